import os, sys
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"fac-via-ppg_amd")]
import numpy as np, torch
from facppg import synth
from waveglow.glow import WN
cfg=dict(synth.WAVEGLOW_CONFIG); sd=synth.waveglow_state_dict(cfg)
B=int(sys.argv[1]) if len(sys.argv)>1 else 12
wn=WN(4,640,**cfg["WN_config"])
torch.nn.utils.remove_weight_norm(wn.start)
for lst in (wn.in_layers,wn.cond_layers,wn.res_skip_layers):
    for conv in lst: torch.nn.utils.remove_weight_norm(conv)
wn.load_state_dict({k[len("WN.0."):]:v for k,v in sd.items() if k.startswith("WN.0.")},strict=True)
wn.train_precision="bf16"; wn=wn.cuda()
a=torch.randn(B,4,1250,device="cuda").requires_grad_(True); s=torch.randn(B,640,1250,device="cuda").requires_grad_(True)
for _ in range(2): wn((a,s)).sum().backward()
torch.cuda.synchronize()
path="/tmp/wg_stamps.txt"
if os.path.exists(path): os.remove(path)
os.environ["FACPPG_WGRAD_STAMPS"]=path
wn((a,s)).sum().backward(); torch.cuda.synchronize()
cur=None; L=[]
for line in open(path):
    if line.startswith("launch"): cur=[]; L.append((line.strip(),cur))
    else: cur.append([int(x) for x in line.split()])
for hdr,rows in L:
    r=np.array(rows,dtype=np.float64); nch=r[:,23]; r=r[:,:23]*0.01
    ok=r[:,2]>0; r=r[ok]; nch=nch[ok]
    d=np.diff(r[:,:22],axis=1)
    print(hdr, "wgs", len(r), "chunks/wg", np.median(nch), "| first stamp->chunk times (us, median):", " ".join("%.2f"%x for x in np.median(d,axis=0)[:12]), "| loop total med %.1f" % np.median(r[:,22]))
