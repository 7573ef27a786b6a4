import sys, time, os, torch
sys.path.insert(0,'.'); sys.path.insert(0,'fac-via-ppg_amd')
from facppg import synth
from oracle import waveglow as owg
n=int(sys.argv[1]); torch.set_num_threads(n)
cfg=dict(synth.WAVEGLOW_CONFIG, hop_length=256); sd=synth.waveglow_state_dict(cfg)
T=int(sys.argv[2])
mel=synth.synthetic_mel(1,T); zs=synth.synthetic_z(1,T*32,cfg)
with torch.no_grad():
    owg.infer(sd,cfg,mel[:,:,:8],0.6,[z[:,:,:8*32] for z in zs])
    t=time.time(); owg.infer(sd,cfg,mel,0.6,zs); dt=time.time()-t
print('threads',n,'T',T,'time %.2f'%dt,'samples/s %.0f'%(T*256/dt), flush=True)
