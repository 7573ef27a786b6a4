"""Hashes of Tacotron2.inference outputs on fixed inputs (batch 1-3): run before and after a decoder change (FACPPG_LIB_OVERRIDE=<other .so>)\nto see whether the bits moved."""
import os, sys, hashlib
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
ROOT = os.environ.get("GRAFT_REPO_ROOT", ROOT)
sys.path[:0] = [ROOT, os.path.join(ROOT, "fac-via-ppg_amd")]
import torch
from facppg import lib as _flib
if os.environ.get("FACPPG_LIB_OVERRIDE"): _flib.LIB_PATH = os.environ["FACPPG_LIB_OVERRIDE"]
from common.hparams import create_hparams_stage
from facppg import synth, pipeline
from script.train_ppg2mel import load_model
for Tin, B in ((200, 1), (57, 1), (120, 2), (90, 3)):
    hp = create_hparams_stage(max_decoder_steps=Tin)
    m = load_model(hp); m.load_state_dict(synth.tacotron_state_dict(hp)); m.eval()
    x, lens = pipeline.pad_ppgs([synth.synthetic_ppg(Tin - 7 * i, seed=i) for i in range(B)])
    out = m.inference(x.cuda(), lengths=lens, seed=3) if B > 1 else m.inference(x.cuda(), seed=3)
    torch.cuda.synchronize()
    hs = [hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12] for t in out[:4] if torch.is_tensor(t)]
    print("Tin %d B %d:" % (Tin, B), hs, flush=True)
