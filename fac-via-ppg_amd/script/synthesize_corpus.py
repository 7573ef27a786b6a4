"""Offline corpus synthesis sharded data-parallel over the GPUs of one node (BASELINE config 4).

No counterpart in the reference (its only multi-GPU code is WaveGlow training, distributed.py);
this is the utterance-batch scaling path named by BASELINE.json's north_star.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      -m script.synthesize_corpus --ppg2mel_model taco.pt --waveglow_model wg.pt \
      --ppg_list ppgs.txt --output_dir out/ --batch_size 16
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist
from scipy.io import wavfile

from common.hparams import create_hparams_stage
from common.utils import load_filepaths, load_waveglow_model
from facppg import pipeline, shard
from script.train_ppg2mel import load_model
from waveglow.denoiser import Denoiser


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ppg2mel_model', required=True)
    ap.add_argument('--waveglow_model', required=True)
    ap.add_argument('--ppg_list', required=True, help='text file, one precomputed PPG .npy path per line')
    ap.add_argument('--output_dir', required=True)
    ap.add_argument('--batch_size', type=int, default=16)
    ap.add_argument('--sigma', type=float, default=0.6)
    ap.add_argument('--denoiser_strength', type=float, default=0.005)
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    paths = load_filepaths(args.ppg_list)
    ppgs = [np.load(p, mmap_mode="r") for p in paths]
    lengths = [p.shape[0] for p in ppgs]

    hparams = create_hparams_stage()
    taco = load_model(hparams)
    taco.load_state_dict(torch.load(args.ppg2mel_model, weights_only=False)['state_dict'])
    taco.eval()
    denoiser = Denoiser(torch.load(args.waveglow_model, weights_only=False)['model'].cuda(), mode='zeros')
    waveglow = load_waveglow_model(args.waveglow_model)

    mine = shard.partition(lengths, world)[rank]
    wavs, ids = [], []
    for batch in shard.batches(mine, lengths, args.batch_size):
        out, _ = pipeline.synthesize([np.asarray(ppgs[i]) for i in batch], taco, waveglow, denoiser, args.sigma,
                                     args.denoiser_strength, return_device=True)
        wavs += out
        ids += batch
    gathered = shard.gather_ragged(wavs, ids, dst=0) if world > 1 else {i: w.cpu() for i, w in zip(ids, wavs)}
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        for i, w in sorted(gathered.items()):
            name = os.path.splitext(os.path.basename(paths[i]))[0] + ".wav"
            wavfile.write(os.path.join(args.output_dir, name), 16000, w.numpy().astype(np.float32)[:, None])
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
