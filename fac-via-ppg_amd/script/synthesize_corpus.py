"""Offline corpus synthesis sharded data-parallel over the GPUs of one node (BASELINE config 4).

No counterpart in the reference (its only multi-GPU code is WaveGlow training, distributed.py:145-170, whose
one-process-per-GPU launch pattern this follows); it is the utterance-batch scaling path named by
BASELINE.json's north_star.  Utterances are independent, so each rank synthesises a length-balanced shard
(facppg.shard.partition) with its own copy of the weights; the only exchanges are an all_gather of
(id, length) and one padded gather of the audio to rank 0, which writes the wavs.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      -m script.synthesize_corpus --ppg2mel_model taco.pt --waveglow_model wg.pt \
      --ppg_list ppgs.txt --output_dir out/ --batch_size 64
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist
from scipy.io import wavfile

from common.utils import load_filepaths
from facppg import shard

FS = 16000   # generate_synthesis.py:56


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ppg2mel_model', required=True)
    ap.add_argument('--waveglow_model', required=True)
    ap.add_argument('--ppg_list', required=True, help='text file, one precomputed PPG .npy path per line')
    ap.add_argument('--output_dir', required=True)
    ap.add_argument('--batch_size', type=int, default=64,
                    help='utterances per synthesis batch; the latency-bound decoder costs the same for 16 or 96 utterances, so larger '
                         'batches are faster (one MI355X, 384 utterances: 7.8 / 8.2 / 8.7 / 8.9 M samples/s at 16 / 32 / 64 / 96); the '
                         'waveforms do not depend on it')
    ap.add_argument('--sigma', type=float, default=0.6)
    ap.add_argument('--denoiser_strength', type=float, default=0.005)
    ap.add_argument('--seed', type=int, default=0,
                    help='base seed: the dropout / noise streams of utterance i are keyed by seed + 2*i alone, so its wav '
                         'does not depend on the batch size, its place in a batch, or the number of GPUs')
    ap.add_argument('--limit_steps_to_input', action='store_true',
                    help='stop each utterance after as many mel frames as it has PPG frames at the latest '
                         '(per-utterance max_decoder_steps; both run at a 10 ms frame shift)')
    ap.add_argument('--no_overlap', action='store_true',
                    help='run the batches strictly one after the other instead of starting the acoustic model of the next batch '
                         'under the vocoder of the current one (same waveforms, slower)')
    ap.add_argument('--dist_backend', default='nccl', help='nccl = RCCL over xGMI (default); gloo for CPU tests')
    ap.add_argument('--hparams', default='',
                    help='comma separated "name=value" overrides of create_hparams_stage (train_ppg2mel.py:291-292), '
                         'e.g. n_symbols=40 for monophone PPGs (data_utils.py:253-258)')
    return ap.parse_args(argv)


def parse_hparams(text):
    """"a=1,b=x" -> create_hparams_stage(a=1, b='x'); values are cast to the type of the default they override."""
    from common.hparams import create_hparams_stage
    base = create_hparams_stage()
    kw = {}
    for item in filter(None, (t.strip() for t in text.split(','))):
        name, _, value = item.partition('=')
        if not hasattr(base, name):
            raise ValueError("unknown hparam %r" % name)          # same failure as hparams.py:233-237
        default = getattr(base, name)
        kw[name] = (value.lower() in ('1', 'true')) if isinstance(default, bool) else type(default)(value)
    return create_hparams_stage(**kw)


def synthesize_shard(synthesizer, ppgs, lengths, rank, world, args):
    """This rank's utterances, longest first, in batches of similar length.  Returns (waveforms, global ids)."""
    mine = shard.partition(lengths, world)[rank]
    batches = list(shard.batches(mine, lengths, args.batch_size))
    jobs = ({"ppgs": [np.asarray(ppgs[i]) for i in batch], "utterance_seeds": [args.seed + 2 * i for i in batch],
             "step_limits": [lengths[i] for i in batch] if args.limit_steps_to_input else None} for batch in batches)
    wavs, ids = [], []
    if hasattr(synthesizer, "stream"):      # software-pipelined: batch i+1's acoustic model runs under batch i's vocoder
        results = synthesizer.stream(jobs, sigma=args.sigma, strength=args.denoiser_strength, return_device=True,
                                     overlap=not getattr(args, "no_overlap", False))
    else:
        results = (synthesizer(return_device=True, sigma=args.sigma, strength=args.denoiser_strength, **job) for job in jobs)
    for batch, (out, _) in zip(batches, results):
        wavs += out
        ids += batch
    return wavs, ids


def collect(wavs, ids, world):
    """Rank 0: {global id: CPU waveform}; other ranks: None."""
    if world > 1:
        return shard.gather_ragged(wavs, ids, dst=0)
    return {i: w.detach().cpu() for i, w in zip(ids, wavs)}


def write_wavs(output_dir, paths, gathered):
    """<output_dir>/<ppg file stem>.wav, float32 [N, 1] at 16 kHz like the CLI's ac.wav (generate_synthesis.py:97-98)."""
    os.makedirs(output_dir, exist_ok=True)
    written = []
    for i, w in sorted(gathered.items()):
        name = os.path.splitext(os.path.basename(paths[i]))[0] + ".wav"
        wavfile.write(os.path.join(output_dir, name), FS, w.numpy().astype(np.float32)[:, None])
        written.append(name)
    return written


def main(argv=None, synthesizer=None):
    """``synthesizer``: a facppg.pipeline.Synthesizer-like callable; built from the two checkpoints when None
    (the CPU tests of the N>1 data path pass a stand-in, since the HIP path needs a GPU)."""
    args = parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.dist_backend == "nccl"
    if on_gpu:
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    try:
        paths = load_filepaths(args.ppg_list)
        ppgs = [np.load(p, mmap_mode="r") for p in paths]
        lengths = [p.shape[0] for p in ppgs]
        if synthesizer is None:
            from facppg.pipeline import Synthesizer
            synthesizer = Synthesizer(args.ppg2mel_model, args.waveglow_model, hparams=parse_hparams(args.hparams))
        wavs, ids = synthesize_shard(synthesizer, ppgs, lengths, rank, world, args)
        gathered = collect(wavs, ids, world)
        if rank == 0:
            assert sorted(gathered) == list(range(len(paths))), "an utterance was lost or duplicated in the gather"
            return write_wavs(args.output_dir, paths, gathered)
        return None
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
