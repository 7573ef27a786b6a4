"""CPU, world_size 2, gloo: the N>1 data path of offline corpus synthesis -- length-balanced
partition, flat weight broadcast, and the ragged gather of audio to rank 0 -- with a stand-in
synthesiser (the HIP path itself needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from facppg import shard


def test_partition_covers_every_utterance_once_and_balances():
    g = np.random.Generator(np.random.PCG64(11))
    lengths = (100 + g.integers(0, 301, size=1024)).tolist()
    parts = shard.partition(lengths, 8)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(1024))
    sums = [sum(lengths[i] for i in p) for p in parts]
    assert max(sums) - min(sums) <= max(lengths)           # round-robin on sorted lengths
    assert all(len(p) == 128 for p in parts)
    for p in parts:                                         # each shard stays length-sorted
        assert [lengths[i] for i in p] == sorted((lengths[i] for i in p), reverse=True)
    assert shard.partition([], 4) == [[], [], [], []]
    assert shard.partition([5], 4) == [[0], [], [], []]     # fewer utterances than ranks
    assert shard.batches(list(range(5)), None, 2) == [[0, 1], [2, 3], [4]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns the weights; everyone else receives them in one flat broadcast
        w = [torch.arange(6, dtype=torch.float32).view(2, 3) * (1 if rank == 0 else 0), torch.ones(4) * (rank == 0)]
        w = shard.broadcast_state(w, src=0)
        assert torch.equal(w[0], torch.arange(6, dtype=torch.float32).view(2, 3)) and w[1].sum() == 4
        mine = shard.partition(lengths, world)[rank]
        # stand-in synthesiser: utterance i -> ramp of its own length tagged with its id
        wavs = [torch.arange(lengths[i], dtype=torch.float32) + 1000 * i for i in mine]
        got = shard.gather_ragged(wavs, mine, dst=0)
        if rank == 0:
            ok = sorted(got) == list(range(len(lengths))) and all(
                torch.equal(got[i], torch.arange(lengths[i], dtype=torch.float32) + 1000 * i) for i in got)
            q.put(ok)
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lengths", [[7, 3, 12, 5, 9], [4], [6, 6, 6, 6]])
def test_gather_ragged_world2_gloo(lengths):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_pad_ppgs_layout_host_and_device_paths_agree():
    """pad_ppgs: [Tin_i, D] frames -> channel-major [B, D, Tmax] with zero tails; the path that uploads
    time-major and transposes on the device (here device='cpu') must give the same tensor as the host path."""
    import numpy as np
    import torch
    from facppg import pipeline
    g = np.random.Generator(np.random.PCG64(3))
    ppgs = [g.random((n, 7), dtype=np.float32) for n in (5, 1, 9)]
    x, lens = pipeline.pad_ppgs(ppgs)
    y, lens2 = pipeline.pad_ppgs(ppgs, device="cpu")
    assert lens == lens2 == [5, 1, 9] and x.shape == (3, 7, 9) and x.dtype == torch.float32
    assert torch.equal(x, y)
    for b, p in enumerate(ppgs):
        assert np.array_equal(x[b, :, :lens[b]].numpy(), p.T) and torch.count_nonzero(x[b, :, lens[b]:]) == 0


class _RampSynthesizer(object):
    """Stand-in for facppg.pipeline.Synthesizer (the HIP path needs a GPU): utterance -> a ramp determined by its
    seed and its PPG (length and first value), on the CPU.  Same call signature as the real one."""

    def __call__(self, ppgs, sigma=0.6, strength=0.005, utterance_seeds=None, step_limits=None, return_device=False):
        assert return_device and utterance_seeds is not None and len(utterance_seeds) == len(ppgs)
        lens = [int(p.shape[0]) for p in ppgs]
        assert lens == sorted(lens, reverse=True)                  # a rank's batches arrive longest first
        tout = list(step_limits) if step_limits is not None else lens
        wavs = [torch.arange(t * 160, dtype=torch.float32) * 1e-3 + float(s) + float(p[0, 0]) for t, s, p in zip(tout, utterance_seeds, ppgs)]
        return wavs, tout

    streamed = 0

    def stream(self, jobs, sigma=0.6, strength=0.005, return_device=True, overlap=True):
        """facppg.pipeline.Synthesizer.stream: one result per job, in order, pulled lazily (script.synthesize_corpus feeds it a
        generator of batches and zips the results with them)."""
        assert return_device and overlap
        for job in jobs:
            self.streamed += 1
            yield self(sigma=sigma, strength=strength, return_device=True, **job)


def _corpus_worker(rank, world, port, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from script import synthesize_corpus
    syn = _RampSynthesizer()
    written = synthesize_corpus.main(argv, synthesizer=syn)
    assert (written is not None) == (rank == 0)
    assert syn.streamed >= 1, "the corpus script did not go through the synthesizer's pipelined stream"


@pytest.mark.parametrize("world", [1, 2])
def test_synthesize_corpus_script_gloo(tmp_path, world):
    """BASELINE config 4's N>1 data path, driven through script.synthesize_corpus.main itself on CPU (gloo): its own
    argument parsing, partition, batching, per-utterance seeds / step limits, ragged gather to rank 0 and WAV writer;
    only the synthesiser is a stand-in.  The files must not depend on the world size."""
    from scipy.io import wavfile
    g = np.random.Generator(np.random.PCG64(11))
    lens = [max(2, int(v) // 40) for v in (100 + g.integers(0, 301, size=37))]       # config-4 length law, scaled down
    paths = []
    for i, n in enumerate(lens):
        paths.append(str(tmp_path / ("u%02d.npy" % i)))
        np.save(paths[-1], np.full((n, 3), i / 64.0, dtype=np.float32))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    out = tmp_path / ("out%d" % world)
    argv = ["--ppg2mel_model", "unused", "--waveglow_model", "unused", "--ppg_list", str(tmp_path / "list.txt"),
            "--output_dir", str(out), "--batch_size", "4", "--seed", "5", "--limit_steps_to_input", "--dist_backend", "gloo"]
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_corpus_worker, args=(r, world, port, argv)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(os.listdir(out)) == ["u%02d.wav" % i for i in range(len(lens))]
    for i, n in enumerate(lens):
        sr, a = wavfile.read(out / ("u%02d.wav" % i))
        assert sr == 16000 and a.dtype == np.float32 and a.shape == (n * 160,)
        assert np.array_equal(a, (np.arange(n * 160, dtype=np.float32) * np.float32(1e-3) + np.float32(5 + 2 * i) + np.float32(i / 64.0)))
