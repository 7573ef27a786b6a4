"""CPU: the host logic of the streamed one-utterance path (facppg.pipeline.ConditioningStream) that needs no device: which blocks
of frames are planned for which utterance length / step limit, and when the path is used at all."""
import types

import pytest

from facppg.pipeline import ConditioningStream


def planner(lag=10):
    cs = ConditioningStream.__new__(ConditioningStream)
    cs.lag = lag
    return cs


@pytest.mark.parametrize("steps,Tin", [(200, 200), (64, 64), (75, 75), (170, 170), (400, 130), (1000, 150), (1000, 1000), (64, 30), (8192, 4000)])
def test_block_plan_invariants(steps, Tin, monkeypatch):
    for var in ("FACPPG_STREAM_CHUNK", "FACPPG_STREAM_LAST", "FACPPG_STREAM_PLAN"):
        monkeypatch.delenv(var, raising=False)
    cs = planner()
    cs.cap = min(steps, -(-(Tin + ConditioningStream.SLACK) // 32) * 32)      # what begin() lays the vocoder-side buffers out for
    cuts = cs.plan(steps, Tin)
    assert len(cuts) <= 120
    prev_end = 0
    for f_new, s_a, s_b in cuts:
        assert s_a == prev_end and s_b > s_a and s_a % 32 == 0 and s_b % 32 == 0      # consecutive blocks of whole 32-frame tiles
        assert f_new == s_b + cs.lag                                                   # mel_post[q] is final once frame q + lag exists
        assert f_new <= min(steps, cs.cap)                                             # never waits for a frame past the step limit or the layout
        assert s_b - s_a <= 128                                                        # k_cond_seed takes at most 4 tiles per pass
        prev_end = s_b
    end = min(steps, Tin)
    planned = [c for c in cuts[:len(cuts) - cs.n_extra]]
    if end - cs.lag >= 32:
        assert planned and planned[-1][2] == (end - cs.lag) // 32 * 32                 # everything that can be final before the expected end
        assert planned[-1][2] - planned[-1][1] == 32 or len(planned) == 1              # ... the last block 32 frames
    assert cs.n_extra <= 2


def test_plan_env_overrides(monkeypatch):
    cs = planner()
    monkeypatch.setenv("FACPPG_STREAM_PLAN", "64,32")
    assert [c[1:] for c in cs.plan(200, 200)][:3] == [(0, 64), (64, 96), (96, 128)]
    monkeypatch.delenv("FACPPG_STREAM_PLAN")
    monkeypatch.setenv("FACPPG_STREAM_CHUNK", "64")
    assert [c[1:] for c in cs.plan(200, 200)] == [(0, 64), (64, 128), (128, 160)]


def test_usable_switches(monkeypatch):
    wn = types.SimpleNamespace(n_layers=8)
    wg = types.SimpleNamespace(WN=[wn], n_group=8)
    taco = types.SimpleNamespace(decoder_workgroups=0)
    for var in ("FACPPG_STREAM", "FACPPG_WG_UNFOLDED", "FACPPG_WG_EDGE_FOLD"):
        monkeypatch.delenv(var, raising=False)
    assert ConditioningStream.usable(taco, wg)
    monkeypatch.setenv("FACPPG_STREAM", "0")
    assert not ConditioningStream.usable(taco, wg)
    monkeypatch.delenv("FACPPG_STREAM")
    monkeypatch.setenv("FACPPG_WG_EDGE_FOLD", "0")
    assert not ConditioningStream.usable(taco, wg)           # the seeds are the folded kernels' accumulators
    monkeypatch.delenv("FACPPG_WG_EDGE_FOLD")
    taco.decoder_workgroups = 32                             # a caller that bounds the decoder runs it under something else
    assert not ConditioningStream.usable(taco, wg)
