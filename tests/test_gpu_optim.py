"""GPU: waveglow.optim.Adam (one HIP launch over all parameters, facppg_adam_step) vs torch.optim.Adam -- the optimiser of
the reference's training loop (src/script/train_waveglow.py:83,134).  Tolerance: fp32 round-off of a few operations per
step (the kernel follows torch._fused_adam_'s arithmetic; bias corrections in double)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(512, 256, 3), (512,), (1,), (4097,), (7, 13), (256, 640, 1), (3, 3), (1 << 20,)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in SHAPES]


def _grads(step, ps):
    g = torch.Generator().manual_seed(100 + step)
    for p in ps:
        p.grad = (torch.randn(p.shape, generator=g) * (0.5 + step)).cuda()


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_matches_torch_and_shares_state_dicts(wd):
    from waveglow.optim import Adam
    a, b = _params(1), _params(1)
    oa = Adam(a, lr=1e-3, weight_decay=wd)
    ob = torch.optim.Adam(b, lr=1e-3, weight_decay=wd, fused=True)
    for step in range(5):
        _grads(step, a)
        _grads(step, b)
        oa.step()
        ob.step()
    assert oa._hip and not oa._hip_off                              # the kernel path ran
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-7), float((x - y).abs().max())
    for x, y in zip(a, b):
        sa, sb = oa.state[x], ob.state[y]
        assert float(sa["step"]) == float(sb["step"]) == 5.0
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    # checkpoints are interchangeable (train_waveglow.py:45-64 stores optimizer.state_dict()): ours -> torch's and back
    oc = torch.optim.Adam(_params(1), lr=1e-3, weight_decay=wd, fused=True)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))           # (a checkpoint round trip: state_dict() hands out references)
    od = Adam(_params(1), lr=1e-3, weight_decay=wd)
    od.load_state_dict(copy.deepcopy(ob.state_dict()))
    c, d = oc.param_groups[0]["params"], od.param_groups[0]["params"]
    with torch.no_grad():
        for src, x, y in zip(a, c, d):
            x.copy_(src)
            y.copy_(src)
    for step in range(5, 7):
        _grads(step, a); _grads(step, c); _grads(step, d)
        oa.step(); oc.step(); od.step()
    for x, y, z in zip(a, c, d):
        assert torch.allclose(x, y, rtol=2e-6, atol=2e-7) and torch.allclose(x, z, rtol=2e-6, atol=2e-7)
    assert float(od.state[d[0]]["step"]) == 7.0
    steps = [st["step"] for st in oa.state_dict()["state"].values()]
    assert len({t.data_ptr() for t in steps}) == len(steps) and all(float(t) == 7.0 for t in steps)   # torch's layout: no aliasing


def test_adam_in_a_hip_graph_and_fallback():
    from waveglow.optim import Adam
    a, b = _params(2), _params(2)
    oa, ob = Adam(a, lr=1e-3), Adam(b, lr=1e-3)
    for p in a + b:
        p.grad = torch.zeros_like(p)
    ga = [p.grad for p in a]
    def fill(ps, step):
        g = torch.Generator().manual_seed(300 + step)
        for p in ps:
            p.grad.copy_((torch.randn(p.shape, generator=g)).cuda())
    fill(a, 0); fill(b, 0)
    oa.step(); ob.step()                                            # one ordinary step builds the launch plan
    graph = torch.cuda.CUDAGraph()
    fill(a, 1); fill(b, 1)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        oa.step()
    graph.replay()                                                  # (capture does not execute)
    ob.step()
    for step in range(2, 5):
        fill(a, step); fill(b, step)
        graph.replay()
        ob.step()
    assert all(p.grad is g for p, g in zip(a, ga))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert float(oa.state[a[0]]["step"]) == 5.0
    # a parameter without gradient: torch's own implementation takes over, with per-parameter step counts again
    b[2].grad = None
    fill(a, 5)
    for p, q in zip(a, b):
        if q.grad is not None:
            q.grad.copy_(p.grad)
    before = b[2].detach().clone()
    ob.step()
    assert ob._hip_off and torch.equal(b[2], before)
    assert ob.state[b[0]]["step"] is not ob.state[b[1]]["step"] and float(ob.state[b[0]]["step"]) == 6.0
