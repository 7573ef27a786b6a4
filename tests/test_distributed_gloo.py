"""CPU, world_size 2, gloo: the data-parallel gradient path of WaveGlow training (reference:
src/waveglow/distributed.py) -- flat parameter broadcast, flat-bucket gradient all-reduce fired
from the last gradient hook, reduce_tensor -- on a small stand-in module."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv1d(3, 4, 3, padding=1)
        self.b = torch.nn.Linear(4, 2)
        self.register_buffer("stat", torch.zeros(3))

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)).mean(2))


def _worker(rank, world, port, q):
    from waveglow.distributed import apply_gradient_allreduce, reduce_tensor
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
        m = Tiny()
        m.stat.fill_(float(rank + 1))
        m = apply_gradient_allreduce(m)
        flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()] + [m.stat])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same_params = all(torch.equal(gathered[0], g) for g in gathered) and float(m.stat[0]) == 1.0
        torch.manual_seed(7)
        xs = [torch.randn(5, 3, 8) for _ in range(world)]   # every rank knows every shard
        for step in range(2):                               # two steps: the hook re-arms
            m.zero_grad()
            m(xs[rank]).pow(2).sum().backward()
            got = [p.grad.clone() for p in m.parameters()]
            ref = Tiny()
            ref.load_state_dict(m.state_dict())
            exp = None
            for x in xs:
                ref.zero_grad()
                ref(x).pow(2).sum().backward()
                g = [p.grad.clone() for p in ref.parameters()]
                exp = g if exp is None else [a + b for a, b in zip(exp, g)]
            ok = all(torch.allclose(a, b / world, atol=1e-6) for a, b in zip(got, exp))
            same_params = same_params and ok
        # a parameter that receives no gradient (frozen) and a second backward on a retained graph (accumulation):
        # the exchange must still run once per backward and leave rank-averaged sums
        m.b.bias.requires_grad_(False)
        m.zero_grad()
        out = m(xs[rank]).pow(2).sum()
        out.backward(retain_graph=True)
        g1 = [p.grad.clone() for p in m.parameters() if p.grad is not None]
        out.backward()
        g2 = [p.grad.clone() for p in m.parameters() if p.grad is not None]
        ga = [torch.zeros_like(g) for g in g1]
        for r in range(world):
            ref.zero_grad()
            ref.load_state_dict(m.state_dict())
            ref(xs[r]).pow(2).sum().backward()
            gr = [p.grad for n, p in ref.named_parameters() if n != "b.bias"]
            ga = [a + b / world for a, b in zip(ga, gr)]
        same_params = same_params and all(torch.allclose(a, b, atol=1e-6) for a, b in zip(g1, ga))
        # second pass: grad = averaged(first) + local(second), then averaged again = avg + avg(local)/... every rank equal
        flat2 = torch.cat([g.reshape(-1) for g in g2])
        both = [torch.zeros_like(flat2) for _ in range(world)]
        dist.all_gather(both, flat2)
        same_params = same_params and all(torch.allclose(both[0], b, atol=1e-6) for b in both)
        same_params = same_params and all(torch.allclose(b, 2 * a, atol=1e-5) for a, b in zip(ga, g2))
        loss = reduce_tensor(torch.tensor(float(rank + 1)), world)
        q.put((rank, bool(same_params), float(loss)))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert res == [(0, True, 1.5), (1, True, 1.5)]
