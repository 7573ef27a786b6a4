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


class FlowNamed(torch.nn.Module):
    """Parameters named like WaveGlow's (WN.<k>.*, convinv.<k>.*, upsample.*): 4 'flows', used in flow order."""

    def __init__(self):
        super().__init__()
        self.upsample = torch.nn.Linear(3, 3)
        self.WN = torch.nn.ModuleList([torch.nn.Linear(3, 3) for _ in range(4)])
        self.convinv = torch.nn.ModuleList([torch.nn.Linear(3, 3, bias=False) for _ in range(4)])

    def forward(self, x):
        x = self.upsample(x)
        for k in range(4):
            x = torch.tanh(self.WN[k](self.convinv[k](x)))
        return x


def test_bucket_plan_follows_backward_order():
    """Buckets hold whole flows, last flow first (its gradients are finished first), the upsampler in the last bucket."""
    from waveglow.distributed import plan_buckets
    m = FlowNamed()
    names = {id(p): n for n, p in m.named_parameters()}
    b = [[names[id(p)] for p in bucket] for bucket in plan_buckets(list(m.named_parameters()), 2)]
    assert sorted(b[0]) == sorted(["WN.2.weight", "WN.2.bias", "WN.3.weight", "WN.3.bias", "convinv.2.weight", "convinv.3.weight"])
    assert sorted(b[1]) == sorted(["WN.0.weight", "WN.0.bias", "WN.1.weight", "WN.1.bias", "convinv.0.weight", "convinv.1.weight",
                                   "upsample.weight", "upsample.bias"])
    t = Tiny()       # no flow naming: split by size over the reversed registration order
    tb = plan_buckets(list(t.named_parameters()), 2)
    assert sum(len(x) for x in tb) == 4 and tb[0][0] is t.b.bias


def _exchange_worker(rank, world, port, q, grad_dtype):
    from waveglow.distributed import GradientExchange, broadcast_parameters
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(5 + rank)
        m = FlowNamed()
        broadcast_parameters(m, 0)
        dt = {"fp32": None, "bf16": torch.bfloat16}[grad_dtype]
        tol = 1e-6 if dt is None else 2e-2
        torch.manual_seed(9)
        xs = [torch.randn(6, 3) for _ in range(world)]
        ref = FlowNamed()
        ref.load_state_dict(m.state_dict())
        exp = None
        for x in xs:
            ref.zero_grad()
            ref(x).pow(2).sum().backward()
            g = [p.grad.clone() / world for p in ref.parameters()]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        ok = True
        # (1) eager hooks: buckets launched from gradient hooks, finished by the engine callback; two steps
        ex = GradientExchange(m, n_buckets=2, grad_dtype=dt).install_hooks()
        for step in range(2):
            m.zero_grad()
            m(xs[rank]).pow(2).sum().backward()
            ok = ok and all(torch.allclose(p.grad, e, atol=tol, rtol=tol) for p, e in zip(m.parameters(), exp))
            if dt is None:      # fp32: the gradients ARE slices of the flat buckets now (no copy back)
                ok = ok and all(p.grad.data_ptr() == v.data_ptr() for b, vs in zip(ex.buckets, ex.views) for p, v in zip(b, vs))
        # (2) the graphed-step protocol: gradients are produced into fixed tensors ("the graph's outputs"), bound once,
        # exchange(static=True) after every "replay"; .grad are bucket views afterwards, the sources stay untouched
        m2 = FlowNamed()
        m2.load_state_dict(ref.state_dict())
        ex2 = GradientExchange(m2, n_buckets=3, grad_dtype=dt)
        m2(xs[rank]).pow(2).sum().backward()
        ex2.bind_static_sources()
        static = [p.grad for p in m2.parameters()]
        local = [g.clone() for g in static]
        for replay in range(2):
            for s, l in zip(static, local):
                s.copy_(l)                                  # what a replay does: rewrite the same tensors
            ex2.exchange(static=True)
            ok = ok and all(torch.allclose(p.grad, e, atol=tol, rtol=tol) for p, e in zip(m2.parameters(), exp))
            if dt is None:
                ok = ok and all(p.grad.data_ptr() != s.data_ptr() for p, s in zip(m2.parameters(), static))
                ok = ok and all(torch.equal(s, l) for s, l in zip(static, local))
        ok = ok and ex2.bytes_per_exchange() == sum(p.numel() for p in m2.parameters()) * (4 if dt is None else 2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_world2(target, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return sorted(q.get(timeout=10) for _ in range(2))


def test_bucketed_exchange_world2_gloo_fp32():
    assert _run_world2(_exchange_worker, "fp32") == [(0, True), (1, True)]


def test_bucketed_exchange_world2_gloo_bf16_links():
    assert _run_world2(_exchange_worker, "bf16") == [(0, True), (1, True)]


def test_launcher_starts_one_process_per_rank(tmp_path):
    """waveglow.distributed.main (distributed.py:145-170): N processes, rank i gets --rank=i and the shared --group_name,
    --num_gpus=N and --config; ranks > 0 log to GPU_<i>.log; a failing rank takes the others down."""
    import sys
    from waveglow import distributed
    mod = tmp_path / "fake_train.py"
    mod.write_text("import sys, time\nprint('ARGS', ' '.join(sys.argv[1:]), flush=True)\n"
                   "if '--boom' in sys.argv and '--rank=1' in sys.argv:\n    sys.exit(3)\n"
                   "if '--boom' in sys.argv:\n    time.sleep(60)\n")
    old = os.environ.get("PYTHONPATH", "")
    os.environ["PYTHONPATH"] = str(tmp_path) + os.pathsep + old
    try:
        codes = distributed.main("cfg.json", str(tmp_path / "logs"), "--epochs 1", num_gpus=3, module="fake_train")
        assert codes == [0, 0, 0]
        for i in (1, 2):
            line = (tmp_path / "logs" / ("GPU_%d.log" % i)).read_text()
            assert "--rank=%d" % i in line and "--num_gpus=3" in line and "--config=cfg.json" in line and "--group_name=group_" in line
            assert "--epochs 1" in line
        import time
        t0 = time.time()
        codes = distributed.main("cfg.json", str(tmp_path / "logs"), "--boom", num_gpus=3, module="fake_train")
        assert codes[1] == 3 and all(c != 0 for c in codes) and time.time() - t0 < 30
    finally:
        os.environ["PYTHONPATH"] = old
