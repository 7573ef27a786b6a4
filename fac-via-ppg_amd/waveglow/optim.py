"""Adam for the WaveGlow training loop as ONE HIP launch over all parameters (facppg_adam_step).

Drop-in for the ``torch.optim.Adam(model.parameters(), lr=learning_rate)`` of the reference's training script
(src/script/train_waveglow.py:83, stepped at :134): it IS a ``torch.optim.Adam`` -- same constructor arguments, same
per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` / ``load_state_dict()`` and the
checkpoints built from them (train_waveglow.py:45-64) are interchangeable with torch's -- whose ``step()`` is replaced:
torch's fused multi-tensor Adam walks WaveGlow's 938 parameters in 27 launches at ~1.9 TB/s (1.27 ms of a 13 ms step);
the HIP kernel streams a pointer table in one launch at the HBM rate.  The launch can be captured in a HIP graph (the
step count lives on the device and is advanced by the launch itself).

Falls back to torch's own implementation -- permanently, for that optimiser -- when a step cannot use the kernel (a
parameter without gradient, non-fp32 or non-contiguous tensors, amsgrad / maximize / decoupled weight decay).
"""
import numpy as np
import torch

from facppg import lib as _lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.setdefault("fused", True)
        kw.setdefault("capturable", True)          # step counts are device tensors (what the kernel advances)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._hip = {}                              # group index -> cached launch plan
        self._hip_off = False

    # ---- launch plan of one parameter group
    def _plan(self, gi, group):
        params = group["params"]
        # keyed on the gradients' ADDRESSES (what the table holds): id(p.grad) can be reused by a new tensor elsewhere
        key = tuple(p.grad.data_ptr() for p in params)
        plan = self._hip.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and plan is None:
            raise RuntimeError("waveglow.optim.Adam: take one ordinary step before capturing (the launch plan uploads tables)")
        L = _lib.load()
        dev = params[0].device
        states = [self.state[p] for p in params]
        if plan is None:
            # one shared device step count for the group (torch keeps one per parameter, all equal): every state's
            # `step` becomes the same tensor, which is what state_dict() then serialises
            step = states[0]["step"].detach().to(device=dev, dtype=torch.float32).reshape(()).clone()
            for st in states:
                st["step"] = step
            chunk = L.facppg_adam_chunk_elems()
            counts = [-(-p.numel() // chunk) for p in params]
            chunks = np.empty((sum(counts), 2), dtype=np.int32)
            o = 0
            for i, c in enumerate(counts):
                chunks[o:o + c, 0] = i
                chunks[o:o + c, 1] = np.arange(c)
                o += c
            base = torch.empty(len(params), 5, dtype=torch.int64)
            for i, (p, st) in enumerate(zip(params, states)):
                base[i, 0], base[i, 1], base[i, 2], base[i, 3], base[i, 4] = p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            plan = {"step": step, "chunks": torch.from_numpy(chunks).to(dev), "n_chunks": int(chunks.shape[0]), "base": base,
                    "slots": [], "next": 0, "table": torch.empty(len(params), 5, dtype=torch.int64, device=dev)}
            self._hip[gi] = plan
        # The table is uploaded from PINNED host memory with an asynchronous copy, so a host buffer must not be rewritten
        # while an earlier upload may still be reading it (a caller that does not synchronise every step: eager steps with
        # zero_grad(set_to_none=True) change the gradient addresses every step), and never once a captured graph has baked
        # a copy node that re-reads it at every replay.  Eager steps rotate two staging buffers, each guarded by an event
        # recorded behind its last upload; a capture takes a buffer of its own (allocated with the plan, i.e. before any
        # capture: nothing may be synchronised and nothing should be allocated while a stream is capturing), which then
        # belongs to that graph for good.
        def staging():
            return torch.empty(len(params), 5, dtype=torch.int64).pin_memory()
        if "slots" not in plan or not plan["slots"]:
            plan["slots"] = [{"host": staging(), "event": None}, {"host": staging(), "event": None}]
            plan["capture_host"] = staging()
        if capturing:
            h = plan["capture_host"] if plan["capture_host"] is not None else staging()
            plan["capture_host"] = None
            plan.setdefault("captured_hosts", []).append(h)     # the graph's copy node reads this buffer at every replay
        else:
            slot = plan["slots"][plan["next"] % 2]
            plan["next"] += 1
            if slot["event"] is not None:
                slot["event"].synchronize()
            h = slot["host"]
        h.copy_(plan["base"])
        h[:, 1] = torch.tensor(key, dtype=torch.int64)
        plan["table"].copy_(h, non_blocking=True)
        if not capturing:
            slot["event"] = torch.cuda.Event()
            slot["event"].record(torch.cuda.current_stream(dev))
            if plan["capture_host"] is None:
                plan["capture_host"] = staging()                # for the next capture
        plan["key"] = key
        return plan

    def _usable(self, group):
        if group["amsgrad"] or group["maximize"] or group.get("decoupled_weight_decay") or group.get("differentiable"):
            return False
        if isinstance(group["lr"], torch.Tensor):
            return False
        for p in group["params"]:
            g = p.grad
            if g is None or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() \
                    or not g.is_contiguous() or g.is_sparse:
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        """train_waveglow.py:134."""
        if self._hip_off or closure is not None or not all(self._usable(g) for g in self.param_groups):
            if self._hip:                                   # un-share the step counts before handing over to torch
                for group in self.param_groups:
                    for p in group["params"]:
                        if p in self.state:
                            self.state[p]["step"] = self.state[p]["step"].clone()
                self._hip = {}
            self._hip_off = True
            return super().step(closure)
        L = _lib.load()
        for gi, group in enumerate(self.param_groups):
            if any(len(self.state[p]) == 0 for p in group["params"]):
                self._init_group(group, [], [], [], [], [], [])        # torch's own lazy state creation
            plan = self._plan(gi, group)
            dev = group["params"][0].device
            beta1, beta2 = group["betas"]
            with torch.cuda.device(dev):
                _lib.check(L.facppg_adam_step(_lib.ptr(plan["table"]), len(group["params"]), _lib.ptr(plan["chunks"]), plan["n_chunks"],
                                              _lib.ptr(plan["step"]), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                              float(group["weight_decay"]), _lib.current_stream(dev)))
        return None

    def state_dict(self):
        """torch.optim.Adam's layout exactly: one `step` tensor PER parameter (the shared device counter is cloned per entry),
        so a checkpoint written from here behaves in torch's own Adam as one written by it -- torch's multi-tensor step
        increments every `step` entry it finds, which must therefore not alias each other."""
        sd = super().state_dict()
        # (the per-parameter dicts super() returns ARE the live state: build new ones instead of editing them)
        sd["state"] = {k: (dict(st, step=st["step"].detach().clone()) if torch.is_tensor(st.get("step")) else st)
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._hip = {}                                      # states were replaced: rebuild the plan (and re-share the step)
