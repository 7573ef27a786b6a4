"""Data-parallel gradient synchronisation for WaveGlow training -- drop-in for the reference's
src/waveglow/distributed.py (its only multi-GPU strategy: data parallelism, one process per GPU).

Same entry points (``init_distributed``, ``reduce_tensor``, ``apply_gradient_allreduce``, the one-process-per-GPU
launcher ``main(config, stdout_dir, args_str)``) with the exchange re-planned for one node of 8 MI355X on xGMI through
``torch.distributed`` (backend "nccl" IS RCCL on ROCm; "gloo" in the CPU tests):

  * parameters: ONE flat broadcast from rank 0 (the reference issues one broadcast per state-dict
    tensor, 938 of them, distributed.py:100-103);
  * gradients (``GradientExchange``): the parameters are dealt into a few BUCKETS in the order the backward pass
    finishes them (WaveGlow: flows 11..8, 7..4, 3..0 + the upsampler).  Every bucket owns a persistent flat buffer;
    when the last gradient of a bucket has been produced it is packed into the buffer (one multi-tensor copy) and
    all-reduced ASYNCHRONOUSLY on a side stream while the backward pass goes on with the earlier flows
    (distributed.py:105-129 reduces one flat bucket per dtype after the whole pass).  After the reduction the
    parameters' ``.grad`` ARE views of the flat buffers: there is no copy back, and the optimiser reads them in place.
    xGMI is point to point (7 links x ~153 GB/s per GPU): RCCL turns a large all_reduce into reduce-scatter +
    all-gather over all links, 2*(7/8)*S per GPU, so a few ~100 MB buckets keep every link busy without paying a
    latency floor per tensor.  ``grad_dtype=torch.bfloat16`` halves the bytes on the links (175.8 MB for WaveGlow):
    the bucket is packed / reduced in bf16 and unpacked into fp32 gradients.
"""
import argparse
import os
import re
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def reduce_tensor(tensor, num_gpus):
    """Mean of a (scalar) tensor over ranks, for logging (distributed.py:37-41)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def init_distributed(rank, num_gpus, group_name, dist_backend, dist_url):
    """distributed.py:43-53; one process per GPU, device = rank % device_count."""
    assert torch.cuda.is_available(), "Distributed mode requires a GPU."
    print("Initializing Distributed")
    torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


def _flatten_dense_tensors(tensors):
    if len(tensors) == 1:
        return tensors[0].contiguous().view(-1)
    return torch.cat([t.contiguous().view(-1) for t in tensors], dim=0)


def _unflatten_dense_tensors(flat, tensors):
    out, off = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat.narrow(0, off, n).view_as(t))
        off += n
    return tuple(out)


def broadcast_parameters(module, src=0):
    """One flat broadcast per dtype of every state-dict tensor (parameters and buffers)."""
    by_dtype = {}
    for t in module.state_dict().values():
        if torch.is_tensor(t) and t.numel() > 0:
            by_dtype.setdefault(t.dtype, []).append(t)
    for tensors in by_dtype.values():
        flat = _flatten_dense_tensors([t.detach() for t in tensors])
        _collective(dist.broadcast, flat, src)
        for t, synced in zip(tensors, _unflatten_dense_tensors(flat, tensors)):
            t.detach().copy_(synced)


def _collective(fn, flat, *args, **kw):
    """``fn(flat, ...)``; a GPU tensor under the gloo backend (single-GPU dry runs of the N > 1 path) is staged
    through the host, since not every gloo build reduces device memory."""
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.cpu()
        fn(host, *args)
        flat.copy_(host)
        return None
    return fn(flat, *args, **kw)


def allreduce_gradients(module):
    """Average the gradients over ranks with one flat all_reduce per dtype (the reference's plan,
    distributed.py:105-129).  Kept for callers that want the exchange as one blocking call; training uses
    GradientExchange."""
    buckets = {}
    for p in module.parameters():
        if p.requires_grad and p.grad is not None:
            buckets.setdefault(p.grad.dtype, []).append(p.grad)
    world = dist.get_world_size()
    for grads in buckets.values():
        flat = _flatten_dense_tensors([g.detach() for g in grads])
        _collective(dist.all_reduce, flat)
        flat /= world
        for g, synced in zip(grads, _unflatten_dense_tensors(flat, grads)):
            g.detach().copy_(synced)


def _flow_of(name):
    m = re.match(r"(?:WN|convinv)\.(\d+)\.", name)
    return int(m.group(1)) if m else None


def plan_buckets(named_params, n_buckets):
    """[(name, parameter)] -> list of buckets (lists of parameters), in the order a backward pass completes them.
    WaveGlow's training direction runs flows 0..n-1 (glow.py:226-247), so its backward finishes flow n-1 first:
    parameters named WN.<k>.* / convinv.<k>.* go to bucket (n-1-k)*n_buckets // n, everything else (the upsampler,
    whose gradient needs every flow's conditioning gradient) to the last one.  Modules without that naming are split
    by cumulative size over the reversed registration order."""
    named_params = [(n, p) for n, p in named_params if p.requires_grad]
    flows = [f for f in (_flow_of(n) for n, _ in named_params) if f is not None]
    buckets = [[] for _ in range(max(1, n_buckets))]
    if flows:
        nf = max(flows) + 1
        for n, p in named_params:
            k = _flow_of(n)
            buckets[len(buckets) - 1 if k is None else (nf - 1 - k) * len(buckets) // nf].append(p)
    else:
        rev = list(reversed(named_params))
        total, run = sum(p.numel() for _, p in rev), 0
        for _, p in rev:
            buckets[min(len(buckets) - 1, run * len(buckets) // max(1, total))].append(p)
            run += p.numel()
    return [b for b in buckets if b]


class GradientExchange(object):
    """Bucketed, overlapped gradient averaging (see the module docstring).

    eager steps:   ``install_hooks()`` once; every ``backward()`` then leaves rank-averaged gradients behind.
    graphed steps: (a) OVERLAPPED (waveglow.graphed, the default on RCCL): the hooks stay installed while the step is
                   captured, so the pack + all_reduce of every bucket become nodes of the replayed graph on a forked
                   branch -- bucket i is on the links while the backward pass of the earlier flows runs, exactly as in
                   an eager step, and the optimiser step follows in the same graph;
                   (b) SERIAL (fallback when the collectives cannot be captured, and the gloo tests): the captured
                   graph produces the gradients; ``bind_static_sources()`` once after the capture, then
                   ``exchange(static=True)`` after every replay (buckets are packed and reduced in a pipeline: bucket
                   i+1 is packed while bucket i is on the links)."""

    def __init__(self, module, n_buckets=3, grad_dtype=None):
        self.module = module
        self.world = dist.get_world_size()
        self.buckets = plan_buckets(list(module.named_parameters()), n_buckets)
        self.params = [p for b in self.buckets for p in b]
        dev = self.params[0].device
        self.cuda = dev.type == "cuda"
        self.comm_dtype = grad_dtype or torch.float32
        self.flat, self.views = [], []
        for b in self.buckets:
            flat = torch.zeros(sum(p.numel() for p in b), dtype=self.comm_dtype, device=dev)
            self.flat.append(flat)
            self.views.append(list(_unflatten_dense_tensors(flat, b)))
        self.bucket_of = {id(p): (i, j) for i, b in enumerate(self.buckets) for j, p in enumerate(b)}
        self.comm_stream = torch.cuda.Stream(device=dev) if self.cuda else None
        self.pending = [None] * len(self.buckets)       # async work handles (or True once launched synchronously)
        self.arrived = [0] * len(self.buckets)
        self.sources = None
        self.queued = False
        self.last_exchange_ms = None
        self.hooked = False
        self._hook_handles = []
        self.capture_group = None                       # see prepare_capture_group

    def prepare_capture_group(self):
        """A process group of its own for the collectives that are CAPTURED into a HIP graph (collective call: every rank, before
        its first capture).  c10d's watchdog thread polls the end events of the eager collectives it still holds; on this HIP
        stack an event query fails -- and the watchdog aborts the process -- once the event's stream has joined a capture
        ("operation not permitted on an event last recorded in a capturing stream"), whether or not that record was captured.
        The stream in question is the process group's own: so the captured collectives get a group whose stream is the only one
        that ever joins a capture and that never runs an eager collective (works issued under capture are not handed to the
        watchdog), while the default group -- warm-up steps, broadcasts, fallback steps -- never sees a capture.  With the default
        group bound to its device (init_process_group(device_id=...)) the new communicator is split off and connected eagerly,
        without a collective; otherwise one eager all_reduce initialises it here and the watchdog is given time to retire it."""
        if self.capture_group is not None or not self.cuda or dist.get_backend() == "gloo":
            return self.capture_group
        self.capture_group = dist.new_group()
        if not getattr(dist.distributed_c10d._get_default_group(), "bound_device_id", None):
            import time
            probe = torch.zeros(1, device=self.flat[0].device)
            dist.all_reduce(probe, group=self.capture_group)
            torch.cuda.synchronize()
            time.sleep(0.5)
        return self.capture_group

    # ---- one bucket
    def _grads(self, i, static):
        if static and self.sources is not None:
            return self.sources[i]
        return [p.grad for p in self.buckets[i]]

    def _launch(self, i, static=False):
        """Pack bucket i into its flat buffer and start its all_reduce -- both on the communication stream, which waits
        for the compute stream first: the compute stream itself is never held up."""
        if self.pending[i] is not None:
            return
        grads, views = self._grads(i, static), self.views[i]
        src = [g for g, v in zip(grads, views) if g is not None and g.data_ptr() != v.data_ptr()]
        dst = [v for g, v in zip(grads, views) if g is not None and g.data_ptr() != v.data_ptr()]
        flat = self.flat[i]

        def pack():
            for g, v in zip(grads, views):
                if g is None:
                    v.zero_()                               # a parameter without a gradient adds nothing
            if src:
                torch._foreach_copy_(dst, src)              # one multi-tensor launch; converts when the links carry bf16

        if self.cuda and dist.get_backend() != "gloo":
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                pack()
                group = self.capture_group if torch.cuda.is_current_stream_capturing() else None
                self.pending[i] = dist.all_reduce(flat, async_op=True, group=group)
        else:
            pack()
            _collective(dist.all_reduce, flat)
            self.pending[i] = True

    def _finish(self, i, static=False):
        work = self.pending[i]
        if work is None:
            return
        if work is not True:
            work.wait()                                     # orders the current stream behind the collective
            torch.cuda.current_stream().wait_stream(self.comm_stream)   # (and behind the pack, for the allocator's sake)
        flat = self.flat[i]
        flat.div_(self.world)
        grads = self._grads(i, static)
        if self.comm_dtype == torch.float32:
            for p, g, v in zip(self.buckets[i], grads, self.views[i]):
                if g is not None:
                    p.grad = v                              # the gradient IS the bucket slice from here on: no copy back
        else:
            live = [(g, v) for g, v in zip(grads, self.views[i]) if g is not None]
            if live:
                torch._foreach_copy_([g for g, _ in live], [v for _, v in live])
        self.pending[i] = None
        self.arrived[i] = 0

    # ---- whole exchange
    def exchange(self, static=False):
        """Average every bucket now (blocking with respect to the current stream).  static=True packs from the
        gradient tensors bound with bind_static_sources() (a replayed graph's outputs)."""
        t0 = t1 = None
        if self.cuda:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        for i in range(len(self.buckets)):
            self._launch(i, static)
        for i in range(len(self.buckets)):
            self._finish(i, static)
        if self.cuda:
            t1.record()
            self._timing = (t0, t1)

    def exchange_ms(self):
        """Device time of the last exchange() on the calling stream (pack + all_reduce + scale), in ms."""
        if not getattr(self, "_timing", None):
            return None
        t0, t1 = self._timing
        t1.synchronize()
        return t0.elapsed_time(t1)

    def bind_static_sources(self):
        """Remember the CURRENT ``.grad`` tensors as the sources to pack from (call once, right after the training
        step has been captured: they are the graph's outputs and are rewritten in place by every replay)."""
        self.sources = [[p.grad for p in b] for b in self.buckets]

    def install_hooks(self):
        """Eager training: a bucket is launched as soon as its last gradient has been accumulated, the rest
        (buckets a frozen / unused parameter kept incomplete) and the waits run from ONE autograd-engine callback after
        the whole pass -- so the exchange happens exactly once per ``backward()`` whatever the set of parameters that
        received gradients (frozen parameters, a second backward on a retained graph and gradient accumulation all
        stay in step across ranks; ranks must agree on the set of parameters with gradients, as in the reference)."""
        def finish_all():
            self.queued = False
            for i in range(len(self.buckets)):
                self._launch(i)
            for i in range(len(self.buckets)):
                self._finish(i)

        def on_grad(param):
            if not self.queued:
                self.queued = True
                torch.autograd.Variable._execution_engine.queue_callback(finish_all)
            i, _ = self.bucket_of[id(param)]
            self.arrived[i] += 1
            # buckets complete in order; launching bucket i only once every earlier one is on its way keeps the
            # collectives in the same sequence on every rank
            if self.arrived[i] == len(self.buckets[i]) and all(self.pending[j] is not None for j in range(i)):
                self._launch(i)

        if self.hooked:
            return self
        self._hook_handles = [p.register_post_accumulate_grad_hook(on_grad) for p in self.params]
        self.hooked = True
        return self

    def remove_hooks(self):
        """Back to explicit ``exchange()`` calls (a graphed step whose capture could not hold the collectives)."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles, self.hooked, self.queued = [], False, False
        self.pending = [None] * len(self.buckets)
        self.arrived = [0] * len(self.buckets)

    def bytes_per_exchange(self):
        return sum(f.numel() * f.element_size() for f in self.flat)


def apply_gradient_allreduce(module, n_buckets=3, grad_dtype=None):
    """Make ``loss.backward()`` on ``module`` leave rank-averaged gradients behind, without changing
    the module's class (distributed.py:90-142): flat parameter broadcast, then a hooked GradientExchange."""
    broadcast_parameters(module, 0)
    ex = GradientExchange(module, n_buckets=n_buckets, grad_dtype=grad_dtype).install_hooks()
    module.gradient_exchange = ex
    module.allreduce_params = lambda: ex.exchange()
    return module


def main(config, stdout_dir, args_str, num_gpus=None, module="script.train_waveglow"):
    """One training process per GPU (distributed.py:145-170): ``python -m script.train_waveglow <args_str>
    --config=<config> --rank=<i> --group_name=group_<time>`` for i in 0..num_gpus-1, rank 0 on this terminal, the others
    logging to <stdout_dir>/GPU_<i>.log.  Returns the exit codes."""
    args_list = ['-m', module]
    args_list += args_str.split(' ') if len(args_str) > 0 else []
    args_list.append('--config={}'.format(config))
    if num_gpus is None:
        num_gpus = torch.cuda.device_count()
    args_list.append('--num_gpus={}'.format(num_gpus))
    args_list.append('--rank=0')
    args_list.append("--group_name=group_{}".format(time.strftime("%Y_%m_%d-%H%M%S")))
    if not os.path.isdir(stdout_dir):
        os.makedirs(stdout_dir)
        os.chmod(stdout_dir, 0o775)
    src_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env["PYTHONPATH"] = src_root + os.pathsep + env.get("PYTHONPATH", "")
    workers = []
    for i in range(num_gpus):
        args_list[-2] = '--rank={}'.format(i)
        stdout = None if i == 0 else open(os.path.join(stdout_dir, "GPU_{}.log".format(i)), "w")
        print(args_list)
        workers.append(subprocess.Popen([str(sys.executable)] + args_list, stdout=stdout, env=env))
    # a rank that dies would leave the others in a collective until RCCL's timeout: take them down with it
    codes = [None] * num_gpus
    while any(c is None for c in codes):
        for i, p in enumerate(workers):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(workers):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            break
        time.sleep(0.2)
    return codes


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', type=str, required=True, help='JSON file for configuration')
    parser.add_argument('-s', '--stdout_dir', type=str, default=".", help='directory to save stoud logs')
    parser.add_argument('-a', '--args_str', type=str, default='',
                        help='double quoted string with space separated key value pairs')
    parser.add_argument('-n', '--num_gpus', type=int, default=None, help='processes to start (default: every visible GPU)')
    args = parser.parse_args()
    sys.exit(max(abs(c) for c in main(args.config, args.stdout_dir, args.args_str, args.num_gpus)))
