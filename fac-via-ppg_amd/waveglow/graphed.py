"""The WaveGlow training step as ONE replayed HIP graph.

A bf16 step (script.train_waveglow: forward, WaveGlowLoss, backward, fused Adam) is ~1 300 kernel launches of 5-40 us each
behind ~938 parameters' worth of Python / autograd bookkeeping: at the reference's batch sizes the host, not the GPU, sets
the step time (batch 3: 22.8 ms eager against 14 ms of kernels).  The step has static shapes (fixed `segment_length`,
`drop_last` batches), so it is captured once -- every launch goes to the capturing stream through the C ABI's stream
argument, all buffers come from the graph's private pool -- and replayed per batch.

``GraphedTrainStep(model, criterion, optimizer)(mel, audio)`` does `warmup` ordinary steps first (they are real optimisation
steps), captures on the next call, and replays from then on.  With ``sync_gradients`` (data parallel: the gradient all-reduce
of waveglow.distributed) the graph holds forward + backward only and the exchange and the optimiser step run after each
replay; without it the optimiser step is inside the graph (the optimiser must then be built with ``capturable=True``).
With ``exchange`` (a waveglow.distributed.GradientExchange) there are two protocols.  OVERLAPPED (default on RCCL): the
exchange's autograd hooks stay installed during the capture, so every bucket's pack + all_reduce is captured on a forked
branch of the graph at the point of the backward pass where its last gradient is complete, joined before the optimiser
step, which is captured too -- a replay overlaps bucket i's transfer with the backward pass of the earlier flows, as an
eager step does, without giving up the graph (the round-3 stepper launched all three buckets AFTER the replay: nothing
overlapped in the mode training actually runs).  SERIAL (fallback if the collectives cannot be captured; gloo): the
replayed graph's gradient tensors are bound as the exchange's static sources and each replay is followed by the bucketed,
pipelined all-reduce.  Either way the parameters' ``.grad`` end up views of the flat buckets (no copy back).
A batch of another shape falls back to an ordinary step on the same gradient buffers; a capture that fails (pinned arena
exhausted, an op that cannot be captured, out of memory in the graph's pool) turns the stepper into plain eager steps for
good, with a warning.
"""
import os
import warnings

import torch

from waveglow.glow import reserve_pinned, take_pinned_arenas


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, warmup=3, sync_gradients=None, exchange=None, expected_shapes=None,
                 overlap_exchange=None):
        """expected_shapes: optional ((mel shape), (audio shape)) of a regular batch -- the capture then waits for a batch
        of that shape instead of locking in whatever the first post-warm-up batch happens to be.
        overlap_exchange: capture the exchange's collectives INSIDE the graph (None: yes on a GPU backend that can, i.e.
        not gloo)."""
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.warmup, self.exchange = warmup, exchange
        if overlap_exchange is None:
            import torch.distributed as dist
            # (FACPPG_TRAIN_CAPTURE_EXCHANGE=0: exchange after each replay instead, should a stack not hold collectives in a graph)
            overlap_exchange = (exchange is not None and exchange.cuda and dist.get_backend() != "gloo"
                                and os.environ.get("FACPPG_TRAIN_CAPTURE_EXCHANGE", "1") != "0")
        self.overlap_exchange = bool(overlap_exchange and exchange is not None)
        if self.overlap_exchange:
            exchange.install_hooks()                 # every backward() -- eager or captured -- now leaves averaged gradients behind
        if exchange is not None and sync_gradients is None:
            sync_gradients = exchange.exchange
        self.sync_gradients = sync_gradients
        self.expected_shapes = expected_shapes
        self.calls = 0
        self.graph = None
        self.graph_holds_step = False                # the optimiser step (and the exchange, if any) is inside the graph
        self.capture_failed = False
        self.static_mel = self.static_audio = self.static_loss = None
        self.side = torch.cuda.Stream()

    # ---- the three parts of a step
    def _forward_backward(self, mel, audio):
        loss = self.criterion(self.model((mel, audio)))
        loss.backward()
        return loss

    def _sync(self, static):
        if self.exchange is not None:
            if not self.exchange.hooked:             # (hooked: the backward pass has already exchanged)
                self.exchange.exchange(static=static)
        elif self.sync_gradients is not None:
            self.sync_gradients()

    def _finish(self):
        """What follows the gradients when it is not part of the graph."""
        if self.sync_gradients is not None and not self.graph_holds_step:
            self._sync(static=True)
            self.optimizer.step()

    def _eager(self, mel, audio, keep_grad_buffers):
        # after capture the .grad tensors ARE the graph's outputs (or, data parallel, the exchange's bucket views): zero
        # them in place instead of dropping them
        self.optimizer.zero_grad(set_to_none=not keep_grad_buffers)
        loss = self._forward_backward(mel, audio)
        self._sync(static=False)
        self.optimizer.step()
        return loss.detach()

    def _capture(self, mel, audio):
        """Capture into locals; the stepper's state changes only once the capture has succeeded."""
        static_mel, static_audio = mel.clone(), audio.clone()
        reserve_pinned()
        self.optimizer.zero_grad(set_to_none=True)   # the captured backward allocates the gradients in the graph's pool
        torch.cuda.synchronize()
        if self.exchange is not None and self.exchange.hooked:
            # The captured collectives run on a process group of their own (GradientExchange.prepare_capture_group): c10d's watchdog
            # aborts the process if it polls an eager collective's event after that group's stream has joined a capture, so the
            # stream that joins captures never carries an eager collective.  (Round 4 slept for a second here instead.)
            self.exchange.prepare_capture_group()
            torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        whole_step = self.sync_gradients is None or (self.exchange is not None and self.exchange.hooked)
        # thread_local: other threads (a DataLoader worker pinning memory, a logger) may touch the allocator meanwhile
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            # (hooked exchange: the backward pass launches every bucket's pack + all_reduce on the communication stream --
            #  a forked branch of the capture -- and joins them in its final callback)
            static_loss = self._forward_backward(static_mel, static_audio).detach()
            if whole_step:
                self.optimizer.step()
        self.graph_holds_step = whole_step
        if self.exchange is not None and not self.exchange.hooked:
            self.exchange.bind_static_sources()
        self.pinned_arenas = take_pinned_arenas()    # the graph replays uploads from these tables: they live with it
        self.static_mel, self.static_audio, self.static_loss, self.graph = static_mel, static_audio, static_loss, graph

    def _regular(self, mel, audio):
        if self.expected_shapes is None:
            return True
        return tuple(mel.shape) == tuple(self.expected_shapes[0]) and tuple(audio.shape) == tuple(self.expected_shapes[1])

    def __call__(self, mel, audio):
        """One optimisation step on (mel, audio); returns the loss (a 0-d tensor)."""
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup and not self.capture_failed:
            # warm-up on a side stream, as torch.cuda.graph asks (lazy initialisations, allocator state, autograd threads)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                loss = self._eager(mel, audio, keep_grad_buffers=False)
            torch.cuda.current_stream().wait_stream(self.side)
            return loss
        if self.capture_failed or (self.graph is None and not self._regular(mel, audio)):
            return self._eager(mel, audio, keep_grad_buffers=False)
        if self.graph is None:
            try:
                try:
                    self._capture(mel, audio)
                except Exception as e:   # noqa: BLE001
                    if not (self.exchange is not None and self.exchange.hooked):
                        raise
                    # the collectives could not be captured on this stack: the serial protocol (graph = forward + backward,
                    # the bucket pipeline after each replay) needs nothing of them inside the capture
                    warnings.warn("capturing the gradient exchange inside the HIP graph failed (%r): exchanging after each replay" % (e,))
                    torch.cuda.synchronize()
                    self.exchange.remove_hooks()
                    self.overlap_exchange = False
                    self._capture(mel, audio)
            except Exception as e:   # noqa: BLE001  (any capture failure: the training run goes on, launch by launch)
                warnings.warn("HIP-graph capture of the training step failed (%r): continuing with eager steps" % (e,))
                self.capture_failed = True
                self.graph = None
                torch.cuda.synchronize()
                return self._eager(mel, audio, keep_grad_buffers=False)
        if mel.shape != self.static_mel.shape or audio.shape != self.static_audio.shape:
            return self._eager(mel, audio, keep_grad_buffers=True)
        self.static_mel.copy_(mel)
        self.static_audio.copy_(audio)
        self.graph.replay()
        self._finish()
        return self.static_loss.clone()
