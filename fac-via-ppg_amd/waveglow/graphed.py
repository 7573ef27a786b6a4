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
A batch of another shape falls back to an ordinary step on the same gradient buffers.
"""
import torch

from waveglow.glow import reserve_pinned


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, warmup=3, sync_gradients=None):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.warmup, self.sync_gradients = warmup, sync_gradients
        self.calls = 0
        self.graph = None
        self.static_mel = self.static_audio = self.static_loss = None
        self.side = torch.cuda.Stream()

    # ---- the three parts of a step
    def _forward_backward(self, mel, audio):
        loss = self.criterion(self.model((mel, audio)))
        loss.backward()
        return loss

    def _finish(self):
        """What follows the gradients when it is not part of the graph."""
        if self.sync_gradients is not None:
            self.sync_gradients()
            self.optimizer.step()

    def _eager(self, mel, audio, keep_grad_buffers):
        # after capture the .grad tensors ARE the graph's outputs: zero them in place instead of dropping them
        self.optimizer.zero_grad(set_to_none=not keep_grad_buffers)
        loss = self._forward_backward(mel, audio)
        if self.sync_gradients is not None:
            self.sync_gradients()
        self.optimizer.step()
        return loss.detach()

    def _capture(self, mel, audio):
        self.static_mel, self.static_audio = mel.clone(), audio.clone()
        reserve_pinned()
        self.optimizer.zero_grad(set_to_none=True)   # the captured backward allocates the gradients in the graph's pool
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the DataLoader's pin-memory thread may allocate pinned memory while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_loss = self._forward_backward(self.static_mel, self.static_audio).detach()
            if self.sync_gradients is None:
                self.optimizer.step()

    def __call__(self, mel, audio):
        """One optimisation step on (mel, audio); returns the loss (a 0-d tensor)."""
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            # warm-up on a side stream, as torch.cuda.graph asks (lazy initialisations, allocator state, autograd threads)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                loss = self._eager(mel, audio, keep_grad_buffers=False)
            torch.cuda.current_stream().wait_stream(self.side)
            return loss
        if self.graph is None:
            self._capture(mel, audio)
        if mel.shape != self.static_mel.shape or audio.shape != self.static_audio.shape:
            return self._eager(mel, audio, keep_grad_buffers=True)
        self.static_mel.copy_(mel)
        self.static_audio.copy_(audio)
        self.graph.replay()
        self._finish()
        return self.static_loss.clone()
