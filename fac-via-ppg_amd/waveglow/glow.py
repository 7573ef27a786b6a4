"""WaveGlow vocoder -- drop-in for the reference's ``waveglow.glow`` (src/waveglow/glow.py).

Same classes, constructor arguments, parameter/state-dict layout and checkpoint behaviour as
the reference (checkpoints pickle ``waveglow.glow.WaveGlow`` objects, train_waveglow.py:56-64,
so the qualified names matter), but ``WaveGlow.infer`` does not run torch ops: it hands raw
device pointers to ``facppg_wg_infer`` in libfacppg_hip.so (include/facppg.h), whose fused
gfx950 kernels implement glow.py:252-293.  The ``torch.nn`` modules below are parameter
containers only; there is no CPU path.

Extensions over the reference signature (all optional, defaults = reference behaviour):
  ``infer(spect, sigma=1.0, z=None, lengths=None, seed=None)``
    z        the three N(0,1) draws of glow.py:261-270,285-290 (list of tensors in call order, or
             one flat tensor); None -> generated on the device (Philox) from ``seed``
    lengths  per-utterance valid frame counts for padded batches: utterance b is synthesised
             exactly as a batch-1 call on spect[b, :, :lengths[b]] would be
    utterance_seeds  B integers: the noise of utterance b depends on utterance_seeds[b] alone
             (facppg_wg_draw_noise), so the batch reproduces B batch-1 calls with the same seeds
"""
import os

import torch

from facppg import lib as _lib


def fused_add_tanh_sigmoid_multiply(input_a, input_b, n_channels):
    """glow.py:33-40.  In the HIP path this is the epilogue of the first GEMM of k_wn_layer;
    the standalone function is kept for API compatibility on tensors the caller already has."""
    n = int(n_channels[0])
    in_act = input_a + input_b
    return torch.tanh(in_act[:, :n, :]) * torch.sigmoid(in_act[:, n:, :])


def _sum_segment(t):
    """(outer, outer_stride, inner) of a float32 CUDA tensor that is `outer` runs of `inner` contiguous elements, or None."""
    if t.dim() == 0 or t.is_contiguous():
        return 1, 0, t.numel()
    if t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.size(2):       # a channel slice of a contiguous [B, C, L] tensor
        return t.size(0), t.stride(0), t.size(1) * t.size(2)
    return None


class _GlowLossFunction(torch.autograd.Function):
    """WaveGlowLoss.forward (glow.py:43-59) as one autograd node: the 13 reductions of the reference (sum(z*z), one sum(log_s)
    per flow) are ONE facppg_segment_sums call and the scalar arithmetic around them a handful of 13-element ops, instead of
    ~50 reduction / add / mul launches forward and as many backward (slice backward + accumulate per flow).  Same value:
    (sum(z*z) / (2 sigma^2) - sum_k sum(log_s_k) - sum_k log_det_W_k) / numel(z); gradients: z / (sigma^2 numel) for z and the
    constant -1 / numel for every log_s element and every log_det_W (returned as stride-0 expansions of one scalar)."""

    @staticmethod
    def forward(ctx, sigma, n_log_s, z, *rest):
        L = _lib.load()
        dev = z.device
        log_s, log_det = rest[:n_log_s], rest[n_log_s:]
        tensors = [z] + list(log_s)
        segs = (_lib.SumSegment * len(tensors))()
        for i, t in enumerate(tensors):
            outer, stride, inner = _sum_segment(t)
            segs[i] = _lib.SumSegment(t.data_ptr(), stride, outer, inner, 1 if i == 0 else 0)
        ws = torch.empty(len(tensors) * 64, dtype=torch.float64, device=dev)
        sums = torch.empty(len(tensors), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_segment_sums(segs, len(tensors), _lib.ptr(ws), ws.numel() * 8, _lib.ptr(sums), _lib.current_stream(dev)))
        n = z.numel()
        total = sums[0] * (0.5 / (sigma * sigma)) - sums[1:].sum()
        if log_det:
            total = total - torch.stack([t.reshape(()) for t in log_det]).sum()
        ctx.save_for_backward(z)
        ctx.sigma, ctx.n = float(sigma), n
        ctx.shapes = [t.shape for t in rest]
        return total / n

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        c = g / ctx.n
        neg = -c
        return (None, None, z * (c * (1.0 / (ctx.sigma * ctx.sigma)))) + tuple(neg.expand(sh) for sh in ctx.shapes)


class WaveGlowLoss(torch.nn.Module):
    """glow.py:43-59 (training loss; a scalar reduction of the flow outputs)."""

    def __init__(self, sigma=1.0):
        super(WaveGlowLoss, self).__init__()
        self.sigma = sigma

    def forward(self, model_output):
        z, log_s_list, log_det_W_list = model_output
        fusable = [z] + list(log_s_list) + list(log_det_W_list)
        if (z.is_cuda and len(log_s_list) < 16 and all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in fusable)
                and all(_sum_segment(t) is not None for t in [z] + list(log_s_list)) and all(t.numel() == 1 for t in log_det_W_list)):
            return _GlowLossFunction.apply(self.sigma, len(log_s_list), z, *log_s_list, *log_det_W_list)
        log_s_total = sum(torch.sum(ls) for ls in log_s_list)
        log_det_total = sum(log_det_W_list)
        loss = torch.sum(z * z) / (2 * self.sigma * self.sigma) - log_s_total - log_det_total
        return loss / (z.size(0) * z.size(1) * z.size(2))


def _effective_weight(conv):
    """Weight of a conv whether or not weight-norm is still attached (Denoiser is handed a model
    that still carries weight_g/weight_v, generate_synthesis.py:58-61)."""
    if hasattr(conv, "weight_g"):
        return torch._weight_norm(conv.weight_v, conv.weight_g, 0)
    return conv.weight


_TN, _HALO = 64, 128   # tile width / zero margin of the HIP layouts (csrc/facppg_wg.hip)


class _WNFunction(torch.autograd.Function):
    """One flow's WN stack (glow.py:154-175) as an autograd node on the HIP kernels.

    forward  = facppg_wn_forward_save: the fused k_wn_layer launches, keeping layer inputs and the
               tanh/sigmoid halves of each gate;
    backward = facppg_wn_backward_data for everything that flows through the data (transposed
               dilated convs, gate derivative, conditioning gradient: exact-fp32 MFMA GEMMs) plus
               facppg_wn_weight_grads for the weight / bias gradients (NT products over batch and positions
               of the saved tensors, exact-fp32 MFMA as well): nothing of the stack runs in torch or rocBLAS.
    Inputs: a0 [B, n_in, L], spect_pad [B, 640, Lr], then start.w, start.b, per layer
    (in.w, in.b, cond.w, cond.b, res_skip.w, res_skip.b), end.w, end.b -- plain effective weights."""

    @staticmethod
    def _weights_struct(ws, struct=None, into=None):
        n_layers = (len(ws) - 4) // 6
        st = into if into is not None else (struct or _lib.WnWeights)()
        st.start_w, st.start_b = ws[0].data_ptr(), ws[1].data_ptr()
        for i in range(n_layers):
            q = ws[2 + 6 * i: 8 + 6 * i]
            st.in_w[i], st.in_b[i], st.cond_w[i], st.cond_b[i], st.rs_w[i], st.rs_b[i] = [t.data_ptr() for t in q]
        st.end_w, st.end_b = ws[-2].data_ptr(), ws[-1].data_ptr()
        return st, n_layers

    @staticmethod
    def forward(ctx, a0, spect_pad, *weights):
        L = _lib.load()
        dev = a0.device
        a0 = a0.contiguous()
        ws_t = [w.detach().float().contiguous() for w in weights]
        st, n_layers = _WNFunction._weights_struct(ws_t)
        B, n_in, Lg = a0.shape
        Lr = spect_pad.shape[2]
        Lp = _HALO + Lr + _HALO
        out = torch.empty(B, 2 * n_in, Lg, device=dev)
        h_all = torch.empty(n_layers + 1, B, 256, Lp, device=dev)
        ts_all = torch.empty(n_layers, B, 512, Lr, device=dev)
        skip = torch.empty(B, 256, Lr, device=dev)
        work = torch.empty(L.facppg_wn_train_workspace_bytes(n_layers, B, Lg), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_wn_forward_save(st, n_in, n_layers, _lib.ptr(a0), _lib.ptr(spect_pad), B, Lg, _lib.ptr(out),
                                                _lib.ptr(h_all), _lib.ptr(ts_all), _lib.ptr(skip), _lib.ptr(work), work.numel(),
                                                _lib.current_stream(dev)))
        ctx.save_for_backward(a0, spect_pad, h_all, ts_all, skip, *ws_t)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.load()
        a0, spect_pad, h_all, ts_all, skip, *ws_t = ctx.saved_tensors
        dev = a0.device
        st, n_layers = _WNFunction._weights_struct(ws_t)
        B, n_in, Lg = a0.shape
        Lr = spect_pad.shape[2]
        dout = dout.contiguous().float()
        dpre_all = torch.empty(n_layers, B, 512, Lr, device=dev)
        dh_all = torch.empty(n_layers + 1, B, 256, Lr, device=dev)
        dskip = torch.empty(B, 256, Lr, device=dev)
        dspect = torch.zeros(B, spect_pad.shape[1], Lr, device=dev)
        da0 = torch.empty_like(a0)
        work = torch.empty(L.facppg_wn_train_workspace_bytes(n_layers, B, Lg), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_wn_backward_data(st, n_in, n_layers, _lib.ptr(dout), _lib.ptr(ts_all), B, Lg, _lib.ptr(dpre_all),
                                                 _lib.ptr(dh_all), _lib.ptr(dskip), _lib.ptr(dspect), _lib.ptr(da0),
                                                 _lib.ptr(work), work.numel(), _lib.current_stream(dev)))

            # the parameter half: every weight gradient is an NT product over batch and positions of two saved tensors (the
            # dilated conv's three taps against shifted views of the layer input, the conditioning conv against the
            # spectrogram, res_skip against tanh * sigmoid, start / end against a0 / skip), every bias gradient a row sum:
            # two launches of the library's fp32 MFMA kernels for the whole stack (torch.bmm(...).sum(0) per product until
            # round 4).  dout has L columns per row, everything else Lr (the kernels mask positions >= L).
            grads = [torch.empty_like(w) for w in ws_t]
            gs, _ = _WNFunction._weights_struct(grads, _lib.WnGrads)
            gwork = torch.empty(L.facppg_wn_weight_grads_workspace_bytes(n_layers), dtype=torch.uint8, device=dev)
            _lib.check(L.facppg_wn_weight_grads(n_in, n_layers, _lib.ptr(a0), _lib.ptr(spect_pad), _lib.ptr(h_all), _lib.ptr(ts_all),
                                                _lib.ptr(skip), _lib.ptr(dout), _lib.ptr(dpre_all), _lib.ptr(dh_all), _lib.ptr(dskip),
                                                B, Lg, gs, _lib.ptr(gwork), gwork.numel(), _lib.current_stream(dev)))
        return (da0, dspect, *grads)


def spect_to_posmajor_bf16(spect, Lg):
    """[B, 640, >= Lg] fp32 channel-major -> the bf16 position-major operand [B, Lr, 640] of the bf16 training kernels
    (rows >= Lg zero)."""
    L = _lib.load()
    spect = spect.detach().float().contiguous()
    B, Cn, ld = spect.shape
    Lr = L.facppg_wn_bf16_padded_len(Lg)
    out = torch.empty(B, Lr, Cn, dtype=torch.bfloat16, device=spect.device)
    with torch.cuda.device(spect.device):
        _lib.check(L.facppg_spect_to_bf16(_lib.ptr(spect), B, Cn, Lg, ld, _lib.ptr(out), _lib.current_stream(spect.device)))
    return out


class _StepShared(object):
    """What the bf16 flows of one training step share: the fp32 position-major conditioning gradient every flow's
    backward adds to (one buffer instead of 12 tensors summed by autograd), consumed by the upsampler's backward.

    (Round 3 tried a second stream here: the weight / bias gradients of a flow and the weight-norm backward enqueued
    behind an event, to run underneath the following flows' data-gradient chains.  Measured slower on one MI355X -- graphed
    13.3 vs 12.2 ms at batch 3, 24.4 vs 24.0 at batch 12 -- and unsound as soon as autograd copies an incoming gradient
    instead of adopting it, which it does on the first stream: removed; profiles/r03_experiments.txt.)"""

    def __init__(self):
        self.dspect_pm = None


class _UpsampleBf16Function(torch.autograd.Function):
    """WaveGlow.upsample + crop + regroup (glow.py:184-186, 214-222) on HIP kernels: forward straight into the bf16
    position-major conditioning operand; backward = the ConvTranspose1d's weight / bias gradients from the shared
    conditioning gradient.  Returns (spect_pm, link): ``link`` is a 1-element tensor threaded through every flow so
    that autograd runs this backward after all of theirs."""

    @staticmethod
    def forward(ctx, mel, up_w, up_b, shared, hop, Lg):
        L = _lib.load()
        dev = mel.device
        mel = mel.detach().float().contiguous()
        w, b = up_w.detach().float().contiguous(), up_b.detach().float().contiguous()
        B, nm, T = mel.shape
        Lr = L.facppg_wn_bf16_padded_len(Lg)
        spect_pm = torch.empty(B, Lr, nm * 8, dtype=torch.bfloat16, device=dev)
        ws = torch.empty(L.facppg_upsample_forward_workspace_bytes(B, T, nm, hop, w.shape[2], Lg), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_upsample_regroup_bf16(_lib.ptr(mel), _lib.ptr(w), _lib.ptr(b), B, T, nm, hop, w.shape[2], Lg,
                                                      _lib.ptr(spect_pm), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))
        ctx.save_for_backward(mel)
        ctx.shared, ctx.hop, ctx.Lg, ctx.wshape = shared, hop, Lg, tuple(w.shape)
        ctx.set_materialize_grads(False)                  # (the link's gradient is None: it only orders this node behind the flows)
        ctx.mark_non_differentiable(spect_pm)
        return spect_pm, torch.zeros(1, device=dev)

    @staticmethod
    def backward(ctx, _dspect_pm, _dlink):
        L = _lib.load()
        (mel,) = ctx.saved_tensors
        dev = mel.device
        B, nm, T = mel.shape
        dw = torch.empty(ctx.wshape, device=dev)
        db = torch.empty(nm, device=dev)
        d = ctx.shared.dspect_pm
        if d is None:                                     # no flow contributed (cannot happen in WaveGlow.forward)
            return None, torch.zeros_like(dw), torch.zeros_like(db), None, None, None
        ws = torch.empty(L.facppg_upsample_backward_workspace_bytes(B, T, nm, ctx.hop, ctx.wshape[2], ctx.Lg), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_upsample_regroup_backward(_lib.ptr(mel), _lib.ptr(d), B, T, nm, ctx.hop, ctx.wshape[2], ctx.Lg, _lib.ptr(dw),
                                                          _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))
        ctx.shared.dspect_pm = None
        return None, dw, db, None, None, None


class _WNFunctionBf16(torch.autograd.Function):
    """One flow's WN stack with bf16 MFMA operands, fp32 accumulation and fp32 gradients (BASELINE config 5).
    forward  = facppg_wn_forward_bf16 (fused dilated-conv + conditioning GEMM with the gate in its epilogue, res/skip
               GEMM with the residual / skip update in its epilogue, per layer);
    backward = facppg_wn_backward_bf16: data gradients AND every weight / bias gradient (NT products over the positions,
               batched over layers and taps) -- nothing of the stack's backward runs in torch or rocBLAS.
    Inputs: a0 [B, n_in, L]; ``route``: stand-alone use (shared None) = spect [B, 640, >= L] fp32, which receives the
    conditioning gradient; inside a training step = the upsampler's ``link`` (the gradient is added to shared.dspect_pm
    instead); spect_pm, the bf16 position-major conditioning operand; then the plain effective weights as in
    _WNFunction."""

    @staticmethod
    def forward(ctx, a0, route, shared, spect_pm, *weights):
        L = _lib.load()
        dev = a0.device
        a0 = a0.detach().float().contiguous()
        ws_t = [w.detach().float().contiguous() for w in weights]
        st, n_layers = _WNFunction._weights_struct(ws_t)
        B, n_in, Lg = a0.shape
        out = torch.empty(B, 2 * n_in, Lg, device=dev)
        state = torch.empty(L.facppg_wn_bf16_state_bytes(n_layers, B, Lg), dtype=torch.uint8, device=dev)
        work = torch.empty(L.facppg_wn_bf16_scratch_bytes(n_layers, B, Lg), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_wn_forward_bf16(st, n_in, n_layers, _lib.ptr(a0), _lib.ptr(spect_pm), B, Lg, _lib.ptr(out),
                                                _lib.ptr(state), state.numel(), _lib.ptr(work), work.numel(), _lib.current_stream(dev)))
        ctx.save_for_backward(a0, spect_pm, state, *ws_t)
        ctx.shared, ctx.route_shape = shared, tuple(route.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.load()
        a0, spect_pm, state, *ws_t = ctx.saved_tensors
        dev = a0.device
        st, n_layers = _WNFunction._weights_struct(ws_t)
        B, n_in, Lg = a0.shape
        dout = dout.contiguous().float()
        grads = [torch.empty_like(w) for w in ws_t]
        gs, _ = _WNFunction._weights_struct(grads, _lib.WnGrads)
        da0 = torch.empty_like(a0)
        work = torch.empty(L.facppg_wn_bf16_scratch_bytes(n_layers, B, Lg), dtype=torch.uint8, device=dev)
        shared = ctx.shared
        accumulate = shared is not None and shared.dspect_pm is not None
        if shared is None or not accumulate:
            dspect_pm = torch.empty(spect_pm.shape, dtype=torch.float32, device=dev)
            if shared is not None:
                shared.dspect_pm = dspect_pm
        else:
            dspect_pm = shared.dspect_pm
        droute = torch.zeros(ctx.route_shape, device=dev)
        with torch.cuda.device(dev):
            s = _lib.current_stream(dev)
            _lib.check(L.facppg_wn_backward_bf16(st, gs, n_in, n_layers, _lib.ptr(a0), _lib.ptr(spect_pm), _lib.ptr(dout), B, Lg,
                                                 _lib.ptr(state), state.numel(), _lib.ptr(da0), _lib.ptr(dspect_pm), 1 if accumulate else 0,
                                                 _lib.ptr(work), work.numel(), s))
            if shared is None:
                _lib.check(L.facppg_posmajor_to_f32(_lib.ptr(dspect_pm), B, droute.shape[1], Lg, _lib.ptr(droute), droute.shape[2], s))
        return (da0, droute, None, None, *grads)


class _GlowStepBf16(object):
    """What the flows of ONE bf16 training step share (train_waveglow.py:126-133 on the HIP kernels): both directions' packed bf16
    operand images of every flow's stack and the summed gate biases (formed ONCE per step: facppg_glow_bf16_begin), every flow's
    saved activations (their zero margins written by the same call), the one set of gradient buffers the flows' backward passes
    take turns on, the <= 8-channel tensors at the flow boundaries, and the output z the early outputs are written straight into.
    The object is held by the group nodes' contexts; the large buffers (packed, states, work) become the nodes' SAVED TENSORS as the
    groups run forward (released by the backward pass like any saved tensor) and the object lets go of them after its last group."""

    def __init__(self, model, flow_weights, B, Lg, dev, shared):
        L = _lib.load()
        c = _lib.ctypes
        self.shared = shared                                # (_StepShared: the conditioning gradient the upsampler's backward consumes)
        self.n_flows, self.nl, self.B, self.Lg, self.dev = model.n_flows, model.WN[0].n_layers, B, Lg, dev
        self.sizes = _lib.GlowBf16Sizes()
        _lib.check(L.facppg_glow_bf16_layout(self.nl, B, Lg, c.byref(self.sizes)))
        nf = self.n_flows
        self.packed = torch.empty(nf * self.sizes.packed_bytes_per_flow, dtype=torch.uint8, device=dev)
        self.states = torch.empty(nf * self.sizes.state_bytes_per_flow, dtype=torch.uint8, device=dev)
        self.work = torch.empty(self.sizes.work_bytes, dtype=torch.uint8, device=dev)
        # channels through every flow and what is split off in front of it (glow.py:231-233)
        self.plan, ch = [], model.n_group
        for k in range(nf):
            early = model.n_early_size if (k % model.n_early_every == 0 and k > 0) else 0
            ch -= early
            self.plan.append((ch, early))
        # the boundary tensors of all flows in ONE allocation: u | z | wn_out | dzp [B, c, Lg] each, logdet [1 + c*c]
        offs, tot = [], 0
        for ch, _ in self.plan:
            offs.append(tot)
            tot += 4 * B * ch * Lg + 1 + ch * ch
        self.small = torch.empty(tot, dtype=torch.float32, device=dev)
        self.flow_small = []
        for (ch, _), o in zip(self.plan, offs):
            n = B * ch * Lg
            u, z, wn, dzp = (self.small[o + i * n:o + (i + 1) * n].view(B, ch, Lg) for i in range(4))
            self.flow_small.append((u, z, wn, dzp, self.small[o + 4 * n:o + 4 * n + 1 + ch * ch]))
        self.z = torch.empty(B, model.n_group, Lg, dtype=torch.float32, device=dev)     # [early outputs ... | last flow's output]
        self.parts = None                                   # backward partial sums, allocated by the first group that runs backward
        self.wts_t = [[w.detach().float().contiguous() for w in fw] for fw in flow_weights]
        self.wts = (_lib.WnWeights * nf)()
        for k in range(nf):
            _WNFunction._weights_struct(self.wts_t[k], into=self.wts[k])
        with torch.cuda.device(dev):
            _lib.check(L.facppg_glow_bf16_begin(self.wts, nf, self.nl, B, Lg, _lib.ptr(self.packed), _lib.ptr(self.states), _lib.ptr(self.work),
                                                _lib.current_stream(dev)))

    def flow_struct(self, k, conv_w, packed=None, states=None):
        """facppg_glow_flow of flow k over the step's buffers (forward: the step's own; backward: the node's saved tensors)."""
        packed = self.packed if packed is None else packed
        states = self.states if states is None else states
        f = _lib.GlowFlow()
        ch, early = self.plan[k]
        u, z, wn, dzp, ld = self.flow_small[k]
        f.w = _lib.ctypes.pointer(self.wts[k])
        f.conv_w, f.logdet, f.ld_scale, f.c, f.early = conv_w.data_ptr(), ld.data_ptr(), float(self.B * self.Lg), ch, early
        f.u, f.z, f.wn_out, f.dzp = u.data_ptr(), z.data_ptr(), wn.data_ptr(), dzp.data_ptr()
        f.packed = packed.data_ptr() + k * self.sizes.packed_bytes_per_flow
        f.state = states.data_ptr() + k * self.sizes.state_bytes_per_flow
        return f


class _FlowGroupBf16Function(torch.autograd.Function):
    """A GROUP of consecutive flows of the training direction (glow.py:228-247) as one autograd node on HIP kernels
    (facppg_glow_bf16_group_forward / _backward): per flow boundary ONE launch forward (end conv + affine coupling of the flow
    below, early split, 1x1 mixing conv and start conv of the flow above) and one backward (which also leaves the partial sums
    of the <= 8-channel weight gradients), the WN layers in between.  The groups are the weight-norm groups = the gradient buckets
    of the data-parallel exchange: a group's weight gradients are complete when its node's backward returns.
    Inputs: audio [B, c_in, L] (contiguous), the upsampler's link, then per flow the mixing matrix [c, c] and the stack's
    effective weights.  Outputs: audio_out, the early outputs of the group's flows, log_s per flow (views of the saved stack
    outputs), B * L * log|det W| per flow."""

    @staticmethod
    def forward(ctx, audio, link, step, k0, nfl, spect_pm, last, *tensors):
        L = _lib.load()
        dev = audio.device
        B, Lg = step.B, step.Lg
        audio = audio.detach().float().contiguous()
        convs = [t.detach().float().contiguous() for t in tensors[:nfl]]
        flows = (_lib.GlowFlow * nfl)()
        c_in = audio.shape[1]
        assert c_in == step.plan[k0][0] + step.plan[k0][1], (c_in, step.plan[k0])
        zc, zoff = step.z, sum(e for _, e in step.plan[:k0])      # early outputs go straight into the model's output z
        earlies = []
        for i in range(nfl):
            flows[i] = step.flow_struct(k0 + i, convs[i])
            ch, early = step.plan[k0 + i]
            if early:
                ev = zc[:, zoff:zoff + early, :]
                flows[i].early_io, flows[i].early_bs = ev.data_ptr(), zc.stride(0)
                earlies.append(ev)
                zoff += early
        c_out = step.plan[k0 + nfl - 1][0]
        if last:       # the last flow's output is the rest of z
            out = zc[:, zoff:zoff + c_out, :]
        else:
            out = torch.empty(B, c_out, Lg, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_glow_bf16_group_forward(flows, nfl, step.nl, _lib.ptr(audio), audio.stride(0), out.data_ptr(), out.stride(0),
                                                        _lib.ptr(spect_pm), B, Lg, _lib.current_stream(dev)))
        # The GBs of saved activations / weight images / gradient buffers belong to the graph like any saved tensor: released by the
        # backward pass (unless the graph is retained), not when the last reference to the loss goes.  Every group node saves the
        # same three tensors; the step object itself lets go of them once its last group has run forward.
        ctx.save_for_backward(step.packed, step.states, step.work)
        if last:
            step.packed = step.states = step.work = None
        ctx.step, ctx.k0, ctx.nfl, ctx.convs, ctx.spect_pm, ctx.n_early = step, k0, nfl, convs, spect_pm, len(earlies)
        ctx.in_shape, ctx.link_shape = tuple(audio.shape), tuple(link.shape)
        log_s = [step.flow_small[k0 + i][2][:, step.plan[k0 + i][0] // 2:, :] for i in range(nfl)]
        log_det = [step.flow_small[k0 + i][4][0] for i in range(nfl)]        # (the kernel stores B * L * log|det W|, glow.py:100)
        return (out, *earlies, *log_s, *log_det)

    @staticmethod
    def backward(ctx, d_out, *rest):
        L = _lib.load()
        step, k0, nfl = ctx.step, ctx.k0, ctx.nfl
        packed, states, work = ctx.saved_tensors
        dev = d_out.device
        B, Lg = step.B, step.Lg
        d_early, d_log_s, d_log_det = rest[:ctx.n_early], rest[ctx.n_early:ctx.n_early + nfl], rest[ctx.n_early + nfl:]
        d_out = d_out.float()
        if d_out.stride(2) != 1 or d_out.stride(1) != Lg:
            d_out = d_out.contiguous()
        if step.parts is None:
            step.parts = torch.empty(step.n_flows * step.sizes.n_parts * step.sizes.part_floats, dtype=torch.float32, device=dev)
        per = step.sizes.n_parts * step.sizes.part_floats
        flows = (_lib.GlowFlow * nfl)()
        keep, grads, gstructs, dconvs = [], [], (_lib.WnGrads * nfl)(), []
        ei = 0
        for i in range(nfl):
            k = k0 + i
            f = step.flow_struct(k, ctx.convs[i], packed, states)
            ch, early = step.plan[k]
            g = [torch.empty_like(w) for w in step.wts_t[k]]
            _WNFunction._weights_struct(g, into=gstructs[i])
            f.g = _lib.ctypes.pointer(gstructs[i])
            dW = torch.empty(ch, ch, dtype=torch.float32, device=dev)
            f.d_conv_w = dW.data_ptr()
            f.part = step.parts.data_ptr() + 4 * per * k
            if early:
                de = d_early[ei]
                ei += 1
                if de is None:
                    de = torch.zeros(B, early, Lg, dtype=torch.float32, device=dev)
                elif de.stride(2) != 1 or de.stride(1) != Lg:
                    de = de.contiguous()
                f.early_io, f.early_bs = de.data_ptr(), de.stride(0)
                keep.append(de)
            dl = d_log_s[i]
            if dl is not None:
                dl = dl.float()
                f.dlog_s, f.dls_b, f.dls_j, f.dls_n = dl.data_ptr(), dl.stride(0), dl.stride(1), dl.stride(2)
                keep.append(dl)
            gd = d_log_det[i]
            if gd is not None:
                gd = gd.float()
                f.g_logdet = gd.data_ptr()
                keep.append(gd)
            flows[i] = f
            grads.append(g)
            dconvs.append(dW)
        d_in = torch.empty(ctx.in_shape, dtype=torch.float32, device=dev)
        shared = step.shared
        accumulate = shared.dspect_pm is not None
        if not accumulate:
            shared.dspect_pm = torch.empty(ctx.spect_pm.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_glow_bf16_group_backward(flows, nfl, step.nl, d_out.data_ptr(), d_out.stride(0), _lib.ptr(d_in), d_in.stride(0),
                                                         _lib.ptr(ctx.spect_pm), _lib.ptr(shared.dspect_pm), 1 if accumulate else 0,
                                                         _lib.ptr(work), B, Lg, _lib.current_stream(dev)))
        if k0 == 0:
            step.parts = None          # (the first group's node runs backward last)
        flat = []
        for dW, g in zip(dconvs, grads):
            flat.append(dW)
        for g in grads:
            flat.extend(g)
        return (d_in, None, None, None, None, None, None, *flat)


class _AssembleZFunction(torch.autograd.Function):
    """torch.cat(output_audio, 1) (glow.py:249) without a copy: the group nodes wrote the early outputs and the last flow's output
    straight into the step's z buffer, of which the inputs here are channel slices; forward hands out that buffer, backward the
    matching slices of its gradient."""

    @staticmethod
    def forward(ctx, step, *pieces):
        ctx.widths = [p.shape[1] for p in pieces]
        # (a NEW tensor object over the buffer: returning step.z itself would hang this node -- and through it the whole step's graph,
        #  whose contexts hold `step` -- on an attribute of `step`: a reference cycle that only the cyclic collector breaks, i.e. every
        #  step's saved activations alive until then)
        return step.z.view(step.z.shape)

    @staticmethod
    def backward(ctx, dz):
        out, o = [], 0
        for w in ctx.widths:
            out.append(dz[:, o:o + w, :])
            o += w
        return (None, *out)


def _conv1x1(W, z, transpose=False):
    """[c, c] x [B, c, L] channel mixing on the HIP flow-edge kernel (facppg_conv1x1)."""
    L = _lib.load()
    z = z.float().contiguous()
    W = W.detach().float().contiguous().to(z.device)
    out = torch.empty_like(z)
    B, c, Lg = z.shape
    with torch.cuda.device(z.device):
        _lib.check(L.facppg_conv1x1(_lib.ptr(W), _lib.ptr(z), _lib.ptr(out), B, c, Lg, 1 if transpose else 0,
                                    _lib.current_stream(z.device)))
    return out


class _LogDetFunction(torch.autograd.Function):
    """torch.logdet(W) of the mixing matrix (glow.py:100) as ONE HIP launch (LU with partial pivoting, c <= 8), with its
    gradient W^-T from the same launch -- torch.logdet and its backward go through rocSOLVER in ~22 small launches per flow
    and direction.  NaN / -inf for a negative / zero determinant, as torch.logdet."""

    @staticmethod
    def forward(ctx, W):
        L = _lib.load()
        Wc = W.detach().float().contiguous()
        out = torch.empty(1 + Wc.numel(), device=Wc.device)
        with torch.cuda.device(Wc.device):
            _lib.check(L.facppg_logdet(_lib.ptr(Wc), Wc.shape[0], _lib.ptr(out), _lib.ptr(out[1:]), _lib.current_stream(Wc.device)))
        ctx.save_for_backward(out)
        ctx.shape = Wc.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return g * out[1:].view(ctx.shape)


class _Conv1x1Function(torch.autograd.Function):
    """The mixing conv of Invertible1x1Conv.forward (glow.py:98-102) as an autograd node on HIP kernels:
    forward W z, backward dz = W^T dout (same kernel, transposed matrix) and dW = sum_{b,l} dout z^T."""

    @staticmethod
    def forward(ctx, W, z):
        ctx.save_for_backward(W, z)
        return _conv1x1(W, z)

    @staticmethod
    def backward(ctx, dout):
        W, z = ctx.saved_tensors
        L = _lib.load()
        dout = dout.float().contiguous()
        B, c, Lg = z.shape
        dz = _conv1x1(W, dout, transpose=True) if ctx.needs_input_grad[1] else None
        dW = None
        if ctx.needs_input_grad[0]:
            zc = z.float().contiguous()
            dW = torch.empty(c, c, device=z.device)
            ws = torch.empty(L.facppg_conv1x1_wgrad_workspace_bytes(c), dtype=torch.uint8, device=z.device)
            with torch.cuda.device(z.device):
                _lib.check(L.facppg_conv1x1_wgrad(_lib.ptr(dout), _lib.ptr(zc), _lib.ptr(dW), B, c, Lg, _lib.ptr(ws), ws.numel(),
                                                  _lib.current_stream(z.device)))
        return dW, dz


_PINNED = {"arenas": [], "used": 0}


def _pinned_like(array):
    """A pinned host copy of a small int64 numpy table, carved from arenas allocated OUTSIDE graph capture (pinning memory
    is not permitted while a stream is capturing; call `reserve_pinned()` before capturing a training step)."""
    n = array.size
    arenas = _PINNED["arenas"]
    if not arenas or _PINNED["used"] + n > arenas[-1].numel():
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("pinned table arena exhausted during graph capture: call waveglow.glow.reserve_pinned() first")
        arenas.append(torch.empty(max(1 << 18, n), dtype=torch.int64).pin_memory())
        _PINNED["used"] = 0
    out = arenas[-1][_PINNED["used"]:_PINNED["used"] + n].view(array.shape)
    _PINNED["used"] += n
    out.copy_(torch.from_numpy(array))
    return out


def reserve_pinned():
    """Make sure a fresh 2 MiB pinned arena is available (enough for every table of a captured training step)."""
    _PINNED["arenas"].append(torch.empty(1 << 18, dtype=torch.int64).pin_memory())
    _PINNED["used"] = 0


def take_pinned_arenas():
    """Hand the current arenas to the caller and start afresh.  A captured HIP graph replays uploads FROM these pinned
    tables, so whoever owns the graph (waveglow.graphed.GraphedTrainStep) must own them too: they then live exactly as long
    as the graph, and recycling the shared arenas (below) can never pull memory from under a replay."""
    arenas, _PINNED["arenas"], _PINNED["used"] = _PINNED["arenas"], [], 0
    return arenas


def _recycle_pinned():
    """Drop every table and the arenas behind them (never called while capturing).  Pinned memory is bounded by this:
    at most 64 tables (~14 KB each) between two recyclings, whatever addresses the caching allocator hands out."""
    _WeightNormAllFunction._tables.clear()
    _PINNED["arenas"], _PINNED["used"] = [], 0


class _WeightNormAllFunction(torch.autograd.Function):
    """w_i = g_i * v_i / ||v_i|| (per output row) for EVERY weight-normed conv of the model in one HIP launch, and the
    matching backward in one more (torch's weight_norm recomputes each conv with its own two kernels: 288 convs ->
    ~600 launches per step).  Inputs v_0, g_0, v_1, g_1, ...; outputs w_0, w_1, ..."""

    _tables = {}   # (device, entries) -> (pinned host table, device table, total rows)

    @staticmethod
    def _table(entries, dev):
        """Device table of one launch.  The caching allocator hands a training loop the same buffers step after step, so
        the table is almost always a cache hit (no upload).  A miss uploads from PINNED memory that the cache keeps alive:
        an upload recorded while a HIP graph is being captured stays valid for every replay."""
        import numpy as np
        key = (str(dev), tuple(entries))
        hit = _WeightNormAllFunction._tables.get(key)
        if hit is None:
            t = np.zeros((len(entries), 6), dtype=np.int64)
            row0 = 0
            for i, (v, g, w, norm, rows, ln) in enumerate(entries):
                t[i] = (v, g, w, norm, row0, rows | (ln << 32))
                row0 += rows
            if len(_WeightNormAllFunction._tables) >= 64 and not torch.cuda.is_current_stream_capturing():
                _recycle_pinned()
            host = _pinned_like(t)
            hit = _WeightNormAllFunction._tables[key] = (host, host.to(dev, non_blocking=True), row0)
        return hit[1], hit[2]

    @staticmethod
    def forward(ctx, *vg):
        L = _lib.load()
        vs = [t.detach().float().contiguous() for t in vg[0::2]]
        gs = [t.detach().float().contiguous() for t in vg[1::2]]
        dev = vs[0].device
        wflat = torch.empty(sum(v.numel() for v in vs), device=dev)
        nflat = torch.empty(sum(v.shape[0] for v in vs), device=dev)
        ws, norms, entries, wo, no = [], [], [], 0, 0
        for v, g in zip(vs, gs):
            w, nr = wflat[wo:wo + v.numel()].view_as(v), nflat[no:no + v.shape[0]]
            wo, no = wo + v.numel(), no + v.shape[0]
            ws.append(w)
            norms.append(nr)
            entries.append((v.data_ptr(), g.data_ptr(), w.data_ptr(), nr.data_ptr(), v.shape[0], v.numel() // v.shape[0]))
        table, total = _WeightNormAllFunction._table(entries, dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_weight_norm_forward(_lib.ptr(table), len(vs), total, _lib.current_stream(dev)))
        ctx.save_for_backward(nflat, *vs, *gs)
        ctx.n = len(vs)
        return tuple(ws)

    @staticmethod
    def backward(ctx, *dws):
        L = _lib.load()
        nflat, *rest = ctx.saved_tensors
        vs, gs = rest[:ctx.n], rest[ctx.n:]
        dev = vs[0].device
        dws = [d.float().contiguous() for d in dws]
        dvflat = torch.empty(sum(v.numel() for v in vs), device=dev)
        dgflat = torch.empty(nflat.numel(), device=dev)
        ins, outs, out, wo, no = [], [], [], 0, 0
        for v, g, dw in zip(vs, gs, dws):
            dv, dg, nr = dvflat[wo:wo + v.numel()].view_as(v), dgflat[no:no + v.shape[0]].view_as(g), nflat[no:no + v.shape[0]]
            wo, no = wo + v.numel(), no + v.shape[0]
            ins.append((v.data_ptr(), g.data_ptr(), dw.data_ptr(), nr.data_ptr(), v.shape[0], v.numel() // v.shape[0]))
            outs.append((dv.data_ptr(), dg.data_ptr(), 0, 0, v.shape[0], v.numel() // v.shape[0]))
            out += [dv, dg]
        tin, total = _WeightNormAllFunction._table(ins, dev)
        tout, _ = _WeightNormAllFunction._table(outs, dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_weight_norm_backward(_lib.ptr(tin), _lib.ptr(tout), ctx.n, total, _lib.current_stream(dev)))
        return tuple(out)


class _AffineFunction(torch.autograd.Function):
    """The affine coupling of one flow in the training direction (glow.py:240-245) as one HIP kernel each way:
    (x = [x0 | x1], wn_out = [b | log_s]) -> cat(x0, exp(log_s) * x1 + b), log_s.
    log_s (the upper half of wn_out, which the loss sums) is an OUTPUT of this node rather than a slice the caller takes of
    wn_out: its gradient arrives here and is added into the kernel's d(wn_out) in place -- instead of a slice backward (a zero
    fill + a copy of the full tensor) and the engine's add of wn_out's two gradients, per flow."""

    @staticmethod
    def forward(ctx, x, wn_out):
        L = _lib.load()
        x, wn_out = x.detach().float().contiguous(), wn_out.detach().float().contiguous()
        y = torch.empty_like(x)
        B, c, Lg = x.shape
        with torch.cuda.device(x.device):
            _lib.check(L.facppg_affine_forward(_lib.ptr(x), _lib.ptr(wn_out), _lib.ptr(y), B, c // 2, Lg, _lib.current_stream(x.device)))
        ctx.save_for_backward(x, wn_out)
        return y, wn_out[:, c // 2:, :]

    @staticmethod
    def backward(ctx, dy, dlog_s):
        L = _lib.load()
        x, wn_out = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx, dwn = torch.empty_like(x), torch.empty_like(wn_out)
        B, c, Lg = x.shape
        with torch.cuda.device(x.device):
            _lib.check(L.facppg_affine_backward(_lib.ptr(x), _lib.ptr(wn_out), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dwn), B, c // 2, Lg,
                                                _lib.current_stream(x.device)))
        if dlog_s is not None:
            dwn[:, c // 2:, :] += dlog_s
        return dx, dwn


class Invertible1x1Conv(torch.nn.Module):
    """glow.py:62-102: parameter container for the c x c mixing matrix.  In inference the
    inverse matrix is applied inside k_flow_end (fused with the affine coupling)."""

    def __init__(self, c):
        super(Invertible1x1Conv, self).__init__()
        self.conv = torch.nn.Conv1d(c, c, kernel_size=1, stride=1, padding=0, bias=False)
        W = torch.linalg.qr(torch.randn(c, c))[0]        # random orthonormal init
        if torch.det(W) < 0:
            W[:, 0] = -1 * W[:, 0]                       # det = +1
        self.conv.weight.data = W.view(c, c, 1).contiguous()

    def inverse_matrix(self):
        """W^-1 as the reference computes and caches it (glow.py:88-95); the cache is keyed by the weight's
        storage, in-place version and device so an optimizer step / .to() never leaves a stale inverse."""
        W = self.conv.weight
        key = (W.data_ptr(), W._version, W.device)
        cached = self.__dict__.get("_W_inverse")
        if cached is None or cached[0] != key:
            cached = (key, W.detach().squeeze(-1).float().inverse())
            self.__dict__["_W_inverse"] = cached
        return cached[1]

    @property
    def W_inverse(self):                                 # the reference's attribute (glow.py:93), [c, c, 1]
        return self.inverse_matrix()[..., None]

    def forward(self, z, reverse=False):
        """glow.py:82-102 as a stand-alone module: z [B, c, L] (GPU) -> W^-1 z when ``reverse``, else
        (W z, B * L * log|det W|); runs the flow-edge channel-mixing kernel (facppg_conv1x1)."""
        _lib.require_cuda(z, "Invertible1x1Conv.forward: z")
        W = self.conv.weight.squeeze(-1)
        if reverse:
            return _conv1x1(self.inverse_matrix(), z)
        log_det_W = z.size(0) * z.size(2) * _LogDetFunction.apply(W.float())
        if torch.is_grad_enabled() and (W.requires_grad or z.requires_grad):
            return _Conv1x1Function.apply(W.float(), z.float()), log_det_W
        return _conv1x1(W, z), log_det_W


class WN(torch.nn.Module):
    """glow.py:105-175: parameter container with the reference's layer layout."""

    def __init__(self, n_in_channels, n_mel_channels, n_layers, n_channels, kernel_size):
        super(WN, self).__init__()
        assert kernel_size % 2 == 1
        assert n_channels % 2 == 0
        self.n_layers = n_layers
        self.n_channels = n_channels
        self.in_layers = torch.nn.ModuleList()
        self.res_skip_layers = torch.nn.ModuleList()
        self.cond_layers = torch.nn.ModuleList()
        wn = torch.nn.utils.weight_norm
        self.start = wn(torch.nn.Conv1d(n_in_channels, n_channels, 1), name='weight')
        end = torch.nn.Conv1d(n_channels, 2 * n_in_channels, 1)
        end.weight.data.zero_()                          # coupling starts as identity
        end.bias.data.zero_()
        self.end = end
        for i in range(n_layers):
            dilation = 2 ** i
            padding = (kernel_size * dilation - dilation) // 2
            self.in_layers.append(wn(torch.nn.Conv1d(
                n_channels, 2 * n_channels, kernel_size, dilation=dilation, padding=padding), name='weight'))
            self.cond_layers.append(wn(torch.nn.Conv1d(n_mel_channels, 2 * n_channels, 1), name='weight'))
            rs = 2 * n_channels if i < n_layers - 1 else n_channels
            self.res_skip_layers.append(wn(torch.nn.Conv1d(n_channels, rs, 1), name='weight'))

    def _weight_convs(self):
        """The convs in the order of the plain weight list (minus the un-normed end conv)."""
        convs = [self.start]
        for i in range(self.n_layers):
            convs += [self.in_layers[i], self.cond_layers[i], self.res_skip_layers[i]]
        return convs

    def _plain_weights(self, effective=None):
        """[start.w, start.b, (in.w, in.b, cond.w, cond.b, res_skip.w, res_skip.b) per layer, end.w, end.b];
        ``effective``: the convs' weights already de-normalised (WaveGlow batches that over all flows)."""
        convs = self._weight_convs()
        eff = effective if effective is not None else [_effective_weight(c) for c in convs]
        ws = []
        for conv, w in zip(convs, eff):
            ws += [w, conv.bias]
        return ws + [self.end.weight, self.end.bias]

    def _check_kernel_config(self):
        """The fused layer kernels are built for one WN shape (config.json:36-40 with n_group 8, 80 mel bins);
        anything else would read weights and saved activations with the wrong strides."""
        n_in, n_cond = self.start.in_channels, self.cond_layers[0].in_channels
        ks = self.in_layers[0].kernel_size[0]
        if not (self.n_channels == 256 and ks == 3 and 1 <= self.n_layers <= 8 and n_cond == 640 and 1 <= n_in <= 4):
            raise _lib.FacppgError(
                "WN on the HIP kernels needs n_channels=256, kernel_size=3, n_layers<=8, 640 conditioning channels "
                "(n_mel_channels*n_group) and n_in<=4; got n_channels=%d kernel_size=%d n_layers=%d cond=%d n_in=%d"
                % (self.n_channels, ks, self.n_layers, n_cond, n_in))

    def forward(self, forward_input):
        """glow.py:154-175 as a stand-alone module: (audio [B, n_in, L], spect [B, 640, L]) on the GPU ->
        [B, 2*n_in, L]; the fused k_wn_layer launches of facppg_wn_forward_save, differentiable through
        facppg_wn_backward_data (the same autograd node the training step uses per flow)."""
        audio, spect = forward_input
        _lib.require_cuda(audio, "WN.forward: audio")
        _lib.require_cuda(spect, "WN.forward: spect")
        self._check_kernel_config()
        Lg = audio.size(2)
        if spect.size(2) != Lg or spect.size(0) != audio.size(0):
            raise _lib.FacppgError("WN.forward: audio %s and spect %s disagree" % (tuple(audio.shape), tuple(spect.shape)))
        if getattr(self, "train_precision", "fp32") == "bf16":
            spect = spect.float().contiguous()
            return _WNFunctionBf16.apply(audio.float().contiguous(), spect, None, spect_to_posmajor_bf16(spect, Lg), *self._plain_weights())
        spect_pad = torch.nn.functional.pad(spect.float(), (0, -(-Lg // _TN) * _TN - Lg)).contiguous()
        return _WNFunction.apply(audio.float().contiguous(), spect_pad, *self._plain_weights())


_INFER_SIDE_STREAMS = {}


def _infer_side_stream(device):
    """The second HIP stream WaveGlow.infer runs the other half of a ragged batch on (one per device, private to this module:
    every use starts by waiting for the caller's stream and ends with the caller's stream waiting for it)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _INFER_SIDE_STREAMS:
        _INFER_SIDE_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return _INFER_SIDE_STREAMS[key]


class WaveGlow(torch.nn.Module):
    """glow.py:178-303."""

    def __init__(self, n_mel_channels, hop_length, n_flows, n_group, n_early_every,
                 n_early_size, WN_config):
        super(WaveGlow, self).__init__()
        self.upsample = torch.nn.ConvTranspose1d(n_mel_channels, n_mel_channels, 1024, stride=hop_length)
        assert n_group % 2 == 0
        self.n_flows = n_flows
        self.n_group = n_group
        self.n_early_every = n_early_every
        self.n_early_size = n_early_size
        self.WN = torch.nn.ModuleList()
        self.convinv = torch.nn.ModuleList()
        n_half = n_group // 2
        n_remaining_channels = n_group
        for k in range(n_flows):
            if k % self.n_early_every == 0 and k > 0:
                n_half = n_half - self.n_early_size // 2
                n_remaining_channels = n_remaining_channels - self.n_early_size
            self.convinv.append(Invertible1x1Conv(n_remaining_channels))
            self.WN.append(WN(n_half, n_mel_channels * n_group, **WN_config))
        self.n_remaining_channels = n_remaining_channels

    @property
    def train_precision(self):
        """Operand precision of the training step's MFMA products: 'fp32' (exact, the default: bit-compatible with
        the reference's fp32 training) or 'bf16' (BASELINE config 5: bf16 operands, fp32 accumulation, fp32 master
        weights and gradients).  Set the attribute, or FACPPG_TRAIN_PRECISION in the environment."""
        import os
        return self.__dict__.get("_train_precision") or os.environ.get("FACPPG_TRAIN_PRECISION", "fp32")

    @train_precision.setter
    def train_precision(self, value):
        if value not in ("fp32", "bf16"):
            raise ValueError("train_precision must be 'fp32' or 'bf16'")
        self.__dict__["_train_precision"] = value

    # ---------------------------------------------------------------- HIP handle management
    def _config(self):
        cfg = _lib.WgConfig()
        cfg.n_mel_channels = self.upsample.in_channels
        cfg.hop_length = self.upsample.stride[0]
        cfg.n_flows, cfg.n_group = self.n_flows, self.n_group
        cfg.n_early_every, cfg.n_early_size = self.n_early_every, self.n_early_size
        cfg.wn_layers = self.WN[0].n_layers
        cfg.wn_channels = self.WN[0].n_channels
        cfg.wn_kernel_size = self.WN[0].in_layers[0].kernel_size[0]
        cfg.upsample_kernel = self.upsample.kernel_size[0]
        cfg.alternate_halves = 1 if getattr(self, "_alternate_halves", False) else 0
        return cfg

    def _flat_weights(self):
        """The plain weight blob in the order include/facppg.h documents."""
        parts = [self.upsample.weight, self.upsample.bias]
        for k in range(self.n_flows):
            wn = self.WN[k]
            parts += [_effective_weight(wn.start), wn.start.bias]
            for i in range(wn.n_layers):
                parts += [_effective_weight(wn.in_layers[i]), wn.in_layers[i].bias,
                          _effective_weight(wn.cond_layers[i]), wn.cond_layers[i].bias,
                          _effective_weight(wn.res_skip_layers[i]), wn.res_skip_layers[i].bias]
            parts += [wn.end.weight, wn.end.bias, self.convinv[k].inverse_matrix(), self.convinv[k].conv.weight.squeeze(-1)]
        return torch.cat([p.detach().float().reshape(-1) for p in parts])

    def _release(self):
        self.__dict__.pop("_facppg_prepared", None)
        self.__dict__.pop("_facppg_cond_stream", None)    # its buffers belong to the handle's weights and device
        h = self.__dict__.pop("_facppg_handle", None)
        if h is not None:
            _lib.load().facppg_wg_destroy(h[0])
        self.__dict__.pop("_facppg_ws", None)

    def last_launch_shape(self):
        """(frames per tile, waves per workgroup, workgroups per launch) of the WN-layer kernels of the most recent
        infer() -- which instantiation of the fused layer kernel ran (facppg_wg_last_launch_shape)."""
        h = self.__dict__.get("_facppg_handle")
        if h is None:
            raise _lib.FacppgError("last_launch_shape: no inference has run on this model yet")
        c = _lib.ctypes
        tile, waves, tiles = c.c_int(0), c.c_int(0), c.c_int(0)
        _lib.check(_lib.load().facppg_wg_last_launch_shape(h[0], c.byref(tile), c.byref(waves), c.byref(tiles)))
        return tile.value, waves.value, tiles.value

    def invalidate_packed_weights(self):
        """Forget the packed MFMA weight images; the next infer()/forward() repacks from the live parameters."""
        self._release()

    def train(self, mode=True):
        self._release()
        return super(WaveGlow, self).train(mode)

    def _handle(self, device):
        h = self.__dict__.get("_facppg_handle")
        if h is not None and h[1] == device and h[2].unchanged():
            return h[0]
        self._release()
        L = _lib.load()
        cfg = self._config()
        blob = self._flat_weights().to(device).contiguous()
        if blob.numel() != L.facppg_wg_weight_count(cfg):
            raise _lib.FacppgError("weight blob has %d values, library expects %d (unsupported config: %s)" % (
                blob.numel(), L.facppg_wg_weight_count(cfg), L.facppg_last_error().decode()))
        out = _lib.ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.facppg_wg_create(cfg, _lib.ptr(blob), blob.numel(), device.index,
                                          _lib.current_stream(device), _lib.ctypes.byref(out)))
        self.__dict__["_facppg_handle"] = (out, device, _lib.WeightIdentity(self))
        return out

    def _apply(self, fn, *a, **k):                       # .cuda()/.to()/.float(): weights moved
        self._release()
        return super(WaveGlow, self)._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._release()
        return super(WaveGlow, self).load_state_dict(*a, **k)

    def __getstate__(self):                              # never pickle device handles
        d = dict(self.__dict__)
        d.pop("_facppg_handle", None)
        d.pop("_facppg_prepared", None)
        d.pop("_facppg_ws", None)
        d.pop("_facppg_cond_stream", None)                # (facppg.pipeline.ConditioningStream: streams, events, GBs of seeds)
        return d

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---------------------------------------------------------------- the hot path
    def _forward_autograd(self, spect, audio):
        """Training forward with a differentiable graph (train_waveglow.py:126-133).  Per flow the WN
        stack -- >99 % of the work -- is one HIP autograd node (_WNFunction fp32 / _WNFunctionBf16); the
        flow edges are HIP nodes too (1x1 mixing conv: _Conv1x1Function, affine coupling: _AffineFunction; in bf16 mode
        the upsampler as well: _UpsampleBf16Function).  What stays in torch is bookkeeping on <= 8-channel tensors: the
        early-output slices, the final cat, logdet of the c x c matrices and the weight-norm parametrisation."""
        F = torch.nn.functional
        g = self.n_group
        self._release()                      # the weights are about to be trained: never serve a stale packed copy
        if _lib.load().facppg_wg_weight_count(self._config()) == 0:      # same check the inference handle makes
            raise _lib.FacppgError("unsupported WaveGlow config for the HIP training kernels: %s"
                                   % _lib.load().facppg_last_error().decode())
        for wn in self.WN:
            wn._check_kernel_config()
        hop, ksz = self.upsample.stride[0], self.upsample.kernel_size[0]
        bf16 = self.train_precision == "bf16"
        audio = audio[:, :audio.size(1) - audio.size(1) % g]
        Lg = audio.size(1) // g
        if bf16:
            # upsample + crop + regroup on HIP, straight into the bf16 position-major operand every flow reads; the
            # flows add their conditioning gradients to one shared fp32 buffer that the upsampler's backward consumes
            if (spect.size(2) - 1) * hop + ksz < audio.size(1):
                raise _lib.FacppgError("upsampled mel is shorter than the audio (glow.py:216)")
            shared = _StepShared()
            spect_pm, link = _UpsampleBf16Function.apply(spect, self.upsample.weight, self.upsample.bias, shared, hop, Lg)
        else:
            # ConvTranspose1d as a matrix product + overlap-add (col2im): both are differentiable torch ops that
            # run on rocBLAS / a native fold kernel; MIOpen's transposed-conv backward falls back to a naive
            # kernel here that costs more than the rest of the step together
            Bm, nm, Tm = spect.shape
            cols = torch.einsum('bit,ijk->bjkt', spect, self.upsample.weight).reshape(Bm, nm * ksz, Tm)
            spect = F.fold(cols, output_size=(1, (Tm - 1) * hop + ksz), kernel_size=(1, ksz), stride=(1, hop)).squeeze(2)
            spect = spect + self.upsample.bias.view(1, -1, 1)
            assert spect.size(2) >= audio.size(1)
            spect = spect[:, :, :audio.size(1)]
            spect = spect.unfold(2, g, g).permute(0, 2, 1, 3)
            spect = spect.contiguous().view(spect.size(0), spect.size(1), -1).permute(0, 2, 1)
            spect_pad = F.pad(spect, (0, -(-Lg // _TN) * _TN - Lg)).contiguous()
        audio = audio.unfold(1, g, g).permute(0, 2, 1)
        # effective weights of the weight-normed convs: one launch (and one more in the backward) per GROUP of flows.  The
        # groups are the gradient buckets of waveglow.distributed.plan_buckets: a group's (v, g) gradients -- 99 % of the
        # gradient bytes -- leave its backward launch as soon as the backward pass is through with those flows, so the
        # data-parallel all-reduce of one bucket runs under the backward of the earlier flows
        ngr = max(1, min(self.n_flows, int(getattr(self, "weight_norm_groups", 3))))
        flow_weights = [None] * self.n_flows
        for gi in range(ngr):
            ks = [k for k in range(self.n_flows) if k * ngr // self.n_flows == gi]
            convs = [c for k in ks for c in self.WN[k]._weight_convs()]
            if all(hasattr(c, "weight_g") for c in convs):
                vg = [t for c in convs for t in (c.weight_v, c.weight_g)]
                eff = list(_WeightNormAllFunction.apply(*vg))
            else:
                eff = [_effective_weight(c) for c in convs]
            per = len(eff) // len(ks)
            for j, k in enumerate(ks):
                flow_weights[k] = self.WN[k]._plain_weights(eff[j * per:(j + 1) * per])
        output_audio, log_s_list, log_det_W_list = [], [], []
        if bf16 and os.environ.get("FACPPG_TRAIN_FLOW_NODES", "0") != "1":
            # one autograd node per weight-norm group of flows; the early outputs and the last flow's output land in ONE buffer (z)
            groups = [[k for k in range(self.n_flows) if k * ngr // self.n_flows == gi] for gi in range(ngr)]
            step = _GlowStepBf16(self, flow_weights, audio.size(0), Lg, audio.device, shared)
            for gi, ks in enumerate(groups):
                tensors = [self.convinv[k].conv.weight.squeeze(-1) for k in ks] + [w for k in ks for w in flow_weights[k]]
                outs = _FlowGroupBf16Function.apply(audio, link, step, ks[0], len(ks), spect_pm, gi == ngr - 1, *tensors)
                ne = sum(1 for k in ks if step.plan[k][1])
                audio = outs[0]
                output_audio += outs[1:1 + ne]
                log_s_list += outs[1 + ne:1 + ne + len(ks)]
                log_det_W_list += outs[1 + ne + len(ks):]
            output_audio.append(audio)
            return _AssembleZFunction.apply(step, *output_audio), log_s_list, log_det_W_list
        for k in range(self.n_flows):
            if k % self.n_early_every == 0 and k > 0:
                output_audio.append(audio[:, :self.n_early_size, :])
                audio = audio[:, self.n_early_size:, :]
            W = self.convinv[k].conv.weight.squeeze(-1)
            log_det_W_list.append(audio.size(0) * audio.size(2) * _LogDetFunction.apply(W.float()))   # glow.py:100, one HIP launch
            audio = _Conv1x1Function.apply(W.float(), audio.contiguous())        # 1x1 mixing conv (c <= 8 channels), HIP fwd + bwd
            n_half = audio.size(1) // 2
            audio_0 = audio[:, :n_half, :]
            if bf16:
                output = _WNFunctionBf16.apply(audio_0.contiguous(), link, shared, spect_pm, *flow_weights[k])
            else:
                output = _WNFunction.apply(audio_0.contiguous(), spect_pad, *flow_weights[k])
            audio, log_s = _AffineFunction.apply(audio, output)   # cat(audio_0, exp(log_s) * audio_1 + b) and log_s, HIP fwd + bwd
            log_s_list.append(log_s)
        output_audio.append(audio)
        return torch.cat(output_audio, 1), log_s_list, log_det_W_list

    def forward(self, forward_input):
        """(mel [B, n_mel, F], audio [B, N]) -> (z [B, n_group, N/n_group], log_s_list, log_det_W_list)
        -- the training direction, glow.py:208-250.  With gradients enabled this builds an autograd
        graph whose heavy nodes run on the HIP kernels (see _forward_autograd); under no_grad it is one
        fused facppg_wg_forward call."""
        spect, audio = forward_input
        _lib.require_cuda(spect, "WaveGlow.forward: spect")
        _lib.require_cuda(audio, "WaveGlow.forward: audio")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_autograd(spect.float(), audio.float())
        L = _lib.load()
        dev = spect.device
        spect = spect.float().contiguous()
        audio = audio.float()
        audio = audio[:, :audio.shape[1] - audio.shape[1] % self.n_group].contiguous()   # unfold drops the tail (glow.py:224)
        B, _, F = spect.shape
        N = audio.shape[1]
        hop = self.upsample.stride[0]
        h = self._handle(dev)
        nbytes = L.facppg_wg_workspace_bytes(h, B, (N + hop - 1) // hop)
        wss = self.__dict__.setdefault("_facppg_ws", {})
        ws = wss.get(0)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            wss[0] = ws
        Lg = N // self.n_group
        z = torch.empty(B, self.n_group, Lg, device=dev)
        log_s_flat = torch.empty(L.facppg_wg_log_s_count(h, B, N), device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_wg_forward(h, _lib.ptr(spect), _lib.ptr(audio), B, F, N, _lib.ptr(z), _lib.ptr(log_s_flat),
                                           _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))
        log_s_list, log_det_W_list, off = [], [], 0
        for k in range(self.n_flows):
            hk = self.convinv[k].conv.weight.shape[0] // 2
            n = B * hk * Lg
            log_s_list.append(log_s_flat[off:off + n].view(B, hk, Lg))
            off += n
            W = self.convinv[k].conv.weight.squeeze(-1).float()
            log_det_W_list.append(B * Lg * _LogDetFunction.apply(W))  # glow.py:100
        return z, log_s_list, log_det_W_list

    def draw_noise(self, utterance_seeds, T, device=None):
        """Per-utterance N(0,1) streams (facppg_wg_draw_noise) for a batch of len(utterance_seeds) mels of T frames,
        flat in the injected-z layout ([B, n_remaining, L] ++ [B, n_early, L] per early output): the values of
        utterance b depend on utterance_seeds[b] only."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        L = _lib.load()
        h = self._handle(dev)
        B, hop = len(utterance_seeds), self.upsample.stride[0]
        sd = _lib.upload([int(v) & 0x7FFFFFFFFFFFFFFF for v in utterance_seeds], torch.int64, dev)
        zt = torch.empty(B * self.n_group * (T * hop // self.n_group), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_wg_draw_noise(h, _lib.ptr(sd), B, T, _lib.ptr(zt), _lib.current_stream(dev)))
        return zt

    # ---- two concurrent half-batches for ragged batches
    _ROUND_COSTS = ((64, 331, 185), (32, 181, 94), (16, 105, 57))   # frames per tile, us per full / half round (csrc/facppg_wg.hip)

    def _tail_loss(self, lengths, hop):
        """Fraction of a WN-layer launch of this ragged batch that the chip idles through in its last round of workgroup
        slots, from the library's measured round costs (512 slots = 2 per CU); 0 for a launch that does not fill the slots
        once (its problem is not the tail, and splitting it only adds launches)."""
        P = hop // 8
        best = None
        for frames, full, half in self._ROUND_COSTS:
            cols = sum(-(-int(n) // 4) * 4 for n in lengths)
            tiles = P * -(-cols // frames)
            rem = tiles % 512
            cost = (tiles // 512) * full + (0 if rem == 0 else half if rem <= 256 else full)
            if best is None or cost < best[0]:
                best = (cost, tiles / 512.0 * full, tiles)
        return 1.0 - best[1] / best[0] if best[0] and best[2] > 512 else 0.0

    def _infer_two_groups(self, spect, sigma, zt, lengths, seed, utterance_seeds):
        """A ragged batch as TWO interleaved half-batches, the second on a side stream.  Utterances are independent and every
        one is computed exactly as in its own batch-1 call, so the split changes no bit of the result; what it changes is
        the tail of every layer launch: a launch runs in rounds of 512 workgroup slots and its last, partial round leaves
        most of the chip idle (16 utterances of 100-400 frames: 4.3 rounds, 6 % of the time) -- with two launch sequences in
        flight, the free slots of one half's last round are taken by the other half's tiles.  Measured on the 16-utterance
        batch of BASELINE config 3: 127.1 -> 120.2 ms (profiles/r03_experiments.txt); three or four groups gain nothing.
        Both halves' inputs are prepared on the caller's stream BEFORE the fork (every host-blocking upload happens there),
        then the two launch sequences are enqueued back to back without touching the host clock again."""
        dev, hop, g8 = spect.device, self.upsample.stride[0], self.n_group
        B, _, T = spect.shape
        order = sorted(range(B), key=lambda i: (-int(lengths[i]), i))
        side = _infer_side_stream(dev)
        cur = torch.cuda.current_stream(dev)
        out = torch.zeros(B, T * hop, dtype=torch.float32, device=dev)
        Lfull = T * hop // g8
        segs = None
        if zt is not None:          # the injected-z layout: [B, n_remaining, L] ++ [B, n_early_size, L] per early output
            chans = [self.n_remaining_channels] + [self.n_early_size] * ((g8 - self.n_remaining_channels) // self.n_early_size)
            segs, off = [], 0
            for c in chans:
                segs.append(zt[off:off + B * c * Lfull].view(B, c, Lfull))
                off += B * c * Lfull
        sel = _lib.upload(order[0::2] + order[1::2] + [int(lengths[i]) for i in order[0::2] + order[1::2]], torch.int64, dev)
        parts, n0 = [], len(order[0::2])
        for gi, idx in enumerate((order[0::2], order[1::2])):
            it = sel[:n0] if gi == 0 else sel[n0:B]
            lt = (sel[B:B + n0] if gi == 0 else sel[B + n0:]).to(torch.int32)
            Tg = max(int(lengths[i]) for i in idx)
            mel_g = spect.index_select(0, it)[:, :, :Tg].contiguous()
            if utterance_seeds is not None:
                z_g = self.draw_noise([utterance_seeds[i] for i in idx], Tg, dev)
            elif segs is not None:
                z_g = torch.cat([sg.index_select(0, it)[:, :, :Tg * hop // g8].reshape(-1) for sg in segs])
            else:
                z_g = None
            a = torch.zeros(len(idx), Tg * hop, dtype=torch.float32, device=dev)
            ws = self._infer_workspace(len(idx), Tg, dev, gi)
            parts.append((it, lt, Tg, mel_g, z_g, a, ws))
        side.wait_stream(cur)
        for gi, (it, lt, Tg, mel_g, z_g, a, ws) in enumerate(parts):
            with torch.cuda.stream(side if gi else cur):
                self._infer_launch(mel_g, lt, z_g, (seed + 0x9E3779B97F4A7C15 * gi) & 0x7FFFFFFFFFFFFFFF, sigma, a, ws)
        cur.wait_stream(side)
        for it, lt, Tg, mel_g, z_g, a, ws in parts:
            out[:, :Tg * hop].index_copy_(0, it, a)
        return out

    def _infer_workspace(self, B, T, dev, slot, handle=None):
        nbytes = _lib.load().facppg_wg_workspace_bytes(handle if handle is not None else self._handle(dev), B, T)
        wss = self.__dict__.setdefault("_facppg_ws", {})
        ws = wss.get(slot)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            wss[slot] = ws
        return ws

    def _infer_launch(self, spect, lt, zt, seed, sigma, audio, ws, handle=None):
        """The launch sequence of one batch on the current stream; no host-side waits."""
        dev = spect.device
        B, _, T = spect.shape
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_wg_infer(handle if handle is not None else self._handle(dev), _lib.ptr(spect), _lib.ptr(lt), _lib.ptr(zt),
                                                   seed & 0xFFFFFFFFFFFFFFFF, float(sigma), B, T, _lib.ptr(audio),
                                                   _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)))

    # ---- seeded inference of ONE utterance: the conditioning part of every layer's gate GEMM formed ahead of time
    def seed_layout(self, T, device):
        """(Tqp, margin, seed_bytes) for an utterance of T frames: the zero-margined mel buffer is [n_mel, Tqp] with frame q at
        column margin + q; the seed buffer has seed_bytes bytes (facppg_wg_seed_layout)."""
        c = _lib.ctypes
        tqp, mg, nb = c.c_int(), c.c_int(), c.c_size_t()
        _lib.check(_lib.load().facppg_wg_seed_layout(self._handle(device), int(T), c.byref(tqp), c.byref(mg), c.byref(nb)))
        return tqp.value, mg.value, nb.value

    def mel_pad(self, mel, handle=None):
        """mel [1, n_mel, T] -> the zero-margined [n_mel, Tqp] buffer cond_seed / infer_seeded read."""
        dev = mel.device
        T = mel.shape[2]
        tqp, _, _ = self.seed_layout(T, dev)
        out = torch.empty(mel.shape[1], tqp, dtype=torch.float32, device=dev)
        m = mel[0]
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_wg_mel_pad(self._handle(dev), _lib.ptr(m), T, m.stride(0), _lib.ptr(out), _lib.current_stream(dev)))
        return out

    def cond_seed(self, melp, T, frame0, nframes, seeds, block_tiles=1, layers_per_workgroup=4, skip=None, handle=None, flows=None,
                  max_workgroups=0, counter=None):
        """Form the gate accumulators' seeds (bias + conditioning sums, k_cond_seed) of frames [frame0, frame0 + nframes) of the
        utterance whose zero-margined mel frames are ``melp``, for every flow (or flows = (first, count)), layer and phase, on the
        current stream.  max_workgroups > 0 bounds the launch (its workgroups then take the work items from ``counter``, a zeroed
        int32 on the device): the CUs it does not fill stay free for whoever else needs one right away."""
        dev = melp.device
        f0, nf = flows if flows is not None else (0, 0)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_wg_cond_seed(handle if handle is not None else self._handle(dev), _lib.ptr(melp), int(T), int(frame0),
                                                       int(nframes), int(block_tiles), int(layers_per_workgroup), int(f0), int(nf), _lib.ptr(seeds),
                                                       seeds.numel() * seeds.element_size(), _lib.ptr(skip), int(max_workgroups), _lib.ptr(counter),
                                                       _lib.current_stream(dev)))

    def infer_seeded(self, melp, T, seeds, seeded_frames, sigma=1.0, z=None, seed=None, handle=None, T_layout=None, flow_events=None):
        """WaveGlow.infer of ONE utterance (glow.py:252-293) whose layers start from ``seeds``: audio [1, T*hop].  Same samples
        as infer() on the same mel frames, bit for bit.  T_layout >= T: the frame count ``melp`` and ``seeds`` were laid out for.
        flow_events: {flow: torch.cuda.Event} the launches of that flow wait for (its seeds are still being formed elsewhere)."""
        T_layout = T if T_layout is None else int(T_layout)
        dev = melp.device
        hop = self.upsample.stride[0]
        zt = None
        if z is not None:
            if isinstance(z, (list, tuple)):
                z = torch.cat([t.to(dev).float().reshape(-1) for t in z])
            zt = z.to(dev).float().contiguous()
            if zt.numel() != self.n_group * (T * hop // self.n_group):
                raise _lib.FacppgError("z has %d values, expected n_group*L = %d" % (zt.numel(), self.n_group * (T * hop // self.n_group)))
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        h = handle if handle is not None else self._handle(dev)
        ws = self._infer_workspace(1, T_layout, dev, 0, h)
        audio = torch.empty(1, T * hop, dtype=torch.float32, device=dev)
        evs = None
        if flow_events:
            evs = (_lib.ctypes.c_void_p * self.n_flows)()
            for k, ev in flow_events.items():
                evs[k] = ev.cuda_event
        with torch.cuda.device(dev):
            _lib.check(_lib.load().facppg_wg_infer_seeded(h, _lib.ptr(melp), T_layout, int(T), _lib.ptr(seeds), int(seeded_frames), _lib.ptr(zt),
                                                          seed & 0xFFFFFFFFFFFFFFFF, float(sigma), _lib.ptr(audio), _lib.ptr(ws), ws.numel(),
                                                          evs, _lib.current_stream(dev)))
        return audio

    def prepare(self, device):
        """Validate (or build) the packed weights for ``device`` NOW and remember that for the next ``infer`` on this thread's
        next call: the check walks ~1000 tensors (0.4 ms of host time); facppg.pipeline runs it while the acoustic model's
        decoder keeps the GPU busy instead of between the two models.  Single use: the next infer() -- whatever path it takes --
        consumes it, and it is only honoured while the handle it validated is still the model's handle."""
        self.__dict__["_facppg_prepared"] = (self._handle(device), device)

    def _checked_handle(self, dev):
        pre = self.__dict__.pop("_facppg_prepared", None)
        cur = self.__dict__.get("_facppg_handle")
        if pre is not None and pre[1] == dev and cur is not None and cur[0] is pre[0]:
            return pre[0]
        return self._handle(dev)

    def infer(self, spect, sigma=1.0, z=None, lengths=None, seed=None, utterance_seeds=None, groups=None):
        """mel [B, n_mel, T] (GPU, fp32) -> audio [B, T*hop]   (glow.py:252-293).
        groups: None = decide from the launch shape (ragged batches whose layer launches would idle through >= 3 % of their
        time in the last round run as two concurrent half-batches, see _infer_two_groups), 1 = one launch sequence, 2 = force
        the two half-batches (needs host-side lengths).  With `seed` alone the noise of a uniform batch is a function of (seed, batch
        layout); a ragged batch with host-side lengths draws per-utterance streams derived from (seed, b), identical in both modes."""
        _lib.require_cuda(spect, "WaveGlow.infer: spect")
        if spect.dtype != torch.float32:
            raise _lib.FacppgError("WaveGlow.infer: fp32 only (the reference's fp16 branch is not built)")
        dev = spect.device
        h = self._checked_handle(dev)   # (ONE validity check of the packed weights per call -- it walks ~1000 tensors -- or none:
        spect = spect.contiguous()      #  prepare(); consumed HERE so that no path -- two half-batches included -- leaves the token behind)
        B, _, T = spect.shape
        hop = self.upsample.stride[0]
        host_lengths = lengths is not None and not torch.is_tensor(lengths)
        if host_lengths and (len(lengths) != B or max(int(n) for n in lengths) > T or min(int(n) for n in lengths) < 1):
            raise _lib.FacppgError("lengths must be B values in [1, T]")
        if groups is None:
            groups = 2 if (host_lengths and B >= 4 and self._tail_loss(lengths, hop) >= 0.03) else 1
        if groups == 2 and (not host_lengths or B < 2):
            raise _lib.FacppgError("WaveGlow.infer: groups=2 needs B >= 2 utterances and their lengths as a host list")
        zt = None
        if utterance_seeds is not None:
            if z is not None or len(utterance_seeds) != B:
                raise _lib.FacppgError("utterance_seeds: B integers, and not together with z")
        elif z is not None:
            if isinstance(z, (list, tuple)):
                z = torch.cat([t.to(dev).float().reshape(-1) for t in z])
            zt = z.to(dev).float().contiguous()
            if zt.numel() != B * self.n_group * (T * hop // self.n_group):
                raise _lib.FacppgError("z has %d values, expected B*n_group*L = %d" % (
                    zt.numel(), B * self.n_group * (T * hop // self.n_group)))
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        if utterance_seeds is None and zt is None and host_lengths and B >= 2:
            # `seed` alone on a ragged batch: give every utterance its own stream derived from (seed, b), so that the noise --
            # and the audio -- do not depend on whether the launch-shape heuristic above picks one launch sequence or two
            # half-batches (whose batch layouts differ)
            utterance_seeds = [(int(seed) * 0x9E3779B97F4A7C15 + (b + 1) * 0xBF58476D1CE4E5B9) & 0x7FFFFFFFFFFFFFFF for b in range(B)]
        if groups == 2:
            return self._infer_two_groups(spect, sigma, zt, lengths, seed, utterance_seeds)
        if utterance_seeds is not None:
            zt = self.draw_noise(utterance_seeds, T, dev)
        lt = None
        if lengths is not None:
            if host_lengths:
                lt = _lib.upload([int(n) for n in lengths], torch.int32, dev)
            else:
                lt = lengths.to(device=dev, dtype=torch.int32).contiguous()
                if lt.numel() != B or int(lt.max()) > T or int(lt.min()) < 1:
                    raise _lib.FacppgError("lengths must be B values in [1, T]")
        audio = torch.zeros(B, T * hop, dtype=torch.float32, device=dev) if lt is not None else \
            torch.empty(B, T * hop, dtype=torch.float32, device=dev)
        self._infer_launch(spect, lt, zt, seed, sigma, audio, self._infer_workspace(B, T, dev, 0, h), h)
        return audio

    @staticmethod
    def remove_weightnorm(model):
        """glow.py:295-303: fold g*v/||v|| into plain weights (load-time only)."""
        waveglow = model
        for wn in waveglow.WN:
            wn.start = torch.nn.utils.remove_weight_norm(wn.start)
            wn.in_layers = remove(wn.in_layers)
            wn.cond_layers = remove(wn.cond_layers)
            wn.res_skip_layers = remove(wn.res_skip_layers)
        if hasattr(waveglow, "_release"):
            waveglow._release()
        return waveglow


def remove(conv_list):
    """glow.py:306-311"""
    out = torch.nn.ModuleList()
    for conv in conv_list:
        out.append(torch.nn.utils.remove_weight_norm(conv))
    return out
