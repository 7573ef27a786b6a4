"""Batched PPG -> mel -> wav synthesis (BASELINE configs 3 and 4).

The reference synthesises one utterance at a time (generate_synthesis.py:90-95).  ``synthesize``
runs the same three stages -- Tacotron2.inference, WaveGlow.infer, Denoiser -- on a padded batch
with per-utterance lengths threaded through every HIP call, so each utterance's result equals
its own batch-1 run; everything stays on the device until the final copy of the audio.
"""
import os
import threading

import numpy as np
import torch


class Synthesizer(object):
    """The three models of the synthesis path loaded from the reference's checkpoint formats and kept on the GPU:
    the PPG->mel state dict (``{'state_dict': ...}``, train_ppg2mel.py:113-119), the pickled WaveGlow module
    (``{'model': ...}``, train_waveglow.py:56-64) once with weight norm removed for synthesis
    (utils.py:177-181) and once as pickled for the denoiser's bias estimate (generate_synthesis.py:58-61)."""

    def __init__(self, ppg2mel_path, waveglow_path, hparams=None, denoiser_mode='zeros'):
        from common.hparams import create_hparams_stage
        from common.utils import load_waveglow_model
        from script.train_ppg2mel import load_model
        from waveglow.denoiser import Denoiser
        self.hparams = hparams if hparams is not None else create_hparams_stage()
        self.tacotron = load_model(self.hparams)
        self.tacotron.load_state_dict(torch.load(ppg2mel_path, weights_only=False)['state_dict'])
        self.tacotron.eval()
        self.denoiser = Denoiser(torch.load(waveglow_path, weights_only=False)['model'].cuda(), mode=denoiser_mode)
        self.waveglow = load_waveglow_model(waveglow_path)

    def __call__(self, ppgs, sigma=0.6, strength=0.005, **kw):
        return synthesize(ppgs, self.tacotron, self.waveglow, self.denoiser, sigma=sigma, strength=strength, **kw)

    def stream(self, jobs, sigma=0.6, strength=0.005, **kw):
        """synthesize_stream over this synthesizer's models: batch i+1's acoustic model under batch i's vocoder."""
        return synthesize_stream(jobs, self.tacotron, self.waveglow, self.denoiser, sigma=sigma, strength=strength, **kw)

    @staticmethod
    def has_utterance(utterance_path):
        """The reference checks the teacher wav itself (generate_synthesis.py:89); here a precomputed PPG next to
        it (or given directly) counts as well, since that is what this build reads."""
        from common.data_utils import ppg_candidates
        return any(os.path.isfile(c) for c in [utterance_path] + ppg_candidates(utterance_path))

    def synthesize_file(self, utterance_path, wav_path, fs, sigma=0.6, strength=0.005, ppg_deps=None):
        """One teacher utterance -> ``wav_path`` (float32 [N, 1] at ``fs``, generate_synthesis.py:90-98)."""
        from common.data_utils import get_ppg
        from scipy.io import wavfile
        wavs, tout = self([get_ppg(utterance_path, ppg_deps)], sigma=sigma, strength=strength)
        wavfile.write(wav_path, fs, wavs[0].astype(np.float32)[:, None])
        return tout[0]


class ConditioningStream(object):
    """One utterance at a time (the metric's "batch = 1" case): work that depends on the mel frames ALONE, done while the decoder
    is still producing them, on the ~180 CUs its launch leaves empty.

    The split decoder publishes every frame the moment it exists (facppg_taco_set_frame_stream); on a second HIP stream, gated
    on those frames (facppg_taco_collect_frames), this object runs per block of frames
      * the postnet as a streaming convolution stack (facppg_taco_postnet_range: same sums, same order, same bits as the one-shot
        launch) straight into the vocoder's zero-margined mel buffer, and
      * the conditioning part of every WaveNet layer's gate GEMM (facppg_wg_cond_seed: bias + the folded conditioning chunks, the
        MFMA sequence the layer kernel itself would run first) into a seed buffer,
    so that behind the decoder only the last frames' share is left, and WaveGlow's layer launches start from the seeds with 12 K
    chunks instead of 17 (first layers 1 instead of 6): 29 % of the vocoder's FLOPs off the critical path, same samples bit for
    bit (tests/test_gpu_stream.py).  Reference: Postnet.forward (model.py:178-184, 604-605), WN.forward's cond_layers
    (glow.py:154-175) and the upsampling they read (glow.py:253-259).

    A pass over a block of frames streams every (flow, layer, phase) conditioning image once -- 2 GB at hop 256 -- whatever the
    block's width (FACPPG_STREAM_CHUNK frames: 32 for utterances up to 320 frames, 64 beyond, the last one before the expected
    end always 32, which keeps the share that has to wait for the decoder's end small).  What the blocks cannot cover -- the frames
    that become final only when the decoder ends -- runs as UNSEEDED 16-frame tiles inside the vocoder's own layer launches
    (k_wn_layer_mixed) rather than as one more pass in front of them."""

    LAG = None   # frames of mel the postnet's output trails its input by (pad * layers; from the model)

    def __init__(self, tacotron, waveglow):
        self.tacotron, self.waveglow = tacotron, waveglow
        self.key = None
        self.active = False
        self.profile = False       # True: hipEvents around every seed pass of the following utterances (pass_ms)
        self.lock = threading.Lock()   # held by synthesize() for the whole utterance (buffers, streams and plan are per model pair)

    @staticmethod
    def usable(tacotron, waveglow):
        """The streamed path exists for the reference's shapes on the folded, phase-major vocoder kernels."""
        if os.environ.get("FACPPG_STREAM", "1") == "0" or os.environ.get("FACPPG_WG_UNFOLDED", "0") not in ("", "0"):
            return False
        if os.environ.get("FACPPG_WG_EDGE_FOLD", "1") == "0" or getattr(tacotron, "decoder_workgroups", 0):
            return False
        return waveglow.WN[0].n_layers == 8 and waveglow.n_group == 8

    SLACK = 64   # frames past the PPG's length the buffers are laid out for (an utterance that runs on past them is finished unstreamed)

    def _buffers(self, dev, steps, cap):
        """Views for one utterance: the frame words cover the decoder's step limit ``steps`` (it publishes every frame it makes), the
        vocoder-side buffers (zero-margined mel, seeds, the streaming postnet's layers) are LAID OUT for ``cap`` <= steps frames --
        the PPG's length plus SLACK, not max_decoder_steps: 64 KiB of seeds per (layer, phase, 32 frames) is 1.9 GB at 288 frames and
        6.4 GB at 1000.  The allocations only ever grow (an utterance of another length re-uses them under its own layout) and the
        two side streams are created once per device."""
        from facppg import lib as _lib
        L = _lib.load()
        hp = self.tacotron._hp
        self.NF = hp["n_acoustic_feat_dims"]
        self.lag = (hp["postnet_kernel_size"] - 1) // 2 * hp["postnet_n_convolutions"]
        if self.key != dev:
            self.store = {}
            # The seed passes fill every CU the decoder leaves; the postnet's launches next to them are a few workgroups each and must
            # not queue for a slot behind thousands of the pass's own (measured: 260 us for a 22 us launch): the seed stream gets the
            # lowest priority, the postnet stream the highest (FACPPG_STREAM_PRIO=0: both default).
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, 0)
            if os.environ.get("FACPPG_STREAM_PRIO", "1") == "0":
                lo = hi = 0
            self.side = torch.cuda.Stream(device=dev, priority=max(lo, hi))          # the seed passes (largest number = lowest priority)
            # frame collection + the streaming postnet, one block ahead of them (FACPPG_STREAM_ONE=1: on the same stream, experiments)
            self.post = self.side if os.environ.get("FACPPG_STREAM_ONE") == "1" else torch.cuda.Stream(device=dev, priority=min(lo, hi))
            self.prio = (lo, hi)
            self.key = dev

        def grown(name, numel, dtype, zero=False):
            t = self.store.get(name)
            if t is None or t.numel() < numel:
                self.store[name] = t = (torch.zeros if zero else torch.empty)(numel, dtype=dtype, device=dev)
            return t[:numel]
        # (the layout queries validate the models' packed-weight handles -- ~1000 tensors, 0.2 - 0.4 ms of host time in front of the
        #  encoder: asked once per (step limit, layout), not per utterance)
        lk = (dev, steps, cap)
        if self.__dict__.get("layout_key") != lk:
            self.layout = self.waveglow.seed_layout(cap, dev) + (L.facppg_taco_postnet_stream_workspace_bytes(self.tacotron._handle(dev), cap),)
            self.layout_key = lk
        self.tqp, self.margin, seed_bytes, post_ws_bytes = self.layout
        # {value, frame + 1} words + the void flags + the work counters + mel_post in the vocoder's zero-margined layout: ONE allocation,
        # zeroed by one launch before every decode
        nw = steps * self.NF + 512
        self.zeroed = grown("zeroed", nw + (self.NF * self.tqp + 1) // 2, torch.int64, zero=True)
        self.words = self.zeroed[:nw]
        self.void = self.words[steps * self.NF:].view(torch.int32)[:512]                  # one per block
        self.counters = self.words[steps * self.NF:].view(torch.int32)[512:]              # one per bounded seed launch
        self.melp = self.zeroed[nw:].view(torch.float32)[:self.NF * self.tqp].view(self.NF, self.tqp)
        self.mel = grown("mel", self.NF * steps, torch.float32, zero=True).view(self.NF, steps)   # collected frames, channel-major
        if "void_host" not in self.store:
            self.store["void_host"] = torch.zeros(512, dtype=torch.int32).pin_memory()
        self.void_host = self.store["void_host"]
        self.seeds = grown("seeds", seed_bytes // 4, torch.float32)
        self.post_ws = grown("post_ws", post_ws_bytes, torch.uint8)

    def footprint_bytes(self):
        """Device memory the stream holds on to between utterances."""
        return sum(t.numel() * t.element_size() for t in getattr(self, "store", {}).values() if t.is_cuda)

    def plan(self, steps, Tin):
        """[(frames of mel needed, first seeded frame, end of seeded frames)] -- blocks that can be formed before the utterance ends
        if it runs to about min(steps, Tin) frames; whatever is not covered is left for finish()."""
        end = min(steps, max(Tin, 1))
        # 32-frame blocks keep every block's postnet chain clear of the previous block's pass (a block every 0.62 ms, chain + pass
        # 0.6 ms) at 2 GB of weight images per block; utterances of several seconds take 64-frame blocks (half the HBM traffic
        # next to the decoder) except for the last one before the expected end
        chunk = max(32, int(os.environ.get("FACPPG_STREAM_CHUNK", "32" if end <= 320 else "64")) // 32 * 32)
        last = max(32, int(os.environ.get("FACPPG_STREAM_LAST", "32")) // 32 * 32)
        s_end = (end - self.lag) // 32 * 32                       # seeded frames before the expected end
        cuts, s = [], 0
        widths = [int(v) // 32 * 32 for v in os.environ.get("FACPPG_STREAM_PLAN", "").split(",") if v.strip()]   # (experiments)
        while s < s_end and len(cuts) < 120:
            rest = s_end - s
            w = rest if rest <= last else min(chunk, rest - last)
            if widths:
                w = max(32, min(widths.pop(0), rest))
            cuts.append((s + w + self.lag, s, s + w))
            s += w
        s_lim = (min(steps, getattr(self, "cap", steps)) - self.lag) // 32 * 32   # the decoder may run on towards its step limit: a few more blocks (inside the layout)
        extra = 0
        while s < s_lim and extra < 2 and len(cuts) < 120:
            w = min(chunk, s_lim - s)
            cuts.append((s + w + self.lag, s, s + w))
            s += w
            extra += 1
        self.n_extra = extra
        return cuts

    # ---- called by Tacotron2.inference
    def begin(self, tacotron, handle, dev, steps, Tin):
        """Before the decoder launch, on its stream: zeroed frame words.  Returns them (None: not usable for this call)."""
        self.active = False
        # Short utterances do not gain: up to 128 frames the unstreamed vocoder runs 16-frame tiles, one per CU, and a layer launch
        # lasts as long as ONE tile either way (measured, tools/stream_T_sweep.sh: 64 frames 6.9 -> 8.1 ms, 100 frames 8.2 -> 9.0
        # streamed; 130 frames 10.0 -> 9.7, 200 frames 13.3 -> 11.5, 1000 frames 53.9 -> 48.5).  FACPPG_STREAM_MIN_FRAMES overrides.
        if not self.usable(tacotron, self.waveglow) or min(steps, Tin) < int(os.environ.get("FACPPG_STREAM_MIN_FRAMES", "128")):
            return None
        self.cap = min(steps, -(-(Tin + self.SLACK) // 32) * 32)
        try:
            self._buffers(dev, steps, self.cap)
        except torch.cuda.OutOfMemoryError:                 # (no room for the seeds next to whatever else lives here: run unstreamed)
            self.key, self.store = None, {}
            return None
        self.dev, self.steps, self.Tin, self.taco_handle = dev, steps, Tin, handle
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(self.side)      # (an utterance that was abandoned half way -- an exception between the decoder and the vocoder --
        cur.wait_stream(self.post)      #  may have left launches behind on the side streams: they read the buffers zeroed next)
        self.zeroed.zero_()
        self.ready = torch.cuda.Event()
        self.ready.record(torch.cuda.current_stream(dev))
        return self.words

    def cancel(self):
        self.active = False

    def enqueue(self, out_len):
        """The decoder has been launched and publishes its frames: enqueue every planned block on the side stream."""
        from facppg import lib as _lib
        L = _lib.load()
        dev, steps = self.dev, self.steps
        self.cuts = self.plan(steps, self.Tin)
        self.wg_handle = self.waveglow._handle(dev)
        f_prev = 0
        lpw = max(1, int(os.environ.get("FACPPG_STREAM_LPW", "1")))
        # The seed passes take the CUs the decoder leaves except `spare` of them, ONE workgroup per CU that holds the CU's whole LDS
        # (a bounded launch does, see facppg_wg_cond_seed): the dispatcher otherwise places the postnet's small workgroups -- and the
        # first launches behind the decoder on the main stream -- next to a pass's workgroups, where they crawl (3-5x, measured).
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        spare = int(os.environ.get("FACPPG_STREAM_SPARE_CUS", "8"))
        bound = max(16, n_cu - self.tacotron.last_decoder_launch()[1] - spare) if spare >= 0 else 0
        self.n_launch = 0
        self.last_final, self.finals, self.flow_events = None, [], {}

        def seed_pass(job):
            s_a, n, bt, void, lo, hi = job
            if self.profile:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
            self.waveglow.cond_seed(self.melp, self.cap, s_a, n, self.seeds, block_tiles=bt, layers_per_workgroup=lpw, skip=void,
                                    handle=self.wg_handle, flows=(lo, hi - lo), max_workgroups=bound,
                                    counter=self.counters[self.n_launch:self.n_launch + 1] if self.n_launch < 500 else None)
            self.n_launch += 1
            if self.profile:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(self.side)
                self.pass_events.append((n, hi - lo, e0, e1))
        self.seed_pass = seed_pass
        self.pass_events = []
        self.post.wait_event(self.ready)
        with torch.cuda.device(dev):
            for k, (f_new, s_a, s_b) in enumerate(self.cuts):
                void = self.void[k:k + 1]
                with torch.cuda.stream(self.post):
                    st = _lib.current_stream(dev)
                    _lib.check(L.facppg_taco_collect_frames(self.taco_handle, _lib.ptr(self.words), _lib.ptr(out_len), f_prev, f_new,
                                                            _lib.ptr(self.mel), steps, _lib.ptr(void),
                                                            _lib.ptr(self.void[k - 1:k]) if k else None, st))
                    _lib.check(L.facppg_taco_postnet_range(self.taco_handle, _lib.ptr(self.mel), steps, f_prev, f_new, 0,
                                                           self.melp.data_ptr() + 4 * self.margin, self.tqp, _lib.ptr(self.post_ws),
                                                           self.post_ws.numel(), self.cap, _lib.ptr(void), st))
                    # (the block's void flag travels to pinned host memory behind its collector, on this stream: when the decoder has
                    #  ended the flags of the blocks it covered have been on the host for milliseconds -- finish() reads them there
                    #  instead of spending a second device->host round trip between the decoder and the vocoder)
                    self.void_host[k:k + 1].copy_(void, non_blocking=True)
                    final = torch.cuda.Event()
                    final.record(self.post)
                    self.last_final = final
                    self.finals.append(final)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(final)                # mel_post is final up to s_b
                    seed_pass((s_a, s_b - s_a, min(4, (s_b - s_a) // 32), void, 0, self.waveglow.n_flows))
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.flow_events = {self.waveglow.n_flows - 1: ev}   # the vocoder's first launches (last flow) wait for the last pass
                f_prev = f_new
        self.active = True

    def finish(self, Tout, out_len):
        """The decoder has ended at Tout frames (known on the host): what the blocks could not cover, on the caller's stream.
        Returns mel_post [1, NF, Tout] (a view of the vocoder's mel buffer, valid until the next streamed utterance on these models),
        or None when the utterance outgrew the stream's layout: the caller then runs the one-shot postnet and the ordinary vocoder."""
        from facppg import lib as _lib
        L = _lib.load()
        dev, steps = self.dev, self.steps
        cur = torch.cuda.current_stream(dev)
        if Tout > self.cap:       # the decoder ran on past the frames the buffers were laid out for (PPG length + SLACK): unstreamed
            cur.wait_stream(self.post)
            cur.wait_stream(self.side)
            self.active = False
            return None
        # what is left of the postnet needs the postnet stream's blocks only; the seed passes' events gate the vocoder's flows (vocode)
        if self.last_final is not None:
            cur.wait_event(self.last_final)
        # Which blocks were really formed: a block the decoder stopped short of is void (f_new > Tout, known from the length), but
        # so is a block whose frames did not arrive within FACPPG_STREAM_WAIT_MS (k_collect_frames gave up: a profiler serialising
        # kernels, cooperative launches queued behind another process) -- its postnet columns and seeds were never written, and
        # every block behind it is void too.  The flags are read back here (the blocks up to the decoder's end finished
        # milliseconds ago; the ones past it return the moment they see the length) and the tail below starts at the first void one.
        n_cov = sum(1 for f_new, _, _ in self.cuts if f_new <= Tout)
        self.void_blocks = 0
        if n_cov:
            self.finals[n_cov - 1].synchronize()          # (block k's flag is copied to the host ahead of its event)
            first_void = next((k for k in range(n_cov) if int(self.void_host[k])), n_cov)
            self.void_blocks, n_cov = n_cov - first_void, first_void
        f_done = s_done = 0
        for f_new, s_a, s_b in self.cuts[:n_cov]:
            f_done, s_done = f_new, s_b
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            if Tout > f_done:
                _lib.check(L.facppg_taco_collect_frames(self.taco_handle, _lib.ptr(self.words), _lib.ptr(out_len), f_done, Tout,
                                                        _lib.ptr(self.mel), steps, None, None, st))
            _lib.check(L.facppg_taco_postnet_range(self.taco_handle, _lib.ptr(self.mel), steps, f_done, Tout, Tout,
                                                   self.melp.data_ptr() + 4 * self.margin, self.tqp, _lib.ptr(self.post_ws),
                                                   self.post_ws.numel(), self.cap, None, st))
        self.Tout, self.seeded = Tout, s_done
        return self.melp[:, self.margin:self.margin + Tout].unsqueeze(0)

    def pass_ms(self):
        """[(frames, flows, ms)] of the seed passes of the most recent utterance (profile = True); synchronises them."""
        out = []
        for n, nf, e0, e1 in self.pass_events:
            e1.synchronize()
            out.append((n, nf, e0.elapsed_time(e1)))
        return out

    # ---- called by the vocoder stage
    def vocode(self, sigma, z=None, seed=None):
        """WaveGlow.infer of the streamed utterance from the seeds (the frames behind the last block get theirs here)."""
        T, s_done = self.Tout, self.seeded
        s_all = -(-T // 32) * 32
        # The frames behind the last block: as unseeded 16-frame tiles inside the layer launches themselves (k_wn_layer_mixed) while
        # the launch still gives every CU at most one workgroup; otherwise (and with FACPPG_STREAM_TAIL=seed) one more seed pass
        # over them in front of the vocoder.
        P = self.waveglow.upsample.stride[0] // 8
        mixed_wgs = P * (s_done // 32 + -(-(T - s_done) // 16))
        seeded_wgs = P * (s_all // 32)
        n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        tail = os.environ.get("FACPPG_STREAM_TAIL", "auto")
        # ... i.e. while the mixed launch needs no more ROUNDS of one workgroup per CU than the all-seeded one would (measured: 230
        # frames, 288 against 256 workgroups: 17.5 against 13.4 ms per step; 300 / 350 / 450 frames, two rounds either way: the
        # mixed launch is 0.35 - 0.85 ms ahead of the extra pass)
        in_kernel = s_done > 0 and s_done < T and tail != "seed" and (tail == "mixed" or -(-mixed_wgs // n_cu) <= -(-seeded_wgs // n_cu))
        if s_all > s_done and not in_kernel:
            self.waveglow.cond_seed(self.melp, self.cap, s_done, s_all - s_done, self.seeds, block_tiles=1, layers_per_workgroup=2,
                                    handle=self.wg_handle)
            s_done = s_all
        self.active = False
        return self.waveglow.infer_seeded(self.melp, T, self.seeds, s_done, sigma=sigma, z=z, seed=seed, handle=self.wg_handle,
                                          T_layout=self.cap, flow_events=self.flow_events)


class StageTimer(object):
    """hipEvent timestamps on the launch stream between the stages of one synthesize() call (the kernels run on
    torch's current stream, so torch.cuda.Event brackets them).  ``mark(name)`` closes stage ``name``."""

    def __init__(self):
        self.marks = []
        self.mark("start")

    def mark(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.marks.append((name, ev))

    def stages_ms(self):
        """{stage: ms} in order, plus 'total'; synchronises the last event."""
        self.marks[-1][1].synchronize()
        out = {}
        for (_, a), (name, b) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        out["total"] = self.marks[0][1].elapsed_time(self.marks[-1][1])
        return out


def pad_ppgs(ppgs, device=None):
    """list of [Tin_i, D] arrays -> ([B, D, Tmax] float32 tensor, lengths list).

    With ``device`` the frames are uploaded as they are (time-major, one contiguous copy each) and
    transposed into the channel-major batch on the GPU; the host-side transpose of a 2 s utterance
    ([200, 5816] floats) costs more than the whole encoder."""
    lens = [int(p.shape[0]) for p in ppgs]
    D = int(ppgs[0].shape[1])
    # (one utterance, or equal lengths: every column is written below -- no zero fill in front of the upload)
    alloc = torch.empty if min(lens) == max(lens) else torch.zeros
    x = alloc(len(ppgs), D, max(lens), dtype=torch.float32, device=device)
    for b, p in enumerate(ppgs):
        t = torch.as_tensor(np.ascontiguousarray(p, dtype=np.float32))
        if device is not None:
            t = t.to(device, non_blocking=True)
        x[b, :, :lens[b]] = t.t()
    return x, lens


def _acoustic(ppgs, tacotron, seed, dropout_masks, utterance_seeds, step_limits, timer=None, while_decoding=None, consumer=None):
    """PPG upload + Tacotron2.inference on the current stream -> (mel_post [B, 80, Tout], [Tout_i]).  Blocks the host once,
    for the decoder's output lengths."""
    dev = next(tacotron.parameters()).device
    x, lens = pad_ppgs(ppgs, device=dev)
    if timer is not None:
        timer.mark("ppg_upload")
    _, mel_post, _, _ = tacotron.inference(x, lengths=lens if len(lens) > 1 else None, dropout_masks=dropout_masks,
                                           seed=seed, utterance_seeds=utterance_seeds, step_limits=step_limits,
                                           while_decoding=while_decoding, frame_consumer=consumer if len(lens) == 1 else None,
                                           **({"timer": timer} if timer is not None else {}))
    if consumer is not None and consumer.active:      # (the vocoder reads the stream's own mel buffer: no copy on the latency path)
        return mel_post, [int(v) for v in tacotron.last_output_lengths]
    return mel_post.contiguous(), [int(v) for v in tacotron.last_output_lengths]


def _vocode(mel_post, tout, waveglow, denoiser, sigma, strength, seed, z, utterance_seeds, timer=None, consumer=None):
    """WaveGlow.infer + Denoiser on the current stream -> audio [B, Tout_max * hop] on the device; no host waits beyond the
    small uploads (lengths, seeds)."""
    hop = waveglow.upsample.stride[0]
    multi = len(tout) > 1
    wg_seeds = None if utterance_seeds is None else [int(v) + 1 for v in utterance_seeds]
    if consumer is not None and consumer.active:
        if wg_seeds is not None:
            z = waveglow.draw_noise(wg_seeds, tout[0], mel_post.device)
        audio = consumer.vocode(sigma, z=z, seed=seed)
    else:
        audio = waveglow.infer(mel_post, sigma=sigma, z=z, lengths=tout if multi else None, seed=seed, utterance_seeds=wg_seeds)
    if timer is not None:
        timer.mark("waveglow")
    if denoiser is not None:
        audio = denoiser(audio, strength=strength, lengths=[t * hop for t in tout] if multi else None)[:, 0]
        if timer is not None:
            timer.mark("denoiser")
    return audio


def synthesize(ppgs, tacotron, waveglow, denoiser=None, sigma=0.6, strength=0.005, seed=None, dropout_masks=None, z=None,
               return_device=False, utterance_seeds=None, step_limits=None, timer=None):
    """Returns (list of float32 waveforms [N_i], list of mel lengths).  Models must be on the GPU.

    One utterance (the latency path, the metric's "batch = 1"): the postnet and the conditioning part of the vocoder's gate GEMMs
    run while the decoder is still producing frames (ConditioningStream; FACPPG_STREAM=0 switches it off) -- same samples.
    utterance_seeds: one integer per utterance -- its dropout and noise streams then depend on that seed alone,
    so the result for an utterance is the same whatever batch, batch size or GPU it is synthesised in.
    step_limits: per-utterance max_decoder_steps (e.g. its PPG length)."""
    if timer is not None:
        timer.__init__()
    hop = waveglow.upsample.stride[0]
    with torch.no_grad():
        dev = next(waveglow.parameters()).device
        consumer = None
        if len(ppgs) == 1 and ConditioningStream.usable(tacotron, waveglow):
            consumer = waveglow.__dict__.get("_facppg_cond_stream")
            if consumer is None or consumer.tacotron is not tacotron:
                consumer = waveglow.__dict__["_facppg_cond_stream"] = ConditioningStream(tacotron, waveglow)
        # The stream's buffers and per-utterance state belong to the model pair: ONE utterance at a time goes through them.  A second
        # thread serving another batch-1 request on the same models meanwhile takes the unstreamed path (same samples).
        if consumer is not None and not consumer.lock.acquire(blocking=False):
            consumer = None
        try:
            mel_post, tout = _acoustic(ppgs, tacotron, seed, dropout_masks, utterance_seeds, step_limits, timer,
                                       # host work under the decoder's milliseconds: the vocoder's weight check (the stream does its own)
                                       while_decoding=lambda: None if (consumer is not None and consumer.active) else waveglow.prepare(dev),
                                       consumer=consumer)
            audio = _vocode(mel_post, tout, waveglow, denoiser, sigma, strength, seed, z, utterance_seeds, timer, consumer)
        finally:
            if consumer is not None:
                consumer.lock.release()
    if return_device:
        return [audio[b, :tout[b] * hop] for b in range(len(tout))], tout
    host = audio.cpu().numpy()
    return [host[b, :tout[b] * hop].copy() for b in range(len(tout))], tout


_ACOUSTIC_STREAMS = {}


def _acoustic_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _ACOUSTIC_STREAMS:
        _ACOUSTIC_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return _ACOUSTIC_STREAMS[key]


def synthesize_stream(jobs, tacotron, waveglow, denoiser=None, sigma=0.6, strength=0.005, return_device=True, overlap=True,
                      acoustic_workgroups=32):
    """A sequence of batches, software-pipelined: generator of (waveforms, mel lengths), one per job, in order.

    jobs: iterable of dicts with ``ppgs`` and optionally ``seed``, ``utterance_seeds``, ``step_limits`` (as synthesize()).
    The acoustic model of batch i+1 (PPG upload, encoder, the latency-bound autoregressive decoder, postnet -- a few
    percent of the chip for ~25 ms) runs on its own HIP stream UNDER the vocoder of batch i (MFMA-bound, ~120 ms for 16
    utterances) instead of in front of it.  The host enqueues vocoder i first, then walks through acoustic i+1, whose one
    blocking read (the decoder's output lengths) it would otherwise spend idle.  Every utterance's samples are those of
    synthesize() on the same job: the stages, their inputs and their random streams are the same, only their placement in
    time differs (tests/test_gpu_e2e.py).  overlap=False runs the same jobs back to back on the caller's stream.

    acoustic_workgroups: while overlapped, the decoder is held to this many CUs (Tacotron2.decoder_workgroups, unless the model
    already carries a bound of its own).  A decoder workgroup owns its CU's LDS, so the vocoder loses every CU the decoder
    sits on; on the whole chip (240 workgroups for 16 utterances) the two stages merely take turns (146.7 -> 140.2 ms per
    batch), on 32 CUs the decoder takes 60 instead of 18 ms -- still hidden -- and the vocoder keeps 7/8 of the chip
    (128.3 ms; profiles/r03_experiments.txt).  The slice width that goes with the bound cuts the decoder's LSTM sums
    differently: samples then equal those of synthesize() with the same Tacotron2.decoder_workgroups bit for bit, and those
    of the unbounded decoder to rounding."""
    dev = next(tacotron.parameters()).device
    hop = waveglow.upsample.stride[0]
    main = torch.cuda.current_stream(dev)
    side = _acoustic_stream(dev) if overlap else main

    def acoustic(job):
        with torch.no_grad(), torch.cuda.stream(side):
            # (the inputs are host arrays: nothing on the caller's stream to wait for -- in particular not the previous vocoder)
            bound = tacotron.decoder_workgroups
            if overlap and not bound and acoustic_workgroups:
                tacotron.decoder_workgroups = int(acoustic_workgroups)
            try:
                mel_post, tout = _acoustic(job["ppgs"], tacotron, job.get("seed"), None, job.get("utterance_seeds"), job.get("step_limits"))
            finally:
                tacotron.decoder_workgroups = bound
            done = torch.cuda.Event()
            done.record(side)
        mel_post.record_stream(main)
        return job, mel_post, tout, done

    def finish(audio, tout):
        if return_device:
            return [audio[b, :tout[b] * hop] for b in range(len(tout))], tout
        host = audio.cpu().numpy()
        return [host[b, :tout[b] * hop].copy() for b in range(len(tout))], tout

    it = iter(jobs)
    first = next(it, None)
    if first is None:
        return
    ready = acoustic(first)
    previous_vocoder = None
    while ready is not None:
        job, mel_post, tout, done = ready
        main.wait_event(done)
        with torch.no_grad():
            audio = _vocode(mel_post, tout, waveglow, denoiser, sigma, strength, job.get("seed"), None, job.get("utterance_seeds"))
        vocoder_done = torch.cuda.Event()
        vocoder_done.record(main)
        nxt = next(it, None)
        if nxt is not None:
            # The acoustic model (~25-60 ms) is several times faster than the vocoder it hides under: left alone, the host would run
            # the acoustic models of ALL remaining jobs under the first vocoders and queue every vocoder call -- each with its output,
            # noise and workspace buffers -- far ahead of its execution.  One job of look-ahead is all the overlap needs: acoustic
            # i+1 is enqueued when vocoder i-1 has finished, i.e. as vocoder i starts.
            if overlap and previous_vocoder is not None:
                previous_vocoder.synchronize()
            ready = acoustic(nxt)       # enqueued behind nothing on its own stream: runs while the vocoder above does
        else:
            ready = None
        previous_vocoder = vocoder_done
        yield finish(audio, tout)
