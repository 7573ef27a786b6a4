"""Tacotron2-style PPG -> mel model -- drop-in for src/common/model.py.

The module tree (names, parameter shapes, state-dict keys: SURVEY.md Appendix B) equals the
reference's so ``load_state_dict(torch.load(ckpt)['state_dict'])`` works unchanged
(generate_synthesis.py:82).  The modules hold parameters only; ``Tacotron2.inference`` packs
them once into a ``facppg_taco`` handle and runs the encoder GEMMs, the BiLSTM kernel, the
persistent decoder kernel and the postnet GEMMs of csrc/facppg_taco.hip through the C ABI.
There is no CPU path.

Extensions over the reference signature (defaults reproduce it):
  ``inference(inputs, lengths=None, dropout_masks=None, seed=None)``
    lengths        valid PPG frames per utterance for a padded batch; every utterance is decoded
                   exactly as its own batch-1 run (the reference only supports batch 1,
                   model.py:524-528)
    dropout_masks  (enc [2, B, Tin, E], dec [steps, 2, B, prenet_dim]) keep-masks in {0,1} for the
                   prenets' always-on dropout (model.py:132-135); None -> drawn on the device
    utterance_seeds  B integers: the dropout draws of utterance b depend on utterance_seeds[b] alone, so the
                   batch reproduces B independent batch-1 calls with the same seeds (facppg_taco_draw_dropout)
    step_limits    B integers: utterance b stops after min(step_limits[b], max_decoder_steps) frames at the
                   latest (per-utterance max_decoder_steps of a padded batch)
"""
import torch
from torch import nn

from common.layers import ConvNorm, LinearNorm
from facppg import lib as _lib


class LocationLayer(nn.Module):
    """model.py:44-60"""

    def __init__(self, attention_n_filters, attention_kernel_size, attention_dim):
        super(LocationLayer, self).__init__()
        self.location_conv = ConvNorm(2, attention_n_filters, kernel_size=attention_kernel_size,
                                      padding=(attention_kernel_size - 1) // 2, bias=False, stride=1, dilation=1)
        self.location_dense = LinearNorm(attention_n_filters, attention_dim, bias=False, w_init_gain='tanh')


class Attention(nn.Module):
    """model.py:63-121 (location-sensitive attention); evaluated inside k_decoder."""

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super(Attention, self).__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(attention_location_n_filters, attention_location_kernel_size, attention_dim)
        self.score_mask_value = -float("inf")
        self.not_so_small_mask = -1000


class Prenet(nn.Module):
    """model.py:124-135: bias-free Linear + ReLU + dropout(p=0.5, ALWAYS on) per layer."""

    def __init__(self, in_dim, sizes):
        super(Prenet, self).__init__()
        in_sizes = [in_dim] + sizes[:-1]
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=False) for i, o in zip(in_sizes, sizes)])


class Postnet(nn.Module):
    """model.py:138-184: five 1-D convolutions (k=5) with BatchNorm, tanh on all but the last."""

    def __init__(self, hparams):
        super(Postnet, self).__init__()
        nf, pe, pk = hparams.n_acoustic_feat_dims, hparams.postnet_embedding_dim, hparams.postnet_kernel_size
        dims = [nf] + [pe] * (hparams.postnet_n_convolutions - 1) + [nf]
        self.convolutions = nn.ModuleList()
        for j in range(hparams.postnet_n_convolutions):
            last = j == hparams.postnet_n_convolutions - 1
            self.convolutions.append(nn.Sequential(
                ConvNorm(dims[j], dims[j + 1], kernel_size=pk, stride=1, padding=(pk - 1) // 2, dilation=1,
                         w_init_gain='linear' if last else 'tanh'),
                nn.BatchNorm1d(dims[j + 1])))


class Encoder(nn.Module):
    """model.py:187-249: prenet, conv bank (conv+BN+ReLU), bidirectional LSTM."""

    def __init__(self, hparams):
        super(Encoder, self).__init__()
        E = hparams.encoder_embedding_dim
        self.prenet = Prenet(hparams.n_symbols, [hparams.symbols_embedding_dim, hparams.symbols_embedding_dim])
        ks = hparams.encoder_kernel_size
        self.convolutions = nn.ModuleList([
            nn.Sequential(ConvNorm(E, E, kernel_size=ks, stride=1, padding=(ks - 1) // 2, dilation=1, w_init_gain='relu'),
                          nn.BatchNorm1d(E))
            for _ in range(hparams.encoder_n_convolutions)])
        self.lstm = nn.LSTM(E, E // 2, 1, batch_first=True, bidirectional=True)


class Decoder(nn.Module):
    """model.py:252-535"""

    def __init__(self, hparams):
        super(Decoder, self).__init__()
        self.n_acoustic_feat_dims = hparams.n_acoustic_feat_dims
        self.encoder_embedding_dim = hparams.encoder_embedding_dim
        self.attention_rnn_dim = hparams.attention_rnn_dim
        self.decoder_rnn_dim = hparams.decoder_rnn_dim
        self.prenet_dim = hparams.prenet_dim
        self.max_decoder_steps = hparams.max_decoder_steps
        self.gate_threshold = hparams.gate_threshold
        self.p_attention_dropout = hparams.p_attention_dropout
        self.p_decoder_dropout = hparams.p_decoder_dropout
        self.attention_window_size = hparams.attention_window_size
        self.prenet = Prenet(hparams.n_acoustic_feat_dims, [hparams.prenet_dim, hparams.prenet_dim])
        self.attention_rnn = nn.LSTMCell(hparams.prenet_dim + hparams.encoder_embedding_dim, hparams.attention_rnn_dim)
        self.attention_layer = Attention(hparams.attention_rnn_dim, hparams.encoder_embedding_dim, hparams.attention_dim,
                                         hparams.attention_location_n_filters, hparams.attention_location_kernel_size)
        self.decoder_rnn = nn.LSTMCell(hparams.attention_rnn_dim + hparams.encoder_embedding_dim, hparams.decoder_rnn_dim, 1)
        self.linear_projection = LinearNorm(hparams.decoder_rnn_dim + hparams.encoder_embedding_dim,
                                            hparams.n_acoustic_feat_dims)
        self.gate_layer = LinearNorm(hparams.decoder_rnn_dim + hparams.encoder_embedding_dim, 1, bias=True,
                                     w_init_gain='sigmoid')


class Tacotron2(nn.Module):
    """model.py:538-610"""

    def __init__(self, hparams):
        super(Tacotron2, self).__init__()
        self.mask_padding = hparams.mask_padding
        self.fp16_run = hparams.fp16_run
        self.n_acoustic_feat_dims = hparams.n_acoustic_feat_dims
        self.encoder = Encoder(hparams)
        self.decoder = Decoder(hparams)
        self.postnet = Postnet(hparams)
        self._hp = {k: getattr(hparams, k) for k in (
            "n_symbols", "symbols_embedding_dim", "encoder_kernel_size", "encoder_n_convolutions", "encoder_embedding_dim",
            "n_acoustic_feat_dims", "prenet_dim", "attention_rnn_dim", "decoder_rnn_dim", "attention_dim",
            "attention_location_n_filters", "attention_location_kernel_size", "attention_window_size",
            "postnet_embedding_dim", "postnet_kernel_size", "postnet_n_convolutions", "gate_threshold")}

    # ---------------------------------------------------------------- HIP handle
    def _config(self):
        c = _lib.TacoConfig()
        for k, v in self._hp.items():
            if k == "attention_window_size":
                v = -1 if v is None else int(v)
            setattr(c, k, v)
        c.gate_threshold = float(self.decoder.gate_threshold)
        c.bn_eps = float(self.postnet.convolutions[0][1].eps)
        return c

    def _flat_weights(self):
        """Plain weight blob in the order include/facppg.h documents."""
        sd = self.state_dict()
        keys = ["encoder.prenet.layers.0.linear_layer.weight", "encoder.prenet.layers.1.linear_layer.weight"]
        for j in range(len(self.encoder.convolutions)):
            p = "encoder.convolutions.%d." % j
            keys += [p + "0.conv.weight", p + "0.conv.bias", p + "1.weight", p + "1.bias", p + "1.running_mean", p + "1.running_var"]
        for sfx in ("", "_reverse"):
            keys += ["encoder.lstm.%s_l0%s" % (n, sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        keys += ["decoder.prenet.layers.0.linear_layer.weight", "decoder.prenet.layers.1.linear_layer.weight"]
        keys += ["decoder.attention_rnn." + n for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        a = "decoder.attention_layer."
        keys += [a + "query_layer.linear_layer.weight", a + "memory_layer.linear_layer.weight", a + "v.linear_layer.weight",
                 a + "location_layer.location_conv.conv.weight", a + "location_layer.location_dense.linear_layer.weight"]
        keys += ["decoder.decoder_rnn." + n for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        keys += ["decoder.linear_projection.linear_layer.weight", "decoder.linear_projection.linear_layer.bias",
                 "decoder.gate_layer.linear_layer.weight", "decoder.gate_layer.linear_layer.bias"]
        for j in range(len(self.postnet.convolutions)):
            p = "postnet.convolutions.%d." % j
            keys += [p + "0.conv.weight", p + "0.conv.bias", p + "1.weight", p + "1.bias", p + "1.running_mean", p + "1.running_var"]
        return torch.cat([sd[k].detach().float().reshape(-1) for k in keys])

    def _release(self):
        h = self.__dict__.pop("_facppg_handle", None)
        if h is not None:
            _lib.load().facppg_taco_destroy(h[0])

    def invalidate_packed_weights(self):
        self._release()

    def train(self, mode=True):
        self._release()
        return super(Tacotron2, self).train(mode)

    # CUs the decoder may occupy (facppg_taco_set_decoder_workgroups); 0 = the whole device.  facppg.pipeline.synthesize_stream
    # sets it while the decoder runs under the previous batch's vocoder.
    decoder_workgroups = 0

    def _handle(self, dev):
        h = self.__dict__.get("_facppg_handle")
        if h is not None and h[1] == dev and h[2].unchanged():
            return h[0]
        self._release()
        L = _lib.load()
        cfg = self._config()
        blob = self._flat_weights().to(dev).contiguous()
        if blob.numel() != L.facppg_taco_weight_count(cfg):
            raise _lib.FacppgError("weight blob has %d values, library expects %d: %s" % (
                blob.numel(), L.facppg_taco_weight_count(cfg), L.facppg_last_error().decode()))
        out = _lib.ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(L.facppg_taco_create(cfg, _lib.ptr(blob), blob.numel(), dev.index, _lib.current_stream(dev),
                                            _lib.ctypes.byref(out)))
        self.__dict__["_facppg_handle"] = (out, dev, _lib.WeightIdentity(self))
        return out

    def last_decoder_launch(self):
        """(mode, workgroups) of the most recent decoder launch (facppg_taco_last_decoder_launch): mode is 'single', 'coop' or
        'split'."""
        h = self.__dict__.get("_facppg_handle")
        if h is None:
            raise _lib.FacppgError("last_decoder_launch: no inference has run on this model yet")
        c = _lib.ctypes
        mode, wgs = c.c_int(), c.c_int()
        _lib.check(_lib.load().facppg_taco_last_decoder_launch(h[0], c.byref(mode), c.byref(wgs)))
        return ("single", "coop", "split")[mode.value], wgs.value

    def _apply(self, fn, *a, **k):
        self._release()
        return super(Tacotron2, self)._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._release()
        return super(Tacotron2, self).load_state_dict(*a, **k)

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_facppg_handle", None)
        return d

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---------------------------------------------------------------- reference surface
    def parse_input(self, inputs):
        if self.fp16_run:
            raise _lib.FacppgError("fp16_run is not built (the reference's README.md:53 says FP16 does not work either)")
        return inputs

    def parse_output(self, outputs, output_lengths=None):
        return outputs

    def forward(self, inputs):
        raise NotImplementedError("teacher-forced training of the PPG->mel model is out of scope (SURVEY.md section 2)")

    def draw_dropout_masks(self, utterance_seeds, Tin, device=None, steps=None):
        """Per-utterance dropout keep-masks (facppg_taco_draw_dropout) in the DEVICE layouts the kernels read:
        (enc uint8 [2, B, symbols_embedding_dim, Tin], dec uint8 [max_decoder_steps, 2, B, prenet_dim]).  Mask bits of
        utterance b depend on utterance_seeds[b] only (not on B or Tin).  For ``inference(dropout_masks=...)``,
        which takes the reference-shaped [2, B, Tin, E] encoder masks, pass ``enc.permute(0, 1, 3, 2)``."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        L = _lib.load()
        h = self._handle(dev)
        B, steps = len(utterance_seeds), int(self.decoder.max_decoder_steps if steps is None else steps)
        sd = torch.tensor([int(v) & 0x7FFFFFFFFFFFFFFF for v in utterance_seeds], dtype=torch.int64, device=dev)
        enc_m = torch.empty(2, B, self._hp["symbols_embedding_dim"], Tin, dtype=torch.uint8, device=dev)
        dec_m = torch.empty(steps, 2, B, self._hp["prenet_dim"], dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.facppg_taco_draw_dropout(h, _lib.ptr(sd), B, Tin, steps, _lib.ptr(enc_m), _lib.ptr(dec_m),
                                                  _lib.current_stream(dev)))
        return enc_m, dec_m

    def inference(self, inputs, lengths=None, dropout_masks=None, seed=None, utterance_seeds=None, step_limits=None, timer=None,
                  while_decoding=None, frame_consumer=None):
        """inputs [B, n_symbols, Tin] (GPU fp32) -> [mel, mel_post, gate, alignments]
        = [B,80,Tout], [B,80,Tout], [B,Tout,1], [B,Tout,Tin]  (model.py:597-610).  For B > 1 the
        outputs are zero beyond each utterance's own Tout, kept in ``self.last_output_lengths``.
        while_decoding: optional callable run on the host after the decoder has been enqueued and before its output lengths
        are read back (facppg.pipeline checks the vocoder's packed weights there: ~0.4 ms that would otherwise sit between the
        acoustic model and the vocoder with the GPU idle).
        frame_consumer: optional (B = 1; facppg.pipeline.ConditioningStream): the split decoder publishes every mel frame the
        moment it exists and the consumer runs the postnet -- and whatever else it wants of the frames -- on a second stream
        WHILE the decoder is still running; mel_post is then the consumer's (the same values bit for bit)."""
        inputs = self.parse_input(inputs)
        _lib.require_cuda(inputs, "Tacotron2.inference: inputs")
        L = _lib.load()
        dev = inputs.device
        x = inputs.float().contiguous()
        B, D, Tin = x.shape
        hp = self._hp
        if D != hp["n_symbols"]:
            raise _lib.FacppgError("inputs have %d symbols, model was built for %d" % (D, hp["n_symbols"]))
        h = self._handle(dev)
        lt = None
        if lengths is not None:
            lt = torch.as_tensor(lengths).to(device=dev, dtype=torch.int32).contiguous()
            if lt.numel() != B or int(lt.max()) > Tin or int(lt.min()) < 1:
                raise _lib.FacppgError("lengths must be B values in [1, Tin]")
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        seed &= 0xFFFFFFFFFFFFFFFF
        enc_m = dec_m = None
        steps = int(self.decoder.max_decoder_steps)
        sl = None
        if step_limits is not None:
            sl = torch.as_tensor(step_limits).to(dtype=torch.int32)
            if sl.numel() != B or int(sl.min()) < 1:
                raise _lib.FacppgError("step_limits must be B values >= 1")
            steps = min(steps, int(sl.max()))          # no utterance can run longer: size the outputs for that
            sl = sl.to(dev).contiguous()
        E, P, NF, AD = hp["encoder_embedding_dim"], hp["prenet_dim"], hp["n_acoustic_feat_dims"], hp["attention_dim"]
        if dropout_masks is not None:
            em, dm = dropout_masks
            enc_m = torch.as_tensor(em).to(dev).to(torch.uint8).reshape(2, B, Tin, E).permute(0, 1, 3, 2).contiguous()
            dec_m = torch.as_tensor(dm).to(dev).to(torch.uint8).contiguous()
            if dec_m.numel() != steps * 2 * B * P:
                raise _lib.FacppgError("decoder masks must be [max_decoder_steps, 2, B, prenet_dim]")
        st = _lib.current_stream(dev)
        if utterance_seeds is not None:
            if dropout_masks is not None or len(utterance_seeds) != B:
                raise _lib.FacppgError("utterance_seeds: B integers, and not together with dropout_masks")
            enc_m, dec_m = self.draw_dropout_masks(utterance_seeds, Tin, dev, steps)
        ws = torch.empty(max(L.facppg_taco_workspace_bytes(h, B, Tin), L.facppg_taco_decode_workspace_bytes(h, B, steps)),
                         dtype=torch.uint8, device=dev)
        # the six zero-initialised outputs as views of ONE zeroed allocation: one fill launch instead of six (each tiny launch
        # costs the stream ~8 us of kernel boundary, and they sit in front of the encoder on the latency path)
        sizes = (B * Tin * E, B * Tin * AD, B * NF * steps, B * steps, B * steps * Tin, B)
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 63) // 64 * 64                      # 256-byte aligned pieces
        arena = torch.zeros(tot, dtype=torch.float32, device=dev)
        memory = arena[offs[0]:offs[0] + sizes[0]].view(B, Tin, E)
        pm = arena[offs[1]:offs[1] + sizes[1]].view(B, Tin, AD)
        mel = arena[offs[2]:offs[2] + sizes[2]].view(B, NF, steps)
        gate = arena[offs[3]:offs[3] + sizes[3]].view(B, steps)
        align = arena[offs[4]:offs[4] + sizes[4]].view(B, steps, Tin)
        out_len = arena[offs[5]:offs[5] + sizes[5]].view(torch.int32)
        _lib.check(L.facppg_taco_set_decoder_workgroups(h, int(self.decoder_workgroups)))
        streaming = False
        if frame_consumer is not None and B == 1:
            words = frame_consumer.begin(self, h, dev, steps, Tin)          # (zeroed on this stream, ahead of the decoder launch)
            _lib.check(L.facppg_taco_set_frame_stream(h, _lib.ptr(words), steps if words is not None else 0))
        with torch.cuda.device(dev):
            _lib.check(L.facppg_taco_encode(h, _lib.ptr(x), _lib.ptr(lt), _lib.ptr(enc_m), seed, B, Tin, _lib.ptr(memory),
                                            _lib.ptr(pm), _lib.ptr(ws), ws.numel(), st))
            if timer is not None:
                timer.mark("encoder")
            try:
                _lib.check(L.facppg_taco_decode(h, _lib.ptr(memory), _lib.ptr(pm), _lib.ptr(lt), _lib.ptr(sl), _lib.ptr(dec_m), seed, B, Tin,
                                                steps, _lib.ptr(mel), _lib.ptr(gate), _lib.ptr(align), _lib.ptr(out_len),
                                                _lib.ptr(ws), ws.numel(), st))
            finally:
                if frame_consumer is not None and B == 1:     # (whatever happened: no later decode publishes into this call's buffer)
                    L.facppg_taco_set_frame_stream(h, None, 0)
            if frame_consumer is not None and B == 1:
                flag = _lib.ctypes.c_int()
                _lib.check(L.facppg_taco_last_decode_streamed(h, _lib.ctypes.byref(flag)))
                streaming = bool(flag.value)
                if streaming:
                    frame_consumer.enqueue(out_len)            # its launches, gated on the frames, on its own stream
                else:
                    frame_consumer.cancel()
            if while_decoding is not None:                     # host work that needs no result of the decoder: the GPU is busy for
                while_decoding()                               # milliseconds, the host would only sit in the read below
            out_len_host = out_len.cpu()                       # the path's single device->host sync
            if timer is not None:
                timer.mark("decoder")
            Tout = int(out_len_host.max())
            if Tout == int(self.decoder.max_decoder_steps):
                print("Warning! Reached max decoder steps")     # model.py:527
            mel_post = frame_consumer.finish(Tout, out_len) if streaming else None   # [1, NF, >= Tout] view: what is left of the postnet
            if mel_post is None:
                mel_post = torch.zeros_like(mel)
                ws2 = torch.empty(L.facppg_taco_postnet_workspace_bytes(h, B, Tout), dtype=torch.uint8, device=dev)
                _lib.check(L.facppg_taco_postnet(h, _lib.ptr(mel), _lib.ptr(out_len), B, Tout, steps, _lib.ptr(mel_post),
                                                 _lib.ptr(ws2), ws2.numel(), st))
            if timer is not None:
                timer.mark("postnet")
        self.last_output_lengths = out_len_host.to(torch.long)
        self.last_memory = memory
        return self.parse_output([mel[:, :, :Tout], mel_post[:, :, :Tout], gate[:, :Tout].unsqueeze(-1), align[:, :Tout]])
