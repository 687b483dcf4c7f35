"""T5Transformer — MI355X implementation of ref: music2midi/transformer.py:10-45.

Same constructor (``config_path``), same attributes (``transformer``,
``tokenizer``, ``spectrogram``, ``conditioning``, ``t5config``) and the same two
entry points: ``forward(ModelInputs) -> output with .loss/.logits`` and
``generate(ModelInputs, **kwargs) -> LongTensor [B, L]``.  The parameters live in
a module tree whose ``state_dict()`` keys equal HuggingFace T5's, so a reference
checkpoint loads unchanged; the arithmetic runs in the HIP library (encoder,
cross-K/V projection, graph-replayed greedy decode) — there is no torch compute
path and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import native
from .config import T5Geometry, load_config
from .input import Conditioning, LogMelSpectrogram, ModelInputs
from .tokenizer import MidiTokenizer

_PRECISIONS = {"fp32": native.PREC_FP32, "bf16": native.PREC_BF16}


# --------------------------------------------------------------------------
# Parameter containers with HuggingFace T5 state-dict names (no compute).
# --------------------------------------------------------------------------
class _Norm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))


class _Attention(nn.Module):
    def __init__(self, g: T5Geometry, has_bias: bool):
        super().__init__()
        self.q = nn.Linear(g.d_model, g.inner_dim, bias=False)
        self.k = nn.Linear(g.d_model, g.inner_dim, bias=False)
        self.v = nn.Linear(g.d_model, g.inner_dim, bias=False)
        self.o = nn.Linear(g.inner_dim, g.d_model, bias=False)
        if has_bias:
            self.relative_attention_bias = nn.Embedding(g.num_buckets, g.num_heads)


class _GatedFF(nn.Module):
    def __init__(self, g: T5Geometry):
        super().__init__()
        self.wi_0 = nn.Linear(g.d_model, g.d_ff, bias=False)
        self.wi_1 = nn.Linear(g.d_model, g.d_ff, bias=False)
        self.wo = nn.Linear(g.d_ff, g.d_model, bias=False)


class _SubLayer(nn.Module):
    def __init__(self, name: str, inner: nn.Module, d: int):
        super().__init__()
        setattr(self, name, inner)
        self.layer_norm = _Norm(d)


class _Block(nn.Module):
    def __init__(self, g: T5Geometry, is_decoder: bool, first: bool):
        super().__init__()
        layers = [_SubLayer("SelfAttention", _Attention(g, first), g.d_model)]
        if is_decoder:
            layers.append(_SubLayer("EncDecAttention", _Attention(g, False), g.d_model))
        layers.append(_SubLayer("DenseReluDense", _GatedFF(g), g.d_model))
        self.layer = nn.ModuleList(layers)


class _Stack(nn.Module):
    def __init__(self, g: T5Geometry, embed: nn.Embedding, is_decoder: bool):
        super().__init__()
        self.embed_tokens = embed   # same module as `shared`: state_dict carries the HF alias keys
        n = g.num_decoder_layers if is_decoder else g.num_layers
        self.block = nn.ModuleList([_Block(g, is_decoder, i == 0) for i in range(n)])
        self.final_layer_norm = _Norm(g.d_model)


class T5Parameters(nn.Module):
    """Holds the T5 weights under HF names; initialised like HF's ``_init_weights``
    (hf: models/t5/modeling_t5.py:563-616) with an UNTIED ``lm_head`` — the
    transformers-4.34 meaning of ``tie_word_embeddings: false`` (ref: config.yaml:23)."""

    def __init__(self, g: T5Geometry):
        super().__init__()
        self.geometry = g
        self.shared = nn.Embedding(g.vocab_size, g.d_model)
        self.encoder = _Stack(g, self.shared, False)
        self.decoder = _Stack(g, self.shared, True)
        self.lm_head = nn.Linear(g.d_model, g.vocab_size, bias=False)
        self.config = SimpleNamespace(**g.as_dict(), is_encoder_decoder=True, tie_word_embeddings=False)
        self._init_weights()

    @torch.no_grad()
    def _init_weights(self):
        g = self.geometry
        d, dk, H, dff = g.d_model, g.d_kv, g.num_heads, g.d_ff
        self.shared.weight.normal_(0.0, 1.0)
        self.lm_head.weight.normal_(0.0, 1.0)
        for m in self.modules():
            if isinstance(m, _Attention):
                m.q.weight.normal_(0.0, (d * dk) ** -0.5)
                m.k.weight.normal_(0.0, d ** -0.5)
                m.v.weight.normal_(0.0, d ** -0.5)
                m.o.weight.normal_(0.0, (H * dk) ** -0.5)
                if hasattr(m, "relative_attention_bias"):
                    m.relative_attention_bias.weight.normal_(0.0, d ** -0.5)
            elif isinstance(m, _GatedFF):
                m.wi_0.weight.normal_(0.0, d ** -0.5)
                m.wi_1.weight.normal_(0.0, d ** -0.5)
                m.wo.weight.normal_(0.0, dff ** -0.5)

    @property
    def device(self) -> torch.device:
        return self.shared.weight.device

    def forward(self, *a, **k):
        raise RuntimeError("T5Parameters only stores weights; use T5Transformer.forward/generate")


class Seq2SeqOutput(dict):
    """Minimal stand-in for HF's Seq2SeqLMOutput: attribute and key access to loss/logits."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


# --------------------------------------------------------------------------
class T5Transformer(nn.Module):
    def __init__(self, config_path, precision: Optional[str] = None):
        super().__init__()
        self.config = load_config(config_path)
        self.geometry = T5Geometry(self.config.model.t5)
        self.t5config = SimpleNamespace(**self.geometry.as_dict())

        self.transformer = T5Parameters(self.geometry)
        self.tokenizer = MidiTokenizer(self.config)
        self.spectrogram = LogMelSpectrogram(
            sample_rate=self.config.model.sample_rate,
            n_mels=self.config.model.t5.d_model,
            **self.config.spectrogram,
        )
        self.conditioning = Conditioning(
            self.config.model.t5.d_model,
            [len(v) for v in self.config.conditioning.values()],
        )
        # The reference runs inference in fp32 (SURVEY.md §5 "precision flags"); fp32 is the
        # default so token ids match it.  "bf16" is the throughput mode (BASELINE config 3).
        self.precision = precision or os.environ.get("M2M_PRECISION", "fp32")
        if self.precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {self.precision!r}")
        self._lock = threading.RLock()   # a session is not re-entrant (threaded Flask dev server)
        self._model_handle = None
        self._model_key = None
        self._session = None
        self._session_key = None
        self._workspace = None
        self._keepalive = None

    # -- native objects ------------------------------------------------------
    def set_precision(self, precision: str) -> "T5Transformer":
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
        if precision != self.precision:
            self.precision = precision
            self._drop_native()
        return self

    def _drop_native(self):
        lib = native.load()
        if self._session is not None:
            lib.m2m_session_destroy(self._session)
            self._session = None
            self._session_key = None
            self._workspace = None
        if self._model_handle is not None:
            lib.m2m_model_destroy(self._model_handle)
            self._model_handle = None
            self._model_key = None

    def __del__(self):
        try:
            self._drop_native()
        except Exception:
            pass

    def _param_key(self):
        ps = list(self.transformer.parameters())
        # _weights_epoch: bumped by the native trainer, whose in-place kernel updates torch's version counters cannot see
        return (self.precision, str(ps[0].device), getattr(self, "_weights_epoch", 0)) + tuple((p.data_ptr(), p._version) for p in ps)

    def _get_model(self):
        """Repack the current parameters into the library (once per weight version)."""
        native.require_gpu()
        dev = self.transformer.device
        if dev.type != "cuda":
            raise native.NativeError("T5Transformer weights are on the CPU: call .cuda() first "
                                     "(the MI355X path has no CPU fallback)")
        key = self._param_key()
        if self._model_handle is not None and key == self._model_key:
            return self._model_handle
        self._drop_native()
        lib = native.load()
        g, t = self.geometry, self.transformer
        keep = []

        def p(param):
            x = param.detach().to(dtype=torch.float32).contiguous()
            keep.append(x)
            return x.data_ptr()

        def attn(a):
            return p(a.q.weight), p(a.k.weight), p(a.v.weight), p(a.o.weight)

        enc = (native.EncLayerWeights * g.num_layers)()
        for i, blk in enumerate(t.encoder.block):
            sa, ff = blk.layer[0], blk.layer[1]
            q, k, v, o = attn(sa.SelfAttention)
            enc[i] = native.EncLayerWeights(p(sa.layer_norm.weight), q, k, v, o, p(ff.layer_norm.weight),
                                            p(ff.DenseReluDense.wi_0.weight), p(ff.DenseReluDense.wi_1.weight),
                                            p(ff.DenseReluDense.wo.weight))
        dec = (native.DecLayerWeights * g.num_decoder_layers)()
        for i, blk in enumerate(t.decoder.block):
            sa, ca, ff = blk.layer[0], blk.layer[1], blk.layer[2]
            q, k, v, o = attn(sa.SelfAttention)
            cq, ck, cv, co = attn(ca.EncDecAttention)
            dec[i] = native.DecLayerWeights(p(sa.layer_norm.weight), q, k, v, o, p(ca.layer_norm.weight), cq, ck, cv, co,
                                            p(ff.layer_norm.weight), p(ff.DenseReluDense.wi_0.weight),
                                            p(ff.DenseReluDense.wi_1.weight), p(ff.DenseReluDense.wo.weight))
        w = native.T5Weights(
            p(t.shared.weight), p(t.lm_head.weight),
            p(t.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight),
            p(t.decoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight),
            p(t.encoder.final_layer_norm.weight), p(t.decoder.final_layer_norm.weight), enc, dec)
        geom = native.T5GeometryC(g.d_model, g.d_ff, g.num_layers, g.num_decoder_layers, g.num_heads, g.d_kv,
                                  g.vocab_size, g.num_buckets, g.max_distance, g.pad_token_id, g.eos_token_id,
                                  g.decoder_start_token_id, g.eps)
        h = C.c_void_p()
        with torch.cuda.device(dev):
            native.check(lib.m2m_model_create(C.byref(geom), C.byref(w), _PRECISIONS[self.precision],
                                              native.stream_handle(dev), C.byref(h)), "m2m_model_create")
        del keep
        self._model_handle, self._model_key = h, key
        return h

    def device_weights_checksum(self) -> int:
        """64-bit checksum of the REPACKED device weights the kernels read (``m2m_model_checksum``).  After the multi-GPU weight
        broadcast every rank's value must be the same (``distributed.verify_replicas``)."""
        h = self._get_model()
        out = C.c_uint64(0)
        dev = self.transformer.device
        with torch.cuda.device(dev):
            native.check(native.load().m2m_model_checksum(h, C.byref(out), native.stream_handle(dev)), "m2m_model_checksum")
        return int(out.value)

    def _get_session(self, B: int, S: int, L: int):
        model = self._get_model()
        lib = native.load()
        if self._session is not None:
            mb, ms, ml = self._session_key
            if B <= mb and S <= ms and L <= ml:
                return self._session
            B, S, L = max(B, mb), max(S, ms), max(L, ml)
            lib.m2m_session_destroy(self._session)
            self._session, self._workspace = None, None
        dev = self.transformer.device
        nbytes = lib.m2m_session_workspace_bytes(model, B, S, L)
        if nbytes < 0:
            native.check(int(nbytes), "m2m_session_workspace_bytes")
        self._workspace = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=dev)
        base = (self._workspace.data_ptr() + 255) // 256 * 256
        h = C.c_void_p()
        with torch.cuda.device(dev):
            native.check(lib.m2m_session_create(model, B, S, L, base, int(nbytes), C.byref(h)), "m2m_session_create")
        self._session, self._session_key = h, (B, S, L)
        return h

    # -- pipeline pieces -----------------------------------------------------
    def encoder_inputs(self, inputs: ModelInputs) -> torch.Tensor:
        """waveform + cond_index -> [B, n_cond + frames, d_model], written in place by the two
        frontend kernels (ref transformer.py:42-43 = spectrogram then conditioning concat)."""
        wav = inputs.input_waveform
        dev = self.transformer.device
        wav = wav.to(dev)
        n_cond = len(self.conditioning.embeds)
        F = self.spectrogram.num_frames(wav.shape[1])
        buf = torch.empty((wav.shape[0], n_cond + F, self.geometry.d_model), device=dev, dtype=torch.float32)
        self.spectrogram.forward_into(wav, buf, n_cond)
        self.conditioning.write_rows(inputs.cond_index, buf)
        return buf

    def _encode(self, x: torch.Tensor, max_dec: int, want_states: bool = False):
        B, S, _ = x.shape
        sess = self._get_session(B, S, max_dec)
        enc_out = torch.empty_like(x) if want_states else None
        with torch.cuda.device(x.device):
            native.check(native.load().m2m_encode(sess, x.data_ptr(), B, S,
                                                  enc_out.data_ptr() if want_states else None,
                                                  native.stream_handle(x.device)), "m2m_encode")
        return sess, enc_out

    @torch.no_grad()
    def encode(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        """Encoder stack only: [B, S, d] -> final encoder states [B, S, d] fp32 (parity hook)."""
        with self._lock:
            x = inputs_embeds.to(self.transformer.device, torch.float32).contiguous()
            _, out = self._encode(x, 8, want_states=True)
            return out

    @torch.no_grad()
    def generate_from_embeds(self, inputs_embeds: torch.Tensor, max_length: int = 20) -> torch.Tensor:
        with self._lock:
            x = inputs_embeds.to(self.transformer.device, torch.float32).contiguous()
            sess, _ = self._encode(x, max_length)
            tokens = torch.empty((x.shape[0], max_length), dtype=torch.long, device=x.device)
            out_len = C.c_int(0)
            with torch.cuda.device(x.device):
                native.check(native.load().m2m_generate_greedy(sess, max_length, tokens.data_ptr(), C.byref(out_len),
                                                               native.stream_handle(x.device)), "m2m_generate_greedy")
            return tokens[:, : out_len.value]

    @torch.no_grad()
    def logits_from_embeds(self, inputs_embeds: torch.Tensor, decoder_input_ids: torch.Tensor) -> torch.Tensor:
        with self._lock:
            x = inputs_embeds.to(self.transformer.device, torch.float32).contiguous()
            ids = decoder_input_ids.to(x.device, torch.long).contiguous()
            B, Ld = ids.shape
            sess, _ = self._encode(x, Ld)
            logits = torch.empty((B, Ld, self.geometry.vocab_size), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                native.check(native.load().m2m_decode_forced(sess, ids.data_ptr(), Ld, logits.data_ptr(),
                                                             native.stream_handle(x.device)), "m2m_decode_forced")
            return logits

    def repack_stats(self):
        """(re-packings, rows moved) of the last greedy decode on the current session: how often the live rows were moved into
        the first slots after a quarter of them had emitted EOS (``m2m_session_repack_stats``)."""
        with self._lock:
            if self._session is None:
                return 0, 0
            a, b = C.c_int(0), C.c_int(0)
            native.check(native.load().m2m_session_repack_stats(self._session, C.byref(a), C.byref(b)), "m2m_session_repack_stats")
            return int(a.value), int(b.value)

    def bench_kernel(self, which: int, self_len: int, iters: int):
        """Time one decode kernel in isolation on the current session (after an encode):
        returns (avg microseconds per launch, algorithmic bytes per launch)."""
        with self._lock:
            if self._session is None:
                raise native.NativeError("bench_kernel needs a prior generate/encode on this model")
            us, nbytes = C.c_float(0), C.c_int64(0)
            dev = self.transformer.device
            with torch.cuda.device(dev):
                native.check(native.load().m2m_bench_kernel(self._session, which, self_len, iters, C.byref(us),
                                                            C.byref(nbytes), native.stream_handle(dev)),
                             "m2m_bench_kernel")
            return float(us.value), int(nbytes.value)

    # -- reference API -------------------------------------------------------
    def forward(self, inputs: ModelInputs, **kwargs):
        """Teacher-forced pass (ref transformer.py:28-39): labels from the tokenizer, pad -> -100,
        decoder inputs = shift_right(labels) (hf: modeling_t5.py:618-637), loss = mean CE over
        non-ignored positions.  Inference-only: no autograd graph is built (training is a
        later row of SURVEY.md §8f)."""
        if kwargs:
            raise NotImplementedError(f"unsupported forward kwargs on the MI355X path: {sorted(kwargs)}")
        g = self.geometry
        labels = self.tokenizer(inputs.notes_batch)
        labels[labels == g.pad_token_id] = -100
        labels = labels.to(self.transformer.device)
        encoder_inputs = self.encoder_inputs(inputs)
        dec_in = torch.full_like(labels, g.decoder_start_token_id)
        dec_in[:, 1:] = labels[:, :-1]
        dec_in[dec_in == -100] = g.pad_token_id
        logits = self.logits_from_embeds(encoder_inputs, dec_in)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, g.vocab_size), labels.reshape(-1), ignore_index=-100)
        return Seq2SeqOutput(loss=loss, logits=logits)

    _GENERATE_DEFAULT_MAX_LENGTH = 20   # HF GenerationConfig default when max_length is not given

    def generate(self, inputs: ModelInputs, **kwargs) -> torch.Tensor:
        """Greedy decode (ref transformer.py:41-45).  The reference only ever passes
        ``max_length`` (ref model.py:58,134); sampling/beam arguments are rejected loudly."""
        max_length = int(kwargs.pop("max_length", self._GENERATE_DEFAULT_MAX_LENGTH))
        if kwargs.pop("do_sample", False) or int(kwargs.pop("num_beams", 1)) != 1:
            raise NotImplementedError("only greedy decoding is implemented on the MI355X path")
        if kwargs:
            raise NotImplementedError(f"unsupported generate kwargs on the MI355X path: {sorted(kwargs)}")
        encoder_inputs = self.encoder_inputs(inputs)
        return self.generate_from_embeds(encoder_inputs, max_length=max_length)
