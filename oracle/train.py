"""Oracle: training step (test infrastructure, see oracle/__init__.py).

Follows ref: music2midi/model.py:27-43 — ``loss = T5Transformer.forward(inputs).loss`` (ref
transformer.py:28-39: HF ``T5ForConditionalGeneration`` forward with labels, transformers 4.34.0),
``loss.backward()`` and ``transformers.optimization.Adafactor(params, warmup_init=True)`` +
``AdafactorSchedule``.

* ``T5TrainOracle`` is a differentiable torch-CPU restatement of the teacher-forced forward (batched over
  positions, explicit softmax attention) on leaf tensors keyed like the reference's state dict under
  ``model.``; gradients come from torch autograd.
* ``AdafactorOracle`` restates ``transformers/optimization.py`` ``Adafactor.step`` for the reference's
  arguments (lr=None, eps=(1e-30, 1e-3), clip_threshold=1.0, decay_rate=-0.8, beta1=None, weight_decay=0,
  scale_parameter=True, relative_step=True, warmup_init=True).

Pinned in the build container against HuggingFace itself — ``T5ForConditionalGeneration`` (eager, untied
head) loss + autograd gradients and ``transformers.optimization.Adafactor`` — by
tests/golden/make_golden.py (``train`` case); tests/test_oracle_golden.py re-checks the fixture.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .t5 import gelu_new, relative_position_bucket

_U64 = np.uint64
_GOLDEN = 0x9E3779B97F4A7C15
SITE_ENC, SITE_DEC, SITE_EMB, SITE_FIN = 0, 1000, 900, 901
PL_PROBS_SELF, PL_SELF_OUT, PL_PROBS_CROSS, PL_CROSS_OUT, PL_MID, PL_FF_OUT = 1, 2, 3, 4, 5, 6


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _U64(_GOLDEN)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


class DropoutMasks:
    """The device's counter-based dropout masks, regenerated on the host (csrc/common.h drop_keep): four consecutive elements
    share one hash — element i of a site is kept iff the 16-bit field (i & 3) of splitmix64(key(site) + (i >> 2)) is
    >= (p * 2^32) >> 16; kept values are scaled by 1 / (1 - p)."""

    def __init__(self, p: float, seed: int, call_index: int = 0):
        self.p = float(p)
        self.thresh = int(float(np.float32(p)) * 4294967296.0) if p > 0 else 0
        self.scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
        self.step_key = _splitmix64(np.asarray([(seed + call_index) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]

    def mask(self, site: int, n: int) -> torch.Tensor:
        """float tensor [n]: 0 or scale."""
        if self.thresh == 0:
            return torch.ones(n)
        with np.errstate(over="ignore"):
            key = _splitmix64(np.asarray([self.step_key + _U64((site * _GOLDEN) & 0xFFFFFFFFFFFFFFFF)], dtype=np.uint64))[0]
            i = np.arange(n, dtype=np.uint64)
            h = _splitmix64(key + (i >> _U64(2)))
        field = (h >> (_U64(16) * (i & _U64(3)))) & _U64(0xFFFF)
        keep = field >= _U64(self.thresh >> 16)
        return torch.from_numpy(keep.astype(np.float32) * np.float32(self.scale))


class _MxLinear(torch.autograd.Function):
    """y = x . W^T as the device's fp8 training mode computes it AND differentiates it (csrc/train.hip Ops::mm / dX / dW,
    csrc/mx8.hip).  Each of the three products quantises ITS OWN operands along ITS OWN reduction dimension:

    * forward  y  = Qk(bf16(x)) . Qk(W)^T          — blocks of 32 along the input features K; the activation is quantised from
                                                     its bf16 storage, the weight from the fp32 master copy;
    * dX       dx = Qn(bf16(dy)) . Qn(W^T)^T       — blocks along the OUTPUT features N: a different quantisation of the same
                                                     weight (``mxq_weights_kernel`` keeps both images), and dy itself is
                                                     quantised (format ``grad_fmt``);
    * dW       dw = bf16(dy)^T . bf16(x)           — the grouped bf16 weight-gradient launch (default), or with ``dw_fp8``
                                                     Qm(dy^T) . Qm(x^T)^T with blocks along the rows M (zero-padded).

    A straight-through estimator over the forward alone (round 2's emulation) differentiates a different function: it leaves dy
    and the second weight image unquantised, which is where the 0.93-0.95 gradient cosine of round 2 came from."""

    @staticmethod
    def forward(ctx, x, w, dw_fp8, grad_fmt):
        from .mx8 import mx_quant_dequant
        xb = x.detach().bfloat16().float()
        ctx.save_for_backward(xb, w.detach())
        ctx.dw_fp8, ctx.grad_fmt = dw_fp8, grad_fmt
        return mx_quant_dequant(xb, "e4m3") @ mx_quant_dequant(w.detach(), "e4m3").T

    @staticmethod
    def backward(ctx, dy):
        from .mx8 import mx_quant_dequant
        xb, w = ctx.saved_tensors
        dyb = dy.bfloat16().float()
        dx = mx_quant_dequant(dyb, ctx.grad_fmt) @ mx_quant_dequant(w.T.contiguous(), "e4m3").T
        dy2, x2 = dyb.reshape(-1, dyb.shape[-1]), xb.reshape(-1, xb.shape[-1])
        if ctx.dw_fp8:
            dw = mx_quant_dequant(dy2.T.contiguous(), ctx.grad_fmt) @ mx_quant_dequant(x2.T.contiguous(), "e4m3").T
        else:
            dw = dy2.T @ x2
        return dx, dw, None, None


class _Round(torch.autograd.Function):
    """bfloat16 rounding where the device's bf16 / fp8 training modes STORE bfloat16: fwd / bwd select whether the value, its
    gradient, or both pass through a bf16 buffer on the device (csrc/train.hip: activations h, qkv, P, attention output, gate pair,
    gated activation are stored in bf16 and so are the gradients dxT, dO, dqkv, dS, dmid, dab, dlogits; the residual stream, its
    gradient and the norm inputs' gradient dh stay fp32)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float() if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None, None


class _SoftmaxBf16(torch.autograd.Function):
    """Attention probabilities as the stripe kernels keep them: P = bf16(softmax(s)) is what is stored, and the backward forms
    dS = bf16(P o (dP - sum_k P o dP)) FROM THAT STORED P (csrc/train.hip attn_stripe_kernel<BWD>)."""

    @staticmethod
    def forward(ctx, s):
        p = torch.softmax(s, dim=-1).bfloat16().float()
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        return (p * (dp - (p * dp).sum(-1, keepdim=True))).bfloat16().float()


def leaf_params(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    """state dict (numpy, keys ``transformer.*`` / ``conditioning.*``) -> fp32 leaf tensors with requires_grad."""
    return {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).clone().requires_grad_(True)
            for k, v in sd.items() if k.startswith(("transformer.", "conditioning."))
            and not k.endswith(("encoder.embed_tokens.weight", "decoder.embed_tokens.weight"))}


class T5TrainOracle:
    def __init__(self, geom, params: Dict[str, torch.Tensor]):
        self.g = geom
        self.p = params
        self.masks = None
        self.mx8 = False          # True: emulate the fp8 training mode's projection products (see _lin)
        self.mx8_dw = False       # ... the weight gradients on MXFP8 too (M2M_FP8_PARTS=fwd,dx,dw)
        self.mx8_grad_fmt = "e4m3"
        # True: bfloat16 rounding at every point where the device's bf16 (and fp8) training modes store bfloat16 — values AND
        # gradients (_Round, _SoftmaxBf16).  Without it the oracle is the fp32 reference arithmetic.  The fp8 comparison needs it:
        # an MXFP8 element is 3 mantissa bits, so two runs whose quantiser INPUTS differ by bf16 rounding noise flip a few percent
        # of the fp8 roundings in every layer, and after twelve layers the two gradients are two different noisy draws around the
        # fp32 one (round 2's cosine 0.93-0.95); with the inputs agreeing to fp32 noise the device has to reproduce the emulation.
        self.bf16 = False

    def w(self, name: str) -> torch.Tensor:
        return self.p["transformer." + name]

    def _norm(self, x, w):
        return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.g.eps))

    def _heads(self, x):
        B, L, _ = x.shape
        return x.view(B, L, self.g.num_heads, self.g.d_kv).transpose(1, 2)

    def _bias(self, table, q_len, k_len, bidirectional):
        rel = torch.arange(k_len)[None, :] - torch.arange(q_len)[:, None]
        b = relative_position_bucket(rel, bidirectional, self.g.num_buckets, self.g.max_distance)
        return table[b].permute(2, 0, 1).unsqueeze(0)

    def _drop(self, x, site):
        """dropout at `site` with the device's mask (element index = flat index of the row-major activation)."""
        if self.masks is None:
            return x
        return x * self.masks.mask(site, x.numel()).view(x.shape)

    def _r(self, x, fwd=True, bwd=True):
        return _Round.apply(x, fwd, bwd) if self.bf16 else x

    def _lin(self, x, wname):
        """x @ W^T.  With ``self.mx8`` the product and BOTH of its gradient products are the fp8 training mode's
        (``_MxLinear``: every product quantises its own operands along its own reduction dimension, as csrc/train.hip does);
        lm_head is not a ``_lin`` product: it stays in the bf16 mode's arithmetic on the device too."""
        w = self.w(wname)
        if getattr(self, "mx8", False):
            y = _MxLinear.apply(x, w, bool(self.mx8_dw), self.mx8_grad_fmt)
        else:
            y = x @ self._r(w, True, False).T           # bf16 mode: the product reads a bf16 copy of the fp32 master weight
        cap = getattr(self, "capture", None)            # tests: {weight name: {}} -> the product's input, output and output gradient
        if cap is not None and wname in cap:
            rec = cap[wname]
            rec["x"], rec["y"] = x.detach().clone(), y.detach().clone()
            y.register_hook(lambda g, rec=rec: rec.__setitem__("dy", g.detach().clone()))
        return y

    def _attn(self, hq, hkv, prefix, bias, site_probs=-1):
        B, Lq, _ = hq.shape
        q = self._heads(self._r(self._lin(hq, prefix + ".q.weight")))
        k = self._heads(self._r(self._lin(hkv, prefix + ".k.weight")))
        v = self._heads(self._r(self._lin(hkv, prefix + ".v.weight")))
        s = q @ k.transpose(2, 3)                      # no 1/sqrt(d_kv) (hf: modeling_t5.py:197)
        if bias is not None:
            s = s + bias
        pr = _SoftmaxBf16.apply(s) if self.bf16 else torch.softmax(s, dim=-1)
        if self.masks is not None:                     # the device stores probabilities with a row pitch of ceil8(Sk)
            Sk = pr.shape[-1]
            ldp = (Sk + 7) // 8 * 8
            m = self.masks.mask(site_probs, B * self.g.num_heads * Lq * ldp).view(B, self.g.num_heads, Lq, ldp)[..., :Sk]
            pr = pr * m
        if self.bf16 and self.masks is not None:
            pr = self._r(pr, True, False)              # the dropped copy is rounded again after the 1/(1-p) scale
        o = self._r((pr @ v).transpose(1, 2).reshape(B, Lq, self.g.inner_dim))
        return self._r(self._lin(o, prefix + ".o.weight"), False, True)      # the branch's gradient enters as bf16 (dxT)

    def _ffn(self, h, prefix, site_mid=-1):
        mid = self._r(gelu_new(self._r(self._lin(h, prefix + ".wi_0.weight"))) * self._r(self._lin(h, prefix + ".wi_1.weight")))
        return self._r(self._lin(self._drop(mid, site_mid), prefix + ".wo.weight"), False, True)

    def encoder_inputs(self, feats: torch.Tensor, cond_idx: torch.Tensor) -> torch.Tensor:
        """ref: music2midi/input.py:57-59 — conditioning rows first, then the (constant) log-mel rows."""
        rows = [self.p[f"conditioning.embeds.{i}.weight"][cond_idx[:, i]] for i in range(cond_idx.shape[1])]
        return torch.cat([torch.stack(rows, dim=1), feats], dim=1)

    def forward(self, feats: torch.Tensor, cond_idx: torch.Tensor, labels: torch.Tensor, masks: "DropoutMasks" = None):
        """-> (mean CE over labels != -100, logits [B, Ld, V]).  `masks`: dropout as hf: modeling_t5.py places it
        (embeddings, attention probabilities, every residual branch, gated activation, final norms) with the device's masks."""
        g = self.g
        self.masks = masks
        x = self._drop(self.encoder_inputs(feats, cond_idx), SITE_ENC + SITE_EMB)
        S = x.shape[1]
        ebias = self._bias(self.w("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"), S, S, True)
        for i in range(g.num_layers):
            p = f"encoder.block.{i}.layer"
            st = SITE_ENC + 16 * i
            h = self._r(self._norm(x, self.w(f"{p}.0.layer_norm.weight")), True, False)
            x = x + self._drop(self._attn(h, h, f"{p}.0.SelfAttention", ebias, st + PL_PROBS_SELF), st + PL_SELF_OUT)
            x = x + self._drop(self._ffn(self._r(self._norm(x, self.w(f"{p}.1.layer_norm.weight")), True, False), f"{p}.1.DenseReluDense", st + PL_MID), st + PL_FF_OUT)
        enc = self._drop(self._r(self._norm(x, self.w("encoder.final_layer_norm.weight")), True, False), SITE_ENC + SITE_FIN)
        B, Ld = labels.shape
        dec_in = torch.full((B, Ld), g.decoder_start_token_id, dtype=torch.long)
        dec_in[:, 1:] = labels[:, :-1]
        dec_in[dec_in == -100] = g.pad_token_id
        y = self._drop(self.w("shared.weight")[dec_in], SITE_DEC + SITE_EMB)
        dbias = self._bias(self.w("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"), Ld, Ld, False)
        causal = torch.full((Ld, Ld), float("-inf")).triu(1)
        for i in range(g.num_decoder_layers):
            p = f"decoder.block.{i}.layer"
            st = SITE_DEC + 16 * i
            h = self._r(self._norm(y, self.w(f"{p}.0.layer_norm.weight")), True, False)
            y = y + self._drop(self._attn(h, h, f"{p}.0.SelfAttention", dbias + causal, st + PL_PROBS_SELF), st + PL_SELF_OUT)
            h = self._r(self._norm(y, self.w(f"{p}.1.layer_norm.weight")), True, False)
            y = y + self._drop(self._attn(h, enc, f"{p}.1.EncDecAttention", None, st + PL_PROBS_CROSS), st + PL_CROSS_OUT)
            y = y + self._drop(self._ffn(self._r(self._norm(y, self.w(f"{p}.2.layer_norm.weight")), True, False), f"{p}.2.DenseReluDense", st + PL_MID), st + PL_FF_OUT)
        hD = self._drop(self._r(self._norm(y, self.w("decoder.final_layer_norm.weight")), True, False), SITE_DEC + SITE_FIN)
        logits = self._r(hD @ self._r(self.w("lm_head.weight"), True, False).T, False, True)      # logits fp32, their gradient stored in bf16
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, g.vocab_size), labels.reshape(-1), ignore_index=-100)
        return loss, logits

    def loss_and_grads(self, feats, cond_idx, labels, masks: "DropoutMasks" = None):
        for t in self.p.values():
            t.grad = None
        loss, logits = self.forward(feats, cond_idx, labels, masks)
        loss.backward()
        return loss.detach(), logits.detach(), {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in self.p.items()}


class AdafactorOracle:
    """transformers.optimization.Adafactor.step restated (see the module docstring for the arguments)."""

    def __init__(self, params: Dict[str, torch.Tensor]):
        self.params = params
        self.state = {k: {"step": 0} for k in params}

    @staticmethod
    def _rms(t):
        return t.norm(2) / (t.numel() ** 0.5)

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        eps1, eps2, clip, decay = 1e-30, 1e-3, 1.0, -0.8
        for k, p in self.params.items():
            g = grads[k].float()
            st = self.state[k]
            factored = g.dim() >= 2
            if st["step"] == 0:
                if factored:
                    st["row"] = torch.zeros(g.shape[:-1])
                    st["col"] = torch.zeros(g.shape[:-2] + g.shape[-1:])
                else:
                    st["v"] = torch.zeros_like(g)
            st["step"] += 1
            t = st["step"]
            rho = min(1e-6 * t, 1.0 / math.sqrt(t))                    # relative_step with warmup_init
            lr = max(eps2, float(self._rms(p))) * rho                  # scale_parameter
            beta2t = 1.0 - math.pow(t, decay)
            upd = g ** 2 + eps1
            if factored:
                st["row"].mul_(beta2t).add_(upd.mean(dim=-1), alpha=1.0 - beta2t)
                st["col"].mul_(beta2t).add_(upd.mean(dim=-2), alpha=1.0 - beta2t)
                r = (st["row"] / st["row"].mean(dim=-1, keepdim=True)).rsqrt_().unsqueeze(-1)
                c = st["col"].unsqueeze(-2).rsqrt()
                upd = r * c * g
            else:
                st["v"].mul_(beta2t).add_(upd, alpha=1.0 - beta2t)
                upd = st["v"].rsqrt() * g
            upd = upd / (self._rms(upd) / clip).clamp(min=1.0)
            p.add_(upd * lr, alpha=-1.0)
