"""Oracle: T5 encoder-decoder + greedy decode (test infrastructure, see oracle/__init__.py).

CPU restatement, in plain torch-CPU tensor ops, of what
ref: music2midi/transformer.py:28-45 calls in transformers 4.34.0
(``T5ForConditionalGeneration`` forward / ``generate``).  ``hf:`` citations are
to the transformers 5.15.0 copy installed in the build container, read with the
three 4.34.0 deltas of SURVEY.md §0.3 applied: separate (untied) ``lm_head``
and no ``d_model**-0.5`` output scaling, explicit (eager) attention, tuple-like
KV cache with argmax / pad-after-EOS / stop-at-max_length greedy semantics.

``emulate="bf16"`` rounds to bfloat16 at exactly the points where the device's
bf16 mode stores bf16 (weights, GEMM inputs, KV caches); all accumulation,
norms, softmax and GELU stay fp32 — see DESIGN.md section 2 (precision modes).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int,
                             max_distance: int) -> torch.Tensor:
    """hf: models/t5/modeling_t5.py:217-262 (rel = key_pos - query_pos)."""
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = torch.abs(rel)
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """hf: activations.py NewGELUActivation."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class T5Oracle:
    """Weights come as a dict of numpy arrays keyed like HF's state dict with a
    ``transformer.`` prefix (see music2midi_amd.synth.t5_state_dict)."""

    def __init__(self, geom, sd: Dict[str, np.ndarray], emulate: str = "fp32"):
        assert emulate in ("fp32", "bf16")
        self.g = geom
        self.emulate = emulate
        self.w = {k[len("transformer."):]: self._rnd(torch.from_numpy(np.asarray(v, dtype=np.float32)).clone())
                  if v.ndim == 2 and "relative_attention_bias" not in k and "shared" not in k
                  else torch.from_numpy(np.asarray(v, dtype=np.float32)).clone()
                  for k, v in sd.items() if k.startswith("transformer.")}
        # embedding table feeds the residual stream (not a GEMM input): stays fp32.

    # -- precision emulation -------------------------------------------------
    def _rnd(self, x: torch.Tensor) -> torch.Tensor:
        return x if self.emulate == "fp32" else x.bfloat16().float()

    # -- pieces --------------------------------------------------------------
    def rmsnorm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """hf: modeling_t5.py:59-72 (no mean subtraction, no bias)."""
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + self.g.eps))

    def _heads(self, x: torch.Tensor) -> torch.Tensor:  # [B,L,inner] -> [B,H,L,dk]
        B, L, _ = x.shape
        return x.view(B, L, self.g.num_heads, self.g.d_kv).transpose(1, 2)

    def _bias(self, table: torch.Tensor, q_len: int, k_len: int, bidirectional: bool,
              q_offset: int = 0) -> torch.Tensor:
        """hf: modeling_t5.py:264-279 -> [1,H,q_len,k_len]."""
        ctx = torch.arange(q_len)[:, None] + q_offset
        mem = torch.arange(k_len)[None, :]
        b = relative_position_bucket(mem - ctx, bidirectional, self.g.num_buckets, self.g.max_distance)
        return table[b].permute(2, 0, 1).unsqueeze(0)

    def _ffn(self, x: torch.Tensor, p: str) -> torch.Tensor:
        """hf: modeling_t5.py:106-123 gated-GELU FFN."""
        w = self.w
        h = self._rnd(x)
        g = self._rnd(gelu_new(h @ w[f"{p}.wi_0.weight"].T) * (h @ w[f"{p}.wi_1.weight"].T))
        return g @ w[f"{p}.wo.weight"].T

    # -- encoder -------------------------------------------------------------
    def encode(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        """hf: modeling_t5.py:663-750 with inputs_embeds, mask all ones, dropout off."""
        w = self.w
        x = inputs_embeds.float()
        B, S, _ = x.shape
        bias = self._bias(w["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                          S, S, True)
        for i in range(self.g.num_layers):
            p = f"encoder.block.{i}"
            a = f"{p}.layer.0.SelfAttention"
            h = self._rnd(self.rmsnorm(x, w[f"{p}.layer.0.layer_norm.weight"]))
            q = self._heads(self._rnd(h @ w[f"{a}.q.weight"].T))
            k = self._heads(self._rnd(h @ w[f"{a}.k.weight"].T))
            v = self._heads(self._rnd(h @ w[f"{a}.v.weight"].T))
            scores = q @ k.transpose(2, 3) + bias            # no 1/sqrt(dk) (hf :197)
            pr = self._rnd(torch.softmax(scores.float(), dim=-1))
            o = (pr @ v).transpose(1, 2).reshape(B, S, self.g.inner_dim)
            x = x + self._rnd(o) @ w[f"{a}.o.weight"].T
            h = self.rmsnorm(x, w[f"{p}.layer.1.layer_norm.weight"])
            x = x + self._ffn(h, f"{p}.layer.1.DenseReluDense")
        return self.rmsnorm(x, w["encoder.final_layer_norm.weight"])

    # -- decoder -------------------------------------------------------------
    def _cross_kv(self, enc_out: torch.Tensor):
        w = self.w
        e = self._rnd(enc_out)
        kv = []
        for i in range(self.g.num_decoder_layers):
            a = f"decoder.block.{i}.layer.1.EncDecAttention"
            kv.append((self._heads(self._rnd(e @ w[f"{a}.k.weight"].T)),
                       self._heads(self._rnd(e @ w[f"{a}.v.weight"].T))))
        return kv

    def _new_cache(self, B: int, max_len: int):
        g = self.g
        return [(torch.zeros(B, g.num_heads, max_len, g.d_kv), torch.zeros(B, g.num_heads, max_len, g.d_kv))
                for _ in range(g.num_decoder_layers)]

    def _dec_bias_table(self, max_len: int) -> torch.Tensor:
        """Causal rel-pos bias as a function of n = q_pos - k_pos >= 0 -> [H, max_len]."""
        table = self.w["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        n = torch.arange(max_len)
        b = relative_position_bucket(-n, False, self.g.num_buckets, self.g.max_distance)
        return table[b].T.contiguous()

    def decode_step(self, tokens: torch.Tensor, t: int, self_kv, cross_kv, bias_tab) -> torch.Tensor:
        """One incremental decoder step at position t (hf: modeling_t5.py:448-509,1031-1047).

        tokens [B] int64 -> logits [B, V] fp32.  Appends K/V at slot t.
        """
        w, g = self.w, self.g
        B = tokens.shape[0]
        x = w["shared.weight"][tokens]                       # [B, d]; T5 does not scale embeddings
        nrel = (t - torch.arange(t + 1))                      # q_pos - k_pos
        for i in range(g.num_decoder_layers):
            p = f"decoder.block.{i}"
            a = f"{p}.layer.0.SelfAttention"
            h = self._rnd(self.rmsnorm(x, w[f"{p}.layer.0.layer_norm.weight"]))
            q = (h @ w[f"{a}.q.weight"].T).view(B, g.num_heads, 1, g.d_kv)
            K, V = self_kv[i]
            K[:, :, t] = self._rnd(h @ w[f"{a}.k.weight"].T).view(B, g.num_heads, g.d_kv)
            V[:, :, t] = self._rnd(h @ w[f"{a}.v.weight"].T).view(B, g.num_heads, g.d_kv)
            scores = q @ K[:, :, : t + 1].transpose(2, 3) + bias_tab[:, nrel][None, :, None, :]
            pr = torch.softmax(scores.float(), dim=-1)
            o = (pr @ V[:, :, : t + 1]).reshape(B, g.inner_dim)
            x = x + self._rnd(o) @ w[f"{a}.o.weight"].T

            c = f"{p}.layer.1.EncDecAttention"
            h = self._rnd(self.rmsnorm(x, w[f"{p}.layer.1.layer_norm.weight"]))
            q = (h @ w[f"{c}.q.weight"].T).view(B, g.num_heads, 1, g.d_kv)
            CK, CV = cross_kv[i]
            pr = torch.softmax((q @ CK.transpose(2, 3)).float(), dim=-1)   # zero bias, no mask (hf :337-342)
            o = (pr @ CV).reshape(B, g.inner_dim)
            x = x + self._rnd(o) @ w[f"{c}.o.weight"].T

            h = self.rmsnorm(x, w[f"{p}.layer.2.layer_norm.weight"])
            x = x + self._ffn(h, f"{p}.layer.2.DenseReluDense")
        h = self._rnd(self.rmsnorm(x, w["decoder.final_layer_norm.weight"]))
        return (h @ w["lm_head.weight"].T).float()           # untied head, no d_model**-0.5 (4.34.0)

    # -- public --------------------------------------------------------------
    @torch.no_grad()
    def generate(self, inputs_embeds: torch.Tensor, max_length: int, return_margins: bool = False,
                 enc_out: Optional[torch.Tensor] = None, logits_hook=None):
        """Greedy decode (hf: generation/utils.py:2783-2973, do_sample=False).

        Returns LongTensor [B, L], L <= max_length; column 0 is the start token;
        rows that finished early are right-padded with pad_token_id; generation
        stops when every row has emitted EOS or L == max_length.
        ``logits_hook(t, logits)`` sees every step's fp32 logits [B, V] (fixture generation: the greedy
        trajectory's logits ARE the teacher-forced logits along its own ids).
        """
        g = self.g
        if enc_out is None:
            enc_out = self.encode(inputs_embeds)
        B = enc_out.shape[0]
        cross = self._cross_kv(enc_out)
        cache = self._new_cache(B, max_length)
        bias_tab = self._dec_bias_table(max_length)
        ids = torch.full((B, 1), g.decoder_start_token_id, dtype=torch.long)
        unfinished = torch.ones(B, dtype=torch.long)
        margins = []
        t = 0
        while ids.shape[1] < max_length:
            logits = self.decode_step(ids[:, -1], t, cache, cross, bias_tab)
            if logits_hook is not None:
                logits_hook(t, logits)
            nxt = torch.argmax(logits, dim=-1)
            if return_margins:
                top2 = torch.topk(logits, 2, dim=-1).values
                margins.append((top2[:, 0] - top2[:, 1]))
            nxt = nxt * unfinished + g.pad_token_id * (1 - unfinished)
            ids = torch.cat([ids, nxt[:, None]], dim=1)
            unfinished = unfinished & (nxt != g.eos_token_id).long()
            t += 1
            if unfinished.max() == 0:
                break
        if return_margins:
            return ids, (torch.stack(margins, dim=1) if margins else torch.zeros(B, 0))
        return ids

    @torch.no_grad()
    def forward(self, inputs_embeds: torch.Tensor, labels: torch.Tensor,
                enc_out: Optional[torch.Tensor] = None):
        """Teacher-forced logits + loss (hf: modeling_t5.py:1011-1054).

        labels [B, Ld] with -100 = ignore.  decoder_input_ids = shift_right(labels)
        (hf :618-637).  Computed incrementally, one position at a time, through the
        same decode_step as generate().
        """
        g = self.g
        if enc_out is None:
            enc_out = self.encode(inputs_embeds)
        B, Ld = labels.shape
        dec_in = torch.full((B, Ld), g.decoder_start_token_id, dtype=torch.long)
        dec_in[:, 1:] = labels[:, :-1]
        dec_in[dec_in == -100] = g.pad_token_id
        cross = self._cross_kv(enc_out)
        cache = self._new_cache(B, Ld)
        bias_tab = self._dec_bias_table(Ld)
        logits = torch.stack([self.decode_step(dec_in[:, t], t, cache, cross, bias_tab)
                              for t in range(Ld)], dim=1)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, g.vocab_size), labels.reshape(-1),
                                                 ignore_index=-100)
        return loss, logits
