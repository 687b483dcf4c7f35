"""GPU: the training step's whole-head attention kernels (csrc/attn_train.hip) on their own, against a plain PyTorch fp32
reference of the same op — hf: modeling_t5.py:159-170 T5Attention (scores = Q K^T without 1/sqrt(d), + relative-position bias,
causal mask for the decoder, softmax in fp32, dropout on the probabilities, context = P~ V) and torch autograd for its backward.
The dropout masks are the step's counter-based hash, regenerated on the host by oracle/train.py::DropoutMasks with the element index
the kernels use, so forward AND backward are checked with dropout on against autograd over the same masks.  Inputs are bf16 (the
training step's storage type); the kernels round the probabilities to bf16 as matrix operands, so the bars are bf16-sized."""
import ctypes as C

import numpy as np
import pytest
import torch

from music2midi_amd import native, synth

pytestmark = pytest.mark.gpu

GOLDEN = 0x9E3779B97F4A7C15


def _inputs(B, H, Sq, Sk, seed, scale=0.6):
    def t(tag, S):
        return torch.from_numpy(synth.normal(seed, tag, (B, S, H * 64), scale)).bfloat16()
    q, k, v = t("q", Sq), t("k", Sk), t("v", Sk)
    bias = torch.from_numpy(synth.normal(seed, "bias", (H, Sq + Sk - 1), 1.5)).float()
    return q, k, v, bias


def _masks(p, seed, site, B, H, Sq, Sk):
    """[B, H, Sq, Sk] float: 0 or 1 / (1 - p), element index ((b*H + h)*Sq + q) * round_up_8(Sk) + k."""
    from oracle.train import DropoutMasks
    dm = DropoutMasks(p, seed, 0)
    ldp = (Sk + 7) // 8 * 8
    m = dm.mask(site, B * H * Sq * ldp).view(B, H, Sq, ldp)[..., :Sk]
    return m, int(dm.step_key), (site * GOLDEN) & 0xFFFFFFFFFFFFFFFF


def _keep_words(B, H, Sq, Sk):
    return torch.zeros((B * H, (Sk + 31) // 32, (Sq + 31) // 32 * 32), dtype=torch.int32, device="cuda")


def _reference(q, k, v, bias, causal, mask):
    B, Sq, HD = q.shape
    H, Sk = HD // 64, k.shape[1]
    qf, kf, vf = (x.float().view(B, -1, H, 64).transpose(1, 2) for x in (q, k, v))      # [B, H, S, 64]
    s = qf @ kf.transpose(2, 3)
    if bias is not None:
        rel = torch.arange(Sk)[None, :] - torch.arange(Sq)[:, None] + Sq - 1
        s = s + bias[:, rel][None]
    if causal:
        s = s + torch.full((Sq, Sk), float("-inf")).triu(1)
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    if mask is not None:
        p = p * mask
    o = (p @ vf).transpose(1, 2).reshape(B, Sq, HD)
    return o, lse


CASES = [  # B, H, Sq, Sk, causal, bias, dropout
    (1, 1, 32, 32, False, False, 0.0),
    (2, 2, 21, 21, False, True, 0.0),
    (2, 2, 14, 14, True, True, 0.0),
    (1, 2, 33, 70, False, False, 0.0),
    (2, 8, 261, 261, False, True, 0.1),
    (2, 8, 256, 256, True, True, 0.1),
    (2, 8, 256, 261, False, False, 0.1),
    (1, 2, 280, 288, False, True, 0.0),
    (1, 1, 5, 3, False, False, 0.1),
    (2, 2, 2, 2, True, True, 0.1),
    (1, 2, 1, 1, True, True, 0.0),
    (3, 2, 9, 9, False, True, 0.1),
    (2, 2, 1, 11, False, False, 0.0),
]


@pytest.mark.parametrize("B,H,Sq,Sk,causal,use_bias,p", CASES)
def test_whole_head_attention_forward_matches_torch(B, H, Sq, Sk, causal, use_bias, p):
    native.require_gpu()
    lib = native.load()
    q, k, v, bias = _inputs(B, H, Sq, Sk, seed=B * 1000 + Sq)
    mask, step_key, salt = (None, 0, 0)
    if p > 0:
        mask, step_key, salt = _masks(p, 77, 3, B, H, Sq, Sk)
    o_ref, lse_ref = _reference(q, k, v, bias if use_bias else None, causal, mask)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    bd = bias.cuda() if use_bias else None
    out = torch.full((B, Sq, H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B * H, Sq), float("nan"), dtype=torch.float32, device="cuda")
    bits = _keep_words(B, H, Sq, Sk)
    native.check(lib.m2m_attn_head_fwd_bf16(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), bd.data_ptr() if use_bias else None, B, H, Sq, Sk,
                                            int(causal), float(p), C.c_uint64(step_key), C.c_uint64(salt), out.data_ptr(), lse.data_ptr(),
                                            bits.data_ptr(), native.stream_handle()), "m2m_attn_head_fwd_bf16")
    torch.cuda.synchronize()
    if p > 0:
        # the keep words the backward pass will read: bit k of word (key block j, query q) = mask[q, 32 j + k] != 0
        w = bits.cpu().view(B * H, (Sk + 31) // 32, (Sq + 31) // 32 * 32)[:, :, :Sq].to(torch.int64) & 0xFFFFFFFF
        keep = (mask.reshape(B * H, Sq, Sk) != 0)
        for j in range((Sk + 31) // 32):
            if causal and 32 * j > Sq - 1:
                continue
            for kk in range(min(32, Sk - 32 * j)):
                got = ((w[:, j, :] >> kk) & 1).bool()
                rows = torch.arange(Sq) // 32 >= j if causal else torch.ones(Sq, dtype=torch.bool)      # (causal: tiles past the diagonal are never visited)
                assert torch.equal(got[:, rows], keep[:, rows, 32 * j + kk]), (j, kk)
    o_dev, lse_dev = out.float().cpu(), lse.cpu().view(B, H, Sq)
    assert torch.isfinite(o_dev).all() and torch.isfinite(lse_dev).all()
    e_lse = float((lse_dev - lse_ref).abs().max())
    e_o = float((o_dev - o_ref).abs().max() / (o_ref.abs().max() + 1e-9))
    print(f"fwd B={B} H={H} Sq={Sq} Sk={Sk} causal={causal} bias={use_bias} p={p}: max |lse err| {e_lse:.2e} (|lse| up to {float(lse_ref.abs().max()):.1f}), "
          f"max |O err| / max |O| {e_o:.2e}")
    assert e_lse < 2e-3 * max(1.0, float(lse_ref.abs().max()))
    assert e_o < 1.5e-2


def _reference_bwd(q, k, v, bias, causal, mask, d_out):
    """autograd over the fp32 reference: (dq, dk, dv, dS [B, H, Sq, Sk])."""
    B, Sq, HD = q.shape
    H, Sk = HD // 64, k.shape[1]
    qf, kf, vf = (x.float().clone().requires_grad_(True) for x in (q, k, v))
    qh, kh, vh = (x.view(B, -1, H, 64).transpose(1, 2) for x in (qf, kf, vf))
    s = qh @ kh.transpose(2, 3)
    if bias is not None:
        rel = torch.arange(Sk)[None, :] - torch.arange(Sq)[:, None] + Sq - 1
        s = s + bias[:, rel][None]
    s.retain_grad()
    sm = s + torch.full((Sq, Sk), float("-inf")).triu(1) if causal else s
    p = torch.softmax(sm, dim=-1)
    if mask is not None:
        p = p * mask
    o = (p @ vh).transpose(1, 2).reshape(B, Sq, HD)
    (o * d_out.float()).sum().backward()
    return qf.grad, kf.grad, vf.grad, s.grad


@pytest.mark.parametrize("B,H,Sq,Sk,causal,use_bias,p", [c for c in CASES if c[2] <= 288 and c[3] <= 288])
def test_whole_head_attention_backward_matches_autograd(B, H, Sq, Sk, causal, use_bias, p):
    native.require_gpu()
    lib = native.load()
    q, k, v, bias = _inputs(B, H, Sq, Sk, seed=B * 1000 + Sq)
    d_out = torch.from_numpy(synth.normal(B + Sq, "dout", (B, Sq, H * 64), 0.05)).bfloat16()
    mask, step_key, salt = (None, 0, 0)
    if p > 0:
        mask, step_key, salt = _masks(p, 77, 3, B, H, Sq, Sk)
    dq_ref, dk_ref, dv_ref, ds_ref = _reference_bwd(q, k, v, bias if use_bias else None, causal, mask, d_out)
    qd, kd, vd, dod = q.cuda(), k.cuda(), v.cuda(), d_out.cuda()
    bd = bias.cuda() if use_bias else None
    bptr = bd.data_ptr() if use_bias else None
    out = torch.empty((B, Sq, H * 64), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty((B * H, Sq), dtype=torch.float32, device="cuda")
    bits = _keep_words(B, H, Sq, Sk)
    native.check(lib.m2m_attn_head_fwd_bf16(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), bptr, B, H, Sq, Sk, int(causal), float(p), C.c_uint64(step_key),
                                            C.c_uint64(salt), out.data_ptr(), lse.data_ptr(), bits.data_ptr(), native.stream_handle()), "m2m_attn_head_fwd_bf16")
    nan = float("nan")
    dq = torch.full_like(qd, nan); dk = torch.full_like(kd, nan); dv = torch.full_like(vd, nan)
    nq, dl = (Sq + 31) // 32, Sk + 31
    diag = torch.full((B * H, nq, dl), nan, dtype=torch.float32, device="cuda") if use_bias else None
    native.check(lib.m2m_attn_head_bwd_bf16(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), lse.data_ptr(), dod.data_ptr(), bptr, B, H, Sq, Sk,
                                            int(causal), float(p), C.c_uint64(step_key), C.c_uint64(salt), bits.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                            diag.data_ptr() if use_bias else None, native.stream_handle()), "m2m_attn_head_bwd_bf16")
    torch.cuda.synchronize()
    errs = {}
    for name, dev, ref in (("dq", dq, dq_ref), ("dk", dk, dk_ref), ("dv", dv, dv_ref)):
        d = dev.float().cpu()
        assert torch.isfinite(d).all(), name
        errs[name] = float((d - ref).norm() / (ref.norm() + 1e-12)), float((d - ref).abs().max() / (ref.abs().max() + 1e-12))
    msg = ", ".join(f"{n} rel l2 {e[0]:.2e} / max {e[1]:.2e}" for n, e in errs.items())
    if use_bias:
        # expected diagonal sums from autograd's dS: entry x of query block i = sum over its rows rr of dS[32 i + rr][x - 31 + rr]
        dsr = ds_ref.reshape(B * H, Sq, Sk)
        want = torch.zeros((B * H, nq, dl))
        for i in range(nq):
            for rr in range(min(32, Sq - 32 * i)):
                want[:, i, 31 - rr: 31 - rr + Sk] += dsr[:, 32 * i + rr, :]
        dd = diag.cpu()
        assert torch.isfinite(dd).all()
        e_diag = float((dd - want).abs().max() / (want.abs().max() + 1e-12))
        msg += f", bias-gradient diagonals max err {e_diag:.2e}"
        assert e_diag < 2e-2
    print(f"bwd B={B} H={H} Sq={Sq} Sk={Sk} causal={causal} bias={use_bias} p={p}: {msg}")
    for n, e in errs.items():
        assert e[0] < 2e-2 and e[1] < 3e-2, (n, e)          # (bf16 operands; delta = rowsum(dO o O) from the bf16-rounded O: largest with three keys)
