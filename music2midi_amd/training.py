"""Training step on MI355X — what ref: music2midi/model.py:27-43 runs through pytorch-lightning:

    outputs = self.model(inputs); loss = outputs.loss          # teacher-forced T5 forward with labels
    loss.backward()                                            # (Lightning)
    Adafactor(self.parameters(), warmup_init=True).step()      # + AdafactorSchedule (relative step)

Forward, backward and the optimizer are HIP kernels behind the C ABI (csrc/train.hip:
``m2m_train_forward_backward`` / ``m2m_adafactor_step``); there is no autograd graph and no torch
compute.  The parameters of the ``T5Transformer`` are re-pointed at views of ONE flat fp32 device buffer
(and their ``.grad`` at views of a second one), so ``state_dict()``, checkpoints and the inference path
see the trained weights, and data-parallel training averages the flat gradient (121.6 MB) over RCCL in four large
pieces, the decoder-side ones while the encoder-side backward still runs (``set_sync_stream`` +
``distributed.all_reduce_gradients_overlapped``; ``distributed.all_reduce_gradients`` is the one-call form).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import native
from .input import ModelInputs

_PRECISIONS = {"fp32": native.PREC_FP32, "bf16": native.PREC_BF16, "fp8": native.PREC_FP8}


class NativeTrainer:
    """Owns the native trainer of one ``T5Transformer`` and the two flat buffers."""

    def __init__(self, module, max_batch: int, max_enc_len: int, max_dec_len: int, precision: Optional[str] = None):
        native.require_gpu()
        self.module = module
        self.precision = precision or os.environ.get("M2M_TRAIN_PRECISION", "bf16")
        if self.precision not in _PRECISIONS:
            raise ValueError(f"training precision must be one of {sorted(_PRECISIONS)}, got {self.precision!r}")
        dev = module.transformer.device
        if dev.type != "cuda":
            raise native.NativeError("training runs on the GPU only: call .cuda() first (there is no CPU fallback)")
        self.device = dev
        self.limits = (int(max_batch), int(max_enc_len), int(max_dec_len))
        g = module.geometry
        geom = native.T5GeometryC(g.d_model, g.d_ff, g.num_layers, g.num_decoder_layers, g.num_heads, g.d_kv, g.vocab_size,
                                  g.num_buckets, g.max_distance, g.pad_token_id, g.eos_token_id, g.decoder_start_token_id, g.eps)
        rows = [e.weight.shape[0] for e in module.conditioning.embeds]
        rows_c = (C.c_int * len(rows))(*rows)
        lib = native.load()
        h = C.c_void_p()
        with torch.cuda.device(dev):
            native.check(lib.m2m_trainer_create(C.byref(geom), len(rows), rows_c, _PRECISIONS[self.precision], *self.limits,
                                                C.byref(h)), "m2m_trainer_create")
        self.handle = h
        self.n_floats = int(lib.m2m_trainer_num_params(h))
        self.layout: Dict[str, tuple] = {}
        info = native.TensorInfo()
        for i in range(lib.m2m_trainer_num_tensors(h)):
            native.check(lib.m2m_trainer_tensor_info(h, i, C.byref(info)), "m2m_trainer_tensor_info")
            shape = (info.rows, info.cols) if info.cols else (info.rows,)
            self.layout[info.name.decode()] = (int(info.offset), shape)
        self.params = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_floats, dtype=torch.float32, device=dev)
        self._adopt_parameters()
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.dropout = 0.0
        self.sync_stream = None
        self.early_ranges = []

    def set_dropout(self, p: float, seed: int = 0):
        """Dropout of the teacher-forced pass (hf T5Config.dropout_rate; 0 = off).  Restarts the mask sequence at `seed`."""
        native.check(native.load().m2m_trainer_set_dropout(self.handle, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF), "m2m_trainer_set_dropout")
        self.dropout = float(p)

    # -- parameters <-> flat buffers ---------------------------------------------
    def _adopt_parameters(self):
        own = dict(self.module.named_parameters())           # shared embedding appears once (aliases are deduplicated)
        missing = [k for k in self.layout if k not in own]
        extra = [k for k in own if k not in self.layout]
        if missing or extra:
            raise native.NativeError(f"trainer layout and module parameters differ: missing {missing[:3]}, unexpected {extra[:3]}")
        with torch.no_grad():
            for name, (off, shape) in self.layout.items():
                p = own[name]
                n = p.numel()
                if tuple(p.shape) != tuple(shape):
                    raise native.NativeError(f"{name}: module shape {tuple(p.shape)} != trainer shape {shape}")
                view = self.params[off:off + n].view(shape)
                view.copy_(p.detach().to(self.device, torch.float32))
                p.data = view
                p.grad = self.grads[off:off + n].view(shape)
        self.module._weights_epoch = getattr(self.module, "_weights_epoch", 0) + 1

    # -- data-parallel overlap -----------------------------------------------------
    def set_sync_stream(self, stream: Optional["torch.cuda.Stream"]):
        """Issue the backward pass in two parts and release `stream` as soon as the decoder-side gradients (``early_ranges``: shared
        embedding + lm_head, decoder blocks) are final, so that their all-reduce — enqueued on `stream` right after
        ``forward_backward`` returns — overlaps the encoder-side backward (``distributed.all_reduce_gradients_overlapped``).
        ``None`` switches the split off.  The gradients are bit-identical either way."""
        lib = native.load()
        native.check(lib.m2m_trainer_set_sync_stream(self.handle, C.c_void_p(stream.cuda_stream) if stream is not None else None),
                     "m2m_trainer_set_sync_stream")
        self.sync_stream = stream
        r = (C.c_int64 * 4)()
        native.check(lib.m2m_trainer_early_grad_ranges(self.handle, r), "m2m_trainer_early_grad_ranges")
        self.early_ranges = [(int(r[0]), int(r[1])), (int(r[2]), int(r[3]))]

    def graph_nodes(self) -> int:
        """Launches per step: nodes of the captured graph(s) of the most recent shape (0 before its capture)."""
        return int(native.load().m2m_trainer_graph_nodes(self.handle))

    def fits(self, B: int, S: int, L: int) -> bool:
        return B <= self.limits[0] and S <= self.limits[1] and L <= self.limits[2]

    def close(self):
        if self.handle is not None:
            native.load().m2m_trainer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- one step ------------------------------------------------------------------
    def forward_backward(self, encoder_inputs: torch.Tensor, cond_index: torch.Tensor, labels: torch.Tensor,
                         want_logits: bool = False, backward: bool = True):
        """encoder_inputs [B, S, d] fp32 (log-mel rows in place, cond rows are filled by the trainer),
        cond_index [B, n_cond] int64, labels [B, Ld] int64 with -100 = ignore.  Returns (loss[1] on the device, logits or None);
        the flat gradient buffer (= every parameter's .grad) is overwritten."""
        B, S, _ = encoder_inputs.shape
        Ld = labels.shape[1]
        assert self.fits(B, S, Ld), (B, S, Ld, self.limits)
        x = encoder_inputs.to(self.device, torch.float32).contiguous()
        idx = cond_index.to(self.device, torch.long).contiguous()
        lab = labels.to(self.device, torch.long).contiguous()
        logits = torch.empty((B, Ld, self.module.geometry.vocab_size), dtype=torch.float32, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            native.check(native.load().m2m_train_forward_backward(
                self.handle, self.params.data_ptr(), x.data_ptr(), idx.data_ptr(), lab.data_ptr(), B, S, Ld, self.loss.data_ptr(),
                self.grads.data_ptr() if backward else None, logits.data_ptr() if want_logits else None,
                native.stream_handle(self.device)), "m2m_train_forward_backward")
        return self.loss, logits

    def optimizer_step(self):
        with torch.cuda.device(self.device):
            native.check(native.load().m2m_adafactor_step(self.handle, self.params.data_ptr(), self.grads.data_ptr(),
                                                          native.stream_handle(self.device)), "m2m_adafactor_step")
        self.module._weights_epoch += 1            # the inference path repacks its weights on next use

    @property
    def step_count(self) -> int:
        return int(native.load().m2m_adafactor_get_step(self.handle))

    def optimizer_state(self) -> dict:
        n = int(native.load().m2m_adafactor_state_floats(self.handle))
        buf = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(native.load().m2m_adafactor_state_export(self.handle, buf.data_ptr(), native.stream_handle(self.device)),
                         "m2m_adafactor_state_export")
        return {"step": self.step_count, "second_moments": buf.cpu()}

    def load_optimizer_state(self, state: dict):
        if "second_moments" not in state:                  # torch / Lightning layout (``optimizer.state_dict()`` of HF's Adafactor)
            state = self.optimizer_state_from_hf(state)
        buf = state["second_moments"].to(self.device, torch.float32).contiguous()
        n = int(native.load().m2m_adafactor_state_floats(self.handle))
        if buf.numel() != n:
            raise native.NativeError(f"optimizer state has {buf.numel()} floats, this model's Adafactor state has {n}")
        with torch.cuda.device(self.device):
            native.check(native.load().m2m_adafactor_state_import(self.handle, buf.data_ptr(), int(state["step"]),
                                                                  native.stream_handle(self.device)), "m2m_adafactor_state_import")
            torch.cuda.synchronize(self.device)

    # -- the optimizer state as transformers.optimization.Adafactor keeps it (what a Lightning .ckpt carries) -------------
    # The native state is one flat buffer in tensor order: a matrix owns R[rows] | C[cols] (HF: exp_avg_sq_row / exp_avg_sq_col), a
    # vector V[n] (HF: exp_avg_sq) — csrc/train.hip build_optimizer.  torch numbers an optimizer's parameters in the order of
    # ``module.parameters()``; the module tree registers its parameters in the order of HuggingFace T5's, so index i here is index
    # i of ``Adafactor(self.parameters())`` in ref: music2midi/model.py:28.
    _HF_GROUP = {"lr": None, "eps": (1e-30, 1e-3), "clip_threshold": 1.0, "decay_rate": -0.8, "beta1": None, "weight_decay": 0.0,
                 "scale_parameter": True, "relative_step": True, "warmup_init": True}

    def _state_slices(self):
        """[(name, shape, offset of R or V, offset of C or None)] in the native tensor order."""
        out, off = [], 0
        for name, (_, shape) in self.layout.items():
            if len(shape) == 2:
                out.append((name, shape, off, off + shape[0]))
                off += shape[0] + shape[1]
            else:
                out.append((name, shape, off, None))
                off += shape[0]
        return out

    def _param_order(self):
        return [n for n, _ in self.module.named_parameters()]       # (aliases of the shared embedding appear once, as in torch)

    def optimizer_state_hf(self) -> dict:
        own = self.optimizer_state()
        flat, step = own["second_moments"], own["step"]
        index = {n: i for i, n in enumerate(self._param_order())}
        params = dict(self.module.named_parameters())
        state = {}
        for name, shape, o_r, o_c in self._state_slices():
            if step == 0:
                continue
            rms = float(params[name].detach().float().norm() / params[name].numel() ** 0.5)
            if o_c is not None:
                state[index[name]] = {"step": step, "exp_avg_sq_row": flat[o_r:o_r + shape[0]].clone(),
                                      "exp_avg_sq_col": flat[o_c:o_c + shape[1]].clone(), "RMS": rms}
            else:
                state[index[name]] = {"step": step, "exp_avg_sq": flat[o_r:o_r + shape[0]].clone(), "RMS": rms}
        return {"state": dict(sorted(state.items())), "param_groups": [dict(self._HF_GROUP, params=list(range(len(index))))]}

    def optimizer_state_from_hf(self, hf: dict) -> dict:
        index = {n: i for i, n in enumerate(self._param_order())}
        n = int(native.load().m2m_adafactor_state_floats(self.handle))
        flat = torch.zeros(n, dtype=torch.float32)
        steps = set()
        st = {int(k): v for k, v in hf.get("state", {}).items()}
        for name, shape, o_r, o_c in self._state_slices():
            e = st.get(index[name])
            if e is None:
                continue
            steps.add(int(e["step"]))
            if o_c is not None:
                flat[o_r:o_r + shape[0]] = torch.as_tensor(e["exp_avg_sq_row"], dtype=torch.float32).reshape(-1)
                flat[o_c:o_c + shape[1]] = torch.as_tensor(e["exp_avg_sq_col"], dtype=torch.float32).reshape(-1)
            else:
                flat[o_r:o_r + shape[0]] = torch.as_tensor(e["exp_avg_sq"], dtype=torch.float32).reshape(-1)
        if len(steps) > 1:
            raise native.NativeError(f"optimizer state with different step counts per parameter {sorted(steps)[:4]}: one step counter is kept")
        return {"step": steps.pop() if steps else 0, "second_moments": flat}


class Adafactor:
    """The object ``configure_optimizers`` hands out: ``step()`` = one native Adafactor(warmup_init=True) update of every
    parameter from its ``.grad`` (ref: music2midi/model.py:27-30)."""

    def __init__(self, owner):
        self._owner = owner

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._owner._native_trainer().optimizer_step()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        tr = self._owner._trainer
        if tr is not None:
            tr.grads.zero_()

    def state_dict(self):
        return self._owner._native_trainer().optimizer_state()

    def load_state_dict(self, state):
        self._owner._native_trainer().load_optimizer_state(state)


class AdafactorSchedule:
    """transformers.optimization.AdafactorSchedule stand-in: the learning rate is internal to Adafactor (relative
    step); ``get_last_lr`` reports rho_t = min(1e-6 t, 1/sqrt(t)) — the factor every tensor's max(1e-3, rms(p)) is scaled by."""

    def __init__(self, optimizer: Adafactor):
        self.optimizer = optimizer

    def step(self):
        pass

    def get_last_lr(self):
        t = max(1, self.optimizer._owner._native_trainer().step_count)
        return [min(1e-6 * t, t ** -0.5)]
