"""ctypes binding of libmusic2midi_amd.so (the C ABI in include/music2midi_amd.h).

There is NO fallback: if the library is missing or a call fails, this module
raises.  The product never routes through ``oracle/`` or a torch CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path
from typing import Optional

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libmusic2midi_amd.so"
_lib: Optional[C.CDLL] = None
_lock = threading.Lock()

PREC_FP32 = 0
PREC_BF16 = 1
PREC_FP8 = 2
KERNEL_DEC_CROSS_ATTN = 0
KERNEL_DEC_SELF_ATTN = 1
KERNEL_DEC_STEP = 2


class NativeError(RuntimeError):
    pass


class FrontendDesc(C.Structure):
    _fields_ = [("n_fft", C.c_int), ("hop_length", C.c_int), ("n_freqs", C.c_int), ("n_mels", C.c_int),
                ("window_host", C.c_void_p), ("fb_host", C.c_void_p)]


class T5GeometryC(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("d_model", "d_ff", "num_layers", "num_decoder_layers", "num_heads", "d_kv", "vocab_size",
                 "num_buckets", "max_distance", "pad_token_id", "eos_token_id", "decoder_start_token_id")] + \
               [("layer_norm_eps", C.c_float)]


class EncLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln0", "q", "k", "v", "o", "ln1", "wi0", "wi1", "wo")]


class DecLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("ln0", "q", "k", "v", "o", "ln1", "cq", "ck", "cv", "co", "ln2", "wi0", "wi1", "wo")]


class T5Weights(C.Structure):
    _fields_ = [("shared", C.c_void_p), ("lm_head", C.c_void_p), ("enc_rel_bias", C.c_void_p),
                ("dec_rel_bias", C.c_void_p), ("enc_final_ln", C.c_void_p), ("dec_final_ln", C.c_void_p),
                ("enc", C.POINTER(EncLayerWeights)), ("dec", C.POINTER(DecLayerWeights))]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 160), ("offset", C.c_int64), ("rows", C.c_int), ("cols", C.c_int)]


# name -> (restype, argtypes); every symbol include/music2midi_amd.h declares.
_SIGNATURES = {
    "m2m_abi_version": (C.c_int, []),
    "m2m_last_error": (C.c_char_p, []),
    "m2m_device_count": (C.c_int, []),
    "m2m_frontend_create": (C.c_int, [C.POINTER(FrontendDesc), C.POINTER(C.c_void_p)]),
    "m2m_frontend_destroy": (None, [C.c_void_p]),
    "m2m_frontend_num_frames": (C.c_int, [C.c_void_p, C.c_int]),
    "m2m_frontend_fb_nnz": (C.c_int, [C.c_void_p]),
    "m2m_logmel_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                 C.c_void_p]),
    "m2m_cond_rows_f32": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "m2m_model_create": (C.c_int, [C.POINTER(T5GeometryC), C.POINTER(T5Weights), C.c_int, C.c_void_p,
                                   C.POINTER(C.c_void_p)]),
    "m2m_model_destroy": (None, [C.c_void_p]),
    "m2m_model_precision": (C.c_int, [C.c_void_p]),
    "m2m_model_param_bytes": (C.c_int64, [C.c_void_p]),
    "m2m_model_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "m2m_rel_bucket": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "m2m_session_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "m2m_session_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                     C.POINTER(C.c_void_p)]),
    "m2m_session_destroy": (None, [C.c_void_p]),
    "m2m_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "m2m_generate_greedy": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "m2m_session_repack_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "m2m_decode_forced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "m2m_trainer_create": (C.c_int, [C.POINTER(T5GeometryC), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p)]),
    "m2m_trainer_destroy": (None, [C.c_void_p]),
    "m2m_trainer_num_params": (C.c_int64, [C.c_void_p]),
    "m2m_trainer_num_tensors": (C.c_int, [C.c_void_p]),
    "m2m_trainer_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TensorInfo)]),
    "m2m_trainer_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "m2m_train_forward_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m2m_trainer_set_dropout": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64]),
    "m2m_trainer_set_sync_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "m2m_trainer_early_grad_ranges": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "m2m_trainer_graph_nodes": (C.c_int, [C.c_void_p]),
    "m2m_adafactor_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m2m_adafactor_get_step": (C.c_int, [C.c_void_p]),
    "m2m_adafactor_state_floats": (C.c_int64, [C.c_void_p]),
    "m2m_adafactor_state_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "m2m_adafactor_state_import": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "m2m_mx8_matmul_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "m2m_mx8_matmul_bf16a": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "m2m_attn_head_fwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m2m_attn_head_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m2m_bench_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                   C.POINTER(C.c_int64), C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> Path:
    return Path(os.environ.get("M2M_LIBRARY", str(_LIB_PATH)))


def load() -> C.CDLL:
    """Load the shared library once; raise NativeError if it is not there."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not path.exists():
            raise NativeError(
                f"{path} not found: the HIP library has not been built. Run "
                "`python -m music2midi_amd.csrc.build` (needs hipcc). There is no CPU fallback.")
        try:
            lib = C.CDLL(str(path))
        except OSError as e:
            raise NativeError(f"cannot load {path}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise NativeError(f"{path} does not export {name} (stale build?)") from e
            fn.restype = res
            fn.argtypes = args
        if lib.m2m_abi_version() != 1:
            raise NativeError(f"ABI version mismatch: library {lib.m2m_abi_version()}, binding 1")
        _lib = lib
        return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().m2m_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what} failed (status {status}): {msg}")


def stream_handle(device=None) -> int:
    """hipStream_t of torch's current stream on `device` as an integer."""
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)


def require_gpu() -> None:
    import torch
    if not torch.cuda.is_available() or load().m2m_device_count() < 1:
        raise NativeError("no HIP device visible: the Music2MIDI MI355X path has no CPU fallback")
