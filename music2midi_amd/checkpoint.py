"""Weights in / out without pytorch-lightning.

A reference checkpoint is a Lightning ``.ckpt``: a ``torch.save``d dict whose
``state_dict`` holds ``model.transformer.*`` (HF T5 names, separate
``lm_head``), ``model.conditioning.embeds.{i}.weight`` and the two torchaudio
buffers under ``model.spectrogram.melspectrogram.*`` (SURVEY.md §3.4;
ref: music2midi/model.py:21-25, evaluate.py:27).  The pickles may reference
Lightning classes (``AttributeDict``, callbacks); the reader resolves an allow-list of tensor-related
globals only and stubs everything else, so an untrusted file cannot execute code.
"""
from __future__ import annotations

import logging
import pickle
from typing import Dict, Mapping

import numpy as np
import torch

_log = logging.getLogger(__name__)


def to_torch_state(sd: Mapping[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def load_t5_state(t5_transformer, sd: Mapping, strict: bool = True) -> None:
    """Load a ``transformer.* / conditioning.* / spectrogram.*`` state dict into a T5Transformer.

    HF alias keys (``encoder.embed_tokens.weight``/``decoder.embed_tokens.weight``) are optional.
    With ``strict=False`` keys the module does not have are dropped and keys the dict does
    not have keep their current values.
    """
    sd = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}
    own = t5_transformer.state_dict()
    for alias in ("transformer.encoder.embed_tokens.weight", "transformer.decoder.embed_tokens.weight"):
        if alias not in sd and "transformer.shared.weight" in sd:
            sd[alias] = sd["transformer.shared.weight"]
    if not strict:
        sd = {k: v for k, v in sd.items() if k in own}
        for k in own:
            if k not in sd:
                sd[k] = own[k]
    t5_transformer.load_state_dict(sd, strict=True)


class _Stub:
    """Inert placeholder for anything a checkpoint's pickle names outside the allow-list below
    (Lightning callbacks, loggers, pathlib paths, omegaconf nodes ...): it accepts any construction
    arguments and any state, and does nothing with them."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)

    def __setitem__(self, key, value):      # pickled mappings/sequences fill themselves through these
        self.__dict__[key] = value

    def append(self, item):
        self.__dict__.setdefault("_items", []).append(item)

    def extend(self, items):
        for it in items:
            self.append(it)


_BUILTINS_OK = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "bytearray",
                "complex", "slice", "range", "object"}
_TORCH_UTILS_OK = {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state",
                   "_rebuild_qtensor"}
_NUMPY_OK = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
             ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
             ("numpy", "ndarray"), ("numpy", "dtype")}
# classes that are a plain mapping in disguise: the hyper-parameter container of a LightningModule
# (ref: music2midi/model.py:25 save_hyperparameters() -> AttributeDict in the pickle)
_DICT_LIKE = {"AttributeDict"}


def _resolve_global(module: str, name: str):
    """Allow-list lookup.  Returns the real object for what a tensor checkpoint legitimately needs and an
    inert stub class for EVERYTHING else — no importable global (os.system, builtins.eval, ...) is ever
    resolved, so loading an untrusted ``.ckpt`` cannot run code."""
    if (module, name) == ("collections", "OrderedDict"):
        import collections
        return collections.OrderedDict
    if module == "builtins" and name in _BUILTINS_OK:
        import builtins
        return getattr(builtins, name)
    if module == "torch._utils" and name in _TORCH_UTILS_OK:
        import torch._utils
        return getattr(torch._utils, name)
    if module == "torch" and hasattr(torch, name):
        obj = getattr(torch, name)
        if isinstance(obj, torch.dtype) or obj in (torch.Size, torch.device):
            return obj
        if isinstance(obj, type) and name.endswith("Storage"):
            return obj
    if module == "torch.storage" and name in ("UntypedStorage", "TypedStorage"):
        import torch.storage
        return getattr(torch.storage, name)
    if module == "torch.nn.parameter" and name in ("Parameter", "Buffer"):
        import torch.nn.parameter
        return getattr(torch.nn.parameter, name, _Stub)
    if (module, name) in _NUMPY_OK:
        import importlib
        return getattr(importlib.import_module(module), name)
    if name in _DICT_LIKE:
        return dict
    return type(name, (_Stub,), {"__module__": module})


class _AllowListUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        return _resolve_global(module, name)


class _AllowListPickle:
    """`pickle_module` for torch.load: only the Unpickler differs from the stdlib module."""
    Unpickler = _AllowListUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _AllowListUnpickler(f, **kw).load()


def read_checkpoint(path) -> dict:
    """torch.load a Lightning-format checkpoint on CPU without Lightning.

    First with torch's own ``weights_only=True`` loader.  Real Lightning checkpoints name classes that
    loader rejects (``AttributeDict``, callbacks, ``pathlib`` paths); only for that specific failure
    (``pickle.UnpicklingError``) the file is read again through an allow-list unpickler that maps every
    global outside the allow-list to an inert stub — never through the unrestricted pickle.  I/O errors
    (missing or truncated file) are not retried and propagate."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        _log.info("checkpoint %s names globals outside torch's weights-only allow-list (%s); "
                  "reading it with the stubbing allow-list unpickler", path, str(e).splitlines()[0][:160])
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_AllowListPickle)


def write_checkpoint(path, module_state: Mapping[str, torch.Tensor], config_path: str = "config.yaml") -> None:
    """Emit the Lightning layout the reference's load_from_checkpoint expects."""
    torch.save({
        "state_dict": {k: v.detach().cpu() for k, v in module_state.items()},
        "hyper_parameters": {"config_path": str(config_path)},
        "pytorch-lightning_version": "2.1.0",
        "epoch": 0,
        "global_step": 0,
    }, path)
