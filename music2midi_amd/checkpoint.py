"""Weights in / out without pytorch-lightning.

A reference checkpoint is a Lightning ``.ckpt``: a ``torch.save``d dict whose
``state_dict`` holds ``model.transformer.*`` (HF T5 names, separate
``lm_head``), ``model.conditioning.embeds.{i}.weight`` and the two torchaudio
buffers under ``model.spectrogram.melspectrogram.*`` (SURVEY.md §3.4;
ref: music2midi/model.py:21-25, evaluate.py:27).  The pickles may reference
Lightning classes (``AttributeDict``); unknown globals are stubbed on load.
"""
from __future__ import annotations

import pickle
from typing import Dict, Mapping

import numpy as np
import torch


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
    """Placeholder for classes a Lightning pickle names but this image lacks."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if name == "AttributeDict":
                return dict
            return type(name, (_Stub,), {})


class _TolerantPickle:
    Unpickler = _TolerantUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def read_checkpoint(path) -> dict:
    """torch.load a Lightning-format checkpoint on CPU, tolerating missing Lightning classes."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)


def write_checkpoint(path, module_state: Mapping[str, torch.Tensor], config_path: str = "config.yaml") -> None:
    """Emit the Lightning layout the reference's load_from_checkpoint expects."""
    torch.save({
        "state_dict": {k: v.detach().cpu() for k, v in module_state.items()},
        "hyper_parameters": {"config_path": str(config_path)},
        "pytorch-lightning_version": "2.1.0",
        "epoch": 0,
        "global_step": 0,
    }, path)
