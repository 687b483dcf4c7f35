"""YAML config with attribute access.

The reference loads ``config.yaml`` with OmegaConf independently in every class
(ref: music2midi/model.py:23, music2midi/transformer.py:13) and then uses both
attribute access (``config.model.sample_rate``) and mapping access
(``**config.model.t5``, ``config.conditioning.values()``).  OmegaConf is not a
dependency here; this module gives the same two access styles on top of PyYAML.
"""
from __future__ import annotations

import os
from collections.abc import Mapping
from pathlib import Path
from typing import Any, Union

import yaml


class ConfigNode(dict):
    """dict that also answers attribute access, recursively.

    It subclasses ``dict`` so ``**node``, ``node.values()``, ``len(node)`` and
    ``isinstance(node, Mapping)`` behave exactly as a plain mapping would —
    which is all the reference relies on from ``DictConfig``.
    """

    def __init__(self, data: Mapping | None = None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError as e:  # match OmegaConf: missing key is an attribute error
            raise AttributeError(f"Missing key {name}") from e

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = _wrap(value)

    def to_dict(self) -> dict:
        return {k: _unwrap(v) for k, v in self.items()}


class ConfigList(list):
    def __init__(self, data=()):
        super().__init__(_wrap(v) for v in data)


def _wrap(v: Any) -> Any:
    if isinstance(v, ConfigNode) or isinstance(v, ConfigList):
        return v
    if isinstance(v, Mapping):
        return ConfigNode(v)
    if isinstance(v, (list, tuple)):
        return ConfigList(v)
    return v


def _unwrap(v: Any) -> Any:
    if isinstance(v, ConfigNode):
        return v.to_dict()
    if isinstance(v, ConfigList):
        return [_unwrap(x) for x in v]
    return v


def load_config(path_or_mapping: Union[str, os.PathLike, Mapping]) -> ConfigNode:
    """Load a reference-schema config (ref: config.yaml:1-50).

    Accepts a path (what the reference passes as ``config_path``) or an already
    parsed mapping (convenient for tests and for the tiny parity configs).
    """
    if isinstance(path_or_mapping, Mapping):
        return ConfigNode(path_or_mapping)
    with open(Path(path_or_mapping), "r") as f:
        return ConfigNode(yaml.safe_load(f))


# The reference's shipped configuration (ref: config.yaml:1-50), restated as data
# so the package works without a config file on disk (bench / smoke / tests).
DEFAULT_CONFIG: dict = {
    "dataset": {
        "sample_rate": 22050,
        "dtw_feature_rate": 50,
        "segment_duration": 3,
        "max_notes_per_second": 30,
        "filter_threshold": {
            "wp_std": 5,
            "max_beat_fluctuation": 1.2,
            "max_note_density": 25,
            "time_diff_ratio": 0.2,
        },
    },
    "spectrogram": {"n_fft": 2048, "hop_length": 256, "f_min": 20.0},
    "model": {
        "sample_rate": 16000,
        "t5": {
            "num_layers": 6,
            "num_decoder_layers": 6,
            "d_model": 384,
            "d_ff": 1152,
            "feed_forward_proj": "gated-gelu",
            "tie_word_embeddings": False,
            "tie_encoder_decoder": False,
            "vocab_size": 400,
            "n_positions": 1024,
            "relative_attention_num_buckets": 32,
            "pad_token_id": 0,
            "bos_token_id": 1,
            "eos_token_id": 2,
            "decoder_start_token_id": 1,
        },
    },
    "tokenizer": {
        "midi_quantize_ms": 50,
        "vocab_size": {"special": 5, "pitch": 128, "time": 200},
        "default_velocity": 80,
    },
    "trainer": {"max_epochs": 800, "accumulate_grad_batches": 1, "log_every_n_steps": 40},
    "dataloader": {"batch_size": 16, "num_workers": 4},
    "inference": {"batch_size": 128},
    "conditioning": {
        "genre": ["electronic", "pop", "rock", "soundtrack", "world_music", "classical"],
        "difficulty": ["beginner", "intermediate", "advanced"],
    },
}


def default_config() -> ConfigNode:
    return ConfigNode(DEFAULT_CONFIG)


# T5 hyper-parameters that the reference leaves to the HF defaults
# (hf: models/t5/configuration_t5.py:44-62): they are not in config.yaml.
T5_DEFAULTS = {
    "num_heads": 8,
    "d_kv": 64,
    "relative_attention_num_buckets": 32,
    "relative_attention_max_distance": 128,
    "layer_norm_epsilon": 1e-6,
    "feed_forward_proj": "relu",
    "pad_token_id": 0,
    "eos_token_id": 1,
    "decoder_start_token_id": None,
    "vocab_size": 32128,
    "d_model": 512,
    "d_ff": 2048,
    "num_layers": 6,
    "num_decoder_layers": None,
}


class T5Geometry:
    """Resolved T5 shape parameters (what ``T5Config(**config.model.t5)`` yields)."""

    def __init__(self, t5: Mapping):
        g = dict(T5_DEFAULTS)
        g.update(dict(t5))
        self.d_model = int(g["d_model"])
        self.d_ff = int(g["d_ff"])
        self.num_layers = int(g["num_layers"])
        ndl = g.get("num_decoder_layers")
        self.num_decoder_layers = int(ndl) if ndl is not None else self.num_layers
        self.num_heads = int(g["num_heads"])
        self.d_kv = int(g["d_kv"])
        self.inner_dim = self.num_heads * self.d_kv
        self.vocab_size = int(g["vocab_size"])
        self.num_buckets = int(g["relative_attention_num_buckets"])
        self.max_distance = int(g["relative_attention_max_distance"])
        self.eps = float(g["layer_norm_epsilon"])
        self.pad_token_id = int(g["pad_token_id"])
        self.eos_token_id = int(g["eos_token_id"])
        dst = g.get("decoder_start_token_id")
        self.decoder_start_token_id = int(dst) if dst is not None else self.pad_token_id
        self.bos_token_id = g.get("bos_token_id")
        ffp = str(g["feed_forward_proj"])
        if ffp not in ("gated-gelu",):
            # The reference only ever configures gated-gelu (ref: config.yaml:22);
            # the device path implements exactly that FFN.
            raise ValueError(
                f"feed_forward_proj={ffp!r} is not supported by the MI355X path "
                "(reference uses 'gated-gelu')"
            )
        self.feed_forward_proj = ffp

    def as_dict(self) -> dict:
        return dict(self.__dict__)
