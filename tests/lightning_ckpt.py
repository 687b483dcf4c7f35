"""Test helper: write a checkpoint shaped like the ones pytorch-lightning 2.1 writes for the reference's
``Music2MIDI`` LightningModule (ref: music2midi/model.py:20-25, train.py:40-41) WITHOUT Lightning.

The pickle must NAME Lightning's classes (``pytorch_lightning.utilities.parsing.AttributeDict`` for the
hyper-parameters that ``save_hyperparameters()`` records, a ``ModelCheckpoint`` callback object, a
``pathlib`` path) the way a real file does; stub modules carrying classes of those qualified names are
installed only while ``torch.save`` runs and removed again, so the reader never sees them."""
import pathlib
import sys
import types

import torch

_MODULES = ("pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing",
            "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint")


def write_lightning_like_ckpt(path, module_state, config_path="config.yaml"):
    mods = {n: types.ModuleType(n) for n in _MODULES}

    class AttributeDict(dict):
        pass

    class ModelCheckpoint:
        def __init__(self):
            self.dirpath = pathlib.PurePosixPath("music2midi/abc123/checkpoints")
            self.best_model_score = torch.tensor(0.4321)
            self.monitor = "val/score"

    AttributeDict.__module__ = "pytorch_lightning.utilities.parsing"
    AttributeDict.__qualname__ = "AttributeDict"
    ModelCheckpoint.__module__ = "pytorch_lightning.callbacks.model_checkpoint"
    ModelCheckpoint.__qualname__ = "ModelCheckpoint"
    mods["pytorch_lightning.utilities.parsing"].AttributeDict = AttributeDict
    mods["pytorch_lightning.callbacks.model_checkpoint"].ModelCheckpoint = ModelCheckpoint
    saved = {n: sys.modules.get(n) for n in _MODULES}
    sys.modules.update(mods)
    try:
        torch.save({
            "epoch": 412, "global_step": 51912, "pytorch-lightning_version": "2.1.0",
            "state_dict": {k: v.detach().cpu().clone() for k, v in module_state.items()},
            "loops": {"fit_loop": {"state_dict": {}, "epoch_progress": {"total": {"ready": 413, "completed": 412}}}},
            "callbacks": {"ModelCheckpoint{'monitor': 'val/score', 'mode': 'max'}": {
                "monitor": "val/score", "best_model_score": torch.tensor(0.4321),
                "best_model_path": "music2midi/abc123/checkpoints/epoch=412-step=51912.ckpt",
                "dirpath": pathlib.PurePosixPath("music2midi/abc123/checkpoints")}},
            "callback_objects": [ModelCheckpoint()],
            "optimizer_states": [{"state": {0: {"step": 51912, "exp_avg_sq_row": torch.zeros(4), "exp_avg_sq_col": torch.zeros(3),
                                                "RMS": 0.031}}, "param_groups": [{"lr": None, "params": [0]}]}],
            "lr_schedulers": [{"base_lrs": [0.0], "last_epoch": 51912}],
            "hparams_name": "kwargs",
            "hyper_parameters": AttributeDict(config_path=str(config_path)),
        }, path)
    finally:
        for n in _MODULES:
            if saved[n] is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = saved[n]
    return path
