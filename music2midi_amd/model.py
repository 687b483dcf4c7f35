"""Music2MIDI — the task wrapper callers use (ref: music2midi/model.py:20-140),
without pytorch-lightning: same constructor, ``load_from_checkpoint``,
``generate(audio_path|audio_y, sr, cond_index)``, ``sample_tokens`` and
``evaluate_batch``; segmentation, zero padding, chunking by
``inference.batch_size``, conditioning broadcast, ``max_length=1024`` and the
sequential token decode follow the reference line by line in behaviour.
``training_step`` / ``configure_optimizers`` run the native forward+backward and Adafactor
(csrc/train.hip) instead of autograd + Lightning.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import distributed as D
from .audio import load_audio as _load_audio
from .checkpoint import load_t5_state, read_checkpoint
from .config import load_config
from .evaluation import evaluate_batch
from .input import ModelInputs
from .transformer import T5Transformer
from .utils import numpy_to_midi


class Music2MIDI(nn.Module):
    def __init__(self, config_path: str, precision: Optional[str] = None):
        super().__init__()
        self.config = load_config(config_path)
        self.model = T5Transformer(config_path, precision=precision)
        self.hparams = {"config_path": config_path}

    # -- checkpoint ----------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config_path: Optional[str] = None, map_location=None,
                             strict: bool = True, **kwargs) -> "Music2MIDI":
        """Lightning-format ``.ckpt`` -> model (ref: evaluate.py:27, webui.py:90)."""
        ckpt = read_checkpoint(checkpoint_path)
        if config_path is None:
            config_path = (ckpt.get("hyper_parameters") or {}).get("config_path", "config.yaml")
        obj = cls(config_path, **kwargs)
        state = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        prefix = "model."
        inner = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
        load_t5_state(obj.model, inner if inner else state, strict=strict)
        return obj

    @property
    def device(self) -> torch.device:
        return self.model.transformer.device

    # -- training surface (ref model.py:27-43) -----------------
    _trainer = None
    global_step = 0

    def _native_trainer(self, B: int = 0, S: int = 0, L: int = 0):
        """The native trainer, (re)built when a batch exceeds the sizes it was created for."""
        from .training import NativeTrainer
        tr = self._trainer
        if tr is None or not tr.fits(B, S, L):
            old = tr.limits if tr is not None else (0, 0, 0)
            state = tr.optimizer_state() if tr is not None and tr.step_count > 0 else None
            if tr is not None:
                tr.close()
            limits = (max(B, old[0], int(self.config.dataloader.batch_size)), max(S, old[1]), max(L, old[2], 64))
            self._trainer = NativeTrainer(self.model, *limits, precision=getattr(self, "train_precision", None))
            if state is None and self._resume_optimizer_state is not None:      # a checkpoint read before the first batch
                state, self._resume_optimizer_state = self._resume_optimizer_state, None
            if state is not None:
                self._trainer.load_optimizer_state(state)
            if D.dist.is_available() and D.dist.is_initialized() and D.dist.get_world_size() > 1 and self._trainer.device.type == "cuda":
                # every pass of this trainer releases the decoder-side gradients half-way (distributed.all_reduce_gradients_overlapped)
                self._trainer.set_sync_stream(torch.cuda.Stream(device=self._trainer.device))
        # dropout as the reference trains: model.train() (ref train.py:33) activates T5Config.dropout_rate (0.1 unless the
        # config says otherwise); .eval() switches it off
        want = float(self.config.model.t5.get("dropout_rate", 0.1)) if self.training else 0.0
        if self._trainer.dropout != want:
            self._trainer.set_dropout(want, seed=int(getattr(self, "seed", 0)) + self.global_step)
        return self._trainer

    _resume_optimizer_state = None

    def configure_optimizers(self):
        """ref model.py:27-30: Adafactor(self.parameters(), warmup_init=True) + AdafactorSchedule."""
        from .training import Adafactor, AdafactorSchedule
        optimizer = Adafactor(self)
        return [optimizer], [AdafactorSchedule(optimizer)]

    # Labels are padded to the longest sequence of the batch (ref: music2midi/tokenizer.py:86-96), so the decoder length changes
    # from batch to batch.  Rounded up to a multiple of LABEL_BUCKET with ignored (-100) positions the loss and every gradient are
    # unchanged — an ignored position contributes nothing and, being causal, is seen by no scored one — while the trainer meets a
    # handful of shapes instead of one per batch and replays their captured graphs (csrc/train.hip GraphSlot).
    LABEL_BUCKET = 16

    def _labels(self, notes_batch) -> torch.Tensor:
        t5 = self.model
        labels = t5.tokenizer(notes_batch)
        labels[labels == t5.geometry.pad_token_id] = -100
        pad = -labels.shape[1] % self.LABEL_BUCKET
        if pad:
            labels = torch.nn.functional.pad(labels, (0, pad), value=-100)
        return labels

    def training_step(self, inputs: ModelInputs, batch_idx):
        """ref model.py:32-43.  Lightning calls backward() on the returned loss; here forward AND backward have
        already been ENQUEUED when this returns: every parameter's ``.grad`` will hold d loss / d parameter (views of one flat
        buffer), ready for ``distributed.all_reduce_gradients`` and ``optimizer.step()``.  Nothing here waits for the device: the
        returned loss and ``self.logged["train/loss"]`` are 0-dim tensors on the GPU (``logged_metrics()`` reads them), so the
        gradient all-reduce the caller enqueues next overlaps the backward pass that is still running."""
        t5 = self.model
        labels = self._labels(inputs.notes_batch)
        x = t5.encoder_inputs(inputs)
        tr = self._native_trainer(x.shape[0], x.shape[1], labels.shape[1])
        loss, _ = tr.forward_backward(x, inputs.cond_index, labels)
        loss = loss[0].clone()                                # the trainer's loss word is rewritten by the next pass
        self.logged = {"train/loss": loss, "batch_size": int(x.shape[0])}
        if (self.global_step + 1) % int(self.config.trainer.log_every_n_steps) == 0:
            self.logged["train/score"] = float(self.evaluate_batch(inputs)[0])
        return loss

    def logged_metrics(self) -> dict:
        """The values of the last step as Lightning would log them: ``self.log(..., sync_dist=True)`` in the reference
        (ref model.py:37,42,49,52) means the MEAN over the data-parallel ranks — one packed all-reduce
        (``distributed.reduce_logged``); this is also where device-resident values are read."""
        return D.reduce_logged(dict(getattr(self, "logged", {}) or {}), device=self.device)

    # -- checkpoints of a training run (ref train.py:41: trainer.fit(..., ckpt_path=args.ckpt)) ------------------------------
    def save_checkpoint(self, path, collective: bool = True) -> None:
        """A Lightning-layout ``.ckpt`` of the run: ``state_dict`` (``model.*`` keys incl. the torchaudio buffers), the optimizer
        state as ``transformers.optimization.Adafactor.state_dict()`` lays it out (``optimizer_states[0]``), ``global_step``,
        ``hyper_parameters`` — what ``load_from_checkpoint`` and ``fit_batches(ckpt_path=)`` (and the reference's own
        ``trainer.fit(ckpt_path=)``) read.  Under data parallelism only global rank 0 writes (as Lightning does), into a temporary
        file that is renamed over ``path`` once complete — a reader never sees a torn file — and every rank leaves through a
        broadcast of rank 0's outcome: a ``resume_from_checkpoint`` that follows on any rank reads the finished file, and when
        rank 0 FAILED (disk full, permissions) every rank raises instead of walking into the next gradient all-reduce without it.
        Like ``Trainer.save_checkpoint`` this is therefore a collective call: every rank must make it.  A caller that guards it
        itself (``if rank == 0: model.save_checkpoint(p, collective=False)``) gets the plain single-process behaviour."""
        failure = None
        if D.is_rank_zero() or not collective:
            try:
                tr = self._trainer
                opt = tr.optimizer_state_hf() if tr is not None else {"state": {}, "param_groups": []}     # reads the device: rank 0 only
                torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
                path = Path(path)
                tmp = path.with_name(f".{path.name}.tmp{os.getpid()}")
                try:
                    torch.save({
                        "epoch": int(getattr(self, "current_epoch", 0)), "global_step": int(self.global_step), "pytorch-lightning_version": "2.1.0",
                        "state_dict": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
                        "callbacks": {}, "optimizer_states": [opt],
                        "lr_schedulers": [{"base_lrs": [0.0], "last_epoch": int(self.global_step), "_step_count": int(self.global_step) + 1}],
                        "hparams_name": "kwargs", "hyper_parameters": dict(self.hparams),
                    }, tmp)
                    os.replace(tmp, path)
                finally:
                    if tmp.exists():
                        tmp.unlink()
            except Exception as e:           # the other ranks are on their way to the barrier: meet them there before raising
                failure = e
        if collective:
            failed = D.broadcast_flag(1 if failure is not None else 0, self.device)
            if failed and failure is None:
                raise RuntimeError(f"save_checkpoint: global rank 0 failed to write {path} (see its error); nothing was saved")
        if failure is not None:
            raise failure

    def resume_from_checkpoint(self, path) -> None:
        """Weights, Adafactor state and step counter of a ``.ckpt`` written by ``save_checkpoint`` or by Lightning."""
        ckpt = read_checkpoint(path)
        state = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        inner = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
        tr = self._trainer
        if tr is not None:                                   # the parameters are views of the trainer's flat buffer: copy in place
            own = dict(self.model.named_parameters())
            with torch.no_grad():
                for k, v in (inner or state).items():
                    if k in own:
                        own[k].copy_(torch.as_tensor(v).to(own[k].device, own[k].dtype))
            self.model._weights_epoch = getattr(self.model, "_weights_epoch", 0) + 1
        else:
            load_t5_state(self.model, inner if inner else state, strict=False)
        self.global_step = int(ckpt.get("global_step", 0))
        opts = ckpt.get("optimizer_states") or []
        opt = opts[0] if opts else None
        if opt is not None and opt.get("state"):
            if tr is not None:
                tr.load_optimizer_state(opt)
            else:
                self._resume_optimizer_state = opt
        if tr is not None:
            tr.dropout = -1.0                                # the mask sequence restarts from the restored step (see _native_trainer)

    def fit_batches(self, batches, optimizer=None, world_size: int = 1, ckpt_path=None, save_path=None, save_every_n_steps: int = 0):
        """Minimal stand-in for ``pl.Trainer.fit`` (ref train.py:40-41): step over an iterable of ModelInputs.  ``ckpt_path``
        resumes a run (weights + Adafactor state + step counter) as ``trainer.fit(..., ckpt_path=)`` does; ``save_path`` is
        (re)written every ``save_every_n_steps`` steps and at the end.  The host never waits for a step: losses stay on the device
        until the loop is over, metrics are reduced over the ranks and read every ``trainer.log_every_n_steps`` steps
        (``self.log_history``)."""
        if ckpt_path is not None:
            self.resume_from_checkpoint(ckpt_path)
        if optimizer is None:
            optimizer = self.configure_optimizers()[0][0]
        log_every = max(1, int(self.config.trainer.log_every_n_steps))
        self.log_history = getattr(self, "log_history", [])
        losses, pending = [], []          # host floats so far / device scalars of the steps since the last read
        for i, batch in enumerate(batches):
            loss = self.training_step(batch, i)
            tr = self._trainer
            if tr.sync_stream is not None:       # data-parallel (set when the trainer was built): decoder-side pieces overlap the encoder backward
                D.all_reduce_gradients_overlapped(tr.grads, tr.early_ranges, tr.sync_stream)
            else:
                D.all_reduce_gradients(tr.grads)
            optimizer.step()
            self.global_step += 1
            pending.append(loss)
            if self.global_step % log_every == 0:
                self.log_history.append(dict(self.logged_metrics(), step=self.global_step))
                # logged_metrics() has just waited for this step: move the window's losses to the host here, so that a long run
                # (119 200 steps in the reference's) holds at most log_every device scalars, not one per step
                losses.extend(float(v) for v in torch.stack(pending).cpu())
                pending = []
            if save_path is not None and save_every_n_steps and self.global_step % save_every_n_steps == 0:
                self.save_checkpoint(save_path)
        if save_path is not None:
            self.save_checkpoint(save_path)
        if pending:
            losses.extend(float(v) for v in torch.stack(pending).cpu())
        return losses

    def validation_step(self, inputs: ModelInputs, batch_idx):
        """ref model.py:45-54: teacher-forced loss + chroma score of a greedy decode; returns the LOSS (as the
        reference does).  Lightning's ``self.log`` does not exist here: the two values are kept in
        ``self.logged`` under the reference's metric names, already reduced over the ranks (``sync_dist=True``)."""
        loss = self.model(inputs).loss
        score = self.evaluate_batch(inputs)[0]
        self.logged = D.reduce_logged({"val/loss": loss, "val/score": float(score), "batch_size": int(inputs.input_waveform.shape[0])},
                                      device=self.device)
        self.logged["batch_size"] = int(self.logged["batch_size"])
        return loss

    # -- inference -----------------------------------------------------------
    @torch.no_grad()
    def evaluate_batch(self, inputs: ModelInputs):
        """(chroma accuracy, decoded MIDI per clip, label MIDI per clip) for one labelled batch — ref
        model.py:55-65.  The decode budget is four tokens per label note of the busiest clip."""
        budget = 4 * max(len(n) for n in inputs.notes_batch)
        token_ids = self.model.generate(inputs, max_length=budget)
        predicted = [numpy_to_midi(n) for n in self.model.tokenizer.decode(token_ids, mode="batched")]
        wanted = [numpy_to_midi(n) for n in inputs.notes_batch]
        return evaluate_batch(wanted, predicted), predicted, wanted

    def generate(self, audio_path: Optional[Union[str, Path]] = None, audio_y: Optional[np.ndarray] = None,
                 sr: Optional[int] = None, cond_index: Optional[list] = None):
        """Specify either audio_path or audio_y as input; returns a MIDI object."""
        return numpy_to_midi(self.generate_notes(audio_path, audio_y, sr, cond_index))

    # The three steps below are what ref model.py:67-140 does inline: validate/load, cut the song into
    # whole segments (zero-padded tail), decode the segments in chunks and stitch the notes together.
    def _resolve_audio(self, audio_path, audio_y, sr) -> np.ndarray:
        model_sr = self.config.model.sample_rate
        if audio_path is None and audio_y is None:
            raise ValueError("Either audio_path or audio_y should be specified")
        if sr is not None:
            assert sr == model_sr
        if audio_y is None:
            audio_y = _load_audio(audio_path, model_sr)
        return np.asarray(audio_y, dtype=np.float32)

    def _segment_length(self) -> int:
        return int(self.config.model.sample_rate * self.config.dataset.segment_duration)

    def generate_notes(self, audio_path=None, audio_y=None, sr=None, cond_index=None) -> np.ndarray:
        """Note array [n, 4] (onset_s, offset_s, pitch, velocity) for a whole recording."""
        samples = self._resolve_audio(audio_path, audio_y, sr)
        seg = self._segment_length()
        n_segments = -(-len(samples) // seg)                       # ceil
        padded = np.zeros(n_segments * seg, dtype=np.float32)
        padded[: len(samples)] = samples
        return self.sample_tokens(torch.from_numpy(padded).to(self.device), seg,
                                  split_duration=self.config.dataset.segment_duration, cond_index=cond_index)

    def _cond_rows(self, n_rows: int, cond_index: Optional[list]) -> torch.Tensor:
        """[n_rows, n_embeds] int64: the same (genre, difficulty) pair for every segment; zeros when None."""
        n_embeds = len(self.model.conditioning.embeds)
        rows = torch.zeros((n_rows, n_embeds))
        if cond_index is not None:
            rows = rows + torch.Tensor(cond_index)
        rows = rows.long()
        self.model.conditioning.check_indices(rows)      # IndexError on the host, as nn.Embedding raises it (ref input.py:57)
        return rows.to(self.device)

    @torch.no_grad()
    def sample_tokens(self, waveform: torch.Tensor, split_size: int, split_duration: float,
                      cond_index: Optional[list] = None) -> np.ndarray:
        """Segments -> chunks of inference.batch_size -> generate(max_length=1024) -> notes."""
        pieces = torch.split(waveform, split_size)
        per_call = int(self.config.inference.batch_size)
        token_rows = []
        for first in range(0, len(pieces), per_call):
            group = pieces[first:first + per_call]
            longest = max(p.shape[0] for p in group)
            wav = waveform.new_zeros((len(group), longest)).to(self.device)
            for row, piece in enumerate(group):                      # right-pad a ragged last piece with zeros
                wav[row, : piece.shape[0]] = piece
            # with a process group (one process per GPU) the chunk's segments are sharded over the ranks and the ids
            # all-gathered back in segment order; a single process decodes the chunk itself
            ids = D.generate_sharded(self.model.generate, ModelInputs(input_waveform=wav, cond_index=self._cond_rows(len(group), cond_index)),
                                     max_length=1024, pad_id=self.model.geometry.pad_token_id)
            token_rows.extend(ids.unbind(0))
        return self.model.tokenizer.decode(token_rows, mode="sequential", duration_per_batch=split_duration)
