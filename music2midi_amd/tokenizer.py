"""MIDI note <-> token-id codec with the reference's vocabulary and call surface
(ref: music2midi/tokenizer.py:18-267): ``MidiTokenizer(config)``, ``__call__``,
``decode(mode="batched"|"sequential")``, ``to_string``, constants PAD/BOS/EOS/
ONSET/OFFSET.  CPU-side and numpy-only (no numba, no ``np.float_``); it is not
on the accelerated path, but callers of ``generate()`` need it unchanged.

Vocabulary (ref: config.yaml:33-37): ids 0-4 special, then ``pitch`` note ids,
then ``time`` step ids of ``midi_quantize_ms`` each.
"""
from __future__ import annotations

from typing import Iterable, List, Literal, Optional, Union

import numpy as np
import torch

PAD = 0
BOS = 1
EOS = 2
ONSET = 3
OFFSET = 4

_SPECIAL_NAMES = {PAD: "PAD", BOS: "BOS", EOS: "EOS", ONSET: "ONSET", OFFSET: "OFFSET"}


class MidiTokenizer:
    def __init__(self, config):
        self.config = config.tokenizer
        self.time_step = self.config.midi_quantize_ms / 1000
        self.pitch_token_offset = self.config.vocab_size.special
        self.time_token_offset = self.pitch_token_offset + self.config.vocab_size.pitch

    # ------------------------------------------------------------------ names
    def to_string(self, tokens) -> List[str]:
        names = []
        for token in tokens:
            if token in _SPECIAL_NAMES:
                names.append(_SPECIAL_NAMES[int(token)])
            elif token >= self.time_token_offset:
                names.append(f"time_{token - self.time_token_offset}")
            elif token >= self.pitch_token_offset:
                names.append(f"note_{token - self.pitch_token_offset}")
            else:
                raise ValueError(f"Invalid token '{token}'")
        return names

    # ----------------------------------------------------------------- encode
    def __call__(self, notes_batch: Iterable[np.ndarray], cutoff_time: Optional[int] = None) -> torch.Tensor:
        """Tokenize a batch of note arrays -> LongTensor [batch, Lmax], right-padded with PAD."""
        assert isinstance(notes_batch, Iterable), "notes should be passed in batch"
        rows = [self._tokenize(notes, cutoff_time) for notes in notes_batch]
        width = max((len(r) for r in rows), default=0)
        out = torch.full((len(rows), width), PAD, dtype=torch.long)
        for i, r in enumerate(rows):
            out[i, : len(r)] = r
        return out

    def _quantize(self, seconds: np.ndarray) -> np.ndarray:
        """seconds -> time-step index: rint(nextafter(x/step, +inf)), clipped to the vocabulary
        (ref tokenizer.py:120-126; the nextafter nudges exact .5 ties upward)."""
        steps = seconds / self.time_step
        steps = np.rint(np.nextafter(steps, steps + 1))
        return np.minimum(steps, self.config.vocab_size.time - 1)

    def _tokenize(self, notes: np.ndarray, cutoff_time: Optional[int] = None) -> torch.Tensor:
        """notes rows: (onset_s, offset_s, pitch, velocity).  One group per distinct time index:
        time token, then ONSET + pitches starting there, then OFFSET + pitches ending there; EOS last."""
        ids: List[float] = []
        if len(notes) > 0:
            notes = np.array(notes, dtype=np.float64, copy=True)
            if cutoff_time is not None:
                notes = notes[notes[:, 0] < cutoff_time]
            onset = notes[:, 0]
            offset = np.maximum(notes[:, 1], onset + self.time_step)   # every note lasts >= 1 step
            on_idx = self._quantize(onset)
            off_idx = self._quantize(offset)
            pitch_ids = notes[:, 2] + self.pitch_token_offset
            for step in np.unique(np.concatenate([on_idx, off_idx])):
                ids.append(step + self.time_token_offset)
                starting = pitch_ids[on_idx == step]
                ending = pitch_ids[off_idx == step]
                if len(starting):
                    ids.append(ONSET)
                    ids.extend(starting.tolist())
                if len(ending):
                    ids.append(OFFSET)
                    ids.extend(ending.tolist())
        ids.append(EOS)
        # the reference builds a float32 tensor and truncates it to int64
        return torch.from_numpy(np.asarray(ids, dtype=np.float32).astype(np.int64))

    # ----------------------------------------------------------------- decode
    def decode(
        self,
        tokens_batch: Iterable[Union[np.ndarray, torch.Tensor]],
        mode: Literal["batched", "sequential"] = "batched",
        duration_per_batch: Optional[float] = None,
        cutoff_time: Optional[int] = None,
    ) -> Union[List[np.ndarray], np.ndarray]:
        """Token rows -> note arrays (seconds).

        ``batched``: every row decoded on its own -> list of [n_i, 4] arrays.
        ``sequential``: row i is segment i of one recording; its time indices are shifted by
        ``i * round(duration_per_batch / time_step)`` steps and all notes are concatenated.
        """
        if mode == "batched":
            return [self._decode(tokens, 0, cutoff_time) for tokens in tokens_batch]
        if mode == "sequential":
            assert (
                duration_per_batch is not None
            ), 'duration_per_batch is required for mode="sequential"'
            steps_per_segment = round(duration_per_batch / self.time_step)
            parts = [self._decode(tokens, i * steps_per_segment, cutoff_time) for i, tokens in enumerate(tokens_batch)]
            return np.concatenate(parts)
        raise ValueError(f"Invalid argument mode={mode}")

    def _decode(self, tokens, start_idx: int = 0, cutoff_time: Optional[int] = None) -> np.ndarray:
        if isinstance(tokens, torch.Tensor):
            tokens = tokens.cpu().numpy()
        notes = self._decode_tokens(np.asarray(tokens), start_idx)
        notes = notes[notes[:, 1] != -1]            # onsets that never got an offset are dropped
        notes[:, :2] = notes[:, :2] * self.time_step
        if cutoff_time is not None:
            notes = notes[notes[:, 0] < cutoff_time]
            notes[:, 1] = np.where(notes[:, 1] > cutoff_time, cutoff_time, notes[:, 1])
        return notes

    def _decode_tokens(self, tokens: np.ndarray, start_idx: int) -> np.ndarray:
        """State machine over (current time index, onset/offset mode, pitch)
        (ref tokenizer.py:169-200): a time token resets mode and pitch; a pitch token under
        ONSET opens a note, under OFFSET closes every still-open earlier note of that pitch."""
        onset_t: List[int] = []
        offset_t: List[int] = []
        pitches: List[int] = []
        velocities: List[int] = []
        time_idx, mode, pitch = -1, -1, -1
        for token in tokens.tolist():
            if token == EOS:
                break
            if token == BOS or token == PAD:
                continue
            if token == ONSET:
                mode = 1
            elif token == OFFSET:
                mode = 0
            if token >= self.time_token_offset:
                time_idx, mode, pitch = start_idx + token - self.time_token_offset, -1, -1
            elif token >= self.pitch_token_offset:
                pitch = token - self.pitch_token_offset
            if time_idx == -1 or mode == -1 or pitch == -1:
                continue
            velocity = mode * self.config.default_velocity
            if velocity:
                onset_t.append(time_idx)
                offset_t.append(-1)
                pitches.append(pitch)
                velocities.append(velocity)
            else:
                for i in range(len(pitches)):
                    if pitches[i] == pitch and offset_t[i] == -1 and onset_t[i] < time_idx:
                        offset_t[i] = time_idx
            pitch = -1
        out = np.zeros((len(pitches), 4), dtype=np.float64)
        if pitches:
            out[:, 0], out[:, 1], out[:, 2], out[:, 3] = onset_t, offset_t, pitches, velocities
        return out
