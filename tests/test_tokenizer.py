"""CPU: MidiTokenizer vs vectors produced by the reference's own tokenizer.py (golden), plus
round-trip properties."""
import json

import numpy as np
import pytest
import torch

from music2midi_amd.config import default_config
from music2midi_amd.tokenizer import BOS, EOS, OFFSET, ONSET, PAD, MidiTokenizer


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.loads((golden_dir / "tokenizer_cases.json").read_text())


@pytest.fixture(scope="module")
def tok():
    return MidiTokenizer(default_config())


def test_constants():
    assert (PAD, BOS, EOS, ONSET, OFFSET) == (0, 1, 2, 3, 4)


def test_encode_matches_reference(tok, cases):
    n = 0
    for c in cases:
        if c["kind"] == "encode":
            notes = np.asarray(c["notes"], dtype=np.float64).reshape(-1, 4)
            got = tok._tokenize(notes, c["cutoff"]).tolist()
            assert got == c["ids"], c["name"]
            n += 1
        elif c["kind"] == "encode_batch":
            batch = [np.asarray(x, dtype=np.float64).reshape(-1, 4) for x in c["notes"]]
            got = tok(batch)
            assert got.dtype == torch.long and got.tolist() == c["ids"]
            n += 1
    assert n >= 19


def test_decode_matches_reference(tok, cases):
    n = 0
    for c in cases:
        if c["kind"] == "decode":
            got = tok._decode(np.asarray(c["ids"]), 0, c["cutoff"])
            want = np.asarray(c["notes"], dtype=np.float64).reshape(-1, 4)
            assert got.shape == want.shape and np.array_equal(got, want), c["name"]
            n += 1
        elif c["kind"] == "decode_sequential":
            got = tok.decode([np.asarray(s) for s in c["ids"]], mode="sequential", duration_per_batch=c["duration"])
            assert np.array_equal(got, np.asarray(c["notes"]).reshape(-1, 4))
            n += 1
        elif c["kind"] == "decode_batched":
            got = tok.decode(torch.tensor(c["ids"]), mode="batched")
            for g, w in zip(got, c["notes"]):
                assert np.array_equal(g, np.asarray(w, dtype=np.float64).reshape(-1, 4))
            n += 1
        elif c["kind"] == "to_string":
            assert tok.to_string(np.asarray(c["ids"])) == c["names"]
            n += 1
    assert n >= 21


def test_error_behaviour(tok):
    with pytest.raises(AssertionError, match="duration_per_batch is required"):
        tok.decode([np.array([2])], mode="sequential")
    with pytest.raises(ValueError, match="Invalid argument mode"):
        tok.decode([np.array([2])], mode="nope")
    with pytest.raises(AssertionError, match="notes should be passed in batch"):
        tok(5)
    with pytest.raises(ValueError, match="Invalid token"):
        tok.to_string([-1])


def test_roundtrip_on_grid_notes(tok):
    """Notes already on the 50 ms grid, distinct pitches: decode(encode(x)) == x."""
    rng = np.random.default_rng(0)
    for _ in range(20):
        n = int(rng.integers(1, 30))
        onset = np.sort(rng.integers(0, 150, n)) * 0.05
        dur = rng.integers(1, 40, n) * 0.05
        pitch = rng.permutation(np.arange(21, 109))[:n].astype(float)
        notes = np.stack([onset, np.minimum(onset + dur, 199 * 0.05), pitch, np.full(n, 80.0)], axis=1)
        notes = notes[notes[:, 1] > notes[:, 0]]
        back = tok.decode(tok([notes]), mode="batched")[0]
        key = lambda a: a[np.lexsort((a[:, 2], a[:, 0]))]
        assert np.allclose(key(back), key(notes), atol=1e-9)


def test_sequential_offsets_by_segment(tok):
    ids = tok([np.array([[0.5, 1.0, 60, 80]])])[0]
    out = tok.decode([ids, ids, ids], mode="sequential", duration_per_batch=3)
    assert np.allclose(out[:, 0], [0.5, 3.5, 6.5]) and np.allclose(out[:, 1], [1.0, 4.0, 7.0])
