"""CPU oracle for the Music2MIDI inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the checker or as the
reported CPU baseline — never as the thing measured or shipped.  The product
path (``music2midi_amd``) fails loudly when the HIP library is missing; it has
no CPU fallback and never routes through this package.

What it restates (arithmetic lives in third-party packages that are NOT under
/root/reference; versions are the reference's pins, ref: environment.yaml):

* ``logmel.py``  — torchaudio==2.1.0 ``transforms.MelSpectrogram`` as called at
  ref: music2midi/input.py:25-41 (``torch.stft`` + ``melscale_fbanks``).
* ``t5.py``      — transformers==4.34.0 ``T5ForConditionalGeneration`` forward
  and greedy ``generate`` as called at ref: music2midi/transformer.py:28-45.
* ``chroma.py``  — pretty_midi==0.2.10 ``get_piano_roll(fs, times)`` + mir_eval==0.6
  ``melody.to_cent_voicing`` / ``raw_chroma_accuracy`` as called at
  ref: music2midi/evaluation.py:21-75 (plain loops; parity unpinned: neither package is in the image).

Pinning.  The reference has no tests, golden vectors or
fixtures, so nothing of its own pins this path.  The oracle is pinned instead
against outputs produced in the build container by the real third-party code
that is importable there: ``torch.stft`` (frontend STFT half), HuggingFace
``T5ForConditionalGeneration`` 5.15.0 forced to eager attention with an untied
``lm_head`` (= 4.34.0 semantics), and the reference's own
``music2midi/tokenizer.py`` imported under three shims.  The generating script
is ``tests/golden/make_golden.py``; its outputs are the fixtures in
``tests/golden/``.  The mel *filterbank* half cannot be cross-checked
(torchaudio is absent from the image): it is pinned only by this restatement
of the published ``melscale_fbanks`` formula — "filterbank parity unpinned".
"""
