"""Drop-in import name: ``music2midi`` resolves to the MI355X implementation, so the reference's
``evaluate.py`` / ``webui.py`` / ``demo.ipynb`` run unchanged against it (start them with
``python -m music2midi_amd.run <script> ...`` — see that module for why a launcher is needed).

The hot-path modules (``model``, ``transformer``, ``input``, ``tokenizer``, ``evaluation``, ``utils``)
live here.  Everything else the reference package holds (``webui_utils``, ``plot_midi``, ``dataset`` —
UI, plotting and training-data code that is not part of the path) is NOT re-implemented or copied:
when a reference checkout is visible its ``music2midi/`` directory is appended to this package's
``__path__``, so ``import music2midi.webui_utils`` (ref: webui.py:9) finds the caller's own file while
``music2midi.model`` still resolves to this implementation (first ``__path__`` entry wins).
"""
import os as _os
import sys as _sys
from pathlib import Path as _Path


def _reference_package_dirs():
    here = _Path(__file__).resolve().parent
    roots = []
    if _os.environ.get("MUSIC2MIDI_REFERENCE"):
        roots.append(_os.environ["MUSIC2MIDI_REFERENCE"])
    roots += [p or "." for p in _sys.path] + [_os.getcwd()]
    out = []
    for root in roots:
        try:
            cand = (_Path(root) / "music2midi").resolve()
        except OSError:
            continue
        if cand != here and cand.is_dir() and str(cand) not in out:
            out.append(str(cand))
    return out


__path__.extend(_reference_package_dirs())           # ours stays first: hot-path modules are never shadowed

from music2midi_amd import __version__  # noqa: E402,F401
from music2midi_amd.evaluation import (evaluate_batch, extract_midi_melody,  # noqa: E402,F401
                                       melody_chroma_accuracy)
from music2midi_amd.model import Music2MIDI  # noqa: E402,F401
