"""Run a reference-shaped caller (``evaluate.py``, ``webui.py``, a notebook export ...) against this
implementation WITHOUT editing it:

    python -m music2midi_amd.run evaluate.py data/ --ckpt last.ckpt          # from the reference checkout
    python -m music2midi_amd.run /path/to/webui.py --ckpt last.ckpt

Why a launcher: a script started as ``python evaluate.py`` gets its own directory as ``sys.path[0]``,
ahead of ``PYTHONPATH`` and of any installed package, so inside a reference checkout
``import music2midi`` always finds the checkout's own ``music2midi/`` (ref: evaluate.py:9-11,
webui.py:9-10).  This module puts the directory that holds the ``music2midi`` shim of THIS repo in
front of the script's directory, then runs the script as ``__main__`` with ``runpy`` — in this same
process (no exec; nothing has touched the GPU yet).  Sub-modules the shim does not provide
(``music2midi.webui_utils``, ``.plot_midi``, ``.dataset``) keep resolving from the caller's checkout:
``music2midi/__init__.py`` appends that directory to the package ``__path__``.
"""
from __future__ import annotations

import os
import runpy
import sys
from pathlib import Path

SHIM_ROOT = Path(__file__).resolve().parents[1]      # holds music2midi/ (shim) and music2midi_amd/


def prepare_path(script: Path) -> None:
    """sys.path = [shim root, script dir, ...the rest without duplicates of the two]."""
    script_dir = str(script.resolve().parent)
    shim = str(SHIM_ROOT)
    rest = [p for p in sys.path if p not in (shim, script_dir, "")]
    sys.path[:] = [shim, script_dir] + rest
    # lets music2midi/__init__.py find the caller's checkout even when cwd is elsewhere
    os.environ.setdefault("MUSIC2MIDI_REFERENCE", script_dir)
    stale = [m for m in sys.modules if m == "music2midi" or m.startswith("music2midi.")]
    for m in stale:                                  # a copy imported before the path was fixed must not survive
        del sys.modules[m]


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if argv else 2)
    script = Path(argv[0])
    if not script.is_file():
        raise SystemExit(f"music2midi_amd.run: {script} is not a file")
    prepare_path(script)
    sys.argv = [str(script)] + argv[1:]
    runpy.run_path(str(script), run_name="__main__")


if __name__ == "__main__":
    main()
