"""CPU: a reference-shaped caller ends up on THIS implementation.

The temp directory is laid out like a reference checkout: its own ``music2midi/`` package (stubs whose
classes carry a marker, plus the UI helper module this repo does not provide) and caller scripts whose
import lines are the ones of ref evaluate.py:9-11 and webui.py:9-10."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

CALLER = '''
import json, sys
from music2midi.evaluation import evaluate_batch          # ref evaluate.py:9
from music2midi.model import Music2MIDI                   # ref evaluate.py:10 / webui.py:10
from music2midi.utils import numpy_to_midi                # ref evaluate.py:11
import music2midi.webui_utils as utils                    # ref webui.py:9
import music2midi, music2midi.model, music2midi.tokenizer, music2midi.input, music2midi.transformer
print(json.dumps({"model": Music2MIDI.__module__, "eval": evaluate_batch.__module__, "midi": numpy_to_midi.__module__,
                  "utils_file": utils.__file__, "utils_marker": utils.MARKER, "argv": sys.argv[1:],
                  "tok": music2midi.tokenizer.MidiTokenizer.__module__, "inp": music2midi.input.ModelInputs.__module__,
                  "t5": music2midi.transformer.T5Transformer.__module__, "main": __name__}))
'''


def _checkout(tmp_path: Path) -> Path:
    co = tmp_path / "reference_checkout"
    pkg = co / "music2midi"
    pkg.mkdir(parents=True)
    (pkg / "__init__.py").write_text("")
    (pkg / "model.py").write_text("class Music2MIDI:\n    STUB = True\n")
    (pkg / "evaluation.py").write_text("def evaluate_batch(a, b):\n    raise RuntimeError('checkout stub')\n")
    (pkg / "utils.py").write_text("def numpy_to_midi(n):\n    raise RuntimeError('checkout stub')\n")
    (pkg / "webui_utils.py").write_text("MARKER = 'from-the-checkout'\n")
    (co / "caller.py").write_text(CALLER)
    return co


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "MUSIC2MIDI_REFERENCE")}


def test_plain_python_resolves_the_checkouts_own_package(tmp_path):
    """The failure the launcher exists for: sys.path[0] (the script's directory) beats PYTHONPATH."""
    co = _checkout(tmp_path)
    (co / "probe.py").write_text("from music2midi.model import Music2MIDI\nprint(Music2MIDI.__module__, getattr(Music2MIDI, 'STUB', False))\n")
    r = subprocess.run([sys.executable, "probe.py"], cwd=co, env=dict(_clean_env(), PYTHONPATH=str(ROOT)),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == ["music2midi.model", "True"]


def test_launcher_runs_a_reference_shaped_caller_on_this_implementation(tmp_path):
    co = _checkout(tmp_path)
    r = subprocess.run([sys.executable, "-m", "music2midi_amd.run", "caller.py", "data_dir", "--ckpt", "x.ckpt"], cwd=co,
                       env=dict(_clean_env(), PYTHONPATH=str(ROOT)), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["model"] == "music2midi_amd.model" and rec["eval"] == "music2midi_amd.evaluation"
    assert rec["midi"] == "music2midi_amd.utils" and rec["tok"] == "music2midi_amd.tokenizer"
    assert rec["inp"] == "music2midi_amd.input" and rec["t5"] == "music2midi_amd.transformer"
    # the UI helper is the CHECKOUT's file (never copied into this repo)
    assert rec["utils_marker"] == "from-the-checkout" and Path(rec["utils_file"]).parent == co / "music2midi"
    assert rec["argv"] == ["data_dir", "--ckpt", "x.ckpt"] and rec["main"] == "__main__"


def test_launcher_from_another_directory_with_absolute_script_path(tmp_path):
    co = _checkout(tmp_path)
    r = subprocess.run([sys.executable, "-m", "music2midi_amd.run", str(co / "caller.py")], cwd=ROOT, env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["model"] == "music2midi_amd.model" and rec["utils_marker"] == "from-the-checkout"


def test_shim_alone_reports_missing_ui_module_clearly(tmp_path):
    """Without any reference checkout in sight the UI helper is simply absent (it is not part of the path)."""
    r = subprocess.run([sys.executable, "-c", "import music2midi.model, importlib.util as u; "
                        "print(u.find_spec('music2midi.webui_utils') is None)"], cwd=tmp_path,
                       env=dict(_clean_env(), PYTHONPATH=str(ROOT)), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "True", r.stderr[-2000:]
