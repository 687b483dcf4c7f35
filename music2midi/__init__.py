"""Drop-in import name: ``music2midi`` resolves to the MI355X implementation, so the
reference's ``evaluate.py`` / ``webui.py`` / ``demo.ipynb`` run unchanged against it."""
from music2midi_amd import __version__  # noqa: F401
