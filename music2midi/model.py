from music2midi_amd.model import Music2MIDI  # noqa: F401
