from music2midi_amd.utils import numpy_to_midi  # noqa: F401
