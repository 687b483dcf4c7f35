from music2midi_amd.input import Conditioning, LogMelSpectrogram, ModelInputs  # noqa: F401
