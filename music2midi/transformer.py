from music2midi_amd.transformer import T5Transformer  # noqa: F401
