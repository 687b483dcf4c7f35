"""music2midi_amd — MI355X-native implementation of the Music2MIDI inference hot path.

Public surface mirrors the reference package (``music2midi``): see
``music2midi_amd.model.Music2MIDI``, ``music2midi_amd.transformer.T5Transformer``,
``music2midi_amd.input`` and ``music2midi_amd.tokenizer``.
"""
__version__ = "0.1.0"
