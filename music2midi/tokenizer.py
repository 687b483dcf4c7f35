from music2midi_amd.tokenizer import BOS, EOS, OFFSET, ONSET, PAD, MidiTokenizer  # noqa: F401
