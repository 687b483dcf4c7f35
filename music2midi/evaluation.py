from music2midi_amd.evaluation import (evaluate_batch, extract_midi_melody,  # noqa: F401
                                       melody_chroma_accuracy)
