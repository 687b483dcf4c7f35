"""GPU, end to end on synthetic music: the whole product path as a caller uses it — waveforms -> fused STFT/log-mel kernel ->
native training step (forward + backward + Adafactor, ref model.py:27-43) for a few hundred steps on eight clips of decaying
harmonic tones with known notes -> KV-cached greedy decode (ref transformer.py:41-45) -> tokenizer.decode -> chroma accuracy
(ref evaluation.py).  If any stage were wrong (frontend, gradients, optimizer, weight hand-over to the inference path, decode
loop, token semantics, metric) the transcription could not come back note for note."""
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_overfit_synthetic_tones_and_transcribe_them_back(precision):
    import train_demo
    torch.manual_seed(0)
    losses, score0, score1 = train_demo.main(steps=600, B=8, precision=precision, verbose=True)
    assert losses[0] > 20.0 and losses[-1] < 0.05, (losses[0], losses[-1])
    assert score0 < 0.3 and score1 > 0.9, (score0, score1)
