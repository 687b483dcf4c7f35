"""Editable install: `pip install -e .` puts `music2midi_amd` and the drop-in `music2midi` alias on the path.
The HIP library is built in-tree (not by pip): `python -m music2midi_amd.csrc.build`."""
from setuptools import find_packages, setup

setup(
    name="music2midi-mi355x",
    version="0.1.0",
    description="MI355X-native Music2MIDI inference hot path (hand-written HIP behind a C ABI)",
    packages=find_packages(include=["music2midi_amd", "music2midi_amd.*", "music2midi", "music2midi.*"]),
    package_data={"music2midi_amd": ["lib/*.so", "csrc/*.hip", "csrc/*.h"]},
    python_requires=">=3.9",
    install_requires=["numpy", "pyyaml", "torch"],
)
