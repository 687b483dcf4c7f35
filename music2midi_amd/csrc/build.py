"""Build libmusic2midi_amd.so for gfx950 with hipcc (in-tree, incremental).

    python -m music2midi_amd.csrc.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
BUILD = CSRC / "build"
LIB = PKG / "lib" / "libmusic2midi_amd.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# diagnostic builds: M2M_BUILD_VARIANT=stamps adds -DM2M_STAMPS and writes lib/libmusic2midi_amd_stamps.so
VARIANT = os.environ.get("M2M_BUILD_VARIANT", "")
EXTRA = os.environ.get("M2M_BUILD_EXTRA", "").split()   # experiment builds: extra -D flags, lib suffix = M2M_BUILD_TAG
TAG = os.environ.get("M2M_BUILD_TAG", "")
if EXTRA and not (TAG or VARIANT):
    raise SystemExit("M2M_BUILD_EXTRA needs M2M_BUILD_TAG (or M2M_BUILD_VARIANT): experiment flags never go into the product library")
FLAGS = FLAGS + EXTRA
if TAG:
    BUILD = CSRC / f"build_{TAG}"
    LIB = PKG / "lib" / f"libmusic2midi_amd_{TAG}.so"
if VARIANT == "stamps":
    FLAGS = FLAGS + ["-DM2M_STAMPS"]
    BUILD = CSRC / "build_stamps"
    LIB = PKG / "lib" / "libmusic2midi_amd_stamps.so"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _digest(src: Path, headers) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in [src, *headers]:
        h.update(p.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    BUILD.mkdir(exist_ok=True)
    LIB.parent.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "music2midi_amd.h"]
    sources = sorted(CSRC.glob("*.hip"))
    jobs = []
    for src in sources:
        obj = BUILD / (src.stem + ".o")
        stamp = BUILD / (src.stem + ".sha")
        dig = _digest(src, headers)
        if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
            continue
        jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        stamp.write_text(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [str(BUILD / (s.stem + ".o")) for s in sources]
    if jobs or not LIB.exists():
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
