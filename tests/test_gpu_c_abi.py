"""The C ABI used from plain C: examples/c_abi_demo.c is compiled with gcc against
include/flacenc_stream.h + libflacenc_amd.so, run on the GPU, and the .flac it reports (size +
FNV-1a hash) must be the oracle's stream for the same signal."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_signal(frames):
    s = 12345
    y1 = y2 = 0
    out = np.empty(frames * 2, dtype=np.int32)
    for i in range(frames):
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        e = ((s >> 16) & 0x3FFF) - 8192
        y = ((58000 * y1 - 29491 * y2) >> 15) + e
        y = max(-30000, min(30000, y))
        y2, y1 = y1, y
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        e2 = ((s >> 16) & 0x7FF) - 1024
        out[2 * i] = y
        out[2 * i + 1] = ((3 * y) >> 2) + e2
    return out


def fnv1a(data):
    h = 1469598103934665603
    for chunk_start in range(0, len(data), 1 << 16):
        for b in data[chunk_start:chunk_start + (1 << 16)]:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("preset,frames", [("best", 30000), ("fast", 20000)])
def test_c_program_matches_oracle(tmp_path, preset, frames):
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(ROOT, "flac-codec_amd")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + libdir, "-lflacenc_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.check_output([exe, str(frames), preset], text=True).split()
    size, digest, n_frames = int(out[0]), int(out[1], 16), int(out[2])
    pcm = make_signal(frames)
    rc, ref, _ = orc.encode_stream(orc.options(preset), 44100, 16, 2, pcm, total_known=True)
    assert rc == 0
    assert size == len(ref)
    assert digest == fnv1a(ref)
    block = 1152 if preset == "fast" else 4096
    assert n_frames == (frames + block - 1) // block
