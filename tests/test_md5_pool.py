"""Multi-stream MD5 engine (csrc/host/md5_mb.cpp): up to 16 chains per engine thread in lockstep, one per
AVX-512 lane.  The digests must be the scalar ones (RFC 1321) whatever the run lengths and however the runs
of different streams interleave; hashlib is the third opinion through the writers (test_gpu_stream's MD5
checks run on the GPU box)."""
import ctypes as C

import pytest

from flac_codec_amd import _lib


def _hooks():
    L = _lib.stream_lib() if hasattr(_lib, "stream_lib") else _lib.lib()
    L.flacenc_md5_selftest.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.flacenc_md5_selftest.restype = C.c_int
    L.flacenc_md5_simd_available.restype = C.c_int
    return L


@pytest.mark.parametrize("streams,runs", [(1, 9), (2, 9), (3, 11), (15, 8), (16, 12), (17, 7), (48, 6), (64, 5), (80, 4)])
def test_pool_digests_match_scalar(streams, runs):
    L = _hooks()
    assert L.flacenc_md5_selftest(streams, runs, 4242 + streams) == 0


def test_scalar_fallback(monkeypatch):
    """FLACENC_MD5_SCALAR is read once per process: run the fallback in a child process."""
    import subprocess
    import sys

    code = ("import ctypes as C, os; os.environ['FLACENC_MD5_SCALAR']='1';"
            "from flac_codec_amd import _lib;"
            "L = _lib.stream_lib() if hasattr(_lib, 'stream_lib') else _lib.lib();"
            "L.flacenc_md5_selftest.argtypes=[C.c_uint32]*3;"
            "assert L.flacenc_md5_simd_available() == 0;"
            "assert L.flacenc_md5_selftest(20, 6, 7) == 0")
    subprocess.run([sys.executable, "-c", code], check=True)
