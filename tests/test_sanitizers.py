"""CPU-only sanitizer job: the threaded host C++ of the product library (host/md5_mb.cpp,
frame_pack.cpp, checksums.cpp, lpc_host.cpp, stream_writer.cpp) built with ASan + UBSan and with TSan, and the
oracle built with ASan + UBSan, exercised by the host-side test files in a child interpreter that has the
sanitizer runtime preloaded.  Never on the GPU build path: the device code is not instrumented and these
libraries are only ever loaded here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flac-codec_amd", "csrc")


def _runtime(name):
    p = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _need_objects():
    # the sanitizer libraries link the HIP objects of the ordinary build; a snapshot without them (the GPU box
    # gets only the finished .so) would spend minutes recompiling the kernels for a CPU-side check
    if not os.path.exists(os.path.join(CSRC, "build", "flacenc_gpu.o")):
        pytest.skip("no HIP objects here (run __graft_entry__.build() first)")


def _make(target, cwd):
    r = subprocess.run(["make", "-C", cwd, target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]


def _run(files, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files,
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    return r.returncode, r.stdout


pytestmark = pytest.mark.skipif(os.environ.get("FLAC_IN_SANITIZER") == "1", reason="already inside the sanitizer child")


def test_host_code_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this image")
    _need_objects()
    _make("build/libflacenc_amd_asan.so", CSRC)
    _make("liboracle_asan.so", os.path.join(ROOT, "oracle"))
    rc, out = _run(["tests/test_md5_pool.py", "tests/test_doc_vectors.py", "tests/test_oracle_kat.py",
                    "tests/test_stream_header.py"],
                   {"LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1",
                    "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
                    "FLACENC_AMD_LIBRARY": os.path.join(CSRC, "build", "libflacenc_amd_asan.so"),
                    "FLAC_ORACLE_LIBRARY": os.path.join(ROOT, "oracle", "liboracle_asan.so"),
                    "FLAC_IN_SANITIZER": "1"})
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert rc == 0, out[-6000:]
    assert " passed" in out


def test_md5_engines_under_tsan():
    tsan = _runtime("libtsan.so")
    if tsan is None:
        pytest.skip("no libtsan in this image")
    _need_objects()
    _make("build/libflacenc_amd_tsan.so", CSRC)
    rc, out = _run(["tests/test_md5_pool.py"],
                   {"LD_PRELOAD": tsan, "TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0:exitcode=66",
                    "FLACENC_AMD_LIBRARY": os.path.join(CSRC, "build", "libflacenc_amd_tsan.so"),
                    "FLAC_IN_SANITIZER": "1"})
    # only races inside the product library count (the interpreter and the HIP runtime are not instrumented)
    reports = [blk for blk in out.split("WARNING: ThreadSanitizer")[1:] if "libflacenc_amd_tsan" in blk]
    assert not reports, reports[0][:6000]
    assert " passed" in out, out[-4000:]
