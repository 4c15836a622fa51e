"""Golden streams made by the REFERENCE encoder itself (tools/ref_golden, run by a maintainer with a Rust toolchain:
tools/ref_golden/README.md).  When tests/golden/ref_flac/ holds them, the oracle's stream -- and on a GPU box the HIP
path's -- must be those bytes: whole-bitstream parity pinned by the reference instead of by review of the restatement
(SURVEY.md 8(c)).  Skipped while the directory is absent (this image has no cargo)."""
import importlib.util
import os

import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_flac")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="no tests/golden/ref_flac (tools/ref_golden needs cargo)")


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return list(mod.cases())


def _ref(name):
    path = os.path.join(REF, name + ".flac")
    if not os.path.isfile(path):
        pytest.skip(f"{name}.flac not generated")
    return open(path, "rb").read()


def test_oracle_streams_are_the_reference_encoders_bytes():
    for name, preset, ov, rate, bps, ch, pcm in _cases():
        want = _ref(name)
        pcm = pcm[: pcm.size - pcm.size % ch]
        rc, data, _ = orc.encode_stream(orc.options(preset, **ov), rate, bps, ch, pcm, total_known=True)
        assert rc == 0 and data == want, f"{name}: the oracle's stream differs from the reference encoder's"


@pytest.mark.gpu
def test_gpu_streams_are_the_reference_encoders_bytes():
    from flac_codec_amd.encode import FlacSampleWriter, Options

    for name, preset, ov, rate, bps, ch, pcm in _cases():
        want = _ref(name)
        opts = getattr(Options, preset)()
        if "max_lpc_order" in ov:
            opts.max_lpc_order(ov["max_lpc_order"] or None)
        if ov.get("padding", 0) < 0:
            opts.no_padding()
        pcm = pcm[: pcm.size - pcm.size % ch]
        w = FlacSampleWriter(None, opts, rate, bps, ch, pcm.size)
        w.write(pcm)
        w.finalize()
        data = w.getvalue()
        w.close()
        assert data == want, f"{name}: the GPU path's stream differs from the reference encoder's"
