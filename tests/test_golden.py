"""Committed golden vectors (tests/golden/oracle_vectors.json, made by make_golden.py):
 * CPU: the oracle still reproduces them (guards against oracle drift);
 * GPU: the HIP path reproduces the same streams byte for byte (SHA-256 of the whole .flac)."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_inputs_and_oracle_reproduce_golden_vectors():
    mod = _cases()
    fresh = mod.build()
    assert set(fresh) == set(VEC)
    for name in VEC:
        assert fresh[name]["pcm_sha256"] == VEC[name]["pcm_sha256"], f"{name}: input generator drifted"
        assert fresh[name] == json.loads(json.dumps(VEC[name])), f"{name}: oracle output drifted"


@pytest.mark.gpu
def test_gpu_streams_match_golden_sha256():
    from flac_codec_amd.encode import FlacSampleWriter, Options

    mod = _cases()
    for name, preset, ov, rate, bps, ch, pcm in mod.cases():
        g = VEC[name]
        assert hashlib.sha256(pcm.tobytes()).hexdigest() == g["pcm_sha256"]
        opts = getattr(Options, preset)()
        if "max_lpc_order" in ov:
            opts.max_lpc_order(ov["max_lpc_order"] or None)
        if ov.get("padding", 0) < 0:
            opts.no_padding()
        w = FlacSampleWriter(None, opts, rate, bps, ch, pcm.size - pcm.size % ch)
        w.write(pcm[: pcm.size - pcm.size % ch])
        w.finalize()
        data = w.getvalue()
        w.close()
        assert len(data) == g["flac_len"], name
        assert hashlib.sha256(data).hexdigest() == g["flac_sha256"], name
