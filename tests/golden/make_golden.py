#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json from the CPU oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference itself cannot be executed here (Rust, no
toolchain), so these vectors pin the ORACLE's output -- decision records, exact bit counts and
the SHA-256 of whole .flac streams -- for the reference's own test inputs and for the seeded
synthetic generators.  They protect against silent drift of the oracle and are the committed
expected outputs the GPU path is compared with on the GPU box."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import _oracle as orc  # noqa: E402
from _pcm import generate_sine_2, read_raw, synth_fast  # noqa: E402


def cases():
    """(name, preset, overrides, rate, bps, channels, pcm)"""
    yield "synth-2ch-24-best", "best", {}, 48000, 24, 2, synth_fast(30, 2, 24, 4096 * 4)
    yield "synth-2ch-16-default", "default", {}, 48000, 16, 2, synth_fast(21, 2, 16, 4096 * 4)
    yield "synth-2ch-16-fixed", "default", {"max_lpc_order": 0}, 48000, 16, 2, synth_fast(20, 2, 16, 4096 * 4)
    yield "synth-8ch-24-best", "best", {}, 192000, 24, 8, synth_fast(40, 8, 24, 4096 * 2)
    yield "synth-2ch-24-order32", "best", {"max_lpc_order": 32}, 96000, 24, 2, synth_fast(50, 2, 24, 4096 * 3)
    yield "synth-2ch-16-fast", "fast", {}, 44100, 16, 2, synth_fast(70, 2, 16, 1152 * 5 + 5)
    yield "sine-2ch-16-default", "default", {}, 44100, 16, 2, generate_sine_2(32767.0, 44100.0, 4096 * 4, 441.0, 0.5, 441.0, 0.0, 1.0)
    yield "wasted-bits", "default", {}, 44100, 16, 1, read_raw("wasted-bits.raw", 16)
    for ch, bps in ((2, 16), (2, 24), (8, 24)):
        yield f"roundtrip-{ch}-{bps}-4777", "default", {"padding": -1}, 44100, bps, ch, read_raw(f"roundtrip-{ch}-{bps}-4777.raw", bps)
    rc, pcm, info = orc.decode_stream(open(os.path.join(HERE, "refdata", "sine.flac"), "rb").read())
    yield "sine.flac-first-3-blocks", "best", {}, 44100, 16, 2, pcm[: 4096 * 3 * 2]


def sub_summary(s, n):
    npart = s.n_partitions
    return {"type": s.type, "wasted": s.wasted, "bps": s.bps, "order": s.order,
            "precision": s.precision, "shift": s.shift, "coeffs": list(s.coeffs[: s.order]) if s.type == 3 else [],
            "method": s.coding_method, "porder": s.partition_order, "npart": npart,
            "rice": list(s.rice[:npart]), "escape": list(s.escape_bits[:npart]), "bits": s.bits}


def build():
    out = {}
    for name, preset, ov, rate, bps, ch, pcm in cases():
        o = orc.options(preset, **ov)
        rc, data, st = orc.encode_stream(o, rate, bps, ch, pcm, total_known=True)
        assert rc == 0, name
        bs = o.block_size
        total = pcm.size // ch
        frames = []
        mat = pcm[: total * ch].reshape(total, ch)
        for f, s in enumerate(range(0, total, bs)):
            planar = np.ascontiguousarray(mat[s:s + bs].T)
            rc, fb, plan = orc.encode_frame(o, rate, bps, planar, frame_number=f)
            assert rc == 0
            frames.append({"assignment": plan.assignment, "bytes": len(fb),
                           "sha256": hashlib.sha256(fb).hexdigest()[:16],
                           "subs": [sub_summary(plan.sub[c], planar.shape[1]) for c in range(ch)]})
            if f >= 3:
                break
        out[name] = {"preset": preset, "overrides": ov, "rate": rate, "bps": bps, "channels": ch,
                     "samples": int(pcm.size), "pcm_sha256": hashlib.sha256(pcm.tobytes()).hexdigest(),
                     "flac_len": len(data), "flac_sha256": hashlib.sha256(data).hexdigest(),
                     "md5": bytes(st.md5).hex(), "frames": frames}
    return out


def export_inputs(dst):
    """The cases' PCM as little-endian ceil(bps / 8)-byte samples + manifest.tsv: the input of tools/ref_golden (the reference
    encoder run by a maintainer with a Rust toolchain)."""
    os.makedirs(dst, exist_ok=True)
    rows = ["# name\tpreset\tmax_lpc_order (-1: preset)\tpadding (-1: none)\trate\tbps\tchannels"]
    for name, preset, ov, rate, bps, ch, pcm in cases():
        width = (bps + 7) // 8
        pcm = pcm[: pcm.size - pcm.size % ch]
        raw = np.ascontiguousarray(pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width]).tobytes()
        with open(os.path.join(dst, name + ".raw"), "wb") as f:
            f.write(raw)
        rows.append("\t".join(str(v) for v in (name, preset, ov.get("max_lpc_order", -1), -1 if ov.get("padding", 0) < 0 else 0,
                                               rate, bps, ch)))
    with open(os.path.join(dst, "manifest.tsv"), "w") as f:
        f.write("\n".join(rows) + "\n")
    print(f"{len(rows) - 1} inputs written to {dst}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--export-inputs":
        export_inputs(sys.argv[2])
        sys.exit(0)
    vec = build()
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1, sort_keys=True)
    print(f"{len(vec)} cases written")
