#!/usr/bin/env python3
"""Stress of the early frame download (FLACGPU_EARLY_DOWNLOAD=1; VERDICT r02 item 6: a stream-ordered D2H copy
queued behind the packing kernels once delivered stale bytes for batches with a generic-path frame).
Many concurrent writers on pooled lanes, EVERY stream ending in a short (generic-path) frame, several batches per
stream; every output is compared byte for byte with the reference bytes of the same stream encoded WITHOUT the knob
(and one of each shape with the oracle).  usage: early_download_stress.py [rounds] [streams] [threads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 96
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 24
early = os.environ.get("FLACGPU_EARLY_DOWNLOAD") == "1"

from _pcm import synth_fast  # noqa: E402
from flac_codec_amd.encode import BatchEncoder, Options  # noqa: E402

B = 4096
shapes = []
for i in range(n_streams):
    frames = 3 + (i * 7) % 29                      # 3..31 whole blocks
    tail = 1 + (i * 977) % (B - 1)                 # a short last frame, always
    shapes.append((frames * B + tail, 100 + i % 13))
streams = [synth_fast(seed, 2, 24, n) for n, seed in shapes]
opts = Options.best().batch_frames(8).pipeline_depth(3)

# reference bytes: the same encoder without the knob (a child process so that the knob is not inherited)
ref_path = "/tmp/early_ref.npz"
if os.environ.get("EARLY_STRESS_CHILD") == "1":
    be = BatchEncoder(opts, threads=threads)
    outs = be.encode(streams, 48000, 24, 2)
    np.savez(ref_path, *[np.frombuffer(bytes(o), dtype=np.uint8) for o in outs])
    sys.exit(0)
import subprocess  # noqa: E402

env = dict(os.environ, EARLY_STRESS_CHILD="1")
env.pop("FLACGPU_EARLY_DOWNLOAD", None)
subprocess.check_call([sys.executable, os.path.abspath(__file__), "1", str(n_streams), str(threads)], env=env)
ref = [a.tobytes() for a in np.load(ref_path).values()]
import _oracle as orc  # noqa: E402

for k in (0, n_streams // 2):
    rc, o, _ = orc.encode_stream(orc.options("best"), 48000, 24, 2, streams[k], total_known=True)
    assert rc == 0 and o == ref[k], "the reference run itself differs from the oracle"

be = BatchEncoder(opts, threads=threads)
bad = 0
t0 = time.time()
for r in range(rounds):
    outs = be.encode(streams, 48000, 24, 2)
    for k, o in enumerate(outs):
        o = bytes(o)
        if o != ref[k]:
            bad += 1
            n = min(len(o), len(ref[k]))
            first = next((i for i in range(n) if o[i] != ref[k][i]), n)
            last = max((i for i in range(n) if o[i] != ref[k][i]), default=n)
            print(f"round {r} stream {k}: differs (len {len(o)} vs {len(ref[k])}); first diff at byte {first}, last at {last}",
                  flush=True)
import ctypes  # noqa: E402
from flac_codec_amd import _lib  # noqa: E402

q, w = ctypes.c_uint64(0), ctypes.c_uint64(0)
_lib.lib().flacgpu_early_download_counters(ctypes.byref(q), ctypes.byref(w))
print(f"early downloads queued {q.value}, completed by a remainder copy {w.value}")
print(f"early_download={'on' if early else 'off'} rounds {rounds} streams {n_streams} threads {threads}: "
      f"{rounds * n_streams} streams encoded, {bad} differ, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
