#!/usr/bin/env python3
"""Where a frame-assembly workgroup's time goes (k_frame64<128,64,*,DIRECT>; a library built with -DFRAME_TIMELINE=1):
    tools/build_variant.sh ftl "-DFRAME_TIMELINE=1"
    FLACENC_AMD_LIBRARY=gpurun_variants/ftl.so python3 tools/frame_timeline.py [--signal ar2|hi] [--config 3]
Stamps (100 MHz) of wave c of frame f in the autocorrelation row of candidate (f, c): 0 start, 1 image zeroed + tables + header,
2 subframe emitted, 3 barrier, 4 CRC-16 done, 5 copied out."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--signal", default="ar2")
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--frames", type=int, default=bench.FRAMES)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch

    w = bench.Workload(torch, a.config, a.frames, 0, 1, 0, 0, signal=a.signal)
    w.only_first = True
    w.prewarm(200)
    for _ in range(4):
        w.step()
    torch.cuda.synchronize()
    an = w.ans[0]
    F, NC = a.frames, 4
    rows = torch.empty(F * NC * 36, dtype=torch.int64, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = hip.hipMemcpy(rows.data_ptr(), an.device_buffer(6), rows.numel() * 8, 3)
    assert rc == 0, rc
    t = rows.cpu().numpy().reshape(F, NC, 36)[:, :2, :9]
    st = t.astype(np.float64) * 0.01
    res = {"signal": a.signal, "config": a.config}

    def q(name, v):
        r = {"mean": float(v.mean()), "p50": float(np.median(v)), "p90": float(np.percentile(v, 90)), "max": float(v.max())}
        print(f"{name:44s} mean {r['mean']:7.2f}  p50 {r['p50']:7.2f}  p90 {r['p90']:7.2f}  max {r['max']:7.2f} us")
        res[name] = r

    q("0-1 zero the image, tables, header", st[:, :, 1] - st[:, :, 0])
    if st[:, :, 6].max() > 0:   # (r06: stamps inside the preparation)
        q("  0-6 scalars in, the wave's loads requested", st[:, :, 6] - st[:, :, 0])
        q("  6-7 image zeroed", st[:, :, 7] - st[:, :, 6])
        q("  7-8 table piece stored, the two waves met", st[:, :, 8] - st[:, :, 7])
        q("  8-1 frame header", st[:, :, 1] - st[:, :, 8])
    q("1-2 plan + samples, residual, Rice emission", st[:, :, 2] - st[:, :, 1])
    q("2-3 barrier (the other subframe)", st[:, :, 3] - st[:, :, 2])
    q("3-4 CRC-16", st[:, :, 4] - st[:, :, 3])
    q("4-5 copy out", st[:, :, 5] - st[:, :, 4])
    q("0-5 workgroup", st[:, :, 5] - st[:, :, 0])
    t0, t1 = st[:, :, 0].min(), st[:, :, 5].max()
    print(f"kernel span {t1 - t0:.2f} us; workgroups in flight on average {((st[:, 0, 5] - st[:, 0, 0]).sum()) / (t1 - t0):.1f} (of 256 CUs x resident)")
    res["span_us"] = float(t1 - t0)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
