#!/usr/bin/env python3
"""Where a persistent candidate workgroup's time goes (k_cand64p; a library built with -DCANDP_TIMELINE=1):
    tools/build_variant.sh timeline "-DCANDP_TIMELINE=1"
    FLACENC_AMD_LIBRARY=gpurun_variants/timeline.so python3 tools/cand_timeline.py [--signal ar2|hi] [--config 3] [--out f.json]
Lane 0 of every wave leaves 100 MHz stamps of its turn's phases in the (consumed) autocorrelation row of its
candidate; this script runs ONE context back to back, reads the rows of the last batch and prints, per phase, the
mean / median / p90 duration, the spread of a frame's four waves at the closing barrier and the turn lengths.
Stamps: 0 turn top, 1 image landed (barrier), 2 samples in registers, 3 order statistics + FIXED tree, 4 exact FIXED count,
5 FIR, 6 analysis done, 7 the frame's waves met (barrier), 8 turn end; [13] XCC_ID, [14] turn index, [15] HW_ID."""
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
    t = rows.cpu().numpy().reshape(F, NC, 36)
    st = t[:, :, :9].astype(np.float64) * 0.01   # microseconds
    turn = t[:, :, 14] & 0xFFFFFFFF
    wg = t[:, :, 14] >> 32
    hw = t[:, :, 15]
    xcc = t[:, :, 13]
    ok = (t[:, :, 0] > 0) & (t[:, :, 8] >= t[:, :, 0])
    print(f"candidates with stamps: {ok.sum()} of {F * NC}")
    res = {"signal": a.signal, "config": a.config, "frames": F}

    def q(name, v):
        v = v[np.isfinite(v)]
        r = {"mean": float(v.mean()), "p50": float(np.median(v)), "p90": float(np.percentile(v, 90)), "max": float(v.max())}
        print(f"{name:44s} mean {r['mean']:7.2f}  p50 {r['p50']:7.2f}  p90 {r['p90']:7.2f}  max {r['max']:7.2f} us")
        res[name] = r

    m = ok
    q("0-1 wait for the image (barrier 1)", (st[:, :, 1] - st[:, :, 0])[m])
    q("1-2 samples LDS -> registers", (st[:, :, 2] - st[:, :, 1])[m])
    has3 = m & (t[:, :, 3] > 0)
    q("2-3 order statistics + FIXED tree", (st[:, :, 3] - st[:, :, 2])[has3])
    has4 = has3 & (t[:, :, 4] > 0)
    q("3-4 exact FIXED count", (st[:, :, 4] - st[:, :, 3])[has4])
    has5 = has4 & (t[:, :, 5] > 0)
    q("4-5 FIR", (st[:, :, 5] - st[:, :, 4])[has5])
    q("5-6 LPC fold + tree + exact", (st[:, :, 6] - st[:, :, 5])[has5])
    q("2-6 analysis (whole body)", (st[:, :, 6] - st[:, :, 2])[m])
    q("6-7 wait for the frame's other waves (barrier 2)", (st[:, :, 7] - st[:, :, 6])[m])
    q("7-8 request + plan store", (st[:, :, 8] - st[:, :, 7])[m])
    q("0-8 turn", (st[:, :, 8] - st[:, :, 0])[m])
    # per frame: slowest - fastest body
    body = st[:, :, 6] - st[:, :, 2]
    allm = m.all(axis=1)
    q("per frame: slowest - fastest wave's analysis", (body.max(axis=1) - body.min(axis=1))[allm])
    q("per frame: slowest wave's analysis", body.max(axis=1)[allm])
    q("per frame: mean wave's analysis", body.mean(axis=1)[allm])
    # kernel span, turns per workgroup
    t0, t1 = st[:, :, 0][m].min(), st[:, :, 8][m].max()
    print(f"kernel span (first turn top to last turn end): {t1 - t0:.2f} us; turns per workgroup: {int(turn[m].max()) + 1}")
    res["span_us"] = float(t1 - t0)
    for k in range(int(turn[m].max()) + 1):
        mk = m & (turn == k)
        if mk.any():
            print(f"  turn {k}: starts {st[:, :, 0][mk].min() - t0:7.2f} .. {st[:, :, 0][mk].max() - t0:7.2f}, ends .. {st[:, :, 8][mk].max() - t0:7.2f};"
                  f" mean length {(st[:, :, 8] - st[:, :, 0])[mk].mean():6.2f}")
    # per workgroup: turns taken, first turn top, last turn end
    wgs = wg[:, 0][allm].astype(np.int64)
    nt = np.bincount(wgs)
    first = np.full(nt.size, np.inf)
    last = np.zeros(nt.size)
    np.minimum.at(first, wgs, st[:, 0, 0][allm])
    np.maximum.at(last, wgs, st[:, :, 8].max(axis=1)[allm])
    live = nt > 0
    q("per workgroup: turns taken", nt[live].astype(np.float64))
    q("per workgroup: first turn top (after the kernel's first)", first[live] - t0)
    q("per workgroup: last turn end (before the kernel's last)", t1 - last[live])
    res["turns_hist"] = {int(k): int(v) for k, v in zip(*np.unique(nt[live], return_counts=True))}
    print("turns per workgroup:", res["turns_hist"])
    # the frames that ended last
    ends = st[:, :, 8].max(axis=1)
    order = np.argsort(-np.where(allm, ends, -np.inf))[:12]
    print("the last frames to end (us before the kernel's end: turn top, phases 0-1 1-2 2-6 6-7 7-8 of the slowest wave; workgroup, its turn):")
    for f in order:
        wv = int(np.argmax(st[f, :, 6] - st[f, :, 2]))
        d = st[f, wv]
        print(f"  frame {int(f):5d}: top {t1 - d[0]:7.2f} end {t1 - ends[f]:6.2f} | {d[1] - d[0]:6.2f} {d[2] - d[1]:6.2f} {d[6] - d[2]:6.2f} {d[7] - d[6]:6.2f} {d[8] - d[7]:6.2f}"
              f" | 2-3 {d[3] - d[2] if t[f, wv, 3] else -1:6.2f} 3-4 {d[4] - d[3] if t[f, wv, 4] else -1:6.2f} 4-5 {d[5] - d[4] if t[f, wv, 5] else -1:6.2f}"
              f" | wg {int(wg[f, 0])} turn {int(turn[f, 0])}")
    # the longest waits at barrier 1: all four waves of the frame, and of the frame the workgroup had before
    b1 = np.where(allm[:, None], st[:, :, 1] - st[:, :, 0], -1).max(axis=1)
    wg0 = wg[:, 0]
    for f in np.argsort(-b1)[:4]:
        prev = [int(x) for x in np.nonzero((wg0 == wg0[f]) & (turn[:, 0] == turn[f, 0] - 1))[0]]
        for ff, tag in [(p_, "previous") for p_ in prev] + [(int(f), "this")]:
            print(f"  wg {int(wg0[ff])} turn {int(turn[ff, 0])} frame {ff} ({tag}); stamps in us before the kernel's end, per wave (simd):")
            for wv in range(4):
                print("     wave", wv, "simd", int((hw[ff, wv] >> 4) & 3), " ".join(f"{t1 - st[ff, wv, k]:7.2f}" if t[ff, wv, k] else "      -" for k in range(9)))
    # placement: waves per (xcc, se, cu, simd)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    key = ((xcc & 15) << 12) | (se << 8) | (cu << 4) | simd
    k0 = key[m & (turn == 0)]   # (a workgroup's first turn)
    u, cnt = np.unique(k0, return_counts=True)
    print(f"turn 0: {len(u)} distinct (xcc, se, cu, simd); waves per SIMD: min {cnt.min()} max {cnt.max()} mean {cnt.mean():.2f}")
    # do the four waves of a frame sit on four different SIMDs?
    fs = simd[allm]
    distinct = np.array([len(set(r)) for r in fs])
    print("distinct SIMDs among a frame's four waves:", {int(v): int((distinct == v).sum()) for v in np.unique(distinct)})
    res["waves_per_simd_turn0"] = {"min": int(cnt.min()), "max": int(cnt.max()), "mean": float(cnt.mean()), "simds": int(len(u))}
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
