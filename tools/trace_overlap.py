#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run of the multi-context bench: how much of the wall time has
0 / 1 / 2 / 3+ kernels in flight, per-kernel durations when overlapped, and the steady-state time per
batch.  usage: trace_overlap.py <dir with *kernel_trace.csv>"""
import collections
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:24]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
# steady state: the k_frame64 launches in the middle of the run
fr = [r for r in rows if r[2] == "k_frame64"]
if len(fr) < 40:
    print("too few k_frame64 launches", len(fr)); sys.exit(0)
lo, hi = fr[len(fr) // 4][1], fr[3 * len(fr) // 4][1]
nb = 3 * len(fr) // 4 - len(fr) // 4
print(f"steady window: {nb} batches in {(hi - lo) / 1e6:.3f} ms -> {(hi - lo) / nb / 1e6:.4f} ms per batch")
win = [r for r in rows if r[1] > lo and r[0] < hi]
ev = []
for s, e, n, q in win:
    ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
ev.sort()
depth, last, hist = 0, lo, collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
tot = hi - lo
print("kernels in flight:", {k: f"{100 * v / tot:.1f}%" for k, v in sorted(hist.items())})
dur = collections.defaultdict(list)
for s, e, n, q in win:
    dur[n].append(e - s)
print("per-kernel mean duration inside the window (ms), launches, share of wall:")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:24s} {sum(v) / len(v) / 1e6:.4f}  x{len(v):4d}  {100 * sum(v) / tot:.1f}%")
# which kernels run while k_cand64 runs
co = collections.Counter()
cands = [r for r in win if r[2].startswith("k_cand64")]
for s, e, n, q in cands:
    for s2, e2, n2, q2 in win:
        if not n2.startswith("k_cand64") or (s2, e2) != (s, e):
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                co[n2] += ov
tc = sum(e - s for s, e, n, q in cands)
print("overlap with k_cand64 (fraction of k_cand64 time):", {k: f"{v / tc:.2f}" for k, v in co.most_common(6)})
