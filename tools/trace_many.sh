#!/bin/bash
# Runs ON THE GPU BOX: kernel + memory-copy timeline of flacenc_encode_many (64 streams), summarised:
# busy fractions of the copy engines and of the kernels inside the steady part of the run.
TAG=${1:-tmany}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/e2e_probe.py ${PROBE_CASE:-64,32,0,2} > $OUT/trace.log 2>&1
cd $ROOT
tail -2 $OUT/trace.log | cut -c1-160
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Name", "?")), int(r.get("Bytes", r.get("Size", 0)) or 0)))
for f in glob.glob(out + "/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = n[n.find("k_"):] if "k_" in n else n
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("<")[0].split("(")[0], 0))
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
lo, hi = t0 + (t1 - t0) * 6 // 10, t0 + (t1 - t0) * 9 // 10      # a late slice: steady calls
win = [e for e in ev if e[0] >= lo and e[1] <= hi]
span = hi - lo
agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n, b in win:
    agg[n][0] += e - s
    agg[n][1] += 1
    agg[n][2] += b
print(f"window {span / 1e6:.1f} ms")
for n, (d, c, b) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    extra = f"  {b / 1e6:.0f} MB  {b / max(d, 1):.1f} GB/s while active" if b else ""
    print(f"  {n:28s} busy {100 * d / span:6.1f}%  x{c:5d}  mean {d / c / 1e3:8.1f} us{extra}")
# union of kernel activity
ks = sorted((s, e) for s, e, n, b in win if not n.startswith("copy"))
busy, cur_s, cur_e = 0, None, None
for s, e in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
print(f"some kernel running {100 * busy / span:.1f}% of the window")
PY
