#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass of the one-context bench per counter group given
# ("A B C" "D E" ...), results summarised per kernel.   tools/pmc_pass.sh <tag> "CTR1 CTR2" ...
TAG=${1:-pmc}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 --no-other-configs $PMC_BENCH_ARGS > $OUT/g$i.log 2>&1
done
cd $ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:58]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if any(x in k for x in ("k_cand64","k_autocorr4","k_frame64","k_deinterleave2","k_sub64")) and "mfma" not in k:
        print(k, {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
