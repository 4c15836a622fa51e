#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the two cheap profile passes of the bench command
#   rocprofv3 --kernel-trace --stats                                      -> gpurun_out/<tag>/stats
#   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES -> gpurun_out/<tag>/valu
# (one encoder context, kernels back to back; counter pass separate from the trace pass).
# tools/summarize_profiles.py <tag> turns them into profiles/<tag>_*.  Extra args go to bench.py.
TAG=${1:-quick}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 20 --warmup 3 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 --no-other-configs "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/valu -- python3 $ROOT/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 --no-other-configs "$@" > $OUT/valu.log 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for f in glob.glob(out+"/stats/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/valu/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "SQ_INSTS_VALU" in v:
        n=sum(v["SQ_INSTS_VALU"])/len(v["SQ_INSTS_VALU"])
        w=sum(v["SQ_WAVES"])/len(v["SQ_WAVES"]) if "SQ_WAVES" in v else 0
        if n>1e6: print(k, "VALU insts/launch %.1fM"%(n/1e6), "waves %d"%w, "per wave %.0f"%(n/max(w,1)))
PY
