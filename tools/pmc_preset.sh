#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --pmc of tools/preset_probe.py <preset> (VALU instructions / waves per launch and
# kernel), once on the default path and once with FLACGPU_NO_DIRECT_SHORT=1.   tools/pmc_preset.sh <tag> <preset>
TAG=${1:-pmc_preset}; PRESET=${2:-fast}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/direct -- python3 $ROOT/tools/preset_probe.py $PRESET > $OUT/direct.log 2>&1
FLACGPU_NO_DIRECT_SHORT=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/split -- python3 $ROOT/tools/preset_probe.py $PRESET > $OUT/split.log 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for mode in ("direct","split"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out+"/"+mode+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(mode, k, {c:round(sum(x)/len(x)) for c,x in v.items()}, len(next(iter(v.values()))))
PY
