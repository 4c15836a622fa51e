#!/bin/bash
# Runs ON THE GPU BOX: the bench's kernel times and four-context step for a list of library variants
# (tools/build_variant.sh), config 3 on both signals and config 5 on its high-order input, `REPS` times each.
#   tools/ab_variants.sh <tag> "<name>[:ENV=val,ENV2=val]" ...     (name "default" = the in-tree library)
#   RUNS="3:ar2 3:hi 5:hi" (default) / MORE_RUNS="2:ar2 4:ar2": config:signal pairs
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for rep in $(seq 1 ${REPS:-2}); do
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
  for run in ${RUNS:-3:ar2 3:hi 5:hi} ${MORE_RUNS}; do   # cfg:signal
    cfg=${run%%:*}; sig=${run##*:}
    ( [ "$name" != default ] && export FLACENC_AMD_LIBRARY=$ROOT/gpurun_variants/$name.so
      for kv in ${envs//,/ }; do export "$kv"; done
      timeout 300 python3 bench.py --config $cfg --signal $sig --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end \
        --no-other-configs --sustained-steps 0 --detail $OUT/${name}_${cfg}${sig}_$rep.json > /dev/null 2> $OUT/${name}_${cfg}${sig}_$rep.err || echo "FAILED $spec $run" )
    python3 - "$OUT/${name}_${cfg}${sig}_$rep.json" "$spec" "$cfg$sig" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:28s} {sys.argv[3]:6s} step {d['ms_per_step']:.4f} one-ctx {d['variants']['one_context_back_to_back']['ms_per_step']:.4f}", {k:v['ms'] for k,v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[2], sys.argv[3], "no result", e)
PY
  done
done
done 2>&1 | tee $OUT/summary.txt
