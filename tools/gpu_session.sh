#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): GPU parity tests, then the bench in the variants named on the
# command line ("VAR=val,VAR2=val2:--bench --flags" entries), everything into gpurun_out/<tag>/.
#   tools/gpu_session.sh <tag> [--no-tests] ["ENV=1:--contexts 1" ...]
TAG=${1:-session}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$1" == "--no-tests" ]; then shift; else
  timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
i=0
for spec in "$@"; do
  envs=${spec%%:*}; flags=${spec#*:}
  i=$((i+1))
  ( for kv in ${envs//,/ }; do [ -n "$kv" ] && export "$kv"; done
    timeout 600 python3 bench.py --detail $OUT/bench_$i.json $flags > $OUT/bench_line_$i.json 2> $OUT/bench_$i.err
    echo "[$i] $spec rc=$?" )
  python3 - "$OUT/bench_$i.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("  value", d["value"], "ms/step", d["ms_per_step"], {k:v["ms"] for k,v in d.get("kernels",{}).items()})
except Exception as e:
    print("  (no json)", e)
PY
done
