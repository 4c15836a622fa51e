#!/bin/bash
# One gpurun call = one box: probe it, and only where the headline step runs under 0.37 ms (the pool's faster boxes) collect the
# round's profiles and two default bench runs there.  usage (GPU box): bash tools/collect_if_fast.sh <tag-prefix> <bench-dir>
cd $GRAFT_REPO_ROOT
P=$(timeout 200 python bench.py --no-cpu-baseline --no-other-configs --no-end-to-end --steps 50 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['extra'].get('sclk_MHz'))")
echo "probe: $P"
MS=$(echo $P | cut -d' ' -f1)
if python3 -c "import sys; sys.exit(0 if float('$MS') < 0.37 else 1)"; then
  FULL=1 bash tools/collect_profiles.sh ${1}_b > /dev/null 2>&1
  bash tools/collect_profiles.sh ${1}_hi --signal hi > /dev/null 2>&1
  bash tools/collect_profiles.sh ${1}_cfg2 --config 2 > /dev/null 2>&1
  bash tools/collect_profiles.sh ${1}_cfg4 --config 4 > /dev/null 2>&1
  bash tools/collect_profiles.sh ${1}_cfg5 --config 5 > /dev/null 2>&1
  bash tools/collect_profiles.sh ${1}_cfg5hi --config 5 --signal hi > /dev/null 2>&1
  mkdir -p gpurun_out/$2
  for i in 1 2; do timeout 900 python bench.py --detail gpurun_out/$2/bench$i.json > gpurun_out/$2/line$i.json 2> gpurun_out/$2/err$i.txt; done
  echo "collected on a fast box: $(cat gpurun_out/${1}_b/build_id.txt)"
else
  echo "slow box: nothing collected"
fi
