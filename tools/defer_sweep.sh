for m in 12 16 24 32 48; do
  export FLACGPU_DEFER_MARGIN16=$m
  for run in "3 ar2" "3 hi" "5 hi" "5 ar2"; do
    cfg=${run%% *}; sig=${run##* }
    python3 bench.py --config $cfg --signal $sig --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-configs --sustained-steps 0 --detail /tmp/d.json > /dev/null 2>/tmp/err.txt
    python3 -c "
import json; d=json.load(open('/tmp/d.json')); a=d['analysis_stats']; print('margin16=$m', '$cfg$sig', 'step', d['ms_per_step'], 'cand', d['kernels']['k_cand64']['ms'], 'decided', a['fixed_count_decided_by_bound'], 'refetched', a['fixed_count_refetched'])"
  done
done
