#!/bin/bash
# Runs ON THE GPU BOX: kernel trace (start/end timestamps of every launch) of the default three-context
# bench, for a timeline analysis of how the batches' kernels overlap (tools/trace_overlap.py).
TAG=${1:-trace3}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 100 "$@" > $OUT/trace.log 2>&1
cd $ROOT
python3 tools/trace_overlap.py $OUT/trace
