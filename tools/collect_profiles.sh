#!/bin/bash
# Runs ON THE GPU BOX (via gpurun).  Collects for the bench command:
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/<tag>/stats
#   2. rocprofv3 --pmc FETCH_SIZE                  -> gpurun_out/<tag>/fetch   (separate pass)
#   3. rocprofv3 --pmc WRITE_SIZE                  -> gpurun_out/<tag>/write   (separate pass)
#   4. rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES -> .../valu
# (counter passes never combine with trace domains other than kernel-trace; see MI355X guide)
# The profiled passes run ONE encoder context (--contexts 1: kernels back to back), so that a
# kernel's average duration in the trace is its stand-alone launch time -- the figure bench.py's
# roofline uses; the final un-profiled bench line runs the default (three contexts in flight).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 20 --warmup 3 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/valu -- python3 $ROOT/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 0 > $OUT/valu.log 2>&1 || true
cd $ROOT && python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -30
