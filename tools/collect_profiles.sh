#!/bin/bash
# Runs ON THE GPU BOX (via gpurun).  Collects for the bench command (extra arguments, e.g. `--config 4`, go to bench.py):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/<tag>/stats
#   2. rocprofv3 --pmc FETCH_SIZE                  -> gpurun_out/<tag>/fetch   (separate pass)
#   3. rocprofv3 --pmc WRITE_SIZE                  -> gpurun_out/<tag>/write   (separate pass)
#   4. rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES -> .../valu
# (counter passes never combine with trace domains other than kernel-trace; see MI355X guide)
# The profiled passes run ONE encoder context (--contexts 1: kernels back to back), so that a
# kernel's average duration in the trace is its stand-alone launch time -- the figure bench.py's
# roofline uses; with FULL=1 a final un-profiled default bench line is added.
#   tools/collect_profiles.sh <tag> [bench.py arguments]; tools/summarize_profiles.py <tag> afterwards
set -e
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
COMMON="--contexts 1 --no-cpu-baseline --no-end-to-end --sustained-steps 0 --prewarm-ms 200 --no-other-configs"
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from flac_codec_amd import _lib; print(_lib.build_id())" > $OUT/build_id.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 20 --warmup 3 $COMMON "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 $COMMON "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 3 --warmup 1 $COMMON "$@" > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/valu -- python3 $ROOT/bench.py --steps 3 --warmup 1 $COMMON "$@" > $OUT/valu.log 2>&1 || true
cd $ROOT
# the full record goes to bench.json (what summarize_profiles.py copies), the compact driver line to bench_line.json
if [ -n "$FULL" ]; then python3 bench.py --steps 20 --warmup 5 --detail $OUT/bench.json "$@" > $OUT/bench_line.json 2> $OUT/bench.err; fi
ls $OUT
