#!/bin/bash
# A/B builds of the library for one gpurun session: tools/build_variant.sh <name> "<extra hipcc flags>"
#   -> gpurun_variants/<name>.so (git-ignored, travels to the GPU box); select with FLACENC_AMD_LIBRARY=gpurun_variants/<name>.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_variants
make -C $ROOT/flac-codec_amd/csrc -j8 BUILD=build_a_$NAME OUT=$ROOT/gpurun_variants/$NAME.so EXTRA="$*" 2>&1 | grep -E "error|ScratchSize \[bytes/lane\]: [1-9]" || true
ls -la $ROOT/gpurun_variants/$NAME.so
