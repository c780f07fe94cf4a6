#!/bin/bash
# build_variant.sh NAME [git-rev]: build libvcf2prot_hip.so of the working tree (or of a revision's csrc) into build_ab/NAME.so
# (V2P_DEFS="-DV2P_WAVE_CHECK" adds preprocessor definitions: development builds with checked gathers)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REV=${2:-}
SRC=$ROOT/vcf2prot_amd/csrc
if [ -n "$REV" ]; then
  TMP=$(mktemp -d); (cd $ROOT && git archive $REV vcf2prot_amd/csrc include | tar -x -C $TMP); SRC=$TMP/vcf2prot_amd/csrc
fi
mkdir -p $ROOT/build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result $V2P_DEFS $SRC/stitch_kernels.hip $( [ -f $SRC/stitch_wave.hip ] && echo $SRC/stitch_wave.hip ) $SRC/build_kernels.hip $SRC/v2p_api.hip $SRC/decode_kernels.hip $SRC/v2p_decode_api.hip -o $ROOT/build_ab/$NAME.so
echo built build_ab/$NAME.so
