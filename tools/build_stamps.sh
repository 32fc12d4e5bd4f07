#!/bin/bash
# DIAGNOSTIC: build a copy of the library whose blind-rotation kernel of one file carries per-phase clock stamps
#   tools/build_stamps.sh blind_rotate_quad.hip TFHE_QUAD_STAMPS tools/exp/libtfhe_quad_stamps.so
# (run with TFHE_HIP_LIB=<that .so>; never shipped -- the stamped kernel overwrites part of its output)
set -e
cd "$(dirname "$0")/.."
SRC=$1; DEF=$2; OUT=$3
OBJ=tfhe_rs_string_amd/csrc/obj
mkdir -p tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -std=c++17 -D$DEF ${EXTRA_DEFS} \
    -c tfhe_rs_string_amd/csrc/$SRC -o /tmp/stamps_$$.o
OTHERS=$(ls $OBJ/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -o $OUT $OTHERS /tmp/stamps_$$.o -fvisibility=hidden
rm -f /tmp/stamps_$$.o
echo built $OUT
