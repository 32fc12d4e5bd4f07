#!/bin/bash
# Host-side sanitizer run (CPU only) of the C-ABI library's host code: api.cpp, wire.cpp, host_api.cpp, twiddle.cpp built with
# AddressSanitizer + UBSan (device objects unchanged), driven by the CPU tests that need no GPU.  tools/asan_cabi.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
OBJ=tfhe_rs_string_amd/csrc/obj
TMP=/tmp/asan_cabi_$$
mkdir -p $TMP
FLAGS="-O1 -g -fPIC -fvisibility=hidden -ffp-contract=off -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer"
for f in api.cpp wire.cpp host_api.cpp; do /opt/rocm/bin/hipcc $FLAGS -Wno-unused-value -Wno-option-ignored -c tfhe_rs_string_amd/csrc/$f -o $TMP/$f.o; done
g++ $FLAGS -c tfhe_rs_string_amd/csrc/twiddle.cpp -o $TMP/twiddle.cpp.o
OTHERS=$(ls $OBJ/*.o | grep -v "/api.cpp.o\|/wire.cpp.o\|/host_api.cpp.o\|/twiddle.cpp.o")
/opt/rocm/bin/hipcc -shared -fsanitize=address,undefined -o $TMP/libtfhe_hip_asan.so $TMP/*.o $OTHERS -fvisibility=hidden
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
set +e
LD_PRELOAD="$ASAN" TFHE_HIP_LIB=$TMP/libtfhe_hip_asan.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_seeded_keys_cpu.py tests/test_standard_keys.py tests/test_cabi_cpu.py -x -q -m "not gpu" "$@"
rc=$?
rm -rf $TMP
exit $rc
