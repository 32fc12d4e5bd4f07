#!/bin/bash
# Host-side sanitizer run (CPU only): the op-graph layer (host_api.cpp / host_ops.hpp / host_split.hpp) on the oracle-backed
# test backend under AddressSanitizer + UBSan, driven by the CPU op-graph tests.  tools/asan_host_ops.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
EMU=tests/emu
cp $EMU/libhost_emu.so /tmp/libhost_emu.so.keep 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o $EMU/libhost_emu.so \
    tfhe_rs_string_amd/csrc/host_api.cpp $EMU/oracle_backend.cpp -Loracle -loracle -Wl,-rpath,$PWD/oracle
touch $EMU/libhost_emu.so
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
set +e
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host_ops_cpu.py tests/test_host_split_cpu.py tests/test_num_blocks.py tests/test_host_ops_fuzz.py -x -q -m "not gpu" "$@"
rc=$?
rm -f $EMU/libhost_emu.so   # the next ordinary test run rebuilds the plain library
exit $rc
