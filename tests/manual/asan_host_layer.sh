#!/bin/bash
# Host-code sanitizer run (CPU only): the op-graph layer (host_ops.hpp / host_api.cpp) over the oracle backend,
# built with ASan + UBSan, through the host-op tests and the randomised cross-check.
set -e
cd "$(dirname "$0")/../.."
bak=$(mktemp)
cp tests/emu/libhost_emu.so "$bak" 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o tests/emu/libhost_emu.so \
    tfhe_rs_string_amd/csrc/host_api.cpp tests/emu/oracle_backend.cpp -Loracle -loracle -Wl,-rpath,"$PWD/oracle"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
    python -m pytest tests/test_host_ops_cpu.py tests/test_host_ops_fuzz.py tests/test_host_split_cpu.py -x -q -p no:cacheprovider
[ -s "$bak" ] && cp "$bak" tests/emu/libhost_emu.so && touch tests/emu/libhost_emu.so
rm -f "$bak"
