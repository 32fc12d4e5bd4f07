#!/bin/bash
for f in "$@"; do
  TFHE_HIP_EXTRA_FLAGS="$f" python3 -c "from tfhe_rs_string_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== flags: $f"; python3 tools/lat.py 2>&1 | grep "B="
done
