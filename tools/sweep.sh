#!/bin/bash
# usage: tools/sweep.sh "<flags1>" "<flags2>" ...   (on the GPU box) -- rebuilds with each flag set and benches
for f in "$@"; do
  TFHE_HIP_EXTRA_FLAGS="$f" python3 -c "from tfhe_rs_string_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== flags: $f"
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_ARGS | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('PBS/s %.0f  br_ms %.3f ks_ms %.3f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['keyswitch_avg_launch_ms']))"
done
