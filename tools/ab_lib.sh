#!/bin/bash
# A/B timing on ONE box, alternating: the library in the tree against another build (TFHE_HIP_LIB).  tools/ab_lib.sh <other.so> [bench args]
other=$1; shift
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for which in tree other; do
    if [ $which = other ]; then export TFHE_HIP_LIB=$GRAFT_REPO_ROOT/$other; else unset TFHE_HIP_LIB; fi
    python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" > /tmp/ab_$which.json 2>/dev/null
    python3 -c "import json; j=json.load(open('/tmp/ab_$which.json')); print('$which rep $rep: %.1f PBS/s, blind rotation %.3f ms' % (j['value'], j['roofline']['avg_launch_ms']))"
  done
done
