#!/bin/bash
# A/B timing on ONE box: the library in the tree and any number of other builds, alternating, three rounds.
#   tools/ab_libs.sh "<bench args>" lib1.so lib2.so ...
args=$1; shift
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  unset TFHE_HIP_LIB
  python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 $args > /tmp/ab.json 2>/dev/null
  python3 -c "import json; j=json.load(open('/tmp/ab.json')); print('tree rep $rep: %.1f PBS/s, blind rotation %.3f ms' % (j['value'], j['roofline']['avg_launch_ms']))"
  for lib in "$@"; do
    TFHE_HIP_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 $args > /tmp/ab.json 2>/dev/null
    python3 -c "import json; j=json.load(open('/tmp/ab.json')); print('$lib rep $rep: %.1f PBS/s, blind rotation %.3f ms' % (j['value'], j['roofline']['avg_launch_ms']))"
  done
done
