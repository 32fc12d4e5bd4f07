#!/bin/bash
# A/B timing on ONE box, alternating: schedule v1 (9230f0f), schedule v2 (exp/fft-schedule-v2), main.  tools/ab_v1_v2.sh <outdir>
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
for rep in 1 2 3; do
  for t in v1tree v2tree main; do
    if [ $t = main ]; then cd $GRAFT_REPO_ROOT; else cd $GRAFT_REPO_ROOT/tools/exp/$t; fi
    python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 > $out/ab_${t}_$rep.json 2>/dev/null
    python3 -c "import json; j=json.load(open('$out/ab_${t}_$rep.json')); print('$t rep $rep: %.1f PBS/s, blind rotation %.3f ms' % (j['value'], j['roofline']['avg_launch_ms']))"
  done
done
