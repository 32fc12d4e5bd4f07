#!/bin/bash
# PMC before / after table of FFT schedule v2 on the throughput kernel.  tools/exp/v1tree = commit 9230f0f
# (schedule v1, before the integer identities), tools/exp/v2tree = branch exp/fft-schedule-v2 (same base + schedule v2), both
# built in the container and staged by hand (git-ignored).  Run on the GPU box: tools/pmc_v1_v2.sh gpurun_out/<dir>
out=$GRAFT_REPO_ROOT/$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $out
for t in v1tree v2tree; do
  cd $GRAFT_REPO_ROOT/tools/exp/$t
  python3 bench.py --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err; echo "$t bench rc=$?"
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d $out/$t/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$t/pmc_$tag.log 2>&1
    echo "$t $tag rc=$?"
  done
  echo "== $t" >> $out/summary.txt
  python3 -c "import json; j=json.load(open('$out/bench_$t.json')); print('   bench: %.1f PBS/s, blind rotation %.3f ms per 1024' % (j['value'], j['roofline']['avg_launch_ms']))" >> $out/summary.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/$t >> $out/summary.txt 2>&1
done
cat $out/summary.txt
