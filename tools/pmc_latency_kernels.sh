#!/bin/bash
# PMC passes (instruction counts, LDS activity) of the latency kernels at their batch sizes; run on the GPU box:
#   tools/pmc_latency_kernels.sh gpurun_out/<dir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for cfg in "5 512" "4 512" "5 256" "3 256"; do
  set -- $cfg
  v=$1; b=$2
  d=$out/v${v}_b${b}
  mkdir -p $d
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d $d/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant $v --batch $b > $d/pmc_$tag.log 2>&1
    echo "v$v B=$b $tag rc=$?"
  done
  echo "== variant $v, B = $b" >> $out/summary.txt
  python3 tools/pmc_summary.py $d >> $out/summary.txt 2>&1
done
cat $out/summary.txt
