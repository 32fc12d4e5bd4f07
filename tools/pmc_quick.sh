#!/bin/bash
# one PMC pass (instruction counts / VALU activity) of the bench's kernels: tools/pmc_quick.sh <outdir> [bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc_quick -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_quick.log 2>&1
echo "rc=$?"
python3 tools/pmc_summary.py $out | grep -A9 "blind_rotate"
