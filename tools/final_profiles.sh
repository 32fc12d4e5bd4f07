#!/bin/bash
# Refreshes the measurements behind profiles/ on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of
# the same command, PMC passes for the throughput kernel (B = 1024) and the latency kernels (B = 256: block, B = 512: quad), BASELINE configs.
# Outputs go to gpurun_out/final/ ; tools/collect_profiles.sh copies what should be judged into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/trace.err; echo "trace rc=$?"
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/trace -name "*kernel_trace.csv" -exec python3 tools/trace_calls.py {} blind_rotate_pair_kernel \; > $out/kernel_trace_pair_calls.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_block -- python3 bench.py --no-cpu-baseline --variant 3 --batch 256 > $out/bench_block_under_rocprof.json 2> $out/trace_block.err; echo "trace block rc=$?"
find $out/trace_block -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_block.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_quad -- python3 bench.py --no-cpu-baseline --batch 512 > $out/bench_quad_under_rocprof.json 2> $out/trace_quad.err; echo "trace quad rc=$?"
find $out/trace_quad -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_quad.csv \;
bash tools/prof_pmc.sh $out/pmc_pair > $out/pmc_pair.log 2>&1; echo "pmc pair rc=$?"
bash tools/prof_pmc.sh $out/pmc_quad --batch 512 > $out/pmc_quad.log 2>&1; echo "pmc quad rc=$?"
bash tools/prof_pmc.sh $out/pmc_block --variant 3 --batch 256 > $out/pmc_block.log 2>&1; echo "pmc block rc=$?"
python3 bench_configs.py > $out/bench_configs.jsonl 2> $out/bench_configs.err; echo "configs rc=$?"
rm -rf $out/trace $out/trace_block $out/trace_quad $out/pmc_pair/pmc_* $out/pmc_block/pmc_* $out/pmc_quad/pmc_*
ls -la $out
