#!/bin/bash
# copies the outputs of tools/final_profiles.sh (gpurun_out/final) into profiles/ under this round's names
# and derives profiles/pmc_derived.json: tools/collect_profiles.sh r02
r=${1:-r03}
f=gpurun_out/final
cp $f/bench.json profiles/${r}_bench.json
cp $f/bench_under_rocprof.json profiles/${r}_bench_under_rocprof.json
cp $f/kernel_stats.csv profiles/${r}_kernel_stats.csv
cp $f/kernel_trace_pair_calls.txt profiles/${r}_kernel_trace_pair_calls.txt
cp $f/bench_block_under_rocprof.json profiles/${r}_bench_block_b256_under_rocprof.json
cp $f/kernel_stats_block.csv profiles/${r}_kernel_stats_block_b256.csv
cp $f/pmc_pair/summary.txt profiles/${r}_pmc_summary.txt
cp $f/pmc_block/summary.txt profiles/${r}_pmc_summary_block_b256.txt
cp $f/bench_quad_under_rocprof.json profiles/${r}_bench_quad_b512_under_rocprof.json
cp $f/kernel_stats_quad.csv profiles/${r}_kernel_stats_quad_b512.csv
cp $f/pmc_quad/summary.txt profiles/${r}_pmc_summary_quad_b512.txt
cp $f/bench_configs.jsonl profiles/${r}_bench_configs.jsonl
python3 tools/pmc_derive.py profiles/${r}_pmc_summary.txt blind_rotate_pair_kernel 1024 742 2 "profiles/${r}_pmc_summary.txt" profiles/pmc_derived.json
# static f64 instruction count of the CMUX loop (hipcc -S + tools/isa_loop_mix.py) next to the counters
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -std=c++17 -S --cuda-device-only \
    -o /tmp/pair_static.s tfhe_rs_string_amd/csrc/blind_rotate_pair.hip 2>/dev/null
python3 tools/isa_loop_mix.py /tmp/pair_static.s blind_rotate_pair_kernel > profiles/${r}_isa_loop_mix_pair.txt
python3 - <<PY
import json, re
mix = open("profiles/${r}_isa_loop_mix_pair.txt").read()
m = re.search(r"VALU (\d+): f64 (\d+)\s+int (\d+)\s+lane swaps (\d+)", mix)
d = json.load(open("profiles/pmc_derived.json"))
d["f64_insts_per_step_wave_static"] = int(m.group(2))
d["valu_insts_per_step_wave_static"] = int(m.group(1))
d["static_source"] = "profiles/${r}_isa_loop_mix_pair.txt (hipcc -S, hot path of the CMUX loop)"
json.dump(d, open("profiles/pmc_derived.json", "w"), indent=1)
print("static:", m.groups())
PY
