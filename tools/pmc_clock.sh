#!/bin/bash
# effective shader clock of a kernel in steady state: GRBM_GUI_ACTIVE (cycles per dispatch) against the HIP-event time
#   tools/pmc_clock.sh <outdir> "<variant> <batch>" ...
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for cfg in "$@"; do
  set -- $cfg
  v=$1; b=$2
  d=$out/clk_v${v}_b${b}
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d -- python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --variant $v --batch $b > $d.log 2>&1
  python3 - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "blind_rotate" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
ms = None
for line in open("$d.log"):
    if line.startswith("{"):
        ms = json.loads(line)["roofline"]["avg_launch_ms"]
g = acc.get("GRBM_GUI_ACTIVE", [0])
tail = g[len(g) // 2:]
w = acc.get("SQ_WAVE_CYCLES", [0])
print("variant $v B=$b: launch %.3f ms under the counters, GRBM_GUI_ACTIVE %.3e cycles/dispatch (last half) => %.2f GHz; SQ_WAVE_CYCLES x4 %.3e" % (ms, sum(tail) / len(tail), sum(tail) / len(tail) / ms / 1e6, 4 * sum(w) / len(w)))
PY
done
