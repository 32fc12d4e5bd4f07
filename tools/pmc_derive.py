#!/usr/bin/env python3
"""profiles/pmc_derived.json from a PMC summary (tools/pmc_summary.py output) -- the static counter figures bench.py
quotes in `roofline` (valu_busy, instructions per CMUX step and wavefront, physical HBM bytes per launch).

  python3 tools/pmc_derive.py <summary.txt> <kernel substring> <batch> <lwe_dim> <waves per ciphertext> <source label> [out.json]

valu_busy = SQ_ACTIVE_INST_VALU / (SIMDs x launch duration in quad-cycles): both counters tick once per 4 clocks, so
SQ_BUSY_CYCLES-independent form: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD.
hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports
half of a wide coalesced read stream)."""
import json
import re
import sys

path, kernel, B, n, wpc, source = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
out = sys.argv[7] if len(sys.argv) > 7 else "profiles/pmc_derived.json"
vals, cur = {}, None
for line in open(path):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.eE+-]+)", line)
    if m and cur and kernel in cur:
        vals[m.group(1)] = float(m.group(2))
waves = B * wpc
steps = waves * n
waves_per_simd = 2  # both kernels run two wavefronts per SIMD
d = {
    "source": source,
    "kernel": kernel,
    "batch": B,
    "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
    "valu_busy": vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"] * waves_per_simd,
    "valu_insts_per_step_wave": vals["SQ_INSTS_VALU"] / steps,
    "lds_insts_per_step_wave": vals["SQ_INSTS_LDS"] / steps,
    "lds_bank_conflict_frac": vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_LDS_IDX_ACTIVE"], 1.0),
    "l2_hit_rate": vals["TCC_HIT_sum"] / max(vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"], 1.0),
    "raw": vals,
}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k != "raw"}, indent=1))
