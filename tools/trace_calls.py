"""Per-call durations of one kernel from a rocprofv3 kernel-trace CSV (the --stats average includes the warm-up launches,
which run slower: clock ramp, cold caches).  python3 tools/trace_calls.py <kernel_trace.csv> <kernel name substring>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(f"{sys.argv[2]}: {len(ms)} calls, ms each, in launch order:")
print(" ".join(f"{x:.3f}" for x in ms))
if len(ms) > 8:
    tail = ms[len(ms) // 2:]
    print(f"mean of all {sum(ms) / len(ms):.3f} ms; mean of the second half {sum(tail) / len(tail):.3f} ms")
