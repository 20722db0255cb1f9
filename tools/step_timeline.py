#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: kernels in start order with their durations and the idle gap before
each (host latency between launches).  usage: python tools/step_timeline.py <kernel_trace.csv> [sweep-kernel substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "sweep_blk_kernel<16, false>"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("need at least three steps in the trace")
a, b = marks[-2], marks[-1]                      # one full step: from the end of the second-to-last big kernel to the end of the last
prev_end = int(rows[a]["End_Timestamp"])
tot_gap = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    tot_gap += max(gap, 0)
    print(f"gap {gap / 1e3:8.1f} us | {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"idle between kernels in this step: {tot_gap / 1e3:.1f} us")
