#!/usr/bin/env python3
"""timeline of the last call in a rocprofv3 --kernel-trace --memory-copy-trace CSV output directory: every kernel and copy with start / end
relative to the first event of the last burst.  usage: trace_summary.py <dir>"""
import csv, glob, os, sys
ev = []
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-48:], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "")), "", r.get("Stream_Id", "")))
ev.sort()
# last burst: events after the last gap > 5 ms
cut = 0
for k in range(1, len(ev)):
    if ev[k][0] - max(e[1] for e in ev[max(0, k - 40):k]) > 5_000_000:
        cut = k
t0 = ev[cut][0]
for s, e, name, q, st in ev[cut:]:
    print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} ms  {(e - s) / 1e6:7.3f}  q={q} s={st} {name}")
