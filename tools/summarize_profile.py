#!/usr/bin/env python3
"""Condenses a gpurun_out/prof/<tag> directory (written by tools/profile.sh) into profiles/<name>.md and updates
profiles/traffic.json.  usage: tools/summarize_profile.py <prof dir> <name> <bench profile> <anchors per launch>"""
import collections
import csv
import glob
import json
import os
import sys

src, name, bench_profile, anchors = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_md = os.path.join(root, "profiles", name + ".md")

lines = [f"# rocprofv3 summary `{name}` (bench.py --profile {bench_profile}, {anchors} anchors per launch)", "",
         "Command: `tools/profile.sh` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 ...`, "
         "then one `rocprofv3 --pmc <set> --kernel-trace` run per counter set.", "", "## kernel stats (--kernel-trace --stats)", "",
         "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
for fn in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(fn)):
        if "mm2c::" in r["Name"]:
            short = r["Name"].split("(")[0].replace("void ", "")
            lines.append(f"| `{short}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['TotalDurationNs'])/1e6:.2f} | {r['Percentage']} |")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "mm2c::" not in k:
            continue
        kk = k.split("(")[0].replace("void ", "")
        agg[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines += ["", "## PMC counters, mean per launch", ""]
traffic = None
for k, d in sorted(agg.items()):
    lines += [f"### `{k}`", "", "| counter | mean per launch | per anchor |", "|---|---|---|"]
    for c, v in sorted(d.items()):
        # the same templated name is launched twice per step (main pass + flagged-task redo pass): report the big one
        v = sorted(v)[len(v) // 2:] if "chain_dp_wave" in k else v
        m = sum(v) / len(v)
        lines.append(f"| {c} | {m:.6g} | {m / anchors:.4g} |")
    if "chain_dp_wave" in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fs = sorted(d["FETCH_SIZE"])[len(d["FETCH_SIZE"]) // 2:]
        ws = sorted(d["WRITE_SIZE"])[len(d["WRITE_SIZE"]) // 2:]
        fetch_kb, write_kb = sum(fs) / len(fs), sum(ws) / len(ws)
        if traffic is not None and traffic["fetch_size_kb"] >= fetch_kb:
            lines.append("")
            continue
        traffic = {"fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                   "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024, "anchors_per_launch": anchors,
                   "note": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                           "WRITE_SIZE as reported; separate --pmc passes", "source": name}
    lines.append("")
if traffic:
    lines += ["## HBM traffic of chain_dp_wave", "",
              f"FETCH_SIZE {traffic['fetch_size_kb']:.4g} KB (x2 correction) + WRITE_SIZE {traffic['write_size_kb']:.4g} KB = "
              f"{traffic['hbm_bytes_per_launch']/1e9:.2f} GB per launch = {traffic['hbm_bytes_per_launch']/anchors:.1f} B per anchor "
              f"(algorithmic: 24 B per anchor)", ""]
    tj = os.path.join(root, "profiles", "traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    allt[bench_profile] = traffic
    json.dump(allt, open(tj, "w"), indent=1)
open(out_md, "w").write("\n".join(lines) + "\n")
print("wrote", out_md)
