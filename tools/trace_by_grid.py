"""Per-(kernel, grid) durations from a rocprofv3 --kernel-trace --output-format csv run: which launches of one kernel (= which
U-Net level) the time goes to.
    python tools/trace_by_grid.py <dir with *_kernel_trace.csv> <kernel-name substring> [out.md]"""
import csv, glob, os, sys
from collections import defaultdict
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
pat = sys.argv[2]
agg = defaultdict(list)
total = 0.0
for f in files:
    for row in csv.DictReader(open(f)):
        d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        total += d
        if pat not in row["Kernel_Name"]: continue
        name = row["Kernel_Name"]
        short = name[name.find(pat):][:60]
        tmpl = name[name.find("<"):name.find(">") + 1] if "<" in name else ""
        agg[(short.split("(")[0], tmpl[:40], int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"])), int(row["Grid_Size_Y"]), int(row["Grid_Size_Z"]))].append(d)
out = [f"all kernels: {total / 1e3:.1f} ms; launches matching '{pat}': {sum(len(v) for v in agg.values())}, {sum(sum(v) for v in agg.values()) / 1e3:.2f} ms\n",
       "| kernel | template | workgroups x | y | z | launches | total ms | avg us | min us | max us |", "|---|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    out.append(f"| {k[0]} | `{k[1]}` | {k[2]} | {k[3]} | {k[4]} | {len(v)} | {sum(v) / 1e3:.2f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 3: open(sys.argv[3], "w").write(txt + "\n")
