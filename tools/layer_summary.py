"""Groups the per-launch CSV of tools/gpu_layer_profile.py by layer class (resolution, Cout, taps, K)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
g = defaultdict(lambda: [0, 0.0, 0.0, [0.0] * 6])
for r in rows:
    k = (int(r["H"]), int(r["Cout"]), int(r["taps0"]), int(r["stride"]), int(r["up"]), int(r["K"]))
    g[k][0] += 1; g[k][1] += float(r["us"]); g[k][2] += float(r["gflop"])
    if r.get("wgs"):
        for i, c in enumerate(("wgs", "cyc_prologue", "cyc_staging", "cyc_kloop", "cyc_epilogue", "cyc_stats")):
            g[k][3][i] += float(r[c])
tot = sum(v[1] for v in g.values())
print(f"total {tot/1e3:.2f} ms")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
    n = v[0]
    line = f"H={k[0]:4d} Cout={k[1]:4d} K={k[5]:5d} taps={k[2]} s={k[3]} up={k[4]}  n={n:3d}  {v[1]/n:7.1f} us/launch ({100*v[1]/tot:4.1f}%)  {v[2]/max(v[1],1e-9)*1e3:6.1f} TF/s"
    if v[3][0] > 0:
        line += f"  wgs={v[3][0]/n:6.0f}  cycles/wg: pro {v[3][1]/n:6.0f} stage {v[3][2]/n:6.0f} kloop {v[3][3]/n:6.0f} epi {v[3][4]/n:6.0f} stats {v[3][5]/n:6.0f}"
    print(line)
