"""Groups the per-launch CSV of tools/gpu_layer_profile.py by layer class (resolution, Cout, taps)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
g = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (int(r["H"]), int(r["Cout"]), int(r["taps0"]), int(r["stride"]), int(r["up"]))
    g[k][0] += 1; g[k][1] += float(r["us"]); g[k][2] += float(r["gflop"])
tot = sum(v[1] for v in g.values())
print(f"total {tot/1e3:.2f} ms  {sum(v[2] for v in g.values())/tot*1e-3*1e3:.1f} TF/s")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
    print(f"H={k[0]:4d} Cout={k[1]:4d} taps={k[2]} s={k[3]} up={k[4]}  n={v[0]:3d}  {v[1]/1e3:7.3f} ms ({100*v[1]/tot:4.1f}%)  {v[2]/max(v[1],1e-9)*1e-3:7.1f} TF/s")
