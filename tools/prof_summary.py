"""Summarise a rocprofv3 results .db: per-kernel time (kernel trace) and PMC sums.
    python tools/prof_summary.py <results.db> [out.md]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
out = []
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
out.append(f"total kernel time {tot:.1f} ms over {sum(r[1] for r in rows)} dispatches\n")
out.append("| kernel | calls | total ms | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|")
for r in rows[:25]:
    out.append(f"| `{r[0][:90]}` | {r[1]} | {r[2]:.1f} | {100*r[2]/tot:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} |")
try:
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    if cols:
        q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name"
        pm = list(cur.execute(q))
        if pm:
            out.append("\nPMC sums per kernel:\n\n| kernel | counter | sum | dispatches |\n|---|---|---|---|")
            for r in pm:
                out.append(f"| `{str(r[0])[:70]}` | {r[1]} | {r[2]:.6g} | {r[3]} |")
except Exception as ex:
    out.append(f"(no PMC table: {ex})")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
