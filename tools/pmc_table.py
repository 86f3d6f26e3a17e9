#!/usr/bin/env python3
"""Per-kernel busy fractions from the three counter passes of tools/pmc_level0.sh:  python tools/pmc_table.py gpurun_out/pmc_level0 [filter]
Busy fraction = counter / (units x kernel time x shader clock); the clock of a kernel = GRBM_GUI_ACTIVE / 8 XCDs / its time in the same
pass; units: 1024 SIMDs (SQ_VALU_MFMA_BUSY_CYCLES), 256 CUs (TA_TA_BUSY_sum, SQ_LDS_IDX_ACTIVE, TCP_PENDING_STALL_CYCLES_sum); the wave
columns are SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES (all quad-cycles)."""
import re, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "conv_"
def load(p):
    t, c = {}, {}
    for l in open(p):
        m = re.match(r"\| `(.*?)` \| (\d+) \| ([0-9.]+) \| [0-9.]+ \| ([0-9.]+) \|", l)
        if m: t[m.group(1)] = (int(m.group(2)), float(m.group(3)) * 1e-3)
        m = re.match(r"\| `(.*?)` \| ([A-Za-z_]+) \| ([0-9.e+]+) \| (\d+) \|", l)
        if m: c.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
    # (the counter rows carry kernel names cut at 80 characters)
    c = {full: v for full in t for short, v in c.items() if full.startswith(short)}
    return t, c
(t1, c1), (t2, c2), (t3, c3) = (load(f"{d}/pass{i}.md") for i in (1, 2, 3))
print("| kernel (launches) | avg us | clock GHz | matrix pipe busy | texture addresser busy | LDS busy | L1 stalled on pending | wave issuing / issue-stalled / parked | per wave: VALU / SALU / LDS / vector-memory reads |")
print("|---|---|---|---|---|---|---|---|---|")
for k in sorted(t3, key=lambda k: -t3[k][1]):
    if flt not in k or k not in c1 or k not in c2 or k not in c3: continue
    n, s3 = t3[k]; s1 = t1[k][1]; s2 = t2[k][1]
    clk = c3[k]["GRBM_GUI_ACTIVE"] / 8 / s3
    mf = c1[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * s1 * clk)
    ta = c3[k]["TA_TA_BUSY_sum"] / (256 * s3 * clk); l1 = c3[k]["TCP_PENDING_STALL_CYCLES_sum"] / (256 * s3 * clk)
    lds = c2[k]["SQ_LDS_IDX_ACTIVE"] / (256 * s2 * clk)
    wc = c1[k]["SQ_WAVE_CYCLES"]; w = c2[k]["SQ_WAVES"]
    print(f"| `{k[9:70]}` ({n}) | {s3 / n * 1e6:.1f} | {clk / 1e9:.2f} | {mf:.3f} | {ta:.3f} | {lds:.3f} | {l1:.3f} | {c1[k]['SQ_ACTIVE_INST_ANY'] / wc:.2f} / {c1[k]['SQ_WAIT_INST_ANY'] / wc:.2f} / {c1[k]['SQ_WAIT_ANY'] / wc:.2f} | "
          f"{c2[k]['SQ_INSTS_VALU'] / w:.0f} / {c2[k]['SQ_INSTS_SALU'] / w:.0f} / {c2[k]['SQ_INSTS_LDS'] / w:.0f} / {c2[k]['SQ_INSTS_VMEM_RD'] / w:.0f} |")
