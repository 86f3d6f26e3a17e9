#!/usr/bin/env python3
"""A/B builds of the library that differ in ONE translation unit:  python tools/build_variant.py <name> <source.hip> [-DFLAG ...]
compiles <source.hip> with the flags, links it with the production objects of every other source (build/obj/prod, made by
__graft_entry__.build()) into pnpflow_amd/libpnpflow_hip_<name>.so, and runs the asm-load ISA audit when the unit is conv_dma.hip.
Select at run time with PNPFLOW_HIP_LIB=pnpflow_amd/libpnpflow_hip_<name>.so."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
G.build()
odir = os.path.join(G.OBJDIR, "var_" + name); os.makedirs(odir, exist_ok=True)
obj = os.path.join(odir, src.replace(".hip", ".o"))
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-save-temps=obj",
       "-Rpass-analysis=kernel-resource-usage"] + flags + ["-c", "-o", obj, os.path.join(G.CSRC, src)]
res = subprocess.run(cmd, cwd=odir, capture_output=True, text=True)
if res.returncode != 0:
    sys.stderr.write(res.stderr); sys.exit(1)
import re
for blk in re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]:
    fn = blk.split("\n")[0].split(" ")[0]
    if "conv_dma_kernel" not in fn and src == "conv_dma.hip":
        continue
    g = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"  {fn[:70]:70s} vgpr {g('VGPRs')} scratch {scr} occ {occ}")
objs = [os.path.join(G.OBJDIR, "prod", s.replace(".hip", ".o")) for s in G.SOURCES if s != src] + [obj]
lib = G.LIB.replace(".so", f"_{name}.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print("built", lib)
if src == "conv_dma.hip":
    asm = [f for f in os.listdir(odir) if f.endswith("gfx950.s")]
    sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit_asm_loads.py"), os.path.join(odir, asm[0])]).returncode)
