"""Copy the files of one `bash tools/r06_final.sh` + `bash tools/pmc_traffic.sh` call from gpurun_out/ into profiles/ under a round tag and print the figures
the docs quote.   python tools/collect_evidence.py r06"""
import json, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(R)
cp = {"gpurun_out/r06f/bench_default.json": f"profiles/{tag}_bench_default.json", "gpurun_out/r06f/pmc_conv_counters.md": f"profiles/{tag}_pmc_conv_counters.md",
      "gpurun_out/r06f/l256_160.txt": f"profiles/{tag}_layers_256_160.txt", "gpurun_out/r06f/l128_160.txt": f"profiles/{tag}_layers_128_160.txt",
      "gpurun_out/r06f/mfma_f16_chain.json": f"profiles/{tag}_mfma_f16_chain.json"}
for w in ("c2_256", "c2", "c5"):
    cp[f"gpurun_out/prof_{tag}/kernel_trace_bench_{w}.md"] = f"profiles/{tag}_kernel_trace_bench_{w}.md"
for n in ("c2_256", "c2", "c3", "c4"):
    for k in ("fetch", "write"):
        cp[f"gpurun_out/pmc/{n}_{k}.md"] = f"profiles/{tag}_pmc_{n}_{k}.md"
cp["gpurun_out/pmc/traffic.json"] = "profiles/traffic.json"
for a, b in cp.items():
    if os.path.isfile(a): shutil.copy(a, b)
    else: print("MISSING", a)
open(f"profiles/{tag}_gputests_final.txt", "w").write("".join(open("gpurun_out/r06f/pytest.log").readlines()[-5:]))
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
t = open("profiles/traffic.json").read().replace('"commit": ""', f'"commit": "{commit}"')
open("profiles/traffic.json", "w").write(t)
d = json.loads(open(f"profiles/{tag}_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], d["power"]["board_power_w_mean"], d["power"]["board_power_w_p95"], d["power"]["sclk_ghz_mean"])
print({k: r[k] for k in ("bound", "achieved", "frac", "traffic", "avg_launch_us", "frac_of_box_ceiling", "share_of_conv_time")})
for k, v in r["classes"].items(): print(" ", k, v["share"], v["algorithmic_tflops"], v["frac"], v["executed_mfma_frac"], v["frac_of_box_ceiling"])
print("mfma probe", r["mfma_ceiling"]["tflops_f16_dense"], "family of box", r["mfma_ceiling"]["family_executed_frac_of_box_ceiling"], "alg", r["mfma_family"]["achieved"], "exec", r["mfma_family"]["mfma_tflops_executed"])
print("stream", r["streaming_ceiling"]["detail"])
print("cpu", d["cpu_baseline"]["value"], "e2e TF", d.get("unet_tflops_end_to_end"))
print({k: (v.get("images_per_s") if isinstance(v, dict) else v) for k, v in d["configs"].items()})
for w in ("c2_256", "c2", "c5"):
    print(w, open(f"profiles/{tag}_kernel_trace_bench_{w}.md").readline().strip())
