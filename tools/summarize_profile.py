"""Condenses gpurun_out/prof_<tag>/ (bench.json, rocprofv3 kernel stats, PMC passes) into profiles/<tag>_*."""
import csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(R, "gpurun_out", "prof_" + tag)
dst = os.path.join(R, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]

rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline (MI355X)\n")
    f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for r in rows:
        f.write("%s,%s,%s,%.0f,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                             r["MinNs"], r["MaxNs"], r["Percentage"]))
pmc = {}
for d, cn in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = {}
    for r in csv.DictReader(open(os.path.join(src, d, "bench_counter_collection.csv"))):
        agg.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    pmc[cn] = {k: {"launches": len(v), "mean_kb": sum(v) / len(v)} for k, v in agg.items() if k.startswith("k_")}
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
key = [k for k in pmc["FETCH_SIZE"] if k.startswith("k_ao_rays<false")][0]
fetch_kb = pmc["FETCH_SIZE"][key]["mean_kb"]
write_kb = pmc["WRITE_SIZE"][key]["mean_kb"]
# MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
# 16-B-per-lane loads (the node / segment fetches are global_load_dwordx4) -> doubled; WRITE_SIZE uncorrected.
traffic = int((2.0 * fetch_kb + write_kb) * 1024)
out = {"tag": tag, "kernel": "k_ao_rays", "FETCH_SIZE_kib_per_launch": fetch_kb, "WRITE_SIZE_kib_per_launch": write_kb,
       "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane loads)",
       "k_ao_rays_hbm_bytes_per_launch": traffic,
       "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
       "collected_with": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (separate passes), bench.py --steps 3 --warmup 1",
       "all_kernels": pmc}
json.dump(out, open(os.path.join(dst, "traffic_%s.json" % tag), "w"), indent=1)
bench["roofline"]["traffic"] = traffic  # same run as the kernel stats above
json.dump(bench, open(os.path.join(dst, "bench_%s.json" % tag), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("k_ao_rays_hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}))
for r in rows[:6]:
    print(short(r["Name"]), r["Calls"], "%.3f ms" % (float(r["AverageNs"]) / 1e6), r["Percentage"] + "%")
