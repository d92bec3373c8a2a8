#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/* (rocprofv3 csv, written by profiles/collect.sh on the GPU box) into the committed summaries:
<tag>_kernel_stats.csv, <tag>_pmc_summary.json, <tag>_vitl_kernel_stats.csv, <tag>_tune_kernel_stats.csv and the bench JSON lines.
    python profiles/summarize.py r02"""
import collections, csv, glob, json, os, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
DST = os.path.join(ROOT, "profiles")
GEMM = "gemm_tn_256x256x64_pp"


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def counters(sub):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(newest(os.path.join(SRC, sub, "*", "*_counter_collection.csv")))):
        per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return per


def by_kernel(per, match):
    agg = collections.defaultdict(list)
    for (name, _), c in per.items():
        if match in name:
            for k, v in c.items():
                agg[k].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}, max((len(v) for v in agg.values()), default=0)


def stats(sub, out_name):
    rows = list(csv.DictReader(open(newest(os.path.join(SRC, sub, "*", "*_kernel_stats.csv")))))
    with open(os.path.join(DST, out_name), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    return rows


rows = stats("trace", f"{TAG}_kernel_stats.csv")
if os.path.isdir(os.path.join(SRC, "trace2")):
    stats("trace2", f"{TAG}_kernel_stats_two_parts.csv")
stats("vitl", f"{TAG}_vitl_kernel_stats.csv")
stats("tune", f"{TAG}_tune_kernel_stats.csv")
for src, dst in (("bench.json", f"{TAG}_bench.json"), ("vitl_bench.json", f"{TAG}_vitl_bench.json"), ("tune_bench.json", f"{TAG}_bench_tune.json")):
    shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
out = {"command": "rocprofv3 --kernel-trace [--stats | --pmc ...] -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype --streams 1",
       "note": "FETCH_SIZE/WRITE_SIZE are KB of L2<->fabric traffic (Infinity-Cache hits included); per MI355X_MICROARCH.md "
               "FETCH_SIZE reads exactly half of a wide coalesced stream on gfx950, so read bytes = 2*FETCH_SIZE*1024."}
fetch, nf = by_kernel(counters("pmc_fetch"), GEMM)
write, _ = by_kernel(counters("pmc_write"), GEMM)
sq, _ = by_kernel(counters("pmc_sq"), GEMM)
grbm, _ = by_kernel(counters("pmc_grbm"), GEMM)
rd, wr = 2 * fetch.get("FETCH_SIZE", 0) * 1024, write.get("WRITE_SIZE", 0) * 1024
out[GEMM] = {"dispatches_averaged": nf, "FETCH_SIZE_KB": fetch.get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get("WRITE_SIZE"),
             "read_bytes_corrected": rd, "write_bytes": wr,
             "l2_hit_rate": write.get("TCC_HIT_sum", 0) / max(write.get("TCC_HIT_sum", 0) + write.get("TCC_MISS_sum", 0), 1),
             "SQ": sq, "GRBM_GUI_ACTIVE": grbm.get("GRBM_GUI_ACTIVE"),
             "mfma_busy_frac_of_simd_cycles": sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(grbm.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024, 1)}
out["gemm_hbm_bytes_per_launch"] = rd + wr
for name in ("attn_heads_kernel", "gemm_tn_128x128x64", "image_tail_kernel", "embed_ln_pre_kernel"):
    f2, _ = by_kernel(counters("pmc_fetch"), name)
    w2, _ = by_kernel(counters("pmc_write"), name)
    out[name] = {"read_bytes_corrected": 2 * f2.get("FETCH_SIZE", 0) * 1024, "write_bytes": w2.get("WRITE_SIZE", 0) * 1024}
json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1200])
for r in rows[:12]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
