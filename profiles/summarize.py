#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/* (rocprofv3 csv, written by profiles/collect.sh on the GPU box) into the committed summaries:
<tag>_kernel_stats.csv, <tag>_pmc_summary.json, <tag>_vitl_kernel_stats.csv, <tag>_tune_kernel_stats.csv and the bench JSON lines.
    python profiles/summarize.py r02"""
import collections, csv, glob, json, os, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
DST = os.path.join(ROOT, "profiles")
# round 5: the block GEMMs of the large batch run on the 384 x 256 family; the im2col-free patch GEMM stays a 256 x 256 flavour
GEMM = "gemm_tn_384x256x32_pp" if TAG >= "r05" else "gemm_tn_256x256x64_pp"
PATCH = "gemm_tn_256x256x64_pp"


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
if os.path.isdir(os.path.join(SRC, "tune_dense")):
    stats("tune_dense", f"{TAG}_tune_dense_kernel_stats.csv")
# (bench.json of the collection run is copied only when no newer line has been committed by hand: the round's final line is taken after the last code change)
for src, dst in ((("bench.json", f"{TAG}_bench.json"),) if not os.path.exists(os.path.join(DST, f"{TAG}_bench.json")) else ()) + (("vitl_bench.json", f"{TAG}_vitl_bench.json"), ("tune_bench.json", f"{TAG}_bench_tune.json"),
                 ("tune_dense_bench.json", f"{TAG}_bench_tune_dense.json")):
    if os.path.exists(os.path.join(SRC, src)):
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
if TAG >= "r05":
    # per flavour (the residual flavour serves out-proj and c_proj alternately: split at the median of its read bytes); algorithmic = A + W (+ residual) read
    pf_, pw_ = counters("pmc_fetch"), counters("pmc_write")
    flav = {}
    for tag_, name_, alg in (("Li2ELi0E", "qkv (LN fused)", 50432 * 768 * 2 + 2304 * 768 * 2), ("Li2ELi1E", "c_fc (LN + QuickGELU fused)", 50432 * 768 * 2 + 3072 * 768 * 2)):
        r_ = [2 * v["FETCH_SIZE"] * 1024 for (k, _), v in pf_.items() if GEMM in k and tag_ in k and v["FETCH_SIZE"] * 2048 > 1.5e8]
        w_ = [v["WRITE_SIZE"] * 1024 for (k, _), v in pw_.items() if GEMM in k and tag_ in k and v["WRITE_SIZE"] * 1024 > 1e8]
        if r_ and w_:
            flav[name_] = {"dispatches": len(r_), "read_bytes_corrected": sum(r_) / len(r_), "write_bytes": sum(w_) / len(w_), "algorithmic_read_bytes": alg,
                           "read_over_algorithmic": sum(r_) / len(r_) / alg}
    r_ = sorted(2 * v["FETCH_SIZE"] * 1024 for (k, _), v in pf_.items() if GEMM in k and "Li1ELi2E" in k)
    if r_:
        h = len(r_) // 2
        for name_, part, alg in (("out_proj (+residual, +partials, in-producer merge)", r_[:h], 2 * 50432 * 768 * 2 + 768 * 768 * 2),
                                 ("c_proj (+residual, +partials, in-producer merge)", r_[h:], 50432 * 3072 * 2 + 50432 * 768 * 2 + 768 * 3072 * 2)):
            flav[name_] = {"dispatches": len(part), "read_bytes_corrected": sum(part) / len(part), "algorithmic_read_bytes": alg, "read_over_algorithmic": sum(part) / len(part) / alg}
    out["gemm_read_traffic_per_shape"] = flav
for name in ("attn_heads_kernel", "gemm_tn_128x128x64", "image_tail_kernel", "embed_ln_pre"):
    f2, _ = by_kernel(counters("pmc_fetch"), name)
    w2, _ = by_kernel(counters("pmc_write"), name)
    out[name] = {"read_bytes_corrected": 2 * f2.get("FETCH_SIZE", 0) * 1024, "write_bytes": w2.get("WRITE_SIZE", 0) * 1024}
# round 4: the im2col-free patch GEMM (IM2COL flavour <T,0,0,true>) on its own; with embed_ln_pre above it is the whole patch path
f3, _ = by_kernel(counters("pmc_fetch"), PATCH + "IDF16_Li0ELi0ELb1")
w3, _ = by_kernel(counters("pmc_write"), PATCH + "IDF16_Li0ELi0ELb1")
out["patch GEMM (im2col-free)"] = {"read_bytes_corrected": 2 * f3.get("FETCH_SIZE", 0) * 1024, "write_bytes": w3.get("WRITE_SIZE", 0) * 1024}
pp = out["patch GEMM (im2col-free)"]["read_bytes_corrected"] + out["patch GEMM (im2col-free)"]["write_bytes"] + out["embed_ln_pre"]["read_bytes_corrected"] + out["embed_ln_pre"]["write_bytes"]
out["patch_path_bytes"] = {"measured": pp, "algorithmic_fp16_image": 256 * 3 * 224 * 224 * 2 + 256 * 197 * 768 * 2 + 768 * 768 * 2,
                           "note": "image in the compute dtype (77 MB) + tokens out (77.5 MB) + weights; the conv output makes one round trip between the two launches"}
# attention: SQ counters of the ViT-B heads kernel (same passes as the GEMM) and of the ViT-L/14@336 stream kernel (its own passes)
sq_h, _ = by_kernel(counters("pmc_sq"), "attn_heads_kernel")
g_h, _ = by_kernel(counters("pmc_grbm"), "attn_heads_kernel")
out["attn_heads_kernel"].update({"SQ": sq_h, "GRBM_GUI_ACTIVE": g_h.get("GRBM_GUI_ACTIVE"),
                                 "mfma_busy_frac_of_simd_cycles": sq_h.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(g_h.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024, 1)})
if os.path.isdir(os.path.join(SRC, "pmc_sq_vitl")):
    sq_s, _ = by_kernel(counters("pmc_sq_vitl"), "attn_stream_kernel")
    g_s, _ = by_kernel(counters("pmc_grbm_vitl"), "attn_stream_kernel")
    out["attn_stream_kernel (ViT-L/14@336, B=128)"] = {"SQ": sq_s, "GRBM_GUI_ACTIVE": g_s.get("GRBM_GUI_ACTIVE"),
                                                       "mfma_busy_frac_of_simd_cycles": sq_s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(g_s.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024, 1)}

# instruction-issue counters (profiles/collect_issue.sh: two passes of 8 SQ counters each).  SQ_INSTS_* count wave instructions (SQ_INSTS_VALU includes
# the MFMAs), SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles summed over waves, SQ_VALU_MFMA_*_CYCLES count cycles summed over SIMDs.
if os.path.isdir(os.path.join(SRC, "pmc_issue1")):
    def issue(sub1, sub2, match):
        a, n = by_kernel(counters(sub1), match)
        b, _ = by_kernel(counters(sub2), match)
        a.update(b)
        d = {"dispatches_averaged": n, "counters": a}
        if a.get("SQ_INSTS_VALU"):
            valu = a["SQ_INSTS_VALU"] - a.get("SQ_INSTS_MFMA", 0)
            d["derived"] = {
                "mfma_instructions": a.get("SQ_INSTS_MFMA"), "other_vector_instructions": valu,
                "lds_instructions": a.get("SQ_INSTS_LDS"), "scalar_instructions": a.get("SQ_INSTS_SALU"), "vmem_instructions": a.get("SQ_INSTS_VMEM"),
                "vector_issue_cycles_per_vector_instruction_incl_mfma": 4 * a.get("SQ_ACTIVE_INST_VALU", 0) / a["SQ_INSTS_VALU"],
                "wave_active_fraction": a.get("SQ_ACTIVE_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 0), 1),
                "mfma_cycles_with_a_vector_instruction_executing_beside": a.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 1),
                "vector_issue_cycles_over_mfma_busy_cycles": 4 * a.get("SQ_ACTIVE_INST_VALU", 0) / max(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 1)}
        return d
    out["issue"] = {
        "qkv GEMM <f16,2,0>": issue("pmc_issue1", "pmc_issue2", GEMM + "IDF16_Li2ELi0"),
        "c_fc GEMM <f16,2,1>": issue("pmc_issue1", "pmc_issue2", GEMM + "IDF16_Li2ELi1"),
        "out-proj / c_proj GEMM <f16,1,2>": issue("pmc_issue1", "pmc_issue2", GEMM + "IDF16_Li1ELi2"),
        "attn_heads_kernel (ViT-B/16, B=256)": issue("pmc_issue1", "pmc_issue2", "attn_heads_kernel")}
    if os.path.isdir(os.path.join(SRC, "pmc_issue1_vitl")):
        out["issue"]["attn_stream_kernel (ViT-L/14@336, B=128)"] = issue("pmc_issue1_vitl", "pmc_issue2_vitl", "attn_stream_kernel")


# ---- the GEMM family per launch shape, from the per-dispatch kernel trace (the persistent grid is 256 workgroups for every shape, so
# the shape is told by the kernel flavour and - for <T,1,2>, which serves out-proj and c_proj alternately - by the dispatch order
# inside a forward, cross-checked against the bimodal durations)
def per_shape(sub, out_name):
    path = newest(os.path.join(SRC, sub, "*", "*_kernel_trace.csv"))
    rows_ = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    shapes = collections.defaultdict(list)
    alt = 0
    for r in rows_:
        n = r["Kernel_Name"]
        if GEMM not in n and PATCH not in n:
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        if "Li2ELi0E" in n:
            key = "qkv (LN fused) N=2304 K=768" if us > 100 else "text tower / small (LN fused)"
        elif "Li2ELi1E" in n:
            key = "c_fc (LN + QuickGELU fused) N=3072 K=768" if us > 150 else "text tower c_fc"
        elif "Li1ELi2E" in n:
            key = ("out_proj (+residual, +partials) N=768 K=768", "c_proj (+residual, +partials) N=768 K=3072")[alt & 1]
            if (us > 150) != bool(alt & 1):      # order lost (e.g. a last block without partials in between): fall back to the duration
                key = "c_proj (+residual, +partials) N=768 K=3072" if us > 150 else "out_proj (+residual, +partials) N=768 K=768"
                alt = 1 if us > 150 else 0
            alt += 1
        elif "Li1ELi0E" in n:
            key = "c_proj last block (+residual) N=768 K=3072" if us > 150 else "out_proj / small (+residual)"
        elif "Li0ELi0E" in n:
            key = "patch embedding GEMM (bias only)"
        else:
            key = "other flavour"
        shapes[key].append(us)
    flops = {"qkv": 2 * 50432 * 2304 * 768, "c_fc": 2 * 50432 * 3072 * 768, "out_proj (": 2 * 50432 * 768 * 768, "c_proj": 2 * 50432 * 768 * 3072}
    with open(os.path.join(DST, out_name), "w") as f:
        w = csv.writer(f)
        w.writerow(["shape", "launches", "avg_us", "min_us", "max_us", "TFLOP/s (B=256)", "frac of 2516.6"])
        for k, v in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
            fl = next((x for p_, x in flops.items() if k.startswith(p_)), None)
            avg = sum(v) / len(v)
            tf = fl / avg * 1e-6 if fl else ""
            w.writerow([k, len(v), f"{avg:.1f}", f"{min(v):.1f}", f"{max(v):.1f}", f"{tf:.0f}" if fl else "", f"{tf / 2516.6:.3f}" if fl else ""])
    print(open(os.path.join(DST, out_name)).read())


per_shape("trace", f"{TAG}_gemm_shapes.csv")
# round 5: counter pass of the tuning step (configs[2]): HBM bytes per GEMM launch over every GEMM family of the step
if os.path.isdir(os.path.join(SRC, "pmc_fetch_tune")):
    tf_, ntf = by_kernel(counters("pmc_fetch_tune"), "gemm_tn_")
    tw_, _ = by_kernel(counters("pmc_write_tune"), "gemm_tn_")
    trd, twr = 2 * tf_.get("FETCH_SIZE", 0) * 1024, tw_.get("WRITE_SIZE", 0) * 1024
    json.dump({"command": "rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --mode tune --dtype bf16 --steps 4 --warmup 2",
               "gemm_dispatches_averaged": ntf, "read_bytes_corrected": trd, "write_bytes": twr, "gemm_hbm_bytes_per_launch": trd + twr,
               "l2_hit_rate": tw_.get("TCC_HIT_sum", 0) / max(tw_.get("TCC_HIT_sum", 0) + tw_.get("TCC_MISS_sum", 0), 1),
               "note": "every gemm_tn_* dispatch of the step (text tower forward / backward at M = 6 160 .. 18 480, image tower at M = 100 864); FETCH_SIZE x 2 per MI355X_MICROARCH.md"},
              open(os.path.join(DST, f"{TAG}_tune_pmc_summary.json"), "w"), indent=1)
if os.path.isdir(os.path.join(SRC, "multicrop")):
    stats("multicrop", f"{TAG}_multicrop_kernel_stats.csv")
for src, dst in (("multicrop_bench.json", f"{TAG}_bench_multicrop.json"),):
    if os.path.exists(os.path.join(SRC, src)):
        shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1200])
for r in rows[:12]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
