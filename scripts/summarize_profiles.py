"""Condense the rocprofv3 output of scripts/collect_profiles.sh into the files kept under
profiles/:  <tag>_bench_default.json, <tag>_bench_kernel_stats.csv (top kernels),
<tag>_bench_rocprof_summary.json (stats + PMC per launch of the Gram kernel + derived numbers).

usage: summarize_profiles.py <raw dir> <out dir> <tag>
"""
import csv, glob, json, os, shutil, sys

GRAM = "syrk_batch_kernel<true, 0, 0>"              # the cfg2 Gram launch (all 8 row blocks, LDS-DMA, 128 x 128 tiles)
M_BLOCK, N = 1038240, 8760                    # rows of X covered by one launch


def find(d, suffix):
    # (the newest: a local gpurun_out/ accumulates the files of earlier collections)
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


def kernel_stats(raw):
    f = find(os.path.join(raw, "stats"), "kernel_stats.csv")
    rows = list(csv.DictReader(open(f))) if f else []
    return f, rows


def gram_trace(raw, sub="stats"):
    f = find(os.path.join(raw, sub), "kernel_trace.csv")
    out = []
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if GRAM in r["Kernel_Name"]:
            out.append(r)
    return out


def pmc(raw, sub):
    f = find(os.path.join(raw, sub), "counter_collection.csv")
    agg = {}
    if not f:
        return agg
    # the big launches only (the same kernel also serves the small refine Grams and bench.py's
    # priming problem): those with at least half the largest grid
    recs = [r for r in csv.DictReader(open(f)) if GRAM in r["Kernel_Name"]]
    gmax = max([int(r["Grid_Size"]) for r in recs] or [0])
    per = {}
    for r in recs:
        if 2 * int(r["Grid_Size"]) < gmax:
            continue
        key = (r["Counter_Name"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    for (name, _), v in per.items():
        a = agg.setdefault(name, [0.0, 0])
        a[0] += v
        a[1] += 1
    return {k: {"per_launch_avg": v[0] / v[1], "launches": v[1]} for k, v in agg.items()}


def main():
    raw, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(out, exist_ok=True)
    if os.path.exists(os.path.join(raw, "bench_default.json")):
        shutil.copy(os.path.join(raw, "bench_default.json"), os.path.join(out, f"{tag}_bench_default.json"))
    if os.path.exists(os.path.join(raw, "bench_randomized.json")):
        shutil.copy(os.path.join(raw, "bench_randomized.json"), os.path.join(out, f"{tag}_bench_randomized.json"))
    fr = find(os.path.join(raw, "stats_rand"), "kernel_stats.csv")
    if fr:
        rr = list(csv.DictReader(open(fr)))
        rr.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        if rr:
            with open(os.path.join(out, f"{tag}_bench_randomized_kernel_stats.csv"), "w", newline="") as g:
                w = csv.DictWriter(g, fieldnames=list(rr[0].keys()))
                w.writeheader()
                w.writerows(rr[:20])
    fp = find(os.path.join(raw, "stats_powerlaw"), "kernel_stats.csv")
    if fp:   # the same step on the gap-free spectrum: what the eigen stage launches
        rp = list(csv.DictReader(open(fp)))
        rp.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        if rp:
            with open(os.path.join(out, f"{tag}_bench_powerlaw_kernel_stats.csv"), "w", newline="") as g:
                w = csv.DictWriter(g, fieldnames=list(rp[0].keys()))
                w.writeheader()
                w.writerows(rp[:25])
    for sub in ("cfg4_k50", "cfg4_k200"):   # cfg4: top kernels of the (warm-up + one timed) randomized SVDs
        fc = find(os.path.join(raw, "stats_" + sub), "kernel_stats.csv")
        if fc:
            rc = list(csv.DictReader(open(fc)))
            rc.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
            if rc:
                with open(os.path.join(out, f"{tag}_{sub}_kernel_stats.csv"), "w", newline="") as g:
                    w = csv.DictWriter(g, fieldnames=list(rc[0].keys()))
                    w.writeheader()
                    w.writerows(rc[:16])
    fcp = find(os.path.join(raw, "pmc_cfg4_k50"), "counter_collection.csv")
    if fcp:   # HBM-side traffic of the tall-skinny kernels at cfg4 (FETCH_SIZE in KB, doubled: gfx950 note)
        per = {}
        for r in csv.DictReader(open(fcp)):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            kn = r["Kernel_Name"]
            name = next((k for k in ("skinny16_kernel", "skinny16_gram_reduce", "skinny_kernel", "syrk_batch_kernel", "xty_small_kernel",
                                     "gemm_tn_reduce", "row_center_scale", "scale_columns") if k in kn), kn[:60])
            if name in ("skinny16_kernel", "syrk_batch_kernel"):
                name += kn[kn.index(name) + len(name):].split(">")[0] + ">"
            a = per.setdefault(name, [0.0, set()])
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
        rows4 = sorted(((k, v[0], len(v[1])) for k, v in per.items()), key=lambda x: -x[1])[:8]
        m4, n4 = 15573600, 3653
        json.dump({"what": "rocprofv3 --pmc FETCH_SIZE over scripts/bench_cfg4.py --k 50 --quick (warm-up + one timed SVD = 2 SVDs, "
                           "12 passes over X of 227.6 GB each)",
                   "algorithmic_bytes_of_X_per_pass": 4.0 * m4 * n4,
                   "kernels": [{"kernel": k, "FETCH_SIZE_KB_sum": v, "dispatches": c, "bytes_corrected": 2.0 * v * 1024,
                                "passes_over_X_equiv": 2.0 * v * 1024 / (4.0 * m4 * n4)} for k, v, c in rows4]},
                  open(os.path.join(out, f"{tag}_cfg4_k50_traffic.json"), "w"), indent=1)
    fsq = find(os.path.join(raw, "pmc_cfg4_sq"), "counter_collection.csv")
    if fsq:   # matrix-core occupancy of the tall-skinny kernels at cfg4 (per kernel family, summed over its dispatches)
        acc = {}
        for r in csv.DictReader(open(fsq)):
            kn = r["Kernel_Name"]
            name = next((k for k in ("skinny16_kernel", "syrk_batch_kernel", "xty_small_kernel") if k in kn), None)
            if name is None:
                continue
            name += kn[kn.index(name) + len(name):].split(">")[0] + ">"
            a = acc.setdefault(name, {"dispatches": set()})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            a["dispatches"].add(r["Dispatch_Id"])
        outk = []
        for name, a in acc.items():
            gui = a.get("GRBM_GUI_ACTIVE", 0.0)
            if gui <= 0:
                continue
            outk.append({"kernel": name, "dispatches": len(a["dispatches"]),
                         "mfma_busy_frac": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024),
                         "lds_bank_conflict_frac_of_lds_active": (a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"])
                         if a.get("SQ_LDS_IDX_ACTIVE") else None,
                         "gpu_active_cycles_per_xcd": gui / 8})
        outk.sort(key=lambda d: -d["gpu_active_cycles_per_xcd"])
        json.dump({"what": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE over "
                           "scripts/bench_cfg4.py --k 50 --quick; mfma_busy_frac = busy cycles / (active cycles x 1024 SIMDs)",
                   "kernels": outk[:6]}, open(os.path.join(out, f"{tag}_cfg4_k50_sq.json"), "w"), indent=1)
    f, rows = kernel_stats(raw)
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    top = rows[:12]
    if top:
        with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as g:
            w = csv.DictWriter(g, fieldnames=list(top[0].keys()))
            w.writeheader()
            w.writerows(rows[:25])
    allg = gram_trace(raw)
    gsz = lambda r: int(r.get("Grid_Size") or r["Grid_Size_X"])
    gmax = max([gsz(r) for r in allg] or [0])
    big = [r for r in allg if 2 * gsz(r) >= gmax]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in big]
    flops = float(M_BLOCK) * N * (N + 1)   # 2 m n (n + 1) / 2: the useful flops of one triangle (as bench.py)
    gram = {}
    if durs:
        r0 = big[0]
        gram = {"launches": len(durs), "avg_ms": sum(durs) / len(durs), "min_ms": min(durs), "max_ms": max(durs),
                "grid_size": r0.get("Grid_Size") or r0["Grid_Size_X"],
                "workgroup": r0.get("Workgroup_Size") or r0["Workgroup_Size_X"],
                "lds_bytes": r0.get("LDS_Block_Size"), "vgpr": r0.get("VGPR_Count"), "sgpr": r0.get("SGPR_Count"),
                "algorithmic_flops_per_launch": flops,
                "achieved_tflops_rocprof": flops / (sum(durs) / len(durs) * 1e-3) / 1e12}
    fetch, write, sq = pmc(raw, "pmc_fetch"), pmc(raw, "pmc_write"), pmc(raw, "pmc_sq")
    traffic, derived = {}, {}
    if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        fk, wk = fetch["FETCH_SIZE"]["per_launch_avg"], write["WRITE_SIZE"]["per_launch_avg"]
        traffic = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                   "bytes_per_launch_corrected": 2.0 * fk * 1024 + wk * 1024,
                   "algorithmic_bytes_per_launch": 4.0 * M_BLOCK * N,
                   "note": "L2->fabric bytes (Infinity Cache hits included); FETCH_SIZE (KB) doubled per "
                           "MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of a 16 B/lane stream at 64 B), "
                           "WRITE_SIZE exact.  The Gram re-reads every 128-column panel once per output tile that "
                           "needs it, so L2 misses >> the bytes of X; they are served by the Infinity Cache and "
                           "the kernel is MFMA-bound."}
    if sq:
        g = lambda k: sq.get(k, {}).get("per_launch_avg")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
            derived["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024)
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        if g("GRBM_GUI_ACTIVE") and gram:
            derived["clock_GHz_under_profile"] = g("GRBM_GUI_ACTIVE") / 8 / (gram["avg_ms"] * 1e-3) / 1e9
        derived["lds_bank_conflict_cycles"] = g("SQ_LDS_BANK_CONFLICT")
    if "TCC_HIT_sum" in write:
        h, m = write["TCC_HIT_sum"]["per_launch_avg"], write["TCC_MISS_sum"]["per_launch_avg"]
        derived["L2_hit_rate"] = h / (h + m)
    summary = {"kernel_stats_top12": top, "fetch": fetch, "write": write, "sq": sq,
               "gram_partial_kernel": gram, "traffic": traffic, "derived": derived,
               "made_by": "scripts/collect_profiles.sh + scripts/summarize_profiles.py"}
    json.dump(summary, open(os.path.join(out, f"{tag}_bench_rocprof_summary.json"), "w"), indent=1)
    print(json.dumps({"gram": gram, "traffic": traffic, "derived": derived}, indent=1))


if __name__ == "__main__":
    main()
