"""Per kernel family of OUR library: matrix-core occupancy and LDS bank conflicts from a rocprofv3 counter pass.
usage: pmc_by_kernel.py <dir with *counter_collection.csv>   (pass: --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)"""
import csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
acc = {}
for r in csv.DictReader(open(f)):
    kn = r["Kernel_Name"]
    if "anonymous namespace" not in kn or "at::" in kn:
        continue
    m = re.search(r"\(anonymous namespace\)::([A-Za-z0-9_]+(<[^(]*>)?)", kn)
    name = m.group(1) if m else kn[:60]
    a = acc.setdefault(name, {"n": set()})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["n"].add(r["Dispatch_Id"])
rows = []
for name, a in acc.items():
    gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if gui <= 0: continue
    rows.append((gui, name, len(a["n"]), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024),
                 a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else 0.0))
tot = sum(r[0] for r in rows)
print(f"{'kernel':70s} {'launches':>8s} {'% of GPU-active cycles':>22s} {'MFMA busy':>10s} {'LDS conflict / LDS active':>26s}")
for gui, name, n, mf, lc in sorted(rows, reverse=True)[:24]:
    print(f"{name[:70]:70s} {n:8d} {100 * gui / tot:22.1f} {mf:10.3f} {lc:26.3f}")
