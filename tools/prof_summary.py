"""Condense rocprofv3 output directories into small CSVs (run on the GPU box).
usage: prof_summary.py <dir> <out_prefix>

Round 5: launches are grouped by (kernel, grid size) -- one kernel name covers launches of very different grids
(one loop, 64 loops, 256 loops of a config; the warm-up and the timed launch of the resident kernel), and a mean over
all of them is a figure nobody can reproduce.  Written:
  <prefix>_kernel_stats.csv          rocprofv3's own per-kernel statistics (--stats), names shortened
  <prefix>_kernel_by_grid.csv        from the kernel trace: per (kernel, grid, workgroup): calls, mean / min / max ns,
                                     registers, scratch, LDS
  <prefix>_pmc_by_grid.csv           from a --pmc pass: per (kernel, grid): launches, mean of every counter
  <prefix>_pmc_per_launch_mean.csv   the round-4 form (per kernel over all its grids), kept for comparison"""
import csv, glob, os, re, sys, collections
d, outp = sys.argv[1], sys.argv[2]
KNOWN = ("iteration_kernel", "queue_kernel", "fit_bf16_mfma_kernel", "fit_bf16_kernel", "fit_kernel_w8", "fit_kernel",
         "lbfgsb_kernel_occ2", "lbfgsb_kernel_w12", "lbfgsb_kernel_w8", "lbfgsb_kernel", "screen_topk_kernel", "rows_kernel",
         "candidates_kernel", "labels_kernel", "evaluate_kernel", "shuffle_kernel", "svgd_big_kernel", "svgd_kernel",
         "append_kernel", "select_kernel", "bore_spin_kernel")


def short(n):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<[^>]*>)?", n)
    base, targs = (m.group(1), m.group(2) or "") if m else (n, "")
    for k in KNOWN:
        if base.endswith(k):
            return k + targs.replace(" ", "")
    return n.split("(")[0][:70]


def grid_of(r):
    if "Grid_Size" in r and r["Grid_Size"] not in ("", None):
        return str(r["Grid_Size"])
    if "Grid_Size_X" in r:
        return "x".join(str(r.get(k, "1")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    return "?"


def wg_of(r):
    if "Workgroup_Size" in r and r["Workgroup_Size"] not in ("", None):
        return str(r["Workgroup_Size"])
    if "Workgroup_Size_X" in r:
        return "x".join(str(r.get(k, "1")) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    return "?"


for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(outp + "_kernel_stats.csv", "w") as o:
        w = csv.writer(o); w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print(open(outp + "_kernel_stats.csv").read())

by = collections.defaultdict(list); res = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), grid_of(r), wg_of(r))
        by[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        res[key] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("Scratch_Size", ""), r.get("LDS_Block_Size", ""))
if by:
    with open(outp + "_kernel_by_grid.csv", "w") as o:
        w = csv.writer(o); w.writerow(["kernel", "grid_work_items", "workgroup", "calls", "avg_ns", "min_ns", "max_ns", "vgpr", "agpr", "scratch", "lds"])
        for key in sorted(by):
            v = by[key]
            w.writerow(list(key) + [len(v), sum(v) / len(v), min(v), max(v)] + list(res[key]))
    print(open(outp + "_kernel_by_grid.csv").read())

acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
accg = collections.defaultdict(lambda: collections.defaultdict(float)); cntg = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); kg = (k, grid_of(r))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); accg[kg][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1; cntg[kg] += 1
if acc:
    names = sorted({c for k in acc for c in acc[k]})
    with open(outp + "_pmc_per_launch_mean.csv", "w") as o:
        w = csv.writer(o); w.writerow(["kernel", "launches"] + names)
        for k in sorted(acc): w.writerow([k, cnt[k]] + [acc[k][c] / max(cnt[k], 1) for c in names])
    with open(outp + "_pmc_by_grid.csv", "w") as o:
        w = csv.writer(o); w.writerow(["kernel", "grid_work_items", "launches"] + names)
        for kg in sorted(accg): w.writerow(list(kg) + [cntg[kg]] + [accg[kg][c] / max(cntg[kg], 1) for c in names])
    print(open(outp + "_pmc_by_grid.csv").read())
