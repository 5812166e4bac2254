"""Condense rocprofv3 output directories into small CSVs (run on the GPU box).
usage: prof_summary.py <dir> <out_prefix>"""
import csv, glob, os, re, sys, collections
d, outp = sys.argv[1], sys.argv[2]
KNOWN = ("iteration_kernel", "fit_bf16_mfma_kernel", "fit_bf16_kernel", "fit_kernel", "lbfgsb_kernel",
         "screen_topk_kernel", "rows_kernel", "candidates_kernel", "labels_kernel", "evaluate_kernel",
         "shuffle_kernel", "svgd_kernel", "append_kernel", "select_kernel", "bore_spin_kernel")
def short(n):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<[^>]*>)?", n)
    base, targs = (m.group(1), m.group(2) or "") if m else (n, "")
    for k in KNOWN:
        if base.endswith(k):
            return k + targs.replace(" ", "")
    return n.split("(")[0][:70]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(outp + "_kernel_stats.csv", "w") as o:
        w = csv.writer(o); w.writerow(["kernel","calls","total_ns","avg_ns","pct","min_ns","max_ns"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print(open(outp + "_kernel_stats.csv").read())
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
if acc:
    names = sorted({c for k in acc for c in acc[k]})
    with open(outp + "_pmc_per_launch_mean.csv", "w") as o:
        w = csv.writer(o); w.writerow(["kernel","launches"] + names)
        for k in sorted(acc): w.writerow([k, cnt[k]] + [acc[k][c] / max(cnt[k], 1) for c in names])
    print(open(outp + "_pmc_per_launch_mean.csv").read())
