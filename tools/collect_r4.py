"""Copy what tools/profile_r4.sh <tag> left under gpurun_out/ into profiles/r4/ (the judged copies) and
rebuild profiles/r4/pmc_traffic.json (with the sha256 of the kernel sources the pass was collected on) from the PMC passes.  usage: python tools/collect_r4.py <tag>"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", "r4")
pairs = {f"{tag}_bench_driver.json": "bench_driver_form.json", f"{tag}_bench_T100.json": "bench_T100.json",
         f"{tag}_bench_under_rocprof.json": "bench_under_rocprof.json",
         f"{tag}_headline_kernel_stats.csv": "headline_kernel_stats.csv", f"{tag}_configs_kernel_stats.csv": "configs_kernel_stats.csv",
         f"{tag}_headline_FETCH_SIZE_pmc_per_launch_mean.csv": "headline_pmc_fetch_per_launch_mean.csv",
         f"{tag}_headline_WRITE_SIZE_pmc_per_launch_mean.csv": "headline_pmc_write_per_launch_mean.csv",
         f"{tag}_headline_sq_pmc_per_launch_mean.csv": "headline_pmc_sq_per_launch_mean.csv",
         f"{tag}_configs_FETCH_SIZE_pmc_per_launch_mean.csv": "configs_pmc_fetch_per_launch_mean.csv",
         f"{tag}_configs_WRITE_SIZE_pmc_per_launch_mean.csv": "configs_pmc_write_per_launch_mean.csv",
         f"{tag}_configs_sq_pmc_per_launch_mean.csv": "configs_pmc_sq_per_launch_mean.csv"}
pairs[f"{tag}_cfg_restarts.txt"] = "wide_restart_launches.txt"
for src, dst in pairs.items():
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
log = open(os.path.join(G, f"{tag}_profile.log")).read()
open(os.path.join(P, "loops_sweep.txt"), "w").write("".join(l + "\n" for l in log.splitlines() if l.startswith("loops ")))


def table(name):
    rows = {}
    for r in csv.DictReader(open(os.path.join(P, name))):
        rows[r["kernel"]] = r
    return rows


bench = json.loads(open(os.path.join(P, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
L, steps, warm = bench["config"]["loops_per_gpu"], bench["steps"], bench["warmup"]
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv) on `python3 bench.py --steps 20 "
                  "--warmup 5 --cpu-seconds 0 --no-configs --repeats 1 --survey-steps 0` (tools/profile_r4.sh, tools/collect_r4.py).  HBM bytes = "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950's FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM "
                  "section).  Resident workgroups: the run is TWO launches of iteration_kernel (warm-up steps, timed steps) that together "
                  "carry loops x (warm-up + steps) loop-iterations.  ",
       "build": "r4 (fit: hardware rcp / sqrt / exp2; lagging loops at wave priority 3; restarts of the wide configs from a queue, "
                "bf16 acquisition kernels on the bf16 matrix cores with one LDS image; shuffles of 65..112 rows drawn by one wave)"}
sys.path.insert(0, ROOT)
import bench as _bench
# the digest the GPU box printed for the snapshot it profiled (bench.py prints traffic_stale when the
# sources have changed since); this tree's digest only if that file is missing
try:
    out["csrc_sha256"] = open(os.path.join(G, f"{tag}_csrc_digest.txt")).read().split()[-1]
except OSError:
    out["csrc_sha256"] = _bench.csrc_digest()
f, w = table("headline_pmc_fetch_per_launch_mean.csv"), table("headline_pmc_write_per_launch_mean.csv")
k = [x for x in f if x.startswith("iteration_kernel")][0]
n, its = int(f[k]["launches"]), L * (steps + warm)
fk, wk = float(f[k]["FETCH_SIZE"]), float(w[k]["WRITE_SIZE"])
out["iteration_kernel"] = {"FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "launches": n, "loop_iterations": its,
                           "hbm_bytes_per_model": (2 * fk + wk) * 1024 * n / its,
                           "fetch_bytes_per_loop_iteration": 2 * fk * 1024 * n / its, "write_bytes_per_loop_iteration": wk * 1024 * n / its}
f, w = table("configs_pmc_fetch_per_launch_mean.csv"), table("configs_pmc_write_per_launch_mean.csv")
out["configs_leg"] = {kk: {"FETCH_SIZE_KB_per_launch": float(f[kk]["FETCH_SIZE"]), "WRITE_SIZE_KB_per_launch": float(w[kk]["WRITE_SIZE"]),
                           "launches": int(f[kk]["launches"])} for kk in f if kk in w and not kk.startswith(("void at::", "__amd"))}
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["iteration_kernel"], indent=1))

sweep = {}
for l in open(os.path.join(P, "loops_sweep.txt")):
    parts = l.split()
    sweep[parts[1].rstrip(":")] = float(parts[2])
json.dump({"_source": "tools/profile_r4.sh: python3 bench.py --steps 40 --warmup 3 --repeats 3 --loops L on one MI355X (median of 3 fresh engines)",
           "build": out["build"], "csrc_sha256": out["csrc_sha256"], "steps": 40, "it_per_s": sweep}, open(os.path.join(P, "loops_sweep.json"), "w"), indent=1)
