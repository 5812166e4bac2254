"""Copy what tools/profile_r6.sh <tag> left under gpurun_out/ into profiles/r6/ (the judged copies) and rebuild
profiles/r6/pmc_traffic.json from the PMC passes -- per (kernel, grid size), with the sha256 of the kernel sources the
pass was collected on.  usage: python tools/collect_r6.py <tag>"""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", "r6")
os.makedirs(P, exist_ok=True)
pairs = {f"{tag}_bench_driver.json": "bench_driver_form.json", f"{tag}_bench_T100.json": "bench_T100.json",
         f"{tag}_bench_under_rocprof.json": "bench_under_rocprof.json", f"{tag}_cfg_restarts.txt": "wide_restart_launches.txt"}
for leg in ("headline", "configs"):
    pairs[f"{tag}_{leg}_kernel_stats.csv"] = f"{leg}_kernel_stats.csv"
    pairs[f"{tag}_{leg}_kernel_by_grid.csv"] = f"{leg}_kernel_by_grid.csv"
    for pmc, nm in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("sq", "sq"), ("sq2", "sq2")):
        pairs[f"{tag}_{leg}_{pmc}_pmc_by_grid.csv"] = f"{leg}_pmc_{nm}_by_grid.csv"
pairs[os.path.join("r6", "optimiser_flops.json")] = "optimiser_flops.json"   # (tools/lbfgsb_flops.py, its own GPU call)
for src, dst in pairs.items():
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing:", src)
log = open(os.path.join(G, f"{tag}_profile.log")).read()
open(os.path.join(P, "loops_sweep.txt"), "w").write("".join(l + "\n" for l in log.splitlines() if l.startswith("loops ")))


def table(name):
    """{(kernel, grid): row}"""
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return {}
    return {(r["kernel"], r["grid_work_items"]): r for r in csv.DictReader(open(path))}


def grid_items(g):
    n = 1
    for p in str(g).split("x"):
        n *= int(p)
    return n


bench = json.loads(open(os.path.join(P, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
L, steps, warm = bench["config"]["loops_per_gpu"], bench["steps"], bench["warmup"]
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, --output-format csv) on `python3 bench.py "
                  "--steps 20 --warmup 5 --cpu-seconds 0 --no-configs --repeats 1 --survey-steps 0` (headline) and `--steps 2 --warmup 1 "
                  "--cpu-seconds 0 --repeats 1 --survey-steps 0` (configs leg); tools/profile_r6.sh, tools/prof_summary.py (per kernel AND "
                  "grid size), tools/collect_r6.py.  HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950's FETCH_SIZE reports half of a wide "
                  "coalesced read (MI355X_MICROARCH.md, HBM section).  The resident kernel: the run is TWO launches (warm-up steps, timed "
                  "steps) of the same grid that together carry loops x (warm-up + steps) loop-iterations.",
       "build": "r6 (optimiser's 2m x 2m matrices as two triangles of one block; formk / matupd shifts dealt to the lanes; fast ELU in the fit; "
                "fused loop kernel for static shape 5)"}
try:
    out["csrc_sha256"] = open(os.path.join(G, f"{tag}_csrc_digest.txt")).read().split()[-1]
except OSError:
    sys.path.insert(0, ROOT)
    import bench as _bench
    out["csrc_sha256"] = _bench.csrc_digest()

f, w, sq = table("headline_pmc_fetch_by_grid.csv"), table("headline_pmc_write_by_grid.csv"), table("headline_pmc_sq_by_grid.csv")
k = [x for x in f if x[0].startswith("iteration_kernel")]
if k:
    k = max(k, key=lambda x: grid_items(x[1]))
    n, its = int(f[k]["launches"]), L * (steps + warm)
    fk, wk = float(f[k]["FETCH_SIZE"]), float(w[k]["WRITE_SIZE"])
    out["iteration_kernel"] = {"kernel": k[0], "grid_work_items": k[1], "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                               "launches": n, "loop_iterations": its, "hbm_bytes_per_model": (2 * fk + wk) * 1024 * n / its,
                               "fetch_bytes_per_loop_iteration": 2 * fk * 1024 * n / its,
                               "write_bytes_per_loop_iteration": wk * 1024 * n / its}
    if k in sq:
        s = {c: float(v) for c, v in sq[k].items() if c.startswith("SQ_")}
        out["iteration_kernel"]["sq_per_launch"] = s
        if s.get("SQ_WAVE_CYCLES"):
            # (wave-cycles in which the wave issued an instruction / all wave-cycles: the bound that means something
            # for a kernel that is a chain of dependent small steps)
            out["iteration_kernel"]["issue_slot_utilisation"] = s.get("SQ_ACTIVE_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
            out["iteration_kernel"]["waiting_share_of_wave_cycles"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]

# configs leg: every kernel at its LARGEST grid (the 256-loop launches) and at its smallest (one loop)
f, w, sq = table("configs_pmc_fetch_by_grid.csv"), table("configs_pmc_write_by_grid.csv"), table("configs_pmc_sq_by_grid.csv")
kt = table("configs_kernel_by_grid.csv") if os.path.exists(os.path.join(P, "configs_kernel_by_grid.csv")) else {}
by_kernel = {}
for (kern, grid) in f:
    if (kern, grid) not in w or kern.startswith(("void at::", "__amd", "at::")):
        continue
    by_kernel.setdefault(kern, []).append(grid)
cfgs = {}
for kern, grids in by_kernel.items():
    for which, g in (("many_loops", max(grids, key=grid_items)), ("one_loop", min(grids, key=grid_items))):
        e = {"grid_work_items": g, "launches": int(f[(kern, g)]["launches"]),
             "hbm_bytes_per_launch": (2 * float(f[(kern, g)]["FETCH_SIZE"]) + float(w[(kern, g)]["WRITE_SIZE"])) * 1024}
        for key in kt:                      # the kernel trace names the grid x, y, z: match by work-item count
            if key[0] == kern and grid_items(key[1]) == grid_items(g):
                e["avg_ns_kernel_trace"] = float(kt[key]["avg_ns"]); e["calls_kernel_trace"] = int(kt[key]["calls"])
                e["hbm_GBs_by_counters"] = e["hbm_bytes_per_launch"] / e["avg_ns_kernel_trace"]
        if (kern, g) in sq:
            s = {c: float(v) for c, v in sq[(kern, g)].items() if c.startswith("SQ_")}
            if s.get("SQ_WAVE_CYCLES"):
                e["issue_slot_utilisation"] = s.get("SQ_ACTIVE_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
                e["valu_instructions_per_launch"] = s.get("SQ_INSTS_VALU")
        cfgs.setdefault(kern, {})[which] = e
out["configs_leg_by_kernel"] = cfgs


def pick(shape, bf, *prefixes):
    """the configs-leg kernel of a phase: name starts with one of `prefixes`, template arguments name the shape"""
    for kern in cfgs:
        if kern.startswith(prefixes) and re.search(rf"<(?:true,|false,)?{shape}[,>]", kern) and (("true" in kern.split("<")[1]) == bf or "fit" in kern):
            return kern
    return None


named = {"cfg2_hartmann6_32-32-1_R256": (2, False), "cfg3_hpo16_64-64-64-1_R1024": (3, False),
         "cfg5_nas32_128-128-1_bf16_R4096": (4, True), "plugin_default_D6": (5, False), "plugin_default_D16": (5, False),
         "plugin_D16_transform_identity": (5, False)}
out["configs"] = {}
for name, (shape, bf) in named.items():
    e = {}
    for phase, prefixes in (("fit", ("fit_bf16_mfma_kernel", "fit_kernel_w8", "fit_kernel") if bf else ("fit_kernel_w8", "fit_kernel")),
                            ("screen", ("screen_topk_kernel",)), ("fg", ("lbfgsb_kernel_w12", "lbfgsb_kernel_w8", "lbfgsb_kernel_occ2", "lbfgsb_kernel"))):
        kern = None
        for pre in prefixes:               # (the many-loops launch is the one with the larger grid: prefer its kernel)
            cand = [c for c in cfgs if c.startswith(pre + "<") and re.search(rf"<(?:true,|false,)?{shape}[,>]", c)]
            if phase != "fit":
                cand = [c for c in cand if (",true" in c) == bf]
            if cand:
                kern = max(cand, key=lambda c: grid_items(cfgs[c]["many_loops"]["grid_work_items"]))
                break
        if kern:
            e[phase] = dict(cfgs[kern]["many_loops"], kernel=kern)
    out["configs"][name] = e
    if name.startswith("plugin_"):
        e["note"] = "the three plugin legs launch the same kernels with the same grids: their launches are averaged together"
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out.get("iteration_kernel"), indent=1))
print(json.dumps(out["configs"], indent=1)[:3000])

sweep = {}
for l in open(os.path.join(P, "loops_sweep.txt")):
    parts = l.split()
    sweep[parts[1].rstrip(":")] = float(parts[2])
json.dump({"_source": "tools/profile_r6.sh: python3 bench.py --steps 40 --warmup 3 --repeats 3 --loops L on one MI355X (median of 3 fresh engines)",
           "build": out["build"], "csrc_sha256": out["csrc_sha256"], "steps": 40, "it_per_s": sweep}, open(os.path.join(P, "loops_sweep.json"), "w"), indent=1)
