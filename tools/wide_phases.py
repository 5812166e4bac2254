"""Phase cycles of the restart kernels of the wide BASELINE configs, summed over the problems of ONE loop
(diagnostic build libbore_hip_stamps.so = -DBORE_STAMPS [-DBORE_SHAPE_MASK=...]; GPU box).
usage: python tools/wide_phases.py <cfg2|cfg3|cfg5> [loops] [restarts per loop]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_stamps.so"))
import numpy as np, torch
import bench
from bore_amd import _lib, ops
lib = _lib.lib()
key = {"cfg2": "cfg2_hartmann6_32-32-1_R256", "cfg3": "cfg3_hpo16_64-64-64-1_R1024", "cfg5": "cfg5_nas32_128-128-1_bf16_R4096"}[sys.argv[1]]
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = bench.WIDE_CONFIGS[key]
D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
if len(sys.argv) > 3:          # (restarts per loop: 4 = one workgroup of four waves, ONE wave per SIMD)
    R = int(sys.argv[3])
acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
desc = _lib.make_desc(D, units, acts, compute=c["compute"])
P = ops.param_count(desc)
rs = np.random.RandomState(0)
th = torch.from_numpy(rs.normal(scale=0.3, size=(loops, P)).astype(np.float32)).cuda()
m, v = torch.zeros_like(th), torch.zeros_like(th)
t = torch.zeros(loops, dtype=torch.int64, device="cuda")
X, y = bench._synthetic(rs, loops, N, D)
z = (y < np.quantile(y, 0.25, axis=1)[:, None]).astype(np.float32)
ops.mlp_fit(desc, th, m, v, t, torch.from_numpy(X.astype(np.float32)).cuda(), torch.from_numpy(z).cuda(), 200, 64, seed=0, want_loss=False)
lo, hi = np.zeros(D), np.ones(D)
x0, _ = ops.sample_screen_topk(desc, th, 0, Ns, lo, hi, R)
for rep in range(2):
    lib.bore_debug_lphases_reset()
    x, fun, jac, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, "identity", True, maxiter=1000, ftol=1e-9)
    torch.cuda.synchronize()
pp = (C.c_ulonglong * (4096 * 64))()
lib.bore_debug_lpp(pp)
pp = np.array(pp, dtype=np.float64).reshape(4096, 64)
used = pp[:, 45] > 0          # rows (problem slots) that ran at least one evaluation
pp = pp[used]
n_prob = max(pp[:, 32 + 0].size, 1)
names = ["cauchy", "formk", "cmprlb", "subsm", "lnsrlb", "matupd", "formt", "head", "freev", "accept", "cachechk", "bfgspair", "d=z-x"]
inf = info.cpu().numpy()
print(f"{sys.argv[1]}: {loops} loops x {R} restarts; nit mean {inf[:, :, 0].mean():.1f}, nfev mean {inf[:, :, 1].mean():.1f}; stamped slots {n_prob} (several problems per slot)")
tot_adv, tot_fg, rounds = pp[:, 13].sum(), pp[:, 14].sum(), pp[:, 45].sum()
print(f"per evaluation: f/g {tot_fg / rounds:.0f} cycles; advance {tot_adv / rounds:.0f} cycles per evaluation; share of f/g {100 * tot_fg / (tot_adv + tot_fg):.1f} %")
for i, nm in enumerate(names):
    cyc, calls = pp[:, i].sum(), pp[:, 32 + i].sum()
    if calls:
        print(f"  {nm:8s}: {100 * cyc / (tot_adv + tot_fg):5.1f} %  {cyc / calls:8.0f} cycles per call, {calls / rounds:5.2f} calls per evaluation")
if os.environ.get("BORE_PHASES_CAUCHY"):      # a -DBORE_STAMPS -DBORE_STAMPS_CAUCHY build: the stages of cauchy
    cnames = {19: "classify + sums over the moving variables", 20: "first middle-matrix product", 21: "per breakpoint: heap",
              22: "per breakpoint: workspace updates", 23: "per breakpoint: middle-matrix part (col > 0)", 24: "tail"}
    calls_c = pp[:, 32 + 0].sum()
    for i, nm in cnames.items():
        cyc, calls = pp[:, i].sum(), pp[:, 32 + i].sum()
        if calls:
            print(f"  cauchy / {nm:48s}: {cyc / max(calls_c, 1):8.0f} cycles per call of cauchy ({calls / max(calls_c, 1):5.2f} marks per call, {cyc / calls:7.0f} cycles each)")
if os.environ.get("BORE_PHASES_FORMK"):       # a -DBORE_STAMPS -DBORE_STAMPS_FORMK build: the stages of formk (and dcsrch)
    fnames = {19: "new rows / column (after an update)", 20: "old parts (entered / left variables)", 21: "assembly of WN",
              22: "Cholesky of block (1,1)", 23: "diagonal check + triangular solves", 24: "block (2,2)", 25: "Cholesky of block (2,2)"}
    calls_f = pp[:, 32 + 1].sum()
    for i, nm in fnames.items():
        cyc, calls = pp[:, i].sum(), pp[:, 32 + i].sum()
        if calls:
            print(f"  formk / {nm:44s}: {cyc / max(calls_f, 1):8.0f} cycles per call of formk ({calls / max(calls_f, 1):5.2f} marks per call)")
