"""Phase cycles of the STANDALONE restart kernel lbfgsb_kernel<1> (no register cap, no scratch) on
the same kind of problems the fused kernel runs: trained 16-16-1 classifiers, 3 restarts from the
screened starts (diagnostic build; GPU box).  Compare with tools/engine_phases.py (fused kernel)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_stamps.so"))
import numpy as np, torch
from bore_amd import _lib, ops
from bore_amd.engine import NativeEngine
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = NativeEngine(np.arange(L), async_loops=True)
eng.run(15)
th = torch.from_numpy(eng.state()[0]).cuda()
desc = eng.desc
lo, hi = np.zeros(2), np.ones(2)
x0, _ = ops.sample_screen_topk(desc, th, 0, 1024, lo, hi, 3, draw_index=15)
names = ["cauchy", "formk", "cmprlb", "subsm", "lnsrlb", "matupd", "formt", "head", "freev", "accept", "cachechk", "bfgspair", "d=z-x"]
GAPS = {15: "outside advance (evaluation + kernel loop)", 18: "tail before return", 19: "gap before formt", 20: "gap entry->head", 21: "gap before cauchy", 22: "gap before freev",
        23: "gap before formk", 24: "gap before cmprlb", 25: "gap before subsm", 26: "gap before d=z-x", 27: "gap before lnsrlb",
        28: "gap before cachechk", 29: "gap before accept", 30: "gap before bfgspair", 31: "gap before matupd"}
for rep in range(2):
    lib.bore_debug_lphases_reset()
    x, fun, jac, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, "identity", True, maxiter=1000, ftol=1e-9)
    torch.cuda.synchronize()
pp = (C.c_ulonglong * (4096 * 64))()
lib.bore_debug_lpp(pp)
pp = np.array(pp, dtype=np.float64).reshape(4096, 64)[:4 * L].reshape(L, 4, 64)[:, :3]
n_prob = L * 3
nfev = info.cpu().numpy()[:, :, 1].sum()
tot_adv, tot_fg = pp[..., 13].sum(), pp[..., 14].sum()
print(f"standalone lbfgsb_kernel<1>, {L} models x 3 restarts: per problem advance {tot_adv / n_prob:.0f} cycles, f/g {tot_fg / n_prob:.0f}; nfev {nfev / n_prob:.1f} rounds {pp[..., 45].sum() / n_prob:.1f}")
acc = 0.0
for i, nm in enumerate(names):
    cyc, calls = pp[..., i].sum(), pp[..., 32 + i].sum()
    acc += cyc
    print(f"  {nm:8s}: {cyc / n_prob:9.0f} cycles per problem, {calls / n_prob:6.2f} calls, {cyc / max(calls, 1):7.0f} per call")
for g, nm in GAPS.items():
    cyc, calls = pp[..., g].sum(), pp[..., 32 + g].sum()
    print(f"  [{nm}]: {cyc / n_prob:9.0f} cycles per problem, {calls / n_prob:6.2f} times, {cyc / max(calls, 1):7.0f} each")
print(f"  rest of advance: {(tot_adv - acc) / n_prob:9.0f} cycles per problem ({100 * (tot_adv - acc) / (tot_adv + tot_fg):5.1f} %); f/g per round {tot_fg / max(pp[..., 45].sum(), 1):.0f}")
