"""The optimiser's OWN float64 arithmetic per restart of each BASELINE config (the restart
roofline counted the network's FLOPs alone): bore_amd/csrc/lbfgsb.h compiled for the host with every float64
operation counted (tests/native/lbfgsb_flops.cpp: `double` replaced by a counting wrapper in that translation unit),
driven on the DEVICE-trained network of one loop of the config -- f / g of every evaluation come from the GPU
(bore_mlp_value_and_input_grad), the starts from the screening kernel -- for a sample of the restarts.  The counted
run must ask for exactly the evaluations the device's own restart of the same start asks for (same header, same
operations; checked).  Writes profiles/r6/optimiser_flops.json, which bench.py's restart roofline reads.
usage (GPU box): python tools/lbfgsb_flops.py [restarts per config]"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bore_amd import _lib, ops

SRC = os.path.join(ROOT, "tests", "native", "lbfgsb_flops.cpp")
SO = os.path.join(ROOT, "tests", "native", "liblbfgsb_flops.so")
HDR = os.path.join(ROOT, "bore_amd", "csrc", "lbfgsb.h")
if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", SRC, "-o", SO], check=True)
lib = C.CDLL(SO)
CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
CONFIGS = dict(bench.WIDE_CONFIGS)
CONFIGS["cfg1_branin2_16-16-1_R3"] = dict(D=2, units=[16, 16, 1], R=64, Ns=1024, N=30, compute="float32")   # (64 starts: a sample)
out = {"_source": "tools/lbfgsb_flops.py on the GPU box: lbfgsb.h's float64 operations counted in a host build "
                  "(tests/native/lbfgsb_flops.cpp), objective evaluated by the device kernels; per restart, mean over the sample",
       "csrc_sha256": bench.csrc_digest(), "configs": {}}
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for name, c in CONFIGS.items():
    D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
    acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
    desc = _lib.make_desc(D, units, acts, compute=c["compute"])
    P = ops.param_count(desc)
    rs = np.random.RandomState(0)
    th = torch.from_numpy(rs.normal(scale=0.3, size=(1, P)).astype(np.float32)).cuda()
    m, v = torch.zeros_like(th), torch.zeros_like(th)
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    X, y = bench._synthetic(rs, 1, N, D)
    z = (y < np.quantile(y, 0.25, axis=1)[:, None]).astype(np.float32)
    ops.mlp_fit(desc, th, m, v, t, torch.from_numpy(X.astype(np.float32)).cuda(), torch.from_numpy(z).cuda(), 200, 64, seed=0, want_loss=False)
    lo, hi = np.zeros(D), np.ones(D)
    x0, _ = ops.sample_screen_topk(desc, th, 0, Ns, lo, hi, R)
    opts = dict(maxiter=1000, ftol=1e-9)
    _, _, _, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, "identity", True, **opts)
    info = info.cpu().numpy()[0]
    starts = x0.cpu().numpy()[0]
    nbd = np.full(D, 2, dtype=np.int32)
    xbuf = torch.empty((1, 1, D), dtype=torch.float64, device="cuda")

    def cb(n_, xp, fp, gp):
        xbuf.copy_(torch.from_numpy(np.ctypeslib.as_array(xp, shape=(n_,)).reshape(1, 1, n_).copy()))
        val, grad = ops.mlp_value_and_input_grad(desc, th, xbuf, "identity", True)
        fp[0] = float(val.cpu().numpy()[0, 0])
        g = grad.cpu().numpy()[0, 0]
        for i in range(n_):
            gp[i] = g[i]

    rows = []
    picks = np.unique(np.linspace(0, R - 1, min(S, R)).astype(int))   # (spread over the ranking: the best starts converge at once)
    for r in picks:
        xo = np.empty(D); fo = C.c_double(); oi = np.zeros(3, dtype=np.int32); oc = np.zeros(2, dtype=np.uint64)
        lib.lbfgsb_flops_minimize(D, 10, dp(np.ascontiguousarray(starts[r])), dp(lo), dp(hi), nbd.ctypes.data_as(C.POINTER(C.c_int)),
                                  C.c_double(1e-9 / np.finfo(float).eps), C.c_double(1e-5), 1000, 15000, 20, CB(cb), dp(xo), C.byref(fo),
                                  oi.ctypes.data_as(C.POINTER(C.c_int)), oc.ctypes.data_as(C.POINTER(C.c_ulonglong)))
        rows.append((int(oi[0]), int(oi[1]), int(oc[0]), int(oc[1]), int(info[r, 0]), int(info[r, 1])))
    a = np.array(rows, dtype=np.float64)
    same = int(np.sum((a[:, 0] == a[:, 4]) & (a[:, 1] == a[:, 5])))
    launch = dict(nit_mean_of_the_launch=float(info[:, 0].mean()), nfev_mean_of_the_launch=float(info[:, 1].mean()))
    nfev = a[:, 1].sum()
    e = {"restarts_counted": len(rows), "same_nit_and_nfev_as_the_device": same,
         "nit_mean": a[:, 0].mean(), "nfev_mean": a[:, 1].mean(),
         "addsubmul_per_restart": a[:, 2].mean(), "divsqrt_per_restart": a[:, 3].mean(),
         "addsubmul_per_evaluation": a[:, 2].sum() / nfev, "divsqrt_per_evaluation": a[:, 3].sum() / nfev,
         "network_flops_per_evaluation_fwd_bwd": 4.0 * sum(i * o for i, o in zip([D] + units[:-1], units)), **launch}
    out["configs"][name] = e
    print(name, json.dumps(e), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r6"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6", "optimiser_flops.json"), "w"), indent=1)
