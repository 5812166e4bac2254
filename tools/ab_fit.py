"""A/B of two builds of libbore_hip.so on the wide fits: bit-identity of the results and time per
Adam step (GPU box).  usage: python tools/ab_fit.py  (runs itself once per library)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("shape3_f32", 16, [64, 64, 64, 1], "float32"), ("shape3_bf16", 16, [64, 64, 64, 1], "bfloat16"),
         ("shape4_bf16", 32, [128, 128, 1], "bfloat16"), ("shape2_f32", 6, [32, 32, 1], "float32"),
         ("shape1_f32", 2, [16, 16, 1], "float32")]


def child(tag):
    import numpy as np, torch
    from bore_amd import _lib, ops
    out = {}
    for name, D, units, compute in CASES:
        rs = np.random.RandomState(7)
        acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
        desc = _lib.make_desc(D, units, acts, compute=compute)
        P = ops.param_count(desc)
        L, N, E = 3, 200, 40
        th = torch.from_numpy(rs.normal(scale=0.2, size=(L, P)).astype(np.float32)).cuda()
        m, v = torch.zeros_like(th), torch.zeros_like(th)
        t = torch.zeros(L, dtype=torch.int64, device="cuda")
        X = torch.from_numpy(rs.uniform(size=(L, N, D)).astype(np.float32)).cuda()
        z = torch.from_numpy((rs.uniform(size=(L, N)) < 0.25).astype(np.float32)).cuda()
        ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=3, want_loss=False)   # also the warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=3, epoch0=E, want_loss=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = E * -(-N // 64)
        print(f"{tag} {name}: {1e6 * dt / steps:7.2f} us per Adam step ({L} models, N={N})", flush=True)
        out[name + "_th"], out[name + "_m"], out[name + "_v"] = (a.cpu().numpy() for a in (th, m, v))
    np.savez(os.path.join(ROOT, "gpurun_out", f"ab_{tag}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    import numpy as np
    libs = {"new": os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip.so"),
            "prev": os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_prev.so")}
    for tag, path in libs.items():
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, BORE_LIB_PATH=path), check=True)
    a, b = (np.load(os.path.join(ROOT, "gpurun_out", f"ab_{t}.npz")) for t in ("new", "prev"))
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        print(f"{k}: {'bit-identical' if same else 'DIFFERENT, max abs diff %g' % np.abs(a[k] - b[k]).max()}")
