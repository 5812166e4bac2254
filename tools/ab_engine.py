"""A/B of library builds on the headline engine (GPU box): for each build named on the command line
(tag of bore_amd/csrc/libbore_hip_<tag>.so, or "default"), a child process runs the 512-loop
asynchronous engine `reps` times (fresh engine, 5 warm-up + 20 timed steps) and prints the median
BO-iterations/s and the device phases; builds alternate so that drift hits them alike.
usage: python tools/ab_engine.py [--reps 7] [--loops 512] [--steps 20] [--check] tagA tagB ...
--check: the builds' trajectories (observations after the run) must be bit-identical to the first's."""
import argparse, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(tag, loops, steps, warm, reps):
    sys.path.insert(0, ROOT)
    import time
    import numpy as np, torch
    from bore_amd.engine import NativeEngine
    vals, dev, digest = [], None, None
    for _ in range(reps):
        eng = NativeEngine(np.arange(loops), async_loops=True,
                           **({"objective": "branin01"} if os.environ.get("BORE_AB_OBJECTIVE", "native") == "native" else {}))
        eng.run(warm)
        eng.take_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eng.take_stats()
        n = max(st["phase_iterations"], 1)
        dev = {k: 1e-3 * st["phase_ns_" + k] / n for k in ("labels", "fit", "screen", "lbfgsb")}
        vals.append(loops * steps / dt)
        if digest is None:
            X, y = eng.observations()
            th = eng.state()[0]
            digest = hashlib.sha256(X.tobytes() + y.tobytes() + th.tobytes()).hexdigest()[:16]
        del eng
    print(json.dumps(dict(tag=tag, median=float(np.median(vals)), min=min(vals), max=max(vals), dev=dev, digest=digest)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--loops", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--child", default=None)
    ap.add_argument("tags", nargs="*")
    a = ap.parse_args()
    if a.child is not None:
        child(a.child, a.loops, a.steps, a.warmup, a.reps)
        sys.exit(0)
    res = {t: [] for t in a.tags}
    for rnd in range(a.rounds):
        for t in a.tags:
            env = dict(os.environ)
            env.pop("BORE_LIB_PATH", None)
            if t != "default":
                env["BORE_LIB_PATH"] = os.path.join(ROOT, "bore_amd", "csrc", f"libbore_hip_{t}.so")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", t, "--reps", str(a.reps), "--loops", str(a.loops),
                                "--steps", str(a.steps), "--warmup", str(a.warmup)], env=env, capture_output=True, text=True)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not lines:
                print(f"{t}: FAILED\n{p.stderr[-1500:]}")
                continue
            res[t].append(json.loads(lines[-1]))
    ref = None
    for t in a.tags:
        if not res[t]:
            continue
        med = sorted(r["median"] for r in res[t])[len(res[t]) // 2]
        d = res[t][-1]["dev"]
        dg = res[t][0]["digest"]
        ref = ref or dg
        print(f"{t:12s}: {med:9.0f} it/s (rounds: {' '.join('%.0f' % r['median'] for r in res[t])}) | device us: "
              + " ".join(f"{k} {v:6.1f}" for k, v in d.items()) + f" sum {sum(d.values()):7.1f} | trajectories {dg}"
              + ("" if not a.check else ("  SAME" if dg == ref else "  DIFFERENT")), flush=True)
