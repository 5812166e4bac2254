"""bench.py as the driver starts it: `python bench.py --gpus N` from a plain shell must start its
own N ranks.  Runs here without a GPU through --dry-run
(an engine stand-in; rank spawning, rendezvous over gloo, barriers, max-over-ranks time and the
gather are the real code)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _run(*argv, timeout=240):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=_plain_env(),
                          capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _record(p, detail_dir=None):
    """What the driver does with a run: the LAST line of the last 8000 bytes of stdout is the record.  It must parse,
    stay under 4 KB (a 31 KB line left the driver with nothing) and carry `roofline` and
    `cpu_baseline`; everything else is in the side file it names."""
    tail = p.stdout.encode()[-8000:].decode(errors="replace")
    last = tail.rstrip("\n").splitlines()[-1]
    assert len(last.encode()) < 4096, len(last)
    d = json.loads(last)
    assert "roofline" in d and "cpu_baseline" in d and "metric" in d and "value" in d
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("{")] == [last]      # ONE JSON line, from rank 0
    if detail_dir is None:
        return d
    with open(os.path.join(str(detail_dir), f"bench_detail_n{d['n_gpus']}.json")) as f:
        return d, json.load(f)


def test_spawns_its_own_ranks_config4_as_written(tmp_path):
    p = _run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--detail-dir", str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    d, _ = _record(p, tmp_path)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "strong"            # 512 loops in total, whatever N
    assert d["config"]["total_loops"] == 512 and d["config"]["loops_per_gpu"] == 256
    assert d["dry_run"] is True and d["value"] is None
    eff = d["efficiency"]
    assert eff["share_loops"] == 256 and eff["total_loops"] == 512
    assert set(("eff_w", "eff_s", "T_1_share", "T_1_total", "T_N_total")) <= set(eff)
    # ... and, in the same line, the weak point: 512 loops on EVERY GPU (what one GPU runs for the headline), with
    # its own eff_w against T(1, 512) -- the curve that means something for a path of independent serial chains
    weak = d["weak_point_512_loops_per_gpu"]
    assert weak["loops_per_gpu"] == 512 and weak["total_loops"] == 1024 and weak["scaling"] == "weak"
    assert weak["T_1"] == eff["T_1_total"] and abs(weak["eff_w"] - weak["value"] / (2 * weak["T_1"])) < 1e-4 * weak["eff_w"]


def test_eight_ranks_the_drivers_widest_launch(tmp_path):
    """The N = 8 flow end to end (spawn, rendezvous, repeated regions with the same count on every
    rank, rank 0's reference points, gather): the line carries measured AND predicted efficiency
    and says how long each part of the flow took and what bounds it."""
    p = _run("--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1", "--detail-dir", str(tmp_path), timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    d, full = _record(p, tmp_path)            # the record (< 4 KB, last line) and the side file it names
    assert d["detail"].endswith("bench_detail_n8.json") and full["n_gpus"] == 8
    assert d["n_gpus"] == 8 and d["config"]["loops_per_gpu"] == 64 and d["scaling"] == "strong"
    eff = d["efficiency"]
    assert set(("eff_w", "eff_s", "T_1_share", "T_1_total", "T_N_total")) <= set(eff)
    assert eff["share_loops"] == 64 and eff["total_loops"] == 512
    pred = d["efficiency_predicted"]          # config 4 as written gains ~1.2x from 8 GPUs: said up front
    assert pred is not None and 0.05 < pred["eff_s"] < 0.5 and pred["eff_w"] == 1.0
    assert 1.0 < pred["gain_over_one_gpu"] < 4.0
    assert d["weak_point_512_loops_per_gpu"]["total_loops"] == 4096
    runs = d["runs"]
    assert runs["n"] >= 2 and runs["min"] <= runs["p25"] <= runs["p75"] <= runs["max"]
    # the backend's own view of the job: eight ranks, each with its own slice of the host cores
    assert d["ranks"] == {"backend": "gloo", "backend_world_size": 8}
    census = full["ranks"]
    assert census["backend_world_size"] == 8 and census["backend"] == "gloo"      # (dry run: gloo)
    assert sorted(r["rank"] for r in census["ranks"]) == list(range(8))
    assert len({r["pid"] for r in census["ranks"]}) == 8
    slices = [tuple(r["host_cores"]) for r in census["ranks"] if r["host_cores"]]
    if len(os.sched_getaffinity(0)) >= 8:          # enough cores for disjoint slices
        assert len(slices) == 8 and len(set(c for s in slices for c in s)) == sum(len(s) for s in slices)
    flow = full["flow_wall_s"]
    # by construction: repeats are bounded by --repeat-budget-s (120 s), the reference points by
    # 2 x 21 engines of the same few steps -- far inside the driver's 600 s
    assert flow["repeated_timed_regions"] < 150 and flow["efficiency_reference_runs"] < 150
    assert "repeats <=" in flow["bounds"]


def test_a_rank_that_dies_early_does_not_hang_the_parent():
    """A rank failing before the rendezvous must end the run at once (not at the process-group
    timeout): rank 1 is told a world size that --gpus contradicts and exits."""
    import time
    t0 = time.time()
    env = _plain_env()
    env["BORE_BENCH_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=200, cwd=ROOT)
    assert p.returncode != 0 and "rank 1 exited" in p.stderr
    assert time.time() - t0 < 60


def test_per_gpu_loops_is_weak_scaling():
    p = _run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", "--loops", "8",
             "--no-efficiency")
    assert p.returncode == 0, p.stderr[-2000:]
    d = _record(p)
    assert d["scaling"] == "weak" and d["config"]["loops_per_gpu"] == 8
    assert d["config"]["total_loops"] == 16 and "efficiency" not in d


def test_torchrun_style_launch_still_works():
    """Under a launcher (RANK / WORLD_SIZE set) the script is the rank it is told to be."""
    env = _plain_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run",
                        "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _record(p)
    assert d["n_gpus"] == 1 and d["config"]["loops_per_gpu"] == 512 and d["scaling"] == "weak"


def test_one_gpu_record_is_compact_whatever_the_detail(tmp_path):
    """--gpus 1 as the driver starts it (dry run), and the compact record built from the LARGEST detail this repo has
    seen: the committed round-5 line (28 KB, the one the driver could not parse)."""
    p = _run("--gpus", "1", "--dry-run", "--steps", "3", "--warmup", "1", "--detail-dir", str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    d, full = _record(p, tmp_path)
    assert d["n_gpus"] == 1 and full["config"]["loops_per_gpu"] == 512
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r5", "bench_driver_form.json")) as f:
        big = json.load(f)
    assert len(json.dumps(big)) > 20000
    text = bench.compact_line(big, "bench_detail_n1.json")
    assert len(text.encode()) < 4096
    c = json.loads(text)
    assert c["value"] == pytest.approx(big["value"], rel=1e-5) and c["roofline"]["frac"] == pytest.approx(big["roofline"]["frac"], rel=1e-5)
    assert c["cpu_baseline"]["cores"] == big["cpu_baseline"]["cores"] and c["cpu_baseline"]["kind"] == "port"
    assert set(c["configs"]) == set(big["configs"])           # one figure per BASELINE config survives
    # a detail ten times larger still gives a record under the limit: optional blocks are dropped, never the contract's keys
    big["configs"] = {f"{k}_{i}": v for i in range(40) for k, v in big["configs"].items()}
    c = json.loads(bench.compact_line(big, "x.json"))
    assert len(json.dumps(c)) < 4096 and "roofline" in c and "cpu_baseline" in c and "configs" not in c


def test_bad_split_is_refused_before_any_rank_starts():
    p = _run("--gpus", "3", "--dry-run")
    assert p.returncode != 0 and "does not divide" in p.stderr


def test_failed_rank_fails_the_run():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU: the ranks fail at torch.cuda.set_device")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0")
    assert p.returncode != 0
    assert "rank 0 exited" in p.stderr or "rank 1 exited" in p.stderr


def test_under_torch_distributed_run_as_the_driver_launches_n_ranks(tmp_path):
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- every rank runs the script, rank 0 alone prints, and what it prints last
    is the compact record."""
    port = str(29000 + os.getpid() % 2000)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--dry-run", "--steps", "2", "--warmup", "0", "--detail-dir", str(tmp_path)],
                       env=_plain_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d, full = _record(p, tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["loops_per_gpu"] == 256
    assert full["ranks"]["backend_world_size"] == 2 and "efficiency" in d
