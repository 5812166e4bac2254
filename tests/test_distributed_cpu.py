"""The N>1 path on CPU: world_size-2 gloo job (the GPU bench uses the same code over RCCL).
Loops are sharded in contiguous blocks, nothing is exchanged on the data path, results are
gathered once at rank 0."""
import json
import os
import subprocess
import sys

import numpy as np

from bore_amd.engine import shard_loop_ids

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shards_are_disjoint_and_complete():
    for world in (1, 2, 4, 8):
        ids = np.concatenate([shard_loop_ids(r, world, 64) for r in range(world)])
        assert np.array_equal(ids, np.arange(world * 64))
    assert np.array_equal(shard_loop_ids(3, 8, 64), np.arange(192, 256))


def test_two_rank_gloo_gather(tmp_path):
    out = tmp_path / "gather.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(HERE, "dist_worker.py"), str(out), "5"]
    subprocess.run(cmd, check=True, env=env, timeout=240)
    res = json.loads(out.read_text())
    rows = np.array(res["rows"])
    assert res["world"] == 2 and res["tmax"] == 2.0            # MAX over ranks
    assert rows.shape == (10, 4)
    assert np.array_equal(rows[:, 0], np.arange(10))           # every loop once, sorted by id
    assert np.allclose(rows[:, 1], rows[:, 0] * 0.5) and np.allclose(rows[:, 3], -rows[:, 0])
