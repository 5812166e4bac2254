"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job exercising the
N>1 path of bench.py (loop sharding + the single result gather) without a GPU."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bore_amd.engine import gather_results, shard_loop_ids  # noqa: E402


class StubEngine:
    """What gather_results needs from an engine: loop ids, per-loop best (x, y), a device."""

    def __init__(self, loop_ids):
        self.loop_ids = loop_ids
        self.device = torch.device("cpu")

    def best(self):
        x = np.stack([np.array([i * 0.5, i * 0.25]) for i in self.loop_ids])
        return x, self.loop_ids.astype(np.float64) * -1.0


def main():
    out_path, loops = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    eng = StubEngine(shard_loop_ids(rank, world, loops))
    # max-over-ranks timing reduction, as bench.py does it
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res = gather_results(eng, world)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(dict(tmax=float(t[0]), rows=res.tolist(), world=world), f)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
