"""A rank of a stand-in job for compairr_amd.dist.spawn_ranks (CPU, gloo): what bench.py's ranks do
around the library -- read RANK / WORLD_SIZE / MASTER_* from the environment, form the group, sum a
matrix over the ranks, rank 0 prints ONE JSON line -- without a GPU.  `--fail R` makes rank R exit 3."""

import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    if "--fail" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail") + 1]):
        sys.exit(3)
    dist.init_process_group("gloo")
    t = torch.full((4,), rank + 1, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"ranks_seen": dist.get_world_size(), "sum": int(t[0]),
                          "argv": sys.argv[1:]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
