"""One rank of the multi-process rehearsal of ``DistributedFusion`` with the REAL HIP operator on a one-GPU box:
every rank on device 0, collectives over gloo (RCCL refuses two ranks on one device; production is nccl, one rank per GPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dist_worker as dw  # noqa: E402
from surfh_amd.fusion import DistributedFusion  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    backend = os.environ.get("DIST_BACKEND", "gloo")           # "nccl" (= RCCL): a world of one rank on a one-GPU box
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = dw.small_problem()
    fus = DistributedFusion(prob, rank=rank, world=world, device=0, split=os.environ.get("DIST_SPLIT", "lambda"))
    y = fus.make_data(prob["maps"], noise_rel=0.0)
    res = fus.lcg(y, mu=1.0, mu_reg=50.0, max_iter=int(os.environ.get("DIST_ITERS", "6")), tol=1e-14,
                  refresh=int(os.environ.get("DIST_REFRESH", "50")))
    ms_it = -1.0
    if os.environ.get("DIST_TIME"):          # per-iteration time of the loop as configured (collectives included)
        import time
        fus.start(y, 1.0, 50.0)
        for _ in range(5):
            fus.step()
        fus._sync()
        t0 = time.perf_counter()
        n = int(os.environ["DIST_TIME"])
        for _ in range(n):
            fus.step()
        fus._sync()
        torch.cuda.synchronize()
        ms_it = (time.perf_counter() - t0) / n * 1e3
    xs = [torch.zeros_like(fus.x) for _ in range(world)]
    dist.all_gather(xs, fus.x)
    same = all(torch.equal(xs[0], t) for t in xs)
    ng = torch.tensor([len(fus.unit_groups)], dtype=torch.int64, device="cuda:0" if backend == "nccl" else "cpu")
    dist.all_reduce(ng, op=dist.ReduceOp.MAX)          # band groups on the rank that has most
    if rank == 0:
        np.savez(os.environ["DIST_OUT"], x=res.x, grad_norm=np.array(res.grad_norm), same=same,
                 n_groups=len(fus.unit_groups), n_groups_max=int(ng.item()), assignment=np.array(repr(fus.assignment)), nosync=bool(fus._nosync), spec=bool(fus.spec),
                 ms_it=ms_it)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
