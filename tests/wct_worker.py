"""One rank of the wavelength-sharded Fourier-domain model on CPU (gloo): the checker stands in for the HIP operator."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import surfh_oracle as orc                       # noqa: E402  (test infrastructure)
from surfh_amd.mixing import ShardedWCT                        # noqa: E402


def inputs():
    rng = np.random.default_rng(11)
    L, T, shape = 26, 3, (40, 36)
    psfs = orc.gaussian_psf(np.linspace(7, 8, L), 0.025)[:, 12:29, 12:29]
    specs = rng.random((T, L)) + 0.5
    pce = rng.random(L) + 0.5
    x = rng.random((T,) + shape)
    y = rng.standard_normal((L,) + shape)
    return psfs, specs, shape, pce, x, y


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    psfs, specs, shape, pce, x, y = inputs()
    m = ShardedWCT(psfs, specs, shape, pce, rank, world, model_factory=orc.WCTOracle)
    cube = m.forward(x)
    adj = m.adjoint(y[m.lo:m.hi])
    hx = m.fwadj(x)
    np.savez(os.environ["WCT_OUT"] + f".{rank}.npz", lo=m.lo, hi=m.hi, cube=cube, adj=adj, fwadj=hx)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
