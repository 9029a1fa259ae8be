"""world_size-2 rehearsal of the channel-sharded CG on CPU with the gloo backend."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,split", [(2, "lambda"), (3, "lambda"), (4, "lambda"), (3, "pointing")])
def test_sharded_cg_matches_single_process(tmp_path, world, split):
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, DIST_OUT=out, DIST_SPLIT=split, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = 29500 + os.getpid() % 400 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    assert bool(got["same"])                      # x is replicated bit-identically on every rank
    print(world, split, str(got["assignment"]), "groups on rank 0:", int(got["n_groups"]))
    if split == "pointing":
        assert int(got["n_groups"]) == 0          # no group-local collective in this layout

    # single-process reference: the checker's lcg on the full (unsharded) operator
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as dw
    from oracle import surfh_oracle as orc
    prob = dw.small_problem()
    full = dw.OracleBackedModel(prob, prob["ifus"], prob["pointings"]).om
    y = full.forward(prob["maps"])
    ref = orc.lcg(full, y, 1.0, 50.0, np.zeros(full.ishape), tol=1e-14, max_iter=6)
    gn, gr = got["grad_norm"], np.array(ref["grad_norm"])
    assert len(gn) == len(gr)
    assert np.max(np.abs(gn - gr) / gr) < 5e-3    # float32 vectors vs float64
    assert np.linalg.norm(got["x"] - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-3
