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


@pytest.mark.parametrize("world", [2, 3])
def test_wavelength_sharded_wct(tmp_path, world):
    """Config 5's multi-GPU form: planes sharded over ranks, forward without exchange, adjoint / fwadj with
    one all-reduce of [T, Na, Nb]."""
    out = str(tmp_path / "wct")
    env = dict(os.environ, WCT_OUT=out, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = 29900 + os.getpid() % 400 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "wct_worker.py")]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import wct_worker as ww
    from oracle import surfh_oracle as orc
    psfs, specs, shape, pce, x, y = ww.inputs()
    full = orc.WCTOracle(psfs, specs, shape, pce)
    parts = [np.load(f"{out}.{r}.npz") for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == specs.shape[1] and all(parts[r]["hi"] == parts[r + 1]["lo"] for r in range(world - 1))
    cube = np.concatenate([p["cube"] for p in parts])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)   # noqa: E731
    assert rel(cube, full.forward(x)) < 1e-12
    for p in parts:                                                  # replicated results after the all-reduce
        assert rel(p["adj"], full.adjoint(y)) < 1e-12 and rel(p["fwadj"], full.fwadj(x)) < 1e-12
        assert np.array_equal(p["adj"], parts[0]["adj"])
