"""Multi-process run of the sharded CG with the real HIP operator (every rank on the one GPU of the box, gloo collectives):
what the gloo CPU tests cannot cover -- the lambda-split branch of ``DistributedFusion.normal`` (forward, group-local
all-reduce of the partial y, adjoint, global all-reduce), the ("planes", a, b) plans of ``partition_balanced``, the
``new_group`` ordering, and the device-resident CG scalars (no host synchronisation per iteration).  RCCL itself is not
exercised: it refuses two ranks on one device ("Multi-GPU")."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,split,refresh,n_pix,partition",
                         [(2, "lambda", 50, 48, None), (3, "lambda", 4, 48, None), (3, "lambda", 4, 48, "balanced"), (3, "pointing", 50, 48, None),
                          (2, "lambda", 50, 128, None), (3, "lambda", 4, 128, "balanced")],
                         ids=["w2-lambda", "w3-lambda-refresh4", "w3-lambda-chunks-refresh4", "w3-pointing", "w2-lambda-spectral",
                              "w3-lambda-chunks-refresh4-spectral"])
def test_sharded_cg_on_the_hip_operator(tmp_path, world, split, refresh, n_pix, partition):
    """n_pix = 128: the ranks' solver vectors are the maps' scaled half spectra (band sharding: normal_spec + all-reduce +
    spectral prior; lambda split: forward_spec, group all-reduce of y, adjoint_spec); the reference stays surfh_cg on maps."""
    out = str(tmp_path / "dist.npz")
    iters = 9
    env = dict(os.environ, DIST_OUT=out, DIST_SPLIT=split, DIST_ITERS=str(iters), DIST_REFRESH=str(refresh), DIST_NPIX=str(n_pix),
               MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", SURFH_REHEARSAL="1")
    if partition:                  # "balanced": the equal-cost contiguous chunks, i.e. ("planes", a, b) plans and ranks in two band groups
        env["SURFH_PARTITION"] = partition
    port = 29300 + os.getpid() % 400 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    assert bool(got["same"]) and bool(got["nosync"])           # replicated bit-identically; scalars stayed on the device
    assert bool(got["spec"]) == (n_pix >= 127)
    print(world, split, str(got["assignment"]), "groups on rank 0:", int(got["n_groups"]), "most on a rank:", int(got["n_groups_max"]), flush=True)
    if split == "lambda" and world == 3:
        assert int(got["n_groups_max"]) >= 1                     # a band is shared: the group-local collective ran
    if partition == "balanced":
        assert "planes" in str(got["assignment"])

    # single-process reference on the same GPU: the library's own CG on the unsharded operator
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as dw
    from surfh_amd.models import spectroSigRLSCT
    prob = dw.small_problem(n_pix)
    m = spectroSigRLSCT(prob["sotf"], prob["templates"], prob["alpha_axis"], prob["beta_axis"], prob["wavel"], prob["ifus"],
                        prob["step_deg"], prob["pointings"], with_ref=False)
    y = m.forward(prob["maps"])
    x, gn, nit = m.cg(y, mu=1.0, mu_reg=50.0, x0=None, max_iter=iters, tol=1e-14, refresh=refresh)
    m.close()
    g = got["grad_norm"]
    assert len(g) == len(gn) == iters + 1
    dev = np.abs(g - gn) / gn
    ex = float(np.linalg.norm(got["x"] - x) / np.linalg.norm(x))
    print("sharded vs single-process CG: grad-norm trace deviation per iteration", [f"{d:.1e}" for d in dev], f"x {ex:.2e}", flush=True)
    # the two runs differ only in fp32 summation order (all-reduce of the ranks' partial products); CG amplifies that from one
    # iteration to the next, so the early iterations pin the arithmetic and the late ones the convergence
    assert np.max(dev[:4]) < 1e-4 and np.max(dev) < 0.1 and ex < 1e-3


@pytest.mark.parametrize("force", [1, 2], ids=["allreduce", "allreduce+band-groups"])
def test_rccl_world_of_one(tmp_path, force):
    """RCCL itself, as far as a one-GPU box allows: ``init_process_group("nccl")`` with one rank, ``new_group([0])`` and the
    per-iteration all-reduce(s) of the loop on the plan's stream actually go through the library (SURFH_FORCE_DIST: 1 = the
    all-reduce of the normal-equation product, 2 = also every band treated as shared: forward, group all-reduce of y, adjoint).
    Same iterates as without the collectives; the per-iteration overhead is printed."""
    iters, n_pix = 9, 128
    out_f, out_p = str(tmp_path / "forced.npz"), str(tmp_path / "plain.npz")
    base = dict(os.environ, DIST_SPLIT="lambda", DIST_ITERS=str(iters), DIST_REFRESH="50", DIST_NPIX=str(n_pix), DIST_BACKEND="nccl",
                DIST_TIME="200", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    for out, f in ((out_f, str(force)), (out_p, "0")):
        env = dict(base, DIST_OUT=out, SURFH_FORCE_DIST=f)
        port = 29700 + os.getpid() % 200 + int(f)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
        subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    a, b = np.load(out_f), np.load(out_p)
    assert bool(a["spec"]) and bool(a["nosync"]) and int(a["n_groups"]) == (2 if force == 2 else 0)
    dev = np.abs(a["grad_norm"] - b["grad_norm"]) / b["grad_norm"]
    ex = float(np.linalg.norm(a["x"] - b["x"]) / np.linalg.norm(b["x"]))
    print(f"RCCL world of one, SURFH_FORCE_DIST={force}: {float(a['ms_it']):.3f} ms per iteration against {float(b['ms_it']):.3f} ms "
          f"without collectives (+{(float(a['ms_it']) - float(b['ms_it'])) * 1e3:.0f} us); r.r within {dev.max():.1e}, x within {ex:.1e}", flush=True)
    assert np.max(dev[:4]) < 1e-4 and np.max(dev) < 0.1 and ex < 1e-3
