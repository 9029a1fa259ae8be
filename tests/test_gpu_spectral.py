"""The solver's vectors in the Fourier domain of the maps (surfh_normal_spec_dev and friends, include/surfh_amd.h): the
building blocks against their map-domain twins and the CG loop of ``DistributedFusion`` in both bases on BASELINE config 2
(band 2A, 251 x 251 x 1024).  Reference: the operator and the priors of surfh/Simulation/fusion_CT.py:16-43,118-162; the
basis change is exact mathematics (unitary transforms, circular differences are diagonal), so both loops follow qmm.lcg."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def fus2():
    import torch
    from surfh_amd import synth
    from surfh_amd.fusion import DistributedFusion
    prob = synth.config2()
    fus = DistributedFusion(prob, rank=0, world=1, device=0)
    assert fus.spec, "config 2 on one MI355X runs the fused transform passes: the spectral-domain loop must be on"
    yield prob, fus, torch
    fus.model.close()


def test_spectral_building_blocks(fus2):
    prob, fus, torch = fus2
    m = fus.model
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    x = torch.as_tensor(rng.standard_normal(m.ishape).astype(np.float32), device=dev)
    z = torch.as_tensor(rng.standard_normal(m.ishape).astype(np.float32), device=dev)
    xt, zt = (torch.zeros(m.spec_size, dtype=torch.float32, device=dev) for _ in range(2))
    m.to_spec_dev(x, xt)
    m.to_spec_dev(z, zt)
    back = torch.empty_like(x)
    m.from_spec_dev(xt, back)
    torch.cuda.synchronize()
    assert rel(back.cpu().numpy(), x.cpu().numpy()) < 2e-6                       # round trip
    n, nv = int(np.prod(m.ishape)), m.spec_size
    d_sp, d_map = m.dot_dev(xt, zt, nv), m.dot_dev(x, z, n)
    assert abs(d_sp - d_map) < 1e-5 * np.sqrt(m.dot_dev(x, x, n) * m.dot_dev(z, z, n))   # orthonormal basis: same inner product
    # normal operator + prior in both bases
    mu, mu_reg = 1.0, 5e3
    q = torch.empty_like(x)
    m.normal_dev(x, q, mu)
    m.prior_add_dev(x, q, mu_reg)
    qt = torch.empty_like(xt)
    m.normal_spec_dev(xt, qt, mu, mu_reg)
    qb = torch.empty_like(x)
    m.from_spec_dev(qt, qb)
    # the two halves separately, prior as its own kernel (the multi-GPU form)
    y = torch.empty(m.osize, dtype=torch.float32, device=dev)
    m.forward_spec_dev(xt, y)
    y_map = torch.empty_like(y)
    m.forward_dev(x, y_map)
    qt2 = torch.empty_like(xt)
    m.adjoint_spec_dev(y, qt2, mu)
    m.prior_spec_add_dev(xt, qt2, mu_reg)
    torch.cuda.synchronize()
    e = dict(normal=rel(qb.cpu().numpy(), q.cpu().numpy()), forward=rel(y.cpu().numpy(), y_map.cpu().numpy()),
             halves=rel(qt2.cpu().numpy(), qt.cpu().numpy()))
    print("spectral building blocks:", e)
    assert max(e.values()) < 5e-6, e


def test_spectral_cg_matches_map_domain_cg(fus2, monkeypatch):
    prob, fus, torch = fus2
    from surfh_amd.fusion import DistributedFusion
    y = fus.make_data(prob["maps"])
    res_s = fus.lcg(y, mu=1.0, mu_reg=5e3, max_iter=10, check_every=100)
    monkeypatch.setenv("SURFH_SPECTRAL_CG", "0")
    ref = DistributedFusion(prob, rank=0, world=1, device=0)
    try:
        assert not ref.spec
        res_m = ref.lcg(y, mu=1.0, mu_reg=5e3, max_iter=10, check_every=100)
    finally:
        ref.model.close()
    gs, gm = np.asarray(res_s.grad_norm), np.asarray(res_m.grad_norm)
    e_g, e_x = float(np.max(np.abs(gs - gm) / gm)), rel(res_s.x, res_m.x)
    print(f"CG in the Fourier domain of the maps vs on the maps, 10 iterations: r.r within {e_g:.2e}, x within {e_x:.2e}")
    assert gs.shape == gm.shape and e_g < 1e-4 and e_x < 1e-4


def test_spectral_domain_on_cooley_tukey_passes():
    """An image size beyond the LDS-resident passes (300 = 4 x 75: dft_ct.hip): the spectral-domain calls run there too -- the
    mix loader takes the solver's scaled spectra, the adjoint's reduction kernel writes them (scale, mu and the quadratic prior
    folded in).  Building blocks against the map-domain calls, ten CG iterations in both bases."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dist_worker import small_problem
    from surfh_amd.fusion import DistributedFusion
    prob = small_problem(300)
    fus = DistributedFusion(prob, rank=0, world=1, device=0)
    m = fus.model
    try:
        assert fus.spec, "N = 300 runs the Cooley-Tukey passes: the spectral-domain loop must be on"
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(8)
        x = torch.as_tensor(rng.standard_normal(m.ishape).astype(np.float32), device=dev)
        xt = torch.zeros(m.spec_size, dtype=torch.float32, device=dev)
        m.to_spec_dev(x, xt)
        mu, mu_reg = 1.0, 50.0
        q = torch.empty_like(x)
        m.normal_dev(x, q, mu)
        m.prior_add_dev(x, q, mu_reg)
        qt = torch.empty_like(xt)
        m.normal_spec_dev(xt, qt, mu, mu_reg)
        qb = torch.empty_like(x)
        m.from_spec_dev(qt, qb)
        y, y_map = (torch.empty(m.osize, dtype=torch.float32, device=dev) for _ in range(2))
        m.forward_spec_dev(xt, y)
        m.forward_dev(x, y_map)
        qt2 = torch.empty_like(xt)
        m.adjoint_spec_dev(y, qt2, mu)
        m.prior_spec_add_dev(xt, qt2, mu_reg)
        torch.cuda.synchronize()
        e = dict(normal=rel(qb.cpu().numpy(), q.cpu().numpy()), forward=rel(y.cpu().numpy(), y_map.cpu().numpy()),
                 halves=rel(qt2.cpu().numpy(), qt.cpu().numpy()))
        print("spectral building blocks, N = 300:", e)
        assert max(e.values()) < 5e-6, e
        yd = fus.make_data(prob["maps"])
        res_s = fus.lcg(yd, mu=1.0, mu_reg=50.0, max_iter=10, check_every=100)
    finally:
        m.close()
    os.environ["SURFH_SPECTRAL_CG"] = "0"
    try:
        ref = DistributedFusion(prob, rank=0, world=1, device=0)
        try:
            assert not ref.spec
            res_m = ref.lcg(yd, mu=1.0, mu_reg=50.0, max_iter=10, check_every=100)
        finally:
            ref.model.close()
    finally:
        del os.environ["SURFH_SPECTRAL_CG"]
    gs, gm = np.asarray(res_s.grad_norm), np.asarray(res_m.grad_norm)
    e_it, e_x = np.abs(gs - gm) / gm, rel(res_s.x, res_m.x)
    print("N = 300, CG in the Fourier domain vs on the maps, r.r per iteration:", " ".join(f"{v:.1e}" for v in e_it), f"; x within {e_x:.2e}")
    # this small problem (96 planes, two 0.6" fields of view, mu_reg = 50) is far worse conditioned than config 2 and r.r falls
    # by eight decades in ten iterations: the two fp32 loops separate as every CG does (tests/test_gpu_distributed.py), fastest
    # at the end; the first iterations carry the comparison
    assert gs.shape == gm.shape and np.max(e_it[:4]) < 1e-4 and np.max(e_it[:9]) < 1e-2 and np.max(e_it) < 0.5 and e_x < 5e-3
