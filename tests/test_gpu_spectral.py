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
