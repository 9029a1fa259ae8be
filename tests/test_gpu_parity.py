"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden vectors
of the real reference.  Everything here needs a real MI355X (`-m gpu`).

Tolerances: index/integer tables are bit-exact (checked on CPU in test_host_geometry.py);
the float path is <= 1e-5 relative L2 against the float64 oracle (north_star), the dot-test
gap is < 1e-6 with fp64-accumulated inner products.
"""
import json
import os

import numpy as np
import pytest

import problems
from helpers import build_model, rel
from oracle import surfh_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
OUT = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
TOL = 1e-5


def note(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_log.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **{k: (float(v) if (np.isscalar(v) and not isinstance(v, str)) else v) for k, v in kw.items()})) + "\n")


# ----------------------------------------------------------------------------------------------
def test_gemm_mfma_selftest():
    from surfh_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    for (M, N, K, sk) in [(64, 64, 16, 1), (128, 128, 64, 1), (64, 128, 48, 1), (192, 64, 160, 1),
                          (128, 192, 256, 2), (256, 128, 384, 3)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        # asymmetric structure catches transposed fragment maps
        A += np.arange(M, dtype=np.float32)[:, None] * 0.01
        B += np.arange(N, dtype=np.float32)[None, :] * 0.02
        Cg = np.empty((M, N), dtype=np.float32)
        _lib.check(L.surfh_gemm_selftest(0, M, N, K, sk, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cg)))
        Cr = A.astype(np.float64) @ B.astype(np.float64)
        e = rel(Cg, Cr)
        note("gemm", M=M, N=N, K=K, sk=sk, err=e)
        assert e < 2e-6, (M, N, K, sk, e)


def test_gemm_two_piece_fp16_selftest():
    """The two-piece fp16 GEMM (gemm_cc16.hip, the 256x256 all-consumer tile with both operands delivered as pieces by
    LDS-DMA): power-of-two operand scales, round-to-nearest split, three products; ragged tiles, split K, K slabs of 1, 2
    and 33 steps.  Element magnitudes spread over 12 decades within the operand (entries far
    below the operand's largest magnitude keep their absolute, not their relative, precision), an all-zero operand, and
    non-negative operands where a truncating split would show a bias."""
    from surfh_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(2)
    os.environ["SURFH_SELFTEST_F16X2"] = "1"
    try:
        for (M, N, K, sk) in [(128, 128, 32, 1), (128, 256, 64, 1), (256, 384, 512, 2), (384, 640, 1056, 3), (64, 128, 96, 1), (1664, 1408, 2112, 2)]:
            A = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-6, 6, (M, K)))).astype(np.float32)
            B = rng.standard_normal((K, N)).astype(np.float32) + np.arange(N, dtype=np.float32)[None, :] * 0.02
            for scale in (1.0, 1e20, 1e-20):
                As = (A * np.float32(scale)).astype(np.float32)
                Cg = np.empty((M, N), dtype=np.float32)
                _lib.check(L.surfh_gemm_selftest(0, M, N, K, sk, _lib.fptr(As), _lib.fptr(B), _lib.fptr(Cg)))
                e = rel(Cg, As.astype(np.float64) @ B.astype(np.float64))
                note("gemm_f16x2_pc", M=M, N=N, K=K, sk=sk, scale=scale, err=e)
                assert e < 5e-7, (M, N, K, sk, scale, e)
        M, N, K = 256, 256, 4096
        A = rng.random((M, K)).astype(np.float32) * 3e4
        B = (rng.random((K, N)) * 0.1).astype(np.float32)
        Cg = np.empty((M, N), dtype=np.float32)
        _lib.check(L.surfh_gemm_selftest(0, M, N, K, 1, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cg)))
        ref = A.astype(np.float64) @ B.astype(np.float64)
        bias = float(np.mean((Cg - ref) / ref))
        note("gemm_f16x2_pc_nonneg", err=rel(Cg, ref), bias=bias)
        assert rel(Cg, ref) < 1e-6 and abs(bias) < 3e-7          # one 4096-long fp32 accumulation chain (bf16 split: -8e-6 at 3072)
        # one scale per row of the data operand: a hot row (1e8 x the others) does not cost the faint rows their precision
        Ah = A.copy()
        Ah[7] *= np.float32(1e8)
        _lib.check(L.surfh_gemm_selftest(0, M, N, K, 1, _lib.fptr(Ah), _lib.fptr(B), _lib.fptr(Cg)))
        refh = Ah.astype(np.float64) @ B.astype(np.float64)
        per_row = np.linalg.norm(Cg - refh, axis=1) / np.linalg.norm(refh, axis=1)
        note("gemm_f16x2_pc_hot_row", worst_row=float(per_row.max()))
        assert per_row.max() < 1e-6
        Z = np.zeros((M, K), dtype=np.float32)
        _lib.check(L.surfh_gemm_selftest(0, M, N, K, 1, _lib.fptr(Z), _lib.fptr(B), _lib.fptr(Cg)))
        assert not Cg.any()
    finally:
        os.environ.pop("SURFH_SELFTEST_F16X2")


def test_gemm_two_piece_fp16_k_step_classes():
    """The same kernel with K-step lists (gemm_cc16.hip "K-step classes"; plan.hip build_klist with the plan's tolerances): a
    constant operand shaped like the spectral response -- sinc^2 around a diagonal, rows normalised, several beta columns side by
    side -- has most of its K steps in the far class (leading fp16 product only); the result stays at the error of the
    three-product kernel.  An operand without structure gets no far steps and runs through the list path unchanged.  Split K
    divides both lists among the slabs."""
    from surfh_amd import _lib
    import ctypes
    L = _lib.load()
    rng = np.random.default_rng(3)
    os.environ["SURFH_SELFTEST_F16X2"] = "2"
    try:
        ks = (ctypes.c_int64 * 2)()
        for (M, N, Lin, ncol, sk) in [(256, 1024, 1152, 2, 1), (384, 1408, 1152, 3, 2), (128, 640, 704, 1, 1)]:
            K = ncol * Lin
            lo, li = np.arange(N)[:, None], np.arange(Lin)[None, :]
            cols = []
            for c in range(ncol):
                z = (li - (lo * (Lin / N) + 3.0 * c)) / 2.3
                w = np.sinc(z) ** 2
                cols.append(w / w.sum(axis=1, keepdims=True))
            W = np.concatenate(cols, axis=1)                               # [N][K]
            B = np.ascontiguousarray(W.T).astype(np.float32)               # the hook takes B as [K][N]
            for name, A in (("nonneg", rng.random((M, K)) * 50 + 1), ("randn", rng.standard_normal((M, K)))):
                A = A.astype(np.float32)
                Cg = np.empty((M, N), dtype=np.float32)
                _lib.check(L.surfh_gemm_selftest(0, M, N, K, sk, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cg)))
                L.surfh_gemm_selftest_ksteps(ks)
                e = rel(Cg, A.astype(np.float64) @ B.astype(np.float64))
                note("gemm_f16x2_klist", M=M, N=N, K=K, sk=sk, data=name, near=int(ks[0]), far=int(ks[1]), err=e)
                assert ks[1] > ks[0] and e < 1e-6, (M, N, K, sk, name, list(ks), e)
        # the adjoint's shape: the constant operand has one row per (beta column, wavelength), K runs over the detector axis; a tile
        # is 64 wavelengths of four neighbouring columns (GemmArgs::permP), incl. a last group with missing columns (5 = 4 + 1)
        for (M, Ldet, Lin, ncol) in [(256, 1024, 1152, 4), (128, 640, 768, 5)]:
            lo, li = np.arange(Ldet)[:, None], np.arange(Lin)[None, :]
            cols = []
            for c in range(ncol):
                w = np.sinc((li - (lo * (Lin / Ldet) + 3.0 * c)) / 2.3) ** 2
                cols.append(w / w.sum(axis=1, keepdims=True))
            Wt = np.ascontiguousarray(np.concatenate(cols, axis=1)).astype(np.float32)      # [K = Ldet][N = ncol * Lin], as the hook wants B
            N = ncol * Lin
            A = (rng.random((M, Ldet)) * 50 + 1).astype(np.float32)
            Cg = np.empty((M, N), dtype=np.float32)
            os.environ["SURFH_SELFTEST_PERM"] = str(Lin)
            try:
                _lib.check(L.surfh_gemm_selftest(0, M, N, Ldet, 1, _lib.fptr(A), _lib.fptr(Wt), _lib.fptr(Cg)))
            finally:
                os.environ.pop("SURFH_SELFTEST_PERM")
            L.surfh_gemm_selftest_ksteps(ks)
            e = rel(Cg, A.astype(np.float64) @ Wt.astype(np.float64))
            note("gemm_f16x2_klist_perm", M=M, N=N, K=Ldet, near=int(ks[0]), far=int(ks[1]), err=e)
            assert ks[1] > ks[0] and e < 1e-6, (M, N, Ldet, list(ks), e)
        M, N, K = 256, 384, 1056
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        Cg = np.empty((M, N), dtype=np.float32)
        _lib.check(L.surfh_gemm_selftest(0, M, N, K, 3, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cg)))
        L.surfh_gemm_selftest_ksteps(ks)
        e = rel(Cg, A.astype(np.float64) @ B.astype(np.float64))
        assert ks[1] == 0 and e < 5e-7, (list(ks), e)
    finally:
        os.environ.pop("SURFH_SELFTEST_F16X2")


@pytest.fixture(scope="module")
def c1():
    cfg = problems.config1()
    om = problems.oracle_model(cfg, box="direct")
    m = build_model(cfg)
    yield cfg, om, m
    m.close()


def oracle_xs(om, tab, blurred):
    """The R-GEMM operand the fused S+box+L+decimation kernel must produce: [(l,b'), (p,s,a)]."""
    Lin = tab.wslice[1] - tab.wslice[0]
    nbs = tab.npix_slit_beta_width
    P, S, _, aout = tab.oshape
    out = np.zeros((Lin, nbs, P, S, aout))
    sub = blurred[tab.wslice[0]:tab.wslice[1]]
    for p in range(P):
        sc = orc.box_sum_direct(tab, orc.gridding(tab, sub, p))
        for s in range(S):
            sl = orc.slicing(tab, sc, s)[:, : aout * tab.srf: tab.srf, :]     # [Lin, aout, nbs]
            out[:, :, p, s, :] = np.transpose(sl, (0, 2, 1))
    return out.reshape(Lin * nbs, P * S * aout)


def test_stage_blurred_cube(c1):
    cfg, om, m = c1
    m.forward(cfg["maps"])
    N = cfg["N"]
    lo, hi = int(m.debug_buffer("info")[0]), int(m.debug_buffer("info")[1])
    b = m.debug_buffer("blurred").transpose(2, 1, 0)[: hi - lo, :N, :N]     # device layout is [beta][alpha][lambda]
    ref = om.blur(cfg["maps"])[lo:hi]
    e = rel(b, ref)
    note("blurred", err=e)
    assert e < TOL
    g = np.load(os.path.join(G, "config1_chain.npz"))
    sel = [k for k in g["lam_sel"] if lo <= k < hi]
    assert rel(b[[k - lo for k in sel]], g["blurred_sel"][: len(sel)]) < TOL       # the real reference's planes


def test_stage_gemm_operand(c1):
    cfg, om, m = c1
    m.forward(cfg["maps"])
    tab = om.channels[0]
    ref = oracle_xs(om, tab, om.blur(cfg["maps"]))                     # [(l,b'), (p,s,a)]
    LinP, shift, nbs, Lin = (int(v) for v in m.debug_buffer("xsinfo:0"))
    xs = m.debug_buffer("xs:0")                                         # device layout [(p,s,a)][b'][LinP]
    ref = ref.reshape(Lin, nbs, -1).transpose(2, 1, 0)
    xs = xs[: ref.shape[0], :, shift: shift + Lin]
    e = rel(xs, ref)
    note("xs", err=e)
    assert e < TOL


def test_forward_vs_oracle_and_reference(c1):
    cfg, om, m = c1
    y = m.forward(cfg["maps"])
    e = rel(y, om.forward(cfg["maps"]))
    g = np.load(os.path.join(G, "config1_chain.npz"))
    eg = rel(y, g["y"])
    note("forward", err=e, err_golden=eg)
    assert y.shape == (om.osize,) and y.dtype == np.float64
    assert e < TOL and eg < TOL


def test_adjoint_exact_vs_oracle(c1):
    cfg, om, m = c1
    u = np.random.default_rng(1).standard_normal(om.osize)
    a = m.adjoint(u)
    e = rel(a, om.adjoint(u))
    note("adjoint", err=e)
    assert a.shape == om.ishape
    assert e < TOL


def test_adjoint_ref_vs_reference(c1):
    cfg, om, m = c1
    g = np.load(os.path.join(G, "config1_chain.npz"))
    u = np.random.default_rng(int(g["u_seed"])).standard_normal(om.osize)
    a = m.adjoint_ref(u)
    e = rel(a, g["adjoint_ref"])
    note("adjoint_ref", err=e, err_oracle=rel(a, om.adjoint_ref(u)))
    assert e < TOL


def test_dottest(c1):
    cfg, om, m = c1
    from surfh_amd import dotgap, dottest
    gaps, ngaps = [], []
    rng = np.random.default_rng(11)
    for _ in range(9):
        v, u = rng.standard_normal(m.isize), rng.standard_normal(m.osize)
        av = np.asarray(m.matvec(v), dtype=np.float64)
        l, r = float(np.vdot(np.asarray(m.rmatvec(u), dtype=np.float64), v)), float(np.vdot(u, av))
        gaps.append(abs(l - r) / abs(r))
        ngaps.append(abs(l - r) / (np.linalg.norm(u) * np.linalg.norm(av)))
    note("dottest", gaps=[float(x) for x in gaps], normalised=[float(x) for x in ngaps])
    # fp32 production path, zero-mean test vectors (test/sandbox_dottest.py:16-27): <u, A v> is a cancelling sum, so what fp32
    # can honestly hold is the gap against the natural scale |u||Av| of the inner product; the strict ratio test runs on the
    # float64-accumulating verification plan below (test_dottest_randn_strict_on_verification_plan)
    assert max(ngaps) < 1e-6
    assert dottest(m, num=2, rng=rng, rtol=1e-4)
    pg = []
    for _ in range(3):       # non-negative test vectors: no cancellation in <u, A v>, strict < 1e-6
        v, u = rng.random(m.isize), rng.random(m.osize)
        l = float(np.vdot(m.rmatvec(u), v)); r = float(np.vdot(u, m.matvec(v)))
        pg.append(abs(l - r) / abs(r))
    note("dottest_uniform", gaps=[float(x) for x in pg])
    assert max(pg) < 1e-6


def test_dottest_randn_strict_on_verification_plan(c1):
    """The reference's dot test as it stands -- randn u and v, |<A^T u, v> - <u, A v>| / |<u, A v>| < 1e-6
    (test/sandbox_dottest.py:16-27, north_star) -- on the verification plan: the same operator and tables with every long
    sum accumulated in float64 (surfh_config.verify).  The plan must also agree with the production plan and the oracle."""
    cfg, om, m = c1
    mv = build_model(cfg, verify=True)
    try:
        rng = np.random.default_rng(11)
        gaps = []
        for _ in range(9):
            v, u = (rng.standard_normal(n).astype(np.float32).astype(np.float64) for n in (mv.isize, mv.osize))   # what the device sees
            l = float(np.vdot(np.asarray(mv.rmatvec(u), dtype=np.float64), v))
            r = float(np.vdot(u, np.asarray(mv.matvec(v), dtype=np.float64)))
            gaps.append(abs(l - r) / abs(r))
        yv, y = mv.forward(cfg["maps"]), m.forward(cfg["maps"])
        u = rng.standard_normal(mv.osize)
        e = dict(fwd_vs_production=rel(yv, y), fwd_vs_oracle=rel(yv, om.forward(cfg["maps"])),
                 adj_vs_oracle=rel(mv.adjoint(u), om.adjoint(u)), adj_ref_vs_oracle=rel(mv.adjoint_ref(u), om.adjoint_ref(u)))
        note("dottest_verify", gaps=[float(x) for x in gaps], **e)
        assert max(gaps) < 1e-6
        assert max(e.values()) < 2e-6
    finally:
        mv.close()


def test_fwadj_and_linearity(c1):
    cfg, om, m = c1
    rng = np.random.default_rng(3)
    x1, x2 = rng.standard_normal(om.ishape), rng.standard_normal(om.ishape)
    assert rel(m.fwadj(x1), m.adjoint(m.forward(x1))) < 1e-6
    # the normal operator hands the forward GEMM's slab sums straight to the adjoint GEMM as fp16 pieces (ymat16_from_cpart):
    # the same bits as the way through y
    assert np.array_equal(m.fwadj(x1), m.adjoint(m.forward(x1)))
    assert np.array_equal(m.fwadj(x2), m.adjoint(m.forward(x2)))
    assert rel(m.forward(2.0 * x1 - 0.5 * x2), 2.0 * m.forward(x1) - 0.5 * m.forward(x2)) < 1e-5


def test_two_channel_overlap():
    cfg = problems.two_channel_small()
    om = problems.oracle_model(cfg, box="direct")
    m = build_model(cfg)
    g = np.load(os.path.join(G, "two_channel.npz"))
    assert np.array_equal(m._idx, g["idx"])
    y = m.forward(cfg["maps"])
    u = np.random.default_rng(int(g["u_seed"])).standard_normal(om.osize)
    e = dict(fwd=rel(y, g["y"]), adj=rel(m.adjoint(u), om.adjoint(u)), adj_ref=rel(m.adjoint_ref(u), g["adjoint_ref"]))
    note("two_channel", **e)
    assert max(e.values()) < TOL
    from surfh_amd import dotgap
    gaps, pg = [], []
    for k in range(5):
        l, r = dotgap(m, np.random.default_rng(40 + k))
        gaps.append(abs(l - r) / abs(r))
        rng = np.random.default_rng(50 + k)
        v, uu = rng.random(m.isize), rng.random(m.osize)
        l = float(np.vdot(m.rmatvec(uu), v)); r = float(np.vdot(uu, m.matvec(v)))
        pg.append(abs(l - r) / abs(r))
    note("two_channel_dot", randn=[float(x) for x in gaps], uniform=[float(x) for x in pg])
    assert max(pg) < 1e-6                                   # non-negative vectors: strict
    mv = build_model(cfg, verify=True)                       # zero-mean vectors: strict on the float64-accumulating plan
    vg = []
    for k in range(5):
        rng = np.random.default_rng(40 + k)
        v, uu = (rng.standard_normal(n).astype(np.float32).astype(np.float64) for n in (mv.isize, mv.osize))
        l = float(np.vdot(np.asarray(mv.rmatvec(uu), dtype=np.float64), v)); r = float(np.vdot(uu, np.asarray(mv.matvec(v), dtype=np.float64)))
        vg.append(abs(l - r) / abs(r))
    mv.close()
    note("two_channel_dot_verify", randn=[float(x) for x in vg])
    # what is left on the verification plan is the fp32 STORAGE of the intermediates (random, ~6e-8 / sqrt(n) of |u||Av|); on this
    # tiny problem (48 x 48 x 96) a badly cancelling draw can reach 1e-6 -- config 1 and config 2 are asserted < 1e-6 per draw
    assert np.median(vg) < 1e-6 and max(vg) < 5e-6
    m.close()


def test_no_lmm_mode():
    """templates=None switches the LMM off (spectroModel.py:60-63,92-95): input is the cube."""
    cfg = dict(problems.two_channel_small())
    cfg["templates"] = None
    om = problems.oracle_model(cfg, box="direct")
    m = build_model(cfg)
    cube = np.random.default_rng(5).random(om.ishape)
    u = np.random.default_rng(6).standard_normal(om.osize)
    e = dict(fwd=rel(m.forward(cube), om.forward(cube)), adj=rel(m.adjoint(u), om.adjoint(u)))
    note("no_lmm", **e)
    assert max(e.values()) < TOL
    m.close()


def test_cg_matches_oracle_lcg(c1):
    cfg, om, m = c1
    y = om.forward(cfg["maps"])
    y = y + np.random.default_rng(1).standard_normal(y.size) * 1e-2 * np.sqrt(np.mean(y ** 2))
    mu, mur, nit = 1.0, 5e3, 12
    ref = orc.lcg(om, y, mu, mur, np.zeros(om.ishape), tol=1e-12, max_iter=nit)
    x, gn, n = m.cg(y, mu=mu, mu_reg=mur, x0=None, max_iter=nit, tol=1e-12)
    gr = np.array(ref["grad_norm"])
    e = rel(x, ref["x"])
    ge = float(np.max(np.abs(gn - gr) / gr))
    note("cg", err_x=e, err_gradnorm=ge, nit=n, gn_first=float(gn[0]), gn_last=float(gn[-1]))
    assert n == nit and len(gn) == nit + 1
    # fp32 vectors vs the float64 oracle over 12 CG iterations of an ill-conditioned system: the first ten iterates' r.r agree
    # to 2e-5 (measured; asserted 2e-4), the last two amplify the rounding of the recurrences (1e-3, 2e-2, 5e-2 measured)
    assert e < 3e-3 and ge < 0.2 and float(np.max(np.abs(gn[:10] - gr[:10]) / gr[:10])) < 2e-4
    # the same solve on the verification plan (every long sum in float64, vectors still fp32): what is left is the vectors' rounding
    mv = build_model(cfg, verify=True)
    try:
        xv, gv, _ = mv.cg(y, mu=mu, mu_reg=mur, x0=None, max_iter=nit, tol=1e-12)
    finally:
        mv.close()
    ev, gev = rel(xv, ref["x"]), float(np.max(np.abs(gv[:10] - gr[:10]) / gr[:10]))
    note("cg_verify", err_x=ev, err_gradnorm_first10=gev, err_gradnorm_last=float(abs(gv[-1] - gr[-1]) / gr[-1]))
    assert ev < 1e-3 and gev < 5e-5 and float(np.max(np.abs(gv - gr) / gr)) < 5e-2
    c = [orc.crit_val(om, y, m.cg(y, mu=mu, mu_reg=mur, max_iter=k)[0], mu, mur) for k in (1, 4, 8)]
    assert c[0] > c[1] > c[2]


def test_device_pointer_api_and_profile(c1):
    import torch
    cfg, om, m = c1
    dev = torch.device("cuda:0")
    x = torch.tensor(cfg["maps"], dtype=torch.float32, device=dev).contiguous()
    y = torch.empty(om.osize, dtype=torch.float32, device=dev)
    q = torch.empty_like(x)
    torch.cuda.synchronize()
    m.profile_reset()
    m.profile_enable(True)
    m.forward_dev(x, y)
    m.normal_dev(x, q, 1.0)
    prof = m.profile()
    m.profile_enable(False)
    torch.cuda.synchronize()
    assert rel(y.cpu().numpy(), om.forward(cfg["maps"])) < TOL
    assert rel(q.cpu().numpy(), om.adjoint(om.forward(cfg["maps"]))) < TOL
    assert prof["gemm_wblur_fwd"][0] == 2 and prof["gemm_wblur_adj"][0] == 1
    assert all(ms >= 0 for _, ms in prof.values())
    n = om.isize
    assert abs(m.dot_dev(x, x, n) - float(np.sum(cfg["maps"].astype(np.float32).astype(np.float64) ** 2))) < 1e-6 * n


def test_distributed_loop_matches_single_call_cg(c1):
    """DistributedFusion on one rank (the benchmark's loop: device vectors, fused step+direction call with one host
    synchronisation per iteration) gives the same iterates with and without the fused call (bit for bit) and agrees with surfh_cg."""
    import torch
    from surfh_amd.fusion import DistributedFusion
    cfg, om, m = c1
    y = om.forward(cfg["maps"])
    x_ref, gn_ref, _ = m.cg(y, mu=1.0, mu_reg=5e3, x0=None, max_iter=8)
    from helpers import make_ifu, make_pointings
    prob = dict(ifus=[make_ifu(s) for s in cfg["specs"]], pointings=make_pointings(cfg), alpha_axis=cfg["alpha_axis"],
                beta_axis=cfg["beta_axis"], wavel=cfg["wavel"], step_deg=cfg["step_deg"], sotf=cfg["sotf"],
                templates=cfg["templates"])
    out = {}
    for fused in (True, False):
        fus = DistributedFusion(prob, rank=0, world=1, device=0)
        if not fused:
            fus.model.cg_iter_dev = None                    # instance attribute shadows the method: two-call path
        yt = torch.as_tensor(np.ascontiguousarray(y, dtype=np.float32), device="cuda:0")
        res = fus.lcg(yt, mu=1.0, mu_reg=5e3, max_iter=8)
        out[fused] = (np.asarray(res.grad_norm), res.x.reshape(x_ref.shape))
        fus.model.close()
    assert np.array_equal(out[True][0], out[False][0]) and np.array_equal(out[True][1], out[False][1])   # same arithmetic
    assert float(np.max(np.abs(out[True][0] - gn_ref) / gn_ref)) < 1e-5 and rel(out[True][1], x_ref) < 1e-5


def test_quad_criterion_mirror(c1):
    """QuadCriterion_MRS.run_method('lcg') / get_crit_val (surfh/Simulation/fusion_CT.py:66-265) on the device CG."""
    from surfh_amd.fusion import QuadCriterion_MRS
    cfg, om, m = c1
    y = om.forward(cfg["maps"])
    crit = QuadCriterion_MRS(1.0, y, m, 5e3, printing=False, gradient="separated")
    res = crit.run_method("lcg", maximum_iterations=8, tolerance=1e-12, value_init=0.5)
    assert res.x.shape == (om.isize,) and res.nit == 8 and len(res.grad_norm) == 9
    ref = orc.lcg(om, y, 1.0, 5e3, np.ones(om.ishape) * 0.5, tol=1e-12, max_iter=8)
    assert rel(res.x.reshape(om.ishape), ref["x"]) < 5e-3
    c_gpu = crit.get_crit_val(res.x)
    c_ref = orc.crit_val(om, y, ref["x"], 1.0, 5e3)
    c_init = orc.crit_val(om, y, np.ones(om.ishape) * 0.5, 1.0, 5e3)
    assert abs(c_gpu - c_ref) / c_ref < 1e-2 and c_gpu < c_init
    res_m = crit.run_method("mmmg", maximum_iterations=8, value_init=0.5)       # the other solver (tests/test_gpu_driver.py)
    assert res_m.nit == 8 and rel(res_m.x, res.x) < 1e-4


def test_error_behaviour():
    """Failures are loud and carry a message: the reference's own (field of view outside the cube -> ValueError,
    cython_2D_interpolation.py:472-478; wrong input size) and the library's limits."""
    import copy
    cfg = problems.config1()
    m = build_model(cfg)
    with pytest.raises(ValueError, match="expected"):
        m.forward(np.zeros((3, 64, 64)))
    with pytest.raises(ValueError, match="size"):
        m.cg(np.zeros(7), max_iter=1)
    m.close()
    far = copy.deepcopy(cfg)
    far["pointings"] = [[(a + 40 * problems.STEP_DEG, b) for a, b in far["pointings"][0]]]
    with pytest.raises(ValueError, match="out of bounds"):
        build_model(far)
    many = dict(cfg)
    many["templates"] = np.ones((9, cfg["templates"].shape[1]))
    with pytest.raises(ValueError, match="at most 8 templates"):
        build_model(many)
    bad = dict(cfg)
    bad["sotf"] = cfg["sotf"][:, :, :-1]
    with pytest.raises(ValueError, match="sotf shape"):
        build_model(bad)
    with pytest.raises(ValueError, match="without templates"):
        build_model(dict(cfg, sotf=None))


@pytest.mark.parametrize("lmm", [True, False])
def test_disjoint_wavelength_windows(lmm):
    """A plan stores only the union of its channels' windows; here that union has a gap."""
    cfg = dict(problems.two_channel_disjoint())
    if not lmm:
        cfg["templates"] = None
    om = problems.oracle_model(cfg, box="direct")
    w = [c.wslice for c in om.channels]
    assert w[0][1] < w[1][0]                                    # really disjoint
    m = build_model(cfg)
    info = m.debug_buffer("info")
    assert int(info[3]) == 2 and int(info[2]) < w[1][1] - w[0][0]     # two segments, fewer planes than the span
    x = cfg["maps"] if lmm else np.random.default_rng(5).random(om.ishape)
    u = np.random.default_rng(6).standard_normal(om.osize)
    e = dict(fwd=rel(m.forward(x), om.forward(x)), adj=rel(m.adjoint(u), om.adjoint(u)),
             adj_ref=rel(m.adjoint_ref(u), om.adjoint_ref(u)))
    note("disjoint_windows", lmm=lmm, **e)
    assert max(e.values()) < TOL
    m.close()


@pytest.mark.parametrize("env", [{"SURFH_DFT_H2": "0"}, {"SURFH_DFT_DENSE": "1"}, {"SURFH_NO_FUSED_MIX": "1"}, {"SURFH_WBLUR_FP32": "1"},
                                 {"SURFH_OVERLAP": "1"}, {"SURFH_GATHER_SORTED": "0"}, {"SURFH_SCATTER_RMW_ALL": "1"},
                                 {"SURFH_SCATTER_GROUPED": "0"}, {"SURFH_GATHER_GROUPED": "0"}, {"SURFH_ADJ_CLEAR": "1"}],
                         ids=["no_fast_transform_kernel", "dense_dft", "unfused_mix", "wblur_fp32", "two_streams", "gather_rows_unsorted",
                              "scatter_rmw_everywhere", "scatter_row_by_row", "gather_row_by_row", "adjoint_clears_accumulator"])
def test_alternative_kernel_paths(env):
    """The A/B kernel paths kept behind environment switches (read at plan creation) stay parity-green.  Only the switches that
    change something at config 1 (64 x 64 x 128, one wavelength chunk) are listed here; the ones whose fast side needs N >= 127,
    several wavelength chunks or a long detector axis are compared in test_alternative_kernel_paths_where_they_engage."""
    cfg = problems.config1()
    om = problems.oracle_model(cfg, box="direct")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = build_model(cfg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
    try:
        y = m.forward(cfg["maps"])
        ey = rel(y, om.forward(cfg["maps"]))
        rng = np.random.default_rng(5)
        v = rng.random(y.shape)
        ea = rel(m.adjoint(v), om.adjoint(v))
        note("alt_path", env=json.dumps(env), fwd=ey, adj=ea)
        assert ey < TOL and ea < TOL
    finally:
        m.close()


@pytest.fixture(scope="module")
def mid():
    """problems.two_channel_mid with the default plan's outputs and the float64 oracle's."""
    cfg = problems.two_channel_mid()
    om = problems.oracle_model(cfg, box="direct")
    rng = np.random.default_rng(12)
    u = rng.standard_normal(om.osize)
    m = build_model(cfg)
    try:
        state = dict(ksteps=[int(v) for v in m.debug_buffer("ksteps")], otf=[int(v) for v in m.debug_buffer("otf")[:2]],
                     rng=[int(v) for v in m.debug_buffer("range")], spec=bool(m.spec_supported()))
        out = dict(fwd=np.asarray(m.forward(cfg["maps"])), adj=np.asarray(m.adjoint(u)))
        # zero-mean maps: sign-cancelling sums with the far K steps on one fp16 product and the OTF's support lists on
        xr = rng.standard_normal(om.ishape)
        state["fwd_randn"] = rel(m.forward(xr), om.forward(xr))
    finally:
        m.close()
    ref = dict(fwd=om.forward(cfg["maps"]), adj=om.adjoint(u))
    return cfg, u, out, ref, state


def test_default_paths_engage_at_mid_size(mid):
    """Every fast path is really in use on the problem the A/B cases below compare on (at config 1 none of them is: one
    wavelength chunk, N = 64, a detector axis of 48 samples)."""
    cfg, u, out, ref, st = mid
    N = cfg["N"]
    print("mid-size plan:", st)
    assert st["ksteps"][1] > 0 and st["ksteps"][3] > 0                     # far K steps in both spectral-blur GEMMs
    assert 0 < st["otf"][0] < st["otf"][1]                                 # some, not all, super-tiles inside the OTF's support
    assert st["rng"][0] > 0 and st["rng"][1] < N                           # cube columns no channel sees
    assert st["spec"]                                                      # fused adjoint tail (spectral-domain solver calls)
    assert rel(out["fwd"], ref["fwd"]) < TOL and rel(out["adj"], ref["adj"]) < TOL      # adjoint: zero-mean data
    assert st["fwd_randn"] < TOL, st["fwd_randn"]


@pytest.mark.parametrize("env,changes_bits", [({"SURFH_WBLUR_FAR": "0"}, True), ({"SURFH_WBLUR_PERM": "0"}, True),
                                              ({"SURFH_OTF_SUPPORT": "0"}, True), ({"SURFH_OTF_RANGES": "0"}, False),
                                              ({"SURFH_ADJ_FUSED": "0"}, True), ({"SURFH_ALPHA_RANGE": "0"}, False),
                                              ({"SURFH_GEMM_GROUPED": "0"}, False), ({"SURFH_DFT_H2": "0"}, True)],
                         ids=["gemm_three_products_everywhere", "gemm_adjoint_plain_tiles", "whole_spectrum", "otf_support_lists_only",
                              "adjoint_tail_separate", "transform_whole_cube", "adjoint_gemms_one_by_one", "transforms_on_dft_ct_2x64"])
def test_alternative_kernel_paths_where_they_engage(mid, env, changes_bits):
    """A/B of the switches whose fast side needs a problem of some size (tests/problems.py two_channel_mid): both sides within
    1e-5 of the float64 oracle; where the two sides run different arithmetic the outputs must differ in their bits (the switch
    did something), where the fast side only skips exact zeros or regroups launches they must not."""
    cfg, u, out, ref, st = mid
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = build_model(cfg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
    try:
        ks = [int(v) for v in m.debug_buffer("ksteps")]
        otf = [int(v) for v in m.debug_buffer("otf")[:2]]
        rg = [int(v) for v in m.debug_buffer("range")]
        y, a = np.asarray(m.forward(cfg["maps"])), np.asarray(m.adjoint(u))
    finally:
        m.close()
    ey, ea = rel(y, ref["fwd"]), rel(a, ref["adj"])
    same = bool(np.array_equal(y, out["fwd"]) and np.array_equal(a, out["adj"]))
    note("alt_path_mid", env=json.dumps(env), fwd=ey, adj=ea, same_bits=same)
    print(f"{env}: forward {ey:.2e} adjoint {ea:.2e}, same bits as the default plan: {same}; ksteps {ks} otf {otf} range {rg}")
    assert ey < TOL and ea < TOL
    assert same != changes_bits, (env, same)
    if "SURFH_WBLUR_FAR" in env:
        assert ks[1] == 0 and ks[3] == 0
    if "SURFH_WBLUR_PERM" in env:
        assert ks[2:] != st["ksteps"][2:]
    if "SURFH_OTF_SUPPORT" in env:
        assert otf[0] == 0 or otf[0] == otf[1]
    if "SURFH_ALPHA_RANGE" in env:
        assert rg[0] == 0 and rg[1] == cfg["N"]


def test_exact_flags_of_the_config(mid):
    """``surfh_config.exact`` switches the two bounded approximations off per plan, without environment variables: the same
    bits as SURFH_WBLUR_FAR=0 + SURFH_OTF_SUPPORT=0."""
    cfg, u, out, ref, st = mid
    m = build_model(cfg, exact=3)
    try:
        ks, otf = [int(v) for v in m.debug_buffer("ksteps")], [int(v) for v in m.debug_buffer("otf")[:2]]
        y, a = np.asarray(m.forward(cfg["maps"])), np.asarray(m.adjoint(u))
    finally:
        m.close()
    assert ks[1] == 0 and ks[3] == 0 and (otf[0] == 0 or otf[0] == otf[1])
    assert rel(y, ref["fwd"]) < TOL and rel(a, ref["adj"]) < TOL
    old = {k: os.environ.get(k) for k in ("SURFH_WBLUR_FAR", "SURFH_OTF_SUPPORT")}
    os.environ.update(SURFH_WBLUR_FAR="0", SURFH_OTF_SUPPORT="0")
    try:
        m2 = build_model(cfg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
    try:
        assert np.array_equal(np.asarray(m2.forward(cfg["maps"])), y) and np.array_equal(np.asarray(m2.adjoint(u)), a)
    finally:
        m2.close()


@pytest.mark.parametrize("maker", [problems.config1, problems.two_channel_small, problems.two_channel_disjoint],
                         ids=["config1", "overlapping_windows", "disjoint_windows"])
def test_adjoint_accumulator_carries_no_state(maker):
    """The exact adjoint accumulates in a cube that is cleared once, at plan creation: every scatter row stores where no earlier
    channel of the same pass has written and read-modify-writes exactly the wavelengths one has.  Whatever ran before -- other
    data, the reference adjoint, a forward -- the result of a call is bit-identical, and zero data give exactly zero."""
    cfg = maker()
    om = problems.oracle_model(cfg, box="direct")
    m = build_model(cfg)
    try:
        rng = np.random.default_rng(21)
        u1, u2 = rng.standard_normal(m.osize), 1e3 * rng.standard_normal(m.osize)
        a1 = m.adjoint(u1)
        assert rel(a1, om.adjoint(u1)) < TOL
        m.adjoint(u2)
        m.adjoint_ref(u2)
        m.forward(cfg["maps"])
        assert np.array_equal(m.adjoint(u1), a1)
        assert np.all(m.adjoint(np.zeros(m.osize)) == 0)
        assert rel(m.adjoint(u2), om.adjoint(u2)) < TOL
    finally:
        m.close()
