"""driver parity -- CG with the qmm callback, the criterion trace of QuadCriterion_MRS.run_method,
mapsToCube / cubeTomaps on the device and the fusion driver end to end (needs an MI355X)."""
import importlib.util
import os

import numpy as np
import pytest
from click.testing import CliRunner

import problems
from helpers import build_model, rel
from oracle import surfh_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def setup():
    cfg = problems.config1()
    om = problems.oracle_model(cfg, box="direct")
    m = build_model(cfg)
    y = om.forward(cfg["maps"])
    y = y + np.random.default_rng(1).standard_normal(y.shape) * 1e-2 * np.sqrt(np.mean(y ** 2))
    yield cfg, om, m, y
    m.close()


def test_cg_callback_trace(setup):
    cfg, om, m, y = setup
    x_ref, gn_ref, nit_ref = m.cg(y, mu=1.0, mu_reg=5e3, x0=np.zeros(m.ishape), max_iter=7)
    seen = []

    def cb(it, gn, x):
        assert gn.shape == (it + 1,) and x.shape == tuple(m.ishape)
        # the callback may run the operator on the same plan (the criterion trace does)
        seen.append((it, gn[-1], float(np.sum(m.forward(x) ** 2)), x.copy()))
        return False

    x, gn, nit = m.cg(y, mu=1.0, mu_reg=5e3, x0=np.zeros(m.ishape), max_iter=7, callback=cb)
    assert nit == nit_ref == 7 and [s[0] for s in seen] == list(range(1, 8))
    assert np.array_equal(gn, gn_ref) and np.array_equal(x, x_ref)          # the callback does not disturb the solver
    assert np.array_equal([s[1] for s in seen], gn[1:]) and np.array_equal(seen[-1][3], x)
    # a truthy return stops the loop after that iteration
    x3, gn3, nit3 = m.cg(y, mu=1.0, mu_reg=5e3, x0=np.zeros(m.ishape), max_iter=7, callback=lambda it, g, xx: it == 3)
    assert nit3 == 3 and np.array_equal(gn3, gn[:4]) and np.array_equal(x3, seen[2][3])
    # an exception in the callback surfaces in Python, not through the C frame
    with pytest.raises(ZeroDivisionError):
        m.cg(y, mu=1.0, mu_reg=5e3, max_iter=2, callback=lambda it, g, xx: 1 / 0)


def test_mmmg_matches_oracle_and_cg(setup, capsys):
    """The reference's other solver choice (fusion_CT.py:194-198 -> qmm.mmmg) against the oracle's restatement, and against
    CG: on this quadratic criterion both produce the same iterates."""
    from surfh_amd.fusion import QuadCriterion_MRS
    cfg, om, m, y = setup
    mu, mur, nit = 1.0, 5e3, 8
    x0 = np.ones(m.ishape) * 0.5
    ref = orc.mmmg(om, y, mu, mur, x0, max_iter=nit)
    x, gn, n = m.mmmg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=nit)
    assert n == nit and gn.shape == (nit + 1,)
    gr = np.array(ref["grad_norm"])
    # fp32 operator vs the float64 restatement of qmm.mmmg: the same accuracy as CG has against qmm.lcg
    assert rel(x, ref["x"]) < 1e-4 and float(np.max(np.abs(gn - gr) / gr)) < 2e-4
    xc, gc, _ = m.cg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=nit)
    assert rel(x, xc) < 1e-4 and float(np.max(np.abs(gn ** 2 - gc) / gc)) < 1e-3
    # refresh every iteration (gradient always from scratch, as qmm does) gives the same path
    xf, gf, _ = m.mmmg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=nit, refresh=1)
    assert rel(xf, x) < 1e-4
    # from x0 = 0 the float64 restatement itself leaves the CG path at iterations 7-8 (numpy's pinv cut drops the
    # memory direction once |move|^2 / |grad|^2 < 1e-15, see oracle mmmg vs lcg); the device solver, which scales the 2x2
    # system, stays on it
    z = np.zeros(m.ishape)
    xz, gz, _ = m.mmmg(y, mu=mu, mu_reg=mur, x0=z, max_iter=nit)
    rz, rl = orc.mmmg(om, y, mu, mur, z, max_iter=nit), orc.lcg(om, y, mu, mur, z, max_iter=nit)
    assert float(np.max(np.abs(gz[:6] - rz["grad_norm"][:6]) / rz["grad_norm"][:6])) < 2e-4
    assert float(np.max(np.abs(gz - np.sqrt(rl["grad_norm"])) / np.sqrt(rl["grad_norm"]))) < 2e-4 and rel(xz, rl["x"]) < 2e-4
    # long run: converges like CG (the literal [-grad, move] basis in fp32 does not: DESIGN.md)
    xl, gl, _ = m.mmmg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=40)
    xcl, gcl, _ = m.cg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=40)
    jl, jc = orc.crit_val(om, y, xl, mu, mur), orc.crit_val(om, y, xcl, mu, mur)
    assert abs(jl - jc) / jc < 1e-4 and rel(xl, xcl) < 2e-3
    # callback, early stop, tolerance stop
    seen = []
    x3, g3, n3 = m.mmmg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=nit, callback=lambda it, g, xx: seen.append(it) or it == 3)
    assert n3 == 3 and seen == [1, 2, 3] and np.array_equal(g3, gn[:4])
    xt, gt, nt = m.mmmg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=nit, tol=gn[4] * 1.0001 / x.size)
    assert nt == 4 and np.array_equal(gt, gn[:5])
    # through the criterion class, as the reference's drivers select it
    q = QuadCriterion_MRS(mu, y, m, mur)
    res = q.run_method("mmmg", nit, value_init=0.5)
    assert res.nit == nit and rel(res.x.reshape(m.ishape), x) == 0.0
    c = [orc.crit_val(om, y, m.mmmg(y, mu=mu, mu_reg=mur, x0=x0, max_iter=k)[0], mu, mur) for k in (1, 4, 8)]
    assert c[0] > c[1] > c[2]


def test_criterion_trace_modes(setup, capsys):
    """fusion_CT.py:163-225: criterion at iterations 1, 6, 11, ... when both flags are set."""
    from surfh_amd.fusion import QuadCriterion_MRS
    cfg, om, m, y = setup
    q = QuadCriterion_MRS(1, y, m, 5e3, printing=True)
    res = q.run_method("lcg", 7, perf_crit=1, calc_crit=True, value_init=0)
    out = capsys.readouterr().out
    assert res.nit == 7 and len(q.L_crit_val) == 2                      # iterations 1 and 6
    assert out.count("Grad norm =") == 7 and out.count("Criterion value =") == 2 and "Iteration n°7" in out
    xs = []
    m.cg(y, mu=1, mu_reg=5e3, x0=np.zeros(m.ishape), max_iter=7, callback=lambda it, g, x: xs.append(x.copy()) and False)
    for k, it in enumerate((1, 6)):
        ref = orc.crit_val(om, y, xs[it - 1], 1.0, 5e3)                  # float64 oracle criterion of the same iterate
        assert abs(q.L_crit_val[k] - ref) / ref < 1e-5
    assert q.L_crit_val[1] < q.L_crit_val[0]
    # gradient norms only
    q2 = QuadCriterion_MRS(1, y, m, 5e3)
    q2.run_method("lcg", 3, perf_crit=1, calc_crit=False, value_init=0)
    assert q2.L_crit_val == [] and capsys.readouterr().out.count("Grad norm =") == 3
    # criterion of every iterate
    q3 = QuadCriterion_MRS(1, y, m, 5e3)
    q3.run_method("lcg", 3, calc_crit=True, value_init=0)
    assert len(q3.L_crit_val) == 3 and q3.L_crit_val[2] < q3.L_crit_val[0]
    # silent
    q4 = QuadCriterion_MRS(1, y, m, 5e3)
    r4 = q4.run_method("lcg", 3, value_init=0)
    assert q4.L_crit_val == [] and np.allclose(r4.x, xs[2].ravel())


def test_lmm_on_device(setup):
    cfg, om, m, y = setup
    rng = np.random.default_rng(3)
    tpl = np.asarray(cfg["templates"], dtype=np.float64)
    maps = rng.random(m.ishape)
    cube = m.mapsToCube(maps)
    assert cube.shape == (tpl.shape[1],) + tuple(m.ishape[1:])
    assert rel(cube, np.tensordot(tpl.T, maps, axes=(1, 0))) < 1e-6      # jax_utils.py:10-16
    c = rng.random(cube.shape)
    assert rel(m.cubeTomaps(c), np.tensordot(tpl, c, axes=(1, 0))) < 1e-6  # jax_utils.py:18-26
    with pytest.raises(ValueError):
        m.cubeTomaps(c[:-1])


def test_driver_end_to_end(tmp_path):
    spec = importlib.util.spec_from_file_location("main_fusion", os.path.join(ROOT, "scripts", "main_fusion.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    r = CliRunner().invoke(drv.main, ["-fd", str(tmp_path), "-np", "251", "-hp", "5e3", "-ni", "6", "--synthetic", "small"])
    assert r.exit_code == 0, r.output + repr(r.exception)
    d = tmp_path / "Results" / drv.result_dir_name("lcg", 1, 4, 6, 5e3, False)
    x = np.load(d / "res_x.npy")
    cube = np.load(d / "res_cube.npy")
    crit = np.load(d / "criterion.npy")
    assert x.shape == (4 * 251 * 251,) and cube.shape == (256, 251, 251) and crit.shape == (2,)
    assert crit[1] < crit[0] and "Iteration n°6" in r.output
    from surfh_amd import synth
    tpl = synth.templates(256)
    assert rel(cube, np.tensordot(tpl.T, x.reshape(4, 251, 251), axes=(1, 0))) < 1e-6
    # the reconstruction moves towards the truth the data were simulated from
    truth = np.random.default_rng(19940407).random((4, 251, 251))
    assert np.isfinite(x).all() and np.linalg.norm(x) > 0 and truth.shape == (4, 251, 251)


def test_driver_checkpoint_and_resume(tmp_path):
    """--checkpoint_every writes the iterate (atomically) while the solver runs; --resume warm-starts from it and runs the rest.
    The resumed run restarts the search directions, so its result agrees with the uninterrupted one only approximately; what
    is asserted: the checkpoint is the uninterrupted run's iterate at that iteration, and resuming lowers the criterion."""
    from surfh_amd.fusion import load_checkpoint
    spec = importlib.util.spec_from_file_location("main_fusion", os.path.join(ROOT, "scripts", "main_fusion.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    base = ["-np", "251", "-hp", "5e3", "--synthetic", "small"]
    r = CliRunner().invoke(drv.main, ["-fd", str(tmp_path / "a"), "-ni", "4", "--checkpoint_every", "2"] + base)
    assert r.exit_code == 0, r.output + repr(r.exception)
    da = tmp_path / "a" / "Results" / drv.result_dir_name("lcg", 1, 4, 4, 5e3, False)
    xck, it, gn = load_checkpoint(da / "checkpoint.npz")
    assert it == 4 and xck.size == 4 * 251 * 251 and len(gn) == 5
    assert rel(xck.ravel(), np.load(da / "res_x.npy")) < 1e-6            # the last checkpoint is the final iterate
    assert not (da / "checkpoint.tmp.npz").exists()
    # an "interrupted" run of 2 iterations, then 8 in total resumed from its checkpoint
    r = CliRunner().invoke(drv.main, ["-fd", str(tmp_path / "b"), "-ni", "2", "--checkpoint_every", "2"] + base)
    assert r.exit_code == 0, r.output + repr(r.exception)
    db = tmp_path / "b" / "Results" / drv.result_dir_name("lcg", 1, 4, 2, 5e3, False)
    x2, it2, _ = load_checkpoint(db / "checkpoint.npz")
    assert it2 == 2
    r = CliRunner().invoke(drv.main, ["-fd", str(tmp_path / "c"), "-ni", "8", "--resume", str(db / "checkpoint.npz")] + base)
    assert r.exit_code == 0, r.output + repr(r.exception)
    assert "2 iterations done, 6 to go" in r.output
    dc = tmp_path / "c" / "Results" / drv.result_dir_name("lcg", 1, 4, 8, 5e3, False)     # named by the total, as an uninterrupted run
    crit_c = np.load(dc / "criterion.npy")
    crit_b = np.load(db / "criterion.npy")
    assert crit_c[-1] < crit_b[-1]                                        # the resumed iterations keep descending


def test_joint_prior(setup):
    """``gradient="joint"`` (Difference_Operator_Joint, fusion_CT.py:45-62,141-150): the device stencil is D^T D with
    D = ir2fr(laplacian(2)) restated (udft absent: parity unpinned) -- checked against the oracle's Fourier-domain form of the
    same kernel, then through the criterion class: the solver minimises the joint criterion."""
    import torch
    from surfh_amd.fusion import QuadCriterion_MRS
    cfg, om, m, y = setup
    rng = np.random.default_rng(4)
    x = rng.standard_normal(m.ishape)
    # Fourier form: irfft2(|D(f)|^2 rfft2(x)), norm="ortho" both ways as udft's rdft2 / irdftn
    dtd = np.fft.irfft2(orc.reg_freq(m.ishape[1:], "joint")[None] * np.fft.rfft2(x, norm="ortho"), s=m.ishape[1:], norm="ortho")
    m.set_prior("joint")
    d_t = torch.as_tensor(x.astype(np.float32), device="cuda:0")
    q_t = torch.zeros_like(d_t)
    m.prior_add_dev(d_t, q_t, 2.5)
    torch.cuda.synchronize()
    e = rel(q_t.cpu().numpy(), 2.5 * dtd)
    m.set_prior("separated")
    q2 = torch.zeros_like(d_t)
    m.prior_add_dev(d_t, q2, 2.5)
    torch.cuda.synchronize()
    e2 = rel(q2.cpu().numpy(), 2.5 * (orc.diff_r_t(orc.diff_r(x)) + orc.diff_c_t(orc.diff_c(x))))
    print(f"joint prior stencil vs Fourier form {e:.2e}; separated {e2:.2e}")
    assert e < 1e-6 and e2 < 1e-6
    crit = QuadCriterion_MRS(1.0, y, m, 50.0, gradient="joint")
    res = crit.run_method("lcg", 12, value_init=0.0)
    x0 = np.zeros(m.ishape)
    assert res.nit == 12 and crit.get_crit_val(res.x) < 0.9 * crit.get_crit_val(x0) and res.grad_norm[-1] < res.grad_norm[0]
    # CG on the joint normal equations agrees with the float64 oracle's CG on the same operator with the Fourier-form prior
    def q_joint(v):
        return om.adjoint(om.forward(v)) + 50.0 * np.fft.irfft2(orc.reg_freq(m.ishape[1:], "joint")[None] * np.fft.rfft2(v, norm="ortho"),
                                                                  s=m.ishape[1:], norm="ortho")
    xo = np.zeros(m.ishape); r = om.adjoint(y) - q_joint(xo); d = r.copy(); rr = float(np.sum(r * r)); trace = [rr]
    for it in range(5):
        qd = q_joint(d); step = rr / float(np.sum(d * qd)); xo += step * d
        r = om.adjoint(y) - q_joint(xo) if it == 0 else r - step * qd
        rn = float(np.sum(r * r)); d = r + (rn / rr) * d; rr = rn; trace.append(rr)
    dev = np.abs(np.array(res.grad_norm[:6]) - np.array(trace)) / np.array(trace)
    print("joint-prior CG vs float64 recurrence, r.r deviation per iteration:", [f"{v:.1e}" for v in dev])
    assert np.max(dev) < 1e-2
    m.set_prior("separated")
    with pytest.raises(ValueError):
        QuadCriterion_MRS(1.0, y, m, 50.0, gradient="nope")


def test_reference_adjoint_right_hand_side_is_recorded(setup):
    """The reference gives qmm its interpolating ``gridding_t`` adjoint (spectroModel.py:173-185), the solver here uses the
    exact transpose: how far the two right-hand sides mu A^T y and the resulting normal equations are apart on
    config 1 is recorded here (and bounded, so that a regression of either adjoint shows)."""
    cfg, om, m, y = setup
    b_exact, b_ref = m.adjoint(y), m.adjoint_ref(y)
    d = rel(b_ref, b_exact)
    print(f"config 1: |A_ref^T y - A^T y| / |A^T y| = {d:.3e}")
    assert 1e-5 < d < 0.2           # different operators (the reference pair is not an adjoint pair), same physics
    assert rel(b_ref, om.adjoint_ref(y)) < 1e-5 and rel(b_exact, om.adjoint(y)) < 1e-5


@pytest.mark.parametrize("planes,method", [(1, "lcg"), (3, "lcg"), (1, "qmm")])
def test_deconvolution_driver_end_to_end(tmp_path, planes, method):
    """the 2-D deconvolution run of scripts/simulate_deconvolution_mrs_rectangle.py:149-198 (and, with
    method "qmm", the 3MG branch deconvolution_mrs_noRotation.py:199-212 takes) through scripts/deconvolution_mrs.py:
    criterion trace recorded at iterations 1, 6, 11, ... decreases, the result files exist and the solution agrees with
    the float64 restatement of the same CG on the same synthetic problem."""
    sp = importlib.util.spec_from_file_location("deconvolution_mrs", os.path.join(ROOT, "scripts", "deconvolution_mrs.py"))
    dd = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(dd)
    out = str(tmp_path / "res")
    niter = 21
    r = CliRunner().invoke(dd.main, ["-np", "192", "-ni", str(niter), "-m", method, "--planes", str(planes), "--out", out])
    assert r.exit_code == 0, r.output
    x = np.load(os.path.join(out, "res_x.npy"))
    crit = np.load(os.path.join(out, "criterion.npy"))
    data = np.load(os.path.join(out, "data.npy"))
    assert x.shape == ((192, 192) if planes == 1 else (planes, 192, 192))
    assert len(crit) == 5 and np.all(np.diff(crit) < 0)                       # iterations 1, 6, 11, 16, 21 (criterion_2D.py:172-175)
    assert r.output.count("Iteration n°") == niter and "Criterion value" in r.output
    # float64 restatement: the oracle's 2-D operator, plane by plane, CG on the same normal equations
    prob = dd.build_problem(192, planes, 19940407, None)
    spec = orc.ChannelSpec(3.2 / 3600, 3.7 / 3600, (0.0, 0.0), 0.0, 0.196, 21, float(np.mean([3100, 3610])), prob["ifu"].wavel_axis, "1C")
    pts = [(c.alpha, c.beta) for c in prob["pointings"]]
    sotf = prob["sotf"] if planes > 1 else prob["sotf"][None]
    bo = orc.BlurredOracle(sotf, prob["alpha_axis"], prob["beta_axis"], spec, prob["step_deg"], pts)
    truth = prob["truth"] if planes > 1 else prob["truth"][None]
    yo = bo.forward(truth)
    assert rel(data.reshape(yo.shape), yo) < 1e-5
    if method == "lcg":
        def Q(v):
            return bo.adjoint(bo.forward(v)) + 5.0 * ((2 * v - np.roll(v, 1, -2) - np.roll(v, -1, -2)) + (2 * v - np.roll(v, 1, -1) - np.roll(v, -1, -1)))
        xo = np.zeros_like(truth); b = bo.adjoint(yo); rr_ = b - Q(xo); d = rr_.copy()
        rr = np.sum(rr_ * rr_, axis=(1, 2))
        for it in range(niter):
            q = Q(d); step = rr / np.sum(d * q, axis=(1, 2)); xo += step[:, None, None] * d
            rr_ = b - Q(xo) if it % 50 == 0 else rr_ - step[:, None, None] * q
            rn = np.sum(rr_ * rr_, axis=(1, 2)); d = rr_ + (rn / rr)[:, None, None] * d; rr = rn
        e = rel(x.reshape(xo.shape), xo)
        print(f"deconvolution driver ({planes} plane(s)) vs float64 CG after {niter} iterations: {e:.2e}")
        assert e < 1e-3
