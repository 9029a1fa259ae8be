"""One-off robustness sweep (GPU): random instrument geometries -- cube size, rotation, field of view, slit count, detector pixel
(srf), detector axis, resolving power, 1-4 pointings at random offsets, one or two channels -- forward, exact adjoint and the
reference-compatible adjoint of the HIP operator against the float64 oracle.  Geometries whose field of view leaves the cube
raise ValueError on both sides (cython_2D_interpolation.py:472-478) and are skipped.
    python3 tools/sweep_geometry.py [n_cases] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers                                  # noqa: E402
import problems                                 # noqa: E402
from oracle import surfh_oracle as orc          # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
STEP, STEP_DEG = problems.STEP, problems.STEP_DEG
bad = done = skipped = 0
t0 = time.time()
for seed in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([48, 64, 80, 96, 127, 130, 160]))
    Lc = int(rng.choice([64, 96, 128, 200, 256, 384]))
    ax = orc.synthetic_axes(N, STEP_DEG)
    lam0 = float(rng.uniform(5.0, 20.0))
    wav = np.linspace(lam0, lam0 * (1.0 + rng.uniform(0.02, 0.08)), Lc)
    specs, pts = [], []
    for c in range(int(rng.integers(1, 3))):
        fov_a = rng.uniform(0.35, 0.6) * N * STEP
        fov_b = rng.uniform(0.35, 0.6) * N * STEP
        n_slit = int(rng.integers(2, 8))
        dpix = float(rng.choice([0.13, 0.196, 0.245, 0.273]))
        Ldet = int(rng.integers(24, 220))
        lo = rng.uniform(0.05, 0.4)
        wdet = np.linspace(wav[0] + lo * (wav[-1] - wav[0]), wav[0] + (lo + rng.uniform(0.3, 0.55)) * (wav[-1] - wav[0]), Ldet)
        spec = orc.ChannelSpec(fov_a / 3600, fov_b / 3600, (0.0, 0.0), float(rng.uniform(-25, 25)), dpix, n_slit,
                               float(rng.uniform(1500, 4000)), wdet, f"S{c}")
        P = int(rng.integers(1, 5))
        d4 = orc.dither4(spec.det_pix_size, spec.beta_width / spec.n_slit)
        pts.append([(a * rng.uniform(0.3, 2.0), b * rng.uniform(0.3, 2.0)) for a, b in d4[:P]])
        specs.append(spec)
    tpl = orc.synthetic_templates(Lc)
    sotf = orc.ir2fr(orc.gaussian_psf(wav, STEP), (N, N))
    maps = rng.random((4, N, N))
    cfg = dict(N=N, Lc=Lc, alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, specs=specs, templates=tpl, sotf=sotf, pointings=pts,
               maps=maps, step_deg=STEP_DEG)
    tag = f"seed {seed}: N {N} Lc {Lc} " + " ".join(f"[{s.n_slit} slits, pix {s.det_pix_size}, angle {s.angle:.1f}, Ldet {len(s.wavel_axis)}, P {len(p)}]"
                                                     for s, p in zip(specs, pts))
    try:
        om = problems.oracle_model(cfg, box="direct")
    except (ValueError, AssertionError) as e:
        skipped += 1
        print(f"{tag}: skipped by the oracle ({type(e).__name__}: {str(e)[:60]})", flush=True)
        continue
    try:
        m = helpers.build_model(cfg, with_ref=True)
    except Exception as e:                       # noqa: BLE001
        bad += 1
        print(f"{tag}: FAIL plan creation {e!r}", flush=True)
        continue
    try:
        u = rng.random(om.osize)
        ef = helpers.rel(m.forward(maps), om.forward(maps))
        ea = helpers.rel(m.adjoint(u), om.adjoint(u))
        er = helpers.rel(m.adjoint_ref(u), om.adjoint_ref(u))
        ok = max(ef, ea, er) < 1e-5
        bad += 0 if ok else 1
        done += 1
        print(f"{tag}: forward {ef:.2e} adjoint {ea:.2e} adjoint_ref {er:.2e} {'ok' if ok else 'FAIL'}  [{time.time() - t0:.0f}s]", flush=True)
    except Exception as e:                       # noqa: BLE001
        bad += 1
        print(f"{tag}: FAIL {e!r}", flush=True)
    finally:
        m.close()
print(f"{done} compared, {skipped} skipped, {bad} failure(s)")
sys.exit(1 if bad else 0)
