#!/usr/bin/env python
"""2-D deconvolution of MRS slit data with the no-rotation operator ``MRSBlurred`` on MI355X -- the run of the reference's
``scripts/simulate_deconvolution_mrs_rectangle.py:100-188`` (and ``scripts/deconvolution_mrs_noRotation.py:100-212``):

    mixed_maps = (0.4 m0 + 0.5 m1 + 0.4 m2 + 0.3 m3) * 1e4          one image from four abundance maps      (:135)
    model      = MRSBlurred(sotf, alpha_axis, beta_axis, ch1c, step, 4 pointings)                           (:137-167)
    data       = model.forward(fliplr(mixed_maps))                                                          (:169)
    criterion  = QuadCriterion_MRS_2D(mu_spectro=1, data, model, mu_reg=5, gradient="separated")            (:190-196)
    result     = criterion.run_method("lcg", 600, perf_crit=1, calc_crit=True, value_init=0)                (:198)

The reference reads its maps, PSF and pointings from the author's disk; here they are synthetic (random maps, the Gaussian
PSF of surfh/ToolsDir/utils.py:40-50 at the chosen wavelength, the 1C field of view without rotation, four integer-pixel
pointings) or come from ``--input`` (an .npz with ``maps [4,N,N]``, ``psf [n,n]`` and optionally ``pointings [P,2]`` in
degrees).  ``--planes L`` solves L wavelength planes at once (BASELINE.json configs[4]).  Results: ``res_x.npy``,
``criterion.npy``, ``data.npy`` under ``--out``.
"""
from __future__ import annotations

import os
import sys
import time

import click
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEP = 0.025                       # arcsec per pixel (simulate_deconvolution_mrs_rectangle.py:66)


def build_problem(npix: int, planes: int, seed: int, inp: str | None):
    from surfh_amd import instru, synth
    step_deg = STEP / 3600
    wl_1c = synth.band_wavelengths("1c")
    rng = np.random.default_rng(seed)
    if inp:
        z = np.load(inp)
        maps = np.asarray(z["maps"], dtype=np.float64)
        npix = maps.shape[-1]
        psf = np.asarray(z["psf"], dtype=np.float64)
        psfs = np.broadcast_to(psf / psf.sum(), (planes,) + psf.shape)
        pts = [tuple(p) for p in z["pointings"]] if "pointings" in z.files else None
    else:
        maps = rng.random((4, npix, npix))
        lam = wl_1c[100] if planes == 1 else np.linspace(wl_1c[100], wl_1c[100] + 0.05, planes)   # instr_wavelength[100] (:73)
        psfs = synth.gaussian_psf(np.atleast_1d(lam), STEP)
        pts = None
    if pts is None:
        s = step_deg
        pts = [(0.0, 0.0), (2 * s, -3 * s), (-4 * s, 1 * s), (3 * s, 5 * s)]
    mixed = (0.4 * maps[0] + 0.5 * maps[1] + 0.4 * maps[2] + 0.3 * maps[3]) * 10000          # :135
    ax = synth.axes(npix, step_deg)
    ch1c = instru.IFU(fov=instru.FOV(3.2 / 3600, 3.7 / 3600, origin=instru.Coord(0, 0), angle=0.0), det_pix_size=0.196,
                      n_slit=21, w_blur=instru.SpectralBlur(float(np.mean([3100, 3610]))), pce=None, wavel_axis=wl_1c, name="1C")
    sotf = synth.ir2fr(psfs, (npix, npix))
    if planes == 1:
        sotf = sotf[0]
    truth = np.fliplr(mixed)                                                                  # :169
    if planes > 1:
        truth = np.stack([truth * (1.0 + 0.1 * k / planes) for k in range(planes)])
    return dict(sotf=sotf, alpha_axis=ax, beta_axis=ax.copy(), ifu=ch1c, step_deg=step_deg,
                pointings=instru.CoordList([instru.Coord(a, b) for a, b in pts]), truth=truth)


@click.command()
@click.option("-np", "--npix", default=251, type=int, help="image size (the reference's maps are 251 x 251)")
@click.option("-hp", "--hyper_parameter", default=5.0, type=float, help="mu_reg (reference: 5)")
@click.option("-ni", "--niter", default=600, type=int, help="iterations (reference: 600)")
@click.option("-m", "--method", default="lcg", type=str, help="'lcg' or anything else for 3MG (criterion_2D.py:190-193)")
@click.option("-vi", "--value_init", default=0.0, type=float)
@click.option("--planes", default=1, type=int, help="wavelength planes solved at once (independent 2-D problems)")
@click.option("--input", "inp", default=None, type=str, help=".npz with maps, psf[, pointings]")
@click.option("--out", default="deconvolution_results", type=str)
@click.option("--quiet", is_flag=True, help="no per-iteration prints (the reference prints every iteration)")
@click.option("--seed", default=19940407, type=int)
@click.option("--device", default=0, type=int)
def main(npix, hyper_parameter, niter, method, value_init, planes, inp, out, quiet, seed, device):
    from surfh_amd.spectro_blind_rectangle import MRSBlurred, QuadCriterion_MRS_2D
    prob = build_problem(npix, planes, seed, inp)
    model = MRSBlurred(prob["sotf"], prob["alpha_axis"], prob["beta_axis"], prob["ifu"], prob["step_deg"], prob["pointings"],
                       device=device)
    simulated_data = model.forward(prob["truth"])
    crit = QuadCriterion_MRS_2D(mu_spectro=1, y_spectro=np.copy(simulated_data), model_spectro=model, mu_reg=hyper_parameter,
                                printing=True, gradient="separated")
    t0 = time.time()
    if quiet:
        res = crit.run_method(method, niter, value_init=value_init)
        crit.L_crit_val = [crit.get_crit_val(np.full(model.ishape, value_init)), crit.get_crit_val(res.x)]
    else:
        res = crit.run_method(method, niter, perf_crit=1, calc_crit=True, value_init=value_init)
    dt = time.time() - t0
    x = np.asarray(res.x).reshape(model.ishape)
    err = float(np.linalg.norm(x - prob["truth"]) / np.linalg.norm(prob["truth"]))
    print(f"{method}: {res.nit} iterations in {dt:.2f} s ({res.nit / dt:.1f} it/s incl. callbacks), relative error to the truth {err:.3e}, "
          f"criterion {crit.L_crit_val[0]:.6e} -> {crit.L_crit_val[-1]:.6e}")
    os.makedirs(out, exist_ok=True)
    np.save(os.path.join(out, "res_x.npy"), x)
    np.save(os.path.join(out, "criterion.npy"), np.asarray(crit.L_crit_val))
    np.save(os.path.join(out, "data.npy"), simulated_data)
    model.close()


if __name__ == "__main__":
    main()
