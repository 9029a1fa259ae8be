"""Synthetic problems used by the tests, expressed with the oracle's types.

Config numbering follows BASELINE.json.  ``config1`` is the reference's own
CPU-runnable case (64x64x128, one synthetic channel); ``band_spec`` gives the 12
real MRS bands with the constants of scripts/main_fusion.py:107-120.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import surfh_oracle as orc  # noqa: E402

STEP = 0.025                 # arcsec  (test/test_fw_ad.py:74-84)
STEP_DEG = STEP / 3600.0

# name: (n_slit, r_min, r_max, det_pix, fov_a ["], fov_b ["], (lambda0, dlambda, n))
BANDS = {
    "1a": (21, 3320, 3710, 0.196, 3.2, 3.7, (4.9004001, 0.0008, 1050)),
    "1b": (21, 3190, 3750, 0.196, 3.2, 3.7, (5.66039985, 0.0008, 1213)),
    "1c": (21, 3100, 3610, 0.196, 3.2, 3.7, (6.53040021, 0.0008, 1400)),
    "2a": (17, 2990, 3110, 0.196, 4.0, 4.8, (7.51065023, 0.0013, 970)),
    "2b": (17, 2750, 3170, 0.196, 4.0, 4.8, (8.67065008, 0.0013, 1124)),
    "2c": (17, 2860, 3300, 0.196, 4.0, 4.8, (10.01065023, 0.0013, 1300)),
    "3a": (16, 2530, 2880, 0.245, 5.2, 6.2, (11.55125019, 0.0025, 769)),
    "3b": (16, 1790, 2640, 0.245, 5.2, 6.2, (13.34125015, 0.0025, 892)),
    "3c": (16, 1980, 2790, 0.245, 5.2, 6.2, (15.41124985, 0.0025, 1028)),
    "4a": (12, 1460, 1930, 0.273, 6.6, 7.7, (17.70300076, 0.006, 542)),
    "4b": (12, 1680, 1760, 0.273, 6.6, 7.7, (20.69300053, 0.006, 632)),
    "4c": (12, 1630, 1330, 0.273, 6.6, 7.7, (24.40299962, 0.006, 717)),
}


def band_wavel(name):
    """The reference's detector axis (global_variables.wavelength_<band>), from the golden geometry file."""
    with np.load(os.path.join(ROOT, "tests", "golden", "bands_geometry.npz")) as z:
        return np.array(z[name + "_wavel"], dtype=np.float64)


def band_spec(name, angle=8.2, wavel_axis=None):
    n_slit, rmin, rmax, dpix, fa, fb, _ = BANDS[name]
    wa = band_wavel(name) if wavel_axis is None else wavel_axis
    return orc.ChannelSpec(fa / 3600, fb / 3600, (0.0, 0.0), angle, dpix, n_slit,
                           float(np.mean([rmin, rmax])), wa, name.upper())


def config1():
    """64x64x128, one synthetic 4-slit channel, 4-point dither (SURVEY.md 8d 'Config 1')."""
    N, Lc = 64, 128
    ax = orc.synthetic_axes(N, STEP_DEG)
    wav = np.linspace(7.50, 7.70, Lc)
    spec = orc.ChannelSpec(0.8 / 3600, 0.9 / 3600, (0.0, 0.0), 8.2, 0.196, 4, 3050.0,
                           np.linspace(7.52, 7.68, 48), "S1")
    tpl = orc.synthetic_templates(Lc)
    sotf = orc.ir2fr(orc.gaussian_psf(wav, STEP), (N, N))
    pts = orc.dither4(spec.det_pix_size, spec.beta_width / spec.n_slit)
    maps = np.random.default_rng(19940407).random((4, N, N))
    return dict(N=N, Lc=Lc, alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, specs=[spec],
                templates=tpl, sotf=sotf, pointings=[pts], maps=maps, step_deg=STEP_DEG)


def two_channel_small():
    """48x48x96 with two overlapping synthetic channels (exercises the lambda-window add, spectroModel.py:176)."""
    N, Lc = 48, 96
    ax = orc.synthetic_axes(N, STEP_DEG)
    wav = np.linspace(7.40, 7.90, Lc)
    s1 = orc.ChannelSpec(0.6 / 3600, 0.7 / 3600, (0.0, 0.0), 8.2, 0.196, 3, 3050.0,
                         np.linspace(7.50, 7.64, 40), "A")
    s2 = orc.ChannelSpec(0.7 / 3600, 0.6 / 3600, (0.0, 0.0), -5.0, 0.196, 2, 2900.0,
                         np.linspace(7.60, 7.80, 36), "B")
    tpl = orc.synthetic_templates(Lc)
    sotf = orc.ir2fr(orc.gaussian_psf(wav, STEP), (N, N))
    p1 = orc.dither4(s1.det_pix_size, s1.beta_width / s1.n_slit)[:2]
    p2 = orc.dither4(s2.det_pix_size, s2.beta_width / s2.n_slit)[:2]
    maps = np.random.default_rng(7).random((4, N, N))
    return dict(N=N, Lc=Lc, alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, specs=[s1, s2],
                templates=tpl, sotf=sotf, pointings=[p1, p2], maps=maps, step_deg=STEP_DEG)


def two_channel_disjoint():
    """Two channels whose wavelength windows do NOT touch: the plan stores two plane segments with a gap."""
    N, Lc = 48, 160
    ax = orc.synthetic_axes(N, STEP_DEG)
    wav = np.linspace(7.20, 8.45, Lc)
    s1 = orc.ChannelSpec(0.6 / 3600, 0.7 / 3600, (0.0, 0.0), 8.2, 0.196, 3, 3050.0, np.linspace(7.32, 7.42, 30), "A")
    s2 = orc.ChannelSpec(0.7 / 3600, 0.6 / 3600, (0.0, 0.0), -5.0, 0.196, 2, 2900.0, np.linspace(8.15, 8.30, 34), "B")
    tpl = orc.synthetic_templates(Lc)
    sotf = orc.ir2fr(orc.gaussian_psf(wav, STEP), (N, N))
    p1 = orc.dither4(s1.det_pix_size, s1.beta_width / s1.n_slit)[:2]
    p2 = orc.dither4(s2.det_pix_size, s2.beta_width / s2.n_slit)[:3]
    maps = np.random.default_rng(8).random((4, N, N))
    return dict(N=N, Lc=Lc, alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, specs=[s1, s2],
                templates=tpl, sotf=sotf, pointings=[p1, p2], maps=maps, step_deg=STEP_DEG)


def two_channel_mid():
    """128x128x512: large enough for every fast path of the HIP operator to engage -- the fused adjoint tail (127 <= N <= 255),
    four 128-wavelength chunks (OTF-support lists with chunk-dependent cutoffs), detector axes of 400 samples at the real
    bands' sampling, long enough for the spectral response to have far tails inside a band in both directions (K-step classes
    of the spectral-blur GEMMs), two channels (grouped adjoint
    GEMMs), a field of view well inside the image (column range of the transforms) -- and small enough for the float64 oracle
    to finish in seconds.  The PSF support is 81x81 so that the long-wavelength OTFs have a clean cutoff."""
    N, Lc = 128, 512
    ax = orc.synthetic_axes(N, STEP_DEG)
    wav = np.linspace(7.40, 8.06, Lc)
    s1 = orc.ChannelSpec(1.0 / 3600, 1.2 / 3600, (0.0, 0.0), 8.2, 0.196, 5, 3050.0, np.linspace(7.45, 7.95, 400), "A")
    s2 = orc.ChannelSpec(1.1 / 3600, 1.0 / 3600, (0.0, 0.0), -5.0, 0.196, 4, 2900.0, np.linspace(7.52, 8.02, 390), "B")
    tpl = orc.synthetic_templates(Lc)
    yy, xx = np.mgrid[0:81, 0:81]
    sig = np.linspace(2.0, 5.0, Lc)
    psf = np.exp(-((yy - 40) ** 2 + (xx - 40) ** 2)[None] / (2.0 * sig[:, None, None] ** 2))
    psf /= psf.sum(axis=(1, 2), keepdims=True)
    sotf = orc.ir2fr(psf, (N, N))
    p1 = orc.dither4(s1.det_pix_size, s1.beta_width / s1.n_slit)
    p2 = orc.dither4(s2.det_pix_size, s2.beta_width / s2.n_slit)[:2]
    maps = np.random.default_rng(9).random((4, N, N))
    return dict(N=N, Lc=Lc, alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, specs=[s1, s2],
                templates=tpl, sotf=sotf, pointings=[p1, p2], maps=maps, step_deg=STEP_DEG)


def oracle_model(cfg, box="fft", gridding="bilinear"):
    return orc.OracleModel(cfg["sotf"], cfg["templates"], cfg["alpha_axis"], cfg["beta_axis"],
                           cfg["wavel"], cfg["specs"], cfg["step_deg"], cfg["pointings"], box=box, gridding=gridding)
