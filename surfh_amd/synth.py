"""Synthetic problems of the named benchmark shapes: axes, linear-ramp
templates, Gaussian PSF -> OTF, the 12 MRS bands with the constants the reference's driver
uses (scripts/main_fusion.py:107-120) and the 4-point sub-pixel dither
(test/test_fw_ad.py:736-741).  Host-side NumPy, setup only."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from . import instru

STEP = 0.025                    # arcsec per cube pixel (test/test_fw_ad.py:74-84)
STEP_DEG = STEP / 3600.0

# band: (n_slit, R_min, R_max, det_pix ["], fov_alpha ["], fov_beta ["], (lambda0, dlambda, n) of the detector axis)
BANDS: Dict[str, tuple] = {
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


_WAVEL_TABLE = None


def band_wavelengths(name: str) -> np.ndarray:
    """Detector wavelength axis of a sub-band: the reference's table ``global_variables.wavelength_<band>``
    (surfh/Others/global_variables.py, read by wavelength_mrs.py:22-46), shipped as data in
    ``surfh_amd/data/mrs_wavelengths.npz`` (written by tests/golden/make_golden.py from the reference; bit-equal,
    tests/test_host_geometry.py).  The tables are CRVAL + CDELT * arange(NAXIS) evaluated in float32 steps, which a
    float64 re-evaluation of that formula reproduces only to 3e-8 um (and band 3B's second sample is off the grid by
    2.5e-4 um in the reference), so the table itself is the single source."""
    global _WAVEL_TABLE
    if _WAVEL_TABLE is None:
        import os
        with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mrs_wavelengths.npz")) as z:
            _WAVEL_TABLE = {k: np.array(z[k], dtype=np.float64) for k in z.files}
    wa = _WAVEL_TABLE[name]
    assert len(wa) == BANDS[name][6][2]
    return wa.copy()


def band_ifu(name: str, angle: float = 8.2, wavel_axis=None) -> instru.IFU:
    n_slit, rmin, rmax, dpix, fa, fb, _ = BANDS[name]
    wa = band_wavelengths(name) if wavel_axis is None else wavel_axis
    return instru.IFU(fov=instru.FOV(fa / 3600, fb / 3600, origin=instru.Coord(0, 0), angle=angle),
                      det_pix_size=dpix, n_slit=n_slit, w_blur=instru.SpectralBlur(float(np.mean([rmin, rmax]))),
                      pce=None, wavel_axis=wa, name=name.upper())


def ir2fr(imp_resp, shape):
    """PSF -> OTF as ``udft.ir2fr`` does (udft 3.4.0; call site scripts/main_fusion.py:98):
    zero-pad, roll the centre floor(n/2) to the origin, un-normalised rfftn."""
    imp_resp = np.asarray(imp_resp)
    nd = len(shape)
    pad = np.zeros(imp_resp.shape[:-nd] + tuple(shape), dtype=imp_resp.dtype)
    pad[(Ellipsis,) + tuple(slice(0, s) for s in imp_resp.shape[-nd:])] = imp_resp
    for k, n in enumerate(imp_resp.shape[-nd:]):
        pad = np.roll(pad, -int(np.floor(n / 2)), axis=imp_resp.ndim - nd + k)
    return np.fft.rfftn(pad, axes=tuple(range(-nd, 0)))


def gaussian_psf(wavel_axis, step_arcsec: float, D: float = 6.5) -> np.ndarray:
    """The reference's synthetic PSF (surfh/ToolsDir/utils.py:40-50): FWHM = lambda/D, 40x40 support."""
    x = np.linspace(-30, 30, 40).reshape((1, -1))
    y = x.reshape((-1, 1))
    w = np.asarray(wavel_axis, dtype=np.float64).reshape((-1, 1, 1))
    sigma = ((w * 1e-6 / D) * 206265) / (step_arcsec * 2.354)
    psf = np.exp(-(x[None] ** 2 + y[None] ** 2) / (2 * sigma ** 2))
    return psf / np.sum(psf, axis=(1, 2), keepdims=True)


def axes(n: int, step_deg: float = STEP_DEG) -> np.ndarray:
    a = np.arange(n).astype(np.float64) * step_deg
    return a - np.mean(a)


def templates(n_lambda: int) -> np.ndarray:
    """Four linear-ramp templates (test/global_variable_testing.py:227-230)."""
    lam = np.arange(n_lambda, dtype=np.float64)
    c = (11.0, 15.0, 16.0, 17.0)
    return np.stack([(0.2 + 0.1 * t) * lam + c[t] for t in range(4)])


def dither4(ifu: instru.IFU) -> instru.CoordList:
    da = (ifu.det_pix_size / 3600) / 4
    db = ifu.slit_beta_width / 4
    return instru.CoordList([instru.Coord(da, db), instru.Coord(-da, db), instru.Coord(da, -db), instru.Coord(-da, -db)])


def problem(bands: Sequence[str], n_lambda: int, lam_range, n_pix: int = 251, lam_stride: int = 1, geometry_only: bool = False) -> dict:
    """A fusion problem on an n_pix^2 x n_lambda cube observed by `bands` with the 4-point dither.
    ``lam_stride`` > 1 keeps every lam_stride-th cube plane (CPU-baseline sample); ``geometry_only`` leaves out the arrays
    (OTF, templates, maps) -- enough to plan the multi-GPU partition."""
    wav = np.linspace(lam_range[0], lam_range[1], n_lambda)[::lam_stride]
    ax = axes(n_pix)
    ifus = [band_ifu(b) for b in bands]
    if geometry_only:
        return dict(bands=list(bands), alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, ifus=ifus,
                    pointings=[dither4(i) for i in ifus], step_deg=STEP_DEG)
    return dict(bands=list(bands), alpha_axis=ax, beta_axis=ax.copy(), wavel=wav, ifus=ifus,
                pointings=[dither4(i) for i in ifus], templates=templates(n_lambda)[:, ::lam_stride],
                sotf=ir2fr(gaussian_psf(wav, STEP), (n_pix, n_pix)), step_deg=STEP_DEG,
                maps=np.random.default_rng(19940407).random((4, n_pix, n_pix)))


def config2(**kw) -> dict:
    """BASELINE.json configs[1]: single channel 2A, 251x251x1024 on [7.41, 8.87] um."""
    return problem(["2a"], 1024, (7.41, 8.87), **kw)


def config3(**kw) -> dict:
    """BASELINE.json configs[2] / the headline metric: 251x251x4000 on [1C[0], 2C[-1]], bands 1C,2A,2B,2C
    (scripts/fusion/fusion_largeMCMO_SigRLSCT_NN_simulated.py:127-134)."""
    return problem(["1c", "2a", "2b", "2c"], 4000, (band_wavelengths("1c")[0], band_wavelengths("2c")[-1]), **kw)


def config4(**kw) -> dict:
    """BASELINE.json configs[3]: all 12 sub-bands, 251x251x8000 on [4.90, 28.70] um."""
    return problem(list(BANDS), 8000, (4.90, 28.70), **kw)
