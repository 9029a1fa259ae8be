"""CPU oracle for the surfh forward/adjoint + CG hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a float64 NumPy/SciPy restatement of the reference algorithm
(sidiso/surfh @ 2025-02-04).  It exists to CHECK the HIP product path and to
serve as the timed CPU baseline in ``bench.py``; it is never the thing that is
shipped.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``surfh_amd/`` imports it.

Pinning: every stage below is compared, in this container, against the real
reference modules imported from ``/root/reference`` (``oracle/ref_harness.py``)
and against the golden vectors those modules produced
(``tests/golden/*.npz``, generator ``tests/golden/make_golden.py``).
Two third-party pieces are absent from the reference tree and are restated
from their published behaviour: ``udft.ir2fr`` (udft 3.4.0) and ``qmm.lcg``
(qmm 0.18.2).  ``lcg`` has no reference-side golden: *parity unpinned* for the
solver (checked against a dense solve instead).

Style: procedural, table driven.  It deliberately does not share code with
``surfh_amd`` (the product's host side mirrors the reference's class API
instead), so the two are independent implementations of the same geometry.

All ``file:line`` citations are relative to ``/root/reference``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import ceil, floor
from typing import List, Sequence, Tuple

import numpy as np
import scipy.fft as sfft


# ----------------------------------------------------------------------------
# Fourier helpers
# ----------------------------------------------------------------------------
def dft(x):
    """Unitary rfft2 on the last two axes (surfh/ToolsDir/python_utils.py:59-71)."""
    return sfft.rfftn(x, axes=(-2, -1), norm="ortho", workers=-1)


def idft(xf, shape):
    """Unitary irfft2 (surfh/ToolsDir/python_utils.py:41-57)."""
    return sfft.irfftn(xf, s=tuple(shape), axes=(-2, -1), norm="ortho", workers=-1)


def ir2fr(imp_resp, shape, real=True):
    """Restatement of ``udft.ir2fr`` (udft 3.4.0, absent from the reference tree).

    Zero-pad the impulse response to ``shape`` on the last ``len(shape)`` axes,
    roll its centre ``floor(n/2)`` to index 0, un-normalised (r)fftn.
    Call sites: surfh/Models/spectroModelChannel.py:81-83,
    scripts/main_fusion.py:98, test/test_fw_ad.py:507.
    """
    imp_resp = np.asarray(imp_resp)
    nd = len(shape)
    lead = imp_resp.shape[:-nd]
    padded = np.zeros(lead + tuple(shape), dtype=imp_resp.dtype)
    padded[(Ellipsis,) + tuple(slice(0, s) for s in imp_resp.shape[-nd:])] = imp_resp
    for k, n in enumerate(imp_resp.shape[-nd:]):
        padded = np.roll(padded, -int(np.floor(n / 2)), axis=imp_resp.ndim - nd + k)
    if real:
        return np.fft.rfftn(padded, axes=tuple(range(-nd, 0)))
    return np.fft.fftn(padded, axes=tuple(range(-nd, 0)))


# ----------------------------------------------------------------------------
# LMM (surfh/ToolsDir/python_utils.py:11-35)
# ----------------------------------------------------------------------------
def lmm_maps2cube(maps, tpls):
    return np.tensordot(tpls.T, maps, axes=(1, 0))


def lmm_cube2maps(cube, tpls):
    return np.tensordot(tpls, cube, axes=(1, 0))


# ----------------------------------------------------------------------------
# Geometry (surfh/Models/instru.py)
# ----------------------------------------------------------------------------
def rotmatrix(degree):
    """instru.py:36-45"""
    t = np.radians(degree)
    return np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])


def pix(value, step):
    """Coord.pix: python (banker's) round to the grid (instru.py:143-145)."""
    return round(value / step) * step


def get_srf(det_pix_size, step_arcsec):
    """instru.py:67-84"""
    return int(det_pix_size // step_arcsec)


def fov_local_axis(width, margin, step):
    """One axis of FOV.local_coords (instru.py:283-304)."""
    start = -width / 2 - margin
    length = width + 2 * margin
    round_start = int(floor(start / step)) * step
    num = int(ceil((length + (start - round_start)) / step))
    return np.arange(num + 1) * step + round_start


def wslice_of(wavel_in, wmin, wmax, margin):
    """IFU.wslice (instru.py:649-658); stop excludes the last in-range plane."""
    lo = np.flatnonzero(wavel_in <= max(wmin - margin, wavel_in.min()))[-1]
    hi = np.flatnonzero(wavel_in >= min(wmax + margin, wavel_in.max()))[0]
    return int(lo), int(hi)


def local2global(la, lb, angle, origin):
    """FOV.local2global (instru.py:306-321)."""
    na, nb = len(la), len(lb)
    A = np.tile(la.reshape((-1, 1)), [1, nb])
    B = np.tile(lb.reshape((1, -1)), [na, 1])
    c = rotmatrix(angle) @ np.vstack((A.ravel(), B.ravel()))
    return c[0].reshape((na, nb)) + origin[0], c[1].reshape((na, nb)) + origin[1]


def global2local(ga, gb, angle, origin):
    """FOV.global2local (instru.py:323-340)."""
    na, nb = len(ga), len(gb)
    ga = ga - origin[0]
    gb = gb - origin[1]
    A = np.tile(ga.reshape((-1, 1)), [1, nb])
    B = np.tile(gb.reshape((1, -1)), [na, 1])
    c = rotmatrix(-angle) @ np.vstack((A.ravel(), B.ravel()))
    return c[0].reshape((na, nb)), c[1].reshape((na, nb))


def spectral_psf(grating_resolution, out_axis, beta, wavelength, scale, n_margin=15, kind="mrs"):
    """SpectralBlur.psfs (instru.py:499-572); kind='dirac' (:564-570) keeps a 1 at the peak of every (lambda', beta)
    response on the margin-extended axis -- the nearest-wavelength selector of Channel.sliceToCube.

    Returns W[lambda', lambda, beta]; normalised over the margin-extended input
    axis and then cropped (row sums are < 1 near the ends of the axis).
    Note ``np.sinc(np.pi * z)`` = sin(pi^2 z)/(pi^2 z): kept as in the reference.
    """
    grating_len = 2 * 0.44245 / np.pi * grating_resolution
    delta_w = min(np.diff(wavelength))
    beta = np.asarray(beta).reshape((1, 1, -1))
    out_axis = np.asarray(out_axis).reshape((-1, 1, 1))
    wavelength = np.asarray(wavelength)
    w_norm = np.concatenate(
        [
            np.linspace(wavelength.min() - n_margin * delta_w, wavelength.min() - delta_w, n_margin - 1),
            wavelength,
            np.linspace(wavelength.max() + delta_w, wavelength.max() + n_margin * delta_w, n_margin - 1),
        ]
    ).reshape((1, -1, 1))
    out = (np.pi * grating_len / w_norm
           * np.sinc(np.pi * grating_len * ((out_axis - scale * beta) / w_norm - 1)) ** 2)
    out /= np.sum(out, axis=1, keepdims=True)
    if kind == "dirac":
        out = (out == np.max(out, axis=1, keepdims=True)).astype(np.float64)
    return out[:, n_margin - 1: -n_margin + 1, :]


# ----------------------------------------------------------------------------
# Bilinear tables (surfh/ToolsDir/cythons_files.pyx:20-154)
# ----------------------------------------------------------------------------
def find_indices(axis, values):
    """Interval index and normalised distance for each value.

    ``x[i] <= v < x[i+1]`` by search on the ACTUAL axis values, interval closed
    on the right end, out-of-range clamped to the first/last interval
    (find_interval_ascending with extrapolate=1, cythons_files.pyx:20-103).
    """
    axis = np.asarray(axis, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    n = len(axis)
    idx = np.clip(np.searchsorted(axis, v, side="right") - 1, 0, n - 2)
    frac = (v - axis[idx]) / (axis[idx + 1] - axis[idx])
    return idx.astype(np.int64), frac


def nearest_index(axis, values):
    """Index of the nearest axis sample (what the cKDTree query of
    nearest_neighbor_interpolation.py:16-20,177,199 returns on a separable grid)."""
    axis = np.asarray(axis, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64).ravel()
    return np.abs(axis[None, :] - v[:, None]).argmin(axis=1)


def bilinear_apply(cube, i0, i1, y0, y1):
    """solve_2D_hypercube (cythons_files.pyx:163-193): same weights, same order."""
    w1 = (1.0 - y0) * (1.0 - y1)
    w2 = (1.0 - y0) * y1
    w3 = y0 * (1.0 - y1)
    w4 = y0 * y1
    out = cube[:, i0, i1] * w1
    out = out + cube[:, i0, i1 + 1] * w2
    out = out + cube[:, i0 + 1, i1] * w3
    out = out + cube[:, i0 + 1, i1 + 1] * w4
    return out


# ----------------------------------------------------------------------------
# Channel description and tables
# ----------------------------------------------------------------------------
@dataclass
class ChannelSpec:
    """What the reference passes as ``instru.IFU`` (instru.py:575-697)."""
    alpha_width: float          # degrees
    beta_width: float           # degrees
    origin: Tuple[float, float]  # degrees
    angle: float                # degrees
    det_pix_size: float         # arcsec
    n_slit: int
    grating_resolution: float
    wavel_axis: np.ndarray      # detector lambda' axis
    name: str = "_"


@dataclass
class ChannelTables:
    spec: ChannelSpec
    srf: int
    origin_pix: Tuple[float, float]
    pointings: List[Tuple[float, float]]
    wslice: Tuple[int, int]
    local_alpha_axis: np.ndarray
    local_beta_axis: np.ndarray
    npix_slit_alpha_width: int
    npix_slit_beta_width: int
    n_alpha_out: int
    slit_slices: List[Tuple[int, int, int, int]]     # (a0, a1, b0, b1) per slit
    slit_weights: List[np.ndarray]                   # [na_s, nb_s] per slit
    wpsf: np.ndarray                                 # [Ldet, Lin, nbeta]
    oshape: Tuple[int, int, int, int]
    grid_idx: List[Tuple[np.ndarray, np.ndarray]] = field(default_factory=list)   # per pointing (i0, i1) [na*nb]
    grid_frac: List[Tuple[np.ndarray, np.ndarray]] = field(default_factory=list)  # per pointing (y0, y1)
    nn_idx: List[np.ndarray] = field(default_factory=list)     # per pointing: flat C-order cube index [na*nb]
    nn_idx_t: List[np.ndarray] = field(default_factory=list)   # per pointing: flat local index per cube pixel [Na*Nb]
    gridding: str = "bilinear"
    wpsf_dirac: np.ndarray = None                    # [Ldet, Lin, nbeta], one-hot (spectroModelChannel.py:92-95,145-155)


def _slit_local_fov(spec: ChannelSpec, s: int):
    """Slicer.slit_local_fov (slicer.py:87-90) -> (alpha_start, alpha_end, beta_start, beta_end).

    LocalFOV is centred at (0,0)+slit_shift; beta bounds are rounded to 9 decimals
    (instru.py:420-434), alpha bounds are not.
    """
    sbw = spec.beta_width / spec.n_slit
    shift_beta = (-spec.beta_width / 2 + sbw / 2) + s * sbw   # IFU.__post_init__ instru.py:612-617
    ob = 0 + shift_beta
    a_start = 0 - spec.alpha_width / 2
    a_end = 0 + spec.alpha_width / 2
    b_start = round(ob - sbw / 2, 9)
    b_end = round(ob + sbw / 2, 9)
    return a_start, a_end, b_start, b_end


def _to_slices(fovb, la, lb):
    """LocalFOV.to_slices (instru.py:436-459)."""
    a_start, a_end, b_start, b_end = fovb
    astep = la[1] - la[0]
    bstep = lb[1] - lb[0]
    a0 = np.flatnonzero(a_start < la + astep / 2)[0]
    a1 = np.flatnonzero(la - astep / 2 < a_end)[-1] + 1
    b0 = np.flatnonzero(b_start < lb + bstep / 2)[0]
    b1 = np.flatnonzero(lb - bstep / 2 < b_end)[-1] + 1
    return int(a0), int(a1), int(b0), int(b1)


def _slit_slices(spec, s, la, lb, npix_a, npix_b, n_alpha_out):
    """Slicer.get_slit_slices incl. both trimming hacks (slicer.py:118-145)."""
    fovb = _slit_local_fov(spec, s)
    a0, a1, b0, b1 = _to_slices(fovb, la, lb)
    if (b1 - b0) > npix_b:
        if abs(lb[b1] - fovb[3]) > abs(lb[b0] - fovb[2]):
            b1 -= 1
        else:
            b0 += 1
    if n_alpha_out % 2 == 0 and n_alpha_out < 28:
        if (a1 - a0) > npix_a:
            a1 -= 1
        elif (a1 - a0) < npix_a:
            a0 -= 2
    return a0, a1, b0, b1


def _fov_weight(fovb, sl, la, lb):
    """Slicer.fov_weight (slicer.py:187-244): only the beta edges are weighted."""
    a0, a1, b0, b1 = sl
    bstep = lb[1] - lb[0]
    sel_b = lb[b0:b1]
    w = np.ones((a1 - a0, b1 - b0))
    if sel_b[0] - bstep / 2 < fovb[2]:
        wg = 1 - abs(sel_b[0] - bstep / 2 - fovb[2]) / bstep
        assert 0 <= wg <= 1
        w[:, 0] = wg
    if sel_b[-1] + bstep / 2 > fovb[3]:
        wg = 1 - abs(sel_b[-1] + bstep / 2 - fovb[3]) / bstep
        assert 0 <= wg <= 1
        w[:, -1] = wg
    return w


def build_channel(spec: ChannelSpec, alpha_axis, beta_axis, wavel_axis, step_degree,
                  pointings: Sequence[Tuple[float, float]], with_grid=True, gridding="bilinear",
                  lam_slice=None) -> ChannelTables:
    """Everything Channel.__init__ / Slicer derive (spectroModelChannel.py:27-108)."""
    srf = get_srf(spec.det_pix_size, step_degree * 3600)               # spectroModel.py:67-70
    origin_pix = (pix(spec.origin[0], step_degree), pix(spec.origin[1], step_degree))   # IFU.pix
    pts = [(pix(p[0], step_degree), pix(p[1], step_degree)) for p in pointings]       # CoordList.pix
    la = fov_local_axis(spec.alpha_width, 5 * step_degree, step_degree)
    lb = fov_local_axis(spec.beta_width, 5 * step_degree, step_degree)
    ws = wslice_of(wavel_axis, spec.wavel_axis[0], spec.wavel_axis[-1], 0.1)

    bstep = beta_axis[1] - beta_axis[0]
    sbw = spec.beta_width / spec.n_slit
    npix_b = int(ceil(sbw / bstep))                                     # slicer.py:45-48
    lstep = la[1] - la[0]
    npix_a = int(ceil(spec.alpha_width / 2 / lstep)) - int(floor(-spec.alpha_width / 2 / lstep))  # :54-62
    n_alpha_out = ceil(npix_a / srf)

    slices = [_slit_slices(spec, s, la, lb, npix_a, npix_b, n_alpha_out) for s in range(spec.n_slit)]
    weights = []
    for s, sl in enumerate(slices):                                      # slicer.py:148-168
        w = _fov_weight(_slit_local_fov(spec, s), sl, la, lb)
        if s > 0 and slices[s - 1][3] - 1 != sl[2]:
            w[:, 0] = 1
        if s < spec.n_slit - 1 and sl[3] - 1 != slices[s + 1][2]:
            w[:, -1] = 1
        weights.append(w)

    beta_in_slit = np.arange(0, npix_b) * bstep                         # spectroModelChannel.py:133-143
    wpsf = spectral_psf(spec.grating_resolution, spec.wavel_axis,
                        beta_in_slit - np.mean(beta_in_slit), wavel_axis[ws[0]:ws[1]],
                        scale=(spec.wavel_axis[1] - spec.wavel_axis[0]) / spec.det_pix_size)
    wpsf_dirac = spectral_psf(spec.grating_resolution, spec.wavel_axis,
                              beta_in_slit - np.mean(beta_in_slit), wavel_axis[ws[0]:ws[1]],
                              scale=(spec.wavel_axis[1] - spec.wavel_axis[0]) / spec.det_pix_size, kind="dirac")
    if lam_slice is not None:      # one of n contiguous parts of the window (the parts' outputs add up)
        lin = ws[1] - ws[0]
        if len(lam_slice) == 3:        # ("planes", a, b)
            _, a, b = lam_slice
        else:                          # (i, n)
            i, n = lam_slice
            a, b = (lin * i) // n, (lin * (i + 1)) // n
        wpsf = wpsf[:, a:b, :]
        wpsf_dirac = wpsf_dirac[:, a:b, :]
        ws = (ws[0] + a, ws[0] + b)

    tab = ChannelTables(spec=spec, srf=srf, origin_pix=origin_pix, pointings=pts, wslice=ws,
                        local_alpha_axis=la, local_beta_axis=lb,
                        npix_slit_alpha_width=npix_a, npix_slit_beta_width=npix_b,
                        n_alpha_out=n_alpha_out, slit_slices=slices, slit_weights=weights, wpsf=wpsf,
                        oshape=(len(pts), spec.n_slit, len(spec.wavel_axis), n_alpha_out), wpsf_dirac=wpsf_dirac)
    if with_grid:
        for p in pts:
            ga, gb = local2global(la, lb, spec.angle, (origin_pix[0] + p[0], origin_pix[1] + p[1]))
            for k, (ax, v) in enumerate(((alpha_axis, ga), (beta_axis, gb))):
                if not (np.all(ax[0] <= v) and np.all(v <= ax[-1])):   # cython_2D_interpolation.py:472-478
                    raise ValueError("One of the requested xi is out of bounds in dimension %d" % k)
            i0, y0 = find_indices(alpha_axis, ga.ravel())
            i1, y1 = find_indices(beta_axis, gb.ravel())
            tab.grid_idx.append((i0, i1))
            tab.grid_frac.append((y0, y1))
            if gridding != "bilinear":
                # NN index recipe of Channel.precompute_mask (spectroModelChannel.py:399-415).  The reference forms
                # k = i_beta*N + i_alpha and applies it to the C-order cube ("nn_ref"); "nn" is the untransposed form.
                N = len(beta_axis)
                ia, ib = nearest_index(alpha_axis, ga), nearest_index(beta_axis, gb)
                tab.nn_idx.append(ib * N + ia if gridding == "nn_ref" else ia * N + ib)
                ca, cb = global2local(alpha_axis, beta_axis, spec.angle, (origin_pix[0] + p[0], origin_pix[1] + p[1]))
                if gridding == "nn_ref":
                    ca, cb = ca.T, cb.T
                tab.nn_idx_t.append(nearest_index(la, ca) * len(lb) + nearest_index(lb, cb))
    tab.gridding = gridding
    return tab


# ----------------------------------------------------------------------------
# Channel operators (surfh/Models/spectroModelChannel.py)
# ----------------------------------------------------------------------------
def gridding(tab: ChannelTables, sub_cube, p):
    """Channel.gridding (:158-177): bilinear cube -> rotated local grid."""
    na, nb = len(tab.local_alpha_axis), len(tab.local_beta_axis)
    if tab.gridding != "bilinear":       # NN_gridding (:201-205)
        return sub_cube.reshape(sub_cube.shape[0], -1)[:, tab.nn_idx[p]].reshape(sub_cube.shape[0], na, nb)
    (i0, i1), (y0, y1) = tab.grid_idx[p], tab.grid_frac[p]
    return bilinear_apply(sub_cube, i0, i1, y0, y1).reshape(sub_cube.shape[0], na, nb)


def gridding_t_ref(tab: ChannelTables, local_cube, p, alpha_axis, beta_axis):
    """Channel.gridding_t (:180-199): the reference's *interpolating* back-projection
    (bilinear local -> global, 0 outside).  NOT the transpose of ``gridding``."""
    if tab.gridding != "bilinear":       # NN_gridding_t (:208-212): gather back, no support mask
        L = local_cube.shape[0]
        return local_cube.reshape(L, -1)[:, tab.nn_idx_t[p]].reshape(L, len(alpha_axis), len(beta_axis))
    org = (tab.origin_pix[0] + tab.pointings[p][0], tab.origin_pix[1] + tab.pointings[p][1])
    ca, cb = global2local(alpha_axis, beta_axis, tab.spec.angle, org)
    la, lb = tab.local_alpha_axis, tab.local_beta_axis
    i0, y0 = find_indices(la, ca.ravel())
    i1, y1 = find_indices(lb, cb.ravel())
    out = bilinear_apply(local_cube, i0, i1, y0, y1)
    oob = (ca.ravel() < la[0]) | (ca.ravel() > la[-1]) | (cb.ravel() < lb[0]) | (cb.ravel() > lb[-1])
    out[:, oob] = 0                                                 # cython_2D_interpolation.py:322-323
    return out.reshape(local_cube.shape[0], len(alpha_axis), len(beta_axis))


def gridding_T(tab: ChannelTables, local_cube, p, n_alpha, n_beta):
    """Exact transpose of ``gridding`` (scatter-add of the same four weights)."""
    L = local_cube.shape[0]
    v = local_cube.reshape(L, -1)
    out = np.zeros((L, n_alpha * n_beta))
    if tab.gridding != "bilinear":       # transpose of the index gather = scatter-add
        for l in range(L):
            out[l] = np.bincount(tab.nn_idx[p], weights=v[l], minlength=n_alpha * n_beta)
        return out.reshape(L, n_alpha, n_beta)
    (i0, i1), (y0, y1) = tab.grid_idx[p], tab.grid_frac[p]
    if L > 64:      # many planes: the same scatter-add as one sparse product (identical sums up to float64 rounding order)
        import scipy.sparse as sp
        nloc = v.shape[1]
        rows = np.concatenate([(i0 + di) * n_beta + (i1 + dj) for di, dj in ((0, 0), (0, 1), (1, 0), (1, 1))])
        vals = np.concatenate([(1 - y0) * (1 - y1), (1 - y0) * y1, y0 * (1 - y1), y0 * y1])
        mat = sp.csr_matrix((vals, (rows, np.tile(np.arange(nloc), 4))), shape=(n_alpha * n_beta, nloc))
        return np.ascontiguousarray((mat @ v.T).T).reshape(L, n_alpha, n_beta)
    for di, dj, w in ((0, 0, (1 - y0) * (1 - y1)), (0, 1, (1 - y0) * y1),
                      (1, 0, y0 * (1 - y1)), (1, 1, y0 * y1)):
        flat = (i0 + di) * n_beta + (i1 + dj)
        for l in range(L):
            out[l] += np.bincount(flat, weights=v[l] * w, minlength=n_alpha * n_beta)
    return out.reshape(L, n_alpha, n_beta)


def _box_filters(tab: ChannelTables):
    """_otf_sr (:81-83) and decalf (:104-108) exactly as the reference forms them."""
    shp = (len(tab.local_alpha_axis), len(tab.local_beta_axis))
    otf_sr = ir2fr(np.ones((tab.srf, 1)), shp)[np.newaxis, ...]
    decal = np.zeros(shp)
    dsi = int((tab.srf - 1) / 2)
    decal[-dsi, -0] = np.sqrt(shp[0] * shp[1])
    return otf_sr, dft(decal)


def box_sum_fft(tab: ChannelTables, local_cube):
    """sum_cube (:220-223), done literally through the two FFTs."""
    otf_sr, decalf = _box_filters(tab)
    shp = local_cube.shape[-2:]
    return idft(dft(local_cube) * (otf_sr * decalf), shp)


def box_sum_fft_t(tab: ChannelTables, local_cube):
    """:258-259"""
    otf_sr, decalf = _box_filters(tab)
    shp = local_cube.shape[-2:]
    return idft(dft(local_cube) * otf_sr.conj() * decalf.conj(), shp)


def box_sum_direct(tab: ChannelTables, local_cube):
    """The same operator as a circular window sum: y[n] = sum_{j<srf} x[(n+j) mod na]."""
    return sum(np.roll(local_cube, -j, axis=1) for j in range(tab.srf))


def box_sum_direct_t(tab: ChannelTables, local_cube):
    return sum(np.roll(local_cube, j, axis=1) for j in range(tab.srf))


def slicing(tab: ChannelTables, cube, s):
    """Slicer.slicing (slicer.py:64-68)."""
    a0, a1, b0, b1 = tab.slit_slices[s]
    return cube[:, a0:a1, b0:b1] * tab.slit_weights[s][np.newaxis]


def slicing_t(tab: ChannelTables, slit, s, local_shape):
    """Slicer.slicing_t (slicer.py:72-84)."""
    out = np.zeros(local_shape)
    a0, a1, b0, b1 = tab.slit_slices[s]
    out[:, a0:a1, b0:b1] = slit * tab.slit_weights[s][np.newaxis]
    return out


def wblur_subsampling(sliced, wpsf):
    """jax_utils.wblur_subSampling (:72-80): out[l',a] = sum_{l,b} W[l',l,b] x[l,a,b]."""
    return np.einsum("klb,lab->ka", wpsf, sliced, optimize=True)


_WT_CACHE = {}


def wblur_t(arr, wpsf):
    """jax_utils.wblur_t (:83-91): x[l,a,b] = sum_l' y[l',a,b] W[l',l,b]."""
    key = (id(wpsf), wpsf.shape)
    wT = _WT_CACHE.get(key)
    if wT is None or wT[0] is not wpsf:                # [b][l][l'] contiguous, built once per spectral PSF
        if len(_WT_CACHE) > 32:
            _WT_CACHE.clear()
        wT = (wpsf, np.ascontiguousarray(wpsf.transpose(2, 1, 0)))
        _WT_CACHE[key] = wT
    # one batched BLAS product over the beta columns: out[b][l][a] = sum_l' W[l'][l][b] y[l'][a][b]
    return np.ascontiguousarray(np.matmul(wT[1], np.ascontiguousarray(arr.transpose(2, 0, 1))).transpose(1, 2, 0))


def channel_forward(tab: ChannelTables, blurred_cube, box="fft", stages=None):
    """Channel.forward (:215-231)."""
    out = np.zeros(tab.oshape)
    sub = blurred_cube[tab.wslice[0]:tab.wslice[1]]
    bs = box_sum_fft if box == "fft" else box_sum_direct
    for p in range(len(tab.pointings)):
        g = gridding(tab, sub, p)
        sc = bs(tab, g)
        if stages is not None:
            stages.setdefault("gridded", []).append(g)
            stages.setdefault("sum_cube", []).append(sc)
        for s in range(tab.spec.n_slit):
            sl = slicing(tab, sc, s)
            out[p, s] = wblur_subsampling(sl, tab.wpsf)[:, : tab.oshape[3] * tab.srf: tab.srf]
    return out.ravel()


def channel_adjoint(tab: ChannelTables, y, alpha_axis, beta_axis, mode="exact", box="fft", stages=None):
    """Channel.adjoint (:234-264).

    mode="ref"   : reference behaviour, S back-projection by interpolation (gridding_t).
    mode="exact" : true transpose of channel_forward (scatter-add S^T).
    """
    Lin = tab.wslice[1] - tab.wslice[0]
    na, nb = len(tab.local_alpha_axis), len(tab.local_beta_axis)
    a0, a1, b0, b1 = tab.slit_slices[0]
    slit_shape = (Lin, a1 - a0, b1 - b0)                      # get_slit_shape_t (slicer.py:179-185)
    y = y.reshape(tab.oshape)
    inter = np.zeros((Lin, len(alpha_axis), len(beta_axis)))
    bst = box_sum_fft_t if box == "fft" else box_sum_direct_t
    wconj = tab.wpsf if not np.iscomplexobj(tab.wpsf) else tab.wpsf.conj()      # :251 wpsf.conj(); real in practice
    for p in range(len(tab.pointings)):
        local = np.zeros((Lin, na, nb))
        for s in range(tab.spec.n_slit):
            over = np.repeat(y[p, s][:, :, None], tab.npix_slit_beta_width, axis=2)
            bts = np.zeros(slit_shape)
            bts[:, : tab.oshape[3] * tab.srf: tab.srf, :] = wblur_t(over, wconj)
            sa0, sa1, sb0, sb1 = tab.slit_slices[s]            # local += slicing_t(...) without the full-size temporary
            local[:, sa0:sa1, sb0:sb1] += bts * tab.slit_weights[s][np.newaxis]
        st = bst(tab, local)
        if stages is not None:
            stages.setdefault("local_cube", []).append(local)
            stages.setdefault("sum_t", []).append(st)
        if mode == "ref":
            dg = gridding_t_ref(tab, st, p, alpha_axis, beta_axis)
        else:
            dg = gridding_T(tab, st, p, len(alpha_axis), len(beta_axis))
        if stages is not None:
            stages.setdefault("degridded", []).append(dg)
        inter += dg
    return inter


def slice_to_cube(tab: ChannelTables, data, alpha_axis, beta_axis, n_lambda):
    """Channel.sliceToCube (:266-301): the data of pointing 0 sent back to the cube through the one-hot spectral
    selector instead of the spectral PSF, then the reference's adjoint chain (slicing_t, transposed box sum,
    interpolating gridding_t); planes outside the channel's window stay 0."""
    Lin = tab.wslice[1] - tab.wslice[0]
    na, nb = len(tab.local_alpha_axis), len(tab.local_beta_axis)
    a0, a1, b0, b1 = tab.slit_slices[0]
    y = np.asarray(data, dtype=np.float64).reshape(tab.oshape)
    local = np.zeros((Lin, na, nb))
    for s in range(tab.spec.n_slit):
        over = np.repeat(y[0, s][:, :, None], tab.npix_slit_beta_width, axis=2)
        bts = np.zeros((Lin, a1 - a0, b1 - b0))
        bts[:, : tab.oshape[3] * tab.srf: tab.srf, :] = wblur_t(over, tab.wpsf_dirac)
        local += slicing_t(tab, bts, s, (Lin, na, nb))
    out = np.zeros((n_lambda, len(alpha_axis), len(beta_axis)))
    out[tab.wslice[0]:tab.wslice[1]] = gridding_t_ref(tab, box_sum_fft_t(tab, local), 0, alpha_axis, beta_axis)
    return out


def realdata_cube_to_slice(tab0: ChannelTables, cube):
    """Channel.realData_cubeToSlice (:303-309).  ``tab0`` is the channel built with the single pointing (0, 0);
    every plane of ``cube`` (one per detector wavelength) is gridded, cut into slits with the edge weights, decimated
    along alpha WITHOUT the box sum and summed over beta."""
    g = gridding(tab0, np.asarray(cube, dtype=np.float64), 0)
    out = np.zeros(tab0.oshape[1:])
    for s in range(tab0.spec.n_slit):
        out[s] = slicing(tab0, g, s)[:, : tab0.oshape[3] * tab0.srf: tab0.srf, :].sum(axis=2)
    return out


def realdata_slice_to_cube(tab0: ChannelTables, slices, cube_dim, alpha_axis, beta_axis):
    """Channel.realData_sliceToCube (:311-336): each slit value spread evenly over the slit's beta columns, zero-stuffed
    along alpha, slicing_t, correlation with the box kernel (``_otf_sr.conj()`` alone -- no ``decalf`` here), gridding_t."""
    L = cube_dim[0]
    na, nb = len(tab0.local_alpha_axis), len(tab0.local_beta_axis)
    a0, a1, b0, b1 = tab0.slit_slices[0]
    local = np.zeros((L, na, nb))
    for s in range(tab0.spec.n_slit):
        sl = np.zeros((L, a1 - a0, b1 - b0))
        sl[:, ::tab0.srf] = np.repeat(np.asarray(slices)[s][:, :, None], tab0.npix_slit_beta_width, axis=2) / tab0.npix_slit_beta_width
        local += slicing_t(tab0, sl, s, (L, na, nb))
    otf_sr, _ = _box_filters(tab0)
    st = idft(dft(local) * otf_sr.conj(), (na, nb))
    return gridding_t_ref(tab0, st, 0, alpha_axis, beta_axis)


# ----------------------------------------------------------------------------
# Full operator (surfh/Models/spectroModel.py)
# ----------------------------------------------------------------------------
class OracleModel:
    """spectroSigRLSCT (spectroModel.py:39-185) in float64."""

    def __init__(self, sotf, templates, alpha_axis, beta_axis, wavelength_axis,
                 specs: Sequence[ChannelSpec], step_degree, pointings, box="fft", gridding="bilinear",
                 lam_slices=None):
        self.sotf = np.asarray(sotf)
        self.templates = None if templates is None else np.asarray(templates, dtype=np.float64)
        self.alpha_axis = np.asarray(alpha_axis, dtype=np.float64)
        self.beta_axis = np.asarray(beta_axis, dtype=np.float64)
        self.wavelength_axis = np.asarray(wavelength_axis, dtype=np.float64)
        self.box = box
        self.channels = [build_channel(sp, self.alpha_axis, self.beta_axis, self.wavelength_axis,
                                       step_degree, pointings[k], gridding=gridding,
                                       lam_slice=None if lam_slices is None else lam_slices[k])
                         for k, sp in enumerate(specs)]
        n_lead = self.templates.shape[0] if self.templates is not None else len(self.wavelength_axis)
        self.ishape = (n_lead, len(self.alpha_axis), len(self.beta_axis))
        self.cube_shape = (len(self.wavelength_axis), len(self.alpha_axis), len(self.beta_axis))
        self._idx = np.cumsum([0] + [int(np.prod(c.oshape)) for c in self.channels])
        self.oshape = (int(self._idx[-1]),)

    @property
    def isize(self):
        return int(np.prod(self.ishape))

    @property
    def osize(self):
        return int(np.prod(self.oshape))

    def blur(self, maps):
        cube = lmm_maps2cube(maps, self.templates) if self.templates is not None else maps   # T  :161
        return idft(dft(cube) * self.sotf, self.ishape[1:])                                  # C  :166

    def forward(self, maps, stages=None):
        blurred = self.blur(np.asarray(maps, dtype=np.float64).reshape(self.ishape))
        if stages is not None:
            stages["blurred"] = blurred
        out = np.zeros(self.oshape)
        for k, tab in enumerate(self.channels):                                              # :168-169
            out[self._idx[k]: self._idx[k + 1]] = channel_forward(tab, blurred, self.box, stages)
        return out

    def _adjoint(self, y, mode, stages=None):
        g = np.zeros(self.cube_shape)
        for k, tab in enumerate(self.channels):                                              # :175-176
            g[tab.wslice[0]:tab.wslice[1]] += channel_adjoint(
                tab, np.asarray(y, dtype=np.float64)[self._idx[k]: self._idx[k + 1]],
                self.alpha_axis, self.beta_axis, mode, self.box, stages)
        if stages is not None:
            stages["global_cube"] = g
        bt = idft(dft(g) * self.sotf.conj(), self.ishape[1:])                                # :178
        return lmm_cube2maps(bt, self.templates) if self.templates is not None else bt       # :181

    def adjoint(self, y, stages=None):
        """Exact transpose of ``forward`` (what CG and the dot-test use)."""
        return self._adjoint(y, "exact", stages)

    def adjoint_ref(self, y, stages=None):
        """Reference behaviour (interpolating ``gridding_t``)."""
        return self._adjoint(y, "ref", stages)

    # aljabr.LinOp flat-vector forms used by dottest (test/sandbox_dottest.py:16-27)
    def matvec(self, x):
        return self.forward(x.reshape(self.ishape)).ravel()

    def rmatvec(self, y):
        return self.adjoint(y.reshape(self.oshape)).ravel()


def dottest_gap(op, rng, num=1):
    """|<A^T u, v> - <u, A v>| / |<u, A v>|  (aljabr.dottest restated, test/sandbox_dottest.py:16-27)."""
    worst = 0.0
    for _ in range(num):
        v = rng.standard_normal(op.isize)
        u = rng.standard_normal(op.osize)
        left = np.vdot(op.rmatvec(u), v)
        right = np.vdot(u, op.matvec(v))
        worst = max(worst, abs(left - right) / abs(right))
    return worst


# ----------------------------------------------------------------------------
# Priors and solver (surfh/Simulation/fusion_CT.py; qmm.lcg restated)
# ----------------------------------------------------------------------------
def diff_r(x):
    """NpDiff_r.forward (fusion_CT.py:23-25)."""
    return -np.diff(np.pad(x, ((0, 0), (1, 0), (0, 0)), "wrap"), axis=1)


def diff_r_t(y):
    """NpDiff_r.adjoint (:27-29)."""
    return np.diff(np.pad(y, ((0, 0), (0, 1), (0, 0)), "wrap"), axis=1)


def diff_c(x):
    """NpDiff_c.forward (:38-40)."""
    return -np.diff(np.pad(x, ((0, 0), (0, 0), (1, 0)), "wrap"), axis=2)


def diff_c_t(y):
    """NpDiff_c.adjoint (:42-43)."""
    return np.diff(np.pad(y, ((0, 0), (0, 0), (0, 1)), "wrap"), axis=2)


def normal_apply(op, x, mu, mu_reg):
    """Q x = mu A^T A x + mu_reg (Dr^T Dr + Dc^T Dc) x  -- the three QuadObjective of fusion_CT.py:130-162."""
    q = mu * op.adjoint(op.forward(x))
    if mu_reg:
        q = q + mu_reg * (diff_r_t(diff_r(x)) + diff_c_t(diff_c(x)))
    return q


def lcg(op, data, mu, mu_reg, x0, tol=1e-12, max_iter=10, refresh=50):
    """Linear CG as ``qmm.lcg`` runs it (qmm 0.18.2 is absent: parity UNPINNED).

    b = mu A^T y ; r = b - Qx ; d = r ; per iteration q = Qd, step = r.r / d.q,
    x += step d, r refreshed from scratch every ``refresh`` iterations (incl. the
    first) else r -= step q, d = r + (r'.r'/r.r) d, stop when sqrt(r.r) < size*tol.
    ``grad_norm`` holds r.r, one entry before the loop and one per iteration.
    Call site: surfh/Simulation/fusion_CT.py:194-225.
    """
    x = np.array(x0, dtype=np.float64, copy=True)
    b = mu * op.adjoint(data)
    r = b - normal_apply(op, x, mu, mu_reg)
    d = r.copy()
    grad_norm = [float(np.sum(r * r))]
    nit = 0
    for it in range(max_iter):
        q = normal_apply(op, d, mu, mu_reg)
        step = grad_norm[-1] / float(np.sum(d * q))
        x += step * d
        if refresh and it % refresh == 0:
            r = b - normal_apply(op, x, mu, mu_reg)
        else:
            r -= step * q
        grad_norm.append(float(np.sum(r * r)))
        d = r + (grad_norm[-1] / grad_norm[-2]) * d
        nit = it + 1
        if np.sqrt(grad_norm[-1]) < x.size * tol:
            break
    return {"x": x, "grad_norm": grad_norm, "nit": nit}


def mmmg(op, data, mu, mu_reg, x0, tol=1e-12, max_iter=10):
    """3MG as ``qmm.mmmg`` runs it on this path's three quadratic objectives (qmm 0.18.2 is absent: parity UNPINNED;
    restated from the published algorithm, Chouzenoux, Idier & Moussaoui 2011, with qmm's loop structure [memory]).

    Objectives: mu |y - A x|^2 (operator V = A), mu_reg |Dr x|^2, mu_reg |Dc x|^2.  Per iteration: grad from scratch,
    ``grad_norm`` <- |grad|, stop when it is below size*tol, directions D = [-grad, move], operator images
    V D = [V(-grad), (V D_prev) step_prev] per objective, step = -pinv(sum_k hyper_k (V_k D)^T (V_k D)) D^T grad
    (the quadratic objective is its own majorant), move = D step, x += move.
    Call site: surfh/Simulation/fusion_CT.py:194-225 (``function = mmmg``)."""
    x = np.array(x0, dtype=np.float64, copy=True)
    b = mu * op.adjoint(data)
    ops = [(mu, op.forward), (mu_reg, diff_r), (mu_reg, diff_c)]
    move = np.zeros_like(x)
    vd = [np.stack([np.zeros_like(f(x)).ravel()] * 2, axis=1) for _, f in ops]
    step = np.ones((2, 1))
    grad_norm = []
    nit = 0
    for it in range(max_iter + 1):
        grad = normal_apply(op, x, mu, mu_reg) - b
        grad_norm.append(float(np.sqrt(np.sum(grad * grad))))
        if it == max_iter or grad_norm[-1] < x.size * tol:
            break
        D = np.stack([-grad.ravel(), move.ravel()], axis=1)
        vd = [np.stack([f(-grad).ravel(), (v @ step).ravel()], axis=1) for (_, f), v in zip(ops, vd)]
        B = sum(h * (v.T @ v) for (h, _), v in zip(ops, vd))
        step = -np.linalg.pinv(B) @ (D.T @ grad.ravel()).reshape(2, 1)
        move = (D @ step).reshape(x.shape)
        x = x + move
        nit = it + 1
    return {"x": x, "grad_norm": grad_norm, "nit": nit}


def crit_val(op, data, x, mu, mu_reg):
    """QuadCriterion_MRS.get_crit_val (fusion_CT.py:242-265)."""
    d = mu * np.sum((data - op.forward(x)) ** 2)
    r = mu_reg * np.sum(diff_r(x) ** 2 + diff_c(x) ** 2)
    return (d + r) / 2


# ----------------------------------------------------------------------------
# Synthetic inputs (formulas from test/global_variable_testing.py:227-234,
# surfh/ToolsDir/utils.py:40-50, test/test_fw_ad.py:736-741)
# ----------------------------------------------------------------------------
def gaussian_psf(wavel_axis, step_arcsec, D=6.5):
    x = np.linspace(-30, 30, 40).reshape((1, -1))
    y = x.reshape((-1, 1))
    psf = np.empty((len(wavel_axis), 40, 40))
    for k, w in enumerate(wavel_axis):
        sigma = ((w * 1e-6 / D) * 206265) / (step_arcsec * 2.354)
        psf[k] = np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return psf / np.sum(psf, axis=(1, 2), keepdims=True)


def synthetic_axes(n, step_degree):
    ax = np.arange(n).astype(np.float64) * step_degree
    return ax - np.mean(ax)


def synthetic_templates(n_lambda):
    lam = np.arange(n_lambda, dtype=np.float64)
    c = (11.0, 15.0, 16.0, 17.0)
    return np.stack([(0.2 + 0.1 * t) * lam + c[t] for t in range(4)])


def dither4(det_pix_size_arcsec, slit_beta_width_deg):
    da = (det_pix_size_arcsec / 3600) / 4
    db = slit_beta_width_deg / 4
    return [(da, db), (-da, db), (da, -db), (-da, -db)]


# ----------------------------------------------------------------------------
# 2-D single-wavelength operator without rotation: MRSBlurred
# (surfh/Models/spectro_blind_rectangle.py:27-332)
# ----------------------------------------------------------------------------
class BlurredOracle:
    """``MRSBlurred``: C (2-D OTF), integer-crop gridding, srf-row window sum, slit window with beta-edge
    weights, alpha decimation, beta sum.  ``sotf`` may carry a leading wavelength axis: the planes are
    independent and are processed as a batch (BASELINE.json configs[4])."""

    def __init__(self, sotf, alpha_axis, beta_axis, spec: ChannelSpec, step_degree, pointings):
        self.sotf = np.asarray(sotf)
        self.batched = self.sotf.ndim == 3
        self.alpha_axis = np.asarray(alpha_axis, dtype=np.float64)
        self.beta_axis = np.asarray(beta_axis, dtype=np.float64)
        self.spec, self.pointings = spec, list(pointings)        # NOT pixelised (:38-39)
        self.srf = get_srf(spec.det_pix_size, step_degree * 3600)
        self.la = fov_local_axis(spec.alpha_width, 5 * step_degree, step_degree)
        self.lb = fov_local_axis(spec.beta_width, 5 * step_degree, step_degree)
        lstep = self.la[1] - self.la[0]
        self.npix_a = int(ceil(spec.alpha_width / 2 / lstep)) - int(floor(-spec.alpha_width / 2 / lstep))   # :90-98
        self.npix_b = int(ceil((spec.beta_width / spec.n_slit) / (self.beta_axis[1] - self.beta_axis[0])))   # :105-108
        self.n_out = ceil(self.npix_a / self.srf)
        self.slices_shape = (len(self.pointings), spec.n_slit, self.n_out)
        self.ishape = (len(self.alpha_axis), len(self.beta_axis))
        self.slit_slices, self.slit_weights = [], []
        for s in range(spec.n_slit):                              # get_slit_slices (:122-149): beta trim only
            fovb = _slit_local_fov(spec, s)
            a0, a1, b0, b1 = _to_slices(fovb, self.la, self.lb)
            if (b1 - b0) > self.npix_b:
                if abs(self.lb[b1] - fovb[3]) > abs(self.lb[b0] - fovb[2]):
                    b1 -= 1
                else:
                    b0 += 1
            self.slit_slices.append((a0, a1, b0, b1))
        for s, sl in enumerate(self.slit_slices):                 # get_slit_weights (:152-172)
            w = _fov_weight(_slit_local_fov(spec, s), sl, self.la, self.lb)
            if s > 0 and self.slit_slices[s - 1][3] - 1 != sl[2]:
                w[:, 0] = 1
            if s < self.npix_b - 1:       # the reference compares with npix_slit_beta_width, not n_slit (:167):
                if sl[3] - 1 != self.slit_slices[s + 1][2]:       # slits >= npix_b-1 keep their fractional last column,
                    w[:, -1] = 1                                  # and n_slit < npix_b raises IndexError as there
            self.slit_weights.append(w)
        self.crops = []
        for p in self.pointings:                                  # gridding (:286-307)
            ia = int(np.abs(self.alpha_axis - p[0]).argmin())
            ib = int(np.abs(self.beta_axis - p[1]).argmin())
            self.crops.append((ia - len(self.la) // 2, ia + len(self.la) // 2 + 1,
                               ib - len(self.lb) // 2, ib + len(self.lb) // 2 + 1))

    def _box(self, img, t=False):
        sh = 1 if t else -1
        return sum(np.roll(img, sh * j, axis=-2) for j in range(self.srf))

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        xb = x if self.batched else x[None]
        sf = self.sotf if self.batched else self.sotf[None]
        blurred = idft(dft(xb) * sf, self.ishape)
        out = np.zeros((xb.shape[0],) + self.slices_shape)
        for p, (a0, a1, b0, b1) in enumerate(self.crops):
            ss = self._box(blurred[:, a0:a1, b0:b1])
            for s, (sa0, sa1, sb0, sb1) in enumerate(self.slit_slices):
                sl = ss[:, sa0:sa1, sb0:sb1] * self.slit_weights[s][None]
                out[:, p, s] = np.sum(sl[:, : self.n_out * self.srf: self.srf], axis=2)
        out = out.reshape(xb.shape[0], -1)
        return out if self.batched else out[0]

    def adjoint(self, data):
        data = np.asarray(data, dtype=np.float64)
        L = self.sotf.shape[0] if self.batched else 1
        d = data.reshape((L,) + self.slices_shape)
        g = np.zeros((L,) + self.ishape)
        na, nb = len(self.la), len(self.lb)
        for p, (a0, a1, b0, b1) in enumerate(self.crops):
            local = np.zeros((L, na, nb))
            for s, (sa0, sa1, sb0, sb1) in enumerate(self.slit_slices):
                over = np.repeat(d[:, p, s][:, :, None], self.npix_b, axis=2)
                bts = np.zeros((L, sa1 - sa0, sb1 - sb0))
                bts[:, : self.n_out * self.srf: self.srf, :] = over
                local[:, sa0:sa1, sb0:sb1] += bts * self.slit_weights[s][None]
            g[:, a0:a1, b0:b1] += self._box(local, t=True)
        sf = self.sotf if self.batched else self.sotf[None]
        out = idft(dft(g) * sf.conj(), self.ishape)
        return out if self.batched else out[0]


    def data_to_img(self, data):
        """``MRSBlurred.data_to_img`` (surfh/Models/spectro_blind_rectangle.py:240-283): the slit data spread evenly over the
        slit's beta columns, put back on the local grid, box-summed, thresholded (< 1 -> 0) with the reference's two column
        patches (local columns 5 <- 6 and 153 <- 152: the local grid must have at least 154 columns), placed in the image
        at every pointing.  Returns (mean over the pointings that cover a pixel, sum over pointings).  The reference leaves the
        mean of uncovered pixels uninitialised (``np.divide(..., where=...)`` without ``out``); 0 here."""
        if self.batched:
            raise ValueError("data_to_img is defined for a single image")
        d = np.asarray(data, dtype=np.float64).reshape(self.slices_shape)
        na, nb = len(self.la), len(self.lb)
        cum = np.zeros((len(self.crops),) + self.ishape)
        for p, (a0, a1, b0, b1) in enumerate(self.crops):
            local = np.zeros((na, nb))
            for s, (sa0, sa1, sb0, sb1) in enumerate(self.slit_slices):
                over = np.repeat(d[p, s][:, None], self.npix_b, axis=1) / self.npix_b
                bts = np.zeros((sa1 - sa0, sb1 - sb0))
                bts[: self.n_out * self.srf: self.srf, :] = over
                local[sa0:sa1, sb0:sb1] += bts * self.slit_weights[s]
            st = self._box(local, t=True)
            st[st < 1] = 0
            st[:, 5] = st[:, 6]
            st[:, 153] = st[:, 152]
            cum[p, a0:a1, b0:b1] = st
        valid = np.sum(cum != 0, axis=0)
        total = np.sum(cum, axis=0)
        return np.divide(total, valid, out=np.zeros(self.ishape), where=valid != 0), total


# ----------------------------------------------------------------------------
# Fourier-domain fused W.C.T operator: Model_WCT (surfh/Models/mixing.py:131-272, di = dj = 1)
# ----------------------------------------------------------------------------
class WCTOracle:
    def __init__(self, psfs_monoch, L_specs, shape_target, L_pce):
        self.shape = tuple(shape_target)
        self.specs = np.asarray(L_specs, dtype=np.float64)
        # H_spec_freq[t, l] = ir2fr(psf[l] pce[l] spec[t, l])   (make_H_spec_freq_sum2, :23-62; kernel_for_sum and
        # rdft2(decal) are identically 1 for di = dj = 1)
        self.otf = ir2fr(np.asarray(psfs_monoch) * np.asarray(L_pce)[:, None, None], self.shape)
        self.ishape = (self.specs.shape[0],) + self.shape
        self.oshape = (self.specs.shape[1],) + self.shape

    def forward(self, x):                                             # :232-245
        xf = dft(np.asarray(x, dtype=np.float64))
        return idft(np.einsum("tl,lij,tij->lij", self.specs, self.otf, xf), self.shape)

    def adjoint(self, y):                                             # :247-268
        yf = dft(np.asarray(y, dtype=np.float64))
        return idft(np.einsum("tl,lij,lij->tij", self.specs, self.otf.conj(), yf), self.shape)

    def fwadj(self, x):                                               # :270-272 via the explicit Hessian (:177-212)
        hth = np.einsum("tl,ul,lij->tuij", self.specs, self.specs, np.abs(self.otf) ** 2)
        return idft(np.einsum("tuij,uij->tij", hth, dft(np.asarray(x, dtype=np.float64))), self.shape)

    def expsol(self, data, mu_reg, gradient="separated"):
        """QuadCriterion3.run_expsol (surfh/ToolsDir/fusion_mixing.py:309-342,348-438): the minimiser of
        |y - H x|^2 + sum_t mu_t |D x_t|^2 in closed form, one T x T solve per frequency:
        (HtH(f) + diag(mu_t |D(f)|^2)) x(f) = (H^T y)(f)."""
        T = self.specs.shape[0]
        mu = np.ones(T) * mu_reg if np.isscalar(mu_reg) else np.asarray(mu_reg, dtype=np.float64)
        hth = np.einsum("tl,ul,lij->ijtu", self.specs, self.specs, np.abs(self.otf) ** 2)
        hth = hth + reg_freq(self.shape, gradient)[:, :, None, None] * np.diag(mu)[None, None]
        b = dft(self.adjoint(data))                                   # [T, Na, Nb/2+1]
        xf = np.linalg.solve(hth, np.moveaxis(b, 0, -1)[..., None])[..., 0]
        return idft(np.moveaxis(xf, -1, 0), self.shape)


def reg_freq(shape, gradient="separated"):
    """|D(f)|^2 on the half spectrum [Na, Nb/2+1] (fusion_mixing.py:364-395): "separated" = |D_row|^2 + |D_col|^2
    of the first-difference kernels [-1, 1]; "joint" = |ir2fr(laplacian(2))|^2 (udft's 3x3 Laplacian, restated)."""
    if gradient == "separated":
        dr = ir2fr(np.array([-1.0, 1.0])[:, None], shape)
        dc = ir2fr(np.array([-1.0, 1.0])[None, :], shape)
        return np.abs(dr) ** 2 + np.abs(dc) ** 2
    if gradient == "joint":
        lap = np.array([[0.0, -1.0, 0.0], [-1.0, 4.0, -1.0], [0.0, -1.0, 0.0]])
        return np.abs(ir2fr(lap, shape)) ** 2
    raise ValueError(gradient)


# ----------------------------------------------------------------------------
# Masked linear mixing model (surfh/Models/mixing.py:276-337 + cythons_files.pyx:370-463), float32 like the reference
# ----------------------------------------------------------------------------
class MixingSTOracle:
    def __init__(self, templates, shape, selection_arr, fast_selection_arr):
        self.tpl = np.asarray(templates, dtype=np.float32)
        self.shape = tuple(shape)                                   # (Na, Nb)
        self.vox = np.asarray(fast_selection_arr).reshape(-1, 3)
        L = self.tpl.shape[1]
        S = np.ones((L,) + self.shape, dtype=np.float32)
        S[selection_arr] = 0
        # c_precompute_TST (:374-392)
        self.TST = np.einsum("pl,ml,lij->mpij", self.tpl.astype(np.float64), self.tpl.astype(np.float64), S.astype(np.float64))

    def forward(self, maps):                                        # c_fast_forward_TST (:400-417)
        maps = np.asarray(maps, dtype=np.float32)
        cube = np.zeros((self.tpl.shape[1],) + self.shape, dtype=np.float64)
        l, i, j = self.vox[:, 0], self.vox[:, 1], self.vox[:, 2]
        np.add.at(cube, (l, i, j), np.einsum("mv,mv->v", maps[:, i, j].astype(np.float64), self.tpl[:, l].astype(np.float64)))
        return cube

    def adjoint(self, cube):                                        # c_fast_adjoint_TST (:447-463)
        cube = np.asarray(cube, dtype=np.float32)
        maps = np.zeros((self.tpl.shape[0],) + self.shape, dtype=np.float64)
        l, i, j = self.vox[:, 0], self.vox[:, 1], self.vox[:, 2]
        for m in range(self.tpl.shape[0]):
            np.add.at(maps[m], (i, j), cube[l, i, j].astype(np.float64) * self.tpl[m, l])
        return maps

    def fwadj(self, maps):                                          # mixing.py:316-317
        return np.sum(self.TST * np.asarray(maps, dtype=np.float64)[np.newaxis], axis=1)
