"""Slit geometry and the per-channel tables the HIP library consumes.

``Slicer`` mirrors the reference's ``surfh.Models.slicer.Slicer`` API (slit slices,
beta-edge weights, ``slicing`` / ``slicing_t``); ``ChannelGeometry`` gathers what
``surfh.Models.spectroModelChannel.Channel.__init__`` derives and freezes it into flat
arrays (``tables()``) with exactly the layout of ``surfh_channel_desc`` in
``include/surfh_amd.h``.  Everything here is setup-time NumPy in float64; the integer
tables are bit-exact with the reference (tests/test_host_geometry.py).
"""
from __future__ import annotations

from math import ceil, floor
from typing import Tuple

import numpy as np

from . import instru


class Slicer:
    """Slit windows of one IFU on its local (alpha, beta) grid (slicer.py:14-244)."""

    def __init__(self, instr: instru.IFU, wavelength_axis, alpha_axis, beta_axis,
                 local_alpha_axis, local_beta_axis, srf: int):
        self.instr = instr
        self.wavelength_axis = wavelength_axis
        self.alpha_axis = alpha_axis
        self.beta_axis = beta_axis
        self.local_alpha_axis = local_alpha_axis
        self.local_beta_axis = local_beta_axis
        self.srf = srf
        self.slices_shape = (instr.n_slit, ceil(self.npix_slit_alpha_width / srf))
        self._cache = {}

    @property
    def wslice(self) -> slice:
        return self.instr.wslice(self.wavelength_axis, 0.1)

    @property
    def slit_beta_width(self):
        return self.instr.fov.beta_width / self.instr.n_slit

    @property
    def npix_slit_beta_width(self) -> int:
        return int(ceil(self.slit_beta_width / (self.beta_axis[1] - self.beta_axis[0])))

    @property
    def slit_alpha_width(self):
        return self.instr.fov.alpha_width

    @property
    def npix_slit_alpha_width(self) -> int:
        step = self.local_alpha_axis[1] - self.local_alpha_axis[0]
        half = self.slit_alpha_width / 2 / step
        return int(ceil(half)) - int(floor(-half))

    def slit_local_fov(self, slit_idx: int) -> instru.LocalFOV:
        return self.instr.slit_fov[slit_idx].local + self.instr.slit_shift[slit_idx]

    def get_slit_slices(self, slit_idx: int) -> Tuple[slice, slice]:
        """Window of slit ``slit_idx``; reproduces both trimming rules of slicer.py:118-145."""
        if slit_idx in self._cache:
            return self._cache[slit_idx]
        lf = self.slit_local_fov(slit_idx)
        sa, sb = lf.to_slices(self.local_alpha_axis, self.local_beta_axis)
        nb = self.npix_slit_beta_width
        if sb.stop - sb.start > nb:
            # one column too many: drop the edge column that is further from the slit bound
            far_end = abs(self.local_beta_axis[sb.stop] - lf.beta_end)
            far_start = abs(self.local_beta_axis[sb.start] - lf.beta_start)
            sb = slice(sb.start, sb.stop - 1) if far_end > far_start else slice(sb.start + 1, sb.stop)
        n_out = self.slices_shape[1]
        if n_out % 2 == 0 and n_out < 28:
            na = self.npix_slit_alpha_width
            if sa.stop - sa.start > na:
                sa = slice(sa.start, sa.stop - 1)
            elif sa.stop - sa.start < na:
                sa = slice(sa.start - 2, sa.stop)
        self._cache[slit_idx] = (sa, sb)
        return sa, sb

    def fov_weight(self, fov: instru.LocalFOV, slices, alpha_axis, beta_axis) -> np.ndarray:
        """Fractional coverage of the first/last beta column (slicer.py:187-244)."""
        sa, sb = slices
        db = beta_axis[1] - beta_axis[0]
        sel = beta_axis[sb]
        w = np.ones((sa.stop - sa.start, sb.stop - sb.start))
        if sel[0] - db / 2 < fov.beta_start:
            first = 1 - abs(sel[0] - db / 2 - fov.beta_start) / db
            assert 0 <= first <= 1, f"Weight of first beta observed pixel in slit must be in [0, 1] ({first:.2f})"
            w[:, 0] = first
        if sel[-1] + db / 2 > fov.beta_end:
            last = 1 - abs(sel[-1] + db / 2 - fov.beta_end) / db
            assert 0 <= last <= 1, f"Weight of last beta observed pixel in slit must be in [0, 1] ({last:.2f})"
            w[:, -1] = last
        return w

    def get_slit_weights(self, slit_idx: int, slices) -> np.ndarray:
        w = self.fov_weight(self.slit_local_fov(slit_idx), slices, self.local_alpha_axis, self.local_beta_axis)
        if slit_idx > 0 and self.get_slit_slices(slit_idx - 1)[1].stop - 1 != slices[1].start:
            w[:, 0] = 1       # previous slit does not share this column
        if slit_idx < self.slices_shape[0] - 1 and slices[1].stop - 1 != self.get_slit_slices(slit_idx + 1)[1].start:
            w[:, -1] = 1      # next slit does not share this column
        return w[np.newaxis, ...]

    def slicing(self, gridded_cube, slit_idx: int):
        sl = self.get_slit_slices(slit_idx)
        return gridded_cube[:, sl[0], sl[1]] * self.get_slit_weights(slit_idx, sl)

    def slicing_t(self, slit, slit_idx: int, local_shape):
        out = np.zeros(local_shape)
        sl = self.get_slit_slices(slit_idx)
        out[:, sl[0], sl[1]] = slit * self.get_slit_weights(slit_idx, sl)
        return out

    def get_slit_shape(self):
        sl = self.get_slit_slices(0)
        return (self.wslice.stop - self.wslice.start, sl[0].stop - sl[0].start, sl[1].stop - sl[1].start)

    get_slit_shape_t = get_slit_shape


def find_indices(axis, values):
    """Lower interval index and normalised distance on the ACTUAL axis values, right end closed,
    out-of-range clamped to the edge intervals (cythons_files.pyx:20-154)."""
    axis = np.asarray(axis, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    idx = np.clip(np.searchsorted(axis, v, side="right") - 1, 0, len(axis) - 2)
    return idx.astype(np.int32), (v - axis[idx]) / (axis[idx + 1] - axis[idx])


def nearest_indices(axis, values):
    """Nearest-sample selection written as a degenerate (index, fraction) pair: the fraction is exactly
    0 or 1, so the bilinear machinery (and its exact transpose) applies the nearest-neighbour gather
    of NN_gridding (spectroModelChannel.py:201-212) without a second code path."""
    axis = np.asarray(axis, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    near = np.abs(axis[None, :] - v[:, None]).argmin(axis=1) if v.size * axis.size < 5e7 else \
        np.clip(np.rint((v - axis[0]) / (axis[1] - axis[0])), 0, len(axis) - 1).astype(np.int64)
    lo = np.minimum(near, len(axis) - 2)
    return lo.astype(np.int32), (near - lo).astype(np.float64)


class ChannelGeometry:
    """Host description of one channel (spectroModelChannel.py:27-108), no arithmetic on cubes."""

    def __init__(self, instr: instru.IFU, alpha_axis, beta_axis, wavel_axis, srf: int,
                 pointings: instru.CoordList, step_degree: float, gridding: str = "bilinear",
                 lam_slice=None, psf_type: str = "mrs", box=None, beta_sum: bool = False, full_window: bool = False):
        """``lam_slice = (i, n)`` keeps only the i-th of n contiguous parts of the channel's wavelength window
        (multi-GPU: a band's lambda range shared by n ranks; the partial outputs add up to the band's output).

        The last four arguments select the variants the reference's slice <-> cube projections use
        (spectroModelChannel.py:266-336): ``psf_type='dirac'`` the one-hot spectral selector ``wpsf_dirac``;
        ``box=(len, shift)`` another alpha window per detector sample (see ``surfh_channel_desc.box_len``);
        ``beta_sum`` no spectral blur, the slit's beta columns are summed; ``full_window`` every plane of the cube is
        observed (the real-data cubes live on the detector's wavelength axis)."""
        if gridding not in ("bilinear", "nn", "nn_ref"):
            raise ValueError("gridding must be 'bilinear', 'nn' or 'nn_ref'")
        self.gridding_mode = gridding
        self.psf_type, self.box, self.beta_sum = psf_type, box, beta_sum
        self.raw_instr, self.raw_pointings = instr, pointings
        self.alpha_axis = np.asarray(alpha_axis, dtype=np.float64)
        self.beta_axis = np.asarray(beta_axis, dtype=np.float64)
        self.global_wavelength_axis = np.asarray(wavel_axis, dtype=np.float64)
        self.step_degree = step_degree
        self.srf = srf
        self.instr = instr.pix(step_degree)
        self.pointings = instru.CoordList(pointings).pix(step_degree)
        self.local_alpha_axis, self.local_beta_axis = self.instr.fov.local_coords(
            step_degree, alpha_margin=5 * step_degree, beta_margin=5 * step_degree)
        self.slicer = Slicer(self.instr, self.global_wavelength_axis, self.alpha_axis, self.beta_axis,
                             self.local_alpha_axis, self.local_beta_axis, srf)
        self.wslice = slice(0, len(self.global_wavelength_axis)) if full_window else self.instr.wslice(self.global_wavelength_axis, 0.1)
        self.band_wslice = self.wslice
        self.lam_slice = lam_slice
        if lam_slice is not None:
            ws0, lin = self.wslice.start, self.wslice.stop - self.wslice.start
            if len(lam_slice) == 3:          # ("planes", a, b): explicit plane offsets inside the window
                _, a, b = lam_slice
            else:                            # (i, n): the i-th of n equal parts
                i, n = lam_slice
                if not (0 <= i < n <= lin):
                    raise ValueError(f"bad lam_slice {lam_slice} for a window of {lin} planes")
                a, b = (lin * i) // n, (lin * (i + 1)) // n
            if not (0 <= a < b <= lin):
                raise ValueError(f"bad lam_slice {lam_slice} for a window of {lin} planes")
            self.wslice = slice(ws0 + a, ws0 + b)
        n_out = ceil(self.slicer.npix_slit_alpha_width / srf)
        n_det = (self.wslice.stop - self.wslice.start) if beta_sum else len(self.instr.wavel_axis)
        self.oshape = (len(self.pointings), self.instr.n_slit, n_det, n_out)
        self.slices_shape = (len(self.pointings), self.instr.n_slit, n_out)
        self.local_im_shape = (len(self.local_alpha_axis), len(self.local_beta_axis))
        self.imshape = (len(self.alpha_axis), len(self.beta_axis))
        self.ishape = (len(self.global_wavelength_axis),) + self.imshape
        self.instr_cube_shape = (self.wslice.stop - self.wslice.start,) + self.imshape
        self._wpsf = None

    @property
    def beta_step(self):
        return self.beta_axis[1] - self.beta_axis[0]

    @property
    def wpsf(self) -> np.ndarray:
        """W[lambda', lambda, beta] (spectroModelChannel.py:133-143); beta offsets are passed in
        degrees and scaled by um/arcsec exactly as the reference does."""
        if self._wpsf is None:
            n = self.slicer.npix_slit_beta_width
            b = np.arange(0, n) * self.beta_step
            # always evaluated on the band's full window (its normalisation runs over that axis), then cut
            full = self.instr.spectral_psf(
                b - np.mean(b), self.global_wavelength_axis[self.band_wslice],
                arcsec2micron=self.instr.wavel_step / self.instr.det_pix_size, type=self.psf_type)
            o = self.wslice.start - self.band_wslice.start
            self._wpsf = np.ascontiguousarray(full[:, o: o + (self.wslice.stop - self.wslice.start), :])
        return self._wpsf

    def grid_tables(self, p: int):
        """Bilinear (i0, i1, y0, y1) of the local grid in the cube; raises like the reference's
        ``bounds_error=True`` (cython_2D_interpolation.py:472-478)."""
        ga, gb = (self.instr.fov + self.pointings[p]).local2global(self.local_alpha_axis, self.local_beta_axis)
        for dim, (ax, v) in enumerate(((self.alpha_axis, ga), (self.beta_axis, gb))):
            if not (np.all(ax[0] <= v) and np.all(v <= ax[-1])):
                raise ValueError("One of the requested xi is out of bounds in dimension %d" % dim)
        if self.gridding_mode == "bilinear":
            i0, y0 = find_indices(self.alpha_axis, ga.ravel())
            i1, y1 = find_indices(self.beta_axis, gb.ravel())
        elif self.gridding_mode == "nn":
            i0, y0 = nearest_indices(self.alpha_axis, ga.ravel())
            i1, y1 = nearest_indices(self.beta_axis, gb.ravel())
        else:
            # the reference's index recipe (spectroModelChannel.py:399-407) builds k = i_beta*N + i_alpha and
            # applies it to the C-order cube, i.e. it reads pixel [i_beta, i_alpha]: kept under "nn_ref"
            if len(self.alpha_axis) != len(self.beta_axis):
                raise ValueError("gridding='nn_ref' needs a square cube")
            i0, y0 = nearest_indices(self.beta_axis, gb.ravel())
            i1, y1 = nearest_indices(self.alpha_axis, ga.ravel())
        return i0, i1, y0, y1

    def gridt_tables(self, p: int):
        """Tables of the reference's back-projection: interpolating ``gridding_t``
        (spectroModelChannel.py:180-199), or for the NN modes the nearest local pixel of every cube
        pixel (``NN_gridding_t``, :208-212, :411-415 -- no support mask, as in the reference)."""
        ca, cb = (self.instr.fov + self.pointings[p]).global2local(self.alpha_axis, self.beta_axis)
        la, lb = self.local_alpha_axis, self.local_beta_axis
        if self.gridding_mode == "bilinear":
            i0, y0 = find_indices(la, ca.ravel())
            i1, y1 = find_indices(lb, cb.ravel())
            inside = ~((ca.ravel() < la[0]) | (ca.ravel() > la[-1]) | (cb.ravel() < lb[0]) | (cb.ravel() > lb[-1]))
            return i0, i1, y0, y1, inside.astype(np.uint8)
        if self.gridding_mode == "nn_ref":      # global pixel [i, j] is read as the point (alpha_j, beta_i)
            ca, cb = ca.T.copy(), cb.T.copy()
        i0, y0 = nearest_indices(la, ca.ravel())
        i1, y1 = nearest_indices(lb, cb.ravel())
        return i0, i1, y0, y1, np.ones(ca.size, dtype=np.uint8)

    def tables(self, with_ref: bool = True) -> dict:
        """Flat C-contiguous arrays in the layout of ``surfh_channel_desc``."""
        S = self.instr.n_slit
        slices = [self.slicer.get_slit_slices(s) for s in range(S)]
        nbs = self.slicer.npix_slit_beta_width
        a0, a1 = slices[0][0].start, slices[0][0].stop
        weights = np.empty((S, nbs))
        for s, sl in enumerate(slices):
            if (sl[0].start, sl[0].stop) != (a0, a1) or sl[1].stop - sl[1].start != nbs:
                raise ValueError(f"slit {s}: window {sl} differs from slit 0 / npix_slit_beta_width={nbs}")
            w = self.slicer.get_slit_weights(s, sl)[0]
            if not np.all(w == w[0:1]):
                raise ValueError("slit weights are expected to be constant along alpha")
            weights[s] = w[0]
        P = len(self.pointings)
        g = [self.grid_tables(p) for p in range(P)]
        t = dict(
            wslice_start=int(self.wslice.start), wslice_stop=int(self.wslice.stop), n_pointings=P, n_slit=S,
            n_lambda_out=int(self.oshape[2]), n_alpha_out=int(self.oshape[3]), srf=int(self.srf),
            na=int(self.local_im_shape[0]), nb=int(self.local_im_shape[1]), alpha0=int(a0),
            n_alpha_slit=int(a1 - a0), n_beta_slit=int(nbs),
            slit_beta0=np.ascontiguousarray([sl[1].start for sl in slices], dtype=np.int32),
            slit_weights=np.ascontiguousarray(weights, dtype=np.float64),
            grid_i0=np.ascontiguousarray(np.stack([x[0] for x in g]), dtype=np.int32),
            grid_i1=np.ascontiguousarray(np.stack([x[1] for x in g]), dtype=np.int32),
            grid_y0=np.ascontiguousarray(np.stack([x[2] for x in g]), dtype=np.float64),
            grid_y1=np.ascontiguousarray(np.stack([x[3] for x in g]), dtype=np.float64),
            wpsf=None if self.beta_sum else np.ascontiguousarray(self.wpsf, dtype=np.float64),
            box_len=0 if self.box is None else int(self.box[0]), box_shift=0 if self.box is None else int(self.box[1]),
        )
        if with_ref:
            r = [self.gridt_tables(p) for p in range(P)]
            t.update(
                gt_i0=np.ascontiguousarray(np.stack([x[0] for x in r]), dtype=np.int32),
                gt_i1=np.ascontiguousarray(np.stack([x[1] for x in r]), dtype=np.int32),
                gt_y0=np.ascontiguousarray(np.stack([x[2] for x in r]), dtype=np.float64),
                gt_y1=np.ascontiguousarray(np.stack([x[3] for x in r]), dtype=np.float64),
                gt_inside=np.ascontiguousarray(np.stack([x[4] for x in r]), dtype=np.uint8),
            )
        return t
