"""Instrument geometry for the MRS operator: the host-side mirror of the reference's
``surfh.Models.instru`` public API (Coord, CoordList, FOV, LocalFOV, SpectralBlur, IFU,
get_srf), so that code written against the reference constructs the same objects.

Only what the forward/adjoint path needs is implemented (setup-time, NumPy, float64).
The numerical conventions that matter for bit-exact tables are kept and cited
(reference paths relative to /root/reference):

* ``Coord.pix`` rounds with Python's banker's ``round``            (instru.py:143-145)
* ``FOV.local_coords`` floor/ceil axis construction                (instru.py:283-304)
* ``LocalFOV.beta_start/beta_end`` rounded to 9 decimals           (instru.py:428-434)
* ``IFU.wslice`` stop index excludes the last in-range plane       (instru.py:649-658)
* ``SpectralBlur.psfs``: ``np.sinc(np.pi * z)``, margin-normalised  (instru.py:499-572)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import ceil, floor
from typing import List, Tuple

import numpy as np

__all__ = ["rotmatrix", "get_srf", "Coord", "CoordList", "FOV", "LocalFOV", "SpectralBlur", "IFU"]


def rotmatrix(degree: float) -> np.ndarray:
    t = np.radians(degree)
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s], [s, c]])


def get_srf(det_pix_size_list, step: float) -> List[int]:
    """Super-resolution factor per channel: det_pix_size // step (instru.py:67-84)."""
    return [int(d // step) for d in det_pix_size_list]


@dataclass
class Coord:
    alpha: float
    beta: float

    @classmethod
    def from_array(cls, arr):
        return cls(arr[0], arr[1])

    def _chk(self, o):
        if not isinstance(o, Coord):
            raise ValueError("`coord` must be a `Coord`")

    def __add__(self, o):
        self._chk(o)
        return Coord(self.alpha + o.alpha, self.beta + o.beta)

    def __sub__(self, o):
        self._chk(o)
        return Coord(self.alpha - o.alpha, self.beta - o.beta)

    def __iadd__(self, o):
        self._chk(o)
        self.alpha += o.alpha
        self.beta += o.beta
        return self

    def __isub__(self, o):
        self._chk(o)
        self.alpha -= o.alpha
        self.beta -= o.beta
        return self

    def rotate(self, degree: float) -> "Coord":
        v = rotmatrix(-degree) @ np.array([[self.alpha], [self.beta]], dtype=np.float32)
        return Coord(float(v[0, 0]), float(v[1, 0]))

    def pix(self, step: float) -> "Coord":
        return Coord(round(self.alpha / step) * step, round(self.beta / step) * step)

    def __array__(self, dtype=None, copy=None):
        return np.array([self.alpha, self.beta]).astype(np.float32 if dtype is None else dtype).reshape((2, 1))


class CoordList(list):
    @classmethod
    def from_array(cls, arr):
        return cls(Coord.from_array(a) for a in arr)

    def pix(self, step: float) -> "CoordList":
        return CoordList(c.pix(step) for c in self)

    alpha_min = property(lambda self: min(c.alpha for c in self))
    alpha_max = property(lambda self: max(c.alpha for c in self))
    beta_min = property(lambda self: min(c.beta for c in self))
    beta_max = property(lambda self: max(c.beta for c in self))
    alpha_mean = property(lambda self: (self.alpha_max + self.alpha_min) / 2)
    beta_mean = property(lambda self: (self.beta_max + self.beta_min) / 2)
    alpha_box = property(lambda self: self.alpha_max - self.alpha_min)
    beta_box = property(lambda self: self.beta_max - self.beta_min)
    box = property(lambda self: (self.alpha_box, self.beta_box))


def _grid(a, b):
    return np.tile(a.reshape((-1, 1)), [1, len(b)]), np.tile(b.reshape((1, -1)), [len(a), 1])


@dataclass
class FOV:
    """A rotated rectangular field of view (degrees)."""
    alpha_width: float
    beta_width: float
    origin: Coord = field(default_factory=lambda: Coord(0, 0))
    angle: float = 0

    def local_coords(self, step, alpha_margin=0, beta_margin=0) -> Tuple[np.ndarray, np.ndarray]:
        def axis(start, length):
            r0 = int(floor(start / step)) * step
            return np.arange(int(ceil((length + (start - r0)) / step)) + 1) * step + r0

        return (axis(-self.alpha_width / 2 - alpha_margin, self.alpha_width + 2 * alpha_margin),
                axis(-self.beta_width / 2 - beta_margin, self.beta_width + 2 * beta_margin))

    def local2global(self, alpha_coords, beta_coords):
        A, B = _grid(alpha_coords, beta_coords)
        c = rotmatrix(self.angle) @ np.vstack((A.ravel(), B.ravel()))
        return c[0].reshape(A.shape) + self.origin.alpha, c[1].reshape(A.shape) + self.origin.beta

    def global2local(self, alpha_coords, beta_coords):
        A, B = _grid(alpha_coords - self.origin.alpha, beta_coords - self.origin.beta)
        c = rotmatrix(-self.angle) @ np.vstack((A.ravel(), B.ravel()))
        return c[0].reshape(A.shape), c[1].reshape(A.shape)

    def coords(self, step, alpha_margin=0, beta_margin=0):
        return self.local2global(*self.local_coords(step, alpha_margin, beta_margin))

    def rotate(self, degree):
        self.angle += degree

    def shift(self, coord):
        self.origin = self.origin + coord

    def _corner(self, sa, sb):
        return Coord(sa * self.alpha_width / 2, sb * self.beta_width / 2).rotate(self.angle) + self.origin

    lower_left = property(lambda self: self._corner(-1, -1))
    lower_right = property(lambda self: self._corner(1, -1))
    upper_right = property(lambda self: self._corner(1, 1))
    upper_left = property(lambda self: self._corner(-1, 1))

    @property
    def vertices(self):
        return (self.lower_left, self.lower_right, self.upper_right, self.upper_left)

    @property
    def bbox(self):
        v = self.vertices
        return (Coord(min(p.alpha for p in v), min(p.beta for p in v)),
                Coord(max(p.alpha for p in v), max(p.beta for p in v)))

    @property
    def local(self):
        return LocalFOV(self)

    def __add__(self, coord):
        return FOV(self.alpha_width, self.beta_width, self.origin + coord, self.angle)

    def __sub__(self, coord):
        return FOV(self.alpha_width, self.beta_width, self.origin - coord, self.angle)


class LocalFOV(FOV):
    """The FOV in its own frame: centred, no angle."""

    def __init__(self, fov: FOV):
        super().__init__(fov.alpha_width, fov.beta_width, Coord(0, 0), angle=0)

    alpha_start = property(lambda self: self.origin.alpha - self.alpha_width / 2)
    alpha_end = property(lambda self: self.origin.alpha + self.alpha_width / 2)
    beta_start = property(lambda self: round(self.origin.beta - self.beta_width / 2, 9))
    beta_end = property(lambda self: round(self.origin.beta + self.beta_width / 2, 9))

    def to_slices(self, alpha_axis, beta_axis):
        da = alpha_axis[1] - alpha_axis[0]
        db = beta_axis[1] - beta_axis[0]
        return (slice(int(np.flatnonzero(self.alpha_start < alpha_axis + da / 2)[0]),
                      int(np.flatnonzero(alpha_axis - da / 2 < self.alpha_end)[-1]) + 1),
                slice(int(np.flatnonzero(self.beta_start < beta_axis + db / 2)[0]),
                      int(np.flatnonzero(beta_axis - db / 2 < self.beta_end)[-1]) + 1))

    def n_alpha(self, step):
        return int(ceil((self.alpha_width / 2) / step)) - int(floor(-self.alpha_width / 2 / step))

    def n_beta(self, step):
        return int(ceil(self.beta_width / 2 / step)) - int(floor(-self.beta_width / 2 / step))

    def __add__(self, coord):
        out = LocalFOV(self)
        out.origin += coord
        return out

    def __sub__(self, coord):
        out = LocalFOV(self)
        out.origin -= coord
        return out


class SpectralBlur:
    """Grating spectral response R = lambda / d_lambda."""

    def __init__(self, grating_resolution: float):
        self.grating_resolution = grating_resolution
        self._n_margin = 15

    @property
    def grating_len(self) -> float:
        return 2 * 0.44245 / np.pi * self.grating_resolution

    def psfs(self, out_axis, beta, wavelength, scale: float = 1, type: str = "mrs") -> np.ndarray:
        """W[lambda', lambda, beta], each detector sample normalised over the margin-extended sky axis."""
        m = self._n_margin
        wavelength = np.asarray(wavelength)
        dw = min(np.diff(wavelength))
        lo, hi = wavelength.min(), wavelength.max()
        ext = np.concatenate([np.linspace(lo - m * dw, lo - dw, m - 1), wavelength,
                              np.linspace(hi + dw, hi + m * dw, m - 1)]).reshape((1, -1, 1))
        oa = np.asarray(out_axis).reshape((-1, 1, 1))
        bt = np.asarray(beta).reshape((1, 1, -1))
        gl = self.grating_len
        out = np.pi * gl / ext * np.sinc(np.pi * gl * ((oa - scale * bt) / ext - 1)) ** 2
        out /= np.sum(out, axis=1, keepdims=True)
        if type == "dirac":
            out = (out == out.max(axis=1, keepdims=True)).astype(out.dtype)      # one-hot at the peak of every (lambda', beta)
        return out[:, m - 1: -m + 1, :]


@dataclass
class IFU:
    """One integral-field channel: FOV, detector pixel, slits, spectral blur, detector wavelength axis."""
    fov: FOV
    det_pix_size: float
    n_slit: int
    w_blur: SpectralBlur
    pce: np.ndarray
    wavel_axis: np.ndarray
    name: str = "_"

    def __post_init__(self):
        sbw = self.slit_beta_width
        self.slit_shift = [Coord(0, -self.fov.beta_width / 2 + sbw / 2) + Coord(0, k * sbw) for k in range(self.n_slit)]
        self.slit_fov = [FOV(self.fov.alpha_width, sbw, self.fov.origin + sh.rotate(self.fov.angle), self.fov.angle)
                         for sh in self.slit_shift]

    wavel_min = property(lambda self: self.wavel_axis[0])
    wavel_max = property(lambda self: self.wavel_axis[-1])
    wavel_step = property(lambda self: self.wavel_axis[1] - self.wavel_axis[0])
    n_wavel = property(lambda self: len(self.wavel_axis))
    slit_beta_width = property(lambda self: self.fov.beta_width / self.n_slit)

    def wslice(self, wavel_input_axis, margin=0) -> slice:
        a = wavel_input_axis
        return slice(int(np.flatnonzero(a <= max(self.wavel_min - margin, a.min()))[-1]),
                     int(np.flatnonzero(a >= min(self.wavel_max + margin, a.max()))[0]))

    def get_name_pix(self):
        return self.name if self.name.endswith("pix") else self.name + "_pix"

    def spectral_psf(self, beta, wavel_input_axis, arcsec2micron, type="mrs"):
        return self.w_blur.psfs(self.wavel_axis, beta, wavel_input_axis, arcsec2micron, type)

    def pix(self, step):
        return IFU(FOV(self.fov.alpha_width, self.fov.beta_width, self.fov.origin.pix(step), self.fov.angle),
                   self.det_pix_size, self.n_slit, self.w_blur, self.pce, self.wavel_axis, self.name + "_pix")
