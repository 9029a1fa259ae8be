"""``Model_WCT``: the reference's Fourier-domain fused "W.C.T" operator
(surfh/Models/mixing.py:131-272, with its decimation di = dj = 1) on the HIP library.

    forward : cube[l] = irfft2( sum_t H[t,l] rfft2(maps[t]) ),  H[t,l] = spec[t,l] pce[l] ir2fr(psf[l])
    adjoint : maps[t] = irfft2( sum_l conj(H[t,l]) rfft2(cube[l]) )
    fwadj   : A^T A through the per-frequency T x T Hessian  sum_l spec[t,l] spec[t',l] |pce[l] otf[l]|^2
    expsol  : (A^T A + diag(mu_t) D^T D)^-1 A^T y, one T x T solve per frequency -- ``QuadCriterion3`` below
              (surfh/ToolsDir/fusion_mixing.py:261-438)

``H`` is never materialised (the reference holds a [T, L, N, N/2+1] complex128 array, mixing.py:40):
the products are formed on the fly from the OTF and the spectra.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .linop import LinOp
from .synth import ir2fr


class Model_WCT(LinOp):
    def __init__(self, psfs_monoch, L_specs, shape_target, L_pce, *, device: int = 0):
        psfs_monoch = np.asarray(psfs_monoch, dtype=np.float64)
        L_specs = np.ascontiguousarray(L_specs, dtype=np.float64)
        L_pce = np.asarray(L_pce, dtype=np.float64)
        assert psfs_monoch.shape[1] <= shape_target[0] and psfs_monoch.shape[2] <= shape_target[1]   # mixing.py:135-136
        self.di = self.dj = 1
        self.shape_target = tuple(int(v) for v in shape_target)
        self.n_spec, self.n_lamb = L_specs.shape
        sotf = np.ascontiguousarray(ir2fr(psfs_monoch * L_pce[:, None, None], self.shape_target), dtype=np.complex128)
        super().__init__(ishape=(self.n_spec,) + self.shape_target, oshape=(self.n_lamb,) + self.shape_target)
        cfg = _lib.Config()
        cfg.n_alpha, cfg.n_beta, cfg.n_lambda, cfg.n_templates = self.shape_target[0], self.shape_target[1], self.n_lamb, self.n_spec
        cfg.templates = _lib.dptr(L_specs)
        cfg.sotf = sotf.view(np.float64).ctypes.data_as(_lib.c_double_p)
        cfg.n_channels, cfg.channels = 0, None
        cfg.device, cfg.stream, cfg.split_k_forward = device, None, 0
        L = _lib.load()
        plan = C.c_void_p()
        _lib.check(L.surfh_plan_create(C.byref(cfg), C.byref(plan)), ValueError)
        self._L, self._plan = L, plan

    def close(self):
        if getattr(self, "_plan", None):
            self._L.surfh_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, fn, x, shape_in, shape_out):
        assert tuple(np.shape(x)) == tuple(shape_in)          # mixing.py:233,248,271
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
        out = np.empty(int(np.prod(shape_out)), dtype=np.float32)
        _lib.check(fn(self._plan, _lib.fptr(a), _lib.fptr(out)))
        return out.astype(np.float64).reshape(shape_out)

    def forward(self, x):
        return self._call(self._L.surfh_wct_forward, x, self.ishape, self.oshape)

    def adjoint(self, y):
        return self._call(self._L.surfh_wct_adjoint, y, self.oshape, self.ishape)

    def fwadj(self, x):
        return self._call(self._L.surfh_wct_fwadj, x, self.ishape, self.ishape)

    def expsol(self, data, L_mu, reg_freq):
        """(H^T H + diag(L_mu) D^T D)^-1 H^T data with |D(f)|^2 = ``reg_freq`` on the half spectrum [Na, Nb/2+1]."""
        assert tuple(np.shape(data)) == tuple(self.oshape)
        mu = np.ascontiguousarray(L_mu, dtype=np.float64)
        reg = np.ascontiguousarray(reg_freq, dtype=np.float64)
        if mu.shape != (self.n_spec,) or reg.shape != (self.shape_target[0], self.shape_target[1] // 2 + 1):
            raise ValueError("L_mu must have one entry per map and reg_freq the half-spectrum shape")
        a = np.ascontiguousarray(np.asarray(data, dtype=np.float32).reshape(-1))
        out = np.empty(int(np.prod(self.ishape)), dtype=np.float32)
        _lib.check(self._L.surfh_wct_expsol(self._plan, _lib.fptr(a), _lib.dptr(mu), _lib.dptr(reg), _lib.fptr(out)),
                   np.linalg.LinAlgError)
        return out.astype(np.float64).reshape(self.ishape)


def regularisation_freq(shape_target, gradient="separated"):
    """|D(f)|^2 of the prior on the half spectrum (Regul_Fusion_Model3, fusion_mixing.py:364-395): "separated" =
    first differences along rows and columns (kernels [-1, 1]), "joint" = udft's 3x3 Laplacian."""
    if gradient == "separated":
        d_row = ir2fr(np.array([-1.0, 1.0])[:, None], shape_target)
        d_col = ir2fr(np.array([-1.0, 1.0])[None, :], shape_target)
        return np.abs(d_row) ** 2 + np.abs(d_col) ** 2
    if gradient == "joint":
        lap = np.array([[0.0, -1.0, 0.0], [-1.0, 4.0, -1.0], [0.0, -1.0, 0.0]])
        return np.abs(ir2fr(lap, shape_target)) ** 2
    raise ValueError(f"gradient must be 'separated' or 'joint', not {gradient!r}")


class QuadCriterion3:
    """Closed-form regularised least squares for ``Model_WCT`` (surfh/ToolsDir/fusion_mixing.py:261-342): same
    constructor and ``run_expsol()``; ``mu_reg`` is one hyper-parameter or one per abundance map."""

    def __init__(self, data, model, mu_reg, printing=False, gradient="separated"):
        self.data, self.model = data, model
        self.n_spec = model.n_spec
        assert isinstance(mu_reg, (float, int, list, np.ndarray))
        self.mu_reg = mu_reg
        if isinstance(mu_reg, (list, np.ndarray)):
            assert len(mu_reg) == self.n_spec
            self.L_mu = np.array(mu_reg, dtype=np.float64)
        else:
            self.L_mu = np.ones(self.n_spec) * mu_reg          # same mu for all maps
        self.shape_of_output = (self.n_spec,) + tuple(model.shape_target)
        self.printing, self.gradient = printing, gradient
        self._reg = regularisation_freq(model.shape_target, gradient)

    def run_expsol(self):
        import time
        t1 = time.time()
        res = self.model.expsol(self.data, self.L_mu, self._reg)
        if self.printing:
            print("Total time needed for expsol = {} sec.".format(round(time.time() - t1, 3)))
        return res


class MixingST(LinOp):
    """Masked linear mixing model (surfh/Models/mixing.py:276-337): the cube exists only on the voxels of
    ``fast_selection_arr`` (``[n, 3]`` rows ``(lambda, i, j)``, e.g. ``np.array(np.where(y_cube > 1e-5)).T``);
    ``selection_arr`` is the complementary NumPy index (``np.where(y_cube < 1e-5)``) whose voxels are zeroed in the
    mask ``S`` behind ``fwadj``'s ``TST``.  Same constructor as the reference; float32 arithmetic like its Cython
    kernels (cythons_files.pyx:370-463)."""

    def __init__(self, templates, alpha_axis, beta_axis, wavel_axis, selection_arr, fast_selection_arr,
                 dtype=np.float64, *, device: int = 0):
        self.templates = np.ascontiguousarray(templates, dtype=np.float64)
        self.alpha_axis, self.beta_axis, self.wavel_axis = alpha_axis, beta_axis, wavel_axis
        self.selection_arr, self.fast_selection_arr = selection_arr, fast_selection_arr
        T, L, na, nb = self.templates.shape[0], len(wavel_axis), len(alpha_axis), len(beta_axis)
        if self.templates.shape[1] != L:
            raise ValueError("templates must be [n_maps, len(wavel_axis)]")
        super().__init__((T, na, nb), (L, na, nb), "MixingModelST", dtype)
        vox = np.ascontiguousarray(np.asarray(fast_selection_arr).reshape(-1, 3), dtype=np.int32)
        S = None
        if selection_arr is not None:                              # fast_precompute_TST, mixing.py:319-327
            S = np.ones((L, na, nb), dtype=np.float32)
            S[selection_arr] = 0
        lib = _lib.load()
        h = C.c_void_p()
        rc = lib.surfh_tst_create(na, nb, L, T, _lib.dptr(self.templates), vox.ctypes.data_as(_lib.c_int32_p), vox.shape[0],
                                  _lib.fptr(S) if S is not None else None, device, C.byref(h))
        if rc:
            raise ValueError(lib.surfh_tst_last_error().decode())
        self._L, self._h = lib, h

    def close(self):
        if getattr(self, "_h", None):
            self._L.surfh_tst_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _run(self, fn, x, shape_in, shape_out):
        if tuple(np.shape(x)) != tuple(shape_in):
            raise ValueError(f"expected shape {tuple(shape_in)}, got {tuple(np.shape(x))}")
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
        out = np.empty(int(np.prod(shape_out)), dtype=np.float32)
        if fn(self._h, _lib.fptr(a), _lib.fptr(out)):
            raise RuntimeError(self._L.surfh_tst_last_error().decode())
        return out.reshape(shape_out)

    def forward(self, maps):
        return self._run(self._L.surfh_tst_forward, maps, self.ishape, self.oshape)

    def adjoint(self, cube):
        return self._run(self._L.surfh_tst_adjoint, cube, self.oshape, self.ishape)

    def fwadj(self, maps):
        return self._run(self._L.surfh_tst_fwadj, maps, self.ishape, self.ishape)

    def mapsToCube(self, maps):
        """Unmasked LMM (mixing.py:330-334)."""
        return np.sum(np.expand_dims(np.asarray(maps), 1) * self.templates[..., np.newaxis, np.newaxis], axis=0)


class ShardedWCT:
    """``Model_WCT`` sharded over wavelength: rank r owns a contiguous slice of the planes.
    ``forward`` returns the rank's slice of the cube (planes are independent, no exchange); ``adjoint`` and ``fwadj`` sum
    over wavelength, so their per-rank partial maps are all-reduced through ``torch.distributed`` -- one collective of
    [T, Na, Nb] per application, the same exchange as the fusion CG.  ``model_factory(psfs, specs, shape, pce)`` replaces
    the HIP operator (CPU rehearsal with the checker)."""

    def __init__(self, psfs_monoch, L_specs, shape_target, L_pce, rank: int = 0, world: int = 1, *, device: int = 0,
                 model_factory=None):
        L = np.asarray(L_specs).shape[1]
        self.rank, self.world = rank, world
        self.lo, self.hi = (L * rank) // world, (L * (rank + 1)) // world
        if self.hi <= self.lo:
            raise ValueError(f"rank {rank} of {world} owns no plane of {L}")
        sl = slice(self.lo, self.hi)
        mk = model_factory or (lambda a, b, c, d: Model_WCT(a, b, c, d, device=device))
        self.model = mk(np.asarray(psfs_monoch)[sl], np.asarray(L_specs)[:, sl], shape_target, np.asarray(L_pce)[sl])
        self.ishape = tuple(self.model.ishape)
        self.oshape = (self.hi - self.lo,) + tuple(shape_target)

    def _allreduce(self, a):
        if self.world == 1:
            return a
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        dist.all_reduce(t)
        return t.numpy()

    def forward(self, x):
        return self.model.forward(x)                 # this rank's planes [lo, hi)

    def adjoint(self, y_local):
        return self._allreduce(self.model.adjoint(y_local))

    def fwadj(self, x):
        return self._allreduce(self.model.fwadj(x))

    def close(self):
        if hasattr(self.model, "close"):
            self.model.close()
