"""``spectroSigRLSCT``: the reference's multi-channel multi-observation MRS operator
(surfh/Models/spectroModel.py:39-185) with the same constructor and methods, evaluated by
the HIP library through the C ABI (include/surfh_amd.h).

    y = Sigma R L S C T x      (operator chain documented at spectroModel.py:25-38)

``forward`` / ``adjoint`` take and return NumPy arrays (float64 out, for drop-in use with
a CPU solver); ``*_dev`` variants take device pointers / torch tensors and are
asynchronous on the plan's stream.  ``adjoint`` is the exact transpose of ``forward``
(what CG and the dot-test need); ``adjoint_ref`` reproduces the reference's
interpolating ``gridding_t`` (spectroModelChannel.py:180-199).
"""
from __future__ import annotations

import ctypes as C
from math import ceil
from typing import List, Optional, Sequence

import numpy as np

from . import _lib, instru
from .geometry import ChannelGeometry
from .linop import LinOp


def _ptr(t):
    """Device pointer of a torch tensor (or a raw int)."""
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class Channel(ChannelGeometry):
    """One channel of the model, as ``model.channels[k]`` (surfh.Models.spectroModelChannel.Channel): the geometry
    plus the reference's slice <-> cube projections its drivers use to look at data and to initialise
    (scripts/fusion/*.py).  Each projection is the library's own gather / GEMM / scatter chain on the GPU, run on a
    small single-channel plane-wise plan built on first use with the variant tables of ``ChannelGeometry``."""
    device = 0

    def _aux(self, key, wavel_axis, pointings, **opts):
        cache = self.__dict__.setdefault("_aux_ops", {})
        if key not in cache:
            cache[key] = spectroSigRLSCT(None, None, self.alpha_axis, self.beta_axis, wavel_axis, [self.raw_instr],
                                         self.step_degree, [instru.CoordList(pointings)], device=self.device,
                                         gridding=self.gridding_mode, channel_opts=opts)
        return cache[key]

    def close(self):
        for op in self.__dict__.pop("_aux_ops", {}).values():
            op.close()

    def sliceToCube(self, data):
        """spectroModelChannel.py:266-301: the data of pointing 0 back in the cube -- every detector sample goes to the
        wavelength plane where its spectral response peaks (``wpsf_dirac``), then the reference's adjoint chain
        (slit weights, transposed box sum, interpolating ``gridding_t``).  Returns [len(wavelength_axis), Na, Nb]."""
        if self.lam_slice is not None:
            raise ValueError("sliceToCube is defined on a whole channel, not on a lambda part")
        op = self._aux("s2c", self.global_wavelength_axis, [self.raw_pointings[0]], psf_type="dirac")
        y0 = np.asarray(data, dtype=np.float64).reshape(self.oshape)[0]
        return op.adjoint_ref(y0)

    def realData_cubeToSlice(self, cube):
        """spectroModelChannel.py:303-309: a cube on the detector's wavelength axis -> [n_slit, L, n_alpha_out]:
        gridding at pointing (0, 0), slit windows with their edge weights, plain alpha decimation, sum over beta."""
        cube = np.asarray(cube)
        if cube.shape != (self.oshape[2],) + self.imshape:
            raise ValueError(f"cube shape {cube.shape} != {(self.oshape[2],) + self.imshape}")
        op = self._aux("c2s", np.asarray(self.raw_instr.wavel_axis, dtype=np.float64), [instru.Coord(0, 0)],
                       beta_sum=True, full_window=True, box=(1, 0))
        L, (S, A) = self.oshape[2], (self.oshape[1], self.oshape[3])
        return op.forward(cube).reshape(L, S, A).transpose(1, 0, 2).copy()          # the plan's beta-sum output is [l][s][a]

    def realData_sliceToCube(self, slices, cube_dim):
        """spectroModelChannel.py:311-336: slit values spread evenly over the slit's beta columns, zero-stuffed along
        alpha, correlated with the box kernel (``_otf_sr.conj()``, no ``decalf``) and back-interpolated at pointing (0, 0)."""
        cube_dim = tuple(int(v) for v in cube_dim)
        if cube_dim != (self.oshape[2],) + self.imshape:
            raise ValueError(f"cube_dim {cube_dim} != {(self.oshape[2],) + self.imshape}")
        op = self._aux("s2c_rd", np.asarray(self.raw_instr.wavel_axis, dtype=np.float64), [instru.Coord(0, 0)],
                       beta_sum=True, full_window=True, box=(self.srf, -int((self.srf - 1) / 2)))
        s = np.asarray(slices, dtype=np.float64).reshape(self.oshape[1:]) / self.slicer.npix_slit_beta_width
        return op.adjoint_ref(np.ascontiguousarray(s.transpose(1, 0, 2)))


class spectroSigRLSCT(LinOp):
    def __init__(self, sotf, templates, alpha_axis, beta_axis, wavelength_axis, instrs: List[instru.IFU],
                 step_degree: float, pointings: Sequence[instru.CoordList], *, device: int = 0,
                 channels: Optional[Sequence[int]] = None, with_ref: bool = True, stream: Optional[int] = None,
                 split_k_forward: int = 0, gridding: str = "bilinear", lam_slices=None, channel_opts: Optional[dict] = None,
                 verify: bool = False, exact: int = 0):
        """``exact``: ``surfh_config.exact`` -- 1: all three fp16 products on every K step of the spectral-blur GEMMs, 2: whole
        spectrum in the transform passes, 3: both (the production kernels without their two bounded approximations).
        ``sotf=None`` (plane-wise plans only, ``templates=None``) means no spatial blur; ``channel_opts`` are the
        ``ChannelGeometry`` variants of the slice <-> cube projections (see ``Channel`` below).  ``verify=True`` builds the
        verification plan (include/surfh_amd.h ``surfh_config.verify``): the same operator with every long sum accumulated in
        float64 -- slow; what the strict dot test with zero-mean test vectors runs on."""
        self.sotf = sotf
        self.alpha_axis = np.asarray(alpha_axis, dtype=np.float64)
        self.beta_axis = np.asarray(beta_axis, dtype=np.float64)
        self.wavelength_axis = np.asarray(wavelength_axis, dtype=np.float64)
        self.step_degree = step_degree
        self.templates = None if templates is None else np.ascontiguousarray(templates, dtype=np.float64)
        self.lmm = self.templates is not None
        self.pointings = pointings
        self.instrs = [i.pix(step_degree) for i in instrs]
        self.srfs = instru.get_srf([i.det_pix_size for i in instrs], step_degree * 3600)
        # every channel's geometry is known to every rank; `channels` selects the ones this plan owns
        self.all_channels = [Channel(ins, self.alpha_axis, self.beta_axis, self.wavelength_axis, srf,
                                     pointings[k], step_degree, gridding=gridding,
                                     lam_slice=None if lam_slices is None else lam_slices[k], **(channel_opts or {}))
                             for k, (srf, ins) in enumerate(zip(self.srfs, instrs))]
        for c in self.all_channels:
            c.device = device
        self.owned = list(range(len(instrs))) if channels is None else [int(c) for c in channels]
        self.channels = [self.all_channels[k] for k in self.owned]
        self.list_wslice = [c.wslice for c in self.channels]
        self.instrs_oshape = [c.oshape for c in self.channels]
        self._idx = np.cumsum([0] + [int(np.prod(s)) for s in self.instrs_oshape])
        self.full_idx = np.cumsum([0] + [int(np.prod(c.oshape)) for c in self.all_channels])
        self.imshape = (len(self.alpha_axis), len(self.beta_axis))
        self.cube_shape = (len(self.wavelength_axis),) + self.imshape
        lead = self.templates.shape[0] if self.lmm else len(self.wavelength_axis)
        super().__init__(ishape=(lead,) + self.imshape, oshape=(int(self._idx[-1]),))

        nkb = self.imshape[1] // 2 + 1
        if sotf is None and self.lmm:
            raise ValueError("sotf=None (no spatial blur) is only defined without templates")
        sotf_c = None if sotf is None else np.ascontiguousarray(sotf, dtype=np.complex128)
        if sotf_c is not None and sotf_c.shape != (self.cube_shape[0], self.imshape[0], nkb):
            raise ValueError(f"sotf shape {sotf_c.shape} != {(self.cube_shape[0], self.imshape[0], nkb)}")
        if self.lmm and self.templates.shape[1] != self.cube_shape[0]:
            raise ValueError("templates must be [T, len(wavelength_axis)]")

        L = _lib.load()
        self._keep = []          # keeps the table arrays alive during plan creation
        descs = (_lib.ChannelDesc * len(self.channels))()
        for d, ch in zip(descs, self.channels):
            t = ch.tables(with_ref=with_ref)   # raises ValueError like the reference if the FoV leaves the cube
            self._keep.append(t)
            for k in ("wslice_start", "wslice_stop", "n_pointings", "n_slit", "n_lambda_out", "n_alpha_out", "srf",
                      "na", "nb", "alpha0", "n_alpha_slit", "n_beta_slit"):
                setattr(d, k, t[k])
            d.slit_beta0 = _lib.iptr(t["slit_beta0"])
            d.slit_weights = _lib.dptr(t["slit_weights"])
            d.grid_i0, d.grid_i1 = _lib.iptr(t["grid_i0"]), _lib.iptr(t["grid_i1"])
            d.grid_y0, d.grid_y1 = _lib.dptr(t["grid_y0"]), _lib.dptr(t["grid_y1"])
            d.wpsf = None if t["wpsf"] is None else _lib.dptr(t["wpsf"])
            d.box_len, d.box_shift = t["box_len"], t["box_shift"]
            if with_ref:
                d.gt_i0, d.gt_i1 = _lib.iptr(t["gt_i0"]), _lib.iptr(t["gt_i1"])
                d.gt_y0, d.gt_y1 = _lib.dptr(t["gt_y0"]), _lib.dptr(t["gt_y1"])
                d.gt_inside = _lib.u8ptr(t["gt_inside"])
        cfg = _lib.Config()
        cfg.n_alpha, cfg.n_beta, cfg.n_lambda = self.imshape[0], self.imshape[1], self.cube_shape[0]
        cfg.n_templates = self.templates.shape[0] if self.lmm else 0
        cfg.templates = _lib.dptr(self.templates) if self.lmm else None
        cfg.sotf = None if sotf_c is None else sotf_c.view(np.float64).ctypes.data_as(_lib.c_double_p)
        cfg.n_channels = len(self.channels)
        cfg.channels = descs
        cfg.device = device
        cfg.stream = C.c_void_p(stream) if stream else None
        cfg.split_k_forward = split_k_forward
        cfg.verify = 1 if verify else 0
        cfg.exact = int(exact)
        plan = C.c_void_p()
        _lib.check(L.surfh_plan_create(C.byref(cfg), C.byref(plan)), ValueError)
        self._L, self._plan = L, plan
        self._keep = []
        assert L.surfh_isize(plan) == self.isize and L.surfh_osize(plan) == self.osize
        self.device = device

    # ---- life cycle -----------------------------------------------------------------------
    def close(self):
        for c in getattr(self, "all_channels", []):
            c.close()
        if getattr(self, "_plan", None):
            self._L.surfh_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def alpha_step(self) -> float:
        return self.alpha_axis[1] - self.alpha_axis[0]

    @property
    def beta_step(self) -> float:
        return self.beta_axis[1] - self.beta_axis[0]

    @property
    def stream(self) -> int:
        return int(self._L.surfh_stream(self._plan) or 0)

    # ---- host-array API (reference calling convention) --------------------------------------
    def _host(self, fn, x, nin, shape_out):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
        if a.size != nin:
            raise ValueError(f"input has {a.size} elements, expected {nin}")
        out = np.empty(int(np.prod(shape_out)), dtype=np.float32)
        _lib.check(fn(self._plan, _lib.fptr(a), _lib.fptr(out)))
        return out.astype(np.float64).reshape(shape_out)

    def forward(self, maps):
        return self._host(self._L.surfh_forward, maps, self.isize, self.oshape)

    def adjoint(self, inarray):
        return self._host(self._L.surfh_adjoint, inarray, self.osize, self.ishape)

    def adjoint_ref(self, inarray):
        return self._host(self._L.surfh_adjoint_ref, inarray, self.osize, self.ishape)

    def fwadj(self, x):
        return self._host(self._L.surfh_fwadj, x, self.isize, self.ishape)

    # ---- device-pointer API (asynchronous on self.stream) ----------------------------------
    def forward_dev(self, maps_t, y_t):
        _lib.check(self._L.surfh_forward_dev(self._plan, _ptr(maps_t), _ptr(y_t)))

    def adjoint_dev(self, y_t, maps_t):
        _lib.check(self._L.surfh_adjoint_dev(self._plan, _ptr(y_t), _ptr(maps_t)))

    def adjoint_ref_dev(self, y_t, maps_t):
        _lib.check(self._L.surfh_adjoint_ref_dev(self._plan, _ptr(y_t), _ptr(maps_t)))

    def normal_dev(self, d_t, q_t, mu: float = 1.0):
        _lib.check(self._L.surfh_normal_dev(self._plan, _ptr(d_t), _ptr(q_t), float(mu)))

    # ---- the same with the solver's vectors in the Fourier domain of the maps (include/surfh_amd.h: surfh_normal_spec_dev) ----
    def spec_supported(self) -> bool:
        return bool(self._L.surfh_spec_supported(self._plan))

    @property
    def spec_size(self) -> int:
        return int(self._L.surfh_spec_size(self._plan))

    def to_spec_dev(self, x_t, xt_t):
        _lib.check(self._L.surfh_to_spec_dev(self._plan, _ptr(x_t), _ptr(xt_t)))

    def from_spec_dev(self, xt_t, x_t):
        _lib.check(self._L.surfh_from_spec_dev(self._plan, _ptr(xt_t), _ptr(x_t)))

    def forward_spec_dev(self, dt_t, y_t):
        _lib.check(self._L.surfh_forward_spec_dev(self._plan, _ptr(dt_t), _ptr(y_t)))

    def adjoint_spec_dev(self, y_t, qt_t, mu: float = 1.0, dt_t=None, mu_reg: float = 0.0):
        _lib.check(self._L.surfh_adjoint_spec_dev(self._plan, _ptr(y_t), _ptr(qt_t), float(mu), _ptr(dt_t) if dt_t is not None else None,
                                                  float(mu_reg)))

    def normal_spec_dev(self, dt_t, qt_t, mu: float = 1.0, mu_reg: float = 0.0):
        _lib.check(self._L.surfh_normal_spec_dev(self._plan, _ptr(dt_t), _ptr(qt_t), float(mu), float(mu_reg)))

    def prior_spec_add_dev(self, dt_t, qt_t, mu_reg: float):
        _lib.check(self._L.surfh_prior_spec_add_dev(self._plan, _ptr(dt_t), _ptr(qt_t), float(mu_reg)))

    def prior_add_dev(self, d_t, q_t, mu_reg: float):
        _lib.check(self._L.surfh_prior_add_dev(self._plan, _ptr(d_t), _ptr(q_t), float(mu_reg)))

    def dot_dev(self, a_t, b_t, n: int) -> float:
        out = C.c_double()
        _lib.check(self._L.surfh_dot_dev(self._plan, _ptr(a_t), _ptr(b_t), int(n), C.byref(out)))
        return out.value

    def cg_step_dev(self, x_t, r_t, d_t, q_t, n: int, rr: float) -> float:
        out = C.c_double()
        _lib.check(self._L.surfh_cg_step_dev(self._plan, _ptr(x_t), _ptr(r_t), _ptr(d_t), _ptr(q_t), int(n), float(rr),
                                             C.byref(out)))
        return out.value

    def cg_iter_dev(self, x_t, r_t, d_t, q_t, n: int, rr: float) -> float:
        """cg_step_dev + cg_dir_dev with one host synchronisation; returns the new r.r."""
        out = C.c_double()
        _lib.check(self._L.surfh_cg_iter_dev(self._plan, _ptr(x_t), _ptr(r_t), _ptr(d_t), _ptr(q_t), int(n), float(rr), C.byref(out)))
        return out.value

    def cg_dir_dev(self, d_t, r_t, n: int, beta: float):
        _lib.check(self._L.surfh_cg_dir_dev(self._plan, _ptr(d_t), _ptr(r_t), int(n), float(beta)))

    # device-resident CG scalars: nothing below synchronises with the host (include/surfh_amd.h)
    def cg_begin_dev(self, r_t, n: int):
        _lib.check(self._L.surfh_cg_begin_dev(self._plan, _ptr(r_t), int(n)))

    def cg_iter_nosync_dev(self, x_t, r_t, d_t, q_t, n: int):
        _lib.check(self._L.surfh_cg_iter_nosync_dev(self._plan, _ptr(x_t), _ptr(r_t), _ptr(d_t), _ptr(q_t), int(n)))

    def cg_xupdate_nosync_dev(self, x_t, d_t, q_t, n: int):
        _lib.check(self._L.surfh_cg_xupdate_nosync_dev(self._plan, _ptr(x_t), _ptr(d_t), _ptr(q_t), int(n)))

    def cg_refresh_nosync_dev(self, r_t, b_t, q_t, d_t, n: int):
        _lib.check(self._L.surfh_cg_refresh_nosync_dev(self._plan, _ptr(r_t), _ptr(b_t), _ptr(q_t), _ptr(d_t), int(n)))

    def cg_trace(self, cap: int = 1 << 16) -> np.ndarray:
        """r.r of every iterate since ``cg_begin_dev`` (synchronises the plan's stream)."""
        out = np.zeros(cap, dtype=np.float64)
        n = self._L.surfh_cg_trace(self._plan, _lib.dptr(out), int(cap))
        if n < 0:
            raise RuntimeError("surfh_cg_trace failed")
        return out[:n].copy()

    def residual_dev(self, r_t, b_t, q_t, n: int):
        _lib.check(self._L.surfh_residual_dev(self._plan, _ptr(r_t), _ptr(b_t), _ptr(q_t), int(n)))

    def set_prior(self, gradient: str = "separated"):
        """The regulariser of ``cg`` / ``mmmg`` / ``prior_add_dev``: "separated" (NpDiff_r / NpDiff_c, the default) or "joint"
        (the Laplacian of Difference_Operator_Joint) -- ``QuadCriterion_MRS(gradient=...)``, fusion_CT.py:98-106."""
        if gradient not in ("separated", "joint"):
            raise ValueError(f"gradient must be 'separated' or 'joint', not {gradient!r}")
        _lib.check(self._L.surfh_set_prior(self._plan, 1 if gradient == "joint" else 0))
        self._prior = gradient

    def get_prior(self) -> str:
        """The regulariser ``set_prior`` selected last ("separated" on a new model)."""
        return getattr(self, "_prior", "separated")

    # ---- solver on one GPU ------------------------------------------------------------------
    def cg(self, data, mu=1.0, mu_reg=0.0, x0=None, max_iter=10, tol=1e-12, refresh=50, callback=None):
        """Device-resident linear CG (qmm.lcg restated).  ``callback(it, grad_norm, x)`` -- the per-iteration callback
        of ``qmm.lcg`` (fusion_CT.py:194-225) -- receives the 1-based iteration, the grad_norm trace so far and the
        current iterate ``[T,Na,Nb]``; it may call ``forward`` / ``adjoint`` on this model; a truthy return stops."""
        return self._solve(self._L.surfh_cg_cb, data, mu, mu_reg, x0, max_iter, tol, refresh, callback)

    def mmmg(self, data, mu=1.0, mu_reg=0.0, x0=None, max_iter=10, tol=1e-12, refresh=50, callback=None):
        """Device-resident 3MG (qmm.mmmg restated for quadratic objectives; the reference's ``method='mmmg'``,
        fusion_CT.py:194-198).  Same arguments as ``cg``; ``grad_norm`` holds |grad| (not squared) of every iterate."""
        return self._solve(self._L.surfh_mmmg, data, mu, mu_reg, x0, max_iter, tol, refresh, callback)

    def _solve(self, fn, data, mu, mu_reg, x0, max_iter, tol, refresh, callback):
        y = np.ascontiguousarray(np.asarray(data, dtype=np.float32).reshape(-1))
        if y.size != self.osize:
            raise ValueError("data size mismatch")
        x0a = None if x0 is None else np.ascontiguousarray(np.asarray(x0, dtype=np.float32).reshape(-1))
        x = np.empty(self.isize, dtype=np.float32)
        gn = np.zeros(max_iter + 1, dtype=np.float64)
        nit = C.c_int32()
        err = []

        def tramp(_user, it, gptr, xptr):
            try:
                g = np.ctypeslib.as_array(gptr, shape=(it + 1,)).copy()
                xi = np.ctypeslib.as_array(xptr, shape=(self.isize,)).astype(np.float64).reshape(self.ishape)
                return 1 if callback(it, g, xi) else 0
            except BaseException as e:          # never unwind through the C frame
                err.append(e)
                return 1

        cb = _lib.CG_CALLBACK(tramp) if callback is not None else _lib.CG_CALLBACK()
        _lib.check(fn(self._plan, _lib.fptr(y), float(mu), float(mu_reg),
                                       _lib.fptr(x0a) if x0a is not None else None, int(max_iter), float(tol),
                                       int(refresh), _lib.fptr(x), _lib.dptr(gn), C.byref(nit), cb, None))
        if err:
            raise err[0]
        return x.astype(np.float64).reshape(self.ishape), gn[: nit.value + 1].copy(), nit.value

    # ---- helpers the reference's drivers call -----------------------------------------------
    def cubeTomaps(self, cube):
        """lmm_cube2maps (spectroModel.py:187-188, jax_utils.py:18-26) on the device."""
        cube = np.ascontiguousarray(np.asarray(cube, dtype=np.float32))
        tpl = np.ascontiguousarray(self.templates, dtype=np.float64)
        if cube.shape != (tpl.shape[1], self.ishape[-2], self.ishape[-1]):
            raise ValueError(f"cube shape {cube.shape} != {(tpl.shape[1], self.ishape[-2], self.ishape[-1])}")
        out = np.empty((tpl.shape[0],) + cube.shape[1:], dtype=np.float32)
        _lib.check(self._L.surfh_cube_to_maps(self._plan, _lib.dptr(tpl), tpl.shape[0], tpl.shape[1], _lib.fptr(cube), _lib.fptr(out)))
        return out.astype(np.float64)

    def mapsToCube(self, maps):
        """lmm_maps2cube (spectroModel.py:190-198, jax_utils.py:10-16) on the device."""
        tpl = np.ascontiguousarray(self.templates, dtype=np.float64)
        maps = np.ascontiguousarray(np.asarray(maps, dtype=np.float32).reshape((tpl.shape[0],) + tuple(self.ishape[-2:])))
        out = np.empty((tpl.shape[1],) + maps.shape[1:], dtype=np.float32)
        _lib.check(self._L.surfh_maps_to_cube(self._plan, _lib.dptr(tpl), tpl.shape[0], tpl.shape[1], _lib.fptr(maps), _lib.fptr(out)))
        return out.astype(np.float64)

    def real_data_janskySR_to_jansky(self, data):
        """Jy/sr -> Jy normalisation of slit data (spectroModel.py:225-239)."""
        out = np.zeros_like(data)
        for k, ch in enumerate(self.channels):
            d = np.array(data[self._idx[k]: self._idx[k + 1]]).reshape(self.instrs_oshape[k])
            for s in range(self.instrs_oshape[k][1]):
                sl = ch.slicer.get_slit_slices(s)
                w = ch.slicer.get_slit_weights(s, sl)
                d[:, s, :, :] = d[:, s, :, :] * np.sum(w[0, 0, :]) * self.srfs[self.owned[k]]
            out[self._idx[k]: self._idx[k + 1]] = d.ravel()
        return out

    # ---- instrumentation ---------------------------------------------------------------------
    def profile_filter(self, prefix=None):
        """Time only the stages whose name starts with ``prefix`` (None: all)."""
        _lib.check(self._L.surfh_profile_filter(self._plan, prefix.encode() if prefix else None))

    def profile_enable(self, on=True):
        _lib.check(self._L.surfh_profile_enable(self._plan, 1 if on else 0))

    def profile_reset(self):
        _lib.check(self._L.surfh_profile_reset(self._plan))

    def profile(self) -> dict:
        """{kernel name: (launches, total ms)} measured with HIP events on the plan's stream."""
        n = self._L.surfh_profile_count(self._plan)
        out = {}
        for i in range(n):
            name, cnt, ms = C.c_char_p(), C.c_int64(), C.c_double()
            _lib.check(self._L.surfh_profile_get(self._plan, i, C.byref(name), C.byref(cnt), C.byref(ms)))
            out[name.value.decode()] = (cnt.value, ms.value)
        return out

    def debug_buffer(self, which: str) -> np.ndarray:
        dims = (C.c_int64 * 4)()
        _lib.check(self._L.surfh_debug_dims(self._plan, which.encode(), dims))
        shape = tuple(int(d) for d in dims)
        if which in ("info", "range", "ksteps", "otf") or which.startswith("xsinfo:"):
            return np.array(shape)
        while len(shape) > 1 and shape[-1] == 1:
            shape = shape[:-1]
        out = np.empty(int(np.prod(shape)), dtype=np.float32)
        n = self._L.surfh_debug_copy(self._plan, which.encode(), _lib.fptr(out), out.size)
        if n < 0:
            raise RuntimeError(self._L.surfh_last_error().decode())
        return out.reshape(shape)
