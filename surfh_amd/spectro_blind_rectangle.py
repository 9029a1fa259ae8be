"""``MRSBlurred``: the 2-D, no-rotation operator of the reference's deconvolution path
(surfh/Models/spectro_blind_rectangle.py:27-332, driver scripts/deconvolution_mrs_noRotation.py:170-212)
evaluated by the HIP library.

    y[p, s, a] = sum_beta  w_s[beta] * boxsum_alpha(crop_p(C x))[alpha0 + a*srf, beta]

C = 2-D OTF multiply, crop = integer-shift gridding (:286-307), box-sum = srf consecutive alpha rows
(:201-204), slit window with beta-edge weights, alpha decimation and beta sum (:206-208).  The crop is
exactly transposable, so ``adjoint`` is both the exact transpose and the reference's adjoint (:212-237).

``sotf`` may be ``[N_alpha, N_beta/2+1]`` (the reference's single image) or
``[L, N_alpha, N_beta/2+1]``: the L wavelength planes are independent problems evaluated as one batch
(BASELINE.json configs[4]); ``forward`` then maps ``[L, N_alpha, N_beta] -> [L, P*S*alpha_out]``.
"""
from __future__ import annotations

import ctypes as C
from math import ceil, floor

import numpy as np

from . import _lib, instru
from .linop import LinOp


class MRSBlurred(LinOp):
    def __init__(self, sotf, alpha_axis, beta_axis, instr: instru.IFU, step_degree: float,
                 pointings: instru.CoordList, *, device: int = 0, stream=None):
        self.sotf = sotf
        self.alpha_axis = np.asarray(alpha_axis, dtype=np.float64)
        self.beta_axis = np.asarray(beta_axis, dtype=np.float64)
        self.step_degree = step_degree
        self.instr = instr                    # not pixelised, as in the reference (:38-39)
        self.pointings = pointings
        self.srf = instru.get_srf([instr.det_pix_size], step_degree * 3600)[0]
        self.local_alpha_axis, self.local_beta_axis = instr.fov.local_coords(step_degree, 5 * step_degree, 5 * step_degree)
        self.local_im_shape = (len(self.local_alpha_axis), len(self.local_beta_axis))
        self.imshape = (len(self.alpha_axis), len(self.beta_axis))
        self.slices_shape = (len(pointings), instr.n_slit, ceil(self.npix_slit_alpha_width / self.srf))
        sotf_c = np.ascontiguousarray(sotf, dtype=np.complex128)
        self.batched = sotf_c.ndim == 3
        if not self.batched:
            sotf_c = sotf_c[None]
        self.n_planes = sotf_c.shape[0]
        if sotf_c.shape[1:] != (self.imshape[0], self.imshape[1] // 2 + 1):
            raise ValueError(f"sotf plane shape {sotf_c.shape[1:]} does not match the image {self.imshape}")
        n_out = int(np.prod(self.slices_shape))
        super().__init__(ishape=((self.n_planes,) if self.batched else ()) + self.imshape,
                         oshape=((self.n_planes, n_out) if self.batched else (n_out,)))

        # ---- tables ---------------------------------------------------------------------------
        S = instr.n_slit
        slices = [self.get_slit_slices(s) for s in range(S)]
        nbs = self.npix_slit_beta_width
        a0, a1 = slices[0][0].start, slices[0][0].stop
        weights = np.empty((S, nbs))
        for s, sl in enumerate(slices):
            if (sl[0].start, sl[0].stop) != (a0, a1) or sl[1].stop - sl[1].start != nbs:
                raise ValueError(f"slit {s}: window {sl} differs from slit 0 / npix_slit_beta_width={nbs}")
            weights[s] = self.get_slit_weights(s, sl)[0][0]
        na, nb = self.local_im_shape
        P = len(pointings)
        i0 = np.empty((P, na * nb), dtype=np.int32)
        i1 = np.empty_like(i0)
        y0 = np.empty((P, na * nb), dtype=np.float64)
        y1 = np.empty_like(y0)
        self.crops = []
        for p, pt in enumerate(pointings):          # integer crop (:286-307) written as degenerate bilinear taps
            ia = int(np.abs(self.alpha_axis - pt.alpha).argmin())
            ib = int(np.abs(self.beta_axis - pt.beta).argmin())
            sa, sb = ia - na // 2, ib - nb // 2
            if sa < 0 or sb < 0 or sa + na > self.imshape[0] or sb + nb > self.imshape[1] or na % 2 == 0 or nb % 2 == 0:
                raise ValueError(f"pointing {p}: the {na}x{nb} field of view does not fit in the {self.imshape} image")
            self.crops.append((sa, sa + na, sb, sb + nb))
            ra = np.repeat(np.arange(sa, sa + na), nb)
            rb = np.tile(np.arange(sb, sb + nb), na)
            la, lb = np.minimum(ra, self.imshape[0] - 2), np.minimum(rb, self.imshape[1] - 2)
            i0[p], i1[p], y0[p], y1[p] = la, lb, ra - la, rb - lb
        self._tab = dict(slit_beta0=np.ascontiguousarray([sl[1].start for sl in slices], dtype=np.int32),
                         slit_weights=np.ascontiguousarray(weights), i0=i0, i1=i1, y0=y0, y1=y1)
        d = _lib.ChannelDesc()
        d.wslice_start, d.wslice_stop = 0, self.n_planes
        d.n_pointings, d.n_slit, d.n_lambda_out, d.n_alpha_out, d.srf = P, S, self.n_planes, self.slices_shape[2], self.srf
        d.na, d.nb, d.alpha0, d.n_alpha_slit, d.n_beta_slit = na, nb, a0, a1 - a0, nbs
        d.slit_beta0, d.slit_weights = _lib.iptr(self._tab["slit_beta0"]), _lib.dptr(self._tab["slit_weights"])
        d.grid_i0, d.grid_i1, d.grid_y0, d.grid_y1 = _lib.iptr(i0), _lib.iptr(i1), _lib.dptr(y0), _lib.dptr(y1)
        d.wpsf = None                                # beta-sum mode (no spectral blur)
        cfg = _lib.Config()
        cfg.n_alpha, cfg.n_beta, cfg.n_lambda, cfg.n_templates = self.imshape[0], self.imshape[1], self.n_planes, 0
        cfg.templates = None
        cfg.sotf = sotf_c.view(np.float64).ctypes.data_as(_lib.c_double_p)
        cfg.n_channels, cfg.channels = 1, C.pointer(d)
        cfg.device, cfg.stream, cfg.split_k_forward = device, (C.c_void_p(stream) if stream else None), 0
        L = _lib.load()
        plan = C.c_void_p()
        _lib.check(L.surfh_plan_create(C.byref(cfg), C.byref(plan)), ValueError)
        self._L, self._plan = L, plan
        assert L.surfh_isize(plan) == self.isize and L.surfh_osize(plan) == self.osize

    def close(self):
        if getattr(self, "_plan", None):
            self._L.surfh_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- geometry (same rules as the reference class) -----------------------------------------------
    @property
    def npix_slit_alpha_width(self) -> int:
        step = self.local_alpha_axis[1] - self.local_alpha_axis[0]
        half = self.instr.fov.alpha_width / 2 / step
        return int(ceil(half)) - int(floor(-half))

    @property
    def slit_beta_width(self):
        return self.instr.fov.beta_width / self.instr.n_slit

    @property
    def npix_slit_beta_width(self) -> int:
        return int(ceil(self.slit_beta_width / (self.beta_axis[1] - self.beta_axis[0])))

    def slit_local_fov(self, slit_idx: int):
        return self.instr.slit_fov[slit_idx].local + self.instr.slit_shift[slit_idx]

    def get_slit_slices(self, slit_idx: int):
        """Beta trimming only: the alpha-length rule of Slicer is commented out here (:122-149)."""
        lf = self.slit_local_fov(slit_idx)
        sa, sb = lf.to_slices(self.local_alpha_axis, self.local_beta_axis)
        if sb.stop - sb.start > self.npix_slit_beta_width:
            far_end = abs(self.local_beta_axis[sb.stop] - lf.beta_end)
            far_start = abs(self.local_beta_axis[sb.start] - lf.beta_start)
            sb = slice(sb.start, sb.stop - 1) if far_end > far_start else slice(sb.start + 1, sb.stop)
        return sa, sb

    def get_slit_weights(self, slit_idx: int, slices):
        sa, sb = slices
        lf = self.slit_local_fov(slit_idx)
        db = self.local_beta_axis[1] - self.local_beta_axis[0]
        sel = self.local_beta_axis[sb]
        w = np.ones((sa.stop - sa.start, sb.stop - sb.start))
        if sel[0] - db / 2 < lf.beta_start:
            w[:, 0] = 1 - abs(sel[0] - db / 2 - lf.beta_start) / db
        if sel[-1] + db / 2 > lf.beta_end:
            w[:, -1] = 1 - abs(sel[-1] + db / 2 - lf.beta_end) / db
        assert np.all((0 <= w) & (w <= 1))
        if slit_idx > 0 and self.get_slit_slices(slit_idx - 1)[1].stop - 1 != sb.start:
            w[:, 0] = 1
        # the reference bounds this test by npix_slit_beta_width, not n_slit (:167): kept, including
        # its IndexError when there are fewer slits than beta columns
        if slit_idx < self.npix_slit_beta_width - 1:
            if sb.stop - 1 != self.get_slit_slices(slit_idx + 1)[1].start:
                w[:, -1] = 1
        return w[np.newaxis, ...]

    # ---- operator -------------------------------------------------------------------------------------
    def _call(self, fn, x, nin, shape_out):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
        if a.size != nin:
            raise ValueError(f"input has {a.size} elements, expected {nin}")
        out = np.empty(int(np.prod(shape_out)), dtype=np.float32)
        _lib.check(fn(self._plan, _lib.fptr(a), _lib.fptr(out)))
        return out.astype(np.float64).reshape(shape_out)

    def forward(self, x):
        return self._call(self._L.surfh_forward, x, self.isize, self.oshape)

    def adjoint(self, data):
        return self._call(self._L.surfh_adjoint, data, self.osize, self.ishape)

    def data_to_img(self, data):
        """The reference's quick-look back-projection of slit data (spectro_blind_rectangle.py:240-283; called by
        scripts/simulate_deconvolution_mrs_rectangle.py:193 and scripts/deconvolution_mrs_single_wavelength.py:159,194): every
        sample spread evenly over its slit's beta columns, put back on the local grid (slit windows with their beta-edge
        weights), summed over the srf-row window (the transposed box), values below 1 zeroed and local columns 5 / 153
        overwritten by their neighbours 6 / 152 as the reference does (so the local grid needs >= 154 columns, as there),
        placed in the image at each pointing.  Returns ``(weighted_mean, global_img)``: the mean over the pointings that
        cover a pixel (0 where none does; the reference leaves those entries uninitialised) and the plain sum.
        Host NumPy: a plotting aid on one image, not part of the operator."""
        if self.batched:
            raise ValueError("data_to_img is defined for a single image")
        if self.local_im_shape[1] < 154:
            raise IndexError(f"data_to_img patches local columns 5 and 153: the local grid has {self.local_im_shape[1]} columns")
        d = np.asarray(data, dtype=np.float64).reshape(self.slices_shape)
        na, nb = self.local_im_shape
        nbs, n_out = self.npix_slit_beta_width, self.slices_shape[2]
        cum = np.zeros((len(self.crops),) + self.imshape)
        for p, (a0, a1, b0, b1) in enumerate(self.crops):
            local = np.zeros((na, nb))
            for s in range(self.instr.n_slit):
                sl = self.get_slit_slices(s)
                w = self.get_slit_weights(s, sl)[0]
                bts = np.zeros((sl[0].stop - sl[0].start, sl[1].stop - sl[1].start))
                bts[: n_out * self.srf: self.srf, :] = np.repeat(d[p, s][:, None], nbs, axis=1) / nbs
                local[sl[0], sl[1]] += bts * w
            st = sum(np.roll(local, j, axis=0) for j in range(self.srf))      # transpose of the circular srf-row window sum
            st[st < 1] = 0
            st[:, 5] = st[:, 6]
            st[:, 153] = st[:, 152]
            cum[p, a0:a1, b0:b1] = st
        valid = np.sum(cum != 0, axis=0)
        total = np.sum(cum, axis=0)
        return np.divide(total, valid, out=np.zeros(self.imshape), where=valid != 0), total

    # ---- solver: regularised least squares by CG, one independent 2-D problem per plane -------------------
    def cg(self, data, mu=1.0, mu_reg=0.0, x0=None, max_iter=10, tol=1e-12, refresh=50, callback=None):
        """Device-resident linear CG on  mu |y - A x|^2 + mu_reg (|Dr x|^2 + |Dc x|^2)  (criterion_2D.py:60-250 with
        `qmm.lcg` restated).  Batched model: every plane is its own problem with its own step sizes; returns
        ``(x, grad_norm, nit)`` with ``grad_norm`` of shape ``[nit+1]`` (single image) or ``[nit+1, n_planes]``.
        ``callback(it, grad_norm, x)`` as for ``spectroSigRLSCT.cg``."""
        return self._solve(self._L.surfh_cg_planes_cb, data, mu, mu_reg, x0, max_iter, tol, refresh, callback)

    # ---- instrumentation (HIP events on the plan's stream, as spectroSigRLSCT) -----------------------------
    def profile_enable(self, on=True):
        _lib.check(self._L.surfh_profile_enable(self._plan, 1 if on else 0))

    def profile_reset(self):
        _lib.check(self._L.surfh_profile_reset(self._plan))

    def profile(self) -> dict:
        out = {}
        for i in range(self._L.surfh_profile_count(self._plan)):
            name, cnt, ms = C.c_char_p(), C.c_int64(), C.c_double()
            _lib.check(self._L.surfh_profile_get(self._plan, i, C.byref(name), C.byref(cnt), C.byref(ms)))
            out[name.value.decode()] = (cnt.value, ms.value)
        return out

    # the same loop on device tensors (torch), no host synchronisation inside: see include/surfh_amd.h
    def forward_dev(self, x_t, y_t):
        _lib.check(self._L.surfh_forward_dev(self._plan, C.c_void_p(x_t.data_ptr()), C.c_void_p(y_t.data_ptr())))

    def cg_begin_dev(self, y_t, x_t, mu=1.0, mu_reg=0.0):
        """``x_t`` [n_planes, Na, Nb] float32 on the plan's device: the start, then the current iterate (updated in place)."""
        _lib.check(self._L.surfh_cg_planes_begin_dev(self._plan, C.c_void_p(y_t.data_ptr()), float(mu), float(mu_reg), C.c_void_p(x_t.data_ptr())))

    def cg_step_dev(self, iters=1, refresh=50):
        _lib.check(self._L.surfh_cg_planes_step_dev(self._plan, int(iters), int(refresh)))

    def cg_rr(self):
        out = np.zeros(self.n_planes, dtype=np.float64)
        _lib.check(self._L.surfh_cg_planes_rr(self._plan, _lib.dptr(out)))
        return out

    def mmmg(self, data, mu=1.0, mu_reg=0.0, x0=None, max_iter=10, tol=1e-12, refresh=50, callback=None):
        """Device-resident 3MG on the same criterion (`qmm.mmmg` restated for quadratic objectives) -- what the 2-D
        deconvolution driver's ``method = "qmm"`` runs (scripts/deconvolution_mrs_noRotation.py:199-212).  Same returns as
        ``cg`` except that ``grad_norm`` holds |grad| (not squared)."""
        return self._solve(self._L.surfh_mmmg_planes_cb, data, mu, mu_reg, x0, max_iter, tol, refresh, callback)

    def _solve(self, fn, data, mu, mu_reg, x0, max_iter, tol, refresh, callback=None):
        y = np.ascontiguousarray(np.asarray(data, dtype=np.float32).reshape(-1))
        if y.size != self.osize:
            raise ValueError("data size mismatch")
        x0a = None if x0 is None else np.ascontiguousarray(np.asarray(x0, dtype=np.float32).reshape(-1))
        if x0a is not None and x0a.size != self.isize:
            raise ValueError("x0 size mismatch")
        x = np.empty(self.isize, dtype=np.float32)
        gn = np.zeros((max_iter + 1, self.n_planes), dtype=np.float64)
        nit = C.c_int32()
        err = []

        def tramp(_user, it, gptr, xptr):
            try:
                g = np.ctypeslib.as_array(gptr, shape=(it + 1, self.n_planes)).copy()
                xi = np.ctypeslib.as_array(xptr, shape=(self.isize,)).astype(np.float64).reshape(self.ishape)
                return 1 if callback(it, g if self.batched else g[:, 0], xi) else 0
            except BaseException as e:          # never unwind through the C frame
                err.append(e)
                return 1

        cb = _lib.CG_CALLBACK(tramp) if callback is not None else _lib.CG_CALLBACK()
        _lib.check(fn(self._plan, _lib.fptr(y), float(mu), float(mu_reg),
                      _lib.fptr(x0a) if x0a is not None else None, int(max_iter), float(tol), int(refresh),
                      _lib.fptr(x), _lib.dptr(gn), C.byref(nit), cb, None))
        if err:
            raise err[0]
        gn = gn[: nit.value + 1]
        return x.astype(np.float64).reshape(self.ishape), (gn if self.batched else gn[:, 0]).copy(), nit.value


class QuadCriterion_MRS_2D:
    """The reference's 2-D criterion (surfh/Simulation/criterion_2D.py:66-250): same constructor, ``run_method('lcg' | 'mmmg')``
    and ``get_crit_val``, on ``MRSBlurred`` (one image, or a stack of independent images solved together)."""

    def __init__(self, mu_spectro, y_spectro, model_spectro, mu_reg, printing=False, gradient="separated"):
        assert isinstance(mu_reg, (float, int, list, np.ndarray))
        if gradient != "separated":
            raise NotImplementedError("only the separated first-difference priors (NpDiff_r / NpDiff_c) are built")
        self.mu_spectro, self.y_spectro, self.model_spectro, self.mu_reg = mu_spectro, y_spectro, model_spectro, mu_reg
        self.shape_of_output = tuple(model_spectro.ishape)
        self.printing, self.gradient, self.it = printing, gradient, 1
        self.L_crit_val = []

    def run_method(self, method="lcg", maximum_iterations=10, tolerance=1e-12, calc_crit=False, perf_crit=None, value_init=0.5):
        assert isinstance(self.mu_reg, (int, float))             # criterion_2D.py:115
        solver = self.model_spectro.cg if method == "lcg" else self.model_spectro.mmmg       # criterion_2D.py:190-193
        init = np.ones(self.shape_of_output) * value_init if isinstance(value_init, (int, float)) else value_init
        assert tuple(np.shape(init)) == self.shape_of_output
        import time
        from .fusion import OptimizeResult
        self.L_crit_val = []
        self.it = 1

        # the four callback modes of criterion_2D.py:163-225 (the same as fusion_CT.py:163-225, see QuadCriterion_MRS.run_method)
        def record_crit(x):
            crit_val = self.get_crit_val(x)
            self.L_crit_val.append(crit_val)
            print(f"Criterion value = {crit_val}\n")

        def print_last_grad_norm(it, gn, x):
            print(f"Iteration n°{self.it}, Grad norm = {np.max(np.atleast_1d(gn[-1]))}")
            self.it = self.it + 1

        def print_last_grad_norm_and_crit(it, gn, x):
            print_last_grad_norm(it, gn, x)
            if self.it % 5 == 2:
                record_crit(x)

        if calc_crit and perf_crit is None:
            print(f"{method} : Criterion calculated at each iteration!")
            callback = lambda it, gn, x: record_crit(x)      # noqa: E731
        elif not calc_crit and perf_crit is not None:
            print(f"{method} : perf_crit calculated at each iteration!")
            callback = print_last_grad_norm
        elif calc_crit and perf_crit is not None:
            print(f"{method} : criterion and gradient printed at each iteration!")
            callback = print_last_grad_norm_and_crit
        else:
            callback = None
        t0 = time.time()
        x, gn, nit = solver(self.y_spectro, mu=self.mu_spectro, mu_reg=self.mu_reg, x0=init,
                            max_iter=maximum_iterations, tol=tolerance, callback=callback)
        last = np.max(np.atleast_1d(gn[-1]))
        last = np.sqrt(last) if method == "lcg" else last           # lcg traces r.r, mmmg |grad|
        res = OptimizeResult(x=x.ravel(), grad_norm=list(gn), nit=nit,
                             success=bool(last < np.prod(self.shape_of_output[-2:]) * tolerance), time=time.time() - t0)
        if self.printing:
            print(f"Total time needed for {method} :", round(res.time, 3))
        return res

    def get_crit_val(self, x_hat):
        """(mu |y - A x|^2 + mu_reg (|Dr x|^2 + |Dc x|^2)) / 2   (criterion_2D.py:252-275), summed over the planes."""
        x_hat = np.asarray(x_hat).reshape(self.shape_of_output)
        data = self.mu_spectro * np.sum((np.asarray(self.y_spectro).reshape(self.model_spectro.oshape)
                                         - self.model_spectro.forward(x_hat)) ** 2)
        dr = np.roll(x_hat, 1, axis=-2) - x_hat
        dc = np.roll(x_hat, 1, axis=-1) - x_hat
        return (data + self.mu_reg * np.sum(dr ** 2 + dc ** 2)) / 2
