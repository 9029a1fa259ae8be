"""Regularised least-squares fusion by linear CG on top of ``spectroSigRLSCT``.

* ``QuadCriterion_MRS`` mirrors the reference's criterion class
  (surfh/Simulation/fusion_CT.py:66-265): same constructor, ``run_method('lcg' | 'mmmg', ...)``,
  ``get_crit_val``; the solvers it drives are the library's device-resident CG and 3MG
  (``qmm.lcg`` / ``qmm.mmmg`` restated, see include/surfh_amd.h:surfh_cg, surfh_mmmg).
* ``DistributedFusion`` is the multi-GPU form: one process per GPU, each rank owns a set of
  (band, pointings) units, x/r/d are replicated, and the only exchange per iteration is one
  RCCL all-reduce (sum) of the partial normal-equation product mu * A_r^T A_r d  ([T, Na, Nb] fp32)
  through ``torch.distributed``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import os

import numpy as np

from . import instru
from .models import spectroSigRLSCT


@dataclass
class OptimizeResult:
    x: np.ndarray
    grad_norm: List[float] = field(default_factory=list)
    nit: int = 0
    success: bool = False
    time: float = 0.0


class QuadCriterion_MRS:
    def __init__(self, mu_spectro, y_spectro, model_spectro, mu_reg, printing=False, gradient="separated"):
        """``gradient="joint"`` regularises with the Laplacian of ``Difference_Operator_Joint`` (fusion_CT.py:45-62); udft's
        ``laplacian(2)`` is absent from the reference tree, the 3 x 3 kernel [[0,-1,0],[-1,4,-1],[0,-1,0]] is restated: parity
        unpinned for that option (the operator is checked against the oracle's Fourier-domain form of the same kernel).
        NOTE on the data term: ``model_spectro.adjoint`` is the exact transpose of ``forward`` (what CG needs); the reference
        hands qmm its interpolating ``gridding_t`` adjoint instead (spectroModel.py:173-185), so its right-hand side
        mu A_ref^T y and hence its iterates differ from the ones computed here (tests/test_gpu_driver.py records by how much)."""
        assert isinstance(mu_reg, (float, int, list, np.ndarray))
        if gradient not in ("separated", "joint"):
            raise ValueError(f"gradient must be 'separated' or 'joint', not {gradient!r}")
        self.mu_spectro, self.y_spectro, self.model_spectro, self.mu_reg = mu_spectro, y_spectro, model_spectro, mu_reg
        self.n_spec = model_spectro.ishape[0]
        self.shape_of_output = tuple(model_spectro.ishape)
        self.printing, self.gradient, self.it = printing, gradient, 1
        self.L_crit_val = []

    def run_method(self, method="lcg", maximum_iterations=10, tolerance=1e-12, calc_crit=False, perf_crit=None,
                   value_init=0.5, checkpoint=None):
        """fusion_CT.py:118-238.  The four callback modes of the reference:
        calc_crit / perf_crit = False/None: none; False/set: print the gradient norm every iteration;
        True/set: print it and, at iterations 1, 6, 11, ... (``self.it % 5 == 2`` after the increment, :172-175),
        evaluate the criterion (one extra forward) into ``L_crit_val``; True/None: the reference hands
        ``get_crit_val`` itself to qmm as the callback, which receives an OptimizeResult and cannot work -- here the
        criterion of every iterate is recorded in ``L_crit_val`` instead.

        ``checkpoint=(path, every)`` (not in the reference, which saves its result once at the end, main_fusion.py:202-204):
        the iterate is written to ``path`` (.npz: x, it, grad_norm) every ``every`` iterations, through a temporary file and a
        rename, so an interrupted run restarts from ``load_checkpoint(path)`` as ``value_init`` (a warm start of the solver,
        fusion_CT.py:122-126: the search directions start afresh)."""
        assert isinstance(self.mu_reg, (int, float))       # fusion_CT.py:119
        solver = self.model_spectro.cg if method == "lcg" else self.model_spectro.mmmg     # fusion_CT.py:194-198
        # the regulariser is state of the plan: select this criterion's for the duration of the solve and put back what was
        # there, so that two criteria sharing one model (different `gradient`) do not change each other's operator
        prior_before = self.model_spectro.get_prior() if hasattr(self.model_spectro, "get_prior") else None
        self.model_spectro.set_prior(self.gradient)                                         # fusion_CT.py:141-162
        if isinstance(value_init, (int, float)):
            init = np.ones(self.shape_of_output) * value_init
        else:
            assert value_init.shape == self.shape_of_output
            init = value_init
        import time
        self.L_crit_val = []

        def record_crit(x):
            crit_val = self.get_crit_val(x)
            self.L_crit_val.append(crit_val)
            print(f"Criterion value = {crit_val}\n")

        def print_last_grad_norm(it, gn, x):
            print(f"Iteration n°{self.it}, Grad norm = {gn[-1]}")
            self.it = self.it + 1

        def print_last_grad_norm_and_crit(it, gn, x):
            print_last_grad_norm(it, gn, x)
            if self.it % 5 == 2:
                record_crit(x)

        if calc_crit and perf_crit is None:
            print(f"{method} : Criterion calculated at each iteration!")
            callback = lambda it, gn, x: record_crit(x)
        elif not calc_crit and perf_crit is not None:
            print(f"{method} : perf_crit calculated at each iteration!")
            callback = print_last_grad_norm
        elif calc_crit and perf_crit is not None:
            print(f"{method} : criterion and gradient printed at each iteration!")
            callback = print_last_grad_norm_and_crit
        else:
            callback = None
        if checkpoint is not None:
            ck_path, ck_every = checkpoint
            ck_every = int(ck_every)
            user_cb, done = callback, [0]

            def callback(it, gn, x):
                done[0] += 1
                if ck_every > 0 and done[0] % ck_every == 0:
                    save_checkpoint(ck_path, x, done[0], gn)
                return user_cb(it, gn, x) if user_cb is not None else None
        t0 = time.time()
        try:
            x, gn, nit = solver(self.y_spectro, mu=self.mu_spectro, mu_reg=self.mu_reg, x0=init,
                                max_iter=maximum_iterations, tol=tolerance, callback=callback)
        finally:
            if prior_before is not None:
                self.model_spectro.set_prior(prior_before)
        last = np.sqrt(gn[-1]) if method == "lcg" else gn[-1]      # lcg traces r.r, mmmg |grad|
        res = OptimizeResult(x=x.ravel(), grad_norm=list(gn), nit=nit,
                             success=bool(last < x.size * tolerance), time=time.time() - t0)
        if self.printing:
            print(f"Total time needed for {method} :", round(res.time, 3))
        return res

    def get_crit_val(self, x_hat):
        """(mu |y - A x|^2 + mu_reg (|Dr x|^2 + |Dc x|^2)) / 2   (fusion_CT.py:242-265)."""
        x_hat = np.asarray(x_hat).reshape(self.shape_of_output)
        data = self.mu_spectro * np.sum((self.y_spectro - self.model_spectro.forward(x_hat)) ** 2)
        if self.gradient == "joint":                 # |D x|^2, D = circular 3 x 3 Laplacian centred on the pixel (:254-255, :45-54)
            dx = 4 * x_hat - np.roll(x_hat, 1, 1) - np.roll(x_hat, -1, 1) - np.roll(x_hat, 1, 2) - np.roll(x_hat, -1, 2)
            return (data + self.mu_reg * np.sum(dx ** 2)) / 2
        dr = np.roll(x_hat, 1, axis=1) - x_hat
        dc = np.roll(x_hat, 1, axis=2) - x_hat
        return (data + self.mu_reg * np.sum(dr ** 2 + dc ** 2)) / 2


# ------------------------------------------------------------------------------------------------
# multi-GPU
# ------------------------------------------------------------------------------------------------
def save_checkpoint(path, x, it, grad_norm):
    """Iterate, iteration count and r.r trace as one .npz, written beside `path` first and renamed over it."""
    path = str(path)
    if not path.endswith(".npz"):
        path += ".npz"
    tmp = path[:-4] + ".tmp.npz"
    np.savez(tmp, x=np.asarray(x, dtype=np.float64), it=np.int64(it), grad_norm=np.asarray(list(grad_norm), dtype=np.float64))
    os.replace(tmp, path)
    return path


def load_checkpoint(path):
    """(x, iterations done, r.r trace) of a file written by ``save_checkpoint``."""
    path = str(path)
    if not path.endswith(".npz"):
        path += ".npz"
    with np.load(path, allow_pickle=False) as f:
        return f["x"], int(f["it"]), f["grad_norm"]


def band_cost(n_pix: int, geo) -> float:
    """Estimated time (microseconds) of one band per CG iteration, from the rates measured on MI355X
    (profiles/r02_bench_config3.json): R/R^T at 445 TFLOP/s algorithmic, the two folded 2-D transforms at
    0.30 us per plane of the band's window (251^2, scaled by N^3; 1.42 ms for the 4743 window planes of config 3),
    gather / scatter / slab sum at 0.143 us per plane."""
    P, S, Ldet, aout = geo.oshape
    Lin = geo.wslice.stop - geo.wslice.start
    nbs = geo.slicer.npix_slit_beta_width
    r_flops = 4.0 * P * S * Ldet * Lin * nbs * aout
    return r_flops / 445e6 + Lin * 0.30 * (n_pix / 251.0) ** 3 + Lin * 0.143 * (n_pix / 251.0) ** 2 * (P / 4.0)


def partition_lambda(costs: Sequence[float], world: int) -> List[List[Tuple[int, Tuple[int, int]]]]:
    """Assign (band, lambda part (i, n)) units to ranks.  world <= bands: whole bands ((0, 1) parts) by
    longest-processing-time greedy; world > bands: every band gets n_k >= 1 ranks (the costliest bands get the
    extra ones) and its wavelength window is cut into n_k contiguous parts.  The R contraction runs over
    lambda, so the parts' outputs ADD: the ranks of a band all-reduce their partial y (one extra, group-local
    collective), while the FFT-conv stage, the gather and the GEMM K range are all divided by n_k."""
    nb = len(costs)
    out: List[List[Tuple[int, Tuple[int, int]]]] = [[] for _ in range(world)]
    if world <= nb:
        load = [0.0] * world
        for k in sorted(range(nb), key=lambda i: -costs[i]):
            r = int(np.argmin(load))
            out[r].append((k, (0, 1)))
            load[r] += costs[k]
        for r in range(world):
            out[r].sort()
        return out
    share = [1] * nb
    for _ in range(world - nb):
        share[int(np.argmax([costs[i] / share[i] for i in range(nb)]))] += 1
    r = 0
    for k in range(nb):
        for i in range(share[k]):
            out[r].append((k, (i, share[k])))
            r += 1
    return out


def partition_balanced(costs: Sequence[float], lins: Sequence[int], world: int, snap: int = 16):
    """Cut the concatenation of all bands' wavelength windows (band order, cost per plane = band cost / planes)
    into `world` contiguous chunks of equal cost.  A rank gets units (band, ("planes", a, b)); a band cut by a
    chunk boundary is shared by the ranks on both sides (its partial outputs are all-reduced in its group).
    Boundaries closer than `snap` planes to a band edge snap to it."""
    total = float(sum(costs))
    out: List[List[Tuple[int, tuple]]] = [[] for _ in range(world)]
    # position -> (band, plane offset)
    def locate(c):
        acc = 0.0
        for k, ck in enumerate(costs):
            if c <= acc + ck or k == len(costs) - 1:
                off = int(round((c - acc) / ck * lins[k]))
                off = min(max(off, 0), lins[k])
                if off < snap:
                    off = 0
                if lins[k] - off < snap:
                    off = lins[k]
                return k, off
            acc += ck
    cuts = [(0, 0)] + [locate(total * (r + 1) / world) for r in range(world - 1)] + [(len(costs) - 1, lins[-1])]
    for r in range(world):
        (k0, a0), (k1, a1) = cuts[r], cuts[r + 1]
        for k in range(k0, k1 + 1):
            a = a0 if k == k0 else 0
            b = a1 if k == k1 else lins[k]
            if b > a:
                out[r].append((k, (0, 1) if (a == 0 and b == lins[k]) else ("planes", a, b)))
    return out


def partition_units(costs: Sequence[float], n_pointings: Sequence[int], world: int) -> List[List[Tuple[int, List[int]]]]:
    """Assign (band, pointing subset) units to ranks.  world <= bands: whole bands, longest-processing-time
    greedy; world > bands: every band gets >= 1 ranks (the costliest bands get the extra ones) and its
    pointings are split contiguously among them (pointings are additive, spectroModelChannel.py:236-262)."""
    nb = len(costs)
    out: List[List[Tuple[int, List[int]]]] = [[] for _ in range(world)]
    if world <= nb:
        load = [0.0] * world
        for k in sorted(range(nb), key=lambda i: -costs[i]):
            r = int(np.argmin(load))
            out[r].append((k, list(range(n_pointings[k]))))
            load[r] += costs[k]
        for r in range(world):
            out[r].sort()
        return out
    share = [1] * nb
    for _ in range(world - nb):
        k = int(np.argmax([costs[i] / share[i] if share[i] < n_pointings[i] else -1.0 for i in range(nb)]))
        share[k] += 1
    r = 0
    for k in range(nb):
        groups = np.array_split(np.arange(n_pointings[k]), share[k])
        for g in groups:
            out[r].append((k, [int(i) for i in g]))
            r += 1
    return out


def unit_share(lins, k, u) -> float:
    """Fraction of band k's cost a lambda unit u carries: (0, 1) whole band, (i, n) one of n equal parts, ("planes", a, b) a plane range."""
    return 1.0 if u == (0, 1) else ((u[2] - u[1]) / lins[k] if len(u) == 3 else 1.0 / u[1])


def assignment_times(costs, lins, ybytes, asg):
    """(per-rank compute loads, per-rank predicted times with the group-local all-reduces of shared bands) of an assignment, us."""
    members = {}
    for r, units in enumerate(asg):
        for k, u in units:
            members.setdefault(k, set()).add(r)
    loads, times = [], []
    for r, units in enumerate(asg):
        load = sum(costs[k] * unit_share(lins, k, u) for k, u in units)
        comm = 0.0
        for k in {k for k, _ in units}:
            m = len(members[k])
            if m > 1:
                comm += 20.0 + 2.0 * (m - 1) / m * ybytes[k] / 50e3
        loads.append(load)
        times.append(load + comm)
    return loads, times


def choose_lambda_assignment(costs, lins, ybytes, world):
    """Whole bands / equal parts of a band (``partition_lambda``) or equal-cost contiguous chunks (``partition_balanced``)?
    Default: whichever has the smaller predicted time of its slowest rank WITH the group-local all-reduces counted -- for every
    band a rank shares with others, 20 us + 2 (m - 1) / m * bytes of that band's partial outputs at 50 GB/s (one xGMI link per
    pair; the link model is unmeasured).  A band is therefore split only when that pays for the extra collective: config 3 on
    4 ranks is one band per GPU (930 us predicted against 1110 us for the equal-cost chunks, which cut 2A / 2B / 2C
    across ranks), on 8 ranks two equal parts per band (655 against 721 us: the chunks put most ranks into TWO groups).
    ``SURFH_PARTITION=balanced``: the compute-only rule of round 2 (whole bands / equal parts when balanced to 5 %, else the
    chunks).  Returns (assignment, per-rank compute loads, per-rank predicted time)."""
    def timed(asg):
        return assignment_times(costs, lins, ybytes, asg)
    cand, bal = partition_lambda(costs, world), partition_balanced(costs, lins, world)
    (lc, tc), (lb, tb) = timed(cand), timed(bal)
    if os.environ.get("SURFH_PARTITION", "comm") == "balanced":
        return (cand, lc, tc) if max(lc) <= 1.05 * max(lb) else (bal, lb, tb)
    return (cand, lc, tc) if max(tc) <= max(tb) else (bal, lb, tb)


def plan_assignment(prob: dict, world: int, split: str = "lambda", with_times: bool = False):
    """The unit assignment ``DistributedFusion`` uses for `world` ranks, with the predicted cost of every rank:
    ``(assignment, loads, imbalance)``, imbalance = max load / mean load - 1 of the compute loads (``with_times``: a fourth
    entry, the predicted per-rank times with the group-local all-reduces -- what the assignment is chosen by).  Host only."""
    from .geometry import ChannelGeometry
    ifus, pts = prob["ifus"], prob["pointings"]
    n_pix = len(prob["alpha_axis"])
    srfs = instru.get_srf([i.det_pix_size for i in ifus], prob["step_deg"] * 3600)
    geos = [ChannelGeometry(i, prob["alpha_axis"], prob["beta_axis"], prob["wavel"], s, p, prob["step_deg"])
            for i, s, p in zip(ifus, srfs, pts)]
    costs = [band_cost(n_pix, g) for g in geos]
    if split == "lambda":
        lins = [g.wslice.stop - g.wslice.start for g in geos]
        ybytes = [4 * int(np.prod(g.oshape)) for g in geos]
        asg, loads, times = choose_lambda_assignment(costs, lins, ybytes, world)
    else:
        asg = partition_units(costs, [len(p) for p in pts], world)
        loads = [sum(costs[k] * len(sel) / len(pts[k]) for k, sel in r) for r in asg]
        times = list(loads)
    mean = sum(loads) / len(loads)
    if with_times:
        return asg, loads, max(loads) / mean - 1.0, times
    return asg, loads, max(loads) / mean - 1.0


class DistributedFusion:
    """One rank of the channel-sharded CG.  ``prob`` is a dict as produced by ``surfh_amd.synth.problem``."""

    def __init__(self, prob: dict, rank: int = 0, world: int = 1, device: int = 0, with_ref: bool = False,
                 model_factory=None, split: str = "lambda"):
        """``model_factory(ifus, pointings, lam_slices)`` replaces the HIP operator (used by the gloo tests on
        CPU, where a checker-backed stand-in exposes the same ``*_dev`` methods on CPU tensors).
        ``split``: how a band is shared when there are more ranks than bands -- "lambda" (its wavelength
        window, partial y all-reduced inside the band's group) or "pointing" (its pointings, no extra
        collective but the band's FFT-conv work is repeated on every rank of the group)."""
        import contextlib
        import torch
        self.torch = torch
        self.rank, self.world, self.device = rank, world, device
        # SURFH_FORCE_DIST=1: run the collectives even in a world of one rank (the all-reduce of the normal-equation product on
        # the plan's stream); =2: also treat every band as shared, i.e. the lambda-split branch (forward, group-local all-reduce
        # of y, adjoint).  For a one-GPU box, where RCCL can only be exercised with one rank (tests/test_gpu_distributed.py).
        self.force = int(os.environ.get("SURFH_FORCE_DIST", "0")) if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        ifus, pts = prob["ifus"], prob["pointings"]
        n_pix = len(prob["alpha_axis"])
        # geometry of every band (cheap) -> costs -> unit assignment, identical on every rank
        from .geometry import ChannelGeometry
        srfs = instru.get_srf([i.det_pix_size for i in ifus], prob["step_deg"] * 3600)
        geos = [ChannelGeometry(i, prob["alpha_axis"], prob["beta_axis"], prob["wavel"], s, p, prob["step_deg"])
                for i, s, p in zip(ifus, srfs, pts)]
        self.costs = [band_cost(n_pix, g) for g in geos]
        self.group = None
        self.unit_groups = []          # (first, last) offsets in this rank's y and the process group, per shared band
        if split == "lambda":
            lins = [g.wslice.stop - g.wslice.start for g in geos]
            ybytes = [4 * int(np.prod(g.oshape)) for g in geos]
            # whole bands / equal parts, or equal-cost contiguous chunks: whichever predicts the faster slowest rank, the
            # group-local all-reduces of shared bands included (choose_lambda_assignment)
            self.assignment, _, self.predicted_us = choose_lambda_assignment(self.costs, lins, ybytes, world)
            self.units = self.assignment[rank]
            my_ifus = [ifus[k] for k, _ in self.units]
            my_pts = [pts[k] for k, _ in self.units]
            my_slices = [None if u == (0, 1) else u for _, u in self.units]
            # one process group per shared band; every rank creates every group, in the same (band) order
            groups = {}
            for k in range(len(ifus)):
                members = [r for r in range(world) if any(kk == k for kk, _ in self.assignment[r])]
                if len(members) > 1 or (self.force >= 2 and members):
                    grp = torch.distributed.new_group(ranks=members)
                    if rank in members:
                        groups[k] = grp
            self._band_groups = groups
            if groups:
                self.group = next(iter(groups.values()))
        else:
            self.assignment = partition_units(self.costs, [len(p) for p in pts], world)
            self.units = self.assignment[rank]
            my_ifus = [ifus[k] for k, _ in self.units]
            my_pts = [instru.CoordList([pts[k][i] for i in sel]) for k, sel in self.units]
            my_slices = None
        if model_factory is None:
            torch.cuda.set_device(device)
            self.tstream = torch.cuda.Stream(device=device)
            self.dev = f"cuda:{device}"
            self._ctx = lambda: torch.cuda.stream(self.tstream)
            self._sync = self.tstream.synchronize
            self.model = spectroSigRLSCT(prob["sotf"], prob["templates"], prob["alpha_axis"], prob["beta_axis"],
                                         prob["wavel"], my_ifus, prob["step_deg"], my_pts, device=device,
                                         with_ref=with_ref, stream=self.tstream.cuda_stream, lam_slices=my_slices)
        else:
            self.tstream = None
            self.dev = "cpu"
            self._ctx = contextlib.nullcontext
            self._sync = lambda: None
            self.model = model_factory(my_ifus, my_pts, my_slices)
        self.n = self.model.isize
        # The solver's vectors live in the Fourier domain of the maps where the operator offers it (surfh_normal_spec_dev: the
        # forward model reads the maps' spectra in the loader of its first transform pass, the adjoint's last pass writes them;
        # no transform of the maps, no padding, no prior kernel inside the iteration).  The basis is orthonormal, so the CG
        # recurrences and r.r are those of the maps.  SURFH_SPECTRAL_CG=0: vectors are the maps.
        self._agree_basis()
        self._finish_init(split)

    def _agree_basis(self):
        """Which basis the solver's vectors live in (the maps, or their scaled half spectra where the operator offers the
        spectral-domain calls -- which depends on the plan AND on its current regulariser: the joint Laplacian has no spectral
        form here).  Called at construction and again by ``start``: a ``set_prior`` between two solves must not leave a stale choice."""
        torch, world = self.torch, self.world
        self.spec = bool(getattr(self.model, "spec_supported", None)) and self.model.spec_supported() and \
            os.environ.get("SURFH_SPECTRAL_CG", "1") != "0"
        if world > 1 or self.force:
            # every rank decides from its own plan (transform kernels, pitch limits, prior): the vectors that are all-reduced
            # must have one basis and one length everywhere, so the ranks agree on the weakest answer
            flag = torch.tensor([1 if self.spec else 0], dtype=torch.int32, device=self.dev if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            self.spec = bool(int(flag.item()))
        self.nv = self.model.spec_size if self.spec else self.n      # floats of a solver vector
        if world > 1 or self.force:
            nvs = torch.tensor([self.nv, -self.nv], dtype=torch.int64, device=self.dev if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(nvs, op=torch.distributed.ReduceOp.MAX)
            assert int(nvs[0]) == self.nv and int(nvs[1]) == -self.nv, "ranks disagree on the length of the solver's vectors"

    def _finish_init(self, split):
        self._ytmp = None
        if split == "lambda" and getattr(self, "_band_groups", None):
            idx = np.cumsum([0] + [int(np.prod(c.oshape)) for c in self.model.channels]) if hasattr(self.model, "channels") \
                else np.asarray(self.model._idx)
            for u, (k, _) in enumerate(self.units):
                if k in self._band_groups:
                    self.unit_groups.append((int(idx[u]), int(idx[u + 1]), self._band_groups[k]))

    def _allreduce(self, t):
        if self.world > 1 or self.force:
            self.torch.distributed.all_reduce(t)

    def _reduce_shared(self, y):
        """Sum the partial outputs of every band this rank shares with others (ascending band order on all ranks)."""
        for a, b, grp in self.unit_groups:
            self.torch.distributed.all_reduce(y[a:b], group=grp)

    def make_data(self, maps, noise_rel=1e-2, seed=1):
        """y_r = A_r maps + N(0, sigma^2), sigma = noise_rel * rms(y_r)."""
        torch = self.torch
        with self._ctx():
            x = torch.as_tensor(np.ascontiguousarray(maps, dtype=np.float32), device=self.dev)
            y = torch.empty(self.model.osize, dtype=torch.float32, device=x.device)
            self.model.forward_dev(x, y)
            self._reduce_shared(y)                  # lambda parts of one band: the partial outputs add up
            if noise_rel:
                # the ranks sharing a band must draw the same noise: one generator per band, seeded by the band
                idx = np.asarray(self.model._idx)
                for u, (k, _) in enumerate(self.units):
                    g = torch.Generator(device=x.device).manual_seed(seed + 1000 * k)
                    seg = y[int(idx[u]): int(idx[u + 1])]
                    seg += torch.randn(seg.shape, generator=g, device=x.device, dtype=torch.float32) * (noise_rel * seg.square().mean().sqrt())
        self._sync()
        return y

    def normal(self, d, q, mu, mu_reg):
        """q = mu A^T A d (summed over ranks) + mu_reg (Dr^T Dr + Dc^T Dc) d."""
        if self.spec:
            m = self.model
            if self.group is None:
                if self.world == 1 and not self.force:
                    m.normal_spec_dev(d, q, mu, mu_reg)            # prior folded into the adjoint's last kernel
                    return
                m.normal_spec_dev(d, q, mu, 0.0)
            else:                                   # y = sum over the band's lambda parts, then each part's A^T
                if self._ytmp is None:
                    self._ytmp = self.torch.empty(m.osize, dtype=self.torch.float32, device=d.device)
                m.forward_spec_dev(d, self._ytmp)
                self._reduce_shared(self._ytmp)
                m.adjoint_spec_dev(self._ytmp, q, mu)
            self._allreduce(q)
            if mu_reg:
                m.prior_spec_add_dev(d, q, mu_reg)
            return
        if self.group is None:
            self.model.normal_dev(d, q, mu)
        else:                                       # y = sum over the band's lambda parts, then each part's A^T
            if self._ytmp is None:
                self._ytmp = self.torch.empty(self.model.osize, dtype=self.torch.float32, device=d.device)
            self.model.forward_dev(d, self._ytmp)
            self._reduce_shared(self._ytmp)
            self.model.adjoint_dev(self._ytmp, q)
            if mu != 1.0:
                q *= mu
        self._allreduce(q)
        if mu_reg:
            self.model.prior_add_dev(d, q, mu_reg)

    def start(self, y, mu=1.0, mu_reg=0.0, x0=None):
        torch, m = self.torch, self.model
        dev = self.dev
        self._agree_basis()              # the regulariser may have changed since construction (set_prior)
        self._nosync = hasattr(m, "cg_iter_nosync_dev")          # the HIP operator keeps the CG scalars on the device
        with self._ctx():
            shape = m.ishape
            if self.spec:
                self.x = torch.zeros(self.nv, dtype=torch.float32, device=dev)
                if x0 is not None:
                    m.to_spec_dev(torch.as_tensor(np.ascontiguousarray(x0, dtype=np.float32), device=dev), self.x)
                self.b = torch.empty_like(self.x)
                self.q = torch.empty_like(self.x)
                m.adjoint_spec_dev(y, self.b, mu)
            else:
                self.x = torch.zeros(shape, dtype=torch.float32, device=dev) if x0 is None else \
                    torch.as_tensor(np.ascontiguousarray(x0, dtype=np.float32), device=dev).clone()
                self.b = torch.empty_like(self.x)
                self.q = torch.empty_like(self.x)
                m.adjoint_dev(y, self.b)
                if mu != 1.0:
                    self.b *= mu
            self._allreduce(self.b)
            self.normal(self.x, self.q, mu, mu_reg)
            self.r = self.b - self.q
            self.d = self.r.clone()
            if self._nosync:
                m.cg_begin_dev(self.r, self.nv)
                self._trace = None
            else:
                self.rr = m.dot_dev(self.r, self.r, self.nv)
                self._trace = [self.rr]
        self.mu, self.mu_reg = mu, mu_reg
        self.it = 0

    def x_maps(self):
        """The current iterate as maps [T, Na, Nb] (a device tensor), whatever basis the solver's vectors live in."""
        if not self.spec:
            return self.x
        with self._ctx():
            out = self.torch.empty(self.model.ishape, dtype=self.torch.float32, device=self.x.device)
            self.model.from_spec_dev(self.x, out)
        self._sync()
        return out

    @property
    def grad_norm(self):
        """r.r of every iterate so far (reading it synchronises the device when the scalars live there)."""
        return list(self.model.cg_trace()) if self._nosync else list(self._trace)

    def step(self, refresh=50):
        """One CG iteration (qmm.lcg loop body; oracle/surfh_oracle.py:lcg documents the recurrences).  With the HIP operator
        nothing here waits for the device: normal operator, all-reduce and vector updates are queued on the plan's stream and
        the step lengths are formed on the device from device-resident scalars."""
        torch, m = self.torch, self.model
        fresh = bool(refresh) and self.it % refresh == 0
        with self._ctx():
            self.normal(self.d, self.q, self.mu, self.mu_reg)
            if self._nosync:
                if fresh:           # residual recomputed from scratch
                    m.cg_xupdate_nosync_dev(self.x, self.d, self.q, self.nv)
                    self.normal(self.x, self.q, self.mu, self.mu_reg)
                    m.cg_refresh_nosync_dev(self.r, self.b, self.q, self.d, self.nv)
                else:
                    m.cg_iter_nosync_dev(self.x, self.r, self.d, self.q, self.nv)
            else:
                rr_new = m.cg_step_dev(self.x, self.r, self.d, self.q, self.nv, self.rr)
                if fresh:
                    self.normal(self.x, self.q, self.mu, self.mu_reg)
                    m.residual_dev(self.r, self.b, self.q, self.nv)
                    rr_new = m.dot_dev(self.r, self.r, self.nv)
                m.cg_dir_dev(self.d, self.r, self.nv, rr_new / self.rr)
                self.rr = rr_new
                self._trace.append(rr_new)
        self.it += 1

    def lcg(self, y, mu=1.0, mu_reg=0.0, x0=None, max_iter=10, tol=1e-12, refresh=50, check_every=8):
        """``check_every``: with device-resident scalars the stopping test reads the trace only every that many iterations
        (the iterates stop at the first multiple of it past the tolerance; ``check_every=1`` reproduces qmm.lcg's count)."""
        self.start(y, mu, mu_reg, x0)
        for i in range(max_iter):
            self.step(refresh)
            if not self._nosync or (i + 1) % check_every == 0 or i + 1 == max_iter:
                if np.sqrt(self.grad_norm[-1]) < self.n * tol:
                    break
        self._sync()
        gn = self.grad_norm
        return OptimizeResult(x=self.x_maps().cpu().numpy().astype(np.float64), grad_norm=gn, nit=self.it,
                              success=bool(np.sqrt(gn[-1]) < self.n * tol))
