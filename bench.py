#!/usr/bin/env python
"""Benchmark of the surfh hot path on MI355X: CG iterations / second.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one iteration of the regularised least-squares linear CG (qmm.lcg loop body,
surfh/Simulation/fusion_CT.py:194-225): one forward + one exact-transpose adjoint of the MRS
operator, the two first-difference priors and the vector updates.  The workload is the
configuration BASELINE.json's metric is quoted on: the 4-channel (1C, 2A, 2B, 2C) 251x251x4000
synthetic cube with a 4-point dither.  It fits one GPU, so N=1 runs all
four bands on one MI355X; for N>1 the same problem is sharded by (band, pointings) units and the
only exchange is one RCCL all-reduce of the [T,251,251] normal-equation product per iteration
(total work fixed -> "strong" scaling).  `--config 2` runs BASELINE.json configs[1]
(single band 2A, 251x251x1024) instead.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how `roofline` and `cpu_baseline`
are formed.  Inputs are resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA peak
MFMA_16BIT_PEAK_TF = 2500.0 # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (the instruction the spectral-blur GEMM issues)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_problem(cfg_name: str, **kw):
    """The synthetic problems of BASELINE.json's configs (surfh_amd/synth.py).  Config 4 (12 sub-bands, 8000 planes) runs on the
    reference driver's default 501 x 501 image (scripts/main_fusion.py:217): channel 4's field of view does not fit 251 pixels
    (the reference raises, cython_2D_interpolation.py:472-478)."""
    from surfh_amd import synth
    if cfg_name == "2":
        return synth.config2(**kw)
    if cfg_name == "4":
        return synth.config4(n_pix=501, **kw)
    return synth.config3(**kw)


WORKLOADS = {
    "2": ("CG-iterations/sec (forward+adjoint) on 251x251x1024 cube", "config2: band 2A, 251x251x1024 cube, 4-point dither, T=4, mu_reg=5e3"),
    "3": ("CG-iterations/sec (forward+adjoint) on 251x251x4000 cube",
          "config3: 4 MRS bands 1C,2A,2B,2C, 251x251x4000 cube, 4-point dither, T=4, mu_reg=5e3"),
    "4": ("CG-iterations/sec (forward+adjoint) on 501x501x8000 cube",
          "config4: all 12 MRS sub-bands, 501x501x8000 cube (the reference driver's default image size; band 4 does not fit 251), "
          "4-point dither, T=4, mu_reg=5e3"),
}


def cpu_baseline(cfg_name: str, budget_s: float):
    """The oracle (float64 NumPy/SciPy port of the reference chain) timed on the host cores on a
    bounded sample of the same workload: every `stride`-th cube plane, full detector axes,
    all bands and pointings; every stage costs ~1/stride of the full problem, so the
    full-size rate is sample_rate / stride."""
    from oracle import surfh_oracle as orc
    from surfh_amd import synth
    stride = {"2": 16, "3": 32, "4": 128}[cfg_name]
    prob = build_problem(cfg_name, lam_stride=stride)
    specs = [orc.ChannelSpec(i.fov.alpha_width, i.fov.beta_width, (0.0, 0.0), i.fov.angle, i.det_pix_size, i.n_slit,
                             i.w_blur.grating_resolution, i.wavel_axis, i.name) for i in prob["ifus"]]
    pts = [[(c.alpha, c.beta) for c in pl] for pl in prob["pointings"]]
    t0 = time.time()
    om = orc.OracleModel(prob["sotf"], prob["templates"], prob["alpha_axis"], prob["beta_axis"], prob["wavel"], specs,
                         prob["step_deg"], pts, box="direct")
    log(f"[cpu_baseline] oracle setup {time.time() - t0:.1f}s, Lc_sample={len(prob['wavel'])}")
    d = np.random.default_rng(0).standard_normal(om.ishape)
    t_all = time.time()
    orc.normal_apply(om, d, 1.0, 5e3)             # warm-up (counts against the budget)
    times = []
    while len(times) < 3 and (not times or (time.time() - t_all) + times[-1] < budget_s):
        t = time.time()
        orc.normal_apply(om, d, 1.0, 5e3)
        times.append(time.time() - t)
    t_it = float(np.median(times))
    return {"value": 1.0 / (t_it * stride), "unit": "it/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle normal-operator application (1 fwd + 1 exact adj + priors) on every {stride}th cube plane "
                      f"({len(prob['wavel'])} of {len(prob['wavel']) * stride} planes), all bands/pointings, full detector axes; "
                      f"{len(times)} timed reps, median {t_it:.2f}s; full-size rate = sample rate / {stride}",
            "seconds_per_sample_iteration": t_it}


def main_config5(args):
    """BASELINE.json configs[4]: the 2-D deconvolution path (scripts/deconvolution_mrs_noRotation.py:100-212 -> criterion_2D.py ->
    qmm.lcg) on 512 x 512 x 2048: band 1C geometry without rotation, four pointings, one independent regularised 2-D problem per
    wavelength plane, all planes batched in one plan.  A step = one CG iteration of every plane (forward + adjoint of MRSBlurred,
    first-difference priors, per-plane step lengths).  The planes do not couple: N ranks take 2048 / N planes each and nothing is
    exchanged (replicas on disjoint plane ranges)."""
    import torch
    import torch.distributed as dist
    from oracle import surfh_oracle as orc            # problem constants only (axes, PSF formula); nothing of it is timed
    from surfh_amd import instru, synth
    from surfh_amd.spectro_blind_rectangle import MRSBlurred
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    N, Lc = 512, 2048
    lo, hi = Lc * rank // world, Lc * (rank + 1) // world            # this rank's planes
    t0 = time.time()
    ax = synth.axes(N)
    wav = np.linspace(6.53, 7.65, Lc)[lo:hi]
    sotf = synth.ir2fr(synth.gaussian_psf(wav, synth.STEP), (N, N))
    ifu = instru.IFU(fov=instru.FOV(3.2 / 3600, 3.7 / 3600, origin=instru.Coord(0, 0), angle=0.0), det_pix_size=0.196, n_slit=21,
                     w_blur=instru.SpectralBlur(3355.0), pce=None, wavel_axis=np.linspace(6.6, 7.6, 10), name="1C")
    sd = synth.STEP_DEG
    pts = instru.CoordList([instru.Coord(a, b) for a, b in [(0.0, 0.0), (2 * sd, -3 * sd), (-4 * sd, 1 * sd), (3 * sd, 5 * sd)]])
    ts = torch.cuda.Stream(device=local)
    mb = MRSBlurred(sotf, ax, ax, ifu, sd, pts, device=local, stream=ts.cuda_stream)
    log(f"[rank {rank}] planes [{lo}, {hi}); problem + plan in {time.time() - t0:.1f}s; oshape {mb.oshape}")
    dev = torch.device(f"cuda:{local}")
    with torch.cuda.stream(ts):
        g = torch.Generator(device=dev).manual_seed(19940407 + rank)
        truth = torch.rand((hi - lo, N, N), generator=g, device=dev, dtype=torch.float32)
        y = torch.empty(int(np.prod(mb.oshape)), dtype=torch.float32, device=dev)
        mb.forward_dev(truth, y)
        y += torch.randn(y.shape, generator=g, device=dev, dtype=torch.float32) * (1e-2 * y.square().mean().sqrt())
        x = torch.zeros_like(truth)
        mu, mu_reg = 1.0, 0.05
        mb.cg_begin_dev(y, x, mu, mu_reg)
    ts.synchronize()
    rr0 = mb.cg_rr()
    prof_all, n_all = {}, 0
    with torch.cuda.stream(ts):
        for i in range(args.warmup):
            if not args.no_profile and i == min(1, args.warmup - 1):
                ts.synchronize()
                mb.profile_reset(); mb.profile_enable(True)
                n_all = args.warmup - i
            mb.cg_step_dev(1)
    if n_all:
        prof_all = mb.profile()
        mb.profile_enable(False)

    def fence():
        ts.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    with torch.cuda.stream(ts):
        mb.cg_step_dev(args.steps)
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    rr1 = mb.cg_rr()
    if rank == 0:
        for name, (cnt, ms) in sorted(prof_all.items(), key=lambda kv: -kv[1][1]):
            log(f"[prof warm-up] {name:28s} launches {cnt:5d}  avg {ms / max(cnt, 1):8.4f} ms  per-step {ms / n_all:8.4f} ms")
        Nf = N * (N // 2 + 1)
        Lr = hi - lo
        stage = {k: v for k, v in prof_all.items() if k.startswith(("dft_", "specmix_"))}
        t_stage = sum(v[1] for v in stage.values()) * 1e-3 / max(n_all, 1)
        # per plane and direction the chain reads the image, reads the OTF and writes the blurred image: 2 N^2 4 + Nf 8 bytes
        b_stage = 2.0 * Lr * (2 * N * N * 4 + Nf * 8)
        dft = {k: v for k, v in prof_all.items() if k.startswith("dft_")}
        dom_cnt = sum(v[0] for v in dft.values())
        dom_ms = sum(v[1] for v in dft.values())
        roof = None
        if dom_cnt:
            # a step holds four 2-D transforms (x -> spectrum and product -> image, in each direction) = eight pass launches
            # (with the OTF products formed inside the passes' loaders -- no specmix launches -- the eight launches share the stage's bytes)
            fused_products = not any(k.startswith("specmix_") for k in prof_all)
            bytes_launch = b_stage / (dom_cnt / max(n_all, 1)) if fused_products else 0.5 * Lr * (Nf * 8 + N * N * 4)
            ach = bytes_launch / (dom_ms / dom_cnt * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "traffic_source": None, "kernel": "dft_ct_kernel" if any(k.startswith("dft_ct_") for k in dft) else "dft_h2_kernel",
                    "launches": dom_cnt, "avg_ms": dom_ms / dom_cnt,
                    "note": ("per launch: an eighth of the stage's algorithmic bytes (the OTF products run inside the passes), planes x (2 N^2 4 + Nf 8) / 4"
                             if fused_products else "per launch: half of a 2-D transform's algorithmic bytes, planes x (N (N/2+1) 8 + N^2 4) / 2") +
                            f"; HIP-event times of the {n_all} warm-up step(s)"}
        out = {"metric": "CG-iterations/sec (forward+adjoint) on 512x512x2048 cube, 2-D deconvolution path", "value": args.steps / el, "unit": "it/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None,
               "dtype": "f32 (storage and accumulation; the DFT passes as 2-piece fp16 splits, 3 products per fp32 product, on the 16-bit matrix cores)",
               "data": "synthetic",
               "config": {"workload": "config5: 2-D deconvolution (MRSBlurred, band 1C geometry, no rotation, 4 pointings) of 2048 independent "
                                      "512x512 planes, mu_reg=0.05, x0=0", "parallelism": f"{world} rank(s), {Lr} planes each, no exchange",
                          "rr_median_first_last": [float(np.median(rr0)), float(np.median(rr1))]},
               "roofline": roof,
               "roofline_fft_conv_stage": {"bound": "hbm", "achieved": b_stage / t_stage / 1e9 if t_stage else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": b_stage / t_stage / 1e9 / HBM_PEAK_GBS if t_stage else None, "ms_per_step": t_stage * 1e3,
                                           "kernel": "+".join(sorted(stage)),
                                           "note": "whole stage (transform passes + OTF products) against 2 x planes x (2 N^2 4 + Nf 8) bytes per step: "
                                                   "per direction the image read, the OTF read, the blurred image written"},
               "stage_ms_per_step": {k: round(v[1] / max(n_all, 1), 4) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][1])},
               "cpu_baseline": None}
        if world == 1 and args.cpu_seconds > 0:
            try:      # the oracle's 2-D operator (float64 port) on a few planes, all host cores through scipy.fft / BLAS
                sel = list(range(0, Lr, max(1, Lr // 8)))[:8]
                spec = orc.ChannelSpec(3.2 / 3600, 3.7 / 3600, (0.0, 0.0), 0.0, 0.196, 21, 3355.0, np.linspace(6.6, 7.6, 10), "1C")
                bo = orc.BlurredOracle(sotf[sel], ax, ax, spec, sd, [(0.0, 0.0), (2 * sd, -3 * sd), (-4 * sd, 1 * sd), (3 * sd, 5 * sd)])
                d = np.random.default_rng(0).standard_normal((len(sel), N, N))
                bo.adjoint(bo.forward(d))
                tt = []
                while len(tt) < 3 and (not tt or sum(tt) + tt[-1] < args.cpu_seconds):
                    t1 = time.time()
                    q = bo.adjoint(bo.forward(d))
                    q = q + 0.05 * (orc.diff_r_t(orc.diff_r(d)) + orc.diff_c_t(orc.diff_c(d)))
                    tt.append(time.time() - t1)
                t_it = float(np.median(tt))
                out["cpu_baseline"] = {"value": len(sel) / Lr / t_it, "unit": "it/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"oracle normal operator + priors on {len(sel)} of {Lr} planes (independent problems: full-size rate = "
                                                 f"sample rate x {len(sel)}/{Lr}); {len(tt)} reps, median {t_it:.2f}s",
                                       "seconds_per_sample_iteration": t_it}
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    mb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed CG iterations (400 x 3 ms: the timed region is >= 1 s and holds the residual refreshes of qmm.lcg, one every 50 iterations)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="3", choices=["2", "3", "4", "5"])
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-verify", action="store_true", help="skip the randn dot test on the float64-accumulating verification plan")
    ap.add_argument("--plan-only", action="store_true", help="no GPU: print the unit assignment and the predicted per-rank cost "
                    "imbalance for --gpus N on the chosen config (gate <= 15 %%, SURVEY.md 8e) and exit")
    args = ap.parse_args()
    if args.plan_only:
        from surfh_amd import synth
        from surfh_amd.fusion import plan_assignment
        prob = build_problem(args.config, geometry_only=True)
        asg, loads, imb, times = plan_assignment(prob, args.gpus, with_times=True)
        print(json.dumps({"config": args.config, "n_gpus": args.gpus, "assignment": repr(asg),
                          "predicted_compute_us_per_iteration_per_rank": [round(v, 1) for v in loads], "compute_imbalance": round(imb, 4),
                          "predicted_us_with_group_allreduces_per_rank": [round(v, 1) for v in times],
                          "note": "chosen by the predicted time of the slowest rank, group-local all-reduces of split bands included "
                                  "(SURFH_PARTITION=balanced: compute-only rule, imbalance gate 15 %); link model unmeasured"}))
        return

    if args.config == "5":
        return main_config5(args)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal on a one-GPU box: SURFH_REHEARSAL=1 puts every rank on device 0 and uses gloo for the
    # collectives (RCCL refuses two ranks on one device); the production path is nccl (= RCCL), one rank per GPU
    rehearsal = os.environ.get("SURFH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))

    from surfh_amd import synth
    from surfh_amd.fusion import DistributedFusion
    t0 = time.time()
    prob = build_problem(args.config)
    log(f"[rank {rank}] problem built in {time.time() - t0:.1f}s")
    t0 = time.time()
    fus = DistributedFusion(prob, rank=rank, world=world, device=local)
    m = fus.model
    log(f"[rank {rank}] plan built in {time.time() - t0:.1f}s; units {fus.units}; osize {m.osize}; "
        f"cube columns x rows the tables touch [a_lo, a_hi) x [b_lo, b_hi) = {[int(v) for v in m.debug_buffer('range')]}")
    y = fus.make_data(prob["maps"])
    mu, mu_reg = 1.0, 5e3
    fus.start(y, mu, mu_reg, x0=None)
    # Warm-up steps are bracketed stage by stage with HIP events (per-stage log, FFT-conv stage roofline).  Two event packets
    # per stage cost 4.6 % of an iteration when all ~45 stages carry them, so the timed region brackets only the dominant
    # kernel group (the one `roofline` reports), found from the warm-up profile.
    prof_all, n_all = {}, 0
    for i in range(args.warmup):
        if not args.no_profile and i == min(1, args.warmup - 1):      # skip the first step (lazy initialisation) when there are more
            fus._sync()
            m.profile_filter(None)
            m.profile_reset()
            m.profile_enable(True)
            n_all = args.warmup - i
        fus.step()
    if n_all:
        prof_all = m.profile()
        m.profile_enable(False)
    dom_prefix = None
    if prof_all:
        tot = {}
        for name, (cnt, ms) in prof_all.items():
            key = "gemm_wblur" if name.startswith("gemm_wblur") else "dft_" if name.startswith("dft_") else name
            tot[key] = tot.get(key, 0.0) + ms
        dom_prefix = max(tot, key=tot.get)

    def fence():
        fus._sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    if not args.no_profile:
        m.profile_filter(dom_prefix)
        m.profile_reset()
        m.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fus.step()
    fence()
    el = time.perf_counter() - t0
    prof = {}
    if not args.no_profile:
        prof = m.profile()
        m.profile_enable(False)
        m.profile_filter(None)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # parity gate reported with the number: the dot test of this rank's operator at benchmark size, inner
    # products accumulated in float64, non-negative test vectors (the physical regime; gate < 1e-6).  Outside the timed region.
    gate = None
    if rank == 0:
        rng = np.random.default_rng(5)
        v, u = rng.random(m.isize), rng.random(m.osize)
        lhs = float(np.vdot(np.asarray(m.rmatvec(u), dtype=np.float64), v))
        rhs = float(np.vdot(u, np.asarray(m.matvec(v), dtype=np.float64)))
        gate = {"dot_test_gap": abs(lhs - rhs) / abs(rhs), "gate": 1e-6,
                "note": "|<A^T u, v> - <u, A v>| / |<u, A v>| of the operator this rank holds, at benchmark size; the <=1e-5 "
                        "forward / adjoint gates against the float64 oracle and the bit-exact index tables are tests/ (pytest -m gpu)"}
        log(f"[parity gate] dot-test gap {gate['dot_test_gap']:.2e}")
        if world == 1 and not args.no_verify:
            # the reference's own dot test (randn vectors, test/sandbox_dottest.py:16-27) on the verification plan: the same
            # operator with every long sum accumulated in float64 (surfh_config.verify); gate < 1e-6
            try:
                from surfh_amd.models import spectroSigRLSCT
                t0 = time.time()
                mv = spectroSigRLSCT(prob["sotf"], prob["templates"], prob["alpha_axis"], prob["beta_axis"], prob["wavel"],
                                     prob["ifus"], prob["step_deg"], prob["pointings"], device=local, with_ref=False, verify=True)
                vr, ur = (rng.standard_normal(n).astype(np.float32).astype(np.float64) for n in (mv.isize, mv.osize))
                lhs = float(np.vdot(np.asarray(mv.rmatvec(ur), dtype=np.float64), vr))
                rhs = float(np.vdot(ur, np.asarray(mv.matvec(vr), dtype=np.float64)))
                gate["dot_test_gap_randn_verification_plan"] = abs(lhs - rhs) / abs(rhs)
                avp = np.asarray(m.matvec(vr), dtype=np.float64)
                gp = float(np.vdot(np.asarray(m.rmatvec(ur), dtype=np.float64), vr)) - float(np.vdot(ur, avp))
                gate["dot_test_gap_randn_production_normalised"] = abs(gp) / (np.linalg.norm(ur) * np.linalg.norm(avp))
                mv.close()
                log(f"[parity gate] randn dot-test gap on the verification plan {gate['dot_test_gap_randn_verification_plan']:.2e} "
                    f"({time.time() - t0:.1f}s); production plan, normalised by |u||Av|: {gate['dot_test_gap_randn_production_normalised']:.2e}")
            except Exception as e:      # the gate must never hide the number
                gate["dot_test_gap_randn_verification_plan"] = None
                gate["verification_error"] = repr(e)

    if rank == 0:
        for name, (cnt, ms) in sorted(prof_all.items(), key=lambda kv: -kv[1][1]):
            log(f"[prof warm-up] {name:28s} launches {cnt:5d}  avg {ms / max(cnt, 1):8.4f} ms  per-step {ms / n_all:8.4f} ms")
        for name, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            log(f"[prof timed]   {name:28s} launches {cnt:5d}  avg {ms / max(cnt, 1):8.4f} ms  per-step {ms / args.steps:8.4f} ms")
        N = len(prob["alpha_axis"])
        Nf = N * (N // 2 + 1)
        info = m.debug_buffer("info")
        Lown = int(info[2])      # planes this rank stores (union of its channels' windows)
        # HIP-event names -> kernel symbols as rocprofv3 reports them (profiles/*_kernel_stats_*.csv)
        def symbol(name):
            if name.startswith("gemm_wblur"):
                if os.environ.get("SURFH_WBLUR_FP32") == "1":
                    return "gemm_f32_kernel<128, 128>"
                return "gemm_nt_f16x2_cc_kernel"
            if name.startswith("specmix_"):       # interleaved spectra where the two-piece fp16 passes run (dft_h2.hip)
                return name + ("_ilv_kernel" if any(k.startswith(("dft_h2_", "dft_ct_")) for k in prof_all) else "_kernel")
            if name.startswith("dft_h2_") and name.endswith("_adjmix"):
                return "dft_h2_adjmix_kernel"      # the adjoint's last pass with the conj(OTF) product and the wavelength reduction fused in
            if name.startswith("dft_h2_"):
                return "dft_h2_kernel"             # four template instances <KIND, MIX> of one kernel (dft_h2.hip)
            if name.startswith("dft_ct_"):
                return "dft_ct_kernel"             # template instances <R, loader, epilogue> of one kernel (dft_ct.hip)
            if name.startswith("spmm_"):
                f16 = os.environ.get("SURFH_WBLUR_FP32") != "1"
                if name == "spmm_gather_fwd" and f16:
                    return "spmm_group_gather_f16_kernel" if os.environ.get("SURFH_GATHER_GROUPED") != "0" and \
                        os.environ.get("SURFH_GATHER_SORTED") != "0" else "spmm_rows_f16_kernel"
                if name == "spmm_scatter_adj" and os.environ.get("SURFH_SCATTER_GROUPED") != "0":
                    return "spmm_group_scatter_kernel"
                return "spmm_rows_kernel"
            if name.startswith("gemm_dft_") and name.endswith("_maps"):
                return "gemm_f32_kernel<64, 64>"
            return name + "_kernel"
        def grouped(pr):
            gr = {}
            for name, (cnt, ms) in pr.items():
                a = gr.setdefault(symbol(name), [0, 0.0])
                a[0] += cnt
                a[1] += ms
            return gr
        groups = grouped(prof)                 # timed region: the dominant kernel group only
        groups_all = grouped(prof_all)         # warm-up steps: every stage
        roof = None
        stage_ms = {k: round(v[1] / max(n_all, 1), 4) for k, v in sorted(groups_all.items(), key=lambda kv: -kv[1][1])}
        traffic_file = os.path.join(ROOT, "profiles", f"r03_pmc_traffic_config{args.config}.json")
        for alt in (f"r02_pmc_traffic_config{args.config}.json", f"r01_final_pmc_traffic_config{args.config}.json"):
            if not os.path.exists(traffic_file):
                traffic_file = os.path.join(ROOT, "profiles", alt)
        pmc = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}
        if groups:
            dom = max(groups, key=lambda k: groups[k][1])
            cnt, ms = groups[dom]
            avg_s = ms / cnt * 1e-3
            per_step = cnt / max(1, args.steps)
            traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
            if dom.startswith("gemm_nt") or dom.startswith("gemm_f32_kernel<128"):
                # R and R^T: 2 * P*S*Ldet*alpha_out * Lin*n_beta flops per channel and direction (SURVEY.md 8d F_iter)
                flops_step = sum(2.0 * 2.0 * np.prod(c.oshape) * (c.wslice.stop - c.wslice.start) * c.slicer.npix_slit_beta_width
                                 for c in m.channels)
                ach = flops_step / per_step / avg_s / 1e12
                nprod = 3.0 if dom.startswith("gemm_nt_f16x2") else 1.0
                peak = MFMA_16BIT_PEAK_TF if nprod > 1.0 else MFMA_F32_PEAK_TF
                roof = {"bound": "mfma", "achieved": nprod * ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": nprod * ach / peak, "traffic": traffic, "kernel": dom, "launches": cnt,
                        "avg_ms": ms / cnt, "algorithmic_fp32_tflops": ach,
                        "note": "matrix-core flops actually issued (each fp32 product = 3 fp16 products of a two-piece split) "
                                "against the dense peak of that instruction"}
            else:
                # FFT-conv stage: a 2-D transform of the owned planes algorithmically moves Lown*(Nf*8 + N^2*4) bytes
                # (SURVEY.md 8d); a CG step holds two (one per direction), each made of a complex pass along alpha and a
                # real <-> complex pass along beta: half of a transform's bytes per launch.
                bytes_launch = 0.5 * Lown * (Nf * 8 + N * N * 4) if dom.startswith(("dft_h2", "dft_ct")) else None
                ach = bytes_launch / avg_s / 1e9 if bytes_launch else None
                roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic, "kernel": dom,
                        "launches": cnt, "avg_ms": ms / cnt}
            if roof is not None:
                # `traffic` is not measured in this run: PMC counters need their own rocprofv3 passes (tools/pmc_traffic.py)
                roof["traffic_source"] = os.path.relpath(traffic_file, ROOT) if traffic is not None else None
        refreshes = sum(1 for i in range(args.warmup, args.warmup + args.steps) if i % 50 == 0)
        out = {
            "metric": WORKLOADS[args.config][0],
            "value": args.steps / el, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (storage and accumulation; products as 2-piece fp16 splits -- 3 products per fp32 product, 1 on the far tails of the spectral response -- on the 16-bit matrix cores, in the spectral-blur GEMM and in the DFT passes)", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config][1],
                       "parallelism": f"{world} rank(s), (band,pointings) units {fus.assignment}",
                       "osize_rank0": int(m.osize), "grad_norm_first_last": [fus.grad_norm[0], fus.grad_norm[-1]]},
            "refreshes_in_timed_region": refreshes,
            "roofline": roof, "parity_gates": gate, "stage_ms_per_step": stage_ms,
            "stage_ms_note": f"per-stage HIP-event times from the {n_all} untimed warm-up step(s) with every stage bracketed; "
                             "the timed region brackets only the kernel group of `roofline`",
        }
        # the HBM-bound half of the path as a whole: everything a step spends on its two 2-D transforms, the OTF product and
        # the wavelength reduction (the DFT passes, the fused adjoint tail or the separate reduction kernel) against the
        # algorithmic bytes of the FFT-conv stage.  Per-stage times of the warm-up steps (every stage bracketed).
        stage_keys = ("dft_h2", "dft_ct", "gemm_dft_rows", "gemm_dft_cols", "specmix_adj", "specmix_fwd")
        dft = [(k, v) for k, v in groups_all.items() if k.startswith(stage_keys)]
        if dft:
            n_l = sum(v[0] for _, v in dft)
            t_step = sum(v[1] for _, v in dft) * 1e-3 / max(n_all, 1)
            ach = 2.0 * Lown * (Nf * 8 + N * N * 4) / t_step / 1e9
            out["roofline_fft_conv_stage"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": ach / HBM_PEAK_GBS, "kernel": "+".join(sorted(k for k, _ in dft)),
                                              "launches_per_step": n_l / max(n_all, 1), "ms_per_step": t_step * 1e3,
                                              "note": "whole stage: algorithmic bytes of a step's two 2-D transforms (OTF read once and "
                                                      "cube written / read once per direction) over the time of every kernel of the stage; "
                                                      "the passes visit only the spectrum tiles inside the OTF's support (otf_tiles_in_support_of), "
                                                      "so they move fewer bytes than the algorithmic count",
                                              "otf_tiles_in_support_of": [int(v) for v in m.debug_buffer("otf")[:2]]}
        # the matrix-core half of the path, whichever group dominates: R / R^T against the fp32-MFMA peak
        gm = [(k, v) for k, v in (groups if dom_prefix == "gemm_wblur" else groups_all).items() if k.startswith("gemm_nt")]
        if gm:
            n_l = sum(v[0] for _, v in gm)
            t_s = sum(v[1] for _, v in gm) * 1e-3
            steps_seen = args.steps if dom_prefix == "gemm_wblur" else max(n_all, 1)
            flops_step = sum(2.0 * 2.0 * np.prod(c.oshape) * (c.wslice.stop - c.wslice.start) * c.slicer.npix_slit_beta_width
                             for c in m.channels)
            ach = flops_step * steps_seen / t_s / 1e12
            # K steps of a tile that keep three 16-bit products per fp32 product ("near") and steps kept as one ("far", gemm_cc16.hip):
            # the two GEMMs of a channel have the same algorithmic flops, so the issued products average over the directions
            ks = [float(v) for v in m.debug_buffer("ksteps")]
            nprod = 0.5 * sum((3.0 * a + b) / (a + b) if a + b > 0 else 3.0 for a, b in ((ks[0], ks[1]), (ks[2], ks[3])))
            out["roofline_spectral_blur_gemm"] = {"bound": "mfma", "achieved": nprod * ach, "peak": MFMA_16BIT_PEAK_TF, "unit": "TFLOP/s",
                                                  "frac": nprod * ach / MFMA_16BIT_PEAK_TF, "kernel": gm[0][0], "launches": n_l,
                                                  "avg_ms": t_s * 1e3 / n_l, "algorithmic_fp32_tflops": ach,
                                                  "note": f"matrix-core flops issued ({nprod:.2f} 16-bit products per fp32 product: three on the K steps near "
                                                          "the spectral response's diagonal, one on its far tails) against the dense fp16 / bf16 MFMA peak",
                                                  "k_steps_near_far_forward_adjoint": [int(v) for v in ks],
                                                  "traffic": pmc.get(gm[0][0], {}).get("hbm_bytes_per_launch")}
        # The iteration as a whole against SURVEY.md 8d's one-pass-per-stage byte model B_iter (OTF read, blurred cube written / read,
        # per channel: cube window read, local cube written and read per pointing, W read once, y written / read; per direction;
        # plus the CG vectors) and its flop count F_iter -- the bound of the path is max(t_HBM, t_MFMA).  The fused gather never
        # materialises the local cubes the model counts, so the achieved figure is a rate of ALGORITHMIC bytes, not of HBM traffic.
        try:
            T = int(m.ishape[0])
            b_ch = 0.0
            for c in m.channels:
                P, S, Ldet, aout = (int(v) for v in c.oshape)
                Lin = c.wslice.stop - c.wslice.start
                na, nb = c.local_im_shape
                b_ch += 4.0 * (Lin * N * N + 2.0 * P * Lin * na * nb + Ldet * Lin * c.slicer.npix_slit_beta_width + P * S * Ldet * aout)
            b_iter = 2.0 * (Lown * Nf * 8.0 + Lown * N * N * 4.0 + b_ch) + 10.0 * T * N * N * 4.0
            f_iter = sum(2.0 * 2.0 * np.prod(c.oshape) * (c.wslice.stop - c.wslice.start) * c.slicer.npix_slit_beta_width for c in m.channels)
            t_hbm, t_f32 = b_iter / (HBM_PEAK_GBS * 1e9), f_iter / (MFMA_F32_PEAK_TF * 1e12)
            t_it = el / args.steps
            out["roofline_iteration"] = {"bound": "hbm" if t_hbm >= t_f32 else "mfma", "achieved": b_iter / t_it / 1e9, "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": t_hbm / t_it, "algorithmic_bytes": b_iter, "algorithmic_flops": f_iter,
                                         "t_hbm_ms": t_hbm * 1e3, "t_mfma_fp32_ms": t_f32 * 1e3, "frac_of_max_bound": max(t_hbm, t_f32) / t_it,
                                         "note": "SURVEY.md 8d: B_iter at the HBM peak and F_iter at the fp32 matrix-core peak against the measured "
                                                 "iteration of this rank's units; `frac` = t_HBM / t_iteration; the products run as fp16 splits, so "
                                                 "the fp32-MFMA time is a reference point, not a bound of this implementation"}
        except Exception as e:
            out["roofline_iteration"] = {"error": repr(e)}
        if world == 1:
            # The exported solver -- surfh_cg, what QuadCriterion_MRS.run_method('lcg') and INTEGRATION.md's qmm.lcg replacement
            # run (fusion_CT.py:194-225) -- outside the timed region: two calls of different length from host buffers; the
            # difference of their wall times over the difference of their iteration counts leaves out the setup they share
            # (copy of y to the device, b = mu A^T y, copy of x back).  tol = 0: no early stop.
            try:
                yh = y.cpu().numpy()
                n1, n2 = 20, 20 + max(50, min(args.steps, 400))
                ts = []
                for nn in (n1, n2, n1, n2):
                    t0c = time.perf_counter()
                    _, gnc, nitc = m.cg(yh, mu=mu, mu_reg=mu_reg, max_iter=nn, tol=0.0)
                    ts.append(time.perf_counter() - t0c)
                    assert nitc == nn
                dt = min(ts[1], ts[3]) - min(ts[0], ts[2])
                out["surfh_cg_it_s"] = (n2 - n1) / dt
                out["surfh_cg_note"] = (f"C-ABI surfh_cg (host buffers in and out) with max_iter {n1} and {n2}: ({n2} - {n1}) iterations / "
                                        f"({min(ts[1], ts[3]):.3f} s - {min(ts[0], ts[2]):.3f} s); ratio to `value`: {(n2 - n1) / dt / (args.steps / el):.3f}")
                log(f"[surfh_cg] {out['surfh_cg_it_s']:.1f} it/s through the C-ABI solver ({out['surfh_cg_note']})")
            except Exception as e:
                out["surfh_cg_it_s"] = None
                out["surfh_cg_note"] = repr(e)
        if world == 1 and args.cpu_seconds > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
            except Exception as e:   # the baseline leg must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
