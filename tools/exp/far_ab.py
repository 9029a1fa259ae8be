"""A/B of the far K-step class of the spectral-blur GEMMs on config 3: run twice (SURFH_WBLUR_FAR=0 / 1), then compare.
    python tools/exp/far_ab.py save out0.npz ; SURFH_WBLUR_FAR=0 python tools/exp/far_ab.py save out1.npz ; python tools/exp/far_ab.py cmp out0.npz out1.npz"""
import sys
import numpy as np

if sys.argv[1] == "save":
    from surfh_amd import synth
    from surfh_amd.models import spectroSigRLSCT
    p = synth.config3()
    m = spectroSigRLSCT(p["sotf"], p["templates"], p["alpha_axis"], p["beta_axis"], p["wavel"], p["ifus"], p["step_deg"],
                        p["pointings"], device=0, with_ref=False)
    print("ksteps (near, far) forward, (near, far) adjoint:", m.debug_buffer("ksteps"))
    rng = np.random.default_rng(0)
    out = {}
    for name, gen in (("pos", rng.random), ("randn", rng.standard_normal)):
        x = gen(m.ishape).astype(np.float32)
        u = gen(m.osize).astype(np.float32)
        out["y_" + name] = np.asarray(m.forward(x))
        out["a_" + name] = np.asarray(m.adjoint(u))
    np.savez(sys.argv[2], **out)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = a[k].astype(np.float64) - b[k].astype(np.float64)
        print(f"{k:10s} rel L2 {np.linalg.norm(d) / np.linalg.norm(b[k].astype(np.float64)):.3e}   max|d| / max|ref| {np.abs(d).max() / np.abs(b[k]).max():.3e}")
