"""The 2-D transforms of the hot path against numpy's float64 FFT, through the Fourier-domain model
``Model_WCT`` (forward = irfft2(sotf * rfft2(sum_t spec[t,l] maps[t])), reference
surfh/Models/mixing.py:102-270; kernels: surfh/ToolsDir/jax_utils.py:30-41 dft / idft).

Sizes on both sides of the limits of the two-piece fp16 passes (dft_h2.hip: 16 < N/2+1 <= 128, matrices
resident in LDS, interleaved complex arrays), of the Cooley-Tukey passes for longer axes (dft_ct.hip: N = R x M with
R = 2, 3, 4 -- 501 = 3 x 167, 512 = 4 x 128, 302 = 2 x 151, mixed factors on the two axes, one axis on each kernel: 300 x 64,
501 x 256) and of the dense fp32 fallback for axes neither covers (257 is prime, 20 too short; planar arrays), even and odd
lengths, rectangular images; inputs built to stress the per-column block
exponent of the fp16 split (hot pixels, spectra spanning 30 decades, rows that grow towards the
centre so that the accumulators are rescaled k-step after k-step).  Tolerances are fp32-level.
"""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 3e-6      # relative L2, fp32 storage + fp32 accumulation (gate of the path: 1e-5)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))


def np_forward(sotf, specs, maps):
    cube_in = np.einsum("tl,tab->lab", specs, maps)
    return np.fft.irfft2(sotf * np.fft.rfft2(cube_in), s=maps.shape[-2:])


def np_adjoint(sotf, specs, cube):
    z = np.fft.irfft2(np.conj(sotf) * np.fft.rfft2(cube), s=cube.shape[-2:])
    return np.einsum("tl,lab->tab", specs, z)


def build(shape, L, T, rng, psf_sigma=None):
    from surfh_amd.mixing import Model_WCT
    from surfh_amd.synth import ir2fr
    hs, ws = min(15, shape[0]), min(13, shape[1])
    yy, xx = np.mgrid[0:hs, 0:ws]
    if psf_sigma is None:
        psf_sigma = np.linspace(1.0, 3.0, L)
    psf_sigma = np.broadcast_to(np.asarray(psf_sigma, dtype=np.float64), (L,))
    psfs = np.exp(-((yy - hs // 2) ** 2 + (xx - ws // 2) ** 2)[None] / (2.0 * psf_sigma[:, None, None] ** 2))
    psfs = psfs * (1.0 + 0.1 * rng.standard_normal(psfs.shape))          # not symmetric: a complex OTF
    psfs /= psfs.sum(axis=(1, 2), keepdims=True)
    specs = rng.random((T, L)) + 0.1
    pce = 0.5 + rng.random(L)
    m = Model_WCT(psfs, specs, shape, pce)
    sotf = ir2fr(psfs * pce[:, None, None], shape)
    return m, sotf, specs


@pytest.mark.parametrize("shape", [(48, 48), (64, 64), (100, 100), (33, 254), (255, 40), (96, 130), (131, 77), (251, 251), (256, 256), (300, 64),
                                   (501, 501), (512, 512), (300, 300), (501, 256), (384, 510), (258, 570), (302, 302),
                                   (257, 64), (20, 48), (255, 501)])
def test_transforms_vs_numpy(shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    L, T = 130, 3
    m, sotf, specs = build(shape, L, T, rng)
    maps = rng.random((T,) + shape)
    cube = rng.standard_normal((L,) + shape)
    ef = rel(m.forward(maps), np_forward(sotf, specs, maps))
    ea = rel(m.adjoint(cube), np_adjoint(sotf, specs, cube))
    m.close()
    print(f"transforms {shape}: forward {ef:.2e} adjoint {ea:.2e}")
    assert ef < TOL and ea < TOL, (shape, ef, ea)


@pytest.mark.parametrize("shape", [(64, 64), (251, 251), (120, 200)])
def test_block_exponent_dynamic_range(shape):
    """Hot pixels 1e7 above the background, OTFs from almost flat to a Gaussian that falls by 30 decades, planes whose
    levels differ by 1e12 inside one wave tile."""
    rng = np.random.default_rng(5)
    L, T = 64, 4
    m, sotf, specs = build(shape, L, T, rng, psf_sigma=np.geomspace(0.4, 6.0, L))
    specs = specs * np.where(np.arange(L) % 3 == 0, 1e6, 1e-6)[None, :]       # neighbouring planes 1e12 apart
    m.close()
    from surfh_amd.mixing import Model_WCT
    from surfh_amd.synth import ir2fr
    # rebuild with the scaled spectra (the plan keeps its own copy)
    hs, ws = min(15, shape[0]), min(13, shape[1])
    yy, xx = np.mgrid[0:hs, 0:ws]
    sig = np.geomspace(0.4, 6.0, L)
    psfs = np.exp(-((yy - hs // 2) ** 2 + (xx - ws // 2) ** 2)[None] / (2.0 * sig[:, None, None] ** 2))
    psfs /= psfs.sum(axis=(1, 2), keepdims=True)
    pce = np.ones(L)
    m = Model_WCT(psfs, specs, shape, pce)
    sotf = ir2fr(psfs, shape)
    maps = rng.random((T,) + shape) * 1e-3
    maps[0, shape[0] // 2, shape[1] // 3] = 1e4
    maps[2, 3, shape[1] - 2] = 3e3
    y = m.forward(maps)
    yr = np_forward(sotf, specs, maps)
    ef = rel(y, yr)
    # plane by plane: no plane may lose its precision to a brighter neighbour
    pe = np.array([np.linalg.norm(y[l] - yr[l]) / np.linalg.norm(yr[l]) for l in range(L)])
    cube = rng.standard_normal((L,) + shape) * np.where(np.arange(L) % 5 == 0, 1e5, 1e-4)[:, None, None]
    cube[7, shape[0] // 2, shape[1] // 2] = 1e9
    ea = rel(m.adjoint(cube), np_adjoint(sotf, specs, cube))
    m.close()
    print(f"dynamic range {shape}: forward {ef:.2e} (worst plane {pe.max():.2e}) adjoint {ea:.2e}")
    assert ef < TOL and pe.max() < 1e-5 and ea < TOL, (ef, pe.max(), ea)


@pytest.mark.parametrize("shape", [(251, 251), (128, 64)])
def test_rows_growing_towards_the_centre(shape):
    """The folded passes walk a column from both ends inwards: magnitudes that double every few rows towards the centre
    force the running block exponent down (and the accumulators to be rescaled) at almost every k-step."""
    rng = np.random.default_rng(11)
    L, T = 32, 2
    m, sotf, specs = build(shape, L, T, rng, psf_sigma=1.5)
    ra = np.minimum(np.arange(shape[0]), shape[0] - np.arange(shape[0]))
    rb = np.minimum(np.arange(shape[1]), shape[1] - np.arange(shape[1]))
    grow = np.exp2(ra[:, None] / 4.0 + rb[None, :] / 4.0)                   # up to 2^62 at 251 x 251
    cube = rng.standard_normal((L,) + shape) * grow[None]
    ea = rel(m.adjoint(cube), np_adjoint(sotf, specs, cube))
    maps = rng.random((T,) + shape) * grow[None]
    ef = rel(m.forward(maps), np_forward(sotf, specs, maps))
    m.close()
    print(f"growing rows {shape}: forward {ef:.2e} adjoint {ea:.2e}")
    assert ef < TOL and ea < TOL, (ef, ea)


@pytest.mark.parametrize("shape,L,T", [((251, 251), 300, 4), ((251, 251), 64, 1), ((128, 64), 130, 3), ((127, 250), 40, 2),
                                       ((255, 40), 200, 4), ((200, 131), 129, 4)])
def test_fused_adjoint_tail(shape, L, T, monkeypatch):
    """The adjoint's last transform pass with the conj(OTF) product and the wavelength reduction in its epilogue
    (dft_h2_adjmix_kernel: reference spectroModel.py:175-181 / mixing.py:177-212) against numpy float64 and against the
    separate pass + reduction kernel (SURFH_ADJ_FUSED=0): first and last admissible row counts (127, 255), a row count
    equal to its padding (128: the unused mirror of row 0 lies behind the array), one and several super-tiles per
    k_beta, workgroup pairs that hold one, two or no super-tile, 1 to 4 templates, wavelength padding inside a tile."""
    rng = np.random.default_rng(shape[0] * 7 + L)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SURFH_ADJ_FUSED", fused)
        rs = np.random.default_rng(5)
        m, sotf, specs = build(shape, L, T, rs)
        cube = rng.standard_normal((L,) + shape) if not out else cube
        out[fused] = np.asarray(m.adjoint(cube), dtype=np.float64)
        if fused == "1":
            ref = np_adjoint(sotf, specs, cube)
        m.close()
    e1, e0, d = rel(out["1"], ref), rel(out["0"], ref), rel(out["1"], out["0"])
    print(f"fused adjoint tail {shape} L={L} T={T}: fused {e1:.2e} separate {e0:.2e} fused vs separate {d:.2e}")
    assert e1 < TOL and e0 < TOL and d < TOL, (e1, e0, d)


@pytest.mark.parametrize("shape,L,sig", [((251, 251), 300, (2.0, 4.5)), ((128, 200), 260, (3.0, 3.0)), ((251, 130), 140, (5.0, 2.5)),
                                         ((501, 300), 260, (2.5, 6.0)), ((512, 512), 140, (7.0, 3.0))])
def test_otf_support_lists(shape, L, sig):
    """Band-limited OTFs (clean Gaussians, widths changing along the wavelength axis): the forward's complex pass and the fused
    adjoint tail visit only the (k_beta, 128-wavelength chunk) super-tiles in which the OTF reaches 2^-24 of its plane's peak
    (plan.hip otf_support) -- chunks with different cutoffs, k_beta values without any tile, a cutoff that rises with the
    wavelength.  Both directions against numpy float64.  The last two shapes run the Cooley-Tukey passes (dft_ct.hip), where the
    lists also limit the rows the adjoint's passes store and the bins its reduction reads."""
    import ctypes
    from surfh_amd.mixing import Model_WCT
    from surfh_amd.synth import ir2fr
    rng = np.random.default_rng(L)
    T = 3
    hs, ws = 81, 81                      # 40 pixels = 9 sigma of the widest PSF: no truncation floor in the OTF
    yy, xx = np.mgrid[0:hs, 0:ws]
    s = np.linspace(sig[0], sig[1], L)
    psfs = np.exp(-((yy - hs // 2) ** 2 + (xx - ws // 2) ** 2)[None] / (2.0 * s[:, None, None] ** 2))
    psfs /= psfs.sum(axis=(1, 2), keepdims=True)
    specs = rng.random((T, L)) + 0.1
    pce = 0.5 + rng.random(L)
    sotf = ir2fr(psfs * pce[:, None, None], shape)
    maps = rng.random((T,) + shape)
    cube = rng.standard_normal((L,) + shape)
    m = Model_WCT(psfs, specs, shape, pce)
    dims = (ctypes.c_int64 * 4)()
    m._L.surfh_debug_dims(m._plan, b"otf", dims)
    yf, ya = np.asarray(m.forward(maps)), np.asarray(m.adjoint(cube))
    m.close()
    ef, ea = rel(yf, np_forward(sotf, specs, maps)), rel(ya, np_adjoint(sotf, specs, cube))
    print(f"otf support {shape} L={L}: {dims[0]} of {dims[1]} super-tiles, forward {ef:.2e} adjoint {ea:.2e}")
    assert 0 < dims[0] < dims[1], list(dims)
    assert ef < TOL and ea < TOL, (ef, ea)
