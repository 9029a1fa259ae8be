"""Host-side geometry of the product (surfh_amd.instru / surfh_amd.geometry) against the
golden tables produced by the real reference, and against the (independent) oracle."""
import os
import re

import numpy as np
import pytest

import problems
from helpers import make_ifu, make_pointings
from oracle import surfh_oracle as orc
from surfh_amd import instru
from surfh_amd.geometry import ChannelGeometry

G = os.path.join(os.path.dirname(__file__), "golden")


def geom(cfg, k=0):
    spec = cfg["specs"][k]
    srf = instru.get_srf([spec.det_pix_size], cfg["step_deg"] * 3600)[0]
    return ChannelGeometry(make_ifu(spec), cfg["alpha_axis"], cfg["beta_axis"], cfg["wavel"], srf,
                           make_pointings(cfg)[k], cfg["step_deg"])


def check_against_golden(ch, g, pre):
    assert ch.srf == int(g[pre + "srf"])
    assert (ch.wslice.start, ch.wslice.stop) == tuple(g[pre + "wslice"])
    assert np.array_equal(ch.local_alpha_axis, g[pre + "local_alpha_axis"])
    assert np.array_equal(ch.local_beta_axis, g[pre + "local_beta_axis"])
    assert (ch.slicer.npix_slit_alpha_width, ch.slicer.npix_slit_beta_width) == tuple(g[pre + "npix_ab"])
    assert tuple(ch.oshape) == tuple(g[pre + "oshape"])
    sl = [ch.slicer.get_slit_slices(s) for s in range(ch.instr.n_slit)]
    assert np.array_equal([[a.start, a.stop, b.start, b.stop] for a, b in sl], g[pre + "slit_slices"])
    w = [ch.slicer.get_slit_weights(s, sl[s])[0] for s in range(ch.instr.n_slit)]
    assert np.array_equal([x[0, 0] for x in w], g[pre + "slit_w_first"])
    assert np.array_equal([x[0, -1] for x in w], g[pre + "slit_w_last"])
    assert np.array_equal([[c.alpha, c.beta] for c in ch.pointings], g[pre + "pointings_pix"])
    assert np.array_equal([ch.instr.fov.origin.alpha, ch.instr.fov.origin.beta], g[pre + "origin_pix"])


def test_config1_tables_bit_exact():
    cfg = problems.config1()
    g = np.load(os.path.join(G, "config1_chain.npz"))
    ch = geom(cfg)
    check_against_golden(ch, g, "c0_")
    assert np.array_equal(ch.wpsf, g["wpsf"])
    t = ch.tables()
    for p in range(4):
        assert np.array_equal(np.stack([t["grid_i0"][p], t["grid_i1"][p]]), g[f"bil_idx_p{p}"])
        assert np.array_equal(np.stack([t["grid_y0"][p], t["grid_y1"][p]]), g[f"bil_frac_p{p}"])
    # and the oracle agrees with the product on every table
    tab = problems.oracle_model(cfg).channels[0]
    assert np.array_equal(t["slit_weights"], np.array([w[0] for w in tab.slit_weights]))
    assert list(t["slit_beta0"]) == [s[2] for s in tab.slit_slices]
    assert (t["alpha0"], t["n_alpha_slit"]) == (tab.slit_slices[0][0], tab.slit_slices[0][1] - tab.slit_slices[0][0])


def test_two_channel_tables():
    cfg = problems.two_channel_small()
    g = np.load(os.path.join(G, "two_channel.npz"))
    for k in range(2):
        check_against_golden(geom(cfg, k), g, f"c{k}_")


def test_real_bands_bit_exact():
    g = np.load(os.path.join(G, "bands_geometry.npz"))
    ax = orc.synthetic_axes(251, problems.STEP_DEG)
    lo, hi, n = g["axis_cfg4"]
    wav4 = np.linspace(lo, hi, int(n))
    for name in problems.BANDS:
        spec = problems.band_spec(name, wavel_axis=g[f"{name}_wavel"])
        cfg = dict(specs=[spec], alpha_axis=ax, beta_axis=ax, wavel=wav4, step_deg=problems.STEP_DEG,
                   pointings=[orc.dither4(spec.det_pix_size, spec.beta_width / spec.n_slit)])
        ch = geom(cfg)
        check_against_golden(ch, g, f"{name}_")
        if name in ("1a", "2a", "4c"):
            assert np.array_equal(ch.wpsf[::97, ::53, :], g[f"{name}_wpsf_sample"])
        if name == "2a":
            t = ch.tables(with_ref=False)
            for p in range(4):
                assert np.array_equal(np.stack([t["grid_i0"][p], t["grid_i1"][p]])[:, ::5], g[f"2a_bil_idx_p{p}"])
                assert np.array_equal(np.stack([t["grid_y0"][p], t["grid_y1"][p]])[:, ::5], g[f"2a_bil_frac_p{p}"])


def test_out_of_bounds_pointing_raises_like_reference():
    cfg = problems.config1()
    spec = cfg["specs"][0]
    big = orc.ChannelSpec(1.0 / 3600, 1.2 / 3600, (0.0, 0.0), 8.2, 0.196, 4, 3050.0, spec.wavel_axis, "big")
    cfg2 = dict(cfg, specs=[big])
    with pytest.raises(ValueError, match="out of bounds"):
        geom(cfg2).tables()


def test_coord_pix_bankers_rounding():
    c = instru.Coord(2.5, 3.5).pix(1.0)
    assert (c.alpha, c.beta) == (2.0, 4.0)      # instru.py:143-145 uses python round()


def test_wslice_excludes_last_plane():
    ifu = make_ifu(orc.ChannelSpec(1e-3, 1e-3, (0, 0), 0, 0.196, 2, 3000.0, np.linspace(7.0, 8.0, 10)))
    ws = ifu.wslice(np.linspace(7.0, 8.0, 24), 0.1)
    assert (ws.start, ws.stop) == (0, 23)        # SURVEY.md 7 hard part 3


def test_nn_tables_match_reference_indices():
    cfg = problems.config1()
    g = np.load(os.path.join(G, "config1_nn_indices.npz"))
    spec = cfg["specs"][0]
    ch = ChannelGeometry(make_ifu(spec), cfg["alpha_axis"], cfg["beta_axis"], cfg["wavel"], 7,
                         make_pointings(cfg)[0], cfg["step_deg"], gridding="nn_ref")
    N = 64
    for p in range(4):
        i0, i1, y0, y1 = ch.grid_tables(p)
        assert set(np.unique(y0)) <= {0.0, 1.0} and set(np.unique(y1)) <= {0.0, 1.0}
        k = (i0 + y0.astype(np.int64)) * N + (i1 + y1.astype(np.int64))       # the single selected pixel, C-order
        assert np.array_equal(k, g[f"nn_idx_p{p}"].ravel())
        j0, j1, z0, z1, inside = ch.gridt_tables(p)
        kt = (j0 + z0.astype(np.int64)) * len(ch.local_beta_axis) + (j1 + z1.astype(np.int64))
        assert np.array_equal(kt, g[f"nn_idx_t_p{p}"].ravel()) and inside.all()
