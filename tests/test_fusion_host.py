"""Host logic of the multi-GPU driver: unit partition, synthetic problems, criterion mirror."""
import numpy as np

from surfh_amd import synth
from surfh_amd.fusion import partition_units


def test_partition_whole_bands():
    costs = [67.5, 85.9, 113.1, 140.9]
    a = partition_units(costs, [4, 4, 4, 4], 2)
    assert sorted(k for r in a for k, _ in r) == [0, 1, 2, 3]
    loads = [sum(costs[k] for k, _ in r) for r in a]
    assert max(loads) / min(loads) < 1.15
    a4 = partition_units(costs, [4] * 4, 4)
    assert all(len(r) == 1 and r[0][1] == [0, 1, 2, 3] for r in a4)
    a1 = partition_units(costs, [4] * 4, 1)
    assert [k for k, _ in a1[0]] == [0, 1, 2, 3]


def test_partition_splits_pointings_beyond_band_count():
    a = partition_units([1.0, 2.0, 3.0, 4.0], [4] * 4, 8)
    assert all(len(r) == 1 for r in a)
    for k in range(4):
        sel = sorted(i for r in a for kk, s in r if kk == k for i in s)
        assert sel == [0, 1, 2, 3]                     # every pointing of every band owned exactly once
    a6 = partition_units([1.0, 2.0, 3.0, 4.0], [4] * 4, 6)
    assert sum(1 for r in a6 for kk, _ in r if kk == 3) == 2 and sum(1 for r in a6 for kk, _ in r if kk == 0) == 1


def test_config_shapes():
    p = synth.config2(lam_stride=64)
    assert p["wavel"].shape == (16,) and p["sotf"].shape == (16, 251, 126) and p["templates"].shape == (4, 16)
    assert [i.name for i in synth.config3(lam_stride=500)["ifus"]] == ["1C", "2A", "2B", "2C"]
    assert len(synth.BANDS) == 12 and synth.band_wavelengths("2a").shape == (970,)


def test_band_wavelength_tables_are_the_reference_tables():
    """synth.band_wavelengths (what scripts/main_fusion.py and the benchmark problems use) == the reference's
    global_variables.wavelength_<band>, bit for bit (golden file written from the imported reference)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bands_geometry.npz"))
    for b in synth.BANDS:
        assert np.array_equal(synth.band_wavelengths(b), g[b + "_wavel"]), b
    p3 = synth.config3(lam_stride=500)
    lo, hi, n = g["axis_cfg3"]
    assert np.array_equal(np.linspace(lo, hi, int(n))[::500], p3["wavel"])


def test_partition_choice(monkeypatch):
    """the assignment the multi-GPU driver will use for 2, 4 and 8 ranks on config 3 (4 bands) and on config 4
    (12 bands, npix 501): every cube plane of every band owned exactly once; config 3 on 4 ranks is one band per GPU (north_star);
    a band is split only where the predicted time of the slowest rank, the group-local all-reduce of the split band's partial
    outputs included, is lower than without the split; the compute-only rule (SURFH_PARTITION=balanced) stays within 15 %."""
    from surfh_amd.fusion import plan_assignment
    p3 = synth.config3(geometry_only=True)
    p4 = synth.config4(n_pix=501, geometry_only=True)
    for prob, worlds in ((p3, (2, 4, 8)), (p4, (2, 4, 8))):
        for world in worlds:
            monkeypatch.setenv("SURFH_PARTITION", "balanced")
            asg_b, loads_b, imb_b, times_b = plan_assignment(prob, world, with_times=True)
            monkeypatch.delenv("SURFH_PARTITION")
            asg, loads, imb, times = plan_assignment(prob, world, with_times=True)
            print(len(prob["ifus"]), "bands on", world, "ranks: loads", [round(v) for v in loads], f"compute imbalance {imb:.3f}, slowest rank "
                  f"{max(times):.0f} us (compute-balanced chunks: {max(times_b):.0f} us, imbalance {imb_b:.3f})")
            assert len(asg) == world and all(len(r) > 0 for r in asg)
            assert imb_b <= 0.15, (world, loads_b)
            assert max(times) <= max(times_b) + 1e-9
            if prob is p3 and world == 4:
                assert sorted(r[0] for r in asg) == [(0, (0, 1)), (1, (0, 1)), (2, (0, 1)), (3, (0, 1))]
            # coverage: the lambda parts of every band tile its window exactly
            from surfh_amd.geometry import ChannelGeometry
            from surfh_amd import instru
            for k in range(len(prob["ifus"])):
                parts = [u for r in asg for kk, u in r if kk == k]
                assert parts, k
                if parts == [(0, 1)]:
                    continue
                if len(parts[0]) == 3:        # ("planes", a, b)
                    iv = sorted((u[1], u[2]) for u in parts)
                    assert iv[0][0] == 0 and all(iv[i][1] == iv[i + 1][0] for i in range(len(iv) - 1))
                else:                          # (i, n) equal parts
                    assert sorted(u[0] for u in parts) == list(range(parts[0][1]))


def test_checkpoint_file_round_trip(tmp_path):
    """save_checkpoint / load_checkpoint: one .npz written through a temporary file, loadable without pickle."""
    from surfh_amd.fusion import load_checkpoint, save_checkpoint
    x = np.random.default_rng(0).random((4, 8, 8)).astype(np.float32)
    p = save_checkpoint(tmp_path / "ck", x, 7, [3.0, 2.0, 1.0])
    assert p.endswith("ck.npz") and not (tmp_path / "ck.tmp.npz").exists()
    x2, it, gn = load_checkpoint(tmp_path / "ck")
    assert it == 7 and x2.dtype == np.float64 and np.array_equal(x2, x.astype(np.float64)) and list(gn) == [3.0, 2.0, 1.0]
    save_checkpoint(tmp_path / "ck.npz", x * 2, 8, [1.0])                 # overwrites in place
    assert load_checkpoint(tmp_path / "ck.npz")[1] == 8
