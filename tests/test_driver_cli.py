"""Host logic of the fusion driver (scripts/main_fusion.py): command line, result naming, slit-data loading.
No GPU needed."""
import importlib.util
import os

import numpy as np
from click.testing import CliRunner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("main_fusion", os.path.join(ROOT, "scripts", "main_fusion.py"))
drv = importlib.util.module_from_spec(spec)
spec.loader.exec_module(drv)


def test_command_line_matches_reference_flags():
    """Same short/long flags and defaults as the reference driver (scripts/main_fusion.py:212-220)."""
    r = CliRunner().invoke(drv.main, ["--help"])
    assert r.exit_code == 0
    for flag in ("-fd, --fusion_dir", "-np, --npix", "-hp, --hyper_parameter", "-ni, --niter", "-nt, --n_templates",
                 "-sd, --scale_data", "-m, --method", "-v, --verbose"):
        assert flag in r.output, flag
    defaults = {p.name: p.default for p in drv.main.params}
    assert defaults["npix"] == 501 and defaults["hyper_parameter"] == 1.0 and defaults["niter"] == 5
    assert defaults["n_templates"] == 4 and defaults["scale_data"] is False and defaults["method"] == "lcg"


def test_result_directory_name():
    # the reference's f-string (main_fusion.py:182) for 12 channels, 4 templates, 50 iterations, mu = 5e3
    assert drv.result_dir_name("lcg", 12, 4, 50, 5e3, False) == "lcg_MC_12_MO_4_Temp_4_nit_50_mu_5.00e+03_SD_False/"
    paths, step, step_angle = drv.initialize_parameters("/data/F")
    assert paths["result_path"] == "/data/F/Results/" and step == 0.025 and abs(step_angle - 0.025 / 3600) < 1e-18


def test_load_data_from_npz(tmp_path):
    """[Ldet, S, a_out] raveled per file -> [S, Ldet, a_out] per pointing; roll angle and targets kept per band."""
    shapes = {"1a": (21, 1050, 19), "2a": (17, 970, 24)}           # (S, Ldet, a_out), main_fusion.py:34-39
    rng = np.random.default_rng(0)
    ref = {}
    for chan, (S, L, A) in shapes.items():
        for k in range(2):
            d = rng.random((L, S, A))
            ref[(chan, k)] = d
            np.savez(tmp_path / f"ch{chan}_{k}.npz", data=d.ravel(), PA_V3=250.0 + k, TARG_RA=1e-4 * k, TARG_DEC=-2e-4 * k)
    dd = drv.load_data(["1a", "2a"], str(tmp_path))
    for chan, (S, L, A) in shapes.items():
        assert len(dd["data"][chan]) == 2 and dd["data"][chan][0].shape == (S, L, A)
        assert np.array_equal(dd["data"][chan][1], ref[(chan, 1)].transpose(1, 0, 2))
        assert dd["rotation"][chan] == 251.0 and dd["target"][chan][1] == (1e-4, -2e-4)
    ifus = drv.create_instruments(dd, ["1a", "2a"])
    assert ifus["2a"].n_slit == 17 and ifus["2a"].fov.angle == -251.0 and ifus["1a"].n_wavel == 1050


def test_deconvolution_driver_command_line():
    """scripts/deconvolution_mrs.py: the reference run's parameters are the defaults
    (scripts/simulate_deconvolution_mrs_rectangle.py:183-198: lcg, 600 iterations, mu_reg 5, value_init 0)."""
    sp = importlib.util.spec_from_file_location("deconvolution_mrs", os.path.join(ROOT, "scripts", "deconvolution_mrs.py"))
    dd = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(dd)
    r = CliRunner().invoke(dd.main, ["--help"])
    assert r.exit_code == 0
    defaults = {p.name: p.default for p in dd.main.params}
    assert defaults["niter"] == 600 and defaults["hyper_parameter"] == 5.0 and defaults["method"] == "lcg" and defaults["value_init"] == 0.0
    prob = dd.build_problem(96, 1, 1, None)
    assert prob["sotf"].shape == (96, 49) and prob["truth"].shape == (96, 96) and len(prob["pointings"]) == 4
    prob3 = dd.build_problem(64, 3, 1, None)
    assert prob3["sotf"].shape == (3, 64, 33) and prob3["truth"].shape == (3, 64, 64)
