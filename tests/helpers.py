"""Build the product model (surfh_amd) from a tests/problems.py config."""
import numpy as np

from surfh_amd import instru
from surfh_amd.models import spectroSigRLSCT


def make_ifu(spec):
    return instru.IFU(fov=instru.FOV(spec.alpha_width, spec.beta_width,
                                     origin=instru.Coord(spec.origin[0], spec.origin[1]), angle=spec.angle),
                      det_pix_size=spec.det_pix_size, n_slit=spec.n_slit,
                      w_blur=instru.SpectralBlur(spec.grating_resolution), pce=None,
                      wavel_axis=spec.wavel_axis, name=spec.name)


def make_pointings(cfg):
    return [instru.CoordList([instru.Coord(a, b) for a, b in pts]) for pts in cfg["pointings"]]


def build_model(cfg, **kw):
    return spectroSigRLSCT(cfg["sotf"], cfg["templates"], cfg["alpha_axis"], cfg["beta_axis"], cfg["wavel"],
                           [make_ifu(s) for s in cfg["specs"]], cfg["step_deg"], make_pointings(cfg), **kw)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) /
                 np.linalg.norm(np.asarray(b, dtype=np.float64)))
