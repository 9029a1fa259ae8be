"""Import the REAL reference hot path from /root/reference  --  TEST INFRASTRUCTURE ONLY.

Works only in the build container (``/root/reference`` does not exist on the GPU
box).  Used by ``tests/golden/make_golden.py`` to generate the golden vectors that
``tests/test_oracle_golden.py`` holds ``oracle/surfh_oracle.py`` to, stage by stage.
Nothing is copied: the reference's modules are imported from where they lie, and its
one native file (``surfh/ToolsDir/cythons_files.pyx``) is cythonized + compiled from
that path with outputs only under a scratch directory OUTSIDE the repository
(``$SURFH_REF_BUILD`` or ``<tmp>/surfh_ref_build``): nothing compiled from the reference
is committed or travels to the GPU box.

The snapshot is mid-refactor and depends on packages that are not installed
(SURVEY.md 8c).  What this harness supplies so that the surviving modules import:

* aliases for modules renamed in the refactor:
    surfh.Models.slicer_new                           -> surfh.Models.slicer
    surfh.DottestModels.MCMO_SigRLSCT_Channel_Model   -> surfh.Models.spectroModelChannel
* ``surfh.ToolsDir.jax_utils`` mapped onto the reference's OWN SciPy/NumPy twins
  in ``surfh/ToolsDir/python_utils.py`` (float64 instead of JAX float32); the
  reference asserts python == jax == cython in test/test_accel_accuracy.py:17-57,252-379.
  ``wblur_subSampling(a, w) := python_utils.wblur(a, w).sum(axis=2)``.
* restated third-party pieces (NOT reference code, flagged in fixture metadata):
    udft.ir2fr   (udft 3.4.0)   -- oracle.surfh_oracle.ir2fr
    udft.dft2 / idft2 / rdft2 / irdftn -- numpy FFTs with norm="ortho" (only for surfh/Models/mixing.py)
    aljabr.LinOp (aljabr 0.4.0) -- minimal ishape/oshape/matvec/rmatvec holder
* empty stubs for packages that are imported but unused on the path:
    jax, astropy, loguru, xarray, numba, matplotlib(Agg), einops is real.
"""
from __future__ import annotations

import importlib
import os
import subprocess
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
import tempfile  # noqa: E402

OUT = os.environ.get("SURFH_REF_BUILD") or os.path.join(tempfile.gettempdir(), "surfh_ref_build")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "surfh"))


def build_cython(force=False) -> str:
    """cythonize + gcc the reference's cythons_files.pyx into the scratch directory OUT (build.py:12-13 flags)."""
    os.makedirs(OUT, exist_ok=True)
    import sysconfig
    so = os.path.join(OUT, "cythons_files" + sysconfig.get_config_var("EXT_SUFFIX"))
    src = os.path.join(REF, "surfh/ToolsDir/cythons_files.pyx")
    if os.path.exists(so) and not force and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    c_file = os.path.join(OUT, "cythons_files.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", src, "-o", c_file])
    inc = sysconfig.get_paths()["include"]
    subprocess.check_call(
        ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-w",
         "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
         "-I", inc, "-I", np.get_include(), c_file, "-o", so])
    os.remove(c_file)
    return so


class _Anything(types.ModuleType):
    """Stub module: any attribute is another stub / a no-op decorator."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(self.__name__ + "." + name)
        sys.modules.setdefault(sub.__name__, sub)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return self


def _stub(name):
    m = _Anything(name)
    sys.modules[name] = m
    return m


def load():
    """Return a namespace with the imported reference modules."""
    if not available():
        raise RuntimeError("reference tree not present (only in the build container)")
    if "surfh_ref_ns" in sys.modules:
        return sys.modules["surfh_ref_ns"]
    so = build_cython()
    import matplotlib
    matplotlib.use("Agg")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import surfh_oracle as orc

    for name in ("qmm", "surfh.Models.spectro", "surfh.Models.spectrolmm", "jax", "jax.numpy", "jax.lax", "astropy", "astropy.units", "astropy.coordinates",
                 "astropy.io", "astropy.io.fits", "loguru", "xarray", "numba", "progressbar",
                 "SharedArray"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["jax"].numpy = sys.modules["jax.numpy"]
    sys.modules["jax"].lax = sys.modules["jax.lax"]
    sys.modules["jax"].jit = lambda f=None, **k: (f if f is not None else (lambda g: g))
    sys.modules["loguru"].logger = types.SimpleNamespace(
        info=lambda *a, **k: None, debug=lambda *a, **k: None, warning=lambda *a, **k: None)

    udft = types.ModuleType("udft")
    udft.ir2fr = orc.ir2fr          # restated third-party (udft 3.4.0)
    # udft's unitary transforms (used only by surfh/Models/mixing.py), restated: numpy FFTs with norm="ortho"
    udft.dft2 = lambda x: np.fft.fft2(x, norm="ortho")
    udft.idft2 = lambda x: np.fft.ifft2(x, norm="ortho")
    udft.rdft2 = lambda x: np.fft.rfft2(x, norm="ortho")
    udft.irdftn = lambda x, shape: np.fft.irfftn(x, s=tuple(shape), axes=tuple(range(-len(shape), 0)), norm="ortho")
    # only imported by surfh/ToolsDir/fusion_mixing.py (the explicit-inverse solver): the discrete Laplacian impulse
    # response of udft 3.4.0 restated [memory] -- centre 2*ndim, -1 on the axis neighbours
    def _laplacian(ndim):
        ir = np.zeros((3,) * ndim)
        c = (1,) * ndim
        ir[c] = 2.0 * ndim
        for ax in range(ndim):
            for d in (0, 2):
                i = list(c)
                i[ax] = d
                ir[tuple(i)] = -1.0
        return ir
    udft.laplacian = _laplacian
    udft.diff_ir = lambda ndim, axis: np.reshape(np.array([0.0, -1.0, 1.0]), [3 if a == axis % ndim else 1 for a in range(ndim)])
    sys.modules["udft"] = udft

    aljabr = types.ModuleType("aljabr")

    class LinOp:                     # restated third-party (aljabr 0.4.0), minimal
        def __init__(self, ishape, oshape, name="_", dtype=np.float64):
            self.ishape, self.oshape, self.name, self.dtype = tuple(ishape), tuple(oshape), name, dtype

        @property
        def isize(self):
            return int(np.prod(self.ishape))

        @property
        def osize(self):
            return int(np.prod(self.oshape))

        def matvec(self, x):
            return np.asarray(self.forward(np.reshape(x, self.ishape))).ravel()

        def rmatvec(self, y):
            return np.asarray(self.adjoint(np.reshape(y, self.oshape))).ravel()

    aljabr.LinOp = LinOp
    aljabr.dottest = lambda *a, **k: None
    sys.modules["aljabr"] = aljabr
    aljabr_linop = types.ModuleType("aljabr.linop")
    aljabr_linop.Shape = tuple
    sys.modules["aljabr.linop"] = aljabr_linop

    # the reference's compiled native module, loaded from the scratch build directory
    import importlib.util
    spec = importlib.util.spec_from_file_location("surfh.ToolsDir.cythons_files", so)
    cyf = importlib.util.module_from_spec(spec)
    import surfh.ToolsDir  # noqa: F401  (real package from /root/reference)
    sys.modules["surfh.ToolsDir.cythons_files"] = cyf
    spec.loader.exec_module(cyf)
    sys.modules["surfh.ToolsDir"].cythons_files = cyf

    python_utils = importlib.import_module("surfh.ToolsDir.python_utils")

    jx = types.ModuleType("surfh.ToolsDir.jax_utils")   # jax twins -> the reference's own scipy twins
    jx.lmm_maps2cube = python_utils.lmm_maps2cube
    jx.lmm_cube2maps = python_utils.lmm_cube2maps
    jx.dft = python_utils.dft
    jx.idft = python_utils.idft
    jx.dft_mult = lambda a, b: python_utils.dft(a) * b
    jx.wblur = python_utils.wblur
    jx.wblur_t = python_utils.wblur_t
    jx.wblur_subSampling = lambda a, w: python_utils.wblur(a, w).sum(axis=2)
    sys.modules["surfh.ToolsDir.jax_utils"] = jx
    sys.modules["surfh.ToolsDir"].jax_utils = jx

    mo = types.ModuleType("surfh.ToolsDir.matrix_op")   # numba duplicates, unused on the path
    sys.modules["surfh.ToolsDir.matrix_op"] = mo
    sys.modules["surfh.ToolsDir"].matrix_op = mo

    instru = importlib.import_module("surfh.Models.instru")
    slicer = importlib.import_module("surfh.Models.slicer")
    sys.modules["surfh.Models.slicer_new"] = slicer
    sys.modules["surfh.Models"].slicer_new = slicer
    cython_utils = importlib.import_module("surfh.ToolsDir.cython_utils")
    nn = importlib.import_module("surfh.ToolsDir.nearest_neighbor_interpolation")
    chan = importlib.import_module("surfh.Models.spectroModelChannel")
    dm = types.ModuleType("surfh.DottestModels")
    dm.MCMO_SigRLSCT_Channel_Model = chan
    sys.modules["surfh.DottestModels"] = dm
    sys.modules["surfh.DottestModels.MCMO_SigRLSCT_Channel_Model"] = chan
    model = importlib.import_module("surfh.Models.spectroModel")
    gv = importlib.import_module("surfh.Others.global_variables")

    ns = types.ModuleType("surfh_ref_ns")
    ns.instru, ns.slicer, ns.python_utils, ns.cython_utils = instru, slicer, python_utils, cython_utils
    ns.cythons_files, ns.nn, ns.channel, ns.model, ns.global_variables = cyf, nn, chan, model, gv
    ns.fusion_mixing = lambda: importlib.import_module("surfh.ToolsDir.fusion_mixing")   # QuadCriterion3 (explicit inverse)
    ns.mixing = lambda: importlib.import_module("surfh.Models.mixing")   # lazy: pulls algorithms.py + shared-memory helpers
    sys.modules["surfh_ref_ns"] = ns
    return ns


def make_ifu(ns, spec):
    """Build a reference ``instru.IFU`` from an oracle ChannelSpec."""
    I = ns.instru
    return I.IFU(fov=I.FOV(spec.alpha_width, spec.beta_width,
                           origin=I.Coord(spec.origin[0], spec.origin[1]), angle=spec.angle),
                 det_pix_size=spec.det_pix_size, n_slit=spec.n_slit,
                 w_blur=I.SpectralBlur(spec.grating_resolution), pce=None,
                 wavel_axis=spec.wavel_axis, name=spec.name)
