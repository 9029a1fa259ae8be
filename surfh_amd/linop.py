"""Minimal linear-operator base and dot-test with the surface the reference uses from
``aljabr`` (aljabr 0.4.0 is not vendored by the reference; in-tree restatement of the
dot-test: test/sandbox_dottest.py:16-27)."""
from __future__ import annotations

import numpy as np


class LinOp:
    def __init__(self, ishape, oshape, name: str = "_", dtype=np.float64):
        self.ishape = tuple(int(s) for s in ishape)
        self.oshape = tuple(int(s) for s in oshape)
        self.name = name
        self.dtype = dtype

    @property
    def isize(self) -> int:
        return int(np.prod(self.ishape))

    @property
    def osize(self) -> int:
        return int(np.prod(self.oshape))

    def forward(self, x):
        raise NotImplementedError

    def adjoint(self, y):
        raise NotImplementedError

    def fwadj(self, x):
        return self.adjoint(self.forward(x))

    def matvec(self, x):
        return np.asarray(self.forward(np.reshape(x, self.ishape))).ravel()

    def rmatvec(self, y):
        return np.asarray(self.adjoint(np.reshape(y, self.oshape))).ravel()

    def __repr__(self):
        return f"{type(self).__name__}(ishape={self.ishape}, oshape={self.oshape})"


def dotgap(linop, rng=None):
    """(<A^T u, v>, <u, A v>) with fp64-accumulated inner products."""
    rng = np.random.default_rng() if rng is None else rng
    v = rng.standard_normal(linop.isize)
    u = rng.standard_normal(linop.osize)
    left = np.vdot(np.asarray(linop.rmatvec(u), dtype=np.float64), v)
    right = np.vdot(u, np.asarray(linop.matvec(v), dtype=np.float64))
    return float(left), float(right)


def dottest(linop, num: int = 1, rtol: float = 1e-5, atol: float = 1e-8, echo: bool = False, rng=None) -> bool:
    ok = True
    for _ in range(num):
        left, right = dotgap(linop, rng)
        if echo:
            print(f"(A^H u)^H v = {left} ~ {right} = u^H (A v)")
        ok = ok and bool(np.allclose(left, right, rtol=rtol, atol=atol))
    return ok
