"""Host-side invariants of the K-step classes of the spectral-blur GEMMs (plan.hip build_klist, through the host-only C-ABI hook
surfh_klist_classify -- no GPU): every K step of a tile is in exactly one class, both lists ascend, and for EVERY row of a tile
the far steps together hold at most 2^-8 of the row's l1 norm and 2^-10 of its l2 norm (what bounds the error of keeping only the
leading fp16 product there); an operand without structure has no far step; the adjoint's tile shape (64 rows of
four neighbouring columns) covers every row once."""
import ctypes

import numpy as np


def classify(B, perm_p=0, perm_lin=0):
    from surfh_amd import _lib
    L = _lib.load()
    n, k = B.shape
    B = np.ascontiguousarray(B, dtype=np.float32)
    nb = k // 32
    cap = (n // 64 + 8) * (2 + nb)
    rec = np.zeros(cap, dtype=np.int32)
    nt = L.surfh_klist_classify(_lib.fptr(B), n, k, k, perm_p, perm_lin, rec.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), cap)
    assert nt > 0, nt
    return rec[: nt * (2 + nb)].reshape(nt, 2 + nb)


def response(n_out, lin, ncol):
    lo, li = np.arange(n_out)[:, None], np.arange(lin)[None, :]
    cols = []
    for c in range(ncol):
        w = np.sinc((li - (lo * (lin / n_out) + 3.0 * c)) / 2.3) ** 2
        cols.append(w / w.sum(axis=1, keepdims=True))
    return np.concatenate(cols, axis=1)          # [n_out][ncol * lin]


def check(B, rec, rows_of_tile):
    nb = B.shape[1] // 32
    a = np.abs(B.astype(np.float64)).reshape(B.shape[0], nb, 32)
    l1, l2 = a.sum(2), (a * a).sum(2)
    for t, r in enumerate(rec):
        nn, nf = int(r[0]), int(r[1])
        near, far = r[2: 2 + nn] & 0xFFFF, r[2 + nn: 2 + nn + nf] & 0xFFFF
        assert nn + nf == nb and np.all(np.diff(near) > 0) and np.all(np.diff(far) > 0)
        assert sorted(np.concatenate([near, far]).tolist()) == list(range(nb))
        rows = rows_of_tile(t)
        rows = rows[rows < B.shape[0]]
        if nf:
            f1 = l1[np.ix_(rows, far)].sum(1)
            f2 = l2[np.ix_(rows, far)].sum(1)
            assert np.all(f1 <= l1[rows].sum(1) / 256 * (1 + 1e-12)), t
            assert np.all(f2 <= l2[rows].sum(1) / 1024 ** 2 * (1 + 1e-12)), t
    return int(rec[:, 0].sum()), int(rec[:, 1].sum())


def test_forward_shape_classes():
    W = response(1024, 1152, 3)                   # [lambda'][beta column, lambda]
    rec = classify(W)
    assert rec.shape[0] == 4
    near, far = check(W, rec, lambda t: np.arange(256 * t, 256 * t + 256))
    assert far > near > 0


def test_adjoint_tiles_of_four_columns():
    lin, ncol = 1152, 5
    Wt = np.ascontiguousarray(response(1024, lin, ncol).T)      # [(beta column, lambda)][lambda']
    rec = classify(Wt, perm_p=4, perm_lin=lin)
    tiles_l = lin // 64
    assert rec.shape[0] == 2 * tiles_l          # column groups (0-3), (4) x chunks of 64 wavelengths

    def rows(t):
        g, lc = divmod(t, tiles_l)
        cols = [c for c in range(4 * g, 4 * g + 4) if c < ncol]
        return np.concatenate([np.arange(c * lin + 64 * lc, c * lin + 64 * lc + 64) for c in cols])

    seen = np.concatenate([rows(t) for t in range(rec.shape[0])])
    assert sorted(seen.tolist()) == list(range(ncol * lin))
    near, far = check(Wt, rec, rows)
    plain = classify(Wt)
    assert far > near and rec[:, 1].sum() > plain[:, 1].sum()       # rows of the same wavelengths share their near steps


def test_operand_without_structure_has_no_far_step():
    B = np.random.default_rng(0).standard_normal((384, 1056))
    rec = classify(B)
    assert rec[:, 1].sum() == 0 and np.all(rec[:, 0] == 33)
