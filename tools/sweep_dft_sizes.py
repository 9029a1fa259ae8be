"""One-off robustness sweep (GPU): the 2-D transforms of Model_WCT against numpy's float64 FFT over many image sizes, so that
every factorisation class of the axis kernels (dft_h2 <= 255, dft_ct N = R M with R = 2, 3, 4, dense fp32 fallback) and their
edge conditions are visited:  python3 tools/sweep_dft_sizes.py [step]  -> one line per size, FAIL lines if above 3e-6."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_dft import build, np_adjoint, np_forward, rel      # noqa: E402

step = int(sys.argv[1]) if len(sys.argv) > 1 else 7
sizes = sorted(set(list(range(34, 640, step)) + [255, 256, 257, 258, 259, 260, 262, 264, 380, 381, 382, 383, 384, 385, 386, 387, 388, 501, 512,
                                                  566, 568, 570, 571, 572, 573, 576, 756, 758, 760, 761, 762, 764, 768]))
bad = 0
t0 = time.time()
for n in sizes:
    for shape in ((n, n), (n, 64), (48, n)):
        if shape != (n, n) and n % 3:       # the rectangular ones on a third of the sizes
            continue
        rng = np.random.default_rng(n * 1000 + shape[1])
        L, T = 128, 3
        try:
            m, sotf, specs = build(shape, L, T, rng)
            maps = rng.random((T,) + shape)
            cube = rng.standard_normal((L,) + shape)
            ef = rel(m.forward(maps), np_forward(sotf, specs, maps))
            ea = rel(m.adjoint(cube), np_adjoint(sotf, specs, cube))
            m.close()
        except Exception as e:              # noqa: BLE001
            print(f"{shape}: EXCEPTION {e!r}", flush=True)
            bad += 1
            continue
        ok = ef < 3e-6 and ea < 3e-6
        bad += 0 if ok else 1
        print(f"{shape}: forward {ef:.2e} adjoint {ea:.2e} {'ok' if ok else 'FAIL'}  [{time.time() - t0:.0f}s]", flush=True)
print(f"{bad} failure(s)")
sys.exit(1 if bad else 0)
