"""Times the emit kernel on fragments that take the generic (raw text) path: every fragment carries one 'N'.
Run on the GPU box:  python scripts/raw_path_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as amd  # noqa: E402

rng = np.random.default_rng(3)
n, L = 200_000, 200
codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
text = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
raw = text.copy()
raw[:, 100] = ord("N")
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "..", "tests", "golden", "patterns_k64_t16_m1_s7.txt")) as f:
    pos, k = amd.parse_patterns(f.readlines(), 64), 64
eng = amd.Engine(0)
eng.set_patterns(pos, k)
for name, arr in (("packed", text), ("raw", raw)):
    batch = eng.upload(amd.PackedReads.from_sequences([bytes(r) for r in arr]))
    eng.features(batch, True, 31)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.features(batch, True, 31)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.2f} ms per {n} fragments = {n / dt / 1e6:.1f} M fragments/s", flush=True)
