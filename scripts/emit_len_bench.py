"""Emit rate by fragment length, fast path on and off (hashes/s make lengths comparable).
Run on the GPU box:  python scripts/emit_len_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as amd  # noqa: E402

rng = np.random.default_rng(3)
k = 64
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "..", "tests", "golden", os.environ.get("OPAL_PATTERN", "patterns_k64_t16_m1_s7.txt"))) as f:
    pos = amd.parse_patterns(f.readlines(), k)
eng = amd.Engine(0)
eng.set_patterns(pos, k)
import sys
for L in ([int(a) for a in sys.argv[1:]] or [100, 200, 250, 300, 512, 600]):
    n = 800_000_000 // L * 2 // len(pos)
    batch = eng.upload(amd.PackedReads.from_codes(rng.integers(0, 4, size=(n, L), dtype=np.uint8)))
    ref = None
    for fast in (tuple(int(x) if x.isdigit() else x == "True" for x in os.environ["OPAL_KT"].split(",")) if os.environ.get("OPAL_KT") else (True, False)):   # 2 / 3: the per-base-term experiment
        eng.set_tuning(emit_fast=fast)
        for _ in range(12):                      # the first dozen launches run below the settled clock
            idx, off = eng.features(batch, True, 31)
            del idx, off
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            idx, off = eng.features(batch, True, 31)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        if ref is None:
            ref = idx.clone()
        assert torch.equal(ref, idx), (L, fast)
        print("L=%3d fast=%-5s %6.2f ms  %7.1f M fragments/s  %6.1f G hashes/s" % (L, fast, dt * 1e3, n / dt / 1e6, idx.numel() / dt / 1e9), flush=True)
        del idx, off
eng.set_tuning()
