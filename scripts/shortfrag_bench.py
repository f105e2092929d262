"""Predict rate by fragment length at the reference CLI's default geometry (-k 64 --num-hash 8; -l/--frag-length defaults to 64:
one window per strand, 18 features + the constant per fragment) and at numHash 1.  Same table and class counts as bench.py.
Run on the GPU box:  python scripts/shortfrag_bench.py [K ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as amd  # noqa: E402
from opal_amd.engine import DeviceBatch  # noqa: E402

dev = torch.device("cuda", 0)
eng = amd.Engine(0)
eng.set_vw_options()
bits = 31
gen = torch.Generator(device=dev)
gen.manual_seed(99)
table = torch.empty((1 << bits) + 4096, dtype=torch.float32, device=dev)
table[:1 << bits].normal_(0.0, 0.05, generator=gen)
here = os.path.dirname(os.path.abspath(__file__))
for K in ([int(a) for a in sys.argv[1:]] or [356, 1564]):
    eng.adopt_weights(table, K, bits)
    for pat in ("patterns_k64_t16_m8_s7.txt", "patterns_k64_t16_m1_s7.txt"):
        with open(os.path.join(here, "..", "tests", "golden", pat)) as f:
            pos = amd.parse_patterns(f.readlines(), 64)
        eng.set_patterns(pos, 64)
        P = pos.shape[0]
        for L in (64, 100, 200):
            F = (L - 63) * P * 2
            n = max(200_000, min(20_000_000, int(3e9 / ((F + 1) * K))))          # ~12 GB of rows per launch
            nb = (L + 3) // 4
            data = torch.randint(0, 256, (n * nb,), dtype=torch.uint8, device=dev, generator=gen)
            batch = DeviceBatch(data, torch.arange(n, dtype=torch.int64, device=dev) * (nb * 4), torch.full((n,), L, dtype=torch.int32, device=dev), n, n * nb)
            out = torch.empty(n, dtype=torch.int32, device=dev)
            for _ in range(2):
                eng.predict(batch, True, out_class=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.predict(batch, True, out_class=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            gbs = n * (nb + 4 * K * (F + 1) + 4) / dt / 1e9
            print("K=%4d P=%d L=%3d  %9d fragments  %8.2f ms  %8.2f M fragments/s  %6.0f GB/s = %4.1f %% of 8 TB/s" % (
                K, P, L, n, dt * 1e3, n / dt / 1e6, gbs, gbs / 80), flush=True)
            del data, batch, out
