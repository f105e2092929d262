"""Predict on ragged fragment lengths vs. a fixed-length batch with the same number of windows (load balance check).
Run on the GPU box:  python scripts/ragged_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as amd  # noqa: E402

rng = np.random.default_rng(3)
K, bits, k = 356, 31, 64
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "..", "tests", "golden", "patterns_k64_t16_m1_s7.txt")) as f:
    pos = amd.parse_patterns(f.readlines(), k)
eng = amd.Engine(0)
eng.set_patterns(pos, k)
w = torch.empty(1 << bits, dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
eng.load_weights(w, K, bits)


def batch_of(lens):
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total, dtype=np.uint8)
    pad = np.zeros((total + 3) // 4 * 4, dtype=np.uint8)
    pad[:total] = codes
    q = pad.reshape(-1, 4)
    data = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)          # base offsets into one packed stream
    return eng.upload(amd.PackedReads(data, off, lens.astype(np.uint32)))


def run(name, lens):
    b = batch_of(lens)
    windows = int(np.maximum(lens - k + 1, 0).sum())
    eng.predict(b, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.predict(b, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {len(lens)} fragments, {windows} windows, {dt * 1e3:.1f} ms = {windows / dt / 1e6:.1f} M windows/s", flush=True)


n = 400_000
run("fixed 332", np.full(n, 332))
run("uniform 64..600", rng.integers(64, 601, size=n))
run("bimodal 100/1500", np.where(rng.random(n) < 0.85, 100, 1500))
run("sorted uniform 64..600", np.sort(rng.integers(64, 601, size=n)))
