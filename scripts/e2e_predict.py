#!/usr/bin/env python3
"""End-to-end timing of `opal.py predict` on a synthetic 1M-fragment FASTA (GPU box only, scratch files in /tmp)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opal_amd import opal, model as model_io

n, L, K, bits = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 200, 356, 31
d = tempfile.mkdtemp(prefix="opal_e2e_", dir="/tmp")
rng = np.random.default_rng(0)
t0 = time.time()
codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
os.makedirs(d + "/test"); os.makedirs(d + "/model")
with open(d + "/test/reads.fasta", "wb") as f:
    hdr = np.char.add(np.char.add(">", np.arange(1, n + 1).astype(str)), "\n").astype("S")
    for s in range(0, n, 100000):
        e = min(n, s + 100000)
        f.write(b"".join(h + q.tobytes() + b"\n" for h, q in zip(hdr[s:e].tolist(), seqs[s:e])))
with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "patterns_k64_t16_m1_s7.txt")) as f:
    open(d + "/model/patterns.txt", "w").write(f.read())
open(d + "/model/vw-dico.txt", "w").write("".join("tax%d\t%d\n" % (i, i) for i in range(1, K + 1)))
idx = rng.integers(0, 1 << bits, size=5_000_000).astype(np.uint32)
val = (rng.standard_normal(idx.size) * 0.05).astype(np.float32)
with open(d + "/model/vw-model_final.model", "wb") as f:
    f.write(model_io._HDR.pack(model_io.MAGIC, 1, bits, K, 6, -1.0, 1.0, idx.size, 0, 0)); idx.tofile(f); val.tofile(f)
print("setup %.1fs, fasta %.0f MB" % (time.time() - t0, os.path.getsize(d + "/test/reads.fasta") / 1e6), flush=True)
import argparse, cProfile, pstats
args = argparse.Namespace(kmer=64, reverse_complement=True)
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
opal.predict(d + "/model", d + "/test", d + "/pred", args)
pr.disable()
print("predict wall %.2fs -> %.0f fragments/s end to end" % (time.time() - t0, n / (time.time() - t0)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
import shutil; shutil.rmtree(d)
