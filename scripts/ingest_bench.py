"""Times opal_ingest_fasta on a synthetic FASTA held in HBM (n 200-bp records, optional multi-line genomes).
Run on the GPU box:  python scripts/ingest_bench.py [n_records] [reads|genomes|both]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as amd  # noqa: E402
from opal_amd._lib import check, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(5)


def reads_fasta(n, L=200):
    rec = np.empty((n, 9 + 1 + L + 1), dtype=np.uint8)          # ">%08d\n" + L bases + "\n"
    rec[:, 0] = ord(">")
    ids = np.arange(n)
    for d in range(8):
        rec[:, 8 - d] = ord("0") + (ids // 10 ** d) % 10
    rec[:, 9] = ord("\n")
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    rec[:, 10:10 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    rec[:, 10 + L] = ord("\n")
    return rec.reshape(-1), codes


def genome_fasta(n_genomes, length, width=70):
    parts = []
    for g in range(n_genomes):
        body = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=length, dtype=np.uint8)]
        lines = np.full((length // width, width + 1), ord("\n"), dtype=np.uint8)
        lines[:, :width] = body[:(length // width) * width].reshape(-1, width)
        parts += [np.frombuffer(b">genome%d some description\n" % g, dtype=np.uint8), lines.reshape(-1)]
    return np.concatenate(parts)


eng = amd.Engine(0)
which = sys.argv[2] if len(sys.argv) > 2 else "both"
cases = []
if which in ("both", "reads"):
    cases.append(("reads (%d x 200 bp)" % n,) + reads_fasta(n))
if which in ("both", "genomes"):
    cases.append(("genomes (64 x 4 Mbp, 70-column lines)", genome_fasta(64, 4_000_000), None))
for name, arr, codes in cases:
    d_text = torch.from_numpy(arr).to(eng.device)
    nbytes = int(arr.size)
    max_rec = int((d_text == 62).sum().item()) + 1
    reads = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    off = torch.empty(max_rec, dtype=torch.int64, device=eng.device)
    ln = torch.empty(max_rec, dtype=torch.int32, device=eng.device)
    nrec, nb = C.c_uint64(), C.c_uint64()

    def run():
        check(lib().opal_ingest_fasta(eng._h, d_text.data_ptr(), nbytes, reads.data_ptr(), nbytes, off.data_ptr(), ln.data_ptr(),
                                      max_rec, C.byref(nrec), C.byref(nb)), eng._h)
    run()
    torch.cuda.synchronize()
    if codes is not None:
        # every record is a 200-base view into the packed stream: the stream is the batch from_codes builds
        want = amd.PackedReads.from_codes(codes)
        assert nrec.value == n and np.array_equal(off.cpu().numpy()[:n].view(np.uint64), want.off)
        assert bool((ln[:n] == 200).all()) and np.array_equal(reads.cpu().numpy()[:want.data.size], want.data)
        print("verified: all %d records decode to the generated reads" % n, flush=True)
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {nbytes / 1e6:.0f} MB of text -> {nrec.value} records, {nb.value / 1e6:.0f} MB packed in {dt * 1e3:.2f} ms "
          f"= {nbytes / dt / 1e9:.1f} GB/s of text", flush=True)
