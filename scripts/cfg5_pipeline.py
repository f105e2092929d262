"""BASELINE.json config 5 end to end on one MI355X: training-feature generation for one batch of the 1564-genome set
(1564 x 4 Mbp resident in HBM, coverage 1 -> 31.3 M fragments of 200 bp): CPython-exact sampling on the host, views
built on the device, hashed features emitted in the reference's order.  Prints the time of every stage.
Run on the GPU box:  python scripts/cfg5_pipeline.py [n_genomes] [genome_len]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as amd  # noqa: E402
from opal_amd._lib import check, lib  # noqa: E402
from opal_amd.engine import DeviceBatch  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1564
GL = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
L = 200
SUB = int(os.environ.get("OPAL_CFG5_SUB", "0")) or G * (GL // L + 1)     # fragments per emit launch: by default the whole batch in one
eng = amd.Engine(0)
dev = eng.device
with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "patterns_k64_t16_m1_s7.txt")) as f:
    eng.set_patterns(amd.parse_patterns(f.readlines(), 64), 64)
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
data = torch.randint(0, 256, (G * GL // 4,), dtype=torch.uint8, device=dev, generator=gen)   # 4 uniform bases per byte
h_off = np.arange(G, dtype=np.uint64) * np.uint64(GL)
h_len = np.full(G, GL, dtype=np.uint32)
gs = amd.GenomeSet(eng, amd.PackedReads(None, h_off, h_len),
                   DeviceBatch(data, torch.from_numpy(h_off.view(np.int64)).to(dev), torch.from_numpy(h_len.view(np.int32)).to(dev),
                               G, int(data.numel())))
rng = amd.Rng(421)


def timed(name, fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-34s %8.1f ms  %8.1f M fragments/s" % (name, dt * 1e3, n / dt / 1e6), flush=True)
    return out, dt


total = 0.0
(pos, gi), dt = timed("sample positions (host, 1 core)", lambda: rng.sample_positions([GL] * G, L, 1.0), G * GL // L)
n = len(pos)
total += dt
batch, dt = timed("upload 12 B/fragment + views", lambda: gs.fragments(pos, gi, L), n)
total += dt
idx = None
t_emit = 0.0
for s in range(0, n, SUB):
    e = min(n, s + SUB)
    sub = DeviceBatch(batch.data, batch.off[s:e], batch.len[s:e], e - s, batch.nbytes)
    offsets = eng.feature_offsets(sub, True)
    tot = int(offsets[-1].item())
    if idx is None:
        idx = torch.empty(min(SUB, n) * 548, dtype=torch.int32, device=dev)
    def emit():
        check(lib().opal_features(eng._h, sub.data.data_ptr(), sub.nbytes, sub.off.data_ptr(), sub.len.data_ptr(), sub.n, 1, 31,
                                  offsets.data_ptr(), idx.data_ptr(), tot), eng._h)
    _, dt = timed("  emit fragments %d..%d" % (s, e), emit, e - s)
    t_emit += dt
print("%-34s %8.1f ms  %8.1f M fragments/s" % ("emit, all sub-batches", t_emit * 1e3, n / t_emit / 1e6))
total += t_emit
print("%-34s %8.1f ms  %8.1f M fragments/s  (%.2e hashes/s)" % ("whole batch", total * 1e3, n / total / 1e6, n * 548 / total))

# the whole of config 5: 15 batches at coverage 1 (readme.txt:116-118 recommends -c 1 --num-batches 15; seeds 421+i as
# opal.py:337-344), the host sampling batch i+1 in a thread while the GPU builds the views of batch i and emits it
import queue
import threading

NB = int(sys.argv[3]) if len(sys.argv) > 3 else 15
q = queue.Queue(maxsize=1)


def producer():
    for b in range(NB):
        r = amd.Rng(421 + b)
        q.put(r.sample_positions([GL] * G, L, 1.0))
    q.put(None)


torch.cuda.synchronize()
t0 = time.perf_counter()
threading.Thread(target=producer, daemon=True).start()
done = 0
while True:
    item = q.get()
    if item is None:
        break
    pos, gi = item
    batch = gs.fragments(pos, gi, L)
    for s in range(0, len(pos), SUB):
        e = min(len(pos), s + SUB)
        sub = DeviceBatch(batch.data, batch.off[s:e], batch.len[s:e], e - s, batch.nbytes)
        offsets = eng.feature_offsets(sub, True)
        check(lib().opal_features(eng._h, sub.data.data_ptr(), sub.nbytes, sub.off.data_ptr(), sub.len.data_ptr(), sub.n, 1, 31,
                                  offsets.data_ptr(), idx.data_ptr(), (e - s) * 548), eng._h)
    done += len(pos)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d batches, sampling overlapped: %d fragments in %.2f s = %.1f M fragments/s  (%.2e hashes/s)" % (
    NB, done, dt, done / dt / 1e6, done * 548 / dt), flush=True)
