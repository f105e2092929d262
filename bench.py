#!/usr/bin/env python3
"""Benchmark of the Opal hot path on MI355X: fused fragment -> LDPC spaced-k-mer hashes -> OAA scores -> class.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the fused predict kernel over one batch of synthetic fragments that is already resident in
HBM.  Default workload = BASELINE.json configs[1]: 356 classes, 1M fragments of 200 bp, k=64, row_weight=16,
numHash=1, both strands (-r), 2^31-slot weight table.  Fragments shard across ranks with no collective
(SURVEY.md §8e): every rank scores its own batch against its own replica of the table ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / mean launch time measured with
HIP events on the launch stream; `cpu_baseline` = the CPU oracle (oracle/, a bit-exact C port) timed on this
box's host cores on a bounded sample of the same batch (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (K, n_frag per GPU, numHash pattern file, bits, mode)
    "cfg1_toy10": dict(K=10, n=10_000, pattern="patterns_k64_t16_m1_s7.txt", bits=18, mode="predict",
                       desc="10 classes, 1e4 x 200bp, numHash=1, b=18 (plumbing)"),
    "cfg2_356sp_1M": dict(K=356, n=1_000_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="predict",
                          desc="356 classes, 1M x 200bp fragments, k=64 t=16 numHash=1, -r, b=31"),
    "cfg3_1564sp": dict(K=1564, n=500_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="predict",
                        desc="1564 classes, 200bp fragments per GPU shard, numHash=1, -r, b=31"),
    "cfg4_numhash4": dict(K=1564, n=200_000, pattern="patterns_k64_t16_m4_s7.txt", bits=31, mode="predict",
                          desc="1564 classes, numHash=4 (P=5, F=1370), -r, b=31"),
    "cfg5_emit": dict(K=1564, n=4_000_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="emit",
                      desc="training-feature generation: sample 200bp fragments from 1564 HBM-resident genomes (zero-copy views) "
                           "and emit their hashed features, numHash=1, -r"),
    "train_356sp": dict(K=356, n=200_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=True,
                        desc="native OAA learner, worst case: first pass from a zeroed table (every class has a non-zero error, "
                             "nothing is skipped) over 200k labelled 200bp fragments, 356 classes, numHash=1, -r, b=31"),
    "train_356sp_warm": dict(K=356, n=200_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=False,
                             desc="native OAA learner, later passes over the same 200k fragments: classes already at their "
                                  "label after the clip are skipped, 356 classes, numHash=1, -r, b=31"),
    "ingest_reads": dict(K=1, n=4_000_000, pattern="patterns_k64_t16_m1_s7.txt", bits=18, mode="ingest",
                         desc="GPU FASTA ingest (SURVEY.md 8f-3): 4M records of 200 bp as FASTA text resident in HBM -> "
                              "2-bit packed fragment batch (opal_ingest_scan + opal_ingest_fasta)"),
    "train_1564sp": dict(K=1564, n=50_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=True,
                         desc="native OAA learner, worst case (zeroed table each step): 50k labelled 200bp fragments, 1564 classes"),
}


def make_genomes_and_draws(rank, n, L, K, glen):
    """Synthetic genomes (uniform ACGT, PCG64 seed 1234) + n fragment draws (position, genome) from the CPython-exact
    sampler (seed 42+rank) through the C-ABI, like `opal.py frag` (reference opal.py:187-196), in shuffled order
    (reads arrive in no particular order; opal.py:363 shuffles training batches with the same stream)."""
    import opal_amd
    n_genomes = max(K, 8)
    g = np.random.Generator(np.random.PCG64(1234)).integers(0, 4, size=(n_genomes, glen), dtype=np.uint8)
    cov = (n * L) / float(n_genomes * glen) * 1.02 + 1e-4
    rng = opal_amd.Rng(42 + rank)
    pos, gen = rng.sample_positions([glen] * n_genomes, L, cov)
    assert len(pos) >= n, (len(pos), n)
    sel = rng.shuffle_perm(len(pos))[:n].astype(np.int64)
    return g, pos[sel].astype(np.int64), gen[sel].astype(np.int64)


def make_fragments(rank, n, L, K):
    """Explicit fragment codes [n, L] cut from low-coverage genomes (a metagenomic sample covers its genomes thinly, so
    weight rows are not re-used between fragments)."""
    g, pos, gen = make_genomes_and_draws(rank, n, L, K, 1_000_000)
    codes = np.empty((n, L), dtype=np.uint8)
    step = 100_000
    ar = np.arange(L, dtype=np.int64)
    for s in range(0, n, step):
        e = min(n, s + step)
        codes[s:e] = g[gen[s:e, None], pos[s:e, None] + ar[None, :]]
    return codes


def algorithmic_bytes_per_fragment(L, F, K, mode):
    if mode == "emit":
        return math.ceil(L / 4) + 4 * F                      # 2-bit read + one u32 per feature
    if mode == "ingest":
        return (1 + 8 + 1 + L + 1) + math.ceil(L / 4) + 12   # the record's text once + its packed bases + (offset, length)
    if mode == "train":
        # sweep 1 reads w rows; sweep 2 reads+writes G rows; sweep 3 reads G rows, reads+writes w rows (opal_train.hip)
        return math.ceil(L / 4) + 6 * 4 * K * (F + 1) + 4
    return math.ceil(L / 4) + 4 * K * (F + 1) + 4             # 2-bit read + (F+1) rows of K floats + class id


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2_356sp_1M", choices=sorted(WORKLOADS))
    ap.add_argument("--nfrag", type=int, default=0, help="override fragments per GPU")
    ap.add_argument("--bits", type=int, default=0, help="override table bits")
    ap.add_argument("--classes", type=int, default=0, help="override the class count K (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=300_000)
    args = ap.parse_args()

    import torch
    import opal_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        # OPAL_DIST_BACKEND=gloo is for rehearsing the N>1 path on a box with fewer GPUs than ranks (ranks then share
        # devices round-robin); the driver's runs use nccl (= RCCL), one rank per GPU
        backend = os.environ.get("OPAL_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    wl = dict(WORKLOADS[args.workload])
    if args.nfrag:
        wl["n"] = args.nfrag
    if args.bits:
        wl["bits"] = args.bits
    if args.classes:
        wl["K"] = args.classes
    K, n, bits, mode = wl["K"], wl["n"], wl["bits"], wl["mode"]
    L, k = 200, 64
    with open(os.path.join(ROOT, "tests", "golden", wl["pattern"])) as f:
        pos = opal_amd.parse_patterns(f.readlines(), k)
    P = pos.shape[0]
    F = (L - k + 1) * P * 2

    dev = torch.device("cuda", local_rank)
    eng = opal_amd.Engine(local_rank)
    eng.set_patterns(pos, k)
    w = None
    if mode == "predict":
        gen = torch.Generator(device=dev)
        gen.manual_seed(99)
        w = torch.empty(1 << bits, dtype=torch.float32, device=dev)
        w.normal_(0.0, 0.05, generator=gen)           # fp32 N(0, 0.05^2) dense table, same on every rank
        eng.load_weights(w, K, bits)
        eng.set_vw_options()                            # VW defaults: --hash strings, constant feature, clip to [-1,1]
    labels = None
    if mode == "predict":
        codes = make_fragments(rank, n, L, max(K, 1))
        batch = eng.upload(opal_amd.PackedReads.from_codes(codes))
    elif mode == "ingest":
        # ">%08d\n" + 200 bases + "\n" per record, uniform ACGT (PCG64 seed 1234 + rank)
        codes = None
        rec = np.empty((n, 10 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord(">")
        ids = np.arange(n)
        for d in range(8):
            rec[:, 8 - d] = ord("0") + (ids // 10 ** d) % 10
        rec[:, 9] = rec[:, 10 + L] = ord("\n")
        rec[:, 10:10 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[
            np.random.Generator(np.random.PCG64(1234 + rank)).integers(0, 4, size=(n, L), dtype=np.uint8)]
        d_text = torch.from_numpy(rec.reshape(-1)).to(dev)
        nbytes = int(d_text.numel())
        ing_reads = torch.empty(nbytes // 4 + 4096, dtype=torch.uint8, device=dev)
        ing_off = torch.empty(n, dtype=torch.int64, device=dev)
        ing_len = torch.empty(n, dtype=torch.int32, device=dev)
    else:
        # genomes packed once and resident in HBM; a batch is 12 bytes of (offset, length) per fragment
        g, gpos, ggen = make_genomes_and_draws(rank, n, L, max(K, 1), 200_000)
        gs = opal_amd.GenomeSet(eng, opal_amd.PackedReads.from_codes(g))
        batch = gs.fragments(gpos, ggen, L)
        codes = None
        labels = torch.from_numpy((ggen % K + 1).astype(np.int32)).to(dev)
    out_class = torch.empty(n, dtype=torch.int32, device=dev)
    pre_step = None
    if mode == "train":
        eng.set_vw_options()
        eng.train_begin(K, bits)
        if wl.get("cold"):
            pre_step = lambda: eng.train_begin(K, bits)   # re-zero w and G outside the timed events

        def step():
            eng.train_batch(batch, labels, True, eta=0.5)
    elif mode == "ingest":
        import ctypes as C
        from opal_amd._lib import check, lib
        c_a, c_b = C.c_uint64(), C.c_uint64()

        def step():
            check(lib().opal_ingest_scan(eng._h, d_text.data_ptr(), nbytes, C.byref(c_a), C.byref(c_b)), eng._h)
            check(lib().opal_ingest_fasta(eng._h, d_text.data_ptr(), nbytes, ing_reads.data_ptr(), ing_reads.numel(), ing_off.data_ptr(),
                                          ing_len.data_ptr(), n, C.byref(c_a), C.byref(c_b)), eng._h)
            assert c_a.value == n
    elif mode == "emit":
        offsets = eng.feature_offsets(batch, True)
        total = int(offsets[-1].item())
        out_idx = torch.empty(total, dtype=torch.int32, device=dev)
        from opal_amd._lib import check, lib

        def step():
            check(lib().opal_features(eng._h, batch.data.data_ptr(), batch.nbytes, batch.off.data_ptr(), batch.len.data_ptr(),
                                      batch.n, 1, bits, offsets.data_ptr(), out_idx.data_ptr(), total), eng._h)
    else:
        def step():
            eng.predict(batch, True, want_scores=False, out_class=out_class)

    for _ in range(args.warmup):
        if pre_step:
            pre_step()
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    t_pre = 0.0
    for a, b in ev:
        if pre_step:                                   # (learner worst case only) untimed re-zeroing between steps
            torch.cuda.synchronize(dev)
            tp = time.perf_counter()
            pre_step()
            torch.cuda.synchronize(dev)
            t_pre += time.perf_counter() - tp
        a.record()      # torch's current stream IS the ctx stream (Engine binds to it), so these bracket the kernel
        step()
        b.record()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0 - t_pre
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank == 0:
        frag_per_s = world * n * args.steps / elapsed
        abytes = algorithmic_bytes_per_fragment(L, F, K, mode) * n
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic, prof = None, {}
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            with open(tp) as f:
                prof = json.load(f).get(args.workload if not (args.nfrag or args.bits or args.classes) else "", {})
                traffic = prof.get("hbm_bytes_per_launch")
        res = {
            "metric": "fragments/sec", "value": frag_per_s, "unit": "fragments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8" if mode == "ingest" else "u32" if mode == "emit" else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "desc": wl["desc"], "classes": K, "fragments_per_gpu": n, "frag_len": L,
                       "kmer": k, "row_weight": 16, "patterns": P, "reverse_complement": True, "bits": bits,
                       "features_per_fragment": F, "parallelism": "fragment-sharded x%d, replicated table, no collective" % world},
            "hashes_per_s": None if mode == "ingest" else frag_per_s * F,
            "kernel": {"predict": eng.predict_kernel_name(), "emit": "emit_idx_kernel", "train": "train_kernel",
                       "ingest": "ing_summary_kernel + ing_scan_kernel + ing_pack_kernel + ing_records_kernel (whole call)"}[mode],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": kern_ms},
        }
        for key in ("valu_issue_frac", "lds_busy_frac"):        # kernels that are ALU/LDS-bound by construction:
            if key in prof:                                     # integer-ALU and LDS utilisation from the committed PMC passes
                res["roofline"][key] = prof[key]
        import shutil
        res["vw_binary_on_box"] = shutil.which("vw")       # time a real vw if one exists (never seen so far)
        if world == 1 and not args.no_cpu_baseline and codes is not None:
            res["cpu_baseline"] = cpu_baseline(codes, pos, k, w, K, bits, mode, min(args.cpu_sample, n), out_class)
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(codes, pos, k, w, K, bits, mode, m, out_class):
    """Times the CPU oracle (the checker, a scalar C port with OpenMP over fragments) on the first m fragments."""
    from oracle import oracle as orc
    orc.build()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L = codes.shape[1]
    blob = np.ascontiguousarray(acgt[codes[:m]].reshape(-1))
    blob = np.concatenate([blob, np.zeros(1, dtype=np.uint8)])
    soff = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
    threads = min(16, orc.max_threads())
    if mode == "emit":
        t0 = time.perf_counter()
        idx, off = orc.features_batch((blob, soff), k, pos, True, bits, nthreads=threads)
        dt = time.perf_counter() - t0
        ok = None
    else:
        wh = w.cpu().numpy()
        t0 = time.perf_counter()
        cls, _, _ = orc.predict((blob, soff), k, pos, True, wh, K, bits, want_scores=False, nthreads=threads)
        dt = time.perf_counter() - t0
        ok = bool(np.array_equal(cls, out_class[:m].cpu().numpy()))   # the timed GPU output agrees with the checker
        m1 = max(1000, m // 50)
        t1 = time.perf_counter()
        orc.predict((blob[:m1 * L + 1], soff[:m1 + 1]), k, pos, True, wh, K, bits, want_scores=False, nthreads=1)
        dt1 = time.perf_counter() - t1
    out = {"value": m / dt, "unit": "fragments/s", "cores": threads, "kind": "port",
           "sample": "first %d fragments of the same batch, oracle/opal_oracle.c with %d OpenMP threads (%.1f s)" % (m, threads, dt)}
    if mode != "emit":
        out["value_1thread"] = m1 / dt1
        out["gpu_matches_oracle_on_sample"] = ok
        out["reference_python_1core"] = "3833 frag/s feature half only (BASELINE.md, survey container); vw absent"
    return out


if __name__ == "__main__":
    main()
