#!/usr/bin/env python3
"""Benchmark of the Opal hot path on MI355X: fused fragment -> LDPC spaced-k-mer hashes -> OAA scores -> class.

    python bench.py --gpus N --steps K --warmup W            (N>1: one rank per GPU, launched by torch.distributed.run -- by the
                                                              caller, or by this script itself when it finds no launcher)

A "step" is one pass of the fused predict kernel over one batch of synthetic fragments that is already resident in
HBM.  Headline workload (`value`, every N) = BASELINE.json configs[1]: 356 classes, 1M fragments of 200 bp per GPU,
k=64, row_weight=16, numHash=1, both strands (-r), 2^31-slot weight table.  Fragments shard across ranks with no
collective: every rank scores its own batch against its own replica of the table ("weak" scaling).

The same run also times the other BASELINE configs with a few steps each and reports them under `configs` in the same
JSON line (max over ranks, like the headline):
  cfg3_1564sp    configs[2]: K=1564, 12.5 M fragments per GPU = the per-GPU share of "100 M fragments over 8 GPUs"
                 (at --gpus 8 the ranks together score exactly that job in one step)
  cfg4_numhash4  configs[3]: K=1564, numHash=4 (P=5, F=1370)
  cfg5_emit      configs[4]: training-feature generation at its stated size: 1564 genomes x 4 Mbp resident in HBM, one batch at
                 coverage 1 (31.28 M fragments drawn by the CPython-exact sampler as views, shuffled) -> hashed features, one launch
Fragments are generated on the device from a seed (uniform ACGT: a 2-bit packed fragment is 50 random bytes), so set-up
takes seconds at any size.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / mean launch time measured with
HIP events on the launch stream; `cpu_baseline` = the CPU oracle (oracle/, a bit-exact C port) timed on this
box's host cores on a bounded sample of the headline batch (rank 0, N=1 only).
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
    "cfg3_1564sp": dict(K=1564, n=12_500_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="predict",
                        desc="1564 classes, 12.5M x 200bp fragments per GPU (= 100M over 8 GPUs), numHash=1, -r, b=31"),
    "cfg4_numhash4": dict(K=1564, n=1_000_000, pattern="patterns_k64_t16_m4_s7.txt", bits=31, mode="predict",
                          desc="1564 classes, numHash=4 (P=5, F=1370), 1M x 200bp fragments, -r, b=31"),
    "numhash8_default": dict(K=1564, n=500_000, pattern="patterns_k64_t16_m8_s7.txt", bits=31, mode="predict",
                             desc="the reference CLI's default --num-hash 8 (P=9, F=2466): 1564 classes, 500k x 200bp fragments, -r, b=31 "
                                  "(not a BASELINE config; --workload only)"),
    "cfg5_emit": dict(K=1564, n=31_280_000, genomes=1564, genome_len=4_000_000, coverage=1.0,
                      pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="emit",
                      desc="training-feature generation, one batch at coverage 1 (readme.txt:116-118: -c 1 --num-batches 15): 1564 genomes x 4 Mbp "
                           "resident in HBM (1.56 GB packed), 31.28 M 200bp fragments drawn by the CPython-exact sampler as zero-copy views, "
                           "shuffled (opal.py:363), hashed features emitted by ONE launch into one 68.6 GB index buffer, numHash=1, -r"),
    "train_356sp": dict(K=356, n=200_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=True,
                        desc="native OAA learner, worst case: first pass from a zeroed table (every class has a non-zero error, "
                             "nothing is skipped) over 200k labelled 200bp fragments, 356 classes, numHash=1, -r, b=31"),
    "train_356sp_warm": dict(K=356, n=200_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=False,
                             desc="native OAA learner, later passes over the same 200k fragments: classes already at their "
                                  "label after the clip are skipped, 356 classes, numHash=1, -r, b=31"),
    "train_356sp_det": dict(K=356, n=200_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=True, tmode=2,
                            desc="native OAA learner, deterministic mode (groups of 8192 examples, slot-range owners), first pass from a "
                                 "zeroed table over 200k labelled 200bp fragments, 356 classes, numHash=1, -r, b=31"),
    "train_356sp_det_warm": dict(K=356, n=200_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=False, tmode=2,
                                 desc="native OAA learner, deterministic mode, later passes over the same 200k fragments"),
    "train_1564sp_det": dict(K=1564, n=50_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=True, tmode=2,
                             desc="native OAA learner, deterministic mode, worst case (zeroed table each step): 50k fragments, 1564 classes"),
    "ingest_reads": dict(K=1, n=4_000_000, pattern="patterns_k64_t16_m1_s7.txt", bits=18, mode="ingest",
                         desc="GPU FASTA ingest (SURVEY.md 8f-3): 4M records of 200 bp as FASTA text resident in HBM -> "
                              "2-bit packed fragment batch (opal_ingest_fasta: one pass over the text)"),
    "train_1564sp": dict(K=1564, n=50_000, pattern="patterns_k64_t16_m1_s7.txt", bits=31, mode="train", cold=True,
                         desc="native OAA learner, worst case (zeroed table each step): 50k labelled 200bp fragments, 1564 classes"),
}
# what the default run times besides the headline: (workload, steps, warmup), bounded so the whole run stays within minutes
EXTRA = (("cfg3_1564sp", 2, 1), ("cfg4_numhash4", 3, 1), ("cfg5_emit", 20, 20))   # (4 ms steps: the first dozen run below the settled clock)
L_FRAG, KMER = 200, 64


def algorithmic_bytes_per_fragment(L, F, K, mode):
    if mode == "emit":
        return math.ceil(L / 4) + 4 * F                      # 2-bit read + one u32 per feature
    if mode == "ingest":
        return (1 + 8 + 1 + L + 1) + math.ceil(L / 4) + 12   # the record's text once + its packed bases + (offset, length)
    if mode == "train":
        # sweep 1 reads w rows; sweep 2 reads+writes G rows; sweep 3 reads G rows, reads+writes w rows (opal_train.hip)
        return math.ceil(L / 4) + 6 * 4 * K * (F + 1) + 4
    return math.ceil(L / 4) + 4 * K * (F + 1) + 4             # 2-bit read + (F+1) rows of K floats + class id


class Bench:
    """One rank's engine, its weight table (one copy in HBM, shared by every predict workload) and the process group."""

    def __init__(self, args):
        import torch
        import opal_amd
        self.torch, self.amd, self.args = torch, opal_amd, args
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:      # under a launcher the launcher's world size is the truth (main() starts one otherwise)
            args.gpus = self.world
        self.dist = None
        if self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # launched by torch.distributed.run (also with one rank)
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
            self.dist = dist
        self.dev = torch.device("cuda", local_rank)
        self.eng = opal_amd.Engine(local_rank)
        self.eng.set_vw_options()                      # VW defaults: --hash strings, constant feature, clip to [-1,1]
        self.table, self.table_bits = None, 0

    def weights(self, bits):
        """fp32 N(0, 0.05^2) dense table of 2^bits floats, seed 99, same on every rank, generated on the
        device with room for the row-wrap mirror of the largest K, so that the engine scores it in place (one copy)."""
        t = self.torch
        if self.table is None or self.table_bits != bits:
            self.table = None
            gen = t.Generator(device=self.dev)
            gen.manual_seed(99)
            self.table = t.empty((1 << bits) + 4096, dtype=t.float32, device=self.dev)
            self.table[:1 << bits].normal_(0.0, 0.05, generator=gen)
            self.table_bits = bits
        return self.table

    def device_fragments(self, n, L, seed):
        """n fragments of L uniform ACGT bases, 2-bit packed (L/4 random bytes each), generated on the device."""
        t = self.torch
        nb = (L + 3) // 4
        gen = t.Generator(device=self.dev)
        gen.manual_seed(seed)
        data = t.randint(0, 256, (n * nb,), dtype=t.uint8, device=self.dev, generator=gen)
        off = t.arange(n, dtype=t.int64, device=self.dev) * (nb * 4)          # packed fragments are addressed in bases
        ln = t.full((n,), L, dtype=t.int32, device=self.dev)
        from opal_amd.engine import DeviceBatch
        return DeviceBatch(data, off, ln, n, n * nb)

    def device_genomes(self, n_genomes, glen, seed):
        """n_genomes uniform ACGT genomes of glen bases, packed and resident in HBM (generated there)."""
        from opal_amd.engine import DeviceBatch
        b = self.device_fragments(n_genomes, glen, seed)
        host = self.amd.PackedReads(None, np.arange(n_genomes, dtype=np.uint64) * np.uint64(glen), np.full(n_genomes, glen, dtype=np.uint32))
        return self.amd.GenomeSet(self.eng, host, DeviceBatch(b.data, b.off, b.len, n_genomes, b.nbytes))

    def views(self, gs, n, L, glen, n_genomes):
        """n fragment draws (position, genome) from the CPython-exact sampler (seed 42+rank) through the C-ABI, like
        `opal.py frag` (reference opal.py:187-196), in shuffled order (opal.py:363 shuffles with the same stream)."""
        cov = (n * L) / float(n_genomes * glen) * 1.02 + 1e-4
        rng = self.amd.Rng(42 + self.rank)
        pos, gen = rng.sample_positions([glen] * n_genomes, L, cov)
        assert len(pos) >= n, (len(pos), n)
        sel = rng.shuffle_perm(len(pos))[:n].astype(np.int64)
        return gs.fragments(pos[sel], gen[sel], L), gen[sel]

    # ---------------------------------------------------------------------------------------------- one workload
    def prepare(self, name, wl):
        """Builds the inputs of a workload in HBM; returns (step, pre_step, info)."""
        t, eng, dev = self.torch, self.eng, self.dev
        K, n, bits, mode = wl["K"], wl["n"], wl["bits"], wl["mode"]
        L, k = L_FRAG, KMER
        with open(os.path.join(ROOT, "tests", "golden", wl["pattern"])) as f:
            pos = self.amd.parse_patterns(f.readlines(), k)
        eng.set_patterns(pos, k)
        P = pos.shape[0]
        F = (L - k + 1) * P * 2
        info = dict(K=K, n=n, bits=bits, mode=mode, P=P, F=F, pos=pos, batch=None, out_class=None)
        pre_step = None
        if mode == "predict":
            eng.adopt_weights(self.weights(bits), K, bits)
            batch = self.device_fragments(n, L, 1234 + self.rank)
            out_class = t.empty(n, dtype=t.int32, device=dev)
            info.update(batch=batch, out_class=out_class, kernel=eng.predict_kernel_name())

            def step():
                eng.predict(batch, True, want_scores=False, out_class=out_class)
        elif mode == "emit":
            n_genomes, glen = wl.get("genomes", max(K, 8)), wl.get("genome_len", 200_000)
            gs = self.device_genomes(n_genomes, glen, 1234)
            per_genome = -(-int(wl.get("coverage", 0.0) * glen) // L)
            if n == n_genomes * per_genome:
                # BASELINE configs[4]: the whole batch `drawfrag -c coverage` makes of these genomes (util/drawfrag.py:36-49:
                # ceil(c*len/l) draws per genome, seed 42 + rank), in the shuffled order train() feeds (opal.py:363)
                rng = self.amd.Rng(42 + self.rank)
                gpos, ggen = rng.sample_positions([glen] * n_genomes, L, wl["coverage"])
                assert len(gpos) == n, (len(gpos), n)
                sel = rng.shuffle_perm(n).astype(np.int64)
                batch = gs.fragments(gpos[sel], ggen[sel], L)
                del gpos, ggen, sel
            else:                                          # --nfrag: a part of that batch (rehearsals)
                batch, _ = self.views(gs, n, L, glen, n_genomes)
            info.update(genomes=n_genomes, genome_len=glen)
            offsets = eng.feature_offsets(batch, True)
            total = int(offsets[-1].item())
            out_idx = t.empty(total, dtype=t.int32, device=dev)
            from opal_amd._lib import check, lib
            info.update(kernel="emit_canon2_kernel", keep=(gs, offsets, out_idx))

            def step():
                check(lib().opal_features(eng._h, batch.data.data_ptr(), batch.nbytes, batch.off.data_ptr(), batch.len.data_ptr(),
                                          batch.n, 1, bits, offsets.data_ptr(), out_idx.data_ptr(), total), eng._h)
        elif mode == "train":
            n_genomes, glen = max(K, 8), 200_000
            gs = self.device_genomes(n_genomes, glen, 1234)
            batch, ggen = self.views(gs, n, L, glen, n_genomes)
            labels = t.from_numpy((ggen.astype(np.int64) % K + 1).astype(np.int32)).to(dev)
            eng.train_begin(K, bits)
            if wl.get("cold"):
                pre_step = lambda: eng.train_begin(K, bits)   # re-zero w and G outside the timed events
            info.update(kernel="train_kernel" if not wl.get("tmode") else "dtrain_score_kernel + sort + dtrain_apply_kernel x2 + dtrain_const_kernel",
                        keep=(gs, labels))

            tmode = wl.get("tmode", 0)

            def step():
                eng.train_batch(batch, labels, True, eta=0.5, sequential=tmode)
        else:   # ingest: ">%08d\n" + 200 bases + "\n" per record, uniform ACGT (PCG64 seed 1234 + rank)
            rec = np.empty((n, 10 + L + 1), dtype=np.uint8)
            rec[:, 0] = ord(">")
            ids = np.arange(n)
            for d in range(8):
                rec[:, 8 - d] = ord("0") + (ids // 10 ** d) % 10
            rec[:, 9] = rec[:, 10 + L] = ord("\n")
            rec[:, 10:10 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[
                np.random.Generator(np.random.PCG64(1234 + self.rank)).integers(0, 4, size=(n, L), dtype=np.uint8)]
            d_text = t.from_numpy(rec.reshape(-1)).to(dev)
            nbytes = int(d_text.numel())
            ing_reads = t.empty(nbytes // 4 + 4096, dtype=t.uint8, device=dev)
            ing_off = t.empty(n, dtype=t.int64, device=dev)
            ing_len = t.empty(n, dtype=t.int32, device=dev)
            import ctypes as C
            from opal_amd._lib import check, lib
            c_a, c_b = C.c_uint64(), C.c_uint64()
            info.update(kernel="ing_fused_kernel + ing_edges_kernel x2 + ing_records_kernel (whole call)")

            def step():
                check(lib().opal_ingest_fasta(eng._h, d_text.data_ptr(), nbytes, ing_reads.data_ptr(), ing_reads.numel(), ing_off.data_ptr(),
                                              ing_len.data_ptr(), n, C.byref(c_a), C.byref(c_b)), eng._h)
                assert c_a.value == n
        return step, pre_step, info

    def timed(self, step, pre_step, steps, warmup):
        """W untimed steps, then exactly `steps` steps between barrier + synchronize on both sides.  Returns (whole-job
        seconds = max over ranks, per-rank seconds up to the rank's own synchronize, mean launch ms from HIP events)."""
        t, dev, dist = self.torch, self.dev, self.dist
        for _ in range(warmup):
            if pre_step:
                pre_step()
            step()
        t.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        t.cuda.synchronize(dev)
        ev = [(t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        t_pre = 0.0
        for a, b in ev:
            if pre_step:                                   # (learner worst case only) untimed re-zeroing between steps
                t.cuda.synchronize(dev)
                tp = time.perf_counter()
                pre_step()
                t.cuda.synchronize(dev)
                t_pre += time.perf_counter() - tp
            a.record()      # torch's current stream IS the ctx stream (Engine binds to it), so these bracket the kernel
            step()
            b.record()
        t.cuda.synchronize(dev)
        mine = time.perf_counter() - t0 - t_pre
        if dist:
            dist.barrier()
        t.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0 - t_pre
        per_rank = [mine]
        if dist:
            on = dev if dist.get_backend() == "nccl" else "cpu"
            tt = t.tensor([elapsed], dtype=t.float64, device=on)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            got = [t.zeros(1, dtype=t.float64, device=on) for _ in range(self.world)]
            dist.all_gather(got, t.tensor([mine], dtype=t.float64, device=on))
            per_rank = [float(x.item()) for x in got]
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return elapsed, per_rank, kern_ms

    def measure(self, name, wl, steps, warmup, traffic_key):
        step, pre_step, info = self.prepare(name, wl)
        elapsed, per_rank, kern_ms = self.timed(step, pre_step, steps, warmup)
        K, n, mode, F = info["K"], info["n"], info["mode"], info["F"]
        frag_per_s = self.world * n * steps / elapsed
        abytes = algorithmic_bytes_per_fragment(L_FRAG, F, K, mode) * n
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        prof = {}
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and traffic_key:
            with open(tp) as f:
                prof = json.load(f).get(traffic_key, {})
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": prof.get("hbm_bytes_per_launch"), "algorithmic_bytes_per_launch": abytes, "kernel_ms": kern_ms}
        if info["mode"] == "train":
            # 6 row passes per feature is what moves when every class of every example has a non-zero step; classes already at
            # their clipped label are skipped (as VW skips zero updates), which at K=1564 happens within the first groups
            roof["note"] = "algorithmic bytes assume no class is skipped: an upper bound on what moved, so frac can exceed 1"
        for key in ("valu_issue_frac", "lds_busy_frac"):       # kernels that are ALU/LDS-bound by construction:
            if key in prof:                                    # integer-ALU and LDS utilisation from the committed PMC passes
                roof[key] = prof[key]
        res = {"value": frag_per_s, "unit": "fragments/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
               "per_rank_ms_per_step": [x / steps * 1e3 for x in per_rank],
               "hashes_per_s": None if mode == "ingest" else frag_per_s * F, "kernel": info["kernel"],
               "dtype": "u8" if mode == "ingest" else "u32" if mode == "emit" else "f32", "roofline": roof,
               "config": {"workload": name, "desc": wl["desc"], "classes": K, "fragments_per_gpu": n, "frag_len": L_FRAG,
                          "kmer": KMER, "row_weight": 16, "patterns": info["P"], "reverse_complement": True, "bits": info["bits"],
                          "features_per_fragment": F,
                          "parallelism": "fragment-sharded x%d, replicated table, no collective" % self.world}}
        if mode == "emit":
            res["config"].update(genomes=info["genomes"], genome_len=info["genome_len"], coverage=n * L_FRAG / (info["genomes"] * info["genome_len"]),
                                 packed_genome_bytes=info["genomes"] * info["genome_len"] // 4, index_buffer_bytes=4 * F * n)
        return res, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2_356sp_1M", choices=sorted(WORKLOADS))
    ap.add_argument("--nfrag", type=int, default=0, help="override fragments per GPU")
    ap.add_argument("--bits", type=int, default=0, help="override table bits")
    ap.add_argument("--classes", type=int, default=0, help="override the class count K (experiments)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling on BASELINE configs[2]: the whole 100 M-fragment, 1564-class job split over the ranks "
                         "(100 M / N fragments per rank, one step = the job; use --steps 1 --warmup 0 at small N)")
    ap.add_argument("--no-extra", action="store_true", help="time the headline workload only (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=600_000)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))     # nothing has touched the GPU yet: the ranks are child processes

    B = Bench(args)
    if args.strong:
        args.workload, args.no_extra = "cfg3_1564sp", True
    overridden = bool(args.nfrag or args.bits or args.classes)
    wl = dict(WORKLOADS[args.workload])
    if args.strong:
        wl["n"] = -(-100_000_000 // B.world)                 # the last fragments of an uneven split are scored twice: never fewer than the job
        wl["desc"] = "1564 classes, 100M x 200bp fragments split over %d GPU(s), numHash=1, -r, b=31 (strong scaling)" % B.world
    if args.nfrag:
        wl["n"] = args.nfrag
    if args.bits:
        wl["bits"] = args.bits
    if args.classes:
        wl["K"] = args.classes

    head, info = B.measure(args.workload, wl, args.steps, args.warmup, None if overridden or args.strong else args.workload)
    res = {"metric": "fragments/sec", "value": head["value"], "unit": "fragments/s", "n_gpus": B.world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
           "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic (uniform ACGT fragments generated on the device from a seed)",
           "config": head["config"], "hashes_per_s": head["hashes_per_s"], "kernel": head["kernel"], "roofline": head["roofline"],
           "per_rank_ms_per_step": head["per_rank_ms_per_step"]}
    if B.rank == 0:
        import shutil
        res["vw_binary_on_box"] = shutil.which("vw")       # time a real vw if one exists (never seen so far)
        if B.world == 1 and not args.no_cpu_baseline and info["mode"] == "predict":
            res["cpu_baseline"] = cpu_baseline(B, info, min(args.cpu_sample, info["n"]))
    del info
    if args.workload == "cfg2_356sp_1M" and not overridden and not args.no_extra:
        res["configs"] = {}
        for name, steps, warmup in EXTRA:
            B.torch.cuda.empty_cache()
            r, inf = B.measure(name, dict(WORKLOADS[name]), steps, warmup, name)
            del inf
            res["configs"][name] = r
    if B.rank == 0:
        print(json.dumps(res), flush=True)
    if B.dist:
        B.dist.barrier()
        B.dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: starts the N ranks itself, as CHILD processes of torch.distributed.run
    (the same command line the driver uses), before this process has initialised the GPU; rank 0's JSON line goes to the
    inherited stdout and the launcher's exit code is returned.  On a box with fewer GPUs than ranks the ranks share the
    devices round-robin and rendezvous over gloo (a rehearsal of the N>1 path, not a measurement)."""
    import socket
    import subprocess
    import torch
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()          # counts devices without creating a GPU context
    if ndev < n:
        env.setdefault("OPAL_DIST_BACKEND", "gloo")
        print("bench.py: %d GPU(s) for %d ranks: ranks share devices, rendezvous over gloo (rehearsal mode)" % (ndev, n), file=sys.stderr, flush=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(B, info, m):
    """Times the CPU oracle (the checker, a scalar C port with OpenMP over fragments) on the first m fragments of the
    batch the GPU just scored, and compares the classes."""
    from oracle import oracle as orc
    orc.build()
    L, k, pos, K, bits = L_FRAG, KMER, info["pos"], info["K"], info["bits"]
    nb = (L + 3) // 4
    packed = info["batch"].data[:m * nb].cpu().numpy().reshape(m, nb)
    codes = ((packed[:, :, None] >> (np.arange(4, dtype=np.uint8) * 2)[None, None, :]) & 3).reshape(m, nb * 4)[:, :L]
    blob = np.ascontiguousarray(np.frombuffer(b"ACGT", dtype=np.uint8)[codes].reshape(-1))
    blob = np.concatenate([blob, np.zeros(1, dtype=np.uint8)])
    soff = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
    threads = min(16, orc.max_threads())
    wh = B.table[:1 << bits].cpu().numpy()
    t0 = time.perf_counter()
    cls, _, _ = orc.predict((blob, soff), k, pos, True, wh, K, bits, want_scores=False, nthreads=threads)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(cls, info["out_class"][:m].cpu().numpy()))   # the timed GPU output agrees with the checker
    m1 = max(1000, m // 50)
    t1 = time.perf_counter()
    orc.predict((blob[:m1 * L + 1], soff[:m1 + 1]), k, pos, True, wh, K, bits, want_scores=False, nthreads=1)
    dt1 = time.perf_counter() - t1
    return {"value": m / dt, "unit": "fragments/s", "cores": threads, "kind": "port",
            "sample": "first %d fragments of the same batch, oracle/opal_oracle.c with %d OpenMP threads (%.1f s)" % (m, threads, dt),
            "value_1thread": m1 / dt1, "gpu_matches_oracle_on_sample": ok,
            "reference_python_1core": "3833 frag/s feature half only (BASELINE.md, survey container); vw absent"}


if __name__ == "__main__":
    main()
