#!/usr/bin/env python3
"""Opal LDPC k-mer hash metagenomic classifier, MI355X-native: same CLI, flags, defaults and files as the
reference's opal.py (frag / train / predict / eval / simulate, reference opal.py:511-645), with the feature
extraction, hashing, scoring and learning done by hand-written HIP kernels in libopal_hip.so instead of Python string
code piped into a `vw` process.

Differences a user can see:
  * no `vw` binary is needed or spawned; `<prefix>_vwps.log` holds this program's own progress lines;
  * `vw-model_final.model` is an opal_amd container (opal_amd/model.py), not a VW regressor;
  * run under `python -m torch.distributed.run --nproc-per-node N` to shard `predict` over N GPUs (no collectives).
"""
import argparse
import glob
import os
import sys
from datetime import datetime

import numpy as np

from . import drawfrag, fasta2skm, ldpc
from . import model as model_io
from .engine import Engine, GenomeSet, Rng, check_geometry, parse_patterns
from .shard import gather_in_rank_order

__version__ = "1.0.0"


# ------------------------------------------------------------------ small file helpers (reference opal.py:52-109)
def unique_lines(file):
    """How many distinct raw lines the file has.  Lines are compared with their newline, as reference opal.py:52-58
    does, so a last line without '\n' counts separately from the same text followed by '\n'."""
    with open(file) as fh:
        return len(set(fh))


def safe_makedirs(directory):
    os.makedirs(directory, exist_ok=True)
    return 0


def extract_column_two(infile, outfile):
    """Second whitespace-separated column of every line ('' when a line has fewer than two), one per output line
    (reference opal.py:65-74)."""
    with open(infile) as src, open(outfile, "w") as dst:
        for row in src:
            cols = row.split()
            dst.write((cols[1] if len(cols) >= 2 else "") + "\n")


def vw_class_to_taxid(inputfile, dicofile, outputfile):
    """Class ids (one per line, '3' or '3.000000') -> taxids through vw-dico.txt (reference opal.py:76-88).
    An id that the dico does not hold raises KeyError, as in the reference."""
    by_class = {}
    with open(dicofile) as fh:
        for ln in fh:
            taxid, klass = ln.strip().split()[:2]            # ValueError on a row with fewer than two words, as there
            by_class[klass] = taxid
    with open(inputfile) as src, open(outputfile, "w") as dst:
        dst.writelines(by_class[str(int(float(ln.strip())))] + "\n" for ln in src)


class PredictionFiles:
    """What `vw -p` followed by vw_class_to_taxid leave on disk (reference opal.py:449-451,484 and :76-88), written
    straight from class-id arrays, a batch at a time: `pred_file` one class id per line, `taxid_file` the taxid of each
    through vw-dico.txt.  A class the dico does not hold raises KeyError, as in the reference."""

    def __init__(self, dicofile, pred_file, taxid_file):
        # the reference looks classes up as str(int(float(line))) (opal.py:87): only rows whose id column is an integer
        # in that canonical spelling can ever match, the others are dead rows
        self.by_class = {}
        with open(dicofile) as fh:
            for ln in fh:
                taxid, klass = ln.strip().split()[:2]
                try:
                    if str(int(klass)) == klass:
                        self.by_class[int(klass)] = taxid
                except ValueError:
                    pass
        top = max(self.by_class) if self.by_class else 0
        self.known = np.zeros(max(top, 0) + 1, dtype=bool)
        for c in self.by_class:
            if c >= 0:
                self.known[c] = True
        self.ids = np.array([("%d\n" % c).encode() for c in range(top + 1)], dtype="S")
        self.tax = np.array([(self.by_class.get(c, "") + "\n").encode("latin-1") for c in range(top + 1)], dtype="S")
        self.pred = open(pred_file, "wb")
        self.taxid = open(taxid_file, "wb")

    def append(self, classes, chunk=8_000_000):
        classes = np.ascontiguousarray(classes, dtype=np.int64)
        for a in range(0, classes.shape[0], chunk):
            part = classes[a:a + chunk]
            ok = (part >= 0) & (part < self.known.shape[0])
            ok[ok] = self.known[part[ok]]
            if not ok.all():
                raise KeyError(str(int(part[~ok][0])))
            # table[idx] gathered at C speed; the NUL padding of the fixed-width entries is dropped from the byte stream
            self.pred.write(self.ids[part].tobytes().replace(b"\0", b""))
            self.taxid.write(self.tax[part].tobytes().replace(b"\0", b""))

    def close(self):
        self.pred.close()
        self.taxid.close()


def write_predictions(classes, dicofile, pred_file, taxid_file):
    """One-shot form of PredictionFiles."""
    out = PredictionFiles(dicofile, pred_file, taxid_file)
    try:
        out.append(classes)
    finally:
        out.close()


def get_fasta_and_taxid(directory):
    """[first *.fasta of the directory, its same-named .taxid]; RuntimeError when either is missing
    (reference opal.py:90-100)."""
    found = glob.glob(directory + "/*.fasta")
    if not found:
        raise RuntimeError("Could not find fasta file in:" + directory)
    taxids = os.path.splitext(found[0])[0] + ".taxid"
    if not os.path.isfile(taxids):
        raise RuntimeError("Could not find matching taxid: " + taxids)
    return [found[0], taxids]


def get_final_model(directory):
    """First file matching *_final.model (reference opal.py:102-109)."""
    found = glob.glob(directory + "/*_final.model")
    if not found:
        raise RuntimeError("Could not find final model file in: " + directory)
    return found[0]


def evaluate_predictions(reffile, predfile):
    '''micro / macro / median per-class accuracy (reference opal.py:111-138)'''
    with open(predfile, "r") as fin:
        pred = fin.read().splitlines()
    with open(reffile, "r") as fin:
        ref = fin.read().splitlines()
    if len(pred) != len(ref):          # the reference builds a pandas frame from both columns (opal.py:123), which refuses
        raise ValueError("All arrays must be of the same length")
    correct = np.array([x == y for x, y in zip(pred, ref)], dtype=bool)
    refs = np.array(ref)
    classes, inv = np.unique(refs, return_inverse=True)
    species = np.bincount(inv, weights=correct, minlength=len(classes)) / np.maximum(np.bincount(inv, minlength=len(classes)), 1)
    micro, macro, median = float(np.mean(correct)), float(np.mean(species)), float(np.median(species))
    print("micro = {:.4f}".format(micro))
    print("macro = {:.4f}".format(macro))
    print("median = {:.4f}".format(median))
    sys.stdout.flush()
    return micro, macro, median


def evaluate_predictions_gpu(reffile, predfile, engine):
    """Same numbers as evaluate_predictions, counted on the GPU (opal_eval_counts): taxid strings are mapped to dense
    ids on the host, the per-class correct/total histograms come from the device."""
    with open(predfile, "r") as fin:
        pred = fin.read().splitlines()
    with open(reffile, "r") as fin:
        ref = fin.read().splitlines()
    if len(pred) != len(ref):
        raise ValueError("All arrays must be of the same length")
    n = len(ref)
    ids = {}
    r = np.fromiter((ids.setdefault(x, len(ids)) for x in ref[:n]), dtype=np.int32, count=n)
    nref = len(ids)
    p = np.fromiter((ids.get(x, -1) for x in pred[:n]), dtype=np.int32, count=n)
    cor, tot = engine.eval_counts(p, r, max(nref, 1))
    species = cor[:nref].astype(np.float64) / np.maximum(tot[:nref], 1)
    micro = float(cor.sum()) / max(n, 1)
    macro, median = float(np.mean(species)), float(np.median(species))
    print("micro = {:.4f}".format(micro))
    print("macro = {:.4f}".format(macro))
    print("median = {:.4f}".format(median))
    sys.stdout.flush()
    return micro, macro, median


def _dist():
    """torch.distributed when launched with WORLD_SIZE>1 (one rank per GPU), else None."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return None, 0, 1, 0
    import torch
    import torch.distributed as dist
    # ranks map to GPUs round-robin (one rank per GPU on a full node).  The data path has no collective; the only
    # exchange is rank 0 collecting the class ids from host memory, so the process group is gloo, not RCCL.
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if not dist.is_initialized():
        dist.init_process_group(os.environ.get("OPAL_DIST_BACKEND", "gloo"))
    return dist, dist.get_rank(), dist.get_world_size(), local


# ------------------------------------------------------------------ frag (reference opal.py:141-206)
def frag(test_dir, frag_dir, args):
    fasta, taxids = get_fasta_and_taxid(test_dir)
    safe_makedirs(frag_dir)
    fasta_out = os.path.join(frag_dir, "test.fragments.fasta")
    gi2taxid_out = os.path.join(frag_dir, "test.fragments.gi2taxid")
    taxid_out = os.path.join(frag_dir, "test.fragments.taxid")
    starttime = datetime.now()
    print("================================================\nDrawing fragments\n{:%Y-%m-%d %H:%M:%S}\n".format(starttime))
    print("frag_length = {}\ncoverage = {}".format(args.frag_length, args.coverage))
    print("------------------------------------------------")
    print("Fasta input:    {}\ntaxids input:   {}\n\nFasta output:   {}\ngi2taxid output:{}\ntaxids output:  {}".format(
        fasta, taxids, fasta_out, gi2taxid_out, taxid_out))
    sys.stdout.flush()
    drawfrag.main(["-i", fasta, "-t", taxids, "-l", str(args.frag_length), "-c", str(args.coverage), "-o", fasta_out,
                   "-g", gi2taxid_out, "-s", str(42)])        # seed 42: reference opal.py:187
    extract_column_two(gi2taxid_out, taxid_out)
    print("------------------------------------------------\nTotal wall clock runtime (sec): {}\n"
          "================================================".format((datetime.now() - starttime).total_seconds()))
    sys.stdout.flush()
    return 0


TRAIN_MODES = {"concurrent": 0, "sequential": 1, "deterministic": 2}      # opal_train_batch's schedule (include/opal_hip.h)


# ------------------------------------------------------------------ the training stream (reference opal.py:329-367)
def batch_label(taxid_line):
    """The label a genome's fragments carry in a training batch: drawfrag writes `<first word of the record name>\t<taxid
    line>` per fragment (util/drawfrag.py:29,47) and extract_column_two keeps the second word of that row (opal.py:65-74),
    i.e. the first word of the taxid line, or '' when the line is blank or missing."""
    words = taxid_line.split()
    return words[0] if words else ""


def draw_training_batch(rng, seed, genome_len, genome_taxid, frag_length, coverage, dico):
    """One batch of the example stream, in the order the reference pipes it to vw (opal.py:329-367): drawfrag.main with
    `-s seed` (random.seed, then one randint per fragment, genome by genome) -> extract_column_two -> the dico update of
    fasta2skm.main_generator (ids in first-seen order of the batch's labels) -> random.shuffle continuing the same stream.
    genome_taxid[g] = line g of the .taxid file ('' when the file is shorter than the fasta).  Returns, per example of the
    shuffled stream, (class id int32, position uint64, genome index uint32); `dico` is updated on disk like the
    reference's vw-dico.txt.  Pinned by tests/golden/train_stream.json (the reference's own functions run in this order)."""
    rng.seed(seed)
    fpos, fgen = rng.sample_positions(genome_len, frag_length, coverage)
    labels = [batch_label(t) for t in genome_taxid]
    sampled = np.unique(fgen).tolist()                       # fragments come genome by genome: first-seen order = genome order
    label2vwid = fasta2skm.update_dictionary((labels[g] + "\n" for g in sampled), dico)
    genome_vwid = np.zeros(len(genome_len), dtype=np.int32)
    for g in sampled:
        genome_vwid[g] = label2vwid[labels[g]]
    perm = rng.shuffle_perm(len(fpos)).astype(np.int64)      # shuffled[i] = original[perm[i]]
    return np.ascontiguousarray(genome_vwid[fgen.astype(np.int64)][perm]), fpos[perm], fgen[perm]


# ------------------------------------------------------------------ train (reference opal.py:208-389)
def train(ref_dir, model_dir, args, engine=None):
    frag_length, coverage, kmer = args.frag_length, args.coverage, args.kmer
    row_weight, hierarchical, num_hash = args.row_weight, args.hierarchical_weight, args.num_hash
    num_batches, num_passes, bits = args.num_batches, args.num_passes, args.bits
    lambda1, lambda2, reverse = args.lambda1, args.lambda2, args.reverse_complement

    fasta, taxids = get_fasta_and_taxid(ref_dir)
    starttime = datetime.now()
    if kmer % row_weight != 0:
        raise ValueError("Row weight [{}] must divide into k-mer length [{}].".format(row_weight, kmer))
    if hierarchical > 0:
        if kmer % hierarchical != 0:
            raise ValueError("Hierarchy middle level [{}] must divide into k-mer length [{}].".format(hierarchical, kmer))
        if hierarchical % row_weight != 0:
            raise ValueError("Row weight[{}] must divide into middle hierarchical structure weight [{}].".format(row_weight, hierarchical))
    check_geometry(kmer, row_weight, num_hash + 1)        # before patterns.txt or the model directory are written
    print("================================================\nTraining using Opal (MI355X-native learner)\n{:%Y-%m-%d %H:%M:%S}\n".format(starttime))
    print("frag_length = {}\ncoverage:       {}\nreverse-complements: {}\nk-mer length:   {}".format(frag_length, coverage, reverse, kmer))
    if hierarchical > 0:
        print("hierarchical:   {}".format(hierarchical))
    print("row weight:     {}\nnum hashes:     {}\nnum batches:    {}\nnum passes:     {}".format(row_weight, num_hash, num_batches, num_passes))
    print("------------------------------------------------\nFasta input:    {}\ntaxids input:   {}\n"
          "------------------------------------------------".format(fasta, taxids))
    num_labels = unique_lines(taxids)                       # --oaa K (reference opal.py:288,311)
    print("Number labels:  {}".format(num_labels))
    sys.stdout.flush()

    safe_makedirs(model_dir)
    dico = os.path.join(model_dir, "vw-dico.txt")
    model_prefix = os.path.join(model_dir, "vw-model")
    pattern_file = os.path.join(model_dir, "patterns.txt")
    final_model_file = model_prefix + "_final.model"
    # --resume (no counterpart in the reference CLI, which passes --save_resume to vw at opal.py:310 but never reads the state
    # back): go on training the model of model_dir, written by an earlier `train --save-resume`, with ITS patterns.txt
    resume = bool(getattr(args, "resume", False))
    save_resume = bool(getattr(args, "save_resume", False))
    if resume:
        if not (os.path.exists(pattern_file) and os.path.exists(final_model_file)):
            raise RuntimeError("--resume needs the patterns.txt and vw-model_final.model of an earlier run in {}".format(model_dir))
    else:
        ldpc.ldpc_write(k=kmer, t=row_weight, _m=num_hash, d=pattern_file)     # reference opal.py:301-302
    with open(pattern_file) as f:
        pos = parse_patterns(f.readlines(), kmer)

    eng = engine or Engine(int(os.environ.get("LOCAL_RANK", "0")))
    eng.set_patterns(pos, kmer)
    eng.set_vw_options()
    batches_done = 0
    if resume:
        hdr = model_io.load_model(final_model_file, eng, resume=True)
        if hdr["K"] != num_labels:
            raise ValueError("--resume: the model has {} classes, the training set {}".format(hdr["K"], num_labels))
        bits, batches_done = hdr["bits"], hdr["batches"]
        eng.train_begin(num_labels, bits, resume=True)
    else:
        eng.train_begin(num_labels, bits)
    # The reference's learner is one vw process fed in order: the same inputs give the same model.  The default schedule
    # here keeps that property (deterministic groups, opal_dtrain.hip: bit-identical models run to run); `concurrent`
    # is the Hogwild schedule (as fast or faster on later passes, models differ slightly between runs), `sequential` the
    # one-wave statement of the rule (slow: for checking).
    mode_name = getattr(args, "train_mode", None) or os.environ.get("OPAL_TRAIN_MODE", "deterministic")
    if mode_name not in TRAIN_MODES:
        raise ValueError("train mode [{}] must be one of {}".format(mode_name, ", ".join(TRAIN_MODES)))
    train_mode = TRAIN_MODES[mode_name]
    log = open(model_prefix + "_vwps.log", "w")
    log.write("opal_amd learner: oaa=%d bits=%d l1=%g l2=%g passes=%d schedule=%s\n" % (num_labels, bits, lambda1, lambda2, num_passes, mode_name))

    with open(taxids, "r") as f:
        taxid_lines = [l.rstrip("\n") for l in f]
    # genomes are split into records and packed on the GPU (opal_ingest_fasta = fasta_reader's records) and stay in
    # HBM; a training batch is only (offset, length, label) per fragment: the fragments drawfrag would write to
    # train.batch-i.fasta are views into the resident genomes
    genomes = GenomeSet.from_fasta(eng, fasta)
    genome_len = [int(x) for x in genomes.lengths]
    genome_taxid = [taxid_lines[g] if g < len(taxid_lines) else "" for g in range(len(genome_len))]

    # Several passes (reference opal.py:316-320: `--cache_file ... --passes N` on the ONE vw process that is fed every
    # batch): vw replays its cache, i.e. the whole stream of all batches in the order it arrived, N-1 more times.
    # With more than one pass vw also sets aside every 10th example of the stream (holdout_period 10) in ALL passes,
    # keeps the regressor of the pass with the lowest holdout loss and stops after 3 passes without improvement
    # (early_terminate 3).  [VW-unverified: documented vw behaviour, no vw here to pin it]
    multipass = num_passes > 1
    HOLDOUT_PERIOD, EARLY_TERMINATE = 10, 3
    stream = []                                              # per batch: (train views, labels, holdout views, labels)
    seen = 0
    seed = 420 + batches_done                                # a resumed run goes on with the seeds the first run would have used
    rng = Rng()
    for i in range(num_batches):
        seed = seed + 1                                      # 421, 422, ... (reference opal.py:329)
        print("Drawing fragments for batch {}".format(i))
        vwid, fpos, fgen = draw_training_batch(rng, seed, genome_len, genome_taxid, frag_length, coverage, dico)
        n_frag = len(fpos)
        if vwid.size and vwid.max() > num_labels:
            raise ValueError("class id %d exceeds --oaa %d" % (vwid.max(), num_labels))
        print("Getting training set ...")
        print("Shuffling training set ...")
        held = (np.arange(seen + 1, seen + n_frag + 1) % HOLDOUT_PERIOD == 0) if multipass else np.zeros(n_frag, dtype=bool)
        seen += n_frag
        keep = ~held
        batch = genomes.fragments(fpos[keep], fgen[keep], frag_length)
        y = np.ascontiguousarray(vwid[keep])
        print("Training on the GPU ...")
        sys.stdout.flush()
        pred = eng.train_batch(batch, y, reverse, eta=0.5, l1=lambda1, l2=lambda2, sequential=train_mode, want_pred=True)
        err = float((pred.cpu().numpy() != y).mean()) if y.size else 0.0
        msg = "batch %d pass 0: %d examples, progressive error %.4f" % (i, y.size, err)
        print(msg)
        log.write(msg + "\n")
        if multipass:
            stream.append((batch, y, genomes.fragments(fpos[held], fgen[held], frag_length), np.ascontiguousarray(vwid[held])))
    w = None
    if multipass:
        def holdout_error():
            wrong = total = 0
            for _, _, hb, hy in stream:
                if hy.size:
                    cls, _ = eng.predict(hb, reverse)
                    wrong += int((cls.cpu().numpy() != hy).sum())
                    total += hy.size
            return wrong / total if total else 0.0
        best = holdout_error()
        w = eng.get_weights()
        log.write("pass 0: holdout error %.4f\n" % best)
        no_win = 0
        for p in range(1, num_passes):
            for batch, y, _, _ in stream:
                eng.train_batch(batch, y, reverse, eta=0.5, l1=lambda1, l2=lambda2, sequential=train_mode)
            e = holdout_error()
            msg = "pass %d: holdout error %.4f" % (p, e)
            print(msg)
            log.write(msg + "\n")
            if e < best:
                best, w, no_win = e, eng.get_weights(), 0
            else:
                no_win += 1
                if no_win == EARLY_TERMINATE:
                    log.write("no improvement in %d passes: stopping, keeping the best regressor\n" % EARLY_TERMINATE)
                    break
    if w is None:
        w = eng.get_weights()
    # --save-resume: the file also carries the accumulators (of the table as trained to the end; with several passes the saved
    # weights may be an earlier pass's, the best on the holdout set) and the number of batches seen
    nnz = model_io.save_model(final_model_file, w, num_labels, bits, g=eng.get_accumulators() if save_resume else None,
                              batches=batches_done + num_batches)
    log.write("saved %s (%d non-zero weights)\n" % (final_model_file, nnz))
    log.close()
    print("------------------------------------------------\nTotal wall clock runtime (sec): {}\n"
          "================================================".format((datetime.now() - starttime).total_seconds()))
    sys.stdout.flush()
    return 0


def _next_header(text, start):
    """Smallest i >= start where a header line begins ('>' right after a line terminator; a lone \\r ends a line too,
    as in the reference's open(path, 'r')), or len(text) when there is none."""
    size, window = len(text), 1 << 24
    i0 = max(start, 1)
    while i0 < size:
        w = np.asarray(text[i0 - 1:min(size, i0 + window)])
        hit = np.flatnonzero((w[1:] == 62) & ((w[:-1] == 10) | (w[:-1] == 13)))
        if hit.size:
            return i0 + int(hit[0])
        i0 += window
    return size


def fasta_pieces(path, chunk_bytes, rank=0, world=1):
    """The file as uint8 arrays of about chunk_bytes each, cut in front of header lines, so that files larger than host
    or GPU memory stream through predict the way the reference streams lines into vw (opal.py:458-469).  Piece i is
    [cut(i*chunk_bytes), cut((i+1)*chunk_bytes)) with cut(x) = start of the first header line at or after x, so every
    piece can be found without looking at the rest of the file: rank r of `world` yields pieces r, r+world, ... (and an
    empty array once the file has run out, so that all ranks make the same number of steps) and reads only those.
    A header resets the reader's state (fasta_functions.py:24-35), so pieces are independent; every piece but the one
    that ends the file gets one extra '>' appended: the reader drops an EMPTY LAST record (fasta_functions.py:20-22),
    which must only happen at the real end of the file, and that sentinel is exactly such a record."""
    size = os.path.getsize(path)
    chunk_bytes = max(1, int(chunk_bytes))
    n_pieces = max(1, -(-size // chunk_bytes))
    rounds = -(-n_pieces // world)
    empty = np.zeros(0, dtype=np.uint8)
    if n_pieces == 1:
        yield np.fromfile(path, dtype=np.uint8) if rank == 0 else empty
        return
    text = np.memmap(path, dtype=np.uint8, mode="r")

    def cut(x):
        return 0 if x <= 0 else (size if x >= size else _next_header(text, x))

    for j in range(rounds):
        i = j * world + rank
        lo, hi = (cut(i * chunk_bytes), cut((i + 1) * chunk_bytes)) if i < n_pieces else (size, size)
        if hi <= lo:
            yield empty
            continue
        piece = np.empty(hi - lo + (1 if hi < size else 0), dtype=np.uint8)
        piece[:hi - lo] = text[lo:hi]
        if hi < size:
            piece[-1] = 62
        yield piece


def default_chunk_bytes(path, world):
    """Piece size of predict: 256 MiB (OPAL_PREDICT_CHUNK_BYTES overrides): small enough that reading, uploading and
    writing results overlap with scoring even for a file of a few hundred MB, large enough to fill the GPU (1.2 M reads); with several ranks at least four pieces per
    rank (but no smaller than 16 MiB) so that the ranks stay evenly loaded."""
    env = os.environ.get("OPAL_PREDICT_CHUNK_BYTES")
    if env:
        return int(env)
    if world <= 1:
        return 256 << 20
    return int(min(256 << 20, max(16 << 20, os.path.getsize(path) // (4 * world) + 1)))


def _uploaded(pieces, device_index):
    """Moves every piece to HBM on a stream of its own, inside the thread that reads the file: the copy of piece i+1
    then runs while the predict kernel works on piece i (the copy engines are independent of the compute units)."""
    side = None
    for piece in pieces:
        if piece.size == 0:
            yield piece
            continue
        import torch
        if side is None:
            torch.cuda.set_device(device_index)
            side = torch.cuda.Stream(device=device_index)
        with torch.cuda.stream(side):
            d = torch.from_numpy(piece).to(torch.device("cuda", device_index))
        side.synchronize()                                   # the bytes are in HBM when the consumer gets the tensor
        yield d


class _Prefetched:
    """Runs a generator in a background thread, `depth` items ahead, starting at once: the file is being read while the
    GPU context and the model are set up, and the next piece is read while the GPU scores the current one."""

    _DONE = object()

    def __init__(self, gen, depth=1):
        import queue
        import threading
        self._q = queue.Queue(maxsize=depth)

        def work():
            try:
                for item in gen:
                    self._q.put(item)
                self._q.put(self._DONE)
            except BaseException as e:      # hand the error to the consumer
                self._q.put(e)

        threading.Thread(target=work, daemon=True).start()

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is self._DONE:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item


# ------------------------------------------------------------------ predict (reference opal.py:392-493)
def predict(model_dir, test_dir, predict_dir, args, engine=None, chunk_bytes=None):
    kmer, reverse = args.kmer, args.reverse_complement
    try:
        fasta = glob.glob(test_dir + "/*.fasta")[0]
    except IndexError:
        raise RuntimeError("No fasta file found in: " + test_dir)
    model = get_final_model(model_dir)
    dico = os.path.join(model_dir, "vw-dico.txt")
    pattern_file = os.path.join(model_dir, "patterns.txt")
    starttime = datetime.now()
    dist, rank, world, local = _dist()
    if rank == 0:
        print("================================================\nPredicting using Opal (MI355X-native scorer)\n{:%Y-%m-%d %H:%M:%S}\n".format(starttime))
        print("k-mer length:   {}\n------------------------------------------------".format(kmer))
        print("Fasta input:    {}\nModel used:     {}\nDict used:      {}\nLDPC patterns:  {}\nreverse-complements: {}".format(
            fasta, model, dico, pattern_file, reverse))
        print("------------------------------------------------")
        sys.stdout.flush()
    safe_makedirs(predict_dir)
    prefix = os.path.join(predict_dir, "test.fragments-db")
    prediction_file = prefix + ".preds.vw"

    # the file's bytes go to HBM as they are, a piece of about 256 MiB at a time (read by a background thread that starts
    # now, before the GPU context exists); records are split and 2-bit packed on the device (opal_ingest_fasta)
    if chunk_bytes is None:
        chunk_bytes = default_chunk_bytes(fasta, world)
    pieces = _Prefetched(_uploaded(fasta_pieces(fasta, chunk_bytes, rank, world), local))
    with open(pattern_file, "r") as f:
        pos = parse_patterns(f.readlines(), kmer)
    eng = engine or Engine(local)
    eng.set_patterns(pos, kmer)
    hdr = model_io.load_model(model, eng)
    if getattr(args, "dump_vw_text", None) and rank == 0:
        ns = argparse.Namespace(input=fasta, taxid=None, kmer=kmer, dico=None, output=args.dump_vw_text, pattern=pattern_file,
                                reverse=reverse)
        with open(args.dump_vw_text, "w", encoding="latin-1") as f:
            f.writelines(fasta2skm.main_generator(ns, engine=eng))
    # every rank scores its own pieces (fragments are independent; no data-path collective); after each
    # round of pieces the class ids are concatenated on rank 0 in piece order = file order
    n_frag, pending = 0, None
    out = PredictionFiles(dico, prediction_file, prefix + '.preds.taxid') if rank == 0 else None   # `vw -p` file + vw_class_to_taxid
    try:
        for piece in pieces:
            cls = None
            if len(piece):
                if hasattr(piece, "record_stream"):
                    piece.record_stream(eng.torch.cuda.current_stream(eng.device))
                batch = eng.ingest_fasta(piece)
                if batch.n:
                    cls, _ = eng.predict(batch, reverse)          # asynchronous: the kernel runs while ...
            if pending is not None:
                out.append(pending)                               # ... the previous round's lines are written
                pending = None
            local_cls = cls.cpu().numpy() if cls is not None else np.zeros(0, dtype=np.int32)
            got = gather_in_rank_order(local_cls, dist)
            if rank == 0:
                pending = got
                n_frag += int(got.shape[0])
            del piece
        if pending is not None:
            out.append(pending)
    finally:
        if out is not None:
            out.close()
    if rank == 0:
        with open(prefix + "_vwps.log", "w") as log:
            log.write("opal_amd scorer: %d fragments, oaa=%d bits=%d, kernel %s, %d rank(s)\n" % (
                n_frag, hdr["K"], hdr["bits"], eng.predict_kernel_name(), world))
        print("------------------------------------------------\nPredicted labels:   {}\nTotal wall clock runtime (sec): {}\n"
              "================================================".format(prefix + '.preds.taxid', (datetime.now() - starttime).total_seconds()))
        sys.stdout.flush()
    if dist:
        dist.barrier()
    return prefix + '.preds.taxid'


# flag table: (short, long, argparse kwargs).  Names, types and defaults are those of reference opal.py:519-545.
_FLAGS = {
    "frag_length": ("-l", "--frag-length", dict(type=int, default=64, help="fragment length drawn from each fasta record")),
    "kmer": ("-k", "--kmer", dict(type=int, default=64, help="k-mer (window) length")),
    "coverage": ("-c", "--coverage", dict(type=float, default=15.0, help="how many times over each genome is covered by fragments")),
    "reverse": ("-r", "--reverse-complement", dict(action="store_true", help="also use the reverse complement of every fragment")),
    "hierarchical": (None, "--hierarchical-weight", dict(type=int, default=-1, help="accepted and validated like the reference; the pattern generator ignores it there too")),
    "row_weight": (None, "--row-weight", dict(type=int, default=16, help="positions kept per spaced k-mer; must divide the k-mer length")),
    "num_hash": (None, "--num-hash", dict(type=int, default=8, help="number of LDPC patterns (hash functions) besides the contiguous prefix")),
    "num_batches": (None, "--num-batches", dict(type=int, default=1, help="number of freshly sampled training batches")),
    "num_passes": (None, "--num-passes", dict(type=int, default=1, help="learner passes over each batch")),
    "bits": (None, "--bits", dict(type=int, default=31, help="log2 of the number of weight slots")),
    "lambda1": (None, "--lambda1", dict(type=float, default=0., help="l1 regularisation")),
    "lambda2": (None, "--lambda2", dict(type=float, default=0., help="l2 regularisation")),
}
_TRAIN_FLAGS = ("frag_length", "coverage", "kmer", "reverse", "num_batches", "num_passes", "num_hash", "row_weight",
                "hierarchical", "bits", "lambda1", "lambda2")
# sub-command -> (help, positional (name, help) pairs, flags)
_COMMANDS = {
    "frag": ("cut a fasta file into random fragments", (("test_dir", "directory with one .fasta and its .taxid"),
                                                       ("frag_dir", "output directory")), ("frag_length", "coverage")),
    "train": ("learn a model from reference genomes", (("train_dir", "directory with one .fasta and its .taxid"),
                                                      ("model_dir", "output directory")), _TRAIN_FLAGS),
    "predict": ("classify fragments with a trained model", (("model_dir", "directory written by train"),
                                                           ("test_dir", "directory with the fragments' .fasta"),
                                                           ("predict_dir", "output directory")), ("reverse", "kmer")),
    "eval": ("score predicted labels against a gold standard", (("reference_file", "one true label per line"),
                                                               ("predicted_labels", "one predicted label per line")), ()),
    "simulate": ("frag, train, predict and eval in one go", (("test_dir", "test genomes"), ("train_dir", "training genomes"),
                                                            ("out_dir", "output directory (1frag, 2model, 3predict)")), _TRAIN_FLAGS),
}


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter, description=__doc__)
    parser.add_argument('--version', action='version', version='%(prog)s ' + __version__)
    sub = parser.add_subparsers(help="sub-commands", dest="mode")
    for name, (text, positionals, flags) in _COMMANDS.items():
        p = sub.add_parser(name, help=text, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        for pos_name, pos_help in positionals:
            p.add_argument(pos_name, help=pos_help)
        if name == "simulate":
            p.add_argument("--do-not-fragment", action="store_true", help="use the test fasta as it is")
        if name == "train":
            p.add_argument("--save-resume", action="store_true",
                           help="also store the learner's state (AdaGrad accumulators) in vw-model_final.model, as vw's --save_resume "
                                "does, so that `train --resume` can go on from it (larger file)")
            p.add_argument("--resume", action="store_true",
                           help="continue training the model of model_dir (written with --save-resume) on further batches, "
                                "with its patterns.txt and dictionary")
        if name in ("train", "simulate"):
            p.add_argument("--train-mode", choices=sorted(TRAIN_MODES), default=None,
                           help="learner schedule (not in the reference, whose vw process is sequential): deterministic = same model "
                                "every run (default, or $OPAL_TRAIN_MODE); concurrent = Hogwild; sequential = one wave, slow")
        if name == "predict":
            p.add_argument("--dump-vw-text", metavar="FILE", default=None,
                           help="debug: also write the VW text lines the reference would pipe to vw (byte-identical)")
        for key in flags:
            short, long_, kw = _FLAGS[key]
            p.add_argument(*([short, long_] if short else [long_]), **kw)
    return parser


def main(argv):
    args = build_parser().parse_args(argv)
    print(args)
    sys.stdout.flush()
    mode = args.mode
    if mode == "simulate":
        fullstarttime = datetime.now()
        print("Full simulation\n{:%Y-%m-%d %H:%M:%S}\nFragment mode: {}".format(fullstarttime, not args.do_not_fragment))
        frag_dir = os.path.join(args.out_dir, '1frag')
        model_dir = os.path.join(args.out_dir, '2model')
        predict_dir = os.path.join(args.out_dir, '3predict')
        if args.do_not_fragment:
            train(args.train_dir, model_dir, args)
            pf = predict(model_dir, args.test_dir, predict_dir, args)
            _, rf = get_fasta_and_taxid(args.test_dir)
        else:
            frag(args.test_dir, frag_dir, args)
            train(args.train_dir, model_dir, args)
            pf = predict(model_dir, frag_dir, predict_dir, args)
            _, rf = get_fasta_and_taxid(frag_dir)
        print("Evaluation reference file: " + rf)
        sys.stdout.flush()
        evaluate_predictions(rf, pf)
        print("Total full sim wall clock runtime (sec): {}".format((datetime.now() - fullstarttime).total_seconds()))
    elif mode == "frag":
        frag(args.test_dir, args.frag_dir, args)
    elif mode == "train":
        train(args.train_dir, args.model_dir, args)
    elif mode == "predict":
        predict(args.model_dir, args.test_dir, args.predict_dir, args)
    elif mode == "eval":
        evaluate_predictions(args.reference_file, args.predicted_labels)


if __name__ == "__main__":
    main(sys.argv[1:])
