"""Host-side driver of libopal_hip.so: packs fragments, owns the per-GPU context, moves buffers with torch.

torch is plumbing only (device memory, streams); every computation on the hot path happens inside
libopal_hip.so (hand-written HIP kernels for gfx950) through the C-ABI of include/opal_hip.h.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

__all__ = ["PackedReads", "Rng", "Engine", "GenomeSet", "parse_patterns", "check_geometry"]


def parse_patterns(lines, kmer):
    """Mirror of fasta2skm.create_pattern_getters (reference util/fasta2skm.py:120-134): returns the [P, t]
    int32 table of positions.  `lines` is file.readlines() of patterns.txt, or None for the identity k-mer.
    Raises AssertionError on a malformed file, exactly where the reference asserts."""
    if lines:
        num_hash, row_weight = [int(x) for x in lines[0].split()[:2]]
        assert kmer % row_weight == 0
        rows = []
        for i in range(num_hash):
            row = [int(x) for x in lines[1 + i].split()]
            assert len(row) == row_weight
            rows.append(row)
    else:
        rows = [list(range(kmer))]
    return np.ascontiguousarray(np.array(rows, dtype=np.int32))


MAX_K, MAX_T, MAX_P = 256, 64, 64       # OPAL_MAX_K / OPAL_MAX_T / OPAL_MAX_P of include/opal_hip.h


def check_geometry(kmer, row_weight, n_patterns):
    """The limits of libopal_hip.so's kernels, checked before anything is written or uploaded (the reference has none:
    its getters are Python itemgetters).  Raises ValueError with the flag to change."""
    if kmer > MAX_K:
        raise ValueError("k-mer length %d is beyond this build's limit of %d (-k/--kmer)" % (kmer, MAX_K))
    if row_weight > MAX_T:
        raise ValueError("%d positions per pattern is beyond this build's limit of %d (--row-weight; without a pattern file the "
                         "whole k-mer is one pattern, so then -k must be <= %d)" % (row_weight, MAX_T, MAX_T))
    if n_patterns > MAX_P:
        raise ValueError("%d patterns is beyond this build's limit of %d (--num-hash must be <= %d)" % (n_patterns, MAX_P, MAX_P - 1))


class PackedReads:
    """A fragment batch in the layout of include/opal_hip.h (host copy): 2-bit packed A/C/G/T, raw bytes else."""

    def __init__(self, data, off, length):
        self.data, self.off, self.len = data, off, length
        self.n = int(len(off))

    def sequences(self):
        """Decodes the batch back into the text of its records (list of bytes): packed fragments are addressed in
        bases and may start anywhere inside a byte, raw ones in bytes (include/opal_hip.h)."""
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        data = np.asarray(self.data, dtype=np.uint8)
        out = []
        for o, lw in zip(self.off.tolist(), self.len.tolist()):
            ln = lw & 0x7FFFFFFF
            if lw & _lib.RAW_FLAG:
                out.append(data[o:o + ln].tobytes())
            else:
                p = np.arange(o, o + ln, dtype=np.uint64)
                out.append(acgt[(data[(p >> np.uint64(2)).astype(np.int64)] >> ((p & np.uint64(3)) * np.uint64(2)).astype(np.uint8)) & 3].tobytes())
        return out

    @staticmethod
    def from_blob(blob: np.ndarray, seq_off: np.ndarray):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        n = len(seq_off) - 1
        L = lib()
        frag_off = np.zeros(max(n, 1), dtype=np.uint64)
        frag_len = np.zeros(max(n, 1), dtype=np.uint32)
        nbytes = C.c_uint64(0)
        check(L.opal_pack_reads(blob.ctypes.data, seq_off.ctypes.data, n, None, 0, frag_off.ctypes.data,
                                frag_len.ctypes.data, C.byref(nbytes)))
        out = np.zeros(max(int(nbytes.value), 1), dtype=np.uint8)
        check(L.opal_pack_reads(blob.ctypes.data, seq_off.ctypes.data, n, out.ctypes.data, out.size, frag_off.ctypes.data,
                                frag_len.ctypes.data, C.byref(nbytes)))
        return PackedReads(out[:max(int(nbytes.value), 1)], frag_off[:n], frag_len[:n])

    @staticmethod
    def from_sequences(seqs):
        """seqs: iterable of bytes (the text of each record, as fasta_reader would yield it)."""
        seqs = list(seqs)
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if seqs:
            off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        blob = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8)
        return PackedReads.from_blob(blob, off)

    @staticmethod
    def from_codes(codes: np.ndarray):
        """codes: uint8 [n, L] of 2-bit base codes (A=0 C=1 G=2 T=3), every fragment the same length."""
        n, L = codes.shape
        nb = (L + 3) // 4
        pad = np.zeros((n, nb * 4), dtype=np.uint8)
        pad[:, :L] = codes
        q = pad.reshape(n, nb, 4)
        data = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8).reshape(-1)
        off = (np.arange(n, dtype=np.uint64) * np.uint64(nb * 4))   # packed fragments are addressed in bases
        return PackedReads(np.ascontiguousarray(data), off, np.full(n, L, dtype=np.uint32))


class Rng:
    """CPython-exact `random` stream (what drawfrag.py:24,38 and opal.py:363 consume)."""

    def __init__(self, seed=None):
        h = C.c_void_p()
        check(lib().opal_rng_create(C.byref(h)))
        self._h = h
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        try:
            lib().opal_rng_destroy(self._h)
        except Exception:
            pass

    def seed(self, seed: int):
        a = abs(int(seed))
        limbs = []
        while a:
            limbs.append(a & 0xFFFFFFFF)
            a >>= 32
        arr = np.array(limbs or [0], dtype=np.uint32)
        check(lib().opal_rng_seed(self._h, arr.ctypes.data, len(arr)))

    def randint(self, a: int, b: int) -> int:
        out = C.c_uint64()
        check(lib().opal_rng_randbelow(self._h, b - a + 1, C.byref(out)))
        return a + int(out.value)

    def sample_positions(self, genome_len, frag_len: int, coverage: float):
        gl = np.ascontiguousarray(np.asarray(genome_len, dtype=np.uint64))
        cap = int(sum(int(coverage * int(x) / frag_len) + 2 for x in gl if x >= frag_len)) + 16
        pos = np.zeros(cap, dtype=np.uint64)
        gen = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64()
        check(lib().opal_sample_positions(self._h, gl.ctypes.data, len(gl), frag_len, float(coverage), pos.ctypes.data,
                                          gen.ctypes.data, cap, C.byref(n)))
        return pos[:n.value], gen[:n.value]

    def shuffle_perm(self, n: int):
        perm = np.arange(n, dtype=np.uint64)
        check(lib().opal_rng_shuffle(self._h, perm.ctypes.data, n))
        return perm


class DeviceBatch:
    def __init__(self, data, off, length, n, nbytes):
        self.data, self.off, self.len, self.n, self.nbytes = data, off, length, n, nbytes


class GenomeSet:
    """Genomes resident in HBM (packed once); fragments are (offset, length) views into them: sampling a batch
    uploads 12 bytes per fragment instead of its bases (zero-copy `drawfrag`, reference util/drawfrag.py:38-39)."""

    def __init__(self, engine, pr: PackedReads, dev=None):
        """pr: host-side batch (only its off/len arrays are used when `dev`, the device copy, is given)."""
        self.engine = engine
        self.host = pr
        self.dev = dev if dev is not None else engine.upload(pr)
        self.lengths = (pr.len & np.uint32(0x7FFFFFFF)).astype(np.uint64)

    @staticmethod
    def from_fasta(engine, path):
        """Genomes straight from a FASTA file through the GPU ingest (the host never walks the sequence text);
        same records as fasta_reader over open(path, 'r') (reference util/fasta_functions.py:9-43)."""
        dev = engine.ingest_fasta(path)
        off = dev.off[:dev.n].cpu().numpy().view(np.uint64)
        ln = dev.len[:dev.n].cpu().numpy().view(np.uint32)
        return GenomeSet(engine, PackedReads(None, off, ln), dev)

    def fragments(self, pos, gen, frag_len: int) -> "DeviceBatch":
        """pos, gen: arrays from Rng.sample_positions (position inside the genome, genome index).  The views are built
        on the device (opal_fragment_views) from the 12 bytes per fragment the sampler produced."""
        t = self.engine.torch
        dev = self.engine.device
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        gen = np.ascontiguousarray(gen, dtype=np.uint32)
        n = int(len(pos))
        assert len(gen) == n
        d_pos = t.from_numpy(pos.view(np.int64)).to(dev)
        d_gen = t.from_numpy(gen.view(np.int32)).to(dev)
        d_off = t.empty(max(n, 1), dtype=t.int64, device=dev)
        d_len = t.empty(max(n, 1), dtype=t.int32, device=dev)
        check(lib().opal_fragment_views(self.engine._h, self.dev.off.data_ptr(), self.dev.len.data_ptr(), self.dev.n,
                                        d_pos.data_ptr(), d_gen.data_ptr(), n, int(frag_len), d_off.data_ptr(), d_len.data_ptr()),
              self.engine._h)
        return DeviceBatch(self.dev.data, d_off, d_len, n, self.dev.nbytes)


class Engine:
    """One per GPU.  Mirrors the predict half of opal.py (reference opal.py:449-469) as in-process calls."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("opal_amd.Engine needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else None
        h = C.c_void_p()
        check(lib().opal_ctx_create(device, stream, C.byref(h)))
        self._h = h
        self.k = self.P = self.t = 0
        self.K = self.bits = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().opal_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ configuration
    def set_patterns(self, pos: np.ndarray, k: int):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        P, t = pos.shape
        check_geometry(k, t, P)
        check(lib().opal_set_patterns(self._h, pos.ctypes.data, P, t, k), self._h)
        self.k, self.P, self.t = k, P, t

    def load_weights(self, w, K: int, bits: int):
        """w: numpy float32 [2^bits] (host) or a torch cuda float32 tensor of that size."""
        n_slots = 1 << bits
        if isinstance(w, np.ndarray):
            w = np.ascontiguousarray(w, dtype=np.float32)
            assert w.size == n_slots
            check(lib().opal_load_weights(self._h, w.ctypes.data, n_slots, K, bits), self._h)
        else:
            assert w.is_cuda and w.dtype == self.torch.float32 and w.numel() == n_slots and w.is_contiguous()
            check(lib().opal_set_weights_device(self._h, w.data_ptr(), n_slots, K, bits), self._h)
        self.K, self.bits = K, bits

    def adopt_weights(self, w, K: int, bits: int):
        """w: torch cuda float32 tensor with at least 2^bits + K + 4 elements whose first 2^bits are the table; it is scored
        in place (no second copy; the tail receives the mirror of the head) and must outlive its use."""
        n_slots = 1 << bits
        assert w.is_cuda and w.dtype == self.torch.float32 and w.is_contiguous() and w.numel() >= n_slots + K + 4
        check(lib().opal_adopt_weights_device(self._h, w.data_ptr(), n_slots, w.numel(), K, bits), self._h)
        self._adopted = w                      # keeps the tensor alive
        self.K, self.bits = K, bits

    def set_tuning(self, predict_nt=-1, emit_qchunk=0, emit_fast=True, train_group=0):
        check(lib().opal_set_tuning(self._h, int(predict_nt), int(emit_qchunk), int(emit_fast), int(train_group)), self._h)

    def set_vw_options(self, hash_all=False, add_constant=True, clip=True, min_label=-1.0, max_label=1.0):
        check(lib().opal_set_vw_options(self._h, int(hash_all), int(add_constant), int(clip), min_label, max_label), self._h)

    def sync(self):
        check(lib().opal_sync(self._h), self._h)

    def predict_kernel_name(self):
        return lib().opal_predict_kernel_name(self._h).decode()

    # ------------------------------------------------------------------ data movement (torch plumbing)
    def upload(self, pr: PackedReads) -> DeviceBatch:
        t = self.torch
        data = t.from_numpy(pr.data).to(self.device)
        off = t.from_numpy(pr.off.view(np.int64)).to(self.device) if pr.n else t.zeros(1, dtype=t.int64, device=self.device)
        ln = t.from_numpy(pr.len.view(np.int32)).to(self.device) if pr.n else t.zeros(1, dtype=t.int32, device=self.device)
        return DeviceBatch(data, off, ln, pr.n, int(pr.data.size))

    # ------------------------------------------------------------------ hot path
    def feature_offsets(self, b: DeviceBatch, reverse: bool):
        t = self.torch
        out = t.empty(b.n + 1, dtype=t.int64, device=self.device)
        check(lib().opal_feature_offsets(self._h, b.len.data_ptr(), b.n, int(reverse), out.data_ptr()), self._h)
        return out

    def features(self, b: DeviceBatch, reverse: bool, bits: int, offsets=None):
        """Emit mode: (idx int32-as-uint32 tensor [total], offsets int64 tensor [n+1]) on the device."""
        t = self.torch
        if offsets is None:
            offsets = self.feature_offsets(b, reverse)
        total = int(offsets[-1].item())
        idx = t.empty(max(total, 1), dtype=t.int32, device=self.device)
        check(lib().opal_features(self._h, b.data.data_ptr(), b.nbytes, b.off.data_ptr(), b.len.data_ptr(), b.n,
                                  int(reverse), bits, offsets.data_ptr(), idx.data_ptr(), total), self._h)
        return idx[:total], offsets

    def feature_names(self, b: DeviceBatch, reverse: bool, offsets=None):
        t = self.torch
        if offsets is None:
            offsets = self.feature_offsets(b, reverse)
        total = int(offsets[-1].item())
        names = t.empty(max(total * self.t, 1), dtype=t.uint8, device=self.device)
        check(lib().opal_feature_names(self._h, b.data.data_ptr(), b.nbytes, b.off.data_ptr(), b.len.data_ptr(), b.n,
                                       int(reverse), offsets.data_ptr(), names.data_ptr(), total), self._h)
        return names[:total * self.t].view(total, self.t) if total else names[:0].view(0, self.t), offsets

    def predict(self, b: DeviceBatch, reverse: bool, want_scores=False, out_class=None, out_scores=None):
        """Fused extract+hash+score: returns (classes int32 [n] (1-based VW ids), scores float32 [n,K] or None)."""
        t = self.torch
        if out_class is None:
            out_class = t.empty(max(b.n, 1), dtype=t.int32, device=self.device)
        if want_scores and out_scores is None:
            out_scores = t.empty((max(b.n, 1), self.K), dtype=t.float32, device=self.device)
        check(lib().opal_predict(self._h, b.data.data_ptr(), b.nbytes, b.off.data_ptr(), b.len.data_ptr(), b.n,
                                 int(reverse), out_class.data_ptr(), out_scores.data_ptr() if want_scores else None), self._h)
        return out_class[:b.n], (out_scores[:b.n] if want_scores else None)

    # ------------------------------------------------------------------ trainer (replaces the vw learner process)
    def train_begin(self, K: int, bits: int, resume=False):
        check(lib().opal_train_begin(self._h, K, bits, int(resume)), self._h)
        self.K, self.bits = K, bits

    def train_batch(self, b: DeviceBatch, labels, reverse: bool, eta=0.5, l1=0.0, l2=0.0, sequential=False, want_pred=False):
        """One pass over the batch in the order given. labels: VW class ids 1..K (numpy int32 or cuda tensor).
        sequential: False/0 = concurrent (Hogwild), True/1 = one wave, 2 = deterministic groups (include/opal_hip.h).
        Returns the progressive predictions (class predicted before each example's update) or None."""
        t = self.torch
        if isinstance(labels, np.ndarray):
            labels = t.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(self.device)
        assert labels.dtype == t.int32 and labels.numel() >= b.n
        pred = t.empty(max(b.n, 1), dtype=t.int32, device=self.device) if want_pred else None
        check(lib().opal_train_batch(self._h, b.data.data_ptr(), b.nbytes, b.off.data_ptr(), b.len.data_ptr(),
                                     labels.data_ptr(), b.n, int(reverse), eta, l1, l2, int(sequential),
                                     pred.data_ptr() if want_pred else None), self._h)
        return pred[:b.n] if want_pred else None

    def get_weights(self, to_host=False):
        """The 2^bits compact weight table (torch cuda tensor, or numpy when to_host)."""
        t = self.torch
        n = 1 << self.bits
        if to_host:
            out = np.empty(n, dtype=np.float32)
            check(lib().opal_get_weights(self._h, out.ctypes.data, 0), self._h)
            return out
        out = t.empty(n, dtype=t.float32, device=self.device)
        check(lib().opal_get_weights(self._h, out.data_ptr(), 1), self._h)
        return out

    def get_accumulators(self):
        """The learner's 2^bits AdaGrad accumulators (torch cuda tensor): the resume state a `--save_resume` model carries
        (reference opal.py:310)."""
        t = self.torch
        out = t.empty(1 << self.bits, dtype=t.float32, device=self.device)
        check(lib().opal_get_accumulators(self._h, out.data_ptr(), 1), self._h)
        return out

    def load_accumulators_sparse(self, idx: np.ndarray, val: np.ndarray):
        """Installs saved accumulators next to the loaded weights; train_begin(K, bits, resume=True) then continues the rule."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        assert idx.size == val.size
        check(lib().opal_load_accumulators_sparse(self._h, idx.ctypes.data, val.ctypes.data, idx.size), self._h)

    def load_weights_sparse(self, idx: np.ndarray, val: np.ndarray, K: int, bits: int):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        assert idx.size == val.size
        check(lib().opal_load_weights_sparse(self._h, idx.ctypes.data, val.ctypes.data, idx.size, K, bits), self._h)
        self.K, self.bits = K, bits

    # ------------------------------------------------------------------ evaluation (reference opal.py:111-131)
    def eval_counts(self, pred, ref, n_classes: int):
        """pred, ref: int32 class ids (numpy or cuda tensors), ref in [0, n_classes). Returns (correct, total) numpy uint64."""
        t = self.torch
        if isinstance(pred, np.ndarray):
            pred = t.from_numpy(np.ascontiguousarray(pred, dtype=np.int32)).to(self.device)
        if isinstance(ref, np.ndarray):
            ref = t.from_numpy(np.ascontiguousarray(ref, dtype=np.int32)).to(self.device)
        n = min(pred.numel(), ref.numel())
        cor = t.empty(n_classes, dtype=t.int64, device=self.device)
        tot = t.empty(n_classes, dtype=t.int64, device=self.device)
        check(lib().opal_eval_counts(self._h, pred.data_ptr(), ref.data_ptr(), n, n_classes, cor.data_ptr(), tot.data_ptr()), self._h)
        return cor.cpu().numpy().view(np.uint64), tot.cpu().numpy().view(np.uint64)

    # ------------------------------------------------------------------ GPU-side FASTA ingest
    def ingest_fasta(self, text) -> DeviceBatch:
        """text: bytes / numpy uint8 / path of a FASTA file / a cuda uint8 tensor.  The file's bytes go to HBM as they are; record splitting,
        newline stripping, the '+' rule and 2-bit packing happen on the device (opal_ingest_fasta)."""
        t = self.torch
        if t.is_tensor(text):                                     # the file's bytes, already in HBM
            assert text.is_cuda and text.dtype == t.uint8 and text.is_contiguous()
            d_text, n = text, int(text.numel())
            if n == 0:
                return self.upload(PackedReads.from_sequences([]))
        else:
            if isinstance(text, str):
                arr = np.fromfile(text, dtype=np.uint8)
            elif isinstance(text, (bytes, bytearray)):
                arr = np.frombuffer(bytes(text), dtype=np.uint8)
            else:
                arr = np.ascontiguousarray(text, dtype=np.uint8)
            n = int(arr.size)
            if n == 0:
                return self.upload(PackedReads.from_sequences([]))
            d_text = t.from_numpy(arr.copy() if not arr.flags.writeable else arr).to(self.device)
        # the call reads the text once and needs no sizes in advance: buffers for records of >= 96 bytes of text each and
        # for a few raw records first; a call that finds them too small reports what the text needs (OPAL_ERANGE).  So that a
        # large file of unusually short records is not read twice, the first guess follows the header density of the text's
        # first MiB (one small copy to the host; a retry stays the safety net for texts whose start is not typical)
        nrec, nb = C.c_uint64(), C.c_uint64()
        max_rec, cap = n // 96 + 1024, n // 4 + n // 32 + 4096
        if n >= (64 << 20):
            head = d_text[:1 << 20].cpu().numpy()
            headers = int(np.count_nonzero((head[1:] == 62) & ((head[:-1] == 10) | (head[:-1] == 13)))) + int(head[0] == 62)
            max_rec = max(max_rec, int(headers * (n / head.size) * 1.25) + 1024)
        for _ in range(3):
            off = t.empty(max_rec, dtype=t.int64, device=self.device)
            ln = t.empty(max_rec, dtype=t.int32, device=self.device)
            reads = t.empty(cap, dtype=t.uint8, device=self.device)
            rc = lib().opal_ingest_fasta(self._h, d_text.data_ptr(), n, reads.data_ptr(), cap, off.data_ptr(), ln.data_ptr(), max_rec,
                                         C.byref(nrec), C.byref(nb))
            if rc != _lib.OPAL_ERANGE:
                break
            max_rec, cap = max(max_rec, int(nrec.value), 1), max(cap, int(nb.value))
        check(rc, self._h)
        if int(nrec.value) * 2 < max_rec:                          # hand the unused part of the optimistic buffers back
            off, ln = off[:int(nrec.value)].clone(), ln[:int(nrec.value)].clone()
        return DeviceBatch(reads, off, ln, int(nrec.value), max(int(nb.value), 1))
