"""ctypes face of oracle/libopal_oracle.so plus the small pure-Python/numpy restatements.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (opal_amd/) must never import this module.

Reference lines restated here (under /root/reference):
  util/fasta2skm.py:120-134  create_pattern_getters -> parse_patterns
  util/fasta2skm.py:19-44    update_dictionary      -> update_dictionary
  util/ldpc.py:10-37,66-92   ldpc / write_out       -> ldpc_matrix / patterns_text
  opal.py:76-88              vw_class_to_taxid      -> vw_class_to_taxid
  opal.py:111-138            evaluate_predictions   -> evaluate
Everything else forwards to opal_oracle.c (see its header for the per-function citations).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libopal_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "opal_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, u32p, u64p, i32p, f32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_float))
        L.orc_murmur3_32.restype = C.c_uint32
        L.orc_murmur3_32.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32]
        L.orc_vw_hashstring.restype = C.c_uint64
        L.orc_vw_hashstring.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
        L.orc_fasta_read.restype = C.c_void_p
        L.orc_fasta_read.argtypes = [C.c_char_p, C.c_uint64, C.c_int]
        L.orc_fasta_count.restype = C.c_uint64
        L.orc_fasta_count.argtypes = [C.c_void_p]
        for nm in ("orc_fasta_name", "orc_fasta_seq"):
            getattr(L, nm).restype = C.c_void_p
            getattr(L, nm).argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.orc_fasta_free.argtypes = [C.c_void_p]
        L.orc_reverse_complement.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        L.orc_num_features.restype = C.c_uint64
        L.orc_num_features.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.orc_vw_line.restype = C.c_uint64
        L.orc_vw_line.argtypes = [C.c_char_p, C.c_uint64, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_feature_hashes.restype = C.c_uint64
        L.orc_feature_hashes.argtypes = [C.c_char_p, C.c_uint64, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int, u64p]
        L.orc_predict.restype = C.c_uint64
        L.orc_predict.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                  i32p, C.c_void_p, C.c_int]
        L.orc_features_batch.restype = C.c_uint64
        L.orc_features_batch.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, u64p, C.c_void_p, C.c_int]
        L.orc_train.restype = None
        L.orc_train.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, i32p, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, i32p]
        L.orc_train_groups.restype = None
        L.orc_train_groups.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, i32p, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_uint64, C.c_int, i32p]
        L.orc_rng_new.restype = C.c_void_p
        L.orc_rng_free.argtypes = [C.c_void_p]
        L.orc_rng_seed_limbs.argtypes = [C.c_void_p, u32p, C.c_int]
        L.orc_rng_randbelow.restype = C.c_uint64
        L.orc_rng_randbelow.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_rng_shuffle.argtypes = [C.c_void_p, u64p, C.c_uint64]
        L.orc_drawfrag.restype = C.c_uint64
        L.orc_drawfrag.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_uint64, C.c_double, u64p, u32p, C.c_uint64]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


# ------------------------------------------------------------------ hashing
def murmur3_32(data: bytes, seed=0) -> int:
    return int(lib().orc_murmur3_32(data, len(data), seed))


def vw_hashstring(data: bytes, seed=0) -> int:
    return int(lib().orc_vw_hashstring(data, len(data), seed))


# ------------------------------------------------------------------ FASTA
def fasta_read(text: bytes, universal_newlines=True):
    """[(name, seq)] as bytes; mirrors fasta_functions.fasta_reader over open(path,'r') (universal newlines)."""
    L = lib()
    h = L.orc_fasta_read(text, len(text), 1 if universal_newlines else 0)
    try:
        out = []
        n = C.c_uint64()
        for i in range(L.orc_fasta_count(h)):
            p = L.orc_fasta_name(h, i, C.byref(n))
            name = C.string_at(p, n.value)
            p = L.orc_fasta_seq(h, i, C.byref(n))
            out.append((name, C.string_at(p, n.value)))
        return out
    finally:
        L.orc_fasta_free(h)


def reverse_complement(s: bytes) -> bytes:
    out = C.create_string_buffer(len(s) + 1)
    lib().orc_reverse_complement(s, len(s), out)
    return out.raw[:len(s)]


# ------------------------------------------------------------------ patterns (fasta2skm.py:120-134)
def parse_patterns(lines, kmer):
    """lines: list of str (file.readlines()) or None. Returns int32 array [P, t]."""
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


def vw_line(seq: bytes, k, pos, reverse, label="1") -> bytes:
    L = lib()
    P, t = pos.shape
    lab = str(label).encode()
    n = L.orc_vw_line(seq, len(seq), k, _p(pos, C.c_int32), P, t, int(reverse), lab, None)
    buf = C.create_string_buffer(n + 1)
    L.orc_vw_line(seq, len(seq), k, _p(pos, C.c_int32), P, t, int(reverse), lab, buf)
    return buf.raw[:n]


def feature_hashes(seq: bytes, k, pos, reverse, hash_all=False):
    L = lib()
    P, t = pos.shape
    nf = L.orc_num_features(len(seq), k, P, int(reverse))
    out = np.zeros(max(nf, 1), dtype=np.uint64)
    got = L.orc_feature_hashes(seq, len(seq), k, _p(pos, C.c_int32), P, t, int(reverse), int(hash_all), _p(out, C.c_uint64))
    assert got == nf
    return out[:nf]


def _concat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8)
    return blob, off


def features_batch(seqs, k, pos, reverse, bits, hash_all=False, nthreads=1):
    """seqs: list of bytes, or (blob uint8 array, off uint64 array). Returns (idx uint32, off uint64[n+1])."""
    L = lib()
    blob, off = seqs if isinstance(seqs, tuple) else _concat(seqs)
    n = len(off) - 1
    P, t = pos.shape
    out_off = np.zeros(n + 1, dtype=np.uint64)
    tot = L.orc_features_batch(blob.ctypes.data, _p(off, C.c_uint64), n, k, _p(pos, C.c_int32), P, t, int(reverse),
                               int(hash_all), bits, _p(out_off, C.c_uint64), None, nthreads)
    idx = np.zeros(max(tot, 1), dtype=np.uint32)
    L.orc_features_batch(blob.ctypes.data, _p(off, C.c_uint64), n, k, _p(pos, C.c_int32), P, t, int(reverse),
                         int(hash_all), bits, _p(out_off, C.c_uint64), idx.ctypes.data, nthreads)
    return idx[:tot], out_off


def predict(seqs, k, pos, reverse, w, K, bits, add_constant=True, clip=True, min_label=-1.0, max_label=1.0,
            hash_all=False, want_scores=True, nthreads=1):
    """Returns (classes int32[n] (1-based), scores float32[n,K] or None, total_features)."""
    L = lib()
    blob, off = seqs if isinstance(seqs, tuple) else _concat(seqs)
    n = len(off) - 1
    P, t = pos.shape
    assert w.dtype == np.float32 and w.size == (1 << bits) and w.flags.c_contiguous
    cls = np.zeros(max(n, 1), dtype=np.int32)
    sc = np.zeros((max(n, 1), K), dtype=np.float32) if want_scores else None
    tot = L.orc_predict(blob.ctypes.data, _p(off, C.c_uint64), n, k, _p(pos, C.c_int32), P, t, int(reverse), int(hash_all),
                        w.ctypes.data, K, bits, int(add_constant), int(clip), min_label, max_label,
                        _p(cls, C.c_int32), sc.ctypes.data if want_scores else None, nthreads)
    return cls[:n], (sc[:n] if want_scores else None), int(tot)


def train(seqs, labels, k, pos, reverse, w, g2, K, bits, eta=0.5, l1=0.0, l2=0.0, min_label=-1.0, max_label=1.0,
          hash_all=False):
    """Sequential restatement of the trainer (updates w and g2 in place). Returns progressive predictions."""
    L = lib()
    blob, off = seqs if isinstance(seqs, tuple) else _concat(seqs)
    n = len(off) - 1
    P, t = pos.shape
    assert w.dtype == np.float32 and g2.dtype == np.float32 and w.size == g2.size == (1 << bits)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    pred = np.zeros(max(n, 1), dtype=np.int32)
    L.orc_train(blob.ctypes.data, _p(off, C.c_uint64), n, k, _p(pos, C.c_int32), P, t, int(reverse), int(hash_all),
                w.ctypes.data, g2.ctypes.data, K, bits, _p(lab, C.c_int32), eta, l1, l2, min_label, max_label,
                _p(pred, C.c_int32))
    return pred[:n]


def train_groups(seqs, labels, k, pos, reverse, w, g2, K, bits, group, pshift, eta=0.5, l1=0.0, l2=0.0, min_label=-1.0,
                 max_label=1.0, hash_all=False):
    """CPU statement of the deterministic group-synchronous trainer (opal_dtrain.hip); updates w and g2 in place."""
    L = lib()
    blob, off = seqs if isinstance(seqs, tuple) else _concat(seqs)
    n = len(off) - 1
    P, t = pos.shape
    assert w.dtype == np.float32 and g2.dtype == np.float32 and w.size == g2.size == (1 << bits)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    pred = np.zeros(max(n, 1), dtype=np.int32)
    L.orc_train_groups(blob.ctypes.data, _p(off, C.c_uint64), n, k, _p(pos, C.c_int32), P, t, int(reverse), int(hash_all),
                       w.ctypes.data, g2.ctypes.data, K, bits, _p(lab, C.c_int32), eta, l1, l2, min_label, max_label,
                       int(group), int(pshift), _p(pred, C.c_int32))
    return pred[:n]


def train_pshift(K, bits):
    """log2 of the trainer's partition size (opal::dtrain_pshift): >= K slots, at most 2^14 partitions."""
    ps = 0
    while (1 << ps) < K:
        ps += 1
    if bits > 14:
        ps = max(ps, bits - 14)
    return min(ps, bits)


def max_threads():
    return int(lib().orc_max_threads())


# ------------------------------------------------------------------ CPython random
class PyRandom:
    """random.seed(int) / randint / shuffle, bit-exact to CPython (drawfrag.py:24,38; opal.py:363)."""

    def __init__(self, seed=None):
        self._h = lib().orc_rng_new()
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        try:
            lib().orc_rng_free(self._h)
        except Exception:
            pass

    def seed(self, seed: int):
        a = abs(int(seed))
        limbs = []
        while a:
            limbs.append(a & 0xFFFFFFFF)
            a >>= 32
        if not limbs:
            limbs = [0]
        arr = (C.c_uint32 * len(limbs))(*limbs)
        lib().orc_rng_seed_limbs(self._h, arr, len(limbs))

    def randbelow(self, n: int) -> int:
        return int(lib().orc_rng_randbelow(self._h, n))

    def randint(self, a: int, b: int) -> int:
        return a + self.randbelow(b - a + 1)

    def shuffle_perm(self, n: int):
        perm = np.arange(n, dtype=np.uint64)
        lib().orc_rng_shuffle(self._h, _p(perm, C.c_uint64), n)
        return perm

    def drawfrag(self, genome_len, size, coverage):
        gl = np.ascontiguousarray(np.asarray(genome_len, dtype=np.uint64))
        cap = int(sum(int(np.ceil(coverage * int(x) / size)) + 2 for x in gl if x >= size)) + 16
        pos = np.zeros(cap, dtype=np.uint64)
        gen = np.zeros(cap, dtype=np.uint32)
        n = lib().orc_drawfrag(self._h, _p(gl, C.c_uint64), len(gl), size, float(coverage), _p(pos, C.c_uint64),
                               _p(gen, C.c_uint32), cap)
        assert n <= cap
        return pos[:n], gen[:n]


def drawfrag_files(fasta_text: bytes, taxid_text: str, size, coverage, seed):
    """Restates drawfrag.py:15-49 end to end on in-memory files. Returns (out_fasta, out_gi2taxid) as str."""
    recs = fasta_read(fasta_text, universal_newlines=True)
    tlines = taxid_text.split("\n")
    rng = PyRandom(seed if seed else None)  # drawfrag.py:23: `if args.seed:` (0/None -> unseeded)
    pos, gen = rng.drawfrag([len(s) for _, s in recs], size, coverage)
    of, og = [], []
    for i, (p, g) in enumerate(zip(pos.tolist(), gen.tolist())):
        name, seq = recs[g]
        tl = tlines[g] if g < len(tlines) else ""
        of.append(">%d\n%s\n" % (i + 1, seq[p:p + size].decode("latin-1")))
        og.append("%s\t%s\n" % (name.decode("latin-1").split()[0], tl))
    return "".join(of), "".join(og)


# ------------------------------------------------------------------ dico (fasta2skm.py:19-44)
def update_dictionary(label_lines, dico_text):
    """label_lines: iterable of str lines; dico_text: existing file contents or None. Returns (dict, new_text)."""
    label2vwid = {}
    vwidset = set()
    if dico_text is not None:
        for line in dico_text.splitlines(True):
            label, vwid_str = line.strip().split()[:2]
            label2vwid[label] = int(vwid_str)
            vwidset.add(int(vwid_str))
    for line in label_lines:
        label = line.rstrip("\n")
        if label not in label2vwid:
            vwid = 1 if not vwidset else max(vwidset) + 1
            label2vwid[label] = vwid
            vwidset.add(vwid)
    return label2vwid, "".join("%s\t%d\n" % (a, b) for a, b in label2vwid.items())


# ------------------------------------------------------------------ ldpc (ldpc.py:10-37, 66-92)
def ldpc_matrix(k, t, _m, permutations):
    """permutations: the (w-1) column permutations the reference draws with np.random.permutation(k)."""
    if k % t != 0:
        raise ValueError("k should be multiple of t!")
    m = (int(np.ceil(_m * 1.0 / (k / t)) + 1)) * (k // t)
    w = m * t // k
    Hb = np.zeros((m // w, k), dtype=bool)
    for i in range(m // w):
        Hb[i, i * t:(i + 1) * t] = True
    blocks = [Hb]
    assert len(permutations) == w - 1
    for perm in permutations:
        blocks.append(Hb[:, np.asarray(perm)])
    return np.vstack(blocks)


def patterns_text(H, _m):
    k = H.shape[1]
    t = int(H[0].sum())
    m = (int(np.ceil(_m * 1.0 / (k / t)) + 1)) * (k // t)
    w = m * t // k
    s = "%d %d\n" % (_m + 1, t)
    s += "".join("%d " % j for j in range(t)) + "\n"
    st = m // w
    for i in range(_m):
        s += "".join("%d " % j for j in range(k) if H[st + i, j]) + "\n"
    return s


# ------------------------------------------------------------------ post-processing (opal.py:76-88, 111-138)
def vw_class_to_taxid(pred_lines, dico_text):
    dico = {}
    for line in dico_text.splitlines():
        txid, vwid = line.strip().split()[:2]
        dico[vwid] = txid
    return ["%s\n" % dico[str(int(float(l.strip())))] for l in pred_lines]


def evaluate(ref, pred):
    """micro / macro / median per-reference-class accuracy (opal.py:111-131); files of unequal length make the
    reference's pandas frame raise ValueError (opal.py:123).  Pinned by tests/golden/opal_helpers.json."""
    if len(ref) != len(pred):
        raise ValueError("All arrays must be of the same length")
    correct = [x == y for x, y in zip(pred, ref)]
    by = {}
    for r, c in zip(ref, correct):
        by.setdefault(r, []).append(c)
    species = [float(np.mean(v)) for _, v in sorted(by.items())]
    return float(np.mean(correct)), float(np.mean(species)), float(np.median(species))
