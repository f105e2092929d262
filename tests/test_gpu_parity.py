"""GPU parity: the HIP path (through the C-ABI of include/opal_hip.h) against the CPU oracle and the
reference-generated golden vectors.  Bit-exact everywhere: indices/offsets/classes are integers, and the
fp32 scores are accumulated in the reference's order so they must match the oracle bit for bit.

VW-side rules (addressing, constant, clip, argmax) are checked against the oracle's restatement only:
"parity unpinned" (no vw binary exists anywhere).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def amd():
    import opal_amd
    opal_amd.lib()
    return opal_amd


@pytest.fixture(scope="module")
def eng(amd):
    e = amd.Engine(0)
    yield e
    e.close()


def _patterns(amd, golden_dir, fname, k):
    if fname is None:
        return amd.parse_patterns(None, k)
    with open(os.path.join(golden_dir, fname)) as f:
        return amd.parse_patterns(f.readlines(), k)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def _rand_seqs(rng, n, L):
    return [bytes(rng.choice(ACGT, size=L)) for _ in range(n)]


# ------------------------------------------------------------------------------ emit mode vs golden
def test_emit_indices_match_reference_golden(amd, eng, oracle, golden_dir):
    cases = _load(golden_dir, "skm.json")
    for c in cases:
        pos = _patterns(amd, golden_dir, c["pattern"], c["k"])
        if pos.shape[1] > 64:
            continue
        eng.set_patterns(pos, c["k"])
        seqs = [s for _, s in oracle.fasta_read(c["fasta"].encode("latin-1"))]
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        want_off = np.concatenate([[0], np.cumsum([len(h) for h in c["h32"]])]).astype(np.int64)
        for bits in (18, 31, 32):
            mask = (1 << bits) - 1
            # golden hashes are pure murmur (sklearn): compare with --hash all ...
            eng.set_vw_options(hash_all=True)
            idx, off = eng.features(batch, c["reverse"], bits)
            assert np.array_equal(off.cpu().numpy(), want_off), c["name"]
            want = np.array([h & mask for hs in c["h32"] for h in hs], dtype=np.uint32)
            assert np.array_equal(_u32(idx), want), (c["name"], bits)
            # ... and the VW default (--hash strings, digit rule) with the oracle
            eng.set_vw_options(hash_all=False)
            idx2, _ = eng.features(batch, c["reverse"], bits)
            oidx, ooff = oracle.features_batch(seqs, c["k"], pos, c["reverse"], min(bits, 32), hash_all=False)
            assert np.array_equal(_u32(idx2), oidx), (c["name"], bits, "digit rule")
    eng.set_vw_options()


def test_vw_text_lines_rebuilt_from_gpu_names(amd, eng, oracle, golden_dir):
    """--dump-vw-text path: names gathered on the GPU + host formatting == the reference's lines, byte for byte."""
    from opal_amd import fasta2skm
    for c in _load(golden_dir, "skm.json"):
        pos = _patterns(amd, golden_dir, c["pattern"], c["k"])
        if pos.shape[1] > 64:
            continue
        eng.set_patterns(pos, c["k"])
        seqs = [s for _, s in oracle.fasta_read(c["fasta"].encode("latin-1"))]
        lines = fasta2skm.vw_text_lines(eng, seqs, c["reverse"], labels=None)
        assert [l.decode("latin-1") for l in lines] == c["lines"], c["name"]


def test_corpus_digests_direct_from_the_reference(amd, eng, golden_dir):
    """GPU output against the REFERENCE's output at fuzz scale, with no oracle in between: ~2100 fragments per geometry
    (P in {2, 5, 9}, -r on/off; lengths 0..600 so both emit kernels run; ACGT, mixed case + N, IUPAC) went through the
    reference's main_generator (tests/golden/make_golden.py gen_skm_corpus); the digests of its VW text and of the
    sklearn-hashed index stream are rebuilt here from opal_feature_names / opal_features."""
    import hashlib
    import sys
    from opal_amd import fasta2skm
    sys.path.insert(0, golden_dir)
    import corpus_gen
    eng.set_vw_options(hash_all=True)          # the golden hashes are pure murmur (sklearn); no name of the corpus is all digits
    for c in _load(golden_dir, "skm_corpus.json"):
        pos = _patterns(amd, golden_dir, c["pattern"], c["k"])
        eng.set_patterns(pos, c["k"])
        seqs = corpus_gen.fragments(c["name"], c["per_alphabet"])
        assert len(seqs) == c["fragments"]
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        idx, off = eng.features(batch, c["reverse"], 32)
        idx, off = _u32(idx), off.cpu().numpy()
        assert idx.size == c["features"], c["name"]
        assert hashlib.sha256(np.diff(off).astype("<u8").tobytes()).hexdigest() == c["counts_sha256"], c["name"]
        assert hashlib.sha256(idx.astype("<u4").tobytes()).hexdigest() == c["idx32_sha256"], c["name"]
        idx31, _ = eng.features(batch, c["reverse"], 31)
        assert hashlib.sha256(_u32(idx31).astype("<u4").tobytes()).hexdigest() == c["idx31_sha256"], c["name"]
        text = hashlib.sha256()
        nbytes = 0
        for line in fasta2skm.vw_text_lines(eng, seqs, c["reverse"], labels=None, chunk=512):
            text.update(line)
            nbytes += len(line)
        assert nbytes == c["text_bytes"], c["name"]
        assert text.hexdigest() == c["text_sha256"], c["name"]
    eng.set_vw_options()
    eng.sync()


# ------------------------------------------------------------------------------ fused predict vs oracle
@pytest.mark.parametrize("K,bits", [(10, 12), (7, 16), (1, 10), (64, 14), (356, 12), (356, 20), (1564, 20), (1563, 18),
                                    (2050, 18), (4100, 16), (130, 31 - 9)])
def test_predict_matches_oracle(amd, eng, oracle, golden_dir, K, bits):
    rng = np.random.default_rng(K * 131 + bits)
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    seqs = _rand_seqs(rng, 37, 200)
    seqs[1] = seqs[1].lower()                      # raw path, case quirk of reverse_complement
    seqs[2] = seqs[2][:17] + b"N" + seqs[2][18:]   # raw path
    seqs[3] = seqs[3][:63]                         # no features: only the constant
    seqs[4] = seqs[4][:64]                         # a single window
    seqs[5] = b""
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    for reverse in (True, False):
        for (addc, clip) in ((True, True), (True, False), (False, False)):
            eng.set_vw_options(add_constant=addc, clip=clip, min_label=-0.25, max_label=0.3)
            cls, sc = eng.predict(batch, reverse, want_scores=True)
            ocls, osc, _ = oracle.predict(seqs, 64, pos, reverse, w, K, bits, add_constant=addc, clip=clip,
                                          min_label=-0.25, max_label=0.3)
            assert np.array_equal(sc.cpu().numpy(), osc), (K, bits, reverse, addc, clip)
            assert np.array_equal(cls.cpu().numpy(), ocls), (K, bits, reverse, addc, clip)
            cls2, none = eng.predict(batch, reverse, want_scores=False)
            assert none is None and np.array_equal(cls2.cpu().numpy(), ocls)
    eng.set_vw_options()


@pytest.mark.parametrize("K", [1, 3, 10, 64, 130, 356, 1563, 1564, 2050])
def test_streaming_load_variants_match_oracle(amd, eng, oracle, golden_dir, K):
    """Tables of 4 GiB and more are scored by the kernel variants with non-temporal row loads; here they are forced
    onto a small table (opal_set_tuning) so that every (VEC, NV) variant is compared with the oracle bit for bit."""
    rng = np.random.default_rng(K)
    bits = 18
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    seqs = _rand_seqs(rng, 50, 200) + [b"", b"ACGT" * 10, bytes(rng.choice(ACGT, size=900))]
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    eng.set_vw_options()
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    ocls, osc, _ = oracle.predict(seqs, 64, pos, True, w, K, bits)
    try:
        for nt in (1, 0):
            eng.set_tuning(predict_nt=nt)
            cls, sc = eng.predict(batch, True, want_scores=True)
            assert np.array_equal(sc.cpu().numpy(), osc) and np.array_equal(cls.cpu().numpy(), ocls), (K, nt)
    finally:
        eng.set_tuning()


def test_adopted_table_scores_in_place(amd, oracle, golden_dir):
    """opal_adopt_weights_device: the ctx scores from the caller's padded buffer (no second copy); same scores as the
    oracle incl. rows that wrap at 2^bits, the caller's table stays as it was, too-small buffers are refused, and a later
    opal_load_weights / train_begin leaves the caller's buffer alone."""
    import torch
    rng = np.random.default_rng(21)
    K, bits = 356, 12                       # 4096 slots: most rows wrap
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    e2 = amd.Engine(0)
    e2.set_patterns(pos, 64)
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    buf = torch.full(((1 << bits) + K + 4 + 7,), 123.0, dtype=torch.float32, device=e2.device)
    buf[:1 << bits] = torch.from_numpy(w).to(e2.device)
    e2.adopt_weights(buf, K, bits)
    seqs = _rand_seqs(rng, 40, 200)
    batch = e2.upload(amd.PackedReads.from_sequences(seqs))
    cls, sc = e2.predict(batch, True, want_scores=True)
    ocls, osc, _ = oracle.predict(seqs, 64, pos, True, w, K, bits)
    assert np.array_equal(sc.cpu().numpy(), osc) and np.array_equal(cls.cpu().numpy(), ocls)
    assert np.array_equal(buf[:1 << bits].cpu().numpy(), w)
    assert np.array_equal(buf[1 << bits:(1 << bits) + K + 4].cpu().numpy(), np.resize(w, K + 4))      # the mirror
    assert float(buf[-1]) == 123.0
    small = torch.zeros((1 << bits) + K, dtype=torch.float32, device=e2.device)
    from opal_amd import _lib
    assert _lib.lib().opal_adopt_weights_device(e2._h, small.data_ptr(), 1 << bits, small.numel(), K, bits) == _lib.OPAL_EINVAL
    assert b"K + 4" in _lib.lib().opal_last_error(e2._h)
    e2.load_weights(w * 2, K, bits)                               # the ctx goes back to a table of its own
    cls2, sc2 = e2.predict(batch, True, want_scores=True)
    assert np.array_equal(sc2.cpu().numpy(), oracle.predict(seqs, 64, pos, True, w * 2, K, bits)[1])
    assert np.array_equal(buf[:1 << bits].cpu().numpy(), w)
    e2.adopt_weights(buf, K, bits)
    e2.train_begin(K, bits)                                       # a fresh table for the learner, not the caller's
    assert float(e2.get_weights().abs().max()) == 0.0 and np.array_equal(buf[:1 << bits].cpu().numpy(), w)
    e2.close()


@pytest.mark.parametrize("pattern,k", [("patterns_k64_t16_m4_s7.txt", 64), ("patterns_k64_t16_m8_s7.txt", 64),
                                       ("patterns_k32_t8_m2_s11.txt", 32), ("patterns_hand_k16_t4_p11.txt", 16),
                                       (None, 20)])
def test_predict_other_geometries_and_long_reads(amd, eng, oracle, golden_dir, pattern, k):
    rng = np.random.default_rng(11)
    pos = _patterns(amd, golden_dir, pattern, k)
    eng.set_patterns(pos, k)
    seqs = _rand_seqs(rng, 9, 200) + [bytes(rng.choice(ACGT, size=L)) for L in (5000, 1537, 449, 450, 451, 3000)]
    seqs[-1] = seqs[-1].lower()  # long raw read
    K, bits = 356, 18
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    cls, sc = eng.predict(batch, True, want_scores=True)
    ocls, osc, _ = oracle.predict(seqs, k, pos, True, w, K, bits)
    assert np.array_equal(sc.cpu().numpy(), osc)
    assert np.array_equal(cls.cpu().numpy(), ocls)
    idx, off = eng.features(batch, True, 31)
    oidx, ooff = oracle.features_batch(seqs, k, pos, True, 31)
    assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff)
    assert np.array_equal(_u32(idx), oidx)


def test_tie_break_and_clip_saturation(amd, eng, oracle, golden_dir):
    """All-equal scores: VW's strict '>' keeps the lowest class; saturating clip makes many ties."""
    rng = np.random.default_rng(3)
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    seqs = _rand_seqs(rng, 16, 200)
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    K, bits = 356, 14
    eng.load_weights(np.zeros(1 << bits, dtype=np.float32), K, bits)
    cls, _ = eng.predict(batch, True)
    assert cls.cpu().numpy().tolist() == [1] * 16
    w = (rng.standard_normal(1 << bits) * 5.0).astype(np.float32)  # everything saturates at +-1
    eng.load_weights(w, K, bits)
    eng.set_vw_options()
    cls, sc = eng.predict(batch, True, want_scores=True)
    ocls, osc, _ = oracle.predict(seqs, 64, pos, True, w, K, bits)
    assert np.array_equal(sc.cpu().numpy(), osc) and np.array_equal(cls.cpu().numpy(), ocls)
    assert (np.abs(osc) == 1.0).mean() > 0.5


def test_empty_batch_and_error_paths(amd, eng, golden_dir):
    from opal_amd._lib import OpalError
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    eng.load_weights(np.zeros(1 << 10, dtype=np.float32), 4, 10)
    batch = eng.upload(amd.PackedReads.from_sequences([]))
    cls, _ = eng.predict(batch, True)
    assert cls.numel() == 0
    idx, off = eng.features(batch, True, 31)
    assert idx.numel() == 0 and off.cpu().numpy().tolist() == [0]
    with pytest.raises(OpalError):
        eng.set_patterns(np.array([[0, 1, 70, 3]], dtype=np.int32), 64)      # position >= k
    with pytest.raises(OpalError):
        eng.set_patterns(np.zeros((1, 5), dtype=np.int32), 64)               # k % t != 0 (fasta2skm.py:125)
    e2 = amd.Engine(0)
    with pytest.raises(OpalError):
        e2.predict(batch, True)                                              # nothing configured
    e2.close()


# ------------------------------------------------------------------------------ full-size properties
def test_properties_at_scale(amd, eng, oracle, golden_dir):
    """Config-2-shaped batch (200 bp, k=64, t=16, numHash=1, -r, K=356): size-independent properties."""
    rng = np.random.default_rng(2024)
    n, L, K, bits = 200_000, 200, 356, 24
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    pr = amd.PackedReads.from_codes(codes)
    batch = eng.upload(pr)
    # (a) emit: checksum of all indices against the multi-threaded oracle, and exact offsets
    blob = ACGT[codes].reshape(-1)
    soff = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    idx, off = eng.features(batch, True, 31)
    oidx, ooff = oracle.features_batch((blob, soff), 64, pos, True, 31, nthreads=0)
    assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff)
    gi = _u32(idx)
    assert int(gi.astype(np.uint64).sum()) == int(oidx.astype(np.uint64).sum())
    assert np.array_equal(gi, oidx)
    # (b) reverse-complement symmetry of the feature multiset: features(rc(x)) = [second half | first half] of features(x)
    rc = (3 - codes[:, ::-1]).astype(np.uint8)
    idx_rc, _ = eng.features(eng.upload(amd.PackedReads.from_codes(np.ascontiguousarray(rc))), True, 31)
    F = 2 * (L - 64 + 1) * pos.shape[0]
    a = gi.reshape(n, F)
    b = _u32(idx_rc).reshape(n, F)
    assert np.array_equal(b[:, :F // 2], a[:, F // 2:]) and np.array_equal(b[:, F // 2:], a[:, :F // 2])
    # (c) predict: sharding invariance (any split of the batch gives the same per-fragment answers) ...
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    eng.set_vw_options(clip=False)
    cls, sc = eng.predict(batch, True, want_scores=True)
    cut = 77_777
    b1 = eng.upload(amd.PackedReads.from_codes(codes[:cut]))
    b2 = eng.upload(amd.PackedReads.from_codes(codes[cut:]))
    c1, s1 = eng.predict(b1, True, want_scores=True)
    c2, s2 = eng.predict(b2, True, want_scores=True)
    assert np.array_equal(np.concatenate([c1.cpu().numpy(), c2.cpu().numpy()]), cls.cpu().numpy())
    assert np.array_equal(np.concatenate([s1.cpu().numpy(), s2.cpu().numpy()]), sc.cpu().numpy())
    # ... exact scaling by a power of two, and agreement with the oracle on a bounded sample
    eng.load_weights(w * np.float32(4.0), K, bits)
    cls4, sc4 = eng.predict(batch, True, want_scores=True)
    assert np.array_equal(sc4.cpu().numpy(), sc.cpu().numpy() * np.float32(4.0))
    assert np.array_equal(cls4.cpu().numpy(), cls.cpu().numpy())
    m = 4000
    ocls, osc, _ = oracle.predict((blob[:m * L + 1], soff[:m + 1]), 64, pos, True, w, K, bits, clip=False, nthreads=0)
    assert np.array_equal(sc.cpu().numpy()[:m], osc) and np.array_equal(cls.cpu().numpy()[:m], ocls)
    eng.set_vw_options()


def test_correctness_run_with_injected_n_and_lowercase(amd, eng, oracle, golden_dir):
    """the throughput inputs are upper-case ACGT; this is the separate correctness run with 0.1 % of the
    bases replaced by N / lower-case letters, so a batch mixes 2-bit packed fragments (table-driven hash) with text
    fragments (register murmur, the reference's RC quirk).  Emit and predict must equal the oracle on every fragment."""
    rng = np.random.default_rng(77)
    n, L, K, bits = 60_000, 200, 356, 22
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    text = ACGT[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    hit = rng.random((n, L)) < 0.001
    repl = np.frombuffer(b"NnacgtRY", dtype=np.uint8)[rng.integers(0, 8, size=int(hit.sum()))]
    text[hit] = repl
    raw_rows = hit.any(axis=1)
    assert 0.1 < raw_rows.mean() < 0.3                               # ~18 % of the fragments carry at least one
    blob = np.concatenate([text.reshape(-1), np.zeros(1, dtype=np.uint8)])
    soff = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    pr = amd.PackedReads.from_blob(blob, soff)
    assert np.array_equal((pr.len & np.uint32(0x80000000)) != 0, raw_rows)
    batch = eng.upload(pr)
    idx, off = eng.features(batch, True, 31)
    oidx, ooff = oracle.features_batch((blob, soff), 64, pos, True, 31, nthreads=0)
    assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff)
    assert np.array_equal(_u32(idx), oidx)
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    eng.set_vw_options()
    cls, sc = eng.predict(batch, True, want_scores=True)
    m = 6000
    ocls, osc, _ = oracle.predict((blob[:m * L + 1], soff[:m + 1]), 64, pos, True, w, K, bits, nthreads=0)
    assert np.array_equal(sc.cpu().numpy()[:m], osc) and np.array_equal(cls.cpu().numpy()[:m], ocls)


def test_zero_copy_fragments_from_resident_genomes(amd, eng, oracle, golden_dir):
    """drawfrag on the GPU side: fragments are views (genome offset + position) into genomes packed once in HBM;
    features and predictions equal those of the explicitly cut fragments (reference util/drawfrag.py:38-39)."""
    rng = np.random.default_rng(77)
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    genomes = [bytes(rng.choice(ACGT, size=L)) for L in (5003, 777, 150, 4096)]
    genomes[1] = genomes[1][:300] + b"NNNNnnacgt" + genomes[1][310:]      # a raw genome
    gs = amd.GenomeSet(eng, amd.PackedReads.from_sequences(genomes))
    r = amd.Rng(421)
    p, g = r.sample_positions([len(x) for x in genomes], 200, 3.0)
    frags = [genomes[gi][pi:pi + 200] for pi, gi in zip(p.tolist(), g.tolist())]
    assert len(frags) > 100 and any(pi % 4 for pi in p.tolist())
    view = gs.fragments(p, g, 200)
    idx, off = eng.features(view, True, 31)
    oidx, ooff = oracle.features_batch(frags, 64, pos, True, 31)
    assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff) and np.array_equal(_u32(idx), oidx)
    K, bits = 20, 16
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    cls, sc = eng.predict(view, True, want_scores=True)
    ocls, osc, _ = oracle.predict(frags, 64, pos, True, w, K, bits)
    assert np.array_equal(sc.cpu().numpy(), osc) and np.array_equal(cls.cpu().numpy(), ocls)
    # a view that would run past the buffer is treated as empty, never read
    bad = gs.fragments(np.array([4000]), np.array([3]), 2000)
    idx2, off2 = eng.features(bad, True, 31)
    assert off2.cpu().numpy().tolist()[0] == 0


def _check_ingest(amd, eng, oracle, text: bytes):
    """The ingested batch decodes to exactly the records of the reference reader; records of upper-case ACGT are 2-bit
    packed (views into one packed stream, addressed in bases), all others raw."""
    recs = oracle.fasta_read(text, universal_newlines=True)
    b = eng.ingest_fasta(text)
    assert b.n == len(recs), (b.n, len(recs))
    if b.n == 0:
        return
    ln = b.len.cpu().numpy()[:b.n].view(np.uint32)
    want = amd.PackedReads.from_sequences([sq for _, sq in recs])
    assert np.array_equal(ln, want.len)                       # lengths and the packed/raw decision of opal_pack_reads
    got = amd.PackedReads(b.data.cpu().numpy()[:b.nbytes], b.off.cpu().numpy()[:b.n].view(np.uint64), ln)
    assert got.sequences() == [sq for _, sq in recs]
    # raw records never overlap each other or the packed stream
    spans = sorted((int(o), int(o) + int(l & 0x7fffffff)) for o, l in zip(got.off, ln) if l & 0x80000000)
    stream_end = (sum(len(sq) for _, sq in recs) + 3) // 4
    for (a0, a1), (b0, _) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert all(a0 >= stream_end and a1 <= b.nbytes for a0, a1 in spans)


def test_gpu_fasta_ingest_matches_reference_reader(amd, eng, oracle, golden_dir):
    """opal_ingest_fasta == fasta_reader over open(path,'r') + opal_pack_reads, on the golden FASTA texts (the lines
    the reference built from them are pinned in skm.json) and on adversarial synthetic files."""
    for c in _load(golden_dir, "skm.json"):
        _check_ingest(amd, eng, oracle, c["fasta"].encode("latin-1"))
    for txt, _ in _load(golden_dir, "primitives.json")["fasta_reader"]:
        _check_ingest(amd, eng, oracle, txt.encode("latin-1"))
    rng = np.random.default_rng(int(os.environ.get("OPAL_FUZZ_SEED", "123")))
    pieces = [b"ACGT", b"acgtn", b"N", b"\n", b"\r\n", b"\r", b">", b">name x\n", b"+", b"+\n", b"\n+\n", b"\n>h\n", b" ", b"@r\n"]
    for trial in range(int(os.environ.get("OPAL_FUZZ_TRIALS", "60"))):
        k = int(rng.integers(1, 60))
        txt = b"".join(pieces[int(i)] * int(rng.integers(1, 4)) for i in rng.integers(0, len(pieces), size=k))
        _check_ingest(amd, eng, oracle, txt)
    # every byte value inside lines, at line starts and next to terminators (the terminator test is a byte-value trick)
    every = b">all\n" + bytes(v for v in range(256) if v not in (10, 13)) + b"\n"
    every += b"".join(b">v%d\n" % v + bytes([v]) + b"AC" + bytes([v]) + b"\nGT" + bytes([v, v]) + b"\n" for v in range(256))
    _check_ingest(amd, eng, oracle, every)
    _check_ingest(amd, eng, oracle, every.replace(b"\n", b"\r\n"))
    parts = []
    for i in range(3000):
        L = int(rng.integers(0, 700))
        body = bytes(rng.choice(ACGT, size=L))
        if i % 7 == 0:
            body = body.lower()
        width = int(rng.integers(20, 90))
        nl = b"\r\n" if i % 5 == 0 else b"\n"
        lines = nl.join(body[j:j + width] for j in range(0, L, width))
        parts.append(b">rec%d desc" % i + nl + lines + nl)
        if i % 11 == 0:
            parts.append(b"+" + nl + b"I" * L + nl)
    big = b"junk before the first header\n" + b"".join(parts)
    assert len(big) > 200 * 4096
    _check_ingest(amd, eng, oracle, big)
    # genome-sized records (the grid-wide path of the packer: > 65 536 bases), packed and raw, between short ones
    def wrapped(body, width=70):
        return b"\n".join(body[j:j + width] for j in range(0, len(body), width)) + b"\n"
    g1 = bytes(rng.choice(ACGT, size=300_001))
    g2 = bytearray(rng.choice(ACGT, size=70_000))
    g2[65_999] = ord("N")
    g3 = bytes(rng.choice(ACGT, size=65_537))
    genomes = (b">short1\nACGTACGT\n>g1 packed genome\n" + wrapped(g1) + b">empty\n>g2 raw genome\n" + wrapped(bytes(g2), 61) +
               b">short2\nacgtn\n>g3\n" + wrapped(g3, 80) + b">edge\n" + wrapped(bytes(rng.choice(ACGT, size=65_536))))
    _check_ingest(amd, eng, oracle, genomes)
    # the C-ABI directly: an exact-fit and a too-small d_reads_out, too few record slots (OPAL_ERANGE reports what the text
    # needs), and opal_ingest_scan's counts
    import ctypes as C
    import torch
    from opal_amd._lib import OPAL_ERANGE, lib
    d_text = torch.from_numpy(np.frombuffer(big, dtype=np.uint8).copy()).to(eng.device)
    ref = eng.ingest_fasta(big)
    max_rec = big.count(b">") + 1
    off = torch.empty(max_rec, dtype=torch.int64, device=eng.device)
    ln = torch.empty(max_rec, dtype=torch.int32, device=eng.device)
    nrec, nb = C.c_uint64(), C.c_uint64()
    for cap, want_rc in ((ref.nbytes, 0), (ref.nbytes - 4, OPAL_ERANGE), (len(big) * 2, 0)):
        reads = torch.full((cap + 64,), 0xEE, dtype=torch.uint8, device=eng.device)
        rc = lib().opal_ingest_fasta(eng._h, d_text.data_ptr(), len(big), reads.data_ptr(), cap, off.data_ptr(), ln.data_ptr(), max_rec,
                                     C.byref(nrec), C.byref(nb))
        assert rc == want_rc, (cap, rc)
        assert bool((reads[cap:] == 0xEE).all())                # never writes past reads_cap
        assert nrec.value == ref.n and nb.value == ref.nbytes    # also what a failed call reports
        if rc == 0:
            got = amd.PackedReads(reads.cpu().numpy()[:nb.value], off.cpu().numpy()[:ref.n].view(np.uint64),
                                  ln.cpu().numpy()[:ref.n].view(np.uint32))
            assert got.sequences() == [sq for _, sq in oracle.fasta_read(big, universal_newlines=True)]
    rc = lib().opal_ingest_fasta(eng._h, d_text.data_ptr(), len(big), reads.data_ptr(), len(big) * 2, off.data_ptr(), ln.data_ptr(), 100,
                                 C.byref(nrec), C.byref(nb))
    assert rc == OPAL_ERANGE and nrec.value == big.count(b">rec") and nb.value == 0
    nh, ns = C.c_uint64(), C.c_uint64()
    assert lib().opal_ingest_scan(eng._h, d_text.data_ptr(), len(big), C.byref(nh), C.byref(ns)) == 0
    seqs_big = [sq for _, sq in oracle.fasta_read(big, universal_newlines=True)]
    assert nh.value == len(seqs_big) == ref.n
    assert ns.value == sum(len(sq) for sq in seqs_big) + len(b"junk before the first header")   # dropped text sits in the stream too
    # a text pointer that is not 16-byte aligned (every thread then loads byte by byte), an unaligned output pointer
    shifted = torch.empty(len(big) + 16, dtype=torch.uint8, device=eng.device)
    for sh in (1, 7):
        shifted[sh:sh + len(big)] = d_text
        reads = torch.zeros(len(big) * 2, dtype=torch.uint8, device=eng.device)
        rc = lib().opal_ingest_fasta(eng._h, shifted.data_ptr() + sh, len(big), reads.data_ptr(), len(big) * 2, off.data_ptr(),
                                     ln.data_ptr(), max_rec, C.byref(nrec), C.byref(nb))
        assert rc == 0 and nrec.value == ref.n
        got = amd.PackedReads(reads.cpu().numpy()[:nb.value], off.cpu().numpy()[:ref.n].view(np.uint64),
                              ln.cpu().numpy()[:ref.n].view(np.uint32))
        assert got.sequences() == [sq for _, sq in oracle.fasta_read(big, universal_newlines=True)]
    from opal_amd._lib import OPAL_EINVAL
    assert lib().opal_ingest_fasta(eng._h, d_text.data_ptr(), len(big), reads.data_ptr() + 2, len(big), off.data_ptr(), ln.data_ptr(),
                                   max_rec, C.byref(nrec), C.byref(nb)) == OPAL_EINVAL
    # end to end: predictions from the ingested batch equal predictions from host-parsed records
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    K, bits = 24, 16
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    eng.load_weights(w, K, bits)
    seqs = [sq for _, sq in oracle.fasta_read(big)]
    c1, _ = eng.predict(eng.ingest_fasta(big), True)
    c2, _ = eng.predict(eng.upload(amd.PackedReads.from_sequences(seqs)), True)
    assert np.array_equal(c1.cpu().numpy(), c2.cpu().numpy())


def test_gpu_fasta_ingest_chain_across_tiles(amd, eng, oracle):
    """Texts of 5-40 tiles (16 KiB each) whose structure is hostile to the chained scan: lines, headers, '+' blocks and CR LF
    pairs that straddle tile borders, long stretches without any line start, tiles without a single sequence byte."""
    rng = np.random.default_rng(int(os.environ.get("OPAL_FUZZ_SEED", "321")))
    pieces = [b"ACGT" * 50, b"acgtn" * 30, b"N" * 100, b"\n", b"\r\n", b"\r", b">", b">name with words\n", b"+\n", b"\n+\n",
              b"\n>h\n", b"I" * 3000, b"ACGT" * 5000, b">" + b"x" * 20000 + b"\n", b"+" + b"q" * 17000 + b"\n", b"\n" * 300]
    for trial in range(int(os.environ.get("OPAL_CHAIN_TRIALS", "12"))):
        target = int(rng.integers(5, 40)) * 16384 + int(rng.integers(-20, 20))
        parts, size = [], 0
        while size < target:
            piece = pieces[int(rng.integers(0, len(pieces)))]
            parts.append(piece)
            size += len(piece)
        _check_ingest(amd, eng, oracle, b"".join(parts)[:target])


def test_gpu_fasta_ingest_resizes_when_its_guess_is_too_small(amd, eng, oracle):
    """Engine.ingest_fasta sizes its buffers by guess (one record per >= 96 bytes of text, few raw records) and repeats the
    call with the sizes OPAL_ERANGE reports: many tiny records, and records that are all raw."""
    tiny = b">r\nAC\n" * 200_000                                  # 1.2 MB, 200 000 records: 15 x the guess
    _check_ingest(amd, eng, oracle, tiny)
    raw = b"".join(b">r%d\n" % i + b"acgtn" * 40 + b"\n" for i in range(20_000))   # every record raw: 4 MB behind the stream
    _check_ingest(amd, eng, oracle, raw)
    both = b"".join(b">x\nacgu\n" for _ in range(150_000))          # too many records AND all of them raw
    _check_ingest(amd, eng, oracle, both)


def test_gpu_fasta_ingest_at_scale(amd, eng):
    """A 150 MB reads file (more tiles than one round of the tile scan holds, so the carried state is exercised) with
    1000 raw records sprinkled in: every packed record is its 200-base view of the stream, every raw one decodes."""
    n, L = 700_000, 200
    rng = np.random.default_rng(77)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    rec = np.empty((n, 10 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord(">")
    ids = np.arange(n)
    for d in range(8):
        rec[:, 8 - d] = ord("0") + (ids // 10 ** d) % 10
    rec[:, 9] = ord("\n")
    rec[:, 10:10 + L] = ACGT[codes]
    rec[:, 10 + L] = ord("\n")
    raw_rows = np.sort(rng.choice(n, size=1000, replace=False))
    rec[raw_rows, 10 + rng.integers(0, L, size=1000)] = ord("N")
    assert rec.size > 8192 * 16384
    b = eng.ingest_fasta(rec.reshape(-1))
    assert b.n == n
    ln = b.len.cpu().numpy()[:n].view(np.uint32)
    off = b.off.cpu().numpy()[:n].view(np.uint64)
    is_raw = np.zeros(n, dtype=bool)
    is_raw[raw_rows] = True
    assert np.array_equal(ln, np.where(is_raw, np.uint32(0x80000000 | L), np.uint32(L)))
    want = amd.PackedReads.from_codes(codes)
    assert np.array_equal(off[~is_raw], want.off[~is_raw])          # record i starts at base 200*i of the stream
    data = b.data.cpu().numpy()
    assert np.array_equal(data[:n * 50].reshape(n, 50)[~is_raw], want.data.reshape(n, 50)[~is_raw])
    got = amd.PackedReads(data[:b.nbytes], off[raw_rows], ln[raw_rows]).sequences()
    assert got == [rec[i, 10:10 + L].tobytes() for i in raw_rows]
    assert int(off[raw_rows].min()) >= n * 50 and len(set(off[raw_rows].tolist())) == 1000


def test_full_size_configs_properties(amd, oracle, golden_dir):
    """BASELINE.json configs 2 and 3 at their real sizes (1M x 200bp, K=356 / K=1564 shard, 2^31-slot table):
    size-independent properties + the oracle on a bounded sample."""
    import torch
    from opal_amd.engine import DeviceBatch
    eng = amd.Engine(0)
    dev = eng.device
    rng = np.random.default_rng(31)
    n, L, bits = 1_000_000, 200, 31
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    batch = eng.upload(amd.PackedReads.from_codes(codes))
    # emit at full size: every index equals the multi-threaded oracle's (and so does the checksum)
    blob = ACGT[codes].reshape(-1)
    soff = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    idx, off = eng.features(batch, True, bits)
    oidx, ooff = oracle.features_batch((blob, soff), 64, pos, True, bits, nthreads=0)
    assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff)
    assert np.array_equal(_u32(idx), oidx)
    del idx, oidx
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    w = torch.empty(1 << bits, dtype=torch.float32, device=dev)
    w.normal_(0.0, 0.05, generator=g)
    wh = w.cpu().numpy()

    def view(lo, hi):
        return DeviceBatch(batch.data, batch.off[lo:hi], batch.len[lo:hi], hi - lo, batch.nbytes)

    for K, m in ((356, 1_000_000), (1564, 250_000)):
        eng.load_weights(w, K, bits)
        eng.set_vw_options(clip=False)
        cls, sc = eng.predict(view(0, m), True, want_scores=True)
        cls, sc = cls.clone(), sc.clone()
        # sharding invariance: three uneven shards give the same per-fragment classes and scores
        cuts = [0, m // 7, m // 7 + m // 3, m]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            c, s = eng.predict(view(lo, hi), True, want_scores=True)
            assert torch.equal(c, cls[lo:hi]) and torch.equal(s, sc[lo:hi])
        # the oracle on a bounded sample (first 1500 fragments), bit for bit
        ms = 1500
        ocls, osc, _ = oracle.predict((blob[:ms * L + 1], soff[:ms + 1]), 64, pos, True, wh, K, bits, clip=False, nthreads=0)
        assert np.array_equal(sc[:ms].cpu().numpy(), osc) and np.array_equal(cls[:ms].cpu().numpy(), ocls)
        # exact power-of-two scaling of the whole table
        w.mul_(2.0)
        eng.load_weights(w, K, bits)
        c2, s2 = eng.predict(view(0, m), True, want_scores=True)
        assert torch.equal(s2, sc * 2.0) and torch.equal(c2, cls)
        w.mul_(0.5)
        # VW defaults (clip to [-1,1]): classes still agree with the oracle on the sample
        eng.load_weights(w, K, bits)
        eng.set_vw_options()
        c3, _ = eng.predict(view(0, ms), True)
        o3, _, _ = oracle.predict((blob[:ms * L + 1], soff[:ms + 1]), 64, pos, True, wh, K, bits, want_scores=False, nthreads=0)
        assert np.array_equal(c3.cpu().numpy(), o3)
        del c2, s2, sc, cls
    eng.close()


def test_cfg3_per_gpu_share_12p5M_fragments(amd, oracle, golden_dir):
    """BASELINE.json configs[2] at its real per-GPU size: 12.5 M fragments of 200 bp (100 M over 8 GPUs), K=1564, 2^31-slot
    table scored in place (opal_adopt_weights_device).  Sharding invariance over three uneven shards (what --gpus N does with
    contiguous ranges) and the oracle, bit for bit, on samples from the start, the middle and the end of the batch."""
    import torch
    from opal_amd.engine import DeviceBatch
    eng = amd.Engine(0)
    dev = eng.device
    n, L, bits, K = 12_500_000, 200, 31, 1564
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    eng.set_patterns(pos, 64)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    data = torch.randint(0, 256, (n * 50,), dtype=torch.uint8, device=dev, generator=g)     # 50 random bytes = 200 uniform bases
    off = torch.arange(n, dtype=torch.int64, device=dev) * L
    ln = torch.full((n,), L, dtype=torch.int32, device=dev)
    w = torch.empty((1 << bits) + K + 4, dtype=torch.float32, device=dev)
    w[:1 << bits].normal_(0.0, 0.05, generator=g)
    eng.adopt_weights(w, K, bits)
    eng.set_vw_options()

    def view(lo, hi):
        return DeviceBatch(data, off[lo:hi], ln[lo:hi], hi - lo, n * 50)

    cls, _ = eng.predict(view(0, n), True)
    cls = cls.clone()
    assert int(cls.min()) >= 1 and int(cls.max()) <= K
    cuts = [0, n // 8 + 3, n // 8 + n // 2, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        c, _ = eng.predict(view(lo, hi), True)
        assert torch.equal(c, cls[lo:hi])
    wh = w[:1 << bits].cpu().numpy()
    ms = 400
    for lo in (0, n // 2 - 7, n - ms):
        packed = data[lo * 50:(lo + ms) * 50].cpu().numpy().reshape(ms, 50)
        codes = ((packed[:, :, None] >> (np.arange(4, dtype=np.uint8) * 2)[None, None, :]) & 3).reshape(ms, L)
        seqs = [ACGT[c].tobytes() for c in codes]
        ocls, _, _ = oracle.predict(seqs, 64, pos, True, wh, K, bits, want_scores=False, nthreads=0)
        assert np.array_equal(cls[lo:lo + ms].cpu().numpy(), ocls), lo
    eng.close()


def test_randomised_geometries_against_oracle(amd, eng, oracle):
    """Seeded fuzz over pattern geometry (k, t, P), fragment lengths (around every tile/buffer boundary), packed/raw
    storage, views at every base phase, K, bits and VW switches: emit and predict must equal the oracle every time."""
    # OPAL_FUZZ_SEED / OPAL_FUZZ_TRIALS: longer campaigns with other seeds (run by hand on the GPU box)
    rng = np.random.default_rng(int(os.environ.get("OPAL_FUZZ_SEED", "20240611")))
    alphabets = [ACGT, np.frombuffer(b"ACGTacgtNn", dtype=np.uint8), np.frombuffer(b"0123456789", dtype=np.uint8)]
    for trial in range(int(os.environ.get("OPAL_FUZZ_TRIALS", "70"))):
        t = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 16, 16, 32, 64]))
        mult = int(rng.integers(1, max(2, min(8, 256 // t) + 1)))
        k = t * mult
        P = int(rng.choice([1, 2, 2, 3, 5, 9, 10, 11, 17, 64]))
        pos = np.stack([np.sort(rng.choice(k, size=t, replace=False)) for _ in range(P)]).astype(np.int32)
        if rng.random() < 0.3:                                   # Opal's files start with a run of consecutive positions
            pos[0] = np.arange(t) + int(rng.integers(0, k - t + 1))
        eng.set_patterns(pos, k)
        lens = [int(x) for x in rng.choice([0, 1, k - 1, k, k + 1, k + 63, k + 64, k + 65, 200, 255, 256, 257, 511, 512, 513,
                                            700, 1536 // max(P, 1) + k, 2000], size=12)]
        seqs = []
        for L in lens:
            a = alphabets[0] if rng.random() < 0.7 else alphabets[int(rng.integers(1, 3))]
            seqs.append(bytes(rng.choice(a, size=max(L, 0))))
        reverse = bool(rng.integers(0, 2))
        hash_all = bool(rng.integers(0, 2))
        bits = int(rng.choice([8, 12, 16, 20, 31, 32]))
        eng.set_vw_options(hash_all=hash_all)
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        idx, off = eng.features(batch, reverse, bits)
        oidx, ooff = oracle.features_batch(seqs, k, pos, reverse, bits, hash_all=hash_all)
        assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff), trial
        assert np.array_equal(_u32(idx), oidx), (trial, k, t, P)
    eng.set_vw_options()


def test_run_patterns_and_unsorted_rows_against_oracle(amd, eng, oracle):
    """The t=16 fast paths: pattern 0 as a run of consecutive positions starting anywhere in the window (gather-free
    path), the identity pattern used without a pattern file (k = t = 16, fasta2skm.py:130-132), a run that is NOT
    row 0, and rows whose positions are not ascending (the name keeps the file's order, fasta2skm.py:127-129)."""
    rng = np.random.default_rng(5150)
    cases = []
    for start in (0, 1, 3, 47, 48):
        rows = [np.arange(start, start + 16)] + [np.sort(rng.choice(64, size=16, replace=False)) for _ in range(2)]
        cases.append((64, np.stack(rows)))
    cases.append((16, np.arange(16)[None, :]))                                             # identity, P = 1
    cases.append((64, np.stack([np.sort(rng.choice(64, size=16, replace=False)), np.arange(8, 24)])))   # run in row 1 only
    cases.append((64, np.stack([rng.permutation(64)[:16], np.arange(16)[::-1]])))          # unsorted rows
    cases.append((240, np.stack([np.arange(224, 240), np.sort(rng.choice(240, size=16, replace=False))])))
    for k, pos in cases:
        pos = pos.astype(np.int32)
        eng.set_patterns(pos, k)
        lens = [0, k - 1, k, k + 1, k + 62, k + 63, k + 64, 199, 200, 255, 256, 257, 300, 1000]
        # text fragments (register murmur path), incl. all-digit names that `--hash strings` reads as numbers
        seqs = [bytes(rng.choice(ACGT, size=L)) for L in lens] + [b"ACGTNNNNacgt" * 30, b"acgt" * 70, b"0123456789" * 30,
                                                                  b"00000000000000000000ACGT1234567890123456789" * 8]
        for reverse, hash_all in ((False, False), (True, False), (True, True)):
            eng.set_vw_options(hash_all=hash_all)
            batch = eng.upload(amd.PackedReads.from_sequences(seqs))
            idx, off = eng.features(batch, reverse, 31)
            oidx, ooff = oracle.features_batch(seqs, k, pos, reverse, 31, hash_all=hash_all)
            assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff), (k, pos[0, 0], reverse)
            assert np.array_equal(_u32(idx), oidx), (k, pos[0, 0], reverse, hash_all)
    eng.set_vw_options()


@pytest.mark.parametrize("P", [3, 4, 5, 6, 7, 8, 9, 10])
def test_straight_line_emit_kernels_for_every_pattern_count(amd, eng, oracle, P):
    """emit_canonP_kernel<P> (numHash 2 .. 8: a run followed by P - 1 gathered rows; P = 10 falls to the general fast kernel):
    fragments around every step boundary (windows per strand 1, 63, 64, 65, 127, 128, 129, 449), both strands and one, rows that
    hold consecutive positions inside a block, views at every base phase of a packed genome (the P dwords a lane stores are
    only dword aligned), text fragments in the same batch (they go to the generic kernel)."""
    import torch
    rng = np.random.default_rng(900 + P)
    k = 64
    rows = [np.arange(16) + int(rng.integers(0, 49))]
    for _ in range(P - 1):
        r = np.sort(rng.choice(k, size=16, replace=False))
        if rng.random() < 0.5:                                   # a block of four consecutive positions inside a gathered row
            b = int(rng.integers(0, 4))
            lo = int(min(r[4 * b], k - 4))
            cand = np.concatenate([r[:4 * b], np.arange(lo, lo + 4), r[4 * b + 4:]])
            if len(np.unique(cand)) == 16 and np.all(np.diff(cand) > 0):
                r = cand
        rows.append(r)
    pos = np.stack(rows).astype(np.int32)
    eng.set_patterns(pos, k)
    lens = [0, 63, 64, 126, 127, 128, 190, 191, 192, 193, 200, 255, 256, 257, 300, 511, 512, 513, 600]
    seqs = [bytes(rng.choice(ACGT, size=L)) for L in lens] + [b"ACGTNacgt" * 30, bytes(rng.choice(ACGT, size=200)).lower()]
    for reverse in (True, False):
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        idx, off = eng.features(batch, reverse, 31)
        oidx, ooff = oracle.features_batch(seqs, k, pos, reverse, 31)
        assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff), (P, reverse)
        assert np.array_equal(_u32(idx), oidx), (P, reverse)
    # views into one packed genome at every base phase (offset mod 16), fragments of 200 and 137 bases
    genome = bytes(rng.choice(ACGT, size=5000))
    gs = amd.GenomeSet(eng, amd.PackedReads.from_sequences([genome]))
    starts = np.arange(0, 1600, 37, dtype=np.uint64)
    for L in (200, 137):
        views = gs.fragments(starts, np.zeros(len(starts), dtype=np.uint32), L)
        idx, off = eng.features(views, True, 32)
        want, _ = oracle.features_batch([genome[int(a):int(a) + L] for a in starts], k, pos, True, 32)
        assert np.array_equal(_u32(idx), want), (P, L)
    eng.sync()


def _pair_eligible(row):
    """opal_set_patterns' rule for the paired form: inside every block of 4, letters (0,1) and (2,3) lie 1 .. 14 - 2m apart;
    one pair of the row may fail that (the spare word serves it)."""
    return sum(not (1 <= row[4 * b + 2 * m + 1] - row[4 * b + 2 * m] <= 14 - 2 * m) for b in range(4) for m in range(2)) <= 1


@pytest.mark.parametrize("case", ["edge", "random2", "random5", "random9", "spare_first_pair", "spare_second_pair", "spare_last_block",
                                  "two_far_pairs", "mixed_rows"])
def test_paired_letter_emit_kernels(amd, eng, oracle, case):
    """The emit kernels read a gathered row's letters two to a word (block_addr_pair, opal_set_patterns): rows at the rule's
    limits (gaps of 14 and 12, a run block inside, rotate 0), random rows for numHash 1 / 4 / 8 (one row in eight needs the spare
    word), rows whose one far pair sits on a block's first or second pair or in the last block (spare word, each mask), and rows
    with two far pairs, alone and among paired ones, which must go the one-letter-per-word way.  Against the oracle, and the
    paired form switched off (opal_set_tuning emit_fast = 4) against it switched on."""
    import torch
    rng = np.random.default_rng(sum(map(ord, case)))
    k = 64
    run = np.arange(16) + 3
    edge = np.array([0, 14, 15, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 50, 51, 63])
    two_far = np.array([0, 15, 16, 17, 18, 33, 34, 35] + list(range(36, 44)))

    def random_eligible():
        while True:
            r = np.sort(rng.choice(k, size=16, replace=False))
            if _pair_eligible(r):
                return r

    if case == "edge":
        rows, want_pairs = [run, edge], True
    elif case.startswith("random"):
        rows, want_pairs = [run] + [random_eligible() for _ in range(int(case[6:]) - 1)], True
    elif case == "spare_first_pair":
        rows, want_pairs = [run, np.array([0, 15] + list(range(16, 30)))], True
    elif case == "spare_second_pair":
        rows, want_pairs = [run, np.array([0, 1, 2, 15] + list(range(16, 28)))], True
    elif case == "spare_last_block":
        rows, want_pairs = [run, np.array(list(range(4, 16)) + [44, 45, 46, 63]), np.array(list(range(0, 12)) + [20, 40, 41, 42])], True
    elif case == "two_far_pairs":
        rows, want_pairs = [run, two_far], False
    else:
        rows, want_pairs = [run, random_eligible(), two_far, edge], False
    assert all(_pair_eligible(r) for r in rows[1:]) == want_pairs
    pos = np.stack(rows).astype(np.int32)
    eng.set_patterns(pos, k)
    lens = [0, 63, 64, 65, 127, 128, 129, 191, 192, 193, 200, 256, 300, 511, 512, 513]
    seqs = [bytes(rng.choice(ACGT, size=L)) for L in lens] + [bytes(rng.choice(ACGT, size=int(L))) for L in rng.integers(64, 400, size=40)]
    seqs.append(bytes(rng.choice(ACGT, size=200)).lower())
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    for reverse in (True, False):
        eng.set_tuning()
        idx, off = eng.features(batch, reverse, 31)
        oidx, ooff = oracle.features_batch(seqs, k, pos, reverse, 31)
        assert np.array_equal(off.cpu().numpy().view(np.uint64), ooff), (case, reverse)
        assert np.array_equal(_u32(idx), oidx), (case, reverse)
        eng.set_tuning(emit_fast=4)
        idx4, _ = eng.features(batch, reverse, 31)
        assert torch.equal(idx4, idx), (case, reverse)
    eng.set_tuning()
    eng.sync()


def test_per_base_term_experiment_matches_the_shipped_emit_kernel(amd, eng, golden_dir):
    """opal_set_tuning(emit_fast = 2 | 3): emit_canon2_kernel with per-base term arrays for the run pattern ("Emit,
    round 3": measured, not adopted).  Same indices as the default kernel, for lengths either side of its 256-base variant's
    limit (longer fragments go to the generic kernel there)."""
    import torch
    rng = np.random.default_rng(77)
    eng.set_patterns(_patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64), 64)
    seqs = [bytes(rng.choice(ACGT, size=int(L))) for L in list(rng.integers(0, 600, size=300)) + [63, 64, 65, 255, 256, 257, 511, 512, 513]]
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    for reverse in (True, False):
        eng.set_tuning()
        want, _ = eng.features(batch, reverse, 31)
        for knob in (2, 3):
            eng.set_tuning(emit_fast=knob)
            got, _ = eng.features(batch, reverse, 31)
            assert torch.equal(got, want), (knob, reverse)
    eng.set_tuning()
    eng.sync()


def test_emit_never_writes_past_out_cap(amd, eng, oracle, golden_dir):
    """out_cap of opal_features / opal_feature_names: fragments whose range passes it are skipped, everything
    before is written as usual, nothing lands beyond the capacity."""
    import torch
    from opal_amd._lib import check, lib
    with open(os.path.join(golden_dir, "patterns_k64_t16_m1_s7.txt")) as f:
        pos = amd.parse_patterns(f.readlines(), 64)
    eng.set_patterns(pos, 64)
    rng = np.random.default_rng(11)
    seqs = _rand_seqs(rng, 40, 200)
    b = eng.upload(amd.PackedReads.from_sequences(seqs))
    off = eng.feature_offsets(b, True)
    total = int(off[-1].item())
    oidx, _ = oracle.features_batch(seqs, 64, pos, True, 31)
    cut = int(off[25].item()) + 7                              # capacity ends inside fragment 25
    idx = torch.full((total,), -1, dtype=torch.int32, device=eng.device)
    check(lib().opal_features(eng._h, b.data.data_ptr(), b.nbytes, b.off.data_ptr(), b.len.data_ptr(), b.n, 1, 31,
                              off.data_ptr(), idx.data_ptr(), cut), eng._h)
    got = _u32(idx)
    keep = int(off[25].item())
    assert np.array_equal(got[:keep], oidx[:keep])
    assert np.all(got[keep:] == 0xffffffff)
    names = torch.full((total * 16,), 7, dtype=torch.uint8, device=eng.device)
    check(lib().opal_feature_names(eng._h, b.data.data_ptr(), b.nbytes, b.off.data_ptr(), b.len.data_ptr(), b.n, 1,
                                   off.data_ptr(), names.data_ptr(), cut), eng._h)
    assert bool((names[keep * 16:] == 7).all()) and not bool((names[:keep * 16] == 7).any())


def test_fragments_outside_the_buffer_are_marked_and_reported(amd, eng, oracle, golden_dir):
    """A fragment whose (offset, length) reaches outside d_reads owns a range of the output (offsets come from the lengths
    alone) but has no bases: its features are written as 0xffffffff / NUL names, every other fragment as usual, and the
    next opal_sync reports how many there were (once)."""
    import torch
    from opal_amd._lib import OPAL_EINVAL, check, lib
    with open(os.path.join(golden_dir, "patterns_k64_t16_m1_s7.txt")) as f:
        pos = amd.parse_patterns(f.readlines(), 64)
    eng.set_patterns(pos, 64)
    rng = np.random.default_rng(12)
    seqs = _rand_seqs(rng, 30, 200) + [b"acgtn" * 40]
    b = eng.upload(amd.PackedReads.from_sequences(seqs))
    off_bad = b.off.clone()
    off_bad[7] = b.nbytes * 4 - 100          # packed fragment: 200 bases from 100 bases before the end of the buffer
    off_bad[30] = b.nbytes - 50              # raw fragment: 200 bytes from 50 bytes before the end
    off = eng.feature_offsets(b, True)
    total = int(off[-1].item())
    oidx, ooff = oracle.features_batch(seqs, 64, pos, True, 31)
    idx = torch.full((total,), 5, dtype=torch.int32, device=eng.device)
    check(lib().opal_features(eng._h, b.data.data_ptr(), b.nbytes, off_bad.data_ptr(), b.len.data_ptr(), b.n, 1, 31,
                              off.data_ptr(), idx.data_ptr(), total), eng._h)
    assert lib().opal_sync(eng._h) == OPAL_EINVAL and b"2 fragment(s)" in lib().opal_last_error(eng._h)
    assert lib().opal_sync(eng._h) == 0
    got = _u32(idx)
    for i in range(31):
        a, e = int(ooff[i]), int(ooff[i + 1])
        if i in (7, 30):
            assert np.all(got[a:e] == 0xffffffff)
        else:
            assert np.array_equal(got[a:e], oidx[a:e])
    names = torch.full((total * 16,), 7, dtype=torch.uint8, device=eng.device)
    check(lib().opal_feature_names(eng._h, b.data.data_ptr(), b.nbytes, off_bad.data_ptr(), b.len.data_ptr(), b.n, 1,
                                   off.data_ptr(), names.data_ptr(), total), eng._h)
    assert lib().opal_sync(eng._h) == OPAL_EINVAL
    nm = names.cpu().numpy().reshape(total, 16)
    assert not nm[int(ooff[7]):int(ooff[8])].any() and not nm[int(ooff[30]):int(ooff[31])].any() and nm[:int(ooff[7])].all()


def test_randomised_predict_switches_against_oracle(amd, eng, oracle):
    rng = np.random.default_rng(int(os.environ.get("OPAL_FUZZ_SEED", "77001")))
    for trial in range(int(os.environ.get("OPAL_FUZZ_TRIALS", "40"))):
        t = int(rng.choice([4, 8, 16, 16, 32]))
        k = t * int(rng.integers(1, 5))
        P = int(rng.choice([1, 2, 3, 5, 9, 12]))
        pos = np.stack([np.sort(rng.choice(k, size=t, replace=False)) for _ in range(P)]).astype(np.int32)
        eng.set_patterns(pos, k)
        seqs = [bytes(rng.choice(ACGT if rng.random() < 0.8 else np.frombuffer(b"ACGTNacgt", dtype=np.uint8),
                                 size=int(rng.choice([0, k - 1, k, k + 7, 150, 200, 256, 257, 600, 1700])))) for _ in range(10)]
        reverse, hash_all = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        addc, clip = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        bits = int(rng.choice([8, 12, 16, 20]))
        K = int(rng.choice([1, 2, 3, 4, 10, 63, 64, 65, 200, 356, 513, 700, 2052]))
        w = (rng.standard_normal(1 << bits) * 0.1).astype(np.float32)
        eng.load_weights(w, K, bits)
        eng.set_vw_options(hash_all=hash_all, add_constant=addc, clip=clip, min_label=-0.5, max_label=0.75)
        # half of the trials score views into one packed "genome" holding all sequences back to back
        if trial % 2:
            genome = b"".join(s for s in seqs if all(c in b"ACGT" for c in s))
            starts, frs, o = [], [], 0
            for s in seqs:
                if all(c in b"ACGT" for c in s):
                    starts.append(o)
                    frs.append(s)
                    o += len(s)
            gs = amd.GenomeSet(eng, amd.PackedReads.from_sequences([genome]))
            # equal-length requirement of GenomeSet.fragments: one call per fragment length
            for s0, fr in zip(starts, frs):
                b1 = gs.fragments(np.array([s0]), np.array([0]), len(fr))
                c1, s1 = eng.predict(b1, reverse, want_scores=True)
                oc, osc, _ = oracle.predict([fr], k, pos, reverse, w, K, bits, add_constant=addc, clip=clip,
                                            min_label=-0.5, max_label=0.75, hash_all=hash_all)
                assert np.array_equal(s1.cpu().numpy(), osc) and np.array_equal(c1.cpu().numpy(), oc), (trial, s0, len(fr))
        else:
            batch = eng.upload(amd.PackedReads.from_sequences(seqs))
            cls, sc = eng.predict(batch, reverse, want_scores=True)
            ocls, osc, _ = oracle.predict(seqs, k, pos, reverse, w, K, bits, add_constant=addc, clip=clip,
                                          min_label=-0.5, max_label=0.75, hash_all=hash_all)
            assert np.array_equal(sc.cpu().numpy(), osc), (trial, k, t, P, K, bits)
            assert np.array_equal(cls.cpu().numpy(), ocls), (trial, k, t, P, K, bits)
    eng.set_vw_options()


def test_fasta2skm_tool_and_dump_vw_text_match_reference_lines(amd, eng, golden_dir, tmp_path):
    """The stand-alone fasta2skm tool (util/fasta2skm.py:99-117 mirror) writes the reference's lines, labels included."""
    from opal_amd import fasta2skm
    cases = {c["name"]: c for c in _load(golden_dir, "skm.json")}
    c = cases["edge_m4_r1"]
    fa = tmp_path / "in.fasta"
    fa.write_bytes(c["fasta"].encode("latin-1"))
    out = tmp_path / "out.vw"
    fasta2skm.main(["-i", str(fa), "-k", "64", "-p", os.path.join(golden_dir, c["pattern"]), "-o", str(out), "-r"])
    assert out.read_bytes().decode("latin-1") == "".join(c["lines"])
    d = _load(golden_dir, "dico.json")[0]
    fa.write_text(d["fasta"])
    tx, dico = tmp_path / "in.taxid", tmp_path / "dico.txt"
    tx.write_text(d["taxid"])
    fasta2skm.main(["-i", str(fa), "-k", "64", "-p", os.path.join(golden_dir, "patterns_k64_t16_m1_s7.txt"), "-t", str(tx),
                    "-d", str(dico), "-o", str(out)])
    assert [l.split(" ", 1)[0] for l in out.read_text().splitlines()] == d["labels"]
    assert dico.read_text() == d["dico_out"]
    with pytest.raises(ValueError):
        fasta2skm.main(["-i", str(fa), "-k", "64", "-t", str(tx)])           # --taxid without --dico (fasta2skm.py:57-58)
    with pytest.raises(ValueError):
        fasta2skm.main(["-k", "64"])                                         # no input (fasta2skm.py:52-53)


def test_plain_c_client_of_the_cabi(oracle, golden_dir, tmp_path):
    """examples/predict_c (C, hipMalloc, no Python/torch in the process) gives the oracle's classes."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "predict_c")
    assert os.path.exists(exe), "examples/predict_c is built by __graft_entry__.build()"
    rng = np.random.default_rng(4)
    K, bits = 356, 18
    with open(os.path.join(golden_dir, "patterns_k64_t16_m1_s7.txt")) as f:
        pos = oracle.parse_patterns(f.readlines(), 64)
    seqs = _rand_seqs(rng, 300, 200)
    seqs[7] = seqs[7].lower()
    seqs[9] = seqs[9][:50]
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    w.tofile(tmp_path / "w.f32")
    (tmp_path / "reads.txt").write_bytes(b"".join(s + b"\n" for s in seqs))
    out = tmp_path / "out.txt"
    r = subprocess.run([exe, os.path.join(golden_dir, "patterns_k64_t16_m1_s7.txt"), "64", str(tmp_path / "w.f32"), str(K),
                        str(bits), str(tmp_path / "reads.txt"), str(out), "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.array([int(x) for x in out.read_text().split()], dtype=np.int32)
    ocls, _, _ = oracle.predict(seqs, 64, pos, True, w, K, bits, want_scores=False)
    assert np.array_equal(got, ocls)
