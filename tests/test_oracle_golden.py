"""Pins the CPU oracle (oracle/) to vectors produced by the reference's own Python (tests/golden/make_golden.py).

Covers SURVEY.md §8c (i)-(vi).  Everything here runs on CPU.
"""
import json
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _patterns(oracle, golden_dir, fname, k):
    if fname is None:
        return oracle.parse_patterns(None, k)
    with open(os.path.join(golden_dir, fname)) as f:
        return oracle.parse_patterns(f.readlines(), k)


# ---------------------------------------------------------------- primitives
def test_reverse_complement(oracle, golden_dir):
    for s, want in _load(golden_dir, "primitives.json")["reverse_complement"]:
        assert oracle.reverse_complement(s.encode("latin-1")).decode("latin-1") == want


def test_fasta_reader_raw_newlines(oracle, golden_dir):
    # generated with io.StringIO(newline='') -> only '\n' ends a line
    for txt, want in _load(golden_dir, "primitives.json")["fasta_reader"]:
        got = oracle.fasta_read(txt.encode("latin-1"), universal_newlines=False)
        assert [[a.decode("latin-1"), b.decode("latin-1")] for a, b in got] == want, repr(txt)


def test_murmur_known_answer(oracle):
    # SURVEY.md §8a row 8 [probed with sklearn]
    assert oracle.murmur3_32(b"TGCAACTCATCGACTC0") == 1729087884
    assert oracle.murmur3_32(b"") == 0
    assert oracle.vw_hashstring(b"12345") == 12345          # VW --hash strings digit rule [VW-unpinned]
    assert oracle.vw_hashstring(b"1234a") == oracle.murmur3_32(b"1234a")


# ---------------------------------------------------------------- (iii)+(iv) VW lines + hashes
def test_vw_lines_and_hashes(oracle, golden_dir):
    cases = _load(golden_dir, "skm.json")
    assert len(cases) >= 15
    nlines = 0
    for c in cases:
        pos = _patterns(oracle, golden_dir, c["pattern"], c["k"])
        recs = oracle.fasta_read(c["fasta"].encode("latin-1"), universal_newlines=True)
        assert len(recs) == len(c["lines"]), c["name"]
        for (name, seq), want_line, want_h in zip(recs, c["lines"], c["h32"]):
            got = oracle.vw_line(seq, c["k"], pos, c["reverse"], "1")
            assert got.decode("latin-1") == want_line, c["name"]
            h = oracle.feature_hashes(seq, c["k"], pos, c["reverse"], hash_all=True)
            assert h.tolist() == want_h, c["name"]
            nlines += 1
    assert nlines > 60


def test_batch_indices_match_per_fragment(oracle, golden_dir):
    c = [x for x in _load(golden_dir, "skm.json") if x["name"] == "edge_m4_r1"][0]
    pos = _patterns(oracle, golden_dir, c["pattern"], c["k"])
    seqs = [s for _, s in oracle.fasta_read(c["fasta"].encode("latin-1"))]
    for bits in (18, 24, 31):
        idx, off = oracle.features_batch(seqs, c["k"], pos, True, bits, hash_all=True, nthreads=2)
        want = [np.array(h, dtype=np.uint64) & ((1 << bits) - 1) for h in c["h32"]]
        assert off.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in want])]).tolist()
        assert idx.tolist() == np.concatenate(want).astype(np.uint32).tolist()


# ---------------------------------------------------------------- (ii)+(vi) sampler, shuffle
def test_cpython_random_streams(oracle, golden_dir):
    for c in _load(golden_dir, "pyrandom.json"):
        r = oracle.PyRandom(int(c["seed"]))
        for n, want in c["draws"]:
            assert r.randint(0, n - 1) == want, (c["seed"], n)
        assert r.shuffle_perm(57).tolist() == c["shuffle57"]
        assert [r.randint(0, 999) for _ in range(5)] == c["after"]


def test_drawfrag_files(oracle, golden_dir):
    for c in _load(golden_dir, "drawfrag.json"):
        of, og = oracle.drawfrag_files(c["fasta"].encode("latin-1"), c["taxid"], c["size"], c["coverage"], c["seed"])
        assert of == c["out_fasta"], (c["size"], c["coverage"], c["seed"])
        assert og == c["out_gi2taxid"]
        # exactly ceil(c*len/l) fragments per genome that is long enough
        n_expected = 0
        for name, seq in oracle.fasta_read(c["fasta"].encode("latin-1")):
            if len(seq) >= c["size"]:
                n = 0
                while n * c["size"] < c["coverage"] * len(seq):
                    n += 1
                n_expected += n
        assert of.count(">") == n_expected


# ---------------------------------------------------------------- (i) patterns
def test_ldpc_construction_and_file_format(oracle, golden_dir):
    meta = _load(golden_dir, "patterns_meta.json")
    for fname, m in meta.items():
        k, t, _m = m["k"], m["t"], m["m"]
        rows = m["H_rows"]
        # recover the permutations the reference drew from its H: block b row i = {perm^-1 positions of block i}
        kt = k // t
        nblocks = len(rows) // kt
        perms = []
        for b in range(1, nblocks):
            perm = np.zeros(k, dtype=np.int64)
            for i in range(kt):
                cols = rows[b * kt + i]
                assert len(cols) == t
                # H_basic[:, perm][i, j] = H_basic[i, perm[j]] -> perm[j] in [i*t,(i+1)*t) for j in cols
                perm[cols] = np.arange(i * t, (i + 1) * t)
            perms.append(perm)
            # partition property: the kt rows of a permuted block cover every position exactly once
            assert sorted(sum((rows[b * kt + i] for i in range(kt)), [])) == list(range(k))
        H = oracle.ldpc_matrix(k, t, _m, perms)
        assert list(H.shape) == m["H_shape"]
        assert [np.nonzero(r)[0].tolist() for r in H] == rows
        with open(os.path.join(golden_dir, fname)) as f:
            assert oracle.patterns_text(H, _m) == f.read()


# ---------------------------------------------------------------- (v) dico
def test_dico_evolution(oracle, golden_dir):
    for c in _load(golden_dir, "dico.json"):
        d, text = oracle.update_dictionary(c["taxid"].splitlines(True), c["dico_in"])
        assert text == c["dico_out"]
        assert [str(d[l.rstrip("\n")]) for l in c["taxid"].splitlines(True)] == c["labels"]


# ---------------------------------------------------------------- scoring: self-consistency only (VW unpinned)
def test_evaluate_matches_reference_output(oracle, golden_dir):
    """oracle.evaluate against the three numbers the reference's evaluate_predictions printed (opal.py:111-131)."""
    for case in _load(golden_dir, "opal_helpers.json")["evaluate_predictions"]:
        if case["error"]:
            with pytest.raises(ValueError):
                oracle.evaluate(case["ref"], case["pred"])
            continue
        micro, macro, median = oracle.evaluate(case["ref"], case["pred"])
        assert "micro = {:.4f}\nmacro = {:.4f}\nmedian = {:.4f}\n".format(micro, macro, median) == case["stdout"]


def test_scoring_matches_numpy_restatement(oracle, golden_dir):
    """orc_predict vs an independent numpy statement of the same documented VW rules.  NOT reference parity:
    no vw binary exists here -> 'parity unpinned'."""
    c = [x for x in _load(golden_dir, "skm.json") if x["name"] == "acgt200_m1_r1"][0]
    pos = _patterns(oracle, golden_dir, c["pattern"], c["k"])
    seqs = [s for _, s in oracle.fasta_read(c["fasta"].encode("latin-1"))]
    rng = np.random.default_rng(5)
    for (K, bits) in [(10, 12), (7, 16), (356, 18)]:
        w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
        cls, sc, tot = oracle.predict(seqs, c["k"], pos, True, w, K, bits, clip=False)
        mask = (1 << bits) - 1
        for i, h in enumerate(c["h32"]):
            hs = [x & mask for x in h] + [11650396 & mask]
            acc = np.zeros(K, dtype=np.float32)
            for x in hs:
                slots = (x * K + np.arange(K)) & mask
                acc = (acc + w[slots]).astype(np.float32)
            assert np.array_equal(acc, sc[i])
            assert cls[i] == int(np.argmax(acc)) + 1
        cls2, sc2, _ = oracle.predict(seqs, c["k"], pos, True, w, K, bits, clip=True, min_label=-0.3, max_label=0.2)
        assert np.array_equal(sc2, np.clip(sc, -0.3, 0.2))
        assert cls2.tolist() == (np.argmax(sc2, axis=1) + 1).tolist()


# ---------------------------------------------------------------- hash-only corpus (digests of the reference's output)
def test_corpus_digests(oracle, golden_dir):
    """~2100 fragments per geometry (lengths 0..600, ACGT / mixed case + N / IUPAC; tests/golden/corpus_gen.py) went through the
    reference's main_generator; skm_corpus.json keeps SHA-256 digests of its lines and of the sklearn-hashed indices."""
    import hashlib
    import sys
    sys.path.insert(0, golden_dir)
    import corpus_gen
    for c in _load(golden_dir, "skm_corpus.json"):
        pos = _patterns(oracle, golden_dir, c["pattern"], c["k"])
        seqs = corpus_gen.fragments(c["name"], c["per_alphabet"])
        assert len(seqs) == c["fragments"]
        idx, off = oracle.features_batch(seqs, c["k"], pos, c["reverse"], 32, hash_all=True, nthreads=4)
        assert idx.size == c["features"], c["name"]
        assert hashlib.sha256(np.diff(off).astype("<u8").tobytes()).hexdigest() == c["counts_sha256"], c["name"]
        assert hashlib.sha256(idx.astype("<u4").tobytes()).hexdigest() == c["idx32_sha256"], c["name"]
        assert hashlib.sha256((idx & np.uint32(0x7FFFFFFF)).astype("<u4").tobytes()).hexdigest() == c["idx31_sha256"], c["name"]
        if c["features"] < 3_000_000:        # the text of the larger cases (up to 160 MB) is checked on the GPU side only
            text = hashlib.sha256()
            for s in seqs:
                text.update(oracle.vw_line(s, c["k"], pos, c["reverse"], "1"))
            assert text.hexdigest() == c["text_sha256"], c["name"]
