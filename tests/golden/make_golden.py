#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference's own Python.

Run ONLY in the build container (needs /root/reference, read-only):

    python tests/golden/make_golden.py

What is committed is data (inputs + the reference's outputs), never reference source.
The reference modules exercised (file:line in /root/reference):
  util/ldpc.py:10-37,66-97        ldpc(), write_out(), ldpc_write()      -> patterns_*.txt
  util/drawfrag.py:15-49          main_not_commandline() via main(argv)   -> drawfrag_*.json
  util/fasta_functions.py:9-54    fasta_reader, reverse_complement, get_all_substrings
  util/fasta2skm.py:19-44,47-84,120-141  update_dictionary, main_generator, gen_features -> skm_*.json, skm_corpus.json
                                  (skm_corpus.json: digests of the lines and of the hashed indices for ~2100 fragments per
                                  geometry; the fragments come from tests/golden/corpus_gen.py, which the tests share)
  CPython `random` (what drawfrag.py:24,38 and opal.py:363 call)          -> pyrandom.json
  opal.py:52-88,111-138           unique_lines, extract_column_two, vw_class_to_taxid, evaluate_predictions
                                                                          -> opal_helpers.json
  opal.py:329-367                 one training batch as train() assembles it: drawfrag.main (seed 421+i) ->
                                  extract_column_two -> fasta2skm.main_generator(taxid, dico) -> random.shuffle
                                  (the reference's own functions called in train()'s order; train() itself cannot run:
                                  it starts `vw`)                         -> train_stream.json
The murmur indices use sklearn.utils.murmurhash3_32 (an independent MurmurHash3_x86_32) because the
reference's hashing lives in the external `vw` binary, which is absent: the VW side
of parity is UNPINNED.
"""
import argparse
import io
import json
import os
import random
import sys
import tempfile

sys.dont_write_bytecode = True
REF = "/root/reference/util"
sys.path.insert(0, REF)

import warnings  # noqa: E402

import numpy as np  # noqa: E402
warnings.simplefilter("ignore", FutureWarning)        # pandas' note on .agg(np.mean) inside the reference
from sklearn.utils import murmurhash3_32  # noqa: E402

sys.path.insert(1, "/root/reference")
import opal as ref_opal  # noqa: E402  (the reference's opal.py; argparse/pandas/sklearn are present)

import drawfrag  # noqa: E402
import fasta2skm  # noqa: E402
import fasta_functions  # noqa: E402
import ldpc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_dna(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


# ---------------------------------------------------------------- (i) patterns
def gen_patterns():
    out = {}
    for (k, t, m, seed) in [(64, 16, 1, 7), (64, 16, 4, 7), (64, 16, 8, 7), (32, 8, 2, 11), (64, 16, 2, 3), (12, 4, 3, 5)]:
        np.random.seed(seed)
        path = os.path.join(HERE, "patterns_k%d_t%d_m%d_s%d.txt" % (k, t, m, seed))
        ldpc.ldpc_write(k=k, t=t, _m=m, d=path)
        np.random.seed(seed)
        H = ldpc.ldpc(k, t, m)
        out[os.path.basename(path)] = {"k": k, "t": t, "m": m, "np_seed": seed, "H_shape": list(H.shape),
                                      "H_rows": [np.nonzero(r)[0].tolist() for r in H]}
    with open(os.path.join(HERE, "patterns_meta.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


# ---------------------------------------------------------------- (ii) sampler
def run_drawfrag(fasta_text, taxid_text, size, coverage, seed):
    with tempfile.TemporaryDirectory() as d:
        fa, tx = os.path.join(d, "in.fasta"), os.path.join(d, "in.taxid")
        of, og = os.path.join(d, "out.fasta"), os.path.join(d, "out.gi2taxid")
        open(fa, "w").write(fasta_text)
        open(tx, "w").write(taxid_text)
        argv = ["-i", fa, "-t", tx, "-l", str(size), "-c", str(coverage), "-o", of, "-g", og]
        if seed is not None:
            argv += ["-s", str(seed)]
        so = sys.stdout
        sys.stdout = io.StringIO()
        try:
            drawfrag.main(argv)
        finally:
            sys.stdout = so
        # the reference never closes its output files; CPython flushes them when the objects die
        import gc
        gc.collect()
        return open(of).read(), open(og).read()


def gen_drawfrag():
    rng = random.Random(2024)
    cases = []
    # genomes incl. one shorter than the fragment, one exactly the fragment length, multi-line FASTA
    g = [rand_dna(rng, n) for n in (3000, 50, 200, 1777, 199, 4096)]

    def fasta(genomes, width=None):
        s = ""
        for i, x in enumerate(genomes):
            s += ">gi|%d|ref|NC_%04d| synthetic genome %d\n" % (i + 1, i, i)
            if width:
                s += "\n".join(x[j:j + width] for j in range(0, len(x), width)) + "\n"
            else:
                s += x + "\n"
        return s

    taxid = "".join("%d\n" % t for t in (562, 1280, 562, 1313, 1280, 4932))
    for (size, cov, seed, width) in [(200, 1.0, 42, None), (200, 0.1, 421, 70), (200, 15.0, 422, None),
                                     (64, 2.5, 42, 60), (200, 1.0, 423, None), (50, 0.37, 7, None)]:
        ft = fasta(g, width)
        of, og = run_drawfrag(ft, taxid, size, cov, seed)
        cases.append({"fasta": ft, "taxid": taxid, "size": size, "coverage": cov, "seed": seed,
                      "out_fasta": of, "out_gi2taxid": og})
    with open(os.path.join(HERE, "drawfrag.json"), "w") as f:
        json.dump(cases, f)


def gen_pyrandom():
    """Raw CPython `random` streams: what drawfrag.py:24,38 (seed/randint) and opal.py:363 (shuffle) consume."""
    out = []
    for seed in (42, 421, 422, 1, 2**31 - 1, 2**32 + 5, 123456789012345678901234567890):
        random.seed(seed)
        ns = [1, 2, 3, 7, 64, 1000, 4801, 65536, 65537, 3999801, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**40 + 123, 2**63 + 1]
        draws = []
        for rep in range(3):
            for n in ns:
                draws.append([n, random.randint(0, n - 1)])
        lst = list(range(57))
        random.shuffle(lst)
        after = [random.randint(0, 999) for _ in range(5)]
        out.append({"seed": str(seed), "draws": draws, "shuffle57": lst, "after": after})
    with open(os.path.join(HERE, "pyrandom.json"), "w") as f:
        json.dump(out, f)


# ---------------------------------------------------------------- (iii)+(iv) features
def skm_lines(fasta_text, pattern_path, k, reverse, taxid_text=None, dico_text=None):
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "in.fasta")
        open(fa, "w", newline="").write(fasta_text)
        tx = dico = None
        if taxid_text is not None:
            tx = os.path.join(d, "in.taxid")
            open(tx, "w").write(taxid_text)
            dico = os.path.join(d, "dico.txt")
            if dico_text is not None:
                open(dico, "w").write(dico_text)
        ns = argparse.Namespace(input=fa, taxid=tx, kmer=k, dico=dico, output=None, pattern=pattern_path,
                                reverse=reverse)
        lines = list(fasta2skm.main_generator(ns))
        dico_out = open(dico).read() if dico else None
        return lines, dico_out


def hashed(lines):
    """Per line: the feature names after '| ' -> unsigned murmur3_32(name, seed 0); index at b bits = h & (2^b-1)."""
    res = []
    for ln in lines:
        body = ln.rstrip("\n").split("|", 1)[1]
        names = body.split()
        res.append([murmurhash3_32(nm.encode("latin-1"), seed=0, positive=True) for nm in names])
    return res


def gen_skm():
    rng = random.Random(99)
    cases = []
    pat = {m: os.path.join(HERE, "patterns_k64_t16_m%d_s7.txt" % m) for m in (1, 4, 8)}
    # -- A: plain ACGT 200bp fragments, P in {2,5,9}, with and without -r
    frs = [rand_dna(rng, 200) for _ in range(6)]
    fa = "".join(">%d\n%s\n" % (i + 1, s) for i, s in enumerate(frs))
    for m in (1, 4, 8):
        nfr = {1: 6, 4: 3, 8: 2}[m]
        fa = "".join(">%d\n%s\n" % (i + 1, s) for i, s in enumerate(frs[:nfr]))
        for rev in (False, True):
            lines, _ = skm_lines(fa, pat[m], 64, rev)
            cases.append({"name": "acgt200_m%d_r%d" % (m, rev), "fasta": fa, "pattern": os.path.basename(pat[m]),
                          "k": 64, "reverse": rev, "lines": lines, "h32": hashed(lines)})
    # -- B: edge cases: lowercase, N, other letters, len<k, len==k, len==k+1, empty record, multi-line, FASTQ '+', CRLF, digits
    edge = []
    edge.append(rand_dna(rng, 150).lower())                       # all lowercase
    edge.append(rand_dna(rng, 120, "ACGTacgtNn"))                 # mixed case + N
    edge.append(rand_dna(rng, 63))                                # len < k  -> zero features
    edge.append(rand_dna(rng, 64))                                # len == k -> one window
    edge.append(rand_dna(rng, 65))                                # two windows
    edge.append(rand_dna(rng, 100, "ACGTRYKMSWBDHVN"))            # IUPAC
    edge.append(rand_dna(rng, 90, "0123456789"))                  # all digits: VW hashes all-digit names numerically
    edge.append(rand_dna(rng, 70, "ACGT") + "XxUu-*." + rand_dna(rng, 30))
    fb = "".join(">e%d some description\n%s\n" % (i, s) for i, s in enumerate(edge))
    fb += ">empty_mid\n>after_empty\n" + rand_dna(rng, 80) + "\n"
    fb += ">multi\n" + "\n".join(rand_dna(rng, 60) for _ in range(3)) + "\n\n" + rand_dna(rng, 11) + "\n"
    fb += ">fq-like\n" + rand_dna(rng, 96) + "\n+\n" + "I" * 96 + "\n"
    fb += ">last_no_newline\n" + rand_dna(rng, 77)
    for m in (1, 4):
        for rev in (False, True):
            lines, _ = skm_lines(fb, pat[m], 64, rev)
            cases.append({"name": "edge_m%d_r%d" % (m, rev), "fasta": fb, "pattern": os.path.basename(pat[m]),
                          "k": 64, "reverse": rev, "lines": lines, "h32": hashed(lines)})
    # -- C: CRLF file (the reference keeps '\r' inside the sequence)
    fc = ">a\r\n" + rand_dna(rng, 40) + "\r\n" + rand_dna(rng, 40) + "\r\n>b\r\n" + rand_dna(rng, 70) + "\r\n"
    lines, _ = skm_lines(fc, pat[1], 64, True)
    cases.append({"name": "crlf_m1_r1", "fasta": fc, "pattern": os.path.basename(pat[1]), "k": 64, "reverse": True,
                  "lines": lines, "h32": hashed(lines)})
    # -- D: other geometry: k=32,t=8,m=2 and k=12,t=4,m=3 (two-digit-free) ; no pattern file (identity k-mer)
    p32 = os.path.join(HERE, "patterns_k32_t8_m2_s11.txt")
    p12 = os.path.join(HERE, "patterns_k12_t4_m3_s5.txt")
    fd = "".join(">%d\n%s\n" % (i, rand_dna(rng, n)) for i, n in enumerate((31, 32, 33, 100, 257)))
    for (pp, k) in ((p32, 32), (p12, 12)):
        lines, _ = skm_lines(fd, pp, k, True)
        cases.append({"name": "geom_k%d" % k, "fasta": fd, "pattern": os.path.basename(pp), "k": k, "reverse": True,
                      "lines": lines, "h32": hashed(lines)})
    lines, _ = skm_lines(fd, None, 20, True)
    cases.append({"name": "nopattern_k20", "fasta": fd, "pattern": None, "k": 20, "reverse": True,
                  "lines": lines, "h32": hashed(lines)})
    # -- E: >=10 patterns (two-digit pattern suffix): hand-made pattern file with 11 rows, t=4, k=16
    p11 = os.path.join(HERE, "patterns_hand_k16_t4_p11.txt")
    prng = random.Random(5)
    with open(p11, "w") as f:
        f.write("11 4\n")
        for i in range(11):
            f.write(" ".join(str(x) for x in sorted(prng.sample(range(16), 4))) + " \n")
    lines, _ = skm_lines(fd, p11, 16, True)
    cases.append({"name": "p11_k16", "fasta": fd, "pattern": os.path.basename(p11), "k": 16, "reverse": True,
                  "lines": lines, "h32": hashed(lines)})
    with open(os.path.join(HERE, "skm.json"), "w") as f:
        json.dump(cases, f)


# ---------------------------------------------------------------- (iii b) hash-only corpus: thousands of fragments, digests only
def gen_skm_corpus():
    """~2100 deterministic fragments (tests/golden/corpus_gen.py: lengths 0..600, alphabets ACGT / ACGTacgtNn / IUPAC) per
    geometry P in {2, 5, 9} and -r on/off through the reference's main_generator (util/fasta2skm.py:47-84); stored per case:
    SHA-256 of the concatenated VW text lines, SHA-256 of the little-endian u32 stream of murmurhash3_32(name) & (2^b - 1)
    for b = 32 and 31 (sklearn's MurmurHash3: independent of the build), SHA-256 of the per-line feature counts."""
    import hashlib
    import corpus_gen
    per_alphabet = 700
    cases = []
    for m in (1, 4, 8):
        pat = os.path.join(HERE, "patterns_k64_t16_m%d_s7.txt" % m)
        for rev in (False, True):
            name = "corpus_m%d_r%d" % (m, rev)
            seqs = corpus_gen.fragments(name, per_alphabet)
            lines, _ = skm_lines(corpus_gen.fasta_text(seqs), pat, 64, rev)
            assert len(lines) == len(seqs)
            text = hashlib.sha256("".join(lines).encode("latin-1"))
            counts, hs = [], []
            for ln in lines:
                names = ln.rstrip("\n").split("|", 1)[1].split()
                counts.append(len(names))
                hs.extend(murmurhash3_32(nm.encode("latin-1"), seed=0, positive=True) for nm in names)
            h = np.array(hs, dtype=np.uint32)
            cases.append({"name": name, "pattern": os.path.basename(pat), "k": 64, "reverse": rev, "per_alphabet": per_alphabet,
                          "fragments": len(seqs), "features": int(h.size), "text_bytes": sum(len(ln) for ln in lines),
                          "text_sha256": text.hexdigest(),
                          "counts_sha256": hashlib.sha256(np.array(counts, dtype="<u8").tobytes()).hexdigest(),
                          "idx32_sha256": hashlib.sha256(h.astype("<u4").tobytes()).hexdigest(),
                          "idx31_sha256": hashlib.sha256((h & np.uint32(0x7FFFFFFF)).astype("<u4").tobytes()).hexdigest()})
            print(name, len(seqs), "fragments", h.size, "features", flush=True)
    with open(os.path.join(HERE, "skm_corpus.json"), "w") as f:
        json.dump(cases, f, indent=0)


# ---------------------------------------------------------------- (v) dico evolution
def gen_dico():
    rng = random.Random(3)
    frs = [rand_dna(rng, 70) for _ in range(7)]
    fa = "".join(">%d\n%s\n" % (i + 1, s) for i, s in enumerate(frs))
    pat = os.path.join(HERE, "patterns_k64_t16_m1_s7.txt")
    cases = []
    tax1 = "562\n1280\n562\nabc\n1313\n1280\nX\n"
    lines, d1 = skm_lines(fa, pat, 64, False, taxid_text=tax1)
    cases.append({"fasta": fa, "taxid": tax1, "dico_in": None, "dico_out": d1, "labels": [l.split(" ", 1)[0] for l in lines]})
    tax2 = "9\n562\nabc\n9\n77\n1313\n5\n"
    lines, d2 = skm_lines(fa, pat, 64, False, taxid_text=tax2, dico_text=d1)
    cases.append({"fasta": fa, "taxid": tax2, "dico_in": d1, "dico_out": d2, "labels": [l.split(" ", 1)[0] for l in lines]})
    # pre-existing dico with gaps / non-sequential ids
    d0 = "foo\t5\nbar\t2\n"
    lines, d3 = skm_lines(fa, pat, 64, False, taxid_text=tax1, dico_text=d0)
    cases.append({"fasta": fa, "taxid": tax1, "dico_in": d0, "dico_out": d3, "labels": [l.split(" ", 1)[0] for l in lines]})
    with open(os.path.join(HERE, "dico.json"), "w") as f:
        json.dump(cases, f)


# ---------------------------------------------------------------- string primitives
def gen_primitives():
    rng = random.Random(17)
    rc = []
    for s in ["", "A", "acgtNnX", "ACGTacgt", rand_dna(rng, 50, "ACGTacgtNnRYxX-*0189 \r")]:
        rc.append([s, fasta_functions.reverse_complement(s)])
    readers = []
    for txt in ["@r1\nACGT\n+\nIIII\n@r2\nGGCC\n+\nIIII\n", "AC\n>a\nAC\n\nGT\n", ">a\r\nAC\r\nGT\r\n", ">x\n>y\nAC\n>z\n",
                ">only\nACGT", "", ">a\nAC\n+\nII\nGG\n>b\nTT\n", ">  spaced name \nA C\n"]:
        readers.append([txt, [list(x) for x in fasta_functions.fasta_reader(io.StringIO(txt, newline=""))]])
    subs = [[s, k, fasta_functions.get_all_substrings(s, k)] for (s, k) in [("ACGTAC", 4), ("ACG", 4), ("ACGT", 4), ("", 1)]]
    with open(os.path.join(HERE, "primitives.json"), "w") as f:
        json.dump({"reverse_complement": rc, "fasta_reader": readers, "get_all_substrings": subs}, f)


# ---------------------------------------------------------------- opal.py host helpers (opal.py:52-88,111-138)
def _captured(fn, *a):
    so = sys.stdout
    sys.stdout = io.StringIO()
    try:
        fn(*a)
        return sys.stdout.getvalue()
    finally:
        sys.stdout = so


def gen_opal_helpers():
    out = {"unique_lines": [], "extract_column_two": [], "vw_class_to_taxid": [], "evaluate_predictions": []}
    with tempfile.TemporaryDirectory() as d:
        a, b, c = os.path.join(d, "a"), os.path.join(d, "b"), os.path.join(d, "c")
        for txt in ["562\n1280\n562\n", "562\n1280\n562", "", "\n\n", "a\nA\n a\na \n", "x\r\nx\n"]:
            open(a, "w", newline="").write(txt)
            out["unique_lines"].append([txt, ref_opal.unique_lines(a)])
        for txt in ["gi|1|\t562\ngi|2|\t1280\n", "one\n\nthree words here\n a  b\tc\n", "x\ty", "", "n\t12 34\n"]:
            open(a, "w").write(txt)
            ref_opal.extract_column_two(a, b)
            out["extract_column_two"].append([txt, open(b).read()])
        dico = "562\t1\n1280\t2\nNC_000913.3\t3\nlabel with\t4\n9\t7\n"
        open(b, "w").write(dico)
        for txt in ["3\n1\n1\n2\n", "3.000000\n1.0\n7\n", " 2 \n", ""]:
            open(a, "w").write(txt)
            ref_opal.vw_class_to_taxid(a, b, c)
            out["vw_class_to_taxid"].append({"preds": txt, "dico": dico, "taxids": open(c).read(), "error": None})
        for txt, err in [("5\n", "KeyError"), ("1\nx\n", "ValueError")]:
            open(a, "w").write(txt)
            try:
                ref_opal.vw_class_to_taxid(a, b, c)
                raise SystemExit("expected " + err)
            except (KeyError, ValueError) as e:
                assert type(e).__name__ == err
            out["vw_class_to_taxid"].append({"preds": txt, "dico": dico, "taxids": None, "error": err})
        rng = random.Random(8)
        ref_big = [str(rng.choice([562, 1280, 1313, 4932, 9, 77])) for _ in range(500)]
        pred_big = [x if rng.random() < 0.8 else str(rng.choice([562, 1280, 5])) for x in ref_big]
        cases = [
            (["1", "2", "2", "3"], ["1", "2", "1", "3"]),
            (["a", "a", "b", "b", "b", "c"], ["a", "b", "b", "b", "a", "x"]),             # class of pred absent from ref, and vice versa
            (["1", "1", "2", "2", "2"], ["1", "1", "2"]),                                  # pred shorter than ref
            (["1", "2"], ["1", "2", "2", "2"]),                                            # pred longer than ref
            (["7"], ["7"]),
            (["1", "2", "3", "4"], ["4", "3", "2", "1"]),
            (ref_big, pred_big),
        ]
        for ref, pred in cases:
            open(a, "w").write("".join(x + "\n" for x in ref))
            open(b, "w").write("".join(x + "\n" for x in pred))
            try:                                   # files of unequal length: pandas refuses the frame (opal.py:123) -> ValueError
                rec = {"ref": ref, "pred": pred, "stdout": _captured(ref_opal.evaluate_predictions, a, b), "error": None}
            except ValueError:
                assert len(ref) != len(pred)
                rec = {"ref": ref, "pred": pred, "stdout": None, "error": "ValueError"}
            out["evaluate_predictions"].append(rec)
    with open(os.path.join(HERE, "opal_helpers.json"), "w") as f:
        json.dump(out, f)


# ---------------------------------------------------------------- the training stream (opal.py:329-367)
def gen_train_stream():
    import hashlib
    rng = random.Random(77)
    lens = (900, 150, 2300, 99, 1200, 640, 100)
    genomes = [rand_dna(rng, n, "ACGT" if i != 4 else "ACGTacgtN") for i, n in enumerate(lens)]
    fasta = ""
    for i, g in enumerate(genomes):
        fasta += ">gi|%d|ref|NC_%04d| genome %d\n" % (i + 1, i, i)
        fasta += "\n".join(g[j:j + 80] for j in range(0, len(g), 80)) + "\n"
    # two genomes share a taxid, one taxid has a second column (extract_column_two keeps its first word only)
    taxid_full = "562\n1280\n562\n77\n1313 strainB\n4932\n1280\n"
    # the last genome has NO taxid line: its label is '' (a second batch would die re-reading the dico line "\t<id>",
    # fasta2skm.py:26, so this case has one batch)
    taxid_short = "562\n1280\n562\n77\n1313 strainB\n4932\n"
    pat = os.path.join(HERE, "patterns_k64_t16_m1_s7.txt")
    cases = []
    for (L, cov, nb, reverse, taxid_text) in [(100, 1.0, 2, True, taxid_full), (64, 2.5, 3, False, taxid_full),
                                              (100, 0.7, 1, True, taxid_short)]:
        with tempfile.TemporaryDirectory() as d:
            fa, tx, dico = os.path.join(d, "train.fasta"), os.path.join(d, "train.taxid"), os.path.join(d, "vw-dico.txt")
            open(fa, "w").write(fasta)
            open(tx, "w").write(taxid_text)
            seed, batches = 420, []
            for i in range(nb):                                   # opal.py:328-367, with the reference's own functions
                seed = seed + 1
                prefix = os.path.join(d, "train.batch-%d" % i)
                _captured(drawfrag.main, ["-i", fa, "-t", tx, "-l", str(L), "-c", str(cov), "-o", prefix + ".fasta",
                                          "-g", prefix + ".gi2taxid", "-s", str(seed)])
                import gc
                gc.collect()                                      # drawfrag never closes its files
                ref_opal.extract_column_two(prefix + ".gi2taxid", prefix + ".taxid")
                ns = argparse.Namespace(input=prefix + ".fasta", taxid=prefix + ".taxid", kmer=64, dico=dico, output=None,
                                        pattern=pat, reverse=reverse)
                training_list = [line.rstrip("\n") for line in fasta2skm.main_generator(ns)]
                frags = [s for _, s in fasta_functions.fasta_reader(open(prefix + ".fasta"))]
                assert len(frags) == len(training_list)
                state = random.getstate()
                random.shuffle(training_list)                     # opal.py:363: the stream drawfrag seeded goes on
                after = random.getstate()
                random.setstate(state)
                order = list(range(len(frags)))
                random.shuffle(order)                             # same length -> same permutation
                assert random.getstate() == after
                labels = [ln.split(" ", 1)[0] for ln in training_list]
                text = "".join("%s\n" % item for item in training_list).encode()      # what opal.py:366-367 pipes to vw
                batches.append({"seed": seed, "labels": labels, "fragments": [frags[j] for j in order],
                                "dico_after": open(dico).read(), "vw_text_sha256": hashlib.sha256(text).hexdigest(),
                                "vw_text_bytes": len(text)})
            cases.append({"fasta": fasta, "taxid": taxid_text, "frag_length": L, "coverage": cov, "reverse": reverse,
                          "pattern": os.path.basename(pat), "kmer": 64, "num_labels": ref_opal.unique_lines(tx),
                          "batches": batches})
    with open(os.path.join(HERE, "train_stream.json"), "w") as f:
        json.dump(cases, f)


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    if "--only-corpus" in sys.argv:
        gen_skm_corpus()
        sys.exit(0)
    gen_opal_helpers()
    gen_train_stream()
    if "--only-new" in sys.argv:
        sys.exit(0)
    gen_skm_corpus()
    gen_patterns()
    gen_drawfrag()
    gen_pyrandom()
    gen_skm()
    gen_dico()
    gen_primitives()
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("golden written to", HERE, "total bytes", total)
