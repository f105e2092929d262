"""CPU-only tests of the product's host side (no GPU, no kernels): file formats, sampler, C-ABI surface."""
import argparse
import io
import json
import os
import re

import numpy as np
import pytest


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_cabi_exports_every_declared_symbol():
    """include/opal_hip.h <-> libopal_hip.so <-> ctypes prototypes agree (no compute calls)."""
    import opal_amd
    from opal_amd import _lib
    L = opal_amd.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "opal_hip.h")).read()
    declared = set(re.findall(r"\b(opal_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libopal_hip.so lacks " + name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert b"gfx950" in L.opal_version()
    # the library beside the sources was built from exactly these sources (a stale or experimental build would otherwise
    # travel to the GPU box unnoticed)
    import __graft_entry__ as entry
    assert L.opal_version().decode().endswith("src " + entry.source_hash()), "opal_amd/libopal_hip.so is stale: run __graft_entry__.build()"


def test_product_fasta_reader_and_rc_match_reference(golden_dir):
    from opal_amd.fasta_functions import fasta_reader, reverse_complement, get_all_substrings
    g = _load(golden_dir, "primitives.json")
    for txt, want in g["fasta_reader"]:
        assert [list(x) for x in fasta_reader(io.StringIO(txt, newline=""))] == want
    for s, want in g["reverse_complement"]:
        assert reverse_complement(s) == want
    for s, k, want in g["get_all_substrings"]:
        assert get_all_substrings(s, k) == want


def test_rng_matches_cpython_golden(golden_dir):
    import opal_amd
    for c in _load(golden_dir, "pyrandom.json"):
        r = opal_amd.Rng(int(c["seed"]))
        for n, want in c["draws"]:
            assert r.randint(0, n - 1) == want
        assert r.shuffle_perm(57).tolist() == c["shuffle57"]
        assert [r.randint(0, 999) for _ in range(5)] == c["after"]


def test_block_sampler_equals_per_call_stream():
    """opal_sample_positions consumes whole tempered blocks of the generator; it must draw exactly what a loop of
    random.randint calls draws (drawfrag.py:36-49), for tiny ranges with many rejections, ranges just above and below a
    power of two, genomes shorter than the fragment, fractional coverage, and leave the stream where that loop would."""
    import opal_amd
    cases = [
        ([200, 201, 202, 203, 1000, 199, 5000], 200, 3.0),
        ([1 << 12, (1 << 12) + 1, (1 << 12) - 1, (1 << 20) + 7], 1, 0.02),
        ([300_000, 10, 70_001, 65_536 + 149], 150, 1.7),
        ([(1 << 32) + 10, 1 << 31], 11, 1e-6),
        ([5000] * 40, 64, 15.0),
        ([777], 200, 0.0),
    ]
    for seed, (lens, fl, cov) in enumerate(cases, start=421):
        fast, slow = opal_amd.Rng(seed), opal_amd.Rng(seed)
        pos, gen = fast.sample_positions(lens, fl, cov)
        want_pos, want_gen = [], []
        for g, L in enumerate(lens):
            if L < fl:
                continue
            covered, tries = 0, 0
            while covered < cov * L:
                tries += 1
                want_pos.append(slow.randint(0, L - fl))
                want_gen.append(g)
                covered += fl
                if tries > 10 * L:
                    break
        assert pos.tolist() == want_pos and gen.tolist() == want_gen
        assert [fast.randint(0, 10 ** 6) for _ in range(700)] == [slow.randint(0, 10 ** 6) for _ in range(700)]
        n = len(want_pos)
        assert fast.shuffle_perm(n + 130).tolist() == _py_shuffle(slow, n + 130)


def _py_shuffle(rng, n):
    x = list(range(n))
    for i in reversed(range(1, n)):                       # Lib/random.py shuffle
        j = rng.randint(0, i)
        x[i], x[j] = x[j], x[i]
    return x


def test_drawfrag_files_match_reference(golden_dir, tmp_path):
    from opal_amd import drawfrag
    for i, c in enumerate(_load(golden_dir, "drawfrag.json")):
        fa, tx = tmp_path / ("in%d.fasta" % i), tmp_path / ("in%d.taxid" % i)
        of, og = tmp_path / ("o%d.fasta" % i), tmp_path / ("o%d.g2t" % i)
        fa.write_text(c["fasta"])
        tx.write_text(c["taxid"])
        drawfrag.main(["-i", str(fa), "-t", str(tx), "-l", str(c["size"]), "-c", str(c["coverage"]), "-o", str(of),
                       "-g", str(og), "-s", str(c["seed"])])
        assert of.read_text() == c["out_fasta"]
        assert og.read_text() == c["out_gi2taxid"]


def test_vectorised_fragment_writer_equals_per_fragment_loop(tmp_path):
    """write_fragment_files (numpy gather, fixed-width header blocks) writes the bytes the reference's per-fragment
    loop writes (drawfrag.py:44-47): across the 9->10, 99->100, 999->1000 header widths, with records shorter than the
    fragment, lower case / N / latin-1 bytes in the text, names with spaces, a missing taxid line."""
    import opal_amd
    from opal_amd import drawfrag
    rng = np.random.default_rng(5)
    recs = []
    for i, L in enumerate([700, 30, 2500, 64, 65, 9000, 1200]):
        seq = "".join(rng.choice(list("ACGTacgtN\xe9"), size=L))
        recs.append(("rec%d some description" % i, seq))
    taxids = ["t%d" % i for i in range(len(recs) - 1)]                 # the last record has no taxid line
    for size, cov, seed in ((64, 9.0, 42), (65, 0.5, 421), (1, 0.2, 7)):
        frags, labels = drawfrag.sample_fragments(recs, taxids, size, cov, seed)
        want_f = "".join(">%d\n%s\n" % (i + 1, s) for i, s in enumerate(frags)).encode("latin-1")
        want_g = "".join("%s\t%s\n" % nl for nl in labels).encode("latin-1")
        n = drawfrag.write_fragment_files(recs, taxids, size, cov, seed, str(tmp_path / "f.fasta"), str(tmp_path / "g.txt"), chunk=100)
        assert n == len(frags) and (size != 64 or n > 1000)
        assert (tmp_path / "f.fasta").read_bytes() == want_f
        assert (tmp_path / "g.txt").read_bytes() == want_g


def test_ldpc_files_match_reference(golden_dir, tmp_path):
    from opal_amd import ldpc
    for fname, m in _load(golden_dir, "patterns_meta.json").items():
        np.random.seed(m["np_seed"])
        out = tmp_path / fname
        ldpc.ldpc_write(k=m["k"], t=m["t"], _m=m["m"], d=str(out))
        assert out.read_text() == open(os.path.join(golden_dir, fname)).read(), fname
        np.random.seed(m["np_seed"])
        H = ldpc.ldpc(m["k"], m["t"], m["m"])
        assert [np.flatnonzero(r).tolist() for r in H] == m["H_rows"]
    with pytest.raises(ValueError):
        ldpc.ldpc(64, 10, 1)


def test_dico_matches_reference(golden_dir, tmp_path):
    from opal_amd import fasta2skm
    for i, c in enumerate(_load(golden_dir, "dico.json")):
        d = tmp_path / ("dico%d.txt" % i)
        if c["dico_in"] is not None:
            d.write_text(c["dico_in"])
        m = fasta2skm.update_dictionary(io.StringIO(c["taxid"]), str(d))
        assert d.read_text() == c["dico_out"]
        assert [str(m[l.rstrip("\n")]) for l in c["taxid"].splitlines(True)] == c["labels"]


def test_pack_reads_layout():
    import opal_amd
    pr = opal_amd.PackedReads.from_sequences([b"ACGTACGTA", b"acgt", b"", b"TTTT", b"ACGN"])
    assert pr.len.tolist() == [9, 4 | 0x80000000, 0, 4, 4 | 0x80000000]
    assert pr.off.tolist() == [0, 3, 7 * 4, 7 * 4, 8]            # packed: base offsets, raw: byte offsets
    assert pr.data[:3].tolist() == [0b11100100, 0b11100100, 0]
    assert bytes(pr.data[3:7]) == b"acgt" and pr.data[7] == 0xFF and bytes(pr.data[8:12]) == b"ACGN"
    codes = np.array([[0, 1, 2, 3, 0, 1, 2, 3, 0]], dtype=np.uint8)
    assert opal_amd.PackedReads.from_codes(codes).data.tolist() == pr.data[:3].tolist()


def test_pattern_parser_errors_like_reference():
    import opal_amd
    with pytest.raises(AssertionError):
        opal_amd.parse_patterns(["2 16\n", "0 1 2\n", "3 4 5\n"], 64)      # row length != row_weight (fasta2skm.py:128)
    with pytest.raises(AssertionError):
        opal_amd.parse_patterns(["1 5\n", "0 1 2 3 4\n"], 64)              # k % row_weight (fasta2skm.py:125)
    assert opal_amd.parse_patterns(None, 5).tolist() == [[0, 1, 2, 3, 4]]


def test_shard_ranges_partition():
    from opal_amd.shard import shard_range
    for n in (0, 1, 7, 100, 12_500_001):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, i, world) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_model_container_roundtrip_on_cpu(tmp_path):
    """opal_amd/model.py without a GPU: header fields, sparse and dense payloads, and the error on foreign files."""
    from opal_amd import model as model_io

    class FakeEngine:                                   # records what load_model hands to the GPU context
        def load_weights(self, w, K, bits):
            self.dense = (np.array(w), K, bits)

        def load_weights_sparse(self, idx, val, K, bits):
            self.sparse = (np.array(idx), np.array(val), K, bits)

        def set_vw_options(self, *a):
            self.opts = a

        def load_accumulators_sparse(self, idx, val):
            self.acc = (np.array(idx), np.array(val))

    rng = np.random.default_rng(0)
    w = np.zeros(1 << 14, dtype=np.float32)
    nz = rng.choice(1 << 14, size=500, replace=False)
    w[nz] = rng.standard_normal(500).astype(np.float32)
    p = str(tmp_path / "vw-model_final.model")
    assert model_io.save_model(p, w, 7, 14, hash_all=True, add_constant=False, clip=True, min_label=-2.0, max_label=3.0) == 500
    h = model_io.read_header(p)
    assert (h["bits"], h["K"], h["nnz"], h["dense"]) == (14, 7, 500, False)
    assert (h["hash_all"], h["add_constant"], h["clip"], h["min_label"], h["max_label"]) == (True, False, True, -2.0, 3.0)
    e = FakeEngine()
    model_io.load_model(p, e)
    idx, val, K, bits = e.sparse
    back = np.zeros_like(w)
    back[idx] = val
    assert np.array_equal(back, w) and (K, bits) == (7, 14) and e.opts == (True, False, True, -2.0, 3.0)
    wd = rng.standard_normal(1 << 10).astype(np.float32)
    p2 = str(tmp_path / "dense_final.model")
    model_io.save_model(p2, wd, 3, 10)
    e2 = FakeEngine()
    assert model_io.load_model(p2, e2)["dense"] and np.array_equal(e2.dense[0], wd)
    # version 2 (`train --save-resume`): the same weights section, then the accumulators and the batch count; a version-1
    # file and a file cut inside the resume section refuse to resume, and a plain load of a version-2 file ignores the section
    g = np.zeros_like(w)
    g[nz] = rng.random(500).astype(np.float32) + 0.5
    p3 = str(tmp_path / "resumable_final.model")
    model_io.save_model(p3, w, 7, 14, g=g, batches=3)
    h3 = model_io.read_header(p3)
    assert (h3["version"], h3["resumable"], h3["batches"], h3["nnz"]) == (2, True, 3, 500)
    assert open(p3, "rb").read()[48:48 + 500 * 8] == open(p, "rb").read()[48:48 + 500 * 8]       # (48 = header size)
    e3 = FakeEngine()
    model_io.load_model(p3, e3, resume=True)
    gi, gv = e3.acc
    gback = np.zeros_like(g)
    gback[gi] = gv
    assert np.array_equal(gback, g) and np.array_equal(e3.sparse[0], idx)
    e4 = FakeEngine()
    model_io.load_model(p3, e4)
    assert not hasattr(e4, "acc") and np.array_equal(e4.sparse[1], val)
    with pytest.raises(RuntimeError, match="no resume state"):
        model_io.load_model(p, FakeEngine(), resume=True)
    cut = tmp_path / "cut_final.model"
    cut.write_bytes(open(p3, "rb").read()[:-100])
    with pytest.raises(RuntimeError, match="truncated"):
        model_io.load_model(str(cut), FakeEngine(), resume=True)
    bad = tmp_path / "x_final.model"
    bad.write_bytes(b"\x06\x00\x00\x008.6.1\x00m")      # how a VW binary regressor starts
    with pytest.raises(RuntimeError):
        model_io.read_header(str(bad))
    # load_model dispatches on the first bytes: a VW binary regressor's header is parsed as far as it goes, then refused
    import struct
    opts = b" --oaa 7 --random_seed 420\0"
    vwbin = (struct.pack("<I", 6) + b"8.6.1\0" + struct.pack("<I", 1) + b"\0" + b"m" + struct.pack("<ffII", -1.0, 1.0, 14, 0) +
             struct.pack("<I", 0) + struct.pack("<I", 0) + struct.pack("<I", len(opts)) + opts + struct.pack("<I", 0xdeadbeef) +
             b"\x00" * 64)
    (tmp_path / "vw_final.model").write_bytes(vwbin)
    assert model_io.model_kind(str(tmp_path / "vw_final.model")) == "vw-binary"
    with pytest.raises(RuntimeError) as ei:
        model_io.load_model(str(tmp_path / "vw_final.model"), FakeEngine())
    msg = str(ei.value)
    assert "version='8.6.1'" in msg and "bits=14" in msg and "--oaa 7" in msg and "header complete" in msg
    with pytest.raises(RuntimeError) as ei:
        model_io.load_model(str(bad), FakeEngine())
    assert "version='8.6.1'" in str(ei.value) and "id string" in str(ei.value) or "32-bit field" in str(ei.value)
    # the --readable_model text form round-trips through its own writer, and load_model accepts it
    txt = str(tmp_path / "readable_final.model")
    assert model_io.export_vw_readable_model(txt, w, 7, 14, min_label=-2.0, max_label=3.0) == 500
    assert model_io.model_kind(txt) == "vw-text"
    m = model_io.import_vw_readable_model(txt)
    back2 = np.zeros_like(w)
    back2[m["idx"]] = m["val"]
    assert np.array_equal(back2, w) and (m["K"], m["bits"], m["min_label"], m["max_label"]) == (7, 14, -2.0, 3.0)
    e3 = FakeEngine()
    h3 = model_io.load_model(txt, e3)
    assert (h3["K"], h3["bits"], h3["nnz"]) == (7, 14, 500) and np.array_equal(np.sort(e3.sparse[0]), np.sort(nz.astype(np.uint32)))


def test_cli_parser_matches_reference_defaults():
    """Flags and defaults of reference opal.py:519-545."""
    from opal_amd import opal
    p = opal.build_parser()
    a = p.parse_args(["train", "in", "out"])
    assert (a.frag_length, a.kmer, a.coverage, a.reverse_complement, a.row_weight, a.num_hash, a.num_batches, a.num_passes,
            a.bits, a.lambda1, a.lambda2, a.hierarchical_weight) == (64, 64, 15.0, False, 16, 8, 1, 1, 31, 0.0, 0.0, -1)
    a = p.parse_args(["simulate", "t", "tr", "o", "-r", "-l", "200", "-c", "0.1", "--do-not-fragment"])
    assert a.do_not_fragment and a.reverse_complement and a.frag_length == 200 and a.coverage == 0.1
    a = p.parse_args(["predict", "m", "t", "p", "-k", "32"])
    assert a.kmer == 32 and not a.reverse_complement and not hasattr(a, "bits")
    a = p.parse_args(["frag", "t", "f"])
    assert (a.frag_length, a.coverage) == (64, 15.0)
    with pytest.raises(ValueError):                      # reference opal.py:248-249
        import argparse
        ns = argparse.Namespace(frag_length=64, coverage=1.0, kmer=64, row_weight=10, hierarchical_weight=-1, num_hash=1,
                                num_batches=1, num_passes=1, bits=20, lambda1=0.0, lambda2=0.0, reverse_complement=False)
        import tempfile
        d = tempfile.mkdtemp()
        open(os.path.join(d, "a.fasta"), "w").write(">g\nACGT\n")
        open(os.path.join(d, "a.taxid"), "w").write("1\n")
        opal.train(d, os.path.join(d, "m"), ns)
    with pytest.raises(RuntimeError):                    # reference opal.py:93-96
        opal.get_fasta_and_taxid(tempfile.mkdtemp())
    # kernel limits are reported before anything is written: --num-hash 64 (65 patterns), a 320-mer
    for bad in (dict(num_hash=64), dict(kmer=320, row_weight=16)):
        d2 = tempfile.mkdtemp()
        open(os.path.join(d2, "a.fasta"), "w").write(">g\nACGT\n")
        open(os.path.join(d2, "a.taxid"), "w").write("1\n")
        ns2 = argparse.Namespace(**{**vars(ns), "row_weight": 16, **bad})
        with pytest.raises(ValueError, match="limit"):
            opal.train(d2, os.path.join(d2, "m"), ns2)
        assert not os.path.exists(os.path.join(d2, "m"))


def test_opal_helpers_match_reference(golden_dir, tmp_path, capsys):
    """unique_lines, extract_column_two, vw_class_to_taxid (+ the array writer) and evaluate_predictions against what the
    reference's own functions returned / wrote / printed (tests/golden/opal_helpers.json, reference opal.py:52-88,111-138)."""
    from opal_amd import opal
    g = _load(golden_dir, "opal_helpers.json")
    a, b, c = str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "c")
    for txt, want in g["unique_lines"]:
        open(a, "w", newline="").write(txt)
        assert opal.unique_lines(a) == want, txt
    for txt, want in g["extract_column_two"]:
        open(a, "w").write(txt)
        opal.extract_column_two(a, b)
        assert open(b).read() == want, txt
    for case in g["vw_class_to_taxid"]:
        open(a, "w").write(case["preds"])
        open(b, "w").write(case["dico"])
        if case["error"]:
            with pytest.raises({"KeyError": KeyError, "ValueError": ValueError}[case["error"]]):
                opal.vw_class_to_taxid(a, b, c)
            if case["error"] == "KeyError":                  # the array writer of predict refuses the same class
                with pytest.raises(KeyError):
                    opal.write_predictions(np.array([int(float(x)) for x in case["preds"].split()]), b, a + ".vw", c)
            continue
        opal.vw_class_to_taxid(a, b, c)
        assert open(c).read() == case["taxids"]
        cls = np.array([int(float(x)) for x in case["preds"].split()], dtype=np.int32)
        opal.write_predictions(cls, b, a + ".vw", c + ".2")
        assert open(c + ".2").read() == case["taxids"]
        assert open(a + ".vw").read() == "".join("%d\n" % x for x in cls)
    capsys.readouterr()
    for case in g["evaluate_predictions"]:
        open(a, "w").write("".join(x + "\n" for x in case["ref"]))
        open(b, "w").write("".join(x + "\n" for x in case["pred"]))
        if case["error"]:
            with pytest.raises(ValueError):
                opal.evaluate_predictions(a, b)
            continue
        opal.evaluate_predictions(a, b)
        assert capsys.readouterr().out == case["stdout"]


def test_training_stream_matches_reference(golden_dir, tmp_path):
    """The example stream train() feeds the learner (class id, genome, position per example, in order) and the dico it
    leaves, against the reference's drawfrag.main -> extract_column_two -> main_generator(taxid, dico) -> random.shuffle
    run in train()'s order for seeds 421, 422, ... (tests/golden/train_stream.json, reference opal.py:329-367)."""
    import opal_amd
    from opal_amd import opal
    from opal_amd.fasta_functions import fasta_reader
    for ci, case in enumerate(_load(golden_dir, "train_stream.json")):
        genomes = [s for _, s in fasta_reader(io.StringIO(case["fasta"]))]
        taxid_lines = case["taxid"].splitlines()
        genome_taxid = [taxid_lines[g] if g < len(taxid_lines) else "" for g in range(len(genomes))]
        tx = tmp_path / ("t%d.taxid" % ci)
        tx.write_text(case["taxid"])
        assert opal.unique_lines(str(tx)) == case["num_labels"]
        dico = str(tmp_path / ("dico%d.txt" % ci))
        rng = opal_amd.Rng()
        L = case["frag_length"]
        for b in case["batches"]:
            vwid, pos, gen = opal.draw_training_batch(rng, b["seed"], [len(x) for x in genomes], genome_taxid, L,
                                                      case["coverage"], dico)
            assert [str(v) for v in vwid.tolist()] == b["labels"]
            assert [genomes[g][p:p + L] for p, g in zip(pos.tolist(), gen.tolist())] == b["fragments"]
            assert open(dico).read() == b["dico_after"]


def test_write_predictions_equals_reference_style_files(tmp_path):
    """write_predictions (array -> `vw -p` file + taxid file) leaves the same bytes as the reference's two steps
    (opal.py:449-451 then vw_class_to_taxid, opal.py:76-88), including the KeyError for a class outside the dico."""
    from opal_amd import opal
    dico = tmp_path / "vw-dico.txt"
    dico.write_text("".join("%s\t%d\n" % (name, c) for c, name in enumerate(["562", "1280", "NC_000913.3", "9"], start=1)))
    cls = np.array([3, 1, 1, 4, 2, 3, 3], dtype=np.int32)
    opal.write_predictions(cls, str(dico), str(tmp_path / "p.vw"), str(tmp_path / "p.taxid"))
    assert (tmp_path / "p.vw").read_text() == "3\n1\n1\n4\n2\n3\n3\n"
    opal.vw_class_to_taxid(str(tmp_path / "p.vw"), str(dico), str(tmp_path / "ref.taxid"))
    assert (tmp_path / "p.taxid").read_bytes() == (tmp_path / "ref.taxid").read_bytes()
    with pytest.raises(KeyError):
        opal.write_predictions(np.array([1, 5]), str(dico), str(tmp_path / "q.vw"), str(tmp_path / "q.taxid"))
    opal.write_predictions(np.zeros(0, dtype=np.int32), str(dico), str(tmp_path / "e.vw"), str(tmp_path / "e.taxid"))
    assert (tmp_path / "e.vw").read_bytes() == b"" and (tmp_path / "e.taxid").read_bytes() == b""


def test_fasta_pieces_preserve_the_reference_readers_records(tmp_path, oracle):
    """predict streams big files in pieces cut in front of header lines (opal.fasta_pieces).  Reading every piece with
    the reference reader's semantics and concatenating the records must give exactly the records of the whole file,
    for every chunk size: CRLF / lone-CR line ends, '+' quality blocks, empty records (mid-file: kept; at the end of
    the file: dropped), text before the first header, a file without any header."""
    from opal_amd import opal
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = [b"junk before the first header\nACGT\n"]
    for i in range(400):
        L = int(rng.integers(0, 300)) if i % 9 else 0                      # every 9th record is empty
        body = bytes(rng.choice(acgt, size=L))
        nl = (b"\n", b"\r\n", b"\r")[i % 3]
        width = int(rng.integers(10, 80))
        parts.append(b">rec%d x" % i + nl + nl.join(body[j:j + width] for j in range(0, L, width)) + (nl if L else b""))
        if i % 7 == 0:
            parts.append(b"+" + nl + b">" * 0 + b"I" * L + nl)
    texts = [b"".join(parts), b"".join(parts) + b">empty_last\n", b"".join(parts[1:]) + b">e1\n>e2\n", b"ACGT\nACGT\n" * 500,
             b">only\nACGT", b""]
    for ti, text in enumerate(texts):
        path = tmp_path / ("t%d.fasta" % ti)
        path.write_bytes(text)
        want = oracle.fasta_read(text)
        for chunk in ((1, 7) if len(text) < 3000 else (7,)) + (64, 1000, 5000, len(text) + 1):
            got, n_pieces = [], 0
            for piece in opal.fasta_pieces(str(path), chunk):
                got += oracle.fasta_read(piece.tobytes())
                n_pieces += 1
            assert [s for _, s in got] == [s for _, s in want], (ti, chunk)
            if 0 < chunk < len(text) // 4 and b">" in text[1:]:
                assert n_pieces > 1
            # the same pieces dealt to 3 ranks: every rank makes the same number of steps, and reading the steps in
            # (round, rank) order gives the file's records
            per_rank = [list(opal.fasta_pieces(str(path), chunk, r, 3)) for r in range(3)]
            assert len({len(x) for x in per_rank}) == 1
            got3 = []
            for rnd in range(len(per_rank[0])):
                for r in range(3):
                    got3 += oracle.fasta_read(per_rank[r][rnd].tobytes())
            assert [s for _, s in got3] == [s for _, s in want], (ti, chunk, "3 ranks")


def test_traffic_json_follows_the_committed_profiles():
    """bench.py quotes roofline.traffic from profiles/traffic.json; the file is generated from the profile summaries
    (scripts/make_traffic_json.py) and must be regenerated whenever a profile is refreshed."""
    import importlib.util
    import json
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_traffic_json", os.path.join(ROOT, "scripts", "make_traffic_json.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        assert json.load(f) == mod.build()
