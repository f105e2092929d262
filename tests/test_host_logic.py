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
