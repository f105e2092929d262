"""Deterministic synthetic fragments for the hash-only corpus (tests/golden/skm_corpus.json).

Shared by tests/golden/make_golden.py (which runs the REFERENCE over these fragments and stores digests of what it
produced) and by the tests (which rebuild the same fragments and compare the oracle's / the GPU's output with those
digests).  The byte stream comes from SHA-256 in counter mode, so it does not depend on any library's RNG.
"""
import hashlib

ALPHABETS = {"acgt": b"ACGT", "mixed": b"ACGTacgtNn", "iupac": b"ACGTRYKMSWBDHVN"}


def _stream(tag: bytes, nbytes: int) -> bytes:
    out = bytearray()
    ctr = 0
    while len(out) < nbytes:
        out += hashlib.sha256(tag + b"/%d" % ctr).digest()
        ctr += 1
    return bytes(out[:nbytes])


def fragments(case: str, per_alphabet: int, max_len: int = 600):
    """-> list of bytes: per_alphabet fragments for each alphabet (in ALPHABETS' order), lengths in 0..max_len drawn
    from the stream (the first few lengths are fixed: 0, k-1, k, k+1 for k = 64, the fast path's limit 512 and 513)."""
    fixed = [0, 63, 64, 65, 512, 513, max_len]
    seqs = []
    for name, alpha in ALPHABETS.items():
        tag = (case + ":" + name).encode()
        lens_raw = _stream(tag + b":len", 2 * per_alphabet)
        for i in range(per_alphabet):
            n = fixed[i] if i < len(fixed) else (lens_raw[2 * i] | (lens_raw[2 * i + 1] << 8)) % (max_len + 1)
            raw = _stream(tag + b":seq%d" % i, n)
            seqs.append(bytes(alpha[b % len(alpha)] for b in raw))
    if not seqs[-1]:
        seqs[-1] = b"ACGT"      # fasta_reader drops an empty LAST record (util/fasta_functions.py:20-22)
    return seqs


def fasta_text(seqs) -> str:
    return "".join(">%d\n%s\n" % (i + 1, s.decode("ascii")) for i, s in enumerate(seqs))
