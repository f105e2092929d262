"""Host-side FASTA/string primitives with the reference's exact semantics (reference util/fasta_functions.py).

These run on the host on purpose: they are file parsing, the part of the path the north-star keeps in Python.
The per-base work (reverse complement, windows, gathers, hashing) happens on the GPU; `reverse_complement`
here exists for host-side tools and tests only.
"""
import re

__all__ = ["fasta_reader", "reverse_complement", "get_all_substrings", "check_acgt", "read_fasta_bytes"]


def fasta_reader(f):
    """Generator of (name, seq) with the quirks of the reference's reader (util/fasta_functions.py:9-43):
    '>' starts a record; lines are concatenated with only the newline stripped; a line starting with '+' mutes
    everything up to the next '>' (FASTQ quality); text before the first '>' is dropped; a record with an empty
    sequence is yielded mid-file but dropped at end of file."""
    seq, name = [], ""
    ignore_line = False
    first_line = True
    for line in f:
        c0 = line[0]
        if c0 == ">":
            if first_line:
                first_line = False
            else:
                yield (name, "".join(seq))
            name = line[1:].rstrip("\n")
            seq = []
            ignore_line = False
        elif c0 == "+":
            ignore_line = True
        elif not ignore_line:
            seq.append(line.rstrip("\n"))
    last = "".join(seq)
    if last != "":
        yield (name, last)


def read_fasta_bytes(path):
    """[(name, seq)] as latin-1 bytes, reading the file the way the reference does (open(path,'r'): universal
    newlines)."""
    with open(path, "r", encoding="latin-1") as f:
        return [(n.encode("latin-1"), s.encode("latin-1")) for n, s in fasta_reader(f)]


_trans = str.maketrans("ATGCatgc", "TACGTACG")


def reverse_complement(dna):
    """util/fasta_functions.py:46-50: lower-case acgt are complemented AND upper-cased, anything else is kept."""
    return dna[::-1].translate(_trans)


def get_all_substrings(input_string, k):
    """Every window of length k, left to right; empty when the string is shorter than k (util/fasta_functions.py:53-54)."""
    n_windows = len(input_string) - k + 1
    return [input_string[start:start + k] for start in range(max(n_windows, 0))]


_ONLY_ACGT = re.compile(r"[ACGTacgt]*$")   # like the reference, `$` also accepts one trailing newline


def check_acgt(s):
    """Truthy when s holds nothing but A/C/G/T in either case (util/fasta_functions.py:57-61)."""
    return _ONLY_ACGT.match(s)
