"""Mirror of the reference's util/drawfrag.py: from each FASTA record draw random substrings of size l covering it
c times (util/drawfrag.py:15-49).  Same argv, same output files, same CPython `random` stream: the positions come
from libopal_hip.so's CPython-exact MT19937 (opal_sample_positions / opal_rng_randbelow).
"""
import argparse
import sys

from .engine import Rng
from .fasta_functions import check_acgt, fasta_reader

__version__ = "0.0.2"
__all__ = ["main", "main_not_commandline", "sample_fragments", "write_fragment_files"]


def sample_fragments(records, taxid_lines, size, coverage, seed, atgc=False, rng=None):
    """records: [(name, seq)] str; taxid_lines: one label per record (missing -> '').  Returns
    (fragments [str], labels [(firstname, taxid)]) in the reference's order.  rng: an opal_amd.Rng to draw
    from (the reference uses the process-wide `random`, which opal.py:363 keeps drawing from for its shuffle)."""
    rng = rng or Rng()
    if seed:                                   # drawfrag.py:23-24: seed 0/None leaves the stream unseeded
        rng.seed(seed)
    frags, labels = [], []
    if not atgc:
        pos, gen = rng.sample_positions([len(s) for _, s in records], size, coverage)
        for p, g in zip(pos.tolist(), gen.tolist()):
            name, seq = records[g]
            frags.append(seq[p:p + size])
            labels.append((name.split()[0], taxid_lines[g] if g < len(taxid_lines) else ""))
        return frags, labels
    # --atgc (never enabled by opal.py): rejected draws still consume the stream (drawfrag.py:36-49)
    for g, (name, seq) in enumerate(records):
        if len(seq) < size:
            continue
        cov, tries, want = 0, 0, coverage * len(seq)
        while cov < want:
            tries += 1
            p = rng.randint(0, len(seq) - size)
            sample = seq[p:p + size]
            if check_acgt(sample):
                cov += size
                frags.append(sample)
                labels.append((name.split()[0], taxid_lines[g] if g < len(taxid_lines) else ""))
            if tries > 10 * len(seq):
                break
    return frags, labels


def _fasta_block(first_num, frags):
    """Bytes of `>{n}\n{fragment}\n` for the rows of frags (uint8 [m, size]) numbered first_num, first_num+1, ...:
    one fixed-width byte matrix per run of numbers with the same digit count."""
    import numpy as np
    m, size = frags.shape
    parts, done = [], 0
    while done < m:
        n0 = first_num + done
        w = len(str(n0))
        cnt = min(m - done, 10 ** w - n0)                 # numbers that still have w digits
        block = np.empty((cnt, w + size + 3), dtype=np.uint8)
        block[:, 0] = ord(">")
        nums = np.arange(n0, n0 + cnt, dtype=np.int64)
        for d in range(w):
            block[:, w - d] = ord("0") + (nums // 10 ** d) % 10
        block[:, w + 1] = ord("\n")
        block[:, w + 2:w + 2 + size] = frags[done:done + cnt]
        block[:, w + 2 + size] = ord("\n")
        parts.append(block.tobytes())
        done += cnt
    return b"".join(parts)


def write_fragment_files(records, taxid_lines, size, coverage, seed, out_path, g2t_path, rng=None, chunk=1 << 20):
    """The two files drawfrag.py:44-47 writes, byte for byte, without a Python loop over fragments: positions from the
    CPython-exact sampler, fragment text gathered per record with numpy.  Returns the number of fragments."""
    import numpy as np
    rng = rng or Rng()
    if seed:
        rng.seed(seed)
    pos, gen = rng.sample_positions([len(s) for _, s in records], size, coverage)
    ar = np.arange(size, dtype=np.int64)
    bounds = np.flatnonzero(np.diff(gen.astype(np.int64), prepend=-1, append=-1))   # runs of equal genome index
    num = 1
    with open(out_path, "wb") as out, open(g2t_path, "wb") as g2t:
        for a, b in zip(bounds[:-1].tolist(), bounds[1:].tolist()):
            g = int(gen[a])
            name, seq = records[g]
            arr = np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)
            for s in range(a, b, chunk):
                p = pos[s:min(b, s + chunk)].astype(np.int64)
                out.write(_fasta_block(num, arr[p[:, None] + ar[None, :]]))
                num += len(p)
            line = "%s\t%s\n" % (name.split()[0], taxid_lines[g] if g < len(taxid_lines) else "")
            g2t.write(line.encode("latin-1") * (b - a))
    return num - 1


def main_not_commandline(args):
    with open(args.input, "r", encoding="latin-1") as f:
        records = list(fasta_reader(f))
    with open(args.taxids, "r") as f:
        taxid_lines = [l.rstrip("\n") for l in f]
    if not args.atgc and args.size > 0:
        write_fragment_files(records, taxid_lines, args.size, args.coverage, args.seed, args.output, args.gi2taxid)
        return
    frags, labels = sample_fragments(records, taxid_lines, args.size, args.coverage, args.seed, args.atgc)
    with open(args.output, "w", encoding="latin-1") as out, open(args.gi2taxid, "w", encoding="latin-1") as g2t:
        out.write("".join(">%d\n%s\n" % (i + 1, s) for i, s in enumerate(frags)))
        g2t.write("".join("%s\t%s\n" % nl for nl in labels))


# same option letters/names as reference util/drawfrag.py:59-66
_OPTIONS = (
    ("-i", "--input", dict(help="fasta file to sample from")),
    ("-l", "--size", dict(type=int, help="fragment length")),
    ("-t", "--taxids", dict(help="one taxid per fasta record, one per line")),
    ("-c", "--coverage", dict(type=float, help="target coverage of every record")),
    ("-g", "--gi2taxid", dict(help="output: '<first word of the record name>\\t<taxid>' per fragment")),
    ("-s", "--seed", dict(type=int, help="seed of the CPython random stream; unset or 0 leaves it unseeded")),
    ("-o", "--output", dict(help="output: fragments as fasta records named 1, 2, ...")),
)


def main(argv):
    parser = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter, description=__doc__)
    parser.add_argument('--version', action='version', version='%(prog)s ' + __version__)
    for short, long_, kw in _OPTIONS:
        parser.add_argument(short, long_, **kw)
    parser.add_argument('--atgc', action='store_true', help='keep only fragments made of A/C/G/T (either case)')
    args = parser.parse_args(argv)
    print(args)
    main_not_commandline(args)


if __name__ == "__main__":
    main(sys.argv[1:])
