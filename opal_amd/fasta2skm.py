"""Mirror of the reference's util/fasta2skm.py interface on top of the GPU engine.

Same names, argument meaning and error behaviour as the reference module:
  update_dictionary(labels, dico_file)      util/fasta2skm.py:19-44
  create_pattern_getters(lines, kmer)       util/fasta2skm.py:120-134  (returns the [P,t] position table)
  main_generator(args)                      util/fasta2skm.py:47-84    (yields the same VW text lines)
The fast path never builds text: opal.predict() feeds packed fragments straight to Engine.predict().  The text
generator exists for parity tests and for `--dump-vw-text`; the characters of every feature are gathered on the
GPU (opal_feature_names), only the string formatting happens here.
"""
import argparse
import itertools
import os
import sys

import numpy as np

from .engine import Engine, PackedReads, check_geometry, parse_patterns
from .fasta_functions import fasta_reader

__all__ = ["update_dictionary", "create_pattern_getters", "main_generator", "main_not_commandline", "main",
           "vw_text_lines", "load_labels"]
__version__ = "0.0.2"

create_pattern_getters = parse_patterns


def update_dictionary(labels, dico_file):
    """Extends the label -> class id table kept in `dico_file` with the labels of `labels` (an iterable of lines) and
    returns it.  Behaviour of util/fasta2skm.py:19-44 as the golden cases record it: rows of an existing file are
    `<label> <id>` (first two words; a later row wins), a label not seen before gets the largest id so far plus one
    (1 in an empty table), the file is rewritten as `<label>\t<id>` rows in first-seen order."""
    table, top = {}, None
    if os.path.isfile(dico_file):
        with open(dico_file, "r") as fh:
            for row in fh:
                name, num = row.strip().split()[:2]          # ValueError on a row with fewer than two words
                table[name] = int(num)
                top = table[name] if top is None else max(top, table[name])
    for row in labels:
        name = row.rstrip("\n")
        if name not in table:
            top = 1 if top is None else top + 1
            table[name] = top
    with open(dico_file, "w") as fh:
        fh.writelines("%s\t%d\n" % item for item in table.items())
    return table


def vw_text_lines(engine: Engine, seqs, reverse, labels=None, chunk=4096):
    """VW text lines (bytes, each ending in '\\n') for sequences `seqs` (list of bytes): '<label> | f f f\\n' with
    f = <t gathered chars><pattern index> (util/fasta2skm.py:81-82,140).  labels: iterable or None (-> 1)."""
    labels = iter(labels) if labels is not None else itertools.repeat(1)
    P, t = engine.P, engine.t
    suffix = [str(i).encode() for i in range(P)]
    out = []
    for c0 in range(0, len(seqs), chunk):
        part = seqs[c0:c0 + chunk]
        batch = engine.upload(PackedReads.from_sequences(part))
        names, off = engine.feature_names(batch, reverse)
        names = names.cpu().numpy()
        off = off.cpu().numpy()
        if P <= 10:
            # one-digit pattern index: every feature is t + 2 bytes "<t chars><digit> ", laid out for the whole chunk at once
            # (every fragment's feature count is a multiple of P, so the pattern index is the global feature index mod P)
            cells = np.empty((names.shape[0], t + 2), dtype=np.uint8)
            cells[:, :t] = names
            cells[:, t] = ord("0") + np.arange(names.shape[0]) % P
            cells[:, t + 1] = ord(" ")
            for i in range(len(part)):
                out.append(("%s | " % next(labels)).encode() + cells[off[i]:off[i + 1]].tobytes()[:-1] + b"\n")
            continue
        for i in range(len(part)):
            rows = names[off[i]:off[i + 1]]
            feats = [rows[f].tobytes() + suffix[f % P] for f in range(rows.shape[0])]
            out.append(("%s | " % next(labels)).encode() + b" ".join(feats) + b"\n")
    return out


def load_labels(args):
    """The label stream of main_generator (util/fasta2skm.py:55-60,70-74)."""
    if args.taxid:
        if not args.dico:
            raise ValueError("If --taxid is set, then so must be --dico")
        with open(args.taxid, "r") as taxid_file:
            label2vwid = update_dictionary(taxid_file, args.dico)
        with open(args.taxid, "r") as taxid_file:
            return [label2vwid[l.rstrip("\n")] for l in taxid_file]
    return None


def main_generator(args, engine=None):
    """Yields the VW text lines the reference yields for the same Namespace(input, taxid, kmer, dico, pattern,
    reverse).  Needs a GPU: the gathers run in libopal_hip.so."""
    if not args.input or not args.kmer:
        raise ValueError("fasta2skm requires input and kmer arguments")
    labels = load_labels(args)
    if args.pattern:
        with open(args.pattern, "r") as pattern_file:
            file_contents = pattern_file.readlines()
    else:
        file_contents = None
    pos = create_pattern_getters(file_contents, args.kmer)
    check_geometry(args.kmer, pos.shape[1], pos.shape[0])
    eng = engine or Engine(0)
    eng.set_patterns(pos, args.kmer)
    with open(args.input, "r", encoding="latin-1") as input_file:
        seqs = [s.encode("latin-1") for _, s in fasta_reader(input_file)]
    lab = iter(labels) if labels is not None else None
    for line in vw_text_lines(eng, seqs, bool(args.reverse), lab):
        yield line.decode("latin-1")


def main_not_commandline(args):
    """Writes the lines of main_generator to args.output ('-' = stdout), like util/fasta2skm.py:87-96."""
    out = sys.stdout if args.output == "-" else open(args.output, "w", encoding="latin-1")
    try:
        for line in main_generator(args):
            out.write(line)
    finally:
        if out is not sys.stdout:
            out.close()


def main(argv):
    """Same options as the reference's stand-alone tool (util/fasta2skm.py:99-117)."""
    parser = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter, description=__doc__)
    parser.add_argument('--version', action='version', version='%(prog)s ' + __version__)
    for short, long_, kw in (("-i", "--input", dict(help="fasta/fastq file to featurise")),
                             ("-k", "--kmer", dict(type=int, help="window length")),
                             ("-t", "--taxid", dict(help="one label per record; without it every line is labelled 1")),
                             ("-p", "--pattern", dict(help="patterns.txt (LDPC positions); without it the whole k-mer is one feature")),
                             ("-d", "--dico", dict(help="label <-> class id table, created/extended (needed with --taxid)")),
                             ("-o", "--output", dict(default="-", help="output file, '-' = stdout"))):
        parser.add_argument(short, long_, **kw)
    parser.add_argument('-r', '--reverse', action='store_true', help='append the reverse complement strand\'s features')
    main_not_commandline(parser.parse_args(argv))


if __name__ == "__main__":
    main(sys.argv[1:])
