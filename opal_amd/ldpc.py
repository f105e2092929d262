"""LDPC (Gallager-style) spaced-k-mer pattern generation: host-side, runs once per training.

Mirror of the reference's util/ldpc.py interface: ldpc(k,t,_m) (util/ldpc.py:10-37), write_out(H,d,_m) (:66-92),
ldpc_write(k,t,_m,d) (:94-97).  The construction, the numpy global-RNG call sequence (one np.random.permutation(k)
per extra block) and the patterns.txt text format are preserved, so the same np.random.seed gives the same file.
"""
import numpy as np

__all__ = ["ldpc", "write_out", "ldpc_write", "pattern_rows"]


def _dims(k, t, _m):
    rows_per_block = k // t
    m = (int(np.ceil(_m * 1.0 / (k / t)) + 1)) * rows_per_block   # total rows of H
    w = m * t // k                                                # number of blocks (ones per column)
    return m, w


def ldpc(k, t, _m):
    """0/1 matrix: a block-diagonal base block (row i covers positions [i*t,(i+1)*t)) stacked with w-1 copies whose
    columns are randomly permuted, so the rows of every block partition the k positions."""
    if k % t != 0:
        raise ValueError('k should be multiple of t!')
    m, w = _dims(k, t, _m)
    base = np.repeat(np.eye(m // w, dtype=bool), t, axis=1)
    blocks = [base]
    for _ in range(w - 1):
        blocks.append(base[:, np.random.permutation(k)])
    return np.vstack(blocks)


def pattern_rows(H, _m):
    """The position lists patterns.txt holds: the contiguous prefix 0..t-1, then the first _m rows of the permuted
    blocks, positions ascending."""
    k = H.shape[1]
    t = int(H[0].sum())
    m, w = _dims(k, t, _m)
    first_permuted = m // w
    return [list(range(t))] + [np.flatnonzero(H[first_permuted + i]).tolist() for i in range(_m)]


def write_out(H, d, _m):
    """patterns.txt: '<_m+1> <t>' then one line per pattern, every position followed by a space."""
    rows = pattern_rows(H, _m)
    with open(d, 'w') as fout:
        fout.write('%d %d\n' % (_m + 1, len(rows[0])))
        for row in rows:
            fout.write(''.join('%d ' % j for j in row) + '\n')


def ldpc_write(k, t, _m, d):
    write_out(ldpc(k, t, _m), d, _m)
