"""The entry point that allocates nothing (opal_feature_offsets_noalloc) against opal_feature_offsets, through the C-ABI:
bit-exact, on two calls in a row, the second with a smaller input (the static scratch keeps the first call's contents)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.gpu
def test_feature_offsets_noalloc_matches_allocating_entry_point():
    import torch
    import opal_amd
    from opal_amd._lib import OPAL_ERANGE, check, lib

    eng = opal_amd.Engine(0)
    with open(os.path.join(GOLD, "patterns_k64_t16_m4_s7.txt")) as f:
        eng.set_patterns(opal_amd.parse_patterns(f.readlines(), 64), 64)
    dev = eng.device
    rng = np.random.default_rng(5)

    def both(n, reverse):
        ln = rng.integers(0, 700, size=max(n, 1), dtype=np.uint32)
        ln[rng.random(max(n, 1)) < 0.1] |= np.uint32(0x80000000)           # raw-flagged fragments count the same way
        d_len = torch.from_numpy(ln.view(np.int32)).to(dev)
        ref = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
        got = torch.full((n + 1,), -2, dtype=torch.int64, device=dev)
        check(lib().opal_feature_offsets(eng._h, d_len.data_ptr(), n, int(reverse), ref.data_ptr()), eng._h)
        check(lib().opal_feature_offsets_noalloc(eng._h, d_len.data_ptr(), n, int(reverse), got.data_ptr()), eng._h)
        eng.sync()
        want = np.zeros(n + 1, dtype=np.int64)
        L = (ln[:n] & np.uint32(0x7FFFFFFF)).astype(np.int64)
        want[1:] = np.cumsum(np.where(L >= 64, (L - 63) * 5 * (2 if reverse else 1), 0))
        assert np.array_equal(ref.cpu().numpy(), want)
        assert torch.equal(ref, got)

    both(3_000_001, True)      # 1 465 tiles: the scratch is left full of this call's totals
    both(70_001, False)        # smaller input right after: must not see the stale totals
    both(1, True)
    both(0, True)
    # beyond the static maximum: a clear error, nothing launched
    d_len = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    rc = lib().opal_feature_offsets_noalloc(eng._h, d_len.data_ptr(), 134217728 + 1, 1, out.data_ptr())
    assert rc == OPAL_ERANGE
    assert b"static scratch" in lib().opal_last_error(eng._h)
    eng.close()
