"""The entry point that allocates nothing (opal_feature_offsets_noalloc) against opal_feature_offsets, through the C-ABI:
bit-exact, on two calls in a row, the second with a smaller input (the static scratch keeps the first call's contents);
and the property it exists for: no device memory is allocated by the call, and it can be captured into a HIP graph."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _engine(opal_amd):
    eng = opal_amd.Engine(0)
    with open(os.path.join(GOLD, "patterns_k64_t16_m4_s7.txt")) as f:
        eng.set_patterns(opal_amd.parse_patterns(f.readlines(), 64), 64)
    return eng


def _want(ln, n, reverse):
    want = np.zeros(n + 1, dtype=np.int64)
    L = (ln[:n] & np.uint32(0x7FFFFFFF)).astype(np.int64)
    want[1:] = np.cumsum(np.where(L >= 64, (L - 63) * 5 * (2 if reverse else 1), 0))
    return want


@pytest.mark.gpu
def test_feature_offsets_noalloc_matches_allocating_entry_point():
    import torch
    import opal_amd
    from opal_amd._lib import OPAL_ERANGE, check, lib

    eng = _engine(opal_amd)
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
        assert np.array_equal(ref.cpu().numpy(), _want(ln, n, reverse))
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


@pytest.mark.gpu
def test_feature_offsets_noalloc_allocates_nothing():
    """A fresh ctx whose allocating opal_feature_offsets has never run (so there is no ctx-owned tile buffer to fall back
    on), a batch far beyond that buffer's initial 1 024 tiles (50 M fragments = 24 415 tiles = 190 KiB of totals), the
    device's free memory read before and after."""
    import torch
    import opal_amd
    from opal_amd._lib import check, lib

    eng = _engine(opal_amd)
    dev = eng.device
    n = 50_000_000
    d_len = torch.randint(0, 700, (n,), dtype=torch.int32, device=dev)
    got = torch.empty(n + 1, dtype=torch.int64, device=dev)
    warm = torch.empty(9, dtype=torch.int64, device=dev)
    check(lib().opal_feature_offsets_noalloc(eng._h, d_len.data_ptr(), 8, 1, warm.data_ptr()), eng._h)   # kernels' code objects loaded
    torch.cuda.synchronize(dev)
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(3):
        check(lib().opal_feature_offsets_noalloc(eng._h, d_len.data_ptr(), n, 1, got.data_ptr()), eng._h)
    torch.cuda.synchronize(dev)
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, (free0, free1)
    want = _want(d_len.cpu().numpy().view(np.uint32), n, True)
    assert np.array_equal(got.cpu().numpy(), want)
    # (hipMemGetInfo moves in the runtime's own chunks: the 190 KiB the allocating entry point takes for the same batch do not
    # show in it, so the equality above only rules out large allocations; the capture test below is the strict one)
    ref = torch.empty(n + 1, dtype=torch.int64, device=dev)
    check(lib().opal_feature_offsets(eng._h, d_len.data_ptr(), n, 1, ref.data_ptr()), eng._h)
    torch.cuda.synchronize(dev)
    assert torch.equal(ref, got)
    eng.close()


@pytest.mark.gpu
def test_feature_offsets_noalloc_under_graph_capture():
    """The call on a capturing stream: capture must succeed (an allocation or a synchronisation inside it would abort the
    capture), and replaying the graph on new lengths gives the allocating entry point's offsets."""
    import torch
    import opal_amd
    from opal_amd._lib import check, lib

    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(dev)
    rng = np.random.default_rng(11)
    n = 300_000
    with torch.cuda.stream(side):
        eng = _engine(opal_amd)                               # the ctx binds to the current stream: `side`
        d_len = torch.zeros(n, dtype=torch.int32, device=dev)
        got = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        check(lib().opal_feature_offsets_noalloc(eng._h, d_len.data_ptr(), n, 1, got.data_ptr()), eng._h)   # warm-up outside the capture
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        check(lib().opal_feature_offsets_noalloc(eng._h, d_len.data_ptr(), n, 1, got.data_ptr()), eng._h)
    for _ in range(2):
        ln = rng.integers(0, 700, size=n, dtype=np.uint32)
        d_len.copy_(torch.from_numpy(ln.view(np.int32)))
        torch.cuda.synchronize(dev)
        graph.replay()
        torch.cuda.synchronize(dev)
        assert np.array_equal(got.cpu().numpy(), _want(ln, n, True))
    del graph
    eng.close()
