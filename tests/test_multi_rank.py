"""N>1 path on CPU: two gloo ranks shard a fragment batch, score their shard, rank 0 concatenates in rank order.

The GPU scorer cannot run here, so each rank uses the CPU oracle as the stand-in scorer; what is under test is the
product's sharding + gather code (opal_amd/shard.py), which is all the multi-GPU design consists of.
"""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from opal_amd.shard import gather_in_rank_order, shard_range
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)                      # same batch on every rank
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [bytes(rng.choice(acgt, size=int(L))) for L in rng.integers(60, 260, size=101)]
    with open(os.path.join(ROOT, "tests", "golden", "patterns_k64_t16_m1_s7.txt")) as f:
        pos = orc.parse_patterns(f.readlines(), 64)
    K, bits = 12, 14
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    lo, hi = shard_range(len(seqs), rank, world)
    cls, _, _ = orc.predict(seqs[lo:hi], 64, pos, True, w, K, bits, want_scores=False)
    full = gather_in_rank_order(cls, dist, dst=0)
    if rank == 0:
        want, _, _ = orc.predict(seqs, 64, pos, True, w, K, bits, want_scores=False)
        ok = full is not None and np.array_equal(full, want)
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): the N ranks are started as CHILD processes of
    torch.distributed.run, the driver's own command line, with the caller's arguments; with fewer GPUs than ranks (none
    here) the rendezvous falls back to gloo.  Nothing in that path may initialise the GPU."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("OPAL_DIST_BACKEND", raising=False)
    assert bench.launch_ranks(4) == 7                                       # the launcher's exit code comes back
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["OPAL_DIST_BACKEND"] == "gloo"                       # no GPU in this container: rehearsal mode
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
