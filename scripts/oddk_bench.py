#!/usr/bin/env python3
"""Predict throughput by class count K, for K that are and are not multiples of 4 (K = distinct taxids, reference
opal.py:288: any value).  Same table, fragments and geometry as bench.py's cfg2 / cfg3, one process, HIP events.

    python scripts/oddk_bench.py [--nfrag 1000000] [--steps 3] [K ...]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("classes", nargs="*", type=int, default=[355, 356, 357, 358, 1562, 1563, 1564, 1565])
    ap.add_argument("--nfrag", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pattern", default="patterns_k64_t16_m1_s7.txt")
    args = ap.parse_args()
    B = bench.Bench(argparse.Namespace(gpus=1))
    for K in args.classes:
        wl = dict(K=K, n=args.nfrag if K < 1000 else max(args.nfrag // 4, 1), pattern=args.pattern, bits=31, mode="predict", desc="K sweep")
        r, info = B.measure("ksweep", wl, args.steps, args.warmup, None)
        del info
        print(json.dumps({"K": K, "kernel": r["kernel"], "fragments": wl["n"], "kernel_ms": r["roofline"]["kernel_ms"],
                          "fragments_per_s": r["value"], "hbm_frac": r["roofline"]["frac"]}), flush=True)


if __name__ == "__main__":
    main()
