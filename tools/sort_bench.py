#!/usr/bin/env python3
"""Device time of the ray-order radix sort (hiprz_selftest_sort) at the sizes and key distributions of the configs' frames.
usage: [SORT_BENCH_CASES=uniform24,coherent24] python3 tools/sort_bench.py [n ...]   (default: 2 073 600 and 8 294 400 keys, 24 bits, + 3 000 000 keys of 32 bits as the tree builder sorts)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayzath_amd.engine import Context  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2_073_600, 8_294_400]
    ctx = Context(0)
    rng = np.random.default_rng(1)
    out = {}
    for n in sizes:
        i = np.arange(n, dtype=np.uint64)
        cell = ((i // 256) * 2654435761 >> 7) & 0x7FFF
        cases = {
            "uniform24": (rng.integers(0, 1 << 24, n, dtype=np.uint32), 24),
            "cells_random_directions24": (((cell << 9) | rng.integers(0, 512, n, dtype=np.uint64)).astype(np.uint32), 24),
            "coherent24": (((cell << 9) | ((i // 64) & 511)).astype(np.uint32), 24),
            "uniform32": (rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), 32),
        }
        only = os.environ.get("SORT_BENCH_CASES")
        for name, (keys, bits) in cases.items():
            if only and name not in only.split(","):
                continue
            bad, us = ctx.selftest_sort(keys, bits, repeats=5)
            out[f"{name}_n{n}"] = {"us": round(us, 1), "violations": bad, "Mkeys_per_s": round(n / us, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
