"""The ray-order radix sort (rayzath_amd/csrc/hiprz_sort.hip) checked directly through hiprz_selftest_sort: a STABLE permutation in key
order, for the key distributions the passes meet (pixel-ordered keys with random direction bits, keys already sorted by their low
digits, one value only, the dead-ray sentinel) and for ragged sizes.  The frames only show a wrong permutation (a pixel traced twice,
another never); an unstable one would be invisible there, and the tree builder relies on stability across its four digit passes."""
import numpy as np
import pytest

from rayzath_amd.engine import Context

pytestmark = pytest.mark.gpu


def key_patterns(n, rng):
    i = np.arange(n, dtype=np.uint64)
    cell = ((i // 256) * 2654435761 >> 7) & 0x7FFF          # one origin cell per 256 consecutive pixels
    yield "uniform", rng.integers(0, 1 << 24, n, dtype=np.uint32)
    yield "cells_with_random_directions", ((cell << 9) | rng.integers(0, 512, n, dtype=np.uint64)).astype(np.uint32)
    yield "one_value", np.full(n, 0x00ABCDEF, dtype=np.uint32)
    yield "mostly_dead_rays", np.where(rng.random(n) < 0.9, 0x00FFFFFE, rng.integers(0, 1 << 24, n, dtype=np.uint32)).astype(np.uint32)
    yield "falling", (np.uint32(0x00FFFFFF) - (i & 0xFFFFFF).astype(np.uint32))
    yield "two_digits_only", (rng.integers(0, 2, n, dtype=np.uint32) * np.uint32(0x00010100))


@pytest.mark.parametrize("n", [1, 63, 257, 4096, 4097, 12289, 300_001, 2_073_600])
def test_sort_is_a_stable_permutation_in_key_order(built, n):
    ctx = Context(0)
    rng = np.random.default_rng(n)
    for name, keys in key_patterns(n, rng):
        bad, us = ctx.selftest_sort(keys, 24)
        assert bad == 0, f"{name}, n={n}: {bad} violations"


def test_sort_beyond_one_chunk_of_tile_counts(built):
    """More than 2 048 tiles of 4 096 keys: the offsets kernel scans a digit's row of tile counts in several chunks and carries the sum on
    (an 8K frame, or two 4K cameras' worth of rays in one context)."""
    ctx = Context(0)
    n = 9_000_001
    rng = np.random.default_rng(7)
    for name, keys in (("uniform", rng.integers(0, 1 << 24, n, dtype=np.uint32)),
                       ("few_digits", (rng.integers(0, 3, n, dtype=np.uint32) * np.uint32(0x00400001)))):
        bad, us = ctx.selftest_sort(keys, 24)
        assert bad == 0, f"{name}: {bad} violations"


@pytest.mark.parametrize("bits", [1, 8, 9, 16, 24, 25, 32])
def test_sort_over_every_digit_count(built, bits):
    """1..4 digit passes; keys carry bits above the ones the sort is told about — it looks at whole bytes, the check does too."""
    ctx = Context(0)
    rng = np.random.default_rng(bits)
    keys = rng.integers(0, 1 << 32, 70_001, dtype=np.uint64).astype(np.uint32)
    bad, us = ctx.selftest_sort(keys, bits)
    assert bad == 0


def test_sort_selftest_reports_a_broken_order(built):
    """The check itself: handed keys, it must accept a correct sort — and the host-side checker is exercised against numpy's stable sort."""
    ctx = Context(0)
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 16, 50_000, dtype=np.uint32)
    bad, us = ctx.selftest_sort(keys, 16, repeats=3)
    assert bad == 0 and us > 0.0
    with pytest.raises(Exception):
        ctx.selftest_sort(keys, 0)
