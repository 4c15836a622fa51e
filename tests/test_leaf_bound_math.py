"""The bound behind `CAND_LEAF_BOUND` (kernels/wave_cand_fixed.inc): 10 + the sum over the 64 leaves of a 4096-sample block of
   k c + (2 S >> k) - 6   (k = the rule's parameter ceil(log2(S / c)), encode.rs:3777-3780; 2 S where S <= c)
never exceeds the bits of the residual block at ANY partition order 0..6 -- Rice partitions with the rule's parameter, escaped
partitions (parameter >= 15 / 31, encode.rs:3787-3800) and all-zero ones -- whatever the residual looks like.  A numpy
restatement of both sides over adversarial distributions (CPU only: the arithmetic, not the kernel)."""
import numpy as np
import pytest

N, LEAF = 4096, 64


def rule_k(c, s):
    k = 0
    while (c << k) < s:
        k += 1
    return k


def block_bits(r, order, level, rice_max):
    """exact bits of the residual block coded at partition order `level` (method + order fields included)"""
    u = np.where(r >= 0, 2 * r, -2 * r - 1).astype(np.int64)
    a = np.abs(r).astype(np.int64)
    plen = N >> level
    bits = 6
    for p in range(1 << level):
        lo, hi = max(p * plen, order), (p + 1) * plen
        c, s = hi - lo, int(a[lo:hi].sum())
        if c <= 0:
            return None               # (the reference refuses such a level)
        if s == 0:
            bits += 4 + 5             # escaped with zero bits per residual
            continue
        k = rule_k(c, s) if s > c else 0
        if k >= rice_max:
            e = int(s).bit_length() - 1 + 2
            bits += 4 + 5 + e * c
        else:
            bits += 4 + int(((u[lo:hi] >> k) + k + 1).sum())
    return bits


def leaf_bound(r, order):
    a = np.abs(r).astype(np.int64)
    lb = 10
    for leaf in range(N // LEAF):
        lo, hi = max(leaf * LEAF, order), (leaf + 1) * LEAF
        c, s = hi - lo, int(a[lo:hi].sum())
        if s <= c:
            lb += 2 * s
        else:
            k = rule_k(c, s)
            lb += k * c + ((2 * s) >> k) - 6
    return lb


def cases(rng):
    for scale in (0.3, 0.6, 1.0, 1.7, 3, 10, 100, 5000, 2 ** 20, 2 ** 27):
        yield np.rint(rng.laplace(0, scale, N)).astype(np.int64)
        yield np.rint(rng.normal(0, scale, N)).astype(np.int64)
        x = np.zeros(N, dtype=np.int64)                                   # spikes in silence
        x[rng.integers(0, N, 40)] = int(scale * 50) + 1
        yield x
        yield np.full(N, int(scale) + 1, dtype=np.int64)                  # constant magnitude (escape beats Rice)
        yield (np.rint(rng.laplace(0, scale, N)) * (rng.random(N) < 0.1)).astype(np.int64)
        env = np.repeat(rng.random(N // LEAF) ** 6, LEAF) * scale * 40      # loud and quiet leaves side by side
        yield np.rint(rng.laplace(0, 1, N) * env).astype(np.int64)
    yield np.zeros(N, dtype=np.int64)


@pytest.mark.parametrize("rice_max", [15, 31])
def test_leaf_bound_never_exceeds_the_block(rice_max):
    rng = np.random.default_rng(20260105)
    checked = 0
    for r in cases(rng):
        r = np.clip(r, -(2 ** 31) + 1, 2 ** 31 - 1)
        for order in (0, 2, 4):
            lb = leaf_bound(r, order)
            for level in range(7):
                b = block_bits(r, order, level, rice_max)
                if b is None:
                    continue
                assert lb <= b, (order, level, lb, b, int(np.abs(r).sum()))
                checked += 1
    assert checked > 1000
