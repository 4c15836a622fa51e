"""The arithmetic the residual hand-over rests on (kernels/wave_cand.inc wave_rice_fold<SIGNS> /
wave_rice_exact_stored<SIGNS>, kernels/pack.inc), checked in numpy on the CPU -- no GPU, no oracle:

  fold      t = r ^ (r >> 31) = zigzag(r) >> 1 < 2^31, kept with r's sign in bit 31:  w = t | (r & 2^31)
  search    the exact bit count sums t >> sh over a lane's samples (sh = k - 1, k the partition's Rice parameter); summed over w
            instead, every negative sample adds 2^(31 - sh): the sum starts at -(neg << (31 - sh)) and arithmetic is mod 2^32
  emission  zigzag(r) (encode.rs:3834-3863: (r << 1) ^ (r >> 31)) is w rotated left by one
"""
import numpy as np
import pytest

M32 = np.uint64(0xFFFFFFFF)


def fold(r):
    r = r.astype(np.int64)
    t = (r ^ (r >> 63)).astype(np.uint64) & np.uint64(0x7FFFFFFF)      # r ^ (r >> 31) on 32-bit values
    w = t | ((r.astype(np.uint64) & M32) & np.uint64(0x80000000))
    return t, w


def zigzag(r):
    r = r.astype(np.int64)
    return (((r << 1) ^ (r >> 31)).astype(np.uint64)) & M32


def samples(rng, n, kind):
    if kind == "small":
        return rng.integers(-300, 300, size=n, dtype=np.int64)
    if kind == "wide":
        return rng.integers(-(1 << 31), 1 << 31, size=n, dtype=np.int64)
    if kind == "negative":
        return -rng.integers(1, 1 << 20, size=n, dtype=np.int64)
    edge = np.array([0, -1, 1, -(1 << 31), (1 << 31) - 1, -(1 << 30), (1 << 30) - 1, (1 << 30), -(1 << 30) - 1], dtype=np.int64)
    return rng.choice(edge, size=n)


@pytest.mark.parametrize("kind", ["small", "wide", "negative", "edges"])
def test_rotation_is_the_zigzag(kind):
    rng = np.random.Generator(np.random.PCG64(31))
    r = samples(rng, 4096, kind)
    t, w = fold(r)
    assert (t < (1 << 31)).all() and (t == zigzag(r) >> np.uint64(1)).all()
    rot = ((w << np.uint64(1)) | (w >> np.uint64(31))) & M32             # v_alignbit_b32(w, w, 31)
    assert (rot == zigzag(r)).all()


@pytest.mark.parametrize("kind", ["small", "wide", "negative", "edges"])
def test_quotient_sum_with_the_sign_bits_taken_off(kind):
    rng = np.random.Generator(np.random.PCG64(32))
    for lane in range(50):
        r = samples(rng, 64, kind)                                       # a lane's 64 samples
        t, w = fold(r)
        neg = np.uint64((r < 0).sum())
        for sh in range(0, 31):
            want = np.uint64(int((t >> np.uint64(sh)).sum()) & 0xFFFFFFFF)
            q = np.uint64((0 - (int(neg) << (31 - sh))) & 0xFFFFFFFF)    # the sum's start
            for v in w:
                q = (q + (v >> np.uint64(sh))) & M32
            assert q == want, (kind, lane, sh)


def test_wide_test_masks_the_sign():
    """write_signed_counted(31) fails for r outside [-2^30, 2^30) (encode.rs:3857): t >= 2^30, looked at without bit 31"""
    r = np.array([(1 << 30) - 1, 1 << 30, -(1 << 30), -(1 << 30) - 1, -1, 0], dtype=np.int64)
    t, w = fold(r)
    assert ((w & np.uint64(0x7FFFFFFF)) >= (1 << 30)).tolist() == [False, True, False, True, False, False]
    assert ((t >= (1 << 30)) == ((r < -(1 << 30)) | (r >= (1 << 30)))).all()
