"""The batch plan of the coalescing front end (csrc/host/coalesce.cpp plan_batches, through the flacenc_coalesce_plan hook): which
run of whole blocks of which stream goes into which batch.  CPU-only.  Invariants, over seeded random stream populations and the
shapes the soak found trouble with: every block of every stream exactly once, a stream's segments in stream order and in
non-decreasing batches, no batch above the cap, and a stream of up to 32 blocks in ONE segment -- its MD5 chain is one run hashed by
one task (r06: a remainder batch at the head of a short stream used to cut it in two, and two tasks hashed one chain)."""
import ctypes as C

import numpy as np
import pytest

from flac_codec_amd import _lib

SOLO = 32


def plan(whole, batch_frames=0, samples_per_block=8192):
    L = _lib.lib()
    L.flacenc_coalesce_plan.restype = C.c_size_t
    L.flacenc_coalesce_plan.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t,
                                        C.POINTER(C.c_uint32)]
    w = (C.c_uint64 * max(1, len(whole)))(*whole)
    cap = 4 * sum(whole) // 1 + 4 * len(whole) + 16
    cap = min(cap, 4_000_000)
    st, fi, nn, ba = (C.c_uint32 * cap)(), (C.c_uint64 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    bc = C.c_uint32(0)
    n = L.flacenc_coalesce_plan(w, len(whole), batch_frames, samples_per_block, st, fi, nn, ba, cap, C.byref(bc))
    assert n <= cap
    return [(st[i], fi[i], nn[i], ba[i]) for i in range(n)], bc.value


def check(whole, batch_frames=0, samples_per_block=8192):
    segs, cap = plan(whole, batch_frames, samples_per_block)
    nxt = [0] * len(whole)
    last_batch = [-1] * len(whole)
    nseg = [0] * len(whole)
    frames = {}
    prev_batch = 0
    for s, first, n, b in segs:
        assert n > 0 and first == nxt[s], f"stream {s}: segment starts at {first}, expected {nxt[s]}"
        assert b >= last_batch[s] and b >= prev_batch, "segments in non-decreasing batches"
        nxt[s] += n
        last_batch[s] = b
        prev_batch = b
        nseg[s] += 1
        frames[b] = frames.get(b, 0) + n
    assert nxt == list(whole), "every block exactly once"
    assert all(v <= cap for v in frames.values()), (max(frames.values()), cap)
    assert sorted(frames) == list(range(len(frames))), "no empty batch"
    for k, wk in enumerate(whole):
        if 0 < wk <= SOLO:
            assert nseg[k] == 1, f"a stream of {wk} blocks cut into {nseg[k]} segments"
    return segs, cap


def test_the_shape_that_cut_a_short_stream():
    check([8] * 7 + [7, 4], batch_frames=64)                       # plan 64 + 3: the 4-block stream meets the 3-frame batch
    segs, cap = check([5, 3, 8, 1, 2, 8, 8, 8, 8, 8, 6], batch_frames=64)
    assert cap == 64


@pytest.mark.parametrize("seed", range(40))
def test_random_populations(seed):
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    n = int(rng.choice([1, 2, 3, 17, 64, 97, 300, 1500]))
    kind = int(rng.integers(4))
    if kind == 0:
        whole = rng.integers(0, 9, n)
    elif kind == 1:
        whole = rng.integers(0, 40, n)
    elif kind == 2:
        whole = rng.choice([0, 1, 8, 31, 32, 33, 64, 512, 2000], n)
    else:
        whole = rng.integers(0, 700, n)
    if whole.sum() == 0:
        whole[0] = 3
    check([int(v) for v in whole], batch_frames=int(rng.choice([0, 4, 64, 100, 1024, 8192])),
          samples_per_block=int(rng.choice([4096, 8192, 32768, 1152 * 2])))


def test_sweep_shapes_and_chain_bound_head():
    for n, f in ((8192, 1), (1024, 8), (256, 32), (64, 128), (32, 256), (16, 512), (64, 512), (1, 5000)):
        segs, cap = check([f] * n)
        if f > SOLO and n < 70 and n * f >= 4 * cap:               # chain-bound: a small batch in front starts all chains together
            first = [s for s in segs if s[3] == 0]
            assert len(first) == n and sum(s[2] for s in first) <= cap // 4
