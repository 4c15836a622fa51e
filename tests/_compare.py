"""Helpers comparing the GPU path's decision records with the oracle's."""
import numpy as np

import _oracle as orc

SRC_MID, SRC_SIDE = 8, 9


def orc_options_for(block_size, max_po, max_lpc, mid_side, exhaustive, window_kind=2,
                    window_param=0.5):
    return orc.options("default", block_size=block_size, max_partition_order=max_po,
                       max_lpc_order=max_lpc, mid_side=int(mid_side), exhaustive=int(exhaustive),
                       window_kind=window_kind, window_param=window_param)


def candidate(planar, source):
    """planar [C][n] int32 -> candidate samples of `source` (before wasted-bit removal)."""
    if source == SRC_MID:
        return ((planar[0].astype(np.int64) + planar[1].astype(np.int64)) >> 1).astype(np.int32)
    if source == SRC_SIDE:
        return (planar[0].astype(np.int64) - planar[1].astype(np.int64)).astype(np.int32)
    return planar[source]


def expected_row(sub, planar, n):
    """The residual row the ABI promises for one subframe (warm-up + residuals / verbatim)."""
    cand = candidate(planar, sub.source)
    shifted = cand >> sub.wasted
    if sub.type in (orc.SUB_FIXED, orc.SUB_LPC):
        o = orc.SubframePlan()
        o.type, o.wasted, o.order, o.shift = sub.type, sub.wasted, sub.order, sub.shift
        for i in range(32):
            o.coeffs[i] = sub.coeffs[i]
        res = orc.subframe_residuals(o, cand)
        return np.concatenate([shifted[: sub.order], res]).astype(np.int32)
    if sub.type == orc.SUB_VERBATIM:
        return shifted.astype(np.int32)
    return shifted[:1].astype(np.int32)


def compare_frame(gpu_fp, gpu_subs, gpu_res_rows, oplan, planar, n, where=""):
    """Raises AssertionError with a precise message on the first difference."""
    assert gpu_fp.assignment == oplan.assignment, f"{where} assignment {gpu_fp.assignment} != {oplan.assignment}"
    assert gpu_fp.block_size == n, f"{where} block size"
    nch = planar.shape[0]
    body = 0
    for c in range(nch):
        g, o = gpu_subs[c], oplan.sub[c]
        w = f"{where} ch{c}"
        assert g.source == oplan.source[c], f"{w} source {g.source} != {oplan.source[c]}"
        for f in ("type", "wasted", "bps", "bits"):
            assert getattr(g, f) == getattr(o, f), f"{w} {f}: gpu {getattr(g, f)} != oracle {getattr(o, f)} (type gpu {g.type} orc {o.type}, order gpu {g.order} orc {o.order})"
        if o.type in (orc.SUB_FIXED, orc.SUB_LPC):
            assert g.order == o.order, f"{w} order {g.order} != {o.order}"
            for f in ("coding_method", "partition_order", "n_partitions"):
                assert getattr(g, f) == getattr(o, f), f"{w} {f}: {getattr(g, f)} != {getattr(o, f)}"
            npart = o.n_partitions
            assert list(g.rice[:npart]) == list(o.rice[:npart]), f"{w} rice {list(g.rice[:npart])} != {list(o.rice[:npart])}"
            assert list(g.escape_bits[:npart]) == list(o.escape_bits[:npart]), f"{w} escape"
            # partition lengths: first chunk short, the rest part_len
            lens = [o.part_len[i] for i in range(npart)]
            nres = n - o.order
            glens = [nres - (npart - 1) * g.part_len] + [g.part_len] * (npart - 1) if npart > 1 else [nres]
            assert glens == lens, f"{w} partition lengths {glens} != {lens}"
        if o.type == orc.SUB_LPC:
            assert (g.precision, g.shift) == (o.precision, o.shift), f"{w} precision/shift"
            assert list(g.coeffs[: o.order]) == list(o.coeffs[: o.order]), f"{w} coeffs"
        row = expected_row(g, planar, n)
        got = gpu_res_rows[c][: len(row)]
        assert np.array_equal(got, row), f"{w} residual row differs at {np.flatnonzero(got != row)[:5]}"
        body += o.bits
    assert gpu_fp.body_bits == body, f"{where} body bits"


def planar_frames(interleaved, channels, block_size):
    """Split interleaved PCM into a list of planar [C][n] frames (last may be short)."""
    pcm = np.asarray(interleaved, dtype=np.int32)
    total = pcm.size // channels
    pcm = pcm[: total * channels].reshape(total, channels)
    out = []
    for s in range(0, total, block_size):
        out.append(np.ascontiguousarray(pcm[s:s + block_size].T))
    return out


def frames_the_reference_cannot_decode(subs, n_frames, channels, block_size, last_len):
    """Frames holding a subframe whose residual chunks do not tile the block.  best_partitions (encode.rs:3865-3896)
    tries every partition count up to 2^min(tz(n), max) and cuts the residuals with rchunks(n / count); when n / count
    is smaller than the predictor order the chunks are FEWER than `count`, and if their number happens to be a power
    of two the candidate survives the filter of :3881 and is written with partition order log2(chunks) (:3903) -- a
    partition length that is not n >> order.  The reference's own decoder (decode.rs:1812-1820: rchunks_mut(n / 2^order),
    `partitions.len() != partition_count` -> InvalidPartitionOrder) rejects such a frame.  It takes a frame of fewer
    than 2 x order samples (a very short last frame); the encoder here reproduces the reference's bytes, and the device
    decoder, like the reference's, flags the frame.  Returns the set of such frame indices."""
    bad = set()
    for f in range(n_frames):
        n = block_size if f + 1 < n_frames else last_len
        for ch in range(channels):
            s = subs[f * channels + ch]
            if s.type in (2, 3) and (s.part_len << s.partition_order) != n:
                bad.add(f)
    return bad
