"""Device-side decoder / verifier (k_decode + k_crc<VERIFY>): decodes the frames the GPU just
packed, without leaving HBM, and must give back the input PCM; corrupting one byte of a frame
must be detected (the reference's tests/corruption.rs does the same with random bit flips)."""
import ctypes as C

import numpy as np
import pytest

from _pcm import read_raw, synth_fast

pytestmark = pytest.mark.gpu


def encode(pcm, ch, bps, block=4096, lpc=12, po=6, rate=48000, mid_side=True, exhaustive=True):
    from flac_codec_amd.gpu import GpuAnalyzer

    total = pcm.size // ch
    n_frames = (total + block - 1) // block
    last = total - (n_frames - 1) * block
    an = GpuAnalyzer(block, po, lpc, mid_side, exhaustive, 2, 0.5, bps, ch, max_frames=n_frames)
    data, off = an.encode_frames(pcm[: total * ch], n_frames, last, 0, rate)
    return an, n_frames, last, data, off


@pytest.mark.parametrize("ch,bps,block,lpc", [(2, 24, 4096, 12), (2, 16, 4096, 0), (1, 16, 1152, 8),
                                              (8, 24, 4096, 12), (2, 24, 4096, 32), (3, 20, 576, 12),
                                              (2, 32, 4096, 12), (2, 8, 256, 4)])
def test_decode_roundtrip_on_device(ch, bps, block, lpc):
    pcm = synth_fast(500 + ch + bps, ch, min(bps, 24), block * 9 + block // 3)
    if bps == 32:
        pcm = (pcm.astype(np.int64) << 8).astype(np.int32)
    if bps == 8:
        pcm = (pcm >> 16).astype(np.int32)
    an, n_frames, last, data, off = encode(pcm, ch, bps, block=block, lpc=lpc)
    res, ms = an.verify_device(48000)
    assert (res.frames, res.bad_structure, res.bad_crc16) == (n_frames, 0, 0)
    assert res.compared_pcm == 1 and res.frames_pcm_differs == 0 and res.samples_differ == 0
    out = an.fetch_decoded(n_frames, last)
    total = pcm.size // ch
    assert np.array_equal(out, pcm[: total * ch])
    an.close()


def test_special_subframes_decode():
    rng = np.random.Generator(np.random.PCG64(7))
    cases = [np.zeros(4096 * 4, dtype=np.int32),                                        # CONSTANT
             rng.integers(-(1 << 23), 1 << 23, size=4096 * 4, dtype=np.int64).astype(np.int32),  # VERBATIM
             (synth_fast(8, 2, 16, 4096 * 2) << 5).astype(np.int32)]                     # wasted bits
    x = np.zeros(4096 * 4, dtype=np.int32)
    x[::997] = 1 << 22                                                                  # escapes / long unary
    cases.append(x)
    for pcm in cases:
        an, n_frames, last, _, _ = encode(pcm, 2, 24)
        res, _ = an.verify_device(48000)
        assert (res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (0, 0, 0)
        assert np.array_equal(an.fetch_decoded(n_frames, last), pcm)
        an.close()
    pcm = read_raw("wasted-bits.raw", 16)
    an, n_frames, last, _, _ = encode(pcm, 1, 16)
    res, _ = an.verify_device(44100)
    assert (res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (0, 0, 0)
    an.close()


def test_corruption_is_detected():  # tests/corruption.rs:29-43 shape
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    pcm = synth_fast(510, 2, 16, 4096 * 8)
    an, n_frames, last, data, off = encode(pcm, 2, 16)
    dbuf = an.device_buffer(4)
    rng = np.random.Generator(np.random.PCG64(3))
    detected = 0
    trials = 12
    for _ in range(trials):
        f = int(rng.integers(0, n_frames))
        pos = int(rng.integers(off[f] + 8, off[f + 1] - 2))
        orig = bytes([data[pos]])
        flipped = bytes([data[pos] ^ (1 << int(rng.integers(0, 8)))])
        assert hip.hipMemcpy(C.c_void_p(dbuf + pos), flipped, 1, 1) == 0
        res, _ = an.verify_device(48000)
        if res.bad_crc16 >= 1 and (res.bad_structure + res.frames_pcm_differs) >= 1:
            detected += 1
        assert res.bad_crc16 == 1            # CRC-16 catches every single-bit error
        assert hip.hipMemcpy(C.c_void_p(dbuf + pos), orig, 1, 1) == 0
    res, _ = an.verify_device(48000)
    assert (res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (0, 0, 0)
    assert detected >= trials - 2   # the decoder itself also notices almost every flip
    an.close()


@pytest.mark.parametrize("ch,bps,block,lpc", [(2, 16, 4096, 12), (8, 24, 576, 12), (1, 16, 1152, 32)])
def test_heavy_corruption_never_hangs_or_spills_over(ch, bps, block, lpc):
    """Several flipped / zeroed / saturated bytes per trial: the decoder must come back, flag only
    frames that were touched, and the CRC-16 must catch at least one of them every time."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    frames = 48
    pcm = synth_fast(600 + ch + bps, ch, bps, block * frames)
    an, n_frames, last, data, off = encode(pcm, ch, bps, block=block, lpc=lpc)
    data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    off = np.asarray(off)
    dbuf = an.device_buffer(4)
    rng = np.random.Generator(np.random.PCG64(11 + ch))
    for _ in range(25):
        bad = data.copy()
        for _ in range(int(rng.integers(1, 30))):
            pos = int(rng.integers(0, len(bad)))
            kind = int(rng.integers(0, 3))
            bad[pos] = bad[pos] ^ (1 << int(rng.integers(0, 8))) if kind == 0 else (0 if kind == 1 else 0xFF)
        touched = {int(np.searchsorted(off, p, side="right") - 1) for p in np.nonzero(bad != data)[0]}
        assert hip.hipMemcpy(C.c_void_p(dbuf), bad.ctypes.data_as(C.c_void_p), len(bad), 1) == 0
        res, _ = an.verify_device(48000)
        assert res.bad_crc16 <= len(touched) and res.bad_structure + res.frames_pcm_differs <= len(touched)
        assert not touched or res.bad_crc16 >= 1
    assert hip.hipMemcpy(C.c_void_p(dbuf), data.ctypes.data_as(C.c_void_p), len(data), 1) == 0
    res, _ = an.verify_device(48000)
    assert (res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (0, 0, 0)
    an.close()
