"""Device-side decoder / verifier (k_decode + k_crc<VERIFY>): decodes the frames the GPU just
packed, without leaving HBM, and must give back the input PCM; corrupting one byte of a frame
must be detected (the reference's tests/corruption.rs does the same with random bit flips)."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as orc
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


# ---- stand-alone decoder (flacgpu_decode_stream): foreign streams, no encoder plan ---------------
REFDATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata")


@pytest.mark.parametrize("name,md5hex", [("sine.flac", "831671b807f97051301e01d68b5c54b3"),
                                          ("all-frames.flac", None), ("seektable.flac", None),
                                          ("comment.flac", None)])
def test_standalone_decoder_on_reference_fixtures(name, md5hex):
    """The libFLAC-made files the reference's own tests hold decode on the GPU to the MD5 in their
    STREAMINFO (sine.flac: the MD5 tests/format.rs checks) and to the oracle decoder's samples."""
    from flac_codec_amd.gpu import decode_stream

    blob = open(os.path.join(REFDATA, name), "rb").read()
    pcm, info = decode_stream(blob)
    assert info.bad_frames == 0 and info.bad_crc16 == 0
    assert info.md5_status == 1, f"{name}: decoded MD5 {bytes(info.decoded_md5).hex()} != STREAMINFO {bytes(info.md5).hex()}"
    if md5hex:
        assert bytes(info.decoded_md5).hex() == md5hex
    rc, ref, oinfo = orc.decode_stream(blob)
    assert rc == 0 and oinfo.md5_ok == 1
    assert info.frames == oinfo.frames and info.decoded_samples == (info.total_samples or info.decoded_samples)
    assert np.array_equal(pcm, ref)


def test_standalone_decoder_on_own_streams_and_corruption():
    from flac_codec_amd.encode import FlacSampleWriter, Options
    from flac_codec_amd.gpu import decode_stream

    for (ch, bps, n, opts) in [(2, 24, 4096 * 9 + 123, Options.best()), (2, 16, 1152 * 20 + 7, Options.fast()),
                               (6, 20, 4096 * 3, Options.default()), (1, 8, 5000, Options.best().block_size(1000)),
                               (2, 24, 4096 * 4, Options.best().max_lpc_order(32))]:
        pcm = synth_fast(4000 + ch + bps, ch, bps, n)
        w = FlacSampleWriter(None, opts, 48000, bps, ch, pcm.size)
        w.write(pcm)
        w.finalize()
        data = w.getvalue()
        w.close()
        out, info = decode_stream(data)
        assert info.md5_status == 1 and info.bad_frames == 0 and np.array_equal(out, pcm)
        # a flipped bit inside a frame is caught (CRC-16 chain breaks -> the scan stops there)
        bad = bytearray(data)
        bad[len(bad) - len(bad) // 8] ^= 0x10   # inside the frames (the first half may be PADDING)
        _, binfo = decode_stream(bytes(bad))
        assert binfo.md5_status != 1 or binfo.bad_frames or binfo.frames != info.frames
