"""Pins the CPU oracle against every known-answer test the reference holds for the
encode hot path.  Values are the literals of the reference's own
unit tests / doc-tests; the file:line of each is given."""
import hashlib
import os

import numpy as np
import pytest

import _oracle as orc

REF = os.path.join(os.path.dirname(__file__), "golden", "refdata")

SINE25 = [0, 16, 31, 44, 54, 61, 64, 63, 58, 49, 38, 24, 8, -8, -24, -38, -49, -58, -63, -64,
          -61, -54, -44, -31, -16]


def test_residual_encoding_1():  # encode.rs:3216-3243
    rc, res = orc.encode_residuals([59, -30], 5, SINE25)
    assert rc == 0
    assert res.tolist() == [2, 2, 2, 3, 3, 3, 2, 2, 3, 0, 0, 0, -1, -1, -1, -3, -2, -2, -2, -1,
                            -1, 0, 0]


def test_residual_encoding_2():  # encode.rs:3245-3272
    samples = [64, 62, 56, 47, 34, 20, 4, -12, -27, -41, -52, -60, -63, -63, -60, -52, -41, -27,
               -12, 4, 20, 34, 47, 56, 62]
    rc, res = orc.encode_residuals([58, -29], 5, samples)
    assert rc == 0
    assert res.tolist() == [2, 2, 0, 1, -1, -1, -1, -2, -2, -2, -1, -3, -2, 0, -1, 1, 0, 2, 2, 2,
                            4, 2, 4]


def test_quantization():  # encode.rs:3404-3476
    rc, q, shift = orc.quantize([0.797774, -0.045362, -0.050136, -0.054254], 10)
    assert (rc, shift, q.tolist()) == (0, 9, [408, -23, -25, -28])
    rc, q, shift = orc.quantize([-0.054687, -0.953216, -0.027115, 0.033537], 10)
    assert (rc, shift, q.tolist()) == (0, 9, [-28, -488, -14, 17])
    rc, _, _ = orc.quantize([0.0, 0.0, 0.0, 0.0], 10)
    assert rc == 1  # ZeroLpCoefficients
    rc, q, shift = orc.quantize([-0.1, 0.1, 10000000.0, -0.2], 10)
    assert (rc, shift, q.tolist()) == (0, 0, [0, 0, 305, 0])
    rc, _, _ = orc.quantize([-0.1, 0.1, 100000000.0, -0.2], 10)
    assert rc == 2  # LpNegativeShiftError


def test_autocorrelation():  # encode.rs:3503-3527
    assert orc.autocorrelate([1.0], 1).tolist() == [1.0]
    assert orc.autocorrelate([1.0, 2.0, 3.0, 4.0, 5.0], 4).tolist() == [55.0, 40.0, 26.0, 14.0, 5.0]
    assert orc.autocorrelate([float(x) for x in SINE25], 4).tolist() == \
        [51408.0, 49792.0, 45304.0, 38466.0, 29914.0]


def _approx(a, b):
    assert abs(a - b) < 1.0e-6, f"{a} != {b}"


def test_lp_coefficients_1():  # encode.rs:3591-3621
    coeffs, errors = orc.lp_coefficients([55.0, 40.0, 26.0, 14.0, 5.0])
    assert len(coeffs) == 4
    for e, x in zip(errors, [25.909091, 25.540351, 25.316142, 25.241623]):
        _approx(e, x)
    expect = [[0.727273], [0.814035, -0.119298], [0.802858, -0.043028, -0.093694],
              [0.797774, -0.045362, -0.050136, -0.054254]]
    for c, x in zip(coeffs, expect):
        assert len(c) == len(x)
        for a, b in zip(c, x):
            _approx(a, b)


def test_lp_coefficients_2():  # encode.rs:3623-3653
    coeffs, errors = orc.lp_coefficients([51408.0, 49792.0, 45304.0, 38466.0, 29914.0])
    assert len(coeffs) == 4
    for e, x in zip(errors, [3181.201369, 495.815931, 495.161449, 494.604514]):
        _approx(e, x)
    expect = [[0.968565], [1.858456, -0.918772], [1.891837, -0.986293, 0.036332],
              [1.890618, -0.953216, -0.027115, 0.033537]]
    for c, x in zip(coeffs, expect):
        for a, b in zip(c, x):
            _approx(a, b)


def test_compute_best_order():  # encode.rs:3704-3745
    bits = orc.subframe_bits_by_order(16, 5, 20, [3181.201369, 495.815931, 495.161449, 494.604514])
    for a, b in zip(bits, [80.977565, 74.685594, 93.853530, 113.025628]):
        _approx(a, b)
    assert orc.compute_best_order(16, 5, 20, [3181.201369, 495.815931, 495.161449, 494.604514]) == 2
    bits = orc.subframe_bits_by_order(16, 10, 4096, [15000.0, 25000.0, 20000.0, 30000.0])
    for a, b in zip(bits, [1812.801817, 3346.934051, 2713.303385, 3935.492805]):
        _approx(a, b)
    # take_while(error > 0.0): a non-positive error ends the candidate list (encode.rs:3668)
    assert len(orc.subframe_bits_by_order(16, 10, 4096, [5.0, 0.0, 7.0])) == 1
    assert orc.compute_best_order(16, 10, 4096, [0.0, 1.0]) == 0  # NoBestLpcOrder


def test_verify_prediction():  # decode.rs:1754-1798 (decoder predict == inverse of the FIR)
    cases = [
        ([-75, 166, 121, -269, -75, -399, 1042], 9,
         [-796, -547, -285, -32, 199, 443, 670, -2, -23, 14, 6, 3, -4, 12, -2, 10],
         [-796, -547, -285, -32, 199, 443, 670, 875, 1046, 1208, 1343, 1454, 1541, 1616, 1663, 1701]),
        ([119, -255, 555, -836, 879, -1199, 1757], 10,
         [-21363, -21951, -22649, -24364, -27297, -26870, -30017, 3157],
         [-21363, -21951, -22649, -24364, -27297, -26870, -30017, -29718]),
        ([709, -2589, 4600, -4612, 1350, 4220, -9743, 12671, -12129, 8586, -3775, -645, 3904,
          -5543, 4373, 182, -6873, 13265, -15417, 11550], 12,
         [213238, 210830, 234493, 209515, 235139, 201836, 208151, 186277, 157720, 148176, 115037,
          104836, 60794, 54523, 412, 17943, -6025, -3713, 8373, 11764, 30094],
         [213238, 210830, 234493, 209515, 235139, 201836, 208151, 186277, 157720, 148176, 115037,
          104836, 60794, 54523, 412, 17943, -6025, -3713, 8373, 11764, 33931]),
    ]
    for coeffs, shift, buf, out in cases:
        qlp = list(reversed(coeffs))
        rc, res = orc.encode_residuals(qlp, shift, out)
        assert rc == 0
        assert res.tolist() == buf[len(qlp):]


def test_frame_bytes_doc_test():  # stream.rs:107-129 (header + CRC-8 0x64), 1645-1677 (CRC-16 0xd33b)
    opts = orc.options("default")
    rc, data, plan = orc.encode_frame(opts, 44100, 16, np.zeros((1, 20), dtype=np.int32))
    assert rc == 0
    assert data == bytes([0xff, 0xf8, 0x69, 0x08, 0x00, 0x13, 0x64, 0x00, 0x00, 0x00, 0xd3, 0x3b])
    assert plan.sub[0].type == orc.SUB_CONSTANT
    # the same 12 bytes are the first frame of the libFLAC-made fixture
    blob = open(os.path.join(REF, "all-frames.flac"), "rb").read()
    assert data in blob


def test_crc_and_md5_primitives():
    assert orc.crc8(bytes([0xff, 0xf8, 0x69, 0x08, 0x00, 0x13])) == 0x64
    assert orc.crc16(bytes([0xff, 0xf8, 0x69, 0x08, 0x00, 0x13, 0x64, 0, 0, 0])) == 0xd33b
    for blob in (b"", b"a", b"abc", bytes(range(256)) * 37):
        assert orc.md5(blob) == hashlib.md5(blob).digest()


def test_streaminfo_bytes_doc_test():  # metadata/mod.rs:1599-1630 layout, via a real stream
    opts = orc.options("default", padding=-1, seektable_mode=0)
    pcm = np.zeros(20, dtype=np.int32)
    rc, data, st = orc.encode_stream(opts, 44100, 16, 1, pcm, total_known=True)
    assert rc == 0
    assert data[:4] == b"fLaC"
    assert data[4] == 0x80 and data[5:8] == b"\x00\x00\x22"  # last-block flag, STREAMINFO, 34 bytes
    si = data[8:42]
    assert si[0:2] == (4096).to_bytes(2, "big") and si[2:4] == (4096).to_bytes(2, "big")
    assert si[4:7] == (12).to_bytes(3, "big") and si[7:10] == (12).to_bytes(3, "big")
    # 20 bits rate, 3 bits channels-1, 5 bits bps-1, 36 bits total samples
    packed = int.from_bytes(si[10:18], "big")
    assert packed >> 44 == 44100 and (packed >> 41) & 7 == 0 and (packed >> 36) & 31 == 15
    assert packed & ((1 << 36) - 1) == 20
    assert si[18:34] == hashlib.md5(bytes(40)).digest()


@pytest.mark.parametrize("name,md5hex,frames", [
    ("sine.flac", "831671b807f97051301e01d68b5c54b3", 49),
    ("all-frames.flac", None, 4),
    ("seektable.flac", None, 4),
    ("comment.flac", None, 4),
])
def test_decoder_on_reference_fixtures(name, md5hex, frames):
    """The decoder restatement (the round-trip verifier of every later test) is pinned by
    libFLAC-produced streams the reference's tests hold: CRC-8/16 and STREAMINFO MD5 verify."""
    blob = open(os.path.join(REF, name), "rb").read()
    rc, pcm, info = orc.decode_stream(blob)
    assert rc == 0
    assert info.frames == frames
    assert info.md5_ok == 1
    assert pcm.size == info.total_samples * info.channels
    if md5hex:
        assert bytes(info.md5).hex() == md5hex
        assert (info.sample_rate, info.channels, info.bps) == (44100, 2, 16)
        assert info.n_seekpoints == 5


def test_frame_number_coding():  # stream.rs:1264-1356: UTF-8-like coding, checked via decode
    opts = orc.options("fast", padding=-1, seektable_mode=0)
    pcm = np.arange(16, dtype=np.int32).reshape(1, 16)
    expect_len = {0: 1, 0x7F: 1, 0x80: 2, 0x7FF: 2, 0x800: 3, 0xFFFF: 3, 0x10000: 4,
                  0x1FFFFF: 4, 0x200000: 5, 0x3FFFFFF: 5, 0x4000000: 6, 0x7FFFFFFF: 6,
                  0x80000000: 7, 0xFFFFFFFFF: 7}
    for fn, nbytes in expect_len.items():
        rc, data, _ = orc.encode_frame(opts, 44100, 16, pcm, frame_number=fn)
        assert rc == 0
        # header = 4 fixed bytes + frame number + 1 byte (block size 16 -> 8-bit field) + CRC-8
        hdr = data[: 4 + nbytes + 1 + 1]
        assert orc.crc8(hdr[:-1]) == hdr[-1]
        first = data[4]
        if nbytes == 1:
            assert first == fn
        else:
            assert bin(first)[2:].zfill(8).startswith("1" * nbytes + "0")
            v = first & ((1 << (7 - nbytes)) - 1)
            for b in data[5:4 + nbytes]:
                assert b >> 6 == 0b10
                v = (v << 6) | (b & 0x3F)
            assert v == fn
