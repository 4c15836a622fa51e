"""Byte vectors the reference holds in its doc-tests (literal data, /root/reference/src/stream.rs and
src/metadata/mod.rs), with the values the reference asserts they parse to.  Every vector is one
subframe of a 20-sample, 16-bit mono block (the frame header 'ff f8 69 08 00 13 64' of
stream.rs:107-129 / 1645-1677 in front of it makes a whole subset frame at 44.1 kHz)."""
import numpy as np

FRAME_HEADER = bytes([0xff, 0xf8, 0x69, 0x08, 0x00, 0x13, 0x64])   # stream.rs:107-129
N, BPS, RATE = 20, 16, 44100

SUB_CONSTANT, SUB_VERBATIM, SUB_FIXED, SUB_LPC = 0, 1, 2, 3


def lpc_samples(coeffs, shift, warm, residuals):
    x = list(warm)
    for r in residuals:
        i = len(x)
        pred = sum(c * x[i - 1 - j] for j, c in enumerate(coeffs)) >> shift
        x.append(r + pred)
    return x


VECTORS = {
    # stream.rs:2081-2100
    "constant": dict(bytes=bytes([0b0_000000_0, 0x00, 0x00]), bits=8 + 16, type=SUB_CONSTANT, order=0,
                     samples=[0] * 20, residuals=[], warm=[0]),
    # stream.rs:2130-2157
    "verbatim": dict(bytes=bytes([0b0_000001_0] + [b for v in range(20) for b in (0x00, v)]), bits=8 + 20 * 16,
                     type=SUB_VERBATIM, order=0, samples=list(range(20)), residuals=[], warm=[]),
    # stream.rs:2190-2223: FIXED order 4, warm-up 0 1 2 3, sixteen zero residuals in one Rice-0 partition
    "fixed4": dict(bytes=bytes([0b0_001100_0, 0x00, 0x00, 0x00, 0x01, 0x00, 0x02, 0x00, 0x03, 0x00, 0x3f, 0xff, 0xc0]),
                   bits=8 + 4 * 16 + 2 + 4 + 4 + 16, type=SUB_FIXED, order=4, samples=list(range(20)),
                   residuals=[0] * 16, warm=[0, 1, 2, 3], rice=0, method=0),
    # stream.rs:2266-2311: LPC order 1, precision 12, shift 11, coefficient 1989, residuals 1 2 2 2 ... (Rice 1);
    # the residual block alone is stream.rs:2719-2752, its partition stream.rs:2902-2925
    "lpc1": dict(bytes=bytes([0b0_100000_0, 0x00, 0x00, 0b1011_0101, 0b1_0111110, 0b00101_000,
                              0x02, 0x88, 0x88, 0x88, 0x88, 0x88, 0x88, 0x88, 0x88, 0x88, 0x80]),
                 bits=8 + 16 + 4 + 5 + 12 + 2 + 4 + 4 + (3 + 18 * 4), type=SUB_LPC, order=1, precision=12,
                 shift=11, coeffs=[1989], warm=[0], residuals=[1] + [2] * 18, rice=1, method=0,
                 samples=lpc_samples([1989], 11, [0], [1] + [2] * 18)),
}
# stream.rs:2719-2752 (whole residual block of the LPC example) and 2902-2925 (its single partition)
RESIDUAL_BLOCK = bytes([0b00_0000_00, 0b01010001] + [0b00010001] * 8 + [0b00010000, 0b00000000])
RESIDUAL_PARTITION = bytes([0b0001_01_0_0] + [0b01_0_001_0_0] * 9)

# metadata/mod.rs:1599-1630
STREAMINFO_BYTES = bytes([0x10, 0x00, 0x10, 0x00, 0x00, 0x00, 0x0c, 0x00, 0x00, 0x0c,
                          0b00001010, 0b11000100, 0b0100_000_0, 0b1111_0000,
                          0b00000000, 0b00000000, 0b00000000, 0b01010000,
                          0xf5, 0x3f, 0x86, 0x87, 0x6d, 0xcd, 0x77, 0x83,
                          0x22, 0x5c, 0x93, 0xba, 0x8a, 0x93, 0x8c, 0x7d])
STREAMINFO_FIELDS = dict(min_block=4096, max_block=4096, min_frame=12, max_frame=12, sample_rate=44100,
                         channels=1, bits_per_sample=16, total_samples=80, md5=STREAMINFO_BYTES[18:])


def residual_row(v):
    """The row the packers take for a subframe (include/flacenc_gpu.h: warm-up then residuals /
    the samples / the constant)."""
    if v["type"] == SUB_VERBATIM:
        return v["samples"]
    if v["type"] == SUB_CONSTANT:
        return [v["samples"][0]]
    return v["warm"] + v["residuals"]


def fill_plan(sub, v):
    """Fill a flacgpu_subframe_plan (ctypes) with the decisions the vector encodes."""
    sub.type, sub.wasted, sub.bps, sub.order = v["type"], 0, BPS, v["order"]
    sub.precision, sub.shift = v.get("precision", 0), v.get("shift", 0)
    sub.coding_method, sub.partition_order, sub.source = v.get("method", 0), 0, 0
    coded = v["type"] in (SUB_FIXED, SUB_LPC)
    sub.n_partitions = 1 if coded else 0
    sub.part_len = N if coded else 0
    sub.bits = v["bits"]
    for j, c in enumerate(v.get("coeffs", [])):
        sub.coeffs[j] = c
    if coded:
        sub.rice[0] = v["rice"]
        sub.escape_bits[0] = 0


def frame_bytes(v, crc16):
    """header + subframe bytes (zero padded to a byte, as the vector is) + CRC-16 of all of it"""
    body = FRAME_HEADER + v["bytes"]
    c = crc16(body)
    return body + bytes([c >> 8, c & 0xFF])


def as_array(xs):
    return np.asarray(xs, dtype=np.int32)
