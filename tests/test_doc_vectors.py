"""The reference's literal doc-test bytes as pins:
  * stream.rs:2081-2100 / 2130-2157 / 2190-2223 / 2266-2311 (CONSTANT, VERBATIM, FIXED-4, LPC-1
    subframes), 2719-2752 / 2902-2925 (residual block / partition), metadata/mod.rs:1599-1630 (STREAMINFO)
CPU part: the oracle's decoder must parse them to the values the reference asserts; the product's
host packer (flacenc_pack_frames) and STREAMINFO serialiser must produce exactly these bytes.
GPU part (-m gpu): the device packer (flacgpu_pack_plans -> k_emit + k_pack + k_crc) must too --
HIP against reference-held bytes with no oracle in between."""
import ctypes as C

import numpy as np
import pytest

import _docvectors as dv
import _oracle as orc


def minimal_stream(frame):
    """fLaC + STREAMINFO (20 samples, 16-bit mono, 44.1 kHz, MD5 unknown = zeros) + one frame"""
    si = bytearray()
    si += (20).to_bytes(2, "big") * 2
    si += len(frame).to_bytes(3, "big") * 2
    packed = (dv.RATE << 44) | (0 << 41) | ((dv.BPS - 1) << 36) | dv.N
    si += packed.to_bytes(8, "big")
    si += bytes(16)
    return b"fLaC" + bytes([0x80, 0, 0, 34]) + bytes(si) + frame


@pytest.mark.parametrize("name", sorted(dv.VECTORS))
def test_oracle_decoder_parses_reference_bytes(name):
    v = dv.VECTORS[name]
    frame = dv.frame_bytes(v, orc.crc16)
    rc, out, info = orc.decode_stream(minimal_stream(frame))
    assert rc == 0, f"{name}: decoder rejected the reference's bytes"
    assert info.frames == 1
    assert list(out) == v["samples"], f"{name}: decoded samples differ from what the reference asserts"


def test_residual_vectors_are_the_tail_of_the_lpc_vector():
    """stream.rs:2719-2752 is the residual block of the LPC example; 2902-2925 its partition shifted by the
    6 bits of coding method + partition order -- the same bits the decoder test above consumed."""
    lpc = dv.VECTORS["lpc1"]["bytes"]
    blk = dv.RESIDUAL_BLOCK
    bits = lambda b: "".join(f"{x:08b}" for x in b)
    tail = bits(lpc)[8 + 16 + 4 + 5 + 12:]
    assert tail.rstrip("0") == bits(blk).rstrip("0")
    assert bits(blk)[6:].rstrip("0") == bits(dv.RESIDUAL_PARTITION).rstrip("0")


def host_pack(v):
    from flac_codec_amd import _lib
    from flac_codec_amd._lib import FramePlan, SubframePlan

    L = _lib.lib()
    plan = (FramePlan * 1)()
    plan[0].assignment, plan[0].channels, plan[0].block_size, plan[0].body_bits = 0, 1, dv.N, v["bits"]
    subs = (SubframePlan * 1)()
    dv.fill_plan(subs[0], v)
    rows = np.zeros(dv.N, dtype=np.int32)
    r = dv.residual_row(v)
    rows[: len(r)] = r
    off = (C.c_uint64 * 2)()
    buf = np.zeros(256, dtype=np.uint8)
    rc = L.flacenc_pack_frames(dv.RATE, dv.BPS, 1, 0, 1, dv.N, C.cast(plan, C.c_void_p), C.cast(subs, C.c_void_p),
                               rows.ctypes.data_as(C.POINTER(C.c_int32)), 1, buf.ctypes.data, buf.size, off)
    assert rc == 0
    return buf[: off[1]].tobytes()


@pytest.mark.parametrize("name", sorted(dv.VECTORS))
def test_host_packer_reproduces_reference_bytes(name):
    v = dv.VECTORS[name]
    assert host_pack(v) == dv.frame_bytes(v, orc.crc16)


def test_streaminfo_serialiser_reproduces_reference_bytes():  # metadata/mod.rs:1599-1630
    from flac_codec_amd import _lib

    L = _lib.lib()
    f = dv.STREAMINFO_FIELDS
    out = (C.c_uint8 * 34)()
    md5 = (C.c_uint8 * 16)(*f["md5"])
    L.flacenc_streaminfo_bytes.argtypes = [C.c_uint32] * 7 + [C.c_uint64, C.c_void_p, C.c_void_p]
    rc = L.flacenc_streaminfo_bytes(f["min_block"], f["max_block"], f["min_frame"], f["max_frame"], f["sample_rate"],
                                    f["channels"], f["bits_per_sample"], f["total_samples"], md5, out)
    assert rc == 0 and bytes(out) == dv.STREAMINFO_BYTES


def test_oracle_streaminfo_layout_against_reference_bytes():
    """The oracle's stream writer on 80 samples: every field but the content-dependent ones must sit where
    the reference's literal puts it (same block sizes, rate / channels / bps, total 80)."""
    opts = orc.options("default", padding=-1, seektable_mode=0)
    rc, data, _ = orc.encode_stream(opts, 44100, 16, 1, np.zeros(80, dtype=np.int32), total_known=True)
    assert rc == 0
    si = data[8:42]
    assert si[:4] == dv.STREAMINFO_BYTES[:4] and si[10:18] == dv.STREAMINFO_BYTES[10:18]
    assert si[4:10] == dv.STREAMINFO_BYTES[4:10]     # an 80-sample all-zero mono frame is 12 bytes as well


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(dv.VECTORS))
def test_device_packer_reproduces_reference_bytes(name):
    from flac_codec_amd import _lib
    from flac_codec_amd._lib import FramePlan, SubframePlan
    from flac_codec_amd.gpu import GpuAnalyzer

    v = dv.VECTORS[name]
    L = _lib.lib()
    an = GpuAnalyzer(dv.N, 0, 8, False, False, 0, 0.0, dv.BPS, 1, max_frames=3)
    # three frames with the same decisions (frame numbers 0, 1, 2): the first must be the literal frame
    plans = (FramePlan * 3)()
    subs = (SubframePlan * 3)()
    for f in range(3):
        plans[f].assignment, plans[f].channels, plans[f].block_size, plans[f].body_bits = 0, 1, dv.N, v["bits"]
        dv.fill_plan(subs[f], v)
    pcm = np.tile(dv.as_array(v["samples"]), 3)
    L.flacgpu_pack_plans.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.POINTER(FramePlan),
                                     C.POINTER(SubframePlan), C.c_uint64, C.c_uint32]
    rc = L.flacgpu_pack_plans(an._h, pcm.ctypes.data_as(C.POINTER(C.c_int32)), 3, dv.N, plans, subs, 0, dv.RATE)
    assert rc == 0, L.flacgpu_last_error()
    data, off = an.fetch_frames(3)
    want = dv.frame_bytes(v, orc.crc16)
    assert data[off[0]:off[1]] == want, f"{name}: device packer differs from the reference's bytes"
    # frames 1 and 2 differ only in the frame number byte, the CRC-8 and the CRC-16
    for f in (1, 2):
        fr = data[off[f]:off[f + 1]]
        assert fr[7:-2] == want[7:-2] and fr[4] == f
        assert orc.crc8(fr[:6]) == fr[6] and orc.crc16(fr[:-2]) == (fr[-2] << 8 | fr[-1])
    an.close()
