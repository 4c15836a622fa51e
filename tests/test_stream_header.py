"""flacenc_stream_header: the bytes in front of the first frame, rebuilt from the frame sizes alone
(what the owner of a stream encoded in shards on several GPUs needs), against the
oracle's whole-stream output (/root/reference/src/encode.rs:1882-1980, 1999-2003, 2024-2110).
CPU only: no analysis lane is created."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _oracle as orc
from _pcm import synth_fast


def frame_sizes(data, first):
    """Frame boundaries of a FLAC stream by the CRC-16 chain (header + CRC-8 valid, the two bytes in
    front are the CRC-16 of the frame)."""
    sizes, p = [], first
    while p < len(data):
        q = p + 8
        while True:
            nxt = data.find(b"\xff\xf8", q)
            if nxt < 0:
                nxt = len(data)
            if orc.crc16(data[p:nxt - 2]) == int.from_bytes(data[nxt - 2:nxt], "big"):
                break
            q = nxt + 1
        sizes.append(nxt - p)
        p = nxt
    return sizes


@pytest.mark.parametrize("preset,ch,bps,n,kw", [
    ("best", 2, 24, 4096 * 9 + 100, {}),
    ("default", 2, 16, 44100 * 25 + 7, {}),                       # several seek points (10 s interval)
    ("fast", 1, 8, 1152 * 30, {"padding": -1}),
    ("default", 3, 16, 4096 * 4, {"seektable_mode": 2, "seektable_value": 2}),
    ("default", 2, 16, 4096 * 3, {"seektable_mode": 0}),
])
def test_header_from_frame_sizes_matches_oracle_stream(preset, ch, bps, n, kw):
    from flac_codec_amd.encode import Options, _COptions, _stream_lib

    pcm = synth_fast(8000 + n % 1000, ch, bps, n)
    oo = orc.options(preset, **kw)
    rc, ref, _ = orc.encode_stream(oo, 44100, bps, ch, pcm, total_known=True)
    assert rc == 0
    # where the frames start: after the metadata blocks
    pos = 4
    while True:
        last, ln = ref[pos] & 0x80, int.from_bytes(ref[pos + 1:pos + 4], "big")
        pos += 4 + ln
        if last:
            break
    sizes = frame_sizes(ref, pos)
    opt = getattr(Options, preset)()
    if kw.get("padding") == -1:
        opt = opt.no_padding()
    if "seektable_mode" in kw:
        opt = opt.no_seektable() if kw["seektable_mode"] == 0 else opt.seektable_frames(kw["seektable_value"])
    co = opt._c_options()
    B = co.block_size
    assert len(sizes) == (n + B - 1) // B
    width = (bps + 7) // 8
    md5 = hashlib.md5(np.ascontiguousarray(pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width]).tobytes()).digest()
    L = _stream_lib()
    L.flacenc_stream_header.argtypes = [C.POINTER(_COptions), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                        C.c_size_t, C.POINTER(C.c_size_t)]
    fs = (C.c_uint32 * len(sizes))(*sizes)
    buf = (C.c_uint8 * (1 << 16))()
    ln = C.c_size_t(0)
    rc = L.flacenc_stream_header(C.byref(co), 44100, bps, ch, n, md5, len(sizes), fs, n - (len(sizes) - 1) * B,
                                 buf, len(buf), C.byref(ln))
    assert rc == 0
    assert bytes(buf[: ln.value]) == ref[:pos]
