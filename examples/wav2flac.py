#!/usr/bin/env python3
"""wav2flac -- counterpart of the reference's examples/wav2flac.rs (lines 37-130): parse the
RIFF/`fmt `/`data` chunks, stream the PCM bytes through FlacByteWriter with Options::default()
(+ nothing else: WAVE_FORMAT_EXTENSIBLE channel masks are ignored here), finalize.

    python examples/wav2flac.py in.wav [out.flac]
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_wav(f):
    riff, _size, wave = struct.unpack("<4sI4s", f.read(12))
    if riff != b"RIFF" or wave != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    fmt = None
    while True:
        hdr = f.read(8)
        if len(hdr) < 8:
            raise ValueError("no data chunk")
        cid, size = struct.unpack("<4sI", hdr)
        if cid == b"fmt ":
            body = f.read(size + (size & 1))
            tag, channels, rate, _br, _align, bps = struct.unpack("<HHIIHH", body[:16])
            if tag not in (1, 0xFFFE):
                raise ValueError("only PCM WAV is supported")
            fmt = (channels, rate, bps)
        elif cid == b"data":
            if fmt is None:
                raise ValueError("data chunk before fmt chunk")
            return fmt, size
        else:
            f.seek(size + (size & 1), 1)


def convert_wav(src, dst):
    from flac_codec_amd.encode import FlacByteWriter, Options

    with open(src, "rb") as f:
        (channels, rate, bps), size = parse_wav(f)
        with open(dst, "wb") as out:
            w = FlacByteWriter(out, Options.default(), rate, bps, channels, size)
            remaining = size
            while remaining:
                chunk = f.read(min(remaining, 1 << 20))
                if not chunk:
                    break
                if bps <= 8:  # 8-bit WAV is unsigned (examples/wav2flac.rs:132-146)
                    chunk = bytes((b - 128) & 0xFF for b in chunk)
                w.write(chunk)
                remaining -= len(chunk)
            w.finalize()
            w.close()


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(src)[0] + ".flac"
    convert_wav(src, dst)
    print(f"{src} -> {dst} ({os.path.getsize(dst)} bytes)")
