"""Stream-level GPU parity: a .flac produced through the reference-shaped writers
(flac_codec_amd.encode, over the C ABI) must be BYTE-IDENTICAL to the oracle's for the same
input and options, and decode (oracle's decoder restatement) back to the input with a valid
MD5 -- the latter is how every encoder test of the reference works (tests/format.rs)."""
import io

import numpy as np
import pytest

import _oracle as orc
from _pcm import generate_sine_2, read_raw, synth, synth_fast

pytestmark = pytest.mark.gpu


def orc_opts_from(o):
    c = o._c
    return orc.options("default", block_size=c.block_size, max_partition_order=c.max_partition_order,
                       max_lpc_order=c.max_lpc_order, mid_side=c.mid_side,
                       exhaustive=c.exhaustive_channel_correlation, window_kind=c.window_kind,
                       window_param=c.window_param, padding=(c.padding if c.padding > 0 else -1),
                       seektable_mode=c.seektable_mode, seektable_value=c.seektable_value)


def check_stream(data, pcm, rate, bps, ch, opts, total_known):
    rc, ref, _ = orc.encode_stream(orc_opts_from(opts), rate, bps, ch, pcm, total_known=total_known)
    assert rc == 0
    assert len(data) == len(ref), (len(data), len(ref))
    if data != ref:
        diff = next(i for i in range(len(ref)) if data[i] != ref[i])
        raise AssertionError(f"first differing byte at {diff} of {len(ref)}")
    rc2, out, info = orc.decode_stream(data)
    assert rc2 == 0 and info.md5_ok == 1
    n = pcm.size - pcm.size % ch
    assert np.array_equal(out, pcm[:n])


def encode_samples(opts, rate, bps, ch, pcm, total_known=True, chunk=None, writer=None):
    from flac_codec_amd.encode import FlacSampleWriter

    w = FlacSampleWriter(writer, opts, rate, bps, ch, pcm.size if total_known else None)
    if chunk is None:
        w.write(pcm)
    else:
        for s in range(0, pcm.size, chunk):
            w.write(pcm[s:s + chunk])
    w.finalize()
    data = w.getvalue() if writer is None else writer.getvalue()
    w.close()
    return data


@pytest.mark.parametrize("host_pack", [False, True])
def test_headline_config_stream(host_pack):  # BASELINE config 3, small: L8, 48 kHz / 24-bit stereo
    from flac_codec_amd.encode import Options

    pcm = synth_fast(100, 2, 24, 4096 * 40 + 777)
    opts = Options.best().batch_frames(16).host_pack(host_pack)
    data = encode_samples(opts, 48000, 24, 2, pcm, chunk=10007)
    check_stream(data, pcm, 48000, 24, 2, opts, True)


def test_wav2flac_config_default_16bit():  # BASELINE config 1 shape (10 s 44.1 kHz/16-bit stereo sine)
    from flac_codec_amd.encode import Options

    pcm = generate_sine_2(32767.0, 44100.0, 441000, 441.0, 0.5, 441.0, 0.0, 1.0)
    opts = Options.default()
    data = encode_samples(opts, 44100, 16, 2, pcm)
    check_stream(data, pcm, 44100, 16, 2, opts, True)
    rc, _, info = orc.decode_stream(data)
    assert info.n_seekpoints == 1  # 10 s of audio, one point per 10 s


@pytest.mark.parametrize("total_known", [True, False])
def test_total_known_and_unknown(total_known):
    from flac_codec_amd.encode import Options

    pcm = synth_fast(101, 2, 16, 4096 * 6 + 5)
    opts = Options.default().seektable_seconds(1 if total_known else 10)
    data = encode_samples(opts, 8000, 16, 2, pcm, total_known=total_known)
    check_stream(data, pcm, 8000, 16, 2, opts, total_known)


def test_python_file_object_sink_with_offset():
    from flac_codec_amd.encode import Options

    pcm = synth_fast(102, 1, 16, 4096 * 3)
    buf = io.BytesIO()
    buf.write(b"JUNK")  # stream_position() != 0 at creation (encode.rs:1941)
    opts = Options.best()
    data = encode_samples(opts, 44100, 16, 1, pcm, writer=buf)
    assert data[:4] == b"JUNK"
    check_stream(data[4:], pcm, 44100, 16, 1, opts, True)


@pytest.mark.parametrize("preset", ["default", "fast", "best"])
@pytest.mark.parametrize("ch,bps", [(1, 8), (2, 16), (2, 24), (4, 24), (8, 16), (2, 32)])
def test_presets_channels(preset, ch, bps):  # tests/format.rs:1248-1384 shape
    from flac_codec_amd.encode import Options

    pcm = synth_fast(103 + ch + bps, ch, min(bps, 24), 4096 * 3 + 100)
    if bps == 32:
        pcm = (pcm.astype(np.int64) << 8).astype(np.int32)
    opts = getattr(Options, preset)().no_padding().host_pack(ch == 4)
    data = encode_samples(opts, 44100, bps, ch, pcm, chunk=4099 * ch)
    check_stream(data, pcm, 44100, bps, ch, opts, True)


@pytest.mark.parametrize("ch", [1, 2, 4, 8])
@pytest.mark.parametrize("bps", [8, 16, 24])
@pytest.mark.parametrize("endian", ["little", "big"])
def test_byte_writer_roundtrip_files(ch, bps, endian):  # tests/format.rs:207-435
    from flac_codec_amd.encode import FlacByteWriter, Options

    pcm = read_raw(f"roundtrip-{ch}-{bps}-4777.raw", bps)
    b = bps // 8
    le = b"".join(int(v).to_bytes(b, "little", signed=True) for v in pcm.tolist())
    raw = le if endian == "little" else b"".join(le[i:i + b][::-1] for i in range(0, len(le), b))
    opts = Options.default().no_padding()
    w = FlacByteWriter(None, opts, 44100, bps, ch, len(raw), endian=endian)
    for s in range(0, len(raw), 1000):  # ragged byte chunks that split samples
        w.write(raw[s:s + 1000])
    w.finalize()
    check_stream(w.getvalue(), pcm, 44100, bps, ch, opts, True)
    w.close()


def test_channel_writer():
    from flac_codec_amd.encode import ChannelCountMismatch, ChannelLengthMismatch, FlacChannelWriter, Options

    pcm = synth_fast(110, 2, 16, 4096 * 2 + 321)
    opts = Options.default()
    w = FlacChannelWriter(None, opts, 44100, 16, 2, pcm.size // 2)
    with pytest.raises(ChannelCountMismatch):
        w.write([pcm[0::2]])
    with pytest.raises(ChannelLengthMismatch):
        w.write([pcm[0::2], pcm[1::2][:-1]])
    half = pcm.size // 4
    w.write([pcm[0::2][:half], pcm[1::2][:half]])
    w.write([pcm[0::2][half:], pcm[1::2][half:]])
    w.finalize()
    check_stream(w.getvalue(), pcm, 44100, 16, 2, opts, True)


def test_stream_writer_frames():  # encode.rs:1050-1290
    from flac_codec_amd.encode import FlacStreamWriter, NonSubsetBitsPerSample, NonSubsetSampleRate, Options

    opts = Options.best()
    w = FlacStreamWriter(None, opts)
    oo = orc_opts_from(opts)
    expect = b""
    fn = 0
    for rate, ch, bps, n, seed in [(44100, 2, 16, 1000, 1), (48000, 1, 24, 4096, 2), (96000, 2, 24, 333, 3),
                                   (44100, 2, 16, 16, 4)]:
        pcm = synth_fast(seed, ch, bps, n)
        w.write(rate, ch, bps, pcm)
        planar = np.ascontiguousarray(pcm.reshape(n, ch).T)
        rc, fb, _ = orc.encode_frame(oo, rate, bps, planar, frame_number=fn, subset=True)
        assert rc == 0
        expect += fb
        fn += 1
    assert w.getvalue() == expect
    with pytest.raises(NonSubsetBitsPerSample):
        w.write(44100, 2, 17, np.zeros(32, dtype=np.int32))
    with pytest.raises(NonSubsetSampleRate):
        w.write(700001, 2, 16, np.zeros(32, dtype=np.int32))
    w.close()


def test_finalize_errors():
    from flac_codec_amd.encode import FlacSampleWriter, NoSamples, Options, SampleCountMismatch

    w = FlacSampleWriter(None, Options.default(), 44100, 16, 2, 1000)
    w.write(np.zeros(500, dtype=np.int32))
    with pytest.raises(SampleCountMismatch):  # encode.rs:2083
        w.finalize()
    w.close()
    w = FlacSampleWriter(None, Options.default(), 44100, 16, 2, None)
    with pytest.raises(NoSamples):  # encode.rs:2090
        w.finalize()
    w.close()


def test_small_files():  # tests/format.rs:16-82
    from flac_codec_amd.encode import Options

    opts = Options.fast().max_lpc_order(16).mid_side(True).no_padding()
    for ch in (1, 2):
        for n in range(1, 11):
            pcm = (np.arange(n * ch, dtype=np.int32) * 37 - 100).astype(np.int32)
            data = encode_samples(opts, 44100, 16, ch, pcm)
            check_stream(data, pcm, 44100, 16, ch, opts, True)


@pytest.mark.parametrize("bs,total", [(33, 31), (33, 35), (256, 258), (2048, 2046), (4608, 4610), (4608, 9218)])
def test_fractional(bs, total):  # tests/format.rs:136-205
    from flac_codec_amd.encode import Options

    noise = read_raw("noise-256k.raw", 16)
    opts = Options.default().block_size(bs)
    pcm = noise[: total * 2]
    data = encode_samples(opts, 44100, 16, 2, pcm)
    check_stream(data, pcm, 44100, 16, 2, opts, True)


def test_vorbis_comment_tags():  # Options::tag, encode.rs:1513-1520; block order :1944-1951
    from flac_codec_amd.encode import Options

    pcm = synth_fast(120, 2, 16, 4096 * 3 + 7)
    for total_known in (True, False):
        opts = Options.default().tag("TITLE", "Test").tag("WAVEFORMATEXTENSIBLE_CHANNEL_MASK", "0x0003")
        data = encode_samples(opts, 44100, 16, 2, pcm, total_known=total_known)
        rc, ref, _ = orc.encode_stream(orc_opts_from(opts), 44100, 16, 2, pcm, total_known=total_known,
                                       tags=["TITLE=Test", "WAVEFORMATEXTENSIBLE_CHANNEL_MASK=0x0003"])
        assert rc == 0 and data == ref
        assert data[42] & 0x7F == 4  # VORBIS_COMMENT directly after STREAMINFO
        assert b"flac-codec 1.3.2" in data[:200]
        rc2, out, info = orc.decode_stream(data)
        assert rc2 == 0 and info.md5_ok == 1 and np.array_equal(out, pcm)
    opts = Options.default().no_padding().no_seektable().comment(["A=b"], vendor_string="me")
    data = encode_samples(opts, 44100, 16, 2, pcm)
    rc, ref, _ = orc.encode_stream(orc_opts_from(opts), 44100, 16, 2, pcm, tags=["A=b"], vendor="me")
    assert rc == 0 and data == ref and data[42] == 0x84  # last-block flag on the comment


def test_concurrent_writers_and_context_pool():
    """Many writers at once on Python threads (contexts come from the pool and go back to it),
    different inputs and options per thread, odd write chunkings (the zero-copy path cuts whole
    batches out of the caller's buffer): every stream byte-identical to the oracle's."""
    from concurrent.futures import ThreadPoolExecutor

    from flac_codec_amd.encode import Options

    def job(i):
        ch, bps = [(2, 16), (2, 24), (1, 16), (2, 16)][i % 4]
        opts = [Options.best, Options.default, Options.fast, Options.best][i % 4]().batch_frames(8)
        block = 1152 if i % 4 == 2 else 4096
        n = block * (20 + i) + 13 * i
        pcm = synth_fast(700 + i, ch, bps, n)
        chunk = [None, 4096 * 8 * ch, 9973, 4096 * 8 * ch * 3 + 5][i % 4]
        data = encode_samples(opts, 44100, bps, ch, pcm, chunk=chunk)
        check_stream(data, pcm, 44100, bps, ch, opts, True)
        return len(data)

    for _ in range(2):   # the second round runs entirely on pooled contexts
        with ThreadPoolExecutor(8) as ex:
            sizes = list(ex.map(job, range(16)))
        assert all(s > 0 for s in sizes)


def test_tuning_and_encode_device_fallbacks():
    from flac_codec_amd.gpu import GpuAnalyzer, GpuError

    pcm = synth_fast(720, 2, 24, 4096 * 12)
    an = GpuAnalyzer(4096, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=12)
    ref, ref_off = an.encode_frames(pcm, 12, 4096, 5, 48000)
    with pytest.raises(GpuError):
        an.set_tuning(99, 1)
    with pytest.raises(GpuError):
        an.set_tuning(an.TUNE_LAG_SPLIT, 3)
    an.set_tuning(an.TUNE_LAG_SPLIT, 2)
    an.set_two_ranges(True)            # 12 frames < 256: one range
    got, off = an.encode_frames(pcm, 12, 4096, 5, 48000)
    assert got == ref and off == ref_off
    got, off = an.encode_frames(pcm[: 4096 * 2 * 11 + 2 * 100], 12, 100, 5, 48000)   # short last frame: generic tail
    assert off[:12] == ref_off[:12] and got[: off[11]] == ref[: ref_off[11]]
    an.close()


def test_first_use_from_many_threads_in_a_fresh_process():
    """The ctypes binding is set up lazily; doing that from 16 threads at once used to expose a
    half-bound library (default int return type = truncated pointer -> crash in getvalue)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, hashlib
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
from concurrent.futures import ThreadPoolExecutor
from _pcm import synth_fast
pcm = synth_fast(900, 2, 16, 4096 * 40)
def job(i):
    from flac_codec_amd.encode import FlacSampleWriter, Options
    w = FlacSampleWriter(None, Options.best().batch_frames(8), 44100, 16, 2, pcm.size)
    w.write(pcm); w.finalize(); d = w.getvalue(); w.close()
    return hashlib.sha256(d).hexdigest()
with ThreadPoolExecutor(16) as ex:
    hs = set(ex.map(job, range(64)))
assert len(hs) == 1, hs
print("ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_early_download_of_batches_with_a_generic_path_frame(monkeypatch):
    """FLACGPU_EARLY_DOWNLOAD=1: the frames' D2H copy is queued behind the packing kernels before the sizes are known
    (the previous batch's size as the guess, the remainder afterwards).  r02 saw stale bytes from such a copy for
    batches with a generic-path frame; the cause was on the host -- the lane's pinned destination was re-allocated when
    the sizes arrived -- and the destination is now sized for the worst case BEFORE submission.  Many
    concurrent writers on pooled lanes, every stream ending in a short (generic-path) frame, ragged lengths so that
    guesses are too small as often as too large: every stream must be the oracle's."""
    import ctypes

    from flac_codec_amd import _lib
    from flac_codec_amd.encode import BatchEncoder, Options

    monkeypatch.setenv("FLACGPU_EARLY_DOWNLOAD", "1")
    B = 4096
    shapes = [((3 + (i * 7) % 19) * B + 1 + (i * 977) % (B - 1), 400 + i % 7) for i in range(48)]
    streams = [synth_fast(seed, 2, 24, n) for n, seed in shapes]
    refs = []
    for s in streams:
        rc, o, _ = orc.encode_stream(orc.options("best"), 48000, 24, 2, s, total_known=True)
        assert rc == 0
        refs.append(o)

    # batch_frames 13: lanes no other test of this process has created (the knob is read when a lane's context is made)
    be = BatchEncoder(Options.best().batch_frames(13).pipeline_depth(3), threads=24)
    q0, w0 = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _lib.lib().flacgpu_early_download_counters(ctypes.byref(q0), ctypes.byref(w0))
    for r in range(6):
        outs = be.encode(streams, 48000, 24, 2)
        for k, o in enumerate(outs):
            assert bytes(o) == refs[k], f"round {r} stream {k}"
    q1, w1 = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _lib.lib().flacgpu_early_download_counters(ctypes.byref(q1), ctypes.byref(w1))
    assert q1.value - q0.value >= 100 and w1.value > w0.value     # the path really ran, remainder copies included


def test_batch_encoder_worst_case_buffers():
    """flacenc_worst_case_bytes sizes BatchEncoder's output buffers: full-scale noise (every frame VERBATIM) at a small
    block size, where frame headers, the bps + 1 side channel and one seek point per frame exceed the PCM size --
    the case the former `PCM size + 1/16` rule failed with FLACENC_ERR_IO."""
    from flac_codec_amd.encode import BatchEncoder, Options

    rng = np.random.Generator(np.random.PCG64(91))
    for bps, block in ((8, 192), (16, 256), (24, 4096)):
        lo, hi = -(1 << (bps - 1)), (1 << (bps - 1))
        streams = [rng.integers(lo, hi, size=2 * (block * 37 + 11), dtype=np.int64).astype(np.int32) for _ in range(3)]
        opts = Options.best().block_size(block)
        outs = BatchEncoder(opts, threads=3).encode(streams, 44100, bps, 2)
        for s, o in zip(streams, outs):
            rc, ref, _ = orc.encode_stream(orc.options("best").copy(block_size=block), 44100, bps, 2, s, total_known=True)
            assert rc == 0 and o == ref
            assert len(o) > s.size * ((bps + 7) // 8)      # really larger than the PCM at stream width
