"""Python mirror of the reference's `flac_codec::encode` module surface
(/root/reference/src/encode.rs): `Options`, `Window`, `FlacSampleWriter`, `FlacByteWriter`,
`FlacChannelWriter`, `FlacStreamWriter` -- same names, argument meaning and error behaviour,
so the parity tests read like the reference's own tests (tests/format.rs).

All work is done by libflacenc_amd.so (C ABI in include/flacenc_stream.h, gfx950 kernels
behind include/flacenc_gpu.h).  There is no CPU fallback.
"""
import ctypes as C
import threading
import io
import os

import numpy as np

from . import _lib


# --------------------------------------------------------------------------------------
# errors: the reference's `Error` (src/lib.rs:59-193) and `OptionsError` (encode.rs:1676-1698)
# --------------------------------------------------------------------------------------
class Error(Exception):
    """flac_codec::Error"""
    code = None


class OptionsError(Exception):
    """flac_codec::encode::OptionsError"""
    code = None


def _mk(name, base, code):
    cls = type(name, (base,), {"code": code})
    return cls


InvalidBlockSize = _mk("InvalidBlockSize", OptionsError, -101)
InvalidLpcOrder = _mk("InvalidLpcOrder", OptionsError, -102)
InvalidMaxPartitions = _mk("InvalidMaxPartitions", OptionsError, -103)
ExcessivePadding = _mk("ExcessivePadding", OptionsError, -104)
InvalidBitsPerSample = _mk("InvalidBitsPerSample", Error, -110)
InvalidSampleRate = _mk("InvalidSampleRate", Error, -111)
ExcessiveChannels = _mk("ExcessiveChannels", Error, -112)
SamplesNotDivisibleByChannels = _mk("SamplesNotDivisibleByChannels", Error, -113)
InvalidTotalSamples = _mk("InvalidTotalSamples", Error, -114)
InvalidTotalBytes = _mk("InvalidTotalBytes", Error, -115)
ExcessiveTotalSamples = _mk("ExcessiveTotalSamples", Error, -116)
SampleCountMismatch = _mk("SampleCountMismatch", Error, -117)
NoSamples = _mk("NoSamples", Error, -118)
ExcessiveFrameNumber = _mk("ExcessiveFrameNumber", Error, -119)
ChannelCountMismatch = _mk("ChannelCountMismatch", Error, -120)
ChannelLengthMismatch = _mk("ChannelLengthMismatch", Error, -121)
NonSubsetSampleRate = _mk("NonSubsetSampleRate", Error, -122)
NonSubsetBitsPerSample = _mk("NonSubsetBitsPerSample", Error, -123)
Io = _mk("Io", Error, -130)
InvalidArgument = _mk("InvalidArgument", Error, -140)
Finalized = _mk("Finalized", Error, -141)
GpuError = _mk("GpuError", Error, -150)
Unsupported = _mk("Unsupported", Error, -151)

_BY_CODE = {c.code: c for c in list(globals().values())
            if isinstance(c, type) and issubclass(c, (Error, OptionsError)) and c.code is not None}


def _check(rc):
    if rc == 0:
        return
    msg = _stream_lib().flacenc_last_error().decode(errors="replace")
    raise _BY_CODE.get(rc, Error)(f"{_BY_CODE.get(rc, Error).__name__} ({rc}) {msg}".strip())


# --------------------------------------------------------------------------------------
# C structs of include/flacenc_stream.h
# --------------------------------------------------------------------------------------
class _COptions(C.Structure):
    _fields_ = [
        ("block_size", C.c_uint32),
        ("max_partition_order", C.c_uint32),
        ("max_lpc_order", C.c_uint32),
        ("mid_side", C.c_uint8),
        ("exhaustive_channel_correlation", C.c_uint8),
        ("window_kind", C.c_uint8),
        ("reserved0", C.c_uint8),
        ("window_param", C.c_float),
        ("padding", C.c_int64),
        ("seektable_mode", C.c_int32),
        ("seektable_value", C.c_uint32),
        ("batch_frames", C.c_uint32),
        ("device", C.c_int32),
        ("pack_threads", C.c_uint32),
        ("host_pack", C.c_uint32),
        ("vendor_string", C.c_char_p),
        ("comment_fields", C.POINTER(C.c_char_p)),
        ("n_comment_fields", C.c_uint32),
        ("pipeline_depth", C.c_uint32),
        ("shared_md5", C.c_uint32),
    ]


_WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
_SEEK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)


class _CSink(C.Structure):
    _fields_ = [("write", _WRITE_FN), ("seek", _SEEK_FN), ("user", C.c_void_p), ("start", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("samples_per_channel", C.c_uint64),
        ("bytes_written", C.c_uint64),
        ("min_frame_size", C.c_uint32),
        ("max_frame_size", C.c_uint32),
        ("md5", C.c_uint8 * 16),
        ("gpu_ms", C.c_double),
        ("pack_ms", C.c_double),
        ("md5_ms", C.c_double),
    ]


class _CJob(C.Structure):
    _fields_ = [
        ("samples", C.POINTER(C.c_int32)),
        ("count", C.c_size_t),
        ("sample_rate", C.c_uint32),
        ("bits_per_sample", C.c_uint32),
        ("channels", C.c_uint32),
        ("reserved", C.c_uint32),
        ("out", C.c_void_p),
        ("out_cap", C.c_size_t),
        ("out_len", C.c_size_t),
        ("status", C.c_int32),
        ("reserved1", C.c_int32),
        ("elapsed_ms", C.c_double),
        ("pack_ms", C.c_double),
        ("gpu_ms", C.c_double),
        ("md5_ms", C.c_double),
        ("start_ms", C.c_double),
    ]


_bound = False
_bind_lock = threading.Lock()


def _stream_lib():
    """The library with the stream-writer entry points bound (once, under a lock: a thread must
    never call through a function whose restype is still the default int)."""
    global _bound
    L = _lib.lib()
    if _bound:
        return L
    with _bind_lock:
        if _bound:
            return L
        vp, ip = C.c_void_p, C.POINTER(C.c_int32)
        po = C.POINTER(_COptions)
        ps = C.POINTER(_CSink)
        for n in ("flacenc_options_default", "flacenc_options_fast", "flacenc_options_best"):
            getattr(L, n).argtypes = [po]
            getattr(L, n).restype = None
        L.flacenc_options_validate.argtypes = [po]
        L.flacenc_sample_writer_new.argtypes = [po, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                C.c_uint64, ps, C.POINTER(vp)]
        L.flacenc_byte_writer_new.argtypes = [po, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                              C.c_uint64, C.c_int, ps, C.POINTER(vp)]
        L.flacenc_channel_writer_new.argtypes = [po, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                 C.c_uint64, ps, C.POINTER(vp)]
        L.flacenc_write_samples.argtypes = [vp, ip, C.c_size_t]
        L.flacenc_write_bytes.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.flacenc_write_channels.argtypes = [vp, C.POINTER(ip), C.c_uint32, C.c_size_t]
        L.flacenc_finalize.argtypes = [vp]
        L.flacenc_writer_free.argtypes = [vp]
        L.flacenc_writer_free.restype = None
        L.flacenc_writer_data.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.flacenc_writer_data.restype = C.POINTER(C.c_uint8)
        L.flacenc_writer_stats.argtypes = [vp, C.POINTER(Stats)]
        L.flacenc_stream_writer_new.argtypes = [po, ps, C.POINTER(vp)]
        L.flacenc_stream_writer_write.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, ip, C.c_size_t]
        L.flacenc_stream_writer_data.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.flacenc_stream_writer_data.restype = C.POINTER(C.c_uint8)
        L.flacenc_stream_writer_free.argtypes = [vp]
        L.flacenc_stream_writer_free.restype = None
        L.flacenc_last_error.restype = C.c_char_p
        L.flacenc_encode_many.argtypes = [po, C.POINTER(_CJob), C.c_size_t, C.c_uint32]
        L.flacenc_encode_many_devices.argtypes = [po, C.POINTER(_CJob), C.c_size_t, C.c_uint32, C.POINTER(C.c_int),
                                                  C.c_uint32]
        L.flacenc_encode_many_coalesced.argtypes = [po, C.POINTER(_CJob), C.c_size_t, C.c_uint32]
        _bound = True
    return L


# --------------------------------------------------------------------------------------
# Window (encode.rs:1711-1720) and Options (encode.rs:1361-1672)
# --------------------------------------------------------------------------------------
class Window:
    """The method to use for windowing the input signal (encode.rs:1713)."""

    def __init__(self, kind, param=0.0):
        self.kind, self.param = kind, float(param)

    Rectangle = None
    Hann = None

    @staticmethod
    def Tukey(p):
        return Window(2, p)

    def __repr__(self):
        return {0: "Rectangle", 1: "Hann"}.get(self.kind, f"Tukey({self.param})")


Window.Rectangle = Window(0)
Window.Hann = Window(1)


class Options:
    """FLAC encoding options (encode.rs:1363).  Builder methods return `self` and raise the
    `OptionsError` the reference's builder returns."""

    def __init__(self, preset="default"):
        self._c = _COptions()
        getattr(_stream_lib(), "flacenc_options_" + preset)(C.byref(self._c))
        self._clobber = False
        self._vendor = None      # VORBIS_COMMENT vendor string (None = reference default)
        self._fields = None      # list of "NAME=value" strings, None = no block

    # presets, encode.rs:1376-1408, 1635-1657
    @classmethod
    def default(cls):
        return cls("default")

    @classmethod
    def fast(cls):
        return cls("fast")

    @classmethod
    def best(cls):
        return cls("best")

    def block_size(self, block_size):  # encode.rs:1418
        if block_size < 16 or block_size > 65535:
            raise InvalidBlockSize("block size must be >= 16")
        self._c.block_size = block_size
        return self

    def max_lpc_order(self, max_lpc_order):  # encode.rs:1430; None = no LPC subframes
        if max_lpc_order is None:
            self._c.max_lpc_order = 0
        else:
            if not 0 < max_lpc_order <= 32:
                raise InvalidLpcOrder("maximum LPC order must be <= 32")
            self._c.max_lpc_order = max_lpc_order
        return self

    def max_partition_order(self, max_partition_order):  # encode.rs:1447
        if not 0 <= max_partition_order <= 15:
            raise InvalidMaxPartitions("max partition order must be <= 15")
        self._c.max_partition_order = max_partition_order
        return self

    def mid_side(self, mid_side):  # encode.rs:1460
        self._c.mid_side = int(bool(mid_side))
        return self

    def window(self, window):  # encode.rs:1465
        self._c.window_kind, self._c.window_param = window.kind, window.param
        return self

    def fast_channel_correlation(self, fast):  # encode.rs:1472
        self._c.exhaustive_channel_correlation = int(not fast)
        return self

    def padding(self, size):  # encode.rs:1486
        if size < 0 or size >= (1 << 24):
            raise ExcessivePadding("padding size is too large for block")
        self._c.padding = size
        return self

    def no_padding(self):  # encode.rs:1505
        self._c.padding = 0
        return self

    def tag(self, field, value):  # encode.rs:1513: adds NAME=value, creating the block if needed
        if "=" in field:
            raise ValueError("field must not contain '='")  # the reference panics (metadata/mod.rs:2357)
        if self._fields is None:
            self._fields = []
        self._fields.append(f"{field}={value}")
        return self

    def comment(self, fields, vendor_string=None):  # encode.rs:1525: replaces the whole block
        self._fields = list(fields)
        self._vendor = vendor_string
        return self

    def _c_options(self):
        """The C struct with the VORBIS_COMMENT pointers filled in (kept alive on self)."""
        if self._fields is None:
            self._c.vendor_string, self._c.n_comment_fields = None, 0
            self._c.comment_fields = None
        else:
            self._keep = (C.c_char_p * max(len(self._fields), 1))(*[f.encode() for f in self._fields])
            self._c.comment_fields = self._keep
            self._c.n_comment_fields = len(self._fields)
            self._c.vendor_string = (self._vendor or "flac-codec 1.3.2").encode()
        return self._c

    def seektable_seconds(self, seconds):  # encode.rs:1568
        self._c.seektable_mode, self._c.seektable_value = (1, seconds & 0xFF) if seconds else (0, 0)
        return self

    def seektable_frames(self, frames):  # encode.rs:1579
        self._c.seektable_mode, self._c.seektable_value = (2, frames) if frames else (0, 0)
        return self

    def no_seektable(self):  # encode.rs:1585
        self._c.seektable_mode = 0
        return self

    def overwrite(self):  # encode.rs:1627
        self._clobber = True
        return self

    # execution knobs without a reference counterpart
    def batch_frames(self, n):
        self._c.batch_frames = n
        return self

    def device(self, ordinal):
        self._c.device = ordinal
        return self

    def pack_threads(self, n):
        self._c.pack_threads = n
        return self

    def pipeline_depth(self, n):
        """Batches in flight per writer (each on its own context / HIP stream / pinned staging)."""
        self._c.pipeline_depth = n
        return self

    def shared_md5(self, on=True):
        """Hash this stream on the shared multi-stream MD5 engines (many concurrent writers)."""
        self._c.shared_md5 = int(bool(on))
        return self

    def host_pack(self, on=True):
        """Keep Rice bit-packing + CRC on the host (default: assembled on the GPU)."""
        self._c.host_pack = int(bool(on))
        return self

    def _open(self, path):  # Options::create, encode.rs:1660-1671
        return open(path, "wb" if self._clobber else "xb")


# --------------------------------------------------------------------------------------
# writers
# --------------------------------------------------------------------------------------
class _Writer:
    def __init__(self, writer):
        self._h = C.c_void_p(None)
        self._py = writer
        self._sink = None
        self._closed = False
        if writer is not None:
            start = writer.tell()

            def _w(_user, data, n):
                try:
                    writer.write(C.string_at(data, n))
                    return 0
                except Exception:
                    return 1

            def _s(_user, off):
                try:
                    writer.seek(off)
                    return 0
                except Exception:
                    return 1

            self._cb = (_WRITE_FN(_w), _SEEK_FN(_s))  # keep alive
            self._sink = _CSink(self._cb[0], self._cb[1], None, start)

    def _sink_ptr(self):
        return C.byref(self._sink) if self._sink is not None else None

    def finalize(self):
        """Finalizes the stream (encode.rs:624): last partial block, SEEKTABLE, STREAMINFO."""
        _check(_stream_lib().flacenc_finalize(self._h))

    def getvalue(self):
        """Bytes of the stream when no writer object was given (memory sink)."""
        n = C.c_size_t(0)
        p = _stream_lib().flacenc_writer_data(self._h, C.byref(n))
        return C.string_at(p, n.value) if p else b""

    def stats(self):
        s = Stats()
        _check(_stream_lib().flacenc_writer_stats(self._h, C.byref(s)))
        return s

    def close(self):
        """Drop: finalize silently and free (encode.rs:399-405)."""
        if self._h:
            _stream_lib().flacenc_writer_free(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _total(total):
    return (0, 0) if total is None else (1, int(total))


class FlacSampleWriter(_Writer):
    """A FLAC writer which accepts samples as signed integers (encode.rs:461)."""

    def __init__(self, writer, options, sample_rate, bits_per_sample, channels, total_samples=None):
        super().__init__(writer)
        ht, tv = _total(total_samples)
        _check(_stream_lib().flacenc_sample_writer_new(
            C.byref(options._c_options()), sample_rate, bits_per_sample, channels, ht, tv, self._sink_ptr(),
            C.byref(self._h)))

    @classmethod
    def new_cdda(cls, writer, options, total_samples=None):  # encode.rs:535
        return cls(writer, options, 44100, 16, 2, total_samples)

    @classmethod
    def create(cls, path, options, sample_rate, bits_per_sample, channels, total_samples=None):
        return cls(options._open(path), options, sample_rate, bits_per_sample, channels, total_samples)

    def write(self, samples):
        """Interleaved samples, any count (encode.rs:558)."""
        s = np.ascontiguousarray(samples, dtype=np.int32)
        _check(_stream_lib().flacenc_write_samples(self._h, s.ctypes.data_as(C.POINTER(C.c_int32)), s.size))


class FlacByteWriter(_Writer, io.RawIOBase):
    """A FLAC writer which accepts samples as bytes (encode.rs:105); `endian` is "little"
    or "big" (the reference's `Endianness` type parameter)."""

    def __init__(self, writer, options, sample_rate, bits_per_sample, channels, total_bytes=None,
                 endian="little"):
        _Writer.__init__(self, writer)
        ht, tv = _total(total_bytes)
        _check(_stream_lib().flacenc_byte_writer_new(
            C.byref(options._c_options()), sample_rate, bits_per_sample, channels, ht, tv,
            int(endian == "big"), self._sink_ptr(), C.byref(self._h)))

    @classmethod
    def endian(cls, writer, endianness, options, sample_rate, bits_per_sample, channels,
               total_bytes=None):  # encode.rs:205
        return cls(writer, options, sample_rate, bits_per_sample, channels, total_bytes, endianness)

    @classmethod
    def new_cdda(cls, writer, options, total_bytes=None):
        return cls(writer, options, 44100, 16, 2, total_bytes)

    @classmethod
    def create(cls, path, options, sample_rate, bits_per_sample, channels, total_bytes=None,
               endian="little"):
        return cls(options._open(path), options, sample_rate, bits_per_sample, channels, total_bytes, endian)

    def write(self, data):  # io::Write::write, encode.rs:359: the whole buffer is consumed
        b = bytes(data)
        _check(_stream_lib().flacenc_write_bytes(self._h, b, len(b)))
        return len(b)

    def writable(self):
        return True

    def close(self):
        _Writer.close(self)


class FlacChannelWriter(_Writer):
    """A FLAC writer which accepts samples as channels of signed integers (encode.rs:735)."""

    def __init__(self, writer, options, sample_rate, bits_per_sample, channels, total_samples=None):
        super().__init__(writer)
        self._channels = channels
        ht, tv = _total(total_samples)
        _check(_stream_lib().flacenc_channel_writer_new(
            C.byref(options._c_options()), sample_rate, bits_per_sample, channels, ht, tv, self._sink_ptr(),
            C.byref(self._h)))

    @classmethod
    def new_cdda(cls, writer, options, total_samples=None):
        return cls(writer, options, 44100, 16, 2, total_samples)

    def write(self, channels):
        """`channels`: one sequence of samples per channel, all of one length (encode.rs:832)."""
        chans = [np.ascontiguousarray(c, dtype=np.int32) for c in channels]
        if len(chans) != self._channels:  # encode.rs:856-859
            raise ChannelCountMismatch("ChannelCountMismatch")
        if any(c.size != chans[0].size for c in chans):  # encode.rs:849-855
            raise ChannelLengthMismatch("ChannelLengthMismatch")
        ptrs = (C.POINTER(C.c_int32) * len(chans))(*[c.ctypes.data_as(C.POINTER(C.c_int32)) for c in chans])
        _check(_stream_lib().flacenc_write_channels(self._h, ptrs, len(chans), chans[0].size))


class FlacStreamWriter:
    """A FLAC writer which outputs header-less subset frames, one per `write` call
    (encode.rs:1050)."""

    def __init__(self, writer, options):
        self._h = C.c_void_p(None)
        self._py = writer
        self._sink = None
        if writer is not None:
            def _w(_user, data, n):
                try:
                    writer.write(C.string_at(data, n))
                    return 0
                except Exception:
                    return 1

            self._cb = (_WRITE_FN(_w), _SEEK_FN(lambda _u, _o: 1))
            self._sink = _CSink(self._cb[0], self._cb[1], None, 0)
        _check(_stream_lib().flacenc_stream_writer_new(
            C.byref(options._c_options()), C.byref(self._sink) if self._sink is not None else None,
            C.byref(self._h)))

    def write(self, sample_rate, channels, bits_per_sample, samples):  # encode.rs:1142
        s = np.ascontiguousarray(samples, dtype=np.int32)
        _check(_stream_lib().flacenc_stream_writer_write(
            self._h, sample_rate, channels, bits_per_sample, s.ctypes.data_as(C.POINTER(C.c_int32)), s.size))

    def write_cdda(self, samples):  # encode.rs:1270
        self.write(44100, 2, 16, samples)

    def getvalue(self):
        n = C.c_size_t(0)
        p = _stream_lib().flacenc_stream_writer_data(self._h, C.byref(n))
        return C.string_at(p, n.value) if p else b""

    def close(self):
        if self._h:
            _stream_lib().flacenc_stream_writer_free(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchEncoder:
    """Many independent streams at once (flacenc_encode_many): the C++ front end keeps `threads` host
    workers busy, each running FlacSampleWriter::new / write / finalize (encode.rs:487, 558, 624) for
    one stream at a time; output buffers are allocated once and reused by later calls."""

    def __init__(self, options, threads=0, devices=None, coalesce=False):
        """devices: None (the options' device), "all" (every visible device) or a list of HIP ordinals -- stream i is
        encoded whole on devices[i mod len] (flacenc_encode_many_devices).  coalesce: streams of one shape share analysis
        batches (flacenc_encode_many_coalesced: many SMALL streams)."""
        self._opts, self._threads, self._devices, self._coalesce = options, threads, devices, coalesce
        self._bufs = []

    def prepare(self, streams, sample_rate, bits_per_sample, channels):
        """The job array of a call (what a C caller would hold): arrays pinned in a handle, output buffers sized for the
        worst case.  run(handle) then is the C entry point alone."""
        L = _stream_lib()
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in streams]
        jobs = (_CJob * len(arrs))()
        co = self._opts._c_options()
        L.flacenc_worst_case_bytes.restype = C.c_size_t
        L.flacenc_worst_case_bytes.argtypes = [C.POINTER(_COptions), C.c_uint32, C.c_uint32, C.c_uint64]
        for i, a in enumerate(arrs):
            # every frame VERBATIM with its headers + a seek point per frame + the metadata: never too small
            cap = int(L.flacenc_worst_case_bytes(C.byref(co), bits_per_sample, channels, a.size // max(1, channels)))
            if i >= len(self._bufs):
                self._bufs.append(np.empty(cap, dtype=np.uint8))
            elif self._bufs[i].size < cap:
                self._bufs[i] = np.empty(cap, dtype=np.uint8)
            j = jobs[i]
            j.samples = a.ctypes.data_as(C.POINTER(C.c_int32))
            j.count = a.size
            j.sample_rate, j.bits_per_sample, j.channels = sample_rate, bits_per_sample, channels
            j.out = self._bufs[i].ctypes.data
            j.out_cap = self._bufs[i].size
        return (jobs, arrs, co)

    def run(self, handle):
        """One call of flacenc_encode_many / _coalesced / _devices on a prepared job array."""
        L = _stream_lib()
        jobs, arrs, co = handle
        if self._coalesce:
            _check(L.flacenc_encode_many_coalesced(C.byref(co), jobs, len(arrs), self._threads))
        elif self._devices is None:
            _check(L.flacenc_encode_many(C.byref(co), jobs, len(arrs), self._threads))
        elif self._devices == "all":
            _check(L.flacenc_encode_many_devices(C.byref(co), jobs, len(arrs), self._threads, None, 0xFFFFFFFF))
        else:
            devs = (C.c_int * len(self._devices))(*self._devices)
            _check(L.flacenc_encode_many_devices(C.byref(co), jobs, len(arrs), self._threads, devs, len(self._devices)))

    def results(self, handle, copy=True):
        jobs, arrs, _ = handle
        self.last_jobs = [{k: getattr(jobs[i], k) for k in ("elapsed_ms", "pack_ms", "gpu_ms", "md5_ms", "start_ms")}
                          for i in range(len(arrs))]
        views = [self._bufs[i][: jobs[i].out_len] for i in range(len(arrs))]
        return [v.tobytes() for v in views] if copy else views

    def encode(self, streams, sample_rate, bits_per_sample, channels, copy=True):
        """streams: list of interleaved int32 arrays.  Returns the .flac bytes of every stream (or
        memoryviews into the reused output buffers with copy=False)."""
        h = self.prepare(streams, sample_rate, bits_per_sample, channels)
        self.run(h)
        return self.results(h, copy)
