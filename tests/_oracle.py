"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
# FLAC_ORACLE_LIBRARY: the sanitizer build of the oracle (oracle/Makefile `asan`)
_LIB_PATH = os.environ.get("FLAC_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")

MAX_CH, MAX_LPC, MAX_PART = 8, 32, 64

SUB_CONSTANT, SUB_VERBATIM, SUB_FIXED, SUB_LPC = 0, 1, 2, 3
WINDOW_RECTANGLE, WINDOW_HANN, WINDOW_TUKEY = 0, 1, 2


class Options(C.Structure):
    _fields_ = [
        ("block_size", C.c_uint32),
        ("max_partition_order", C.c_uint32),
        ("mid_side", C.c_int32),
        ("max_lpc_order", C.c_int32),
        ("window_kind", C.c_int32),
        ("window_param", C.c_float),
        ("exhaustive", C.c_int32),
        ("padding", C.c_int32),
        ("seektable_mode", C.c_int32),
        ("seektable_value", C.c_uint32),
    ]

    def copy(self, **kw):
        o = Options.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            setattr(o, k, v)
        return o


class SubframePlan(C.Structure):
    _fields_ = [
        ("type", C.c_uint8),
        ("wasted", C.c_uint8),
        ("bps", C.c_uint8),
        ("order", C.c_uint8),
        ("precision", C.c_uint8),
        ("shift", C.c_uint8),
        ("coding_method", C.c_uint8),
        ("partition_order", C.c_uint8),
        ("n_partitions", C.c_uint32),
        ("bits", C.c_uint32),
        ("coeffs", C.c_int32 * MAX_LPC),
        ("rice", C.c_uint8 * MAX_PART),
        ("escape_bits", C.c_uint8 * MAX_PART),
        ("part_len", C.c_uint16 * MAX_PART),
    ]


class FramePlan(C.Structure):
    _fields_ = [
        ("assignment", C.c_uint8),
        ("channels", C.c_uint8),
        ("block_size", C.c_uint16),
        ("frame_bytes", C.c_uint32),
        ("source", C.c_uint8 * MAX_CH),
        ("sub", SubframePlan * MAX_CH),
    ]


class StreamStats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("samples_written", C.c_uint64),
        ("min_frame_size", C.c_uint32),
        ("max_frame_size", C.c_uint32),
        ("md5", C.c_uint8 * 16),
        ("first_frame_offset", C.c_uint64),
    ]


class VorbisComment(C.Structure):
    _fields_ = [("vendor", C.c_char_p), ("fields", C.POINTER(C.c_char_p)), ("n_fields", C.c_uint32)]


class DecodedInfo(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint32),
        ("bps", C.c_uint32),
        ("min_block", C.c_uint32),
        ("max_block", C.c_uint32),
        ("min_frame", C.c_uint32),
        ("max_frame", C.c_uint32),
        ("total_samples", C.c_uint64),
        ("md5", C.c_uint8 * 16),
        ("md5_ok", C.c_int),
        ("frames", C.c_uint64),
        ("n_seekpoints", C.c_uint32),
    ]


_lib = None


def build():
    """(Re)build oracle/liboracle.so with gcc when missing or stale."""
    src = os.path.join(ORACLE_DIR, "flac_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "flac_oracle.h")
    if os.environ.get("FLAC_ORACLE_LIBRARY"):
        return _LIB_PATH   # a build made by somebody else (the sanitizer run)
    if (not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_window_generate.argtypes = [C.c_int, C.c_float, C.c_uint32, dp]
        L.orc_window_generate.restype = None
        L.orc_autocorrelate.argtypes = [dp, C.c_uint32, C.c_uint32, dp]
        L.orc_lp_coefficients.argtypes = [dp, C.c_int, C.c_void_p, dp]
        L.orc_subframe_bits_by_order.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, dp, C.c_int, dp]
        L.orc_compute_best_order.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, dp, C.c_int]
        L.orc_quantize.argtypes = [C.c_int, dp, C.c_uint32, ip, C.POINTER(C.c_uint32)]
        L.orc_encode_residuals.argtypes = [C.c_int, ip, C.c_uint32, ip, C.c_uint32, ip]
        L.orc_lpc_precision.argtypes = [C.c_uint32]
        L.orc_lpc_precision.restype = C.c_uint32
        L.orc_encode_frame.argtypes = [
            C.POINTER(Options), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ip), C.c_uint32,
            C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
            C.POINTER(C.c_size_t), C.POINTER(FramePlan)]
        L.orc_subframe_residuals.argtypes = [C.POINTER(SubframePlan), ip, C.c_uint32, ip]
        L.orc_encode_stream.argtypes = [
            C.POINTER(Options), C.c_uint32, C.c_uint32, C.c_uint32, ip, C.c_uint64, C.c_int,
            C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(StreamStats)]
        L.orc_encode_stream_vc.argtypes = [
            C.POINTER(Options), C.POINTER(VorbisComment), C.c_uint32, C.c_uint32, C.c_uint32, ip,
            C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
            C.POINTER(StreamStats)]
        L.orc_decode_stream.argtypes = [
            C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
            C.POINTER(DecodedInfo)]
        L.orc_md5.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_md5.restype = None
        L.orc_crc8.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_crc8.restype = C.c_uint8
        L.orc_crc16.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_crc16.restype = C.c_uint16
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        for n in ("orc_options_default", "orc_options_fast", "orc_options_best"):
            getattr(L, n).argtypes = [C.POINTER(Options)]
            getattr(L, n).restype = None
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def options(preset="default", **kw):
    o = Options()
    getattr(lib(), "orc_options_" + preset)(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def window(kind, p, n):
    w = np.empty(n, dtype=np.float64)
    lib().orc_window_generate(kind, p, n, _dptr(w))
    return w


def autocorrelate(windowed, max_order):
    windowed = np.ascontiguousarray(windowed, dtype=np.float64)
    out = np.zeros(MAX_LPC + 1, dtype=np.float64)
    cnt = lib().orc_autocorrelate(_dptr(windowed), len(windowed), max_order, _dptr(out))
    return out[:cnt].copy()


def lp_coefficients(ac):
    ac = np.ascontiguousarray(ac, dtype=np.float64)
    coeffs = np.zeros((MAX_LPC, MAX_LPC), dtype=np.float64)
    errors = np.zeros(MAX_LPC, dtype=np.float64)
    cnt = lib().orc_lp_coefficients(_dptr(ac), len(ac), coeffs.ctypes.data, _dptr(errors))
    return [coeffs[i, : i + 1].copy() for i in range(cnt)], errors[:cnt].copy()


def subframe_bits_by_order(bps, precision, n, errors):
    errors = np.ascontiguousarray(errors, dtype=np.float64)
    bits = np.zeros(MAX_LPC, dtype=np.float64)
    cnt = lib().orc_subframe_bits_by_order(bps, precision, n, _dptr(errors), len(errors), _dptr(bits))
    return bits[:cnt].copy()


def compute_best_order(bps, precision, n, errors):
    errors = np.ascontiguousarray(errors, dtype=np.float64)
    return lib().orc_compute_best_order(bps, precision, n, _dptr(errors), len(errors))


def quantize(coeffs, precision):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    q = np.zeros(MAX_LPC, dtype=np.int32)
    shift = C.c_uint32(0)
    rc = lib().orc_quantize(len(coeffs), _dptr(coeffs), precision, _iptr(q), C.byref(shift))
    return rc, q[: len(coeffs)].copy(), shift.value


def encode_residuals(qlp, shift, samples):
    qlp = np.ascontiguousarray(qlp, dtype=np.int32)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    out = np.zeros(max(len(samples), 1), dtype=np.int32)
    rc = lib().orc_encode_residuals(len(qlp), _iptr(qlp), shift, _iptr(samples), len(samples), _iptr(out))
    return rc, out[: len(samples) - len(qlp)].copy()


def encode_frame(opts, sample_rate, bps, planar, frame_number=0, subset=False):
    """planar: int32 array [channels][n].  Returns (rc, frame_bytes, FramePlan)."""
    planar = np.ascontiguousarray(planar, dtype=np.int32)
    nch, n = planar.shape
    ptrs = (C.POINTER(C.c_int32) * nch)(*[_iptr(planar[c]) for c in range(nch)])
    out = C.c_void_p(None)
    out_len = C.c_size_t(0)
    out_cap = C.c_size_t(0)
    plan = FramePlan()
    rc = lib().orc_encode_frame(C.byref(opts), sample_rate, bps, nch, ptrs, n, frame_number,
                                int(subset), C.byref(out), C.byref(out_len), C.byref(out_cap),
                                C.byref(plan))
    data = C.string_at(out, out_len.value) if out.value else b""
    if out.value:
        lib().orc_free(out)
    return rc, data, plan


def subframe_residuals(sub_plan, cand_samples):
    cand = np.ascontiguousarray(cand_samples, dtype=np.int32)
    out = np.zeros(max(len(cand), 1), dtype=np.int32)
    cnt = lib().orc_subframe_residuals(C.byref(sub_plan), _iptr(cand), len(cand), _iptr(out))
    return out[:cnt].copy()


def encode_stream(opts, sample_rate, bps, channels, interleaved, total_known=True, threads=1,
                  tags=None, vendor=None):
    """Returns (rc, flac_bytes, StreamStats).  tags: list of "NAME=value" strings."""
    s = np.ascontiguousarray(interleaved, dtype=np.int32)
    out = C.c_void_p(None)
    out_len = C.c_size_t(0)
    st = StreamStats()
    vc = None
    if tags is not None:
        arr = (C.c_char_p * max(len(tags), 1))(*[t.encode() for t in tags])
        vc = VorbisComment(vendor.encode() if vendor else None, arr, len(tags))
    rc = lib().orc_encode_stream_vc(C.byref(opts), C.byref(vc) if vc is not None else None,
                                    sample_rate, bps, channels, _iptr(s), s.size,
                                    int(total_known), threads, C.byref(out), C.byref(out_len),
                                    C.byref(st))
    data = C.string_at(out, out_len.value) if out.value else b""
    if out.value:
        lib().orc_free(out)
    return rc, data, st


def decode_stream(data):
    """Returns (rc, interleaved int32 array, DecodedInfo)."""
    out = C.c_void_p(None)
    cnt = C.c_uint64(0)
    info = DecodedInfo()
    rc = lib().orc_decode_stream(data, len(data), C.byref(out), C.byref(cnt), C.byref(info))
    if rc != 0 or not out.value:
        return rc, np.zeros(0, dtype=np.int32), info
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(cnt.value,)).copy()
    lib().orc_free(out)
    return rc, arr, info


def md5(data):
    out = C.create_string_buffer(16)
    lib().orc_md5(bytes(data), len(data), out)
    return out.raw


def crc8(data):
    return lib().orc_crc8(bytes(data), len(data))


def crc16(data):
    return lib().orc_crc16(bytes(data), len(data))
