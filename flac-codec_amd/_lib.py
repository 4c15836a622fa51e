"""ctypes loader for libflacenc_amd.so (the C-ABI shared library of this package).

The library holds the hand-written gfx950 kernels; there is NO CPU fallback: if it is
missing or fails to load, importing anything that needs it raises.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLACENC_AMD_LIBRARY: another build of the same library (the sanitizer builds of csrc/Makefile `sanitize`)
LIB_PATH = os.environ.get("FLACENC_AMD_LIBRARY") or os.path.join(_HERE, "libflacenc_amd.so")

MAX_CHANNELS = 8
MAX_LPC_ORDER = 32
MAX_PARTITIONS = 64
N_KERNELS = 12


class GpuOptions(C.Structure):
    """flacgpu_options (include/flacenc_gpu.h)."""
    _fields_ = [
        ("block_size", C.c_uint32),
        ("max_partition_order", C.c_uint32),
        ("max_lpc_order", C.c_uint32),
        ("mid_side", C.c_uint8),
        ("exhaustive_channel_correlation", C.c_uint8),
        ("window_kind", C.c_uint8),
        ("reserved", C.c_uint8),
        ("window_param", C.c_float),
    ]


class SubframePlan(C.Structure):
    """flacgpu_subframe_plan."""
    _fields_ = [
        ("type", C.c_uint8),
        ("wasted", C.c_uint8),
        ("bps", C.c_uint8),
        ("order", C.c_uint8),
        ("precision", C.c_uint8),
        ("shift", C.c_uint8),
        ("coding_method", C.c_uint8),
        ("partition_order", C.c_uint8),
        ("source", C.c_uint8),
        ("reserved", C.c_uint8 * 3),
        ("n_partitions", C.c_uint32),
        ("part_len", C.c_uint32),
        ("bits", C.c_uint32),
        ("coeffs", C.c_int32 * MAX_LPC_ORDER),
        ("rice", C.c_uint8 * MAX_PARTITIONS),
        ("escape_bits", C.c_uint8 * MAX_PARTITIONS),
    ]


class FramePlan(C.Structure):
    """flacgpu_frame_plan."""
    _fields_ = [
        ("assignment", C.c_uint8),
        ("channels", C.c_uint8),
        ("block_size", C.c_uint16),
        ("body_bits", C.c_uint32),
    ]


class VerifyResult(C.Structure):
    """flacgpu_verify_result."""
    _fields_ = [
        ("frames", C.c_uint32),
        ("bad_structure", C.c_uint32),
        ("bad_crc16", C.c_uint32),
        ("frames_pcm_differs", C.c_uint32),
        ("samples_differ", C.c_uint32),
        ("compared_pcm", C.c_uint32),
    ]


class StreamInfo(C.Structure):
    """flacgpu_stream_info."""
    _fields_ = [
        ("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32),
        ("min_block", C.c_uint32), ("max_block", C.c_uint32),
        ("frames", C.c_uint32), ("bad_frames", C.c_uint32), ("bad_crc16", C.c_uint32),
        ("total_samples", C.c_uint64), ("decoded_samples", C.c_uint64),
        ("md5", C.c_uint8 * 16), ("decoded_md5", C.c_uint8 * 16),
        ("md5_status", C.c_uint32), ("reserved", C.c_uint32),
    ]


class ShardCounters(C.Structure):
    """flacgpu_shard_counters: the four integers that cross shards of a stream (include/flacenc_gpu.h)."""
    _fields_ = [("frames", C.c_uint64), ("bytes", C.c_uint64), ("min_frame", C.c_uint64), ("max_frame", C.c_uint64)]

    def as_list(self):
        return [int(self.frames), int(self.bytes), int(self.min_frame), int(self.max_frame)]


class Segment(C.Structure):
    """flacgpu_segment: a run of whole blocks of one stream inside a batch of several (include/flacenc_gpu.h)."""
    _fields_ = [("pcm", C.c_void_p), ("n_frames", C.c_uint32), ("reserved", C.c_uint32), ("first_frame_number", C.c_uint64)]


class GpuStats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint32),
        ("lpc_failed", C.c_uint32),
        ("order_ties", C.c_uint32),
        ("log2_edge", C.c_uint32),
        ("order_ties_resolved", C.c_uint32),
        ("fir_recheck", C.c_uint32),
        ("fir_rechecked", C.c_uint32),
        ("fixed_decided", C.c_uint32),
        ("fixed_refetched", C.c_uint32),
    ]


_lib = None


class LibraryMissing(RuntimeError):
    pass


_load_lock = threading.Lock()


def lib():
    """Load libflacenc_amd.so.  Fails loudly when the HIP extension is not built.  Thread-safe:
    writers are used from many threads, and a half-bound library (default `int` return types
    truncate pointers) must never be visible."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        if _lib is None:
            _lib = _load()
    return _lib


def _load():
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    ip = C.POINTER(C.c_int32)
    L.flacgpu_create.argtypes = [C.POINTER(GpuOptions), C.c_uint32, C.c_uint32, C.c_int,
                                 C.c_uint32, C.POINTER(vp)]
    L.flacgpu_destroy.argtypes = [vp]
    L.flacgpu_destroy.restype = None
    L.flacgpu_last_error.restype = C.c_char_p
    L.flacgpu_analyze.argtypes = [vp, ip, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(FramePlan),
                                  C.POINTER(SubframePlan), ip]
    L.flacgpu_analyze_device.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp]
    L.flacgpu_fetch.argtypes = [vp, C.POINTER(FramePlan), C.POINTER(SubframePlan), ip]
    L.flacgpu_get_stats.argtypes = [vp, C.POINTER(GpuStats)]
    L.flacgpu_resolve.argtypes = [vp]
    L.flacgpu_device_buffer.argtypes = [vp, C.c_int]
    L.flacgpu_device_buffer.restype = vp
    L.flacgpu_set_timing.argtypes = [vp, C.c_int]
    L.flacgpu_get_kernel_ms.argtypes = [vp, C.POINTER(C.c_float * N_KERNELS)]
    L.flacgpu_pack_device.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.flacgpu_set_tuning.argtypes = [vp, C.c_int, C.c_int]
    L.flacgpu_encode_device.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, vp]
    L.flacgpu_fetch_frames.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]
    L.flacgpu_encode_frames.argtypes = [vp, ip, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64)]
    L.flacenc_pack_frames.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                      C.c_uint32, C.c_void_p, C.c_void_p, ip, C.c_uint32,
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.flacgpu_decode_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_int, ip, C.c_size_t, C.POINTER(StreamInfo)]
    L.flacgpu_pack_plans.argtypes = [vp, ip, C.c_uint32, C.c_uint32, C.POINTER(FramePlan), C.POINTER(SubframePlan),
                                     C.c_uint64, C.c_uint32]
    L.flacgpu_host_alloc.argtypes = [C.c_size_t]
    L.flacgpu_host_alloc.restype = vp
    L.flacgpu_host_free.argtypes = [vp]
    L.flacgpu_host_free.restype = None
    L.flacgpu_current_device.argtypes = []
    L.flacgpu_packed_input_supported.argtypes = [vp, C.c_uint32]
    L.flacgpu_encode_packed_async.argtypes = [vp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                              C.c_uint32]
    L.flacgpu_frames_ready.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    L.flacgpu_fetch_frames_async.argtypes = [vp, C.c_void_p, C.c_size_t]
    L.flacgpu_wait.argtypes = [vp]
    L.flacgpu_verify_device.argtypes = [vp, C.c_uint32, C.c_uint64, C.POINTER(VerifyResult),
                                        C.POINTER(C.c_float)]
    L.flacgpu_fetch_decoded.argtypes = [vp, ip]
    L.flacgpu_experiment_mfma_autocorr.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                                   C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    L.flacgpu_kernel_name.argtypes = [C.c_int]
    L.flacgpu_kernel_name.restype = C.c_char_p
    L.flacgpu_packed_cap.argtypes = [vp]
    L.flacgpu_packed_cap.restype = C.c_size_t
    L.flacgpu_pipeline_create.argtypes = [C.POINTER(GpuOptions), C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                          C.POINTER(vp)]
    L.flacgpu_pipeline_destroy.argtypes = [vp]
    L.flacgpu_pipeline_destroy.restype = None
    L.flacgpu_pipeline_submit.argtypes = [vp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    L.flacgpu_pipeline_retire.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_uint64)),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.flacgpu_pipeline_in_flight.argtypes = [vp]
    L.flacgpu_pipeline_in_flight.restype = C.c_uint32
    L.flacgpu_pipeline_depth.argtypes = [vp]
    L.flacgpu_pipeline_depth.restype = C.c_uint32
    L.flacgpu_link_probe.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.flacgpu_encode_segments_device.argtypes = [vp, C.POINTER(Segment), C.c_uint32, C.c_uint32, vp]
    L.flacgpu_encode_segments.argtypes = [vp, C.POINTER(Segment), C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.flacgpu_encode_segments_packed_async_host.argtypes = [vp, C.c_void_p, C.c_uint32, C.POINTER(Segment), C.c_uint32,
                                                            C.c_uint32, C.c_void_p, C.c_size_t]
    sc = C.POINTER(ShardCounters)
    u64p = C.POINTER(C.c_uint64)
    L.flacgpu_merge_counters.argtypes = [sc, C.c_uint32, sc, u64p]
    L.flacgpu_shard_range.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]
    L.flacgpu_shard_range.restype = None
    L.flacgpu_device_count.argtypes = []
    L.flacgpu_multi_create.argtypes = [C.POINTER(GpuOptions), C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.flacgpu_multi_destroy.argtypes = [vp]
    L.flacgpu_multi_destroy.restype = None
    L.flacgpu_multi_shards.argtypes = [vp]
    L.flacgpu_multi_shards.restype = C.c_uint32
    L.flacgpu_multi_device_of.argtypes = [vp, C.c_uint32]
    L.flacgpu_multi_host_copy_stats.argtypes = [vp, u64p, u64p, C.POINTER(C.c_uint32)]
    L.flacgpu_device_numa_info.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.flacgpu_multi_encode.argtypes = [vp, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                       C.c_void_p, C.c_size_t, u64p, u64p, sc, sc]
    L.flacgpu_multi_encode_device.argtypes = [vp, C.c_uint32, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    L.flacgpu_multi_wait.argtypes = [vp]
    L.flacgpu_multi_counters.argtypes = [vp, sc, sc]
    L.flacgpu_multi_last_context.argtypes = [vp, C.c_uint32]
    L.flacgpu_multi_last_context.restype = vp
    L.flacgpu_rccl_available.argtypes = []
    L.flacgpu_rccl_allgather_counters.argtypes = [vp, vp, sc, sc, C.c_uint32, C.POINTER(C.c_uint32),
                                                  C.POINTER(C.c_uint32)]
    L.flacgpu_build_id.argtypes = []
    L.flacgpu_build_id.restype = C.c_char_p
    return L


def build_id():
    """flacgpu_build_id() of the loaded library (hash of the sources it was built from)."""
    return lib().flacgpu_build_id().decode()


def source_build_id():
    """The same hash computed from the sources in the tree (None when they are not there): differs from build_id()
    when the library is stale."""
    import hashlib

    src = os.path.join(_HERE, "csrc")
    files = []
    for sub, pat in (("", ".hip"), ("host", ""), ("kernels", "")):
        d = os.path.join(src, sub)
        if not os.path.isdir(d):
            return None
        files += [os.path.join(sub, f) for f in os.listdir(d)
                  if f.endswith(pat) and os.path.isfile(os.path.join(d, f))]
    files.append("Makefile")
    h = hashlib.sha256()
    for f in sorted(files):      # make's $(sort): byte order
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


def exported_symbols():
    """Names of the dynamic symbols the shared library exports (for the ABI test)."""
    import subprocess

    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB_PATH], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}
