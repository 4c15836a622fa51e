"""Thin Python handle on the analysis C ABI (include/flacenc_gpu.h).

`GpuAnalyzer` replaces the analysis half of the reference's `encode_frame`
(/root/reference/src/encode.rs:2259-2406) for a batch of frames.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FramePlan, GpuOptions, GpuStats, SubframePlan

LAYOUT_INTERLEAVED, LAYOUT_PLANAR = 0, 1


class GpuError(RuntimeError):
    def __init__(self, code, where):
        msg = _lib.lib().flacgpu_last_error().decode(errors="replace")
        super().__init__(f"{where}: flacgpu error {code}: {msg}")
        self.code = code


class GpuAnalyzer:
    def __init__(self, block_size, max_partition_order, max_lpc_order, mid_side, exhaustive,
                 window_kind, window_param, bits_per_sample, channels, max_frames, device=-1):
        L = _lib.lib()
        o = GpuOptions(block_size, max_partition_order, max_lpc_order or 0, int(bool(mid_side)),
                       int(bool(exhaustive)), window_kind, 0, window_param)
        self._h = C.c_void_p(None)
        self.block_size, self.channels, self.max_frames = block_size, channels, max_frames
        self.bits_per_sample = bits_per_sample
        self.last_frames = 0   # frames of the batch submitted last (sizes the offsets arrays)
        rc = L.flacgpu_create(C.byref(o), bits_per_sample, channels, device, max_frames,
                              C.byref(self._h))
        if rc:
            raise GpuError(rc, "flacgpu_create")

    def close(self):
        if self._h:
            _lib.lib().flacgpu_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _alloc(self, n_frames):
        plans = (FramePlan * n_frames)()
        subs = (SubframePlan * (n_frames * self.channels))()
        res = np.empty((n_frames, self.channels, self.block_size), dtype=np.int32)
        return plans, subs, res

    def analyze(self, pcm, n_frames, last_frame_len, layout=LAYOUT_INTERLEAVED):
        """pcm: host int32 array.  Returns (frame plans, subframe plans, residuals)."""
        self.last_frames = n_frames
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        plans, subs, res = self._alloc(n_frames)
        rc = _lib.lib().flacgpu_analyze(
            self._h, pcm.ctypes.data_as(C.POINTER(C.c_int32)), layout, n_frames, last_frame_len,
            plans, subs, res.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            raise GpuError(rc, "flacgpu_analyze")
        return plans, subs, res

    def analyze_device(self, device_ptr, n_frames, last_frame_len, layout=LAYOUT_INTERLEAVED,
                       stream=None):
        self.last_frames = n_frames
        rc = _lib.lib().flacgpu_analyze_device(self._h, C.c_void_p(device_ptr), layout, n_frames,
                                               last_frame_len, C.c_void_p(stream or 0))
        if rc:
            raise GpuError(rc, "flacgpu_analyze_device")

    def fetch(self, n_frames, want_residuals=True):
        plans, subs, res = self._alloc(n_frames)
        rc = _lib.lib().flacgpu_fetch(
            self._h, plans, subs,
            res.ctypes.data_as(C.POINTER(C.c_int32)) if want_residuals else None)
        if rc:
            raise GpuError(rc, "flacgpu_fetch")
        return plans, subs, res

    TUNE_TWO_RANGES, TUNE_LAG_SPLIT, TUNE_BLOCKING_WAIT, TUNE_COPY_INPUT, TUNE_CHUNK_MSAMPLES = 1, 2, 3, 4, 5

    def set_tuning(self, key, value):
        rc = _lib.lib().flacgpu_set_tuning(self._h, key, value)
        if rc:
            raise GpuError(rc, "flacgpu_set_tuning")

    def set_two_ranges(self, on):
        """Cut big batches of 4096-sample frames into two frame ranges on two HIP streams
        (FLACGPU_TUNE_TWO_RANGES); off by default."""
        self.set_tuning(self.TUNE_TWO_RANGES, 1 if on else 0)

    def encode_device(self, device_ptr, n_frames, last_frame_len, first_frame_number, sample_rate,
                      layout=LAYOUT_INTERLEAVED, stream=None):
        """analyze_device + pack_device in one call (see set_two_ranges)."""
        self.last_frames = n_frames
        rc = _lib.lib().flacgpu_encode_device(self._h, C.c_void_p(device_ptr), layout, n_frames,
                                              last_frame_len, first_frame_number, sample_rate,
                                              C.c_void_p(stream or 0))
        if rc:
            raise GpuError(rc, "flacgpu_encode_device")

    def encode_segments(self, segments, sample_rate):
        """segments: [(host int32 array of whole blocks, first_frame_number), ...] of streams of this context's shape, encoded
        as ONE batch (flacgpu_encode_segments).  Returns (bytes, offsets[total_frames + 1]): frames in segment order."""
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a, _ in segments]
        per = self.block_size * self.channels
        segs = (_lib.Segment * len(arrs))()
        total = 0
        for i, (a, (_, first)) in enumerate(zip(arrs, segments)):
            assert a.size % per == 0 and a.size
            segs[i].pcm, segs[i].n_frames, segs[i].first_frame_number = a.ctypes.data, a.size // per, first
            total += a.size // per
        self.last_frames = total
        off = (C.c_uint64 * (total + 1))()
        tot = C.c_uint64(0)
        cap = total * (per * 4 + 64) + 256
        buf = np.empty(cap, dtype=np.uint8)
        rc = _lib.lib().flacgpu_encode_segments(self._h, segs, len(arrs), sample_rate, C.c_void_p(buf.ctypes.data), cap, off,
                                                C.byref(tot))
        if rc:
            raise GpuError(rc, "flacgpu_encode_segments")
        return buf[: tot.value].tobytes(), [int(v) for v in off]

    def encode_segments_device(self, segments, sample_rate, stream=None):
        """segments: [(device pointer, n_frames, first_frame_number), ...]; asynchronous (fetch_frames() for the result)."""
        segs = (_lib.Segment * len(segments))()
        total = 0
        for i, (ptr, n, first) in enumerate(segments):
            segs[i].pcm, segs[i].n_frames, segs[i].first_frame_number = ptr, n, first
            total += n
        self.last_frames = total
        rc = _lib.lib().flacgpu_encode_segments_device(self._h, segs, len(segments), sample_rate, C.c_void_p(stream or 0))
        if rc:
            raise GpuError(rc, "flacgpu_encode_segments_device")

    def pack_plans(self, pcm, n_frames, last_frame_len, plans, subframes, first_frame_number, sample_rate):
        """flacgpu_pack_plans: device-side frame assembly of caller-supplied decisions.  Returns (bytes, offsets)."""
        a = np.ascontiguousarray(pcm, dtype=np.int32)
        rc = _lib.lib().flacgpu_pack_plans(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), n_frames, last_frame_len, plans,
                                           subframes, first_frame_number, sample_rate)
        if rc:
            raise GpuError(rc, "flacgpu_pack_plans")
        self.last_frames = n_frames
        return self.fetch_frames(n_frames)

    def pack_device(self, first_frame_number, sample_rate, stream=None):
        """Device-side frame assembly of the last analysed batch (bytes stay in HBM)."""
        rc = _lib.lib().flacgpu_pack_device(self._h, first_frame_number, sample_rate,
                                            C.c_void_p(stream or 0))
        if rc:
            raise GpuError(rc, "flacgpu_pack_device")

    def fetch_frames(self, n_frames=None):
        """Returns (bytes, offsets[n_frames+1]) of the frames packed on the device."""
        held = self.last_frames or n_frames
        if n_frames is not None and held != n_frames:
            raise ValueError(f"the context holds a batch of {held} frames, not {n_frames}")
        n_frames = held
        off = (C.c_uint64 * (n_frames + 1))()
        total = C.c_uint64(0)
        L = _lib.lib()
        rc = L.flacgpu_fetch_frames(self._h, None, 0, off, C.byref(total))
        if rc not in (0, -5):
            raise GpuError(rc, "flacgpu_fetch_frames")
        buf = np.empty(total.value, dtype=np.uint8)
        rc = L.flacgpu_fetch_frames(self._h, buf.ctypes.data, buf.size, off, C.byref(total))
        if rc:
            raise GpuError(rc, "flacgpu_fetch_frames")
        return buf.tobytes(), list(off)

    def encode_frames(self, pcm, n_frames, last_frame_len, first_frame_number, sample_rate,
                      layout=LAYOUT_INTERLEAVED):
        """analyze + pack + fetch on host PCM; returns (bytes, offsets)."""
        self.last_frames = n_frames
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        cap = pcm.size * 4 + n_frames * 128 + 1024
        buf = np.empty(cap, dtype=np.uint8)
        off = (C.c_uint64 * (n_frames + 1))()
        total = C.c_uint64(0)
        rc = _lib.lib().flacgpu_encode_frames(
            self._h, pcm.ctypes.data_as(C.POINTER(C.c_int32)), layout, n_frames, last_frame_len,
            first_frame_number, sample_rate, buf.ctypes.data, cap, off, C.byref(total))
        if rc:
            raise GpuError(rc, "flacgpu_encode_frames")
        return buf[: total.value].tobytes(), list(off)

    def encode_frames_pinned(self, pcm, n_frames, last_frame_len, first_frame_number, sample_rate, repeat=1,
                             layout=LAYOUT_INTERLEAVED):
        """flacgpu_encode_frames with BOTH host buffers in pinned memory (flacgpu_host_alloc), which is what a caller
        that cares about the PCIe leg hands over: the copies run at the link's rate instead of through the runtime's
        staging of pageable memory.  `repeat` calls back to back; returns (bytes, offsets, seconds per call)."""
        import time

        self.last_frames = n_frames
        L = _lib.lib()
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        cap = pcm.size * 4 + n_frames * 128 + 1024
        hin = L.flacgpu_host_alloc(pcm.nbytes + 64)
        hout = L.flacgpu_host_alloc(cap)
        if not hin or not hout:
            if hin:
                L.flacgpu_host_free(hin)
            if hout:
                L.flacgpu_host_free(hout)
            raise MemoryError("flacgpu_host_alloc")
        try:
            C.memmove(hin, pcm.ctypes.data, pcm.nbytes)
            off = (C.c_uint64 * (n_frames + 1))()
            total = C.c_uint64(0)
            times = []
            for _ in range(repeat):
                t = time.perf_counter()
                rc = L.flacgpu_encode_frames(self._h, C.cast(hin, C.POINTER(C.c_int32)), layout, n_frames, last_frame_len,
                                             first_frame_number, sample_rate, hout, cap, off, C.byref(total))
                times.append(time.perf_counter() - t)
                if rc:
                    raise GpuError(rc, "flacgpu_encode_frames")
            return C.string_at(hout, total.value), list(off), times
        finally:
            L.flacgpu_host_free(hin)
            L.flacgpu_host_free(hout)

    def encode_packed(self, pcm_le, bytes_per_sample, n_frames, last_frame_len, first_frame_number,
                      sample_rate, pinned=True):
        """The asynchronous host path in one go (flacgpu_encode_packed_async -> frames_ready ->
        fetch_frames_async -> wait): pcm_le = interleaved little-endian samples of bytes_per_sample
        bytes (uint8 array).  Returns (bytes, offsets)."""
        self.last_frames = n_frames
        L = _lib.lib()
        src = np.ascontiguousarray(pcm_le, dtype=np.uint8)
        hin = hout = None
        try:
            if pinned:
                hin = L.flacgpu_host_alloc(src.size + 64)
                C.memmove(hin, src.ctypes.data, src.size)
            rc = L.flacgpu_encode_packed_async(self._h, hin if pinned else src.ctypes.data, bytes_per_sample,
                                               n_frames, last_frame_len, first_frame_number, sample_rate)
            if rc:
                raise GpuError(rc, "flacgpu_encode_packed_async")
            offp = C.POINTER(C.c_uint64)()
            total = C.c_uint64(0)
            rc = L.flacgpu_frames_ready(self._h, C.byref(offp), C.byref(total))
            if rc:
                raise GpuError(rc, "flacgpu_frames_ready")
            off = [offp[i] for i in range(n_frames + 1)]
            if pinned:
                hout = L.flacgpu_host_alloc(total.value + 64)
                dst = hout
            else:
                buf = np.empty(total.value + 64, dtype=np.uint8)
                dst = buf.ctypes.data
            rc = L.flacgpu_fetch_frames_async(self._h, dst, total.value + 64)
            if rc:
                raise GpuError(rc, "flacgpu_fetch_frames_async")
            rc = L.flacgpu_wait(self._h)
            if rc:
                raise GpuError(rc, "flacgpu_wait")
            data = C.string_at(dst, total.value)
            return data, off
        finally:
            if hin:
                L.flacgpu_host_free(hin)
            if hout:
                L.flacgpu_host_free(hout)

    def encode_segments_packed(self, pcm_le, bytes_per_sample, segments, sample_rate):
        """flacgpu_encode_segments_packed_async_host in one go: pcm_le = the segments' whole blocks back to back (uint8 array of
        little-endian samples of bytes_per_sample bytes, or int32 samples viewed as bytes when bytes_per_sample == 4);
        segments = [(n_frames, first_frame_number), ...].  Pinned buffers both ways.  Returns (bytes, offsets)."""
        L = _lib.lib()
        src = np.ascontiguousarray(pcm_le).view(np.uint8).reshape(-1)
        n_frames = sum(n for n, _ in segments)
        self.last_frames = n_frames
        segs = (_lib.Segment * len(segments))()
        for i, (n, first) in enumerate(segments):
            segs[i].pcm = None
            segs[i].n_frames = n
            segs[i].first_frame_number = first
        cap = L.flacgpu_packed_cap(self._h)
        hin = L.flacgpu_host_alloc(src.size + 64)
        hout = L.flacgpu_host_alloc(cap)
        try:
            C.memmove(hin, src.ctypes.data, src.size)
            rc = L.flacgpu_encode_segments_packed_async_host(self._h, hin, bytes_per_sample, segs, len(segments), sample_rate,
                                                             hout, cap)
            if rc:
                raise GpuError(rc, "flacgpu_encode_segments_packed_async_host")
            offp = C.POINTER(C.c_uint64)()
            total = C.c_uint64(0)
            rc = L.flacgpu_frames_ready(self._h, C.byref(offp), C.byref(total))
            if rc:
                raise GpuError(rc, "flacgpu_frames_ready")
            off = [offp[i] for i in range(n_frames + 1)]
            rc = L.flacgpu_fetch_frames_async(self._h, hout, cap)
            if rc:
                raise GpuError(rc, "flacgpu_fetch_frames_async")
            rc = L.flacgpu_wait(self._h)
            if rc:
                raise GpuError(rc, "flacgpu_wait")
            return C.string_at(hout, total.value), off
        finally:
            L.flacgpu_host_free(hin)
            L.flacgpu_host_free(hout)

    def packed_input_supported(self, bytes_per_sample):
        return bool(_lib.lib().flacgpu_packed_input_supported(self._h, bytes_per_sample))

    def verify_device(self, sample_rate, first_frame_number=0):
        """Decode the frames packed last on the GPU, check CRC-16 and compare with the analysed
        PCM.  Returns (VerifyResult, kernel_ms)."""
        res = _lib.VerifyResult()
        ms = C.c_float(0)
        rc = _lib.lib().flacgpu_verify_device(self._h, sample_rate, first_frame_number,
                                              C.byref(res), C.byref(ms))
        if rc:
            raise GpuError(rc, "flacgpu_verify_device")
        return res, ms.value

    def fetch_decoded(self, n_frames, last_frame_len):
        n = ((n_frames - 1) * self.block_size + last_frame_len) * self.channels
        out = np.empty(n, dtype=np.int32)
        rc = _lib.lib().flacgpu_fetch_decoded(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            raise GpuError(rc, "flacgpu_fetch_decoded")
        return out

    def device_buffer(self, which):
        return _lib.lib().flacgpu_device_buffer(self._h, which)

    def experiment_mfma_autocorr(self):
        """EXPERIMENT (not on the product path): f64-MFMA autocorrelation of the last batch.
        Returns dict(ms, compared, params_differ, max_rel_err)."""
        ms, cmp_, diff, err = C.c_float(0), C.c_uint32(0), C.c_uint32(0), C.c_double(0)
        rc = _lib.lib().flacgpu_experiment_mfma_autocorr(self._h, C.byref(ms), C.byref(cmp_),
                                                         C.byref(diff), C.byref(err))
        if rc:
            raise GpuError(rc, "flacgpu_experiment_mfma_autocorr")
        return {"ms": ms.value, "compared": cmp_.value, "params_differ": diff.value,
                "max_rel_err": err.value}

    def stats(self):
        s = GpuStats()
        rc = _lib.lib().flacgpu_get_stats(self._h, C.byref(s))
        if rc:
            raise GpuError(rc, "flacgpu_get_stats")
        return s

    def handed_subframes(self):
        """(handed, subframes, enabled) of the last batch: subframes whose residual the candidate kernel handed to the frame
        kernel (flacgpu_handed_subframes)"""
        L = _lib.lib()
        h, n, e = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
        L.flacgpu_handed_subframes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        rc = L.flacgpu_handed_subframes(self._h, C.byref(h), C.byref(n), C.byref(e))
        if rc:
            raise GpuError(rc, "flacgpu_handed_subframes")
        return h.value, n.value, bool(e.value)

    def set_timing(self, on=True):
        _lib.lib().flacgpu_set_timing(self._h, int(on))

    def kernel_ms(self):
        arr = (C.c_float * _lib.N_KERNELS)()
        _lib.lib().flacgpu_get_kernel_ms(self._h, C.byref(arr))
        L = _lib.lib()
        return {L.flacgpu_kernel_name(i).decode(): float(arr[i]) for i in range(_lib.N_KERNELS)
                if arr[i] > 0}


def decode_stream(data, device=-1, want_pcm=True):
    """Stand-alone GPU decode + verify of a whole FLAC stream (flacgpu_decode_stream): returns
    (interleaved int32 PCM or None, StreamInfo)."""
    L = _lib.lib()
    info = _lib.StreamInfo()
    data = bytes(data)
    rc = L.flacgpu_decode_stream(data, len(data), device, None, 0, C.byref(info))   # sizes + verification
    if rc:
        raise GpuError(rc, "flacgpu_decode_stream")
    if not want_pcm:
        return None, info
    out = np.empty(info.decoded_samples * info.channels, dtype=np.int32)
    rc = L.flacgpu_decode_stream(data, len(data), device, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size,
                                 C.byref(info))
    if rc:
        raise GpuError(rc, "flacgpu_decode_stream")
    return out, info


def host_pack_frames(sample_rate, bits_per_sample, channels, first_frame_number, n_frames,
                     row_stride, plans, subs, rows, threads=1):
    """Host bit-packing of an analysed batch (flacenc_pack_frames); returns (bytes, offsets)."""
    L = _lib.lib()
    off = (C.c_uint64 * (n_frames + 1))()
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    args = (sample_rate, bits_per_sample, channels, first_frame_number, n_frames, row_stride,
            C.cast(plans, C.c_void_p), C.cast(subs, C.c_void_p),
            rows.ctypes.data_as(C.POINTER(C.c_int32)), threads)
    rc = L.flacenc_pack_frames(*args, None, 0, off)
    if rc not in (0, -140):
        raise RuntimeError(f"flacenc_pack_frames: {rc}")
    buf = np.empty(off[n_frames], dtype=np.uint8)
    rc = L.flacenc_pack_frames(*args, buf.ctypes.data, buf.size, off)
    if rc:
        raise RuntimeError(f"flacenc_pack_frames: {rc}")
    return buf.tobytes(), list(off)


class PinnedBuffer:
    """Pinned host memory from flacgpu_host_alloc, viewed as a numpy uint8 array (`.array`)."""

    def __init__(self, nbytes):
        self._p = _lib.lib().flacgpu_host_alloc(nbytes)
        if not self._p:
            raise MemoryError("flacgpu_host_alloc")
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    @property
    def address(self):
        return self._p

    def close(self):
        if self._p:
            self.array = None
            _lib.lib().flacgpu_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """flacgpu_pipeline_* (include/flacenc_gpu.h): `depth` encoder contexts taking consecutive batches in rotation --
    upload of batch n, kernels of n - 1 and the frames of n - 2 in flight together.  PCM comes from PINNED buffers
    (PinnedBuffer) and must stay untouched until its batch has been retired."""

    def __init__(self, block_size, max_partition_order, max_lpc_order, mid_side, exhaustive, window_kind, window_param,
                 bits_per_sample, channels, max_frames, depth=4, device=-1):
        L = _lib.lib()
        o = GpuOptions(block_size, max_partition_order, max_lpc_order or 0, int(bool(mid_side)),
                       int(bool(exhaustive)), window_kind, 0, window_param)
        self._h = C.c_void_p(None)
        rc = L.flacgpu_pipeline_create(C.byref(o), bits_per_sample, channels, device, max_frames, depth, C.byref(self._h))
        if rc:
            raise GpuError(rc, "flacgpu_pipeline_create")
        self.depth = depth

    def in_flight(self):
        return _lib.lib().flacgpu_pipeline_in_flight(self._h)

    def submit(self, pinned_address, bytes_per_sample, n_frames, last_frame_len, first_frame_number, sample_rate):
        """Returns False (nothing queued) when every slot holds a batch: retire() first."""
        rc = _lib.lib().flacgpu_pipeline_submit(self._h, C.c_void_p(pinned_address), bytes_per_sample, n_frames,
                                                last_frame_len, first_frame_number, sample_rate)
        if rc == -6:
            return False
        if rc:
            raise GpuError(rc, "flacgpu_pipeline_submit")
        return True

    def retire(self, copy=True):
        """The oldest batch in flight: (frame bytes, offsets[n_frames + 1]); with copy=False a (address, total,
        offsets pointer, n_frames) tuple valid until the next submit."""
        frames = C.c_void_p(None)
        off = C.POINTER(C.c_uint64)()
        n = C.c_uint32(0)
        total = C.c_uint64(0)
        rc = _lib.lib().flacgpu_pipeline_retire(self._h, C.byref(frames), C.byref(off), C.byref(n), C.byref(total))
        if rc:
            raise GpuError(rc, "flacgpu_pipeline_retire")
        if not copy:
            return frames.value, total.value, off, n.value
        return C.string_at(frames.value, total.value), [off[i] for i in range(n.value + 1)]

    def close(self):
        if self._h:
            _lib.lib().flacgpu_pipeline_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_counters_c(per_shard):
    """flacgpu_merge_counters on [[frames, bytes, min_frame, max_frame], ...]: (merged record, shard byte offsets)."""
    n = len(per_shard)
    arr = (_lib.ShardCounters * n)(*[_lib.ShardCounters(*[int(v) for v in c]) for c in per_shard])
    merged = _lib.ShardCounters()
    offs = (C.c_uint64 * n)()
    rc = _lib.lib().flacgpu_merge_counters(arr, n, C.byref(merged), offs)
    if rc:
        raise GpuError(rc, "flacgpu_merge_counters")
    return merged.as_list(), [int(v) for v in offs]


def shard_range_c(total_frames, shards, shard):
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    _lib.lib().flacgpu_shard_range(total_frames, shards, shard, C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


class MultiDevice:
    """flacgpu_multi_* (include/flacenc_gpu.h "several GPUs"): one process, one shard per listed device, batches of
    contiguous frames of one stream dealt to the shards in turn, the four-integer records merged on the host.  `devices=None`: every visible device;
    an ordinal may be listed more than once (a 1-GPU box then exercises the whole multi-shard path)."""

    def __init__(self, block_size, max_partition_order, max_lpc_order, mid_side, exhaustive, window_kind, window_param,
                 bits_per_sample, channels, max_frames, devices=None, depth=2):
        L = _lib.lib()
        o = GpuOptions(block_size, max_partition_order, max_lpc_order or 0, int(bool(mid_side)),
                       int(bool(exhaustive)), window_kind, 0, window_param)
        self._h = C.c_void_p(None)
        devs = (C.c_int * len(devices))(*devices) if devices else None
        rc = L.flacgpu_multi_create(C.byref(o), bits_per_sample, channels, devs, len(devices) if devices else 0,
                                    max_frames, depth, C.byref(self._h))
        if rc:
            raise GpuError(rc, "flacgpu_multi_create")
        self.shards = L.flacgpu_multi_shards(self._h)
        self.devices = [L.flacgpu_multi_device_of(self._h, k) for k in range(self.shards)]
        self.block_size, self.channels, self.bits_per_sample = block_size, channels, bits_per_sample

    def encode(self, pcm, n_frames, last_frame_len, first_frame_number, sample_rate, bytes_per_sample=4):
        """Host PCM of one stream's run of blocks (int32 array, or a uint8 array of packed little-endian samples) ->
        (frame bytes, offsets[n_frames + 1], per-shard records, merged record)."""
        L = _lib.lib()
        pcm = np.ascontiguousarray(pcm, dtype=np.int32 if bytes_per_sample == 4 else np.uint8)
        off = (C.c_uint64 * (n_frames + 1))()
        total = C.c_uint64(0)
        per = (_lib.ShardCounters * self.shards)()
        merged = _lib.ShardCounters()
        args = (self._h, C.c_void_p(pcm.ctypes.data), bytes_per_sample, n_frames, last_frame_len, first_frame_number,
                sample_rate)
        cap = n_frames * (self.block_size * self.channels * ((self.bits_per_sample + 7) // 8 + 1) + 64)
        buf = np.empty(cap, dtype=np.uint8)
        rc = L.flacgpu_multi_encode(*args, C.c_void_p(buf.ctypes.data), cap, off, C.byref(total), per, C.byref(merged))
        if rc:
            raise GpuError(rc, "flacgpu_multi_encode")
        return (buf[: total.value].tobytes(), [int(v) for v in off], [c.as_list() for c in per], merged.as_list())

    def encode_raw(self, address, n_frames, last_frame_len, first_frame_number, sample_rate, bytes_per_sample=4):
        """flacgpu_multi_encode on a caller-held (pinned) buffer, frames into a reused output array: the timed form."""
        L = _lib.lib()
        cap = n_frames * (self.block_size * self.channels * ((self.bits_per_sample + 7) // 8 + 1) + 64)
        if getattr(self, "_out", None) is None or self._out.size < cap:
            self._out = np.empty(cap, dtype=np.uint8)
            self._out[:] = 0      # touched: page faults are not the encoder's
        total = C.c_uint64(0)
        rc = L.flacgpu_multi_encode(self._h, C.c_void_p(address), bytes_per_sample, n_frames, last_frame_len, first_frame_number,
                                    sample_rate, C.c_void_p(self._out.ctypes.data), cap, None, C.byref(total), None, None)
        if rc:
            raise GpuError(rc, "flacgpu_multi_encode")
        return total.value

    def host_copy_stats(self):
        """(bytes handed out in `out`, bytes the host copied to put them there, shard threads bound near their GPU)"""
        a, b, n = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        rc = _lib.lib().flacgpu_multi_host_copy_stats(self._h, C.byref(a), C.byref(b), C.byref(n))
        if rc:
            raise GpuError(rc, "flacgpu_multi_host_copy_stats")
        return a.value, b.value, n.value

    @staticmethod
    def numa_info(device):
        """(NUMA node of the device's PCI function or -1, its local CPU list as sysfs spells it or '')"""
        node = C.c_int(-1)
        buf = C.create_string_buffer(1024)
        rc = _lib.lib().flacgpu_device_numa_info(device, C.byref(node), buf, 1024)
        if rc:
            raise GpuError(rc, "flacgpu_device_numa_info")
        return node.value, buf.value.decode()

    def encode_device(self, shard, device_ptr, n_frames, last_frame_len, first_frame_number, sample_rate,
                      layout=LAYOUT_INTERLEAVED):
        rc = _lib.lib().flacgpu_multi_encode_device(self._h, shard, C.c_void_p(device_ptr), layout, n_frames,
                                                    last_frame_len, first_frame_number, sample_rate)
        if rc:
            raise GpuError(rc, "flacgpu_multi_encode_device")

    def wait(self):
        rc = _lib.lib().flacgpu_multi_wait(self._h)
        if rc:
            raise GpuError(rc, "flacgpu_multi_wait")

    def counters(self):
        per = (_lib.ShardCounters * self.shards)()
        merged = _lib.ShardCounters()
        rc = _lib.lib().flacgpu_multi_counters(self._h, per, C.byref(merged))
        if rc:
            raise GpuError(rc, "flacgpu_multi_counters")
        return [c.as_list() for c in per], merged.as_list()

    def fetch_last(self, shard, n_frames):
        """Frame bytes and offsets of shard `shard`'s last resident batch."""
        h = _lib.lib().flacgpu_multi_last_context(self._h, shard)
        off = (C.c_uint64 * (n_frames + 1))()
        total = C.c_uint64(0)
        rc = _lib.lib().flacgpu_fetch_frames(C.c_void_p(h), None, 0, off, C.byref(total))
        if rc not in (0, -5):
            raise GpuError(rc, "flacgpu_fetch_frames")
        buf = np.empty(total.value, dtype=np.uint8)
        rc = _lib.lib().flacgpu_fetch_frames(C.c_void_p(h), C.c_void_p(buf.ctypes.data), buf.size, off, C.byref(total))
        if rc:
            raise GpuError(rc, "flacgpu_fetch_frames")
        return buf.tobytes(), [int(v) for v in off]

    def close(self):
        if self._h:
            _lib.lib().flacgpu_multi_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_allgather_counters(nccl_comm, local, cap_ranks=64, stream=None):
    """flacgpu_rccl_allgather_counters: every rank's [frames, bytes, min_frame, max_frame] over the caller's RCCL
    communicator (an ncclComm_t as an integer / c_void_p); returns (records in rank order, this rank)."""
    mine = _lib.ShardCounters(*[int(v) for v in local])
    out = (_lib.ShardCounters * cap_ranks)()
    n, me = C.c_uint32(0), C.c_uint32(0)
    rc = _lib.lib().flacgpu_rccl_allgather_counters(C.c_void_p(nccl_comm), C.c_void_p(stream or 0), C.byref(mine), out,
                                                    cap_ranks, C.byref(n), C.byref(me))
    if rc:
        raise GpuError(rc, "flacgpu_rccl_allgather_counters")
    return [out[i].as_list() for i in range(n.value)], int(me.value)
