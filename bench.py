#!/usr/bin/env python3
"""bench.py -- FLAC encode hot path on MI355X.

Metric (BASELINE.json): Msamples/s encode at "level 8" (= reference Options::best():
block 4096, LPC order <= 12, partition order <= 6, mid-side, exhaustive channel
correlation), 48 kHz / 24-bit stereo, output bit-exact vs the reference restatement.

A "step" is one pass of the hot path over one batch of synthetic PCM that is already
resident in HBM (interleaved int32, the layout FlacSampleWriter::write receives):
de-interleave -> fixed/LPC analysis -> channel-assignment decision -> residuals -> frame
assembly (headers, Rice bit-packing, CRC-8/16) -> finished FLAC frame bytes in HBM.
Stream bookkeeping that the reference keeps per stream on the host (MD5 of the PCM,
metadata rewrite) is outside the step; the `end_to_end` block of the output line measures the
whole host PCM -> .flac bytes path (PCIe, MD5, container included) separately.

Usage:  python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}]
  N > 1 without WORLD_SIZE in the environment: this process starts N ranks of itself (one per
  GPU, before it touches any GPU) and relays rank 0's line; under torchrun the ranks are taken
  from the environment and --gpus must equal WORLD_SIZE.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

BLOCK = 4096
FRAMES = 8192          # 8192 blocks x 4096 per GPU
DISTINCT = 512         # distinct synthetic frames, tiled to FRAMES (all of them are oracle-checked)

# SURVEY.md section 0 (level mapping) + 8(d) (configs -> inputs)
CONFIGS = {
    2: dict(rate=48000, bps=16, ch=2, lpc=0, po=5, level="L5-fixed",
            text="48 kHz/16-bit stereo, blocksize 4096, level 5 with fixed predictors only "
                 "(Options::default().max_lpc_order(None): partition order 5, mid-side, exhaustive)"),
    3: dict(rate=48000, bps=24, ch=2, lpc=12, po=6, level="L8",
            text="48 kHz/24-bit stereo, blocksize 4096, level 8 (Options::best: LPC order 12, "
                 "partition order 6, mid-side, exhaustive)"),
    4: dict(rate=192000, bps=24, ch=8, lpc=12, po=6, level="L8",
            text="192 kHz/24-bit 8-channel, blocksize 4096, level 8 (Options::best), contiguous frame "
                 "ranges per GPU"),
    5: dict(rate=96000, bps=24, ch=2, lpc=32, po=6, level="L8x",
            text="96 kHz/24-bit stereo, blocksize 4096, level 8 exhaustive (Options::best + LPC order "
                 "32, partition order 6)"),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md
F64_PEAK_GFLOPS = 78600.0   # f64 vector peak as stated (FMA = 2 flop); separately rounded mul + add: half of it
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4   # wave64 instructions / s at the nominal clock


HI_SECTIONS = {3: 6, 4: 6, 5: 16}    # resonator sections of the high-order input per config (AR(2 x sections))


def make_pcm(seed, frames, channels, bps, signal="ar2", sections=6):
    """signal "ar2": SURVEY.md 8(d)'s generator (one 2-pole resonator: the encoder answers with LPC order 2);
    "hi": tests/_pcm.py synth_hi, cascaded resonators whose model order changes every 8 frames (orders 8-12 with 6
    sections, 22-32 with 16) together with the relation between the channels.
    The DISTINCT base frames are generated once per box and configuration and kept under /dev/shm (the ranks of a
    multi-GPU run all use the same few buffers: whoever comes first writes the file, the others read it)."""
    from _pcm import synth_fast, synth_hi

    n = BLOCK * min(DISTINCT, frames)
    key = f"flacbench_{signal}_{seed}_{channels}_{bps}_{n}_{sections}.npy"
    shm = os.environ.get("FLAC_BENCH_CACHE", "/dev/shm")
    path = os.path.join(shm, key) if os.path.isdir(shm) and os.access(shm, os.W_OK) else None
    base = None
    if path and os.path.exists(path):
        try:
            base = np.load(path)
            if base.size != n * channels:
                base = None
        except Exception:
            base = None
    if base is None:
        base = (synth_hi(seed, channels, bps, n, sections=sections) if signal == "hi"
                else synth_fast(seed, channels, bps, n))
        if path:
            try:   # written under a private name, then renamed: a reader never sees half a file
                tmp = f"{path}.{os.getpid()}.tmp"
                with open(tmp, "wb") as fh:
                    np.save(fh, base)
                os.replace(tmp, path)
            except OSError:
                pass
    reps = (frames + DISTINCT - 1) // DISTINCT
    return np.tile(base, reps)[: frames * BLOCK * channels]


class GpuSensors:
    """The shader clock and the socket power of the GPU this rank runs on, read from its hwmon files while a timed loop runs
    (amdgpu: freq1_input in Hz, power1_input in uW, power1_cap).  The boxes of the pool do not all clock alike -- one met in r06
    holds 2.13-2.16 GHz under this load where the others hold 2.3-2.4 -- and `value` moves with it."""

    def __init__(self, torch, device):
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(device)
            want = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            import glob
            for d in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % want):
                if os.path.exists(os.path.join(d, "freq1_input")):
                    self.dir = d
        except Exception:
            self.dir = None
        self.samples = []
        self._stop = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def start(self, period=0.004):
        if not self.dir:
            return
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                f, pw = self._read("freq1_input"), self._read("power1_input")
                if f and pw:
                    self.samples.append((f / 1e6, pw / 1e6))
                self._stop.wait(period)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        if not self._stop:
            return None
        self._stop.set()
        self._th.join()
        if not self.samples:
            return None
        fs = [a for a, _ in self.samples]
        ps = [b for _, b in self.samples]
        cap = self._read("power1_cap")
        return {"sclk_MHz": round(statistics.median(fs)), "sclk_MHz_min_max": [round(min(fs)), round(max(fs))],
                "socket_power_W": round(statistics.median(ps)), "power_cap_W": round(cap / 1e6) if cap else None,
                "samples": len(fs), "source": "hwmon freq1_input / power1_input of the rank's GPU during the sustained loop"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_gpus_without_hip():
    """GPUs this process may use, WITHOUT loading the HIP runtime (on this pool a process that has touched the
    GPU must not be the one whose children exec; torch.cuda.device_count() can fall back to hipGetDeviceCount):
    the KFD topology (nodes with SIMDs are GPUs) narrowed by the visibility variables."""
    import glob

    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(props):
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                    n += 1
        except OSError:
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, popen=subprocess.Popen):
    """--gpus N without torchrun: N children of this script, one per GPU.  Nothing in this (parent) process
    loads torch or the HIP runtime (tests/test_bench_launch.py checks /proc/self/maps at spawn time)."""
    have = count_gpus_without_hip()
    if have < n and os.environ.get("FLAC_BENCH_SHARE_DEVICE") != "1":
        sys.stderr.write(f"bench.py: --gpus {n} asked for, but this box has {have} GPU(s); "
                         f"refusing to report a line for fewer ranks than requested\n")
        sys.exit(2)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def orc_options(orc, cfg):
    oo = orc.options("best" if cfg["lpc"] >= 12 else "default")
    return oo.copy(max_lpc_order=cfg["lpc"], max_partition_order=cfg["po"])


def cpu_baseline(orc, cfg, pcm, frames_total):
    """The oracle (C restatement of the reference, NOT the Rust binary) on this box's host cores:
    median of 5 runs after one warm-up, each on a bounded sample of the same workload."""
    C = cfg["ch"]
    oo = orc_options(orc, cfg)

    def run(frames, threads):
        sample = pcm[: frames * BLOCK * C]
        times = []
        ref = None
        for i in range(6):
            t = time.perf_counter()
            rc, out, _ = orc.encode_stream(oo, cfg["rate"], cfg["bps"], C, sample, total_known=True,
                                           threads=threads)
            dt = time.perf_counter() - t
            assert rc == 0
            if ref is None:
                ref = out
            assert out == ref
            if i:
                times.append(dt)
        return sample.size / statistics.median(times) / 1e6, frames

    per_frame = BLOCK * C
    f1 = max(64, min(frames_total, int(2048 * 8192 / per_frame)))
    fn = max(64, min(frames_total, int(4096 * 8192 / per_frame)))
    ncores = os.cpu_count() or 1
    v1, n1 = run(f1, 1)
    fj_threads = 4 if C <= 2 else min(16, 2 * C)
    vf, nf = run(f1, -fj_threads)
    v4, n4 = run(fn, 4)
    va, na = run(fn, ncores)
    return {"value": round(v1, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"first {n1} frames of the bench batch, full encode to an in-memory .flac "
                      f"(MD5 + analysis + bit-pack + CRC), 1 thread, median of 5 after 1 warm-up",
            "reference_fork_join": {"value": round(vf, 3), "cores": fj_threads, "frames": nf,
                                    "note": "the reference's own task structure (encode.rs:2690-2745, 2906-2928, "
                                            "3964-4010): frames strictly sequential, per frame L || R then M || S, "
                                            "each subframe FIXED || LPC (<= 4 concurrent tasks for stereo, <= 16 for "
                                            "8 channels), on a small work-helping pool"},
            "four_threads": {"value": round(v4, 3), "cores": 4, "frames": n4,
                             "note": "frame-parallel restatement on 4 threads (not what the reference does): an "
                                     "upper bound of what its fork-join can reach with 4 cores"},
            "all_cores_frame_parallel": {"value": round(va, 3), "cores": ncores, "frames": na,
                                         "note": "not something the reference does"}}


def host_capacity(bytes_per_sample):
    """What the host side of this box can do: logical CPUs, the cgroup CPU quota, and the aggregate
    rate of concurrent MD5 chains (hashlib).  Every stream's MD5 is a serial chain on one host core
    (encode.rs:571, 1292-1318), so the many-stream end-to-end rate cannot exceed this aggregate."""
    import hashlib

    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    ncpu = len(os.sched_getaffinity(0))
    eff = min(ncpu, quota) if quota else ncpu
    n = max(1, min(int(eff), 64))
    buf = os.urandom(24 << 20)
    done = [0.0] * n

    def work(i):
        hashlib.md5(buf).digest()
        done[i] = time.perf_counter()

    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t
    return {"logical_cpus": ncpu, "cgroup_cpu_quota": quota, "usable_cpus": eff,
            "md5_aggregate_Msamples/s": round(n * len(buf) / bytes_per_sample / dt / 1e6, 1),
            "md5_threads": n,
            "note": "aggregate of concurrent SCALAR MD5 chains (hashlib), one per usable CPU: what one-thread-per-stream "
                    "hashing can reach on this box; the writers' shared 16-lane AVX-512 engines (host/md5_mb.cpp) are "
                    "not bound by it"}


def end_to_end(cfg, pcm, device, orc, batch_frames=0):
    """Host-resident PCM -> finished .flac bytes in host memory through the public writer surface
    (FlacSampleWriter::new / write / finalize, encode.rs:487-627): H2D, kernels, D2H, MD5, seek
    table and metadata rewrite included.  One stream, then many concurrent streams sharing the GPU."""
    from flac_codec_amd.encode import FlacSampleWriter, Options
    from flac_codec_amd.gpu import GpuAnalyzer

    C, bps, rate = cfg["ch"], cfg["bps"], cfg["rate"]

    def opts():
        o = Options.best() if cfg["lpc"] >= 12 else Options.default()
        o = o.max_lpc_order(cfg["lpc"] or None).max_partition_order(cfg["po"]).device(device)
        if batch_frames:
            o = o.batch_frames(batch_frames)
        return o

    def encode(samples):
        """returns (.flac bytes, stats, seconds from FlacSampleWriter::new to the end of finalize: the
        stream then sits complete in the writer's host memory; copying it into a Python bytes object is
        outside the time)"""
        t = time.perf_counter()
        w = FlacSampleWriter(None, opts(), rate, bps, C, samples.size)
        w.write(samples)
        w.finalize()
        dt = time.perf_counter() - t
        data = w.getvalue()
        st = w.stats()
        w.close()
        return data, st, dt

    out = {}
    import torch

    # host PCM -> frames in host memory with both link directions in flight (flacgpu_pipeline_*), two batch sizes; measured
    # FIRST: the writers below leave a pool of 64 lanes (128 HIP streams) and parked threads behind, beside which this loop
    # ran 5-10 % slower
    # (64 batches per timed loop: with 16 the loop's ramp -- the first uploads with nothing to overlap -- and its drain were
    # 6-10 % of it, the "loss inside the bench" of VERDICT r04 item 9: bench.pipelined_pcie from a bare script gives 11.35 /
    # 13.5 Gsamples/s with 16 batches and 12.67 / 14.9 with 64, tools/pp_bench_fn.py)
    # (r06: 1024-frame batches six deep -- finer interleaving of upload, kernels and frame stores: 13.7 / 16.9 Gsamples/s against
    # 12.6 / 13.9 with 2048-frame batches four deep on one box, tools/pp_scan.py, profiles/r06_pp_scan.json)
    out["pipelined_pcie"] = pipelined_pcie(torch, cfg, pcm, device, orc, 1024, depth=6, batches=128)
    out["pipelined_pcie"]["batch_8192"] = {k: v for k, v in pipelined_pcie(torch, cfg, pcm, device, orc, FRAMES, depth=3,
                                                                            batches=16).items() if k not in ("link", "note")}
    out["host"] = host_capacity((bps + 7) // 8)   # (after the pipelined leg: its 16 hashing threads use up the CPU quota of the period)
    # one stream: 2048 blocks (~3 minutes of 48 kHz audio)
    one = pcm[: 2048 * BLOCK * C]
    encode(one[: 64 * BLOCK * C])           # warm-up (context creation, staging buffers)
    times = []
    st = None
    for _ in range(5):
        _, st, dt = encode(one)
        times.append(dt)
    rc, ref, _ = orc.encode_stream(orc_options(orc, cfg), rate, bps, C, one[: 256 * BLOCK * C],
                                   total_known=True)
    small = encode(one[: 256 * BLOCK * C])[0]
    assert rc == 0 and small == ref, "end-to-end stream differs from the oracle's .flac"
    dt = statistics.median(times)
    md5_rate = one.size / (st.md5_ms * 1e-3) / 1e6 if st.md5_ms > 0 else None
    out["one_stream"] = {"Msamples/s": round(one.size / dt / 1e6, 1), "frames": 2048,
                         "md5_thread_Msamples/s": round(md5_rate, 1) if md5_rate else None,
                         "byte_identical_to_oracle": True,
                         "note": "median of 5; the stream MD5 is a serial chain on one host thread "
                                 "(encode.rs:571, 1292-1318) and caps a single stream"}
    # ---- many streams, host PCM -> .flac bytes in caller buffers, MD5 included: the two C++ front ends.  Timed: the C entry
    # point alone on a prepared job array (what a C / Rust caller holds); a pause of one quota period between calls lets the cgroup's
    # CPU quota recover (a call of a dozen busy threads uses half of a 100 ms period's 16 CPUs; back to back they meet the throttle).
    from flac_codec_amd.encode import BatchEncoder

    usable = int(out["host"]["usable_cpus"])
    link = out["pipelined_pcie"]["link"]
    width = (bps + 7) // 8
    link_limit = link["h2d_GB/s"] / width * 1e3     # Msamples/s the upload direction alone allows at the stream's width

    def windows(n, f):
        """n streams of f blocks: windows into the bench's PCM (distinct where it is long enough, overlapping otherwise)"""
        total = pcm.size // (BLOCK * C)
        step = max(1, (total - f) // max(1, n - 1)) if n > 1 else 0
        return [pcm[i * step * BLOCK * C: (i * step + f) * BLOCK * C] for i in range(n)]

    def front_end(streams, coalesce, threads, reps):
        enc = BatchEncoder(opts(), threads=threads, coalesce=coalesce)
        h = enc.prepare(streams, rate, bps, C)
        enc.run(h)                                   # warm-up: contexts, pinned staging, output buffers
        got = [bytes(v) for v in enc.results(h, copy=False)]
        ts = []
        for _ in range(reps):
            time.sleep(0.12)   # (a period of the cgroup's CPU quota: every call starts with a full one)
            t = time.perf_counter()
            enc.run(h)
            ts.append(time.perf_counter() - t)
        return got, statistics.median(ts), min(ts)

    def front_end_pair(streams, threads, reps, rounds=3):
        """Both front ends on the same streams, their calls in alternating rounds (the host legs move by a quarter from minute to
        minute on a shared host: a comparison needs both sides under the same weather); medians over all rounds."""
        encs = [BatchEncoder(opts(), threads=0, coalesce=True), BatchEncoder(opts(), threads=threads, coalesce=False)]
        hs = [e.prepare(streams, rate, bps, C) for e in encs]
        gots = []
        for e, h in zip(encs, hs):
            e.run(h)
            gots.append([bytes(v) for v in e.results(h, copy=False)])
        assert gots[0] == gots[1], "the coalescing front end and the per-stream writers disagree"
        ts = ([], [])
        for _ in range(rounds):
            for k in (0, 1):
                for _ in range(reps[k]):
                    time.sleep(0.12)
                    t = time.perf_counter()
                    encs[k].run(hs[k])
                    ts[k].append(time.perf_counter() - t)
        return statistics.median(ts[0]), statistics.median(ts[1])

    def rate_of(n, f, dt):
        return round(n * f * BLOCK * C / dt / 1e6, 1)

    # (the coalescing front end's legs FIRST: the per-stream writers keep dozens of threads and pooled lanes behind)
    n_small, f_small = 1024, 8
    smalls = windows(n_small, f_small)
    small_co, small_co_dt, small_co_best = front_end(smalls, True, 0, 7)
    # 64 streams of 512 blocks each
    n_streams, f_streams = 64, 512
    streams = windows(n_streams, f_streams)
    n_threads = min(n_streams, 16)   # per-stream writers: they mostly wait; more threads than the CPU quota meet the throttle
    co, co_dt, co_best = front_end(streams, True, 0, 5)
    wide = windows(256, 128)
    _, wide_dt, _ = front_end(wide, True, 0, 5)
    pw, pw_dt, pw_best = front_end(streams, False, n_threads, 5)
    assert co == pw, "the coalescing front end and the per-stream writers disagree"
    for i in (0, n_streams - 1):
        rc, ref, _ = orc.encode_stream(orc_options(orc, cfg), rate, bps, C, streams[i], total_known=True)
        assert rc == 0 and co[i] == ref, "a batch-encoded stream differs from the oracle's .flac"
    lane_gbs = md5_lane_rate()
    chain_bound = n_streams * lane_gbs / width * 1e3 if lane_gbs else None
    out["many_streams"] = {
        "Msamples/s": rate_of(n_streams, f_streams, co_dt), "best_Msamples/s": rate_of(n_streams, f_streams, co_best),
        "per_stream_writers_Msamples/s": rate_of(n_streams, f_streams, pw_dt), "streams": n_streams, "frames_per_stream": f_streams,
        "frac_of_link": round(rate_of(n_streams, f_streams, co_dt) / link_limit, 4),
        "md5_chain_bound_Msamples/s": round(chain_bound, 1) if chain_bound else None,
        "md5_GB/s_per_chain": lane_gbs,
        "same_volume_256_streams_Msamples/s": rate_of(256, 128, wide_dt),
        "same_volume_256_streams_frac_of_link": round(rate_of(256, 128, wide_dt) / link_limit, 4),
        "host_cores": os.cpu_count(), "usable_cpus": usable, "per_stream_writer_threads": n_threads, "byte_identical_to_oracle": True,
        "note": "median of 5 calls (C entry point on a prepared job array, 120 ms between calls) of flacenc_encode_many_coalesced -- "
                "all streams' frames through one ring of pinned staging buffers, stream-width uploads, csrc/host/coalesce.cpp -- and of "
                "flacenc_encode_many (a writer per stream); host PCM -> .flac bytes in caller buffers, MD5 included.  A stream's "
                "MD5 is ONE serial chain (encode.rs:571, 1292-1318) whose speed is the latency of its dependent steps "
                "(md5_GB/s_per_chain, measured here): 64 streams cannot be hashed faster than md5_chain_bound whatever the "
                "engines' width; the same volume as 256 streams is not bound by it"}
    # ---- many SMALL streams: 1024 streams of 8 blocks each (0.7 s of audio), and the sweep over stream lengths (8192 blocks
    # in all): shared batches against one writer per stream, bytes compared
    co, co_dt, co_best = small_co, small_co_dt, small_co_best
    pw, pw_dt, _ = front_end(smalls, False, n_threads, 3)
    rc, ref, _ = orc.encode_stream(orc_options(orc, cfg), rate, bps, C, smalls[5], total_known=True)
    assert rc == 0 and co[5] == ref and co == pw
    sweep = []
    for f in (1, 2, 4, 16, 32, 64, 128, 256, 512):
        n = 8192 // f
        ss = windows(n, f)
        # (rows where a few dozen MD5 chains bound both front ends alike get more rounds: their difference is the run's noise)
        c_dt, w_dt = front_end_pair(ss, n_threads, (3, 1 if f < 16 else 2), rounds=5 if f >= 256 else 3)
        sweep.append({"streams": n, "blocks": f, "coalesced_Msamples/s": rate_of(n, f, c_dt), "per_stream_writers_Msamples/s": rate_of(n, f, w_dt)})
    for n, f, ss in ((n_small, f_small, smalls), (n_streams, f_streams, streams)):
        c_dt, w_dt = front_end_pair(ss, n_threads, (3, 2), rounds=5 if f >= 256 else 3)
        sweep.append({"streams": n, "blocks": f, "coalesced_Msamples/s": rate_of(n, f, c_dt), "per_stream_writers_Msamples/s": rate_of(n, f, w_dt)})
    sweep.sort(key=lambda r: r["blocks"])
    out["many_small_streams"] = {
        "streams": n_small, "frames_per_stream": f_small,
        "coalesced_Msamples/s": rate_of(n_small, f_small, co_dt), "coalesced_best_Msamples/s": rate_of(n_small, f_small, co_best),
        "one_writer_per_stream_Msamples/s": rate_of(n_small, f_small, pw_dt),
        "coalesced_frac_of_link": round(rate_of(n_small, f_small, co_dt) / link_limit, 4),
        "stream_length_sweep": sweep,
        "coalesced_wins_at_every_length": all(r["coalesced_Msamples/s"] >= r["per_stream_writers_Msamples/s"] for r in sweep),
        # (where a few dozen long streams' MD5 chains bound both front ends -- 16 x 512, 32 x 256, 64 x 512 -- the two are the
        # same speed and the strict comparison is decided by the run's noise)
        "coalesced_at_least_0.97_of_the_writers_at_every_length":
            all(r["coalesced_Msamples/s"] >= 0.97 * r["per_stream_writers_Msamples/s"] for r in sweep),
        "byte_identical": True,
        "note": "flacenc_encode_many_coalesced against flacenc_encode_many, host PCM -> .flac bytes, MD5 included, bytes "
                "compared at every length (one stream with the oracle); the sweep's rows: both front ends' calls in three alternating "
                "rounds on the same streams (five for the rows of 256 and 512 blocks), medians over the rounds; r05's coalescing front end uploaded int32 from pageable "
                "memory, one synchronous batch per worker: 1.2-1.3 Gsamples/s at 1024 x 8"}
    # PCIe-inclusive batch call: H2D + kernels + D2H, no MD5 / container
    an = GpuAnalyzer(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, bps, C, max_frames=1024, device=device)
    batch = pcm[: 1024 * BLOCK * C]
    an.encode_frames(batch, 1024, BLOCK, 0, rate)
    times = []
    for _ in range(5):
        t = time.perf_counter()
        an.encode_frames(batch, 1024, BLOCK, 0, rate)
        times.append(time.perf_counter() - t)
    ref_bytes, ref_off = an.encode_frames(batch, 1024, BLOCK, 0, rate)
    pin_bytes, pin_off, pin_times = an.encode_frames_pinned(batch, 1024, BLOCK, 0, rate, repeat=6)
    assert pin_bytes == ref_bytes and pin_off == ref_off
    an.close()
    out["encode_frames_pcie_inclusive"] = {
        "Msamples/s": round(batch.size / statistics.median(times) / 1e6, 1), "frames": 1024,
        "pinned_host_buffers_Msamples/s": round(batch.size / statistics.median(pin_times[1:]) / 1e6, 1),
        "note": "flacgpu_encode_frames, one context, synchronous: H2D + kernels + D2H; first figure from pageable "
                "numpy arrays (a fresh output array per call), second one with both host buffers from "
                "flacgpu_host_alloc (the copies at the link's rate)"}
    # the headline batch through the same call: 8192 frames, pinned buffers (268 MB in, the frames out)
    big = GpuAnalyzer(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, bps, C, max_frames=FRAMES, device=device)
    whole = pcm[: FRAMES * BLOCK * C]
    if whole.size == FRAMES * BLOCK * C:
        _, _, big_times = big.encode_frames_pinned(whole, FRAMES, BLOCK, 0, rate, repeat=4)
        out["encode_frames_pcie_inclusive"]["pinned_8192_frames_Msamples/s"] = round(
            whole.size / statistics.median(big_times[1:]) / 1e6, 1)
    big.close()
    return out




def md5_lane_rate():
    """GB/s of ONE MD5 chain inside a full 16-lane group (flacenc_md5_probe): the per-stream bound of the front ends"""
    import ctypes as C

    from flac_codec_amd import _lib

    L = _lib.lib()
    if not hasattr(L, "flacenc_md5_probe"):
        return None
    L.flacenc_md5_probe.restype = C.c_double
    L.flacenc_md5_probe.argtypes = [C.c_uint32, C.c_uint32]
    v = L.flacenc_md5_probe(16, 4096)
    return round(v / 16, 3) if v > 0 else None


SIGNAL_TEXT = {"ar2": "SURVEY 8(d) generator: one Q15 2-pole resonator per channel + dither (the encoder answers with LPC order 2)",
               "hi": "tests/_pcm.py synth_hi: AR(<= %d) -- cascaded Q15 resonators, model order and channel relation changing "
                     "every 8 frames, +-1 LSB dither"}
ASSIGNMENT_NAMES = {0: "independent", 8: "left_side", 9: "side_right", 10: "mid_side"}
SUBFRAME_NAMES = {0: "constant", 1: "verbatim", 2: "fixed", 3: "lpc"}
def pcie_probe(device, mib=256):
    """What the host link carries (flacgpu_link_probe, pinned memory, GB/s; tools/ubench/pcie_duplex.hip stand-alone):
    each direction alone by copy engine, both at once by copy engines, and upload by copy engine while a KERNEL stores
    into pinned host memory -- the shape of the asynchronous host path, whose frames k_frame64 writes over the link."""
    import ctypes as C

    from flac_codec_amd import _lib

    def run(up, down):
        v = C.c_double(0.0)
        rc = _lib.lib().flacgpu_link_probe(device, mib << 20, up, down, C.byref(v))
        assert rc == 0, _lib.lib().flacgpu_last_error()
        return round(v.value, 1)

    return {"h2d_GB/s": run(1, 0), "d2h_GB/s": run(0, 1), "both_by_copy_engines_sum_GB/s": run(1, 1),
            "engine_up_kernel_stores_down_sum_GB/s": run(1, 2), "kernel_loads_up_kernel_stores_down_sum_GB/s": run(2, 2)}


def pipelined_pcie(torch, cfg, pcm, device, orc, batch_frames, depth=4, batches=12):
    """The full-duplex host -> host batch loop (flacgpu_pipeline_*, include/flacenc_gpu.h; what encode.rs:558-585 drives):
    pinned host PCM -> finished frames in pinned host memory, `depth` contexts in rotation, no MD5.  Both upload widths:
    int32 samples (4 B) and the stream-width little-endian samples (3 B at 24 bits).  Output checked against the
    synchronous call."""
    from flac_codec_amd.gpu import GpuAnalyzer, PinnedBuffer, Pipeline

    C, bps, rate = cfg["ch"], cfg["bps"], cfg["rate"]
    link = pcie_probe(device)
    out = {"link": link, "batch_frames": batch_frames, "depth": depth, "batches_timed": batches}
    F = min(batch_frames, pcm.size // (BLOCK * C))
    batch = np.ascontiguousarray(pcm[: F * BLOCK * C])
    an = GpuAnalyzer(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, bps, C, max_frames=F, device=device)
    ref_bytes, ref_off = an.encode_frames(batch, F, BLOCK, 0, rate)
    an.close()
    width = (bps + 7) // 8
    le = np.ascontiguousarray(batch.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width]).reshape(-1)
    down = len(ref_bytes) / batch.size
    for name, bpsam, src in (("int32", 4, batch.view(np.uint8)), (f"packed_{width}_byte", width, le)):
        pipe = Pipeline(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, bps, C, max_frames=F, depth=depth, device=device)
        bufs = [PinnedBuffer(src.size) for _ in range(depth)]
        for b in bufs:
            b.array[:] = src
        # first round: every slot once (context warm-up), checked against the synchronous call
        for i in range(depth):
            assert pipe.submit(bufs[i].address, bpsam, F, BLOCK, 0, rate)
        for i in range(depth):
            data, off = pipe.retire()
            assert data == ref_bytes and off == ref_off, f"pipelined {name} batch differs from flacgpu_encode_frames"
        best = 0.0
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(batches):
                if pipe.in_flight() == depth:
                    pipe.retire(copy=False)
                assert pipe.submit(bufs[i % depth].address, bpsam, F, BLOCK, 0, rate)
            while pipe.in_flight():
                pipe.retire(copy=False)
            best = max(best, batches * batch.size / (time.perf_counter() - t) / 1e6)
        pipe.close()
        for b in bufs:
            b.close()
        # what the link allows: each direction alone, and the two together when they share one ceiling
        lim = min(link["h2d_GB/s"] / bpsam, link["d2h_GB/s"] / down, link["engine_up_kernel_stores_down_sum_GB/s"] / (bpsam + down))
        lim_fd = min(link["h2d_GB/s"] / bpsam, link["d2h_GB/s"] / down)
        out[name] = {"Msamples/s": round(best, 1), "bytes_up_per_sample": bpsam, "bytes_down_per_sample": round(down, 3),
                     "link_GB/s_used_up": round(best * bpsam / 1e3, 1), "link_GB/s_used_down": round(best * down / 1e3, 1),
                     "link_limit_Msamples/s": round(lim * 1e3, 1), "frac_of_link": round(best / (lim * 1e3), 4),
                     "frac_of_full_duplex_link": round(best / (lim_fd * 1e3), 4), "byte_identical_to_synchronous_call": True}
    out["note"] = ("best of 5 loops of `batches` batches through flacgpu_pipeline_submit / _retire; frac_of_full_duplex_link = "
                   "Msamples/s over min(h2d / bytes up, d2h / bytes down), the copy-engine rate of each direction ALONE measured "
                   "in this run; link_limit also honours what the two directions reach TOGETHER in the path's own shape "
                   "(copy engine up, kernel stores down)")
    return out


KERNEL_PROFILE_NAMES = {"k_autocorr": "k_autocorr4", "k_deinterleave": "k_deinterleave2", "k_pack": "k_frame64",
                        "k_cand64": "k_cand64p"}


class Workload:
    """One BASELINE configuration resident on this rank's GPU: `contexts` encoder contexts on their own HIP
    streams, each reading ITS OWN device buffer of synthetic PCM (a streaming encoder never re-reads its input:
    distinct buffers keep one context's batch out of the Infinity Cache of the next)."""

    def __init__(self, torch, cfg_id, frames, first_frame, contexts, device, seed_rank, lag_split=0, pcm=None,
                 signal="ar2"):
        from flac_codec_amd.gpu import GpuAnalyzer

        self.torch = torch
        self.cfg_id, self.cfg = cfg_id, CONFIGS[cfg_id]
        c = self.cfg
        self.C, self.BPS, self.RATE, self.LPC, self.PO = c["ch"], c["bps"], c["rate"], c["lpc"], c["po"]
        self.F, self.first_frame = frames, first_frame
        # context i encodes its own PCM (seed differs); `pcm` given: every context the caller's samples (strong
        # scaling: the rank's range of ONE stream)
        self.signal = signal
        # (the same `contexts` buffers on every rank, rotated by the rank: one generation per box, /dev/shm above)
        self.pcm = [pcm if pcm is not None else make_pcm(1000 + 16 * cfg_id + 101 * ((i + seed_rank) % contexts), frames, self.C,
                                                         self.BPS, signal, HI_SECTIONS.get(cfg_id, 6))
                    for i in range(contexts)]
        self.d_pcm = [torch.from_numpy(p).cuda() for p in self.pcm]
        self.ans = [GpuAnalyzer(BLOCK, self.PO, self.LPC, True, True, 2, 0.5, self.BPS, self.C, max_frames=frames,
                                device=device) for _ in range(contexts)]
        if lag_split:
            for a in self.ans:
                a.set_tuning(a.TUNE_LAG_SPLIT, lag_split)
        self.streams = [torch.cuda.Stream() for _ in self.ans]
        self.n = 0
        self.same_buffer = False     # A/B: every context reads buffer 0 (what the r02 bench did)
        self.only_first = False      # one context, kernels back to back

    def step(self):   # analysis + frame assembly of one whole batch (flacgpu_encode_device)
        i = 0 if self.only_first else self.n % len(self.ans)
        self.n += 1
        src = self.d_pcm[0 if self.same_buffer else i]
        self.ans[i].encode_device(src.data_ptr(), self.F, BLOCK, self.first_frame, self.RATE,
                                  stream=self.streams[i].cuda_stream)

    def prewarm(self, ms):
        torch = self.torch
        t = time.perf_counter()
        steps = 0
        while (time.perf_counter() - t) * 1e3 < ms:
            for _ in range(8):
                self.step()
            torch.cuda.synchronize()
            steps += 8
        return steps

    def timed(self, k, dist=None):
        torch = self.torch
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            self.step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def parity(self, orc, experiment=False):
        """EVERY context's last batch: all distinct frames byte-compared with the oracle, the whole batch decoded
        back and compared with its input on the device.  Returns (ok, identical, differ, compressed_bytes, verify)."""
        from _compare import orc_options_for

        oopts = orc_options_for(BLOCK, self.PO, self.LPC, True, True)
        check = min(DISTINCT, self.F)
        ok, identical, differ = True, 0, 0
        compressed = None
        verify = None
        C = self.C
        self.order_histogram, self.assignment_histogram, self.subframe_types = {}, {}, {}
        for i, an in enumerate(self.ans):
            # one more batch from the context's own buffer through the timed entry point (the A/B loops after the
            # timed region may have left another buffer's frames in it)
            an.encode_device(self.d_pcm[i].data_ptr(), self.F, BLOCK, self.first_frame, self.RATE,
                             stream=self.streams[i].cuda_stream)
            self.torch.cuda.synchronize()
            data, off = an.fetch_frames(self.F)
            if compressed is None:
                compressed = off[self.F]
            pcm = self.pcm[i]
            for f in range(check):
                planar = np.ascontiguousarray(pcm[f * BLOCK * C:(f + 1) * BLOCK * C].reshape(BLOCK, C).T)
                rc, fb, plan = orc.encode_frame(oopts, self.RATE, self.BPS, planar, frame_number=self.first_frame + f)
                if rc == 0:    # what the ORACLE decided for this frame: winning subframes by type and LPC order
                    a = ASSIGNMENT_NAMES.get(plan.assignment, str(plan.assignment))
                    self.assignment_histogram[a] = self.assignment_histogram.get(a, 0) + 1
                    for c in range(C):
                        sp = plan.sub[c]
                        t = SUBFRAME_NAMES[sp.type]
                        self.subframe_types[t] = self.subframe_types.get(t, 0) + 1
                        if sp.type == 3:
                            self.order_histogram[sp.order] = self.order_histogram.get(sp.order, 0) + 1
                if rc != 0 or data[off[f]:off[f + 1]] != fb:
                    differ += 1
                    if not experiment:
                        sys.stderr.write(f"config {self.cfg_id} context {i}: frame {f} differs from the oracle\n")
                        ok = False
                        break
                else:
                    identical += 1
            an.analyze_device(self.d_pcm[i].data_ptr(), self.F, BLOCK)
            an.pack_device(self.first_frame, self.RATE)
            vres, vms = an.verify_device(self.RATE, self.first_frame)
            if (vres.bad_structure, vres.bad_crc16, vres.frames_pcm_differs) != (0, 0, 0):
                ok = False
            if verify is None:
                verify = {"kernel_ms": round(vms, 3), "frames": vres.frames, "compared_pcm": bool(vres.compared_pcm),
                          "Msamples/s": round(self.F * BLOCK * C / (vms * 1e-3) / 1e6, 1)}
        return ok, identical, differ, compressed, verify

    def kernel_times(self, reps=10):
        """Per-kernel durations (HIP events on the launch stream inside the library), taken from steady-state
        batches: the clocks are re-warmed right before, three untimed passes, then `reps` passes back to back."""
        an, d = self.ans[0], self.d_pcm[0]
        self.prewarm(150.0)
        an.set_timing(True)
        acc = {}
        for r in range(3 + reps):
            an.analyze_device(d.data_ptr(), self.F, BLOCK)
            ms_a = an.kernel_ms()
            an.pack_device(self.first_frame, self.RATE)
            ms_b = an.kernel_ms()
            if r >= 3:
                for k, v in {**ms_a, **ms_b}.items():
                    acc[k] = acc.get(k, 0.0) + v / reps
        an.set_timing(False)
        return {k: v for k, v in acc.items() if v > 0}

    def handed(self):
        """What the candidate kernel hands to the frame kernel in one batch (flacgpu_handed_subframes of context 0's last
        batch): subframes handed with their residual, bytes written for them."""
        try:
            h, n, on = self.ans[0].handed_subframes()
        except Exception:
            return None
        if not on:
            return {"enabled": False}
        return {"enabled": True, "subframes": n, "handed": h, "frac": round(h / max(1, n), 4), "bytes_written": 4.0 * h * BLOCK}

    def algorithmic(self, compressed_bytes):
        """algorithmic bytes / flops per launch (DESIGN.md "Kernels")"""
        F, C = self.F, self.C
        n_cand = (4 if C == 2 else C) * F          # L, R, M, S per stereo frame; one per channel otherwise
        cand_samples = n_cand * BLOCK
        return {
            "k_deinterleave": ("hbm", 8.0 * F * BLOCK * C),
            "k_autocorr": ("f64", 2.0 * BLOCK * (self.LPC + 1) * n_cand),
            # fused FIXED + LPC + Rice search: every candidate's samples are read once, residuals
            # never leave registers, 280-byte plan out
            "k_cand64": ("hbm", 4.0 * cand_samples + 280.0 * n_cand),
            # k_frame64 (reported in the k_pack slot): samples in, finished frame bytes out
            "k_pack": ("hbm", 4.0 * F * BLOCK * C + compressed_bytes),
        }

    def kernels_report(self, compressed_bytes):
        """Per-kernel durations priced against each kernel's own bound: HBM bytes for the integer kernels, separately
        rounded f64 mul+add for the autocorrelation (an FMA is bit-identical only where the product is exact -- under
        window values of exactly 1.0, about half of a Tukey(0.5) block, where the kernels do fuse; elsewhere 39.3 TFLOP/s
        is what the f64 VALU can issue as mul + add; the datasheet's 78.6 TFLOP/s counts an FMA as two)."""
        acc = self.kernel_times()
        alg = self.algorithmic(compressed_bytes)
        kernels = {}
        for k, ms in acc.items():
            entry = {"ms": round(ms, 4)}
            if k in alg:
                kind, amount = alg[k]
                if kind == "hbm":
                    entry["GB/s"] = round(amount / (ms * 1e-3) / 1e9, 1)
                    entry["hbm_roofline_frac"] = round(entry["GB/s"] / HBM_PEAK_GBS, 4)
                else:
                    entry["GFLOP/s"] = round(amount / (ms * 1e-3) / 1e9, 1)
                    entry["f64_frac_of_78.6_TF"] = round(entry["GFLOP/s"] / F64_PEAK_GFLOPS, 4)
                    entry["f64_frac_of_39.3_TF_mul_add"] = round(entry["GFLOP/s"] / (F64_PEAK_GFLOPS / 2), 4)
            kernels[k] = entry
        priced = {k: v for k, v in kernels.items() if k in alg}
        # dominant kernel = the longest launch among the priced kernels (HBM- and f64-priced alike); launches within 3 %
        # of it count as tied (run-to-run noise decides their order) and the tie goes to an HBM-priced kernel, then to
        # the larger algorithmic amount
        longest = max(v["ms"] for v in priced.values())
        tied = [k for k, v in priced.items() if v["ms"] >= 0.97 * longest]
        dom = max(tied, key=lambda k: (alg[k][0] == "hbm", alg[k][1]))
        return kernels, dom, alg

    def close(self):
        for a in self.ans:
            a.close()
        self.d_pcm = None


def roofline_of(kernels, dom, alg, traffic, traffic_src, valu, stale, handed=None):
    """The `roofline` object of the dominant kernel, with the bound that prices it."""
    kind, amount = alg[dom]
    k = kernels[dom]
    if kind == "hbm":
        r = {"kernel": dom, "bound": "hbm", "achieved": k["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(k["GB/s"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
             "algorithmic_bytes": amount}
    else:
        r = {"kernel": dom, "bound": "f64-valu", "achieved": round(k["GFLOP/s"] / 1e3, 3), "peak": F64_PEAK_GFLOPS / 1e3,
             "unit": "TFLOP/s", "frac": round(k["GFLOP/s"] / F64_PEAK_GFLOPS, 4),
             "frac_of_mul_add_peak_39.3": round(k["GFLOP/s"] / (F64_PEAK_GFLOPS / 2), 4),
             "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flops": amount,
             "note": "exact left-fold autocorrelation on the VALU: separately rounded f64 mul and add under the window's "
                     "taper (an FMA or a matrix core would change the sums there), one FMA per term under its flat middle "
                     "(integer factors, exact product: identical rounding); 78.6 TFLOP/s is the stated f64 peak, 39.3 what "
                     "mul + add can reach"}
    r["avg_launch_ms"] = k["ms"]
    if handed and handed.get("enabled") and dom == "k_cand64" and kind == "hbm":
        # the candidate kernel also WRITES the winners' residuals for the frame kernel (r06): not counted in `achieved` /
        # `frac`, which keep r05's definition (4 B per candidate sample read + the plans); SURVEY 8(d) prices the FIR stage
        # at 4 B read + 4 B residual written, which is what the second figure adds for the handed subframes
        with_res = (amount + handed["bytes_written"]) / (k["ms"] * 1e-3) / 1e9
        r["handed_residuals"] = {"subframes": handed["subframes"], "handed": handed["handed"], "bytes_written": handed["bytes_written"],
                                 "achieved_with_them_GB/s": round(with_res, 1), "frac_with_them": round(with_res / HBM_PEAK_GBS, 4),
                                 "note": "k_cand64p stores the folded LPC residual of every subframe an LPC candidate wins and "
                                         "k_frame64 reads it back instead of both channels (no second FIR); `achieved` and `frac` "
                                         "above do NOT count these bytes"}
    r["dominance_rule"] = ("longest launch among the priced kernels (HBM bytes or f64 flops); launches within 3 % are tied, "
                           "an HBM-priced kernel wins a tie (every kernel's own fraction is in kernels.*)")
    r["counters_stale"] = stale
    if valu:
        r["valu_issue"] = valu     # the integer kernels are bound by VALU instruction issue, not by HBM
    return r


def running_build_id():
    from flac_codec_amd import _lib

    return _lib.build_id()


def profile_figures(cfg_id, dom, dom_ms, signal="ar2"):
    """HBM bytes and VALU instructions per launch of the dominant kernel from the committed rocprofv3 --pmc passes of
    this same command (tools/collect_profiles.sh; corrected as MI355X_MICROARCH.md prescribes) -- deterministic
    for a given input, workload AND BUILD -- and the instruction-issue floor of tools/issue_floor.py.  Every collection
    carries the build id of the library it was taken with (`_build_id`, flacgpu_build_id); the fourth value returned is
    True when any figure used comes from a collection of ANOTHER build than the running library (or of an unknown
    one): the counters next to this run's times are then stale."""
    import re

    traffic = traffic_src = valu = None
    variant = ("" if cfg_id == 3 else f"cfg{cfg_id}") + ("hi" if signal == "hi" else "")
    pat = r"^r\d+_" + (variant if variant else "[a-z]") + "_%s\\.json$"
    pdir = os.path.join(ROOT, "profiles")
    key = KERNEL_PROFILE_NAMES.get(dom, dom)
    mine = running_build_id()
    stale = False

    def newest(kind):
        names = sorted(f for f in os.listdir(pdir) if re.match(pat % kind, f))
        if not names:
            return None, None
        return names[-1], json.load(open(os.path.join(pdir, names[-1])))

    try:
        name, t = newest("traffic")
        k2 = key if key in t else (key[:-1] if key.endswith("p") and key[:-1] in t else None)
        if k2 is None and key == "k_frame64" and "k_sub64" in t:
            k2 = "k_sub64"     # 5..8 channels: one workgroup per subframe assembles the frames
        if k2:
            traffic = t[k2]["hbm_bytes"]
            traffic_src = {"source": "profiles/" + name, "kernel": k2, "build_id": t.get("_build_id"),
                           "fetch_bytes": t[k2].get("fetch_bytes"), "write_bytes": t[k2].get("write_bytes")}
            stale |= t.get("_build_id") != mine
    except Exception:
        pass
    try:
        name, vt = newest("valu")
        k2 = key if key in vt else (key[:-1] if key.endswith("p") and key[:-1] in vt else None)
        if k2 is None and key == "k_frame64" and "k_sub64" in vt:
            k2 = "k_sub64"
        if k2 and vt[k2].get("SQ_INSTS_VALU"):
            insts = vt[k2]["SQ_INSTS_VALU"]
            valu = {"wave_insts_per_launch": insts, "source": "profiles/" + name, "build_id": vt.get("_build_id"),
                    "achieved_Ginst/s": round(insts / (dom_ms * 1e-3) / 1e9, 1),
                    "nominal_peak_Ginst/s": round(VALU_ISSUE_PEAK / 1e9, 1),
                    "frac_of_nominal_4_cycle_peak": round(insts / (dom_ms * 1e-3) / VALU_ISSUE_PEAK, 4),
                    "valu_active_per_wave_cycle": vt[k2].get("valu_active_per_wave_cycle"),
                    "note": "the 4-cycle peak is a model; attainable_ms below is the floor from the kernel's own "
                            "instruction mix at MEASURED per-class issue costs (profiles/r03_issue_rate_ubench.json)"}
            stale |= vt.get("_build_id") != mine
    except Exception:
        pass
    try:
        floors = sorted(f for f in os.listdir(pdir) if re.match(r"^r\d+_issue_floor\.json$", f))
        fl = json.load(open(os.path.join(pdir, floors[-1])))
        e = fl.get(f"config{cfg_id}" + ("hi" if signal == "hi" else ""), {}).get(dom)
        if e and valu is not None:
            valu["attainable_ms"] = e["attainable_ms"]
            valu["attainable_source"] = f"profiles/{floors[-1]} (tools/issue_floor.py)"
            valu["attainable_build_id"] = fl.get("_build_id")
            valu["frac_of_attainable"] = round(e["attainable_ms"] / dom_ms, 4)
            stale |= fl.get("_build_id") != mine
    except Exception:
        pass
    if traffic is None and valu is None:
        stale = None      # nothing read from a collection
    return traffic, traffic_src, valu, stale


def batch_sweep(torch, cfg_id, args, device, sizes=(256, 512, 1024, 2048, 4096, 8192)):
    """Step time by batch size (frames per flacgpu_encode_device call), the same multi-context loop: what many small
    streams and a strong-scaled stream (frames / GPUs per rank) get.  per_sample_rate_vs_8192 = 1 means no small-batch
    penalty."""
    cfg = CONFIGS[cfg_id]
    base = make_pcm(1000 + 16 * cfg_id, max(sizes), cfg["ch"], cfg["bps"])
    rows = {}
    for n in sizes:
        w = Workload(torch, cfg_id, n, 0, args.contexts, device, 0, pcm=base[: n * BLOCK * cfg["ch"]])
        w.prewarm(60.0)
        steps = max(20, min(400, int(20 * 8192 / n)))
        ms = w.timed(steps) / steps * 1e3
        w.only_first = True
        ms1 = w.timed(max(10, steps // 2)) / max(10, steps // 2) * 1e3
        w.close()
        rows[str(n)] = {"ms_per_step": round(ms, 4), "Msamples/s": round(n * BLOCK * cfg["ch"] / (ms * 1e-3) / 1e6, 1),
                        "one_context_ms": round(ms1, 4)}
    full = rows[str(max(sizes))]["Msamples/s"]
    for r in rows.values():
        r["per_sample_rate_vs_8192"] = round(r["Msamples/s"] / full, 4)
    # the same small batches COALESCED: max(sizes) / n segments of n frames each (of different streams: shuffled places of
    # the buffer, their own frame numbers) as ONE batch through flacgpu_encode_segments_device -- what a caller with many
    # small streams submits instead of one batch per stream
    from flac_codec_amd.gpu import GpuAnalyzer

    F, C, per = max(sizes), cfg["ch"], BLOCK * cfg["ch"]
    d = [torch.from_numpy(make_pcm(1000 + 16 * cfg_id + 101 * i, F, C, cfg["bps"])).cuda() for i in range(args.contexts)]
    ans = [GpuAnalyzer(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, cfg["bps"], C, max_frames=F, device=device)
           for _ in range(args.contexts)]
    streams = [torch.cuda.Stream() for _ in ans]
    rng = np.random.Generator(np.random.PCG64(5))
    for n in sizes[:-1]:
        k = F // n
        order = rng.permutation(k)
        segs = [[(d[i].data_ptr() + int(j) * n * per * 4, n, 1000 * int(j)) for j in order] for i in range(len(ans))]
        for rep in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            steps = 24
            for it in range(steps):
                i = it % len(ans)
                ans[i].encode_segments_device(segs[i], cfg["rate"], stream=streams[i].cuda_stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / steps * 1e3
        rows[str(n)]["coalesced"] = {"segments_per_batch": k, "ms_per_batch_of_%d_frames" % F: round(ms, 4),
                                     "Msamples/s": round(F * per / (ms * 1e-3) / 1e6, 1),
                                     "per_sample_rate_vs_8192": round(F * per / (ms * 1e-3) / 1e6 / full, 4)}
    for a in ans:
        a.close()
    return rows


def histogram_summary(w, cfg_id):
    """What the oracle decided for the distinct frames of every context (Workload.parity)."""
    subs = sum(w.subframe_types.values()) or 1
    orders = {str(k): v for k, v in sorted(w.order_histogram.items())}
    at_least = 20 if cfg_id == 5 else 8
    return {"order_histogram": orders, "subframe_types": w.subframe_types, "channel_assignments": w.assignment_histogram,
            "winning_subframes": subs,
            f"frac_of_winning_subframes_at_lpc_order_>={at_least}":
                round(sum(v for k, v in w.order_histogram.items() if k >= at_least) / subs, 4)}


def measure_other_config(torch, cfg_id, args, device, orc, signal="ar2"):
    """A BASELINE configuration that is not the headline one (or the headline one on another input signal), at its SURVEY
    8(d) size: ms/step over `steps` timed steps of the same multi-context loop, every context's distinct frames against
    the oracle (with the histogram of the oracle's decisions), the dominant kernel and its recomputed roofline fraction."""
    t_wall = time.perf_counter()
    w = Workload(torch, cfg_id, args.frames, 0, args.contexts, device, 0, signal=signal)
    pre = w.prewarm(200.0)
    for _ in range(3):
        w.step()
    steps = args.other_steps
    el = w.timed(steps)
    ms = el / steps * 1e3
    ok, identical, differ, compressed, verify = w.parity(orc)
    assert ok, f"config {cfg_id} ({signal}): parity check failed"
    kernels, dom, alg = w.kernels_report(compressed)
    traffic, traffic_src, valu, stale = profile_figures(cfg_id, dom, kernels[dom]["ms"], signal)
    st0 = w.ans[0].stats()
    cfg = w.cfg
    out = {"workload": cfg["text"], "level": cfg["level"], "signal": SIGNAL_TEXT[signal] if signal != "hi" else
           SIGNAL_TEXT[signal] % (2 * HI_SECTIONS.get(cfg_id, 6)),
           "frames_per_step": w.F, "contexts": len(w.ans), "steps": steps,
           "ms_per_step": round(ms, 4), "Msamples/s": round(w.F * BLOCK * w.C / (ms * 1e-3) / 1e6, 1),
           "prewarm_steps": pre,
           "frames_byte_identical_to_oracle": identical, "frames_checked": identical + differ,
           "frames_round_tripped_on_device": w.F * len(w.ans),
           "oracle_decisions": histogram_summary(w, cfg_id),
           "dominant_kernel": roofline_of(kernels, dom, alg, traffic, traffic_src, valu, stale, w.handed()),
           "fixed_count": {"decided_by_bound": st0.fixed_decided, "refetched": st0.fixed_refetched,
                           "note": "candidates of context 0 since it was created whose exact FIXED bit count was put off "
                                   "behind the LPC half (Params::defer_fixed): skipped / counted after a re-fetch"},
           "kernels": kernels,
           "compression_ratio": round(compressed / (w.F * BLOCK * w.C * ((w.BPS + 7) // 8)), 4)}
    w.close()
    out["wall_s"] = round(time.perf_counter() - t_wall, 1)
    return out


FINAL_LINE_LIMIT = 8192      # hard cap of the record the driver parses (a 21 KB line did not parse)
FINAL_LINE_TARGET = 4096


def _pick(d, *keys):
    return {k: d.get(k) for k in keys} if isinstance(d, dict) else None


def _short(text, n):
    return text if text is None or len(text) <= n else text[: n - 3] + "..."


def compact_record(out, detail_path=None):
    """The record printed as the LAST stdout line: the driver's contract keys, `roofline`, `cpu_baseline` and a dozen
    scalar extras -- no prose, no per-config blocks.  Everything else (other_configs, batch_sweep, end_to_end, kernels,
    notes) lives in the detail file written beside it (`write_detail`)."""
    cfgd = out.get("config") or {}
    rec = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    rec["config"] = {"workload": _short(cfgd.get("workload"), 200), "baseline_config": cfgd.get("baseline_config"),
                     "level": cfgd.get("level"), "frames_per_step": cfgd.get("frames_per_step"),
                     "frames_per_gpu": cfgd.get("frames_per_gpu"), "contexts": cfgd.get("contexts"),
                     "parallelism": cfgd.get("parallelism")}
    r = out.get("roofline")
    if r:
        rr = _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes",
                   "algorithmic_flops", "avg_launch_ms", "counters_stale")
        rec["roofline"] = {k: v for k, v in rr.items() if v is not None or k in ("traffic", "counters_stale")}
        v = r.get("valu_issue") or {}
        if v.get("frac_of_attainable") is not None:
            rec["roofline"]["frac_of_issue_floor"] = v["frac_of_attainable"]
        h = r.get("handed_residuals")
        if h:   # (bytes the kernel also writes for the frame kernel: NOT in `achieved` / `frac`)
            rec["roofline"]["handed_residual_bytes"] = h["bytes_written"]
            rec["roofline"]["frac_with_handed_residuals"] = h["frac_with_them"]
    else:
        rec["roofline"] = None
    c = out.get("cpu_baseline")
    if c:
        rec["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                               "sample": _short(c.get("sample"), 160)}
        for k in ("reference_fork_join", "all_cores_frame_parallel"):
            if c.get(k):
                rec["cpu_baseline"][k] = _pick(c[k], "value", "cores")
    else:
        rec["cpu_baseline"] = None
    extras = {"build_id": out.get("build_id"), "hbm_bound_fraction": out.get("hbm_bound_fraction"),
              "compression_ratio": out.get("compression_ratio"), "experiment": out.get("experiment")}
    if out.get("sustained"):
        extras["sustained_ms_per_step"] = out["sustained"]["ms_per_step"]
        ck = out["sustained"].get("gpu_clock")
        if ck:   # (the pool's boxes do not clock alike: `value` moves with this)
            extras["sclk_MHz"] = ck["sclk_MHz"]
            extras["socket_power_W"] = ck["socket_power_W"]
    var = out.get("variants") or {}
    if var.get("one_context_back_to_back"):
        extras["one_context_ms_per_step"] = var["one_context_back_to_back"]["ms_per_step"]

    def cfg_scalars(e):
        d = e.get("dominant_kernel") or {}
        r = {"ms_per_step": e.get("ms_per_step"), "kernel": d.get("kernel"), "kernel_ms": d.get("avg_launch_ms"),
             "frac": d.get("frac"), "identical": e.get("frames_byte_identical_to_oracle"),
             "checked": e.get("frames_checked")}
        fc = e.get("fixed_count") or {}
        if fc.get("decided_by_bound") or fc.get("refetched"):
            r["fixed_count_skipped"] = fc["decided_by_bound"]
            r["fixed_count_refetched"] = fc["refetched"]
        return r

    if var.get("high_order_input"):
        extras["high_order"] = cfg_scalars(var["high_order_input"])
    for name, e in sorted((out.get("other_configs") or {}).items()):
        extras[name] = cfg_scalars(e)
        if e.get("high_order_input"):
            extras[name + "_high_order"] = cfg_scalars(e["high_order_input"])
    kern = out.get("kernels") or {}
    extras["kernel_ms"] = {k: v.get("ms") for k, v in kern.items()}
    p = out.get("parity") or {}
    extras["parity"] = {"identical_per_rank": p.get("frames_byte_identical_to_oracle_per_rank"),
                        "checked_per_rank": p.get("frames_checked_per_rank"),
                        "round_tripped_per_rank": p.get("frames_round_tripped_on_device_per_rank"),
                        "ranks": p.get("ranks_checked")}
    st = out.get("analysis_stats") or {}
    extras["fir_recheck"] = st.get("fir_recheck")
    e2e = out.get("end_to_end") or {}
    pp = e2e.get("pipelined_pcie") if isinstance(e2e, dict) else None
    if isinstance(pp, dict):
        extras["pipelined_pcie"] = {k: _pick(pp[k], "Msamples/s", "frac_of_link") for k in ("int32", "packed_3_byte")
                                    if isinstance(pp.get(k), dict)}
    if isinstance(e2e, dict):   # the feeding paths (host PCM -> .flac bytes, MD5 included) beside the device-resident `value`
        ms, mss = e2e.get("many_streams"), e2e.get("many_small_streams")
        fed = {}
        if isinstance(ms, dict):
            fed["many_streams"] = _pick(ms, "Msamples/s", "per_stream_writers_Msamples/s", "frac_of_link", "md5_chain_bound_Msamples/s",
                                        "same_volume_256_streams_Msamples/s", "same_volume_256_streams_frac_of_link")
        if isinstance(mss, dict):
            fed["many_small_streams"] = _pick(mss, "coalesced_Msamples/s", "one_writer_per_stream_Msamples/s", "coalesced_frac_of_link",
                                              "coalesced_wins_at_every_length")
        one = e2e.get("one_stream")
        if isinstance(one, dict):
            fed["one_stream_Msamples/s"] = one.get("Msamples/s")
        if fed:
            extras["fed"] = fed
    if isinstance(out.get("host_to_host"), dict):
        extras["host_to_host"] = _pick(out["host_to_host"], "Msamples/s", "host_copy_GB/s", "copies_per_output_byte",
                                       "shard_threads_near_their_gpu", "error")
    sc = out.get("shard_counters")
    if isinstance(sc, dict):
        extras["shards"] = _pick(sc, "ranks_seen", "backend", "total_frames", "total_bytes", "min_frame", "max_frame")
    rec["extra"] = extras
    rec["detail"] = detail_path
    return rec


def final_line(out, detail_path=None):
    """Strict JSON, one line, below FINAL_LINE_LIMIT bytes whatever the detail holds: extras are dropped (largest first)
    if a future block ever pushes the record over the target."""
    rec = compact_record(out, detail_path)
    line = json.dumps(rec, allow_nan=False, separators=(", ", ": "))
    extra = rec["extra"]
    while len(line) > FINAL_LINE_TARGET and extra:
        biggest = max(extra, key=lambda k: len(json.dumps(extra[k])))
        del extra[biggest]
        line = json.dumps(rec, allow_nan=False, separators=(", ", ": "))
    assert len(line) < FINAL_LINE_LIMIT and "\n" not in line
    return line


def write_detail(out, path):
    """The full record (every block the compact line leaves out) as a file; returns the path written or None."""
    for cand in ([path] if path else []) + [os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                                           os.path.join(ROOT, "bench_detail.json")]:
        try:
            if os.path.dirname(cand) and not os.path.isdir(os.path.dirname(cand)):
                continue
            with open(cand, "w") as fh:
                json.dump(out, fh)
                fh.write("\n")
            return os.path.relpath(cand, ROOT) if cand.startswith(ROOT) else cand
        except OSError:
            continue
    return None


def metric_text(cfg, bit_exact):
    what = {2: "level 5 (fixed predictors only), 48kHz/16-bit stereo", 3: "level 8, 48kHz/24-bit stereo",
            4: "level 8, 192kHz/24-bit 8-channel", 5: "level 8 exhaustive (LPC order 32), 96kHz/24-bit stereo"}[cfg]
    return f"Msamples/s encode at {what}" + ("; bit-exact vs reference" if bit_exact else "; NOT bit-exact (experiment)")


def in_process(args):
    """--in-process: ONE process drives all N devices through the C ABI's flacgpu_multi_* (include/flacenc_gpu.h "several
    GPUs": a shard -- contexts in rotation, own HIP streams -- per device, contiguous frame ranges, the four-integer
    records merged on the host); no torch.distributed, no RCCL: the path has no data exchange.  Same contract as the
    one-process-per-GPU run: W warm-up steps, K timed steps bracketed by a sync of every device, whole-job throughput."""
    import torch

    import _oracle as orc
    from _compare import orc_options_for
    from flac_codec_amd.gpu import MultiDevice

    N = args.gpus
    share = os.environ.get("FLAC_BENCH_SHARE_DEVICE") == "1"   # TEST: every shard on GPU 0
    if not share and torch.cuda.device_count() < N:
        sys.stderr.write(f"bench.py: --gpus {N} but only {torch.cuda.device_count()} visible; refusing to report\n")
        sys.exit(3)
    devices = [0] * N if share else list(range(N))
    cfg = CONFIGS[args.config]
    C, BPS, RATE = cfg["ch"], cfg["bps"], cfg["rate"]
    F, ctxs = args.frames, args.contexts
    md = MultiDevice(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, BPS, C, max_frames=F, devices=devices, depth=ctxs)
    pcm = [[make_pcm(1000 + 16 * args.config + 101 * ((i + k) % ctxs), F, C, BPS, args.signal,
                     HI_SECTIONS.get(args.config, 6)) for i in range(ctxs)] for k in range(N)]
    bufs = [[torch.from_numpy(p).to(f"cuda:{devices[k]}") for p in pcm[k]] for k in range(N)]

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    n = [0]

    def step():   # shard k encodes frames [k F, (k + 1) F) of the job: contiguous ranges per device
        i = n[0] % ctxs
        n[0] += 1
        for k in range(N):
            md.encode_device(k, bufs[k][i].data_ptr(), F, BLOCK, k * F, RATE)

    sync()
    t = time.perf_counter()
    pre = 0
    while (time.perf_counter() - t) * 1e3 < args.prewarm_ms:
        for _ in range(8):
            step()
        md.wait()
        pre += 8
    for _ in range(args.warmup):
        step()
    md.wait()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    md.wait()
    sync()
    elapsed = time.perf_counter() - t0
    per_shard, merged = md.counters()
    # parity: every shard's last batch -- distinct frames byte for byte against the oracle
    oopts = orc_options_for(BLOCK, cfg["po"], cfg["lpc"], True, True)
    last = (n[0] - 1) % ctxs
    identical = 0
    for k in range(N):
        data, off = md.fetch_last(k, F)
        p = pcm[k][last]
        for f in range(min(DISTINCT, F)):
            planar = np.ascontiguousarray(p[f * BLOCK * C:(f + 1) * BLOCK * C].reshape(BLOCK, C).T)
            rc, fb, _ = orc.encode_frame(oopts, RATE, BPS, planar, frame_number=k * F + f)
            assert rc == 0 and data[off[f]:off[f + 1]] == fb, f"shard {k}: frame {f} differs from the oracle"
            identical += 1
    md.close()
    del bufs
    # host -> host through flacgpu_multi_encode: pinned PCM of ONE stream (N x F frames at the stream's width), batches of 1024
    # frames dealt to the shards in turn, every retired batch copied once into its place in `out` -- what the host side moves
    host_leg = None
    try:
        from flac_codec_amd.gpu import PinnedBuffer

        width = (BPS + 7) // 8
        whole = np.concatenate([pcm[k][0] for k in range(N)])
        le = np.ascontiguousarray(whole.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width]).reshape(-1)
        pin = PinnedBuffer(le.size)
        pin.array[:] = le
        mh = MultiDevice(BLOCK, cfg["po"], cfg["lpc"], True, True, 2, 0.5, BPS, C, max_frames=1024, devices=devices, depth=6)
        body, off, _, _ = mh.encode(pin.array, N * F, BLOCK, 0, RATE, bytes_per_sample=width)      # warm-up + parity
        for f in (0, N * F - 1):
            src = pcm[f // F][0][(f % F) * BLOCK * C:((f % F) + 1) * BLOCK * C]
            rc, fb, _ = orc.encode_frame(oopts, RATE, BPS, np.ascontiguousarray(src.reshape(BLOCK, C).T), frame_number=f)
            assert rc == 0 and body[off[f]:off[f + 1]] == fb, f"host leg: frame {f} differs from the oracle"
        o0, c0, _ = mh.host_copy_stats()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            mh.encode_raw(pin.address, N * F, BLOCK, 0, RATE, bytes_per_sample=width)
            ts.append(time.perf_counter() - t)
        o1, c1, near = mh.host_copy_stats()
        dt = statistics.median(ts)
        host_leg = {"Msamples/s": round(whole.size / dt / 1e6, 1), "frames": N * F, "batch_frames": 1024, "depth": 6,
                    "bytes_up_per_sample": width, "host_copy_GB/s": round((c1 - c0) / 3 / dt / 1e9, 2),
                    "copies_per_output_byte": round((c1 - c0) / max(1, o1 - o0), 4), "shard_threads_near_their_gpu": near,
                    "note": "flacgpu_multi_encode: pinned stream-width PCM in, frames in a pageable caller buffer out; median of 3"}
        mh.close()
        pin.close()
    except Exception as e:   # (diagnostic leg: never costs the contract line)
        host_leg = {"error": repr(e)}
    # the dominant kernel's roofline on device 0 (one context, kernels back to back), as in the per-rank run
    w = Workload(torch, args.config, F, 0, 1, devices[0], 0, signal=args.signal)
    w.ans[0].encode_device(w.d_pcm[0].data_ptr(), F, BLOCK, 0, RATE, stream=w.streams[0].cuda_stream)
    torch.cuda.synchronize()
    _, off0 = w.ans[0].fetch_frames(F)
    kernels, dom, alg = w.kernels_report(off0[F])
    traffic = traffic_src = valu = stale = None
    if F == FRAMES:
        traffic, traffic_src, valu, stale = profile_figures(args.config, dom, kernels[dom]["ms"], args.signal)
    roofline = roofline_of(kernels, dom, alg, traffic, traffic_src, valu, stale)
    w.close()
    samples_per_step = N * F * BLOCK * C
    ms = elapsed / args.steps * 1e3
    out = {
        "metric": metric_text(args.config, True), "value": round(samples_per_step * args.steps / elapsed / 1e6, 2),
        "unit": "Msamples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32/i64 (+f64 LPC analysis)",
        "data": "synthetic",
        "config": {"workload": f"config {args.config}: {cfg['text']}; {F} frames per GPU per step, PCM resident in HBM, "
                               f"ONE process driving {N} device shard(s) through flacgpu_multi_* ({ctxs} contexts each)",
                   "baseline_config": args.config, "level": cfg["level"], "frames_per_gpu": F, "frames_per_step": N * F,
                   "parallelism": f"frame ranges x{N}, in-process", "contexts": ctxs, "devices": devices,
                   "prewarm": f"{pre} untimed steps"},
        "roofline": roofline, "cpu_baseline": None, "kernels": kernels, "build_id": running_build_id(),
        "hbm_bound_fraction": round((8.0 * samples_per_step / N) / (ms * 1e-3) / 8e12, 4),
        "parity": {"frames_byte_identical_to_oracle_per_rank": identical // N, "frames_checked_per_rank": identical // N,
                   "ranks_checked": N},
        "shard_counters": {"total_frames": merged[0], "total_bytes": merged[1], "min_frame": merged[2],
                           "max_frame": merged[3], "frames_per_rank": [c[0] for c in per_shard], "ranks_seen": N,
                           "backend": "in-process (flacgpu_merge_counters)"},
        "host_to_host": host_leg,
    }
    print(final_line(out, write_detail(out, args.detail)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (36 ms of timed region at the headline workload; the sustained leg runs 200 more)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (SURVEY.md 8(d)); 3 is the headline metric")
    ap.add_argument("--frames", type=int, default=FRAMES,
                    help="FLAC frames per step: per GPU (weak scaling) or of the whole stream (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank encodes --frames frames of its own; strong: ONE stream of --frames frames, "
                         "contiguous frame ranges per rank (SURVEY.md 8(d) config 4)")
    ap.add_argument("--emit-flac", default=None,
                    help="strong scaling: rank 0 writes the whole .flac (frames gathered from the ranks, metadata rebuilt "
                         "from the gathered sizes) to this path")
    ap.add_argument("--signal", choices=("ar2", "hi"), default="ar2",
                    help="input signal of the headline loop: ar2 = SURVEY 8(d)'s generator (the contract; LPC order 2 wins "
                         "everywhere), hi = tests/_pcm.py synth_hi (high LPC orders win) -- for profile collections of "
                         "`variants.high_order_input`; the default line reports both")
    ap.add_argument("--only-batch-sweep", action="store_true", help="print the batch_sweep block alone and exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs block (BASELINE configs 2, 4, 5 next to the headline one)")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of every other_configs entry")
    ap.add_argument("--e2e-batch-frames", type=int, default=0,
                    help="batch_frames of the writers in the end_to_end block (0: the library default)")
    ap.add_argument("--contexts", type=int, default=3, choices=(1, 2, 3, 4, 5, 6),
                    help="encoder contexts consecutive batches rotate through (multi-buffering; r06: three -- with the hand-over four are 1 %% slower, profiles/r06_contexts.json)")
    ap.add_argument("--lag-split", type=int, default=0, choices=(0, 2, 4),
                    help="waves the autocorrelation lags are split over (0: the library default)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps run before the warm-up until this much wall time has passed "
                         "(the clocks of an idle GPU take longer to ramp than a few batches)")
    ap.add_argument("--experiment", choices=("mfma_autocorr",), default=None,
                    help="mfma_autocorr: the f64-MFMA autocorrelation (re-associated sums, NOT bit-exact) in place "
                         "of the exact kernel; the line is labelled and frames that differ from the oracle are "
                         "counted instead of failing the run")
    ap.add_argument("--detail", default=None,
                    help="where the full record goes (default: gpurun_out/bench_detail.json if that directory exists, "
                         "else bench_detail.json beside bench.py); stdout carries ONE compact line")
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="steps of the additional long timed loop reported as `sustained`")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process drives all --gpus devices through the C ABI's flacgpu_multi_* (no torch.distributed); "
                         "default: one process per GPU")
    args = ap.parse_args()

    if args.in_process:
        return in_process(args)
    if args.experiment == "mfma_autocorr":
        os.environ["FLACGPU_TEST_KNOBS"] = "1"            # test-only knobs are ignored without this
        os.environ["FLACGPU_EXPERIMENT_MFMA_AC"] = "1"   # read by the library when a context is created
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("FLAC_BENCH_SHARE_DEVICE") == "1"   # TEST: all ranks on GPU 0, gloo collectives

    import torch

    if share:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)\n")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="gloo" if share else "nccl", rank=rank, world_size=world)
        dist = dist_mod

    from flac_codec_amd.parallel import gather_shard_counters, shard_range

    if args.only_batch_sweep:
        print(json.dumps({f"config{c}": batch_sweep(torch, c, args, local_rank) for c in (3, 4)}))
        return
    cfg = CONFIGS[args.config]
    C, BPS, RATE, MAX_LPC = cfg["ch"], cfg["bps"], cfg["rate"], cfg["lpc"]
    strong = args.scaling == "strong"
    whole = None
    if strong:
        # ONE stream of --frames frames; this rank encodes the contiguous range [lo, hi) with the frame numbers
        # the whole stream gives them (every rank generates the same synthetic stream and takes its range)
        lo, hi = shard_range(args.frames, world, rank)
        whole = make_pcm(1000 + 16 * args.config, args.frames, C, BPS)
        F, first_frame = hi - lo, lo
        if F == 0:
            sys.stderr.write(f"bench.py: rank {rank} got no frames ({args.frames} frames over {world} ranks)\n")
            sys.exit(2)
        w = Workload(torch, args.config, F, first_frame, args.contexts, local_rank, 0, args.lag_split,
                     pcm=whole[lo * BLOCK * C: hi * BLOCK * C])
        total_frames = args.frames
    else:
        F, first_frame = args.frames, rank * args.frames      # contiguous frame ranges per GPU (8(e))
        w = Workload(torch, args.config, F, first_frame, args.contexts, local_rank, rank, args.lag_split,
                     signal=args.signal)
        total_frames = F * world
    an = w.ans[0]

    prewarm_steps = w.prewarm(args.prewarm_ms)
    for _ in range(args.warmup):
        w.step()
    elapsed = w.timed(args.steps, dist)
    samples_per_step = total_frames * BLOCK * C
    value = samples_per_step * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    sustained = None
    if args.sustained_steps > 0:
        sensors = GpuSensors(torch, local_rank)
        sensors.start()
        el = w.timed(args.sustained_steps, dist)
        clock = sensors.stop()
        sustained = {"steps": args.sustained_steps, "ms_per_step": round(el / args.sustained_steps * 1e3, 4),
                     "value": round(samples_per_step * args.sustained_steps / el / 1e6, 2), "unit": "Msamples/s"}
        if clock:
            sustained["gpu_clock"] = clock
    # A/B of the input buffers and of the context count (same run, same clocks)
    variants = {}
    if not strong and len(w.ans) > 1:
        w.same_buffer = True
        el = w.timed(args.steps, dist)
        w.same_buffer = False
        variants["every_context_reads_one_buffer"] = {
            "ms_per_step": round(el / args.steps * 1e3, 4),
            "note": "what the r02 bench timed: all contexts re-read the same 256 MiB (an Infinity Cache of 256 MB under it); "
                    "`value` is measured with one buffer per context"}
    w.only_first = True
    el = w.timed(args.steps, dist)
    w.only_first = False
    one_ctx_ms = el / args.steps * 1e3
    variants["one_context_back_to_back"] = {"ms_per_step": round(one_ctx_ms, 4)}
    # per-shard counters (frames, bytes, min/max frame size) gathered over RCCL: the only
    # cross-GPU exchange of the path (seek-table offsets / STREAMINFO)
    counters = gather_shard_counters(an, F, dist)

    # ---- parity precondition on this very batch, on EVERY rank and EVERY context: all distinct frames
    # byte-identical to the oracle, every frame of the batch round-tripped on the device
    import _oracle as orc

    ok, identical, differ, compressed_bytes, verify = w.parity(orc, experiment=bool(args.experiment))
    st = an.stats()
    if st.order_ties:
        sys.stderr.write(f"rank {rank}: {st.order_ties} candidates inside the libm-sensitive order band\n")
    okv = 1 if ok else 0
    if dist:
        t = torch.tensor([okv], dtype=torch.int32, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        okv = int(t.item())
    assert okv == 1, "parity check failed on at least one rank"

    emitted = None
    if strong and args.emit_flac:
        # the stream itself: frames of the last batch gathered to rank 0, metadata from the gathered sizes
        from flac_codec_amd.encode import Options
        from flac_codec_amd.parallel import finish_sharded_stream

        data, off = an.fetch_frames(F)
        o = Options.best() if cfg["lpc"] >= 12 else Options.default()
        o = o.max_lpc_order(cfg["lpc"] or None).max_partition_order(cfg["po"])
        flac = finish_sharded_stream(bytes(data[: off[F]]), [off[i + 1] - off[i] for i in range(F)], whole, o, RATE, BPS, C,
                                     dist)
        if rank == 0:
            with open(args.emit_flac, "wb") as fh:
                fh.write(flac)
            emitted = {"path": args.emit_flac, "bytes": len(flac)}

    out = None
    if rank == 0:
        analysis_stats = {"lpc_failed": st.lpc_failed, "order_ties": st.order_ties,
                          "order_ties_resolved_on_host": st.order_ties_resolved, "log2_edge": st.log2_edge,
                          "fir_recheck": st.fir_recheck, "fir_rechecked": st.fir_rechecked,
                          "fixed_count_decided_by_bound": st.fixed_decided, "fixed_count_refetched": st.fixed_refetched,
                          "candidates": (4 if C == 2 else C) * F}
        kernels, dom, alg = w.kernels_report(compressed_bytes)
        sum_kernels = sum(v["ms"] for v in kernels.values())
        variants["one_context_back_to_back"]["sum_of_kernel_ms"] = round(sum_kernels, 4)
        variants["one_context_back_to_back"]["sum_over_step"] = round(sum_kernels / one_ctx_ms, 4)
        mfma_exp = None
        if 1 <= MAX_LPC <= 16 and C == 2 and not strong:
            # experiment: the same autocorrelation on the f64 matrix cores (NOT bit-exact, not used)
            an.analyze_device(w.d_pcm[0].data_ptr(), F, BLOCK)
            mf = an.experiment_mfma_autocorr()
            issued = 4 * F * (BLOCK // 64) * 2 * 2048.0       # MFMAs x 2048 flop (16x16x4 f64)
            mfma_exp = {"kernel": "k_autocorr_mfma", "ms": round(mf["ms"], 4),
                        "issued_TFLOP/s": round(issued / (mf["ms"] * 1e-3) / 1e12, 2),
                        "mfma_util_vs_78.6_TF": round(issued / (mf["ms"] * 1e-3) / 78.6e12, 4),
                        "max_rel_err_of_a_lag": mf["max_rel_err"],
                        "candidates_compared": mf["compared"],
                        "candidates_whose_quantised_lpc_params_change": mf["params_differ"],
                        "note": "re-associated sums are not the reference's left fold; kept off the product path"}
        traffic = traffic_src = valu = stale = None
        if F == FRAMES:
            traffic, traffic_src, valu, stale = profile_figures(args.config, dom, kernels[dom]["ms"], args.signal)
        roofline = roofline_of(kernels, dom, alg, traffic, traffic_src, valu, stale, w.handed())
        headline_decisions = histogram_summary(w, args.config)

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(orc, cfg, w.pcm[0], F)
        # (the headline workload's contexts and device buffers are released BEFORE the host-path legs: with them alive
        # the pipelined leg measured 5-10 % below its stand-alone figure)
        pcm0 = w.pcm[0]
        n_ctx = len(w.ans)
        w.close()
        torch.cuda.empty_cache()
        e2e = None
        if world == 1 and not args.no_end_to_end:
            e2e = end_to_end(cfg, pcm0, local_rank, orc, args.e2e_batch_frames)
        others = None
        if world == 1 and not args.no_other_configs and not args.experiment and args.frames == FRAMES:
            others = {}
            for cid in sorted(CONFIGS):
                if cid != args.config:
                    others[f"config{cid}"] = measure_other_config(torch, cid, args, local_rank, orc)
                    if cid == 5:
                        # the order-32 configuration on an input that makes the encoder choose orders >= 20
                        others[f"config{cid}"]["high_order_input"] = measure_other_config(torch, cid, args, local_rank, orc,
                                                                                          signal="hi")
            variants["batch_sweep"] = {f"config{c}": batch_sweep(torch, c, args, local_rank) for c in (3, 4)}
            if CONFIGS[args.config]["lpc"] and args.signal == "ar2":
                # the headline configuration on an input on which the encoder chooses HIGH LPC orders (SURVEY's generator
                # makes it choose order 2 everywhere: `value` times a 2-tap FIR)
                variants["high_order_input"] = measure_other_config(torch, args.config, args, local_rank, orc, signal="hi")
        out = {
            "metric": metric_text(args.config, differ == 0),
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "i32/i64 (+f64 LPC analysis)",
            "data": "synthetic",
            "config": {"signal": SIGNAL_TEXT[args.signal] if args.signal != "hi" else
                       SIGNAL_TEXT["hi"] % (2 * HI_SECTIONS.get(args.config, 6)),
                       "workload": f"config {args.config}: {cfg['text']}; "
                                   + (f"ONE stream of {total_frames} frames per step cut into contiguous frame ranges over "
                                      f"{world} rank(s)" if strong else f"{F} frames per GPU per step")
                                   + f", PCM resident in HBM, frame bytes produced in HBM; consecutive batches rotate "
                                     f"through {len(w.ans)} encoder context(s) on separate HIP streams, each reading its "
                                     f"own input buffer",
                       "baseline_config": args.config, "level": cfg["level"],
                       "frames_per_gpu": F, "frames_per_step": total_frames, "parallelism": f"frame ranges x{world}",
                       "contexts": n_ctx, "distinct_input_buffers": n_ctx if not strong else 1,
                       "prewarm": f"{prewarm_steps} untimed steps ({args.prewarm_ms:.0f} ms) before the warm-up"},
            "sustained": sustained,
            "variants": variants,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "other_configs": others,
            "kernels": kernels,
            "oracle_decisions": headline_decisions,
            "build_id": running_build_id(),
            "mfma_autocorr_experiment": mfma_exp,
            "device_verify": verify,
            "compression_ratio": round(compressed_bytes / (F * BLOCK * C * ((BPS + 7) // 8)), 4),
            "hbm_bound_fraction": round((8.0 * samples_per_step / world) / (ms_per_step * 1e-3) / 8e12, 4),
            "experiment": args.experiment,
            "parity": {"frames_byte_identical_to_oracle_per_rank": identical, "frames_checked_per_rank": identical + differ,
                       "contexts_checked": len(w.ans),
                       "frames_round_tripped_on_device_per_rank": F * len(w.ans), "ranks_checked": world},
            "analysis_stats": analysis_stats,
            "shard_counters": counters,
            "emitted_stream": emitted,
        }
        del pcm0
    else:
        w.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        detail_path = write_detail(out, args.detail)
        print(final_line(out, detail_path), flush=True)


if __name__ == "__main__":
    main()
