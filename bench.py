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
metadata rewrite) is outside the step; DESIGN.md states its cost and the PCIe-inclusive rate.

Usage:  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torchrun)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

BLOCK = 4096
CHANNELS = 2
BPS = 24
RATE = 48000
MAX_LPC = 12
MAX_PO = 6
FRAMES = 8192          # 8192 blocks x 4096
DISTINCT = 512         # distinct synthetic frames, tiled to FRAMES


def make_pcm(seed, frames):
    from _pcm import synth_fast

    base = synth_fast(seed, CHANNELS, BPS, BLOCK * min(DISTINCT, frames))
    reps = (frames + DISTINCT - 1) // DISTINCT
    return np.tile(base, reps)[: frames * BLOCK * CHANNELS]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES, help="FLAC frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts", type=int, default=3, choices=(1, 2, 3, 4),
                    help="encoder contexts consecutive batches rotate through (multi-buffering)")
    ap.add_argument("--lag-split", type=int, default=0, choices=(0, 2, 4),
                    help="waves the autocorrelation lags are split over (0: the library default)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world)
        dist = dist_mod

    from flac_codec_amd.gpu import GpuAnalyzer
    from flac_codec_amd.parallel import gather_shard_counters

    F = args.frames
    pcm = make_pcm(1000 + rank, F)                 # every rank encodes its own frame range
    d_pcm = torch.from_numpy(pcm).cuda()
    an = GpuAnalyzer(BLOCK, MAX_PO, MAX_LPC, True, True, 2, 0.5, BPS, CHANNELS, max_frames=F,
                     device=local_rank)
    first_frame = rank * F                          # contiguous frame ranges per GPU (8(e))
    # Multi-buffering, as a streaming encoder runs: consecutive batches rotate through a few
    # encoder contexts (each with its own plans / output buffer in HBM) on their own HIP streams,
    # so the HBM-bound, latency-bound and VALU-bound kernels of neighbouring batches overlap; a
    # context's output stays valid until the context is used again.  --contexts 1 runs every
    # batch back to back on one context.
    ans = [an]
    for _ in range(args.contexts - 1):
        ans.append(GpuAnalyzer(BLOCK, MAX_PO, MAX_LPC, True, True, 2, 0.5, BPS, CHANNELS, max_frames=F,
                               device=local_rank))
    if args.lag_split:
        for a in ans:
            a.set_tuning(a.TUNE_LAG_SPLIT, args.lag_split)
    streams = [torch.cuda.Stream() for _ in ans]
    step_no = [0]

    def step():   # analysis + frame assembly of one whole batch (flacgpu_encode_device)
        i = step_no[0] % len(ans)
        step_no[0] += 1
        ans[i].encode_device(d_pcm.data_ptr(), F, BLOCK, first_frame, RATE,
                             stream=streams[i].cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-shard counters (frames, bytes, min/max frame size) gathered over RCCL: the only
    # cross-GPU exchange of the path (seek-table offsets / STREAMINFO)
    counters = gather_shard_counters(an, F, dist)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = F * BLOCK * CHANNELS * world
    value = samples_per_step * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    out = None
    if rank == 0:
        # ---- parity precondition on this very batch: first frames byte-identical to the oracle
        import _oracle as orc
        from _compare import orc_options_for

        data, off = an.fetch_frames(F)
        oopts = orc_options_for(BLOCK, MAX_PO, MAX_LPC, True, True)
        check = 16
        for f in range(check):
            planar = np.ascontiguousarray(
                pcm[f * BLOCK * CHANNELS:(f + 1) * BLOCK * CHANNELS].reshape(BLOCK, CHANNELS).T)
            rc, fb, _ = orc.encode_frame(oopts, RATE, BPS, planar, frame_number=first_frame + f)
            assert rc == 0 and data[off[f]:off[f + 1]] == fb, f"frame {f} differs from the oracle"
        compressed_bytes = off[F]
        st = an.stats()
        analysis_stats = {"lpc_failed": st.lpc_failed, "order_ties": st.order_ties, "log2_edge": st.log2_edge,
                          "candidates": 4 * F}

        # ---- per-kernel durations (HIP events on the launch stream), one extra timed pass
        an.set_timing(True)
        reps = 5
        acc = {}
        for _ in range(reps):
            an.analyze_device(d_pcm.data_ptr(), F, BLOCK)
            ms_a = an.kernel_ms()
            an.pack_device(first_frame, RATE)
            ms_b = an.kernel_ms()
            for k, v in {**ms_a, **ms_b}.items():
                acc[k] = acc.get(k, 0.0) + v / reps
        an.set_timing(False)
        # device-side decode + CRC verification of the frames just packed (round trip in HBM)
        an.analyze_device(d_pcm.data_ptr(), F, BLOCK)
        an.pack_device(first_frame, RATE)
        vres, vms = an.verify_device(RATE, first_frame)
        assert (vres.bad_structure, vres.bad_crc16, vres.frames_pcm_differs) == (0, 0, 0)
        verify = {"kernel_ms": round(vms, 3), "frames": vres.frames, "compared_pcm": bool(vres.compared_pcm),
                  "Msamples/s": round(F * BLOCK * CHANNELS / (vms * 1e-3) / 1e6, 1)}
        # experiment: the same autocorrelation on the f64 matrix cores (NOT bit-exact, not used)
        an.analyze_device(d_pcm.data_ptr(), F, BLOCK)
        mf = an.experiment_mfma_autocorr()
        issued = 4 * F * (BLOCK // 64) * 2 * 2048.0       # MFMAs x 2048 flop (16x16x4 f64)
        mfma_exp = {"kernel": "k_autocorr_mfma", "ms": round(mf["ms"], 4),
                    "issued_TFLOP/s": round(issued / (mf["ms"] * 1e-3) / 1e12, 2),
                    "mfma_util_vs_78.6_TF": round(issued / (mf["ms"] * 1e-3) / 78.6e12, 4),
                    "max_rel_err_of_a_lag": mf["max_rel_err"],
                    "candidates_compared": mf["compared"],
                    "candidates_whose_quantised_lpc_params_change": mf["params_differ"],
                    "note": "re-associated sums are not the reference's left fold; kept off the product path"}
        n_cand = 4 * F                      # L, R, M, S per stereo frame
        cand_samples = n_cand * BLOCK
        # algorithmic bytes / flops per launch (DESIGN.md "Kernels")
        alg = {
            "k_deinterleave": ("hbm", 8.0 * F * BLOCK * CHANNELS),
            "k_fixed": ("hbm", 8.0 * cand_samples),
            "k_autocorr": ("f64", 2.0 * BLOCK * (MAX_LPC + 1) * n_cand),
            "k_fir": ("hbm", 8.0 * cand_samples),
            # fused FIXED + LPC + Rice search: every candidate's samples are read once, residuals
            # never leave registers, 280-byte plan out
            "k_cand64": ("hbm", 4.0 * cand_samples + 280.0 * n_cand),
            "k_emit": ("hbm", 8.0 * F * BLOCK * CHANNELS),
            "k_pack": ("hbm", 4.0 * F * BLOCK * CHANNELS + compressed_bytes),
            "k_crc": ("hbm", float(compressed_bytes)),
        }
        kernels = {}
        for k, ms in acc.items():
            entry = {"ms": round(ms, 4)}
            if k in alg and ms > 0:
                kind, amount = alg[k]
                if kind == "hbm":
                    entry["GB/s"] = round(amount / (ms * 1e-3) / 1e9, 1)
                else:
                    entry["GFLOP/s"] = round(amount / (ms * 1e-3) / 1e9, 1)
            kernels[k] = entry
        hbm_kernels = {k: v for k, v in kernels.items() if "GB/s" in v}
        dom = max(hbm_kernels, key=lambda k: hbm_kernels[k]["ms"])
        achieved = hbm_kernels[dom]["GB/s"]
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes of this same
        # command, tools/collect_profiles.sh; corrected as MI355X_MICROARCH.md prescribes)
        traffic = None
        traffic_src = None
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
            if prof and F == FRAMES:
                t = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
                fast = {"k_fixed": "k_fixed16", "k_fir": "k_fir16", "k_autocorr": "k_autocorr3",
                        "k_deinterleave": "k_deinterleave2"}
                key = fast.get(dom, dom) if fast.get(dom, dom) in t else dom
                if key in t:
                    traffic = t[key]["hbm_bytes"]
                    traffic_src = {"source": "profiles/" + prof[-1], "kernel": key,
                                   "fetch_bytes": t[key].get("fetch_bytes"), "write_bytes": t[key].get("write_bytes")}
        except Exception:
            traffic = None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes": alg[dom][1], "avg_launch_ms": hbm_kernels[dom]["ms"]}
        if dom == "k_cand64":
            # SURVEY 8(d) prices the stages this kernel fuses separately (K1 FIXED 8 B, K4 FIR 8 B,
            # K5 Rice search 4 B per candidate sample); the fusion removes all but one 4-byte read.
            fused = 20.0 * cand_samples
            roofline["unfused_accounting"] = {
                "bytes": fused, "equivalent_GB/s": round(fused / (hbm_kernels[dom]["ms"] * 1e-3) / 1e9, 1),
                "fir_stage_alone_GB/s": round(8.0 * cand_samples / (hbm_kernels[dom]["ms"] * 1e-3) / 1e9, 1),
                "note": "K1 8 B + K4 8 B + K5 4 B per candidate sample (SURVEY 8(d)); `achieved` above "
                        "counts only the bytes the fused kernel still has to move"}
        # The integer kernels are bound by VALU instruction issue, not by HBM: one wave64 VALU
        # instruction holds a SIMD for 4 cycles, so the chip issues at most 1024 SIMDs x clk / 4
        # wave-instructions per second.  Instruction counts per launch come from the SQ_INSTS_VALU
        # pass of tools/collect_profiles.sh (deterministic for a given input).
        try:
            vprof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_valu.json"))
            if vprof and F == FRAMES:
                vt = json.load(open(os.path.join(ROOT, "profiles", vprof[-1])))
                key = {"k_fixed": "k_fixed16", "k_fir": "k_fir16", "k_autocorr": "k_autocorr3",
                       "k_deinterleave": "k_deinterleave2"}.get(dom, dom)
                if key in vt and vt[key].get("SQ_INSTS_VALU"):
                    insts = vt[key]["SQ_INSTS_VALU"]
                    peak_issue = 1024 * 2.4e9 / 4
                    roofline["valu_issue"] = {
                        "wave_insts_per_launch": insts, "source": vprof[-1],
                        "achieved_Ginst/s": round(insts / (hbm_kernels[dom]["ms"] * 1e-3) / 1e9, 1),
                        "peak_Ginst/s": round(peak_issue / 1e9, 1),
                        "frac": round(insts / (hbm_kernels[dom]["ms"] * 1e-3) / peak_issue, 4),
                        "valu_active_per_wave_cycle": vt[key].get("valu_active_per_wave_cycle")}
        except Exception:
            pass

        # ---- CPU baseline: the oracle (C restatement of the reference, NOT the Rust binary),
        # timed on this box's host cores over the same workload
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_frames = min(F, 4096)
            sample = pcm[: cpu_frames * BLOCK * CHANNELS]
            oo = orc.options("best")
            t1 = time.perf_counter()
            rc, ref, _ = orc.encode_stream(oo, RATE, BPS, CHANNELS, sample, total_known=True, threads=1)
            dt1 = time.perf_counter() - t1
            assert rc == 0
            ncores = os.cpu_count() or 1
            t2 = time.perf_counter()
            rc, ref2, _ = orc.encode_stream(oo, RATE, BPS, CHANNELS, sample, total_known=True,
                                            threads=ncores)
            dtn = time.perf_counter() - t2
            assert ref == ref2
            # the reference forks at most 4 tasks per stereo frame (L || R, then M || S, each
            # FIXED || LPC; encode.rs:2690-2745, 2906-2928) and walks frames sequentially: 4
            # frame-parallel threads of the restatement bound that from above
            t3 = time.perf_counter()
            rc, ref3, _ = orc.encode_stream(oo, RATE, BPS, CHANNELS, sample, total_known=True, threads=4)
            dt4 = time.perf_counter() - t3
            assert ref == ref3
            cpu = {"value": round(sample.size / dt1 / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                   "kind": "port",
                   "sample": f"first {cpu_frames} frames of the bench batch, full encode to an "
                             f"in-memory .flac (MD5 + analysis + bit-pack + CRC), 1 thread",
                   "four_threads": {"value": round(sample.size / dt4 / 1e6, 3), "cores": 4,
                                    "note": "upper bound of the reference's per-frame fork-join "
                                            "(at most 4 concurrent tasks per stereo frame)"},
                   "all_cores_frame_parallel": {"value": round(sample.size / dtn / 1e6, 3),
                                                "cores": ncores,
                                                "note": "not something the reference does"}}
        out = {
            "metric": "Msamples/s encode at level 8, 48kHz/24-bit stereo; bit-exact vs reference",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32/i64 (+f64 LPC analysis)",
            "data": "synthetic",
            "config": {"workload": "48 kHz/24-bit stereo, blocksize 4096, level 8 (Options::best: "
                                   "LPC order 12, partition order 6, mid-side, exhaustive), "
                                   f"{F} frames per GPU per step, PCM resident in HBM, frame bytes "
                                   f"produced in HBM; consecutive batches rotate through {len(ans)} "
                                   "encoder context(s) on separate HIP streams",
                       "frames_per_gpu": F, "parallelism": f"frame ranges x{world}",
                       "contexts": len(ans)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernels": kernels,
            "mfma_autocorr_experiment": mfma_exp,
            "device_verify": verify,
            "compression_ratio": round(compressed_bytes / (F * BLOCK * CHANNELS * 3), 4),
            "hbm_bound_fraction": round((8.0 * samples_per_step / world) / (ms_per_step * 1e-3) / 8e12, 4),
            "parity_checked_frames": check,
            "analysis_stats": analysis_stats,
            "shard_counters": counters,
        }
    for a in ans:
        a.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
