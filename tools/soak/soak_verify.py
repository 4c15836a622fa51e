import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from flac_codec_amd.gpu import GpuAnalyzer
from _pcm import synth_fast
SECONDS = float(os.environ.get("SOAK_SECONDS", "120"))
F, B = 4096, 4096
rng = np.random.Generator(np.random.PCG64(5))
cfgs = [(2, 24, 12), (2, 16, 8), (1, 24, 12), (2, 24, 32), (4, 20, 12), (8, 24, 12), (6, 16, 8), (3, 24, 12), (8, 16, 10)]   # (8 / 6: k_sub64; 3, 4, 8: read in place, r04)
tot_frames = tot_batches = 0
t_end = time.time() + SECONDS
per = SECONDS / len(cfgs)
for (ch, bps, lpc) in cfgs:
    pool = []
    for s in range(3):
        x = synth_fast(900 + s + ch + bps, ch, bps, F * B)
        if s == 1:   # wasted bits + silence + noise patches
            x = x.reshape(F, B * ch).copy()
            x[::7] = 0
            x[3::11] = (x[3::11] >> 3) << 3
            x[5::13] = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=x[5::13].shape, dtype=np.int64).astype(np.int32)
            x = x.reshape(-1)
        pool.append(torch.from_numpy(np.ascontiguousarray(x)).cuda())
    an = GpuAnalyzer(B, 6, lpc, True, True, 2, 0.5, bps, ch, max_frames=F)
    t0 = time.time(); n = 0
    while time.time() - t0 < per:
        d = pool[n % len(pool)]
        an.encode_device(d.data_ptr(), F, B, n * F, 48000)
        res, _ = an.verify_device(48000, n * F)
        assert (res.frames, res.bad_structure, res.bad_crc16, res.frames_pcm_differs, res.samples_differ) == (F, 0, 0, 0, 0), (ch, bps, lpc, n, res.bad_structure, res.bad_crc16, res.frames_pcm_differs)
        assert res.compared_pcm == 1
        n += 1
    tot_batches += n; tot_frames += n * F
    print(f"cfg ch={ch} bps={bps} lpc={lpc}: {n} batches verified", flush=True)
    an.close()
print(f"soak_verify: {tot_batches} batches, {tot_frames} frames, {tot_frames * B / 1e9:.1f} G samples per channel encoded and decoded back on the device, 0 mismatches")
