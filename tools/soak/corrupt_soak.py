import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from flac_codec_amd.gpu import GpuAnalyzer
from _pcm import synth_fast
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
rng = np.random.Generator(np.random.PCG64(11))
tot = det = 0
for (ch, bps, block, lpc) in [(2, 16, 4096, 12), (2, 24, 1152, 8), (1, 16, 4096, 32), (8, 24, 576, 12), (2, 24, 4096, 0)]:
    F = 64
    pcm = synth_fast(600 + ch + bps, ch, bps, block * F)
    an = GpuAnalyzer(block, 6, lpc, True, True, 2, 0.5, bps, ch, max_frames=F)
    data, off = an.encode_frames(pcm, F, block, 0, 48000)
    data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    dbuf = an.device_buffer(4)
    for trial in range(120):
        bad = data.copy()
        nflip = int(rng.integers(1, 40))
        frames = set()
        for _ in range(nflip):
            pos = int(rng.integers(0, len(bad)))
            kind = int(rng.integers(0, 3))
            if kind == 0: bad[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1: bad[pos] = 0
            else: bad[pos] = 0xFF
            frames.add(int(np.searchsorted(off, pos, side="right") - 1))
        changed = {int(np.searchsorted(off, p, side="right") - 1) for p in np.nonzero(bad != data)[0]}
        assert hip.hipMemcpy(C.c_void_p(dbuf), bad.ctypes.data_as(C.c_void_p), len(bad), 1) == 0
        res, _ = an.verify_device(48000)
        tot += 1
        # every changed frame must fail the CRC-16 or the structure / PCM check
        flagged = res.bad_crc16
        assert res.bad_structure + res.frames_pcm_differs <= len(changed) + 0, (res.bad_structure, res.frames_pcm_differs, len(changed))
        assert res.bad_crc16 <= len(changed)
        if len(changed) == 0 or res.bad_crc16 >= 1: det += 1
    assert hip.hipMemcpy(C.c_void_p(dbuf), data.ctypes.data_as(C.c_void_p), len(data), 1) == 0
    res, _ = an.verify_device(48000)
    assert (res.bad_structure, res.bad_crc16, res.frames_pcm_differs) == (0, 0, 0)
    an.close()
    print("config", ch, bps, block, lpc, "ok", flush=True)
print("corrupt soak:", tot, "trials, detected", det)
