import os, sys, time, hashlib, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from flac_codec_amd.gpu import GpuAnalyzer
from _pcm import synth_fast
F, B = 8192, 4096
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 120
pcm = synth_fast(11, 2, 24, F * B)
d = torch.from_numpy(pcm).cuda()
ans = [GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, 2, max_frames=F) for _ in range(3)]
for a in ans:
    a.set_tuning(a.TUNE_LAG_SPLIT, 2)
    if os.environ.get("SOAK_TWO_RANGES"): a.set_two_ranges(True)
streams = [torch.cuda.Stream() for _ in ans]
ref = None
n = 0
bad = 0
while time.time() < t_end:
    for rep in range(50):
        for i, a in enumerate(ans):
            a.encode_device(d.data_ptr(), F, B, 0, 48000, stream=streams[i].cuda_stream)
    torch.cuda.synchronize()
    for a in ans:
        data, off = a.fetch_frames(F)
        h = hashlib.sha256(data).hexdigest()
        if ref is None: ref = h
        if h != ref: bad += 1
        n += 1
print(f"soak: {n} outputs hashed over {50 * n} batches, mismatches {bad}, sha256 {ref[:16]}")
for a in ans: a.close()
