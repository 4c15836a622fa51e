"""Step time of the `best` settings for 1 / 2 / 3 / 4 / 6 / 8 channels (24-bit, 4096-sample blocks, 67 M samples per
batch, four contexts, interleaved PCM resident in HBM).  `python3 tools/channel_probe.py [channels ...]`"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import torch         # noqa: E402
from _pcm import synth_fast   # noqa: E402
from flac_codec_amd.gpu import GpuAnalyzer   # noqa: E402

B = 4096
for C in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8]:
    F = 8192 * 2 // C
    base = synth_fast(91 + C, C, 24, B * 256)
    pcm = np.tile(base, (F + 255) // 256)[: F * B * C]
    d = torch.from_numpy(pcm).cuda()
    ans = [GpuAnalyzer(B, 6, 12, True, True, 2, 0.5, 24, C, max_frames=F) for _ in range(4)]
    ss = [torch.cuda.Stream() for _ in ans]
    for i in range(12):
        ans[i % 4].encode_device(d.data_ptr(), F, B, 0, 48000, stream=ss[i % 4].cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(60):
        ans[i % 4].encode_device(d.data_ptr(), F, B, 0, 48000, stream=ss[i % 4].cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 60
    ans[0].set_timing(True)
    ans[0].analyze_device(d.data_ptr(), F, B)
    k = ans[0].kernel_ms()
    ans[0].pack_device(0, 48000)
    k.update(ans[0].kernel_ms())
    print(f"{C} ch: {dt*1e3:.3f} ms per {F} frames = {F*B*C/dt/1e9:.1f} Gsamples/s; kernels {({a: round(b, 3) for a, b in k.items() if b > 0})}")
    for a in ans:
        a.close()
