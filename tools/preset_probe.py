import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from _pcm import synth_fast
from flac_codec_amd.gpu import GpuAnalyzer
CASES = (("fast", 1152, 3, 0, False, False), ("default", 4096, 5, 8, True, True), ("best", 4096, 6, 12, True, True),
         # the other wave block lengths at the `best` settings (direct input since r03; FLACGPU_NO_DIRECT_SHORT=1 for A/B)
         ("best-1024", 1024, 6, 12, True, True), ("best-1152", 1152, 6, 12, True, True),
         ("best-2048", 2048, 6, 12, True, True), ("best-2304", 2304, 6, 12, True, True))
for name, B, po, lpc, ms, ex in [c for c in CASES if len(sys.argv) < 2 or c[0] in sys.argv[1:]]:
    F = 8192 * 4096 // B
    base = synth_fast(77, 2, 16, B * 512)
    pcm = np.tile(base, (F + 511) // 512)[: F * B * 2]
    d = torch.from_numpy(pcm).cuda()
    NCTX = int(__import__('os').environ.get('PROBE_CONTEXTS', '4'))
    ans = [GpuAnalyzer(B, po, lpc, ms, ex, 2, 0.5, 16, 2, max_frames=F) for _ in range(NCTX)]
    ss = [torch.cuda.Stream() for _ in ans]
    for i in range(16):
        ans[i % NCTX].encode_device(d.data_ptr(), F, B, 0, 44100, stream=ss[i % NCTX].cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(100):
        ans[i % NCTX].encode_device(d.data_ptr(), F, B, 0, 44100, stream=ss[i % NCTX].cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 100
    ans[0].set_timing(True)
    ans[0].analyze_device(d.data_ptr(), F, B); ans[0].pack_device(0, 44100); torch.cuda.synchronize()
    k = ans[0].kernel_ms()
    print(f"{name}: {dt*1e3:.3f} ms per {F} frames of {B} = {F*B*2/dt/1e9:.1f} Gsamples/s; kernels {({a: round(b, 3) for a, b in k.items()})}")
    for a in ans: a.close()
