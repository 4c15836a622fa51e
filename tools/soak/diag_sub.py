import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "soak"))
import diag_case as D
from flac_codec_amd.gpu import GpuAnalyzer
seed = int(sys.argv[1])
c, pcm = D.describe(seed)
B, C = c["block"], c["channels"]
def run(x, nf, last, tag, **over):
    cc = dict(c); cc.update(over)
    an = GpuAnalyzer(B, cc["max_po"], cc["max_lpc"], cc["mid_side"], cc["exhaustive"], cc["window"][0], cc["window"][1], cc["bps"], C, max_frames=nf)
    an.encode_frames(x, nf, last, cc["first"], cc["rate"])
    res, _ = an.verify_device(cc["rate"], cc["first"])
    plans, subs, _ = an.fetch(nf)
    print(tag, {k: getattr(res, k) for k, _ in res._fields_})
    for f in range(nf):
        for ch in range(C):
            s = subs[f * C + ch]
            print("   frame", f, "ch", ch, "type", s.type, "order", s.order, "bits", s.bits, "wasted", s.wasted, "bps", s.bps, "po", s.partition_order, "src", s.source, "rice", list(s.rice[:2]), "esc", list(s.escape_bits[:2]))
    an.close()
run(pcm[: 2 * B * C], 2, B, "frames 0-1")
run(pcm[2 * B * C:], 1, c["last"], "frame 2 alone")
run(pcm[2 * B * C:], 1, c["last"], "frame 2 alone lpc0", max_lpc=0)
run(pcm[2 * B * C:], 1, c["last"], "frame 2 alone lpc12", max_lpc=12)
