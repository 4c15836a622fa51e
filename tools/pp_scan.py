"""Scan of the pipelined host -> host leg (bench.pipelined_pcie): depth x batch size, with and without GPU_MAX_HW_QUEUES.
   python3 tools/pp_scan.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, _oracle as orc
cfg = bench.CONFIGS[3]
pcm = bench.make_pcm(1048, 8192, 2, 24)
out = []
for frames, depth, batches in ((2048, 4, 64), (2048, 6, 64), (4096, 4, 32), (1024, 6, 128), (2048, 3, 64), (2048, 4, 64)):
    r = bench.pipelined_pcie(torch, cfg, pcm, 0, orc, frames, depth=depth, batches=batches)
    rec = {"frames": frames, "depth": depth, "int32": r["int32"]["Msamples/s"], "int32_frac": r["int32"]["frac_of_link"],
           "packed3": r["packed_3_byte"]["Msamples/s"], "packed3_frac": r["packed_3_byte"]["frac_of_link"], "link": r["link"]}
    print(json.dumps(rec), flush=True)
    out.append(rec)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
