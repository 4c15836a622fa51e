"""diagnostic: bench.pipelined_pcie called from a bare script"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, _oracle as orc
cfg = bench.CONFIGS[3]
pcm = bench.make_pcm(1048, 2048, 2, 24)
r = bench.pipelined_pcie(torch, cfg, pcm, 0, orc, 2048, depth=4, batches=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
print(r["int32"]["Msamples/s"], r["packed_3_byte"]["Msamples/s"])
