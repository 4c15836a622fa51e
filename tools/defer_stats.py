import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench, torch
for cfg, sig in ((3, "ar2"), (3, "hi"), (5, "hi")):
    w = bench.Workload(torch, cfg, 8192, 0, 1, 0, 0, signal=sig)
    w.only_first = True
    w.step(); torch.cuda.synchronize()
    s0 = w.ans[0].stats()
    w.step(); torch.cuda.synchronize()
    s1 = w.ans[0].stats()
    print(cfg, sig, "per batch: decided", s1.fixed_decided - s0.fixed_decided, "refetched", s1.fixed_refetched - s0.fixed_refetched, "of", 8192 * 4)
