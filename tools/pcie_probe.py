#!/usr/bin/env python3
"""Host <-> device copy rates of this box (pinned memory): one copy at a time and several streams at once,
each direction alone and both together -- the ceiling of the host path (3 B/sample up, ~1.6 B/sample down)."""
import time

import torch

n = 256 << 20
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]


def run(h2d, d2h, streams):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(streams):
        with torch.cuda.stream(ss[i]):
            if h2d:
                d[i].copy_(h[i], non_blocking=True)
            if d2h:
                h[(i + 2) % 4 if h2d else i].copy_(d[(i + 2) % 4 if h2d else i], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return streams * n * (int(h2d) + int(d2h)) / dt / 1e9


for streams in (1, 2, 4):
    run(True, False, streams)
    print(f"streams {streams}: H2D {run(True, False, streams):.1f} GB/s, D2H {run(False, True, streams):.1f} GB/s, "
          f"both {run(True, True, min(streams, 2)):.1f} GB/s (sum)")
