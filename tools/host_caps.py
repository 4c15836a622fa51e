#!/usr/bin/env python3
"""Host-side capacity of the box the bench runs on: logical CPUs, cgroup CPU quota, and how the
stream MD5 (one serial chain per stream, hashlib's MD5 here) scales with concurrent streams.
The many-stream end-to-end rate cannot exceed the aggregate MD5 rate."""
import hashlib, os, threading, time, json

def read(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None

info = {"os.cpu_count": os.cpu_count(), "sched_getaffinity": len(os.sched_getaffinity(0)),
        "cgroup cpu.max": read("/sys/fs/cgroup/cpu.max"),
        "cgroup v1 quota/period": (read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")),
        "loadavg": read("/proc/loadavg")}
buf = os.urandom(48 << 20)   # 16 Msamples of 24-bit audio
def work(out, i):
    t = time.perf_counter(); hashlib.md5(buf).digest(); out[i] = time.perf_counter() - t
scaling = {}
for n in (1, 2, 4, 8, 16, 32, 64):
    out = [0] * n
    ths = [threading.Thread(target=work, args=(out, i)) for i in range(n)]
    t = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t
    scaling[n] = {"aggregate_GB/s": round(n * len(buf) / dt / 1e9, 2), "aggregate_Msamples/s_24bit": round(n * len(buf) / 3 / dt / 1e6),
                  "slowest_thread_s": round(max(out), 3)}
info["md5_scaling"] = scaling
print(json.dumps(info, indent=1))
