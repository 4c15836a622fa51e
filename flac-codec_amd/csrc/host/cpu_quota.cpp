// cpu_quota.cpp -- how many CPUs the process may really use (the thread counts of the host front ends and of the MD5 engines).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "host_internal.h"

// CPUs this process may really use: the cgroup's CPU quota (a container with 16 CPUs' worth of time on a 256-thread host
// is throttled for the rest of the period once its threads have burnt the quota -- more runnable threads than that only add
// stalls), else the hardware threads
namespace flacenc_host {
unsigned usable_cpus() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
            char q[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) {
                const long quota = std::atol(q);
                if (quota > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long>(1, (quota + period / 2) / period));
            }
            std::fclose(f);
        } else {
            long quota = -1, period = 0;
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
                std::fclose(g);
            }
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g, "%ld", &period) != 1) period = 0;
                std::fclose(g);
            }
            if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long>(1, (quota + period / 2) / period));
        }
        return hw;
    }();
    return n;
}
}  // namespace flacenc_host
