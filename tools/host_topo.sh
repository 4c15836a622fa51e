#!/bin/bash
# Host topology of the box the bench runs on: what the feeding paths (packing, MD5 engines, pinned staging) have to live with.
# Writes gpurun_out/host_topo.txt.  No GPU work.
out=${1:-gpurun_out/host_topo.txt}
mkdir -p "$(dirname "$out")"
{
echo "== lscpu"; lscpu 2>/dev/null | head -40
echo "== cgroup cpu.max"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
echo "== cpuset"; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; grep -i "cpus_allowed_list\|mems_allowed_list" /proc/self/status
echo "== numa nodes"; for n in /sys/devices/system/node/node*; do echo "$n: cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4 $5}')"; done
echo "== gpu numa"; for c in /sys/class/drm/card*/device; do echo "$c numa_node=$(cat $c/numa_node 2>/dev/null) local_cpulist=$(cat $c/local_cpulist 2>/dev/null) vendor=$(cat $c/vendor 2>/dev/null)"; done
echo "== kfd topology"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n: $(grep -E 'simd_count|cpu_cores_count|drm_render_minor|domain|location_id' $n/properties 2>/dev/null | tr '\n' ' ')"; done
echo "== thread siblings of cpu0"; cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list
echo "== meminfo"; head -5 /proc/meminfo
echo "== loadavg"; cat /proc/loadavg
echo "== numactl"; which numactl && numactl -H
echo "== rocm-smi topo"; /opt/rocm/bin/rocm-smi --showtoponuma 2>/dev/null | head -20
} > "$out" 2>&1
echo "wrote $out"
