#!/bin/bash
# sample the shader clock and power while the probe loop runs
cd $GRAFT_REPO_ROOT
for n in 1 4; do
  python3 tools/kprobe.py --contexts $n --steps 20000 --own-buffers > /tmp/kp_$n.log 2>&1 &
  PID=$!
  sleep 6
  for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | head -3 | tr '\n' ' '; echo; sleep 0.5; done
  wait $PID
  tail -1 /tmp/kp_$n.log
done
