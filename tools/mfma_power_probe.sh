#!/bin/bash
# The register-only MFMA loop (tools/mfma_power_probe.hip) with one constant operand pair and with random operands, socket power
# read next to it with rocm-smi once a second (GPU box).   bash tools/mfma_power_probe.sh > gpurun_out/mfma_power.txt
set -e
hipcc -O3 --offload-arch=gfx950 "$(dirname "$0")/mfma_power_probe.hip" -o /tmp/mfma_power_probe 2>/dev/null
for mode in "const 16" "random 16" "random 32"; do
  /tmp/mfma_power_probe $mode > "/tmp/probe_$mode.txt" &
  pid=$!
  sleep 3
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 1; done
  wait $pid
  cat "/tmp/probe_$mode.txt" | sed -n '1p;6,9p'
done
