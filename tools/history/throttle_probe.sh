#!/bin/bash
# Is the long-epoch submission stall the container's CFS quota (cpu.max, 100 ms period)?  Runs the epoch probe with the default
# OpenMP pool and with a small one, printing the cgroup's throttle counters around each run (measurement tool).
stat() { grep -E 'nr_periods|nr_throttled|throttled_usec' /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
cat /sys/fs/cgroup/cpu.max
for omp in default 4; do
  echo "== OMP_NUM_THREADS=$omp"; stat
  if [ "$omp" = default ]; then python tools/epoch_sync_probe.py; else OMP_NUM_THREADS=$omp MKL_NUM_THREADS=$omp python tools/epoch_sync_probe.py; fi
  stat
done
