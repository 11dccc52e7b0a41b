#!/bin/bash
# Round 6: the throughput-bound half of 5_9_adjmatrix_size.py (M = K = 8192 .. 32768) on the standalone bench, both engines.
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed -o /tmp/kbench tools/kbench.hip
for mk in ${SIZES:-8192 16384 32768}; do
  for n in ${NS:-16 64 256 1024}; do
    reps=20
    MFMA=1 /tmp/kbench $mk $mk $n 1 1 1 $reps | sed 's/^/mfma     /'
    if [ -z "$NOPOP" ]; then /tmp/kbench $mk $mk $n 1 1 1 $reps | sed 's/^/popcount /'; fi
  done
done
