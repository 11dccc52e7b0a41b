#!/bin/bash
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed -DQGTC_STREAM_TUNE -DQGTC_STAMPS"
hipcc $F -o /tmp/kbs tools/kbench.hip
echo "per wave (rows 0-7: workgroup 0, 8-15: workgroup 100; waves 0-3 multiply, 4-7 fetch), s_memtime ticks from the wave's start."
echo "multiplying waves: 2 barrier 0 passed, 4 barrier 5 passed, 6 step (5, 0) done, 8 barrier 6 passed, 10 step (6, 0) done, 11 loop end, 12 tile summed."
echo "fetching waves: 1 group 0 landed, 2 barrier 0 passed, 3 group 5 landed, 4 barrier 5 passed, 7 group 6 landed, 8 barrier 6 passed, 11 loop end, 12 tile summed."
for n in ${NS:-64}; do
echo "== 32768^2 x $n"; SLOTS16=1 MFMA=1 /tmp/kbs 32768 32768 $n 1 1 1 50 | head -18
echo "== 32768^2 x $n no DMA, no zero-step test"; ABL_NODMA=1 NOZS=1 SLOTS16=1 MFMA=1 /tmp/kbs 32768 32768 $n 1 1 1 50 | head -10
echo "== 32768^2 x $n X all zero"; SLOTS16=1 MFMA=1 /tmp/kbs 32768 32768 $n 1 1 1 50 0.0 | head -10
done
echo "== wide kernel 16384 x 16384 x 1024 (k_bitmm_fp4_wide has its own stamps; slot layout differs)"; MFMA=1 /tmp/kbs 16384 16384 1024 1 1 1 20 | head -3
