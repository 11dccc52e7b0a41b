#!/bin/bash
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed -DQGTC_STREAM_TUNE -DQGTC_STAMPS"
hipcc $F -o /tmp/kbs tools/kbench.hip
echo "per wave (rows 0-7: workgroup 0, 8-15: workgroup 100), s_memtime ticks from the wave's start: 1 group 0 landed, 2 barrier 0 passed, 3 g5 own DMAs landed, 4 barrier 5 passed, 6 multiply(g4) done, 7 g6 landed, 8 barrier 6 passed, 10 multiply(g5) done, 11 loop end, 12 tile summed"
for n in ${NS:-64}; do
echo "== 32768^2 x $n"; SLOTS16=1 MFMA=1 /tmp/kbs 32768 32768 $n 1 1 1 50 | head -4
echo "== 32768^2 x $n no DMA"; ABL_NODMA=1 NOZS=1 SLOTS16=1 MFMA=1 /tmp/kbs 32768 32768 $n 1 1 1 50 | head -4
echo "== 32768^2 x $n X all zero"; SLOTS16=1 MFMA=1 /tmp/kbs 32768 32768 $n 1 1 1 50 0.0 | head -4
done
echo "== wide kernel 16384 x 16384 x 1024 (k_bitmm_fp4_wide has its own stamps; slot layout differs)"; MFMA=1 /tmp/kbs 16384 16384 1024 1 1 1 20 | head -3
