#!/bin/bash
# Round 6: the long-K kernel on the standalone bench (timing-only switches under -DQGTC_STREAM_TUNE)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed -DQGTC_STREAM_TUNE"
hipcc $F -o /tmp/kb tools/kbench.hip
for S in ${SIZES:-32768}; do
for n in ${NS:-64}; do
echo "== $S x $S x $n"
echo -n "default                             "; MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "all-ones X                          "; MFMA=1 /tmp/kb $S $S $n 1 1 1 20 1.0
echo -n "no zero-step test (NOZS)            "; NOZS=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "X all zero (every step skipped)     "; MFMA=1 /tmp/kb $S $S $n 1 1 1 20 0.0
echo -n "X out of L2 (ABL_X)                 "; ABL_X=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "no DMA at all (ABL_NODMA), NOZS     "; NOZS=1 ABL_NODMA=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "128-row tiles (QGTC_STREAM_RF=4)    "; QGTC_STREAM_RF=4 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "64-row tiles (QGTC_STREAM_RF=2)     "; QGTC_STREAM_RF=2 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
done
done
