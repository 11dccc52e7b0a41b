#!/bin/bash
# Round 6: the long-K kernel on the standalone bench (timing-only switches under -DQGTC_STREAM_TUNE)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed -DQGTC_STREAM_TUNE"
hipcc $F -o /tmp/kb tools/kbench.hip
for S in ${SIZES:-32768}; do
for n in ${NS:-64}; do
echo "== $S x $S x $n"
echo -n "default                             "; MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "X all zero (zero skip)              "; MFMA=1 /tmp/kb $S $S $n 1 1 1 20 0.0
echo -n "no DMA, NOZS                        "; NOZS=1 ABL_NODMA=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "no DMA, no barrier, NOZS            "; NOZS=1 ABL_NODMA=1 ABL_NOBAR=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "no DMA, no fragment reads, NOZS     "; NOZS=1 ABL_NODMA=1 ABL_NOLDS=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
echo -n "no DMA, no reads, no barrier, NOZS  "; NOZS=1 ABL_NODMA=1 ABL_NOLDS=1 ABL_NOBAR=1 MFMA=1 /tmp/kb $S $S $n 1 1 1 20
done
done
