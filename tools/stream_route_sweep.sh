#!/bin/bash
# Round 6: where does the long-K kernel (k_bitmm_fp4_stream; MFMA=1 takes it wherever it applies) beat k_bitmm_fp4_skinny (QGTC_NO_STREAM=1)?
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed"
hipcc $F -o /tmp/kb tools/kbench.hip
for K in 8192 16384 32768; do
for M in 1024 2048 4096 8192 16384 32768 65536; do
for n in 16 32 64 128 256; do
a=$(MFMA=1 /tmp/kb $M $K $n 1 1 1 20 | sed 's/.*: \([0-9.]*\) us.*/\1/')
b=$(QGTC_NO_STREAM=1 MFMA=1 /tmp/kb $M $K $n 1 1 1 20 | sed 's/.*: \([0-9.]*\) us.*/\1/')
a4=$(QGTC_STREAM_RF=4 MFMA=1 /tmp/kb $M $K $n 1 1 1 20 | sed 's/.*: \([0-9.]*\) us.*/\1/')
a2=$(QGTC_STREAM_RF=2 MFMA=1 /tmp/kb $M $K $n 1 1 1 20 | sed 's/.*: \([0-9.]*\) us.*/\1/')
c=$(AUTO=1 /tmp/kb $M $K $n 1 1 1 20 | sed 's/.*: \([0-9.]*\) us.*/\1/')
echo "M=$M K=$K N=$n  stream $a (rf4 $a4 rf2 $a2)  skinny $b  auto $c"
done; done; done
