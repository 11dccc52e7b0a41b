#!/bin/bash
# Round 6: in-kernel stamps of the generic AND + popcount kernel on the two product shapes of a --bit_width 32 epoch (hidden 16)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-pass-failed"
hipcc $F -DQGTC_STAMPS -o /tmp/kbs tools/kbench.hip
hipcc $F -o /tmp/kb tools/kbench.hip
echo "stamps: 0 start, 1 first stage known, 2 its loads issued, 3 landed, 4 in LDS, 5 multiplied, 7 loop end, 8 partial tiles stored, 9 barrier, 11 summed, 12 re-quantised, 14 planes stored, 15 end"
echo "== X.W: 1213 x 128 x 16, 32 x 32 planes (3 and 1 non-zero)"
XPLANES=3 WPLANES=1 /tmp/kb 1213 128 16 32 32 32 200
XPLANES=3 WPLANES=1 /tmp/kbs 1213 128 16 32 32 32 20 | head -4
echo "== A.T: 1213 x 1213 x 16, 1 x 32 planes (10 non-zero)"
WPLANES=10 /tmp/kb 1213 1213 16 1 32 32 200 0.005
WPLANES=10 /tmp/kbs 1213 1213 16 1 32 32 20 0.005 | head -4
