// qgtc_mfma.hip — third translation unit of libqgtc_hip.so (compiled in parallel with the others): the
// 128 x 128-tile matrix-core engine (bitmm_mfma.hip.h) and its launchers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "qgtc.h"

#include "common.hip.h"
#include "bitmm_popcount.hip.h"   // MMShape, requant, the DPP ORs (templates only: nothing is instantiated here)
#include "bitmm_mfma.hip.h"
#include "launch_common.hip.h"
#include "launch_mfma.hip.h"
