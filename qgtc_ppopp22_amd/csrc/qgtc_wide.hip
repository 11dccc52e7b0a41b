// qgtc_wide.hip — fourth translation unit of libqgtc_hip.so (compiled in parallel with the others): the FP4 matrix-core
// kernel for wide right operands (bitmm_fp4_wide.hip.h) and its launcher.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "qgtc.h"

#include "common.hip.h"
#include "bitmm_popcount.hip.h"   // MMShape (templates only: nothing is instantiated here)
#include "bitmm_mfma.hip.h"       // vector types
#include "fp4_rowblock.hip.h"  // requant_pack16
#include "fp4_expand.hip.h"
#include "bitmm_fp4_wide.hip.h"
#include "launch_common.hip.h"
#include "launch_wide.hip.h"
