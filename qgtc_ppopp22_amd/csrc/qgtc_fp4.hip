// qgtc_fp4.hip — second translation unit of libqgtc_hip.so (compiled in parallel with qgtc_hip.hip): the FP4
// matrix-core kernels for narrow right operands (bitmm_fp4_one / _skinny), the row-block kernels and their launchers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "qgtc.h"

#include "common.hip.h"
#include "bitmm_popcount.hip.h"   // MMShape, requant, the DPP ORs (templates only: nothing is instantiated here)
#include "bitmm_mfma.hip.h"       // expand_word_fp4, or_with_partner_half, vector types
#include "bitmm_fp4_skinny.hip.h"
#include "bitmm_fp4_one.hip.h"
#include "fp4_rowblock.hip.h"
#include "bitmm_fp4_rows.hip.h"
#include "fp4_rbw_common.hip.h"
#include "bitmm_fp4_rbw.hip.h"
#include "launch_common.hip.h"
#include "launch_fp4.hip.h"
