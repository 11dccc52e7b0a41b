// qgtc_chainx.hip — sixth translation unit of libqgtc_hip.so (compiled in parallel with the others): the chain entries at 5 .. 8 bits
// and at 129 .. 256 columns (bitmm_fp4_rbx.hip.h) and their launchers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "qgtc.h"

#include "common.hip.h"
#include "bitmm_popcount.hip.h"   // MMShape (templates only: nothing is instantiated here)
#include "bitmm_mfma.hip.h"       // expand_word_fp4, or_with_partner_half, vector types
#include "fp4_rowblock.hip.h"     // strip_operand
#include "fp4_rbw_common.hip.h"
#include "bitmm_fp4_rbx.hip.h"
#include "launch_common.hip.h"
#include "launch_chainx.hip.h"
