// qgtc_stream.hip — translation unit of libqgtc_hip.so (compiled in parallel with the others): the FP4 matrix-core kernel for
// long K and narrow right operands (bitmm_fp4_stream.hip.h: the throughput-bound shapes of 5_9_adjmatrix_size.py) and its launcher.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "qgtc.h"

#include "common.hip.h"
#include "bitmm_popcount.hip.h"   // requant (templates only: nothing is instantiated here)
#include "bitmm_mfma.hip.h"       // vector types
#include "bitmm_fp4_stream.hip.h"
#include "launch_common.hip.h"
#include "launch_stream.hip.h"
