// ncc_bf16_kernel instantiations (float32 pixels as two bfloat16 pieces on the bf16 matrix cores): its own translation unit.
#include "mtm_bf16.hip.h"

namespace mtm {

Bf16Fn bf16_kernel(int mb) { return mb == 2 ? (Bf16Fn)ncc_bf16_kernel<2> : mb == 1 ? (Bf16Fn)ncc_bf16_kernel<1> : nullptr; }

}  // namespace mtm
