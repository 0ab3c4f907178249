// ncc_bf16_kernel instantiations (float32 pixels as two bfloat16 pieces on the bf16 matrix cores): its own translation unit.
#include "mtm_bf16.hip.h"

namespace mtm {

Bf16Fn bf16_kernel(int mb, int np) {
    if (np == 1) return mb == 2 ? (Bf16Fn)ncc_bf16_kernel<2, 1> : mb == 1 ? (Bf16Fn)ncc_bf16_kernel<1, 1> : nullptr;
    return mb == 2 ? (Bf16Fn)ncc_bf16_kernel<2, 3> : mb == 1 ? (Bf16Fn)ncc_bf16_kernel<1, 3> : nullptr;
}

}  // namespace mtm
