// ncc_mfma_kernel instantiations, unit 5: the two-row tiling - 16 templates x 2 consecutive output rows per wave
// (classes of more than 16 templates up to 64 wide, one channel, methods 2..5), plain and with the fused global extremum.
// The headline kernel (4K x 32 templates 64x64, TM_CCOEFF_NORMED) is ncc_mfma_kernel<2, 5, true, false, false, 1, false, true>.
#include "mtm_mfma.hip.h"

namespace mtm {

MfmaFn mfma_kernel_rows(const MfmaSel& s) {
    const int xd = s.exact_div ? 1 : 0;
    if (!s.r2 || s.method < 2 || s.method > 5 || s.masked || s.rm || s.kp || s.ch != 1) return nullptr;
    // [extremum][exact][method - 2]
#define MTM_MF_ROWS(MB, X, E) {ncc_mfma_kernel<MB, 2, X, false, false, 1, E, true>, ncc_mfma_kernel<MB, 3, X, false, false, 1, E, true>,   \
                              ncc_mfma_kernel<MB, 4, X, false, false, 1, E, true>, ncc_mfma_kernel<MB, 5, X, false, false, 1, E, true>}
    static const MfmaFn kMfmaR2Fns[2][2][4] = {{MTM_MF_ROWS(2, false, false), MTM_MF_ROWS(2, true, false)},
                                               {MTM_MF_ROWS(2, false, true), MTM_MF_ROWS(2, true, true)}};
#undef MTM_MF_ROWS
    if (s.mb == 2) return kMfmaR2Fns[s.ext ? 1 : 0][xd][s.method - 2];
    return nullptr;
}

}  // namespace mtm
