// ncc_mfma_kernel instantiations, unit 3 of 4: plain tiling with three channels (RGB) and / or the fused global extremum
// (N_object == 1: cv2.minMaxLoc inside the epilogue).
#include "mtm_mfma.hip.h"

namespace mtm {

MfmaFn mfma_kernel_ext(const MfmaSel& s) {
    const int xd = s.exact_div ? 1 : 0, mbi = s.mb - 1;
    if (s.rm || s.kp || s.r2 || s.mb < 1 || s.mb > 2 || s.method < 0 || s.method > 5) return nullptr;
#define MTM_MF_C3(MB, X) {ncc_mfma_kernel<MB, 0, X, false, false, 3>, ncc_mfma_kernel<MB, 1, X, false, false, 3>,   \
                         ncc_mfma_kernel<MB, 2, X, false, false, 3>, ncc_mfma_kernel<MB, 3, X, false, false, 3>,   \
                         ncc_mfma_kernel<MB, 4, X, false, false, 3>, ncc_mfma_kernel<MB, 5, X, false, false, 3>}
    static const MfmaFn kMfmaC3Fns[2][2][6] = {{MTM_MF_C3(1, false), MTM_MF_C3(2, false)},
                                               {MTM_MF_C3(1, true), MTM_MF_C3(2, true)}};                 // [exact][MB - 1][method]
#undef MTM_MF_C3
#define MTM_MF_EXT(MB, X) {ncc_mfma_kernel<MB, 0, X, false, false, 1, true>, ncc_mfma_kernel<MB, 1, X, false, false, 1, true>,   \
                          ncc_mfma_kernel<MB, 2, X, false, false, 1, true>, ncc_mfma_kernel<MB, 3, X, false, false, 1, true>,   \
                          ncc_mfma_kernel<MB, 4, X, false, false, 1, true>, ncc_mfma_kernel<MB, 5, X, false, false, 1, true>}
    static const MfmaFn kMfmaExtFns[2][2][6] = {{MTM_MF_EXT(1, false), MTM_MF_EXT(2, false)},
                                                {MTM_MF_EXT(1, true), MTM_MF_EXT(2, true)}};
#undef MTM_MF_EXT
#define MTM_MF_EXTC3(MB, X) {ncc_mfma_kernel<MB, 0, X, false, false, 3, true>, ncc_mfma_kernel<MB, 1, X, false, false, 3, true>,   \
                            ncc_mfma_kernel<MB, 2, X, false, false, 3, true>, ncc_mfma_kernel<MB, 3, X, false, false, 3, true>,   \
                            ncc_mfma_kernel<MB, 4, X, false, false, 3, true>, ncc_mfma_kernel<MB, 5, X, false, false, 3, true>}
    static const MfmaFn kMfmaExtC3Fns[2][2][6] = {{MTM_MF_EXTC3(1, false), MTM_MF_EXTC3(2, false)},
                                                  {MTM_MF_EXTC3(1, true), MTM_MF_EXTC3(2, true)}};
#undef MTM_MF_EXTC3
    // fused global extremum of masked classes (binary uint8 mask, methods 0..3; reciprocal-normalisation builds only:
    // MTM_OPT_EXACT_DIV calls keep the maps + extremum_kernel route)
#define MTM_MF_EXTM(MB) {ncc_mfma_kernel<MB, 0, false, true, false, 1, true>, ncc_mfma_kernel<MB, 1, false, true, false, 1, true>,   \
                        ncc_mfma_kernel<MB, 2, false, true, false, 1, true>, ncc_mfma_kernel<MB, 3, false, true, false, 1, true>}
    static const MfmaFn kMfmaExtMaskedFns[2][4] = {MTM_MF_EXTM(1), MTM_MF_EXTM(2)};
#undef MTM_MF_EXTM
    if (s.ext && s.masked) return (s.method > 3 || xd || s.ch != 1) ? nullptr : kMfmaExtMaskedFns[mbi][s.method];
    if (s.masked) return nullptr;
    if (s.ext) return s.ch == 3 ? kMfmaExtC3Fns[xd][mbi][s.method] : kMfmaExtFns[xd][mbi][s.method];
    return s.ch == 3 ? kMfmaC3Fns[xd][mbi][s.method] : nullptr;
}

}  // namespace mtm
