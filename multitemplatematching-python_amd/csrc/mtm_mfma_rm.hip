// ncc_mfma_kernel instantiations, unit 2 of 4: the row-multiplexed tiling (classes of <= 16 templates) in all its forms -
// one or three channels, masked, fused global extremum, packed K, raw mode.
#include "mtm_mfma.hip.h"

namespace mtm {

MfmaFn mfma_kernel_rm(const MfmaSel& s) {
    const int xd = s.exact_div ? 1 : 0, e = s.ext ? 1 : 0;
    if (!s.rm || s.mb != 2) return nullptr;
    if (s.method == kMfRaw) return s.kp ? nullptr : (MfmaFn)ncc_mfma_kernel<2, kMfRaw, false, false, true>;
    if (s.method < 0 || s.method > 5) return nullptr;
    if (s.kp) {                                         // packed K: the normalised methods 1 / 3 / 5 (masked: 1 / 3)
        if (!(s.method & 1)) return nullptr;
        const int m2 = (s.method - 1) / 2;
#define MTM_MF_RMKP(X, E) {ncc_mfma_kernel<2, 1, X, false, true, 1, E, false, true>,                                  \
                          ncc_mfma_kernel<2, 3, X, false, true, 1, E, false, true>,                                  \
                          ncc_mfma_kernel<2, 5, X, false, true, 1, E, false, true>}
#define MTM_MF_RMKPM(X, E) {ncc_mfma_kernel<2, 1, X, true, true, 1, E, false, true>, ncc_mfma_kernel<2, 3, X, true, true, 1, E, false, true>}
#define MTM_MF_RMKP3(X, E) {ncc_mfma_kernel<2, 1, X, false, true, 3, E, false, true>,                                 \
                           ncc_mfma_kernel<2, 3, X, false, true, 3, E, false, true>,                                 \
                           ncc_mfma_kernel<2, 5, X, false, true, 3, E, false, true>}
        static const MfmaFn kMfmaRmKpFns[2][2][3] = {{MTM_MF_RMKP(false, false), MTM_MF_RMKP(true, false)},
                                                     {MTM_MF_RMKP(false, true), MTM_MF_RMKP(true, true)}};
        static const MfmaFn kMfmaRmKpMaskedFns[2][2] = {MTM_MF_RMKPM(false, false), MTM_MF_RMKPM(true, false)};
        static const MfmaFn kMfmaRmKpMaskedExtFns[2] = MTM_MF_RMKPM(false, true);    // reciprocal normalisation only
        static const MfmaFn kMfmaRmKpC3Fns[2][2][3] = {{MTM_MF_RMKP3(false, false), MTM_MF_RMKP3(true, false)},
                                                       {MTM_MF_RMKP3(false, true), MTM_MF_RMKP3(true, true)}};
#undef MTM_MF_RMKP
#undef MTM_MF_RMKPM
#undef MTM_MF_RMKP3
        if (s.masked) return m2 > 1 ? nullptr : (s.ext ? (xd ? nullptr : kMfmaRmKpMaskedExtFns[m2]) : kMfmaRmKpMaskedFns[xd][m2]);
        if (s.ch == 3) return kMfmaRmKpC3Fns[e][xd][m2];
        return kMfmaRmKpFns[e][xd][m2];
    }
#define MTM_MF_RM(X, M) {ncc_mfma_kernel<2, 0, X, M, true>, ncc_mfma_kernel<2, 1, X, M, true>,                      \
                        ncc_mfma_kernel<2, 2, X, M, true>, ncc_mfma_kernel<2, 3, X, M, true>,                      \
                        ncc_mfma_kernel<2, 4, X, false, true>, ncc_mfma_kernel<2, 5, X, false, true>}
    static const MfmaFn kMfmaRmFns[2][2][6] = {{MTM_MF_RM(false, false), MTM_MF_RM(true, false)},
                                               {MTM_MF_RM(false, true), MTM_MF_RM(true, true)}};       // [masked][exact][method]
#undef MTM_MF_RM
#define MTM_MF_RMC3(X) {ncc_mfma_kernel<2, 0, X, false, true, 3>, ncc_mfma_kernel<2, 1, X, false, true, 3>,   \
                       ncc_mfma_kernel<2, 2, X, false, true, 3>, ncc_mfma_kernel<2, 3, X, false, true, 3>,   \
                       ncc_mfma_kernel<2, 4, X, false, true, 3>, ncc_mfma_kernel<2, 5, X, false, true, 3>}
    static const MfmaFn kMfmaRmC3Fns[2][6] = {MTM_MF_RMC3(false), MTM_MF_RMC3(true)};
#undef MTM_MF_RMC3
#define MTM_MF_RMEXT(X) {ncc_mfma_kernel<2, 0, X, false, true, 1, true>, ncc_mfma_kernel<2, 1, X, false, true, 1, true>,   \
                        ncc_mfma_kernel<2, 2, X, false, true, 1, true>, ncc_mfma_kernel<2, 3, X, false, true, 1, true>,   \
                        ncc_mfma_kernel<2, 4, X, false, true, 1, true>, ncc_mfma_kernel<2, 5, X, false, true, 1, true>}
    static const MfmaFn kMfmaRmExtFns[2][6] = {MTM_MF_RMEXT(false), MTM_MF_RMEXT(true)};
#undef MTM_MF_RMEXT
#define MTM_MF_RMEXTC3(X) {ncc_mfma_kernel<2, 0, X, false, true, 3, true>, ncc_mfma_kernel<2, 1, X, false, true, 3, true>,   \
                          ncc_mfma_kernel<2, 2, X, false, true, 3, true>, ncc_mfma_kernel<2, 3, X, false, true, 3, true>,   \
                          ncc_mfma_kernel<2, 4, X, false, true, 3, true>, ncc_mfma_kernel<2, 5, X, false, true, 3, true>}
    static const MfmaFn kMfmaRmExtC3Fns[2][6] = {MTM_MF_RMEXTC3(false), MTM_MF_RMEXTC3(true)};
#undef MTM_MF_RMEXTC3
    // fused global extremum of masked classes: binary uint8 mask, methods 0..3, reciprocal normalisation only
    static const MfmaFn kMfmaRmExtMaskedFns[4] = {ncc_mfma_kernel<2, 0, false, true, true, 1, true>,
                                                  ncc_mfma_kernel<2, 1, false, true, true, 1, true>,
                                                  ncc_mfma_kernel<2, 2, false, true, true, 1, true>,
                                                  ncc_mfma_kernel<2, 3, false, true, true, 1, true>};
    if (s.ext && s.masked) return (s.method > 3 || xd) ? nullptr : kMfmaRmExtMaskedFns[s.method];
    if (s.ext) return s.ch == 3 ? kMfmaRmExtC3Fns[xd][s.method] : kMfmaRmExtFns[xd][s.method];
    if (s.ch == 3) return s.masked ? nullptr : kMfmaRmC3Fns[xd][s.method];
    return kMfmaRmFns[s.masked ? 1 : 0][xd][s.method];
}

}  // namespace mtm
