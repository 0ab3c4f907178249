// ncc_mfma_kernel instantiations, unit 4 of 4: packed K on the plain tiling (template widths that are not multiples of
// 64; the normalised methods 1 / 3 / 5) - one or three channels, masked, fused global extremum.
#include "mtm_mfma.hip.h"

namespace mtm {

MfmaFn mfma_kernel_kp(const MfmaSel& s) {
    const int xd = s.exact_div ? 1 : 0, mbi = s.mb - 1, e = s.ext ? 1 : 0;
    if (!s.kp || s.rm || s.r2 || s.mb < 1 || s.mb > 2 || s.method < 1 || s.method > 5 || !(s.method & 1)) return nullptr;
    const int m2 = (s.method - 1) / 2;                   // methods 1 / 3 / 5 -> 0 / 1 / 2
#define MTM_MF_KP(MB, X, E) {ncc_mfma_kernel<MB, 1, X, false, false, 1, E, false, true>,                              \
                            ncc_mfma_kernel<MB, 3, X, false, false, 1, E, false, true>,                              \
                            ncc_mfma_kernel<MB, 5, X, false, false, 1, E, false, true>}
    static const MfmaFn kMfmaKpFns[2][2][2][3] = {
        {{MTM_MF_KP(1, false, false), MTM_MF_KP(2, false, false)}, {MTM_MF_KP(1, true, false), MTM_MF_KP(2, true, false)}},
        {{MTM_MF_KP(1, false, true), MTM_MF_KP(2, false, true)}, {MTM_MF_KP(1, true, true), MTM_MF_KP(2, true, true)}}};   // [extremum][exact][MB - 1][m2]
#undef MTM_MF_KP
    // masked (methods 1 / 3; the fused extremum only with the reciprocal normalisation)
#define MTM_MF_KPM(MB, X, E) {ncc_mfma_kernel<MB, 1, X, true, false, 1, E, false, true>, ncc_mfma_kernel<MB, 3, X, true, false, 1, E, false, true>}
    static const MfmaFn kMfmaKpMaskedFns[2][2][2] = {{MTM_MF_KPM(1, false, false), MTM_MF_KPM(2, false, false)},
                                                     {MTM_MF_KPM(1, true, false), MTM_MF_KPM(2, true, false)}};
    static const MfmaFn kMfmaKpMaskedExtFns[2][2] = {MTM_MF_KPM(1, false, true), MTM_MF_KPM(2, false, true)};
#undef MTM_MF_KPM
#define MTM_MF_KP3(MB, X, E) {ncc_mfma_kernel<MB, 1, X, false, false, 3, E, false, true>,                             \
                             ncc_mfma_kernel<MB, 3, X, false, false, 3, E, false, true>,                             \
                             ncc_mfma_kernel<MB, 5, X, false, false, 3, E, false, true>}
    static const MfmaFn kMfmaKpC3Fns[2][2][2][3] = {
        {{MTM_MF_KP3(1, false, false), MTM_MF_KP3(2, false, false)}, {MTM_MF_KP3(1, true, false), MTM_MF_KP3(2, true, false)}},
        {{MTM_MF_KP3(1, false, true), MTM_MF_KP3(2, false, true)}, {MTM_MF_KP3(1, true, true), MTM_MF_KP3(2, true, true)}}};
#undef MTM_MF_KP3
    if (s.masked) return m2 > 1 ? nullptr : (s.ext ? (xd ? nullptr : kMfmaKpMaskedExtFns[mbi][m2]) : kMfmaKpMaskedFns[xd][mbi][m2]);
    if (s.ch == 3) return kMfmaKpC3Fns[e][xd][mbi][m2];
    return kMfmaKpFns[e][xd][mbi][m2];
}

}  // namespace mtm
