// ncc_mfma_kernel instantiations, unit 1 of 4: plain tiling (one channel, compile-time method, masked or not; the generic
// run-time-method epilogue), raw mode (slabs, sum I^2 M, the first uint16 byte-plane pass) and the
// uint16 finishing pass.  Built as its own translation unit so that the ~300 instantiations compile in parallel.
#include "mtm_mfma.hip.h"

namespace mtm {

MfmaFn mfma_kernel_plain(const MfmaSel& s) {
    const int xd = s.exact_div ? 1 : 0, mbi = s.mb - 1;
    if (s.mb < 1 || s.mb > 2) return nullptr;
    if (s.method == kMfRaw) {                                   // biased int32 accumulators as they are
        if (s.mb != 2) return nullptr;
        return s.kp ? (MfmaFn)ncc_mfma_kernel<2, kMfRaw, false, false, false, 1, false, false, true>
                    : (MfmaFn)ncc_mfma_kernel<2, kMfRaw, false, false>;
    }
    if (s.method == kMfU16) {                                   // uint16 finishing pass: [packed K][extremum][exact]
        static const MfmaFn kU16Fns[2][2][2] = {
            {{ncc_mfma_kernel<2, kMfU16, false, false>, ncc_mfma_kernel<2, kMfU16, true, false>},
             {ncc_mfma_kernel<2, kMfU16, false, false, false, 1, true>, ncc_mfma_kernel<2, kMfU16, true, false, false, 1, true>}},
            {{ncc_mfma_kernel<2, kMfU16, false, false, false, 1, false, false, true>,
              ncc_mfma_kernel<2, kMfU16, true, false, false, 1, false, false, true>},
             {ncc_mfma_kernel<2, kMfU16, false, false, false, 1, true, false, true>,
              ncc_mfma_kernel<2, kMfU16, true, false, false, 1, true, false, true>}}};
        return s.mb == 2 ? kU16Fns[s.kp ? 1 : 0][s.ext ? 1 : 0][xd] : nullptr;
    }
    if (s.method < -1 || s.method > 5) return nullptr;
    if (s.r2) return nullptr;                                   // (mtm_mfma_rows.hip)
    // [masked][exact][MB - 1][0: generic, 1 + method]; masked classes only reach here with methods 0..3 and one channel
#define MTM_MF_ROW(MB, X, M) {ncc_mfma_kernel<MB, -1, X, false>, ncc_mfma_kernel<MB, 0, X, M>, ncc_mfma_kernel<MB, 1, X, M>, \
                             ncc_mfma_kernel<MB, 2, X, M>, ncc_mfma_kernel<MB, 3, X, M>, ncc_mfma_kernel<MB, 4, X, false>,  \
                             ncc_mfma_kernel<MB, 5, X, false>}
    static const MfmaFn kMfmaFns[2][2][2][7] = {
        {{MTM_MF_ROW(1, false, false), MTM_MF_ROW(2, false, false)}, {MTM_MF_ROW(1, true, false), MTM_MF_ROW(2, true, false)}},
        {{MTM_MF_ROW(1, false, true), MTM_MF_ROW(2, false, true)}, {MTM_MF_ROW(1, true, true), MTM_MF_ROW(2, true, true)}}};
#undef MTM_MF_ROW
    return kMfmaFns[s.masked ? 1 : 0][xd][mbi][1 + s.method];
}

// Dispatch between the four units; same decision order as the launcher has always used.
MfmaFn mfma_kernel(const MfmaSel& s) {
    if (s.method == kMfRaw && s.rm) return mfma_kernel_rm(s);
    if (s.method == kMfRaw || s.method == kMfU16) return mfma_kernel_plain(s);
    if (s.kp) return s.rm ? mfma_kernel_rm(s) : mfma_kernel_kp(s);
    if (s.r2) return mfma_kernel_rows(s);
    if (s.rm) return mfma_kernel_rm(s);
    if (s.ext || s.ch == 3) return mfma_kernel_ext(s);
    return mfma_kernel_plain(s);
}

}  // namespace mtm
