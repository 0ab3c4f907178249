// DPP wave shifts on gfx950 (rowmax_mask of ncc_mfma_kernel relies on them): wave_shr:1 -> lane i receives lane i - 1,
// wave_shl:1 -> lane i receives lane i + 1, lanes without a source (0 / 63) and lanes whose source is inactive keep `old`.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o, int active_below) {
    const int lane = threadIdx.x;
    o[lane] = o[64 + lane] = -777.f;
    if (lane < active_below) {
        const float v = 100.f + lane;
        const int l = __builtin_amdgcn_update_dpp(__float_as_int(-1.0f), __float_as_int(v), 0x138, 0xf, 0xf, false);
        const int r = __builtin_amdgcn_update_dpp(__float_as_int(-2.0f), __float_as_int(v), 0x130, 0xf, 0xf, false);
        o[lane] = __int_as_float(l);
        o[64 + lane] = __int_as_float(r);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 128 * 4);
    float h[128];
    int bad = 0;
    for (int act : {64, 40}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, act);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int i = 0; i < act; ++i) {
            const float el = i == 0 ? -1.f : 100.f + (i - 1);
            const float er = (i == 63 || i + 1 >= act) ? -2.f : 100.f + (i + 1);
            if (h[i] != el || h[64 + i] != er) {
                if (bad < 8) printf("active %d lane %d: left %.0f (expected %.0f) right %.0f (expected %.0f)\n", act, i, h[i], el, h[64 + i], er);
                ++bad;
            }
        }
    }
    printf("dpp wave shifts: %s (%d mismatches) cycles\n", bad ? "UNEXPECTED" : "as expected", bad);
    return bad ? 1 : 0;
}
