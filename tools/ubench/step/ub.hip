// Cycles per K step of the shipped inline-asm step (32 MFMAs + operand shifts), operands in registers only:
// what the step costs without loads, waits and loop control.  1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
namespace mtm {
template <int MB>
__device__ __forceinline__ void mfma_step(v4i (&acc)[MB][16], const v4i qa, const v4i qb, const v4i (&a)[MB]);
#include "../../../multitemplatematching-python_amd/csrc/mtm_mfma_step_asm.inc"
}

template <int V>
__global__ __launch_bounds__(512, 1) void k(int iters, int waves_active, int* out) {
    if ((int)(threadIdx.x >> 6) >= waves_active) return;
    v4i acc[2][16];
    for (int m = 0; m < 2; ++m)
        for (int c = 0; c < 16; ++c) acc[m][c] = v4i{0, 0, 0, 0};
    v4i qa = {(int)threadIdx.x, 2, 3, 4}, qb = {5, 6, 7, 8};
    v4i a[2] = {{1, 2, 3, 4}, {5, 6, 7, (int)threadIdx.x}};
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) mtm::mfma_step<2>(acc, qa, qb, a);
        else if constexpr (V == 1) mtm::mfma_step2_fused(acc, qa, qb, a);
        else mtm::mfma_step2_fused_b(acc, qa, qb, a);
        asm volatile("" : "+v"(qa), "+v"(qb), "+v"(a[0]), "+v"(a[1]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int s = 0;
    for (int m = 0; m < 2; ++m)
        for (int c = 0; c < 16; ++c) s += acc[m][c].x ^ acc[m][c].w;
    if (s == 0x7fffffff) out[0] = s;
}

int main() {
    int* d; (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int v = 0; v < 3; ++v)
    for (int waves = 4; waves <= 8; waves += 4) {
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, iters, waves, d);
            else if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, iters, waves, d);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, iters, waves, d);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double cyc = best * 1e-3 * 2.34e9 / iters / (waves / 4);
        printf("asm K step variant %d (0 four statements, 1 one statement, 2 one statement + direct F), %d wave(s)/SIMD: %.3f ms -> %.0f cycles per step per wave = %.2f cycles per MFMA at 2.34 GHz (%s)\n", v, waves / 4,
               best, cyc, cyc / 32, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
