// Cycles per MFMA of the inline-asm K step with two (shipped two-row variant) and three MFMA groups (three-row variant:
// 48 MFMAs share one set of operand shifts), operands in registers only, 1 and 2 waves per SIMD.  Random-looking operands
// (the chip clocks to its power budget; constants would flatter both).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
namespace mtm {
template <int MB>
__device__ __forceinline__ void mfma_step(v4i (&acc)[MB][16], const v4i qa, const v4i qb, const v4i (&a)[MB]);
#include "../../../multitemplatematching-python_amd/csrc/mtm_mfma_step_asm.inc"
}

template <int MB>
__global__ __launch_bounds__(512, 1) void k(int iters, int waves_active, int* out, unsigned long long* cyc) {
    if ((int)(threadIdx.x >> 6) >= waves_active) return;
    v4i acc[MB][16];
    for (int m = 0; m < MB; ++m)
        for (int c = 0; c < 16; ++c) acc[m][c] = v4i{0, 0, 0, 0};
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (int)h; };
    v4i qa = {rnd(), rnd(), rnd(), rnd()}, qb = {rnd(), rnd(), rnd(), rnd()};
    v4i a[MB];
    for (int m = 0; m < MB; ++m) a[m] = v4i{rnd(), rnd(), rnd(), rnd()};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        mtm::mfma_step<MB>(acc, qa, qb, a);
        // operands change from step to step (a rotation is enough to keep the multipliers toggling)
        qa = v4i{qa.y, qa.z, qa.w, qb.x};
        qb = v4i{qb.y, qb.z, qb.w, qa.x ^ it};
        asm volatile("" : "+v"(qa), "+v"(qb));
#pragma unroll
        for (int m = 0; m < MB; ++m) asm volatile("" : "+v"(a[m]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int m = 0; m < MB; ++m)
        for (int c = 0; c < 16; ++c) s += acc[m][c].x ^ acc[m][c].w;
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 128) cyc[0] = t1 - t0;
}

template <int MB>
void run(const char* name, int* d, unsigned long long* dc) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 4; waves <= 8; waves += 4) {
        float best = 1e9; unsigned long long cy = 0;
        for (int r = 0; r < 4; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<MB>, dim3(256), dim3(512), 0, 0, iters, waves, d, dc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; (void)hipMemcpy(&cy, dc, 8, hipMemcpyDeviceToHost); }
        }
        const double mfmas = (double)iters * 16 * MB * (waves / 4);      // per SIMD
        printf("%s, %d wave(s)/SIMD: %.3f ms, shader cycles per MFMA (per SIMD) %.2f, effective clock %.0f MHz, %.1f TOP/s (%s)\n", name,
               waves / 4, best, (double)cy / ((double)iters * 16 * MB) / (waves / 4), (double)cy / (best * 1e-3) / 1e6,
               mfmas * 1024 * 32768.0 / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
}

int main() {
    int* d; (void)hipMalloc(&d, 64);
    unsigned long long* dc; (void)hipMalloc(&dc, 64);
    run<2>("two groups  (32 MFMAs + 47 VALU per step)", d, dc);
    run<3>("three groups (48 MFMAs + 47 VALU per step)", d, dc);
    return 0;
}
