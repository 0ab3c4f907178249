// What does the int8 matrix pipe deliver when NOTHING but MFMAs is issued - and how much of that is the chip's power
// budget?  v_mfma_i32_16x16x64_i8 back to back into 32 independent accumulators, two waves per SIMD, every CU busy, for
// (a) all-zero operands, (b) operands that never change, (c) random operands that change with every instruction (eight
// random register sets taken in turn: what a correlation of white-noise bytes feeds the multipliers).  The score kernel's
// operands are (c).  Prints TOP/s and the shader clock measured in the kernel (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(int iters, int* out, float* mhz) {
    v4i acc[32];
    for (int c = 0; c < 32; ++c) acc[c] = v4i{0, 0, 0, 0};
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return MODE == 0 ? 0 : (int)(h ^ (h >> 15)); };
    v4i a[8], b[8];
    for (int s = 0; s < 8; ++s) {
        a[s] = v4i{rnd(), rnd(), rnd(), rnd()};
        b[s] = v4i{rnd(), rnd(), rnd(), rnd()};
    }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int s = MODE == 2 ? (c & 7) : 0;          // MODE 2: another operand pair for every instruction
            acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b[(s + (c >> 3)) & 7], acc[c], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int c = 0; c < 32; ++c) s += acc[c].x ^ acc[c].w;
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 128 && r1 > r0) mhz[0] = (float)((double)(t1 - t0) * 100.0 / (double)(r1 - r0));
}

template <int MODE>
void run(const char* name, int* d, float* dm) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 40000;
    float best = 1e9, clk = 0;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, d, dm);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; (void)hipMemcpy(&clk, dm, 4, hipMemcpyDeviceToHost); }
    }
    const double ops = (double)iters * 32 * 8 * 256 * 32768.0;      // 8 waves x 256 work-groups x 32 MFMAs x 2 * 16 * 16 * 64
    const double tops = ops / (best * 1e-3) / 1e12;
    printf("%-46s %.3f ms  %7.1f TOP/s  shader clock %.0f MHz  -> %.2f cycles per MFMA and SIMD (%s)\n", name, best, tops, clk,
           clk * 1e6 * (best * 1e-3) / ((double)iters * 32 * 2), hipGetErrorString(hipGetLastError()));
}

int main() {
    int* d; (void)hipMalloc(&d, 64);
    float* dm; (void)hipMalloc(&dm, 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("MFMA only, all-zero operands", d, dm);
        run<1>("MFMA only, constant random operands", d, dm);
        run<2>("MFMA only, random operands changing every MFMA", d, dm);
    }
    return 0;
}
