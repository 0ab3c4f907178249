// tools/ubench/power asks what the int8 matrix pipe sustains on operands that change every instruction with the tile the
// score kernel uses (v_mfma_i32_16x16x64_i8: 3.96 POP/s at 2.0 GHz).  This one asks the same of v_mfma_i32_32x32x32_i8: twice
// the multiply-accumulates per instruction from the same 2 x 16 operand bytes per lane, i.e. HALF the operand register reads
// per MAC.  If the power budget - not the issue rate - caps the pipe, the larger tile should hold a higher clock.
// 128 accumulator registers per wave either way (8 x v16i against 32 x v4i), two waves per SIMD, every CU busy.
// hipcc -O3 --offload-arch=gfx950 ub.hip -o ub && ./ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int MODE>
__global__ __launch_bounds__(512, 1) void k(int iters, int* out, float* mhz) {
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return MODE == 0 ? 0 : (int)(h ^ (h >> 15)); };
    v4i a[8], b[8];
    for (int s = 0; s < 8; ++s) {
        a[s] = v4i{rnd(), rnd(), rnd(), rnd()};
        b[s] = v4i{rnd(), rnd(), rnd(), rnd()};
    }
    int sum = 0;
    unsigned long long t0, r0, t1, r1;
    if constexpr (SHAPE == 16) {
        v4i acc[32];
        for (int c = 0; c < 32; ++c) acc[c] = v4i{0, 0, 0, 0};
        t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const int s = MODE == 2 ? (c & 7) : 0;
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b[(s + (c >> 3)) & 7], acc[c], 0, 0, 0);
            }
        }
        t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        for (int c = 0; c < 32; ++c) sum += acc[c].x ^ acc[c].w;
    } else {
        v16i acc[8];
        for (int c = 0; c < 8; ++c)
            for (int e = 0; e < 16; ++e) acc[c][e] = 0;
        t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)       // 16 instructions = the MACs of the 32 above
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int s = MODE == 2 ? ((c + 3 * rep) & 7) : 0;
                    acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[(s + rep + (c >> 1)) & 7], acc[c], 0, 0, 0);
                }
        }
        t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        for (int c = 0; c < 8; ++c) sum += acc[c][0] ^ acc[c][15];
    }
    if (sum == 0x7fffffff) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 128 && r1 > r0) mhz[0] = (float)((double)(t1 - t0) * 100.0 / (double)(r1 - r0));
}

template <int SHAPE, int MODE>
void run(const char* name, int* d, float* dm) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 40000;
    float best = 1e9, clk = 0;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(512), 0, 0, iters, d, dm);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; (void)hipMemcpy(&clk, dm, 4, hipMemcpyDeviceToHost); }
    }
    const double ops = (double)iters * 32 * 8 * 256 * 32768.0;      // the same multiply-accumulates per iteration for both shapes
    const double tops = ops / (best * 1e-3) / 1e12;
    printf("%-58s %.3f ms  %7.1f TOP/s  shader clock %.0f MHz (%s)\n", name, best, tops, clk, hipGetErrorString(hipGetLastError()));
}

int main() {
    int* d; (void)hipMalloc(&d, 64);
    float* dm; (void)hipMalloc(&dm, 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 0>("16x16x64, all-zero operands", d, dm);
        run<32, 0>("32x32x32, all-zero operands", d, dm);
        run<16, 2>("16x16x64, random operands changing every MFMA", d, dm);
        run<32, 2>("32x32x32, random operands changing every MFMA", d, dm);
    }
    return 0;
}
