// Micro-benchmark: do i8 MFMA and VALU (fp64 / fp32 / dot4) work from two waves on the same SIMD
// overlap on gfx950?  Work-group = 512 threads = 8 waves (2 per SIMD): waves 0-3 run role A, waves
// 4-7 run role B.  hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap && ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));

enum { R_NONE = 0, R_MFMA = 1, R_F64 = 2, R_F32 = 3, R_DOT4 = 4, R_F64DIV = 5 };

__device__ __forceinline__ void run_role(int role, int iters, int* out, int seed) {
    if (role == R_MFMA) {
        v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed + 4, 5, 6, 7};
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c7, 0, 0, 0);
        }
        v4i s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        if (s.x == 0x12345678) out[0] = s.y;
    } else if (role == R_F64 || role == R_F64DIV) {
        double x0 = seed * 1e-3 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        const double m = 1.0000001, k = 1e-9;
        for (int i = 0; i < iters; ++i) {
            if (role == R_F64) {
                x0 = fma(x0, m, k); x1 = fma(x1, m, k); x2 = fma(x2, m, k); x3 = fma(x3, m, k);
                x4 = fma(x4, m, k); x5 = fma(x5, m, k); x6 = fma(x6, m, k); x7 = fma(x7, m, k);
            } else {
                x0 = x1 / x0; x1 = x2 / x1; x2 = x3 / x2; x3 = x4 / x3;
                x4 = x5 / x4; x5 = x6 / x5; x6 = x7 / x6; x7 = x0 / x7;
            }
        }
        double s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        if (s == 0.123) out[0] = 1;
    } else if (role == R_F32) {
        float x0 = seed * 1e-3f + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        const float m = 1.0001f, k = 1e-6f;
        for (int i = 0; i < iters; ++i) {
            x0 = fmaf(x0, m, k); x1 = fmaf(x1, m, k); x2 = fmaf(x2, m, k); x3 = fmaf(x3, m, k);
            x4 = fmaf(x4, m, k); x5 = fmaf(x5, m, k); x6 = fmaf(x6, m, k); x7 = fmaf(x7, m, k);
        }
        float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        if (s == 0.123f) out[0] = 1;
    } else if (role == R_DOT4) {
        unsigned a = seed * 2654435761u, b = a ^ 0x55aa55aa;
        unsigned c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_udot4(a, b, c0, false); c1 = __builtin_amdgcn_udot4(a, b, c1, false);
            c2 = __builtin_amdgcn_udot4(a, b, c2, false); c3 = __builtin_amdgcn_udot4(a, b, c3, false);
            c4 = __builtin_amdgcn_udot4(a, b, c4, false); c5 = __builtin_amdgcn_udot4(a, b, c5, false);
            c6 = __builtin_amdgcn_udot4(a, b, c6, false); c7 = __builtin_amdgcn_udot4(a, b, c7, false);
        }
        unsigned s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        if (s == 0x12345678u) out[0] = 1;
    }
}

__global__ __launch_bounds__(512) void k(int roleA, int roleB, int itA, int itB, int* out) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;          // wave-uniform
    run_role(__builtin_amdgcn_readfirstlane(role), wave < 4 ? itA : itB, out, threadIdx.x);
}

static float timeit(int ra, int rb, int ia, int ib, int* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, ra, rb, ia, ib, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    int* d; hipMalloc(&d, 64);
    const char* names[] = {"none", "mfma_i8_16x16x64", "fp64_fma", "fp32_fma", "dot4_u8", "fp64_div"};
    const int it[] = {0, 20000, 40000, 80000, 80000, 4000};   // x8 instructions per iteration
    for (int r = 1; r <= 5; ++r) {
        float t = timeit(r, R_NONE, it[r], 0, d);
        printf("alone  %-18s %8.3f ms  (%.2f cycles/instr/wave @2.1GHz)\n", names[r], t, t * 1e-3 * 2.1e9 / (it[r] * 8.0));
    }
    for (int r = 1; r <= 5; ++r) {
        float t = timeit(r, r, it[r], it[r], d);
        printf("same   %-18s x2 %8.3f ms\n", names[r], t);
    }
    for (int r = 2; r <= 5; ++r) {
        float ta = timeit(R_MFMA, R_NONE, it[1], 0, d), tb = timeit(r, R_NONE, it[r], 0, d);
        float t = timeit(R_MFMA, r, it[1], it[r], d);
        printf("mixed  mfma + %-12s %8.3f ms   alone: %.3f + %.3f -> overlap %.0f%%\n", names[r], t, ta, tb,
               100.0 * (ta + tb - t) / (ta < tb ? ta : tb));
    }
    return 0;
}
