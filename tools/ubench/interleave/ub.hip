// Does an integer VALU instruction issued by the SAME wave right behind an MFMA execute in the MFMA's
// shadow?  (tools/ubench/overlap.hip answers the question for two different waves: no for int/fp32.)
// Patterns per iteration, one or two waves per SIMD:
//   M   : 8 independent v_mfma_i32_16x16x64_i8
//   V   : 24 independent v_alignbyte_b32
//   MV  : 8 x (1 MFMA + 3 alignbyte), interleaved
//   M+V : 8 MFMA then 24 alignbyte (blocked)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE, int OP>
__global__ __launch_bounds__(512) void k(int iters, int waves_active, int* out) {
    if ((int)(threadIdx.x >> 6) >= waves_active) return;
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {(int)threadIdx.x * 3, 5, 6, 7};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{0, 0, 0, 0};
    unsigned w[25];
    for (int i = 0; i < 25; ++i) w[i] = threadIdx.x * 7 + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 3) {
            if (OP == 0) {
#pragma unroll
                for (int i = 0; i < 24; ++i) w[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], 1);
            } else if (OP == 1) {          // 12 v_pk_mov_b32 (2 dwords each)
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    v2u x = {w[2 * i], w[2 * i + 1]}, y = {w[(2 * i + 2) % 24], w[(2 * i + 3) % 24]}, z;
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(z) : "v"(x), "v"(y));
                    w[2 * i] = z.x; w[2 * i + 1] = z.y;
                }
            } else if (OP == 2) {          // 24 v_mov_b32
#pragma unroll
                for (int i = 0; i < 24; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(w[(i + 5) % 24]));
            } else {                       // 24 v_dot4 (VOP3P, as tools/ubench/overlap.hip)
#pragma unroll
                for (int i = 0; i < 24; ++i) w[i] = __builtin_amdgcn_udot4(w[(i + 1) % 24], w[(i + 2) % 24], w[i], false);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0\n\t"
                             "v_alignbyte_b32 %3, %4, %3, 1\n\t"
                             "v_alignbyte_b32 %5, %6, %5, 1\n\t"
                             "v_alignbyte_b32 %7, %8, %7, 1"
                             : "+v"(c[i]), "+v"(a), "+v"(b), "+v"(w[3 * i]), "+v"(w[3 * i + 1]), "+v"(w[3 * i + 2]),
                               "+v"(w[(3 * i + 3) % 24]), "+v"(w[(3 * i + 4) % 24]), "+v"(w[(3 * i + 5) % 24]));
            }
        }
        asm volatile("" : "+v"(w[0]), "+v"(w[24]));
    }
    v4i s = c[0];
    for (int i = 1; i < 8; ++i) s += c[i];
    unsigned t = 0;
    for (int i = 0; i < 25; ++i) t ^= w[i];
    if (s.x == 0x12345678 && t == 77) out[0] = s.y;
}

template <int MODE, int OP>
static void run(const char* name, int* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 4; waves <= 8; waves += 4) {
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<MODE, OP>), dim3(256), dim3(512), 0, 0, iters, waves, d);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-34s %d wave(s)/SIMD: %7.3f ms -> %6.1f cycles per iteration per wave (@2.34 GHz)\n", name, waves / 4, best,
               best * 1e-3 * 2.34e9 / iters / (waves / 4));
    }
}

int main() {
    int* d; (void)hipMalloc(&d, 64);
    run<0, 0>("M   (8 MFMA)", d);
    run<1, 0>("V   (24 alignbyte)", d);
    run<2, 0>("MV  (8 x [MFMA + 3 alignbyte])", d);
    run<3, 0>("M+V (8 MFMA, then 24 alignbyte)", d);
    run<1, 1>("V   (12 v_pk_mov_b32)", d);
    run<3, 1>("M+V (8 MFMA, then 12 v_pk_mov_b32)", d);
    run<1, 2>("V   (24 v_mov_b32)", d);
    run<3, 2>("M+V (8 MFMA, then 24 v_mov_b32)", d);
    run<1, 3>("V   (24 v_dot4_u32_u8)", d);
    run<3, 3>("M+V (8 MFMA, then 24 v_dot4_u32_u8)", d);
    return 0;
}
