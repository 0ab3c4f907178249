// Micro-benchmark: issue rate of v_mfma_i32_16x16x64_i8 from 1 / 2 waves per SIMD when the A/B
// operands come from several different register tuples (as in ncc_mfma_kernel) instead of one.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NB, int NA>
__global__ __launch_bounds__(512) void k(int iters, int waves_active, int* out) {
    if ((int)(threadIdx.x >> 6) >= waves_active) return;
    v4i a[2], b[8];
    for (int i = 0; i < 2; ++i) a[i] = v4i{(int)threadIdx.x + i, 2, 3, 4};
    for (int i = 0; i < 8; ++i) b[i] = v4i{(int)threadIdx.x * 3 + i, 5, 6, 7};
    v4i c[16];
    for (int i = 0; i < 16; ++i) c[i] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i % NA], b[(i / 2) % NB], c[i], 0, 0, 0);
        // keep the operands "changing" for the compiler without real work
        asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    }
    v4i s = c[0];
    for (int i = 1; i < 16; ++i) s += c[i];
    if (s.x == 0x12345678) out[0] = s.y;
}

template <int NB, int NA>
static void run(const char* name, int* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 10000;
    for (int waves = 4; waves <= 8; waves += 4) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<NB, NA>), dim3(256), dim3(512), 0, 0, iters, waves, d);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double per = best * 1e-3 * 2.1e9 / ((double)iters * 16 * (waves / 4));
        printf("%-28s %d wave(s)/SIMD: %7.3f ms  -> %.2f cycles per MFMA per SIMD (@2.1 GHz)\n", name, waves / 4, best, per);
    }
}

int main() {
    int* d; (void)hipMalloc(&d, 64);
    run<1, 1>("1 B tuple, 1 A tuple", d);
    run<2, 2>("2 B tuples, 2 A tuples", d);
    run<8, 2>("8 B tuples, 2 A tuples", d);
    run<8, 1>("8 B tuples, 1 A tuple", d);
    return 0;
}
