// Do byte-unaligned ds_read_b128 work on gfx950 (SH_MEM alignment mode) and what do they cost?
// hipcc -O3 --offload-arch=gfx950 ub.hip -o ub && ./ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(int off, int iters, int* out, int* check) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[32 * 1024];
    for (int i = threadIdx.x; i < 32 * 1024; i += 256) smem[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the access pattern of ncc_mfma_kernel: lane (j, q) reads the 16 bytes at chunk j + q (+ byte offset)
    const int j = lane & 15, q = lane >> 4;
    const uint8_t* p = smem + wave * 4096 + (j + q) * 16 + off;
    v4i acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v4i v;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p), "i"(0));
            acc += v;
            p += 336;
        }
        p -= 8 * 336;
    }
    if (acc.x == 0x12345678) out[0] = acc.y;
    if (blockIdx.x == 0 && iters == 1) {
        // correctness: first read of every lane
        v4i v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(smem + wave * 4096 + (j + q) * 16 + off)));
        int ok = 1;
        const uint8_t* b = (const uint8_t*)&v;
        for (int i = 0; i < 16; ++i) ok &= b[i] == (uint8_t)((wave * 4096 + (j + q) * 16 + off + i) * 7 + 3);
        check[threadIdx.x] = ok;
    }
}

int main() {
    int *out, *check;
    hipMalloc(&out, 64); hipMalloc(&check, 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int off = 0; off < 8; ++off) {
        hipMemset(check, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, off, 1, out, check);
        int h[256]; hipMemcpy(h, check, 1024, hipMemcpyDeviceToHost);
        int good = 0; for (int i = 0; i < 256; ++i) good += h[i];
        const int iters = 20000;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * 2), dim3(256), 0, 0, off, iters, out, check);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // per CU: 2 WGs x 4 waves x iters x 8 reads x 1 KiB
        const double bytes_per_cu = 2.0 * 4 * iters * 8 * 1024;
        printf("byte offset %d: %3d/256 lanes correct (%s), %.3f ms -> %.1f B/clk/CU @2.34GHz\n", off, good,
               hipGetErrorString(hipGetLastError()), best, bytes_per_cu / (best * 1e-3 * 2.34e9));
    }
    return 0;
}
