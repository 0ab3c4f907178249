// Does an H2D copy (+ a small kernel) on a second stream overlap a long compute kernel that owns every
// VGPR?  And how fast is a host memcpy into hipHostMalloc memory?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

__global__ __launch_bounds__(256, 2) void busy(float* out, int iters) {
    float acc[96];
    for (int i = 0; i < 96; ++i) acc[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i] = fmaf(acc[i], 1.0001f, 0.5f);
    float s = 0;
    for (int i = 0; i < 96; ++i) s += acc[i];
    if (s == 1.2345f) out[0] = s;
}
__global__ void touch(const unsigned char* in, unsigned char* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] ^ 0x80;
}

int main() {
    const size_t N = 3840 * 2160;
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned char *d_in, *d_out; float* d_f;
    hipMalloc(&d_in, N); hipMalloc(&d_out, N); hipMalloc(&d_f, 64);
    std::vector<unsigned char> pageable(N, 7);
    void* pinned[3]; const unsigned flags[3] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocWriteCombined};
    const char* fname[3] = {"default", "noncoherent", "writecombined"};
    for (int k = 0; k < 3; ++k) {
        if (hipHostMalloc(&pinned[k], N, flags[k]) != hipSuccess) { pinned[k] = nullptr; continue; }
        auto t0 = clk::now(); std::memcpy(pinned[k], pageable.data(), N); auto t1 = clk::now();
        std::memcpy(pinned[k], pageable.data(), N); auto t2 = clk::now();
        printf("host memcpy 8.3 MB into hipHostMalloc(%s): first %.3f ms, second %.3f ms\n", fname[k], ms(t0, t1), ms(t1, t2));
    }
    void* reg = malloc(N); memset(reg, 1, N);
    { auto t0 = clk::now(); hipError_t e = hipHostRegister(reg, N, hipHostRegisterDefault); auto t1 = clk::now();
      printf("hipHostRegister 8.3 MB: %.3f ms (%s)\n", ms(t0, t1), hipGetErrorString(e));
      auto t2 = clk::now(); std::memcpy(reg, pageable.data(), N); auto t3 = clk::now();
      printf("host memcpy into registered malloc memory: %.3f ms\n", ms(t2, t3)); }
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int iters = 60000;   // ~ a few ms
    for (int mode = 0; mode < 5; ++mode) {
        const void* src = mode == 0 ? (void*)pageable.data() : mode == 1 ? pinned[0] : mode == 2 ? pinned[1] : mode == 3 ? reg : nullptr;
        const char* mname[] = {"pageable", "pinned default", "pinned noncoherent", "registered", "no copy"};
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = clk::now();
            hipLaunchKernelGGL(busy, dim3(256 * 2 * 8), dim3(256), 0, a, d_f, iters);
            hipStreamQuery(a);
            auto t1 = clk::now();
            if (src) {
                hipMemcpyAsync(d_in, src, N, hipMemcpyHostToDevice, b);
                hipLaunchKernelGGL(touch, dim3((N + 255) / 256), dim3(256), 0, b, d_in, d_out, N);
            }
            hipEventRecord(ev, b);
            hipStreamQuery(b);
            auto t2 = clk::now();
            hipEventSynchronize(ev);
            auto t3 = clk::now();
            hipStreamSynchronize(a);
            auto t4 = clk::now();
            if (rep) printf("%-20s launch %.3f | enqueue copy %.3f | copy+touch done at %.3f | busy kernel done at %.3f ms\n",
                            mname[mode], ms(t0, t1), ms(t1, t2), ms(t0, t3), ms(t0, t4));
        }
    }
    return 0;
}
