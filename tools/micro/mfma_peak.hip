// mfma_peak.hip - sustained fp32 MFMA rate of this box (pure register loop, no memory): the ceiling the implicit-GEMM kernel
// is priced against in DESIGN.md.  build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, long long* clk) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f - threadIdx.x * 1e-3f;
    long long t0 = __builtin_readcyclecounter();
    long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long m1 = __builtin_amdgcn_s_memtime();
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = m1 - m0;
    }
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    long long* clk;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, int blocks, int iters, int nacc, double flop_per_mfma, auto launch) {
        launch(blocks, iters);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            launch(blocks, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[2];
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double fl = (double)blocks * 4 * iters * nacc * flop_per_mfma;
            printf("%-28s blocks=%4d  %8.3f ms  %7.1f TFLOP/s   shader clk %.0f MHz (block 0: %lld cyc, memtime %lld)\n", name, blocks, ms,
                   fl / ms / 1e9, h[0] / (ms * 1e3), h[0], h[1]);
        }
    };
    for (int blocks : {256, 512, 1024, 2048}) {
        run("32x32x2 4acc", blocks, 20000 * 512 / blocks, 4, 4096.0, [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it, clk); });
    }
    run("32x32x2 2acc", 512, 40000, 2, 4096.0, [&](int b, int it) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(256), 0, 0, out, it, clk); });
    run("32x32x2 1acc", 512, 80000, 1, 4096.0, [&](int b, int it) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(256), 0, 0, out, it, clk); });
    run("16x16x4 8acc", 512, 40000, 8, 2048.0, [&](int b, int it) { hipLaunchKernelGGL(k16<8>, dim3(b), dim3(256), 0, 0, out, it); });
    // long sustained run (~2 s) to see the clock settle
    run("32x32x2 4acc sustained", 512, 2000000, 4, 4096.0, [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it, clk); });
    return 0;
}
