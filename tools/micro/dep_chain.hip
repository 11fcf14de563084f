// dep_chain.hip - what ONE dependent stage of a decode step costs on this device, independent of any GEMV code.
// A chain of N kernels on one stream, each an "all-to-all seam" in miniature: every workgroup reads the WHOLE vector its predecessor
// wrote (B x d fp32 = 32 KB, fresh data produced on other CUs), reduces it, and writes its own slice of the next vector.
//   variant 0: empty kernels                      -> launch / dependency overhead of a kernel boundary alone
//   variant 1: read the predecessor's vector      -> + the fresh-data read every seam pays
//   variant 2: variant 1 + a 64 KB weight stream  -> + weights a GEMV column tile streams from HBM (issued together with the x read)
// Reported per kernel, eager launches and one hipGraph replay.  Build: hipcc --offload-arch=gfx950 -O2 tools/micro/dep_chain.hip -o dep_chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int VEC = 16 * 512;  // B = 16 sequences x d = 512

template <int VARIANT>
__global__ __launch_bounds__(512) void stage(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w) {
    if (VARIANT == 0) return;
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc = 0.f;
    float4 wv[8];
    if (VARIANT == 2) {
        const float4* wp = reinterpret_cast<const float4*>(w + (size_t)blockIdx.x * 16384) + tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = wp[i * 512];  // 8 x 512 x 16 B = 64 KB per workgroup, all in flight at once
    }
    const float4* ip = reinterpret_cast<const float4*>(in) + tid;
    float4 xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = ip[i * 512];  // the whole 32 KB vector, one round trip
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
        if (VARIANT == 2) acc += 1e-9f * (wv[i].x + wv[i].y + wv[i].z + wv[i].w + wv[i + 4].x + wv[i + 4].y + wv[i + 4].z + wv[i + 4].w);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid < VEC / (int)gridDim.x) {
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += red[i];
        out[blockIdx.x * (VEC / gridDim.x) + tid] = 1e-6f * s + 0.5f;
    }
}

// variant 2 + PADK x 1 KB of straight-line vector code, ID makes distinct copies at distinct addresses: a chain that cycles through
// several big kernels shows what instruction-cache misses add to a latency-bound stage (the decode step cycles through 5 kernels
// of 7-15 KB per layer)
template <int PADK, int ID>
__global__ __launch_bounds__(512) void stage_padded(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 wv[8], xv[4];
    const float4* wp = reinterpret_cast<const float4*>(w + (size_t)blockIdx.x * 16384) + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = wp[i * 512];
    const float4* ip = reinterpret_cast<const float4*>(in) + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = ip[i * 512];
    float acc = (float)ID;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        acc += xv[i].x + xv[i].y + xv[i].z + xv[i].w + 1e-9f * (wv[i].x + wv[i].y + wv[i].z + wv[i].w + wv[i + 4].x + wv[i + 4].y + wv[i + 4].z + wv[i + 4].w);
    float r0 = acc, r1 = acc + 1.f, r2 = acc + 2.f, r3 = acc + 3.f, r4 = acc + 4.f, r5 = acc + 5.f, r6 = acc + 6.f, r7 = acc + 7.f;
#pragma unroll
    for (int i = 0; i < PADK * 16; ++i) {  // 8 x 8-byte VOP3 = 64 B per trip
        r0 = fmaf(r0, 1.0000001f, 1e-7f); r1 = fmaf(r1, 1.0000002f, 1e-7f); r2 = fmaf(r2, 1.0000003f, 1e-7f); r3 = fmaf(r3, 1.0000004f, 1e-7f);
        r4 = fmaf(r4, 1.0000005f, 1e-7f); r5 = fmaf(r5, 1.0000006f, 1e-7f); r6 = fmaf(r6, 1.0000007f, 1e-7f); r7 = fmaf(r7, 1.0000008f, 1e-7f);
    }
    acc = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid < VEC / (int)gridDim.x) {
        float sacc = 0.f;
        for (int i = 0; i < 8; ++i) sacc += red[i];
        out[blockIdx.x * (VEC / gridDim.x) + tid] = 1e-9f * sacc + 0.5f;
    }
}

template <int PADK>
static int run_padded(int n_kernels, int grid, int distinct, float* a, float* b, const float* w, hipStream_t s) {
    typedef void (*kern_t)(const float*, float*, const float*);
    const kern_t ks[6] = {stage_padded<PADK, 0>, stage_padded<PADK, 1>, stage_padded<PADK, 2>, stage_padded<PADK, 3>, stage_padded<PADK, 4>,
                          stage_padded<PADK, 5>};
    auto chain = [&]() {
        for (int i = 0; i < n_kernels; ++i) hipLaunchKernelGGL(ks[i % distinct], dim3(grid), dim3(512), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, w);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) chain();
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        chain();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("stage + %2d KB of straight-line code, cycling %d distinct kernel(s), grid %3d: %6.2f us per kernel (eager)\n", PADK, distinct, grid,
           1e3f * best / n_kernels);
    return 0;
}

template <int VARIANT>
static int run(const char* name, int n_kernels, int grid, float* a, float* b, const float* w, hipStream_t s) {
    auto chain = [&]() {
        for (int i = 0; i < n_kernels; ++i) hipLaunchKernelGGL(stage<VARIANT>, dim3(grid), dim3(512), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, w);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) chain();
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        chain();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float bestg = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        bestg = ms < bestg ? ms : bestg;
    }
    printf("%-46s grid %3d: %6.2f us per kernel eager, %6.2f us per kernel in a replayed hipGraph (chain of %d)\n", name, grid,
           1e3f * best / n_kernels, 1e3f * bestg / n_kernels, n_kernels);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    return 0;
}

int main() {
    float *a, *b, *w;
    CK(hipMalloc(&a, VEC * 4));
    CK(hipMalloc(&b, VEC * 4));
    CK(hipMalloc(&w, (size_t)256 * 16384 * 4 + 65536));
    CK(hipMemset(a, 0, VEC * 4));
    CK(hipMemset(b, 0, VEC * 4));
    CK(hipMemset(w, 0, (size_t)256 * 16384 * 4 + 65536));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int N = 620;  // ten decode steps' worth of launches
    for (int grid : {128, 256}) {
        if (run<0>("empty kernel", N, grid, a, b, w, s)) return 1;
        if (run<1>("reads the predecessor's 32 KB vector", N, grid, a, b, w, s)) return 1;
        if (run<2>("same + 64 KB of weights per workgroup", N, grid, a, b, w, s)) return 1;
    }
    for (int distinct : {1, 6}) {
        if (run_padded<1>(N, 128, distinct, a, b, w, s)) return 1;
        if (run_padded<8>(N, 128, distinct, a, b, w, s)) return 1;
        if (run_padded<16>(N, 128, distinct, a, b, w, s)) return 1;
    }
    return 0;
}
