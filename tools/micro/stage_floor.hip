// stage_floor.hip - the data-movement SKELETON of the UniSE decode step at 16 sequences: the same five
// launches per layer as csrc/lm_decode.hip - same grids, same workgroup sizes, the same bytes pulled per workgroup from the same kind of place
// (fresh activations another launch just wrote / weights that stream from HBM once per step / K-V rows that stream from HBM), every load of a
// wave issued before anything else, one LDS exchange + barrier, the same bytes stored - and NOTHING ELSE: no MFMA, no RoPE, no softmax, no
// epilogue arithmetic.  What a skeleton launch costs is the floor of the real launch on this device: the kernels of lm_decode.hip can only
// differ from it by their arithmetic and their instruction stream.
//
//   launch        grid (x, y[, z = prefetch plane]) x threads   fresh bytes / workgroup      streamed bytes / workgroup   stored / workgroup
//   qkv           96 x 2 (x 2)  x 512                           16 KB (8 rows of x)          32 KB weights                512 B
//   attention     8 x 16 x S    x 512                           256 B (q)                    128 KB K / V (HBM)           272 B
//   o_proj        128 x 2 (x 2) x 512                           32 KB (partial records)      8 KB weights                 128 B
//   fused MLP     128 x 2 (x 2) x 512                           16 KB (8 rows of x)          96 KB weights                16 KB
//   MLP reduce    256           x 64                            16 KB (partials)             -                            128 B
//
// Weights are distinct per layer (12 x 16 MB: more than the L2s and most of the Infinity Cache hold next to 26 MB of K / V per layer), K / V
// distinct per layer (12 x 26 MB).  PF=1: the second z-plane of qkv / o_proj / MLP touches one word per 64 bytes of the NEXT launch's weights
// (csrc/lm_decode.h PfArgs), as the real step does.  Per-kernel times: run under `rocprofv3 --kernel-trace --stats`; the program prints the
// per-step time of the chain.  Build: hipcc --offload-arch=gfx950 -O2 tools/micro/stage_floor.hip -o tools/micro/stage_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                             \
        }                                                                         \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Pf {
    const char* p;
    long long tile_bytes;
    int n_tiles;
};

__device__ __forceinline__ void prefetch_plane(const Pf& pf, int id, int count) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned acc = 0;
    if (pf.p) {
        const long long tb = pf.tile_bytes, step = (long long)nthr * 64;
        for (int j = id; j < pf.n_tiles; j += count) {
            const char* base = pf.p + (long long)j * tb;
            for (long long o = (long long)tid * 64; o < tb; o += 4 * step) {
                unsigned v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long oo = o + u * step;
                    v[u] = oo < tb ? *reinterpret_cast<const unsigned*>(base + oo) : 0u;
                }
                acc += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
    }
    asm volatile("" ::"v"(acc));
}

// One skeleton stage.  NT threads; FV float4 of fresh data per thread (the region is shared by the workgroups of one y-plane when
// fresh_per_wg == 0, private to the workgroup otherwise); WV float4 of streamed data per thread (non-temporal, private to the workgroup);
// OV4 float4 stored per workgroup.  ID only makes distinct kernels (distinct names in the trace, distinct code addresses).
template <int ID, int NT, int FV, int WV>
__global__ __launch_bounds__(NT) void skel(const f32x4* __restrict__ fresh, int fresh_per_wg, const f32x4* __restrict__ w, f32x4* __restrict__ out,
                                            int ov4, const Pf pf) {
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (ID != 1 && __builtin_expect(blockIdx.z != 0, 0)) {  // the attention launch uses z for its key split, it carries no prefetch plane
        prefetch_plane(pf, wg, gridDim.x * gridDim.y);
        return;
    }
    const int wgz = ID == 1 ? wg + gridDim.x * gridDim.y * blockIdx.z : wg;
    __shared__ float red[NT / 64 > 0 ? NT / 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 wv[WV > 0 ? WV : 1], fv[FV > 0 ? FV : 1];
    // a column tile's weights are shared by its row groups (the second reader hits the XCD's L2); K / V rows are private to the workgroup
    const f32x4* wp = w + (size_t)(ID == 1 ? wgz : (int)blockIdx.x) * WV * NT + tid;
#pragma unroll
    for (int i = 0; i < WV; ++i) wv[i] = __builtin_nontemporal_load(wp + i * NT);
    const f32x4* fp = fresh + (size_t)(fresh_per_wg ? wgz : (int)blockIdx.y) * FV * NT + tid;
#pragma unroll
    for (int i = 0; i < FV; ++i) fv[i] = fp[i * NT];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < FV; ++i) acc += fv[i].x + fv[i].y + fv[i].z + fv[i].w;
#pragma unroll
    for (int i = 0; i < WV; ++i) acc += 1e-9f * (wv[i].x + wv[i].y + wv[i].z + wv[i].w);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    const f32x4 o4 = {1e-9f * s + 0.5f, 0.25f, 0.125f, 1e-9f * s};
    for (int i = tid; i < ov4; i += NT) out[(size_t)wgz * ov4 + i] = o4;
}

int main(int argc, char** argv) {
    const int PF = argc > 1 ? atoi(argv[1]) : 1;
    const int steps = argc > 2 ? atoi(argv[2]) : 50;
    const int S = argc > 3 ? atoi(argv[3]) : 2;  // key splits of the attention launch (mean over a 16-segment generate: 2.1)
    const int L = 12;
    const size_t MB = 1 << 20;
    const size_t w_layer = 16 * MB, kv_layer = (size_t)8 * 16 * S * 128 * 1024;
    char *wbuf, *kvbuf;
    f32x4 *a, *b;
    CK(hipMalloc(&wbuf, L * w_layer + MB));
    CK(hipMalloc(&kvbuf, L * kv_layer + MB));
    CK(hipMalloc(&a, 8 * MB));
    CK(hipMalloc(&b, 8 * MB));
    CK(hipMemset(wbuf, 0, L * w_layer + MB));
    CK(hipMemset(kvbuf, 0, L * kv_layer + MB));
    CK(hipMemset(a, 0, 8 * MB));
    CK(hipMemset(b, 0, 8 * MB));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const unsigned pz = PF ? 2 : 1;
    auto step = [&]() {
        f32x4 *in = a, *out = b;
        auto swap = [&]() { f32x4* t = in; in = out; out = t; };
        for (int l = 0; l < L; ++l) {
            const char* wl = wbuf + (size_t)l * w_layer;  // [qkv 3 MB | o 1 MB | mlp 12 MB]
            const char* wn = wbuf + (size_t)((l + 1) % L) * w_layer;
            const f32x4* wq = reinterpret_cast<const f32x4*>(wl);
            const f32x4* wo = reinterpret_cast<const f32x4*>(wl + 3 * MB);
            const f32x4* wm = reinterpret_cast<const f32x4*>(wl + 4 * MB);
            const f32x4* kv = reinterpret_cast<const f32x4*>(kvbuf + (size_t)l * kv_layer);
            Pf none{nullptr, 0, 0};
            Pf pq = PF ? Pf{wl + 3 * MB, 8 * 1024, 128} : none;          // qkv launch -> o_proj weights (128 tiles of 8 KB)
            Pf po = PF ? Pf{wl + 4 * MB, 96 * 1024, 128} : none;         // o_proj launch -> fused-MLP weights (128 slices of 96 KB)
            Pf pm = PF ? Pf{wn, 32 * 1024, 96} : none;                   // MLP launch -> the next layer's qkv weights (96 tiles of 32 KB)
            hipLaunchKernelGGL((skel<0, 512, 2, 4>), dim3(96, 2, pz), dim3(512), 0, s, in, 0, wq, out, 32, pq);       // qkv
            swap();
            hipLaunchKernelGGL((skel<1, 512, 1, 16>), dim3(8, 16, S), dim3(512), 0, s, in, 0, kv, out, 17, none);    // attention
            swap();
            hipLaunchKernelGGL((skel<2, 512, 4, 1>), dim3(128, 2, pz), dim3(512), 0, s, in, 0, wo, out, 8, po);       // o_proj
            swap();
            hipLaunchKernelGGL((skel<3, 512, 2, 12>), dim3(128, 2, pz), dim3(512), 0, s, in, 0, wm, out, 1024, pm);   // fused MLP
            swap();
            hipLaunchKernelGGL((skel<4, 64, 16, 0>), dim3(256, 1, 1), dim3(64), 0, s, in, 1, wq, out, 8, none);      // MLP reduce
            swap();
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) step();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < steps; ++r) step();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("skeleton step (12 layers x 5 launches, prefetch plane %d, S = %d): %.1f us per step, %.2f us per launch\n", PF, S, 1e3f * ms / steps,
           1e3f * ms / steps / (L * 5));
    return 0;
}
