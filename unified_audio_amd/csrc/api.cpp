// api.cpp - error channel and the kernel-level C-ABI entry points.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace qa {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace qa

namespace qa {
// ---- live GEMM profiler: (start, stop) event pairs per launch, reduced per tile configuration on read-out
struct ProfRec {
    hipEvent_t a, b;
    int cfg;
    double flops, bytes;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t prof_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
bool profile_enabled() { return g_prof_on; }
static bool g_serial = [] {
    const char* e = getenv("QA_SERIAL");
    return e && *e && *e != '0';
}();
bool serial_mode() { return g_serial; }
void profile_record_begin(int cfg, double flops, double bytes, hipStream_t s) {
    ProfRec r{prof_event(), prof_event(), cfg, flops, bytes};
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}
void profile_record_end(hipStream_t s) { (void)hipEventRecord(g_prof.back().b, s); }
}  // namespace qa

using namespace qa;

extern "C" {

int qa_set_serial(int on) {
    g_serial = on != 0;
    return QA_OK;
}

int qa_profile_begin(void) {
    for (auto& r : g_prof) {
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_prof.clear();
    g_prof_on = true;
    return QA_OK;
}

// out[cfg*4 + {0,1,2,3}] = {algorithmic FLOPs, elapsed ms, launches, algorithmic bytes} for cfg in {128x32, 128x64, 128x128}
int qa_profile_end(double* out, int32_t n_out) {
    g_prof_on = false;
    if (!out || n_out < PROF_NCFG * 4) {
        set_error("qa_profile_end: need room for %d doubles", PROF_NCFG * 4);
        return QA_ERR_INVALID;
    }
    for (int i = 0; i < PROF_NCFG * 4; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        QA_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        QA_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        out[r.cfg * 4 + 0] += r.flops;
        out[r.cfg * 4 + 1] += ms;
        out[r.cfg * 4 + 2] += 1.0;
        out[r.cfg * 4 + 3] += r.bytes;
    }
    return QA_OK;
}

int qa_sconv_geometry(int64_t L, int32_t ksize, int32_t stride, int64_t* T_out, int32_t* pad_left, int32_t* pad_right) {
    QA_REQUIRE(L > 0 && ksize >= stride && stride >= 1 && T_out && pad_left && pad_right, "qa_sconv_geometry: bad argument");
    const int pad_total = ksize - stride;
    const int64_t t = ceil_div(L, stride);
    *T_out = t;
    *pad_right = pad_total / 2 + (int32_t)(t * stride - L);
    *pad_left = pad_total - pad_total / 2;
    return QA_OK;
}

int64_t qa_resolve_frame(int64_t r, int64_t L, int32_t max_pad, int32_t pad_mode) {
    const int Lp = (L <= max_pad) ? max_pad + 1 : (int)L;
    return resolve_frame((int)r, (int)L, Lp, pad_mode);
}

int qa_version(void) { return QA_VERSION; }
const char* qa_last_error(void) { return g_err; }

int qa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int qa_rvq_search(const float* x, int64_t n_vec, const float* codebooks, int32_t Q, int32_t K, int32_t D,
                  int64_t* indices, float* quantized_out, void* stream) {
    if (!x || !codebooks || !indices) {
        set_error("qa_rvq_search: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(n_vec >= 0 && Q > 0 && K > 0 && D > 0, "qa_rvq_search: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* e2 = nullptr;  // [Q*K] code norms followed by the search workspace
    const size_t e2n = (size_t)round_up((int64_t)Q * K, 64);
    QA_HIP(hipMallocAsync(reinterpret_cast<void**>(&e2), sizeof(float) * (e2n + rvq_scratch_floats(n_vec, K, D)), s));
    int st = launch_rvq_norms(codebooks, e2, Q * K, D, s);
    if (st == QA_OK)
        st = launch_rvq_search(x, n_vec, codebooks, e2, Q, K, D, reinterpret_cast<long long*>(indices), quantized_out, D, e2 + e2n, s);
    (void)hipFreeAsync(e2, s);
    return st;
}

int qa_rvq_lookup(const int64_t* indices, int64_t n_vec, const float* codebooks, int32_t Q, int32_t K, int32_t D,
                  float* out, void* stream) {
    if (!indices || !codebooks || !out) {
        set_error("qa_rvq_lookup: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(n_vec >= 0 && Q > 0 && K > 0 && D > 0, "qa_rvq_lookup: bad shape");
    return launch_rvq_lookup(reinterpret_cast<const long long*>(indices), n_vec, codebooks, Q, K, D, out, D,
                             static_cast<hipStream_t>(stream));
}

int qa_conv1d_cl(const qa_conv_args* args, void* stream) {
    if (!args) {
        set_error("qa_conv1d_cl: null argument");
        return QA_ERR_INVALID;
    }
    ConvParams p;
    QA_TRY(conv_params_from_args(*args, &p));
    return launch_conv_gemm(p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
