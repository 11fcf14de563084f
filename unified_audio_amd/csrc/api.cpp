// api.cpp - error channel and the kernel-level C-ABI entry points.
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include <atomic>
#include <cstring>
#include <mutex>

#include "kernels.h"
#include "knobs.h"

namespace qa {
// ---- the knob table (knobs.h): values start from the environment, qa_set_knob() overrides at run time
namespace {
struct KnobRow {
    const char* name;
    long long def;
    const char* doc;
};
const KnobRow g_knob_rows[K_COUNT] = {
#define QA_KNOB_ROW(id, name, def, doc) {name, (long long)(def), doc},
    QA_KNOB_TABLE(QA_KNOB_ROW)
#undef QA_KNOB_ROW
};
std::atomic<long long> g_knob_val[K_COUNT];  // relaxed: a knob is an independent integer, read at launch time by any host thread
std::once_flag g_knob_once;
void knob_init() {
    for (int i = 0; i < K_COUNT; ++i) {
        const char* e = getenv(g_knob_rows[i].name);
        g_knob_val[i].store((e && *e) ? atoll(e) : g_knob_rows[i].def, std::memory_order_relaxed);
    }
}
}  // namespace
long long knob(Knob k) {
    std::call_once(g_knob_once, knob_init);
    return g_knob_val[k].load(std::memory_order_relaxed);
}
void knob_set(Knob k, long long v) {
    std::call_once(g_knob_once, knob_init);
    g_knob_val[k].store(v, std::memory_order_relaxed);
}
int raise_dynamic_lds(const void* kernel, int bytes) {
    struct Raised { const void* kernel; int dev, bytes; };
    static std::mutex mu;
    static std::vector<Raised> done;  // largest size raised so far per (kernel, device): a later, larger request raises again
    int dev = 0;
    QA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (Raised& r : done)
        if (r.kernel == kernel && r.dev == dev) {
            if (bytes <= r.bytes) return QA_OK;
            QA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            r.bytes = bytes;
            return QA_OK;
        }
    QA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.push_back(Raised{kernel, dev, bytes});
    return QA_OK;
}
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
    int M, N, K, ksize, flags;  // shape of the launch (QA_GEMM_SHAPES report)
};
static bool g_prof_on = false;   // conv_gemm launches
static bool g_prof_hbm = false;  // byte-bound kernels (HbmKind): recorded as cfg = PROF_NCFG + kind
static const char* const g_hbm_names[HK_NKINDS] = {"rownorm_kernel (RMSNorm / LayerNorm)", "dwconv_kernel (+ LayerNorm)", "gn_partial + gn_apply (GroupNorm)",
                                                   "rope_kernel", "istft_spec_kernel", "istft_ola_kernel", "stft_post_kernel", "rvq_lookup_kernel",
                                                   "rvq_pick_kernel", "seanet_block_kernel<32> (fused conv0 + block)", "conv_in_kernel"};
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
bool profile_hbm_enabled() { return g_prof_hbm; }
HbmProf::HbmProf(int kind, double bytes, hipStream_t stream) : s(stream), on(g_prof_hbm) {
    if (on) profile_record_begin(PROF_NCFG + kind, 0.0, bytes, s, nullptr);
}
HbmProf::~HbmProf() {
    if (on) profile_record_end(s);
}
bool serial_mode() { return knob(K_SERIAL) != 0; }
void profile_record_begin(int cfg, double flops, double bytes, hipStream_t s, const ConvParams* p) {
    ProfRec r{prof_event(), prof_event(), cfg, flops, bytes, 0, 0, 0, 0, 0};
    if (p) {
        r.M = p->M; r.N = p->N; r.K = p->K; r.ksize = p->ksize;
        r.flags = (p->res ? 1 : 0) | (p->gate ? 2 : 0) | (p->prologue ? 4 : 0) | (p->act ? 8 : 0);
    }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}
void profile_record_end(hipStream_t s) { (void)hipEventRecord(g_prof.back().b, s); }
}  // namespace qa

using namespace qa;

extern "C" {

int qa_set_serial(int on) {
    knob_set(K_SERIAL, on != 0);
    return QA_OK;
}

int qa_knob_count(void) { return K_COUNT; }

int qa_knob_info(int32_t index, const char** name, int64_t* value, int64_t* default_value, const char** doc) {
    QA_REQUIRE(index >= 0 && index < K_COUNT, "qa_knob_info: index %d out of range", index);
    if (name) *name = g_knob_rows[index].name;
    if (value) *value = knob((Knob)index);
    if (default_value) *default_value = g_knob_rows[index].def;
    if (doc) *doc = g_knob_rows[index].doc;
    return QA_OK;
}

static int knob_index(const char* name) {
    if (name)
        for (int i = 0; i < K_COUNT; ++i)
            if (std::strcmp(name, g_knob_rows[i].name) == 0) return i;
    set_error("unknown knob '%s' (qa_knob_info enumerates them)", name ? name : "(null)");
    return -1;
}

int qa_set_knob(const char* name, int64_t value) {
    const int i = knob_index(name);
    if (i < 0) return QA_ERR_INVALID;
    knob_set((Knob)i, value);
    return QA_OK;
}

int qa_get_knob(const char* name, int64_t* value) {
    const int i = knob_index(name);
    if (i < 0 || !value) return QA_ERR_INVALID;
    *value = knob((Knob)i);
    return QA_OK;
}

int qa_profile_begin_ex(int32_t mask) {
    for (auto& r : g_prof) {
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_prof.clear();
    g_prof_on = (mask & 1) != 0;
    g_prof_hbm = (mask & 2) != 0;
    return QA_OK;
}
int qa_profile_begin(void) { return qa_profile_begin_ex(1); }

int qa_profile_hbm_kinds(void) { return HK_NKINDS; }
const char* qa_profile_hbm_name(int32_t kind) { return (kind >= 0 && kind < HK_NKINDS) ? g_hbm_names[kind] : ""; }

// out[kind*3 + {0,1,2}] = {algorithmic bytes, elapsed ms, launches} of the byte-bound kernels recorded since qa_profile_begin_ex(2 | ..);
// call BEFORE qa_profile_end (which closes the recording); does not clear.
int qa_profile_end_hbm(double* out, int32_t n_out) {
    if (!out || n_out < HK_NKINDS * 3) {
        set_error("qa_profile_end_hbm: need room for %d doubles", HK_NKINDS * 3);
        return QA_ERR_INVALID;
    }
    for (int i = 0; i < HK_NKINDS * 3; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        if (r.cfg < PROF_NCFG) continue;
        QA_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        QA_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        const int k = r.cfg - PROF_NCFG;
        out[k * 3 + 0] += r.bytes;
        out[k * 3 + 1] += ms;
        out[k * 3 + 2] += 1.0;
    }
    return QA_OK;
}

// out[cfg*4 + {0,1,2,3}] = {algorithmic FLOPs, elapsed ms, launches, algorithmic bytes} for cfg in {128x32, 128x64, 128x128, 64x128, 64x64}
int qa_profile_end(double* out, int32_t n_out) {
    g_prof_on = false;
    g_prof_hbm = false;
    if (!out || n_out < PROF_NCFG * 4) {
        set_error("qa_profile_end: need room for %d doubles", PROF_NCFG * 4);
        return QA_ERR_INVALID;
    }
    for (int i = 0; i < PROF_NCFG * 4; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        if (r.cfg >= PROF_NCFG) continue;  // byte-bound kernels: qa_profile_end_hbm
        QA_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        QA_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        out[r.cfg * 4 + 0] += r.flops;
        out[r.cfg * 4 + 1] += ms;
        out[r.cfg * 4 + 2] += 1.0;
        out[r.cfg * 4 + 3] += r.bytes;
    }
    // QA_GEMM_SHAPES=<path>: per-shape table of the profiled launches (tuning aid; summaries go to profiles/)
    if (const char* path = g_prof.empty() ? nullptr : getenv("QA_GEMM_SHAPES")) {
        struct Agg { int M, N, K, ksize, flags, cfg; double ms, flops; long n; };
        std::vector<Agg> agg;
        for (auto& r : g_prof) {
            if (r.cfg >= PROF_NCFG) continue;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            Agg* hit = nullptr;
            for (auto& a : agg)
                if (a.M == r.M && a.N == r.N && a.K == r.K && a.ksize == r.ksize && a.flags == r.flags && a.cfg == r.cfg) hit = &a;
            if (!hit) {
                agg.push_back(Agg{r.M, r.N, r.K, r.ksize, r.flags, r.cfg, 0.0, 0.0, 0});
                hit = &agg.back();
            }
            hit->ms += ms;
            hit->flops += r.flops;
            hit->n += 1;
        }
        if (FILE* f = fopen(path, "w")) {
            double tot = 0.0;
            for (auto& a : agg) tot += a.ms;
            fprintf(f, "| M | N | K | ksize | flags(res1 gate2 pro4 act8) | cfg(0=128x32 1=128x64 2=128x128 3=64x128 4=64x64) | launches | total ms | share | avg us | TFLOP/s |\n|---|---|---|---|---|---|---|---|---|---|---|\n");
            for (auto& a : agg)
                fprintf(f, "| %d | %d | %d | %d | %d | %d | %ld | %.3f | %.3f | %.1f | %.1f |\n", a.M, a.N, a.K, a.ksize, a.flags, a.cfg, a.n, a.ms,
                        a.ms / tot, 1e3 * a.ms / a.n, a.flops / (a.ms * 1e-3) / 1e12);
            fclose(f);
        }
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

int qa_rownorm(const float* x, const float* w, const float* b, float* y, int64_t rows, int32_t C, float eps, int32_t mode, void* stream) {
    if (!x || !w || !y || rows < 0) {
        set_error("qa_rownorm: null argument");
        return QA_ERR_INVALID;
    }
    if (rows == 0) return QA_OK;
    if (mode == NORM_RMS) return launch_rmsnorm(x, w, y, rows, C, eps, static_cast<hipStream_t>(stream));
    if (mode == NORM_LAYER) return launch_layernorm(x, w, b, y, rows, C, eps, static_cast<hipStream_t>(stream));
    set_error("qa_rownorm: mode %d (1 RMSNorm, 2 LayerNorm)", mode);
    return QA_ERR_INVALID;
}

int qa_dwconv_cl(const float* x, const float* w_kc, const float* bias, const float* ln_w, const float* ln_b, float* y, int64_t B, int64_t T, int32_t C,
                 int32_t ksize, int32_t pad_left, float eps, void* stream) {
    if (!x || !w_kc || !bias || !y || (ln_w == nullptr) != (ln_b == nullptr)) {
        set_error("qa_dwconv_cl: null argument (ln_w and ln_b come together)");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(B > 0 && T > 0 && B * T < (1LL << 31) && pad_left >= -1, "qa_dwconv_cl: bad shape");
    return launch_dwconv(x, w_kc, bias, ln_w, ln_b, y, (int)B, (int)T, C, ksize, eps, static_cast<hipStream_t>(stream), pad_left);
}

int qa_codes_check_async(const int64_t* codes, int64_t n, int64_t lo, int64_t limit, int64_t* bad_count_dev, void* stream) {
    if (!codes || !bad_count_dev) {
        set_error("qa_codes_check_async: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(n >= 0 && limit > lo, "qa_codes_check_async: bad argument");
    return launch_codes_count(reinterpret_cast<const long long*>(codes), n, lo, limit, reinterpret_cast<unsigned long long*>(bad_count_dev),
                              static_cast<hipStream_t>(stream));
}

int qa_codes_check(const int64_t* codes, int64_t n, int64_t limit, int64_t* bad, void* stream) {
    if (!codes || !bad) {
        set_error("qa_codes_check: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(n >= 0 && limit > 0, "qa_codes_check: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long* dev = nullptr;
    QA_HIP(hipMallocAsync(reinterpret_cast<void**>(&dev), sizeof(unsigned long long), s));
    int st = launch_codes_check(reinterpret_cast<const long long*>(codes), n, limit, dev, s);
    unsigned long long host = 0;
    hipError_t e = hipSuccess;
    if (st == QA_OK) e = hipMemcpyAsync(&host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (st == QA_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFreeAsync(dev, s);
    if (st != QA_OK) return st;
    QA_HIP(e);
    *bad = (int64_t)host;
    return QA_OK;
}

// torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann, lowpass_filter_width = 6, rolloff = 0.99; index arithmetic in
// double, kernel stored as fp32 - what transforms.Resample precomputes) for the reduced ratio orig / new
static void resample_taps(int orig, int nw, std::vector<float>* taps, int* width_out) {
    const double lpw = 6.0, rolloff = 0.99, pi = 3.14159265358979323846;
    const double base = std::min(orig, nw) * rolloff;
    const int width = (int)std::ceil(lpw * orig / base);
    const int kt = 2 * width + orig;
    taps->assign((size_t)nw * kt, 0.f);
    const double scale = base / orig;
    for (int i = 0; i < nw; ++i)
        for (int j = 0; j < kt; ++j) {
            double t = ((double)(-i) / nw + (double)(j - width) / orig) * base;
            t = std::max(-lpw, std::min(lpw, t));
            const double c = std::cos(t * pi / lpw / 2.0);
            const double window = c * c;
            const double tp = t * pi;
            const double sinc = tp == 0.0 ? 1.0 : std::sin(tp) / tp;
            (*taps)[(size_t)i * kt + j] = (float)(sinc * window * scale);
        }
    *width_out = width;
}
static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

int64_t qa_resample_length(int64_t T, int32_t orig_freq, int32_t new_freq) {
    if (T < 0 || orig_freq <= 0 || new_freq <= 0) return QA_ERR_INVALID;
    const int g = gcd_i(orig_freq, new_freq);
    const int64_t o = orig_freq / g, n = new_freq / g;
    return (n * T + o - 1) / o;
}

int qa_resample(const float* wav, int64_t B, int64_t T, int32_t orig_freq, int32_t new_freq, float* out, void* stream) {
    if (!wav || !out) {
        set_error("qa_resample: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(B > 0 && T > 0 && orig_freq > 0 && new_freq > 0, "qa_resample: bad argument");
    const int g = gcd_i(orig_freq, new_freq);
    const int orig = orig_freq / g, nw = new_freq / g;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (orig == nw) {
        QA_HIP(hipMemcpyAsync(out, wav, sizeof(float) * (size_t)B * T, hipMemcpyDeviceToDevice, s));
        return QA_OK;
    }
    std::vector<float> taps;
    int width = 0;
    resample_taps(orig, nw, &taps, &width);
    float* dev = nullptr;
    QA_HIP(hipMallocAsync(reinterpret_cast<void**>(&dev), taps.size() * sizeof(float), s));
    hipError_t e = hipMemcpyAsync(dev, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice, s);
    int st = QA_OK;
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // `taps` is a stack-lifetime host buffer
    if (e == hipSuccess)
        st = launch_resample(wav, dev, out, (int)B, T, qa_resample_length(T, orig_freq, new_freq), orig, nw, width, 2 * width + orig, s);
    (void)hipFreeAsync(dev, s);
    QA_HIP(e);
    return st;
}

}  // extern "C"
