// lm.cpp - UniSE decoder-only AR-LM (placeholder until the generate loop lands; fails loudly, never falls back).
#include "host_util.h"

using namespace qa;

struct qa_lm {
    int unused;
};

extern "C" {

int qa_lm_create(qa_lm** out, const qa_lm_spec*, const qa_tensor*, int64_t, int) {
    if (out) *out = nullptr;
    set_error("qa_lm_create: the UniSE LM path is not built into this library yet");
    return QA_ERR_UNSUPPORTED;
}
void qa_lm_destroy(qa_lm* lm) { delete lm; }
int qa_lm_generate(qa_lm*, int32_t, const float*, int64_t, const float*, int64_t, int64_t, int32_t, int32_t, float,
                   int32_t, float, int64_t*, int64_t*, void*) {
    set_error("qa_lm_generate: the UniSE LM path is not built into this library yet");
    return QA_ERR_UNSUPPORTED;
}
}
