// Error plumbing, version string, per-device launch prerequisites and the debug knobs of libtokensgen_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "common.h"
#include "tokensgen_hip.h"

static thread_local char g_err[512] = "";

extern "C" int tg_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* tg_last_error_string(void) { return g_err; }
extern "C" const char* tg_version(void) { return "tokensgen_hip 0.1 (gfx950)"; }

// ---- per-device prerequisites (common.h) ----
int tg_device_cus(void) {
    static std::atomic<int> cus[256];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int>& slot = cus[dev & 255];
    int v = slot.load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        slot.store(v, std::memory_order_relaxed);
    }
    return v;
}

static unsigned long long* once_word(TgOnce& once, unsigned long long* bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    *bit = 1ull << (dev & 63);
    return &once.mask[(dev >> 6) & 3];
}

bool tg_done_on_device(TgOnce& once) {
    unsigned long long bit;
    const unsigned long long* w = once_word(once, &bit);
    return (__atomic_load_n(w, __ATOMIC_ACQUIRE) & bit) != 0;
}

// called AFTER the attribute call has returned: a second host thread that sees the bit may launch at once.  Two threads that both find the bit
// clear both set the attribute (idempotent) and both mark it.
void tg_mark_on_device(TgOnce& once) {
    unsigned long long bit;
    unsigned long long* w = once_word(once, &bit);
    __atomic_fetch_or(w, bit, __ATOMIC_RELEASE);
}

// ---- debug knobs: dispatch overrides of the cross-check tests; see tg_debug_set in the header ----
namespace {
struct KnobDef { const char* name; long def; };
const KnobDef kKnobs[TG_KNOB_COUNT] = {{"TG_ATTN_PP_MIN_WG", 1024}, {"TG_ATTN_FIXEDM", 1}, {"TG_ATTN_SPLIT", 1}, {"TG_GEMM_W4", 1},
                                        {"TG_CONV_SPLITK", 1},       {"TG_CONV_HALO", 1},  {"TG_CONV_W4", 1}};
std::atomic<long> g_knob[TG_KNOB_COUNT];
std::atomic<unsigned> g_knob_set{0};
}  // namespace

long tg_knob(TgKnob k) { return (g_knob_set.load(std::memory_order_relaxed) >> k) & 1u ? g_knob[k].load(std::memory_order_relaxed) : kKnobs[k].def; }

extern "C" int tg_debug_set(const char* knob, long value) {
    for (int i = 0; knob && i < TG_KNOB_COUNT; ++i)
        if (!strcmp(knob, kKnobs[i].name)) {
            g_knob[i].store(value, std::memory_order_relaxed);
            g_knob_set.fetch_or(1u << i, std::memory_order_relaxed);
            return TG_OK;
        }
    return tg_set_error(TG_ERR_ARG, "tg_debug_set: unknown knob '%s'", knob ? knob : "(null)");
}

extern "C" int tg_debug_get(const char* knob, long* value) {
    for (int i = 0; knob && value && i < TG_KNOB_COUNT; ++i)
        if (!strcmp(knob, kKnobs[i].name)) {
            *value = tg_knob((TgKnob)i);
            return TG_OK;
        }
    return tg_set_error(TG_ERR_ARG, "tg_debug_get: unknown knob '%s'", knob ? knob : "(null)");
}

extern "C" const char* tg_debug_knob_name(int index) { return index >= 0 && index < TG_KNOB_COUNT ? kKnobs[index].name : nullptr; }
