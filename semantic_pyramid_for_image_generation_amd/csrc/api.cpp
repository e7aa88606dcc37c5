// Error plumbing and version of the C ABI (include/sempyr.h).
#include <cstdarg>
#include <cstdio>
#include "../../include/sempyr.h"

static thread_local char g_err[512] = "";

extern "C" void sp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sp_last_error_string(void) { return g_err; }
extern "C" int sp_version(void) { return SP_VERSION; }

static thread_local const char* g_route = "";
extern "C" void sp_note_route(const char* name) { g_route = name; }
extern "C" const char* sp_last_route(void) { return g_route; }

// knobs for tests and A/B runs (-1 = built-in default); the library itself never reads the environment
struct SpTuneInit { int v[SP_TUNE_COUNT]; SpTuneInit() { for (int i = 0; i < SP_TUNE_COUNT; ++i) v[i] = -1; } };
static SpTuneInit g_tune_init;
extern int* const sp_g_tune;
int* const sp_g_tune = g_tune_init.v;
extern "C" int sp_set_tuning(int32_t key, int32_t value) {
    if (key < 0 || key >= SP_TUNE_COUNT) { sp_set_error("sp_set_tuning: unknown key %d", key); return SP_ERR_INVALID; }
    sp_g_tune[key] = value;
    return SP_OK;
}
