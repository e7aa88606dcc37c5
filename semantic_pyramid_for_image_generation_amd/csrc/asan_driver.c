// Host-only driver for asan_host.sh: every C-ABI call here returns before any kernel launch (queries, tuning, argument checks).
#include <stdio.h>
#include <string.h>
#include "sempyr.h"

static int fails = 0;
#define EXPECT(cond, what) do { if (!(cond)) { printf("FAIL %s (%s)\n", what, sp_last_error_string()); ++fails; } } while (0)

int main(void) {
    EXPECT(sp_version() == SP_VERSION, "version");
    // tuning table: every key settable, out-of-range keys rejected
    for (int k = 0; k < SP_TUNE_COUNT; ++k) { EXPECT(sp_set_tuning(k, 1) == SP_OK, "set_tuning"); EXPECT(sp_set_tuning(k, -1) == SP_OK, "reset"); }
    EXPECT(sp_set_tuning(-1, 0) == SP_ERR_INVALID && sp_set_tuning(SP_TUNE_COUNT, 0) == SP_ERR_INVALID, "set_tuning range");
    EXPECT(strlen(sp_last_error_string()) > 0, "error string");
    // workspace planners over the layer shapes of the step (batch 2 / 20 / 32, channel factors 0.5 / 1 / 2 / 4) and odd ones
    const int batches[] = {1, 2, 3, 20, 32};
    const int chans[] = {3, 8, 16, 64, 65, 72, 128, 136, 256, 264, 512, 520, 768, 1024, 1536};
    const int maps[] = {2, 4, 8, 16, 32, 64, 128, 256};
    long queries = 0;
    for (int dt = 0; dt < 2; ++dt)
        for (unsigned b = 0; b < sizeof(batches) / sizeof(int); ++b)
            for (unsigned i = 0; i < sizeof(chans) / sizeof(int); ++i)
                for (unsigned o = 0; o < sizeof(chans) / sizeof(int); ++o)
                    for (unsigned m = 0; m < sizeof(maps) / sizeof(int); ++m)
                        for (int k = 1; k <= 3; k += 2) {
                            const int e = dt == SP_F32 ? 4 : 8;
                            const int cin_p = (chans[i] + e - 1) / e * e;
                            if ((long)batches[b] * maps[m] * maps[m] * cin_p > (1L << 28)) continue;
                            int64_t bytes = -1, floats = -1;
                            for (int det = 0; det < 2; ++det) {
                                sp_set_tuning(SP_TUNE_DETERMINISTIC, det);
                                EXPECT(sp_conv2d_workspace(batches[b], maps[m], maps[m], cin_p, chans[o], k, dt, &bytes) == SP_OK && bytes >= 0, "conv workspace");
                                EXPECT(sp_conv2d_wgrad_workspace(batches[b], maps[m], maps[m], cin_p, chans[o], k, dt, &floats) == SP_OK && floats >= 0, "wgrad workspace");
                                queries += 2;
                            }
                            sp_set_tuning(SP_TUNE_DETERMINISTIC, -1);
                        }
    // argument-check failure paths of the launching entry points (they return before touching the device)
    sp_conv_params p;
    memset(&p, 0, sizeof p);
    EXPECT(sp_conv2d_igemm(NULL, NULL) == SP_ERR_INVALID, "null params");
    EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "null tensors");
    int dummy;
    p.x = p.w = &dummy; p.y = &dummy;
    p.ksize = 5; EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "ksize");
    p.ksize = 3; p.n = 1; p.h = 8; p.w_ = 32; p.cin_p = 12; p.cout = 64; p.ldy = 64; p.dtype = SP_BF16;
    EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "cin_p multiple");
    p.cin_p = 16; p.ldy = 32; EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "ldy");
    p.ldy = 64; p.dtype = 7; EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "dtype");
    p.dtype = SP_BF16; p.pool2 = 3; EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "pool2 range");
    p.pool2 = 1; p.cout = 24; p.ldy = 24; EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "pool2 shape");
    p.pool2 = 0; p.in_up2 = 1; p.ksize = 1; EXPECT(sp_conv2d_igemm(&p, NULL) == SP_ERR_INVALID, "in_up2 shape");
    int64_t out;
    EXPECT(sp_conv2d_workspace(0, 8, 8, 8, 8, 3, SP_BF16, &out) == SP_ERR_INVALID, "workspace bad dims");
    EXPECT(sp_conv2d_wgrad_workspace(1, 8, 8, 8, 8, 2, SP_BF16, &out) == SP_ERR_INVALID, "wgrad workspace bad ksize");
    printf("asan_driver: %ld planner queries, %d failures\n", queries, fails);
    return fails != 0;
}
