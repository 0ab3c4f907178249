/*
 * A plain C99 host of libmtm_hip.so: no Python, no C++ types - the boundary is the C ABI of
 * include/mtm_hip.h.  Plants a 24x24 patch of a synthetic image as template, searches it with
 * TM_CCOEFF_NORMED and prints the hits after NMS.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -o c_host \
 *       -Lmultitemplatematching-python_amd/MTM -lmtm_hip -Wl,-rpath,$PWD/multitemplatematching-python_amd/MTM
 *
 * Exit code 0: found the planted patch; 3: no GPU visible (the library has no CPU fallback); 1: error.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mtm_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != MTM_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mtm_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    enum { ROWS = 240, COLS = 320, TH = 24, TW = 24, PY = 100, PX = 200 };
    static uint8_t image[ROWS][COLS];
    uint32_t s = 12345u;
    int y, x;
    mtm_ctx* ctx = NULL;
    mtm_templ templ;
    mtm_hit hits[64];
    int32_t keep[64];
    int64_t n = 0, n_keep = 0, i;
    int found = 0;

    if (mtm_abi_version() != MTM_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch\n");
        return 1;
    }
    if (mtm_device_count() < 1) {
        fprintf(stderr, "no GPU visible: %s\n", "libmtm_hip has no CPU fallback");
        return 3;
    }
    for (y = 0; y < ROWS; ++y)
        for (x = 0; x < COLS; ++x) {
            s = s * 1664525u + 1013904223u;
            image[y][x] = (uint8_t)(s >> 24);
        }
    memset(&templ, 0, sizeof(templ));
    templ.px = &image[PY][PX];               /* a view into the image: row stride = image row */
    templ.mask = NULL;
    templ.rows = TH;
    templ.cols = TW;
    templ.chans = 1;
    templ.dtype = MTM_U8;
    templ.row_stride = COLS;

    CHECK(mtm_ctx_create(&ctx, 0));
    CHECK(mtm_set_image(ctx, image, ROWS, COLS, 1, MTM_U8, COLS));
    CHECK(mtm_set_templates(ctx, &templ, 1, MTM_TM_CCOEFF_NORMED));
    CHECK(mtm_find_matches(ctx, MTM_PEAKS_LOCAL, 0.5, hits, 64, &n));
    CHECK(mtm_nms(hits, n, 0.5, 0, -1, 0.25, keep, &n_keep));
    for (i = 0; i < n_keep; ++i) {
        const mtm_hit* h = &hits[keep[i]];
        printf("hit: template %d at (x=%d, y=%d, w=%d, h=%d) score %.6f\n", h->templ_idx, h->x, h->y, h->w, h->h,
               (double)h->score);
        if (h->x == PX && h->y == PY && h->score > 0.999f) found = 1;
    }
    mtm_ctx_destroy(ctx);
    return found ? 0 : 1;
}
