/*
 * mtm_hip.h - C ABI of libmtm_hip.so, the MI355X (gfx950) implementation of the
 * Multi-Template-Matching hot path.
 *
 * The reference (multi-template-matching/MultiTemplateMatching-Python) has no FFI of its own: it
 * is pure Python calling OpenCV / scikit-image / scipy.  The boundary this library sits behind is
 * therefore the set of third-party calls on the reference's hot path; each entry point below
 * names the reference call site(s) it replaces (file:line in the reference tree).  The Python
 * host layer (multitemplatematching-python_amd/MTM) binds these with ctypes and re-exposes the
 * reference's module-level API unchanged; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative MTM_E_* code;
 *     mtm_last_error() returns a thread-local message for the last failure on this thread.
 *   - the caller owns every host buffer for the duration of a call; the library owns all device
 *     memory, freed in mtm_ctx_destroy().
 *   - images are row-major, (rows, cols) or (rows, cols, chans) interleaved, with an explicit row
 *     stride in bytes (numpy views with contiguous pixels are passed without a copy).
 *   - a context is bound to one GPU and is single-caller (not re-entrant); calls block until
 *     their results are on the host unless documented otherwise.
 */
#ifndef MTM_HIP_H
#define MTM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTM_ABI_VERSION 9

/* pixel types (after the dtype policy of MTM/__init__.py:71-74: uint8 stays, all else float32) */
#define MTM_U8  0
#define MTM_F32 1
#define MTM_U16 2   /* the reference casts uint16 to float32 (exactly) before cv2; passing the uint16 pixels as they
                       are gives the same results (every sum is an exact integer) and lets single-channel
                       uint16 image + uint16 templates run on the int8 matrix cores (byte-plane decomposition)
                       instead of the float64 kernel */

/* OpenCV's TemplateMatchModes values, used as raw ints by the reference
 * (MTM/__init__.py:56,95,247 defaults; :78,:216,:227,:232 comparisons) */
#define MTM_TM_SQDIFF        0
#define MTM_TM_SQDIFF_NORMED 1
#define MTM_TM_CCORR         2
#define MTM_TM_CCORR_NORMED  3
#define MTM_TM_CCOEFF        4
#define MTM_TM_CCOEFF_NORMED 5

/* peak extraction mode */
#define MTM_PEAKS_LOCAL  0   /* MTM/__init__.py:231-235: local maxima (methods 2-5) / minima (0,1) */
#define MTM_PEAKS_GLOBAL 1   /* MTM/__init__.py:225-230: N_object == 1, cv2.minMaxLoc              */

/* 3x3 maximum-filter border rule of skimage.feature.peak_local_max (MTM/__init__.py:45; the reference leaves
 * scikit-image unpinned, setup.py:24).  The two rules differ only where the filtered map is negative at its
 * border: local MINIMA (methods 0/1, the map is negated, MTM/__init__.py:53) and negative thresholds. */
#define MTM_BORDER_CONSTANT 0  /* skimage <= 0.18: pad with 0 (border minima are never peaks) */
#define MTM_BORDER_NEAREST  1  /* skimage >= 0.19: replicate the edge - the default */

/* kernel selection for the uint8 score-map path (mtm_set_option MTM_OPT_KERNEL) */
#define MTM_KERNEL_AUTO  0
#define MTM_KERNEL_NAIVE 1   /* one thread per output pixel, scalar loop: the in-library cross-check */
#define MTM_KERNEL_DOT4  2   /* LDS-tiled sliding window on v_dot4_u32_u8 */
#define MTM_KERNEL_MFMA  3   /* implicit-GEMM sliding window on v_mfma_i32_16x16x64_i8 */
#define MTM_KERNEL_MFMA16 4  /* reported in mtm_timing.kernel_used only: uint16 pixels as four byte-plane
                                correlations on the MFMA kernel (selected by MTM_KERNEL_AUTO / _MFMA) */

#define MTM_KERNEL_MFMA_F32 5 /* reported in mtm_timing.kernel_used only: float32 pixels as two bfloat16 pieces on the
                                bf16 matrix cores, ~1e-5 of the normalised score (selected by MTM_KERNEL_AUTO /
                                _MFMA for unmasked float32 image + templates; MTM_F32_MFMA=0: float64 kernel) */

#define MTM_OPT_KERNEL      1
#define MTM_OPT_PEAK_BORDER 2
#define MTM_OPT_HIT_CAPACITY 3
#define MTM_OPT_DOT4_VARIANT 4  /* register-blocking variant of the dot4 kernel (tuning) */
#define MTM_OPT_EXACT_DIV 5     /* 1 (default since round 5: measured free on the hits-only path): IEEE division in the MFMA
                                   epilogue, bit-identical to the other kernels and to the oracle;
                                   0: correctly rounded reciprocals, <= 1 ulp(float32) on ~1e-8 of the pixels (round 1-4's
                                   default; what the fused global extremum of MASKED classes still uses under 1);
                                   2: strict - that extremum too goes through maps + extremum_kernel */
#define MTM_OPT_HITS_ONLY 6     /* 1 (default): mtm_find_matches does not write the score maps to memory when every
                                   template runs the int8 MFMA kernel: in local-extrema mode the peaks come from
                                   the in-kernel candidate list, in global-extremum mode (unmasked 1- or
                                   3-channel templates) the per-template best is kept inside the score kernel;
                                   0: always materialise the maps.  Results are identical either way. */
#define MTM_OPT_F32_MFMA 7      /* float32 images (every non-uint8, non-uint16 input: MTM/__init__.py:71-74), unmasked
                                   templates; the normalised methods, and the raw-sum methods for the global extremum
                                   and for local extrema against a threshold (listed by rigorous per-pixel error bounds
                                   of the sum, re-scored exactly; the raw sums' maps always take the float64 kernel).
                                   1 (default): scores on the bf16 matrix cores (within ~1e-5 of cv2's float64 result)
                                   as a SCREEN - every output whose exact score COULD be a peak, the global extremum or
                                   pass the threshold is re-scored with the float64 arithmetic of the exact kernel.
                                   Since round 5 "could" is decided by a per-output error bound, not by margins:
                                   |bf16 ratio - exact ratio| <= eps sqrt(sum (I - mu)^2) / sq * (sqrt(sum (T - mean)^2) /
                                   templ_norm), large where it has to be (a low-contrast window beside a much brighter
                                   region) and ~5e-5 on textured windows (one hardware assumption, stated in
                                   bf16_rig_eps and measured by tests/test_gpu_parity.py::test_float32_error_bound_holds);
                                   mtm_find_matches then returns the exact kernel's hit lists.  Round 6: where only a list
                                   leaves the kernel (hits-only mode: local extrema against a threshold, the global
                                   extremum, templates with masks) the screen first runs with ONE bfloat16 piece product
                                   instead of three - a third of the matrix-core work, eps ~2^-7 instead of ~2^-15, the
                                   same bound and the same exact re-scoring, so the same lists; a list that overflows
                                   repeats the launch with three products and the next calls start there
                                   (mtm_timing.f32_pieces says which ran).  Score maps read back with mtm_score_map keep
                                   the ~1e-5 tolerance (three products, always);
                                   0: the float64 kernel for everything (10x slower, maps exact to rounding);
                                   2: bf16 scores as they are, no re-scoring;
                                   3: as 1 without the one-product tier;
                                   4: diagnostic - as 2 with one piece product (the screen's raw scores: what the tests
                                   measure its bound on; never a result).  Environment: MTM_F32_MFMA. */

/* error codes */
#define MTM_OK            0
#define MTM_E_INVALID    -1  /* bad argument */
#define MTM_E_HIP        -2  /* a HIP runtime call failed (message has the hipError string) */
#define MTM_E_NO_DEVICE  -3  /* no usable GPU */
#define MTM_E_STATE      -4  /* call order: image / templates not set */
#define MTM_E_OVERFLOW   -5  /* output buffer too small; *n_out holds the required capacity */
#define MTM_E_COMM       -6  /* RCCL failure */

typedef struct mtm_ctx mtm_ctx;

/* One template ("unit": a template, or a rotation/scale variant the caller appended to
 * listTemplates).  mask == NULL for no mask; a mask has the template's shape and dtype
 * (the policy of MTM/__init__.py:76-88 is applied by the host layer before this call). */
typedef struct mtm_templ {
    const void* px;
    const void* mask;
    int32_t rows, cols, chans, dtype;
    int64_t row_stride;       /* bytes */
    int64_t mask_row_stride;  /* bytes */
} mtm_templ;

/* One detection: the record MTM builds at MTM/__init__.py:241 (label replaced by the template's
 * index in the list).  24 bytes; also the record exchanged between ranks. */
typedef struct mtm_hit {
    int32_t templ_idx;
    int32_t x, y, w, h;
    float   score;
} mtm_hit;

/* timing of the last mtm_find_matches call, measured with HIP events on the context's stream */
typedef struct mtm_timing {
    float total_ms;      /* first kernel launch -> last kernel done.  Calls whose image arrives in row bands
                            (mtm_find_matches_image, one bandable size class) start this clock once the FIRST band's copy is
                            on its way - with a pageable source that copy call blocks the host while the rows are staged -
                            so total_ms of a banded call excludes up to that band's upload (25 % of the image by default);
                            unbanded calls include the whole upload.  Wall-clock figures (bench.py `value`) are not affected. */
    float score_ms;      /* window statistics + score-map kernels                        */
    float peaks_ms;      /* peak-extraction kernels                                      */
    float ncc_kernel_ms; /* the dominant score-map kernel(s) alone: time during which at least one launch ran */
    int32_t ncc_launches;
    int32_t kernel_used; /* MTM_KERNEL_* actually dispatched for the uint8 path           */
    int64_t n_hits;
    int32_t hits_only;   /* 1: the last mtm_find_matches ran without materialising the score maps; 2: maps in memory and
                            the peak pass over the flagged row segments only (dense images) */
    float   sclk_mhz;    /* shader clock the score kernel ran at, measured inside it (s_memtime ticks per
                            s_memrealtime tick x 100 MHz) by one mid-grid work-group; 0 when not measured */
    float   ncc_sum_ms;  /* plain sum of the score-kernel launch durations (= ncc_kernel_ms unless launches of a
                            banded call overlapped; what a profiler's per-launch average times the count gives) */
    int32_t f32_route;   /* float32 images on the bf16 matrix cores (MTM_OPT_F32_MFMA = 1), how the exact decisions were
                            reached: 0 not such a call, 1 kernel candidates re-scored, 2 map scan + neighbourhoods
                            re-scored, 3 the float64 kernel after all (lists overflowed, or classes it has to run anyway), 4 (round 6)
                            templates with masks: two raw bf16 correlations as a screen, every output that could pass the threshold
                            re-scored with the float64 kernel's own chains, "below" placeholders elsewhere (maps not published) */
    int32_t sq_launches; /* masked classes on the matrix cores: launches of the sum I^2 M pass (one per masked class) ... */
    float   masked_stat_ms; /* ... and the time they took (sum of the passes' own event pairs; 0 without masked classes) */
    int32_t f32_pieces;  /* (ABI 8) float32 images: bfloat16 piece products of the last matrix-core launch - 3 (scores to ~1e-5), or 1:
                            the one-product screen of the hits-only refined routes (f32_route 1 and 4; listing by a bound of
                            2^-7 of the norms' product, exact re-scoring decides; an overflowing list repeats the launch with 3);
                            0: no such launch */
} mtm_timing;

/* ---- device / context ------------------------------------------------------------------- */
int         mtm_abi_version(void);
int         mtm_device_count(void);                 /* 0 when no GPU is visible */
const char* mtm_last_error(void);
int         mtm_ctx_create(mtm_ctx** out, int device_id);
void        mtm_ctx_destroy(mtm_ctx* ctx);
int         mtm_set_option(mtm_ctx* ctx, int option, int64_t value);
int         mtm_get_option(mtm_ctx* ctx, int option, int64_t* value);     /* the current value of an MTM_OPT_* option */
/* Test support (ABI 7; nothing of the reference's interface corresponds to it).  The reference's functions are pure
 * functions of their arguments (MTM/__init__.py:92, :238-241); a kernel that reads memory it did not write in this call -
 * register-spill slots, LDS, a recycled work buffer - breaks that silently, depending on what earlier launches of the
 * process left behind (round 5's uint16 finding).  This call leaves a byte pattern (0xFF: NaN as float32 /
 * float64; 0x7F: huge finite values) in the places such a read would hit: every wave slot's scratch memory, every CU's
 * LDS, and the context's per-call work buffers.  tests/test_gpu_parity.py runs it ahead of every parity test. */
#define MTM_POISON_SCRATCH 1
#define MTM_POISON_LDS     2
#define MTM_POISON_ARENAS  4
int         mtm_debug_poison(mtm_ctx* ctx, int pattern_byte, int what);
/* Test support (ABI 9).  The IEEE-division epilogues of the single-channel uint8 score kernel obtain (float)(num / t) -
 * the value OpenCV's common_matchTemplate stores, SURVEY 8a-5 - from a reciprocal product plus an integer test that sends the
 * quotients next to a float32 rounding boundary through the division itself (csrc/mtm_device_util.hip.h,
 * quotient_as_float).  This call runs that function against the division on n_cases operand triples shaped like the
 * epilogue's, half of them constructed to straddle a rounding boundary (both instantiations of the function: the general
 * one, which also guards quotients in the float32 denormal range, and the epilogues', whose operands cannot produce
 * those): out4 = {cases run, results that differ in any
 * bit (must be 0), cases that took the division, largest |num * rr - num / t| seen in ulp(double) (the bound in the
 * source is 6, the test's margin 32)}. */
int         mtm_debug_quotient_check(mtm_ctx* ctx, uint64_t n_cases, uint64_t seed, uint64_t* out4);
/* Page-locked host memory for pixel buffers (optional).  The reference's caller hands over whatever numpy holds
 * (MTM/__init__.py:247 `image`) - pageable memory, which the runtime stages through its own pinned buffers while the
 * upload call blocks.  An image kept in memory from mtm_host_alloc crosses PCIe as a plain DMA transfer behind the call
 * (MTM.pinned_empty wraps it as a numpy array).  NULL on failure (message in mtm_last_error). */
void*       mtm_host_alloc(size_t bytes);
void        mtm_host_free(void* p);

/* ---- inputs ------------------------------------------------------------------------------ */
/* Upload the search image (already cropped to searchBox by the host layer, MTM/__init__.py:140-144)
 * and build its integral images.  Replaces the per-template image handling inside
 * cv2.matchTemplate (MTM/__init__.py:92). */
int mtm_set_image(mtm_ctx* ctx, const void* px, int rows, int cols, int chans, int dtype,
                  int64_t row_stride_bytes);
/* Same, but the image is area-downscaled by an integer factor on the device while it is laid out
 * (the search image becomes rows/factor x cols/factor; remainder rows/columns are dropped).  Replaces
 * the host-side cv2.resize(image, smallDim, interpolation=cv2.INTER_AREA) the reference's speed-up
 * recipe runs before matching (tutorials/Tutorial3-SpeedingUp.ipynb:395).  uint8 rounding follows
 * OpenCV's integer-factor INTER_AREA path: factor 2 -> (sum + 2) >> 2, else
 * rint((float)sum * (1.f / factor^2)). */
int mtm_set_image_downscaled(mtm_ctx* ctx, const void* px, int rows, int cols, int chans, int dtype,
                             int64_t row_stride_bytes, int factor);

/* Upload all templates of one matchTemplates/findMatches call and fix the method.  Template
 * statistics (cv::meanStdDev) are computed here.  Replaces the per-template arguments of
 * cv2.matchTemplate (MTM/__init__.py:92). */
int mtm_set_templates(mtm_ctx* ctx, const mtm_templ* templs, int n_templ, int method);

/* The caller's step BEFORE the hot path, on the device: the reference's users append rotated / flipped / rescaled
 * copies of their templates to listTemplates on the host (tutorials/Tutorial2-Template_Augmentation.ipynb:313,
 * np.rot90; multi-scale copies).  Here the caller hands over the BASES and a list of variants; every
 * (base, variant) pair becomes one unit, base-major (unit index = base * n_variants + variant), exactly as if
 * the copies had been passed to mtm_set_templates.  A unit is a view of a source kept on the device - the base,
 * or an area-resized copy a kernel makes of it - read through the variant's reflection / rotation, and the
 * operand packs of the score kernel are gathered from those views: no per-unit pixel work on the host, no
 * per-unit upload.  uint8 bases (1-4 channels), uint8 masks (transformed like their template).
 * Order of the steps of a variant: resize, np.fliplr, np.flipud, np.rot90(k). */
typedef struct mtm_variant {
    int32_t rot90;     /* 0..3 quarter turns counter-clockwise (np.rot90(a, k))                                   */
    int32_t flip_lr;   /* 1: np.fliplr                                                                          */
    int32_t flip_ud;   /* 1: np.flipud                                                                          */
    int32_t rows, cols;/* > 0: area-resize the base to rows x cols first (exact rational area average, rounded
                          half up: MTM.augment.resize_area); 0, 0 = keep the size                               */
    int32_t down;      /* > 1: integer-factor area downscale first, OpenCV INTER_AREA rounding
                          (MTM.augment.downscale); exclusive with rows / cols                                   */
} mtm_variant;
int mtm_set_templates_augmented(mtm_ctx* ctx, const mtm_templ* bases, int n_bases,
                                const mtm_variant* variants, int n_variants, int method);

/* ---- the hot path -------------------------------------------------------------------------- */
/* cv2.matchTemplate(image, template, method, mask) for template `templ_idx`
 * (MTM/__init__.py:92, via computeScoreMap :56-92): float32 (rows-h+1, cols-w+1) to host memory. */
int mtm_score_map(mtm_ctx* ctx, int templ_idx, float* out, int64_t out_row_stride_bytes);

/* The per-template pipeline of _multi_compute (MTM/__init__.py:222-241) for every template set by
 * mtm_set_templates, batched: score maps, then either local extrema above/below `score_threshold`
 * (skimage peak_local_max / scipy find_peaks semantics, :22-53) or the global extremum
 * (cv2.minMaxLoc, :226).  Hits come back ordered by template index, then descending quality, then
 * row-major position; coordinates are relative to the uploaded image.  The threshold is the
 * python float of the reference call; it is narrowed to float32 for the comparison with the
 * float32 map exactly as numpy does.  On MTM_E_OVERFLOW *n_out is the capacity needed. */
int mtm_find_matches(mtm_ctx* ctx, int mode, double score_threshold,
                     mtm_hit* out, int64_t capacity, int64_t* n_out);

/* One call for "this image, these templates": mtm_set_image + mtm_find_matches without the round trip to the
 * host in between - what one MTM.matchTemplates / findMatches call does with the image it is given
 * (MTM/__init__.py:95-177; the templates come from mtm_set_templates).  The image becomes the context's
 * current image.  Where the layout allows (one unmasked single-channel uint8 size class on the matrix-core
 * kernel) the image crosses PCIe in row bands, and the score kernel of the rows already there runs under the
 * transfer of the next band: the upload costs little more than its first band.  Results are those of the two
 * separate calls. */
int mtm_find_matches_image(mtm_ctx* ctx, const void* px, int rows, int cols, int chans, int dtype,
                           int64_t row_stride_bytes, int mode, double score_threshold,
                           mtm_hit* out, int64_t capacity, int64_t* n_out);

/* mtm_find_matches_image (local extrema) followed by MTM's non-maxima suppression - what MTM.matchTemplates does with
 * the hits of all templates (MTM/__init__.py:290-296 -> MTM/NMS.py:53-84 -> cv2.dnn.NMSBoxes): hits whose score passes
 * `score_threshold` (1 - score for TM_SQDIFF_NORMED, as NMS.py:73-75 transforms it), best first, each kept unless it
 * overlaps an already kept one by more than `max_overlap` (intersection over union, OpenCV's float expression); at most
 * `n_object` of them (n_object < 0: all).  Returns the kept hits in that order - the list mtm_nms would select from the
 * list mtm_find_matches_image returns.  Of thousands of peaks (dense images) the ones a neighbourhood's best peak suppresses
 * are dropped on the device and never cross PCIe; mtm_timing.n_hits is the number of peaks before the suppression. */
int mtm_find_matches_image_nms(mtm_ctx* ctx, const void* px, int rows, int cols, int chans, int dtype,
                               int64_t row_stride_bytes, double score_threshold, double max_overlap, int64_t n_object,
                               mtm_hit* out, int64_t capacity, int64_t* n_out);

/* One step of the process-per-GPU form in one native call (round 5): search this rank's shard of the caller's template
 * list (the context's current templates; global_idx[i] = list position of local template i), exchange the ranks' hit
 * lists through the context's communicator (mtm_comm_init; without one, or with one rank: no exchange), merge them in
 * template order and run MTM's non-maxima suppression - every rank returns the same kept hits, best first.  The
 * reference's fan-in of per-template results (MTM/__init__.py:173-177) + MTM/NMS.py:53-84 across processes.  A rank
 * without units passes n_local_templ = 0 (px may be NULL) and still takes part in the collective. */
int mtm_find_matches_image_sharded_nms(mtm_ctx* ctx, const void* px, int rows, int cols, int chans, int dtype,
                                       int64_t row_stride_bytes, double score_threshold, double max_overlap,
                                       int64_t n_object, int method, const int32_t* global_idx, int n_local_templ,
                                       mtm_hit* out, int64_t capacity, int64_t* n_out);

/* Stream form of mtm_find_matches ("thousands of images", reference
 * tutorials/Tutorial3-SpeedingUp.ipynb:564: same templates, one image after the other): returns the
 * hits of the CURRENT image exactly like mtm_find_matches and makes `next_px` the current image for
 * the following call.  The next image is uploaded and converted on a second HIP stream while the
 * kernels of the current image run, so its PCIe transfer costs no wall-clock.
 * The caller's buffer is only read during the call.  On MTM_E_OVERFLOW the swap has happened too
 * (fetch the hits with mtm_last_hits). */
int mtm_find_matches_next(mtm_ctx* ctx, int mode, double score_threshold, mtm_hit* out, int64_t capacity,
                          int64_t* n_out, const void* next_px, int rows, int cols, int chans, int dtype,
                          int64_t row_stride_bytes);

/* Split form of mtm_find_matches for hosts that want to work while the GPU does: _async queues everything
 * the call needs on the context's stream (statistics, score kernels, the fetch of the candidate list) and
 * returns without waiting for the GPU; _wait synchronises, extracts the peaks and delivers the hits exactly
 * like mtm_find_matches (MTM_E_OVERFLOW: fetch them with mtm_last_hits).  mtm_find_matches is the two
 * back to back.  Between the two the context must not be used for anything else; one call in flight per
 * context.  (The reference gets its overlap from a thread pool per call, MTM/__init__.py:172-175; a Python
 * host holds the GIL while it builds the hit list of the previous image, so the overlap has to come from
 * the native side.) */
int mtm_find_matches_async(mtm_ctx* ctx, int mode, double score_threshold);
int mtm_find_matches_wait(mtm_ctx* ctx, mtm_hit* out, int64_t capacity, int64_t* n_out);

/* The hit list of the last mtm_find_matches call again, without recomputing anything: the way to
 * collect the result after MTM_E_OVERFLOW told the caller the capacity it needs. */
int mtm_last_hits(mtm_ctx* ctx, mtm_hit* out, int64_t capacity, int64_t* n_out);

/* The score map of template `templ_idx` exactly as the last mtm_find_matches call computed it inside its
 * batched launch (cv2.matchTemplate, MTM/__init__.py:92) - nothing is recomputed.  Only valid when that call
 * materialised the maps (MTM_OPT_HITS_ONLY = 0, or any class that is not on the MFMA kernel); MTM_E_STATE
 * otherwise.  Parity tests use it to check the production launch at full size against the oracle. */
int mtm_last_score_map(mtm_ctx* ctx, int templ_idx, float* out, int64_t out_row_stride_bytes);

int mtm_get_timing(mtm_ctx* ctx, mtm_timing* out);

/* cv2.dnn.NMSBoxes as MTM.NMS uses it (MTM/NMS.py:73-82): keep hits with score > threshold
 * (float32, strict), stable sort by descending score, greedy IoU suppression with
 * overlap <= max_overlap, optional truncation to n_object (-1 = no limit).  `ascending` applies
 * the 1-score transform of MTM/NMS.py:73-75.  Host code (C++), no GPU needed.
 * keep[] receives indices into hits[]; capacity of keep must be >= n. */
int mtm_nms(const mtm_hit* hits, int64_t n, double score_threshold, int ascending,
            int64_t n_object, double max_overlap, int32_t* keep, int64_t* n_keep);

/* ---- multi-GPU, one process: a group of per-device contexts -------------------------------- */
/* The units (templates / rotations / scales) of a search are independent given the image - the reference runs
 * one thread-pool task per template (MTM/__init__.py:172-175).  A group holds one context and one worker thread
 * per listed device (a device may be listed more than once: several contexts on one GPU).  A search shards the
 * units over the devices by longest-processing-time-first on their multiply-accumulate cost
 * out_px * w * h * C (* 2 with a mask), every device uploads the image and searches its shard concurrently
 * (mtm_set_templates + mtm_find_matches_image; unchanged templates stay resident per device), and the hit lists
 * are merged on the host in template order: the result equals the single-device call.  No collective: in one
 * process every list is in host memory when its worker returns (the one-process-per-GPU form with the RCCL
 * all-gather follows below). */
typedef struct mtm_group mtm_group;
int      mtm_group_create(mtm_group** out, const int* device_ids, int n_devices);
void     mtm_group_destroy(mtm_group* g);
int      mtm_group_size(const mtm_group* g);
mtm_ctx* mtm_group_ctx(mtm_group* g, int i);                       /* per-device context (options, timing) */
int      mtm_group_set_option(mtm_group* g, int option, int64_t value);   /* on every context */
/* the partition a search would use: device_of_unit[i] = index (0 .. size-1) of the device unit i goes to */
int      mtm_group_shards(const mtm_group* g, const mtm_templ* templs, int n_templ, int method, int rows, int cols,
                          int32_t* device_of_unit);
/* mtm_set_templates + mtm_find_matches_image over all devices; hits ordered as mtm_find_matches orders them */
int      mtm_group_find_matches(mtm_group* g, const mtm_templ* templs, int n_templ, int method,
                                const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes,
                                int mode, double score_threshold, mtm_hit* out, int64_t capacity, int64_t* n_out);
/* The same + MTM's non-maxima suppression on the merged list inside the call (round 5; what mtm_find_matches_image_nms is
 * to a single context): local extrema, the kept hits in cv2.dnn.NMSBoxes' order, at most n_object of them (-1: all).
 * Replaces MTM.matchTemplates -> findMatches + NMS (MTM/__init__.py:289-296) for a device group in ONE native call. */
int      mtm_group_find_matches_nms(mtm_group* g, const mtm_templ* templs, int n_templ, int method,
                                    const void* px, int rows, int cols, int chans, int dtype, int64_t row_stride_bytes,
                                    double score_threshold, double max_overlap, int64_t n_object,
                                    mtm_hit* out, int64_t capacity, int64_t* n_out);
int      mtm_group_last_hits(mtm_group* g, mtm_hit* out, int64_t capacity, int64_t* n_out);
/* The hit exchange of a group as north_star / SURVEY 8e name it: a single process, ncclCommInitAll over the group's
 * devices, one stream per device, ONE all-gather of fixed-size slots of 24-byte hit records per device inside
 * ncclGroupStart / ncclGroupEnd; rank 0's gathered list is merged and returned (the reference's fan-in,
 * MTM/__init__.py:173-177, followed by the one global NMS of MTM/NMS.py:78).  mtm_group_comm_init() creates the
 * communicators and selects that exchange; it fails with MTM_E_COMM - and the group keeps the host merge - when RCCL is
 * missing or a device is listed twice (RCCL takes one rank per device).  mtm_group_comm_ranks(): ncclCommCount of rank 0
 * (0 without communicators).  mtm_group_set_exchange() switches between the two; both deliver the same list. */
#define MTM_GROUP_EXCHANGE_HOST 0
#define MTM_GROUP_EXCHANGE_RCCL 1
int      mtm_group_comm_init(mtm_group* g);
int      mtm_group_comm_ranks(mtm_group* g);
int      mtm_group_set_exchange(mtm_group* g, int kind);
int      mtm_group_exchange_used(const mtm_group* g);            /* what the last mtm_group_find_matches used */

/* ---- multi-GPU: one process per GPU, templates sharded across ranks (north_star) ------------ */
/* RCCL all-gather of per-rank hit lists over xGMI.  The 128-byte unique id is created on rank 0
 * and distributed by the caller's bootstrap (torch.distributed store, MPI, a file...). */
#define MTM_COMM_ID_BYTES 128
int mtm_comm_unique_id(void* id_out /* MTM_COMM_ID_BYTES */);
int mtm_comm_init(mtm_ctx* ctx, const void* id, int n_ranks, int rank);
/* all-gather: every rank contributes n_local hits; out receives the concatenation in rank order.
 * counts_out[n_ranks] receives the per-rank counts.  Collective call with a deadline: if the other ranks do not
 * arrive within MTM_COMM_TIMEOUT_S seconds (environment; default 300, 0 = wait for ever) the communicator is
 * aborted (ncclCommAbort), the call returns MTM_E_COMM and the context goes back to single-rank operation. */
int mtm_comm_allgather_hits(mtm_ctx* ctx, const mtm_hit* local, int64_t n_local,
                            mtm_hit* out, int64_t capacity, int64_t* counts_out, int64_t* n_out);
/* The result of the last mtm_comm_allgather_hits again (no communication): the way to collect it after
 * MTM_E_OVERFLOW told the caller the capacity it needs.  The collective itself is never repeated - a rank whose
 * buffer was large enough has already left it. */
int mtm_comm_last_gather(mtm_ctx* ctx, mtm_hit* out, int64_t capacity, int64_t* counts_out, int64_t* n_out);
int mtm_comm_destroy(mtm_ctx* ctx);
/* The same exchange for n contexts of ONE process (what mtm_group_comm_init / mtm_group_find_matches use): communicators
 * by ncclCommInitAll over the contexts' devices (context i becomes rank i; MTM_E_COMM if two contexts share a device),
 * and one call that queues every rank's all-gather inside ncclGroupStart / ncclGroupEnd, each on its context's stream, and
 * returns rank 0's gathered list.  mtm_comm_count(): ncclCommCount of the context's communicator, 0 without one. */
int mtm_comm_init_all(mtm_ctx* const* ctxs, int n);
int mtm_comm_count(mtm_ctx* ctx);
int mtm_comm_allgather_hits_all(mtm_ctx* const* ctxs, int n, const mtm_hit* const* local, const int64_t* n_local,
                                mtm_hit* out, int64_t capacity, int64_t* counts_out, int64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* MTM_HIP_H */
