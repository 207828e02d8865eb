/*
 * sgmx.h -- C ABI of the MI355X-native semi-global matching engine (libsgmx.so).
 *
 * This is the drop-in boundary for the dense-stereo hot path of
 * ivankreso/stereo-vision.  The reference has no FFI layer of its own: its seam
 * is the C++ class API of reconstruction/base (recon::StereoSGM, recon::SGM).
 * Every entry point below names the reference interface it replaces
 * (file:line relative to the upstream tree).  include/sgmx_recon.hpp holds the
 * C++ adapter classes with the reference's own signatures on top of this ABI;
 * INTEGRATION.md shows how a maintainer binds it.
 *
 * Conventions
 *  - plain C, no C++/torch types; all functions return an int status
 *    (SGMX_OK == 0) and never throw or abort; sgmx_last_error() gives the text.
 *  - "host" entry points take host pointers and are synchronous.
 *  - "_dev" entry points take device pointers (hipMalloc'd memory on the
 *    context's device) plus a hipStream_t passed as void*; they only enqueue
 *    work on that stream and return; the caller synchronises.  Calls on one
 *    context share its scratch: a call on another stream than the previous one
 *    first waits (on the device) for that call's last launch group, so calls
 *    through one context never overlap; use one context per concurrent stream.
 *  - a context owns all device scratch and is reused across calls; one context
 *    per (host thread, device).  Nothing is written to the file system.
 *  - volumes are [y][x][d] with d fastest; images are row-major.
 */
#ifndef SGMX_H
#define SGMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SGMX_API __attribute__((visibility("default")))
#else
#define SGMX_API
#endif

enum {
    SGMX_OK = 0,
    SGMX_ERR_INVALID_ARG = 1, /* null pointer, bad size, unsupported parameter */
    SGMX_ERR_RANGE = 2,       /* penalties / disparity range outside the uint16 path-cost guarantee */
    SGMX_ERR_NO_DEVICE = 3,   /* no usable HIP device (the engine has no CPU fallback) */
    SGMX_ERR_HIP = 4,         /* a HIP runtime call failed; see sgmx_last_error */
    SGMX_ERR_OOM = 5,
    SGMX_ERR_STATE = 6        /* tap requested before a compute call, wrong context mode, ... */
};

/* Replaces struct StereoSGMParams (reconstruction/base/stereo_sgm.h:15-21) and
 * struct SGMParams (reconstruction/base/sgm.h:18-23): the first four fields are
 * theirs, same meaning.  The rest are extensions; zero selects the default. */
typedef struct sgmx_params {
    int disp_range;  /* number of disparities D, d in [0, D); 2 <= D <= 512.  Past D = 256 a winner
                        d >= 256 gives 256 (d + ds) >= 65536: the output keeps the low 16 bits, as
                        the reference's x86-64 builds do (stereo_sgm.h:309,312)      */
    int window_sz;   /* census window, odd, 1..31 (an even window reads past the image in the reference); m = (wsz-1)/2; ignored by the float path */
    int penalty1;    /* P1 >= 0                                                                 */
    int penalty2;    /* P2 >= 0.  Census entry points: 36*255 + 8*P2 and P1 + 9*255 + P2 must stay below the
                        uint16 guards (SGMX_ERR_RANGE from the call otherwise); the float pipelines take any */
    int num_paths;   /* 8 (default; order of stereo_sgm.cc:120-122) or 4 (its axis-aligned subset) */
    int device;      /* HIP device ordinal (default 0)                                           */
    int max_batch;   /* stereo pairs processed per launch group (default 1); sizes the scratch   */
    int flags;       /* 0 or SGMX_FLAG_MATERIALIZE (SGMX_FLAG_NO_SMALL_GROUPS: accepted, ignored)  */
} sgmx_params;

/* flags: keep the data-cost volume C and the summed-cost volume S in HBM as the reference does
 * (one streaming pass over both per direction).  Without it -- and when penalty2 <= 255 -- the
 * census path recomputes C from the census images and stores one byte per voxel and direction
 * instead; the results are identical.  penalty2 > 255 always uses the materialised form. */
#define SGMX_FLAG_MATERIALIZE 1
/* Deprecated, accepted and ignored: round 1's contexts ran single pairs through the materialised form unless this
 * flag was set; the volume-free form now serves every group size.  Kept so that callers built against that header
 * still create their contexts. */
#define SGMX_FLAG_NO_SMALL_GROUPS 2

typedef struct sgmx_ctx sgmx_ctx;

/* Library version string, e.g. "sgmx 0.1 (gfx950)". */
SGMX_API const char *sgmx_version(void);

/* Construct / destroy.  Replaces recon::StereoSGM::StereoSGM(StereoSGMParams&)
 * (stereo_sgm.h:56) and recon::SGM::SGM(SGMParams&) (sgm.h:64).  Fails with
 * SGMX_ERR_NO_DEVICE when no GPU is present -- there is no CPU path. */
SGMX_API int sgmx_create(const sgmx_params *params, sgmx_ctx **out);
SGMX_API void sgmx_destroy(sgmx_ctx *ctx);

/* Text of the last error on this context (ctx may be NULL for create errors). */
SGMX_API const char *sgmx_last_error(const sgmx_ctx *ctx);

/* Census/Hamming SGM on one rectified 8-bit pair; replaces
 * void recon::StereoSGM::compute(cv::Mat& left, cv::Mat& right, cv::Mat& disp)
 * (stereo_sgm.h:57, stereo_sgm.cc:15-188, COST_CENSUS build).  left/right: H rows
 * of W bytes, row pitch stride_* bytes.  disp: H rows of W uint16 (value = 256 *
 * disparity, 0 = invalid, zero border of width m), row pitch stride_disp BYTES.
 * The reference allocates disp itself (stereo_sgm.h:280); here the caller does. */
SGMX_API int sgmx_compute_census_u8(sgmx_ctx *ctx, const uint8_t *left, const uint8_t *right, int W, int H,
                                    int stride_left, int stride_right, uint16_t *disp, int stride_disp);

/* The same for n pairs stored back to back (pair i at left + i*H*W etc., dense
 * rows); what a driver such as scripts/run_sgm_batch_census.py:27-34 gets by
 * looping over files, in one call.  Processes max_batch pairs per launch group. */
SGMX_API int sgmx_compute_census_u8_batch(sgmx_ctx *ctx, int n, const uint8_t *left, const uint8_t *right, int W, int H,
                                          uint16_t *disp);

/* Page-locked host memory for the buffers handed to the host-memory entry points (no counterpart in
 * the reference, whose images live in cv::Mat heap memory): copies from and to it run at PCIe speed
 * and overlap the compute stream -- the batch call reaches 4 650 pairs/s from such buffers against
 * 3 300 from pageable ones at KITTI size.  Any host memory still works.  The context only selects
 * the device; the memory outlives it and is released with sgmx_host_free. */
SGMX_API int sgmx_host_alloc(sgmx_ctx *ctx, size_t bytes, void **out);
SGMX_API int sgmx_host_free(sgmx_ctx *ctx, void *ptr);

/* Device-resident form of the batch call: inputs and outputs already in HBM
 * (dense), work enqueued on `stream`.  n may exceed max_batch. */
SGMX_API int sgmx_compute_census_u8_dev(sgmx_ctx *ctx, int n, const uint8_t *d_left, const uint8_t *d_right, int W,
                                        int H, uint16_t *d_disp, void *stream);

/* sgmx_compute_census_u8_dev for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that must not reach hipMalloc / hipFree -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same arguments, same kernels and bit-identical results, but every
 * intermediate buffer (census images, the byte volumes of the directions, border rings, the pre-median image)
 * is a slice of one static 64 MiB __device__ array per GPU (sgmx_noalloc_scratch_bytes()) instead of the
 * context's scratch, no event is created, and nothing is allocated or freed on this path (sgmx_create, which
 * looks the array's address up once per device, is not part of it).
 * Limits: the volume-free formulation only (penalty2 <= 255, no SGMX_FLAG_MATERIALIZE), WTA row tables within
 * 64 KiB (width up to ~3 900), and one pair's buffers must fit the array.  One pair needs (h w Dp) bytes per
 * direction (h, w = cropped size, Dp = disp_range rounded up to 32, 64, 128, 256, 384 or 512) plus about 10 bytes
 * per pixel: 384x288 at D=16 takes 15.2 MB with 4 paths and 29.0 MB with 8 (4 resp. 2 pairs per launch group fit);
 * 288x200 at D=64, 8 paths 29.5 MB; 640x480 at D=64, 8 paths 159 MB: too large -> SGMX_ERR_INVALID_ARG with the byte
 * count (use sgmx_compute_census_u8_dev).  n pairs run in launch groups of as many pairs as fit (at most max_batch).
 * Under stream capture (hipStreamBeginCapture on `stream`, any capture mode) the call records its memsets and
 * kernels into the graph and nothing else: tests/test_zz_no_alloc.py captures two calls in global capture mode,
 * where hipMalloc / hipFree / synchronous copies would fail the capture, and replays the graph.
 * All calls on one GPU share that GPU's array, whatever the context: calls that overlap in time on one GPU
 * race; the array keeps the previous call's bytes and the call clears what must start at zero. */
SGMX_API int sgmx_compute_census_u8_dev_noalloc(sgmx_ctx *ctx, int n, const uint8_t *d_left, const uint8_t *d_right, int W,
                                                int H, uint16_t *d_disp, void *stream);
SGMX_API size_t sgmx_noalloc_scratch_bytes(void);

/* Float data-cost SGM; replaces the part of
 * cv::Mat recon::SGM::Compute(const std::string, const std::string) (sgm.h:65,
 * sgm.cc:14-174) that follows the data-cost loop: `cost` is the [H][W][D] float
 * volume the reference builds at sgm.cc:76-84 (FLT_MAX where x < d, sgm.cc:55),
 * D == params.disp_range.  disp: H x W uint16, NO padding (sgm.h:291), dense. */
SGMX_API int sgmx_compute_cost_f32(sgmx_ctx *ctx, const float *cost, int W, int H, int D, uint16_t *disp);
SGMX_API int sgmx_compute_cost_f32_dev(sgmx_ctx *ctx, int n, const float *d_cost, int W, int H, int D,
                                       uint16_t *d_disp, void *stream);

/* The whole of cv::Mat recon::SGM::Compute(left_desc_path, right_desc_path) (sgm.h:65,
 * sgm.cc:14-174) after the two descriptor files are in memory (LoadRepresentationFromFile,
 * sgm.h:378-410): desc_left/right are the [H][W][K] float tensors; the L2 data cost of
 * sgm.cc:76-84 is built on the GPU (float32, differences summed in index order), then the float
 * path above runs on it.  disp: H x W uint16, dense. */
SGMX_API int sgmx_compute_desc_f32(sgmx_ctx *ctx, const float *desc_left, const float *desc_right, int W, int H, int K,
                                   uint16_t *disp);

/* recon::SGMStereo, the reference's second float pipeline (reconstruction/base/new_sgm/sgm_stereo.{h,cc};
 * the alternate binary of scripts/run_sgm_batch_cnn.py:7): eight textbook paths (no carry-over between
 * directions) on the left volume and on the right volume derived from it (sgm_stereo.cc:208-246),
 * first-minimum WTA with the pipeline's own sub-pixel estimate (:362-402), speckle filter (100 pixels,
 * difference 2; :413-488), correct / mismatch / occlusion states (:490-531), interpolation (:534-557),
 * round(256 * d) and a 5x5 median (:79-85).  Uses params.disp_range / penalty1 / penalty2 (penalty1 <
 * penalty2 as SetSmoothnessCostParameters demands, :16-28; num_paths is ignored, the pipeline has eight);
 * W >= D and D <= 256 are required.
 *   _cost_f32: replaces void SGMStereo::Compute(const std::string data_cost_path, cv::Mat*) (sgm_stereo.h:29,
 *     sgm_stereo.cc:52-56) after LoadDataCostFromFile has read the [H][W][D] float volume (:602-627).
 *   _desc_f32: replaces void SGMStereo::Compute(left_descriptors_path, right_descriptors_path, cv::Mat*)
 *     (sgm_stereo.h:26-28, sgm_stereo.cc:38-50) after the two descriptor files are in memory; the data cost
 *     (:158-181, last valid cost repeated where x < d) is built on the GPU.
 * disp: H x W uint16, dense. */
SGMX_API int sgmx_sgmstereo_cost_f32(sgmx_ctx *ctx, const float *cost, int W, int H, int D, int consistency_threshold,
                                     uint16_t *disp);
SGMX_API int sgmx_sgmstereo_cost_f32_dev(sgmx_ctx *ctx, int n, const float *d_cost, int W, int H, int D,
                                         int consistency_threshold, uint16_t *d_disp, void *stream);
SGMX_API int sgmx_sgmstereo_desc_f32(sgmx_ctx *ctx, const float *desc_left, const float *desc_right, int W, int H, int K,
                                     int consistency_threshold, uint16_t *disp);

/* The KITTI stereo evaluation the batch scripts' outputs go through (reconstruction/evaluation): n result
 * maps and their two ground-truth maps (non-occluded, all), all in the 16-bit on-disk format (value = 256 *
 * d, 0 = invalid; io_disp.h:220-243), host pointers, dense.  Per image, as the loop body of eval()
 * (evaluate_stereo.cc:158-175): D_ipol = interpolateBackground(D_orig) (io_disp.h:108-186), then
 * disparityErrorsOutlier and disparityErrorsAverage (evaluate_stereo.cc:21-119) against both ground truths.
 * errors: n x 26 floats = noc_out[11], occ_out[11], noc_avg[2], occ_avg[2].  ipol (optional): D_ipol as
 * D_ipol.write stores it.  A ground truth without a valid pixel is an error (the reference throws). */
SGMX_API int sgmx_kitti_eval_u16(sgmx_ctx *ctx, int n, const uint16_t *disp, const uint16_t *gt_noc, const uint16_t *gt_occ,
                                 int W, int H, float *errors, uint16_t *ipol);

/* CalculateDepthImage (reconstruction/main/create_rgbd_dataset/main.cc:12-27) on a 16-bit disparity image as
 * compute returns it: d = value / 256 (:93); depth = round(100 * focal * baseline / d) centimetres where
 * d >= 1, else 0 (calib[0], calib[4] at :13-14).  Host pointers, dense.  SGMX_ERR_RANGE when a depth
 * reaches 65536 (the reference throws, :22). */
SGMX_API int sgmx_depth_from_disparity_u16(sgmx_ctx *ctx, const uint16_t *disp, int W, int H, double focal, double baseline,
                                           uint16_t *depth);

/* The callers' pre-filter cv::GaussianBlur(img, img, cv::Size(3,3), sigma) on an 8-bit image
 * (reconstruction/main/dense_stereo/main.cc:83-85, stereo_eval/main.cc:36-38), host pointers,
 * src and dst may be the same buffer.  OpenCV is not part of the reference tree and its version is
 * unpinned: this is the 2.4-era integer path (kernel rounded to 8 fractional bits, int32 row pass,
 * rounded column pass, reflect-101 border). */
SGMX_API int sgmx_gaussian_blur3_u8(sgmx_ctx *ctx, const uint8_t *src, int W, int H, int stride_src, double sigma,
                                    uint8_t *dst, int stride_dst);

/* The same pre-filter fused into the census entry points: once enabled, sgmx_compute_census_u8, _batch and _dev
 * blur both images of every pair on the device (same arithmetic as sgmx_gaussian_blur3_u8, bit for bit) before
 * the census transform -- what dense_stereo/main.cc:83-85 and stereo_eval/main.cc:36-38 do to every image on the
 * host before compute(), without the two host round trips per pair.  sigma = 0: OpenCV's default for a 3x3
 * kernel (0.8).  The no-allocation entry point refuses contexts with the pre-blur on. */
SGMX_API int sgmx_set_preblur3(sgmx_ctx *ctx, int enabled, double sigma);

/* ---- stage taps (host pointers; for parity tests and debugging) ---- */

/* StereoCosts::census_transform (stereo_costs.cc:36-61): out is (H-2m) x (W-2m) uint32. */
SGMX_API int sgmx_census_u32(sgmx_ctx *ctx, const uint8_t *img, int W, int H, int stride, uint32_t *out);

/* The data-cost loop stereo_sgm.cc:75-101 with hamming_dist (stereo_costs.h:120-132)
 * and the 255 initialisation (stereo_sgm.cc:43): census maps are h x w uint32,
 * out is [h][w][D] uint8, D == params.disp_range. */
SGMX_API int sgmx_hamming_volume_u8(sgmx_ctx *ctx, const uint32_t *census_left, const uint32_t *census_right, int w,
                                    int h, uint8_t *out);

/* compute_hamming (core/hamming.h:67-68, core/hamming.c:215-236):
 * dis[j*na + i] = popcount(a_i xor b_j) over ncodes bytes. */
SGMX_API int sgmx_hamming_matrix_u16(sgmx_ctx *ctx, uint16_t *dis, const uint8_t *a, const uint8_t *b, int na, int nb,
                                     int ncodes);

/* Thresholded matching of binary signatures, the rest of the core/hamming library
 * (core/hamming.h:73-104).  Signatures are ncodes bytes each; a match is a pair whose Hamming
 * distance is <= ht.  Matches come in the reference's loop order (query i outer, base j inner):
 *   idx[2p] = i, idx[2p+1] = j, hams[p] = distance      (core/hamming.c:743-762, 817-837)
 * The reference's prealloc forms trust the caller's buffers; here `capacity` (in matches) is
 * checked: when it is too small nothing is written, *nptr holds the count needed and the call
 * returns SGMX_ERR_INVALID_ARG.  Typical use, as in the reference: count, allocate, prealloc.
 *   sgmx_match_hamming_count           match_hamming_count           core/hamming.h:75-76
 *   sgmx_crossmatch_hamming_count      crossmatch_hamming_count      core/hamming.h:78-79  (pairs j > i of one set)
 *   sgmx_match_hamming_thres_prealloc  match_hamming_thres_prealloc  core/hamming.h:90-92
 *   sgmx_crossmatch_hamming_prealloc   crossmatch_hamming_prealloc   core/hamming.h:103-104 */
SGMX_API int sgmx_match_hamming_count(sgmx_ctx *ctx, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht, int ncodes,
                                      size_t *nptr);
SGMX_API int sgmx_crossmatch_hamming_count(sgmx_ctx *ctx, const uint8_t *dbs, long n, int ht, int ncodes, size_t *nptr);
SGMX_API int sgmx_match_hamming_thres_prealloc(sgmx_ctx *ctx, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht,
                                               int ncodes, int *idx, uint16_t *hams, size_t capacity, size_t *nptr);
SGMX_API int sgmx_crossmatch_hamming_prealloc(sgmx_ctx *ctx, const uint8_t *dbs, long n, int ht, int ncodes, int *idx,
                                              uint16_t *hams, size_t capacity, size_t *nptr);

/* The allocating and "threaded" forms of the same library (core/hamming.h:84-96, 108-112).
 *   sgmx_hammatch                   hammatch_t                  core/hamming.h:52-56 (same layout: two ints, a uint16)
 *   sgmx_match_hamming_thres        match_hamming_thres         core/hamming.h:84-86, core/hamming.c:533-577
 *   sgmx_crossmatch_hamming         crossmatch_hamming          core/hamming.h:95-96, core/hamming.c:768-807
 *     *hmptr receives a malloc'ed array of at least max(bufsize, 1) records (the reference mallocs bufsize and grows
 *     it by realloc), *nptr the number of matches, records in the reference's loop order; release with
 *     sgmx_free (= free()).  On an error return *hmptr is NULL (argument errors) or a valid malloc'ed pointer to
 *     release.
 *   sgmx_compute_hamming_thread     compute_hamming_thread      core/hamming.h:108-109, core/hamming.c:849-860
 *     the same matrix as compute_hamming.
 *   sgmx_match_hamming_thres_nt     match_hamming_thres_nt      core/hamming.h:111-112, core/hamming.c:863-919
 *     The reference cuts both sets into blocks of 128 signatures and writes block pair by block pair: matches are
 *     ordered by block pair (bl = 0 .. nblock1*nblock2-1; queries of block bl / nblock1, base block bl % nblock1),
 *     then query, then base, and the stored indices are LOCAL to the block pair (0..127) -- reproduced as is.  Its
 *     block arithmetic stays inside the sets only when both sets have the same number of blocks; other shapes are
 *     SGMX_ERR_INVALID_ARG here (undefined behaviour there).  nt is accepted and ignored.  *keys (2 ints per match)
 *     and *ham are malloc'ed; the reference returns the count, here it goes to *nmatches.  Release with sgmx_free. */
typedef struct sgmx_hammatch {
    int qid;        /* query id */
    int bid;        /* base id */
    uint16_t score; /* Hamming distance */
} sgmx_hammatch;
SGMX_API int sgmx_match_hamming_thres(sgmx_ctx *ctx, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht, int ncodes,
                                      size_t bufsize, sgmx_hammatch **hmptr, size_t *nptr);
SGMX_API int sgmx_crossmatch_hamming(sgmx_ctx *ctx, const uint8_t *dbs, long n, int ht, int ncodes, long bufsize,
                                     sgmx_hammatch **hmptr, size_t *nptr);
SGMX_API int sgmx_compute_hamming_thread(sgmx_ctx *ctx, uint16_t *dis, const uint8_t *a, const uint8_t *b, int na, int nb,
                                         int ncodes);
SGMX_API int sgmx_match_hamming_thres_nt(sgmx_ctx *ctx, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht,
                                         int ncodes, int nt, int **keys, uint16_t **ham, size_t *nmatches);
SGMX_API void sgmx_free(void *ptr);

/* After a compute call, copy stages of pair `index` (0-based within the last
 * launch group: a call runs its pairs in launches of at most max_batch pairs,
 * on the volume-free census path also of at most 32 pairs and at most 208 MB
 * of census images -- 3 pairs at 3840 x 2160 with disp_range 256 -- so that
 * the images stay cache resident between the direction sweeps; a call with
 * one pair always has index 0) to the host: the data cost C [h][w][D] (uint8 for the census
 * path -- costs at stereo_sgm.cc:31-51; float for the float path), the summed
 * path costs S [h][w][D] (aggr_costs, stereo_sgm.cc:133 / sgm.cc:116; uint16 here,
 * uint32 in the reference, same values), the disparity image before the median
 * (get_disparity_image_uint16, stereo_sgm.h:275-317 / sgm.h:285-332).  Any
 * pointer may be NULL. */
SGMX_API int sgmx_get_stages_census(sgmx_ctx *ctx, int index, uint8_t *C, uint16_t *S, uint16_t *raw_disp);
/* Which pairs the taps can reach: the last launch group of the last compute call held `count` pairs, the first of
 * them pair `first_pair` of that call -- tap index i is the call's pair first_pair + i.  (E.g. 40 KITTI-size pairs
 * with max_batch 40 run as launches of 32 + 8: first_pair 32, count 8.) */
SGMX_API int sgmx_last_launch_group(const sgmx_ctx *ctx, int *first_pair, int *count);
SGMX_API int sgmx_get_stages_f32(sgmx_ctx *ctx, int index, float *S, uint16_t *raw_disp);

/* Stages of volume `index` of the last SGMStereo launch group: summed costs of the left pass (sum_cost_,
 * sgm_stereo.cc:357), left / right integer disparities after the speckle filter (:62-65, 413), states after
 * the interpolation ('c', 'm', 'o'; :490-557), the 16-bit image before the median (:79-80). */
SGMX_API int sgmx_get_stages_sgmstereo(sgmx_ctx *ctx, int index, float *S_left, int32_t *left_disp, int32_t *right_disp,
                                       uint8_t *states, uint16_t *raw_disp);

/* GetDisparityImage (stereo_sgm.h:243-257): WTA on the raw data cost of pair
 * `index` of the last census launch group, (h+2m) x (w+2m) uint8.  The reference
 * writes this to ./data_cost_img.png on every call (stereo_sgm.cc:184-185); here
 * it is computed only on request and never written to disk. */
SGMX_API int sgmx_get_data_cost_wta_u8(sgmx_ctx *ctx, int index, uint8_t *out);

/* ---- measurement ---- */

enum {
    SGMX_K_CENSUS = 0,
    SGMX_K_HAMMING = 1,
    SGMX_K_AGGREGATE = 2, /* all directions; one launch per direction per launch group */
    SGMX_K_WTA = 3,
    SGMX_K_MEDIAN = 4,
    SGMX_K_HAMLIB = 5, /* the distance and matching kernels of the core/hamming entry points */
    SGMX_K_COUNT = 6
};

/* SGMX_PROFILE_ALL (1): every kernel launch is bracketed by two HIP events on the launch
 * stream.  SGMX_PROFILE_SWEEPS (2): only the direction sweeps of the volume-free census path, and the sweeps of a
 * launch group share their boundary events (9 markers per group of 8 sweeps instead of 26 for all its kernels:
 * at KITTI size the markers of mode 1 cost 1.4 % of a step, those of mode 2 a third of that); the other classes
 * then report nothing.  0: off.  sgmx_kernel_stats synchronises, returns the accumulated
 * duration and launch count of one kernel class and clears them. */
#define SGMX_PROFILE_ALL 1
#define SGMX_PROFILE_SWEEPS 2
SGMX_API int sgmx_set_profiling(sgmx_ctx *ctx, int mode);
SGMX_API int sgmx_kernel_stats(sgmx_ctx *ctx, int kernel_class, double *total_ms, long *launches);
/* The SGMX_K_AGGREGATE time again, split by direction: k = index of the direction in the schedule (8 paths: the
 * order of stereo_sgm.cc:120-122, (-1,-1) (-1,0) (-1,1) (0,-1) (0,1) (1,-1) (1,0) (1,1) as (DIRX, DIRY); 4 paths:
 * its axis-aligned subset).  Same semantics as sgmx_kernel_stats: synchronises, returns and clears. */
SGMX_API int sgmx_direction_stats(sgmx_ctx *ctx, int k, double *total_ms, long *launches);

/* Algorithmic HBM bytes of one pair on the census path, B = V*(5N+1) + 10*W*H
 * (SURVEY.md 8d), and of its aggregation launches alone, 5*N*V - 2*V. */
SGMX_API double sgmx_algorithmic_bytes_census(const sgmx_ctx *ctx, int W, int H, double *aggregate_only);

#ifdef __cplusplus
}
#endif
#endif /* SGMX_H */
