/*
 * sgmstereo_oracle.c -- CPU restatement of recon::SGMStereo, the reference's second float SGM
 * pipeline (reconstruction/base/new_sgm/sgm_stereo.{h,cc}).
 *
 * TEST INFRASTRUCTURE ONLY (see sgm_oracle.c): loaded by tests/ as the checker, never by the product.
 *
 * PARITY UNPINNED: the reference holds no fixture, output or test for this pipeline, and the class
 * needs OpenCV and Eigen, which this image lacks, so it is not rebuilt here.  The restatement below
 * follows sgm_stereo.cc statement by statement (same buffers, same loop order, same float/double
 * expression types) and each function cites the lines it follows.  cv::medianBlur(ksize 5) is
 * OpenCV arithmetic (not in the reference tree): median of the 5x5 neighbourhood, replicated border.
 *
 * Layouts: cost volumes float [y][x][d], d fastest; images row-major.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* float -> uint16 at sgm_stereo.cc:79-80 is undefined past 65535 (disparities >= 256); the reference's x86-64 builds
 * keep the low 16 bits of cvttss2si's result (0 for NaN / infinities / |r| >= 2^31).  Same rule as
 * cast_u16_like_x86 in sgm_oracle.c. */
static uint16_t ns_cast_u16_like_x86(double r)
{
    if (!(fabs(r) < 2147483648.0))
        return 0;
    return (uint16_t)(uint32_t)(int32_t)r;
}

#define ORACLE_API __attribute__((visibility("default")))

enum { NS_PATHS = 4 }; /* sgm_stereo.h:17 kNumPaths: paths per pass */

/* sgm_stereo.cc:158-181 ComputeLeftCostImage: ||fL[y][x] - fR[y][x-d]|| for x >= d; for x < d the
 * cost of the previous disparity is repeated (:174), i.e. the last valid one.  The K squared
 * differences are summed in index order in float32 (Eigen's own order is build dependent). */
ORACLE_API void oracle_ns_left_cost_desc(const float *fl, const float *fr, int w, int h, int K, int D, float *C)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int d = 0; d < D; d++) {
                const size_t idx = ((size_t)y * w + x) * D + d;
                if (x >= d) {
                    const float *a = fl + ((size_t)y * w + x) * K, *b = fr + ((size_t)y * w + x - d) * K;
                    float s = 0.f;
                    for (int k = 0; k < K; k++) {
                        const float df = a[k] - b[k];
                        s += df * df;
                    }
                    C[idx] = sqrtf(s);
                } else {
                    C[idx] = C[idx - 1];
                }
            }
}

/* sgm_stereo.cc:208-246 ComputeRightCostImage(): right[y][x-d][d] = left[y][x][d] (:215-233), then
 * for the last D-1 columns the disparities that would look past the image repeat the last valid
 * value (:235-244).  The reference indexes out of bounds when w < D; callers must ensure w >= D. */
ORACLE_API int oracle_ns_right_cost(const float *left, int w, int h, int D, float *right)
{
    if (w < D)
        return 1;
    const size_t step = (size_t)w * D;
    for (int y = 0; y < h; y++) {
        const float *lrow = left + step * y;
        float *rrow = right + step * y;
        for (int x = 0; x < D; x++) {
            const float *lp = lrow + (size_t)D * x;
            float *rp = rrow + (size_t)D * x;
            for (int d = 0; d <= x; d++) {
                *rp = *lp;
                rp -= D - 1;
                lp++;
            }
        }
        for (int x = D; x < w; x++) {
            const float *lp = lrow + (size_t)D * x;
            float *rp = rrow + (size_t)D * x;
            for (int d = 0; d < D; d++) {
                *rp = *lp;
                rp -= D - 1;
                lp++;
            }
        }
        for (int x = w - D + 1; x < w; x++) {
            const int maxd = w - x;
            const float last = rrow[(size_t)D * x + maxd - 1];
            for (int d = maxd; d < D; d++)
                rrow[(size_t)D * x + d] = last;
        }
    }
    return 0;
}

/* sgm_stereo.cc:416-488 SpeckleFilter: 4-connected regions of non-zero pixels whose neighbouring
 * values differ by at most maxDifference; every pixel of a region of at most maxSpeckleSize
 * pixels becomes 0.  Literal: explicit stack, labels, regionTypes, raster order. */
ORACLE_API void oracle_ns_speckle(int *image, int w, int h, int maxSpeckleSize, int maxDifference)
{
    const size_t n = (size_t)w * h;
    int *labels = calloc(n, sizeof(int));
    unsigned char *small = calloc(n + 2, 1); /* regionTypes */
    int *stack = malloc(sizeof(int) * (n + 1));
    int cur = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int pix = w * y + x;
            if (image[pix] == 0)
                continue;
            if (labels[pix] > 0) {
                if (small[labels[pix]])
                    image[pix] = 0;
                continue;
            }
            int sp = 0, total = 0;
            stack[sp++] = pix;
            cur++;
            labels[pix] = cur;
            while (sp) {
                const int c = stack[--sp];
                const int cx = c % w, cy = c / w;
                total++;
                const float v = (float)image[c]; /* CostType pixelValue */
                if (cx < w - 1 && labels[c + 1] == 0 && image[c + 1] != 0 && fabsf(v - image[c + 1]) <= maxDifference) {
                    labels[c + 1] = cur;
                    stack[sp++] = c + 1;
                }
                if (cx > 0 && labels[c - 1] == 0 && image[c - 1] != 0 && fabsf(v - image[c - 1]) <= maxDifference) {
                    labels[c - 1] = cur;
                    stack[sp++] = c - 1;
                }
                if (cy < h - 1 && labels[c + w] == 0 && image[c + w] != 0 && fabsf(v - image[c + w]) <= maxDifference) {
                    labels[c + w] = cur;
                    stack[sp++] = c + w;
                }
                if (cy > 0 && labels[c - w] == 0 && image[c - w] != 0 && fabsf(v - image[c - w]) <= maxDifference) {
                    labels[c - w] = cur;
                    stack[sp++] = c - w;
                }
            }
            if (total <= maxSpeckleSize) {
                small[cur] = 1;
                image[pix] = 0;
            }
        }
    free(labels);
    free(small);
    free(stack);
}

/* sgm_stereo.cc:248-414 PerformSGM: two raster passes (top-left, then bottom-right), four paths
 * each -- r0 the previous pixel of the row, r1/r2/r3 the previous row's x-step, x, x+step -- row
 * buffers lr_curr/lr_prev swapped after every row, sum_cost accumulated path by path in the order
 * the loops visit them (:357); a path that does not exist contributes the data cost alone (:333).
 * After the second pass of a row: first-minimum argmin (:371-378) and, when subpix != NULL, the
 * sub-pixel estimate with its + 0.5 (:383-400; int + float / float / double + double, cast to
 * float).  Ends with SpeckleFilter(100, 2) on the integer image (:413).  S_out (optional) gets
 * sum_cost. */
ORACLE_API void oracle_ns_perform_sgm(const float *cost, int w, int h, int D, float P1, float P2, int *disp, float *subpix,
                                      float *S_out)
{
    const size_t nvox = (size_t)w * h * D, lr_size = (size_t)w * D;
    float *sum = calloc(nvox, sizeof(float));
    float *lr_curr[NS_PATHS], *lr_prev[NS_PATHS], *min_curr[NS_PATHS], *min_prev[NS_PATHS];
    for (int i = 0; i < NS_PATHS; i++) {
        lr_curr[i] = calloc(lr_size, sizeof(float));
        lr_prev[i] = calloc(lr_size, sizeof(float));
        /* :252-253 memset(ptr, kCostMax, ...) stores the low byte of (int)FLT_MAX; whatever it is,
         * every entry is written before it is read, so zeros serve */
        min_curr[i] = calloc(w, sizeof(float));
        min_prev[i] = calloc(w, sizeof(float));
    }
    for (int pass = 0; pass < 2; pass++) {
        const int startX = pass == 0 ? 0 : w - 1, endX = pass == 0 ? w : -1, stepX = pass == 0 ? 1 : -1;
        const int startY = pass == 0 ? 0 : h - 1, endY = pass == 0 ? h : -1, stepY = pass == 0 ? 1 : -1;
        for (int y = startY; y != endY; y += stepY) {
            const float *crow = cost + (size_t)y * w * D;
            float *srow = sum + (size_t)y * w * D;
            for (int x = startX; x != endX; x += stepX) {
                const float *lr_p[NS_PATHS] = {NULL, NULL, NULL, NULL};
                float min_p[NS_PATHS] = {0.f, 0.f, 0.f, 0.f};
                const size_t xs = (size_t)x * D;
                float *cur_p[NS_PATHS];
                for (int r = 0; r < NS_PATHS; r++)
                    cur_p[r] = lr_curr[r] + xs;
                if (x != startX) {
                    lr_p[0] = lr_curr[0] + (size_t)(x - stepX) * D;
                    min_p[0] = min_curr[0][x - stepX];
                }
                if (y != startY) {
                    lr_p[2] = lr_prev[2] + xs;
                    min_p[2] = min_prev[2][x];
                    if (x != startX) {
                        lr_p[1] = lr_prev[1] + (size_t)(x - stepX) * D;
                        min_p[1] = min_prev[1][x - stepX];
                    }
                    if (x != endX - stepX) {
                        lr_p[3] = lr_prev[3] + (size_t)(x + stepX) * D;
                        min_p[3] = min_prev[3][x + stepX];
                    }
                }
                const float *dc = crow + xs;
                for (int r = 0; r < NS_PATHS; r++)
                    min_curr[r][x] = FLT_MAX;
                float *sp = srow + xs;
                for (int d = 0; d < D; d++)
                    for (int r = 0; r < NS_PATHS; r++) {
                        cur_p[r][d] = dc[d];
                        if (lr_p[r] != NULL) {
                            float agg = min_p[r] + P2;
                            if (lr_p[r][d] < agg)
                                agg = lr_p[r][d]; /* std::min(min + P2, lr[d]) */
                            if (d > 0) {
                                const float t = lr_p[r][d - 1] + P1;
                                if (agg > t)
                                    agg = t;
                            }
                            if (d < D - 1) {
                                const float t = lr_p[r][d + 1] + P1;
                                if (agg > t)
                                    agg = t;
                            }
                            cur_p[r][d] += agg - min_p[r];
                        }
                        if (min_curr[r][x] > cur_p[r][d])
                            min_curr[r][x] = cur_p[r][d];
                        sp[d] += cur_p[r][d];
                    }
            }
            if (pass == 1) {
                for (int x = 0; x < w; x++) {
                    const float *s = srow + (size_t)D * x;
                    float best = s[0];
                    int bd = 0;
                    for (int d = 1; d < D; d++)
                        if (s[d] < best) {
                            best = s[d];
                            bd = d;
                        }
                    disp[(size_t)w * y + x] = bd;
                    if (subpix) {
                        if (bd > 0 && bd < D - 1) {
                            const float c = s[bd], l = s[bd - 1], r = s[bd + 1];
                            if (r < l)
                                subpix[(size_t)w * y + x] = (float)(bd + (float)(r - l) / (c - l) / 2.0 + 0.5);
                            else
                                subpix[(size_t)w * y + x] = (float)(bd + (float)(r - l) / (c - r) / 2.0 + 0.5);
                        } else {
                            subpix[(size_t)w * y + x] = (float)bd;
                        }
                    }
                }
            }
            for (int r = 0; r < NS_PATHS; r++) { /* :406-407 std::swap of the pointer arrays */
                float *t = lr_curr[r];
                lr_curr[r] = lr_prev[r];
                lr_prev[r] = t;
                t = min_curr[r];
                min_curr[r] = min_prev[r];
                min_prev[r] = t;
            }
        }
    }
    if (S_out)
        memcpy(S_out, sum, nvox * sizeof(float));
    oracle_ns_speckle(disp, w, h, 100, 2);
    free(sum);
    for (int i = 0; i < NS_PATHS; i++) {
        free(lr_curr[i]);
        free(lr_prev[i]);
        free(min_curr[i]);
        free(min_prev[i]);
    }
}

/* sgm_stereo.cc:490-531 EnforceLeftRightConsistency: 'c' correct, 'm' mismatch, 'o' occlusion.
 * The mismatch search runs d = 0 .. min(x, D) inclusive (:512). */
ORACLE_API void oracle_ns_lr_states(const int *left, const int *right, int w, int h, int D, int thr, char *states)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int li = w * y + x;
            const int ld = left[li];
            if (ld > x) {
                states[li] = 'o';
                continue;
            }
            if (abs(ld - right[li - ld]) <= thr) {
                states[li] = 'c';
                continue;
            }
            char st = 'o';
            const int dmax = x < D ? x : D;
            for (int d = 0; d <= dmax; d++)
                if (d != ld && abs(ld - right[w * y + x - d]) <= thr) {
                    st = 'm';
                    break;
                }
            states[li] = st;
        }
}

/* sgm_stereo.cc:534-557 InterpolateDisparities, literal (including the test of disp_states[idx]
 * rather than [i] inside the fill loop, :548, which makes the fill stop after the pixel itself). */
ORACLE_API void oracle_ns_interpolate(char *states, float *disp, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int idx = w * y + x;
            if (states[idx] != 'o')
                continue;
            int cidx = -1;
            for (int i = 1; i <= x; i++)
                if (states[idx - i] == 'c') {
                    cidx = idx - i;
                    break;
                }
            if (cidx >= 0)
                for (int i = idx; i > cidx; i--)
                    if (states[idx] == 'o') {
                        disp[i] = disp[cidx];
                        states[i] = 'c';
                    }
        }
}

/* cv::medianBlur(img, img, 5) on CV_16U (sgm_stereo.cc:85): median of the 25 replicate-bordered
 * neighbours, from a copy of the input. */
ORACLE_API void oracle_median5_u16(const uint16_t *in, int W, int H, uint16_t *out)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint16_t v[25];
            int n = 0;
            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++) {
                    int yy = y + j, xx = x + i;
                    yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                    xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                    v[n++] = in[(size_t)yy * W + xx];
                }
            for (int a = 1; a < 25; a++) {
                const uint16_t t = v[a];
                int b = a - 1;
                while (b >= 0 && v[b] > t) {
                    v[b + 1] = v[b];
                    b--;
                }
                v[b + 1] = t;
            }
            out[(size_t)y * W + x] = v[12];
        }
}

/* sgm_stereo.cc:58-101 ComputeSGM on a left data-cost volume (what Compute(data_cost_path) holds
 * after LoadDataCostFromFile, :602-629, and Compute(left, right) after ComputeCostImage, :150-155):
 * left and right SGM, LR states, interpolation, round(subpix * 256.0) (:79-80; float * double,
 * std::round, to float, to uint16), median 5.  Optional taps: S_left [h][w][D], left/right integer
 * disparities after the speckle filter, states after interpolation, raw = image before the median. */
ORACLE_API int oracle_ns_compute(const float *left_cost, int w, int h, int D, int P1, int P2, int thr, uint16_t *disp,
                                 float *S_left, int *left_disp_out, int *right_disp_out, char *states_out, uint16_t *raw_out)
{
    const size_t npix = (size_t)w * h;
    float *right_cost = malloc(npix * D * sizeof(float));
    float *subpix = malloc(npix * sizeof(float));
    int *ld = malloc(npix * sizeof(int)), *rd = malloc(npix * sizeof(int));
    char *states = malloc(npix);
    uint16_t *raw = malloc(npix * sizeof(uint16_t));
    int rc = 2;
    if (right_cost && subpix && ld && rd && states && raw) {
        rc = oracle_ns_right_cost(left_cost, w, h, D, right_cost);
        if (rc == 0) {
            oracle_ns_perform_sgm(left_cost, w, h, D, (float)P1, (float)P2, ld, subpix, S_left);
            oracle_ns_perform_sgm(right_cost, w, h, D, (float)P1, (float)P2, rd, NULL, NULL);
            oracle_ns_lr_states(ld, rd, w, h, D, thr, states);
            oracle_ns_interpolate(states, subpix, w, h);
            for (size_t i = 0; i < npix; i++) {
                const float scaled = (float)round(subpix[i] * 256.0);
                raw[i] = ns_cast_u16_like_x86((double)scaled);
            }
            oracle_median5_u16(raw, w, h, disp);
            if (left_disp_out)
                memcpy(left_disp_out, ld, npix * sizeof(int));
            if (right_disp_out)
                memcpy(right_disp_out, rd, npix * sizeof(int));
            if (states_out)
                memcpy(states_out, states, npix);
            if (raw_out)
                memcpy(raw_out, raw, npix * sizeof(uint16_t));
        }
    }
    free(right_cost);
    free(subpix);
    free(ld);
    free(rd);
    free(states);
    free(raw);
    return rc;
}

/* ------------------------------------------------------------------ */
/* KITTI stereo evaluation (reconstruction/evaluation)  */
/* ------------------------------------------------------------------ */

/* io_disp.h:220-232 readDisparityMap: 16-bit gray value 0 = invalid (-1), else value / 256.0. */
ORACLE_API void oracle_kitti_read(const uint16_t *png, size_t n, float *data)
{
    for (size_t i = 0; i < n; i++)
        data[i] = png[i] == 0 ? -1.f : (float)(((float)png[i]) / 256.0);
}

/* io_disp.h:234-243 writeDisparityMap: (uint16_t)max(d * 256.0, 1.0) for valid pixels, 0 otherwise. */
ORACLE_API void oracle_kitti_write(const float *data, size_t n, uint16_t *png)
{
    for (size_t i = 0; i < n; i++) {
        const double s = data[i] * 256.0;
        png[i] = data[i] >= 0 ? (uint16_t)(s > 1.0 ? s : 1.0) : 0;
    }
}

/* io_disp.h:108-186 interpolateBackground, literal: per row, gaps between two valid pixels take the
 * smaller of the two, then the row is extrapolated to the left and to the right; then every column
 * is extrapolated to the top and to the bottom. */
ORACLE_API void oracle_kitti_interpolate_background(float *d, int width, int height)
{
#define VALID(u, v) (d[(size_t)(v) * width + (u)] >= 0)
#define AT(u, v) d[(size_t)(v) * width + (u)]
    for (int v = 0; v < height; v++) {
        int count = 0;
        for (int u = 0; u < width; u++) {
            if (VALID(u, v)) {
                if (count >= 1) {
                    const int u1 = u - count, u2 = u - 1;
                    if (u1 > 0 && u2 < width - 1) {
                        const float a = AT(u1 - 1, v), b = AT(u2 + 1, v);
                        const float ip = b < a ? b : a; /* std::min(a, b) */
                        for (int uc = u1; uc <= u2; uc++)
                            AT(uc, v) = ip;
                    }
                }
                count = 0;
            } else {
                count++;
            }
        }
        for (int u = 0; u < width; u++)
            if (VALID(u, v)) {
                for (int u2 = 0; u2 < u; u2++)
                    AT(u2, v) = AT(u, v);
                break;
            }
        for (int u = width - 1; u >= 0; u--)
            if (VALID(u, v)) {
                for (int u2 = u + 1; u2 <= width - 1; u2++)
                    AT(u2, v) = AT(u, v);
                break;
            }
    }
    for (int u = 0; u < width; u++) {
        for (int v = 0; v < height; v++)
            if (VALID(u, v)) {
                for (int v2 = 0; v2 < v; v2++)
                    AT(u, v2) = AT(u, v);
                break;
            }
        for (int v = height - 1; v >= 0; v--)
            if (VALID(u, v)) {
                for (int v2 = v + 1; v2 <= height - 1; v2++)
                    AT(u, v2) = AT(u, v);
                break;
            }
    }
#undef VALID
#undef AT
}

/* evaluate_stereo.cc:21-77 disparityErrorsOutlier: errors[2i] = share of ground-truth pixels whose
 * error exceeds i+1 pixels, errors[2i+1] the same over pixels the result also holds, errors[10] the
 * density; float counters and float divisions, pixels visited column by column.  Returns 1 where
 * the reference throws (no valid ground-truth pixel). */
ORACLE_API int oracle_kitti_errors_outlier(const float *gt, const float *orig, const float *ipol, int width, int height,
                                           float *errors /* 11 */)
{
    for (int i = 0; i < 11; i++)
        errors[i] = 0;
    int num_pixels = 0, num_pixels_result = 0;
    for (int u = 0; u < width; u++)
        for (int v = 0; v < height; v++) {
            const size_t k = (size_t)v * width + u;
            if (gt[k] >= 0) {
                const float d_err = fabsf(gt[k] - ipol[k]);
                for (int i = 0; i < 5; i++)
                    if (d_err > (float)(i + 1))
                        errors[i * 2]++;
                num_pixels++;
                if (orig[k] >= 0) {
                    for (int i = 0; i < 5; i++)
                        if (d_err > (float)(i + 1))
                            errors[i * 2 + 1]++;
                    num_pixels_result++;
                }
            }
        }
    if (num_pixels == 0)
        return 1;
    const float np = (float)num_pixels > 1.0f ? (float)num_pixels : 1.0f;
    const float nr = (float)num_pixels_result > 1.0f ? (float)num_pixels_result : 1.0f;
    for (int i = 0; i < 10; i += 2)
        errors[i] /= np;
    if (num_pixels_result > 0)
        for (int i = 1; i < 10; i += 2)
            errors[i] /= nr;
    errors[10] = (float)num_pixels_result / np;
    return 0;
}

/* evaluate_stereo.cc:79-119 disparityErrorsAverage: float sums of the absolute error in column-major
 * pixel order, over ground-truth pixels and over those the result also holds. */
ORACLE_API void oracle_kitti_errors_average(const float *gt, const float *orig, const float *ipol, int width, int height,
                                            float *errors /* 2 */)
{
    errors[0] = errors[1] = 0;
    int num_pixels = 0, num_pixels_result = 0;
    for (int u = 0; u < width; u++)
        for (int v = 0; v < height; v++) {
            const size_t k = (size_t)v * width + u;
            if (gt[k] >= 0) {
                const float d_err = fabsf(gt[k] - ipol[k]);
                errors[0] += d_err;
                num_pixels++;
                if (orig[k] >= 0) {
                    errors[1] += d_err;
                    num_pixels_result++;
                }
            }
        }
    errors[0] /= (float)num_pixels > 1.0f ? (float)num_pixels : 1.0f;
    errors[1] /= (float)num_pixels_result > 1.0f ? (float)num_pixels_result : 1.0f;
}

/* One image of the loop in eval() (evaluate_stereo.cc:158-175): result and the two ground-truth maps
 * as 16-bit images; out = noc_out[11], occ_out[11], noc_avg[2], occ_avg[2]; ipol_png (optional) =
 * D_ipol as D_ipol.write would store it.  Returns 1 where the reference throws. */
ORACLE_API int oracle_kitti_eval_image(const uint16_t *disp, const uint16_t *gt_noc, const uint16_t *gt_occ, int width,
                                       int height, float *out /* 26 */, uint16_t *ipol_png)
{
    const size_t n = (size_t)width * height;
    float *orig = malloc(n * 4), *ipol = malloc(n * 4), *noc = malloc(n * 4), *occ = malloc(n * 4);
    oracle_kitti_read(disp, n, orig);
    memcpy(ipol, orig, n * 4);
    oracle_kitti_interpolate_background(ipol, width, height);
    oracle_kitti_read(gt_noc, n, noc);
    oracle_kitti_read(gt_occ, n, occ);
    int rc = oracle_kitti_errors_outlier(noc, orig, ipol, width, height, out);
    rc |= oracle_kitti_errors_outlier(occ, orig, ipol, width, height, out + 11);
    oracle_kitti_errors_average(noc, orig, ipol, width, height, out + 22);
    oracle_kitti_errors_average(occ, orig, ipol, width, height, out + 24);
    if (ipol_png)
        oracle_kitti_write(ipol, n, ipol_png);
    free(orig);
    free(ipol);
    free(noc);
    free(occ);
    return rc;
}

/* reconstruction/main/create_rgbd_dataset/main.cc:12-27 CalculateDepthImage on the 16-bit disparity
 * (converted to CV_64F with 1/256 at :93).  Returns 1 where the reference throws (z >= 65536). */
ORACLE_API int oracle_depth_image(const uint16_t *disp, int W, int H, double f, double b, uint16_t *depth)
{
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const double d = disp[i] * (1.0 / 256.0);
        depth[i] = 0;
        if (d >= 1.0) {
            const double z = 100.0 * f * b / d;
            if (z >= 65536)
                return 1;
            depth[i] = (uint16_t)round(z);
        }
    }
    return 0;
}
