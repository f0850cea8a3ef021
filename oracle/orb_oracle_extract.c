/*
 * orb_oracle_extract.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see orb_oracle.h) for
 * ORBextractor::operator() of the reference (/root/reference/src/ORBextractor.cc).
 *
 * Every function cites the reference lines it restates.  The OpenCV primitives the
 * reference calls are restated from the generic C++ paths of OpenCV 2.4.9-3.4.0
 * (imgproc resize/filter/smooth, features2d fast/fast_score, core copy/mathfuncs);
 * OpenCV is not part of the reference tree, so those are PARITY UNPINNED.
 *
 * Build with -ffp-contract=off and without -march flags: the float expressions
 * below must round after every operation (no FMA contraction).
 */
#include "orb_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define PATCH_SIZE 31
#define HALF_PATCH_SIZE 15
#define EDGE_THRESHOLD 19
#define MAX_LEVELS 32

static const int8_t k_pattern[1024] = {
#include "../include/orbx_pattern.inc"
};

/* ------------------------------------------------------------------------- */
/* OpenCV scalar helpers                                                      */
/* ------------------------------------------------------------------------- */

/* cvRound: nearest integer, ties to even.  On x86-64 OpenCV compiles it to the 32-bit cvtsd2si, which returns
 * INT_MIN for NaN and for values no int can hold; lrint (64-bit there) would wrap differently, so that case is explicit. */
int orc_cv_round(double v)
{
    if (!(v >= -2147483648.5 && v <= 2147483647.5)) return INT_MIN;
    const double r = rint(v);
    return r > 2147483647.0 || r < -2147483648.0 ? INT_MIN : (int)r;
}
static int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static int cv_ceil(double v)  { int i = (int)v; return i + (i < v); }
static int16_t sat_short_from_float(float v)
{
    int i = orc_cv_round(v);
    return (int16_t)(i < SHRT_MIN ? SHRT_MIN : i > SHRT_MAX ? SHRT_MAX : i);
}

/* cv::fastAtan2 (degrees), scalar path of OpenCV core/mathfuncs.cpp; called at
 * src/ORBextractor.cc:102.  All float, each operation rounded. */
float orc_fast_atan2(float y, float x)
{
    const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 =  0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 =  0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

void orc_fast_atan2_array(const float* y, const float* x, int n, float* out)
{
    for (int i = 0; i < n; ++i) out[i] = orc_fast_atan2(y[i], x[i]);
}

/* BORDER_REFLECT_101 index map (core/copy.cpp borderInterpolate). */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* cv::copyMakeBorder(..., BORDER_REFLECT_101[+ISOLATED]) as used at src/ORBextractor.cc:1121-1127:
 * the ROI already sits inside the padded buffer; only the frame is written. */
void orc_border101(uint8_t* roi, int w, int h, ptrdiff_t stride, int pad)
{
    for (int y = -pad; y < h + pad; ++y) {
        int sy = reflect101(y, h);
        uint8_t* drow = roi + (ptrdiff_t)y * stride;
        const uint8_t* srow = roi + (ptrdiff_t)sy * stride;
        if (y >= 0 && y < h) {
            for (int x = -pad; x < 0; ++x) drow[x] = srow[reflect101(x, w)];
            for (int x = w; x < w + pad; ++x) drow[x] = srow[reflect101(x, w)];
        } else {
            for (int x = -pad; x < w + pad; ++x) drow[x] = srow[reflect101(x, w)];
        }
    }
}

/* cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1 (imgproc resize: fixed-point
 * Q11 coefficients, HResizeLinear + VResizeLinear<uchar,int,short>); called at
 * src/ORBextractor.cc:1119.  An exact 2x2 decimation is rerouted to INTER_AREA as OpenCV does. */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, ptrdiff_t sstride,
                          uint8_t* dst, int dw, int dh, ptrdiff_t dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = orc_cv_round(scale_x), iscale_y = orc_cv_round(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        /* ResizeAreaFast, scale 2: rounded mean of the 2x2 block */
        for (int dy = 0; dy < dh; ++dy) {
            const uint8_t* s0 = src + (ptrdiff_t)(2 * dy) * sstride;
            const uint8_t* s1 = s0 + sstride;
            uint8_t* d = dst + (ptrdiff_t)dy * dstride;
            for (int dx = 0; dx < dw; ++dx)
                d[dx] = (uint8_t)((s0[2 * dx] + s0[2 * dx + 1] + s1[2 * dx] + s1[2 * dx + 1] + 2) >> 2);
        }
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int16_t* alpha = (int16_t*)malloc(sizeof(int16_t) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        alpha[2 * dx]     = sat_short_from_float((1.f - fx) * 2048);
        alpha[2 * dx + 1] = sat_short_from_float(fx * 2048);
    }
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * dw);
    rows[1] = (int*)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        int16_t b0 = sat_short_from_float((1.f - fy) * 2048);
        int16_t b1 = sat_short_from_float(fy * 2048);
        for (int k = 0; k < 2; ++k) {
            int r = sy + k;
            r = r < 0 ? 0 : (r < sh ? r : sh - 1);
            const uint8_t* S = src + (ptrdiff_t)r * sstride;
            int* D = rows[k];
            int dx = 0;
            for (; dx < xmax; ++dx) {
                int sx = xofs[dx];
                D[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
            }
            for (; dx < dw; ++dx) D[dx] = S[xofs[dx]] * 2048;
        }
        uint8_t* d = dst + (ptrdiff_t)dy * dstride;
        for (int x = 0; x < dw; ++x)
            d[x] = (uint8_t)((((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
    }
    free(rows[0]); free(rows[1]); free(xofs); free(alpha);
}

/* ------------------------------------------------------------------------- */
/* FAST-9_16 (features2d/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) */
/* ------------------------------------------------------------------------- */

static void fast_offsets(int pixel[25], ptrdiff_t stride)
{
    static const int off[16][2] = {
        {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
        {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    int k;
    for (k = 0; k < 16; ++k) pixel[k] = off[k][0] + off[k][1] * (int)stride;
    for (; k < 25; ++k) pixel[k] = pixel[k - 16];
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static int corner_score16(const uint8_t* ptr, const int pixel[], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = imin((int)d[k + 1], (int)d[k + 2]);
        a = imin(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, (int)d[k + 4]);
        a = imin(a, (int)d[k + 5]);
        a = imin(a, (int)d[k + 6]);
        a = imin(a, (int)d[k + 7]);
        a = imin(a, (int)d[k + 8]);
        a0 = imax(a0, imin(a, (int)d[k]));
        a0 = imax(a0, imin(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = imax((int)d[k + 1], (int)d[k + 2]);
        b = imax(b, (int)d[k + 3]);
        b = imax(b, (int)d[k + 4]);
        b = imax(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, (int)d[k + 6]);
        b = imax(b, (int)d[k + 7]);
        b = imax(b, (int)d[k + 8]);
        b0 = imin(b0, imax(b, (int)d[k]));
        b0 = imin(b0, imax(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

int orc_fast9_score(const uint8_t* center, ptrdiff_t stride, int threshold)
{
    int pixel[25];
    fast_offsets(pixel, stride);
    return corner_score16(center, pixel, threshold);
}

/* cv::FAST(img, kps, threshold, nonmax) on a w x h sub-image, as called per cell at
 * src/ORBextractor.cc:808-814.  Output in raster order, cell-local coordinates. */
int orc_fast9(const uint8_t* img, int w, int h, ptrdiff_t stride, int threshold, int nonmax,
              int16_t* ox, int16_t* oy, uint8_t* oscore, int cap)
{
    const int K = 8, N = 25;
    int i, j, k, pixel[25], nout = 0;
    if (w < 7 || h < 7) return 0; /* loops below are empty; also keeps the row buffers in range */
    fast_offsets(pixel, stride);
    threshold = imin(imax(threshold, 0), 255);

    uint8_t tab[512];
    for (i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    uint8_t* bufmem = (uint8_t*)calloc((size_t)w * 3, 1);
    int* cpmem = (int*)malloc(sizeof(int) * (size_t)(w + 1) * 3);
    uint8_t* buf[3] = {bufmem, bufmem + w, bufmem + 2 * w};
    int* cpbuf[3] = {cpmem + 1, cpmem + 1 + (w + 1), cpmem + 1 + 2 * (w + 1)};

    for (i = 3; i < h - 2; i++) {
        const uint8_t* ptr = img + (ptrdiff_t)i * stride + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, (size_t)w);
        int ncorners = 0;

        if (i < h - 3) {
            for (j = 3; j < w - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t* t = &tab[0] - v + 255;
                int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
                d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
                d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
                d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
                d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
                d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];

                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;

        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (k = 0; k < ncorners; k++) {
            j = cornerpos[k];
            int score = prev[j];
            if (!nonmax ||
                (score > prev[j + 1] && score > prev[j - 1] &&
                 score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                 score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                if (nout < cap) { ox[nout] = (int16_t)j; oy[nout] = (int16_t)(i - 1); oscore[nout] = (uint8_t)score; }
                nout++;
            }
        }
    }
    free(bufmem); free(cpmem);
    return nout;
}

/* ------------------------------------------------------------------------- */
/* GaussianBlur 7x7 sigma 2 (imgproc smooth.cpp/filter.cpp, 8u fixed-point)   */
/* ------------------------------------------------------------------------- */

/* getGaussianKernel(7, 2.0, CV_32F) then the 8u smooth-symmetric conversion to Q8 ints. */
void orc_gauss_taps(int32_t taps[7])
{
    const int n = 7;
    const double sigma = 2.0;
    double scale2x = -0.5 / (sigma * sigma), sum = 0;
    float cf[7];
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2x * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) cf[i] = (float)(cf[i] * sum);
    for (int i = 0; i < n; ++i) taps[i] = orc_cv_round(cf[i] * 256.0f);
}

/* GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on a standalone w x h image
 * (the reference blurs a clone, src/ORBextractor.cc:1084-1085, so the border comes from the
 * image itself).  rounding 0: generic FixedPtCastEx (sum + 32768) >> 16;
 * rounding 1: the x86 SIMD column pass (float, ties to even) for x < (w & ~3), generic for the tail. */
void orc_gauss7(const uint8_t* src, int w, int h, ptrdiff_t sstride, uint8_t* dst, ptrdiff_t dstride, int rounding)
{
    int32_t taps[7];
    orc_gauss_taps(taps);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (ptrdiff_t)y * sstride;
        for (int x = 0; x < w; ++x) {
            int32_t acc = 0;
            for (int k = 0; k < 7; ++k) acc += taps[k] * s[reflect101(x + k - 3, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    int simd_w = rounding ? (w & ~3) : 0;
    for (int y = 0; y < h; ++y) {
        uint8_t* d = dst + (ptrdiff_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int32_t acc = 0;
            for (int k = 0; k < 7; ++k) acc += taps[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
            int32_t v;
            if (x < simd_w) v = (acc + 32767 + ((acc >> 16) & 1)) >> 16; /* exact value, ties to even */
            else v = (acc + 32768) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* Extractor object                                                           */
/* ------------------------------------------------------------------------- */

typedef struct level_data {
    int w, h;                 /* ROI size */
    uint8_t* padded;          /* (w+38) x (h+38) */
    uint8_t* blurred;         /* w x h or NULL */
    int ncand, cand_cap;
    int16_t *cx, *cy; uint8_t* cs;
    int nkp, kp_cap;
    orc_keypoint* kps;
} level_data;

struct orc_extractor {
    orc_params p;
    double scaleFactor;       /* include/ORBextractor.h:223: the member is a double */
    float mvScaleFactor[MAX_LEVELS], mvLevelSigma2[MAX_LEVELS];
    float mvInvScaleFactor[MAX_LEVELS], mvInvLevelSigma2[MAX_LEVELS];
    int mnFeaturesPerLevel[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 2];
    level_data lv[MAX_LEVELS];
};

/* ORBextractor::ORBextractor, src/ORBextractor.cc:409-469 */
orc_extractor* orc_extractor_create(const orc_params* p)
{
    if (!p || p->nlevels < 1 || p->nlevels > MAX_LEVELS || p->nfeatures < 0) return NULL;
    orc_extractor* ex = (orc_extractor*)calloc(1, sizeof(*ex));
    ex->p = *p;
    ex->scaleFactor = p->scale_factor;
    const int nlevels = p->nlevels;
    ex->mvScaleFactor[0] = 1.0f;
    ex->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        ex->mvScaleFactor[i] = (float)(ex->mvScaleFactor[i - 1] * ex->scaleFactor);
        ex->mvLevelSigma2[i] = ex->mvScaleFactor[i] * ex->mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
        ex->mvInvScaleFactor[i] = 1.0f / ex->mvScaleFactor[i];
        ex->mvInvLevelSigma2[i] = 1.0f / ex->mvLevelSigma2[i];
    }
    float factor = (float)(1.0f / ex->scaleFactor);
    float nDesired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        ex->mnFeaturesPerLevel[level] = orc_cv_round(nDesired);
        sum += ex->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    ex->mnFeaturesPerLevel[nlevels - 1] = imax(p->nfeatures - sum, 0);

    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) ex->umax[v] = orc_cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (ex->umax[v0] == ex->umax[v0 + 1]) ++v0;
        ex->umax[v] = v0;
        ++v0;
    }
    return ex;
}

static void level_free(level_data* l)
{
    free(l->padded); free(l->blurred); free(l->cx); free(l->cy); free(l->cs); free(l->kps);
    memset(l, 0, sizeof(*l));
}

void orc_extractor_destroy(orc_extractor* ex)
{
    if (!ex) return;
    for (int i = 0; i < MAX_LEVELS; ++i) level_free(&ex->lv[i]);
    free(ex);
}

int orc_extractor_tables(const orc_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                         int32_t* features_per_level, int32_t* umax16)
{
    for (int i = 0; i < ex->p.nlevels; ++i) {
        if (scale) scale[i] = ex->mvScaleFactor[i];
        if (inv_scale) inv_scale[i] = ex->mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = ex->mvLevelSigma2[i];
        if (inv_sigma2) inv_sigma2[i] = ex->mvInvLevelSigma2[i];
        if (features_per_level) features_per_level[i] = ex->mnFeaturesPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = ex->umax[i];
    return ex->p.nlevels;
}

/* ORBextractor::ComputePyramid, src/ORBextractor.cc:1106-1131 */
static void compute_pyramid(orc_extractor* ex, const uint8_t* img, int w, int h, ptrdiff_t stride)
{
    for (int level = 0; level < ex->p.nlevels; ++level) {
        float scale = ex->mvInvScaleFactor[level];
        int lw = orc_cv_round((float)w * scale), lh = orc_cv_round((float)h * scale);
        level_data* L = &ex->lv[level];
        level_free(L);
        L->w = lw; L->h = lh;
        ptrdiff_t ps = lw + 2 * EDGE_THRESHOLD;
        L->padded = (uint8_t*)calloc((size_t)ps * (lh + 2 * EDGE_THRESHOLD), 1);
        uint8_t* roi = L->padded + EDGE_THRESHOLD * ps + EDGE_THRESHOLD;
        if (level != 0) {
            level_data* P = &ex->lv[level - 1];
            ptrdiff_t pps = P->w + 2 * EDGE_THRESHOLD;
            const uint8_t* proi = P->padded + EDGE_THRESHOLD * pps + EDGE_THRESHOLD;
            orc_resize_linear_u8(proi, P->w, P->h, pps, roi, lw, lh, ps);
        } else {
            for (int y = 0; y < lh; ++y) memcpy(roi + y * ps, img + y * stride, (size_t)lw);
        }
        orc_border101(roi, lw, lh, ps, EDGE_THRESHOLD);
    }
}

/* ---- DistributeOctTree, src/ORBextractor.cc:480-762 ---- */

typedef struct onode {
    struct onode *prev, *next;     /* std::list links */
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int* keys; int nkeys;          /* indices into the candidate arrays, input order preserved */
    int noMore;
    long seq;                      /* creation sequence: canonical stand-in for the heap address used as sort tie-break */
} onode;

typedef struct olist { onode *head, *tail; int size; long seq; } olist;

static onode* onode_new(int cap) { onode* n = (onode*)calloc(1, sizeof(onode)); n->keys = (int*)malloc(sizeof(int) * (size_t)(cap > 0 ? cap : 1)); return n; }
static void onode_free(onode* n) { free(n->keys); free(n); }
static void olist_push_back(olist* l, onode* n) { n->seq = l->seq++; n->next = NULL; n->prev = l->tail; if (l->tail) l->tail->next = n; else l->head = n; l->tail = n; l->size++; }
static void olist_push_front(olist* l, onode* n) { n->seq = l->seq++; n->prev = NULL; n->next = l->head; if (l->head) l->head->prev = n; else l->tail = n; l->head = n; l->size++; }
static onode* olist_erase(olist* l, onode* n) { onode* nx = n->next; if (n->prev) n->prev->next = n->next; else l->head = n->next; if (n->next) n->next->prev = n->prev; else l->tail = n->prev; l->size--; onode_free(n); return nx; }

/* ExtractorNode::DivideNode, src/ORBextractor.cc:480-536 */
static void divide_node(const onode* p, const int16_t* kx, const int16_t* ky, onode* c[4])
{
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    for (int i = 0; i < 4; ++i) c[i] = onode_new(p->nkeys);
    onode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;

    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;

    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;

    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;

    for (int i = 0; i < p->nkeys; ++i) {
        int k = p->keys[i];
        float x = (float)kx[k], y = (float)ky[k];
        if (x < n1->URx) {
            if (y < n1->BRy) n1->keys[n1->nkeys++] = k;
            else n3->keys[n3->nkeys++] = k;
        } else if (y < n1->BRy) n2->keys[n2->nkeys++] = k;
        else n4->keys[n4->nkeys++] = k;
    }
    for (int i = 0; i < 4; ++i) if (c[i]->nkeys == 1) c[i]->noMore = 1;
}

typedef struct szptr { int size; onode* node; } szptr;
static int szptr_cmp(const void* a, const void* b)
{
    const szptr *A = (const szptr*)a, *B = (const szptr*)b;
    if (A->size != B->size) return A->size < B->size ? -1 : 1;
    if (A->node->seq != B->node->seq) return A->node->seq < B->node->seq ? -1 : 1;
    return 0;
}

int orc_distribute_octree(const int16_t* kx, const int16_t* ky, const uint8_t* score, int n,
                          int minX, int maxX, int minY, int maxY, int N,
                          int32_t* out_index, int cap)
{
    /* :542-544 */
    const int nIni = (int)round((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return -2; /* the reference divides by zero here */
    const float hX = (float)(maxX - minX) / nIni;

    olist L = {0, 0, 0, 0};
    onode** ini = (onode**)malloc(sizeof(onode*) * (size_t)nIni);
    for (int i = 0; i < nIni; i++) {            /* :551-562 */
        onode* ni = onode_new(n);
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        olist_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n; i++) {               /* :565-569 */
        size_t r = (size_t)((float)kx[i] / hX);
        if (r >= (size_t)nIni) r = (size_t)nIni - 1; /* out of range is UB in the reference; cannot occur for x < maxX-minX */
        ini[r]->keys[ini[r]->nkeys++] = i;
    }
    free(ini);
    for (onode* it = L.head; it;) {             /* :571-584 */
        if (it->nkeys == 1) { it->noMore = 1; it = it->next; }
        else if (it->nkeys == 0) it = olist_erase(&L, it);
        else it = it->next;
    }

    int bFinish = 0;
    szptr* vsp = (szptr*)malloc(sizeof(szptr) * (size_t)(4 * (n + 4) + 16));
    szptr* vprev = (szptr*)malloc(sizeof(szptr) * (size_t)(4 * (n + 4) + 16));
    int nsp = 0;

    while (!bFinish) {                          /* :593-738 */
        int prevSize = L.size;
        int nToExpand = 0;
        nsp = 0;
        onode* lit = L.head;
        while (lit) {
            if (lit->noMore) { lit = lit->next; continue; }
            onode* c[4];
            divide_node(lit, kx, ky, c);
            for (int q = 0; q < 4; ++q) {
                if (c[q]->nkeys > 0) {
                    olist_push_front(&L, c[q]);
                    if (c[q]->nkeys > 1) { nToExpand++; vsp[nsp].size = c[q]->nkeys; vsp[nsp].node = c[q]; nsp++; }
                } else onode_free(c[q]);
            }
            lit = olist_erase(&L, lit);
        }
        if (L.size >= N || L.size == prevSize) bFinish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = L.size;
                int nprev = nsp;
                memcpy(vprev, vsp, sizeof(szptr) * (size_t)nprev);
                nsp = 0;
                qsort(vprev, (size_t)nprev, sizeof(szptr), szptr_cmp);
                for (int j = nprev - 1; j >= 0; j--) {
                    onode* c[4];
                    divide_node(vprev[j].node, kx, ky, c);
                    for (int q = 0; q < 4; ++q) {
                        if (c[q]->nkeys > 0) {
                            olist_push_front(&L, c[q]);
                            if (c[q]->nkeys > 1) { vsp[nsp].size = c[q]->nkeys; vsp[nsp].node = c[q]; nsp++; }
                        } else onode_free(c[q]);
                    }
                    olist_erase(&L, vprev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;
            }
        }
    }
    free(vsp); free(vprev);

    int nout = 0;                               /* :741-759 */
    for (onode* it = L.head; it; it = it->next) {
        int best = it->keys[0];
        float maxResponse = (float)score[best];
        for (int k = 1; k < it->nkeys; k++) {
            if ((float)score[it->keys[k]] > maxResponse) { best = it->keys[k]; maxResponse = (float)score[best]; }
        }
        if (nout < cap) out_index[nout] = best;
        nout++;
    }
    while (L.head) olist_erase(&L, L.head);
    return nout;
}

/* IC_Angle, src/ORBextractor.cc:76-103 */
static float ic_angle(const uint8_t* center, ptrdiff_t step, const int* u_max)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* glibc's sinf / cosf since 2.28 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h; constants of
 * __sincosf_table), restated for |y| < 120: one reduction by pi/2, then a double-precision polynomial.  use_fma = 1
 * evaluates `a + b * c` fused wherever the library's x86-64 FMA build (__sinf_fma / __cosf_fma, picked by its ifunc
 * on CPUs with FMA and AVX2) does; use_fma = 0 rounds every operation (the SSE2 build).  tests/test_oracle_cpu.py
 * walks every float angle in [0, 360) and finds this host's libm equal to one of the two. */
static double gl_mad(int use_fma, double a, double b, double c) { return use_fma ? fma(a, b, c) : a * b + c; }   /* built with -ffp-contract=off: a * b + c rounds twice */
static float gl_poly(int use_fma, double x, double x2, int negate, int cosine)
{
    const double sg = negate ? -1.0 : 1.0;     /* __sincosf_table[1]: c0..c4 negated, s1..s3 unchanged */
    if (!cosine) {
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        double x3 = x * x2;
        double t = gl_mad(use_fma, x2, s3, s2);
        double x7 = x3 * x2;
        double u = gl_mad(use_fma, x3, s1, x);
        return (float)gl_mad(use_fma, x7, t, u);
    }
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
                 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double t2 = gl_mad(use_fma, x2, c4, c3);
    double t1 = gl_mad(use_fma, x2, c1, c0);
    double x6 = x4 * x2;
    double c = gl_mad(use_fma, x4, c2, t1);
    return (float)gl_mad(use_fma, x6, t2, c);
}
static void gl_sincosf(int use_fma, float y, float* sn, float* cs)
{
    uint32_t bits; memcpy(&bits, &y, 4);
    const uint32_t top12 = (bits >> 20) & 0x7FFu;
    double x = y;
    if (top12 < 0x3F4u) {                       /* abstop12(y) < abstop12(pio4) */
        if (top12 < 0x398u) { *sn = y; *cs = 1.0f; return; }
        double x2 = x * x;
        *sn = gl_poly(use_fma, x, x2, 0, 0);
        *cs = gl_poly(use_fma, x, x2, 0, 1);
        return;
    }
    double r = x * 0x1.45F306DC9C883p+23;                 /* reduce_fast, !TOINT_INTRINSICS */
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double hpi = 0x1.921FB54442D18p0;
    x = use_fma ? fma(-(double)n, hpi, x) : x - n * hpi;
    double x2 = x * x;
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double xs = x * sign[n & 3];
    *sn = gl_poly(use_fma, xs, x2, n & 2, n & 1);
    *cs = gl_poly(use_fma, xs, x2, n & 2, !(n & 1));
}

void orc_trig(int trig_mode, float theta, float* cos_out, float* sin_out)
{
    if (trig_mode == 1 || trig_mode == 2) gl_sincosf(trig_mode == 1, theta, sin_out, cos_out);
    else if (trig_mode == 3) { *cos_out = cosf(theta); *sin_out = sinf(theta); }     /* what src/ORBextractor.cc:112 calls */
    else { *cos_out = (float)cos((double)theta); *sin_out = (float)sin((double)theta); }
}

static const float k_factorPI = (float)(3.1415926535897932384626433832795 / 180.f);   /* src/ORBextractor.cc:63 */

void orc_trig_range(int trig_mode, uint32_t first_bits, uint32_t count, float* cos_out, float* sin_out)
{
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t b = first_bits + i;
        float angle; memcpy(&angle, &b, 4);
        orc_trig(trig_mode, angle * k_factorPI, cos_out + i, sin_out + i);
    }
}

uint64_t orc_trig_count_differences(int mode_a, int mode_b, uint32_t first_bits, uint32_t count)
{
    uint64_t n = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t b = first_bits + i;
        float angle, ca, sa, cb, sb; memcpy(&angle, &b, 4);
        orc_trig(mode_a, angle * k_factorPI, &ca, &sa);
        orc_trig(mode_b, angle * k_factorPI, &cb, &sb);
        n += memcmp(&ca, &cb, 4) != 0 || memcmp(&sa, &sb, 4) != 0;
    }
    return n;
}

/* computeOrbDescriptor, src/ORBextractor.cc:106-146.  cos/sin: the reference calls the float overloads (glibc
 * cosf/sinf), trig_mode 3 here; trig_mode 1 / 2 restate that routine so that the device can follow it bit for bit;
 * trig_mode 0, the default, is the correctly rounded value (float)cos((double)angle), which glibc 2.35 cosf/sinf
 * reproduces for all but ~0.04%/0.09% of the float angles in [0,360) deg. */
static void orb_descriptor(const orc_keypoint* kpt, const uint8_t* img, ptrdiff_t step, uint8_t* desc, int trig_mode)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = kpt->angle * factorPI;
    float a, b;
    orc_trig(trig_mode, angle, &a, &b);
    const uint8_t* center = img + (ptrdiff_t)orc_cv_round(kpt->y) * step + orc_cv_round(kpt->x);
    const int8_t* pat = k_pattern;
#define GET_VALUE(idx) \
    center[orc_cv_round((float)pat[2 * (idx)] * b + (float)pat[2 * (idx) + 1] * a) * step + \
           orc_cv_round((float)pat[2 * (idx)] * a - (float)pat[2 * (idx) + 1] * b)]
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; ++j) {
            int t0 = GET_VALUE(2 * j), t1 = GET_VALUE(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
}

/* ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:764-852 */
static int compute_keypoints_octree(orc_extractor* ex)
{
    const float W = 30;
    for (int level = 0; level < ex->p.nlevels; ++level) {
        level_data* L = &ex->lv[level];
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = L->w - EDGE_THRESHOLD + 3;
        const int maxBorderY = L->h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        if (!(width > 0) || !(height > 0)) return 2;
        int nCols = (int)(width / W);
        int nRows = (int)(height / W);
        /* :785-786 with nCols or nRows == 0: width / nCols is a float division (inf, no trap); the cell loops below then
         * run zero times and the level hands DistributeOctTree an empty list.  wCell / hCell are never used in that case. */
        if (nCols <= 0 || nRows <= 0) nCols = nRows = 0;
        const int wCell = nCols ? (int)ceilf(width / nCols) : 0;
        const int hCell = nRows ? (int)ceilf(height / nRows) : 0;

        ptrdiff_t ps = L->w + 2 * EDGE_THRESHOLD;
        const uint8_t* roi = L->padded + EDGE_THRESHOLD * ps + EDGE_THRESHOLD;

        L->cand_cap = (L->w * L->h) / 4 + 64;
        L->cx = (int16_t*)malloc(sizeof(int16_t) * (size_t)L->cand_cap);
        L->cy = (int16_t*)malloc(sizeof(int16_t) * (size_t)L->cand_cap);
        L->cs = (uint8_t*)malloc((size_t)L->cand_cap);
        L->ncand = 0;
        const int cellcap = 4096;
        int16_t tx[4096], ty[4096]; uint8_t ts[4096];

        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
                const uint8_t* cell = roi + (ptrdiff_t)y0 * ps + x0;
                int nk = orc_fast9(cell, x1 - x0, y1 - y0, ps, ex->p.ini_th_fast, 1, tx, ty, ts, cellcap);
                if (nk == 0) nk = orc_fast9(cell, x1 - x0, y1 - y0, ps, ex->p.min_th_fast, 1, tx, ty, ts, cellcap);
                if (nk > cellcap) return 4;
                for (int k = 0; k < nk; ++k) {
                    if (L->ncand >= L->cand_cap) return 4;
                    L->cx[L->ncand] = (int16_t)(tx[k] + j * wCell);
                    L->cy[L->ncand] = (int16_t)(ty[k] + i * hCell);
                    L->cs[L->ncand] = ts[k];
                    L->ncand++;
                }
            }
        }

        const int N = ex->mnFeaturesPerLevel[level];
        int cap = L->ncand + 8;
        int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
        /* with no candidates the reference's DistributeOctTree returns an empty vector whatever nIni is (:565-569 never run) */
        int nsel = L->ncand == 0 && nCols == 0 ? 0
                 : orc_distribute_octree(L->cx, L->cy, L->cs, L->ncand, minBorderX, maxBorderX, minBorderY, maxBorderY, N, sel, cap);
        if (nsel < 0) { free(sel); return 2; }
        L->kp_cap = nsel > 0 ? nsel : 1;
        L->kps = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)L->kp_cap);
        L->nkp = nsel;
        const int scaledPatchSize = (int)(PATCH_SIZE * ex->mvScaleFactor[level]);
        for (int k = 0; k < nsel; ++k) {
            orc_keypoint* kp = &L->kps[k];
            kp->x = (float)L->cx[sel[k]] + minBorderX;
            kp->y = (float)L->cy[sel[k]] + minBorderY;
            kp->size = (float)scaledPatchSize;
            kp->angle = -1;
            kp->response = (float)L->cs[sel[k]];
            kp->octave = level;
            kp->class_id = -1;
        }
        free(sel);
    }
    for (int level = 0; level < ex->p.nlevels; ++level) {   /* :850-851 */
        level_data* L = &ex->lv[level];
        ptrdiff_t ps = L->w + 2 * EDGE_THRESHOLD;
        const uint8_t* roi = L->padded + EDGE_THRESHOLD * ps + EDGE_THRESHOLD;
        for (int k = 0; k < L->nkp; ++k) {
            orc_keypoint* kp = &L->kps[k];
            kp->angle = ic_angle(roi + (ptrdiff_t)orc_cv_round(kp->y) * ps + orc_cv_round(kp->x), ps, ex->umax);
        }
    }
    return 0;
}

/* ORBextractor::operator(), src/ORBextractor.cc:1042-1104 */
int orc_extract(orc_extractor* ex, const uint8_t* img, int w, int h, ptrdiff_t stride,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    if (n) *n = 0;
    if (!img || w <= 0 || h <= 0) return 0; /* empty image: silent return (:1045) */
    compute_pyramid(ex, img, w, h, stride);
    int rc = compute_keypoints_octree(ex);
    if (rc) return rc;
    int total = 0;
    for (int level = 0; level < ex->p.nlevels; ++level) total += ex->lv[level].nkp;
    if (total > cap) { if (n) *n = total; return 4; }
    int offset = 0;
    for (int level = 0; level < ex->p.nlevels; ++level) {
        level_data* L = &ex->lv[level];
        if (L->nkp == 0) continue;
        ptrdiff_t ps = L->w + 2 * EDGE_THRESHOLD;
        const uint8_t* roi = L->padded + EDGE_THRESHOLD * ps + EDGE_THRESHOLD;
        L->blurred = (uint8_t*)malloc((size_t)L->w * L->h);
        orc_gauss7(roi, L->w, L->h, ps, L->blurred, L->w, ex->p.blur_rounding);
        for (int k = 0; k < L->nkp; ++k)
            orb_descriptor(&L->kps[k], L->blurred, L->w, desc + (size_t)(offset + k) * 32, ex->p.trig_mode);
        float scale = ex->mvScaleFactor[level];
        for (int k = 0; k < L->nkp; ++k) {
            orc_keypoint kp = L->kps[k];
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            kps[offset + k] = kp;
        }
        offset += L->nkp;
    }
    if (n) *n = total;
    return 0;
}

int orc_level_size(const orc_extractor* ex, int level, int* w, int* h)
{
    if (level < 0 || level >= ex->p.nlevels) return 1;
    *w = ex->lv[level].w; *h = ex->lv[level].h;
    return 0;
}
const uint8_t* orc_level_padded(const orc_extractor* ex, int level, int* pw, int* ph)
{
    if (level < 0 || level >= ex->p.nlevels) return NULL;
    if (pw) *pw = ex->lv[level].w + 2 * EDGE_THRESHOLD;
    if (ph) *ph = ex->lv[level].h + 2 * EDGE_THRESHOLD;
    return ex->lv[level].padded;
}
const uint8_t* orc_level_blurred(const orc_extractor* ex, int level)
{
    if (level < 0 || level >= ex->p.nlevels) return NULL;
    return ex->lv[level].blurred;
}
int orc_level_candidates(const orc_extractor* ex, int level, int16_t* x, int16_t* y, uint8_t* score, int cap)
{
    if (level < 0 || level >= ex->p.nlevels) return -1;
    const level_data* L = &ex->lv[level];
    int m = L->ncand < cap ? L->ncand : cap;
    for (int i = 0; i < m; ++i) { x[i] = L->cx[i]; y[i] = L->cy[i]; score[i] = L->cs[i]; }
    return L->ncand;
}
int orc_level_keypoints(const orc_extractor* ex, int level, orc_keypoint* kps, int cap)
{
    if (level < 0 || level >= ex->p.nlevels) return -1;
    const level_data* L = &ex->lv[level];
    int m = L->nkp < cap ? L->nkp : cap;
    for (int i = 0; i < m; ++i) kps[i] = L->kps[i];
    return L->nkp;
}

/* ------------------------------------------------------------------------- */
/* Frame::ComputeStereoMatches, sub-pixel part, src/Frame.cc:551-639           */
/* ------------------------------------------------------------------------- */
typedef struct { int dist; int idx; } dist_idx;
static int dist_idx_cmp(const void* a, const void* b)
{
    const dist_idx* x = (const dist_idx*)a; const dist_idx* y = (const dist_idx*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);     /* sort(vector<pair<int,int>>), :626 */
}

void orc_stereo_refine(const orc_extractor* left, const orc_extractor* right,
                       const orc_keypoint* kl, int nl, const orc_keypoint* kr,
                       const int32_t* best_dist, const int32_t* best_idx,
                       float bf, float min_d, float max_d, float* uright, float* depth, int32_t* sad)
{
    const int thOrbDist = (100 + 50) / 2;                /* (TH_HIGH+TH_LOW)/2, :471 */
    dist_idx* v = (dist_idx*)malloc(sizeof(dist_idx) * (size_t)(nl > 0 ? nl : 1));
    int nv = 0;
    for (int iL = 0; iL < nl; ++iL) {
        uright[iL] = -1.0f; depth[iL] = -1.0f; sad[iL] = -1;
        if (best_dist[iL] < 0 || best_dist[iL] >= thOrbDist) continue;           /* :552 */
        const int oct = kl[iL].octave;
        const float uL = kl[iL].x;
        const float uR0 = kr[best_idx[iL]].x;
        const float scaleFactor = left->mvInvScaleFactor[oct];
        const float scaleduL = roundf(kl[iL].x * scaleFactor);
        const float scaledvL = roundf(kl[iL].y * scaleFactor);
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const level_data* ll = &left->lv[oct];
        const level_data* lr = &right->lv[oct];
        const ptrdiff_t psl = ll->w + 2 * EDGE_THRESHOLD, psr = lr->w + 2 * EDGE_THRESHOLD;
        const uint8_t* imL = ll->padded + EDGE_THRESHOLD * psl + EDGE_THRESHOLD;   /* mvImagePyramid[oct] = the ROI */
        const uint8_t* imR = lr->padded + EDGE_THRESHOLD * psr + EDGE_THRESHOLD;
        const float iniu = scaleduR0 + L - w;
        const float endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= lr->w) continue;                                  /* :575 */
        const int yl = (int)scaledvL, xl = (int)scaleduL, xr = (int)scaleduR0;
        /* rowRange/colRange outside the level throw in OpenCV (does not happen on extractor output): no match */
        if (yl - w < 0 || yl + w >= ll->h || xl - w < 0 || xl + w >= ll->w || xr - L - w < 0) continue;
        int bestDist = 2147483647, bestincR = 0;
        float vDists[11];
        const float cL = (float)imL[(ptrdiff_t)yl * psl + xl];
        for (int incR = -L; incR <= L; ++incR) {
            const float cR = (float)imR[(ptrdiff_t)yl * psr + xr + incR];
            float dist = 0;                                                       /* cv::norm(IL, IR, NORM_L1): integer-valued, exact */
            for (int dy = -w; dy <= w; ++dy)
                for (int dx = -w; dx <= w; ++dx) {
                    const float a = (float)imL[(ptrdiff_t)(yl + dy) * psl + xl + dx] - cL;
                    const float b = (float)imR[(ptrdiff_t)(yl + dy) * psr + xr + incR + dx] - cR;
                    dist += fabsf(a - b);
                }
            if (dist < (float)bestDist) { bestDist = (int)dist; bestincR = incR; }
            vDists[L + incR] = dist;
        }
        if (bestincR == -L || bestincR == L) continue;                            /* :594 */
        const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = left->mvScaleFactor[oct] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = uL - bestuR;
        if (disparity >= min_d && disparity < max_d) {
            if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
            depth[iL] = bf / disparity;
            uright[iL] = bestuR;
            sad[iL] = bestDist;
            v[nv].dist = bestDist; v[nv].idx = iL; ++nv;
        }
    }
    if (nv > 0) {                                                                 /* :626-639; empty vector: nothing to filter */
        qsort(v, (size_t)nv, sizeof(dist_idx), dist_idx_cmp);
        const float median = (float)v[nv / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; --i) {
            if ((float)v[i].dist < thDist) break;
            uright[v[i].idx] = -1.0f; depth[v[i].idx] = -1.0f;
        }
    }
    free(v);
}

/* cv::cvtColor to grey, 8-bit: OpenCV builds three 256-entry tables tab[i] = i*coeff (the red one offset by 2^13) and
 * sums one entry of each, which is this expression (modules/imgproc/src/color.cpp, RGB2Gray<uchar>, yuv_shift = 14). */
void orc_cvt_color_to_gray(const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, int blue_first,
                           uint8_t* dst, ptrdiff_t dst_stride)
{
    const int R2Y = 4899, G2Y = 9617, B2Y = 1868;
    const int c0 = blue_first ? B2Y : R2Y, c2 = blue_first ? R2Y : B2Y;
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + y * src_stride;
        uint8_t* d = dst + y * dst_stride;
        for (int x = 0; x < w; ++x, s += channels)
            d[x] = (uint8_t)((s[0] * c0 + s[1] * G2Y + s[2] * c2 + (1 << 13)) >> 14);
    }
}

/* cvUndistortPoints without R, with P = K (modules/imgproc/src/undistort.cpp).  k[] holds k1,k2,p1,p2,k3 and zeros for the
 * rational terms k4..k6, so the numerator of icdist is 1 + 0 = 1.  RR = P * I has exact entries fx, 0, cx / 0, fy, cy / 0, 0, 1. */
void orc_undistort_points(const float* xy, int n, float ffx, float ffy, float fcx, float fcy,
                          float fk1, float fk2, float fp1, float fp2, float fk3, float* xy_out)
{
    const double fx = ffx, fy = ffy, cx = fcx, cy = fcy;
    const double k[8] = {fk1, fk2, fp1, fp2, fk3, 0, 0, 0};
    const double ifx = 1. / fx, ify = 1. / fy;
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
    for (int i = 0; i < n; ++i) {
        double x = xy[2 * i], y = xy[2 * i + 1], x0, y0;
        x0 = x = (x - cx) * ifx;
        y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

/* ---- cv::remap, INTER_LINEAR, 8-bit, float maps ------------------------------------------------------------------ */
#define RM_TAB 32                 /* INTER_TAB_SIZE = 1 << INTER_BITS (5) */
#define RM_SCALE 32768            /* INTER_REMAP_COEF_SCALE = 1 << 15 */
static short rm_tab[RM_TAB * RM_TAB * 4 + 8];      /* BilinearTab_i: static, zero until filled (the sum fix-up looks past an entry) */
static int rm_tab_ready = 0;

static short rm_sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* initInterTab2D(INTER_LINEAR, fixpt = true) */
static void rm_init_tab(void)
{
    if (rm_tab_ready) return;
    float t1[RM_TAB * 2];
    const float scale = 1.f / RM_TAB;
    for (int i = 0; i < RM_TAB; ++i) { const float x = i * scale; t1[i * 2] = 1.f - x; t1[i * 2 + 1] = x; }   /* interpolateLinear */
    const int ksize = 2;
    short* itab = rm_tab;
    for (int i = 0; i < RM_TAB; ++i)
        for (int j = 0; j < RM_TAB; ++j, itab += ksize * ksize) {
            int isum = 0;
            for (int k1 = 0; k1 < ksize; ++k1) {
                const float vy = t1[i * ksize + k1];
                for (int k2 = 0; k2 < ksize; ++k2) {
                    const float v = vy * t1[j * ksize + k2];
                    isum += itab[k1 * ksize + k2] = rm_sat_short(orc_cv_round((double)(v * RM_SCALE)));
                }
            }
            if (isum != RM_SCALE) {
                const int diff = isum - RM_SCALE;
                const int ksize2 = ksize / 2;
                int Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; ++k1)
                    for (int k2 = ksize2; k2 < ksize2 + 2; ++k2) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    rm_tab_ready = 1;
}

void orc_remap_linear_u8(const uint8_t* src, int sw, int sh, ptrdiff_t sstride,
                         const float* map_x, const float* map_y, ptrdiff_t map_stride,
                         uint8_t* dst, int dw, int dh, ptrdiff_t dstride)
{
    rm_init_tab();
    const unsigned width1 = (unsigned)(sw - 1 > 0 ? sw - 1 : 0), height1 = (unsigned)(sh - 1 > 0 ? sh - 1 : 0);
    for (int dy = 0; dy < dh; ++dy) {
        const float* sX = map_x + dy * map_stride;
        const float* sY = map_y + dy * map_stride;
        uint8_t* D = dst + dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            /* RemapInvoker, CV_32FC1 pair of maps, bilinear */
            const int fx = orc_cv_round((double)(sX[dx] * RM_TAB)), fy = orc_cv_round((double)(sY[dx] * RM_TAB));
            const int v = (fy & (RM_TAB - 1)) * RM_TAB + (fx & (RM_TAB - 1));
            const int sx = rm_sat_short(fx >> 5), sy = rm_sat_short(fy >> 5);
            const short* w = rm_tab + v * 4;
            int val;
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const uint8_t* S = src + sy * sstride + sx;
                val = S[0] * w[0] + S[1] * w[1] + S[sstride] * w[2] + S[sstride + 1] * w[3];
            } else {
                if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) { D[dx] = 0; continue; }       /* cval */
                /* borderInterpolate(p, len, BORDER_CONSTANT) = p inside, -1 outside */
                const int sx0 = (unsigned)sx < (unsigned)sw ? sx : -1, sx1 = (unsigned)(sx + 1) < (unsigned)sw ? sx + 1 : -1;
                const int sy0 = (unsigned)sy < (unsigned)sh ? sy : -1, sy1 = (unsigned)(sy + 1) < (unsigned)sh ? sy + 1 : -1;
                const int v0 = sx0 >= 0 && sy0 >= 0 ? src[sy0 * sstride + sx0] : 0;
                const int v1 = sx1 >= 0 && sy0 >= 0 ? src[sy0 * sstride + sx1] : 0;
                const int v2 = sx0 >= 0 && sy1 >= 0 ? src[sy1 * sstride + sx0] : 0;
                const int v3 = sx1 >= 0 && sy1 >= 0 ? src[sy1 * sstride + sx1] : 0;
                val = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
            }
            const int r = (val + (1 << 14)) >> 15;                                                     /* FixedPtCast<int, uchar, 15> */
            D[dx] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}
