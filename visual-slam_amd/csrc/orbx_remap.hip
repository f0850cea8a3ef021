// MI355X (gfx950) kernel for the stereo rectification that precedes extraction in the reference's stereo front ends:
// cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) with the CV_32F maps of cv::initUndistortRectifyMap
// (Examples/Stereo/stereo_euroc.cc:97-98, 136-137; Examples/ROS/ORB_SLAM2/src/ros_stereo.cc:106-107, 161-162).
// OpenCV (2.4 / 3.x, modules/imgproc/src/imgwarp.cpp) turns the float maps into a source pixel (shorts) plus a 5+5 bit
// fraction, and blends the 2x2 neighbourhood with Q15 weights; pixels outside the source are 0 (BORDER_CONSTANT).
// A byte gather bound by cache-line traffic: one thread produces four adjacent destination pixels.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>
#include <algorithm>
#include "orbx_internal.h"

using namespace orbx;

#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
            return ORBX_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

namespace {

// cvRound(v * INTER_TAB_SIZE): v * 32 is exact in float; values no int can hold (and NaN) become INT_MIN as with the
// x86 conversion instructions OpenCV's cvRound compiles to
__device__ __forceinline__ int fixed5(float v)
{
    const float s = v * 32.0f;
    if (!(fabsf(s) < 2147483648.0f)) return (int)0x80000000;
    return (int)rintf(s);
}
__device__ __forceinline__ int sat_short(int v) { return min(max(v, -32768), 32767); }

// per destination pixel: source column and row of the top-left neighbour (two shorts in one word) and the fraction index
// (fy << 5 | fx), the XY / A arrays cv::remap builds block by block on every call
__global__ __launch_bounds__(256) void k_remap_prepare(const float* __restrict__ mx, const float* __restrict__ my, long long map_stride,
                                                       int dw, int dh, uint32_t* __restrict__ xy, uint16_t* __restrict__ frac)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int sx = fixed5(mx[(long long)y * map_stride + x]), sy = fixed5(my[(long long)y * map_stride + x]);
    xy[(size_t)y * dw + x] = (uint32_t)(sat_short(sx >> 5) & 0xFFFF) | ((uint32_t)sat_short(sy >> 5) << 16);
    frac[(size_t)y * dw + x] = (uint16_t)(((sy & 31) << 5) | (sx & 31));
}

__device__ __forceinline__ uint32_t remap_pixel(const uint8_t* __restrict__ src, long long stride, int sw, int sh, uint32_t xyw, uint32_t f)
{
    const int sx = (int)(short)(xyw & 0xFFFF), sy = (int)(short)(xyw >> 16);
    const int a = (int)(f & 31), b = (int)(f >> 5);
    // Q15 weights of the 32 x 32 table: products of the two 1-D linear kernels, exact; the entry for a zero fraction is
    // (32767, 0, 0, 1): 32768 does not fit a short and the table's sum correction puts the missing unit on the last tap
    int w00 = (32 - b) * (32 - a) * 32, w01 = (32 - b) * a * 32, w10 = b * (32 - a) * 32, w11 = b * a * 32;
    if (f == 0) { w00 = 32767; w11 = 1; }
    int v00, v01, v10, v11;
    if ((unsigned)sx < (unsigned)max(sw - 1, 0) && (unsigned)sy < (unsigned)max(sh - 1, 0)) {
        const uint8_t* p = src + (long long)sy * stride + sx;
        v00 = p[0]; v01 = p[1]; v10 = p[stride]; v11 = p[stride + 1];
    } else {
        if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return 0;           // wholly outside: the border value
        const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw;
        const bool y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
        v00 = x0 && y0 ? src[(long long)sy * stride + sx] : 0;
        v01 = x1 && y0 ? src[(long long)sy * stride + sx + 1] : 0;
        v10 = x0 && y1 ? src[(long long)(sy + 1) * stride + sx] : 0;
        v11 = x1 && y1 ? src[(long long)(sy + 1) * stride + sx + 1] : 0;
    }
    const int sum = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    return (uint32_t)min((sum + (1 << 14)) >> 15, 255);          // the table form, kept literal on this (rare) path
}

struct __attribute__((packed, aligned(1))) u64_unaligned { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) u128_aligned4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) u64_aligned2 { uint32_t x, y; };

// The Q15 table entry for fraction (a, b) / 32 is 32 * (32-b or b) * (32-a or a) (exact), so the blend equals
// ((32-b) * ((32-a) v00 + a v01) + b * ((32-a) v10 + a v11) + 512) >> 10; the zero-fraction entry (32767, 0, 0, 1) gives
// v00 + ((v11 - v00 + 16384) >> 15) = v00, which the same expression yields.  wx = (32-a) | a << 8 and sel picks the two
// adjacent source bytes of a row out of an 8-byte window.
__device__ __forceinline__ uint32_t blend_rows(uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi, uint32_t sel, uint32_t wx, uint32_t ib, uint32_t b)
{
    const uint32_t top = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(a_hi, a_lo, sel), wx, 0u, false);
    const uint32_t bot = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(b_hi, b_lo, sel), wx, 0u, false);
    return (ib * top + b * bot + 512u) >> 10;
}

constexpr int kRemapFrames = 4;      // frames per thread: the map entries and weights of a quad are set up once for all of them

__global__ __launch_bounds__(256) void k_remap(const uint8_t* __restrict__ src, long long src_stride, long long src_frame, int sw, int sh,
                                               const uint32_t* __restrict__ xy, const uint16_t* __restrict__ frac, int dw, int dh, int qpr,
                                               uint8_t* __restrict__ dst, long long dst_stride, long long dst_frame, int n_frames)
{
    const int q = blockIdx.x * 256 + threadIdx.x;          // quad index inside the frame
    if (q >= qpr * dh) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const int f0 = blockIdx.y * kRemapFrames, nf = min(kRemapFrames, n_frames - f0);
    const uint8_t* s = src + (long long)f0 * src_frame;
    uint8_t* d = dst + (long long)f0 * dst_frame + (long long)y * dst_stride + x;
    const size_t m = (size_t)y * dw + x;
    if (x + 4 <= dw) {
        const u128_aligned4 p = *reinterpret_cast<const u128_aligned4*>(xy + m);
        const u64_aligned2 fr = *reinterpret_cast<const u64_aligned2*>(frac + m);
        const uint32_t pk[4] = {p.x, p.y, p.z, p.w};
        const uint32_t fk[4] = {fr.x & 0xFFFFu, fr.x >> 16, fr.y & 0xFFFFu, fr.y >> 16};
        // smooth maps: the four pixels read at most three adjacent source rows and eight adjacent columns, so three 8-byte
        // loads serve all sixteen taps (a pixel whose row is one below the quad's first takes rows 1 and 2)
        int sxk[4], syk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { sxk[k] = (int)(short)(pk[k] & 0xFFFF); syk[k] = (int)(short)(pk[k] >> 16); }
        const int sxb = min(min(sxk[0], sxk[1]), min(sxk[2], sxk[3])), syb = min(min(syk[0], syk[1]), min(syk[2], syk[3]));
        bool fast = syb >= 0 && sxb >= 0 && sxb + 8 <= sw;
        uint32_t sel[4], wx[4], wb[4], wib[4];
        bool down[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = sxk[k] - sxb, dy = syk[k] - syb;
            fast = fast && (unsigned)off <= 6u && (unsigned)dy <= 1u && syk[k] < sh - 1;
            down[k] = dy != 0;
            const uint32_t a = fk[k] & 31u;
            sel[k] = 0x0c0c0000u | (uint32_t)off | ((uint32_t)(off + 1) << 8);
            wx[k] = (32u - a) | (a << 8);
            wb[k] = fk[k] >> 5; wib[k] = 32u - wb[k];
        }
        const bool aligned = (((uintptr_t)d) & 3) == 0 && (dst_frame & 3) == 0;
        if (fast) {
            const uint8_t* r0 = s + (long long)syb * src_stride + sxb;
            const long long third = (long long)(min(syb + 2, sh - 1) - syb) * src_stride;     // row syb+2 exists whenever a pixel needs it
            u64_unaligned ra[kRemapFrames], rb[kRemapFrames], rc[kRemapFrames];
#pragma unroll
            for (int f = 0; f < kRemapFrames; ++f)
                if (f < nf) {
                    ra[f] = *reinterpret_cast<const u64_unaligned*>(r0 + f * src_frame);
                    rb[f] = *reinterpret_cast<const u64_unaligned*>(r0 + f * src_frame + src_stride);
                    rc[f] = *reinterpret_cast<const u64_unaligned*>(r0 + f * src_frame + third);
                }
#pragma unroll
            for (int f = 0; f < kRemapFrames; ++f)
                if (f < nf) {
                    uint32_t out = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        out |= blend_rows(down[k] ? rb[f].x : ra[f].x, down[k] ? rb[f].y : ra[f].y, down[k] ? rc[f].x : rb[f].x,
                                          down[k] ? rc[f].y : rb[f].y, sel[k], wx[k], wib[k], wb[k]) << (8 * k);
                    uint8_t* o = d + f * dst_frame;
                    if (aligned) *reinterpret_cast<uint32_t*>(o) = out;
                    else for (int k = 0; k < 4; ++k) o[k] = (uint8_t)(out >> (8 * k));
                }
        } else {
            for (int f = 0; f < nf; ++f) {
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) out |= remap_pixel(s + f * src_frame, src_stride, sw, sh, pk[k], fk[k]) << (8 * k);
                uint8_t* o = d + f * dst_frame;
                if (aligned) *reinterpret_cast<uint32_t*>(o) = out;
                else for (int k = 0; k < 4; ++k) o[k] = (uint8_t)(out >> (8 * k));
            }
        }
    } else {
        for (int f = 0; f < nf; ++f)
            for (int k = 0; k < 4 && x + k < dw; ++k)
                d[f * dst_frame + k] = (uint8_t)remap_pixel(s + f * src_frame, src_stride, sw, sh, xy[m + k], frac[m + k]);
    }
}

} // namespace

struct orbx_remapper {
    int device = 0, sw = 0, sh = 0, dw = 0, dh = 0;
    uint32_t* d_xy = nullptr; uint16_t* d_frac = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *d_src = nullptr, *d_dst = nullptr, *h_src = nullptr, *h_dst = nullptr;     // staging of the host entry point
};

extern "C" {

int orbx_remap_create(int device, const float* map_x, const float* map_y, ptrdiff_t map_stride, int dst_w, int dst_h,
                      int src_w, int src_h, orbx_remapper** out)
{
    if (!out) return ORBX_ERR_BAD_ARG;
    *out = nullptr;
    if (!map_x || !map_y || dst_w < 1 || dst_h < 1 || src_w < 1 || src_h < 1 || src_w > 32767 || src_h > 32767 || map_stride < dst_w) {
        set_error("orbx_remap_create: bad argument"); return ORBX_ERR_BAD_ARG;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return ORBX_ERR_NO_DEVICE; }
    if (device >= count) { set_error("device index out of range"); return ORBX_ERR_BAD_ARG; }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    orbx_remapper* r = new orbx_remapper();
    struct Guard { orbx_remapper* p; ~Guard() { orbx_remap_destroy(p); } } guard{r};
    HIP_CHECK(hipGetDevice(&r->device));
    r->sw = src_w; r->sh = src_h; r->dw = dst_w; r->dh = dst_h;
    HIP_CHECK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    const size_t n = (size_t)dst_w * dst_h;
    struct Maps { float *x = nullptr, *y = nullptr; ~Maps() { if (x) (void)hipFree(x); if (y) (void)hipFree(y); } } maps;
    HIP_CHECK(hipMalloc(&maps.x, sizeof(float) * n)); HIP_CHECK(hipMalloc(&maps.y, sizeof(float) * n));
    float *d_mx = maps.x, *d_my = maps.y;
    HIP_CHECK(hipMalloc(&r->d_xy, sizeof(uint32_t) * n)); HIP_CHECK(hipMalloc(&r->d_frac, sizeof(uint16_t) * n));
    HIP_CHECK(hipMemcpy2D(d_mx, sizeof(float) * dst_w, map_x, sizeof(float) * map_stride, sizeof(float) * dst_w, dst_h, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy2D(d_my, sizeof(float) * dst_w, map_y, sizeof(float) * map_stride, sizeof(float) * dst_w, dst_h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_remap_prepare, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, r->stream, d_mx, d_my, (long long)dst_w, dst_w, dst_h, r->d_xy, r->d_frac);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(r->stream));
    guard.p = nullptr;
    *out = r;
    return ORBX_OK;
}

void orbx_remap_destroy(orbx_remapper* r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    void* ptrs[] = {r->d_xy, r->d_frac, r->d_src, r->d_dst};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (r->h_src) (void)hipHostFree(r->h_src);
    if (r->h_dst) (void)hipHostFree(r->h_dst);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

int orbx_remap_batch_device(orbx_remapper* r, const uint8_t* d_src, ptrdiff_t src_row_stride, ptrdiff_t src_frame_stride, int n_frames,
                            uint8_t* d_dst, ptrdiff_t dst_row_stride, ptrdiff_t dst_frame_stride, void* stream)
{
    if (!r || !d_src || !d_dst || n_frames < 0 || src_row_stride < r->sw || dst_row_stride < r->dw) { set_error("orbx_remap_batch_device: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (n_frames == 0) return ORBX_OK;
    HIP_CHECK(hipSetDevice(r->device));
    const int qpr = (r->dw + 3) / 4;
    hipLaunchKernelGGL(k_remap, dim3((unsigned)(((long long)qpr * r->dh + 255) / 256), (n_frames + kRemapFrames - 1) / kRemapFrames), dim3(256), 0,
                       (hipStream_t)stream, d_src, (long long)src_row_stride, (long long)src_frame_stride, r->sw, r->sh, r->d_xy, r->d_frac,
                       r->dw, r->dh, qpr, d_dst, (long long)dst_row_stride, (long long)dst_frame_stride, n_frames);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_remap_image(orbx_remapper* r, const uint8_t* src, ptrdiff_t src_stride, uint8_t* dst, ptrdiff_t dst_stride)
{
    if (!r || !src || !dst || src_stride < r->sw || dst_stride < r->dw) { set_error("orbx_remap_image: bad argument"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(r->device));
    const size_t sbytes = (size_t)r->sw * r->sh, dbytes = (size_t)r->dw * r->dh;
    if (!r->d_src) {
        HIP_CHECK(hipMalloc(&r->d_src, sbytes)); HIP_CHECK(hipMalloc(&r->d_dst, dbytes));
        HIP_CHECK(hipHostMalloc(&r->h_src, sbytes, hipHostMallocDefault)); HIP_CHECK(hipHostMalloc(&r->h_dst, dbytes, hipHostMallocDefault));
    }
    for (int y = 0; y < r->sh; ++y) std::memcpy(r->h_src + (size_t)y * r->sw, src + (ptrdiff_t)y * src_stride, (size_t)r->sw);
    HIP_CHECK(hipMemcpyAsync(r->d_src, r->h_src, sbytes, hipMemcpyHostToDevice, r->stream));
    const int rc = orbx_remap_batch_device(r, r->d_src, r->sw, (ptrdiff_t)sbytes, 1, r->d_dst, r->dw, (ptrdiff_t)dbytes, r->stream);
    if (rc) return rc;
    HIP_CHECK(hipMemcpyAsync(r->h_dst, r->d_dst, dbytes, hipMemcpyDeviceToHost, r->stream));
    HIP_CHECK(hipStreamSynchronize(r->stream));
    for (int y = 0; y < r->dh; ++y) std::memcpy(dst + (ptrdiff_t)y * dst_stride, r->h_dst + (size_t)y * r->dw, (size_t)r->dw);
    return ORBX_OK;
}

} // extern "C"
