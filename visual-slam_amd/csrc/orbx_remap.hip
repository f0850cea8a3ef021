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
    return (uint32_t)min((sum + (1 << 14)) >> 15, 255);
}

struct __attribute__((packed, aligned(1))) u64_unaligned { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) u128_aligned4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) u64_aligned2 { uint32_t x, y; };

__device__ __forceinline__ uint32_t blend(int v00, int v01, int v10, int v11, uint32_t f)
{
    const int a = (int)(f & 31), b = (int)(f >> 5);
    int w00 = (32 - b) * (32 - a) * 32, w01 = (32 - b) * a * 32, w10 = b * (32 - a) * 32, w11 = b * a * 32;
    if (f == 0) { w00 = 32767; w11 = 1; }
    return (uint32_t)min((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15, 255);
}

__global__ __launch_bounds__(256) void k_remap(const uint8_t* __restrict__ src, long long src_stride, long long src_frame, int sw, int sh,
                                               const uint32_t* __restrict__ xy, const uint16_t* __restrict__ frac, int dw, int dh, int qpr,
                                               uint8_t* __restrict__ dst, long long dst_stride, long long dst_frame)
{
    const int q = blockIdx.x * 256 + threadIdx.x;          // quad index inside the frame
    if (q >= qpr * dh) return;
    const int y = q / qpr, x = (q - y * qpr) * 4;
    const uint8_t* s = src + (long long)blockIdx.y * src_frame;
    uint8_t* d = dst + (long long)blockIdx.y * dst_frame + (long long)y * dst_stride + x;
    const size_t m = (size_t)y * dw + x;
    uint32_t out = 0;
    bool done = false;
    if (x + 4 <= dw) {
        // smooth maps: the four pixels read one source row pair and at most eight adjacent columns, so two 8-byte loads
        // serve all sixteen taps
        const u128_aligned4 p = *reinterpret_cast<const u128_aligned4*>(xy + m);
        const u64_aligned2 fr = *reinterpret_cast<const u64_aligned2*>(frac + m);
        const uint32_t pk[4] = {p.x, p.y, p.z, p.w};
        const uint32_t fk[4] = {fr.x & 0xFFFFu, fr.x >> 16, fr.y & 0xFFFFu, fr.y >> 16};
        const int sy = (int)(short)(p.x >> 16), sx0 = (int)(short)(p.x & 0xFFFF);
        bool fast = (unsigned)sy < (unsigned)max(sh - 1, 0) && sx0 >= 0 && sx0 + 8 <= sw;
        int off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            off[k] = (int)(short)(pk[k] & 0xFFFF) - sx0;
            fast = fast && (int)(short)(pk[k] >> 16) == sy && (unsigned)off[k] <= 6u;
        }
        if (fast) {
            const uint8_t* r0 = s + (long long)sy * src_stride + sx0;
            const u64_unaligned a = *reinterpret_cast<const u64_unaligned*>(r0), b = *reinterpret_cast<const u64_unaligned*>(r0 + src_stride);
            const unsigned long long ra = ((unsigned long long)a.y << 32) | a.x, rb = ((unsigned long long)b.y << 32) | b.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sh8 = 8 * off[k];
                out |= blend((int)((ra >> sh8) & 255), (int)((ra >> (sh8 + 8)) & 255), (int)((rb >> sh8) & 255), (int)((rb >> (sh8 + 8)) & 255), fk[k]) << (8 * k);
            }
            done = true;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) out |= remap_pixel(s, src_stride, sw, sh, pk[k], fk[k]) << (8 * k);
            done = true;
        }
    }
    if (!done) {
        for (int k = 0; k < 4; ++k)
            if (x + k < dw) out |= remap_pixel(s, src_stride, sw, sh, xy[m + k], frac[m + k]) << (8 * k);
    }
    if (x + 4 <= dw && (((uintptr_t)d) & 3) == 0) *reinterpret_cast<uint32_t*>(d) = out;
    else for (int k = 0; k < 4 && x + k < dw; ++k) d[k] = (uint8_t)(out >> (8 * k));
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
    HIP_CHECK(hipGetDevice(&r->device));
    r->sw = src_w; r->sh = src_h; r->dw = dst_w; r->dh = dst_h;
    HIP_CHECK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    const size_t n = (size_t)dst_w * dst_h;
    float *d_mx = nullptr, *d_my = nullptr;
    HIP_CHECK(hipMalloc(&d_mx, sizeof(float) * n)); HIP_CHECK(hipMalloc(&d_my, sizeof(float) * n));
    HIP_CHECK(hipMalloc(&r->d_xy, sizeof(uint32_t) * n)); HIP_CHECK(hipMalloc(&r->d_frac, sizeof(uint16_t) * n));
    HIP_CHECK(hipMemcpy2D(d_mx, sizeof(float) * dst_w, map_x, sizeof(float) * map_stride, sizeof(float) * dst_w, dst_h, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy2D(d_my, sizeof(float) * dst_w, map_y, sizeof(float) * map_stride, sizeof(float) * dst_w, dst_h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_remap_prepare, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, r->stream, d_mx, d_my, (long long)dst_w, dst_w, dst_h, r->d_xy, r->d_frac);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(r->stream));
    (void)hipFree(d_mx); (void)hipFree(d_my);
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
    hipLaunchKernelGGL(k_remap, dim3((unsigned)(((long long)qpr * r->dh + 255) / 256), n_frames), dim3(256), 0, (hipStream_t)stream, d_src,
                       (long long)src_row_stride, (long long)src_frame_stride, r->sw, r->sh, r->d_xy, r->d_frac, r->dw, r->dh, qpr, d_dst,
                       (long long)dst_row_stride, (long long)dst_frame_stride);
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
