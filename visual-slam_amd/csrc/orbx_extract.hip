// MI355X (gfx950) kernels and context of the ORB extractor: ORBextractor::operator()
// (src/ORBextractor.cc:1042-1104) for a batch of equally sized frames resident in HBM.
//
// Launch sequence per batch (all on one stream, grids over (tile, frame)):
//   k_level0      copy + REFLECT_101 border                      ComputePyramid :1124-1128
//   k_resize x(L-1)  fixed-point bilinear + border, level l from l-1  ComputePyramid :1117-1123
//   k_fast_cells  per-cell FAST-9 score, NMS, threshold fallback   ComputeKeyPointsOctTree :788-828
//   k_octree      quadtree selection, one workgroup per level      DistributeOctTree :538-762
//   k_blur        7x7 Gaussian of every level                      :1084-1085
//   k_describe    IC angle, rBRIEF on the blurred level, output record  :76-146, :1086-1102
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <thread>
#include <string>
#include "orbx_internal.h"
#include "octree_algo.h"

using namespace orbx;

#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
            return ORBX_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

namespace orbx {
int raise_dynamic_lds(const void* fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> high;       // (kernel, device) -> limit set so far
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& h = high[std::make_pair(fn, dev)];
    if (bytes <= h) return (int)hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) h = bytes;
    return (int)e;
}
} // namespace orbx

namespace {

// Per-level constants as the kernels see them.
struct DevLevel {
    int w, h, stride, rows;
    unsigned long long offset, roi;
    int maxBorderX, maxBorderY;
    int cell_first, cell_count, cand_offset, cand_cap;
    int quota, nIni;
    float hX;
    int kp_offset, kp_cap;
    int area2x, rx_offset, ry_offset, rb_offset;
    float scale, kp_size;
    int key_offset;   // offset of this level's compact key array inside one frame's key scratch
    int fast_resize;  // every aligned group of 4 destination columns reads <= 8 consecutive source bytes
    int rg_offset;    // first group record of this level
    int gpr;          // groups per row of the resize kernel
    unsigned gpr_magic;
    int cpr;          // 16-byte chunks per ROI row, ceil(w/16)
    unsigned cpr_magic; // ceil(2^32 / cpr): row of flat chunk index i is umulhi(i, cpr_magic)
    int tpr;            // 16-byte columns per row of the tiled blurred level
    int upr;            // 16-pixel units per row of the blur kernel
    unsigned upr_magic;
    unsigned long long blur_offset;   // of the tiled blurred level inside one frame's blurred block
    int border_threads; // threads of k_borders for this level
    float inv_scale;    // mvInvScaleFactor[level]
    int direct_ok;      // a key's (cell, index inside the cell) fits the 24-bit rank of the quadtree's direct form
};

typedef unsigned short us16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ uint32_t load_u32(const uint8_t* p) { return *reinterpret_cast<const u32_unaligned*>(p); }

// Sum over the 64 lanes as a wave-uniform value: four DPP adds inside each row of 16 lanes (lane pairs, quad halves,
// rotations by 4 and 8), then the four row sums meet in scalar registers.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);      // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);      // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);     // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);     // row_ror:8
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}


// LDS hand-off between the lanes of one wavefront: DS operations of a wave execute in issue order,
// so only the compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Frame-major work split over grids of 8 x chunks x ceil(B/8) workgroups.  Workgroups are dealt round-robin over the
// 8 XCDs in linear order, x fastest (observed behaviour, relied on for locality only), so workgroups with the same
// blockIdx.x share an L2: all chunks of frame f go to x = f mod 8, one frame after the other, and the lines of a frame
// that several chunks touch are fetched from memory once.  (Three grid dimensions instead of one spare every wavefront
// a division by the chunk count.)
__device__ __forceinline__ bool frame_chunk(int n_frames, int& frame, uint32_t& chunk)
{
    chunk = blockIdx.y;
    frame = (int)(blockIdx.z * 8u + blockIdx.x);
    return frame < n_frames;
}
__host__ inline dim3 frame_chunk_grid(unsigned chunks, int n_frames) { return dim3(8u, chunks, (unsigned)((n_frames + 7) / 8)); }

__device__ __forceinline__ int reflect101(int p, int len)
{
    // one reflection is enough for |overshoot| <= 19 < len
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Level 0: copy the input into the padded pyramid buffer and write the REFLECT_101 frame.
// One thread writes one aligned dword of a padded row.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Pyramid.  Interior pixels are produced 16 per thread in flat (row-major) order, so a wavefront writes
// 1 KiB of consecutive bytes; the REFLECT_101 frames of all levels are filled by one separate launch
// (a frame pixel is a copy of the interior pixel it mirrors), which keeps the interior kernels free of
// divergent border code.
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) u128_unaligned { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) u64_unaligned { uint32_t x, y; };

// flat chunk index -> (row, chunk in row) with a precomputed reciprocal (exact for the index range used)
__device__ __forceinline__ void split_index(uint32_t i, uint32_t cpr, uint32_t magic, uint32_t& row, uint32_t& col)
{
    row = __umulhi(i, magic);
    col = i - row * cpr;
}

// Store chunk ch of ROI row y and the parts of the REFLECT_101 frame that mirror it: the dword left of the ROI
// (columns -4..-1 repeat columns 4..1) with chunk 0, and frame row -j / h-1+j with ROI row j / h-1-j
// (j = 1..kFrameFill).  Frame columns right of the ROI are ordinary columns of the last chunk(s).
__device__ __forceinline__ void store_chunk_with_frame(uint8_t* roi, int stride, int h, uint32_t y, uint32_t ch, const uint4 v)
{
    uint8_t* row = roi + (unsigned long long)y * stride;
    const uint32_t left = __builtin_amdgcn_perm(v.y, v.x, 0x01020304u);
    *reinterpret_cast<uint4*>(row + 16 * ch) = v;
    if (ch == 0) *reinterpret_cast<uint32_t*>(row - 4) = left;
    const bool top = y - 1u < (uint32_t)kFrameFill, bottom = (uint32_t)(h - 2) - y < (uint32_t)kFrameFill;
    if (top || bottom) {
        uint8_t* mrow = roi + (long long)(top ? -(int)y : 2 * (h - 1) - (int)y) * stride;
        *reinterpret_cast<uint4*>(mrow + 16 * ch) = v;
        if (ch == 0) *reinterpret_cast<uint32_t*>(mrow - 4) = left;
    }
}

// cv::cvtColor to grey for 8-bit pixels (the conversion Tracking::GrabImage* applies before extraction,
// src/Tracking.cc:254-267): (c0*p0 + c1*p1 + c2*p2 + 2^13) >> 14 with the Q14 weights R 4899, G 9617, B 1868
// in the channel order of the input; a fourth channel is ignored.
template <int CH> __device__ __forceinline__ uint32_t grey_at(const uint8_t* row, int x, int c0, int c1, int c2)
{
    if (CH == 1) return row[x];
    const uint8_t* p = row + CH * x;
    return (uint32_t)(p[0] * c0 + p[1] * c1 + p[2] * c2 + 8192) >> 14;
}

template <int CH>
__global__ __launch_bounds__(256) void k_level0_colour(const uint8_t* __restrict__ img, long long row_stride, long long frame_stride,
                                                       uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, DevLevel L, int n_frames,
                                                       int c0, int c1, int c2)
{
    int frame; uint32_t blk;
    if (!frame_chunk(n_frames, frame, blk)) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= (uint32_t)(L.cpr * L.h)) return;
    uint32_t y, ch;
    split_index(i, L.cpr, L.cpr_magic, y, ch);
    const uint8_t* row = img + (long long)frame * frame_stride + (long long)y * row_stride;
    uint32_t t[4] = {0, 0, 0, 0};
    if ((int)(16 * ch + 16) <= L.w) {
        // 16 pixels = CH*4 dwords; the byte positions are compile-time constants, so each channel is one bit-field extract
        uint32_t d[CH * 4];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const u128_unaligned v = *reinterpret_cast<const u128_unaligned*>(row + CH * 16 * ch + 16 * k);
            d[4 * k] = v.x; d[4 * k + 1] = v.y; d[4 * k + 2] = v.z; d[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int b = CH * k;
            const uint32_t p0 = (d[b >> 2] >> (8 * (b & 3))) & 255u, p1 = (d[(b + 1) >> 2] >> (8 * ((b + 1) & 3))) & 255u,
                           p2 = (d[(b + 2) >> 2] >> (8 * ((b + 2) & 3))) & 255u;
            t[k >> 2] |= ((p0 * c0 + p1 * c1 + p2 * c2 + 8192u) >> 14) << (8 * (k & 3));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = (int)(16 * ch) + k;
            if (x < L.w + kFrameFill) t[k >> 2] |= grey_at<CH>(row, reflect101(x, L.w), c0, c1, c2) << (8 * (k & 3));
        }
    }
    store_chunk_with_frame(pyr + (unsigned long long)frame * pyr_bytes + L.roi, L.stride, L.h, y, ch, make_uint4(t[0], t[1], t[2], t[3]));
}

__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ img, long long row_stride, long long frame_stride,
                                                uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, DevLevel L, int n_frames)
{
    int frame; uint32_t blk;
    if (!frame_chunk(n_frames, frame, blk)) return;
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= (uint32_t)(L.cpr * L.h)) return;
    uint32_t y, ch;
    split_index(i, L.cpr, L.cpr_magic, y, ch);
    const uint8_t* s = img + (long long)frame * frame_stride + (long long)y * row_stride + 16 * ch;
    uint4 v;
    if ((int)(16 * ch + 16) <= L.w) {
        const u128_unaligned t = *reinterpret_cast<const u128_unaligned*>(s);
        v = make_uint4(t.x, t.y, t.z, t.w);
    } else {                                   // last chunks of a row: the ROI's end and the frame columns right of it
        uint32_t t[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = (int)(16 * ch) + k;
            if (x < L.w + kFrameFill) t[k >> 2] |= (uint32_t)s[reflect101(x, L.w) - (int)(16 * ch)] << (8 * (k & 3));
        }
        v = make_uint4(t[0], t[1], t[2], t[3]);
    }
    store_chunk_with_frame(pyr + (unsigned long long)frame * pyr_bytes + L.roi, L.stride, L.h, y, ch, v);
}

// Level 0 already in place (the frames were written into the context's input view): only the part of the REFLECT_101
// frame that later stages read is missing.  Per frame: threads 0 .. h-1 take one ROI row each (the dword left of it repeats
// columns 4..1, the three pixels right of it repeat columns w-2..w-4, written as the two dwords that hold them), then
// 6 * cpr threads copy the three rows above and below the ROI (rows 1..3 and h-2..h-4 with the side pixels they were just
// given -- by other threads, so those six rows get their side pixels computed here again, not read back).
__device__ __forceinline__ void level0_frame_block(uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, const DevLevel& L, int frame, uint32_t blk)
{
    const uint32_t i = blk * 256 + threadIdx.x;
    uint8_t* roi = pyr + (unsigned long long)frame * pyr_bytes + L.roi;
    const uint32_t nside = (uint32_t)L.h, ncopy = 6u * (uint32_t)L.cpr;
    if (i < nside) {
        uint8_t* row = roi + (unsigned long long)i * L.stride;
        const uint2 a = *reinterpret_cast<const uint2*>(row);                       // columns 0..7 (rows are 32-byte aligned)
        *reinterpret_cast<uint32_t*>(row - 4) = __builtin_amdgcn_perm(a.y, a.x, 0x01020304u);
        // columns w .. w+2 = columns w-2, w-3, w-4; the bytes around them (the ROI's last columns, the column after) are kept
        uint8_t t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = row[L.w - 2 - k];
#pragma unroll
        for (int k = 0; k < 3; ++k) row[L.w + k] = t[k];
    } else if (i < nside + ncopy) {
        const uint32_t j = i - nside, r = j / (uint32_t)L.cpr, ch = j - r * (uint32_t)L.cpr;     // r 0..2: above, 3..5: below
        const int src = r < 3 ? (int)r + 1 : L.h - 2 - (int)(r - 3), dst = r < 3 ? -(int)r - 1 : L.h + (int)(r - 3);
        const uint8_t* srow = roi + (long long)src * L.stride;
        uint8_t* drow = roi + (long long)dst * L.stride;
        uint4 v = *reinterpret_cast<const uint4*>(srow + 16 * ch);
        if ((int)(16 * ch + 16) > L.w) {                                           // the chunk reaches the frame columns right of the ROI
            uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int x = (int)(16 * ch) + k;
                if (x >= L.w && x < L.w + kFrameFill) w4[k >> 2] = (w4[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((uint32_t)srow[2 * L.w - 2 - x] << (8 * (k & 3)));
            }
            v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        *reinterpret_cast<uint4*>(drow + 16 * ch) = v;
        if (ch == 0) *reinterpret_cast<uint32_t*>(drow - 4) = __builtin_amdgcn_perm(v.y, v.x, 0x01020304u);
    }
}

__global__ __launch_bounds__(256) void k_level0_frame(uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, DevLevel L, int n_frames)
{
    int frame; uint32_t blk;
    if (!frame_chunk(n_frames, frame, blk)) return;
    level0_frame_block(pyr, pyr_bytes, L, frame, blk);
}

// Level l from level l-1: cv::resize INTER_LINEAR for 8-bit (Q11 weights, two-stage rounding).
// A thread owns one group of four adjacent destination columns over four consecutive rows, and consecutive
// lanes own consecutive groups, so each load or store of a wavefront touches a few cache lines.  The four
// columns read at most 8 consecutive source bytes per row: they arrive as one 4-byte-aligned 12-byte load
// (an unaligned 8-byte load costs the address unit about twice as much) shifted into place by two
// v_alignbyte_b32; one v_perm_b32 pulls a column's two neighbours out of that window as 16-bit lanes, one
// v_dot2_u32_u16 applies the weights.  All loads of a thread are issued before the first store.
// kSharedRows: the four destination rows read source rows base+r or base+r+1 (and the row below), which holds
// for every scale factor up to 4/3; then six source rows are loaded instead of eight and the row constants come
// as one 16-byte record per row block.  Otherwise each destination row loads its own two rows.
// Frame columns are ordinary groups (their table entries repeat the mirror column); frame rows are second
// stores of the rows they mirror.
constexpr int kResizeRows = 4;
struct __attribute__((packed, aligned(4))) u96_aligned4 { uint32_t x, y, z; };

__device__ __forceinline__ uint2 resize_window(const uint8_t* p, uint32_t shift)     // 8 bytes at p + shift, p 4-byte aligned
{
    const u96_aligned4 t = *reinterpret_cast<const u96_aligned4*>(p);
    return make_uint2(__builtin_amdgcn_alignbyte(t.y, t.x, shift), __builtin_amdgcn_alignbyte(t.z, t.y, shift));
}

// bits 32..47 of the product of two 24-bit values (v_mul_hi_u32_u24)
__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b)
{
    return (uint32_t)(((unsigned long long)(a & 0xFFFFFFu) * (unsigned long long)(b & 0xFFFFFFu)) >> 32);
}

// frame_blocks: workgroups behind the resize's own that write the REFLECT_101 frame of the source level (level 0 when the
// frames came by the input view: nothing of the resize reads those pixels -- what its windows touch of them meets a
// zero weight -- and the blur, which does, runs behind all the resizes; as a launch of its own the three workgroups per
// frame cost a whole launch latency)
template <bool kSharedRows>
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, DevLevel S, DevLevel D,
                                                const ResizeY* __restrict__ ryt, const ResizeRowBlock* __restrict__ rrb,
                                                const ResizeGroupA* __restrict__ rga, const ResizeGroupB* __restrict__ rgb, int n_frames,
                                                uint32_t resize_blocks)
{
    const uint32_t nthreads = (uint32_t)(D.gpr * ((D.h + kResizeRows - 1) / kResizeRows));
    int frame; uint32_t blk;
    if (!frame_chunk(n_frames, frame, blk)) return;
    if (blk >= resize_blocks) { level0_frame_block(pyr, pyr_bytes, S, frame, blk - resize_blocks); return; }
    const uint32_t i = blk * 256 + threadIdx.x;
    if (i >= nthreads) return;
    uint32_t rb, gx;
    split_index(i, D.gpr, D.gpr_magic, rb, gx);
    uint8_t* base = pyr + (unsigned long long)frame * pyr_bytes;
    const uint4 ta = *reinterpret_cast<const uint4*>(rga + D.rg_offset + gx);   // sx_first, 4 byte offsets, w0, w1
    const uint2 tb = *reinterpret_cast<const uint2*>(rgb + D.rg_offset + gx);   // w2, w3
    const uint32_t wt[4] = {ta.z, ta.w, tb.x, tb.y};
    uint32_t sel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t off = (ta.y >> (8 * k)) & 0xFFu;
        sel[k] = (off | (off << 16)) + 0x0c010c00u;                             // bytes off, off+1 -> 16-bit lanes 0, 1
    }
    // all products below fit 24-bit operands: rows < 2^15, strides < 2^13, weights <= 2048, sums >> 4 < 2^15
    // addresses: a wavefront-uniform base plus an unsigned 32-bit lane offset, so that no load or store needs 64-bit vector
    // arithmetic (a tenth of the kernel's vector instructions otherwise).  First source columns are never negative (frame
    // groups repeat their mirror column); the destination base sits kFrameFill rows and one group left / above the ROI.
    const uint32_t shift = ta.x & 3u;                                           // level origins and strides are multiples of 32
    const uint8_t* sbase = base + S.roi;
    const uint32_t scol = ta.x - shift;
    uint8_t* dbase = base + D.roi - 4 - (long long)kFrameFill * D.stride;
    const uint32_t dcol = 4u * gx + (uint32_t)kFrameFill * (uint32_t)D.stride;
    const int y0 = (int)rb * kResizeRows;
    uint2 r0[kResizeRows], r1[kResizeRows];
    uint32_t b0[kResizeRows], b1[kResizeRows];
    if constexpr (kSharedRows) {
        // The row block's record by scalar loads: the 64 consecutive threads of a wavefront lie in one row block or two
        // (three only on levels narrower than 256 pixels), so its record and the next one's come as wavefront constants
        // and a lane picks its own, instead of 64 lanes asking the address unit for the same 16 bytes (3840x2160: pyramid
        // stage 470 -> 458 us per 32 frames; 640x480: no change -- the row loads wait for the record either way).
        const uint32_t rb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb);
        const uint4* rec_of = reinterpret_cast<const uint4*>(rrb + D.rb_offset);
        uint4 tr;
        if (__all(rb <= rb0 + 1u)) {
            const uint32_t rb1 = min(rb0 + 1u, (uint32_t)((D.h + kResizeRows - 1) / kResizeRows) - 1u);
            const uint4 t0 = rec_of[rb0], t1 = rec_of[rb1];
            const bool first = rb == rb0;
            tr = make_uint4(first ? t0.x : t1.x, first ? t0.y : t1.y, first ? t0.z : t1.z, first ? t0.w : t1.w);
        } else tr = rec_of[rb];
        const uint32_t rec[4] = {tr.x, tr.y, tr.z, tr.w};
        const int srow0 = (int)((tr.x >> 25) | ((tr.y >> 25) << 7) | ((tr.z >> 25) << 14));
        uint2 rows[kResizeRows + 2];
#pragma unroll
        for (int k = 0; k < kResizeRows + 2; ++k)
            rows[k] = resize_window(sbase + (scol + __umul24((uint32_t)min(srow0 + k, S.h - 1), (uint32_t)S.stride)), shift);
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) {
            const bool up = rec[r] & (1u << 24);
            r0[r] = up ? rows[r + 1] : rows[r];
            r1[r] = up ? rows[r + 2] : rows[r + 1];
            b0[r] = rec[r] & 0xFFFu; b1[r] = (rec[r] >> 12) & 0xFFFu;
        }
    } else {
        // rows past the bottom edge repeat the last row and are not stored
        ResizeY ey[kResizeRows];
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) ey[r] = ryt[D.ry_offset + min(y0 + r, D.h - 1)];
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) {
            r0[r] = resize_window(sbase + (scol + __umul24((uint32_t)ey[r].sy0, (uint32_t)S.stride)), shift);
            r1[r] = resize_window(sbase + (scol + __umul24((uint32_t)ey[r].sy1, (uint32_t)S.stride)), shift);
            b0[r] = (uint32_t)(int)ey[r].b0; b1[r] = (uint32_t)(int)ey[r].b1;
        }
    }
    // OpenCV's vertical pass, ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2, with each product as one
    // 24-bit high multiply: (b << 12) * (H & ~15) = b * (H >> 4) * 2^16, whose bits 32.. are the shifted product
    // (b <= 2^11 after the host's check, H < 2^19: both factors fit 24 bits).  The four result bytes are packed by
    // byte selects, which also drop everything above bit 7 as the cast to uchar does.
    uint32_t out[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) {
        const uint32_t c0 = b0[r] << 12, c1 = b1[r] << 12;
        uint32_t b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const us16x2 w = __builtin_bit_cast(us16x2, wt[k]);
            const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(r0[r].y, r0[r].x, sel[k])), w, 0u, false);
            const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(r1[r].y, r1[r].x, sel[k])), w, 0u, false);
            b[k] = mul_hi_u24(c0, h0 & ~15u) + mul_hi_u24(c1, h1 & ~15u) + 2u;             // <= 1026: four times the pixel
        }
        // >> 2 on two sums at a time (16 bits apart: what the shift moves from the upper sum into the lower half lands
        // above the byte that is kept), then one byte select
        // (the pairing as an instruction: the compiler otherwise folds the + 2 of the upper sum into a constant of its own)
        uint32_t u01, u23;
        asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(u01) : "v"(b[1]), "v"(b[0]));
        asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(u23) : "v"(b[3]), "v"(b[2]));
        out[r] = __builtin_amdgcn_perm(u23 >> 2, u01 >> 2, 0x06040200u);
    }
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) {
        const int y = y0 + r;
        if (y >= D.h) break;
        const uint32_t v = out[r];
        *reinterpret_cast<uint32_t*>(dbase + (dcol + __umul24((uint32_t)y, (uint32_t)D.stride))) = v;
        // frame row -j repeats row j, row h-1+j repeats row h-1-j (j = 1..kFrameFill)
        const bool top = (uint32_t)(y - 1) < (uint32_t)kFrameFill, bottom = (uint32_t)(D.h - 2 - y) < (uint32_t)kFrameFill;
        if (top || bottom)
            *reinterpret_cast<uint32_t*>(dbase + (4u * gx + __umul24((uint32_t)(kFrameFill + (top ? -y : 2 * (D.h - 1) - y)), (uint32_t)D.stride))) = v;
    }
}

// Generic form (per-pixel table, exact 2x decimation rerouted to the 2x2 mean as OpenCV does): used for
// levels whose geometry does not fit the 8-byte window of the fast kernel.
__global__ __launch_bounds__(256) void k_resize_generic(uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, DevLevel S, DevLevel D,
                                                        const ResizeX* __restrict__ rxt, const ResizeY* __restrict__ ryt)
{
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= D.w || y >= D.h) return;
    uint8_t* base = pyr + (unsigned long long)blockIdx.z * pyr_bytes;
    const ResizeY ey = ryt[D.ry_offset + y];
    const uint8_t* s0 = base + S.roi + (long long)ey.sy0 * S.stride;
    const uint8_t* s1 = base + S.roi + (long long)ey.sy1 * S.stride;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = min(x0 + k, D.w - 1);
        const ResizeX ex = rxt[D.rx_offset + xx];
        const int p00 = s0[ex.sx], p01 = s0[ex.sx + 1], p10 = s1[ex.sx], p11 = s1[ex.sx + 1];
        uint32_t b;
        if (D.area2x) {
            b = (uint32_t)((p00 + p01 + p10 + p11 + 2) >> 2);
        } else {
            const int h0 = p00 * ex.a0 + p01 * ex.a1;
            const int h1 = p10 * ex.a0 + p11 * ex.a1;
            b = (uint32_t)((((ey.b0 * (h0 >> 4)) >> 16) + ((ey.b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xFFu;
        }
        v |= b << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(base + D.roi + (unsigned long long)y * D.stride + x0) = v;
}

// REFLECT_101 frames of every level (cv::copyMakeBorder, src/ORBextractor.cc:1121-1127).  One thread per
// dword of the frame area: nb rows above and below the ROI, and in the rows between them ld dwords left of
// the ROI and rd dwords from its right end.  No stage reads more than 3 pixels outside the ROI (the Gaussian
// of a pixel at least 1 inside), so the batch path fills nb = 3, ld = 1, rd = 2; the full 19-pixel frame
// (nb = 19, ld = 8, rd = to the end of the padded row) is filled when a padded level is asked for.
__host__ __device__ inline int border_threads(int w, int h, int stride, int nb, int ld, int rdmax)
{
    const int rfirst = (kRoiX + w) >> 2;
    const int rd = min(rdmax, (stride >> 2) - rfirst);
    return 2 * nb * (rfirst + rd - (8 - ld)) + h * (ld + rd);
}
__global__ __launch_bounds__(256) void k_borders(uint8_t* __restrict__ pyr, unsigned long long pyr_bytes, const DevLevel* __restrict__ lv,
                                                 int nb, int ld, int rdmax)
{
    const DevLevel& L = lv[blockIdx.y];
    const int rfirst = (kRoiX + L.w) >> 2;               // first dword touching the right frame
    const int rd = min(rdmax, (L.stride >> 2) - rfirst);
    const int d0 = 8 - ld;                               // first dword of a full frame row
    const int span = rfirst + rd - d0;
    const int side = ld + rd;
    const int nfull = 2 * nb * span;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nfull + L.h * side) return;
    int prow, d;
    if (i < nfull) {
        const int r = i / span;
        d = d0 + (i - r * span);
        prow = r < nb ? kEdge - nb + r : kEdge + L.h + (r - nb);
    } else {
        const int j = i - nfull;
        const int r = j / side, k = j - r * side;
        prow = kEdge + r;
        d = k < ld ? d0 + k : rfirst + (k - ld);
    }
    uint8_t* base = pyr + (unsigned long long)blockIdx.z * pyr_bytes;
    const uint8_t* srow = base + L.roi + (long long)reflect101(prow - kEdge, L.h) * L.stride;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = 4 * d + k - kRoiX;
        uint32_t b = 0;
        if (xx >= -kEdge && xx < L.w + kEdge) b = srow[reflect101(xx, L.w)];
        v |= b << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(base + L.offset + (unsigned long long)prow * L.stride + 4 * d) = v;
}

// ---------------------------------------------------------------------------------------------
// FAST-9 per cell (cv::FAST(cell, kps, T, true) with T = iniThFAST, else minThFAST).
//
// corner at threshold T  <=>  s >= T with  s = max over the 16 arcs of 9 contiguous ring pixels of
// min(one-signed difference) - 1, which is also the score OpenCV stores.  Non-maximum suppression
// compares the 8 neighbours inside the cell's own detection area only (outside counts as 0), so
// the set of local maxima does not depend on T; T only filters it.  A cell therefore needs one
// score pass: keep maxima with s >= ini if there is one, otherwise those with s >= min.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxCellDim = 60;      // cells are at most 59 wide (wCell = ceil(width / floor(width/30)))
constexpr int kSmallCellDim = 38;    // the usual case (wCell 30..38): smaller LDS tiles, more resident wavefronts
constexpr int kSmallQueue = 480;     // survivor list of a usual cell (a third of its pixels): 5 KB of LDS per cell, 32 cells per CU
constexpr int kBigQueue = 1024;
// cells (wavefronts) per workgroup.  One: a workgroup's LDS and wave slots come free when its cell is done, not when the
// slowest of several cells is (cells differ by the corners they hold); measured 333 -> 321 us per 256 frames against four.
constexpr int kFastWavesPerBlock = 1;

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// smallest / largest of sixteen values with three-input instructions (a two-input minimum of 32-bit values issues no
// faster than a three-input one on this chip)
// (written as instructions: left to itself the compiler shares two-input minima between neighbouring triples, which
// needs fewer operations but more issue time here)
__device__ __forceinline__ int min3_op(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int max3_op(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int min2_op(int a, int b) { int r; asm("v_min_i32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int max2_op(int a, int b) { int r; asm("v_max_i32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int min16i(const int (&m)[16])
{
    // three partial chains, so that consecutive instructions do not wait for each other
    const int a = min3_op(min3_op(m[0], m[1], m[2]), m[3], m[4]);
    const int b = min3_op(min3_op(m[5], m[6], m[7]), m[8], m[9]);
    const int c = min3_op(min3_op(m[10], m[11], m[12]), m[13], m[14]);
    return min3_op(min3_op(a, b, c), m[15], m[15]);
}
__device__ __forceinline__ int max16i(const int (&m)[16])
{
    const int a = max3_op(max3_op(m[0], m[1], m[2]), m[3], m[4]);
    const int b = max3_op(max3_op(m[5], m[6], m[7]), m[8], m[9]);
    const int c = max3_op(max3_op(m[10], m[11], m[12]), m[13], m[14]);
    return max3_op(max3_op(a, b, c), m[15], m[15]);
}

// exact score of the pixel at tile address t (ring offsets in tile coordinates, row stride TS).  With ring pixels p and
// centre v: max over the arcs of min over an arc of (v - p) is v - A, A the smallest arc maximum of p; the bright side
// likewise is B - v, B the largest arc minimum -- so the ring is reduced as it is read, without the sixteen differences.
template <int TS>
__device__ __forceinline__ int fast_score(const uint8_t* t)
{
    const int v = t[0];
    const int p[16] = {t[3 * TS], t[3 * TS + 1], t[2 * TS + 2], t[TS + 3], t[3], t[-TS + 3], t[-2 * TS + 2], t[-3 * TS + 1],
                       t[-3 * TS], t[-3 * TS - 1], t[-2 * TS - 2], t[-TS - 3], t[-3], t[TS - 3], t[2 * TS - 2], t[3 * TS - 1]};
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo3[k] = min3_op(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        hi3[k] = max3_op(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
    int arcmax[16], arcmin[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        arcmax[k] = max3_op(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);     // ring k..k+8
        arcmin[k] = min3_op(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    }
    return max(v - min16i(arcmax), max16i(arcmin) - v) - 1;
}

// One side of the exact score: DARK gives v - A (A the smallest arc maximum), otherwise B - v (B the largest arc
// minimum); the pixel is a corner on that side at threshold T iff the value exceeds T, and its score is then the larger
// side's value - 1.  A survivor of the pretest on one side only cannot be a corner on the other side (the other side's
// value is at most its pretest margin, which did not exceed T), so a corner found on one side alone has that side's score.
// flip: 0, or 0xFF for a lane that wants the bright side from the DARK form -- with every pixel complemented, B - v of the
// pixels is v' - A' of the complements (a wavefront whose lanes differ in side pays seventeen exclusive-ors, not two passes).
template <int TS, bool DARK>
__device__ __forceinline__ int fast_side(const uint8_t* t, const int flip = 0)
{
    const int v = t[0] ^ flip;
    const int p[16] = {t[3 * TS] ^ flip, t[3 * TS + 1] ^ flip, t[2 * TS + 2] ^ flip, t[TS + 3] ^ flip, t[3] ^ flip, t[-TS + 3] ^ flip,
                       t[-2 * TS + 2] ^ flip, t[-3 * TS + 1] ^ flip, t[-3 * TS] ^ flip, t[-3 * TS - 1] ^ flip, t[-2 * TS - 2] ^ flip,
                       t[-TS - 3] ^ flip, t[-3] ^ flip, t[TS - 3] ^ flip, t[2 * TS - 2] ^ flip, t[3 * TS - 1] ^ flip};
    int e3[16], two[8];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        e3[k] = DARK ? max3_op(p[k], p[(k + 1) & 15], p[(k + 2) & 15]) : min3_op(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    // The arc k..k+8 is e3[k], e3[k+3], e3[k+6].  Arcs k and k+3 share two of the three, and
    // min(max(a,b,c), max(b,c,d)) = max(b, c, min(a,d)): two arcs cost two operations and leave one value, and the sixteen
    // arcs fall into the eight pairs (k, k+3), k = 0, 6, 12, 2, 8, 14, 4, 10 (mod 16).
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = (6 * i) & 15;
        const int a = e3[k], b = e3[(k + 3) & 15], c = e3[(k + 6) & 15], d = e3[(k + 9) & 15];
        two[i] = DARK ? max3_op(b, c, min2_op(a, d)) : min3_op(b, c, max2_op(a, d));
    }
    if constexpr (DARK) return v - min2_op(min3_op(min3_op(two[0], two[1], two[2]), min3_op(two[3], two[4], two[5]), two[6]), two[7]);
    else return max2_op(max3_op(max3_op(two[0], two[1], two[2]), max3_op(two[3], two[4], two[5]), two[6]), two[7]) - v;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t u) { return __builtin_bit_cast(s16x2, u); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 pmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// bytes POS and POS+1 of the 12-byte window {w0,w1,w2}, zero-extended into two 16-bit lanes
template <int POS>
__device__ __forceinline__ s16x2 pair16(uint32_t w0, uint32_t w1, uint32_t w2)
{
    if constexpr (POS + 1 <= 7)
        return as_s16x2(__builtin_amdgcn_perm(w1, w0, 0x0c000c00u | (uint32_t)POS | ((uint32_t)(POS + 1) << 16)));
    else
        return as_s16x2(__builtin_amdgcn_perm(w2, w1, 0x0c000c00u | (uint32_t)(POS - 4) | ((uint32_t)(POS - 3) << 16)));
}

// Necessary condition for two horizontally adjacent pixels at once (16-bit packed lanes): an arc of 9
// ring pixels holds one pixel of every antipodal pair, so a dark corner needs, in each of the four even pairs
// (0,8) (2,10) (4,12) (6,14), one member below v - t, that is max_k min(p_k, p_k+8) < v - t; a bright corner needs
// min_k max(p_k, p_k+8) > v + t.  qd / qb: per lane a value whose sign bit is set exactly when the pixel survives as a
// dark / bright candidate at threshold t (thr holds t in both lanes).
template <int J0>
__device__ __forceinline__ void fast_pretest(const uint32_t (&m3)[3], const uint32_t (&m2)[3], const uint32_t (&c)[3],
                                             const uint32_t (&p2)[3], const uint32_t (&p3)[3], const s16x2 thr, uint32_t& qd, uint32_t& qb)
{
    const s16x2 v = pair16<3 + J0>(c[0], c[1], c[2]);
    const s16x2 r0 = pair16<3 + J0>(p3[0], p3[1], p3[2]);    // ( 0,+3)
    const s16x2 r8 = pair16<3 + J0>(m3[0], m3[1], m3[2]);    // ( 0,-3)
    const s16x2 r4 = pair16<6 + J0>(c[0], c[1], c[2]);       // (+3, 0)
    const s16x2 r12 = pair16<0 + J0>(c[0], c[1], c[2]);      // (-3, 0)
    const s16x2 r2 = pair16<5 + J0>(p2[0], p2[1], p2[2]);    // (+2,+2)
    const s16x2 r14 = pair16<1 + J0>(p2[0], p2[1], p2[2]);   // (-2,+2)
    const s16x2 r6 = pair16<5 + J0>(m2[0], m2[1], m2[2]);    // (+2,-2)
    const s16x2 r10 = pair16<1 + J0>(m2[0], m2[1], m2[2]);   // (-2,-2)
    const s16x2 hi = pmax(pmax(pmin(r0, r8), pmin(r4, r12)), pmax(pmin(r2, r10), pmin(r6, r14)));
    const s16x2 lo = pmin(pmin(pmax(r0, r8), pmax(r4, r12)), pmin(pmax(r2, r10), pmax(r6, r14)));
    qd = as_u32(hi - (v - thr));
    qb = as_u32((v + thr) - lo);
}

// number of set bits of m below the calling lane
__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// divisions by the number of four-pixel groups in a cell row, as tables (the divisor is uniform: scalar loads)
struct GroupsPerRow { int rows[16]; int magic[16]; };
constexpr GroupsPerRow make_groups_per_row()
{
    GroupsPerRow t{};
    for (int g = 1; g < 16; ++g) { t.rows[g] = 64 / g; t.magic[g] = (65536 + g - 1) / g; }
    return t;
}
__constant__ const GroupsPerRow kGroupsPerRow = make_groups_per_row();

// Diagnostic build (-DORBX_FAST_STAMPS, tools/fast_stamps.py): wave cycles between the phases of k_fast_cells, one row of
// eight sums per wavefront; a stamp itself costs about 560 cycles.
#ifdef ORBX_FAST_STAMPS
constexpr int kStampWaves = 256 * 1024;
__device__ uint32_t g_fast_stamps[kStampWaves][8];
#define FSTAMP(i) do { wave_sync(); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_fast_stamps[stamp_row_][i] += (uint32_t)(t_ - t_prev_); t_prev_ = t_; } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
#ifndef ORBX_FAST_WPE_MIN
#define ORBX_FAST_WPE_MIN 7
#define ORBX_FAST_WPE_MAX 8
#endif
// What one wavefront needs to know about its cell, in launch order (read with scalar loads: one round trip before the
// tile can be fetched, where cell list -> cell record -> level record took three).
struct FastCell {
    uint32_t src;        // byte offset of the cell sub-image's first pixel inside a frame's pyramid
    uint32_t stride;     // bytes per row of the level
    uint16_t cw, ch;     // detection area
    uint16_t offx, offy; // j*wCell, i*hCell
    uint32_t out;        // first candidate slot of the cell inside a frame's slot array
    uint32_t cap;        // slots reserved
    uint32_t cell;       // index into the per-frame count array
    uint32_t pad;
};

// One wavefront per cell, WPB cells per workgroup; no workgroup barrier anywhere.  Seven wavefronts per SIMD are asked
// for, not eight: at 64 registers the score pass spills (the spill traffic cost the pipeline 2 %), and between five and
// eight wavefronts per SIMD the kernel's time does not change -- it is bound by vector issue.
//   tile load -> per threshold attempt { pass 1: packed pretest of 256 pixels per round, survivors kept as bits ->
//   queued -> pass 2: exact score of the queued survivors -> pass 3: strict local maxima among the corners }
//   -> every maximum finds its own rank in raster order and writes its key.
// TS/TR: tile row stride (bytes) and rows, SS/SR: score tile stride and rows, MR: rows of the maxima masks,
// QCAP: entries of the survivor / corner list.  The LDS per workgroup decides how many wavefronts a CU holds, and the
// kernel lives on that (its phases wait on LDS and memory round trips): the list is sized for the usual cell, not for
// the worst one.  A cell with more survivors than the list holds takes the dense path: its rounds are queued and
// scored one at a time, no corner list is kept, and passes 3 and the emission walk all pixels of the cell instead.
template <int TS, int TR, int SS, int SR, int MR, int QCAP, int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(TR <= kSmallCellDim + 6 ? ORBX_FAST_WPE_MIN : 3, TR <= kSmallCellDim + 6 ? ORBX_FAST_WPE_MAX : 8))) void k_fast_cells(const uint8_t* __restrict__ pyr, unsigned long long pyr_bytes,
                                                    const FastCell* __restrict__ fcells, int nlist,
                                                    uint32_t* __restrict__ cand, int cand_total,
                                                    uint16_t* __restrict__ cellcnt, int ncells,
                                                    int iniTh, int minTh, int n_frames)
{
    static_assert(QCAP >= 256 && QCAP % 8 == 0 && (SR * SS) % 16 == 0 && MR <= 64, "list holds at least one round");
    __shared__ __align__(16) uint8_t s_tile[WPB][TR * TS];
    __shared__ __align__(16) uint8_t s_score[WPB][SR * SS];
    __shared__ unsigned long long s_mask[WPB][MR];
    __shared__ uint16_t s_queue[WPB][QCAP];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // everything derived from it stays scalar
    int frame; uint32_t blk;
    if (!frame_chunk(n_frames, frame, blk)) return;
    if ((int)blk * WPB + wave >= nlist) return;
    const FastCell c = fcells[(int)blk * WPB + wave];
#ifdef ORBX_FAST_STAMPS
    const uint32_t stamp_row_ = ((uint32_t)frame * (uint32_t)nlist + blk) % kStampWaves;
    if (lane < 8) g_fast_stamps[stamp_row_][lane] = 0;
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
    const uint8_t* src = pyr + (unsigned long long)frame * pyr_bytes + c.src;
    uint8_t* tile = s_tile[wave];
    uint8_t* score = s_score[wave];
    unsigned long long* rowmask = s_mask[wave];
    uint16_t* queue = s_queue[wave];
    const int cw = c.cw, ch = c.ch;
    const int tw = cw + 6, th = ch + 6;
    {
        // 16 bytes per lane, TS/16 lanes per tile row, RPI rows per turn: at most NIT turns, written out so that every load
        // is issued before the first store and the addresses are a uniform base plus a 32-bit lane offset (left to itself
        // the compiler unrolls the loop eightfold around 64-bit multiply-adds)
        constexpr int LPR = TS / 16, RPI = 64 / LPR, NIT = (TR + RPI - 1) / RPI;
        const int r0 = (int)(__umul24((uint32_t)lane, (256u + LPR - 1) / LPR) >> 8);   // lane / LPR (exact for lane < 64, LPR 3 or 5)
        int j;                                                   // lane % LPR (as an instruction: the compiler picks the 64-bit multiply-add)
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(j) : "v"(r0), "n"(-LPR), "v"(lane));
        static_assert(LPR == 3 || LPR == 5, "the reciprocal above is checked for these");
        const bool act = lane < LPR * RPI && 16 * j < tw;
        const uint32_t goff = __umul24((uint32_t)r0, c.stride) + 16u * (uint32_t)j, gstep0 = (uint32_t)RPI * c.stride;
        uint32_t gstep = gstep0, gstep2 = 2u * gstep0;
        asm volatile("" : "+s"(gstep), "+s"(gstep2));      // (kept as scalars: folded back into the offsets they become 64-bit multiply-adds)
        const uint32_t loff = (uint32_t)(r0 * TS + 16 * j);
        u128_unaligned t[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (act && r0 + it * RPI < th) t[it] = *reinterpret_cast<const u128_unaligned*>(src + (goff + (it == 0 ? 0u : it == 1 ? gstep : it == 2 ? gstep2 : (uint32_t)it * gstep)));
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (act && r0 + it * RPI < th) *reinterpret_cast<uint4*>(tile + (loff + (uint32_t)(it * RPI * TS))) = make_uint4(t[it].x, t[it].y, t[it].z, t[it].w);
    }
    FSTAMP(0);
    // pretest lanes: groups of four pixels, gpr per row (<= 15), as many whole rows per round as 64 lanes hold
    const int gpr = (cw + 3) >> 2;
    const int rows_per_round = kGroupsPerRow.rows[gpr];              // 64 / gpr
    const int lrow = (int)(__umul24((uint32_t)lane, (uint32_t)kGroupsPerRow.magic[gpr]) >> 16);   // lane / gpr (exact for lane < 64, gpr <= 15)
    const int g = lane - lrow * gpr;
    const int x = 4 * g;
    const int npx = lrow < rows_per_round ? min(cw - x, 4) : 0;            // pixels of the lane inside the cell
    const int qbase = lrow * 64 + x;                                       // list entry of the lane's first pixel in round 0
    // Survivor bits of a lane: one bit per (pixel of its four, round).  Cells of the small variant take at most eight
    // rounds, so a byte per pixel holds them (pixel k in byte k); larger cells take up to sixteen, a half-word per pixel.
    // Round r of R sits at bit kRoundBits - R + r of its field (the fields are shifted right once per round).
    constexpr bool kSmall = TR <= kSmallCellDim + 6;
    constexpr int kRoundBits = kSmall ? 8 : 16;
    static_assert(!kSmall || (kSmallCellDim + (64 / ((kSmallCellDim + 3) / 4)) - 1) / (64 / ((kSmallCellDim + 3) / 4)) < 8, "rounds of a small cell must leave the top bit of a byte free");
    static_assert(kSmall || (kMaxCellDim + 3) / (64 / ((kMaxCellDim + 3) / 4)) < 16, "rounds of a large cell must leave the top bit of a half-word free");
    typedef typename std::conditional<kSmall, uint32_t, unsigned long long>::type Bits;
    const Bits colbits = kSmall ? (Bits)(npx >= 4 ? 0xFFFFFFFFu : (1u << (8 * npx)) - 1u)
                                : (npx >= 4 ? ~(Bits)0 : ((Bits)1 << (16 * npx)) - (Bits)1);

    // exact two-sided scores of queue[0..n) (dense path): corners (score >= T) go to the score tile
    auto score_queued = [&](int n, int T) {
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            if (e < n) {
                const int i = queue[e];
                const int yy = i >> 6, xx = i & 63;
                const int sc = fast_score<TS>(&tile[(yy + 3) * TS + xx + 3]);
                if (sc >= T) score[(yy + 1) * SS + xx + 1] = (uint8_t)sc;
            }
        }
    };
    // exact value of the queued survivors, dark list queue[0 .. nd) and bright list queue[nd .. nd + nb), each on its own
    // side: corners (value > T) go to the score tile with score value - 1 and are compacted to queue[0 ..) (a scored entry
    // is written at or below the slot its lane just read, so no waiting entry is overwritten); returns their number.
    // The two lists are walked as one, 64 entries a turn, so that only one turn is partly filled; the turn that holds
    // the end of the dark list and the start of the bright one takes the dark form with the bright lanes' pixels
    // complemented.
    auto score_lists = [&](int nd, int nb, int T) -> int {
        const int n = nd + nb;
        int kept = 0;
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            bool keep = false;
            int i = 0;
            if (e < n) {
                i = queue[e];
                const int yy = i >> 6, xx = i & 63;
                const uint8_t* t = &tile[(yy + 3) * TS + xx + 3];
                int val;
                if (min(e0 + 64, n) <= nd) val = fast_side<TS, true>(t);
                else if (e0 >= nd) val = fast_side<TS, false>(t);
                else val = fast_side<TS, true>(t, e >= nd ? 0xFF : 0);
                keep = val > T;
                // (a pixel queued on both sides is a corner on one at most: two arcs of nine do not fit the ring of sixteen)
                if (keep) score[(yy + 1) * SS + xx + 1] = (uint8_t)(val - 1);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) queue[kept + lanes_below(m)] = (uint16_t)i;
            kept += __popcll(m);
        }
        return kept;
    };
    // pixel (xx, yy) of the cell is a corner with a score above its eight neighbours inside the cell
    auto strict_max = [&](int yy, int xx) -> bool {
        const uint8_t* sp = &score[(yy + 1) * SS + xx + 1];
        const int sc = sp[0];
        return sc > sp[-1] && sc > sp[1] && sc > sp[-SS - 1] && sc > sp[-SS] && sc > sp[-SS + 1] &&
               sc > sp[SS - 1] && sc > sp[SS] && sc > sp[SS + 1];
    };

    // cv::FAST(cell, T, nonmax) at T = iniThFAST; only if that returns nothing, at T = minThFAST (:808-815).  Pixels
    // scoring below T play no part at threshold T (a kept pixel beats them anyway), so each attempt scores only
    // its own corners; nearly every cell is done after the first.
    unsigned long long mask = 0;
    int nsq = 0;                                                     // corners of the last attempt: queue[0..nsq)
    bool dense = false;                                              // ... unless the list could not hold the survivors
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int T = max(attempt == 0 ? iniTh : minTh, 1);          // score >= T  <=>  corner at threshold T (T >= 1)
        for (int i = lane; i < ((ch + 2) * SS + 15) >> 4; i += 64) reinterpret_cast<uint4*>(score)[i] = make_uint4(0, 0, 0, 0);
        if (lane < MR) rowmask[lane] = 0;
        wave_sync();
        FSTAMP(1);
        const s16x2 thr = {(short)T, (short)T};
        // pass 1: every lane keeps the survivors of its four pixels per round in its own bit sets, dark and bright
        // candidates apart: nothing is queued while the rounds run
        Bits dbits = 0, bbits = 0;
        uint32_t ad01 = 0, ad23 = 0, ab01 = 0, ab23 = 0;             // large cells: pixels 0,1 / 2,3 as half-words
        int R = 0;
        for (int y0 = 0; y0 < ch; y0 += rows_per_round, ++R) {
            const int y = y0 + lrow;
            if constexpr (kSmall) { dbits >>= 1; bbits >>= 1; }
            else { ad01 >>= 1; ad23 >>= 1; ab01 >>= 1; ab23 >>= 1; }
            if (y < ch && npx > 0) {
                const uint32_t* row = reinterpret_cast<const uint32_t*>(&tile[y * TS]) + g;   // tile row y = image row y-3
                constexpr int RS = TS / 4;
                const uint32_t m3[3] = {row[0], row[1], row[2]};
                const uint32_t m2[3] = {row[RS], row[RS + 1], row[RS + 2]};
                const uint32_t cc[3] = {row[3 * RS], row[3 * RS + 1], row[3 * RS + 2]};
                const uint32_t p2[3] = {row[5 * RS], row[5 * RS + 1], row[5 * RS + 2]};
                const uint32_t p3[3] = {row[6 * RS], row[6 * RS + 1], row[6 * RS + 2]};
                uint32_t qd01, qb01, qd23, qb23;
                fast_pretest<0>(m3, m2, cc, p2, p3, thr, qd01, qb01);
                fast_pretest<2>(m3, m2, cc, p2, p3, thr, qd23, qb23);
                if constexpr (kSmall) {
                    // the four sign bytes side by side, pixel k in byte k
                    dbits |= __builtin_amdgcn_perm(qd23, qd01, 0x07050301u) & 0x80808080u;
                    bbits |= __builtin_amdgcn_perm(qb23, qb01, 0x07050301u) & 0x80808080u;
                } else {
                    ad01 |= qd01 & 0x80008000u; ad23 |= qd23 & 0x80008000u;
                    ab01 |= qb01 & 0x80008000u; ab23 |= qb23 & 0x80008000u;
                }
            }
        }
        if constexpr (!kSmall) {
            dbits = (Bits)ad01 | ((Bits)ad23 << 32);
            bbits = (Bits)ab01 | ((Bits)ab23 << 32);
        }
        dbits &= colbits; bbits &= colbits;
        FSTAMP(2);
        // the survivors in `bits` go to the queue from entry `off` on (the order inside the queue does not matter)
        const int rstep = rows_per_round * 64;
        const int qb0 = qbase - (kRoundBits - R) * rstep;
        auto lowest = [](Bits bits) -> int { return sizeof(Bits) == 4 ? __ffs((int)bits) - 1 : __ffsll((long long)bits) - 1; };
        // (the multiply-add spelled out: the compiler knows both factors are small, drops the 24-bit form and then picks
        // the 64-bit multiply-add, which issues at a quarter of the rate, for the innermost loop of the queueing)
        auto entry_of = [&](int bpos) -> uint16_t {
            uint32_t e;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(e) : "v"((uint32_t)(bpos & (kRoundBits - 1))), "s"((uint32_t)rstep), "v"((uint32_t)(qb0 + (bpos >> (kSmall ? 3 : 4)))));
            return (uint16_t)e;
        };
        auto enqueue = [&](Bits bits, int off) {
            while (bits) {
                queue[off++] = entry_of(lowest(bits));
                bits &= bits - 1;
            }
        };
        // one prefix sum for both lists: dark counts in the low half, bright counts in the high half
        const int mine_d = __popcll((unsigned long long)dbits), mine_b = __popcll((unsigned long long)bbits);
        const int incl = ot_wave_scan_incl(mine_d | (mine_b << 16));
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        const int nd = tot & 0xFFFF, nb = tot >> 16;
        dense = nd + nb > QCAP;
        if (!dense) {
            enqueue(dbits, (incl & 0xFFFF) - mine_d);
            enqueue(bbits, nd + (incl >> 16) - mine_b);
            wave_sync();
            FSTAMP(3);
            // pass 2: exact value of the queued survivors, each on its list's side; the corners end up in queue[0..nsq)
            nsq = score_lists(nd, nb, T);
            wave_sync();
            FSTAMP(4);
            // pass 3: strict local maxima inside the cell among the corners, one lane per corner; row masks of the maxima
            for (int e = lane; e < nsq; e += 64) {
                const int i = queue[e];
                const int yy = i >> 6, xx = i & 63;
                if (strict_max(yy, xx)) atomicOr(&rowmask[yy], 1ull << xx);
            }
        } else {
            // more survivors than the list holds: one round (at most 256 of them) at a time, both sides, scores only
            const Bits ubits = dbits | bbits;
            Bits rsel = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) rsel |= (Bits)1 << (kRoundBits * k);
            for (int r = 0; r < R; ++r) {
                const Bits rb = ubits & (rsel << (kRoundBits - R + r));
                const int rmine = __popcll((unsigned long long)rb);
                const int rincl = ot_wave_scan_incl(rmine);
                const int n = __builtin_amdgcn_readlane(rincl, 63);
                enqueue(rb, rincl - rmine);
                wave_sync();
                score_queued(n, T);
                wave_sync();
            }
            // (row by row, a lane per column -- cells are under 64 wide: a flat index would cost every cell, dense or not,
            // the set-up of a division by the cell width, which the compiler moves to the top of the kernel)
            for (int yy = 0; yy < ch; ++yy)
                if (lane < cw && score[(yy + 1) * SS + lane + 1] && strict_max(yy, lane)) atomicOr(&rowmask[yy], 1ull << lane);
        }
        wave_sync();
        FSTAMP(5);
        mask = lane < MR ? rowmask[lane] : 0ull;                      // lane r owns row r (rows >= ch are empty)
        if (__any(mask != 0ull) || max(minTh, 1) >= T) break;         // a second attempt at a threshold not below T finds nothing new
        wave_sync();
    }
    // raster-ordered emission: row r's maxima start at the number of maxima in the rows above it (lane r owns row r for
    // the prefix sum) and follow each other from left to right -- every lane walks the set bits of its own row mask, so
    // the loop takes as many turns as the fullest row has maxima (two or three), with no pass over the corner list
    const int cnt = __popcll(mask);
    const int inc = ot_wave_scan_incl(cnt);
    const int total = __builtin_amdgcn_readlane(inc, 63);
    if (total) {
        uint32_t* out = cand + (unsigned long long)frame * cand_total + c.out;
        const uint32_t keybase = (((uint32_t)(3 + c.offx) << 20) | ((uint32_t)(3 + c.offy) << 8)) + ((uint32_t)lane << 8);
        const uint8_t* srow = &score[(lane + 1) * SS + 1];
        int pos = inc - cnt;
        unsigned long long m = mask;
        while (m) {
            const int xx = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            if (pos < (int)c.cap) out[pos] = keybase + ((uint32_t)xx << 20) + (uint32_t)srow[xx];
            ++pos;
        }
    }
    if (lane == 0) cellcnt[(unsigned long long)frame * ncells + c.cell] = (uint16_t)min(total, (int)c.cap);
    FSTAMP(6);
    FSTAMP(7);
}

// ---------------------------------------------------------------------------------------------
// Quadtree selection: one workgroup per (level, frame).  Gathers the per-cell candidate slots into a
// dense key array (cell-row-major, raster inside a cell: the reference's vToDistributeKeys order),
// then runs the level-synchronous DistributeOctTree of octree_algo.h.
// ---------------------------------------------------------------------------------------------
constexpr int kOctThreads = 1024;    // launch bound; the launch uses orbx_extractor::oct_threads

// LDS layout: the node tables of OtShared (capacity NC nodes), then room for KL keys (4 B) and node ids (2 B)
__device__ size_t ot_carve(OtShared& s, unsigned char* mem, int NC, int nthreads, int TC, int XY)
{
    size_t o = 0;
    s.cur = reinterpret_cast<OtNode*>(mem + o); o += sizeof(OtNode) * NC;
    s.nxt = reinterpret_cast<OtNode*>(mem + o); o += sizeof(OtNode) * NC;
    s.cc = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * 4 * NC;
    s.candPos = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * NC;
    s.pi = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * NC;
    s.cByPi = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * NC;
    s.keep = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * NC;
    s.posToCand = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * NC;
    o = (o + 7) & ~(size_t)7;
    s.cc2 = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * 2 * NC;
    s.scal = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * 16;
    s.scanTmp = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * (nthreads / 64 + 2);
    s.tcnt = reinterpret_cast<int32_t*>(mem + o); o += sizeof(int32_t) * TC;
    s.tpos = reinterpret_cast<uint16_t*>(mem + o); o += sizeof(uint16_t) * TC;
    s.ent0 = reinterpret_cast<uint16_t*>(mem + o); o += TC ? sizeof(uint16_t) * NC : 0;
    s.ent1 = reinterpret_cast<uint16_t*>(mem + o); o += TC ? sizeof(uint16_t) * NC : 0;
    s.tcap = TC;
    s.xs = reinterpret_cast<uint16_t*>(mem + o); o += sizeof(uint16_t) * XY;
    s.ys = s.xs;                    // set per level: the rows follow the level's columns
    s.xycap = XY;
    return (o + 15) & ~(size_t)15;
}
size_t ot_lds_bytes(int NC, int nthreads, int KL, int TC, int XY)
{
    size_t o = sizeof(OtNode) * 2 * NC + sizeof(int32_t) * (4 + 5) * NC + 8 + sizeof(int32_t) * 2 * NC + sizeof(int32_t) * 16 + sizeof(int32_t) * (nthreads / 64 + 2)
             + (size_t)TC * 6 + (TC ? sizeof(uint16_t) * 2 * NC : 0) + sizeof(uint16_t) * XY;
    o = (o + 15) & ~(size_t)15;
    return o + (size_t)KL * 6 + 16;
}
// entries of the quadtree's count tables that a level with quota N and nIni roots can put to use (kMaxTableCells at most)
constexpr int kMaxTableCells = 4096;
constexpr size_t kOctLdsLimit = 158 * 1024;          // of the 160 KB a workgroup can have
int ot_table_cells(int N, int nIni)
{
    return ot_table_size(nIni, ot_table_depth(N, nIni, kMaxTableCells));
}

// gathers the per-cell candidate slots of a level into the dense key array K (cell-row-major, raster inside a cell: the
// reference's vToDistributeKeys order) and runs the quadtree on it
template <int LOOK>
__device__ __attribute__((always_inline)) int octree_level(OtShared& s, uint32_t* K, uint16_t* NO, int nk, const DevLevel& L,
                                                            const CellRec* __restrict__ cells, const uint16_t* __restrict__ cnt,
                                                            const uint32_t* __restrict__ slots, int NCmax, uint32_t* __restrict__ out)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    int base = 0;
    for (int c0 = 0; c0 < L.cell_count; c0 += NCmax) {
        const int m = min(NCmax, L.cell_count - c0);
        for (int i = tid; i < m; i += NT) s.keep[i] = cnt[c0 + i];
        __syncthreads();
        const int tot = ot_block_scan(s.keep, m, s.scanTmp);
        for (int i = tid; i < m; i += NT) {
            const CellRec& cr = cells[L.cell_first + c0 + i];
            const int n = cnt[c0 + i];
            const int o = base + s.keep[i];
            for (int j = 0; j < n; ++j) K[o + j] = slots[cr.slot + j];
        }
        __syncthreads();
        base += tot;
    }
    __threadfence_block();
    __syncthreads();
    OtMemKeysT<LOOK> ks{K, NO, nk};
    return ot_distribute(s, ks, K, L.maxBorderX - kMinBorder, L.maxBorderY - kMinBorder, L.nIni, L.hX, L.quota, out, L.kp_cap);
}

// The same selection without the dense key array, for a level with enough candidates for the count tables: the steps
// then never touch a key, so the keys are only read twice, where the FAST cells left them -- once to count the cells
// of the tables' deepest level, once to let every key compete for its node (found through the leaf table of
// ot_distribute_tables: one look-up instead of a walk down the tables).  A key's rank in input order (cell-row-major,
// raster inside a cell, the order the gather above lays the keys out in) is (cell << kDirectCellBits | index inside the
// cell).  One thread per FAST cell, kDirectBatch keys loaded at a time; runs of keys with the same target act once.
// Returns the length of the list, or -1 when the steps have to leave the tables: the caller then takes the dense form.
constexpr int kDirectCellBits = 10;
#ifndef ORBX_DIRECT_BATCH
#define ORBX_DIRECT_BATCH 8
#endif
constexpr int kDirectBatch = ORBX_DIRECT_BATCH;
__device__ __attribute__((always_inline)) int octree_level_direct(OtShared& s, const DevLevel& L, int D, const CellRec* __restrict__ cells,
                                                                   const uint16_t* __restrict__ cnt, const uint32_t* __restrict__ slots,
                                                                   uint32_t* __restrict__ out)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    const int W = L.maxBorderX - kMinBorder, H = L.maxBorderY - kMinBorder;
    s.ys = s.xs + W;
    const int total = ot_table_size(L.nIni, D), offD = ot_table_off(L.nIni, D);
    for (int i = tid; i < total; i += NT) s.tcnt[i] = 0;
    ot_build_axis_tables(s, W, H, L.nIni, L.hX, D);
    __syncthreads();
    uint16_t* leaf = reinterpret_cast<uint16_t*>(s.cc2);      // free until the steps' last use of it, and long enough: fewer than 4 N cells at depth D
    unsigned* best = reinterpret_cast<unsigned*>(s.keep);
    auto key_pass = [&](auto assign) {
        constexpr bool ASSIGN = decltype(assign)::value;
        int* deep = s.tcnt + offD;
        for (int ci = tid; ci < L.cell_count; ci += NT) {
            const int n = cnt[ci];
            if (n == 0) continue;
            const uint32_t* ks = slots + cells[L.cell_first + ci].slot;
            int cur = -1, run = 0; unsigned v = 0;
            auto flush = [&]() {
                if (cur < 0) return;
                if (!ASSIGN) atomicAdd(&deep[cur], run);
                else if (v > best[cur]) atomicMax(&best[cur], v);
            };
            for (int j0 = 0; j0 < n; j0 += kDirectBatch) {
                uint32_t key[kDirectBatch];
#pragma unroll
                for (int u = 0; u < kDirectBatch; ++u) key[u] = j0 + u < n ? ks[j0 + u] : 0u;
#pragma unroll
                for (int u = 0; u < kDirectBatch; ++u) {
                    if (j0 + u >= n) break;
                    const int idx = ot_key_cell(s, key[u]);
                    const int tgt = ASSIGN ? (int)leaf[idx] : idx;
                    if (tgt != cur) { flush(); cur = tgt; run = 0; v = 0; }
                    ++run;
                    v = max(v, ((key[u] & 0xFFu) << 24) | (0xFFFFFFu - (((unsigned)ci << kDirectCellBits) | (unsigned)(j0 + u))));
                }
            }
            flush();
        }
    };
    key_pass(std::false_type{});
    __syncthreads();
    const int n = ot_distribute_tables(s, H, L.nIni, L.hX, L.quota, D, leaf);
    if (n < 0) return -1;
    for (int p = tid; p < n; p += NT) best[p] = 0u;
    __syncthreads();
    key_pass(std::true_type{});
    __syncthreads();
    for (int p = tid; p < n && p < L.kp_cap; p += NT) {
        const unsigned c = 0xFFFFFFu - (best[p] & 0xFFFFFFu);
        out[p] = slots[cells[L.cell_first + (c >> kDirectCellBits)].slot + (c & ((1u << kDirectCellBits) - 1u))];
    }
    return n;
}

template <bool kTablesInDeviceMemory>
__global__ __launch_bounds__(kOctThreads) void k_octree(const DevLevel* __restrict__ lv, const CellRec* __restrict__ cells,
                                                        const uint32_t* __restrict__ cand, int cand_total,
                                                        const uint16_t* __restrict__ cellcnt, int ncells,
                                                        uint32_t* __restrict__ keys, uint16_t* __restrict__ nodeOf, int key_total,
                                                        uint32_t* __restrict__ lvlkp, int kp_total,
                                                        int32_t* __restrict__ lvlcnt, int nlevels, int NCmax, int KL, int TC, int XY,
                                                        unsigned char* __restrict__ gmem, unsigned long long gstride)
{
    extern __shared__ __align__(16) unsigned char ot_lds[];
    __shared__ int s_nk;
    const int tid = threadIdx.x;
    const int level = blockIdx.y, frame = blockIdx.x;
    const DevLevel& L = lv[level];
    OtShared s;
    const int NT = blockDim.x;
    // the node tables live in LDS; a quota too large for it (thousands of keypoints in one level) takes a slice of device
    // memory instead (KL == 0): the waves of a workgroup share one CU's L1, so barriers order it the same way.  A template
    // parameter rather than a run-time choice, so that the LDS form keeps its DS instructions.
    unsigned char* ot_mem = kTablesInDeviceMemory ? gmem + ((unsigned long long)frame * nlevels + level) * gstride : ot_lds;
    const size_t keyOff = ot_carve(s, ot_mem, NCmax, kOctThreads, TC, XY);

    const uint16_t* cnt = cellcnt + (unsigned long long)frame * ncells + L.cell_first;
    const uint32_t* slots = cand + (unsigned long long)frame * cand_total + L.cand_offset;
    // number of candidates of this level: decides where the key arrays live
    int part = 0;
    for (int i = tid; i < L.cell_count; i += NT) part += cnt[i];
    part = wave_sum_i32(part);
    if (tid == 0) s_nk = 0;
    __syncthreads();
    if ((tid & 63) == 0) atomicAdd(&s_nk, part);
    __syncthreads();
    const int nk = s_nk;
    // no candidates (always so for a level too small to hold a FAST cell): DistributeOctTree returns nothing, whatever nIni is
    if (nk == 0) {
        if (tid == 0) lvlcnt[frame * nlevels + level] = 0;
        return;
    }
    // two copies of the same code, one per home of the key arrays: with the address space known at compile time the LDS
    // copy gets DS instructions (a pointer chosen at run time would make every key access a FLAT one)
    int n = -1;
    // (keys that fit the LDS arrays are faster in the dense form: 640x480 87 us per 256 frames against 96)
    if (L.direct_ok && nk > KL && nk >= kOtTableMinKeysPerThread * NT && (L.maxBorderX - kMinBorder) + (L.maxBorderY - kMinBorder) <= XY) {
        const int D = ot_table_depth(L.quota, L.nIni, TC);
        if (D > 0) n = octree_level_direct(s, L, D, cells, cnt, slots, lvlkp + (unsigned long long)frame * kp_total + L.kp_offset);
        __syncthreads();
    }
    if (n >= 0) {}
    else if (nk <= KL)
        n = octree_level<4>(s, reinterpret_cast<uint32_t*>(ot_mem + keyOff), reinterpret_cast<uint16_t*>(ot_mem + keyOff + (size_t)KL * 4), nk,
                         L, cells, cnt, slots, NCmax, lvlkp + (unsigned long long)frame * kp_total + L.kp_offset);
    else
        n = octree_level<4>(s, keys + (unsigned long long)frame * key_total + L.key_offset,
                         nodeOf + (unsigned long long)frame * key_total + L.key_offset, nk,
                         L, cells, cnt, slots, NCmax, lvlkp + (unsigned long long)frame * kp_total + L.kp_offset);
    if (tid == 0) lvlcnt[frame * nlevels + level] = min(n, L.kp_cap);
}

// standalone form for parity tests: keys already dense
__global__ __launch_bounds__(kOctThreads) void k_octree_single(const uint32_t* __restrict__ keys, uint16_t* __restrict__ nodeOf, int nk,
                                                               int W, int H, int nIni, float hX, int N,
                                                               uint32_t* __restrict__ out, int outCap, int32_t* __restrict__ nOut, int NC,
                                                               int KL, int TC, int XY, unsigned char* __restrict__ gmem)
{
    extern __shared__ __align__(16) unsigned char ot_lds[];
    OtShared s;
    unsigned char* ot_mem = gmem ? gmem : ot_lds;
    const size_t keyOff = ot_carve(s, ot_mem, NC, kOctThreads, TC, XY);
    const uint32_t* K = keys; uint16_t* NO = nodeOf;
    if (nk <= KL) {                     // KL == 0 keeps the keys in device memory
        uint32_t* kl = reinterpret_cast<uint32_t*>(ot_mem + keyOff);
        for (int k = threadIdx.x; k < nk; k += blockDim.x) kl[k] = keys[k];
        K = kl;
        NO = reinterpret_cast<uint16_t*>(ot_mem + keyOff + (size_t)KL * 4);
        __syncthreads();
    }
    OtMemKeys ks{K, NO, nk};
    const int n = ot_distribute(s, ks, K, W, H, nIni, hX, N, out, outCap);
    if (threadIdx.x == 0) *nOut = n;
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2 of every pyramid level (src/ORBextractor.cc:1084-1085): Q8 taps
// 18,34,49,55,49,34,18 in both directions, one rounding (sum + 32768) >> 16 (cv::GaussianBlur's 8-bit
// fixed-point path; with rounding mode 1 ties go to even for the columns an x86 SIMD row loop covers).
// The reference blurs a clone of the level, so the filter's REFLECT_101 border is the pyramid's own frame.
//
// One thread owns 4 adjacent columns and slides down 16 rows: per input row three dwords are loaded, the
// horizontal pass is two v_dot4_u32_u8 per pixel on byte-aligned windows (v_alignbyte_b32), horizontal sums
// of two consecutive rows are packed into one dword per pixel, and the vertical pass is four
// v_dot2_u32_u16 per output pixel over the last four row pairs.  No LDS.
// ---------------------------------------------------------------------------------------------
constexpr int kBlurRows = 16;
struct __attribute__((packed, aligned(4))) u96 { uint32_t a, b, c; };

// Output layout: every blurred level is tiled, 8 rows x 16 bytes per 128-byte tile (tile (y >> 3, x >> 4), row y & 7 at byte
// 16 * (y & 7)), tpr tiles per tile row: the 37x37 window of a keypoint then spans ~19 cache lines instead of ~51.  A
// wavefront's 64 threads are 16 units of 16 pixels x 16 rows; the results pass through LDS so that every store of the
// wavefront writes eight complete lines (8 lanes x 16 bytes each).
constexpr int kBlurUnitBytes = 16 * kBlurRows + 16;   // one unit in LDS, padded against bank conflicts
template <int MODE>
__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blr, unsigned long long pyr_bytes,
                                              unsigned long long blur_bytes, const DevLevel* __restrict__ lv,
                                              const uint32_t* __restrict__ chunks, int n_frames)
{
    __shared__ __align__(16) uint8_t s_stage[4][16 * kBlurUnitBytes];
    int frame; uint32_t chunk;
    if (!frame_chunk(n_frames, frame, chunk)) return;
    // (level << 16 | workgroup inside the level) of every workgroup of a frame: one scalar load, where a division by the
    // workgroups per level cost every thread some twenty-five vector instructions, and no empty workgroups behind the
    // small levels
    const uint32_t entry = chunks[chunk];
    const uint32_t level = entry >> 16, blk = entry & 0xFFFFu;
    const DevLevel& L = lv[level];
    const uint32_t upr = (uint32_t)L.upr;                              // units (4 threads, 16 pixels) per strip row
    const uint32_t strips = (uint32_t)((L.h + kBlurRows - 1) / kBlurRows);
    const uint32_t nunits = upr * strips;
    if (blk * 64u >= nunits) return;                                   // the whole workgroup is past the level
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i = blk * 256 + threadIdx.x;                        // thread = (unit i >> 2, quarter i & 3)
    uint8_t* stage = s_stage[wave];
    uint32_t strip, uc;                                                // of the thread's unit, also for the units past the level
    split_index(i >> 2, upr, L.upr_magic, strip, uc);
    const uint32_t my_unit = (strip << 16) | uc;
    if ((i >> 2) < nunits) {
        const int y0 = (int)strip * kBlurRows, x = (int)(16 * uc + 4 * (i & 3));
        // wavefront-uniform base (three rows up and four bytes left of the ROI origin) plus a 32-bit lane offset
        const uint32_t stride = (uint32_t)L.stride;
        const uint8_t* src = pyr + (unsigned long long)frame * pyr_bytes + L.roi - 3 * stride - 4;
        const uint32_t vo = __umul24((uint32_t)y0, stride) + (uint32_t)x;
        const bool even_ties = MODE != 0 && x < (L.w & ~3);
        uint32_t* out = reinterpret_cast<uint32_t*>(stage + ((lane >> 2) * kBlurUnitBytes + 4 * (lane & 3)));   // + 16 bytes per row
        constexpr uint32_t T0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), T1 = 49u | (34u << 8) | (18u << 16);
        // horizontal taps 18 34 49 55 49 34 18 centred on pixel x+k, cut into the words that meet t.a / t.b / t.c
        constexpr uint32_t A0 = (18u << 8) | (34u << 16) | (49u << 24), B0 = 55u | (49u << 8) | (34u << 16) | (18u << 24);
        constexpr uint32_t A1 = (18u << 16) | (34u << 24), B1 = 49u | (55u << 8) | (49u << 16) | (34u << 24), C1 = 18u;
        constexpr uint32_t A2 = 18u << 24, B2 = 34u | (49u << 8) | (55u << 16) | (49u << 24), C2 = 34u | (18u << 8);
        constexpr uint32_t E0 = 18u | (34u << 16), E1 = 49u | (55u << 16), E2 = 49u | (34u << 16), E3 = 18u;
        constexpr uint32_t O0 = 18u << 16, O1 = 34u | (49u << 16), O2 = 55u | (49u << 16), O3 = 34u | (18u << 16);
        uint32_t P[4][4];               // last four row pairs: per pixel, sums of rows 2j | 2j+1 << 16
#pragma unroll
        for (int j = 0; j < kBlurRows / 2 + 3; ++j) {
            uint32_t h[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const u96 t = *reinterpret_cast<const u96*>(src + (vo + (uint32_t)(2 * j + r) * stride));
                // t.a = pixels x-4..x-1, t.b = x..x+3, t.c = x+4..x+7: instead of shifting the pixels under one pair of
                // tap words, each output uses the tap words shifted to its position (10 dot products per row, no byte align)
                h[r][0] = __builtin_amdgcn_udot4(t.b, B0, __builtin_amdgcn_udot4(t.a, A0, 0u, false), false);
                h[r][1] = __builtin_amdgcn_udot4(t.c, C1, __builtin_amdgcn_udot4(t.b, B1, __builtin_amdgcn_udot4(t.a, A1, 0u, false), false), false);
                h[r][2] = __builtin_amdgcn_udot4(t.c, C2, __builtin_amdgcn_udot4(t.b, B2, __builtin_amdgcn_udot4(t.a, A2, 0u, false), false), false);
                h[r][3] = __builtin_amdgcn_udot4(t.c, T1, __builtin_amdgcn_udot4(t.b, T0, 0u, false), false);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) P[j & 3][k] = h[0][k] | (h[1][k] << 16);
            if (j >= 3) {
                const int m = j - 3;    // output rows 2m, 2m+1 of the strip: row pairs m .. m+3
                uint32_t te[4], to[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#define ORBX_DOT2(a, w, acc) __builtin_amdgcn_udot2(__builtin_bit_cast(us16x2, (uint32_t)(a)), __builtin_bit_cast(us16x2, (uint32_t)(w)), acc, false)
                    const uint32_t d0 = P[m & 3][k], d1 = P[(m + 1) & 3][k], d2 = P[(m + 2) & 3][k], d3 = P[(m + 3) & 3][k];
                    // the rounding constant of (sum + 32768) >> 16 rides in as the accumulator's start value
                    constexpr uint32_t kStart = MODE == 0 ? 32768u : 0u;
                    const uint32_t se = ORBX_DOT2(d3, E3, ORBX_DOT2(d2, E2, ORBX_DOT2(d1, E1, ORBX_DOT2(d0, E0, kStart))));
                    const uint32_t so = ORBX_DOT2(d3, O3, ORBX_DOT2(d2, O2, ORBX_DOT2(d1, O1, ORBX_DOT2(d0, O0, kStart))));
#undef ORBX_DOT2
                    if (MODE == 0) { te[k] = se; to[k] = so; }
                    else {
                        te[k] = se + (even_ties ? 32767u + ((se >> 16) & 1u) : 32768u);
                        to[k] = so + (even_ties ? 32767u + ((so >> 16) & 1u) : 32768u);
                    }
                }
                // bits 16..24 of each sum hold the rounded value (at most 257): gather as 16-bit pairs, clamp, pack bytes
                const us16x2 lim = {255, 255};
                const uint32_t e01 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(te[1], te[0], 0x07060302u)), lim));
                const uint32_t e23 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(te[3], te[2], 0x07060302u)), lim));
                const uint32_t o01 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(to[1], to[0], 0x07060302u)), lim));
                const uint32_t o23 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(to[3], to[2], 0x07060302u)), lim));
                out[4 * (2 * m)] = __builtin_amdgcn_perm(e23, e01, 0x06040200u);
                out[4 * (2 * m + 1)] = __builtin_amdgcn_perm(o23, o01, 0x06040200u);
            }
        }
    }
    wave_sync();
    // lanes 8k..8k+7 store the eight 16-byte rows of one tile: unit c >> 4 of the wavefront, strip row c & 15
    uint8_t* dst = blr + (unsigned long long)frame * blur_bytes + L.blur_offset;
    const uint32_t tile_row_bytes = 128u * (uint32_t)L.tpr;
    const uint32_t unit0 = (blk * 256u + 64u * (uint32_t)wave) >> 2;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const uint32_t c = 64u * s4 + (uint32_t)lane;
        const uint32_t unit = unit0 + (c >> 4), r = c & 15u;
        // (strip, unit in the strip) of unit c >> 4 of the wavefront: lane 4 * (c >> 4) worked them out for its own unit above
        const uint32_t su = (uint32_t)__shfl((int)my_unit, (int)(4u * (c >> 4)));
        const uint32_t y = (su >> 16) * kBlurRows + r, ucs = su & 0xFFFFu;
        if (unit < nunits && y < (uint32_t)L.h) {
            const uint4 v = *reinterpret_cast<const uint4*>(stage + ((c >> 4) * kBlurUnitBytes + 16 * r));
            // (nontemporal: nothing reads the blurred levels before k_describe, three kernels on; the pyramid that
            // k_fast_cells is about to read keeps its place in the caches: headline +0.7 %)
            typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(dst + (__umul24(y >> 3, tile_row_bytes) + 128u * ucs + 16u * (y & 7u))));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Describe: one wavefront per selected keypoint.  The intensity centroid (radius 15) is summed straight
// from the unblurred level, the 37x37 window of the blurred level that the rotated pattern can reach is
// staged in LDS, and the 256 binary tests read it there.
// ---------------------------------------------------------------------------------------------
constexpr int kBlurW = 37, kWinChunks = 4;      // window rows are four 16-byte chunks in LDS, aligned as in memory

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2 scalar path; every operation rounded (no contraction)
    constexpr float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    constexpr float p1 = 0.9997878412794807f * rad2deg, p3 = -0.3258083974640975f * rad2deg;
    constexpr float p5 = 0.1555786518463281f * rad2deg, p7 = -0.04432655554792128f * rad2deg;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    const float eps = 2.22044605e-16f; // (float)DBL_EPSILON
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sin and cos of x in [0, 2*pi] in double precision (error ~1e-16), rounded to float by the caller: the
// correctly rounded float value of cos/sin except when the true value lies within ~1e-16 (relative)
// of a float rounding boundary.  Cody-Waite reduction by pi/2 and Taylor polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_0_2pi(double x, double& sn, double& cs)
{
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = -1.0 / 1307674368000.0;
    ps = fma(ps, z, 1.0 / 6227020800.0);
    ps = fma(ps, z, -1.0 / 39916800.0);
    ps = fma(ps, z, 1.0 / 362880.0);
    ps = fma(ps, z, -1.0 / 5040.0);
    ps = fma(ps, z, 1.0 / 120.0);
    ps = fma(ps, z, -1.0 / 6.0);
    const double s = fma(r * z, ps, r);
    double pc = 1.0 / 20922789888000.0;
    pc = fma(pc, z, -1.0 / 87178291200.0);
    pc = fma(pc, z, 1.0 / 479001600.0);
    pc = fma(pc, z, -1.0 / 3628800.0);
    pc = fma(pc, z, 1.0 / 40320.0);
    pc = fma(pc, z, -1.0 / 720.0);
    pc = fma(pc, z, 1.0 / 24.0);
    pc = fma(pc, z, -0.5);
    const double c = fma(pc, z, 1.0);
    const int n = (int)k & 3;
    sn = (n == 0) ? s : (n == 1) ? c : (n == 2) ? -s : -c;
    cs = (n == 0) ? c : (n == 1) ? -s : (n == 2) ? -c : s;
}

// glibc's sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h since 2.28: a double-precision
// polynomial after one reduction by pi/2), for x = theta in [0, 120): what src/ORBextractor.cc:112 calls through the
// float overloads of cos / sin.  FMA = true follows the build x86-64 CPUs with FMA and AVX2 run (the ifunc'ed
// __sinf_fma / __cosf_fma: the compiler contracted `a + b * c` into fused multiply-adds, read off the library's code);
// FMA = false is the same source evaluated operation by operation (the SSE2 build).  Constants: __sincosf_table.
template <bool FMA> __device__ __forceinline__ double glibc_mad(double a, double b, double c)   // a * b + c
{
    return FMA ? __fma_rn(a, b, c) : __dadd_rn(__dmul_rn(a, b), c);
}
template <bool FMA> __device__ __forceinline__ float glibc_sincos_poly(double x, double x2, bool negate, bool cosine)
{
    const double sg = negate ? -1.0 : 1.0;                     // __sincosf_table[1] holds c0..c4 negated
    if (!cosine) {
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        const double x3 = __dmul_rn(x, x2);
        const double t = glibc_mad<FMA>(x2, s3, s2);           // s1' = s2 + x2 * s3
        const double x7 = __dmul_rn(x3, x2);
        const double u = glibc_mad<FMA>(x3, s1, x);            // s = x + x3 * s1
        return __double2float_rn(glibc_mad<FMA>(x7, t, u));    // s + x7 * s1'
    }
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
                 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double x4 = __dmul_rn(x2, x2);
    const double t2 = glibc_mad<FMA>(x2, c4, c3);              // c2' = c3 + x2 * c4
    const double t1 = glibc_mad<FMA>(x2, c1, c0);              // c1' = c0 + x2 * c1
    const double x6 = __dmul_rn(x4, x2);
    const double c = glibc_mad<FMA>(x4, c2, t1);               // c = c1' + x4 * c2
    return __double2float_rn(glibc_mad<FMA>(x6, t2, c));       // c + x6 * c2'
}
template <bool FMA> __device__ __forceinline__ void glibc_sincosf(float y, float& sn, float& cs)
{
    const uint32_t top12 = (__float_as_uint(y) >> 20) & 0x7FFu;
    double x = (double)y;
    if (top12 < 0x3F4u) {                                      // abstop12(y) < abstop12(pi/4)
        if (top12 < 0x398u) { sn = y; cs = 1.0f; return; }     // |y| < 2^-12
        const double x2 = __dmul_rn(x, x);
        sn = glibc_sincos_poly<FMA>(x, x2, false, false);
        cs = glibc_sincos_poly<FMA>(x, x2, false, true);
        return;
    }
    // reduce_fast without round-to-int instructions: n = round(x * 2/pi) through the 2^24-scaled constant
    const double r = __dmul_rn(x, 0x1.45F306DC9C883p+23);
    const int n = ((int)r + 0x800000) >> 24;
    const double hpi = 0x1.921FB54442D18p0;
    x = FMA ? __fma_rn(-(double)n, hpi, x) : __dsub_rn(x, __dmul_rn((double)n, hpi));
    const double x2 = __dmul_rn(x, x);
    const double xs = __dmul_rn(x, (n + 1) & 2 ? -1.0 : 1.0);  // sign[n & 3] = {1, -1, -1, 1}
    const bool neg = n & 2;
    sn = glibc_sincos_poly<FMA>(xs, x2, neg, n & 1);           // sinf: sinf_poly(x * s, x2, p, n)
    cs = glibc_sincos_poly<FMA>(xs, x2, neg, !(n & 1));        // cosf: sinf_poly(x * s, x2, p, n ^ 1)
}

// cos and sin of theta as orbx_params.trig_mode defines them
__device__ __forceinline__ void trig_by_mode(int mode, float theta, float& a, float& b)
{
    if (mode == 1) glibc_sincosf<true>(theta, b, a);
    else if (mode == 2) glibc_sincosf<false>(theta, b, a);
    else {
        double sd, cd;
        sincos_0_2pi((double)theta, sd, cd);
        a = (float)cd; b = (float)sd;
    }
}

// cos / sin of angle * (float)(pi / 180) for the float angles whose bit patterns are first .. first + count - 1
__global__ __launch_bounds__(256) void k_trig_range(int mode, uint32_t first, uint32_t count, float* __restrict__ cs, float* __restrict__ sn)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    constexpr float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);
    float a, b;
    trig_by_mode(mode, __fmul_rn(__uint_as_float(first + i), factorPI), a, b);
    cs[i] = a; sn[i] = b;
}

// Byte weights of the intensity-centroid sums for the eight dwords of a disc row |v| read from column u = -16:
// wu = u + 16 inside the disc, wm = 1 inside the disc, 0 outside.
struct IcTable { uint32_t wu[16][8]; uint32_t wm[16][8]; };
constexpr IcTable make_ic_table()
{
    constexpr int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    IcTable t{};
    for (int v = 0; v < 16; ++v)
        for (int i = 0; i < 8; ++i) {
            uint32_t wu = 0, wm = 0;
            for (int b = 0; b < 4; ++b) {
                const int u = 4 * i + b - 16;
                if (u >= -um[v] && u <= um[v]) { wu |= (uint32_t)(u + 16) << (8 * b); wm |= 1u << (8 * b); }
            }
            t.wu[v][i] = wu; t.wm[v][i] = wm;
        }
    return t;
}
__device__ __constant__ IcTable c_ic = make_ic_table();

// the 256 test pairs as floats (x0, y0, x1, y1)
struct PatternF { float v[1024]; };
constexpr PatternF make_pattern_f()
{
    constexpr signed char src[1024] = {
#include "../../include/orbx_pattern.inc"
    };
    PatternF t{};
    for (int i = 0; i < 1024; ++i) t.v[i] = (float)src[i];
    return t;
}
__device__ __constant__ __align__(16) PatternF c_pattern_f = make_pattern_f();

// A wavefront describes kKpPerWave consecutive keypoint slots one after the other, so that what depends only on
// the lane (disc weights, load offsets in units of rows and bytes) is set up once.
constexpr int kKpPerWave = 4;

// kOwnBlur: there are no blurred levels -- the wavefront fetches the 43x43 pixels of the unblurred level around its keypoint
// and blurs the 37x37 window itself, in LDS, with the arithmetic of k_blur (same taps, same single rounding, same groups
// of four columns).  That is about 1.6 k blurred pixels per keypoint: cheaper than blurring every level when the image
// has many more pixels than keypoints x 1.6 k (3840x2160 with 8000 features), dearer otherwise (orbx_extractor::own_blur).
constexpr int kRawW = 43, kRawRows = 44;           // window rows fetched (one spare row: the last row pair reads it, nothing uses it)
constexpr int kHGroups = 10;                       // aligned groups of four columns that hold the 37 window columns (first one: the group of byte a + 3)
template <bool kOwnBlur, int kBlurMode>
__global__ __launch_bounds__(256) void k_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blr, unsigned long long pyr_bytes,
                                                  unsigned long long blur_bytes, const DevLevel* __restrict__ lv, int nlevels,
                                                  const uint32_t* __restrict__ lvlkp, int kp_total,
                                                  const int32_t* __restrict__ lvlcnt,
                                                  orbx_keypoint* __restrict__ okp, uint8_t* __restrict__ odesc,
                                                  int32_t* __restrict__ ocount, int cap_per_frame, int n_frames, int trig_mode)
{
    // (own blur: one leading chunk, so that the group left of the first one can be read, then the raw rows; the blurred
    // window later takes the place of the raw rows)
    __shared__ __align__(16) uint4 s_win[4][kOwnBlur ? 1 + kRawRows * kWinChunks : kBlurW * kWinChunks + 1];
    __shared__ __align__(16) uint4 s_h[kOwnBlur ? 4 : 1][kOwnBlur ? (kRawRows / 2) * kHGroups : 1];   // horizontal sums, two rows per word
    __shared__ __align__(16) float4 s_pat[256];
    __shared__ __align__(16) uint32_t s_ic[256];
    // results of a wavefront's four keypoints, stored together at the end: 4 x 32 descriptor bytes, 4 x 7 record words
    // (eight apart), 4 output rows (-1: slot not in use)
    __shared__ __align__(16) unsigned long long s_desc[4][4 * 4];
    __shared__ __align__(16) uint32_t s_rec[4][4 * 8];
    __shared__ __align__(16) int32_t s_row[4][4];
    static_assert(sizeof(PatternF) == 256 * sizeof(float4) && sizeof(IcTable) == 256 * sizeof(uint32_t), "one entry per thread");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // slot arithmetic stays scalar
    int frame; uint32_t chunk;
    if (!frame_chunk(n_frames, frame, chunk)) return;                    // (the same for the whole workgroup)
    // The lane constants of every wavefront -- 4 KB of pattern points, 1 KB of disc weights -- come in once per workgroup
    // and reach the four wavefronts through LDS: fetched per wavefront they were a quarter of the kernel's requests to
    // the address unit, which is what bounds it.
    s_pat[threadIdx.x] = reinterpret_cast<const float4*>(c_pattern_f.v)[threadIdx.x];
    s_ic[threadIdx.x] = reinterpret_cast<const uint32_t*>(&c_ic)[threadIdx.x];
    __syncthreads();
    const int slot0 = ((int)chunk * 4 + wave) * kKpPerWave;
    if (slot0 >= kp_total) return;
    uint4* win = s_win[wave] + (kOwnBlur ? 1 : 0);
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);

    // lane constants.  Window: 16-byte chunk l = lane, lane + 64 of the 37 x 3 chunks -> (row, byte).  Disc: lane
    // 2*row + half reads 16 bytes of disc row `row` from column -16 + 16*half, with the weights of its four dwords.
    uint32_t wcol[3], wrem[3];          // chunk column and row offset from the first tile row of window slot lane + 64 k
#pragma unroll
    for (int k = 0; k < 3; ++k) { const uint32_t sl = lane + 64 * k; wcol[k] = sl / 48u; wrem[k] = sl - 48u * wcol[k]; }
    const bool disc_lane = lane < 2 * kPatch;
    const int crow = min(lane >> 1, kPatch - 1), chalf = lane & 1;
    const int cv = crow - kHalfPatch;
    uint32_t cwu[4], cwm[4];
    {
        const int av = cv < 0 ? -cv : cv;
        const uint4 wu4 = reinterpret_cast<const uint4*>(s_ic)[2 * av + chalf];            // IcTable::wu[av][4 * chalf ..]
        const uint4 wm4 = reinterpret_cast<const uint4*>(s_ic)[32 + 2 * av + chalf];       // IcTable::wm[av][4 * chalf ..]
        cwu[0] = disc_lane ? wu4.x : 0u; cwu[1] = disc_lane ? wu4.y : 0u; cwu[2] = disc_lane ? wu4.z : 0u; cwu[3] = disc_lane ? wu4.w : 0u;
        cwm[0] = disc_lane ? wm4.x : 0u; cwm[1] = disc_lane ? wm4.y : 0u; cwm[2] = disc_lane ? wm4.z : 0u; cwm[3] = disc_lane ? wm4.w : 0u;
    }

    // (level, index, output row) of the wave's slots and their keys, all scalar
    int jlevel[kKpPerWave], jidx[kKpPerWave], jout[kKpPerWave];
#pragma unroll
    for (int q = 0; q < kKpPerWave; ++q) { jlevel[q] = -1; jidx[q] = 0; jout[q] = 0; }
    int total = 0;
    for (int l = 0; l < nlevels; ++l) {
        const int cnt = lvlcnt[frame * nlevels + l];
        const int o = lv[l].kp_offset;
#pragma unroll
        for (int q = 0; q < kKpPerWave; ++q) {
            const int i = slot0 + q - o;
            if (i >= 0 && i < cnt && total + i < cap_per_frame) { jlevel[q] = l; jidx[q] = i; jout[q] = total + i; }
        }
        total += cnt;
    }
    if (slot0 == 0 && lane == 0) ocount[frame] = total;
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kKpPerWave; ++q) s_row[wave][q] = jlevel[q] >= 0 ? jout[q] : -1;
    }
    uint32_t jkey[kKpPerWave];
#pragma unroll
    for (int q = 0; q < kKpPerWave; ++q)
        jkey[q] = jlevel[q] >= 0 ? lvlkp[(unsigned long long)frame * kp_total + lv[jlevel[q]].kp_offset + jidx[q]] : 0u;

    // Schedule: the disc loads of all four keypoints and the window loads of the first two go out first; the four
    // orientations are then computed together, keypoint q in lane q (one pass through fastAtan2 and the double-precision
    // sin / cos instead of four); each keypoint's 256 tests run with the window loads of the keypoint two ahead in flight.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct Window { u32x4 w[3]; };
    auto key_xy = [&](int q, int& kx, int& ky) __attribute__((always_inline)) {
        kx = (int)(jkey[q] >> 20) + kMinBorder; ky = (int)((jkey[q] >> 8) & 0xFFF) + kMinBorder;
    };
    auto fetch_disc = [&](int q, u32x4& d) __attribute__((always_inline)) {
        d = u32x4{0u, 0u, 0u, 0u};
        if (jlevel[q] < 0 || !disc_lane) return;
        const DevLevel& L = lv[jlevel[q]];
        int kx, ky; key_xy(q, kx, ky);
        // disc rows ky-15..ky+15 of the unblurred level, bytes kx-16..kx+15
        const uint8_t* csrc = pyr + (unsigned long long)frame * pyr_bytes + L.roi + (long long)(ky - kHalfPatch) * L.stride + (kx - 16);
        const u128_unaligned t = *reinterpret_cast<const u128_unaligned*>(csrc + (__umul24((uint32_t)crow, (uint32_t)L.stride) + 16u * (uint32_t)chalf));
        d = u32x4{t.x, t.y, t.z, t.w};
    };
    auto fetch_window = [&](int q, Window& f) __attribute__((always_inline)) {
        if (jlevel[q] < 0) return;
        const DevLevel& L = lv[jlevel[q]];
        int kx, ky; key_xy(q, kx, ky);
        if constexpr (kOwnBlur) {
            // rows ky-21..ky+21 of the unblurred level, the four 16-byte chunks from the one that holds column kx-21: slot
            // lane + 64 k is chunk (slot & 3) of row slot >> 2, so four lanes read 64 contiguous bytes
            const uint8_t* rsrc = pyr + (unsigned long long)frame * pyr_bytes + L.roi + (long long)(ky - 21) * L.stride + (long long)((kx - 21) & ~15);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t sl = (uint32_t)lane + 64u * k;
                if ((sl >> 2) < (uint32_t)kRawW) f.w[k] = *reinterpret_cast<const u32x4*>(rsrc + (__umul24(sl >> 2, (uint32_t)L.stride) + 16u * (sl & 3u)));
            }
            return;
        }
        // blurred window rows ky-18..ky+18, the four 16-byte columns from the one holding column kx-18, out of the
        // tiled level: tile (y >> 3, column) holds row y at byte 16 * (y & 7).  The 192 slots of the three loads walk
        // the memory order -- column, tile row (six cover the 37 rows wherever they start), row inside the tile -- so every
        // aligned group of four lanes reads 64 contiguous bytes; slots outside the window stay idle.
        const uint8_t* bsrc = blr + (unsigned long long)frame * blur_bytes + L.blur_offset + 128u * (uint32_t)((kx - 18) >> 4);
        const uint32_t tile_row_bytes = 128u * (uint32_t)L.tpr;
        const uint32_t ytile = (uint32_t)(ky - 18) >> 3, off7 = (uint32_t)(ky - 18) & 7u;
        auto tiled = [&](int k) __attribute__((always_inline)) {
            return __umul24(ytile + (wrem[k] >> 3), tile_row_bytes) + 128u * wcol[k] + 16u * (wrem[k] & 7u);
        };
        // the 37 columns start at byte (kx-18) & 15 of the first chunk: the fourth chunk of a row is only touched when
        // that offset is 12 or more (a quarter of the keypoints)
        const uint32_t last = ((uint32_t)(kx - 18) & 15u) >= 12u ? 4u : 3u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (wrem[k] - off7 < (uint32_t)kBlurW && wcol[k] < last) f.w[k] = *reinterpret_cast<const u32x4*>(bsrc + tiled(k));
    };
    // IC_Angle (src/ORBextractor.cc:76-103): integer moments over the 749-pixel disc, four pixels per
    // v_dot4_u32_u8.  m10 = sum u*I = sum (u+16)*I - 16*sum I,  m01 = sum v*I; bytes outside the disc have weight 0.
    auto moments = [&](const u32x4& d, int& m10, int& m01) __attribute__((always_inline)) {
        uint32_t su = 0, sm = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            su = __builtin_amdgcn_udot4(d[k], cwu[k], su, false);
            sm = __builtin_amdgcn_udot4(d[k], cwm[k], sm, false);
        }
        m10 = (int)su - 16 * (int)sm;                      // lanes outside the disc hold zero weights
        m01 = cv * (int)sm;
        m10 = wave_sum_i32(m10); m01 = wave_sum_i32(m01);
    };
    static_assert(kKpPerWave == 4, "the schedule below is written out for four keypoints");
    u32x4 disc[kKpPerWave];
    Window fa, fb;
#pragma unroll
    for (int q = 0; q < kKpPerWave; ++q) fetch_disc(q, disc[q]);
    fetch_window(0, fa);
    fetch_window(1, fb);
    int my10 = 0, my01 = 0;
#pragma unroll
    for (int q = 0; q < kKpPerWave; ++q) {
        int m10, m01;
        moments(disc[q], m10, m01);
        if (lane == q) { my10 = m10; my01 = m01; }
    }
    const float my_angle = fast_atan2_deg((float)my01, (float)my10);
    // computeOrbDescriptor (src/ORBextractor.cc:107-146)
    constexpr float factorPI = (float)(3.1415926535897932384626433832795 / 180.0); // (float)(CV_PI/180.f)
    float my_a, my_b;
    trig_by_mode(trig_mode, __fmul_rn(my_angle, factorPI), my_a, my_b);

    auto stage = [&](int q, const Window& f) __attribute__((always_inline)) {      // window registers -> LDS
        if (jlevel[q] < 0) return;
        wave_sync();        // the previous keypoint's tests have read the window
        u32x4* winv = reinterpret_cast<u32x4*>(win);
        int kx, ky; key_xy(q, kx, ky);
        if constexpr (kOwnBlur) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t sl = (uint32_t)lane + 64u * k;
                if ((sl >> 2) < (uint32_t)kRawW) winv[sl] = f.w[k];
            }
            wave_sync();
            const DevLevel& L = lv[jlevel[q]];
            const uint32_t a = (uint32_t)(kx - 21) & 15u;              // byte of column kx-21 inside its chunk row
            const uint32_t G0 = (a + 3u) >> 2;                         // group of window column 0 (its centre byte is a + 3)
            const int xg0 = ((kx - 21) & ~15) + 4 * (int)G0;           // image column of that group's first pixel
            const uint32_t* rawd = reinterpret_cast<const uint32_t*>(win);
            constexpr uint32_t T0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), T1 = 49u | (34u << 8) | (18u << 16);
            constexpr uint32_t A0 = (18u << 8) | (34u << 16) | (49u << 24), B0 = 55u | (49u << 8) | (34u << 16) | (18u << 24);
            constexpr uint32_t A1 = (18u << 16) | (34u << 24), B1 = 49u | (55u << 8) | (49u << 16) | (34u << 24), C1 = 18u;
            constexpr uint32_t A2 = 18u << 24, B2 = 34u | (49u << 8) | (55u << 16) | (49u << 24), C2 = 34u | (18u << 8);
            constexpr uint32_t E0 = 18u | (34u << 16), E1 = 49u | (55u << 16), E2 = 49u | (34u << 16), E3 = 18u;
            constexpr uint32_t O0 = 18u << 16, O1 = 34u | (49u << 16), O2 = 55u | (49u << 16), O3 = 34u | (18u << 16);
            static_assert(kHGroups == 10 && (kRawRows / 2) * kHGroups <= 256, "the reciprocal below divides by ten, for task numbers under 256");
            // horizontal pass: task = (row pair, group); per pixel the sums of rows 2p | 2p+1 << 16, as in k_blur
            uint4* hv = s_h[wave];
#pragma unroll
            for (int it = 0; it < ((kRawRows / 2) * kHGroups + 63) / 64; ++it) {
                const uint32_t t = (uint32_t)lane + 64u * it;
                if (t < (uint32_t)((kRawRows / 2) * kHGroups)) {
                    const uint32_t pr = (t * 6554u) >> 16, gi = t - pr * (uint32_t)kHGroups;      // t / 10 for t < 256
                    uint32_t h[2][4];
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const uint32_t* rp = rawd + ((2u * pr + r) * 16u + G0 + gi);
                        const uint32_t ta = rp[-1], tb = rp[0], tc = rp[1];
                        h[r][0] = __builtin_amdgcn_udot4(tb, B0, __builtin_amdgcn_udot4(ta, A0, 0u, false), false);
                        h[r][1] = __builtin_amdgcn_udot4(tc, C1, __builtin_amdgcn_udot4(tb, B1, __builtin_amdgcn_udot4(ta, A1, 0u, false), false), false);
                        h[r][2] = __builtin_amdgcn_udot4(tc, C2, __builtin_amdgcn_udot4(tb, B2, __builtin_amdgcn_udot4(ta, A2, 0u, false), false), false);
                        h[r][3] = __builtin_amdgcn_udot4(tc, T1, __builtin_amdgcn_udot4(tb, T0, 0u, false), false);
                    }
                    // (the pairing as one byte select: the compiler makes a shift and an or of `a | b << 16`; not as inline
                    // assembly -- behind a dot product the compiler must see the reader to keep the required distance)
                    hv[t] = make_uint4(__builtin_amdgcn_perm(h[1][0], h[0][0], 0x05040100u), __builtin_amdgcn_perm(h[1][1], h[0][1], 0x05040100u),
                                       __builtin_amdgcn_perm(h[1][2], h[0][2], 0x05040100u), __builtin_amdgcn_perm(h[1][3], h[0][3], 0x05040100u));
                }
            }
            wave_sync();
            // vertical pass: task = (pair of window rows 2m, 2m+1, group) over the row pairs m .. m+3; the blurred bytes go
            // where the raw ones were, window column c at byte a + 3 + c of its 64-byte row
            uint32_t* outd = reinterpret_cast<uint32_t*>(win);
#pragma unroll
            for (int it = 0; it < (((kBlurW + 1) / 2) * kHGroups + 63) / 64; ++it) {
                const uint32_t t = (uint32_t)lane + 64u * it;
                if (t < (uint32_t)(((kBlurW + 1) / 2) * kHGroups)) {
                    const uint32_t m = (t * 6554u) >> 16, gi = t - m * (uint32_t)kHGroups;
                    const uint4 p0 = hv[m * kHGroups + gi], p1 = hv[(m + 1) * kHGroups + gi], p2 = hv[(m + 2) * kHGroups + gi], p3 = hv[(m + 3) * kHGroups + gi];
                    const uint32_t d0[4] = {p0.x, p0.y, p0.z, p0.w}, d1[4] = {p1.x, p1.y, p1.z, p1.w}, d2[4] = {p2.x, p2.y, p2.z, p2.w}, d3[4] = {p3.x, p3.y, p3.z, p3.w};
                    const bool even_ties = kBlurMode != 0 && xg0 + 4 * (int)gi < (L.w & ~3);
                    uint32_t te[4], to[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#define ORBX_DOT2(a_, w_, acc) __builtin_amdgcn_udot2(__builtin_bit_cast(us16x2, (uint32_t)(a_)), __builtin_bit_cast(us16x2, (uint32_t)(w_)), acc, false)
                        // the rounding constant of (sum + 32768) >> 16 rides in as the accumulator's start value
                        constexpr uint32_t kStart = kBlurMode == 0 ? 32768u : 0u;
                        const uint32_t se = ORBX_DOT2(d3[k], E3, ORBX_DOT2(d2[k], E2, ORBX_DOT2(d1[k], E1, ORBX_DOT2(d0[k], E0, kStart))));
                        const uint32_t so = ORBX_DOT2(d3[k], O3, ORBX_DOT2(d2[k], O2, ORBX_DOT2(d1[k], O1, ORBX_DOT2(d0[k], O0, kStart))));
#undef ORBX_DOT2
                        if (kBlurMode == 0) { te[k] = se; to[k] = so; }
                        else {
                            te[k] = se + (even_ties ? 32767u + ((se >> 16) & 1u) : 32768u);
                            to[k] = so + (even_ties ? 32767u + ((so >> 16) & 1u) : 32768u);
                        }
                    }
                    const us16x2 lim = {255, 255};
                    const uint32_t e01 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(te[1], te[0], 0x07060302u)), lim));
                    const uint32_t e23 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(te[3], te[2], 0x07060302u)), lim));
                    const uint32_t o01 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(to[1], to[0], 0x07060302u)), lim));
                    const uint32_t o23 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us16x2, __builtin_amdgcn_perm(to[3], to[2], 0x07060302u)), lim));
                    outd[(2u * m) * 16u + G0 + gi] = __builtin_amdgcn_perm(e23, e01, 0x06040200u);
                    if (2u * m + 1u < (uint32_t)kBlurW) outd[(2u * m + 1u) * 16u + G0 + gi] = __builtin_amdgcn_perm(o23, o01, 0x06040200u);
                }
            }
            return;
        }
        const uint32_t off7 = (uint32_t)(ky - 18) & 7u;
        const uint32_t last = ((uint32_t)(kx - 18) & 15u) >= 12u ? 4u : 3u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (wrem[k] - off7 < (uint32_t)kBlurW && wcol[k] < last) winv[(wrem[k] - off7) * kWinChunks + wcol[k]] = f.w[k];
    };
    auto process = [&](int q) __attribute__((always_inline)) {
        if (jlevel[q] < 0) return;
        const int level = jlevel[q];
        const DevLevel& L = lv[level];
        int kx, ky; key_xy(q, kx, ky);
        const int resp = (int)(jkey[q] & 0xFF);
        const float angle = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_angle), q));
        const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_a), q));
        const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_b), q));
        const int wshift = kOwnBlur ? (int)((uint32_t)(kx - 21) & 15u) + 3 : (kx - 18) & 15;
        wave_sync();
        // cvRound by adding 1.5 * 2^23: the low bits of the sum's encoding are the rounded integer (ties to even)
        constexpr float kMagic = 12582912.f;
        constexpr uint32_t kBias = 0x4B400000u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (read where it is used: held in registers for the whole kernel the 256 pairs cost sixteen of them and the
            // eighth wavefront per SIMD)
            const float4 pt = s_pat[g * 64 + lane];
            const uint32_t r0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(pt.x, b), __fmul_rn(pt.y, a)), kMagic)) - kBias;
            const uint32_t c0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(pt.x, a), __fmul_rn(pt.y, b)), kMagic)) - kBias;
            const uint32_t r1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(pt.z, b), __fmul_rn(pt.w, a)), kMagic)) - kBias;
            const uint32_t c1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(pt.z, a), __fmul_rn(pt.w, b)), kMagic)) - kBias;
            const int t0 = wb[(int)(18 + r0) * (16 * kWinChunks) + (int)(18 + c0) + wshift];
            const int t1 = wb[(int)(18 + r1) * (16 * kWinChunks) + (int)(18 + c1) + wshift];
            const unsigned long long bits = __ballot(t0 < t1);
            if (lane == 0) s_desc[wave][4 * q + g] = bits;
        }
        if (lane == 0) {
            float x = (float)kx, y = (float)ky;
            if (level != 0) { x = __fmul_rn(x, L.scale); y = __fmul_rn(y, L.scale); }
            // orbx_keypoint: x, y, size, angle, response, octave, class_id
            uint4* rec = reinterpret_cast<uint4*>(&s_rec[wave][8 * q]);
            rec[0] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(L.kp_size), __float_as_uint(angle));
            rec[1] = make_uint4(__float_as_uint((float)resp), (uint32_t)level, 0xFFFFFFFFu, 0u);
        }
    };
    stage(0, fa); fetch_window(2, fa); process(0);
    stage(1, fb); fetch_window(3, fb); process(1);
    stage(2, fa); process(2);
    stage(3, fb); process(3);
    // One store instruction for the four descriptors (lane 4q + g: the g-th eight bytes of keypoint q) and one for the four
    // records (lane 7q + f: word f of keypoint q), instead of six single-lane stores per keypoint -- those were a seventh
    // of the kernel's time.  Consecutive slots mostly have consecutive output rows, so the lanes' addresses adjoin.
    static_assert(sizeof(orbx_keypoint) == 28, "seven words per record");
    wave_sync();
    if (lane < 4 * kKpPerWave) {
        const int row = s_row[wave][lane >> 2];
        if (row >= 0) reinterpret_cast<unsigned long long*>(odesc + ((size_t)frame * cap_per_frame + row) * 32)[lane & 3] = s_desc[wave][lane];
    }
    if (lane < 7 * kKpPerWave) {
        const int q = (lane * 37) >> 8, f = lane - 7 * q;                  // lane / 7 for lane < 28
        const int row = s_row[wave][q];
        if (row >= 0) reinterpret_cast<uint32_t*>(okp + ((size_t)frame * cap_per_frame + row))[f] = s_rec[wave][8 * q + f];
    }
}

// ---------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches, sub-pixel part (src/Frame.cc:551-623).  A wavefront takes kRefineGroup consecutive left
// keypoints: a lane each reads its keypoint's match and decides whether it is refined at all (Hamming distance below
// (TH_HIGH + TH_LOW) / 2: about a tenth of them), then the wavefront works through the ones that are, one after the
// other with all its lanes -- 11x11 windows of the two extractors' pyramids at the keypoint's level, SAD (L1 norm of the
// centre-subtracted windows: integer-valued, so exact in integers) for the 11 column shifts, parabola fit and disparity
// test in float exactly as written in the reference -- and every lane stores its own keypoint's result.
// ---------------------------------------------------------------------------------------------
constexpr int kRefineGroup = 16;
__global__ __launch_bounds__(256) void k_stereo_refine(const uint8_t* __restrict__ pyrL, const uint8_t* __restrict__ pyrR,
                                                       unsigned long long pyr_bytes, const DevLevel* __restrict__ lv, int nlevels,
                                                       const orbx_keypoint* __restrict__ kl, const int32_t* __restrict__ nl_arr,
                                                       const orbx_keypoint* __restrict__ kr, int cap,
                                                       const int32_t* __restrict__ best_dist, const int32_t* __restrict__ best_idx,
                                                       float bf, float min_d, float max_d,
                                                       float* __restrict__ uright, float* __restrict__ depth, int32_t* __restrict__ sad)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.y;
    const int i0 = (blockIdx.x * 4 + wave) * kRefineGroup;
    const int nl = min(nl_arr[pair], cap);
    if (i0 >= nl) return;
    constexpr int w = 5, R = 5;
    constexpr int thOrbDist = (100 + 50) / 2;
    // every lane: what its own keypoint needs (the part of :551-585 that involves no pixels).  Keypoints whose windows
    // would leave the level get no match (rowRange/colRange outside the level throw in OpenCV; keypoints lie at least 19
    // pixels inside their level, so this does not happen on extractor output).
    const bool mine = lane < kRefineGroup && i0 + lane < nl;
    const size_t row = (size_t)pair * cap + i0 + (mine ? lane : 0);
    float kx = 0.f, lscale = 0.f, scaleduR0 = 0.f;
    uint32_t offL = 0, offR = 0, stride = 0;                 // of the window centres inside a frame's pyramid; bytes per row
    bool cand = false;
    if (mine) {
        const int bd = best_dist[row];
        if (bd >= 0 && bd < thOrbDist) {
            const orbx_keypoint k = kl[row];
            if (k.octave >= 0 && k.octave < nlevels) {
                const DevLevel& L = lv[k.octave];
                const float uR0 = kr[(size_t)pair * cap + best_idx[row]].x;
                const float sf = L.inv_scale;
                const float scaleduL = roundf(__fmul_rn(k.x, sf)), scaledvL = roundf(__fmul_rn(k.y, sf));
                scaleduR0 = roundf(__fmul_rn(uR0, sf));
                const float iniu = scaleduR0 + (float)(R - w), endu = scaleduR0 + (float)(R + w + 1);
                const int yl = (int)scaledvL, xl = (int)scaleduL, xr = (int)scaleduR0;
                const bool inside = yl - w >= 0 && yl + w < L.h && xl - w >= 0 && xl + w < L.w && xr - R - w >= 0;
                if (!(iniu < 0.f || endu >= (float)L.w) && inside) {
                    cand = true;
                    kx = k.x; lscale = L.scale; stride = (uint32_t)L.stride;
                    offL = (uint32_t)L.roi + (uint32_t)yl * stride + (uint32_t)xl;
                    offR = (uint32_t)L.roi + (uint32_t)yl * stride + (uint32_t)xr;
                }
            }
        }
    }
    // window position of the lane in the two passes over the 121 pixels (pass 1: 57 lanes)
    const int dy0 = (lane * 373 >> 12) - w, dx0 = lane - (dy0 + w) * (2 * w + 1) - w;               // lane / 11 for lane < 128
    const int pos1 = lane + 64;
    const int dy1 = (pos1 * 373 >> 12) - w, dx1 = pos1 - (dy1 + w) * (2 * w + 1) - w;
    const bool in1 = pos1 < (2 * w + 1) * (2 * w + 1);
    const uint8_t* imL = pyrL + (unsigned long long)pair * pyr_bytes;
    const uint8_t* imR = pyrR + (unsigned long long)pair * pyr_bytes;
    // the wavefront works through its candidates: SAD of the centre-subtracted 11x11 windows for the 11 shifts,
    // |(l - cL) - (r_i - cR_i)| = |(l - r_i) - (cL - cR_i)| with the centre pair taken from the lane that holds it (pass 0,
    // lane 60), so that one round of loads serves a keypoint
    int r_best = -1, r_inc = 0, r_prev = 0, r_next = 0;
    unsigned long long todo = __ballot(cand);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t oL = (uint32_t)__builtin_amdgcn_readlane((int)offL, src), oR = (uint32_t)__builtin_amdgcn_readlane((int)offR, src);
        const int st = __builtin_amdgcn_readlane((int)stride, src);
        const uint8_t* pl0 = imL + oL + (dy0 * st + dx0);
        const uint8_t* pr0 = imR + oR + (dy0 * st + dx0 - R);
        const uint8_t* pl1 = imL + oL + (in1 ? dy1 * st + dx1 : 0);
        const uint8_t* pr1 = imR + oR + (in1 ? dy1 * st + dx1 - R : -R);
        const int l0 = pl0[0], l1 = pl1[0];
        int t0[2 * R + 1], t1[2 * R + 1];
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) { t0[i] = l0 - (int)pr0[i]; t1[i] = l1 - (int)pr1[i]; }
        int acc[2 * R + 1];
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) {
            const int c = __builtin_amdgcn_readlane(t0[i], 60);
            acc[i] = wave_sum_i32(abs(t0[i] - c) + (in1 ? abs(t1[i] - c) : 0));
        }
        int bestDist = 2147483647, bestinc = 0;
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) if (acc[i] < bestDist) { bestDist = acc[i]; bestinc = i - R; }
        int prev = 0, next = 0;
#pragma unroll
        for (int i = 1; i < 2 * R; ++i) if (i - R == bestinc) { prev = acc[i - 1]; next = acc[i + 1]; }
        if (lane == src) { r_best = bestDist; r_inc = bestinc; r_prev = prev; r_next = next; }
    }
    // every lane: parabola fit and disparity test of its own keypoint (:600-623)
    float out_u = -1.0f, out_d = -1.0f;
    int out_s = -1;
    if (cand && r_inc != -R && r_inc != R) {
        const float d1 = (float)r_prev, d2 = (float)r_best, d3 = (float)r_next;
        const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
        if (!(deltaR < -1.f || deltaR > 1.f)) {
            float bestuR = __fmul_rn(lscale, __fadd_rn(__fadd_rn(scaleduR0, (float)r_inc), deltaR));
            float disparity = __fsub_rn(kx, bestuR);
            if (disparity >= min_d && disparity < max_d) {
                if (disparity <= 0.f) { disparity = 0.01f; bestuR = (float)((double)kx - 0.01); }
                out_d = __fdiv_rn(bf, disparity);
                out_u = bestuR;
                out_s = r_best;
            }
        }
    }
    if (mine) { uright[row] = out_u; depth[row] = out_d; sad[row] = out_s; }
}

// Median filter of the accepted SAD distances (src/Frame.cc:626-639).  The reference sorts (distance, left index) pairs
// and uses the distance of element size/2.
constexpr int kMedianRegs = 24;      // distances a lane keeps in registers: pairs of up to 64 * 24 left keypoints take the fast path
__global__ __launch_bounds__(256) void k_stereo_median(const int32_t* __restrict__ nl_arr, int cap, const int32_t* __restrict__ sad,
                                                       float* __restrict__ uright, float* __restrict__ depth, int n_pairs)
{
    // one wavefront per pair; the median (element size/2 of the sorted distances, src/Frame.cc:626-629) is a value, so it
    // is found by bisection on the value with a count per step: no sort, no workgroup barrier.  The distances are read
    // once and kept in registers (rejected ones as -1 -> 0x7FFFFFFF, never counted); only pairs with more keypoints than
    // the registers hold walk device memory in every step.
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const int nl = min(nl_arr[pair], cap);
    const size_t base = (size_t)pair * cap;
    const bool in_regs = nl <= 64 * kMedianRegs;
    int mine[kMedianRegs];
    int nv = 0;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < kMedianRegs; ++k) {
            const int i = lane + 64 * k;
            const int sv = i < nl ? sad[base + i] : -1;
            mine[k] = sv >= 0 ? sv : 0x7FFFFFFF;
            nv += sv >= 0;
        }
    } else {
        for (int i = lane; i < nl; i += 64) nv += sad[base + i] >= 0;
    }
    nv = wave_sum_i32(nv);
    if (nv == 0) return;
    const int kth = nv / 2;
    int lo = 0, hi = 1 << 16;                               // least v with #{s : 0 <= s <= v} > kth; an 11x11 SAD is at most 30 855
    while (lo < hi) {
        const int v = lo + ((hi - lo) >> 1);
        int cnt = 0;
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kMedianRegs; ++k) cnt += mine[k] <= v;
        } else {
            for (int i = lane; i < nl; i += 64) { const int sv = sad[base + i]; cnt += sv >= 0 && sv <= v; }
        }
        cnt = wave_sum_i32(cnt);
        if (cnt > kth) hi = v; else lo = v + 1;
    }
    const float thDist = 1.5f * 1.4f * (float)lo;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < kMedianRegs; ++k) {
            const int i = lane + 64 * k;
            if (mine[k] != 0x7FFFFFFF && !((float)mine[k] < thDist)) { uright[base + i] = -1.0f; depth[base + i] = -1.0f; }
        }
    } else {
        for (int i = lane; i < nl; i += 64) {
            const int sv = sad[base + i];
            if (sv >= 0 && !((float)sv < thDist)) { uright[base + i] = -1.0f; depth[base + i] = -1.0f; }
        }
    }
}

// cv::undistortPoints(src, dst, K, dist, noArray(), K) for the reference's camera model (OpenCV 2.4.9 - 3.4.1
// cvUndistortPoints: K and the coefficients widened to double, five iterations, no early exit).  One thread per point;
// every operation rounds on its own (the file is built with contraction off), as in a baseline x86-64 build of OpenCV.
__device__ __forceinline__ void undistort_point(float px, float py, double fx, double fy, double cx, double cy,
                                                double k1, double k2, double p1, double p2, double k3, float& ox, float& oy)
{
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    double x = px, y = py;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        // the rational-model numerator (k4..k6 = 0) evaluates to exactly 1
        const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // projection through P = K: rows (fx 0 cx), (0 fy cy), (0 0 1), each evaluated left to right
    const double xx = fx * x + 0.0 * y + cx;
    const double yy = 0.0 * x + fy * y + cy;
    const double ww = 1.0 / (0.0 * x + 0.0 * y + 1.0);
    ox = (float)(xx * ww);
    oy = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort_keypoints(const orbx_keypoint* __restrict__ kps, const int32_t* __restrict__ counts, int cap,
                                                             orbx_camera cam, int copy_only, orbx_keypoint* __restrict__ out)
{
    const int frame = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(counts[frame], cap)) return;
    orbx_keypoint k = kps[(size_t)frame * cap + i];
    if (!copy_only) undistort_point(k.x, k.y, cam.fx, cam.fy, cam.cx, cam.cy, cam.k1, cam.k2, cam.p1, cam.p2, cam.k3, k.x, k.y);
    out[(size_t)frame * cap + i] = k;
}

__global__ __launch_bounds__(256) void k_undistort_points(const float2* __restrict__ xy, int n, orbx_camera cam, float2* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 p = xy[i];
    float2 o;
    undistort_point(p.x, p.y, cam.fx, cam.fy, cam.cx, cam.cy, cam.k1, cam.k2, cam.p1, cam.p2, cam.k3, o.x, o.y);
    out[i] = o;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------------------------
struct orbx_extractor {
    orbx_params params;
    Geometry geo;
    int device = 0;
    int max_batch = 0;
    hipStream_t stream = nullptr;
    std::vector<DevLevel> hlv;
    DevLevel* d_lv = nullptr;
    CellRec* d_cells = nullptr;
    FastCell* d_fast_cells = nullptr;  // per launch slot of k_fast_cells: the n_small cells of at most kSmallCellDim pixels, then the n_big others
    int n_small = 0, n_big = 0;
    ResizeX* d_rx = nullptr;
    ResizeY* d_ry = nullptr;
    ResizeRowBlock* d_rrb = nullptr;
    ResizeGroupA* d_rga = nullptr;
    ResizeGroupB* d_rgb = nullptr;
    uint8_t* d_pyr = nullptr;
    uint8_t* d_blur = nullptr;   // Gaussian-blurred copy of every level (ROI pixels only), tiled for the descriptor windows
    int max_blur_threads = 0;
    uint32_t* d_blur_chunks = nullptr;   // (level << 16 | workgroup inside the level) of every workgroup of k_blur for one frame
    int n_blur_chunks = 0;
    int generic_levels = 0;  // levels produced by k_resize_generic
    uint32_t* d_cand = nullptr;
    uint16_t* d_cellcnt = nullptr;
    uint32_t* d_keys = nullptr;
    uint16_t* d_nodeof = nullptr;
    uint32_t* d_lvlkp = nullptr;
    int32_t* d_lvlcnt = nullptr;
    // staging for the host entry points
    float* d_pts = nullptr; float* h_pts = nullptr; int pts_cap = 0;   // staging of orbx_undistort_points
    uint8_t* d_img = nullptr;
    int channels = 1, c0 = 0, c1 = 0, c2 = 0;   // input layout: interleaved channels and their Q14 grey weights
    orbx_keypoint* d_okp = nullptr;
    uint8_t* d_odesc = nullptr;
    int32_t* d_ocount = nullptr;
    uint8_t* h_img = nullptr;            // pinned host mirrors of the staging buffers
    orbx_keypoint* h_okp = nullptr;
    uint8_t* h_odesc = nullptr;
    int32_t* h_ocount = nullptr;
    hipGraphExec_t graph1 = nullptr;     // captured single-frame sequence of orbx_extract / orbx_extract_batch(n_frames = 1)
    bool no_graph = false;
    uint8_t* h_level = nullptr;          // pinned bounce buffer of orbx_pyramid_level
    size_t h_level_bytes = 0;
    int key_total = 0;
    int max_border_threads = 0;
    int NCmax = 0;
    int TC = 0;                     // entries of the quadtree's count tables
    int XY = 0;                     // entries of its per-column / per-row cell index tables
    int KL = 0;              // keys of one level that fit the quadtree's LDS key arrays
    int oct_threads = 256;
    size_t oct_lds = 0;
    unsigned char* d_octmem = nullptr;   // device-memory working set of the quadtree when it does not fit LDS
    size_t oct_gstride = 0;
    int last_frames = 0;
    bool profiling = false;

    hipEvent_t ev[ORBX_STAGE_COUNT + 1] = {};
    hipEvent_t ev_done = nullptr;        // recorded behind the last kernel of every batch, on the stream the batch ran on
    hipEvent_t ev_fork = nullptr;        // orbx_extract_batch_device_multi: recorded on the caller's stream, waited for by the contexts' streams
    bool own_blur = false;               // k_describe blurs each keypoint's window itself; no blurred levels (see k_describe)
    void* sr_dev = nullptr; uint8_t* sr_pin = nullptr; size_t sr_bytes = 0;   // staging of the host form of the stereo refinement
    hipStream_t copy_stream = nullptr;   // uploads of the chunked host batch path
    std::vector<hipEvent_t> chunk_ev;    // one per chunk: upload done
    float stage_ms[ORBX_STAGE_COUNT] = {};
};

// The context's own stream catches up with the last batch (which may have run on a caller's stream), then the host waits.
static hipError_t sync_context(orbx_extractor* ex)
{
    const hipError_t e = hipStreamWaitEvent(ex->stream, ex->ev_done, 0);
    return e != hipSuccess ? e : hipStreamSynchronize(ex->stream);
}

static int select_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return ORBX_ERR_NO_DEVICE; }
    if (device >= count) { set_error("device index out of range"); return ORBX_ERR_BAD_ARG; }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    return ORBX_OK;
}

constexpr int kHostChunk = 32;       // most frames per stage of the host batch pipeline
constexpr int kCopyThreads = 4;      // host threads that move frames into / results out of the pinned mirrors

// runs fn(begin, end) over [0, n) split into kCopyThreads contiguous parts (the calling thread takes the last one)
template <typename F> static void parallel_rows(int n, F fn)
{
    const int parts = std::min(kCopyThreads, std::max(1, n / 64));
    if (parts <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (int p = 0; p + 1 < parts; ++p) th.emplace_back([=]() { fn((int)((long long)n * p / parts), (int)((long long)n * (p + 1) / parts)); });
    fn((int)((long long)n * (parts - 1) / parts), n);
    for (std::thread& t : th) t.join();
}

extern "C" {

#ifdef ORBX_FAST_STAMPS
int orbx_debug_fast_stamps(unsigned long long* out16)
{
    // sums of the per-wave rows over the first out16[15] rows (the waves of the last launch)
    const size_t rows = (size_t)out16[15];
    std::vector<uint32_t> h(rows * 8);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fast_stamps), h.size() * 4));
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (size_t r = 0; r < rows; ++r) for (int i = 0; i < 8; ++i) out16[i] += h[r * 8 + i];
    out16[15] = rows;
    return ORBX_OK;
}
#endif

int orbx_extractor_create(const orbx_params* p, int width, int height, int max_batch, int device, orbx_extractor** out)
{
    if (!p || !out || max_batch < 1 || max_batch > 65535) { set_error("orbx_extractor_create: bad argument (max_batch must be 1..65535)"); return ORBX_ERR_BAD_ARG; }
    *out = nullptr;
    Geometry geo;
    int rc = build_geometry(*p, width, height, geo);
    if (rc) return rc;
    if (geo.max_cell_dim > kMaxCellDim) { set_error("cell larger than 60 pixels"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
    rc = select_device(device);
    if (rc) return rc;
    orbx_extractor* ex = new orbx_extractor();
    struct Guard { orbx_extractor* e; ~Guard() { orbx_extractor_destroy(e); } } guard{ex};    // a failed step releases what was allocated
    ex->params = *p; ex->geo = geo; ex->max_batch = max_batch;
    HIP_CHECK(hipGetDevice(&ex->device));
    ex->hlv.resize(geo.nlevels);
    int key_off = 0, ncmax = 0;
    std::vector<std::tuple<int, int, int> > pending_tc;      // per level: candidates of a busy image (one pixel in 64), count-table cells it would use
    for (int l = 0; l < geo.nlevels; ++l) {
        const LevelGeom& g = geo.lv[l];
        DevLevel d{};
        d.w = g.w; d.h = g.h; d.stride = g.stride; d.rows = g.rows; d.offset = g.offset; d.roi = g.roi;
        d.maxBorderX = g.maxBorderX; d.maxBorderY = g.maxBorderY;
        d.cell_first = g.cell_first; d.cell_count = g.cell_count; d.cand_offset = g.cand_offset; d.cand_cap = g.cand_cap;
        d.quota = g.quota; d.nIni = g.nIni; d.hX = g.hX; d.kp_offset = g.kp_offset; d.kp_cap = g.kp_cap;
        d.area2x = g.area2x; d.rx_offset = g.rx_offset; d.ry_offset = g.ry_offset; d.rb_offset = g.rb_offset; d.scale = g.scale; d.inv_scale = geo.tables.inv_scale[l]; d.kp_size = g.kp_size;
        d.key_offset = key_off;
        d.fast_resize = g.fast_resize;
        if (l > 0 && !g.fast_resize) ++ex->generic_levels;
        d.rg_offset = g.rg_offset; d.gpr = g.gpr; d.gpr_magic = g.gpr_magic;
        d.tpr = g.tpr; d.upr = g.upr; d.upr_magic = g.upr_magic; d.blur_offset = g.blur_offset;
        d.cpr = g.cpr; d.cpr_magic = g.cpr_magic;
        d.border_threads = border_threads(g.w, g.h, g.stride, kEdge, 8, 1 << 20);
        {
            int max_cap = 0;
            for (int i = 0; i < g.cell_count; ++i) max_cap = std::max(max_cap, (int)geo.cells[g.cell_first + i].cap);
            d.direct_ok = g.cell_count < (1 << (24 - kDirectCellBits)) && max_cap <= (1 << kDirectCellBits) && !getenv("ORBX_OCT_NO_DIRECT");
        }
        ex->max_border_threads = std::max(ex->max_border_threads, border_threads(g.w, g.h, g.stride, kFrameFill, 1, 2));
        ex->max_blur_threads = std::max(ex->max_blur_threads, 4 * g.upr * ((g.h + kBlurRows - 1) / kBlurRows));
        key_off += (g.cand_cap + 3) & ~3;
        ncmax = std::max(ncmax, g.kp_cap);
        pending_tc.push_back(g.cell_count > 0 ? std::make_tuple(g.w * g.h / 64, ot_table_cells(g.quota, g.nIni), g.maxBorderX - kMinBorder + g.maxBorderY - kMinBorder) : std::make_tuple(0, 0, 0));
        ex->hlv[l] = d;
    }
    ex->key_total = key_off;
    ex->NCmax = std::max(ncmax + 8, 64);
    if (ex->NCmax >= 65535) { set_error("a level's quota of 65527 keypoints or more overflows the quadtree's 16-bit node ids"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
    // Workgroup size of the quadtree kernel: its key passes scale with the candidates of a level (~1 % of the pixels).
    // Measured per batch: 640x480 256 threads 0.10 ms, 512 +15 %, 1024 2x; 3840x2160 256 threads 2.14 ms, 512 1.19, 1024 0.75.
    {
        const long px = (long)width * height;
        ex->oct_threads = px >= 2000000 ? 1024 : px >= 1000000 ? 512 : 256;
    }
    // count tables: room for the levels that a busy image gives enough keys per thread to use them (a level that gets
    // there against this estimate runs with the room the others left, or walks the keys in every sweep as before)
    for (const auto& t : pending_tc)
        if (std::get<0>(t) >= kOtTableMinKeysPerThread * ex->oct_threads) {
            ex->TC = std::max(ex->TC, std::get<1>(t));
            ex->XY = std::max(ex->XY, (std::get<2>(t) + 7) & ~7);
        }
    // ... as long as the node tables stay in LDS with them
    if (ot_lds_bytes(ex->NCmax, kOctThreads, 0, ex->TC, ex->XY) > kOctLdsLimit) { ex->TC = 0; ex->XY = 0; }
    // LDS key arrays: as many keys as leave four workgroups resident per CU (160 KB / 4 each).  Measured per 256 frames
    // with the count tables in place, budgets of 23 / 26 / 31 / 40 / 52 KB: 640x480 98 / 97 / 95 / 87 / 115 us (at 40 KB
    // the largest level's keys just fit), 480x640 92 -> 86, 752x480 101 -> 103, 1241x376 and 1920x1080 unchanged (their
    // node tables leave no room for keys either way).  Below 512 keys the LDS arrays are not worth having.
    {
        const long room = 40L * 1024 - (long)ot_lds_bytes(ex->NCmax, kOctThreads, 0, ex->TC, ex->XY);
        ex->KL = (int)std::max(0L, room / 6) / 128 * 128;
        if (ex->KL < 512) ex->KL = 0;
    }
    ex->oct_lds = ot_lds_bytes(ex->NCmax, kOctThreads, ex->KL, ex->TC, ex->XY);
    if (ex->oct_lds > kOctLdsLimit) {        // node tables beyond LDS: per-workgroup slices of device memory, keys there too
        ex->KL = 0;
        ex->oct_gstride = (ot_lds_bytes(ex->NCmax, kOctThreads, 0, ex->TC, ex->XY) + 255) & ~(size_t)255;
        ex->oct_lds = 64;
        HIP_CHECK(hipMalloc(&ex->d_octmem, ex->oct_gstride * geo.nlevels * (size_t)max_batch));
    }
    HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_octree<false>), ex->oct_lds));
    {
        // own blur when blurring every keypoint's 43 x 37 + 37 x 37 window costs less than blurring every level.  Measured
        // (us per launch, blurred levels -> own blur): a window costs about as much as 950 level pixels -- 1241x376 / 2000
        // features, 700 pixels per keypoint slot: 1221 -> 1284; 752x480 / 1200, 890: 887 -> 882; 640x480 / 1000, 940: 780 ->
        // 773 (four contexts: no difference); 1920x1080 / 2000, 3300: 1147 -> 989; 3840x2160 / 8000, 3300: 2360 -> 2043,
        // 14.3 k -> 17 k frames/s on four contexts.  ORBX_OWN_BLUR=0 / 1 forces a form (tests, A/B runs).
        long pyr_px = 0;
        for (int l = 0; l < geo.nlevels; ++l) pyr_px += (long)geo.lv[l].w * geo.lv[l].h;
        ex->own_blur = (long)geo.kp_total * 1200 < pyr_px;
        if (const char* e = getenv("ORBX_OWN_BLUR")) ex->own_blur = atoi(e) != 0;
    }
    HIP_CHECK(hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&ex->ev_done, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ex->ev_fork, hipEventDisableTiming));
    const size_t B = (size_t)max_batch;
    HIP_CHECK(hipMalloc(&ex->d_lv, sizeof(DevLevel) * geo.nlevels));
    HIP_CHECK(hipMalloc(&ex->d_cells, sizeof(CellRec) * std::max<size_t>(geo.cells.size(), 1)));
    HIP_CHECK(hipMalloc(&ex->d_rx, sizeof(ResizeX) * std::max<size_t>(geo.rx.size(), 1)));
    HIP_CHECK(hipMalloc(&ex->d_ry, sizeof(ResizeY) * std::max<size_t>(geo.ry.size(), 1)));
    HIP_CHECK(hipMalloc(&ex->d_rrb, sizeof(ResizeRowBlock) * std::max<size_t>(geo.rrb.size(), 1)));
    HIP_CHECK(hipMemcpy(ex->d_lv, ex->hlv.data(), sizeof(DevLevel) * geo.nlevels, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(ex->d_cells, geo.cells.data(), sizeof(CellRec) * geo.cells.size(), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> small, big;
        for (size_t i = 0; i < geo.cells.size(); ++i)
            (std::max(geo.cells[i].cw, geo.cells[i].ch) <= kSmallCellDim ? small : big).push_back((int32_t)i);
        ex->n_small = (int)small.size(); ex->n_big = (int)big.size();
        small.insert(small.end(), big.begin(), big.end());
        std::vector<FastCell> fc(std::max<size_t>(small.size(), 1));
        for (size_t i = 0; i < small.size(); ++i) {
            const CellRec& c = geo.cells[small[i]];
            const LevelGeom& g = geo.lv[c.level];
            const size_t src = g.roi + (size_t)c.y0 * g.stride + c.x0;
            if (src + (size_t)(c.ch + 6) * g.stride > 0xFFFFFFFFull) { set_error("pyramid of one frame beyond 4 GiB"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
            FastCell& f = fc[i];
            f.src = (uint32_t)src; f.stride = (uint32_t)g.stride;
            f.cw = (uint16_t)c.cw; f.ch = (uint16_t)c.ch; f.offx = (uint16_t)c.offx; f.offy = (uint16_t)c.offy;
            f.out = (uint32_t)(g.cand_offset + c.slot); f.cap = (uint32_t)c.cap; f.cell = (uint32_t)small[i]; f.pad = 0;
        }
        HIP_CHECK(hipMalloc(&ex->d_fast_cells, sizeof(FastCell) * fc.size()));
        HIP_CHECK(hipMemcpy(ex->d_fast_cells, fc.data(), sizeof(FastCell) * fc.size(), hipMemcpyHostToDevice));
    }
    {
        // every per-frame grid puts its chunk count into gridDim.y (at most 65535): refuse here, by name, what a launch
        // would otherwise refuse later with a bare HIP error
        long worst = 0; const char* which = "";
        auto see = [&](long chunks, const char* name) { if (chunks > worst) { worst = chunks; which = name; } };
        see(((long)ex->hlv[0].cpr * ex->hlv[0].h + 255) / 256, "level-0 copy");
        for (int l = 1; l < geo.nlevels; ++l)
            see(((long)ex->hlv[l].gpr * ((ex->hlv[l].h + kResizeRows - 1) / kResizeRows) + 255) / 256
                + (l == 1 ? ((long)ex->hlv[0].h + 6 * ex->hlv[0].cpr + 255) / 256 : 0), "resize");
        see((ex->n_small + kFastWavesPerBlock - 1) / kFastWavesPerBlock, "FAST cells");
        see((ex->n_big + kFastWavesPerBlock - 1) / kFastWavesPerBlock, "FAST cells");
        { long nb = 0; for (int l = 0; l < geo.nlevels; ++l) nb += (ex->hlv[l].upr * ((ex->hlv[l].h + kBlurRows - 1) / kBlurRows) + 63) / 64; see(nb, "blur"); }
        see((geo.kp_total + 4 * kKpPerWave - 1) / (4 * kKpPerWave), "describe");
        if (worst > 65535) {
            set_error(std::string("image too large for this parameter set: the ") + which + " stage needs " + std::to_string(worst) +
                      " workgroups per frame, the launch grid holds 65535");
            return ORBX_ERR_UNSUPPORTED_GEOMETRY;
        }
    }
    if (!geo.rx.empty()) HIP_CHECK(hipMemcpy(ex->d_rx, geo.rx.data(), sizeof(ResizeX) * geo.rx.size(), hipMemcpyHostToDevice));
    if (!geo.ry.empty()) HIP_CHECK(hipMemcpy(ex->d_ry, geo.ry.data(), sizeof(ResizeY) * geo.ry.size(), hipMemcpyHostToDevice));
    if (!geo.rrb.empty()) HIP_CHECK(hipMemcpy(ex->d_rrb, geo.rrb.data(), sizeof(ResizeRowBlock) * geo.rrb.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&ex->d_rga, sizeof(ResizeGroupA) * std::max<size_t>(geo.rga.size(), 1)));
    HIP_CHECK(hipMalloc(&ex->d_rgb, sizeof(ResizeGroupB) * std::max<size_t>(geo.rgb.size(), 1)));
    if (!geo.rga.empty()) HIP_CHECK(hipMemcpy(ex->d_rga, geo.rga.data(), sizeof(ResizeGroupA) * geo.rga.size(), hipMemcpyHostToDevice));
    if (!geo.rgb.empty()) HIP_CHECK(hipMemcpy(ex->d_rgb, geo.rgb.data(), sizeof(ResizeGroupB) * geo.rgb.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&ex->d_pyr, geo.pyr_bytes * B + 256));
    HIP_CHECK(hipMemset(ex->d_pyr, 0, geo.pyr_bytes * B + 256));
    if (!ex->own_blur) {         // (an own-blur context gets its blurred levels when the debug download first asks for them)
        HIP_CHECK(hipMalloc(&ex->d_blur, geo.blur_bytes * B + 256));
        HIP_CHECK(hipMemset(ex->d_blur, 0, geo.blur_bytes * B + 256));
    }
    HIP_CHECK(hipMalloc(&ex->d_cand, sizeof(uint32_t) * (size_t)geo.cand_total * B));
    HIP_CHECK(hipMalloc(&ex->d_cellcnt, sizeof(uint16_t) * geo.cells.size() * B));
    HIP_CHECK(hipMalloc(&ex->d_keys, sizeof(uint32_t) * (size_t)ex->key_total * B));
    HIP_CHECK(hipMalloc(&ex->d_nodeof, sizeof(uint16_t) * (size_t)ex->key_total * B));
    HIP_CHECK(hipMalloc(&ex->d_lvlkp, sizeof(uint32_t) * (size_t)geo.kp_total * B));
    HIP_CHECK(hipMalloc(&ex->d_lvlcnt, sizeof(int32_t) * geo.nlevels * B));
    {
        std::vector<uint32_t> bc;
        for (int l = 0; l < geo.nlevels; ++l) {
            const int units = ex->hlv[l].upr * ((ex->hlv[l].h + kBlurRows - 1) / kBlurRows);
            for (int b = 0; b < (units + 63) / 64; ++b) bc.push_back(((uint32_t)l << 16) | (uint32_t)b);
            if ((units + 63) / 64 > 0xFFFF) { set_error("image too large: a level's blur needs more than 65535 workgroups"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
        }
        ex->n_blur_chunks = (int)bc.size();
        HIP_CHECK(hipMalloc(&ex->d_blur_chunks, sizeof(uint32_t) * std::max<size_t>(bc.size(), 1)));
        HIP_CHECK(hipMemcpy(ex->d_blur_chunks, bc.data(), sizeof(uint32_t) * bc.size(), hipMemcpyHostToDevice));
    }
    for (auto& e : ex->ev) HIP_CHECK(hipEventCreate(&e));
    ex->no_graph = getenv("ORBX_NO_GRAPH") != nullptr;
    guard.e = nullptr;
    *out = ex;
    return ORBX_OK;
}

void orbx_extractor_destroy(orbx_extractor* ex)
{
    if (!ex) return;
    (void)hipSetDevice(ex->device);
    if (ex->stream) (void)hipStreamSynchronize(ex->stream);
    void* ptrs[] = {ex->d_pts, ex->d_octmem, ex->d_lv, ex->d_cells, ex->d_fast_cells, ex->d_rx, ex->d_ry, ex->d_rrb, ex->d_rga, ex->d_rgb, ex->d_pyr, ex->d_blur, ex->d_cand, ex->d_cellcnt, ex->d_keys, ex->d_nodeof,
                    ex->d_lvlkp, ex->d_lvlcnt, ex->d_img, ex->d_okp, ex->d_odesc, ex->d_ocount, ex->d_blur_chunks};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    void* hptrs[] = {ex->h_pts, ex->h_img, ex->h_okp, ex->h_odesc, ex->h_ocount, ex->h_level};
    for (void* p : hptrs) if (p) (void)hipHostFree(p);
    for (auto& e : ex->ev) if (e) (void)hipEventDestroy(e);
    if (ex->graph1) (void)hipGraphExecDestroy(ex->graph1);
    for (hipEvent_t e : ex->chunk_ev) (void)hipEventDestroy(e);
    if (ex->ev_done) (void)hipEventDestroy(ex->ev_done);
    if (ex->ev_fork) (void)hipEventDestroy(ex->ev_fork);
    if (ex->sr_dev) (void)hipFree(ex->sr_dev);
    if (ex->sr_pin) (void)hipHostFree(ex->sr_pin);
    if (ex->copy_stream) (void)hipStreamDestroy(ex->copy_stream);
    if (ex->stream) (void)hipStreamDestroy(ex->stream);
    delete ex;
}

// All kernels of frames [f0, f0 + B) of a batch, on stream st.
static int launch_frames(orbx_extractor* ex, hipStream_t st, int f0, int B, bool prof, const uint8_t* d_images, ptrdiff_t row_stride,
                         ptrdiff_t frame_stride, orbx_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap_per_frame)
{
    const Geometry& G = ex->geo;
    // per-frame arrays of this range
    d_images += (ptrdiff_t)f0 * frame_stride;
    d_kps += (size_t)f0 * cap_per_frame; d_desc += (size_t)f0 * cap_per_frame * 32; d_counts += f0;
    uint8_t* const pyr = ex->d_pyr + (size_t)f0 * G.pyr_bytes;
    uint8_t* const blur = ex->d_blur ? ex->d_blur + (size_t)f0 * G.blur_bytes : nullptr;
    uint32_t* const cand = ex->d_cand + (size_t)f0 * G.cand_total;
    uint16_t* const cellcnt = ex->d_cellcnt + (size_t)f0 * G.cells.size();
    uint32_t* const keys = ex->d_keys + (size_t)f0 * ex->key_total;
    uint16_t* const nodeof = ex->d_nodeof + (size_t)f0 * ex->key_total;
    uint32_t* const lvlkp = ex->d_lvlkp + (size_t)f0 * G.kp_total;
    int32_t* const lvlcnt = ex->d_lvlcnt + (size_t)f0 * G.nlevels;
    if (prof) HIP_CHECK(hipEventRecord(ex->ev[0], st));
    {
        const DevLevel& L0 = ex->hlv[0];
        dim3 grid(frame_chunk_grid((L0.cpr * L0.h + 255) / 256, B));
        bool frame_on_resize = false;
        if (ex->channels == 1 && d_images == pyr + L0.roi && row_stride == (ptrdiff_t)L0.stride && frame_stride == (ptrdiff_t)G.pyr_bytes) {
            // the frames already lie where level 0 lives (orbx_extractor_input_view): no copy, only the REFLECT_101 frame,
            // which rides on the launch of level 1 when that is the fast resize kernel
            if (G.nlevels >= 2 && ex->hlv[1].fast_resize) frame_on_resize = true;
            else
                hipLaunchKernelGGL(k_level0_frame, frame_chunk_grid((unsigned)((L0.h + 6 * L0.cpr + 255) / 256), B), dim3(256), 0, st, pyr,
                                   (unsigned long long)G.pyr_bytes, L0, B);
        } else if (ex->channels == 1)
            hipLaunchKernelGGL(k_level0, grid, dim3(256), 0, st, d_images, (long long)row_stride, (long long)frame_stride,
                               pyr, (unsigned long long)G.pyr_bytes, L0, B);
        else if (ex->channels == 3)
            hipLaunchKernelGGL(k_level0_colour<3>, grid, dim3(256), 0, st, d_images, (long long)row_stride, (long long)frame_stride,
                               pyr, (unsigned long long)G.pyr_bytes, L0, B, ex->c0, ex->c1, ex->c2);
        else
            hipLaunchKernelGGL(k_level0_colour<4>, grid, dim3(256), 0, st, d_images, (long long)row_stride, (long long)frame_stride,
                               pyr, (unsigned long long)G.pyr_bytes, L0, B, ex->c0, ex->c1, ex->c2);
        for (int l = 1; l < G.nlevels; ++l) {
            const DevLevel& D = ex->hlv[l];
            if (D.fast_resize) {
                const unsigned rblocks = (unsigned)((D.gpr * ((D.h + kResizeRows - 1) / kResizeRows) + 255) / 256);
                const unsigned fblocks = l == 1 && frame_on_resize ? (unsigned)((L0.h + 6 * L0.cpr + 255) / 256) : 0u;
                dim3 g2(frame_chunk_grid(rblocks + fblocks, B));
                if (D.fast_resize == 2)
                    hipLaunchKernelGGL(k_resize<true>, g2, dim3(256), 0, st, pyr, (unsigned long long)G.pyr_bytes, ex->hlv[l - 1], D,
                                       ex->d_ry, ex->d_rrb, ex->d_rga, ex->d_rgb, B, rblocks);
                else
                    hipLaunchKernelGGL(k_resize<false>, g2, dim3(256), 0, st, pyr, (unsigned long long)G.pyr_bytes, ex->hlv[l - 1], D,
                                       ex->d_ry, ex->d_rrb, ex->d_rga, ex->d_rgb, B, rblocks);
            } else {
                dim3 g2((D.w + 255) / 256, (D.h + 3) / 4, B);
                hipLaunchKernelGGL(k_resize_generic, g2, dim3(256), 0, st, pyr, (unsigned long long)G.pyr_bytes, ex->hlv[l - 1], D,
                                   ex->d_rx, ex->d_ry);
            }
        }
        if (ex->generic_levels) {               // the generic resize writes no frame pixels
            dim3 gb((ex->max_border_threads + 255) / 256, G.nlevels, B);
            hipLaunchKernelGGL(k_borders, gb, dim3(256), 0, st, pyr, (unsigned long long)G.pyr_bytes, ex->d_lv, kFrameFill, 1, 2);
        }
    }
    if (prof) HIP_CHECK(hipEventRecord(ex->ev[1], st));
    {
        // the blurred levels are only read by k_describe, but the blur runs here, next to the pyramid it reads: with the
        // blur directly in front of it k_describe took 6 % longer (measured, 187 against 176 us per 256 frames)
        dim3 gblur(frame_chunk_grid((unsigned)ex->n_blur_chunks, B));
        if (ex->own_blur) {}
        else if (ex->params.blur_rounding)
            hipLaunchKernelGGL(k_blur<1>, gblur, dim3(256), 0, st, pyr, blur, (unsigned long long)G.pyr_bytes, (unsigned long long)G.blur_bytes, ex->d_lv, ex->d_blur_chunks, B);
        else
            hipLaunchKernelGGL(k_blur<0>, gblur, dim3(256), 0, st, pyr, blur, (unsigned long long)G.pyr_bytes, (unsigned long long)G.blur_bytes, ex->d_lv, ex->d_blur_chunks, B);
    }
    if (prof) HIP_CHECK(hipEventRecord(ex->ev[2], st));
    {
        // cells up to 38 pixels (the usual 30..38) run with small LDS tiles (5 KB per cell: 32 wavefronts per CU); the few larger
        // ones (a level whose border-to-border span is just under a multiple of 30) get their own launch
        const int nc = (int)G.cells.size();
        if (ex->n_small)
            hipLaunchKernelGGL((k_fast_cells<48, kSmallCellDim + 6, 40, kSmallCellDim + 2, 40, kSmallQueue, kFastWavesPerBlock>),
                               dim3(frame_chunk_grid((ex->n_small + kFastWavesPerBlock - 1) / kFastWavesPerBlock, B)), dim3(64 * kFastWavesPerBlock), 0, st, pyr, (unsigned long long)G.pyr_bytes, ex->d_fast_cells, ex->n_small, cand, G.cand_total, cellcnt, nc,
                               ex->params.ini_th_fast, ex->params.min_th_fast, B);
        if (ex->n_big)
            hipLaunchKernelGGL((k_fast_cells<80, kMaxCellDim + 8, 64, kMaxCellDim + 2, 64, kBigQueue, kFastWavesPerBlock>),
                               dim3(frame_chunk_grid((ex->n_big + kFastWavesPerBlock - 1) / kFastWavesPerBlock, B)), dim3(64 * kFastWavesPerBlock), 0, st, pyr, (unsigned long long)G.pyr_bytes, ex->d_fast_cells + ex->n_small, ex->n_big, cand, G.cand_total, cellcnt, nc,
                               ex->params.ini_th_fast, ex->params.min_th_fast, B);
    }
    if (prof) HIP_CHECK(hipEventRecord(ex->ev[3], st));
    {
        // frame is the fast grid index: consecutive workgroups go to different XCDs, so every XCD gets an equal
        // share of each level (a level-major order would put all level-0 quadtrees, the largest, on one XCD)
        dim3 grid(B, G.nlevels);
        if (ex->d_octmem)
            hipLaunchKernelGGL(k_octree<true>, grid, dim3(ex->oct_threads), ex->oct_lds, st, ex->d_lv, ex->d_cells, cand, G.cand_total,
                               cellcnt, (int)G.cells.size(), keys, nodeof, ex->key_total, lvlkp, G.kp_total, lvlcnt, G.nlevels, ex->NCmax, ex->KL, ex->TC, ex->XY,
                               ex->d_octmem + (size_t)f0 * G.nlevels * ex->oct_gstride, (unsigned long long)ex->oct_gstride);
        else
            hipLaunchKernelGGL(k_octree<false>, grid, dim3(ex->oct_threads), ex->oct_lds, st, ex->d_lv, ex->d_cells, cand, G.cand_total,
                               cellcnt, (int)G.cells.size(), keys, nodeof, ex->key_total, lvlkp, G.kp_total, lvlcnt, G.nlevels, ex->NCmax, ex->KL, ex->TC, ex->XY,
                               (unsigned char*)nullptr, 0ull);
    }
    if (prof) HIP_CHECK(hipEventRecord(ex->ev[4], st));
    {
        dim3 grid(frame_chunk_grid((G.kp_total + 4 * kKpPerWave - 1) / (4 * kKpPerWave), B));
#define ORBX_LAUNCH_DESCRIBE(OWN, MODE) hipLaunchKernelGGL((k_describe<OWN, MODE>), grid, dim3(256), 0, st, pyr, blur, (unsigned long long)G.pyr_bytes, (unsigned long long)G.blur_bytes, \
                ex->d_lv, G.nlevels, lvlkp, G.kp_total, lvlcnt, d_kps, d_desc, d_counts, cap_per_frame, B, ex->params.trig_mode)
        if (!ex->own_blur) ORBX_LAUNCH_DESCRIBE(false, 0);
        else if (ex->params.blur_rounding) ORBX_LAUNCH_DESCRIBE(true, 1);
        else ORBX_LAUNCH_DESCRIBE(true, 0);
#undef ORBX_LAUNCH_DESCRIBE
    }
    if (prof) HIP_CHECK(hipEventRecord(ex->ev[5], st));
    HIP_CHECK(hipGetLastError());
    // what reads this batch's pyramids or tables on another stream waits for this event (not recorded while the
    // single-frame sequence is being captured into its graph: that sequence only ever runs on the context's own stream)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    HIP_CHECK(hipStreamIsCapturing(st, &capturing));
    if (capturing == hipStreamCaptureStatusNone) HIP_CHECK(hipEventRecord(ex->ev_done, st));
    return ORBX_OK;
}

int orbx_extract_batch_device(orbx_extractor* ex, const uint8_t* d_images, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                              int n_frames, orbx_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap_per_frame,
                              void* stream)
{
    if (!ex || !d_images || !d_kps || !d_desc || !d_counts) { set_error("orbx_extract_batch_device: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (n_frames < 1 || n_frames > ex->max_batch) { set_error("n_frames outside [1, max_batch]"); return ORBX_ERR_BAD_ARG; }
    if (row_stride < (ptrdiff_t)ex->geo.width * ex->channels || cap_per_frame < 1) { set_error("row_stride < width * channels or cap_per_frame < 1"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(ex->device));
    const int rc = launch_frames(ex, (hipStream_t)stream, 0, n_frames, ex->profiling, d_images, row_stride, frame_stride, d_kps, d_desc,
                                 d_counts, cap_per_frame);
    if (rc == ORBX_OK) ex->last_frames = n_frames;
    return rc;
}

int orbx_extract_batch_device_multi(orbx_extractor* const* exs, int n_contexts, const uint8_t* d_images, ptrdiff_t row_stride,
                                    ptrdiff_t frame_stride, int n_frames, orbx_keypoint* d_kps, uint8_t* d_desc,
                                    int32_t* d_counts, int cap_per_frame, void* stream, int flags)
{
    if (!exs || n_contexts < 1 || !d_kps || !d_desc || !d_counts || n_frames < 1 || cap_per_frame < 1 || (flags & ~ORBX_MULTI_NO_JOIN)) {
        set_error("orbx_extract_batch_device_multi: bad argument"); return ORBX_ERR_BAD_ARG;
    }
    const int per = (n_frames + n_contexts - 1) / n_contexts;
    for (int c = 0; c < n_contexts; ++c) {
        orbx_extractor* ex = exs[c];
        if (!ex || ex->device != exs[0]->device || ex->geo.width != exs[0]->geo.width || ex->geo.height != exs[0]->geo.height || ex->channels != exs[0]->channels) {
            set_error("orbx_extract_batch_device_multi: the contexts must share device, image size and input format"); return ORBX_ERR_BAD_ARG;
        }
        if (std::min(per, n_frames - c * per) > ex->max_batch) { set_error("orbx_extract_batch_device_multi: a context's share exceeds its max_batch"); return ORBX_ERR_BAD_ARG; }
        if (d_images && row_stride < (ptrdiff_t)ex->geo.width * ex->channels) { set_error("row_stride < width * channels"); return ORBX_ERR_BAD_ARG; }
        if (!d_images && ex->channels != 1) { set_error("orbx_extract_batch_device_multi: the input views hold grey frames"); return ORBX_ERR_BAD_ARG; }
    }
    HIP_CHECK(hipSetDevice(exs[0]->device));
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    HIP_CHECK(hipStreamIsCapturing(st, &capturing));
    if (capturing != hipStreamCaptureStatusNone) { set_error("orbx_extract_batch_device_multi: not under stream capture (it orders its streams with events of its own)"); return ORBX_ERR_BAD_ARG; }
    // fork: what the caller's stream has queued so far (the producer of the frames) comes first on every context's stream
    HIP_CHECK(hipEventRecord(exs[0]->ev_fork, st));
    for (int c = 0; c < n_contexts; ++c) {
        const int f0 = c * per, nf = std::min(per, n_frames - f0);
        if (nf <= 0) break;
        orbx_extractor* ex = exs[c];
        HIP_CHECK(hipStreamWaitEvent(ex->stream, exs[0]->ev_fork, 0));
        const Geometry& G = ex->geo;
        const uint8_t* img = d_images ? d_images + (ptrdiff_t)f0 * frame_stride : ex->d_pyr + G.lv[0].roi;
        const ptrdiff_t rs = d_images ? row_stride : (ptrdiff_t)G.lv[0].stride, fs = d_images ? frame_stride : (ptrdiff_t)G.pyr_bytes;
        const int rc = launch_frames(ex, ex->stream, 0, nf, false, img, rs, fs, d_kps + (size_t)f0 * cap_per_frame,
                                     d_desc + (size_t)f0 * cap_per_frame * 32, d_counts + f0, cap_per_frame);
        if (rc) return rc;
        ex->last_frames = nf;
        // join: the caller's stream continues behind this context's last kernel
        if (!(flags & ORBX_MULTI_NO_JOIN)) HIP_CHECK(hipStreamWaitEvent(st, ex->ev_done, 0));
    }
    return ORBX_OK;
}

int orbx_extract_multi_join(orbx_extractor* const* exs, int n_contexts, void* stream)
{
    if (!exs || n_contexts < 1) { set_error("orbx_extract_multi_join: bad argument"); return ORBX_ERR_BAD_ARG; }
    for (int c = 0; c < n_contexts; ++c) if (!exs[c]) { set_error("orbx_extract_multi_join: null context"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(exs[0]->device));
    for (int c = 0; c < n_contexts; ++c) HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, exs[c]->ev_done, 0));
    return ORBX_OK;
}

int orbx_extractor_set_input_format(orbx_extractor* ex, int format)
{
    if (!ex || format < ORBX_INPUT_GRAY || format > ORBX_INPUT_BGRA) { set_error("orbx_extractor_set_input_format: bad argument"); return ORBX_ERR_BAD_ARG; }
    const int channels = format == ORBX_INPUT_GRAY ? 1 : (format == ORBX_INPUT_RGB || format == ORBX_INPUT_BGR ? 3 : 4);
    const bool blue_first = format == ORBX_INPUT_BGR || format == ORBX_INPUT_BGRA;
    HIP_CHECK(hipSetDevice(ex->device));
    HIP_CHECK(sync_context(ex));
    if (channels != ex->channels) {              // the image staging buffers are sized by the input layout
        if (ex->d_img) (void)hipFree(ex->d_img);
        if (ex->h_img) (void)hipHostFree(ex->h_img);
        ex->d_img = nullptr; ex->h_img = nullptr;
        for (void** p : {(void**)&ex->d_okp, (void**)&ex->d_odesc, (void**)&ex->d_ocount}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        for (void** p : {(void**)&ex->h_okp, (void**)&ex->h_odesc, (void**)&ex->h_ocount}) { if (*p) (void)hipHostFree(*p); *p = nullptr; }
    }
    if (ex->graph1) { (void)hipGraphExecDestroy(ex->graph1); ex->graph1 = nullptr; }   // captured with the old first kernel
    ex->channels = channels;
    ex->c0 = blue_first ? 1868 : 4899; ex->c1 = 9617; ex->c2 = blue_first ? 4899 : 1868;   // B2Y, G2Y, R2Y in Q14
    return ORBX_OK;
}

int orbx_undistort_keypoints_batch_device(orbx_extractor* ex, const orbx_camera* cam, const orbx_keypoint* d_kps, const int32_t* d_counts,
                                          int cap_per_frame, int n_frames, orbx_keypoint* d_kps_un, void* stream)
{
    if (!ex || !cam || !d_kps || !d_counts || !d_kps_un || cap_per_frame < 1 || n_frames < 0) { set_error("orbx_undistort_keypoints_batch_device: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (n_frames == 0) return ORBX_OK;
    if (!(cam->fx != 0.f) || !(cam->fy != 0.f)) { set_error("camera focal length is zero"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(ex->device));
    hipLaunchKernelGGL(k_undistort_keypoints, dim3((cap_per_frame + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream, d_kps, d_counts,
                       cap_per_frame, *cam, cam->k1 == 0.0f ? 1 : 0, d_kps_un);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_undistort_points(orbx_extractor* ex, const orbx_camera* cam, const float* xy, int n, float* xy_out)
{
    if (!ex || !cam || n < 0) { set_error("orbx_undistort_points: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (n == 0) return ORBX_OK;
    if (!xy || !xy_out) { set_error("orbx_undistort_points: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!(cam->fx != 0.f) || !(cam->fy != 0.f)) { set_error("camera focal length is zero"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(ex->device));
    if (ex->pts_cap < n) {
        if (ex->d_pts) (void)hipFree(ex->d_pts);
        if (ex->h_pts) (void)hipHostFree(ex->h_pts);
        ex->d_pts = nullptr; ex->h_pts = nullptr; ex->pts_cap = 0;
        const int cap = std::max(n, 4096);
        HIP_CHECK(hipMalloc(&ex->d_pts, sizeof(float) * 4 * (size_t)cap));
        HIP_CHECK(hipHostMalloc(&ex->h_pts, sizeof(float) * 4 * (size_t)cap, hipHostMallocDefault));
        ex->pts_cap = cap;
    }
    std::memcpy(ex->h_pts, xy, sizeof(float) * 2 * (size_t)n);
    HIP_CHECK(hipMemcpyAsync(ex->d_pts, ex->h_pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ex->stream));
    hipLaunchKernelGGL(k_undistort_points, dim3((n + 255) / 256), dim3(256), 0, ex->stream, reinterpret_cast<const float2*>(ex->d_pts), n, *cam,
                       reinterpret_cast<float2*>(ex->d_pts + 2 * (size_t)ex->pts_cap));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(ex->h_pts + 2 * (size_t)ex->pts_cap, ex->d_pts + 2 * (size_t)ex->pts_cap, sizeof(float) * 2 * (size_t)n,
                             hipMemcpyDeviceToHost, ex->stream));
    HIP_CHECK(sync_context(ex));
    std::memcpy(xy_out, ex->h_pts + 2 * (size_t)ex->pts_cap, sizeof(float) * 2 * (size_t)n);
    return ORBX_OK;
}

static int ensure_staging(orbx_extractor* ex)
{
    if (ex->d_img) return ORBX_OK;
    const size_t B = (size_t)ex->max_batch;
    const Geometry& G = ex->geo;
    HIP_CHECK(hipMalloc(&ex->d_img, (size_t)G.width * G.height * ex->channels * B + 16));
    HIP_CHECK(hipMalloc(&ex->d_okp, sizeof(orbx_keypoint) * (size_t)G.out_cap * B));
    HIP_CHECK(hipMalloc(&ex->d_odesc, (size_t)32 * G.out_cap * B));
    HIP_CHECK(hipMalloc(&ex->d_ocount, sizeof(int32_t) * B));
    // pinned mirrors: pageable 2-D copies cost milliseconds per frame, pinned 1-D copies tens of microseconds
    HIP_CHECK(hipHostMalloc(&ex->h_img, (size_t)G.width * G.height * ex->channels * B + 16, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(&ex->h_okp, sizeof(orbx_keypoint) * (size_t)G.out_cap * B, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(&ex->h_odesc, (size_t)32 * G.out_cap * B, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(&ex->h_ocount, sizeof(int32_t) * B, hipHostMallocDefault));
    return ORBX_OK;
}

int orbx_extract_batch(orbx_extractor* ex, const uint8_t* images, ptrdiff_t row_stride, ptrdiff_t frame_stride, int n_frames,
                       orbx_keypoint* kps, uint8_t* desc, int32_t* counts, int cap_per_frame)
{
    if (!ex || !images || !kps || !desc || !counts) { set_error("orbx_extract_batch: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (n_frames < 1 || n_frames > ex->max_batch) { set_error("n_frames outside [1, max_batch]"); return ORBX_ERR_BAD_ARG; }
    if (row_stride < (ptrdiff_t)ex->geo.width * ex->channels) { set_error("row_stride < width * channels"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(ex->device));
    int rc = ensure_staging(ex);
    if (rc) return rc;
    const Geometry& G = ex->geo;
    const size_t rbytes = (size_t)G.width * ex->channels, fbytes = rbytes * G.height;
    const bool pipelined = n_frames >= 48 && !ex->profiling;
    for (int f = 0; f < n_frames && !pipelined; ++f) {
        const uint8_t* src = images + (ptrdiff_t)f * frame_stride;
        uint8_t* dst = ex->h_img + f * fbytes;
        if ((size_t)row_stride == rbytes) std::memcpy(dst, src, fbytes);
        else for (int y = 0; y < G.height; ++y) std::memcpy(dst + (size_t)y * rbytes, src + (ptrdiff_t)y * row_stride, rbytes);
    }
    if (pipelined) {
        // many frames: chunks of up to kHostChunk frames flow through a pipeline: the host copies chunk c+1 into the pinned mirror
        // (several threads) while chunk c uploads on the copy stream and chunk c-1 runs its kernels and downloads its
        // output slots on the compute stream
        if (!ex->copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&ex->copy_stream, hipStreamNonBlocking));
        const int chunk = std::min(kHostChunk, std::max(16, (n_frames + 3) / 4));     // four stages when the batch allows, never tiny ones
        const int nchunks = (n_frames + chunk - 1) / chunk;
        while ((int)ex->chunk_ev.size() < nchunks) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ex->chunk_ev.push_back(e); }
        for (int c = 0; c < nchunks; ++c) {
            const int f0 = c * chunk, nf = std::min(chunk, n_frames - f0);
            parallel_rows(nf * G.height, [&](int r0, int r1) {
                for (int r = r0; r < r1; ++r) {
                    const int f = f0 + r / G.height, y = r % G.height;
                    std::memcpy(ex->h_img + (size_t)f * fbytes + (size_t)y * rbytes, images + (ptrdiff_t)f * frame_stride + (ptrdiff_t)y * row_stride, rbytes);
                }
            });
            HIP_CHECK(hipMemcpyAsync(ex->d_img + (size_t)f0 * fbytes, ex->h_img + (size_t)f0 * fbytes, fbytes * nf, hipMemcpyHostToDevice, ex->copy_stream));
            HIP_CHECK(hipEventRecord(ex->chunk_ev[c], ex->copy_stream));
            HIP_CHECK(hipStreamWaitEvent(ex->stream, ex->chunk_ev[c], 0));
            rc = launch_frames(ex, ex->stream, f0, nf, false, ex->d_img, (ptrdiff_t)rbytes, (ptrdiff_t)fbytes, ex->d_okp, ex->d_odesc, ex->d_ocount, G.out_cap);
            if (rc) return rc;
            HIP_CHECK(hipMemcpyAsync(ex->h_ocount + f0, ex->d_ocount + f0, sizeof(int32_t) * nf, hipMemcpyDeviceToHost, ex->stream));
            HIP_CHECK(hipMemcpyAsync(ex->h_okp + (size_t)f0 * G.out_cap, ex->d_okp + (size_t)f0 * G.out_cap, sizeof(orbx_keypoint) * (size_t)G.out_cap * nf,
                                     hipMemcpyDeviceToHost, ex->stream));
            HIP_CHECK(hipMemcpyAsync(ex->h_odesc + (size_t)f0 * G.out_cap * 32, ex->d_odesc + (size_t)f0 * G.out_cap * 32, (size_t)32 * G.out_cap * nf,
                                     hipMemcpyDeviceToHost, ex->stream));
        }
        ex->last_frames = n_frames;
        HIP_CHECK(sync_context(ex));
        int status = ORBX_OK;
        for (int f = 0; f < n_frames; ++f) {
            counts[f] = ex->h_ocount[f];
            if (counts[f] > cap_per_frame) { status = ORBX_ERR_CAPACITY; set_error("cap_per_frame smaller than the keypoint count"); }
        }
        parallel_rows(n_frames, [&](int a0, int a1) {
            for (int f = a0; f < a1; ++f) {
                const int n = std::min(ex->h_ocount[f], cap_per_frame);
                if (n <= 0) continue;
                std::memcpy(kps + (size_t)f * cap_per_frame, ex->h_okp + (size_t)f * G.out_cap, sizeof(orbx_keypoint) * n);
                std::memcpy(desc + (size_t)f * cap_per_frame * 32, ex->h_odesc + (size_t)f * G.out_cap * 32, (size_t)32 * n);
            }
        });
        return status;
    }
    // a few frames: one round trip (counts and whole output slots together); many frames: counts first, then what is used
    const bool one_trip = n_frames < 48;
    auto enqueue = [&]() -> int {
        HIP_CHECK(hipMemcpyAsync(ex->d_img, ex->h_img, fbytes * n_frames, hipMemcpyHostToDevice, ex->stream));
        const int r = launch_frames(ex, ex->stream, 0, n_frames, ex->profiling, ex->d_img, (ptrdiff_t)rbytes, (ptrdiff_t)fbytes, ex->d_okp, ex->d_odesc,
                                    ex->d_ocount, G.out_cap);
        if (r) return r;
        HIP_CHECK(hipMemcpyAsync(ex->h_ocount, ex->d_ocount, sizeof(int32_t) * n_frames, hipMemcpyDeviceToHost, ex->stream));
        if (one_trip) {
            HIP_CHECK(hipMemcpyAsync(ex->h_okp, ex->d_okp, sizeof(orbx_keypoint) * (size_t)G.out_cap * n_frames, hipMemcpyDeviceToHost, ex->stream));
            HIP_CHECK(hipMemcpyAsync(ex->h_odesc, ex->d_odesc, (size_t)32 * G.out_cap * n_frames, hipMemcpyDeviceToHost, ex->stream));
        }
        return ORBX_OK;
    };
    if (n_frames == 1 && !ex->profiling && !ex->no_graph) {
        // the per-frame call of the reference's callers: the whole sequence (upload, 13 kernels, download) is captured once
        // into a HIP graph and replayed, which removes most of the per-launch latency
        if (!ex->graph1) {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(ex->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue();
            const hipError_t e = hipStreamEndCapture(ex->stream, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            HIP_CHECK(e);
            const hipError_t ei = hipGraphInstantiate(&ex->graph1, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            HIP_CHECK(ei);
        }
        HIP_CHECK(hipGraphLaunch(ex->graph1, ex->stream));
        ex->last_frames = 1;
    } else {
        rc = enqueue();
        if (rc) return rc;
        ex->last_frames = n_frames;
    }
    HIP_CHECK(sync_context(ex));
    if (!one_trip) {
        for (int f = 0; f < n_frames; ++f) {
            const int n = std::min(ex->h_ocount[f], G.out_cap);
            if (n <= 0) continue;
            HIP_CHECK(hipMemcpyAsync(ex->h_okp + (size_t)f * G.out_cap, ex->d_okp + (size_t)f * G.out_cap, sizeof(orbx_keypoint) * n,
                                     hipMemcpyDeviceToHost, ex->stream));
            HIP_CHECK(hipMemcpyAsync(ex->h_odesc + (size_t)f * G.out_cap * 32, ex->d_odesc + (size_t)f * G.out_cap * 32, (size_t)32 * n,
                                     hipMemcpyDeviceToHost, ex->stream));
        }
        HIP_CHECK(sync_context(ex));
    }
    int status = ORBX_OK;
    for (int f = 0; f < n_frames; ++f) {
        int n = ex->h_ocount[f];
        counts[f] = n;
        if (n > cap_per_frame) { status = ORBX_ERR_CAPACITY; set_error("cap_per_frame smaller than the keypoint count"); n = cap_per_frame; }
        if (n > 0) {
            std::memcpy(kps + (size_t)f * cap_per_frame, ex->h_okp + (size_t)f * G.out_cap, sizeof(orbx_keypoint) * n);
            std::memcpy(desc + (size_t)f * cap_per_frame * 32, ex->h_odesc + (size_t)f * G.out_cap * 32, (size_t)32 * n);
        }
    }
    return status;
}

int orbx_extract(orbx_extractor* ex, const uint8_t* image, ptrdiff_t stride, orbx_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    if (n) *n = 0;
    if (!ex || !n) { set_error("orbx_extract: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!image) return ORBX_OK; // empty image: silent return, outputs untouched (src/ORBextractor.cc:1045-1046)
    int32_t count = 0;
    int rc = orbx_extract_batch(ex, image, stride, 0, 1, kps, desc, &count, cap);
    *n = count;
    return rc;
}

int orbx_extractor_input_view(orbx_extractor* ex, uint8_t** d_images, ptrdiff_t* row_stride, ptrdiff_t* frame_stride)
{
    if (!ex || !d_images || !row_stride || !frame_stride) { set_error("orbx_extractor_input_view: null pointer"); return ORBX_ERR_BAD_ARG; }
    const LevelGeom& L = ex->geo.lv[0];
    *d_images = ex->d_pyr + L.roi;
    *row_stride = (ptrdiff_t)L.stride;
    *frame_stride = (ptrdiff_t)ex->geo.pyr_bytes;
    return ORBX_OK;
}

int orbx_pyramid_level_device(orbx_extractor* ex, int frame, int level, const uint8_t** d_roi, int* w, int* h, ptrdiff_t* stride)
{
    if (!ex || frame < 0 || frame >= ex->max_batch || level < 0 || level >= ex->geo.nlevels) return ORBX_ERR_BAD_ARG;
    const LevelGeom& L = ex->geo.lv[level];
    if (d_roi) *d_roi = ex->d_pyr + (size_t)frame * ex->geo.pyr_bytes + L.roi;
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (stride) *stride = L.stride;
    return ORBX_OK;
}

int orbx_pyramid_level(orbx_extractor* ex, int frame, int level, uint8_t* dst, ptrdiff_t dst_stride)
{
    if (!ex || !dst || frame < 0 || frame >= ex->max_batch || level < 0 || level >= ex->geo.nlevels) return ORBX_ERR_BAD_ARG;
    const LevelGeom& L = ex->geo.lv[level];
    if (dst_stride < L.w + 2 * kEdge) return ORBX_ERR_BAD_ARG;
    HIP_CHECK(hipSetDevice(ex->device));
    HIP_CHECK(sync_context(ex));
    // the batch path fills only the 3 innermost frame pixels: complete this level's frame first
    hipLaunchKernelGGL(k_borders, dim3((ex->hlv[level].border_threads + 255) / 256), dim3(256), 0, ex->stream,
                       ex->d_pyr + (size_t)frame * ex->geo.pyr_bytes, 0ull, ex->d_lv + level, kEdge, 8, 1 << 20);
    HIP_CHECK(sync_context(ex));
    // one contiguous copy of the padded level into a pinned buffer, then the rows to the caller (a pageable 2-D copy
    // costs milliseconds)
    const size_t bytes = (size_t)L.stride * L.rows;
    if (ex->h_level_bytes < bytes) {
        if (ex->h_level) (void)hipHostFree(ex->h_level);
        ex->h_level = nullptr; ex->h_level_bytes = 0;
        HIP_CHECK(hipHostMalloc(&ex->h_level, bytes, hipHostMallocDefault));
        ex->h_level_bytes = bytes;
    }
    HIP_CHECK(hipMemcpyAsync(ex->h_level, ex->d_pyr + (size_t)frame * ex->geo.pyr_bytes + L.offset, bytes, hipMemcpyDeviceToHost, ex->stream));
    HIP_CHECK(sync_context(ex));
    for (int y = 0; y < L.rows; ++y)
        std::memcpy(dst + (size_t)y * dst_stride, ex->h_level + (size_t)y * L.stride + (kRoiX - kEdge), (size_t)L.w + 2 * kEdge);
    return ORBX_OK;
}

int orbx_pyramid_levels(orbx_extractor* ex, int frame, uint8_t* const* dst, const ptrdiff_t* dst_stride)
{
    if (!ex || !dst || !dst_stride || frame < 0 || frame >= ex->max_batch) return ORBX_ERR_BAD_ARG;
    const Geometry& G = ex->geo;
    for (int l = 0; l < G.nlevels; ++l)
        if (!dst[l] || dst_stride[l] < G.lv[l].w + 2 * kEdge) return ORBX_ERR_BAD_ARG;
    HIP_CHECK(hipSetDevice(ex->device));
    // complete the 19-pixel frames of every level in one launch, then one copy of the frame's whole pyramid block
    hipLaunchKernelGGL(k_borders, dim3((ex->hlv[0].border_threads + 255) / 256, G.nlevels, 1), dim3(256), 0, ex->stream,
                       ex->d_pyr + (size_t)frame * G.pyr_bytes, 0ull, ex->d_lv, kEdge, 8, 1 << 20);
    HIP_CHECK(hipGetLastError());
    if (ex->h_level_bytes < G.pyr_bytes) {
        if (ex->h_level) (void)hipHostFree(ex->h_level);
        ex->h_level = nullptr; ex->h_level_bytes = 0;
        HIP_CHECK(hipHostMalloc(&ex->h_level, G.pyr_bytes, hipHostMallocDefault));
        ex->h_level_bytes = G.pyr_bytes;
    }
    HIP_CHECK(hipMemcpyAsync(ex->h_level, ex->d_pyr + (size_t)frame * G.pyr_bytes, G.pyr_bytes, hipMemcpyDeviceToHost, ex->stream));
    HIP_CHECK(sync_context(ex));
    for (int l = 0; l < G.nlevels; ++l) {
        const LevelGeom& L = G.lv[l];
        for (int y = 0; y < L.rows; ++y)
            std::memcpy(dst[l] + (size_t)y * dst_stride[l], ex->h_level + L.offset + (size_t)y * L.stride + (kRoiX - kEdge), (size_t)L.w + 2 * kEdge);
    }
    return ORBX_OK;
}

int orbx_stereo_refine_batch_device(orbx_extractor* left, orbx_extractor* right,
                                    const orbx_keypoint* d_kl, const int32_t* d_nl, const orbx_keypoint* d_kr, int cap_per_frame,
                                    int n_pairs, const int32_t* d_best_dist, const int32_t* d_best_idx, float bf, float min_d,
                                    float max_d, float* d_uright, float* d_depth, int32_t* d_sad, void* stream)
{
    if (!left || !right || !d_kl || !d_nl || !d_kr || !d_best_dist || !d_best_idx || !d_uright || !d_depth || !d_sad) {
        set_error("orbx_stereo_refine_batch_device: null pointer"); return ORBX_ERR_BAD_ARG;
    }
    if (left->device != right->device || left->geo.width != right->geo.width || left->geo.height != right->geo.height ||
        left->geo.nlevels != right->geo.nlevels || left->params.scale_factor != right->params.scale_factor) {
        set_error("orbx_stereo_refine_batch_device: the two extractors must share device, image size and pyramid"); return ORBX_ERR_BAD_ARG;
    }
    if (n_pairs < 1 || n_pairs > left->max_batch || n_pairs > right->max_batch || cap_per_frame < 1 || cap_per_frame > 16000) {
        set_error("orbx_stereo_refine_batch_device: n_pairs outside the contexts' batch or cap_per_frame outside [1, 16000]"); return ORBX_ERR_BAD_ARG;
    }
    HIP_CHECK(hipSetDevice(left->device));
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipStreamWaitEvent(st, left->ev_done, 0));       // the pyramids of both contexts' last batches must be complete
    HIP_CHECK(hipStreamWaitEvent(st, right->ev_done, 0));
    hipLaunchKernelGGL(k_stereo_refine, dim3((cap_per_frame + 4 * kRefineGroup - 1) / (4 * kRefineGroup), n_pairs), dim3(256), 0, st, left->d_pyr, right->d_pyr,
                       (unsigned long long)left->geo.pyr_bytes, left->d_lv, left->geo.nlevels, d_kl, d_nl, d_kr, cap_per_frame,
                       d_best_dist, d_best_idx, bf, min_d, max_d, d_uright, d_depth, d_sad);
    hipLaunchKernelGGL(k_stereo_median, dim3((n_pairs + 3) / 4), dim3(256), 0, st, d_nl, cap_per_frame, d_sad, d_uright, d_depth, n_pairs);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_stereo_refine(orbx_extractor* left, orbx_extractor* right, const orbx_keypoint* kl, int nl, const orbx_keypoint* kr, int nr,
                       const int32_t* best_dist, const int32_t* best_idx, float bf, float min_d, float max_d,
                       float* uright, float* depth)
{
    if (!left || !right || nl < 0 || nr < 0) { set_error("orbx_stereo_refine: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (nl == 0) return ORBX_OK;
    if (!kl || !best_dist || !best_idx || !uright || !depth || (nr > 0 && !kr)) { set_error("orbx_stereo_refine: null pointer"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < nl; ++i)
        if (best_dist[i] >= 0 && (best_idx[i] < 0 || best_idx[i] >= nr)) { set_error("orbx_stereo_refine: best_idx out of range"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(left->device));
    const int cap = std::max(nl, std::max(nr, 1));
    // one pinned block up ([left keypoints | right keypoints | best_dist | best_idx | nl]), one device block
    // (the same, then sad | uright | depth), one block down; both grow with the frame and stay with the left context
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_kl = 0, o_kr = o_kl + up(sizeof(orbx_keypoint) * cap), o_bd = o_kr + up(sizeof(orbx_keypoint) * cap),
                 o_bi = o_bd + up(sizeof(int32_t) * cap), o_nl = o_bi + up(sizeof(int32_t) * cap), o_in_end = o_nl + 256,
                 o_sad = o_in_end, o_ur = o_sad + up(sizeof(int32_t) * cap), o_dp = o_ur + up(sizeof(float) * cap), o_end = o_dp + up(sizeof(float) * cap);
    if (left->sr_bytes < o_end) {
        if (left->sr_dev) (void)hipFree(left->sr_dev);
        if (left->sr_pin) (void)hipHostFree(left->sr_pin);
        left->sr_dev = nullptr; left->sr_pin = nullptr; left->sr_bytes = 0;
        HIP_CHECK(hipMalloc(&left->sr_dev, o_end));
        HIP_CHECK(hipHostMalloc((void**)&left->sr_pin, o_end, hipHostMallocDefault));
        left->sr_bytes = o_end;
    }
    uint8_t* h = left->sr_pin; uint8_t* d = static_cast<uint8_t*>(left->sr_dev);
    std::memcpy(h + o_kl, kl, sizeof(orbx_keypoint) * nl);
    if (nr) std::memcpy(h + o_kr, kr, sizeof(orbx_keypoint) * nr);
    std::memcpy(h + o_bd, best_dist, sizeof(int32_t) * nl);
    std::memcpy(h + o_bi, best_idx, sizeof(int32_t) * nl);
    std::memcpy(h + o_nl, &nl, sizeof(int32_t));
    hipStream_t st = left->stream;
    HIP_CHECK(hipMemcpyAsync(d, h, o_in_end, hipMemcpyHostToDevice, st));
    const int rc = orbx_stereo_refine_batch_device(left, right, reinterpret_cast<const orbx_keypoint*>(d + o_kl), reinterpret_cast<const int32_t*>(d + o_nl),
                                                   reinterpret_cast<const orbx_keypoint*>(d + o_kr), cap, 1, reinterpret_cast<const int32_t*>(d + o_bd),
                                                   reinterpret_cast<const int32_t*>(d + o_bi), bf, min_d, max_d, reinterpret_cast<float*>(d + o_ur),
                                                   reinterpret_cast<float*>(d + o_dp), reinterpret_cast<int32_t*>(d + o_sad), st);
    if (rc != ORBX_OK) return rc;
    HIP_CHECK(hipMemcpyAsync(h + o_ur, d + o_ur, o_end - o_ur, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    std::memcpy(uright, h + o_ur, sizeof(float) * nl);
    std::memcpy(depth, h + o_dp, sizeof(float) * nl);
    return ORBX_OK;
}

int orbx_debug_trig_range(int trig_mode, uint32_t first_bits, uint32_t count, float* d_cos, float* d_sin, void* stream)
{
    if (trig_mode < 0 || trig_mode > 2 || (count && (!d_cos || !d_sin))) { set_error("orbx_debug_trig_range: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (count > (1u << 30)) { set_error("orbx_debug_trig_range: at most 2^30 angles per call"); return ORBX_ERR_BAD_ARG; }
    if (!count) return ORBX_OK;
    hipLaunchKernelGGL(k_trig_range, dim3((count + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, trig_mode, first_bits, count, d_cos, d_sin);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_debug_blurred_level(orbx_extractor* ex, int frame, int level, uint8_t* dst, ptrdiff_t dst_stride)
{
    if (!ex || !dst || frame < 0 || frame >= ex->max_batch || level < 0 || level >= ex->geo.nlevels) return ORBX_ERR_BAD_ARG;
    const LevelGeom& L = ex->geo.lv[level];
    if (dst_stride < L.w) return ORBX_ERR_BAD_ARG;
    HIP_CHECK(hipSetDevice(ex->device));
    HIP_CHECK(sync_context(ex));
    if (ex->own_blur) {
        // this context's k_describe blurs its windows itself: make the blurred levels of the last batch now
        const Geometry& G = ex->geo;
        if (!ex->d_blur) {
            HIP_CHECK(hipMalloc(&ex->d_blur, G.blur_bytes * (size_t)ex->max_batch + 256));
            HIP_CHECK(hipMemset(ex->d_blur, 0, G.blur_bytes * (size_t)ex->max_batch + 256));
        }
        dim3 gblur(frame_chunk_grid((unsigned)ex->n_blur_chunks, ex->max_batch));
        if (ex->params.blur_rounding)
            hipLaunchKernelGGL(k_blur<1>, gblur, dim3(256), 0, ex->stream, ex->d_pyr, ex->d_blur, (unsigned long long)G.pyr_bytes, (unsigned long long)G.blur_bytes, ex->d_lv, ex->d_blur_chunks, ex->max_batch);
        else
            hipLaunchKernelGGL(k_blur<0>, gblur, dim3(256), 0, ex->stream, ex->d_pyr, ex->d_blur, (unsigned long long)G.pyr_bytes, (unsigned long long)G.blur_bytes, ex->d_lv, ex->d_blur_chunks, ex->max_batch);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ex->stream));
    }
    // the device copy is tiled (8 rows x 16 bytes per 128-byte tile, tpr tiles per tile row)
    const size_t bytes = (size_t)((L.h + 7) / 8) * L.tpr * 128;
    std::vector<uint8_t> tiled(bytes);
    HIP_CHECK(hipMemcpy(tiled.data(), ex->d_blur + (size_t)frame * ex->geo.blur_bytes + L.blur_offset, bytes, hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; ++y)
        for (int x = 0; x < L.w; ++x)
            dst[(size_t)y * dst_stride + x] = tiled[((size_t)(y >> 3) * L.tpr + (x >> 4)) * 128 + 16 * (y & 7) + (x & 15)];
    return ORBX_OK;
}

int orbx_debug_candidates(orbx_extractor* ex, int frame, int level, uint32_t* packed, int cap, int* n)
{
    if (!ex || !n || frame < 0 || frame >= ex->max_batch || level < 0 || level >= ex->geo.nlevels) return ORBX_ERR_BAD_ARG;
    HIP_CHECK(hipSetDevice(ex->device));
    HIP_CHECK(sync_context(ex));
    const Geometry& G = ex->geo;
    const LevelGeom& L = G.lv[level];
    std::vector<uint16_t> cnt(L.cell_count);
    std::vector<uint32_t> slots(std::max(L.cand_cap, 1));
    HIP_CHECK(hipMemcpy(cnt.data(), ex->d_cellcnt + (size_t)frame * G.cells.size() + L.cell_first, sizeof(uint16_t) * L.cell_count, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(slots.data(), ex->d_cand + (size_t)frame * G.cand_total + L.cand_offset, sizeof(uint32_t) * L.cand_cap, hipMemcpyDeviceToHost));
    int total = 0;
    for (int c = 0; c < L.cell_count; ++c) {
        const CellRec& cr = G.cells[L.cell_first + c];
        for (int j = 0; j < cnt[c]; ++j) {
            if (total < cap) packed[total] = slots[cr.slot + j];
            ++total;
        }
    }
    *n = total;
    return total > cap ? ORBX_ERR_CAPACITY : ORBX_OK;
}

int orbx_debug_level_keypoints(orbx_extractor* ex, int frame, int level, uint32_t* packed, int cap, int* n)
{
    if (!ex || !n || frame < 0 || frame >= ex->max_batch || level < 0 || level >= ex->geo.nlevels) return ORBX_ERR_BAD_ARG;
    HIP_CHECK(hipSetDevice(ex->device));
    HIP_CHECK(sync_context(ex));
    const Geometry& G = ex->geo;
    const LevelGeom& L = G.lv[level];
    int32_t cnt = 0;
    HIP_CHECK(hipMemcpy(&cnt, ex->d_lvlcnt + (size_t)frame * G.nlevels + level, sizeof(int32_t), hipMemcpyDeviceToHost));
    *n = cnt;
    if (cnt > cap) return ORBX_ERR_CAPACITY;
    if (cnt > 0) HIP_CHECK(hipMemcpy(packed, ex->d_lvlkp + (size_t)frame * G.kp_total + L.kp_offset, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

int orbx_distribute_octree(int device, const uint32_t* packed_keys, int n, int minX, int maxX, int minY, int maxY, int N,
                           uint32_t* out, int cap, int* n_out)
{
    return orbx_distribute_octree_ex(device, packed_keys, n, minX, maxX, minY, maxY, N, out, cap, n_out, 0);
}

int orbx_distribute_octree_ex(int device, const uint32_t* packed_keys, int n, int minX, int maxX, int minY, int maxY, int N,
                              uint32_t* out, int cap, int* n_out, int key_storage)
{
    if (!n_out || n < 0 || (n > 0 && !packed_keys) || maxX <= minX || maxY <= minY || N < 0 || n >= (1 << 24)) { set_error("orbx_distribute_octree: bad argument"); return ORBX_ERR_BAD_ARG; }
    const int W = maxX - minX, H = maxY - minY;
    const int nIni = (int)std::round((float)W / H);
    if (nIni <= 0 || nIni > kMaxRoots || W >= 4096 || H >= 4096) { set_error("orbx_distribute_octree: unsupported box"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
    const float hX = (float)W / nIni;
    const int NC = std::max(std::max(N + 3, 4 * nIni) + 9, 64);
    if (NC >= 65535) { set_error("orbx_distribute_octree: N of 65523 or more overflows the 16-bit node ids"); return ORBX_ERR_UNSUPPORTED_GEOMETRY; }
    int KL = key_storage == 0 ? 4096 : 0;
    int TC = ot_table_cells(N, nIni), XY = (W + H + 7) & ~7;
    if (ot_lds_bytes(NC, kOctThreads, 0, TC, XY) > kOctLdsLimit) { TC = 0; XY = 0; }
    size_t lds = ot_lds_bytes(NC, kOctThreads, KL, TC, XY);
    if (lds > kOctLdsLimit && KL) { KL = 0; lds = ot_lds_bytes(NC, kOctThreads, 0, TC, XY); }
    const bool in_device_memory = lds > kOctLdsLimit;       // node tables beyond LDS
    if (in_device_memory) { KL = 0; lds = 64; }
    int rc = select_device(device);
    if (rc) return rc;
    uint32_t *d_keys = nullptr, *d_out = nullptr; uint16_t* d_no = nullptr; int32_t* d_n = nullptr; unsigned char* d_mem = nullptr;
    if (in_device_memory) HIP_CHECK(hipMalloc(&d_mem, ot_lds_bytes(NC, kOctThreads, 0, TC, XY)));
    HIP_CHECK(hipMalloc(&d_keys, sizeof(uint32_t) * std::max(n, 1)));
    HIP_CHECK(hipMalloc(&d_no, sizeof(uint16_t) * std::max(n, 1)));
    HIP_CHECK(hipMalloc(&d_out, sizeof(uint32_t) * NC));
    HIP_CHECK(hipMalloc(&d_n, sizeof(int32_t)));
    if (n) HIP_CHECK(hipMemcpy(d_keys, packed_keys, sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_octree_single), lds));
    hipLaunchKernelGGL(k_octree_single, dim3(1), dim3(256), lds, 0, d_keys, d_no, n, W, H, nIni, hX, N, d_out, NC, d_n, NC, KL, TC, XY, d_mem);
    HIP_CHECK(hipGetLastError());
    int32_t cnt = 0;
    HIP_CHECK(hipMemcpy(&cnt, d_n, sizeof(int32_t), hipMemcpyDeviceToHost));
    *n_out = cnt;
    int status = ORBX_OK;
    if (cnt > cap) { status = ORBX_ERR_CAPACITY; cnt = cap; }
    if (cnt > 0) HIP_CHECK(hipMemcpy(out, d_out, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
    (void)hipFree(d_keys); (void)hipFree(d_no); (void)hipFree(d_out); (void)hipFree(d_n); (void)hipFree(d_mem);
    return status;
}

int orbx_extractor_set_profiling(orbx_extractor* ex, int on)
{
    if (!ex) return ORBX_ERR_BAD_ARG;
    ex->profiling = on != 0;
    return ORBX_OK;
}

int orbx_extractor_stage_ms(orbx_extractor* ex, float* ms)
{
    if (!ex || !ms) return ORBX_ERR_BAD_ARG;
    if (!ex->profiling) { set_error("profiling is off"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(ex->device));
    HIP_CHECK(hipEventSynchronize(ex->ev[ORBX_STAGE_COUNT]));
    for (int s = 0; s < ORBX_STAGE_COUNT; ++s) HIP_CHECK(hipEventElapsedTime(&ms[s], ex->ev[s], ex->ev[s + 1]));
    return ORBX_OK;
}

} // extern "C"
