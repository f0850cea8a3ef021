// MI355X (gfx950) kernels of the descriptor matcher: the 256-bit Hamming loops of ORBmatcher
// (src/ORBmatcher.cc:1650-1666 DescriptorDistance and the best / second-best bookkeeping of the
// Search* routines) and of Frame::ComputeStereoMatches (src/Frame.cc:466-549).
//
// The windowed and row-band searches take distances as 8 x (v_xor_b32 + v_bcnt_u32_b32) per pair.  The brute-force
// search against a large map is the one dense contraction of the library (every query against every row): it runs on
// the matrix cores as an exact int8 product (k_best2_mfma below).
#include <hip/hip_runtime.h>
#include <mutex>
#include <cstring>
#include <cmath>
#include <vector>
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

constexpr int kIdxBits = 22;                       // row index inside a chunk
constexpr uint32_t kKeyInit = 256u << kIdxBits;    // distance 256, never beaten by a real row (strict '<')
constexpr int kBlock = 256;

struct Part { uint32_t best, second; };            // packed keys: distance << 22 | row-in-chunk

__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c)
{
    return max(min(a, b), min(max(a, b), c));      // folds to v_med3_u32
}

// Brute-force best / second-best on the matrix cores.  With a map row t as 256 bytes of 0 / 1 and a query q as 256 bytes of +1 (bit clear)
// / -1 (bit set), the integer dot product is popc(t) - 2 popc(q & t), so distance(q, t) = popc(q) + dot: an exact int8
// contraction, v_mfma_i32_16x16x64_i8 (4 per 16 rows x 16 queries, 16 cycles each: four times the pairs per clock of the
// xor / popcount form, which needs 19 vector instructions per 64 pairs).  A wavefront owns a run of map rows and kMfmaQT tiles of 16 queries (their +-1 bytes stay in
// registers); per tile of 16 rows each lane expands 16 bits of its row per k-step (nibble * 0x00204081 & 0x01010101) as the
// A operand.  The order of the 256 bit positions inside the contraction is free as long as both operands agree: lane group
// g = lane >> 4 takes bytes 8 g .. 8 g + 7 of its row or query (one 8-byte load), k-step s the s-th 16 bits of them.
// D has the query on the lane and 4 rows in the registers: the tile's smallest dot is compared with the lane's current
// second-best first, and only a tile that can change a lane's result walks its 4 keys (distance << 22 | row-in-run: the
// minimum over keys is the lowest distance and, on ties, the lowest row).  The four lane groups hold the same queries over different rows and merge at the end.
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kMfmaQT = 8;                         // query tiles of 16 per wavefront
constexpr int kMfmaQueries = 16 * kMfmaQT;         // queries per workgroup (its four waves share them over different rows)

__device__ __forceinline__ uint32_t bits4_to_bytes(uint32_t nib) { return __umul24(nib, 0x00204081u) & 0x01010101u; }   // nib < 16
template <int HALF> __device__ __forceinline__ v4i bits16_to_bytes(uint32_t w)      // the low (HALF 0) or high 16 bits of w
{
    v4i a;
    a.x = (int)bits4_to_bytes((w >> (16 * HALF)) & 15u);
    a.y = (int)bits4_to_bytes((w >> (16 * HALF + 4)) & 15u);
    a.z = (int)bits4_to_bytes((w >> (16 * HALF + 8)) & 15u);
    a.w = (int)bits4_to_bytes((w >> (16 * HALF + 12)) & 15u);
    return a;
}

// How data-dependent the search is: tiles of 16 rows x 16 queries whose keys were walked (counted in a scalar register,
// added once per wavefront), for orbx_debug_best2_tile_walks.
__device__ unsigned long long g_best2_walked;

// grid: (ceil(nchunks / 4), query tiles of kMfmaQueries); wave w of block x owns chunk 4 x + w
__global__ __launch_bounds__(kBlock, 2) void k_best2_mfma(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t, long long nt,
                                                          int rows_per_chunk, int nchunks, const orbx_best2* __restrict__ seed,
                                                          Part* __restrict__ parts)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, grp = lane >> 4;
    // byte -> its eight bits as bytes of 0 / 1, for the row expansion in the loop (two LDS reads per 16 bits instead of
    // twelve vector instructions: the vector pipe is what the loop has least of)
    __shared__ uint2 s_lut[256];
    s_lut[threadIdx.x] = make_uint2(bits4_to_bytes(threadIdx.x & 15u), bits4_to_bytes(threadIdx.x >> 4));
    __syncthreads();
    const int chunk = blockIdx.x * 4 + wave;
    if (chunk >= nchunks) return;
    const long long r0 = (long long)chunk * rows_per_chunk;
    const int nrows = (int)min((long long)rows_per_chunk, nt - r0);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut);

    v4i B[kMfmaQT][4];
    int pq[kMfmaQT], thr[kMfmaQT];
    uint32_t best[kMfmaQT], second[kMfmaQT];
#pragma unroll
    for (int qt = 0; qt < kMfmaQT; ++qt) {
        const int qi = min((int)blockIdx.y * kMfmaQueries + qt * 16 + n, nq - 1);
        const uint4 lo = *reinterpret_cast<const uint4*>(q + (size_t)qi * 8), hi = *reinterpret_cast<const uint4*>(q + (size_t)qi * 8 + 4);
        pq[qt] = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        const uint2 mine = *reinterpret_cast<const uint2*>(q + (size_t)qi * 8 + 2 * grp);
        const uint32_t w[2] = {mine.x, mine.y};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4i e = (s & 1) ? bits16_to_bytes<1>(w[s >> 1]) : bits16_to_bytes<0>(w[s >> 1]);
            // bit set -> -1 (0xFF), bit clear -> +1
            B[qt][s].x = (int)(((uint32_t)e.x * 0xFEu) | 0x01010101u); B[qt][s].y = (int)(((uint32_t)e.y * 0xFEu) | 0x01010101u);
            B[qt][s].z = (int)(((uint32_t)e.z * 0xFEu) | 0x01010101u); B[qt][s].w = (int)(((uint32_t)e.w * 0xFEu) | 0x01010101u);
        }
        // with a seed (the second-best distance over a sample of the rows) only rows that can still matter are ever walked
        thr[qt] = (seed ? seed[qi].second : 256) - pq[qt];
        best[qt] = kKeyInit; second[qt] = kKeyInit;
    }

    // rows: lane (n, grp) reads bytes 8 grp .. 8 grp + 7 of row tb + n, two tiles ahead of the tile being multiplied
    // (rows past the end of the run repeat its last row)
    const uint32_t* rows = t + (size_t)r0 * 8 + 2 * grp;
    auto row_at = [&](int r) __attribute__((always_inline)) { return *reinterpret_cast<const uint2*>(rows + (size_t)min(r, nrows - 1) * 8); };
    uint2 n1 = row_at(n), n2 = row_at(16 + n);
    int walked = 0;
    for (int tb = 0; tb < nrows; tb += 16) {
        const uint32_t w[2] = {n1.x, n1.y};
        n1 = n2;
        n2 = row_at(tb + 32 + n);
        v4i acc[kMfmaQT];
        const v4i zero = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t ws = w[s >> 1];
            const uint2 e0 = *reinterpret_cast<const uint2*>(lut + (s & 1 ? (ws >> 13) & 0x7F8u : (ws << 3) & 0x7F8u));
            const uint2 e1 = *reinterpret_cast<const uint2*>(lut + (s & 1 ? (ws >> 21) & 0x7F8u : (ws >> 5) & 0x7F8u));
            const v4i a = {(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
#pragma unroll
            for (int qt = 0; qt < kMfmaQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, B[qt][s], s == 0 ? zero : acc[qt], 0, 0, 0);
        }
        const bool tail = tb + 16 > nrows;
        const uint32_t rowbase = (uint32_t)tb + 4u * (uint32_t)grp;          // register j holds row rowbase + j
#pragma unroll
        for (int qt = 0; qt < kMfmaQT; ++qt) {
            const v4i& d = acc[qt];
            const int m = min(min(d[0], d[1]), min(d[2], d[3]));
            if (__any(m <= thr[qt]) || tail) {
                ++walked;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t r = rowbase + (uint32_t)j;
                    uint32_t key = ((uint32_t)(d[j] + pq[qt]) << kIdxBits) + r;
                    if (tail && r >= (uint32_t)nrows) key = kKeyInit;
                    second[qt] = med3u(best[qt], second[qt], key);
                    best[qt] = min(best[qt], key);
                }
                thr[qt] = min(thr[qt], (int)(second[qt] >> kIdxBits) - pq[qt]);
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < kMfmaQT; ++qt) {
        uint32_t b = best[qt], s2 = second[qt];
#pragma unroll
        for (int x = 16; x <= 32; x <<= 1) {
            const uint32_t ob = __shfl_xor(b, x, 64), os = __shfl_xor(s2, x, 64);
            s2 = min(max(b, ob), min(s2, os));
            b = min(b, ob);
        }
        const int qi = (int)blockIdx.y * kMfmaQueries + qt * 16 + n;
        if (grp == 0 && qi < nq) { Part p; p.best = b; p.second = s2; parts[(size_t)chunk * nq + qi] = p; }
    }
    if (lane == 0) atomicAdd(&g_best2_walked, (unsigned long long)walked);
}

// one thread per query: fold the per-chunk partial results in chunk (= index) order
__global__ __launch_bounds__(256) void k_best2_fold(const Part* __restrict__ parts, int nchunks, int nq, int rows_per_chunk,
                                                    long long index_base, orbx_best2* __restrict__ out)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int best = 256, second = 256;
    long long idx = -1;
    for (int c = 0; c < nchunks; ++c) {
        const Part p = parts[(size_t)c * nq + qi];
        const int b = (int)(p.best >> kIdxBits), s = (int)(p.second >> kIdxBits);
        if (b < best) {
            second = min(best, s);
            best = b;
            idx = index_base + (long long)c * rows_per_chunk + (long long)(p.best & ((1u << kIdxBits) - 1u));
        } else {
            second = min(second, b);
        }
        second = min(second, s);
    }
    orbx_best2 o;
    o.best = best; o.best_idx = (int32_t)idx; o.second = second; o.pad = 0;
    out[qi] = o;
}

// merge shard results (shard-major): lowest distance, then lowest global index; second = second
// smallest of the union, which is what one sequential pass over the whole set yields
__global__ __launch_bounds__(256) void k_best2_merge(const orbx_best2* __restrict__ parts, int n_parts, int nq, orbx_best2* __restrict__ out)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int best = 256, second = 256, idx = -1;
    for (int p = 0; p < n_parts; ++p) {
        const orbx_best2 r = parts[(size_t)p * nq + qi];
        if (r.best_idx < 0) continue;
        if (r.best < best || (r.best == best && r.best_idx < idx)) {
            second = min(second, best);
            best = r.best; idx = r.best_idx;
        } else {
            second = min(second, r.best);
        }
        second = min(second, r.second);
    }
    orbx_best2 o;
    o.best = best; o.best_idx = idx; o.second = second; o.pad = 0;
    out[qi] = o;
}

// CSR distances: one wavefront per query row
__global__ __launch_bounds__(256) void k_csr_dist(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ t,
                                                  const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cand,
                                                  int32_t* __restrict__ dist)
{
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    uint32_t qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = q[(size_t)qi * 8 + i];
    const int b = rowptr[qi], e = rowptr[qi + 1];
    for (int c = b + lane; c < e; c += 64) {
        const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)cand[c] * 8);
        const uint4 lo = tp[0], hi = tp[1];
        int d = __popc(qv[0] ^ lo.x) + __popc(qv[1] ^ lo.y) + __popc(qv[2] ^ lo.z) + __popc(qv[3] ^ lo.w) +
                __popc(qv[4] ^ hi.x) + __popc(qv[5] ^ hi.y) + __popc(qv[6] ^ hi.z) + __popc(qv[7] ^ hi.w);
        dist[c] = d;
    }
}

// Stereo row-band search (src/Frame.cc:476-549).  One workgroup per pair builds the reference's vRowIndices table
// (for every image row the right keypoints whose band of +-2*scale rows covers it; counts and offsets in LDS, the index
// lists in a device-memory slice of the workgroup), then one thread per left keypoint walks the list of its row only.
// The best candidate is the minimum of distance << 16 | index, which is what the reference's strict '<' over lists in
// ascending index order keeps, so the order in which the lists are filled does not matter.
// The index lists live in LDS when they fit the entries the launch reserved there (they do for the usual 1000-2000
// keypoints: a walk is a chain of dependent reads, and from device memory each link costs a round trip); a left
// keypoint's walk takes four list entries a turn, so that up to four descriptor fetches are in flight.
constexpr int kStereoThreads = 1024;
__global__ __launch_bounds__(kStereoThreads) void k_stereo(const orbx_keypoint* __restrict__ kl, const uint32_t* __restrict__ dl, const int32_t* __restrict__ nl_arr,
                                                const orbx_keypoint* __restrict__ kr, const uint32_t* __restrict__ dr, const int32_t* __restrict__ nr_arr,
                                                int cap, int nl_fixed, int nr_fixed,
                                                const float* __restrict__ scale, int n_rows, float min_d, float max_d,
                                                uint16_t* __restrict__ lists, unsigned long long list_stride, int lds_list_entries,
                                                int32_t* __restrict__ best_dist, int32_t* __restrict__ best_idx)
{
    extern __shared__ int32_t s_mem[];
    int32_t* s_start = s_mem;                      // n_rows + 1: first list entry of a row
    int32_t* s_fill = s_mem + (n_rows + 1);        // n_rows + 1: counts, then fill positions
    float* s_x = reinterpret_cast<float*>(s_mem + 2 * (n_rows + 1));      // right keypoint columns
    uint8_t* s_oct = reinterpret_cast<uint8_t*>(s_x + cap);               // right keypoint octaves
    uint16_t* s_list = reinterpret_cast<uint16_t*>(s_oct + ((cap + 15) & ~15));   // lds_list_entries index-list entries
    __shared__ int32_t s_part[256];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int pair = blockIdx.x;
    const int nl = nl_arr ? min(nl_arr[pair], cap) : nl_fixed;
    const int nr = nr_arr ? min(nr_arr[pair], cap) : nr_fixed;
    const size_t base = (size_t)pair * cap;
    uint16_t* list = lists + (unsigned long long)pair * list_stride;
    for (int i = tid; i <= n_rows; i += NT) s_fill[i] = 0;
    __syncthreads();
    // bands of the right keypoints (:483-493), clipped to the image rows
    for (int i = tid; i < nr; i += NT) {
        const orbx_keypoint k = kr[base + i];
        const float r = __fmul_rn(2.0f, scale[k.octave]);
        const int maxr = min((int)ceilf(__fadd_rn(k.y, r)), n_rows - 1), minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
        s_x[i] = k.x; s_oct[i] = (uint8_t)k.octave;
        for (int yi = minr; yi <= maxr; ++yi) atomicAdd(&s_fill[yi], 1);
    }
    __syncthreads();
    // exclusive prefix sum of the row counts: a contiguous chunk per thread, then a scan of the chunk totals
    const int chunk = (n_rows + 256) / 256;                       // the first 256 threads do the scan
    const int c0 = min(tid * chunk, n_rows + 1), c1 = min(c0 + chunk, n_rows + 1);
    int sum = 0;
    if (tid < 256) {
        for (int i = c0; i < c1; ++i) sum += s_fill[i];
        s_part[tid] = sum;
    }
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid < 256 && tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        if (tid < 256) s_part[tid] += v;
        __syncthreads();
    }
    if (tid < 256) {
        int run = s_part[tid] - sum;
        for (int i = c0; i < c1; ++i) { const int c = s_fill[i]; s_start[i] = run; s_fill[i] = run; run += c; }
    }
    __syncthreads();
    const int total_entries = s_start[n_rows];                      // (every thread reads the same word)
    auto fill_and_walk = [&](auto* lst) __attribute__((always_inline)) {
        for (int i = tid; i < nr; i += NT) {
            const orbx_keypoint k = kr[base + i];
            const float r = __fmul_rn(2.0f, scale[k.octave]);
            const int maxr = min((int)ceilf(__fadd_rn(k.y, r)), n_rows - 1), minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
            for (int yi = minr; yi <= maxr; ++yi) lst[atomicAdd(&s_fill[yi], 1)] = (uint16_t)i;
        }
        __threadfence_block();
        __syncthreads();
        for (int iL = tid; iL < nl; iL += NT) {
            const orbx_keypoint k = kl[base + iL];
            const int levelL = k.octave;
            const int row = (int)k.y;                                   // vRowIndices[vL], :509-511
            const float minU = __fsub_rn(k.x, max_d), maxU = __fsub_rn(k.x, min_d);
            const bool rowok = row >= 0 && row < n_rows;
            const int b = rowok ? s_start[row] : 0, e = rowok ? s_start[row + 1] : 0;
            const bool skipped = e == b || (maxU < 0);                  // :513-520
            uint32_t best = (100u << 16);                               // bestDist = TH_HIGH, bestIdxR = 0 (:522-523)
            if (!skipped) {
                uint32_t qv[8];
                { const uint4* qp = reinterpret_cast<const uint4*>(dl + (base + iL) * 8);
                  const uint4 a = qp[0], c = qp[1];
                  qv[0] = a.x; qv[1] = a.y; qv[2] = a.z; qv[3] = a.w; qv[4] = c.x; qv[5] = c.y; qv[6] = c.z; qv[7] = c.w; }
                for (int c = b; c < e; c += 4) {
                    int idx[4]; bool ok[4]; uint4 lo[4], hi[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool in = c + u < e;
                        const int i = in ? (int)lst[c + u] : 0;
                        const int o = s_oct[i];
                        const float uR = s_x[i];
                        idx[u] = i;
                        ok[u] = in && !(o < levelL - 1 || o > levelL + 1) && uR >= minU && uR <= maxU;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ok[u]) { const uint4* tp = reinterpret_cast<const uint4*>(dr + (base + idx[u]) * 8); lo[u] = tp[0]; hi[u] = tp[1]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ok[u]) {
                            const uint32_t d = __popc(qv[0] ^ lo[u].x) + __popc(qv[1] ^ lo[u].y) + __popc(qv[2] ^ lo[u].z) + __popc(qv[3] ^ lo[u].w) +
                                               __popc(qv[4] ^ hi[u].x) + __popc(qv[5] ^ hi[u].y) + __popc(qv[6] ^ hi[u].z) + __popc(qv[7] ^ hi[u].w);
                            best = min(best, (d << 16) | (uint32_t)idx[u]);
                        }
                }
            }
            best_dist[base + iL] = skipped ? -1 : (int)(best >> 16);
            best_idx[base + iL] = skipped ? -1 : (int)(best & 0xFFFF);
        }
    };
    if (total_entries <= lds_list_entries) fill_and_walk(s_list);
    else fill_and_walk(list);
}

// Windowed candidate generation on the device.  One wavefront per query walks the keypoints in grid order, 64 at a time:
// a keypoint is a candidate when it lies strictly inside the square window and passes the level filter (for keypoints that
// are in the grid this is exactly the set the reference's walk over the covered cells returns, and ballot compaction keeps
// the walk's order).  First launch counts, a one-workgroup scan turns counts into row offsets, second launch writes
// candidate indices and distances.
struct KpLite { float x, y; int32_t octave; };

__device__ __forceinline__ bool in_window(const orbx::WindowQuery& q, const KpLite& k)
{
    const bool check = q.minLevel > 0 || q.maxLevel >= 0;
    if (check && (k.octave < q.minLevel || (q.maxLevel >= 0 && k.octave > q.maxLevel))) return false;
    return fabsf(__fsub_rn(k.x, q.x)) < q.r && fabsf(__fsub_rn(k.y, q.y)) < q.r;
}

template <bool kFill>
__global__ __launch_bounds__(256) void k_window(const orbx::WindowQuery* __restrict__ qs, const uint32_t* __restrict__ qdesc, int nq,
                                                const KpLite* __restrict__ kp, const uint32_t* __restrict__ d2,
                                                const int32_t* __restrict__ order, int n_in_grid,
                                                int32_t* __restrict__ rowptr, int32_t* __restrict__ cand, int32_t* __restrict__ dist, int cap)
{
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const orbx::WindowQuery q = qs[qi];
    uint32_t qv[8];
    int base = 0;
    if (kFill) {
        base = rowptr[qi];
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = qdesc[(size_t)qi * 8 + i];
    }
    int count = 0;
    if (q.r >= 0.0f)
        for (int j0 = 0; j0 < n_in_grid; j0 += 64) {
            const int j = j0 + lane;
            int idx = -1;
            bool on = false;
            if (j < n_in_grid) { idx = order[j]; on = in_window(q, kp[idx]); }
            const unsigned long long mask = __ballot(on);
            if (kFill && on) {
                const int pos = base + count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (pos < cap) {
                    const uint4* tp = reinterpret_cast<const uint4*>(d2 + (size_t)idx * 8);
                    const uint4 lo = tp[0], hi = tp[1];
                    cand[pos] = idx;
                    dist[pos] = __popc(qv[0] ^ lo.x) + __popc(qv[1] ^ lo.y) + __popc(qv[2] ^ lo.z) + __popc(qv[3] ^ lo.w) +
                                __popc(qv[4] ^ hi.x) + __popc(qv[5] ^ hi.y) + __popc(qv[6] ^ hi.z) + __popc(qv[7] ^ hi.w);
                }
            }
            count += __popcll(mask);
        }
    if (!kFill && lane == 0) rowptr[qi] = count;
}

// exclusive prefix sum of nq counts in place, total in rowptr[nq]; one workgroup
__global__ __launch_bounds__(1024) void k_window_scan(int32_t* __restrict__ rowptr, int nq)
{
    __shared__ int32_t s_part[1024];
    const int tid = threadIdx.x;
    const int chunk = (nq + 1023) / 1024;
    const int c0 = min(tid * chunk, nq), c1 = min(c0 + chunk, nq);
    int sum = 0;
    for (int i = c0; i < c1; ++i) sum += rowptr[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int i = c0; i < c1; ++i) { const int c = rowptr[i]; rowptr[i] = run; run += c; }
    if (tid == 1023) rowptr[nq] = s_part[1023];
}

// rows a right keypoint's band can span: ceil(y + r) - floor(y - r) + 1 <= 2r + 3 with r = 2 * scale
// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:408-523) for a batch of frame pairs, every step on the device:
// one wavefront per pair.  The octave-0 keypoints of frame 2 (the only ones a level-0 window returns, :428) go to LDS
// with their descriptors; the queries (octave-0 keypoints of frame 1) are then walked IN ORDER, because each accepted
// match changes what later queries may take (vMatchedDistance / vnMatches21, :447, :466-473).  Per query the 64 lanes
// share the candidates: window test, Hamming distance, the `vMatchedDistance[i2] <= dist` skip; the best candidate is
// the minimum of distance << 32 | visiting order, the visiting order of Frame::GetFeaturesInArea being grid column,
// grid row, keypoint index (src/Frame.cc:350-376: strict `<` lets the first visited of equal distances win); the second
// best distance is the second smallest of the multiset, which is what the sequential update of :449-459 yields.
// Rotation histogram, ComputeThreeMaxima, the removal of the other bins and the update of vbPrevMatched follow.
// Per keypoint of capacity: 44 bytes that are written once (descriptor, position, visiting-order key: the frame part) and
// 9 bytes of state of the sequential pass.  Both live in LDS while they fit (kFrameInGlobal false); for larger
// capacities -- the reference's initialisation extractor runs at twice nFeatures, src/Tracking.cc:136-137 -- the frame
// part moves to a per-pair slice of a device-memory arena (served by the CU's L1 / L2) and LDS keeps the state only.
// ---------------------------------------------------------------------------------------------
constexpr int kHistoLength = 30;
constexpr int kThLow = 50;
constexpr int kInitFramePerKeypoint = 44, kInitStatePerKeypoint = 9;
constexpr size_t kMatchLdsLimit = 150 * 1024;
static size_t init_frame_bytes(int cap) { return ((size_t)cap * kInitFramePerKeypoint + 15) & ~(size_t)15; }
static size_t init_state_bytes(int cap) { return (size_t)cap * kInitStatePerKeypoint + 16; }

__device__ __forceinline__ int rotation_bin_dev(float a1, float a2)      // :476-482 (factor = 1/HISTO_LENGTH: bins 0..12 only)
{
    const float factor = 1.0f / kHistoLength;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == kHistoLength) bin = 0;
    return bin;
}

// inclusive prefix sum over the 64 lanes (all active): DPP shifts inside rows of 16, then the row totals
__device__ __forceinline__ int wave_scan_incl_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

// minimum over the 64 lanes (all active) as a wave-uniform value: the prefix pattern of the sum above with min, whose
// last lane then holds the minimum of all (six DPP steps instead of six LDS round trips through ds_bpermute)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, 0x111, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, 0x112, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, 0x114, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, 0x118, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, 0x142, 0xA, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, 0x143, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit keys: the high words first, then the low words of the lanes that hold the smallest high word
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mh = wave_min_u32(hi);
    const uint32_t ml = wave_min_u32(hi == mh ? lo : ~0u);
    return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ int wave_min_i32(int v)          // values >= 0 only
{
    return (int)wave_min_u32((uint32_t)v);
}

template <bool kFrameInGlobal>
__global__ __launch_bounds__(64) void k_init_match(const orbx_keypoint* __restrict__ k1, const uint32_t* __restrict__ d1, const int32_t* __restrict__ n1v,
                                                   const orbx_keypoint* __restrict__ k2, const uint32_t* __restrict__ d2, const int32_t* __restrict__ n2v,
                                                   int cap, float minX, float minY, float invW, float invH,
                                                   float* __restrict__ prev_xy, float window, float nnratio, int check_ori,
                                                   int32_t* __restrict__ matches12, int32_t* __restrict__ nmatches,
                                                   unsigned char* __restrict__ arena, unsigned long long arena_stride)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int pair = blockIdx.x;
    // frame 2, octave 0, in keypoint order: descriptor, position, visiting-order key (written once) ...
    const size_t frame_bytes = ((size_t)cap * kInitFramePerKeypoint + 15) & ~(size_t)15;
    unsigned char* fbase = kFrameInGlobal ? arena + (size_t)pair * arena_stride : smem;
    uint4* s_desc = reinterpret_cast<uint4*>(fbase);                                 // 2 x uint4 per entry
    float2* s_xy = reinterpret_cast<float2*>(fbase + (size_t)cap * 32);
    uint32_t* s_okey = reinterpret_cast<uint32_t*>(fbase + (size_t)cap * 40);        // cellx << 26 | celly << 20 | index
    // ... and the state of the sequential pass (always LDS); indices as 16 bits (the capacity is below 65535), 0xFFFF = none
    unsigned char* mbase = kFrameInGlobal ? smem : smem + frame_bytes;
    uint16_t* s_m21 = reinterpret_cast<uint16_t*>(mbase);                            // vnMatches21
    uint16_t* s_mdist = reinterpret_cast<uint16_t*>(mbase + (size_t)cap * 2);        // vMatchedDistance (0xFFFF = INT_MAX)
    uint16_t* s_qlist = reinterpret_cast<uint16_t*>(mbase + (size_t)cap * 4);        // octave-0 keypoints of frame 1
    uint16_t* s_m12 = reinterpret_cast<uint16_t*>(mbase + (size_t)cap * 6);          // vnMatches12
    uint8_t* s_bin = mbase + (size_t)cap * 8;                                        // histogram bin a query was pushed to (255: none)
    __shared__ int s_hist[kHistoLength];
    constexpr uint16_t kNone = 0xFFFFu;

    const size_t off = (size_t)pair * cap;
    k1 += off; k2 += off; d1 += off * 8; d2 += off * 8; prev_xy += off * 2; matches12 += off;
    const int n1 = min(n1v[pair], cap), n2 = min(n2v[pair], cap);

    // compaction of the octave-0 keypoints that lie in the 64 x 48 grid (Frame::PosInGrid, src/Frame.cc:382-392)
    int n0 = 0;
    for (int j0 = 0; j0 < n2; j0 += 64) {
        const int j = j0 + lane;
        bool on = false; float x = 0.f, y = 0.f; int px = 0, py = 0;
        if (j < n2) {
            const orbx_keypoint kp = k2[j];
            x = kp.x; y = kp.y;
            px = (int)roundf(__fmul_rn(__fsub_rn(x, minX), invW));
            py = (int)roundf(__fmul_rn(__fsub_rn(y, minY), invH));
            on = kp.octave == 0 && px >= 0 && px < 64 && py >= 0 && py < 48;
        }
        const unsigned long long m = __ballot(on);
        if (on) {
            const int pos = n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint4* tp = reinterpret_cast<const uint4*>(d2 + (size_t)j * 8);
            s_desc[2 * pos] = tp[0]; s_desc[2 * pos + 1] = tp[1];
            s_xy[pos] = make_float2(x, y);
            s_okey[pos] = ((uint32_t)px << 26) | ((uint32_t)py << 20) | (uint32_t)j;
            s_m21[pos] = kNone; s_mdist[pos] = 0xFFFFu;
        }
        n0 += __popcll(m);
    }
    int nq = 0;
    for (int i0 = 0; i0 < n1; i0 += 64) {
        const int i = i0 + lane;
        const bool on = i < n1 && k1[i].octave <= 0;                       // :424-426 skips level1 > 0
        const unsigned long long m = __ballot(on);
        if (on) s_qlist[nq + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
        nq += __popcll(m);
        if (i < n1) { s_m12[i] = kNone; s_bin[i] = 255; }
    }
    if (lane < kHistoLength) s_hist[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    int count = 0;
    // the next query's descriptor and window centre are fetched while the current one is matched
    uint32_t qd[8]; float qx = 0.f, qy = 0.f; int qi = -1;
    auto fetch = [&](int e) {
        if (e < nq) {
            qi = s_qlist[e];
            const uint4* tp = reinterpret_cast<const uint4*>(d1 + (size_t)qi * 8);
            const uint4 lo = tp[0], hi = tp[1];
            qd[0] = lo.x; qd[1] = lo.y; qd[2] = lo.z; qd[3] = lo.w; qd[4] = hi.x; qd[5] = hi.y; qd[6] = hi.z; qd[7] = hi.w;
            qx = prev_xy[2 * qi]; qy = prev_xy[2 * qi + 1];
        }
    };
    fetch(0);
    for (int e = 0; e < nq; ++e) {
        const int i1 = qi;
        uint32_t cd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cd[k] = qd[k];
        const float cx = qx, cy = qy;
        fetch(e + 1);
        unsigned long long best = ~0ull;                                   // distance << 32 | visiting order
        int second = INT_MAX, bestj = -1;
        for (int j0 = 0; j0 < n0; j0 += 64) {
            const int j = j0 + lane;
            if (j >= n0) continue;
            const float2 p = s_xy[j];
            if (!(fabsf(__fsub_rn(p.x, cx)) < window && fabsf(__fsub_rn(p.y, cy)) < window)) continue;     // src/Frame.cc:370-374
            const uint4 lo = s_desc[2 * j], hi = s_desc[2 * j + 1];
            const int dist = __popc(cd[0] ^ lo.x) + __popc(cd[1] ^ lo.y) + __popc(cd[2] ^ lo.z) + __popc(cd[3] ^ lo.w) +
                             __popc(cd[4] ^ hi.x) + __popc(cd[5] ^ hi.y) + __popc(cd[6] ^ hi.z) + __popc(cd[7] ^ hi.w);
            if ((int)s_mdist[j] <= dist) continue;                         // :447 (0xFFFF stands for INT_MAX)
            const unsigned long long key = ((unsigned long long)(uint32_t)dist << 32) | s_okey[j];
            if (key < best) { if (best != ~0ull) second = (int)(best >> 32); best = key; bestj = j; }
            else if (dist < second) second = dist;
        }
        const unsigned long long wbest = wave_min_u64(best);
        if (wbest == ~0ull) continue;                                      // no candidate passed
        const bool winner = best == wbest;                                 // keys are unique: one lane
        const int mine = winner ? second : (best == ~0ull ? INT_MAX : (int)(best >> 32));
        const int bestDist2 = wave_min_i32(mine);
        const int bestDist = (int)(wbest >> 32);
        if (bestDist <= kThLow && (float)bestDist < __fmul_rn((float)bestDist2, nnratio)) {       // :462-464
            const int wj = __builtin_amdgcn_readlane(bestj, __ffsll((long long)__ballot(winner)) - 1);
            const int i2 = (int)(s_okey[wj] & 0xFFFFFu);
            if (lane == 0) {
                const uint16_t prev = s_m21[wj];
                if (prev != kNone) s_m12[prev] = kNone;                    // :466-470
                s_m12[i1] = (uint16_t)i2; s_m21[wj] = (uint16_t)i1; s_mdist[wj] = (uint16_t)bestDist;
                if (check_ori) { const int bin = rotation_bin_dev(k1[i1].angle, k2[i2].angle); s_bin[i1] = (uint8_t)bin; ++s_hist[bin]; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    // rotation consistency (:489-515)
    int ind1 = -1, ind2 = -1, ind3 = -1;
    if (check_ori) {
        int m1 = 0, m2 = 0, m3 = 0;
        for (int i = 0; i < kHistoLength; ++i) {                           // ComputeThreeMaxima, :1604-1645 (uniform: every lane the same)
            const int sz = s_hist[i];
            if (sz > m1) { m3 = m2; m2 = m1; m1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > m2) { m3 = m2; m2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > m3) { m3 = sz; ind3 = i; }
        }
        if ((float)m2 < __fmul_rn(0.1f, (float)m1)) { ind2 = -1; ind3 = -1; }
        else if ((float)m3 < __fmul_rn(0.1f, (float)m1)) ind3 = -1;
    }
    for (int i0 = 0; i0 < n1; i0 += 64) {
        const int i = i0 + lane;
        int mt = -1;
        if (i < n1) {
            mt = s_m12[i] == kNone ? -1 : (int)s_m12[i];
            const int bin = s_bin[i];
            if (check_ori && bin != 255 && bin != ind1 && bin != ind2 && bin != ind3) mt = -1;
            matches12[i] = mt;
            if (mt >= 0) { prev_xy[2 * i] = k2[mt].x; prev_xy[2 * i + 1] = k2[mt].y; }         // :518-520
        }
        count += __popcll(__ballot(mt >= 0));
    }
    for (int i = n1 + lane; i < cap; i += 64) matches12[i] = -1;
    if (lane == 0) nmatches[pair] = count;
}

// ---------------------------------------------------------------------------------------------
// The projection searches of ORBmatcher for a batch of (frame, query list) calls, every step on the device: one
// wavefront per call.  FORM 0: SearchByProjection(Frame&, const Frame&, th, bMono) and the relocalisation form
// (src/ORBmatcher.cc:1381-1470, :1531-1598: best-1 within th_dist, rotation histogram).  FORM 1: SearchByProjection(Frame&,
// const vector<MapPoint*>&, th) (:68-125: best and second best with their levels, ratio test when the levels agree).
// The frame's features that lie in the 64 x 48 grid are bucketed by grid cell into LDS (counting sort; cells column by
// column, as Frame::mGrid is walked), each with descriptor, position, octave, angle, right coordinate, claim state and a
// visiting-order key (column, row, index: the order Frame::GetFeaturesInArea returns candidates in, which the strict `<`
// of the best-candidate loops turns into the tie-break).  The queries are walked IN ORDER (a claim changes what later
// queries may take); per query the cells its window covers (src/Frame.cc:332-346) form one run of LDS entries per grid
// column, the runs are flattened over the 64 lanes, and the best (and second best) candidate is the wave minimum of
// distance << 32 | visiting order.  A query whose map point has no observations (q_transient) does not block the
// feature it takes (:87-89, :1406-1408): its claim is stored as -2 - query.
// ---------------------------------------------------------------------------------------------
constexpr int kThHigh = 100;

constexpr int kProjFramePerKeypoint = 53;         // descriptor 32, position 8, visiting-order key 4, angle 4, right coordinate 4, octave 1
static size_t proj_frame_bytes(int cap) { return ((size_t)cap * kProjFramePerKeypoint + 15) & ~(size_t)15; }
static size_t proj_state_bytes(int cap, int qcap) { return (((size_t)cap * 4 + 15) & ~(size_t)15) + 3080 * 4 + (size_t)qcap * 3 + 16; }

// kFrameInGlobal: the per-feature data that is written once (53 bytes per feature) lives in a per-call slice of a
// device-memory arena instead of LDS, which then holds the claim states, the cell table and the accepted-match list only
template <int FORM, bool kFrameInGlobal>
__global__ __launch_bounds__(64) void k_proj_match(const float* __restrict__ qu, const float* __restrict__ qv, const float* __restrict__ qradius,
                                                   const float* __restrict__ qur, const int32_t* __restrict__ qlevel, const float* __restrict__ qangle,
                                                   const uint32_t* __restrict__ qdesc, const uint8_t* __restrict__ qtransient,
                                                   const int32_t* __restrict__ nqv, int qcap,
                                                   const orbx_keypoint* __restrict__ k2, const uint32_t* __restrict__ d2, const float* __restrict__ uright2,
                                                   const int32_t* __restrict__ n2v, int cap,
                                                   float minX, float minY, float invW, float invH, int mode, int th_dist, float nnratio, int check_ori,
                                                   int32_t* __restrict__ claimed, int32_t* __restrict__ nmatches,
                                                   unsigned char* __restrict__ arena, unsigned long long arena_stride)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const size_t frame_bytes = ((size_t)cap * kProjFramePerKeypoint + 15) & ~(size_t)15;
    unsigned char* fbase = kFrameInGlobal ? arena + (size_t)blockIdx.x * arena_stride : smem;
    uint4* s_desc = reinterpret_cast<uint4*>(fbase);                                 // 2 x uint4 per entry
    float2* s_xy = reinterpret_cast<float2*>(fbase + (size_t)cap * 32);
    uint32_t* s_okey = reinterpret_cast<uint32_t*>(fbase + (size_t)cap * 40);        // column << 26 | row << 20 | index
    float* s_angle = reinterpret_cast<float*>(fbase + (size_t)cap * 44);
    float* s_ur = reinterpret_cast<float*>(fbase + (size_t)cap * 48);
    uint8_t* s_oct = fbase + (size_t)cap * 52;
    unsigned char* mbase = kFrameInGlobal ? smem : smem + frame_bytes;
    int32_t* s_claim = reinterpret_cast<int32_t*>(mbase);
    const size_t o_cells = (size_t)(mbase - smem) + (((size_t)cap * 4 + 15) & ~(size_t)15);
    int32_t* s_cell = reinterpret_cast<int32_t*>(smem + o_cells);                    // [3073] first entry of every cell, column-major
    uint16_t* s_hj = reinterpret_cast<uint16_t*>(smem + o_cells + 3080 * 4);         // accepted matches: entry ...
    uint8_t* s_hb = smem + o_cells + 3080 * 4 + (size_t)qcap * 2;                    // ... and histogram bin
    __shared__ int s_hist[kHistoLength];
    __shared__ int s_run_b[64], s_run_i[64];                                         // per grid column of a window: first entry, inclusive length prefix

    const int lane = threadIdx.x;
    const int pair = blockIdx.x;
    const size_t foff = (size_t)pair * cap, qoff = (size_t)pair * qcap;
    k2 += foff; d2 += foff * 8; claimed += foff; if (uright2) uright2 += foff;
    qu += qoff; qv += qoff; qradius += qoff; qlevel += qoff; qdesc += qoff * 8;
    if (qur) qur += qoff; if (qangle) qangle += qoff; if (qtransient) qtransient += qoff;
    const int n2 = min(n2v[pair], cap), nq = min(nqv[pair], qcap);
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
    auto cell_of = [&](const orbx_keypoint& kp, int& px, int& py) -> bool {         // Frame::PosInGrid, src/Frame.cc:382-392
        px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, minX), invW));
        py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, minY), invH));
        return px >= 0 && px < 64 && py >= 0 && py < 48;
    };

    // counting sort of the in-grid features by cell: counts, exclusive prefix (lane c owns grid column c), placement
    for (int i = lane; i < 3073; i += 64) s_cell[i] = 0;
    if (lane < kHistoLength) s_hist[lane] = 0;
    wsync();
    for (int j = lane; j < n2; j += 64) {
        int px, py;
        if (cell_of(k2[j], px, py)) atomicAdd(&s_cell[px * 48 + py], 1);
    }
    wsync();
    {
        int sum = 0;
        for (int r = 0; r < 48; ++r) sum += s_cell[lane * 48 + r];
        int run = wave_scan_incl_i32(sum) - sum;
        for (int r = 0; r < 48; ++r) { const int c = s_cell[lane * 48 + r]; s_cell[lane * 48 + r] = run; run += c; }
        if (lane == 63) s_cell[3072] = run;
    }
    wsync();
    for (int j0 = 0; j0 < n2; j0 += 64) {
        const int j = j0 + lane;
        if (j < n2) {
            const orbx_keypoint kp = k2[j];
            int px, py;
            if (cell_of(kp, px, py)) {
                const int pos = atomicAdd(&s_cell[px * 48 + py], 1);               // the cell's cursor; the starts are restored below
                const uint4* tp = reinterpret_cast<const uint4*>(d2 + (size_t)j * 8);
                s_desc[2 * pos] = tp[0]; s_desc[2 * pos + 1] = tp[1];
                s_xy[pos] = make_float2(kp.x, kp.y);
                s_okey[pos] = ((uint32_t)px << 26) | ((uint32_t)py << 20) | (uint32_t)j;
                s_angle[pos] = kp.angle;
                s_ur[pos] = uright2 ? uright2[j] : -1.0f;
                s_claim[pos] = claimed[j];
                s_oct[pos] = (uint8_t)kp.octave;
            }
        }
    }
    wsync();
    // after the placement cell c's counter holds the start of cell c + 1: shift back by one cell
    {
        int v[48];
#pragma unroll
        for (int r = 0; r < 48; ++r) { const int i = lane * 48 + r; v[r] = i == 0 ? 0 : s_cell[i - 1]; }
        wsync();
#pragma unroll
        for (int r = 0; r < 48; ++r) s_cell[lane * 48 + r] = v[r];
    }
    wsync();

    int count = 0, nacc = 0;
    uint32_t nd[8]; float nu = 0.f, nv = 0.f, nr = -1.f, nur = 0.f, nang = 0.f; int nlev = 0, ntr = 0;
    auto fetch = [&](int e) {
        if (e < nq) {
            const uint4* tp = reinterpret_cast<const uint4*>(qdesc + (size_t)e * 8);
            const uint4 lo = tp[0], hi = tp[1];
            nd[0] = lo.x; nd[1] = lo.y; nd[2] = lo.z; nd[3] = lo.w; nd[4] = hi.x; nd[5] = hi.y; nd[6] = hi.z; nd[7] = hi.w;
            nu = qu[e]; nv = qv[e]; nr = qradius[e]; nlev = qlevel[e];
            nur = qur ? qur[e] : 0.f; nang = qangle ? qangle[e] : 0.f; ntr = qtransient ? qtransient[e] : 0;
        }
    };
    fetch(0);
    for (int e = 0; e < nq; ++e) {
        uint32_t cd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cd[k] = nd[k];
        const float u = nu, v = nv, r = nr, ur = nur, ang = nang; const int lev = nlev, transient = ntr;
        fetch(e + 1);
        // level band of the window (src/Frame.cc:348, :357-363)
        int minLevel, maxLevel;
        if (FORM == 1) { minLevel = lev - 1; maxLevel = lev; }
        else if (mode == 1) { minLevel = lev; maxLevel = -1; }
        else if (mode == 2) { minLevel = 0; maxLevel = lev; }
        else { minLevel = lev - 1; maxLevel = lev + 1; }
        const bool checkLevels = minLevel > 0 || maxLevel >= 0;
        // cells the window covers (src/Frame.cc:332-346)
        const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, minX), r), invW)));
        const int cx1 = min(63, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, minX), r), invW)));
        const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, minY), r), invH)));
        const int cy1 = min(47, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, minY), r), invH)));
        if (!(r >= 0.0f) || cx0 >= 64 || cx1 < 0 || cy0 >= 48 || cy1 < 0 || cx1 < cx0 || cy1 < cy0) continue;
        // one run of entries per covered grid column; inclusive prefix of the run lengths
        const int ncols = cx1 - cx0 + 1;
        int rb = 0, rl = 0;
        if (lane < ncols) { rb = s_cell[(cx0 + lane) * 48 + cy0]; rl = s_cell[(cx0 + lane) * 48 + cy1 + 1] - rb; }
        const int incl = wave_scan_incl_i32(rl);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) continue;
        s_run_b[lane] = rb - (incl - rl);                                  // entry = s_run_b[column] + t for t in the column's span
        s_run_i[lane] = incl;
        wsync();
        unsigned long long best = ~0ull, second = ~0ull;                   // distance << 32 | visiting order
        int bestj = -1, secondj = -1;
        for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            if (t >= total) continue;
            int c = 0;                                                     // first column whose inclusive prefix exceeds t
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) if (c + step <= ncols && s_run_i[c + step - 1] <= t) c += step;
            const int j = s_run_b[c] + t;
            const int oct = s_oct[j];
            if (checkLevels && (oct < minLevel || (maxLevel >= 0 && oct > maxLevel))) continue;
            const float2 p = s_xy[j];
            if (!(fabsf(__fsub_rn(p.x, u)) < r && fabsf(__fsub_rn(p.y, v)) < r)) continue;       // src/Frame.cc:370-374
            if (s_claim[j] >= 0) continue;                                                      // :87-89, :1406-1408, :1544-1545
            const float urj = s_ur[j];
            if (urj > 0.0f && fabsf(__fsub_rn(ur, urj)) > r) continue;                           // :91-96, :1410-1416
            const uint4 lo = s_desc[2 * j], hi = s_desc[2 * j + 1];
            const int dist = __popc(cd[0] ^ lo.x) + __popc(cd[1] ^ lo.y) + __popc(cd[2] ^ lo.z) + __popc(cd[3] ^ lo.w) +
                             __popc(cd[4] ^ hi.x) + __popc(cd[5] ^ hi.y) + __popc(cd[6] ^ hi.z) + __popc(cd[7] ^ hi.w);
            const unsigned long long key = ((unsigned long long)(uint32_t)dist << 32) | s_okey[j];
            if (key < best) { second = best; secondj = bestj; best = key; bestj = j; }
            else if (key < second) { second = key; secondj = j; }
        }
        const unsigned long long wbest = wave_min_u64(best);
        if (wbest == ~0ull) continue;
        const bool winner = best == wbest;
        const int wlane = __ffsll((long long)__ballot(winner)) - 1;
        const int wj = __builtin_amdgcn_readlane(bestj, wlane);
        const int bestDist = (int)(wbest >> 32);
        bool accept;
        if (FORM == 0) accept = bestDist <= th_dist;                       // :1428, :1556
        else {
            // second best in visiting order among equal distances = the second smallest key (:100-113)
            const unsigned long long mine = winner ? second : best;
            const unsigned long long wsecond = wave_min_u64(mine);
            accept = bestDist <= kThHigh;                                  // :116
            if (accept && wsecond != ~0ull) {
                const bool w2 = mine == wsecond;
                const int l2 = __ffsll((long long)__ballot(w2)) - 1;
                const int sj = __builtin_amdgcn_readlane(winner ? secondj : bestj, l2);
                const int bestDist2 = (int)(wsecond >> 32);
                if (s_oct[wj] == s_oct[sj] && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) accept = false;     // :118-119
            }
        }
        if (accept) {
            if (lane == 0) {
                s_claim[wj] = transient ? -2 - e : e;                       // :123, :1431
                if (FORM == 0 && check_ori) {
                    const int bin = rotation_bin_dev(ang, s_angle[wj]);    // :1436-1444
                    s_hj[nacc] = (uint16_t)wj; s_hb[nacc] = (uint8_t)bin; ++s_hist[bin];
                }
            }
            ++nacc; ++count;
            wsync();
        }
    }
    wsync();
    if (FORM == 0 && check_ori) {                                          // :1449-1470
        int ind1 = -1, ind2 = -1, ind3 = -1, m1 = 0, m2 = 0, m3 = 0;
        for (int i = 0; i < kHistoLength; ++i) {
            const int sz = s_hist[i];
            if (sz > m1) { m3 = m2; m2 = m1; m1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > m2) { m3 = m2; m2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > m3) { m3 = sz; ind3 = i; }
        }
        if ((float)m2 < __fmul_rn(0.1f, (float)m1)) { ind2 = -1; ind3 = -1; }
        else if ((float)m3 < __fmul_rn(0.1f, (float)m1)) ind3 = -1;
        for (int a0 = 0; a0 < nacc; a0 += 64) {
            const int a = a0 + lane;
            bool drop = false;
            if (a < nacc) { const int bin = s_hb[a]; drop = bin != ind1 && bin != ind2 && bin != ind3; }
            if (drop) s_claim[s_hj[a]] = -1;                               // every entry of a rejected bin clears its slot and counts down
            count -= __popcll(__ballot(drop));
        }
        wsync();
    }
    const int nin = s_cell[3072];
    for (int j = lane; j < nin; j += 64) claimed[s_okey[j] & 0xFFFFFu] = s_claim[j];
    if (lane == 0) nmatches[pair] = count;
}

static int stereo_band_rows(const float* scale_factors, int nlevels)
{
    float smax = 0.f;
    for (int i = 0; i < nlevels; ++i) smax = std::max(smax, scale_factors[i]);
    return (int)(4.0f * smax) + 3;
}
static size_t stereo_lds_bytes(int n_rows, int cap) { return sizeof(int32_t) * 2 * ((size_t)n_rows + 1) + sizeof(float) * (size_t)cap + (size_t)((cap + 15) & ~15) + 16; }
// index-list entries kept in LDS behind the row table: what the lists can hold at most, up to 64 KB (two workgroups per CU still fit)
static int stereo_lds_list_entries(size_t table_bytes, size_t list_stride)
{
    if (const char* e = getenv("ORBX_STEREO_LIST_IN_LDS")) if (atoi(e) == 0) return 0;      // test switch: lists in device memory
    const size_t room = table_bytes < 150 * 1024 ? 150 * 1024 - table_bytes : 0;
    return (int)std::min<size_t>(std::min<size_t>(list_stride, 32768), room / sizeof(uint16_t));
}

} // namespace

// Scratch of orbx_hamming_best2_device_static: the per-run partial results (kStaticPartBytes) followed by the per-query
// seed records of the two-pass search (kStaticSeedBytes).  Every GPU the library runs on holds its own copy until the
// process ends.  In host code this name only stands for the symbol: the memory is reached through the per-device address
// that hipGetSymbolAddress returns (orbx_matcher::static_scratch), never through the name.  (No `used` attribute here:
// the host-side reference below is what makes the compiler export the symbol from the device image; with the attribute
// it stays local there and the lookup cannot find it.)
constexpr size_t kStaticMaxQueries = 16384;
constexpr size_t kStaticPartBytes = 4u << 20;
constexpr size_t kStaticSeedBytes = sizeof(orbx_best2) * kStaticMaxQueries;
static __device__ __align__(256) unsigned char scratch[kStaticPartBytes + kStaticSeedBytes];

// Calls of the static entry point on one GPU share that GPU's scratch whatever matcher or stream they come from, so they
// are put in order: one event per device (created with the first matcher of that device, never destroyed), waited for in
// front of a call's launches and recorded behind them, under a host lock so that two host threads cannot interleave the
// wait / launch / record of their calls.  Neither allocates.  A call under stream capture skips both (an event wait would
// become part of the caller's graph): captured calls stay the caller's to order.
constexpr int kMaxStaticDevices = 64;
static hipEvent_t g_static_done[kMaxStaticDevices] = {};
static std::mutex g_static_mu[kMaxStaticDevices];
static std::mutex g_static_create_mu;

struct orbx_matcher {
    int device = 0;
    void* static_scratch = nullptr;                // this device's address of `scratch`, resolved at create
    hipStream_t stream = nullptr;
    void* ws = nullptr; size_t ws_bytes = 0;       // partial results of the brute-force kernel
    void* seed = nullptr; size_t seed_bytes = 0;   // its per-query result over the first rows (threshold seed of the full pass)
    void* stage[8] = {}; size_t stage_bytes[8] = {}; // staging buffers of the host entry points
    void* win = nullptr; size_t win_bytes = 0;       // device arena of the windowed candidate generation
    void* arena = nullptr; size_t arena_bytes = 0;   // frame part of the batched device matchers when it does not fit LDS (grows; calls on one matcher must not overlap)
    uint8_t* pin = nullptr; size_t pin_bytes = 0;    // pinned host mirror of it (one upload, two small downloads per call)
};

static int reserve(void** p, size_t* have, size_t need)
{
    if (*have >= need && *p) return ORBX_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    HIP_CHECK(hipMalloc(p, std::max<size_t>(need, 256)));
    *have = std::max<size_t>(need, 256);
    return ORBX_OK;
}

constexpr int64_t kSeedRows = 16384;             // rows of the seeding pass of large maps

// Row runs of the matrix-core kernel: one per wavefront, a multiple of 32 rows, enough of them to give every SIMD two waves,
// and as long as that allows (a lane's second-best tightens with the rows it has seen, and fewer tiles walk their keys).
static void chunking(int nq, int64_t nt, int& rows_per_chunk, int& nchunks)
{
    const int qtiles = (nq + kMfmaQueries - 1) / kMfmaQueries;
    const int64_t want = std::max<int64_t>(4, (2048 + qtiles - 1) / std::max(qtiles, 1));
    int64_t rpc = (nt + want - 1) / want;
    rpc = std::max<int64_t>((rpc + 31) & ~(int64_t)31, 256);
    rpc = std::min<int64_t>(rpc, (1 << kIdxBits));
    rows_per_chunk = (int)rpc;
    nchunks = (int)((nt + rpc - 1) / rpc);
    if (nchunks < 1) nchunks = 1;
}

extern "C" {

int orbx_matcher_create(int device, orbx_matcher** out)
{
    if (!out) return ORBX_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return ORBX_ERR_NO_DEVICE; }
    if (device >= count) { set_error("device index out of range"); return ORBX_ERR_BAD_ARG; }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    orbx_matcher* m = new orbx_matcher();
    struct Guard { orbx_matcher* p; ~Guard() { orbx_matcher_destroy(p); } } guard{m};
    HIP_CHECK(hipGetDevice(&m->device));
    HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    // the matcher's device is current here: this is that device's copy of the array
    HIP_CHECK(hipGetSymbolAddress(&m->static_scratch, HIP_SYMBOL(scratch)));
    if (m->device < kMaxStaticDevices) {
        std::lock_guard<std::mutex> lock(g_static_create_mu);
        if (!g_static_done[m->device]) HIP_CHECK(hipEventCreateWithFlags(&g_static_done[m->device], hipEventDisableTiming));
    }
    guard.p = nullptr;
    *out = m;
    return ORBX_OK;
}

void orbx_matcher_destroy(orbx_matcher* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->ws) (void)hipFree(m->ws);
    if (m->seed) (void)hipFree(m->seed);
    for (void* p : m->stage) if (p) (void)hipFree(p);
    if (m->win) (void)hipFree(m->win);
    if (m->arena) (void)hipFree(m->arena);
    if (m->pin) (void)hipHostFree(m->pin);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int64_t orbx_hamming_best2_workspace_bytes(int nq, int64_t nt)
{
    int rpc, nch, rpc1, nch1;
    chunking(nq, nt, rpc, nch);
    chunking(nq, std::min<int64_t>(nt, kSeedRows), rpc1, nch1);  // the seeding pass of large maps
    return (int64_t)sizeof(Part) * std::max(nch, nch1) * nq + (int64_t)sizeof(orbx_best2) * nq;
}

// Launches of the brute-force search; parts / seed are device scratch of at least best2_part_bytes / nq records.
static int launch_best2(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt, int64_t index_base, orbx_best2* d_out,
                        hipStream_t st, Part* parts, orbx_best2* seed_buf, bool seeded, int rpc, int nch, int rpc1, int nch1)
{
    const int qtiles = (nq + kMfmaQueries - 1) / kMfmaQueries;
    const orbx_best2* d_seed = nullptr;
    if (seeded) {
        hipLaunchKernelGGL(k_best2_mfma, dim3((nch1 + 3) / 4, qtiles), dim3(kBlock), 0, st, reinterpret_cast<const uint32_t*>(d_q), nq,
                           reinterpret_cast<const uint32_t*>(d_t), (long long)kSeedRows, rpc1, nch1, (const orbx_best2*)nullptr, parts);
        hipLaunchKernelGGL(k_best2_fold, dim3((nq + 255) / 256), dim3(256), 0, st, (const Part*)parts, nch1, nq, rpc1, 0LL, seed_buf);
        d_seed = seed_buf;
    }
    if (nt == 0) nch = 0;
    if (nch > 0)
        hipLaunchKernelGGL(k_best2_mfma, dim3((nch + 3) / 4, qtiles), dim3(kBlock), 0, st, reinterpret_cast<const uint32_t*>(d_q), nq,
                           reinterpret_cast<const uint32_t*>(d_t), (long long)nt, rpc, nch, d_seed, parts);
    hipLaunchKernelGGL(k_best2_fold, dim3((nq + 255) / 256), dim3(256), 0, st, (const Part*)parts, nch, nq, rpc, (long long)index_base, d_out);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

static int check_best2_args(const char* who, const orbx_matcher* m, const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                            int64_t index_base, const orbx_best2* d_out)
{
    if (!m || !d_q || !d_out || nq < 1 || nt < 0 || (nt > 0 && !d_t)) { set_error(std::string(who) + ": bad argument"); return ORBX_ERR_BAD_ARG; }
    if (index_base < 0 || index_base + nt > 0x7FFFFFFFLL) { set_error("global row index exceeds int32"); return ORBX_ERR_BAD_ARG; }
    // the kernel reads queries as 16-byte and rows as 8-byte vectors
    if (((uintptr_t)d_q & 15u) || ((uintptr_t)d_t & 7u)) { set_error(std::string(who) + ": queries must be 16-byte aligned, map rows 8-byte aligned"); return ORBX_ERR_BAD_ARG; }
    return ORBX_OK;
}

int orbx_hamming_best2_device(orbx_matcher* m, const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt, int64_t index_base,
                              orbx_best2* d_out, void* stream)
{
    int rc = check_best2_args("orbx_hamming_best2_device", m, d_q, nq, d_t, nt, index_base, d_out);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(m->device));
    int rpc, nch, rpc1 = 0, nch1 = 0;
    chunking(nq, nt, rpc, nch);
    // Large maps: a first pass over the first kSeedRows rows gives every query a second-best distance that no row of the final
    // answer exceeds; seeded with it, the full pass walks the keys of few tiles instead of warming every row run up from 256.
    const bool seeded = nt >= 16 * kSeedRows;
    if (seeded) chunking(nq, kSeedRows, rpc1, nch1);
    if ((rc = reserve(&m->ws, &m->ws_bytes, sizeof(Part) * (size_t)std::max(nch, nch1) * nq))) return rc;
    if (seeded && (rc = reserve(&m->seed, &m->seed_bytes, sizeof(orbx_best2) * (size_t)nq))) return rc;
    return launch_best2(d_q, nq, d_t, nt, index_base, d_out, (hipStream_t)stream, reinterpret_cast<Part*>(m->ws),
                        reinterpret_cast<orbx_best2*>(m->seed), seeded, rpc, nch, rpc1, nch1);
}

// The same search for callers that must not allocate (HIP graph capture, serving loops): the partial results and the seed
// records live in this device's copy of `scratch`, reached through the address resolved at orbx_matcher_create.  Nothing
// in the scratch has to start at zero: every record the fold kernels read is written first by the search launch in front
// of them on the same stream.
int orbx_hamming_best2_device_static(orbx_matcher* m, const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt, int64_t index_base,
                                     orbx_best2* d_out, void* stream)
{
    int rc = check_best2_args("orbx_hamming_best2_device_static", m, d_q, nq, d_t, nt, index_base, d_out);
    if (rc) return rc;
    if (!m->static_scratch) { set_error("orbx_hamming_best2_device_static: the device scratch was not resolved at create"); return ORBX_ERR_HIP; }
    int rpc, nch, rpc1 = 0, nch1 = 0;
    chunking(nq, nt, rpc, nch);
    const bool seeded = nt >= 16 * kSeedRows;
    if (seeded) chunking(nq, kSeedRows, rpc1, nch1);
    if ((size_t)nq > kStaticMaxQueries || sizeof(Part) * (size_t)std::max(nch, nch1) * (size_t)nq > kStaticPartBytes) {
        set_error("orbx_hamming_best2_device_static: needs more than the fixed scratch (at most 16384 queries and 4 MiB of partial "
                  "results, see orbx_hamming_best2_workspace_bytes); use orbx_hamming_best2_device");
        return ORBX_ERR_CAPACITY;
    }
    HIP_CHECK(hipSetDevice(m->device));
    unsigned char* base = static_cast<unsigned char*>(m->static_scratch);
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    HIP_CHECK(hipStreamIsCapturing(st, &capturing));
    const bool ordered = capturing == hipStreamCaptureStatusNone && m->device < kMaxStaticDevices && g_static_done[m->device];
    if (!ordered)
        return launch_best2(d_q, nq, d_t, nt, index_base, d_out, st, reinterpret_cast<Part*>(base),
                            reinterpret_cast<orbx_best2*>(base + kStaticPartBytes), seeded, rpc, nch, rpc1, nch1);
    std::lock_guard<std::mutex> lock(g_static_mu[m->device]);
    HIP_CHECK(hipStreamWaitEvent(st, g_static_done[m->device], 0));      // behind the last call that used this GPU's scratch
    rc = launch_best2(d_q, nq, d_t, nt, index_base, d_out, st, reinterpret_cast<Part*>(base),
                      reinterpret_cast<orbx_best2*>(base + kStaticPartBytes), seeded, rpc, nch, rpc1, nch1);
    HIP_CHECK(hipEventRecord(g_static_done[m->device], st));
    return rc;
}

int orbx_debug_best2_tile_walks(orbx_matcher* m, uint64_t* walked, int reset)
{
    if (!m || !walked) { set_error("orbx_debug_best2_tile_walks: null pointer"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(m->device));
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long v = 0;
    HIP_CHECK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_best2_walked), sizeof(v)));
    *walked = v;
    if (reset) { v = 0; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_best2_walked), &v, sizeof(v))); }
    return ORBX_OK;
}

int orbx_best2_merge_device(orbx_matcher* m, const orbx_best2* d_parts, int n_parts, int nq, orbx_best2* d_out, void* stream)
{
    if (!m || !d_parts || !d_out || n_parts < 1 || nq < 1) { set_error("orbx_best2_merge_device: bad argument"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_best2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, d_parts, n_parts, nq, d_out);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_hamming_best2(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int64_t nt, orbx_best2* out)
{
    if (!m || !q || !out || nq < 1 || nt < 0 || (nt > 0 && !t)) { set_error("orbx_hamming_best2: bad argument"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(m->device));
    int rc;
    if ((rc = reserve(&m->stage[0], &m->stage_bytes[0], (size_t)nq * 32))) return rc;
    if ((rc = reserve(&m->stage[1], &m->stage_bytes[1], (size_t)nt * 32))) return rc;
    if ((rc = reserve(&m->stage[2], &m->stage_bytes[2], sizeof(orbx_best2) * (size_t)nq))) return rc;
    HIP_CHECK(hipMemcpyAsync(m->stage[0], q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    if (nt) HIP_CHECK(hipMemcpyAsync(m->stage[1], t, (size_t)nt * 32, hipMemcpyHostToDevice, m->stream));
    rc = orbx_hamming_best2_device(m, (const uint8_t*)m->stage[0], nq, (const uint8_t*)m->stage[1], nt, 0, (orbx_best2*)m->stage[2], m->stream);
    if (rc) return rc;
    HIP_CHECK(hipMemcpyAsync(out, m->stage[2], sizeof(orbx_best2) * (size_t)nq, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

int orbx_hamming_csr(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, const int32_t* rowptr, const int32_t* cand,
                     int32_t* dist_out)
{
    if (!m || nq < 0 || nt < 0 || !rowptr) { set_error("orbx_hamming_csr: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (nq == 0) return ORBX_OK;
    const int total = rowptr[nq];
    if (total == 0) return ORBX_OK;
    if (!q || !t || !cand || !dist_out) { set_error("orbx_hamming_csr: null pointer"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < nq; ++i) if (rowptr[i] > rowptr[i + 1] || rowptr[i] < 0) { set_error("rowptr not monotone"); return ORBX_ERR_BAD_ARG; }
    for (int c = 0; c < total; ++c) if (cand[c] < 0 || cand[c] >= nt) { set_error("candidate index out of range"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(m->device));
    int rc;
    if ((rc = reserve(&m->stage[0], &m->stage_bytes[0], (size_t)nq * 32))) return rc;
    if ((rc = reserve(&m->stage[1], &m->stage_bytes[1], (size_t)nt * 32))) return rc;
    if ((rc = reserve(&m->stage[3], &m->stage_bytes[3], sizeof(int32_t) * (size_t)(nq + 1)))) return rc;
    if ((rc = reserve(&m->stage[4], &m->stage_bytes[4], sizeof(int32_t) * (size_t)total))) return rc;
    if ((rc = reserve(&m->stage[5], &m->stage_bytes[5], sizeof(int32_t) * (size_t)total))) return rc;
    HIP_CHECK(hipMemcpyAsync(m->stage[0], q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->stage[1], t, (size_t)nt * 32, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->stage[3], rowptr, sizeof(int32_t) * (size_t)(nq + 1), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->stage[4], cand, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_csr_dist, dim3((nq + 3) / 4), dim3(256), 0, m->stream, (const uint32_t*)m->stage[0], nq, (const uint32_t*)m->stage[1],
                       (const int32_t*)m->stage[3], (const int32_t*)m->stage[4], (int32_t*)m->stage[5]);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(dist_out, m->stage[5], sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

int orbx_search_for_initialization_batch_device(orbx_matcher* m, const orbx_keypoint* d_k1, const uint8_t* d_d1, const int32_t* d_n1,
                                                const orbx_keypoint* d_k2, const uint8_t* d_d2, const int32_t* d_n2, int cap, int n_pairs,
                                                float minX, float maxX, float minY, float maxY, float* d_prev_matched_xy, int window,
                                                float nnratio, int check_ori, int32_t* d_matches12, int32_t* d_nmatches, void* stream)
{
    if (!m || cap < 1 || n_pairs < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_search_for_initialization_batch_device: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (n_pairs == 0) return ORBX_OK;
    if (!d_k1 || !d_d1 || !d_n1 || !d_k2 || !d_d2 || !d_n2 || !d_prev_matched_xy || !d_matches12 || !d_nmatches) { set_error("orbx_search_for_initialization_batch_device: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (cap >= 65535) { set_error("orbx_search_for_initialization_batch_device: per-frame capacity must be below 65535 (16-bit match indices)"); return ORBX_ERR_CAPACITY; }
    HIP_CHECK(hipSetDevice(m->device));
    const float invW = 64.0f / (maxX - minX), invH = 48.0f / (maxY - minY);      // Frame::mfGridElementWidthInv / HeightInv, src/Frame.cc:101-102
    const size_t lds_all = init_frame_bytes(cap) + init_state_bytes(cap);
    if (lds_all <= kMatchLdsLimit) {
        HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_init_match<false>), lds_all));
        hipLaunchKernelGGL(k_init_match<false>, dim3(n_pairs), dim3(64), lds_all, (hipStream_t)stream, d_k1, reinterpret_cast<const uint32_t*>(d_d1), d_n1,
                           d_k2, reinterpret_cast<const uint32_t*>(d_d2), d_n2, cap, minX, minY, invW, invH, d_prev_matched_xy, (float)window, nnratio,
                           check_ori, d_matches12, d_nmatches, (unsigned char*)nullptr, 0ull);
    } else {
        // frame part in a device-memory arena of the matcher (one slice per pair), state in LDS
        const size_t lds = init_state_bytes(cap);
        if (lds > kMatchLdsLimit) {
            set_error("orbx_search_for_initialization_batch_device: per-frame capacity above 17000 keypoints: the state of the sequential pass (9 bytes per keypoint) does not fit the LDS of one CU");
            return ORBX_ERR_CAPACITY;
        }
        const size_t stride = init_frame_bytes(cap);
        int rc;
        if ((rc = reserve(&m->arena, &m->arena_bytes, stride * (size_t)n_pairs))) return rc;
        HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_init_match<true>), lds));
        hipLaunchKernelGGL(k_init_match<true>, dim3(n_pairs), dim3(64), lds, (hipStream_t)stream, d_k1, reinterpret_cast<const uint32_t*>(d_d1), d_n1,
                           d_k2, reinterpret_cast<const uint32_t*>(d_d2), d_n2, cap, minX, minY, invW, invH, d_prev_matched_xy, (float)window, nnratio,
                           check_ori, d_matches12, d_nmatches, (unsigned char*)m->arena, (unsigned long long)stride);
    }
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_search_by_projection_batch_device(orbx_matcher* m, int form,
                                           const float* d_qu, const float* d_qv, const float* d_qradius, const float* d_qur,
                                           const int32_t* d_qlevel, const float* d_qangle, const uint8_t* d_qdesc, const uint8_t* d_q_transient,
                                           const int32_t* d_nq, int qcap,
                                           const orbx_keypoint* d_k2, const uint8_t* d_d2, const float* d_uright2, const int32_t* d_n2, int cap,
                                           int n_calls, float minX, float maxX, float minY, float maxY, int mode, int th_dist, float nnratio,
                                           int check_ori, int32_t* d_claimed, int32_t* d_nmatches, void* stream)
{
    if (!m || form < 0 || form > 1 || cap < 1 || qcap < 1 || n_calls < 0 || !(maxX > minX) || !(maxY > minY) || mode < 0 || mode > 2) {
        set_error("orbx_search_by_projection_batch_device: bad argument"); return ORBX_ERR_BAD_ARG;
    }
    if (n_calls == 0) return ORBX_OK;
    if (!d_qu || !d_qv || !d_qradius || !d_qlevel || !d_qdesc || !d_nq || !d_k2 || !d_d2 || !d_n2 || !d_claimed || !d_nmatches ||
        (d_uright2 && !d_qur) || (form == 0 && check_ori && !d_qangle)) {
        set_error("orbx_search_by_projection_batch_device: null pointer"); return ORBX_ERR_BAD_ARG;
    }
    if (cap >= (1 << 16) || qcap >= (1 << 20)) { set_error("orbx_search_by_projection_batch_device: capacities must stay below 65536 features / 1048576 queries"); return ORBX_ERR_CAPACITY; }
    HIP_CHECK(hipSetDevice(m->device));
    const float invW = 64.0f / (maxX - minX), invH = 48.0f / (maxY - minY);      // src/Frame.cc:101-102
    const size_t lds_all = proj_frame_bytes(cap) + proj_state_bytes(cap, qcap);
    const bool in_lds = lds_all <= kMatchLdsLimit;
    const size_t lds = in_lds ? lds_all : proj_state_bytes(cap, qcap);
    if (lds > kMatchLdsLimit) {
        set_error("orbx_search_by_projection_batch_device: 4 bytes per feature plus 3 bytes per query of capacity (claim states and accepted matches) do not fit the LDS of one CU");
        return ORBX_ERR_CAPACITY;
    }
    unsigned char* arena = nullptr;
    const size_t stride = proj_frame_bytes(cap);
    if (!in_lds) {
        int rc;
        if ((rc = reserve(&m->arena, &m->arena_bytes, stride * (size_t)n_calls))) return rc;
        arena = (unsigned char*)m->arena;
    }
#define ORBX_LAUNCH_PROJ(FORM, G)                                                                                                              \
    do {                                                                                                                                       \
        HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_proj_match<FORM, G>), lds));                                   \
        hipLaunchKernelGGL((k_proj_match<FORM, G>), dim3(n_calls), dim3(64), lds, (hipStream_t)stream, d_qu, d_qv, d_qradius, d_qur, d_qlevel, \
                           d_qangle, reinterpret_cast<const uint32_t*>(d_qdesc), d_q_transient, d_nq, qcap, d_k2,                              \
                           reinterpret_cast<const uint32_t*>(d_d2), d_uright2, d_n2, cap, minX, minY, invW, invH, mode, th_dist, nnratio,      \
                           check_ori, d_claimed, d_nmatches, arena, (unsigned long long)stride);                                               \
    } while (0)
    if (form == 0) { if (in_lds) ORBX_LAUNCH_PROJ(0, false); else ORBX_LAUNCH_PROJ(0, true); }
    else { if (in_lds) ORBX_LAUNCH_PROJ(1, false); else ORBX_LAUNCH_PROJ(1, true); }
#undef ORBX_LAUNCH_PROJ
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_stereo_match_batch_device(orbx_matcher* m, const orbx_keypoint* d_kl, const uint8_t* d_dl, const int32_t* d_nl,
                                   const orbx_keypoint* d_kr, const uint8_t* d_dr, const int32_t* d_nr, int cap_per_frame, int n_pairs,
                                   const float* scale_factors, int nlevels, int n_rows, float min_d, float max_d,
                                   int32_t* d_best_dist, int32_t* d_best_idx, void* stream)
{
    if (!m || !d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !scale_factors || !d_best_dist || !d_best_idx ||
        cap_per_frame < 1 || cap_per_frame > 65535 || n_pairs < 1 || nlevels < 1 || nlevels > kMaxLevels) {
        set_error("orbx_stereo_match_batch_device: bad argument"); return ORBX_ERR_BAD_ARG;
    }
    HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = reserve(&m->stage[6], &m->stage_bytes[6], sizeof(float) * kMaxLevels))) return rc;
    const size_t list_stride = (size_t)cap_per_frame * stereo_band_rows(scale_factors, nlevels);
    if ((rc = reserve(&m->stage[7], &m->stage_bytes[7], sizeof(uint16_t) * list_stride * (size_t)n_pairs))) return rc;
    const size_t table_lds = stereo_lds_bytes(n_rows, cap_per_frame);
    if (n_rows < 1 || table_lds > 150 * 1024) { set_error("image rows / keypoint capacity too large for the row table in LDS"); return ORBX_ERR_BAD_ARG; }
    const int lds_list = stereo_lds_list_entries(table_lds, list_stride);
    const size_t lds = table_lds + sizeof(uint16_t) * (size_t)lds_list;
    HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_stereo), lds));
    HIP_CHECK(hipMemcpyAsync(m->stage[6], scale_factors, sizeof(float) * nlevels, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_stereo, dim3(n_pairs), dim3(kStereoThreads), lds, st, d_kl, (const uint32_t*)d_dl, d_nl,
                       d_kr, (const uint32_t*)d_dr, d_nr, cap_per_frame, 0, 0, (const float*)m->stage[6], n_rows, min_d, max_d,
                       (uint16_t*)m->stage[7], (unsigned long long)list_stride, lds_list, d_best_dist, d_best_idx);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_stereo_match(orbx_matcher* m, const orbx_keypoint* kl, const uint8_t* dl, int nl, const orbx_keypoint* kr, const uint8_t* dr, int nr,
                      const float* scale_factors, int nlevels, int n_rows, float min_d, float max_d, int32_t* best_dist, int32_t* best_idx)
{
    if (!m || nl < 0 || nr < 0 || !scale_factors || nlevels < 1 || nlevels > kMaxLevels || nr > 65535) { set_error("orbx_stereo_match: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (nl == 0) return ORBX_OK;
    if (!kl || !dl || !best_dist || !best_idx || (nr > 0 && (!kr || !dr))) { set_error("orbx_stereo_match: null pointer"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < nl; ++i) if (kl[i].octave < 0 || kl[i].octave >= nlevels) { set_error("left octave out of range"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < nr; ++i) if (kr[i].octave < 0 || kr[i].octave >= nlevels) { set_error("right octave out of range"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(m->device));
    const int cap = std::max(nl, std::max(nr, 1));
    int rc;
    if ((rc = reserve(&m->stage[0], &m->stage_bytes[0], sizeof(orbx_keypoint) * (size_t)cap))) return rc;
    if ((rc = reserve(&m->stage[1], &m->stage_bytes[1], (size_t)cap * 32))) return rc;
    if ((rc = reserve(&m->stage[2], &m->stage_bytes[2], sizeof(orbx_keypoint) * (size_t)cap))) return rc;
    if ((rc = reserve(&m->stage[3], &m->stage_bytes[3], (size_t)cap * 32))) return rc;
    if ((rc = reserve(&m->stage[4], &m->stage_bytes[4], sizeof(int32_t) * (size_t)cap))) return rc;
    if ((rc = reserve(&m->stage[5], &m->stage_bytes[5], sizeof(int32_t) * (size_t)cap))) return rc;
    if ((rc = reserve(&m->stage[6], &m->stage_bytes[6], sizeof(float) * kMaxLevels))) return rc;
    HIP_CHECK(hipMemcpyAsync(m->stage[0], kl, sizeof(orbx_keypoint) * (size_t)nl, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(m->stage[1], dl, (size_t)nl * 32, hipMemcpyHostToDevice, m->stream));
    if (nr) {
        HIP_CHECK(hipMemcpyAsync(m->stage[2], kr, sizeof(orbx_keypoint) * (size_t)nr, hipMemcpyHostToDevice, m->stream));
        HIP_CHECK(hipMemcpyAsync(m->stage[3], dr, (size_t)nr * 32, hipMemcpyHostToDevice, m->stream));
    }
    HIP_CHECK(hipMemcpyAsync(m->stage[6], scale_factors, sizeof(float) * nlevels, hipMemcpyHostToDevice, m->stream));
    const size_t list_stride = (size_t)cap * stereo_band_rows(scale_factors, nlevels);
    if ((rc = reserve(&m->stage[7], &m->stage_bytes[7], sizeof(uint16_t) * list_stride))) return rc;
    const size_t table_lds = stereo_lds_bytes(n_rows, cap);
    if (n_rows < 1 || table_lds > 150 * 1024) { set_error("image rows / keypoint count too large for the row table in LDS"); return ORBX_ERR_BAD_ARG; }
    const int lds_list = stereo_lds_list_entries(table_lds, list_stride);
    const size_t lds = table_lds + sizeof(uint16_t) * (size_t)lds_list;
    HIP_CHECK((hipError_t)raise_dynamic_lds(reinterpret_cast<const void*>(k_stereo), lds));
    hipLaunchKernelGGL(k_stereo, dim3(1), dim3(kStereoThreads), lds, m->stream, (const orbx_keypoint*)m->stage[0], (const uint32_t*)m->stage[1],
                       (const int32_t*)nullptr, (const orbx_keypoint*)m->stage[2], (const uint32_t*)m->stage[3], (const int32_t*)nullptr, cap, nl, nr,
                       (const float*)m->stage[6], n_rows, min_d, max_d, (uint16_t*)m->stage[7], (unsigned long long)list_stride, lds_list,
                       (int32_t*)m->stage[4], (int32_t*)m->stage[5]);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(best_dist, m->stage[4], sizeof(int32_t) * (size_t)nl, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipMemcpyAsync(best_idx, m->stage[5], sizeof(int32_t) * (size_t)nl, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

} // extern "C"

namespace orbx {
int csr_distances(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, const int32_t* rowptr, const int32_t* cand, int32_t* dist_out)
{
    return orbx_hamming_csr(m, q, nq, t, nt, rowptr, cand, dist_out);
}
int window_csr(orbx_matcher* m, const WindowQuery* q, const uint8_t* qdesc, int nq, const orbx_keypoint* k2, const uint8_t* d2, int n2,
               const int* grid_order, int n_in_grid, std::vector<int32_t>& rowptr, std::vector<int32_t>& cand, std::vector<int32_t>& dist)
{
    rowptr.assign((size_t)nq + 1, 0); cand.clear(); dist.clear();
    if (nq == 0 || n2 == 0 || n_in_grid == 0) return ORBX_OK;
    HIP_CHECK(hipSetDevice(m->device));
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // arena: [queries | query descriptors | keypoints | descriptors | grid order | rowptr | cand | dist]
    const size_t o_q = 0, o_qd = o_q + up(sizeof(WindowQuery) * nq), o_kp = o_qd + up((size_t)nq * 32), o_d2 = o_kp + up(sizeof(KpLite) * n2),
                 o_ord = o_d2 + up((size_t)n2 * 32), o_row = o_ord + up(sizeof(int32_t) * n_in_grid), o_out = o_row + up(sizeof(int32_t) * (nq + 1));
    size_t cap = std::max<size_t>((size_t)64 * nq, 1 << 16);      // candidate capacity; grown when a call needs more
    bool filled = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const size_t need = o_out + up(sizeof(int32_t) * cap) * 2;
        if (m->win_bytes < need) {
            filled = false;
            if (m->win) (void)hipFree(m->win);
            if (m->pin) (void)hipHostFree(m->pin);
            m->win = nullptr; m->pin = nullptr; m->win_bytes = m->pin_bytes = 0;
            HIP_CHECK(hipMalloc(&m->win, need));
            HIP_CHECK(hipHostMalloc((void**)&m->pin, need, hipHostMallocDefault));
            m->win_bytes = m->pin_bytes = need;
        }
        uint8_t* h = m->pin; uint8_t* d = static_cast<uint8_t*>(m->win);
        if (!filled) {
            filled = true;
            std::memcpy(h + o_q, q, sizeof(WindowQuery) * nq);
            std::memcpy(h + o_qd, qdesc, (size_t)nq * 32);
            KpLite* kl = reinterpret_cast<KpLite*>(h + o_kp);
            for (int i = 0; i < n2; ++i) { kl[i].x = k2[i].x; kl[i].y = k2[i].y; kl[i].octave = k2[i].octave; }
            std::memcpy(h + o_d2, d2, (size_t)n2 * 32);
            std::memcpy(h + o_ord, grid_order, sizeof(int32_t) * n_in_grid);
        }
        HIP_CHECK(hipMemcpyAsync(d, h, o_row, hipMemcpyHostToDevice, m->stream));
        const dim3 grid((nq + 3) / 4), block(256);
        int32_t* d_row = reinterpret_cast<int32_t*>(d + o_row);
        int32_t* d_cand = reinterpret_cast<int32_t*>(d + o_out);
        int32_t* d_dist = reinterpret_cast<int32_t*>(d + o_out + up(sizeof(int32_t) * cap));
        hipLaunchKernelGGL(k_window<false>, grid, block, 0, m->stream, reinterpret_cast<const WindowQuery*>(d + o_q), reinterpret_cast<const uint32_t*>(d + o_qd), nq,
                           reinterpret_cast<const KpLite*>(d + o_kp), reinterpret_cast<const uint32_t*>(d + o_d2), reinterpret_cast<const int32_t*>(d + o_ord),
                           n_in_grid, d_row, d_cand, d_dist, (int)cap);
        hipLaunchKernelGGL(k_window_scan, dim3(1), dim3(1024), 0, m->stream, d_row, nq);
        hipLaunchKernelGGL(k_window<true>, grid, block, 0, m->stream, reinterpret_cast<const WindowQuery*>(d + o_q), reinterpret_cast<const uint32_t*>(d + o_qd), nq,
                           reinterpret_cast<const KpLite*>(d + o_kp), reinterpret_cast<const uint32_t*>(d + o_d2), reinterpret_cast<const int32_t*>(d + o_ord),
                           n_in_grid, d_row, d_cand, d_dist, (int)cap);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(h + o_row, d_row, sizeof(int32_t) * (nq + 1), hipMemcpyDeviceToHost, m->stream));
        HIP_CHECK(hipStreamSynchronize(m->stream));
        const int32_t* hrow = reinterpret_cast<const int32_t*>(h + o_row);
        const size_t total = (size_t)hrow[nq];
        if (total > cap) { cap = total; continue; }               // rare: rerun with room for every candidate
        rowptr.assign(hrow, hrow + nq + 1);
        if (total) {
            HIP_CHECK(hipMemcpyAsync(h + o_out, d_cand, sizeof(int32_t) * total, hipMemcpyDeviceToHost, m->stream));
            HIP_CHECK(hipMemcpyAsync(h + o_out + up(sizeof(int32_t) * cap), d_dist, sizeof(int32_t) * total, hipMemcpyDeviceToHost, m->stream));
            HIP_CHECK(hipStreamSynchronize(m->stream));
            const int32_t* hc = reinterpret_cast<const int32_t*>(h + o_out);
            const int32_t* hd = reinterpret_cast<const int32_t*>(h + o_out + up(sizeof(int32_t) * cap));
            cand.assign(hc, hc + total); dist.assign(hd, hd + total);
        }
        return ORBX_OK;
    }
    set_error("window_csr: candidate capacity");
    return ORBX_ERR_CAPACITY;
}
int matcher_device(const orbx_matcher* m) { return m->device; }
void* matcher_stream(const orbx_matcher* m) { return m->stream; }
int matcher_stage(orbx_matcher* m, int slot, size_t bytes, void** out)
{
    if (slot < 0 || slot >= 8) return ORBX_ERR_BAD_ARG;
    const int rc = reserve(&m->stage[slot], &m->stage_bytes[slot], bytes);
    *out = m->stage[slot];
    return rc;
}
}
