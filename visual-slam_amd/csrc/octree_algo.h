// Level-synchronous formulation of DistributeOctTree (src/ORBextractor.cc:538-762).
//
// The reference walks a std::list, splitting nodes one at a time.  Each of its passes is a
// data-parallel "split step" over a subset S of the list's nodes, processed in an order pi:
//   coarse sweep (:605-664): S = every node with more than one key, pi = list order;
//   fine pass   (:675-736): S = the largest nodes first (sorted by key count, ties by creation
//                           sequence -- the reference's tie is the heap address, which no rerun
//                           reproduces), cut as soon as the list holds N nodes.
// After a step the list reads: children of the node processed LAST first (order n4,n3,n2,n1, empty
// ones dropped) ... children of the node processed first, then every unsplit node in its old order.
// Keys never move: each key carries the list position of its node, a node with one key keeps that
// key in its record, and the best key of a larger node is found at the end by an atomic max.
//
// Most steps never touch the keys at all: while every node with more than one key is split in every step, the
// nodes are cells of the full quadtree, whose key counts one counting pass tabulates (see "Count tables" below);
// the keys are handed to their nodes when the steps are over, or when a step has to go below the tables' depth.
//
// The same source builds for the device (one workgroup per (frame, level)) and, with
// ORBX_OCTREE_HOST_SIM, as a single-thread host simulation used by the CPU unit tests.
#pragma once
#include <stdint.h>

#ifdef ORBX_OCTREE_HOST_SIM
#include <cmath>
#define OT_DEV
// split steps by kind (0: on the count tables, 1: walking the keys), for the tests
static int ot_sim_sweeps[2];
#define OT_STAT(i) (++ot_sim_sweeps[i])
#define OT_HD static inline
#define OT_TID 0
#define OT_NT 1
#define OT_SYNC() ((void)0)
static inline int ot_atomic_add(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned ot_atomic_max(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline float ot_fdiv(float a, float b) { return a / b; }
static inline void ot_count(int* base, int idx) { if (idx >= 0) base[idx] += 1; }
#else
#define OT_DEV __device__ __forceinline__
#define OT_STAT(i) ((void)0)
#define OT_HD static __host__ __device__ inline
#define OT_TID ((int)threadIdx.x)
#define OT_NT ((int)blockDim.x)
#define OT_SYNC() __syncthreads()
static __device__ __forceinline__ int ot_atomic_add(int* p, int v) { return atomicAdd(p, v); }
static __device__ __forceinline__ unsigned ot_atomic_max(unsigned* p, unsigned v) { return atomicMax(p, v); }
static __device__ __forceinline__ float ot_fdiv(float a, float b) { return __fdiv_rn(a, b); }
// Inclusive prefix sum over the 64 lanes (all active) with DPP adds: shifts by 1, 2, 4, 8 inside each row of 16 lanes (a lane
// without a source adds 0), then the last lane of row 0 / 2 into row 1 / 3 and lane 31 into rows 2 and 3.  Six dependent
// vector instructions instead of six LDS round trips through ds_bpermute.
// (Written as instructions: from the builtin form the compiler makes three per step -- a zero, a DPP move, an add --
// where one add with a DPP operand does; a lane without a source adds 0 (bound_ctrl), a row outside the row mask keeps
// its value.  A DPP operand written by the previous vector instruction needs two wait states.)
static __device__ __forceinline__ int ot_wave_scan_incl(int v)
{
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"     // -> rows 1, 3
                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"           // -> rows 2, 3
                 : "+v"(v));
    return v;
}
// base[idx] += 1 for every lane with idx >= 0.  Every lane of the wavefront must call it (idx < 0: nothing to
// add).  Keys are stored in raster order, so neighbouring lanes mostly hit the same node and quadrant: each
// run of equal targets adds once, with the run's length, instead of up to 64 same-address atomics.
static __device__ __forceinline__ void ot_count(int* base, int idx)
{
    const int lane = threadIdx.x & 63;
    const int prev = __builtin_amdgcn_update_dpp(idx, idx, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps its own, and is a head anyway)
    const bool head = lane == 0 || idx != prev;
    const unsigned long long m = __ballot(head);
    const unsigned long long above = lane == 63 ? 0ull : m >> (lane + 1);
    const int len = above ? __ffsll((long long)above) : 64 - lane;      // lanes up to the next run
    if (head && idx >= 0) atomicAdd(&base[idx], len);
}
#endif

namespace orbx {

constexpr uint16_t kOtDead = 0xFFFF;
constexpr int kOtSingle = 1 << 30;   // flag on a child position: the child holds exactly one key

struct OtNode {            // 16 bytes
    int16_t x0, x1, y0, y1;
    int32_t cnt;           // keys inside
    uint32_t aux;          // cnt == 1: the packed key itself; cnt > 1: creation sequence number
};

// Shared (LDS) working set of one octree.  Capacities: nodes NC, keys live in global memory.
struct OtShared {
    OtNode* cur;           // [NC] current list, position = index
    OtNode* nxt;           // [NC]
    int32_t* cc;           // [NC*4] per candidate: child key counts, later child positions
    int32_t* candPos;      // [NC] list position of candidate e
    int32_t* pi;           // [NC] processing index of candidate e
    int32_t* cByPi;        // [NC] non-empty child count by processing index -> exclusive scan
    int32_t* keep;         // [NC] unsplit flag by position -> exclusive scan -> new position
    int32_t* posToCand;    // [NC] candidate index of a position (or -1)
    int32_t* cc2;          // [NC*2] 8-byte aligned scratch: 64-bit sort keys of the fine pass
    int32_t* scal;         // [16] scalars
    int32_t* scanTmp;      // [NT/64 + 2] block-scan scratch (device)
    int32_t* tcnt;         // [tcap] key counts of the full quadtree's cells down to the table depth (see ot_table_depth)
    uint16_t* tpos;        // [tcap] 0xFFFF: the cell's node has been split; else, once the keys are handed over, its list position
    uint16_t* ent0;        // [NC] table entry of every node of cur / nxt while the steps run on the tables (tcap > 0 only)
    uint16_t* ent1;        // [NC]
    int tcap;              // capacity of the two tables in entries (0: no tables)
    uint16_t* xs;          // [xycap] per key column / row: its share of the key's cell index at the tables' depth (see ot_build_axis_tables)
    uint16_t* ys;
    int xycap;             // entries of xs and ys together
};
enum { OT_N = 0, OT_E = 1, OT_C = 2, OT_RSTAR = 3, OT_NEXP = 4, OT_SEQ = 5, OT_TMP = 6, OT_NSPLIT = 7 };

// Count tables.  Every coarse sweep, and every fine step that does not reach N, splits every node with more than one key,
// so after t such steps those nodes are exactly the non-empty depth-t cells of the full quadtree under the roots, and a
// key's cell at any depth follows from its coordinates alone.  One pass over the keys therefore counts the cells of the
// deepest level D; sums give the levels above; the steps then work on node records only (their child counts are table
// look-ups), and one more pass at the end hands every key to its node and ranks it there.  Two passes over the keys
// instead of two to three per step.  A step that would split nodes of depth D hands the keys over first, and the steps
// from there on walk them.
// Level d (0 = the roots) holds nIni * 4^d cells, cell index = root * 4^d + quadrant path, levels stored back to back.
// Used from kOtTableMinKeysPerThread keys per thread on (below that a pass over the keys is a single round anyway).
constexpr int kOtTableMinKeysPerThread = 2;
OT_HD int ot_table_size(int nIni, int d) { long long s = 0, c = nIni; for (int i = 0; i <= d; ++i) { s += c; c *= 4; } return s > 0x7FFFFFFF ? 0x7FFFFFFF : (int)s; }
OT_HD int ot_table_off(int nIni, int d) { return d <= 0 ? 0 : ot_table_size(nIni, d - 1); }
// Depth of the tables: the first level with at least N cells (the steps stop when the list reaches N nodes, so only
// clustered key sets take them deeper), as far as the capacity allows.
OT_HD int ot_table_depth(int N, int nIni, int tcap)
{
    int d = 0; long long cells = nIni;
    while (cells < N && d < 7) { cells *= 4; ++d; }
    while (d > 0 && (ot_table_size(nIni, d) > tcap || ((long long)nIni << (2 * d)) > 65536)) --d;   // cell indices are 16-bit
    return d;
}

// exclusive prefix sum of a[0..n) in place; returns the total.  All threads call it.
static OT_DEV int ot_block_scan(int32_t* a, int n, int32_t* tmp)
{
#ifdef ORBX_OCTREE_HOST_SIM
    (void)tmp;
    int run = 0;
    for (int i = 0; i < n; ++i) { int v = a[i]; a[i] = run; run += v; }
    return run;
#else
    const int tid = OT_TID, nt = OT_NT;
    if (n <= nt) {
        // one element per thread: wave-level scan, then the totals of the waves in front
        const int lane1 = tid & 63, wave1 = tid >> 6;
        const int v = tid < n ? a[tid] : 0;
        const int inc1 = ot_wave_scan_incl(v);
        if (lane1 == 63) tmp[wave1] = inc1;
        __syncthreads();
        const int nw1 = (nt + 63) >> 6;
        int base1 = 0, total1 = 0;
        for (int w = 0; w < nw1; ++w) { const int t = tmp[w]; total1 += t; if (w < wave1) base1 += t; }
        if (tid < n) a[tid] = base1 + inc1 - v;
        __syncthreads();
        return total1;
    }
    const int chunk = (n + nt - 1) / nt;
    const int b = tid * chunk, e = min(b + chunk, n);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += a[i];
    // inclusive scan of per-thread sums inside the wave
    const int lane = tid & 63, wave = tid >> 6;
    const int inc = ot_wave_scan_incl(sum);
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    const int nw = (nt + 63) >> 6;
    if (tid == 0) { int run = 0; for (int w = 0; w < nw; ++w) { int v = tmp[w]; tmp[w] = run; run += v; } tmp[nw] = run; }
    __syncthreads();
    int run = tmp[wave] + inc - sum;
    for (int i = b; i < e; ++i) { int v = a[i]; a[i] = run; run += v; }
    const int total = tmp[nw];
    __syncthreads();
    return total;
#endif
}

static OT_DEV int ot_key_x(uint32_t k) { return (int)(k >> 20); }
static OT_DEV int ot_key_y(uint32_t k) { return (int)((k >> 8) & 0xFFF); }

// quadrant of a key inside node nd (DivideNode, :480-525): 0=n1 (top-left), 1=n2, 2=n3, 3=n4
static OT_DEV int ot_quadrant(const OtNode& nd, uint32_t key)
{
    const int hx = (nd.x1 - nd.x0 + 1) >> 1;   // ceil((x1-x0)/2.f)
    const int hy = (nd.y1 - nd.y0 + 1) >> 1;
    const int right = ot_key_x(key) >= nd.x0 + hx;
    const int bottom = ot_key_y(key) >= nd.y0 + hy;
    return right + 2 * bottom;
}

static OT_DEV void ot_child_box(const OtNode& p, int q, OtNode& c)
{
    const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1);
    const int my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
    c.x0 = (int16_t)((q & 1) ? mx : p.x0);
    c.x1 = (int16_t)((q & 1) ? p.x1 : mx);
    c.y0 = (int16_t)((q & 2) ? my : p.y0);
    c.y1 = (int16_t)((q & 2) ? p.y1 : my);
}

static OT_DEV int ot_root_of_x(int x, float hX, int nIni)
{
    const int r = (int)ot_fdiv((float)x, hX);
    return r > nIni - 1 ? nIni - 1 : r;
}
static OT_DEV int ot_key_root(uint32_t key, float hX, int nIni) { return ot_root_of_x(ot_key_x(key), hX, nIni); }

// The split lines of ot_child_box depend on one axis only: whether a key goes right at depth d follows from its column
// and the columns of its cell, whether it goes down from its row.  So a key's cell index at depth D (root * 4^D + two bits
// per level, right + 2 * bottom, first level on top) is xs[column] | ys[row], with xs holding the root and the `right` bits
// and ys the `bottom` bits, and its cell at depth d < D is that index >> 2 (D - d).  W + H short walks replace one per key.
static OT_DEV void ot_build_axis_tables(OtShared& s, int W, int H, int nIni, float hX, int D)
{
    const int tid = OT_TID, nt = OT_NT;
    for (int i = tid; i < W + H; i += nt) {
        const bool isx = i < W;
        const int v = isx ? i : i - W;
        int lo = 0, hi = (int16_t)H, idx = 0;
        if (isx) {
            const int r = ot_root_of_x(v, hX, nIni);
            lo = (int16_t)(int)(hX * (float)r); hi = (int16_t)(int)(hX * (float)(r + 1));
            idx = r << (2 * D);
        }
        for (int d = 0; d < D; ++d) {
            const int mid = lo + ((hi - lo + 1) >> 1);
            const int far = v >= mid;
            if (far) lo = mid; else hi = mid;
            idx |= far << (2 * (D - 1 - d) + (isx ? 0 : 1));
        }
        (isx ? s.xs : s.ys)[v] = (uint16_t)idx;
    }
}
static OT_DEV int ot_key_cell(const OtShared& s, uint32_t key) { return (int)s.xs[ot_key_x(key)] | (int)s.ys[ot_key_y(key)]; }

// Key storage.  Thread tid owns the keys tid, tid+nt, tid+2nt, ... ("slot" j = key tid + j*nt) for the whole
// run, and only the owner ever touches a key's node id.  The arrays live in LDS when a level's keys fit
// there and in device memory otherwise.
// LOOK: how many keys a thread looks up before it commits them (ot_for_keys4): 4 hides the LDS latency; 8 or 16 for
// arrays in device memory gained nothing at 3840 x 2160 (some ninety keys per thread and pass) and cost registers.
template <int LOOK>
struct OtMemKeysT {
    static constexpr int kLook = LOOK;
    const uint32_t* keys; uint16_t* node; int nk;
    OT_DEV uint32_t key(int, int k) const { return keys[k]; }
    OT_DEV uint16_t get(int, int k) const { return node[k]; }
    OT_DEV void set(int, int k, uint16_t v) { node[k] = v; }
};
typedef OtMemKeysT<4> OtMemKeys;
// f(slot, key index) for every key the calling thread owns
template <class KS, class F>
static OT_DEV void ot_for_keys(KS& ks, int tid, int nt, F&& f)
{
    int j = 0;
    for (int k = tid; k < ks.nk; k += nt, ++j) f(j, k);
}

// Two-phase form in which every thread runs the same number of rounds, so `commit` may use wavefront-wide
// operations: `look` (loads only; returns an int, -1 meaning nothing to do) runs for four keys before `commit`
// (stores, atomics) runs for them, and the four dependent load chains overlap instead of running back to back.
// Past the last key commit is called with -1.
template <class KS, class L, class C>
static OT_DEV void ot_for_keys4(KS& ks, int tid, int nt, L&& look, C&& commit)
{
    constexpr int U = KS::kLook;
    int j = 0;
    for (int k0 = 0; k0 < ks.nk; k0 += U * nt, j += U) {
        int v[U];
#ifndef ORBX_OCTREE_HOST_SIM
#pragma unroll
#endif
        for (int u = 0; u < U; ++u) { const int k = k0 + tid + u * nt; v[u] = k < ks.nk ? look(j + u, k) : -1; }
#ifndef ORBX_OCTREE_HOST_SIM
#pragma unroll
#endif
        for (int u = 0; u < U; ++u) commit(j + u, k0 + tid + u * nt, v[u]);
    }
}

// Coarse sweep (:605-664): every node with more than one key is split, in list order.  One packed prefix sum
// (children in the low half, unsplit nodes in the high half) places everything.
// TABLES: sweep number t of a run on the count tables: pc[p] is the table entry of node p (all nodes with more than one
// key are depth-t cells), the child counts are look-ups, pn receives the entries of the new list, and no key is touched.
template <bool TABLES, class KS>
static OT_DEV void ot_coarse_step(OtShared& s, KS& ks, int t = 0, int nIni = 0, const uint16_t* pc = nullptr, uint16_t* pn = nullptr)
{
    const int tid = OT_TID, nt = OT_NT;
    const int n = s.scal[OT_N];
    const int offT = TABLES ? ot_table_off(nIni, t) : 0, offC = TABLES ? ot_table_off(nIni, t + 1) : 0;
    if (TABLES) {
        for (int p = tid; p < n; p += nt)
            if (s.cur[p].cnt > 1) {
                const int32_t* c = &s.tcnt[offC + 4 * (pc[p] - offT)];
                s.cc[4 * p] = c[0]; s.cc[4 * p + 1] = c[1]; s.cc[4 * p + 2] = c[2]; s.cc[4 * p + 3] = c[3];
            }
        if (tid == 0) s.scal[OT_NEXP] = 0;
    } else {
        for (int i = tid; i < 4 * n; i += nt) s.cc[i] = 0;
        if (tid == 0) s.scal[OT_NEXP] = 0;
        OT_SYNC();
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int {
                const uint16_t p = ks.get(j, k);
                return p == kOtDead ? -1 : 4 * p + ot_quadrant(s.cur[p], ks.key(j, k));
            },
            [&](int, int, int target) { ot_count(s.cc, target); });
    }
    OT_SYNC();
    for (int p = tid; p < n; p += nt) {
        const int* c = &s.cc[4 * p];
        s.keep[p] = s.cur[p].cnt > 1 ? (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) : (1 << 16);
    }
    OT_SYNC();
    const int tot = ot_block_scan(s.keep, n, s.scanTmp);
    const int C = tot & 0xFFFF, nkeep = tot >> 16;
    const int seqBase = s.scal[OT_SEQ];
    for (int p = tid; p < n; p += nt) {
        const int cex = s.keep[p] & 0xFFFF, kex = s.keep[p] >> 16;
        if (s.cur[p].cnt > 1) {
            int* c = &s.cc[4 * p];
            const int cn = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
            const int bs = C - cex - cn;     // children of the node visited last come first
            int before = 0, nexp = 0;
            for (int q = 0; q < 4; ++q) {
                const int cnt = c[q];
                if (cnt > 0) {
                    const int np = bs + (cn - 1 - before);   // n4 sits at the front of the block
                    OtNode ch;
                    ot_child_box(s.cur[p], q, ch);
                    ch.cnt = cnt;
                    ch.aux = (uint32_t)(seqBase + cex + before);
                    s.nxt[np] = ch;
                    if (TABLES) pn[np] = (uint16_t)(offC + 4 * (pc[p] - offT) + q);
                    c[q] = cnt == 1 ? np | kOtSingle : np;   // the key of a one-key child is parked in the child
                    nexp += cnt > 1;
                    ++before;
                } else c[q] = -1;
            }
            if (nexp) ot_atomic_add(&s.scal[OT_NEXP], nexp);
            if (TABLES) s.tpos[pc[p]] = 0xFFFF;
        } else {
            s.nxt[C + kex] = s.cur[p];       // nodes with one key keep their relative order at the back
            if (TABLES) pn[C + kex] = pc[p];
        }
    }
    OT_SYNC();
    if (!TABLES)
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int {
                const uint16_t p = ks.get(j, k);
                if (p == kOtDead) return -1;
                return s.cc[4 * p + ot_quadrant(s.cur[p], ks.key(j, k))];
            },
            [&](int j, int k, int np) {
                if (np < 0) return;
                if (np & kOtSingle) { s.nxt[np & 0xFFFF].aux = ks.key(j, k); ks.set(j, k, kOtDead); }
                else ks.set(j, k, (uint16_t)np);
            });
    if (tid == 0) { s.scal[OT_N] = C + nkeep; s.scal[OT_SEQ] = seqBase + C; }
    OT_SYNC();
}

// A key's node in the current list while the steps run on the count tables: walking down from its root the key stops at
// the first cell that holds it alone (that node keeps the key in its record: kOtSingle is set) or whose node has not
// been split (tpos has the list position of every node of the list, and 0xFFFF for the cells whose nodes were split).
static OT_DEV int ot_walk_to_node(const OtShared& s, uint32_t key, int nIni, int D)
{
    const int cell = ot_key_cell(s, key);
    int off = 0, cells = nIni;
    for (int d = 0; d <= D; ++d) {
        const int e = off + (cell >> (2 * (D - d)));
        const int pos = s.tpos[e];
        if (s.tcnt[e] == 1) return pos | kOtSingle;
        if (pos != 0xFFFF) return pos;
        off += cells; cells *= 4;
    }
    return -1;      // not reached: no step on the tables splits a node of depth D
}

// End of a run on the count tables with steps still to come: every key goes to its node.
template <class KS>
static OT_DEV void ot_tables_to_keys(OtShared& s, KS& ks, int nIni, int D, const uint16_t* pc)
{
    const int tid = OT_TID, nt = OT_NT;
    const int n = s.scal[OT_N];
    for (int p = tid; p < n; p += nt) s.tpos[pc[p]] = (uint16_t)p;
    OT_SYNC();
    ot_for_keys4(ks, tid, nt,
        [&](int j, int k) -> int { return ot_walk_to_node(s, ks.key(j, k), nIni, D); },
        [&](int j, int k, int np) {
            if (np < 0) return;
            if (np & kOtSingle) { s.cur[np & 0xFFFF].aux = ks.key(j, k); ks.set(j, k, kOtDead); }
            else ks.set(j, k, (uint16_t)np);
        });
    OT_SYNC();
}

// One split step.  fine == 0: coarse sweep; fine == 1: sorted pass with the cut at N.
// Updates s.scal[OT_N], OT_NEXP, OT_SEQ; the caller swaps cur/nxt after the call.
// TABLES: step number t of a run on the count tables (see ot_coarse_step): child counts are look-ups, no key is touched.
template <bool TABLES, class KS>
static OT_DEV void ot_split_step(OtShared& s, KS& ks, int N, int fine, int t = 0, int nIni = 0, const uint16_t* pc = nullptr, uint16_t* pn = nullptr)
{
    const int tid = OT_TID, nt = OT_NT;
    const int n = s.scal[OT_N];
    const int offT = TABLES ? ot_table_off(nIni, t) : 0, offC = TABLES ? ot_table_off(nIni, t + 1) : 0;

    // candidates = nodes with more than one key, in list order
    for (int p = tid; p < n; p += nt) s.keep[p] = s.cur[p].cnt > 1 ? 1 : 0;
    OT_SYNC();
    const int E = ot_block_scan(s.keep, n, s.scanTmp);
    for (int p = tid; p < n; p += nt) {
        const int e = s.keep[p];
        if (s.cur[p].cnt > 1) {
            s.candPos[e] = p; s.posToCand[p] = e;
            const int32_t* c = TABLES ? &s.tcnt[offC + 4 * (pc[p] - offT)] : nullptr;
            s.cc[4 * e] = TABLES ? c[0] : 0; s.cc[4 * e + 1] = TABLES ? c[1] : 0; s.cc[4 * e + 2] = TABLES ? c[2] : 0; s.cc[4 * e + 3] = TABLES ? c[3] : 0;
        }
        else s.posToCand[p] = -1;
    }
    OT_SYNC();
    // child key counts of every candidate
    if (!TABLES) {
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int {
                const uint16_t p = ks.get(j, k);
                return p == kOtDead ? -1 : 4 * s.posToCand[p] + ot_quadrant(s.cur[p], ks.key(j, k));
            },
            [&](int, int, int target) { ot_count(s.cc, target); });
        OT_SYNC();
    }
    // processing order
    if (!fine) {
        for (int e = tid; e < E; e += nt) s.pi[e] = e;
        if (tid == 0) s.scal[OT_RSTAR] = E - 1;
        OT_SYNC();
    } else {
        // rank by (count desc, sequence desc): the reference sorts ascending and walks from the back (:683-684)
        // (count and sequence packed into one 64-bit sort key per candidate first: one load and one compare per pair)
        unsigned long long* rk = reinterpret_cast<unsigned long long*>(s.cc2);
        for (int e = tid; e < E; e += nt) { const OtNode& a = s.cur[s.candPos[e]]; rk[e] = ((unsigned long long)(uint32_t)a.cnt << 32) | a.aux; }
        OT_SYNC();
        for (int e = tid; e < E; e += nt) {
            const unsigned long long a = rk[e];
            int r = 0;
            for (int m = 0; m < E; ++m) r += rk[m] > a;
            s.pi[e] = r;
        }
        OT_SYNC();
        // growth by rank, inclusive prefix, first rank at which the list reaches N (:729-730)
        for (int e = tid; e < E; e += nt) {
            const int* c = &s.cc[4 * e];
            s.cByPi[s.pi[e]] = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) - 1;
        }
        if (tid == 0) s.scal[OT_RSTAR] = E - 1;
        OT_SYNC();
        ot_block_scan(s.cByPi, E, s.scanTmp);  // exclusive prefix of growth
        for (int e = tid; e < E; e += nt) {
            const int r = s.pi[e];
            const int* c = &s.cc[4 * e];
            const int g = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) - 1;
            if (n + s.cByPi[r] + g >= N) {
#ifdef ORBX_OCTREE_HOST_SIM
                if (r < s.scal[OT_RSTAR]) s.scal[OT_RSTAR] = r;
#else
                atomicMin(&s.scal[OT_RSTAR], r);
#endif
            }
        }
        OT_SYNC();
    }
    const int rstar = s.scal[OT_RSTAR];
    // non-empty child count by processing index (only the split ones), exclusive scan
    for (int e = tid; e < E; e += nt) {
        const int* c = &s.cc[4 * e];
        if (s.pi[e] <= rstar) s.cByPi[s.pi[e]] = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
    }
    OT_SYNC();
    const int nsplit = rstar + 1 < E ? rstar + 1 : E;
    const int C = ot_block_scan(s.cByPi, nsplit, s.scanTmp);
    // unsplit nodes keep their relative order behind the new children
    for (int p = tid; p < n; p += nt) {
        const int e = s.posToCand[p];
        s.keep[p] = (e >= 0 && s.pi[e] <= rstar) ? 0 : 1;
    }
    OT_SYNC();
    ot_block_scan(s.keep, n, s.scanTmp);
    if (tid == 0) s.scal[OT_NEXP] = 0;
    OT_SYNC();
    const int seqBase = s.scal[OT_SEQ];
    // write the new list
    for (int p = tid; p < n; p += nt) {
        const int e = s.posToCand[p];
        if (e >= 0 && s.pi[e] <= rstar) {
            int* c = &s.cc[4 * e];
            const int cex = s.cByPi[s.pi[e]];
            const int cn = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
            const int bs = C - cex - cn;     // block of the node processed last comes first
            int before = 0, nexp = 0;
            for (int q = 0; q < 4; ++q) {
                const int cnt = c[q];
                if (cnt > 0) {
                    const int after = cn - 1 - before;        // n4 sits at the front of the block
                    const int np = bs + after;
                    OtNode ch;
                    ot_child_box(s.cur[p], q, ch);
                    ch.cnt = cnt;
                    ch.aux = (uint32_t)(seqBase + cex + before);
                    s.nxt[np] = ch;
                    if (TABLES) pn[np] = (uint16_t)(offC + 4 * (pc[p] - offT) + q);
                    c[q] = np;
                    nexp += cnt > 1;
                    ++before;
                } else c[q] = -1;
            }
            if (nexp) ot_atomic_add(&s.scal[OT_NEXP], nexp);
            if (TABLES) s.tpos[pc[p]] = 0xFFFF;
        } else {
            const int np = C + s.keep[p];
            s.nxt[np] = s.cur[p];
            if (TABLES) pn[np] = pc[p];
            s.keep[p] = np;
        }
    }
    OT_SYNC();
    // keys follow their nodes
    if (!TABLES)
    ot_for_keys4(ks, tid, nt,
        [&](int j, int k) -> int {
            const uint16_t p = ks.get(j, k);
            if (p == kOtDead) return -1;
            const int e = s.posToCand[p];
            if (e >= 0 && s.pi[e] <= rstar) {
                const int np = s.cc[4 * e + ot_quadrant(s.cur[p], ks.key(j, k))];
                return s.nxt[np].cnt == 1 ? np | kOtSingle : np;
            }
            return s.keep[p];
        },
        [&](int j, int k, int np) {
            if (np < 0) return;
            if (np & kOtSingle) { s.nxt[np & 0xFFFF].aux = ks.key(j, k); ks.set(j, k, kOtDead); }
            else ks.set(j, k, (uint16_t)np);
        });
    OT_SYNC();
    if (tid == 0) {
        s.scal[OT_N] = C + (n - nsplit);
        s.scal[OT_SEQ] = seqBase + C;
        s.scal[OT_NSPLIT] = nsplit;
    }
    OT_SYNC();
}

// Whole DistributeOctTree for one key set held by the key store ks (keys packed x<<20|y<<8|score,
// coordinates relative to minBorder; keys: the same keys in memory, read at the end for the winners).
// H: maxY-minY, nIni/hX: roots (:542-544).  out receives the selected keys in list order; returns their
// number (all threads).
template <class KS>
static OT_DEV int ot_distribute(OtShared& s, KS& ks, const uint32_t* keys,
                                int W, int H, int nIni, float hX, int N, uint32_t* out, int outCap)
{
    const int tid = OT_TID, nt = OT_NT;
    // the tables pay from two keys per thread on (measured per 256 frames of 640x480 98 -> 95 us, 1241x376 188 -> 154,
    // per 128 of 1920x1080 431 -> 284, per 64 of 3840x2160 1204 -> 539); below that a key pass is a single round anyway
    const int D = ks.nk >= kOtTableMinKeysPerThread * nt && W + H <= s.xycap ? ot_table_depth(N, nIni, s.tcap) : 0;
    uint16_t* pc = s.ent0;          // table entries of the nodes of cur / nxt while the steps run on the count tables
    uint16_t* pn = s.ent1;
    bool tables = D > 0;
    if (tables) s.ys = s.xs + W;
    // roots (:551-569)
    for (int i = tid; i < nIni; i += nt) {
        OtNode r;
        r.x0 = (int16_t)(int)(hX * (float)i);
        r.x1 = (int16_t)(int)(hX * (float)(i + 1));
        r.y0 = 0; r.y1 = (int16_t)H;
        r.cnt = 0; r.aux = (uint32_t)i;
        s.nxt[i] = r;
        s.cc[i] = 0;
    }
    if (tid == 0) { s.scal[OT_SEQ] = nIni; }
    if (tables) {
        const int total = ot_table_size(nIni, D);
        for (int i = tid; i < total; i += nt) s.tcnt[i] = 0;
        ot_build_axis_tables(s, W, H, nIni, hX, D);
        OT_SYNC();
        int32_t* deepest = s.tcnt + ot_table_off(nIni, D);
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int { return ot_key_cell(s, ks.key(j, k)); },
            [&](int, int, int idx) { ot_count(deepest, idx); });
        OT_SYNC();
        for (int d = D - 1; d >= 0; --d) {
            const int32_t* below = s.tcnt + ot_table_off(nIni, d + 1);
            int32_t* here = s.tcnt + ot_table_off(nIni, d);
            const int cells = ot_table_size(nIni, d) - ot_table_off(nIni, d);
            for (int i = tid; i < cells; i += nt) here[i] = below[4 * i] + below[4 * i + 1] + below[4 * i + 2] + below[4 * i + 3];
            OT_SYNC();
        }
        for (int i = tid; i < nIni; i += nt) s.cc[i] = s.tcnt[i];
    } else {
        OT_SYNC();
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int { return ot_key_root(ks.key(j, k), hX, nIni); },
            [&](int j, int k, int r) {
                if (r >= 0) ks.set(j, k, (uint16_t)r);
                ot_count(s.cc, r);
            });
    }
    OT_SYNC();
    // drop empty roots (:571-584); a root with one key keeps it
    for (int i = tid; i < nIni; i += nt) { s.nxt[i].cnt = s.cc[i]; s.keep[i] = s.cc[i] > 0 ? 1 : 0; }
    OT_SYNC();
    const int n0 = ot_block_scan(s.keep, nIni, s.scanTmp);
    for (int i = tid; i < nIni; i += nt)
        if (s.nxt[i].cnt > 0) { s.cur[s.keep[i]] = s.nxt[i]; if (tables) pc[s.keep[i]] = (uint16_t)i; }
    OT_SYNC();
    if (!tables)
        ot_for_keys(ks, tid, nt, [&](int j, int k) {
            const int np = s.keep[ks.get(j, k)];
            if (s.cur[np].cnt == 1) { s.cur[np].aux = ks.key(j, k); ks.set(j, k, kOtDead); }
            else ks.set(j, k, (uint16_t)np);
        });
    if (tid == 0) s.scal[OT_N] = n0;
    OT_SYNC();

    // main loop (:593-738)
    bool finish = false;
    int sweeps = 0;
    while (!finish) {
        const int prevSize = s.scal[OT_N];
        OT_SYNC();
        // a step on the tables needs the counts one level below its nodes: past the tables' depth the keys take over
        if (tables && sweeps == D) { ot_tables_to_keys(s, ks, nIni, D, pc); tables = false; }
        if (tables) {
            ot_coarse_step<true>(s, ks, sweeps, nIni, pc, pn);
            uint16_t* t = pc; pc = pn; pn = t;
            OT_STAT(0);
        } else {
            ot_coarse_step<false>(s, ks);
            OT_STAT(1);
        }
        { OtNode* t = s.cur; s.cur = s.nxt; s.nxt = t; }
        ++sweeps;
        const int n = s.scal[OT_N], nToExpand = s.scal[OT_NEXP];
        if (n >= N || n == prevSize) finish = true;
        else if (n + nToExpand * 3 > N) {
            // (a fine step that does not reach N splits every node with more than one key, as a coarse sweep does: those
            // nodes stay the cells of one depth, which is all a step on the tables relies on)
            while (!finish) {
                const int prev2 = s.scal[OT_N];
                OT_SYNC();
                if (tables && sweeps == D) { ot_tables_to_keys(s, ks, nIni, D, pc); tables = false; }
                if (tables) {
                    ot_split_step<true>(s, ks, N, 1, sweeps, nIni, pc, pn);
                    uint16_t* t = pc; pc = pn; pn = t;
                    OT_STAT(0);
                } else {
                    ot_split_step<false>(s, ks, N, 1);
                    OT_STAT(1);
                }
                { OtNode* t = s.cur; s.cur = s.nxt; s.nxt = t; }
                ++sweeps;
                const int n2 = s.scal[OT_N];
                if (n2 >= N || n2 == prev2) finish = true;
            }
        }
    }
    OT_SYNC();
    // best key per node (:741-759): largest response, first in input order on ties
    const int n = s.scal[OT_N];
    unsigned* best = reinterpret_cast<unsigned*>(s.keep);
    for (int p = tid; p < n; p += nt) { best[p] = 0; if (tables) s.tpos[pc[p]] = (uint16_t)p; }
    OT_SYNC();
    if (tables)       // the run ended on the tables: the keys find their nodes and compete in the same pass
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int { return ot_walk_to_node(s, ks.key(j, k), nIni, D); },
            [&](int j, int k, int np) {
                if (np < 0) return;
                const uint32_t key = ks.key(j, k);
                if (np & kOtSingle) s.cur[np & 0xFFFF].aux = key;
                else {
                    // (most keys lose against what their node already holds: a plain read spares them the atomic, and
                    // same-address atomics of one wavefront run one after the other)
                    const unsigned v = ((key & 0xFFu) << 24) | (0xFFFFFFu - (unsigned)k);
                    if (v > best[np]) ot_atomic_max(&best[np], v);
                }
            });
    else
        ot_for_keys4(ks, tid, nt,
            [&](int j, int k) -> int {
                const uint16_t p = ks.get(j, k);
                return p == kOtDead ? -1 : (int)(((unsigned)p << 8) | (ks.key(j, k) & 0xFFu));      // node, response
            },
            [&](int, int k, int v) {
                if (v >= 0) {
                    const unsigned c = ((unsigned)(v & 0xFF) << 24) | (0xFFFFFFu - (unsigned)k);
                    if (c > best[v >> 8]) ot_atomic_max(&best[v >> 8], c);
                }
            });
    OT_SYNC();
    for (int p = tid; p < n && p < outCap; p += nt) {
        const OtNode& nd = s.cur[p];
        out[p] = nd.cnt == 1 ? nd.aux : keys[0xFFFFFFu - (best[p] & 0xFFFFFFu)];
    }
    OT_SYNC();
    return n;
}

// The steps of DistributeOctTree on the count tables alone, for a caller that counts, hands over and ranks the keys
// itself: on entry the tables' deepest level (depth D > 0) holds the key counts of its cells.  Runs the same steps as
// ot_distribute does while it is on its tables and then writes, for every cell of depth D that holds a key, the list
// position of the node its keys belong to (leaf[cell]: the first cell on the way down from the root that is a node of
// the final list).  The winner of a node is then the largest (response, first in input order) among the keys that map
// to it, whether the node holds one key or many.  Returns the length of the list, or -1 when a step would have to
// split nodes of depth D (clustered keys): such a level goes through ot_distribute with its keys instead.
static OT_DEV int ot_distribute_tables(OtShared& s, int H, int nIni, float hX, int N, int D, uint16_t* leaf)
{
    const int tid = OT_TID, nt = OT_NT;
    uint16_t* pc = s.ent0;
    uint16_t* pn = s.ent1;
    OtMemKeys ks{nullptr, nullptr, 0};        // steps on the tables never touch a key
    for (int i = tid; i < nIni; i += nt) {
        OtNode r;
        r.x0 = (int16_t)(int)(hX * (float)i);
        r.x1 = (int16_t)(int)(hX * (float)(i + 1));
        r.y0 = 0; r.y1 = (int16_t)H;
        r.cnt = 0; r.aux = (uint32_t)i;
        s.nxt[i] = r;
    }
    if (tid == 0) { s.scal[OT_SEQ] = nIni; }
    const int offD = ot_table_off(nIni, D), cellsD = ot_table_size(nIni, D) - offD;
    OT_SYNC();
    for (int d = D - 1; d >= 0; --d) {
        const int32_t* below = s.tcnt + ot_table_off(nIni, d + 1);
        int32_t* here = s.tcnt + ot_table_off(nIni, d);
        const int cells = ot_table_size(nIni, d) - ot_table_off(nIni, d);
        for (int i = tid; i < cells; i += nt) here[i] = below[4 * i] + below[4 * i + 1] + below[4 * i + 2] + below[4 * i + 3];
        OT_SYNC();
    }
    // drop empty roots (:571-584)
    for (int i = tid; i < nIni; i += nt) { s.nxt[i].cnt = s.tcnt[i]; s.keep[i] = s.tcnt[i] > 0 ? 1 : 0; }
    OT_SYNC();
    const int n0 = ot_block_scan(s.keep, nIni, s.scanTmp);
    for (int i = tid; i < nIni; i += nt)
        if (s.nxt[i].cnt > 0) { s.cur[s.keep[i]] = s.nxt[i]; pc[s.keep[i]] = (uint16_t)i; }
    if (tid == 0) s.scal[OT_N] = n0;
    OT_SYNC();
    // main loop (:593-738), as in ot_distribute
    bool finish = n0 == 0;
    int sweeps = 0;
    while (!finish) {
        const int prevSize = s.scal[OT_N];
        OT_SYNC();
        if (sweeps == D) return -1;
        ot_coarse_step<true>(s, ks, sweeps, nIni, pc, pn);
        { uint16_t* t = pc; pc = pn; pn = t; }
        { OtNode* t = s.cur; s.cur = s.nxt; s.nxt = t; }
        ++sweeps;
        const int n = s.scal[OT_N], nToExpand = s.scal[OT_NEXP];
        if (n >= N || n == prevSize) finish = true;
        else if (n + nToExpand * 3 > N) {
            while (!finish) {
                const int prev2 = s.scal[OT_N];
                OT_SYNC();
                if (sweeps == D) return -1;
                ot_split_step<true>(s, ks, N, 1, sweeps, nIni, pc, pn);
                { uint16_t* t = pc; pc = pn; pn = t; }
                { OtNode* t = s.cur; s.cur = s.nxt; s.nxt = t; }
                ++sweeps;
                const int n2 = s.scal[OT_N];
                if (n2 >= N || n2 == prev2) finish = true;
            }
        }
    }
    OT_SYNC();
    const int n = s.scal[OT_N];
    for (int p = tid; p < n; p += nt) s.tpos[pc[p]] = (uint16_t)p;
    OT_SYNC();
    for (int i = tid; i < cellsD; i += nt) {
        if (s.tcnt[offD + i] == 0) continue;
        int off = 0, cells = nIni;
        for (int d = 0; d <= D; ++d) {
            const int e = off + (i >> (2 * (D - d)));
            const int pos = s.tpos[e];
            if (s.tcnt[e] == 1 || pos != 0xFFFF) { leaf[i] = (uint16_t)pos; break; }
            off += cells; cells *= 4;
        }
    }
    OT_SYNC();
    return n;
}

} // namespace orbx
