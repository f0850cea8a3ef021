// MI355X (gfx950) kernels of the two Hamming consumers that follow extraction in the reference:
//   - MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:483-548): per map point the observation whose
//     descriptor has the least median distance to the others;
//   - the DBoW2 vocabulary descent behind Frame::ComputeBoW (src/Frame.cc:395-402):
//     TemplatedVocabulary::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1272) with FORB::distance
//     (FORB.cpp:81-101).
// Both are popcount work on 32-byte rows (8 x v_xor_b32 + v_bcnt_u32_b32 per pair), bound by instruction issue and by
// the latency of dependent loads down the tree; nothing here is a dense contraction, so no MFMA.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <mutex>
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

__device__ __forceinline__ int hamming256(const uint32_t (&a)[8], const uint4 lo, const uint4 hi)
{
    return __popc(a[0] ^ lo.x) + __popc(a[1] ^ lo.y) + __popc(a[2] ^ lo.z) + __popc(a[3] ^ lo.w) +
           __popc(a[4] ^ hi.x) + __popc(a[5] ^ hi.y) + __popc(a[6] ^ hi.z) + __popc(a[7] ^ hi.w);
}

// ---- distinctive descriptors -------------------------------------------------------------------------------------
// One thread per descriptor row.  Rows of a group sit in adjacent lanes, so the group's descriptors are read from the
// same few cache lines by all of them.  The median of a row of the distance matrix is found without storing the row:
// a bisection on the value (0..256) that recounts the distances not above the pivot, nine passes over the group.
__global__ __launch_bounds__(256) void k_distinctive_median(const uint4* __restrict__ desc, const int32_t* __restrict__ group_ptr,
                                                            int n_groups, long long n_rows, int32_t* __restrict__ median)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    int lo = 0, hi = n_groups;                       // last group whose first row is <= row
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((long long)group_ptr[mid] <= row) lo = mid; else hi = mid; }
    const int b = group_ptr[lo], e = group_ptr[lo + 1];
    if (row < b || row >= e) { median[row] = -1; return; }     // a row outside every group (group_ptr[0] > 0 or a gap)
    const int n = e - b, kth = (n - 1) >> 1;                   // sorted[(int)(0.5*(n-1))], src/MapPoint.cc:535
    uint32_t mine[8];
    { const uint4 a = desc[2 * row], c = desc[2 * row + 1];
      mine[0] = a.x; mine[1] = a.y; mine[2] = a.z; mine[3] = a.w; mine[4] = c.x; mine[5] = c.y; mine[6] = c.z; mine[7] = c.w; }
    int vlo = 0, vhi = 256;                                    // least v with #{j : d(row, j) <= v} > kth
    while (vlo < vhi) {
        const int v = (vlo + vhi) >> 1;
        int cnt = 0;
        for (int j = b; j < e; ++j) cnt += hamming256(mine, desc[2 * (long long)j], desc[2 * (long long)j + 1]) <= v ? 1 : 0;
        if (cnt > kth) vhi = v; else vlo = v + 1;
    }
    median[row] = vlo;
}

// one thread per group: the first row with the least median (strict '<' of src/MapPoint.cc:537)
__global__ __launch_bounds__(256) void k_distinctive_pick(const int32_t* __restrict__ median, const int32_t* __restrict__ group_ptr,
                                                          int n_groups, int32_t* __restrict__ best)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const int b = group_ptr[g], e = group_ptr[g + 1];
    int bm = 0x7FFFFFFF, bi = -1;
    for (int j = b; j < e; ++j) { const int m = median[j]; if (m < bm) { bm = m; bi = j - b; } }
    best[g] = bi;
}

// ---- vocabulary descent ------------------------------------------------------------------------------------------
// Sixteen lanes per feature: lane c of the group measures child c, c + 16, ... of the current node (the children of a
// node are 32-byte rows stored back to back, so a k = 10 node is 320 contiguous bytes), then a 4-step butterfly over
// keys (distance << 16 | child position) leaves the nearest child, the first one on ties, in every lane.
__global__ __launch_bounds__(256) void k_voc_descend(const uint32_t* __restrict__ desc, const int32_t* __restrict__ counts, int cap,
                                                     long long n_rows, const int32_t* __restrict__ child_first,
                                                     const uint4* __restrict__ slot_desc, const int32_t* __restrict__ slot_node,
                                                     const int32_t* __restrict__ word_of, int nid_level,
                                                     int32_t* __restrict__ word, int32_t* __restrict__ node)
{
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n_rows) return;
    if (counts) {
        const long long f = row / cap;
        if (row - f * cap >= counts[f]) { if (sub == 0) { word[row] = -1; node[row] = -1; } return; }
    }
    uint32_t q[8];
    { const uint4* qp = reinterpret_cast<const uint4*>(desc + row * 8);
      const uint4 a = qp[0], c = qp[1];
      q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = c.x; q[5] = c.y; q[6] = c.z; q[7] = c.w; }
    int cur = 0, level = 0, nid = nid_level <= 0 ? 0 : -1;
    int first = child_first[0], nch = child_first[1] - first;
    while (nch > 0) {                                          // uniform over the 16 lanes of a feature
        uint32_t key = 0xFFFFFFFFu;
        for (int c = sub; c < nch; c += 16) {
            const uint4 lo = slot_desc[2 * (size_t)(first + c)], hi = slot_desc[2 * (size_t)(first + c) + 1];
            key = min(key, ((uint32_t)hamming256(q, lo, hi) << 16) | (uint32_t)c);
        }
        // minimum over the feature's 16 lanes (one DPP row): lane pairs, quad halves, rotations by 4 and 8
        key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xF, 0xF, false));    // quad_perm:[1,0,3,2]
        key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x4E, 0xF, 0xF, false));    // quad_perm:[2,3,0,1]
        key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x124, 0xF, 0xF, false));   // row_ror:4
        key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x128, 0xF, 0xF, false));   // row_ror:8
        cur = slot_node[first + (int)(key & 0xFFFFu)];
        if (++level == nid_level) nid = cur;
        first = child_first[cur];
        nch = child_first[cur + 1] - first;
    }
    if (sub == 0) { word[row] = word_of[cur]; node[row] = nid >= 0 ? nid : cur; }
}

} // namespace

struct orbx_vocabulary {
    int device = 0;
    int k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0;
    int32_t* d_child_first = nullptr; uint4* d_slot_desc = nullptr; int32_t* d_slot_node = nullptr; int32_t* d_word_of = nullptr;
    std::vector<double> word_weight;                 // weight of the leaf behind each word
    // staging of the host entry point; one caller at a time (Frame::ComputeBoW and KeyFrame::ComputeBoW run in
    // different threads of the reference against the one shared vocabulary)
    std::mutex mu;
    hipStream_t stream = nullptr;
    void* d_stage = nullptr; size_t stage_bytes = 0;
    int32_t* h_ids = nullptr; size_t h_ids_rows = 0;  // pinned: word ids then node ids
};

static int launch_descend(const orbx_vocabulary* v, const uint8_t* d_desc, const int32_t* d_counts, int cap, long long n_rows,
                          int levelsup, int32_t* d_word, int32_t* d_node, hipStream_t st)
{
    if (n_rows == 0) return ORBX_OK;
    const long long blocks = (n_rows + 15) / 16;
    if (blocks > 0x7FFFFFFFLL) { set_error("too many features for one launch"); return ORBX_ERR_BAD_ARG; }
    hipLaunchKernelGGL(k_voc_descend, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(d_desc), d_counts, cap,
                       n_rows, v->d_child_first, v->d_slot_desc, v->d_slot_node, v->d_word_of, v->L - levelsup, d_word, d_node);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

extern "C" {

int orbx_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes,
                           const int32_t* parent, const uint8_t* is_leaf, const uint8_t* descriptors, const double* weight,
                           orbx_vocabulary** out)
{
    if (!out) return ORBX_ERR_BAD_ARG;
    *out = nullptr;
    // the ranges TemplatedVocabulary::loadFromTextFile accepts (TemplatedVocabulary.h:1372)
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3 || n_nodes < 2 ||
        !parent || !is_leaf || !descriptors || !weight) { set_error("orbx_vocabulary_create: bad argument"); return ORBX_ERR_BAD_ARG; }
    std::vector<int32_t> first((size_t)n_nodes + 1, 0);
    for (int i = 1; i < n_nodes; ++i) {
        if (parent[i] < 0 || parent[i] >= i) { set_error("vocabulary: a node must follow its parent"); return ORBX_ERR_BAD_ARG; }
        first[(size_t)parent[i] + 1]++;
    }
    for (int i = 0; i < n_nodes; ++i) first[(size_t)i + 1] += first[i];
    std::vector<int32_t> fill(first.begin(), first.end() - 1), slot_node((size_t)n_nodes - 1), word_of((size_t)n_nodes, -1);
    std::vector<uint8_t> slot_desc((size_t)(n_nodes - 1) * 32);
    for (int i = 1; i < n_nodes; ++i) {
        const int s = fill[parent[i]]++;
        slot_node[s] = i;
        std::memcpy(&slot_desc[(size_t)s * 32], descriptors + (size_t)i * 32, 32);
    }
    orbx_vocabulary* v = new orbx_vocabulary();
    struct Guard { orbx_vocabulary* p; ~Guard() { orbx_vocabulary_destroy(p); } } guard{v};
    for (int i = 1; i < n_nodes; ++i) {
        const bool leaf = is_leaf[i] != 0, childless = first[(size_t)i + 1] == first[i];
        if (leaf != childless) { set_error("vocabulary: leaf flag disagrees with the children lists"); return ORBX_ERR_BAD_ARG; }
        if (leaf) { word_of[i] = v->n_words++; v->word_weight.push_back(weight[i]); }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return ORBX_ERR_NO_DEVICE; }
    if (device >= count) { set_error("device index out of range"); return ORBX_ERR_BAD_ARG; }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipGetDevice(&v->device));
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_nodes = n_nodes;
    HIP_CHECK(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    HIP_CHECK(hipMalloc(&v->d_child_first, sizeof(int32_t) * first.size()));
    HIP_CHECK(hipMalloc(&v->d_slot_desc, slot_desc.size()));
    HIP_CHECK(hipMalloc(&v->d_slot_node, sizeof(int32_t) * slot_node.size()));
    HIP_CHECK(hipMalloc(&v->d_word_of, sizeof(int32_t) * word_of.size()));
    HIP_CHECK(hipMemcpy(v->d_child_first, first.data(), sizeof(int32_t) * first.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(v->d_slot_desc, slot_desc.data(), slot_desc.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(v->d_slot_node, slot_node.data(), sizeof(int32_t) * slot_node.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(v->d_word_of, word_of.data(), sizeof(int32_t) * word_of.size(), hipMemcpyHostToDevice));
    guard.p = nullptr;
    *out = v;
    return ORBX_OK;
}

void orbx_vocabulary_destroy(orbx_vocabulary* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    void* ptrs[] = {v->d_child_first, v->d_slot_desc, v->d_slot_node, v->d_word_of, v->d_stage};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (v->h_ids) (void)hipHostFree(v->h_ids);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

int orbx_vocabulary_words(const orbx_vocabulary* v) { return v ? v->n_words : 0; }

int orbx_vocabulary_descend_device(orbx_vocabulary* v, const uint8_t* d_desc, const int32_t* d_counts, int cap_per_frame,
                                   int n_frames, int levelsup, int32_t* d_word, int32_t* d_node, void* stream)
{
    if (!v || !d_desc || !d_word || !d_node || cap_per_frame < 1 || n_frames < 0) { set_error("orbx_vocabulary_descend_device: bad argument"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(v->device));
    return launch_descend(v, d_desc, d_counts, cap_per_frame, (long long)cap_per_frame * n_frames, levelsup, d_word, d_node, (hipStream_t)stream);
}

int orbx_vocabulary_transform(orbx_vocabulary* v, const uint8_t* desc, int n, int levelsup,
                              uint32_t* bow_ids, double* bow_values, int32_t* n_words,
                              uint32_t* fv_node_ids, int32_t* fv_node_ptr, int32_t* fv_features, int32_t* fv_n_nodes)
{
    if (!v || n < 0 || !n_words || !fv_node_ptr || !fv_n_nodes) { set_error("orbx_vocabulary_transform: bad argument"); return ORBX_ERR_BAD_ARG; }
    *n_words = 0; *fv_n_nodes = 0; fv_node_ptr[0] = 0;
    if (n == 0) return ORBX_OK;
    if (!desc || !bow_ids || !bow_values || !fv_node_ids || !fv_features) { set_error("orbx_vocabulary_transform: null pointer"); return ORBX_ERR_BAD_ARG; }
    std::vector<int32_t> ids((size_t)n * 2);
    {
        std::lock_guard<std::mutex> lock(v->mu);
        HIP_CHECK(hipSetDevice(v->device));
        const size_t need = (size_t)n * 32 + sizeof(int32_t) * 2 * (size_t)n;
        if (v->stage_bytes < need) {
            if (v->d_stage) (void)hipFree(v->d_stage);
            v->d_stage = nullptr; v->stage_bytes = 0;
            HIP_CHECK(hipMalloc(&v->d_stage, need));
            v->stage_bytes = need;
        }
        if (v->h_ids_rows < (size_t)n) {
            if (v->h_ids) (void)hipHostFree(v->h_ids);
            v->h_ids = nullptr; v->h_ids_rows = 0;
            HIP_CHECK(hipHostMalloc((void**)&v->h_ids, sizeof(int32_t) * 2 * (size_t)n + (size_t)n * 32, hipHostMallocDefault));
            v->h_ids_rows = (size_t)n;
        }
        uint8_t* h_desc = reinterpret_cast<uint8_t*>(v->h_ids + 2 * v->h_ids_rows);
        std::memcpy(h_desc, desc, (size_t)n * 32);
        uint8_t* d_desc = static_cast<uint8_t*>(v->d_stage);
        int32_t* d_ids = reinterpret_cast<int32_t*>(d_desc + (size_t)n * 32);
        HIP_CHECK(hipMemcpyAsync(d_desc, h_desc, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
        const int rc = launch_descend(v, d_desc, nullptr, n, n, levelsup, d_ids, d_ids + n, v->stream);
        if (rc) return rc;
        HIP_CHECK(hipMemcpyAsync(v->h_ids, d_ids, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyDeviceToHost, v->stream));
        HIP_CHECK(hipStreamSynchronize(v->stream));
        std::memcpy(ids.data(), v->h_ids, sizeof(int32_t) * 2 * (size_t)n);
    }
    const int32_t* word = ids.data(); const int32_t* node = ids.data() + n;
    // features whose word is not stopped (weight > 0, TemplatedVocabulary.h:1170), ordered by word / by node and then
    // by feature index: the order in which a std::map keyed by id meets them
    std::vector<int32_t> kept;
    kept.reserve((size_t)n);
    for (int i = 0; i < n; ++i) if (v->word_weight[(size_t)word[i]] > 0) kept.push_back(i);
    std::vector<int32_t> by_word(kept), by_node(kept);
    std::stable_sort(by_word.begin(), by_word.end(), [&](int32_t a, int32_t b) { return word[a] < word[b]; });
    std::stable_sort(by_node.begin(), by_node.end(), [&](int32_t a, int32_t b) { return node[a] < node[b]; });
    const bool tf = v->weighting == 0 || v->weighting == 1;          // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    int nw = 0;
    for (size_t p = 0; p < by_word.size();) {
        const int32_t w = word[by_word[p]];
        const double wt = v->word_weight[(size_t)w];
        double sum = wt;                                             // BowVector::addWeight, BowVector.cpp:34-46
        size_t e = p + 1;
        for (; e < by_word.size() && word[by_word[e]] == w; ++e) if (tf) sum += wt;
        bow_ids[nw] = (uint32_t)w; bow_values[nw] = sum; ++nw;
        p = e;
    }
    const bool must = v->scoring != 5;                               // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    if (tf && nw > 0 && !must) { const double nd = nw; for (int i = 0; i < nw; ++i) bow_values[i] /= nd; }   // :1177-1183
    if (must) {                                                      // BowVector::normalize, BowVector.cpp:62-84
        double norm = 0.0;
        if (v->scoring != 1) { for (int i = 0; i < nw; ++i) norm += std::fabs(bow_values[i]); }
        else { for (int i = 0; i < nw; ++i) norm += bow_values[i] * bow_values[i]; norm = std::sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < nw; ++i) bow_values[i] /= norm;
    }
    int nn = 0;
    for (size_t p = 0; p < by_node.size(); ++p) {
        const int32_t nd = node[by_node[p]];
        if (p == 0 || nd != node[by_node[p - 1]]) { fv_node_ids[nn] = (uint32_t)nd; fv_node_ptr[nn] = (int32_t)p; ++nn; }
        fv_features[p] = by_node[p];
    }
    fv_node_ptr[nn] = (int32_t)by_node.size();
    *n_words = nw; *fv_n_nodes = nn;
    return ORBX_OK;
}

int orbx_distinctive_descriptors_device(orbx_matcher* m, const uint8_t* d_desc, const int32_t* d_group_ptr, int n_groups,
                                        int64_t n_rows, int32_t* d_best, void* stream)
{
    if (!m || n_groups < 0 || n_rows < 0 || n_rows > 0x7FFFFFFFLL) { set_error("orbx_distinctive_descriptors_device: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (n_groups == 0) return ORBX_OK;
    if (!d_group_ptr || !d_best || (n_rows > 0 && !d_desc)) { set_error("orbx_distinctive_descriptors_device: null pointer"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(matcher_device(m)));
    hipStream_t st = (hipStream_t)stream;
    void* d_median = nullptr;
    int rc = matcher_stage(m, 7, sizeof(int32_t) * (size_t)std::max<int64_t>(n_rows, 1), &d_median);
    if (rc) return rc;
    if (n_rows > 0)
        hipLaunchKernelGGL(k_distinctive_median, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_desc),
                           d_group_ptr, n_groups, (long long)n_rows, (int32_t*)d_median);
    hipLaunchKernelGGL(k_distinctive_pick, dim3((n_groups + 255) / 256), dim3(256), 0, st, (const int32_t*)d_median, d_group_ptr, n_groups, d_best);
    HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbx_distinctive_descriptors(orbx_matcher* m, const uint8_t* desc, const int32_t* group_ptr, int n_groups, int32_t* best)
{
    if (!m || n_groups < 0) { set_error("orbx_distinctive_descriptors: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (n_groups == 0) return ORBX_OK;
    if (!group_ptr || !best) { set_error("orbx_distinctive_descriptors: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (group_ptr[0] != 0) { set_error("group_ptr must start at 0"); return ORBX_ERR_BAD_ARG; }
    for (int g = 0; g < n_groups; ++g) if (group_ptr[g] > group_ptr[g + 1]) { set_error("group_ptr not monotone"); return ORBX_ERR_BAD_ARG; }
    const int64_t n_rows = group_ptr[n_groups];
    if (n_rows > 0 && !desc) { set_error("orbx_distinctive_descriptors: null pointer"); return ORBX_ERR_BAD_ARG; }
    HIP_CHECK(hipSetDevice(matcher_device(m)));
    hipStream_t st = (hipStream_t)matcher_stream(m);
    void *d_desc = nullptr, *d_ptr = nullptr, *d_best = nullptr;
    int rc;
    if ((rc = matcher_stage(m, 0, (size_t)std::max<int64_t>(n_rows, 1) * 32, &d_desc))) return rc;
    if ((rc = matcher_stage(m, 1, sizeof(int32_t) * ((size_t)n_groups + 1), &d_ptr))) return rc;
    if ((rc = matcher_stage(m, 2, sizeof(int32_t) * (size_t)n_groups, &d_best))) return rc;
    if (n_rows) HIP_CHECK(hipMemcpyAsync(d_desc, desc, (size_t)n_rows * 32, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(d_ptr, group_ptr, sizeof(int32_t) * ((size_t)n_groups + 1), hipMemcpyHostToDevice, st));
    rc = orbx_distinctive_descriptors_device(m, (const uint8_t*)d_desc, (const int32_t*)d_ptr, n_groups, n_rows, (int32_t*)d_best, st);
    if (rc) return rc;
    HIP_CHECK(hipMemcpyAsync(best, d_best, sizeof(int32_t) * (size_t)n_groups, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return ORBX_OK;
}

} // extern "C"
