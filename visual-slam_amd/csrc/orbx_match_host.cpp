// Windowed matchers of ORBmatcher on plain arrays.  The candidate lists come from the 64x48 frame
// grid (host, order-defining), every Hamming distance is computed on the device in one CSR launch,
// and the claim / override pass that the reference runs in query order stays a sequential host loop
// over the precomputed distances: it is what makes the result depend on query order
// (src/ORBmatcher.cc:447, 466-473, 87-89, 1406-1408).
#include "orbx_internal.h"
#include <cmath>
#include <climits>
#include <vector>

using namespace orbx;

namespace {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30; // src/ORBmatcher.cc:37-39

// rotation histogram bin (src/ORBmatcher.cc:478-483); factor = 1/HISTO_LENGTH, so bins 0..12 only
inline int rotation_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

struct Csr {
    std::vector<int32_t> rowptr, cand, dist;
    std::vector<int> scratch;
    void start(int nq) { rowptr.assign(1, 0); rowptr.reserve(nq + 1); cand.clear(); }
    void push_row(const std::vector<int>& v) { cand.insert(cand.end(), v.begin(), v.end()); rowptr.push_back((int32_t)cand.size()); }
    void push_empty() { rowptr.push_back((int32_t)cand.size()); }
    int run(orbx_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt)
    {
        dist.assign(cand.size(), 0);
        return csr_distances(m, q, nq, t, nt, rowptr.data(), cand.data(), dist.data());
    }
    // windowed searches: the frame grid is built here (its cell_items is the order the reference's walk visits keypoints
    // in), the windows are evaluated and the distances computed on the device
    std::vector<WindowQuery> wq;
    void window(float x, float y, float r, int minLevel, int maxLevel) { wq.push_back(WindowQuery{x, y, r, (int16_t)minLevel, (int16_t)maxLevel}); }
    void window_none() { wq.push_back(WindowQuery{0.f, 0.f, -1.f, 0, 0}); }
    int run_windows(orbx_matcher* m, const uint8_t* qdesc, const orbx_keypoint* k2, const uint8_t* d2, int n2,
                    float minX, float maxX, float minY, float maxY)
    {
        FrameGrid grid;
        grid.build(k2, n2, minX, maxX, minY, maxY);
        return window_csr(m, wq.data(), qdesc, (int)wq.size(), k2, d2, n2, grid.cell_items.data(), grid.cell_start.back(), rowptr, cand, dist);
    }
};

bool octaves_ok(const orbx_keypoint* k, int n)
{
    for (int i = 0; i < n; ++i) if (k[i].octave < 0 || k[i].octave >= kMaxLevels) return false;
    return true;
}

} // namespace

extern "C" {

// ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:408-523
int orbx_search_for_initialization(orbx_matcher* m, const orbx_keypoint* k1, const uint8_t* d1, int n1,
                                   const orbx_keypoint* k2, const uint8_t* d2, int n2,
                                   float minX, float maxX, float minY, float maxY, float* prev_matched_xy, int window,
                                   float nnratio, int check_ori, int32_t* matches12, int* nmatches)
{
    if (!m || !nmatches || n1 < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_search_for_initialization: bad argument"); return ORBX_ERR_BAD_ARG; }
    if ((n1 > 0 && (!k1 || !d1 || !prev_matched_xy || !matches12)) || (n2 > 0 && (!k2 || !d2))) { set_error("orbx_search_for_initialization: null pointer"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    Csr csr;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) { csr.window_none(); continue; }
        csr.window(prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], (float)window, level1, level1);
    }
    int rc = csr.run_windows(m, d1, k2, d2, n2, minX, maxX, minY, maxY);
    if (rc) return rc;

    int count = 0;
    std::vector<int> matchedDistance(n2, INT_MAX), matches21(n2, -1);
    std::vector<int> hist[HISTO_LENGTH];
    for (int i1 = 0; i1 < n1; ++i1) {
        const int b = csr.rowptr[i1], e = csr.rowptr[i1 + 1];
        if (b == e) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = b; c < e; ++c) {
            const int i2 = csr.cand[c], dist = csr.dist[c];
            if (matchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; --count; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDistance[bestIdx2] = bestDist;
            ++count;
            if (check_ori) hist[rotation_bin(k1[i1].angle, k2[bestIdx2].angle)].push_back(i1);
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int32_t)hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbx_compute_three_maxima(sizes, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : hist[i]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; --count; }
        }
    }
    for (int i1 = 0; i1 < n1; ++i1)
        if (matches12[i1] >= 0) {
            prev_matched_xy[2 * i1] = k2[matches12[i1]].x;
            prev_matched_xy[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    *nmatches = count;
    return ORBX_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:68-125
int orbx_search_by_projection_mappoints(orbx_matcher* m, const float* qu, const float* qv, const float* qr, const float* qur,
                                        const int32_t* qlevel, const uint8_t* qdesc, int nq,
                                        const orbx_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                        float minX, float maxX, float minY, float maxY, float nnratio, const uint8_t* q_transient, int32_t* claimed, int* nmatches)
{
    if (!m || !nmatches || nq < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_search_by_projection_mappoints: bad argument"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    if (nq == 0 || n2 == 0) return ORBX_OK;
    if (!qu || !qv || !qr || !qlevel || !qdesc || !k2 || !d2 || !claimed || (uright2 && !qur)) { set_error("orbx_search_by_projection_mappoints: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!octaves_ok(k2, n2)) { set_error("octave out of range"); return ORBX_ERR_BAD_ARG; }
    Csr csr;
    for (int i = 0; i < nq; ++i) csr.window(qu[i], qv[i], qr[i], qlevel[i] - 1, qlevel[i]);
    int rc = csr.run_windows(m, qdesc, k2, d2, n2, minX, maxX, minY, maxY);
    if (rc) return rc;
    int count = 0;
    for (int i = 0; i < nq; ++i) {
        const int b = csr.rowptr[i], e = csr.rowptr[i + 1];
        if (b == e) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = b; c < e; ++c) {
            const int idx = csr.cand[c];
            if (claimed[idx] >= 0) continue;
            if (uright2 && uright2[idx] > 0) {
                const float er = std::fabs(qur[i] - uright2[idx]);
                if (er > qr[i]) continue;
            }
            const int dist = csr.dist[c];
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k2[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = k2[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            claimed[bestIdx] = q_transient && q_transient[i] ? -2 - i : i;     // a point without observations does not block (:87-89)
            ++count;
        }
    }
    *nmatches = count;
    return ORBX_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), src/ORBmatcher.cc:1381-1470
int orbx_search_by_projection_lastframe(orbx_matcher* m, const float* qu, const float* qv, const float* qradius, const float* qur,
                                        const int32_t* qoctave, const float* qangle, const uint8_t* qdesc, int nq,
                                        const orbx_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                        float minX, float maxX, float minY, float maxY, int mode, int th_dist, int check_ori,
                                        const uint8_t* q_transient, int32_t* claimed, int* nmatches)
{
    if (!m || !nmatches || nq < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY) || mode < 0 || mode > 2) { set_error("orbx_search_by_projection_lastframe: bad argument"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    if (nq == 0 || n2 == 0) return ORBX_OK;
    if (!qu || !qv || !qradius || !qoctave || !qdesc || !k2 || !d2 || !claimed || (uright2 && !qur) || (check_ori && !qangle)) { set_error("orbx_search_by_projection_lastframe: null pointer"); return ORBX_ERR_BAD_ARG; }
    Csr csr;
    for (int i = 0; i < nq; ++i) {
        const int o = qoctave[i];
        if (mode == 1) csr.window(qu[i], qv[i], qradius[i], o, -1);
        else if (mode == 2) csr.window(qu[i], qv[i], qradius[i], 0, o);
        else csr.window(qu[i], qv[i], qradius[i], o - 1, o + 1);
    }
    int rc = csr.run_windows(m, qdesc, k2, d2, n2, minX, maxX, minY, maxY);
    if (rc) return rc;
    int count = 0;
    std::vector<int> hist[HISTO_LENGTH];
    for (int i = 0; i < nq; ++i) {
        const int b = csr.rowptr[i], e = csr.rowptr[i + 1];
        if (b == e) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = b; c < e; ++c) {
            const int i2 = csr.cand[c];
            if (claimed[i2] >= 0) continue;
            if (uright2 && uright2[i2] > 0) {
                const float er = std::fabs(qur[i] - uright2[i2]);
                if (er > qradius[i]) continue;
            }
            const int dist = csr.dist[c];
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= th_dist) {
            claimed[bestIdx2] = q_transient && q_transient[i] ? -2 - i : i;   // a point without observations does not block (:1406-1408)
            ++count;
            if (check_ori) hist[rotation_bin(qangle[i], k2[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int32_t)hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbx_compute_three_maxima(sizes, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int i2 : hist[i]) { claimed[i2] = -1; --count; }
        }
    }
    *nmatches = count;
    return ORBX_OK;
}

} // extern "C"

namespace {

// Walk of two feature vectors in step (src/ORBmatcher.cc:183-267): calls f(k1, k2) for every node id both hold.
// (lower_bound on the map is a jump to the first node id not below the other side's; a linear merge visits the same pairs.)
template <class F>
void for_shared_nodes(const orbx_feature_vector& a, const orbx_feature_vector& b, F&& f)
{
    int i = 0, j = 0;
    while (i < a.n_nodes && j < b.n_nodes) {
        if (a.node_ids[i] == b.node_ids[j]) { f(i, j); ++i; ++j; }
        else if (a.node_ids[i] < b.node_ids[j]) ++i;
        else ++j;
    }
}

bool feature_vector_ok(const orbx_feature_vector* fv, int n)
{
    if (!fv || fv->n_nodes < 0) return false;
    if (fv->n_nodes == 0) return true;
    if (!fv->node_ids || !fv->node_ptr || fv->node_ptr[0] != 0) return false;
    for (int k = 0; k < fv->n_nodes; ++k) {
        if (fv->node_ptr[k + 1] < fv->node_ptr[k]) return false;
        if (k > 0 && fv->node_ids[k] <= fv->node_ids[k - 1]) return false;
    }
    if (fv->node_ptr[fv->n_nodes] > 0 && !fv->features) return false;
    for (int c = 0; c < fv->node_ptr[fv->n_nodes]; ++c) if (fv->features[c] < 0 || fv->features[c] >= n) return false;
    return true;
}

// One CSR row per side-1 feature that takes part, in processing order; the row's candidates are the side-2 list of the node.
struct BowRows {
    Csr csr;
    std::vector<int> idx1;            // side-1 feature of each row
    std::vector<uint8_t> q;           // its descriptor, gathered
    template <class Take>
    void build(const orbx_feature_vector& fv1, const orbx_feature_vector& fv2, const uint8_t* d1, Take&& take)
    {
        csr.start(0);
        std::vector<int> v;
        for_shared_nodes(fv1, fv2, [&](int a, int b) {
            v.assign(fv2.features + fv2.node_ptr[b], fv2.features + fv2.node_ptr[b + 1]);
            for (int c = fv1.node_ptr[a]; c < fv1.node_ptr[a + 1]; ++c) {
                const int i1 = fv1.features[c];
                if (!take(i1)) continue;
                idx1.push_back(i1);
                q.insert(q.end(), d1 + (size_t)i1 * 32, d1 + (size_t)i1 * 32 + 32);
                csr.push_row(v);
            }
        });
    }
};

} // namespace

// Shared candidate loop of ORBmatcher::Fuse (src/ORBmatcher.cc:893-952, :1052-1082) and SearchBySim3 (:1192-1222, :1272-1302)
static int project_best1(orbx_matcher* m, const float* qu, const float* qv, const float* qradius, const int32_t* qlevel, const uint8_t* qdesc,
                         const uint8_t* qvalid, int nq, const orbx_keypoint* k2, const uint8_t* d2, int n2,
                         float minX, float maxX, float minY, float maxY,
                         const float* qur, const float* uright2, const float* inv_level_sigma2,
                         int32_t* best_idx, int32_t* best_dist)
{
    Csr csr;
    for (int i = 0; i < nq; ++i) {
        if (!qvalid || qvalid[i]) csr.window(qu[i], qv[i], qradius[i], qlevel[i] - 1, qlevel[i]);     // GetFeaturesInArea + the level test
        else csr.window_none();
    }
    int rc = csr.run_windows(m, qdesc, k2, d2, n2, minX, maxX, minY, maxY);
    if (rc) return rc;
    for (int i = 0; i < nq; ++i) {
        int bestDist = 0x7FFFFFFF, bestIdx = -1;
        for (int c = csr.rowptr[i]; c < csr.rowptr[i + 1]; ++c) {
            const int idx = csr.cand[c];
            if (qur) {                                           // reprojection test of Fuse(pKF, vpMapPoints, th), :917-941
                const orbx_keypoint& kp = k2[idx];
                const float ex = qu[i] - kp.x, ey = qv[i] - kp.y;
                if (uright2[idx] >= 0) {
                    const float er = qur[i] - uright2[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kp.octave] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kp.octave] > 5.99) continue;
                }
            }
            if (csr.dist[c] < bestDist) { bestDist = csr.dist[c]; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
    return ORBX_OK;
}

extern "C" {

int orbx_project_best1(orbx_matcher* m, const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                       const uint8_t* qdesc, int nq, const orbx_keypoint* k2, const uint8_t* d2, int n2,
                       float minX, float maxX, float minY, float maxY,
                       const float* qur, const float* uright2, const float* inv_level_sigma2, int nlevels,
                       int32_t* best_idx, int32_t* best_dist)
{
    if (!m || nq < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_project_best1: bad argument"); return ORBX_ERR_BAD_ARG; }
    if (nq == 0) return ORBX_OK;
    if (!best_idx || !best_dist) { set_error("orbx_project_best1: null pointer"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < nq; ++i) { best_idx[i] = -1; best_dist[i] = 0x7FFFFFFF; }
    if (n2 == 0) return ORBX_OK;
    if (!qu || !qv || !qradius || !qlevel || !qdesc || !k2 || !d2) { set_error("orbx_project_best1: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (qur && (!uright2 || !inv_level_sigma2 || nlevels < 1 || nlevels > kMaxLevels)) { set_error("orbx_project_best1: the reprojection test needs uright2 and inv_level_sigma2"); return ORBX_ERR_BAD_ARG; }
    if (!octaves_ok(k2, n2)) { set_error("octave out of range"); return ORBX_ERR_BAD_ARG; }
    if (qur) for (int i = 0; i < n2; ++i) if (k2[i].octave >= nlevels) { set_error("octave outside inv_level_sigma2"); return ORBX_ERR_BAD_ARG; }
    return project_best1(m, qu, qv, qradius, qlevel, qdesc, nullptr, nq, k2, d2, n2, minX, maxX, minY, maxY, qur, uright2, inv_level_sigma2,
                         best_idx, best_dist);
}

int orbx_search_by_sim3(orbx_matcher* m,
                        const uint8_t* valid1, const float* u1in2, const float* v1in2, const float* radius1, const int32_t* level1,
                        const uint8_t* mpdesc1, const orbx_keypoint* k1, const uint8_t* d1, int n1,
                        const uint8_t* valid2, const float* u2in1, const float* v2in1, const float* radius2, const int32_t* level2,
                        const uint8_t* mpdesc2, const orbx_keypoint* k2, const uint8_t* d2, int n2,
                        float minX, float maxX, float minY, float maxY, int32_t* match12, int* nfound)
{
    if (!m || !nfound || n1 < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_search_by_sim3: bad argument"); return ORBX_ERR_BAD_ARG; }
    *nfound = 0;
    if (n1 > 0 && !match12) { set_error("orbx_search_by_sim3: null pointer"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    if (!valid1 || !u1in2 || !v1in2 || !radius1 || !level1 || !mpdesc1 || !k1 || !d1 || !valid2 || !u2in1 || !v2in1 || !radius2 || !level2 ||
        !mpdesc2 || !k2 || !d2) { set_error("orbx_search_by_sim3: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!octaves_ok(k1, n1) || !octaves_ok(k2, n2)) { set_error("octave out of range"); return ORBX_ERR_BAD_ARG; }
    std::vector<int32_t> b1(n1), dd1(n1), b2(n2), dd2(n2);
    int rc = project_best1(m, u1in2, v1in2, radius1, level1, mpdesc1, valid1, n1, k2, d2, n2, minX, maxX, minY, maxY, nullptr, nullptr, nullptr,
                           b1.data(), dd1.data());
    if (rc) return rc;
    rc = project_best1(m, u2in1, v2in1, radius2, level2, mpdesc2, valid2, n2, k1, d1, n1, minX, maxX, minY, maxY, nullptr, nullptr, nullptr,
                       b2.data(), dd2.data());
    if (rc) return rc;
    int found = 0;
    for (int i1 = 0; i1 < n1; ++i1) {                            // :1224-1227, :1304-1307, agreement :1313-1326
        const int idx2 = dd1[i1] <= TH_HIGH ? b1[i1] : -1;
        if (idx2 < 0) continue;
        const int idx1 = dd2[idx2] <= TH_HIGH ? b2[idx2] : -1;
        if (idx1 == i1) { match12[i1] = idx2; ++found; }
    }
    *nfound = found;
    return ORBX_OK;
}

} // extern "C"

extern "C" {

// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:362-401
int orbx_search_by_projection_keyframe(orbx_matcher* m, const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                                       const uint8_t* qdesc, int nq, const orbx_keypoint* k2, const uint8_t* d2, int n2,
                                       float minX, float maxX, float minY, float maxY, int32_t* claimed, int* nmatches)
{
    if (!m || !nmatches || nq < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_search_by_projection_keyframe: bad argument"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    if (nq == 0 || n2 == 0) return ORBX_OK;
    if (!qu || !qv || !qradius || !qlevel || !qdesc || !k2 || !d2 || !claimed) { set_error("orbx_search_by_projection_keyframe: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!octaves_ok(k2, n2)) { set_error("octave out of range"); return ORBX_ERR_BAD_ARG; }
    Csr csr;
    // GetFeaturesInArea(u, v, radius) and the level test of the loop (:365, :381-384) visit the same features in the
    // same order as the level-filtered window
    for (int i = 0; i < nq; ++i) csr.window(qu[i], qv[i], qradius[i], qlevel[i] - 1, qlevel[i]);
    int rc = csr.run_windows(m, qdesc, k2, d2, n2, minX, maxX, minY, maxY);
    if (rc) return rc;
    int count = 0;
    for (int i = 0; i < nq; ++i) {
        int bestDist = 256, bestIdx = -1;
        for (int c = csr.rowptr[i]; c < csr.rowptr[i + 1]; ++c) {
            const int idx = csr.cand[c];
            if (claimed[idx] >= 0) continue;
            if (csr.dist[c] < bestDist) { bestDist = csr.dist[c]; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { claimed[bestIdx] = i; ++count; }
    }
    *nmatches = count;
    return ORBX_OK;
}

// ORBmatcher::SearchByBoW, src/ORBmatcher.cc:159-291 (keyframe_form 0) and :525-658 (keyframe_form 1)
int orbx_search_by_bow(orbx_matcher* m,
                       const uint8_t* d1, const float* angle1, const uint8_t* eligible1, int n1, const orbx_feature_vector* fv1,
                       const uint8_t* d2, const float* angle2, const uint8_t* eligible2, int n2, const orbx_feature_vector* fv2,
                       float nnratio, int check_ori, int keyframe_form, int32_t* match12, int32_t* match21, int* nmatches)
{
    if (!m || !nmatches || n1 < 0 || n2 < 0) { set_error("orbx_search_by_bow: bad argument"); return ORBX_ERR_BAD_ARG; }
    if ((n1 > 0 && (!d1 || !angle1 || !eligible1 || !match12)) || (n2 > 0 && (!d2 || !angle2 || !match21))) { set_error("orbx_search_by_bow: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!feature_vector_ok(fv1, n1) || !feature_vector_ok(fv2, n2)) { set_error("orbx_search_by_bow: malformed feature vector"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    for (int i = 0; i < n2; ++i) match21[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    BowRows rows;
    rows.build(*fv1, *fv2, d1, [&](int i1) { return eligible1[i1] != 0; });
    const int nr = (int)rows.idx1.size();
    if (nr == 0) return ORBX_OK;
    int rc = rows.csr.run(m, rows.q.data(), nr, d2, n2);
    if (rc) return rc;

    int count = 0;
    std::vector<std::pair<int, int>> hist[HISTO_LENGTH];
    for (int r = 0; r < nr; ++r) {
        const int i1 = rows.idx1[r];
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int c = rows.csr.rowptr[r]; c < rows.csr.rowptr[r + 1]; ++c) {
            const int i2 = rows.csr.cand[c];
            if (match21[i2] >= 0 || (eligible2 && !eligible2[i2])) continue;     // :212 / :579
            const int dist = rows.csr.dist[c];
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        const bool low = keyframe_form ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW;  // :601 / :231
        if (low && (float)bestDist1 < nnratio * (float)bestDist2) {
            match12[i1] = bestIdx2; match21[bestIdx2] = i1; ++count;
            if (check_ori) hist[rotation_bin(angle1[i1], angle2[bestIdx2])].push_back({i1, bestIdx2});
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int32_t)hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbx_compute_three_maxima(sizes, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (const auto& pr : hist[i]) { match12[pr.first] = -1; match21[pr.second] = -1; --count; }
        }
    }
    *nmatches = count;
    return ORBX_OK;
}

// ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:660-826
int orbx_search_for_triangulation(orbx_matcher* m,
                                  const orbx_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, const float* uright1, int n1,
                                  const orbx_feature_vector* fv1,
                                  const orbx_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, const float* uright2, int n2,
                                  const orbx_feature_vector* fv2,
                                  const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                  int nlevels, int only_stereo, int check_ori, int32_t* match12, int* nmatches)
{
    if (!m || !nmatches || n1 < 0 || n2 < 0 || !F12 || !scale_factors2 || !level_sigma2_2 || nlevels < 1 || nlevels > kMaxLevels) {
        set_error("orbx_search_for_triangulation: bad argument"); return ORBX_ERR_BAD_ARG;
    }
    if ((n1 > 0 && (!k1 || !d1 || !has_mp1 || !uright1 || !match12)) || (n2 > 0 && (!k2 || !d2 || !has_mp2 || !uright2))) {
        set_error("orbx_search_for_triangulation: null pointer"); return ORBX_ERR_BAD_ARG;
    }
    if (!feature_vector_ok(fv1, n1) || !feature_vector_ok(fv2, n2)) { set_error("orbx_search_for_triangulation: malformed feature vector"); return ORBX_ERR_BAD_ARG; }
    for (int i = 0; i < n2; ++i) if (k2[i].octave < 0 || k2[i].octave >= nlevels) { set_error("orbx_search_for_triangulation: octave out of range"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    BowRows rows;
    rows.build(*fv1, *fv2, d1, [&](int i1) { return !has_mp1[i1] && (!only_stereo || uright1[i1] >= 0); });   // :702-712
    const int nr = (int)rows.idx1.size();
    if (nr == 0) return ORBX_OK;
    int rc = rows.csr.run(m, rows.q.data(), nr, d2, n2);
    if (rc) return rc;

    int count = 0;
    std::vector<int> hist[HISTO_LENGTH];
    for (int r = 0; r < nr; ++r) {
        const int i1 = rows.idx1[r];
        const bool bStereo1 = uright1[i1] >= 0;
        const orbx_keypoint& kp1 = k1[i1];
        // epipolar line of kp1 in image 2 (CheckDistEpipolarLine, :143-145)
        const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
        const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
        const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
        const float den = a * a + b * b;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int cc = rows.csr.rowptr[r]; cc < rows.csr.rowptr[r + 1]; ++cc) {
            const int i2 = rows.csr.cand[cc];
            if (has_mp2[i2]) continue;                                           // :728 (vbMatched2 is never set)
            const bool bStereo2 = uright2[i2] >= 0;
            if (only_stereo && !bStereo2) continue;
            const int dist = rows.csr.dist[cc];
            if (dist > TH_LOW || dist > bestDist) continue;
            const orbx_keypoint& kp2 = k2[i2];
            if (!bStereo1 && !bStereo2) {
                const float distex = ex - kp2.x, distey = ey - kp2.y;
                if (distex * distex + distey * distey < 100 * scale_factors2[kp2.octave]) continue;
            }
            const float num = a * kp2.x + b * kp2.y + c;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (dsqr < 3.84 * level_sigma2_2[kp2.octave]) { bestIdx2 = i2; bestDist = dist; }   // double comparison, as written
        }
        if (bestIdx2 >= 0) {
            match12[i1] = bestIdx2; ++count;
            if (check_ori) hist[rotation_bin(kp1.angle, k2[bestIdx2].angle)].push_back(i1);
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int32_t)hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orbx_compute_three_maxima(sizes, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int i1 : hist[i]) { match12[i1] = -1; --count; }
        }
    }
    *nmatches = count;
    return ORBX_OK;
}

} // extern "C"
