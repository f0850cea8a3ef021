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
    FrameGrid grid;
    grid.build(k2, n2, minX, maxX, minY, maxY);
    Csr csr;
    csr.start(n1);
    std::vector<int> v;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) { csr.push_empty(); continue; }
        grid.query(prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], (float)window, level1, level1, v);
        csr.push_row(v);
    }
    int rc = csr.run(m, d1, n1, d2, n2);
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
                                        float minX, float maxX, float minY, float maxY, float nnratio, int32_t* claimed, int* nmatches)
{
    if (!m || !nmatches || nq < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY)) { set_error("orbx_search_by_projection_mappoints: bad argument"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    if (nq == 0 || n2 == 0) return ORBX_OK;
    if (!qu || !qv || !qr || !qlevel || !qdesc || !k2 || !d2 || !claimed || (uright2 && !qur)) { set_error("orbx_search_by_projection_mappoints: null pointer"); return ORBX_ERR_BAD_ARG; }
    if (!octaves_ok(k2, n2)) { set_error("octave out of range"); return ORBX_ERR_BAD_ARG; }
    FrameGrid grid;
    grid.build(k2, n2, minX, maxX, minY, maxY);
    Csr csr;
    csr.start(nq);
    std::vector<int> v;
    for (int i = 0; i < nq; ++i) {
        grid.query(qu[i], qv[i], qr[i], qlevel[i] - 1, qlevel[i], v);
        csr.push_row(v);
    }
    int rc = csr.run(m, qdesc, nq, d2, n2);
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
            claimed[bestIdx] = i;
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
                                        int32_t* claimed, int* nmatches)
{
    if (!m || !nmatches || nq < 0 || n2 < 0 || !(maxX > minX) || !(maxY > minY) || mode < 0 || mode > 2) { set_error("orbx_search_by_projection_lastframe: bad argument"); return ORBX_ERR_BAD_ARG; }
    *nmatches = 0;
    if (nq == 0 || n2 == 0) return ORBX_OK;
    if (!qu || !qv || !qradius || !qoctave || !qdesc || !k2 || !d2 || !claimed || (uright2 && !qur) || (check_ori && !qangle)) { set_error("orbx_search_by_projection_lastframe: null pointer"); return ORBX_ERR_BAD_ARG; }
    FrameGrid grid;
    grid.build(k2, n2, minX, maxX, minY, maxY);
    Csr csr;
    csr.start(nq);
    std::vector<int> v;
    for (int i = 0; i < nq; ++i) {
        const int o = qoctave[i];
        if (mode == 1) grid.query(qu[i], qv[i], qradius[i], o, -1, v);
        else if (mode == 2) grid.query(qu[i], qv[i], qradius[i], 0, o, v);
        else grid.query(qu[i], qv[i], qradius[i], o - 1, o + 1, v);
        csr.push_row(v);
    }
    int rc = csr.run(m, qdesc, nq, d2, n2);
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
            claimed[bestIdx2] = i;
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
