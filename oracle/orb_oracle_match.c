/*
 * orb_oracle_match.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see orb_oracle.h) for the
 * descriptor-matching hot loops of the reference: ORBmatcher (src/ORBmatcher.cc) and the
 * Frame helpers they use (src/Frame.cc).  Plain arrays stand in for Frame / MapPoint.
 */
#include "orb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1650-1666 (the SWAR popcount, literally) */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* Brute-force best/second over a whole descriptor set: the inner loop shape shared by every
 * Search* routine (strict '<' so the first candidate wins ties), e.g. src/ORBmatcher.cc:439-460. */
void orc_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int64_t nt, orc_best2* out)
{
    for (int i = 0; i < nq; ++i) {
        int best = 256, second = 256, idx = -1;
        for (int64_t j = 0; j < nt; ++j) {
            int d = orc_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < best) { second = best; best = d; idx = (int)j; }
            else if (d < second) second = d;
        }
        out[i].best = best; out[i].best_idx = idx; out[i].second = second; out[i].pad = 0;
    }
}

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1604-1645 */
void orc_compute_three_maxima(const int32_t* sizes, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ---- Frame grid: AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea, src/Frame.cc:230-245, 327-392 ---- */
struct orc_frame_grid {
    const orc_keypoint* kps; int n;
    float mnMinX, mnMaxX, mnMinY, mnMaxY, invW, invH;
    int* cell_start;  /* COLS*ROWS+1, cell id = ix*ROWS+iy */
    int* cell_items;
};

orc_frame_grid* orc_grid_create(const orc_keypoint* kps, int n, float minX, float maxX, float minY, float maxY)
{
    orc_frame_grid* g = (orc_frame_grid*)calloc(1, sizeof(*g));
    g->kps = kps; g->n = n;
    g->mnMinX = minX; g->mnMaxX = maxX; g->mnMinY = minY; g->mnMaxY = maxY;
    g->invW = (float)FRAME_GRID_COLS / (maxX - minX);     /* src/Frame.cc:101-102 */
    g->invH = (float)FRAME_GRID_ROWS / (maxY - minY);
    const int nc = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    g->cell_start = (int*)calloc((size_t)nc + 1, sizeof(int));
    g->cell_items = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* cell_of = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int px = (int)roundf((kps[i].x - g->mnMinX) * g->invW);
        int py = (int)roundf((kps[i].y - g->mnMinY) * g->invH);
        if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) { cell_of[i] = -1; continue; }
        cell_of[i] = px * FRAME_GRID_ROWS + py;
        g->cell_start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) g->cell_start[c + 1] += g->cell_start[c];
    int* fill = (int*)calloc((size_t)nc, sizeof(int));
    for (int i = 0; i < n; ++i) {
        if (cell_of[i] < 0) continue;
        g->cell_items[g->cell_start[cell_of[i]] + fill[cell_of[i]]++] = i;
    }
    free(fill); free(cell_of);
    return g;
}

void orc_grid_destroy(orc_frame_grid* g)
{
    if (!g) return;
    free(g->cell_start); free(g->cell_items); free(g);
}

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

int orc_grid_features_in_area(const orc_frame_grid* g, float x, float y, float r, int minLevel, int maxLevel,
                              int32_t* out, int cap)
{
    int nout = 0;
    const int nMinCellX = imax(0, (int)floorf((x - g->mnMinX - r) * g->invW));
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    const int nMaxCellX = imin(FRAME_GRID_COLS - 1, (int)ceilf((x - g->mnMinX + r) * g->invW));
    if (nMaxCellX < 0) return 0;
    const int nMinCellY = imax(0, (int)floorf((y - g->mnMinY - r) * g->invH));
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    const int nMaxCellY = imin(FRAME_GRID_ROWS - 1, (int)ceilf((y - g->mnMinY + r) * g->invH));
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            int c = ix * FRAME_GRID_ROWS + iy;
            for (int j = g->cell_start[c]; j < g->cell_start[c + 1]; ++j) {
                const orc_keypoint* kp = &g->kps[g->cell_items[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x;
                const float disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (nout < cap) out[nout] = g->cell_items[j];
                    nout++;
                }
            }
        }
    }
    return nout;
}

static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;       /* src/ORBmatcher.cc:416: the inverse of the bin COUNT, so only bins 0..12 occur */
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:408-523 */
int orc_search_for_initialization(const orc_keypoint* k1, const uint8_t* d1, int n1,
                                  const orc_keypoint* k2, const uint8_t* d2, int n2,
                                  float minX, float maxX, float minY, float maxY,
                                  float* prev_matched_xy, int window,
                                  float nnratio, int check_ori, int32_t* matches12)
{
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { rotHist[i] = (int*)malloc(sizeof(int) * (size_t)(n1 + 1)); rotN[i] = 0; }
    int* vMatchedDistance = (int*)malloc(sizeof(int) * (size_t)(n2 + 1));
    int* vnMatches21 = (int*)malloc(sizeof(int) * (size_t)(n2 + 1));
    for (int i = 0; i < n2; ++i) { vMatchedDistance[i] = INT_MAX; vnMatches21[i] = -1; }
    int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    orc_frame_grid* g = orc_grid_create(k2, n2, minX, maxX, minY, maxY);

    for (int i1 = 0; i1 < n1; i1++) {
        int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        int nc = orc_grid_features_in_area(g, prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], (float)window, level1, level1, vIndices2, n2);
        if (nc == 0) continue;
        const uint8_t* da = d1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            int i2 = vIndices2[c];
            int dist = orc_descriptor_distance(da, d2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    int bin = rot_bin(k1[i1].angle, k2[bestIdx2].angle);
                    rotHist[bin][rotN[bin]++] = i1;
                }
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                int idx1 = rotHist[i][j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            prev_matched_xy[2 * i1] = k2[matches12[i1]].x;
            prev_matched_xy[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    orc_grid_destroy(g);
    for (int i = 0; i < HISTO_LENGTH; ++i) free(rotHist[i]);
    free(vMatchedDistance); free(vnMatches21); free(vIndices2);
    return nmatches;
}

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th): candidate loop and acceptance,
 * src/ORBmatcher.cc:68-125.  qr is r*scale[level] (line 69); claimed[idx] >= 0 plays the role of
 * "F.mvpMapPoints[idx] with Observations()>0" (line 87-89) and receives the query index (line 123). */
int orc_search_by_projection_mappoints(const float* qu, const float* qv, const float* qr, const float* qur,
                                       const int32_t* qlevel, const uint8_t* qdesc, int nq,
                                       const orc_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                       float minX, float maxX, float minY, float maxY,
                                       float nnratio, const uint8_t* q_transient, int32_t* claimed)
{
    int nmatches = 0;
    orc_frame_grid* g = orc_grid_create(k2, n2, minX, maxX, minY, maxY);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    for (int i = 0; i < nq; ++i) {
        const int lvl = qlevel[i];
        int nc = orc_grid_features_in_area(g, qu[i], qv[i], qr[i], lvl - 1, lvl, vIndices, n2);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = vIndices[c];
            if (claimed[idx] >= 0) continue;
            if (uright2 && uright2[idx] > 0) {
                const float er = fabsf(qur[i] - uright2[idx]);
                if (er > qr[i]) continue;
            }
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, d2 + (size_t)idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k2[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = k2[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            claimed[bestIdx] = (q_transient && q_transient[i]) ? -2 - i : i;   /* F.mvpMapPoints[bestIdx]=pMP; it blocks later queries only if pMP->Observations()>0 */
            nmatches++;
        }
    }
    free(vIndices);
    orc_grid_destroy(g);
    return nmatches;
}

/* SearchByProjection(Frame&, const Frame&, th, bMono): candidate loop, acceptance and rotation
 * histogram, src/ORBmatcher.cc:1381-1470 (and the KeyFrame variant :1531-1598 with th_dist = ORBdist). */
int orc_search_by_projection_lastframe(const float* qu, const float* qv, const float* qradius, const float* qur,
                                       const int32_t* qoctave, const float* qangle, const uint8_t* qdesc, int nq,
                                       const orc_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                                       float minX, float maxX, float minY, float maxY,
                                       int mode, int th_dist, int check_ori, const uint8_t* q_transient, int32_t* claimed)
{
    int nmatches = 0;
    orc_frame_grid* g = orc_grid_create(k2, n2, minX, maxX, minY, maxY);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { rotHist[i] = (int*)malloc(sizeof(int) * (size_t)(nq + 1)); rotN[i] = 0; }
    for (int i = 0; i < nq; ++i) {
        const int oct = qoctave[i];
        const float radius = qradius[i];
        int nc;
        if (mode == 1) nc = orc_grid_features_in_area(g, qu[i], qv[i], radius, oct, -1, vIndices, n2);
        else if (mode == 2) nc = orc_grid_features_in_area(g, qu[i], qv[i], radius, 0, oct, vIndices, n2);
        else nc = orc_grid_features_in_area(g, qu[i], qv[i], radius, oct - 1, oct + 1, vIndices, n2);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = vIndices[c];
            if (claimed[i2] >= 0) continue;
            if (uright2 && uright2[i2] > 0) {
                const float er = fabsf(qur[i] - uright2[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, d2 + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= th_dist) {
            claimed[bestIdx2] = (q_transient && q_transient[i]) ? -2 - i : i;  /* CurrentFrame.mvpMapPoints[bestIdx2]=pMP; blocks only with observations */
            nmatches++;
            if (check_ori) {
                int bin = rot_bin(qangle[i], k2[bestIdx2].angle);
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (int j = 0; j < rotN[i]; j++) { claimed[rotHist[i][j]] = -1; nmatches--; }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(rotHist[i]);
    free(vIndices);
    orc_grid_destroy(g);
    return nmatches;
}

/* Frame::ComputeStereoMatches, Hamming part, src/Frame.cc:466-549 */
void orc_stereo_match(const orc_keypoint* kl, const uint8_t* dl, int nl,
                      const orc_keypoint* kr, const uint8_t* dr, int nr,
                      const float* scale_factors, int nlevels, int n_rows,
                      float min_d, float max_d, int32_t* best_dist, int32_t* best_idx)
{
    (void)nlevels;
    /* row table :476-493 */
    int* row_count = (int*)calloc((size_t)n_rows + 1, sizeof(int));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const int maxr = (int)ceilf(kpY + r);
        const int minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < n_rows) row_count[yi + 1]++;
    }
    for (int y = 0; y < n_rows; ++y) row_count[y + 1] += row_count[y];
    int* items = (int*)malloc(sizeof(int) * (size_t)(row_count[n_rows] + 1));
    int* fill = (int*)calloc((size_t)n_rows, sizeof(int));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const int maxr = (int)ceilf(kpY + r);
        const int minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < n_rows) items[row_count[yi] + fill[yi]++] = iR;
    }
    for (int iL = 0; iL < nl; iL++) {
        best_dist[iL] = -1; best_idx[iL] = -1;
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        const int row = (int)vL;                 /* vRowIndices[vL]: float index truncates */
        if (row < 0 || row >= n_rows) continue;
        const int c0 = row_count[row], c1 = row_count[row + 1];
        if (c0 == c1) continue;
        const float minU = uL - max_d, maxU = uL - min_d;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; int bestIdxR = 0;
        for (int c = c0; c < c1; ++c) {
            const int iR = items[c];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        best_dist[iL] = bestDist; best_idx[iL] = bestIdxR;
    }
    free(fill); free(items); free(row_count);
}

/* ------------------------------------------------------------------------- */
/* Vocabulary-node matchers                                                    */
/* ------------------------------------------------------------------------- */
/* std::map::lower_bound on the flattened feature vector: first node index whose id is not below `id` */
static int fv_lower_bound(const orc_feature_vector* fv, uint32_t id)
{
    int k = 0;
    while (k < fv->n_nodes && fv->node_ids[k] < id) ++k;
    return k;
}

/* ORBmatcher::SearchByBoW, src/ORBmatcher.cc:159-291 (keyframe_form 0) / :525-658 (keyframe_form 1) */
int orc_search_by_bow(const uint8_t* d1, const float* angle1, const uint8_t* eligible1, int n1, const orc_feature_vector* fv1,
                      const uint8_t* d2, const float* angle2, const uint8_t* eligible2, int n2, const orc_feature_vector* fv2,
                      float nnratio, int check_ori, int keyframe_form, int32_t* match12, int32_t* match21)
{
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    for (int i = 0; i < n2; ++i) match21[i] = -1;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { rotHist[i] = (int*)malloc(sizeof(int) * (size_t)(n1 + 1)); rotN[i] = 0; }
    int it1 = 0, it2 = 0;
    while (it1 < fv1->n_nodes && it2 < fv2->n_nodes) {
        if (fv1->node_ids[it1] == fv2->node_ids[it2]) {
            for (int c1 = fv1->node_ptr[it1]; c1 < fv1->node_ptr[it1 + 1]; ++c1) {
                const int idx1 = fv1->features[c1];
                if (!eligible1[idx1]) continue;                          /* !pMP || pMP->isBad() */
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int c2 = fv2->node_ptr[it2]; c2 < fv2->node_ptr[it2 + 1]; ++c2) {
                    const int idx2 = fv2->features[c2];
                    if (match21[idx2] >= 0) continue;                    /* vpMapPointMatches[realIdxF] / vbMatched2[idx2] */
                    if (eligible2 && !eligible2[idx2]) continue;
                    const int dist = orc_descriptor_distance(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (keyframe_form ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2; match21[bestIdx2] = idx1;
                        if (check_ori) { const int bin = rot_bin(angle1[idx1], angle2[bestIdx2]); rotHist[bin][rotN[bin]++] = idx1; }
                        nmatches++;
                    }
                }
            }
            it1++; it2++;
        } else if (fv1->node_ids[it1] < fv2->node_ids[it2]) it1 = fv_lower_bound(fv1, fv2->node_ids[it2]);
        else it2 = fv_lower_bound(fv2, fv1->node_ids[it1]);
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) {
                const int idx1 = rotHist[i][j];
                match21[match12[idx1]] = -1; match12[idx1] = -1; nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(rotHist[i]);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:140-157 */
static int check_dist_epipolar_line(const orc_keypoint* kp1, const orc_keypoint* kp2, const float* F12, const float* sigma2_2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2_2[kp2->octave];
}

/* ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:660-826 */
int orc_search_for_triangulation(const orc_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, const float* uright1, int n1,
                                 const orc_feature_vector* fv1,
                                 const orc_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, const float* uright2, int n2,
                                 const orc_feature_vector* fv2,
                                 const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                 int only_stereo, int check_ori, int32_t* match12)
{
    (void)n2;
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { rotHist[i] = (int*)malloc(sizeof(int) * (size_t)(n1 + 1)); rotN[i] = 0; }
    int it1 = 0, it2 = 0;
    while (it1 < fv1->n_nodes && it2 < fv2->n_nodes) {
        if (fv1->node_ids[it1] == fv2->node_ids[it2]) {
            for (int c1 = fv1->node_ptr[it1]; c1 < fv1->node_ptr[it1 + 1]; ++c1) {
                const int idx1 = fv1->features[c1];
                if (has_mp1[idx1]) continue;
                const int bStereo1 = uright1[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const orc_keypoint* kp1 = &k1[idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int c2 = fv2->node_ptr[it2]; c2 < fv2->node_ptr[it2 + 1]; ++c2) {
                    const int idx2 = fv2->features[c2];
                    if (has_mp2[idx2]) continue;                          /* vbMatched2 is never set in the reference */
                    const int bStereo2 = uright2[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orc_descriptor_distance(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const orc_keypoint* kp2 = &k2[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2->x, distey = ey - kp2->y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[kp2->octave]) continue;
                    }
                    if (check_dist_epipolar_line(kp1, kp2, F12, level_sigma2_2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2; nmatches++;
                    if (check_ori) { const int bin = rot_bin(kp1->angle, k2[bestIdx2].angle); rotHist[bin][rotN[bin]++] = idx1; }
                }
            }
            it1++; it2++;
        } else if (fv1->node_ids[it1] < fv2->node_ids[it2]) it1 = fv_lower_bound(fv1, fv2->node_ids[it2]);
        else it2 = fv_lower_bound(fv2, fv1->node_ids[it1]);
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_compute_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotN[i]; j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(rotHist[i]);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:362-401 */
int orc_search_by_projection_keyframe(const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                                      const uint8_t* qdesc, int nq, const orc_keypoint* k2, const uint8_t* d2, int n2,
                                      float minX, float maxX, float minY, float maxY, int32_t* claimed)
{
    int nmatches = 0;
    orc_frame_grid* g = orc_grid_create(k2, n2, minX, maxX, minY, maxY);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    for (int i = 0; i < nq; ++i) {
        const int nPredictedLevel = qlevel[i];
        int nc = orc_grid_features_in_area(g, qu[i], qv[i], qradius[i], 0, -1, vIndices, n2);      /* :365, no level filter */
        if (nc == 0) continue;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = vIndices[c];
            if (claimed[idx] >= 0) continue;
            const int kpLevel = k2[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, d2 + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { claimed[bestIdx] = i; nmatches++; }
    }
    free(vIndices);
    orc_grid_destroy(g);
    return nmatches;
}

/* Candidate loop of ORBmatcher::Fuse: src/ORBmatcher.cc:893-952 (reproj = 1: the chi-square tests :917-941, bestDist
 * starts at 256) and :1052-1082 (reproj = 0: bestDist starts at INT_MAX).  best_dist[i] = INT_MAX when no candidate passed. */
void orc_fuse_candidates(const float* qu, const float* qv, const float* qur, const float* qradius, const int32_t* qlevel,
                         const uint8_t* qdesc, int nq, const orc_keypoint* k2, const uint8_t* d2, const float* uright2, int n2,
                         const float* inv_level_sigma2, float minX, float maxX, float minY, float maxY, int reproj,
                         int32_t* best_idx, int32_t* best_dist)
{
    orc_frame_grid* g = orc_grid_create(k2, n2, minX, maxX, minY, maxY);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1; best_dist[i] = INT_MAX;
        const float u = qu[i], v = qv[i];
        const int nPredictedLevel = qlevel[i];
        const int nc = orc_grid_features_in_area(g, u, v, qradius[i], 0, -1, vIndices, n2);        /* KeyFrame::GetFeaturesInArea */
        if (nc == 0) continue;
        int bestDist = reproj ? 256 : INT_MAX, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = vIndices[c];
            const orc_keypoint* kp = &k2[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (reproj) {
                if (uright2[idx] >= 0) {
                    const float ex = u - kp->x, ey = v - kp->y, er = qur[i] - uright2[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp->x, ey = v - kp->y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, d2 + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestIdx >= 0) { best_idx[i] = bestIdx; best_dist[i] = bestDist; }
    }
    free(vIndices);
    orc_grid_destroy(g);
}

/* one direction of SearchBySim3, src/ORBmatcher.cc:1188-1228 */
static void sim3_direction(const uint8_t* valid, const float* qu, const float* qv, const float* qradius, const int32_t* qlevel,
                           const uint8_t* qdesc, int nq, const orc_keypoint* kt, const uint8_t* dt, int nt,
                           float minX, float maxX, float minY, float maxY, int32_t* vnMatch)
{
    orc_frame_grid* g = orc_grid_create(kt, nt, minX, maxX, minY, maxY);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt + 1));
    for (int i = 0; i < nq; ++i) {
        vnMatch[i] = -1;
        if (!valid[i]) continue;
        const int nPredictedLevel = qlevel[i];
        const int nc = orc_grid_features_in_area(g, qu[i], qv[i], qradius[i], 0, -1, vIndices, nt);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = vIndices[c];
            if (kt[idx].octave < nPredictedLevel - 1 || kt[idx].octave > nPredictedLevel) continue;
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, dt + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
    }
    free(vIndices);
    orc_grid_destroy(g);
}

/* ORBmatcher::SearchBySim3 candidate loops and agreement, src/ORBmatcher.cc:1147-1328.  Returns nFound. */
int orc_search_by_sim3(const uint8_t* valid1, const float* u1in2, const float* v1in2, const float* radius1, const int32_t* level1,
                       const uint8_t* mpdesc1, const orc_keypoint* k1, const uint8_t* d1, int n1,
                       const uint8_t* valid2, const float* u2in1, const float* v2in1, const float* radius2, const int32_t* level2,
                       const uint8_t* mpdesc2, const orc_keypoint* k2, const uint8_t* d2, int n2,
                       float minX, float maxX, float minY, float maxY, int32_t* match12)
{
    int32_t* vnMatch1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 + 1));
    int32_t* vnMatch2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    sim3_direction(valid1, u1in2, v1in2, radius1, level1, mpdesc1, n1, k2, d2, n2, minX, maxX, minY, maxY, vnMatch1);
    sim3_direction(valid2, u2in1, v2in1, radius2, level2, mpdesc2, n2, k1, d1, n1, minX, maxX, minY, maxY, vnMatch2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; ++i1) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { match12[i1] = idx2; nFound++; }
        }
    }
    free(vnMatch1); free(vnMatch2);
    return nFound;
}
