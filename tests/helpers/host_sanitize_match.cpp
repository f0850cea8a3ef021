// The host-side half of the matchers (frame grid, candidate rows, sequential claim / ratio / rotation-histogram passes of
// orbx_match_host.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer.  The two device drivers it calls are replaced
// HERE, in this test program only, by plain loops, so that the index-heavy host code can run on a machine without a GPU;
// liborbx itself has no such path.
#include "../../visual-slam_amd/csrc/orbx_internal.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

struct orbx_matcher { int unused; };

namespace orbx {
int csr_distances(orbx_matcher*, const uint8_t* q, int nq, const uint8_t* t, int nt, const int32_t* rowptr, const int32_t* cand, int32_t* dist_out)
{
    for (int i = 0; i < nq; ++i)
        for (int c = rowptr[i]; c < rowptr[i + 1]; ++c) {
            if (cand[c] < 0 || cand[c] >= nt) return ORBX_ERR_BAD_ARG;
            dist_out[c] = orbx_descriptor_distance(q + (size_t)i * 32, t + (size_t)cand[c] * 32);
        }
    return ORBX_OK;
}
int window_csr(orbx_matcher*, const WindowQuery* q, const uint8_t* qdesc, int nq, const orbx_keypoint* k2, const uint8_t* d2, int n2,
               const int* grid_order, int n_in_grid, std::vector<int32_t>& rowptr, std::vector<int32_t>& cand, std::vector<int32_t>& dist)
{
    rowptr.assign(1, 0); cand.clear(); dist.clear();
    for (int i = 0; i < nq; ++i) {
        if (q[i].r >= 0)
            for (int j = 0; j < n_in_grid; ++j) {
                const orbx_keypoint& k = k2[grid_order[j]];
                const bool check = q[i].minLevel > 0 || q[i].maxLevel >= 0;
                if (check && (k.octave < q[i].minLevel || (q[i].maxLevel >= 0 && k.octave > q[i].maxLevel))) continue;
                if (std::fabs(k.x - q[i].x) < q[i].r && std::fabs(k.y - q[i].y) < q[i].r) {
                    cand.push_back(grid_order[j]);
                    dist.push_back(orbx_descriptor_distance(qdesc + (size_t)i * 32, d2 + (size_t)grid_order[j] * 32));
                }
            }
        rowptr.push_back((int32_t)cand.size());
    }
    (void)n2;
    return ORBX_OK;
}
}

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

int main()
{
    std::mt19937 rng(11);
    auto uni = [&](int a, int b) { return (int)(rng() % (unsigned)(b - a + 1)) + a; };
    auto unf = [&](float a, float b) { return a + (b - a) * (float)(rng() % 100000) / 100000.f; };
    orbx_matcher m{0};
    float scale[8], sigma2[8], inv_sigma2[8];
    for (int l = 0; l < 8; ++l) { scale[l] = std::pow(1.2f, (float)l); sigma2[l] = scale[l] * scale[l]; inv_sigma2[l] = 1.f / sigma2[l]; }
    for (int it = 0; it < 40; ++it) {
        const int n1 = it % 10 == 0 ? uni(0, 3) : uni(50, 900), n2 = it % 10 == 1 ? uni(0, 3) : uni(50, 900);
        const float W = 640.f, H = 480.f;
        std::vector<orbx_keypoint> k1(n1), k2(n2);
        std::vector<uint8_t> d1((size_t)n1 * 32 + 32), d2((size_t)n2 * 32 + 32);
        for (auto& b : d2) b = (uint8_t)rng();
        for (int i = 0; i < n2; ++i) k2[i] = orbx_keypoint{unf(-5, W + 5), unf(-5, H + 5), 31.f, unf(0, 360), 20.f, uni(0, 7), -1};
        for (int i = 0; i < n1; ++i) {
            const int j = n2 ? uni(0, n2 - 1) : 0;
            k1[i] = n2 ? k2[j] : orbx_keypoint{unf(0, W), unf(0, H), 31.f, 0.f, 20.f, uni(0, 7), -1};
            k1[i].x += unf(-6, 6); k1[i].y += unf(-6, 6); k1[i].octave = uni(0, 7);
            if (n2) std::memcpy(&d1[(size_t)i * 32], &d2[(size_t)j * 32], 32);
            for (int b = 0; b < 20; ++b) d1[(size_t)i * 32 + uni(0, 31)] ^= (uint8_t)(1u << uni(0, 7));
        }
        std::vector<float> prev(2 * (size_t)n1 + 2), qu(n1 + 1), qv(n1 + 1), qr(n1 + 1), qur(n1 + 1), ang(n1 + 1), ur2(n2 + 1);
        std::vector<int32_t> lvl(n1 + 1), m12(n1 + 1), m21(n2 + 1), claimed(n2 + 1), bi(n1 + 1), bd(n1 + 1);
        std::vector<uint8_t> valid1(n1 + 1), valid2(n2 + 1), transient(n1 + 1);
        for (int i = 0; i < n1; ++i) {
            prev[2 * i] = k1[i].x; prev[2 * i + 1] = k1[i].y; qu[i] = k1[i].x; qv[i] = k1[i].y; lvl[i] = k1[i].octave;
            qr[i] = unf(2, 40) * scale[lvl[i]]; qur[i] = qu[i] - unf(1, 30); ang[i] = k1[i].angle; valid1[i] = rng() % 5 != 0; transient[i] = rng() % 3 == 0;
        }
        for (int i = 0; i < n2; ++i) { ur2[i] = rng() % 3 ? k2[i].x - unf(1, 30) : -1.f; valid2[i] = rng() % 5 != 0; }
        int nm = -1;
        EXPECT(orbx_search_for_initialization(&m, k1.data(), d1.data(), n1, k2.data(), d2.data(), n2, 0, W, 0, H, prev.data(), uni(5, 120), 0.9f, it & 1,
                                              m12.data(), &nm) == ORBX_OK);
        int cnt = 0;
        for (int i = 0; i < n1; ++i) { EXPECT(m12[i] >= -1 && m12[i] < n2); cnt += m12[i] >= 0; }
        EXPECT(cnt == nm);
        for (int mode = 0; mode < 3; ++mode) {
            for (int i = 0; i < n2; ++i) claimed[i] = rng() % 7 == 0 ? (1 << 20) : -1;
            EXPECT(orbx_search_by_projection_lastframe(&m, qu.data(), qv.data(), qr.data(), qur.data(), lvl.data(), ang.data(), d1.data(), n1, k2.data(),
                                                       d2.data(), it % 3 ? ur2.data() : nullptr, n2, 0, W, 0, H, mode, 100, 1, it % 2 ? transient.data() : nullptr, claimed.data(), &nm) == ORBX_OK);
            for (int i = 0; i < n2; ++i) EXPECT(claimed[i] == -1 || claimed[i] == (1 << 20) || (claimed[i] >= 0 && claimed[i] < n1) || (claimed[i] <= -2 && -2 - claimed[i] < n1));
        }
        for (int i = 0; i < n2; ++i) claimed[i] = -1;
        EXPECT(orbx_search_by_projection_mappoints(&m, qu.data(), qv.data(), qr.data(), qur.data(), lvl.data(), d1.data(), n1, k2.data(), d2.data(),
                                                   ur2.data(), n2, 0, W, 0, H, 0.8f, it % 2 ? nullptr : transient.data(), claimed.data(), &nm) == ORBX_OK);
        for (int i = 0; i < n2; ++i) claimed[i] = rng() % 9 == 0 ? 5 : -1;
        EXPECT(orbx_search_by_projection_keyframe(&m, qu.data(), qv.data(), qr.data(), lvl.data(), d1.data(), n1, k2.data(), d2.data(), n2, 0, W, 0, H,
                                                  claimed.data(), &nm) == ORBX_OK);
        EXPECT(orbx_project_best1(&m, qu.data(), qv.data(), qr.data(), lvl.data(), d1.data(), n1, k2.data(), d2.data(), n2, 0, W, 0, H,
                                  it & 1 ? qur.data() : nullptr, ur2.data(), inv_sigma2, 8, bi.data(), bd.data()) == ORBX_OK);
        for (int i = 0; i < n1; ++i) EXPECT(bi[i] >= -1 && bi[i] < n2);
        // both directions of SearchBySim3
        std::vector<float> u2(n2 + 1), v2(n2 + 1), r2(n2 + 1); std::vector<int32_t> l2(n2 + 1);
        for (int i = 0; i < n2; ++i) { u2[i] = k2[i].x; v2[i] = k2[i].y; l2[i] = k2[i].octave; r2[i] = 7.5f * scale[l2[i]]; }
        EXPECT(orbx_search_by_sim3(&m, valid1.data(), qu.data(), qv.data(), qr.data(), lvl.data(), d1.data(), k1.data(), d1.data(), n1, valid2.data(),
                                   u2.data(), v2.data(), r2.data(), l2.data(), d2.data(), k2.data(), d2.data(), n2, 0, W, 0, H, m12.data(), &nm) == ORBX_OK);
        // vocabulary-node matchers: features filed under a node read off their descriptor
        auto fvof = [&](const std::vector<uint8_t>& d, int n, std::vector<uint32_t>& ids, std::vector<int32_t>& ptr, std::vector<int32_t>& feat) {
            ids.clear(); ptr.assign(1, 0); feat.clear();
            for (uint32_t node = 0; node < 32; ++node) {
                const size_t before = feat.size();
                for (int i = 0; i < n; ++i) if ((uint32_t)(d[(size_t)i * 32] >> 3) == node) feat.push_back(i);
                if (feat.size() != before) { ids.push_back(node * 3 + 1); ptr.push_back((int32_t)feat.size()); }
            }
        };
        std::vector<uint32_t> id1, id2; std::vector<int32_t> p1, p2, f1, f2;
        fvof(d1, n1, id1, p1, f1); fvof(d2, n2, id2, p2, f2);
        orbx_feature_vector fv1{id1.data(), p1.data(), f1.data(), (int32_t)id1.size()}, fv2{id2.data(), p2.data(), f2.data(), (int32_t)id2.size()};
        std::vector<float> a1(n1 + 1), a2(n2 + 1), ur1(n1 + 1);
        for (int i = 0; i < n1; ++i) { a1[i] = k1[i].angle; ur1[i] = rng() % 3 ? k1[i].x - 5.f : -1.f; }
        for (int i = 0; i < n2; ++i) a2[i] = k2[i].angle;
        for (int form = 0; form < 2; ++form) {
            EXPECT(orbx_search_by_bow(&m, d1.data(), a1.data(), valid1.data(), n1, &fv1, d2.data(), a2.data(), form ? valid2.data() : nullptr, n2, &fv2,
                                      0.7f, 1, form, m12.data(), m21.data(), &nm) == ORBX_OK);
            for (int i = 0; i < n1; ++i) EXPECT(m12[i] >= -1 && m12[i] < n2);
            for (int i = 0; i < n2; ++i) EXPECT(m21[i] >= -1 && m21[i] < n1);
        }
        const float F12[9] = {0.f, -1e-4f, 0.02f, 1e-4f, 0.f, -0.03f, -0.02f, 0.03f, 0.f};
        EXPECT(orbx_search_for_triangulation(&m, k1.data(), d1.data(), valid1.data(), ur1.data(), n1, &fv1, k2.data(), d2.data(), valid2.data(), ur2.data(),
                                             n2, &fv2, F12, 320.f, 240.f, scale, sigma2, 8, it & 1, 1, m12.data(), &nm) == ORBX_OK);
        std::vector<int32_t> sd(n1 + 1), si(n1 + 1);
    }
    std::printf("host_sanitize_match: %d failed expectations\n", fails);
    return fails ? 1 : 0;
}
