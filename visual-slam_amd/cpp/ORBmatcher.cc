#include "ORBmatcher.h"
#ifdef ORBX_MATCHER_CLASS
#define ORBmatcher ORBX_MATCHER_CLASS      // see ORBmatcher.h
#endif
#include <algorithm>
#include <stdexcept>
#include <string>
#include <cstring>

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static void check(int status, const char* what)
{
    if (status != ORBX_OK)
        throw std::runtime_error(std::string(what) + ": " + orbx_status_string(status) + " (" + orbx_last_error() + ")");
}

static const orbx_keypoint* as_orbx(const std::vector<cv::KeyPoint>& v)
{
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "keypoint layouts must agree");
    return reinterpret_cast<const orbx_keypoint*>(v.data());
}

static std::vector<unsigned char> packed_rows(const cv::Mat& d)
{
    std::vector<unsigned char> out((size_t)d.rows * 32);
    for (int r = 0; r < d.rows; ++r) std::memcpy(out.data() + (size_t)r * 32, d.ptr(r), 32);
    return out;
}

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
ORBmatcher::~ORBmatcher() { orbx_matcher_destroy(mpContext); }

void ORBmatcher::SetDevice(int device)
{
    if (device != mnDevice) { orbx_matcher_destroy(mpContext); mpContext = nullptr; mnDevice = device; }
}

orbx_matcher* ORBmatcher::Context()
{
    if (!mpContext) check(orbx_matcher_create(mnDevice, &mpContext), "ORBmatcher: context");
    return mpContext;
}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    return orbx_descriptor_distance(a.ptr(), b.ptr());
}

int ORBmatcher::DistinctiveDescriptor(const std::vector<cv::Mat>& vDescriptors)
{
    if (vDescriptors.empty()) return -1;
    std::vector<unsigned char> rows(vDescriptors.size() * 32);
    for (size_t i = 0; i < vDescriptors.size(); ++i) std::memcpy(&rows[i * 32], vDescriptors[i].ptr(), 32);
    const int32_t ptr[2] = {0, (int32_t)vDescriptors.size()};
    int32_t best = -1;
    check(orbx_distinctive_descriptors(Context(), rows.data(), ptr, 1, &best), "DistinctiveDescriptor");
    return best;
}

void ORBmatcher::DistinctiveDescriptors(const cv::Mat& descriptorRows, const std::vector<int>& vGroupPtr, std::vector<int>& vBestIdx)
{
    vBestIdx.assign(vGroupPtr.empty() ? 0 : vGroupPtr.size() - 1, -1);
    if (vBestIdx.empty()) return;
    const std::vector<unsigned char> rows = packed_rows(descriptorRows);
    check(orbx_distinctive_descriptors(Context(), rows.data(), vGroupPtr.data(), (int)vBestIdx.size(), vBestIdx.data()), "DistinctiveDescriptors");
}

void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3)
{
    std::vector<int32_t> sizes(L);
    for (int i = 0; i < L; ++i) sizes[i] = (int32_t)histo[i].size();
    orbx_compute_three_maxima(sizes.data(), L, &ind1, &ind2, &ind3);
}

int ORBmatcher::SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                        std::vector<int>& vnMatches12, int windowSize)
{
    const int n1 = (int)F1.keysUn->size(), n2 = (int)F2.keysUn->size();
    vnMatches12.assign(n1, -1);
    if ((int)vbPrevMatched.size() != n1) throw std::runtime_error("SearchForInitialization: vbPrevMatched size");
    std::vector<unsigned char> d1 = packed_rows(*F1.descriptors), d2 = packed_rows(*F2.descriptors);
    int nmatches = 0;
    static_assert(sizeof(cv::Point2f) == 8, "Point2f layout");
    check(orbx_search_for_initialization(Context(), as_orbx(*F1.keysUn), d1.data(), n1, as_orbx(*F2.keysUn), d2.data(), n2,
                                         F2.minX, F2.maxX, F2.minY, F2.maxY, reinterpret_cast<float*>(vbPrevMatched.data()),
                                         windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &nmatches),
          "SearchForInitialization");
    return nmatches;
}

namespace {
struct QueryArrays {
    std::vector<float> u, v, r, ur, angle; std::vector<int32_t> level; std::vector<unsigned char> desc, transient;
    explicit QueryArrays(const std::vector<ProjectedQuery>& q)
    {
        const size_t n = q.size();
        u.resize(n); v.resize(n); r.resize(n); ur.resize(n); angle.resize(n); level.resize(n); desc.resize(n * 32); transient.resize(n);
        for (size_t i = 0; i < n; ++i) {
            u[i] = q[i].u; v[i] = q[i].v; r[i] = q[i].radius; ur[i] = q[i].uRight; angle[i] = q[i].angle; level[i] = q[i].level;
            std::memcpy(desc.data() + i * 32, q[i].descriptor, 32);
            transient[i] = q[i].hasObservations ? 0 : 1;
        }
    }
};
}

int ORBmatcher::SearchByProjection(const FrameView& F, const std::vector<ProjectedQuery>& vQueries, std::vector<int>& vnClaimed)
{
    const int n2 = (int)F.keysUn->size();
    if ((int)vnClaimed.size() != n2) throw std::runtime_error("SearchByProjection: vnClaimed size");
    QueryArrays q(vQueries);
    std::vector<unsigned char> d2 = packed_rows(*F.descriptors);
    int nmatches = 0;
    check(orbx_search_by_projection_mappoints(Context(), q.u.data(), q.v.data(), q.r.data(), q.ur.data(), q.level.data(), q.desc.data(),
                                              (int)vQueries.size(), as_orbx(*F.keysUn), d2.data(), F.uRight ? F.uRight->data() : nullptr, n2,
                                              F.minX, F.maxX, F.minY, F.maxY, mfNNratio, q.transient.data(), vnClaimed.data(), &nmatches),
          "SearchByProjection");
    return nmatches;
}

int ORBmatcher::SearchByProjection(const FrameView& F, const std::vector<ProjectedQuery>& vQueries, std::vector<int>& vnClaimed, int mode,
                                   int thDist)
{
    const int n2 = (int)F.keysUn->size();
    if ((int)vnClaimed.size() != n2) throw std::runtime_error("SearchByProjection: vnClaimed size");
    QueryArrays q(vQueries);
    std::vector<unsigned char> d2 = packed_rows(*F.descriptors);
    int nmatches = 0;
    check(orbx_search_by_projection_lastframe(Context(), q.u.data(), q.v.data(), q.r.data(), q.ur.data(), q.level.data(), q.angle.data(),
                                              q.desc.data(), (int)vQueries.size(), as_orbx(*F.keysUn), d2.data(),
                                              F.uRight ? F.uRight->data() : nullptr, n2, F.minX, F.maxX, F.minY, F.maxY, mode, thDist,
                                              mbCheckOrientation ? 1 : 0, q.transient.data(), vnClaimed.data(), &nmatches),
          "SearchByProjection");
    return nmatches;
}

namespace {
// DBoW2::FeatureVector -> orbx_feature_vector (std::map iterates node ids in ascending order)
struct FlatFeatureVector {
    std::vector<uint32_t> ids; std::vector<int32_t> ptr, feat; orbx_feature_vector fv;
    explicit FlatFeatureVector(const FeatureVectorMap& m)
    {
        ptr.push_back(0);
        for (const auto& kv : m) {
            ids.push_back(kv.first);
            for (unsigned int i : kv.second) feat.push_back((int32_t)i);
            ptr.push_back((int32_t)feat.size());
        }
        fv.node_ids = ids.data(); fv.node_ptr = ptr.data(); fv.features = feat.data(); fv.n_nodes = (int32_t)ids.size();
    }
};
std::vector<unsigned char> flags(const std::vector<bool>& v, size_t n)
{
    std::vector<unsigned char> o(n, 0);
    for (size_t i = 0; i < n && i < v.size(); ++i) o[i] = v[i] ? 1 : 0;
    return o;
}
std::vector<float> angles(const std::vector<cv::KeyPoint>& k)
{
    std::vector<float> a(k.size());
    for (size_t i = 0; i < k.size(); ++i) a[i] = k[i].angle;
    return a;
}
} // namespace

int ORBmatcher::SearchByProjectionKeyFrame(const FrameView& KF, const std::vector<ProjectedQuery>& vQueries, std::vector<int>& vnClaimed)
{
    const int n2 = (int)KF.keysUn->size(), nq = (int)vQueries.size();
    vnClaimed.resize(n2, -1);
    if (nq == 0 || n2 == 0) return 0;
    std::vector<float> u(nq), v(nq), r(nq);
    std::vector<int> level(nq);
    std::vector<unsigned char> desc((size_t)nq * 32);
    for (int i = 0; i < nq; ++i) {
        u[i] = vQueries[i].u; v[i] = vQueries[i].v; r[i] = vQueries[i].radius; level[i] = vQueries[i].level;
        std::copy(vQueries[i].descriptor, vQueries[i].descriptor + 32, desc.begin() + (size_t)i * 32);
    }
    std::vector<unsigned char> d2 = packed_rows(*KF.descriptors);
    int nmatches = 0;
    check(orbx_search_by_projection_keyframe(Context(), u.data(), v.data(), r.data(), level.data(), desc.data(), nq, as_orbx(*KF.keysUn),
                                             d2.data(), n2, KF.minX, KF.maxX, KF.minY, KF.maxY, vnClaimed.data(), &nmatches),
          "SearchByProjection");
    return nmatches;
}

void ORBmatcher::FuseCandidates(const FrameView& KF, const std::vector<ProjectedQuery>& vQueries, bool bCheckReprojection,
                                const std::vector<float>& vInvLevelSigma2, std::vector<int>& vnBestIdx, std::vector<int>& vnBestDist)
{
    const int n2 = (int)KF.keysUn->size(), nq = (int)vQueries.size();
    vnBestIdx.assign(nq, -1); vnBestDist.assign(nq, 0x7FFFFFFF);
    if (nq == 0 || n2 == 0) return;
    if (bCheckReprojection && !KF.uRight) throw std::runtime_error("FuseCandidates: the reprojection test needs mvuRight");
    QueryArrays q(vQueries);
    std::vector<unsigned char> d2 = packed_rows(*KF.descriptors);
    check(orbx_project_best1(Context(), q.u.data(), q.v.data(), q.r.data(), q.level.data(), q.desc.data(), nq, as_orbx(*KF.keysUn), d2.data(), n2,
                             KF.minX, KF.maxX, KF.minY, KF.maxY, bCheckReprojection ? q.ur.data() : nullptr,
                             bCheckReprojection ? KF.uRight->data() : nullptr, bCheckReprojection ? vInvLevelSigma2.data() : nullptr,
                             (int)vInvLevelSigma2.size(), vnBestIdx.data(), vnBestDist.data()), "Fuse");
}

int ORBmatcher::SearchBySim3(const FrameView& KF1, const std::vector<ProjectedQuery>& vQueries1, const std::vector<bool>& vbValid1,
                             const FrameView& KF2, const std::vector<ProjectedQuery>& vQueries2, const std::vector<bool>& vbValid2,
                             std::vector<int>& vnMatches12)
{
    const int n1 = (int)KF1.keysUn->size(), n2 = (int)KF2.keysUn->size();
    if ((int)vQueries1.size() != n1 || (int)vQueries2.size() != n2) throw std::runtime_error("SearchBySim3: one query per feature");
    vnMatches12.assign(n1, -1);
    if (n1 == 0 || n2 == 0) return 0;
    QueryArrays q1(vQueries1), q2(vQueries2);
    std::vector<unsigned char> d1 = packed_rows(*KF1.descriptors), d2 = packed_rows(*KF2.descriptors), f1 = flags(vbValid1, n1), f2 = flags(vbValid2, n2);
    int nFound = 0;
    check(orbx_search_by_sim3(Context(), f1.data(), q1.u.data(), q1.v.data(), q1.r.data(), q1.level.data(), q1.desc.data(), as_orbx(*KF1.keysUn),
                              d1.data(), n1, f2.data(), q2.u.data(), q2.v.data(), q2.r.data(), q2.level.data(), q2.desc.data(),
                              as_orbx(*KF2.keysUn), d2.data(), n2, KF2.minX, KF2.maxX, KF2.minY, KF2.maxY, vnMatches12.data(), &nFound),
          "SearchBySim3");
    return nFound;
}

int ORBmatcher::SearchByBoW(const FrameView& KF, const std::vector<bool>& vbGoodMapPointKF, const FeatureVectorMap& vFeatVecKF,
                            const FrameView& F, const std::vector<cv::KeyPoint>& vKeysF, const FeatureVectorMap& vFeatVecF,
                            std::vector<int>& vnMatchesF)
{
    const int n1 = (int)KF.keysUn->size(), n2 = (int)vKeysF.size();
    std::vector<unsigned char> d1 = packed_rows(*KF.descriptors), d2 = packed_rows(*F.descriptors), e1 = flags(vbGoodMapPointKF, n1);
    std::vector<float> a1 = angles(*KF.keysUn), a2 = angles(vKeysF);
    FlatFeatureVector f1(vFeatVecKF), f2(vFeatVecF);
    std::vector<int> m12(n1, -1);
    vnMatchesF.assign(n2, -1);
    int nmatches = 0;
    check(orbx_search_by_bow(Context(), d1.data(), a1.data(), e1.data(), n1, &f1.fv, d2.data(), a2.data(), nullptr, n2, &f2.fv,
                             mfNNratio, mbCheckOrientation ? 1 : 0, 0, m12.data(), vnMatchesF.data(), &nmatches), "SearchByBoW");
    return nmatches;
}

int ORBmatcher::SearchByBoW(const FrameView& KF1, const std::vector<bool>& vbGoodMapPoint1, const FeatureVectorMap& vFeatVec1,
                            const FrameView& KF2, const std::vector<bool>& vbGoodMapPoint2, const FeatureVectorMap& vFeatVec2,
                            std::vector<int>& vnMatches12)
{
    const int n1 = (int)KF1.keysUn->size(), n2 = (int)KF2.keysUn->size();
    std::vector<unsigned char> d1 = packed_rows(*KF1.descriptors), d2 = packed_rows(*KF2.descriptors);
    std::vector<unsigned char> e1 = flags(vbGoodMapPoint1, n1), e2 = flags(vbGoodMapPoint2, n2);
    std::vector<float> a1 = angles(*KF1.keysUn), a2 = angles(*KF2.keysUn);
    FlatFeatureVector f1(vFeatVec1), f2(vFeatVec2);
    std::vector<int> m21(n2, -1);
    vnMatches12.assign(n1, -1);
    int nmatches = 0;
    check(orbx_search_by_bow(Context(), d1.data(), a1.data(), e1.data(), n1, &f1.fv, d2.data(), a2.data(), e2.data(), n2, &f2.fv,
                             mfNNratio, mbCheckOrientation ? 1 : 0, 1, vnMatches12.data(), m21.data(), &nmatches), "SearchByBoW");
    return nmatches;
}

int ORBmatcher::SearchForTriangulation(const FrameView& KF1, const std::vector<bool>& vbMapPoint1, const FeatureVectorMap& vFeatVec1,
                                       const FrameView& KF2, const std::vector<bool>& vbMapPoint2, const FeatureVectorMap& vFeatVec2,
                                       const float F12[9], float ex, float ey, const std::vector<float>& vScaleFactors2,
                                       const std::vector<float>& vLevelSigma2_2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                                       const bool bOnlyStereo)
{
    const int n1 = (int)KF1.keysUn->size(), n2 = (int)KF2.keysUn->size();
    if (!KF1.uRight || !KF2.uRight) throw std::runtime_error("SearchForTriangulation: FrameView::uRight (mvuRight) is required");
    std::vector<unsigned char> d1 = packed_rows(*KF1.descriptors), d2 = packed_rows(*KF2.descriptors);
    std::vector<unsigned char> h1 = flags(vbMapPoint1, n1), h2 = flags(vbMapPoint2, n2);
    FlatFeatureVector f1(vFeatVec1), f2(vFeatVec2);
    std::vector<int> m12(n1, -1);
    int nmatches = 0;
    check(orbx_search_for_triangulation(Context(), as_orbx(*KF1.keysUn), d1.data(), h1.data(), KF1.uRight->data(), n1, &f1.fv,
                                        as_orbx(*KF2.keysUn), d2.data(), h2.data(), KF2.uRight->data(), n2, &f2.fv, F12, ex, ey,
                                        vScaleFactors2.data(), vLevelSigma2_2.data(), (int)vScaleFactors2.size(), bOnlyStereo ? 1 : 0,
                                        mbCheckOrientation ? 1 : 0, m12.data(), &nmatches), "SearchForTriangulation");
    vMatchedPairs.clear();                                                       // :815-823
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return nmatches;
}

void ORBmatcher::StereoRowSearch(const std::vector<cv::KeyPoint>& vKeysLeft, const cv::Mat& descLeft,
                                 const std::vector<cv::KeyPoint>& vKeysRight, const cv::Mat& descRight,
                                 const std::vector<float>& vScaleFactors, int nRows, float minD, float maxD,
                                 std::vector<int>& vBestDist, std::vector<int>& vBestIdxR)
{
    const int nl = (int)vKeysLeft.size(), nr = (int)vKeysRight.size();
    vBestDist.assign(nl, -1); vBestIdxR.assign(nl, -1);
    std::vector<unsigned char> dl = packed_rows(descLeft), dr = packed_rows(descRight);
    check(orbx_stereo_match(Context(), as_orbx(vKeysLeft), dl.data(), nl, as_orbx(vKeysRight), dr.data(), nr, vScaleFactors.data(),
                            (int)vScaleFactors.size(), nRows, minD, maxD, vBestDist.data(), vBestIdxR.data()),
          "StereoRowSearch");
}

void ORBmatcher::ComputeStereoMatches(ORBextractor& extractorLeft, ORBextractor& extractorRight,
                                      const std::vector<cv::KeyPoint>& vKeysLeft, const cv::Mat& descLeft,
                                      const std::vector<cv::KeyPoint>& vKeysRight, const cv::Mat& descRight,
                                      float bf, float b, std::vector<float>& vuRight, std::vector<float>& vDepth)
{
    const int nl = (int)vKeysLeft.size(), nr = (int)vKeysRight.size();
    vuRight.assign(nl, -1.0f); vDepth.assign(nl, -1.0f);                       // src/Frame.cc:468-469
    if (nl == 0) return;
    const float minD = 0, maxD = bf / b;                                       // :496-498
    std::vector<int> vBestDist, vBestIdxR;
    StereoRowSearch(vKeysLeft, descLeft, vKeysRight, descRight, extractorLeft.GetScaleFactors(), extractorLeft.ImageHeight(),
                    minD, maxD, vBestDist, vBestIdxR);
    check(orbx_stereo_refine(extractorLeft.Handle(), extractorRight.Handle(), as_orbx(vKeysLeft), nl, as_orbx(vKeysRight), nr,
                             vBestDist.data(), vBestIdxR.data(), bf, minD, maxD, vuRight.data(), vDepth.data()),
          "ComputeStereoMatches");
}

} // namespace ORB_SLAM2
