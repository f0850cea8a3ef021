// Drop-in for the matching loops of the reference's ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:56-330).
// Frame / KeyFrame / MapPoint belong to the reference's map data model and are not rebuilt here; the
// Search* methods take a FrameView (the fields of Frame the reference reads) and plain query arrays.
// A maintainer of the reference keeps the original signatures and forwards to these from the places cited
// on each method.
#ifndef ORBX_ORBMATCHER_H
#define ORBX_ORBMATCHER_H

#include <map>
#include <utility>
#include <vector>
#include "cv_compat.h"
#include "ORBextractor.h"
#include "../../include/orbx.h"

// Inside the reference tree the reference keeps its own ORB_SLAM2::ORBmatcher (the Frame& / KeyFrame* signatures,
// implemented by cpp/reference_adapters/ORBmatcher_adapters.cc on top of this class): build with
// -DORBX_MATCHER_CLASS=OrbxMatcherEngine there so that the two classes have different names.
#ifdef ORBX_MATCHER_CLASS
#define ORBmatcher ORBX_MATCHER_CLASS
#endif

namespace ORB_SLAM2
{

// What the matchers read from a Frame: mvKeysUn, mDescriptors (N x 32), mvuRight, and the static image bounds
// mnMinX.. (src/Frame.cc:457-463).
struct FrameView {
    const std::vector<cv::KeyPoint>* keysUn = nullptr;
    const cv::Mat* descriptors = nullptr;
    const std::vector<float>* uRight = nullptr;   // may be null (monocular)
    float minX = 0, maxX = 0, minY = 0, maxY = 0;
};

// One projected map point / last-frame keypoint (the values the reference computes before its candidate loop).
struct ProjectedQuery {
    float u, v;          // projection in the current frame
    float radius;        // search radius, already multiplied by the scale factor of the level
    float uRight;        // projected right-image column (stereo consistency test); ignored when FrameView::uRight is null
    int level;           // predicted level (map points) or octave of the last-frame keypoint
    float angle;         // keypoint angle of the source observation (rotation histogram)
    const unsigned char* descriptor; // 32 bytes
    // the map point has observations (pMP->Observations() > 0): its match blocks the feature for later queries
    // (src/ORBmatcher.cc:87-89, :1406-1408).  False for the temporal points of stereo / RGB-D tracking: such a match
    // is reported as -2 - query in vnClaimed and a later query may take the feature over, as in the reference.
    bool hasObservations = true;
};

// What DBoW2::FeatureVector is (Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22): node id -> indices of the features under it.
// A DBoW2::FeatureVector converts to this reference implicitly.
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVectorMap;

class ORBmatcher
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    ~ORBmatcher();
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    // src/ORBmatcher.cc:1650-1666
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // src/ORBmatcher.cc:408-523.  vbPrevMatched and vnMatches12 as in the reference.
    int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10);

    // src/ORBmatcher.cc:45-129 (candidate loop and acceptance).  vnClaimed[i] >= 0 marks frame keypoints that already
    // hold a map point with observations; a new match stores the query index.  Returns the number of matches.
    int SearchByProjection(const FrameView& F, const std::vector<ProjectedQuery>& vQueries, std::vector<int>& vnClaimed);

    // src/ORBmatcher.cc:1331-1473 and 1475-1602: best-1 within thDist (TH_HIGH or ORBdist), rotation histogram.
    // mode 0: levels level-1..level+1, 1: forward motion (>= level), 2: backward (0..level).
    int SearchByProjection(const FrameView& CurrentFrame, const std::vector<ProjectedQuery>& vQueries, std::vector<int>& vnClaimed,
                           int mode, int thDist);

    // src/ORBmatcher.cc:293-406, SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th): candidate loop and acceptance
    // (best-1 within TH_LOW among free features of level-1..level).  ProjectedQuery::radius = th * scale[level];
    // uRight and angle are not used.  vnClaimed as above.
    int SearchByProjectionKeyFrame(const FrameView& KF, const std::vector<ProjectedQuery>& vQueries, std::vector<int>& vnClaimed);

    // The candidate loop of both Fuse forms (src/ORBmatcher.cc:893-952 and :1052-1082): per projected map point (u, v,
    // radius = th * scale[level], level, descriptor; uRight = the projected right coordinate, :873) the nearest feature of
    // the key frame within the radius at level-1..level.  bCheckReprojection applies the chi-square test of
    // Fuse(pKF, vpMapPoints, th) (:917-941; needs KF.uRight and vInvLevelSigma2).  vnBestIdx[i] = -1 when none;
    // the caller accepts vnBestDist[i] <= TH_LOW and replaces / adds map points (:955-974, :1085-1099).
    void FuseCandidates(const FrameView& KF, const std::vector<ProjectedQuery>& vQueries, bool bCheckReprojection,
                        const std::vector<float>& vInvLevelSigma2, std::vector<int>& vnBestIdx, std::vector<int>& vnBestDist);

    // src/ORBmatcher.cc:1105-1329, SearchBySim3: vQueries1[i1] = feature i1 of key frame 1 projected into key frame 2 (descriptor =
    // its map point's), vbValid1[i1] = passed the tests of :1153-1186; likewise 2 into 1.  vnMatches12[i1] = feature of key frame 2
    // (the reference stores its map point in vpMatches12[i1], :1322) or -1.  Returns nFound.
    int SearchBySim3(const FrameView& KF1, const std::vector<ProjectedQuery>& vQueries1, const std::vector<bool>& vbValid1,
                     const FrameView& KF2, const std::vector<ProjectedQuery>& vQueries2, const std::vector<bool>& vbValid2,
                     std::vector<int>& vnMatches12);

    // src/ORBmatcher.cc:159-291, SearchByBoW(KeyFrame*, Frame&, ...).  vbGoodMapPointKF[i]: key-frame feature i has a map
    // point that is not bad; vKeysF = F.mvKeys (the reference takes the frame's angles from there, :241).
    // vnMatchesF[iF] = key-frame feature whose map point the reference stores in vpMapPointMatches[iF], or -1.
    int SearchByBoW(const FrameView& KF, const std::vector<bool>& vbGoodMapPointKF, const FeatureVectorMap& vFeatVecKF,
                    const FrameView& F, const std::vector<cv::KeyPoint>& vKeysF, const FeatureVectorMap& vFeatVecF,
                    std::vector<int>& vnMatchesF);

    // src/ORBmatcher.cc:525-658, SearchByBoW(KeyFrame*, KeyFrame*, ...).  vnMatches12[i1] = feature of key frame 2 whose
    // map point the reference stores in vpMatches12[i1], or -1.
    int SearchByBoW(const FrameView& KF1, const std::vector<bool>& vbGoodMapPoint1, const FeatureVectorMap& vFeatVec1,
                    const FrameView& KF2, const std::vector<bool>& vbGoodMapPoint2, const FeatureVectorMap& vFeatVec2,
                    std::vector<int>& vnMatches12);

    // src/ORBmatcher.cc:660-826.  vbMapPoint: the feature has a map point (any); FrameView::uRight must be set (mvuRight).
    // F12 row-major; (ex, ey) the epipole in image 2 (:667-673); scale factors and level sigma^2 of key frame 2.
    int SearchForTriangulation(const FrameView& KF1, const std::vector<bool>& vbMapPoint1, const FeatureVectorMap& vFeatVec1,
                               const FrameView& KF2, const std::vector<bool>& vbMapPoint2, const FeatureVectorMap& vFeatVec2,
                               const float F12[9], float ex, float ey, const std::vector<float>& vScaleFactors2,
                               const std::vector<float>& vLevelSigma2_2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo);

    // Frame::ComputeStereoMatches, Hamming part (src/Frame.cc:466-549).
    void StereoRowSearch(const std::vector<cv::KeyPoint>& vKeysLeft, const cv::Mat& descLeft,
                         const std::vector<cv::KeyPoint>& vKeysRight, const cv::Mat& descRight,
                         const std::vector<float>& vScaleFactors, int nRows, float minD, float maxD,
                         std::vector<int>& vBestDist, std::vector<int>& vBestIdxR);

    // Frame::ComputeStereoMatches whole (src/Frame.cc:466-639): the row search above, then the 11x11 SAD refinement on
    // the pyramids the two extractors hold from their last operator() call, parabola fit, disparity test and median
    // filter.  bf and b are Frame::mbf and Frame::mb; vuRight / vDepth are mvuRight / mvDepth.
    void ComputeStereoMatches(ORBextractor& extractorLeft, ORBextractor& extractorRight,
                              const std::vector<cv::KeyPoint>& vKeysLeft, const cv::Mat& descLeft,
                              const std::vector<cv::KeyPoint>& vKeysRight, const cv::Mat& descRight,
                              float bf, float b, std::vector<float>& vuRight, std::vector<float>& vDepth);

    // MapPoint::ComputeDistinctiveDescriptors, the selection at src/MapPoint.cc:513-542: index (BestIdx) of the descriptor
    // with the least median distance to the others; vDescriptors as the reference collects it (:500-508).  -1 if empty.
    int DistinctiveDescriptor(const std::vector<cv::Mat>& vDescriptors);
    // The same for many map points at once (local mapping after a key frame, or the whole map after a loop closure):
    // descriptor rows of all points back to back, point g owning rows vGroupPtr[g] .. vGroupPtr[g+1]).
    void DistinctiveDescriptors(const cv::Mat& descriptorRows, const std::vector<int>& vGroupPtr, std::vector<int>& vBestIdx);

    static void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    void SetDevice(int device);

protected:
    orbx_matcher* Context();
    float mfNNratio;
    bool mbCheckOrientation;
    orbx_matcher* mpContext = nullptr;
    int mnDevice = -1;
};

} // namespace ORB_SLAM2

#ifdef ORBX_MATCHER_CLASS
#undef ORBmatcher
#endif

#endif
