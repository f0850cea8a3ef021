// Compile-only mirror of the reference's class ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:56-330: the
// declarations the adapters must define, line of each on its right).  The reference's header says `vector` and `pair`
// unqualified in places (:199, :233, :248); they are qualified here and mean the same types.
#pragma once
#include <set>
#include <utility>
#include <vector>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
namespace ORB_SLAM2 {
class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);                                                  // :66
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);                                      // :69
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);        // :84
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);  // :104
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound,
                           const float th, const int ORBdist);                                              // :119
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                           std::vector<MapPoint*>& vpMatched, int th);                                      // :135
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);                    // :150
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);                   // :162
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10);                        // :179
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                               std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo);   // :198-199
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12,
                     const cv::Mat& R12, const cv::Mat& t12, const float th);                               // :219
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);               // :233
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th,
             std::vector<MapPoint*>& vpReplacePoint);                                                       // :248
    static const int TH_LOW;                                                                                // :259
    static const int TH_HIGH;                                                                               // :267
    static const int HISTO_LENGTH;                                                                          // :272
protected:
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);   // :291
    float RadiusByViewingCos(const float& viewCos);                                                         // :298
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);         // :313
    float mfNNratio;                                                                                        // :321
    bool mbCheckOrientation;                                                                                // :329
};
}
