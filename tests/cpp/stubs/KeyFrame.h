// Compile-only mirror of the members of ORB_SLAM2::KeyFrame that the matcher adapters touch
// (reference include/KeyFrame.h, line of each declaration on its right).  Declarations only.
#pragma once
#include <cstddef>
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
namespace ORB_SLAM2 {
class MapPoint;
class KeyFrame {
public:
    cv::Mat GetCameraCenter();                               // :133
    cv::Mat GetRotation();                                   // :142
    cv::Mat GetTranslation();                                // :145
    void AddMapPoint(MapPoint* pMP, const size_t& idx);      // :276
    std::set<MapPoint*> GetMapPoints();                      // :319
    std::vector<MapPoint*> GetMapPointMatches();             // :322
    MapPoint* GetMapPoint(const size_t& idx);                // :338
    bool IsInImage(const float& x, const float& y) const;    // :362
    const float fx, fy, cx, cy, invfx, invfy, mbf, mb, mThDepth;   // :551
    const int N;                                             // :554
    const std::vector<cv::KeyPoint> mvKeysUn;                // :562
    const std::vector<float> mvuRight;                       // :563
    const cv::Mat mDescriptors;                              // :567
    DBoW2::FeatureVector mFeatVec;                           // :573
    const std::vector<float> mvScaleFactors;                 // :587
    const std::vector<float> mvLevelSigma2;                  // :589
    const std::vector<float> mvInvLevelSigma2;               // :591
    const int mnMinX, mnMinY, mnMaxX, mnMaxY;                // :595-598
};
}
