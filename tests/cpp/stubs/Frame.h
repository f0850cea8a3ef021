// Compile-only mirror of the members of ORB_SLAM2::Frame that the matcher adapters touch
// (reference include/Frame.h, line of each declaration on its right).  Declarations only.
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
namespace ORB_SLAM2 {
using std::vector;
class MapPoint;
class Frame {
public:
    static float fx, fy, cx, cy;                             // :134-137
    float mbf;                                               // :143
    float mb;                                                // :146
    int N;                                                   // :153
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;           // :158
    std::vector<cv::KeyPoint> mvKeysUn;                      // :159
    std::vector<float> mvuRight;                             // :163
    DBoW2::FeatureVector mFeatVec;                           // :168
    cv::Mat mDescriptors, mDescriptorsRight;                 // :171
    std::vector<MapPoint*> mvpMapPoints;                     // :174
    std::vector<bool> mvbOutlier;                            // :177
    cv::Mat mTcw;                                            // :185
    vector<float> mvScaleFactors;                            // :198
    vector<float> mvLevelSigma2;                             // :200
    vector<float> mvInvLevelSigma2;                          // :201
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;             // :204-207
};
}
