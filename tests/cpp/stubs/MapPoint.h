// Compile-only mirror of the members of ORB_SLAM2::MapPoint that the matcher adapters touch
// (reference include/MapPoint.h, line of each declaration on its right).  Declarations only.
#pragma once
#include <cstddef>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class Frame;
class MapPoint {
public:
    cv::Mat GetWorldPos();                                   // :106
    cv::Mat GetNormal();                                     // :116
    int Observations();                                      // :133
    void AddObservation(KeyFrame* pKF, size_t idx);          // :139
    int GetIndexInKeyFrame(KeyFrame* pKF);                   // :161
    bool IsInKeyFrame(KeyFrame* pKF);                        // :169
    bool isBad();                                            // :189
    void Replace(MapPoint* pMP);                             // :192
    cv::Mat GetDescriptor();                                 // :223
    float GetMinDistanceInvariance();                        // :237
    float GetMaxDistanceInvariance();                        // :245
    int PredictScale(const float& currentDist, KeyFrame* pKF);   // :254
    int PredictScale(const float& currentDist, Frame* pF);       // :255
    float mTrackProjX;                                       // :291
    float mTrackProjY;                                       // :292
    float mTrackProjXR;                                      // :293
    bool mbTrackInView;                                      // :294
    int mnTrackScaleLevel;                                   // :295
    float mTrackViewCos;                                     // :300
};
}
