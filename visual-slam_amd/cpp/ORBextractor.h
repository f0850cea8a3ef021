// Drop-in for the reference's ORB_SLAM2::ORBextractor (include/ORBextractor.h:74-282): same constructor,
// operator(), getters and the public mvImagePyramid member; the work runs on the MI355X through the C ABI
// of include/orbx.h.  ExtractorNode / DistributeOctTree / ComputePyramid are not members here: they live
// in the device kernels.
#ifndef ORBX_ORBEXTRACTOR_H
#define ORBX_ORBEXTRACTOR_H

#include <vector>
#include "cv_compat.h"
#include "../../include/orbx.h"

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    // include/ORBextractor.h:83-84
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // include/ORBextractor.h:102-104.  The mask is ignored, as in the reference.  An empty image returns
    // silently and leaves the outputs untouched (src/ORBextractor.cc:1045-1046); any other failure throws
    // std::runtime_error where the reference would surface a cv::Exception.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // include/ORBextractor.h:150: level images as ROIs of (w+38) x (h+38) buffers with the REFLECT_101 frame,
    // refreshed by every operator() call (read by Frame::ComputeStereoMatches, src/Frame.cc:563-580).
    std::vector<cv::Mat> mvImagePyramid;

    // Extensions (no counterpart in the reference): skip the device-to-host copy of the pyramid when the
    // caller does not read mvImagePyramid (monocular tracking); choose the GPU; pick the GaussianBlur tie
    // rounding (orbx_params::blur_rounding).
    void SetPyramidDownload(bool on) { mbDownloadPyramid = on; }
    void SetDevice(int device);
    // Colour frames straight into operator(): ORBX_INPUT_RGB / BGR / RGBA / BGRA make the first pyramid kernel apply the
    // grey conversion Tracking::GrabImage* does on the host (src/Tracking.cc:254-267), so that call site can drop its cvtColor.
    void SetInputFormat(int format);
    void SetBlurRounding(int mode);
    // cos / sin of the keypoint angle (orbx_params::trig_mode): 1 (default) / 2 glibc's cosf / sinf bit for bit, 0 correctly rounded
    // (what src/ORBextractor.cc:112 calls on a glibc system)
    void SetTrigMode(int mode);
    // Frame::UndistortKeyPoints (src/Frame.cc:404-434) for the keypoints of the last operator() call's image size:
    // cam = mK and mDistCoef (k3 = 0 when the settings file has none).  k1 == 0 copies, as the reference does.
    void UndistortKeyPoints(const std::vector<cv::KeyPoint>& vKeys, const orbx_camera& cam, std::vector<cv::KeyPoint>& vKeysUn);
    // Frame::ComputeImageBounds (src/Frame.cc:436-464) for an image of cols x rows
    void ComputeImageBounds(int cols, int rows, const orbx_camera& cam, float& minX, float& maxX, float& minY, float& maxY);
    orbx_extractor* Handle() { return mpContext; }
    int ImageHeight() const { return mnHeight; }    // mvImagePyramid[0].rows of the last operator() call

protected:
    void EnsureContext(int width, int height);

    int nfeatures;
    double scaleFactor;      // a double in the reference too (include/ORBextractor.h:223)
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

    orbx_params mParams;
    orbx_extractor* mpContext = nullptr;
    int mnFormat = ORBX_INPUT_GRAY;
    int mnWidth = 0, mnHeight = 0, mnDevice = -1;
    bool mbDownloadPyramid = true;
    std::vector<orbx_keypoint> mvStaging;
};

} //namespace ORB_SLAM2

#endif
