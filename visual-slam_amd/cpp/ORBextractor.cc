#include "ORBextractor.h"
#include <stdexcept>
#include <string>
#include <cstring>
#include <algorithm>

namespace ORB_SLAM2
{

static void check(int status, const char* what)
{
    if (status != ORBX_OK)
        throw std::runtime_error(std::string(what) + ": " + orbx_status_string(status) + " (" + orbx_last_error() + ")");
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    mParams.nfeatures = _nfeatures; mParams.scale_factor = _scaleFactor; mParams.nlevels = _nlevels;
    mParams.ini_th_fast = _iniThFAST; mParams.min_th_fast = _minThFAST; mParams.blur_rounding = 0;
    // cos / sin as the reference's own callee computes them: glibc's cosf / sinf (src/ORBextractor.cc:111-112), bit for bit
    mParams.trig_mode = 1;
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
    check(orbx_scale_tables(&mParams, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                            mvInvLevelSigma2.data(), mnFeaturesPerLevel.data(), umax.data()), "ORBextractor");
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { orbx_extractor_destroy(mpContext); }

void ORBextractor::SetDevice(int device)
{
    if (device != mnDevice) { orbx_extractor_destroy(mpContext); mpContext = nullptr; mnDevice = device; }
}

void ORBextractor::SetBlurRounding(int mode)
{
    if (mode != mParams.blur_rounding) { orbx_extractor_destroy(mpContext); mpContext = nullptr; mParams.blur_rounding = mode; }
}

void ORBextractor::SetTrigMode(int mode)
{
    if (mode != mParams.trig_mode) { orbx_extractor_destroy(mpContext); mpContext = nullptr; mParams.trig_mode = mode; }
}

void ORBextractor::SetInputFormat(int format)
{
    if (format < ORBX_INPUT_GRAY || format > ORBX_INPUT_BGRA) throw std::runtime_error("ORBextractor: unknown input format");
    mnFormat = format;
    if (mpContext) check(orbx_extractor_set_input_format(mpContext, format), "ORBextractor: input format");
}

void ORBextractor::EnsureContext(int width, int height)
{
    if (mpContext && width == mnWidth && height == mnHeight) return;
    orbx_extractor_destroy(mpContext);
    mpContext = nullptr;
    check(orbx_extractor_create(&mParams, width, height, 1, mnDevice, &mpContext), "ORBextractor: context");
    check(orbx_extractor_set_input_format(mpContext, mnFormat), "ORBextractor: input format");
    mnWidth = width; mnHeight = height;
    mvStaging.resize(orbx_keypoint_capacity(&mParams, width, height));
}

void ORBextractor::UndistortKeyPoints(const std::vector<cv::KeyPoint>& vKeys, const orbx_camera& cam, std::vector<cv::KeyPoint>& vKeysUn)
{
    vKeysUn = vKeys;
    if (cam.k1 == 0.0f || vKeys.empty()) return;                        // src/Frame.cc:406-410
    if (!mpContext) throw std::runtime_error("ORBextractor::UndistortKeyPoints: call operator() first");
    std::vector<float> xy(vKeys.size() * 2), out(vKeys.size() * 2);
    for (size_t i = 0; i < vKeys.size(); ++i) { xy[2 * i] = vKeys[i].pt.x; xy[2 * i + 1] = vKeys[i].pt.y; }
    check(orbx_undistort_points(mpContext, &cam, xy.data(), (int)vKeys.size(), out.data()), "ORBextractor: undistort");
    for (size_t i = 0; i < vKeys.size(); ++i) { vKeysUn[i].pt.x = out[2 * i]; vKeysUn[i].pt.y = out[2 * i + 1]; }
}

void ORBextractor::ComputeImageBounds(int cols, int rows, const orbx_camera& cam, float& minX, float& maxX, float& minY, float& maxY)
{
    if (cam.k1 == 0.0f) { minX = 0.0f; maxX = (float)cols; minY = 0.0f; maxY = (float)rows; return; }
    if (!mpContext) throw std::runtime_error("ORBextractor::ComputeImageBounds: call operator() first");
    const float corners[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
    float un[8];
    check(orbx_undistort_points(mpContext, &cam, corners, 4, un), "ORBextractor: undistort");
    minX = std::min(un[0], un[4]); maxX = std::max(un[2], un[6]);     // src/Frame.cc:451-454
    minY = std::min(un[1], un[3]); maxY = std::max(un[5], un[7]);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors)
{
    if (_image.empty())
        return;
    const cv::Mat image = _image.getMat();
    const int channels = mnFormat == ORBX_INPUT_GRAY ? 1 : (mnFormat <= ORBX_INPUT_BGR ? 3 : 4);
#ifdef ORBX_WITH_OPENCV
    if (image.type() != CV_8UC(channels)) throw std::runtime_error("ORBextractor: image type does not match the input format");
    const int width = image.cols;
#else
    // the stand-in Mat counts bytes: a colour image is passed as a matrix of width * channels columns
    if (image.type() != CV_8UC1 || image.cols % channels) throw std::runtime_error("ORBextractor: image must be 8-bit, width * channels columns");
    const int width = image.cols / channels;
#endif
    EnsureContext(width, image.rows);

    const int cap = (int)mvStaging.size();
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0;
    check(orbx_extract(mpContext, image.ptr(), (ptrdiff_t)image.step, mvStaging.data(), desc.ptr(), cap, &n), "ORBextractor: extract");

    if (n == 0)
        _descriptors.release();
    else {
        _descriptors.create(n, 32, CV_8U);                       // src/ORBextractor.cc:1067-1068
        cv::Mat out = _descriptors.getMat();
        for (int r = 0; r < n; ++r) std::memcpy(out.ptr(r), desc.ptr(r), 32);
    }
    _keypoints.clear();
    _keypoints.reserve(n);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "keypoint layouts must agree");
    for (int i = 0; i < n; ++i) {
        const orbx_keypoint& k = mvStaging[i];
        _keypoints.push_back(cv::KeyPoint(k.x, k.y, k.size, k.angle, k.response, k.octave, k.class_id));
    }

    if (mbDownloadPyramid) {
        std::vector<cv::Mat> padded(nlevels);
        std::vector<uint8_t*> ptrs(nlevels);
        std::vector<ptrdiff_t> strides(nlevels);
        std::vector<int> ws(nlevels), hs(nlevels);
        for (int level = 0; level < nlevels; ++level) {
            check(orbx_level_size(&mParams, width, image.rows, level, &ws[level], &hs[level]), "ORBextractor: level size");
            padded[level] = cv::Mat(hs[level] + 38, ws[level] + 38, CV_8UC1);
            ptrs[level] = padded[level].ptr(); strides[level] = (ptrdiff_t)padded[level].step;
        }
        check(orbx_pyramid_levels(mpContext, 0, ptrs.data(), strides.data()), "ORBextractor: pyramid");
        for (int level = 0; level < nlevels; ++level)
            mvImagePyramid[level] = padded[level](cv::Rect(19, 19, ws[level], hs[level]));     // an ROI of the padded buffer, as in the reference
    }
}

} //namespace ORB_SLAM2
