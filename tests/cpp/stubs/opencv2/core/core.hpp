// Compile-only stand-in for <opencv2/core/core.hpp>: the types of cv_standin.h plus DECLARATIONS (no definitions) of the
// cv::Mat arithmetic that the reference-tree adapters use.  Syntax checks only; nothing here runs.
#pragma once
#define ORBX_STANDIN_DECLARE_ARITHMETIC
#include "../../../../../visual-slam_amd/cpp/cv_standin.h"
#define CV_8UC(n) (((n) - 1) << 3)
namespace cv {
Mat operator*(const Mat& a, const Mat& b);      // matrix product (MatExpr in OpenCV; converts to Mat)
Mat operator*(double s, const Mat& a);
Mat operator*(const Mat& a, double s);
Mat operator/(const Mat& a, double s);
Mat operator+(const Mat& a, const Mat& b);
Mat operator-(const Mat& a, const Mat& b);
Mat operator-(const Mat& a);
double norm(const Mat& a);
}
