// Stand-ins for the few OpenCV types that appear in the reference's ORBextractor / ORBmatcher signatures
// (include/ORBextractor.h:102-104, include/ORBmatcher.h), with OpenCV's layouts: KeyPoint is 28 bytes, Mat rows are
// `step` bytes apart.  Included by cv_compat.h when the real OpenCV is not available.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };

struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Rect { int x, y, width, height; Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };

// 8-bit single-channel matrix with shared ownership and ROI views; enough for images, pyramid levels and
// N x 32 descriptor matrices.
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, void* ext, size_t step_ = 0) : rows(r), cols(c), step(step_ ? step_ : (size_t)c), data((uint8_t*)ext) {}
    void create(int r, int c, int /*type*/)
    {
        if (r == rows && c == cols && owner_ && step == (size_t)c) return;
        const size_t bytes = (size_t)r * (size_t)c;
        owner_ = std::shared_ptr<uint8_t>(new uint8_t[bytes > 0 ? bytes : 1], std::default_delete<uint8_t[]>());   // uninitialised, as cv::Mat::create leaves it
        rows = r; cols = c; step = (size_t)c; data = owner_.get();
    }
    void release() { owner_.reset(); rows = cols = 0; step = 0; data = nullptr; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return step == (size_t)cols; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + c * sizeof(T)); }
    template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + c * sizeof(T)); }
    Mat row(int r) const { Mat m; m.owner_ = owner_; m.rows = 1; m.cols = cols; m.step = step; m.data = data + (size_t)r * step; return m; }
    Mat operator()(const Rect& roi) const
    {
        Mat m; m.owner_ = owner_; m.rows = roi.height; m.cols = roi.width; m.step = step;
        m.data = data + (size_t)roi.y * step + roi.x; return m;
    }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
#ifdef ORBX_STANDIN_DECLARE_ARITHMETIC
    // declarations only, for compile checks of code written against the real cv::Mat (tests/cpp/stubs)
    template <typename T> T& at(int i);
    template <typename T> const T& at(int i) const;
    Mat col(int c) const;
    Mat t() const;
    double dot(const Mat& m) const;
#endif
    Mat clone() const
    {
        Mat m; if (empty()) return m;
        m.create(rows, cols, CV_8UC1);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols);
        return m;
    }
private:
    std::shared_ptr<uint8_t> owner_;
};

// The proxy types of OpenCV's signatures, with the part of their interface the drop-in classes use and nothing else:
// like the real cv::_InputArray / cv::_OutputArray they are NOT matrices, so code that compiles against these also
// compiles against the real headers (getMat() to reach the matrix, create() / release() to size an output).
class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }
private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

} // namespace cv
