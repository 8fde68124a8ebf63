/*
 * TEST INFRASTRUCTURE -- stand-in for the part of OpenCV 4.2 the unmodified reference sources use: cv::Mat as a reference-counted
 * 2-D array (utility.cpp:103-104, Removerter.cpp:121,570-572: constructor with a fill value, at<T>(), rows / cols, element-wise
 * float subtraction) plus what utility.h:114-127 (convertColorMappedImg) needs to compile.  OpenCV is not in /root/reference.
 * The colour mapping is visualisation only (the images go to RViz topics, Removerter.cpp:580-585) and is skipped unless
 * refshim::viz_enabled() (REFSHIM_VIZ=1): the arithmetic below then follows cv::MatExpr loosely and is NOT a parity statement.
 * Original code.
 */
#ifndef REFSHIM_CV_H
#define REFSHIM_CV_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "refshim/ros.h"

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32SC1 4
#define CV_32FC1 5

namespace refshim {
inline bool viz_enabled()
{
    static const bool v = std::getenv("REFSHIM_VIZ") != nullptr;
    return v;
}
}

namespace cv {

struct Scalar {
    double v[4];
    static Scalar all(double x) { return Scalar{{x, x, x, x}}; }
};

enum ColormapTypes { COLORMAP_JET = 2 };

class Mat {
public:
    int rows = 0, cols = 0;

    Mat() {}
    Mat(int r, int c, int type, const Scalar& s) : rows(r), cols(c), type_(type)
    {
        alloc();
        const size_t n = (size_t)r * c;
        if (type == CV_32FC1) { float* p = ptr<float>(); for (size_t i = 0; i < n; ++i) p[i] = (float)s.v[0]; }
        else if (type == CV_32SC1) { int* p = ptr<int>(); for (size_t i = 0; i < n; ++i) p[i] = (int)s.v[0]; }
        else std::memset(buf_->data(), (int)s.v[0], buf_->size());
    }
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type) { alloc(); }

    int type() const { return type_; }
    bool empty() const { return !buf_ || rows == 0 || cols == 0; }
    template <class T> T& at(int r, int c) { return ptr<T>()[(size_t)r * cols + c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>()[(size_t)r * cols + c]; }
    template <class T> T* ptr() { return reinterpret_cast<T*>(buf_->data()); }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(buf_->data()); }

    void convertTo(Mat& dst, int rtype) const
    {
        if (!refshim::viz_enabled()) { dst = *this; return; }
        Mat out(rows, cols, rtype);
        const size_t n = (size_t)rows * cols;
        for (size_t i = 0; i < n; ++i) {
            const double v = type_ == CV_32FC1 ? (double)ptr<float>()[i] : type_ == CV_32SC1 ? (double)ptr<int>()[i] : (double)ptr<uint8_t>()[i];
            long q = std::isnan(v) ? 0 : std::lrint(v);
            out.ptr<uint8_t>()[i] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
        }
        dst = out;
    }

    static size_t elem_size(int type) { return type == CV_8UC1 ? 1 : type == CV_8UC3 ? 3 : 4; }

private:
    void alloc() { buf_ = std::make_shared<std::vector<uint8_t>>((size_t)rows * cols * elem_size(type_)); }
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<uint8_t>> buf_;      /* shared like cv::Mat's reference-counted header */
};

/* scan_rimg - map_rimg (Removerter.cpp:572 etc.): element-wise, float for CV_32F images */
inline Mat operator-(const Mat& a, const Mat& b)
{
    Mat out(a.rows, a.cols, a.type());
    const size_t n = (size_t)a.rows * a.cols;
    if (a.type() == CV_32FC1) { for (size_t i = 0; i < n; ++i) out.ptr<float>()[i] = a.ptr<float>()[i] - b.ptr<float>()[i]; }
    else if (a.type() == CV_32SC1) { for (size_t i = 0; i < n; ++i) out.ptr<int>()[i] = a.ptr<int>()[i] - b.ptr<int>()[i]; }
    else { for (size_t i = 0; i < n; ++i) out.ptr<uint8_t>()[i] = (uint8_t)(a.ptr<uint8_t>()[i] - b.ptr<uint8_t>()[i]); }
    return out;
}

namespace detail {
template <class F> inline Mat map_scalar(const Mat& a, F f)
{
    if (!refshim::viz_enabled()) return a;           /* visualisation arithmetic only (utility.h:121) */
    Mat out(a.rows, a.cols, a.type());
    const size_t n = (size_t)a.rows * a.cols;
    if (a.type() == CV_32FC1) { for (size_t i = 0; i < n; ++i) out.ptr<float>()[i] = (float)f((double)a.ptr<float>()[i]); }
    else if (a.type() == CV_32SC1) { for (size_t i = 0; i < n; ++i) out.ptr<int>()[i] = (int)std::lrint(f((double)a.ptr<int>()[i])); }
    else { for (size_t i = 0; i < n; ++i) out.ptr<uint8_t>()[i] = (uint8_t)std::lrint(f((double)a.ptr<uint8_t>()[i])); }
    return out;
}
}
inline Mat operator-(const Mat& a, double s) { return detail::map_scalar(a, [s](double v) { return v - s; }); }
inline Mat operator*(double s, const Mat& a) { return detail::map_scalar(a, [s](double v) { return s * v; }); }
inline Mat operator*(const Mat& a, double s) { return detail::map_scalar(a, [s](double v) { return s * v; }); }
inline Mat operator/(const Mat& a, double s) { return detail::map_scalar(a, [s](double v) { return v / s; }); }

inline void applyColorMap(const Mat& src, Mat& dst, int)
{
    if (!refshim::viz_enabled()) { dst = src; return; }
    Mat out(src.rows, src.cols, CV_8UC3);
    const size_t n = (size_t)src.rows * src.cols;
    for (size_t i = 0; i < n; ++i) {
        const double x = src.ptr<uint8_t>()[i] / 255.0;
        const double ch[3] = {std::min(4 * x + 0.5, -4 * x + 2.5), std::min(4 * x - 0.5, -4 * x + 3.5), std::min(4 * x - 1.5, -4 * x + 4.5)};
        for (int c = 0; c < 3; ++c) out.ptr<uint8_t>()[3 * i + c] = (uint8_t)std::lrint(255.0 * std::min(1.0, std::max(0.0, ch[c])));
    }
    dst = out;
}

} // namespace cv

namespace cv_bridge {
class CvImage {
public:
    CvImage(const std_msgs::Header& h, const std::string& enc, const cv::Mat& img) : header_(h), encoding_(enc), image_(img) {}
    sensor_msgs::ImagePtr toImageMsg() const
    {
        sensor_msgs::ImagePtr m(new sensor_msgs::Image());
        m->header = header_; m->encoding = encoding_; m->height = image_.rows; m->width = image_.cols;
        return m;
    }
private:
    std_msgs::Header header_;
    std::string encoding_;
    cv::Mat image_;
};
}

#endif
