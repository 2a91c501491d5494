// Plain value types standing in for the OpenCV/Eigen types on the reference's call surface
// (cv::Point2f, cv::KeyPoint, 8UC1 cv::Mat, Eigen::VectorXd calibration vectors, 4x4 cv::Mat extrinsics).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>

namespace feature_tracker {

struct Point2f {
    float x = 0.f, y = 0.f;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

// field-for-field the members of cv::KeyPoint that the reference touches
struct KeyPoint {
    Point2f pt;
    float size = 7.f;
    float angle = -1.f;
    float response = 0.f;
    int octave = 0;
    int class_id = -1;
};

// borrowed view of a grayscale image (cv::Mat of type 8UC1)
struct Image8 {
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;   // bytes per row
    Image8() = default;
    Image8(const uint8_t* d, int c, int r, size_t s) : data(d), cols(c), rows(r), step(s ? s : (size_t)c) {}
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

using Calib8 = std::array<double, 8>;     // fx, fy, cx, cy, d1, d2, d3, d4
using Mat4d = std::array<double, 16>;     // row-major 4x4 rigid transform
using Mat3d = std::array<double, 9>;
using Mat3f = std::array<float, 9>;
using Vec3d = std::array<double, 3>;

}  // namespace feature_tracker
