// TEST-ONLY stand-in for the part of cv::Mat include/detection_6d_foundationpose_amd.hpp touches (see Eigen/Dense here).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5

namespace cv {
struct Size {
  int width = 0, height = 0;
  bool operator==(const Size &o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size &o) const { return !(*this == o); }
};

class Mat {
public:
  int rows = 0, cols = 0;
  uint8_t *data = nullptr;
  Mat() = default;
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type), own_(std::make_shared<std::vector<uint8_t>>((size_t)r * c * elem())) { data = own_->data(); }
  Mat(int r, int c, int type, void *ext) : rows(r), cols(c), data((uint8_t *)ext), type_(type) {}
  int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
  bool isContinuous() const { return true; }
  bool empty() const { return data == nullptr; }
  Size size() const { return Size{cols, rows}; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    std::memcpy(m.data, data, (size_t)rows * cols * elem());
    return m;
  }

private:
  size_t elem() const { return type_ == CV_32FC1 ? 4 : (type_ == CV_8UC3 ? 3 : 1); }
  int type_ = CV_8UC1;
  std::shared_ptr<std::vector<uint8_t>> own_;
};
}  // namespace cv
