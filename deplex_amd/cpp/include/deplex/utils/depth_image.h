// deplex/utils/depth_image.h — C++ mirror of deplex::utils::DepthImage (cpp/deplex/include/deplex/utils/depth_image.h:24-56).
// The back-projection arithmetic (utils/depth_image.cpp:55-78) runs in the HIP kernel behind dpx_backproject.
// PNG decoding is file IO and outside the accelerated path (the reference vendors stb_image for it): the path
// constructor goes through the library's reader (dpx_read_depth_png, same conversion rules as the reference's
// stbi_load_16(..., STBI_grey) call); a decoded 16-bit buffer can be passed directly as well.  readIntrinsics mirrors
// utils/eigen_io.cpp:41-55.
#pragma once
#include <array>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../../../include/deplex_amd.h"

namespace deplex {
namespace utils {
/** 3x3 intrinsics, row-major (K(i,j) = k[3*i+j]). */
using Intrinsics = std::array<float, 9>;

inline Intrinsics readIntrinsics(std::string const& intrinsics_path) {
  std::ifstream in(intrinsics_path);
  if (!in.is_open()) throw std::runtime_error("Error: Couldn't open intrinsics file " + intrinsics_path);
  Intrinsics K{};
  for (int i = 0; i < 9; ++i) in >> K[i];
  return K;
}

class DepthImage {
 public:
  DepthImage() : width_(0), height_(0) {}
  /** The reference's constructor (utils/depth_image.cpp:27-41): reads a PNG as one 16-bit grey channel. */
  explicit DepthImage(std::string const& image_path) : width_(0), height_(0) { reset(image_path); }
  DepthImage(uint16_t const* data, int32_t height, int32_t width) : image_(data, data + static_cast<size_t>(height) * width), width_(width), height_(height) {}
  /** utils/depth_image.cpp:43-50; throws std::runtime_error("Error: Couldn't read image <path>") like the reference. */
  void reset(std::string const& image_path) {
    uint16_t* data = nullptr;
    int32_t h = 0, w = 0;
    if (dpx_read_depth_png(image_path.c_str(), &h, &w, &data) != DPX_OK) throw std::runtime_error(dpx_last_error());
    image_.assign(data, data + static_cast<size_t>(h) * w);
    dpx_free(data);
    width_ = w;
    height_ = h;
  }
  int32_t getWidth() const { return width_; }
  int32_t getHeight() const { return height_; }
  uint16_t const* data() const { return image_.data(); }
  /** N x 3 row-major float cloud; z in raw depth units (no scaling), like the reference. */
  std::vector<float> toPointCloud(Intrinsics const& K, int32_t device = 0) const {
    std::vector<float> xyz(static_cast<size_t>(3) * width_ * height_);
    if (dpx_backproject(image_.data(), height_, width_, K[0], K[4], K[2], K[5], xyz.data(), device) != DPX_OK)
      throw std::runtime_error(dpx_last_error());
    return xyz;
  }

 private:
  std::vector<uint16_t> image_;
  int32_t width_, height_;
};
}  // namespace utils
}  // namespace deplex
