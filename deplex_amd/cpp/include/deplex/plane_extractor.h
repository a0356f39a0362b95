// deplex/plane_extractor.h — C++ mirror of deplex::PlaneExtractor (cpp/deplex/include/deplex/plane_extractor.h:28-56)
// over the C-ABI.  Move-only like the reference; construction and process() throw std::runtime_error with the
// reference's messages (plane_extractor.cpp:161-164, :188-194).
//
// The reference signature is  Eigen::VectorXi process(Eigen::MatrixX3f const&).  Eigen is not a dependency of this
// repository; the portable overloads take the organised cloud as a row-major float buffer, and the Eigen-typed
// overload is enabled automatically when <Eigen/Core> is on the include path.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

#include "config.h"

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define DEPLEX_AMD_HAS_EIGEN 1
#endif

namespace deplex {
class PlaneExtractor {
 public:
  PlaneExtractor(int32_t image_height, int32_t image_width, config::Config config = config::Config(), int32_t max_batch = 1,
                 int32_t device = 0)
      : height_(image_height), width_(image_width) {
    const dpx_config c = config.to_c();
    if (dpx_create(&c, image_height, image_width, max_batch, device, &h_) != DPX_OK) throw std::runtime_error(dpx_last_error());
  }
  ~PlaneExtractor() { dpx_destroy(h_); }
  PlaneExtractor(PlaneExtractor&& op) noexcept : h_(op.h_), height_(op.height_), width_(op.width_) { op.h_ = nullptr; }
  PlaneExtractor& operator=(PlaneExtractor&& op) noexcept {
    if (this != &op) {
      dpx_destroy(h_);
      h_ = op.h_;
      height_ = op.height_;
      width_ = op.width_;
      op.h_ = nullptr;
    }
    return *this;
  }
  PlaneExtractor(PlaneExtractor const&) = delete;
  PlaneExtractor& operator=(PlaneExtractor const&) = delete;

  /** Organised cloud, n_points x 3 row-major float (row index = image_row * W + image_col) -> one label per point. */
  std::vector<int32_t> process(float const* xyz_row_major, int64_t n_points) {
    std::vector<int32_t> labels(static_cast<size_t>(n_points > 0 ? n_points : 0), 0);
    if (dpx_process_pcd(h_, xyz_row_major, n_points, labels.data()) != DPX_OK) throw std::runtime_error(dpx_last_error());
    return labels;
  }
  std::vector<int32_t> process(std::vector<float> const& xyz_row_major) {
    return process(xyz_row_major.data(), static_cast<int64_t>(xyz_row_major.size() / 3));
  }
#ifdef DEPLEX_AMD_HAS_EIGEN
  Eigen::VectorXi process(Eigen::MatrixX3f const& pcd_array) {
    Eigen::Matrix<float, Eigen::Dynamic, 3, Eigen::RowMajor> xyz = pcd_array;  // same re-layout as cell_grid.cpp:21
    Eigen::VectorXi labels = Eigen::VectorXi::Zero(pcd_array.rows());
    if (dpx_process_pcd(h_, xyz.data(), pcd_array.rows(), labels.data()) != DPX_OK) throw std::runtime_error(dpx_last_error());
    return labels;
  }
#endif
  /** Fused DepthImage::toPointCloud + process for `batch` uint16 frames (not in the reference). */
  std::vector<int32_t> processDepthBatch(uint16_t const* depth, int32_t batch, float fx, float fy, float cx, float cy) {
    std::vector<int32_t> labels(static_cast<size_t>(batch) * height_ * width_);
    if (dpx_process_depth_batch(h_, depth, batch, fx, fy, cx, cy, labels.data()) != DPX_OK) throw std::runtime_error(dpx_last_error());
    return labels;
  }
  /** Fitted planes of frame `frame` of the last call (the reference keeps them private, plane_extractor.cpp:222). */
  std::vector<dpx_plane> planes(int32_t frame = 0) {
    int32_t n = 0;
    if (dpx_get_planes(h_, frame, nullptr, 0, &n) != DPX_OK) throw std::runtime_error(dpx_last_error());
    std::vector<dpx_plane> out(static_cast<size_t>(n));
    if (n > 0 && dpx_get_planes(h_, frame, out.data(), n, &n) != DPX_OK) throw std::runtime_error(dpx_last_error());
    return out;
  }
  dpx_handle handle() const { return h_; }

 private:
  dpx_handle h_ = nullptr;
  int32_t height_, width_;
};
}  // namespace deplex
