// deplex/plane_extractor.h — C++ mirror of deplex::PlaneExtractor (cpp/deplex/include/deplex/plane_extractor.h:28-56)
// over the C-ABI.  Move-only like the reference; construction and process() throw std::runtime_error with the
// reference's messages (plane_extractor.cpp:161-164, :188-194).
//
// The reference signature is  Eigen::VectorXi process(Eigen::MatrixX3f const&).  Eigen is not a dependency of this
// repository; the portable overloads take the organised cloud as a row-major float buffer, and the Eigen-typed
// overload is enabled automatically when <Eigen/Core> is on the include path.
#pragma once
#include <cstdint>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
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
  /** The same with caller-provided output (e.g. a slice of a page-locked buffer, dpx_host_alloc). */
  void processDepthBatch(uint16_t const* depth, int32_t batch, float fx, float fy, float cx, float cy, int32_t* labels) {
    if (dpx_process_depth_batch(h_, depth, batch, fx, fy, cx, cy, labels) != DPX_OK) throw std::runtime_error(dpx_last_error());
  }
  /** Cell-level output of the same computation (what the reference holds before toImageLabels, plane_extractor.cpp:455-470):
   *  the merged 1-based label of every cell, batch x nv x nh, and the batch's plane records packed frame after frame
   *  (n_planes[f] records for frame f).  80x fewer bytes cross PCIe than with the pixel map. */
  struct CellResult {
    std::vector<int32_t> cell_labels;
    std::vector<int32_t> n_planes;
    std::vector<dpx_plane> planes;
  };
  CellResult processDepthBatchCells(uint16_t const* depth, int32_t batch, float fx, float fy, float cx, float cy) {
    int32_t nh = 0, nv = 0;
    if (dpx_grid_dims(h_, &nh, &nv) != DPX_OK) throw std::runtime_error(dpx_last_error());
    CellResult r;
    r.cell_labels.resize(static_cast<size_t>(batch) * nh * nv);
    r.n_planes.resize(static_cast<size_t>(batch));
    int64_t total = 0;
    if (dpx_process_depth_batch_cells(h_, depth, batch, fx, fy, cx, cy, r.cell_labels.data(), r.n_planes.data(), nullptr, 0, &total) != DPX_OK)
      throw std::runtime_error(dpx_last_error());
    r.planes.resize(static_cast<size_t>(total));  // (sized from the count: one copy out of the handle's staging)
    if (dpx_get_packed_planes(h_, r.planes.data(), total, nullptr) != DPX_OK) throw std::runtime_error(dpx_last_error());
    r.planes.resize(static_cast<size_t>(total));
    return r;
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
  int32_t height() const { return height_; }
  int32_t width() const { return width_; }

 private:
  dpx_handle h_ = nullptr;
  int32_t height_, width_;
};

/** A batch of HOST frames over several GPUs of one node (not in the reference, which is single-threaded): frames are
 *  independent (an extractor holds only per-call scratch, plane_extractor.cpp:281,428), so the batch is cut into contiguous
 *  ranges frames[g*B/G, (g+1)*B/G), one per entry of `devices` (a device may be listed more than once), each processed by
 *  its own host thread through its own extractor in chunks of max_batch frames; the label maps land in the caller's
 *  (ideally page-locked) output at the range's offset.  No exchange between devices. */
class ShardedExtractor {
 public:
  /** `slots` batches of `max_batch` frames are in flight per device (dpx_pipeline with host-buffer submissions: every slot's
   *  stream carries copy in, kernels, copy out).  Set GPU_MAX_HW_QUEUES >= slots per device before HIP initialises. */
  ShardedExtractor(int32_t image_height, int32_t image_width, std::vector<int32_t> const& devices, config::Config config = config::Config(),
                   int32_t max_batch = 64, int32_t slots = 3)
      : height_(image_height), width_(image_width), max_batch_(max_batch) {
    if (devices.empty()) throw std::runtime_error("ShardedExtractor needs at least one device");
    const dpx_config c = config.to_c();
    for (int32_t d : devices) {
      shards_.emplace_back(image_height, image_width, config, max_batch, d);
      dpx_pipeline p = nullptr;
      if (dpx_pipeline_create(&c, image_height, image_width, max_batch, d, slots, 1, &p) != DPX_OK) {
        for (dpx_pipeline q : pipes_) dpx_pipeline_destroy(q);
        throw std::runtime_error(dpx_last_error());
      }
      pipes_.push_back(p);
    }
  }
  ~ShardedExtractor() {
    for (dpx_pipeline p : pipes_) dpx_pipeline_destroy(p);
  }
  ShardedExtractor(ShardedExtractor const&) = delete;
  ShardedExtractor& operator=(ShardedExtractor const&) = delete;
  static std::pair<int64_t, int64_t> shardRange(int64_t n_frames, int64_t rank, int64_t world) {
    return {n_frames * rank / world, n_frames * (rank + 1) / world};
  }
  /** depth: batch x H x W uint16 in host memory (page-locked: the copies run asynchronously); labels: batch x H x W int32 */
  void processDepthBatch(uint16_t const* depth, int64_t batch, float fx, float fy, float cx, float cy, int32_t* labels) {
    const size_t n = static_cast<size_t>(height_) * width_;
    const int64_t world = static_cast<int64_t>(shards_.size());
    std::vector<std::thread> threads;
    std::vector<std::string> errors(shards_.size());
    for (int64_t g = 0; g < world; ++g) {
      const auto range = shardRange(batch, g, world);
      if (range.second <= range.first) continue;
      threads.emplace_back([&, g, range] {
        try {
          dpx_pipeline p = pipes_[static_cast<size_t>(g)];
          int64_t s = range.first;
          for (; s + max_batch_ <= range.second; s += max_batch_)  // whole chunks stream through the shard's pipeline
            if (dpx_pipeline_submit_depth_host(p, depth + static_cast<size_t>(s) * n, max_batch_, fx, fy, cx, cy, labels + static_cast<size_t>(s) * n) != DPX_OK)
              throw std::runtime_error(dpx_last_error());
          if (s < range.second)  // the remainder: one blocking call
            shards_[static_cast<size_t>(g)].processDepthBatch(depth + static_cast<size_t>(s) * n, static_cast<int32_t>(range.second - s), fx, fy, cx, cy,
                                                             labels + static_cast<size_t>(s) * n);
          if (dpx_pipeline_sync(p) != DPX_OK) throw std::runtime_error(dpx_last_error());
        } catch (std::exception const& e) {
          errors[static_cast<size_t>(g)] = e.what();
        }
      });
    }
    for (auto& t : threads) t.join();
    for (auto const& e : errors)
      if (!e.empty()) throw std::runtime_error(e);
  }
  size_t shards() const { return shards_.size(); }

 private:
  int32_t height_, width_, max_batch_;
  std::vector<PlaneExtractor> shards_;
  std::vector<dpx_pipeline> pipes_;
};

/** A stream of HOST batches through one GPU (not in the reference): `slots` batches are in flight, each on its own
 *  stream carrying copy in, kernels, copy out (dpx_pipeline_submit_depth_host / _cells_host), so the copies of one batch
 *  overlap the kernels of the others.  Buffers handed to submit() must stay valid and untouched until sync(); page-locked
 *  memory (dpx_host_alloc) makes the copies asynchronous.  Every submission has `batch` frames. */
class StreamExtractor {
 public:
  StreamExtractor(int32_t image_height, int32_t image_width, int32_t batch, config::Config config = config::Config(), int32_t slots = 3,
                  int32_t device = 0) {
    const dpx_config c = config.to_c();
    if (dpx_pipeline_create(&c, image_height, image_width, batch, device, slots, 1, &p_) != DPX_OK) throw std::runtime_error(dpx_last_error());
    batch_ = batch;
  }
  ~StreamExtractor() { dpx_pipeline_destroy(p_); }
  StreamExtractor(StreamExtractor const&) = delete;
  StreamExtractor& operator=(StreamExtractor const&) = delete;
  /** labels: batch x H x W int32 (the reference's label maps) */
  void submit(uint16_t const* depth, float fx, float fy, float cx, float cy, int32_t* labels) {
    if (dpx_pipeline_submit_depth_host(p_, depth, batch_, fx, fy, cx, cy, labels) != DPX_OK) throw std::runtime_error(dpx_last_error());
  }
  /** cell_labels: batch x nv x nh int32 (dpx_expand_cell_labels turns them into label maps on the host) */
  void submitCells(uint16_t const* depth, float fx, float fy, float cx, float cy, int32_t* cell_labels) {
    if (dpx_pipeline_submit_depth_cells_host(p_, depth, batch_, fx, fy, cx, cy, cell_labels) != DPX_OK) throw std::runtime_error(dpx_last_error());
  }
  void sync() {
    if (dpx_pipeline_sync(p_) != DPX_OK) throw std::runtime_error(dpx_last_error());
  }

 private:
  dpx_pipeline p_ = nullptr;
  int32_t batch_ = 0;
};
}  // namespace deplex
