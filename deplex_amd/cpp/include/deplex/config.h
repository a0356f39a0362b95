// deplex/config.h — C++ mirror of the reference's deplex::config::Config over the C-ABI.
// Same public fields, defaults and constructors as cpp/deplex/include/deplex/config.h:29-82 of prime-slam/deplex;
// Config(std::string const&) parses the .ini through dpx_config_from_ini (config.cpp:28-80) and throws
// std::runtime_error("Couldn't open ini file: " + path) like the reference.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "../../../../include/deplex_amd.h"

namespace deplex {
namespace config {
struct Config {
 public:
  Config() = default;
  /** The reference's map constructor is an empty stub (config.cpp:25-26); kept for source compatibility. */
  explicit Config(std::unordered_map<std::string, std::string> const&) {}
  explicit Config(std::string const& config_path) {
    dpx_config c;
    if (dpx_config_from_ini(config_path.c_str(), &c) != DPX_OK) throw std::runtime_error(dpx_last_error());
    from_c(c);
  }

  int32_t patch_size = 10;
  int32_t histogram_bins_per_coord = 20;
  float min_cos_angle_merge = 0.90f;
  float max_merge_dist = 500.f;
  int32_t min_region_growing_candidate_size = 5;
  int32_t min_region_growing_cells_activated = 4;
  float min_region_planarity_score = 0.55f;
  float depth_sigma_coeff = 1.425e-6f;
  float depth_sigma_margin = 10.f;
  int32_t min_pts_per_cell = 3;
  float depth_discontinuity_threshold = 160.f;
  int32_t max_number_depth_discontinuity = 1;
  bool ransac_refinement = false;
  int32_t ransac_max_iterations = 1000;
  float ransac_threshold = 1.f;
  float ransac_inliers_ratio = 0.9f;

  dpx_config to_c() const {
    dpx_config c;
    c.patch_size = patch_size;
    c.histogram_bins_per_coord = histogram_bins_per_coord;
    c.min_cos_angle_merge = min_cos_angle_merge;
    c.max_merge_dist = max_merge_dist;
    c.min_region_growing_candidate_size = min_region_growing_candidate_size;
    c.min_region_growing_cells_activated = min_region_growing_cells_activated;
    c.min_region_planarity_score = min_region_planarity_score;
    c.depth_sigma_coeff = depth_sigma_coeff;
    c.depth_sigma_margin = depth_sigma_margin;
    c.min_pts_per_cell = min_pts_per_cell;
    c.depth_discontinuity_threshold = depth_discontinuity_threshold;
    c.max_number_depth_discontinuity = max_number_depth_discontinuity;
    c.ransac_refinement = ransac_refinement ? 1 : 0;
    c.ransac_max_iterations = ransac_max_iterations;
    c.ransac_threshold = ransac_threshold;
    c.ransac_inliers_ratio = ransac_inliers_ratio;
    return c;
  }

 private:
  void from_c(dpx_config const& c) {
    patch_size = c.patch_size;
    histogram_bins_per_coord = c.histogram_bins_per_coord;
    min_cos_angle_merge = c.min_cos_angle_merge;
    max_merge_dist = c.max_merge_dist;
    min_region_growing_candidate_size = c.min_region_growing_candidate_size;
    min_region_growing_cells_activated = c.min_region_growing_cells_activated;
    min_region_planarity_score = c.min_region_planarity_score;
    depth_sigma_coeff = c.depth_sigma_coeff;
    depth_sigma_margin = c.depth_sigma_margin;
    min_pts_per_cell = c.min_pts_per_cell;
    depth_discontinuity_threshold = c.depth_discontinuity_threshold;
    max_number_depth_discontinuity = c.max_number_depth_discontinuity;
    ransac_refinement = c.ransac_refinement != 0;
    ransac_max_iterations = c.ransac_max_iterations;
    ransac_threshold = c.ransac_threshold;
    ransac_inliers_ratio = c.ransac_inliers_ratio;
  }
};
}  // namespace config
}  // namespace deplex
