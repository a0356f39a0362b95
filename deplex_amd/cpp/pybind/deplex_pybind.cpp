// deplex_pybind.cpp — pybind11 module with the reference's binding surface (cpp/pybind/deplex_pybind.cpp:19-25,
// plane_extraction/plane_extraction.cpp:22-38, utils/utils.cpp:24-38) over the C++ mirror of this repository:
//   pybind.plane_extraction.Config(path), .PlaneExtractor(image_height, image_width, config=Config()).process(pcd_array)
//   pybind.utils.DepthImage(image_path): .height, .width, .transform_to_pcd(intrinsics), .reset(image_path)
// Additions beyond the reference (marked): default Config(), config fields, batched depth entry, plane records.
// PNG decoding is file IO outside the accelerated path; the reference vendors stb_image for it, this binding asks
// Pillow (verified identical on both reference frames).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../include/deplex/deplex.h"

namespace py = pybind11;
using deplex::PlaneExtractor;
using deplex::config::Config;

namespace {

using FloatArray = py::array_t<float, py::array::c_style | py::array::forcecast>;

// utils::DepthImage with the reference's path constructor (depth_image.cpp:27-53), decoded by the library (dpx_read_depth_png)
class PyDepthImage {
 public:
  explicit PyDepthImage(std::string const& image_path) { reset(image_path); }
  void reset(std::string const& image_path) { image_.reset(image_path); }
  int32_t height() const { return image_.getHeight(); }
  int32_t width() const { return image_.getWidth(); }
  // Eigen::MatrixXf toPointCloud(Eigen::Matrix3f const&) -> (N, 3) float32 (depth_image.cpp:55-78)
  py::array_t<float> transform_to_pcd(FloatArray intrinsics) const {
    if (intrinsics.size() != 9) throw std::runtime_error("intrinsics must be a 3x3 matrix");
    deplex::utils::Intrinsics K;
    std::memcpy(K.data(), intrinsics.data(), 9 * sizeof(float));
    std::vector<float> xyz;
    {
      py::gil_scoped_release nogil;
      xyz = image_.toPointCloud(K);
    }
    py::array_t<float> out({static_cast<py::ssize_t>(xyz.size() / 3), static_cast<py::ssize_t>(3)});
    std::memcpy(out.mutable_data(), xyz.data(), xyz.size() * sizeof(float));
    return out;
  }

 private:
  deplex::utils::DepthImage image_;
};

py::array_t<int32_t> process(PlaneExtractor& self, FloatArray pcd_array) {
  if (pcd_array.ndim() != 2 || pcd_array.shape(1) != 3) throw py::type_error("process(): pcd_array must have shape (N, 3)");
  const int64_t n = pcd_array.shape(0);
  py::array_t<int32_t> labels(static_cast<py::ssize_t>(n));
  {
    py::gil_scoped_release nogil;  // (the reference holds the GIL during process)
    if (dpx_process_pcd(self.handle(), n ? pcd_array.data() : nullptr, n, n ? labels.mutable_data() : nullptr) != DPX_OK)
      throw std::runtime_error(dpx_last_error());
  }
  return labels;
}

}  // namespace

PYBIND11_MODULE(_pybind, m) {
  m.doc() = "deplex_amd: pybind11 surface of prime-slam/deplex over the MI355X library";

  py::module m_pe = m.def_submodule("plane_extraction", "Module with Plane Extraction algorithm");
  py::class_<Config>(m_pe, "Config")
      .def(py::init<std::string>(), py::arg("path"))
      .def(py::init<>())  // addition: the C++ default constructor (config.h:51), used as PlaneExtractor's default
      .def_readwrite("patch_size", &Config::patch_size)  // additions: the public fields of config::Config
      .def_readwrite("histogram_bins_per_coord", &Config::histogram_bins_per_coord)
      .def_readwrite("min_cos_angle_merge", &Config::min_cos_angle_merge)
      .def_readwrite("max_merge_dist", &Config::max_merge_dist)
      .def_readwrite("min_region_growing_candidate_size", &Config::min_region_growing_candidate_size)
      .def_readwrite("min_region_growing_cells_activated", &Config::min_region_growing_cells_activated)
      .def_readwrite("min_region_planarity_score", &Config::min_region_planarity_score)
      .def_readwrite("ransac_refinement", &Config::ransac_refinement);
  py::class_<PlaneExtractor>(m_pe, "PlaneExtractor")
      .def(py::init([](int image_height, int image_width, Config config) {
             return new PlaneExtractor(image_height, image_width, config);
           }),
           py::arg("image_height"), py::arg("image_width"), py::arg("config") = Config())
      .def("process", &process, py::arg("pcd_array"))
      .def("planes",  // addition: fitted plane records of the last call as (normal[3], d, mean[3], mse, score, n_pts) rows
           [](PlaneExtractor& self) {
             std::vector<dpx_plane> pl = self.planes(0);
             py::array_t<float> out({static_cast<py::ssize_t>(pl.size()), static_cast<py::ssize_t>(10)});
             float* o = out.mutable_data();
             for (size_t i = 0; i < pl.size(); ++i) {
               const dpx_plane& q = pl[i];
               const float row[10] = {q.normal[0], q.normal[1], q.normal[2], q.d, q.mean[0], q.mean[1], q.mean[2], q.mse, q.score,
                                      static_cast<float>(q.n_pts)};
               std::memcpy(o + 10 * i, row, sizeof(row));
             }
             return out;
           });

  py::module m_utils = m.def_submodule("utils", "Plane Extraction utilities");
  py::class_<PyDepthImage>(m_utils, "DepthImage")
      .def(py::init<std::string>(), py::arg("image_path"))
      .def_property_readonly("height", &PyDepthImage::height)
      .def_property_readonly("width", &PyDepthImage::width)
      .def("transform_to_pcd", &PyDepthImage::transform_to_pcd, py::arg("intrinsics"))
      .def("reset", &PyDepthImage::reset, py::arg("image_path"));
}
