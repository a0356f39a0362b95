// C++ usage of the mirror API, shaped like the reference's examples/process_cloud.cpp:24-36 and its gtest cases
// (cpp/tests/test_plane_extractor.cpp:27-88).  Input: a raw little-endian uint16 depth dump (H*W values) + .K file.
// Build:  g++ -std=c++17 -I deplex_amd/cpp/include process_cloud.cpp -L deplex_amd/lib -ldeplex_amd -Wl,-rpath,...
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>

#include <deplex/deplex.h>

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s depth.u16 height width intrinsics.K [config.ini]\n", argv[0]);
    return 2;
  }
  const int h = std::atoi(argv[2]), w = std::atoi(argv[3]);
  std::vector<uint16_t> raw(static_cast<size_t>(h) * w);
  std::ifstream f(argv[1], std::ios::binary);
  if (!f.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size() * 2))) {
    std::fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  try {
    deplex::utils::DepthImage image(raw.data(), h, w);
    const auto K = deplex::utils::readIntrinsics(argv[4]);
    deplex::config::Config config = argc > 5 ? deplex::config::Config(std::string(argv[5])) : deplex::config::Config();
    deplex::PlaneExtractor algorithm(image.getHeight(), image.getWidth(), config);
    const std::vector<float> points = image.toPointCloud(K);
    std::vector<int32_t> labels = algorithm.process(points);
    const int NUMBER_OF_RUNS = 10;
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < NUMBER_OF_RUNS; ++i) labels = algorithm.process(points);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / NUMBER_OF_RUNS;
    std::cout << "planes " << *std::max_element(labels.begin(), labels.end()) << " labels " << labels.size() << " mean_us " << us
              << " fps " << 1e6 / us << " fitted " << algorithm.planes().size() << "\n";
    // batched depth entry points: pixel labels, the cell-level form expanded on the host, and two shards on one device
    {
      const int B = 5;
      std::vector<uint16_t> frames(static_cast<size_t>(B) * h * w);
      for (int k = 0; k < B; ++k) std::copy(raw.begin(), raw.end(), frames.begin() + static_cast<size_t>(k) * h * w);
      deplex::PlaneExtractor batch(h, w, config, B);
      const std::vector<int32_t> pix = batch.processDepthBatch(frames.data(), B, K[0], K[4], K[2], K[5]);
      const auto cells = batch.processDepthBatchCells(frames.data(), B, K[0], K[4], K[2], K[5]);
      std::vector<int32_t> expanded(pix.size());
      if (dpx_expand_cell_labels(h, w, config.patch_size, cells.cell_labels.data(), B, expanded.data()) != DPX_OK) throw std::runtime_error(dpx_last_error());
      int32_t* pinned = static_cast<int32_t*>(dpx_host_alloc(pix.size() * sizeof(int32_t)));
      if (!pinned) throw std::runtime_error(dpx_last_error());
      deplex::ShardedExtractor sharded(h, w, {0, 0}, config, 2);
      sharded.processDepthBatch(frames.data(), B, K[0], K[4], K[2], K[5], pinned);
      bool same = std::equal(pix.begin(), pix.end(), pinned) && expanded == pix && std::equal(labels.begin(), labels.end(), pix.begin());
      // a stream of host batches: three in flight, pixel maps and cell labels
      {
        deplex::StreamExtractor stream(h, w, B, config);
        std::vector<int32_t> out_a(pix.size(), -1), out_b(pix.size(), -1), out_c(cells.cell_labels.size(), -1);
        std::fill(pinned, pinned + pix.size(), -1);
        stream.submit(frames.data(), K[0], K[4], K[2], K[5], pinned);
        stream.submit(frames.data(), K[0], K[4], K[2], K[5], out_a.data());
        stream.submitCells(frames.data(), K[0], K[4], K[2], K[5], out_c.data());
        stream.submit(frames.data(), K[0], K[4], K[2], K[5], out_b.data());
        stream.sync();
        same = same && std::equal(pix.begin(), pix.end(), pinned) && out_a == pix && out_b == pix && out_c == cells.cell_labels;
      }
      dpx_host_free(pinned);
      std::cout << "batch " << B << " cells_equal_pixels_equal_sharded_equal_streamed " << (same ? 1 : 0) << " planes_frame0 " << cells.n_planes[0] << " total "
                << cells.planes.size() << "\n";
      if (!same) return 1;
    }
    // the reference's throwing cases
    try {
      deplex::config::Config bad;
      bad.patch_size = 0;
      deplex::PlaneExtractor x(h, w, bad);
      std::cout << "ERROR: patch_size 0 did not throw\n";
      return 1;
    } catch (std::runtime_error const& e) {
      std::cout << "throws: " << e.what() << "\n";
    }
    try {
      deplex::PlaneExtractor x(h / 2, w / 2);
      x.process(points);
      std::cout << "ERROR: wrong shape did not throw\n";
      return 1;
    } catch (std::runtime_error const& e) {
      std::cout << "throws: " << e.what() << "\n";
    }
  } catch (std::exception const& e) {
    std::cerr << "failed: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
