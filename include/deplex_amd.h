/* deplex_amd.h — C-ABI of the MI355X-native depth-plane extractor.
 *
 * Drop-in boundary for prime-slam/deplex's hot path PlaneExtractor::process().  Every entry point
 * names the reference interface it replaces (paths relative to the reference checkout).  Plain
 * pointers and sizes only; no C++/torch types.  The library is libdeplex_amd.so, built by
 * __graft_entry__.build() with hipcc --offload-arch=gfx950.  All compute runs in hand-written HIP
 * kernels; there is no CPU fallback — calls fail with DPX_ERR_HIP / DPX_ERR_NO_DEVICE without a GPU.
 *
 * Threading: a handle is not re-entrant (like a reference PlaneExtractor instance, whose labels_map_
 * is mutable scratch, plane_extractor.cpp:76); distinct handles are independent.
 */
#ifndef DEPLEX_AMD_H
#define DEPLEX_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* POD mirror of deplex::config::Config, same field order and defaults
 * (cpp/deplex/include/deplex/config.h:51-81). */
typedef struct dpx_config {
  int32_t patch_size;                         /* patchSize                      10 */
  int32_t histogram_bins_per_coord;           /* histogramBinsPerCoord          20 */
  float min_cos_angle_merge;                  /* minCosAngleForMerge          0.90 */
  float max_merge_dist;                       /* maxMergeDist                  500 */
  int32_t min_region_growing_candidate_size;  /* minRegionGrowingCandidateSize   5 */
  int32_t min_region_growing_cells_activated; /* minRegionGrowingCellsActivated  4 */
  float min_region_planarity_score;           /* minRegionPlanarityScore      0.55 */
  float depth_sigma_coeff;                    /* depthSigmaCoeff          1.425e-6 */
  float depth_sigma_margin;                   /* depthSigmaMargin               10 */
  int32_t min_pts_per_cell;                   /* minPtsPerCell                   3 */
  float depth_discontinuity_threshold;        /* depthDiscontinuityThreshold   160 */
  int32_t max_number_depth_discontinuity;     /* maxNumberDepthDiscontinuity     1 */
  int32_t ransac_refinement;                  /* ransacRefinement                0 */
  int32_t ransac_max_iterations;              /* ransacMaxIterations          1000 */
  float ransac_threshold;                     /* ransacThreshold                 1 */
  float ransac_inliers_ratio;                 /* ransacInliersRatio            0.9 */
} dpx_config;

/* Per-cell record the plane-fit kernel writes (64 bytes, one per grid cell, cell = row*nh + col).
 * Fields follow CellSegmentStat / CellSegment (cell_segment_stat.h:62-69, cell_segment.h). */
typedef struct dpx_cell {
  float sum[3];    /* coord_sum_ */
  float pp[6];     /* variance_: xx xy xz yy yz zz */
  float normal[3]; /* normal_ */
  float d;         /* d_ */
  float mse;       /* mse_ (FLT_MAX when the cell is not valid) */
  float tol;       /* merge_tolerance_ */
  uint32_t info;   /* bit0 valid, bit1 planar, bit2 eigensolver used QL, bits 8..31 histogram bin + 1 (0 = none) */
} dpx_cell;

/* Per-plane record (one per grown segment, in creation order; state after findMergedLabels,
 * plane_extractor.cpp:394-426).  The reference keeps these private; exported here because the
 * acceptance criteria grade fitted plane parameters. */
typedef struct dpx_plane {
  float sum[3];
  float pp[6];
  float mean[3];
  float normal[3];
  float d, mse, score;
  int32_t n_pts;       /* nr_pts_ */
  int32_t n_cells;     /* cells activated by the grow that created the segment */
  int32_t merge_label; /* plane_merge_labels[i], 0-based; label in the image = merge_label + 1 */
} dpx_plane;

typedef struct dpx_frame_stats {
  int32_t planar_cells;
  int32_t grows;
  int32_t activations;
  int32_t segments;
  int32_t merges;
  int32_t overflow; /* non-zero if an internal capacity was exceeded (results for the frame invalid) */
} dpx_frame_stats;

typedef struct dpx_handle_s* dpx_handle;
typedef struct dpx_pipeline_s* dpx_pipeline;

enum {
  DPX_OK = 0,
  DPX_ERR_PATCH_ZERO = -1, /* reference throws at construction, plane_extractor.cpp:161-164 */
  DPX_ERR_SHAPE = -2,      /* reference throws in process(), plane_extractor.cpp:188-194 */
  DPX_ERR_UNDEFINED = -3,  /* configuration the reference leaves undefined (negative patchSize, minPtsPerCell==0, ...) */
  DPX_ERR_HIP = -4,
  DPX_ERR_ARG = -5,
  DPX_ERR_CAPACITY = -6,
  DPX_ERR_NO_DEVICE = -7,
  DPX_ERR_IO = -8
};

/* Kernel indices for dpx_get_kernel_ms(). */
enum { DPX_K_PLANEFIT = 0, DPX_K_CELLGRAPH = 1, DPX_K_REGIONS = 2, DPX_K_LABELS = 3, DPX_K_COUNT = 4 };

/* Config::Config() defaults — config.h:51-81. */
void dpx_default_config(dpx_config* cfg);

/* Config::Config(std::string const& config_path) — config.cpp:28-80 (same key names, no trimming, '#' comments,
 * std::stoi/std::stof prefix parsing).  DPX_ERR_IO with message "Couldn't open ini file: <path>" if unreadable,
 * DPX_ERR_ARG if a value does not parse (the reference lets std::invalid_argument escape). */
int dpx_config_from_ini(const char* path, dpx_config* cfg);

/* PlaneExtractor::PlaneExtractor(int32_t image_height, int32_t image_width, config::Config) —
 * plane_extractor.h:39, plane_extractor.cpp:153-176,182-183.  Also allocates device scratch for up to
 * max_batch frames per call on HIP device `device`. */
int dpx_create(const dpx_config* cfg, int32_t image_height, int32_t image_width, int32_t max_batch, int32_t device,
               dpx_handle* out);

/* PlaneExtractor::~PlaneExtractor — plane_extractor.cpp:178. */
void dpx_destroy(dpx_handle h);

/* Message of the last failing call on this thread (the reference's exception text for its throwing cases). */
const char* dpx_last_error(void);

/* Eigen::VectorXi PlaneExtractor::process(Eigen::MatrixX3f const& pcd_array) — plane_extractor.h:48,
 * plane_extractor.cpp:185-283.  xyz: n_points x 3 row-major float32 organised cloud in HOST memory
 * (row index = image_row*W + image_col); labels: n_points int32 in HOST memory, 0 = non-planar.
 * Returns DPX_ERR_SHAPE (with the reference's message) when n_points != H*W. */
int dpx_process_pcd(dpx_handle h, const float* xyz, int64_t n_points, int32_t* labels);

/* Batched form of the same call: `batch` clouds back to back. */
int dpx_process_pcd_batch(dpx_handle h, const float* xyz, int32_t batch, int32_t* labels);

/* DepthImage::toPointCloud(K) followed by PlaneExtractor::process() for `batch` depth frames, fused on the
 * device (utils/depth_image.cpp:55-78 + plane_extractor.cpp:185-283).  depth: batch x H x W uint16 (HOST),
 * labels: batch x H x W int32 (HOST). */
int dpx_process_depth_batch(dpx_handle h, const uint16_t* depth, int32_t batch, float fx, float fy, float cx,
                            float cy, int32_t* labels);

/* The same call with CELL-LEVEL output: what PlaneExtractor::process() knows before toImageLabels
 * (plane_extractor.cpp:455-470) blows it up to one int32 per pixel.  cell_labels: batch x nv x nh int32 (HOST), the
 * merged 1-based label of every cell (0 = none) — exactly what the pixel map holds for all pixels of the cell;
 * n_planes (batch int32, may be NULL) and planes (HOST, may be NULL): the frames' plane records packed back to
 * back, frame f's first record at the sum of n_planes[0..f); planes_cap = records the buffer can hold; *planes_total
 * (may be NULL) receives the number of records of the batch.  DPX_ERR_CAPACITY (records up to planes_cap copied) if
 * the buffer is too small.  Device-to-host traffic per 640x480 frame: 12 KB + ~3 KB instead of 1.2 MB — the
 * host-pointer entry points are PCIe-bound, so this is ~3x the end-to-end frame rate.
 * Not defined with ransac_refinement (which edits single pixels): DPX_ERR_ARG. */
int dpx_process_depth_batch_cells(dpx_handle h, const uint16_t* depth, int32_t batch, float fx, float fy, float cx,
                                  float cy, int32_t* cell_labels, int32_t* n_planes, dpx_plane* planes,
                                  int64_t planes_cap, int64_t* planes_total);
/* The packed plane records of the LAST dpx_process_depth_batch_cells call on this handle that asked for counts (n_planes or
 * planes_total non-NULL), from the handle's page-locked staging — for callers that size their buffer from *planes_total
 * first (pass planes = NULL to the processing call, then call this once).  Same capacity rule. */
int dpx_get_packed_planes(dpx_handle h, dpx_plane* planes, int64_t planes_cap, int64_t* planes_total);
/* Impl::toImageLabels (plane_extractor.cpp:455-470) on the host, for callers of the cell-level form that want a pixel
 * map after all: labels[f][row][col] = cell_labels[f][row / P][col / P] (0 outside the nv*P x nh*P grid; P = the
 * stored patch size min(patch_size, H, W), nh = W / patch_size, nv = H / patch_size).  Plain C loop, no device. */
int dpx_expand_cell_labels(int32_t image_height, int32_t image_width, int32_t patch_size, const int32_t* cell_labels,
                           int32_t batch, int32_t* labels);

/* Page-locked host memory for the host-pointer entry points (hipHostMalloc): copies from / to such buffers go
 * straight over PCIe instead of through the runtime's staging buffer, and one buffer can be the target of several
 * devices' copies at once (disjoint slices: deplex_amd.ShardedExtractor).  NULL on failure. */
void* dpx_host_alloc(size_t bytes);
void dpx_host_free(void* p);

/* Device-pointer, stream-ordered forms (no copies, no synchronisation): d_* are device pointers on the
 * handle's device, hip_stream is a hipStream_t (NULL = default stream).  Used by pipelines and bench.py. */
int dpx_process_depth_batch_device(dpx_handle h, const void* d_depth, int32_t batch, float fx, float fy, float cx,
                                   float cy, void* d_labels, void* hip_stream);
int dpx_process_pcd_batch_device(dpx_handle h, const void* d_xyz, int32_t batch, void* d_labels, void* hip_stream);

/* Streamed processing (the reference's examples/process_sequence.cpp:30-53 loop, pipelined).  `slots` groups of
 * `group` (1..32) submissions are in flight, each slot with its own scratch and stream.  A submission is recorded and
 * the whole group is launched when it is full (or at dpx_pipeline_sync): ONE plane-fit launch over all its frames (each
 * submission keeps its own input buffer and intrinsics), then region growing and label expansion, also one launch
 * sequence per group (the GPU executes only a handful of kernels concurrently and a launch per small batch leaves the
 * chip under-filled at its start and end; group = 1 restores launch-per-submission).  Submissions must all have the
 * batch size given at creation.
 * submit() is asynchronous on device pointers (the first submission with an intrinsics tuple the process has not seen
 * for this image shape blocks the host for the ~0.2 ms of its exact-division check, on a stream of its own — it does not
 * wait for the slot's group in flight; dpx_prepare_intrinsics() moves that ahead of time); results of a submission are
 * complete after dpx_pipeline_sync() (which
 * also flushes a partially filled group).  The caller must not reuse d_depth / d_labels of a submission before that.
 * dpx_pipeline_slot() exposes the per-slot handle (frames of a slot = its group's submissions back to back). */
int dpx_pipeline_create(const dpx_config* cfg, int32_t image_height, int32_t image_width, int32_t batch, int32_t device,
                        int32_t slots, int32_t group, dpx_pipeline* out);
void dpx_pipeline_destroy(dpx_pipeline p);
int dpx_pipeline_submit_depth(dpx_pipeline p, const void* d_depth, int32_t batch, float fx, float fy, float cx, float cy,
                              void* d_labels);
/* A submission with CELL-LEVEL output (see dpx_process_depth_batch_cells): d_cell_labels receives batch x nv x nh int32
 * (device memory), the merged 1-based label of every cell; the label expansion launch — a quarter of the pipeline's GPU
 * time and two thirds of its HBM writes — is left out for the group.  A group holds one output form: switching forms
 * closes the open group first.  The plane records of a slot's last group are read with dpx_get_planes on
 * dpx_pipeline_slot() after dpx_pipeline_sync().  DPX_ERR_ARG with ransac_refinement. */
int dpx_pipeline_submit_depth_cells(dpx_pipeline p, const void* d_depth, int32_t batch, float fx, float fy, float cx, float cy,
                                    void* d_cell_labels);
/* HOST-buffer submissions: depth frames in host memory in, pixel label maps (batch x H x W int32) or cell labels
 * (batch x nv x nh int32) in host memory out.  The frames are copied to a staging buffer of the slot on the slot's
 * stream at submission, the results back after the group's kernels, so the copies of one slot overlap the kernels of
 * the others (three slots keep the PCIe link busy: 90 k frames/s at 640x480 with cell labels, 38 k with pixel maps,
 * DESIGN.md section 6).  Both buffers must stay valid and untouched until dpx_pipeline_sync(); page-locked memory
 * (dpx_host_alloc) makes the copies asynchronous, pageable memory works but blocks in the call.  May be mixed with
 * device-buffer submissions. */
int dpx_pipeline_submit_depth_host(dpx_pipeline p, const uint16_t* h_depth, int32_t batch, float fx, float fy, float cx, float cy,
                                   int32_t* h_labels);
int dpx_pipeline_submit_depth_cells_host(dpx_pipeline p, const uint16_t* h_depth, int32_t batch, float fx, float fy, float cx,
                                         float cy, int32_t* h_cell_labels);
/* n submissions in one call (same batch size and intrinsics, one input and one output buffer each): what a caller that
 * already holds several batches does instead of n calls — the order and the grouping are those of n single calls. */
int dpx_pipeline_submit_depth_many(dpx_pipeline p, int32_t n, const void* const* d_depth, int32_t batch, float fx, float fy,
                                   float cx, float cy, void* const* d_labels);
/* The same for organised clouds (float32 [batch][H*W][3], the input of PlaneExtractor::process, plane_extractor.h:48).
 * A group holds one input form; switching forms closes the open group first. */
int dpx_pipeline_submit_pcd(dpx_pipeline p, const void* d_xyz, int32_t batch, void* d_labels);
int dpx_pipeline_sync(dpx_pipeline p);
/* Launches the submissions recorded so far without waiting for them (a partially filled group is launched as it is and
 * the next submission starts a new group on the next slot).  For callers that want the first results of a stream early,
 * or that want the groups in flight to be at different stages (docs/EXPERIMENTS.md section 2, short runs). */
int dpx_pipeline_flush(dpx_pipeline p);
dpx_handle dpx_pipeline_slot(dpx_pipeline p, int32_t slot);

/* Optional: validate intrinsics ahead of the first depth call (the fused kernel divides by fx, fy through a
 * reciprocal sequence that is checked exhaustively against IEEE division for this image geometry; the check
 * synchronises once per new intrinsics tuple, so call this before capturing a HIP graph).  *fast_division
 * receives 1 if the fast sequence is exact for these intrinsics, 0 if the kernel will use IEEE division. */
int dpx_prepare_intrinsics(dpx_handle h, float fx, float fy, float cx, float cy, int32_t* fast_division);

/* DepthImage::DepthImage(path) / reset(path) — utils/depth_image.cpp:27-53: decodes a PNG (host file IO) to ONE 16-bit
 * grey channel with the conversion rules of the reference's stbi_load_16(..., STBI_grey) call (colour -> luma, alpha
 * dropped, palette expanded, 8-bit data widened as v * 257).  *data is allocated by the library (height * width
 * values, row-major); release it with dpx_free.  DPX_ERR_IO with the reference's message if the file cannot be read. */
int dpx_read_depth_png(const char* path, int32_t* height, int32_t* width, uint16_t** data);
void dpx_free(void* p);

/* DepthImage::toPointCloud — utils/depth_image.cpp:55-78: depth H x W uint16 (HOST) -> xyz N x 3 float32 (HOST). */
int dpx_backproject(const uint16_t* depth, int32_t height, int32_t width, float fx, float fy, float cx, float cy,
                    float* xyz, int32_t device);

/* Results of the last process call on this handle (synchronises the handle's last stream). */
int dpx_grid_dims(dpx_handle h, int32_t* nh, int32_t* nv);
int dpx_get_cells(dpx_handle h, int32_t frame, dpx_cell* out);               /* nh*nv records */
int dpx_get_cell_edges(dpx_handle h, int32_t frame, uint8_t* out);           /* nh*nv bytes: bit0 planar, bits2..5 edge U,D,L,R */
int dpx_get_cell_labels(dpx_handle h, int32_t frame, int32_t* out);          /* nh*nv merged labels (toImageLabels input) */
int dpx_get_planes(dpx_handle h, int32_t frame, dpx_plane* out, int32_t cap, int32_t* n_out);
int dpx_get_frame_stats(dpx_handle h, int32_t frame, dpx_frame_stats* out);

/* Measurement hooks. */
/* When enabled, every process call brackets each kernel with HIP events on the launch stream (a ring of the
 * last 128 calls is kept; enabling resets it). */
int dpx_set_profiling(dpx_handle h, int32_t enable);
/* Average milliseconds per kernel (DPX_K_*) over the profiled calls recorded since enabling; synchronises. */
int dpx_get_kernel_ms(dpx_handle h, float* ms /* DPX_K_COUNT */);
/* Launch only the plane-fit kernel `iters` times on `hip_stream` over a device-resident batch and return the
 * average duration in ms measured with HIP events on that stream (input_kind: 0 depth uint16, 1 xyz float32). */
int dpx_time_planefit(dpx_handle h, const void* d_in, int32_t input_kind, int32_t batch, float fx, float fy,
                      float cx, float cy, int32_t iters, void* hip_stream, float* avg_ms);
/* The same for a pipeline group: ONE launch over `nsub` (1..32) device buffers of `per_sub` frames each, the launch
 * shape dpx_pipeline uses (nsub * per_sub <= max_batch of the handle). */
int dpx_time_planefit_group(dpx_handle h, const void* const* d_in, int32_t nsub, int32_t input_kind, int32_t per_sub,
                            float fx, float fy, float cx, float cy, int32_t iters, void* hip_stream, float* avg_ms);
/* Algorithmic bytes one plane-fit launch moves for `batch` frames (input + cell records). */
int64_t dpx_planefit_bytes(dpx_handle h, int32_t input_kind, int32_t batch);

/* Diagnostic build of the region-growing kernel: shader-clock cycles per phase of frame `frame` of the last call
 * (out24, 32 words: P0 state load, P1 seeds+BFS, P2 accumulate, P3 fits, P4 labels, P5 merge, P6 final, total; sums
 * over the sweep rounds of arg-max, seeds + speculative grows + commit, wave-wide BFS; [11] = rounds | grows committed
 * by rounds << 32; BFS steps, grows, candidates, activations; [16..20] = cycles in list scans, cycles in speculative
 * grows, per-lane scan steps, 64-entry scan steps, speculative pops; [21..27] = why rounds stopped, cycles in claims /
 * resolution + commit, rounds that took a single event and their cycles; [28] = rounds whose events came from several
 * chunks of the bin list, [29] = grows given up beyond a round's 64th item, [30] = rounds that continued a sweep without an
 * arg-max, [31] = cycles of the CSR merge walk).  enable != 0 switches the diagnostic
 * instantiation on for later calls; never used when timing. */
#define DPX_DEBUG_PHASE_WORDS 32
int dpx_debug_region_phases(dpx_handle h, int32_t enable, int32_t frame, uint64_t* out32 /* DPX_DEBUG_PHASE_WORDS */);
/* The same switch also times the launches of the core-based region path (dpx_regions3.h; frames beyond ~8 k cells):
 * out32[1] = cycles of the sequence launch (k_regsim), [2] = cycles of the slowest wave of the order launch, [11] = rounds,
 * [13..15] = grows, candidates, activations, [19] = 64-entry list-scan steps, [21] = events that fired alone ("wide"),
 * [22] = events skipped because their bin lost a cell, [23] = wide events with a one-way edge out of their core,
 * [28] = rounds with multi-cell cores. */
int dpx_debug_cores_phases(dpx_handle h, int32_t frame, uint64_t* out32 /* DPX_DEBUG_PHASE_WORDS */);
/* Which kernels a handle runs, as "key=value;..." (regions = tier2 | tier1 | tier0 | tier0-split | cores | v1, plus the
 * variant switches read at dpx_create): ties a result or a timing to its variant.  Returns the length written (without
 * the terminating 0), or a negative error. */
int dpx_get_options(dpx_handle h, char* buf, int32_t cap);
/* Diagnostic: copies one intermediate array of the core-based region path for `frame` of the last call into `out`
 * (which: 0 parent / root per cell [C uint32], 1 csize [C], 2 npairs [C], 3 pstart [C], 4 corepairs [C], 5 seed lists
 * [C x 4 uint32], 6 bin offsets [bins + 1], 7 per-bin cell counts [bins], 8 cursors [4], 9 sequence header [48 int32],
 * 10 cand_grow [pcap x 4], 11 queue [C], 12 cellq [C], 13 cell-graph words [C]).  Returns the bytes written. */
int64_t dpx_debug_read_cores(dpx_handle h, int32_t which, int32_t frame, void* out, int64_t cap_bytes);
/* Diagnostic (handles created with DPX_RESIDENCY=1 in the environment): average shader-clock lifetime of a frame's
 * workgroup in the three region launches (preparation, grow, tail) over the frames of the last call — how long each
 * holds its LDS while other kernels share the chip. */
/* Diagnostic wave trace: installs a device buffer of 65536 slots x `cap_per_slot` entries (4 x uint64 each: start, end in
 * 100 MHz ticks, XCC<<32 | HW_ID, kind | wave-in-block<<4 | blockIdx<<8; kinds 1 plane fit, 2 preparation, 3 grow,
 * 4 tail, 5 labels) that every wave of the pipeline's kernels appends to (slot = its hardware wave slot: XCC, CU, SIMD,
 * wave id); d_buf = NULL switches the trace off.  counts (65536 words) receives the entries written per slot since
 * the previous call.  Synchronises the device. */
int dpx_debug_trace(int32_t device, uint64_t* d_buf, uint32_t cap_per_slot, uint32_t* counts);
int dpx_debug_residency(dpx_handle h, double* out4 /* [3] = grow lifetime in 100 MHz real-time ticks */);

/* Exercises dpx_math.h and the 3x3 eigensolver on the device (parity tests): n inputs each.
 * which: 0 atan2(a,b) 1 sin(a) 2 cos(a) 3 acos(a) 4 sqrt(a).  Host pointers. */
int dpx_test_math(int32_t which, const double* a, const double* b, double* out, int64_t n, int32_t device);
/* A: n x 6 (a00 a01 a02 a11 a12 a22); Q: n x 9 row-major; w: n x 3; rc: n (0 analytic, 1 QL, -1 no convergence). */
int dpx_test_eig3(const double* A, double* Q, double* w, int32_t* rc, int64_t n, int32_t device);

const char* dpx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DEPLEX_AMD_H */
