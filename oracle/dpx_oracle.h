/* dpx_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a scalar CPU restatement of prime-slam/deplex's PlaneExtractor::process()
 * (reference: cpp/deplex/src/deplex/plane_extractor.cpp:187-283 and the files it calls), written
 * without Eigen, with the reference's float32 operation order.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (deplex_amd/) never links, imports or executes anything under oracle/.
 *
 * Parity pin: see oracle/README.md — pinned by the reference's own known-answer tests
 * (34 planes on the TUM frame for both input variants, the all-zero and throwing cases), and its
 * 3x3 eigensolver is checked bit-for-bit against the reference's own libs/dsyev C sources built into
 * oracle/_ref/ by oracle/Makefile.
 */
#ifndef DPX_ORACLE_H
#define DPX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors deplex::config::Config (cpp/deplex/include/deplex/config.h:51-81), same defaults. */
typedef struct orc_config {
  int32_t patch_size;                         /* 10   */
  int32_t histogram_bins_per_coord;           /* 20   */
  float min_cos_angle_merge;                  /* 0.90 */
  float max_merge_dist;                       /* 500  */
  int32_t min_region_growing_candidate_size;  /* 5    */
  int32_t min_region_growing_cells_activated; /* 4    */
  float min_region_planarity_score;           /* 0.55 */
  float depth_sigma_coeff;                    /* 1.425e-6 */
  float depth_sigma_margin;                   /* 10   */
  int32_t min_pts_per_cell;                   /* 3    */
  float depth_discontinuity_threshold;        /* 160  */
  int32_t max_number_depth_discontinuity;     /* 1    */
  int32_t ransac_refinement;                  /* 0    */
  int32_t ransac_max_iterations;              /* 1000 */
  float ransac_threshold;                     /* 1    */
  float ransac_inliers_ratio;                 /* 0.9  */
} orc_config;

/* One record per grid cell, in cell-id order (cell = cell_row * nh + cell_col). */
typedef struct orc_cell {
  float sum[3];    /* coord_sum_            (cell_segment_stat.cpp:31) */
  float pp[6];     /* variance_ xx xy xz yy yz zz (cell_segment_stat.cpp:32) */
  float mean[3];   /* mean_                 (:33) */
  float normal[3]; /* normal_               (:76) */
  float d;         /* d_                    (:77) */
  float mse;       /* mse_                  (:79); FLT_MAX for invalid cells (:27) */
  float score;     /* score_                (:80); 0 for invalid cells */
  float tol;       /* merge_tolerance_      (cell_segment.cpp:34) */
  int32_t valid;   /* passed hasValidPoints && isDepthContinuous (cell_segment.cpp:27-30) */
  int32_t planar;  /* is_planar_            (cell_segment.cpp:32) */
  int32_t bin;     /* histogram bin, -1 when not planar (normals_histogram.cpp:44) */
  int32_t used_ql; /* 1 when the eigensolver fell back to QL (dsyevh3.c:103,118) */
} orc_cell;

/* One record per grown segment (plane_segments[i], plane_extractor.cpp:337), AFTER findMergedLabels. */
typedef struct orc_plane {
  float sum[3];
  float pp[6];
  float mean[3];
  float normal[3];
  float d, mse, score;
  int32_t n_pts;       /* nr_pts_ */
  int32_t n_cells;     /* cells activated by the grow that created the segment */
  int32_t merge_label; /* plane_merge_labels[i] (0-based) */
} orc_plane;

typedef struct orc_stats {
  int32_t nh, nv;
  int32_t planar_cells;
  int32_t grows;          /* iterations of the createPlaneSegments loop that reached growSeed */
  int32_t activations;    /* total cells activated over all grows */
  int32_t max_grow;
  int32_t segments;       /* plane_segments.size() */
  int32_t merges;         /* number of (col -> plane) merges performed */
  int32_t ql_fallbacks;   /* eigen solves that took the QL path (cells + segments + merges) */
  int32_t eig_solves;
} orc_stats;

/* math flavour: 0 = glibc libm (what the reference links), 1 = deplex_amd/csrc/dpx_math.h
 * (bit-identical to the device code).  Process-global. */
void orc_set_math(int flavour);
int orc_get_math(void);

void orc_default_config(orc_config* cfg);

/* DepthImage::toPointCloud (utils/depth_image.cpp:55-78): xyz is N x 3 row-major float32. */
void orc_backproject(const uint16_t* depth, int32_t height, int32_t width, float fx, float fy, float cx,
                     float cy, float* xyz);

/* PlaneExtractor(h, w, cfg) + process(xyz).  n_points is pcd_array.rows().
 * Returns 0 on success; -1 patch_size==0 (ctor throw, plane_extractor.cpp:161-164);
 * -2 shape mismatch (process throw, :188-194); -3 configuration the reference leaves undefined.
 * err (may be NULL) receives the reference's exception message.
 * cells_out: nh*nv records or NULL.  planes_out: up to planes_cap records or NULL.
 * cell_labels_out: nv*nh int32 (labels_map_ before toImageLabels) or NULL. */
int orc_process(const orc_config* cfg, int32_t height, int32_t width, const float* xyz, int64_t n_points,
                int32_t* labels, orc_cell* cells_out, orc_plane* planes_out, int32_t planes_cap,
                int32_t* cell_labels_out, orc_stats* stats, char* err, int32_t err_cap);

/* Convenience: back-project then process, the depth path of the GPU pipeline. */
int orc_process_depth(const orc_config* cfg, int32_t height, int32_t width, const uint16_t* depth, float fx,
                      float fy, float cx, float cy, int32_t* labels, orc_cell* cells_out,
                      orc_plane* planes_out, int32_t planes_cap, int32_t* cell_labels_out, orc_stats* stats,
                      char* err, int32_t err_cap);

/* Frame-parallel batch over `threads` std::threads (one extractor each) — the CPU baseline leg. */
int orc_process_depth_batch(const orc_config* cfg, int32_t height, int32_t width, const uint16_t* depth,
                            int32_t batch, float fx, float fy, float cx, float cy, int32_t* labels,
                            int32_t threads);

/* The reference's dsyevh3 restated (libs/dsyev/src/dsyevh3.c:31-134).  A, Q row-major 3x3.
 * Returns 0 analytic path, 1 QL fallback, -1 QL did not converge. */
int orc_eig3(const double* A, double* Q, double* w);

/* CellSegmentStat(points) on an arbitrary n x 3 row-major matrix (used by the refinement test,
 * cpp/tests/test_refinement.cpp:29-41).  Returns mse. */
float orc_points_mse(const float* xyz, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
