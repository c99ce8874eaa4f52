/* nirrt_pointops.h — C ABI of the PointNet++ guidance sampler's point operators in libnirrt_hip.so (gfx950).
 *
 * The reference evaluates these with stock PyTorch ops / open3d calls; each entry names what it replaces.  Pointers marked
 * DEVICE are addresses in HBM (e.g. torch.cuda tensor.data_ptr()); `stream` is a hipStream_t (0 = default stream).
 * All functions return 0 on success, < 0 on error (-1 bad argument, -2 HIP error, -4 bad device); nothing throws.
 */
#ifndef NIRRT_POINTOPS_H
#define NIRRT_POINTOPS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* farthest_point_sample, pointnet_pointnet2/models/pointnet2_utils.py:65-86: xyz DEVICE f32 (B, N, 3), start DEVICE i64 (B,)
 * (the reference draws it with torch.randint on the CPU generator, :77), out DEVICE i64 (B, S).  Ties -> lowest index
 * (torch.max).  One persistent workgroup per cloud. */
int nirrt_pn2_fps(const float *xyz, int B, int N, int S, const int64_t *start, int64_t *out, void *stream);

/* query_ball_point, pointnet2_utils.py:89-109: the first K indices (ascending) with squared distance <= r2, padded with the
 * first hit; xyz DEVICE f32 (B, N, 3), new_xyz DEVICE f32 (B, S, 3), out DEVICE i64 (B, S, K).  B*S must be a multiple of 4. */
int nirrt_pn2_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, int K, float r2, int64_t *out, void *stream);

/* Ragged batches (round 6): clouds of DIFFERENT sizes in one forward.  The reference classifies one cloud at a time
 * (pointnet2_wrapper.py:43-58), so a batch may hold any clouds as long as each is treated as if it were alone: only the first
 * set-abstraction level looks at the cloud itself (farthest-point sampling and the ball queries over its n points); everything
 * behind it works on 1024 / 256 / 64 / 16 sampled points or per point.  n_valid DEVICE i32 (B,): cloud b has n_valid[b] <= N
 * points in its N-point row (S <= n_valid[b], N <= 2048 for the sampling); the padding is never read.  n_valid = NULL: the
 * functions above. */
int nirrt_pn2_fps_ragged(const float *xyz, int B, int N, int S, const int64_t *start, const int32_t *n_valid, int64_t *out, void *stream);
int nirrt_pn2_ball_query_ragged(const float *xyz, const float *new_xyz, int B, int N, int S, int K, float r2, const int32_t *n_valid,
                                int64_t *out, void *stream);

/* three nearest coarse points of every fine point (PointNetFeaturePropagation, pointnet2_utils.py:295-299: sort of
 * square_distance, first three): xyz1 DEVICE f32 (B, N, 3), xyz2 DEVICE f32 (B, S, 3) -> dist DEVICE f32 (B, N, 3) squared
 * distances ascending, idx DEVICE i64 (B, N, 3). */
int nirrt_pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, float *dist, int64_t *idx, void *stream);

/* One radius branch of PointNetSetAbstractionMsg.forward (pointnet2_utils.py:236-262) fused into one kernel on
 * v_mfma_f32_16x16x4_f32 tiles: gather the K members of every centroid's group ([features, xyz - centroid], :247-250), three
 * folded conv1x1+BatchNorm+ReLU layers, maximum over the members.  DEVICE pointers: feats f32 (B, N, C), xyz f32 (B, N, 3),
 * new_xyz f32 (B, S, 3), gidx i64 (B, S, K) from nirrt_pn2_ball_query; w1t / w2t / w3t = the layers' weights TRANSPOSED
 * (C_in x C_out row-major; w1t has cin_pad >= C + 3 rows, a multiple of 4, zero rows behind the real ones), b1 / b2 / b3
 * their biases; C1, C2, C3 multiples of 16, C3 <= 128.  Writes out[b, s, out_off : out_off + C3] of out f32 (B, S, out_stride).
 * Returns -3 (and does nothing) when the branch's weights + activation tiles exceed the 160 KB of LDS. */
int nirrt_pn2_sa_mlp(const float *feats, const float *xyz, const float *new_xyz, const int64_t *gidx, int B, int N, int S, int K, int C,
                     int cin_pad, const float *w1t, const float *b1, int C1, const float *w2t, const float *b2, int C2,
                     const float *w3t, const float *b3, int C3, float *out, int out_stride, int out_off, void *stream);

/* The network's (6, n) float32 input block of guidance clouds that are already resident (classify_path_points,
 * pointnet_pointnet2/pointnet2_wrapper.py:43-58: pc_normalize of the float32 cloud, start / goal masks of
 * get_point_cloud_mask_around_points with the float64 cloud, "neither" channel), bit-equal to the numpy evaluation.  DEVICE
 * pointers: clouds f64 (n_clouds, stride_pts, 3) (z = 0 for planar clouds), rows i32 (n_rows,) = the clouds to process (all of
 * n points), starts / goals f64 (n_rows, 3), out f32 (n_rows, 6, n).  n <= 12288. */
int nirrt_pn2_net_input(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n, const double *starts,
                        const double *goals, double radius, float *out, void *stream);
/* ragged: n_each DEVICE i32 (n_rows,) points of each cloud (<= n); out f32 (n_rows, 6, n), ZEROED by the caller - the columns
 * behind a cloud's own points are left alone; mean and largest norm of pc_normalize are the cloud's own */
int nirrt_pn2_net_input_ragged(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n, const int32_t *n_each,
                               const double *starts, const double *goals, double radius, float *out, void *stream);

/* the same block with GIVEN indicator channels: start_masks / goal_masks DEVICE bytes, cloud `row` at + row * mask_stride (the
 * masks of the neural-connect rounds are re-seeded at boundary points, pointnet2_wrapper_connect_bfs.py:181-216) */
int nirrt_pn2_net_input_masks(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n,
                              const uint8_t *start_masks, const uint8_t *goal_masks, int64_t mask_stride, float *out, void *stream);
/* (ragged, as nirrt_pn2_net_input_ragged) */
int nirrt_pn2_net_input_masks_ragged(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n, const int32_t *n_each,
                                     const uint8_t *start_masks, const uint8_t *goal_masks, int64_t mask_stride, float *out, void *stream);

/* Input rows of the set-abstraction levels whose MLP runs as library GEMMs: sample_and_group's concatenation
 * (pointnet2_utils.py:247-250) in one pass.  DEVICE pointers as for nirrt_pn2_sa_mlp; C a multiple of 4;
 * out f32 (B * S * K, C + 4) = [feats[b, gidx], xyz[b, gidx] - new_xyz[b, s], 0] (the zero column keeps rows 16-byte aligned). */
int nirrt_pn2_group_rows(const float *feats, const float *xyz, const float *new_xyz, const int64_t *gidx, int B, int N, int S, int K,
                         int C, float *out, void *stream);

/* Input rows of a feature-propagation level (PointNetFeaturePropagation.forward, pointnet2_utils.py:295-309): inverse-distance
 * weights of the three neighbours from nirrt_pn2_three_nn (dist, idx), interpolation of the coarse features and concatenation
 * behind the fine level's own: out f32 (B * N, C1 + C2) = [feats1[b, n], sum_j w_j feats2[b, idx_j]], w_j = (1 / (d_j + 1e-8)) /
 * sum.  feats1 DEVICE f32 (B, N, C1) or NULL with C1 = 0, feats2 DEVICE f32 (B, S, C2); C1, C2 multiples of 4. */
int nirrt_pn2_fp_rows(const float *feats1, const float *feats2, const float *dist, const int64_t *idx, int B, int N, int S, int C1,
                      int C2, float *out, void *stream);

/* open3d PointCloud.farthest_point_down_sample as called by datasets/point_cloud_mask_utils.py:69-72,170-173 and
 * datasets_3d/point_cloud_mask_utils_3d.py:49-53,196-199 (un-vendored dependency, behaviour restated: start at point 0,
 * greedy max-min squared distance in float64, first maximum on ties).  HOST pointers: pts (N, 3) f64 row-major,
 * sel (N,) bytes out (1 = kept; the caller keeps the survivors in their original order).  N <= 16384. */
int nirrt_fps_f64(const double *pts, int N, int num_samples, unsigned char *sel, int device_id);

/* the same for n_clouds clouds in ONE launch (one workgroup per cloud): pts / sel hold the clouds back to back,
 * cnt[b] points and num_samples[b] <= cnt[b] survivors each (HOST pointers). */
int nirrt_fps_f64_batch(const double *pts, int n_clouds, const int *cnt, const int *num_samples, unsigned char *sel, int device_id);

/* Guidance clouds generated on the device: generate_rectangle_point_cloud / ellipsoid_point_cloud_sampling
 * (datasets/point_cloud_mask_utils.py:35-73, 104-174) and generate_rectangle_point_cloud_3d / ellipsoid_point_cloud_sampling_3d
 * (datasets_3d/point_cloud_mask_utils_3d.py:83-113, 132-200) for n_jobs problems at once - candidates from each problem's numpy generator outputs (resident in HBM, consumed like
 * rng.random_sample / rng.uniform would: 2 words per double), free-space filter, farthest-point down-sampling, survivors in their
 * original order.  jobs: HOST array whose pointers are DEVICE addresses; clouds: DEVICE (n_jobs, n_points, 3) f64 out; n_cand /
 * n_out: HOST out (candidates after the filters / points of the cloud).  The caller advances the problem's generator by
 * 2 * (2 or 3) * n_raw words.  The 3D ellipsoid candidates (ellipsoid_point_cloud_sampling_3d, point_cloud_mask_utils_3d.py:
 * 132-200) go through the device's sin / cos: equal to the host's within a few ulp (<= 1e-9 on the coordinates), not bit for bit. */
typedef struct nirrt_cloud_job {
    const uint32_t *words;     /* DEVICE */
    const uint8_t *free_tab;   /* DEVICE, 2D: (h + 1) x (w + 1), 1 = the 2 x 2 pixel block around the integer position is free */
    const double *balls;       /* DEVICE, 3D: (n_ball, 4) */
    const double *boxes;       /* DEVICE, 3D: (n_box, 6) */
    int32_t mode;              /* 0 whole image (2D), 1 ellipse (2D), 2 whole box (3D), 3 ellipsoid (3D); one batch = 2D jobs or 3D jobs */
    int32_t w, h, n_ball, n_box, pad;
    double a[20];              /* mode 0: w, h; mode 1: (C.L)[0][0], [0][1], [1][0], [1][1], x_center[0], [1]; mode 2: lo[3], (hi - lo)[3];
                                  mode 3: C.L row-major [9], x_center [3], range lo [3], range hi [3] */
    double clearance;
} nirrt_cloud_job;
int nirrt_guidance_clouds(const nirrt_cloud_job *jobs, int n_jobs, int n_raw, int n_points, double *clouds, int *n_cand, int *n_out,
                          int device_id);

/* Neural connect for a batch of clouds (PNGWrapper.generate_connected_path_points, wrapper{,_3d}/pointnet_pointnet2/
 * pointnet2_wrapper_connect_bfs.py:76-240, with bfs_point_cloud_visualization / get_boundary_mask /
 * select_heuristic_boundary_point of wrapper/utils/bfs_connect_heuristic.py:80-139, 5-29, 142-181), one workgroup per cloud.
 * The reference's float32 numpy arithmetic is restated op by op (float32 differences, (d0^2 + d1^2) [+ d2^2], correctly rounded
 * sqrt, strict `< radius`).  jobs: HOST array whose pointers are DEVICE addresses; n <= 2048 points.
 *   nirrt_connect_round: path_mask |= pred; breadth-first reachability start -> goal (then goal -> start) over the predicted
 *     points; has_path[b] = 1 ends the rounds of cloud b; otherwise boundary masks (jobs[b].boundary, (2, n)) and the heuristic
 *     boundary point of each search: seed_idx (n_jobs, 2), -1 = no boundary point; tie (n_jobs, 2) != 0: two boundary points
 *     share a key of the rank heuristic - numpy's argsort is not stable, the caller repeats that choice with numpy on the
 *     boundary mask.  has_path / seed_idx / tie: HOST outputs.
 *   nirrt_connect_masks: start / goal masks of the next classification from seeds (HOST (n_jobs, 2)): -2 = the start / goal
 *     state (the masks before the first round), -1 = keep the mask, >= 0 = that cloud point. */
typedef struct nirrt_connect_job {
    const double *cloud;     /* DEVICE (n, 3) f64, z = 0 for planar clouds (the float32 cloud of the reference = its rounding) */
    const uint8_t *pred;     /* DEVICE (n,): this round's path_pred != 0 */
    uint8_t *path_mask;      /* DEVICE (n,) in / out: path_pred_mask, the union of the predictions so far */
    uint8_t *start_mask;     /* DEVICE (n,) in / out */
    uint8_t *goal_mask;      /* DEVICE (n,) in / out */
    uint8_t *boundary;       /* DEVICE (2, n) out */
    int32_t n, dim;
    double start[3], goal[3];
} nirrt_connect_job;
int nirrt_connect_round(const nirrt_connect_job *jobs, int n_jobs, double radius, int32_t *has_path, int32_t *seed_idx, int32_t *tie,
                        int device_id);
int nirrt_connect_masks(const nirrt_connect_job *jobs, int n_jobs, double radius, const int32_t *seed_idx, int device_id);

/* The library's restatement of glibc 2.35's atan2 / sin / cos (csrc/glibc235_libm.inc: what math.atan2 / math.cos / math.sin of
 * the reference's new_state, rrt_star_2d.py:67-78, and np.sin / np.cos of irrt_star_3d.py:146-158 resolve to on an x86-64 FMA host
 * with that libm) evaluated ON THE DEVICE for the caller's arguments: fn 0 = atan2(a[i], b[i]), 1 = sin(a[i]), 2 = cos(a[i]);
 * a, b, out HOST f64 (n,).  A caller compares `out` with its own libm bit for bit before trusting the device-side steer / samplers
 * on this host (nirrt_star_amd/_hip.libm_check does, once per process); arguments whose restated path the translator could not
 * express (|x| > 1e8, Inf) return NaN. */
int nirrt_libm_probe(int32_t fn, int64_t n, const double *a, const double *b, double *out, int device_id);

#ifdef __cplusplus
}
#endif
#endif /* NIRRT_POINTOPS_H */
