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

/* three nearest coarse points of every fine point (PointNetFeaturePropagation, pointnet2_utils.py:295-299: sort of
 * square_distance, first three): xyz1 DEVICE f32 (B, N, 3), xyz2 DEVICE f32 (B, S, 3) -> dist DEVICE f32 (B, N, 3) squared
 * distances ascending, idx DEVICE i64 (B, N, 3). */
int nirrt_pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, float *dist, int64_t *idx, void *stream);

/* open3d PointCloud.farthest_point_down_sample as called by datasets/point_cloud_mask_utils.py:69-72,170-173 and
 * datasets_3d/point_cloud_mask_utils_3d.py:49-53,196-199 (un-vendored dependency, behaviour restated: start at point 0,
 * greedy max-min squared distance in float64, first maximum on ties).  HOST pointers: pts (N, 3) f64 row-major,
 * sel (N,) bytes out (1 = kept; the caller keeps the survivors in their original order).  N <= 16384. */
int nirrt_fps_f64(const double *pts, int N, int num_samples, unsigned char *sel, int device_id);

/* the same for n_clouds clouds in ONE launch (one workgroup per cloud): pts / sel hold the clouds back to back,
 * cnt[b] points and num_samples[b] <= cnt[b] survivors each (HOST pointers). */
int nirrt_fps_f64_batch(const double *pts, int n_clouds, const int *cnt, const int *num_samples, unsigned char *sel, int device_id);

#ifdef __cplusplus
}
#endif
#endif /* NIRRT_POINTOPS_H */
