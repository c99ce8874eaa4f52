// pointops.hip — PointNet++ point operators for the guidance sampler (gfx950), C ABI, raw device
// pointers (torch tensors' data_ptr) + an explicit HIP stream.  Semantics follow the reference's
// torch code (pointnet_pointnet2/models/pointnet2_utils.py): farthest_point_sample :65-86,
// query_ball_point :89-109 (with square_distance :21-42), 3-NN of PointNetFeaturePropagation :295-299.
// float32 arithmetic, -ffp-contract=off; the dot products of square_distance are a forward FMA chain.
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define FPS_NT 1024
#define FPS_MAX_PER_THREAD 8   // N <= 8192

// one workgroup per batch element; cloud in LDS, running min-distance in registers
__global__ __launch_bounds__(FPS_NT) void k_fps(const float *__restrict__ xyz, int N, int S, const long long *__restrict__ start,
                                               long long *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *px = reinterpret_cast<float *>(smem);
    float *py = px + N;
    float *pz = py + N;
    float *rv = pz + N;                       // [16] wave maxima
    int *ri = reinterpret_cast<int *>(rv + 16);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    for (int i = tid; i < N; i += FPS_NT) { px[i] = p[3 * i]; py[i] = p[3 * i + 1]; pz[i] = p[3 * i + 2]; }
    float dist[FPS_MAX_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FPS_MAX_PER_THREAD; j++) dist[j] = 1e10f;
    int far = (int)start[b];
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (tid == 0) out[(size_t)b * S + s] = far;
        const float cx = px[far], cy = py[far], cz = pz[far];
        float bv = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < FPS_MAX_PER_THREAD; j++) {
            int i = tid + j * FPS_NT;
            if (i < N) {
                float dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
                float d = dx * dx + dy * dy + dz * dz;
                if (d < dist[j]) dist[j] = d;
                if (dist[j] > bv) { bv = dist[j]; bi = i; }   // ascending i within the thread: first max kept
            }
        }
        // argmax, lowest index on ties
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float ov = __shfl_xor(bv, off);
            int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { rv[w] = bv; ri[w] = bi; }
        __syncthreads();
        bv = rv[0]; bi = ri[0];
#pragma unroll
        for (int i = 1; i < FPS_NT / 64; i++) {
            float ov = rv[i];
            int oi = ri[i];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        far = bi;
    }
}

// wave64 max of a 64-bit key without LDS traffic: four DPP steps (quad swaps, half-row and row mirrors)
// leave every 16-lane row holding its maximum, v_readlane pulls the four row maxima into scalars
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long k)
{
    unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
    unsigned ol = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    unsigned oh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    unsigned long long o = ((unsigned long long)oh << 32) | ol;
    return o > k ? o : k;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    k = dpp_max_step<0xB1>(k);    // quad_perm [1,0,3,2]
    k = dpp_max_step<0x4E>(k);    // quad_perm [2,3,0,1]
    k = dpp_max_step<0x141>(k);   // row_half_mirror
    k = dpp_max_step<0x140>(k);   // row_mirror
    unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
    unsigned long long r = 0;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, row * 16) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)lo, row * 16);
        r = o > r ? o : r;
    }
    return r;
}

// register-resident variant for N <= 64*NW*PPL: every lane keeps PPL points and their running
// min-distances in registers, so a step is VALU work + a DPP wave reduction of the packed key
// (distance bits, ~index): distances are non-negative, so their bit patterns order like the values
// and the complemented index makes the lowest index win ties, as in k_fps.  NW > 1 waves exchange
// their maxima through double-buffered LDS slots: one barrier per step.  The LDS copy of the cloud
// only serves the centroid lookup.  Same arithmetic and tie-breaking as k_fps.
template <int PPL, int NW>
__global__ __launch_bounds__(64 * NW) void k_fps_wave(const float *__restrict__ xyz, int N, int S, const long long *__restrict__ start,
                                                     long long *__restrict__ out)
{
    constexpr int NT = 64 * NW;
    __shared__ float lp[NT * PPL * 3];
    __shared__ unsigned long long slot[2][NW];
    const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    float px[PPL], py[PPL], pz[PPL], dist[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        int i = tid + NT * j;
        px[j] = py[j] = pz[j] = 0.f;
        dist[j] = 1e10f;
        if (i < N) {
            px[j] = p[3 * i]; py[j] = p[3 * i + 1]; pz[j] = p[3 * i + 2];
            lp[3 * i] = px[j]; lp[3 * i + 1] = py[j]; lp[3 * i + 2] = pz[j];
        }
    }
    __syncthreads();
    int far = (int)start[b];
    for (int s = 0; s < S; s++) {
        if (tid == 0) out[(size_t)b * S + s] = far;
        const float cx = lp[3 * far], cy = lp[3 * far + 1], cz = lp[3 * far + 2];
        float bv = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            int i = tid + NT * j;
            if (i < N) {
                float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
                float d = dx * dx + dy * dy + dz * dz;
                if (d < dist[j]) dist[j] = d;
                if (dist[j] > bv) { bv = dist[j]; bi = i; }
            }
        }
        unsigned long long key = bv < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bv) << 32) | (unsigned)~bi);
        key = wave_max_u64(key);
        if (NW > 1) {
            if ((tid & 63) == 0) slot[s & 1][w] = key;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NW; i++) {
                unsigned long long o = slot[s & 1][i];
                key = o > key ? o : key;
            }
        }
        far = (int)~(unsigned)key;
    }
}

__device__ __forceinline__ float sqdist_ref(float qx, float qy, float qz, float sq, float x, float y, float z)
{
    // square_distance: -2*src.dst^T, += sum(src^2), += sum(dst^2)   (src = query)
    float dot = __builtin_fmaf(qz, z, __builtin_fmaf(qy, y, qx * x));
    float sp = x * x + y * y + z * z;
    return (-2.f * dot + sq) + sp;
}

// one wave per query: first K indices (ascending) with sqrdist <= r2, padded with the first hit
__global__ __launch_bounds__(256) void k_ball_query(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N, int S,
                                                   int K, float r2, long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // global query id over B*S
    const int b = (int)(q / S);
    const float *p = xyz + (size_t)b * N * 3;
    const float qx = new_xyz[q * 3], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
    const float sq = qx * qx + qy * qy + qz * qz;
    long long *o = out + q * K;
    int cnt = 0, first = -1;
    for (int base = 0; base < N && cnt < K; base += 64) {
        int i = base + lane;
        bool hit = false;
        if (i < N) hit = !(sqdist_ref(qx, qy, qz, sq, p[3 * i], p[3 * i + 1], p[3 * i + 2]) > r2);
        unsigned long long m = __ballot(hit);
        if (m) {
            if (first < 0) first = base + __builtin_ctzll(m);
            int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && pos < K) o[pos] = i;
            cnt += __popcll(m);
        }
    }
    if (cnt > K) cnt = K;
    // reference pads with the first group member; an empty ball yields index N there (never happens:
    // the query point itself is always a member) - we pad with 0 in that impossible case
    if (first < 0) first = 0;
    for (int k = cnt + lane; k < K; k += 64) o[k] = first;
}

// one thread per fine point: 3 nearest coarse points (distance ascending, index ascending on ties).
// 64-thread workgroups, grid (ceil(N/64), B); the coarse cloud is staged through LDS in chunks and read back as wave-uniform
// broadcasts, so the scan is VALU-bound instead of waiting on one global load per candidate.
#define NN_CHUNK 1024
__global__ __launch_bounds__(64) void k_three_nn(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int S,
                                                long long total, float *__restrict__ dist_out, long long *__restrict__ idx_out)
{
    __shared__ float cs[NN_CHUNK * 3];
    const int b = blockIdx.y, loc = blockIdx.x * 64 + threadIdx.x;
    const long long g = (long long)b * N + loc;
    const float *c = xyz2 + (size_t)b * S * 3;
    const bool live = loc < N && g < total;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = xyz1[g * 3]; qy = xyz1[g * 3 + 1]; qz = xyz1[g * 3 + 2]; }
    const float sq = qx * qx + qy * qy + qz * qz;
    float d0 = 3.4e38f, d1 = 3.4e38f, d2 = 3.4e38f;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int base = 0; base < S; base += NN_CHUNK) {
        const int m = min(NN_CHUNK, S - base);
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * m; k += 64) cs[k] = c[3 * base + k];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < m; j++) {
            float d = sqdist_ref(qx, qy, qz, sq, cs[3 * j], cs[3 * j + 1], cs[3 * j + 2]);
            int jj = base + j;
            if (d < d0) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = jj; }
            else if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = jj; }
            else if (d < d2) { d2 = d; i2 = jj; }
        }
    }
    if (!live) return;
    dist_out[g * 3] = d0; dist_out[g * 3 + 1] = d1; dist_out[g * 3 + 2] = d2;
    idx_out[g * 3] = i0; idx_out[g * 3 + 1] = i1; idx_out[g * 3 + 2] = i2;
}

extern "C" int nirrt_pn2_fps(const float *xyz, int B, int N, int S, const int64_t *start, int64_t *out, void *stream)
{
    if (N > FPS_NT * FPS_MAX_PER_THREAD || N <= 0 || S <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    const long long *sp = (const long long *)start;
    long long *op = (long long *)out;
    if (N <= 64) hipLaunchKernelGGL((k_fps_wave<1, 1>), dim3(B), dim3(64), 0, st, xyz, N, S, sp, op);
    else if (N <= 256) hipLaunchKernelGGL((k_fps_wave<4, 1>), dim3(B), dim3(64), 0, st, xyz, N, S, sp, op);
    else if (N <= 1024) hipLaunchKernelGGL((k_fps_wave<4, 4>), dim3(B), dim3(256), 0, st, xyz, N, S, sp, op);
    else if (N <= 2048) hipLaunchKernelGGL((k_fps_wave<8, 4>), dim3(B), dim3(256), 0, st, xyz, N, S, sp, op);
    else {
        size_t lds = sizeof(float) * 3 * (size_t)N + 16 * sizeof(float) + 16 * sizeof(int);
        hipLaunchKernelGGL(k_fps, dim3(B), dim3(FPS_NT), lds, st, xyz, N, S, sp, op);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, int K, float r2, int64_t *out,
                                    void *stream)
{
    long long queries = (long long)B * S;
    if (queries % 4 != 0) return -1;   // S is a multiple of 16 in this network
    hipLaunchKernelGGL(k_ball_query, dim3((unsigned)(queries / 4)), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, S, K, r2,
                       (long long *)out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, float *dist, int64_t *idx, void *stream)
{
    long long total = (long long)B * N;
    hipLaunchKernelGGL(k_three_nn, dim3((unsigned)((N + 63) / 64), (unsigned)B), dim3(64), 0, (hipStream_t)stream, xyz1, xyz2, N, S,
                       total, dist, (long long *)idx);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// float64 farthest-point down-sampling of the guidance cloud (the reference calls open3d's
// PointCloud.farthest_point_down_sample, datasets/point_cloud_mask_utils.py:69-72,170-173): start at
// point 0, greedy max-min squared distance, first maximum on ties; the caller keeps the selected points
// in their original order.  One workgroup, running min-distances in registers, cloud read from L2.
// ------------------------------------------------------------------------------------------------
#define FPS64_MAX_PER_THREAD 16   // N <= 16384

// one workgroup per cloud: cloud b = points [off[b], off[b] + cnt[b]) of the concatenated SoA buffer
// (x of all clouds, then y, then z; `total` points in all), ns[b] survivors, sel concatenated likewise
__global__ __launch_bounds__(FPS_NT) void k_fps_f64(const double *__restrict__ buf, long long total, const long long *__restrict__ off,
                                                   const int *__restrict__ cnt, const int *__restrict__ ns,
                                                   unsigned char *__restrict__ sel_all)
{
    __shared__ double rv[FPS_NT / 64];
    __shared__ int ri[FPS_NT / 64];
    const long long o = off[blockIdx.x];
    const int N = cnt[blockIdx.x], S = ns[blockIdx.x];
    if (N <= S) return;   // nothing to drop (the callers keep such clouds whole)
    const double *x = buf + o, *y = buf + total + o, *z = buf + 2 * total + o;
    unsigned char *sel = sel_all + o;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double dist[FPS64_MAX_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FPS64_MAX_PER_THREAD; j++) dist[j] = __builtin_inf();
    for (int i = tid; i < N; i += FPS_NT) sel[i] = 0;
    int far = 0;
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (tid == 0) sel[far] = 1;
        const double cx = x[far], cy = y[far], cz = z[far];
        double bv = -1.;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < FPS64_MAX_PER_THREAD; j++) {
            int i = tid + j * FPS_NT;
            if (i < N) {
                double dx = x[i] - cx, dy = y[i] - cy, dz = z[i] - cz;
                double d = dx * dx + dy * dy + dz * dz;
                if (d < dist[j]) dist[j] = d;
                if (dist[j] > bv) { bv = dist[j]; bi = i; }
            }
        }
#pragma unroll
        for (int off2 = 32; off2 >= 1; off2 >>= 1) {
            double ov = __shfl_xor(bv, off2);
            int oi = __shfl_xor(bi, off2);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { rv[w] = bv; ri[w] = bi; }
        __syncthreads();
        bv = rv[0]; bi = ri[0];
#pragma unroll
        for (int i = 1; i < FPS_NT / 64; i++) {
            double ov = rv[i];
            int oi = ri[i];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        far = bi;
    }
}

struct FpsScratch { void *p = nullptr; size_t cap = 0; };
static std::mutex g_fps_mu;
static FpsScratch g_fps_scratch[16];

// Down-sampling of n_clouds clouds in ONE launch (one workgroup each).  Host pointers in / out: pts = the clouds' (cnt[b], 3)
// row-major f64 points back to back, sel = their keep-bytes back to back (1 = kept), num_samples[b] < cnt[b].
extern "C" int nirrt_fps_f64_batch(const double *pts, int n_clouds, const int *cnt, const int *num_samples, unsigned char *sel,
                                   int device_id)
{
    if (!pts || !sel || !cnt || !num_samples || n_clouds <= 0) return -1;
    if (hipSetDevice(device_id) != hipSuccess) return -4;
    long long total = 0;
    long long *h_off = (long long *)malloc(sizeof(long long) * (size_t)n_clouds);
    for (int b = 0; b < n_clouds; b++) {
        if (cnt[b] <= 0 || num_samples[b] <= 0 || num_samples[b] > cnt[b] || cnt[b] > FPS_NT * FPS64_MAX_PER_THREAD) { free(h_off); return -1; }
        h_off[b] = total;
        total += cnt[b];
    }
    double *h = (double *)malloc(sizeof(double) * 3 * (size_t)total);
    for (long long i = 0; i < total; i++) { h[i] = pts[3 * i]; h[total + i] = pts[3 * i + 1]; h[2 * total + i] = pts[3 * i + 2]; }
    // device scratch: ONE grow-only allocation per device, kept between calls - a batch run refreshes clouds while the
    // other half of the batch is inside a persistent launch, and hipFree would wait for that launch
    const size_t a256 = 255;
    const size_t b_pts = (sizeof(double) * 3 * (size_t)total + a256) & ~a256, b_sel = ((size_t)total + a256) & ~a256;
    const size_t b_off = (sizeof(long long) * (size_t)n_clouds + a256) & ~a256, b_int = (sizeof(int) * (size_t)n_clouds + a256) & ~a256;
    const size_t need = b_pts + b_sel + b_off + 2 * b_int;
    int rc = 0;
    std::lock_guard<std::mutex> hold(g_fps_mu);
    FpsScratch &sc = g_fps_scratch[device_id & 15];
    if (sc.cap < need) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.cap = 0;
        const size_t want = need + need / 2;
        if (hipMalloc(&sc.p, want) == hipSuccess) sc.cap = want; else rc = -2;
    }
    char *base = (char *)sc.p;
    double *d = (double *)base;
    unsigned char *ds = (unsigned char *)(base + b_pts);
    long long *d_off = (long long *)(base + b_pts + b_sel);
    int *d_cnt = (int *)(base + b_pts + b_sel + b_off), *d_ns = (int *)(base + b_pts + b_sel + b_off + b_int);
    if (!rc && (hipMemcpy(d, h, sizeof(double) * 3 * (size_t)total, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_off, h_off, sizeof(long long) * (size_t)n_clouds, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_cnt, cnt, sizeof(int) * (size_t)n_clouds, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_ns, num_samples, sizeof(int) * (size_t)n_clouds, hipMemcpyHostToDevice) != hipSuccess))
        rc = -2;
    if (!rc) {
        hipLaunchKernelGGL(k_fps_f64, dim3(n_clouds), dim3(FPS_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                           (const int *)d_cnt, (const int *)d_ns, ds);
        if (hipGetLastError() != hipSuccess || hipMemcpy(sel, ds, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) rc = -2;
    }
    free(h);
    free(h_off);
    return rc;
}

// one cloud: pts (N,3) row-major f64, sel (N,) bytes (1 = kept)
extern "C" int nirrt_fps_f64(const double *pts, int N, int num_samples, unsigned char *sel, int device_id)
{
    if (N <= 0 || num_samples <= 0) return -1;
    return nirrt_fps_f64_batch(pts, 1, &N, &num_samples, sel, device_id);
}

// ------------------------------------------------------------------------------------------------
// Guidance clouds generated ON the device (datasets/point_cloud_mask_utils.py:35-73, 104-174; datasets_3d/..._3d.py:83-113):
// the over-sampled candidates are drawn from the problem's numpy generator outputs (resident in HBM), filtered, down-sampled
// (k_fps_f64) and compacted without a host round trip.  Arithmetic restated op by op: legacy RandomState doubles
// (a = w0 >> 5, b = w1 >> 6, (a * 2^26 + b) / 2^53), uniform(lo, hi) = lo + (hi - lo) * u, the ellipse transform as numpy's
// dgemm evaluates it for K = 3 (fma chain in k order: fma(a1, s1, a0 * s0), the z term adds an exact zero), np.linalg.norm
// (axis) = sqrt(x*x + y*y), astype(int) = truncation.  The 3D ellipsoid candidates go through sin / cos and stay on the host.
// ------------------------------------------------------------------------------------------------
struct nirrt_cloud_job {
    const unsigned *words;          // DEVICE: generator outputs from the problem's current position (>= 2 * n_doubles of them)
    const unsigned char *free_tab;  // DEVICE, 2D: (h + 1) x (w + 1) table "the 2 x 2 pixel block around this integer position is free"
    const double *balls;            // DEVICE, 3D: (n_ball, 4) cx, cy, cz, r
    const double *boxes;            // DEVICE, 3D: (n_box, 6) x, y, z, w, h, d
    int mode;                       // 0: whole image (2D), 1: ellipse (2D), 2: whole box (3D)
    int w, h, n_ball, n_box, pad;
    double a[8];                    // mode 0: w, h; mode 1: CL00, CL01, CL10, CL11, cx, cy; mode 2: lo[3], hi - lo [3]
    double clearance;               // 3D obstacle inflation
};

__device__ __forceinline__ double mt_double(const unsigned *w, long long j)
{
    const unsigned a = w[2 * j] >> 5, b = w[2 * j + 1] >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

#define CAND_NT 256
// one workgroup per cloud: candidate i of cloud b -> (x, y, z) + keep flag; survivors compacted in order into the SoA
// buffer region [b * n_raw, (b + 1) * n_raw) (x of all clouds, then y, then z), cnt[b] = their number
__global__ __launch_bounds__(CAND_NT) void k_cloud_candidates(const nirrt_cloud_job *jobs, int n_raw, double *buf, long long total, int *cnt)
{
    __shared__ int wave_tot[CAND_NT / 64];
    __shared__ int base_s;
    const nirrt_cloud_job jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *ox = buf + (long long)blockIdx.x * n_raw, *oy = ox + total, *oz = oy + total;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_raw; i0 += CAND_NT) {
        const int i = i0 + tid;
        double x = 0., y = 0., z = 0.;
        bool keep = false;
        if (i < n_raw) {
            if (jb.mode == 2) {
                x = jb.a[0] + jb.a[3] * mt_double(jb.words, 3ll * i);
                y = jb.a[1] + jb.a[4] * mt_double(jb.words, 3ll * i + 1);
                z = jb.a[2] + jb.a[5] * mt_double(jb.words, 3ll * i + 2);
                bool inside = false;
                for (int o = 0; o < jb.n_ball; o++) {
                    const double *c = jb.balls + 4 * o;
                    const double rc = c[3] + jb.clearance;
                    const double dx = x - c[0], dy = y - c[1], dz = z - c[2];
                    inside = inside || (dx * dx + dy * dy + dz * dz < rc * rc);
                }
                for (int o = 0; o < jb.n_box; o++) {
                    const double *bx = jb.boxes + 6 * o;
                    const double cl = jb.clearance;
                    inside = inside || (bx[0] - cl <= x && x <= bx[0] + bx[3] + cl && bx[1] - cl <= y && y <= bx[1] + bx[4] + cl &&
                                        bx[2] - cl <= z && z <= bx[2] + bx[5] + cl);
                }
                keep = !inside;
            } else {
                const double u0 = mt_double(jb.words, 2ll * i), u1 = mt_double(jb.words, 2ll * i + 1);
                bool ok = true;
                if (jb.mode == 0) {
                    x = u0 * jb.a[0]; y = u1 * jb.a[1];
                } else {
                    const double s0 = -1.0 + 2.0 * u0, s1 = -1.0 + 2.0 * u1;
                    ok = __builtin_sqrt(s0 * s0 + s1 * s1) <= 1.0;
                    x = __builtin_fma(jb.a[1], s1, jb.a[0] * s0) + jb.a[4];
                    y = __builtin_fma(jb.a[3], s1, jb.a[2] * s0) + jb.a[5];
                    ok = ok && 0.0 <= x && x <= (double)jb.w && 0.0 <= y && y <= (double)jb.h;
                }
                int ix = (int)x + 1, iy = (int)y + 1;
                ix = ix < 0 ? 0 : (ix > jb.w ? jb.w : ix);
                iy = iy < 0 ? 0 : (iy > jb.h ? jb.h : iy);
                keep = ok && jb.free_tab[iy * (jb.w + 1) + ix] != 0;
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = base_s, tot = 0;
#pragma unroll
        for (int k = 0; k < CAND_NT / 64; k++) { if (k < wv) off += wave_tot[k]; tot += wave_tot[k]; }
        if (keep) {
            const int p = off + __popcll(m & ((1ull << lane) - 1ull));
            ox[p] = x; oy[p] = y; oz[p] = z;
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    if (tid == 0) cnt[blockIdx.x] = base_s;
}

// survivors of cloud b (sel bytes over its region of the SoA buffer) -> out (b, n_points, 3) row-major, in order; n_out[b]
__global__ __launch_bounds__(CAND_NT) void k_cloud_compact(const double *buf, long long total, int n_raw, const int *cnt, int n_points,
                                                           const unsigned char *sel, double *out, int *n_out)
{
    __shared__ int wave_tot[CAND_NT / 64];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = cnt[blockIdx.x];
    const double *x = buf + (long long)blockIdx.x * n_raw, *y = x + total, *z = y + total;
    const unsigned char *sl = sel + (long long)blockIdx.x * n_raw;
    double *o = out + (long long)blockIdx.x * n_points * 3;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += CAND_NT) {
        const int i = i0 + tid;
        const bool keep = i < N && (N <= n_points || sl[i] != 0);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = base_s, tot = 0;
#pragma unroll
        for (int k = 0; k < CAND_NT / 64; k++) { if (k < wv) off += wave_tot[k]; tot += wave_tot[k]; }
        if (keep) {
            const int p = off + __popcll(m & ((1ull << lane) - 1ull));
            if (p < n_points) { o[3 * p] = x[i]; o[3 * p + 1] = y[i]; o[3 * p + 2] = z[i]; }
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    if (tid == 0) n_out[blockIdx.x] = base_s < n_points ? base_s : n_points;
}

static FpsScratch g_cloud_scratch[16];

// Candidates -> down-sampling -> compaction for n_jobs clouds, all on `device_id`.  jobs: HOST array (its pointers are DEVICE
// addresses); clouds: DEVICE (n_jobs, n_points, 3) f64 out; n_cand / n_out: HOST out (candidates that survived the filters / points
// of the final cloud, <= n_points).  n_raw <= 16384.
extern "C" int nirrt_guidance_clouds(const nirrt_cloud_job *jobs, int n_jobs, int n_raw, int n_points, double *clouds, int *n_cand,
                                     int *n_out, int device_id)
{
    if (!jobs || !clouds || !n_cand || !n_out || n_jobs <= 0 || n_raw <= 0 || n_points <= 0 || n_raw > FPS_NT * FPS64_MAX_PER_THREAD) return -1;
    if (hipSetDevice(device_id) != hipSuccess) return -4;
    const long long total = (long long)n_jobs * n_raw;
    const size_t a256 = 255;
    const size_t b_pts = (sizeof(double) * 3 * (size_t)total + a256) & ~a256, b_sel = ((size_t)total + a256) & ~a256;
    const size_t b_off = (sizeof(long long) * (size_t)n_jobs + a256) & ~a256, b_int = (sizeof(int) * (size_t)n_jobs + a256) & ~a256;
    const size_t b_job = (sizeof(nirrt_cloud_job) * (size_t)n_jobs + a256) & ~a256;
    const size_t need = b_pts + b_sel + b_off + 3 * b_int + b_job;
    std::lock_guard<std::mutex> hold(g_fps_mu);
    FpsScratch &sc = g_cloud_scratch[device_id & 15];
    if (sc.cap < need) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.cap = 0;
        const size_t want = need + need / 2;
        if (hipMalloc(&sc.p, want) != hipSuccess) return -2;
        sc.cap = want;
    }
    char *base = (char *)sc.p;
    double *d = (double *)base;
    unsigned char *ds = (unsigned char *)(base + b_pts);
    long long *d_off = (long long *)(base + b_pts + b_sel);
    int *d_cnt = (int *)(base + b_pts + b_sel + b_off), *d_ns = d_cnt + b_int / sizeof(int), *d_nout = d_ns + b_int / sizeof(int);
    nirrt_cloud_job *d_jobs = (nirrt_cloud_job *)(base + b_pts + b_sel + b_off + 3 * b_int);
    long long *h_off = (long long *)malloc(sizeof(long long) * (size_t)n_jobs);
    int *h_ns = (int *)malloc(sizeof(int) * (size_t)n_jobs);
    for (int b = 0; b < n_jobs; b++) { h_off[b] = (long long)b * n_raw; h_ns[b] = n_points; }
    int rc = 0;
    if (hipMemcpy(d_jobs, jobs, sizeof(nirrt_cloud_job) * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_off, h_off, sizeof(long long) * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ns, h_ns, sizeof(int) * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess)
        rc = -2;
    free(h_off);
    free(h_ns);
    if (!rc) {
        hipLaunchKernelGGL(k_cloud_candidates, dim3(n_jobs), dim3(CAND_NT), 0, 0, (const nirrt_cloud_job *)d_jobs, n_raw, d, total, d_cnt);
        // k_fps_f64 leaves clouds with cnt <= num_samples alone (k_cloud_compact keeps all of their points)
        hipLaunchKernelGGL(k_fps_f64, dim3(n_jobs), dim3(FPS_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                           (const int *)d_cnt, (const int *)d_ns, ds);
        hipLaunchKernelGGL(k_cloud_compact, dim3(n_jobs), dim3(CAND_NT), 0, 0, (const double *)d, total, n_raw, (const int *)d_cnt, n_points,
                           (const unsigned char *)ds, clouds, d_nout);
        if (hipGetLastError() != hipSuccess || hipMemcpy(n_cand, d_cnt, sizeof(int) * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(n_out, d_nout, sizeof(int) * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess)
            rc = -2;
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Fused set-abstraction branch (PointNetSetAbstractionMsg.forward, pointnet2_utils.py:236-262, one radius): gather the K
// grouped points of a centroid, run the three 1x1-conv + BatchNorm (folded: W, b) + ReLU layers and take the maximum over
// the K members - without ever writing the grouped tensor (B, S, K, C_in) or an intermediate activation to HBM.
//
// One wave owns 32 rows at a time (one group of K = 32 members or two groups of 16) as TWO 16-row tiles that go through
// v_mfma_f32_16x16x4_f32 side by side (fp32 in, fp32 accumulate - the reference computes in fp32):
//     A (16 x 4)  lane l holds A[l % 16][k(l / 16, step)]     activations
//     B (4 x 16)  lane l holds B[k(l / 16, step)][l % 16]     weights
//     D (16 x 16) lane l, register v holds D[4 * (l / 16) + v][l % 16]
// The reduction index is assigned kq-major, k(kq, step) = kq * (K / 4) + step (any assignment is a valid order of the same
// dot product as long as A and B agree), so that the operands of FOUR consecutive steps are 16 contiguous bytes: a layer
// loads its A operands ONCE (K / 16 ds_read_b128 per tile, kept in registers - the activation tile is dead afterwards and
// the layer's output overwrites it) and every weight fragment with one ds_read_b128 that feeds 4 steps x 2 tiles = 8 MFMAs
// on two independent accumulators.  (Round 2: one ds_read_b32 per operand per MFMA on a single dependent chain.)
// Weights sit in LDS as [column][k] rows of stride K + 4 floats (16-byte aligned rows that start 4 banks apart).
// Input rows are [features of the member, xyz(member) - xyz(centroid)] (:247-250), zero-padded to a multiple of 16 channels.
// ------------------------------------------------------------------------------------------------
typedef float float4_t __attribute__((ext_vector_type(4)));

#define SA_WAVES 4          // waves per workgroup when the branch's LDS footprint allows (the host picks 4, 3 or 2)
#define SA_KMAX 128         // widest layer input (C_in padded, C1, C2)

struct SaMlpArgs {
    const float *feats;     // (B, N, C)
    const float *xyz;       // (B, N, 3)
    const float *new_xyz;   // (B, S, 3)
    const long long *gidx;  // (B, S, K)
    float *out;             // (B, S, out_stride) - this branch writes columns [out_off, out_off + C3)
    const float *w1t, *w2t, *w3t;   // W^T row-major (cin_src x C1), (C1 x C2), (C2 x C3): as packed by the host
    const float *b1, *b2, *b3;
    int B, N, S, K, C, cin_src, Cin, C1, C2, C3, out_stride, out_off;   // Cin = C + 3 padded to a multiple of 16
};

// one layer on the wave's two tiles: act (32 rows x stride sa floats, K_ inputs) -> C_ outputs; LAST: maximum over each group's
// rows instead of a new activation tile
template <bool LAST>
__device__ __forceinline__ void sa_layer(float *act, int sa, const float *W, const float *bias, int K_, int C_, int lane, int gsz,
                                         float (&best)[2][8])
{
    const int row = lane & 15, kq = lane >> 4, kc = K_ >> 2;   // this lane's k range: [kq * kc, (kq + 1) * kc)
    const int sw = K_ + 4;
    float4_t a0[SA_KMAX / 16], a1[SA_KMAX / 16];
#pragma unroll
    for (int j = 0; j < SA_KMAX / 16; j++) {
        if (4 * j < kc) {
            a0[j] = *reinterpret_cast<const float4_t *>(act + row * sa + kq * kc + 4 * j);
            a1[j] = *reinterpret_cast<const float4_t *>(act + (16 + row) * sa + kq * kc + 4 * j);
        }
    }
    // (same wave reads and later writes `act`: the reads above have returned before the first write below is issued only if we
    // wait for them - the compiler's lgkmcnt tracking does that, the values are consumed by the MFMAs first)
    for (int ct = 0, ci = 0; ct < C_; ct += 16, ci++) {
        float4_t acc0, acc1;
        const float bv = bias[ct + row];
        acc0[0] = bv; acc0[1] = bv; acc0[2] = bv; acc0[3] = bv;
        acc1 = acc0;
        const float *wrow = W + (ct + row) * sw + kq * kc;
#pragma unroll
        for (int j = 0; j < SA_KMAX / 16; j++) {
            if (4 * j < kc) {
                const float4_t wf = *reinterpret_cast<const float4_t *>(wrow + 4 * j);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j][u], wf[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j][u], wf[u], acc1, 0, 0, 0);
                }
            }
        }
        if (!LAST) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                act[(4 * kq + v) * sa + ct + row] = acc0[v] > 0.f ? acc0[v] : 0.f;
                act[(16 + 4 * kq + v) * sa + ct + row] = acc1[v] > 0.f ? acc1[v] : 0.f;
            }
        } else {
            // maximum over the rows of each tile, per output column (ReLU first); gsz == 32: one group spans both tiles
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int v = 0; v < 4; v++) { m0 = fmaxf(m0, acc0[v]); m1 = fmaxf(m1, acc1[v]); }
            m0 = fmaxf(m0, __shfl_xor(m0, 16)); m0 = fmaxf(m0, __shfl_xor(m0, 32));
            m1 = fmaxf(m1, __shfl_xor(m1, 16)); m1 = fmaxf(m1, __shfl_xor(m1, 32));
            if (gsz == 32) { best[0][ci] = fmaxf(m0, m1); } else { best[0][ci] = m0; best[1][ci] = m1; }
        }
    }
}

__global__ __launch_bounds__(64 * SA_WAVES) void k_sa_mlp(SaMlpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sa_lds[];
    const int nw = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Cin = a.Cin, C1 = a.C1, C2 = a.C2, C3 = a.C3;
    // weights as [column][k], rows of stride K + 4
    float *W1 = sa_lds;
    float *W2 = W1 + C1 * (Cin + 4);
    float *W3 = W2 + C2 * (C1 + 4);
    float *Bs = W3 + C3 * (C2 + 4);           // C1 + C2 + C3 biases
    int kmax = Cin > C1 ? Cin : C1;
    kmax = kmax > C2 ? kmax : C2;
    const int sa = kmax + 4;                  // activation row stride (16-byte aligned rows, 4 banks apart)
    float *act = Bs + C1 + C2 + C3 + (size_t)w * 32 * sa;
    for (int i = tid; i < C1 * Cin; i += blockDim.x) { const int c = i / Cin, k = i - c * Cin; W1[c * (Cin + 4) + k] = k < a.cin_src ? a.w1t[k * C1 + c] : 0.f; }
    for (int i = tid; i < C2 * C1; i += blockDim.x) { const int c = i / C1, k = i - c * C1; W2[c * (C1 + 4) + k] = a.w2t[k * C2 + c]; }
    for (int i = tid; i < C3 * C2; i += blockDim.x) { const int c = i / C2, k = i - c * C2; W3[c * (C2 + 4) + k] = a.w3t[k * C3 + c]; }
    for (int i = tid; i < C1; i += blockDim.x) Bs[i] = a.b1[i];
    for (int i = tid; i < C2; i += blockDim.x) Bs[C1 + i] = a.b2[i];
    for (int i = tid; i < C3; i += blockDim.x) Bs[C1 + C2 + i] = a.b3[i];
    __syncthreads();
    const int gsz = a.K;                                   // 16 or 32 members per group
    const int gpp = 32 / gsz;                              // groups per 32-row pass
    const long long groups = (long long)a.B * a.S;
    const long long passes = (groups + gpp - 1) / gpp;
    const int CC = a.C + 3;
    for (long long ps = (long long)blockIdx.x * nw + w; ps < passes; ps += (long long)gridDim.x * nw) {
        const long long g0 = ps * gpp;
        // gather: element e of the 32 x Cin tile = (row e / Cin, channel e % Cin); consecutive lanes read consecutive channels
        for (int e = lane; e < 32 * Cin; e += 64) {
            const int r = e / Cin, c = e - r * Cin;
            const long long g = g0 + r / gsz;
            float v = 0.f;
            if (g < groups && c < CC) {
                const int b = (int)(g / a.S);
                const long long idx = a.gidx[g * a.K + (r % gsz)];
                if (c < a.C) v = a.feats[((long long)b * a.N + idx) * a.C + c];
                else {
                    const float p = a.xyz[((long long)b * a.N + idx) * 3 + (c - a.C)];
                    v = p - a.new_xyz[g * 3 + (c - a.C)];
                }
            }
            act[r * sa + c] = v;
        }
        float best[2][8];
        sa_layer<false>(act, sa, W1, Bs, Cin, C1, lane, gsz, best);
        sa_layer<false>(act, sa, W2, Bs + C1, C1, C2, lane, gsz, best);
        sa_layer<true>(act, sa, W3, Bs + C1 + C2, C2, C3, lane, gsz, best);
        if (lane < 16) {
            for (int q = 0; q < gpp; q++) {
                if (g0 + q < groups) {
                    float *o = a.out + (g0 + q) * a.out_stride + a.out_off;
                    for (int ct = 0, ci = 0; ct < C3; ct += 16, ci++) o[ct + lane] = best[q][ci];
                }
            }
        }
    }
}

// One branch of a set-abstraction level.  DEVICE pointers; w*t are the folded weights TRANSPOSED (C_in x C_out, row-major), the
// first with cin_pad >= C + 3 rows (zero rows behind the real ones); C1, C2, C3 multiples of 16 and <= 128, K = 16 or 32.
// LDS: 4 * (C1 * (Cin + 4) + C2 * (C1 + 4) + C3 * (C2 + 4) + C1 + C2 + C3 + waves * 32 * (max(Cin, C1, C2) + 4)) bytes with
// Cin = C + 3 rounded up to 16; the launcher takes 4, 3 or 2 waves per workgroup, whatever fits 160 KB, and returns -3 when not
// even 2 do (the caller then uses library GEMMs).
extern "C" int nirrt_pn2_sa_mlp(const float *feats, const float *xyz, const float *new_xyz, const int64_t *gidx, int B, int N, int S,
                                int K, int C, int cin_pad, const float *w1t, const float *b1, int C1, const float *w2t, const float *b2,
                                int C2, const float *w3t, const float *b3, int C3, float *out, int out_stride, int out_off, void *stream)
{
    if (C1 % 16 || C2 % 16 || C3 % 16 || C3 > 128 || C1 > SA_KMAX || C2 > SA_KMAX || cin_pad < C + 3 || (K != 16 && K != 32) || B <= 0 || S <= 0) return -1;
    const int Cin = (C + 3 + 15) / 16 * 16;
    if (Cin > SA_KMAX) return -3;
    int kmax = Cin > C1 ? Cin : C1;
    kmax = kmax > C2 ? kmax : C2;
    const size_t fixed = (size_t)C1 * (Cin + 4) + (size_t)C2 * (C1 + 4) + (size_t)C3 * (C2 + 4) + C1 + C2 + C3;
    int nw = SA_WAVES;
    size_t lds = 0;
    for (; nw >= 2; nw--) {
        lds = sizeof(float) * (fixed + (size_t)nw * 32 * (kmax + 4));
        if (lds <= 160 * 1024) break;
    }
    if (nw < 2) return -3;
    // the attribute belongs to the CURRENT device's copy of the kernel (a process may drive several GPUs): set on every call,
    // it is a host-side table write
    if (hipFuncSetAttribute((const void *)k_sa_mlp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    SaMlpArgs a;
    a.feats = feats; a.xyz = xyz; a.new_xyz = new_xyz; a.gidx = (const long long *)gidx; a.out = out;
    a.w1t = w1t; a.b1 = b1; a.w2t = w2t; a.b2 = b2; a.w3t = w3t; a.b3 = b3;
    a.B = B; a.N = N; a.S = S; a.K = K; a.C = C; a.cin_src = cin_pad; a.Cin = Cin; a.C1 = C1; a.C2 = C2; a.C3 = C3;
    a.out_stride = out_stride; a.out_off = out_off;
    const long long groups = (long long)B * S;
    const long long passes = (groups + (32 / K) - 1) / (32 / K);
    long long grid = (passes + nw - 1) / nw;
    if (grid > 2048) grid = 2048;      // grid-stride over the passes: the weights are staged once per workgroup
    hipLaunchKernelGGL(k_sa_mlp, dim3((unsigned)grid), dim3(64 * nw), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
