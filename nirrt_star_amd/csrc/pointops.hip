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
// Fused set-abstraction branch (PointNetSetAbstractionMsg.forward, pointnet2_utils.py:236-262, one radius): gather the K
// grouped points of a centroid, run the three 1x1-conv + BatchNorm (folded: W, b) + ReLU layers and take the maximum over
// the K members - without ever writing the grouped tensor (B, S, K, C_in) or an intermediate activation to HBM.
//
// One wave owns one group at a time; its rows are processed as 16-row tiles through v_mfma_f32_16x16x4_f32 (fp32 in,
// fp32 accumulate - the reference computes in fp32):
//     A (16 x 4)  lane l holds A[l % 16][l / 16]          activations, read from the wave's LDS tile
//     B (4 x 16)  lane l holds B[l / 16][l % 16]          W^T, read from the workgroup's LDS copy of the weights
//     D (16 x 16) lane l, register v holds D[4 * (l / 16) + v][l % 16]
// Layer l: for every 16-column tile, acc = bias; acc += A[:, k0:k0+4] . W^T[k0:k0+4, tile] over k0; ReLU; the tile goes to the
// other LDS activation buffer (layers 1, 2) or is reduced over its 16 rows in registers (layer 3: max over the 4 registers
// of a lane, then over the 4 lane groups with two DPP-free shuffles) and written out.  Input rows are
// [features of the member, xyz(member) - xyz(centroid)] (:247-250), zero-padded to a multiple of 4 channels.
// ------------------------------------------------------------------------------------------------
typedef float float4_t __attribute__((ext_vector_type(4)));

#define SA_WAVES 4

struct SaMlpArgs {
    const float *feats;     // (B, N, C)
    const float *xyz;       // (B, N, 3)
    const float *new_xyz;   // (B, S, 3)
    const long long *gidx;  // (B, S, K)
    float *out;             // (B, S, out_stride) - this branch writes columns [out_off, out_off + C3)
    const float *w1t, *b1, *w2t, *b2, *w3t, *b3;   // W^T row-major (C_in_pad x C1), (C1 x C2), (C2 x C3); biases
    int B, N, S, K, C, Cin_pad, C1, C2, C3, out_stride, out_off;
};

__global__ __launch_bounds__(64 * SA_WAVES) void k_sa_mlp(SaMlpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sa_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Cin = a.Cin_pad, C1 = a.C1, C2 = a.C2, C3 = a.C3;
    float *W1 = sa_lds;                       // Cin x C1
    float *W2 = W1 + Cin * C1;                // C1 x C2
    float *W3 = W2 + C1 * C2;                 // C2 x C3
    float *Bs = W3 + C2 * C3;                 // C1 + C2 + C3 biases
    const int sA = (Cin > C2 ? Cin : C2) + 1, sB = C1 + 1;      // row strides (odd: rows fall into different banks)
    float *bufA = Bs + C1 + C2 + C3 + (size_t)w * 16 * (sA + sB);   // this wave's tiles: input / layer-2 output ...
    float *bufB = bufA + 16 * sA;                                   // ... and layer-1 output
    for (int i = tid; i < Cin * C1; i += 64 * SA_WAVES) W1[i] = a.w1t[i];
    for (int i = tid; i < C1 * C2; i += 64 * SA_WAVES) W2[i] = a.w2t[i];
    for (int i = tid; i < C2 * C3; i += 64 * SA_WAVES) W3[i] = a.w3t[i];
    for (int i = tid; i < C1; i += 64 * SA_WAVES) Bs[i] = a.b1[i];
    for (int i = tid; i < C2; i += 64 * SA_WAVES) Bs[C1 + i] = a.b2[i];
    for (int i = tid; i < C3; i += 64 * SA_WAVES) Bs[C1 + C2 + i] = a.b3[i];
    __syncthreads();
    const int row = lane & 15, kq = lane >> 4;            // A / B operand coordinates of this lane
    const long long groups = (long long)a.B * a.S;
    const int CC = a.C + 3;
    for (long long g = (long long)blockIdx.x * SA_WAVES + w; g < groups; g += (long long)gridDim.x * SA_WAVES) {
        const int b = (int)(g / a.S);
        const float *cen = a.new_xyz + g * 3;
        const float cx = cen[0], cy = cen[1], cz = cen[2];
        float best[8];                                     // running maxima of this lane's output columns (C3 <= 128)
#pragma unroll
        for (int i = 0; i < 8; i++) best[i] = -3.4e38f;
        for (int r0 = 0; r0 < a.K; r0 += 16) {             // 16-row tiles of the group
            // gather: element e of the tile = (row e / Cin, channel e % Cin); consecutive lanes read consecutive channels
            for (int e = lane; e < 16 * Cin; e += 64) {
                const int r = e / Cin, c = e - r * Cin;
                float v = 0.f;
                if (r0 + r < a.K && c < CC) {
                    const long long idx = a.gidx[g * a.K + r0 + r];
                    if (c < a.C) v = a.feats[((long long)b * a.N + idx) * a.C + c];
                    else {
                        const float p = a.xyz[((long long)b * a.N + idx) * 3 + (c - a.C)];
                        v = p - (c - a.C == 0 ? cx : (c - a.C == 1 ? cy : cz));
                    }
                }
                bufA[r * sA + c] = v;
            }
            // rows beyond K (K < 16 never happens in this network; K is 16 or 32) would be padding: keep them out of the max
            // layer 1: bufA (16 x Cin) -> bufB (16 x C1)
            for (int ct = 0; ct < C1; ct += 16) {
                float4_t acc;
                const float bias = Bs[ct + row];
                acc[0] = bias; acc[1] = bias; acc[2] = bias; acc[3] = bias;
                for (int k0 = 0; k0 < Cin; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bufA[row * sA + k0 + kq], W1[(k0 + kq) * C1 + ct + row], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++) bufB[(4 * kq + v) * sB + ct + row] = acc[v] > 0.f ? acc[v] : 0.f;
            }
            // layer 2: bufB (16 x C1) -> bufA (16 x C2)
            for (int ct = 0; ct < C2; ct += 16) {
                float4_t acc;
                const float bias = Bs[C1 + ct + row];
                acc[0] = bias; acc[1] = bias; acc[2] = bias; acc[3] = bias;
                for (int k0 = 0; k0 < C1; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bufB[row * sB + k0 + kq], W2[(k0 + kq) * C2 + ct + row], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++) bufA[(4 * kq + v) * sA + ct + row] = acc[v] > 0.f ? acc[v] : 0.f;
            }
            // layer 3: bufA (16 x C2) -> maximum over the rows, per output column
            for (int ct = 0, ci = 0; ct < C3; ct += 16, ci++) {
                float4_t acc;
                const float bias = Bs[C1 + C2 + ct + row];
                acc[0] = bias; acc[1] = bias; acc[2] = bias; acc[3] = bias;
                for (int k0 = 0; k0 < C2; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bufA[row * sA + k0 + kq], W3[(k0 + kq) * C3 + ct + row], acc, 0, 0, 0);
                float m = -3.4e38f;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const float x = acc[v] > 0.f ? acc[v] : 0.f;
                    if (r0 + 4 * kq + v < a.K) m = x > m ? x : m;
                }
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                best[ci] = fmaxf(best[ci], m);             // every lane now holds the column maximum of its `row` column
            }
        }
        if (lane < 16) {
            float *o = a.out + g * a.out_stride + a.out_off;
            for (int ct = 0, ci = 0; ct < C3; ct += 16, ci++) o[ct + lane] = best[ci];
        }
    }
}

// One branch of a set-abstraction level.  DEVICE pointers; w*t are the folded weights TRANSPOSED (C_in x C_out, row-major), the
// first with its C + 3 input rows padded with zero rows to cin_pad (a multiple of 4); C1, C2, C3 multiples of 16, C3 <= 128.
// Needs 4 * (cin_pad * C1 + C1 * C2 + C2 * C3 + C1 + C2 + C3 + 4 * 16 * (max(cin_pad, C2) + C1 + 2)) bytes of LDS (<= 160 KB):
// returns -3 when the level does not fit (the caller then uses library GEMMs).
extern "C" int nirrt_pn2_sa_mlp(const float *feats, const float *xyz, const float *new_xyz, const int64_t *gidx, int B, int N, int S,
                                int K, int C, int cin_pad, const float *w1t, const float *b1, int C1, const float *w2t, const float *b2,
                                int C2, const float *w3t, const float *b3, int C3, float *out, int out_stride, int out_off, void *stream)
{
    if (C1 % 16 || C2 % 16 || C3 % 16 || C3 > 128 || cin_pad % 4 || cin_pad < C + 3 || K <= 0 || B <= 0 || S <= 0) return -1;
    const int sA = (cin_pad > C2 ? cin_pad : C2) + 1, sB = C1 + 1;
    const size_t lds = sizeof(float) * ((size_t)cin_pad * C1 + (size_t)C1 * C2 + (size_t)C2 * C3 + C1 + C2 + C3 +
                                        (size_t)SA_WAVES * 16 * (sA + sB));
    if (lds > 160 * 1024) return -3;
    // the attribute belongs to the CURRENT device's copy of the kernel (a process may drive several GPUs): set on every call,
    // it is a host-side table write
    if (hipFuncSetAttribute((const void *)k_sa_mlp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    SaMlpArgs a;
    a.feats = feats; a.xyz = xyz; a.new_xyz = new_xyz; a.gidx = (const long long *)gidx; a.out = out;
    a.w1t = w1t; a.b1 = b1; a.w2t = w2t; a.b2 = b2; a.w3t = w3t; a.b3 = b3;
    a.B = B; a.N = N; a.S = S; a.K = K; a.C = C; a.Cin_pad = cin_pad; a.C1 = C1; a.C2 = C2; a.C3 = C3;
    a.out_stride = out_stride; a.out_off = out_off;
    const long long groups = (long long)B * S;
    long long grid = (groups + SA_WAVES - 1) / SA_WAVES;
    if (grid > 2048) grid = 2048;      // grid-stride over the groups: the weights are staged once per workgroup
    hipLaunchKernelGGL(k_sa_mlp, dim3((unsigned)grid), dim3(64 * SA_WAVES), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
