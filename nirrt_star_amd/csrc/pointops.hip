// pointops.hip — PointNet++ point operators for the guidance sampler (gfx950), C ABI, raw device
// pointers (torch tensors' data_ptr) + an explicit HIP stream.  Semantics follow the reference's
// torch code (pointnet_pointnet2/models/pointnet2_utils.py): farthest_point_sample :65-86,
// query_ball_point :89-109 (with square_distance :21-42), 3-NN of PointNetFeaturePropagation :295-299.
// float32 arithmetic, -ffp-contract=off; the dot products of square_distance are a forward FMA chain.
#include <mutex>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// glibc 2.35's sin / cos (restated; every lane its own argument): the 3D ellipsoid candidates of the guidance clouds
#define GLIBC_NS glibc235v
#define GLIBC_UNIFORM 0
#include "glibc235_device.inc"
#undef GLIBC_NS
#undef GLIBC_UNIFORM

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
// nv != nullptr (ragged batch): cloud b holds nv[b] <= N points in its N-point row of xyz, the rest of the row is padding that
// is never looked at - the picks are those of a launch over that cloud alone.
template <int PPL, int NW>
__global__ __launch_bounds__(64 * NW) void k_fps_wave(const float *__restrict__ xyz, int N_stride, int S, const long long *__restrict__ start,
                                                     long long *__restrict__ out, const int *__restrict__ nv)
{
    constexpr int NT = 64 * NW;
    __shared__ float lp[NT * PPL * 3];
    __shared__ unsigned long long slot[2][NW];
    const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
    const float *p = xyz + (size_t)b * N_stride * 3;
    const int N = nv ? nv[b] : N_stride;
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
// (nv != nullptr: only the first nv[b] points of cloud b's N-point row exist, see k_fps_wave)
__global__ __launch_bounds__(256) void k_ball_query(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N_stride, int S,
                                                   int K, float r2, long long *__restrict__ out, const int *__restrict__ nv)
{
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // global query id over B*S
    const int b = (int)(q / S);
    const float *p = xyz + (size_t)b * N_stride * 3;
    const int N = nv ? nv[b] : N_stride;
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

// n_valid (DEVICE, (B,) int32, or NULL): ragged batch - cloud b has n_valid[b] <= N points, S <= n_valid[b] (N <= 2048 then)
extern "C" int nirrt_pn2_fps_ragged(const float *xyz, int B, int N, int S, const int64_t *start, const int32_t *n_valid, int64_t *out, void *stream)
{
    if (N > FPS_NT * FPS_MAX_PER_THREAD || N <= 0 || S <= 0 || (n_valid && N > 2048)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const long long *sp = (const long long *)start;
    long long *op = (long long *)out;
    const int *nv = (const int *)n_valid;
    if (N <= 64) hipLaunchKernelGGL((k_fps_wave<1, 1>), dim3(B), dim3(64), 0, st, xyz, N, S, sp, op, nv);
    else if (N <= 256) hipLaunchKernelGGL((k_fps_wave<4, 1>), dim3(B), dim3(64), 0, st, xyz, N, S, sp, op, nv);
    else if (N <= 1024) hipLaunchKernelGGL((k_fps_wave<4, 4>), dim3(B), dim3(256), 0, st, xyz, N, S, sp, op, nv);
    else if (N <= 2048) hipLaunchKernelGGL((k_fps_wave<8, 4>), dim3(B), dim3(256), 0, st, xyz, N, S, sp, op, nv);
    else {
        size_t lds = sizeof(float) * 3 * (size_t)N + 16 * sizeof(float) + 16 * sizeof(int);
        hipLaunchKernelGGL(k_fps, dim3(B), dim3(FPS_NT), lds, st, xyz, N, S, sp, op);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_fps(const float *xyz, int B, int N, int S, const int64_t *start, int64_t *out, void *stream)
{
    return nirrt_pn2_fps_ragged(xyz, B, N, S, start, nullptr, out, stream);
}

extern "C" int nirrt_pn2_ball_query_ragged(const float *xyz, const float *new_xyz, int B, int N, int S, int K, float r2,
                                           const int32_t *n_valid, int64_t *out, void *stream)
{
    long long queries = (long long)B * S;
    if (queries % 4 != 0) return -1;   // S is a multiple of 16 in this network
    hipLaunchKernelGGL(k_ball_query, dim3((unsigned)(queries / 4)), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, S, K, r2,
                       (long long *)out, (const int *)n_valid);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, int K, float r2, int64_t *out,
                                    void *stream)
{
    return nirrt_pn2_ball_query_ragged(xyz, new_xyz, B, N, S, K, r2, nullptr, out, stream);
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
#define FPS64_NT 512              // 8 waves at <= 96 VGPRs: a workgroup fits NEXT to the 8 - 16 one-wave trees a CU holds while the
#define FPS64_MAX_PER_THREAD 32   // other half of a guided batch is inside its persistent launch (N <= 16384).  (Round 2: 1024 threads
                                  // at 128 VGPRs = a whole CU's register file - every refresh waited for the running launch to drain.)

// one workgroup per cloud: cloud b = points [off[b], off[b] + cnt[b]) of the concatenated SoA buffer
// (x of all clouds, then y, then z; `total` points in all), ns[b] survivors, sel concatenated likewise.
// open3d's farthest_point_down_sample restated: start at point 0, greedy max-min SQUARED distance in float64, first maximum
// on ties.  The running minimum distances stay in registers; has_z == 0 (planar clouds: z is all zeros) skips that column.
template <int MAXP>   // points per thread: 20 (N <= 10240, the guidance clouds' 5 x 2048 candidates) or 32
__global__ __launch_bounds__(FPS64_NT, 5) void k_fps_f64(const double *__restrict__ buf, long long total, const long long *__restrict__ off,
                                                        const int *__restrict__ cnt, const int *__restrict__ ns,
                                                        unsigned char *__restrict__ sel_all, int has_z)
{
    __shared__ double rv[FPS64_NT / 64];
    __shared__ int ri[FPS64_NT / 64];
    const long long o = off[blockIdx.x];
    const int N = cnt[blockIdx.x], S = ns[blockIdx.x];
    if (N <= S) return;   // nothing to drop (the callers keep such clouds whole)
    const double *x = buf + o, *y = buf + total + o, *z = buf + 2 * total + o;
    unsigned char *sel = sel_all + o;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double dist[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; j++) dist[j] = __builtin_inf();
    for (int i = tid; i < N; i += FPS64_NT) sel[i] = 0;
    int far = 0;
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (tid == 0) sel[far] = 1;
        const double cx = x[far], cy = y[far], cz = has_z ? z[far] : 0.;
        double bv = -1.;
        int bi = 0x7fffffff;
        // (one 32-bit byte offset per point on top of the three scalar bases: separate 64-bit addresses per point and column
        // would need more registers than the running minima themselves)
        unsigned ob = (unsigned)tid * 8u;
        asm volatile("" : "+v"(ob));   // (keeps the per-point addresses from being formed once, outside the loop over the steps, and spilled)
#pragma unroll
        for (int j = 0; j < MAXP; j++) {
            const int i = tid + j * FPS64_NT;
            if (i < N) {
                const double px = *(const double *)((const char *)x + ob), py = *(const double *)((const char *)y + ob);
                const double pz = has_z ? *(const double *)((const char *)z + ob) : 0.;
                const double dx = px - cx, dy = py - cy, dz = pz - cz;
                const double d = dx * dx + dy * dy + dz * dz;
                if (d < dist[j]) dist[j] = d;
                if (dist[j] > bv) { bv = dist[j]; bi = i; }
            }
            ob += FPS64_NT * 8u;
            // four points' loads in flight at a time: hoisting all of them above the arithmetic spills the running minima
            if ((j & 3) == 3) asm volatile("" ::: "memory");
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
        for (int i = 1; i < FPS64_NT / 64; i++) {
            double ov = rv[i];
            int oi = ri[i];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        far = bi;
    }
}

// (round 6) the same down-sampling with the cloud IN REGISTERS.  k_fps_f64 re-reads every point of the cloud from the L2 in every
// one of its S steps (N x 24 bytes: 245 KB per step and cloud, 500 MB per cloud, > 100 GB for a refresh of 256 clouds) and starts
// every step with a dependent load of the last pick's coordinates; a step took 8 us, a cloud 16 ms - and a refresh cannot take
// less than one cloud does, however few clouds are due (at pc_update_cost_ratio = 1.0: 770 refreshes per step of the bench).
// Here a thread keeps its MAXP points and their running minima in registers (N <= 512 x 20 = the 5 x 2048 candidates of a guidance
// cloud: 80 doubles per thread), a wave's winner travels through LDS together with its coordinates, and one barrier per step is
// left (two alternating LDS rows).  Same arithmetic, same order of comparisons: the pick sequence is k_fps_f64's.
// wave64 argmax of (value >= 0 as its bit pattern, lowest index on ties) without LDS traffic: four DPP steps leave every 16-lane row
// holding its winner, v_readlane pulls the four row winners into scalars (wave_max_u64's scheme with the index as a third word -
// six ds_bpermute round trips were a third of a step)
template <int CTRL>
__device__ __forceinline__ void dpp_argmax_step(unsigned long long &k, int &i)
{
    const unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
    const unsigned ol = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const unsigned oh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)oh << 32) | ol;
    const bool take = o > k || (o == k && oi < i);
    k = take ? o : k;
    i = take ? oi : i;
}

__device__ __forceinline__ void wave_argmax_u64(unsigned long long &k, int &i)
{
    dpp_argmax_step<0xB1>(k, i);    // quad_perm [1,0,3,2]
    dpp_argmax_step<0x4E>(k, i);    // quad_perm [2,3,0,1]
    dpp_argmax_step<0x141>(k, i);   // row_half_mirror
    dpp_argmax_step<0x140>(k, i);   // row_mirror
    const unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
    unsigned long long rk = 0;
    int ri = 0x7fffffff;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, row * 16) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)lo, row * 16);
        const int oi = __builtin_amdgcn_readlane(i, row * 16);
        const bool take = row == 0 || o > rk || (o == rk && oi < ri);
        rk = take ? o : rk;
        ri = take ? oi : ri;
    }
    k = rk; i = ri;
}

template <int MAXP, bool HAS_Z>
__global__ __launch_bounds__(FPS64_NT) void k_fps_f64_reg(const double *__restrict__ buf, long long total, const long long *__restrict__ off,
                                                        const int *__restrict__ cnt, const int *__restrict__ ns,
                                                        unsigned char *__restrict__ sel_all)
{
    constexpr int NWV = FPS64_NT / 64;
    __shared__ unsigned long long rk[2][NWV];
    __shared__ double rx[2][NWV], ry[2][NWV], rz[2][NWV];
    __shared__ int ri[2][NWV];
    const long long o = off[blockIdx.x];
    const int N = cnt[blockIdx.x], S = ns[blockIdx.x];
    if (N <= S) return;   // nothing to drop (the callers keep such clouds whole)
    const double *x = buf + o, *y = buf + total + o, *z = buf + 2 * total + o;
    unsigned char *sel = sel_all + o;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double px[MAXP], py[MAXP], pz[MAXP], dist[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; j++) {
        const int i = tid + j * FPS64_NT;
        const bool in = i < N;
        px[j] = in ? x[i] : 0.;
        py[j] = in ? y[i] : 0.;
        pz[j] = (HAS_Z && in) ? z[i] : 0.;
        // a slot past the end never wins: every real maximum is >= 0 > -1, and min(d, -1) stays -1.  (Its arithmetic is done all the
        // same: a branch around it - even a uniform one - ends the basic block, and the MAXP independent points of a step are what the
        // scheduler interleaves to cover the latency of each point's dependent chain)
        dist[j] = in ? __builtin_inf() : -1.;
    }
    for (int i = tid; i < N; i += FPS64_NT) sel[i] = 0;
    double cx = x[0], cy = y[0], cz = HAS_Z ? z[0] : 0.;
    int far = 0;
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (tid == 0) sel[far] = 1;
        double bv = -1.;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < MAXP; j++) {
            const double dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            const double d = dx * dx + dy * dy + dz * dz;
            dist[j] = __builtin_fmin(dist[j], d);      // (`if (d < dist) dist = d` of k_fps_f64: no NaN ever gets here)
            if (dist[j] > bv) { bv = dist[j]; bi = tid + j * FPS64_NT; }   // ascending index within the thread: the first maximum is kept
        }
        // squared distances are >= 0: their bit patterns order like the values (a thread without a point: key 0, index INT_MAX)
        unsigned long long key = bv < 0. ? 0ull : (unsigned long long)__double_as_longlong(bv);
        wave_argmax_u64(key, bi);
        // the wave's winner (a scalar pair now) goes to LDS with its coordinates.  Those are read back from the cloud with SCALAR
        // loads (bi is the same in every lane): picking them out of the owner lane's registers was a chain of 6 x MAXP conditional
        // moves per step and wave - a fifth of the step
        const int p = s & 1;
        {
            const int li = bi < N ? bi : 0;      // (a wave without a point - small clouds - reports key 0 / index INT_MAX: it never wins)
            const double wx = x[li], wy = y[li], wz = HAS_Z ? z[li] : 0.;
            if (lane == 0) { rk[p][w] = key; ri[p][w] = bi; rx[p][w] = wx; ry[p][w] = wy; rz[p][w] = wz; }
        }
        __syncthreads();
        key = rk[p][0]; bi = ri[p][0];
        int bw = 0;
#pragma unroll
        for (int i = 1; i < NWV; i++) {
            const unsigned long long ok = rk[p][i];
            const int oi = ri[p][i];
            if (ok > key || (ok == key && oi < bi)) { key = ok; bi = oi; bw = i; }
        }
        far = bi;
        cx = rx[p][bw]; cy = ry[p][bw]; cz = HAS_Z ? rz[p][bw] : 0.;
        // (no second barrier: the next step writes the OTHER row, and nobody gets two steps ahead of a wave that still reads this one)
    }
}

struct FpsScratch { void *p = nullptr; size_t cap = 0; };
// NIRRT_FPS64_REG=0: the L2-streaming kernel of rounds 2 - 5 for clouds of up to 10240 points too (A/B and the parity test of the two)
static bool fps64_in_registers()
{
    const char *e = getenv("NIRRT_FPS64_REG");
    return !(e && *e == '0');
}
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
        if (cnt[b] <= 0 || num_samples[b] <= 0 || num_samples[b] > cnt[b] || cnt[b] > FPS64_NT * FPS64_MAX_PER_THREAD) { free(h_off); return -1; }
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
        int nmax = 0;
        for (int b = 0; b < n_clouds; b++) nmax = cnt[b] > nmax ? cnt[b] : nmax;
        if (nmax <= 20 * FPS64_NT && fps64_in_registers())
            hipLaunchKernelGGL((k_fps_f64_reg<20, true>), dim3(n_clouds), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                               (const int *)d_cnt, (const int *)d_ns, ds);
        else if (nmax <= 20 * FPS64_NT)
            hipLaunchKernelGGL(k_fps_f64<20>, dim3(n_clouds), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                               (const int *)d_cnt, (const int *)d_ns, ds, 1);
        else
            hipLaunchKernelGGL(k_fps_f64<32>, dim3(n_clouds), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                               (const int *)d_cnt, (const int *)d_ns, ds, 1);
        if (hipGetLastError() != hipSuccess || hipMemcpy(sel, ds, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) rc = -2;
    }
    free(h);
    free(h_off);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// nirrt_libm_probe: the restated glibc routines (csrc/glibc235_libm.inc) evaluated on the device for arguments the CALLER chooses -
// the Python binding compares them with the host's own math.atan2 / math.sin / math.cos before it trusts the device-side steer
// (a host whose libm is not glibc 2.35 / x86-64 / FMA computes other last bits in the REFERENCE: nirrt_star_amd/_hip.libm_check)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_libm_probe(int fn, long long n, const double *a, const double *b, double *out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double r;
    if (fn == 0) r = glibc235v::glibc_atan2(a[i], b[i]);
    else if (fn == 1) r = glibc235v::sin_fn(a[i]);
    else r = glibc235v::cos_fn(a[i]);
    out[i] = r;
}

extern "C" int nirrt_libm_probe(int32_t fn, int64_t n, const double *a, const double *b, double *out, int device_id)
{
    if (fn < 0 || fn > 2 || n < 0 || (n > 0 && (!a || !out || (fn == 0 && !b)))) return -1;
    if (n == 0) return 0;
    if (hipSetDevice(device_id) != hipSuccess) { (void)hipGetLastError(); return -4; }
    double *d = nullptr;
    if (hipMalloc(&d, sizeof(double) * 3 * (size_t)n) != hipSuccess) { (void)hipGetLastError(); return -2; }
    int rc = 0;
    // test hook (tests/test_libm_check_gpu.py): NIRRT_LIBM_PROBE_FLIP=1 flips the lowest bit of one word of the sin / cos table for
    // the duration of the probe - what a table that does not belong to the host's libm looks like to the caller
    const char *flip = getenv("NIRRT_LIBM_PROBE_FLIP");
    const bool do_flip = flip && *flip && *flip != '0';
    uint64_t saved[440];
    if (do_flip) {
        if (hipMemcpyFromSymbol(saved, HIP_SYMBOL(glibc235v::T_sincos), sizeof(saved)) != hipSuccess) rc = -2;
        if (!rc) {
            uint64_t bad[440];
            for (int i = 0; i < 440; i++) bad[i] = saved[i] ^ ((i % 4 == 0) ? 1ull : 0ull);   // the leading word of every entry
            if (hipMemcpyToSymbol(HIP_SYMBOL(glibc235v::T_sincos), bad, sizeof(bad)) != hipSuccess) rc = -2;
        }
    }
    if (!rc && hipMemcpy(d, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = -2;
    if (!rc && fn == 0 && hipMemcpy(d + n, b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = -2;
    if (!rc) {
        hipLaunchKernelGGL(k_libm_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (int)fn, (long long)n, (const double *)d,
                           (const double *)(d + n), d + 2 * n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d + 2 * n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = -2;
    }
    if (do_flip && hipMemcpyToSymbol(HIP_SYMBOL(glibc235v::T_sincos), saved, sizeof(saved)) != hipSuccess) rc = -2;
    (void)hipFree(d);
    if (rc == -2) (void)hipGetLastError();
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
// (axis) = sqrt(x*x + y*y), astype(int) = truncation.  The 3D ellipsoid candidates (point_cloud_mask_utils_3d.py:132-200) go
// through sin / cos: the device's own (OCML), so they agree with the host's to a few ulp, not bit for bit - like the informed
// samples of the 3D loop.
// ------------------------------------------------------------------------------------------------
struct nirrt_cloud_job {
    const unsigned *words;          // DEVICE: generator outputs from the problem's current position (>= 2 * n_doubles of them)
    const unsigned char *free_tab;  // DEVICE, 2D: (h + 1) x (w + 1) table "the 2 x 2 pixel block around this integer position is free"
    const double *balls;            // DEVICE, 3D: (n_ball, 4) cx, cy, cz, r
    const double *boxes;            // DEVICE, 3D: (n_box, 6) x, y, z, w, h, d
    int mode;                       // 0: whole image (2D), 1: ellipse (2D), 2: whole box (3D), 3: ellipsoid (3D)
    int w, h, n_ball, n_box, pad;
    double a[20];                   // mode 0: w, h; mode 1: CL00, CL01, CL10, CL11, cx, cy; mode 2: lo[3], hi - lo [3];
                                    // mode 3: C.L row-major [9], x_center [3], range lo [3], range hi [3]
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
            if (jb.mode >= 2) {
                bool inside = false;
                if (jb.mode == 2) {
                    x = jb.a[0] + jb.a[3] * mt_double(jb.words, 3ll * i);
                    y = jb.a[1] + jb.a[4] * mt_double(jb.words, 3ll * i + 1);
                    z = jb.a[2] + jb.a[5] * mt_double(jb.words, 3ll * i + 2);
                } else {
                    // rng.uniform(0, 1, n), rng.uniform(0, pi, n), rng.uniform(0, 2 pi, n): three arrays drawn one after the other;
                    // spherical coordinates -> unit ball -> (C.L) . s + x_center, then the range test
                    const double PI = 3.141592653589793;
                    const double rr = 0.0 + (1.0 - 0.0) * mt_double(jb.words, (long long)i);
                    const double th = 0.0 + (PI - 0.0) * mt_double(jb.words, (long long)n_raw + i);
                    const double ph = 0.0 + (2 * PI - 0.0) * mt_double(jb.words, 2ll * n_raw + i);
                    // (np.sin / np.cos of numpy 2.2 are libm's for float64: the restated glibc functions, one argument per lane)
                    const double st = glibc235v::sin_fn(th), ct = glibc235v::cos_fn(th);
                    const double sp = glibc235v::sin_fn(ph), cp = glibc235v::cos_fn(ph);
                    const double s0 = rr * st * cp, s1 = rr * st * sp, s2 = rr * ct;
                    x = __builtin_fma(jb.a[2], s2, __builtin_fma(jb.a[1], s1, jb.a[0] * s0)) + jb.a[9];
                    y = __builtin_fma(jb.a[5], s2, __builtin_fma(jb.a[4], s1, jb.a[3] * s0)) + jb.a[10];
                    z = __builtin_fma(jb.a[8], s2, __builtin_fma(jb.a[7], s1, jb.a[6] * s0)) + jb.a[11];
                    inside = !(jb.a[12] <= x && x <= jb.a[15] && jb.a[13] <= y && y <= jb.a[16] && jb.a[14] <= z && z <= jb.a[17]);
                }
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
    if (!jobs || !clouds || !n_cand || !n_out || n_jobs <= 0 || n_raw <= 0 || n_points <= 0 || n_raw > FPS64_NT * FPS64_MAX_PER_THREAD) return -1;
    for (int b = 0; b < n_jobs; b++)   // one batch = one dimensionality (the down-sampling kernel takes has_z for the whole launch)
        if (jobs[b].mode < 0 || jobs[b].mode > 3 || (jobs[b].mode >= 2) != (jobs[0].mode >= 2)) return -1;
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
    std::vector<long long> h_off((size_t)n_jobs);
    std::vector<int> h_ns((size_t)n_jobs, n_points);
    for (int b = 0; b < n_jobs; b++) h_off[b] = (long long)b * n_raw;
    int rc = 0;
    if (hipMemcpy(d_jobs, jobs, sizeof(nirrt_cloud_job) * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_off, h_off.data(), sizeof(long long) * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ns, h_ns.data(), sizeof(int) * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess)
        rc = -2;
    if (!rc) {
        // The three kernels run on the NULL stream on purpose: the generator outputs they read are complete (nirrt_generator_words
        // waits for its launch), and the clouds they write are read next by torch's default stream - the legacy stream orders
        // that without events.  (With NIRRT_BATCH_GROUPS >= 2 - off by default, see batch.run_batch - the launch waits
        // for the other group's persistent kernel when the tree streams are blocking ones.)
        hipLaunchKernelGGL(k_cloud_candidates, dim3(n_jobs), dim3(CAND_NT), 0, 0, (const nirrt_cloud_job *)d_jobs, n_raw, d, total, d_cnt);
        // k_fps_f64 leaves clouds with cnt <= num_samples alone (k_cloud_compact keeps all of their points)
        if (n_raw <= 20 * FPS64_NT && fps64_in_registers()) {
            if (jobs[0].mode >= 2)
                hipLaunchKernelGGL((k_fps_f64_reg<20, true>), dim3(n_jobs), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                                   (const int *)d_cnt, (const int *)d_ns, ds);
            else
                hipLaunchKernelGGL((k_fps_f64_reg<20, false>), dim3(n_jobs), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                                   (const int *)d_cnt, (const int *)d_ns, ds);
        } else if (n_raw <= 20 * FPS64_NT)
            hipLaunchKernelGGL(k_fps_f64<20>, dim3(n_jobs), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                               (const int *)d_cnt, (const int *)d_ns, ds, jobs[0].mode >= 2 ? 1 : 0);
        else
            hipLaunchKernelGGL(k_fps_f64<32>, dim3(n_jobs), dim3(FPS64_NT), 0, 0, (const double *)d, total, (const long long *)d_off,
                               (const int *)d_cnt, (const int *)d_ns, ds, jobs[0].mode >= 2 ? 1 : 0);
        hipLaunchKernelGGL(k_cloud_compact, dim3(n_jobs), dim3(CAND_NT), 0, 0, (const double *)d, total, n_raw, (const int *)d_cnt, n_points,
                           (const unsigned char *)ds, clouds, d_nout);
        if (hipGetLastError() != hipSuccess || hipMemcpy(n_cand, d_cnt, sizeof(int) * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(n_out, d_nout, sizeof(int) * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess)
            rc = -2;
    }
    return rc;
}

typedef float float4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// The network's input block of a guidance cloud (pointnet2_wrapper.py:43-58), assembled where the cloud already is: xyz in
// float32 shifted to the centroid and scaled by the largest norm (pc_normalize, pointnet2_utils.py:13-18), the start / goal
// indicator channels (get_point_cloud_mask_around_points, point_cloud_mask_utils.py:20-31: float64 distance strictly inside
// the radius) and the "neither" channel.  The arithmetic follows numpy's: np.mean over axis 0 of an (N, 3) float32 array adds
// the rows in order (one thread per coordinate walks the cloud), the norm is (x^2 + y^2) + z^2, no contraction.
// One workgroup per cloud; rows[] picks the clouds of one size n out of the batch's (n_clouds, stride_pts, 3) f64 block.
// ------------------------------------------------------------------------------------------------
#define NI_NT 256
// smask / gmask != nullptr: the indicator channels are given (one byte per point, row `row` at + row * mask_stride) - the masks of
// the neural-connect rounds are re-seeded at boundary points (k_connect_masks) instead of following from the start / goal states
// n_each != nullptr (ragged batch): cloud b has n_each[b] <= n_in points; the blocks are n_in wide all the same (the caller zeroes
// the columns behind a cloud's own points) and everything that depends on the cloud's size (mean, largest norm) uses n_each[b]
__global__ __launch_bounds__(NI_NT) void k_net_input(const double *__restrict__ clouds, long long stride_pts, const int *__restrict__ rows,
                                                     int n_in, const double *__restrict__ starts, const double *__restrict__ goals,
                                                     double radius, float *__restrict__ out_all, const unsigned char *__restrict__ smask,
                                                     const unsigned char *__restrict__ gmask, long long mask_stride,
                                                     const int *__restrict__ n_each)
{
    extern __shared__ float ni_lds[];          // 3 * n_in coordinates + reduction slots
    float *px = ni_lds, *py = px + n_in, *pz = py + n_in;
    const int n = n_each ? n_each[blockIdx.x] : n_in;
    __shared__ float mean[3];
    __shared__ float red[NI_NT / 64];
    const int b = blockIdx.x, row = rows[b], tid = threadIdx.x;
    const double *c = clouds + (long long)row * stride_pts * 3;
    const bool given = smask != nullptr;
    const double sx = given ? 0. : starts[3 * b], sy = given ? 0. : starts[3 * b + 1], sz = given ? 0. : starts[3 * b + 2];
    const double gx = given ? 0. : goals[3 * b], gy = given ? 0. : goals[3 * b + 1], gz = given ? 0. : goals[3 * b + 2];
    float *o = out_all + (long long)b * 6 * n_in;
    for (int i = tid; i < n; i += NI_NT) {
        const double x = c[3 * i], y = c[3 * i + 1], z = c[3 * i + 2];
        px[i] = (float)x; py[i] = (float)y; pz[i] = (float)z;
        float sm, gm;
        if (given) {
            sm = smask[(long long)row * mask_stride + i] ? 1.f : 0.f;
            gm = gmask[(long long)row * mask_stride + i] ? 1.f : 0.f;
        } else {
            double dx = x - sx, dy = y - sy, dz = z - sz;
            sm = sqrt((dx * dx + dy * dy) + dz * dz) < radius ? 1.f : 0.f;
            dx = x - gx; dy = y - gy; dz = z - gz;
            gm = sqrt((dx * dx + dy * dy) + dz * dz) < radius ? 1.f : 0.f;
        }
        o[3 * n_in + i] = sm;
        o[4 * n_in + i] = gm;
        o[5 * n_in + i] = (sm + gm) == 0.f ? 1.f : 0.f;
    }
    __syncthreads();
    if (tid < 3) {
        const float *p = tid == 0 ? px : tid == 1 ? py : pz;
        float acc = 0.f;
        for (int i = 0; i < n; i++) acc = acc + p[i];
        mean[tid] = acc / (float)n;
    }
    __syncthreads();
    const float mx = mean[0], my = mean[1], mz = mean[2];
    float big = 0.f;
    for (int i = tid; i < n; i += NI_NT) {
        const float x = px[i] - mx, y = py[i] - my, z = pz[i] - mz;
        px[i] = x; py[i] = y; pz[i] = z;
        big = fmaxf(big, sqrtf((x * x + y * y) + z * z));
    }
    for (int off = 32; off; off >>= 1) big = fmaxf(big, __shfl_xor(big, off));
    if ((tid & 63) == 0) red[tid >> 6] = big;
    __syncthreads();
    big = red[0];
    for (int w = 1; w < NI_NT / 64; w++) big = fmaxf(big, red[w]);
    for (int i = tid; i < n; i += NI_NT) {
        o[i] = px[i] / big;
        o[n_in + i] = py[i] / big;
        o[2 * n_in + i] = pz[i] / big;
    }
}

// n_each (DEVICE, (n_rows,) int32, or NULL): ragged batch - row b's cloud has n_each[b] <= n points; out is (n_rows, 6, n) and the
// caller has zeroed it (the columns behind a cloud's own points are not written)
extern "C" int nirrt_pn2_net_input_ragged(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n, const int32_t *n_each,
                                          const double *starts, const double *goals, double radius, float *out, void *stream)
{
    if (n_rows <= 0 || n <= 0 || n > 12288 || stride_pts < n) return -1;
    hipLaunchKernelGGL(k_net_input, dim3((unsigned)n_rows), dim3(NI_NT), sizeof(float) * 3 * (size_t)n, (hipStream_t)stream, clouds,
                       (long long)stride_pts, rows, n, starts, goals, radius, out, (const unsigned char *)nullptr, (const unsigned char *)nullptr, 0ll,
                       (const int *)n_each);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_net_input(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n, const double *starts,
                                   const double *goals, double radius, float *out, void *stream)
{
    return nirrt_pn2_net_input_ragged(clouds, stride_pts, rows, n_rows, n, nullptr, starts, goals, radius, out, stream);
}

extern "C" int nirrt_pn2_net_input_masks_ragged(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n,
                                                const int32_t *n_each, const uint8_t *start_masks, const uint8_t *goal_masks,
                                                int64_t mask_stride, float *out, void *stream)
{
    if (n_rows <= 0 || n <= 0 || n > 12288 || stride_pts < n || mask_stride < n || !start_masks || !goal_masks) return -1;
    hipLaunchKernelGGL(k_net_input, dim3((unsigned)n_rows), dim3(NI_NT), sizeof(float) * 3 * (size_t)n, (hipStream_t)stream, clouds,
                       (long long)stride_pts, rows, n, (const double *)nullptr, (const double *)nullptr, 0.0, out, start_masks, goal_masks,
                       (long long)mask_stride, (const int *)n_each);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_net_input_masks(const double *clouds, int64_t stride_pts, const int32_t *rows, int n_rows, int n,
                                         const uint8_t *start_masks, const uint8_t *goal_masks, int64_t mask_stride, float *out, void *stream)
{
    return nirrt_pn2_net_input_masks_ragged(clouds, stride_pts, rows, n_rows, n, nullptr, start_masks, goal_masks, mask_stride, out, stream);
}

// ------------------------------------------------------------------------------------------------
// Neural connect (wrapper/pointnet_pointnet2/pointnet2_wrapper_connect_bfs.py:76-240 with the helpers of
// wrapper/utils/bfs_connect_heuristic.py:5-29, 80-139, 142-181), one workgroup per cloud, every open cloud of a batch in one
// launch.  The reference works on the float32 cloud with numpy: np.linalg.norm(axis) of float32 differences = correctly rounded
// float32 sqrt of the float32 sum (d0^2 + d1^2) [+ d2^2], compared with the radius in float32 - restated here op by op.
//   * breadth-first search from one end over {end, other end, points predicted "path" so far}, edge <=> distance < radius: only
//     the SET reached matters (has_path <=> the other end is adjacent to the end or to a reached point; without a path the
//     reached set is the end's whole component, whatever the visiting order), so the search runs level by level in parallel;
//   * boundary = reached points with a not-predicted point strictly within the radius;
//   * next seed = the boundary point with the smallest (rank of g + h, ascending) + (rank of g, descending), first on ties of
//     that sum.  numpy's argsort is not stable: equal keys are reported (tie) and the caller repeats the choice for that cloud
//     with numpy itself;
//   * start / goal masks of the next classification = points strictly within the radius of the seeds (k_connect_masks).
// ------------------------------------------------------------------------------------------------
struct nirrt_connect_job {
    const double *cloud;        // DEVICE (n, 3) f64 (z = 0 for planar clouds)
    const unsigned char *pred;  // DEVICE (n,): this round's path_pred != 0
    unsigned char *path_mask;   // DEVICE (n,) in / out: union of the predictions so far (path_pred_mask)
    unsigned char *start_mask;  // DEVICE (n,) in / out: start / goal masks of the NEXT classification
    unsigned char *goal_mask;
    unsigned char *boundary;    // DEVICE (2, n) out: boundary masks of the two searches (start -> goal, goal -> start)
    int n, dim;
    double start[3], goal[3];
};
#define CN_NT 256
#define CN_MAX 2048

__device__ __forceinline__ float cn_dist(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return sqrtf((dx * dx + dy * dy) + dz * dz);   // (planar clouds: dz == 0 adds an exact zero)
}

__global__ __launch_bounds__(CN_NT) void k_connect_round(const nirrt_connect_job *jobs, float radius, int *has_path, int *seed_idx, int *tie)
{
    __shared__ float px[CN_MAX], py[CN_MAX], pz[CN_MAX];
    __shared__ unsigned char um[CN_MAX], vis[CN_MAX], bf[CN_MAX];
    __shared__ unsigned short fr[2][CN_MAX], bidx[CN_MAX];
    __shared__ float kg[CN_MAX], ks[CN_MAX];
    __shared__ int cnt[2], flag, best_score, best_k;
    const nirrt_connect_job jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, n = jb.n;
    for (int i = tid; i < n; i += CN_NT) {
        px[i] = (float)jb.cloud[3 * i]; py[i] = (float)jb.cloud[3 * i + 1]; pz[i] = jb.dim == 3 ? (float)jb.cloud[3 * i + 2] : 0.f;
        const unsigned char u = (unsigned char)((jb.path_mask[i] | jb.pred[i]) ? 1 : 0);   // ((mask + pred) > 0)
        um[i] = u;
        jb.path_mask[i] = u;
    }
    if (tid == 0) { has_path[blockIdx.x] = 0; seed_idx[2 * blockIdx.x] = -1; seed_idx[2 * blockIdx.x + 1] = -1; tie[2 * blockIdx.x] = 0; tie[2 * blockIdx.x + 1] = 0; }
    const float sxyz[3] = {(float)jb.start[0], (float)jb.start[1], jb.dim == 3 ? (float)jb.start[2] : 0.f};
    const float gxyz[3] = {(float)jb.goal[0], (float)jb.goal[1], jb.dim == 3 ? (float)jb.goal[2] : 0.f};
    __syncthreads();
    for (int d = 0; d < 2; d++) {
        const float ax = d == 0 ? sxyz[0] : gxyz[0], ay = d == 0 ? sxyz[1] : gxyz[1], az = d == 0 ? sxyz[2] : gxyz[2];
        const float bx = d == 0 ? gxyz[0] : sxyz[0], by = d == 0 ? gxyz[1] : sxyz[1], bz = d == 0 ? gxyz[2] : sxyz[2];
        // ---- the component of end a among the predicted points ----
        if (tid == 0) { cnt[0] = 0; cnt[1] = 0; flag = cn_dist(ax, ay, az, bx, by, bz) < radius ? 1 : 0; }
        __syncthreads();
        for (int i = tid; i < n; i += CN_NT) {
            const bool hit = um[i] && cn_dist(px[i], py[i], pz[i], ax, ay, az) < radius;
            vis[i] = hit ? 1 : 0;
            if (hit) fr[0][atomicAdd(&cnt[0], 1)] = (unsigned short)i;
        }
        __syncthreads();
        int cur = 0;
        for (;;) {
            const int nf = cnt[cur];
            if (nf == 0) break;
            for (int i = tid; i < n; i += CN_NT) {
                if (um[i] && !vis[i]) {
                    const float x = px[i], y = py[i], z = pz[i];
                    bool hit = false;
                    for (int f = 0; f < nf && !hit; f++) {
                        const int q = fr[cur][f];
                        hit = cn_dist(x, y, z, px[q], py[q], pz[q]) < radius;
                    }
                    if (hit) { vis[i] = 1; fr[cur ^ 1][atomicAdd(&cnt[cur ^ 1], 1)] = (unsigned short)i; }
                }
            }
            __syncthreads();
            if (tid == 0) cnt[cur] = 0;
            cur ^= 1;
            __syncthreads();
        }
        // has_path: the other end is adjacent to this end or to a reached point
        for (int i = tid; i < n; i += CN_NT)
            if (vis[i] && cn_dist(px[i], py[i], pz[i], bx, by, bz) < radius) flag = 1;
        __syncthreads();
        if (flag) {   // (uniform) connected: the rounds end here (pointnet2_wrapper_connect_bfs.py:178-179)
            if (tid == 0) has_path[blockIdx.x] = 1;
            return;
        }
        // ---- boundary: reached points with a not-predicted point strictly within the radius ----
        for (int i = tid; i < n; i += CN_NT) {
            bool b = false;
            if (vis[i]) {
                const float x = px[i], y = py[i], z = pz[i];
                for (int q = 0; q < n && !b; q++) b = !um[q] && cn_dist(x, y, z, px[q], py[q], pz[q]) < radius;
            }
            bf[i] = b ? 1 : 0;
            jb.boundary[(size_t)d * n + i] = b ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) {   // boundary points in index order (np.where)
            int nb = 0;
            for (int i = 0; i < n; i++) if (bf[i]) bidx[nb++] = (unsigned short)i;
            cnt[1] = nb; best_score = 0x7fffffff; best_k = 0x7fffffff;
        }
        __syncthreads();
        const int nb = cnt[1];
        if (nb > 0) {
            // g = |b - a|, h = |b - other end| (float32), key = g + h
            for (int k = tid; k < nb; k += CN_NT) {
                const int i = bidx[k];
                const float g = cn_dist(px[i], py[i], pz[i], ax, ay, az), h = cn_dist(px[i], py[i], pz[i], bx, by, bz);
                kg[k] = g; ks[k] = g + h;
            }
            __syncthreads();
            bool any_tie = false;
            for (int k = tid; k < nb; k += CN_NT) {
                const float g = kg[k], s_ = ks[k];
                int tr = 0, gr = 0;
                for (int j = 0; j < nb; j++) {
                    const float gj = kg[j], sj = ks[j];
                    tr += sj < s_ ? 1 : 0;
                    gr += gj > g ? 1 : 0;
                    if (j != k && (sj == s_ || gj == g)) any_tie = true;
                }
                atomicMin(&best_score, tr + gr);
                fr[0][k] = (unsigned short)(tr + gr);   // (the frontier lists are dead; the keys are still being read by other threads)
            }
            __syncthreads();
            for (int k = tid; k < nb; k += CN_NT)
                if ((int)fr[0][k] == best_score) atomicMin(&best_k, k);   // np.argmax: the first maximum of -(rank sum)
            if (any_tie) tie[2 * blockIdx.x + d] = 1;
            __syncthreads();
            if (tid == 0) seed_idx[2 * blockIdx.x + d] = bidx[best_k];
        }
        __syncthreads();
    }
}

// start / goal masks of the next classification: seed -2 = the start / goal state itself (the masks before the first round), -1 =
// keep the mask (no boundary point: pointnet2_wrapper_connect_bfs.py:181-183), >= 0: that cloud point
__global__ __launch_bounds__(CN_NT) void k_connect_masks(const nirrt_connect_job *jobs, float radius, const int *seed_idx)
{
    const nirrt_connect_job jb = jobs[blockIdx.x];
    for (int d = 0; d < 2; d++) {
        const int sd = seed_idx[2 * blockIdx.x + d];
        if (sd == -1) continue;
        float ax, ay, az;
        if (sd == -2) {
            const double *p = d == 0 ? jb.start : jb.goal;
            ax = (float)p[0]; ay = (float)p[1]; az = jb.dim == 3 ? (float)p[2] : 0.f;
        } else {
            ax = (float)jb.cloud[3 * sd]; ay = (float)jb.cloud[3 * sd + 1]; az = jb.dim == 3 ? (float)jb.cloud[3 * sd + 2] : 0.f;
        }
        unsigned char *m = d == 0 ? jb.start_mask : jb.goal_mask;
        for (int i = threadIdx.x; i < jb.n; i += CN_NT) {
            const float x = (float)jb.cloud[3 * i], y = (float)jb.cloud[3 * i + 1], z = jb.dim == 3 ? (float)jb.cloud[3 * i + 2] : 0.f;
            m[i] = cn_dist(x, y, z, ax, ay, az) < radius ? 1 : 0;
        }
    }
}

static FpsScratch g_connect_scratch[16];
static int connect_stage(const nirrt_connect_job *jobs, int n_jobs, int n_ints, int device_id, nirrt_connect_job **d_jobs, int **d_ints)
{
    if (!jobs || n_jobs <= 0) return -1;
    for (int b = 0; b < n_jobs; b++)
        if (jobs[b].n <= 0 || jobs[b].n > CN_MAX || (jobs[b].dim != 2 && jobs[b].dim != 3)) return -1;
    if (hipSetDevice(device_id) != hipSuccess) return -4;
    const size_t b_job = (sizeof(nirrt_connect_job) * (size_t)n_jobs + 255) & ~(size_t)255;
    const size_t need = b_job + sizeof(int) * (size_t)n_ints * (size_t)n_jobs + 256;
    FpsScratch &sc = g_connect_scratch[device_id & 15];
    if (sc.cap < need) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.cap = 0;
        if (hipMalloc(&sc.p, 2 * need) != hipSuccess) return -2;
        sc.cap = 2 * need;
    }
    *d_jobs = (nirrt_connect_job *)sc.p;
    *d_ints = (int *)((char *)sc.p + b_job);
    return hipMemcpy(*d_jobs, jobs, sizeof(nirrt_connect_job) * (size_t)n_jobs, hipMemcpyHostToDevice) == hipSuccess ? 0 : -2;
}

// one neural-connect round for n_jobs clouds (their new predictions in jobs[].pred): has_path (n_jobs,), seed_idx (n_jobs, 2) and
// tie (n_jobs, 2) are HOST outputs; the boundary masks stay on the device (jobs[].boundary)
extern "C" int nirrt_connect_round(const nirrt_connect_job *jobs, int n_jobs, double radius, int32_t *has_path, int32_t *seed_idx,
                                   int32_t *tie, int device_id)
{
    if (!has_path || !seed_idx || !tie) return -1;
    std::lock_guard<std::mutex> hold(g_fps_mu);
    nirrt_connect_job *d_jobs = nullptr;
    int *d = nullptr;
    int rc = connect_stage(jobs, n_jobs, 5, device_id, &d_jobs, &d);
    if (rc) return rc;
    int *d_has = d, *d_seed = d + n_jobs, *d_tie = d + 3 * n_jobs;
    hipLaunchKernelGGL(k_connect_round, dim3(n_jobs), dim3(CN_NT), 0, 0, (const nirrt_connect_job *)d_jobs, (float)radius, d_has, d_seed, d_tie);
    if (hipGetLastError() != hipSuccess || hipMemcpy(has_path, d_has, sizeof(int) * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(seed_idx, d_seed, sizeof(int) * 2 * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(tie, d_tie, sizeof(int) * 2 * (size_t)n_jobs, hipMemcpyDeviceToHost) != hipSuccess)
        return -2;
    return 0;
}

// start / goal masks from seeds (HOST (n_jobs, 2): -2 start / goal state, -1 keep, >= 0 cloud point)
extern "C" int nirrt_connect_masks(const nirrt_connect_job *jobs, int n_jobs, double radius, const int32_t *seed_idx, int device_id)
{
    if (!seed_idx) return -1;
    std::lock_guard<std::mutex> hold(g_fps_mu);
    nirrt_connect_job *d_jobs = nullptr;
    int *d = nullptr;
    int rc = connect_stage(jobs, n_jobs, 2, device_id, &d_jobs, &d);
    if (rc) return rc;
    for (int b = 0; b < n_jobs; b++)
        for (int k = 0; k < 2; k++)
            if (seed_idx[2 * b + k] < -2 || seed_idx[2 * b + k] >= jobs[b].n) return -1;
    if (hipMemcpy(d, seed_idx, sizeof(int) * 2 * (size_t)n_jobs, hipMemcpyHostToDevice) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_connect_masks, dim3(n_jobs), dim3(CN_NT), 0, 0, (const nirrt_connect_job *)d_jobs, (float)radius, (const int *)d);
    return hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Input rows of the levels whose MLP runs as library GEMMs (SA3, SA4, every feature-propagation level): ONE pass that writes
// the GEMM's A matrix, instead of torch gathers + multiply + sum + cat (five kernels and four intermediate tensors per level).
// float4 throughout: channel counts are multiples of 4 (the host checks and otherwise keeps the torch path).
// ------------------------------------------------------------------------------------------------

// grouped rows (pointnet2_utils.py:247-250): out[(b, s, k), :] = [feats[b, gidx[b, s, k], 0:C], xyz[b, gidx] - new_xyz[b, s], 0]
// (C + 4 columns: the zero column keeps the rows 16-byte aligned; the first layer's weight gets a zero column to match)
__global__ __launch_bounds__(256) void k_group_rows(const float4_t *__restrict__ feats, const float *__restrict__ xyz,
                                                    const float *__restrict__ new_xyz, const long long *__restrict__ gidx, int N, int S,
                                                    int K, int C4, long long rows, float4_t *__restrict__ out)
{
    const int W4 = C4 + 1;
    const long long total = rows * W4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / W4;
        const int c = (int)(e - row * W4);
        const long long g = row / K;                       // (b, s)
        const long long src = (g / S) * N + gidx[row];
        float4_t v;
        if (c < C4) v = feats[src * C4 + c];
        else {
            v[0] = xyz[src * 3] - new_xyz[g * 3];
            v[1] = xyz[src * 3 + 1] - new_xyz[g * 3 + 1];
            v[2] = xyz[src * 3 + 2] - new_xyz[g * 3 + 2];
            v[3] = 0.f;
        }
        out[e] = v;
    }
}

// feature-propagation rows (pointnet2_utils.py:295-309): out[(b, n), :] = [feats1[b, n, 0:C1], sum_j w_j feats2[b, idx[b, n, j], 0:C2]]
// with w_j = (1 / (d_j + 1e-8)) / sum_j (1 / (d_j + 1e-8)), the three terms added in order j = 0, 1, 2 (no contraction: -ffp-contract=off)
__global__ __launch_bounds__(256) void k_fp_rows(const float4_t *__restrict__ feats1, const float4_t *__restrict__ feats2,
                                                 const float *__restrict__ dist, const long long *__restrict__ idx, int N, int S, int C14,
                                                 int C24, long long rows, float4_t *__restrict__ out)
{
    const int W4 = C14 + C24;
    const long long total = rows * W4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / W4;
        const int c = (int)(e - row * W4);
        float4_t v;
        if (c < C14) v = feats1[row * C14 + c];
        else {
            const float r0 = 1.0f / (dist[row * 3] + 1e-8f), r1 = 1.0f / (dist[row * 3 + 1] + 1e-8f), r2 = 1.0f / (dist[row * 3 + 2] + 1e-8f);
            const float norm = (r0 + r1) + r2;
            const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
            const long long b = row / N;
            const int cc = c - C14;
            const float4_t a0 = feats2[(b * S + idx[row * 3]) * C24 + cc];
            const float4_t a1 = feats2[(b * S + idx[row * 3 + 1]) * C24 + cc];
            const float4_t a2 = feats2[(b * S + idx[row * 3 + 2]) * C24 + cc];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = (a0[u] * w0 + a1[u] * w1) + a2[u] * w2;
        }
        out[e] = v;
    }
}

extern "C" int nirrt_pn2_group_rows(const float *feats, const float *xyz, const float *new_xyz, const int64_t *gidx, int B, int N, int S,
                                    int K, int C, float *out, void *stream)
{
    if (B <= 0 || S <= 0 || K <= 0 || C <= 0 || C % 4 || ((uintptr_t)feats | (uintptr_t)out) % 16) return -1;
    const long long rows = (long long)B * S * K, total = rows * (C / 4 + 1);
    long long grid = (total + 255) / 256;
    if (grid > 256 * 64) grid = 256 * 64;
    hipLaunchKernelGGL(k_group_rows, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float4_t *)feats, xyz, new_xyz,
                       (const long long *)gidx, N, S, K, C / 4, rows, (float4_t *)out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nirrt_pn2_fp_rows(const float *feats1, const float *feats2, const float *dist, const int64_t *idx, int B, int N, int S,
                                 int C1, int C2, float *out, void *stream)
{
    if (B <= 0 || N <= 0 || S <= 0 || C2 <= 0 || C1 < 0 || C1 % 4 || C2 % 4 || ((uintptr_t)feats1 | (uintptr_t)feats2 | (uintptr_t)out) % 16)
        return -1;
    const long long rows = (long long)B * N, total = rows * ((C1 + C2) / 4);
    long long grid = (total + 255) / 256;
    if (grid > 256 * 64) grid = 256 * 64;
    hipLaunchKernelGGL(k_fp_rows, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float4_t *)feats1,
                       (const float4_t *)feats2, dist, (const long long *)idx, N, S, C1 / 4, C2 / 4, rows, (float4_t *)out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
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

// one layer on the wave's two tiles: act (32 rows x stride sa floats, K_ = 16 * KC4 inputs) -> C_ outputs; LAST: maximum over
// each group's rows instead of a new activation tile.  KC4 (float4 operands per lane) is a template parameter so that the loop
// over k is straight-line code: with a run-time bound every ds_read_b128 sat in its own basic block right in front of the eight
// MFMAs it feeds and its latency was paid every time.  The weight fragments of column tile ct + 16 are read while tile ct
// multiplies (two register sets, ping-pong).
// maximum over the 16 lanes of a DPP row; valid in lane 15 of the row (row_shr 1, 2, 4, 8 with the lanes shifted in from outside
// the row keeping their own value)
__device__ __forceinline__ float row_max16(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x111, 0xf, 0xf, false)));   // row_shr:1
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x112, 0xf, 0xf, false)));   // row_shr:2
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x114, 0xf, 0xf, false)));   // row_shr:4
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x118, 0xf, 0xf, false)));   // row_shr:8
    return v;
}

template <bool LAST, int KC4>
__device__ __forceinline__ void sa_layer(float *act, int sa, const float *W, const float *bias, int C_, int lane, int gsz)
{
    constexpr int K_ = 16 * KC4, kc = 4 * KC4, sw = K_ + 4;
    const int row = lane & 15, kq = lane >> 4;                 // this lane's k range: [kq * kc, (kq + 1) * kc)
    float4_t a0[KC4], a1[KC4];
#pragma unroll
    for (int j = 0; j < KC4; j++) {
        a0[j] = *reinterpret_cast<const float4_t *>(act + row * sa + kq * kc + 4 * j);
        a1[j] = *reinterpret_cast<const float4_t *>(act + (16 + row) * sa + kq * kc + 4 * j);
    }
    const float *wbase = W + row * sw + kq * kc;
    // fragments + bias of a column tile; `ct` is clamped so that the read-ahead past the last tile stays inside the layer (no
    // branch between two tiles: the compiler may then run the epilogue of one under the MFMAs of the next)
    // The product is formed TRANSPOSED: D^T (16 channels x 16 rows) = W (16 x 4) . A^T (4 x 16) - the weight fragment goes in as the
    // first operand, the activations as the second (the register contents are the same as for A . W^T: lane l holds
    // W[ct + l % 16][k(l / 16, step)] and A[l % 16][k(l / 16, step)]).  Lane l then holds FOUR CONSECUTIVE CHANNELS
    // ct + 4 * (l / 16) + v of row l % 16: the next layer's input row is written with ONE ds_write_b128 per tile instead of four
    // scattered ds_write_b32 (round 3: 39 % of the LDS-active cycles were bank conflicts of that epilogue), and the bias is a vector.
    auto wload = [&](float4_t (&wf)[KC4], float4_t &bv, int ct) {
        ct = ct < C_ ? ct : C_ - 16;
#pragma unroll
        for (int j = 0; j < KC4; j++) wf[j] = *reinterpret_cast<const float4_t *>(wbase + ct * sw + 4 * j);
        if (!LAST) bv = *reinterpret_cast<const float4_t *>(bias + ct + 4 * kq);
        else { const float b = bias[ct + row]; bv[0] = b; bv[1] = b; bv[2] = b; bv[3] = b; }   // (last layer: not transposed, see epi)
    };
    auto mm = [&](const float4_t (&wf)[KC4], const float4_t &bv, float4_t &acc0, float4_t &acc1) {
        acc0 = bv;
        acc1 = bv;
#pragma unroll
        for (int j = 0; j < KC4; j++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!LAST) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][u], a0[j][u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][u], a1[j][u], acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j][u], wf[j][u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j][u], wf[j][u], acc1, 0, 0, 0);
                }
            }
        }
    };
    auto epi = [&](const float4_t &acc0, const float4_t &acc1, int ct) {
        if (!LAST) {
            float4_t r0, r1;
#pragma unroll
            for (int v = 0; v < 4; v++) { r0[v] = acc0[v] > 0.f ? acc0[v] : 0.f; r1[v] = acc1[v] > 0.f ? acc1[v] : 0.f; }
            // (same wave reads and later writes `act`: a0 / a1 were loaded above and LDS operations of a wave complete in order)
            *reinterpret_cast<float4_t *>(act + row * sa + ct + 4 * kq) = r0;
            *reinterpret_cast<float4_t *>(act + (16 + row) * sa + ct + 4 * kq) = r1;
        } else {
            // the last layer is NOT transposed (lane l holds rows 4 * (l / 16) + v of column l % 16): the maximum over the rows is a
            // maximum over this lane's four registers (ReLU first); the four lanes of a column leave their partial maxima in
            // rows kq (first tile) and 16 + kq (second tile) of the - by now dead - activation tile, the caller folds them.  (With
            // the transposed product the maximum runs across 16 lanes: 32 DPP steps per column tile pair - measured slower.)
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int v = 0; v < 4; v++) { m0 = fmaxf(m0, acc0[v]); m1 = fmaxf(m1, acc1[v]); }
            const int ms = C_ + 16;                   // 4 * ms <= 16 * sa: the caller sizes sa >= C3 / 4 + 4
            act[kq * ms + ct + row] = m0;
            act[16 * sa + kq * ms + ct + row] = m1;
        }
    };
    // Two tiles per trip; the fragments of the NEXT trip are read at the end of this one (the scheduler sinks LDS reads towards
    // their first use - placed anywhere earlier they ended up one by one, each with its own s_waitcnt, inside the MFMA stream of
    // the tile that needed them; it cannot sink them across the back edge).  The first tile's epilogue runs under the second
    // tile's MFMAs.
    float4_t wA[KC4], wB[KC4], x0, x1, y0, y1;
    float4_t bA, bB;
    wload(wA, bA, 0);
    wload(wB, bB, 16);
    int ct = 0;
    for (; ct + 16 < C_; ct += 32) {
        mm(wA, bA, x0, x1);
        mm(wB, bB, y0, y1);
        epi(x0, x1, ct);
        wload(wA, bA, ct + 32);
        wload(wB, bB, ct + 48);
        epi(y0, y1, ct + 16);
    }
    if (ct < C_) { mm(wA, bA, x0, x1); epi(x0, x1, ct); }
}

// run-time width -> the instantiation (K_ is a multiple of 16, at most 16 * KA)
template <bool LAST, int KA>
__device__ __forceinline__ void sa_layer_k(float *act, int sa, const float *W, const float *bias, int K_, int C_, int lane, int gsz)
{
    const int k4 = K_ >> 4;
    if (k4 == 1) sa_layer<LAST, 1>(act, sa, W, bias, C_, lane, gsz);
    else if (k4 == 2) sa_layer<LAST, 2>(act, sa, W, bias, C_, lane, gsz);
    else if constexpr (KA > 2) {
        if (k4 == 3) sa_layer<LAST, 3>(act, sa, W, bias, C_, lane, gsz);
        else if (k4 == 4) sa_layer<LAST, 4>(act, sa, W, bias, C_, lane, gsz);
        else if (k4 == 5) sa_layer<LAST, 5>(act, sa, W, bias, C_, lane, gsz);
        else if (k4 == 6) sa_layer<LAST, 6>(act, sa, W, bias, C_, lane, gsz);
        else if (k4 == 7) sa_layer<LAST, 7>(act, sa, W, bias, C_, lane, gsz);
        else sa_layer<LAST, 8>(act, sa, W, bias, C_, lane, gsz);
    }
}

// The gather of a pass is software-pipelined against the MFMAs of the pass before it: the member indices of pass p + 2 and the
// feature / xyz vectors of pass p + 1 are in flight (registers) while pass p multiplies.  VW = vector width of a feature row
// load (4, 2 or 1 floats: the widest that divides C); a lane owns the same (row, channel) positions of the 32 x C tile in every
// pass, so their decomposition is done once.  (Round 2 / early round 3: scalar loads behind a per-element index load, nothing
// else to run on the SIMD while they were in flight - the gather, not the MFMAs, was 80 % of the kernel.)
// Two shapes are instantiated: PF = 2 prefetch vectors per lane and layer inputs up to 32 wide (SA1: 4 waves per SIMD hide what
// is left of the latencies of its short passes) and PF = 16 / up to 128 wide (SA2: the weights take most of a CU's LDS, one
// workgroup per CU, a wave has its SIMD's register file to itself).

template <int VW> struct SaVec;
template <> struct SaVec<4> { typedef float4_t T; };
template <> struct SaVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct SaVec<1> { typedef float T; };

template <int VW, int SA_PF, int KA>
__global__ __launch_bounds__(64 * SA_WAVES, (KA <= 2 ? 2 : 1)) void k_sa_mlp(SaMlpArgs a)
{
    typedef typename SaVec<VW>::T vec_t;
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
    kmax = kmax > C3 / 4 ? kmax : C3 / 4;     // (the last layer parks 4 rows of C3 + 16 partial maxima in each half tile)
    const int sa = kmax + 4;                  // activation row stride (16-byte aligned rows, 4 banks apart)
    float *act = Bs + C1 + C2 + C3 + (size_t)w * 32 * sa;
    // staging: read W^T (k-major, columns fastest) in storage order - coalesced - and scatter into the [column][k] rows; the grid
    // is one resident set of workgroups, so this happens once per CU slot, not once per 2048th of the work
    for (int i = tid; i < C1 * Cin; i += blockDim.x) { const int k = i / C1, c = i - k * C1; W1[c * (Cin + 4) + k] = k < a.cin_src ? a.w1t[i] : 0.f; }
    for (int i = tid; i < C2 * C1; i += blockDim.x) { const int k = i / C2, c = i - k * C2; W2[c * (C1 + 4) + k] = a.w2t[i]; }
    for (int i = tid; i < C3 * C2; i += blockDim.x) { const int k = i / C3, c = i - k * C3; W3[c * (C2 + 4) + k] = a.w3t[i]; }
    for (int i = tid; i < C1; i += blockDim.x) Bs[i] = a.b1[i];
    for (int i = tid; i < C2; i += blockDim.x) Bs[C1 + i] = a.b2[i];
    for (int i = tid; i < C3; i += blockDim.x) Bs[C1 + C2 + i] = a.b3[i];
    __syncthreads();
    const int gsz = a.K;                                   // 16 or 32 members per group
    const int gpp = 32 / gsz;                              // groups per 32-row pass
    const long long groups = (long long)a.B * a.S;
    const long long passes = (groups + gpp - 1) / gpp;
    const long long stride = (long long)gridDim.x * nw;
    const int C = a.C, CC = a.C + 3;
    const int nv = C / VW, nvec = 32 * nv;                 // vectors per row / per tile
    // this lane's positions in the tile: vector j covers row pr[j], channels pc[j] .. pc[j] + VW - 1 (-1: none)
    int pr[SA_PF], pc[SA_PF];
#pragma unroll
    for (int j = 0; j < SA_PF; j++) {
        const int e = j * 64 + lane;
        pr[j] = e < nvec ? e / nv : -1;
        pc[j] = e < nvec ? (e - (e / nv) * nv) * VW : 0;
    }
    // xyz: element e < 96 = (row e / 3, axis e % 3); lanes 0..63 take e = lane, lanes 0..31 also e = 64 + lane
    const int xr0 = lane / 3, xa0 = lane - 3 * xr0;
    const int xr1 = (64 + lane) / 3, xa1 = (64 + lane) - 3 * xr1;
    const bool x1 = lane < 32;

    // pipeline registers
    vec_t pf[SA_PF];
    float px0 = 0.f, px1 = 0.f, pn0 = 0.f, pn1 = 0.f;
    // lanes 0..31: source row of tile row `lane` = rbase + ridx.  The loaded member index is NOT touched here (ridx is consumed a
    // whole pass later, in issue()): combining it with the batch offset at this point put an s_waitcnt vmcnt(0) - behind the
    // vector loads just issued - in front of every MFMA phase.
    auto load_rows = [&](long long ps, int &rbase, int &ridx) {
        rbase = -1; ridx = 0;
        if (lane < 32 && ps < passes) {
            const long long g = ps * gpp + lane / gsz;
            if (g < groups) { rbase = (int)(g / a.S) * a.N; ridx = (int)a.gidx[ps * 32 + lane]; }
        }
    };
    auto issue = [&](int rbase, int ridx, long long ps) {  // this pass's row offsets (lanes 0..31)
        const int ro = rbase < 0 ? -1 : rbase + ridx;
#pragma unroll
        for (int j = 0; j < SA_PF; j++) {
            if (j * 64 < nvec) {
                const int r = __shfl(ro, pr[j] < 0 ? 0 : pr[j]);
                vec_t v;
                if constexpr (VW == 1) v = 0.f; else v = (vec_t)(0.f);
                if (pr[j] >= 0 && r >= 0) v = *reinterpret_cast<const vec_t *>(a.feats + (long long)r * C + pc[j]);
                pf[j] = v;
            }
        }
        const int r0 = __shfl(ro, xr0), r1 = __shfl(ro, x1 ? xr1 : 0);
        const long long g0 = ps * gpp;
        px0 = 0.f; pn0 = 0.f; px1 = 0.f; pn1 = 0.f;
        if (r0 >= 0) { px0 = a.xyz[(long long)r0 * 3 + xa0]; pn0 = a.new_xyz[(g0 + xr0 / gsz) * 3 + xa0]; }
        if (x1 && r1 >= 0) { px1 = a.xyz[(long long)r1 * 3 + xa1]; pn1 = a.new_xyz[(g0 + xr1 / gsz) * 3 + xa1]; }
    };

    // the maxima of a pass leave for HBM at the top of the NEXT pass (from registers), after that pass's tile went to the LDS:
    // every wait for prefetched vectors then only has loads and stores in front of it that had a whole MFMA phase to finish
    float oreg[4] = {0.f, 0.f, 0.f, 0.f};
    long long og0 = -1;
    auto flush = [&]() {
        if (og0 < 0) return;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q < gpp && og0 + q < groups) {
                float *o = a.out + (og0 + q) * a.out_stride + a.out_off;
                if (lane < C3) o[lane] = oreg[2 * q];
                if (lane + 64 < C3) o[lane + 64] = oreg[2 * q + 1];
            }
        }
    };
    long long ps = (long long)blockIdx.x * nw + w;
    int rb_next, ri_next;
    load_rows(ps, rb_next, ri_next);
    issue(rb_next, ri_next, ps);
    load_rows(ps + stride, rb_next, ri_next);
    for (; ps < passes; ps += stride) {
        // the tile of this pass: registers -> LDS (channels [0, C) features, [C, C + 3) relative xyz, zero up to Cin)
#pragma unroll
        for (int j = 0; j < SA_PF; j++)
            if (j * 64 < nvec && pr[j] >= 0) *reinterpret_cast<vec_t *>(act + pr[j] * sa + pc[j]) = pf[j];
        act[xr0 * sa + C + xa0] = px0 - pn0;
        if (x1) act[xr1 * sa + C + xa1] = px1 - pn1;
        for (int c = CC + (lane >> 5); c < Cin; c += 2) act[(lane & 31) * sa + c] = 0.f;
        flush();
        // next pass's vectors and the indices of the one after it go out before the MFMAs of this one
        issue(rb_next, ri_next, ps + stride);
        load_rows(ps + 2 * stride, rb_next, ri_next);
        sa_layer_k<false, KA>(act, sa, W1, Bs, Cin, C1, lane, gsz);
        sa_layer_k<false, KA>(act, sa, W2, Bs + C1, C1, C2, lane, gsz);
        sa_layer_k<true, KA>(act, sa, W3, Bs + C1 + C2, C2, C3, lane, gsz);
        og0 = ps * gpp;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = lane + 64 * h;
            if (c < C3) {
                const int ms = C3 + 16;
                float m0 = act[c], m1 = act[16 * sa + c];
#pragma unroll
                for (int r = 1; r < 4; r++) { m0 = fmaxf(m0, act[r * ms + c]); m1 = fmaxf(m1, act[16 * sa + r * ms + c]); }
                if (gpp == 1) oreg[h] = fmaxf(m0, m1);
                else { oreg[h] = m0; oreg[2 + h] = m1; }
            }
        }
    }
    flush();
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
    const int kin = kmax;                     // widest layer INPUT: picks the instantiation
    kmax = kmax > C3 / 4 ? kmax : C3 / 4;
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
    const uintptr_t fa = (uintptr_t)feats;      // widest vector load the rows' width and the base address allow
    const int vw = (C % 4 == 0 && fa % 16 == 0) ? 4 : (C % 2 == 0 && fa % 8 == 0) ? 2 : 1;
    const int npf = (32 * (C / vw) + 63) / 64;
    if (npf > 16 || (long long)B * N >= (1ll << 31)) return -3;
    const bool small = npf <= 2 && kin <= 32;
    void (*kern)(SaMlpArgs) = small ? (vw == 4 ? k_sa_mlp<4, 2, 2> : vw == 2 ? k_sa_mlp<2, 2, 2> : k_sa_mlp<1, 2, 2>)
                                    : (vw == 4 ? k_sa_mlp<4, 16, 8> : vw == 2 ? k_sa_mlp<2, 16, 8> : k_sa_mlp<1, 16, 8>);
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    SaMlpArgs a;
    a.feats = feats; a.xyz = xyz; a.new_xyz = new_xyz; a.gidx = (const long long *)gidx; a.out = out;
    a.w1t = w1t; a.b1 = b1; a.w2t = w2t; a.b2 = b2; a.w3t = w3t; a.b3 = b3;
    a.B = B; a.N = N; a.S = S; a.K = K; a.C = C; a.cin_src = cin_pad; a.Cin = Cin; a.C1 = C1; a.C2 = C2; a.C3 = C3;
    a.out_stride = out_stride; a.out_off = out_off;
    const long long groups = (long long)B * S;
    const long long passes = (groups + (32 / K) - 1) / (32 / K);
    // grid-stride over the passes with exactly one resident set of workgroups: the weights are staged once per workgroup
    int dev = 0, ncu = 256, per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 64 * nw, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    long long grid = (passes + nw - 1) / nw;
    if (grid > (long long)ncu * per_cu) grid = (long long)ncu * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
