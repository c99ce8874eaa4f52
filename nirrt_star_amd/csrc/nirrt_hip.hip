// nirrt_hip.hip — kernels + C ABI (include/nirrt_hip.h) of libnirrt_hip.so, gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
#include "nirrt_device.hpp"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// kernels: one workgroup (NT threads) per tree
// ------------------------------------------------------------------------------------------------
#define NT 256   // 4 wave64 = one wave per SIMD; several trees share a CU

// The workgroup's LDS working set lives at file scope so that the (non-inlined) loop-body function
// addresses it as LDS (ds_* instructions) instead of through a generic pointer.
__shared__ Lds<NT> g_lds;

template <int D>
__global__ __launch_bounds__(NT) void k_init(TreeDev *tp)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    stage_obstacles<NT>(s, t);
    if (threadIdx.x == 0) {
        t.n_gc = 0; t.n_sol = 0; t.status = 0;
        t.sol_dirty = 1; t.gc_dirty = 1; t.sol_best = -1; t.gc_best = -1;
        t.sol_best_cost = __builtin_inf(); t.gc_best_cost = __builtin_inf();
    }
    __syncthreads();
    // child lists + exact cost cache of the current tree (n == 1 after create/reset; n > 1 after upload)
    int n = t.n;
    for (int i = threadIdx.x; i < n; i += NT) t.first_child[i] = -1;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < n; i++) link_child(t, i, t.aux[i].parent);
    __syncthreads();
    if (threadIdx.x == 0) {   // uploaded vertices may lie outside the range box
        double cm = t.cmax;
        for (int i = 0; i < n; i++)
            for (int k = 0; k < D; k++) cm = fmax(cm, fabs(t.c[k][i]));
        t.cmax = cm;
    }
    for (int i = threadIdx.x; i < n; i += NT) {
#pragma unroll
        for (int k = 0; k < D; k++) t.cf[k][i] = (float)t.c[k][i];
        VRec vr;
        vr.x = t.c[0][i]; vr.y = t.c[1][i]; vr.z = D == 3 ? t.c[D - 1][i] : 0.;
        vr.cost = walk_cost<D>(t, i);
        t.vrec[i] = vr;
    }
    __syncthreads();
    // goal-candidate list over the current vertices, ascending
    for (int i = 0; i < n; i++) {  // uniform loop; cheap for n == 1, acceptable for test uploads
        double v[D];
        load_vertex<D>(t, i, v);
        wg_goal_candidate<D, NT>(s, t, i, v);
    }
}

template <int D>
__global__ __launch_bounds__(NT) void k_nearest(TreeDev *tp, double q0, double q1, double q2, int *out_idx)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    double q[3] = {q0, q1, q2};
    int bi = wg_nearest<D, NT>(s, t, t.n, q);
    if (threadIdx.x == 0) *out_idx = bi;
}

template <int D>
__global__ __launch_bounds__(NT) void k_collision_batch(TreeDev *tp, long long n_seg, const double *seg, unsigned char *out)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    stage_obstacles<NT>(s, t);
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n_seg; i += (long long)gridDim.x * NT) {
        double a[D], b[D];
#pragma unroll
        for (int k = 0; k < D; k++) { a[k] = seg[i * 2 * D + k]; b[k] = seg[i * 2 * D + D + k]; }
        out[i] = seg_all<D, NT>(s, a, b, t.clearance) ? 1 : 0;
    }
}

template <int D>
__global__ __launch_bounds__(NT) void k_points(TreeDev *tp, long long n, const double *pts, unsigned char *inside,
                                               unsigned char *valid)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    stage_obstacles<NT>(s, t);
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        double p[D];
#pragma unroll
        for (int k = 0; k < D; k++) p[k] = pts[i * D + k];
        bool in = point_in_obs<D, NT>(s, p, t.clearance);
        if (inside) inside[i] = in ? 1 : 0;
        if (valid) valid[i] = (point_in_range<D>(t, p) && !in) ? 1 : 0;
    }
}

template <int D>
__global__ __launch_bounds__(NT) void k_near(TreeDev *tp, double q0, double q1, double q2, int new_idx, int *out_k)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    stage_obstacles<NT>(s, t);
    double q[3] = {q0, q1, q2};
    int k = wg_near<D, NT>(s, t, t.n, q, new_idx);   // result left in t.nr_idx[0..k)
    if (threadIdx.x == 0) *out_k = k;
}

template <int D>
__global__ __launch_bounds__(NT) void k_cost(TreeDev *tp, long long n_idx, const long long *idx, double *out)
{
    TreeDev &t = *tp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n_idx; i += (long long)gridDim.x * NT)
        out[i] = walk_cost<D>(t, (int)idx[i]);
}

template <int D>
__global__ __launch_bounds__(NT) void k_goal_parent(TreeDev *tp, int *out_idx, double *out_len)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    int gp;
    double len;
    wg_goal_parent<D, NT>(s, t, gp, len);
    if (threadIdx.x == 0) { *out_idx = gp; *out_len = len; }
}

template <int D>
__global__ __launch_bounds__(NT) void k_best_solution(TreeDev *tp, int *out_idx, double *out_c)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    double cb;
    int xb;
    wg_best_solution<D, NT>(s, t, cb, xb);
    if (threadIdx.x == 0) { *out_idx = xb; *out_c = cb; }
}

template <int D>
__global__ __launch_bounds__(NT, 4) void k_step(TreeDev *tp, double q0, double q1, double q2, int host_steer, int nearest_in,
                                             unsigned flags, nirrt_step_result *res)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *tp;
    stage_obstacles<NT>(s, t);
    double q[3] = {q0, q1, q2};
    wg_iteration<D, NT>(s, t, q, host_steer != 0, nearest_in, flags, res);
    double cb;
    int xb;
    wg_report<D, NT>(s, t, flags, cb, xb);
    if (threadIdx.x == 0) {
        res->c_best = cb; res->x_best = xb; res->n_solutions = t.n_sol; res->n = t.n; res->status = t.status;
    }
}

// The persistent loops call the loop body through a real function call: inlined into the loop the
// compiler hoists the tree descriptor into registers across iterations and spills.
template <int D>
__device__ __noinline__ int iteration_call(TreeDev *tp, double q0, double q1, double q2, unsigned flags, int pref_ni,
                                           int has_next, double n0, double n1, double n2)
{
    double q[3] = {q0, q1, q2};
    double qn[3] = {n0, n1, n2};
    wg_iteration<D, NT>(g_lds, *tp, q, false, 0, flags, nullptr, pref_ni, has_next ? qn : nullptr);
    return g_lds.bc_i[6];   // nearest index of the next sample if this iteration's Near scan ran, else -1
}

template <int D>
__device__ __noinline__ double report_call(TreeDev *tp, unsigned flags)
{
    double cb;
    int xb;
#ifdef NIRRT_PROFILE
    long long t0_ = wall_clock64();
#endif
    wg_report<D, NT>(g_lds, *tp, flags, cb, xb);
#ifdef NIRRT_PROFILE
    if (threadIdx.x == 0) tp->prof[7] += wall_clock64() - t0_;
#endif
    return cb;
}

// persistent loop, replayed samples: block b owns trees[b]
struct RunDev {
    unsigned flags;
    int pad;
    long long iters;
    const double *samples;  // (n_trees, iters, D) or nullptr
    double *cost_trace;     // (n_trees, iters) or nullptr
    long long *iters_done;  // (n_trees,)
};

template <int D>
__global__ __launch_bounds__(NT, 4) void k_run_replay(TreeDev *const *trees, RunDev a)
{
    Lds<NT> &s = g_lds;
    TreeDev &t = *trees[blockIdx.x];
    stage_obstacles<NT>(s, t);
    const double *smp = a.samples + (long long)blockIdx.x * a.iters * D;
    double *trace = a.cost_trace ? a.cost_trace + (long long)blockIdx.x * a.iters : nullptr;
    long long k = 0;
    int pref = -1;
    for (; k < a.iters; k++) {
        double q[3] = {0., 0., 0.}, qn[3] = {0., 0., 0.};
        const int has_next = k + 1 < a.iters;
#pragma unroll
        for (int c = 0; c < D; c++) {
            q[c] = smp[k * D + c];
            if (has_next) qn[c] = smp[(k + 1) * D + c];
        }
        // the Near scan of this iteration also answers the next iteration's nearest query (one pass per iteration)
        pref = iteration_call<D>(&t, q[0], q[1], q[2], a.flags, pref, has_next, qn[0], qn[1], qn[2]);
        if (trace || (a.flags & NIRRT_F_STOP_FIRST)) {
            double cb = report_call<D>(&t, a.flags);
            if (trace && threadIdx.x == 0) trace[k] = cb;
            if ((a.flags & NIRRT_F_STOP_FIRST) && cb < __builtin_inf()) { k++; break; }
        }
        if (t.status != 0) { k++; break; }
    }
    if (threadIdx.x == 0) a.iters_done[blockIdx.x] = k;
}

// ------------------------------------------------------------------------------------------------
// in-kernel sampling: consumes raw MT19937 32-bit outputs exactly like numpy's legacy RandomState
// (random_sample: a = w>>5, b = w>>6, (a*2^26+b)/2^53; uniform(lo,hi) = lo + (hi-lo)*u) and CPython's
// random.random() (same 53-bit construction), so the host generators can be advanced by the
// reported word counts afterwards.
// ------------------------------------------------------------------------------------------------
struct WordStream {
    const unsigned *w;
    long long n, pos;
    __device__ __forceinline__ bool has(long long k) const { return pos + k <= n; }
    __device__ __forceinline__ double next_double()
    {
        unsigned a = w[pos] >> 5, b = w[pos + 1] >> 6;
        pos += 2;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// SampleFree (rrt_base_2d.py:46-52 / rrt_base_3d.py:49-58); false = stream ran dry
template <int D>
__device__ __forceinline__ bool sample_free(const Lds<NT> &s, const TreeDev &t, WordStream &np, double *out)
{
    for (;;) {
        if (!np.has(2 * D)) return false;
#pragma unroll
        for (int k = 0; k < D; k++) {
            double lo = t.lo[k] + t.clearance, hi = t.hi[k] - t.clearance;
            out[k] = lo + (hi - lo) * np.next_double();
        }
        if (!point_in_obs<D, NT>(s, out, t.clearance)) return true;
    }
}

// SampleInformedSubset (irrt_star_2d.py:121-151 / irrt_star_3d.py:117-158).  The two matrix
// products go through BLAS in the reference; the forms below are what OpenBLAS 0.3.29 (numpy 2.2.6
// wheel, Haswell/Zen kernels) evaluates for these shapes (tests/test_host_sampling.py pins them on
// the host): C.L -> RN(C[i][j]*r[j]);  2D (3,3)x(3,1) with x2 = 0 -> fma(a0,x0,a1*x1);
// 3D (3,3)x(3,) -> fma(a2,x2,fma(a0,x0,a1*x1)).
template <int D>
__device__ __forceinline__ bool sample_informed(const Lds<NT> &s, const TreeDev &t, WordStream &np, WordStream &py,
                                                double c_max, double *out)
{
    const double c_min = t.c_min;
    double rad = c_max * c_max - c_min * c_min;
    double eps = rad < 0 ? 1e-6 : 0.;
    double r0 = c_max / 2.0;
    double r1 = __builtin_sqrt(rad + eps) / 2.0;
    double CL[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        CL[i][0] = t.CL_C[3 * i + 0] * r0;
        CL[i][1] = t.CL_C[3 * i + 1] * r1;
        CL[i][2] = t.CL_C[3 * i + 2] * r1;
    }
    for (;;) {
        double xb[3];
        if (D == 2) {
            // SampleUnitBall: python random.uniform(-1, 1) twice until inside the open unit disk
            for (;;) {
                if (!py.has(4)) return false;
                xb[0] = -1.0 + 2.0 * py.next_double();
                xb[1] = -1.0 + 2.0 * py.next_double();
                if (xb[0] * xb[0] + xb[1] * xb[1] < 1) break;
            }
            xb[2] = 0.;
#pragma unroll
            for (int i = 0; i < 2; i++) out[i] = __builtin_fma(CL[i][0], xb[0], CL[i][1] * xb[1]) + t.x_center[i];
        } else {
            // spherical coordinates with three numpy uniforms (not volume-uniform; reproduced as is)
            if (!np.has(6)) return false;
            const double PI = 3.141592653589793;
            double rr = 0.0 + (1.0 - 0.0) * np.next_double();
            double theta = 0.0 + (PI - 0.0) * np.next_double();
            double phi = 0.0 + (2 * PI - 0.0) * np.next_double();
            xb[0] = rr * sin(theta) * cos(phi);
            xb[1] = rr * sin(theta) * sin(phi);
            xb[2] = rr * cos(theta);
#pragma unroll
            for (int i = 0; i < 3; i++)
                out[i] = __builtin_fma(CL[i][2], xb[2], __builtin_fma(CL[i][0], xb[0], CL[i][1] * xb[1])) + t.x_center[i];
        }
        // Utils.is_valid: inside the clearance-shrunk range and outside every inflated obstacle
        if (point_in_range<D>(t, out) && !point_in_obs<D, NT>(s, out, t.clearance)) return true;
    }
}

// SamplePointCloud (nirrt_star_png_2d.py:129-130): np.random.randint(0, m) of the legacy generator = masked
// rejection on single 32-bit outputs (rng = m-1; no draw when rng == 0)
template <int D>
__device__ __forceinline__ int sample_cloud(const TreeDev &t, WordStream &np, double *out)
{
    const unsigned rng = (unsigned)(t.pc_n - 1);
    unsigned idx = 0;
    if (rng != 0) {
        unsigned mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        do {
            if (!np.has(1)) return 0;
            idx = np.w[np.pos++] & mask;
        } while (idx > rng);
    }
#pragma unroll
    for (int k = 0; k < D; k++) out[k] = t.pc[(size_t)idx * D + k];
    return 1;
}

struct RunSampleDev {
    unsigned flags;
    int pad;
    long long iters;
    const unsigned *const *np_words;
    const long long *n_np;
    const unsigned *const *py_words;
    const long long *n_py;
    long long *np_used;
    long long *py_used;
    double *cost_trace;
    long long *iters_done;
    int *stop_code;   // per tree: 0 done, NIRRT_E_STREAM, NIRRT_E_CAPACITY
};

// persistent loop with in-kernel sampling (RRT*: SampleFree; IRRT*: informed once a solution exists)
template <int D>
__global__ __launch_bounds__(NT, 4) void k_run_sample(TreeDev *const *trees, RunSampleDev a)
{
    Lds<NT> &s = g_lds;
    const int b = blockIdx.x;
    TreeDev &t = *trees[b];
    stage_obstacles<NT>(s, t);
    WordStream np = {a.np_words[b], a.n_np[b], 0};
    WordStream py = {a.py_words ? a.py_words[b] : nullptr, a.py_words ? a.n_py[b] : 0, 0};
    double *trace = a.cost_trace ? a.cost_trace + (long long)b * a.iters : nullptr;
    const bool irrt = (a.flags & NIRRT_F_IRRT) != 0;
    const bool png = (a.flags & NIRRT_F_PNG) != 0;
    const bool reports = irrt || (a.flags & NIRRT_F_GOAL_SCAN);
    long long k = 0;
    int stop = 0;
    // cb = best cost on the current tree: what IRRT* samples with at the top of the next iteration
    // (irrt_star_2d.py:51-53) and what planning_random records after each iteration (:223-229, :241)
    double cb = reports ? report_call<D>(&t, a.flags) : __builtin_inf();

    // draw node_rand for the coming iteration with the reference's policy (thread 0), broadcast through LDS.
    // returns 0 or a stop code; on failure the stream positions are left where they were.
    auto draw = [&](double cbest, double *q) -> int {
        if (threadIdx.x == 0) {
            double v[3] = {0., 0., 0.};
            long long np0 = np.pos, py0 = py.pos;
            int ok = 1, code = NIRRT_E_STREAM;
            bool from_cloud = false;
            if (png) {
                if (!np.has(2)) ok = 0;
                else from_cloud = np.next_double() < t.pc_rate;     // np.random.random() < pc_sample_rate
            }
            if (ok) {
                if (from_cloud) {
                    if (t.pc_n <= 0) { ok = 0; code = NIRRT_E_ARG; }   // empty prediction: the reference raises in randint(0, 0)
                    else ok = sample_cloud<D>(t, np, v);
                } else {
                    ok = (irrt && cbest < __builtin_inf()) ? sample_informed<D>(s, t, np, py, cbest, v) : sample_free<D>(s, t, np, v);
                }
            }
            if (!ok) { np.pos = np0; py.pos = py0; }
            s.bc_d[0] = v[0]; s.bc_d[1] = v[1]; s.bc_d[2] = v[2];
            s.bc_i[0] = ok ? 0 : code;
        }
        __syncthreads();
        int rc = s.bc_i[0];
        q[0] = s.bc_d[0]; q[1] = s.bc_d[1]; q[2] = s.bc_d[2];
        __syncthreads();
        return rc;
    };

    // Software pipeline: the sample of iteration k+1 is drawn BEFORE iteration k runs, with the best cost known at
    // that point, so that iteration k's Near scan can also answer iteration k+1's nearest query.  Tree operations
    // never touch the generators, so the early draw consumes exactly the words the reference's draw would - unless
    // iteration k changes the best cost (or ends the run): then the draw is undone (stream positions restored) and
    // repeated with the new value, and the prefetched nearest index is dropped.
    double q[3], qn[3] = {0., 0., 0.};
    bool have_q = false, spec = false;
    int pref = -1;
    long long sp_np = 0, sp_py = 0;
    for (; k < a.iters; k++) {
        if (!have_q) {
            // NIRRT*: the guidance cloud is refreshed by the host (PointNet++) once the best cost has dropped below
            // pc_update_cost_ratio * c_update (nirrt_star_png_2d.py:114-116) -> hand control back before sampling
            if (png && cb < t.pc_ratio * t.c_update) { stop = NIRRT_E_CLOUD; break; }
            stop = draw(cb, q);
            if (stop) break;
            pref = -1;
        }
        have_q = false;
        // speculative draw for iteration k+1 (thread 0 owns the stream state)
        spec = false;
        if (k + 1 < a.iters) {
            sp_np = np.pos; sp_py = py.pos;
            spec = draw(cb, qn) == 0;
        }
        pref = iteration_call<D>(&t, q[0], q[1], q[2], a.flags, pref, spec ? 1 : 0, qn[0], qn[1], qn[2]);
        const double cb_new = reports ? report_call<D>(&t, a.flags) : cb;
        if (trace && threadIdx.x == 0) trace[k] = cb_new;
        bool leave = false;
        if (t.status != 0) { stop = t.status; leave = true; }
        else if ((a.flags & NIRRT_F_STOP_FIRST) && cb_new < __builtin_inf()) leave = true;
        const bool keep = spec && !leave && cb_new == cb && !(png && cb_new < t.pc_ratio * t.c_update);
        if (spec && !keep) { np.pos = sp_np; py.pos = sp_py; pref = -1; }   // undo the early draw (only thread 0's copy matters)
        cb = cb_new;
        if (leave) { k++; break; }
        if (keep) { q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; have_q = true; }
    }
    if (threadIdx.x == 0) {
        a.iters_done[b] = k;
        a.np_used[b] = np.pos;
        a.py_used[b] = py.pos;
        a.stop_code[b] = stop;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return NIRRT_E_HIP;                                                               \
        }                                                                                     \
    } while (0)

// host twin of hypot_py (CPython vector_norm) for nirrt_upload's edge-length column
static double host_hypot_py(int n, const double *d)
{
    const double T27 = 134217729.0;
    double vec[3], mx = 0.0;
    for (int i = 0; i < n; i++) { vec[i] = std::fabs(d[i]); if (vec[i] > mx) mx = vec[i]; }
    if (mx == 0.0) return mx;
    int max_e;
    (void)std::frexp(mx, &max_e);
    double scale = std::ldexp(1.0, -max_e);
    volatile double x, oldcsum, csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0, t, hi, lo, h;
    for (int i = 0; i < n; i++) {
        x = vec[i] * scale;
        t = x * T27; hi = t - (t - x); lo = x - hi;
        x = hi * hi; oldcsum = csum; csum = csum + x; frac1 = frac1 + ((oldcsum - csum) + x);
        x = 2.0 * hi * lo; oldcsum = csum; csum = csum + x; frac2 = frac2 + ((oldcsum - csum) + x);
        frac3 = frac3 + lo * lo;
    }
    h = std::sqrt(csum - 1.0 + (frac1 + frac2 + frac3));
    x = h; t = x * T27; hi = t - (t - x); lo = x - hi;
    x = -hi * hi; oldcsum = csum; csum = csum + x; frac1 = frac1 + ((oldcsum - csum) + x);
    x = -2.0 * hi * lo; oldcsum = csum; csum = csum + x; frac2 = frac2 + ((oldcsum - csum) + x);
    x = -lo * lo; oldcsum = csum; csum = csum + x; frac3 = frac3 + ((oldcsum - csum) + x);
    x = csum - 1.0 + (frac1 + frac2 + frac3);
    return (h + x / (2.0 * h)) / scale;
}

struct Scratch {  // pinned, device-visible result slots
    nirrt_step_result step;
    int i[4];
    double d[4];
};

struct nirrt_tree {
    nirrt_config cfg;
    int dim;
    int cap;
    int device;
    hipStream_t stream;
    TreeDev host;    // host mirror of the descriptor (pointers are device pointers)
    TreeDev *dev;    // descriptor in HBM
    double *near_r;  // device table
    Scratch *scratch;      // pinned host memory
    Scratch *scratch_dev;  // device alias of the same memory
    double *pc_dev;        // guidance cloud (nirrt_set_cloud)
};

extern "C" const char *nirrt_last_error(void) { return g_err.c_str(); }

extern "C" int nirrt_device_count(int *count)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
    if (count) *count = c;
    return NIRRT_OK;
}

#define DISPATCH_DIM(t, KERNEL, grid, ...)                                                     \
    do {                                                                                       \
        if ((t)->dim == 2) hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(NT), 0, (t)->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<3>, dim3(grid), dim3(NT), 0, (t)->stream, __VA_ARGS__); \
    } while (0)

static int sync_check(nirrt_tree *t)
{
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(t->stream));
    return NIRRT_OK;
}

static int push_desc(nirrt_tree *t)
{
    HIPCHK(hipMemcpyAsync(t->dev, &t->host, sizeof(TreeDev), hipMemcpyHostToDevice, t->stream));
    return NIRRT_OK;
}

extern "C" int nirrt_destroy(nirrt_tree *t)
{
    if (!t) return NIRRT_OK;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    TreeDev &h = t->host;
    void *bufs[] = {h.cf[0], h.cf[1], h.cf[2], h.st_c[0], h.st_c[1], h.st_c[2], h.c[0], h.c[1], h.c[2], h.aux, h.vrec, h.nr_cost, h.first_child, h.next_sib, h.prev_sib, h.bfs_q, h.st_idx,
                    h.nr_idx, h.nr_flag, h.nr_dist, h.nr_c0, h.nr_c1, h.sol, h.sol_line, h.gc_idx, h.gc_dist, h.gc_col,
                    t->near_r};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (t->pc_dev) (void)hipFree(t->pc_dev);
    if (t->dev) (void)hipFree(t->dev);
    if (t->scratch) (void)hipHostFree(t->scratch);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return NIRRT_OK;
}

extern "C" int nirrt_reset(nirrt_tree *t)
{
    if (!t) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    for (int k = 0; k < t->dim; k++)
        HIPCHK(hipMemcpyAsync(t->host.c[k], &t->cfg.x_start[k], sizeof(double), hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipMemsetAsync(t->host.aux, 0, sizeof(Aux) * (size_t)(t->cap + SCAN_PAD), t->stream));  // parent 0, elen 0, mark 0
    t->host.n = 1;
    t->host.n_sol = 0;
    t->host.n_gc = 0;
    t->host.status = 0;
    t->host.stamp = 0;
    t->host.scan_elems = 0;
    t->host.alg_elems = 0;
    int rc = push_desc(t);
    if (rc) return rc;
    DISPATCH_DIM(t, k_init, 1, t->dev);
    return sync_check(t);
}

extern "C" int nirrt_create(const nirrt_config *cfg, nirrt_tree **out)
{
    if (!cfg || !out) { g_err = "null argument"; return NIRRT_E_ARG; }
    *out = nullptr;
    if (cfg->dim != 2 && cfg->dim != 3) { g_err = "dim must be 2 or 3"; return NIRRT_E_ARG; }
    if (cfg->iter_max < 0 || cfg->iter_max + 1 > (1ll << 30)) { g_err = "iter_max out of range"; return NIRRT_E_ARG; }
    if (cfg->n_round < 0 || cfg->n_round > MAX_OBS || cfg->n_box < 0 || cfg->n_box > MAX_OBS) {
        g_err = "too many obstacles (NIRRT_MAX_OBSTACLES per kind)";
        return NIRRT_E_CAPACITY;
    }
    if ((cfg->n_round > 0 && !cfg->round_obs) || (cfg->n_box > 0 && !cfg->box_obs)) { g_err = "null obstacle table"; return NIRRT_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        g_err = "no HIP device visible";
        return NIRRT_E_NODEVICE;
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_err = "device_id out of range"; return NIRRT_E_ARG; }

    nirrt_tree *t = new nirrt_tree();
    std::memset(&t->host, 0, sizeof(TreeDev));
    t->cfg = *cfg;
    t->cfg.round_obs = nullptr;
    t->cfg.box_obs = nullptr;
    t->dim = cfg->dim;
    t->cap = (int)(cfg->iter_max + 1);
    t->device = cfg->device_id;
    t->stream = nullptr;
    t->dev = nullptr; t->near_r = nullptr; t->scratch = nullptr; t->scratch_dev = nullptr; t->pc_dev = nullptr;
    const int D = t->dim;
    auto fail = [&](int rc) { nirrt_destroy(t); return rc; };
#define HIPCHK_T(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return fail(NIRRT_E_HIP);                                                         \
        }                                                                                     \
    } while (0)
    HIPCHK_T(hipSetDevice(t->device));
    HIPCHK_T(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    TreeDev &h = t->host;
    const size_t np = (size_t)t->cap + SCAN_PAD;   // padded element count of every per-vertex array
    for (int k = 0; k < D; k++) {
        HIPCHK_T(hipMalloc(&h.c[k], sizeof(double) * np));
        HIPCHK_T(hipMemset(h.c[k], 0, sizeof(double) * np));
    }
    for (int k = 0; k < D; k++) HIPCHK_T(hipMalloc(&h.st_c[k], sizeof(double) * np));
    for (int k = 0; k < D; k++) {
        HIPCHK_T(hipMalloc(&h.cf[k], sizeof(float) * np));
        HIPCHK_T(hipMemset(h.cf[k], 0, sizeof(float) * np));
    }
    {
        double cmax = 0.;
        for (int k = 0; k < D; k++) cmax = std::fmax(cmax, std::fmax(std::fabs(cfg->range_lo[k]), std::fabs(cfg->range_hi[k])));
        h.cmax = cmax;
    }
    HIPCHK_T(hipMalloc(&h.aux, sizeof(Aux) * np));
    HIPCHK_T(hipMalloc(&h.vrec, sizeof(VRec) * np));
    HIPCHK_T(hipMalloc(&h.nr_cost, sizeof(double) * np));
    HIPCHK_T(hipMalloc(&h.first_child, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.next_sib, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.prev_sib, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.bfs_q, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.st_idx, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.nr_idx, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.nr_flag, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.nr_dist, sizeof(double) * np));
    HIPCHK_T(hipMalloc(&h.nr_c0, sizeof(double) * np));
    HIPCHK_T(hipMalloc(&h.nr_c1, sizeof(double) * np));
    h.cap = t->cap;
    h.dim = D;
    h.cap_sol = t->cap;
    HIPCHK_T(hipMalloc(&h.sol, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.sol_line, sizeof(double) * np));
    HIPCHK_T(hipMalloc(&h.gc_idx, sizeof(int) * np));
    HIPCHK_T(hipMalloc(&h.gc_dist, sizeof(double) * np));
    HIPCHK_T(hipMalloc(&h.gc_col, np));
    HIPCHK_T(hipMalloc(&t->near_r, sizeof(double) * (size_t)(t->cap + 1)));
    HIPCHK_T(hipMalloc(&t->dev, sizeof(TreeDev)));
    HIPCHK_T(hipHostMalloc((void **)&t->scratch, sizeof(Scratch), hipHostMallocMapped));
    HIPCHK_T(hipHostGetDevicePointer((void **)&t->scratch_dev, t->scratch, 0));
    // Near radius table with the host libm (the reference's math.sqrt/math.log/float pow):
    // rrt_star_2d.py:133  r = min(gamma*sqrt(log(n)/n), step_len);  rrt_star_3d.py:134 cube root
    {
        std::vector<double> r((size_t)t->cap + 1, 0.0);
        for (int n = 1; n <= t->cap; n++) {
            double x = std::log((double)n) / (double)n;
            double v = D == 2 ? cfg->search_radius * std::sqrt(x) : cfg->search_radius * std::pow(x, 1 / 3.);
            r[(size_t)n] = v < cfg->step_len ? v : cfg->step_len;
        }
        HIPCHK_T(hipMemcpy(t->near_r, r.data(), sizeof(double) * r.size(), hipMemcpyHostToDevice));
    }
    h.near_r = t->near_r;
    for (int k = 0; k < 3; k++) {
        h.start[k] = k < D ? cfg->x_start[k] : 0.;
        h.goal[k] = k < D ? cfg->x_goal[k] : 0.;
        h.lo[k] = k < D ? cfg->range_lo[k] : 0.;
        h.hi[k] = k < D ? cfg->range_hi[k] : 0.;
    }
    h.step_len = cfg->step_len;
    h.clearance = cfg->clearance;
    h.n_round = cfg->n_round;
    h.n_box = cfg->n_box;
    for (int i = 0; i < cfg->n_round; i++) {
        const double *c = cfg->round_obs + (size_t)i * (D + 1);
        h.rnd[i][0] = c[0]; h.rnd[i][1] = c[1]; h.rnd[i][2] = D == 3 ? c[2] : 0.; h.rnd[i][3] = c[D];
    }
    for (int i = 0; i < cfg->n_box; i++) {
        const double *b = cfg->box_obs + (size_t)i * 2 * D;
        h.box[i][0] = b[0]; h.box[i][1] = b[1]; h.box[i][2] = D == 3 ? b[2] : 0.;
        h.box[i][3] = b[D]; h.box[i][4] = b[D + 1]; h.box[i][5] = D == 3 ? b[D + 2] : 0.;
    }
    h.c_min = 0.;
    for (int k = 0; k < 9; k++) h.CL_C[k] = (k % 4 == 0) ? 1. : 0.;
    h.pc = nullptr; h.pc_n = 0; h.pad2 = 0; h.pc_rate = 0.; h.pc_ratio = 0.; h.c_update = std::numeric_limits<double>::infinity();
    int rc = nirrt_reset(t);
    if (rc) return fail(rc);
    *out = t;
    return NIRRT_OK;
#undef HIPCHK_T
}

extern "C" int nirrt_num_vertices(nirrt_tree *t, int64_t *n)
{
    if (!t || !n) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    TreeDev tmp;
    HIPCHK(hipMemcpyAsync(&tmp, t->dev, sizeof(TreeDev), hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    *n = tmp.n;
    return NIRRT_OK;
}

extern "C" int nirrt_upload(nirrt_tree *t, int64_t n, const double *vertices, const int64_t *parents)
{
    if (!t || !vertices || !parents || n < 1 || n > t->cap) { g_err = "bad upload arguments"; return NIRRT_E_ARG; }
    HIPCHK(hipSetDevice(t->device));
    const int D = t->dim;
    std::vector<double> col((size_t)n);
    for (int k = 0; k < D; k++) {
        for (int64_t i = 0; i < n; i++) col[(size_t)i] = vertices[i * D + k];
        HIPCHK(hipMemcpy(t->host.c[k], col.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    }
    std::vector<Aux> ax((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (parents[i] < 0 || parents[i] >= n) { g_err = "parent index out of range"; return NIRRT_E_ARG; }
        double d[3] = {0., 0., 0.};
        for (int k = 0; k < D; k++) d[k] = vertices[i * D + k] - vertices[parents[i] * D + k];
        ax[(size_t)i].elen = i == 0 ? 0. : host_hypot_py(D, d);
        ax[(size_t)i].parent = (int)parents[i];
        ax[(size_t)i].pad = 0;
    }
    HIPCHK(hipMemcpy(t->host.aux, ax.data(), sizeof(Aux) * (size_t)n, hipMemcpyHostToDevice));
    t->host.n = (int)n;
    t->host.n_sol = 0;
    t->host.n_gc = 0;
    t->host.status = 0;
    int rc = push_desc(t);
    if (rc) return rc;
    DISPATCH_DIM(t, k_init, 1, t->dev);
    return sync_check(t);
}

extern "C" int nirrt_download(nirrt_tree *t, double *vertices, int64_t *parents, int64_t *n_out)
{
    if (!t) return NIRRT_E_ARG;
    int64_t n = 0;
    int rc = nirrt_num_vertices(t, &n);
    if (rc) return rc;
    const int D = t->dim;
    if (vertices) {
        std::vector<double> col((size_t)n);
        for (int k = 0; k < D; k++) {
            HIPCHK(hipMemcpy(col.data(), t->host.c[k], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; i++) vertices[i * D + k] = col[(size_t)i];
        }
    }
    if (parents) {
        std::vector<Aux> ax((size_t)n);
        HIPCHK(hipMemcpy(ax.data(), t->host.aux, sizeof(Aux) * (size_t)n, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) parents[i] = ax[(size_t)i].parent;
    }
    if (n_out) *n_out = n;
    return NIRRT_OK;
}

extern "C" int nirrt_nearest(nirrt_tree *t, const double *q, int64_t *idx)
{
    if (!t || !q || !idx) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    DISPATCH_DIM(t, k_nearest, 1, t->dev, q[0], q[1], t->dim == 3 ? q[2] : 0., &t->scratch_dev->i[0]);
    int rc = sync_check(t);
    if (rc) return rc;
    *idx = t->scratch->i[0];
    return NIRRT_OK;
}

static int grid_for(long long n)
{
    long long g = (n + NT - 1) / NT;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

extern "C" int nirrt_collision_batch(nirrt_tree *t, int64_t n_seg, const double *seg, uint8_t *out)
{
    if (!t || n_seg < 0 || (n_seg > 0 && (!seg || !out))) return NIRRT_E_ARG;
    if (n_seg == 0) return NIRRT_OK;
    HIPCHK(hipSetDevice(t->device));
    double *d_seg = nullptr;
    unsigned char *d_out = nullptr;
    size_t bytes = sizeof(double) * (size_t)n_seg * 2 * t->dim;
    HIPCHK(hipMalloc(&d_seg, bytes));
    HIPCHK(hipMalloc(&d_out, (size_t)n_seg));
    HIPCHK(hipMemcpyAsync(d_seg, seg, bytes, hipMemcpyHostToDevice, t->stream));
    DISPATCH_DIM(t, k_collision_batch, grid_for(n_seg), t->dev, (long long)n_seg, (const double *)d_seg, d_out);
    HIPCHK(hipMemcpyAsync(out, d_out, (size_t)n_seg, hipMemcpyDeviceToHost, t->stream));
    int rc = sync_check(t);
    (void)hipFree(d_seg);
    (void)hipFree(d_out);
    return rc;
}

extern "C" int nirrt_points_in_obs(nirrt_tree *t, int64_t n, const double *pts, uint8_t *inside, uint8_t *valid)
{
    if (!t || n < 0 || (n > 0 && !pts)) return NIRRT_E_ARG;
    if (n == 0) return NIRRT_OK;
    HIPCHK(hipSetDevice(t->device));
    double *d_pts = nullptr;
    unsigned char *d_in = nullptr, *d_va = nullptr;
    size_t bytes = sizeof(double) * (size_t)n * t->dim;
    HIPCHK(hipMalloc(&d_pts, bytes));
    HIPCHK(hipMalloc(&d_in, (size_t)n));
    HIPCHK(hipMalloc(&d_va, (size_t)n));
    HIPCHK(hipMemcpyAsync(d_pts, pts, bytes, hipMemcpyHostToDevice, t->stream));
    DISPATCH_DIM(t, k_points, grid_for(n), t->dev, (long long)n, (const double *)d_pts, d_in, d_va);
    if (inside) HIPCHK(hipMemcpyAsync(inside, d_in, (size_t)n, hipMemcpyDeviceToHost, t->stream));
    if (valid) HIPCHK(hipMemcpyAsync(valid, d_va, (size_t)n, hipMemcpyDeviceToHost, t->stream));
    int rc = sync_check(t);
    (void)hipFree(d_pts);
    (void)hipFree(d_in);
    (void)hipFree(d_va);
    return rc;
}

extern "C" int nirrt_near(nirrt_tree *t, const double *node_new, int64_t new_idx, int64_t *k, int64_t *idx_out, int64_t cap)
{
    if (!t || !node_new || !k) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    DISPATCH_DIM(t, k_near, 1, t->dev, node_new[0], node_new[1], t->dim == 3 ? node_new[2] : 0., (int)new_idx,
                 &t->scratch_dev->i[0]);
    int rc = sync_check(t);
    if (rc) return rc;
    int kk = t->scratch->i[0];
        *k = kk;
    if (idx_out && kk > 0) {
        std::vector<int> tmp((size_t)kk);
        HIPCHK(hipMemcpy(tmp.data(), t->host.nr_idx, sizeof(int) * (size_t)kk, hipMemcpyDeviceToHost));
        for (int i = 0; i < kk && i < cap; i++) idx_out[i] = tmp[(size_t)i];
    }
    return NIRRT_OK;
}

extern "C" int nirrt_cost(nirrt_tree *t, int64_t n_idx, const int64_t *idx, double *out)
{
    if (!t || n_idx < 0 || (n_idx > 0 && (!idx || !out))) return NIRRT_E_ARG;
    if (n_idx == 0) return NIRRT_OK;
    HIPCHK(hipSetDevice(t->device));
    long long *d_idx = nullptr;
    double *d_out = nullptr;
    HIPCHK(hipMalloc(&d_idx, sizeof(long long) * (size_t)n_idx));
    HIPCHK(hipMalloc(&d_out, sizeof(double) * (size_t)n_idx));
    HIPCHK(hipMemcpyAsync(d_idx, idx, sizeof(long long) * (size_t)n_idx, hipMemcpyHostToDevice, t->stream));
    DISPATCH_DIM(t, k_cost, grid_for(n_idx), t->dev, (long long)n_idx, (const long long *)d_idx, d_out);
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)n_idx, hipMemcpyDeviceToHost, t->stream));
    int rc = sync_check(t);
    (void)hipFree(d_idx);
    (void)hipFree(d_out);
    return rc;
}

extern "C" int nirrt_search_goal_parent(nirrt_tree *t, int64_t *idx, double *path_len)
{
    if (!t || !idx) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    DISPATCH_DIM(t, k_goal_parent, 1, t->dev, &t->scratch_dev->i[0], &t->scratch_dev->d[0]);
    int rc = sync_check(t);
    if (rc) return rc;
    *idx = t->scratch->i[0];
    if (path_len) *path_len = t->scratch->d[0];
    return NIRRT_OK;
}

extern "C" int nirrt_best_solution(nirrt_tree *t, double *c_best, int64_t *x_best)
{
    if (!t || !c_best || !x_best) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    DISPATCH_DIM(t, k_best_solution, 1, t->dev, &t->scratch_dev->i[0], &t->scratch_dev->d[0]);
    int rc = sync_check(t);
    if (rc) return rc;
    *x_best = t->scratch->i[0];
    *c_best = t->scratch->d[0];
    return NIRRT_OK;
}

extern "C" int nirrt_solutions(nirrt_tree *t, int64_t *n_sol, int64_t *out, int64_t cap)
{
    if (!t || !n_sol) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    TreeDev tmp;
    HIPCHK(hipMemcpyAsync(&tmp, t->dev, sizeof(TreeDev), hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    *n_sol = tmp.n_sol;
    if (out && tmp.n_sol > 0) {
        std::vector<int> s((size_t)tmp.n_sol);
        HIPCHK(hipMemcpy(s.data(), t->host.sol, sizeof(int) * (size_t)tmp.n_sol, hipMemcpyDeviceToHost));
        for (int i = 0; i < tmp.n_sol && i < cap; i++) out[i] = s[(size_t)i];
    }
    return NIRRT_OK;
}

static int do_step(nirrt_tree *t, const double *p, int host_steer, int64_t nearest_idx, uint32_t flags, nirrt_step_result *res)
{
    if (!t || !p || !res) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    DISPATCH_DIM(t, k_step, 1, t->dev, p[0], p[1], t->dim == 3 ? p[2] : 0., host_steer, (int)nearest_idx, (unsigned)flags,
                 &t->scratch_dev->step);
    int rc = sync_check(t);
    if (rc) return rc;
    *res = t->scratch->step;
    if (res->status) { g_err = "capacity exceeded inside step"; return res->status; }
    return NIRRT_OK;
}

extern "C" int nirrt_step(nirrt_tree *t, const double *node_rand, uint32_t flags, nirrt_step_result *res)
{
    return do_step(t, node_rand, 0, 0, flags, res);
}

extern "C" int nirrt_extend(nirrt_tree *t, int64_t nearest_idx, const double *node_new, uint32_t flags, nirrt_step_result *res)
{
    if (t && (nearest_idx < 0 || nearest_idx >= t->cap)) return NIRRT_E_ARG;
    return do_step(t, node_new, 1, nearest_idx, flags, res);
}

extern "C" int nirrt_set_informed(nirrt_tree *t, double c_min, const double *x_center, const double *C)
{
    if (!t || !x_center || !C) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    // patch the three fields in the device descriptor (the kernels own the rest of it)
    t->host.c_min = c_min;
    for (int k = 0; k < 3; k++) t->host.x_center[k] = k < t->dim ? x_center[k] : 0.;
    for (int k = 0; k < 9; k++) t->host.CL_C[k] = C[k];
    size_t off = offsetof(TreeDev, c_min);
    HIPCHK(hipMemcpyAsync((char *)t->dev + off, (char *)&t->host + off, sizeof(TreeDev) - off, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return NIRRT_OK;
}

extern "C" int nirrt_set_cloud(nirrt_tree *t, int64_t n, const double *pts, double sample_rate, double update_cost_ratio,
                               double c_update)
{
    if (!t || n < 0 || (n > 0 && !pts)) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    if (t->pc_dev) { (void)hipFree(t->pc_dev); t->pc_dev = nullptr; }
    if (n > 0) {
        HIPCHK(hipMalloc(&t->pc_dev, sizeof(double) * (size_t)n * t->dim));
        HIPCHK(hipMemcpy(t->pc_dev, pts, sizeof(double) * (size_t)n * t->dim, hipMemcpyHostToDevice));
    }
    t->host.pc = t->pc_dev;
    t->host.pc_n = (int)n;
    t->host.pad2 = 0;
    t->host.pc_rate = sample_rate;
    t->host.pc_ratio = update_cost_ratio;
    t->host.c_update = c_update;
    size_t off = offsetof(TreeDev, pc);
    HIPCHK(hipMemcpyAsync((char *)t->dev + off, (char *)&t->host + off, sizeof(TreeDev) - off, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return NIRRT_OK;
}

static int run_sampling(nirrt_tree *const *trees, int32_t n_trees, const nirrt_run_args *a)
{
    nirrt_tree *t0 = trees[0];
    const int D = t0->dim;
    hipStream_t st = t0->stream;
    if (!a->np_words || !a->n_np || !a->np_used || !a->py_used || !a->iters_done) {
        g_err = "nirrt_run: sampling mode needs np_words, n_np, np_used, py_used, iters_done";
        return NIRRT_E_ARG;
    }
    const bool need_py = (a->flags & NIRRT_F_IRRT) && D == 2;
    if (need_py && (!a->py_words || !a->n_py)) { g_err = "nirrt_run: 2D IRRT* sampling needs py_words"; return NIRRT_E_ARG; }
    std::vector<void *> to_free;
    auto cleanup = [&]() { for (void *p : to_free) (void)hipFree(p); };
#define HIPCHK_R(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            cleanup();                                                                        \
            return NIRRT_E_HIP;                                                               \
        }                                                                                     \
    } while (0)
    auto dalloc = [&](size_t bytes, void **out) -> hipError_t {
        hipError_t e = hipMalloc(out, bytes ? bytes : 8);
        if (e == hipSuccess) to_free.push_back(*out);
        return e;
    };
    std::vector<TreeDev *> ptrs((size_t)n_trees);
    std::vector<long long> scan0((size_t)n_trees, 0), alg0((size_t)n_trees, 0);
    for (int i = 0; i < n_trees; i++) {
        ptrs[(size_t)i] = trees[i]->dev;
        TreeDev tmp;
        HIPCHK_R(hipMemcpy(&tmp, trees[i]->dev, sizeof(TreeDev), hipMemcpyDeviceToHost));
        scan0[(size_t)i] = tmp.scan_elems;
        alg0[(size_t)i] = tmp.alg_elems;
    }
    // word streams -> device (one slab per generator) unless they already live there
    std::vector<const unsigned *> npp((size_t)n_trees), pyp((size_t)n_trees, nullptr);
    std::vector<long long> nnp((size_t)n_trees), npy((size_t)n_trees, 0);
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t *const *src = pass == 0 ? a->np_words : a->py_words;
        const int64_t *cnt = pass == 0 ? a->n_np : a->n_py;
        if (!src) continue;
        std::vector<const unsigned *> &dst = pass == 0 ? npp : pyp;
        std::vector<long long> &dn = pass == 0 ? nnp : npy;
        size_t total = 0;
        for (int i = 0; i < n_trees; i++) { dn[(size_t)i] = cnt[i]; total += (size_t)cnt[i]; }
        if (a->inputs_on_device) {
            for (int i = 0; i < n_trees; i++) dst[(size_t)i] = src[i];
        } else {
            unsigned *slab = nullptr;
            HIPCHK_R(dalloc(sizeof(unsigned) * total, (void **)&slab));
            size_t off = 0;
            for (int i = 0; i < n_trees; i++) {
                if (cnt[i] > 0) HIPCHK_R(hipMemcpyAsync(slab + off, src[i], sizeof(unsigned) * (size_t)cnt[i], hipMemcpyHostToDevice, st));
                dst[(size_t)i] = slab + off;
                off += (size_t)cnt[i];
            }
        }
    }
    TreeDev **d_ptrs = nullptr;
    const unsigned **d_npp = nullptr, **d_pyp = nullptr;
    long long *d_nnp = nullptr, *d_npy = nullptr, *d_npu = nullptr, *d_pyu = nullptr, *d_done = nullptr;
    int *d_stop = nullptr;
    double *d_trace = nullptr;
    const size_t nt = (size_t)n_trees;
    HIPCHK_R(dalloc(sizeof(void *) * nt, (void **)&d_ptrs));
    HIPCHK_R(dalloc(sizeof(void *) * nt, (void **)&d_npp));
    HIPCHK_R(dalloc(sizeof(void *) * nt, (void **)&d_pyp));
    HIPCHK_R(dalloc(sizeof(long long) * nt, (void **)&d_nnp));
    HIPCHK_R(dalloc(sizeof(long long) * nt, (void **)&d_npy));
    HIPCHK_R(dalloc(sizeof(long long) * nt, (void **)&d_npu));
    HIPCHK_R(dalloc(sizeof(long long) * nt, (void **)&d_pyu));
    HIPCHK_R(dalloc(sizeof(long long) * nt, (void **)&d_done));
    HIPCHK_R(dalloc(sizeof(int) * nt, (void **)&d_stop));
    if (a->cost_trace) HIPCHK_R(dalloc(sizeof(double) * nt * (size_t)a->iters, (void **)&d_trace));
    HIPCHK_R(hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(void *) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_npp, npp.data(), sizeof(void *) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_pyp, pyp.data(), sizeof(void *) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_nnp, nnp.data(), sizeof(long long) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_npy, npy.data(), sizeof(long long) * nt, hipMemcpyHostToDevice, st));
    RunSampleDev rd;
    rd.flags = a->flags; rd.pad = 0; rd.iters = a->iters;
    rd.np_words = d_npp; rd.n_np = d_nnp; rd.py_words = a->py_words ? d_pyp : nullptr; rd.n_py = d_npy;
    rd.np_used = d_npu; rd.py_used = d_pyu; rd.cost_trace = d_trace; rd.iters_done = d_done; rd.stop_code = d_stop;
    hipEvent_t e0, e1;
    HIPCHK_R(hipEventCreate(&e0));
    HIPCHK_R(hipEventCreate(&e1));
    HIPCHK_R(hipEventRecord(e0, st));
    if (D == 2) hipLaunchKernelGGL(k_run_sample<2>, dim3(n_trees), dim3(NT), 0, st, (TreeDev *const *)d_ptrs, rd);
    else hipLaunchKernelGGL(k_run_sample<3>, dim3(n_trees), dim3(NT), 0, st, (TreeDev *const *)d_ptrs, rd);
    HIPCHK_R(hipEventRecord(e1, st));
    HIPCHK_R(hipGetLastError());
    HIPCHK_R(hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK_R(hipEventElapsedTime(&ms, e0, e1));
    if (a->kernel_ms) *a->kernel_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::vector<long long> done(nt), npu(nt), pyu(nt);
    std::vector<int> stop(nt);
    HIPCHK_R(hipMemcpy(done.data(), d_done, sizeof(long long) * nt, hipMemcpyDeviceToHost));
    HIPCHK_R(hipMemcpy(npu.data(), d_npu, sizeof(long long) * nt, hipMemcpyDeviceToHost));
    HIPCHK_R(hipMemcpy(pyu.data(), d_pyu, sizeof(long long) * nt, hipMemcpyDeviceToHost));
    HIPCHK_R(hipMemcpy(stop.data(), d_stop, sizeof(int) * nt, hipMemcpyDeviceToHost));
    if (a->cost_trace) HIPCHK_R(hipMemcpy(a->cost_trace, d_trace, sizeof(double) * nt * (size_t)a->iters, hipMemcpyDeviceToHost));
    int rc_all = NIRRT_OK;
    for (int i = 0; i < n_trees; i++) {
        a->iters_done[i] = done[(size_t)i];
        a->np_used[i] = npu[(size_t)i];
        a->py_used[i] = pyu[(size_t)i];
        if (a->status) a->status[i] = stop[(size_t)i];
        if (a->scan_elems || a->alg_elems) {
            TreeDev tmp;
            HIPCHK_R(hipMemcpy(&tmp, trees[i]->dev, sizeof(TreeDev), hipMemcpyDeviceToHost));
            if (a->scan_elems) a->scan_elems[i] = tmp.scan_elems - scan0[(size_t)i];
            if (a->alg_elems) a->alg_elems[i] = tmp.alg_elems - alg0[(size_t)i];
        }
        if (stop[(size_t)i] == NIRRT_E_CAPACITY) rc_all = NIRRT_E_CAPACITY;
    }
    cleanup();
    return rc_all;   // NIRRT_E_STREAM is reported per tree in status[] (the caller refills and resumes)
#undef HIPCHK_R
}

/* debug: per-phase tick counters (all zero unless built with -DNIRRT_PROFILE) */
extern "C" int nirrt_debug_prof(nirrt_tree *t, int64_t *out16)
{
    if (!t || !out16) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    TreeDev tmp;
    HIPCHK(hipMemcpy(&tmp, t->dev, sizeof(TreeDev), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) out16[i] = tmp.prof[i];
    return NIRRT_OK;
}

extern "C" int nirrt_run(nirrt_tree *const *trees, int32_t n_trees, const nirrt_run_args *a)
{
    if (!trees || n_trees <= 0 || !a || a->iters < 0) return NIRRT_E_ARG;
    nirrt_tree *t0 = trees[0];
    for (int i = 0; i < n_trees; i++) {
        if (!trees[i] || trees[i]->device != t0->device || trees[i]->dim != t0->dim) {
            g_err = "nirrt_run: all trees must share device and dim";
            return NIRRT_E_ARG;
        }
    }
    HIPCHK(hipSetDevice(t0->device));
    for (int i = 1; i < n_trees; i++) HIPCHK(hipStreamSynchronize(trees[i]->stream));
    if (!a->samples) return run_sampling(trees, n_trees, a);
    const int D = t0->dim;
    hipStream_t st = t0->stream;
    for (int i = 1; i < n_trees; i++) HIPCHK(hipStreamSynchronize(trees[i]->stream));
    std::vector<TreeDev *> ptrs((size_t)n_trees);
    for (int i = 0; i < n_trees; i++) ptrs[(size_t)i] = trees[i]->dev;
    TreeDev **d_ptrs = nullptr;
    double *d_samples = nullptr, *d_trace = nullptr;
    long long *d_done = nullptr;
    size_t sbytes = sizeof(double) * (size_t)n_trees * (size_t)a->iters * D;
    HIPCHK(hipMalloc(&d_ptrs, sizeof(TreeDev *) * (size_t)n_trees));
    std::vector<long long> scan0((size_t)n_trees, 0), alg0((size_t)n_trees, 0);
    for (int i = 0; i < n_trees; i++) {
        TreeDev tmp;
        HIPCHK(hipMemcpy(&tmp, trees[i]->dev, sizeof(TreeDev), hipMemcpyDeviceToHost));
        scan0[(size_t)i] = tmp.scan_elems;
        alg0[(size_t)i] = tmp.alg_elems;
    }
    if (a->inputs_on_device) d_samples = const_cast<double *>(a->samples);
    else HIPCHK(hipMalloc(&d_samples, sbytes ? sbytes : 8));
    HIPCHK(hipMalloc(&d_done, sizeof(long long) * (size_t)n_trees));
    if (a->cost_trace) HIPCHK(hipMalloc(&d_trace, sizeof(double) * (size_t)n_trees * (size_t)a->iters));
    HIPCHK(hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(TreeDev *) * (size_t)n_trees, hipMemcpyHostToDevice, st));
    if (!a->inputs_on_device) HIPCHK(hipMemcpyAsync(d_samples, a->samples, sbytes, hipMemcpyHostToDevice, st));
    RunDev rd;
    rd.flags = a->flags; rd.pad = 0; rd.iters = a->iters; rd.samples = d_samples; rd.cost_trace = d_trace; rd.iters_done = d_done;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
    if (D == 2) hipLaunchKernelGGL(k_run_replay<2>, dim3(n_trees), dim3(NT), 0, st, (TreeDev *const *)d_ptrs, rd);
    else hipLaunchKernelGGL(k_run_replay<3>, dim3(n_trees), dim3(NT), 0, st, (TreeDev *const *)d_ptrs, rd);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (a->kernel_ms) *a->kernel_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::vector<long long> done((size_t)n_trees);
    HIPCHK(hipMemcpy(done.data(), d_done, sizeof(long long) * (size_t)n_trees, hipMemcpyDeviceToHost));
    if (a->cost_trace)
        HIPCHK(hipMemcpy(a->cost_trace, d_trace, sizeof(double) * (size_t)n_trees * (size_t)a->iters, hipMemcpyDeviceToHost));
    int rc_all = NIRRT_OK;
    for (int i = 0; i < n_trees; i++) {
        if (a->iters_done) a->iters_done[i] = done[(size_t)i];
        if (a->np_used) a->np_used[i] = 0;
        if (a->py_used) a->py_used[i] = 0;
        TreeDev tmp;
        HIPCHK(hipMemcpy(&tmp, trees[i]->dev, sizeof(TreeDev), hipMemcpyDeviceToHost));
        if (a->status) a->status[i] = tmp.status;
        if (a->scan_elems) a->scan_elems[i] = tmp.scan_elems - scan0[(size_t)i];
        if (a->alg_elems) a->alg_elems[i] = tmp.alg_elems - alg0[(size_t)i];
        if (tmp.status) rc_all = tmp.status;
    }
    (void)hipFree(d_ptrs);
    if (!a->inputs_on_device) (void)hipFree(d_samples);
    (void)hipFree(d_done);
    if (d_trace) (void)hipFree(d_trace);
    return rc_all;
}
