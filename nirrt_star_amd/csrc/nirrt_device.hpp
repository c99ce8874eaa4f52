// nirrt_device.hpp — gfx950 device code of the RRT*/IRRT* inner loop.
//
// Execution model: ONE workgroup (NT threads = NT/64 wave64s) owns ONE tree.  Every O(n) pass
// (nearest, Near) is a coalesced grid-stride scan of the SoA coordinate arrays by the whole
// workgroup with a wave64 __shfl_xor reduction + an LDS cross-wave step; everything O(k) (fan of
// segment tests, parent-chain cost walks, choose-parent, rewire) runs lane-parallel out of LDS.
// The same device functions back the one-kernel-per-primitive entry points, the fused
// one-iteration kernel and the persistent many-trees loop.
//
// Arithmetic: float64 everywhere, compiled with -ffp-contract=off; each distance uses the formula
// the reference resolves to at that call site:
//   np.hypot            -> hypot_np()   glibc 2.35 __hypot, non-FMA kernel, IEEE ops only
//   math.hypot          -> hypot_py<D>()  CPython 3.10 vector_norm
//   np.linalg.norm axis -> norm_axis<D>() sqrt of left-to-right unfused sum of squares
//   np.linalg.norm 1-D  -> norm_1d<D>()   sqrt of the BLAS ddot forward FMA chain
//   np.dot (2-vectors)  -> dot_blas<D>()
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nirrt_hip.h"

#define NT 1024            // threads per workgroup (16 wave64)
#define NW (NT / 64)
#define NEAR_CAP NIRRT_NEAR_CAPACITY
#define MAX_OBS NIRRT_MAX_OBSTACLES

// ------------------------------------------------------------------------------------------------
// per-tree state in HBM
// ------------------------------------------------------------------------------------------------
struct TreeDev {
    // flat SoA vertex store
    double *c[3];   // x[cap], y[cap], z[cap]
    int *parent;    // parent[cap], parent[0] = 0
    int cap;
    int n;          // num_vertices
    int dim;
    int status;     // sticky NIRRT_E_* code
    // IRRT*: path_solutions (goal-parent indices, duplicates allowed)
    int *sol;
    int n_sol;
    int cap_sol;
    // RRT*: vertices within step_len of the goal (ascending index) with their distance and the
    // result of the vertex->goal segment test (vertices never move, so this is append-only)
    int *gc_idx;
    double *gc_dist;
    unsigned char *gc_col;
    int n_gc;
    int pad0;
    // Near radius r(n) = min(gamma*sqrt(ln n/n), step_len) [2D] / cube root [3D], tabulated on the
    // host with glibc (rrt_star_2d.py:133, rrt_star_3d.py:134); index = num_vertices
    const double *near_r;
    // problem constants
    double start[3], goal[3];
    double step_len, clearance;
    double lo[3], hi[3];
    int n_round, n_box;
    double rnd[MAX_OBS][4];  // cx, cy, cz, r
    double box[MAX_OBS][6];  // x, y, z, w, h, d
    // informed sampling constants (IRRT*.init, irrt_star_2d.py:35-40 / irrt_star_3d.py:32-36)
    double c_min;
    double x_center[3];
    double Crot[9];          // rotation to world frame, row-major 3x3
};

// ------------------------------------------------------------------------------------------------
// LDS working set of one workgroup
// ------------------------------------------------------------------------------------------------
struct Lds {
    int n_round, n_box;
    double rnd[MAX_OBS][4];
    double box[MAX_OBS][6];
    double red_val[NW];
    int red_idx[NW];
    int wave_tot[NW];
    int counter;
    int flag;
    int near_idx[NEAR_CAP];
    int near_idx2[NEAR_CAP];
    double near_dist[NEAR_CAP];
    double near_dist2[NEAR_CAP];
    double near_c0[NEAR_CAP];  // cost(j)
    double near_c1[NEAR_CAP];  // cost(new) if parent[new] were j
    int near_col[NEAR_CAP];
    double bc_d[8];
    int bc_i[8];
};

// ------------------------------------------------------------------------------------------------
// distance primitives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double hypot_np(double x, double y)
{
    // glibc 2.35 sysdeps/ieee754/dbl-64/e_hypot.c (non-FMA kernel).  Coordinates are finite.
    const double SCALE = 0x1p-600, LARGE_VAL = 0x1p+511, TINY_VAL = 0x1p-459, EPS = 0x1p-54;
    x = fabs(x);
    y = fabs(y);
    double ax = x < y ? y : x;
    double ay = x < y ? x : y;
    double s = 1.0;
    if (ax > LARGE_VAL) {
        if (ay <= ax * EPS) return ax + ay;
        ax *= SCALE; ay *= SCALE; s = 0x1p+600;
    } else if (ay < TINY_VAL) {
        if (ax >= ay / EPS) return ax + ay;
        ax /= SCALE; ay /= SCALE; s = SCALE;
    } else if (ax >= ay / EPS) {
        return ax + ay;
    }
    double h = __builtin_sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return s == 1.0 ? h : h * s;
}

template <int D>
__device__ __forceinline__ double hypot_py(const double *d)
{
    // CPython 3.10 Modules/mathmodule.c vector_norm()
    const double T27 = 134217729.0;
    double vec[D], mx = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) {
        vec[i] = fabs(d[i]);
        if (vec[i] > mx) mx = vec[i];
    }
    if (mx == 0.0) return mx;
    int max_e;
    (void)frexp(mx, &max_e);
    double scale = ldexp(1.0, -max_e);
    double x, oldcsum, csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0, t, hi, lo, h;
#pragma unroll
    for (int i = 0; i < D; i++) {
        x = vec[i] * scale;
        t = x * T27;
        hi = t - (t - x);
        lo = x - hi;
        x = hi * hi;
        oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
        x = 2.0 * hi * lo;
        oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
        frac3 += lo * lo;
    }
    h = __builtin_sqrt(csum - 1.0 + (frac1 + frac2 + frac3));
    x = h;
    t = x * T27;
    hi = t - (t - x);
    lo = x - hi;
    x = -hi * hi;
    oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
    x = -2.0 * hi * lo;
    oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
    x = -lo * lo;
    oldcsum = csum; csum += x; frac3 += (oldcsum - csum) + x;
    x = csum - 1.0 + (frac1 + frac2 + frac3);
    return (h + x / (2.0 * h)) / scale;
}

template <int D>
__device__ __forceinline__ double norm_axis(const double *d)
{
    double s = d[0] * d[0] + d[1] * d[1];
    if (D == 3) s = s + d[2] * d[2];
    return __builtin_sqrt(s);
}

template <int D>
__device__ __forceinline__ double dot_blas(const double *u, const double *w)
{
    double s = u[0] * w[0];
    s = __builtin_fma(u[1], w[1], s);
    if (D == 3) s = __builtin_fma(u[2], w[2], s);
    return s;
}

template <int D>
__device__ __forceinline__ double norm_1d(const double *d)
{
    return __builtin_sqrt(dot_blas<D>(d, d));
}

// distance of the O(n) scans and of choose_parent / rewire / goal scan
template <int D>
__device__ __forceinline__ double dist_scan(const double *d)
{
    if (D == 2) return hypot_np(d[0], d[1]);
    return norm_axis<3>(d);
}

// ------------------------------------------------------------------------------------------------
// segment / point tests against ONE obstacle (the AABB prefilter of the reference is kept: it
// decides which obstacles reach the exact test, and the exact tests are not monotone in it)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool line_intersection(const double *l1, const double *l2)
{
    // collision_check_utils.py:8-30; l = {x0,y0,x1,y1}
    double xd0 = l1[0] - l1[2], xd1 = l2[0] - l2[2];
    double yd0 = l1[1] - l1[3], yd1 = l2[1] - l2[3];
    double div = xd0 * yd1 - xd1 * yd0;
    if (div == 0) return false;
    double d0 = l1[0] * l1[3] - l1[1] * l1[2];
    double d1 = l2[0] * l2[3] - l2[1] * l2[2];
    double x = (d0 * xd1 - d1 * xd0) / div;
    double y = (d0 * yd1 - d1 * yd0) / div;
    const double eps = 1e-6;
    return fmin(l1[0], l1[2]) - eps <= x && x <= fmax(l1[0], l1[2]) + eps &&
           fmin(l1[1], l1[3]) - eps <= y && y <= fmax(l1[1], l1[3]) + eps &&
           fmin(l2[0], l2[2]) - eps <= x && x <= fmax(l2[0], l2[2]) + eps &&
           fmin(l2[1], l2[3]) - eps <= y && y <= fmax(l2[1], l2[3]) + eps;
}

// 2D circle: AABB prefilter (collision_check_utils.py:177-190) + check_collision_line_single_circle (:33-60)
__device__ __forceinline__ bool seg_round_2d(const double *a, const double *b, const double *c, double clr)
{
    double cx = c[0], cy = c[1], cr = c[3];
    double lx0 = fmin(a[0], b[0]), ly0 = fmin(a[1], b[1]), lx1 = fmax(a[0], b[0]), ly1 = fmax(a[1], b[1]);
    double X1 = cx - cr - clr, Y1 = cy - cr - clr, X2 = cx + cr + clr, Y2 = cy + cr + clr;
    if (!(lx0 <= X2 && lx1 >= X1 && ly0 <= Y2 && ly1 >= Y1)) return false;
    double R = cr + clr;
    double lv[2] = {b[0] - a[0], b[1] - a[1]};
    double L = norm_1d<2>(lv);
    if (L == 0) {
        double pc[2] = {a[0] - cx, a[1] - cy};
        return norm_1d<2>(pc) <= cr + clr;
    }
    double dir[2] = {lv[0] / L, lv[1] / L};
    double sc[2] = {cx - a[0], cy - a[1]};
    double proj = dot_blas<2>(sc, dir);
    double t = fmin(fmax(proj, 0.0), L);
    double cp[2] = {t * dir[0] + a[0], t * dir[1] + a[1]};
    double dc[2] = {cx - cp[0], cy - cp[1]};
    return norm_1d<2>(dc) <= R;
}

// 2D rectangle: AABB prefilter (:191-204) + check_collision_line_single_rectangle (:98-130)
__device__ __forceinline__ bool seg_box_2d(const double *a, const double *b, const double *r, double clr)
{
    double rx = r[0], ry = r[1], rw = r[3], rh = r[4];
    double x0 = rx - clr, y0 = ry - clr, x1 = rx + rw + clr, y1 = ry + rh + clr;
    double lx0 = fmin(a[0], b[0]), ly0 = fmin(a[1], b[1]), lx1 = fmax(a[0], b[0]), ly1 = fmax(a[1], b[1]);
    if (!(lx0 <= x1 && lx1 >= x0 && ly0 <= y1 && ly1 >= y0)) return false;
    if (x0 <= a[0] && a[0] <= x1 && y0 <= a[1] && a[1] <= y1) return true;
    if (x0 <= b[0] && b[0] <= x1 && y0 <= b[1] && b[1] <= y1) return true;
    double l1[4] = {a[0], a[1], b[0], b[1]};
    double e0[4] = {x0, y0, x1, y0}, e1[4] = {x1, y0, x1, y1}, e2[4] = {x1, y1, x0, y1}, e3[4] = {x0, y1, x0, y0};
    return line_intersection(l1, e0) || line_intersection(l1, e1) || line_intersection(l1, e2) ||
           line_intersection(l1, e3);
}

// 3D ball: AABB prefilter (collision_check_utils_3d.py:169-181) + check_collision_line_single_ball (:3-38)
__device__ __forceinline__ bool seg_round_3d(const double *p0, const double *p1, const double *c, double clr)
{
    double cr = c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double l0 = fmin(p0[k], p1[k]), l1 = fmax(p0[k], p1[k]);
        double A1 = c[k] - cr - clr, A2 = c[k] + cr + clr;
        if (!(l0 <= A2 && l1 >= A1)) return false;
    }
    double r = cr + clr;
    double l[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    if (norm_1d<3>(l) == 0) {
        double pc[3] = {p0[0] - c[0], p0[1] - c[1], p0[2] - c[2]};
        return norm_1d<3>(pc) <= cr + clr;
    }
    double d1[3] = {c[0] - p0[0], c[1] - p0[1], c[2] - p0[2]};
    double t = (1 / (l[0] * l[0] + l[1] * l[1] + l[2] * l[2])) * (l[0] * d1[0] + l[1] * d1[1] + l[2] * d1[2]);
    double r2 = r * r;  // reference: numpy scalar r**2 (= pow(r,2)); identical for the integer-valued radii used
    if (t <= 0) {
        return d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2] <= r2;
    } else if (t >= 1) {
        double d2[3] = {c[0] - p1[0], c[1] - p1[1], c[2] - p1[2]};
        return d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2] <= r2;
    } else if (0 < t && t < 1) {
        double x[3] = {p0[0] + t * l[0], p0[1] + t * l[1], p0[2] + t * l[2]};
        double k[3] = {c[0] - x[0], c[1] - x[1], c[2] - x[2]};
        return k[0] * k[0] + k[1] * k[1] + k[2] * k[2] <= r2;
    }
    return false;
}

// 3D box: AABB prefilter (:182-203) + check_collision_line_single_box (:41-84)
__device__ __forceinline__ bool seg_box_3d(const double *p0, const double *p1, const double *b, double clr)
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double l0 = fmin(p0[k], p1[k]), l1 = fmax(p0[k], p1[k]);
        double A1 = b[k] - clr, A2 = b[k] + b[3 + k] + clr;
        if (!(l0 <= A2 && l1 >= A1)) return false;
    }
    double mid[3] = {(p0[0] + p1[0]) / 2, (p0[1] + p1[1]) / 2, (p0[2] + p1[2]) / 2};
    double dir[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    double dist = norm_1d<3>(dir);
    if (dist == 0) {
        return b[0] - clr <= p0[0] && p0[0] <= b[0] + b[3] + clr && b[1] - clr <= p0[1] && p0[1] <= b[1] + b[4] + clr &&
               b[2] - clr <= p0[2] && p0[2] <= b[2] + b[5] + clr;
    }
    double I[3] = {dir[0] / dist, dir[1] / dist, dir[2] / dist};
    double hl = dist / 2;
    double P[3] = {b[0] + b[3] / 2, b[1] + b[4] / 2, b[2] + b[5] / 2};
    double E[3] = {b[3] / 2 + clr, b[4] / 2 + clr, b[5] / 2 + clr};
    double T[3] = {P[0] - mid[0], P[1] - mid[1], P[2] - mid[2]};
    if (fabs(T[0]) > (E[0] + hl * fabs(I[0]))) return false;
    if (fabs(T[1]) > (E[1] + hl * fabs(I[1]))) return false;
    if (fabs(T[2]) > (E[2] + hl * fabs(I[2]))) return false;
    double r;
    r = E[1] * fabs(I[2]) + E[2] * fabs(I[1]);
    if (fabs(T[1] * I[2] - T[2] * I[1]) > r) return false;
    r = E[0] * fabs(I[2]) + E[2] * fabs(I[0]);
    if (fabs(T[2] * I[0] - T[0] * I[2]) > r) return false;
    r = E[0] * fabs(I[1]) + E[1] * fabs(I[0]);
    if (fabs(T[0] * I[1] - T[1] * I[0]) > r) return false;
    return true;
}

// segment vs obstacle #o of the LDS tables (o < n_round: round, else box)
template <int D>
__device__ __forceinline__ bool seg_obstacle(const Lds &s, int o, const double *a, const double *b, double clr)
{
    if (o < s.n_round) {
        if (D == 2) return seg_round_2d(a, b, s.rnd[o], clr);
        return seg_round_3d(a, b, s.rnd[o], clr);
    }
    o -= s.n_round;
    if (D == 2) return seg_box_2d(a, b, s.box[o], clr);
    return seg_box_3d(a, b, s.box[o], clr);
}

// whole segment test by ONE lane (used in lane-parallel fans over many segments)
template <int D>
__device__ __forceinline__ bool seg_all(const Lds &s, const double *a, const double *b, double clr)
{
    int M = s.n_round + s.n_box;
    for (int o = 0; o < M; o++)
        if (seg_obstacle<D>(s, o, a, b, clr)) return true;
    return false;
}

// points_in_circles / points_in_balls: strict <  (collision_check_utils.py:292, _3d.py:299)
// points_in_rectangles / points_in_boxes: inclusive (:254, _3d.py:260)
template <int D>
__device__ __forceinline__ bool point_in_obs(const Lds &s, const double *p, double clr)
{
    for (int i = 0; i < s.n_round; i++) {
        double rc = s.rnd[i][3] + clr;
        double q = (p[0] - s.rnd[i][0]) * (p[0] - s.rnd[i][0]) + (p[1] - s.rnd[i][1]) * (p[1] - s.rnd[i][1]);
        if (D == 3) q = q + (p[2] - s.rnd[i][2]) * (p[2] - s.rnd[i][2]);
        if (q < rc * rc) return true;
    }
    for (int i = 0; i < s.n_box; i++) {
        bool in = true;
#pragma unroll
        for (int k = 0; k < D; k++) {
            double mx = s.box[i][k] + s.box[i][3 + k] + clr, mn = s.box[i][k] - clr;
            in = in && (mn <= p[k]) && (p[k] <= mx);
        }
        if (in) return true;
    }
    return false;
}

// points_in_range: the range as one rectangle tested with clearance = -clearance (:330-351)
template <int D>
__device__ __forceinline__ bool point_in_range(const TreeDev &t, const double *p)
{
    double clr = -t.clearance;
    bool in = true;
#pragma unroll
    for (int k = 0; k < D; k++) {
        double w = t.hi[k] - t.lo[k];
        double mx = t.lo[k] + w + clr, mn = t.lo[k] - clr;
        in = in && (mn <= p[k]) && (p[k] <= mx);
    }
    return in;
}

// ------------------------------------------------------------------------------------------------
// workgroup collectives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_obstacles(Lds &s, const TreeDev &t)
{
    int tid = threadIdx.x;
    if (tid == 0) { s.n_round = t.n_round; s.n_box = t.n_box; }
    for (int i = tid; i < t.n_round * 4; i += NT) s.rnd[i / 4][i % 4] = t.rnd[i / 4][i % 4];
    for (int i = tid; i < t.n_box * 6; i += NT) s.box[i / 6][i % 6] = t.box[i / 6][i % 6];
    __syncthreads();
}

// lexicographic (value, index) minimum over the workgroup; every thread gets the result.
// Ties keep the LOWEST index (np.argmin).  Threads with nothing pass idx = INT_MAX, v = +inf.
__device__ __forceinline__ void block_argmin(Lds &s, double &v, int &idx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(v, off);
        int oi = __shfl_xor(idx, off);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect red_* reuse
    if (lane == 0) { s.red_val[w] = v; s.red_idx[w] = idx; }
    __syncthreads();
    v = s.red_val[0];
    idx = s.red_idx[0];
#pragma unroll
    for (int i = 1; i < NW; i++) {
        double ov = s.red_val[i];
        int oi = s.red_idx[i];
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// workgroup OR
__device__ __forceinline__ bool block_any(bool p) { return __syncthreads_or(p ? 1 : 0) != 0; }

// ordered compaction: threads with keep get their output slot (ascending thread order); returns total
__device__ __forceinline__ int block_compact(Lds &s, bool keep, int &pos)
{
    unsigned long long m = __ballot(keep);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int pre = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s.wave_tot[w] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        int c = s.wave_tot[i];
        if (i < w) off += c;
        tot += c;
    }
    pos = off + pre;
    return tot;
}

// ------------------------------------------------------------------------------------------------
// tree primitives (workgroup scope)
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void load_vertex(const TreeDev &t, int i, double *v)
{
#pragma unroll
    for (int k = 0; k < D; k++) v[k] = t.c[k][i];
}

// nearest_neighbor: argmin_i dist(q, v_i), lowest index on ties
template <int D>
__device__ __forceinline__ int wg_nearest(Lds &s, const TreeDev &t, int n, const double *q, double &best_d)
{
    double bd = __builtin_inf();
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += NT) {
        double d[D];
#pragma unroll
        for (int k = 0; k < D; k++) d[k] = q[k] - t.c[k][i];
        double h = dist_scan<D>(d);
        if (h < bd) { bd = h; bi = i; }
    }
    block_argmin(s, bd, bi);
    best_d = bd;
    return bi;
}

// cost walk leaf -> root (RRTBase.cost).  Returns cost(idx); if c1 != nullptr also the cost the
// vertex `from` would have if its parent were idx:  ((e(from,idx) + e(idx,p)) + ...)  which is how
// cost(from) sums when walking from `from` (cost() restarts its accumulator at the leaf).
template <int D>
__device__ __forceinline__ double walk_cost(const TreeDev &t, int idx, const double *from, double *c1)
{
    double v[D];
    load_vertex<D>(t, idx, v);
    double acc0 = 0., acc1 = 0.;
    if (c1) {
        double d[D];
#pragma unroll
        for (int k = 0; k < D; k++) d[k] = from[k] - v[k];
        acc1 = hypot_py<D>(d);
    }
    int guard = t.cap + 1;
    while (idx != 0 && guard-- > 0) {
        int p = t.parent[idx];
        double pv[D], d[D];
        load_vertex<D>(t, p, pv);
#pragma unroll
        for (int k = 0; k < D; k++) { d[k] = v[k] - pv[k]; v[k] = pv[k]; }
        double e = hypot_py<D>(d);
        acc0 += e;
        acc1 += e;
        idx = p;
    }
    if (c1) *c1 = acc1;
    return acc0;
}

// steer (new_state).  2D: rrt_star_2d.py:67-78, device atan2/cos/sin; 3D: rrt_star_3d.py:67-78, IEEE only.
template <int D>
__device__ __forceinline__ void steer(const TreeDev &t, const double *from, const double *to, double *out)
{
    double d[D];
#pragma unroll
    for (int k = 0; k < D; k++) d[k] = to[k] - from[k];
    double dist = hypot_py<D>(d);
    double m = dist < t.step_len ? dist : t.step_len;
    if (D == 2) {
        double theta = atan2(d[1], d[0]);
        out[0] = from[0] + m * cos(theta);
        out[1] = from[1] + m * sin(theta);
    } else {
        double dir[3] = {0., 0., 0.};
        if (dist != 0) {
#pragma unroll
            for (int k = 0; k < D; k++) dir[k] = d[k] / dist;
        }
#pragma unroll
        for (int k = 0; k < D; k++) out[k] = from[k] + m * dir[k];
    }
}

// segment test spread over the workgroup: lane o tests obstacle o, OR-reduced
template <int D>
__device__ __forceinline__ bool wg_collision(const Lds &s, const double *a, const double *b, double clr)
{
    int M = s.n_round + s.n_box;
    bool hit = false;
    for (int o = threadIdx.x; o < M; o += NT) hit = hit || seg_obstacle<D>(s, o, a, b, clr);
    return block_any(hit);
}

// Near set of node_new on the current tree (find_near_neighbors).  On return
// s.near_idx[0..k) ascending, s.near_dist[0..k) the matching scan distances.  Returns k (or -1 on
// Near-capacity overflow).
template <int D>
__device__ __forceinline__ int wg_near(Lds &s, const TreeDev &t, int n, const double *node_new, int new_idx)
{
    const double r = t.near_r[n];
    const double clr = t.clearance;
    if (threadIdx.x == 0) s.counter = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) {
        double d[D];
#pragma unroll
        for (int k = 0; k < D; k++) d[k] = node_new[k] - t.c[k][i];
        double h = dist_scan<D>(d);
        if (h <= r) {
            int pos = atomicAdd(&s.counter, 1);
            if (pos < NEAR_CAP) { s.near_idx2[pos] = i; s.near_dist2[pos] = h; }
        }
    }
    __syncthreads();
    int kraw = s.counter;
    if (kraw > NEAR_CAP) return -1;
    // rank sort -> ascending vertex index (np.where order)
    for (int a = threadIdx.x; a < kraw; a += NT) {
        int me = s.near_idx2[a], rank = 0;
        for (int b = 0; b < kraw; b++) rank += (s.near_idx2[b] < me);
        s.near_idx[rank] = me;
        s.near_dist[rank] = s.near_dist2[a];
        s.near_col[a] = 0;
    }
    __syncthreads();
    // fan of segment tests (node_new -> v_j) x obstacles, one (segment, obstacle) pair per lane
    int M = s.n_round + s.n_box;
    if (M > 0) {
        int pairs = kraw * M;
        for (int p = threadIdx.x; p < pairs; p += NT) {
            int j = p / M, o = p - j * M;
            double vj[D];
            load_vertex<D>(t, s.near_idx[j], vj);
            if (seg_obstacle<D>(s, o, node_new, vj, clr)) s.near_col[j] = 1;
        }
    }
    __syncthreads();
    // stable filter: collision-free and != new_idx
    int k = 0;
    for (int base = 0; base < kraw; base += NT) {
        int a = base + threadIdx.x;
        bool keep = a < kraw && s.near_col[a] == 0 && s.near_idx[a] != new_idx;
        int vi = 0;
        double vd = 0;
        if (a < kraw) { vi = s.near_idx[a]; vd = s.near_dist[a]; }
        int pos;
        int tot = block_compact(s, keep, pos);
        if (keep) { s.near_idx2[k + pos] = vi; s.near_dist2[k + pos] = vd; }
        k += tot;
    }
    __syncthreads();
    for (int a = threadIdx.x; a < k; a += NT) {
        s.near_idx[a] = s.near_idx2[a];
        s.near_dist[a] = s.near_dist2[a];
    }
    __syncthreads();
    return k;
}

// find_best_path_solution (irrt_star_2d.py:84-97): argmin_s cost(sol[s]) + Line(v, goal), first minimum
template <int D>
__device__ __forceinline__ void wg_best_solution(Lds &s, const TreeDev &t, double &c_best, int &x_best)
{
    double bv = __builtin_inf();
    int bs = 0x7fffffff;
    int ns = t.n_sol;
    for (int q = threadIdx.x; q < ns; q += NT) {
        int idx = t.sol[q];
        double v[D], d[D];
        load_vertex<D>(t, idx, v);
#pragma unroll
        for (int k = 0; k < D; k++) d[k] = t.goal[k] - v[k];
        double c = walk_cost<D>(t, idx, nullptr, nullptr) + hypot_py<D>(d);
        if (c < bv) { bv = c; bs = q; }
    }
    block_argmin(s, bv, bs);
    c_best = bv;
    x_best = (ns > 0 && bs != 0x7fffffff) ? t.sol[bs] : -1;
    if (ns > 0 && bs == 0x7fffffff) x_best = t.sol[0];  // all +inf cannot happen (costs finite); keep argmin semantics
}

// search_goal_parent (rrt_star_2d.py:101-117) over the maintained candidate list + path length
template <int D>
__device__ __forceinline__ void wg_goal_parent(Lds &s, const TreeDev &t, int &gp, double &path_len)
{
    double bv = __builtin_inf();
    int bq = 0x7fffffff;
    int ng = t.n_gc;
    for (int q = threadIdx.x; q < ng; q += NT) {
        double c = __builtin_inf();
        if (!t.gc_col[q]) c = walk_cost<D>(t, t.gc_idx[q], nullptr, nullptr) + t.gc_dist[q];
        if (c < bv) { bv = c; bq = q; }
    }
    block_argmin(s, bv, bq);
    if (ng == 0) { gp = -1; path_len = __builtin_inf(); return; }
    if (bq == 0x7fffffff) bq = 0;  // every candidate collides: np.argmin of all-inf = 0
    gp = t.gc_idx[bq];
    // get_path_len(extract_path(gp)): sum of segment norms, goal <- gp <- ... <- start
    if (threadIdx.x == 0) {
        double len = 0., prev[D], v[D], d[D];
#pragma unroll
        for (int k = 0; k < D; k++) prev[k] = t.goal[k];
        int i = gp, guard = t.cap + 1;
        for (;;) {
            load_vertex<D>(t, i, v);
#pragma unroll
            for (int k = 0; k < D; k++) { d[k] = prev[k] - v[k]; prev[k] = v[k]; }
            len += norm_axis<D>(d);
            if (i == 0 || guard-- <= 0) break;
            i = t.parent[i];
        }
        s.bc_d[7] = len;
    }
    __syncthreads();
    path_len = s.bc_d[7];
    __syncthreads();
}

// bookkeeping when a vertex (idx, coordinates v) has just been appended: RRT* goal-candidate list.
// Block-uniform control flow; `v` identical in all threads.
template <int D>
__device__ __forceinline__ void wg_goal_candidate(Lds &s, TreeDev &t, int idx, const double *v)
{
    double d[D];
#pragma unroll
    for (int k = 0; k < D; k++) d[k] = t.goal[k] - v[k];
    double h = dist_scan<D>(d);
    if (h <= t.step_len) {  // uniform
        bool col = wg_collision<D>(s, v, t.goal, t.clearance);
        if (threadIdx.x == 0) {
            int q = t.n_gc;
            t.gc_idx[q] = idx;
            t.gc_dist[q] = h;
            t.gc_col[q] = col ? 1 : 0;
            t.n_gc = q + 1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// one loop body (rrt_star_2d.py:37-55 / irrt_star_2d.py:54-73)
//   host_steer: node_in = node_new computed by the caller and nearest_in its nearest index
//   else      : node_in = node_rand
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void wg_iteration(Lds &s, TreeDev &t, const double *node_in, bool host_steer,
                                             int nearest_in, unsigned flags, nirrt_step_result *res)
{
    const int tid = threadIdx.x;
    const double clr = t.clearance;
    int n = t.n;
    int ni;
    double node_new[D], nearest[D];
    if (host_steer) {
        ni = nearest_in;
        load_vertex<D>(t, ni, nearest);
#pragma unroll
        for (int k = 0; k < D; k++) node_new[k] = node_in[k];
    } else {
        double bd;
        ni = wg_nearest<D>(s, t, n, node_in, bd);
        load_vertex<D>(t, ni, nearest);
        steer<D>(t, nearest, node_in, node_new);
    }
    if (res && tid == 0) {
        res->collided = 0; res->inserted = 0; res->nearest_idx = ni; res->new_idx = -1; res->n_near = 0;
        res->reparented = 0; res->n_rewired = 0; res->in_goal = 0; res->status = 0; res->reserved = 0;
        res->node_new[0] = node_new[0]; res->node_new[1] = node_new[1]; res->node_new[2] = D == 3 ? node_new[D - 1] : 0.;
    }
    bool collided = wg_collision<D>(s, nearest, node_new, clr);
    int new_idx = -1;
    if (!collided) {
        double diff[D];
#pragma unroll
        for (int k = 0; k < D; k++) diff[k] = node_new[k] - nearest[k];
        const bool dup = norm_1d<D>(diff) < 1e-8;
        bool inserted = false;
        if (dup) {
            new_idx = ni;
#pragma unroll
            for (int k = 0; k < D; k++) node_new[k] = nearest[k];
        } else if (n >= t.cap) {
            if (tid == 0) t.status = NIRRT_E_CAPACITY;
            new_idx = -2;  // tree full: treat as a no-op iteration
        } else {
            new_idx = n;
            if (tid == 0) {
#pragma unroll
                for (int k = 0; k < D; k++) t.c[k][new_idx] = node_new[k];
                t.parent[new_idx] = ni;
                t.n = n + 1;
            }
            n = n + 1;
            inserted = true;
            __syncthreads();
            wg_goal_candidate<D>(s, t, new_idx, node_new);
        }
        if (new_idx >= 0) {
            int k = wg_near<D>(s, t, n, node_new, new_idx);
            if (k < 0) {
                if (tid == 0) t.status = NIRRT_E_CAPACITY;
                k = 0;
            }
            int reparented = 0, n_rewired = 0;
            if (k > 0) {
                // parent-chain walks: lane j < k walks neighbour j, lane k walks `nearest`
                double c0 = 0., c1 = 0.;
                if (tid < k) {
                    c0 = walk_cost<D>(t, s.near_idx[tid], node_new, &c1);
                    s.near_c0[tid] = c0;
                    s.near_c1[tid] = c1;
                } else if (tid == k) {
                    c0 = walk_cost<D>(t, ni, node_new, &c1);
                    // curr_node_new_cost: rrt_star_2d.py:45 (same point) / :51
                    double curr = c0;
                    if (!dup) curr = c0 + hypot_py<D>(diff);
                    s.bc_d[0] = curr;
                    s.bc_d[1] = dup ? c0 : c1;  // cost(new) while parent[new] is unchanged
                }
                // (k may exceed NT only if NEAR_CAP > NT; NEAR_CAP == NT here)
                __syncthreads();
                // choose_parent (rrt_star_2d.py:80-90)
                double cand = __builtin_inf();
                int cj = 0x7fffffff;
                if (tid < k) { cand = s.near_c0[tid] + s.near_dist[tid]; cj = tid; }
                block_argmin(s, cand, cj);
                const double curr = s.bc_d[0];
                double new_cost = s.bc_d[1];
                if (cand < curr) {
                    reparented = 1;
                    if (tid == 0) t.parent[new_idx] = s.near_idx[cj];
                    new_cost = s.near_c1[cj];
                    __syncthreads();
                    if (dup) {
                        // node_new is an existing vertex that just moved in the tree: every neighbour
                        // below it changed cost -> re-walk before rewiring
                        if (tid < k) s.near_c0[tid] = walk_cost<D>(t, s.near_idx[tid], nullptr, nullptr);
                        __syncthreads();
                    }
                }
                // rewire (rrt_star_2d.py:92-99): sequential semantics.  All decisions up to and
                // including the first "true" are exact with the costs in hand; after a re-parenting the
                // remaining neighbours are re-walked (a rewired vertex may be their ancestor).
                int start = 0;
                while (start < k) {
                    int first = 0x7fffffff;
                    double dummy = __builtin_inf();
                    if (tid >= start && tid < k && s.near_c0[tid] > new_cost + s.near_dist[tid]) { first = tid; dummy = 0.; }
                    block_argmin(s, dummy, first);
                    if (first == 0x7fffffff) break;
                    if (tid == 0) t.parent[s.near_idx[first]] = new_idx;
                    n_rewired++;
                    start = first + 1;
                    __syncthreads();
                    if (start < k) {
                        if (tid >= start && tid < k) s.near_c0[tid] = walk_cost<D>(t, s.near_idx[tid], nullptr, nullptr);
                        __syncthreads();
                    }
                }
            }
            int in_goal = 0;
            if (flags & NIRRT_F_IRRT) {
                // InGoalRegion (rrt_base_2d.py:87-89): Line(node_new, goal) < step_len and collision-free
                double d[D];
#pragma unroll
                for (int kk = 0; kk < D; kk++) d[kk] = t.goal[kk] - node_new[kk];
                if (hypot_py<D>(d) < t.step_len) {
                    if (!wg_collision<D>(s, node_new, t.goal, clr)) {
                        in_goal = 1;
                        if (tid == 0) {
                            if (t.n_sol < t.cap_sol) { t.sol[t.n_sol] = new_idx; t.n_sol = t.n_sol + 1; }
                            else t.status = NIRRT_E_CAPACITY;
                        }
                        __syncthreads();
                    }
                }
            }
            if (res && tid == 0) {
                res->inserted = inserted ? 1 : 0; res->new_idx = new_idx; res->n_near = k;
                res->reparented = reparented; res->n_rewired = n_rewired; res->in_goal = in_goal;
                res->node_new[0] = node_new[0]; res->node_new[1] = node_new[1];
                res->node_new[2] = D == 3 ? node_new[D - 1] : 0.;
            }
        }
    } else if (res && tid == 0) {
        res->collided = 1;
    }
    __syncthreads();
    if (res) {
        double cb = __builtin_inf();
        int xb = -1;
        if (flags & NIRRT_F_IRRT) wg_best_solution<D>(s, t, cb, xb);
        else if (flags & NIRRT_F_GOAL_SCAN) wg_goal_parent<D>(s, t, xb, cb);
        if (tid == 0) {
            res->c_best = cb; res->x_best = xb; res->n_solutions = t.n_sol; res->n = t.n; res->status = t.status;
        }
    }
    __syncthreads();
}
