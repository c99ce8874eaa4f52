// nirrt_hip.hip — kernels + C ABI (include/nirrt_hip.h) of libnirrt_hip.so, gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
#include "nirrt_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <tuple>
#include <vector>

// ------------------------------------------------------------------------------------------------
// launch-argument structs shared by both kernel variants
// ------------------------------------------------------------------------------------------------
struct RunDev {
    unsigned flags;
    int pad;
    long long iters;
    const double *samples;  // (n_trees, iters, D) or nullptr
    double *cost_trace;     // (n_trees, iters) or nullptr
    long long *iters_done;  // (n_trees,)
};

// MT19937: the 624 state words after one more twist (genrand's refill), one wave, all 64 lanes.  Lane l holds words l, l + 64,
// ... of the block (ten registers, the last row 48 words wide).  Word i needs old[i], old[i + 1] and word (i + 397) mod 624,
// which is an OLD word for i < 227 and a NEW one - word i - 227, three to four rows back - from there on: rows are produced in
// order, the neighbours come through ds_bpermute / v_readlane, nothing goes through memory until the block is complete.
static __device__ __noinline__ void mt_next_block(const GAS unsigned *old_k, GAS unsigned *new_k)
{
    const int l = lane_id();
    unsigned o[10], nw[10];
#pragma unroll
    for (int r = 0; r < 10; r++) o[r] = (64 * r + l) < MT_N ? old_k[64 * r + l] : 0u;
    auto lane_of = [](unsigned v, int src) -> unsigned { return (unsigned)__builtin_amdgcn_ds_bpermute((src & 63) << 2, (int)v); };
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // old[i + 1]: the next lane of this row, lane 0 of the next row for l == 63; word 623 pairs with the NEW word 0
        unsigned o1 = lane_of(o[r], l + 1);
        if (r < 9) { const unsigned nx = (unsigned)__builtin_amdgcn_readlane((int)o[r + 1], 0); o1 = l == 63 ? nx : o1; }
        else { const unsigned n0 = (unsigned)__builtin_amdgcn_readlane((int)nw[0], 0); o1 = l == 47 ? n0 : o1; }
        const unsigned y = (o[r] & 0x80000000u) | (o1 & 0x7fffffffu);
        unsigned m;
        if (r <= 2) {          // i + 397 = 64 (r + 6) + l + 13: old rows r + 6 / r + 7
            const unsigned a = lane_of(o[r + 6], l + 13), b = lane_of(o[r + 7], l + 13);
            m = l + 13 < 64 ? a : b;
        } else if (r == 3) {   // i < 227 <=> l < 35: old word i + 397 (row 9, lane l + 13); else new word i - 227 (row 0, lane l - 35)
            const unsigned a = lane_of(o[9], l + 13), b = lane_of(nw[0], l - 35);
            m = l < 35 ? a : b;
        } else {               // new word i - 227 = 64 (r - 4) + l + 29: rows r - 4 / r - 3
            const unsigned a = lane_of(nw[r - 4], l + 29), b = lane_of(nw[r - 3], l - 35);
            m = l < 35 ? a : b;
        }
        nw[r] = m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
#pragma unroll
    for (int r = 0; r < 10; r++)
        if (64 * r + l < MT_N) new_k[64 * r + l] = nw[r];
    __threadfence_block();   // the window refill that follows reads the block back (other lanes' words)
}

__device__ __forceinline__ unsigned mt_temper(unsigned y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// Raw 32-bit outputs of a generator, consumed in order.  Executed by ALL lanes of wave 0 with identical values:
// lane i keeps word base+i of a 64-word window in a register, a word is fetched with v_readlane, and the window is
// refilled with one coalesced load when the position leaves it (so a draw costs no memory round trip most of the time).
// Word mode: the outputs were produced by the caller (w[0 .. n)).  Generator mode: the tree's own MT19937 stream (MtGen) -
// w points at the stream's two key blocks, a refill tempers the state words of the block the position is in and produces
// the next block when the position enters it; n is the limit of the current draw (see draw_call).
struct WordStream {
    const GAS unsigned *w;
    long long n, pos;
    long long base;   // first word of the window, -1: nothing loaded
    unsigned reg;     // this lane's word of the window
    int wn;           // words in the window
    int gen;          // generator mode: highest block produced so far
    int genmode;
    __device__ __forceinline__ bool has(long long k) const { return pos + k <= n; }
    __device__ __forceinline__ void refill()   // (inline: an out-of-line member would force the stream's state out of registers)
    {
        base = pos;
        const int lane = lane_id();
        if (!genmode) {
            const long long i = base + (long long)lane;
            wn = 64;
            reg = i < n ? w[i] : 0u;
            return;
        }
        const long long b = pos / MT_N;               // wave-uniform; b is gen - 1, gen or gen + 1
        const int j = (int)(pos - b * MT_N);
        if (b > gen) {
            mt_next_block(w + (gen & 1) * MT_N, const_cast<GAS unsigned *>(w) + ((gen + 1) & 1) * MT_N);
            gen++;
        }
        wn = MT_N - j < 64 ? MT_N - j : 64;
        reg = mt_temper(lane < wn ? w[(int)(b & 1) * MT_N + j + lane] : 0u);
    }
    __device__ __forceinline__ unsigned next_word()
    {
        long long rel = pos - base;
        if (base < 0 || rel < 0 || rel >= wn) {   // wave-uniform
            refill();
            rel = 0;
        }
        const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)reg, __builtin_amdgcn_readfirstlane((int)rel));
        pos++;
        return v;
    }
    __device__ __forceinline__ double next_double()
    {
        unsigned a = next_word() >> 5, b = next_word() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // generator mode: the block that holds output pos - 1 (what get_state() shows) must stay in the two-block ring while a
    // draw that may be undone runs: such a draw may not enter block anchor + 2
    __device__ __forceinline__ long long undo_limit() const { return ((pos > 0 ? (pos - 1) / MT_N : 0) + 2) * MT_N; }
};

// (RunSampleDev / PoolDev: nirrt_device.hpp - the run loop keeps its launch arguments in LDS)

// Three instantiations of every kernel; nirrt_run picks by batch size so that the CU's 16 wave slots are busy:
//   slim   64 threads (one wave per tree, 12 trees per CU = every wave slot at 168 VGPRs): batches of more than 2048 trees;
//   narrow 128 threads (8 trees per CU): with the grid index an iteration is a chain of short dependent phases, so trees
//          in flight per CU is what counts (measured on 2048 problems: 1.2x IRRT*, 1.7x RRT* over 256-thread workgroups;
//          256 was best while the O(n) scans dominated);
//   wide   256 threads: one or a few trees - measured for one 50k-iteration problem: IRRT* 2.10 s (128 threads 2.66 s,
//          1024 threads 2.24 s), RRT* 1.4x faster than with 1024 threads, which had been best for the scans.

#ifndef NIRRT_BODY_ATTR
#define NIRRT_BODY_ATTR __noinline__   // loop body of the persistent kernels: a real call (see iteration_call)
#endif
#ifndef NIRRT_DRAW_ATTR
#define NIRRT_DRAW_ATTR __noinline__
#endif
#ifndef NIRRT_NT_NARROW
#define NIRRT_NT_NARROW 128
#endif
#define NT NIRRT_NT_NARROW
namespace narrow {
#include "nirrt_kernels.inc"
}
#undef NT
#ifndef NIRRT_NT_WIDE
#define NIRRT_NT_WIDE 256
#endif
#define NT NIRRT_NT_WIDE
namespace wide {
#include "nirrt_kernels.inc"
}
#undef NT
#define NT 64
namespace slim {
#include "nirrt_kernels.inc"
}
#undef NT
#define NT_NARROW NIRRT_NT_NARROW
#define NT_WIDE NIRRT_NT_WIDE
#define NT_SLIM 64
static_assert(NT_WIDE / 64 <= LDS_NW_MAX && NT_NARROW / 64 <= LDS_NW_MAX, "LdsData reduction slots: raise LDS_NW_MAX");
static_assert(offsetof(TreeHotH, pc) > offsetof(TreeHotH, CL_C) && offsetof(TreeHotH, g_rec) > offsetof(TreeHotH, c_update),
              "nirrt_set_informed / nirrt_set_cloud patch contiguous field ranges of the descriptor");
// 12 one-wave trees per CU (3 waves per SIMD at 168 VGPRs): 160 KB / 12 in allocation granules of 1280 bytes = 12800 bytes each
static_assert(sizeof(LdsData) + 256 <= 12800, "LdsData (+ the 256 B of LDS the module's other kernels declare: one module-wide allocation) must fit 12 times into a CU's 160 KB of LDS");
// nirrt_run: batches larger than the 2048 workgroup slots of the 128-thread kernels run one wave per tree (16 trees per
// CU with 10 KB of LDS): measured on 4096 problems 12.8 vs 10.9 M it/s (IRRT*), 39.0 vs 26.3 M it/s (RRT*); at 2048
// problems the 128-thread kernels win (IRRT* 10.9 vs 8.6) or tie (RRT*).  NIRRT_SLIM_MIN_TREES overrides the threshold.
// The knobs are read on every call (tests switch them inside one process).
static int env_int(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    return (e && *e) ? std::atoi(e) : dflt;
}
static int slim_min_trees() { return env_int("NIRRT_SLIM_MIN_TREES", 2049); }
// nirrt_run: batches up to this many trees use the 256-thread kernels (4 trees per CU fill its 16 wave slots); measured
// on 1024 problems: IRRT* (hundreds of Near members per iteration) 8.5 vs 7.3 M it/s, RRT* 15.2 vs 16.9 M it/s.
// NIRRT_WIDE_MAX_TREES overrides both (0 = always the 128-thread kernels).
static int wide_max_trees(unsigned flags)
{
    const int env = env_int("NIRRT_WIDE_MAX_TREES", -1);
    if (env >= 0) return env;
    return (flags & NIRRT_F_IRRT) ? 1024 : 256;
}
// NIRRT_FORCE_VARIANT=slim|narrow|wide pins EVERY kernel launch of the library (primitives, step, persistent loops) to
// one instantiation: the parity suite runs whole under each of them (tests/test_hip_variants.py).
enum Variant { V_AUTO = 0, V_SLIM, V_NARROW, V_WIDE };
static Variant forced_variant()
{
    const char *e = std::getenv("NIRRT_FORCE_VARIANT");
    if (!e || !*e) return V_AUTO;
    if (!std::strcmp(e, "slim")) return V_SLIM;
    if (!std::strcmp(e, "narrow")) return V_NARROW;
    if (!std::strcmp(e, "wide")) return V_WIDE;
    return V_AUTO;
}
#define WIDE_MIN_VERTICES 16000   // ... once the trees are (or will grow) this big

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

#include <atomic>
struct ScratchBlock { void *base; std::atomic<int> refs; };
#define PC_OWN_POINTS 4096   // guidance-cloud points that fit the per-tree arena (pc_n_points is 2048 in the reference's configs)
struct nirrt_tree {
    nirrt_config cfg;
    int dim;
    int cap;
    int device;
    hipStream_t stream;
    TreeDev host;    // host mirror of the descriptor (pointers are device pointers)
    TreeDev *dev;    // descriptor in HBM
    TreeDev **self_dev;   // one-element device array holding `dev` (kernels take arrays of descriptors)
    void *arena;     // ONE device range holding every per-tree array, the descriptor and the Near-radius table
    size_t arena_bytes;
    bool arena_pooled;   // carved out of a pool chunk (see ArenaPool) / an allocation of its own
    double *near_r;  // device table (inside the arena)
    Scratch *scratch;      // pinned host memory (a slot of sblk)
    Scratch *scratch_dev;  // device alias of the same memory
    struct ScratchBlock *sblk;   // the pinned block the slot lives in, shared by the trees of one nirrt_create_batch call
    double *pc_dev;        // guidance cloud (nirrt_set_cloud): pc_own (inside the arena, PC_OWN_POINTS points) or an allocation of its own
    double *pc_own;
    MtGen *mt;             // the tree's generators (inside the arena)
    long long last_n;      // num_vertices as of the last call that reported it (kernel-variant choice only)
};

extern "C" const char *nirrt_last_error(void) { return g_err.c_str(); }
extern "C" int nirrt_abi_version(void) { return NIRRT_ABI_VERSION; }

extern "C" int nirrt_device_count(int *count)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
    if (count) *count = c;
    return NIRRT_OK;
}

#define LAUNCH_V(var, dimv, KERNEL, grid, st, ...)                                             \
    do {                                                                                       \
        switch (var) {                                                                         \
        case V_SLIM:                                                                           \
            if ((dimv) == 2) hipLaunchKernelGGL(slim::KERNEL<2>, dim3(grid), dim3(NT_SLIM), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL(slim::KERNEL<3>, dim3(grid), dim3(NT_SLIM), 0, st, __VA_ARGS__); \
            break;                                                                             \
        case V_NARROW:                                                                         \
            if ((dimv) == 2) hipLaunchKernelGGL(narrow::KERNEL<2>, dim3(grid), dim3(NT_NARROW), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL(narrow::KERNEL<3>, dim3(grid), dim3(NT_NARROW), 0, st, __VA_ARGS__); \
            break;                                                                             \
        default:                                                                               \
            if ((dimv) == 2) hipLaunchKernelGGL(wide::KERNEL<2>, dim3(grid), dim3(NT_WIDE), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL(wide::KERNEL<3>, dim3(grid), dim3(NT_WIDE), 0, st, __VA_ARGS__); \
            break;                                                                             \
        }                                                                                      \
    } while (0)

// grid-stride / one-off kernels: 256 threads unless a variant is forced
#define DISPATCH_DIM(t, KERNEL, grid, ...)                                                     \
    do {                                                                                       \
        const Variant fv_ = forced_variant();                                                  \
        LAUNCH_V(fv_ == V_AUTO ? V_WIDE : fv_, (t)->dim, KERNEL, grid, (t)->stream, __VA_ARGS__); \
    } while (0)

// single-tree kernels: by tree size
#define DISPATCH_BY_SIZE(t, KERNEL, ...)                                                        \
    do {                                                                                       \
        const Variant fv_ = forced_variant();                                                  \
        const Variant v_ = fv_ != V_AUTO ? fv_ : ((t)->last_n >= WIDE_MIN_VERTICES ? V_WIDE : V_NARROW); \
        LAUNCH_V(v_, (t)->dim, KERNEL, 1, (t)->stream, __VA_ARGS__);                           \
    } while (0)

// persistent loops: by batch size and algorithm (see the notes above the three namespaces)
static Variant run_variant(int n_trees, unsigned flags, long long n_hi, long long iters)
{
    const Variant fv = forced_variant();
    if (fv != V_AUTO) return fv;
    if (n_trees >= slim_min_trees()) return V_SLIM;
    if (n_trees <= wide_max_trees(flags) && n_hi + iters >= WIDE_MIN_VERTICES) return V_WIDE;
    return V_NARROW;
}

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

// Tree arenas come out of large device chunks (NIRRT_POOL_CHUNK_MB, default 4096; 0 = one hipMalloc per tree): thousands of
// separately mapped 12 MB allocations cost one address-translation entry per 2 MB each, while a few multi-GB chunks are
// mapped with large fragments - the loop is a latency-bound chase across ~100 GB of trees, and translation misses are part of
// every round trip.  Bump allocation inside a chunk, free lists by size per device (an idle block serves a request of up to
// its own size, down to half of it).  A chunk whose last tree is destroyed starts over empty, and all but one idle chunk per
// device go back to the driver at once (nirrt_pool_trim returns that one too): a long-lived process that plans batches of
// varying iter_max neither grows without bound nor starves other users of the device (torch) of memory.
#include <map>
#include <mutex>
namespace {
struct ArenaPool {
    std::mutex mu;
    struct Chunk { int device; char *base; size_t size, used; long live; };
    std::vector<Chunk> chunks;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_list;   // (device, bytes) -> arenas handed back
    static size_t chunk_bytes()
    {
        const char *e = std::getenv("NIRRT_POOL_CHUNK_MB");
        const long long mb = (e && *e) ? std::atoll(e) : 4096;
        return mb <= 0 ? 0 : (size_t)mb << 20;
    }
    Chunk *owner(const void *p)
    {
        for (Chunk &c : chunks)
            if ((const char *)p >= c.base && (const char *)p < c.base + c.size) return &c;
        return nullptr;
    }
    // returns nullptr when pooling is off or the request is small / larger than a chunk (the caller then uses hipMalloc);
    // *got = size of the block handed out (>= bytes: an idle block of another tree size is reused if it is large enough)
    void *take(int device, size_t bytes, size_t *got)
    {
        const size_t cb = chunk_bytes();
        *got = bytes;
        if (cb == 0 || bytes < ((size_t)1 << 20) || bytes > cb) return nullptr;
        std::lock_guard<std::mutex> g(mu);
        // the smallest idle block of this device that fits
        auto it = free_list.lower_bound({device, bytes});
        while (it != free_list.end() && it->first.first == device && it->second.empty()) ++it;
        if (it != free_list.end() && it->first.first == device && it->first.second <= 2 * bytes) {   // (not a block twice the size: it would pin the room of two trees)
            void *p = it->second.back();
            it->second.pop_back();
            *got = it->first.second;
            owner(p)->live++;
            return p;
        }
        const size_t A = (size_t)2 << 20;   // arenas start on 2 MB boundaries
        const size_t need = (bytes + A - 1) / A * A;
        Chunk *c = nullptr;
        for (Chunk &k : chunks)
            if (k.device == device && k.used + need <= k.size) c = &k;
        if (!c) {
            void *b = nullptr;
            if (hipMalloc(&b, cb) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            chunks.push_back(Chunk{device, (char *)b, cb, 0, 0});
            c = &chunks.back();
        }
        void *p = c->base + c->used;
        c->used += need;
        c->live++;
        return p;
    }
    void give(int device, size_t bytes, void *p)
    {
        std::lock_guard<std::mutex> g(mu);
        Chunk *c = owner(p);
        c->live--;
        if (c->live > 0) { free_list[{device, bytes}].push_back(p); return; }
        // the chunk's last tree is gone: it starts over empty (its idle blocks are forgotten) - a process that creates and destroys
        // trees of many sizes does not grow without bound, and the chunk can serve any size again
        for (auto &kv : free_list) {
            auto &v = kv.second;
            v.erase(std::remove_if(v.begin(), v.end(), [&](void *q) { return (char *)q >= c->base && (char *)q < c->base + c->size; }), v.end());
        }
        c->used = 0;
        // more than one idle chunk per device is more than the next batch needs to start: the others go back to the driver
        int idle = 0;
        for (const Chunk &k : chunks) idle += (k.device == device && k.live == 0) ? 1 : 0;
        if (idle > 1) {
            const Chunk dead = *c;
            chunks.erase(chunks.begin() + (c - chunks.data()));
            (void)hipSetDevice(dead.device);
            (void)hipFree(dead.base);
        }
    }
    // chunks none of whose arenas is in use go back to the driver
    void trim()
    {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < chunks.size();) {
            Chunk c = chunks[i];
            if (c.live > 0) { i++; continue; }
            for (auto &kv : free_list) {
                auto &v = kv.second;
                v.erase(std::remove_if(v.begin(), v.end(), [&](void *p) { return (char *)p >= c.base && (char *)p < c.base + c.size; }), v.end());
            }
            (void)hipSetDevice(c.device);
            (void)hipFree(c.base);
            chunks.erase(chunks.begin() + (long)i);
        }
    }
};
ArenaPool g_pool;

// Launch scratch (argument tables, counters, traces of nirrt_run): blocks are kept and handed out again, rounded up to powers of
// two.  hipFree waits for the whole device - with two batches launched from two host threads the thread that finished first sat
// in its cleanup until the other thread's persistent kernel had ended.
struct ScratchPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> idle;   // (device, bytes) -> block
    static size_t round_up(size_t b) { size_t r = 256; while (r < b) r <<= 1; return r; }
    hipError_t take(int device, size_t bytes, void **out, size_t *got)
    {
        if (bytes > ((size_t)64 << 20)) { *got = 0; return hipMalloc(out, bytes); }   // word slabs of host-fed runs: exact, not kept
        const size_t r = round_up(bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = idle.find({device, r});
            if (it != idle.end()) { *out = it->second; *got = r; idle.erase(it); return hipSuccess; }
        }
        *got = r;
        return hipMalloc(out, r);
    }
    void give(int device, size_t bytes, void *p)
    {
        if (bytes == 0) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> g(mu);
        idle.insert({{device, bytes}, p});
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto &kv : idle) { (void)hipSetDevice(kv.first.first); (void)hipFree(kv.second); }
        idle.clear();
    }
};
ScratchPool g_scratch;
}

/* MT19937 outputs in bulk (host): numpy's legacy RandomState and CPython's random.Random are this generator, and the loops
 * consume their raw 32-bit outputs.  key: the 624 state words, *pos: position inside the block (624 = used up), both updated in
 * place the way the generators themselves would be after n outputs (pass copies to peek). */
extern "C" int nirrt_mt19937_fill(uint32_t *key, int32_t *pos, int64_t n, uint32_t *out)
{
    if (!key || !pos || n < 0 || (n > 0 && !out) || *pos < 0 || *pos > 624) return NIRRT_E_ARG;
    int p = *pos;
    for (int64_t i = 0; i < n;) {
        if (p >= 624) {
            // the next block: three stretches whose inputs are all in place (the reference implementation's loop, unrolled by range)
            int k = 0;
            for (; k < 624 - 397; k++) {
                const uint32_t y = (key[k] & 0x80000000u) | (key[k + 1] & 0x7fffffffu);
                key[k] = key[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; k < 623; k++) {
                const uint32_t y = (key[k] & 0x80000000u) | (key[k + 1] & 0x7fffffffu);
                key[k] = key[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            const uint32_t y = (key[623] & 0x80000000u) | (key[0] & 0x7fffffffu);
            key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            p = 0;
        }
        const int64_t take = std::min<int64_t>(624 - p, n - i);
        for (int64_t j = 0; j < take; j++) {
            uint32_t y = key[p + j];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[i + j] = y;
        }
        p += (int)take;
        i += take;
    }
    *pos = p;
    return NIRRT_OK;
}

// ---- the trees' own generators ------------------------------------------------------------------------------------
// one wave per tree; keys: (n, 624) words per stream, pos: (n,) - a stream whose key pointer is null is left alone
__global__ __launch_bounds__(64) void k_set_generators(TreeDev *const *trees, const unsigned *np_key, const int *np_pos,
                                                       const unsigned *py_key, const int *py_pos)
{
    MtGen *m = trees[blockIdx.x]->mt;
    const int l = threadIdx.x;
    for (int g = 0; g < 2; g++) {
        const unsigned *key = g == 0 ? np_key : py_key;
        const int *pos = g == 0 ? np_pos : py_pos;
        if (!key) continue;
        for (int i = l; i < MT_N; i += 64) m->key[g][0][i] = key[(size_t)blockIdx.x * MT_N + i];
        if (l == 0) { m->pos[g] = pos[blockIdx.x]; m->gen[g] = 0; }
    }
}

// the state as the generator's own get_state() / getstate() would show it: the block that holds the last consumed output and
// the position inside it (1..624; 0 only if nothing was consumed from a state handed over at position 0)
__global__ __launch_bounds__(64) void k_get_generators(TreeDev *const *trees, unsigned *np_key, int *np_pos, unsigned *py_key, int *py_pos)
{
    const MtGen *m = trees[blockIdx.x]->mt;
    const int l = threadIdx.x;
    for (int g = 0; g < 2; g++) {
        unsigned *key = g == 0 ? np_key : py_key;
        int *pos = g == 0 ? np_pos : py_pos;
        if (!key) continue;
        const long long p = m->pos[g];
        const long long blk = p > 0 ? (p - 1) / MT_N : 0;
        for (int i = l; i < MT_N; i += 64) key[(size_t)blockIdx.x * MT_N + i] = m->key[g][blk & 1][i];
        if (l == 0) pos[blockIdx.x] = (int)(p - blk * MT_N);
    }
}

// the next n outputs of stream `which` of every tree, written to out + tree * stride; the streams move on by n
__global__ __launch_bounds__(64) void k_generator_words(TreeDev *const *trees, int which, long long n, unsigned *out, long long stride)
{
    MtGen *m = trees[blockIdx.x]->mt;
    const int l = threadIdx.x;
    WordStream ws = {(const GAS unsigned *)&m->key[which][0][0], 0, m->pos[which], -1, 0u, 0, m->gen[which], 1};
    unsigned *o = out + (size_t)blockIdx.x * (size_t)stride;
    long long done = 0;
    while (done < n) {   // a window (the rest of a block, at most 64 outputs) per trip
        ws.refill();
        const long long take = ws.wn < n - done ? ws.wn : n - done;
        if (l < take) o[done + l] = ws.reg;
        done += take;
        ws.pos += take;
    }
    if (l == 0) { m->pos[which] = ws.pos; m->gen[which] = ws.gen; }
}

namespace {
// trees of one device -> device array of their descriptors (launch scratch from the pool)
struct TreeList {
    TreeDev **d = nullptr;
    size_t got = 0;
    int device = 0;
    // the host copy of the table lives as long as the list: tree_list() hands it to an ASYNCHRONOUS copy, and the stream is
    // synchronized by the list's user, after tree_list() has returned (round 6: it was a local of tree_list() - a pageable source
    // that is freed while the copy is pending is whatever the heap holds by then; the candidate cause of the round's two rare
    // "Memory access fault" aborts, both in guided runs, whose every refresh comes through here)
    std::vector<TreeDev *> host;
    ~TreeList() { if (d) g_scratch.give(device, got, d); }
};
}
static int tree_list(nirrt_tree *const *trees, int32_t n_trees, const char *who, TreeList &tl)
{
    if (!trees || n_trees <= 0) { g_err = std::string(who) + ": no trees"; return NIRRT_E_ARG; }
    nirrt_tree *t0 = trees[0];
    for (int i = 0; i < n_trees; i++)
        if (!trees[i] || trees[i]->device != t0->device) { g_err = std::string(who) + ": all trees must live on one device"; return NIRRT_E_ARG; }
    HIPCHK(hipSetDevice(t0->device));
    for (int i = 0; i < n_trees; i++) HIPCHK(hipStreamSynchronize(trees[i]->stream));
    tl.host.resize((size_t)n_trees);
    for (int i = 0; i < n_trees; i++) tl.host[(size_t)i] = trees[i]->dev;
    tl.device = t0->device;
    HIPCHK(g_scratch.take(t0->device, sizeof(TreeDev *) * (size_t)n_trees, (void **)&tl.d, &tl.got));
    HIPCHK(hipMemcpyAsync(tl.d, tl.host.data(), sizeof(TreeDev *) * (size_t)n_trees, hipMemcpyHostToDevice, t0->stream));
    return NIRRT_OK;
}

/* np.random.set_state / random.setstate for the trees' own generators: (key, pos) as get_state() / getstate() expose them */
extern "C" int nirrt_set_generators(nirrt_tree *const *trees, int32_t n_trees, const uint32_t *np_key, const int32_t *np_pos,
                                    const uint32_t *py_key, const int32_t *py_pos)
{
    TreeList tl;
    int rc = tree_list(trees, n_trees, "nirrt_set_generators", tl);
    if (rc) return rc;
    if ((np_key && !np_pos) || (py_key && !py_pos)) { g_err = "nirrt_set_generators: a key table needs its positions"; return NIRRT_E_ARG; }
    for (int i = 0; i < n_trees; i++)
        if ((np_key && (np_pos[i] < 0 || np_pos[i] > MT_N)) || (py_key && (py_pos[i] < 0 || py_pos[i] > MT_N))) {
            g_err = "nirrt_set_generators: position outside 0..624";
            return NIRRT_E_ARG;
        }
    hipStream_t st = trees[0]->stream;
    const size_t kb = sizeof(uint32_t) * MT_N * (size_t)n_trees, pb = sizeof(int32_t) * (size_t)n_trees;
    char *slab = nullptr;
    size_t got = 0;
    HIPCHK(g_scratch.take(tl.device, 2 * (kb + pb) + 1024, (void **)&slab, &got));
    unsigned *d_k[2] = {(unsigned *)slab, (unsigned *)(slab + kb)};
    int *d_p[2] = {(int *)(slab + 2 * kb), (int *)(slab + 2 * kb + ((pb + 255) & ~(size_t)255))};
    hipError_t e = hipSuccess;
    if (np_key) { e = hipMemcpyAsync(d_k[0], np_key, kb, hipMemcpyHostToDevice, st); if (e == hipSuccess) e = hipMemcpyAsync(d_p[0], np_pos, pb, hipMemcpyHostToDevice, st); }
    if (e == hipSuccess && py_key) { e = hipMemcpyAsync(d_k[1], py_key, kb, hipMemcpyHostToDevice, st); if (e == hipSuccess) e = hipMemcpyAsync(d_p[1], py_pos, pb, hipMemcpyHostToDevice, st); }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_set_generators, dim3(n_trees), dim3(64), 0, st, (TreeDev *const *)tl.d, np_key ? d_k[0] : nullptr, d_p[0],
                           py_key ? d_k[1] : nullptr, d_p[1]);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    g_scratch.give(tl.device, got, slab);
    if (e != hipSuccess) { g_err = std::string("nirrt_set_generators: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    return NIRRT_OK;
}

/* np.random.get_state / random.getstate of the trees' own generators */
extern "C" int nirrt_get_generators(nirrt_tree *const *trees, int32_t n_trees, uint32_t *np_key, int32_t *np_pos, uint32_t *py_key,
                                    int32_t *py_pos)
{
    TreeList tl;
    int rc = tree_list(trees, n_trees, "nirrt_get_generators", tl);
    if (rc) return rc;
    if ((np_key && !np_pos) || (py_key && !py_pos)) { g_err = "nirrt_get_generators: a key table needs its positions"; return NIRRT_E_ARG; }
    hipStream_t st = trees[0]->stream;
    const size_t kb = sizeof(uint32_t) * MT_N * (size_t)n_trees, pb = sizeof(int32_t) * (size_t)n_trees;
    char *slab = nullptr;
    size_t got = 0;
    HIPCHK(g_scratch.take(tl.device, 2 * (kb + pb) + 1024, (void **)&slab, &got));
    unsigned *d_k[2] = {(unsigned *)slab, (unsigned *)(slab + kb)};
    int *d_p[2] = {(int *)(slab + 2 * kb), (int *)(slab + 2 * kb + ((pb + 255) & ~(size_t)255))};
    hipLaunchKernelGGL(k_get_generators, dim3(n_trees), dim3(64), 0, st, (TreeDev *const *)tl.d, np_key ? d_k[0] : nullptr, d_p[0],
                       py_key ? d_k[1] : nullptr, d_p[1]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && np_key) { e = hipMemcpyAsync(np_key, d_k[0], kb, hipMemcpyDeviceToHost, st); if (e == hipSuccess) e = hipMemcpyAsync(np_pos, d_p[0], pb, hipMemcpyDeviceToHost, st); }
    if (e == hipSuccess && py_key) { e = hipMemcpyAsync(py_key, d_k[1], kb, hipMemcpyDeviceToHost, st); if (e == hipSuccess) e = hipMemcpyAsync(py_pos, d_p[1], pb, hipMemcpyDeviceToHost, st); }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    g_scratch.give(tl.device, got, slab);
    if (e != hipSuccess) { g_err = std::string("nirrt_get_generators: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    return NIRRT_OK;
}

/* the next n_words raw outputs of every tree's numpy (which = 0) or python (which = 1) generator, produced on the device */
extern "C" int nirrt_generator_words(nirrt_tree *const *trees, int32_t n_trees, int32_t which, int64_t n_words, uint32_t *out,
                                     int64_t stride, int32_t out_on_device)
{
    if ((which != 0 && which != 1) || n_words < 0 || stride < n_words || (n_words > 0 && !out)) { g_err = "nirrt_generator_words: bad arguments"; return NIRRT_E_ARG; }
    TreeList tl;
    int rc = tree_list(trees, n_trees, "nirrt_generator_words", tl);
    if (rc || n_words == 0) return rc;
    hipStream_t st = trees[0]->stream;
    unsigned *d_out = out;
    size_t got = 0;
    const size_t bytes = sizeof(uint32_t) * (size_t)stride * (size_t)n_trees;
    if (!out_on_device) HIPCHK(g_scratch.take(tl.device, bytes, (void **)&d_out, &got));
    hipLaunchKernelGGL(k_generator_words, dim3(n_trees), dim3(64), 0, st, (TreeDev *const *)tl.d, (int)which, (long long)n_words, d_out,
                       (long long)stride);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !out_on_device) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (!out_on_device) g_scratch.give(tl.device, got, d_out);
    if (e != hipSuccess) { g_err = std::string("nirrt_generator_words: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    return NIRRT_OK;
}

extern "C" int nirrt_pool_trim(void)
{
    g_pool.trim();
    g_scratch.trim();
    return NIRRT_OK;
}

extern "C" int nirrt_destroy(nirrt_tree *t)
{
    if (!t) return NIRRT_OK;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->pc_dev && t->pc_dev != t->pc_own) (void)hipFree(t->pc_dev);
    if (t->arena && t->arena_pooled) g_pool.give(t->device, t->arena_bytes, t->arena);
    else if (t->arena) (void)hipFree(t->arena);
    if (t->sblk && --t->sblk->refs == 0) {
        if (t->sblk->base) (void)hipHostFree(t->sblk->base);
        delete t->sblk;
    }
    delete t;      // (the stream belongs to the device's pool)
    return NIRRT_OK;
}

static void host_mirror_reset(nirrt_tree *t)
{
    t->host.n = 1;
    t->last_n = 1;
    t->host.n_sol = 0;
    t->host.n_gc = 0;
    t->host.status = 0;
    for (int i = 0; i < NSTAT; i++) t->host.stat[i] = 0;
}

extern "C" int nirrt_reset(nirrt_tree *t)
{
    if (!t) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    host_mirror_reset(t);
    DISPATCH_DIM(t, k_init, 1, (TreeDev *const *)t->self_dev, 1);   // back to the single start vertex, on the device
    return sync_check(t);
}

/* all trees of a batch back to their single start vertex in ONE launch (one workgroup per tree) */
extern "C" int nirrt_reset_batch(nirrt_tree *const *trees, int32_t n_trees)
{
    if (!trees || n_trees <= 0) return NIRRT_E_ARG;
    nirrt_tree *t0 = trees[0];
    for (int i = 0; i < n_trees; i++)
        if (!trees[i] || trees[i]->device != t0->device || trees[i]->dim != t0->dim) { g_err = "nirrt_reset_batch: all trees must share device and dim"; return NIRRT_E_ARG; }
    HIPCHK(hipSetDevice(t0->device));
    for (int i = 1; i < n_trees; i++) HIPCHK(hipStreamSynchronize(trees[i]->stream));
    std::vector<TreeDev *> ptrs((size_t)n_trees);
    for (int i = 0; i < n_trees; i++) { ptrs[(size_t)i] = trees[i]->dev; host_mirror_reset(trees[i]); }
    TreeDev **d_ptrs = nullptr;
    HIPCHK(hipMalloc(&d_ptrs, sizeof(TreeDev *) * (size_t)n_trees));
    HIPCHK(hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(TreeDev *) * (size_t)n_trees, hipMemcpyHostToDevice, t0->stream));
    DISPATCH_DIM(t0, k_init, n_trees, (TreeDev *const *)d_ptrs, 1);
    int rc = sync_check(t0);
    (void)hipFree(d_ptrs);
    return rc;
}

static hipStream_t tree_stream(int device)
{
    static std::mutex mu;
    static std::map<int, std::vector<hipStream_t>> pools;
    static std::map<int, size_t> next;
    std::lock_guard<std::mutex> g(mu);
    std::vector<hipStream_t> &p = pools[device];
    if (p.empty()) {   // all of a device's streams at its first tree: one-time runtime allocations, like the code objects
        for (int i = 0; i < 32; i++) {
            hipStream_t st = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            p.push_back(st);
        }
        if (p.empty()) return nullptr;
    }
    return p[next[device]++ % p.size()];
}

// the Near radius r(n) = min(gamma * f(n), step_len) with f(n) = sqrt(log n / n) (2D) / (log n / n)^(1/3) (3D), host libm
// (rrt_star_2d.py:133, rrt_star_3d.py:134): f depends on the dimension only - tabulated once per capacity, not per tree
static const std::vector<double> &near_factor(int D, int cap)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::vector<double>> tabs;
    std::lock_guard<std::mutex> g(mu);
    std::vector<double> &f = tabs[{D, cap}];
    if (f.empty()) {
        f.assign((size_t)cap + 1, 0.0);
        for (int n = 1; n <= cap; n++) {
            const double x = std::log((double)n) / (double)n;
            f[(size_t)n] = D == 2 ? std::sqrt(x) : std::pow(x, 1 / 3.);
        }
    }
    return f;
}

// NIRRT_CREATE_PROFILE=1: host seconds per stage of nirrt_create, printed every 1024 trees (set-up of a large batch is outside the
// benchmark's timed step but it is what a user waits for first)
#include <chrono>
static double g_create_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static long g_create_n = 0;
struct CreateTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    CreateTimer() : on(std::getenv("NIRRT_CREATE_PROFILE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(int slot)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        g_create_s[slot] += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    }
    void done()
    {
        if (!on) return;
        if (++g_create_n % 1024 == 0)
            fprintf(stderr, "nirrt_create x %ld: stream %.3f arena %.3f memset %.3f pinned %.3f near_r %.3f desc+reset %.3f s\n", g_create_n,
                    g_create_s[0], g_create_s[1], g_create_s[2], g_create_s[3], g_create_s[4], g_create_s[5]);
    }
};

// ---- creation: host-side preparation per tree, ONE device pass per batch ---------------------------------------------------------
// nirrt_create is the batch of one.  Per tree nothing waits for the device: the arena is carved out of a pool chunk, the descriptor
// is filled in on the host and copied asynchronously; one k_init launch (mode 2) over the batch clears what must be clear in a
// (possibly recycled) arena - the rewire stamps and the generators -, tabulates the Near radii and builds the one-vertex trees.
// Round 5 did per tree: four hipMemsets (8 MB of vertex / tree records that the loop overwrites before it reads them), a pinned
// allocation, a 400 KB synchronous copy of the Near-radius table, a synchronous copy of the descriptor's address and a k_init launch
// with its own synchronisation - 5.4 s for the 8192 trees of the bench.
static int validate_config(const nirrt_config *cfg)
{
    if (cfg->dim != 2 && cfg->dim != 3) { g_err = "dim must be 2 or 3"; return NIRRT_E_ARG; }
    if (cfg->iter_max < 0 || cfg->iter_max + 1 > (1ll << 30)) { g_err = "iter_max out of range"; return NIRRT_E_ARG; }
    if (cfg->n_round < 0 || cfg->n_round > MAX_OBS || cfg->n_box < 0 || cfg->n_box > MAX_OBS) {
        g_err = "too many obstacles (NIRRT_MAX_OBSTACLES per kind)";
        return NIRRT_E_CAPACITY;
    }
    if ((cfg->n_round > 0 && !cfg->round_obs) || (cfg->n_box > 0 && !cfg->box_obs)) { g_err = "null obstacle table"; return NIRRT_E_ARG; }
    if (4 * cfg->n_round + 6 * cfg->n_box > OB_POOL) {
        g_err = "obstacle tables exceed the LDS pool (4 * n_round + 6 * n_box <= NIRRT_OBSTACLE_POOL)";
        return NIRRT_E_CAPACITY;
    }
    return NIRRT_OK;
}

// f(n) of the Near radius (near_factor) resident on the device, once per (device, dimension, capacity)
static const double *near_factor_dev(int device, int D, int cap)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, double *> tabs;
    const std::vector<double> &f = near_factor(D, cap);
    std::lock_guard<std::mutex> g(mu);
    double *&d = tabs[std::make_tuple(device, D, cap)];
    if (!d) {
        if (hipMalloc(&d, sizeof(double) * f.size()) != hipSuccess) { (void)hipGetLastError(); d = nullptr; return nullptr; }
        if (hipMemcpy(d, f.data(), sizeof(double) * f.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); d = nullptr; return nullptr; }
    }
    return d;
}

// everything of a tree that needs no answer from the device
static int prepare_tree(const nirrt_config *cfg, nirrt_tree **out, CreateTimer &tm)
{
    nirrt_tree *t = new nirrt_tree();
    std::memset(&t->host, 0, sizeof(TreeDev));
    t->cfg = *cfg;
    t->cfg.round_obs = nullptr;
    t->cfg.box_obs = nullptr;
    t->dim = cfg->dim;
    t->cap = (int)(cfg->iter_max + 1);
    t->device = cfg->device_id;
    t->stream = nullptr;
    t->arena = nullptr; t->arena_bytes = 0; t->arena_pooled = false; t->pc_own = nullptr; t->mt = nullptr; t->dev = nullptr; t->self_dev = nullptr; t->near_r = nullptr; t->scratch = nullptr; t->scratch_dev = nullptr; t->pc_dev = nullptr;
    t->sblk = nullptr;
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
    // A tree's calls are ordered on its stream.  Thousands of trees do not get a stream each (0.44 ms per hipStreamCreate, and a
    // handful of hardware queues behind them anyway): the trees of a device share a pool of 32, dealt round-robin.
    t->stream = tree_stream(t->device);
    if (!t->stream) return fail(NIRRT_E_HIP);
    tm.lap(0);
    TreeDev &h = t->host;
    const size_t np = (size_t)t->cap + SCAN_PAD;   // padded element count of every per-vertex array
    // uniform-grid index: 256^2 / 32^3 cells over the range box (2D, measured at the bench configuration: 256^2 visits
    // 12 % fewer slots than 128^2 and is 5 % faster; 3D IRRT*, 4096 trees, round 5: 32^3 visits 2490 slots per iteration where
    // 16^3 visited 3600 - 11.3 vs 10.7 M it/s; 24^3: 9.9, 40^3: 10.7 with twice the whole-tree fallbacks)
    h.g_G = D == 2 ? 256 : 32;
    if (const char *e = std::getenv("NIRRT_GRID_G")) h.g_G = std::min(D == 2 ? 1024 : 64, std::max(1, std::atoi(e)));
    h.g_ncell = D == 2 ? h.g_G * h.g_G : h.g_G * h.g_G * h.g_G;
    h.g_lgG = 0;   // tiled g_start (nirrt_device.hpp, grid_word) for power-of-two grids; NIRRT_GRID_TILE=0 keeps it row-major (A/B, tests)
    if (h.g_G >= 8 && (h.g_G & (h.g_G - 1)) == 0 && env_int("NIRRT_GRID_TILE", 1) != 0)
        for (int g = h.g_G; g > 1; g >>= 1) h.g_lgG++;
    // second level over the vertices appended since the last full rebuild: 32^2 / 8^3 coarse cells, re-sorted every 64 insertions
    h.g_G2 = D == 2 ? 32 : 8;
    if (const char *e = std::getenv("NIRRT_GRID_G2")) h.g_G2 = std::min(D == 2 ? 64 : 16, std::max(1, std::atoi(e)));
    h.g_ncell2 = D == 2 ? h.g_G2 * h.g_G2 : h.g_G2 * h.g_G2 * h.g_G2;
    // One allocation per tree (~12 MB at 50k vertices in 2D), carved into its arrays: a workgroup's scattered accesses then
    // fall into half a dozen 2 MB pages instead of ~35 separately placed buffers of 50 - 2400 KB (address translation, not
    // HBM, is what a latency-bound chase across 8192 such trees pays for).  Arrays the loop body touches every iteration
    // come first, next to each other.
    {
        struct Piece { void **dst; size_t bytes; };
        std::vector<Piece> pieces;
        auto want = [&](auto **dst, size_t count) { pieces.push_back({(void **)dst, sizeof(**dst) * count}); };
        want(&t->dev, 1);
        want(&t->self_dev, 1);
        want(&t->near_r, (size_t)t->cap + 1);
        want(&t->pc_own, (size_t)PC_OWN_POINTS * 3);
        want(&t->mt, 1);
        want(&h.vrec, np);
        want(&h.topo, np);
        want(&h.g_rec, np * (size_t)(D == 2 ? SlotBytes<2>::value : SlotBytes<3>::value));   // (char elements: packed slot records)
        want(&h.g_start, (size_t)h.g_ncell + 1);
        want(&h.g_start2, (size_t)h.g_ncell2 + 1);
        want(&h.sol, np); want(&h.sol_line, np); want(&h.sol_val, np);
        want(&h.gc_idx, np); want(&h.gc_dist, np); want(&h.gc_col, np);
        want(&h.nr_idx, np); want(&h.nr_m, np);
        want(&h.bfs_q, np); want(&h.bfs_fc, np); want(&h.chain_g, np);
        want(&h.tie_stamp, np);
        want(&h.g_cnt, (size_t)std::max(h.g_ncell, h.g_ncell2)); want(&h.g_rank, np);   // (both levels' rebuilds count in it)
        const size_t A = 256;
        size_t total = 0;
        for (const Piece &pc : pieces) total += (pc.bytes + A - 1) / A * A;
        t->arena_bytes = total;
        t->arena = g_pool.take(t->device, total, &t->arena_bytes);   // (arena_bytes = the block's size: what goes back to the pool)
        t->arena_pooled = t->arena != nullptr;
        if (!t->arena) { t->arena_bytes = total; HIPCHK_T(hipMalloc(&t->arena, total)); }
        size_t off = 0;
        for (const Piece &pc : pieces) { *pc.dst = (char *)t->arena + off; off += (pc.bytes + A - 1) / A * A; }
    }
    tm.lap(1);
    // (no memsets: the loop writes a vertex's records before anything reads them; k_init clears the rewire stamps and the generators)
    h.mt = t->mt;
    h.cap = t->cap;
    h.dim = D;
    h.cap_sol = t->cap;
    h.g_ns = 0;
    h.g_ns2 = 0;
    h.g_rho = 0.;
    h.g_min = GRID_MIN_VERTICES;
    h.g_every = GRID_REBUILD_EVERY;
    if (const char *e = std::getenv("NIRRT_GRID_MIN")) h.g_min = std::max(2, std::atoi(e));         // test / tuning knobs
    if (const char *e = std::getenv("NIRRT_GRID_REBUILD")) h.g_every = std::max(1, std::atoi(e));
    h.g_every2 = std::max(1, h.g_every / 16);
    if (const char *e = std::getenv("NIRRT_GRID_REBUILD2")) h.g_every2 = std::max(1, std::atoi(e));   // >= g_every: no second level
    for (int k = 0; k < 3; k++) {
        double ext = k < D ? cfg->range_hi[k] - cfg->range_lo[k] : 1.0;
        if (!(ext > 0.)) ext = 1.0;
        h.g_inv_h[k] = (double)h.g_G / ext;
        h.g_margin[k] = ext / (double)h.g_G / 256.0;
        h.g_inv_h2[k] = (double)h.g_G2 / ext;
        h.g_margin2[k] = ext / (double)h.g_G2 / 256.0;
        h.g_h[k] = 1.0 / h.g_inv_h[k];
        h.g_h2[k] = 1.0 / h.g_inv_h2[k];
    }
    tm.lap(2);
    // Near radius r(n) = min(gamma * f(n), step_len) (rrt_star_2d.py:133, rrt_star_3d.py:134): f(n) with the host's libm once per
    // capacity (near_factor), resident on the device; k_init multiplies by the tree's gamma - the same IEEE product and comparison
    // the host made per tree in round 5 (400 KB copied per tree)
    h.near_f = near_factor_dev(t->device, D, t->cap);
    if (!h.near_f) { g_err = "Near-radius table: device allocation failed"; return fail(NIRRT_E_HIP); }
    h.gamma = cfg->search_radius;
    h.self_slot = t->self_dev;
    h.near_r = t->near_r;
    tm.lap(4);
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
    h.n = 1;
    host_mirror_reset(t);
    *out = t;
    return NIRRT_OK;
#undef HIPCHK_T
}

// the device pass for prepared trees of ONE device and dimension (ts[0 .. n)): pinned result slots (one block for all of them),
// descriptors, one k_init launch
static int finish_trees(nirrt_tree **ts, int n, CreateTimer &tm)
{
    nirrt_tree *t0 = ts[0];
    HIPCHK(hipSetDevice(t0->device));
    ScratchBlock *blk = new ScratchBlock();
    blk->base = nullptr;
    blk->refs = 0;
    if (hipHostMalloc(&blk->base, sizeof(Scratch) * (size_t)n, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        delete blk;
        g_err = "hipHostMalloc of the result slots failed";
        return NIRRT_E_HIP;
    }
    Scratch *dev_alias = nullptr;
    hipError_t e = hipHostGetDevicePointer((void **)&dev_alias, blk->base, 0);
    for (int i = 0; i < n; i++) {   // (every tree holds a reference from here on: nirrt_destroy releases the block with the last one)
        ts[i]->sblk = blk;
        blk->refs++;
        ts[i]->scratch = (Scratch *)blk->base + i;
        ts[i]->scratch_dev = dev_alias ? dev_alias + i : nullptr;
    }
    if (e != hipSuccess) { g_err = std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    tm.lap(3);
    hipStream_t st = t0->stream;
    std::vector<TreeDev *> ptrs((size_t)n);
    for (int i = 0; i < n; i++) {
        ptrs[(size_t)i] = ts[i]->dev;
        HIPCHK(hipMemcpyAsync(ts[i]->dev, &ts[i]->host, sizeof(TreeDev), hipMemcpyHostToDevice, st));
    }
    TreeDev **d_ptrs = nullptr;
    size_t got = 0;
    HIPCHK(g_scratch.take(t0->device, sizeof(TreeDev *) * (size_t)n, (void **)&d_ptrs, &got));
    e = hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(TreeDev *) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const Variant fv = forced_variant();
        LAUNCH_V(fv == V_AUTO ? V_WIDE : fv, t0->dim, k_init, n, st, (TreeDev *const *)d_ptrs, 2);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    g_scratch.give(t0->device, got, d_ptrs);
    if (e != hipSuccess) { g_err = std::string("nirrt_create: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    tm.lap(5);
    return NIRRT_OK;
}

extern "C" int nirrt_create_batch(const nirrt_config *cfgs, int32_t n, nirrt_tree **out)
{
    CreateTimer tm;
    if (!cfgs || !out || n <= 0) { g_err = "null argument"; return NIRRT_E_ARG; }
    for (int i = 0; i < n; i++) out[i] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        g_err = "no HIP device visible";
        return NIRRT_E_NODEVICE;
    }
    for (int i = 0; i < n; i++) {
        int rc = validate_config(&cfgs[i]);
        if (rc) return rc;
        if (cfgs[i].device_id < 0 || cfgs[i].device_id >= ndev) { g_err = "device_id out of range"; return NIRRT_E_ARG; }
    }
    auto undo = [&](int rc) {
        for (int i = 0; i < n; i++) { if (out[i]) nirrt_destroy(out[i]); out[i] = nullptr; }
        return rc;
    };
    for (int i = 0; i < n; i++) {
        int rc = prepare_tree(&cfgs[i], &out[i], tm);
        if (rc) return undo(rc);
    }
    // one device pass per (device, dimension) group, in the caller's order within a group
    std::vector<char> done((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (done[(size_t)i]) continue;
        std::vector<nirrt_tree *> grp;
        for (int j = i; j < n; j++)
            if (!done[(size_t)j] && out[j]->device == out[i]->device && out[j]->dim == out[i]->dim) { grp.push_back(out[j]); done[(size_t)j] = 1; }
        int rc = finish_trees(grp.data(), (int)grp.size(), tm);
        if (rc) return undo(rc);
    }
    for (int i = 0; i < n; i++) tm.done();
    return NIRRT_OK;
}

extern "C" int nirrt_create(const nirrt_config *cfg, nirrt_tree **out)
{
    if (!cfg || !out) { g_err = "null argument"; return NIRRT_E_ARG; }
    return nirrt_create_batch(cfg, 1, out);
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
    // the tree is rooted at the start state (RRTBase.__init__: vertices[0] = x_start, rrt_base_2d.py:27): the loop body
    // relies on it (cost(v) >= |v - x_start|, the floor of the Near stash), so a frozen tree with another root is refused
    for (int k = 0; k < D; k++)
        if (vertices[k] != t->cfg.x_start[k]) { g_err = "nirrt_upload: vertex 0 must be x_start (the tree is rooted at the start state)"; return NIRRT_E_ARG; }
    {   // coordinates go straight into the per-vertex records (k_init fills in the costs and the slot records)
        std::vector<VRec> rec((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            VRec &r = rec[(size_t)i];
            r.x = vertices[i * D]; r.y = vertices[i * D + 1]; r.z = D == 3 ? vertices[i * D + 2] : 0.; r.cost = 0.;
        }
        HIPCHK(hipMemcpy(t->host.vrec, rec.data(), sizeof(VRec) * (size_t)n, hipMemcpyHostToDevice));
    }
    std::vector<Topo> ax((size_t)n);   // each vertex's own edge (e[0], a[0]); k_init derives the rest of the records
    std::memset(ax.data(), 0, sizeof(Topo) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (parents[i] < 0 || parents[i] >= n) { g_err = "parent index out of range"; return NIRRT_E_ARG; }
        double d[3] = {0., 0., 0.};
        for (int k = 0; k < D; k++) d[k] = vertices[i * D + k] - vertices[parents[i] * D + k];
        ax[(size_t)i].e[0] = i == 0 ? 0. : host_hypot_py(D, d);
        ax[(size_t)i].a[0] = (int)parents[i];
    }
    HIPCHK(hipMemcpy(t->host.topo, ax.data(), sizeof(Topo) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(t->host.tie_stamp, 0, sizeof(int) * ((size_t)t->cap + SCAN_PAD)));   // the descriptor push below restarts rw_stamp at 0
    t->host.n = (int)n;
    t->last_n = n;
    t->host.n_sol = 0;
    t->host.n_gc = 0;
    t->host.status = 0;
    int rc = push_desc(t);
    if (rc) return rc;
    DISPATCH_DIM(t, k_init, 1, (TreeDev *const *)t->self_dev, 0);
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
        std::vector<VRec> rec((size_t)n);
        HIPCHK(hipMemcpy(rec.data(), t->host.vrec, sizeof(VRec) * (size_t)n, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            const VRec &r = rec[(size_t)i];
            vertices[i * D] = r.x; vertices[i * D + 1] = r.y;
            if (D == 3) vertices[i * D + 2] = r.z;
        }
    }
    if (parents) {
        std::vector<Topo> ax((size_t)n);
        HIPCHK(hipMemcpy(ax.data(), t->host.topo, sizeof(Topo) * (size_t)n, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) parents[i] = ax[(size_t)i].a[0];
    }
    if (n_out) *n_out = n;
    return NIRRT_OK;
}

extern "C" int nirrt_nearest(nirrt_tree *t, const double *q, int64_t *idx)
{
    if (!t || !q || !idx) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    DISPATCH_BY_SIZE(t, k_nearest, t->dev, q[0], q[1], t->dim == 3 ? q[2] : 0., &t->scratch_dev->i[0]);
    int rc = sync_check(t);
    if (rc) return rc;
    *idx = t->scratch->i[0];
    return NIRRT_OK;
}

static int grid_for(long long n)
{
    long long g = (n + NT_WIDE - 1) / NT_WIDE;
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
        std::sort(tmp.begin(), tmp.end());   // the kernel leaves the members in visiting order; np.where lists them ascending
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
    DISPATCH_BY_SIZE(t, k_step, t->dev, p[0], p[1], t->dim == 3 ? p[2] : 0., host_steer, (int)nearest_idx, (unsigned)flags,
                     &t->scratch_dev->step);
    int rc = sync_check(t);
    if (rc) return rc;
    *res = t->scratch->step;
    t->last_n = res->n;
    if (res->status) {
        g_err = res->status == NIRRT_E_LIBM ? "the restated libm routines returned NaN for this steer (argument outside their supported domain)"
                                            : "capacity exceeded inside step";
        return res->status;
    }
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
    const size_t off = offsetof(TreeHotH, c_min), end = offsetof(TreeHotH, pc);   // c_min, x_center, CL_C
    HIPCHK(hipMemcpyAsync((char *)t->dev + off, (char *)&t->host + off, end - off, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return NIRRT_OK;
}

// IRRTStar.init for a whole batch: tree i gets (c_min[i], x_center[3 i ..], C[9 i ..]) - one copy and one launch instead of a copy
// and a synchronisation per tree
__global__ void k_set_informed(TreeDev *const *trees, int n, const double *vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TreeDev *t = trees[i];
    const double *v = vals + (size_t)i * 13;
    t->c_min = v[0];
    for (int k = 0; k < 3; k++) t->x_center[k] = v[1 + k];
    for (int k = 0; k < 9; k++) t->CL_C[k] = v[4 + k];
}

extern "C" int nirrt_set_informed_batch(nirrt_tree *const *trees, int32_t n_trees, const double *c_min, const double *x_center, const double *C)
{
    if (!c_min || !x_center || !C) return NIRRT_E_ARG;
    TreeList tl;
    int rc = tree_list(trees, n_trees, "nirrt_set_informed_batch", tl);
    if (rc) return rc;
    std::vector<double> vals((size_t)n_trees * 13);
    for (int i = 0; i < n_trees; i++) {
        nirrt_tree *t = trees[i];
        double *v = &vals[(size_t)i * 13];
        v[0] = c_min[i];
        for (int k = 0; k < 3; k++) v[1 + k] = k < t->dim ? x_center[(size_t)i * 3 + k] : 0.;
        for (int k = 0; k < 9; k++) v[4 + k] = C[(size_t)i * 9 + k];
        t->host.c_min = v[0];
        for (int k = 0; k < 3; k++) t->host.x_center[k] = v[1 + k];
        for (int k = 0; k < 9; k++) t->host.CL_C[k] = v[4 + k];
    }
    hipStream_t st = trees[0]->stream;
    double *d = nullptr;
    size_t got = 0;
    HIPCHK(g_scratch.take(tl.device, sizeof(double) * vals.size(), (void **)&d, &got));
    hipError_t e = hipMemcpyAsync(d, vals.data(), sizeof(double) * vals.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_set_informed, dim3((n_trees + 255) / 256), dim3(256), 0, st, (TreeDev *const *)tl.d, (int)n_trees, (const double *)d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    g_scratch.give(tl.device, got, d);
    if (e != hipSuccess) { g_err = std::string("nirrt_set_informed_batch: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    return NIRRT_OK;
}

// Utils.is_collision for ONE segment per tree, each against its own tree's obstacles (the free-segment probe of a batch of
// problems: is the straight start-goal segment free?): one thread per tree, the tables read from the descriptors
template <int D>
__global__ void k_collision_each(TreeDev *const *trees, int n, const double *seg, unsigned char *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TreeDev *t = trees[i];
    double a[D], b[D];
#pragma unroll
    for (int k = 0; k < D; k++) { a[k] = seg[(size_t)i * 2 * D + k]; b[k] = seg[(size_t)i * 2 * D + D + k]; }
    const double clr = t->clearance;
    bool hit = false;
    for (int o = 0; o < t->n_round && !hit; o++) hit = D == 2 ? seg_round_2d(a, b, t->rnd[o], clr) : seg_round_3d(a, b, t->rnd[o], clr);
    for (int o = 0; o < t->n_box && !hit; o++) hit = D == 2 ? seg_box_2d(a, b, t->box[o], clr) : seg_box_3d(a, b, t->box[o], clr);
    out[i] = hit ? 1 : 0;
}

extern "C" int nirrt_collision_each(nirrt_tree *const *trees, int32_t n_trees, const double *seg, uint8_t *out)
{
    if (!seg || !out) return NIRRT_E_ARG;
    TreeList tl;
    int rc = tree_list(trees, n_trees, "nirrt_collision_each", tl);
    if (rc) return rc;
    const int D = trees[0]->dim;
    for (int i = 0; i < n_trees; i++)
        if (trees[i]->dim != D) { g_err = "nirrt_collision_each: all trees must share dim"; return NIRRT_E_ARG; }
    hipStream_t st = trees[0]->stream;
    const size_t sb = sizeof(double) * (size_t)n_trees * 2 * D;
    char *d = nullptr;
    size_t got = 0;
    HIPCHK(g_scratch.take(tl.device, sb + (size_t)n_trees + 256, (void **)&d, &got));
    hipError_t e = hipMemcpyAsync(d, seg, sb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        if (D == 2) hipLaunchKernelGGL(k_collision_each<2>, dim3((n_trees + 63) / 64), dim3(64), 0, st, (TreeDev *const *)tl.d, (int)n_trees, (const double *)d, (unsigned char *)(d + sb));
        else hipLaunchKernelGGL(k_collision_each<3>, dim3((n_trees + 63) / 64), dim3(64), 0, st, (TreeDev *const *)tl.d, (int)n_trees, (const double *)d, (unsigned char *)(d + sb));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + sb, (size_t)n_trees, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    g_scratch.give(tl.device, got, d);
    if (e != hipSuccess) { g_err = std::string("nirrt_collision_each: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    return NIRRT_OK;
}

extern "C" int nirrt_set_cloud(nirrt_tree *t, int64_t n, const double *pts, double sample_rate, double update_cost_ratio,
                               double c_update)
{
    if (!t || n < 0 || (n > 0 && !pts)) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    // a refresh replaces the cloud of a stopped tree: the usual 2048-point clouds live in the tree's arena (no allocation,
    // no implicit device synchronisation of hipFree per refresh - a batch refreshes thousands of clouds per run)
    if (t->pc_dev && t->pc_dev != t->pc_own) (void)hipFree(t->pc_dev);
    t->pc_dev = nullptr;
    if (n > 0) {
        if (n <= PC_OWN_POINTS) t->pc_dev = t->pc_own;
        else HIPCHK(hipMalloc(&t->pc_dev, sizeof(double) * (size_t)n * t->dim));
        HIPCHK(hipMemcpyAsync(t->pc_dev, pts, sizeof(double) * (size_t)n * t->dim, hipMemcpyHostToDevice, t->stream));
    }
    t->host.pc = t->pc_dev;
    t->host.pc_n = (int)n;
    t->host.pad2 = 0;
    t->host.pc_rate = sample_rate;
    t->host.pc_ratio = update_cost_ratio;
    t->host.c_update = c_update;
    const size_t off = offsetof(TreeHotH, pc), end = offsetof(TreeHotH, g_rec);   // pc, pc_n, pc_rate, pc_ratio, c_update
    HIPCHK(hipMemcpyAsync((char *)t->dev + off, (char *)&t->host + off, end - off, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return NIRRT_OK;
}

// update_point_cloud's last step for a whole batch (nirrt_star_png_2d.py:166-174: self.path_point_cloud_pred = pc[path_pred
// .nonzero()[0]]): tree i gets the points of its cloud whose prediction is non-zero, in order, plus the sampling-policy
// scalars - straight into its arena and descriptor, one workgroup per tree, no per-tree copies.
struct SetCloudJob {
    TreeDev *tree;
    double *pc_own;            // the tree's cloud buffer (PC_OWN_POINTS points)
    const double *cloud;       // DEVICE (n, 3) f64
    const unsigned char *pred; // DEVICE (n,) bytes
    int n, dim;
    double rate, ratio, c_update;
};
__global__ __launch_bounds__(256) void k_set_clouds(const SetCloudJob *jobs, int *n_path)
{
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    const SetCloudJob jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < jb.n; i0 += 256) {
        const int i = i0 + tid;
        const bool keep = i < jb.n && jb.pred[i] != 0;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = base_s, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (k < wv) off += wave_tot[k]; tot += wave_tot[k]; }
        if (keep) {
            const int p = off + __popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < jb.dim; k++) jb.pc_own[(size_t)p * jb.dim + k] = jb.cloud[3 * (size_t)i + k];
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    if (tid == 0) {
        TreeDev *t = jb.tree;
        t->pc = jb.pc_own; t->pc_n = base_s; t->pad2 = 0;
        t->pc_rate = jb.rate; t->pc_ratio = jb.ratio; t->c_update = jb.c_update;
        n_path[blockIdx.x] = base_s;
    }
}

extern "C" int nirrt_set_cloud_batch(nirrt_tree *const *trees, int32_t n_trees, const double *clouds, int64_t cloud_stride,
                                     const int32_t *n_points, const uint8_t *pred, int64_t pred_stride, double sample_rate,
                                     double update_cost_ratio, const double *c_update, int32_t *n_path_out)
{
    if (!trees || n_trees <= 0 || !clouds || !n_points || !pred || !c_update) return NIRRT_E_ARG;
    nirrt_tree *t0 = trees[0];
    HIPCHK(hipSetDevice(t0->device));
    std::vector<SetCloudJob> jobs((size_t)n_trees);
    for (int i = 0; i < n_trees; i++) {
        nirrt_tree *t = trees[i];
        if (!t || t->device != t0->device || t->dim != t0->dim || n_points[i] < 0 || n_points[i] > PC_OWN_POINTS) {
            g_err = "nirrt_set_cloud_batch: trees must share device and dim, clouds hold at most 4096 points";
            return NIRRT_E_ARG;
        }
        HIPCHK(hipStreamSynchronize(t->stream));
        if (t->pc_dev && t->pc_dev != t->pc_own) (void)hipFree(t->pc_dev);
        t->pc_dev = t->pc_own;
        jobs[(size_t)i] = SetCloudJob{t->dev, t->pc_own, clouds + (size_t)i * (size_t)cloud_stride, pred + (size_t)i * (size_t)pred_stride,
                                      n_points[i], t->dim, sample_rate, update_cost_ratio, c_update[i]};
    }
    // grow-only scratch kept per device: this runs while the other half of a batch is inside a persistent launch, and hipFree
    // would wait for that launch
    static std::mutex mu;
    static void *scratch[16] = {nullptr};
    static size_t scratch_cap[16] = {0};
    std::lock_guard<std::mutex> hold(mu);
    const size_t need = (sizeof(SetCloudJob) + sizeof(int)) * (size_t)n_trees + 256;
    const int dv = t0->device & 15;
    if (scratch_cap[dv] < need) {
        if (scratch[dv]) (void)hipFree(scratch[dv]);
        scratch[dv] = nullptr; scratch_cap[dv] = 0;
        HIPCHK(hipMalloc(&scratch[dv], 2 * need));
        scratch_cap[dv] = 2 * need;
    }
    SetCloudJob *d_jobs = (SetCloudJob *)scratch[dv];
    int *d_n = (int *)((char *)scratch[dv] + ((sizeof(SetCloudJob) * (size_t)n_trees + 255) & ~(size_t)255));
    std::vector<int> n_path((size_t)n_trees);
    hipError_t e = hipMemcpyAsync(d_jobs, jobs.data(), sizeof(SetCloudJob) * (size_t)n_trees, hipMemcpyHostToDevice, t0->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_set_clouds, dim3(n_trees), dim3(256), 0, t0->stream, (const SetCloudJob *)d_jobs, d_n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(n_path.data(), d_n, sizeof(int) * (size_t)n_trees, hipMemcpyDeviceToHost, t0->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t0->stream);
    if (e != hipSuccess) { g_err = std::string("nirrt_set_cloud_batch: ") + hipGetErrorString(e); return NIRRT_E_HIP; }
    for (int i = 0; i < n_trees; i++) {   // host mirrors follow (later per-tree patches copy from them)
        nirrt_tree *t = trees[i];
        t->host.pc = t->pc_own; t->host.pc_n = n_path[(size_t)i]; t->host.pad2 = 0;
        t->host.pc_rate = sample_rate; t->host.pc_ratio = update_cost_ratio; t->host.c_update = c_update[i];
        if (n_path_out) n_path_out[i] = n_path[(size_t)i];
    }
    return NIRRT_OK;
}

// {n, status, stat[NSTAT]} of every tree of a batch in one array: one small kernel + one copy instead of a descriptor
// read per tree (8192 trees per launch in the bench)
#define COLLECT_W (NSTAT + 2)
__global__ void k_collect(TreeDev *const *trees, int n, long long *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const TreeDev *t = trees[i];
        long long *o = out + (size_t)i * COLLECT_W;
        o[0] = t->n;
        o[1] = t->status;
        for (int j = 0; j < NSTAT; j++) o[2 + j] = t->stat[j];
    }
}

// per-tree counters of one launch: the descriptor accumulates since reset, a launch reports the difference
// (after / before: rows of k_collect)
static void report_stats(const nirrt_run_args *a, int i, const long long *after, const long long *before)
{
    const long long *sa = after + 2, *sb = before + 2;
    if (a->scan_elems) a->scan_elems[i] = sa[ST_VISITED] - sb[ST_VISITED];
    if (a->alg_elems) a->alg_elems[i] = sa[ST_ALG] - sb[ST_ALG];
    if (a->stats) {
        for (int j = 0; j < NSTAT; j++) a->stats[(size_t)i * NSTAT + j] = sa[j] - sb[j];
        a->stats[(size_t)i * NSTAT + ST_T0] = sa[ST_T0];
        a->stats[(size_t)i * NSTAT + ST_T1] = sa[ST_T1];
        a->stats[(size_t)i * NSTAT + ST_CBEST] = sa[ST_CBEST];
    }
}

// Lane groups of one nirrt_run call must run AT THE SAME TIME (the few wide workgroups of the heavy trees next to the many
// one-wave trees).  Tree streams do not guarantee that: thousands of them share the handful of hardware queues, and two
// launches that land on the same queue run one after the other (measured: the 64-lane group started the moment the 256-lane
// group ended).  Three streams per device, created back to back (consecutive streams get consecutive hardware queues), the
// first with the highest priority - it carries the widest, longest-running trees.
// One set per calling THREAD and device: concurrent nirrt_run calls from worker threads (run_batch with NIRRT_BATCH_GROUPS /
// NIRRT_BATCH_INFLIGHT > 1) must not share streams - their launches would serialize and each call's event timing would contain
// the other call's kernels.  (The few streams of a worker thread that ends are not destroyed: HIP may already be shutting down.)
static hipStream_t *group_streams(int device)
{
    struct Set { hipStream_t st[3]; };
    thread_local std::map<int, Set> sets;
    auto it = sets.find(device);
    if (it == sets.end()) {
        Set s;
        int lo = 0, hi = 0;   // (numerically lower = higher priority)
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
        for (int i = 0; i < 3; i++)
            if (hipStreamCreateWithPriority(&s.st[i], hipStreamNonBlocking, i == 0 ? hi : lo) != hipSuccess) {
                (void)hipGetLastError();
                for (int j = 0; j < i; j++) (void)hipStreamDestroy(s.st[j]);
                return nullptr;
            }
        it = sets.emplace(device, s).first;
    }
    return it->second.st;
}

// workgroups of the time-sliced loop that are resident at once (occupancy of the kernel x compute units)
static int resident_workgroups(Variant v, int D, int device)
{
    int per_cu = 0, cus = 0;
    const void *fn = nullptr;
    int threads = NT_SLIM;
    switch (v) {
    case V_SLIM: fn = D == 2 ? (const void *)slim::k_run_pool<2> : (const void *)slim::k_run_pool<3>; threads = NT_SLIM; break;
    case V_NARROW: fn = D == 2 ? (const void *)narrow::k_run_pool<2> : (const void *)narrow::k_run_pool<3>; threads = NT_NARROW; break;
    default: fn = D == 2 ? (const void *)wide::k_run_pool<2> : (const void *)wide::k_run_pool<3>; threads = NT_WIDE; break;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * cus;
}

static int run_sampling(nirrt_tree *const *trees, int32_t n_trees, const nirrt_run_args *a)
{
    nirrt_tree *t0 = trees[0];
    const int D = t0->dim;
    hipStream_t st = t0->stream;
    const bool gen_mode = a->np_words == nullptr;   // the trees' own generators (nirrt_set_generators) produce the words
    if (!a->np_used || !a->py_used || !a->iters_done || (!gen_mode && !a->n_np)) {
        g_err = "nirrt_run: sampling mode needs np_used, py_used, iters_done (+ n_np with np_words)";
        return NIRRT_E_ARG;
    }
    if (gen_mode && a->py_words) { g_err = "nirrt_run: py_words without np_words (generator mode uses the trees' own streams for both)"; return NIRRT_E_ARG; }
    const bool need_py = (a->flags & NIRRT_F_IRRT) && D == 2;
    if (!gen_mode && need_py && (!a->py_words || !a->n_py)) { g_err = "nirrt_run: 2D IRRT* sampling needs py_words"; return NIRRT_E_ARG; }
    std::vector<std::pair<void *, size_t>> to_free;
    const int dev_id = t0->device;
    auto cleanup = [&]() { for (auto &p : to_free) g_scratch.give(dev_id, p.second, p.first); to_free.clear(); };
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
        size_t got = 0;
        hipError_t e = g_scratch.take(dev_id, bytes ? bytes : 8, out, &got);
        if (e == hipSuccess) to_free.push_back({*out, got});
        return e;
    };
    // Launch groups.  By default the whole batch runs on one kernel instantiation (run_variant).  With lanes_hint the
    // caller asks for wider workgroups for some trees (64 / 128 / 256 lanes): the batch is split into up to three groups,
    // each launched on its own stream at the same time - a tree that is known to be heavy (its Near sets will be large) gets
    // more lanes instead of holding up the launch on one wave.  perm[j] = index of the tree at launch position j.
    long long n_hi = 0;
    for (int i = 0; i < n_trees; i++) n_hi = std::max(n_hi, trees[i]->last_n);
    const Variant v_all = run_variant(n_trees, a->flags, n_hi, a->iters);
    std::vector<Variant> want((size_t)n_trees, v_all);
    if (a->lanes_hint && forced_variant() == V_AUTO) {
        for (int i = 0; i < n_trees; i++) {
            const int h = a->lanes_hint[i];
            if (h == 64) want[(size_t)i] = V_SLIM; else if (h == 128) want[(size_t)i] = V_NARROW; else if (h == 256) want[(size_t)i] = V_WIDE;
            else if (h != 0) { g_err = "nirrt_run: lanes_hint[i] must be 0, 64, 128 or 256"; return NIRRT_E_ARG; }
        }
    }
    std::vector<int> perm;
    perm.reserve((size_t)n_trees);
    struct Group { Variant v; int b0, b1; hipStream_t st; hipEvent_t e0, e1; };
    std::vector<Group> groups;
    for (Variant v : {V_WIDE, V_NARROW, V_SLIM}) {   // widest first: those are the long-running trees
        const int b0 = (int)perm.size();
        for (int i = 0; i < n_trees; i++)
            if (want[(size_t)i] == v) perm.push_back(i);
        if ((int)perm.size() > b0) groups.push_back(Group{v, b0, (int)perm.size(), trees[perm[(size_t)b0]]->stream, nullptr, nullptr});
    }
    // The kernels are launched and timed on streams that belong to the calling thread (one group: the first of its set).  The
    // trees' own streams come from a pool of 32 per device: two nirrt_run calls from two threads could land on the same one -
    // their launches would serialize and each call's kernel_ms would contain the other's kernels.
    if (hipStream_t *gs = group_streams(t0->device))
        for (size_t gi = 0; gi < groups.size(); gi++) groups[gi].st = gs[groups.size() > 1 ? gi : 2];   // (one group: a normal-priority stream)
    bool identity = true;
    for (int j = 0; j < n_trees; j++) identity = identity && perm[(size_t)j] == j;
    std::vector<TreeDev *> ptrs((size_t)n_trees);
    for (int j = 0; j < n_trees; j++) ptrs[(size_t)j] = trees[perm[(size_t)j]]->dev;
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
        for (int j = 0; j < n_trees; j++) { dn[(size_t)j] = cnt[perm[(size_t)j]]; total += (size_t)dn[(size_t)j]; }
        if (a->inputs_on_device) {
            for (int j = 0; j < n_trees; j++) dst[(size_t)j] = src[perm[(size_t)j]];
        } else {
            unsigned *slab = nullptr;
            HIPCHK_R(dalloc(sizeof(unsigned) * total, (void **)&slab));
            size_t off = 0;
            for (int j = 0; j < n_trees; j++) {
                const int i = perm[(size_t)j];
                if (cnt[i] > 0) HIPCHK_R(hipMemcpyAsync(slab + off, src[i], sizeof(unsigned) * (size_t)cnt[i], hipMemcpyHostToDevice, st));
                dst[(size_t)j] = slab + off;
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
    // (a tree whose budget is 0 is never run by the time-sliced kernel: its outputs are zeros, not what the scratch pool held)
    HIPCHK_R(hipMemsetAsync(d_npu, 0, sizeof(long long) * nt, st));
    HIPCHK_R(hipMemsetAsync(d_pyu, 0, sizeof(long long) * nt, st));
    HIPCHK_R(hipMemsetAsync(d_done, 0, sizeof(long long) * nt, st));
    HIPCHK_R(hipMemsetAsync(d_stop, 0, sizeof(int) * nt, st));
    HIPCHK_R(hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(void *) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_npp, npp.data(), sizeof(void *) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_pyp, pyp.data(), sizeof(void *) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_nnp, nnp.data(), sizeof(long long) * nt, hipMemcpyHostToDevice, st));
    HIPCHK_R(hipMemcpyAsync(d_npy, npy.data(), sizeof(long long) * nt, hipMemcpyHostToDevice, st));
    long long *d_each = nullptr;
    std::vector<long long> each;      // (function scope: the asynchronous copy below reads it; the stream is synchronized further down)
    if (a->iters_each) {
        each.resize(nt);
        for (int j = 0; j < n_trees; j++) {
            const long long e = a->iters_each[perm[(size_t)j]];
            if (e < 0 || e > a->iters) { g_err = "nirrt_run: iters_each[i] must be in [0, iters]"; cleanup(); return NIRRT_E_ARG; }
            each[(size_t)j] = e;
        }
        HIPCHK_R(dalloc(sizeof(long long) * nt, (void **)&d_each));
        HIPCHK_R(hipMemcpyAsync(d_each, each.data(), sizeof(long long) * nt, hipMemcpyHostToDevice, st));
    }
    int *d_ahead = nullptr;
    if (a->run_ahead) {
        std::vector<int> ah(nt);
        for (int j = 0; j < n_trees; j++) ah[(size_t)j] = a->run_ahead[perm[(size_t)j]] != 0;
        HIPCHK_R(dalloc(sizeof(int) * nt, (void **)&d_ahead));
        HIPCHK_R(hipMemcpyAsync(d_ahead, ah.data(), sizeof(int) * nt, hipMemcpyHostToDevice, st));
        HIPCHK_R(hipStreamSynchronize(st));   // (the vector goes out of scope)
    }
    int *d_park = nullptr;
    if (a->park_limit > 0 && (a->flags & NIRRT_F_PNG)) {
        HIPCHK_R(dalloc(256, (void **)&d_park));
        HIPCHK_R(hipMemsetAsync(d_park, 0, 256, st));
    }
    long long *d_col = nullptr;   // k_collect rows before / after the launch
    HIPCHK_R(dalloc(sizeof(long long) * 2 * nt * COLLECT_W, (void **)&d_col));
    hipLaunchKernelGGL(k_collect, dim3((n_trees + 255) / 256), dim3(256), 0, st, (TreeDev *const *)d_ptrs, n_trees, d_col);
    HIPCHK_R(hipStreamSynchronize(st));   // inputs in place before any group's stream starts
    for (Group &g : groups) {
        RunSampleDev rd;
        const int o = g.b0;
        rd.iters_each = d_each ? d_each + o : nullptr;
        rd.flags = a->flags; rd.pad = 0; rd.iters = a->iters;
        rd.np_words = gen_mode ? nullptr : d_npp + o; rd.n_np = d_nnp + o; rd.py_words = a->py_words ? d_pyp + o : nullptr; rd.n_py = d_npy + o;
        rd.np_used = d_npu + o; rd.py_used = d_pyu + o; rd.cost_trace = d_trace ? d_trace + (size_t)o * (size_t)a->iters : nullptr;
        rd.iters_done = d_done + o; rd.stop_code = d_stop + o;
        rd.park = d_park; rd.park_limit = a->park_limit; rd.pad2 = 0;
        HIPCHK_R(hipEventCreate(&g.e0));
        HIPCHK_R(hipEventCreate(&g.e1));
        // more trees than the GPU holds at once: a resident set of workgroups shares them in time slices (k_run_pool)
        const int n_g = g.b1 - g.b0;
        int resident = gen_mode ? resident_workgroups(g.v, D, t0->device) : 0;
        if (const int forced = env_int("NIRRT_POOL_RESIDENT", 0)) resident = std::min(resident, forced);   // (tests: a small resident set)
        long long slice = a->slice_iters;
        if (slice == 0) {
            const long long e = env_int("NIRRT_SLICE", -1);
            slice = e >= 0 ? e : std::max<long long>(128, (a->iters + 47) / 48);   // (measured at 50 000 iterations: 3125 -> 49.4, 1024 -> 50.9, 512 -> 51.0 M it/s)
            if (e == 0) slice = -1;
        }
        if (resident > 0 && n_g > resident && slice > 0 && slice < a->iters) {
            int *d_pool = nullptr;   // [0] ticket counter, then state[], round[], fin[]
            HIPCHK_R(dalloc(sizeof(int) * (3 * (size_t)n_g + 64), (void **)&d_pool));
            HIPCHK_R(hipMemsetAsync(d_pool, 0, sizeof(int) * (3 * (size_t)n_g + 64), g.st));
            PoolDev pd;
            pd.n_trees = n_g; pd.pad = 0; pd.quantum = slice; pd.ticket = (unsigned *)d_pool;
            pd.state = d_pool + 64; pd.round = d_pool + 64 + n_g; pd.fin = d_pool + 64 + 2 * n_g;
            pd.ahead = d_ahead ? d_ahead + o : nullptr;
            HIPCHK_R(hipEventRecord(g.e0, g.st));
            LAUNCH_V(g.v, D, k_run_pool, resident, g.st, (TreeDev *const *)(d_ptrs + o), rd, pd);
        } else {
            HIPCHK_R(hipEventRecord(g.e0, g.st));
            LAUNCH_V(g.v, D, k_run_sample, n_g, g.st, (TreeDev *const *)(d_ptrs + o), rd);
        }
        HIPCHK_R(hipEventRecord(g.e1, g.st));
        HIPCHK_R(hipGetLastError());
    }
    float ms_max = 0.f;
    for (Group &g : groups) {
        HIPCHK_R(hipStreamSynchronize(g.st));
    }
    for (Group &g : groups) {
        // first group's start -> this group's end: groups that each fill the GPU run one after the other, whatever their streams
        float ms = 0.f;
        HIPCHK_R(hipEventElapsedTime(&ms, groups[0].e0, g.e1));
        ms_max = std::max(ms_max, ms);
    }
    for (Group &g : groups) {
        (void)hipEventDestroy(g.e0);
        (void)hipEventDestroy(g.e1);
    }
    if (a->kernel_ms) *a->kernel_ms = ms_max;   // device time of the launch: start of the first group to the end of the last
    hipLaunchKernelGGL(k_collect, dim3((n_trees + 255) / 256), dim3(256), 0, st, (TreeDev *const *)d_ptrs, n_trees, d_col + nt * COLLECT_W);
    std::vector<long long> col(2 * nt * COLLECT_W);
    HIPCHK_R(hipMemcpyAsync(col.data(), d_col, sizeof(long long) * col.size(), hipMemcpyDeviceToHost, st));
    HIPCHK_R(hipStreamSynchronize(st));
    std::vector<long long> done(nt), npu(nt), pyu(nt);
    std::vector<int> stop(nt);
    HIPCHK_R(hipMemcpyAsync(done.data(), d_done, sizeof(long long) * nt, hipMemcpyDeviceToHost, st));
    HIPCHK_R(hipMemcpyAsync(npu.data(), d_npu, sizeof(long long) * nt, hipMemcpyDeviceToHost, st));
    HIPCHK_R(hipMemcpyAsync(pyu.data(), d_pyu, sizeof(long long) * nt, hipMemcpyDeviceToHost, st));
    HIPCHK_R(hipMemcpyAsync(stop.data(), d_stop, sizeof(int) * nt, hipMemcpyDeviceToHost, st));
    HIPCHK_R(hipStreamSynchronize(st));
    if (a->cost_trace) {
        if (identity) HIPCHK_R(hipMemcpy(a->cost_trace, d_trace, sizeof(double) * nt * (size_t)a->iters, hipMemcpyDeviceToHost));
        else
            for (int j = 0; j < n_trees; j++)
                HIPCHK_R(hipMemcpy(a->cost_trace + (size_t)perm[(size_t)j] * (size_t)a->iters, d_trace + (size_t)j * (size_t)a->iters,
                                   sizeof(double) * (size_t)a->iters, hipMemcpyDeviceToHost));
    }
    int rc_all = NIRRT_OK;
    for (int j = 0; j < n_trees; j++) {
        const int i = perm[(size_t)j];
        a->iters_done[i] = done[(size_t)j];
        a->np_used[i] = npu[(size_t)j];
        a->py_used[i] = pyu[(size_t)j];
        if (a->status) a->status[i] = stop[(size_t)j];
        const long long *before = &col[(size_t)j * COLLECT_W], *after = &col[(nt + (size_t)j) * COLLECT_W];
        report_stats(a, i, after, before);
        trees[i]->last_n = after[0];
        if (stop[(size_t)j] == NIRRT_E_CAPACITY) rc_all = NIRRT_E_CAPACITY;
    }
    cleanup();
    return rc_all;   // NIRRT_E_STREAM is reported per tree in status[] (the caller refills and resumes)
#undef HIPCHK_R
}

/* debug: per-phase tick counters (all zero unless built with -DNIRRT_PROFILE) */
extern "C" int nirrt_debug_prof(nirrt_tree *t, int64_t *out24)
{
    if (!t || !out24) return NIRRT_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    TreeDev tmp;
    HIPCHK(hipMemcpy(&tmp, t->dev, sizeof(TreeDev), hipMemcpyDeviceToHost));
    for (int i = 0; i < 24; i++) out24[i] = tmp.prof[i];
    return NIRRT_OK;
}

extern "C" int nirrt_run(nirrt_tree *const *trees, int32_t n_trees, const nirrt_run_args *a)
{
    if (!trees || n_trees <= 0 || !a || a->iters < 0) return NIRRT_E_ARG;
    if (a->struct_size != sizeof(nirrt_run_args)) {
        g_err = "nirrt_run: nirrt_run_args.struct_size does not match this library (NIRRT_ABI_VERSION " + std::to_string(NIRRT_ABI_VERSION) +
                ", sizeof " + std::to_string(sizeof(nirrt_run_args)) + "): recompile the caller against include/nirrt_hip.h";
        return NIRRT_E_ARG;
    }
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
    // (launched and timed on a stream of the calling thread, see run_sampling)
    HIPCHK(hipStreamSynchronize(t0->stream));
    hipStream_t *gs_ = group_streams(t0->device);
    hipStream_t st = gs_ ? gs_[2] : t0->stream;
    std::vector<TreeDev *> ptrs((size_t)n_trees);
    for (int i = 0; i < n_trees; i++) ptrs[(size_t)i] = trees[i]->dev;
    TreeDev **d_ptrs = nullptr;
    double *d_samples = nullptr, *d_trace = nullptr;
    long long *d_done = nullptr;
    size_t sbytes = sizeof(double) * (size_t)n_trees * (size_t)a->iters * D;
    HIPCHK(hipMalloc(&d_ptrs, sizeof(TreeDev *) * (size_t)n_trees));
    long long *d_col = nullptr;
    HIPCHK(hipMalloc(&d_col, sizeof(long long) * 2 * (size_t)n_trees * COLLECT_W));
    if (a->inputs_on_device) d_samples = const_cast<double *>(a->samples);
    else HIPCHK(hipMalloc(&d_samples, sbytes ? sbytes : 8));
    HIPCHK(hipMalloc(&d_done, sizeof(long long) * (size_t)n_trees));
    if (a->cost_trace) HIPCHK(hipMalloc(&d_trace, sizeof(double) * (size_t)n_trees * (size_t)a->iters));
    HIPCHK(hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(TreeDev *) * (size_t)n_trees, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_collect, dim3((n_trees + 255) / 256), dim3(256), 0, st, (TreeDev *const *)d_ptrs, n_trees, d_col);
    if (!a->inputs_on_device) HIPCHK(hipMemcpyAsync(d_samples, a->samples, sbytes, hipMemcpyHostToDevice, st));
    RunDev rd;
    rd.flags = a->flags; rd.pad = 0; rd.iters = a->iters; rd.samples = d_samples; rd.cost_trace = d_trace; rd.iters_done = d_done;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
    long long n_hi = 0;
    for (int i = 0; i < n_trees; i++) n_hi = std::max(n_hi, trees[i]->last_n);
    LAUNCH_V(run_variant(n_trees, a->flags, n_hi, a->iters), D, k_run_replay, n_trees, st, (TreeDev *const *)d_ptrs, rd);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (a->kernel_ms) *a->kernel_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    hipLaunchKernelGGL(k_collect, dim3((n_trees + 255) / 256), dim3(256), 0, st, (TreeDev *const *)d_ptrs, n_trees, d_col + (size_t)n_trees * COLLECT_W);
    std::vector<long long> col(2 * (size_t)n_trees * COLLECT_W);
    HIPCHK(hipMemcpyAsync(col.data(), d_col, sizeof(long long) * col.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d_col);
    std::vector<long long> done((size_t)n_trees);
    HIPCHK(hipMemcpy(done.data(), d_done, sizeof(long long) * (size_t)n_trees, hipMemcpyDeviceToHost));
    if (a->cost_trace)
        HIPCHK(hipMemcpy(a->cost_trace, d_trace, sizeof(double) * (size_t)n_trees * (size_t)a->iters, hipMemcpyDeviceToHost));
    int rc_all = NIRRT_OK;
    for (int i = 0; i < n_trees; i++) {
        if (a->iters_done) a->iters_done[i] = done[(size_t)i];
        if (a->np_used) a->np_used[i] = 0;
        if (a->py_used) a->py_used[i] = 0;
        const long long *before = &col[(size_t)i * COLLECT_W], *after = &col[((size_t)n_trees + (size_t)i) * COLLECT_W];
        if (a->status) a->status[i] = (int)after[1];
        trees[i]->last_n = after[0];
        report_stats(a, i, after, before);
        if (after[1]) rc_all = (int)after[1];
    }
    (void)hipFree(d_ptrs);
    if (!a->inputs_on_device) (void)hipFree(d_samples);
    (void)hipFree(d_done);
    if (d_trace) (void)hipFree(d_trace);
    return rc_all;
}
