// nirrt_device.hpp — gfx950 device code of the RRT*/IRRT* inner loop.
//
// Execution model: ONE workgroup of NT threads (NT/64 wave64s) owns ONE tree; a launch runs many
// trees (one per workgroup, several workgroups per CU) so that one tree's dependent-latency phases
// (parent-chain walks, barriers) overlap other trees' work.
//
//   * nearest / Near.  ONE fused visit per iteration answers find_near_neighbors(node_new), choose_parent's argmin
//     and the NEXT iteration's nearest_neighbor query.  Vertices are kept ordered by cell of a uniform grid
//     (float64 SoA mirror g_x / g_cost / g_idx, rebuilt by a counting sort every GRID_REBUILD_EVERY insertions)
//     followed by the not yet ordered tail, which is read straight from the per-vertex records; a query streams
//     the rows of cells that meet its ball + the tail.  Every visited vertex is decided in float64: squared
//     distance against a 2^-48 guard band, the reference's own distance formula (glibc hypot / sqrt) inside the
//     band and for every member.  Members are not stored as a list: the visit keeps the running
//     argmin of cost + dist for choose_parent and leaves (index, cost - dist) pairs in an LDS stash (spilling to
//     HBM only beyond the stash capacity), from which rewire later picks, in ascending index order, the few members
//     that can pass `cost(j) > cost(new) + dist`.  Only members whose margin cost - dist exceeds the straight
//     distance root -> new (a lower bound of cost(new)) are kept: the others cannot pass that test.
//   * cost(v) is the reference's leaf->root sum (math.hypot per edge, added in that order).  It is kept
//     EXACT in a per-vertex cache: a walk chases one 48-byte record per FOUR hops, and a re-parented
//     vertex has its whole subtree (child / sibling links) re-walked right away, so every cost the loop
//     compares is the value the reference's un-cached cost() would return.
//
// Arithmetic: float64, compiled with -ffp-contract=off; per-call-site formulas of SURVEY.md App. A:
//   np.hypot -> hypot_np()   math.hypot -> hypot_py<D>()   np.linalg.norm(axis) -> norm_axis<D>()
//   np.linalg.norm 1-D -> norm_1d<D>()   np.dot -> dot_blas<D>()
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/nirrt_hip.h"

// Tree arrays live in HBM: the device code addresses them through address-space-1 ("global") pointers, so that the
// compiler emits global_load / global_store (vmcnt only) instead of flat_* instructions - a flat access also counts on
// lgkmcnt, i.e. every wait for an LDS read would wait for the loads in flight as well.  Host code (and the descriptor as
// stored in HBM) uses plain pointers of the same size: TreeHotH is the storage type, TreeHot the device's view of it.
#if defined(__HIP_DEVICE_COMPILE__)
#define GAS __attribute__((address_space(1)))
#else
#define GAS
#endif
template <typename T> struct gp_plain { typedef T *type; };
template <typename T> struct gp_global { typedef GAS T *type; };

// whole-record loads / stores through a global pointer (16-byte pieces; the records are 16 / 32 / 48 bytes)
typedef unsigned nirrt_v4u __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T ldg(const GAS T *p)
{
    static_assert(sizeof(T) % 16 == 0, "records are multiples of 16 bytes");
    T out;
    const GAS nirrt_v4u *g = (const GAS nirrt_v4u *)p;
    nirrt_v4u *o = (nirrt_v4u *)&out;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) o[i] = g[i];
    return out;
}
template <typename T>
__device__ __forceinline__ void stg(GAS T *p, const T &v)
{
    static_assert(sizeof(T) % 16 == 0, "records are multiples of 16 bytes");
    GAS nirrt_v4u *g = (GAS nirrt_v4u *)p;
    const nirrt_v4u *o = (const nirrt_v4u *)&v;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) g[i] = o[i];
}

struct Hop4;
struct Topo;
#ifndef NIRRT_WAVES_PER_EU
#define NIRRT_WAVES_PER_EU 3   // second __launch_bounds__ argument of the persistent kernels (register budget 512 / this).
// 3 waves per SIMD = 168 VGPRs = 12 one-wave trees per CU (3072 per GPU).  At 4 (128 VGPRs, 4096 trees resident) the four phase
// functions spilled 100+ registers per iteration: the launch's WRITE_SIZE was the scratch frames, 27 KB per iteration in 2D and
// 35.8 KB in 3D - exactly (224 + 176 + 96) / (256 + 192 + 112) bytes per lane x 64 - against ~1.2 KB of tree updates; frames at
// 168 VGPRs: 48 + 32 + 16 bytes.  Per-tree time 4.28 -> 2.70 s (IRRT* 2D), launch 11.7 -> 10.0 s; RRT* 2D 54 -> 63.5 M it/s,
// 3D 45 -> 58.7 M it/s.  (2 waves per SIMD, no spills at all: 2.37 s per tree but only 2048 slots - launch 11.6 s.)
#endif
// Out-of-line device functions (NIRRT_FN) get their register budget from the kernels that reach them: the compiler gives a
// callee the union of its callers' waves-per-EU ranges, so EVERY kernel of the library carries the same second launch
// bound - one kernel without it would lift the limit for the shared callees, and with them for every kernel that calls them
// `static`: internal linkage makes the definition exact, which is what lets the inter-procedural register allocation use
// the callee's real clobber set at the call sites (a template's linkonce_odr body "may be replaced" and counts as unknown)
#define NIRRT_FN static __noinline__

#define MAX_OBS NIRRT_MAX_OBSTACLES
#ifndef LDS_POOL
#define LDS_POOL 800     // 8-byte LDS slots shared by the obstacle tables (4 per round + 6 per box obstacle) and the Near stash
#endif
#define OB_POOL NIRRT_OBSTACLE_POOL   // slots the obstacle tables may take (the rest, >= 256 entries, is the stash)
#define SCAN_PAD 256     // extra elements allocated behind every per-vertex array (vector loads may overrun n)
#define WALK_R (NHOP == 8 ? 2 : 4)   // parent chains chased concurrently per lane (their records live in registers)
#ifndef CHAIN_MAX
#define CHAIN_MAX 96     // LDS slots for the new->root edge-length sequence (deeper chains continue in HBM, t.chain_g)
#endif
#define GRID_MIN_VERTICES 2048   // smaller trees are visited whole (everything is "tail")
#define GRID_REBUILD_EVERY 2048  // vertices appended behind the cell-ordered part before it is rebuilt (round 6: 2048, coarse level every
                                 // 128 - a full rebuild re-sorts the whole tree, and with the other traffic of an iteration cut it showed:
                                 // 1024 / 2048 / 4096: 58.3 / 59.3 / 57.0 M it/s on the default line, RRT* 2D 77.5 / 80.4 at 1024 / 2048)
#define GRID_RG_MAX 64           // rows of cells per query (<= the smallest workgroup: one row per thread); larger boxes fall
                                 // back to visiting the whole tree
#ifndef NEAR_STASH
#define NEAR_STASH LDS_POOL      // upper limit of Near members whose (index, bound) pair stays in LDS (the pool slots the
#endif                           // obstacle tables leave free); the rest spills to nr_idx / nr_m.  Test builds set it to 8.
#ifndef GRID_U_WIDE
#define GRID_U_WIDE 2            // ... of the 256-lane kernels (one or a few heavy trees).  Round 5: 3 and 4 spill inside the visit and lose
                                 // everywhere the 256-lane kernels run (1000-problem set 18.2 / 16.6 / 15.4 M it/s, one tree alone
                                 // 1.66 / 1.73 / 1.82 s, 3D IRRT* 11.0 / 10.9 / 10.3)
#endif
#ifndef GRID_U
#define GRID_U 2                 // slots per lane and trip of a grid visit (measured: 2 beats 4 by 12 %, 8 spills).  Round 4: TWO trips
                                 // requested ahead of the one being evaluated (three register sets taking turns, loop unrolled by
                                 // three, no spills in the loop) ran 16 % SLOWER (43.7 vs 52.0 M it/s): more loads in flight per
                                 // wave only lengthen the queues - the visit is bound by what the memory system delivers for
                                 // scattered 32-byte records (3.8 TB/s at the HBM), not by the latency one wave sees
#endif
#define REBUILD_U 4               // vertices per lane and trip of an index rebuild
#ifndef LIST_U
#define LIST_U 8                 // solution / goal-candidate list entries per lane and trip
#endif
#define RG_BITS 11264            // flat offsets of a range list that the start-bit map covers (1408 B of LDS: what 12 trees per CU leave -
                                 // the module's LDS, this struct + 256 B of other kernels' variables, must stay within 12800 B)
#define GRID_N 1u                // range serves the Near query
#define GRID_Q 2u                // range serves the nearest query

// per-tree counters of one launch (nirrt_run_args.stats; accumulated in LDS, flushed when the kernel ends)
#define ST_VISITED 0     // slots visited by the fused nearest / Near passes
#define ST_VISIT_B 1     // bytes those visits read (28 / 36 B per cell-ordered slot, 32 B per tail record)
#define ST_MEMBERS 2     // Near members (collision-free, not the new vertex itself)
#define ST_SPILLED 3     // members that did not fit the LDS stash
#define ST_HOPREC 4      // 48-byte chain records read by cost walks (cost(new) + subtree re-costing)
#define ST_ROUNDS 5      // rewire candidates examined (one 32-byte record each)
#define ST_REWIRED 6     // vertices re-parented by rewire
#define ST_RECOST 7      // vertices re-costed below re-parented vertices
#define ST_LISTSCAN 8    // goal-candidate list entries re-evaluated (12 B + one 32-byte record each)
#define ST_SOLSCAN 21    // solution list entries re-evaluated (8 B each: the cached cost + Line(v, goal) of round 6)
#define ST_INSERTED 9    // vertices appended
#define ST_REBUILT 10    // vertices passed through index rebuilds (32 B read + 28 / 36 B + 4 B written each)
#define ST_REVISITS 11   // additional visits of a widened nearest box
#define ST_BRUTE 12      // queries answered by visiting the whole tree (no index yet / box too large / tie)
#define ST_ITERS 13
#define ST_T0 14         // wall_clock64 (100 MHz) when the tree's loop started / ended
#define ST_T1 15
#define ST_CBEST 17      // bit pattern of the best cost when a sampling loop ended (absolute, like T0 / T1; what NIRRT* compared with ratio * c_update)
#define ST_RROUNDS 18    // rewire rounds that re-parented something (batched path)
#define ST_RSEQ 19       // vertices re-parented one at a time (candidate list did not fit / small-limits build)
#define ST_BUSY 20       // device ticks (100 MHz) the tree's loop was running (= T1 - T0 unless the launch was time-sliced)
#define ST_ALG 16        // vertices the REFERENCE algorithm scans for the same iterations: n per nearest_neighbor + n per find_near_neighbors
#define NSTAT NIRRT_N_STATS

// optional per-phase cycle accounting (build with -DNIRRT_PROFILE; scripts/perf_phases.py reads prof[])
#ifdef NIRRT_PROFILE
#define PROF_DECL long long prof_t0 = wall_clock64();
#define PROF(slot)                                                     \
    do {                                                               \
        long long now_ = wall_clock64();                               \
        if (threadIdx.x == 0) s.tree_g->prof[slot] += now_ - prof_t0; \
        prof_t0 = now_;                                                \
    } while (0)
#else
#define PROF_DECL
#define PROF(slot)
#endif

// Values that are identical in every lane by construction (read from LDS / the tree descriptor, results of the
// workgroup-wide reductions) are moved to scalar registers explicitly: control flow that depends on them compiles to
// scalar branches instead of exec-masked regions, and they stop occupying vector registers.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long uni(long long b)
{
    int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}
template <class T>
__device__ __forceinline__ T *uni(T *v)
{
    long long b = (long long)v;
    int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return (T *)(((long long)hi << 32) | (unsigned)lo);
}

// Thread index inside the workgroup.  One-wave workgroups (the slim instantiation: most of the library's work) take it from the
// lane counter - two VALU instructions wherever it is needed - instead of the work-item id register: a value every function of
// the loop body needs would have to be kept alive (saved and reloaded) across each of the loop's calls, and with no function
// asking for it the kernels do not even have to pass it down.
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int NT>
__device__ __forceinline__ int tidx()
{
    if (NT == 64) return lane_id();
    return (int)threadIdx.x;
}

// ------------------------------------------------------------------------------------------------
// per-tree state in HBM
// ------------------------------------------------------------------------------------------------
// NHOP hops of the parent chain in one record: a cost() walk needs one memory round trip per NHOP edges.
// e[j] / a[j] = length / upper end of the j-th edge above the vertex; beyond the root the entries are (0, 0), and
// adding 0.0 to the (non-negative) running sum leaves it bit-identical, so a walk may run over the end of the chain.
// Round 5: 8 hops (96 bytes) instead of 4 - cost(new), the one long dependent pointer chase of an iteration (~40 edges at
// 50 000 vertices), is 5 round trips instead of 10.
#ifndef NHOP
#define NHOP 8
#endif
struct __attribute__((aligned(16))) Hop4 {
    double e[NHOP];
    int a[NHOP];
};

// the hop part (first 12 * NHOP bytes) of a vertex's record
__device__ __forceinline__ Hop4 ld_hop(const GAS Topo *p) { return ldg(reinterpret_cast<const GAS Hop4 *>(p)); }
__device__ __forceinline__ void st_hop(GAS Topo *p, const Hop4 &h) { stg(reinterpret_cast<GAS Hop4 *>(p), h); }

__device__ __forceinline__ Hop4 hop_shift(const Hop4 &p, double e0, int a0)
{
    Hop4 r;
    r.e[0] = e0; r.a[0] = a0;
#pragma unroll
    for (int j = 1; j < NHOP; j++) { r.e[j] = p.e[j - 1]; r.a[j] = p.a[j - 1]; }
    return r;
}

// Everything about a vertex's place in the tree in ONE record (one 128-byte line with NHOP = 8): the hops above it (a[0] = its
// parent, e[0] = math.hypot(v - v_parent), the term cost() adds for this vertex; 0 for the root), its child-list links and
// its list-membership flags.  A cost walk, a re-parenting and a subtree traversal each touch one record per vertex.
struct __attribute__((aligned(NHOP == 8 ? 128 : 64))) Topo {
    double e[NHOP];
    int a[NHOP];
    int fc, ns, ps;   // first child, next / previous sibling (-1 = none); the root is nobody's child
    int flags;        // bit 0 = in sol[], bit 1 = in gc_idx[] (a re-costed listed vertex invalidates the cached best)
    // (round 6) what used to be looked up in arrays of their own - a 4-byte read that cost a 128-byte line of HBM traffic each -
    // rides in the record's last 16 bytes, which every reader of the record has fetched anyway:
    int slot;         // slot of the vertex in the grid index (valid for vertices below g_ns2; slot = index above)
    int sol_q;        // first position of the vertex in sol[] (valid if flags & 1): where its cached cost + Line(v, goal) lives
    int pad_[2];
};
static_assert(sizeof(Topo) == (NHOP == 8 ? 128 : 64) || NHOP != 8, "the tree record is one 128-byte line");
// the links + flags of a vertex (the 16 bytes behind the hop part) in one load
struct __attribute__((aligned(16))) TopoLinks { int fc, ns, ps, flags; };
__device__ __forceinline__ TopoLinks ld_links(const GAS Topo *p)
{
    return ldg(reinterpret_cast<const GAS TopoLinks *>(reinterpret_cast<const GAS char *>(p) + sizeof(Hop4)));
}
__device__ __forceinline__ void st_links(GAS Topo *p, const TopoLinks &l)
{
    stg(reinterpret_cast<GAS TopoLinks *>(reinterpret_cast<GAS char *>(p) + sizeof(Hop4)), l);
}
// a whole record without its padding: hop part + links
__device__ __forceinline__ void ld_topo(const GAS Topo *p, Hop4 &h, TopoLinks &l) { h = ld_hop(p); l = ld_links(p); }

// random-access twin of a vertex: one 32-byte record (half a 64-byte sector) holds everything the O(k) phases
// need about a Near candidate - coordinates and the exact cost(v) - so a candidate costs ONE sector read
struct __attribute__((aligned(32))) VRec {
    double x, y, z;
    double cost;   // exact cost(v) of the CURRENT tree (see walk_chains / wg_recost_subtree)
};

// One slot of the grid index (see "uniform-grid index"): what a query needs about a vertex - coordinates, exact cost(v) (kept in
// step with vrec[v].cost) and the vertex index - PACKED (round 6): 28 bytes in 2D {x, y, cost, id}, 36 in 3D {x, y, cost, z, id}.
// The visit streams rows of these records and its bytes are what an iteration's memory traffic mostly is: rounds 3 - 5 kept a
// 32-byte record (4 bytes of padding in 2D) and, in 3D, the index in an array of its own - a second stream with a 128-byte line of
// its own for every row of a handful of slots.  Records are only 4-byte aligned; the loads below say so.
#ifdef NIRRT_SLOT_PAD   // A/B builds only: the record stride of rounds 3 - 5 (32 bytes in 2D) / a 40-byte 3D record
template <int D> struct SlotBytes { static constexpr int value = D == 2 ? 32 : 40; };
#else
template <int D> struct SlotBytes { static constexpr int value = D == 2 ? 28 : 36; };
#endif
#define SLOT_COST_OFF 16   // cost sits at the same place in both layouts

// The part of a tree descriptor the loop body touches: copied into LDS when a kernel starts (hot_enter) and written back when
// it ends (hot_leave), so that a pointer or a counter of the tree costs an LDS read instead of a dependent scalar load from HBM
// (12 - 16 trees per CU x ~700 B of descriptor do not live in the scalar cache).
template <template <typename> class P>
struct TreeHotT {
    typename P<Topo>::type topo;     // topo[cap]: parent chain (4 hops), child-list links, flags
    typename P<VRec>::type vrec;     // vrec[cap]: coordinates + exact cost by vertex index (random access, download)
    typename P<int>::type bfs_q;     // scratch queue for subtree traversals
    typename P<int>::type bfs_fc;    // ... first child of each queue entry (-2: not known, read the record)
    // rewire: tie_stamp[v] == rw_stamp <=> v is a Near member of THIS pass that is collision-free and fails the reference's test
    // by less than the list threshold (a "near-tie").  Such members stay off the candidate list; if an earlier re-parenting of
    // the pass re-costs one of them, the re-costing finds the stamp and puts it on the list for an exact re-test.
    typename P<int>::type tie_stamp;
    typename P<double>::type chain_g;   // edge lengths of the chain new -> root beyond the first CHAIN_MAX (which live in LDS)
    int cap;
    int n;          // num_vertices
    int dim;
    int status;     // sticky NIRRT_E_* code
    // continuation of the LDS Near stash (members beyond NEAR_STASH; the nirrt_near primitive puts all of them here)
    typename P<int>::type nr_idx;
    typename P<double>::type nr_m;
    // IRRT*: path_solutions (goal-parent indices, duplicates allowed) + cached costs
    typename P<int>::type sol;
    typename P<double>::type sol_line;   // Line(v, goal) of each solution vertex (static)
    // cost(sol[q]) + sol_line[q] as of the vertex's last re-costing (+inf at the later positions of a vertex listed twice: the
    // first one has the same value and the lower position, so it is the one np.argmin finds).  find_best_path_solution's scan reads
    // this array front to back instead of gathering the vertices' records - one 128-byte line of HBM traffic per list entry before
    typename P<double>::type sol_val;
    int n_sol;
    int cap_sol;
    int sol_dirty;   // some parent changed since sol_best was computed
    int sol_best;    // argmin position in sol[] (first minimum), -1 if none
    double sol_best_cost;
    // RRT*: vertices within step_len of the goal (ascending index), distance, segment test result
    typename P<int>::type gc_idx;
    typename P<double>::type gc_dist;
    typename P<unsigned char>::type gc_col;
    int n_gc;
    int gc_dirty;
    int gc_best;
    int pad1;
    double gc_best_cost;
    // Near radius r(n), tabulated on the host with glibc (rrt_star_2d.py:133, rrt_star_3d.py:134)
    typename P<const double>::type near_r;
    // problem constants
    double start[3], goal[3];
    double step_len, clearance;
    double lo[3], hi[3];
    int n_round, n_box;
    // informed sampling constants (IRRT*.init, irrt_star_2d.py:35-40 / irrt_star_3d.py:32-36)
    double c_min;
    double x_center[3];
    double CL_C[9];          // rotation-to-world matrix C, row-major 3x3
    // NIRRT* point-cloud guidance (nirrt_star_png_2d.py:99-130): predicted path points + policy scalars
    typename P<const double>::type pc;        // (pc_n, dim) row-major
    int pc_n;
    int pad2;
    double pc_rate;          // pc_sample_rate
    double pc_ratio;         // pc_update_cost_ratio
    double c_update;         // best cost at the last cloud refresh (inf before the first solution)
    // uniform-grid index (see "uniform-grid index" below): one 32-byte record per SLOT - slots [0, g_ns) hold vertices
    // [0, g_ns) ordered by cell, slots [g_ns, n) the vertices appended since (slot i = vertex i): a query streams slot ranges only
    typename P<char>::type g_rec;             // packed slot records (SlotBytes<D> each): coordinates, exact cost(v), vertex index
    typename P<int>::type g_start;            // g_start[c] .. g_start[c+1]: slots of cell c inside [0, g_ns); g_ncell + 1 entries
    typename P<int>::type g_cnt;              // rebuild scratch: per-cell counters
    typename P<int>::type g_rank;             // rebuild scratch: rank of vertex i inside its cell
    int g_ns;                // vertices covered by the cell-ordered part (0: index not built yet)
    int g_G;                 // cells per axis
    int g_ncell;             // g_G ^ dim
    int g_min;               // smallest tree that gets indexed (GRID_MIN_VERTICES; env NIRRT_GRID_MIN)
    int g_every;             // rebuild interval in vertices (GRID_REBUILD_EVERY; env NIRRT_GRID_REBUILD)
    int rw_stamp;            // number of the current rewire pass (see tie_stamp)
    double g_rho;            // running estimate of the nearest-vertex distance of the samples (first box of a nearest query)
    double g_inv_h[3];       // cells per unit length, per axis
    double g_margin[3];      // slack added to every query box
    // second, coarse level over the vertices appended since the last (full) rebuild: slots [g_ns, g_ns2) hold vertices
    // [g_ns, g_ns2) ordered by cell of a G2^D grid (re-sorted every g_every2 insertions - a few hundred vertices), so a query
    // reads a few coarse cells of them instead of all; only slots [g_ns2, n) are visited unconditionally
    typename P<int>::type g_start2;   // g_start2[c] .. g_start2[c+1]: slots (absolute) of coarse cell c; g_ncell2 + 1 entries
    int g_ns2;
    int g_G2;
    int g_ncell2;
    int g_every2;
    double g_inv_h2[3];
    double g_margin2[3];
    double g_h[3], g_h2[3];   // cell sizes of both levels (1 / g_inv_h: the host's division is the device's, both correctly rounded)
    // g_start is stored TILED when g_G is a power of two >= 8 (g_lgG = its log2, else 0 = plain row-major): the words of 4 rows x 8
    // columns of cells share one 128-byte line.  A query asks for two words per row of its box (first cell / cell behind the last);
    // row-major, the ~10 rows of a Near box touched 10 - 20 lines of HBM for 20 words, tiled they touch 5 - 6 (grid_word below)
    int g_lgG, pad3;
};
using TreeHotH = TreeHotT<gp_plain>;    // as stored in HBM and as the host fills it in
using TreeHot = TreeHotT<gp_global>;    // the device code's view (same layout)

// The tree's OWN generators, resident in its arena: stream 0 = numpy's legacy RandomState, stream 1 = CPython's random.Random
// (both MT19937; the reference draws from the process-global ones inside the loop: rrt_base_2d.py:46-52, irrt_star_2d.py:121-151).
// A stream is the endless sequence of the generator's 32-bit outputs; pos counts outputs from the start of block 0 (the state
// nirrt_set_generators handed over), block b = the 624 state words after b twists.  The last two blocks are kept (key[s][b & 1])
// so that a speculative draw can be undone across a block boundary; the state as get_state() shows it is (key of the block
// holding output pos - 1, pos inside it in 1..624).
#define MT_N 624
struct MtGen {
    long long pos[2];
    int gen[2];              // highest block produced so far
    int pad[2];
    unsigned key[2][2][MT_N];
};

// the whole descriptor in HBM: hot part first, then what only kernel prologues / epilogues and the host touch
struct TreeDev : TreeHotH {
    long long stat[NSTAT];   // counters since creation / reset (ST_*); a launch reports the difference
    double rnd[MAX_OBS][4];  // cx, cy, cz, r
    double box[MAX_OBS][6];  // x, y, z, w, h, d
    long long prof[24];      // NIRRT_PROFILE: wall_clock64 ticks (100 MHz) per phase
    MtGen *mt;               // the tree's generators (inside its arena)
    // creation (k_init mode 2): the Near radii are tabulated on the device from f(n) (host libm, one table per capacity) and gamma
    const double *near_f;
    double gamma;
    TreeDev **self_slot;     // one-element device array that is to hold the descriptor's own address
};
static_assert(sizeof(TreeHot) % 8 == 0 && sizeof(TreeHot) == sizeof(TreeHotH), "hot_enter / hot_leave copy 8-byte words");

// ------------------------------------------------------------------------------------------------
// launch arguments of the persistent sampling loops (k_run_sample / k_run_pool)
// ------------------------------------------------------------------------------------------------
struct RunSampleDev {
    unsigned flags;
    int pad;
    long long iters;
    const unsigned *const *np_words;   // nullptr: generator mode - the trees' own MT19937 streams (MtGen) produce the words
    const long long *n_np;
    const unsigned *const *py_words;
    const long long *n_py;
    long long *np_used;
    long long *py_used;
    double *cost_trace;
    long long *iters_done;
    int *stop_code;   // per tree: 0 done, NIRRT_E_STREAM, NIRRT_E_CAPACITY
    const long long *iters_each;   // optional per-tree iteration budgets (<= iters)
    int *park;                     // optional (nirrt_run_args.park_limit): trees of this call that stopped for a cloud refresh so far
    int park_limit, pad2;
};

// time-sliced launches (k_run_pool): the work queue of a launch group
struct PoolDev {
    int n_trees, pad;
    long long quantum;     // iterations per slice
    unsigned *ticket;      // next slice to hand out (ticket i = a slice of tree i mod n_trees)
    int *state;            // per tree: bit 0 = being run, bits 1.. = tickets booked on it while it was being run
    int *round;            // per tree: slices completed
    int *fin;              // per tree: run ended early (nothing left for later slices)
    const int *ahead;      // per tree (optional): never waits for its turn (nirrt_run_args.run_ahead)
};

// What one run of the sampling loop (run_tree: a whole launch, or one time slice of it) works from.  It lives in LDS: the loop
// reads a field when it needs it instead of carrying launch arguments in registers across the calls of the loop body - the
// time-sliced kernel of round 4 kept them in VGPRs that every iteration reloaded from scratch (~30 dwords per lane).
// the loop's own state: one copy per wave (every wave computes the same values and reads back only what it wrote itself, so no
// barrier is needed between a write and the reads that follow a call)
struct RunLoop {
    long long k, n_it;    // iteration of this slice / iterations this slice runs
    long long t_begin;
    double cb;            // best cost on the current tree
    int have_q, spec, stop, ended;
};
struct RunCtx {
    RunLoop loop[4];      // (LDS_NW_MAX waves)
    RunSampleDev a;
    TreeDev *tg;          // the tree's descriptor in HBM
    int *fin;             // time-sliced launches: fin[b] = the tree's run has ended (nullptr: one launch per tree)
    long long k0;         // first iteration of this slice
    long long budget;     // iterations of this slice
    int b;                // launch position of the tree (row of the per-tree outputs)
    int sliced;
};

// ------------------------------------------------------------------------------------------------
// LDS working set of one workgroup
// ------------------------------------------------------------------------------------------------
// Generator state of one tree between draws: stream positions and the 64-word windows (see WordStream in
// nirrt_hip.hip) live in LDS, so the persistent loop carries no sampler registers across the loop-body call.
struct StreamState {
    const GAS unsigned *w;   // word mode: the caller's raw outputs; generator mode: the two key blocks of the stream (MtGen::key[s])
    long long n, pos, base;
    long long pos0;          // position when the kernel started (words used by the launch = pos - pos0)
    int genmode;             // 1: the tree's own MT19937 generator produces the words (n is a per-draw limit, see draw_call)
    int gen;                 // generator mode: highest block produced so far
    int wn;                  // words in the register window (64; fewer at the end of a generator block)
    int pad;
    unsigned buf[64];
};

// ONE LDS object for every kernel of the library (both workgroup sizes): with a single module-wide variable the
// compiler can give it the same address in every kernel, so the non-inlined loop-body functions address it with
// constant offsets (-mllvm -amdgpu-lower-module-lds-strategy=module) instead of looking its offset up per kernel.
#define LDS_NW_MAX 4    // waves of the widest workgroup (256 threads)
struct LdsData {
    TreeHot hot;                  // this workgroup's copy of the descriptor's hot part (see TreeHot)
    TreeDev *tree_g;              // the descriptor in HBM (counters, profile slots, obstacle tables)
    int n_round, n_box;
    int stash_off, stash_cap;     // first pool slot / number of slots of the Near stash
    // round obstacles (cx, cy, cz, r) first, then boxes (x, y, z, w, h, d), then the Near stash: stash_cap doubles
    // cost(member) - dist(member, new) followed by stash_cap vertex indices (12 bytes per member)
    double pool[LDS_POOL];
    double red_val[LDS_NW_MAX];
    double red_val2[LDS_NW_MAX];
    int red_idx[LDS_NW_MAX];
    int wave_tot[LDS_NW_MAX];
    StreamState st[2];            // [0] numpy, [1] python
    double bc_d[8];
    int bc_i[8];
    // edge lengths along the chain new -> root of the current iteration (every vertex re-costed in this
    // iteration hangs below `new`, so its walk ends with exactly this sequence)
    int chain_len;          // edges on the chain new -> root: the first CHAIN_MAX lengths in chainE, the rest in t.chain_g
    double chainE[CHAIN_MAX];
    // grid queries: slot ranges of the cell-ordered mirror to visit (rows of cells)
    int rg_n, hit_cnt, mem_cnt;   // query: ranges listed, members stashed, members seen
    // constants of the samplers (copied from the descriptor once per kernel)
    double k_lo[3], k_hi[3], k_clr, k_cmin, k_xc[3], k_CLC[9];
    Hop4 hop_new;                 // copy of hop[new_idx] of the current iteration (thread 0 reads it when re-parenting)
    int new_next, new_fc;         // thread 0: next sibling of the vertex inserted this iteration / head of its child list
    int ob_n;                     // obstacles whose inflated box meets the Near ball's box
    unsigned char ob_list[2 * MAX_OBS];
    int rg_beg[GRID_RG_MAX + 1], rg_len[GRID_RG_MAX + 1];   // + 1: the appended vertices [g_ns, n)
    unsigned char rg_flag[GRID_RG_MAX + 4];
    // (round 6) where a lane's flat offset falls in the range list, without walking the list: bit f of rg_bits is set iff a range
    // starts at flat offset f; the range of offset f is (number of set bits at or below f) - 1 = one popcount per lane on the word
    // of its wave's 64-offset window (a wave-uniform LDS read) + the starts before the window, which the wave carries along.
    // rg_cb[r] = (start offset of range r | its flag << 16, first slot).  Lists of more than RG_BITS slots keep the walk.
    unsigned long long rg_bits[RG_BITS / 64];
    int rg_cb[GRID_RG_MAX + 1][2] __attribute__((aligned(8)));
    int rg_total, rg_fast;
    struct {                      // arguments / results of wg_query_fn
        double pn[3], q[3], r, floor_m, cand;
        int lazy;                 // collision filter of the Near members deferred to the members that can matter (wg_query_fn)
        int n, want, new_idx, lds_cap, ni, cj;
    } qa;
    struct {                      // state of the loop body handed from phase to phase (it_extend / it_connect / it_book)
        double node_in[3], q_next[3], node_new[3], edge_new, cost_ni, r_query;   // q_next: the next iteration's node_rand (has_next)
        long long alg;
        long long sp_pos[2];      // persistent loops, thread 0: generator positions before the early draw
        nirrt_step_result *res;
        unsigned flags;
        int has_next, host_steer, ni, pref_ni, next_ni, collided, new_idx, n, dup, inserted, dup_parent, dup_ns, dup_ps, dup_fc, k, reparented, n_rewired;
    } it;
    RunCtx run;                   // persistent sampling loops: launch arguments + the slice to run (thread 0 fills it in)
    int n_cand;                   // rewire: members whose stashed margin reaches cost(new) (the stash is compacted to them)
    int cand_listed;              // ... all of them are in the LDS list (else: the spilled part is searched per round)
    long long stat[NSTAT];
};
template <int NT>
using Lds = LdsData;   // the per-instantiation name the device functions use
__shared__ LdsData g_lds;   // the one LDS object of the library (see above); out-of-line helpers address it directly


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

// CPython 3.10 Modules/mathmodule.c vector_norm() on (x, y[, z]); z == 0 contributes exact zeros,
// so the 2-argument call is the same code with z = 0.
NIRRT_FN __device__ double hypot_py3(double x0, double x1, double x2, int nd)
{
    const double T27 = 134217729.0;
    double vec[3] = {fabs(x0), fabs(x1), fabs(x2)};
    double mx = vec[0] > vec[1] ? vec[0] : vec[1];
    if (nd == 3) mx = vec[2] > mx ? vec[2] : mx;
    if (mx == 0.0) return mx;
    int max_e;
    (void)frexp(mx, &max_e);
    double scale = ldexp(1.0, -max_e);
    double x, oldcsum, csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0, t, hi, lo, h;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i < nd) {
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
__device__ __forceinline__ double hypot_py(const double *d)
{
    return hypot_py3(d[0], d[1], D == 3 ? d[D - 1] : 0.0, D);
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

// the reference's distance in the O(n) scans and in choose_parent / rewire / goal scan
template <int D>
__device__ __forceinline__ double dist_scan(const double *d)
{
    if (D == 2) return hypot_np(d[0], d[1]);
    return norm_axis<3>(d);
}

// cheap monotone proxy of dist_scan used to filter: plain sum of squares (relative error < 2^-51)
template <int D>
__device__ __forceinline__ double dist2(const double *d)
{
    double s = d[0] * d[0] + d[1] * d[1];
    if (D == 3) s = s + d[2] * d[2];
    return s;
}
// |dist_scan^2 / dist2 - 1| < 2^-50; decisions whose squared operands differ by more than this
// band are already decided by dist2, anything inside is re-decided with dist_scan itself.
#define BAND_HI (1.0 + 0x1p-48)
#define BAND_LO (1.0 - 0x1p-48)

// ------------------------------------------------------------------------------------------------
// segment / point tests against ONE obstacle (the AABB prefilter of the reference is kept)
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

// obstacle tables in the LDS pool: round obstacles first (4 doubles each), boxes behind them (6 doubles each)
__device__ __forceinline__ const double *ob_rnd(const LdsData &s, int i) { return &s.pool[4 * i]; }
__device__ __forceinline__ const double *ob_box(const LdsData &s, int i) { return &s.pool[4 * s.n_round + 6 * i]; }
// Near stash entry p (p < s.stash_cap): margin in pool[stash_off + p], index in the int array behind the margins
__device__ __forceinline__ int *stash_ids(LdsData &s) { return reinterpret_cast<int *>(&s.pool[s.stash_off + s.stash_cap]); }
__device__ __forceinline__ void stash_put(LdsData &s, int p, int id, double m)
{
    s.pool[s.stash_off + p] = m;
    stash_ids(s)[p] = id;
}

// segment vs obstacle #o of the LDS tables (o < n_round: round, else box)
template <int D, int NT>
__device__ __forceinline__ bool seg_obstacle(const Lds<NT> &s, int o, const double *a, const double *b, double clr)
{
    if (o < s.n_round) {
        if (D == 2) return seg_round_2d(a, b, ob_rnd(s, o), clr);
        return seg_round_3d(a, b, ob_rnd(s, o), clr);
    }
    o -= s.n_round;
    if (D == 2) return seg_box_2d(a, b, ob_box(s, o), clr);
    return seg_box_3d(a, b, ob_box(s, o), clr);
}

// the AABB prefilter alone (same arithmetic as the first lines of the seg_* tests above: c - r - clr, c + r + clr for
// a circle / ball, x - clr, x + w + clr for a rectangle / box)
template <int D, int NT>
__device__ __forceinline__ bool seg_aabb_pass(const Lds<NT> &s, int o, const double *l0, const double *l1)
{
    // l0 / l1 = per-axis min / max of the segment end points
    const double clr = s.k_clr;
    bool pass = true;
    if (o < s.n_round) {
        const double *c = ob_rnd(s, o);
        const double cr = c[3];
#pragma unroll
        for (int k = 0; k < D; k++) pass = pass && (l0[k] <= c[k] + cr + clr) && (l1[k] >= c[k] - cr - clr);
    } else {
        const double *b = ob_box(s, o - s.n_round);
#pragma unroll
        for (int k = 0; k < D; k++) pass = pass && (l0[k] <= b[k] + b[3 + k] + clr) && (l1[k] >= b[k] - clr);
    }
    return pass;
}

// whole segment test by ONE lane (lane-parallel fans over many segments)
template <int D, int NT>
__device__ __forceinline__ bool seg_all(const Lds<NT> &s, const double *a, const double *b, double clr)
{
    int M = s.n_round + s.n_box;
    for (int o = 0; o < M; o++)
        if (seg_obstacle<D, NT>(s, o, a, b, clr)) return true;
    return false;
}

// points_in_circles / points_in_balls: strict <  (collision_check_utils.py:292, _3d.py:299)
// points_in_rectangles / points_in_boxes: inclusive (:254, _3d.py:260)
template <int D, int NT>
__device__ __forceinline__ bool point_in_round(const Lds<NT> &s, int i, const double *p, double clr)
{
    const double *c = ob_rnd(s, i);
    double rc = c[3] + clr;
    double q = (p[0] - c[0]) * (p[0] - c[0]) + (p[1] - c[1]) * (p[1] - c[1]);
    if (D == 3) q = q + (p[2] - c[2]) * (p[2] - c[2]);
    return q < rc * rc;
}
template <int D, int NT>
__device__ __forceinline__ bool point_in_box(const Lds<NT> &s, int i, const double *p, double clr)
{
    const double *b = ob_box(s, i);
    bool in = true;
#pragma unroll
    for (int k = 0; k < D; k++) {
        double mx = b[k] + b[3 + k] + clr, mn = b[k] - clr;
        in = in && (mn <= p[k]) && (p[k] <= mx);
    }
    return in;
}

template <int D, int NT>
__device__ __forceinline__ bool point_in_obs(const Lds<NT> &s, const double *p, double clr)
{
    for (int i = 0; i < s.n_round; i++)
        if (point_in_round<D, NT>(s, i, p, clr)) return true;
    for (int i = 0; i < s.n_box; i++)
        if (point_in_box<D, NT>(s, i, p, clr)) return true;
    return false;
}

// the same test spread over the lanes of ONE wave (every lane active, identical p): lane i takes obstacle i
template <int D, int NT>
__device__ __forceinline__ bool point_in_obs_wave(const Lds<NT> &s, const double *p, double clr)
{
    const int lane = lane_id();
    bool hit = false;
    for (int i = lane; i < s.n_round; i += 64) hit = hit || point_in_round<D, NT>(s, i, p, clr);
    for (int i = lane; i < s.n_box; i += 64) hit = hit || point_in_box<D, NT>(s, i, p, clr);
    return __ballot(hit) != 0ull;
}

// points_in_range: the range as one rectangle tested with clearance = -clearance (:330-351)
template <int D>
__device__ __forceinline__ bool point_in_range(const TreeHot &t, const double *p)
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

template <int D, int NT>
__device__ __forceinline__ bool point_in_range_lds(const Lds<NT> &s, const double *p)
{
    double clr = -s.k_clr;
    bool in = true;
#pragma unroll
    for (int k = 0; k < D; k++) {
        double w = s.k_hi[k] - s.k_lo[k];
        double mx = s.k_lo[k] + w + clr, mn = s.k_lo[k] - clr;
        in = in && (mn <= p[k]) && (p[k] <= mx);
    }
    return in;
}

// ------------------------------------------------------------------------------------------------
// workgroup collectives
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void stage_obstacles(Lds<NT> &s, const TreeDev &t)
{
    int tid = tidx<NT>();
    if (tid == 0) {
        s.n_round = t.n_round; s.n_box = t.n_box;
        s.stash_off = 4 * t.n_round + 6 * t.n_box;
        const int room = ((LDS_POOL - s.stash_off) * 2) / 3;   // 8 + 4 bytes per entry
        s.stash_cap = room < NEAR_STASH ? room : NEAR_STASH;
    }
    if (tid < NSTAT) s.stat[tid] = 0;
    if (tid < 3) { s.k_lo[tid] = t.lo[tid]; s.k_hi[tid] = t.hi[tid]; s.k_xc[tid] = t.x_center[tid]; }
    if (tid < 9) s.k_CLC[tid] = t.CL_C[tid];
    if (tid == 0) { s.k_clr = t.clearance; s.k_cmin = t.c_min; }
    const int nr4 = t.n_round * 4;   // the host refuses worlds whose tables exceed OB_POOL
    for (int i = tid; i < nr4; i += NT) s.pool[i] = t.rnd[i / 4][i % 4];
    for (int i = tid; i < t.n_box * 6; i += NT) s.pool[nr4 + i] = t.box[i / 6][i % 6];
    __syncthreads();
}

// kernel prologue: stage the obstacle tables and the hot part of the descriptor; returns the LDS copy every device function
// works on from here
template <int NT>
__device__ __forceinline__ TreeHot &hot_enter(Lds<NT> &s, TreeDev *tg)
{
    const long long *src = reinterpret_cast<const long long *>(static_cast<const TreeHotH *>(tg));
    long long *dst = reinterpret_cast<long long *>(&s.hot);
    for (int i = tidx<NT>(); i < (int)(sizeof(TreeHot) / 8); i += NT) dst[i] = src[i];
    if (tidx<NT>() == 0) s.tree_g = tg;
    stage_obstacles<NT>(s, *tg);   // ends with a barrier
    return s.hot;
}

// kernel epilogue: hot part back to HBM + this launch's LDS counters added to the descriptor's
template <int NT>
__device__ __forceinline__ void hot_leave(Lds<NT> &s)
{
    __syncthreads();
    TreeDev *tg = s.tree_g;
    const long long *src = reinterpret_cast<const long long *>(&s.hot);
    long long *dst = reinterpret_cast<long long *>(static_cast<TreeHotH *>(tg));
    for (int i = tidx<NT>(); i < (int)(sizeof(TreeHot) / 8); i += NT) dst[i] = src[i];
    if (tidx<NT>() == 0)
        for (int i = 0; i < NSTAT; i++)
            if (i != ST_T0 && i != ST_T1) tg->stat[i] += s.stat[i];
}

// wave64 reductions without LDS traffic: four DPP steps (quad swaps, half-row and row mirrors) leave every 16-lane row
// holding its result, v_readlane pulls the four row results into scalars (the result is wave-uniform)
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = dpp_mov<CTRL>((int)b), hi = dpp_mov<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) { return __int_as_float(dpp_mov<CTRL>(__float_as_int(x))); }
__device__ __forceinline__ int lane_get(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ float lane_get(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ double lane_get(double x, int l)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// (value, index) -> lexicographic minimum over the wave, lowest index on ties
template <typename T>
__device__ __forceinline__ void wave_argmin(T &v, int &idx)
{
#define NIRRT_ARGMIN_STEP(C)                                             \
    {                                                                    \
        const T ov = dpp_mov<C>(v);                                      \
        const int oi = dpp_mov<C>(idx);                                  \
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }       \
    }
    NIRRT_ARGMIN_STEP(0xB1) NIRRT_ARGMIN_STEP(0x4E) NIRRT_ARGMIN_STEP(0x141) NIRRT_ARGMIN_STEP(0x140)
#undef NIRRT_ARGMIN_STEP
    T rv = lane_get(v, 0);
    int ri = lane_get(idx, 0);
#pragma unroll
    for (int row = 1; row < 4; row++) {
        const T ov = lane_get(v, 16 * row);
        const int oi = lane_get(idx, 16 * row);
        if (ov < rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
    }
    v = rv;
    idx = ri;
}

template <typename T>
__device__ __forceinline__ T wave_min(T v)
{
#define NIRRT_MIN_STEP(C) { const T ov = dpp_mov<C>(v); v = ov < v ? ov : v; }
    NIRRT_MIN_STEP(0xB1) NIRRT_MIN_STEP(0x4E) NIRRT_MIN_STEP(0x141) NIRRT_MIN_STEP(0x140)
#undef NIRRT_MIN_STEP
    T rv = lane_get(v, 0);
#pragma unroll
    for (int row = 1; row < 4; row++) {
        const T ov = lane_get(v, 16 * row);
        rv = ov < rv ? ov : rv;
    }
    return rv;
}

// lexicographic (value, index) minimum over the workgroup; every thread gets the result.
// Ties keep the LOWEST index (np.argmin).  Threads with nothing pass idx = INT_MAX, v = +inf.
template <int NT>
__device__ __forceinline__ void block_argmin(Lds<NT> &s, double &v, int &idx)
{
    wave_argmin(v, idx);
    int lane = lane_id(), w = tidx<NT>() >> 6;
    __syncthreads();  // protect red_* reuse
    if (lane == 0) { s.red_val[w] = v; s.red_idx[w] = idx; }
    __syncthreads();
    v = s.red_val[0];
    idx = s.red_idx[0];
#pragma unroll
    for (int i = 1; i < NT / 64; i++) {
        double ov = s.red_val[i];
        int oi = s.red_idx[i];
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// minimum index over the workgroup (INT_MAX if none)
template <int NT>
__device__ __forceinline__ int block_min_int(Lds<NT> &s, int v)
{
    v = wave_min(v);
    int lane = lane_id(), w = tidx<NT>() >> 6;
    __syncthreads();
    if (lane == 0) s.red_idx[w] = v;
    __syncthreads();
    v = s.red_idx[0];
#pragma unroll
    for (int i = 1; i < NT / 64; i++) v = s.red_idx[i] < v ? s.red_idx[i] : v;
    return v;
}

// (the library's reduction reads blockDim for the number of waves, so every function on the call path carries the implicit
// kernel arguments and the workgroup ids in saved scalar registers; a hand-written ballot + LDS version drops those and measures
// 2.4 % SLOWER on the default bench - 51.0 vs 52.2 M it/s, same box, twice - so the library's version stays)
template <int NT>
__device__ __forceinline__ bool block_any(bool p)
{
    if (NT == 64) return __ballot(p) != 0ull;   // one wave: no LDS, no barrier, and no implicit kernel arguments on the call path
    return __syncthreads_or(p ? 1 : 0) != 0;
}

// ordered compaction: threads with keep get their output slot (ascending thread order); returns total
template <int NT>
__device__ __forceinline__ int block_compact(Lds<NT> &s, bool keep, int &pos)
{
    unsigned long long m = __ballot(keep);
    int lane = lane_id(), w = tidx<NT>() >> 6;
    int pre = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s.wave_tot[w] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
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
__device__ __forceinline__ void load_vertex(const TreeHot &t, int i, double *v)
{
#pragma unroll
    for (int k = 0; k < D; k++) v[k] = k == 0 ? t.vrec[i].x : (k == 1 ? t.vrec[i].y : t.vrec[i].z);
}

// slot record helpers
typedef double nirrt_v2d __attribute__((ext_vector_type(2)));
typedef unsigned nirrt_v3u __attribute__((ext_vector_type(3)));
typedef nirrt_v2d nirrt_v2d_a4 __attribute__((aligned(4)));   // (the same vectors at the 4-byte alignment packed records have)
typedef nirrt_v3u nirrt_v3u_a4 __attribute__((aligned(4)));
typedef double nirrt_f64_a4 __attribute__((aligned(4)));
template <int D>
__device__ __forceinline__ GAS char *slot_ptr(const TreeHot &t, int sl)
{
    return (GAS char *)t.g_rec + (long long)sl * SlotBytes<D>::value;
}
template <int D>
__device__ __forceinline__ void slot_store(const TreeHot &t, int sl, const double *v, double cost, int id)
{
    GAS char *p = slot_ptr<D>(t, sl);
    nirrt_v2d xy;
    xy.x = v[0]; xy.y = v[1];
    *(GAS nirrt_v2d_a4 *)p = xy;
    const unsigned long long cb = (unsigned long long)__double_as_longlong(cost);
    if (D == 2) {
        nirrt_v3u b;
        b.x = (unsigned)cb; b.y = (unsigned)(cb >> 32); b.z = (unsigned)id;
        *(GAS nirrt_v3u_a4 *)(p + 16) = b;
    } else {
        nirrt_v2d cz;
        cz.x = cost; cz.y = v[D - 1];
        *(GAS nirrt_v2d_a4 *)(p + 16) = cz;
        *(GAS int *)(p + 32) = id;
    }
}
// the cost of the vertex in slot sl changed (every re-costing writes the vertex record and this)
template <int D>
__device__ __forceinline__ void slot_set_cost(const TreeHot &t, int sl, double cost)
{
    *(GAS nirrt_f64_a4 *)(slot_ptr<D>(t, sl) + SLOT_COST_OFF) = cost;
}
// the fields of slot sl, fetched with loads whose every register is used (x, y: 16 bytes; 2D: cost + index 12 bytes;
// 3D: cost + z 16 bytes and the index word) - a wider load with a dead lane would let the register allocator recycle that
// lane and wait for the load right after issuing it
template <int D>
__device__ __forceinline__ void slot_load(const TreeHot &t, int sl, double &x, double &y, double &z, double &c, int &id)
{
    const GAS char *p = slot_ptr<D>(t, sl);
    const nirrt_v2d xy = *(const GAS nirrt_v2d_a4 *)p;
    x = xy.x; y = xy.y;
    if (D == 2) {
        const nirrt_v3u b = *(const GAS nirrt_v3u_a4 *)(p + 16);
        c = __longlong_as_double(((long long)b.y << 32) | (long long)b.x);
        id = (int)b.z;
        z = 0.;
    } else {
        const nirrt_v2d cz = *(const GAS nirrt_v2d_a4 *)(p + 16);
        c = cz.x; z = cz.y;
        id = *(const GAS int *)(p + 32);
    }
}

// out-of-line forms of the rare, long paths of a visit (one copy each in the code; the visit loop stays compact)
NIRRT_FN __device__ double hypot_np_cold(double x, double y) { return hypot_np(x, y); }
template <int D>
__device__ __forceinline__ double dist_scan_cold(double dx, double dy, double dz)
{
    if (D == 2) return hypot_np_cold(dx, dy);
    return __builtin_sqrt(dx * dx + dy * dy + dz * dz);
}
// exact segment test against obstacle #o of the LDS tables
template <int D>
NIRRT_FN __device__ bool seg_obstacle_cold(int o, double ax, double ay, double az, double bx, double by, double bz)
{
    const double a[3] = {ax, ay, az}, b[3] = {bx, by, bz};
    return seg_obstacle<D, 64>(g_lds, o, a, b, g_lds.k_clr);
}

// exact (reference-formula) nearest scan over every slot; only used when a visit sees a near-tie
template <int D, int NT>
NIRRT_FN __device__ int wg_nearest_exact(int n, double q0, double q1, double q2)
{
    Lds<NT> &s = g_lds;
    const TreeHot &t = g_lds.hot;
    double bd = __builtin_inf();
    int bi = 0x7fffffff;
    for (int sl = tidx<NT>(); sl < n; sl += NT) {
        double gx, gy, gz, gc;
        int id;
        slot_load<D>(t, sl, gx, gy, gz, gc, id);
        double d[3] = {q0 - gx, q1 - gy, D == 3 ? q2 - gz : 0.};
        double h = dist_scan<D>(d);
        if (h < bd || (h == bd && id < bi)) { bd = h; bi = id; }
    }
    block_argmin<NT>(s, bd, bi);
    return bi;
}

// workgroup reduction of the per-lane (smallest d2, its index, second-smallest d2) triples of a nearest visit.
// Returns the winner's index, or -1 when a second vertex lies inside the guard band of the minimum (the caller then
// lets the reference formula decide, wg_nearest_exact); *g1_out = the smallest squared distance seen (inf: nothing).
template <int D, int NT>
__device__ __forceinline__ int wg_nearest_finish(Lds<NT> &s, double m1, int i1, double m2, double *g1_out)
{
    constexpr int NW = NT / 64;
    const int lane = lane_id(), w = tidx<NT>() >> 6;
    double wm = m1;
    int wi = i1;
    wave_argmin(wm, wi);
    double ws = (i1 == wi) ? m2 : m1;  // this lane's best that is NOT the wave winner
    ws = wave_min(ws);
    __syncthreads();
    if (lane == 0) { s.red_val[w] = wm; s.red_idx[w] = wi; s.red_val2[w] = ws; }
    __syncthreads();
    double g1 = s.red_val[0];
    int gi = s.red_idx[0], gw = 0;
#pragma unroll
    for (int i = 1; i < NW; i++) {
        double ov = s.red_val[i];
        int oi = s.red_idx[i];
        if (ov < g1 || (ov == g1 && oi < gi)) { g1 = ov; gi = oi; gw = i; }
    }
    double g2 = __builtin_inf();
#pragma unroll
    for (int i = 0; i < NW; i++) {
        double c = (i == gw) ? s.red_val2[i] : s.red_val[i];
        g2 = c < g2 ? c : g2;
    }
    *g1_out = g1;
    if (g1 == __builtin_inf()) return -1;
    return (g2 <= g1 * BAND_HI) ? -1 : gi;
}

// ------------------------------------------------------------------------------------------------
// uniform-grid index.  The reference scans every vertex for nearest_neighbor and find_near_neighbors; the
// answers only depend on the vertices inside a small ball around the query, so every vertex has a 32-byte SLOT record
// (coordinates, exact cost, index): slots [0, g_ns) hold vertices [0, g_ns) ordered by cell of a G^D grid over the range
// box (rebuilt by a counting sort every GRID_REBUILD_EVERY insertions), slots [g_ns, n) the vertices appended since, in
// insertion order.  A query visits the rows of cells (contiguous slot ranges) that intersect its box plus the appended
// range - ONE loop over a list of slot ranges.  Completeness: cell(x) is monotone in x per axis, so every vertex within
// `rad` of p (per axis) lies in a cell of grid_box(p, rad) (g_margin is slack on top); vertices outside the range box
// fall into border cells, which extend to infinity.  The order inside a cell is whatever the atomics produce; no result
// depends on it (minima are reduced as (value, index) pairs, rewire selects by index).
// ------------------------------------------------------------------------------------------------
// position of cell c's word in g_start (c = row-major cell number, or g_ncell for the end word): see TreeHot::g_lgG
__device__ __forceinline__ int grid_word(const TreeHot &t, int c)
{
    const int lg = t.g_lgG;
    if (lg == 0 || c >= t.g_ncell) return c;
    const int G = 1 << lg, x = c & (G - 1), y = (c >> lg) & (G - 1), slab = c >> (2 * lg);
    return (slab << (2 * lg)) + ((((y >> 2) << (lg - 3)) + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7);
}

template <int L = 0>   // L = 0: the G^D grid of the cell-ordered part; L = 1: the coarse grid of the second level
__device__ __forceinline__ int grid_cell_axis(const TreeHot &t, int k, double x)
{
    const double a = (x - t.lo[k]) * (L == 0 ? t.g_inv_h[k] : t.g_inv_h2[k]);
    const int G = L == 0 ? t.g_G : t.g_G2;
    return a <= 0.0 ? 0 : (a >= (double)G ? G - 1 : (int)a);
}

template <int D, int L = 0>
__device__ __forceinline__ void grid_box(const TreeHot &t, const double *p, double rad, int (&c0)[3], int (&c1)[3])
{
    c0[2] = 0; c1[2] = 0;
#pragma unroll
    for (int k = 0; k < D; k++) {
        const double m = L == 0 ? t.g_margin[k] : t.g_margin2[k];
        c0[k] = grid_cell_axis<L>(t, k, p[k] - rad - m);
        c1[k] = grid_cell_axis<L>(t, k, p[k] + rad + m);
    }
}

__device__ __forceinline__ int grid_rows(const int (&c0)[3], const int (&c1)[3])
{
    return (c1[1] - c0[1] + 1) * (c1[2] - c0[2] + 1);
}

// exclusive prefix sum over the workgroup (thread order); returns the total
template <int NT>
__device__ __forceinline__ int block_excl_scan(Lds<NT> &s, int v, int &off)
{
    const int lane = lane_id(), w = tidx<NT>() >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) s.wave_tot[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        int c = s.wave_tot[i];
        if (i < w) base += c;
        tot += c;
    }
    off = base + inc - v;
    return tot;
}

// counting sort of vertices [0, n) by cell into the slot records (one 32-byte store per vertex + its pos[] entry)
template <int D, int NT>
NIRRT_FN __device__ void wg_grid_rebuild(int n)
{
    Lds<NT> &s = g_lds;
    TreeHot &t = g_lds.hot;
    const int tid = tidx<NT>(), nc = t.g_ncell, G = t.g_G;
    for (int c = tid; c < nc; c += NT) t.g_cnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](const VRec &v) -> int {
        int c = grid_cell_axis(t, 0, v.x) + G * grid_cell_axis(t, 1, v.y);
        if (D == 3) c += G * G * grid_cell_axis(t, 2, v.z);
        return c;
    };
    for (int i0 = tid; i0 < n; i0 += NT * REBUILD_U) {
        VRec v[REBUILD_U];
        int cell[REBUILD_U], rk[REBUILD_U];
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++)
            if (i0 + u * NT < n) v[u] = ldg(&t.vrec[i0 + u * NT]);
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) cell[u] = i0 + u * NT < n ? cell_of(v[u]) : -1;
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++)
            rk[u] = cell[u] >= 0 ? __hip_atomic_fetch_add(&t.g_cnt[cell[u]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++)
            if (cell[u] >= 0) t.g_rank[i0 + u * NT] = rk[u];
    }
    __syncthreads();
    const int per = (nc + NT - 1) / NT, b = tid * per;
    int sum = 0;
    for (int j = 0; j < per; j++)
        if (b + j < nc) sum += t.g_cnt[b + j];
    int run;
    block_excl_scan<NT>(s, sum, run);
    for (int j = 0; j < per; j++) {
        if (b + j < nc) {
            int c = t.g_cnt[b + j];
            t.g_start[grid_word(t, b + j)] = run;
            run += c;
        }
    }
    if (tid == 0) t.g_start[nc] = n;
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += NT * REBUILD_U) {
        VRec v[REBUILD_U];
        int rk[REBUILD_U], st[REBUILD_U];
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) {
            const int i = i0 + u * NT;
            rk[u] = 0;
            if (i < n) { v[u] = ldg(&t.vrec[i]); rk[u] = t.g_rank[i]; }
        }
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) st[u] = i0 + u * NT < n ? t.g_start[grid_word(t, cell_of(v[u]))] : 0;
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) {
            const int i = i0 + u * NT;
            if (i < n) {
                const int sl = st[u] + rk[u];
                const double xyz[3] = {v[u].x, v[u].y, v[u].z};
                slot_store<D>(t, sl, xyz, v[u].cost, i);
                t.topo[i].slot = sl;
            }
        }
    }
    if (tid == 0) { t.g_ns = n; t.g_ns2 = n; s.stat[ST_REBUILT] += n; }
    __syncthreads();
}

// counting sort of the vertices appended since the last full rebuild, [g_ns, n), by coarse cell into slots [g_ns, n)
template <int D, int NT>
NIRRT_FN __device__ void wg_grid_rebuild2(int n)
{
    Lds<NT> &s = g_lds;
    TreeHot &t = g_lds.hot;
    const int tid = tidx<NT>(), nc = t.g_ncell2, G = t.g_G2, v0 = t.g_ns, m = n - v0;
    for (int c = tid; c < nc; c += NT) t.g_cnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](const VRec &v) -> int {
        int c = grid_cell_axis<1>(t, 0, v.x) + G * grid_cell_axis<1>(t, 1, v.y);
        if (D == 3) c += G * G * grid_cell_axis<1>(t, 2, v.z);
        return c;
    };
    for (int j0 = tid; j0 < m; j0 += NT * REBUILD_U) {
        VRec v[REBUILD_U];
        int cell[REBUILD_U], rk[REBUILD_U];
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++)
            if (j0 + u * NT < m) v[u] = ldg(&t.vrec[v0 + j0 + u * NT]);
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) cell[u] = j0 + u * NT < m ? cell_of(v[u]) : -1;
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++)
            rk[u] = cell[u] >= 0 ? __hip_atomic_fetch_add(&t.g_cnt[cell[u]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++)
            if (cell[u] >= 0) t.g_rank[v0 + j0 + u * NT] = rk[u];
    }
    __syncthreads();
    const int per = (nc + NT - 1) / NT, b = tid * per;
    int sum = 0;
    for (int j = 0; j < per; j++)
        if (b + j < nc) sum += t.g_cnt[b + j];
    int run;
    block_excl_scan<NT>(s, sum, run);
    run += v0;
    for (int j = 0; j < per; j++) {
        if (b + j < nc) {
            int c = t.g_cnt[b + j];
            t.g_start2[b + j] = run;
            run += c;
        }
    }
    if (tid == 0) t.g_start2[nc] = n;
    __syncthreads();
    for (int j0 = tid; j0 < m; j0 += NT * REBUILD_U) {
        VRec v[REBUILD_U];
        int rk[REBUILD_U], st[REBUILD_U];
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) {
            const int j = j0 + u * NT;
            rk[u] = 0;
            if (j < m) { v[u] = ldg(&t.vrec[v0 + j]); rk[u] = t.g_rank[v0 + j]; }
        }
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) st[u] = j0 + u * NT < m ? t.g_start2[cell_of(v[u])] : 0;
#pragma unroll
        for (int u = 0; u < REBUILD_U; u++) {
            const int j = j0 + u * NT;
            if (j < m) {
                const int i = v0 + j, sl = st[u] + rk[u];
                const double xyz[3] = {v[u].x, v[u].y, v[u].z};
                slot_store<D>(t, sl, xyz, v[u].cost, i);
                t.topo[i].slot = sl;
            }
        }
    }
    if (tid == 0) { t.g_ns2 = n; s.stat[ST_REBUILT] += m; }
    __syncthreads();
}

// A range list goes to LDS through rg_publish: thread i contributes range i (mine), thread 0 may add one more behind them.
// Barriers inside; every thread calls it.  (Out of line: four call sites, two of them on rare paths.)  Default build (the walk):
// ranges as they are, total by a wave sum.  -DNIRRT_RG_FAST=1: empty ranges are dropped, the kept ones get their start offsets
// (exclusive scan) and - for lists within RG_BITS slots - their start bits.
#ifndef NIRRT_RG_FAST
// 1: the start-bit map below (built because round 5's review asked for the walk out of `fetch`).  Measured on one box, each twice
// (profiles/r06_sweep6 / 8 / 10): the walk wins wherever a workgroup has more than one wave - a publish with the bit map is a
// compaction, a scan and atomics over up to four waves, four times per query - one problem alone 1.88 -> 1.77 s, the 1000-problem
// set 16.2 -> 17.2 M it/s, 256 problems 5.1 -> 5.5 M - and is level or ahead on the one-wave lines (default 55.5 / 55.2 vs 56.1 /
// 56.1, RRT* 3D 76.0 vs 74.7 the other way).  A lane leaves its range once per ~40 slots: the lookup was never the cost.
// What did cost was the publish itself: without compaction and scan (below) the walk gained another 2.6 % on the default line,
// 7.5 % on the 1000-problem set and 8 % for one problem alone (profiles/r06_sweep11_lean_publish.txt).
#define NIRRT_RG_FAST 0
#endif
template <int NT>
NIRRT_FN __device__ void rg_publish(bool mine, int my_beg, int my_len, unsigned my_flag, int extra_beg, int extra_len, unsigned extra_flag)
{
    Lds<NT> &s = g_lds;
    const int tid = tidx<NT>();
#if !NIRRT_RG_FAST
    {
        // the walk needs neither compaction nor start offsets: range i goes to slot i as it is (empty ranges are stepped over), the
        // total is a wave sum - every contributor sits in wave 0 (at most GRID_RG_MAX = 64 rows, thread i holds row i).  Two barriers
        // instead of the six of the compaction + scan below: on the 128- / 256-lane workgroups a barrier is the expensive part
        static_assert(GRID_RG_MAX <= 64, "rows of a range list live in wave 0");
        __syncthreads();      // (the list the visit before read is free)
        const int len = mine && my_len > 0 ? my_len : 0;
        const int R = __popcll(__ballot(mine));      // (wave 0: the contributors are threads 0 .. R - 1)
        if (mine) { s.rg_beg[tid] = my_beg; s.rg_len[tid] = len; s.rg_flag[tid] = (unsigned char)my_flag; }
        if (tid < 64) {
            int sum = len;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
            if (tid == 0) {
                const bool has_extra = extra_len > 0;
                if (has_extra) { s.rg_beg[R] = extra_beg; s.rg_len[R] = extra_len; s.rg_flag[R] = (unsigned char)extra_flag; }
                s.rg_n = R + (has_extra ? 1 : 0); s.rg_total = sum + (has_extra ? extra_len : 0); s.rg_fast = 0;
            }
        }
        __syncthreads();
        return;
    }
#endif
    const bool keep = mine && my_len > 0;
    int pos = 0, cs = 0;
    int R = block_compact<NT>(s, keep, pos);
    int total = block_excl_scan<NT>(s, keep ? my_len : 0, cs);
    const bool has_extra = extra_len > 0;     // (uniform: every thread passes the same extra range)
    const int R_all = R + (has_extra ? 1 : 0), total_all = total + (has_extra ? extra_len : 0);
    const bool fast = NIRRT_RG_FAST && R_all >= 1 && total_all <= RG_BITS;
    if (fast)
        for (int w = tid; w < (total_all + 63) / 64; w += NT) s.rg_bits[w] = 0ull;
    __syncthreads();
    if (keep) {
        s.rg_beg[pos] = my_beg; s.rg_len[pos] = my_len; s.rg_flag[pos] = (unsigned char)my_flag;
        if (fast) {
            s.rg_cb[pos][0] = cs | (int)(my_flag << 16); s.rg_cb[pos][1] = my_beg;
            atomicOr(&s.rg_bits[cs >> 6], 1ull << (cs & 63));
        }
    }
    if (tid == 0) {
        if (has_extra) {
            s.rg_beg[R] = extra_beg; s.rg_len[R] = extra_len; s.rg_flag[R] = (unsigned char)extra_flag;
            if (fast) {
                s.rg_cb[R][0] = total | (int)(extra_flag << 16); s.rg_cb[R][1] = extra_beg;
                atomicOr(&s.rg_bits[total >> 6], 1ull << (total & 63));
            }
        }
        s.rg_n = R_all; s.rg_total = total_all; s.rg_fast = fast ? 1 : 0;
    }
    __syncthreads();
}

// sqrt to a few ulp (hardware reciprocal square root estimate + two coupled Newton steps, no special-case handling):
// enough where a tolerance of 2^-40 follows; about a third of the instructions of the IEEE sequence
__device__ __forceinline__ double sqrt_fast(double v)
{
    if (!(v > 0.)) return 0.;
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}

// result of the Near part of a query
struct NearResult {
    int k;          // members: collision-free vertices within r of node_new other than new_idx
    double cand;    // min over members of cost(j) + dist(j, new)   (inf if none)
    int cj;         // its vertex index, lowest index on ties (np.argmin over the ascending neighbour list)
    int n_stash;    // members on the stash (<= k): those whose margin cost - dist exceeds the caller's floor
};

// one visited slot in registers
struct SlotRegs {
    double x, y, z, c;
    int id;
    unsigned fl;
};

// ONE fused query (find_near_neighbors rrt_star_2d.py:125-144 + choose_parent's argmin :80-90 + the nearest_neighbor
// rrt_base_2d.py:94-107 of another point):
//   pn != nullptr: Near set of pn with radius r on the current tree -> *nr; member (index, bound) pairs go to the
//                  stash: entries [0, lds_cap) in LDS, the rest in t.nr_idx / t.nr_m;
//   q  != nullptr: *ni = argmin_i dist(q, v_i), lowest index on ties (np.argmin).
// Structure: a list of slot ranges in LDS is visited by ONE loop (GRID_U slots per lane and trip, the next trip's records
// requested before the current trip is evaluated); the list is first the appended vertices [g_ns, n) - visited while the
// g_start words of the cell rows are in flight -, then the rows, then (nearest only) whatever a widened box adds.
// The query is a real call with its arguments and results in LDS (s.qa): its loops then get a register allocation of their
// own instead of sharing the file with everything the loop body keeps alive around them (which spilled into those loops).
template <int D, int NT>
NIRRT_FN __device__ void wg_query_fn()
{
    Lds<NT> &s = g_lds;
    const TreeHot &t = g_lds.hot;
    const int tid = tidx<NT>(), lane = tid & 63, G = t.g_G;
    const unsigned long long lt = (1ull << lane) - 1ull;
    __syncthreads();   // s.qa is in place
    if (tid == 0) { s.ob_n = 0; s.hit_cnt = 0; s.mem_cnt = 0; }
    const int ns = uni(t.g_ns);
    const int n = uni(s.qa.n);
    const bool wantN = (uni(s.qa.want) & 1) != 0, wantQ = (uni(s.qa.want) & 2) != 0;
    // everything the loop compares with is the same in every lane: scalar registers
    const double r = uni(s.qa.r), floor_m = uni(s.qa.floor_m);
    const int new_idx = uni(s.qa.new_idx), lds_cap = uni(s.qa.lds_cap);
    const bool lazy = uni(s.qa.lazy) != 0;
    const double r2 = r * r, r2lo = r2 * BAND_LO, r2hi = r2 * BAND_HI;
    const double pnx = uni(s.qa.pn[0]), pny = uni(s.qa.pn[1]), pnz = D == 3 ? uni(s.qa.pn[2]) : 0.;
    const double qx = uni(s.qa.q[0]), qy = uni(s.qa.q[1]), qz = D == 3 ? uni(s.qa.q[2]) : 0.;
    const double pnv[3] = {pnx, pny, pnz}, qv[3] = {qx, qy, qz};
    long long visited = 0;
    int revisits = 0, brutes = 0;
    PROF_DECL
    __syncthreads();
    // obstacles whose inflated box meets the box of the Near ball: a segment new -> v_j (|v_j - new| <= r per axis)
    // can only pass the AABB prefilter of those.  Same comparisons as seg_aabb_pass, on a superset of every segment's box.
    if (wantN) {
        const int M = s.n_round + s.n_box;
        for (int o = tid; o < M; o += NT) {
            const double rb = r * (1.0 + 1e-9);
            double l0[3], l1[3];
#pragma unroll
            for (int c = 0; c < D; c++) { l0[c] = pnv[c] - rb; l1[c] = pnv[c] + rb; }
            if (seg_aabb_pass<D, NT>(s, o, l0, l1)) s.ob_list[atomicAdd(&s.ob_n, 1)] = (unsigned char)o;
        }
    }
    // row `row` of box (c0, c1) of grid level L -> slot range.  ball != nullptr: only the cells of the row that the ball
    // (ball, rad) can reach are kept (the corner cells of the box are dropped)
    auto row_range = [&](auto lvl, const int (&c0)[3], const int (&c1)[3], int row, const double *ball, double rad, int &rb, int &rl) {
        constexpr int L = decltype(lvl)::value;
        const int GL = L == 0 ? G : t.g_G2;
        const int ny_ = c1[1] - c0[1] + 1;
        const int cy = c0[1] + row % ny_, cz = c0[2] + row / ny_;
        int x0 = c0[0], x1 = c1[0];
        bool empty = false;
        if (ball) {
            // distance from the ball centre to the row's slab in y (and z); border cells extend to infinity
            const double m0 = L == 0 ? t.g_margin[0] : t.g_margin2[0];
            double rem = (rad + m0) * (rad + m0);
#pragma unroll
            for (int k = 1; k < D; k++) {
                const int ck = k == 1 ? cy : cz;
                const double h = L == 0 ? t.g_h[k] : t.g_h2[k];
                const double a = ck == 0 ? -__builtin_inf() : t.lo[k] + ck * h;
                const double b = ck == GL - 1 ? __builtin_inf() : t.lo[k] + (ck + 1) * h;
                double dk = fmax(fmax(a - ball[k], ball[k] - b), 0.0) - (L == 0 ? t.g_margin[k] : t.g_margin2[k]);
                if (dk > 0.0) rem -= dk * dk;
            }
            if (rem < 0.0) empty = true;
            else {
                const double w = __builtin_sqrt(rem) + m0;
                x0 = max(x0, grid_cell_axis<L>(t, 0, ball[0] - w));
                x1 = min(x1, grid_cell_axis<L>(t, 0, ball[0] + w));
                if (x1 < x0) empty = true;
            }
        }
        const int base = (cz * GL + cy) * GL;
        int b = 0, e = 0;
        if (!empty) {
            if (L == 0) { b = t.g_start[grid_word(t, base + x0)]; e = t.g_start[grid_word(t, base + x1 + 1)]; }
            else { b = t.g_start2[base + x0]; e = t.g_start2[base + x1 + 1]; }
        }
        rb = b; rl = e;     // first slot / end slot: nothing is computed from the two loads here, so they stay in flight
    };
    const std::integral_constant<int, 0> LV0;
    const std::integral_constant<int, 1> LV1;
    int nb0[3] = {0, 0, 0}, nb1[3] = {0, 0, 0}, qb0[3] = {0, 0, 0}, qb1[3] = {0, 0, 0};
    int cb0[3] = {0, 0, 0}, cb1[3] = {0, 0, 0}, db0[3] = {0, 0, 0}, db1[3] = {0, 0, 0};   // the same boxes on the coarse level
    int rowsN = 0, rowsQ = 0, rows2N = 0, rows2Q = 0;
    // whole-tree visit: no index yet, or a box of more rows than the range list holds (never the case for the Near
    // radius of an indexed tree)
    bool brute = ns == 0;
    const int ns2 = uni(t.g_ns2);
    bool lvl2 = !brute && ns2 > ns;       // a coarse level exists: rows of it are listed like rows of the main level
    bool coarse_whole = false;            // ... or, when the range list has no room for them, it is visited whole
    if (!brute) {
        if (wantN) { grid_box<D>(t, pnv, r, nb0, nb1); rowsN = grid_rows(nb0, nb1); }
        if (wantQ) {
            // first guess: the cells within 1.5x the typical nearest distance seen so far (the cell of q itself at first)
            grid_box<D>(t, qv, 1.5 * t.g_rho, qb0, qb1);
            rowsQ = grid_rows(qb0, qb1);
        }
        if (rowsN + rowsQ > GRID_RG_MAX) brute = true;
        else if (lvl2) {
            if (wantN) { grid_box<D, 1>(t, pnv, r, cb0, cb1); rows2N = grid_rows(cb0, cb1); }
            if (wantQ) { grid_box<D, 1>(t, qv, 1.5 * t.g_rho, db0, db1); rows2Q = grid_rows(db0, db1); }
            if (rowsN + rowsQ + rows2N + rows2Q > GRID_RG_MAX) { coarse_whole = true; rows2N = 0; rows2Q = 0; }
        }
    }
    if (brute) lvl2 = false;
    // (everything about the boxes is the same in every lane: scalar registers)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        nb0[k] = uni(nb0[k]); nb1[k] = uni(nb1[k]); qb0[k] = uni(qb0[k]); qb1[k] = uni(qb1[k]);
        cb0[k] = uni(cb0[k]); cb1[k] = uni(cb1[k]); db0[k] = uni(db0[k]); db1[k] = uni(db1[k]);
    }
    rowsN = uni(rowsN); rowsQ = uni(rowsQ); rows2N = uni(rows2N); rows2Q = uni(rows2Q);
    const int rows1 = rowsN + rowsQ, rowsAll = rows1 + rows2N + rows2Q;
    const unsigned fl_all = (wantN ? GRID_N : 0u) | (wantQ ? GRID_Q : 0u);
    // the rows' slot ranges (two g_start loads per row; at most GRID_RG_MAX <= NT rows: one row per thread) are
    // requested now and consumed after the first pass of the loop
    int rb = 0, rl = 0;
    if (!brute) {
        if (tid < rowsN) row_range(LV0, nb0, nb1, tid, pnv, r, rb, rl);
        else if (tid < rows1) row_range(LV0, qb0, qb1, tid - rowsN, nullptr, 0., rb, rl);
        else if (tid < rows1 + rows2N) row_range(LV1, cb0, cb1, tid - rows1, pnv, r, rb, rl);
        else if (tid < rowsAll) row_range(LV1, db0, db1, tid - rows1 - rows2N, nullptr, 0., rb, rl);
    }
    auto publish = [&](bool mine, int my_beg, int my_len, unsigned my_flag, int extra_beg, int extra_len, unsigned extra_flag) {
        rg_publish<NT>(mine, my_beg, my_len, my_flag, extra_beg, extra_len, extra_flag);
    };
    double m1 = __builtin_inf(), m2 = __builtin_inf();   // nearest: smallest / second-smallest squared distance of this lane
    int i1 = 0x7fffffff;
    // Near: this lane's best cost + dist (ba: by sqrt(v) until b_exact, then the reference's value) and its index
    double ba = __builtin_inf();
    bool b_exact = false;
    int cj = 0x7fffffff;
    // 2D: offset to and cost of that member (its slot record's own doubles): the exact value is computed from registers -
    // re-reading the record for it cost a scattered 32-byte load per lane and query (64 cache lines of HBM traffic and a dependent
    // round trip at the end of every visit)
    double bdx = 0., bdy = 0., bcost = 0.;
    int n_mem = 0;                                        // members seen by this wave
    __syncthreads();
    const int n_ob = wantN ? uni(s.ob_n) : 0;
    if (brute) brutes++;
    PROF(20);
    // one visited slot; true = Near member (sm = cost - dist, what rewire compares with cost(new) later: members of a dense,
    // well optimised tree sit within 1e-5 of that threshold by the dozen, so the margin is kept in full float64)
    // Written with selects instead of nested branches (the common path is straight-line code; only the rare long paths -
    // guard-band re-decision, obstacle tests, near-ties of the running minimum - sit behind branches).
    auto process = [&](const SlotRegs &p, double &sm) -> bool {
        // (round 5) the loop is bound by instruction issue as much as by latency: a trip slot whose 64 lanes all sit in rows listed
        // for ONE of the two questions skips the other question's arithmetic altogether (wave-uniform branches: rows are listed
        // Near rows first, nearest rows after them, so most trip slots are pure)
        if (__ballot((p.fl & GRID_Q) != 0) != 0ull) {
            const bool isQ = (p.fl & GRID_Q) != 0;
            const double dx = qx - p.x, dy = qy - p.y;
            double wv = dx * dx + dy * dy;
            if (D == 3) { const double dz = qz - p.z; wv = wv + dz * dz; }
            const bool lt1 = isQ && (wv < m1 || (wv == m1 && p.id < i1));
            const double m2n = (isQ && wv < m2) ? wv : m2;
            m2 = lt1 ? m1 : m2n;
            m1 = lt1 ? wv : m1;
            i1 = lt1 ? p.id : i1;
        }
        if (__ballot((p.fl & GRID_N) != 0) == 0ull) { sm = 0.; return false; }   // nobody here answers the Near question
        const double dx = pnx - p.x, dy = pny - p.y, dz = D == 3 ? pnz - p.z : 0.;
        double v = dx * dx + dy * dy;
        if (D == 3) v = v + dz * dz;
        const bool in = (p.fl & GRID_N) != 0 && v <= r2hi && p.id != new_idx;
        // d_j: the reference's own distance (glibc hypot in 2D) decides inside the guard band and is what choose_parent
        // adds - but it is only evaluated where it can matter: sqrt(v) is within 2 ulp of it, which settles every
        // member that is not in the band and cannot reach this lane's best cost + dist so far
        const double ds = sqrt_fast(v);
        bool exact = D == 3;                    // 3D: dist_scan IS sqrt of this very sum
        double dj = ds;
        bool hit = in && v <= r2lo;
        if (in && !hit) {                       // inside the guard band
            if (!exact) { dj = dist_scan_cold<D>(dx, dy, dz); exact = true; }
            hit = dj <= r;
        }
        // The Near set is the collision-free hits (find_near_neighbors, rrt_star_2d.py:101-110).  lazy (the persistent loops): the
        // segment test runs only for a hit that can still become this lane's choose_parent candidate; the others go to the stash
        // untested and it_connect tests the few of them that reach cost(new) - a member is used in exactly those two places.  (3D
        // trees with thousands of Near members spent 90 % of their time testing every member against every obstacle in reach.)
        const double c = p.c + dj;
        bool col = false;
        if (n_ob > 0) {                         // uniform
            if (hit && (!lazy || c <= ba * (1.0 + 0x1p-40))) {
                double l0[3], l1[3];
                l0[0] = fmin(pnx, p.x); l1[0] = fmax(pnx, p.x); l0[1] = fmin(pny, p.y); l1[1] = fmax(pny, p.y);
                if (D == 3) { l0[D - 1] = fmin(pnz, p.z); l1[D - 1] = fmax(pnz, p.z); }
                for (int j = 0; j < n_ob && !col; j++) {
                    const int o = s.ob_list[j];
                    if (seg_aabb_pass<D, NT>(s, o, l0, l1)) col = seg_obstacle_cold<D>(o, pnx, pny, pnz, p.x, p.y, D == 3 ? p.z : 0.);
                }
            }
        }
        const bool member = hit && !col;
        sm = p.c - ds;               // margin for rewire's search (tolerance there >> 2 ulp)
        // choose_parent's argmin of cost_j + d_j with the reference's own d_j: the lane keeps its best member by the value
        // with sqrt(v) (within 2 ulp of d_j) and settles it exactly only when another member comes within 2^-40 of it; a
        // lane's best is evaluated exactly ONCE, after the visit.  (Evaluating at every new per-lane minimum put a call - and
        // with it a drain of the loads in flight - into nearly every trip.)
        const bool better = member && c < ba * (1.0 - 0x1p-40);             // clearly better than the lane's best
        const bool close = member && !better && c <= ba * (1.0 + 0x1p-40);   // too close to call: the reference's values decide
        ba = better ? c : ba;
        cj = better ? p.id : cj;
        b_exact = better ? exact : b_exact;
        if (D == 2) { bdx = better ? dx : bdx; bdy = better ? dy : bdy; bcost = better ? p.c : bcost; }
        if (close) {
            if (!b_exact) { ba = bcost + hypot_np_cold(bdx, bdy); b_exact = true; }
            const double ce = exact ? c : p.c + dist_scan_cold<D>(dx, dy, dz);
            if (ce < ba || (ce == ba && p.id < cj)) { ba = ce; cj = p.id; }
        }
        return member;
    };
    // members of one trip slot -> stash positions (wave ballot + one LDS atomic per wave); every lane of the wave is here.
    // Only members whose margin exceeds floor_m are kept: the caller passes a lower bound of cost(new), below which
    // rewire's test cannot pass (wg_iteration); the others are merely counted.
    auto stash = [&](bool member, int id, double sm) {
        n_mem += __popcll(__ballot(member));
        const bool keep = member && sm > floor_m;
        const unsigned long long mk = __ballot(keep);
        if (mk) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s.hit_cnt, __popcll(mk));
            base = __builtin_amdgcn_readfirstlane(base);
            if (keep) {
                const int p = base + __popcll(mk & lt);
                if (p < lds_cap) stash_put(s, p, id, sm);
                else { t.nr_idx[p - lds_cap] = id; t.nr_m[p - lds_cap] = sm; }
            }
        }
    };
    int stage = brute ? 2 : 0;   // 0: the appended range is being visited, the rows come next; 1: rows visited; 2: nearest widening / whole tree
    int pass = 0;
    double ring = 0.;
    int result_ni = -1;
    {
        // first pass: the vertices no level covers yet (and the coarse level whole if its rows found no room in the list)
        const int beg = brute ? 0 : (coarse_whole || !lvl2 ? ns : ns2);
        publish(tid == 0, beg, n - beg, fl_all, 0, 0, 0u);
    }
    // slots per lane and trip: GRID_U on the one- / two-wave workgroups (many trees share the memory system), GRID_U_WIDE on
    // the 256-lane ones (one or a few heavy trees: fewer, larger trips - a trip is a dependent round trip)
    constexpr int GU = NT >= 256 ? GRID_U_WIDE : GRID_U;
    constexpr int NW = NT / 64;
    const int wv = NT == 64 ? 0 : (tid >> 6);
    const unsigned long long le_mask = (2ull << lane) - 1ull;   // (lane 63: all ones)
    // one trip loop for both ways of finding a lane's slot: `fetch(Fbase, o)` loads the record of flat offset Fbase + tid
    // (Fbase = the workgroup's window, uniform and a multiple of NT).  The loads are issued unconditionally (lanes past the end
    // read slot 0 and get flag 0): a load behind a branch would make the number of loads in flight unknown to the wait-count
    // insertion, which then waits for all of them
    auto run_trips = [&](int total, auto &fetch) {
        SlotRegs cur[GU], nxt[GU];
#pragma unroll
        for (int u = 0; u < GU; u++) fetch(u * NT, cur[u]);
#pragma nounroll
        for (int f0 = 0; f0 < total; f0 += NT * GU) {
#pragma unroll
            for (int u = 0; u < GU; u++) fetch(f0 + NT * GU + u * NT, nxt[u]);   // (past the end: dummy loads)
#pragma unroll
            for (int u = 0; u < GU; u++) {
                if (f0 + u * NT < total) {   // uniform
                    double sm = 0.;
                    const bool member = cur[u].fl ? process(cur[u], sm) : false;
                    if (wantN && __ballot(member) != 0ull) stash(member, cur[u].id, sm);
                }
            }
#pragma unroll
            for (int u = 0; u < GU; u++) cur[u] = nxt[u];
        }
    };
#pragma nounroll
    for (;;) {
        // ---------------- visit the ranges listed in s.rg_* ----------------
        {
            const int total = uni(s.rg_total);
            visited += total;
            if (NIRRT_RG_FAST && __builtin_expect(uni(s.rg_fast) != 0, 1)) {      // (compiled out of the default build: one trip loop resident)
                // range of a flat offset = (range starts at or below it) - 1: the wave's 64-offset window is one word of the start-bit
                // map (a uniform LDS read), the starts before the window are carried along (windows are fetched in ascending order)
                const int n_words = (total + 63) >> 6;
                int rbase = 0;   // range starts before the workgroup's current window row
                auto fetch = [&](int Fbase, SlotRegs &o) {
                    const int off = Fbase + tid, w0 = Fbase >> 6;
                    unsigned long long M = 0ull;
                    int before = rbase, row = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        const unsigned long long m = w0 + k < n_words ? uni((long long)s.rg_bits[w0 + k]) : 0ull;
                        const int c = __popcll(m);
                        if (k < wv) before += c;
                        if (k == wv) M = m;
                        row += c;
                    }
                    rbase += row;
                    const bool live = off < total;
                    int r = before + __popcll(M & le_mask) - 1;
                    r = live ? r : 0;
                    const long long cb = *reinterpret_cast<const long long *>(&s.rg_cb[r][0]);   // (one 8-byte LDS read, unconditional)
                    const int c0 = (int)cb, b0 = (int)(cb >> 32);
                    const int sl = live ? b0 + (off - (c0 & 0xffff)) : 0;
                    slot_load<D>(t, sl, o.x, o.y, o.z, o.c, o.id);
                    o.fl = live ? (unsigned)c0 >> 16 : 0u;
                };
                run_trips(total, fetch);
            } else {
                // one range (the whole tree, the appended vertices) or more slots than the start-bit map covers: a lane's flat offsets
                // only grow (with u and with the trip), so its position in the range list is carried along
                int rr = 0, r_lo = 0, r_hi = s.rg_len[0], r_beg = s.rg_beg[0];
                unsigned r_flag = (unsigned)s.rg_flag[0];
                auto fetch = [&](int Fbase, SlotRegs &o) {
                    const int off = Fbase + tid;
                    const bool live = off < total;
                    if (live) {
                        while (off >= r_hi) {
                            rr++;
                            r_lo = r_hi;
                            r_hi += s.rg_len[rr]; r_beg = s.rg_beg[rr]; r_flag = (unsigned)s.rg_flag[rr];
                        }
                    }
                    const int sl = live ? r_beg + (off - r_lo) : 0;
                    slot_load<D>(t, sl, o.x, o.y, o.z, o.c, o.id);
                    o.fl = live ? r_flag : 0u;
                };
                run_trips(total, fetch);
            }
        }
        if (stage == 0) {
            // the rows of cells: the range list goes to LDS now (their g_start words have had the first pass to arrive)
            const unsigned my_flag = (tid < rowsN || (tid >= rows1 && tid < rows1 + rows2N)) ? GRID_N : GRID_Q;
            publish(tid < rowsAll, rb, rl - rb, my_flag, 0, 0, 0u);
            stage = 1;
            PROF(13);
            if (rowsAll > 0) continue;
        }
        if (!wantQ) break;
        // widen the box until it provably contains the nearest vertex: nothing found -> one more ring of cells (twice),
        // something found -> the box of the ball through the winner's distance (then final)
        double g1;
        const int gi = wg_nearest_finish<D, NT>(s, m1, i1, m2, &g1);
        if (brute) {   // the whole tree was visited
            result_ni = gi >= 0 ? gi : wg_nearest_exact<D, NT>(n, qx, qy, qz);
            break;
        }
        int eb0[3], eb1[3];
        if (g1 == __builtin_inf()) {
            double h = 0.;
#pragma unroll
            for (int k = 0; k < D; k++) h = fmax(h, t.g_h[k]);
            ring += h;
            grid_box<D>(t, qv, ring, eb0, eb1);
        } else {
            // every vertex that could beat (or tie with) the winner lies within its distance of q
            const double rad = __builtin_sqrt(g1 * BAND_HI) * (1.0 + 1e-9);
            grid_box<D>(t, qv, rad, eb0, eb1);
            bool covered = true;
#pragma unroll
            for (int k = 0; k < D; k++) covered = covered && eb0[k] >= qb0[k] && eb1[k] <= qb1[k];
            if (covered && lvl2 && !coarse_whole) {   // the same ball on the coarse level
                int fb0[3], fb1[3];
                grid_box<D, 1>(t, qv, rad, fb0, fb1);
#pragma unroll
                for (int k = 0; k < D; k++) covered = covered && fb0[k] >= db0[k] && fb1[k] <= db1[k];
            }
            if (covered) {
                result_ni = gi >= 0 ? gi : wg_nearest_exact<D, NT>(n, qx, qy, qz);
                if (tid == 0) {   // statistics only: no decision depends on it
                    TreeHot &tw = const_cast<TreeHot &>(t);
                    tw.g_rho = 0.875 * tw.g_rho + 0.125 * __builtin_sqrt(g1);
                }
                break;
            }
        }
        const int rowsE = grid_rows(eb0, eb1);
        // another visit, nearest only, with a fresh reduction (a vertex seen twice would look like its own runner-up)
        m1 = __builtin_inf(); m2 = __builtin_inf(); i1 = 0x7fffffff;
        if (rowsE > GRID_RG_MAX || pass >= 3 || (g1 == __builtin_inf() && pass >= 2)) {
            brute = true;
            brutes++;
            publish(tid == 0, 0, n, GRID_Q, 0, 0, 0u);
        } else {
            revisits++;
            int eb = 0, el = 0;
            if (tid < rowsE) row_range(LV0, eb0, eb1, tid, nullptr, 0., eb, el);
            publish(tid < rowsE, eb, el - eb, GRID_Q, ns, n - ns, GRID_Q);   // + everything behind the cell-ordered part
#pragma unroll
            for (int k = 0; k < 3; k++) { qb0[k] = eb0[k]; qb1[k] = eb1[k]; }
            coarse_whole = true;   // (everything behind the cell-ordered part has been visited whole from here on)
        }
        stage = 2;
        pass++;
    }
    PROF(14);
    double cand = ba;
    if (wantN) {
        if (lane == 0 && n_mem) atomicAdd(&s.mem_cnt, n_mem);
        if (D == 2 && cj != 0x7fffffff && !b_exact) cand = bcost + hypot_np(bdx, bdy);   // every lane's best with the reference's distance
        block_argmin<NT>(s, cand, cj);   // lexicographic (value, index): np.argmin's first minimum of the ascending list (barriers inside)
    }
    if (tid == 0) {
        s.qa.ni = result_ni;
        s.qa.cand = cand; s.qa.cj = cj;
        s.stat[ST_VISITED] += visited; s.stat[ST_VISIT_B] += visited * SlotBytes<D>::value; s.stat[ST_REVISITS] += revisits; s.stat[ST_BRUTE] += brutes;
        if (wantN) {
            const int ks = s.hit_cnt;
            s.stat[ST_MEMBERS] += s.mem_cnt;
            if (ks > lds_cap) s.stat[ST_SPILLED] += ks - lds_cap;
        }
    }
    __syncthreads();
    PROF(15);
}

template <int D, int NT>
__device__ __forceinline__ void wg_query(Lds<NT> &s, const TreeHot &t, int n, const double *pn, double r, int new_idx,
                                         const double *q, int *ni, NearResult *nr, int lds_cap,
                                         double floor_m = -__builtin_inf(), bool lazy = false)
{
    if (tidx<NT>() == 0) {   // the arguments are the same in every thread
        s.qa.n = n; s.qa.want = (pn ? 1 : 0) | (q ? 2 : 0); s.qa.lazy = lazy ? 1 : 0;
        s.qa.r = r; s.qa.floor_m = floor_m; s.qa.new_idx = new_idx; s.qa.lds_cap = lds_cap;
#pragma unroll
        for (int k = 0; k < 3; k++) { s.qa.pn[k] = (pn && k < D) ? pn[k] : 0.; s.qa.q[k] = (q && k < D) ? q[k] : 0.; }
    }
    wg_query_fn<D, NT>();   // barriers at both ends
    if (ni) *ni = uni(s.qa.ni);
    if (nr) {
        nr->cand = uni(s.qa.cand);
        nr->cj = uni(s.qa.cj);
        nr->k = uni(s.mem_cnt);
        nr->n_stash = uni(s.hit_cnt);
    }
}

// nearest_neighbor: argmin_i dist(q, v_i), lowest index on ties (np.argmin)
template <int D, int NT>
__device__ __forceinline__ int wg_nearest(Lds<NT> &s, const TreeHot &t, int n, const double *q)
{
    int ni = -1;
    wg_query<D, NT>(s, t, n, nullptr, 0., -1, q, &ni, nullptr, 0);
    return ni;
}

// chase parent chains leaf -> root (RRTBase.cost, rrt_base_2d.py:54-61): acc = 0; acc += elen[v]; v = parent[v] ...
// Up to WALK_R chains per lane are walked concurrently so that their dependent 16-byte loads overlap.
// idx[r] <= 0: inactive slot (the root costs 0).  A chain stops early when it reaches `stop_at` (> 0):
// the caller then continues the very same left-to-right sum with the cached tail of that vertex.
template <int D>
__device__ __forceinline__ int walk_chains(const TreeHot &t, int (&idx)[WALK_R], double (&acc)[WALK_R], int stop_at)
{
    int guard = t.cap + 1, nrec = 0;
    for (;;) {
        bool any = false;
#pragma unroll
        for (int r = 0; r < WALK_R; r++) any = any || (idx[r] > 0 && idx[r] != stop_at);
        if (!any || guard-- <= 0) break;
        Hop4 h[WALK_R];
#pragma unroll
        for (int r = 0; r < WALK_R; r++)
            if (idx[r] > 0 && idx[r] != stop_at) { h[r] = ld_hop(&t.topo[idx[r]]); nrec++; }
#pragma unroll
        for (int r = 0; r < WALK_R; r++) {
            if (idx[r] > 0 && idx[r] != stop_at) {
#pragma unroll
                for (int j = 0; j < NHOP; j++) {
                    if (idx[r] > 0 && idx[r] != stop_at) {
                        acc[r] += h[r].e[j];
                        idx[r] = h[r].a[j];
                    }
                }
            }
        }
    }
    return nrec;
}

// single chain
template <int D>
__device__ __forceinline__ double walk_cost(const TreeHot &t, int i)
{
    double acc = 0.;
    int guard = t.cap + 1;
    while (i > 0 && guard-- > 0) {
        acc += t.topo[i].e[0];
        i = t.topo[i].a[0];
    }
    return acc;
}

// child-list maintenance (one thread)
__device__ __forceinline__ void link_child(TreeHot &t, int v, int p)
{
    int f = t.topo[p].fc;
    t.topo[v].ns = f;
    t.topo[v].ps = -1;
    if (f >= 0) t.topo[f].ps = v;
    t.topo[p].fc = v;
}
__device__ __forceinline__ void unlink_child(TreeHot &t, int v, int p)
{
    int nx = t.topo[v].ns, pv = t.topo[v].ps;
    if (pv >= 0) t.topo[pv].ns = nx; else t.topo[p].fc = nx;
    if (nx >= 0) t.topo[nx].ps = pv;
}

// finish a cost walk that has reached `new`: add the recorded edge lengths new -> root, in that order (LDS part, then HBM part)
__device__ __forceinline__ double chain_finish(const LdsData &s, const TreeHot &t, double acc, int clen)
{
    const int n_lds = clen < CHAIN_MAX ? clen : CHAIN_MAX;
    for (int i = 0; i < n_lds; i++) acc += s.chainE[i];
    for (int i = CHAIN_MAX; i < clen; i++) acc += t.chain_g[i];
    return acc;
}

// rewire's candidate list lives where the Near stash was: ids in the stash's index area, one state byte per entry in its
// (dead) margin area
#define CAND_PASS 1u      // passes the reference's test with its current cost
#define CAND_DIRTY 2u     // cost changed since it was tested (by a re-parenting the reference performs BEFORE this member's turn)
#define CAND_BLOCKED 4u   // passes, but a passing ancestor with a lower index is re-parented first: tested again afterwards
#define CAND_DONE 8u      // re-parented in this pass
__device__ __forceinline__ unsigned char *cand_state(LdsData &s) { return reinterpret_cast<unsigned char *>(&s.pool[s.stash_off]); }

// Vertices were just re-parented under `through` (records and child lists already relinked): refresh the exact cost of
// everything below them.  The queue t.bfs_q[0, n_src) holds the re-parented vertices ("sources"; t.g_rank[] the source each
// queue entry descends from), s.bc_i[4] = n_src.  Breadth-first over the child lists, then one leaf->root walk per collected
// vertex from queue position walk_from on (WALK_R chains per lane).  Every such walk passes through `through` (= new_idx,
// the vertex they were all just hung under): it is chased in memory only up to there and finished from s.chainE, the
// edge-length sequence through -> root recorded this iteration - the same additions in the same order as a full walk.
// The cost lands in the vertex's record and in its slot record.  n_list > 0: a re-costed vertex that is on rewire's candidate
// list (first n_list stash ids) is marked for re-testing if its source has the LOWER index (the reference re-parents that
// source before the member's turn; a source with a higher index comes after it and must not change the member's test).
#define BFS_FRONT 32   // frontier entries kept in LDS per level (6 * BFS_FRONT ints: two buffers of vertex / first child / source)
template <int D, int NT>
NIRRT_FN __device__ void wg_recost_queue_fn(int n_src_in, int walk_from_in, int through_in, int n_list_in, int front_off_in, int stamp_in)
{
    // stamp != 0: the candidate list holds passing members only; a re-costed vertex that carries the pass's near-tie stamp
    // (and hangs below a source of lower index) joins the list for an exact re-test (s.n_cand grows; at most list_cap entries)
    const int stamp = uni(stamp_in);
    // front_off >= 0: byte offset (from the Near stash's margin area) of 6 * BFS_FRONT ints the traversal may use for its
    // frontier: a level of up to BFS_FRONT vertices then costs ONE memory round trip (the records of its children) instead
    // of re-reading the queue and the parents' records first - what counts for the long chains of degenerate trees
    Lds<NT> &s = g_lds;
    TreeHot &t = g_lds.hot;
    const int n_src = uni(n_src_in), walk_from = uni(walk_from_in), through = uni(through_in), n_list = uni(n_list_in);
    const int front_off = uni(front_off_in);
    int *fr = front_off >= 0 ? reinterpret_cast<int *>(cand_state(s) + front_off) : nullptr;
    const int tid = tidx<NT>();
    const int ns = uni(t.g_ns2);   // vertices below it have their slot in pos[]
#ifdef NIRRT_PROFILE
    long long rq0_ = wall_clock64();
#endif
    int head = 0, tail = n_src, level = 0, cur = 0;
    bool in_lds = false;   // the sources come from the global queue (their child lists have just been edited: read afresh)
    while (head < tail) {   // one BFS level per trip; uniform
        for (int i = head + tid; i < tail; i += NT) {
            int u, su, c;
            if (in_lds) { const int *f = fr + cur * 3 * BFS_FRONT; u = f[i - head]; c = f[BFS_FRONT + i - head]; su = f[2 * BFS_FRONT + i - head]; }
            else { u = t.bfs_q[i]; su = t.g_rank[i]; c = t.bfs_fc[i]; }
            if (c == -2) c = t.topo[u].fc;
            // the records of the NHOP - 1 levels below a re-parented vertex mention its edge too
            Hop4 hu;
            if (level < NHOP - 1 && c >= 0) hu = ld_hop(&t.topo[u]);
            while (c >= 0) {
                GAS Topo &hc = t.topo[c];
                const int c_ns = hc.ns, c_fc = hc.fc;   // one record: the sibling link now, the child link for the next level
                const int pos = atomicAdd(&s.bc_i[4], 1);
                t.bfs_q[pos] = c;
                t.g_rank[pos] = su;
                t.bfs_fc[pos] = c_fc;
                const int lp = pos - tail;
                if (fr && lp < BFS_FRONT) { int *f = fr + (cur ^ 1) * 3 * BFS_FRONT; f[lp] = c; f[BFS_FRONT + lp] = c_fc; f[2 * BFS_FRONT + lp] = su; }
                if (level < NHOP - 1) {   // entry 0 of the child's record (its own edge) is unchanged
#pragma unroll
                    for (int j = 1; j < NHOP; j++) { hc.e[j] = hu.e[j - 1]; hc.a[j] = hu.a[j - 1]; }
                }
                c = c_ns;
            }
        }
        __syncthreads();
        head = tail;
        tail = s.bc_i[4];
        in_lds = fr != nullptr && tail - head <= BFS_FRONT;
        cur ^= 1;
        level++;
        __syncthreads();
    }
#ifdef NIRRT_PROFILE
    { const long long n_ = wall_clock64(); if (tid == 0) { s.tree_g->prof[11] += n_ - rq0_; s.tree_g->prof[23] += level; } rq0_ = n_; }
#endif
    int nrec = 0;
    for (int base = walk_from; base < tail; base += NT * WALK_R) {
        int idx[WALK_R], who[WALK_R], slot[WALK_R], src[WALK_R];
        double acc[WALK_R];
#pragma unroll
        for (int r = 0; r < WALK_R; r++) {
            int i = base + r * NT + tid;
            who[r] = i < tail ? t.bfs_q[i] : -1;
            src[r] = i < tail ? t.g_rank[i] : 0;
            idx[r] = who[r];
            acc[r] = 0.;
        }
#pragma unroll
        for (int r = 0; r < WALK_R; r++) slot[r] = (who[r] >= 0 && who[r] < ns) ? t.topo[who[r]].slot : who[r];   // appended vertices: slot = index
        const int clen = s.chain_len;
        nrec += walk_chains<D>(t, idx, acc, through);
#pragma unroll
        for (int r = 0; r < WALK_R; r++) {
            if (who[r] >= 0 && idx[r] == through) acc[r] = chain_finish(s, t, acc[r], clen);
        }
#pragma unroll
        for (int r = 0; r < WALK_R; r++) {
            if (who[r] >= 0) {
                t.vrec[who[r]].cost = acc[r];
                slot_set_cost<D>(t, slot[r], acc[r]);
                const int li = t.topo[who[r]].flags;
                const int ts = stamp ? t.tie_stamp[who[r]] : 0;
                if (li & 1) { const int q = t.topo[who[r]].sol_q; t.sol_val[q] = acc[r] + t.sol_line[q]; t.sol_dirty = 1; }
                if (li & 2) t.gc_dirty = 1;
                if (n_list > 0 && src[r] < who[r]) {
                    int *ids = stash_ids(s);
                    bool listed = false;
                    for (int a = 0; a < n_list; a++)
                        if (ids[a] == who[r]) { listed = true; if (!(cand_state(s)[a] & CAND_DONE)) cand_state(s)[a] = CAND_DIRTY; }
                    if (!listed && stamp && ts == stamp) {
                        const int p = atomicAdd(&s.n_cand, 1);
                        if (p < s.stash_cap) { ids[p] = who[r]; cand_state(s)[p] = CAND_DIRTY; }
                        else t.status = NIRRT_E_CAPACITY;   // more re-tests than the list holds (64 places are kept free for them)
                    }
                }
            }
        }
    }
    // counters: every lane adds its own records (LDS atomic), the member count is uniform
    if (nrec) atomicAdd((unsigned long long *)&s.stat[ST_HOPREC], (unsigned long long)nrec);
    if (tid == 0) s.stat[ST_RECOST] += tail;
    __syncthreads();
}

template <int D, int NT>
__device__ __forceinline__ void wg_recost_queue(Lds<NT> &s, TreeHot &t, int n_src, int walk_from, int through, int n_list, int front_off = -1,
                                                int stamp = 0)
{
    wg_recost_queue_fn<D, NT>(n_src, walk_from, through, n_list, front_off, stamp);
}

// one re-parented vertex v: its own cost and everything below it
template <int D, int NT>
__device__ __forceinline__ void wg_recost_subtree(Lds<NT> &s, TreeHot &t, int v, int through, int n_list = 0)
{
    __syncthreads();
    if (tidx<NT>() == 0) { t.bfs_q[0] = v; t.g_rank[0] = v; t.bfs_fc[0] = -2; s.bc_i[4] = 1; }
    __syncthreads();
    wg_recost_queue<D, NT>(s, t, 1, 0, through, n_list);
}

// record the edge-length sequence new_idx -> root (LDS + HBM continuation) and return cost(new_idx) (thread 0 walks; uniform
// result).  The first record of the chain is s.hop_new (thread 0 has just written or read it).
template <int D, int NT>
__device__ __forceinline__ double wg_chain_of_new(Lds<NT> &s, const TreeHot &t, int new_idx)
{
    if (tidx<NT>() == 0) {
        double acc = 0.;
        int i = new_idx, len = 0, guard = t.cap + 1, nrec = 0;
        Hop4 h = s.hop_new;
        while (i > 0 && guard-- > 0) {
            if (nrec > 0) h = ld_hop(&t.topo[i]);
            nrec++;
#pragma unroll
            for (int j = 0; j < NHOP; j++) {
                if (i > 0) {
                    acc += h.e[j];
                    if (len < CHAIN_MAX) s.chainE[len] = h.e[j]; else t.chain_g[len] = h.e[j];
                    len++;
                    i = h.a[j];
                }
            }
        }
        s.chain_len = len;
        s.bc_d[6] = acc;
        s.stat[ST_HOPREC] += nrec > 0 ? nrec - 1 : 0;
    }
    __syncthreads();
    double c = s.bc_d[6];
    return c;
}

// glibc 2.35's sin / cos / atan2 (the x86-64 FMA variants: what math.sin / math.cos / math.atan2 of the reference's interpreter
// call on an FMA + AVX2 host), restated instruction by instruction - see csrc/glibc235_libm.inc.  With them the 2D steer is
// bit-identical to the reference's (rounds 1-4 used the device's own libm: vertices within 1e-9, and on degenerate trees - free
// straight start-goal segment, hundreds of near-ties per rewiring pass - an ulp was enough to flip a parent now and then).
#define GLIBC_NS glibc235
#define GLIBC_UNIFORM 1
#include "glibc235_device.inc"
#undef GLIBC_NS
#undef GLIBC_UNIFORM

// steer (new_state).  2D: rrt_star_2d.py:67-78 with the reference's own libm functions (above); 3D: rrt_star_3d.py:67-78, IEEE only.
template <int D>
__device__ __forceinline__ void steer(const TreeHot &t, const double *from, const double *to, double *out)
{
    double d[D];
#pragma unroll
    for (int k = 0; k < D; k++) d[k] = to[k] - from[k];
    double dist = hypot_py<D>(d);
    double m = dist < t.step_len ? dist : t.step_len;
    if (D == 2) {
        const glibc235::v2d cs = glibc235::cos_sin_atan2(d[1], d[0]);
        out[0] = from[0] + m * cs.x;
        out[1] = from[1] + m * cs.y;
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

// segment test spread over the workgroup: lane o tests obstacle o, OR-reduced (one out-of-line copy: the loop body calls it
// from three places)
template <int D, int NT>
NIRRT_FN __device__ bool wg_collision_fn(double ax, double ay, double az, double bx, double by, double bz)
{
    const Lds<NT> &s = g_lds;
    const double a[3] = {ax, ay, az}, b[3] = {bx, by, bz};
    const double clr = s.k_clr;
    int M = s.n_round + s.n_box;
    bool hit = false;
    for (int o = tidx<NT>(); o < M; o += NT) hit = hit || seg_obstacle<D, NT>(s, o, a, b, clr);
    return block_any<NT>(hit);
}
template <int D, int NT>
__device__ __forceinline__ bool wg_collision(const Lds<NT> &s, const double *a, const double *b, double clr)
{
    // (clr is the tree's clearance, which is what s.k_clr holds)
    return wg_collision_fn<D, NT>(a[0], a[1], D == 3 ? a[D - 1] : 0., b[0], b[1], D == 3 ? b[D - 1] : 0.);
}

// find_near_neighbors as a primitive (nirrt_near): every member index goes to t.nr_idx[0, k) in visiting order (the host
// sorts them ascending, which is the reference's np.where order); returns k.
template <int D, int NT>
__device__ __forceinline__ int wg_near_list(Lds<NT> &s, TreeHot &t, int n, const double *node_new, int new_idx)
{
    NearResult nr;
    wg_query<D, NT>(s, t, n, node_new, t.near_r[n], new_idx, nullptr, nullptr, &nr, 0);
    return nr.k;
}

// find_best_path_solution (irrt_star_2d.py:84-97): argmin_s cost(sol[s]) + Line(v_s, goal), first minimum.
// The costs come from the exact cache; the argmin is only redone when a listed vertex was re-costed (sol_dirty),
// a solution appended in between competes with the standing minimum (strict <, so the first minimum stays).
template <int D, int NT>
__device__ __forceinline__ void wg_best_solution(Lds<NT> &s, TreeHot &t, double &c_best, int &x_best, bool want_x = true)
{
    const int tid = tidx<NT>();
    const int ns = t.n_sol;
    if (ns == 0) { c_best = __builtin_inf(); x_best = -1; return; }
    if (t.sol_dirty) {   // uniform
        double bv = __builtin_inf();
        int bs = 0x7fffffff;
        for (int q0 = tid; q0 < ns; q0 += NT * LIST_U) {   // LIST_U entries per lane and trip, loads issued back to back
            double c[LIST_U];                               // (cost(sol[q]) + Line(sol[q], goal), kept current by every re-costing)
#pragma unroll
            for (int u = 0; u < LIST_U; u++) c[u] = q0 + u * NT < ns ? t.sol_val[q0 + u * NT] : __builtin_inf();
#pragma unroll
            for (int u = 0; u < LIST_U; u++)
                if (q0 + u * NT < ns && c[u] < bv) { bv = c[u]; bs = q0 + u * NT; }
        }
        block_argmin<NT>(s, bv, bs);
        if (bs == 0x7fffffff) bs = 0;
        if (tid == 0) { t.sol_dirty = 0; t.sol_best = bs; t.sol_best_cost = bv; s.stat[ST_SOLSCAN] += ns; }
        __syncthreads();
    }
    c_best = t.sol_best_cost;
    x_best = want_x ? t.sol[t.sol_best] : -1;   // the persistent loops only need the cost: one dependent load less
}

// append a solution (InGoalRegion true).  Uniform; thread 0 writes.
template <int D, int NT>
__device__ __forceinline__ void wg_append_solution(Lds<NT> &s, TreeHot &t, int idx, const double *v)
{
    if (tidx<NT>() == 0) {
        if (t.n_sol < t.cap_sol) {
            int q = t.n_sol;
            double d[D];
#pragma unroll
            for (int k = 0; k < D; k++) d[k] = t.goal[k] - v[k];
            double line = hypot_py<D>(d);
            t.sol[q] = idx;
            t.sol_line[q] = line;
            const int fl = t.topo[idx].flags;
            const double c = t.vrec[idx].cost + line;
            // a vertex listed before ("same point" iterations append it again): its first position keeps the value
            t.sol_val[q] = (fl & 1) ? __builtin_inf() : c;
            if (!(fl & 1)) { t.topo[idx].flags = fl | 1; t.topo[idx].sol_q = q; }
            if (!t.sol_dirty) {
                if (q == 0 || c < t.sol_best_cost) { t.sol_best = q; t.sol_best_cost = c; }
            }
            t.n_sol = q + 1;
        } else {
            t.status = NIRRT_E_CAPACITY;
        }
    }
    __syncthreads();
}

// search_goal_parent (rrt_star_2d.py:101-117) over the maintained candidate list + path length
template <int D, int NT>
__device__ __forceinline__ void wg_goal_parent(Lds<NT> &s, TreeHot &t, int &gp, double &path_len)
{
    const int tid = tidx<NT>();
    const int ng = t.n_gc;
    if (ng == 0) { gp = -1; path_len = __builtin_inf(); return; }
    if (t.gc_dirty) {
        double bv = __builtin_inf();
        int bq = 0x7fffffff;
        for (int q0 = tid; q0 < ng; q0 += NT * LIST_U) {
            int vi[LIST_U];
            double di[LIST_U], co[LIST_U];
            bool col[LIST_U];
#pragma unroll
            for (int u = 0; u < LIST_U; u++) {
                const int q = q0 + u * NT;
                vi[u] = 0; di[u] = 0.; col[u] = true;
                if (q < ng) { vi[u] = t.gc_idx[q]; di[u] = t.gc_dist[q]; col[u] = t.gc_col[q] != 0; }
            }
#pragma unroll
            for (int u = 0; u < LIST_U; u++) co[u] = q0 + u * NT < ng ? t.vrec[vi[u]].cost : 0.;
#pragma unroll
            for (int u = 0; u < LIST_U; u++) {
                const double c = col[u] ? __builtin_inf() : co[u] + di[u];
                if (q0 + u * NT < ng && c < bv) { bv = c; bq = q0 + u * NT; }
            }
        }
        block_argmin<NT>(s, bv, bq);
        if (bq == 0x7fffffff) bq = 0;  // every candidate collides: np.argmin of all-inf = 0
        if (tid == 0) { t.gc_dirty = 0; t.gc_best = bq; t.gc_best_cost = bv; s.stat[ST_LISTSCAN] += ng; }
        __syncthreads();
    }
    gp = t.gc_idx[t.gc_best];
    // get_path_len(extract_path(gp)): sum of segment norms, goal <- gp <- ... <- start
    if (tid == 0) {
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
            i = t.topo[i].a[0];
        }
        s.bc_d[7] = len;
    }
    __syncthreads();
    path_len = s.bc_d[7];
    __syncthreads();
}

// bookkeeping when a vertex (idx, coordinates v) joined the tree: RRT* goal-candidate list.
// Block-uniform control flow; `v` identical in all threads; cost[idx] must be final.
template <int D, int NT>
__device__ __forceinline__ void wg_goal_candidate(Lds<NT> &s, TreeHot &t, int idx, const double *v)
{
    double d[D];
#pragma unroll
    for (int k = 0; k < D; k++) d[k] = t.goal[k] - v[k];
    if (dist2<D>(d) > t.step_len * t.step_len * BAND_HI) return;   // uniform: clearly outside
    double h = dist_scan<D>(d);
    if (h <= t.step_len) {  // uniform
        bool col = wg_collision<D, NT>(s, v, t.goal, t.clearance);
        if (tidx<NT>() == 0) {
            int q = t.n_gc;
            t.gc_idx[q] = idx;
            t.gc_dist[q] = h;
            t.topo[idx].flags |= 2;
            t.gc_col[q] = col ? 1 : 0;
            if (!t.gc_dirty) {
                double c = col ? __builtin_inf() : t.vrec[idx].cost + h;
                if (q == 0 || c < t.gc_best_cost) { t.gc_best = q; t.gc_best_cost = c; }
            }
            t.n_gc = q + 1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// one loop body (rrt_star_2d.py:37-55 / irrt_star_2d.py:54-73) in four out-of-line phases that hand their state on through
// LDS (s.it): each phase gets a register allocation of its own, so nothing one phase keeps alive spills into the loops of
// another.   it_extend: nearest / steer / edge test / insertion -> wg_query_fn -> it_connect: choose_parent, cost(new),
// rewire -> it_book: goal bookkeeping.
//   host_steer: node_in = node_new computed by the caller and nearest_in its nearest index
//   else      : node_in = node_rand
// ------------------------------------------------------------------------------------------------
template <int D, int NT>
NIRRT_FN __device__ void it_extend()
{
    // pref_ni >= 0: nearest_neighbor(node_in) already known from the previous iteration's fused query.
    // q_next != nullptr: the next iteration's node_rand; if this iteration runs a Near query its nearest index is
    // left in s.bc_i[6] (else -1) for the caller to pass back as pref_ni.
    Lds<NT> &s = g_lds;
    TreeHot &t = g_lds.hot;
    const int tid = tidx<NT>();
    const double clr = t.clearance;
    int n = uni(t.n);
    const bool host_steer = uni(s.it.host_steer) != 0;
    const int nearest_in = uni(s.it.ni), pref_ni = uni(s.it.pref_ni);
    nirrt_step_result *res = s.it.res;
    const double node_in[3] = {uni(s.it.node_in[0]), uni(s.it.node_in[1]), uni(s.it.node_in[2])};
    long long alg = host_steer ? 0 : n;
    PROF_DECL
    // keep the cell-ordered part of the grid index within GRID_REBUILD_EVERY vertices of the tree
    if (n >= t.g_min && n - t.g_ns >= t.g_every) wg_grid_rebuild<D, NT>(n);   // uniform
    else if (t.g_ns > 0 && n - t.g_ns2 >= t.g_every2) wg_grid_rebuild2<D, NT>(n);   // the coarse level over what was appended since
    PROF(12);
    int ni;
    double node_new[D], nearest[D];
    if (host_steer) {
        ni = nearest_in;
#pragma unroll
        for (int k = 0; k < D; k++) node_new[k] = node_in[k];
    } else {
        if (pref_ni >= 0) ni = pref_ni;
        else ni = wg_nearest<D, NT>(s, t, n, node_in);
        PROF(0);
    }
    ni = uni(ni);
    // coordinates and exact cost of the nearest vertex in one 32-byte record; the Near radii the iteration can need
    // (tree size unchanged for a "same point", + 1 otherwise) ride along in the same round trip
    // ... and so does the vertex's tree record (parent chain + head of its child list: what an insertion under it needs)
    const VRec vnear = ldg(&t.vrec[ni]);
    Hop4 hnear;
    TopoLinks tnear;
    ld_topo(&t.topo[ni], hnear, tnear);
    const double r_same = t.near_r[n], r_grown = t.near_r[n + 1 <= t.cap ? n + 1 : n];
    nearest[0] = vnear.x; nearest[1] = vnear.y;
    if (D == 3) nearest[D - 1] = vnear.z;
    if (!host_steer) steer<D>(t, nearest, node_in, node_new);
    if (res && tid == 0) {
        res->collided = 0; res->inserted = 0; res->nearest_idx = ni; res->new_idx = -1; res->n_near = 0;
        res->reparented = 0; res->n_rewired = 0; res->in_goal = 0; res->status = 0; res->reserved = 0;
        res->node_new[0] = node_new[0]; res->node_new[1] = node_new[1]; res->node_new[2] = D == 3 ? node_new[D - 1] : 0.;
    }
    bool collided = wg_collision<D, NT>(s, nearest, node_new, clr);
    // the restated libm routines return NaN on the paths their translation does not cover (|theta| > 1e8, Inf: unreachable for
    // atan2's results, but nothing else would notice): no NaN vertex enters the tree - the iteration is dropped and the run ends
    if (D == 2 && !host_steer && (node_new[0] != node_new[0] || node_new[1] != node_new[1])) {   // (a NaN end point collides with nothing)
        collided = true;
        if (tid == 0) t.status = NIRRT_E_LIBM;
    }
    PROF(1);
    int new_idx = -1;
    bool dup_ = false, inserted_ = false;
    double edge_new_ = 0.;
    if (!collided) {
        double diff[D];
#pragma unroll
        for (int k = 0; k < D; k++) diff[k] = node_new[k] - nearest[k];
        const bool dup = norm_1d<D>(diff) < 1e-8;
        const double edge_new = dup ? 0. : hypot_py<D>(diff);   // Line(nearest, new) == cost() term of new
        bool inserted = false;
        if (dup) {
            new_idx = ni;
            if (tid == 0) s.hop_new = hnear;   // only thread 0 reads it back
#pragma unroll
            for (int k = 0; k < D; k++) node_new[k] = nearest[k];
        } else if (n >= t.cap) {
            if (tid == 0) t.status = NIRRT_E_CAPACITY;
            new_idx = -2;  // tree full: treat as a no-op iteration
        } else {
            new_idx = n;
            if (tid == 0) {
                // (the nearest vertex's record arrived with its coordinates: stores only)
                const int fc_ni = tnear.fc;
                const Hop4 hn = hop_shift(hnear, edge_new, ni);
                s.hop_new = hn;
                // link_child(new_idx, ni) with the head read above; new's own links are remembered for a re-parenting
                TopoLinks ln;
                ln.fc = -1; ln.ns = fc_ni; ln.ps = -1; ln.flags = 0;
                st_hop(&t.topo[new_idx], hn);
                st_links(&t.topo[new_idx], ln);
                VRec vr;
                vr.x = node_new[0]; vr.y = node_new[1]; vr.z = D == 3 ? node_new[D - 1] : 0.; vr.cost = 0.;
                stg(&t.vrec[new_idx], vr);
                slot_store<D>(t, new_idx, node_new, 0., new_idx);   // its slot (appended vertices: slot = index)
                if (fc_ni >= 0) t.topo[fc_ni].ps = new_idx;
                t.topo[ni].fc = new_idx;
                s.new_next = fc_ni;
                s.new_fc = -1;
                t.n = n + 1;
                s.stat[ST_INSERTED] += 1;
            }
            n = n + 1;
            inserted = true;
            __syncthreads();
        }
        dup_ = dup; inserted_ = inserted; edge_new_ = edge_new;
    }
    PROF(6);
    // hand over (the values are the same in every thread)
    if (tid == 0) {
        s.it.collided = collided ? 1 : 0; s.it.new_idx = new_idx; s.it.ni = ni; s.it.n = n; s.it.alg = alg;
        if (!collided) {
#pragma unroll
            for (int k = 0; k < D; k++) s.it.node_new[k] = node_new[k];
            s.it.dup = dup_ ? 1 : 0; s.it.inserted = inserted_ ? 1 : 0; s.it.edge_new = edge_new_; s.it.cost_ni = vnear.cost;
            s.it.r_query = inserted_ ? r_grown : r_same;
            s.it.dup_parent = hnear.a[0]; s.it.dup_ns = tnear.ns; s.it.dup_ps = tnear.ps; s.it.dup_fc = tnear.fc;
        }
    }
    __syncthreads();
}

template <int D, int NT>
NIRRT_FN __device__ void it_connect()
{
    Lds<NT> &s = g_lds;
    TreeHot &t = g_lds.hot;
    const int tid = tidx<NT>();
    const int new_idx = uni(s.it.new_idx), ni = uni(s.it.ni);
    const bool dup = uni(s.it.dup) != 0;
    const double edge_new = uni(s.it.edge_new);
    double node_new[D];
#pragma unroll
    for (int kk = 0; kk < D; kk++) node_new[kk] = uni(s.it.node_new[kk]);
    struct { double cost; } vnear = {uni(s.it.cost_ni)};
    struct { int a[1]; int ns, ps, fc; } tnear = {{uni(s.it.dup_parent)}, uni(s.it.dup_ns), uni(s.it.dup_ps), uni(s.it.dup_fc)};
    struct { int k, n_stash, cj; double cand; } nr = {uni(s.mem_cnt), uni(s.hit_cnt), uni(s.qa.cj), uni(s.qa.cand)};
    const int cap_lds = uni(s.stash_cap);
    PROF_DECL
    int k_out = 0, reparented_out = 0, n_rewired_out = 0;
    {
        {
            // Near members.  lazy query: nr.k counts hits, filtered or not; the set is non-empty iff choose_parent has a candidate
            // (a lane filters its hits until one is free), and rewire filters its candidates below
            const bool lazy = uni(s.qa.lazy) != 0;
            const int k = lazy ? (nr.cand < __builtin_inf() ? (nr.k > 0 ? nr.k : 1) : 0) : nr.k;
            const int ks = nr.n_stash;     // ... of which on the stash
            const int n_ob = uni(s.ob_n);  // obstacles in reach of the Near ball (listed by the query)
            // is_collision(new, vertex id) against the query's obstacle list, one lane
            auto blocked = [&](int id) -> bool {
                if (!lazy || n_ob == 0) return false;
                const VRec vr = ldg(&t.vrec[id]);
                const double pz = D == 3 ? node_new[D - 1] : 0., vz = D == 3 ? vr.z : 0.;
                double l0[3], l1[3];
                l0[0] = fmin(node_new[0], vr.x); l1[0] = fmax(node_new[0], vr.x); l0[1] = fmin(node_new[1], vr.y); l1[1] = fmax(node_new[1], vr.y);
                if (D == 3) { l0[D - 1] = fmin(pz, vz); l1[D - 1] = fmax(pz, vz); }
                bool col = false;
                for (int j = 0; j < n_ob && !col; j++) {
                    const int o = s.ob_list[j];
                    if (seg_aabb_pass<D, NT>(s, o, l0, l1)) col = seg_obstacle_cold<D>(o, node_new[0], node_new[1], pz, vr.x, vr.y, vz);
                }
                return col;
            };
            PROF(2);
            int reparented = 0, n_rewired = 0;
            // curr_node_new_cost (rrt_star_2d.py:45 "same point" / :51)
            const double cost_ni = vnear.cost;   // no tree operation since the load changes a cost
            const double curr = dup ? cost_ni : cost_ni + edge_new;
            int best_parent = -1;
            // choose_parent (rrt_star_2d.py:80-90): argmin over the Near set of cost(j) + dist, first minimum
            if (k > 0 && nr.cand < curr) { reparented = 1; best_parent = nr.cj; }
            PROF(3);
            if (reparented) {
                if (tid == 0) {
                    // loads first (one round trip), then the stores
                    const VRec vb = ldg(&t.vrec[best_parent]);
                    const Hop4 hp = ld_hop(&t.topo[best_parent]);
                    const int fc_bp = t.topo[best_parent].fc;
                    int old_p, nx, pv;
                    if (dup) { old_p = tnear.a[0]; nx = tnear.ns; pv = tnear.ps; }   // new_idx == ni: its record is at hand
                    else { old_p = ni; nx = s.new_next; pv = -1; }   // just inserted at the head of ni's children
                    double d[D];
                    d[0] = node_new[0] - vb.x; d[1] = node_new[1] - vb.y;
                    if (D == 3) d[D - 1] = node_new[D - 1] - vb.z;
                    const double el = hypot_py<D>(d);
                    // unlink from the old parent
                    if (pv >= 0) t.topo[pv].ns = nx; else t.topo[old_p].fc = nx;
                    if (nx >= 0) t.topo[nx].ps = pv;
                    s.hop_new = hop_shift(hp, el, best_parent);
                    st_hop(&t.topo[new_idx], s.hop_new);
                    // link under the new parent; if that is the old parent again (its Near distance can beat the steer
                    // edge by an ulp) the head read above may be new_idx itself: use the list as the unlink left it
                    const int head = (best_parent == old_p && pv < 0) ? nx : fc_bp;
                    t.topo[new_idx].ns = head;
                    t.topo[new_idx].ps = -1;
                    if (head >= 0) t.topo[head].ps = new_idx;
                    t.topo[best_parent].fc = new_idx;
                    s.new_next = head;
                }
                __syncthreads();
            }
            // cost(new) = the one long pointer chase of the iteration; its edge lengths stay in LDS for the
            // re-costing below.  An existing vertex that moved (same-point case) takes its subtree along.
            double new_cost = cost_ni;
            if (k > 0 || !dup) {
                new_cost = wg_chain_of_new<D, NT>(s, t, new_idx);
                if (dup) {
                    if (reparented) wg_recost_subtree<D, NT>(s, t, new_idx, new_idx);
                } else {
                    if (tid == 0) { t.vrec[new_idx].cost = new_cost; slot_set_cost<D>(t, new_idx, new_cost); }
                }
            }
            PROF(4);
            if (ks > 0) {
                // rewire (rrt_star_2d.py:92-99), sequential semantics: members in ascending index order, each tested with
                // its CURRENT cost.  Costs only ever drop during a rewire pass, so a member can only pass
                // `cost(j) > cost(new) + d_j` if its stashed margin cost(j) - d_j (visit time, rounding ~1e-13) reaches
                // cost(new): the stash is searched for the lowest such index above the last one handled, that member's
                // record is read afresh and the reference's test decides; a re-parented vertex's subtree is re-costed
                // before the search resumes.
                const double thr = new_cost - (1e-10 + 1e-12 * new_cost);
                const int k_lds = ks < cap_lds ? ks : cap_lds;
                int *ids = stash_ids(s);
                unsigned char *state = cand_state(s);
                const int list_cap = cap_lds;   // the LDS part always fits (it shrinks in place); spilled candidates may not
                // The stash is compacted IN PLACE to the candidates - the members whose margin reaches cost(new), few unless the
                // tree is degenerate; candidates from the spilled part are appended while there is room.  (One trip = read a
                // slice, barrier, write: a write lands at or below the slice just read.)
                // Pass 1 classifies the stash entries whose margin reaches cost(new) WITHOUT touching the stash: collision filter and,
                // if there are many of them, the reference's test with the member's current cost.  A degenerate tree (free straight segment: hundreds of
                // members within 1e-10 of cost(new)) has a few passing members among hundreds of near-ties, and more candidates
                // than the list holds used to send every re-parenting down the one-at-a-time path (7.5 of 10 per iteration in the
                // slowest tree of the bench).  Now the list takes the PASSING members only (`fresh`); the near-ties get the pass's
                // stamp in tie_stamp[] and come back only if an earlier re-parenting re-costs them (wg_recost_queue_fn).
                auto passes_now = [&](int id) -> bool {
                    const VRec vr = ldg(&t.vrec[id]);
                    return vr.cost > new_cost + dist_scan_cold<D>(vr.x - node_new[0], vr.y - node_new[1], D == 3 ? vr.z - node_new[D - 1] : 0.);
                };
                __syncthreads();
                if (tid == 0) { s.n_cand = 0; s.cand_listed = 1; s.bc_i[1] = 0; s.bc_i[2] = 0; }
                __syncthreads();
                unsigned m_pass = 0u, m_tie = 0u;   // this lane's entries of the LDS part, one bit per trip (<= 13 trips of 64 lanes)
                {
                    // 1a: who is a candidate at all (margin + collision filter)
                    int trip = 0, nc_ = 0;
                    for (int base = 0; base < k_lds; base += NT, trip++) {
                        const int a = base + tid;
                        if (a < k_lds && s.pool[s.stash_off + a] >= thr && !blocked(ids[a])) { m_tie |= 1u << trip; nc_++; }
                    }
                    // spilled part (large Near sets only): the verdict replaces the margin - +inf passing, DBL_MAX candidate / near-tie,
                    // -inf neither - so that the later passes (and the one-at-a-time path, which asks `>= thr`) need not repeat the tests
                    for (int a0 = cap_lds + tid; a0 < ks; a0 += 4 * NT) {   // 4 entries per lane in flight
                        double mg[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) mg[u] = a0 + u * NT < ks ? t.nr_m[a0 + u * NT - cap_lds] : -__builtin_inf();
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (a0 + u * NT < ks) {
                                double code = -__builtin_inf();
                                if (mg[u] >= thr && !blocked(t.nr_idx[a0 + u * NT - cap_lds])) { code = 1.7976931348623157e308; nc_++; }
                                t.nr_m[a0 + u * NT - cap_lds] = code;
                            }
                        }
                    }
                    if (nc_) atomicAdd(&s.bc_i[2], nc_);
                }
                __syncthreads();
                const int n_candidates = uni(s.bc_i[2]);
                int n_passing = 0;
                // 1b (only when the candidates would crowd the list): the reference's test for each of them
                const bool classify = cap_lds >= 128 && n_candidates > 64;
                if (classify) {
                    int trip = 0, np_ = 0;
                    for (int base = 0; base < k_lds; base += NT, trip++) {
                        if (((m_tie >> trip) & 1u) && passes_now(ids[base + tid])) { m_tie &= ~(1u << trip); m_pass |= 1u << trip; np_++; }
                    }
                    for (int a0 = cap_lds + tid; a0 < ks; a0 += 4 * NT) {
                        double mg[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) mg[u] = a0 + u * NT < ks ? t.nr_m[a0 + u * NT - cap_lds] : -__builtin_inf();
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (mg[u] >= thr && passes_now(t.nr_idx[a0 + u * NT - cap_lds])) { t.nr_m[a0 + u * NT - cap_lds] = __builtin_inf(); np_++; }
                        }
                    }
                    if (np_) atomicAdd(&s.bc_i[1], np_);
                    __syncthreads();
                    n_passing = uni(s.bc_i[1]);
                }
                // fresh: passing members only on the list, 64 places kept free for near-ties that come back
                const bool fresh = classify && n_passing <= cap_lds - 64;
                int stamp = 0;
                if (fresh) {
                    stamp = uni(t.rw_stamp) + 1;
                    __syncthreads();
                    if (tid == 0) t.rw_stamp = stamp;
                }
                // Pass 2: the stash is compacted IN PLACE to the list (one trip = read a slice, barrier, write: a write lands at or
                // below the slice just read); entries from the spilled part are appended while there is room.
                {
                    int trip = 0;
                    for (int base = 0; base < k_lds; base += NT, trip++) {
                        const int a = base + tid;
                        const int id = a < k_lds ? ids[a] : 0;
                        const bool ps = (m_pass >> trip) & 1u, ti = (m_tie >> trip) & 1u;
                        const bool c = fresh ? ps : (ps || ti);
                        __syncthreads();
                        if (c) {
                            const int p = atomicAdd(&s.n_cand, 1);
                            if (p < list_cap) ids[p] = id; else s.cand_listed = 0;
                        }
                        if (fresh && ti) t.tie_stamp[id] = stamp;
                        __syncthreads();
                    }
                    for (int a0 = cap_lds + tid; a0 < ks; a0 += 4 * NT) {
                        double mg[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) mg[u] = a0 + u * NT < ks ? t.nr_m[a0 + u * NT - cap_lds] : -__builtin_inf();
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (mg[u] >= thr) {
                                const int id = t.nr_idx[a0 + u * NT - cap_lds];
                                const bool ps = mg[u] == __builtin_inf();
                                if (fresh ? ps : true) {
                                    const int p = atomicAdd(&s.n_cand, 1);
                                    if (p < list_cap) ids[p] = id; else s.cand_listed = 0;
                                } else {
                                    t.tie_stamp[id] = stamp;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                PROF(8);
                const int n_cand = uni(s.n_cand);
                // every candidate on the list: the margins are dead now, their bytes hold the candidates' state.  Not all on the
                // list (more candidates than the stash has room for): the LDS part cannot be trusted to be complete ->
                // every round re-tests the spilled stash too (slow, rare).
                const bool listed_all = uni(s.cand_listed) != 0;
                int n_list = n_cand < list_cap ? n_cand : list_cap;   // (fresh: grows when a stamped near-tie is re-costed)
                for (int a = tid; a < n_list; a += NT) state[a] = CAND_DIRTY;
                if (tid == 0) s.stat[ST_ROUNDS] += n_candidates;
                __syncthreads();
                int last = -1;
                if (dup && tid == 0) s.new_fc = tnear.fc;   // head of an existing vertex's child list (a fresh vertex's is in LDS already)
                // ---- sequential path (more candidates than the list holds / tiny stash) ----
                // the re-parenting of one member (thread 0) + re-costing of what hangs below it
                auto rewire_one = [&](int vj, const double *d) {
                    if (tid == 0) {
                        // loads first (one round trip), then the stores
                        const bool leaf = t.topo[vj].fc < 0;
                        const int old_p = t.topo[vj].a[0], nx = t.topo[vj].ns, pv = t.topo[vj].ps;
                        const int li = t.topo[vj].flags;
                        const int slot = vj < t.g_ns2 ? t.topo[vj].slot : vj;
                        const int fc_new = s.new_fc;   // head of new's child list: kept in LDS during the pass
                        const double el = hypot_py<D>(d);
                        if (pv >= 0) t.topo[pv].ns = nx; else t.topo[old_p].fc = nx;
                        if (nx >= 0) t.topo[nx].ps = pv;
                        st_hop(&t.topo[vj], hop_shift(s.hop_new, el, new_idx));
                        const int head = (old_p == new_idx && pv < 0) ? nx : fc_new;   // vj may already hang under new ("same point")
                        t.topo[vj].ns = head;
                        t.topo[vj].ps = -1;
                        if (head >= 0) t.topo[head].ps = vj;
                        t.topo[new_idx].fc = vj;
                        s.new_fc = vj;
                        // a leaf (the common case) has nothing below it: its new cost is its edge followed by the
                        // recorded chain new -> root, the same additions in the same order as a walk
                        const int clen = s.chain_len;
                        int fast = 0;
                        if (leaf) {
                            double acc = 0.;
                            acc += el;
                            acc = chain_finish(s, t, acc, clen);
                            t.vrec[vj].cost = acc;
                            slot_set_cost<D>(t, slot, acc);
                            if (li & 1) { const int q = t.topo[vj].sol_q; t.sol_val[q] = acc + t.sol_line[q]; t.sol_dirty = 1; }
                            if (li & 2) t.gc_dirty = 1;
                            fast = 1;
                        }
                        s.bc_i[7] = fast;
                        s.stat[ST_REWIRED] += 1;
                        s.stat[ST_RSEQ] += 1;
                    }
                    n_rewired++;
                    __syncthreads();
                    PROF(9);
                    if (!s.bc_i[7]) wg_recost_subtree<D, NT>(s, t, vj, new_idx, n_list);   // uniform
                    PROF(10);
                };
                // Candidates are tested in parallel against their CURRENT records, and tested again only after their cost changed
                // (a re-costed vertex marks itself on the list): the lowest index that passes is re-parented, its subtree re-costed,
                // and the next round looks at what is left.  Sequential round trips = vertices actually rewired + 1.
                auto passes = [&](int id) -> bool {
                    const VRec vr = ldg(&t.vrec[id]);
                    double d[D];
                    d[0] = vr.x - node_new[0]; d[1] = vr.y - node_new[1];
                    if (D == 3) d[D - 1] = vr.z - node_new[D - 1];
                    return vr.cost > new_cost + dist_scan<D>(d);
                };
                // ---- batched rounds -----------------------------------------------------------------------------------------
                // The reference's loop (rrt_star_2d.py:92-99) re-parents in ascending index order, each member tested with its
                // CURRENT cost; a re-parenting changes the costs of exactly the vertices below the re-parented one.  So a passing
                // member whose ancestors hold no passing member of LOWER index is tested with the cost the reference will see at
                // its turn, whatever happens to the others: all such members of a round are re-parented TOGETHER (they all end up
                // as children of `new`; child order carries no meaning), their subtrees are re-costed in ONE multi-source
                // traversal, and only members that sit below a lower-index source are tested again.  Rounds = longest chain of
                // such dependencies + 1 (usually 1-2) instead of one round per re-parented vertex.
                const bool batch_ok = listed_all && cap_lds >= 128;   // (room for the relink chunk behind the state bytes; small test builds take the sequential path)
                if (batch_ok) {
                    int *ch_v = reinterpret_cast<int *>(state + ((cap_lds + 7) & ~7)), *ch_pv = ch_v + 64, *ch_nx = ch_v + 128;
                    // behind the chunk arrays: the frontier buffers of the subtree traversal, if the margin area has the room
                    const int front_off = (((cap_lds + 7) & ~7) + 768 + 24 * BFS_FRONT <= 8 * cap_lds) ? ((cap_lds + 7) & ~7) + 768 : -1;
                    const int lane = tid & 63;
                    const int ns_ = uni(t.g_ns2);   // vertices below it have their slot in pos[]
                    const double bound = new_cost - (1e-9 + 1e-11 * new_cost);   // an ancestor that passes costs more than cost(new)
                    const int clen = s.chain_len;
                    VRec vr;
                    TopoLinks tp;      // (single: the candidate's links + parent stay in registers; its hops are read when phase B needs them)
                    int tp_a0 = 0;
                    int slot = 0;
                    for (;;) {
                        __syncthreads();
                        if (tid == 0) { s.bc_i[5] = 0; s.bc_i[4] = 0; s.bc_i[7] = 0; }
                        __syncthreads();
                        if (fresh) { const int nc = uni(s.n_cand); n_list = nc < list_cap ? nc : list_cap; }
                        const bool single = n_list <= 64;   // every candidate has its own lane of wave 0: records stay in registers across the phases
                        // phase A: test what changed
                        for (int base = 0; base < n_list; base += NT) {
                            const int a = base + tid;
                            if (a < n_list) {
                                unsigned st = state[a];
                                if (st & CAND_DIRTY) {
                                    const int id = ids[a];
                                    vr = ldg(&t.vrec[id]);
                                    if (single) { tp = ld_links(&t.topo[id]); tp_a0 = t.topo[id].a[0]; slot = id < ns_ ? t.topo[id].slot : id; }
                                    const double dx = vr.x - node_new[0], dy = vr.y - node_new[1], dz = D == 3 ? vr.z - node_new[D - 1] : 0.;
                                    st = vr.cost > new_cost + dist_scan_cold<D>(dx, dy, dz) ? CAND_PASS : 0u;
                                    state[a] = (unsigned char)st;
                                }
                                if (st & CAND_PASS) {   // the passing members of this round, compact (phase B looks ancestors up in it)
                                    const int pi = atomicAdd(&s.bc_i[5], 1);
                                    if (pi < 192) ch_v[pi] = ids[a];
                                }
                            }
                        }
                        __syncthreads();
                        const int n_pass = uni(s.bc_i[5]);
                        PROF(21);
                        if (n_pass == 0) break;
                        if (tid == 0) s.stat[ST_RROUNDS] += 1;
                        // phase B: a passing member with a passing ancestor of lower index waits for that one
                        for (int base = 0; base < n_list; base += NT) {
                            const int a = base + tid;
                            if (a < n_list && state[a] == CAND_PASS) {
                                const int id = ids[a];
                                Hop4 h = ld_hop(&t.topo[id]);
                                double cst = single ? vr.cost : t.vrec[id].cost;
                                bool blocked = false, stop = false;
                                for (int guard = 0; guard <= t.cap && !stop; guard++) {
#pragma unroll
                                    for (int j = 0; j < NHOP; j++) {
                                        if (!stop) {
                                            const int anc = h.a[j];
                                            cst -= h.e[j];   // ~ cost(anc)
                                            if (anc <= 0 || cst < bound) stop = true;
                                            else if (anc < id) {
                                                if (n_pass <= 192) {
                                                    for (int b = 0; b < n_pass; b++)
                                                        if (ch_v[b] == anc) { blocked = true; stop = true; }
                                                } else {
                                                    for (int b = 0; b < n_list; b++)
                                                        if (ids[b] == anc && (state[b] & CAND_PASS)) { blocked = true; stop = true; }
                                                }
                                            }
                                        }
                                    }
                                    if (!stop) h = ld_hop(&t.topo[h.a[NHOP - 1]]);
                                }
                                if (blocked) state[a] = (unsigned char)(CAND_PASS | CAND_BLOCKED);
                            }
                        }
                        __syncthreads();
                        PROF(22);
                        // phase C: re-parent the others, 64 at a time (wave 0): unlink (runs of adjacent siblings resolved in LDS),
                        // new records + costs, then the whole chunk is pushed onto new's child list
                        // A long list (hundreds of near-tie candidates in degenerate trees) holds its few passing members anywhere:
                        // their list positions are compacted first (into the traversal's frontier buffer, idle until the
                        // re-costing), so that they are re-parented in ceil(n / 64) chunks instead of one chunk per stretch of 64
                        // list positions that happens to hold one - every chunk waits for its stores three times
                        int *go_a = (!single && front_off >= 0) ? reinterpret_cast<int *>(state + front_off) : nullptr;
                        int n_go = -1;
                        if (go_a) {
                            if (tid == 0) s.bc_i[3] = 0;
                            __syncthreads();
                            for (int base = 0; base < n_list; base += NT) {
                                const int a = base + tid;
                                if (a < n_list && state[a] == CAND_PASS) {
                                    const int p = atomicAdd(&s.bc_i[3], 1);
                                    if (p < 6 * BFS_FRONT) go_a[p] = a;
                                }
                            }
                            __syncthreads();
                            n_go = uni(s.bc_i[3]);
                            if (n_go > 6 * BFS_FRONT) n_go = -1;   // (more than the buffer holds: chunks by list position)
                        }
                        for (int base = 0; base < (n_go >= 0 ? n_go : n_list); base += 64) {
                            int a = base + tid;
                            bool mine;
                            if (n_go >= 0) { mine = tid < 64 && a < n_go; a = mine ? go_a[a] : 0; }
                            else mine = tid < 64 && a < n_list && state[a] == CAND_PASS;
                            if (n_go < 0 && !block_any<NT>(mine)) continue;   // (uniform) nothing to re-parent among these 64
                            int v = -1, pv = -1, nx = -1, old_p = -1, fc = -1, flg = 0;
                            if (mine) {
                                v = ids[a];
                                if (!single) { vr = ldg(&t.vrec[v]); tp = ld_links(&t.topo[v]); tp_a0 = t.topo[v].a[0]; slot = v < ns_ ? t.topo[v].slot : v; }
                                pv = tp.ps; nx = tp.ns; old_p = tp_a0; fc = tp.fc; flg = tp.flags;
                            }
                            if (tid < 64) { ch_v[tid] = v; ch_pv[tid] = pv; ch_nx[tid] = nx; }
                            __syncthreads();
                            double el = 0.;
                            if (mine) {
                                auto find = [&](int x) -> int {
                                    int r = -1;
                                    for (int j = 0; j < 64; j++) if (ch_v[j] == x) r = j;
                                    return r;
                                };
                                if (pv < 0 || find(pv) < 0) {   // first of a run of chunk members that are adjacent siblings
                                    int after = nx;
                                    while (after >= 0) {
                                        const int j = find(after);
                                        if (j < 0) break;
                                        after = ch_nx[j];
                                    }
                                    if (pv >= 0) t.topo[pv].ns = after;
                                    else if (old_p == new_idx) s.new_fc = after;   // head of new's own list (kept in LDS until the push)
                                    else t.topo[old_p].fc = after;
                                    if (after >= 0) t.topo[after].ps = pv;
                                }
                                double d[D];
                                d[0] = vr.x - node_new[0]; d[1] = vr.y - node_new[1];
                                if (D == 3) d[D - 1] = vr.z - node_new[D - 1];
                                el = hypot_py<D>(d);
                                // cost: its edge followed by the recorded chain new -> root, the same additions in the same order as a walk
                                double acc = 0.;
                                acc += el;
                                acc = chain_finish(s, t, acc, clen);
                                t.vrec[v].cost = acc;
                                slot_set_cost<D>(t, slot, acc);
                                if (flg & 1) { const int q = t.topo[v].sol_q; t.sol_val[q] = acc + t.sol_line[q]; t.sol_dirty = 1; }
                                if (flg & 2) t.gc_dirty = 1;
                                state[a] = (unsigned char)CAND_DONE;
                            }
                            __syncthreads();
                            if (tid < 64) {
                                const unsigned long long m = __ballot(mine);
                                if (m) {
                                    const int old_head = s.new_fc;
                                    if (mine) {
                                        const unsigned long long above = lane == 63 ? 0ull : (m & ~((2ull << lane) - 1ull));
                                        const unsigned long long below = m & ((1ull << lane) - 1ull);
                                        const int succ = above ? ch_v[__ffsll((long long)above) - 1] : old_head;
                                        const int pred = below ? ch_v[63 - __clzll((long long)below)] : -1;
                                        // hop part + sibling links; the child-list head of v stays as it is in memory (a chunk member that
                                        // was v's first child has just updated it)
                                        st_hop(&t.topo[v], hop_shift(s.hop_new, el, new_idx));
                                        t.topo[v].ns = succ;
                                        t.topo[v].ps = pred;
                                        if (!above && old_head >= 0) t.topo[old_head].ps = v;
                                        if (!below) { s.new_fc = v; t.topo[new_idx].fc = v; }
                                        if (fc >= 0) {   // something hangs below it
                                            const int qp = atomicAdd(&s.bc_i[4], 1);
                                            t.bfs_q[qp] = v;
                                            t.g_rank[qp] = v;
                                            t.bfs_fc[qp] = -2;   // (a chunk member that was v's first child has just edited the list)
                                        }
                                    }
                                    if (lane == 0) { s.bc_i[7] += __popcll(m); s.stat[ST_REWIRED] += __popcll(m); }
                                }
                            }
                            __syncthreads();
                        }
                        PROF(9);
                        n_rewired += uni(s.bc_i[7]);
                        const int n_src = uni(s.bc_i[4]);
                        if (n_src > 0) wg_recost_queue<D, NT>(s, t, n_src, n_src, new_idx, n_list, front_off, stamp);   // uniform
                        for (int a = tid; a < n_list; a += NT)
                            if (state[a] & CAND_BLOCKED) state[a] = (unsigned char)CAND_DIRTY;
                        PROF(10);
                    }
                } else
                while (n_cand > 0) {
                    int first = 0x7fffffff;
                    for (int a = tid; a < n_list; a += NT) {
                        const int id = ids[a];
                        if (id > last) {
                            unsigned st = state[a];
                            if (st & CAND_DIRTY) { st = passes(id) ? CAND_PASS : 0u; state[a] = (unsigned char)st; }
                            if ((st & CAND_PASS) && id < first) first = id;
                        }
                    }
                    if (!listed_all) {   // candidates that found no room on the list: straight from the spilled stash, every round
                        for (int a = cap_lds + tid; a < ks; a += NT) {
                            const int id = t.nr_idx[a - cap_lds];
                            if (id > last && id < first && t.nr_m[a - cap_lds] >= thr && passes(id) && !blocked(id)) first = id;
                        }
                    }
                    first = uni(block_min_int<NT>(s, first));
                    if (first == 0x7fffffff) break;
                    last = first;
                    const VRec vr = ldg(&t.vrec[first]);
                    double d[D];
                    d[0] = vr.x - node_new[0]; d[1] = vr.y - node_new[1];
                    if (D == 3) d[D - 1] = vr.z - node_new[D - 1];
                    rewire_one(first, d);
                }
            }
            k_out = k; reparented_out = reparented; n_rewired_out = n_rewired;
        }
    }
    PROF(5);
    if (tid == 0) { s.it.k = k_out; s.it.reparented = reparented_out; s.it.n_rewired = n_rewired_out; }
    __syncthreads();
}

template <int D, int NT>
NIRRT_FN __device__ void it_book()
{
    Lds<NT> &s = g_lds;
    TreeHot &t = g_lds.hot;
    const int tid = tidx<NT>();
    const int new_idx = uni(s.it.new_idx);
    const bool inserted = uni(s.it.inserted) != 0;
    const unsigned flags = (unsigned)uni((int)s.it.flags);
    const double clr = t.clearance;
    nirrt_step_result *res = s.it.res;
    double node_new[D];
#pragma unroll
    for (int kk = 0; kk < D; kk++) node_new[kk] = uni(s.it.node_new[kk]);
    PROF_DECL
    {
        {
            if (inserted) wg_goal_candidate<D, NT>(s, t, new_idx, node_new);
            int in_goal = 0;
            if (flags & NIRRT_F_IRRT) {
                // InGoalRegion (rrt_base_2d.py:87-89): Line(node_new, goal) < step_len and collision-free
                double d[D];
#pragma unroll
                for (int kk = 0; kk < D; kk++) d[kk] = t.goal[kk] - node_new[kk];
                if (dist2<D>(d) < t.step_len * t.step_len * BAND_HI && hypot_py<D>(d) < t.step_len) {
                    if (!wg_collision<D, NT>(s, node_new, t.goal, clr)) {
                        in_goal = 1;
                        wg_append_solution<D, NT>(s, t, new_idx, node_new);
                    }
                }
            }
            if (res && tid == 0) {
                res->inserted = inserted ? 1 : 0; res->new_idx = new_idx; res->n_near = s.it.k;
                res->reparented = s.it.reparented; res->n_rewired = s.it.n_rewired; res->in_goal = in_goal;
                res->node_new[0] = node_new[0]; res->node_new[1] = node_new[1];
                res->node_new[2] = D == 3 ? node_new[D - 1] : 0.;
            }
        }
    }
    PROF(6);
    __syncthreads();
}

// One iteration over the arguments thread 0 has left in s.it (node_in, host_steer, ni, pref_ni, flags, res, has_next, q_next);
// starts with the barrier that publishes them.
// pref_ni >= 0: nearest_neighbor(node_in) already known from the previous iteration's fused query.
// has_next: q_next is the next iteration's node_rand; if this iteration runs a Near query its nearest index is
// left in s.bc_i[6] (else -1) for the caller to pass back as pref_ni (the persistent loops: s.it.pref_ni is set to it as well).
template <int D, int NT>
__device__ __forceinline__ void wg_iteration_body(Lds<NT> &s, TreeHot &t)
{
    const int tid = tidx<NT>();
    __syncthreads();
    it_extend<D, NT>();
    nirrt_step_result *res = uni(s.it.res);
    int next_ni = -1;
    long long alg = s.it.alg;
    const int new_idx = uni(s.it.new_idx);
    if (!uni(s.it.collided)) {
        if (new_idx >= 0) {
            // the fused query: Near members of node_new (stash + choose_parent's argmin) and the next sample's nearest vertex.
            // Rewire can only re-parent a member j with cost(j) - d_j > cost(new) (rrt_star_2d.py:95), and cost(new) - a sum of
            // edge lengths along a polyline root -> new, each within an ulp - is at least the straight distance root -> new
            // up to 1e-13 relative.  Members whose margin stays below that floor are counted but not kept: in a converged
            // tree (costs close to straight-line distances: exactly the problems with thousands of Near members) that is
            // nearly all of them, and the stash stays inside LDS.
            double node_new[D], d_root[D];
#pragma unroll
            for (int kk = 0; kk < D; kk++) { node_new[kk] = uni(s.it.node_new[kk]); d_root[kk] = node_new[kk] - t.start[kk]; }
            const double lb_new = __builtin_sqrt(dist2<D>(d_root));
            const double floor_m = lb_new - (1e-9 + 1e-11 * lb_new);
            const int n = uni(s.it.n);
            // (the step kernel reports the size of the filtered Near set: it filters every hit; the loops only need the set's uses)
            wg_query<D, NT>(s, t, n, node_new, uni(s.it.r_query), new_idx, uni(s.it.has_next) ? s.it.q_next : nullptr, &next_ni, nullptr,
                            uni(s.stash_cap), floor_m, res == nullptr);
            alg += n;
            it_connect<D, NT>();
            // goal bookkeeping only concerns vertices within step_len of the goal (wg_goal_candidate / InGoalRegion test the same
            // distance again, with the reference's formulas); the step kernel's result record is filled in there too
            double d_goal[D];
#pragma unroll
            for (int kk = 0; kk < D; kk++) d_goal[kk] = t.goal[kk] - node_new[kk];
            if (res != nullptr || dist2<D>(d_goal) <= t.step_len * t.step_len * BAND_HI) it_book<D, NT>();
        }
    } else if (res && tid == 0) {
        res->collided = 1;
    }
    if (tid == 0) { s.stat[ST_ITERS] += 1; s.stat[ST_ALG] += alg; s.bc_i[6] = next_ni; s.it.pref_ni = next_ni; }
    __syncthreads();
}

// The same iteration for the persistent sampling loops (no result record; the Near set's collision filter deferred), written so
// that NO value stays in a register across a phase call: a phase clobbers every register - that is what lets it run without
// saving any - so whatever the caller keeps across the call is stored to scratch before and reloaded after it, per lane and per
// iteration.  Everything a segment between two calls needs is (re)read from LDS.  Leaves the next sample's nearest vertex in
// s.it.pref_ni (-1: no query ran).
template <int D, int NT>
__device__ __forceinline__ void wg_iteration_flat()
{
    Lds<NT> &s = g_lds;
    __syncthreads();
    it_extend<D, NT>();
    if (!uni(s.it.collided) && uni(s.it.new_idx) >= 0) {
        {
            // floor of the Near stash: see wg_iteration_body
            const TreeHot &t = g_lds.hot;
            double node_new[D], d_root[D];
#pragma unroll
            for (int kk = 0; kk < D; kk++) { node_new[kk] = uni(s.it.node_new[kk]); d_root[kk] = node_new[kk] - t.start[kk]; }
            const double lb_new = __builtin_sqrt(dist2<D>(d_root));
            const double floor_m = lb_new - (1e-9 + 1e-11 * lb_new);
            if (tidx<NT>() == 0) {
                const int n = s.it.n;
                const bool has_next = s.it.has_next != 0;
                s.qa.n = n; s.qa.want = 1 | (has_next ? 2 : 0); s.qa.lazy = 1;
                s.qa.r = s.it.r_query; s.qa.floor_m = floor_m; s.qa.new_idx = s.it.new_idx; s.qa.lds_cap = s.stash_cap;
#pragma unroll
                for (int k = 0; k < 3; k++) { s.qa.pn[k] = k < D ? node_new[k] : 0.; s.qa.q[k] = (has_next && k < D) ? s.it.q_next[k] : 0.; }
                s.it.alg += n;
            }
        }
        wg_query_fn<D, NT>();   // barriers at both ends
        if (tidx<NT>() == 0) s.it.next_ni = s.qa.ni;
        it_connect<D, NT>();
        {
            // goal bookkeeping only concerns vertices within step_len of the goal (wg_goal_candidate / InGoalRegion test the same
            // distance again, with the reference's formulas)
            const TreeHot &t = g_lds.hot;
            double d_goal[D];
#pragma unroll
            for (int kk = 0; kk < D; kk++) d_goal[kk] = t.goal[kk] - uni(s.it.node_new[kk]);
            if (dist2<D>(d_goal) <= t.step_len * t.step_len * BAND_HI) it_book<D, NT>();
        }
    } else if (tidx<NT>() == 0) {
        s.it.next_ni = -1;
    }
    if (tidx<NT>() == 0) { s.stat[ST_ITERS] += 1; s.stat[ST_ALG] += s.it.alg; s.bc_i[6] = s.it.next_ni; s.it.pref_ni = s.it.next_ni; }
    __syncthreads();
}

template <int D, int NT>
__device__ __forceinline__ void wg_iteration(Lds<NT> &s, TreeHot &t, const double *node_in, bool host_steer,
                                             int nearest_in, unsigned flags, nirrt_step_result *res)
{
    if (tidx<NT>() == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) s.it.node_in[k] = k < D ? node_in[k] : 0.;
        s.it.host_steer = host_steer ? 1 : 0; s.it.ni = nearest_in; s.it.pref_ni = -1; s.it.flags = flags; s.it.res = res;
        s.it.has_next = 0;
    }
    wg_iteration_body<D, NT>(s, t);
}

// end-of-iteration report shared by the step kernel and the persistent loops
template <int D, int NT>
__device__ __forceinline__ void wg_report(Lds<NT> &s, TreeHot &t, unsigned flags, double &cb, int &xb, bool want_x = true)
{
    cb = __builtin_inf();
    xb = -1;
    if (flags & NIRRT_F_IRRT) wg_best_solution<D, NT>(s, t, cb, xb, want_x);
    else if (flags & NIRRT_F_GOAL_SCAN) wg_goal_parent<D, NT>(s, t, xb, cb);
}
