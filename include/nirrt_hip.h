/*
 * nirrt_hip.h — C ABI of libnirrt_hip.so, the MI355X (gfx950) implementation of the
 * RRT* / Informed-RRT* planning inner loop of tedhuang96/nirrt_star.
 *
 * The reference is pure Python and has no FFI; each entry point below replaces one Python-level
 * function of the reference's hot path (file:line cited per function, paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add to call them from path_planning_classes{,_3d}/.
 *
 * Conventions
 *   - plain C: opaque handle, plain pointers + sizes, no C++/torch types.
 *   - every function returns 0 on success or a negative NIRRT_E_* code; nothing throws.
 *   - the caller owns all host buffers (C-contiguous float64 / int64 / uint8, like the numpy
 *     arrays the reference passes around); the library owns device memory until nirrt_destroy.
 *   - one tree = one opaque handle; a handle is not thread-safe.  The trees of a device share a pool of 32 HIP streams (a
 *     tree's own calls are ordered on its stream; two trees may share one); nirrt_run launches and times its kernels on streams
 *     that belong to the CALLING THREAD, so concurrent nirrt_run calls from different threads neither serialize nor see each
 *     other's kernels in kernel_ms.
 *   - vertices cross the boundary as (n, dim) row-major float64 (the reference's
 *     `self.vertices[:n]`), parents as int64 (`self.vertex_parents[:n]`).  In HBM a tree is ONE device range (its arena)
 *     holding records, not coordinate columns: a 32-byte vertex record {x, y, z, cost(v)} and a 128-byte
 *     tree record {eight hops of the parent chain with their edge lengths, child-list links, flags, slot number} per vertex, a packed
 *     28- / 36-byte slot record {coordinates, cost(v), vertex index} per vertex of the two-level uniform-grid index (cell-ordered part, coarse level over the recent insertions,
 *     unsorted rest), the solution / goal-candidate lists, the Near-radius table, the guidance cloud and the tree's two
 *     MT19937 generators.  Everything is float64 / int32 on the device; there are no float32 copies.
 *   - nearest_neighbor / find_near_neighbors answers come from the grid index on large trees and from whole scans on small
 *     ones; both give exactly the reference's answers (ties, ordering).
 *     Environment knobs read by nirrt_create: NIRRT_GRID_MIN (smallest indexed tree), NIRRT_GRID_REBUILD / NIRRT_GRID_REBUILD2
 *     (rebuild intervals of the two levels), NIRRT_GRID_G / NIRRT_GRID_G2 (cells per axis), NIRRT_POOL_CHUNK_MB (arena pool);
 *     by nirrt_run: NIRRT_WIDE_MAX_TREES, NIRRT_SLIM_MIN_TREES, NIRRT_FORCE_VARIANT.
 *   - all planner arithmetic is float64 and follows the reference's per-call-site formulas
 *     (SURVEY.md Appendix A); integer bookkeeping (indices, parents, n) is exact.
 */
#ifndef NIRRT_HIP_H
#define NIRRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIRRT_OK 0
#define NIRRT_E_ARG (-1)      /* bad argument (dim, sizes, NULL)                      */
#define NIRRT_E_HIP (-2)      /* a HIP runtime call failed (see nirrt_last_error)      */
#define NIRRT_E_CAPACITY (-3) /* tree / Near-set / obstacle / solution capacity hit    */
#define NIRRT_E_NODEVICE (-4) /* no gfx950 device visible                              */
#define NIRRT_E_STREAM (-5)   /* random-word stream exhausted inside nirrt_run         */
#define NIRRT_E_CLOUD (-6)    /* nirrt_run + NIRRT_F_PNG: guidance cloud refresh due (not an error) */
#define NIRRT_E_LIBM (-7)     /* the restated libm routines left their supported domain (a NaN steer result): the iteration was
                                 dropped and the run stopped instead of inserting a NaN vertex */
#define NIRRT_E_PARK (-8)     /* nirrt_run with park_limit: the launch ended early because enough trees were waiting for a
                                 guidance-cloud refresh (not an error: resume with the remaining budget) */

/* Version of this header's structs and entry points.  nirrt_abi_version() returns the value the LIBRARY was built with: a caller
 * compiled against another header must not pass it structs (nirrt_run_args grew in rounds 4, 5 and 6).  nirrt_run_args also
 * carries its own size in its first member and nirrt_run refuses a struct of another size with NIRRT_E_ARG. */
#define NIRRT_ABI_VERSION 6

#define NIRRT_MAX_OBSTACLES 64 /* per kind (round / box) */
#define NIRRT_OBSTACLE_POOL 640 /* = 4 * 64 + 6 * 64: the tables live in LDS, 4 * n_round + 6 * n_box doubles of them */
#define NIRRT_N_STATS 24       /* per-tree counters of a nirrt_run launch, see nirrt_run_args.stats */
#define NIRRT_NEAR_CAPACITY 0 /* unlimited: Near-set scratch is sized like the tree */

typedef struct nirrt_tree nirrt_tree;

/* Problem description = the constructor arguments of RRTStar2D/3D (rrt_star_2d.py:10-30,
 * rrt_base_2d.py:8-37) plus the obstacle tables Utils builds (rrt_utils_2d.py:5-17,
 * rrt_utils_3d.py:7-20). */
typedef struct nirrt_config {
    int32_t dim;       /* 2 or 3 */
    int32_t device_id; /* HIP device ordinal */
    int64_t iter_max;  /* capacity = 1 + iter_max vertices */
    double x_start[3];
    double x_goal[3];
    double step_len;
    double search_radius; /* gamma of the RRT* Near radius */
    double clearance;
    double range_lo[3]; /* x_range[0], y_range[0], z_range[0] */
    double range_hi[3];
    int32_t n_round;         /* circles (2D) / balls (3D) */
    const double *round_obs; /* (n_round, dim+1): cx, cy[, cz], r */
    int32_t n_box;           /* rectangles (2D) / boxes (3D) */
    const double *box_obs;   /* (n_box, 2*dim): x, y[, z], w, h[, d] */
} nirrt_config;

/* flags for nirrt_step / nirrt_extend / nirrt_run */
#define NIRRT_F_IRRT 1u      /* IRRT*: InGoalRegion bookkeeping + best-solution report           */
#define NIRRT_F_GOAL_SCAN 2u /* RRT* planning_random: search_goal_parent + path length each step */
#define NIRRT_F_PNG 8u        /* nirrt_run: NIRRT* sampling policy (nirrt_star_png_2d.py:99-130) with the cloud of nirrt_set_cloud */
#define NIRRT_F_STOP_FIRST 4u /* nirrt_run: leave the loop right after the iteration that yields the first finite
                                 best cost (phase 1 of planning_random, rrt_star_2d.py:230-232, irrt_star_2d.py:245-248) */

/* What one loop body did (rrt_star_2d.py:37-55 / irrt_star_2d.py:51-73). */
typedef struct nirrt_step_result {
    int32_t collided;   /* edge nearest->new hit an obstacle: iteration ended there      */
    int32_t inserted;   /* a vertex was appended (0 in the "same point" case, :41-45)    */
    int64_t nearest_idx;
    int64_t new_idx;    /* index of node_new, -1 if collided                              */
    int32_t n_near;     /* size of the filtered Near set                                  */
    int32_t reparented; /* choose_parent changed parent[new]                              */
    int32_t n_rewired;
    int32_t in_goal;    /* IRRT*: node_new appended to path_solutions                     */
    int64_t n;          /* num_vertices after the iteration                               */
    double node_new[3];
    double c_best;      /* F_IRRT: find_best_path_solution cost (inf if none);
                           F_GOAL_SCAN: get_path_len(extract_path(search_goal_parent()))  */
    int64_t x_best;     /* goal-parent vertex index of that solution, -1 if none          */
    int64_t n_solutions;
    int32_t status;     /* 0 or NIRRT_E_CAPACITY                                          */
    int32_t reserved;
} nirrt_step_result;

/* ---- lifetime ------------------------------------------------------------------------------ */
const char *nirrt_last_error(void);
int nirrt_abi_version(void);
int nirrt_device_count(int *count);
/* RRTBase2D/3D.__init__ (rrt_base_2d.py:8-37, rrt_base_3d.py:8-38): allocate the tree in HBM,
 * vertex 0 = x_start, parent[0] = 0, num_vertices = 1. */
int nirrt_create(const nirrt_config *cfg, nirrt_tree **out);
/* the same for n problems at once (the planner objects of an evaluation set, eval_planning_2d.py:83-136): per tree only host work
 * (its arena comes out of the pool, its descriptor is filled in and copied asynchronously), then ONE device pass over the batch.
 * out[0 .. n) are ordinary handles (nirrt_destroy each); on failure nothing is left allocated and every out[i] is NULL. */
int nirrt_create_batch(const nirrt_config *cfgs, int32_t n, nirrt_tree **out);
int nirrt_destroy(nirrt_tree *t);
int nirrt_reset(nirrt_tree *t);
/* Trees of at least 1 MB are carved out of multi-GB device chunks (NIRRT_POOL_CHUNK_MB, default 4096, 0 = one allocation per
 * tree; no counterpart in the reference, whose arrays are numpy's).  A destroyed tree's block serves the next tree that fits it;
 * a chunk whose last tree is destroyed starts over empty, and all but one idle chunk per device go back to the driver right away.
 * nirrt_pool_trim hands the remaining idle chunks back too - a process that is done with a batch and starts helpers that need the
 * memory calls this */
int nirrt_pool_trim(void);
/* Host helper: n raw 32-bit outputs of an MT19937 generator - numpy's legacy RandomState (np.random.seed, what rrt_base_2d.py /
 * rrt_star_2d.py draw from) and CPython's random.Random (irrt_star_2d.py's informed sampling) are this generator, and nirrt_run
 * consumes their raw outputs (np_words / py_words).  key (624 words) and *pos (0..624, 624 = block used up) are the generator's
 * state as get_state() / getstate() expose it; both are updated in place to the state after n outputs.  No device involved. */
int nirrt_mt19937_fill(uint32_t *key, int32_t *pos, int64_t n, uint32_t *out);
/* the same for every tree of a batch (same device and dim) in ONE launch, one workgroup per tree: the planner objects of an
 * evaluation set are single-use in the reference (demo_planning_2d.py:90); a benchmark step re-plans the same problems */
int nirrt_reset_batch(nirrt_tree *const *trees, int32_t n_trees);
/* The tree's OWN generators.  The reference draws from two process-global MT19937 generators inside the loop: numpy's legacy
 * RandomState (SampleFree rrt_base_2d.py:46-52, 3D SampleUnitBall irrt_star_3d.py:146-158, SamplePointCloud
 * nirrt_star_png_2d.py:129-130) and CPython's random (2D SampleUnitBall irrt_star_2d.py:146-151).  Every tree carries its own
 * pair in HBM: nirrt_set_generators = np.random.set_state / random.setstate (key (n_trees, 624) words + pos (n_trees,) in
 * 0..624, as get_state() / getstate() expose them; a NULL key table leaves that stream alone), nirrt_get_generators = the
 * state after whatever the device consumed, in the very representation get_state() would show (the block of the last
 * consumed output, pos in 1..624).  nirrt_run with np_words == NULL draws from them: the twist runs in the tree's wave
 * (three stretches of the recurrence over 64 lanes), outputs are tempered on read; nothing is generated on the host. */
int nirrt_set_generators(nirrt_tree *const *trees, int32_t n_trees, const uint32_t *np_key, const int32_t *np_pos,
                         const uint32_t *py_key, const int32_t *py_pos);
int nirrt_get_generators(nirrt_tree *const *trees, int32_t n_trees, uint32_t *np_key, int32_t *np_pos, uint32_t *py_key,
                         int32_t *py_pos);
/* the next n_words raw outputs of every tree's numpy (which = 0) / python (which = 1) generator, produced on the device and
 * consumed (rng.random_sample / rng.uniform of the guidance-cloud candidates, datasets/point_cloud_mask_utils.py:81-131, read
 * them in place): tree i's words at out + i * stride; out is a DEVICE pointer if out_on_device != 0, else host memory */
int nirrt_generator_words(nirrt_tree *const *trees, int32_t n_trees, int32_t which, int64_t n_words, uint32_t *out, int64_t stride,
                          int32_t out_on_device);
/* test/bring-up helper: load a frozen tree (vertices (n,dim) f64, parents (n,) i64); vertex 0 must be x_start - the tree is
 * rooted at the start state like the reference's (rrt_base_2d.py:27) - else NIRRT_E_ARG */
int nirrt_upload(nirrt_tree *t, int64_t n, const double *vertices, const int64_t *parents);
/* `self.vertices[:n]`, `self.vertex_parents[:n]`, `self.num_vertices`; either pointer may be NULL */
int nirrt_download(nirrt_tree *t, double *vertices, int64_t *parents, int64_t *n);
int nirrt_num_vertices(nirrt_tree *t, int64_t *n);

/* ---- primitives (one kernel each; bring-up + parity tests) ---------------------------------- */
/* nearest_neighbor, rrt_base_2d.py:94-107 / rrt_base_3d.py:100-113 (np.argmin: lowest index on ties) */
int nirrt_nearest(nirrt_tree *t, const double *q, int64_t *idx);
/* Utils.is_collision, rrt_utils_2d.py:19-33 -> check_collision_line_circles_rectangles
 * (collision_check_utils.py:158-218); 3D rrt_utils_3d.py:22-36 -> check_collision_line_balls_boxes
 * (collision_check_utils_3d.py:151-216).  seg = (n_seg, 2, dim) f64; out[i] = 0/1. */
int nirrt_collision_batch(nirrt_tree *t, int64_t n_seg, const double *seg, uint8_t *out);
/* the same test for ONE segment per tree, each against its own tree's obstacles, in one launch (a batch's "is the straight
 * start-goal segment free?" probes): seg = (n_trees, 2, dim) f64, out[i] = 0/1; trees of one device and dimension */
int nirrt_collision_each(nirrt_tree *const *trees, int32_t n_trees, const double *seg, uint8_t *out);
/* Utils.is_inside_obs / Utils.is_valid (rrt_utils_2d.py:35-79, rrt_utils_3d.py:39-86);
 * pts = (n, dim); either output may be NULL. */
int nirrt_points_in_obs(nirrt_tree *t, int64_t n, const double *pts, uint8_t *inside, uint8_t *valid);
/* find_near_neighbors, rrt_star_2d.py:125-144 / rrt_star_3d.py:125-145: ascending indices of the
 * collision-free vertices within r(n) of node_new, excluding new_idx; *k = count (<= cap written). */
int nirrt_near(nirrt_tree *t, const double *node_new, int64_t new_idx, int64_t *k, int64_t *idx_out, int64_t cap);
/* RRTBase.cost, rrt_base_2d.py:54-61 / rrt_base_3d.py:60-67, for a batch of vertex indices */
int nirrt_cost(nirrt_tree *t, int64_t n_idx, const int64_t *idx, double *out);
/* search_goal_parent (rrt_star_2d.py:101-117): *idx = -1 for None; *path_len = get_path_len of the
 * extracted path (rrt_base_2d.py:79-85), inf if None */
int nirrt_search_goal_parent(nirrt_tree *t, int64_t *idx, double *path_len);
/* find_best_path_solution (irrt_star_2d.py:84-97): *x_best = -1 and *c_best = inf if no solution */
int nirrt_best_solution(nirrt_tree *t, double *c_best, int64_t *x_best);
/* `self.path_solutions` */
int nirrt_solutions(nirrt_tree *t, int64_t *n_sol, int64_t *out, int64_t cap);

/* IRRTStar.init (irrt_star_2d.py:35-40 / irrt_star_3d.py:32-36): informed-sampling frame computed by
 * the caller exactly like the reference (math.hypot, numpy SVD): c_min, x_center (dim), C (3x3 row-major).
 * Only needed before nirrt_run with in-kernel IRRT* sampling. */
int nirrt_set_informed(nirrt_tree *t, double c_min, const double *x_center, const double *C);
/* ... for a batch (same device): c_min (n,), x_center (n, 3), C (n, 9) - one copy, one launch */
int nirrt_set_informed_batch(nirrt_tree *const *trees, int32_t n_trees, const double *c_min, const double *x_center, const double *C);

/* NIRRT* guidance state for nirrt_run + NIRRT_F_PNG: the predicted path points `self.path_point_cloud_pred`
 * ((n, dim) f64), pc_sample_rate, pc_update_cost_ratio and c_update (nirrt_star_png_2d.py:56-63,99-130).  The
 * kernel returns status NIRRT_E_CLOUD as soon as c_best < ratio * c_update: the caller runs
 * update_point_cloud (PointNet++), calls nirrt_set_cloud again with c_update = c_best and resumes. */
int nirrt_set_cloud(nirrt_tree *t, int64_t n, const double *pts, double sample_rate, double update_cost_ratio, double c_update);

/* ---- one whole iteration --------------------------------------------------------------------- */
/* Loop body of RRTStar2D.planning (rrt_star_2d.py:37-55) / IRRTStar2D.planning
 * (irrt_star_2d.py:54-73) given node_rand: nearest -> steer -> edge collision -> insert -> Near ->
 * choose_parent -> rewire [-> InGoalRegion].  Steer runs on the device: 2D with glibc 2.35's atan2 / cos / sin restated
 * (csrc/glibc235_libm.inc: bit-identical to the reference on an x86-64 FMA host with that libm - nirrt_libm_probe lets a caller
 * check its own host), 3D with IEEE operations only; a NaN result ends the call with NIRRT_E_LIBM. */
int nirrt_step(nirrt_tree *t, const double *node_rand, uint32_t flags, nirrt_step_result *res);
/* Same body with the steer done by the caller (for hosts whose libm is not the restated one): nearest_idx from
 * nirrt_nearest, node_new = new_state(node_nearest, node_rand) computed on the host. */
int nirrt_extend(nirrt_tree *t, int64_t nearest_idx, const double *node_new, uint32_t flags, nirrt_step_result *res);

/* ---- device-resident loop over many trees ----------------------------------------------------- */
/* One persistent workgroup per tree runs `iters` loop bodies back to back without leaving the GPU.
 *   samples != NULL : node_rand is replayed from samples[(i*iters + k)*dim ...] (host, (n_trees,
 *                     iters, dim) f64) - valid whenever sampling does not depend on the tree
 *                     (RRT*: SampleFree, rrt_base_2d.py:46-52).
 *   samples == NULL : sampling happens in the kernel (SampleFree / SampleInformedSubset,
 *                     irrt_star_2d.py:99-151, irrt_star_3d.py:95-158) and consumes raw MT19937
 *                     32-bit outputs of the two generators the reference uses, exactly as
 *                     numpy's legacy RandomState / CPython's `random` would.
 *                     np_words == NULL (the normal case): the outputs come from the trees' own generators
 *                       (nirrt_set_generators), twisted and tempered in the kernel; a stream never runs dry
 *                       (NIRRT_E_STREAM only if ONE draw rejects 2^22 outputs: free space empty).
 *                     np_words != NULL: outputs produced by the caller -
 *                       np_words[i] : stream of numpy's global RandomState  (n_np per tree)
 *                       py_words[i] : stream of python's `random` module    (n_py per tree)
 *                     On return np_used[i] / py_used[i] say how many words each tree consumed.
 * cost_trace (optional, (n_trees, iters) f64): entry k = best cost AFTER iteration k, which is what
 *   planning_random's path_len_list holds at index k in both planners -
 *   F_IRRT: find_best_path_solution (irrt_star_2d.py:239-241 after its [1:] shift);
 *   F_GOAL_SCAN: get_path_len(extract_path(search_goal_parent())) (rrt_star_2d.py:223-229).
 * iters_done[i] < iters only if a word stream ran dry or a capacity was hit (status[i] != 0). */
typedef struct nirrt_run_args {
    uint32_t struct_size;     /* = sizeof(nirrt_run_args) of the caller's header (anything else: NIRRT_E_ARG) */
    uint32_t flags;
    int32_t inputs_on_device; /* != 0: samples / np_words[i] / py_words[i] are DEVICE pointers already resident
                                 in HBM (e.g. torch.cuda tensors); 0: host pointers, copied in by nirrt_run */
    int32_t park_limit;    /* sampling mode + NIRRT_F_PNG: > 0 = end the launch as soon as this many trees have stopped for a cloud
                            refresh (NIRRT_E_CLOUD): the others leave their loops at the next iteration boundary with status
                            NIRRT_E_PARK and their remaining budget.  A guided launch otherwise lasts as long as its slowest tree
                            while the trees whose cloud is due idle in their slots - with the 3D demo's pc_update_cost_ratio = 1.0
                            (demo_planning_3d.py:21: a refresh on EVERY improvement) most of a launch's slots were idle most of
                            the time.  0 = off.  Results never depend on it (launch boundaries never do). */
    int64_t iters;
    const double *samples;
    const uint32_t *const *np_words;
    const int64_t *n_np;
    const uint32_t *const *py_words;
    const int64_t *n_py;
    double *cost_trace;
    int64_t *np_used;
    int64_t *py_used;
    int64_t *iters_done;
    int32_t *status;
    double *kernel_ms;   /* optional: device time of the persistent kernel (hipEvent) */
    int64_t *scan_elems; /* optional (n_trees,): vertices actually streamed by the O(n) passes of this call (the Near
                            pass of iteration k also answers iteration k+1's nearest query, so a pass counts once) */
    int64_t *alg_elems;  /* optional (n_trees,): vertices the reference algorithm scans for the same iterations - n per
                            nearest_neighbor + n per find_near_neighbors - i.e. algorithmic bytes = alg_elems * dim * 8
                            (B_iter = 2*n*D*8) */
    int64_t *stats;      /* optional (n_trees, NIRRT_N_STATS): what this launch did per tree -
                            [0] slots visited by the fused nearest / Near passes, [1] bytes those visits read (28 B per slot
                            record in 2D, 36 B in 3D: packed records with the vertex index inside), [2] Near members, [3] members spilled out of LDS, [4] tree records (96 B of hops, eight
                            hops each) read by cost walks,
                            [5] rewire candidates examined, [6] vertices rewired, [7] vertices re-costed,
                            [8] goal-candidate list entries re-evaluated (12 B + one 32-byte record each), [9] vertices inserted,
                            [10] vertices passed through index rebuilds, [11] widened nearest visits,
                            [12] whole-tree visits, [13] iterations, [14] / [15] device wall clock (100 MHz ticks)
                            when the tree's loop started / ended, [16] = alg_elems, [17] sampling mode: bit pattern (IEEE double) of
                            the best cost on the tree when its loop ended - what a NIRRT_E_CLOUD stop compared with
                            update_cost_ratio * c_update, so that the host needs no extra launch per stopped tree -
                            [18] rewire rounds that re-parented something, [19] vertices re-parented one at a time
                            (candidate list larger than its LDS room), [20] device ticks the tree's loop was running (the launch may
                            share its workgroups among the trees in time slices: then [14] / [15] span the idle time in between),
                            [21] solution list entries re-evaluated (8 B each: the cached cost + Line(v, goal) that every re-costing keeps current),
                            [22..23] reserved */
    const int64_t *iters_each; /* optional (n_trees,), sampling mode: tree i runs at most iters_each[i] <= iters iterations
                            (trees of one batch resumed after stopping at different iterations, e.g. NIRRT_E_CLOUD);
                            cost_trace rows stay `iters` long */
    const int32_t *lanes_hint; /* optional (n_trees,), sampling mode: workgroup size wanted for tree i - 64, 128 or 256 lanes, 0 = the
                            batch's default.  Trees with different sizes are launched as concurrent groups on their own streams:
                            a tree known to be heavy (large Near sets) gets more lanes instead of holding up the launch on one
                            wave.  Results never depend on it. */
    int64_t slice_iters;   /* sampling mode with the trees' own generators, more trees than the GPU holds at once: the resident
                            workgroups share ALL trees round-robin in time slices of this many iterations (every tree advances at the
                            same pace, so the launch does not end with a long drain of half-empty compute units; a tree whose slice
                            is still running when its next turn comes simply keeps its workgroup - nobody waits).  0 = the library
                            chooses (iters / 48, at least 128; env NIRRT_SLICE overrides, 0 there = off), < 0 = off (one workgroup
                            per tree for the whole launch).  Results never depend on it. */
    const int32_t *run_ahead; /* optional (n_trees,), time-sliced launches: run_ahead[i] != 0 = tree i never waits for its turn - the
                            workgroup that runs it carries straight on with its next slice.  For trees KNOWN to be long (a free
                            straight start-goal segment: their iterations get slower as the tree grows): taking turns while their
                            slices are still short costs them time they cannot make up once the slices are long, and the launch
                            ends with the last of them (round 5: one such tree, 6.9 s of work spread over 9.5 s of a launch whose
                            other 8191 trees were done after 7.5 s).  Results never depend on it. */
} nirrt_run_args;
int nirrt_run(nirrt_tree *const *trees, int32_t n_trees, const nirrt_run_args *args);

/* update_point_cloud's last step (nirrt_star_png_2d.py:166-174) for a batch of stopped trees in ONE launch: tree i keeps the
 * points of its cloud whose prediction is non-zero, in order, and the policy scalars of nirrt_set_cloud.  clouds: DEVICE
 * f64, cloud i at clouds + i * cloud_stride doubles, (n_points[i], 3) row-major (<= 4096 points); pred: DEVICE bytes, row i at
 * pred + i * pred_stride; n_points, c_update: HOST; n_path_out: optional HOST out (points kept per tree). */
int nirrt_set_cloud_batch(nirrt_tree *const *trees, int32_t n_trees, const double *clouds, int64_t cloud_stride,
                          const int32_t *n_points, const uint8_t *pred, int64_t pred_stride, double sample_rate,
                          double update_cost_ratio, const double *c_update, int32_t *n_path_out);

/* debug aid: per-phase device tick counters of the loop body (zeros unless the library was built
 * with -DNIRRT_PROFILE); out24 = int64[24] */
int nirrt_debug_prof(nirrt_tree *t, int64_t *out24);

#ifdef __cplusplus
}
#endif
#endif /* NIRRT_HIP_H */
