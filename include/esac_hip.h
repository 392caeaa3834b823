/*
 * esac_hip.h -- C ABI of the MI355X-native ESAC hypothesis/inlier hot path.
 *
 * This is the drop-in boundary: a plain-C shared library (libesac_hip.so) with
 * no torch / pybind types in any signature.  It replaces what the reference
 * binds through pybind11 in code/esac/esac.cpp:513-516
 *     m.def("forward",  &esac_forward)   // esac.cpp:64-190
 *     m.def("backward", &esac_backward)  // esac.cpp:213-511 -> esac_hip_backward
 * The Python module `esac` (esac.py -> esac_amd/api.py) binds these entry points
 * with ctypes and keeps the reference's positional `esac.forward(...)` signature;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; esac_hip_last_error()
 *     returns a thread-local message (the Python layer raises RuntimeError,
 *     the exception type pybind11 turns the reference's c10::Error into).
 *   - pointers named d_* are DEVICE pointers on the context's GPU, h_* are HOST.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - tensors are dense row-major: scene coordinates [E,3,H,W] float32
 *     (esac_types.h:45 coord_t), hypothesis assignment [N] int64
 *     (esac_types.h:46 hyp_assign_t).
 *   - there is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef ESAC_HIP_H
#define ESAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESAC_HIP_ABI_VERSION 6

/* reference compile-time constants (esac.cpp:44-45) */
#define ESAC_MAX_SAMPLING_TRIES 1000000
#define ESAC_MAX_REF_STEPS 100
#define ESAC_MAX_BATCH 1024 /* frames per esac_hip_forward_batch call */

/* Scalar arguments of esac_forward (esac.cpp:64-77) plus the knobs the
 * reference hard-codes or hides (RNG key, limits, multi-GPU shard offset). */
typedef struct esac_hip_params {
    int32_t E, H, W;          /* sceneCoordinates.size(0,2,3)  esac.cpp:87-88      */
    int32_t N;                /* hypAssignment.size(0)         esac.cpp:90         */
    int32_t shift_x, shift_y; /* esac.cpp:67-68                                    */
    float focal, ppx, ppy;    /* esac.cpp:69-71; camMat is float, esac.cpp:93-97   */
    float inlier_thresh;      /* tau,   esac.cpp:72                                */
    float inlier_alpha;       /* alpha, esac.cpp:73                                */
    float inlier_beta;        /* beta,  esac.cpp:74                                */
    float max_reproj;         /* esac.cpp:75                                       */
    int32_t sub_sampling;     /* esac.cpp:76                                       */
    uint64_t seed;            /* RNG key; reference: fixed 1305, thread_rand.h:103 */
    uint64_t call;            /* call counter (reference RNG state persists across calls) */
    int32_t max_tries;        /* <=0 -> ESAC_MAX_SAMPLING_TRIES                    */
    int32_t max_ref_steps;    /* <0  -> ESAC_MAX_REF_STEPS                         */
    int32_t hyp_offset;       /* global index of local hypothesis 0 (multi-GPU sharding; RNG and
                                 tie-breaks use global indices so results do not depend on the
                                 number of ranks) */
    float rescore_margin;     /* fast-score band re-scored exactly; <=0 -> alpha*(ESAC_DEFAULT_MARGIN + 2/(H*W)) */
    const int32_t* d_hyp_index; /* optional DEVICE int32[N]: global index of each local hypothesis, for
                                 shards that are not a contiguous range (expert-sharded multi-GPU);
                                 NULL -> hyp_offset + i */
    int32_t flags;            /* ESAC_FLAG_* below, 0 = default */
    int32_t expert_base;      /* added to the winner's (local) expert index in ESAC_RES_EXPERT: a rank that holds only a
                                 slice of the experts' maps (multi-GPU, esac_hip_shard_balanced) passes the global id of
                                 its first map so that the record carries what esac.cpp:189 returns; 0 otherwise */
} esac_hip_params;

/* esac_hip_forward / _batch: score EVERY hypothesis in the reference's mixed float/double arithmetic
 * (esac_util.h:235-260, 292-360) instead of ranking with the fp32 stream and re-scoring only the contenders.
 * The score vector, ESAC_RES_PROB and ESAC_RES_ENTROPY are then the reference's own values (softMax / entropy,
 * esac_util.h:461-497) for every hypothesis; the winner and the pose are the same either way. */
#define ESAC_FLAG_EXACT_SCORES 1
/* Shape of the fp32 ranking score (results are the same to fp32 rounding; default: chosen from grid size and N).
 * TILED: map tiles stationary in registers, hypotheses bucketed by expert stream past them (large maps);
 * STREAM: one hypothesis per workgroup streams its expert's whole map (small, cache-resident maps).
 * Both rank only.  One documented difference from the reference's projection (esac_util.h:302-305, `z ? 1/z : 1`): a cell
 * whose camera-frame depth is EXACTLY 0 in fp32 gets the clamped error maxReproj in the TILED stream (rcp(0) = inf)
 * instead of the x,y-as-is projection; the exact re-score of the contenders and the refinement follow the reference. */
#define ESAC_FLAG_SCORE_TILED 2
#define ESAC_FLAG_SCORE_STREAM 4
/* Sampling reads the maps through a packed (x,y,z,0)-per-cell copy made at the start of the call (default: only when
 * the maps are far larger than the caches and several experts are in play).  Results are unchanged. */
#define ESAC_FLAG_PACK_MAPS 8
/* Sampling without the screen: every try of every hypothesis is solved and decided by the fp64 route (the reference's own
 * loop, esac_util.h:152-223, try by try).  The default route screens the tries of long searches (wrong-expert hypotheses)
 * with a one-sided fp32 test first and decides only what the screen cannot rule out; the accepted
 * try is the same either way -- this flag is the guaranteed route, several times slower on such hypotheses, like
 * ESAC_FLAG_EXACT_SCORES for the scores. */
#define ESAC_FLAG_EXACT_SAMPLING 16
/* d_scores_out is indexed by GLOBAL hypothesis index (d_hyp_index[i], or hyp_offset + i) instead of by local position:
 * a multi-GPU shard writes its scores straight into its slots of the exchange buffer. */
#define ESAC_FLAG_SCORES_BY_INDEX 32
/* Take the two guaranteed routes (ESAC_FLAG_EXACT_SCORES | ESAC_FLAG_EXACT_SAMPLING) WHERE THEY ARE FREE: a single frame of ONE
 * expert with N * H * W <= 2^21 (BASELINE configs[0] / [1]: 64 or 256 hypotheses on the 60x80 grid).  There the exact score of
 * every hypothesis is one ~10 us launch on the whole chip, its softmax statistics run in the refinement kernel's prologue and
 * the sampler has no screened stage anyway: the call costs what the default route costs (bench.py: `value` runs with this flag,
 * `value_fast` without it) and the score vector, ESAC_RES_PROB and ESAC_RES_ENTROPY are the reference's own values.  Everywhere
 * else the flag changes nothing (several experts, many hypotheses, large maps, batches: the fp32 ranking stream + exact re-score of
 * the contenders is several times faster there).  esac.forward() of the Python module sets it unless told otherwise. */
#define ESAC_FLAG_AUTO_EXACT 64
#define ESAC_AUTO_EXACT_MAX_WORK (1 << 21) /* N * H * W up to which ESAC_FLAG_AUTO_EXACT applies */
/* The winner is refined by ONE workgroup whatever the shape (no team on one XCD, no cooperating workgroups): the route that
 * needs no co-residency of several workgroups.  What a blocking call falls back to by itself after a team time-out; a caller of
 * ASYNCHRONOUS calls (the multi-GPU exchange: esac_hip_pick_record returned -12) sets it to run the frame again. */
#define ESAC_FLAG_REFINE_SOLO 128

#define ESAC_DEFAULT_MARGIN 1e-3f

/* Layout of the result record (doubles).  Written by the refinement kernel. */
enum {
    ESAC_RES_SCORE = 0,      /* exact soft-inlier score of the winner (esac_util.h:235-260)     */
    ESAC_RES_HYP = 1,        /* winner hypothesis index (GLOBAL = hyp_offset + local)            */
    ESAC_RES_EXPERT = 2,     /* hypAssignment[winner]  (esac.cpp:189, the return value)          */
    ESAC_RES_RVEC = 3,       /* refined rvec[3], tvec[3] (scene->camera, OpenCV convention)      */
    ESAC_RES_TVEC = 6,
    ESAC_RES_POSE = 9,       /* 16 values: float(inverse([R t;0 1])) row-major (esac.cpp:182-187) */
    ESAC_RES_REF_STEPS = 25, /* accepted re-fits in refineHyp (esac_util.h:396-452)              */
    ESAC_RES_INLIERS = 26,   /* inlier count of the last accepted set                            */
    ESAC_RES_PROB = 27,      /* softmax probability of the winner (esac.cpp:157)                 */
    ESAC_RES_ENTROPY = 28,   /* entropy of the hypothesis distribution (esac.cpp:158)            */
    ESAC_RES_CONTENDERS = 29,/* how many hypotheses were re-scored exactly                       */
    ESAC_RES_LM_ITERS = 30,  /* total LM iterations spent in refinement                          */
    ESAC_RES_VALID = 31,     /* d_result_out only: 1.0 once a record has been written (multi-GPU exchange buffers
                                are zero-padded; the host copy carries the inlier-map buffer index here instead);
                                3.0: the workgroups sharing the refinement timed out -- no record, and not an empty shard */
    ESAC_RES_DOUBLES = 32
};

/* Stage buffers that can be read back / written for stage-wise parity tests. */
enum {
    ESAC_BUF_HYPS = 0,       /* double[N][6]  rvec,tvec per hypothesis (esac_types.h:40 pose_t)  */
    ESAC_BUF_SAMPLE_XY = 1,  /* int32[N][4][2] sampled cells (esac_util.h:187 sampledPoints)     */
    ESAC_BUF_TRIES = 2,      /* int32[N] accepted try index, -1 = budget exhausted               */
    ESAC_BUF_SCORES = 3,     /* double[N] scores (fp32-path value, exact for re-scored ones)     */
    ESAC_BUF_RESULT = 4,     /* double[ESAC_RES_DOUBLES]                                          */
    ESAC_BUF_INLIER_MAP = 5, /* uint8[H*W] last accepted inlier set (esac_util.h:440 inlierMap)  */
    ESAC_BUF_INLIER_COUNTS = 6, /* int32[ESAC_MAX_REF_STEPS+1] inlier count seen at each step    */
    ESAC_BUF_WINNER_ERRS = 7,   /* float[H*W] reprojection errors of the refined pose; only kept after
                                   esac_hip_set_debug(ctx, ESAC_DEBUG_ERROR_IMAGE)               */
    ESAC_BUF_EXACT_FLAGS = 8,   /* uint8[N] 1 where ESAC_BUF_SCORES holds an exact re-score      */
    ESAC_BUF_CYCLES = 9,        /* int64[32] shader-cycle counters of the refinement kernel (profiling) */
    /* stage outputs of the most recent esac_hip_backward */
    ESAC_BUF_BWD_PROBS = 10,       /* double[N]   selection probabilities (softmax of the exact scores)      */
    ESAC_BUF_BWD_LOSSES = 11,      /* double[N]   pose loss of every (refined) hypothesis                    */
    ESAC_BUF_BWD_REF_HYPS = 12,    /* double[N,6] refined poses (initial pose where p < PROB_THRESH)          */
    ESAC_BUF_BWD_SCORE_GRADS = 13, /* double[N]   d expected loss / d score                                  */
    ESAC_BUF_BWD_SLOTS = 14,       /* int32[N]    hypothesis index per slot, first h_out[1] entries valid    */
    ESAC_BUF_BWD_SLOT_INFO = 15,   /* int32[min(N,ESAC_BWD_MAX_SLOTS),4] per slot: accepted map buffer (-1 none),
                                      inliers of the last accepted step, accepted steps, LM iterations        */
    ESAC_BUF_BWD_DLOSS = 16,       /* double[min(N,ESAC_BWD_MAX_SLOTS),6] d loss / d refined pose per slot    */
    ESAC_BUF_BWD_PATH1 = 17,       /* double[k,3,H,W] gradient slabs of the first k slots, path I (unweighted)  */
    ESAC_BUF_BWD_PATH2 = 18,       /* double[k,3,H,W] the same for path II; k = bytes / (3*H*W*8) <= #slots     */
    ESAC_BUF_BWD_TEAM_INFO = 20,   /* int32[4] training path: [0] 1 when the slots of the most recent esac_hip_backward were refined by
                                      teams of 8 workgroups (one XCD each) instead of one workgroup per slot -- blocking calls
                                      that THEMSELVES select <= 32 hypotheses (decided on the device), grids of 1024..8192 cells; [1] calls that issued the team launch so
                                      far; [2] of them, calls in which a team timed out and the slots were refined again by one
                                      workgroup each (teams stay off on this context afterwards); [3] slots of the last call */
    ESAC_BUF_SPEC_INFO = 21,       /* int32[4] speculative forward (ESAC_DEBUG_NO_SPECULATION): [0] forward calls on this context that took the
                                      speculative route, [1] those among them whose speculation failed (the refinement ran again for
                                      the true winner), [2] 1 when the most recent forward call was speculative, [3] 1 when its
                                      speculation failed */
    ESAC_BUF_SPEC_FLAGS = 22,      /* uint8[N] after a speculative call: 1 where the sampler's first pass left the hypothesis to the straggler chain */
    ESAC_BUF_REFINE_INFO = 19      /* int32[8] how the most recent winner refinement ran (refineHyp, esac_util.h:378-454):
                                      [0] 0 one workgroup, 1 cooperating workgroups (grids beyond one LDS list), 2 a team on
                                      one XCD (small grids); [1] workgroups sharing it; [2] XCD census of a team: byte x = members
                                      that ran on XCD x, x = 0..3, [7] the same for XCDs 4..7; [3] 1 when all members shared one
                                      XCD; [4] exchanges between them; [5] 1 when an exchange timed out; [6] bits 0..29: blocking calls on this
                                      context so far whose team timed out and were refined again by one workgroup, bit 30: the
                                      context has stopped asking for teams (two time-outs in a row; esac_hip_set_refine_team)   */
};

/* Hypotheses that take part in the training expectation: selection probability >= PROB_THRESH = 0.001
 * (esac_derivative.h:33), so never more than 1000 whatever N is. */
#define ESAC_BWD_MAX_SLOTS 1000

typedef struct esac_hip_ctx esac_hip_ctx;

/* library / device */
int esac_hip_abi_version(void);
const char* esac_hip_last_error(void);
int esac_hip_device_count(void);

/* A context owns the device workspaces for one GPU; not thread-safe, one call in
 * flight per context (the reference extension is not re-entrant either:
 * static RNG, thread_rand.cpp:4-5). */
int esac_hip_create(esac_hip_ctx** ctx, int device);
int esac_hip_destroy(esac_hip_ctx* ctx);

/*
 * The whole of esac_forward (esac.cpp:64-190) on the device:
 *   sample+P3P -> soft-inlier scores -> select (+exact re-score of the contenders)
 *   -> refine the winner -> pose.
 * d_scene_coords  [E,3,H,W] float32, d_hyp_assign [N] int64 (device).
 * d_scores_out    optional device double[N]  (the score vector; all-reduced across ranks by the caller)
 * d_result_out    optional device double[ESAC_RES_DOUBLES]
 * h_result_out    optional host   double[ESAC_RES_DOUBLES]; when non-NULL the call blocks until the
 *                 refinement kernel has delivered the record (it stores it into pinned host memory and
 *                 the host polls an epoch word: no copy kernel, no completion-signal round trip);
 *                 the reference call is blocking too.
 */
int esac_hip_forward(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                     const esac_hip_params* p, void* stream, double* d_scores_out,
                     double* d_result_out, double* h_result_out);

/*
 * B frames in ONE set of launches (new, beside the drop-in call): one blocking call per frame
 * occupies a single CU for its refinement tail, so independent frames are what fills the other 255.
 * d_scene_coords  frame b at d_scene_coords + b * sc_frame_stride (elements; 0 = every frame uses the same maps),
 *                 each [E,3,H,W]; d_hyp_assign [B,N]; p describes ONE frame (N = hypotheses per frame).
 * Frame b draws the RNG streams of call p->call + b: the batch equals B sequential esac_hip_forward calls with consecutive
 * call counters -- every discrete output identical, the poses bit for bit when the single calls refine on the batch's route
 * (esac_hip_set_refine_team(ctx, 8): a batch's teams are 8 per frame, a single call's default team is chosen per grid), else to
 * the rounding of the LM sums (~1e-9).
 * Outputs are frame-major: d_scores_out [B,N], d_result_out / h_result_out [B,ESAC_RES_DOUBLES].
 */
int esac_hip_forward_batch(esac_hip_ctx* ctx, int B, const float* d_scene_coords, int64_t sc_frame_stride,
                           const int64_t* d_hyp_assign, const esac_hip_params* p, void* stream,
                           double* d_scores_out, double* d_result_out, double* h_result_out);

/*
 * Multi-GPU exchange (new; the reference has no multi-device path): d_records = `world` result records of
 * ESAC_RES_DOUBLES doubles each (device), e.g. the tail of the all-reduced buffer [N scores | world records] with
 * all-zero records for ranks without hypotheses (ESAC_RES_VALID marks real ones).  Picks the global winner -- highest
 * exact score, lowest global hypothesis index on ties (esac_util.h:519) -- on the device and delivers it to
 * h_record_out (blocking, through pinned memory).  -11: no rank contributed a record.  -12: the refinement team of at least one
 * rank timed out (ESAC_RES_VALID = 3 in its record): no winner is declared without that rank's candidate; every rank reads the
 * same records and gets the same status, so all of them run the frame again with ESAC_FLAG_REFINE_SOLO (esac_amd/distributed.py).
 * d_zero / n_zero: optional (NULL / 0) device doubles the same launch sets to 0 -- a caller that alternates between two
 * exchange buffers hands over the one the NEXT call will use, so that no call starts with a memset of its own.
 */
int esac_hip_pick_record(esac_hip_ctx* ctx, const double* d_records, int world, void* stream, double* h_record_out, double* d_zero,
                         int n_zero);

/*
 * The ONE collective of the multi-GPU path -- all-reduce(SUM) of the exchange buffer [N scores | world x 32-double records] over
 * RCCL / xGMI -- issued by the library itself on the launch stream (new).  esac_hip_comm_unique_id: rank 0 makes an RCCL unique
 * id (ESAC_COMM_ID_BYTES bytes) and hands it to every rank by whatever means the caller has (esac_amd/distributed.py: one
 * broadcast over the torch.distributed process group, once); esac_hip_comm_init: every rank joins with it (collective,
 * blocking); esac_hip_allreduce_sum: in place on `stream`, asynchronous.  -13: no communicator on this context.  RCCL is bound
 * when the first of these is called (dlopen of librccl.so.1: the copy the process already holds, else ROCm's), not at link time:
 * a single-GPU process never loads it; -14: it could not be loaded.  -300 - r: RCCL returned ncclResult_t r.
 */
#define ESAC_COMM_ID_BYTES 128
int esac_hip_comm_unique_id(void* out_id, size_t bytes);
int esac_hip_comm_init(esac_hip_ctx* ctx, int nranks, int rank, const void* unique_id, size_t bytes);
int esac_hip_comm_destroy(esac_hip_ctx* ctx);
int esac_hip_allreduce_sum(esac_hip_ctx* ctx, double* d_buf, size_t count, void* stream);
/* What the context's communicator ITSELF reports: out[0] = ncclCommCount, out[1] = ncclCommUserRank, out[2] = ncclCommCuDevice
 * (-1 when the loaded RCCL lacks the query), out[3] = the context's GPU.  -13: no communicator.  (bench.py: `ranks_seen`.) */
int esac_hip_comm_info(esac_hip_ctx* ctx, int32_t out[4]);

/*
 * Load-balanced multi-GPU shard, built on the device (new; SURVEY.md 8e: "a load-balanced assignment from the
 * hypAssignment histogram", test_esac.py:178).  The hypotheses are ordered by (expert, index) -- the stable counting sort
 * of d_hyp_assign [N] -- and rank r of `world` takes the sorted positions [r*N/world, (r+1)*N/world) (remainder to the
 * first ranks): every rank gets N/world hypotheses (+-1) whatever the gating distribution is, its hypotheses belong to a
 * CONTIGUOUS range of experts [first, last], and only the experts at the two ends of that range are shared with a
 * neighbour (their maps are needed on both ranks).  The plan is a pure function of the assignment vector, so every rank
 * computes the same one without communication.
 * d_index_out   int32[n_local]  global indices of this rank's hypotheses (pass as esac_hip_params.d_hyp_index)
 * d_assign_out  int64[n_local]  their experts minus `expert_base` (pass as d_hyp_assign of the forward call; values
 *               outside [0,E) are copied unchanged so that the forward call still reports them)
 * d_info_out    optional int32[4]: first expert, last expert, n_local, 1 if a value was outside [0,E)
 * n_local = N/world (+1 for rank < N % world) is known to the caller.  expert_base: 0 when the forward call sees all E
 * maps; the rank's first expert when it sees only the maps [first, last] (then also esac_hip_params.expert_base).
 * One launch on `stream`, asynchronous.  E <= 4096.
 */
int esac_hip_shard_balanced(esac_hip_ctx* ctx, const int64_t* d_hyp_assign, int N, int E, int world, int rank,
                            int expert_base, void* stream, int32_t* d_index_out, int64_t* d_assign_out, int32_t* d_info_out);

/*
 * esac_backward (esac.cpp:213-520): expected pose loss over the hypothesis distribution and its gradient wrt the
 * scene coordinates, everything on the device.
 * Status -10: hypAssignment held a value outside [0,E) (the reference reads out of bounds there).
 * d_out_gradients [E,3,H,W] float32 (device), ACCUMULATED into (`+=`, esac.cpp:491-508) -- the caller zeroes it,
 *                 as train_esac.py:176 does.
 * h_gt_pose       host float[16], the ground-truth camera pose (4x4 row-major; gtPose, esac.cpp:219).
 * w_loss_rot / w_loss_trans / loss_cut: wLossRot, wLossTrans, lossCut (esac.cpp:220-222).
 * h_out           optional host double[4]: expected loss (the return value of esac_backward), number of
 *                 hypotheses with p >= PROB_THRESH, entropy of the distribution, 1 if an assignment was out of range.  When non-NULL the call
 *                 blocks on `stream` (the reference call is blocking); otherwise it is asynchronous.
 * Uses the same Philox streams as esac_hip_forward for (seed, call): the hypotheses of a backward call are the
 * hypotheses of the forward call with the same counter.  Hypothesis sharding (hyp_offset / d_hyp_index) is
 * rejected: the softmax expectation needs every hypothesis on one device.
 */
int esac_hip_backward(esac_hip_ctx* ctx, const float* d_scene_coords, float* d_out_gradients,
                      const int64_t* d_hyp_assign, const float* h_gt_pose, float w_loss_rot, float w_loss_trans,
                      float loss_cut, const esac_hip_params* p, void* stream, double* h_out);

/* The same phases one at a time (asynchronous on `stream`), for stage-wise parity
 * tests and for callers that interleave other work.  Order: sample, score, select, refine. */
int esac_hip_sample(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                    const esac_hip_params* p, void* stream);   /* esac_util.h:129-225 */
int esac_hip_score(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                   const esac_hip_params* p, void* stream);    /* esac.cpp:131-147   */
int esac_hip_select(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                    const esac_hip_params* p, void* stream);   /* esac.cpp:153-155   */
int esac_hip_refine(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                    const esac_hip_params* p, void* stream);   /* esac.cpp:167-187   */

/* Exact (reference-arithmetic) scoring of every hypothesis; slow path used by tests
 * and by callers that want scores identical to esac_util.h:235-260 for all N. */
int esac_hip_score_exact(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                         const esac_hip_params* p, void* stream);

/* Waits for the device; -10 when the most recent SAMPLING launch on this context (esac_hip_forward / _batch / _sample /
 * _backward) met a hypAssignment value outside [0,E), -12 when the workgroups sharing its most recent winner refinement
 * could not synchronise (blocking calls report the former themselves and, when a small-grid team timed out, run the
 * refinement again in one workgroup; asynchronous ones -- no host result pointer -- cannot: their device record then
 * carries ESAC_RES_VALID = 3 and this function counts the time-out towards the same two-in-a-row latch), else 0.
 * Out-of-range values never cause an out-of-bounds read: such hypotheses are evaluated against expert 0. */
int esac_hip_check(esac_hip_ctx* ctx);

/* Stage buffer access (synchronous). `bytes` must match the buffer size for (which, N, H, W). */
int esac_hip_read(esac_hip_ctx* ctx, int which, void* h_dst, size_t bytes);
int esac_hip_write_hyps(esac_hip_ctx* ctx, const double* h_hyps, int N);

/* Time of the most recent launch of each phase on this context in milliseconds
 * (hipEvents around each kernel; the StopWatch prints of esac.cpp:124,149,161,179).
 * out[0..4] = sample, score, select+rescore, refine, total; out[5] = an empty event interval (the
 * bracketing overhead contained in each of the four phase figures).  Synchronises on the last event. */
int esac_hip_phase_ms(esac_hip_ctx* ctx, float out[6]);
/* Mean duration of the score kernel itself over every launch since timing was enabled, measured on the
 * device: max(end) - min(start) over its workgroups on the constant 100 MHz wall clock -- the figure
 * rocprofv3's kernel trace reports (hipEvents around one ~3 us launch also contain the launch gap).
 * Synchronises the device. */
int esac_hip_score_span_ms(esac_hip_ctx* ctx, float* mean_ms, int* launches);
/* Mean GPU time (ms) of each stage -- sample, score, select(+re-score), refine -- for this input: the chain runs once,
 * then each stage is launched `reps` times back to back between one pair of hipEvents on `stream` (includes the ~1.5 us
 * dependent-kernel boundary per launch, excludes the per-event overhead that brackets around single launches carry).
 * Blocking.  out_ms[0..3]. */
int esac_hip_time_stages(esac_hip_ctx* ctx, const float* d_scene_coords, const int64_t* d_hyp_assign,
                         const esac_hip_params* p, void* stream, int reps, float out_ms[4]);
/* Where the HOST's time of the most recent blocking esac_hip_forward / _batch on this context went (CLOCK_MONOTONIC, always
 * recorded: six clock reads, ~0.15 us): out_ns[0..5] = nanoseconds after entry at which the argument block was ready, the
 * sampling / score / refinement launch calls had returned, the result record had landed in pinned memory, and the call
 * returned; out_ns[6] = the entry time itself (absolute, for the caller's own stamps around the call); out_ns[7] unused.
 * What remains of a step beyond the kernels' own durations -- the "host turn" -- is [1] + the gap to the first kernel's
 * start + ([5] - [4]) + the caller's own time between calls (scripts/dev/host_turn.py puts them side by side). */
int esac_hip_host_turn(esac_hip_ctx* ctx, double out_ns[8]);
/* The same stamps as MEANS over the blocking esac_hip_forward / _batch calls since the last reset (running sums kept by the
 * library: the caller's timed loop is not disturbed by collecting them): out_ns[0..5] as above, out_ns[6] = mean time between the
 * return of one call and the entry of the next (the caller's own share of a step), out_ns[7] = number of calls averaged.
 * reset != 0 clears the sums afterwards. */
int esac_hip_host_turn_mean(esac_hip_ctx* ctx, double out_ns[8], int reset);
/* enable/disable the per-phase events (off by default: zero overhead).  enabled = k > 1 samples every k-th forward
 * call only, starting with the next one (the events themselves cost GPU time: an empty pair reads ~5 us). */
int esac_hip_set_timing(esac_hip_ctx* ctx, int enabled);
/* debug options (off by default). ESAC_DEBUG_ERROR_IMAGE: the refinement also stores the reprojection-error image of
 * the pose it is refining (ESAC_BUF_WINNER_ERRS); nothing downstream needs it, so the stores are skipped otherwise. */
#define ESAC_DEBUG_ERROR_IMAGE 1
/* ESAC_DEBUG_COOP_STALL (tests only): the cooperating refinement workgroups of a large-grid call wait for one workgroup
 * more than was launched, with a short spin limit -- exercises the "not all workgroups became resident" failure path
 * (status -12) without having to occupy the GPU. */
#define ESAC_DEBUG_COOP_STALL 2
/* ESAC_DEBUG_TEAM_SPREAD (tests only): the members of a refinement team are launched as CONSECUTIVE workgroups, which the
 * hardware places on different XCDs -- the placement the team's exchange must survive (it is correct at any placement,
 * only slower across XCDs); ESAC_BUF_REFINE_INFO[3] then reads 0. */
#define ESAC_DEBUG_TEAM_SPREAD 4
/* ESAC_DEBUG_NO_SPECULATION (tests, A/B measurements): esac_hip_forward runs its kernels strictly one after the other on the
 * caller's stream.  By default a single-frame call with several experts and at most 8192 hypotheses (the screened sampling
 * route, the fp32 ranking stream, grids below 32768 cells) is SPECULATIVE: once the sampler's first pass (32 tries per
 * hypothesis) has settled the hypotheses of usable experts, those are scored and the best of them (fp32 ranking) is refined on
 * the caller's stream, while on two streams the context owns the sampler's straggler chain -- the wrong-expert hypotheses that
 * need ~10^3 tries each and practically never win -- with the stragglers' scores, and the selection among the settled hypotheses
 * (band, exact re-scores) run beside that refinement; a join kernel then completes the selection over all hypotheses (band,
 * re-scores, softmax statistics, argmax: the serial route's arithmetic statement by statement) and delivers the record, or, when
 * the winner is not the hypothesis that was refined, has the refinement run again (a second launch, enqueued with the call on the
 * caller's stream and gated on the join's verdict: work the caller enqueues behind the call sees the final outputs).  Every
 * output -- poses, scores, flags, statistics, the record -- is what the serial order produces (ESAC_BUF_SPEC_INFO counts). */
#define ESAC_DEBUG_NO_SPECULATION 8
/* ESAC_DEBUG_SPEC_SECOND_BEST (tests only): the speculative refinement starts from the SECOND-best settled hypothesis of the fp32
 * ranking instead of the best -- the situation in which the fp32 stream and the reference arithmetic order two near-equal scores
 * differently.  The join must find that the refined hypothesis is not the winner and the gated second refinement must deliver
 * the serial route's outputs. */
#define ESAC_DEBUG_SPEC_SECOND_BEST 16
/* ESAC_DEBUG_SPEC_LOSE_CHAIN (tests only): the last launch of the straggler chain's stream -- the one that reports "the chain is
 * done" -- is left out, as if it had failed.  The join's wait is bounded (20 ms): it must report status 5, a blocking call must run
 * again in stream order and return that route's outputs, and the context must stop speculating; an asynchronous call's device
 * record carries ESAC_RES_VALID = 3 (esac_hip_pick_record: -12). */
#define ESAC_DEBUG_SPEC_LOSE_CHAIN 32
int esac_hip_set_debug(esac_hip_ctx* ctx, int flags);

/* The winner's refinement (refineHyp, esac_util.h:378-454) on a single frame whose grid fits one workgroup's LDS list
 * (1024 <= H*W <= 32768 cells) is shared by a TEAM of workgroups on one XCD, each owning a slice of the cells: `members` of
 * them (2..ESAC_REFINE_TEAM_MAX = the CUs of an XCD; default ESAC_REFINE_TEAM_DEFAULT: measured 8 = 19 > 10, 12, 16 at 60x80),
 * never more than give every lane of a member one cell (ceil(H*W / 256)) and never fewer than a member's lanes can hold
 * the cells of (ceil(H*W / 1024): four per lane); 0 or 1: one workgroup refines, as on every other shape.
 * Results do not depend on the setting beyond the rounding of the LM sums (every discrete output is identical).
 * esac_hip_forward_batch with up to 32 frames gives every frame's winner a team of 8 (all 32 teams are resident together);
 * larger batches refine with one workgroup per frame.
 * A member waits at most 1 ms (wall clock) for the others at an exchange: a team whose members do not all become resident
 * (a shared / partitioned GPU) gives up, the blocking call refines again in ONE workgroup and returns that result; two such
 * calls in a row and the context stops asking for teams (ESAC_BUF_REFINE_INFO[6] bit 30) -- every call would pay the wait
 * first -- until 1000 further calls have passed or this function is called again (it re-arms the training path's slot
 * teams too). */
#define ESAC_REFINE_TEAM_MAX 32
#define ESAC_REFINE_TEAM_DEFAULT 8
/* The default POLICY (a fresh context; esac_hip_set_refine_team(ctx, ESAC_REFINE_TEAM_AUTO) restores it): ESAC_REFINE_TEAM_DEFAULT
 * members, or -- round 6 -- the smallest team of at most 16 that lowers the cells a lane holds: every lane of a member walks as
 * many cells per pass as its fullest lane owns, eight members of 600 cells hold three per lane on the 60x80 grid, TEN of 480
 * hold two (a pass 0.3 us shorter, the call 2.5 us; teams of up to 16 exchange without an LDS stage).  Batches and the training
 * path's slot teams stay at 8 per team (32 teams of 8 are the chip's 256 CUs).  A number asks for exactly that many. */
#define ESAC_REFINE_TEAM_AUTO (-1)
int esac_hip_set_refine_team(esac_hip_ctx* ctx, int members);

/* How a blocking call waits for its result record (written by the last kernel into pinned host memory):
 * ESAC_WAIT_SPIN (default) polls the epoch word -- lowest latency, one host core busy for the ~0.2 ms of the call;
 * ESAC_WAIT_YIELD polls with sched_yield() between reads (a DataLoader-heavy caller keeps its cores);
 * ESAC_WAIT_BLOCK sleeps in hipStreamSynchronize (adds the completion-signal round trip, ~15-20 us). */
#define ESAC_WAIT_SPIN 0
#define ESAC_WAIT_YIELD 1
#define ESAC_WAIT_BLOCK 2
int esac_hip_set_wait(esac_hip_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* ESAC_HIP_H */
