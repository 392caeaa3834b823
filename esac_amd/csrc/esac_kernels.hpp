// esac_kernels.hpp -- kernel argument block shared by the launchers and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace esac {

// mirrors of the ESAC_RES_* layout in include/esac_hip.h (checked by static_assert in esac_capi.hip)
constexpr int ESAC_RES_SCORE_K = 0, ESAC_RES_HYP_K = 1, ESAC_RES_EXPERT_K = 2, ESAC_RES_RVEC_K = 3,
              ESAC_RES_POSE_K = 9, ESAC_RES_REF_STEPS_K = 25, ESAC_RES_INLIERS_K = 26, ESAC_RES_PROB_K = 27,
              ESAC_RES_ENTROPY_K = 28, ESAC_RES_CONTENDERS_K = 29, ESAC_RES_LM_ITERS_K = 30;
constexpr int ESAC_MAX_REF_STEPS_K = 100;
constexpr int ESAC_BWD_SLOTS_K = 1000;     // = ESAC_BWD_MAX_SLOTS (include/esac_hip.h)
constexpr int ESAC_REFINE_LDS_CAP = 8192;  // correspondences the refinement kernel stages in LDS (128 KiB of the CU's 160 KiB)
constexpr int ESAC_REFINE_THREADS = 256;   // 4 wavefronts = one per SIMD of the one CU a refinement occupies
constexpr int ESAC_ERR_UNROLL = 8;         // cells per lane in flight in the error pass
constexpr int ESAC_REFINE_COOP_MAX = 256;  // workgroups that may share one refinement (one per CU: all must be resident)
constexpr int ESAC_REFINE_TEAM_DEFAULT_K = 8;  // = ESAC_REFINE_TEAM_DEFAULT (include/esac_hip.h)
constexpr int ESAC_REFINE_TEAM_MAX_K = 32;   // ... on a small grid: a team on ONE XCD (its 32 CUs; esac_refine_team.hip)
constexpr int ESAC_REFINE_TEAM_MIN_CELLS = 1024;  // smaller grids are refined by one workgroup (a pass is shorter than an exchange)
constexpr int ESAC_TEAM_BATCH_MAX = 32;           // frames of a batch that are refined by teams of 8 (32 teams = the chip's 256 CUs at once)
constexpr int ESAC_TEAM_GRANULES = 2 * ESAC_REFINE_TEAM_MAX_K * 32;  // 16-byte granules of one team's exchanges: [parity][member][value]
// How long a team member waits at an exchange before it gives up, in ticks of the 100 MHz wall clock: all members become
// resident within microseconds of each other or not at all (a shared / partitioned GPU, the caller's own kernels holding
// the CUs), and a round is ~3 us -- 1 ms; the training path's slot teams drain group by group behind each other (a
// refinement is ~0.1 ms): 8 ms
constexpr long ESAC_TEAM_SPIN_LIMIT = 100000, ESAC_TEAM_SPIN_LIMIT_SLOTS = 800000;
constexpr int ESAC_PIN_DOUBLES = 36;       // pinned host slot per frame: result record [32] + epoch word + status word + check word + pad
// The pinned record is handed over WITHOUT a system-scope fence: the kernel stores the 34 words and a 35th that is a
// checksum of them (one store instruction), the host accepts a slot once its epoch word is the call's and the check word
// fits the other 34 -- whatever order the words arrive in across PCIe, a partly updated slot fails the check and is
// simply polled again.  pin_mix(bits of word k, k), XOR-ed over k = 0..33, is the check word.
__host__ __device__ inline unsigned long long pin_mix(unsigned long long bits, int k) {
    const unsigned long long h = (bits + (unsigned long long)(k + 1) * 0x9E3779B97F4A7C15ull) * 0xD6E8FEB86659FD93ull;
    const int r = k & 63;
    return r ? (h << r) | (h >> (64 - r)) : h;
}
#ifdef __HIPCC__
// A full wavefront hands 34 words (lane k < 34: word k; word 32 = the call's epoch, 33 = status) + their check word to the pinned
// slot: one store instruction, no fence (see above).
__device__ __forceinline__ void pin_deliver(double* pin, double v) {
    const int lane = threadIdx.x & 63;
    unsigned long long h = lane < 34 ? pin_mix((unsigned long long)__double_as_longlong(v), lane) : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h ^= __shfl_xor(h, o);
    if (lane < 35) pin[lane] = lane < 34 ? v : __longlong_as_double((long long)h);
}
#endif
constexpr int ESAC_FLAG_EXACT_SCORES_K = 1;  // = ESAC_FLAG_EXACT_SCORES (include/esac_hip.h)
constexpr int ESAC_FLAG_EXACT_SAMPLING_K = 16, ESAC_FLAG_SCORES_BY_INDEX_K = 32;  // = ESAC_FLAG_* (checked in esac_capi.hip)
constexpr int ESAC_SELECT_SPLIT = 16;          // cell ranges (workgroups) per contender in k_select_rescore when H*W >= 32768
constexpr int ESAC_CAND_DOUBLES = 26;          // record parked with an accepted entry of the "maybe" list: 6 + 9 + 6 + 4 doubles (+1 pad)
constexpr int ESAC_SAMPLE_LIST_PER_HYP = 16;   // capacity of the prescreen's global "maybe" list, per hypothesis in flight
constexpr int ESAC_TILED_HC = 256;            // hypotheses per chunk of the tile-stationary score kernel
constexpr int ESAC_TILED_MAX_EXPERTS = 4096;  // experts its bucketing kernel counts in LDS

// Correspondence list of one refinement: every wavefront owns a region = its share of the cells of each error-pass
// trip, rounded up to whole trips.  corr_entries(P) is the size of the whole list (>= P, < P + 2048).
__host__ __device__ inline int corr_region(int P) {
    constexpr int per_trip = ESAC_REFINE_THREADS * ESAC_ERR_UNROLL;
    return (P + per_trip - 1) / per_trip * (64 * ESAC_ERR_UNROLL);
}
__host__ __device__ inline long long corr_entries(int P) { return (long long)corr_region(P) * (ESAC_REFINE_THREADS / 64); }

// Training path (esac_hip_backward).  "slot" = position in the ordered list of hypotheses whose selection
// probability reaches PROB_THRESH (esac_derivative.h:33) -- at most 1000 of them exist for any N.
struct BwdArgs {
    int* sel;             // [N] hypothesis index per slot, ascending
    int* n_sel;           // [1]
    double* probs;        // [N] softmax of the exact scores
    double* losses;       // [N]
    double* ref_hyps;     // [N,6] refined pose (initial pose for unselected hypotheses)
    double* sgrad;        // [N] d expected loss / d score (esac_derivative.h:405-420)
    double* dloss;        // [cap,6] d loss / d refined pose
    uint8_t* maps;        // [cap,2,P] alternating inlier maps of the slot's refinement
    int* map_info;        // [cap,4] accepted buffer (-1: none), inliers of the last accepted step, steps, LM iterations
    void* corr_lists;     // [cap,corr_entries(P)] 16-byte correspondences (only for grids above LDS_CAP)
    double* grad1;        // [cap,3,P] path I: refined pose -> coordinates (unweighted, esac.cpp:375-463)
    double* grad2;        // [cap,3,P] path II: score -> coordinates (esac_derivative.h:205-330)
    double* out;          // [4] expected loss, number of slots, entropy, 0
    float* out_grad;      // [E,3,H,W] accumulated into (+=)
    double gt[16];        // ground-truth camera pose, row-major (double of the float input)
    double gt_pose[6];    // trans2pose(gt) (esac_util.h:555-568)
    double w_rot, w_trans, cut;
    int cap;
    // slot refinement by TEAMS (esac_refine_team.hip): 8 workgroups of one XCD per slot; 0 = one workgroup per slot
    int team;
    int team_max_slots;              // the slot-team launch refines only when n_sel <= this, the one-workgroup-per-slot launch only when
                                     // n_sel > this (0: no team launch): the route is a function of the call's own selection, decided
                                     // on the device -- not of what an earlier call selected
    double* team_gran;               // [cap][2][32][32] granules (16 B each) of the slots' exchanges
    unsigned long long team_tag;     // tag of the slot-team launch: the downstream kernels skip their work when it failed
    int* arrived;                    // [1] workgroups of k_bwd_accumulate that are done: the last one delivers the call's record to the
                                     // pinned slot (KArgs::result_pin) and leaves this at zero
};

constexpr int ESAC_SPEC_CNT_FAN = 32, ESAC_SPEC_CNT_STRIDE = 32;  // counters of the second level; ints between two counters (128 bytes)
constexpr int ESAC_SPEC_CNT_INTS = ESAC_SPEC_CNT_STRIDE * (1 + ESAC_SPEC_CNT_FAN);
struct KArgs {
    // inputs (device)
    const float* sc;        // [E,3,H,W]
    const int64_t* assign;  // [N]
    int E, H, W, N;
    int shift_x, shift_y, sub;
    float focal, ppx, ppy;
    float tau, alpha, beta, max_reproj;
    uint64_t seed, call;
    int max_tries, max_ref_steps, hyp_offset;
    int first_try;             // sampling continues from this try (two-phase throughput shape), 0 otherwise
    int handover;              // k_sample leaves a hypothesis SAMPLE_PENDING once it has spent this many tries
    float margin;
    int flags;                 // ESAC_FLAG_* (include/esac_hip.h)
    int expert_base;           // added to the winner's expert index in the record (esac_hip_params.expert_base)
    const int32_t* hyp_index;  // optional [N] global hypothesis indices
    // workspaces (device)
    double* hyps;         // [N,6]
    double* hyps_R;       // [N,9] R(rvec) as the sampler formed it (rodrigues_vec2mat of the stored rvec): what the exact
                          // re-score multiplies with -- recomputing it costs every re-scoring workgroup ~2.5 us of sin / cos latency
    float* rt32;          // [N,12] float(R(rvec)), float(t)
    int* sample_xy;       // [N,8]
    int* tries;           // [N]
    const float4* sc4;    // [E,H*W] (x,y,z,0) records of the maps for the sampler's gathers, or null (small maps: planar reads hit L2)
    int* samp_resume;     // [N] k_sample_prescreen: first try it did NOT screen (k_sample_screened<true> resumes there)
    int* samp_round;      // [N] next 64-try round of a pending hypothesis: handed out IN ORDER to whichever wavefront asks (k_sample_prescreen)
    unsigned long long* best_try;  // [N] k_sample_decide: (lowest accepted try << 32 | list position) among the listed ones, ~0: none
    double* samp_cand;    // [samp_cap,ESAC_CAND_DOUBLES] solved hypothesis of an accepted list entry: rvec,tvec | R | 12 floats rt32 | 8 ints cells
    int* samp_entries;    // [samp_cap] (frame * N + hypothesis, try) pairs: the tries the screen could not rule out
    int* samp_count;      // [4 + 2048] entries appended to samp_entries (may exceed samp_cap: clamp); hypotheses appended to samp_pending
    int* samp_pending;    // [N] (frame * N + hypothesis) of every hypothesis the first passes left SAMPLE_PENDING
    int samp_cap;
    float* fast_scores;   // [N]
    double* scores;       // [N]
    uint8_t* exact_flag;  // [N]
    int* n_contenders;    // [1] hypotheses inside the band of the fp32 maximum
    double* sel_partials; // [N, ESAC_SELECT_SPLIT] k_select_rescore on large grids: a contender's exact sum per cell range
    int* sel_arrived;     // [N] cell ranges of a contender done (the last one sums the partials in order and resets this to 0)
    double* stats;        // [4] max, sum exp, entropy
    float* errs;          // [P]
    uint8_t* inlier_map;  // [2,P] two alternating buffers; result[31] names the accepted one
    void* corr_list;      // [corr_entries(P)] 16-byte correspondences: refinement fallback when the grid exceeds LDS
    int* inlier_counts;   // [ESAC_MAX_REF_STEPS_K+1]
    double* result;       // [32]
    long long* cycles;    // [32] shader-cycle counters of the refinement kernel's sections (profiling aid)
    long long* tstamps;   // [2N] per-workgroup (start,end) wall-clock stamps of the score kernel, or nullptr
    long long* span_acc;  // [2] accumulated score-kernel span (100 MHz ticks) and launch count
    unsigned long long* status;  // [1] epoch of the last call whose hypAssignment held a value outside [0,E) (0: never)
    // cooperating refinement workgroups (esac_refine.hip, struct Coop); coop_slice is set by the launcher
    double* coop_partials;              // [2][ESAC_REFINE_COOP_MAX][32]
    unsigned long long* coop_counter;   // [2] arrival counter, "a barrier timed out" flag
    int coop_slice;
    int coop_max;                       // workgroups of the cooperative kernel the device holds at once (refine_coop_capacity)
    int coop_extra;                     // ESAC_DEBUG_COOP_STALL: workgroups the barrier waits for beyond those launched (0 normally)
    unsigned long long coop_tag;        // launch number << 20: tags the exchange granules and the failure word of THIS refinement launch
    int team;                           // members of a refinement team on small grids (0: one workgroup refines; esac_hip_set_refine_team)
    int team_auto;                      // 1: a.team == the default stands for "the smallest team <= 16 that lowers the cells per lane" (refine_team_members)
    int team_stride;                    // the team's members are the workgroups blockIdx.x % team_stride == 0 (8: one XCD; 1: debug, spread)
    int solo;                           // 1: launch_refine takes ONE workgroup per refinement whatever the shape (the retry after a shared
                                        // refinement timed out: neither a team nor cooperating workgroups, both need co-residency again)
    int* refine_info;                   // [8] mode, members, XCD census (4 bits per XCD), same-XCD flag, exchanges, failed
    int fold_select;                    // 1: the team kernel also runs the selection (softmax statistics, band of contenders, their exact
                                        // re-score) in its prologue: no k_select_rescore launch in front of it (refine_folds_select);
                                        // 2: ESAC_FLAG_EXACT_SCORES -- every score is exact already, the prologue only takes the softmax
                                        // statistics and the argmax: no k_stats_exact launch (refine_folds_exact_stats)
    // SPECULATIVE forward (round 6; esac_capi.hip: forward_impl, several experts): after the sampler's first pass the hypotheses
    // it SETTLED are scored, selected from and their winner refined on the launch stream, while the straggler chain
    // (k_sample_prescreen .. k_sample_screened: wrong-expert hypotheses that practically never win) and the stragglers' scores run
    // beside them on a stream of the context's own; k_spec_join then folds the stragglers in and delivers -- or finds that a
    // straggler wins after all, and the refinement runs again for it.  Every output is what the serial order produces.
    int spec_mode;          // 0: off.  Score / selection kernels: 1 = the settled hypotheses only (spec_flag[h] == 0).
                            // Refinement kernels: 2 = speculative (the hypothesis to refine is the fp32 argmax of the settled ones --
                            // the selection runs beside the refinement; no record leaves the workspace; the status word goes to
                            // spec_state[1], then "done" to spec_state[6]; nothing settled at all: give up at once)
    uint8_t* spec_flag;     // [N] 1: the sampler's first pass left hypothesis h to the straggler chain (written there, read-only afterwards)
    double* spec_state;     // [8] [0] the epoch of the call whose speculation FAILED (k_spec_join), [1] status word of the speculative
                            // refinement, [2] failed speculations so far, [3] "the chain may start" = the epoch of the call whose
                            // speculative refinement has STARTED (its workgroups are resident), [4] "the chain is done" = the epoch
                            // of the call whose straggler chain has finished, [5] hand-offs that timed out so far, [6] "the speculative
                            // refinement is done", [7] "the join is done" (each the epoch of the call it belongs to)
    int* spec_cnt;          // [ESAC_SPEC_CNT_INTS] arrival counters of k_score_stragglers' workgroups, two levels (the last arrival writes
                            // spec_state[4]; every counter is left at zero)
    int spec_gate;          // refinement kernels: 1 = wait for "the join is done", then return at once unless spec_state[0] is this call's
                            // epoch (the second refinement is enqueued with the call and runs only when the speculation failed)
    int spec_debug;         // 1: ESAC_DEBUG_SPEC_SECOND_BEST
    // tile-stationary score (esac_score_tiled.hip); null / 0 when the call uses the per-hypothesis stream
    int* order;           // [N] hypothesis at sorted position pos (sorted by expert)
    float* rt_sorted;     // [N,12] rt32 rows in sorted order
    int* chunks;          // [n_chunks_max,4] expert, first sorted position, count (<= ESAC_TILED_HC), 0
    int* n_chunks;        // [1]
    int* bucket_fill;     // [E] hypotheses per expert, then (k_bucket_scan) the next free sorted position of each expert
    float* partials;      // [n_sub,N] partial sums per (sub-tile, sorted position)
    int n_sub, n_chunks_max;
    // caller-visible outputs written by the kernels themselves (no copy kernels on the critical path)
    double* scores_user;  // optional device [N]: the score vector
    double* result_user;  // optional device [32]: the result record
    double* result_pin;   // optional pinned HOST memory [ESAC_PIN_DOUBLES] (device-visible): result record, epoch word,
                          // status word (1.0: this call's hypAssignment held an out-of-range value)
    double epoch;         // value stored into result_pin[32] after the record (the host polls it)
    double sample_epoch;  // epoch of the launch that sampled the hypotheses in the workspace: what KArgs::status is compared with
    // batched calls: frame b = blockIdx.y works on its own slice of every buffer (device_common.hpp:frame_view)
    int frames;                 // B >= 1
    long long sc_frame_stride;  // elements between the coordinate tensors of consecutive frames
    BwdArgs bwd;                // training path only
};

void launch_sample(const KArgs& a, hipStream_t s);
// speculative forward: can this call's sampler be split into a first pass and a straggler chain (several experts, a few thousand
// hypotheses at most, the screened route)?  launch_sample_split: the first pass on `s` (its last kernel signals `fork`), the chain
// on `side` behind that event
bool sample_can_split(const KArgs& a);
int launch_sample_split(const KArgs& a, hipStream_t s, KArgs* chain, int* chain_waves);
void launch_sample_stragglers_on(const KArgs& chain, int waves, hipStream_t side);
void launch_score_stragglers(const KArgs& a, hipStream_t side);  // fp32 scores of the stragglers + the "chain is done" word
void launch_spec_join(const KArgs& a, hipStream_t s);
// hand-offs between the two streams of a speculative call through words in device memory, not events: a one-wavefront kernel that
// waits until spec_state[which] is this call's epoch
void launch_spec_wait(const KArgs& a, int which, hipStream_t s);
void launch_hyps_to_rt32(const KArgs& a, hipStream_t s);
void launch_score_fast(const KArgs& a, hipStream_t s);
void launch_score(const KArgs& a, hipStream_t s);  // the fp32 score in the shape the C ABI chose (a.partials != null: tiled)
void launch_score_tiled(const KArgs& a, hipStream_t s);
void launch_bucket_order(const KArgs& a, hipStream_t s);  // esac_score_tiled.hip: order[] = hypotheses sorted by expert
int tiled_sub_tiles(int P);
void launch_select_rescore(const KArgs& a, hipStream_t s);
void launch_rescore_all(const KArgs& a, hipStream_t s);
void launch_stats_exact(const KArgs& a, hipStream_t s);
void launch_pick_record(const double* records, int world, double* pin, double epoch, double* zero, int n_zero, hipStream_t s);
void launch_shard_balanced(const int64_t* assign, int N, int E, int world, int rank, int expert_base, int32_t* index_out,
                           int64_t* assign_out, int32_t* info_out, hipStream_t s);
int refine_coop_capacity();              // resident workgroups of the cooperative refinement kernel on the current device
int refine_coop_slice(const KArgs& a);  // cells per cooperating refinement workgroup, 0: one workgroup refines
int refine_team_members(const KArgs& a);  // members of the team that refines a small grid, 0: one workgroup refines
unsigned long long launch_refine(const KArgs& a, hipStream_t s);  // returns the tag of a shared (cooperative / team) launch, 0 otherwise
unsigned long long launch_refine_team(const KArgs& a, hipStream_t s);  // esac_refine_team.hip; requires refine_team_members(a) > 0
unsigned long long next_refine_tag();
bool refine_folds_select(const KArgs& a);  // the refinement launch of this call can (and will) do the selection itself
bool refine_folds_exact_stats(const KArgs& a);  // ... the softmax statistics of the exact scores (ESAC_FLAG_EXACT_SCORES)
// training path (esac_backward.hip, esac_refine.hip)
void launch_refine_slots(const KArgs& a, hipStream_t s);
bool refine_slots_can_team(const KArgs& a);            // grid fits a team of 8 (1024 .. 8192 cells)
unsigned long long launch_refine_slots_team(KArgs& a, hipStream_t s);  // esac_refine_team.hip; sets a.bwd.team_tag, returns it
void launch_bwd_select(const KArgs& a, hipStream_t s);
void launch_bwd_loss(const KArgs& a, hipStream_t s);
void launch_bwd_paths(const KArgs& a, hipStream_t s);  // path I and path II of every slot, one launch
void launch_bwd_accumulate(const KArgs& a, hipStream_t s);

}  // namespace esac
