// esac_capi.hip -- C ABI (include/esac_hip.h) over the HIP kernels.
// Host side of the drop-in boundary: argument validation with the reference's
// failure convention (everything that would have thrown c10::Error / cv::Exception
// through pybind11 becomes a negative status + message), workspace ownership,
// kernel launches on the caller's stream, optional per-phase hipEvent timers
// (the StopWatch prints of esac.cpp:124,149,161,179).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sched.h>
#include <time.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/esac_hip.h"
#include "esac_kernels.hpp"
#include "pose_math.hpp"

using namespace esac;

// RCCL's handful of types, declared here (see rccl() below: the library is bound by dlopen, its headers are not needed)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
}

static_assert(ESAC_RES_SCORE == ESAC_RES_SCORE_K && ESAC_RES_HYP == ESAC_RES_HYP_K && ESAC_RES_EXPERT == ESAC_RES_EXPERT_K &&
                  ESAC_RES_RVEC == ESAC_RES_RVEC_K && ESAC_RES_POSE == ESAC_RES_POSE_K &&
                  ESAC_RES_REF_STEPS == ESAC_RES_REF_STEPS_K && ESAC_RES_INLIERS == ESAC_RES_INLIERS_K &&
                  ESAC_RES_PROB == ESAC_RES_PROB_K && ESAC_RES_ENTROPY == ESAC_RES_ENTROPY_K &&
                  ESAC_RES_CONTENDERS == ESAC_RES_CONTENDERS_K && ESAC_RES_LM_ITERS == ESAC_RES_LM_ITERS_K &&
                  ESAC_MAX_REF_STEPS == ESAC_MAX_REF_STEPS_K && ESAC_BWD_MAX_SLOTS == ESAC_BWD_SLOTS_K &&
                  ESAC_FLAG_EXACT_SCORES == ESAC_FLAG_EXACT_SCORES_K && ESAC_FLAG_EXACT_SAMPLING == ESAC_FLAG_EXACT_SAMPLING_K &&
                  ESAC_FLAG_SCORES_BY_INDEX == ESAC_FLAG_SCORES_BY_INDEX_K && ESAC_REFINE_TEAM_MAX == ESAC_REFINE_TEAM_MAX_K &&
                  ESAC_REFINE_TEAM_DEFAULT == ESAC_REFINE_TEAM_DEFAULT_K &&
                  (ESAC_FLAG_AUTO_EXACT & (ESAC_FLAG_EXACT_SCORES | ESAC_FLAG_EXACT_SAMPLING | ESAC_FLAG_SCORES_BY_INDEX)) == 0,
              "result layout drifted between include/esac_hip.h and esac_kernels.hpp");

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_OK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(-100 - (int)_e, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// Makes the context's GPU current for the duration of one entry point and restores the caller's device afterwards
// (torch and every other runtime user read the same thread-local "current device").
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != dev) prev = cur;
        if (cur != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct esac_hip_ctx {
    int device = 0;
    int capN = 0, capP = 0, capB = 0;  // capN / capP count elements over ALL frames of a batch
    KArgs ws{};  // only the workspace pointers are kept here
    int lastN = 0, lastH = 0, lastW = 0;
    bool timing = false;
    int timing_period = 1;       // record the phase events / device-side stamps on every timing_period-th forward call
    long long timing_calls = 0;  // forward calls since timing was enabled
    bool keep_errs = false;  // esac_hip_set_debug: store the winner's error image
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    double* h_pin = nullptr;  // pinned, device-visible host buffer: result record [32] + epoch word
    double* d_pin = nullptr;  // its device address
    double epoch = 0;         // bumped by every entry point: hand-off word of the pinned record
    double sample_epoch = 0;  // epoch of the most recent SAMPLING launch: what the device-side status word is tagged with
    int wait_mode = ESAC_WAIT_SPIN;
    int coop_max = 0;         // cooperative refinement workgroups this device holds at once (refine_coop_capacity)
    bool coop_stall = false;  // ESAC_DEBUG_COOP_STALL
    int team = ESAC_REFINE_TEAM_DEFAULT;  // members of the refinement team on small grids (esac_hip_set_refine_team; 0: one workgroup)
    bool team_spread = false;             // ESAC_DEBUG_TEAM_SPREAD: the members are consecutive workgroups (one per XCD)
    bool team_auto = true;                // the default team size is chosen per grid (ESAC_REFINE_TEAM_AUTO=0: exactly the default, A/B)
    bool team_auto_env_off = false;
    unsigned long long refine_tag = 0;    // tag of the most recent shared (cooperative / team) refinement launch, 0: none yet
    unsigned long long checked_tag = 0;   // the failed launch esac_hip_check has already counted as a strike
    bool refine_was_team = false;         // the most recent forward's refinement launch was a team's
    long long team_fallbacks = 0;         // blocking calls whose team timed out and were refined again by one workgroup
    int team_strikes = 0;                 // consecutive forward calls whose team timed out; at ESAC_TEAM_STRIKES the context stops asking
    bool team_latched_off = false;        // ... for teams (a caller that keeps the GPU's CUs busy on another stream would otherwise pay the
    long long solo_since_latch = 0;       // time-out on every frame); re-armed after ESAC_TEAM_REARM_CALLS calls or by esac_hip_set_refine_team
    bool fold_select = true;              // the team kernel may run the selection in its prologue (ESAC_FOLD_SELECT=0: measurement scripts)
    int last_nsel = 0;                    // slots the most recent blocking esac_hip_backward refined (0: none yet; reported, decides nothing)
    bool slot_teams = true;               // training path: slots may be refined by teams (off after a time-out until
                                          // esac_hip_set_refine_team re-arms it; ESAC_SLOT_TEAMS=0)
    long long slot_team_calls = 0, slot_team_fallbacks = 0;
    bool last_bwd_teams = false;          // the most recent esac_hip_backward refined its slots by teams
    BwdArgs bws{};  // training-path workspace (pointers only), sized for bN hypotheses, bP cells, bcap slots
    int bN = 0, bP = 0, bcap = 0;
    bool b_lists = false;
    float4* sc4 = nullptr;  // packed copy of the maps for the sampler (ensure_pack_ws)
    long long sc4_cells = 0;
    // tile-stationary score workspace (ensure_tiled_ws)
    int tN = 0, tChunks = 0;
    long long tPart = 0;
    bool rt32_stale = false;  // esac_hip_write_hyps ran: the fp32 [R|t] rows are rebuilt by the next esac_hip_score
    double host_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // esac_hip_host_turn: where the host's time of the most recent blocking forward went
    double host_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... summed over the blocking forward calls since the last reset (esac_hip_host_turn_mean)
    double host_last_return = 0;                    // CLOCK_MONOTONIC at which the previous blocking forward returned
    long long host_n = 0;
    // speculative forward (forward_impl): the straggler chain of the sampler runs on this stream beside the launch stream
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;          // ... and the selection among the settled hypotheses + the join on this one, beside the speculative refinement
    bool spec_off = false, spec_env_off = false;  // ESAC_DEBUG_NO_SPECULATION / ESAC_SPECULATE=0
    bool spec_second_best = false;        // ESAC_DEBUG_SPEC_SECOND_BEST
    bool spec_lose_chain = false;         // ESAC_DEBUG_SPEC_LOSE_CHAIN
    long long spec_calls = 0;             // forward calls that took the speculative route
    double last_spec_epoch = 0;           // epoch of the most recent speculative call (0: the most recent forward was not)
    ncclComm_t comm = nullptr;  // esac_hip_comm_init: this context's rank in an RCCL communicator (the multi-GPU score exchange)
    int comm_ranks = 0, comm_rank = 0;
};

static inline double now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

extern "C" int esac_hip_abi_version(void) { return ESAC_HIP_ABI_VERSION; }
extern "C" const char* esac_hip_last_error(void) { return g_err; }
extern "C" int esac_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

constexpr int ESAC_SLOT_TEAMS_MAX = 32;  // training path: slots refined by teams when the call selects at most this many
constexpr int ESAC_TEAM_STRIKES = 2;
constexpr long long ESAC_TEAM_REARM_CALLS = 1000;

// RCCL, bound at the first esac_hip_comm_* call (see "the one collective of the multi-GPU path" below).  The handful of
// types and entry points this file needs are declared HERE (the stable NCCL 2.x C API: a 128-byte unique id, an opaque
// communicator, ncclResult_t 0 = success, ncclSum = 0, ncclDouble = 8), not taken from <rccl/rccl.h>: a single-GPU build of the
// library needs neither RCCL's headers at compile time nor its shared object at run time.
static_assert(sizeof(ncclUniqueId) == ESAC_COMM_ID_BYTES, "unique id size");
constexpr ncclResult_t ncclSuccess = 0;
constexpr int NCCL_SUM = 0, NCCL_DOUBLE = 8;  // ncclRedOp_t ncclSum, ncclDataType_t ncclFloat64
struct Rccl {
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;      // what the communicator itself reports (esac_hip_comm_info)
    ncclResult_t (*comm_user_rank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*comm_cu_device)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};
static const Rccl& rccl() {
    static const Rccl bound = [] {
        Rccl r;
        void* h = nullptr;
        // the copy the process already holds first (torch ships its own librccl.so under another path: two copies of RCCL in
        // one process would each bring up their own transports), then the loader's search path, then ROCm's
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return r;
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(h, "ncclAllReduce"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(h, "ncclGetErrorString"));
        r.comm_count = reinterpret_cast<decltype(r.comm_count)>(dlsym(h, "ncclCommCount"));
        r.comm_user_rank = reinterpret_cast<decltype(r.comm_user_rank)>(dlsym(h, "ncclCommUserRank"));
        r.comm_cu_device = reinterpret_cast<decltype(r.comm_cu_device)>(dlsym(h, "ncclCommCuDevice"));
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.error_string;
        return r;
    }();
    return bound;
}
static void drop_comm(esac_hip_ctx* c) {
    if (c->comm) (void)rccl().comm_destroy(c->comm);  // a communicator exists only if RCCL was bound
    c->comm = nullptr;
    c->comm_ranks = 0;
}

static void free_ws(esac_hip_ctx* c) {
    void* ptrs[] = {c->ws.hyps,       c->ws.hyps_R,      c->ws.rt32,         c->ws.sample_xy, c->ws.tries,      c->ws.samp_resume, c->ws.samp_round, c->ws.best_try, c->ws.samp_cand, c->ws.samp_entries, c->ws.samp_count, c->ws.samp_pending, c->ws.fast_scores,
                    c->ws.scores,     c->ws.exact_flag,   c->ws.n_contenders, c->ws.sel_partials, c->ws.sel_arrived, c->ws.stats,
                    c->ws.errs,       c->ws.inlier_map,   c->ws.inlier_counts, c->ws.result, c->ws.corr_list, c->ws.cycles, c->ws.tstamps, c->ws.span_acc,
                    c->ws.status,     c->ws.coop_partials, c->ws.coop_counter, c->ws.refine_info, c->ws.order,        c->ws.rt_sorted,  c->ws.chunks,     c->ws.n_chunks,  c->ws.partials, c->ws.bucket_fill,
                    c->ws.spec_flag,  c->ws.spec_state,  c->ws.spec_cnt};
    c->tN = c->tChunks = 0;
    c->tPart = 0;
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->ws = KArgs{};
    c->capN = c->capP = c->capB = 0;
}

extern "C" int esac_hip_create(esac_hip_ctx** out, int device) {
    if (!out) return fail(-1, "esac_hip_create: null ctx pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(-2, "esac_hip_create: no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(-3, "esac_hip_create: device %d out of range [0,%d)", device, n);
    DeviceGuard guard(device);
    esac_hip_ctx* c = new esac_hip_ctx();
    c->device = device;
    for (auto& ev : c->ev) HIP_OK(hipEventCreate(&ev));
    HIP_OK(hipHostMalloc((void**)&c->h_pin, (size_t)ESAC_PIN_DOUBLES * ESAC_MAX_BATCH * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_pin, 0, (size_t)ESAC_PIN_DOUBLES * ESAC_MAX_BATCH * sizeof(double));
    HIP_OK(hipHostGetDevicePointer((void**)&c->d_pin, c->h_pin, 0));
    c->coop_max = refine_coop_capacity();  // CUs x resident workgroups of the cooperative refinement kernel on THIS device
    if (const char* e = getenv("ESAC_REFINE_TEAM")) {  // start value of esac_hip_set_refine_team (measurement scripts)
        const int g = atoi(e);
        c->team = g < 2 ? 0 : (g > ESAC_REFINE_TEAM_MAX ? ESAC_REFINE_TEAM_MAX : g);
    }
    if (const char* e = getenv("ESAC_FOLD_SELECT")) c->fold_select = atoi(e) != 0;
    if (const char* e = getenv("ESAC_REFINE_TEAM_AUTO")) c->team_auto_env_off = atoi(e) == 0;
    if (c->team_auto_env_off || getenv("ESAC_REFINE_TEAM")) c->team_auto = false;  // (an explicit start value is an explicit size)
    if (const char* e = getenv("ESAC_SLOT_TEAMS")) c->slot_teams = atoi(e) != 0;
    if (const char* e = getenv("ESAC_SPECULATE")) c->spec_off = c->spec_env_off = atoi(e) == 0;
    *out = c;
    return 0;
}

static void free_bws(esac_hip_ctx* c) {
    void* ptrs[] = {c->bws.sel,   c->bws.n_sel, c->bws.probs,    c->bws.losses,     c->bws.ref_hyps, c->bws.sgrad, c->bws.dloss,
                    c->bws.maps,  c->bws.map_info, c->bws.corr_lists, c->bws.grad1, c->bws.grad2,   c->bws.out, c->bws.team_gran, c->bws.arrived};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->bws = BwdArgs{};
    c->bN = c->bP = c->bcap = 0;
    c->b_lists = false;
}

extern "C" int esac_hip_destroy(esac_hip_ctx* c) {
    if (!c) return 0;
    DeviceGuard guard(c->device);
    free_ws(c);
    free_bws(c);
    if (c->sc4) (void)hipFree(c->sc4);
    drop_comm(c);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->side2) (void)hipStreamDestroy(c->side2);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    delete c;
    return 0;
}

template <typename T>
static int alloc(T** p, size_t n) {
    HIP_OK(hipMalloc((void**)p, n * sizeof(T)));
    return 0;
}

static int ensure_ws(esac_hip_ctx* c, int N1, int P1, int B = 1) {
    const long long N = (long long)N1 * B, P = (long long)P1 * B;
    if (N <= c->capN && P <= c->capP && B <= c->capB) return 0;
    if (N > 0x7fffffffLL || P > 0x7fffffffLL) return fail(-4, "batch too large");
    HIP_OK(hipDeviceSynchronize());
    const int nN = N > c->capN ? (int)N : c->capN, nP = P > c->capP ? (int)P : c->capP;
    const int nB = B > c->capB ? B : c->capB;
    // hypotheses handed in through esac_hip_write_hyps (and the status word) survive a growing workspace
    double* old_hyps = c->ws.hyps;
    const size_t old_n = (size_t)c->capN;
    unsigned long long old_status = 0;
    if (c->ws.status) HIP_OK(hipMemcpy(&old_status, c->ws.status, sizeof(old_status), hipMemcpyDeviceToHost));
    c->ws.hyps = nullptr;
    free_ws(c);
    int rc = 0;
    rc |= alloc(&c->ws.hyps, (size_t)nN * 6);
    rc |= alloc(&c->ws.hyps_R, (size_t)nN * 9);
    rc |= alloc(&c->ws.rt32, (size_t)nN * 12);
    rc |= alloc(&c->ws.status, (size_t)1);
    // the exchange buffer of shared refinements: partial sums of cooperating workgroups [2][256][32] doubles, or the granules
    // of up to ESAC_TEAM_BATCH_MAX teams (16 bytes each)
    static_assert((size_t)ESAC_TEAM_BATCH_MAX * ESAC_TEAM_GRANULES * 2 >= (size_t)2 * ESAC_REFINE_COOP_MAX * 32, "exchange buffer");
    rc |= alloc(&c->ws.coop_partials, (size_t)ESAC_TEAM_BATCH_MAX * ESAC_TEAM_GRANULES * 2);
    rc |= alloc(&c->ws.coop_counter, (size_t)2);
    rc |= alloc(&c->ws.refine_info, (size_t)8);
    rc |= alloc(&c->ws.sample_xy, (size_t)nN * 8);
    rc |= alloc(&c->ws.tries, (size_t)nN);
    rc |= alloc(&c->ws.samp_resume, (size_t)nN);
    rc |= alloc(&c->ws.samp_round, (size_t)nN);
    rc |= alloc(&c->ws.best_try, (size_t)nN);
    rc |= alloc(&c->ws.samp_cand, (size_t)nN * ESAC_SAMPLE_LIST_PER_HYP * ESAC_CAND_DOUBLES);
    rc |= alloc(&c->ws.samp_entries, (size_t)nN * 2 * ESAC_SAMPLE_LIST_PER_HYP);  // (hypothesis, try) pairs
    rc |= alloc(&c->ws.samp_count, (size_t)4 + 2 * 1024);  // list counters + 1024 x 2 per-expert counters (esac_kernels.hip: expert_stats)
    rc |= alloc(&c->ws.samp_pending, (size_t)nN);
    rc |= alloc(&c->ws.fast_scores, (size_t)nN);
    rc |= alloc(&c->ws.scores, (size_t)nN);
    rc |= alloc(&c->ws.exact_flag, (size_t)nN);
    rc |= alloc(&c->ws.n_contenders, (size_t)4 * nB);
    rc |= alloc(&c->ws.sel_partials, (size_t)nN * ESAC_SELECT_SPLIT);
    rc |= alloc(&c->ws.sel_arrived, (size_t)nN);
    rc |= alloc(&c->ws.stats, (size_t)4 * nB);
    rc |= alloc(&c->ws.errs, (size_t)nP);
    rc |= alloc(&c->ws.inlier_map, (size_t)nP * 2);  // two buffers, see esac_refine.hip
    {
        char* cl = nullptr;
        rc |= alloc(&cl, ((size_t)nP + (size_t)2048 * nB) * 16);  // sum over frames of corr_entries(P) < P + 2048 each
        c->ws.corr_list = cl;
    }
    rc |= alloc(&c->ws.inlier_counts, (size_t)(ESAC_MAX_REF_STEPS + 1) * nB);
    rc |= alloc(&c->ws.result, (size_t)ESAC_RES_DOUBLES * nB);
    rc |= alloc(&c->ws.cycles, (size_t)32);
    rc |= alloc(&c->ws.tstamps, (size_t)nN * 2);
    rc |= alloc(&c->ws.span_acc, (size_t)2);
    rc |= alloc(&c->ws.spec_flag, (size_t)nN);
    rc |= alloc(&c->ws.spec_state, (size_t)8);
    rc |= alloc(&c->ws.spec_cnt, (size_t)ESAC_SPEC_CNT_INTS);
    if (rc) {
        if (old_hyps) (void)hipFree(old_hyps);
        return rc;
    }
    HIP_OK(hipMemset(c->ws.span_acc, 0, 2 * sizeof(long long)));
    HIP_OK(hipMemset(c->ws.spec_flag, 0, (size_t)nN));
    HIP_OK(hipMemset(c->ws.spec_state, 0, 8 * sizeof(double)));
    HIP_OK(hipMemset(c->ws.spec_cnt, 0, ESAC_SPEC_CNT_INTS * sizeof(int)));
    HIP_OK(hipMemset(c->ws.coop_counter, 0, 2 * sizeof(unsigned long long)));  // [1]: tag of the last failed shared refinement (esac_hip_check)
    HIP_OK(hipMemset(c->ws.coop_partials, 0, (size_t)ESAC_TEAM_BATCH_MAX * ESAC_TEAM_GRANULES * 2 * sizeof(double)));  // (also the teams' granules)
    HIP_OK(hipMemset(c->ws.refine_info, 0, 8 * sizeof(int)));
    HIP_OK(hipMemset(c->ws.samp_count, 0, (4 + 2 * 1024) * sizeof(int)));       // the screened chain leaves them at zero (esac_kernels.hip)
    HIP_OK(hipMemset(c->ws.hyps, 0, (size_t)nN * 6 * sizeof(double)));
    HIP_OK(hipMemcpy(c->ws.status, &old_status, sizeof(old_status), hipMemcpyHostToDevice));
    if (old_hyps) {
        if (old_n) HIP_OK(hipMemcpy(c->ws.hyps, old_hyps, old_n * 6 * sizeof(double), hipMemcpyDeviceToDevice));
        (void)hipFree(old_hyps);
    }
    HIP_OK(hipMemset(c->ws.result, 0, (size_t)ESAC_RES_DOUBLES * nB * sizeof(double)));
    HIP_OK(hipMemset(c->ws.n_contenders, 0, (size_t)4 * nB * sizeof(int)));
    HIP_OK(hipMemset(c->ws.sel_arrived, 0, (size_t)nN * sizeof(int)));  // k_select_rescore leaves it zero after every call
    c->capN = nN;
    c->capP = nP;
    c->capB = nB;
    return 0;
}

// Which shape the fp32 score runs in.  Per-hypothesis stream (k_score_fast): every hypothesis re-reads its expert's map,
// fine while a map is L2-resident and hypotheses are few.  Tile-stationary (esac_score_tiled.hip): each map tile is read
// once per chunk of <= 256 hypotheses -- pays when a map no longer fits the caches next to the other experts' maps
// (full-resolution 480x640 maps: 3.7 MB each) and enough hypotheses share it.  ESAC_FLAG_SCORE_TILED / _STREAM override.
static bool want_tiled(const esac_hip_params* p, const float* d_sc, int B) {
    const long long P = (long long)p->H * p->W;
    const bool legal = B == 1 && (p->W & 3) == 0 && (reinterpret_cast<uintptr_t>(d_sc) & 15) == 0 &&
                       p->E <= ESAC_TILED_MAX_EXPERTS && P >= 4;
    const long long partial_bytes = (long long)tiled_sub_tiles((int)P) * p->N * 4;
    if (!legal || partial_bytes > (4LL << 30) || (p->flags & ESAC_FLAG_SCORE_STREAM)) return false;
    // the tile kernel folds k = |beta| log2(e) into the pose rows and multiplies by 2^(-+k tau) after the exp2: beyond
    // k tau ~ 126 that constant under- / overflows (scores NaN or saturated); the stream keeps the subtraction in the
    // exponent and is right for any parameters
    if (!(fabsf(p->inlier_beta) * 1.4426950408889634f * fabsf(p->inlier_thresh) <= 100.0f)) return false;
    if (p->flags & ESAC_FLAG_SCORE_TILED) return true;
    return P >= 32768 && p->N >= 64;
}

// Packed (x,y,z,0) copy of the maps for the sampler: worth one extra pass over the maps when they are far beyond the L2s
// (every random 4-byte gather would otherwise fetch its own cache line, three per cell) and hypotheses of several experts
// will need many tries.  Single frames only.
static bool want_pack(const esac_hip_params* p, int B) {
    if (B != 1) return false;
    if (p->flags & ESAC_FLAG_PACK_MAPS) return true;
    return p->E > 1 && (long long)p->E * p->H * p->W * 12 >= (32LL << 20) && p->N >= 256;
}
static int ensure_pack_ws(esac_hip_ctx* c, long long cells) {
    if (cells <= c->sc4_cells) return 0;
    HIP_OK(hipDeviceSynchronize());
    if (c->sc4) (void)hipFree(c->sc4);
    c->sc4 = nullptr;
    c->sc4_cells = 0;
    HIP_OK(hipMalloc((void**)&c->sc4, (size_t)cells * sizeof(float4)));
    c->sc4_cells = cells;
    return 0;
}

static int ensure_tiled_ws(esac_hip_ctx* c, int N, int P, int E) {
    const int n_sub = tiled_sub_tiles(P);
    const int chunks = N / ESAC_TILED_HC + (E < N ? E : N) + 1;
    const long long part = (long long)n_sub * N;
    if (N <= c->tN && chunks <= c->tChunks && part <= c->tPart) return 0;
    HIP_OK(hipDeviceSynchronize());
    void* ptrs[] = {c->ws.order, c->ws.rt_sorted, c->ws.chunks, c->ws.n_chunks, c->ws.partials, c->ws.bucket_fill};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    c->ws.order = nullptr; c->ws.rt_sorted = nullptr; c->ws.chunks = nullptr; c->ws.n_chunks = nullptr; c->ws.partials = nullptr;
    c->ws.bucket_fill = nullptr;
    const int nN = N > c->tN ? N : c->tN, nC = chunks > c->tChunks ? chunks : c->tChunks;
    const long long nP = part > c->tPart ? part : c->tPart;
    c->tN = c->tChunks = 0;
    c->tPart = 0;
    int rc = 0;
    rc |= alloc(&c->ws.order, (size_t)nN);
    rc |= alloc(&c->ws.rt_sorted, (size_t)nN * 12);
    rc |= alloc(&c->ws.chunks, (size_t)nC * 4);
    rc |= alloc(&c->ws.n_chunks, (size_t)4);
    rc |= alloc(&c->ws.partials, (size_t)nP);
    rc |= alloc(&c->ws.bucket_fill, (size_t)ESAC_TILED_MAX_EXPERTS);
    if (rc) return rc;
    c->tN = nN; c->tChunks = nC; c->tPart = nP;
    return 0;
}

// Validation: what the reference leaves to accessor<>() / OpenCV asserts.
static int make_args(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, KArgs* out,
                     int B = 1, long long sc_frame_stride = 0) {
    if (!c) return fail(-1, "null context");
    if (!p) return fail(-1, "null params");
    if (!d_sc || !d_assign) return fail(-1, "null scene-coordinate or assignment pointer");
    if (p->E <= 0 || p->N <= 0) return fail(-4, "E=%d, N=%d must be positive", p->E, p->N);
    if (p->H < 3 || p->W < 3 || (int64_t)(p->H - 1) * (p->W - 1) < 4)
        return fail(-4, "grid %dx%d too small: 4 distinct cells must exist in [0,W-2]x[0,H-2] (esac_util.h:164-176)", p->H, p->W);
    if ((int64_t)p->H * p->W > (int64_t)1 << 28 || p->H > 65535 || p->W > 65535)
        return fail(-4, "grid %dx%d too large (at most 65535 rows / columns, 2^28 cells)", p->H, p->W);
    if (p->sub_sampling <= 0) return fail(-4, "subSampling=%d must be positive", p->sub_sampling);
    {   // pixel centres col*sub + sub/2 - shift (esac_util.h:64-66) are formed in int32 on the device
        const int64_t lim = 0x7fffffffLL, half = p->sub_sampling / 2;
        const int64_t xs[4] = {half - p->shift_x, (int64_t)(p->W - 1) * p->sub_sampling + half - p->shift_x,
                               half - p->shift_y, (int64_t)(p->H - 1) * p->sub_sampling + half - p->shift_y};
        for (int64_t v : xs)
            if (v > lim || v < -lim) return fail(-4, "pixel positions overflow int32 (subSampling=%d, shift=(%d,%d))", p->sub_sampling, p->shift_x, p->shift_y);
    }
    if (!(p->focal > 0)) return fail(-4, "focal length must be positive");
    const int P = p->H * p->W;
    if (B < 1 || B > ESAC_MAX_BATCH) return fail(-4, "batch size %d outside [1,%d]", B, ESAC_MAX_BATCH);
    int rc = ensure_ws(c, p->N, P, B);
    if (rc) return rc;
    const bool tiled = want_tiled(p, d_sc, B);
    if (tiled && (rc = ensure_tiled_ws(c, p->N, P, p->E))) return rc;
    const bool pack = want_pack(p, B);
    if (pack && (rc = ensure_pack_ws(c, (long long)p->E * P))) return rc;
    KArgs a = c->ws;
    a.sc4 = pack ? c->sc4 : nullptr;
    if (tiled) {
        a.n_sub = tiled_sub_tiles(P);
        a.n_chunks_max = p->N / ESAC_TILED_HC + (p->E < p->N ? p->E : p->N) + 1;
    } else {
        a.partials = nullptr;  // launch_score: per-hypothesis stream
    }
    a.frames = B;
    a.sc_frame_stride = sc_frame_stride;
    a.sc = d_sc;
    a.assign = d_assign;
    a.E = p->E; a.H = p->H; a.W = p->W; a.N = p->N;
    a.shift_x = p->shift_x; a.shift_y = p->shift_y; a.sub = p->sub_sampling;
    a.focal = p->focal; a.ppx = p->ppx; a.ppy = p->ppy;
    a.tau = p->inlier_thresh; a.alpha = p->inlier_alpha; a.beta = p->inlier_beta; a.max_reproj = p->max_reproj;
    a.seed = p->seed; a.call = p->call;
    a.max_tries = p->max_tries > 0 ? p->max_tries : ESAC_MAX_SAMPLING_TRIES;
    a.max_ref_steps = p->max_ref_steps >= 0 ? (p->max_ref_steps < ESAC_MAX_REF_STEPS ? p->max_ref_steps : ESAC_MAX_REF_STEPS)
                                            : ESAC_MAX_REF_STEPS;
    a.hyp_offset = p->hyp_offset;
    a.hyp_index = p->d_hyp_index;
    a.expert_base = p->expert_base;
    a.coop_max = c->coop_max;
    a.coop_extra = c->coop_stall ? 1 : 0;
    a.team = c->team;  // (the forward entry points fold the time-out latch in: forward_team)
    a.team_auto = c->team_auto ? 1 : 0;
    a.team_stride = c->team_spread ? 1 : 8;
    a.solo = 0;
    a.spec_mode = 0;
    a.spec_gate = 0;
    a.spec_debug = 0;
    a.spec_flag = nullptr;  // (forward_impl hands the flags to the kernels of a speculative call only)
    a.samp_cap = (int)(((long long)c->capN * ESAC_SAMPLE_LIST_PER_HYP) > 0x7fffffffLL ? 0x7fffffff : (long long)c->capN * ESAC_SAMPLE_LIST_PER_HYP);
    a.flags = p->flags;
    c->epoch += 1.0;  // every call gets its own epoch: result hand-off word and the tag of the status word
    a.epoch = c->epoch;
    a.sample_epoch = c->sample_epoch;  // launches that sample call mark_sampling() and overwrite this
    // (device-side span stamps only on sampled calls in timing mode: forward_impl clears tstamps otherwise)
    if (!c->keep_errs) a.errs = nullptr;
    // band of the fp32 maximum that is re-scored exactly: the stream's rounding (<= 2e-5 * alpha measured) plus two
    // cells' weight -- an ill-conditioned projection (scene point next to the camera centre) can put a cell on the other
    // side of tau under fp32, which moves a score by alpha / (H*W); on small grids that exceeds alpha * 1e-3
    a.margin = p->rescore_margin > 0 ? p->rescore_margin : fabsf(p->inlier_alpha) * (ESAC_DEFAULT_MARGIN + 2.0f / (float)P);
    c->lastN = p->N; c->lastH = p->H; c->lastW = p->W;
    *out = a;
    return 0;
}

// The forward path's team request: off while the context is latched (two team time-outs in a row, see forward_impl).  The latch is
// the FORWARD path's: the training path's slot teams have a switch of their own (esac_hip_ctx::slot_teams).
static void forward_team(const esac_hip_ctx* c, KArgs& a) {
    if (c->team_latched_off) a.team = 0;
    if (a.flags & ESAC_FLAG_REFINE_SOLO) {
        a.team = 0;
        a.solo = 1;
    }
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-200 - (int)e, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// A launch that (re)samples the hypotheses: the status word (out-of-range hypAssignment) is tagged with ITS epoch, and
// every later stage / check on this context compares against that -- not against the epoch of whatever call came last.
static void mark_sampling(esac_hip_ctx* c, KArgs& a) {
    c->sample_epoch = a.epoch;
    a.sample_epoch = a.epoch;
}

// stage entry points: validate, make the context's GPU current, launch one phase on the caller's stream
template <typename Launch>
static int run_stage(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, void* stream,
                     const char* what, Launch launch) {
    if (!c) return fail(-1, "null context");
    DeviceGuard guard(c->device);
    KArgs a;
    int rc = make_args(c, d_sc, d_assign, p, &a);
    if (rc) return rc;
    launch(c, a, (hipStream_t)stream);
    return check_launch(what);
}
extern "C" int esac_hip_sample(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, void* stream) {
    return run_stage(c, d_sc, d_assign, p, stream, "k_sample", [](esac_hip_ctx* cc, const KArgs& a0, hipStream_t s) {
        KArgs a = a0;
        mark_sampling(cc, a);
        cc->rt32_stale = false;
        launch_sample(a, s);
    });
}
extern "C" int esac_hip_score(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, void* stream) {
    return run_stage(c, d_sc, d_assign, p, stream, "k_score_fast", [](esac_hip_ctx* cc, const KArgs& a, hipStream_t s) {
        if (cc->rt32_stale) {  // hypotheses came from esac_hip_write_hyps: their fp32 [R|t] rows need the maps' origins
            launch_hyps_to_rt32(a, s);
            cc->rt32_stale = false;
        }
        launch_score(a, s);
    });
}
extern "C" int esac_hip_select(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, void* stream) {
    return run_stage(c, d_sc, d_assign, p, stream, "k_select_rescore", [](esac_hip_ctx* cc, const KArgs& a, hipStream_t s) {
        if (cc->rt32_stale) {  // hypotheses came from esac_hip_write_hyps: their rotation matrices have not been formed yet
            launch_hyps_to_rt32(a, s);
            cc->rt32_stale = false;
        }
        launch_select_rescore(a, s);
    });
}
extern "C" int esac_hip_refine(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, void* stream) {
    return run_stage(c, d_sc, d_assign, p, stream, "k_refine", [](esac_hip_ctx* cc, const KArgs& a0, hipStream_t s) {
        KArgs a = a0;
        forward_team(cc, a);
        cc->refine_tag = launch_refine(a, s);
    });
}
extern "C" int esac_hip_score_exact(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p, void* stream) {
    return run_stage(c, d_sc, d_assign, p, stream, "k_rescore(all)", [](esac_hip_ctx* cc, const KArgs& a, hipStream_t s) {
        if (cc->rt32_stale) {
            launch_hyps_to_rt32(a, s);
            cc->rt32_stale = false;
        }
        launch_rescore_all(a, s);
        launch_stats_exact(a, s);  // softmax statistics of the exact scores (the record's probability / entropy)
    });
}

// Blocking calls: the last kernel stores the record and the call's epoch word into pinned host memory (one ESAC_PIN_DOUBLES slot
// per frame: record, epoch word, status word, check word) and the host polls.  A slot has landed when its epoch word is this
// call's AND its check word fits the other 34 words (the kernel stores them without a fence: esac_kernels.hpp, pin_mix).
static int wait_record(esac_hip_ctx* c, hipStream_t s, int B, double want, const char* who) {
    auto all_landed = [&]() {
        for (int b = 0; b < B; b++) {
            const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(c->h_pin + (size_t)b * ESAC_PIN_DOUBLES);
            if (*(const volatile double*)(c->h_pin + (size_t)b * ESAC_PIN_DOUBLES + 32) != want) return false;
            unsigned long long h = 0;
            for (int k = 0; k < 34; k++) h ^= pin_mix(w[k], k);
            if (h != w[34]) return false;
        }
        return true;
    };
    bool landed = false;
    if (c->wait_mode == ESAC_WAIT_BLOCK) {
        HIP_OK(hipStreamSynchronize(s));
        landed = all_landed();
    } else {
        const bool yield = c->wait_mode == ESAC_WAIT_YIELD;
        for (long spins = 0; spins < 200000000L; spins++) {
            if (all_landed()) {
                landed = true;
                break;
            }
            if (yield) sched_yield();
            if ((spins & (yield ? 63 : 1023)) == (yield ? 63 : 1023) && hipStreamQuery(s) == hipSuccess) {  // stream idle: kernels are done (or failed)
                landed = all_landed();
                break;
            }
        }
    }
    if (!landed) {
        HIP_OK(hipStreamSynchronize(s));
        if (!all_landed()) return fail(-9, "%s did not deliver a result record", who);
    }
    return 0;
}

static int forward_impl(esac_hip_ctx* c, const float* d_sc, long long sc_frame_stride, const int64_t* d_assign,
                        const esac_hip_params* p, int B, void* stream, double* d_scores_out, double* d_result_out,
                        double* h_result_out) {
    if (!c) return fail(-1, "null context");
    const double t_entry = now_ns();
    DeviceGuard guard(c->device);
    KArgs a;
    int rc = make_args(c, d_sc, d_assign, p, &a, B, sc_frame_stride);
    if (rc) return rc;
    if (c->team_latched_off && ++c->solo_since_latch > ESAC_TEAM_REARM_CALLS) {  // (blocking or not: every forward call counts)
        c->team_latched_off = false;               // try a team again; one more time-out latches at once
        c->team_strikes = ESAC_TEAM_STRIKES - 1;
    }
    forward_team(c, a);
    c->host_ns[6] = t_entry;
    c->host_ns[0] = now_ns() - t_entry;
    hipStream_t s = (hipStream_t)stream;
    a.scores_user = d_scores_out;
    a.result_user = d_result_out;
    a.result_pin = h_result_out ? c->d_pin : nullptr;
    // ESAC_FLAG_AUTO_EXACT: the guaranteed routes where they are free (include/esac_hip.h)
    if ((a.flags & ESAC_FLAG_AUTO_EXACT) && B == 1 && a.E == 1 && (long long)a.N * a.H * a.W <= ESAC_AUTO_EXACT_MAX_WORK)
        a.flags |= ESAC_FLAG_EXACT_SCORES | ESAC_FLAG_EXACT_SAMPLING;
    // events and stamps cost GPU time themselves (an empty event pair reads ~5 us): sample every timing_period-th call
    const bool tm = c->timing && (c->timing_calls++ % c->timing_period) == 0;
    const bool exact = (a.flags & ESAC_FLAG_EXACT_SCORES) != 0;
    // device-side span stamps: only the per-hypothesis stream (k_score_fast) writes them, and k_select_rescore reduces them --
    // a call whose selection runs in the refinement kernel's prologue keeps ITS launch sequence under timing (the phase
    // events then bracket what an untimed call runs) and takes no stamps
    if (!tm || a.partials || exact) a.tstamps = nullptr;
    if (a.tstamps) {
        KArgs probe = a;
        probe.tstamps = nullptr;
        if (c->fold_select && refine_folds_select(probe)) a.tstamps = nullptr;
    }
    // Several experts: the sampler's straggler chain (wrong-expert hypotheses, which practically never win) runs BESIDE the scoring,
    // selection and refinement of what the first pass settled (KArgs::spec_mode; k_spec_join makes the outputs the serial order's)
    // (not where the selection runs in the refinement kernel's prologue -- a single frame of <= 256 hypotheses on a team: that
    // route re-scores its contenders member by member, another summation order than k_select_rescore's and k_spec_join's)
    const bool spec = !c->spec_off && B == 1 && !exact && sample_can_split(a) && !a.partials && (long long)a.H * a.W < 32768 &&
                      !(c->fold_select && refine_folds_select(a));
    c->last_spec_epoch = 0;
    bool spec_ok = spec;
    if (spec_ok && s != nullptr) {  // a stream that is being captured into a graph takes no launches on other streams beside it
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) spec_ok = false;
        (void)hipGetLastError();
    }
    if (spec_ok && !c->side) {
        HIP_OK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    }
    if (spec_ok && !c->side2) {
        HIP_OK(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
    }
    if (tm) HIP_OK(hipEventRecord(c->ev[0], s));
    c->rt32_stale = false;
    mark_sampling(c, a);
    if (spec_ok) {
        c->spec_calls++;
        c->last_spec_epoch = a.epoch;
        a.tstamps = nullptr;
        a.fold_select = 0;
        a.spec_flag = c->ws.spec_flag;
        KArgs chain;
        int chain_waves = 0;
        launch_sample_split(a, s, &chain, &chain_waves);
        if ((rc = check_launch("k_sample (first pass)"))) return rc;
        c->host_ns[1] = now_ns() - t_entry;
        if (tm) HIP_OK(hipEventRecord(c->ev[1], s));
        // The two streams hand over through WORDS in device memory (spec_state[3]: "the chain may start", [4]: "the chain is
        // done"), not through events: an event between two streams costs the waiting side 8-13 us on this platform even when it is
        // long satisfied (profiles/r06_ab_speculation.txt).  Whoever waits is enqueued BEHIND the launch it waits for (host order
        // below), so that even two streams that share a hardware queue cannot wait for each other; every wait is bounded in wall time.
        KArgs as = a;  // the settled hypotheses: score, selection, refinement of their winner -- no record leaves the workspace
        as.spec_mode = 1;
        as.result_user = nullptr;
        as.result_pin = nullptr;
        launch_score(as, s);
        if ((rc = check_launch("k_score_fast (settled)"))) return rc;
        c->host_ns[2] = now_ns() - t_entry;
        if (tm) HIP_OK(hipEventRecord(c->ev[2], s));
        // The chain starts when the speculative refinement has its CUs, not when the first pass is done: its thousands of
        // single-wavefront workgroups fill every SIMD of the chip, and whatever the launch stream starts while it is in full
        // swing finds no CU to run on until it has drained (measured: started behind the first pass, the selection took 27 us
        // instead of 10; started behind the score kernel, the refinement's team waited 34 us for its CUs).
        // The SELECTION among the settled hypotheses is not on the critical path either (round 6, second half).  The refinement starts
        // from the fp32 argmax of the settled hypotheses (spec_mode 2: spec_pick_fast) right behind the score kernel; the selection
        // kernel (band, exact re-scores) and the join behind it run on a second stream of the context's own, the join resident and
        // waiting when the refinement and the chain finish ("the refinement is done": spec_state[6]); the gated second refinement
        // on the caller's stream waits for the join's verdict ("the join is done": spec_state[7]).  Every waiter is enqueued
        // behind what it waits for.  (With selection and join on the caller's stream, in front of and behind the refinement:
        // cfg3 0.1404 ms against 0.1288, cfg4 0.1915 against 0.184, same box -- profiles/r06_ab_select_beside.txt.)
        if (tm) HIP_OK(hipEventRecord(c->ev[3], s));
        KArgs ar = as;
        ar.spec_mode = 2;
        ar.spec_debug = c->spec_second_best ? 1 : 0;
        c->refine_tag = launch_refine(ar, s);  // (its first workgroup opens the chain: spec_open_chain)
        c->refine_was_team = refine_team_members(ar) > 0;
        if ((rc = check_launch("k_refine (speculative)"))) return rc;
        // (host order: the selection first -- the join waits behind it; a launch call is 3-4 us of host time, and the chain's five
        // in front of it would hold the selection back by 20 us.  The JOIN is enqueued behind the chain it waits for.)
        launch_spec_wait(a, 3, c->side2);
        launch_select_rescore(as, c->side2);
        if ((rc = check_launch("k_select_rescore (settled)"))) return rc;
        launch_spec_wait(a, 3, c->side);
        launch_sample_stragglers_on(chain, chain_waves, c->side);
        if (!c->spec_lose_chain) launch_score_stragglers(a, c->side);  // behind the chain on the side stream; its last workgroup writes "the chain is done"
        if ((rc = check_launch("straggler chain"))) return rc;
        launch_spec_join(a, c->side2);
        if ((rc = check_launch("k_spec_join"))) return rc;
        {
            // The second refinement is enqueued NOW and returns at once unless the join marked the speculation as failed
            // (KArgs::spec_gate): a failed speculation then costs the refinement, not a host round trip on top of it (and an
            // asynchronous call could not look at the join's verdict anyway); a speculation that held has delivered its record
            // before the gate opens
            KArgs ag = a;
            ag.spec_gate = 1;
            const unsigned long long tag2 = launch_refine(ag, s);
            (void)tag2;  // (esac_hip_check follows the speculative launch's tag: a team time-out there is the common case of the two)
            if ((rc = check_launch("k_refine (gated)"))) return rc;
        }
        c->host_ns[3] = now_ns() - t_entry;
    } else {
        launch_sample(a, s);
        if ((rc = check_launch("k_sample"))) return rc;
        c->host_ns[1] = now_ns() - t_entry;
        if (tm) HIP_OK(hipEventRecord(c->ev[1], s));
        // ESAC_FLAG_EXACT_SCORES: every hypothesis scored in the reference's arithmetic (esac_util.h:235-260), softmax
        // statistics from those scores -- the score vector, probability and entropy are then the reference's own values
        if (exact) launch_rescore_all(a, s);
        else       launch_score(a, s);
        if ((rc = check_launch(exact ? "k_rescore(all)" : "k_score_fast"))) return rc;
        c->host_ns[2] = now_ns() - t_entry;
        if (tm) HIP_OK(hipEventRecord(c->ev[2], s));
        // a single frame of <= 256 hypotheses that a team refines: the selection runs in that kernel's prologue
        a.fold_select = !c->fold_select ? 0 : refine_folds_select(a) ? 1 : refine_folds_exact_stats(a) ? 2 : 0;
        if (exact) {
            if (a.fold_select != 2) launch_stats_exact(a, s);
        } else if (!a.fold_select) {
            launch_select_rescore(a, s);
        }
        if ((rc = check_launch(exact ? "k_stats_exact" : "k_select_rescore"))) return rc;
        if (tm) HIP_OK(hipEventRecord(c->ev[3], s));
        c->refine_tag = launch_refine(a, s);
        c->refine_was_team = refine_team_members(a) > 0;
        if ((rc = check_launch("k_refine"))) return rc;
        c->host_ns[3] = now_ns() - t_entry;
    }
    if (tm) {
        HIP_OK(hipEventRecord(c->ev[4], s));
        // an EMPTY interval: what two adjacent hipEventRecord calls measure with nothing in between,
        // i.e. the part of every bracketed phase that is not kernel time
        HIP_OK(hipEventRecord(c->ev[5], s));
        HIP_OK(hipEventRecord(c->ev[6], s));
        c->ev_valid = true;
    }
    if (h_result_out) {
        // the refinement kernel stores the record and then the epoch word into pinned host memory
        // (one ESAC_PIN_DOUBLES slot per frame: record, epoch word, status word)
        if ((rc = wait_record(c, s, B, c->epoch, "esac_hip_forward: the refinement kernel"))) return rc;
        if (spec_ok && c->h_pin[33] == 5.0) {
            // the context's own stream never reported the straggler chain as done within 20 ms (it shares a hardware queue with
            // the caller's stream and something else is holding that queue, or its launch failed): no more speculation on this
            // context, and this call again -- the serial route
            c->spec_off = c->spec_env_off = true;
            HIP_OK(hipStreamSynchronize(c->side));
            if (c->side2) HIP_OK(hipStreamSynchronize(c->side2));
            HIP_OK(hipStreamSynchronize(s));
            return forward_impl(c, d_sc, sc_frame_stride, d_assign, p, B, stream, d_scores_out, d_result_out, h_result_out);
        }
        c->host_ns[4] = now_ns() - t_entry;
        bool team_failed = false;
        for (int b = 0; b < B; b++) team_failed |= c->h_pin[(size_t)b * ESAC_PIN_DOUBLES + 33] == 3.0;
        const bool was_team = refine_team_members(a) > 0;
        if (team_failed && was_team) {
            // the members of a team did not all become resident in time (a shared or partitioned GPU, or the caller's own
            // kernels on another stream holding the CUs): the same refinement(s) in ONE workgroup each -- the hypotheses, scores
            // and selection of this call are still in the workspace.  Twice in a row and the context stops asking for teams
            // (every call would pay the time-out first) until it is re-armed.
            c->team_fallbacks++;
            if (++c->team_strikes >= ESAC_TEAM_STRIKES && !c->team_latched_off) {
                c->team_latched_off = true;
                c->solo_since_latch = 0;
            }
            c->epoch += 1.0;
            a.epoch = c->epoch;
            a.team = 0;
            a.solo = 1;
            if (a.fold_select) {  // the selection was that kernel's too
                if (a.fold_select == 2) launch_stats_exact(a, s);
                else                    launch_select_rescore(a, s);
                a.fold_select = 0;
            }
            c->refine_tag = launch_refine(a, s);
            if ((rc = check_launch("k_refine (one workgroup, after a team time-out)"))) return rc;
            if ((rc = wait_record(c, s, B, c->epoch, "esac_hip_forward: the refinement kernel"))) return rc;
        } else if (was_team) {
            c->team_strikes = 0;
        }
        __sync_synchronize();
        bool bad_assign = false;
        for (int b = 0; b < B; b++) {
            memcpy(h_result_out + (size_t)b * ESAC_RES_DOUBLES, (const void*)(c->h_pin + (size_t)b * ESAC_PIN_DOUBLES),
                   ESAC_RES_DOUBLES * sizeof(double));
            bad_assign |= c->h_pin[(size_t)b * ESAC_PIN_DOUBLES + 33] == 1.0;
        }
        for (int b = 0; b < B; b++)
            if (c->h_pin[(size_t)b * ESAC_PIN_DOUBLES + 33] == 3.0)
                return fail(-12, "esac_hip_forward: the cooperating refinement workgroups could not synchronise (not all of them became resident)");
        if (bad_assign)
            return fail(-10, "hypAssignment holds a value outside [0,%d) (device-resident tensor; such hypotheses were scored against expert 0)", p->E);
    }
    c->host_ns[5] = now_ns() - t_entry;
    if (h_result_out) {  // running sums (seven additions: the caller's timed loop is not touched by reading them later)
        for (int k = 0; k < 6; k++) c->host_sum[k] += c->host_ns[k];
        if (c->host_last_return > 0) c->host_sum[6] += t_entry - c->host_last_return;  // the caller's time between two calls
        c->host_last_return = t_entry + c->host_ns[5];
        c->host_n++;
    }
    return 0;
}

extern "C" int esac_hip_host_turn_mean(esac_hip_ctx* c, double out_ns[8], int reset) {
    if (!c || !out_ns) return fail(-1, "esac_hip_host_turn_mean: null argument");
    const double n = c->host_n > 0 ? (double)c->host_n : 1.0;
    for (int k = 0; k < 6; k++) out_ns[k] = c->host_sum[k] / n;
    out_ns[6] = c->host_n > 1 ? c->host_sum[6] / (double)(c->host_n - 1) : 0.0;
    out_ns[7] = (double)c->host_n;
    if (reset) {
        for (double& v : c->host_sum) v = 0;
        c->host_n = 0;
        c->host_last_return = 0;
    }
    return 0;
}

extern "C" int esac_hip_host_turn(esac_hip_ctx* c, double out_ns[8]) {
    if (!c || !out_ns) return fail(-1, "esac_hip_host_turn: null argument");
    for (int k = 0; k < 8; k++) out_ns[k] = c->host_ns[k];
    return 0;
}

// Mean GPU time of each stage of the forward chain for THIS input: the chain runs once, then every stage is launched
// `reps` times back to back between one pair of hipEvents on `stream` (stages are idempotent given their inputs).
// A host-side loop around single launches cannot do this for ~5 us kernels: it is bound by the caller's launch rate.
extern "C" int esac_hip_time_stages(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p,
                                    void* stream, int reps, float out_ms[4]) {
    if (!c || !out_ms || reps < 1) return fail(-1, "esac_hip_time_stages: bad argument");
    DeviceGuard guard(c->device);
    KArgs a;
    int rc = make_args(c, d_sc, d_assign, p, &a);
    if (rc) return rc;
    a.tstamps = nullptr;
    forward_team(c, a);
    hipStream_t s = (hipStream_t)stream;
    c->rt32_stale = false;
    mark_sampling(c, a);
    if ((a.flags & ESAC_FLAG_AUTO_EXACT) && a.E == 1 && (long long)a.N * a.H * a.W <= ESAC_AUTO_EXACT_MAX_WORK)
        a.flags |= ESAC_FLAG_EXACT_SCORES | ESAC_FLAG_EXACT_SAMPLING;
    const bool exact = (a.flags & ESAC_FLAG_EXACT_SCORES) != 0;
    // as esac_hip_forward would run it: stage 2 is then part of stage 3 (reads 0)
    a.fold_select = !c->fold_select ? 0 : refine_folds_select(a) ? 1 : refine_folds_exact_stats(a) ? 2 : 0;
    auto stage = [&](int k) {
        switch (k) {
            case 0: launch_sample(a, s); break;
            case 1: if (exact) launch_rescore_all(a, s); else launch_score(a, s); break;
            case 2: if (exact) { if (a.fold_select != 2) launch_stats_exact(a, s); } else if (!a.fold_select) launch_select_rescore(a, s); break;
            default: c->refine_tag = launch_refine(a, s); break;
        }
    };
    for (int k = 0; k < 4; k++) stage(k);
    if ((rc = check_launch("forward chain"))) return rc;
    hipEvent_t ev[8];
    for (auto& e : ev) HIP_OK(hipEventCreate(&e));
    for (int k = 0; k < 4; k++) {
        stage(k);  // one untimed launch: the timed ones then start from the same (warm) state
        HIP_OK(hipEventRecord(ev[2 * k], s));
        for (int r = 0; r < reps; r++) stage(k);
        HIP_OK(hipEventRecord(ev[2 * k + 1], s));
    }
    if ((rc = check_launch("stage timing"))) return rc;
    HIP_OK(hipEventSynchronize(ev[7]));
    for (int k = 0; k < 4; k++) {
        HIP_OK(hipEventElapsedTime(&out_ms[k], ev[2 * k], ev[2 * k + 1]));
        out_ms[k] /= (float)reps;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return 0;
}

// Multi-GPU exchange (esac_amd/distributed.py): winner among the per-rank records of the all-reduced buffer.
extern "C" int esac_hip_pick_record(esac_hip_ctx* c, const double* d_records, int world, void* stream, double* h_record_out, double* d_zero,
                                    int n_zero) {
    if (!c || !d_records || !h_record_out || world < 1 || n_zero < 0 || (n_zero > 0 && !d_zero))
        return fail(-1, "esac_hip_pick_record: bad argument");
    DeviceGuard guard(c->device);
    hipStream_t s = (hipStream_t)stream;
    c->epoch += 1.0;
    const double want = c->epoch;
    launch_pick_record(d_records, world, c->d_pin, want, d_zero, n_zero, s);
    int rc = check_launch("k_pick_record");
    if (rc) return rc;
    volatile double* word = c->h_pin + 32;
    bool landed = false;
    for (long spins = 0; spins < 200000000L; spins++) {
        if (*word == want) {
            landed = true;
            break;
        }
        if ((spins & 1023) == 1023 && hipStreamQuery(s) == hipSuccess) {
            landed = *word == want;
            break;
        }
    }
    if (!landed) {
        HIP_OK(hipStreamSynchronize(s));
        if (*word != want) return fail(-9, "esac_hip_pick_record: the kernel did not deliver a record");
    }
    __sync_synchronize();
    memcpy(h_record_out, (const void*)c->h_pin, ESAC_RES_DOUBLES * sizeof(double));
    if (c->h_pin[33] == 3.0)
        return fail(-12, "esac_hip_pick_record: the refinement team of at least one rank timed out (its record carries ESAC_RES_VALID = 3); "
                         "every rank sees the same records: run the frame again with ESAC_FLAG_REFINE_SOLO");
    if (c->h_pin[33] == 2.0) return fail(-11, "esac_hip_pick_record: no rank produced a hypothesis");
    return 0;
}

// ---- the one collective of the multi-GPU path, straight on RCCL (esac_amd/distributed.py bootstraps the id over the caller's
// process group; the per-frame data path then never enters torch.distributed, whose enqueue of a collective costs the host
// 20-27 us a call: bench.py sharded_world1).  RCCL is bound at the first esac_hip_comm_* call, not at link time: a single-GPU
// process never maps the 570 MB library, and one that already holds a copy (torch ships its own librccl.so.1) gets that copy.
#define RCCL_BOUND() \
    if (!rccl().ok) return fail(-14, "RCCL is not available in this process (librccl.so.1 could not be loaded)")
#define NCCL_OK(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) return fail(-300 - (int)_r, "%s: %s", #expr, rccl().error_string(_r));       \
    } while (0)
extern "C" int esac_hip_comm_unique_id(void* out, size_t bytes) {
    if (!out || bytes < sizeof(ncclUniqueId)) return fail(-1, "esac_hip_comm_unique_id: need %zu bytes", sizeof(ncclUniqueId));
    RCCL_BOUND();
    ncclUniqueId id;
    NCCL_OK(rccl().get_unique_id(&id));
    memcpy(out, &id, sizeof(id));
    return 0;
}
extern "C" int esac_hip_comm_init(esac_hip_ctx* c, int nranks, int rank, const void* unique_id, size_t bytes) {
    if (!c || !unique_id || bytes < sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(-1, "esac_hip_comm_init: bad argument");
    RCCL_BOUND();
    DeviceGuard guard(c->device);
    drop_comm(c);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    NCCL_OK(rccl().comm_init_rank(&c->comm, nranks, id, rank));
    c->comm_ranks = nranks;
    c->comm_rank = rank;
    return 0;
}
extern "C" int esac_hip_comm_destroy(esac_hip_ctx* c) {
    if (!c) return fail(-1, "null context");
    DeviceGuard guard(c->device);
    drop_comm(c);
    return 0;
}
extern "C" int esac_hip_allreduce_sum(esac_hip_ctx* c, double* d_buf, size_t count, void* stream) {
    if (!c || !d_buf) return fail(-1, "esac_hip_allreduce_sum: null argument");
    if (!c->comm) return fail(-13, "esac_hip_allreduce_sum: no communicator (esac_hip_comm_init)");
    DeviceGuard guard(c->device);
    NCCL_OK(rccl().all_reduce(d_buf, d_buf, count, NCCL_DOUBLE, NCCL_SUM, c->comm, (hipStream_t)stream));
    return 0;
}
// What the communicator ITSELF reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), beside what the context was told
// and the GPU it is bound to: the proof a multi-GPU bench line carries of how many ranks RCCL saw (bench.py: ranks_seen).
extern "C" int esac_hip_comm_info(esac_hip_ctx* c, int32_t out[4]) {
    if (!c || !out) return fail(-1, "esac_hip_comm_info: null argument");
    if (!c->comm) return fail(-13, "esac_hip_comm_info: no communicator (esac_hip_comm_init)");
    int count = c->comm_ranks, rank = c->comm_rank, dev = -1;
    if (rccl().comm_count) NCCL_OK(rccl().comm_count(c->comm, &count));
    if (rccl().comm_user_rank) NCCL_OK(rccl().comm_user_rank(c->comm, &rank));
    if (rccl().comm_cu_device) NCCL_OK(rccl().comm_cu_device(c->comm, &dev));
    out[0] = count;
    out[1] = rank;
    out[2] = dev;
    out[3] = c->device;
    return 0;
}

extern "C" int esac_hip_forward(esac_hip_ctx* c, const float* d_sc, const int64_t* d_assign, const esac_hip_params* p,
                                void* stream, double* d_scores_out, double* d_result_out, double* h_result_out) {
    return forward_impl(c, d_sc, 0, d_assign, p, 1, stream, d_scores_out, d_result_out, h_result_out);
}

extern "C" int esac_hip_forward_batch(esac_hip_ctx* c, int B, const float* d_sc, int64_t sc_frame_stride,
                                      const int64_t* d_assign, const esac_hip_params* p, void* stream,
                                      double* d_scores_out, double* d_result_out, double* h_result_out) {
    return forward_impl(c, d_sc, (long long)sc_frame_stride, d_assign, p, B, stream, d_scores_out, d_result_out, h_result_out);
}

// ---------------------------------------------------------------- training path
// `cap` = slots (hypotheses with p >= PROB_THRESH) the slab workspace must hold; it only ever grows
static int ensure_bws(esac_hip_ctx* c, int N, int P, int cap) {
    const bool lists = P > ESAC_REFINE_LDS_CAP;
    if (N <= c->bN && P <= c->bP && cap <= c->bcap && (!lists || c->b_lists)) return 0;
    HIP_OK(hipDeviceSynchronize());
    const int nN = N > c->bN ? N : c->bN, nP = P > c->bP ? P : c->bP, ncap = cap > c->bcap ? cap : c->bcap;
    const bool nlists = lists || c->b_lists;
    free_bws(c);
    int rc = 0;
    rc |= alloc(&c->bws.sel, (size_t)nN);
    rc |= alloc(&c->bws.n_sel, (size_t)4);
    rc |= alloc(&c->bws.probs, (size_t)nN);
    rc |= alloc(&c->bws.losses, (size_t)nN);
    rc |= alloc(&c->bws.ref_hyps, (size_t)nN * 6);
    rc |= alloc(&c->bws.sgrad, (size_t)nN);
    const size_t rows = (size_t)(nN < ESAC_BWD_MAX_SLOTS ? nN : ESAC_BWD_MAX_SLOTS);  // small per-slot tables: worst case
    rc |= alloc(&c->bws.dloss, rows * 6);
    rc |= alloc(&c->bws.maps, (size_t)ncap * 2 * nP);
    rc |= alloc(&c->bws.map_info, rows * 4);
    if (nlists) {
        char* cl = nullptr;
        rc |= alloc(&cl, (size_t)ncap * ((size_t)nP + 2048) * 16);  // corr_entries(P) < P + 2048 per slot
        c->bws.corr_lists = cl;
    }
    rc |= alloc(&c->bws.grad1, (size_t)ncap * nP * 3);
    rc |= alloc(&c->bws.grad2, (size_t)ncap * nP * 3);
    rc |= alloc(&c->bws.out, (size_t)4);
    rc |= alloc(&c->bws.arrived, (size_t)1);
    rc |= alloc(&c->bws.team_gran, (size_t)ncap * 2 * ESAC_REFINE_TEAM_MAX * 32 * 2);  // 16-byte granules: [slot][parity][member][value]
    if (rc) {
        free_bws(c);
        return rc;
    }
    HIP_OK(hipMemset(c->bws.n_sel, 0, 4 * sizeof(int)));
    HIP_OK(hipMemset(c->bws.arrived, 0, sizeof(int)));
    HIP_OK(hipMemset(c->bws.team_gran, 0, (size_t)ncap * 2 * ESAC_REFINE_TEAM_MAX * 32 * 2 * sizeof(double)));
    c->bN = nN; c->bP = nP; c->bcap = ncap; c->b_lists = nlists;
    return 0;
}

// general 4x4 inverse, Gauss-Jordan with partial pivoting (cv::Mat::inv() of trans2pose, esac_util.h:557)
static bool inv4_host(const double A[16], double Ai[16]) {
    double M[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            M[i][j] = A[4 * i + j];
            M[i][4 + j] = i == j;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (M[piv][col] == 0) return false;
        if (piv != col)
            for (int j = 0; j < 8; j++) {
                const double t = M[piv][j];
                M[piv][j] = M[col][j];
                M[col][j] = t;
            }
        const double d = 1.0 / M[col][col];
        for (int j = 0; j < 8; j++) M[col][j] *= d;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            const double f = M[r][col];
            if (f == 0) continue;
            for (int j = 0; j < 8; j++) M[r][j] -= f * M[col][j];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) Ai[4 * i + j] = M[i][4 + j];
    return true;
}

// nearest rotation of a 3x3 (orthogonal polar factor = U*Vt of its SVD, what cv::Rodrigues applies to a matrix
// input): Newton iteration X <- (X + X^-T) / 2, quadratic from the ~1e-7 non-orthonormality of a float pose
static void nearest_rotation_host(double R[9]) {
    for (int it = 0; it < 20; it++) {
        const double* a = R;
        const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
        const double c10 = a[2] * a[7] - a[1] * a[8], c11 = a[0] * a[8] - a[2] * a[6], c12 = a[1] * a[6] - a[0] * a[7];
        const double c20 = a[1] * a[5] - a[2] * a[4], c21 = a[2] * a[3] - a[0] * a[5], c22 = a[0] * a[4] - a[1] * a[3];
        const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
        if (det == 0) return;
        const double invT[9] = {c00 / det, c01 / det, c02 / det, c10 / det, c11 / det, c12 / det, c20 / det, c21 / det, c22 / det};
        double delta = 0;
        for (int k = 0; k < 9; k++) {
            const double n = 0.5 * (R[k] + invT[k]);
            delta += fabs(n - R[k]);
            R[k] = n;
        }
        if (delta < 1e-15) break;
    }
}

extern "C" int esac_hip_backward(esac_hip_ctx* c, const float* d_sc, float* d_out_gradients, const int64_t* d_assign,
                                 const float* h_gt_pose, float w_loss_rot, float w_loss_trans, float loss_cut,
                                 const esac_hip_params* p, void* stream, double* h_out) {
    if (!d_out_gradients || !h_gt_pose) return fail(-1, "esac_hip_backward: null gradient tensor or ground-truth pose");
    if (!c) return fail(-1, "null context");
    DeviceGuard guard(c->device);
    KArgs a;
    int rc = make_args(c, d_sc, d_assign, p, &a);
    if (rc) return rc;
    if (p->E > 65535) return fail(-4, "esac_hip_backward: at most 65535 experts (one grid row per expert in the accumulation kernel)");
    if (p->d_hyp_index || p->hyp_offset)
        return fail(-4, "esac_hip_backward: sharded calls are not supported (the expectation needs every hypothesis)");
    const int P = p->H * p->W;
    // Slot workspace (two 3P-double slabs + two inlier maps per slot).  How many hypotheses reach PROB_THRESH is only
    // known on the device: a blocking call starts from what earlier calls needed (at least 64 slots) and, when the
    // selection overflows it, grows the workspace and runs selection..accumulation again -- the accumulation kernel
    // adds nothing on overflow, so the caller's tensor is untouched by the aborted pass.  An asynchronous call
    // (h_out == NULL) cannot look at the count and reserves the worst case min(N, 1000).
    const int worst = p->N < ESAC_BWD_MAX_SLOTS ? p->N : ESAC_BWD_MAX_SLOTS;
    int cap = worst;
    if (h_out) {
        cap = c->bcap > 64 ? c->bcap : 64;
        if (cap > worst) cap = worst;
    }
    double Ti[16], gt[16], gt_pose[6];
    for (int i = 0; i < 16; i++) gt[i] = (double)h_gt_pose[i];
    if (!inv4_host(gt, Ti)) return fail(-4, "esac_hip_backward: the ground-truth pose is singular");
    double Rg[9] = {Ti[0], Ti[1], Ti[2], Ti[4], Ti[5], Ti[6], Ti[8], Ti[9], Ti[10]};
    nearest_rotation_host(Rg);
    rodrigues_mat2vec(Rg, gt_pose);
    gt_pose[3] = Ti[3]; gt_pose[4] = Ti[7]; gt_pose[5] = Ti[11];
    a.tstamps = nullptr;
    hipStream_t s = (hipStream_t)stream;
    c->rt32_stale = false;
    mark_sampling(c, a);
    launch_sample(a, s);                                        // esac.cpp:276
    if ((rc = check_launch("k_sample"))) return rc;
    launch_rescore_all(a, s);                                   // esac.cpp:295-316, reference arithmetic for every hypothesis
    if ((rc = check_launch("k_rescore(all)"))) return rc;
    // Slot refinement (esac.cpp:328-347): by teams of 8 (esac_refine_team.hip) when THIS call's selection holds few enough slots
    // that every team has an XCD's CUs to itself (<= 32), one workgroup per slot otherwise.  The count is only known on the
    // device, so both launches are issued and each returns at once when the other one's case applies (team_max_slots): the
    // route is a function of the call's own inputs, not of what an earlier call on the context selected.  A blocking call can
    // refine again with one workgroup per slot should a team time out; an asynchronous one cannot and does not use teams.
    bool use_teams = h_out && c->slot_teams;
    for (int attempt = 0;; attempt++) {
        if ((rc = ensure_bws(c, p->N, P, cap))) return rc;
        a.bwd = c->bws;
        a.bwd.cap = cap;
        a.bwd.team = 0;
        a.bwd.team_tag = 0;
        a.bwd.out_grad = d_out_gradients;
        a.bwd.w_rot = (double)w_loss_rot;
        a.bwd.w_trans = (double)w_loss_trans;
        a.bwd.cut = (double)loss_cut;
        for (int i = 0; i < 16; i++) a.bwd.gt[i] = gt[i];
        for (int i = 0; i < 6; i++) a.bwd.gt_pose[i] = gt_pose[i];
        const bool teams = use_teams && refine_slots_can_team(a);
        a.bwd.team_max_slots = teams ? ESAC_SLOT_TEAMS_MAX : 0;
        launch_bwd_select(a, s);                                    // esac.cpp:319-331
        if ((rc = check_launch("k_bwd_select"))) return rc;
        if (teams) {
            launch_refine_slots_team(a, s);                         // a team per slot, when n_sel <= team_max_slots
            c->slot_team_calls++;
        }
        launch_refine_slots(a, s);                                  // one workgroup per slot otherwise
        if ((rc = check_launch("k_refine(slots)"))) return rc;
        launch_bwd_loss(a, s);                                      // esac.cpp:354-362 + dLoss + softmax derivative
        if ((rc = check_launch("k_bwd_loss"))) return rc;
        launch_bwd_paths(a, s);                                     // esac.cpp:375-463 (path I) and :470-488 (path II), side by side
        if ((rc = check_launch("k_bwd_paths"))) return rc;
        // the last workgroup of the accumulation hands the call's record (h_out's four values + "a slot team timed out") to the
        // pinned slot: no copies, no stream-completion round trip (two hipMemcpyAsync + hipStreamSynchronize before)
        KArgs acc = a;
        acc.result_pin = h_out ? c->d_pin : nullptr;
        launch_bwd_accumulate(acc, s);                              // esac.cpp:491-508
        if ((rc = check_launch("k_bwd_accumulate"))) return rc;
        if (!h_out) return 0;
        if ((rc = wait_record(c, s, 1, a.epoch, "esac_hip_backward: the accumulation kernel"))) return rc;
        __sync_synchronize();
        for (int k = 0; k < 4; k++) h_out[k] = c->h_pin[k];
        const bool team_failed = c->h_pin[4] == 1.0;
        if (teams && team_failed) {  // a team timed out: nothing was accumulated; one workgroup per slot from here on
            c->slot_team_fallbacks++;
            c->slot_teams = false;
            use_teams = false;
            attempt--;
            c->epoch += 1.0;
            a.epoch = c->epoch;
            continue;
        }
        c->last_nsel = (int)h_out[1];
        c->last_bwd_teams = teams && c->last_nsel <= ESAC_SLOT_TEAMS_MAX;
        const int needed = (int)h_out[1];
        if (needed <= cap || attempt >= 1) break;  // one retry suffices: the second pass is sized by the true count
        cap = needed + 31 > worst ? worst : (needed + 31) / 32 * 32;
        c->epoch += 1.0;
        a.epoch = c->epoch;
    }
    if (h_out[3] != 0.0)
        return fail(-10, "hypAssignment holds a value outside [0,%d) (device-resident tensor; such hypotheses were scored against expert 0)", p->E);
    return 0;
}

// Asynchronous calls (no host result) cannot report an out-of-range hypAssignment themselves: this waits for the
// device and returns -10 when the most recent call on the context flagged one, 0 otherwise.
extern "C" int esac_hip_check(esac_hip_ctx* c) {
    if (!c) return fail(-1, "null context");
    DeviceGuard guard(c->device);
    if (!c->ws.status) return 0;
    HIP_OK(hipDeviceSynchronize());
    unsigned long long st = 0, coop[2] = {0, 0};
    HIP_OK(hipMemcpy(&st, c->ws.status, sizeof(st), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(coop, c->ws.coop_counter, sizeof(coop), hipMemcpyDeviceToHost));
    if (c->refine_tag != 0 && coop[1] == c->refine_tag) {  // the failure word carries the tag of the launch that failed: only the most recent one counts
        // asynchronous calls learn of a team time-out here (or from the pick of the multi-GPU exchange, whose caller then asks here):
        // the same two-strikes latch as the blocking call's, so that a GPU whose CUs are held by someone else does not cost every
        // frame the 1 ms wait
        if (c->refine_was_team && c->checked_tag != c->refine_tag) {
            c->checked_tag = c->refine_tag;
            c->team_fallbacks++;
            if (++c->team_strikes >= ESAC_TEAM_STRIKES && !c->team_latched_off) {
                c->team_latched_off = true;
                c->solo_since_latch = 0;
            }
        }
        return fail(-12, "the cooperating refinement workgroups of the most recent call could not synchronise (not all of them became resident)");
    }
    if (st != 0 && (double)st == c->sample_epoch) return fail(-10, "hypAssignment held a value outside [0,E) in the most recent sampling call");
    return 0;
}

// Multi-GPU: this rank's share of a load-balanced split of the hypotheses, built on the device (one launch, no host
// round trip); see include/esac_hip.h.
extern "C" int esac_hip_shard_balanced(esac_hip_ctx* c, const int64_t* d_assign, int N, int E, int world, int rank, int expert_base,
                                       void* stream, int32_t* d_index_out, int64_t* d_assign_out, int32_t* d_info_out) {
    if (!c || !d_assign || !d_index_out || !d_assign_out) return fail(-1, "esac_hip_shard_balanced: null argument");
    if (N <= 0 || E <= 0 || E > ESAC_TILED_MAX_EXPERTS) return fail(-4, "esac_hip_shard_balanced: N=%d, E=%d (1 <= E <= %d)", N, E, ESAC_TILED_MAX_EXPERTS);
    if (world < 1 || rank < 0 || rank >= world) return fail(-4, "esac_hip_shard_balanced: rank %d of %d", rank, world);
    DeviceGuard guard(c->device);
    launch_shard_balanced(d_assign, N, E, world, rank, expert_base, d_index_out, d_assign_out, d_info_out, (hipStream_t)stream);
    return check_launch("k_shard_balanced");
}

extern "C" int esac_hip_set_wait(esac_hip_ctx* c, int mode) {
    if (!c) return fail(-1, "null context");
    if (mode != ESAC_WAIT_SPIN && mode != ESAC_WAIT_YIELD && mode != ESAC_WAIT_BLOCK) return fail(-4, "esac_hip_set_wait: unknown mode %d", mode);
    c->wait_mode = mode;
    return 0;
}

extern "C" int esac_hip_read(esac_hip_ctx* c, int which, void* h_dst, size_t bytes) {
    if (!c || !h_dst) return fail(-1, "esac_hip_read: null argument");
    DeviceGuard guard(c->device);
    const size_t N = (size_t)c->lastN, P = (size_t)c->lastH * c->lastW;
    const void* src = nullptr;
    size_t want = 0;
    switch (which) {
        case ESAC_BUF_HYPS: src = c->ws.hyps; want = N * 6 * sizeof(double); break;
        case ESAC_BUF_SAMPLE_XY: src = c->ws.sample_xy; want = N * 8 * sizeof(int32_t); break;
        case ESAC_BUF_TRIES: src = c->ws.tries; want = N * sizeof(int32_t); break;
        case ESAC_BUF_SCORES: src = c->ws.scores; want = N * sizeof(double); break;
        case ESAC_BUF_RESULT: src = c->ws.result; want = ESAC_RES_DOUBLES * sizeof(double); break;
        case ESAC_BUF_INLIER_MAP: {
            // the refinement kernel alternates between two map buffers; result[31] names the one that
            // holds the last ACCEPTED inlier set (-1: no re-fit was accepted -> all zeros)
            if (bytes != P) return fail(-7, "esac_hip_read: inlier map holds %zu bytes, caller asked for %zu", P, bytes);
            HIP_OK(hipDeviceSynchronize());
            double which_buf = -1;
            HIP_OK(hipMemcpy(&which_buf, c->ws.result + 31, sizeof(double), hipMemcpyDeviceToHost));
            if (which_buf < 0) {
                memset(h_dst, 0, P);
                return 0;
            }
            HIP_OK(hipMemcpy(h_dst, c->ws.inlier_map + (which_buf > 0.5 ? P : 0), P, hipMemcpyDeviceToHost));
            return 0;
        }
        case ESAC_BUF_INLIER_COUNTS: src = c->ws.inlier_counts; want = (ESAC_MAX_REF_STEPS + 1) * sizeof(int32_t); break;
        case ESAC_BUF_WINNER_ERRS:
            if (!c->keep_errs) return fail(-6, "esac_hip_read: the error image is only kept after esac_hip_set_debug(ctx, ESAC_DEBUG_ERROR_IMAGE)");
            src = c->ws.errs; want = P * sizeof(float); break;
        case ESAC_BUF_EXACT_FLAGS: src = c->ws.exact_flag; want = N; break;
        case ESAC_BUF_SPEC_FLAGS: src = c->ws.spec_flag; want = N; break;
        case ESAC_BUF_CYCLES: src = c->ws.cycles; want = 32 * sizeof(long long); break;
        case ESAC_BUF_BWD_TEAM_INFO: {
            if (bytes != 4 * sizeof(int32_t)) return fail(-7, "esac_hip_read: the slot-team info holds 16 bytes, caller asked for %zu", bytes);
            const int32_t info[4] = {c->last_bwd_teams ? 1 : 0, (int32_t)c->slot_team_calls, (int32_t)c->slot_team_fallbacks, (int32_t)c->last_nsel};
            memcpy(h_dst, info, sizeof(info));
            return 0;
        }
        case ESAC_BUF_SPEC_INFO: {
            if (bytes != 4 * sizeof(int32_t)) return fail(-7, "esac_hip_read: the speculation info holds 16 bytes, caller asked for %zu", bytes);
            double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c->ws.spec_state) {
                HIP_OK(hipDeviceSynchronize());
                HIP_OK(hipMemcpy(st, c->ws.spec_state, sizeof(st), hipMemcpyDeviceToHost));
            }
            const int32_t info[4] = {(int32_t)c->spec_calls, (int32_t)st[2], c->last_spec_epoch != 0 ? 1 : 0,
                                     c->last_spec_epoch != 0 && st[0] == c->last_spec_epoch ? 1 : 0};
            memcpy(h_dst, info, sizeof(info));
            return 0;
        }
        case ESAC_BUF_REFINE_INFO: {
            if (bytes != 8 * sizeof(int32_t)) return fail(-7, "esac_hip_read: the refinement info holds 32 bytes, caller asked for %zu", bytes);
            if (!c->ws.refine_info) return fail(-6, "esac_hip_read: buffer %d is empty (no call has run yet)", which);
            HIP_OK(hipDeviceSynchronize());
            int32_t info[8];
            HIP_OK(hipMemcpy(info, c->ws.refine_info, sizeof(info), hipMemcpyDeviceToHost));
            info[6] = (int32_t)((c->team_fallbacks & 0x3fffffff) | (c->team_latched_off ? 0x40000000 : 0));
            memcpy(h_dst, info, sizeof(info));
            return 0;
        }
        case ESAC_BUF_BWD_PROBS: src = c->bws.probs; want = N * sizeof(double); break;
        case ESAC_BUF_BWD_LOSSES: src = c->bws.losses; want = N * sizeof(double); break;
        case ESAC_BUF_BWD_REF_HYPS: src = c->bws.ref_hyps; want = N * 6 * sizeof(double); break;
        case ESAC_BUF_BWD_SCORE_GRADS: src = c->bws.sgrad; want = N * sizeof(double); break;
        case ESAC_BUF_BWD_SLOTS: src = c->bws.sel; want = N * sizeof(int32_t); break;
        case ESAC_BUF_BWD_SLOT_INFO: src = c->bws.map_info; want = (size_t)(c->lastN < ESAC_BWD_MAX_SLOTS ? c->lastN : ESAC_BWD_MAX_SLOTS) * 4 * sizeof(int32_t); break;
        case ESAC_BUF_BWD_DLOSS: src = c->bws.dloss; want = (size_t)(c->lastN < ESAC_BWD_MAX_SLOTS ? c->lastN : ESAC_BWD_MAX_SLOTS) * 6 * sizeof(double); break;
        case ESAC_BUF_BWD_PATH1:
        case ESAC_BUF_BWD_PATH2: {
            // [slots,3,P] doubles; the caller asks for the first k slots (k = bytes / (3 P 8))
            const size_t slab = 3 * P * sizeof(double);
            const size_t cap = (size_t)c->bcap;  // slots the slab workspace holds (>= the slots of the last call)
            src = which == ESAC_BUF_BWD_PATH1 ? c->bws.grad1 : c->bws.grad2;
            if (!src || slab == 0) return fail(-6, "esac_hip_read: buffer %d is empty (no backward call has run yet)", which);
            if (bytes == 0 || bytes % slab || bytes / slab > cap)
                return fail(-7, "esac_hip_read: buffer %d is read in whole slabs of %zu bytes, at most %zu", which, slab, cap);
            want = bytes;
            break;
        }
        default: return fail(-5, "esac_hip_read: unknown buffer id %d", which);
    }
    if (!src || want == 0) return fail(-6, "esac_hip_read: buffer %d is empty (no call has run yet)", which);
    if (bytes != want) return fail(-7, "esac_hip_read: buffer %d holds %zu bytes, caller asked for %zu", which, want, bytes);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h_dst, src, want, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int esac_hip_write_hyps(esac_hip_ctx* c, const double* h_hyps, int N) {
    if (!c || !h_hyps || N <= 0) return fail(-1, "esac_hip_write_hyps: bad argument");
    DeviceGuard guard(c->device);
    int rc = ensure_ws(c, N, c->capP > 0 ? c->capP : 1);
    if (rc) return rc;
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(c->ws.hyps, h_hyps, (size_t)N * 6 * sizeof(double), hipMemcpyHostToDevice));
    // the fp32 [R | t] rows of the streaming score are relative to each expert map's origin (device_common.hpp:
    // map_centre): they are rebuilt by the next esac_hip_score, which knows the maps
    c->rt32_stale = true;
    c->lastN = N;
    return 0;
}

extern "C" int esac_hip_set_refine_team(esac_hip_ctx* c, int members) {
    if (!c) return fail(-1, "null context");
    if (members < ESAC_REFINE_TEAM_AUTO || members > ESAC_REFINE_TEAM_MAX)
        return fail(-4, "esac_hip_set_refine_team: %d members (0..%d, or ESAC_REFINE_TEAM_AUTO)", members, ESAC_REFINE_TEAM_MAX);
    // ESAC_REFINE_TEAM_AUTO: back to the default policy (the size chosen per grid); a number: exactly that many
    c->team_auto = members == ESAC_REFINE_TEAM_AUTO && !c->team_auto_env_off;
    c->team = members == ESAC_REFINE_TEAM_AUTO ? ESAC_REFINE_TEAM_DEFAULT : members < 2 ? 0 : members;
    c->team_latched_off = false;  // an explicit request re-arms the forward teams and the training path's slot teams
    c->team_strikes = 0;
    c->slot_teams = true;
    return 0;
}

extern "C" int esac_hip_set_debug(esac_hip_ctx* c, int flags) {
    if (!c) return fail(-1, "null context");
    c->keep_errs = (flags & ESAC_DEBUG_ERROR_IMAGE) != 0;
    c->coop_stall = (flags & ESAC_DEBUG_COOP_STALL) != 0;
    c->team_spread = (flags & ESAC_DEBUG_TEAM_SPREAD) != 0;
    c->spec_off = c->spec_env_off || (flags & ESAC_DEBUG_NO_SPECULATION) != 0;
    c->spec_second_best = (flags & ESAC_DEBUG_SPEC_SECOND_BEST) != 0;
    c->spec_lose_chain = (flags & ESAC_DEBUG_SPEC_LOSE_CHAIN) != 0;
    return 0;
}

extern "C" int esac_hip_set_timing(esac_hip_ctx* c, int enabled) {
    if (!c) return fail(-1, "null context");
    c->timing = enabled != 0;
    c->timing_period = enabled > 1 ? enabled : 1;
    c->timing_calls = 0;
    c->ev_valid = false;
    if (c->ws.span_acc) {
        DeviceGuard guard(c->device);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemset(c->ws.span_acc, 0, 2 * sizeof(long long)));
    }
    return 0;
}

extern "C" int esac_hip_phase_ms(esac_hip_ctx* c, float out[6]) {
    if (!c || !out) return fail(-1, "esac_hip_phase_ms: null argument");
    if (!c->timing || !c->ev_valid) return fail(-8, "esac_hip_phase_ms: timing is off or no forward has run");
    DeviceGuard guard(c->device);
    HIP_OK(hipEventSynchronize(c->ev[4]));
    for (int i = 0; i < 4; i++) HIP_OK(hipEventElapsedTime(&out[i], c->ev[i], c->ev[i + 1]));
    HIP_OK(hipEventElapsedTime(&out[4], c->ev[0], c->ev[4]));
    HIP_OK(hipEventSynchronize(c->ev[6]));
    HIP_OK(hipEventElapsedTime(&out[5], c->ev[5], c->ev[6]));
    return 0;
}

extern "C" int esac_hip_score_span_ms(esac_hip_ctx* c, float* mean_ms, int* launches) {
    if (!c || !mean_ms) return fail(-1, "esac_hip_score_span_ms: null argument");
    if (!c->ws.span_acc) return fail(-8, "esac_hip_score_span_ms: no forward has run");
    DeviceGuard guard(c->device);
    HIP_OK(hipDeviceSynchronize());
    // mean device-side span of the score kernel since timing was enabled (100 MHz wall clock -> ms)
    long long acc[2] = {0, 0};
    HIP_OK(hipMemcpy(acc, c->ws.span_acc, sizeof(acc), hipMemcpyDeviceToHost));
    *mean_ms = acc[1] > 0 ? (float)((double)acc[0] / (double)acc[1] * 1e-5) : 0.0f;
    if (launches) *launches = (int)acc[1];
    return 0;
}
