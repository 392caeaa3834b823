// esac_refine_team.hip -- draw(argmax) + refineHyp + pose2trans of ONE frame on a grid of a few thousand cells, shared by a
// TEAM of workgroups on one XCD (round 4): 8 on the 60x80 grid, up to 32 on grids of up to 32768 cells.
//
// Reference: refineHyp esac_util.h:378-454 (the loop), cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess) reached from
// esac_util.h:426-436 (the re-fit: CvLevMarq, 6 parameters, max_iter 20, eps FLT_EPSILON), draw esac_util.h:505-530,
// pose2trans esac_util.h:537-548.  esac_refine.hip holds the same refinement for one workgroup (and for cooperating
// workgroups on grids beyond one LDS list); this file is the latency shape of the headline call: a 60x80 grid, where one
// workgroup spends 150 us on ~33 dependent passes of 2-6 us each.
//
// What is different here:
//   * CELLS LIVE IN REGISTERS.  Member g of G owns the cells [g P / G, (g+1) P / G); with G = 8 that is <= 1024 cells, at
//     most 4 per lane (600 at 60x80: 3 per lane), loaded once.  No correspondence list, no compaction, no LDS but the
//     reduction scratch: an inlier set is a bit mask per lane.
//   * ONE KIND OF PASS at a pose: the fp64 projection chain of every owned cell (lm_math.hpp: the LM chain, contraction and
//     Newton reciprocal allowed), from it (a) the squared residual over the set the running re-fit works on, (b) `err < tau`
//     for EVERY cell -- decided by the fp64 error against tau +- a band that covers what the reference's float rounding
//     can do, the bit-exact project_exact_err sequence for what is left (practically never) -- i.e. the NEXT inlier set,
//     its count and its squared residual, (c) the 24 moments of the normal equations over the running set or over the
//     next one.  27 sums per pass.
//   * FUSED ROUNDS.  The reference evaluates the error image at a pose, then starts the re-fit with a pass at the SAME
//     pose; and it ends a re-fit with a pass at the pose whose error image it evaluates next.  Whether an LM trial, if
//     accepted, ends the re-fit (iteration 20, or relative step < FLT_EPSILON) is known BEFORE the trial is evaluated
//     (it depends on the step, not on the residual), so such a trial is evaluated as a full pass with the moments over
//     the next set: accepted -- the common case -- it IS the next step's error image and first LM pass.  Rejected, the
//     re-fit goes on with a larger lambda from the normal equations of the last accepted point (they stay in registers,
//     team_step); nothing was overwritten (the sets are register masks).  ~25 rounds per frame instead of ~33, same
//     accepted points, same decisions.
//   * ONE HOP PER ROUND between the members: refine_common.hpp, tagged granules.  The chain-rule matrices of the pose are
//     computed while the exchange is in flight.
//   * COUNTED IN INSTRUCTIONS.  A member runs one wavefront per SIMD; such a wavefront issues one VALU instruction every
//     ~4.8 cycles and waits 10 for a dependent fp64 result (scripts/dev/lat_probe.hip): a round is ~1,100 instructions + the
//     hop + two LDS round trips, and it is the COUNT that binds.  Hence ONE step site, a loop that carries only the pose,
//     the cells and a few registers, the arrays ahead of the pad in ONE LDS allocation (16-bit offsets: no address
//     materialised per access), accumulators that die at their reduction, the series of the rotation at a quarter of the
//     angle (16 constants of two scalar moves each instead of 48), the normal equations in units of f.
//   * THE SERIAL SECTION DEALT TO LANES (round 5, lm_lanes.hpp).  Every lane of every wavefront used to compute all 27
//     entries of the (rvec, tvec)-space normal equations and the whole 6x6 LDL^T alike (~350 instructions a round).  Now
//     lane j of a 16-lane row holds column j: the totals of a pass are gathered per lane from LDS (6 loads instead of 27),
//     the two 3x3-block products and a Gauss-Jordan elimination run on v_fmac_f64_dpp ... row_newbcast (a DPP broadcast
//     folded into the FMA: 6 instructions for a row of outputs, 5 for an elimination step), the chain-rule matrices are
//     formed column-wise in the lanes while the exchange is in flight, and what a rejected trial needs again is 26
//     registers instead of an LDS stash.  Same-box A/B of the headline call: 0.1255-0.130 -> 0.1144-0.1155 ms.
//   * THE SELECTION IN THE PROLOGUE (a.fold_select: single frames of <= 256 hypotheses, the headline call).  softMax /
//     entropy statistics of the fp32 scores, the band of contenders and their re-score in reference arithmetic
//     (esac_util.h:235-260, 461-530 -- what k_select_rescore does in a launch of its own) run here: every member scores
//     ITS cells for every contender, one exchange adds them up, every member picks the winner from the same totals.  A
//     launch, its boundary and a 256-workgroup ramp less on a call whose selection re-scores one to three hypotheses.
// All members carry the same pose and take the same decisions from bitwise identical sums; member 0 writes the outputs,
// every member writes its cells of the inlier map once, at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lm_lanes.hpp"
#include "refine_common.hpp"

namespace esac {

constexpr int TEAM_CPL_MAX = 4;   // cells per lane a member can hold: slices of up to 1024 cells
constexpr int TEAM_NSUM = 27;     // 24 moments | residual^2 over the running set | residual^2 over the next set | size of the next set
constexpr int TEAM_LDS_PAD = 96 * 1024;  // LDS nobody touches: one member per CU (two would share its SIMDs)

template <int CPL>
struct TeamCells {
    double X[CPL], Y[CPL], Z[CPL], px[CPL], py[CPL];  // scene point and pixel position (float inputs, exact in double)
    int cell[CPL];                                     // grid cell, -1: none (ragged end of the slice)
    unsigned live;                                     // bit p: cell p exists AND its scene point is finite.  A point that is not can
                                                       // never be an inlier (its error is NaN: esac_util.h:358 then :404); it is kept
                                                       // out of every set and its coordinates read 0 here, so that the 0/1 weights of a
                                                       // pass never meet a NaN (0 * NaN would poison all 24 moments)
};

// What a lane keeps for the lane-dealt LM step (lm_lanes.hpp): loop-invariant registers
struct LaneConst {
    double hot[6];  // 1 in lane k of every 16-lane row
    double keep;    // 1 in lanes 3..6 of every row
    int off[6];     // byte offsets of this lane's X_0..2, Y_0..2 among the totals in LDS (lm_lane_slot)
};
struct LmLaneTable {
    unsigned short off[16][6];
};
constexpr LmLaneTable lm_lane_table() {
    LmLaneTable t{};
    for (int l = 0; l < 16; l++)
        for (int w = 0; w < 6; w++) t.off[l][w] = (unsigned short)(8 * lm_lane_slot(l, w));
    return t;
}
__device__ const LmLaneTable LM_LANE_TABLE = lm_lane_table();

// the decision band of `err < tau` on the fp64 squared error (see error_pass_impl in esac_refine.hip, second screen): the
// reference rounds the projection to float before taking the difference, so its error differs from the fp64 value by
// at most ~sqrt(2)/2 ulp of the pixel coordinate (+1e-6); 3.5x of head-room on that.
struct TauBand {
    double lo2, hi2;   // e2 < lo2: inlier for certain; e2 > hi2: outlier for certain
    float tau, tau_below, max_reproj;
    bool all_in;       // maxReproj < tau: every clamped error is below tau (esac_util.h:358 then :404)
};
__device__ __forceinline__ TauBand make_band(const KArgs& a) {
    const double band2 = 2.4e-7 * ((double)(a.W > a.H ? a.W : a.H) * a.sub + abs(a.shift_x) + abs(a.shift_y) + (double)a.tau) + 2e-6;
    const double tlo = (double)a.tau - band2, thi = (double)a.tau + band2;
    return TauBand{tlo > 0 ? tlo * tlo : -1.0, thi * thi, a.tau, nextafterf(a.tau, -INFINITY), a.max_reproj, a.max_reproj < a.tau};
}

// One pass at `param` (see the header).  in: run_set = the set the re-fit works on, use_next = moments over the next set
// instead.  out: next_set; the 27 totals over the team in LDS (s_tot: [0, 27) the totals, [32, 59) their negatives, what
// the lane-dealt step gathers from), tail = totals 24..26 (identical in every thread of every member); M, K = chain-rule
// matrices at param, column j in lane j of every row (the left Jacobian of SO(3) and [t]x of it, lm_lanes.hpp).
template <int CPL, int WIDE>
__device__ __forceinline__ void team_pass(const KArgs& a, const TeamCells<CPL>& cl, const double (&param)[6], const Cam& cam, const TauBand& band,
                                          unsigned run_set, bool use_next, unsigned& next_set, double (&tail)[3], const LaneConst& lc, double (&M)[3],
                                          double (&K)[3], Coop& co, double* s_part, double* s_tot, double* s_x, long long* g_cyc) {
    CYC_DECL;
    CYC_BEGIN();
    double R[9];
    LmTrig tg;
    lm_pose_rotation(param, R, tg);
    CYC_END(4);
    CYC_BEGIN();
    bool on[CPL];
#pragma unroll
    for (int p = 0; p < CPL; p++) on[p] = (cl.live >> p) & 1u;
    LmTerms<CPL> q;  // x, y, 1/z, residual of every owned cell (all zero for a missing one)
    lm_point_terms<CPL>(R, param + 3, cam, cl.X, cl.Y, cl.Z, cl.px, cl.py, on, q);
    // (b) the next inlier set -- only where the pass is (also) an error image: an LM trial that cannot end its re-fit needs
    // the residual and the moments over the running set, nothing else
    unsigned nset = 0;
    double e2[CPL];
    bool undecided[CPL];
    bool any_undecided = false;
#pragma unroll
    for (int p = 0; p < CPL; p++) {
        e2[p] = __builtin_fma(q.ex[p], q.ex[p], q.ey[p] * q.ey[p]);
        undecided[p] = false;
    }
    if (use_next) {
#pragma unroll
        for (int p = 0; p < CPL; p++) {
            const bool in2 = e2[p] < band.lo2, out2 = e2[p] > band.hi2;  // NaN: neither
            undecided[p] = on[p] && !band.all_in && !(in2 || out2);
            any_undecided |= undecided[p];
            if (on[p] && (in2 || band.all_in)) nset |= 1u << p;
        }
        // (a branch hint on the two paths a round practically never takes -- this one and the pseudo-inverse step: where the compiler
        // lays them out is worth 0.9 us of the call; the kernel is bound by the instructions it fetches and issues, profiles/r06_ab_team10.txt.
        // Hints on the cheap rare branches -- a dead exchange, the time-out -- measured 0.3 us the other way.)
        if (__builtin_expect(__any(any_undecided) != 0, 0)) {  // the reference's own sequence, op by op (rodrigues as cv::Rodrigues, contraction off)
            double Rx[9];
            rodrigues_vec2mat<false>(param, Rx, nullptr);
#pragma unroll
            for (int p = 0; p < CPL; p++) {
                if (!__any(undecided[p])) continue;
                float err = project_exact_err(Rx, param + 3, cam, (float)cl.X[p], (float)cl.Y[p], (float)cl.Z[p], (float)cl.px[p], (float)cl.py[p]);
                err = err < band.max_reproj ? err : band.max_reproj;  // std::min(l, maxReproj), esac_util.h:358
                if (undecided[p] && err < band.tau) nset |= 1u << p;
            }
        }
    }
    next_set = nset;
    // (ESAC_DEBUG_ERROR_IMAGE is served by the one-workgroup kernel -- refine_team_members: a debug option's stores and its exact
    // projections inside THIS loop cost every call 1.2 us: the code is here whether it runs or not)
    // (a), (c): 0/1 weights the optimiser cannot see through (one basic block, no chain sunk under a branch)
    double sums[TEAM_NSUM];
#pragma unroll
    for (int k = 0; k < TEAM_NSUM; k++) sums[k] = 0;
    LmTerms<CPL> m;
#pragma unroll
    for (int p = 0; p < CPL; p++) {
        double w_run = (run_set >> p) & 1u ? 1.0 : 0.0, w_next = (nset >> p) & 1u ? 1.0 : 0.0;
        asm volatile("" : "+v"(w_run), "+v"(w_next));
        const double w = use_next ? w_next : w_run;
        m.x[p] = q.x[p] * w;
        m.y[p] = q.y[p] * w;
        m.iz[p] = q.iz[p] * w;
        m.ex[p] = q.ex[p] * w;
        m.ey[p] = q.ey[p] * w;
        m.w[p] = w;
        sums[24] = __builtin_fma(w_run, e2[p], sums[24]);
        sums[25] = __builtin_fma(w_next, e2[p], sums[25]);
        sums[26] += w_next;
    }
    lm_accumulate_moments<CPL>(m, sums);
    CYC_END(5);
    CYC_BEGIN();
    // wavefront totals -> LDS -> this member's totals (threads < 27) -> granules -> every member adds all members'
    // (publishing per WAVEFRONT -- no LDS round, four granules to poll per thread -- measured 4 % slower per pass)
    wave_totals28_to_lds<TEAM_NSUM>(sums, s_part);
    team_publish<TEAM_NSUM>(workgroup_total28<REFINE_B>(s_part), co);
    lm_lane_chain<double>(tg, param + 3, lc.hot, M, K);  // while the exchange is in flight
    team_collect_lds<TEAM_NSUM, true, WIDE>(co, s_tot, s_x);
    tail[0] = s_tot[24];
    tail[1] = s_tot[25];
    tail[2] = s_tot[26];
    CYC_END(6);
    CYC_ADD(9, 1);
}

// One LM step, the only site: param = prev - solve(JtJ with diag *= 1 + lambda, JtErr), dealt to the lanes of a row
// (lm_lanes.hpp: ~190 instructions where every lane computing all 27 outputs of the transform and the whole LDL^T was ~350).
// fresh: `param` was accepted (or is the start of a re-fit) -- it becomes `prev`, and the rows c[] of the normal equations
// in (rvec, tvec) space are built from this pass's totals in LDS (per-lane gather) and the chain-rule columns M, K at
// that pose; otherwise (a rejected trial) c[] and prev are what they were: they live in registers across rounds (12 + 12:
// a lane holds one column, not the matrix -- round 4 kept U21 / g6 / prev in an LDS stash because 66 uniform registers
// would not stay).  lambda = 10^lambda_lg10 from a table in LDS (the binary exponentiation + division of pow10_int is ~600
// cycles of dependent work).  Returns whether the new trial, if accepted, ends the re-fit (iters: accepted iterations so far).
__device__ __forceinline__ bool team_step(bool fresh, const LaneConst& lc, const double (&M)[3], const double (&K)[3], int lambda_lg10, int iters,
                                          double (&param)[6], double (&prev)[6], double (&c)[6], double& dg, double inv_f, const double* s_tot,
                                          const double* s_pow10, double* s_part, long long* g_cyc) {
    CYC_DECL;
    CYC_BEGIN();
    const double lambda = s_pow10[lambda_lg10 + 16];
    if (fresh) {
#pragma unroll
        for (int k = 0; k < 6; k++) prev[k] = param[k];
        // in units of the focal length: JtJ / f^2 and JtErr / f (the damping is relative, the pivot tests are relative: the
        // step of these equations is f times the step -- six multiplications below instead of 27 of the sums)
        double X[3], Y[3];
        const char* const base = reinterpret_cast<const char*>(s_tot);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            X[k] = *reinterpret_cast<const double*>(base + lc.off[k]);
            Y[k] = *reinterpret_cast<const double*>(base + lc.off[3 + k]);
        }
        lm_lane_transform<double>(X, Y, M, K, lc.hot, lc.keep, c, dg);
    }
    CYC_PIN(c, 6);
    CYC_END(7);
    CYC_BEGIN();
    double dx[6];
    if (!fresh) CYC_ADD(17, 1);  // (profiling build: rejected trials ...
    if (__builtin_expect(!lm_lane_solve<double>(c, dg, lc.hot, lambda, dx), 0)) {
        CYC_ADD(18, 1);          // ... and pseudo-inverse steps of the frame)
        double U21[21], g6[6];
        lm_lane_to_u21<double>(c, U21, g6);
        lm_solve6_pinv(U21, g6, lambda, dx, s_part);
    }
    double dn = 0, pn = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        param[k] = prev[k] - dx[k] * inv_f;
        dn += (param[k] - prev[k]) * (param[k] - prev[k]);
        pn += prev[k] * prev[k];
    }
    CYC_PIN(param, 6);
    CYC_END(8);
    return iters + 1 >= 20 || relative_step_below_eps(dn, pn);
}

// ---- the selection, folded into the prologue (see the header).  N <= REFINE_B: one hypothesis per thread.
// Returns the winner (local index); win_score, nc and the record inputs come back through the references.  Writes what
// k_select_rescore writes: scores / exact_flag / scores_user of every hypothesis, n_contenders, stats (member 0).
constexpr int TEAM_SEL_CHUNK = 16;  // contenders re-scored per exchange (16 x 12 pose values fit one load per thread)
template <int CPL, int WIDE>
__device__ __forceinline__ int team_select(const KArgs& a, const float fs, const TeamCells<CPL>& cl, bool cells_loaded, int cell0, int cell1, const Cam& cam, Coop& co,
                                           bool writer, double* s_part, double* s_tot, double* s_x, double* s_best, int* s_besti, int* s_bestg, int* s_list,
                                           double* s_rt, double& win_score, int& nc_out, RecordInputs& rec_in) {
    constexpr int B = REFINE_B;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int P = a.H * a.W;
    // fp32 maximum (NaN-ignoring) and the band of contenders (fs: this thread's hypothesis' fp32 score, -inf beyond N -- loaded by the
    // caller with the first trip to memory of the kernel)
    float m = fs;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float* s_maxf = reinterpret_cast<float*>(s_best);
    if (lane == 0) s_maxf[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_maxf[0], s_maxf[1]), fmaxf(s_maxf[2], s_maxf[3]));
    const float band = m - a.margin;
    const bool cont = t < a.N && fs >= band;
    // softmax statistics (esac_util.h:461-497) from the fp32-path scores, in double; number of contenders
    double acc[3] = {0, 0, 0};
    if (t < a.N) {
        const double d = (double)fs - (double)m;
        const double ex = exp(d);
        acc[0] = ex;
        acc[1] = ex * d;
        acc[2] = cont ? 1.0 : 0.0;
    }
    block_sum<3, B>(acc, s_part, s_tot);
    const int nc = (int)acc[2];
    nc_out = nc;
    const double entropy = log2(acc[0]) - acc[1] / (acc[0] * 0.6931471805599453);  // -sum p log2 p, p = exp(d) / S
    if (writer && t == 0) {
        a.n_contenders[0] = nc;
        a.stats[0] = (double)m;
        a.stats[1] = acc[0];
        a.stats[2] = entropy;
    }
    if (writer && t < a.N && !cont) {
        a.scores[t] = (double)fs;
        if (a.scores_user) a.scores_user[user_slot(a, t)] = (double)fs;
        a.exact_flag[t] = 0;
    }
    // the contenders, ascending
    int* s_wcnt = s_besti;
    const unsigned long long bal = __ballot(cont);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    {
        int base = 0;
        for (int w = 0; w < wave; w++) base += s_wcnt[w];
        if (cont) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = t;
    }
    __syncthreads();
    // exact re-score, TEAM_SEL_CHUNK contenders per exchange: every member its cells, reference arithmetic op by op
    double bs = -INFINITY;
    int bi = 0x7fffffff, bg = 0x7fffffff;
    const float scale = a.alpha / a.W / a.H;  // float / int / int (esac_util.h:256)
    for (int c0 = 0; c0 < nc; c0 += TEAM_SEL_CHUNK) {
        const int cnt = nc - c0 < TEAM_SEL_CHUNK ? nc - c0 : TEAM_SEL_CHUNK;
        if (t < cnt * 12) {  // R as the sampler formed it (hyps_R), t: 12 values per contender
            const int ci = t / 12, k = t - ci * 12, h = s_list[c0 + ci];
            s_rt[t] = k < 9 ? a.hyps_R[(size_t)h * 9 + k] : a.hyps[(size_t)h * 6 + 3 + (k - 9)];
        }
        __syncthreads();
        for (int ci = 0; ci < cnt; ci++) {
            const int h = s_list[c0 + ci];
            double R[9], tv[3];
#pragma unroll
            for (int k = 0; k < 9; k++) R[k] = s_rt[ci * 12 + k];
#pragma unroll
            for (int k = 0; k < 3; k++) tv[k] = s_rt[ci * 12 + 9 + k];
            const float* __restrict__ mx = a.sc + (size_t)expert_of(a, h) * 3 * P;
            double v = 0;
#pragma unroll
            for (int p = 0; p < CPL; p++) {
                const int i = cell0 + p * B + t;
                if (i >= cell1) continue;
                float X, Y, Z, px, py;
                if (cells_loaded && ((cl.live >> p) & 1u)) {  // one expert: the map is the one in the registers (a non-finite point: as the reference reads it)
                    X = (float)cl.X[p]; Y = (float)cl.Y[p]; Z = (float)cl.Z[p]; px = (float)cl.px[p]; py = (float)cl.py[p];
                } else {
                    const int row = i / a.W, col = i - row * a.W;
                    X = mx[i]; Y = mx[P + i]; Z = mx[2 * P + i];
                    px = (float)cell_pxi(a, col); py = (float)cell_pyi(a, row);
                }
                float err = project_exact_err(R, tv, cam, X, Y, Z, px, py);
                err = err < a.max_reproj ? err : a.max_reproj;  // std::min(l, maxReproj), esac_util.h:358
                v += soft_inlier_exact(err, a.tau, a.beta);
            }
            v = wave_sum(v);
            if (lane == 0) s_part[wave * 28 + ci] = v;
        }
        __syncthreads();
        double tot[TEAM_SEL_CHUNK];
        team_publish<TEAM_SEL_CHUNK>(t < TEAM_SEL_CHUNK ? (s_part[t] + s_part[28 + t]) + (s_part[56 + t] + s_part[84 + t]) : 0.0, co);
        team_collect<TEAM_SEL_CHUNK, WIDE>(tot, co, s_tot, s_x);
        if (co.dead) break;
        if (t < cnt) {
            const int h = s_list[c0 + t];
            double sc = s_tot[t];
            sc *= scale;  // double *= float
            if (writer) {
                a.scores[h] = sc;
                if (a.scores_user) a.scores_user[user_slot(a, h)] = sc;
                a.exact_flag[h] = 1;
            }
            const int g = global_hyp(a, h);
            if (sc > bs || (sc == bs && g < bg)) {  // (one contender per thread and chunk: ascending h, so only `>` ever fires)
                bs = sc;
                bi = h;
                bg = g;
            }
        }
        __syncthreads();  // s_rt / s_part / s_tot are rewritten by the next chunk
    }
    // draw(probs, training=false): argmax of the exact scores, first (global) index on ties (esac_util.h:512-529)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double os = __shfl_xor(bs, o);
        const int oi = __shfl_xor(bi, o);
        const int og = __shfl_xor(bg, o);
        if (os > bs || (os == bs && og < bg)) {
            bs = os;
            bi = oi;
            bg = og;
        }
    }
    __syncthreads();
    if (lane == 0) {
        s_best[wave] = bs;
        s_besti[wave] = bi;
        s_bestg[wave] = bg;
    }
    __syncthreads();
    bs = s_best[0];
    bi = s_besti[0];
    bg = s_bestg[0];
#pragma unroll
    for (int w = 1; w < B / 64; w++) {
        const double os = s_best[w];
        const int oi = s_besti[w];
        const int og = s_bestg[w];
        if (os > bs || (os == bs && og < bg)) {
            bs = os;
            bi = oi;
            bg = og;
        }
    }
    const int win = bi == 0x7fffffff ? 0 : bi;
    // the winner's score: a contender's exact one; with no contender at all (every score NaN) what the selection kernel
    // leaves in scores[0]
    win_score = bi == 0x7fffffff ? (double)a.fast_scores[0] : bs;
    rec_in = RecordInputs{exp(win_score - (double)m) / acc[0], entropy, a.status[0]};
    return win;
}

// ---- ESAC_FLAG_EXACT_SCORES, folded (a.fold_select == 2; N <= REFINE_B: one hypothesis per thread): k_rescore has scored every
// hypothesis in reference arithmetic -- softMax / entropy over those scores (esac_util.h:461-497) and draw's argmax
// (esac_util.h:512-529, first global index on ties), what k_stats_exact + refine_pick_winner do in a launch of their own.
// Every member computes the same from the same scores; no exchange.
__device__ __forceinline__ int team_select_exact(const KArgs& a, const double sc, bool writer, double* s_part, double* s_tot, double* s_best, int* s_besti,
                                                 int* s_bestg, double& win_score, int& nc_out, RecordInputs& rec_in) {
    constexpr int B = REFINE_B;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double m = sc;  // (sc: this thread's hypothesis' exact score, -inf beyond N)  fmax ignores NaN, as k_stats_exact
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if (lane == 0) s_best[wave] = m;
    __syncthreads();
    m = fmax(fmax(s_best[0], s_best[1]), fmax(s_best[2], s_best[3]));
    double acc[2] = {0, 0};
    if (t < a.N) {
        const double d = sc - m;
        const double ex = exp(d);
        acc[0] = ex;
        acc[1] = ex * d;
    }
    block_sum<2, B>(acc, s_part, s_tot);
    const double entropy = log2(acc[0]) - acc[1] / (acc[0] * 0.6931471805599453);  // -sum p log2 p, p = exp(d) / S
    nc_out = a.N;
    if (writer && t == 0) {
        a.n_contenders[0] = a.N;
        a.stats[0] = m;
        a.stats[1] = acc[0];
        a.stats[2] = entropy;
    }
    double bs = t < a.N ? sc : -INFINITY;
    int bi = t < a.N && sc > -INFINITY ? t : 0x7fffffff, bg = bi == 0x7fffffff ? 0x7fffffff : global_hyp(a, t);
    if (!(bs > -INFINITY)) {  // NaN or -inf: never the maximum (refine_pick_winner: `s > bs` is false for it)
        bs = -INFINITY;
        bi = bg = 0x7fffffff;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double os = __shfl_xor(bs, o);
        const int oi = __shfl_xor(bi, o);
        const int og = __shfl_xor(bg, o);
        if (os > bs || (os == bs && og < bg)) {
            bs = os;
            bi = oi;
            bg = og;
        }
    }
    __syncthreads();
    if (lane == 0) {
        s_best[wave] = bs;
        s_besti[wave] = bi;
        s_bestg[wave] = bg;
    }
    __syncthreads();
    bs = s_best[0];
    bi = s_besti[0];
    bg = s_bestg[0];
#pragma unroll
    for (int w = 1; w < B / 64; w++) {
        const double os = s_best[w];
        const int oi = s_besti[w];
        const int og = s_bestg[w];
        if (os > bs || (os == bs && og < bg)) {
            bs = os;
            bi = oi;
            bg = og;
        }
    }
    const int win = bi == 0x7fffffff ? 0 : bi;
    // the winner's score: the maximum found (no second trip to memory); with no score above -inf at all what hypothesis 0 holds
    win_score = bi == 0x7fffffff ? a.scores[0] : bs;
    rec_in = RecordInputs{exp(win_score - m) / acc[0], entropy, a.status[0]};
    return win;
}

// TEAM_SLOTS (training path, esac.cpp:328-347): one team of 8 per selection slot.  Blocks come in groups of 64 = 8 teams, one
// per XCD: block L is member (L % 64) / 8 of the team on XCD L % 8 of group L / 64, which refines slot (L / 64) * 8 + L % 8.
// Every block of such a launch is a member; teams of slots beyond n_sel leave at once.  No winner pick, no selection, no
// record: the refined pose goes to bwd.ref_hyps, the trace to bwd.map_info, the last accepted inlier set to buffer 0 of
// the slot's maps.  Teams become resident in dispatch order, so a launch with more teams than the chip holds at once
// (32) drains group by group; a team never waits for a later one.
// TEAM_FRAMES (esac_hip_forward_batch, round 5): the same block -> team mapping with FRAME (L / 64) * 8 + L % 8 of a batch of
// at most ESAC_TEAM_BATCH_MAX frames in place of the slot -- every frame's winner gets what the single call's gets (8 CUs of
// one XCD, the selection folded in for <= 256 hypotheses), each team with its own granules.
enum : int { TEAM_SINGLE = 0, TEAM_SLOTS = 1, TEAM_FRAMES = 2 };
template <int CPL, int MODE = TEAM_SINGLE, int WIDE = 8>
__global__ __launch_bounds__(REFINE_B) void k_refine_team(KArgs a) {
    static_assert(MODE == TEAM_SINGLE || WIDE == 8, "slot and batch teams have eight members");
    constexpr int B = REFINE_B;
    constexpr bool SLOTS = MODE == TEAM_SLOTS;
    if (MODE == TEAM_SINGLE && (blockIdx.x % a.team_stride) != 0) return;  // the other seven of every eight exist for placement: block b runs on XCD b % 8
    const int slot = MODE != TEAM_SINGLE ? (int)(blockIdx.x >> 6) * 8 + (int)(blockIdx.x & 7) : 0;  // slot / frame of this block's team
    if (SLOTS && slot >= a.bwd.n_sel[0]) return;
    if (SLOTS && a.bwd.n_sel[0] > a.bwd.team_max_slots) return;  // (many slots: one workgroup each is the better shape, launch_refine_slots does it)
    if (MODE == TEAM_FRAMES) {
        if (slot >= a.frames) return;
        frame_view(a, slot);
        if (slot != 0) a.refine_info = nullptr;  // the info words describe frame 0's refinement
    }
    // ONE allocation, the arrays first: their addresses then fit the 16-bit offset field of the LDS instructions (one base
    // register for all of them).  As separate variables they were laid out behind the 96 KB pad, and every access
    // materialised its own address first -- an extra instruction per LDS access on a section every lane walks.
    struct Lds {
        double part[28 * (B / 64) > 84 ? 28 * (B / 64) : 84];  // block reductions; scratch of the pseudo-inverse step
        double tot[LM_LANE_SLOTS];                   // an exchange's totals [0, 27) | zeros | their negatives [32, 59) | zeros (lm_lanes.hpp)
        double pow10[34];                            // 10^-16 .. 10^16: the LM damping factors
        double x[TEAM_MAX * 32];                     // exchanges of more than 8 members: the polled values, [member][value]
        double rt[TEAM_SEL_CHUNK * 12];              // folded selection: the poses of a chunk of contenders
        double best[B / 64];
        int besti[B / 64];
        int bestg[B / 64];
        int list[B];                                 // folded selection: the contenders, ascending
        int coop_dead;
        char pad[TEAM_LDS_PAD];                      // LDS nobody touches: one member per CU
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    double* const s_part = lds.part;
    double* const s_tot = lds.tot;
    double* const s_x = lds.x;
    double* const s_best = lds.best;
    int* const s_besti = lds.besti;
    int* const s_bestg = lds.bestg;
    int* const s_list = lds.list;
    double* const s_rt = lds.rt;
    double* const s_pow10 = lds.pow10;
    if (a.team_stride < 0) lds.pad[threadIdx.x] = 1;  // (never: keeps the allocation)
    if (MODE == TEAM_SINGLE && spec_gate_closed(a, &lds.coop_dead)) return;  // (a gated launch: only when the join found the speculation failed)
    if (MODE == TEAM_SINGLE) spec_open_chain(a);
    const int P = a.H * a.W;
    const Cam cam = make_cam(a);
    long long g_cyc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)g_cyc;
    CYC_DECL;
#ifdef ESAC_PROFILE_CYCLES
    const long long cyc_start = clock64();
#endif
    CYC_BEGIN();
    Coop co{1, 0, nullptr, nullptr, nullptr, 0ull, 1, 0L, &lds.coop_dead, false, nullptr, 0ull, false};
    if (SLOTS) {
        coop_init(co, a, 8, (int)(blockIdx.x & 63) >> 3, ESAC_TEAM_SPIN_LIMIT_SLOTS);
        co.gran = reinterpret_cast<u32x4*>(a.bwd.team_gran) + (size_t)slot * ESAC_TEAM_GRANULES;
        co.tag = a.bwd.team_tag;
    } else if (MODE == TEAM_FRAMES) {
        coop_init(co, a, 8, (int)(blockIdx.x & 63) >> 3, ESAC_TEAM_SPIN_LIMIT);
        co.gran += (size_t)slot * ESAC_TEAM_GRANULES;
    } else {
        coop_init(co, a, (int)gridDim.x / a.team_stride, (int)blockIdx.x / a.team_stride, ESAC_TEAM_SPIN_LIMIT);
    }
    co.expect = co.G;
    if (a.coop_extra && co.g == co.G - 1) return;  // ESAC_DEBUG_COOP_STALL: the last member never shows up
    // first exchange, in flight while the winner is looked up: a census of the XCDs the members run on (64^XCC_ID each)
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    team_publish<1>((double)(1ull << (6 * (xcc & 7))), co);
    // ... and once it is in: do all members share this one's XCD?  (the census is the same number in every member, so they all
    // answer alike: G in one 6-bit field.)  Then the exchanges that follow stay in that XCD's L2 (gran_store).
    auto note_census = [&](double cz) {
        co.local = !co.dead && (unsigned long long)cz == ((unsigned long long)co.G << (6 * (xcc & 7)));
    };
    if (threadIdx.x < 33) s_pow10[threadIdx.x] = pow10_int((int)threadIdx.x - 16);  // (visible after the barrier of the winner pick)
    if (threadIdx.x < LM_LANE_SLOTS) s_tot[threadIdx.x] = 0.0;  // the zero slots stay zero: an exchange writes [0, 27) and [32, 59)
    // the lane-dealt LM step's constants of this lane (opaque to the optimiser: values to keep, not to rematerialise per round)
    LaneConst lc;
    {
        const int l16 = (int)threadIdx.x & 15;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            lc.hot[k] = l16 == k ? 1.0 : 0.0;
            lc.off[k] = (int)LM_LANE_TABLE.off[l16][k];
            asm volatile("" : "+v"(lc.hot[k]), "+v"(lc.off[k]));
        }
        lc.keep = (l16 >= 3 && l16 <= 6) ? 1.0 : 0.0;
        asm volatile("" : "+v"(lc.keep));
    }
    const bool writer = co.g == 0;
    if (writer && !SLOTS)
        for (int i = threadIdx.x; i <= ESAC_MAX_REF_STEPS_K; i += B) a.inlier_counts[i] = -1;  // (drained by the barrier of the winner pick)
    const int cell0 = (int)((long long)P * co.g / co.G), cell1 = (int)((long long)P * (co.g + 1) / co.G);

    // this lane's cells: with one expert the map is known before the winner is (one dependent load less on the way in)
    TeamCells<CPL> cl;
    auto load_cells = [&](const float* __restrict__ mx) {
        cl.live = 0;
#pragma unroll
        for (int p = 0; p < CPL; p++) {
            const int i = cell0 + p * B + (int)threadIdx.x;
            const bool have = i < cell1;
            const int ic = have ? i : cell0;
            cl.cell[p] = have ? i : -1;
            const int row = ic / a.W, col = ic - row * a.W;
            const float fx = mx[ic], fy = mx[P + ic], fz = mx[2 * P + ic];
            const bool fin = have && fabsf(fx) <= 3.0e38f && fabsf(fy) <= 3.0e38f && fabsf(fz) <= 3.0e38f;  // false for NaN / inf
            if (fin) cl.live |= 1u << p;
            cl.X[p] = fin ? (double)fx : 0.0;
            cl.Y[p] = fin ? (double)fy : 0.0;
            cl.Z[p] = fin ? (double)fz : 0.0;
            cl.px[p] = (double)(float)cell_pxi(a, col);  // Point2f of the integer pixel centre (esac_util.h:64-66, :180)
            cl.py[p] = (double)(float)cell_pyi(a, row);
        }
    };
    if (a.E == 1) load_cells(a.sc);
    // the folded selection has one hypothesis per thread: its pose is fetched WITH its score, and the winner's thread hands it round
    // through LDS -- a barrier instead of a second, dependent trip to memory between the argmax and the first pass
    // (and so is its score: a workgroup barrier waits for the loads in flight, so whatever the selection loaded behind its first
    // barrier was a trip of its own)
    double my_hyp[6] = {0, 0, 0, 0, 0, 0};
    double my_score = -INFINITY;
    float my_fast = -INFINITY;
    if (!SLOTS && a.fold_select && (int)threadIdx.x < a.N) {
#pragma unroll
        for (int k = 0; k < 6; k++) my_hyp[k] = a.hyps[(size_t)threadIdx.x * 6 + k];
        if (a.fold_select == 2) my_score = a.scores[threadIdx.x];
        else                    my_fast = a.fast_scores[threadIdx.x];
    }
    int nc = 0, win;
    double win_score = 0;
    RecordInputs rec_in{0.0, 0.0, 0ull};
    double census[1] = {0.0};
    if (SLOTS) {
        win = a.bwd.sel[slot];
        __syncthreads();  // (s_pow10, s_coop_dead)
    } else if (a.fold_select == 2) {
        __syncthreads();  // (s_pow10, s_coop_dead)
        win = team_select_exact(a, my_score, writer, s_part, s_tot, s_best, s_besti, s_bestg, win_score, nc, rec_in);
    } else if (a.fold_select) {
        __syncthreads();  // (s_pow10, s_coop_dead)
        team_collect<1, WIDE>(census, co, s_tot, s_x);
        note_census(census[0]);
        win = team_select<CPL, WIDE>(a, my_fast, cl, a.E == 1, cell0, cell1, cam, co, writer, s_part, s_tot, s_x, s_best, s_besti, s_bestg, s_list, s_rt,
                               win_score, nc, rec_in);
    } else {
        if (MODE == TEAM_SINGLE && a.spec_mode == 2) {
            // speculative, the selection running beside this kernel: the fp32 argmax of the settled hypotheses; score, probability,
            // entropy and contender count of the record are k_spec_join's to fill in
            win = spec_pick_fast<B>(a, s_best, s_besti, s_bestg);
            if (spec_nothing_to_refine(a, win, writer)) return;
            if (writer && threadIdx.x < 64) rec_in = RecordInputs{0.0, 0.0, a.status[0]};
        } else {
            nc = a.n_contenders[0];
            win = refine_pick_winner<B>(a, s_best, s_besti, s_bestg);
            win_score = a.scores[win];
            if (writer && threadIdx.x < 64) rec_in = refine_record_inputs(a, win_score);
        }
    }
    const int e = expert_of(a, win);
    double pose[6];
    if (!SLOTS && a.fold_select) {
        if ((int)threadIdx.x == win) {
#pragma unroll
            for (int k = 0; k < 6; k++) s_rt[k] = my_hyp[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; k++) pose[k] = s_rt[k];
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) pose[k] = a.hyps[(size_t)win * 6 + k];
    }
    if (a.E != 1) load_cells(a.sc + (size_t)e * 3 * P);
    if (SLOTS || a.fold_select != 1) {
        team_collect<1, WIDE>(census, co, s_tot, s_x);
        note_census(census[0]);
    }
    CYC_END(1);

    // ---- refineHyp (esac_util.h:378-454) around ONE pass site
    const TauBand band = make_band(a);
    double sums[3];        // totals 24..26 of a pass: residual^2 over the running set / over the next set, size of the next set
    double M[3], K[3];     // chain-rule columns at the pose of the pass (lane-dealt)
    double c[6], dg = 0.0, prev[6];  // the normal equations' rows and diagonal (lane-dealt) and the point they were built at: what a
                                     // rejected trial needs again
#pragma unroll
    for (int k = 0; k < 6; k++) c[k] = prev[k] = 0.0;
    unsigned run_set = 0, next_set = 0, acc_set = 0;  // sets as bit masks over this lane's cells: the running re-fit's, the one the last
                                                      // full pass found, the last ACCEPTED step's (inlierMap, esac_util.h:440)
    // The loop carries the pose (`param`), this lane's cells, a handful of scalars and what a REJECTED trial needs again: this
    // lane's column of the normal equations and the point they were built at (c, dg, prev above: 13 registers since the LM step
    // is dealt to the lanes -- round 4 kept the whole matrix, 66 uniform values, in an LDS stash per wavefront).  ONE site each
    // for the pass, the normal equations and the solve.
    double param[6];
#pragma unroll
    for (int k = 0; k < 6; k++) param[k] = pose[k];
    int accepted = 0, last_inliers = 0, lm_total = 0, rstep = 0;
    unsigned best_inliers = 4;
    // LM state: lambda = 10^lambda_lg10, iterations of the running re-fit, |err|^2 at `prev`
    int lambda_lg10 = -3, iters = 0;
    double prev_err2 = 0;
    bool in_refit = false;   // false: the next pass is the error image at `param` with the first LM pass of its set
    // Does the trial at `param`, if accepted, end the re-fit?  (CvLevMarq: ++iters >= max_iter, or the relative step
    // cvNorm(param, prevParam, CV_RELATIVE_L2) < eps)  -- a function of the step alone, known when the step is taken.
    bool ends_refit = true;
    const double inv_f = 1.0 / cam.fx;
    for (;;) {
        team_pass<CPL, WIDE>(a, cl, param, cam, band, run_set, ends_refit, next_set, sums, lc, M, K, co, s_part, s_tot, s_x, g_cyc);
        if (co.dead) break;  // an exchange timed out: the sums are garbage, the call reports it
        CYC_BEGIN();
        bool fresh = true;  // normal equations from this pass (else: state CHECK_ERR failed, retry from `prev` with a larger lambda)
        if (in_refit && trial_rejected(sums[0], prev_err2, ends_refit) && ++lambda_lg10 <= 16) {
            fresh = false;
        } else {
            if (in_refit) {
                lambda_lg10 = lambda_lg10 - 1 > -16 ? lambda_lg10 - 1 : -16;
                ++iters;
                if (!ends_refit) {  // state CALC_J at the accepted point
                    prev_err2 = sums[0];
                } else {  // the re-fit is done: its last trial is the refined pose, and this pass was its error image
                    lm_total += iters;
                    accepted++;
                    last_inliers = (int)best_inliers;
                    acc_set = run_set;
                    in_refit = false;
                    rstep++;
                }
            }
            if (!in_refit) {
                // error image at `param` (reproErrs, esac.cpp:169 / esac_util.h:445-452): next_set, its size, its normal equations
                if (rstep >= a.max_ref_steps) break;  // the reference also evaluates the errors of its last re-fit
                const int n_inl = (int)sums[2];
                if (!SLOTS && writer && threadIdx.x == 0) a.inlier_counts[rstep] = n_inl;
                if ((unsigned)n_inl <= best_inliers) break;  // converged (esac_util.h:417-419)
                best_inliers = (unsigned)n_inl;
                // the re-fit over next_set starts from `param`: this pass already was its first one (iters == 0: prevErrNorm = |err(initial pose)|)
                run_set = next_set;
                in_refit = true;
                lambda_lg10 = -3;
                iters = 0;
                prev_err2 = sums[1];
            }
        }
        CYC_END(16);
        ends_refit = team_step(fresh, lc, M, K, lambda_lg10, iters, param, prev, c, dg, inv_f, s_tot, s_pow10, s_part, g_cyc);
    }
    // the refined pose: the last accepted re-fit's (esac_util.h:439); a pass that ended the loop was evaluated AT it
    // (or at the initial pose when no re-fit was accepted; in_refit here only after a time-out: the point the trial left)
#pragma unroll
    for (int k = 0; k < 6; k++) pose[k] = in_refit ? prev[k] : param[k];

    // inlierMap of the last accepted step (esac_util.h:440), every member its cells; buffer 0 (result[31] names it)
    if (accepted > 0) {
        uint8_t* const map0 = SLOTS ? a.bwd.maps + (size_t)slot * 2 * P : a.inlier_map;
#pragma unroll
        for (int p = 0; p < CPL; p++)
            if (cl.cell[p] >= 0) map0[cl.cell[p]] = (uint8_t)((acc_set >> p) & 1u);
    }
    if (SLOTS) {
        if (writer && threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) a.bwd.ref_hyps[(size_t)win * 6 + k] = pose[k];
            int* mi = a.bwd.map_info + 4 * slot;
            mi[0] = accepted > 0 ? 0 : -1;
            mi[1] = last_inliers;
            mi[2] = accepted;
            mi[3] = lm_total;
        }
        return;
    }
    CYC_BEGIN();
    if (threadIdx.x < 64 && writer) {  // (s_part: the last exchange's barrier lies behind every reader of it)
        refine_write_record(a, rec_in, pose, win, win_score, e, nc, accepted, last_inliers, lm_total, accepted > 0 ? 0 : -1, REFINE_TEAM, co,
                            (unsigned long long)census[0], s_part);
#ifdef ESAC_PROFILE_CYCLES
        if (threadIdx.x == 0) {
            CYC_END(3);
            g_cyc[0] = clock64() - cyc_start;
            for (int k = 0; k < 24; k++) a.cycles[k] = g_cyc[k];
        }
#endif
    }
}

// Members of the team that refines a single frame (0: one workgroup refines): grids of 1024 .. 32768 cells; as many members
// as give every lane ONE cell (the pass is then as short as it gets: 19 at 60x80), at most what was asked for
// (esac_hip_set_refine_team) and the CUs of one XCD, at least so many that a member's slice fits its lanes' registers
// (four cells per lane); the device must hold the whole launch at once.
int refine_team_members(const KArgs& a) {
    const int P = a.H * a.W;
    if (a.team < 2 || P > TEAM_MAX * TEAM_CPL_MAX * REFINE_B || P < ESAC_REFINE_TEAM_MIN_CELLS || !a.coop_partials) return 0;
    if (a.errs) return 0;  // ESAC_DEBUG_ERROR_IMAGE: the error image is written by the one-workgroup kernel (k_refine)
    if (a.frames != 1) {  // a small batch: a team of 8 per frame (k_refine_team<CPL, TEAM_FRAMES>), all teams resident together
        return a.frames <= ESAC_TEAM_BATCH_MAX && P <= 8 * TEAM_CPL_MAX * REFINE_B && a.coop_max >= 8 * ESAC_TEAM_BATCH_MAX ? 8 : 0;
    }
    int G = (P + REFINE_B - 1) / REFINE_B;
    if (G > a.team) G = a.team;
    if (G > TEAM_MAX) G = TEAM_MAX;
    const int need = (P + TEAM_CPL_MAX * REFINE_B - 1) / (TEAM_CPL_MAX * REFINE_B);
    if (G < need) G = need;
    if (G < 2) G = 2;
    // the default size (round 6): a pass costs a lane its cells, and every lane of a member walks as many as its fullest lane
    // holds -- eight members of 600 cells hold THREE per lane on the 60x80 grid, ten of 480 hold two.  Up to 16 members exchange
    // without LDS staging (refine_common.hpp: team_collect_lds), so: the smallest team <= 16 that lowers the cells per lane.
    if (a.team_auto && a.team == ESAC_REFINE_TEAM_DEFAULT_K && G <= 16) {
        const int cpl = ((P + G - 1) / G + REFINE_B - 1) / REFINE_B;
        if (cpl >= 2) {
            const int G2 = (P + REFINE_B * (cpl - 1) - 1) / (REFINE_B * (cpl - 1));
            if (G2 > G && G2 <= 16) G = G2;
        }
    }
    if (a.coop_max < G * (a.team_stride > 0 ? a.team_stride : 8)) return 0;
    return G;
}

// Training path: a team of 8 per selection slot (k_refine_team<CPL, true>) -- grids whose eighth fits a member's registers
bool refine_slots_can_team(const KArgs& a) {
    const int P = a.H * a.W;
    return a.team >= 2 && a.frames == 1 && P >= ESAC_REFINE_TEAM_MIN_CELLS && P <= 8 * TEAM_CPL_MAX * REFINE_B && a.bwd.team_gran != nullptr;
}
unsigned long long launch_refine_slots_team(KArgs& a, hipStream_t s) {
    a.bwd.team_tag = next_refine_tag();
    a.bwd.team = 8;
    const int P = a.H * a.W;
    const int cpl = ((P + 7) / 8 + REFINE_B - 1) / REFINE_B;
    int cap = a.N < a.bwd.cap ? a.N : a.bwd.cap;
    if (cap > a.bwd.team_max_slots) cap = a.bwd.team_max_slots;  // (a selection of more slots is refined by launch_refine_slots)
    const dim3 grid((unsigned)((cap + 7) / 8) * 64), block(REFINE_B);
    switch (cpl) {
        case 1: hipLaunchKernelGGL((k_refine_team<1, TEAM_SLOTS>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_refine_team<2, TEAM_SLOTS>), grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_refine_team<3, TEAM_SLOTS>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_refine_team<4, TEAM_SLOTS>), grid, block, 0, s, a); break;
    }
    return a.bwd.team_tag;
}

// The selection can run in the team kernel's prologue: a team refines this call, one hypothesis per thread of a member, the
// default score route (ESAC_FLAG_EXACT_SCORES has statistics of its own), no device-side span stamps to reduce.
bool refine_folds_select(const KArgs& a) {
    return refine_team_members(a) > 0 && a.N <= REFINE_B && !(a.flags & ESAC_FLAG_EXACT_SCORES_K) && !a.tstamps;
}
// ESAC_FLAG_EXACT_SCORES: the scores are final when the refinement starts; their softmax statistics and argmax are a few
// reductions over <= 256 values -- in the team kernel's prologue instead of a launch of their own (k_stats_exact)
bool refine_folds_exact_stats(const KArgs& a) {
    return refine_team_members(a) > 0 && a.N <= REFINE_B && (a.flags & ESAC_FLAG_EXACT_SCORES_K) != 0;
}

unsigned long long launch_refine_team(const KArgs& a, hipStream_t s) {
    const int G = refine_team_members(a);
    KArgs b = a;
    if (b.team_stride <= 0) b.team_stride = 8;
    b.coop_tag = next_refine_tag();
    const int P = a.H * a.W;
    const int slice = (P + G - 1) / G;  // the largest member slice: floor((g+1) P / G) - floor(g P / G) <= ceil(P / G)
    const int cpl = (slice + REFINE_B - 1) / REFINE_B;
    if (a.frames != 1) {  // a team of 8 per frame, eight teams per 64 blocks (one per XCD)
        const dim3 grid((unsigned)((a.frames + 7) / 8) * 64), block(REFINE_B);
        switch (cpl) {
            case 1: hipLaunchKernelGGL((k_refine_team<1, TEAM_FRAMES>), grid, block, 0, s, b); break;
            case 2: hipLaunchKernelGGL((k_refine_team<2, TEAM_FRAMES>), grid, block, 0, s, b); break;
            case 3: hipLaunchKernelGGL((k_refine_team<3, TEAM_FRAMES>), grid, block, 0, s, b); break;
            default: hipLaunchKernelGGL((k_refine_team<4, TEAM_FRAMES>), grid, block, 0, s, b); break;
        }
        return b.coop_tag;
    }
    const dim3 grid(G * b.team_stride), block(REFINE_B);
    // one instantiation per (cells per lane, members it can collect): see team_collect_lds
#define ESAC_LAUNCH_TEAM(CPL_)                                                                                 \
    do {                                                                                                       \
        if (G <= 8)       hipLaunchKernelGGL((k_refine_team<CPL_, TEAM_SINGLE, 8>), grid, block, 0, s, b);     \
        else if (G <= 16) hipLaunchKernelGGL((k_refine_team<CPL_, TEAM_SINGLE, 16>), grid, block, 0, s, b);    \
        else              hipLaunchKernelGGL((k_refine_team<CPL_, TEAM_SINGLE, 32>), grid, block, 0, s, b);    \
    } while (0)
    switch (cpl) {
        case 1: ESAC_LAUNCH_TEAM(1); break;
        case 2: ESAC_LAUNCH_TEAM(2); break;
        case 3: ESAC_LAUNCH_TEAM(3); break;
        default: ESAC_LAUNCH_TEAM(4); break;
    }
#undef ESAC_LAUNCH_TEAM
    return b.coop_tag;
}

}  // namespace esac
