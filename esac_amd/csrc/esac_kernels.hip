// esac_kernels.hip -- the ESAC hypothesis/inlier hot path as HIP kernels for gfx950 (MI355X).
//
// Reference path: code/esac/esac.cpp:64-190 (esac_forward).  Phases -> kernels:
//   K1 k_sample        sampleHypotheses + safeSolvePnP(P3P)        esac_util.h:129-225
//   K2 k_score_fast    getReproErrs + getHypScores, fused, fp32    esac.cpp:131-147, esac_util.h:235-363
//   K3 k_select_rescore  softMax / entropy / argmax band + reference-arithmetic (fp64) re-score of the
//                        contenders in the same launch                esac_util.h:461-530
//   K3b k_rescore      reference-arithmetic score of every hypothesis (training path)
//   K4 k_refine        draw(argmax) + refineHyp + pose2trans       esac_util.h:378-454,505-548
//
// Mapping to CDNA4: K1 = one hypothesis per workgroup, a lane evaluates one sampling try with the whole P3P in fp64
// registers (single frames: the four candidates of a try on two lanes; thousands of hypotheses: four hypotheses per
// wavefront for their first 16 tries), `ballot` picks the lowest accepted try (= the try a sequential loop would stop
// at).  K2 = one hypothesis per workgroup, the H x W map streamed with 16-byte coalesced loads (x/y/z planes), ~30 fp32
// VALU ops per cell, DPP wavefront reductions; no MFMA: there is no dense contraction anywhere on this path.
// K4 (esac_refine.hip) = one workgroup, LM normal equations reduced with DPP + v_permlane swaps, pose state kept
// redundantly in every lane so no broadcast is needed.
//
// Precision split: the streaming score (K2) runs in fp32 and only ranks; every hypothesis within `margin` of the fp32
// maximum is re-scored by K3 with the reference's exact mixed float/double arithmetic, and the winner is the argmax of
// those exact scores (first index on ties, esac_util.h:519).  All discrete decisions of refinement (inlier tests,
// stopping rule) use the exact arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pose_math.hpp"
#include "p3p_screen.hpp"
#include "rng.hpp"
#include "esac_kernels.hpp"
#include "device_common.hpp"

namespace esac {

// ================================================================= K1: sample + P3P
// Stored pose and acceptance of a solved sample: the reference keeps (rvec, tvec), re-expands it when projecting
// (esac_util.h:202) and accepts iff the 4 sampled points reproject within tau (esac_util.h:210-221, norm in double).
__device__ __forceinline__ bool accept_sample(const double Rp[9], const double Tp[3], const float (&Pf)[4][3], const double (&mu)[4],
                                              const double (&mv)[4], const Cam& cam, double tau, double rvec[3], double T[3],
                                              double R[9]) {
    rodrigues_mat2vec(Rp, rvec);
    rodrigues_vec2mat<false>(rvec, R, nullptr);
    T[0] = Tp[0]; T[1] = Tp[1]; T[2] = Tp[2];
    bool accepted = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double Xd = Pf[j][0], Yd = Pf[j][1], Zd = Pf[j][2];
        double x = R[0] * Xd + R[1] * Yd + R[2] * Zd + T[0];
        double y = R[3] * Xd + R[4] * Yd + R[5] * Zd + T[1];
        double z = R[6] * Xd + R[7] * Yd + R[8] * Zd + T[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        const float u = (float)(x * cam.fx + cam.cx), v = (float)(y * cam.fy + cam.cy);
        const float dx = (float)mu[j] - u, dy = (float)mv[j] - v;
        const double nrm = sqrt((double)dx * dx + (double)dy * dy);
        if (nrm < tau) continue;
        accepted = false;
    }
    return accepted;
}

// A solved sample whose 4th point already misses tau by a clear margin under the chosen candidate cannot pass the
// acceptance test: the stored pose differs from the candidate only by the rvec round trip (~1e-12 px) and the
// reference's float rounding of the projection (< 1e-4 px).  Such a try needs neither the matrix->vector->matrix
// conversion nor the four reprojections -- unless it is the LAST try of the budget, whose state must remain.
__device__ __forceinline__ bool cannot_pass(double reproj2, double tau) {
    const double lim = tau + 0.01;
    return reproj2 > lim * lim;  // NaN: false -> the full test decides
}

// the 4 cells of try t of hypothesis gh, their scene points and pixel positions.  With the planar [E,3,H,W] layout a
// cell is three 4-byte reads from three cache lines; when the maps are too large for the caches (KArgs::sc4, see
// k_pack_cells) the sampler reads one 16-byte (x, y, z, -) record per cell instead.
__device__ __forceinline__ void gather_sample(const KArgs& a, const float* __restrict__ map, int P, const Philox& rng, uint32_t gh,
                                              uint32_t t, int (&cx)[4], int (&cy)[4], V3 (&Pt)[4], float (&Pf)[4][3],
                                              double (&mu)[4], double (&mv)[4]) {
    draw_cells(rng, gh, t, a.W, a.H, cx, cy);
    const float4* __restrict__ map4 = a.sc4 ? a.sc4 + (size_t)(map - a.sc) / 3 : nullptr;  // (map - sc) / 3 = expert * P
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int idx = cy[j] * a.W + cx[j];
        if (map4) {
            const float4 v = map4[idx];
            Pf[j][0] = v.x; Pf[j][1] = v.y; Pf[j][2] = v.z;
        } else {
            Pf[j][0] = map[idx];
            Pf[j][1] = map[P + idx];
            Pf[j][2] = map[2 * P + idx];
        }
        Pt[j] = V3{(double)Pf[j][0], (double)Pf[j][1], (double)Pf[j][2]};
        mu[j] = (double)cell_px(a, cx[j]);
        mv[j] = (double)cell_py(a, cy[j]);
    }
}

// planar [E,3,H,W] -> [E,H*W] records (x, y, z, 0): one coalesced pass (12 B read, 16 B written per cell), paid once per
// call when the sampler's random 4-byte gathers would otherwise pull three cache lines per cell from HBM / Infinity Cache
__global__ __launch_bounds__(256) void k_pack_cells(KArgs a) {
    const size_t P = (size_t)a.H * a.W, total = P * a.E;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t e = i / P, c = i - e * P;
        const float* m = a.sc + e * 3 * P;
        const_cast<float4*>(a.sc4)[i] = make_float4(m[c], m[P + c], m[2 * P + c], 0.0f);
    }
}

// float [R | t + R c] for the fp32 scoring stream, c = origin of the expert's map (device_common.hpp:map_centre);
// the shifted translation is formed in double
__device__ __forceinline__ void store_rt32(const KArgs& a, int h, const float* __restrict__ map, const double R[9], const double T[3]) {
    const Centre c = map_centre(a, map);
    const double cx = c.x, cy = c.y, cz = c.z;
    float* rt = a.rt32 + (size_t)h * 12;
#pragma unroll
    for (int k = 0; k < 9; k++) rt[k] = (float)R[k];
    rt[9] = (float)(R[0] * cx + R[1] * cy + R[2] * cz + T[0]);
    rt[10] = (float)(R[3] * cx + R[4] * cy + R[5] * cz + T[1]);
    rt[11] = (float)(R[6] * cx + R[7] * cy + R[8] * cz + T[2]);
}

__device__ __forceinline__ void store_hypothesis(const KArgs& a, int h, const float* __restrict__ map, const double rvec[3],
                                                 const double T[3], const double R[9], const int (&cx)[4], const int (&cy)[4],
                                                 int tries_val) {
    double* hp = a.hyps + (size_t)h * 6;
    hp[0] = rvec[0]; hp[1] = rvec[1]; hp[2] = rvec[2];
    hp[3] = T[0]; hp[4] = T[1]; hp[5] = T[2];
    double* hr = a.hyps_R + (size_t)h * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) hr[k] = R[k];
    store_rt32(a, h, map, R, T);
    int* sx = a.sample_xy + (size_t)h * 8;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        sx[2 * j] = cx[j];
        sx[2 * j + 1] = cy[j];
    }
    a.tries[h] = tries_val;
}

// How hard is an expert's map for the sampler?  Two counters per expert (bin e % 1024, behind the four list counters of
// samp_count; a single frame, several experts, a few thousand hypotheses -- the shape in which most pending hypotheses are
// easy ones): hypotheses assigned / hypotheses the first tries left pending.  An expert on whose map (nearly) every
// hypothesis is still pending needs ~10^3 tries per hypothesis (wrong expert); one that settled most of its hypotheses
// needs a dozen more for the few that missed.  k_sample_prescreen hands out its helper wavefronts by that -- scheduling
// only: which wavefront screens which 64-try round never changes what is accepted.
constexpr int ESAC_STAT_BINS = 1024;
constexpr int ESAC_CLASSES_MAX_N = 8192;
__host__ __device__ __forceinline__ bool expert_stats_on(const KArgs& a) { return a.E > 1 && a.frames == 1 && a.N <= ESAC_CLASSES_MAX_N; }
__device__ __forceinline__ int* expert_stats(const KArgs& a, int e) { return a.samp_count + 4 + 2 * (e & (ESAC_STAT_BINS - 1)); }

constexpr int SAMPLE_PENDING = -2;  // tries[h] between the two phases of the throughput-shaped sampling
// hypothesis h of this frame goes on to the screened chain (k_sample_prescreen works from the list).  Called by the
// lanes of a wavefront that have one (`mine`; all lanes must call): one atomic per wavefront reserves the list slots.
__device__ __forceinline__ void mark_pending(const KArgs& a, int h, int e, bool mine) {
    const unsigned long long m = __ballot(mine);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(a.samp_count + 1, __popcll(m));
    base = __shfl(base, leader);
    if (!mine) return;
    a.tries[h] = SAMPLE_PENDING;
    if (a.spec_flag) a.spec_flag[h] = 1;  // speculative forward: a straggler -- the launch stream's kernels leave it alone
    a.best_try[h] = ~0ull;          // k_sample_decide: lowest accepted try << 32 | its list position
    a.samp_resume[h] = 0x7fffffff;  // k_sample_prescreen: first try not screened yet
    a.samp_round[h] = 0;
    a.samp_pending[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)blockIdx.y * a.N + h;
    if (expert_stats_on(a)) atomicAdd(expert_stats(a, e) + 1, 1);
}
constexpr int FIRST_PHASE_TRIES = 32;  // tries per hypothesis before the screened chain takes over
constexpr int ESAC_FIRST_WIDE_MAX = 8192;  // up to this many hypotheses: one pass, 32 lanes per hypothesis (else two passes of 16)

// Throughput shape, first phase: a hypothesis on a usable map is accepted within its first few tries, so a whole
// wavefront per hypothesis solves ~60 P3P problems nobody needs.  Here a wavefront serves SEVERAL hypotheses, TRIES tries
// each: tries [first_try, first_try + TRIES).  What is not accepted stays pending for the next pass and finally for the
// screened chain (k_pending_list -> k_sample_prescreen ...).
// TRIES = 16: a wavefront serves four hypotheses, 16 tries each per pass (thousands of hypotheses: two passes, the second
// one only for wavefronts with a hypothesis still open); TRIES = 32: two hypotheses, 32 tries in one pass -- with a few
// thousand hypotheses (config 4: 4096) the 16-try shape is 1024 wavefronts, half of what the chip holds, twice in a row
// (47 us; one pass of 32: 30 us).
template <int TRIES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sample_first(KArgs a) {
    constexpr int HPW = 64 / TRIES;  // hypotheses per wavefront
    frame_view(a);
    const int lane = threadIdx.x, grp = lane / TRIES, t = a.first_try + (lane & (TRIES - 1));
    const int h = blockIdx.x * HPW + grp;
    const int hc = h < a.N ? h : a.N - 1;
    const bool mine_pending = h < a.N && (a.first_try == 0 || a.tries[hc] == SAMPLE_PENDING);
    if (!__any(mine_pending)) return;  // all hypotheses of this wavefront are done
    const bool active = mine_pending && t < a.max_tries;
    if (a.first_try == 0 && h < a.N && (lane & (TRIES - 1)) == 0) flag_bad_assignment(a, h);
    const int e = expert_of(a, hc);
    const int P = a.H * a.W;
    const float* __restrict__ map = a.sc + (size_t)e * 3 * P;
    const Philox rng(a.seed, a.call);
    const Cam cam = make_cam(a);
    int cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0};
    double rvec[3] = {0, 0, 0}, T[3] = {0, 0, 0};
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool accepted = false;
    if (active) {
        V3 Pt[4];
        float Pf[4][3];
        double mu[4], mv[4], Rp[9], Tp[3];
        gather_sample(a, map, P, rng, (uint32_t)global_hyp(a, hc), (uint32_t)t, cx, cy, Pt, Pf, mu, mv);
        double reproj2 = 0;
        if (p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2) && (t == a.max_tries - 1 || !cannot_pass(reproj2, (double)a.tau)))
            accepted = accept_sample(Rp, Tp, Pf, mu, mv, cam, (double)a.tau, rvec, T, R);
    }
    const unsigned long long m = __ballot(accepted);
    const unsigned mine = (unsigned)(m >> (TRIES * grp)) & (TRIES == 32 ? 0xffffffffu : 0xffffu);
    if (mine_pending) {
        if (mine) {
            const int first = __ffs((int)mine) - 1;
            if ((lane & (TRIES - 1)) == first) store_hypothesis(a, h, map, rvec, T, R, cx, cy, a.first_try + first);
        } else if (a.max_tries <= a.first_try + TRIES) {
            if (t == a.max_tries - 1) store_hypothesis(a, h, map, rvec, T, R, cx, cy, -1);  // budget exhausted: last state remains
        } else if ((lane & (TRIES - 1)) == 0) {
            a.tries[h] = SAMPLE_PENDING;  // k_pending_list gathers what the last pass leaves pending
        }
    }
}

// The hypotheses the first passes left pending, as a list for the screened chain (k_sample_prescreen): a ballot prefix per
// wavefront, a block scan, ONE global atomic per 1024 hypotheses.  (Appending from k_sample_first itself -- 15,000 single-
// lane atomics on one address at config 5a -- cost 20 us; one per wavefront still 10.)
__global__ __launch_bounds__(1024) void k_pending_list(KArgs a) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    __shared__ int s_stat[2 * ESAC_STAT_BINS];
    frame_view(a);
    const int h = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // first_try == 0: no first phase has run (launch_sample: many hypotheses on several experts) -- everything is pending
    const bool mine = h < a.N && (a.first_try == 0 || a.tries[h] == SAMPLE_PENDING);
    if (a.first_try == 0 && h < a.N) {
        flag_bad_assignment(a, h);
        a.tries[h] = SAMPLE_PENDING;
    }
    if (a.spec_flag && h < a.N) a.spec_flag[h] = mine ? 1 : 0;  // speculative forward: what the first pass settled / left to the chain
    const unsigned long long m = __ballot(mine);
    if (lane == 0) s_wave[wave] = __popcll(m);
    const bool stats = expert_stats_on(a);  // assigned / pending per expert (see expert_stats)
    if (stats) {
        s_stat[threadIdx.x] = 0;
        s_stat[threadIdx.x + 1024] = 0;
        __syncthreads();
        if (h < a.N) {
            const int bin = expert_of(a, h) & (ESAC_STAT_BINS - 1);
            atomicAdd(s_stat + 2 * bin, 1);
            if (mine) atomicAdd(s_stat + 2 * bin + 1, 1);
        }
    }
    __syncthreads();
    if (stats) {
        const int v0 = s_stat[threadIdx.x], v1 = s_stat[threadIdx.x + 1024];
        if (v0) atomicAdd(a.samp_count + 4 + threadIdx.x, v0);
        if (v1) atomicAdd(a.samp_count + 4 + threadIdx.x + 1024, v1);
    }
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; w++) {
            const int c = s_wave[w];
            s_wave[w] = tot;
            tot += c;
        }
        s_base = tot ? atomicAdd(a.samp_count + 1, tot) : 0;
    }
    __syncthreads();
    if (!mine) return;
    a.best_try[h] = ~0ull;          // k_sample_decide: lowest accepted try << 32 | its list position
    a.samp_resume[h] = 0x7fffffff;  // k_sample_prescreen: first try not screened yet
    a.samp_round[h] = 0;
    a.samp_pending[s_base + s_wave[wave] + __popcll(m & ((1ull << lane) - 1ull))] = (int)blockIdx.y * a.N + h;
}

// SAMPLE_B = lanes per hypothesis.  A single call wants latency (4 wavefronts per hypothesis: nearly every hypothesis
// is accepted in round one, the other CUs are idle anyway); thousands of hypotheses in flight (many experts, batched
// frames) want throughput (one wavefront, one try per lane, no wasted solves).
// LPT = lanes per try in the first rounds (tries below SAMPLE_B): the (up to four) candidates of a try -- lengths +
// alignment + 4th-point error per quartic root, the long part of the solver -- are dealt to LPT lanes, 4 / LPT each, and
// the lanes agree on the winner through shuffles.  LPT = 4: the dependent chain of a try is one candidate long, 64 tries a
// round at 256 lanes (config 3's shape: two hypotheses per CU).  LPT = 2 (the single frame, round 4): two candidates
// per lane, 128 tries a round -- a hypothesis on a usable map is accepted at try ~15 on average and 1.8 % of them need
// more than 64, so with 256 hypotheses three frames in four paid a second 64-try round (8.5 us); one in twenty needs a
// second 128-try one.  The chip holds no more than this: the fp64 solver with its decision path needs ~445 registers,
// one wavefront per SIMD, and 256 hypotheses x 4 wavefronts fill the 1024 SIMDs.
// A hypothesis that needs hundreds of tries (wrong expert) continues with one try per lane, the throughput shape.
template <int SAMPLE_B, int LPT>
__global__ __launch_bounds__(SAMPLE_B) void k_sample(KArgs a) {
    static_assert(LPT == 1 || LPT == 2 || LPT == 4, "lanes per try");
    constexpr int CPLN = 4 / LPT;  // candidates per lane in the shared rounds
    __shared__ int s_first[2][SAMPLE_B / 64];
    __shared__ double s_pose[CPLN > 1 ? SAMPLE_B * 12 : 1];  // [value][lane]: the best candidate's pose so far, per lane
    frame_view(a);
    const int h = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = expert_of(a, h);
    const int P = a.H * a.W;
    const float* __restrict__ map = a.sc + (size_t)e * 3 * P;
    const Philox rng(a.seed, a.call);
    const Cam cam = make_cam(a);
    const uint32_t gh = (uint32_t)global_hyp(a, h);
    const double tau = (double)a.tau;
    if (a.first_try > 0 && a.tries[h] != SAMPLE_PENDING) return;  // phase 2 of the throughput shape: done in phase 1
    if (a.first_try == 0 && threadIdx.x == 0) flag_bad_assignment(a, h);
    if (a.handover != 0x7fffffff && threadIdx.x == 0 && expert_stats_on(a)) atomicAdd(expert_stats(a, e), 1);  // (see expert_stats)

    int parity = 0;
    for (int base = a.first_try, TRIES = 0; base < a.max_tries; base += TRIES, parity ^= 1) {
        const bool quad = LPT > 1 && base < SAMPLE_B;  // workgroup-uniform: the first SAMPLE_B tries go LPT lanes a try
        TRIES = quad ? SAMPLE_B / LPT : SAMPLE_B;
        const int t = base + (quad ? (int)threadIdx.x / LPT : (int)threadIdx.x);
        const int sub = threadIdx.x & (LPT - 1);  // shared rounds only: this lane evaluates roots sub, sub + LPT, ...
        bool holder = !quad;                      // the lane that carries the try's final state (pose or zero pose)
        const bool active = t < a.max_tries;
        int cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0};
        double rvec[3] = {0, 0, 0}, T[3] = {0, 0, 0};
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        bool accepted = false;
        if (active) {
            V3 Pt[4];
            float Pf[4][3];
            double mu[4], mv[4];
            gather_sample(a, map, P, rng, gh, (uint32_t)t, cx, cy, Pt, Pf, mu, mv);
            double Rp[9], Tp[3], reproj2 = 0;
            bool solved;
            if (LPT > 1 && quad) {
                P3PSetup S;
                const bool ok = p3p_setup(Pt, mu, mv, cam, S);
                // Step k: the lanes evaluate roots k * LPT .. k * LPT + LPT - 1 (this lane: root k * LPT + sub), exchange
                // (valid, error) and continue the reference's sequential scan over the candidates (same `>` rule, same NaN
                // behaviour) -- every lane carries the scan's state, and the lane whose candidate has just become the best
                // keeps its pose: after the last step the winner's lane holds the winner's pose.
                const int lane0 = lane & ~(LPT - 1);
                bool have = false;
                double min_reproj = 0;
                int win = -1;
#pragma nounroll
                for (int k = 0; k < CPLN; k++) {
                    const int root = k * LPT + sub;
                    const double x = root == 0 ? S.x[0] : root == 1 ? S.x[1] : root == 2 ? S.x[2] : S.x[3];
                    double R1[9], T1[3], rp = 0;
                    const bool vk = ok && root < S.n && p3p_candidate(S, x, Pt, mu[3], mv[3], cam, R1, T1, rp);
#pragma unroll
                    for (int j = 0; j < LPT; j++) {
                        const bool vi = __shfl((int)vk, lane0 + j) != 0;
                        const double ri = __shfl(rp, lane0 + j);
                        if (vi && (!have || min_reproj > ri)) {
                            have = true;
                            min_reproj = ri;
                            win = k * LPT + j;
                        }
                    }
                    if (win == root) {  // (more than one step: the pose waits in LDS, not in 24 registers across the next solve)
#pragma unroll
                        for (int q = 0; q < 9; q++) {
                            if (CPLN > 1) s_pose[q * SAMPLE_B + threadIdx.x] = R1[q];
                            else Rp[q] = R1[q];
                        }
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            if (CPLN > 1) s_pose[(9 + q) * SAMPLE_B + threadIdx.x] = T1[q];
                            else Tp[q] = T1[q];
                        }
                    }
                }
                solved = have && (win & (LPT - 1)) == sub;
                if (CPLN > 1 && solved) {
#pragma unroll
                    for (int q = 0; q < 9; q++) Rp[q] = s_pose[q * SAMPLE_B + threadIdx.x];
#pragma unroll
                    for (int q = 0; q < 3; q++) Tp[q] = s_pose[(9 + q) * SAMPLE_B + threadIdx.x];
                }
                holder = solved || (!have && sub == 0);
                reproj2 = min_reproj;
            } else {
                solved = p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2);
            }
            if (solved && (t == a.max_tries - 1 || !cannot_pass(reproj2, tau)))
                accepted = accept_sample(Rp, Tp, Pf, mu, mv, cam, tau, rvec, T, R);
            // a failed solve leaves the zero pose (safeSolvePnP, esac_util.h:107-111)
        }
        // lowest accepted try of the round = the try the reference's sequential loop stops at
        const unsigned long long m = __ballot(accepted);
        if (lane == 0) {
            const int first_lane = __ffsll((long long)m) - 1;
            s_first[parity][wave] = m ? base + (quad ? wave * (64 / LPT) + first_lane / LPT : wave * 64 + first_lane) : 0x7fffffff;
        }
        __syncthreads();
        int first = s_first[parity][0];
#pragma unroll
        for (int w = 1; w < SAMPLE_B / 64; w++) first = min(first, s_first[parity][w]);
        const bool last_round = base + TRIES >= a.max_tries;
        int writer = -1, tries_val = -1;
        if (first != 0x7fffffff) {
            writer = first;
            tries_val = first;
        } else if (last_round) {
            writer = a.max_tries - 1;  // budget exhausted: state of the last try remains
        }
        if (writer >= 0) {
            if (t == writer && holder) store_hypothesis(a, h, map, rvec, T, R, cx, cy, tries_val);
            if (a.spec_flag && threadIdx.x == 0) a.spec_flag[h] = 0;  // settled by this pass
            return;
        }
        if (base + TRIES >= a.handover) {  // a straggler (wrong expert): the spread, screened search takes over from here
            if (threadIdx.x < 64) mark_pending(a, h, e, threadIdx.x == 0);
            return;
        }
    }
}

// ---- throughput shape, tries >= first_try: screened sampling -----------------------------------------------------------
// A hypothesis of a wrong expert needs ~10^3 tries (its 4 points only pass tau by luck), and half of the fp64 work of a
// try -- per candidate root the least-squares triangle alignment with its Newton iterations, the pose conversion and
// the reprojections -- is spent on candidates whose 4th point misses its pixel by hundreds of pixels.  Here every try
// first runs the roots and depths of the fp64 route (p3p_setup, p3p_candidate_lengths: the same doubles) and then the fp32
// SCREEN of p3p_screen.hpp instead of the alignment; only a try the screen cannot rule out ("maybe": 4th point within
// tau + SCREEN_MARGIN, or a numerically delicate configuration) gets the full fp64 decision, exactly as before.  The
// screen is one-sided (calibrated on 4e7 tries: scripts/dev/p3p_screen_probe.py), so the accepted try is still the try
// the reference's sequential loop stops at (esac_util.h:152-223).
// One wavefront per hypothesis, 64 tries per round.  "Maybe" tries are queued in LDS (in try order) and decided 64 lanes
// wide: a lone maybe would otherwise cost the whole wavefront a full fp64 solve.  The queue is flushed when it holds
// SCREEN_FLUSH tries, when the screen itself sees a 4th point within tau (almost certainly the accepted try), and at
// the end of the budget.
constexpr int ESAC_CHAIN_WAVES = 8192;  // wavefronts of the screened search that work whatever the number of pending hypotheses is
constexpr int ESAC_RESIDENT_WAVES = 2048;  // k_sample_prescreen: 256 CUs x 4 SIMDs x 2
constexpr float SCREEN_MARGIN = 3.0f;  // pixels; the largest screen error of an fp64-accepted try in calibration: tau + 0.008
constexpr int SCREEN_FLUSH = 8;
constexpr int SCREEN_QUEUE = 128;  // >= SCREEN_FLUSH - 1 + 64

// The screening loop alone, for TWO wavefronts per SIMD: the full fp64 decision (alignment, pose conversion, best-candidate
// bookkeeping) is what pushes k_sample_screened to ~400 registers and one wavefront per SIMD, where every dependent fp64
// operation is paid at its full ~40-cycle latency.  The throughput shape therefore runs as a chain of four launches:
//   k_sample_prescreen  no decision code (256 registers, two wavefronts per SIMD): walks the tries of a pending
//                       hypothesis, appends every "maybe" try to ONE global list (h, try) and stops after the first
//                       round in which the screen itself sees the 4th point within tau (that try is accepted in all
//                       but a handful of cases), or when the budget is spent;
//   k_sample_decide     one LANE per listed try, all hypotheses together (a wrong-expert hypothesis lists 2-3 tries out
//                       of ~10^3): the full fp64 decision, atomicMin of the accepted try per hypothesis;
//   k_sample_screened<true>  one wavefront per pending hypothesis: commits the accepted try (copies the record the
//                       decision parked with its list entry); the rare hypothesis whose stop turned out a false alarm
//                       continues from its resume round, screening and deciding in one kernel; a spent budget leaves the
//                       state of the last try.
// Every try below a hypothesis' resume point has been screened, every "maybe" among them decided: the minimum accepted
// try is the try the reference's sequential loop stops at (esac_util.h:152-223).
// gridDim.z wavefronts share a hypothesis.  The rounds (64 tries each) are handed out IN ORDER by a per-hypothesis counter
// to whichever wavefront asks next (samp_round) -- not round-robin: with more workgroups than the chip holds at once,
// wavefront 0 of a hypothesis would walk rounds 0, Z, 2Z, ... alone, long past the accepted try sitting in a round whose
// owner has not been scheduled yet (measured: 4 instead of 2 wavefronts per hypothesis made config 5a 66 % slower).  A
// wavefront leaves as soon as the round it is handed starts at or beyond samp_resume[h] -- the lowest point at which some
// wavefront saw a strong candidate (atomicMin; 0x7fffffff until then).  With few hypotheses pending (a single frame with
// some wrong-expert stragglers) that divides the length of the tail by the number of wavefronts that are resident.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sample_prescreen(KArgs a) {
    // Every wavefront of the launch works: wavefront L serves pending hypothesis L % count of the list the first passes
    // built (k_sample: mark_pending; k_pending_list after the first passes), so a frame with 50 stragglers among 1024 hypotheses puts ~80
    // wavefronts on each of them instead of dispatching 65,000 workgroups that find their own hypothesis settled, and a
    // launch sized for the chip (a few thousand wavefronts) is enough whatever the number of pending hypotheses is.
    const int count = min(a.samp_count[1], a.N * a.frames);
    if (count == 0) return;
    const long long L = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    // ... with one exception: a single expert.  Every pending hypothesis is then an EASY one (a true-expert hypothesis that
    // missed its first 32 tries and needs a dozen more): one wavefront each settles it in a round, and the 28 each that a
    // launch sized for "every hypothesis pending" would put on the few thousand stragglers of a 256-frame batch are 28
    // start-up chains of dependent loads for nothing.  (Wrong-expert stragglers want every helper they can get: capping
    // them at eight per hypothesis cost config 4 18 us.)
    if (a.E == 1 && L >= ESAC_CHAIN_WAVES && L >= count) return;
    const int lane = threadIdx.x;
    const int entry = a.samp_pending[(int)(L % count)];
    const int fr = entry / a.N, h = entry - fr * a.N;
    frame_view(a, fr);
    const int e = expert_of(a, h);
    // Few hypotheses pending (several helpers of a hypothesis are resident AT ONCE and walk its rounds in lockstep): an EASY
    // hypothesis -- its expert settled most of its hypotheses in the first tries -- is accepted in its first round, and
    // every further helper is a wasted round (and a dozen "maybe" tries for k_sample_decide) that a wrong-expert straggler
    // (dozens of rounds) is waiting for.  Two helpers for those; the others take every wavefront they can get.
    // (Config 4, ~500 pending of which ~110 on wrong experts: 81 -> 70 us, 62 with the fresh stop flag below; config 3,
    // where up to half of the ~60 pending are easy: 40.5 -> 28.7 us with both, k_sample_decide 19.3 -> 12.6 us.)
    // Measured and dropped (round 3): wavefronts that look for another open hypothesis instead of leaving (213 us: a
    // search is three dependent gathers and costs more than a round), a list of the hard ones built by an extra kernel
    // with the spare wavefronts leaving after one scalar load (60 + 5 us), late helpers redirected to an open neighbour
    // (62 us) -- the kernel's length is neither the supply of helpers nor the wavefronts that find nothing to do.
    if (expert_stats_on(a) && L >= 2LL * count) {
        const int* st = expert_stats(a, e);
        if (2 * st[1] < st[0]) return;
    }
    const int P = a.H * a.W;
    const float* __restrict__ map = a.sc + (size_t)e * 3 * P;
    const Philox rng(a.seed, a.call);
    const Cam cam = make_cam(a);
    const uint32_t gh = (uint32_t)global_hyp(a, h);
    const float thr = a.tau + SCREEN_MARGIN;
    int* resume = a.samp_resume + h;
    // the stop flag one round old is fine while a hypothesis has one helper at a time (thousands pending); with the chip's
    // 2048 resident wavefronts on a few dozen hypotheses it is a whole superfluous round of every one of them
    const bool fresh = count <= ESAC_RESIDENT_WAVES;
    // Two round trips to L2 per 64-try round used to sit on this loop's critical path (~1.5 us each against ~6 us of
    // arithmetic, at two wavefronts per SIMD): the ticket for the round (atomicAdd with return) and, whenever a lane said
    // "maybe" (a third of the rounds), the slot in the global list.  The ticket for the NEXT round is now drawn while the
    // current one is evaluated (a ticket drawn and never used is harmless: every way out of this loop means that later
    // rounds need no screening), and the "maybe" tries are parked in LDS and flushed with one atomic when the wavefront
    // leaves (or 64 of them have gathered).
    __shared__ int s_park[128];  // tries; the wavefront serves ONE hypothesis
    int parked = 0;              // wave-uniform
    auto flush = [&]() {
        // every parked try gets a slot of the global list; what does not fit any more moves the resume point of the
        // hypothesis back to the (64-aligned) round of the first try that was dropped
        for (int q0 = 0; q0 < parked; q0 += 64) {
            const int nq = min(64, parked - q0);
            int pos0 = 0;
            if (lane == 0) pos0 = atomicAdd(a.samp_count, nq);
            pos0 = __shfl(pos0, 0);
            if (lane < nq) {
                const int t = s_park[q0 + lane];
                if (pos0 + lane < a.samp_cap) reinterpret_cast<int2*>(a.samp_entries)[pos0 + lane] = make_int2(entry, t);
                else atomicMin(resume, a.first_try + ((t - a.first_try) & ~63));
            }
        }
        parked = 0;
    };
    int r_next = 0;
    if (lane == 0) r_next = atomicAdd(a.samp_round + h, 1);
    int stop_at = __hip_atomic_load(resume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (bool again = false;; again = true) {
        const int r = __shfl(r_next, 0);  // (waits for the ticket drawn one round ago)
        const long long base = a.first_try + 64LL * r;
        if (base >= a.max_tries) break;
        if (fresh && again) stop_at = __hip_atomic_load(resume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (base >= stop_at) break;  // (otherwise read one round ago as well: a stale value costs at most one superfluous round)
        if (lane == 0) r_next = atomicAdd(a.samp_round + h, 1);
        stop_at = __hip_atomic_load(resume, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int t = (int)base + lane;
        bool maybe = false, strong = false;
        if (t < a.max_tries) {
            int cx[4], cy[4];
            V3 Pt[4];
            float Pf[4][3];
            double mu[4], mv[4];
            gather_sample(a, map, P, rng, gh, (uint32_t)t, cx, cy, Pt, Pf, mu, mv);
            ScreenSetup S;  // the screen's private (contracted, fast-cubic) copy of the roots and depths
            if (screen_setup(Pt, mu, mv, cam, S)) {
                const float err = p3p_screen_roots(S, Pf, (float)mu[3], (float)mv[3], a.focal, a.ppx, a.ppy);
                maybe = !(err > thr);
                strong = maybe && err >= 0.0f && err <= a.tau;
            }
        }
        const unsigned long long m = __ballot(maybe);
        if (m) {
            if (maybe) s_park[parked + __popcll(m & ((1ull << lane) - 1ull))] = t;
            parked += __popcll(m);
            if (__any(strong)) {  // everything up to and including this round is (about to be) listed: resume after it
                if (lane == 0) atomicMin(resume, (int)(base + 64 < a.max_tries ? base + 64 : a.max_tries));
                break;
            }
            if (parked > 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                flush();
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    flush();
}

// the fp64 route's decision for every listed try, lowest accepted try per hypothesis.  One LANE per try when the list is
// long (thousands of pending hypotheses: throughput); FOUR lanes per try -- lane q evaluates the candidate of quartic root
// q, as in k_sample's first rounds -- when it holds few entries (a frame with some dozens of stragglers: the kernel is
// then one dependent chain long, and the chain of a try is one candidate instead of up to four: 21 -> 12 us)
__global__ __launch_bounds__(64) void k_sample_decide(KArgs a0) {
    const int n = min(a0.samp_count[0], a0.samp_cap);
    const bool quad = 4LL * n <= (long long)gridDim.x * 64;  // kernel-uniform
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int i = quad ? slot >> 2 : slot;
    if ((quad ? blockIdx.x * 16 : blockIdx.x * 64) >= n) return;
    bool accepted = false;
    int hg = 0, t = 0;
    if (i < n) {
        const int2 ent = reinterpret_cast<const int2*>(a0.samp_entries)[i];
        hg = ent.x;
        t = ent.y;
        const int fr = hg / a0.N, h = hg - fr * a0.N;
        // the frame's view of the inputs, by hand (this kernel's grid is not per frame)
        const float* sc = a0.sc + (size_t)fr * a0.sc_frame_stride;
        const int64_t* assign = a0.assign + (size_t)fr * a0.N;
        const long long ev = a0.E == 1 ? 0 : assign[h];
        const int e = (unsigned long long)ev < (unsigned long long)a0.E ? (int)ev : 0;
        const int P = a0.H * a0.W;
        const float* __restrict__ map = sc + (size_t)e * 3 * P;
        const Philox rng(a0.seed, a0.call + (uint64_t)fr);
        const Cam cam = make_cam(a0);
        const uint32_t gh = (uint32_t)global_hyp(a0, h);
        int cx[4], cy[4];
        V3 Pt[4];
        float Pf[4][3];
        double mu[4], mv[4], Rp[9], Tp[3], reproj2 = 0;
        gather_sample(a0, map, P, rng, gh, (uint32_t)t, cx, cy, Pt, Pf, mu, mv);
        bool solved;
        if (quad) {  // the four lanes of this entry replay p3p_4pt's scan over the candidates (k_sample: same rule, same NaN behaviour)
            const int lane = threadIdx.x, root = lane & 3, quad0 = lane & ~3;
            P3PSetup S;
            const bool ok = p3p_setup(Pt, mu, mv, cam, S);
            const double x = root == 0 ? S.x[0] : root == 1 ? S.x[1] : root == 2 ? S.x[2] : S.x[3];
            double reproj = 0;
            const bool valid = ok && root < S.n && p3p_candidate(S, x, Pt, mu[3], mv[3], cam, Rp, Tp, reproj);
            bool have = false;
            double min_reproj = 0;
            int win = -1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool vi = __shfl((int)valid, quad0 + k) != 0;
                const double ri = __shfl(reproj, quad0 + k);
                if (vi && (!have || min_reproj > ri)) {
                    have = true;
                    min_reproj = ri;
                    win = k;
                }
            }
            solved = have && win == root;  // the lane that holds the chosen candidate carries on
            reproj2 = min_reproj;
        } else {
            solved = p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2);
        }
        if (solved && !cannot_pass(reproj2, (double)a0.tau)) {
            double rvec[3], T[3], R[9];
            accepted = accept_sample(Rp, Tp, Pf, mu, mv, cam, (double)a0.tau, rvec, T, R);
            if (accepted) {  // park the solved hypothesis with its list entry: the commit kernel only copies the winner's
                double* cd = a0.samp_cand + (size_t)i * ESAC_CAND_DOUBLES;
                cd[0] = rvec[0]; cd[1] = rvec[1]; cd[2] = rvec[2]; cd[3] = T[0]; cd[4] = T[1]; cd[5] = T[2];
#pragma unroll
                for (int k = 0; k < 9; k++) cd[6 + k] = R[k];
                const Centre c = map_centre(a0, map);
                float* cf = reinterpret_cast<float*>(cd + 15);  // 12 floats: [R | t + R c] (store_rt32)
#pragma unroll
                for (int k = 0; k < 9; k++) cf[k] = (float)R[k];
                cf[9] = (float)(R[0] * (double)c.x + R[1] * (double)c.y + R[2] * (double)c.z + T[0]);
                cf[10] = (float)(R[3] * (double)c.x + R[4] * (double)c.y + R[5] * (double)c.z + T[1]);
                cf[11] = (float)(R[6] * (double)c.x + R[7] * (double)c.y + R[8] * (double)c.z + T[2]);
                int* ci = reinterpret_cast<int*>(cd + 21);     // 8 ints: the sampled cells
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    ci[2 * j] = cx[j];
                    ci[2 * j + 1] = cy[j];
                }
            }
        }
    }
    // lowest accepted try per hypothesis; the list position rides in the low word so that the winner's record is found
    if (accepted) atomicMin(a0.best_try + hg, ((unsigned long long)(unsigned)t << 32) | (unsigned)i);
}

// Commit pre-pass for calls with thousands of pending hypotheses: one LANE per hypothesis copies the record of its
// accepted try (see the COMMIT block of k_sample_screened<true>, which does the same with a whole wavefront per
// hypothesis -- fine for a frame's few stragglers, 33 us of dependent loads at 15,000); what it settles, the resume
// kernel finds settled.
__global__ __launch_bounds__(256) void k_sample_commit(KArgs a) {
    frame_view(a);
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= a.N || a.tries[h] != SAMPLE_PENDING) return;
    const unsigned long long found = a.best_try[h];
    if (found == ~0ull || (int)(found >> 32) >= a.samp_resume[h]) return;  // not settled: k_sample_screened<true> decides
    const double* cd = a.samp_cand + (size_t)(unsigned)found * ESAC_CAND_DOUBLES;
#pragma unroll
    for (int k = 0; k < 6; k++) a.hyps[(size_t)h * 6 + k] = cd[k];
#pragma unroll
    for (int k = 0; k < 9; k++) a.hyps_R[(size_t)h * 9 + k] = cd[6 + k];
#pragma unroll
    for (int k = 0; k < 6; k++) reinterpret_cast<double*>(a.rt32 + (size_t)h * 12)[k] = cd[15 + k];
#pragma unroll
    for (int k = 0; k < 4; k++) reinterpret_cast<double*>(a.sample_xy + (size_t)h * 8)[k] = cd[21 + k];
    a.tries[h] = (int)(found >> 32);
}

// RESUME: the end of the screened chain -- commit the accepted try of a pending hypothesis, or continue where
// k_sample_prescreen stopped.
template <bool RESUME>
__global__ __launch_bounds__(64) void k_sample_screened(KArgs a) {
    __shared__ int s_queue[SCREEN_QUEUE];
    frame_view(a);
    const int h = blockIdx.x, lane = threadIdx.x;
    // the last kernel of the screened chain leaves the chain's counters (list lengths, per-expert counters) at zero for the
    // next call: nothing after k_sample_decide reads them, and a fill in front of every sampling launch is a 3 us
    // launch and a kernel boundary on a 250 us call (esac_capi.hip zeroes them when the workspace is allocated)
    if (RESUME && blockIdx.x == 0 && blockIdx.y == 0) {
        // (the length of the pending list survives in n_contenders[1]: the speculative route scores exactly those hypotheses
        // behind this kernel, k_score_stragglers)
        if (lane == 0) a.n_contenders[1] = min(a.samp_count[1], a.N * a.frames);
        for (int i = lane; i < 4 + 2 * ESAC_STAT_BINS; i += 64) a.samp_count[i] = 0;
    }
    // (RESUME: the list builders marked what is pending -- also when the chain started at try 0)
    if ((RESUME || a.first_try > 0) && a.tries[h] != SAMPLE_PENDING) return;
    if (!RESUME && a.first_try == 0 && lane == 0) flag_bad_assignment(a, h);
    const int e = expert_of(a, h);
    const int P = a.H * a.W;
    const float* __restrict__ map = a.sc + (size_t)e * 3 * P;
    const Philox rng(a.seed, a.call);
    const Cam cam = make_cam(a);
    const uint32_t gh = (uint32_t)global_hyp(a, h);
    const double tau = (double)a.tau;
    const float thr = a.tau + SCREEN_MARGIN;
    int qcount = 0;  // wave-uniform
    long long first = a.first_try;
    if (RESUME) {
        // COMMIT: the lowest accepted try the decision kernel found for this hypothesis.  It only counts if every try below
        // it has been screened: with several wavefronts per hypothesis one of them may have run ahead of the final resume
        // point (a "strong" candidate that the decision then rejected) -- such an entry is ignored here and found again, in
        // order, by the search below.  The record (rvec, tvec | 12 floats rt32 | 8 ints cells = 16 doubles) is copied by 16 lanes.
        const unsigned long long found = a.best_try[h];
        int resume = a.samp_resume[h];
        if (found != ~0ull && (int)(found >> 32) < resume) {
            const double* cd = a.samp_cand + (size_t)(unsigned)found * ESAC_CAND_DOUBLES;
            if (lane < 6) a.hyps[(size_t)h * 6 + lane] = cd[lane];
            else if (lane < 15) a.hyps_R[(size_t)h * 9 + lane - 6] = cd[lane];
            else if (lane < 21) reinterpret_cast<double*>(a.rt32 + (size_t)h * 12)[lane - 15] = cd[lane];  // 12 floats = 6 doubles (48-byte rows: 8-byte aligned)
            else if (lane < 25) reinterpret_cast<double*>(a.sample_xy + (size_t)h * 8)[lane - 21] = cd[lane];  // 8 ints = 4 doubles
            if (lane == 0) a.tries[h] = (int)(found >> 32);
            return;
        }
        // no accepted try below the resume point: a false alarm / a full list (continue from there), or no stop at all
        // (the whole budget has been screened: the pass below re-solves the last try, whose state remains)
        first = resume > a.max_tries ? a.max_tries : resume;
    }
    for (long long base = first;; base += 64) {
        const bool more = base < a.max_tries;
        bool strong = false;
        if (more) {
            const int t = (int)base + lane;
            bool maybe = false;
            if (t < a.max_tries) {
                int cx[4], cy[4];
                V3 Pt[4];
                float Pf[4][3];
                double mu[4], mv[4];
                gather_sample(a, map, P, rng, gh, (uint32_t)t, cx, cy, Pt, Pf, mu, mv);
                ScreenSetup S;  // the screen's private (contracted, fast-cubic) copy of the roots and depths
                if (screen_setup(Pt, mu, mv, cam, S)) {
                    const float err = p3p_screen_roots(S, Pf, (float)mu[3], (float)mv[3], a.focal, a.ppx, a.ppy);
                    maybe = !(err > thr);                        // delicate (-1) and NaN included
                    strong = maybe && err >= 0.0f && err <= a.tau;  // the screen itself sees an inlier
                }
            }
            const unsigned long long m = __ballot(maybe);
            if (maybe) s_queue[qcount + __popcll(m & ((1ull << lane) - 1ull))] = t;  // try order: lanes ascend, rounds ascend
            qcount += __popcll(m);
        }
        if (!(!more || __any(strong) || qcount >= SCREEN_FLUSH)) continue;
        // ---- decide the queued tries in the fp64 route, 64 at a time (the queue never holds more than 7 + 64); ONE code
        // site (the kernel must stay inside the instruction cache): when the budget is exhausted without an accepted
        // try, a last pass re-solves the final try, whose state remains (esac_util.h:152-223 leaves the pose of the last
        // iteration, a failed solve the zero pose)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        bool final_pass = false;
        for (int q0 = 0;; q0 += 64) {
            if (!final_pass && q0 >= qcount) {
                if (more) break;
                final_pass = true;
            }
            const bool mine = final_pass ? lane == 0 : q0 + lane < qcount;
            int cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0};
            double rvec[3] = {0, 0, 0}, T[3] = {0, 0, 0};
            double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            bool accepted = false;
            int t = 0;
            if (mine) {
                t = final_pass ? a.max_tries - 1 : s_queue[q0 + lane];
                V3 Pt[4];
                float Pf[4][3];
                double mu[4], mv[4], Rp[9], Tp[3], reproj2 = 0;
                gather_sample(a, map, P, rng, gh, (uint32_t)t, cx, cy, Pt, Pf, mu, mv);
                if (p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2) && (final_pass || !cannot_pass(reproj2, tau)))
                    accepted = accept_sample(Rp, Tp, Pf, mu, mv, cam, tau, rvec, T, R);
            }
            const unsigned long long am = __ballot(accepted);
            if (am) {  // lowest queue position = lowest try index
                if (lane == __ffsll((long long)am) - 1) store_hypothesis(a, h, map, rvec, T, R, cx, cy, t);
                return;
            }
            if (final_pass) {
                if (lane == 0) store_hypothesis(a, h, map, rvec, T, R, cx, cy, -1);
                return;
            }
        }
        qcount = 0;
    }
}

// (rvec,tvec) -> float [R | t + R c] for the fp32 scoring stream (hypotheses handed in through esac_hip_write_hyps)
__global__ void k_hyps_to_rt32(KArgs a) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= a.N) return;
    double R[9];
    const double* hp = a.hyps + (size_t)h * 6;
    const double r[3] = {hp[0], hp[1], hp[2]};
    const double t[3] = {hp[3], hp[4], hp[5]};
    rodrigues_vec2mat<false>(r, R, nullptr);
#pragma unroll
    for (int k = 0; k < 9; k++) a.hyps_R[(size_t)h * 9 + k] = R[k];
    store_rt32(a, h, a.sc + (size_t)expert_of(a, h) * 3 * a.H * a.W, R, t);
}

// ================================================================= K2: fused soft-inlier score, fp32
struct PoseF {
    float r0, r1, r2, r3, r4, r5, r6, r7, r8, t0, t1, t2;
};

// one cell: project, clamp, 1 - sigmoid(beta*(err - tau)) = 1 / (1 + exp(beta*(err - tau)))
__device__ __forceinline__ float soft_inlier_fast(const PoseF& p, float fx, float fy, float cx, float cy, float X,
                                                  float Y, float Z, float px, float py, float max_reproj,
                                                  float beta_log2e, float tau) {
    const float xc = fmaf(p.r0, X, fmaf(p.r1, Y, fmaf(p.r2, Z, p.t0)));
    const float yc = fmaf(p.r3, X, fmaf(p.r4, Y, fmaf(p.r5, Z, p.t1)));
    const float zc = fmaf(p.r6, X, fmaf(p.r7, Y, fmaf(p.r8, Z, p.t2)));
    const float iz = (zc != 0.0f) ? __builtin_amdgcn_rcpf(zc) : 1.0f;
    const float du = px - fmaf(fx, xc * iz, cx);
    const float dv = py - fmaf(fy, yc * iz, cy);
    const float err = fminf(__builtin_amdgcn_sqrtf(fmaf(du, du, dv * dv)), max_reproj);
    const float ex = __builtin_amdgcn_exp2f((err - tau) * beta_log2e);
    return __builtin_amdgcn_rcpf(1.0f + ex);
}

// two cells at a time with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: measured 24 % faster than the scalar
// form in the tile-stationary kernel); component-wise the same operations in the same order as soft_inlier_fast
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 soft_inlier_fast2(const PoseF& p, float fx, float fy, float cx, float cy, f32x2 X, f32x2 Y, f32x2 Z,
                                                   f32x2 px, float py, float max_reproj, float beta_log2e, float tau) {
    const f32x2 xc = __builtin_elementwise_fma(splat2(p.r0), X, __builtin_elementwise_fma(splat2(p.r1), Y, __builtin_elementwise_fma(splat2(p.r2), Z, splat2(p.t0))));
    const f32x2 yc = __builtin_elementwise_fma(splat2(p.r3), X, __builtin_elementwise_fma(splat2(p.r4), Y, __builtin_elementwise_fma(splat2(p.r5), Z, splat2(p.t1))));
    const f32x2 zc = __builtin_elementwise_fma(splat2(p.r6), X, __builtin_elementwise_fma(splat2(p.r7), Y, __builtin_elementwise_fma(splat2(p.r8), Z, splat2(p.t2))));
    // err = sqrt(d2n) / |zc|, d2n = ((px - cx) zc - fx xc)^2 + ((py - cy) zc - fy yc)^2, as d2n * rsq(d2n zc^2): one
    // transcendental instead of rcp + sqrt (esac_score_tiled.hip: soft_inlier_tile2 has the edge cases)
    const f32x2 da = __builtin_elementwise_fma(px - splat2(cx), zc, -(splat2(fx) * xc));
    const f32x2 db = __builtin_elementwise_fma(splat2(py - cy), zc, -(splat2(fy) * yc));
    const f32x2 d2n = __builtin_elementwise_fma(da, da, db * db);
    const f32x2 q = __builtin_elementwise_fma(d2n, zc * zc, splat2(1e-36f));
    // (q - q: 0, or NaN when d2n zc^2 overflowed -- a hypothesis far beyond the scene: rsq(inf) = 0 would read err = 0, a perfect
    // inlier; the NaN ends in the clamp below like the rcp + sqrt form did.  One packed add per pair of cells.)
    const f32x2 er = __builtin_elementwise_fma(d2n, f32x2{__builtin_amdgcn_rsqf(q.x), __builtin_amdgcn_rsqf(q.y)}, q - q);
    const f32x2 err = {fminf(er.x, max_reproj), fminf(er.y, max_reproj)};
    const f32x2 arg = (err - splat2(tau)) * splat2(beta_log2e);
    const f32x2 den = splat2(1.0f) + f32x2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    return f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
}

// the fp32 score of hypothesis h by one workgroup of B threads; returns it in thread 0 (already stored to fast_scores[h])
template <int B>
__device__ __forceinline__ float score_fast_one(const KArgs& a, int h, float* s_w) {
    // optional device-side span measurement (timing mode): the kernel's duration is
    // max(end) - min(start) over its workgroups, on the constant 100 MHz wall clock
    long long t_start = 0;
    if (a.tstamps && threadIdx.x == 0) t_start = wall_clock64();
    const int e = expert_of(a, h);
    const int P = a.H * a.W;
    const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
    const float* __restrict__ my = mx + P;
    const float* __restrict__ mz = my + P;
    const float* rt = a.rt32 + (size_t)h * 12;  // wave-uniform -> scalar loads
    const PoseF p{rt[0], rt[1], rt[2], rt[3], rt[4], rt[5], rt[6], rt[7], rt[8], rt[9], rt[10], rt[11]};
    const Centre o = map_centre(a, mx);  // rt32's translation is relative to this origin (store_rt32)
    const float fx = a.focal, fy = a.focal, cx = a.ppx, cy = a.ppy;
    const float beta_log2e = a.beta * 1.4426950408889634f;
    float acc = 0.0f;
    if ((a.W & 3) == 0) {
        // 4 consecutive cells of one row per lane: 16-byte coalesced loads from each plane
        const int nq = P >> 2;
        const int wq = a.W >> 2;
        const float4* __restrict__ qx = reinterpret_cast<const float4*>(mx);
        const float4* __restrict__ qy = reinterpret_cast<const float4*>(my);
        const float4* __restrict__ qz = reinterpret_cast<const float4*>(mz);
        const float step = (float)a.sub;
        for (int i = threadIdx.x; i < nq; i += B) {
            const float4 X = qx[i], Y = qy[i], Z = qz[i];
            const int row = i / wq;
            const int col = (i - row * wq) << 2;
            const float py = cell_py(a, row);
            const float px = cell_px(a, col);
            const f32x2 s01 = soft_inlier_fast2(p, fx, fy, cx, cy, f32x2{X.x - o.x, X.y - o.x}, f32x2{Y.x - o.y, Y.y - o.y}, f32x2{Z.x - o.z, Z.y - o.z},
                                                f32x2{px, px + step}, py, a.max_reproj, beta_log2e, a.tau);
            const f32x2 s23 = soft_inlier_fast2(p, fx, fy, cx, cy, f32x2{X.z - o.x, X.w - o.x}, f32x2{Y.z - o.y, Y.w - o.y}, f32x2{Z.z - o.z, Z.w - o.z},
                                                f32x2{px + 2 * step, px + 3 * step}, py, a.max_reproj, beta_log2e, a.tau);
            acc += s01.x;  // the same summation order as the scalar form
            acc += s01.y;
            acc += s23.x;
            acc += s23.y;
        }
    } else {
        for (int i = threadIdx.x; i < P; i += B) {
            const int row = i / a.W;
            const int col = i - row * a.W;
            acc += soft_inlier_fast(p, fx, fy, cx, cy, mx[i] - o.x, my[i] - o.y, mz[i] - o.z, cell_px(a, col), cell_py(a, row),
                                    a.max_reproj, beta_log2e, a.tau);
        }
    }
    const float w = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    float out = 0.0f;
    if (threadIdx.x == 0) {
        double tot = 0;
#pragma unroll
        for (int k = 0; k < B / 64; k++) tot += (double)s_w[k];
        const float scale = a.alpha / a.W / a.H;  // float / int / int (esac_util.h:256)
        out = (float)(tot * (double)scale);
        a.fast_scores[h] = out;
        if (a.tstamps) {
            a.tstamps[2 * h] = t_start;
            a.tstamps[2 * h + 1] = wall_clock64();
        }
    }
    return out;
}

template <int B>
__global__ __launch_bounds__(B) void k_score_fast(KArgs a) {
    __shared__ float s_w[B / 64];
    frame_view(a);
    const int h = blockIdx.x;
    if (a.spec_mode == 1 && a.spec_flag[h] != 0) return;  // speculative forward: the settled hypotheses only (the stragglers: k_score_stragglers)
    (void)score_fast_one<B>(a, h, s_w);
}

// Speculative forward, the side stream's last kernel: the fp32 scores of the STRAGGLERS -- the pending list the chain worked
// through (samp_pending; its length in n_contenders[1], k_sample_screened<true>) -- and, from the last workgroup to finish, the
// "chain is done" word the join polls (spec_state[4]).  The scores are written through (sc1) and every workgroup arrives at a
// counter once its store has reached memory: no launch of its own for the word, no release fence per workgroup (an L2 write-back
// each).  A single frame (the route's condition).
template <int B>
__global__ __launch_bounds__(B) void k_score_stragglers(KArgs a) {
    __shared__ float s_w[B / 64];
    const int count = a.n_contenders[1];
    // only the workgroups that have a straggler to score arrive at the counter (1024 arrivals at one address were 5 us of this
    // kernel; a frame has a few hundred stragglers at most); no straggler at all: workgroup 0 writes the word
    const int nwork = count < (int)gridDim.x ? count : (int)gridDim.x;
    if ((int)blockIdx.x >= (nwork > 0 ? nwork : 1)) return;
    for (int e = blockIdx.x; e < count; e += gridDim.x) {
        const int h = a.samp_pending[e];
        const float v = score_fast_one<B>(a, h, s_w);
        if (threadIdx.x == 0) __hip_atomic_store(a.fast_scores + h, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();  // (s_w is rewritten by the next hypothesis)
    }
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // arrivals in two levels: workgroup b at counter 1 + b % 32 (each on a cache line of its own), the last one there at counter
        // 0 -- at most ~32 atomics on any one address (the 900 stragglers of the 12-expert shape at ONE counter were 4 us of this
        // kernel, which is the end of the call's critical path there)
        bool last = true;
        if (nwork > 1) {
            const int j = (int)blockIdx.x & (ESAC_SPEC_CNT_FAN - 1);
            const int expect = (nwork - j + ESAC_SPEC_CNT_FAN - 1) / ESAC_SPEC_CNT_FAN;  // workgroups b < nwork with b % FAN == j
            int* const sub = a.spec_cnt + ESAC_SPEC_CNT_STRIDE * (1 + j);
            last = __hip_atomic_fetch_add(sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1;
            if (last) {
                __hip_atomic_store(sub, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int groups = nwork < ESAC_SPEC_CNT_FAN ? nwork : ESAC_SPEC_CNT_FAN;
                last = __hip_atomic_fetch_add(a.spec_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
                if (last) __hip_atomic_store(a.spec_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (last) __hip_atomic_store(a.spec_state + 4, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ================================================================= K3: select + exact re-score of the contenders
// softMax / entropy statistics (esac_util.h:461-497) from the fp32-path scores, the band of contenders
// `score >= max - margin`, and the reference-arithmetic re-score of every contender -- ONE launch:
// every workgroup finds the fp32 maximum itself (N floats, trivial), then walks its strided share of the
// hypotheses; a contender is re-scored by the whole workgroup, anything else keeps its fp32 score.  Workgroup 0
// also reduces the statistics.  (One launch gap and one tiny kernel less on the single-frame critical path.)
// WIDE = false: at most one hypothesis per workgroup and no cell ranges (a single frame of <= 256 hypotheses on a grid that
// fits the caches: the headline call) -- the list, the barriers around it and the cross-workgroup sum compile away
template <int B, bool WIDE>
__global__ __launch_bounds__(B) void k_select_rescore(KArgs a) {
    __shared__ double s_part[3 * (B / 64)];
    __shared__ double s_tot[3];
    __shared__ float s_max[B / 64];
    frame_view(a);
    const int P = a.H * a.W;
    const Cam cam = make_cam(a);
    // speculative forward (spec_mode 1): the selection among the hypotheses the sampler's first pass settled -- a straggler has
    // no pose and no score yet; it keeps out of the maximum, the band and the statistics, and nothing of it is written but
    // exact_flag = 0 (k_spec_join completes the picture when the straggler chain is done)
    const bool spec = a.spec_mode == 1;
    auto fast_score = [&](int i) { return spec && a.spec_flag[i] ? -INFINITY : a.fast_scores[i]; };
    // max (NaN-ignoring)
    float m = -INFINITY;
    for (int i = threadIdx.x; i < a.N; i += B) m = fmaxf(m, fast_score(i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_max[0];
#pragma unroll
    for (int k = 1; k < B / 64; k++) m = fmaxf(m, s_max[k]);
    const float band = m - a.margin;

    // large grids: gridDim.z workgroups share a contender, workgroup z sums the cells of range z; the last one to arrive
    // adds the partial sums in range order (a fixed order: the score does not depend on who arrived when)
    const int nz = WIDE ? gridDim.z : 1, z = WIDE ? blockIdx.z : 0;
    const int per = ((P + nz - 1) / nz + B - 1) / B * B;  // cells per range, a multiple of the workgroup size
    const int c0 = z * per, c1 = min(P, c0 + per);
    __shared__ int s_last, s_nc;
    __shared__ int s_cont[WIDE ? B : 1];
    // this workgroup's share of the hypotheses (h = blockIdx.x + gridDim.x * j) is classified by all threads at once --
    // walking it one hypothesis at a time is a chain of dependent loads (64 of them at N = 16384: ~60 us per workgroup)
    const bool single = !WIDE || a.N <= (int)gridDim.x;  // one hypothesis per workgroup: no list, no barriers
    for (int j0 = 0; blockIdx.x + (long long)gridDim.x * j0 < a.N; j0 += B) {
        int nc;
        if (single) {
            const float fs = fast_score(blockIdx.x);  // workgroup-uniform
            nc = fs >= band ? 1 : 0;
            if (!nc && threadIdx.x == 0 && z == 0) {
                if (!(spec && a.spec_flag[blockIdx.x])) {
                    a.scores[blockIdx.x] = (double)fs;
                    if (a.scores_user) a.scores_user[user_slot(a, blockIdx.x)] = (double)fs;
                }
                a.exact_flag[blockIdx.x] = 0;
            }
        } else {
            if (threadIdx.x == 0) s_nc = 0;
            __syncthreads();
            const long long hh = blockIdx.x + (long long)gridDim.x * (j0 + threadIdx.x);
            if (hh < a.N) {
                const int h = (int)hh;
                const float fs = fast_score(h);
                if (fs >= band) {
                    s_cont[atomicAdd(&s_nc, 1)] = h;  // at most B entries per pass
                } else if (z == 0) {
                    if (!(spec && a.spec_flag[h])) {
                        a.scores[h] = (double)fs;
                        if (a.scores_user) a.scores_user[user_slot(a, h)] = (double)fs;
                    }
                    a.exact_flag[h] = 0;
                }
            }
            __syncthreads();
            nc = s_nc;
        }
        for (int ci = 0; ci < nc; ci++) {
            const int h = single ? (int)blockIdx.x : s_cont[ci];
            const int e = expert_of(a, h);
            const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
            const double* hp = a.hyps + (size_t)h * 6;
            const double t[3] = {hp[3], hp[4], hp[5]};
            double R[9];  // rodrigues_vec2mat(rvec) as the sampler stored it (the reference re-expands rvec, esac_util.h:302)
#pragma unroll
            for (int k = 0; k < 9; k++) R[k] = a.hyps_R[(size_t)h * 9 + k];
            double acc[1] = {0};
            for (int i = c0 + threadIdx.x; i < c1; i += B) {
                const int row = i / a.W, col = i - row * a.W;
                float err = project_exact_err(R, t, cam, mx[i], mx[P + i], mx[2 * P + i], cell_px(a, col), cell_py(a, row));
                err = err < a.max_reproj ? err : a.max_reproj;  // std::min(l, maxReproj), esac_util.h:358
                acc[0] += soft_inlier_exact(err, a.tau, a.beta);
            }
            block_sum<1, B>(acc, s_part, s_tot);
            if (nz > 1) {
                if (threadIdx.x == 0) {
                    __hip_atomic_store(a.sel_partials + (size_t)h * ESAC_SELECT_SPLIT + z, acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    const int arrived = atomicAdd(a.sel_arrived + h, 1);
                    s_last = arrived == nz - 1;
                    if (s_last) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        double sum = 0;
                        for (int k = 0; k < nz; k++)
                            sum += __hip_atomic_load(a.sel_partials + (size_t)h * ESAC_SELECT_SPLIT + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        acc[0] = sum;
                        a.sel_arrived[h] = 0;  // ready for the next call
                    }
                }
                __syncthreads();
                if (!s_last) continue;  // (workgroup-uniform)
            }
            if (threadIdx.x == 0) {
                const float scale = a.alpha / a.W / a.H;
                double sc = acc[0];
                sc *= scale;  // double *= float
                a.scores[h] = sc;
                if (a.scores_user) a.scores_user[user_slot(a, h)] = sc;
                a.exact_flag[h] = 1;
            }
            __syncthreads();
        }
        if (!single) __syncthreads();
    }
    if (blockIdx.x != 0 || z != 0) return;

    // softmax statistics (esac_util.h:461-497) from the fp32-path scores, in double; number of contenders
    double acc[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < a.N; i += B) {
        if (spec && a.spec_flag[i]) continue;
        const float s = a.fast_scores[i];
        const double d = (double)s - (double)m;
        const double ex = exp(d);
        acc[0] += ex;
        acc[1] += ex * d;
        acc[2] += (s >= band) ? 1.0 : 0.0;
    }
    block_sum<3, B>(acc, s_part, s_tot);
    if (threadIdx.x == 0) {
        a.n_contenders[0] = (int)acc[2];
        a.stats[0] = (double)m;  // max
        a.stats[1] = acc[0];     // sum exp(s - max)
        // entropy = -sum p log2 p,  p = exp(d)/S  ->  log2(S) - (sum exp(d) d) / (S ln 2)
        a.stats[2] = log2(acc[0]) - acc[1] / (acc[0] * 0.6931471805599453);
    }
    if (a.tstamps) {  // timing mode: span of the score kernel that just ran = max(end) - min(start)
        __shared__ long long s_lo[B / 64], s_hi[B / 64];
        long long lo = 0x7fffffffffffffffLL, hi = 0;
        for (int i = threadIdx.x; i < a.N; i += B) {
            const long long t0 = a.tstamps[2 * i], t1 = a.tstamps[2 * i + 1];
            lo = t0 < lo ? t0 : lo;
            hi = t1 > hi ? t1 : hi;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long ol = __shfl_xor(lo, o), oh = __shfl_xor(hi, o);
            lo = ol < lo ? ol : lo;
            hi = oh > hi ? oh : hi;
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[threadIdx.x >> 6] = lo;
            s_hi[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < B / 64; k++) {
                lo = s_lo[k] < lo ? s_lo[k] : lo;
                hi = s_hi[k] > hi ? s_hi[k] : hi;
            }
            a.span_acc[0] += hi - lo;
            a.span_acc[1] += 1;
        }
    }
}

// ================================================================= speculative forward: the join
// The launch stream has scored the hypotheses the sampler's first pass SETTLED and refined the best of them (fp32 ranking); on a
// second stream of the context's own the selection among the settled hypotheses (k_select_rescore, spec_mode 1) has run beside that
// refinement, and this kernel behind it; on the first one the straggler chain and the stragglers' fp32 scores.  This kernel (ONE
// workgroup, the arithmetic of k_select_rescore<1024, true> statement by statement, so that every number is the serial route's) is
// resident before refinement and chain are done, loads what the selection left, waits for their two "done" words, and completes the
// selection over ALL hypotheses:
//   * fp32 maximum and band over all of them; a settled hypothesis that was a contender of the narrower (settled-only) band but is
//     not one of the final band gets its fp32 score back (what the serial route leaves there); a straggler outside the band gets
//     its fp32 score, one inside is re-scored in reference arithmetic (esac_util.h:235-260) -- practically never: stragglers are
//     wrong-expert hypotheses;
//   * softMax / entropy statistics over all fp32 scores (esac_util.h:461-497), number of contenders;
//   * draw's argmax over the exact scores of the final band, first global index on ties (esac_util.h:512-529).
// It is the hypothesis that was refined (what else, with stragglers that score a hundredth of it -- unless the fp32 stream and
// the reference arithmetic order two near-equal scores differently): the record the refinement left in the workspace gets the
// exact score, the final probability / entropy / contender count and goes to the caller (device record, pinned host slot).
// Otherwise spec_state[0] = this call's epoch: the second refinement launch, enqueued with the call on the caller's stream and
// waiting for this kernel's "done" word (spec_state[7]), runs -- the workspace now holds exactly what k_select_rescore would have
// left: refine_pick_winner finds the true winner.
// Hand-off words instead of events: an event between two streams costs the waiting side 8-13 us on this
// platform even when it is long satisfied (scripts/dev/fork_join.hip, profiles/r06_*timeline*), a polled word ~1 us.
// The waits are bounded in wall time: a word that never comes (the other stream's launch failed) costs ESAC_SPEC_WAIT_TICKS, is
// counted in spec_state[5] and reported by the join (status 5) -- never a hang.
__global__ __launch_bounds__(64) void k_spec_wait(KArgs a, int which) {
    if (threadIdx.x == 0 && !spec_wait_word(a, which)) a.spec_state[5] += 1.0;  // (timed out: the chain runs late, the results stay right)
}
void launch_spec_wait(const KArgs& a, int which, hipStream_t s) { hipLaunchKernelGGL(k_spec_wait, dim3(1), dim3(64), 0, s, a, which); }

// "The join is done" (its verdict in spec_state[0], every output it patched written back): what the gated second refinement on
// the caller's stream waits for.  Called by the join's first wavefront behind its stores.
__device__ __forceinline__ void spec_join_done(const KArgs& a) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(a.spec_state + 7, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int B>
__global__ __launch_bounds__(B) void k_spec_join(KArgs a) {
    constexpr int K = ESAC_FIRST_WIDE_MAX / B;  // hypotheses per thread (sample_can_split: N <= ESAC_FIRST_WIDE_MAX)
    __shared__ double s_part[3 * (B / 64)];
    __shared__ double s_tot[3];
    __shared__ float s_max[B / 64];
    __shared__ int s_nc;
    __shared__ int s_cont[B];
    __shared__ double s_best[B / 64];
    __shared__ int s_besti[B / 64], s_bestg[B / 64];
    const int P = a.H * a.W;
    const Cam cam = make_cam(a);
    // this thread's hypotheses h = threadIdx.x + k B (the order in which k_select_rescore's threads walk them), loaded once: the
    // kernel is a handful of dependent passes over a few thousand values, and every pass that goes back to memory is a
    // round trip on the critical path of the call
    float fs[K];
    double sc0[K];  // (and the score each hypothesis holds now: the winner pick below then needs no second trip to memory)
    uint8_t strag[K], exact[K];
    __shared__ int s_chain_ok;
    // what the LAUNCH stream's own kernels left (they finished before this launch started): flags, the settled hypotheses' scores,
    // the speculative refinement's record and status word -- in flight while thread 0 waits for the other stream
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int h = (int)threadIdx.x + k * B;
        const bool in = h < a.N;
        sc0[k] = in ? a.scores[h] : 0.0;
        strag[k] = in ? a.spec_flag[h] : 0;
        exact[k] = in ? a.exact_flag[h] : 0;
    }
    if (threadIdx.x == 0) {
        s_nc = 0;
        // the straggler chain with the stragglers' scores and the speculative refinement are OTHER streams': wait for their "done" words
        s_chain_ok = spec_wait_word(a, 4, 1, 6);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // what the other streams' kernels wrote, not what this CU's caches hold
    const bool chain_ok = s_chain_ok != 0;
    // the speculative refinement's record and status word
    const double rec_pre = threadIdx.x < 32 ? a.result[threadIdx.x] : threadIdx.x == 33 ? a.spec_state[1] : 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int h = (int)threadIdx.x + k * B;
        fs[k] = h < a.N ? a.fast_scores[h] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; k++) m = fmaxf(m, fs[k]);  // (-inf for the slots beyond N: fmaxf ignores them like the loop bound does)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_max[0];
#pragma unroll
    for (int k = 1; k < B / 64; k++) m = fmaxf(m, s_max[k]);
    const float band = m - a.margin;
    // the final band: stragglers inside it are listed for the exact re-score (practically never any); what is outside it and
    // holds an exact score of the narrower band -- or is a straggler -- gets its fp32 score
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int h = (int)threadIdx.x + k * B;
        if (h >= a.N) continue;
        if (fs[k] >= band) {
            if (strag[k]) {
                const int pos = atomicAdd(&s_nc, 1);
                if (pos < B) s_cont[pos] = h;  // (more than B of them: the overflow pass below)
            }
        } else if (strag[k] || exact[k]) {
            a.scores[h] = (double)fs[k];
            if (a.scores_user) a.scores_user[user_slot(a, h)] = (double)fs[k];
            a.exact_flag[h] = 0;
            exact[k] = 0;
        }
    }
    __syncthreads();
    const int n_list = s_nc;
    for (int pass = 0; pass * B < n_list; pass++) {
        if (pass > 0) {  // beyond B stragglers inside the band: list the next B in thread order (never seen; kept for completeness)
            __syncthreads();
            if (threadIdx.x == 0) s_nc = 0;
            __syncthreads();
            for (int k = 0; k < K; k++) {
                const int h = (int)threadIdx.x + k * B;
                if (h < a.N && fs[k] >= band && strag[k] && !a.exact_flag[h]) {
                    const int pos = atomicAdd(&s_nc, 1);
                    if (pos < B) s_cont[pos] = h;
                }
            }
            __syncthreads();
        }
        const int nc = min(pass == 0 ? n_list : s_nc, B);
        for (int ci = 0; ci < nc; ci++) {
            // (the order of the list is whatever the atomics made it: every entry is re-scored by the whole workgroup, one after the other)
            const int hc = s_cont[ci];
            const int e = expert_of(a, hc);
            const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
            const double* hp = a.hyps + (size_t)hc * 6;
            const double t[3] = {hp[3], hp[4], hp[5]};
            double R[9];
#pragma unroll
            for (int k = 0; k < 9; k++) R[k] = a.hyps_R[(size_t)hc * 9 + k];
            double acc[1] = {0};
            for (int i = threadIdx.x; i < P; i += B) {
                const int row = i / a.W, col = i - row * a.W;
                float err = project_exact_err(R, t, cam, mx[i], mx[P + i], mx[2 * P + i], cell_px(a, col), cell_py(a, row));
                err = err < a.max_reproj ? err : a.max_reproj;  // std::min(l, maxReproj), esac_util.h:358
                acc[0] += soft_inlier_exact(err, a.tau, a.beta);
            }
            block_sum<1, B>(acc, s_part, s_tot);
            if (threadIdx.x == 0) {
                const float scale = a.alpha / a.W / a.H;
                double sc = acc[0];
                sc *= scale;  // double *= float
                a.scores[hc] = sc;
                if (a.scores_user) a.scores_user[user_slot(a, hc)] = sc;
                a.exact_flag[hc] = 1;
            }
            __syncthreads();
        }
    }
    if (n_list > 0) {  // (workgroup-uniform) the re-scored stragglers' flags, as this thread's registers hold them
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++)
            if (strag[k] && fs[k] >= band) exact[k] = 1;
    }
    // softmax statistics over ALL fp32-path scores (k_select_rescore's own loop and reduction)
    double acc[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < K; k++) {
        if ((int)threadIdx.x + k * B >= a.N) continue;
        const float s = fs[k];
        const double d = (double)s - (double)m;
        const double ex = exp(d);
        acc[0] += ex;
        acc[1] += ex * d;
        acc[2] += (s >= band) ? 1.0 : 0.0;
    }
    block_sum<3, B>(acc, s_part, s_tot);
    const double entropy = log2(acc[0]) - acc[1] / (acc[0] * 0.6931471805599453);
    if (threadIdx.x == 0) {
        a.n_contenders[0] = (int)acc[2];
        a.stats[0] = (double)m;
        a.stats[1] = acc[0];
        a.stats[2] = entropy;
    }
    // draw(probs, training=false): refine_pick_winner's rule over the final band
    double bs = -INFINITY;
    int bi = 0x7fffffff, bg = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int h = (int)threadIdx.x + k * B;
        if (h >= a.N || !exact[k]) continue;
        const int g = global_hyp(a, h);
        const double s = strag[k] ? a.scores[h] : sc0[k];  // (a straggler inside the band: re-scored by this kernel)
        if (s > bs || (s == bs && g < bg)) {
            bs = s;
            bi = h;
            bg = g;
        }
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
    if ((threadIdx.x & 63) == 0) {
        s_best[threadIdx.x >> 6] = bs;
        s_besti[threadIdx.x >> 6] = bi;
        s_bestg[threadIdx.x >> 6] = bg;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    bs = s_best[0];
    bi = s_besti[0];
    bg = s_bestg[0];
#pragma unroll
    for (int w = 1; w < B / 64; w++) {
        const double os = s_best[w];
        const int og = s_bestg[w];
        if (os > bs || (os == bs && og < bg)) {
            bs = os;
            bi = s_besti[w];
            bg = og;
        }
    }
    const int win = bi == 0x7fffffff ? 0 : bi;  // (no contender at all -- every score NaN: hypothesis 0, as refine_pick_winner)
    const double win_score = bi == 0x7fffffff ? a.scores[0] : bs;
    const int lane = threadIdx.x;
    const double status = !chain_ok ? 5.0 : __shfl(rec_pre, 33);
    if (!chain_ok) {  // the other stream never reported: nothing here can be trusted -- the host runs the call again, serially (status 5)
        if (a.result_pin) pin_deliver(a.result_pin, lane == 32 ? a.epoch : lane == 33 ? 5.0 : 0.0);
        if (a.result_user && lane == 31) a.result_user[31] = 3.0;  // (asynchronous callers: not a record -- esac_hip_pick_record: -12)
        if (lane == 0) {
            a.spec_state[0] = 0.0;
            a.spec_state[5] += 1.0;
        }
        spec_join_done(a);
        return;
    }
    // the speculative refinement's record is final when it refined THE winner (and ran to its end: a team that timed out is the
    // host's business, status 3, exactly as on the serial route)
    const double rec_hyp = __shfl(rec_pre, ESAC_RES_HYP_K), rec_score = __shfl(rec_pre, ESAC_RES_SCORE_K);
    // (the refinement started from the fp32 argmax of the settled hypotheses and knew no exact score -- the record gets it here)
    const bool held = rec_hyp == (double)global_hyp(a, win);
    if (!held && status != 3.0) {
        if (lane == 0) {
            a.spec_state[0] = a.epoch;
            a.spec_state[2] += 1.0;  // failed speculations on this context so far (ESAC_BUF_SPEC_INFO)
        }
        spec_join_done(a);
        return;
    }
    if (lane == 0) a.spec_state[0] = 0.0;
    double v = lane < 32 ? rec_pre : lane == 32 ? a.epoch : lane == 33 ? status : 0.0;
    if (lane == ESAC_RES_SCORE_K) v = win_score;
    if (lane == ESAC_RES_PROB_K) v = exp(win_score - (double)m) / acc[0];  // (held: win_score is the record's score)
    if (lane == ESAC_RES_ENTROPY_K) v = entropy;
    if (lane == ESAC_RES_CONTENDERS_K) v = (double)(int)acc[2];
    if (lane < 32) {
        a.result[lane] = v;
        if (a.result_user) a.result_user[lane] = lane == 31 ? (status == 3.0 ? 3.0 : 1.0) : v;  // ESAC_RES_VALID (refine_write_record)
    }
    if (a.result_pin) pin_deliver(a.result_pin, v);
    spec_join_done(a);
}
void launch_spec_join(const KArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_spec_join<1024>, dim3(1), dim3(1024), 0, s, a); }

// ================================================================= K3b: exact score of EVERY hypothesis
// (training path esac.cpp:295-316, esac_hip_score_exact)
template <int B>
__global__ __launch_bounds__(B) void k_rescore(KArgs a) {
    __shared__ double s_part[B / 64];
    __shared__ double s_tot[1];
    frame_view(a);
    const int n = a.N;
    const int P = a.H * a.W;
    const Cam cam = make_cam(a);
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        const int h = c;
        const int e = expert_of(a, h);
        const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
        const double* hp = a.hyps + (size_t)h * 6;
        const double t[3] = {hp[3], hp[4], hp[5]};
        double R[9];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = a.hyps_R[(size_t)h * 9 + k];
        double acc[1] = {0};
        for (int i = threadIdx.x; i < P; i += B) {
            const int row = i / a.W, col = i - row * a.W;
            float err = project_exact_err(R, t, cam, mx[i], mx[P + i], mx[2 * P + i], cell_px(a, col), cell_py(a, row));
            err = err < a.max_reproj ? err : a.max_reproj;  // std::min(l, maxReproj), esac_util.h:358
            acc[0] += soft_inlier_exact(err, a.tau, a.beta);
        }
        block_sum<1, B>(acc, s_part, s_tot);
        if (threadIdx.x == 0) {
            const float scale = a.alpha / a.W / a.H;
            double s = acc[0];
            s *= scale;  // double *= float
            a.scores[h] = s;
            if (a.scores_user) a.scores_user[user_slot(a, h)] = s;
            a.exact_flag[h] = 1;
        }
        __syncthreads();
    }
}

// softMax / entropy (esac_util.h:461-497) over the EXACT scores of all N hypotheses (ESAC_FLAG_EXACT_SCORES: k_rescore
// has scored every hypothesis in reference arithmetic, all of them are contenders): max, sum exp(s - max), entropy.
template <int B>
__global__ __launch_bounds__(B) void k_stats_exact(KArgs a) {
    __shared__ double s_part[2 * (B / 64)];
    __shared__ double s_tot[2];
    __shared__ double s_max[B / 64];
    frame_view(a);
    double m = -INFINITY;
    for (int i = threadIdx.x; i < a.N; i += B) m = fmax(m, a.scores[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_max[0];
#pragma unroll
    for (int k = 1; k < B / 64; k++) m = fmax(m, s_max[k]);
    double acc[2] = {0, 0};
    for (int i = threadIdx.x; i < a.N; i += B) {
        const double d = a.scores[i] - m;
        const double ex = exp(d);
        acc[0] += ex;
        acc[1] += ex * d;
    }
    block_sum<2, B>(acc, s_part, s_tot);
    if (threadIdx.x == 0) {
        a.n_contenders[0] = a.N;
        a.stats[0] = m;
        a.stats[1] = acc[0];
        a.stats[2] = log2(acc[0]) - acc[1] / (acc[0] * 0.6931471805599453);  // -sum p log2 p, p = exp(d) / S
    }
}

// Multi-GPU: the all-reduced exchange buffer ends with one 32-double record per rank (zero where a rank had no
// hypothesis; ESAC_RES_VALID = slot 31 marks a real one).  Global winner = highest exact score, lowest GLOBAL hypothesis
// index on ties (esac_util.h:519 "first max") -- picked here and handed to the host through pinned memory, like the
// single-GPU record.  One wavefront.
// zero / n_zero: optional -- the OTHER exchange buffer of the caller's pair, cleared here for the next call (the caller
// alternates between two buffers, so no call starts with a memset of its own: esac_amd/distributed.py).
__global__ __launch_bounds__(256) void k_pick_record(const double* __restrict__ records, int world, double* __restrict__ pin, double epoch,
                                                     double* __restrict__ zero, int n_zero) {
    for (int i = threadIdx.x; i < n_zero; i += 256) zero[i] = 0.0;
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double bs = -INFINITY, bh = INFINITY;
    int br = -1;
    bool failed = false;  // a rank's refinement team timed out (ESAC_RES_VALID = 3): no winner may be declared without its candidate
    for (int r = lane; r < world; r += 64) {
        const double* rec = records + (size_t)r * 32;
        failed |= rec[31] == 3.0;
        if (rec[31] != 1.0) continue;
        const double s = rec[0], h = rec[1];
        if (br < 0 || s > bs || (s == bs && h < bh)) {
            bs = s;
            bh = h;
            br = r;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double os = __shfl_xor(bs, o), oh = __shfl_xor(bh, o);
        const int orr = __shfl_xor(br, o);
        if (orr >= 0 && (br < 0 || os > bs || (os == bs && oh < bh))) {
            bs = os;
            bh = oh;
            br = orr;
        }
    }
    failed = __any(failed);
    if (lane < 32) pin[lane] = br >= 0 ? records[(size_t)br * 32 + lane] : 0.0;
    __threadfence_system();
    if (lane == 0) {
        pin[33] = failed ? 3.0 : br >= 0 ? 0.0 : 2.0;  // 3: a rank's team timed out; 2: no rank contributed a record
        __threadfence_system();
        *reinterpret_cast<volatile double*>(pin + 32) = epoch;
    }
}
void launch_pick_record(const double* records, int world, double* pin, double epoch, double* zero, int n_zero, hipStream_t s) {
    hipLaunchKernelGGL(k_pick_record, dim3(1), dim3(256), 0, s, records, world, pin, epoch, zero, n_zero);
}

// ================================================================= multi-GPU: load-balanced shard of the hypotheses
// esac_hip_shard_balanced (include/esac_hip.h).  Order the hypotheses by (expert, index) -- the stable counting sort of the
// assignment vector -- and give rank r the sorted positions [lo, hi) = its N / world share: every rank gets the same
// number of hypotheses (+-1) whatever the gating distribution, a rank's hypotheses belong to a contiguous range of
// experts, and only the first and last expert of that range can be shared with a neighbour.  The plan is a pure function
// of the assignment vector: every rank runs this kernel on its own GPU and they agree without talking.
// One workgroup.  (1) histogram in LDS, exclusive scan -> first sorted position of every expert; (2) the two experts
// that straddle lo / hi need the EXACT within-expert rank of their hypotheses (which of them fall on this side of the
// cut): every thread counts them over its contiguous slice of indices, a block scan turns the counts into ranks;
// (3) emit: hypotheses of interior experts are appended through an LDS fill counter per expert (their order inside the
// shard is irrelevant: RNG streams, scores and tie-breaks all go by GLOBAL index), hypotheses of the two cut experts go to
// the slot their exact rank gives them.  Values outside [0, E) count as expert 0 (device_common.hpp:expert_of) and are
// copied unchanged, so the forward call on the shard still reports them.
constexpr int SHARD_B = 1024;
__global__ __launch_bounds__(SHARD_B) void k_shard_balanced(const int64_t* __restrict__ assign, int N, int E, int world, int rank,
                                                            int expert_base, int32_t* __restrict__ index_out,
                                                            int64_t* __restrict__ assign_out, int32_t* __restrict__ info_out) {
    __shared__ int s_cnt[ESAC_TILED_MAX_EXPERTS];    // histogram, then fill level
    __shared__ int s_start[ESAC_TILED_MAX_EXPERTS];  // first sorted position of each expert
    __shared__ int s_scan[2][SHARD_B];
    __shared__ int s_cut[2];  // the expert that contains sorted position lo, the one that contains hi - 1
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int base = N / world, rem = N % world;
    const int lo = rank * base + (rank < rem ? rank : rem), hi = lo + base + (rank < rem ? 1 : 0);
    auto expert = [&](int h) {
        const long long e = assign[h];
        return (unsigned long long)e < (unsigned long long)E ? (int)e : 0;
    };
    for (int e = tid; e < E; e += SHARD_B) s_cnt[e] = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int h = tid; h < N; h += SHARD_B) {
        const long long e = assign[h];
        if ((unsigned long long)e >= (unsigned long long)E) s_bad = 1;
        atomicAdd(&s_cnt[expert(h)], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: every thread owns EPT consecutive experts
    constexpr int EPT = ESAC_TILED_MAX_EXPERTS / SHARD_B;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = tid * EPT + k;
        mine += e < E ? s_cnt[e] : 0;
    }
    auto block_excl = [&](int v) {  // exclusive prefix of v over the workgroup (Hillis-Steele in LDS)
        int cur = 0;
        s_scan[0][tid] = v;
        __syncthreads();
        for (int off = 1; off < SHARD_B; off <<= 1) {
            const int t = s_scan[cur][tid] + (tid >= off ? s_scan[cur][tid - off] : 0);
            s_scan[cur ^ 1][tid] = t;
            cur ^= 1;
            __syncthreads();
        }
        const int r = s_scan[cur][tid] - v;
        __syncthreads();
        return r;
    };
    {
        int pos = block_excl(mine);
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int e = tid * EPT + k;
            if (e < E) {
                const int c = s_cnt[e];
                s_start[e] = pos;
                if (c > 0 && pos <= lo && lo < pos + c) s_cut[0] = e;
                if (c > 0 && pos < hi && hi <= pos + c) s_cut[1] = e;
                pos += c;
            }
        }
    }
    __syncthreads();
    if (hi <= lo) {  // more ranks than hypotheses: nothing for this one
        if (info_out && tid == 0) { info_out[0] = 0; info_out[1] = -1; info_out[2] = 0; info_out[3] = s_bad; }
        return;
    }
    const int e_lo = s_cut[0], e_hi = s_cut[1];
    // exact within-expert ranks for the two cut experts: contiguous index slice per thread
    const int per = (N + SHARD_B - 1) / SHARD_B;
    const int h0 = tid * per, h1 = min(N, h0 + per);
    int c_lo = 0, c_hi = 0;
    for (int h = h0; h < h1; h++) {
        const int e = expert(h);
        c_lo += e == e_lo;
        c_hi += e == e_hi;
    }
    int r_lo = block_excl(c_lo), r_hi = e_hi == e_lo ? 0 : block_excl(c_hi);
    if (e_hi == e_lo) r_hi = r_lo;
    for (int e = tid; e < E; e += SHARD_B) s_cnt[e] = 0;  // fill level of the interior experts
    __syncthreads();
    for (int h = h0; h < h1; h++) {
        const long long raw = assign[h];
        const int e = (unsigned long long)raw < (unsigned long long)E ? (int)raw : 0;
        int pos;  // sorted position of h (exact for the cut experts, any free slot of its expert otherwise)
        if (e == e_lo) pos = s_start[e] + r_lo++;
        else if (e == e_hi) pos = s_start[e] + r_hi++;
        else if (e > e_lo && e < e_hi) pos = s_start[e] + atomicAdd(&s_cnt[e], 1);
        else continue;
        if (e == e_lo && e == e_hi) r_hi = r_lo;
        if (pos < lo || pos >= hi) continue;
        index_out[pos - lo] = h;
        assign_out[pos - lo] = (unsigned long long)raw < (unsigned long long)E ? (long long)(e - expert_base) : raw;
    }
    if (info_out && tid == 0) { info_out[0] = e_lo; info_out[1] = e_hi; info_out[2] = hi - lo; info_out[3] = s_bad; }
}
void launch_shard_balanced(const int64_t* assign, int N, int E, int world, int rank, int expert_base, int32_t* index_out,
                           int64_t* assign_out, int32_t* info_out, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_balanced, dim3(1), dim3(SHARD_B), 0, s, assign, N, E, world, rank, expert_base, index_out, assign_out, info_out);
}

// ---------------------------------------------------------------- launchers
void launch_stats_exact(const KArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_stats_exact<256>, dim3(1, a.frames), dim3(256), 0, s, a);
}
// the chain that finishes hypotheses left SAMPLE_PENDING at try b.first_try (see k_sample_prescreen); `waves` wavefronts
// in all work through the list of pending hypotheses
static void launch_sample_stragglers(const KArgs& b, int waves, hipStream_t s) {
    const int gx = b.N < waves ? b.N : waves;  // (the shape only spreads the linear wavefront index over three dimensions)
    const int gz = (waves + gx * b.frames - 1) / (gx * b.frames);
    hipLaunchKernelGGL(k_sample_prescreen, dim3(gx, b.frames, gz < 1 ? 1 : gz), dim3(64), 0, s, b);
    hipLaunchKernelGGL(k_sample_decide, dim3((b.samp_cap + 63) / 64), dim3(64), 0, s, b);
    if ((long long)b.N * b.frames >= 8192) hipLaunchKernelGGL(k_sample_commit, dim3((b.N + 255) / 256, b.frames), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_sample_screened<true>, dim3(b.N, b.frames), dim3(64), 0, s, b);
}

void launch_sample(const KArgs& a, hipStream_t s) {
    const long long total = (long long)a.N * a.frames;
    if (a.sc4) hipLaunchKernelGGL(k_pack_cells, dim3(2048), dim3(256), 0, s, a);
    // entries of the "maybe" list, hypotheses of the pending list (+ the per-expert counters behind them, see expert_stats)
    // are zero between calls: the last kernel of the screened chain clears them (k_sample_screened<true>).  A fill in front
    // of every sampling launch cost the headline call, which never appends to them, 5 us (0.2058 -> 0.2010 ms).
    KArgs b = a;
    b.handover = 0x7fffffff;
    // Few hypotheses in flight: latency.  A workgroup per hypothesis (the candidates of a try on two or four lanes at first)
    // settles a hypothesis of the right expert within its first round; with several experts the stragglers are handed to
    // the spread, screened search after `handover` tries (every wavefront of that launch works, rounds handed out in order).  Beyond ~10^3
    // hypotheses (several experts) a workgroup per hypothesis no longer fits the chip in one wave of workgroups: the
    // first 32 tries run four hypotheses per wavefront and the screened chain finishes the rest.
    // ESAC_FLAG_EXACT_SAMPLING: no screen anywhere -- every try is solved and decided by the fp64 route (k_sample walks a
    // straggler's whole budget itself, one try per lane; the throughput shape finishes with k_sample<64> instead of the
    // screened chain)
    const bool exact = (a.flags & ESAC_FLAG_EXACT_SAMPLING_K) != 0;
    const bool handover = a.E > 1 && a.max_tries > 1024 && !exact;
constexpr int ESAC_LATENCY_MAX = 1024;
constexpr int ESAC_HANDOVER = 32;  // (64: k_sample<128> 41 us + screened search 29 us at config 3; 32: 30 + 31 us)
    // wavefronts of the screened chain: every one of them works whatever the number of pending hypotheses is (they take the
    // 64-try rounds of the hypotheses on the list in order), so the launch is sized for the chip -- 2048 wavefronts are
    // resident at two per SIMD -- with some slack for the tail; when (nearly) every hypothesis is pending, as in the
    // 50-expert workloads, eight per hypothesis measured best (A/B on one box, config 5a: 4 / 8 / 32 per hypothesis ->
    // 2.00 / 1.95 / 2.18 ms)
constexpr int ESAC_CHAIN_PER_HYP = 8;
    const long long w8 = (a.E == 1 ? 1LL : (long long)ESAC_CHAIN_PER_HYP) * total;
    // (every pending hypothesis needs at least ONE wavefront: wavefront L serves list entry L % count, so a launch smaller
    // than the list would leave its tail unscreened -- beyond 131072 hypotheses the launch grows with them)
    const long long wcap = total > 131072 ? total : 131072;
    const int waves = (int)(w8 < ESAC_CHAIN_WAVES ? ESAC_CHAIN_WAVES : (w8 > wcap ? wcap : w8));
    if (total <= (handover ? ESAC_LATENCY_MAX : 1024)) {
        if (handover) b.handover = ESAC_HANDOVER;
        // up to 256 hypotheses: four wavefronts each, two lanes per try (128 tries per round, one workgroup per CU at this
        // kernel's ~445 registers: the chip is full).  Beyond that the workgroups queue up behind each other (1024
        // hypotheses: four ~12 us rounds back to back, 51 us measured): two wavefronts per hypothesis, four lanes per try
        // (32 tries per round -- 93 % of the hypotheses of a usable map are settled in it) put two hypotheses on a CU at a
        // time.
        // 513 .. 1024 hypotheses that hand their stragglers over (round 6): ONE wavefront per hypothesis, two lanes per try -- the
        // same 32 tries in one round, a chain of two candidates instead of one, and all 1024 wavefronts resident at once instead of
        // 2048 in two waves of workgroups: 29.9 -> 25.4 us at config 3 (four lanes per try at one wavefront, two rounds of 16
        // tries: 29.9 again).  Without a hand-over (one expert, ESAC_FLAG_EXACT_SAMPLING) a straggler walks its whole budget in this
        // kernel, one try per lane: two wavefronts per hypothesis halve that tail (config 3 on the guaranteed routes: 0.56 ms
        // against 0.89 with one)
        if (total <= 256)                    hipLaunchKernelGGL((k_sample<256, 2>), dim3(a.N, a.frames), dim3(256), 0, s, b);
        else if (total <= 512 || !handover)  hipLaunchKernelGGL((k_sample<128, 4>), dim3(a.N, a.frames), dim3(128), 0, s, b);
        else                                 hipLaunchKernelGGL((k_sample<64, 2>), dim3(a.N, a.frames), dim3(64), 0, s, b);
    } else if (total <= 4096 && !handover) {
        hipLaunchKernelGGL((k_sample<128, 1>), dim3(a.N, a.frames), dim3(128), 0, s, b);
    } else {  // throughput: passes of 16 tries with four hypotheses per wavefront, then the unaccepted rest by the screened chain
        if (total <= ESAC_FIRST_WIDE_MAX) {
            hipLaunchKernelGGL(k_sample_first<32>, dim3((a.N + 1) / 2, a.frames), dim3(64), 0, s, b);
            b.first_try += 32;
        } else if (a.E > 1 && !exact) {
            // tens of thousands of hypotheses over many experts (config 5: 16384 over 50, Dirichlet gating): nearly all of them
            // sit on wrong experts, where 32 tries in full fp64 are 32 solves for nothing -- the screened chain takes them
            // from try 0 (a hypothesis of the right expert costs it one screened round and a handful of fp64 decisions)
        } else {
            for (int pass = 0; pass < FIRST_PHASE_TRIES / 16 && b.first_try < a.max_tries; pass++) {
                hipLaunchKernelGGL(k_sample_first<16>, dim3((a.N + 3) / 4, a.frames), dim3(64), 0, s, b);
                b.first_try += 16;
            }
        }
        if (b.first_try < a.max_tries) {
            if (exact) {
                hipLaunchKernelGGL((k_sample<64, 1>), dim3(a.N, a.frames), dim3(64), 0, s, b);  // every try solved in full
            } else {
                hipLaunchKernelGGL(k_pending_list, dim3((a.N + 1023) / 1024, a.frames), dim3(1024), 0, s, b);
                // (the list stays in hypothesis order: expert-major and dealt to the XCDs, the full-resolution workload's
                // gathers hit L2 at 0.59 instead of 0.07 and fetch 4.8 GB instead of 11.1 GB per call -- and the kernel takes
                // the same 1.7 ms: it does not wait for them.  profiles/r04_cfg5b_pending_order.txt, LAB_NOTES.md)
                launch_sample_stragglers(b, waves, s);
            }
        }
        return;
    }
    if (handover) {
        b.first_try = b.handover;
        launch_sample_stragglers(b, waves, s);
    }
}
// ---- speculative forward: the sampler in two parts (see KArgs::spec_mode).  The shapes that have a FIRST PASS which settles
// most hypotheses and a straggler chain behind it: several experts, a single frame, the screened route, at most
// ESAC_FIRST_WIDE_MAX hypotheses (beyond that the chain takes every hypothesis from try 0: nothing is settled early).
constexpr int ESAC_SPLIT_LATENCY_MAX = 1024, ESAC_SPLIT_HANDOVER = 32;  // = launch_sample's ESAC_LATENCY_MAX / ESAC_HANDOVER
bool sample_can_split(const KArgs& a) {
    return a.frames == 1 && a.E > 1 && a.max_tries > 1024 && !(a.flags & ESAC_FLAG_EXACT_SAMPLING_K) && a.N <= ESAC_FIRST_WIDE_MAX &&
           a.first_try == 0;
}
// The first pass on `s` -- its last kernel completes `fork` (the kernel's own completion signal: hipExtLaunchKernelGGL; an event
// recorded behind it costs the launch stream 5 us, scripts/dev/fork_join.hip) -- and the chain on `side` behind that event.
// Exactly the kernels, arguments and order of launch_sample for these shapes.  Returns 0, or the hipError_t of the event wait.
// The first pass on `s`; the chain's arguments (where it takes over, how many wavefronts) come back in `chain`.
// Exactly the kernels, arguments and order of launch_sample for these shapes.
int launch_sample_split(const KArgs& a, hipStream_t s, KArgs* chain, int* chain_waves) {
    const long long total = a.N;
    if (a.sc4) hipLaunchKernelGGL(k_pack_cells, dim3(2048), dim3(256), 0, s, a);
    KArgs b = a;
    b.handover = 0x7fffffff;
    const long long w8 = (long long)8 * total;  // ESAC_CHAIN_PER_HYP (launch_sample)
    *chain_waves = (int)(w8 < ESAC_CHAIN_WAVES ? ESAC_CHAIN_WAVES : (w8 > 131072 ? 131072 : w8));
    if (total <= ESAC_SPLIT_LATENCY_MAX) {
        b.handover = ESAC_SPLIT_HANDOVER;
        if (total <= 256)      hipLaunchKernelGGL((k_sample<256, 2>), dim3(a.N, 1), dim3(256), 0, s, b);
        else if (total <= 512) hipLaunchKernelGGL((k_sample<128, 4>), dim3(a.N, 1), dim3(128), 0, s, b);
        else                   hipLaunchKernelGGL((k_sample<64, 2>), dim3(a.N, 1), dim3(64), 0, s, b);
        b.first_try = b.handover;
    } else {
        hipLaunchKernelGGL(k_sample_first<32>, dim3((a.N + 1) / 2, 1), dim3(64), 0, s, b);
        b.first_try += 32;
        hipLaunchKernelGGL(k_pending_list, dim3((a.N + 1023) / 1024, 1), dim3(1024), 0, s, b);
    }
    *chain = b;
    return 0;
}
void launch_sample_stragglers_on(const KArgs& chain, int waves, hipStream_t side) { launch_sample_stragglers(chain, waves, side); }
// the side stream's last kernel (see k_score_stragglers): behind the chain
void launch_score_stragglers(const KArgs& a, hipStream_t side) {
    KArgs b = a;
    b.tstamps = nullptr;
    b.spec_mode = 0;
    const int grid = a.N < 1024 ? a.N : 1024;
    if ((long long)a.N <= 2048) hipLaunchKernelGGL(k_score_stragglers<512>, dim3(grid), dim3(512), 0, side, b);
    else                        hipLaunchKernelGGL(k_score_stragglers<256>, dim3(grid), dim3(256), 0, side, b);
}

void launch_hyps_to_rt32(const KArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_hyps_to_rt32, dim3((a.N + 255) / 256), dim3(256), 0, s, a);
}
void launch_score_fast(const KArgs& a, hipStream_t s) {
    // few hypotheses in flight (a single frame): the launch is ramp + load latency, 8 wavefronts per hypothesis hide
    // more of it (device span 2.96 -> 2.57 us at 256 hypotheses); many hypotheses: 4 wavefronts stream best
    if ((long long)a.N * a.frames <= 2048) hipLaunchKernelGGL(k_score_fast<512>, dim3(a.N, a.frames), dim3(512), 0, s, a);
    else               hipLaunchKernelGGL(k_score_fast<256>, dim3(a.N, a.frames), dim3(256), 0, s, a);
}
void launch_score(const KArgs& a, hipStream_t s) {
    if (a.partials) launch_score_tiled(a, s);
    else            launch_score_fast(a, s);
}
void launch_select_rescore(const KArgs& a, hipStream_t s) {
    // few contenders, latency matters: 16 wavefronts per workgroup; a single frame spreads its hypotheses over up to
    // 256 workgroups (a contender gets a CU to itself), batched frames over 16 each (the frames fill the chip)
    // a contender on a 480x640 grid is 0.15 ms of one workgroup in reference arithmetic: ESAC_SELECT_SPLIT workgroups share
    // it there (and the hypotheses are spread over that many fewer columns: every workgroup finds the fp32 maximum itself)
    const int split = (long long)a.H * a.W >= 32768 ? ESAC_SELECT_SPLIT : 1;
    const int cap = (a.frames > 1 ? 16 : 256) / split;
    const int grid = a.N < cap ? a.N : (cap < 1 ? 1 : cap);
constexpr int ESAC_SELECT_B = 1024;  // threads of the single-frame variant (A/B: scripts/dev/variants.sh)
    if (split == 1 && a.N <= grid) hipLaunchKernelGGL((k_select_rescore<ESAC_SELECT_B, false>), dim3(grid, a.frames), dim3(ESAC_SELECT_B), 0, s, a);
    else                           hipLaunchKernelGGL((k_select_rescore<1024, true>), dim3(grid, a.frames, split), dim3(1024), 0, s, a);
}
void launch_rescore_all(const KArgs& a, hipStream_t s) {
    const int grid = a.N < 4096 ? a.N : 4096;  // bulk exact scoring: 4 wavefronts per hypothesis are enough
    // few hypotheses (a single frame of <= 256: the guaranteed route of the headline call): a cell is ~120 DEPENDENT fp64
    // operations (IEEE division, sqrt, exp), and with one wavefront per SIMD every one of them waits out its 10 cycles --
    // 16 wavefronts per hypothesis put four on every SIMD, whose chains interleave (9.8 -> 5.x us at 256 hypotheses)
    if ((long long)a.N * a.frames <= 256) hipLaunchKernelGGL(k_rescore<1024>, dim3(grid, a.frames), dim3(1024), 0, s, a);
    else                                  hipLaunchKernelGGL(k_rescore<256>, dim3(grid, a.frames), dim3(256), 0, s, a);
}

}  // namespace esac
