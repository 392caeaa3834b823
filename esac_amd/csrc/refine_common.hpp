// refine_common.hpp -- what the two refinement translation units share (esac_refine.hip: one workgroup per refinement /
// cooperating workgroups on grids beyond one LDS list; esac_refine_team.hip: a team of workgroups on one XCD for the
// small grids): the rare pseudo-inverse LM step, cycle-counter macros, the exchange between workgroups that share one
// refinement, winner pick and result record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"
#include "esac_kernels.hpp"
#include "bwd_math.hpp"
#include "lm_math.hpp"
#include "pose_math.hpp"

namespace esac {

// The rare branch of an LM step: the damped normal matrix is singular to rounding (lm_solve6 returned false), so the
// step is pinv(A) * g with eigenvalues below 2 eps sum|w| dropped -- cv::solve(DECOMP_SVD) inside CvLevMarq, the
// route the CPU library always takes (bwd_math.hpp:pinv_sym6_jacobi is the same algorithm, unrolled into registers).
// Here it must cost the common path nothing: ONE lane runs rolled loops over matrices in LDS (run-time indices, a few
// hundred bytes of code, no extra registers), the others wait.  Every lane reaches this together (the LM state is
// replicated), so the barriers are uniform.  `lds`: >= 84 doubles of scratch nobody else touches meanwhile.
__device__ __forceinline__ void lm_solve6_pinv(const double (&U21)[21], const double (&g)[6], double lambda, double (&dx)[6], double* lds) {
    double* A = lds;        // [6][6]
    double* V = lds + 36;   // [6][6]
    double* out = lds + 72; // [6]
    double* gs = lds + 78;  // [6]
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) {
                const double v = (i == j) ? U21[k] * (1. + lambda) : U21[k];
                A[i * 6 + j] = v;
                A[j * 6 + i] = v;
                k++;
            }
#pragma unroll
        for (int i = 0; i < 6; i++) gs[i] = g[i];
        for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 60; sweep++) {
            double off = 0;
            for (int i = 0; i < 6; i++)
                for (int j = i + 1; j < 6; j++) off += A[i * 6 + j] * A[i * 6 + j];
            if (off == 0) break;
            for (int p = 0; p < 6; p++)
                for (int q = p + 1; q < 6; q++) {
                    const double apq = A[p * 6 + q];
                    const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                    if (!(fabs(theta) <= 1.7976931348623157e308)) t = 0;
                    if (apq == 0) t = 0;
                    const double c = 1 / sqrt(t * t + 1), sn = t * c;
                    for (int m = 0; m < 6; m++) {
                        const double akp = A[m * 6 + p], akq = A[m * 6 + q];
                        A[m * 6 + p] = c * akp - sn * akq;
                        A[m * 6 + q] = sn * akp + c * akq;
                    }
                    for (int m = 0; m < 6; m++) {
                        const double apk = A[p * 6 + m], aqk = A[q * 6 + m];
                        A[p * 6 + m] = c * apk - sn * aqk;
                        A[q * 6 + m] = sn * apk + c * aqk;
                    }
                    for (int m = 0; m < 6; m++) {
                        const double vkp = V[m * 6 + p], vkq = V[m * 6 + q];
                        V[m * 6 + p] = c * vkp - sn * vkq;
                        V[m * 6 + q] = sn * vkp + c * vkq;
                    }
                }
        }
        double thresh = 0;
        for (int i = 0; i < 6; i++) thresh += fabs(A[i * 7]);
        thresh *= 2 * 2.220446049250313e-16;
        for (int i = 0; i < 6; i++) out[i] = 0;
        for (int m = 0; m < 6; m++) {
            const double w = A[m * 7];
            if (!(fabs(w) > thresh)) continue;
            double proj = 0;
            for (int j = 0; j < 6; j++) proj += V[j * 6 + m] * gs[j];
            proj /= w;
            for (int i = 0; i < 6; i++) out[i] += V[i * 6 + m] * proj;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; i++) dx[i] = out[i];
    __syncthreads();
}

constexpr int REFINE_B = ESAC_REFINE_THREADS;  // 4 wavefronts = one per SIMD of the one CU this kernel occupies
constexpr int LDS_CAP = ESAC_REFINE_LDS_CAP;   // correspondences staged in LDS (128 KiB of the CU's 160 KiB)
constexpr int ERR_UNROLL = ESAC_ERR_UNROLL;    // points per lane in flight in the exact error pass
constexpr int LM_NP = 2;              // correspondences per lane in flight in an LM pass

// Section cycle counters (clock64 = shader clock), enabled with -DESAC_PROFILE_CYCLES; index:
// 0 total, 1 argmax, 2 error image + compaction, 3 pose2trans + result record, 4 rodrigues+chain, 5 point loop, 6 block_sum,
// 7 transform, 8 solve, 9 number of passes, 10-15 error-pass sub-sections, 16 LM accept / reject / termination logic,
// 17 rejected LM trials, 18 pseudo-inverse steps (team kernel: counts, not cycles)
#ifdef ESAC_PROFILE_CYCLES
#define CYC_DECL long long cyc_t0_
#define CYC_BEGIN() cyc_t0_ = clock64()
#define CYC_END(idx) g_cyc[idx] += clock64() - cyc_t0_
#define CYC_ADD(idx, v) g_cyc[idx] += (v)
// register-only arithmetic may be scheduled across the clock reads: pinning a section's inputs after its first read and its
// outputs before its second one keeps the section's work between them
#define CYC_PIN(arr, n)                                                  \
    do {                                                                 \
        _Pragma("unroll") for (int pin_k = 0; pin_k < (n); pin_k++) asm volatile("" : "+v"((arr)[pin_k])); \
    } while (0)
#else
#define CYC_DECL
#define CYC_BEGIN()
#define CYC_END(idx)
#define CYC_ADD(idx, v)
#define CYC_PIN(arr, n)
#endif

struct __attribute__((aligned(16))) Corr {
    float x, y, z;
    uint32_t row_col;  // grid cell of the correspondence: row << 16 | col (H, W <= 65535, checked by the C ABI); its pixel
                       // position is col * sub + sub / 2 - shift_x (createSampling, esac_util.h:64-66), any magnitude
};

__device__ __forceinline__ int cell_pxi(const KArgs& a, int col) { return col * a.sub + a.sub / 2 - a.shift_x; }
__device__ __forceinline__ int cell_pyi(const KArgs& a, int row) { return row * a.sub + a.sub / 2 - a.shift_y; }

// 10^k, |k| <= 31, by binary exponentiation (the CPU library evaluates exp(k*log(10)))
__device__ __forceinline__ double pow10_int(int k) {
    double r = 1.0;
    const int n = k < 0 ? -k : k;
    double b = 10.0;
#pragma unroll
    for (int bit = 0; bit < 5; bit++) {
        if (n & (1 << bit)) r *= b;
        b *= b;
    }
    return k < 0 ? 1.0 / r : r;
}

// CvLevMarq's CHECK_ERR: is the trial's error norm larger than the one at the last accepted point (then the trial is rejected
// and repeated with 10 x lambda)?  sqrt(err2) > sqrt(prev2), decided without the square roots unless the two are a few ulp
// apart (a double-precision square root is ~40 dependent instructions on the serial section of every pass; sqrt is monotone
// and correctly rounded: err2 <= prev2 (or a NaN) can never give a larger root, and a relative gap of 8 eps separates the
// roots by more than their rounding).
// A trial that ENDS its re-fit if accepted (iteration 20, or a relative step below FLT_EPSILON: the re-fit has converged) is
// treated differently inside that band.  There the two error norms agree to a few ulp and `>` is decided by the last bit of two sums
// of thousands of terms -- a coin the reference's arithmetic and this one's do not share.  A rejection sends CvLevMarq up
// the lambda ladder (x10 a round) until the step has shrunk below the noise: up to 14 more rounds (35 us of the headline
// call, measured: scripts/dev/call8_probe.py) that end at a point within the size of that last step -- ~1e-9 rad / m -- of
// the trial, with the same iteration count (rejected trials do not count) and the re-fit over either way.  Inside the band
// (err <= prev (1 + 8 eps)) such a trial is therefore ACCEPTED; a clear increase is rejected as ever, and trials that do
// not end their re-fit keep the reference's comparison (an acceptance there would change lambda's course).
__device__ __forceinline__ bool trial_rejected(double err2, double prev2, bool ends_refit) {
    if (!(err2 > prev2)) return false;
    if (err2 > prev2 * (1.0 + 8.0 * DBL_EPSILON)) return true;
    if (ends_refit) return false;
    asm volatile("; norms a few ulp apart");  // (keeps the two square roots BEHIND the branch: as plain arithmetic they are
                                              // if-converted and run on every pass, ~45 instructions)
    return sqrt(err2) > sqrt(prev2);
}
// cvNorm(param, prevParam, CV_RELATIVE_L2) < FLT_EPSILON, i.e. sqrt(dn) / (sqrt(pn) + DBL_EPSILON) < eps, dn = |param - prev|^2,
// pn = |prev|^2: decided on the squares wherever the answer is clear of the threshold by 1e-9 (relative), by the
// reference's own expression in between.  sqrt(pn) <= max(1, pn) bounds the DBL_EPSILON term from above.
__device__ __forceinline__ bool relative_step_below_eps(double dn, double pn) {
    const double e2 = (double)FLT_EPSILON * (double)FLT_EPSILON;
    const double lo = e2 * pn;
    if (dn < lo * (1.0 - 1e-9)) return true;
    const double hi = lo + e2 * DBL_EPSILON * (2.0 * (pn > 1.0 ? pn : 1.0) + DBL_EPSILON);
    if (dn > hi * (1.0 + 1e-9)) return false;
    return sqrt(dn) / (sqrt(pn) + DBL_EPSILON) < (double)FLT_EPSILON;
}

// ---- cooperating workgroups ------------------------------------------------------------------------------------------------
// One refinement can be shared by G workgroups: workgroup g owns a slice of the cells -- its part of every error pass and,
// in its own LDS, the correspondences found there -- and every reduction (inlier count, the 24 moments of an LM pass)
// becomes: workgroup sum -> exchange -> every workgroup adds the G contributions in the same fixed order.  All workgroups
// then hold bitwise identical sums, take the same LM / stopping decisions and carry the same pose: nothing is ever
// broadcast.  Two exchanges exist:
//
// REFINE_COOP (grids beyond the LDS list, 480x640: a pass is ~700 us of one CU, up to 256 workgroups anywhere on the chip):
// partial[g] by device-scope (sc1, write-through) stores, drained; one lane arrives at a monotonic counter and polls it
// with s_sleep; the partials are read back with device-scope loads (~3 us a round, noise against the pass).
//
// REFINE_TEAM (round 4; the 60x80 grid of the headline call, where a pass is 2-6 us and that barrier costs more than it
// saves): up to 8 workgroups exchange TAGGED GRANULES in ONE hop.  A granule is 16 bytes {double v, u64 tag},
// tag = (launch epoch << 20 | exchange number) ^ bits(v), written by one 16-byte sc1 store; every workgroup polls all
// G x NV granules with L1-bypassing 16-byte loads (thread t -> value t >> 3 of member t & 7) until the tag fits the value,
// then three DPP stages add the members' contributions.  No counter, no flag, no fence; a torn or stale granule fails the
// tag test and is simply polled again.  Buffers alternate by exchange parity (a member can be at most one exchange ahead
// of the slowest).  Measured (scripts/dev/xcd_exchange.hip, profiles/r04_xcd_exchange.txt): 0.75 us per exchange for 8
// workgroups on one XCD, 1.4 us across 8 XCDs.  The launcher therefore starts 8 G workgroups and keeps those with
// blockIdx.x % 8 == 0 -- observed placement: block b runs on XCD b % 8, so the members share an L2 -- but NOTHING depends
// on that placement except speed: sc1 stores are valid hand-offs between any two CUs.  Every member reads its XCC_ID and
// the first exchange (sc1) carries a census of them into the refinement's info words; when it shows ONE XCD -- the same
// answer in every member -- the exchanges after it use plain stores: the granules never leave that XCD's L2 (gran_store).
// Both exchanges spin with a bound; a time-out (a member never became resident: shared or partitioned GPU) marks the
// launch as failed, every member winds down, and the host re-runs the refinement in one workgroup (blocking calls) or
// reports -12 (esac_hip_check).
enum : int { REFINE_SOLO = 0, REFINE_COOP = 1, REFINE_TEAM = 2 };
constexpr int TEAM_MAX = ESAC_REFINE_TEAM_MAX_K;  // members of a team: the CUs of one XCD

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Coop {
    int G, g;                      // number of cooperating workgroups, this one's index (G == 1: no cooperation)
    double* partials;              // REFINE_COOP: [2][G][32]
    unsigned long long* counter;   // REFINE_COOP: monotonic arrival counter, zeroed by the launcher; COOP_POISON is or-ed in on a time-out
    unsigned long long* failed;    // the launch tag of the most recent launch in which an exchange timed out (what the host reads)
    unsigned long long arrivals;   // exchanges passed so far (same in every thread of every workgroup)
    int expect;                    // workgroups an exchange waits for (= G; ESAC_DEBUG_COOP_STALL: G + 1, never reached)
    long spin_limit;               // REFINE_COOP: polls before the barrier gives up; REFINE_TEAM: ticks of the 100 MHz wall clock
    int* s_dead;                   // LDS flag: an exchange of this launch timed out somewhere
    bool dead;                     // ... as every thread of the workgroup saw it after its last exchange (workgroup-uniform)
    u32x4* gran;                   // REFINE_TEAM: [2][TEAM_MAX][32] granules
    unsigned long long tag;        // this launch's tag: (launch number << 20); the low 20 bits count a team's exchanges
    bool local;                    // REFINE_TEAM: every member runs on the same XCD (known after the census exchange; team_note_census)
};
// A workgroup that gives up at the counter barrier sets this bit: every waiter (now and at every later barrier) sees its
// target reached at once and reads the failure out of the same value -- nobody spins a second time.  (An OR: several
// workgroups timing out together cannot wrap the counter.)
constexpr unsigned long long COOP_POISON = 1ull << 62;

// Every thread of a workgroup that shares a refinement: its place among the G workgroups and the launch's exchange state.
// (s_dead is ordered before its first use by the workgroup barrier of the winner pick.)
__device__ __forceinline__ void coop_init(Coop& co, const KArgs& a, int G, int g, long spin_limit) {
    co.G = G;
    co.g = g;
    co.partials = a.coop_partials;
    co.gran = reinterpret_cast<u32x4*>(a.coop_partials);
    co.counter = a.coop_counter;
    co.failed = a.coop_counter + 1;
    co.tag = a.coop_tag;
    co.expect = G + a.coop_extra;
    co.spin_limit = spin_limit;
    co.local = false;
    if (threadIdx.x == 0) *co.s_dead = 0;
}

__device__ __forceinline__ void coop_mark_failed(Coop& co) {
    __hip_atomic_store(co.failed, co.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *co.s_dead = 1;
}

// ---- REFINE_TEAM: the tagged-granule exchange
__device__ __forceinline__ u32x4 gran_load(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// `local`: the census of the launch's first exchange found every member on ONE XCD -- the granule may stay in that XCD's L2 (a
// plain store; the pollers' loads find it there).  Otherwise sc1: written through to the memory side, a valid hand-off between
// any two CUs -- but the granule is then visible when the write has reached the address's HOME (which of the four I/O dies
// holds its 4 KB page): 0.09 us per exchange more for a page next to the team's XCD, 0.26 us for one across the package --
// the "67 or 71 us from process to process" of round 5 (scripts/dev/gran_shift_probe.py: address bit 13 of the buffer).
__device__ __forceinline__ void gran_store(u32x4* p, u32x4 v, bool local) {
    if (local)
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// thread t < NV publishes `own` = this member's total of value t for the exchange that team_collect() then completes
template <int NV>
__device__ __forceinline__ void team_publish(double own, const Coop& co) {
    static_assert(NV <= 32, "32 granules per member");
    if (co.dead) return;
    if (threadIdx.x < NV) {
        const unsigned long long want = co.tag | (co.arrivals + 1ull);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(own), tg = want ^ bits;
        u32x4* buf = co.gran + (size_t)(co.arrivals & 1ull) * (TEAM_MAX * 32);
        gran_store(buf + co.g * 32 + threadIdx.x, u32x4{(unsigned)bits, (unsigned)(bits >> 32), (unsigned)tg, (unsigned)(tg >> 32)}, co.local);
    }
}
// v[k] <- sum over the members of their value k, members added in one fixed order: bitwise identical in every member.
// REFINE_B = 256 threads.  Up to 8 members: thread t polls value t >> 3 of member t & 7, three DPP stages add the
// members.  More (up to 32): the G x 32 granules of an exchange are contiguous -- thread t polls granules t, t + 256, ...
// (member g >> 5, value g & 31; up to four 16-byte loads in flight), the values go through LDS (s_x: 32 x 32 doubles), thread
// t adds members (t & 7), (t & 7) + 8, ... of value t >> 3, the same three DPP stages finish.  s_tot: >= 32 doubles nobody
// else touches until the next workgroup barrier.
// team_collect_lds: the totals stay in LDS -- s_tot[k], and with NEG s_tot[32 + k] = -total (what the lane-dealt LM step
// gathers per lane, lm_lanes.hpp) -- visible to every thread when it returns; team_collect reads all of them back.
// WIDE = how many members the INSTANTIATION can collect: 8, 16 or 32 (co.expect <= WIDE; the launcher picks the kernel by the
// team's size).  One path per instantiation: as run-time branches the three paths, inlined at every exchange site, grew the
// team kernel by 2,000 instructions and cost the eight-member teams 1.5 us a call (round 6, profiles/r06_ab_team10.txt).
template <int NV, bool NEG, int WIDE>
__device__ __forceinline__ void team_collect_lds(Coop& co, double* s_tot, double* s_x = nullptr) {
    static_assert(NV <= 32 && REFINE_B == 256 && TEAM_MAX <= 32, "poll layout");
    static_assert(WIDE == 8 || WIDE == 16 || WIDE == 32, "members an instantiation collects");
    if (co.dead) return;
    const unsigned long long want = co.tag | (co.arrivals + 1ull);
    const u32x4* buf = co.gran + (size_t)(co.arrivals & 1ull) * (TEAM_MAX * 32);
    const int k = threadIdx.x >> 3, j = threadIdx.x & 7;
    double val = 0.0;
    bool timed_out = false;
    auto fits = [&](const u32x4& g, double& out) {
        const unsigned long long bits = (unsigned long long)g.x | ((unsigned long long)g.y << 32);
        const unsigned long long tg = (unsigned long long)g.z | ((unsigned long long)g.w << 32);
        out = __longlong_as_double((long long)bits);
        return (tg ^ bits) == want;
    };
    // The wait is bounded in TIME (co.spin_limit ticks of the 100 MHz wall clock, ESAC_TEAM_SPIN_LIMIT: a poll is an L2 round
    // trip whose duration depends on what else the chip is doing); the clock is first read when a poll has failed, then every
    // 64 polls -- together with the failure word: another member gave up (it carries this launch's tag), no point in
    // waiting out the limit.
    long long t_first = 0;
    auto give_up = [&](long spins) {
        if (spins == 1) {
            t_first = wall_clock64();
            return false;
        }
        if ((spins & 63) != 0) return false;
        return wall_clock64() - t_first > co.spin_limit || __hip_atomic_load(co.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == co.tag;
    };
    constexpr bool row16 = WIDE == 16;
    double val_b = 0.0;  // (9 .. 16 members: this thread's second value, 16 + t >> 4)
    if constexpr (WIDE == 16) {
        // 9 .. 16 members (round 6: ten members of 480 cells hold TWO cells per lane on the 60x80 grid where eight hold three):
        // thread t polls values t >> 4 and 16 + (t >> 4) of member t & 15 -- two granule loads in flight -- and four DPP stages
        // add the members of a 16-lane row in one fixed order.  No LDS staging, no barrier in front of the sums (the path for
        // up to 32 members below costs ~0.25 us a round: with it ten members measured 0.6 us SLOWER than eight).
        const int j16 = threadIdx.x & 15, k16 = threadIdx.x >> 4;
        const bool need_a = j16 < co.expect && k16 < NV, need_b = j16 < co.expect && 16 + k16 < NV;
        const u32x4* pa = buf + (need_a ? j16 * 32 + k16 : 0);
        const u32x4* pb = buf + (need_b ? j16 * 32 + 16 + k16 : 0);
        if (need_a) {
            for (long spins = 1;; spins++) {
                u32x4 ga, gb;
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(ga), "=&v"(gb)
                             : "v"(pa), "v"(pb)
                             : "memory");
                const bool oka = fits(ga, val), okb = !need_b || fits(gb, val_b);
                if (oka && okb) break;
                if (give_up(spins)) {
                    timed_out = true;
                    val = 0.0;
                    val_b = 0.0;
                    break;
                }
            }
        }
        if (!need_b) val_b = 0.0;
    } else if constexpr (WIDE == 8) {
        if (j < co.expect && k < NV) {
            const u32x4* p = buf + j * 32 + k;
            for (long spins = 1;; spins++) {
                if (fits(gran_load(p), val)) break;
                if (give_up(spins)) {
                    timed_out = true;
                    val = 0.0;
                    break;
                }
            }
        }
    } else {
        // four granules per thread, their loads in flight together (clamped addresses for the ones this thread does not
        // need); polled again until every needed one carries this exchange's tag
        static_assert(TEAM_MAX * 32 / REFINE_B == 4, "four granules per thread");
        const int total = co.expect * 32;
        const u32x4* ptr[4];
        bool need[4];
        double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int g = (int)threadIdx.x + i * REFINE_B;
            need[i] = g < total && (g & 31) < NV;
            ptr[i] = buf + (g < total ? g : 0);
        }
        for (long spins = 1;; spins++) {
            u32x4 g0, g1, g2, g3;
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
                         : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3])
                         : "memory");
            const bool ok0 = !need[0] || fits(g0, x[0]), ok1 = !need[1] || fits(g1, x[1]);
            const bool ok2 = !need[2] || fits(g2, x[2]), ok3 = !need[3] || fits(g3, x[3]);
            if (ok0 && ok1 && ok2 && ok3) break;
            if (give_up(spins)) {
                timed_out = true;
                break;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (need[i]) s_x[(int)threadIdx.x + i * REFINE_B] = timed_out ? 0.0 : x[i];
        if (timed_out) coop_mark_failed(co);
        timed_out = false;
        barrier_lds();
        if (k < NV)
            for (int m = j; m < co.expect; m += 8) val += s_x[m * 32 + k];  // members j, j + 8, j + 16, j + 24 in that order
    }
    if (timed_out) coop_mark_failed(co);
    val += dpp_move<0xB1>(val);   // lanes (0,1) (2,3) (4,5) (6,7)
    val += dpp_move<0x4E>(val);   // quads
    val += dpp_move<0x141>(val);  // all eight
    if constexpr (row16) {
        val += dpp_move<0x140>(val);  // row_mirror: the other eight members of the row
        val_b += dpp_move<0xB1>(val_b);
        val_b += dpp_move<0x4E>(val_b);
        val_b += dpp_move<0x141>(val_b);
        val_b += dpp_move<0x140>(val_b);
        const int j16 = threadIdx.x & 15, k16 = threadIdx.x >> 4;
        if (j16 == 0) {
            if (k16 < NV) {
                s_tot[k16] = val;
                if (NEG) s_tot[32 + k16] = -val;
            }
            if (16 + k16 < NV) {
                s_tot[16 + k16] = val_b;
                if (NEG) s_tot[48 + k16] = -val_b;
            }
        }
    } else if (j == 0 && k < NV) {
        s_tot[k] = val;
        if (NEG) s_tot[32 + k] = -val;
    }
    barrier_lds();
    co.dead = *co.s_dead != 0;
    co.arrivals += 1ull;
}
template <int NV, int WIDE>
__device__ __forceinline__ void team_collect(double (&v)[NV], Coop& co, double* s_tot, double* s_x = nullptr) {
    if (co.dead) {  // (v is DEFINED on every way out: the caller's accumulators then die at its reduction and are
                    // reduced in place -- left untouched here they stay live and every one of them is copied first)
#pragma unroll
        for (int kk = 0; kk < NV; kk++) v[kk] = 0.0;
        return;
    }
    team_collect_lds<NV, false, WIDE>(co, s_tot, s_x);
#pragma unroll
    for (int kk = 0; kk < NV; kk++) v[kk] = s_tot[kk];
}

// ---- draw(probs, training=false): argmax of the exact scores, first (global) index on ties
//      (esac_util.h:512-529; softmax is monotone, so the argmax of the scores is the argmax of the probabilities).
// Every thread returns the winner's (local) hypothesis index; contains a workgroup barrier.
template <int B>
__device__ __forceinline__ int refine_pick_winner(const KArgs& a, double* s_best, int* s_besti, int* s_bestg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double bs = -INFINITY;
    int bi = 0x7fffffff, bg = 0x7fffffff;
    for (int h = threadIdx.x; h < a.N; h += B) {
        if (!a.exact_flag[h]) continue;  // contenders = the hypotheses that were re-scored exactly
        const int g = global_hyp(a, h);
        const double s = a.scores[h];
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
    return bi == 0x7fffffff ? 0 : bi;
}

// ---- speculative forward (KArgs::spec_mode, esac_kernels.hip: k_spec_join)
// A gated launch (the second refinement of a speculative call is enqueued with the call, whatever the join will find) runs only
// when the join marked this call's speculation as failed.  The join runs on a stream of the context's own: thread 0 waits for its
// "done" word first.  s_flag: an int of LDS.  Workgroup-uniform; contains barriers when the launch is gated.
__device__ __forceinline__ bool spec_gate_closed(const KArgs& a, int* s_flag) {
    if (!a.spec_gate) return false;
    {
        // (the join itself waits up to ESAC_SPEC_WAIT_TICKS for its two words before it reports status 5: this wait outlasts it, so
        // that the caller's stream never falls idle in front of the join's report)
        if (threadIdx.x == 0) *s_flag = spec_wait_word(a, 7, 4) ? 1 : 0;
        __syncthreads();
        const bool came = *s_flag != 0;
        __syncthreads();  // (s_flag is the caller's to reuse)
        if (!came) return true;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // what the join and the selection beside it wrote, not what this CU's caches hold
        return __hip_atomic_load(a.spec_state + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch;
    }
}
// "The straggler chain may start": the first workgroup of the SPECULATIVE refinement is running, i.e. the launch has its CUs.
// (The chain's thousands of single-wavefront workgroups fill every SIMD; a refinement launched into that waits for it to drain.)
__device__ __forceinline__ void spec_open_chain(const KArgs& a) {
    if (a.spec_mode == 2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) spec_word_set(a, 3);
}
// "The speculative refinement is done" -- its record is in the workspace, its status word in spec_state[1].  Called by a
// whole wavefront of the workgroup that owns the outputs, behind its stores.
__device__ __forceinline__ void spec_refine_done(const KArgs& a) {
    if (a.spec_mode != 2) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(a.spec_state + 6, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The speculative refinement found nothing to refine (every hypothesis a straggler, or every settled score NaN): the record says so
// (hypothesis -1: k_spec_join then sends the call to the second refinement).  Workgroup-uniform; every workgroup sharing the
// refinement takes the same way out.  `win`: spec_pick_fast's (none: 0x7fffffff).
__device__ __forceinline__ bool spec_nothing_to_refine(const KArgs& a, int win, bool writer) {
    if (a.spec_mode != 2 || win != 0x7fffffff) return false;
    if (writer && threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            a.result[ESAC_RES_HYP_K] = -1.0;
            a.spec_state[1] = 0.0;
        }
        spec_refine_done(a);
    }
    return true;
}
// The hypothesis the speculative refinement works on is the fp32 argmax among the SETTLED hypotheses (first global
// index on ties) -- the selection proper (band, exact re-scores: k_select_rescore) runs beside the refinement, and k_spec_join
// checks that its winner is this one (the two differ only when the fp32 stream and the reference arithmetic order two
// near-equal scores differently).  Every thread returns the (local) index, 0x7fffffff: none.  Contains a workgroup barrier.
template <int B>
__device__ __forceinline__ int spec_pick_fast_but(const KArgs& a, int skip, double* s_best, int* s_besti, int* s_bestg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float bs = -INFINITY;
    int bi = 0x7fffffff, bg = 0x7fffffff;
    for (int h = threadIdx.x; h < a.N; h += B) {
        if (a.spec_flag[h] || h == skip) continue;
        const float s = a.fast_scores[h];
        const int g = global_hyp(a, h);
        if (s > bs || (s == bs && g < bg)) {  // (NaN: never)
            bs = s;
            bi = h;
            bg = g;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float os = __shfl_xor(bs, o);
        const int oi = __shfl_xor(bi, o);
        const int og = __shfl_xor(bg, o);
        if (os > bs || (os == bs && og < bg)) {
            bs = os;
            bi = oi;
            bg = og;
        }
    }
    __syncthreads();  // (s_best may still be read: a second pick)
    if (lane == 0) {
        s_best[wave] = (double)bs;
        s_besti[wave] = bi;
        s_bestg[wave] = bg;
    }
    __syncthreads();
    double ds = s_best[0];
    bi = s_besti[0];
    bg = s_bestg[0];
#pragma unroll
    for (int w = 1; w < B / 64; w++) {
        const double os = s_best[w];
        const int oi = s_besti[w];
        const int og = s_bestg[w];
        if (os > ds || (os == ds && og < bg)) {
            ds = os;
            bi = oi;
            bg = og;
        }
    }
    return bi;
}
template <int B>
__device__ __forceinline__ int spec_pick_fast(const KArgs& a, double* s_best, int* s_besti, int* s_bestg) {
    int win = spec_pick_fast_but<B>(a, -1, s_best, s_besti, s_bestg);
    if (a.spec_debug == 1 && win != 0x7fffffff) {  // ESAC_DEBUG_SPEC_SECOND_BEST (tests): the runner-up, if there is one
        const int second = spec_pick_fast_but<B>(a, win, s_best, s_besti, s_bestg);
        if (second != 0x7fffffff) win = second;
    }
    return win;
}

// ---- pose2trans (esac_util.h:537-548) and the result record.
// What the record needs besides the refinement's own outcome is known when the kernel starts (the selection kernel wrote
// it): loaded there, so that the end of the kernel is arithmetic and stores only.
struct RecordInputs {
    double prob, entropy;       // selection probability of the winner, entropy of the distribution (esac.cpp:157-158)
    unsigned long long status;  // epoch of the last sampling launch that met an out-of-range hypAssignment
};
__device__ __forceinline__ RecordInputs refine_record_inputs(const KArgs& a, double win_score) {
    const double smax = a.stats[0], ssum = a.stats[1];
    return RecordInputs{exp(win_score - smax) / ssum, a.stats[2], a.status[0]};
}

// Called by ALL 64 lanes of the first wavefront of the workgroup that owns the outputs.  s_rec: >= 36 doubles of LDS that
// nobody else touches any more.  census: XCD census of a team (hex digit x = members on XCD x), 0 otherwise.
__device__ __forceinline__ void refine_write_record(const KArgs& a, const RecordInputs& in, const double (&pose)[6], int win, double win_score, int e,
                                                    int nc, int accepted, int last_inliers, int lm_total, int map_buf, int mode, const Coop& co,
                                                    unsigned long long census, double* s_rec) {
    const int lane = threadIdx.x & 63;
    // an exchange between the workgroups sharing this refinement timed out: the record is not to be trusted
    const bool coop_failed = mode != REFINE_SOLO && (co.dead || __hip_atomic_load(co.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == co.tag);
    if (lane == 0) {
        double R[9];
        rodrigues_vec2mat<false>(pose, R, nullptr);
        double T[16];
        pose_to_inverse_transform(R, pose + 3, T);
        s_rec[ESAC_RES_SCORE_K] = win_score;
        s_rec[ESAC_RES_HYP_K] = (double)global_hyp(a, win);
        s_rec[ESAC_RES_EXPERT_K] = (double)(e + a.expert_base);
#pragma unroll
        for (int k = 0; k < 6; k++) s_rec[ESAC_RES_RVEC_K + k] = pose[k];
#pragma unroll
        for (int k = 0; k < 16; k++) s_rec[ESAC_RES_POSE_K + k] = (double)(float)T[k];
        s_rec[ESAC_RES_REF_STEPS_K] = (double)accepted;
        s_rec[ESAC_RES_INLIERS_K] = (double)last_inliers;
        s_rec[ESAC_RES_PROB_K] = in.prob;
        s_rec[ESAC_RES_ENTROPY_K] = in.entropy;
        s_rec[ESAC_RES_CONTENDERS_K] = (double)nc;
        s_rec[ESAC_RES_LM_ITERS_K] = (double)lm_total;
        s_rec[31] = (double)map_buf;  // which inlier-map buffer holds the last accepted set (-1: none)
        s_rec[32] = a.epoch;
        // status word: 1 = out-of-range hypAssignment seen by the launch that sampled, 3 = the shared refinement timed out
        s_rec[33] = coop_failed ? 3.0 : (in.status == (unsigned long long)a.sample_epoch) ? 1.0 : 0.0;
        if (a.refine_info) {
            int same = 0;
            for (int x = 0; x < 8; x++) same |= ((census >> (6 * x)) & 63ull) == (unsigned long long)co.G;
            unsigned lo = 0, hi = 0;  // bytes: members on XCD 0..3 / 4..7
            for (int x = 0; x < 4; x++) {
                lo |= (unsigned)((census >> (6 * x)) & 63ull) << (8 * x);
                hi |= (unsigned)((census >> (6 * (x + 4))) & 63ull) << (8 * x);
            }
            a.refine_info[0] = mode;
            a.refine_info[1] = co.G;
            a.refine_info[2] = (int)lo;
            a.refine_info[7] = (int)hi;
            a.refine_info[3] = mode == REFINE_TEAM ? same : 0;
            a.refine_info[4] = (int)co.arrivals;
            a.refine_info[5] = coop_failed ? 1 : 0;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wavefront: its LDS operations complete in order
    const double v = lane < 34 ? s_rec[lane] : 0.0;
    if (a.spec_mode == 2 && lane == 33) a.spec_state[1] = v;  // speculative refinement: the status word waits for k_spec_join
    if (lane < 32) {
        a.result[lane] = v;
        // ESAC_RES_VALID: lets a zero-padded exchange buffer tell "no record" (0) from a record (1); 3 = the workgroups sharing this
        // refinement timed out -- not a record either, but one the consumer must not take for an empty shard (k_pick_record: -12)
        if (a.result_user) a.result_user[lane] = lane == 31 ? (coop_failed ? 3.0 : 1.0) : v;
    }
    if (a.result_pin) {
        // straight into pinned host memory (the host polls instead of waiting for a copy kernel + stream-completion signal:
        // ~15-20 us of a blocking call's latency), 34 words + their check word, no fence (esac_kernels.hpp: pin_mix)
        pin_deliver(a.result_pin, v);
    }
    spec_refine_done(a);
}

}  // namespace esac
