// device_common.hpp -- cross-lane reductions and small helpers shared by the kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "esac_kernels.hpp"
#include "pose_math.hpp"

namespace esac {

// ---------------------------------------------------------------- cross-lane sums
// CTRL must be a pattern that gives EVERY lane a source (quad_perm, row_mirror, row_half_mirror: all this file uses):
// the move then has no "old" operand.  (__builtin_amdgcn_update_dpp(v, v, ...) ties the destination to a copy of v --
// two extra v_mov_b32 per double and stage, 56 of them in one 28-value wavefront reduction.)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
}
// all 64 lanes end up with the same (bitwise identical) total
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    {
        const unsigned x = __float_as_uint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    {
        const unsigned x = __float_as_uint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    return v;
}

// Sum NV doubles over a workgroup of B threads.  Fixed combination order ->
// run-to-run deterministic.  s_part: NV*(B/64) doubles, s_tot: NV doubles.
template <int NV, int B>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_part, double* s_tot) {
    constexpr int NW = B / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const double w = wave_sum(v[k]);
        if (lane == 0) s_part[wave * NV + k] = w;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) t += s_part[w * NV + threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = s_tot[k];
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)): with global stores in flight every barrier then costs a full store
// round trip to L2 (~1 us).  Use where the data exchanged through the barrier lives in LDS and the
// global stores of the phase are outputs nobody in this workgroup reads back.
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- transposed reduction of 28 doubles per lane ------------------------------------------------
// A dependent fp64 add has ~32 cycles of latency on gfx950 (issue: 4) and the pipeline is in-order,
// so 27 separate 6-stage butterflies cost ~6k cycles.  v_permlane32_swap / v_permlane16_swap
// exchange half-waves / odd-even rows between TWO registers in one instruction:
//     swap32(a, b): a' = [a.lo | b.lo], b' = [a.hi | b.hi]   ->  a' + b' = [sum_halves(a) | sum_halves(b)]
// i.e. one swap + one add reduces two values by one stage AND halves the number of live values.
// 28 values -> 14 -> 7 registers; inside the 16-lane rows the same idea with bank-masked DPP moves (a DPP move
// can leave the lanes of chosen 4-lane banks untouched):
//     pair8(a, b):  x = a, lanes 8-15 <- b[l-8];  y = b, lanes 0-7 <- a[l+8]   ->  x + y = [a: 8 + 8 | b: 8 + 8]
//     pair4(a, b):  the same between neighbouring banks                        ->  banks [a | b | a' | b'] ...
// 7 -> 4 -> 2 registers, and two plain butterflies inside the quads: ~115 instructions where 27 separate butterflies
// are ~600.  Afterwards lane 16 r + 4 b of the two registers holds a wavefront total (see wave_totals28_to_lds).
__device__ __forceinline__ double swap32_pairsum(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap16_pairsum(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// Stage 1 of the workgroup sum of up to 28 doubles per thread: every wavefront leaves ITS 28 totals in
// s_part[wave * 28 ...] (ends with a workgroup barrier).  Stage 2 (workgroup_total28): thread t < 28 adds the wavefronts'
// totals of value t in wavefront order.
// keep, with the lanes of the 4-lane banks in BANKS (bit b: lanes 4b .. 4b+3 of every row) replaced by from[CTRL]
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_merge(double keep, double from) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(from), CTRL, 0xf, BANKS, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(from), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_pair8(double a, double b) {  // lanes 0-7: a[l] + a[l+8], lanes 8-15: b[l-8] + b[l]
    return dpp_merge<0x128, 0xC>(a, b) + dpp_merge<0x128, 0x3>(b, a);  // row_ror:8 (l ^ 8 either way round)
}
__device__ __forceinline__ double dpp_pair4(double a, double b) {  // banks 0, 2: a[l] + a[l+4]; banks 1, 3: b[l-4] + b[l]
    return dpp_merge<0x114, 0xA>(a, b) + dpp_merge<0x104, 0x5>(b, a);  // row_shr:4 (lane l reads l - 4), row_shl:4 (l + 4)
}
template <int NV>
__device__ __forceinline__ void wave_totals28_to_lds(const double (&v)[NV], double* s_part) {
    static_assert(NV <= 28, "at most 28 values");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double w[14], u[7];
#pragma unroll
    for (int p = 0; p < 14; p++) {
        const double a = (2 * p < NV) ? v[2 * p] : 0.0;
        const double b = (2 * p + 1 < NV) ? v[2 * p + 1] : 0.0;
        w[p] = swap32_pairsum(a, b);
    }
#pragma unroll
    for (int q = 0; q < 7; q++) u[q] = swap16_pairsum(w[2 * q], w[2 * q + 1]);
    // row r of u[q] now holds 16 partial sums of value 4q + sub(r), sub(r) = (r & 1) << 1 | r >> 1
    const double r0 = dpp_pair8(u[0], u[1]), r1 = dpp_pair8(u[2], u[3]), r2 = dpp_pair8(u[4], u[5]);
    const double r3 = u[6] + dpp_move<0x128>(u[6]);  // (both halves of a row)
    double q0 = dpp_pair4(r0, r1), q1 = dpp_pair4(r2, r3);  // banks of a row: u0 u2 u1 u3 | u4 u6 u5 u6
    q0 += dpp_move<0xB1>(q0);
    q1 += dpp_move<0xB1>(q1);
    q0 += dpp_move<0x4E>(q0);
    q1 += dpp_move<0x4E>(q1);
    if ((lane & 3) == 0) {
        const int r = lane >> 4, bank = (lane >> 2) & 3;
        const int sub = ((r & 1) << 1) | (r >> 1);
        const int m = ((bank & 1) << 1) | (bank >> 1);  // 0 2 1 3
        s_part[wave * 28 + 4 * m + sub] = q0;
        if (bank != 3) s_part[wave * 28 + 4 * (4 + m) + sub] = q1;
    }
    barrier_lds();
}
template <int B>
__device__ __forceinline__ double workgroup_total28(const double* s_part) {  // meaningful in threads < 28
    double t = 0;
    if (threadIdx.x < 28) {
#pragma unroll
        for (int k = 0; k < B / 64; k++) t += s_part[k * 28 + threadIdx.x];
    }
    return t;
}

// Sum up to 28 doubles per thread over a workgroup of B threads; every thread receives all totals.
// s_part: 28*(B/64) doubles, s_tot: 28 doubles.  Deterministic (fixed combination order).
template <int NV, int B>
__device__ __forceinline__ void block_sum28(double (&v)[NV], double* s_part, double* s_tot) {
    wave_totals28_to_lds<NV>(v, s_part);
    const double t = workgroup_total28<B>(s_part);
    if (threadIdx.x < 28) s_tot[threadIdx.x] = t;
    barrier_lds();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = s_tot[k];
}

// Batched calls (esac_hip_forward_batch): workgroups with blockIdx.y = b serve frame b.  Every per-call
// buffer is laid out frame-major, so a frame's view is the same KArgs with offset pointers; frame b draws
// the RNG streams of call + b, which makes a batch bit-identical to B sequential calls.
__device__ __forceinline__ void frame_view(KArgs& a, int fr) {
    if (fr == 0) return;
    const size_t N = (size_t)a.N, P = (size_t)a.H * a.W, f = (size_t)fr;
    a.sc += f * a.sc_frame_stride;
    a.assign += f * N;
    a.call += (uint64_t)fr;
    a.hyps += f * N * 6;
    a.hyps_R += f * N * 9;
    a.rt32 += f * N * 12;
    a.sample_xy += f * N * 8;
    a.tries += f * N;
    a.samp_resume += f * N;
    a.samp_round += f * N;
    a.best_try += f * N;
    a.fast_scores += f * N;
    a.scores += f * N;
    a.exact_flag += f * N;
    a.n_contenders += f * 4;
    a.sel_partials += (size_t)f * N * ESAC_SELECT_SPLIT;
    a.sel_arrived += f * N;
    a.stats += f * 4;
    if (a.errs) a.errs += f * P;
    a.inlier_map += f * 2 * P;
    a.corr_list = static_cast<char*>(a.corr_list) + f * (size_t)corr_entries(a.H * a.W) * 16;
    a.inlier_counts += f * (ESAC_MAX_REF_STEPS_K + 1);
    a.result += f * 32;
    a.tstamps = nullptr;  // the span measurement follows frame 0 only
    if (a.scores_user) a.scores_user += f * N;
    if (a.result_user) a.result_user += f * 32;
    if (a.result_pin) a.result_pin += f * ESAC_PIN_DOUBLES;
}

__device__ __forceinline__ void frame_view(KArgs& a) { frame_view(a, (int)blockIdx.y); }

// expert of hypothesis h.  With a single expert the answer is known without the (dependent, ~0.5 us) load every kernel
// would otherwise start with; hypAssignment values other than 0 are meaningless there (esac.cpp:189 would return them).
// A value outside [0,E) (the reference would index out of bounds, esac_util.h:183; only a device-resident assignment can
// get this far, the host API range-checks CPU tensors) is mapped to expert 0 here -- never an out-of-bounds read -- and
// reported by the sampling kernel through KArgs::status (flag_bad_assignment), which the host turns into an error.
__device__ __forceinline__ int expert_of(const KArgs& a, int h) {
    if (a.E == 1) return 0;
    const long long e = a.assign[h];
    return (unsigned long long)e < (unsigned long long)a.E ? (int)e : 0;
}
__device__ __forceinline__ void flag_bad_assignment(const KArgs& a, int h) {
    if (a.E == 1) return;
    const long long e = a.assign[h];
    if ((unsigned long long)e >= (unsigned long long)a.E) atomicMax(a.status, (unsigned long long)a.sample_epoch);
}

__device__ __forceinline__ Cam make_cam(const KArgs& a) {
    // camMat is a float matrix widened to double by the solver (esac.cpp:93-97)
    return Cam{(double)a.focal, (double)a.focal, (double)a.ppx, (double)a.ppy};
}
// createSampling (esac_util.h:64-66): integer pixel centre of cell (x,y), then Point2f
// global hypothesis index: keys the RNG stream and breaks ties, independent of the sharding
__device__ __forceinline__ int global_hyp(const KArgs& a, int h) { return a.hyp_index ? a.hyp_index[h] : a.hyp_offset + h; }
// where hypothesis h's score goes in the caller's score vector (ESAC_FLAG_SCORES_BY_INDEX: by global index)
__device__ __forceinline__ size_t user_slot(const KArgs& a, int h) {
    return (a.flags & ESAC_FLAG_SCORES_BY_INDEX_K) ? (size_t)global_hyp(a, h) : (size_t)h;
}
__device__ __forceinline__ float cell_px(const KArgs& a, int x) { return (float)(x * a.sub + a.sub / 2 - a.shift_x); }
__device__ __forceinline__ float cell_py(const KArgs& a, int y) { return (float)(y * a.sub + a.sub / 2 - a.shift_y); }

// Origin of the fp32 scoring stream for one expert's map: the scene point of the middle cell (0 when it is not finite).
// The stream evaluates R*(X - c) + (t + R*c) with the second term formed in double when the hypothesis is stored, so its
// rounding error scales with the extent of the scene around c, not with the distance of the scene from the world origin
// (world-frame maps of outdoor scenes sit ~1e3 m out: float(t) alone would shift every projection by ~1e-2 px).
struct Centre {
    float x, y, z;
};
__device__ __forceinline__ Centre map_centre(const KArgs& a, const float* __restrict__ map) {
    const int P = a.H * a.W, mid = (a.H >> 1) * a.W + (a.W >> 1);
    Centre c{map[mid], map[P + mid], map[2 * P + mid]};
    const bool ok = fabsf(c.x) <= 3.0e38f && fabsf(c.y) <= 3.0e38f && fabsf(c.z) <= 3.0e38f;  // false for NaN / inf
    if (!ok) c = Centre{0.0f, 0.0f, 0.0f};
    return c;
}


// ---- speculative forward (KArgs::spec_mode): hand-offs between the streams of a call
// Hand-off words between the streams of a speculative call: the epoch of the call they belong to, written through to memory
// (sc1) and polled past the caches -- valid between any two CUs.
__device__ __forceinline__ void spec_word_set(const KArgs& a, int which) {
    __hip_atomic_store(a.spec_state + which, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// Every wait for such a word is bounded in wall time: a word that never comes (the other stream's launch failed, or something
// else holds its queue) costs ESAC_SPEC_WAIT_TICKS and is reported (status 5: the host runs the call again in stream order) -- never a hang.
constexpr long long ESAC_SPEC_WAIT_TICKS = 2000000;  // 20 ms of the 100 MHz wall clock
#ifndef ESAC_SPEC_POLL_SLEEP
#define ESAC_SPEC_POLL_SLEEP 4  // x 64 cycles between two polls
#endif
__device__ __forceinline__ bool spec_wait_word(const KArgs& a, int which, int patience = 1, int which2 = -1) {
    // (relaxed loads past the caches while waiting, ONE acquire when the word is there: an acquire per poll invalidates the XCD's
    // L2 every 0.2 us under whatever else runs on it -- with the join and the ten members of the gated refinement polling
    // beside the straggler chain, k_sample_decide took 23 us instead of 13.6.  which2: a second word to wait for, polled in
    // the same trip to memory.)
    const long long t0 = wall_clock64();
    for (;;) {
        const double w1 = __hip_atomic_load(a.spec_state + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double w2 = which2 >= 0 ? __hip_atomic_load(a.spec_state + which2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.epoch;
        if (w1 == a.epoch && w2 == a.epoch) break;
        __builtin_amdgcn_s_sleep(ESAC_SPEC_POLL_SLEEP);
        if (wall_clock64() - t0 > patience * ESAC_SPEC_WAIT_TICKS) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
}  // namespace esac
