// esac_refine.hip -- K4: draw(argmax) + refineHyp + pose2trans in ONE workgroup, no host round trips.
//
// Reference: esac.cpp:153-187, refineHyp esac_util.h:378-454, draw esac_util.h:505-530,
// pose2trans esac_util.h:537-548.  The reference runs this serially on one CPU thread
// (esac.cpp:167 is outside every omp region); here it is the latency-critical tail of
// the whole call.  On gfx950 a DEPENDENT fp64 op costs ~40 cycles (issue: 4) and the
// pipeline is in-order, so everything below is organised to keep independent fp64
// work adjacent in the instruction stream and the dependent chains short:
//   * one fused pass decides `err < tau` for every cell (fp32 screen, fp64 screen, bit-exact evaluation only inside
//     the last band), counts the inliers and appends them to a WAVE-LOCAL list (16 B per correspondence in LDS:
//     x,y,z + packed pixel) -- no cross-wavefront prefix, one barrier per pass;
//   * per-point LM work: with fx = fy the normal matrix is a set of 24 monomial moments in (x, y, 1/z)
//     (lm_math.hpp, ~68 fp64 ops per correspondence), Newton reciprocal instead of IEEE division, the projection
//     chain of the next correspondences software-pipelined against the accumulation of the current ones;
//   * the 24 fp64 sums of a pass are reduced with v_permlane32/16_swap pair-sums
//     (halving the live values per stage) + DPP row stages (device_common.hpp);
//   * every lane carries the 6-parameter LM state redundantly: all lanes take the same
//     branches from the same reduced sums, so nothing is broadcast;
//   * the pass at a trial point already accumulates the normal equations there
//     (speculative Jacobian): an accepted step -- the common case -- costs one pass.
// All discrete decisions (inlier test `err < tau`, stopping rule, argmax) use the
// reference's exact arithmetic (project_exact_err, contraction off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "refine_common.hpp"

namespace esac {

// One lane: arrive at the barrier and wait for the others.  On a time-out (a workgroup of this launch never became
// resident: the GPU is shared, partitioned, or smaller than the launcher assumed) the counter is poisoned so that every
// other waiter -- now and at all later barriers -- falls through immediately, the failure is flagged for the host, and the
// LDS flag tells this workgroup to wind down.
__device__ __forceinline__ void coop_arrive_and_wait(Coop& co) {
    __hip_atomic_fetch_add(co.counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (co.arrivals + 1ull) * (unsigned long long)co.expect;
    long spins = 0;
    unsigned long long seen;
    while ((seen = __hip_atomic_load(co.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > co.spin_limit) {
            seen = __hip_atomic_fetch_or(co.counter, COOP_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | COOP_POISON;
            break;
        }
    }
    if (seen & COOP_POISON) coop_mark_failed(co);
}

// v[0..NV) <- sum over the G workgroups of their v (every thread of a workgroup enters with the same v, leaves with the
// same total; s_tot: >= 28 doubles of LDS scratch).  REFINE_COOP only.
template <int NV>
__device__ __forceinline__ void coop_allreduce(double (&v)[NV], Coop& co, double* s_tot, double* s_part) {
    static_assert(NV <= 28, "partials hold 32 doubles per workgroup");
    if (co.G == 1 || co.dead) return;  // dead: the sums are meaningless from here on, the callers wind the refinement down
    double* buf = co.partials + (size_t)(co.arrivals & 1ull) * co.G * 32;
    __syncthreads();
    // publish: the sums go through LDS (static register indices only), NV lanes store them, ONE lane releases them at
    // device scope and arrives
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) s_tot[k] = v[k];
    }
    __syncthreads();
    // publish with device-scope (write-through, sc1) stores, drained before the arrival; read with device-scope loads: no
    // release / acquire fence (each ~1.7 us: an L2 write-back / an L1 invalidate) on either side of the barrier
    if (threadIdx.x < NV) {
        __hip_atomic_store(buf + (size_t)co.g * 32 + threadIdx.x, s_tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // NV <= 28: all of them lanes of wavefront 0, as is the arriving lane
    }
    if (threadIdx.x == 0) coop_arrive_and_wait(co);
    __syncthreads();
    co.dead = *co.s_dead != 0;  // written before the barrier above by lane 0, read by everyone: workgroup-uniform
    // gather: 8 groups of 32 lanes, group j adds the partials of workgroups j, j+8, j+16, ... (loads independent of each
    // other), then value k adds its 8 group sums -- one fixed order for every workgroup: bitwise identical totals
    {
        const int k = threadIdx.x & 31, j = threadIdx.x >> 5;
        double t = 0;
        if (k < NV)
            for (int w = j; w < co.G; w += 8) t += __hip_atomic_load(buf + (size_t)w * 32 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_part[j * 32 + k] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) t += s_part[j * 32 + threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = s_tot[k];
    __syncthreads();
    co.arrivals += 1ull;
}

// Fused pass over the whole grid at `pose`:
//   a.errs[i]  = min(reprojection error, maxReproj)                (getReproErrs, esac_util.h:292-360);
//                reference-exact near tau, fp32-accurate (~1e-3 px) elsewhere -- see the screening below;
//                only stored when the caller asked for it (esac_hip_set_debug), nothing downstream reads it
//   map_out[i] = err < tau                                         (localInlierMap, esac_util.h:401-414)
//   list[...]  = the inliers, compacted per wavefront: wavefront w owns list[w * corr_region(P) ...], n_wave entries
// Returns the inlier count (same value in every thread).
// Cooperative form: this workgroup handles the cells [cell0, cell0 + Pn) of the P-cell grid (cell0 a multiple of the
// trip size); the returned count is the total over all workgroups.
template <int B, bool VEC, int MODE, typename ListPtr>
__device__ __forceinline__ int error_pass_impl(const KArgs& a, const float* __restrict__ mx, int P, const double pose[6],
                                               const Cam& cam, ListPtr list, int& n_wave, uint8_t* __restrict__ map_out,
                                               int* s_wcnt, long long* g_cyc, int cell0 = 0, int Pn = -1, Coop* co = nullptr,
                                               double* s_tot = nullptr, double* s_part = nullptr) {
    if (Pn < 0) Pn = P;
    CYC_DECL;
    // VEC: every lane owns G groups of 4 CONSECUTIVE cells per trip (W % 4 == 0: a group never straddles a
    // row, planes are 16-byte aligned) -> float4 loads, one float4 + one packed-byte store per group.
    // Otherwise: U cells per lane strided by B, scalar accesses.
    constexpr int G = VEC ? ERR_UNROLL / 4 : ERR_UNROLL;  // load groups per lane per trip
    constexpr int L = VEC ? 4 : 1;                // cells per group
    constexpr int U = G * L, NW = B / 64;
    static_assert(ERR_UNROLL % 4 == 0, "ERR_UNROLL must be a multiple of 4");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double R[9];
    rodrigues_vec2mat<false>(pose, R, nullptr);
    float Rf[9], tf[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Rf[k] = (float)R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) tf[k] = (float)pose[3 + k];
    // fp32 screening band (see below): first-order bound of the fp32 evaluation's error with 10 ulp of
    // head-room on every term.  Camera-frame coordinates carry ~eps*(|Xc|+2|t|), amplified by
    // f/zc*(1+|x/z|); the pixel-space terms are bounded by eps*(image extent + tau).
    const float tmag2 = 2.0f * (fabsf(tf[0]) + fabsf(tf[1]) + fabsf(tf[2]));
    const float kf = 1.2e-6f * a.focal;
    const float kpix = 1.2e-6f * 4.0f * ((float)(a.W + a.H) * (float)a.sub + fabsf(a.ppx) + fabsf(a.ppy) + a.tau);
    // second screen (fp64): the reference's float rounding of the projected pixel, see below
    const double band2 = 2.4e-7 * ((double)(a.W > a.H ? a.W : a.H) * a.sub + abs(a.shift_x) + abs(a.shift_y) + (double)a.tau) + 2e-6;
    const double tlo = (double)a.tau - band2, thi = (double)a.tau + band2;
    const double tau_lo2 = tlo > 0 ? tlo * tlo : -1.0, tau_hi2 = thi * thi;
    const float tau_below = nextafterf(a.tau, -INFINITY);
    // a lane's groups advance by B*L cells per step: (row, col) kept incrementally, one division in total
    const int stepR = (B * L) / a.W, stepC = B * L - stepR * a.W;
    int row = (cell0 + (int)threadIdx.x * L) / a.W, col = cell0 + (int)threadIdx.x * L - row * a.W;
    const int region = corr_region(Pn);  // list entries per wavefront
    const auto wlist = list + (size_t)wave * region;
    int wcount = 0;
    const int cell_end = cell0 + Pn;
    for (int start = cell0; start < cell_end; start += B * U) {
        const bool full = start + B * U <= cell_end;  // wave-uniform: no bounds checks in full trips
        float X[U], Y[U], Z[U], pxf[U], pyf[U], errv[U];
        int coli[U], rowi[U], cell[G];
        bool flag[U];
        CYC_BEGIN();
#pragma unroll
        for (int g = 0; g < G; g++) {  // all loads first: one memory latency for the U cells
            const int i = start + (g * B + (int)threadIdx.x) * L;
            cell[g] = i;
            const int ic = full ? i : (i < cell_end ? i : cell_end - L);
            if (VEC) {
                const float4 vx = *reinterpret_cast<const float4*>(mx + ic);
                const float4 vy = *reinterpret_cast<const float4*>(mx + P + ic);
                const float4 vz = *reinterpret_cast<const float4*>(mx + 2 * P + ic);
                X[4 * g] = vx.x; X[4 * g + 1] = vx.y; X[4 * g + 2] = vx.z; X[4 * g + 3] = vx.w;
                Y[4 * g] = vy.x; Y[4 * g + 1] = vy.y; Y[4 * g + 2] = vy.z; Y[4 * g + 3] = vy.w;
                Z[4 * g] = vz.x; Z[4 * g + 1] = vz.y; Z[4 * g + 2] = vz.z; Z[4 * g + 3] = vz.w;
            } else {
                X[g] = mx[ic];
                Y[g] = mx[P + ic];
                Z[g] = mx[2 * P + ic];
            }
#pragma unroll
            for (int l = 0; l < L; l++) {
                coli[g * L + l] = col + l;
                rowi[g * L + l] = row;
                pxf[g * L + l] = (float)cell_pxi(a, col + l);
                pyf[g * L + l] = (float)cell_pyi(a, row);
            }
            col += stepC;
            row += stepR;
            if (col >= a.W) {
                col -= a.W;
                row++;
            }
        }
        CYC_END(10);
        CYC_BEGIN();
        // fp32 screening: a cell whose fp32 error is farther from tau than the bound above cannot change
        // side under the reference arithmetic.  Only cells inside that band -- a few per ten thousand --
        // pay for the exact fp64 evaluation, so `err < tau` is decided exactly everywhere.
        // Two cells per packed instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): with ONE wavefront on the SIMD a
        // packed FMA issues in 6.5 cycles against 5.5 for a plain one (scripts/dev/valu_rate.hip) -- 1.7x on the
        // arithmetic of this pass; component by component the same operations in the same order as the plain form.
        bool need[U];
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto sp = [](float v) { return f2{v, v}; };
            auto fma2 = [](f2 x, f2 y, f2 z) { return __builtin_elementwise_fma(x, y, z); };
            auto abs2 = [](f2 v) { return f2{fabsf(v.x), fabsf(v.y)}; };
            const f2 Xp = {X[u], X[u + 1]}, Yp = {Y[u], Y[u + 1]}, Zp = {Z[u], Z[u + 1]};
            const f2 xc = fma2(sp(Rf[0]), Xp, fma2(sp(Rf[1]), Yp, fma2(sp(Rf[2]), Zp, sp(tf[0]))));
            const f2 yc = fma2(sp(Rf[3]), Xp, fma2(sp(Rf[4]), Yp, fma2(sp(Rf[5]), Zp, sp(tf[1]))));
            const f2 zc = fma2(sp(Rf[6]), Xp, fma2(sp(Rf[7]), Yp, fma2(sp(Rf[8]), Zp, sp(tf[2]))));
            const f2 iz = {(zc.x != 0.0f) ? __builtin_amdgcn_rcpf(zc.x) : 1.0f, (zc.y != 0.0f) ? __builtin_amdgcn_rcpf(zc.y) : 1.0f};
            const f2 du = f2{pxf[u], pxf[u + 1]} - fma2(sp(a.focal), xc * iz, sp(a.ppx));
            const f2 dv = f2{pyf[u], pyf[u + 1]} - fma2(sp(a.focal), yc * iz, sp(a.ppy));
            const f2 d2 = fma2(du, du, dv * dv);
            errv[u] = __builtin_amdgcn_sqrtf(d2.x);
            errv[u + 1] = __builtin_amdgcn_sqrtf(d2.y);
            const f2 aiz = abs2(iz), axy = abs2(xc) + abs2(yc);
            const f2 guard = fma2(sp(kf) * aiz * (axy + abs2(zc) + sp(tmag2)), fma2(axy, aiz, sp(1.0f)), sp(kpix));
            // not finite, or within the band -> the second screen / the reference-exact evaluation decides
            need[u] = !(fabsf(errv[u] - a.tau) > guard.x);
            need[u + 1] = !(fabsf(errv[u + 1] - a.tau) > guard.y);
        }
        CYC_END(11);
        CYC_BEGIN();
        // Second screen, per sub-step and only where some lane asked for it (wave-uniform branches): the error in
        // fp64 with a Newton reciprocal and no square root (~25 dependent ops instead of the ~70 of the reference
        // sequence).  The reference rounds the projection to float before taking the difference, so its error differs
        // from the fp64 value by at most ~sqrt(2)/2 ulp of the pixel coordinate (+1e-6); `band2` leaves 3.5x of
        // head-room on that.  What is still undecided inside that band gets the bit-exact evaluation.
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!__any(need[u])) continue;
            CYC_ADD(15, 1);
            const double Xd = X[u], Yd = Y[u], Zd = Z[u];
            const double xc = R[0] * Xd + R[1] * Yd + R[2] * Zd + pose[3];
            const double yc = R[3] * Xd + R[4] * Yd + R[5] * Zd + pose[4];
            const double zc = R[6] * Xd + R[7] * Yd + R[8] * Zd + pose[5];
            const double iz = zc ? fast_rcp(zc) : 1.0;
            const double du = (double)pxf[u] - (xc * iz * cam.fx + cam.cx);
            const double dv = (double)pyf[u] - (yc * iz * cam.fy + cam.cy);
            const double e2 = du * du + dv * dv;
            const bool in2 = e2 < tau_lo2, out2 = e2 > tau_hi2;  // NaN: neither
            const bool undecided = need[u] && !(in2 || out2);
            if (need[u] && in2) errv[u] = fminf(errv[u], tau_below);   // keep the fp32 value on the decided side
            if (need[u] && out2) errv[u] = fmaxf(errv[u], a.tau);
            if (__any(undecided)) {
                const float exact = project_exact_err(R, pose + 3, cam, X[u], Y[u], Z[u], pxf[u], pyf[u]);
                if (undecided) errv[u] = exact;
            }
        }
        CYC_END(12);
        CYC_BEGIN();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const bool in_range = full || cell[g] < cell_end;
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int u = g * L + l;
                errv[u] = errv[u] < a.max_reproj ? errv[u] : a.max_reproj;  // std::min(l, maxReproj), esac_util.h:358
                flag[u] = in_range && (errv[u] < a.tau);
            }
            if (in_range) {
                if (VEC) {
                    if (a.errs)  // debug option: the error image itself is nobody's input
                        *reinterpret_cast<float4*>(a.errs + cell[g]) = make_float4(errv[4 * g], errv[4 * g + 1], errv[4 * g + 2], errv[4 * g + 3]);
                    *reinterpret_cast<uint32_t*>(map_out + cell[g]) = (flag[4 * g] ? 1u : 0u) | (flag[4 * g + 1] ? 0x100u : 0u) |
                                                                        (flag[4 * g + 2] ? 0x10000u : 0u) | (flag[4 * g + 3] ? 0x1000000u : 0u);
                } else {
                    if (a.errs) a.errs[cell[g]] = errv[g];
                    map_out[cell[g]] = flag[g] ? 1 : 0;
                }
            }
        }
        // wave-local compaction: every wavefront appends to ITS OWN region of the list (ballot prefix, no
        // cross-wavefront exchange, no barrier); the order (wavefront, trip, sub-step, lane) is fixed -> deterministic
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned long long m = __ballot(flag[u]);
            if (flag[u]) {
                const int slot = wcount + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < region)
                    wlist[slot] = Corr{X[u], Y[u], Z[u], ((uint32_t)rowi[u] << 16) | (uint32_t)coli[u]};
            }
            wcount += __popcll(m);
        }
        CYC_END(13);
    }
    CYC_BEGIN();
    if (lane == 0) s_wcnt[wave] = wcount;
    barrier_lds();
    int base = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) base += s_wcnt[w];
    n_wave = wcount < region ? wcount : region;
    if (MODE != REFINE_SOLO && co->G > 1) {  // total over the cooperating workgroups (exact in double: counts < 2^28)
        double cnt[1] = {(double)base};
        coop_allreduce<1>(cnt, *co, s_tot, s_part);
        base = co->dead ? 0 : (int)cnt[0];  // dead: "no inliers" ends the refinement loop at once
    }
    CYC_END(14);
    return base;
}

// One pass over the compacted correspondences at `param`: residual norm^2 (returned) and the
// (rvec,tvec)-space normal equations U21 / g6 at that point.  `list` / `n`: this wavefront's region and its fill.
// Pixel position of grid cell (col, row): px = col * sub + offx (createSampling, esac_util.h:64-66).
struct PxMap {
    int sub, offx, offy;
};
template <int B, int MODE, typename ListPtr>
__device__ __forceinline__ double lm_pass(ListPtr list, int n, const double param[6], const Cam& cam, const PxMap& pm, double U21[21],
                                          double g6[6], double* s_part, double* s_tot, long long* g_cyc, Coop* co = nullptr) {
    CYC_DECL;
    CYC_BEGIN();
    double R[9], Mw[3][3];
    LmTrig tg;
    lm_pose_rotation(param, R, tg);
    lm_pose_left_jacobian(tg, Mw);
    CYC_END(4);
    CYC_BEGIN();
    double mom[LM_NMOM];
#pragma unroll
    for (int k = 0; k < LM_NMOM; k++) mom[k] = 0;
    // LM_NP correspondences per lane per trip; every wavefront walks the region of the list it filled itself.
    // Software-pipelined: the terms of trip k+1 (a dependent chain per point) are computed in the same basic block
    // as the 24 accumulator chains of trip k, so the in-order pipeline always has independent work to issue.
    const int lane = threadIdx.x & 63;
    const int trips = (n + LM_NP * 64 - 1) / (LM_NP * 64);
    auto fetch = [&](int trip, LmTerms<LM_NP>& out) {
        double X[LM_NP], Y[LM_NP], Z[LM_NP], mxp[LM_NP], myp[LM_NP];
        bool on[LM_NP];
#pragma unroll
        for (int p = 0; p < LM_NP; p++) {
            const int jj = (trip * LM_NP + p) * 64 + lane;
            on[p] = jj < n;
            const Corr c = list[min(jj, n - 1)];  // clamped, unconditional: no control flow inside the pipelined body
            X[p] = (double)c.x; Y[p] = (double)c.y; Z[p] = (double)c.z;
            mxp[p] = (double)((int)(c.row_col & 0xffffu) * pm.sub + pm.offx);
            myp[p] = (double)((int)(c.row_col >> 16) * pm.sub + pm.offy);
        }
        lm_point_terms<LM_NP>(R, param + 3, cam, X, Y, Z, mxp, myp, on, out);
    };
    if (trips > 0) {
        LmTerms<LM_NP> cur;
        fetch(0, cur);
        for (int trip = 1; trip < trips; trip++) {
            LmTerms<LM_NP> nxt;
            fetch(trip, nxt);
            lm_accumulate_moments<LM_NP>(cur, mom);
            cur = nxt;
        }
        lm_accumulate_moments<LM_NP>(cur, mom);
    }
    CYC_END(5);
    CYC_BEGIN();
    block_sum28<LM_NMOM, B>(mom, s_part, s_tot);
    if (MODE == REFINE_COOP) coop_allreduce<LM_NMOM>(mom, *co, s_tot, s_part);
    CYC_END(6);
    CYC_BEGIN();
    CYC_PIN(mom, LM_NMOM);
    double acc[LM_NACC];
    lm_moments_to_acc(mom, cam.fx, acc);
    lm_transform_t(acc, Mw, param + 3, U21, g6);  // (K = [t]x Mw folded in: ~130 fused operations instead of ~235)
    CYC_PIN(U21, 21);
    CYC_PIN(g6, 6);
    CYC_END(7);
    CYC_ADD(9, 1);
    return acc[26];
}

// cv::solvePnP(ITERATIVE, useExtrinsicGuess): CvLevMarq with 6 parameters, max_iter 20, eps FLT_EPSILON,
// lambda = 10^k from k = -3, k++ while a step made the error worse (<= 16), k-- after an accepted step.
// Written as ONE loop around ONE lm_pass call site (the kernel must stay inside the instruction cache:
// with the pass inlined at several sites the code grew to 170 KB and every phase ran from cold code).
template <int B, int MODE, typename ListPtr>
__device__ __forceinline__ int lm_refit(ListPtr list, int n, double pose[6], const Cam& cam, const PxMap& pm, double* s_part,
                                        double* s_tot, const double* s_pow10, long long* g_cyc, Coop* co = nullptr) {
    CYC_DECL;
    double param[6], prev[6];
#pragma unroll
    for (int k = 0; k < 6; k++) param[k] = prev[k] = pose[k];
    double U21[21], g6[6];    // normal equations at `prev` (state CALC_J)
    double U21t[21], g6t[6];  // ... at the point just evaluated
    double prev_err2 = 0;  // |err|^2 at `prev`: CvLevMarq's norms are compared on their squares (refine_common.hpp: trial_rejected)
    int lambda_lg10 = -3, iters = 0;
    bool have_base = false;
    for (;;) {
        // residual norm and (speculatively) the normal equations at `param`
        const double err2 = lm_pass<B, MODE>(list, n, param, cam, pm, U21t, g6t, s_part, s_tot, g_cyc, co);
        if (co && co->dead) break;  // a barrier timed out: the sums are garbage, the call reports -12
        CYC_BEGIN();
        CYC_PIN(g6t, 6);
        bool accept;
        if (!have_base) {
            have_base = true;  // iters == 0: prevErrNorm = |err(initial pose)|
            accept = true;
        } else {
            // does this trial, if accepted, end the re-fit?  (a function of the step alone: iteration 20, or
            // cvNorm(param, prevParam, CV_RELATIVE_L2) < eps) -- such a trial is not sent up the lambda ladder over a
            // last-bit difference of the two error norms (refine_common.hpp: trial_rejected)
            double dn = 0, pn = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                dn += (param[k] - prev[k]) * (param[k] - prev[k]);
                pn += prev[k] * prev[k];
            }
            const bool would_end = iters + 1 >= 20 || relative_step_below_eps(dn, pn);
            if (trial_rejected(err2, prev_err2, would_end) && ++lambda_lg10 <= 16) {
                accept = false;  // state CHECK_ERR failed: retry from `prev` with a larger lambda
            } else {
                lambda_lg10 = lambda_lg10 - 1 > -16 ? lambda_lg10 - 1 : -16;
                ++iters;
                if (would_end) break;
                accept = true;
            }
        }
        if (accept) {  // state CALC_J at the accepted point
            prev_err2 = err2;
#pragma unroll
            for (int k = 0; k < 21; k++) U21[k] = U21t[k];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                g6[k] = g6t[k];
                prev[k] = param[k];
            }
        }
        CYC_PIN(g6, 6);
        CYC_PIN(prev, 6);
        CYC_END(16);
        // step(): param = prev - solve(JtJ with diag *= 1 + lambda, JtErr)
        double dx[6];
        CYC_BEGIN();
        const double lambda = s_pow10[lambda_lg10 + 16];  // 10^k from a table in LDS (pow10_int: ~600 cycles of dependent work)
        if (!lm_solve6(U21, g6, lambda, dx)) lm_solve6_pinv(U21, g6, lambda, dx, s_part);
        CYC_PIN(dx, 6);
        CYC_END(8);
#pragma unroll
        for (int k = 0; k < 6; k++) param[k] = prev[k] - dx[k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) pose[k] = param[k];
    return iters;
}

// GLOBAL_LIST: correspondence list in global memory (grids with more than LDS_CAP cells), else in LDS.
// VEC: 16-byte accesses in the error pass (W % 4 == 0 and a 16-byte aligned coordinate tensor).
// SLOTS: training path -- workgroup b refines the hypothesis of selection slot b (esac.cpp:328-347) and leaves
// its refined pose and inlier maps in the BwdArgs buffers instead of picking the winner and writing the record.
// MODE: REFINE_SOLO one workgroup per refinement; REFINE_COOP: gridDim.x workgroups share one (refine_common.hpp, struct
// Coop): workgroup g owns a.coop_slice cells, its correspondences live in its own LDS list (GLOBAL_LIST must be false: a
// slice never exceeds LDS_CAP cells), workgroup 0 writes the outputs.  (REFINE_TEAM, the shared refinement of the small
// grids, is a kernel of its own: esac_refine_team.hip.)
template <int B, bool GLOBAL_LIST, bool VEC, bool SLOTS, int MODE = REFINE_SOLO>
__global__ __launch_bounds__(B) void k_refine(KArgs a) {
    constexpr bool SHARED = MODE != REFINE_SOLO;
    static_assert(!SHARED || (!GLOBAL_LIST && !SLOTS), "cooperating workgroups keep their slices' lists in LDS; winner refinement only");
    static_assert(MODE == REFINE_SOLO || MODE == REFINE_COOP, "esac_refine_team.hip holds the team kernel");
    // ONE allocation, the small arrays first: their addresses then fit the 16-bit offset field of the LDS instructions.
    // (As separate variables they were laid out behind the 128 KB list, and every access to them materialised its own
    // address first.)
    struct Lds {
        double part[MODE == REFINE_COOP ? 256 : 28 * (B / 64)];  // block reductions; the REFINE_COOP gather uses 8 x 32
        double tot[32];
        double pow10[34];  // 10^-16 .. 10^16: the LM damping factors
        double best[B / 64];
        int besti[B / 64];
        int bestg[B / 64];
        int wcnt[B / 64];  // inliers each wavefront put into its region of the list
        int coop_dead;
        int pad_[3];
        Corr list[GLOBAL_LIST ? 1 : LDS_CAP];
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    Corr* const s_list = lds.list;
    double* const s_part = lds.part;
    double* const s_tot = lds.tot;
    double* const s_best = lds.best;
    int* const s_besti = lds.besti;
    int* const s_bestg = lds.bestg;
    int* const s_wcnt = lds.wcnt;
    double* const s_pow10 = lds.pow10;
    static_assert(B == REFINE_B, "corr_region() assumes the refinement workgroup size");
    if (threadIdx.x < 33) s_pow10[threadIdx.x] = pow10_int((int)threadIdx.x - 16);  // (visible after the barriers of the winner pick)
    frame_view(a);
    const int P = a.H * a.W;
    const Cam cam = make_cam(a);
    const PxMap pm{a.sub, a.sub / 2 - a.shift_x, a.sub / 2 - a.shift_y};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long g_cyc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)g_cyc;
    CYC_DECL;
#ifdef ESAC_PROFILE_CYCLES
    const long long cyc_start = clock64();
#endif
    CYC_BEGIN();
    if (!SLOTS && spec_gate_closed(a, &lds.coop_dead)) return;
    if (!SLOTS) spec_open_chain(a);
    if (SLOTS && (int)blockIdx.x >= a.bwd.n_sel[0]) return;
    if (SLOTS && a.bwd.team_max_slots > 0 && a.bwd.n_sel[0] <= a.bwd.team_max_slots) return;  // few slots: the team launch refines them
    Coop co{1, 0, nullptr, nullptr, nullptr, 0ull, 1, 0L, &lds.coop_dead, false, nullptr, 0ull, false};
    int cell0 = 0, Pn = P;  // this workgroup's cells: [cell0, cell0 + Pn)
    if (SHARED) {
        coop_init(co, a, (int)gridDim.x, (int)blockIdx.x, a.coop_extra ? (1L << 12) : (1L << 25));  // polls: ~seconds; the stall test gives up after ~1 ms
        cell0 = co.g * a.coop_slice;
        Pn = P - cell0 < a.coop_slice ? P - cell0 : a.coop_slice;
    }
    const bool writer = !SHARED || co.g == 0;  // the workgroup that owns the outputs

    // ---- draw(probs, training=false): argmax of the exact scores, first (global) index on ties
    // (spec_mode 2 -- speculative with the selection running beside this kernel: the fp32 argmax of the settled hypotheses; score,
    // probability, entropy and contender count of the record are k_spec_join's to fill in)
    const bool fast_pick = !SLOTS && a.spec_mode == 2;
    const int nc = SLOTS || fast_pick ? 0 : a.n_contenders[0];
    const int picked = SLOTS ? 0 : fast_pick ? spec_pick_fast<B>(a, s_best, s_besti, s_bestg) : refine_pick_winner<B>(a, s_best, s_besti, s_bestg);
    const int win = SLOTS ? a.bwd.sel[blockIdx.x] : picked;
    if (fast_pick && spec_nothing_to_refine(a, win, writer)) return;
    const double win_score = fast_pick ? 0.0 : a.scores[win];
    RecordInputs rec_in{0.0, 0.0, 0ull};
    if (!SLOTS && writer && threadIdx.x < 64) rec_in = fast_pick ? RecordInputs{0.0, 0.0, a.status[0]} : refine_record_inputs(a, win_score);
    const int e = expert_of(a, win);
    const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;

    double pose[6];
#pragma unroll
    for (int k = 0; k < 6; k++) pose[k] = a.hyps[(size_t)win * 6 + k];
    if (!SLOTS && writer)
        for (int i = threadIdx.x; i <= ESAC_MAX_REF_STEPS_K; i += B) a.inlier_counts[i] = -1;
    __syncthreads();
    CYC_END(1);

    // ---- refineHyp (esac_util.h:378-454): one error-pass site, one re-fit site
    Corr* const list = !GLOBAL_LIST ? s_list
                       : SLOTS      ? reinterpret_cast<Corr*>(a.bwd.corr_lists) + (size_t)blockIdx.x * corr_entries(P)
                                    : reinterpret_cast<Corr*>(a.corr_list);
    uint8_t* const maps = SLOTS ? a.bwd.maps + (size_t)blockIdx.x * 2 * P : a.inlier_map;
    const Corr* const my_list = list + (size_t)wave * corr_region(Pn);  // the region this wavefront fills and reads
    int n_wave = 0;
    int accepted = 0, last_inliers = 0, lm_total = 0, map_buf = -1;
    int cur = 0;  // map buffer the next error pass writes
    unsigned best_inliers = 4;
    for (int rstep = 0;; rstep++) {
        // error image of the current pose (reproErrs[hypIdx], esac.cpp:169, then esac_util.h:445-452),
        // this step's inlier set and its compacted correspondence list
        CYC_BEGIN();
        __syncthreads();  // every lane is done reading the list before it is rebuilt
        const int n_inl = error_pass_impl<B, VEC, MODE>(a, mx, P, pose, cam, list, n_wave, maps + (size_t)cur * P, s_wcnt, g_cyc, cell0, Pn,
                                                        SHARED ? &co : nullptr, s_tot, s_part);
        __syncthreads();
        CYC_END(2);
        if (rstep >= a.max_ref_steps) break;  // the reference also evaluates the errors of its last re-fit
        if (!SLOTS && writer && threadIdx.x == 0) a.inlier_counts[rstep] = n_inl;
        if ((unsigned)n_inl <= best_inliers) break;  // converged (esac_util.h:417-419)
        best_inliers = (unsigned)n_inl;
        lm_total += lm_refit<B, MODE>(my_list, n_wave, pose, cam, pm, s_part, s_tot, s_pow10, g_cyc, SHARED ? &co : nullptr);
        accepted++;
        last_inliers = n_inl;
        map_buf = cur;  // inlierMap = this step's set (esac_util.h:440)
        cur ^= 1;
    }

    if (SLOTS) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) a.bwd.ref_hyps[(size_t)win * 6 + k] = pose[k];
            int* mi = a.bwd.map_info + 4 * blockIdx.x;
            mi[0] = map_buf;
            mi[1] = last_inliers;
            mi[2] = accepted;
            mi[3] = lm_total;
        }
        return;
    }
    // ---- pose2trans (esac_util.h:537-548) and the result record
    CYC_BEGIN();
    __syncthreads();  // s_part is free: the record is staged there
    if (threadIdx.x < 64 && writer) {
        refine_write_record(a, rec_in, pose, win, win_score, e, nc, accepted, last_inliers, lm_total, map_buf, MODE, co, 0ull, s_part);
    }
    if (threadIdx.x == 0 && writer) {
#ifdef ESAC_PROFILE_CYCLES
        CYC_END(3);
        g_cyc[0] = clock64() - cyc_start;
        for (int k = 0; k < 24; k++) a.cycles[k] = g_cyc[k];
#endif
    }
}

// Slice of cells per cooperating workgroup (0: the refinement stays in one workgroup): grids beyond the LDS list whose
// rows vectorise, single frames; 8192 cells (four trips of the error pass: 2 trips measured 478 us, 1 trip 604, 4 trips 437
// at 480x640) unless that needs more than 256 workgroups.
int refine_coop_slice(const KArgs& a) {
    const int P = a.H * a.W;
    const bool vec = (a.W & 3) == 0 && (reinterpret_cast<uintptr_t>(a.sc) & 15) == 0;
    // a.coop_max = workgroups of the cooperative kernel this DEVICE holds at once (refine_coop_capacity, queried when
    // the context was created): all of them must be resident for the barrier to complete
    const int gmax = a.coop_max < ESAC_REFINE_COOP_MAX ? a.coop_max : ESAC_REFINE_COOP_MAX;
    if (P <= LDS_CAP || !vec || a.frames != 1 || !a.coop_partials || gmax < 2) return 0;
    constexpr int trip = REFINE_B * ERR_UNROLL;
    int slice = 4 * trip;
    if ((P + slice - 1) / slice > gmax) slice = ((P + gmax - 1) / gmax + trip - 1) / trip * trip;
    return slice <= LDS_CAP ? slice : 0;  // 0: the grid needs more resident workgroups than the device has -> one workgroup, global list
}

// Workgroups of the cooperative refinement kernel the current device can hold at once: CUs x workgroups per CU (1: the
// kernel's 128 KiB correspondence list fills a CU's LDS).  0 when the occupancy query fails.
int refine_coop_capacity() {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_refine<REFINE_B, false, true, false, REFINE_COOP>, REFINE_B, 0) != hipSuccess) return 0;
    const long long cap = (long long)cus * per_cu;
    return cap > ESAC_REFINE_COOP_MAX ? ESAC_REFINE_COOP_MAX : (int)cap;
}

// Every shared launch gets its own tag (process-wide launch number << 20): what its exchange granules and its failure word
// are stamped with, so that nothing has to be cleared between launches.
unsigned long long next_refine_tag() {
    static std::atomic<unsigned long long> seq{0};
    return (seq.fetch_add(1) + 1ull) << 20;
}

unsigned long long launch_refine(const KArgs& a, hipStream_t s) {
    constexpr int B = REFINE_B;
    const bool global_list = a.H * a.W > LDS_CAP;
    // 16-byte accesses: W % 4 == 0 keeps every row, plane (P % 4 == 0) and expert map 16-byte aligned
    const bool vec = (a.W & 3) == 0 && (reinterpret_cast<uintptr_t>(a.sc) & 15) == 0;
    if (!a.solo && refine_team_members(a) > 0) return launch_refine_team(a, s);  // single frames (and small batches) on grids up to 32768 cells
    const int slice = a.solo ? 0 : refine_coop_slice(a);
    if (slice > 0) {
        KArgs b = a;
        b.coop_slice = slice;
        b.coop_tag = next_refine_tag();
        const int G = (a.H * a.W + slice - 1) / slice;
        (void)hipMemsetAsync(a.coop_counter, 0, sizeof(unsigned long long), s);
        hipLaunchKernelGGL((k_refine<B, false, true, false, REFINE_COOP>), dim3(G), dim3(B), 0, s, b);
        return b.coop_tag;
    }
    if (global_list) {
        if (vec) hipLaunchKernelGGL((k_refine<B, true, true, false>), dim3(1, a.frames), dim3(B), 0, s, a);
        else     hipLaunchKernelGGL((k_refine<B, true, false, false>), dim3(1, a.frames), dim3(B), 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((k_refine<B, false, true, false>), dim3(1, a.frames), dim3(B), 0, s, a);
        else     hipLaunchKernelGGL((k_refine<B, false, false, false>), dim3(1, a.frames), dim3(B), 0, s, a);
    }
    return 0ull;
}

// One workgroup per selection slot; slots beyond n_sel (known only on the device) return at once.
void launch_refine_slots(const KArgs& a, hipStream_t s) {
    constexpr int B = REFINE_B;
    const bool global_list = a.H * a.W > LDS_CAP;
    const bool vec = (a.W & 3) == 0 && (reinterpret_cast<uintptr_t>(a.sc) & 15) == 0;
    const dim3 grid(a.N < a.bwd.cap ? a.N : a.bwd.cap);
    if (global_list) {
        if (vec) hipLaunchKernelGGL((k_refine<B, true, true, true>), grid, dim3(B), 0, s, a);
        else     hipLaunchKernelGGL((k_refine<B, true, false, true>), grid, dim3(B), 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((k_refine<B, false, true, true>), grid, dim3(B), 0, s, a);
        else     hipLaunchKernelGGL((k_refine<B, false, false, true>), grid, dim3(B), 0, s, a);
    }
}

}  // namespace esac
