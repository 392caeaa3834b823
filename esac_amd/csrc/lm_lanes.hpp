// lm_lanes.hpp -- the serial section of an LM round (normal equations twist space -> (rvec, tvec) space, damping, 6x6 solve)
// DEALT TO THE LANES of a 16-lane DPP row (round 5).
//
// Reference: cv::solvePnP(SOLVEPNP_ITERATIVE) reached from refineHyp, esac_util.h:426-436 -- CvLevMarq's
// (JtJ with diag *= 1 + lambda) dx = JtErr of the 6 pose parameters.  lm_math.hpp:lm_transform / pose_math.hpp:lm_solve6
// are the same map with every lane of every wavefront computing all 27 outputs and the whole LDL^T alike: ~330
// instructions a round on a kernel (esac_refine_team.hip) whose wavefronts run alone on their SIMD and are bound by
// instruction ISSUE (scripts/dev/lat_probe.hip: a lone wavefront issues one VALU instruction per ~4.8 cycles; a dependent
// fp64 operation waits 10).  Here lane j of a row holds COLUMN j of a small matrix and products run on
//     v_fmac_f64_dpp  d, s0, s1  row_newbcast:k        d += s0[lane k of this row] * s1
// -- the one DPP form a double-precision ALU operation has on gfx9 (VOP2 + row_newbcast), same issue cost and latency as a
// plain FMA (measured) -- so a 3x3-block product is 6 instructions for a whole row of outputs and one Gauss-Jordan
// elimination step is 5.  All four rows of a wavefront (and all wavefronts, and all members of a team) compute the same
// values from the same totals; nothing is exchanged.
//
// Layout (lane = lane within the row; lanes 7-15 carry zeros / values nobody reads):
//     M_k, K_k  (k < 3)   lane j < 3: Mw[k][j], ([t]x Mw)[k][j]        chain-rule matrices of the pose, column j
//     X_k, Y_k  (k < 3)   lane j < 3: Aww[k][j], Awv[j][k];  lane 3+j: Awv[k][j], Avv[k][j];  lane 6: g_w[k], g_v[k]
//                         (twist-space sums gathered per lane from the totals in LDS, lm_lane_slot)
//     Z_i = sum_k M_k[i] X_k + K_k[i] Y_k     lane j < 3: (Mw^T Aww + K^T Awv^T)[i][j]; lane 3+j: U_rt[i][j]; lane 6: g_r[i]
//     c_i (i < 3)  = Z_i (lanes 3-6) | sum_k Z_i[k] M_k + Z_i[3+k] K_k (lanes 0-2)  = row i of the (rvec,tvec)-space system
//     c_3+m        = Y_m (lanes 3-6) | Z_j[3+m] in lane j < 3                        = row 3+m   (column 6: the right-hand side)
// then a_k = c_k with the diagonal * (1 + lambda), Gauss-Jordan over the 6 pivots in place, dx_k = a_k[6] / pivot_k.
// Equal to lm_transform + lm_solve6 to rounding (tests/test_device_math_host.py runs this very source on a 16-lane
// emulation of the row; the GPU tests run the device build).
#pragma once
#include "lm_math.hpp"

namespace esac {

// ---- where lane `lane` of a row finds X_k (which = k) / Y_k (which = 3 + k) among the 64 doubles the exchange leaves in
// LDS: [0, 27) the totals (24 moments | ...), [27, 32) zeros, [32, 59) the NEGATED totals, [59, 64) zeros.
// lm_moments_to_acc with f = 1 (the normal equations in units of the focal length) written as signed moment indices.
constexpr int LM_LANE_ZERO_SLOT = 27;
constexpr int LM_LANE_NEG = 32;   // slot of -total[k]: LM_LANE_NEG + k
constexpr int LM_LANE_SLOTS = 64;
ESAC_HD constexpr int lm_lane_slot(int lane, int which) {
    // accumulator layout entry a -> signed moment: +m: m, -m: 32 + m, structural zero: LM_LANE_ZERO_SLOT
    constexpr int Zs = LM_LANE_ZERO_SLOT;
    const int acc[26] = {15, 32 + 14, 32 + 0, 32 + 9, 32 + 10, 12, 16, 32 + 1, 11, 9, 32 + 13, 2, 32 + 8, 7, Zs, 3, 32 + 4, 3, 32 + 5, 6,
                         32 + 17, 18, 19, 20, 21, 32 + 22};
    const int Aww[3][3] = {{0, 1, 2}, {1, 6, 7}, {2, 7, 11}};
    const int Awv[3][3] = {{3, 4, 5}, {8, 9, 10}, {12, 13, 14}};
    const int Avv[3][3] = {{15, -1, 16}, {-1, 17, 18}, {16, 18, 19}};
    const int k = which % 3;
    const bool y = which >= 3;
    int a = -1;
    if (lane < 3) a = y ? Awv[lane][k] : Aww[k][lane];
    else if (lane < 6) a = y ? Avv[k][lane - 3] : Awv[k][lane - 3];
    else if (lane == 6) a = (y ? 23 : 20) + k;
    return a < 0 ? Zs : acc[a];
}

// ---- host emulation of one DPP row: the generic code below runs on it in the CPU tests
struct Row16 {
    double l[16];
    Row16() = default;
    ESAC_HD Row16(double v) {
        for (int i = 0; i < 16; i++) l[i] = v;
    }
};
ESAC_HD Row16 operator*(const Row16& a, const Row16& b) {
    Row16 r;
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] * b.l[i];
    return r;
}
ESAC_HD Row16 operator-(const Row16& a, const Row16& b) {
    Row16 r;
    for (int i = 0; i < 16; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
ESAC_HD Row16 operator-(const Row16& a) {
    Row16 r;
    for (int i = 0; i < 16; i++) r.l[i] = -a.l[i];
    return r;
}
ESAC_HD Row16 lfma(const Row16& a, const Row16& b, const Row16& c) {
    Row16 r;
    for (int i = 0; i < 16; i++) r.l[i] = __builtin_fma(a.l[i], b.l[i], c.l[i]);
    return r;
}
template <int K>
ESAC_HD Row16 lane_bc(const Row16& x) {
    return Row16(x.l[K]);
}
template <int K>
ESAC_HD void lane_fmac_bc(Row16& d, const Row16& s0, const Row16& s1) {
    const double b = s0.l[K];
    for (int i = 0; i < 16; i++) d.l[i] = __builtin_fma(b, s1.l[i], d.l[i]);
}
ESAC_HD Row16 lane_rcp_neg(const Row16& p) {
    Row16 r;
    for (int i = 0; i < 16; i++) r.l[i] = -1.0 / p.l[i];
    return r;
}
ESAC_HD bool lane_gt_lanes6(const Row16& a, const Row16& b) {
    bool ok = true;
    for (int i = 0; i < 6; i++) ok &= a.l[i] > b.l[i];
    return ok;
}
template <int I>
ESAC_HD Row16 lane_zrow(const Row16 (&M)[3], const Row16 (&K)[3], const Row16 (&X)[3], const Row16 (&Y)[3]) {
    Row16 z(0.0);
    for (int k = 0; k < 3; k++) lane_fmac_bc<I>(z, M[k], X[k]);
    for (int k = 0; k < 3; k++) lane_fmac_bc<I>(z, K[k], Y[k]);
    return z;
}
ESAC_HD Row16 lane_urow(const Row16& z, const Row16& keep, const Row16 (&M)[3], const Row16 (&K)[3]) {
    Row16 c = z * keep;
    lane_fmac_bc<0>(c, z, M[0]);
    lane_fmac_bc<1>(c, z, M[1]);
    lane_fmac_bc<2>(c, z, M[2]);
    lane_fmac_bc<3>(c, z, K[0]);
    lane_fmac_bc<4>(c, z, K[1]);
    lane_fmac_bc<5>(c, z, K[2]);
    return c;
}
template <int Mi>
ESAC_HD Row16 lane_trow(const Row16& y, const Row16& keep, const Row16 (&Z)[3], const Row16 (&hot)[6]) {
    Row16 c = y * keep;
    lane_fmac_bc<3 + Mi>(c, Z[0], hot[0]);
    lane_fmac_bc<3 + Mi>(c, Z[1], hot[1]);
    lane_fmac_bc<3 + Mi>(c, Z[2], hot[2]);
    return c;
}
template <int K>
ESAC_HD void lane_gj_step(Row16 (&a)[6], const Row16& nrow) {
    for (int i = 0; i < 6; i++)
        if (i != K) lane_fmac_bc<K>(a[i], a[i], nrow);
}

// ---- the device's row: one double per lane
ESAC_HD double lfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#if defined(__HIP_DEVICE_COMPILE__)
// `old` is never seen (row_newbcast gives every lane a source): the empty asm "defines" it without an instruction
// (a frozen undef is materialised as v_mov_b64 0)
template <int K>
__device__ __forceinline__ double lane_bc(double x) {
    double u;
    asm("" : "=v"(u));
    return __builtin_amdgcn_update_dpp(u, x, 0x150 + K, 0xf, 0xf, false);
}
// -1 / p, correctly rounded: v_rcp_f64 is good to 2^-24.4, x0 (1 + e + e^2) with e = 1 - p x0 then equals the rounded quotient
// on every input tried (scripts/dev/lat_probe.hip, 2^20 inputs: as two Newton steps, one dependent operation less)
__device__ __forceinline__ double lane_rcp_neg(double p) {
    const double x0 = __builtin_amdgcn_rcp(p);
    const double e = __builtin_fma(-p, x0, 1.0);
    const double e2 = __builtin_fma(e, e, e);
    return __builtin_fma(-x0, e2, -x0);
}
__device__ __forceinline__ bool lane_gt_lanes6(double a, double b) {  // a > b in lanes 0..5 (of row 0; every row holds the same)
    return (__ballot(a > b) & 0x3full) == 0x3full;                    // (NaN: false)
}
// The blocks below are inline assembly because the compiler does not fold a DPP move into v_fmac_f64 (it emits
// v_mov_b64_dpp + v_fmac_f64: two issue slots per term).  Hazards the assembler's recogniser cannot see inside an asm
// statement, handled by hand: a VALU write of a VGPR needs 2 wait states before a DPP read of it (every block starts with
// s_nop 1 unless its DPP sources are as old as the block before it; inside a block no DPP source is written), an SALU write
// of EXEC needs 5 before any DPP operation (s_nop 4 in the first block of the transform and of the solve: either may be the
// first thing behind a branch).  Not volatile: pure functions of their operands.
#define LANE_DPP(k) " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
template <int I>
__device__ __forceinline__ double lane_zrow(const double (&M)[3], const double (&K)[3], const double (&X)[3], const double (&Y)[3]) {
    double z;
    static_assert(I >= 0 && I < 3, "row of Z");
#define LANE_ZROW(i, nop)                                                                                                      \
    asm(nop "v_mov_b64 %0, 0\n\t"                                                                                             \
        "v_fmac_f64_dpp %0, %1, %7" LANE_DPP(i) "v_fmac_f64_dpp %0, %2, %8" LANE_DPP(i) "v_fmac_f64_dpp %0, %3, %9" LANE_DPP(i)   \
        "v_fmac_f64_dpp %0, %4, %10" LANE_DPP(i) "v_fmac_f64_dpp %0, %5, %11" LANE_DPP(i) "v_fmac_f64_dpp %0, %6, %12" LANE_DPP(i) \
        : "=&v"(z)                                                                                                         \
        : "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(Y[0]), "v"(Y[1]), "v"(Y[2]))
    if (I == 0) LANE_ZROW(0, "s_nop 4\n\t");
    if (I == 1) LANE_ZROW(1, "");
    if (I == 2) LANE_ZROW(2, "");
#undef LANE_ZROW
    return z;
}
__device__ __forceinline__ double lane_urow(double z, double keep, const double (&M)[3], const double (&K)[3]) {
    double c = z * keep;
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %1, %2" LANE_DPP(0) "v_fmac_f64_dpp %0, %1, %3" LANE_DPP(1) "v_fmac_f64_dpp %0, %1, %4" LANE_DPP(2)
        "v_fmac_f64_dpp %0, %1, %5" LANE_DPP(3) "v_fmac_f64_dpp %0, %1, %6" LANE_DPP(4) "v_fmac_f64_dpp %0, %1, %7" LANE_DPP(5)
        : "+v"(c)
        : "v"(z), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(K[0]), "v"(K[1]), "v"(K[2]));
    return c;
}
template <int Mi>
__device__ __forceinline__ double lane_trow(double y, double keep, const double (&Z)[3], const double (&hot)[6]) {
    double c = y * keep;
    static_assert(Mi >= 0 && Mi < 3, "row 3 + Mi");
#define LANE_TROW(k)                                                                                                     \
    asm("s_nop 1\n\t"                                                                                                    \
        "v_fmac_f64_dpp %0, %1, %4" LANE_DPP(k) "v_fmac_f64_dpp %0, %2, %5" LANE_DPP(k) "v_fmac_f64_dpp %0, %3, %6" LANE_DPP(k) \
        : "+v"(c)                                                                                                        \
        : "v"(Z[0]), "v"(Z[1]), "v"(Z[2]), "v"(hot[0]), "v"(hot[1]), "v"(hot[2]))
    if (Mi == 0) LANE_TROW(3);
    if (Mi == 1) LANE_TROW(4);
    if (Mi == 2) LANE_TROW(5);
#undef LANE_TROW
    return c;
}
// a_i += a_i[lane K] * nrow for the five rows i != K
template <int K>
__device__ __forceinline__ void lane_gj_step(double (&a)[6], double nrow) {
    static_assert(K >= 0 && K < 6, "pivot");
    constexpr int r0 = K > 0 ? 0 : 1, r1 = K > 1 ? 1 : 2, r2 = K > 2 ? 2 : 3, r3 = K > 3 ? 3 : 4, r4 = K > 4 ? 4 : 5;
#define LANE_GJ(k, nop)                                                                                                      \
    asm(nop                                                                                                               \
        "v_fmac_f64_dpp %0, %0, %5" LANE_DPP(k) "v_fmac_f64_dpp %1, %1, %5" LANE_DPP(k) "v_fmac_f64_dpp %2, %2, %5" LANE_DPP(k) \
        "v_fmac_f64_dpp %3, %3, %5" LANE_DPP(k) "v_fmac_f64_dpp %4, %4, %5" LANE_DPP(k)                                      \
        : "+v"(a[r0]), "+v"(a[r1]), "+v"(a[r2]), "+v"(a[r3]), "+v"(a[r4])                                                   \
        : "v"(nrow))
    if (K == 0) LANE_GJ(0, "s_nop 4\n\t");
    if (K == 1) LANE_GJ(1, "s_nop 1\n\t");
    if (K == 2) LANE_GJ(2, "s_nop 1\n\t");
    if (K == 3) LANE_GJ(3, "s_nop 1\n\t");
    if (K == 4) LANE_GJ(4, "s_nop 1\n\t");
    if (K == 5) LANE_GJ(5, "s_nop 1\n\t");
#undef LANE_GJ
}
#undef LANE_DPP
#else
// host pass of the kernels' translation units: the device overloads must exist for the kernels to parse; never run
template <int K>
ESAC_HD double lane_bc(double x) { return x; }
ESAC_HD double lane_rcp_neg(double p) { return -1.0 / p; }
ESAC_HD bool lane_gt_lanes6(double a, double b) { return a > b; }
template <int I>
ESAC_HD double lane_zrow(const double (&)[3], const double (&)[3], const double (&)[3], const double (&)[3]) { return 0.0; }
ESAC_HD double lane_urow(double z, double, const double (&)[3], const double (&)[3]) { return z; }
template <int Mi>
ESAC_HD double lane_trow(double y, double, const double (&)[3], const double (&)[6]) { return y; }
template <int K>
ESAC_HD void lane_gj_step(double (&)[6], double) {}
#endif

// ---- chain-rule matrices of the pose, column j in lane j < 3 (zero elsewhere: hot[] is): Mw = J_l(r) = A I + B [r]x + C r r^T
// (lm_pose_left_jacobian), K = [t]x Mw.  hot[k]: 1 in lane k of the row, 0 elsewhere.
template <class T>
ESAC_HD void lm_lane_chain(const LmTrig& tg, const double t[3], const T (&hot)[6], T (&M)[3], T (&K)[3]) {
    if (tg.identity) {
        M[0] = hot[0];
        M[1] = hot[1];
        M[2] = hot[2];
    } else {
        const T rx(tg.rx), ry(tg.ry), rz(tg.rz), A(tg.A), B(tg.B);
        const T C((1. - tg.A) * fast_rcp(tg.x));
        const T rj = lfma(rx, hot[0], lfma(ry, hot[1], rz * hot[2]));  // r_j
        const T crj = C * rj;
        const T c0 = lfma(ry, hot[2], -(rz * hot[1])), c1 = lfma(rz, hot[0], -(rx * hot[2])), c2 = lfma(rx, hot[1], -(ry * hot[0]));  // r x e_j
        M[0] = lfma(A, hot[0], lfma(B, c0, crj * rx));
        M[1] = lfma(A, hot[1], lfma(B, c1, crj * ry));
        M[2] = lfma(A, hot[2], lfma(B, c2, crj * rz));
    }
    const T t0(t[0]), t1(t[1]), t2(t[2]);
    K[0] = lfma(t1, M[2], -(t2 * M[1]));  // t x (column of Mw)
    K[1] = lfma(t2, M[0], -(t0 * M[2]));
    K[2] = lfma(t0, M[1], -(t1 * M[0]));
}

// ---- twist-space sums (gathered X, Y) -> the rows of the (rvec, tvec)-space system, see the header
// dg: the diagonal, entry (j, j) in lane j < 6 (what the solve's pivot test compares against)
template <class T>
ESAC_HD void lm_lane_transform(const T (&X)[3], const T (&Y)[3], const T (&M)[3], const T (&K)[3], const T (&hot)[6], const T& keep, T (&c)[6], T& dg) {
    T Z[3];
    Z[0] = lane_zrow<0>(M, K, X, Y);
    Z[1] = lane_zrow<1>(M, K, X, Y);
    Z[2] = lane_zrow<2>(M, K, X, Y);
    c[0] = lane_urow(Z[0], keep, M, K);
    c[1] = lane_urow(Z[1], keep, M, K);
    c[2] = lane_urow(Z[2], keep, M, K);
    c[3] = lane_trow<0>(Y[0], keep, Z, hot);
    c[4] = lane_trow<1>(Y[1], keep, Z, hot);
    c[5] = lane_trow<2>(Y[2], keep, Z, hot);
    dg = c[0] * hot[0];
#pragma unroll
    for (int k = 1; k < 6; k++) dg = lfma(c[k], hot[k], dg);
}

// ---- (A with diag *= 1 + lambda) dx = g by Gauss-Jordan without pivoting over the columns in the lanes (the matrix is SPD
// for any non-degenerate inlier set).  Returns false -- dx is then meaningless -- when a pivot falls below 1e-12 of its
// (damped) diagonal entry, as lm_solve6: the caller takes the pseudo-inverse route (lm_solve6_pinv), as the CPU library
// always does.  The six pivots are tested together at the end: row k keeps its pivot in lane k (later steps add
// multiples of entries the elimination has already cancelled).  dx comes back uniform (every lane of the row).
template <int K, class T>
ESAC_HD void lm_lane_pivot(T (&a)[6], T& inv_k) {
    const T p = lane_bc<K>(a[K]);
    const T ninv = lane_rcp_neg(p);
    inv_k = -ninv;
    lane_gj_step<K>(a, a[K] * ninv);
}
template <class T>
ESAC_HD bool lm_lane_solve(const T (&c)[6], const T& dg, const T (&hot)[6], double lambda, T (&dx)[6]) {
    T a[6], inv[6];
    const T lam(lambda);
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = lfma(c[k], lam * hot[k], c[k]);
    lm_lane_pivot<0>(a, inv[0]);
    lm_lane_pivot<1>(a, inv[1]);
    lm_lane_pivot<2>(a, inv[2]);
    lm_lane_pivot<3>(a, inv[3]);
    lm_lane_pivot<4>(a, inv[4]);
    lm_lane_pivot<5>(a, inv[5]);
    T own = a[0] * hot[0];
#pragma unroll
    for (int k = 1; k < 6; k++) own = lfma(a[k], hot[k], own);
    const bool ok = lane_gt_lanes6(own, T(1e-12 * (1. + lambda)) * dg);
#pragma unroll
    for (int k = 0; k < 6; k++) dx[k] = lane_bc<6>(a[k] * inv[k]);
    return ok;
}

// the system back in the layout lm_solve6 / lm_solve6_pinv take (the rare pseudo-inverse step): every value uniform
template <class T>
ESAC_HD void lm_lane_to_u21(const T (&c)[6], T (&U21)[21], T (&g6)[6]) {
    U21[0] = lane_bc<0>(c[0]); U21[1] = lane_bc<1>(c[0]); U21[2] = lane_bc<2>(c[0]); U21[3] = lane_bc<3>(c[0]); U21[4] = lane_bc<4>(c[0]); U21[5] = lane_bc<5>(c[0]);
    U21[6] = lane_bc<1>(c[1]); U21[7] = lane_bc<2>(c[1]); U21[8] = lane_bc<3>(c[1]); U21[9] = lane_bc<4>(c[1]); U21[10] = lane_bc<5>(c[1]);
    U21[11] = lane_bc<2>(c[2]); U21[12] = lane_bc<3>(c[2]); U21[13] = lane_bc<4>(c[2]); U21[14] = lane_bc<5>(c[2]);
    U21[15] = lane_bc<3>(c[3]); U21[16] = lane_bc<4>(c[3]); U21[17] = lane_bc<5>(c[3]);
    U21[18] = lane_bc<4>(c[4]); U21[19] = lane_bc<5>(c[4]);
    U21[20] = lane_bc<5>(c[5]);
#pragma unroll
    for (int k = 0; k < 6; k++) g6[k] = lane_bc<6>(c[k]);
}

}  // namespace esac
