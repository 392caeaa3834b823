// lm_math.hpp -- per-correspondence terms and the parameter-space transform of the
// Levenberg-Marquardt pose re-fit (cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess=true)
// reached from refineHyp, esac_util.h:426-436).
//
// The CPU solver differentiates the projection wrt (rvec, tvec) per point through
// dR/drvec (27 values, ~60 flops per point).  Here the per-point Jacobian is taken
// wrt a camera-frame twist (w, v):  dXc = w x Xc + v   (12 flops per point, two
// structural zeros), the 6x6 normal matrix is reduced in twist space, and mapped
// ONCE per iteration to (rvec, tvec) space with the chain rule
//     (w, v) = M6 (drvec, dtvec),  M6 = [[Mw, 0], [[t]x Mw, I]],  Mw_j = vee(dR/dr_j R^T)
//     JtJ_rt = M6^T JtJ_twist M6,   JtErr_rt = M6^T JtErr_twist
// which is the same matrix the CPU solver builds (J_rt = J_twist M6 exactly), so
// damping (diag *= 1 + lambda), steps and the termination test are unchanged.
// FMA contraction is allowed here: these sums feed an iterative solver whose
// fixed point does not depend on the rounding of individual products.
#pragma once
#include "pose_math.hpp"

namespace esac {

// accumulator layout: 20 upper-triangle entries of the twist normal matrix (entry (3,4) is
// structurally zero and skipped), 6 gradient entries, 1 squared residual.
// lm_accumulate_point / lm_chain / lm_to_rvec_space are the straightforward entry-by-entry route: the kernels take
// the moment route further down, the host tests check one against the other and against a numeric Jacobian.
constexpr int LM_NACC = 27;

template <bool WITH_J>
ESAC_HD void lm_accumulate_point(const double R[9], const double t[3], const Cam& cam, double X, double Y, double Z,
                                 double mx, double my, double* acc) {
#pragma clang fp contract(fast)
    const double Xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    const double Yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    const double Zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double iz = Zc ? fast_rcp(Zc) : 1;  // cvProjectPoints2's guard, no cheirality test
    const double x = Xc * iz, y = Yc * iz;
    const double ex = (x * cam.fx + cam.cx) - mx;
    const double ey = (y * cam.fy + cam.cy) - my;
    if (WITH_J) {
        const double zz = Zc * iz;  // 1 unless Zc == 0
        // u row: (a0 a1 a2 | a3 0 a5), v row: (b0 b1 b2 | 0 b4 b5)
        const double a3 = cam.fx * iz, a5 = -a3 * x;
        const double b4 = cam.fy * iz, b5 = -b4 * y;
        const double a0 = -cam.fx * x * y, a1 = cam.fx * (zz + x * x), a2 = -cam.fx * y;
        const double b0 = -cam.fy * (zz + y * y), b1 = cam.fy * x * y, b2 = cam.fy * x;
        acc[0] += a0 * a0 + b0 * b0;
        acc[1] += a0 * a1 + b0 * b1;
        acc[2] += a0 * a2 + b0 * b2;
        acc[3] += a0 * a3;
        acc[4] += b0 * b4;
        acc[5] += a0 * a5 + b0 * b5;
        acc[6] += a1 * a1 + b1 * b1;
        acc[7] += a1 * a2 + b1 * b2;
        acc[8] += a1 * a3;
        acc[9] += b1 * b4;
        acc[10] += a1 * a5 + b1 * b5;
        acc[11] += a2 * a2 + b2 * b2;
        acc[12] += a2 * a3;
        acc[13] += b2 * b4;
        acc[14] += a2 * a5 + b2 * b5;
        acc[15] += a3 * a3;
        acc[16] += a3 * a5;
        acc[17] += b4 * b4;
        acc[18] += b4 * b5;
        acc[19] += a5 * a5 + b5 * b5;
        acc[20] += a0 * ex + b0 * ey;
        acc[21] += a1 * ex + b1 * ey;
        acc[22] += a2 * ex + b2 * ey;
        acc[23] += a3 * ex;
        acc[24] += b4 * ey;
        acc[25] += a5 * ex + b5 * ey;
    }
    acc[26] += ex * ex + ey * ey;
}

// ---- the per-correspondence work of an LM pass, as kernels use it ------------------------------------------------
// With fx == fy == f (the only camera this path ever builds, esac.cpp:93-97) the twist Jacobian rows are
//     u: f (-xy, 1+x^2, -y | iz, 0, -iz x)      v: f (-(1+y^2), xy, x | 0, iz, -iz y)
// (x, y = normalised image coordinates, iz = 1/Zc), and the 20 normal-matrix sums of lm_accumulate_point collapse to
// sums of MONOMIALS in (x, y, iz) -- several of them to plain sums (sum a2*a0 + b2*b0 = -f^2 sum x, one entry is
// identically zero): 24 moments, 31 fused ops per correspondence instead of 42, and the ten Jacobian entries are
// never formed.  lm_moments_to_acc() maps the reduced moments back to the accumulator layout above.
// (Zc*iz is taken as exactly 1; the reference's `z ? 1/z : 1` guard differs only at Zc == 0.0 exactly.)
constexpr int LM_NMOM = 24;

// Stage 1 -- projection: a DEPENDENT chain per correspondence (transform, Newton reciprocal, residual).
// on[p] == false switches correspondence p off (ragged tail of the list) through a 0/1 WEIGHT the optimiser cannot see
// through: a plain select lets it sink the whole chain under a branch, which splits the software-pipelined loop body.
template <int NP>
struct LmTerms {
    double x[NP], y[NP], iz[NP], ex[NP], ey[NP], w[NP];
};

template <int NP>
ESAC_HD void lm_point_terms(const double R[9], const double t[3], const Cam& cam, const double (&X)[NP],
                            const double (&Y)[NP], const double (&Z)[NP], const double (&mx)[NP],
                            const double (&my)[NP], const bool (&on)[NP], LmTerms<NP>& o) {
#pragma clang fp contract(fast)
    double Xc[NP], Yc[NP], Zc[NP], iz[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        Xc[p] = R[0] * X[p] + R[1] * Y[p] + R[2] * Z[p] + t[0];
        Yc[p] = R[3] * X[p] + R[4] * Y[p] + R[5] * Z[p] + t[1];
        Zc[p] = R[6] * X[p] + R[7] * Y[p] + R[8] * Z[p] + t[2];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // Newton reciprocal, the NP chains interleaved step by step
#pragma unroll
    for (int p = 0; p < NP; p++) iz[p] = __builtin_amdgcn_rcp(Zc[p]);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        double e[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) e[p] = __builtin_fma(-Zc[p], iz[p], 1.0);
#pragma unroll
        for (int p = 0; p < NP; p++) iz[p] = __builtin_fma(iz[p], e[p], iz[p]);
    }
#pragma unroll
    for (int p = 0; p < NP; p++) iz[p] = Zc[p] ? iz[p] : 1;
#else
#pragma unroll
    for (int p = 0; p < NP; p++) iz[p] = Zc[p] ? 1.0 / Zc[p] : 1;
#endif
#pragma unroll
    for (int p = 0; p < NP; p++) {
        o.w[p] = on[p] ? 1.0 : 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(o.w[p]));
#endif
        o.iz[p] = iz[p] * o.w[p];  // iz = 0 makes x, y and every moment below vanish for a switched-off point
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        o.x[p] = Xc[p] * o.iz[p];
        o.y[p] = Yc[p] * o.iz[p];
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        o.ex[p] = ((o.x[p] * cam.fx + cam.cx) - mx[p]) * o.w[p];
        o.ey[p] = ((o.y[p] * cam.fx + cam.cy) - my[p]) * o.w[p];
    }
}

// Stage 2 -- 24 independent accumulator chains (FMA / add INTO the accumulator; the NP contributions of one
// accumulator are dependent, so accumulators sit in the inner position).
template <int NP>
ESAC_HD void lm_accumulate_moments(const LmTerms<NP>& q, double* m) {
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const double x = q.x[p], y = q.y[p], iz = q.iz[p], ex = q.ex[p], ey = q.ey[p], w = q.w[p];
        const double xx = x * x, yy = y * y, xy = x * y;
        const double r2 = xx + yy, ox = 1.0 + xx, oy = 1.0 + yy, qq = 1.0 + r2, q1 = 1.0 + qq;
        const double p1 = x * iz, p2 = y * iz, iz2 = iz * iz;
        const double oxw = ox * w, oyw = oy * w;
        m[0] += x;
        m[1] += y;
        m[2] += r2;
        m[3] += iz2;
        m[4] = __builtin_fma(iz2, x, m[4]);
        m[5] = __builtin_fma(iz2, y, m[5]);
        m[6] = __builtin_fma(iz2, r2, m[6]);
        m[7] += p1;
        m[8] += p2;
        m[9] = __builtin_fma(xy, iz, m[9]);
        m[10] = __builtin_fma(oy, iz, m[10]);
        m[11] = __builtin_fma(ox, iz, m[11]);
        m[12] = __builtin_fma(p2, qq, m[12]);
        m[13] = __builtin_fma(p1, qq, m[13]);
        m[14] = __builtin_fma(xy, q1, m[14]);
        m[15] = __builtin_fma(xy, xy, __builtin_fma(oyw, oy, m[15]));
        m[16] = __builtin_fma(xy, xy, __builtin_fma(oxw, ox, m[16]));
        m[17] = __builtin_fma(xy, ex, __builtin_fma(oy, ey, m[17]));
        m[18] = __builtin_fma(ox, ex, __builtin_fma(xy, ey, m[18]));
        m[19] = __builtin_fma(x, ey, __builtin_fma(-y, ex, m[19]));
        m[20] = __builtin_fma(iz, ex, m[20]);
        m[21] = __builtin_fma(iz, ey, m[21]);
        m[22] = __builtin_fma(p1, ex, __builtin_fma(p2, ey, m[22]));
        m[23] = __builtin_fma(ex, ex, __builtin_fma(ey, ey, m[23]));
    }
}

// reduced moments -> the accumulator layout of lm_accumulate_point (f = focal length)
ESAC_HD void lm_moments_to_acc(const double m[LM_NMOM], double f, double acc[LM_NACC]) {
    const double f2 = f * f;
    acc[0] = f2 * m[15];
    acc[1] = -f2 * m[14];
    acc[2] = -f2 * m[0];
    acc[3] = -f2 * m[9];
    acc[4] = -f2 * m[10];
    acc[5] = f2 * m[12];
    acc[6] = f2 * m[16];
    acc[7] = -f2 * m[1];
    acc[8] = f2 * m[11];
    acc[9] = f2 * m[9];
    acc[10] = -f2 * m[13];
    acc[11] = f2 * m[2];
    acc[12] = -f2 * m[8];
    acc[13] = f2 * m[7];
    acc[14] = 0.0;
    acc[15] = f2 * m[3];
    acc[16] = -f2 * m[4];
    acc[17] = f2 * m[3];
    acc[18] = -f2 * m[5];
    acc[19] = f2 * m[6];
    acc[20] = -f * m[17];
    acc[21] = f * m[18];
    acc[22] = f * m[19];
    acc[23] = f * m[20];
    acc[24] = f * m[21];
    acc[25] = -f * m[22];
    acc[26] = m[23];
}

// twist-space sums -> (rvec,tvec)-space normal equations.  U21: upper triangle row-major, g6.
// chain-rule matrices of the pose (18 values; they are all the transform needs, so dR/dr need not stay live)
struct LmChain {
    double Mw[3][3], K[3][3];
};

ESAC_HD void lm_chain(const double R[9], const double dRdr[27], const double t[3], LmChain& ch) {
#pragma clang fp contract(fast)
    double (&Mw)[3][3] = ch.Mw;
    double (&K)[3][3] = ch.K;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double* D = dRdr + 9 * j;
        Mw[0][j] = D[6] * R[3] + D[7] * R[4] + D[8] * R[5];  // S[2][1]
        Mw[1][j] = D[0] * R[6] + D[1] * R[7] + D[2] * R[8];  // S[0][2]
        Mw[2][j] = D[3] * R[0] + D[4] * R[1] + D[5] * R[2];  // S[1][0]
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {  // K[:,j] = t x Mw[:,j]
        K[0][j] = t[1] * Mw[2][j] - t[2] * Mw[1][j];
        K[1][j] = t[2] * Mw[0][j] - t[0] * Mw[2][j];
        K[2][j] = t[0] * Mw[1][j] - t[1] * Mw[0][j];
    }
}

// R(rvec) and the chain-rule matrices in closed form, without dR/drvec:  Mw_j = vee(dR/dr_j R^T) is column j of the
// LEFT Jacobian of SO(3),   J_l(r) = (s/th) I + (1 - s/th) n n^T + ((1-c)/th) [n]x,   n = r/th
// (same matrices as lm_chain(rodrigues_vec2mat<true>) to rounding, ~40 flops instead of ~400 on the per-pass
// dependent chain every lane walks before it can touch its first correspondence).
// In two halves: the rotation (what a pass needs before its first correspondence) and the chain-rule matrices (what
// only the transform AFTER the pass's reduction needs -- a team computes them while its exchange is in flight).
//
// No square root, no sin / cos for |r| <= sqrt(10) (every pose this path meets): with x = |r|^2
//     R   = I + A(x) [r]x + B(x) [r]x^2,        A = sin(th)/th,  B = (1 - cos(th))/th^2,
//     J_l = A(x) I + B(x) [r]x + C(x) r r^T,    C = (th - sin(th))/th^3 = (1 - A)/x,
// and A, B are entire functions of x: Taylor polynomials of degree 7 at a quarter of the angle (Estrin's scheme) and two
// angle doublings are good to 3.3e-16 absolute on [0, 10] (checked against 80-digit arithmetic; the degree-15 polynomials
// at the full angle: 2.7e-16 -- and 32 constants of two scalar moves each on a section every lane walks, where this has
// 16): ~10 dependent operations where sqrt -> division -> sincos is ~25.  Beyond |r|^2 = 10 the trigonometric route
// (1 / th from v_rsq_f64).
struct LmTrig {
    double rx, ry, rz, x;  // rvec, |rvec|^2
    double A, B;           // sin(th)/th, (1 - cos(th))/th^2
    bool identity;         // |rvec| < DBL_EPSILON
};

// sum_k c[k] y^k, k < 8: pairs, quads (Estrin)
ESAC_HD double lm_estrin8(const double (&c)[8], double y, double y2, double y4) {
#pragma clang fp contract(fast)
    const double p0 = c[0] + c[1] * y, p1 = c[2] + c[3] * y, p2 = c[4] + c[5] * y, p3 = c[6] + c[7] * y;
    return (p0 + p1 * y2) + (p2 + p3 * y2) * y4;
}

ESAC_HD void lm_pose_rotation(const double param[6], double R[9], LmTrig& tg) {
#pragma clang fp contract(fast)
    const double rx = param[0], ry = param[1], rz = param[2];
    const double x = rx * rx + ry * ry + rz * rz;
    tg.rx = rx; tg.ry = ry; tg.rz = rz; tg.x = x;
    tg.identity = x < DBL_EPSILON * DBL_EPSILON;
    const bool series = x <= 10.0;
    if (tg.identity) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        tg.A = 1;
        tg.B = 0.5;
        return;
    }
    double A, B;
    if (series) {
        // at a quarter of the angle (y = x / 16 <= 0.625: degree 7, next term 0.625^8 / 17! = 7e-17), then two angle
        // doublings:  cos(t) = 1 - t^2 B(t),  A(2t) = A(t) cos(t),  B(2t) = A(t)^2 / 2
        const double cA[8] = {1.00000000000000000e+00, -1.66666666666666657e-01, 8.33333333333333322e-03, -1.98412698412698413e-04,
                              2.75573192239858925e-06, -2.50521083854417202e-08, 1.60590438368216133e-10, -7.64716373181981641e-13};
        const double cB[8] = {5.00000000000000000e-01, -4.16666666666666644e-02, 1.38888888888888894e-03, -2.48015873015873016e-05,
                              2.75573192239858883e-07, -2.08767569878681002e-09, 1.14707455977297245e-11, -4.77947733238738525e-14};
        const double y = x * 0.0625, y2 = y * y, y4 = y2 * y2;
        const double A4 = lm_estrin8(cA, y, y2, y4), B4 = lm_estrin8(cB, y, y2, y4);
        const double A2 = A4 * (1. - y * B4), B2 = (0.5 * A4) * A4;
        A = A2 * (1. - (x * 0.25) * B2);
        B = (0.5 * A2) * A2;
    } else {  // (a NaN pose also lands here and stays NaN)
#if defined(__HIP_DEVICE_COMPILE__)
        double itheta = __builtin_amdgcn_rsq(x);
        itheta = itheta * __builtin_fma(-0.5 * x, itheta * itheta, 1.5);
        itheta = itheta * __builtin_fma(-0.5 * x, itheta * itheta, 1.5);
        const double theta = x * itheta;
        double c, s;
        sincos(theta, &s, &c);
#else
        const double theta = sqrt(x), itheta = 1. / theta;
        const double c = cos(theta), s = sin(theta);
#endif
        A = s * itheta;
        B = (1. - c) * (itheta * itheta);
    }
    tg.A = A;
    tg.B = B;
    const double Bx = B * rx, By = B * ry, Bz = B * rz;
    // R = I + A [r]x + B (r r^T - x I)
    const double d = 1. - B * x;
    R[0] = d + Bx * rx;       R[1] = Bx * ry - A * rz;  R[2] = Bx * rz + A * ry;
    R[3] = Bx * ry + A * rz;  R[4] = d + By * ry;       R[5] = By * rz - A * rx;
    R[6] = Bx * rz - A * ry;  R[7] = By * rz + A * rx;  R[8] = d + Bz * rz;
}

ESAC_HD void lm_pose_left_jacobian(const LmTrig& tg, double (&Mw)[3][3]) {
#pragma clang fp contract(fast)
    if (tg.identity) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Mw[i][j] = (i == j) ? 1.0 : 0.0;
        return;
    }
    const double rx = tg.rx, ry = tg.ry, rz = tg.rz, x = tg.x, A = tg.A, B = tg.B;
    // C = (1 - A) / x: the cancellation in 1 - A costs C a relative 1e-16 / (x / 6), but C only ever multiplies r r^T
    // (entries <= x), so J_l keeps an absolute 1e-16 at every x (a third polynomial here was 16 more constants and
    // FMAs on a section every lane walks)
    const double C = (1. - A) * fast_rcp(x);
    const double Cx = C * rx, Cy = C * ry, Cz = C * rz;
    // J_l = A I + B [r]x + C r r^T
    Mw[0][0] = A + Cx * rx;       Mw[0][1] = Cx * ry - B * rz;  Mw[0][2] = Cx * rz + B * ry;
    Mw[1][0] = Cx * ry + B * rz;  Mw[1][1] = A + Cy * ry;       Mw[1][2] = Cy * rz - B * rx;
    Mw[2][0] = Cx * rz - B * ry;  Mw[2][1] = Cy * rz + B * rx;  Mw[2][2] = A + Cz * rz;
}

ESAC_HD void lm_pose_chain_rest(const LmTrig& tg, const double t[3], LmChain& ch) {
#pragma clang fp contract(fast)
    double (&Mw)[3][3] = ch.Mw;
    double (&K)[3][3] = ch.K;
    lm_pose_left_jacobian(tg, Mw);
#pragma unroll
    for (int j = 0; j < 3; j++) {  // K[:,j] = t x Mw[:,j]
        K[0][j] = t[1] * Mw[2][j] - t[2] * Mw[1][j];
        K[1][j] = t[2] * Mw[0][j] - t[0] * Mw[2][j];
        K[2][j] = t[0] * Mw[1][j] - t[1] * Mw[0][j];
    }
}

ESAC_HD void lm_pose_chain(const double param[6], double R[9], LmChain& ch) {
    LmTrig tg;
    lm_pose_rotation(param, R, tg);
    lm_pose_chain_rest(tg, param + 3, ch);
}

ESAC_HD void lm_transform(const double acc[LM_NACC], const LmChain& ch, double U21[21], double g6[6]);

ESAC_HD void lm_to_rvec_space(const double acc[LM_NACC], const double R[9], const double dRdr[27], const double t[3],
                              double U21[21], double g6[6]) {
    LmChain ch;
    lm_chain(R, dRdr, t, ch);
    lm_transform(acc, ch, U21, g6);
}

ESAC_HD void lm_transform(const double acc[LM_NACC], const LmChain& ch, double U21[21], double g6[6]) {
#pragma clang fp contract(fast)
    const double (&Mw)[3][3] = ch.Mw;
    const double (&K)[3][3] = ch.K;
    // Mw[i][j] = vee(dR/dr_j * R^T)[i]
    const double Aww[3][3] = {{acc[0], acc[1], acc[2]}, {acc[1], acc[6], acc[7]}, {acc[2], acc[7], acc[11]}};
    const double Awv[3][3] = {{acc[3], acc[4], acc[5]}, {acc[8], acc[9], acc[10]}, {acc[12], acc[13], acc[14]}};
    const double Avv[3][3] = {{acc[15], 0.0, acc[16]}, {0.0, acc[17], acc[18]}, {acc[16], acc[18], acc[19]}};
    double T1[3][3], T2[3][3];  // T1 = Aww Mw + Awv K ; T2 = Awv^T Mw + Avv K
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s1 += Aww[i][k] * Mw[k][j] + Awv[i][k] * K[k][j];
                s2 += Awv[k][i] * Mw[k][j] + Avv[i][k] * K[k][j];
            }
            T1[i][j] = s1;
            T2[i][j] = s2;
        }
    double rr[3][3], rt[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                s1 += Mw[k][i] * T1[k][j] + K[k][i] * T2[k][j];
                s2 += Mw[k][i] * Awv[k][j] + K[k][i] * Avv[k][j];
            }
            rr[i][j] = s1;
            rt[i][j] = s2;
        }
    int n = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = i; j < 3; j++) U21[n++] = 0.5 * (rr[i][j] + rr[j][i]);
#pragma unroll
        for (int j = 0; j < 3; j++) U21[n++] = rt[i][j];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) U21[n++] = Avv[i][j];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        g6[i] = Mw[0][i] * acc[20] + Mw[1][i] * acc[21] + Mw[2][i] * acc[22] + K[0][i] * acc[23] + K[1][i] * acc[24] +
                K[2][i] * acc[25];
        g6[3 + i] = acc[23 + i];
    }
}

// The same map with K = [t]x Mw folded in (a team's passes, esac_refine_team.hip): with Tx = [t]x
//     U_rr = Mw^T B Mw,   B = Aww + E Tx + (Awv Tx)^T,   E = Awv + Tx^T Avv;     U_rt = Mw^T E;     U_tt = Avv
//     g_r  = Mw^T (g_w + g_v x t),   g_t = g_v
// -- products with Tx are two terms per entry, B is symmetric (six entries): ~130 fused ops instead of ~235 on the serial
// section every lane walks between a pass and the next step.  Equal to lm_transform to rounding (host test).
ESAC_HD void lm_transform_t(const double acc[LM_NACC], const double (&Mw)[3][3], const double t[3], double U21[21], double g6[6]) {
#pragma clang fp contract(fast)
    const double t0 = t[0], t1 = t[1], t2 = t[2];
    const double Aww[3][3] = {{acc[0], acc[1], acc[2]}, {acc[1], acc[6], acc[7]}, {acc[2], acc[7], acc[11]}};
    const double Awv[3][3] = {{acc[3], acc[4], acc[5]}, {acc[8], acc[9], acc[10]}, {acc[12], acc[13], acc[14]}};
    const double Avv[3][3] = {{acc[15], 0.0, acc[16]}, {0.0, acc[17], acc[18]}, {acc[16], acc[18], acc[19]}};
    // (X Tx)[i][:] = (X[i][1] t2 - X[i][2] t1,  X[i][2] t0 - X[i][0] t2,  X[i][0] t1 - X[i][1] t0)   (row x t, negated: -(t x row))
    // (Tx^T Y)[:][j] = -(t x Y[:][j]) = Y[:][j] x t
    // (E = Awv + Tx^T Avv written out: Avv[0][1] = Avv[1][0] = 0 and Awv[2][2] = acc[14] = 0 are structural zeros, and a
    // product with a literal 0.0 is an instruction the compiler may not drop)
    double E[3][3];
    E[0][0] = Awv[0][0] - Avv[2][0] * t1;
    E[1][0] = Awv[1][0] + (Avv[2][0] * t0 - Avv[0][0] * t2);
    E[2][0] = Awv[2][0] + Avv[0][0] * t1;
    E[0][1] = Awv[0][1] + (Avv[1][1] * t2 - Avv[2][1] * t1);
    E[1][1] = Awv[1][1] + Avv[2][1] * t0;
    E[2][1] = Awv[2][1] - Avv[1][1] * t0;
    E[0][2] = Awv[0][2] + (Avv[1][2] * t2 - Avv[2][2] * t1);
    E[1][2] = Awv[1][2] + (Avv[2][2] * t0 - Avv[0][2] * t2);
    E[2][2] = Avv[0][2] * t1 - Avv[1][2] * t0;
    double ET[3][3], AT[3][3];  // E Tx, Awv Tx
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ET[i][0] = E[i][1] * t2 - E[i][2] * t1;
        ET[i][1] = E[i][2] * t0 - E[i][0] * t2;
        ET[i][2] = E[i][0] * t1 - E[i][1] * t0;
        AT[i][2] = Awv[i][0] * t1 - Awv[i][1] * t0;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        AT[i][0] = Awv[i][1] * t2 - Awv[i][2] * t1;
        AT[i][1] = Awv[i][2] * t0 - Awv[i][0] * t2;
    }
    AT[2][0] = Awv[2][1] * t2;
    AT[2][1] = -(Awv[2][0] * t2);
    double B[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            // B is symmetric analytically; one of the two roundings serves both entries
            const double bij = Aww[i][j] + (ET[i][j] + AT[j][i]);
            B[i][j] = bij;
            B[j][i] = bij;
        }
    double C[3][3];  // B Mw
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i][j] = B[i][0] * Mw[0][j] + B[i][1] * Mw[1][j] + B[i][2] * Mw[2][j];
    int n = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = i; j < 3; j++) U21[n++] = Mw[0][i] * C[0][j] + Mw[1][i] * C[1][j] + Mw[2][i] * C[2][j];
#pragma unroll
        for (int j = 0; j < 3; j++) U21[n++] = Mw[0][i] * E[0][j] + Mw[1][i] * E[1][j] + Mw[2][i] * E[2][j];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) U21[n++] = Avv[i][j];
    const double gw[3] = {acc[20] + (acc[24] * t2 - acc[25] * t1), acc[21] + (acc[25] * t0 - acc[23] * t2), acc[22] + (acc[23] * t1 - acc[24] * t0)};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        g6[i] = Mw[0][i] * gw[0] + Mw[1][i] * gw[1] + Mw[2][i] * gw[2];
        g6[3 + i] = acc[23 + i];
    }
}

}  // namespace esac
