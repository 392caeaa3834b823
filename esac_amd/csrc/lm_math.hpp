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
// structurally zero and skipped), 6 gradient entries, 1 squared residual
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

// NP correspondences at once, stage by stage, so that their independent fp64 chains are adjacent
// in the (in-order) instruction stream; per element the same arithmetic as lm_accumulate_point<true>.
// MASKED: wgt[p] in {0,1} switches correspondence p off (ragged tail of the list).
template <int NP, bool MASKED>
ESAC_HD void lm_accumulate_points(const double R[9], const double t[3], const Cam& cam, const double (&X)[NP],
                                  const double (&Y)[NP], const double (&Z)[NP], const double (&mx)[NP],
                                  const double (&my)[NP], const double (&wgt)[NP], double* acc) {
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
    double x[NP], y[NP], ex[NP], ey[NP], zz[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        x[p] = Xc[p] * iz[p];
        y[p] = Yc[p] * iz[p];
        zz[p] = Zc[p] * iz[p];
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        ex[p] = (x[p] * cam.fx + cam.cx) - mx[p];
        ey[p] = (y[p] * cam.fy + cam.cy) - my[p];
    }
    double a0[NP], a1[NP], a2[NP], a3[NP], a5[NP], b0[NP], b1[NP], b2[NP], b4[NP], b5[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        a3[p] = cam.fx * iz[p];
        b4[p] = cam.fy * iz[p];
        a0[p] = -cam.fx * x[p] * y[p];
        a1[p] = cam.fx * (zz[p] + x[p] * x[p]);
        a2[p] = -cam.fx * y[p];
        b0[p] = -cam.fy * (zz[p] + y[p] * y[p]);
        b1[p] = cam.fy * x[p] * y[p];
        b2[p] = cam.fy * x[p];
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        a5[p] = -a3[p] * x[p];
        b5[p] = -b4[p] * y[p];
    }
    if (MASKED) {  // every accumulated term is a product of two of these, so scaling them by 0 removes the point
#pragma unroll
        for (int p = 0; p < NP; p++) {
            a0[p] *= wgt[p]; a1[p] *= wgt[p]; a2[p] *= wgt[p]; a3[p] *= wgt[p]; a5[p] *= wgt[p];
            b0[p] *= wgt[p]; b1[p] *= wgt[p]; b2[p] *= wgt[p]; b4[p] *= wgt[p]; b5[p] *= wgt[p];
            ex[p] *= wgt[p]; ey[p] *= wgt[p];
        }
    }
    // 27 independent accumulator chains; the NP contributions of one accumulator are dependent, so
    // iterate accumulators in the inner position
#pragma unroll
    for (int p = 0; p < NP; p++) {
        acc[0] += a0[p] * a0[p] + b0[p] * b0[p];
        acc[1] += a0[p] * a1[p] + b0[p] * b1[p];
        acc[2] += a0[p] * a2[p] + b0[p] * b2[p];
        acc[3] += a0[p] * a3[p];
        acc[4] += b0[p] * b4[p];
        acc[5] += a0[p] * a5[p] + b0[p] * b5[p];
        acc[6] += a1[p] * a1[p] + b1[p] * b1[p];
        acc[7] += a1[p] * a2[p] + b1[p] * b2[p];
        acc[8] += a1[p] * a3[p];
        acc[9] += b1[p] * b4[p];
        acc[10] += a1[p] * a5[p] + b1[p] * b5[p];
        acc[11] += a2[p] * a2[p] + b2[p] * b2[p];
        acc[12] += a2[p] * a3[p];
        acc[13] += b2[p] * b4[p];
        acc[14] += a2[p] * a5[p] + b2[p] * b5[p];
        acc[15] += a3[p] * a3[p];
        acc[16] += a3[p] * a5[p];
        acc[17] += b4[p] * b4[p];
        acc[18] += b4[p] * b5[p];
        acc[19] += a5[p] * a5[p] + b5[p] * b5[p];
        acc[20] += a0[p] * ex[p] + b0[p] * ey[p];
        acc[21] += a1[p] * ex[p] + b1[p] * ey[p];
        acc[22] += a2[p] * ex[p] + b2[p] * ey[p];
        acc[23] += a3[p] * ex[p];
        acc[24] += b4[p] * ey[p];
        acc[25] += a5[p] * ex[p] + b5[p] * ey[p];
        acc[26] += ex[p] * ex[p] + ey[p] * ey[p];
    }
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

}  // namespace esac
