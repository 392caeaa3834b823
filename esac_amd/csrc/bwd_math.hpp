// bwd_math.hpp -- per-hypothesis / per-cell arithmetic of the training path (esac.backward).
//
// What each routine stands in for (reference file:line):
//   norm_jac_row     d ||proj - px|| / d (rvec,tvec) of one cell      esac_util.h:333-351, esac.cpp:417-431
//   dproject_dobj    d ||proj - px|| / d (x,y,z) of one cell          esac_derivative.h:47-102
//   pose_loss        loss() on the inverted (camera) transforms       esac_loss.h:45-83, esac.cpp:357-360
//   pose_dloss       dLoss(): 1x6 derivative wrt (rvec,tvec)          esac_loss.h:94-210 (keeps the sqrt(loss) quirk)
//   inv_spd6         (J^T J)^-1 for the pseudo-inverse of esac.cpp:434
// Host+device (ESAC_HD) like pose_math.hpp, so the CPU test-suite can run them against its checker.
#pragma once
#include "pose_math.hpp"

namespace esac {

constexpr double kRefEps = 0.00000001;      // EPS, esac_util.h:39
constexpr double kRefPi = 3.1415926;        // PI,  esac_util.h:40 (calcAngularDistance)
constexpr double kCvPi = 3.1415926535897932384626433832795;
constexpr double kMaxLoss = 10000000.0;     // MAXLOSS, esac_loss.h:33
constexpr double kProbThresh = 0.001;       // PROB_THRESH, esac_derivative.h:33

// Row of the residual-norm Jacobian of one correspondence; false (and a zero row) when the reference skips it.
ESAC_HD bool norm_jac_row(const double R[9], const double dRdr[27], const double t[3], const Cam& cam, float X, float Y,
                          float Z, float px, float py, float max_reproj, double row[6]) {
    double u, v, Ju[6], Jv[6];
    pnp_point_terms(R, dRdr, t, cam, (double)X, (double)Y, (double)Z, 0.0, 0.0, u, v, Ju, Jv);  // residual vs 0 = projection
    const float uf = (float)u, vf = (float)v;  // projections are Point2f
    const float dx = uf - px, dy = vf - py;
    double err = sqrt((double)dx * dx + (double)dy * dy);
    err = err > kRefEps ? err : kRefEps;  // std::max(norm, EPS)
#pragma unroll
    for (int k = 0; k < 6; k++) row[k] = 0;
    if (err > max_reproj) return false;
    const double a = 1 / err * (uf - px), b = 1 / err * (vf - py);
#pragma unroll
    for (int k = 0; k < 6; k++) row[k] = a * Ju[k] + b * Jv[k];
    return true;
}

ESAC_HD void dproject_dobj(float ptx, float pty, float ox, float oy, float oz, const double R[9], const double t[3],
                           float focal, float ppx_f, float ppy_f, float max_reproj, double out[3]) {
    const double f = (double)focal, ppx = (double)ppx_f, ppy = (double)ppy_f;  // read from the FLOAT camera matrix
    out[0] = out[1] = out[2] = 0;
    const double X = R[0] * ox + R[1] * oy + R[2] * oz + t[0];
    const double Y = R[3] * ox + R[4] * oy + R[5] * oz + t[1];
    const double Z = R[6] * ox + R[7] * oy + R[8] * oz + t[2];
    if (fabs(Z) < kRefEps) return;
    const double px = f * X / Z + ppx;
    const double py = f * Y / Z + ppy;
    double err = sqrt((ptx - px) * (ptx - px) + (pty - py) * (pty - py));
    if (err > max_reproj) return;
    err += kRefEps;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double pxd = f * R[0 + k] / Z - f * X / Z / Z * R[6 + k];
        const double pyd = f * R[3 + k] / Z - f * Y / Z / Z * R[6 + k];
        out[k] = 0.5 / err * (2 * (ptx - px) * -pxd + 2 * (pty - py) * -pyd);
    }
}

// loss(pose2trans(pose), gtTrans): gt = row-major 4x4 camera pose (double of the float input)
ESAC_HD double pose_loss(const double pose[6], const double gt[16], double wRot, double wTrans, double cut) {
    double R[9], T[16];
    rodrigues_vec2mat<false>(pose, R, nullptr);
    pose_to_inverse_transform(R, pose + 3, T);
    double tr = 0;  // trace(rot2 * rot1^T): diagonal entries of the product first, then their sum
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double d = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) d += gt[4 * i + k] * T[4 * i + k];
        tr += d;
    }
    tr = tr < 3.0 ? (tr > -1.0 ? tr : -1.0) : 3.0;
    const double rotErr = 180 * acos((tr - 1.0) / 2.0) / kRefPi;
    double d2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) d2 += (T[4 * i + 3] - gt[4 * i + 3]) * (T[4 * i + 3] - gt[4 * i + 3]);
    const double tErr = sqrt(d2);
    double loss = wRot * rotErr + wTrans * tErr;
    if (loss > cut) loss = sqrt(cut * loss);
    return loss < kMaxLoss ? loss : kMaxLoss;
}

// dLoss(est, gt): est, gt are scene poses (rvec,tvec); gt = trans2pose(gtTrans)
ESAC_HD void pose_dloss(const double est[6], const double gt[6], double wRot, double wTrans, double cut, double jac[6]) {
    double R1[9], dRod[27], R2[9];
    rodrigues_vec2mat<true>(est, R1, dRod);
    rodrigues_vec2mat<false>(gt, R2, nullptr);
#pragma unroll
    for (int k = 0; k < 6; k++) jac[k] = 0;
    double tr = 0;  // trace(rot1 * rot2^T)
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double d = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) d += R1[3 * a + c] * R2[3 * a + c];
        tr += d;
    }
    tr = tr < 3.0 ? (tr > -1.0 ? tr : -1.0) : 3.0;
    const double rotErr = 180 * acos((tr - 1.0) / 2.0) / kCvPi;
    double invT1[3], invT2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        invT1[i] = R1[0 + i] * est[3] + R1[3 + i] * est[4] + R1[6 + i] * est[5];  // rot1^T * t1
        invT2[i] = R2[0 + i] * gt[3] + R2[3 + i] * gt[4] + R2[6 + i] * gt[5];
    }
    double d2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) d2 += (invT1[i] - invT2[i]) * (invT1[i] - invT2[i]);
    const double tErr = sqrt(d2);
    double loss = wRot * rotErr + wTrans * tErr;
    bool cutLoss = false;
    if (loss > cut) {
        loss = sqrt(loss);  // sic (esac_loss.h:133-137): sqrt(loss), not sqrt(cut*loss) -- kept bug-compatible
        cutLoss = true;
    }
    if (loss > kMaxLoss) return;
    if ((tErr + rotErr) < kRefEps) return;
    double dD[3];
#pragma unroll
    for (int i = 0; i < 3; i++) dD[i] = (invT1[i] - invT2[i]) / tErr;
#pragma unroll
    for (int j = 0; j < 3; j++) {  // dDist_dInvT1 (1x3) * invRot1 (3x3)
        double s = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) s += dD[i] * R1[3 * j + i];
        jac[3 + j] += s * wTrans;
    }
    double g9[9];
#pragma unroll
    for (int k = 0; k < 9; k++) g9[k] = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) g9[3 * j + i] += dD[i] * est[3 + j];
#pragma unroll
    for (int m = 0; m < 3; m++) {  // ... * dInvT1_dInvRot1 (3x9) * dRod^T (9x3)
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += g9[k] * dRod[9 * m + k];
        jac[m] += s * wTrans;
    }
    const double fac = 180 / kCvPi * -1 / sqrt(3 - tr * tr + 2 * tr);
#pragma unroll
    for (int m = 0; m < 3; m++) {  // dTrace * dRotDiff^T * dRod^T: d diffRot[a][a] / d rot1[a][c] = rot2[a][c]
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += R2[k] * dRod[9 * m + k];
        jac[m] += fac * s * wRot;
    }
    if (cutLoss) {
#pragma unroll
        for (int k = 0; k < 6; k++) jac[k] *= 0.5 / loss;
    }
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 6; k++) nan |= (jac[k] != jac[k]);
    if (nan) {
#pragma unroll
        for (int k = 0; k < 6; k++) jac[k] = 0;
    }
}

// Inverse of the symmetric positive definite 6x6 J^T J (upper triangle row-major in U21) through LDL^T.
// The reference takes the SVD pseudo-inverse (esac.cpp:434); the two coincide whenever J^T J has full rank,
// which any non-degenerate inlier set gives.  Returns false when a pivot falls below 1e-7 of its diagonal entry
// (rank deficient or badly conditioned): the caller then takes pinv_sym6_jacobi, the reference's own route.
ESAC_HD bool inv_spd6(const double U21[21], double Ainv[36]) {
#pragma clang fp contract(fast)
    double A[6][6];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) {
            A[i][j] = U21[k];
            A[j][i] = U21[k];
            k++;
        }
    double L[6][6], Dinv[6], W[6][6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
#pragma unroll
        for (int m = 0; m < j; m++) d -= W[j][m] * L[j][m];
        if (!(d > 1e-7 * A[j][j])) ok = false;
        Dinv[j] = 1. / d;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = A[i][j];
#pragma unroll
            for (int m = 0; m < j; m++) s -= W[i][m] * L[j][m];
            W[i][j] = s;
            L[i][j] = s * Dinv[j];
        }
    }
    // columns of the inverse: solve L D L^T x = e_c
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double y[6], x[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < i; m++) s -= L[i][m] * y[m];
            y[i] = s;
        }
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            double s = y[i] * Dinv[i];
#pragma unroll
            for (int m = i + 1; m < 6; m++) s -= L[m][i] * x[m];
            x[i] = s;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) Ainv[i * 6 + c] = x[i];
    }
    return ok;
}

// Pseudo-inverse of a symmetric 6x6 the way the reference obtains it (cv::Mat::inv(DECOMP_SVD), esac.cpp:434): eigen-
// decomposition by cyclic Jacobi rotations, eigenvalues below 2*eps*sum|w| dropped.  Used when inv_spd6 meets a
// (numerically) rank-deficient or badly conditioned J^T J -- degenerate inlier sets of garbage hypotheses -- where
// "the inverse" and the pseudo-inverse part ways.  Fully unrolled (static register indices); ~10k flops, rare.
ESAC_HD void pinv_sym6_jacobi(const double U21[21], double Ainv[36]) {
    double A[6][6], V[6][6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) {
                A[i][j] = U21[k];
                A[j][i] = U21[k];
                k++;
            }
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i + 1; j < 6; j++) off += A[i][j] * A[i][j];
        if (off == 0) break;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p][q];
                const double theta = (A[q][q] - A[p][p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                if (!(fabs(theta) <= 1.7976931348623157e308)) t = 0;  // apq negligible (theta = inf / nan)
                if (apq == 0) t = 0;                                   // identity rotation = the reference's `continue`
                const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    double thresh = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) thresh += fabs(A[i][i]);
    thresh *= 2 * 2.220446049250313e-16;
#pragma unroll
    for (int i = 0; i < 36; i++) Ainv[i] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double w = A[k][k];
        const double inv = fabs(w) > thresh ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) Ainv[i * 6 + j] += inv != 0.0 ? V[i][k] * V[j][k] / w : 0.0;
    }
}

}  // namespace esac
