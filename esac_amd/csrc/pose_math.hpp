// pose_math.hpp -- fp64 pose arithmetic kept in registers (CDNA4 / gfx950).
//
// Everything here is written for one lane solving one problem: no local arrays
// with run-time indices (they would go to scratch), loops over candidates are
// unrolled with compile-time indices, comparisons keep the sense the CPU
// libraries use so NaNs take the same branch.
//
// What each routine stands in for on the reference's hot path:
//   rodrigues_*        cv::Rodrigues            (esac_util.h:540, via projectPoints / solvePnP)
//   p3p_4pt            cv::solvePnP(SOLVEPNP_P3P) called by safeSolvePnP (esac_util.h:85-114,189-197)
//   project_exact_err  cv::projectPoints + error (esac_util.h:309-319,355-360)
//   pnp_point_terms    one row pair of cvProjectPoints2's Jacobian inside solvePnP(ITERATIVE)
// The file is compiled with -ffp-contract=off: fp64 results then follow IEEE
// op-by-op, like the CPU code the results are compared with.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#define ESAC_HD __host__ __device__ __forceinline__

namespace esac {

struct Cam {
    double fx, fy, cx, cy;
};

// ---------------------------------------------------------------- Rodrigues
// vector -> matrix (row-major R[9]); optional d vec(R) / d r as J[i*9+k], i = r component.
template <bool WITH_JAC>
ESAC_HD void rodrigues_vec2mat(const double r[3], double R[9], double* J) {
    double rx = r[0], ry = r[1], rz = r[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        if (WITH_JAC) {
#pragma unroll
            for (int k = 0; k < 27; k++) J[k] = 0;
            J[5] = -1; J[15] = -1; J[19] = -1;
            J[7] = 1;  J[11] = 1;  J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c;
    const double itheta = 1. / theta;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double xx = rx * rx, xy = rx * ry, xz = rx * rz, yy = ry * ry, yz = ry * rz, zz = rz * rz;
    // R = c*I + (1-c)*n*nT + s*[n]x
    R[0] = c + c1 * xx;      R[1] = c1 * xy - s * rz; R[2] = c1 * xz + s * ry;
    R[3] = c1 * xy + s * rz; R[4] = c + c1 * yy;      R[5] = c1 * yz - s * rx;
    R[6] = c1 * xz - s * ry; R[7] = c1 * yz + s * rx; R[8] = c + c1 * zz;
    if (WITH_JAC) {
        // dR/dr_i = -s n_i I + (s - 2 c1/theta) n_i n nT + (c1/theta)(e_i nT + n e_iT)
        //           + (c - s/theta) n_i [n]x + (s/theta) [e_i]x
        const double a2 = c1 * itheta, a4 = s * itheta;
        const double nn[9] = {xx, xy, xz, xy, yy, yz, xz, yz, zz};
        const double nx[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        const double n[3] = {rx, ry, rz};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double ni = n[i];
            const double a0 = -s * ni, a1 = (s - 2 * c1 * itheta) * ni, a3 = (c - s * itheta) * ni;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int row = k / 3, col = k % 3;
                // (e_i nT + n e_iT)[row][col] = (row==i)*n[col] + (col==i)*n[row]
                const double sym = (row == i ? n[col] : 0.0) + (col == i ? n[row] : 0.0);
                // [e_i]x[row][col]
                double ex = 0.0;
                if (i == 0) ex = (row == 1 && col == 2) ? -1.0 : (row == 2 && col == 1) ? 1.0 : 0.0;
                if (i == 1) ex = (row == 0 && col == 2) ? 1.0 : (row == 2 && col == 0) ? -1.0 : 0.0;
                if (i == 2) ex = (row == 0 && col == 1) ? -1.0 : (row == 1 && col == 0) ? 1.0 : 0.0;
                J[i * 9 + k] = a0 * (row == col ? 1.0 : 0.0) + a1 * nn[k] + a2 * sym + a3 * nx[k] + a4 * ex;
            }
        }
    }
}

// matrix -> vector (R orthonormal to rounding: the SVD clean-up of cv::Rodrigues is omitted, as documented with the CPU checker)
ESAC_HD void rodrigues_mat2vec(const double R[9], double r[3]) {
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// 1/d to full double precision without the IEEE division sequence: v_rcp_f64 + two Newton steps
// (5 dependent ops instead of ~10; a dependent fp64 op costs ~32 cycles on gfx950).  Only used
// where the result feeds an iterative solver, never in the reference-arithmetic ("exact") routines.
ESAC_HD double fast_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double x = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    return x;
#else
    return 1.0 / d;
#endif
}

// ---------------------------------------------------------------- projection
// Reference arithmetic for one reprojection error: fp64 projection with the
// `z ? 1/z : 1` guard and no cheirality test, float result, float pixel
// difference, norm accumulated in double, cast to float (esac_util.h:355-358).
ESAC_HD float project_exact_err(const double R[9], const double t[3], const Cam& cam,
                                                   float X, float Y, float Z, float px, float py) {
    const double Xd = X, Yd = Y, Zd = Z;
    double x = R[0] * Xd + R[1] * Yd + R[2] * Zd + t[0];
    double y = R[3] * Xd + R[4] * Yd + R[5] * Zd + t[1];
    double z = R[6] * Xd + R[7] * Yd + R[8] * Zd + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    const float u = (float)(x * cam.fx + cam.cx);
    const float v = (float)(y * cam.fy + cam.cy);
    const float dx = px - u, dy = py - v;
    return (float)sqrt((double)dx * dx + (double)dy * dy);
}

// The same arithmetic for U points at once, written stage by stage: gfx950 issues in order and a
// dependent fp64 op has ~32 cycles of latency, so the U independent chains must be adjacent in the
// instruction stream to overlap.  Element-wise identical to project_exact_err.
template <int U>
ESAC_HD void project_exact_err_batch(const double R[9], const double t[3], const Cam& cam, const float (&X)[U],
                                     const float (&Y)[U], const float (&Z)[U], const float (&px)[U],
                                     const float (&py)[U], float (&err)[U]) {
    double x[U], y[U], z[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const double Xd = X[u], Yd = Y[u], Zd = Z[u];
        x[u] = R[0] * Xd + R[1] * Yd + R[2] * Zd + t[0];
        y[u] = R[3] * Xd + R[4] * Yd + R[5] * Zd + t[1];
        z[u] = R[6] * Xd + R[7] * Yd + R[8] * Zd + t[2];
    }
#pragma unroll
    for (int u = 0; u < U; u++) z[u] = z[u] ? 1. / z[u] : 1;
#pragma unroll
    for (int u = 0; u < U; u++) {
        x[u] *= z[u];
        y[u] *= z[u];
    }
    double n2[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const float uu = (float)(x[u] * cam.fx + cam.cx);
        const float vv = (float)(y[u] * cam.fy + cam.cy);
        const float dx = px[u] - uu, dy = py[u] - vv;
        n2[u] = (double)dx * dx + (double)dy * dy;
    }
#pragma unroll
    for (int u = 0; u < U; u++) err[u] = (float)sqrt(n2[u]);
}

// one soft-inlier term, reference arithmetic (esac_util.h:248-250)
ESAC_HD double soft_inlier_exact(float err, float tau, float beta) {
    double soft = beta * (err - tau);  // float ops, widened afterwards
    soft = 1 / (1 + exp(-soft));
    return 1 - soft;
}

// ---------------------------------------------------------------- quartic
// Closed-form real roots (Ferrari through the first real root of the resolvent
// cubic; MathWorld "Quartic Equation"/"Cubic Equation"), same branch structure
// as the solver behind cv::solvePnP(P3P) so that both find the same root sets.
ESAC_HD int cubic_first_roots(double a, double b, double c, double d, double& x0, double& x1,
                                                 double& x2) {
    const double kPi = 3.1415926535897932384626433832795;
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        const double delta = c * c - 4 * b * d;
        if (delta < 0) return 0;
        const double inv_2a = 0.5 / b;
        if (delta == 0) {
            x0 = -c * inv_2a;
            x1 = x0;
            return 1;
        }
        const double sq = sqrt(delta);
        x0 = (-c + sq) * inv_2a;
        x1 = (-c - sq) * inv_2a;
        return 2;
    }
    const double inv_a = 1. / a;
    const double b_a = inv_a * b, b_a2 = b_a * b_a;
    const double c_a = inv_a * c;
    const double d_a = inv_a * d;
    const double Q = (3 * c_a - b_a2) / 9;
    const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) {
            x0 = x1 = x2 = -b_a_3;
            return 3;
        }
        x0 = pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        const double theta = acos(R / sqrt(-Q3));
        const double sqrt_Q = sqrt(-Q);
        x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * cos((theta + 2 * kPi) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * cos((theta + 4 * kPi) / 3.0) - b_a_3;
        return 3;
    }
    const double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

ESAC_HD int quartic_real_roots(double a, double b, double c, double d, double e, double& x0,
                                                  double& x1, double& x2, double& x3) {
    if (a == 0) {
        x3 = 0;
        return cubic_first_roots(b, c, d, e, x0, x1, x2);
    }
    const double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    const int n = cubic_first_roots(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
    if (n == 0) return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double R = sqrt(R2);
    const double inv_R = 1. / R;
    int nb = 0;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0)
            D2 = E2 = -1;
        else {
            const double sq = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sq;
            E2 = D2 - 4 * sq;
        }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        const double D = sqrt(D2);
        nb = 2;
        x0 = R_2 + 0.5 * D - b_4;
        x1 = x0 - D;
    }
    if (E2 >= 0) {
        const double E = sqrt(E2);
        if (nb == 0) {
            x0 = -R_2 + 0.5 * E - b_4;
            x1 = x0 - E;
            nb = 2;
        } else {
            x2 = -R_2 + 0.5 * E - b_4;
            x3 = x2 - E;
            nb = 4;
        }
    }
    return nb;
}

// ---------------------------------------------------------------- P3P
struct V3 {
    double x, y, z;
};
ESAC_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
ESAC_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
ESAC_HD V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
ESAC_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ESAC_HD V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
ESAC_HD V3 unit(V3 a) { return (1. / sqrt(dot(a, a))) * a; }
ESAC_HD V3 matvec(const double R[9], V3 p) {
    return {R[0] * p.x + R[1] * p.y + R[2] * p.z, R[3] * p.x + R[4] * p.y + R[5] * p.z,
            R[6] * p.x + R[7] * p.y + R[8] * p.z};
}

// Rigid alignment R*P_k + T = Q_k of two (near-)congruent triangles.  The CPU
// library solves the least-squares absolute-orientation problem with Horn's
// quaternion method (4x4 Jacobi eigen-solve); here the same optimum is reached
// without an eigen-solve: orthonormal triads give the exact answer for congruent
// triangles, and Newton steps on the Procrustes objective (Cayley update, exactly
// orthonormal) remove the difference that remains when the quartic root is off
// and the triangles are not congruent (1 step typically, a few on bad roots).
ESAC_HD void align_triangles(V3 P0, V3 P1, V3 P2, V3 Q0, V3 Q1, V3 Q2, double R[9], double T[3]) {
    const V3 e1 = unit(P1 - P0);
    const V3 e3 = unit(cross(e1, P2 - P0));
    const V3 e2 = cross(e3, e1);
    const V3 f1 = unit(Q1 - Q0);
    const V3 f3 = unit(cross(f1, Q2 - Q0));
    const V3 f2 = cross(f3, f1);
    R[0] = f1.x * e1.x + f2.x * e2.x + f3.x * e3.x; R[1] = f1.x * e1.y + f2.x * e2.y + f3.x * e3.y; R[2] = f1.x * e1.z + f2.x * e2.z + f3.x * e3.z;
    R[3] = f1.y * e1.x + f2.y * e2.x + f3.y * e3.x; R[4] = f1.y * e1.y + f2.y * e2.y + f3.y * e3.y; R[5] = f1.y * e1.z + f2.y * e2.z + f3.y * e3.z;
    R[6] = f1.z * e1.x + f2.z * e2.x + f3.z * e3.x; R[7] = f1.z * e1.y + f2.z * e2.y + f3.z * e3.y; R[8] = f1.z * e1.z + f2.z * e2.z + f3.z * e3.z;

    const double third = 1. / 3;
    const V3 Pc = third * (P0 + P1 + P2), Qc = third * (Q0 + Q1 + Q2);
    const V3 p0 = P0 - Pc, p1 = P1 - Pc, p2 = P2 - Pc;
    const V3 q0 = Q0 - Qc, q1 = Q1 - Qc, q2 = Q2 - Qc;
    // Newton on g(w) = sum q_k . exp([w]x) a_k (exact Hessian, so convergence stays quadratic when the
    // triangles are not congruent): (sum (a.q) I - (a qT + q aT)/2) w = sum a x q
    for (int it = 0; it < 8; it++) {
        const V3 a0 = matvec(R, p0), a1 = matvec(R, p1), a2 = matvec(R, p2);
        const double s = dot(a0, q0) + dot(a1, q1) + dot(a2, q2);
        const double m00 = s - (a0.x * q0.x + a1.x * q1.x + a2.x * q2.x);
        const double m11 = s - (a0.y * q0.y + a1.y * q1.y + a2.y * q2.y);
        const double m22 = s - (a0.z * q0.z + a1.z * q1.z + a2.z * q2.z);
        const double m01 = -0.5 * (a0.x * q0.y + a0.y * q0.x + a1.x * q1.y + a1.y * q1.x + a2.x * q2.y + a2.y * q2.x);
        const double m02 = -0.5 * (a0.x * q0.z + a0.z * q0.x + a1.x * q1.z + a1.z * q1.x + a2.x * q2.z + a2.z * q2.x);
        const double m12 = -0.5 * (a0.y * q0.z + a0.z * q0.y + a1.y * q1.z + a1.z * q1.y + a2.y * q2.z + a2.z * q2.y);
        const V3 g = cross(a0, q0) + cross(a1, q1) + cross(a2, q2);
        // symmetric 3x3 solve by cofactors
        const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
        const double c11 = m00 * m22 - m02 * m02, c12 = m01 * m02 - m00 * m12, c22 = m00 * m11 - m01 * m01;
        const double det = m00 * c00 + m01 * c01 + m02 * c02;
        const double idet = 1. / det;
        V3 w = {idet * (c00 * g.x + c01 * g.y + c02 * g.z), idet * (c01 * g.x + c11 * g.y + c12 * g.z),
                idet * (c02 * g.x + c12 * g.y + c22 * g.z)};
        const double ww = dot(w, w);
        if (!(ww < 0.25)) break;   // no sensible step (degenerate sample): keep the current rotation
        if (!(ww > 1e-30)) break;  // converged
        const bool last = ww < 1e-14;  // quadratic convergence: the step after |w| < 1e-7 is below rounding
        // Cayley: exp([w]x) ~= ((1-|h|^2) I + 2 h hT + 2 [h]x) / (1+|h|^2), h = w/2
        const V3 h = 0.5 * w;
        const double hh = dot(h, h), k = 1. / (1. + hh);
        double C[9];
        C[0] = k * (1 - hh + 2 * h.x * h.x); C[1] = k * (2 * h.x * h.y - 2 * h.z);     C[2] = k * (2 * h.x * h.z + 2 * h.y);
        C[3] = k * (2 * h.x * h.y + 2 * h.z); C[4] = k * (1 - hh + 2 * h.y * h.y);     C[5] = k * (2 * h.y * h.z - 2 * h.x);
        C[6] = k * (2 * h.x * h.z - 2 * h.y); C[7] = k * (2 * h.y * h.z + 2 * h.x);     C[8] = k * (1 - hh + 2 * h.z * h.z);
        double Rn[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Rn[i * 3 + j] = C[i * 3] * R[j] + C[i * 3 + 1] * R[3 + j] + C[i * 3 + 2] * R[6 + j];
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = Rn[i];
        if (last) break;
    }
    const V3 RPc = matvec(R, Pc);
    T[0] = Qc.x - RPc.x; T[1] = Qc.y - RPc.y; T[2] = Qc.z - RPc.z;
}

// 4-point P3P (Gao, Hou, Tang, Cheng, PAMI 2003; main branch): up to four poses
// from points 0..2, the one with the smallest reprojection error of point 3 wins.
// Split in two so that the (up to four) candidates can be evaluated by different lanes: p3p_setup = everything up to
// the real roots of the quartic, p3p_candidate(i) = lengths + alignment + 4th-point error of root i.  p3p_4pt runs
// them in sequence; both routes execute the same operations per candidate, so they agree bit for bit.
struct P3PSetup {
    double mu[3], mv[3], mk[3];
    double dist2, a, b, p, q, r;
    double a2, b2, p2, q2, r2, pqr, ab, a_2, a_4, r3, pr2, r3q, inv_b0;
    double x[4];
    int n;
};

ESAC_HD bool p3p_setup(const V3 P[4], const double mu_px[4], const double mv_px[4], const Cam& cam, P3PSetup& S) {
    const double inv_fx = 1. / cam.fx, inv_fy = 1. / cam.fy, cx_fx = cam.cx / cam.fx, cy_fy = cam.cy / cam.fy;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        S.mu[i] = inv_fx * mu_px[i] - cx_fx;
        S.mv[i] = inv_fy * mv_px[i] - cy_fy;
        const double norm = sqrt(S.mu[i] * S.mu[i] + S.mv[i] * S.mv[i] + 1);
        S.mk[i] = 1. / norm;
        S.mu[i] *= S.mk[i];
        S.mv[i] *= S.mk[i];
    }
    const double* mu = S.mu;
    const double* mv = S.mv;
    const double* mk = S.mk;
    const V3 d12 = P[1] - P[2], d02 = P[0] - P[2], d01 = P[0] - P[1];
    const double dist0 = sqrt(d12.x * d12.x + d12.y * d12.y + d12.z * d12.z);
    const double dist1 = sqrt(d02.x * d02.x + d02.y * d02.y + d02.z * d02.z);
    const double dist2 = sqrt(d01.x * d01.x + d01.y * d01.y + d01.z * d01.z);
    const double cos0 = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    const double cos1 = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    const double cos2 = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];

    const double p = cos0 * 2, q = cos1 * 2, r = cos2 * 2;
    const double inv_d22 = 1. / (dist2 * dist2);
    const double a = inv_d22 * (dist0 * dist0);
    const double b = inv_d22 * (dist1 * dist1);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    S.n = 0;
    if (p2 + q2 + r2 - pqr - 1 == 0) return false;
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return false;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return false;

    double x0 = 0, x1 = 0, x2r = 0, x3 = 0;
    const int n = quartic_real_roots(A, B, C, D, E, x0, x1, x2r, x3);
    if (n == 0) return false;
    S.n = n;
    S.x[0] = x0; S.x[1] = x1; S.x[2] = x2r; S.x[3] = x3;
    S.dist2 = dist2; S.a = a; S.b = b; S.p = p; S.q = q; S.r = r;
    S.a2 = a2; S.b2 = b2; S.p2 = p2; S.q2 = q2; S.r2 = r2; S.pqr = pqr; S.ab = ab; S.a_2 = a_2; S.a_4 = a_4;
    S.r3 = r2 * r; S.pr2 = p * r2; S.r3q = S.r3 * q;
    S.inv_b0 = 1. / b0;
    return true;
}

// Depths (X, Y, Z along the three viewing rays) of the candidate of root x: false when it is not a valid solution.
// (Split out of p3p_candidate so that the fp32 sampling screen, p3p_screen.hpp, evaluates exactly the candidates the
// fp64 route evaluates -- same doubles, same validity tests.)
ESAC_HD bool p3p_candidate_lengths(const P3PSetup& S, double x, double& X, double& Y, double& Z) {
    const double a = S.a, b = S.b, p = S.p, q = S.q, r = S.r;
    const double a2 = S.a2, b2 = S.b2, p2 = S.p2, q2 = S.q2, r2 = S.r2, pqr = S.pqr, ab = S.ab, a_2 = S.a_2, a_4 = S.a_4;
    const double r3 = S.r3, pr2 = S.pr2, r3q = S.r3q;
    (void)pqr;
    if (x <= 0) return false;
    const double xx = x * x;
    // same association order as the CPU solver: b1 suffers heavy cancellation, and the two
    // sides only agree on ill-conditioned samples if they round the same way
    const double b1 =
        ((1 - a - b) * xx + (q * a - q) * x + 1 - a + b) *
        (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
          (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) +
           pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * xx +
         (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) +
          r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
          pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
         2 * r3q * (a_2 - b - a2 + ab - 1) +
         pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
         p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    if (b1 <= 0) return false;
    const double y = S.inv_b0 * b1;
    const double v = xx + y * y - x * y * r;
    if (v <= 0) return false;
    Z = S.dist2 / sqrt(v);
    X = x * Z;
    Y = y * Z;
    return true;
}

// candidate of root x: false when it is not a valid solution
ESAC_HD bool p3p_candidate(const P3PSetup& S, double x, const V3 P[4], const double mu3_px, const double mv3_px,
                           const Cam& cam, double R[9], double T[3], double& reproj) {
    double X, Y, Z;
    if (!p3p_candidate_lengths(S, x, X, Y, Z)) return false;
    const V3 Q0 = {X * S.mu[0], X * S.mv[0], X * S.mk[0]};
    const V3 Q1 = {Y * S.mu[1], Y * S.mv[1], Y * S.mk[1]};
    const V3 Q2 = {Z * S.mu[2], Z * S.mv[2], Z * S.mk[2]};
    align_triangles(P[0], P[1], P[2], Q0, Q1, Q2, R, T);
    const double X3p = R[0] * P[3].x + R[1] * P[3].y + R[2] * P[3].z + T[0];
    const double Y3p = R[3] * P[3].x + R[4] * P[3].y + R[5] * P[3].z + T[1];
    const double Z3p = R[6] * P[3].x + R[7] * P[3].y + R[8] * P[3].z + T[2];
    const double mu3p = cam.cx + cam.fx * X3p / Z3p;
    const double mv3p = cam.cy + cam.fy * Y3p / Z3p;
    reproj = (mu3p - mu3_px) * (mu3p - mu3_px) + (mv3p - mv3_px) * (mv3p - mv3_px);
    return true;
}

// obj: 4 scene points, img: 4 pixel positions.  Returns false when there is no solution.
ESAC_HD bool p3p_4pt(const V3 P[4], const double mu_px[4], const double mv_px[4], const Cam& cam,
                                        double Rbest[9], double Tbest[3], double* best_reproj = nullptr) {
    P3PSetup S;
    if (!p3p_setup(P, mu_px, mv_px, cam, S)) return false;
    bool have = false;
    double min_reproj = 0;
    // not unrolled: four inlined copies of the alignment bought nothing (each candidate is one dependent chain)
    // and cost registers and instruction cache
#pragma nounroll
    for (int i = 0; i < S.n; i++) {
        const double x = (i == 0) ? S.x[0] : (i == 1) ? S.x[1] : (i == 2) ? S.x[2] : S.x[3];
        double R[9], T[3], reproj;
        if (!p3p_candidate(S, x, P, mu_px[3], mv_px[3], cam, R, T, reproj)) continue;
        if (!have || min_reproj > reproj) {
            have = true;
            min_reproj = reproj;
#pragma unroll
            for (int k = 0; k < 9; k++) Rbest[k] = R[k];
            Tbest[0] = T[0]; Tbest[1] = T[1]; Tbest[2] = T[2];
        }
    }
    if (best_reproj) *best_reproj = min_reproj;  // squared pixel error of the 4th point under the chosen candidate
    return have;
}

// ---------------------------------------------------------------- LM terms
// One correspondence of solvePnP(ITERATIVE): residual (ex,ey) = projection - pixel
// in double, and the two Jacobian rows wrt (rvec, tvec).
ESAC_HD void pnp_point_terms(const double R[9], const double dRdr[27], const double t[3],
                                                const Cam& cam, double X, double Y, double Z, double mx, double my,
                                                double& ex, double& ey, double Ju[6], double Jv[6]) {
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    ex = (x * cam.fx + cam.cx) - mx;
    ey = (y * cam.fy + cam.cy) - my;
    Ju[3] = cam.fx * z; Ju[4] = 0;          Ju[5] = -cam.fx * x * z;
    Jv[3] = 0;          Jv[4] = cam.fy * z; Jv[5] = -cam.fy * y * z;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double* dR = dRdr + 9 * j;
        const double dx0 = X * dR[0] + Y * dR[1] + Z * dR[2];
        const double dy0 = X * dR[3] + Y * dR[4] + Z * dR[5];
        const double dz0 = X * dR[6] + Y * dR[7] + Z * dR[8];
        Ju[j] = cam.fx * (z * (dx0 - x * dz0));
        Jv[j] = cam.fy * (z * (dy0 - y * dz0));
    }
}

ESAC_HD void pnp_point_residual(const double R[9], const double t[3], const Cam& cam, double X,
                                                   double Y, double Z, double mx, double my, double& ex, double& ey) {
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    ex = (x * cam.fx + cam.cx) - mx;
    ey = (y * cam.fy + cam.cy) - my;
}

// Damped normal equations of one LM step: (JtJ with diag*(1+lambda)) dx = JtErr.
// JtJ arrives as the 21 upper-triangle sums.  LDL^T (the matrix is SPD for any non-degenerate inlier set; the CPU
// library takes an SVD solve, which gives the same step there).  Returns false -- dx is then meaningless -- when a
// pivot falls below 1e-12 of its diagonal entry (rank-deficient inlier set, e.g. collinear points, once lambda has
// shrunk): the caller then takes the SVD route itself (lm_solve6_pinv), as the CPU library always does.
ESAC_HD bool lm_solve6(const double U21[21], const double g[6], double lambda, double dx[6]) {
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
#pragma unroll
    for (int i = 0; i < 6; i++) A[i][i] *= 1. + lambda;
    // A = L D L^T (unit lower L): 6 reciprocals on the dependent chain, no square roots
    double L[6][6], Dinv[6], W[6][6];  // W[i][m] = L[i][m] * D[m]
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
#pragma unroll
        for (int m = 0; m < j; m++) d -= W[j][m] * L[j][m];
        if (!(d > 1e-12 * A[j][j])) ok = false;
        const double inv = fast_rcp(d);  // six reciprocals sit on the serial chain of every LM iteration
        Dinv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = A[i][j];
#pragma unroll
            for (int m = 0; m < j; m++) s -= W[i][m] * L[j][m];
            W[i][j] = s;
            L[i][j] = s * inv;
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = g[i];
#pragma unroll
        for (int m = 0; m < i; m++) s -= L[i][m] * y[m];
        y[i] = s;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = y[i] * Dinv[i];
#pragma unroll
        for (int m = i + 1; m < 6; m++) s -= L[m][i] * dx[m];
        dx[i] = s;
    }
    return ok;
}

// camera transform = inverse of the scene pose (esac_util.h:537-548), rigid inverse
ESAC_HD void pose_to_inverse_transform(const double R[9], const double t[3], double T[16]) {
    T[0] = R[0]; T[1] = R[3]; T[2] = R[6];
    T[4] = R[1]; T[5] = R[4]; T[6] = R[7];
    T[8] = R[2]; T[9] = R[5]; T[10] = R[8];
    T[3] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);
    T[7] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
    T[11] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

}  // namespace esac
