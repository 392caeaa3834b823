/*
 * esac_oracle.c -- CPU oracle (plain C11 + OpenMP) for the ESAC forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see esac_oracle.h.  PARITY UNPINNED (no buildable
 * reference, no reference golden vectors): every OpenCV stand-in below is a
 * restatement of the published algorithm from memory and says so.
 *
 * Structure follows the reference phase by phase so that it can double as the
 * timed CPU baseline (per-hypothesis full-map projection, `omp parallel for`
 * over hypotheses, serial winner refinement):
 *
 *   createSampling      esac_util.h:53-70    -> px_of()
 *   sampleHypotheses    esac_util.h:129-225  -> sample_hypotheses()
 *   safeSolvePnP(P3P)   esac_util.h:85-114   -> esac_oracle_p3p()        [OpenCV p3p.cpp, from memory]
 *   cv::projectPoints   esac_util.h:202,312  -> project_point()          [OpenCV cvProjectPoints2, from memory]
 *   getReproErrs        esac_util.h:274-363  -> repro_errs()
 *   getHypScores        esac_util.h:235-260  -> hyp_scores()
 *   softMax/entropy/draw esac_util.h:461-530 -> soft_max(), entropy(), draw_argmax()
 *   refineHyp           esac_util.h:378-454  -> refine_hyp()
 *   solvePnP(ITERATIVE) via esac_util.h:426  -> esac_oracle_lm_pnp()     [OpenCV CvLevMarq, from memory]
 *   pose2trans          esac_util.h:537-548  -> esac_oracle_pose2trans()
 *   esac_forward        esac.cpp:64-190      -> esac_oracle_forward()
 *
 * RNG: the reference's per-OpenMP-thread mt19937 (thread_rand.cpp:13-42) cannot
 * be reproduced on a GPU (state depends on thread count and call history), so
 * both this oracle and the HIP path use the same counter-based Philox4x32-10
 * keyed on (seed, call, hypothesis, try, block).  Range semantics are kept:
 * irand(0, imW-1) with the wrapper's exclusive upper bound (thread_rand.cpp:68-71)
 * gives x in [0, W-2], y in [0, H-2].
 */
#define _GNU_SOURCE
#include "esac_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPS_REF 0.00000001 /* esac_util.h:39 */
#define CV_PI_ 3.1415926535897932384626433832795

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int esac_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11).  Counter-based; shared with the HIP  */
/* path (esac_amd/csrc/rng.hpp re-implements it independently).               */
/* ------------------------------------------------------------------------- */
void esac_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline int mulhi_range(uint32_t u, int n) { /* uniform on [0,n) */
    return (int)(((uint64_t)u * (uint64_t)(uint32_t)n) >> 32);
}

/* Draw the 4 DISTINCT cells of try (hyp,tr) (esac_util.h:164-176).  Block k of
 * the stream yields candidate cells 2k and 2k+1; duplicates are skipped, which
 * is the reference's "j--; continue" redraw. */
void esac_oracle_draw_cells(uint64_t seed, uint64_t call, uint32_t hyp, uint32_t tr,
                            int W, int H, int32_t xy[8]) {
    uint64_t key64 = seed + call * 0x9E3779B97F4A7C15ull;
    uint32_t key[2] = {(uint32_t)key64, (uint32_t)(key64 >> 32)};
    int have = 0;
    for (uint32_t k = 0; have < 4; k++) {
        uint32_t ctr[4] = {hyp, tr, k, 0x45534143u}, o[4];
        esac_oracle_philox4x32(ctr, key, o);
        for (int half = 0; half < 2 && have < 4; half++) {
            int x = mulhi_range(o[2 * half + 0], W - 1); /* irand(0, imW-1) -> [0, W-2] */
            int y = mulhi_range(o[2 * half + 1], H - 1);
            int dup = 0;
            for (int j = 0; j < have; j++)
                if (xy[2 * j] == x && xy[2 * j + 1] == y) dup = 1;
            if (dup) continue;
            xy[2 * have] = x;
            xy[2 * have + 1] = y;
            have++;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Polynomial solvers  [OpenCV calib3d/polynom_solver.cpp, from memory;       */
/* MathWorld "Cubic Equation"/"Quartic Equation" closed forms]                */
/* ------------------------------------------------------------------------- */
static int solve_deg2(double a, double b, double c, double* x1, double* x2) {
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) {
        *x1 = -b * inv_2a;
        *x2 = *x1;
        return 1;
    }
    double sqrt_delta = sqrt(delta);
    *x1 = (-b + sqrt_delta) * inv_2a;
    *x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

static int solve_deg3(double a, double b, double c, double d, double* x0, double* x1, double* x2) {
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a;
    double c_a = inv_a * c;
    double d_a = inv_a * d;

    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;

    if (Q == 0) {
        if (R == 0) {
            *x0 = *x1 = *x2 = -b_a_3;
            return 3;
        } else {
            *x0 = pow(2 * R, 1 / 3.0) - b_a_3;
            return 1;
        }
    }
    if (D <= 0) {
        /* three real roots */
        double theta = acos(R / sqrt(-Q3));
        double sqrt_Q = sqrt(-Q);
        *x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        *x1 = 2 * sqrt_Q * cos((theta + 2 * CV_PI_) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * cos((theta + 4 * CV_PI_) / 3.0) - b_a_3;
        return 3;
    }
    /* one real root */
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

int esac_oracle_solve_deg4(double a, double b, double c, double d, double e, double roots[4]) {
    double *x0 = &roots[0], *x1 = &roots[1], *x2 = &roots[2], *x3 = &roots[3];
    if (a == 0) {
        *x3 = 0;
        return solve_deg3(b, c, d, e, x0, x1, x2);
    }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;

    /* resolvent cubic; only its first root is used */
    double r0, r1, r2;
    int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
    if (n == 0) return 0;

    double R2 = 0.25 * b2 - c + r0, R;
    if (R2 < 0) return 0;
    R = sqrt(R2);
    double inv_R = 1. / R;

    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0)
            D2 = E2 = -1;
        else {
            double sqrt_temp = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
            E2 = D2 - 4 * sqrt_temp;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = sqrt(D2);
        nb_real_roots = 2;
        double D_2 = 0.5 * D;
        *x0 = R_2 + D_2 - b_4;
        *x1 = *x0 - D;
    }
    if (E2 >= 0) {
        double E = sqrt(E2);
        double E_2 = 0.5 * E;
        if (nb_real_roots == 0) {
            *x0 = -R_2 + E_2 - b_4;
            *x1 = *x0 - E;
            nb_real_roots = 2;
        } else {
            *x2 = -R_2 + E_2 - b_4;
            *x3 = *x2 - E;
            nb_real_roots = 4;
        }
    }
    return nb_real_roots;
}

/* ------------------------------------------------------------------------- */
/* P3P: Gao, Hou, Tang, Cheng, "Complete Solution Classification for the      */
/* Perspective-Three-Point Problem", PAMI 25(8) 2003 -- as used by            */
/* cv::solvePnP(SOLVEPNP_P3P) [OpenCV calib3d/p3p.cpp, from memory].          */
/* ------------------------------------------------------------------------- */

/* Cyclic Jacobi eigen-decomposition of a symmetric 4x4 (Numerical Recipes
 * `jacobi`, as in p3p::jacobi_4x4).  A: row-major 16, destroyed. */
static int jacobi_4x4(double* A, double* D, double* U) {
    double B[4], Z[4];
    memset(U, 0, 16 * sizeof(double));
    U[0] = U[5] = U[10] = U[15] = 1.0;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    memcpy(D, B, 4 * sizeof(double));
    memset(Z, 0, 4 * sizeof(double));

    for (int iter = 0; iter < 50; iter++) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return 1;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            double* pAij = A + 5 * i + 1;
            for (int j = i + 1; j < 4; j++) {
                double Aij = *pAij;
                double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) &&
                    fabs(D[j]) + eps_machine == fabs(D[j]))
                    *pAij = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh))
                        t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh;
                    D[i] -= hh; D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t);
                    double s = t * c;
                    double tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; k++) {
                        double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = i + 1; k <= j - 1; k++) {
                        double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = j + 1; k < 4; k++) {
                        double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (int k = 0; k < 4; k++) {
                        double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
                pAij++;
            }
        }
        for (int i = 0; i < 4; i++) B[i] += Z[i];
        memcpy(D, B, 4 * sizeof(double));
        memset(Z, 0, 4 * sizeof(double));
    }
    return 0;
}

/* Absolute orientation (Horn 1987, unit quaternion) of 3 scene points onto 3
 * camera-frame points M_end: R*X + T = M_end. */
static int align3(const double M_end[3][3], const double X[3][3], double R[9], double T[3]) {
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; i++) {
        C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
        C_start[i] = (X[0][i] + X[1][i] + X[2][i]) / 3;
    }
    double s[9];
    for (int j = 0; j < 3; j++) {
        s[0 * 3 + j] = (X[0][0] * M_end[0][j] + X[1][0] * M_end[1][j] + X[2][0] * M_end[2][j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (X[0][1] * M_end[0][j] + X[1][1] * M_end[1][j] + X[2][1] * M_end[2][j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (X[0][2] * M_end[0][j] + X[1][2] * M_end[1][j] + X[2][2] * M_end[2][j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0] + s[4] + s[8];
    Qs[1 * 4 + 1] = s[0] - s[4] - s[8];
    Qs[2 * 4 + 2] = s[4] - s[8] - s[0];
    Qs[3 * 4 + 3] = s[8] - s[0] - s[4];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];

    jacobi_4x4(Qs, evs, U);

    int i_ev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; i++)
        if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];

    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q1_2 - q0_3);    R[2] = 2. * (q1_3 + q0_2);
    R[3] = 2. * (q1_2 + q0_3);    R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q2_3 - q0_1);
    R[6] = 2. * (q1_3 - q0_2);    R[7] = 2. * (q2_3 + q0_1);    R[8] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++)
        T[i] = C_end[i] - (R[i * 3 + 0] * C_start[0] + R[i * 3 + 1] * C_start[1] + R[i * 3 + 2] * C_start[2]);
    return 1;
}

/* |PA|,|PB|,|PC| for the main branch of Gao's classification. distances =
 * |BC|,|AC|,|AB|; cosines = cos BPC, APC, APB. */
static int solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3]) {
    double p = cosines[0] * 2;
    double q = cosines[1] * 2;
    double r = cosines[2] * 2;

    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);

    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;

    /* reality condition (P and the three points must not be coplanar) */
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;

    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;

    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;

    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0) return 0;

    double real_roots[4];
    int n = esac_oracle_solve_deg4(A, B, C, D, E, real_roots);
    if (n == 0) return 0;

    int nb_solutions = 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    double inv_b0 = 1. / b0;

    for (int i = 0; i < n; i++) {
        double x = real_roots[i];
        if (x <= 0) continue;
        double x2 = x * x;

        double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) +
               pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) +
              r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) +
             pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));

        if (b1 <= 0) continue;

        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;

        double Z = distances[2] / sqrt(v);
        double X = x * Z;
        double Y = y * Z;
        lengths[nb_solutions][0] = X;
        lengths[nb_solutions][1] = Y;
        lengths[nb_solutions][2] = Z;
        nb_solutions++;
    }
    return nb_solutions;
}

int esac_oracle_p3p_all(const double* obj3, const double* img3, double fx, double fy,
                        double cx, double cy, double* Rs, double* ts) {
    double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double mu[3], mv[3], mk[3];
    for (int i = 0; i < 3; i++) {
        mu[i] = inv_fx * img3[2 * i] - cx_fx;
        mv[i] = inv_fy * img3[2 * i + 1] - cy_fy;
        double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm;
        mu[i] *= mk[i];
        mv[i] *= mk[i];
    }
    const double *P0 = obj3, *P1 = obj3 + 3, *P2 = obj3 + 6;
    double distances[3];
    distances[0] = sqrt((P1[0] - P2[0]) * (P1[0] - P2[0]) + (P1[1] - P2[1]) * (P1[1] - P2[1]) + (P1[2] - P2[2]) * (P1[2] - P2[2]));
    distances[1] = sqrt((P0[0] - P2[0]) * (P0[0] - P2[0]) + (P0[1] - P2[1]) * (P0[1] - P2[1]) + (P0[2] - P2[2]) * (P0[2] - P2[2]));
    distances[2] = sqrt((P0[0] - P1[0]) * (P0[0] - P1[0]) + (P0[1] - P1[1]) * (P0[1] - P1[1]) + (P0[2] - P1[2]) * (P0[2] - P1[2]));
    double cosines[3];
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];

    double lengths[4][3];
    int n = solve_for_lengths(lengths, distances, cosines);

    int nb_solutions = 0;
    double X[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) X[i][j] = obj3[3 * i + j];
    for (int i = 0; i < n; i++) {
        double M_orig[3][3];
        for (int k = 0; k < 3; k++) {
            M_orig[k][0] = lengths[i][k] * mu[k];
            M_orig[k][1] = lengths[i][k] * mv[k];
            M_orig[k][2] = lengths[i][k] * mk[k];
        }
        if (!align3(M_orig, X, Rs + 9 * nb_solutions, ts + 3 * nb_solutions)) continue;
        nb_solutions++;
    }
    return nb_solutions;
}

/* 4-point variant: candidate with the smallest reprojection error of point 3 wins. */
int esac_oracle_p3p(const double* obj, const double* img, double fx, double fy,
                    double cx, double cy, double rvec[3], double tvec[3]) {
    double Rs[4 * 9], ts[4 * 3];
    int n = esac_oracle_p3p_all(obj, img, fx, fy, cx, cy, Rs, ts);
    if (n == 0) return 0;
    int ns = 0;
    double min_reproj = 0;
    const double X3 = obj[9], Y3 = obj[10], Z3 = obj[11];
    const double mu3 = img[6], mv3 = img[7];
    for (int i = 0; i < n; i++) {
        const double* R = Rs + 9 * i;
        const double* t = ts + 3 * i;
        double X3p = R[0] * X3 + R[1] * Y3 + R[2] * Z3 + t[0];
        double Y3p = R[3] * X3 + R[4] * Y3 + R[5] * Z3 + t[1];
        double Z3p = R[6] * X3 + R[7] * Y3 + R[8] * Z3 + t[2];
        double mu3p = cx + fx * X3p / Z3p;
        double mv3p = cy + fy * Y3p / Z3p;
        double reproj = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
        if (i == 0 || min_reproj > reproj) {
            ns = i;
            min_reproj = reproj;
        }
    }
    esac_oracle_rodrigues_mat2vec(Rs + 9 * ns, rvec);
    tvec[0] = ts[3 * ns]; tvec[1] = ts[3 * ns + 1]; tvec[2] = ts[3 * ns + 2];
    return 1;
}

/* ------------------------------------------------------------------------- */
/* Rodrigues  [OpenCV cvRodrigues2, from memory]                              */
/* ------------------------------------------------------------------------- */
void esac_oracle_rodrigues_vec2mat(const double r_in[3], double R[9], double J[27]) {
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        memset(R, 0, 9 * sizeof(double));
        R[0] = R[4] = R[8] = 1;
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c;
    double itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    /* R = cos*I + (1-cos)*r*rT + sin*[r]x */
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        /* J is 3x9: row i = d vec(R) / d r_i */
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        static const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                          0, 0, 1, 0, 0, 0, -1, 0, 0,
                                          0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

/* Matrix -> vector.  Deviation: OpenCV first re-orthonormalises R by SVD
 * (R = U*Vt); inputs here are always rotations built from a unit quaternion or
 * Rodrigues, orthonormal to rounding, so that step is omitted. */
void esac_oracle_rodrigues_mat2vec(const double R[9], double r[3]) {
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0)
            rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5;
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

static void jacobi_sym(int n, double* A, double* w, double* V);

/* cv::Rodrigues(matrix -> vector) the way OpenCV runs it on a general 3x3 input: R is first replaced by the
 * nearest rotation U*Vt of its SVD, i.e. the orthogonal polar factor R (R^T R)^(-1/2).  Matters where the
 * reference feeds a FLOAT matrix (trans2pose on the ground-truth pose, esac_util.h:555-568; the float entries
 * are orthonormal to ~1e-7 only).  The P3P path keeps the plain routine above. */
void esac_oracle_rodrigues_mat2vec_svd(const double R[9], double r[3]) {
    double A[9], w[3], V[9], M[9], Ro[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += R[3 * k + i] * R[3 * k + j];
            A[3 * i + j] = s;
        }
    jacobi_sym(3, A, w, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += V[3 * i + k] * V[3 * j + k] / sqrt(w[k]);
            M[3 * i + j] = s;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += R[3 * i + k] * M[3 * k + j];
            Ro[3 * i + j] = s;
        }
    esac_oracle_rodrigues_mat2vec(Ro, r);
}

/* ------------------------------------------------------------------------- */
/* Pinhole projection  [OpenCV cvProjectPoints2 without distortion, from      */
/* memory]: fp64 compute, `z = z ? 1/z : 1`, no cheirality test, float output */
/* ------------------------------------------------------------------------- */
static inline void project_point(const double R[9], const double t[3], double fx, double fy,
                                 double cx, double cy, double X, double Y, double Z,
                                 float* u, float* v) {
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    *u = (float)(x * fx + cx);
    *v = (float)(y * fy + cy);
}

void esac_oracle_project(const double rvec[3], const double tvec[3], double fx, double fy,
                         double cx, double cy, const float* pts3, int n, float* uv) {
    double R[9];
    esac_oracle_rodrigues_vec2mat(rvec, R, NULL);
    for (int i = 0; i < n; i++)
        project_point(R, tvec, fx, fy, cx, cy, pts3[3 * i], pts3[3 * i + 1], pts3[3 * i + 2],
                      &uv[2 * i], &uv[2 * i + 1]);
}

/* cv::norm(Point2f) accumulates in double: sqrt((double)x*x + (double)y*y) */
static inline double norm2f(float dx, float dy) {
    return sqrt((double)dx * dx + (double)dy * dy);
}

/* ------------------------------------------------------------------------- */
/* LM pose refit = cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess=true)   */
/* [OpenCV cvFindExtrinsicCameraParams2 + CvLevMarq, from memory]:            */
/*   6 params, 2n residuals, max_iter 20, eps FLT_EPSILON, lambda = 10^k with */
/*   k from -3, diag(JtJ) *= 1+lambda, solve by SVD, k++ while the error got  */
/*   worse (<=16), k-- after an accepted step, stop on iteration count or     */
/*   ||param-prev||/||prev|| < eps.                                           */
/* ------------------------------------------------------------------------- */

/* Jacobi eigen-decomposition for symmetric n x n (n<=6), used as the SVD of the
 * (symmetric PSD) damped normal matrix. */
static void jacobi_sym(int n, double* A, double* w, double* V) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off == 0) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                if (!isfinite(theta)) t = 0; /* apq negligible */
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* x = pinv(A) b, singular values below 2*DBL_EPSILON*sum(w) dropped (cv::SVD::backSubst) */
static void solve_svd_sym6(const double A_in[36], const double b[6], double x[6]) {
    double A[36], w[6], V[36];
    memcpy(A, A_in, sizeof(A));
    jacobi_sym(6, A, w, V);
    double thresh = 0;
    for (int i = 0; i < 6; i++) thresh += fabs(w[i]);
    thresh *= 2 * DBL_EPSILON;
    for (int i = 0; i < 6; i++) x[i] = 0;
    for (int k = 0; k < 6; k++) {
        if (!(fabs(w[k]) > thresh)) continue;
        double d = 0;
        for (int i = 0; i < 6; i++) d += V[i * 6 + k] * b[i];
        d /= w[k];
        for (int i = 0; i < 6; i++) x[i] += V[i * 6 + k] * d;
    }
}

/* residuals (proj - m) and, if JtJ != NULL, JtJ (6x6) and JtErr (6) */
static double lm_eval(const float* obj, const float* img, int n, double fx, double fy, double cx,
                      double cy, const double param[6], double* JtJ, double* JtErr) {
    double R[9], dRdr[27];
    esac_oracle_rodrigues_vec2mat(param, R, JtJ ? dRdr : NULL);
    const double* t = param + 3;
    if (JtJ) {
        memset(JtJ, 0, 36 * sizeof(double));
        memset(JtErr, 0, 6 * sizeof(double));
    }
    double err2 = 0;
    for (int i = 0; i < n; i++) {
        double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        /* cvProjectPoints2 writes into the double `err` matrix, then cvSub(err, m) */
        double ex = (x * fx + cx) - (double)img[2 * i];
        double ey = (y * fy + cy) - (double)img[2 * i + 1];
        err2 += ex * ex + ey * ey;
        if (JtJ) {
            double Ju[6], Jv[6];
            /* dp/dt */
            Ju[3] = fx * z; Ju[4] = 0;      Ju[5] = -fx * x * z;
            Jv[3] = 0;      Jv[4] = fy * z; Jv[5] = -fy * y * z;
            /* dp/dr through dR/dr */
            for (int j = 0; j < 3; j++) {
                const double* dR = dRdr + 9 * j;
                double dx0 = X * dR[0] + Y * dR[1] + Z * dR[2];
                double dy0 = X * dR[3] + Y * dR[4] + Z * dR[5];
                double dz0 = X * dR[6] + Y * dR[7] + Z * dR[8];
                Ju[j] = fx * (z * (dx0 - x * dz0));
                Jv[j] = fy * (z * (dy0 - y * dz0));
            }
            for (int a = 0; a < 6; a++) {
                JtErr[a] += Ju[a] * ex + Jv[a] * ey;
                for (int b = a; b < 6; b++) JtJ[a * 6 + b] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
            }
        }
    }
    if (JtJ)
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < a; b++) JtJ[a * 6 + b] = JtJ[b * 6 + a];
    return sqrt(err2);
}

static void lm_step(const double JtJ[36], const double JtErr[6], int lambdaLg10,
                    const double prev[6], double param[6]) {
    double lambda = exp(lambdaLg10 * log(10.));
    double A[36], dx[6];
    memcpy(A, JtJ, sizeof(A));
    for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
    solve_svd_sym6(A, JtErr, dx);
    for (int i = 0; i < 6; i++) param[i] = prev[i] - dx[i];
}

int esac_oracle_lm_pnp(const float* obj, const float* img, int n, double fx, double fy,
                       double cx, double cy, double pose[6]) {
    const int max_iter = 20;
    const double epsilon = (double)FLT_EPSILON;
    double param[6], prev[6], JtJ[36], JtErr[6];
    memcpy(param, pose, sizeof(param));
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX;
    for (;;) {
        /* state CALC_J */
        double errAtParam = lm_eval(obj, img, n, fx, fy, cx, cy, param, JtJ, JtErr);
        memcpy(prev, param, sizeof(prev));
        lm_step(JtJ, JtErr, lambdaLg10, prev, param);
        if (iters == 0) prevErrNorm = errAtParam;
        /* state CHECK_ERR (possibly repeated with a larger lambda) */
        double errNorm;
        for (;;) {
            errNorm = lm_eval(obj, img, n, fx, fy, cx, cy, param, NULL, NULL);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    lm_step(JtJ, JtErr, lambdaLg10, prev, param);
                    continue;
                }
            }
            break;
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; i++) {
            dn += (param[i] - prev[i]) * (param[i] - prev[i]);
            pn += prev[i] * prev[i];
        }
        /* cvNorm(param, prevParam, CV_RELATIVE_L2) = ||a-b|| / (||b|| + DBL_EPSILON) */
        double rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
        if (++iters >= max_iter || rel < epsilon) break;
        prevErrNorm = errNorm;
    }
    memcpy(pose, param, sizeof(param));
    return iters;
}

/* ------------------------------------------------------------------------- */
/* pose2trans (esac_util.h:537-548): T = [R t; 0 1], returned INVERTED by a   */
/* generic LU inverse (cv::Mat::inv default DECOMP_LU, partial pivoting).     */
/* ------------------------------------------------------------------------- */
/* generic 4x4 inverse: LU with partial pivoting on the augmented matrix (cv::Mat::inv, DECOMP_LU) */
void esac_oracle_inv4(const double M[16], double Minv[16]) {
    double A[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) A[i][j] = j < 4 ? M[4 * i + j] : (j - 4 == i);
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (piv != c)
            for (int j = 0; j < 8; j++) { double tmp = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = tmp; }
        double d = 1. / A[c][c];
        for (int r = c + 1; r < 4; r++) {
            double f = A[r][c] * d;
            for (int j = c; j < 8; j++) A[r][j] -= f * A[c][j];
        }
    }
    for (int c = 3; c >= 0; c--) {
        double d = 1. / A[c][c];
        for (int j = 4; j < 8; j++) {
            double s = A[c][j];
            for (int k = c + 1; k < 4; k++) s -= A[c][k] * A[k][j];
            A[c][j] = s * d;
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) Minv[4 * i + j] = A[i][4 + j];
}

void esac_oracle_pose2trans(const double pose[6], double Tinv[16]) {
    double R[9], T[16];
    esac_oracle_rodrigues_vec2mat(pose, R, NULL);
    for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = pose[3 + i];
    }
    esac_oracle_inv4(T, Tinv);
}

/* ------------------------------------------------------------------------- */
/* Reference-level helpers                                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
    const esac_oracle_args* a;
    double fx, fy, cx, cy;
} ctx_t;

static inline float sc_at(const esac_oracle_args* a, int e, int c, int y, int x) {
    return a->scene_coords[e * a->sc_stride[0] + c * a->sc_stride[1] + y * a->sc_stride[2] + x * a->sc_stride[3]];
}
static inline int64_t assign_at(const esac_oracle_args* a, int h) {
    return a->hyp_assign[(int64_t)h * a->assign_stride];
}
/* createSampling, esac_util.h:64-66 (integer arithmetic) */
static inline void px_of(const esac_oracle_args* a, int x, int y, int* px, int* py) {
    *px = x * a->sub_sampling + a->sub_sampling / 2 - a->shift_x;
    *py = y * a->sub_sampling + a->sub_sampling / 2 - a->shift_y;
}

/* getReproErrs without Jacobian (esac_util.h:292-319,355-360): errs stored (y,x) */
static void repro_errs(const ctx_t* c, const double pose[6], int expert, float* errs) {
    const esac_oracle_args* a = c->a;
    double R[9];
    esac_oracle_rodrigues_vec2mat(pose, R, NULL);
    for (int x = 0; x < a->W; x++)
        for (int y = 0; y < a->H; y++) {
            int px, py;
            px_of(a, x, y, &px, &py);
            float u, v;
            project_point(R, pose + 3, c->fx, c->fy, c->cx, c->cy, sc_at(a, expert, 0, y, x),
                          sc_at(a, expert, 1, y, x), sc_at(a, expert, 2, y, x), &u, &v);
            float dx = (float)px - u, dy = (float)py - v; /* Point2f subtraction */
            float l = (float)norm2f(dx, dy);
            /* std::min(l, maxReproj) = (maxReproj < l) ? maxReproj : l (esac_util.h:358): a NaN error -- a non-finite scene
             * coordinate -- STAYS NaN, and with it the score of every hypothesis of this expert (esac_util.h:248-250) */
            errs[y * a->W + x] = a->max_reproj < l ? a->max_reproj : l;
        }
}

/* one term of getHypScores (esac_util.h:248-250) */
static inline double soft_inlier_term(float err, float thresh, float beta) {
    double softThreshold = beta * (err - thresh); /* float arithmetic, then widened */
    softThreshold = 1 / (1 + exp(-softThreshold));
    return 1 - softThreshold;
}

static double hyp_score_from_errs(const esac_oracle_args* a, const float* errs) {
    double s = 0;
    for (int x = 0; x < a->W; x++)
        for (int y = 0; y < a->H; y++) s += soft_inlier_term(errs[y * a->W + x], a->inlier_thresh, a->inlier_beta);
    float scale = a->inlier_alpha / a->W / a->H; /* float / int / int, esac_util.h:256 */
    s *= scale;
    return s;
}

/* one sampling try (esac_util.h:154-223). returns 1 if accepted */
static int sample_try(const ctx_t* c, int h, int expert, uint32_t t, int32_t xy[8], double pose[6]) {
    const esac_oracle_args* a = c->a;
    if (a->rng_mode == ESAC_RNG_CALLBACK) {
        int have = 0;
        while (have < 4) {
            int x = a->irand_cb(0, a->W - 1, a->irand_user);
            int y = a->irand_cb(0, a->H - 1, a->irand_user);
            int dup = 0;
            for (int j = 0; j < have; j++)
                if (xy[2 * j] == x && xy[2 * j + 1] == y) dup = 1;
            if (dup) continue;
            xy[2 * have] = x; xy[2 * have + 1] = y; have++;
        }
    } else {
        const uint32_t gh = a->hyp_index ? (uint32_t)a->hyp_index[h] : (uint32_t)h;
        esac_oracle_draw_cells(a->seed, a->call, gh, t, a->W, a->H, xy);
    }
    double obj[12], img[8];
    for (int j = 0; j < 4; j++) {
        int px, py;
        px_of(a, xy[2 * j], xy[2 * j + 1], &px, &py);
        img[2 * j] = (float)px; img[2 * j + 1] = (float)py;
        for (int k = 0; k < 3; k++) obj[3 * j + k] = sc_at(a, expert, k, xy[2 * j + 1], xy[2 * j]);
    }
    if (!esac_oracle_p3p(obj, img, c->fx, c->fy, c->cx, c->cy, pose, pose + 3)) {
        memset(pose, 0, 6 * sizeof(double)); /* safeSolvePnP failure, esac_util.h:107-111 */
        return 0;
    }
    double R[9];
    esac_oracle_rodrigues_vec2mat(pose, R, NULL);
    for (int j = 0; j < 4; j++) {
        float u, v;
        project_point(R, pose + 3, c->fx, c->fy, c->cx, c->cy, (float)obj[3 * j], (float)obj[3 * j + 1],
                      (float)obj[3 * j + 2], &u, &v);
        float dx = (float)img[2 * j] - u, dy = (float)img[2 * j + 1] - v;
        if (norm2f(dx, dy) < a->inlier_thresh) continue; /* double < float */
        return 0;
    }
    return 1;
}

/* softMax, entropy, draw(training=false): esac_util.h:461-530 */
static void soft_max(const double* scores, int n, double* sf) {
    double maxScore = 0;
    for (int i = 0; i < n; i++)
        if (i == 0 || scores[i] > maxScore) maxScore = scores[i];
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        sf[i] = exp(scores[i] - maxScore);
        sum += sf[i];
    }
    for (int i = 0; i < n; i++) sf[i] /= sum;
}
static double entropy(const double* dist, int n) {
    double e = 0;
    for (int i = 0; i < n; i++)
        if (dist[i] > 0) e -= dist[i] * log2(dist[i]);
    return e;
}
static int draw_argmax(const double* probs, int n) {
    double maxProb = -1;
    int maxIdx = 0;
    for (int idx = 0; idx < n; idx++) {
        if (probs[idx] < EPS_REF) continue;
        if (maxProb < 0 || probs[idx] > maxProb) {
            maxProb = probs[idx];
            maxIdx = idx;
        }
    }
    return maxIdx;
}

/* refineHyp, esac_util.h:378-454 */
static int refine_hyp(const ctx_t* c, const float* reproErrs, int expert, int maxRefSteps, double pose[6],
                      uint8_t* inlierMap, int32_t* counts, int32_t* lm_iters) {
    const esac_oracle_args* a = c->a;
    const int P = a->H * a->W;
    float* localErrs = (float*)malloc(sizeof(float) * P);
    float* localImg = (float*)malloc(sizeof(float) * 2 * P);
    float* localObj = (float*)malloc(sizeof(float) * 3 * P);
    uint8_t* localMap = (uint8_t*)malloc(P);
    memcpy(localErrs, reproErrs, sizeof(float) * P);
    if (inlierMap) memset(inlierMap, 0, P);
    unsigned bestInliers = 4;
    int accepted = 0;
    for (int rStep = 0; rStep < maxRefSteps; rStep++) {
        unsigned n = 0;
        memset(localMap, 0, P);
        for (int x = 0; x < a->W; x++)
            for (int y = 0; y < a->H; y++)
                if (localErrs[y * a->W + x] < a->inlier_thresh) {
                    int px, py;
                    px_of(a, x, y, &px, &py);
                    localImg[2 * n] = (float)px; localImg[2 * n + 1] = (float)py;
                    for (int k = 0; k < 3; k++) localObj[3 * n + k] = sc_at(a, expert, k, y, x);
                    localMap[y * a->W + x] = 1;
                    n++;
                }
        if (counts) counts[rStep] = (int32_t)n;
        if (n <= bestInliers) break; /* converged */
        bestInliers = n;
        /* n > 4 here, so the reference always takes SOLVEPNP_ITERATIVE; it returns true */
        int it = esac_oracle_lm_pnp(localObj, localImg, (int)n, c->fx, c->fy, c->cx, c->cy, pose);
        if (lm_iters) *lm_iters += it;
        if (inlierMap) memcpy(inlierMap, localMap, P);
        accepted++;
        repro_errs(c, pose, expert, localErrs);
    }
    free(localErrs); free(localImg); free(localObj); free(localMap);
    return accepted;
}

int esac_oracle_forward(esac_oracle_args* a) {
    if (!a || !a->scene_coords || !a->hyp_assign || a->N <= 0 || a->H < 3 || a->W < 3 || a->E <= 0) return -1;
    if ((int64_t)(a->W - 1) * (a->H - 1) < 4) return -2;
    if (a->rng_mode == ESAC_RNG_CALLBACK && !a->irand_cb) return -3;
    const int N = a->N, P = a->H * a->W;
    const int max_tries = a->max_tries > 0 ? a->max_tries : ESAC_ORACLE_MAX_TRIES;
    const int max_ref = a->max_ref_steps >= 0 ? a->max_ref_steps : ESAC_ORACLE_MAX_REF_STEPS;
    for (int h = 0; h < N; h++) {
        int64_t e = assign_at(a, h);
        if (e < 0 || e >= a->E) return -4;
    }
#ifdef _OPENMP
    int nthreads = a->num_threads > 0 ? a->num_threads : omp_get_max_threads();
    if (a->rng_mode == ESAC_RNG_CALLBACK) nthreads = 1; /* sequential stream */
#else
    int nthreads = 1;
#endif
    (void)nthreads;
    ctx_t c;
    c.a = a;
    /* camMat is a float matrix (esac.cpp:93-97) that OpenCV widens to double */
    c.fx = (double)a->focal; c.fy = (double)a->focal; c.cx = (double)a->ppx; c.cy = (double)a->ppy;

    double* hyps = (double*)calloc((size_t)N * 6, sizeof(double));
    int32_t* sxy = (int32_t*)calloc((size_t)N * 8, sizeof(int32_t));
    int32_t* tries = (int32_t*)calloc((size_t)N, sizeof(int32_t));
    double* scores = (double*)calloc((size_t)N, sizeof(double));
    double* probs = (double*)calloc((size_t)N, sizeof(double));
    /* the reference materialises one error image per hypothesis (esac.cpp:128-140);
       do the same while it fits, else fuse (identical arithmetic) */
    const int materialise = ((size_t)N * (size_t)P * sizeof(float)) <= ((size_t)1 << 30);
    float* errs = materialise ? (float*)malloc((size_t)N * P * sizeof(float)) : NULL;
    float* werrs = (float*)malloc((size_t)P * sizeof(float));
    double t0 = now_ms(), t1;

    /* ---- sampleHypotheses (esac_util.h:152-224) ---- */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int h = 0; h < N; h++) {
        if (a->in_hyps) { /* test hook: hypotheses handed in, no sampling */
            memcpy(hyps + 6 * h, a->in_hyps + 6 * h, 6 * sizeof(double));
            tries[h] = 0;
            continue;
        }
        int expert = (int)assign_at(a, h);
        int ok = 0;
        int t;
        for (t = 0; t < max_tries; t++) {
            ok = sample_try(&c, h, expert, (uint32_t)t, sxy + 8 * h, hyps + 6 * h);
            if (ok) break;
        }
        tries[h] = ok ? t : -1;
    }
    t1 = now_ms();
    if (a->out_phase_ms) a->out_phase_ms[0] = t1 - t0;
    t0 = t1;

    /* ---- getReproErrs + getHypScores (esac.cpp:131-147) ---- */
    if (materialise) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int h = 0; h < N; h++) repro_errs(&c, hyps + 6 * h, (int)assign_at(a, h), errs + (size_t)h * P);
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int h = 0; h < N; h++) scores[h] = hyp_score_from_errs(a, errs + (size_t)h * P);
    } else {
#pragma omp parallel num_threads(nthreads)
        {
            float* tmp = (float*)malloc((size_t)P * sizeof(float));
#pragma omp for schedule(static)
            for (int h = 0; h < N; h++) {
                repro_errs(&c, hyps + 6 * h, (int)assign_at(a, h), tmp);
                scores[h] = hyp_score_from_errs(a, tmp);
            }
            free(tmp);
        }
    }
    t1 = now_ms();
    if (a->out_phase_ms) a->out_phase_ms[1] = t1 - t0;
    t0 = t1;

    /* ---- softMax, entropy, draw (esac.cpp:153-155) ---- */
    soft_max(scores, N, probs);
    double ent = entropy(probs, N);
    int hypIdx = draw_argmax(probs, N);
    t1 = now_ms();
    if (a->out_phase_ms) a->out_phase_ms[2] = t1 - t0;
    t0 = t1;

    if (a->out_hyps) memcpy(a->out_hyps, hyps, (size_t)N * 6 * sizeof(double));

    /* ---- refineHyp on the winner only, serial (esac.cpp:167-177) ---- */
    int expertW = (int)assign_at(a, hypIdx);
    if (materialise)
        memcpy(werrs, errs + (size_t)hypIdx * P, (size_t)P * sizeof(float));
    else
        repro_errs(&c, hyps + 6 * hypIdx, expertW, werrs);
    if (a->out_winner_errs) memcpy(a->out_winner_errs, werrs, (size_t)P * sizeof(float));
    if (a->out_inlier_counts)
        for (int i = 0; i <= max_ref; i++) a->out_inlier_counts[i] = -1;
    if (a->out_lm_iters) *a->out_lm_iters = 0;
    int acc = refine_hyp(&c, werrs, expertW, max_ref, hyps + 6 * hypIdx, a->out_inlier_map, a->out_inlier_counts,
                         a->out_lm_iters);
    t1 = now_ms();
    if (a->out_phase_ms) a->out_phase_ms[3] = t1 - t0;

    /* ---- pose2trans -> outPose (esac.cpp:182-187) ---- */
    double T[16];
    esac_oracle_pose2trans(hyps + 6 * hypIdx, T);
    if (a->out_pose)
        for (int i = 0; i < 16; i++) a->out_pose[i] = (float)T[i];

    if (a->out_sample_xy) memcpy(a->out_sample_xy, sxy, (size_t)N * 8 * sizeof(int32_t));
    if (a->out_tries) memcpy(a->out_tries, tries, (size_t)N * sizeof(int32_t));
    if (a->out_scores) memcpy(a->out_scores, scores, (size_t)N * sizeof(double));
    if (a->out_probs) memcpy(a->out_probs, probs, (size_t)N * sizeof(double));
    if (a->out_entropy) *a->out_entropy = ent;
    if (a->out_winner) *a->out_winner = hypIdx;
    if (a->out_refined) memcpy(a->out_refined, hyps + 6 * hypIdx, 6 * sizeof(double));
    if (a->out_ref_steps) *a->out_ref_steps = acc;

    free(hyps); free(sxy); free(tries); free(scores); free(probs); free(errs); free(werrs);
    return expertW;
}

#include "esac_oracle_bwd.inc"
