// host_math_probe.cpp -- TEST-ONLY: compiles the device math header (pose_math.hpp) for the host so the
// CPU test-suite (-m "not gpu") can exercise the exact source the kernels use against the oracle.
// Built by tests/native/build.py with hipcc (host pass only is used). Not part of the product library.
#include "../../esac_amd/csrc/pose_math.hpp"
#include "../../esac_amd/csrc/lm_math.hpp"
#include "../../esac_amd/csrc/bwd_math.hpp"
#include "../../esac_amd/csrc/lm_lanes.hpp"
using namespace esac;
extern "C" {
int probe_p3p(const double* obj, const double* img, double fx, double fy, double cx, double cy, double* rvec, double* tvec, double* Rout) {
    V3 P[4]; double mu[4], mv[4];
    for (int j = 0; j < 4; j++) { P[j] = V3{obj[3*j], obj[3*j+1], obj[3*j+2]}; mu[j] = img[2*j]; mv[j] = img[2*j+1]; }
    Cam cam{fx, fy, cx, cy};
    double R[9], T[3];
    if (!p3p_4pt(P, mu, mv, cam, R, T)) return 0;
    rodrigues_mat2vec(R, rvec);
    for (int i = 0; i < 3; i++) tvec[i] = T[i];
    if (Rout) for (int i = 0; i < 9; i++) Rout[i] = R[i];
    return 1;
}
int probe_quartic(double a, double b, double c, double d, double e, double* roots) {
    return quartic_real_roots(a, b, c, d, e, roots[0], roots[1], roots[2], roots[3]);
}
void probe_rodrigues(const double* r, double* R, double* J) { rodrigues_vec2mat<true>(r, R, J); }
void probe_mat2vec(const double* R, double* r) { rodrigues_mat2vec(R, r); }
float probe_exact_err(const double* R, const double* t, double fx, double fy, double cx, double cy, float X, float Y, float Z, float px, float py) {
    Cam cam{fx, fy, cx, cy};
    return project_exact_err(R, t, cam, X, Y, Z, px, py);
}
void probe_lm_solve6(const double* U21, const double* g, double lambda, double* dx) { lm_solve6(U21, g, lambda, dx); }
// normal equations of one LM iteration exactly as k_refine builds them (twist-space sums + chain rule)
void probe_lm_normal(const float* obj, const float* img, int n, const double* pose, double fx, double fy, double cx, double cy,
                     double* U21, double* g6, double* e2) {
    Cam cam{fx, fy, cx, cy};
    double R[9], dRdr[27], acc[LM_NACC];
    rodrigues_vec2mat<true>(pose, R, dRdr);
    for (int k = 0; k < LM_NACC; k++) acc[k] = 0;
    for (int i = 0; i < n; i++)
        lm_accumulate_point<true>(R, pose + 3, cam, obj[3*i], obj[3*i+1], obj[3*i+2], img[2*i], img[2*i+1], acc);
    lm_to_rvec_space(acc, R, dRdr, pose + 3, U21, g6);
    *e2 = acc[26];
}
// the same normal equations through the kernels' own route: closed-form chain + monomial moments (fx == fy)
void probe_lm_normal_moments(const float* obj, const float* img, int n, const double* pose, double f, double cx, double cy,
                             double* U21, double* g6, double* e2) {
    Cam cam{f, f, cx, cy};
    double R[9], mom[LM_NMOM], acc[LM_NACC];
    LmChain ch;
    lm_pose_chain(pose, R, ch);
    for (int k = 0; k < LM_NMOM; k++) mom[k] = 0;
    for (int i = 0; i < n; i += 2) {  // pairs, the second one switched off past the end: exercises the 0/1 weight
        double X[2], Y[2], Z[2], mx[2], my[2];
        bool on[2];
        for (int p = 0; p < 2; p++) {
            const int j = i + p < n ? i + p : i;
            on[p] = i + p < n;
            X[p] = obj[3*j]; Y[p] = obj[3*j+1]; Z[p] = obj[3*j+2]; mx[p] = img[2*j]; my[p] = img[2*j+1];
        }
        LmTerms<2> t;
        lm_point_terms<2>(R, pose + 3, cam, X, Y, Z, mx, my, on, t);
        lm_accumulate_moments<2>(t, mom);
    }
    lm_moments_to_acc(mom, f, acc);
    lm_transform(acc, ch, U21, g6);
    *e2 = acc[26];
}
// lm_transform_t (K folded in) against lm_transform on the same twist-space sums: max |difference| relative to the largest entry
double probe_transform_t_diff(const double* acc, const double* pose) {
    double R[9], U1[21], g1[6], U2[21], g2[6];
    LmChain ch;
    lm_pose_chain(pose, R, ch);
    lm_transform(acc, ch, U1, g1);
    lm_transform_t(acc, ch.Mw, pose + 3, U2, g2);
    double d = 0, mU = 0, mg = 0;
    for (int k = 0; k < 21; k++) mU = fmax(mU, fabs(U1[k]));
    for (int k = 0; k < 6; k++) mg = fmax(mg, fabs(g1[k]));
    for (int k = 0; k < 21; k++) d = fmax(d, fabs(U1[k] - U2[k]) / mU);
    for (int k = 0; k < 6; k++) d = fmax(d, fabs(g1[k] - g2[k]) / mg);
    return d;
}
// The lane-dealt serial section of an LM round (lm_lanes.hpp, round 5) on the 16-lane row emulation: the 27 totals of a pass
// (24 moments | ...) and a pose in, the (rvec, tvec)-space system it builds (U21, g6 through lm_lane_to_u21), the step of
// its Gauss-Jordan solve and its pivot verdict out.  The test compares with lm_moments_to_acc + lm_transform + lm_solve6.
int probe_lane_step(const double* sums27, const double* pose, double lambda, double* U21, double* g6, double* dx) {
    double lds[64];
    for (int k = 0; k < 64; k++) lds[k] = 0.0;
    for (int k = 0; k < 27; k++) { lds[k] = sums27[k]; lds[32 + k] = -sums27[k]; }
    Row16 X[3], Y[3], hot[6], keep, M[3], K[3], c[6], dg, U[21], g[6], d[6];
    for (int l = 0; l < 16; l++) {
        for (int k = 0; k < 3; k++) { X[k].l[l] = lds[lm_lane_slot(l, k)]; Y[k].l[l] = lds[lm_lane_slot(l, 3 + k)]; }
        for (int k = 0; k < 6; k++) hot[k].l[l] = l == k ? 1.0 : 0.0;
        keep.l[l] = (l >= 3 && l <= 6) ? 1.0 : 0.0;
    }
    double R[9];
    LmTrig tg;
    lm_pose_rotation(pose, R, tg);
    lm_lane_chain<Row16>(tg, pose + 3, hot, M, K);
    lm_lane_transform<Row16>(X, Y, M, K, hot, keep, c, dg);
    lm_lane_to_u21<Row16>(c, U, g);
    for (int k = 0; k < 21; k++) U21[k] = U[k].l[9];  // (any lane: uniform)
    for (int k = 0; k < 6; k++) g6[k] = g[k].l[9];
    const bool ok = lm_lane_solve<Row16>(c, dg, hot, lambda, d);
    for (int k = 0; k < 6; k++) dx[k] = d[k].l[11];
    return ok ? 1 : 0;
}
// closed-form chain (lm_pose_chain) vs the chain built from dR/drvec (lm_chain): max |difference| over R, Mw, K
double probe_chain_diff(const double* pose) {
    double R1[9], dRdr[27], R2[9];
    LmChain a, b;
    rodrigues_vec2mat<true>(pose, R1, dRdr);
    lm_chain(R1, dRdr, pose + 3, a);
    lm_pose_chain(pose, R2, b);
    double d = 0;
    for (int k = 0; k < 9; k++) d = fmax(d, fabs(R1[k] - R2[k]));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            d = fmax(d, fabs(a.Mw[i][j] - b.Mw[i][j]));
            d = fmax(d, fabs(a.K[i][j] - b.K[i][j]) / (1.0 + fabs(a.K[i][j])));
        }
    return d;
}
void probe_pose_chain(const double* pose, double* R, double* Mw, double* K) {
    LmChain ch;
    lm_pose_chain(pose, R, ch);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { Mw[3 * i + j] = ch.Mw[i][j]; K[3 * i + j] = ch.K[i][j]; }
}
// ---- training-path routines (bwd_math.hpp)
double probe_pose_loss(const double* pose, const double* gt16, double wR, double wT, double cut) { return pose_loss(pose, gt16, wR, wT, cut); }
void probe_pose_dloss(const double* est, const double* gt6, double wR, double wT, double cut, double* jac) { pose_dloss(est, gt6, wR, wT, cut, jac); }
void probe_dproject_dobj(float ptx, float pty, float ox, float oy, float oz, const double* rvec, const double* t, float focal,
                         float ppx, float ppy, float max_reproj, double* out) {
    double R[9];
    rodrigues_vec2mat<false>(rvec, R, nullptr);
    dproject_dobj(ptx, pty, ox, oy, oz, R, t, focal, ppx, ppy, max_reproj, out);
}
int probe_norm_jac_row(const double* rvec, const double* t, float focal, float ppx, float ppy, float X, float Y, float Z, float px,
                       float py, float max_reproj, double* row) {
    Cam cam{(double)focal, (double)focal, (double)ppx, (double)ppy};
    double R[9], dRdr[27];
    rodrigues_vec2mat<true>(rvec, R, dRdr);
    return norm_jac_row(R, dRdr, t, cam, X, Y, Z, px, py, max_reproj, row) ? 1 : 0;
}
int probe_inv_spd6(const double* U21, double* Ainv) { return inv_spd6(U21, Ainv) ? 1 : 0; }
void probe_pinv_sym6(const double* U21, double* Ainv) { pinv_sym6_jacobi(U21, Ainv); }
}
