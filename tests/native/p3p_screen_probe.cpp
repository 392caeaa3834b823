// p3p_screen_probe.cpp -- TEST-ONLY host build of the fp32 sampling screen (esac_amd/csrc/p3p_screen.hpp) next to the fp64
// route: random 4-cell tries on a given map, decision of the fp64 route (P3P + the 4-point tau test exactly as k_sample's
// accept_sample) vs the screen's smallest 4th-point error.  Built by tests/native/build.py:build_screen_probe();
// used by tests/test_device_math_host.py (the screen must never reject a try the fp64 route accepts) and by the
// calibration run scripts/dev/p3p_screen_probe.py.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

// which guard of the screen's private copy turned a try into "maybe" (p3p_screen.hpp: ESAC_SCREEN_STAT)
static thread_local long long g_screen_stat[10];
#define ESAC_SCREEN_STAT(k) (g_screen_stat[k]++)
// histograms of -log10 of selected ratios (calibration aid): ESAC_SCREEN_HIST(k, v)
static long long g_screen_hist[12][24];
#define ESAC_SCREEN_HIST(k, v)                                                           \
    do {                                                                                 \
        const double v_ = (v);                                                           \
        int bin_ = !(v_ > 0) ? 23 : (int)floor(-log10(v_)) + 2;                          \
        bin_ = bin_ < 0 ? 0 : bin_ > 22 ? 22 : bin_;                                     \
        __atomic_fetch_add(&g_screen_hist[k][bin_], 1, __ATOMIC_RELAXED);                \
    } while (0)
#include "../../esac_amd/csrc/p3p_screen.hpp"
#include "../../esac_amd/csrc/pose_math.hpp"
using namespace esac;

static inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static bool accept64(const double Rp[9], const double Tp[3], const float Pf[4][3], const double mu[4], const double mv[4], const Cam& cam, double tau) {
    double rvec[3], R[9];
    rodrigues_mat2vec(Rp, rvec);
    rodrigues_vec2mat<false>(rvec, R, nullptr);
    for (int j = 0; j < 4; j++) {
        const double Xd = Pf[j][0], Yd = Pf[j][1], Zd = Pf[j][2];
        double x = R[0] * Xd + R[1] * Yd + R[2] * Zd + Tp[0];
        double y = R[3] * Xd + R[4] * Yd + R[5] * Zd + Tp[1];
        double z = R[6] * Xd + R[7] * Yd + R[8] * Zd + Tp[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float u = (float)(x * cam.fx + cam.cx), v = (float)(y * cam.fy + cam.cy);
        const float dx = (float)mu[j] - u, dy = (float)mv[j] - v;
        if (!(sqrt((double)dx * dx + (double)dy * dy) < tau)) return false;
    }
    return true;
}

// out[0] tries, [1] accepted by fp64, [2] screen "delicate", [3] largest screen error among fp64-accepted tries,
// [4 + k] tries the screen keeps ("maybe") at margin margins[k], [12 + k] fp64-accepted tries it would REJECT there;
// [20] largest |screen err - fp64 4th-point err| over solved tries with fp64 err < 200 px; [21] fp64-solved tries;
// [30..38] how often guard 1..9 of the private copy (ESAC_SCREEN_STAT) fired
extern "C" void probe_screen(const float* coords, int H, int W, int sub, int shx, int shy, float f, float cx, float cy, float tau,
                             uint64_t seed, long long n_tries, const float* margins, int n_margins, int mode, double* out) {
    const int P = H * W;
    const Cam cam{(double)f, (double)f, (double)cx, (double)cy};
    double acc[40];
    memset(acc, 0, sizeof(acc));
#pragma omp parallel
    {
        double loc[40];
        memset(loc, 0, sizeof(loc));
        memset(g_screen_stat, 0, sizeof(g_screen_stat));
        uint64_t s = seed;
#ifdef _OPENMP
        s += 7919ull * (uint64_t)omp_get_thread_num();
#endif
#pragma omp for schedule(static)
        for (long long it = 0; it < n_tries; it++) {
            int cxs[4], cys[4];
            for (int j = 0; j < 4; j++) {
                for (;;) {
                    const uint64_t r = splitmix(s);
                    const int x = (int)((r & 0xffffffffu) % (uint32_t)(W - 1)), y = (int)((r >> 32) % (uint32_t)(H - 1));
                    bool dup = false;
                    for (int k = 0; k < j; k++) dup |= cxs[k] == x && cys[k] == y;
                    if (!dup) { cxs[j] = x; cys[j] = y; break; }
                }
            }
            float Pf[4][3], muf[4], mvf[4];
            V3 Pt[4];
            double mu[4], mv[4];
            for (int j = 0; j < 4; j++) {
                const int idx = cys[j] * W + cxs[j];
                Pf[j][0] = coords[idx]; Pf[j][1] = coords[P + idx]; Pf[j][2] = coords[2 * P + idx];
                Pt[j] = V3{(double)Pf[j][0], (double)Pf[j][1], (double)Pf[j][2]};
                muf[j] = (float)(cxs[j] * sub + sub / 2 - shx); mvf[j] = (float)(cys[j] * sub + sub / 2 - shy);
                mu[j] = muf[j]; mv[j] = mvf[j];
            }
            double Rp[9], Tp[3], reproj2 = 0;
            const bool solved = p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2);
            const bool ok = solved && accept64(Rp, Tp, Pf, mu, mv, cam, (double)tau);
            int reason = 0;
            float e;
            if (mode == 3 || mode == 7) {  // 7 = also print false rejects; what k_sample_prescreen / k_sample_screened run: the screen's private fast copy of the roots
                ScreenSetup S;
                e = screen_setup(Pt, mu, mv, cam, S) ? p3p_screen_roots(S, Pf, muf[3], mvf[3], f, cx, cy) : INFINITY;
                if (mode == 7 && ok && e != ESAC_SCREEN_MAYBE && !(e <= tau + 3.0f)) {
#pragma omp critical
                    {
                        printf("FALSE REJECT fast-copy screen e=%g fp64 e=%g\n", e, sqrt(reproj2));
                        for (int j = 0; j < 4; j++) printf("   Q %a %a %a %.1f %.1f\n", Pf[j][0], Pf[j][1], Pf[j][2], muf[j], mvf[j]);
                    }
                }
            } else {  // mode 1 / 2 (2 = also print false rejects): the exact route's own roots (p3p_setup), fp32 per-root work
                P3PSetup S;
                e = p3p_setup(Pt, mu, mv, cam, S) ? p3p_screen_roots(S, Pf, muf[3], mvf[3], f, cx, cy) : INFINITY;
            }
            if (reason > 0 && reason < 10) loc[22 + (reason > 5 ? 5 : reason)] += 1; else if (reason >= 10) loc[28] += 1;
            if (mode == 2 && ok && !(e <= tau + 20.0f) && e != ESAC_SCREEN_MAYBE) {
#pragma omp critical
                {
                    static int shown = 0;
                    if (shown++ < 12) {
                        P3PSetup S;
                        p3p_setup(Pt, mu, mv, cam, S);
                        const double t1 = S.p2 * (S.a - 1 + S.b), t2 = S.r2 * (S.a - 1 - S.b), t3 = S.pqr, t4 = S.a * S.pqr;
                        const double temp = t1 + t2 + t3 - t4;
                        printf("FALSE REJECT screen e=%g fp64 e=%g n=%d roots %g %g %g %g  a=%g b=%g p=%g q=%g r=%g temp=%g (terms %g) dist %g\n", e, sqrt(reproj2), S.n,
                               S.x[0], S.x[1], S.x[2], S.x[3], S.a, S.b, S.p, S.q, S.r, temp, fabs(t1) + fabs(t2) + fabs(t3) + fabs(t4), S.dist2);
                        for (int i = 0; i < S.n; i++) {
                            const double x = S.x[i], xx = x * x, v = (xx + 1 - S.q * x) / S.b, disc = S.p * S.p - 4 * (1 - S.a * v);
                            double R[9], T[3], rp = -1;
                            const bool val = p3p_candidate(S, x, Pt, mu[3], mv[3], cam, R, T, rp);
                            printf("   root %d x=%g v=%g disc=%g  fp64 candidate valid=%d reproj=%g\n", i, x, v, disc, (int)val, val ? sqrt(rp) : -1.0);
                        }
                    }
                }
            }
            loc[0] += 1;
            loc[1] += ok;
            const bool delicate = e == ESAC_SCREEN_MAYBE || !(e == e);
            loc[2] += delicate;
            if (ok && !delicate && e > loc[3]) loc[3] = e;
            for (int k = 0; k < n_margins && k < 8; k++) {
                const bool maybe = !(e > tau + margins[k]);
                loc[4 + k] += maybe;
                loc[12 + k] += ok && !maybe;
            }
            if (solved) {
                loc[21] += 1;
                const double e64 = sqrt(reproj2);
                if (!delicate && e64 < 200 && e < 1e30f) {
                    const double d = fabs((double)e - e64);
                    if (d > loc[20]) loc[20] = d;
                }
            }
        }
        for (int k = 1; k < 10; k++) loc[29 + k] = (double)g_screen_stat[k];  // out[30..38]: guards 1..9 of the private copy
#pragma omp critical
        {
            for (int k = 0; k < 40; k++) {
                if (k == 3 || k == 20) acc[k] = loc[k] > acc[k] ? loc[k] : acc[k];
                else acc[k] += loc[k];
            }
        }
    }
    memcpy(out, acc, sizeof(acc));  // out: 40 doubles; [23..27] delicate by quartic trigger 1..5, [28] other bail-outs
}

// the two quartic solvers side by side: out[0] = n of the exact route (quartic_real_roots), out[1..4] its roots,
// out[5] = n of the screen's fast copy (quartic_roots_fast; -1 = "not reproducible here": the screen reports maybe), out[6..9] its roots
extern "C" void probe_quartic(double a, double b, double c, double d, double e, double* out) {
    double x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
    out[0] = quartic_real_roots(a, b, c, d, e, x[0], x[1], x[2], x[3]);
    float rel[2] = {0, 0};
    out[5] = quartic_roots_fast(a, b, c, d, e, 0, 0, 0, 0, 0, y[0], y[1], y[2], y[3], rel[0], rel[1]);
    for (int i = 0; i < 4; i++) { out[1 + i] = x[i]; out[6 + i] = y[i]; }
}


extern "C" void probe_hist(long long* out, int clear) {
    memcpy(out, g_screen_hist, sizeof(g_screen_hist));
    if (clear) memset(g_screen_hist, 0, sizeof(g_screen_hist));
}

// ---- the sampling loop of k_sample on the host (device math compiled for the host, glibc's libm): first accepted try of
// hypothesis gh under the Philox stream (seed, call), -1 when the budget is spent.  Used to tell apart "the device's
// ALGORITHM differs from the oracle's" from "the device's libm / contraction differs" (scripts/dev/exact_route_probe.py).
static inline uint32_t umulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static void philox_block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = umulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = umulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static void draw_cells_host(uint64_t seed, uint64_t call, uint32_t hyp, uint32_t tr, int W, int H, int cx[4], int cy[4]) {
    const uint64_t key = seed + call * 0x9E3779B97F4A7C15ull;
    int have = 0;
    for (uint32_t k = 0; have < 4; k++) {
        uint32_t o[4];
        philox_block((uint32_t)key, (uint32_t)(key >> 32), hyp, tr, k, 0x45534143u, o);
        for (int half = 0; half < 2 && have < 4; half++) {
            const int x = (int)umulhi32(o[2 * half], (uint32_t)(W - 1)), y = (int)umulhi32(o[2 * half + 1], (uint32_t)(H - 1));
            bool dup = false;
            for (int j = 0; j < have; j++) dup |= cx[j] == x && cy[j] == y;
            if (!dup) { cx[have] = x; cy[have] = y; have++; }
        }
    }
}
extern "C" void probe_first_accept(const float* coords, int H, int W, int sub, float f, float cx, float cy, float tau, uint64_t seed, uint64_t call,
                                   const int* hyps, int n_hyps, int max_tries, int* tries_out) {
    const int P = H * W;
    const Cam cam{(double)f, (double)f, (double)cx, (double)cy};
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n_hyps; i++) {
        int found = -1;
        for (int t = 0; t < max_tries && found < 0; t++) {
            int cxs[4], cys[4];
            draw_cells_host(seed, call, (uint32_t)hyps[i], (uint32_t)t, W, H, cxs, cys);
            float Pf[4][3];
            V3 Pt[4];
            double mu[4], mv[4];
            for (int j = 0; j < 4; j++) {
                const int idx = cys[j] * W + cxs[j];
                Pf[j][0] = coords[idx]; Pf[j][1] = coords[P + idx]; Pf[j][2] = coords[2 * P + idx];
                Pt[j] = V3{(double)Pf[j][0], (double)Pf[j][1], (double)Pf[j][2]};
                mu[j] = (float)(cxs[j] * sub + sub / 2); mv[j] = (float)(cys[j] * sub + sub / 2);
            }
            double Rp[9], Tp[3], reproj2 = 0;
            if (p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2) && accept64(Rp, Tp, Pf, mu, mv, cam, (double)tau)) found = t;
        }
        tries_out[i] = found;
    }
}

// one sample under the microscope (scripts/dev/screen_adversarial.py: replaying a false reject): both routes' roots and,
// per root, the fp64 candidate's 4th-point error next to the screen's
extern "C" void probe_one(const float* pts /*4x3*/, const float* px /*4x2*/, float f, float cx, float cy) {
    const Cam cam{(double)f, (double)f, (double)cx, (double)cy};
    float Pf[4][3], muf[4], mvf[4];
    V3 Pt[4];
    double mu[4], mv[4];
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 3; k++) Pf[j][k] = pts[3 * j + k];
        Pt[j] = V3{(double)Pf[j][0], (double)Pf[j][1], (double)Pf[j][2]};
        muf[j] = px[2 * j]; mvf[j] = px[2 * j + 1];
        mu[j] = muf[j]; mv[j] = mvf[j];
    }
    P3PSetup E;
    const bool oke = p3p_setup(Pt, mu, mv, cam, E);
    printf("exact route: setup %d n=%d roots %.17g %.17g %.17g %.17g  a=%g b=%g p=%g q=%g r=%g\n", (int)oke, E.n, E.x[0], E.x[1], E.x[2], E.x[3], E.a, E.b, E.p, E.q, E.r);
    ScreenScene sc;
    const bool scok = screen_scene(Pf, sc);
    const float mu3[3] = {(float)E.mu[0], (float)E.mu[1], (float)E.mu[2]}, mv3[3] = {(float)E.mv[0], (float)E.mv[1], (float)E.mv[2]},
                mk3[3] = {(float)E.mk[0], (float)E.mk[1], (float)E.mk[2]};
    for (int i = 0; oke && i < E.n; i++) {
        double R[9], T[3], rp = -1, X, Y, Z;
        const bool val = p3p_candidate(E, E.x[i], Pt, mu[3], mv[3], cam, R, T, rp);
        const bool len = p3p_candidate_lengths(E, E.x[i], X, Y, Z);
        const float es = (scok && len) ? screen_candidate(sc, mu3, mv3, mk3, (float)X, (float)Y, (float)Z, muf[3], mvf[3], f, cx, cy, ESAC_SCREEN_CONGRUENCE) : -2.f;
        printf("   root %d x=%.17g  fp64 candidate valid=%d 4th-point err %.6f | lengths %d X=%.9g Y=%.9g Z=%.9g screen err %.6f\n", i, E.x[i], (int)val,
               val ? sqrt(rp) : -1.0, (int)len, X, Y, Z, es);
    }
    ScreenSetup S;
    const bool oks = screen_setup(Pt, mu, mv, cam, S);
    printf("fast copy : setup %d n=%d roots %.17g %.17g %.17g %.17g  dx01=%g dx23=%g\n", (int)oks, S.n, S.x[0], S.x[1], S.x[2], S.x[3], (double)S.dx01, (double)S.dx23);
    for (int i = 0; oks && i < S.n; i++) {
        double X = 0, Y = 0, Z = 0;
        const int ok = screen_lengths(S, S.x[i], i < 2 ? S.dx01 : S.dx23, X, Y, Z);
        const float es = (scok && ok == 1) ? screen_candidate(sc, mu3, mv3, mk3, (float)X, (float)Y, (float)Z, muf[3], mvf[3], f, cx, cy, ESAC_SCREEN_CONGRUENCE) : -2.f;
        printf("   root %d x=%.17g  lengths %d X=%.9g Y=%.9g Z=%.9g screen err %.6f\n", i, S.x[i], ok, X, Y, Z, es);
    }
    printf("screen on the fast copy: %g ; on the exact roots: %g\n", oks ? p3p_screen_roots(S, Pf, muf[3], mvf[3], f, cx, cy) : -1.f,
           oke ? p3p_screen_roots(E, Pf, muf[3], mvf[3], f, cx, cy) : -1.f);
    double Rp[9], Tp[3], reproj2 = 0;
    const bool solved = p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2);
    printf("p3p_4pt: solved %d 4th-point err %.6f accept64 %d\n", (int)solved, sqrt(reproj2), (int)(solved && accept64(Rp, Tp, Pf, mu, mv, cam, 10.0)));
}

// the same as numbers: out[0] screen on the private fast copy, out[1] screen on the exact route's roots (ESAC_SCREEN_MAYBE = -1),
// out[2] 4th-point error of the fp64 route's pose, out[3] 1 if the fp64 route accepts the try at tau
extern "C" void probe_one_values(const float* pts, const float* px, float f, float cx, float cy, float tau, double* out) {
    const Cam cam{(double)f, (double)f, (double)cx, (double)cy};
    float Pf[4][3], muf[4], mvf[4];
    V3 Pt[4];
    double mu[4], mv[4];
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 3; k++) Pf[j][k] = pts[3 * j + k];
        Pt[j] = V3{(double)Pf[j][0], (double)Pf[j][1], (double)Pf[j][2]};
        muf[j] = px[2 * j]; mvf[j] = px[2 * j + 1];
        mu[j] = muf[j]; mv[j] = mvf[j];
    }
    P3PSetup E;
    ScreenSetup S;
    out[0] = screen_setup(Pt, mu, mv, cam, S) ? (double)p3p_screen_roots(S, Pf, muf[3], mvf[3], f, cx, cy) : INFINITY;
    out[1] = p3p_setup(Pt, mu, mv, cam, E) ? (double)p3p_screen_roots(E, Pf, muf[3], mvf[3], f, cx, cy) : INFINITY;
    double Rp[9], Tp[3], reproj2 = 0;
    const bool solved = p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2);
    out[2] = solved ? sqrt(reproj2) : -1.0;
    out[3] = solved && accept64(Rp, Tp, Pf, mu, mv, cam, (double)tau) ? 1.0 : 0.0;
}
