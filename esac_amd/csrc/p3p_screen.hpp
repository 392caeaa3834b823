// p3p_screen.hpp -- fp32 SCREEN for one sampling try (esac_util.h:152-223): can this try possibly be accepted?
//
// A hypothesis on a wrong expert's map needs ~10^3 tries before four random cells happen to be consistent with one
// pose; each try is a 4-point P3P solve in fp64 (pose_math.hpp: ~700-3000 dependent fp64 operations at ~40 cycles of
// latency each) whose 4th point then misses tau by hundreds of pixels.  The screen runs the same algebra in fp32 --
// dependent fp32 operations cost a fraction of the fp64 latency and need half the registers, so several wavefronts fit a
// SIMD -- and answers a ONE-SIDED question: `false` only when every candidate pose puts the 4th point so far from its
// pixel that no rounding of the fp32 route can bridge the gap; `true` ("maybe") for everything else, including every
// numerically delicate configuration (near-double roots, tiny pivots, non-finite intermediates).  A "maybe" try is then
// decided by the fp64 route exactly as before, so the accepted try -- the one the reference's sequential loop stops at --
// is unchanged; only the work spent on hopeless tries shrinks.  The margin and the delicate-case triggers were
// calibrated against the fp64 route on 10^8+ tries of all frame kinds (scripts/dev/p3p_screen_probe.cpp,
// tests/test_device_math_host.py::test_fp32_screen_never_rejects_an_accepted_try).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#include "pose_math.hpp"

namespace esac {

struct V3f {
    float x, y, z;
};
ESAC_HD V3f operator-(V3f a, V3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
ESAC_HD V3f operator*(float s, V3f a) { return {s * a.x, s * a.y, s * a.z}; }
ESAC_HD float dotf(V3f a, V3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ESAC_HD V3f crossf(V3f a, V3f b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

#define ESAC_SCREEN_MAYBE (-1.0f)
constexpr float ESAC_SCREEN_CONGRUENCE = 1e-3f;

// ---- the screen's private copy of the roots and depths ------------------------------------------------------------------------
// p3p_setup / p3p_candidate_lengths (pose_math.hpp) follow the CPU solver operation by operation -- IEEE mul and add, the
// library's acos / cos / pow -- because the DECISION must round like the reference's on ill-conditioned samples.  The
// screen only needs the same roots and depths to ~1e-13: a "maybe" is decided by that exact route anyway, and a rejected
// try has its 4th point hundreds of pixels out.  So this copy lets the compiler contract mul + add into FMAs (a quarter
// fewer fp64 instructions in the coefficient polynomials) and replaces the fp64 library calls of the resolvent cubic
// (~300 dependent instructions for acos + cos, or pow) by an fp32 seed polished with Newton steps in fp64.  Candidate
// validity (x > 0, b1 > 0, v > 0) can differ from the exact route only where those quantities are within ~1e-13 of
// zero relative to their terms; the calibration probe runs THIS code against the exact route.
// 1/d and sqrt(d) to ~1 ulp without the IEEE division / square-root sequences (v_div_scale + v_div_fmas + v_div_fixup,
// the exponent-scaled v_rsq iteration: ~10-15 dependent instructions each, ~40 of them per try): reciprocal / reciprocal
// square root estimate + Newton steps.  Arguments here are squared lengths, cosines and polynomial coefficients of
// ordinary magnitude; zero, negative and non-finite arguments give what the callers' tests expect (inf / NaN / 0).
ESAC_HD double scr_rcp(double d) { return fast_rcp(d); }
// 1 / v, 1 / sqrt(v), sqrt(v) in single precision to the hardware's 1 ulp: the bookkeeping of the bounds and the fp32
// geometry of the screen (triads, projection) need no correct rounding, and the IEEE sequences cost ~10 instructions each
ESAC_HD float scr_rcpf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(v);
#else
    return 1.0f / v;
#endif
}
ESAC_HD float scr_rsqf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(v);
#else
    return 1.0f / sqrtf(v);
#endif
}
ESAC_HD float scr_sqrtf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(v);
#else
    return sqrtf(v);
#endif
}
ESAC_HD double scr_sqrt(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(fast)
    if (!(d > 0)) return d == 0 ? 0.0 : sqrt(d);  // 0 -> 0, negative / NaN -> NaN as the library
    double y = __builtin_amdgcn_rsq(d);            // ~1e-8 relative
    y = y * (1.5 - 0.5 * d * y * y);               // Newton on 1/sqrt: ~1e-16
    double s = d * y;
    s = __builtin_fma(0.5 * y, __builtin_fma(-s, s, d), s);  // one correction of the root itself
    return s;
#else
    return sqrt(d);
#endif
}

struct ScreenSetup {
    double mu[3], mv[3], mk[3];
    double dist2, a, b, p, q, r, inv_b0;
    double x[4];
    float dx01, dx23;  // bound on the difference between roots (0, 1) / (2, 3) here and in the exact route (quartic_roots_fast)
    float sf, m136;    // 1 + a + b and 136 (1 + a + b)^2: magnitudes of the terms of b1's factors (screen_lengths)
    int n;
};

// Host builds of the probe count which guard turned a try into "maybe" (tests/native/p3p_screen_probe.cpp defines the macro)
#ifndef ESAC_SCREEN_STAT
#define ESAC_SCREEN_STAT(k)
#endif
#ifndef ESAC_SCREEN_HIST
#define ESAC_SCREEN_HIST(k, v)
#endif

// cos(acos(c) / 3), |c| <= 1: fp32 seed, Newton on 4u^3 - 3u - c = 0 (the seed is ~1e-7 off: two steps reach rounding);
// next to c = -1 (u = 1/2 is a double root there) the library route is kept
ESAC_HD double cos_third_acos(double c) {
#pragma clang fp contract(fast)
    if (c < -0.999) return cos(acos(c) * (1.0 / 3.0));
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cos_f32 on an argument in [0, pi/3]: the library cosf carries a large-argument reduction with a stack array
    double u = (double)__cosf(acosf((float)c) * (1.0f / 3.0f));
#else
    double u = (double)cosf(acosf((float)c) * (1.0f / 3.0f));
#endif
#pragma unroll
    for (int it = 0; it < 3; it++) {  // the fast fp32 cosine seeds to ~1e-6: three steps reach rounding with margin
        const double u2 = u * u;
        const double fv = (4.0 * u2 - 3.0) * u - c, df = 12.0 * u2 - 3.0;
        u -= fv * scr_rcp(df);
    }
    return u;
}
// a^(1/3), a > 0: fp32 seed, two Newton steps on u^3 = a
ESAC_HD double cbrt_pos(double a) {
#pragma clang fp contract(fast)
    if (!(a > 1e-300 && a < 1e300)) return pow(a, 1.0 / 3.0);
    // scale into float range through the exponent: a = m * 8^k
#if defined(__HIP_DEVICE_COMPILE__)
    const int ex = __builtin_amdgcn_frexp_exp(a);  // a = m * 2^ex, m in [0.5, 1); no out-parameter (= no stack slot)
    const double m = __builtin_amdgcn_frexp_mant(a);
#else
    int ex;
    const double m = frexp(a, &ex);
#endif
    const int k = (ex - (ex < 0 ? 2 : 0)) / 3;  // floor-ish division by 3
    const double ms = ldexp(m, ex - 3 * k);  // in [2^-3, 2^3)
    double u = (double)cbrtf((float)ms);
#pragma unroll
    for (int it = 0; it < 2; it++) u -= (u * u * u - ms) * scr_rcp(3.0 * u * u);
    return ldexp(u, k);
}

// ---- how far can this copy be from the exact route? -----------------------------------------------------------------------------
// Both routes evaluate the same formulas on the same inputs; they differ in ROUNDING: IEEE mul / add and the library's
// acos / cos / pow there, contracted FMAs, Newton reciprocals / roots and a polished fp32 seed here.  The screen may only
// judge a try when that difference cannot change a discrete outcome (number of real roots, validity of a candidate) or
// move a depth by more than the screening margin absorbs.  So every quantity a decision hangs on carries a FIRST-ORDER
// BOUND of the difference between two correctly-ordered evaluations of it:
//   * a quartic coefficient: SCREEN_CU * (sum of the magnitudes of the terms it is summed from) -- SCREEN_CU = 64 eps
//     covers the <= 16 roundings of either evaluation plus the few-eps relative differences of a, b, p, q, r themselves;
//   * propagated through the normalisation, the resolvent cubic's coefficients, its root r0 (residual bound / slope:
//     the conditioning of the root, arbitrarily bad next to a multiple root), Ferrari's R^2, D^2, E^2 -- including the
//     cancellation INSIDE 4bc - 8d - b^3 and the 1 / R amplification -- down to the roots x and, in screen_lengths, to
//     b1(x) and the depth ratio y = b1 / b0.
// Where R^2, D^2, E^2, a root's sign or b1's sign is not SCREEN_SIG times larger than its bound, or a root / depth ratio
// is not reproducible to SCREEN_REL, the try is "maybe" (decided by the exact route).  The probe
// (tests/native/p3p_screen_probe.cpp, scripts/dev/screen_adversarial.py) runs this code against the exact route on
// planar / fronto-parallel / spherical / warped maps where every sample is a near-double-root configuration.
constexpr double SCREEN_CU = (2 * 2.220446049250313e-16);
constexpr double SCREEN_SIG = 250.0;
constexpr double SCREEN_REL = 1e-2;
constexpr double SCREEN_FERRARI_SAFETY = 1e3;
// real roots of the quartic, Ferrari through the first real root of the resolvent cubic: the branch structure of
// quartic_real_roots / cubic_first_roots (pose_math.hpp), contracted arithmetic, fast cubic root.
// ua..ue: bounds of the coefficients' differences between the two routes; dx01 / dx23: bound of the difference of the
// roots of each pair (they share it: x = +-R/2 +- sqrt(D2)/2 - b/4).
// Returns the number of real roots, or -1: "not reproducible here" (the caller reports maybe).
ESAC_HD int quartic_roots_fast(double a, double b, double c, double d, double e, float ua, float ub, float uc, float ud, float ue,
                               double& x0, double& x1, double& x2, double& x3, float& dx01, float& dx23) {
#pragma clang fp contract(fast)
    // (the verdicts of the guards are collected in `bad` and returned once: every early exit is a save / restore of the
    // execution mask and a branch in the scalar unit, which this two-wavefronts-per-SIMD kernel pays in full)
    bool bad = !((float)fabs(a) > (float)SCREEN_SIG * ua);  // (a == 0 included) the leading coefficient carries no digits
    const double inv_a = scr_rcp(a);
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    // The bounds are bookkeeping, not arithmetic: single precision (half the registers; an overflow ends as "maybe")
    const float ia = (float)fabs(inv_a);
    const float fb = (float)fabs(b), fc = (float)fabs(c), fd = (float)fabs(d), fe = (float)fabs(e), fb2 = fb * fb;
    // normalised coefficients: (uB + |b| uA) / |A|
    const float rua = ua * ia;
    ub = ub * ia + fb * rua; uc = uc * ia + fc * rua; ud = ud * ia + fd * rua; ue = ue * ia + fe * rua;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    // resolvent: y^3 - c y^2 + (d b - 4 e) y + (4 c e - d^2 - b^2 e)
    const double cb = -c, cc = d * b - 4 * e, cd = 4 * c * e - d * d - b2 * e;
    const float eps4 = 8.9e-16f;
    const float ucc = fd * ub + fb * ud + 4 * ue + eps4 * (fd * fb + 4 * fe);
    const float ucd = 4 * (fc * ue + fe * uc) + 2 * fd * ud + 2 * fb * fe * ub + fb2 * ue + eps4 * (4 * fc * fe + fd * fd + fb2 * fe);
    const double Q = (3 * cc - cb * cb) * (1. / 9.), R = (9 * cb * cc - 27 * cd - 2 * cb * cb * cb) * (1. / 54.);
    const double Q3 = Q * Q * Q, D = Q3 + R * R;
    const double cb3 = (1. / 3.) * cb;
    double r0;
    if (Q == 0) {
        if (R == 0) r0 = -cb3;
        else r0 = (R > 0 ? cbrt_pos(2 * R) : -cbrt_pos(-2 * R)) - cb3;  // pow(2R, 1/3) is NaN for R < 0 in the exact route: rare, harmless here
    } else if (D <= 0) {
        const double sq = scr_sqrt(-Q3);
        double arg = R * scr_rcp(sq);
        arg = arg > 1. ? 1. : (arg < -1. ? -1. : arg);
        r0 = 2 * scr_sqrt(-Q) * cos_third_acos(arg) - cb3;
    } else {
        const double AD = cbrt_pos(fabs(R) + scr_sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
        const double BD = (AD == 0) ? 0 : -Q * scr_rcp(AD);
        r0 = AD + BD - cb3;
    }
    // The resolvent root.  er: what the two closed-form evaluations (the library's acos / cos / pow there, a
    // Newton-polished fp32 seed here) do to it -- they agree to the conditioning of the root, (rounding of the cubic at
    // r0) / (its slope there); ec: what the coefficient differences do to it, through the same slope.  Both blow up next
    // to a multiple root of the cubic, which is where a double root of the quartic puts it.
    const float fr = (float)fabs(r0);
    const float islope = scr_rcpf((float)fabs((3 * r0 + 2 * cb) * r0 + cc));  // slope 0: inf -> "maybe" below
    const float er = 4e-16f * (((fr + fc) * fr + (float)fabs(cc)) * fr + (float)fabs(cd)) * islope;
    const float ec = ((uc * fr + ucc) * fr + ucd) * islope;
    // Ferrari takes square roots of quantities that cancel: R2 = (half the difference of the two quadratic factors'
    // linear terms)^2, D2 and E2 = squared separations of the root pairs, with v ~ 1 / sqrt(R2)
    const double R2 = 0.25 * b2 - c + r0;
    const float uR2 = ec + 0.5f * fb * ub + uc;  // coefficient part of R2's bound
    const float fR2 = (float)fabs(R2);
    ESAC_SCREEN_HIST(0, er / fR2);
    ESAC_SCREEN_HIST(1, uR2 / fR2);
    bad |= !(fR2 > (float)(SCREEN_FERRARI_SAFETY * 10) * er + (float)SCREEN_SIG * uR2);
    if (R2 < 0) return bad ? -1 : 0;
    const double Rr = scr_sqrt(R2), iRr = scr_rcp(Rr);
    const double num = 4 * bc - 8 * d - b3;
    const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * num * iRr;
    const double D2 = u + v, E2 = u - v;
    const float fiRr = (float)iRr, fiR2 = fiRr * fiRr, fv = (float)fabs(v);
    const float unum = 4 * (fc * ub + fb * uc) + 8 * ud + 3 * fb2 * ub + eps4 * (4 * fb * fc + 8 * fd + fb * fb2);
    const float uD = 1.5f * fb * ub + 2 * uc + uR2 + 0.25f * unum * fiRr + 0.5f * fv * uR2 * fiR2;  // coefficient part of D2's / E2's bound
    const float rD = er * (1 + 0.5f * fv * fiR2) + 1e-16f * (0.75f * fb2 + 2 * fc + fR2 + fv);       // rounding part
    const float fD2 = (float)fabs(D2), fE2 = (float)fabs(E2);
    ESAC_SCREEN_HIST(2, rD / fminf(fD2, fE2));
    ESAC_SCREEN_HIST(3, uD / fminf(fD2, fE2));
    {
        const float dv = (float)SCREEN_FERRARI_SAFETY * rD + (float)SCREEN_SIG * uD;
        bad |= !(fminf(fD2, fE2) > dv);
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * Rr;
    // bound of a root's difference: x = +-R/2 +- sqrt(D2)/2 - b/4
    const float dxR = 0.25f * (uR2 + er) * fiRr + 0.25f * ub;
    // scalars, not an indexed array: a run-time index (or an array the optimiser cannot split) ends up in scratch memory
    int nb = 0;
    if (D2 >= 0) {
        const double Ds = scr_sqrt(D2);
        x0 = R_2 + 0.5 * Ds - b_4;
        x1 = x0 - Ds;
        dx01 = dxR + 0.25f * (uD + rD) * scr_rcpf((float)Ds);
        nb = 2;
    }
    if (E2 >= 0) {
        const double Es = scr_sqrt(E2);
        const double xa = -R_2 + 0.5 * Es - b_4, xb = xa - Es;
        const float dx = dxR + 0.25f * (uD + rD) * scr_rcpf((float)Es);
        if (nb == 0) {
            x0 = xa; x1 = xb;
            dx01 = dx;
            nb = 2;
        } else {
            x2 = xa; x3 = xb;
            dx23 = dx;
            nb = 4;
        }
    }
    return bad ? -1 : nb;
}

// p3p_setup for the screen: false = no candidate at all (the exact route returns false on the same conditions);
// n < 0 marks a degenerate configuration the caller must treat as "maybe"
ESAC_HD bool screen_setup(const V3 P[4], const double mu_px[4], const double mv_px[4], const Cam& cam, ScreenSetup& S) {
#pragma clang fp contract(fast)
    const double inv_fx = scr_rcp(cam.fx), inv_fy = scr_rcp(cam.fy), cx_fx = cam.cx * inv_fx, cy_fy = cam.cy * inv_fy;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        S.mu[i] = inv_fx * mu_px[i] - cx_fx;
        S.mv[i] = inv_fy * mv_px[i] - cy_fy;
        S.mk[i] = scr_rcp(scr_sqrt(S.mu[i] * S.mu[i] + S.mv[i] * S.mv[i] + 1));
        S.mu[i] *= S.mk[i];
        S.mv[i] *= S.mk[i];
    }
    const double* mu = S.mu;
    const double* mv = S.mv;
    const double* mk = S.mk;
    const V3 d12 = P[1] - P[2], d02 = P[0] - P[2], d01 = P[0] - P[1];
    const double s0 = d12.x * d12.x + d12.y * d12.y + d12.z * d12.z;
    const double s1 = d02.x * d02.x + d02.y * d02.y + d02.z * d02.z;
    const double s2 = d01.x * d01.x + d01.y * d01.y + d01.z * d01.z;
    const double p = 2 * (mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2]);
    const double q = 2 * (mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2]);
    const double r = 2 * (mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1]);
    const double inv_d22 = scr_rcp(s2);
    const double a = inv_d22 * s0, b = inv_d22 * s1;
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    S.n = 0;
    if (p2 + q2 + r2 - pqr - 1 == 0) return false;
    const double ab = a * b, a_2 = 2 * a, a_4 = 4 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return false;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return false;
    // magnitudes of the terms each coefficient is summed from (a, b > 0; single precision like every bound here), in units
    // of M = (1 + a + b)^2: A <= 2 M, B and D <= (6 |q| + |p r|) M, C <= 16 M, E <= 4 M (|p|, |q|, |r| <= 2; each bound is
    // within ~3x of the exact sum of magnitudes, which SCREEN_SIG absorbs)
    const float sf = 1 + (float)a + (float)b, M = sf * sf;
    const float fpr = (float)fabs(pr), fq = (float)fabs(q);
    const float mA = 2 * M, mB = (6 * fq + fpr) * M, mC = 16 * M, mD = mB, mE = 4 * M;
    const float mT = ((float)p2 + (float)r2 + fq * fpr) * sf;
    const float cu = (float)SCREEN_CU;
    if (!((float)fabs(temp) > (float)SCREEN_SIG * cu * mT)) {  // b0's sign / zero test is rounding
        ESAC_SCREEN_STAT(4);
        S.n = -1;
        return true;
    }
    // (temp significant to SCREEN_SIG = 1e3 bounds: b0 = b temp^2 is reproducible to 2e-3 / SCREEN_SIG, inside SCREEN_REL)
    double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    float e01 = 0, e23 = 0;
    const int n = quartic_roots_fast(A, B, C, D, E, cu * mA, cu * mB, cu * mC, cu * mD, cu * mE, x0, x1, x2, x3, e01, e23);
    if (n == 0) return false;
    S.n = n;  // -1: degenerate
    S.x[0] = x0; S.x[1] = x1; S.x[2] = x2; S.x[3] = x3;
    S.dx01 = e01; S.dx23 = e23;
    S.sf = sf; S.m136 = 136.0f * M;
    S.dist2 = scr_sqrt(s2); S.a = a; S.b = b; S.p = p; S.q = q; S.r = r;
    S.inv_b0 = scr_rcp(b0);
    return true;
}

// p3p_candidate_lengths for the screen (same polynomial, contracted).  1: valid, depths in X, Y, Z; 0: the candidate the
// fp64 route skips too; -1: a validity test (x > 0, b1 > 0, v > 0) or the depth ratio is within the rounding bound of
// flipping / not reproducible to SCREEN_REL -- the try is "maybe".  dx: bound of the root's difference (quartic_roots_fast).
ESAC_HD int screen_lengths(const ScreenSetup& S, double x, float dx, double& X, double& Y, double& Z) {
#pragma clang fp contract(fast)
    const double a = S.a, b = S.b, p = S.p, q = S.q, r = S.r;
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, ab = a * b, a_2 = 2 * a, a_4 = 4 * a;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    ESAC_SCREEN_HIST(4, dx / (float)fabs(x));
    if (!(dx <= (float)SCREEN_REL * (float)fabs(x))) {  // the root's sign or value is not reproducible (NaN included)
        ESAC_SCREEN_STAT(5);
        return -1;
    }
    if (x <= 0) return 0;
    const double xx = x * x;
    // b1(x) = f1(x) * f2(x): a quadratic times a cubic whose coefficients do not depend on the root
    const double g2 = 1 - a - b, g1 = q * a - q, g0 = 1 - a + b;
    const double k3 = r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1);
    const double k2 = r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2);
    const double k1 = r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
                      pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b);
    const double k0 = 2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
                      p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1));
    const double f1 = g2 * xx + g1 * x + g0;
    const double f2 = (k3 * x + k2) * xx + k1 * x + k0;
    const double b1 = f1 * f2;
    // How far can b1 -- its sign, and the depth ratio y = b1 / b0 -- be from the exact route's?  By the rounding of the two
    // factors (SCREEN_CU of the magnitudes of their terms: |p|, |q|, |r| <= 2 bounds the bracketed sums of f2 by 8 M, 72 M,
    // 136 M, 136 M with M = (1 + a + b)^2 -- loose by up to ~30x, which only matters where f2 has lost ten digits to
    // cancellation) and by what the root's own difference does to it (first order: |b1'(x)| dx).  Bookkeeping: fp32.
    const float xf = (float)x, xxf = xf * xf, sf = 1 + (float)a + (float)b;
    const float f1f = (float)fabs(f1), f2f = (float)fabs(f2);
    const float m1 = sf * (xxf + 1) + 2 * (1 + (float)a) * xf;
    const float m2 = 136 * sf * sf * (xf + 1) * (xxf + 1);
    const float db1x = fabsf((2 * (float)g2 * xf + (float)g1) * (float)f2 + (float)f1 * ((3 * (float)k3 * xf + 2 * (float)k2) * xf + (float)k1)) * dx;
    const float db1 = (f1f * m2 + f2f * m1) * (float)SCREEN_CU + db1x;
    const float fb1 = f1f * f2f;
    ESAC_SCREEN_HIST(5, (f1f * m2 + f2f * m1) * (float)SCREEN_CU / fb1);
    ESAC_SCREEN_HIST(6, db1x / fb1);
    ESAC_SCREEN_HIST(7, f1f / m1);
    ESAC_SCREEN_HIST(8, f2f / m2);
    const double y = S.inv_b0 * b1;
    const double v = xx + y * y - x * y * r;
    // sign of b1 or the value of y = b1 / b0 not reproducible (covers b1 ~ 0 and NaN); v is a positive definite form of
    // (x, y) unless |r| = 2: a sign at rounding distance (only looked at for the candidates the exact route keeps, b1 > 0)
    if (!(fb1 * (float)SCREEN_REL > db1) || (b1 > 0 && !(v > 1e-9 * (xx + y * y)))) {
        ESAC_SCREEN_STAT(6);
        return -1;
    }
    if (b1 <= 0) return 0;
    Z = S.dist2 * scr_rcp(scr_sqrt(v));
    X = x * Z;
    Y = y * Z;
    return 1;
}

// ---- screen over the fp64 roots and depths --------------------------------------------------------------------------------
// fp32 Ferrari is too fragile to screen with (the probe finds ~40 % of random tries within rounding distance of a branch
// threshold), and the depth ratio y must be the fp64 route's own b1 / b0 -- on ill-conditioned samples (base points a few
// cells apart) that quotient is set by cancellation, not by the P3P equations, and it is what the reference accepts or
// rejects.  So roots (p3p_setup) and depths (p3p_candidate_lengths) stay the fp64 route's, same doubles, same validity
// tests.  What the screen replaces is the rest of every candidate: the least-squares triangle alignment with its Newton
// iterations and the fp64 projection of the 4th point (~350 dependent fp64 operations) by orthonormal triads and an fp32
// projection (~90 fp32 operations).  Triads give the alignment exactly when the two triangles are congruent; how far
// they are from that is measured, and a candidate whose camera-frame triangle misses the scene triangle's side lengths
// by more than 1e-3 is reported as "maybe" instead of being judged.
// Returns the smallest 4th-point error over the candidates (+inf: none), or ESAC_SCREEN_MAYBE.
// scene side of a try: orthonormal triad on the three base points and the 4th point's coordinates in it
struct ScreenScene {
    V3f P0, e1, e2, e3;
    float c1, c2, c3;  // 4th point in the triad
    float l1, l2, l3;  // squared side lengths |P1-P0|^2, |P2-P0|^2, |P2-P1|^2
};
// false: coincident or (near-)collinear base points -- the caller reports "maybe"
ESAC_HD bool screen_scene(const float (&Pf)[4][3], ScreenScene& sc) {
    const V3f P0{Pf[0][0], Pf[0][1], Pf[0][2]}, P1{Pf[1][0], Pf[1][1], Pf[1][2]}, P2{Pf[2][0], Pf[2][1], Pf[2][2]},
        P3{Pf[3][0], Pf[3][1], Pf[3][2]};
    const V3f pe1 = P1 - P0, pe2 = P2 - P0, pe3 = P2 - P1;
    sc.P0 = P0;
    sc.l1 = dotf(pe1, pe1); sc.l2 = dotf(pe2, pe2); sc.l3 = dotf(pe3, pe3);
    if (!(sc.l1 > 0) || !(sc.l2 > 0) || !(sc.l3 > 0)) return false;
    sc.e1 = scr_rsqf(sc.l1) * pe1;
    V3f e3 = crossf(sc.e1, pe2);
    const float n3 = dotf(e3, e3);
    if (!(n3 > 1e-8f * sc.l2)) return false;  // (near-)collinear sample
    sc.e3 = scr_rsqf(n3) * e3;
    sc.e2 = crossf(sc.e3, sc.e1);
    const V3f w = P3 - P0;
    sc.c1 = dotf(w, sc.e1); sc.c2 = dotf(w, sc.e2); sc.c3 = dotf(w, sc.e3);
    return true;
}
// one candidate: depths X, Y, Z of the base points along their unit bearings (mu, mv, mk) -> reprojection error of the 4th
// point in pixels, or ESAC_SCREEN_MAYBE.  `congruence`: largest relative mismatch of the squared side lengths at which the
// camera-frame triangle still counts as the scene triangle (beyond it the least-squares alignment of the fp64 route and
// the triads here are different rigid motions).
ESAC_HD float screen_candidate(const ScreenScene& sc, const float (&mu)[3], const float (&mv)[3], const float (&mk)[3], float X, float Y,
                               float Z, float mu3_px, float mv3_px, float f, float cx, float cy, float congruence) {
    if (!(fabsf(X) < 1e18f && fabsf(Y) < 1e18f && fabsf(Z) < 1e18f)) return ESAC_SCREEN_MAYBE;  // NaN / overflow
    const V3f Q0{X * mu[0], X * mv[0], X * mk[0]}, Q1{Y * mu[1], Y * mv[1], Y * mk[1]}, Q2{Z * mu[2], Z * mv[2], Z * mk[2]};
    const V3f qe1 = Q1 - Q0, qe2 = Q2 - Q0, qe3 = Q2 - Q1;
    const float m1 = dotf(qe1, qe1), m2 = dotf(qe2, qe2), m3s = dotf(qe3, qe3);
    if (!(fabsf(m1 - sc.l1) <= congruence * sc.l1) || !(fabsf(m2 - sc.l2) <= congruence * sc.l2) || !(fabsf(m3s - sc.l3) <= congruence * sc.l3))
        return ESAC_SCREEN_MAYBE;
    const float dm = fmaxf(fmaxf(fabsf(m1 - sc.l1), fabsf(m2 - sc.l2)), fabsf(m3s - sc.l3));  // largest mismatch of a squared side, see below
    const V3f f1 = scr_rsqf(m1) * qe1;
    V3f f3 = crossf(f1, qe2);
    const float m3 = dotf(f3, f3);
    if (!(m3 > 1e-8f * m2)) return ESAC_SCREEN_MAYBE;
    f3 = scr_rsqf(m3) * f3;
    const V3f f2 = crossf(f3, f1);
    const float Xc = Q0.x + sc.c1 * f1.x + sc.c2 * f2.x + sc.c3 * f3.x;
    const float Yc = Q0.y + sc.c1 * f1.y + sc.c2 * f2.y + sc.c3 * f3.y;
    const float Zc = Q0.z + sc.c1 * f1.z + sc.c2 * f2.z + sc.c3 * f3.z;
    if (!(fabsf(Zc) > 1e-3f * (fabsf(Xc) + fabsf(Yc) + 1e-6f))) return ESAC_SCREEN_MAYBE;  // 4th point next to the camera plane
    // The camera-frame triangle is never exactly the scene triangle (the depths carry the rounding of the roots), and the
    // fp64 route's least-squares alignment and the triads above settle that mismatch -- dl metres of side length --
    // differently: their 4th points differ by ~lever * (side mismatch) =: dl metres, which is f * dl / |Zc| pixels.  Harmless at metres of
    // depth; with the 4th point millimetres from the camera centre it is the whole margin (found by the 3.6e10-try host
    // campaign of round 3: a sample whose 4th cell repeats a base point, camera 3.5 mm from it, 14.0 px here against
    // 9.98 px in the fp64 route).  Half a pixel of it is allowed.
    // (dm of a squared side length is dm / (2 * side) metres of side; the lever is 1 + |P3 - P0| / side; both with the
    // shortest side, formed here from what is live anyway: no register held across the candidates)
    const float is = scr_rsqf(fminf(sc.l1, fminf(sc.l2, sc.l3)));
    const float dl = dm * 0.5f * is * (1.0f + scr_sqrtf(sc.c1 * sc.c1 + sc.c2 * sc.c2 + sc.c3 * sc.c3) * is);
    if (!(2.0f * f * dl <= 0.5f * fabsf(Zc))) {
        ESAC_SCREEN_STAT(9);
        return ESAC_SCREEN_MAYBE;
    }
    const float iz = scr_rcpf(Zc);
    const float du = cx + f * Xc * iz - mu3_px, dv = cy + f * Yc * iz - mv3_px;
    const float epx = scr_sqrtf(du * du + dv * dv);
    if (!(epx == epx)) return ESAC_SCREEN_MAYBE;
    return epx;
}

template <typename Setup, typename Lengths>
ESAC_HD float p3p_screen_roots_t(const Setup& S, Lengths lengths, const float (&Pf)[4][3], float mu3_px, float mv3_px, float f, float cx,
                               float cy) {
    if (S.n < 0) return ESAC_SCREEN_MAYBE;  // degenerate quartic
    const float mu[3] = {(float)S.mu[0], (float)S.mu[1], (float)S.mu[2]}, mv[3] = {(float)S.mv[0], (float)S.mv[1], (float)S.mv[2]},
                mk[3] = {(float)S.mk[0], (float)S.mk[1], (float)S.mk[2]};
    ScreenScene sc;  // shared by all candidates
    if (!screen_scene(Pf, sc)) return ESAC_SCREEN_MAYBE;
    float best = INFINITY;
    // the roots as VALUES before the loop: a select chain over struct members inside it is turned into an indexed load,
    // which keeps the whole setup struct in memory (scratch / LDS) instead of registers
    const double xs0 = S.x[0], xs1 = S.x[1], xs2 = S.x[2], xs3 = S.x[3];
    // not unrolled: four inlined copies of the b1 polynomial push the sampling kernel out of the instruction cache
#pragma nounroll
    for (int i = 0; i < S.n; i++) {
        const double x = i == 0 ? xs0 : i == 1 ? xs1 : i == 2 ? xs2 : xs3;
        double Xd, Yd, Zd;
        const int ok = lengths(S, x, i, Xd, Yd, Zd);
        if (ok < 0) return ESAC_SCREEN_MAYBE;  // validity or depths not reproducible between the two routes
        if (ok == 0) continue;                 // the candidates the fp64 route skips
        const float epx = screen_candidate(sc, mu, mv, mk, (float)Xd, (float)Yd, (float)Zd, mu3_px, mv3_px, f, cx, cy, ESAC_SCREEN_CONGRUENCE);
        if (epx == ESAC_SCREEN_MAYBE) return ESAC_SCREEN_MAYBE;
        best = fminf(best, epx);
    }
    return best;
}

// the screen on the exact route's own setup (same doubles), and on its private fast copy
ESAC_HD float p3p_screen_roots(const P3PSetup& S, const float (&Pf)[4][3], float mu3_px, float mv3_px, float f, float cx, float cy) {
    return p3p_screen_roots_t(S, [](const P3PSetup& s, double x, int, double& X, double& Y, double& Z) { return p3p_candidate_lengths(s, x, X, Y, Z) ? 1 : 0; },
                              Pf, mu3_px, mv3_px, f, cx, cy);
}
ESAC_HD float p3p_screen_roots(const ScreenSetup& S, const float (&Pf)[4][3], float mu3_px, float mv3_px, float f, float cx, float cy) {
    const float d01 = S.dx01, d23 = S.dx23;  // values, see p3p_screen_roots_t
    return p3p_screen_roots_t(S, [=](const ScreenSetup& s, double x, int i, double& X, double& Y, double& Z) {
        return screen_lengths(s, x, i < 2 ? d01 : d23, X, Y, Z); },
                              Pf, mu3_px, mv3_px, f, cx, cy);
}

}  // namespace esac
