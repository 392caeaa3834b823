// p3p_coarse.hpp -- COARSE fp32 screen for one sampling try (esac_util.h:152-223), in front of the fp64 screen of
// p3p_screen.hpp: can the 4th point of this try come anywhere near its pixel under ANY pose through the three base points?
//
// The fp64 screen follows the CPU solver's own quartic (its roots and depths are the doubles the decision uses), which is
// what makes it tight (margin 3 px) -- and what keeps it at ~1400 dependent fp64 instructions and two wavefronts per
// SIMD.  A wrong-expert hypothesis spends ~10^3 tries on samples whose 4th point misses by hundreds of pixels; for those
// a geometric answer is enough.  This screen solves the same three-point problem by a different, fp32-friendly route:
//   * the three depth constraints  l_i^2 + l_j^2 - 2 c_ij l_i l_j = a_ij  as two homogeneous conics in (l1 : l2 : l3);
//   * one real root of the 3x3 pencil's cubic gives a degenerate conic = a pair of lines (split through its adjugate);
//   * each line reduces the problem to a quadratic in l3 / l2: up to four depth triples;
//   * every triple is polished by Gauss-Newton steps on the constraints themselves, so the accuracy of a candidate
//     depends on the conditioning of the solution, not on the cancellation inside the elimination;
//   * the a-posteriori congruence test of screen_candidate() (camera-frame triangle = scene triangle to 1e-3) is what
//     certifies a candidate; an uncertified one, a clamped discriminant, a vanishing pivot -> "maybe".
// (The pencil-of-conics elimination is the one published as "Lambda Twist", Persson & Nordberg, ECCV 2018; this is an
// independent fp32 implementation with a different line-splitting step.)
// ONE-SIDED use only: a try is dropped when this returns an error above tau + COARSE_MARGIN; everything else goes on to
// the fp64 screen and the fp64 decision unchanged.  Calibrated against the fp64 decision by
// tests/native/p3p_screen_probe.cpp (mode 4) / tests/test_device_math_host.py.
#pragma once
#include "p3p_screen.hpp"

namespace esac {

#ifndef ESAC_COARSE_CONGRUENCE
#define ESAC_COARSE_CONGRUENCE 1e-3f
#endif
#ifndef ESAC_COARSE_ROOT_K
#define ESAC_COARSE_ROOT_K 1e-5f     // |adj| * |p'(root)| against (terms of p at the root) * |A|
#endif
#ifndef ESAC_COARSE_DISC_TOL
#define ESAC_COARSE_DISC_TOL 1e-3f   // |discriminant| relative to its terms below which a point pair counts as double
#endif
#ifndef ESAC_COARSE_COEF_EPS
#define ESAC_COARSE_COEF_EPS 2e-6f   // rounding of the restricted conic's coefficients, per unit of cancellation
#endif
#ifndef ESAC_COARSE_GAO_TOL
#define ESAC_COARSE_GAO_TOL 2e-3f    // |temp| relative to its terms below which the fp64 route's depth quotient is rounding
#endif
#ifndef ESAC_COARSE_LINE_TOL
#define ESAC_COARSE_LINE_TOL 1e-4f   // |adj| relative to |A|^2 below which the two lines count as coincident
#endif
#define COARSE_BAIL(code) do { if (reason) *reason = (code); return ESAC_SCREEN_MAYBE; } while (0)

ESAC_HD float coarse_rsqrt(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(v);
#else
    return 1.0f / sqrtf(v);
#endif
}
ESAC_HD float coarse_rcp(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(v);
#else
    return 1.0f / v;
#endif
}

// one real root of g^3 + b g^2 + c g + d in closed form (to be polished by the caller)
ESAC_HD float coarse_cubic_root(float b, float c, float d) {
    const float b3 = b * (1.0f / 3.0f);
    const float Q = (3.0f * c - b * b) * (1.0f / 9.0f), R = (9.0f * b * c - 27.0f * d - 2.0f * b * b * b) * (1.0f / 54.0f);
    const float Q3 = Q * Q * Q, D = Q3 + R * R;
    if (D > 0.0f) {
        const float s = cbrtf(fabsf(R) + sqrtf(D));
        const float S = R < 0.0f ? -s : s;
        return S - (S != 0.0f ? Q * coarse_rcp(S) : 0.0f) - b3;
    }
    // three real roots: the outermost one on the side of R (the best separated from the other two)
    const float sq = sqrtf(fmaxf(-Q, 0.0f));
    float arg = sq > 0.0f ? fabsf(R) * coarse_rcp(sq * sq * sq) : 0.0f;
    arg = fminf(arg, 1.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    const float cs = __cosf(acosf(arg) * (1.0f / 3.0f));
#else
    const float cs = cosf(acosf(arg) * (1.0f / 3.0f));
#endif
    return (R < 0.0f ? -2.0f : 2.0f) * sq * cs - b3;
}

// smallest 4th-point reprojection error (pixels) over the pose candidates of this try; +inf: no candidate;
// ESAC_SCREEN_MAYBE: not certain.  `reason` (probe only): what made it uncertain.
ESAC_HD float p3p_coarse_err(const float (&Pf)[4][3], const float (&mu_px)[4], const float (&mv_px)[4], float f, float cx, float cy,
                             int* reason = nullptr) {
    const float inv_f = coarse_rcp(f);
    float mu[3], mv[3], mk[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float u = (mu_px[i] - cx) * inv_f, v = (mv_px[i] - cy) * inv_f;
        const float k = coarse_rsqrt(u * u + v * v + 1.0f);
        mu[i] = u * k; mv[i] = v * k; mk[i] = k;
    }
    ScreenScene sc;
    if (!screen_scene(Pf, sc)) COARSE_BAIL(10);
    // squared sides, scaled to O(1)
    const float amax = fmaxf(sc.l1, fmaxf(sc.l2, sc.l3));
    const float ia = coarse_rcp(amax);
    const float a12 = sc.l1 * ia, a13 = sc.l2 * ia, a23 = sc.l3 * ia;
    if (!(fminf(a12, fminf(a13, a23)) > 1e-6f)) COARSE_BAIL(10);
    const V3f y1{mu[0], mv[0], mk[0]}, y2{mu[1], mv[1], mk[1]}, y3{mu[2], mv[2], mk[2]};
    const float c12 = dotf(y1, y2), c13 = dotf(y1, y3), c23 = dotf(y2, y3);
    const float b12 = -2.0f * c12, b13 = -2.0f * c13, b23 = -2.0f * c23;
    // sin^2 of the angles between the bearings from the cross products (1 - c^2 cancels for neighbouring cells)
    const V3f x12 = crossf(y1, y2), x13 = crossf(y1, y3), x23 = crossf(y2, y3);
    const float s12 = dotf(x12, x12), s13 = dotf(x13, x13), s23 = dotf(x23, x23);
    const float blob = c12 * c23 * c13 - 1.0f;
    const float p3 = a13 * (a23 * s13 - a13 * s23);
    const float p2 = 2.0f * blob * a23 * a13 + a13 * (2.0f * a12 + a13) * s23 + a23 * (a23 - a12) * s13;
    const float p1 = a23 * (a13 - a23) * s12 - a12 * a12 * s23 - 2.0f * a12 * (blob * a23 + a13 * s23);
    const float p0 = a12 * (a12 * s23 - a23 * s12);
    // The solver this screen stands in front of (pose_math.hpp p3p_setup / p3p_candidate_lengths, after the CPU library's
    // Gao-style elimination) takes the second depth ratio from a quotient b1 / (b temp^2); where `temp` cancels, that
    // quotient is set by rounding, the "solution" misses the constraints by percents -- and may still be ACCEPTED, because
    // acceptance only asks for four reprojections within tau.  Such a try has no counterpart among the exact solutions
    // this screen enumerates: it must go through.
    {
        const float ia12 = coarse_rcp(a12);
        const float ga = a23 * ia12, gb = a13 * ia12, gp = 2.0f * c23, gq = 2.0f * c13, gr = 2.0f * c12;
        const float t1 = gp * gp * (ga - 1.0f + gb), t2 = gr * gr * (ga - 1.0f - gb), t3 = gp * gq * gr, t4 = ga * t3;
        if (!(fabsf(t1 + t2 + t3 - t4) > ESAC_COARSE_GAO_TOL * (fabsf(t1) + fabsf(t2) + fabsf(t3) + fabsf(t4)))) COARSE_BAIL(11);
    }
    // magnitudes of the terms each coefficient was summed from (their rounding is what limits the root)
    const float ablob = fabsf(blob);
    const float T3 = a13 * (a23 * s13 + a13 * s23);
    const float T2 = 2.0f * ablob * a23 * a13 + a13 * (2.0f * a12 + a13) * s23 + a23 * (a23 + a12) * s13;
    const float T1 = a23 * (a13 + a23) * s12 + a12 * a12 * s23 + 2.0f * a12 * (ablob * a23 + a13 * s23);
    const float T0 = a12 * (a12 * s23 + a23 * s12);
    // a real root of det(gs D1 - gc D2) = p3 gc^3 + p2 gc^2 gs + p1 gc gs^2 + p0 gs^3, as v = gc / gs or gs / gc with |v| <= 1
    // (an isosceles-like sample has p3 ~ 0 and its only real root at gc / gs -> infinity)
    bool inv = fabsf(p3) < fabsf(p0);
    float k3 = inv ? p0 : p3, k2 = inv ? p1 : p2, k1 = inv ? p2 : p1, k0 = inv ? p3 : p0;
    if (!(fabsf(k3) > 0.0f)) COARSE_BAIL(12);
    const float ik3 = coarse_rcp(k3);
    float v = coarse_cubic_root(k2 * ik3, k1 * ik3, k0 * ik3);
    if (!(v == v)) COARSE_BAIL(13);
    if (fabsf(v) > 1.0f) {
        v = coarse_rcp(v);
        inv = !inv;
        float tmp = k3; k3 = k0; k0 = tmp;
        tmp = k2; k2 = k1; k1 = tmp;
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {  // Newton on the cubic itself (unnormalised: k3 may be tiny in this orientation)
        const float fv = ((k3 * v + k2) * v + k1) * v + k0, df = (3.0f * k3 * v + 2.0f * k2) * v + k1;
        if (df != 0.0f) v -= fv * coarse_rcp(df);
    }
    // uncertainty of the root: rounding of the cubic at the root over its slope there (large next to a multiple root)
    const float av = fabsf(v);
    const float perr = ((((inv ? T0 : T3) * av + (inv ? T1 : T2)) * av + (inv ? T2 : T1)) * av + (inv ? T3 : T0));
    const float slope = fabsf((3.0f * k3 * v + 2.0f * k2) * v + k1);
    if (!(slope > 0.0f)) COARSE_BAIL(13);
    const float gc = inv ? 1.0f : v, gs = inv ? v : 1.0f;
    // degenerate member gs D1 - gc D2 of the pencil (symmetric)
    const float A00 = a23 * (gs - gc), A01 = 0.5f * a23 * b12 * gs, A02 = -0.5f * gc * a23 * b13;
    const float A11 = gs * (a23 - a12) + gc * a13, A12 = 0.5f * b23 * (gc * a13 - gs * a12), A22 = gc * (a13 - a23) - gs * a12;
    // adjugate = (product of the two non-zero eigenvalues) * n n^T, n = the point the two lines share
    const float B00 = A11 * A22 - A12 * A12, B11 = A00 * A22 - A02 * A02, B22 = A00 * A11 - A01 * A01;
    const float B01 = A02 * A12 - A01 * A22, B02 = A01 * A12 - A02 * A11, B12 = A01 * A02 - A00 * A12;
    float bd, n0, n1, n2;
    if (fabsf(B00) >= fabsf(B11) && fabsf(B00) >= fabsf(B22)) {
        bd = B00; n0 = B00; n1 = B01; n2 = B02;
    } else if (fabsf(B11) >= fabsf(B22)) {
        bd = B11; n0 = B01; n1 = B11; n2 = B12;
    } else {
        bd = B22; n0 = B02; n1 = B12; n2 = B22;
    }
    const float smax = fmaxf(fmaxf(fabsf(A00), fabsf(A11)), fmaxf(fmaxf(fabsf(A22), fabsf(A01)), fmaxf(fabsf(A02), fabsf(A12))));
    const float btol = ESAC_COARSE_LINE_TOL * smax * smax;
    // cancellation in gs D1 - gc D2 (entries of D1, D2 are O(max side^2) = O(1) here)
    const float kappa = (fabsf(gs) + fabsf(gc)) * coarse_rcp(smax);
    // the two lines (nearly) coincide -- every solution is a near-double one -- or form a conjugate pair (no real point
    // besides their intersection).  A root off by dv moves the adjugate by ~ dv * |A|: below a multiple of that neither the
    // pair nor the sign is determined at fp32
    if (!(fabsf(bd) * slope > ESAC_COARSE_ROOT_K * perr * smax) || !(fabsf(bd) > btol)) COARSE_BAIL(17);
    if (bd > 0.0f) return INFINITY;
    {
        const float ib = coarse_rsqrt(-bd);
        n0 *= ib; n1 *= ib; n2 *= ib;
    }
    // A + [n]x has rank one: (a multiple of) line_a line_b^T
    const float C00 = A00, C01 = A01 - n2, C02 = A02 + n1;
    const float C10 = A01 + n2, C11 = A11, C12 = A12 - n0;
    const float C20 = A02 - n1, C21 = A12 + n0, C22 = A22;
    const float r0n = C00 * C00 + C01 * C01 + C02 * C02, r1n = C10 * C10 + C11 * C11 + C12 * C12, r2n = C20 * C20 + C21 * C21 + C22 * C22;
    const float k0n = C00 * C00 + C10 * C10 + C20 * C20, k1n = C01 * C01 + C11 * C11 + C21 * C21, k2n = C02 * C02 + C12 * C12 + C22 * C22;
    float La[3], Lb[3];
    if (r0n >= r1n && r0n >= r2n) { La[0] = C00; La[1] = C01; La[2] = C02; }
    else if (r1n >= r2n) { La[0] = C10; La[1] = C11; La[2] = C12; }
    else { La[0] = C20; La[1] = C21; La[2] = C22; }
    if (k0n >= k1n && k0n >= k2n) { Lb[0] = C00; Lb[1] = C10; Lb[2] = C20; }
    else if (k1n >= k2n) { Lb[0] = C01; Lb[1] = C11; Lb[2] = C21; }
    else { Lb[0] = C02; Lb[1] = C12; Lb[2] = C22; }
    const float depth_scale = sqrtf(amax);
    float best = INFINITY;
#pragma unroll
    for (int ln = 0; ln < 2; ln++) {
        const float g0 = ln ? Lb[0] : La[0], g1 = ln ? Lb[1] : La[1], g2 = ln ? Lb[2] : La[2];
        // q . (l1, l2, l3) = 0: eliminate the depth with the largest coefficient.  The constraints are symmetric under a
        // relabelling of the points, so the elimination is written once, for "the first" depth, on permuted inputs:
        // sel 0: (1, 2, 3)   sel 1: (2, 1, 3)   sel 2: (3, 1, 2)
        const int sel = (fabsf(g0) >= fabsf(g1) && fabsf(g0) >= fabsf(g2)) ? 0 : (fabsf(g1) >= fabsf(g2) ? 1 : 2);
        const float q0 = sel == 0 ? g0 : (sel == 1 ? g1 : g2), q1 = sel == 0 ? g1 : g0, q2 = sel == 2 ? g1 : g2;
        const float e12 = sel == 2 ? a13 : a12, e13 = sel == 0 ? a13 : a23, e23 = sel == 0 ? a23 : (sel == 1 ? a13 : a12);
        const float h12 = sel == 2 ? b13 : b12, h13 = sel == 0 ? b13 : b23, h23 = sel == 0 ? b23 : (sel == 1 ? b13 : b12);
        if (!(fabsf(q0) > 0.0f)) COARSE_BAIL(14);
        const float iq = -coarse_rcp(q0);
        const float w0 = q1 * iq, w1 = q2 * iq;  // first = w0 * second + w1 * third, |w| <= 1
        // the three constraint forms restricted to the line, in t = third / second (second = 1):
        const float m12a = w1 * w1, m12b = (2.0f * w0 + h12) * w1, m12c = (w0 + h12) * w0 + 1.0f;
        const float m13a = (w1 + h13) * w1 + 1.0f, m13b = (2.0f * w1 + h13) * w0, m13c = w0 * w0;
        // every member of the pencil vanishes at the solutions and, on the line, they are all proportional: take the one
        // with the largest coefficients (a member close to the degenerate one is ~0 on the line)
        const float d1a = e23 * m12a - e12, d1b = e23 * m12b - e12 * h23, d1c = e23 * m12c - e12;
        const float d2a = e23 * m13a - e13, d2b = e23 * m13b - e13 * h23, d2c = e23 * m13c - e13;
        const float d3a = e13 * m12a - e12 * m13a, d3b = e13 * m12b - e12 * m13b, d3c = e13 * m12c - e12 * m13c;
        const float n1 = fabsf(d1a) + fabsf(d1b) + fabsf(d1c), n2 = fabsf(d2a) + fabsf(d2b) + fabsf(d2c), n3 = fabsf(d3a) + fabsf(d3b) + fabsf(d3c);
        const int pick = (n1 >= n2 && n1 >= n3) ? 1 : (n2 >= n3 ? 2 : 3);
        const float qa = pick == 1 ? d1a : (pick == 2 ? d2a : d3a), qb = pick == 1 ? d1b : (pick == 2 ? d2b : d3b),
                    qc = pick == 1 ? d1c : (pick == 2 ? d2c : d3c);
        // a root at t -> infinity or t -> 0 is a solution with one depth ~ 0 (the camera next to a scene point): legal for the
        // solver this screen stands in front of, and its sign is rounding here
        if (!(fabsf(qa) > 1e-3f * (fabsf(qb) + fabsf(qc))) || !(fabsf(qc) > 1e-3f * (fabsf(qa) + fabsf(qb)))) COARSE_BAIL(15);
        float disc = qb * qb - 4.0f * qa * qc;
        // how well the sign of the discriminant is known: relative to its own terms, and to the rounding of the coefficients
        // (differences of O(1) products, through line coefficients that lost log10(kappa) digits when A was formed)
        const float dscale = qb * qb + 4.0f * fabsf(qa * qc);
        const float dband = fmaxf(ESAC_COARSE_DISC_TOL * dscale, ESAC_COARSE_COEF_EPS * kappa * (fabsf(qa) + fabsf(qb) + fabsf(qc)));
        bool dclamped = false;
        if (disc < 0.0f) {
            if (disc < -dband) continue;  // no real point on this line
            disc = 0.0f;
            dclamped = true;
        } else if (disc <= dband) {
            dclamped = true;
        }
        // a (near-)double point: the pose through it is ill-conditioned (camera next to the danger cylinder), fp32 depths
        // cannot place the 4th point -- not ours to judge
        if (dclamped) COARSE_BAIL(18);
        const float sq = sqrtf(disc);
        const float qq = -0.5f * (qb + (qb < 0.0f ? -sq : sq));
        float tr[2];
        tr[0] = qa != 0.0f ? qq * coarse_rcp(qa) : -1.0f;
        tr[1] = qq != 0.0f ? qc * coarse_rcp(qq) : -1.0f;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float t = tr[k];
            if (!(t == t)) COARSE_BAIL(15);
            if (!(t > 0.0f)) continue;
            const float den = (t + h23) * t + 1.0f;  // > 0: |h23| < 2
            const float m2 = sqrtf(e23 * coarse_rcp(den));
            const float m3 = t * m2;
            const float m1 = w0 * m2 + w1 * m3;
            float l1 = sel == 0 ? m1 : m2, l2 = sel == 0 ? m2 : (sel == 1 ? m1 : m3), l3 = sel == 2 ? m1 : m3;
            // Gauss-Newton on the three constraints
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const float r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
                const float r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
                const float r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
                const float ja = 2.0f * l1 + b12 * l2, jb = 2.0f * l2 + b12 * l1;
                const float jc = 2.0f * l1 + b13 * l3, jd = 2.0f * l3 + b13 * l1;
                const float je = 2.0f * l2 + b23 * l3, jf = 2.0f * l3 + b23 * l2;
                const float det = -ja * jd * je - jb * jc * jf;
                if (!(fabsf(det) > 1e-12f)) break;
                const float id = coarse_rcp(det);
                const float s1 = id * (-jd * je * r1 - jb * jf * r2 + jb * jd * r3);
                const float s2 = id * (-jc * jf * r1 + ja * jf * r2 - ja * jd * r3);
                const float s3 = id * (jc * je * r1 - ja * je * r2 - jb * jc * r3);
                // at a double solution the Jacobian is singular and the step is noise: keep the point (the congruence test judges it)
                if (!(fabsf(s1) + fabsf(s2) + fabsf(s3) < 0.05f * (fabsf(l1) + fabsf(l2) + fabsf(l3)))) break;
                l1 -= s1; l2 -= s2; l3 -= s3;
            }
            if (!(l1 > 0.0f && l2 > 0.0f && l3 > 0.0f)) {
                if (!(l1 == l1 && l2 == l2 && l3 == l3)) COARSE_BAIL(16);
                // a depth within rounding of zero: the sign is not ours to decide
                if (fmaxf(fmaxf(-l1, -l2), -l3) < 1e-3f) COARSE_BAIL(17);
                continue;
            }
            const float epx = screen_candidate(sc, mu, mv, mk, l1 * depth_scale, l2 * depth_scale, l3 * depth_scale, mu_px[3], mv_px[3], f, cx, cy,
                                               ESAC_COARSE_CONGRUENCE);
            if (epx == ESAC_SCREEN_MAYBE) COARSE_BAIL(19);
            best = fminf(best, epx);
        }
    }
    return best;
}

// The decision this screen is used for.  An ACCEPTED try is not always an exact solution of the three-point problem: the
// solver behind the decision rounds (Ferrari roots, a quotient of cancelling polynomials), and acceptance only asks for
// four reprojections within tau -- so the accepted pose may sit where the base points reproject up to tau off, and the
// exact solutions enumerated here then see the 4th point off by more than tau.  To first order the 4th point follows the
// base points with its barycentric weights w.r.t. the base triangle in the image: shift <= tau * sum |w_i| =: tau * lever.
// Calibration (tests/native/p3p_screen_probe.cpp, 4e8 tries on ten maps): (err - tau) <= 0.37 * tau * lever over all
// accepted tries.  A try is dropped only above tau + max(COARSE_MARGIN, tau * lever): >= 2.7 x that envelope everywhere.
#ifndef ESAC_COARSE_MARGIN
#define ESAC_COARSE_MARGIN 40.0f
#endif
ESAC_HD float coarse_lever(const float (&mu)[4], const float (&mv)[4]) {
    const float ux = mu[1] - mu[0], uy = mv[1] - mv[0], vx = mu[2] - mu[0], vy = mv[2] - mv[0], px = mu[3] - mu[0], py = mv[3] - mv[0];
    const float d = ux * vy - vx * uy;
    if (!(fabsf(d) > 0.0f)) return INFINITY;  // collinear base pixels
    const float id = coarse_rcp(d);
    const float w1 = (px * vy - vx * py) * id, w2 = (ux * py - px * uy) * id;
    return fabsf(1.0f - w1 - w2) + fabsf(w1) + fabsf(w2);
}
// true: the try goes on to the fp64 screen.  *strong: the coarse screen itself sees the 4th point within tau.
ESAC_HD bool p3p_coarse_maybe(const float (&Pf)[4][3], const float (&mu_px)[4], const float (&mv_px)[4], float f, float cx, float cy, float tau,
                              bool* strong = nullptr) {
    const float err = p3p_coarse_err(Pf, mu_px, mv_px, f, cx, cy);
    if (strong) *strong = err >= 0.0f && err <= tau;
    return !(err > tau + fmaxf(ESAC_COARSE_MARGIN, tau * coarse_lever(mu_px, mv_px)));  // "not certain" (-1) and NaN included
}

}  // namespace esac
