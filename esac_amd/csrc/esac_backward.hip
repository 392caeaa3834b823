// esac_backward.hip -- the training path of the extension (esac_backward, code/esac/esac.cpp:213-520) as HIP
// kernels for gfx950.  Forward kernels re-used: k_sample, k_rescore(all) (exact scores of every hypothesis)
// and k_refine in its SLOTS mode (one workgroup per hypothesis that takes part in the expectation).
//
//   K5 k_bwd_select      softMax, entropy, ordered list of hypotheses with p >= PROB_THRESH   esac.cpp:319-331
//   K6 k_bwd_loss        loss per hypothesis, expected loss, dLoss, d expectation / d score    esac.cpp:354-362,
//                                                                                              esac_derivative.h:405-420
//   K7 k_bwd_path1       refined pose -> scene coordinates through the last re-fit             esac.cpp:375-463
//   K8 k_bwd_path2       score -> scene coordinates (direct + via the 4 sampled points)        esac_derivative.h:205-330
//   K9 k_bwd_accumulate  outGradients += p_h * pathI_h + pathII_h, hypotheses in order          esac.cpp:491-508
//
// The reference allocates per-hypothesis (1 x 3P) and (P x 3) double matrices on the host heap for ALL N
// hypotheses and adds them into the float tensor one hypothesis after the other.  Here only the <= 1000
// hypotheses that can pass the probability threshold own a slab, one workgroup fills each slab, and K9 walks the
// slabs in hypothesis order per output element, so the float `+=` chain rounds exactly like the reference's.
// Everything is fp64 VALU work on small per-hypothesis problems plus one streaming pass over the H x W cells per
// slot: no contraction large enough for MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bwd_math.hpp"
#include "device_common.hpp"
#include "esac_kernels.hpp"
#include "pose_math.hpp"

namespace esac {

constexpr int BWD_B = 512;  // threads per slot in the gradient kernels: 8 wavefronts, 256-VGPR budget each

// ================================================================= K5: softmax + ordered selection
template <int B>
__global__ __launch_bounds__(B) void k_bwd_select(KArgs a) {
    __shared__ double s_part[2 * (B / 64)];
    __shared__ double s_tot[2];
    __shared__ double s_max[B / 64];
    __shared__ int s_wcount[B / 64];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double m = -INFINITY;
    for (int i = threadIdx.x; i < a.N; i += B) {
        const double s = a.scores[i];
        m = s > m ? s : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(m, o);
        m = om > m ? om : m;
    }
    if (lane == 0) s_max[wave] = m;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    m = s_max[0];
#pragma unroll
    for (int w = 1; w < B / 64; w++) m = s_max[w] > m ? s_max[w] : m;
    double acc[2] = {0, 0};
    for (int i = threadIdx.x; i < a.N; i += B) acc[0] += exp(a.scores[i] - m);
    block_sum<2, B>(acc, s_part, s_tot);
    const double sum = acc[0];
    // probabilities, entropy, the pose every unselected hypothesis keeps, and the ordered selection:
    // rounds of B consecutive hypotheses, ballot prefix inside a wavefront, wavefront offsets through LDS
    double ent = 0;
    for (int base = 0; base < a.N; base += B) {
        const int i = base + (int)threadIdx.x;
        bool pick = false;
        if (i < a.N) {
            const double p = exp(a.scores[i] - m) / sum;
            a.bwd.probs[i] = p;
            if (p > 0) ent -= p * log2(p);
#pragma unroll
            for (int k = 0; k < 6; k++) a.bwd.ref_hyps[(size_t)i * 6 + k] = a.hyps[(size_t)i * 6 + k];
            pick = !(p < kProbThresh);
        }
        const unsigned long long bal = __ballot(pick);
        if (lane == 0) s_wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wcount[w];
        if (pick) {
            const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (slot < a.bwd.cap) a.bwd.sel[slot] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < B / 64; w++) tot += s_wcount[w];
            s_base += tot;
        }
        __syncthreads();
    }
    acc[0] = ent;
    acc[1] = 0;
    block_sum<2, B>(acc, s_part, s_tot);
    if (threadIdx.x == 0) {
        a.bwd.n_sel[0] = s_base < a.bwd.cap ? s_base : a.bwd.cap;
        a.bwd.n_sel[1] = s_base;  // unclamped: > cap tells the host to grow the slot workspace and run the call again
        a.stats[0] = m;
        a.stats[1] = sum;
        a.stats[2] = acc[0];
    }
}

// ================================================================= K6: losses, expectation, dLoss
template <int B>
__global__ __launch_bounds__(B) void k_bwd_loss(KArgs a) {
    __shared__ double s_part[B / 64];
    __shared__ double s_tot[1];
    const double wR = a.bwd.w_rot, wT = a.bwd.w_trans, cut = a.bwd.cut;
    double acc[1] = {0};
    for (int h = threadIdx.x; h < a.N; h += B) {
        const double* hp = a.bwd.ref_hyps + (size_t)h * 6;
        const double pose[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
        const double l = pose_loss(pose, a.bwd.gt, wR, wT, cut);
        a.bwd.losses[h] = l;
        acc[0] += a.bwd.probs[h] * l;
    }
    block_sum<1, B>(acc, s_part, s_tot);
    const double expected = acc[0];
    // d E[loss] / d score_i = p_i l_i - p_i sum_j p_j l_j  (softmax derivative, esac_derivative.h:405-420)
    for (int h = threadIdx.x; h < a.N; h += B) {
        const double p = a.bwd.probs[h];
        a.bwd.sgrad[h] = p < kProbThresh ? 0.0 : p * a.bwd.losses[h] - p * expected;
    }
    const int n_sel = a.bwd.n_sel[0];
    for (int slot = threadIdx.x; slot < n_sel; slot += B) {
        const double* hp = a.bwd.ref_hyps + (size_t)a.bwd.sel[slot] * 6;
        const double pose[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
        double j[6];
        pose_dloss(pose, a.bwd.gt_pose, wR, wT, cut, j);
#pragma unroll
        for (int k = 0; k < 6; k++) a.bwd.dloss[(size_t)slot * 6 + k] = j[k];
    }
    if (threadIdx.x == 0) {
        a.bwd.out[0] = expected;
        a.bwd.out[1] = (double)a.bwd.n_sel[1];
        a.bwd.out[2] = a.stats[2];
        a.bwd.out[3] = (a.status[0] == (unsigned long long)a.sample_epoch) ? 1.0 : 0.0;  // out-of-range hypAssignment (k_sample)
    }
}

// ================================================================= K7: path I
// One workgroup per slot.  jacobeanR = -(J^T J)^-1 J^T over the inliers of the last accepted refinement step
// (rows = d residual norm / d pose at the refined pose), clamped to zero as a whole when any entry exceeds 10;
// the slab entry of inlier q is dLoss (1x6) * jacobeanR[:, q] * dProject/dObj (1x3).
// The rare branch of path I: J^T J is rank deficient to rounding (inv_spd6 returned false) -- the reference's own route,
// the SVD pseudo-inverse (bwd_math.hpp:pinv_sym6_jacobi: the same operations in the same order).  Here it must cost the
// common path nothing: ONE lane runs rolled loops over matrices in LDS (a few hundred bytes of code, no registers beyond the
// loop's own -- unrolled into registers it took the kernel to 480 of them and one wavefront per SIMD), the others wait.  Every
// lane reaches this together (U21 is the same in all of them), so the barriers are uniform.  lds: >= 108 doubles.
__device__ __forceinline__ void pinv_sym6_lds(const double (&U21)[21], double (&Ainv)[36], double* lds) {
    double* A = lds;        // [6][6]
    double* V = lds + 36;   // [6][6]
    double* out = lds + 72; // [6][6]
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) {
                A[i * 6 + j] = U21[k];
                A[j * 6 + i] = U21[k];
                k++;
            }
        for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 60; sweep++) {
            double off = 0;
            for (int i = 0; i < 6; i++)
                for (int j = i + 1; j < 6; j++) off += A[i * 6 + j] * A[i * 6 + j];
            if (off == 0) break;
            for (int p = 0; p < 6; p++)
                for (int q = p + 1; q < 6; q++) {
                    const double apq = A[p * 6 + q];
                    const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                    if (!(fabs(theta) <= 1.7976931348623157e308)) t = 0;  // apq negligible (theta = inf / nan)
                    if (apq == 0) t = 0;                                   // identity rotation = the reference's `continue`
                    const double c = 1 / sqrt(t * t + 1), sn = t * c;
                    for (int m = 0; m < 6; m++) {
                        const double akp = A[m * 6 + p], akq = A[m * 6 + q];
                        A[m * 6 + p] = c * akp - sn * akq;
                        A[m * 6 + q] = sn * akp + c * akq;
                    }
                    for (int m = 0; m < 6; m++) {
                        const double apk = A[p * 6 + m], aqk = A[q * 6 + m];
                        A[p * 6 + m] = c * apk - sn * aqk;
                        A[q * 6 + m] = sn * apk + c * aqk;
                    }
                    for (int m = 0; m < 6; m++) {
                        const double vkp = V[m * 6 + p], vkq = V[m * 6 + q];
                        V[m * 6 + p] = c * vkp - sn * vkq;
                        V[m * 6 + q] = sn * vkp + c * vkq;
                    }
                }
        }
        double thresh = 0;
        for (int i = 0; i < 6; i++) thresh += fabs(A[i * 7]);
        thresh *= 2 * 2.220446049250313e-16;
        for (int i = 0; i < 36; i++) out[i] = 0;
        for (int m = 0; m < 6; m++) {
            const double w = A[m * 7];
            if (!(fabs(w) > thresh)) continue;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) out[i * 6 + j] += V[i * 6 + m] * V[j * 6 + m] / w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 36; i++) Ainv[i] = out[i];
    __syncthreads();
}

template <int B>
__device__ __forceinline__ void bwd_path1(const KArgs& a, int slot, double* s_part, double* s_tot, double* s_max) {
    if (slot >= a.bwd.n_sel[0]) return;
    const int h = a.bwd.sel[slot];
    const int P = a.H * a.W;
    double* __restrict__ g = a.bwd.grad1 + (size_t)slot * P * 3;
    const int* mi = a.bwd.map_info + 4 * slot;
    const int buf = mi[0], n_inl = mi[1];
    if (buf < 0 || n_inl < 4) {  // no accepted re-fit (empty inlier map) or fewer than 4 inliers: zero gradient
        for (int i = threadIdx.x; i < 3 * P; i += B) g[i] = 0;
        return;
    }
    const uint8_t* __restrict__ map = a.bwd.maps + ((size_t)slot * 2 + buf) * P;
    const int e = expert_of(a, h);
    const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
    const Cam cam = make_cam(a);
    const double* hp = a.bwd.ref_hyps + (size_t)h * 6;
    const double rv[3] = {hp[0], hp[1], hp[2]};
    const double t[3] = {hp[3], hp[4], hp[5]};
    double R[9], dRdr[27];
    rodrigues_vec2mat<true>(rv, R, dRdr);

    double U[21];
#pragma unroll
    for (int k = 0; k < 21; k++) U[k] = 0;
    for (int i = threadIdx.x; i < P; i += B) {
        if (!map[i]) continue;
        const int row = i / a.W, col = i - row * a.W;
        double r6[6];
        norm_jac_row(R, dRdr, t, cam, mx[i], mx[P + i], mx[2 * P + i], cell_px(a, col), cell_py(a, row), a.max_reproj, r6);
        int k = 0;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p; q < 6; q++) U[k++] += r6[p] * r6[q];
    }
    block_sum28<21, B>(U, s_part, s_tot);
    double Ainv[36];
    if (!inv_spd6(U, Ainv)) pinv_sym6_lds(U, Ainv, s_part);  // workgroup-uniform: every lane holds the same sums
    double dL[6];
#pragma unroll
    for (int k = 0; k < 6; k++) dL[k] = a.bwd.dloss[(size_t)slot * 6 + k];

    double vmax = 0;
    for (int i = threadIdx.x; i < P; i += B) {
        double o0 = 0, o1 = 0, o2 = 0;
        if (map[i]) {
            const int row = i / a.W, col = i - row * a.W;
            const float X = mx[i], Y = mx[P + i], Z = mx[2 * P + i];
            const float px = cell_px(a, col), py = cell_py(a, row);
            double r6[6], c6[6], dNdO[3];
            norm_jac_row(R, dRdr, t, cam, X, Y, Z, px, py, a.max_reproj, r6);
#pragma unroll
            for (int p = 0; p < 6; p++) {
                double s = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) s += -Ainv[p * 6 + q] * r6[q];
                c6[p] = s;
                const double av = fabs(s);
                vmax = av > vmax ? av : vmax;
            }
            dproject_dobj(px, py, X, Y, Z, R, t, a.focal, a.ppx, a.ppy, a.max_reproj, dNdO);
#pragma unroll
            for (int p = 0; p < 6; p++) {
                o0 += dL[p] * (c6[p] * dNdO[0]);
                o1 += dL[p] * (c6[p] * dNdO[1]);
                o2 += dL[p] * (c6[p] * dNdO[2]);
            }
        }
        g[i] = o0;
        g[(size_t)P + i] = o1;
        g[2 * (size_t)P + i] = o2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(vmax, o);
        vmax = ov > vmax ? ov : vmax;
    }
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = vmax;
    __syncthreads();
    vmax = s_max[0];
#pragma unroll
    for (int w = 1; w < B / 64; w++) vmax = s_max[w] > vmax ? s_max[w] : vmax;
    if (vmax > 10) {  // "clamping for stability" (esac.cpp:436-437): the whole pseudo-inverse is dropped
        for (int i = threadIdx.x; i < 3 * P; i += B) g[i] = 0;
    }
}

// ================================================================= K8: path II
// One workgroup per slot.  d score / d coordinates of the INITIAL hypothesis: every cell directly through its own
// reprojection error, and the four sampled cells through the pose (dPNP: central differences of the 4-point solver
// with a float step of 1e-3, esac_derivative.h:128-185).  sum_cells (dRE * dErr/dPose) * dPose/dObj is linear in the
// per-cell term, so the 6 pose-space sums are reduced first and multiplied with the 6x12 dPNP matrix once.
template <int B>
__device__ __forceinline__ void bwd_path2(const KArgs& a, int slot, double* s_part, double* s_tot, double (*s_sol)[6], double* s_J, int& s_bad) {
    if (slot >= a.bwd.n_sel[0]) return;
    const int h = a.bwd.sel[slot];
    const int P = a.H * a.W;
    double* __restrict__ g = a.bwd.grad2 + (size_t)slot * P * 3;
    const int e = expert_of(a, h);
    const float* __restrict__ mx = a.sc + (size_t)e * 3 * P;
    const Cam cam = make_cam(a);
    const int* sxy = a.sample_xy + (size_t)h * 8;
    if (threadIdx.x == 0) s_bad = 0;
    if (threadIdx.x < 72) s_J[threadIdx.x] = 0;
    __syncthreads();

    // ---- dPNP: lane 2q / 2q+1 solves with coordinate q (point q/3, axis q%3) moved by +eps / -eps
    if (threadIdx.x < 18) {
        const float eps = 0.001f;
        const int q = threadIdx.x >> 1;
        const bool minus = threadIdx.x & 1;
        float obj[12];
        double mu[4], mv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = sxy[2 * j + 1] * a.W + sxy[2 * j];
            obj[3 * j] = mx[idx];
            obj[3 * j + 1] = mx[P + idx];
            obj[3 * j + 2] = mx[2 * P + idx];
            mu[j] = (double)cell_px(a, sxy[2 * j]);
            mv[j] = (double)cell_py(a, sxy[2 * j + 1]);
        }
        // the reference perturbs in place (+eps, -2eps, +eps), which does not always restore the float value:
        // coordinates handled before q carry that residue into this solve
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float up = obj[k] + eps;
            const float down = up - 2 * eps;
            const float back = down + eps;
            obj[k] = k < q ? back : (k == q ? (minus ? down : up) : obj[k]);
        }
        V3 Pt[4];
#pragma unroll
        for (int j = 0; j < 4; j++) Pt[j] = V3{(double)obj[3 * j], (double)obj[3 * j + 1], (double)obj[3 * j + 2]};
        double Rp[9], Tp[3], rvec[3];
        if (p3p_4pt(Pt, mu, mv, cam, Rp, Tp)) {
            rodrigues_mat2vec(Rp, rvec);
            double* o = s_sol[threadIdx.x];
            o[0] = rvec[0]; o[1] = rvec[1]; o[2] = rvec[2];
            o[3] = Tp[0]; o[4] = Tp[1]; o[5] = Tp[2];
        } else {
            atomicOr(&s_bad, 1);
        }
    }
    __syncthreads();
    double jv = 0;
    if (threadIdx.x < 54) {
        const int k = threadIdx.x / 9, q = threadIdx.x - 9 * k;
        const float eps = 0.001f;
        jv = (s_sol[2 * q][k] - s_sol[2 * q + 1][k]) / (double)(2 * eps);
        if (jv != jv) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    // getMax(abs) > 10 -> the whole matrix is dropped (esac_derivative.h:274-275)
    double amax = bad ? 0.0 : fabs(jv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(amax, o);
        amax = ov > amax ? ov : amax;
    }
    if (threadIdx.x < 54 && !bad && !(amax > 10)) {  // all 54 entries sit in wavefront 0
        const int k = threadIdx.x / 9, q = threadIdx.x - 9 * k;
        s_J[k * 12 + q] = jv;
    }
    __syncthreads();

    // ---- every cell: direct term + pose-space sums
    const double* hp = a.hyps + (size_t)h * 6;
    const double rv[3] = {hp[0], hp[1], hp[2]};
    const double t[3] = {hp[3], hp[4], hp[5]};
    double R[9], dRdr[27];
    rodrigues_vec2mat<true>(rv, R, dRdr);
    const float scale = a.alpha / a.W / a.H;
    const double sg = a.bwd.sgrad[h];
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < P; i += B) {
        const int row = i / a.W, col = i - row * a.W;
        const float X = mx[i], Y = mx[P + i], Z = mx[2 * P + i];
        const float px = cell_px(a, col), py = cell_py(a, row);
        float err = project_exact_err(R, t, cam, X, Y, Z, px, py);
        err = err < a.max_reproj ? err : a.max_reproj;
        double st = a.beta * (err - a.tau);  // float product, widened
        st = 1 / (1 + exp(-st));
        double dRE = -st * (1 - st) * a.beta * sg;
        dRE *= scale;
        double dPdO[3], r6[6];
        dproject_dobj(px, py, X, Y, Z, R, t, a.focal, a.ppx, a.ppy, a.max_reproj, dPdO);
        g[i] = dPdO[0] * dRE;
        g[(size_t)P + i] = dPdO[1] * dRE;
        g[2 * (size_t)P + i] = dPdO[2] * dRE;
        norm_jac_row(R, dRdr, t, cam, X, Y, Z, px, py, a.max_reproj, r6);
#pragma unroll
        for (int k = 0; k < 6; k++) S[k] += dRE * r6[k];
    }
    block_sum28<6, B>(S, s_part, s_tot);
    __syncthreads();  // the slab writes above are visible to the lanes that add the support terms
    if (threadIdx.x < 12) {
        const int m = threadIdx.x;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += S[k] * s_J[k * 12 + m];
        const int j = m / 3, c = m - 3 * j;
        const int idx = sxy[2 * j + 1] * a.W + sxy[2 * j];
        g[(size_t)c * P + idx] += s;
    }
}

// ================================================================= K7 + K8 in ONE launch
// Neither path reads the other's output (path I: the refined pose, its inlier set, dLoss; path II: the initial hypothesis, its four
// cells, d E / d score), and with a few dozen slots either one is a chain one workgroup long on an otherwise empty chip: workgroups
// [0, slots) take path I, [slots, 2 slots) path II -- 47 + 36 us in sequence became max(path I, path II), with no second stream
// and no event between them (round 6 measured that variant: the fork and the join cost 7 us each).
template <int B>
__global__ __launch_bounds__(B) void k_bwd_paths(KArgs a) {
    __shared__ double s_part[28 * (B / 64) > 108 ? 28 * (B / 64) : 108];
    __shared__ double s_tot[28];
    __shared__ double s_max[B / 64];
    __shared__ double s_sol[18][6];
    __shared__ double s_J[72];
    __shared__ int s_bad;
    const int slots = (int)gridDim.x >> 1;
    if ((int)blockIdx.x < slots) bwd_path1<B>(a, (int)blockIdx.x, s_part, s_tot, s_max);
    else                         bwd_path2<B>(a, (int)blockIdx.x - slots, s_part, s_tot, s_sol, s_J, s_bad);
}

// ================================================================= K9: ordered accumulation into the float tensor
// grid = (tiles of 3P elements, experts).  Wavefront 0 first compacts, in slot order, the slots whose hypothesis
// belongs to this expert (ballot prefix) together with their probabilities into LDS; then one thread per tensor
// element walks that list with unconditional, coalesced slab loads (slabs are planar like the tensor), U loads in
// flight, while the float `+=` chain itself stays in slot order.
__global__ __launch_bounds__(256) void k_bwd_accumulate(KArgs a) {
    __shared__ int s_slot[ESAC_BWD_SLOTS_K];
    __shared__ double s_prob[ESAC_BWD_SLOTS_K];
    __shared__ int s_count;
    const int P = a.H * a.W;
    const int e = blockIdx.y;
    const int n_sel = a.bwd.n_sel[0];
    // more slots than the workspace holds: nothing is accumulated, the host grows it and retries.  The slots were refined by
    // teams and one of them timed out (a member never became resident): nothing is accumulated either, the host refines the
    // slots again with one workgroup each.  (Both are the same answer in every workgroup of the launch.)
    const bool team_failed = a.bwd.team && __hip_atomic_load(a.coop_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.bwd.team_tag;
    const bool skip = a.bwd.n_sel[1] > a.bwd.cap || team_failed;
    if (!skip) {
        if (threadIdx.x < 64) {
            int count = 0;
            for (int base = 0; base < n_sel; base += 64) {
                const int slot = base + (int)threadIdx.x;
                int h = 0;
                bool mine = false;
                if (slot < n_sel) {
                    h = a.bwd.sel[slot];
                    mine = expert_of(a, h) == e;
                }
                const unsigned long long bal = __ballot(mine);
                if (mine) {
                    const int pos = count + __popcll(bal & ((1ull << threadIdx.x) - 1ull));
                    s_slot[pos] = slot;
                    s_prob[pos] = a.bwd.probs[h];
                }
                count += __popcll(bal);
            }
            if (threadIdx.x == 0) s_count = count;
        }
        __syncthreads();
        const int n = s_count;
        const int rem = blockIdx.x * blockDim.x + threadIdx.x;  // c * P + cell
        if (n > 0 && rem < 3 * P) {
            float* o = a.bwd.out_grad + (size_t)e * 3 * P + rem;
            float v = *o;
            constexpr int U = 8;
            for (int base = 0; base < n; base += U) {
                double t[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = base + u < n ? base + u : n - 1;  // clamped: the loads are unconditional
                    const size_t k = (size_t)s_slot[q] * 3 * P + rem;
                    t[u] = s_prob[q] * a.bwd.grad1[k] + a.bwd.grad2[k];
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (base + u < n) v = (float)((double)v + t[u]);  // float += double (esac.cpp:501-506)
            }
            *o = v;
        }
    }
    // The call's record -- expected loss, slots, entropy, out-of-range flag (k_bwd_loss wrote them), "a slot team timed out" --
    // goes to the pinned slot the host polls (as the forward's record does: no copy, no stream-completion round trip), by the
    // LAST workgroup of this launch to finish: every one arrives at a counter once its stores are issued.
    if (!a.result_pin) return;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    int last = 0;
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(a.bwd.arrived, 1) == (int)(gridDim.x * gridDim.y) - 1;
        if (last) a.bwd.arrived[0] = 0;  // (every workgroup has arrived: ready for the next call)
    }
    if (!__shfl(last, 0)) return;
    const int lane = threadIdx.x;
    const double v = lane < 4 ? a.bwd.out[lane] : lane == 4 ? (team_failed ? 1.0 : 0.0) : lane == 32 ? a.epoch : 0.0;
    pin_deliver(a.result_pin, v);
}

// ---------------------------------------------------------------- launchers
static inline int slot_grid(const KArgs& a) { return a.N < a.bwd.cap ? a.N : a.bwd.cap; }

void launch_bwd_select(const KArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_bwd_select<1024>, dim3(1), dim3(1024), 0, s, a); }
void launch_bwd_loss(const KArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_bwd_loss<BWD_B>, dim3(1), dim3(BWD_B), 0, s, a); }
void launch_bwd_paths(const KArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_bwd_paths<BWD_B>, dim3(2 * slot_grid(a)), dim3(BWD_B), 0, s, a);
}
void launch_bwd_accumulate(const KArgs& a, hipStream_t s) {
    const int per_expert = 3 * a.H * a.W;
    hipLaunchKernelGGL(k_bwd_accumulate, dim3((unsigned)((per_expert + 255) / 256), a.E), dim3(256), 0, s, a);
}

}  // namespace esac
