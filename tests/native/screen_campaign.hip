// screen_campaign.hip -- TEST / MEASUREMENT ONLY: the sampling screen against the fp64 route ON THE DEVICE.
//
// The screen (esac_amd/csrc/p3p_screen.hpp) must never reject a try the exact route accepts.  tests/native/
// p3p_screen_probe.cpp checks that on the host build of the same source -- but there scr_rcpf / scr_rsqf / scr_sqrtf,
// the Newton seeds and every contracted expression are IEEE operations of the host compiler, not the v_rcp_f32 /
// v_rsq_f32 / v_sqrt_f32 estimates and FMA contraction the kernels run.  This file compiles the SAME headers for gfx950
// and runs the campaign where the product runs: every thread draws four distinct random cells per try (splitmix, as the
// host probe), decides the try by the exact route (p3p_4pt + the reference's 4-point acceptance test, exactly what
// k_sample_decide / k_sample do) and by the screen as k_sample_prescreen evaluates it (screen_setup + p3p_screen_roots),
// and counts: tries, fp64-accepted, "maybe" at the kernels' margin, fp64-accepted tries the screen would REJECT at
// margins 0.5 / 1 / 2 / 3 px (the kernels use 3), the largest screen error of an accepted try.
// Built by tests/native/build.py (hipcc --offload-arch=gfx950), driven by scripts/dev/screen_campaign_device.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../esac_amd/csrc/pose_math.hpp"
#include "../../esac_amd/csrc/p3p_screen.hpp"

using namespace esac;

__device__ __forceinline__ uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// the reference's acceptance test of a solved sample (esac_util.h:202-221): cv::Rodrigues round trip, projectPoints
// arithmetic, all four errors < tau
__device__ bool accept64(const double Rp[9], const double Tp[3], const float Pf[4][3], const double mu[4], const double mv[4], const Cam& cam, double tau) {
    double rvec[3], R[9];
    rodrigues_mat2vec(Rp, rvec);
    rodrigues_vec2mat<false>(rvec, R, nullptr);
    for (int j = 0; j < 4; j++)
        if (!((double)project_exact_err(R, Tp, cam, Pf[j][0], Pf[j][1], Pf[j][2], (float)mu[j], (float)mv[j]) < tau)) return false;
    return true;
}

// out (per launch, atomically accumulated): [0] tries, [1] fp64-accepted, [2] screen "maybe" (delicate),
// [3] maybe or within tau + 3 px (what goes to the fp64 decision), [4..7] fp64-accepted tries rejected at margin 0.5 / 1 / 2 / 3 px;
// maxerr: largest screen error of an fp64-accepted, non-delicate try (float bits, atomicMax on non-negative floats)
__global__ __launch_bounds__(64) void k_campaign(const float* __restrict__ coords, int H, int W, int sub, float f, float cx, float cy, float tau,
                                                 uint64_t seed, int tries_per_thread, unsigned long long* out, unsigned* maxerr) {
    const int P = H * W;
    const Cam cam{(double)f, (double)f, (double)cx, (double)cy};
    uint64_t s = seed + 0x632BE59BD9B4E019ull * (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long n_acc = 0, n_maybe = 0, n_keep = 0, rej[4] = {0, 0, 0, 0};
    float worst = 0.0f;
    for (int it = 0; it < tries_per_thread; it++) {
        int cxs[4], cys[4];
        for (int j = 0; j < 4; j++) {
            for (;;) {
                const uint64_t r = splitmix(s);
                const int x = (int)((r & 0xffffffffu) % (uint32_t)(W - 1)), y = (int)((r >> 32) % (uint32_t)(H - 1));
                bool dup = false;
                for (int k = 0; k < j; k++) dup |= cxs[k] == x && cys[k] == y;
                if (!dup) {
                    cxs[j] = x;
                    cys[j] = y;
                    break;
                }
            }
        }
        float Pf[4][3], muf[4], mvf[4];
        V3 Pt[4];
        double mu[4], mv[4];
        for (int j = 0; j < 4; j++) {
            const int idx = cys[j] * W + cxs[j];
            Pf[j][0] = coords[idx]; Pf[j][1] = coords[P + idx]; Pf[j][2] = coords[2 * P + idx];
            Pt[j] = V3{(double)Pf[j][0], (double)Pf[j][1], (double)Pf[j][2]};
            muf[j] = (float)(cxs[j] * sub + sub / 2); mvf[j] = (float)(cys[j] * sub + sub / 2);
            mu[j] = muf[j]; mv[j] = mvf[j];
        }
        double Rp[9], Tp[3], reproj2 = 0;
        const bool solved = p3p_4pt(Pt, mu, mv, cam, Rp, Tp, &reproj2);
        const bool ok = solved && accept64(Rp, Tp, Pf, mu, mv, cam, (double)tau);
        ScreenSetup S;
        const float e = screen_setup(Pt, mu, mv, cam, S) ? p3p_screen_roots(S, Pf, muf[3], mvf[3], f, cx, cy) : INFINITY;
        const bool delicate = e == ESAC_SCREEN_MAYBE || !(e == e);
        n_acc += ok;
        n_maybe += delicate;
        n_keep += delicate || !(e > tau + 3.0f);
        const float margins[4] = {0.5f, 1.0f, 2.0f, 3.0f};
        for (int k = 0; k < 4; k++) rej[k] += ok && !delicate && (e > tau + margins[k]);
        if (ok && !delicate && e > worst) worst = e;
    }
    // one atomic per wavefront and counter
    unsigned long long v[8] = {(unsigned long long)tries_per_thread, n_acc, n_maybe, n_keep, rej[0], rej[1], rej[2], rej[3]};
    for (int k = 0; k < 8; k++) {
        unsigned long long t = v[k];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (threadIdx.x == 0) atomicAdd(out + k, t);
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
    if (threadIdx.x == 0) atomicMax(maxerr, __float_as_uint(worst));
}

// host entry: h_coords float[3,H,W]; out: 8 counters + [8] = largest screen error of an accepted try.  Returns 0 / HIP error.
extern "C" int screen_campaign(const float* h_coords, int H, int W, int sub, float f, float cx, float cy, float tau, uint64_t seed,
                               int blocks, int tries_per_thread, int launches, double* out) {
    float* d_coords = nullptr;
    unsigned long long* d_out = nullptr;
    unsigned* d_max = nullptr;
    const size_t bytes = (size_t)3 * H * W * sizeof(float);
    if (hipMalloc((void**)&d_coords, bytes) != hipSuccess || hipMalloc((void**)&d_out, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&d_max, sizeof(unsigned)) != hipSuccess)
        return 1;
    (void)hipMemcpy(d_coords, h_coords, bytes, hipMemcpyHostToDevice);
    (void)hipMemset(d_out, 0, 8 * sizeof(unsigned long long));
    (void)hipMemset(d_max, 0, sizeof(unsigned));
    for (int l = 0; l < launches; l++)
        hipLaunchKernelGGL(k_campaign, dim3(blocks), dim3(64), 0, 0, d_coords, H, W, sub, f, cx, cy, tau, seed + 0x9E3779B97F4A7C15ull * (uint64_t)(l + 1),
                           tries_per_thread, d_out, d_max);
    const hipError_t e = hipDeviceSynchronize();
    unsigned long long h[8];
    unsigned hm = 0;
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hm, d_max, sizeof(hm), hipMemcpyDeviceToHost);
    for (int k = 0; k < 8; k++) out[k] = (double)h[k];
    float fm;
    memcpy(&fm, &hm, sizeof(fm));
    out[8] = fm;
    (void)hipFree(d_coords);
    (void)hipFree(d_out);
    (void)hipFree(d_max);
    return e == hipSuccess ? 0 : (int)e;
}
