// lat_probe.hip -- round 5: what a DEPENDENT instruction costs a lone wavefront on gfx950, by kernel wall time.
// The team refinement kernel runs one wavefront per SIMD; round 4 priced its serial sections by instruction COUNT at the
// issue rate measured with 8 independent chains (valu_rate.hip).  This probe measures the other axis: the latency of a
// dependent chain (v_fma_f64, v_rcp_f64, v_fma_f64 with a row_newbcast DPP operand, an LDS round trip, v_readlane -> VALU)
// for 1..16 independent chains, and the accuracy of the raw v_rcp_f64 and of its one-step / e + e^2 refinements.
//   hipcc --offload-arch=gfx950 -O3 lat_probe.hip -o lat_probe && ./lat_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

// KIND 0: v_fma_f64   1: v_rcp_f64 (+ nothing)   2: v_fma_f64 with DPP row_newbcast on src0   3: LDS write + read back
//      (2: v_fmac_f64_dpp d += bcast(d) * a -- the only DPP form of a DP ALU op on gfx9: VOP2 + row_newbcast)
//      4: v_readlane x2 -> v_fma_f64 with the SGPR pair   5: v_fma_f32   6: v_mul_f64   7: v_add_f64
//      8: v_mov_b32 dpp quad_perm x2 + v_add_f64 (one butterfly stage of a double)
template <int KIND, int CH>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    __shared__ double lds[256 * 16];
    double x[CH];
    float xf[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        x[c] = 1.0 + 1e-3 * (threadIdx.x + c);
        xf[c] = 1.0f + 1e-3f * (threadIdx.x + c);
    }
    const double m = 0.9999999, a = 1e-9;
    const float mf = 0.9999999f, af = 1e-9f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(m), "v"(a));
                if (KIND == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c]));
                if (KIND == 2) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[c]) : "v"(a));
                if (KIND == 3) {
                    lds[threadIdx.x * 16 + c] = x[c];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    x[c] = lds[(threadIdx.x ^ 1) * 16 + c];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (KIND == 4) {
                    const long long b = __double_as_longlong(x[c]);
                    const int lo = __builtin_amdgcn_readlane((int)b, 3), hi = __builtin_amdgcn_readlane((int)(b >> 32), 3);
                    const double s = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
                    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x[c]) : "s"(s), "v"(m), "v"(a));
                }
                if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"(mf), "v"(af));
                if (KIND == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (KIND == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if (KIND == 8) {
                    const long long b = __double_as_longlong(x[c]);
                    const int lo = __builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xf, 0xf, false), hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0xB1, 0xf, 0xf, false);
                    const double t = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(t));
                    asm volatile("v_mul_f64 %0, %0, 0.5" : "+v"(x[c]));
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += x[c] + (double)xf[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_ghz = 2.4;

template <int KIND, int CH>
void run(const char* name, int ops_per_slot) {
    double* out;
    hipMalloc(&out, 256 * 8 * sizeof(double));
    const int iters = 4000;
    hipLaunchKernelGGL((k<KIND, CH>), dim3(8), dim3(256), 0, 0, out, iters);  // 8 workgroups: one per XCD, one wavefront per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, CH>), dim3(8), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double slots = (double)iters * 16 * CH;
    const double ns_slot = ms * 1e6 / slots;
    printf("%-34s chains %2d: %7.2f ns per chain step = %6.1f cycles @%.1f GHz; per wave-instruction issued %6.2f cycles\n", name, CH, ns_slot * CH,
           ns_slot * CH * g_ghz, g_ghz, ns_slot * g_ghz / ops_per_slot);
    hipFree(out);
}

// accuracy of v_rcp_f64 and its refinements over random inputs
__global__ void k_rcp_acc(const double* d, double* e0, double* e1, double* e2, double* e3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = d[i];
    const double x0 = __builtin_amdgcn_rcp(v);
    const double e = __builtin_fma(-v, x0, 1.0);
    const double x1 = __builtin_fma(x0, e, x0);                       // one Newton step
    const double x2 = __builtin_fma(x0, __builtin_fma(e, e, e), x0);  // x0 (1 + e + e^2)
    double x3 = x1;                                                   // two Newton steps (fast_rcp)
    x3 = __builtin_fma(x3, __builtin_fma(-v, x3, 1.0), x3);
    e0[i] = x0; e1[i] = x1; e2[i] = x2; e3[i] = x3;
}

int main() {
    printf("dependent-chain latency of a lone wavefront per SIMD (8 workgroups x 256 threads), kernel wall time\n");
#define ROW(K, name, ops) run<K, 1>(name, ops); run<K, 2>(name, ops); run<K, 4>(name, ops); run<K, 8>(name, ops); run<K, 16>(name, ops);
    ROW(0, "v_fma_f64", 1)
    ROW(6, "v_mul_f64", 1)
    ROW(7, "v_add_f64", 1)
    ROW(5, "v_fma_f32", 1)
    ROW(1, "v_rcp_f64", 1)
    ROW(2, "v_fmac_f64 dpp row_newbcast", 1)
    ROW(8, "2 v_mov_dpp + v_add_f64 + v_mul_f64", 4)
    ROW(4, "2 v_readlane + v_fma_f64(sgpr)", 3)
    run<3, 1>("LDS write + read back (b64)", 2);
    run<3, 4>("LDS write + read back (b64)", 2);
    // accuracy
    const int n = 1 << 20;
    double* h = (double*)malloc(n * sizeof(double));
    srand(7);
    for (int i = 0; i < n; i++) {
        const double mant = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9;
        const int ex = rand() % 80 - 40;
        h[i] = ldexp(mant, ex) * ((rand() & 1) ? 1 : -1);
    }
    double *d, *e[4];
    hipMalloc(&d, n * sizeof(double));
    for (auto& p : e) hipMalloc(&p, n * sizeof(double));
    hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp_acc, dim3(n / 256), dim3(256), 0, 0, d, e[0], e[1], e[2], e[3], n);
    double* r = (double*)malloc(n * sizeof(double));
    const char* names[4] = {"v_rcp_f64 raw", "+ one Newton step", "x0 (1 + e + e^2)", "+ two Newton steps (fast_rcp)"};
    for (int q = 0; q < 4; q++) {
        hipMemcpy(r, e[q], n * sizeof(double), hipMemcpyDeviceToHost);
        double worst = 0;
        long exact = 0;
        for (int i = 0; i < n; i++) {
            const long double t = 1.0L / (long double)h[i];
            const double rel = (double)fabsl(((long double)r[i] - t) / t);
            if (rel > worst) worst = rel;
            if (r[i] == 1.0 / h[i]) exact++;
        }
        printf("%-32s max relative error %.3e (2^%.1f), equal to the correctly rounded 1/d in %.2f %% of %d inputs\n", names[q], worst, log2(worst), 100.0 * exact / n, n);
    }
    return 0;
}
