// valu_rate.hip -- issue cost of VALU instruction classes on gfx950, one wavefront per SIMD and four: v_fma_f32,
// v_rcp_f32, v_exp_f32, v_sqrt_f32, v_pk_fma_f32, v_fma_f64 (8 independent chains each, so the dependent-op latency does
// not show).  The figure scripts/profile_to_json.py prices instruction counts with is the FIRST column: kernel wall time
// (HIP events) x the nominal 2.4 GHz / wave-instructions issued per SIMD.  clock64() is printed next to it for
// reference only: it is NOT the shader clock under load (its rate against wall time wanders between 0.6 and 2.4 GHz
// with occupancy), so "cycles" read from it do not convert to time.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
    float c1 = 0.999f + 1e-9f * threadIdx.x, c2 = 0.001f, c3 = 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w[4], p1 = {0.999f, 0.999f}, p2 = {0.001f, 0.001f};
    double d[8], d1 = 0.999, d2 = 0.001;
    for (int i = 0; i < 4; i++) w[i] = f2{v[2 * i], v[2 * i + 1]};
    for (int i = 0; i < 8; i++) d[i] = v[i];
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // inline asm: the exact instruction, no SLP packing into v_pk_*
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                if (KIND == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0\n\tv_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c3));
                if (KIND == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 4) asm volatile("v_rcp_f32 %0, %0\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                if (KIND == 5 && (i & 1) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[i / 2]) : "v"(p1), "v"(p2));
                if (KIND == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d1), "v"(d2));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += v[i] + (float)d[i];
    for (int i = 0; i < 4; i++) s += w[i].x + w[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 1024 * sizeof(long long));
    const int iters = 2000;
    k<KIND><<<256, threads>>>(out, cyc, iters);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<KIND><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; i++) mean += h[i]; mean /= 256;
    const double insts = (double)iters * (KIND == 5 ? 32 : 64) * per_iter;  // wave-instructions per wave (KIND 5: a packed instruction every other slot)
    const int waves_per_simd = threads / 256;
    printf("%-28s %d wave(s)/SIMD: %5.2f cycles per wave-instruction (wall time x 2.4 GHz / instructions per SIMD)   [kernel %.3f ms; clock64: %.2f ticks per instruction per wave = %.2f GHz against wall time]\n",
           name, waves_per_simd, ms * 1e-3 * 2.4e9 / (insts * waves_per_simd), ms, mean / insts, mean / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32", 1, 256);
    run<0>("v_fma_f32", 1, 1024);
    run<1>("v_rcp_f32", 1, 256);
    run<1>("v_rcp_f32", 1, 1024);
    run<2>("v_exp_f32 + v_mul", 2, 256);
    run<2>("v_exp_f32 + v_mul", 2, 1024);
    run<3>("v_sqrt_f32", 1, 256);
    run<3>("v_sqrt_f32", 1, 1024);
    run<4>("v_rcp + 3 v_fma", 4, 256);
    run<4>("v_rcp + 3 v_fma", 4, 1024);
    run<5>("v_pk_fma_f32 (per pk instr)", 1, 256);
    run<5>("v_pk_fma_f32 (per pk instr)", 1, 1024);
    run<6>("v_fma_f64", 1, 256);
    run<6>("v_fma_f64", 1, 1024);
    return 0;
}
