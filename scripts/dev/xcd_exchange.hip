// xcd_exchange.hip -- what does ONE all-to-all exchange of 24 doubles between G workgroups cost on MI355X, and where do
// the workgroups of a launch land?  (round 4: the refinement's cooperation on the 60x80 grid)
//
// Protocol under test ("tagged granules", one hop): workgroup m stores its 24 values as 16-byte granules
// {double v, u64 tag}, tag = (epoch << 20 | round) ^ bits(v); every workgroup polls all G x 24 granules with L1-bypassing
// (sc1) 16-byte loads until the tag fits, then adds them in one fixed order.  No counter, no fence, no separate flag.
// Buffers are double-buffered by round parity (a workgroup can be at most one round ahead of the slowest).
//   mode 0: plain stores  (stay in the XCD's L2: only valid when all G workgroups share an XCD)
//   mode 1: sc1 stores    (write-through: valid at any placement)
//   mode 2: sc0 sc1 stores
// stride 8: members are blocks 0, 8, 16, ... (observed placement: block b -> XCD b % 8, so one XCD); stride 1: blocks
// 0..G-1 (G different XCDs).  Every block reports its XCC_ID and HW_ID.  `work` dependent fp64 FMAs between rounds.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/dev/xcd_exchange.hip -o /tmp/xcd && /tmp/xcd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u4 load_sc1(const u4* p) {
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void store_gran(u4* p, u4 v) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k(u4* gran, long long* out, int* place, int G, int stride, int rounds, int work, unsigned long long epoch) {
    extern __shared__ double lds[];  // big dynamic allocation: one workgroup per CU, like the refinement kernel
    int xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) {
        place[2 * blockIdx.x] = xcc;
        place[2 * blockIdx.x + 1] = hwid;
    }
    if (blockIdx.x % stride) return;
    const int m = blockIdx.x / stride;
    if (m >= G) return;
    const int k24 = threadIdx.x & 31, j = threadIdx.x >> 5;
    long long bad = 0, spins = 0;
    double carry = 1.0;
    const long long w0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        // "work": a dependent chain every lane walks (stands for the pass between two reductions)
        for (int i = 0; i < work; i++) carry = __builtin_fma(carry, 0.999999, 1e-9);
        u4* buf = gran + (size_t)(r & 1) * 8 * 32;
        if (threadIdx.x < 24) {
            const double v = (double)(r * 3 + m * 7 + (int)threadIdx.x) + carry * 0.0;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            const unsigned long long tag = ((epoch << 20) | (unsigned long long)r) ^ bits;
            u4 g = {(unsigned)bits, (unsigned)(bits >> 32), (unsigned)tag, (unsigned)(tag >> 32)};
            store_gran<MODE>(buf + m * 32 + threadIdx.x, g);
        }
        double val = 0;
        if (j < G && k24 < 24) {
            const unsigned long long want = (epoch << 20) | (unsigned long long)r;
            for (;;) {
                const u4 g = load_sc1(buf + j * 32 + k24);
                const unsigned long long bits = (unsigned long long)g.x | ((unsigned long long)g.y << 32);
                const unsigned long long tag = (unsigned long long)g.z | ((unsigned long long)g.w << 32);
                if ((tag ^ bits) == want) {
                    val = __longlong_as_double((long long)bits);
                    break;
                }
                spins++;
                if (spins > (1ll << 22)) break;
            }
        }
        lds[threadIdx.x] = val;
        __syncthreads();
        if (threadIdx.x < 24) {
            double t = 0;
            for (int w = 0; w < G; w++) t += lds[w * 32 + threadIdx.x];
            double expect = 0;
            for (int w = 0; w < G; w++) expect += (double)(r * 3 + w * 7 + (int)threadIdx.x);
            if (t != expect) bad++;
            lds[256 + threadIdx.x] = t;
        }
        __syncthreads();
        carry += lds[256 + 3] * 1e-30;
    }
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) out[4 * m] = w1 - w0;
    if (threadIdx.x < 24 && bad) atomicAdd((unsigned long long*)&out[4 * m + 1], (unsigned long long)bad);
    if (threadIdx.x == 32) out[4 * m + 2] = spins;
    if (threadIdx.x == 1) out[4 * m + 3] = (long long)carry;
}

int main() {
    u4* gran;
    long long* out;
    int* place;
    hipMalloc(&gran, 2 * 8 * 32 * 16);
    hipMalloc(&out, 4096);
    hipMalloc(&place, 2 * 256 * 4);
    hipMemset(gran, 0, 2 * 8 * 32 * 16);
    const int rounds = 4000;
    unsigned long long epoch = 1;
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // placement of a 64-block launch
    {
        hipMemset(out, 0, 4096);
        hipLaunchKernelGGL(k<1>, dim3(64), dim3(256), lds, 0, gran, out, place, 1, 8, 1, 0, epoch++);
        int h[128];
        hipMemcpy(h, place, sizeof h, hipMemcpyDeviceToHost);
        printf("placement (block: xcc, se, cu):");
        for (int b = 0; b < 64; b++) {
            if (b % 8 == 0) printf("\n ");
            printf(" %2d:%d/%d/%2d", b, h[2 * b] & 0xf, (h[2 * b + 1] >> 13) & 7, (h[2 * b + 1] >> 8) & 15);
        }
        printf("\n");
    }
    for (int work : {0, 200}) {
        for (int stride : {8, 1}) {
            for (int mode : {0, 1, 2}) {
                if (stride == 1 && mode == 0) continue;  // plain stores across XCDs are not a valid hand-off
                for (int G : {1, 2, 4, 8}) {
                    hipMemset(out, 0, 4096);
                    const dim3 grid(G * stride);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), lds, 0, gran, out, place, G, stride, rounds, work, epoch++);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), lds, 0, gran, out, place, G, stride, rounds, work, epoch++);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), lds, 0, gran, out, place, G, stride, rounds, work, epoch++);
                    long long h[32];
                    int pl[128];
                    if (hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
                    hipMemcpy(pl, place, sizeof pl, hipMemcpyDeviceToHost);
                    long long bad = 0;
                    for (int m = 0; m < G; m++) bad += h[4 * m + 1];
                    printf("work %3d stride %d mode %d G %d: %.3f us per round (wg0), spins/round %.1f, wrong sums %lld, xcc of members:", work, stride, mode, G,
                           h[0] * 0.01 / rounds, (double)h[2] / rounds, bad);
                    for (int m = 0; m < G; m++) printf(" %d", pl[2 * m * stride] & 0xf);
                    printf("\n");
                }
            }
        }
    }
    return 0;
}
