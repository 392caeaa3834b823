// fork_join.hip -- round 6: what it costs to run two kernel chains of ONE call side by side on two streams of one GPU.
// Kernels spin for a set time on the 100 MHz wall clock and stamp their own start / end; the gaps are read from the stamps:
//   A: k1 (20 us) -> [fork] -> k2 (40 us) -> [join] -> k3 (5 us)        B: [after k1] kB (15 us)
// fork variants: an event recorded on A after k1 (default flags / DisableTiming / DisableTiming|ReleaseToDevice), k1's own
// completion event (hipExtLaunchKernelGGL stopEvent); join: hipStreamWaitEvent(A, event recorded on B after kB / kB's stop event).
// Also: a DEVICE-side hand-off with no event at all -- kB is launched on B right away and polls a flag that the last workgroup
// of k1 sets (only safe when kB's workgroups cannot keep k1's from becoming resident: a handful of small workgroups).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__device__ __forceinline__ long long wall() { long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__global__ void k_spin(long long ticks, long long* stamp, int* flag_set, int* flag_wait) {
    if (flag_wait && threadIdx.x == 0) {
        while (__hip_atomic_load(flag_wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const long long t0 = wall();
    while (wall() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamp[0] = t0; stamp[1] = wall(); }
    if (flag_set && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag_set, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    long long* d_st; CK(hipMalloc(&d_st, 8 * sizeof(long long)));
    int* d_flag; CK(hipMalloc(&d_flag, sizeof(int)));
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const char* names[] = {"no fork (A only: k1 k2 k3, kB after k3)", "event default flags", "event DisableTiming", "event DisableTiming|ReleaseToDevice",
                           "kernel stop events (hipExtLaunchKernelGGL)", "device flag (kB launched at once, polls)"};
    for (int mode = 0; mode < 6; mode++) {
        hipEvent_t ef, ej;
        const unsigned fl = mode == 1 ? 0 : mode == 3 ? (hipEventDisableTiming | hipEventReleaseToDevice) : hipEventDisableTiming;
        CK(hipEventCreateWithFlags(&ef, fl)); CK(hipEventCreateWithFlags(&ej, fl));
        std::vector<double> g12, g1B, g23, tot;
        for (int rep = 0; rep < 60; rep++) {
            CK(hipMemsetAsync(d_flag, 0, sizeof(int), A));
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
            if (mode == 0) {
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, 2000LL, d_st + 0, nullptr, nullptr);
                hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, A, 4000LL, d_st + 2, nullptr, nullptr);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, A, 500LL, d_st + 6, nullptr, nullptr);
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, A, 1500LL, d_st + 4, nullptr, nullptr);
            } else if (mode <= 3) {
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, 2000LL, d_st + 0, nullptr, nullptr);
                CK(hipEventRecord(ef, A));
                CK(hipStreamWaitEvent(B, ef, 0));
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, B, 1500LL, d_st + 4, nullptr, nullptr);
                CK(hipEventRecord(ej, B));
                hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, A, 4000LL, d_st + 2, nullptr, nullptr);
                CK(hipStreamWaitEvent(A, ej, 0));
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, A, 500LL, d_st + 6, nullptr, nullptr);
            } else if (mode == 4) {
                hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, nullptr, ef, 0, 2000LL, d_st + 0, nullptr, nullptr);
                CK(hipStreamWaitEvent(B, ef, 0));
                hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, B, nullptr, ej, 0, 1500LL, d_st + 4, nullptr, nullptr);
                hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, A, 4000LL, d_st + 2, nullptr, nullptr);
                CK(hipStreamWaitEvent(A, ej, 0));
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, A, 500LL, d_st + 6, nullptr, nullptr);
            } else {
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, 2000LL, d_st + 0, d_flag, nullptr);
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, B, 1500LL, d_st + 4, d_flag + 0, d_flag);  // (sets the flag again: harmless)
                hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, A, 4000LL, d_st + 2, nullptr, nullptr);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, A, 500LL, d_st + 6, nullptr, nullptr);
            }
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
            long long st[8]; CK(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
            if (rep < 10) continue;
            g12.push_back((st[2] - st[1]) * 0.01); g1B.push_back((st[4] - st[1]) * 0.01); g23.push_back((st[6] - st[3]) * 0.01);
            tot.push_back((std::max(st[7], st[5]) - st[0]) * 0.01);
        }
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%-46s k1.end->k2.start %6.2f us | k1.end->kB.start %6.2f us | k2.end->k3.start %6.2f us | k1.start->last end %7.2f us\n", names[mode], med(g12),
               med(g1B), med(g23), med(tot));
        (void)hipEventDestroy(ef); (void)hipEventDestroy(ej);
    }
    return 0;
}
