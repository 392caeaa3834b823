// launch_cost.hip -- round 5: what hipLaunchKernelGGL costs the HOST on this box: an empty kernel with an 8-byte and with a
// ~1 KB argument block, back to back and as the first launch after the stream went idle (what a blocking call's first
// launch is), with and without a hipStreamQuery in between.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
struct Big { double v[120]; };
__global__ void k_small(double* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_big(Big b, double* p) { if (p && threadIdx.x == 9999) p[0] = b.v[3]; }
static double now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
int main() {
    double* d; hipMalloc(&d, 64);
    volatile double* pin; hipHostMalloc((void**)&pin, 64, hipHostMallocMapped);
    hipStream_t s_own; hipStreamCreate(&s_own);
    Big b{};
    for (int which = 0; which < 2; which++) {
    hipStream_t s = which ? (hipStream_t)0 : s_own;
    printf("---- %s\n", which ? "the NULL (legacy default) stream -- torch's current stream unless the caller set another" : "a created stream");
    for (int mode = 0; mode < 6; mode++) {
        double first = 0, second = 0, third = 0;
        const int reps = 300;
        for (int r = 0; r < reps + 20; r++) {
            // idle stream
            hipStreamSynchronize(s);
            if (mode == 2 || mode == 3) (void)hipStreamQuery(s);
            if (mode >= 4) { for (volatile int w = 0; w < 20000; w++) {} }  // ~50 us of host spinning, stream long idle
            const double t0 = now_us();
            if (mode & 1) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, b, d); else hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d);
            const double t1 = now_us();
            if (mode & 1) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, b, d); else hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d);
            const double t2 = now_us();
            if (mode & 1) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, b, d); else hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, d);
            const double t3 = now_us();
            if (r >= 20) { first += t1 - t0; second += t2 - t1; third += t3 - t2; }
        }
        printf("%-58s first %.2f us, second %.2f us, third %.2f us\n",
               mode == 0 ? "8-byte args, after hipStreamSynchronize" : mode == 1 ? "968-byte args, after hipStreamSynchronize" :
               mode == 2 ? "8-byte args, after sync + hipStreamQuery" : mode == 3 ? "968-byte args, after sync + hipStreamQuery" :
               mode == 4 ? "8-byte args, after sync + 50 us of host spinning" : "968-byte args, after sync + 50 us of host spinning",
               first / reps, second / reps, third / reps);
    }
    }
    return 0;
}
