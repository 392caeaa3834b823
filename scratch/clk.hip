#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(long long* out, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  double x = threadIdx.x * 1e-3 + 1.0;
  for (int i = 0; i < iters; i++) x = x * 1.0000001 + 1e-9;   // dependent fp64 fma chain
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[3];
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("wallclock rate kHz %d, clockRate kHz %d\n", rate, clk);
  for (int rep = 0; rep < 6; rep++) {
    int iters = rep < 3 ? 20000 : 2000000;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, iters);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("iters %d: clock64 %lld wall %lld -> cycles/iter %.2f, eff clock %.1f MHz\n", iters, h[0], h[1], (double)h[0]/iters, (double)h[0]/h[1]*rate/1000.0);
  }
  // full-chip load
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 200000);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("full chip: cycles/iter %.2f eff clock %.1f MHz\n", (double)h[0]/200000, (double)h[0]/h[1]*rate/1000.0);
  }
  return 0;
}
