#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH>
__global__ void k(long long* out, double* sink, int iters) {
  double x[CH];
  for (int c = 0; c < CH; c++) x[c] = threadIdx.x * 1e-3 + c;
  const double m = 1.0000001, a = 1e-9;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CH; c++) x[c] = __builtin_fma(x[c], m, a);
  }
  long long c1 = clock64();
  double s = 0;
  for (int c = 0; c < CH; c++) s += x[c];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(int threads, long long* d, double* sink) {
  const int iters = 20000;
  hipLaunchKernelGGL(k<CH>, dim3(1), dim3(threads), 0, 0, d, sink, iters);
  long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("threads %4d chains %2d: %.2f cycles per fma per wave (%.2f per fma instr issued on the SIMD)\n", threads, CH,
         (double)h / iters / CH, (double)h / iters / CH / ((threads + 255) / 256));
}
int main() {
  long long* d; double* sink; hipMalloc(&d, 64); hipMalloc(&sink, 8 * 4096);
  for (int threads : {64, 256, 512, 1024}) { run<1>(threads, d, sink); run<4>(threads, d, sink); run<8>(threads, d, sink); run<16>(threads, d, sink); run<32>(threads, d, sink); }
  return 0;
}
