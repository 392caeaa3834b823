// Cross-workgroup barrier latency on MI355X: G workgroups exchange 28 doubles + a flag per round through global memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
// stride: active workgroups are blockIdx.x = stride * m (stride 8 keeps all members on one XCD: WG i -> XCD i % 8)
__global__ __launch_bounds__(256) void k(double* part, unsigned* flag, long long* out, int G, int stride, int rounds) {
  if (blockIdx.x % stride) return;
  const int m = blockIdx.x / stride;
  if (m >= G) return;
  long long w0 = wall_clock64();
  double acc = 0;
  for (int r = 1; r <= rounds; r++) {
    // publish 28 partials for this round (parity-buffered), then bump the round counter
    if (threadIdx.x < 28) part[((r & 1) * G + m) * 32 + threadIdx.x] = (double)(r + m + threadIdx.x);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r * G)) {}
    }
    __syncthreads();
    if (threadIdx.x < 28) {
      double t = 0;
      for (int g = 0; g < G; g++) t += __hip_atomic_load(&part[((r & 1) * G + g) * 32 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += t;
    }
  }
  long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * m] = w1 - w0; }
  if (threadIdx.x == 1) out[2 * m + 1] = (long long)acc;
}
int main() {
  double* part; unsigned* flag; long long* out;
  hipMalloc(&part, 2 * 64 * 32 * 8); hipMalloc(&flag, 64); hipMalloc(&out, 1024);
  const int rounds = 2000;
  for (int stride : {1, 8}) for (int G : {1, 2, 4, 8, 16}) {
    hipMemset(flag, 0, 64);
    hipLaunchKernelGGL(k, dim3(G * stride), dim3(256), 0, 0, part, flag, out, G, stride, rounds);
    long long h[64]; hipMemcpy(h, out, 2 * G * 8, hipMemcpyDeviceToHost);
    printf("stride %d G %2d: %.3f us per round (wg0), check %lld\n", stride, G, h[0] * 0.01 / rounds, h[1]);
  }
  return 0;
}
