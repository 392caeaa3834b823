// filler.hip -- TEST-ONLY: occupies the CUs of ONE XCD with workgroups that leave one by one (tests/test_gpu_semantics.py:
// a refinement team whose members become resident too far apart must time out within a bounded spin, fall back to one
// workgroup and, after two such calls, stop asking for teams).  Every workgroup takes 100 KB of LDS -- a team member needs
// 111 KB, so none fits beside it -- reads the XCD it landed on, leaves at once unless that is `xcd`, and otherwise spins until
// (its arrival number + 1) * step has passed on the 100 MHz wall clock.  Built by tests/native/build.py; not product code.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void k_filler(int xcd, long long step_ticks, unsigned* counter, int touch) {
    __shared__ volatile char pad[100 * 1024];
    __shared__ unsigned s_idx;
    pad[(threadIdx.x * 1601u + (unsigned)touch) % (100u * 1024u)] = (char)touch;  // (keeps the allocation)
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((xcc & 7) != xcd) return;
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) s_idx = atomicAdd(counter, 1u);
    __syncthreads();
    const long long until = t0 + (long long)(s_idx + 1) * step_ticks;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
}

extern "C" int filler_launch(void* stream, int xcd, float step_ms, int blocks, unsigned* d_counter) {
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_counter, 0, sizeof(unsigned), s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_filler, dim3(blocks), dim3(64), 0, s, xcd, (long long)(step_ms * 1e5f), d_counter, 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
