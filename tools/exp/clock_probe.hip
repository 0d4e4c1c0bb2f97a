// EXPERIMENT: effective shader clock while another kernel loads the chip.  One wave spins for ~dur_us and reports
// (shader cycles elapsed) / (100 MHz wall ticks elapsed).
#include <hip/hip_runtime.h>
extern "C" __global__ void clock_probe(unsigned long long* out, long long wall_ticks) {
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = clock64();
    unsigned long long r1 = r0;
    while ((long long)(r1 - r0) < wall_ticks) r1 = wall_clock64();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
extern "C" void launch_clock_probe(unsigned long long* out, long long wall_ticks, hipStream_t s) {
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s, out, wall_ticks);
}
