// clock_probe.hip — what clock do sparse, latency-bound launches run at?  A chain of tiny dependent kernels (one workgroup
// each, like the tail rounds of a solve) reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) at both ends of a fixed spin; the ratio is the engine clock during that kernel.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/clock_probe.hip -o tools/probes/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_spin(long long* out, int spin) {
  const long long c0 = clock64(), w0 = wall_clock64();
  long long c = c0;
  while (c - c0 < spin) c = clock64();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}
int main(int argc, char** argv) {
  const int n = 2000, spin = argc > 1 ? atoi(argv[1]) : 20000, grid = argc > 2 ? atoi(argv[2]) : 1;
  long long* d; hipMalloc(&d, sizeof(long long) * 2 * grid * n);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, 0, d + 2 * grid * i, spin);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * grid * n); hipMemcpy(h.data(), d, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double cs = 0, ws = 0; for (int i = 0; i < n; ++i) { cs += h[2 * grid * i]; ws += h[2 * grid * i + 1]; }
    printf("grid %d spin %d cycles: wall-clock rate %d kHz | engine clock during the kernels %.0f MHz | %.2f us per launch (spin alone %.2f us at that clock)\n",
           grid, spin, rate, cs / ws * rate / 1e3, 1e3 * ms / n, (cs / n) / (cs / ws * rate / 1e3));
  }
  return 0;
}
