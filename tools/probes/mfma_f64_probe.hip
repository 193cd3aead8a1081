// Probe the lane -> element maps of v_mfma_f64_4x4x4f64 (4 blocks of a 4x4x4 FP64 product per wave).
// For every lane lb, B is one-hot at lb and A holds (lane+1): the non-zero D lanes tell which A lanes
// met B's element.  Build: hipcc --offload-arch=gfx950 -O2 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(double* out) {  // out[lb][lane]
  const int lane = threadIdx.x;
  for (int lb = 0; lb < 64; ++lb) {
    const double a = (double)(lane + 1);
    const double b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[lb * 64 + lane] = d;
  }
}

__global__ void chain_latency(double* out, long long* cyc, int n) {
  const int lane = threadIdx.x;
  double a = 1.0 + 1e-3 * lane, d = 1.0 / (1 + lane);
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.0, 0, 0, 0);  // D feeds B
  const long long t1 = clock64();
  double e = 1.0 / (1 + lane);
  for (int i = 0; i < n; ++i) e = fma(e, a, 0.25);  // dependent v_fma_f64 chain
  const long long t2 = clock64();
  out[lane] = d + e;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

int main() {
  double* d;
  long long* c;
  hipMalloc(&d, 64 * 64 * sizeof(double));
  hipMalloc(&c, 16);
  probe<<<1, 64>>>(d);
  std::vector<double> h(64 * 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  for (int lb = 0; lb < 64; ++lb) {
    printf("B lane %2d ->", lb);
    for (int l = 0; l < 64; ++l)
      if (h[lb * 64 + l] != 0.0) printf("  D[%2d]=A[%2d]", l, (int)h[lb * 64 + l] - 1);
    printf("\n");
  }
  chain_latency<<<1, 64>>>(d, c, 256);
  long long hc[2];
  hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("dependent chain, cycles per op: mfma_f64_4x4x4 %.1f   v_fma_f64 %.1f\n", hc[0] / 256.0, hc[1] / 256.0);
  return 0;
}
