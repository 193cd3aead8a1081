// Accuracy of v_rcp_f64 (and of 1 / 2 Newton steps on top of it) against IEEE division.
// Build: hipcc --offload-arch=gfx950 -O2 rcp_probe.hip -o rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double r = __builtin_amdgcn_rcp(a);
  r0[i] = r;
  r = fma(fma(-a, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-a, r, 1.0), r, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (s >> 11) * (1.0 / 9007199254740992.0);
    x[i] = ldexp(1.0 + u, (int)(s % 61) - 30);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  long ne1 = 0, ne2 = 0;
  for (int i = 0; i < n; ++i) {
    double t = 1.0 / x[i];
    e0 = fmax(e0, fabs(a[i] - t) / t);
    e1 = fmax(e1, fabs(b[i] - t) / t);
    e2 = fmax(e2, fabs(c[i] - t) / t);
    ne1 += b[i] != t;
    ne2 += c[i] != t;
  }
  printf("max relative error: v_rcp_f64 %.3e | +1 Newton %.3e (%ld of %d differ from 1/x) | +2 Newton %.3e (%ld differ)\n", e0, e1, ne1, n, e2, ne2);
  return 0;
}
