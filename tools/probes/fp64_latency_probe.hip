// Issue rate and dependent latency of the FP64 instructions the step kernel lives on, one wave on one SIMD.
// K independent chains of N dependent ops each; cycles/op = (t1-t0)/(N*K).  K=1 gives latency, large K the issue rate.
// Build: hipcc --offload-arch=gfx950 -O2 fp64_latency_probe.hip -o fp64_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K, int OP>
__global__ void chains(double* out, long long* cyc, int n, double a, double b) {
  const int lane = threadIdx.x;
  double x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = 1.0 / (1 + lane + k);
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (OP == 0) x[k] = fma(x[k], a, b);
      if (OP == 1) x[k] = x[k] * a;
      if (OP == 2) x[k] = x[k] + b;
      if (OP == 3) x[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[k], 0.0, 0, 0, 0);
      if (OP == 4) x[k] = __builtin_amdgcn_rcp(x[k]);
      if (OP == 5) {
        int lo = __double2loint(x[k]), hi = __double2hiint(x[k]);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
        x[k] = __hiloint2double(hi, lo);
      }
      if (OP == 6) x[k] = __shfl_xor(x[k], 1, 64);
      if (OP == 8 || OP == 9) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const unsigned lo = __double2loint(x[k]), hi = __double2hiint(x[k]);
        const u2 l = OP == 8 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const u2 h = OP == 8 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x[k] = __hiloint2double((int)h.y, (int)l.y);
      }
      if (OP == 10) x[k] = __shfl(x[k], (lane & 56) | 3, 64);  // ds_bpermute
      if (OP == 11) x[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[k]), 9), __builtin_amdgcn_readlane(__double2loint(x[k]), 9)) + b;
      if (OP == 12) { const double r0 = __builtin_amdgcn_rcp(x[k]); const double r1 = fma(fma(-x[k], r0, 1.0), r0, r0); x[k] = fma(fma(-x[k], r1, 1.0), r1, r1); }
      if (OP == 7) { float f = (float)x[k]; f = fmaf(f, (float)a, (float)b); x[k] = f; }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += x[k];
  out[lane] = s;
  if (lane == 0) cyc[0] = t1 - t0;
}

template <int K, int OP>
void run(const char* name, double* d, long long* c) {
  const int n = 512;
  chains<K, OP><<<1, 64>>>(d, c, n, 1.0000001, 1e-9);
  chains<K, OP><<<1, 64>>>(d, c, n, 1.0000001, 1e-9);
  long long h;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-28s K=%d : %6.1f cycles/op\n", name, K, (double)h / (n * K));
}

int main() {
  double* d;
  long long* c;
  hipMalloc(&d, 64 * 8);
  hipMalloc(&c, 8);
#define ALL(OP, NAME) run<1, OP>(NAME, d, c); run<2, OP>(NAME, d, c); run<4, OP>(NAME, d, c); run<8, OP>(NAME, d, c);
  ALL(0, "v_fma_f64")
  ALL(1, "v_mul_f64")
  ALL(2, "v_add_f64")
  ALL(3, "v_mfma_f64_4x4x4")
  ALL(4, "v_rcp_f64")
  ALL(5, "dpp quad_perm mov (x2)")
  ALL(6, "shfl_xor f64 (ds_swizzle/bpermute)")
  ALL(7, "cvt+v_fma_f32+cvt")
  ALL(8, "permlane32_swap (x2, +movs)")
  ALL(9, "permlane16_swap (x2, +movs)")
  ALL(10, "__shfl f64 (ds_bpermute x2)")
  ALL(11, "readlane x2 + v_add_f64")
  ALL(12, "rcp + 2 Newton (5 ops)")
  return 0;
}
