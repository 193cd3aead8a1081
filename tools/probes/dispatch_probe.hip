// What a workgroup costs the dispatcher: launches of N workgroups that return at once, by workgroup size, LDS allocation
// and register budget (the obstacle kernel's workgroups: 256 threads, 31 KB, 96 VGPRs).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NT, int WAVES>
__global__ __launch_bounds__(NT, WAVES) void k_empty(const int* __restrict__ p, int* out, int n) {
  extern __shared__ double sm[];
  if (p[0] == 12345) { sm[threadIdx.x] = 1.0; __syncthreads(); out[blockIdx.x] = (int)sm[(threadIdx.x + 1) % NT]; }
}
template <int NT, int WAVES>
__global__ __launch_bounds__(NT, WAVES) void k_load(const int* __restrict__ p, int* out, int n) {
  extern __shared__ double sm[];
  const int v = p[blockIdx.x & 1023];   // one dependent scalar load, like the job-list entry
  if (v == 12345) { sm[threadIdx.x] = 1.0; __syncthreads(); out[blockIdx.x] = (int)sm[(threadIdx.x + 1) % NT]; }
}
// the same, but the kernel owns scratch memory (a dynamically indexed local array on a path never taken) / 96 VGPRs
template <int NT, int WAVES>
__global__ __launch_bounds__(NT, WAVES) void k_scratch(const int* __restrict__ p, int* out, int n) {
  extern __shared__ double sm[];
  const int v = p[blockIdx.x & 1023];
  if (v == 12345) {
    double loc[40];
    for (int i = 0; i < 40; ++i) loc[i] = sm[(threadIdx.x + i) % NT];
    __syncthreads();
    double acc = 0;
    for (int i = 0; i < 40; ++i) acc += loc[(p[i] + i) % 40];
    out[blockIdx.x] = (int)acc;
  }
}
template <int NT, int WAVES>
__global__ __launch_bounds__(NT, WAVES) void k_vgpr(const int* __restrict__ p, int* out, int n) {
  extern __shared__ double sm[];
  const int v = p[blockIdx.x & 1023];
  if (v == 12345) {
    double loc[44];
#pragma unroll
    for (int i = 0; i < 44; ++i) loc[i] = sm[(threadIdx.x + i) % NT];
    __syncthreads();
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 44; ++i) acc = fma(acc, loc[i], sm[i]);
    out[blockIdx.x] = (int)acc;
  }
}
template <typename K>
static float time_it(K kern, int grid, int nt, size_t lds, const int* p, int* out, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, 0, p, out, grid);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, 0, p, out, grid);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}
int main() {
  int *p, *out; hipMalloc(&p, 4096); hipMemset(p, 0, 4096); hipMalloc(&out, 1 << 22);
  for (int grid : {1280, 6240, 24960}) {
    printf("grid %5d | 256 thr, no LDS %.1f us | 256 thr, 31 KB %.1f | 256 thr 31 KB + load %.1f | 64 thr no LDS %.1f | 512 thr 62 KB %.1f | 1024 thr 62 KB %.1f | 128 thr 15 KB %.1f\n", grid,
           time_it(k_empty<256, 5>, grid, 256, 0, p, out), time_it(k_empty<256, 5>, grid, 256, 31 * 1024, p, out),
           time_it(k_load<256, 5>, grid, 256, 31 * 1024, p, out), time_it(k_empty<64, 5>, grid, 64, 0, p, out),
           time_it(k_empty<512, 5>, grid / 2, 512, 62 * 1024, p, out), time_it(k_empty<1024, 4>, grid / 4, 1024, 62 * 1024, p, out),
           time_it(k_empty<128, 5>, grid * 2, 128, 15 * 1024, p, out));
  }
  for (int grid : {1280, 6240, 24960})
    printf("grid %5d | 256 thr 31 KB + load %.1f us | ... + scratch %.1f | ... + many VGPRs %.1f\n", grid, time_it(k_load<256, 5>, grid, 256, 31 * 1024, p, out),
           time_it(k_scratch<256, 5>, grid, 256, 31 * 1024, p, out), time_it(k_vgpr<256, 5>, grid, 256, 31 * 1024, p, out));
  return 0;
}
