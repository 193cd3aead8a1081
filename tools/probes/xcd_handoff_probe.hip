// Does data written by one kernel stay in the writing XCD's L2 for the next kernel on the same stream?  Kernel A: block b
// writes a 4 KB record (plain stores).  Kernel B: block b reads the record of block b + shift by a chain of DEPENDENT loads
// (each load's address comes from the previous value) and reports clock64 ticks per load.  shift = 0, 8, 16: same XCD as the
// writer (block b runs on XCD b % 8); shift = 1 ... 7: another XCD.  Also with a third kernel in between that touches
// nothing of it.   build: hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_handoff_probe.hip -o /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REC 512  // 8-byte words per record
__global__ void k_write(unsigned long long* d, int nb, int salt) {
  unsigned long long* r = d + (size_t)blockIdx.x * REC;
  // a permutation chain inside the record: word i holds the index of the next word (stride 17 walk over 512 words, one per 64 B line... 8 words per line)
  for (int i = threadIdx.x; i < REC; i += blockDim.x) r[i] = (unsigned long long)((i + 8 * 17 + salt * 8) % REC);
}
__global__ void k_idle(int* p) { if (p && threadIdx.x == 999) *p = 0; }
__global__ void k_read(const unsigned long long* d, int nb, int shift, long long* ticks, unsigned long long* sink) {
  if (threadIdx.x != 0) return;
  const unsigned long long* r = d + (size_t)((blockIdx.x + shift) % nb) * REC;
  unsigned long long idx = 0;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < 32; ++i) idx = r[idx];
  const long long t1 = clock64();
  ticks[blockIdx.x] = (t1 - t0) / 32;
  sink[blockIdx.x] = idx;
}
int main() {
  const int nb = 1024;
  unsigned long long *d, *sink;
  long long* ticks;
  hipMalloc(&d, (size_t)nb * REC * 8);
  hipMalloc(&sink, nb * 8);
  hipMalloc(&ticks, nb * 8);
  std::vector<long long> h(nb);
  for (int between = 0; between < 2; ++between)
    for (int shift : {0, 8, 16, 1, 2, 4, 7}) {
      std::vector<long long> med;
      for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k_write, dim3(nb), dim3(64), 0, 0, d, nb, rep + shift);
        if (between) hipLaunchKernelGGL(k_idle, dim3(nb), dim3(64), 0, 0, (int*)nullptr);
        hipLaunchKernelGGL(k_read, dim3(nb), dim3(64), 0, 0, d, nb, shift, ticks, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ticks, nb * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        med.push_back(h[nb / 2]);
      }
      std::sort(med.begin(), med.end());
      printf("kernel in between: %d  reader reads block b + %2d (%s XCD): median ticks per dependent load %lld\n", between, shift, shift % 8 == 0 ? "same " : "other", med[2]);
    }
  return 0;
}
