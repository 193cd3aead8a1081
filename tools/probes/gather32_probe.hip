// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of k_obstacle_gram:
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports half their bytes) and
// calls other widths and WRITE_SIZE uncalibrated.  Kernels with a KNOWN byte count each:
//   k_stream16     coalesced 16 B / lane streaming read of the whole array                       (the guide's case)
//   k_gather32     one random, 32-B aligned 32-B record per lane (2 x 16-B loads) out of an array >> L2 + Infinity Cache:
//                  every record a request of its own, no reuse (the obstacle kernel's voxel-record gather, cold)
//   k_gather32_hot the same out of a 16 MB array (fits L2 + Infinity Cache): what re-use does to the counter
//   k_store8       coalesced plain 8-B stores of a whole array                                  (blocks, records out)
//   k_store8_far   one 8-B store per 64-B line (the rest of the line untouched): partial-line writes
//   k_store64      whole 64-B records stored by 8 lanes x 8 B (the evaluation records)
// The probe prints the bytes each kernel moves by construction and its own event timing (an upper bound on the true
// request size: bytes / time cannot exceed what HBM delivers); tools/counter_calibration.py puts rocprofv3's counters of
// the same launches next to them.   build: hipcc --offload-arch=gfx950 -O3 tools/probes/gather32_probe.hip -o /tmp/gather32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__global__ void k_fill(double* a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)(i & 1023);
}
__global__ void k_stream16(const double2* a, size_t n16, double* sink) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == -1.0) *sink = s;
}
// per lane `per` records; record index = hash(lane id, k) mod n_rec; a record = 32 B = two 16-B loads.  TAG: a kernel name
// of its own per use (rocprofv3 reports counters by kernel name).  PAIR > 0: the lane reads a SECOND record PAIR bytes
// behind the first (first records 128-B aligned): PAIR = 32 the other half of the same 64 B, 64 the other half of the
// same 128-B line, 128 the next line -- how many requests that costs says what one request fetches.
template <int TAG, int PAIR>
__global__ void k_gather32(const double2* a, size_t n_rec, int per, uint64_t salt, double* sink) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  for (int k = 0; k < per; ++k) {
    size_t r = (size_t)(mix(gid * 64 + k + salt) % n_rec);
    if (PAIR > 0) r &= ~(size_t)7;  // 256-B aligned record pair
    const double2 v0 = a[2 * r], v1 = a[2 * r + 1];
    s += v0.x + v0.y + v1.x + v1.y;
    if (PAIR > 0) {
      const double2 w0 = a[2 * r + PAIR / 16], w1 = a[2 * r + PAIR / 16 + 1];
      s += w0.x + w0.y + w1.x + w1.y;
    }
  }
  if (s == -1.0) *sink = s;
}
__global__ void k_store8(double* a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = 1.0;
}
__global__ void k_store8_far(double* a, size_t n_lines) {  // word 3 of every 64-B line
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) a[8 * i + 3] = 2.0;
}
__global__ void k_store64(double* a, size_t n_rec, uint64_t salt) {  // 8 lanes write one whole 64-B record at a random place
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t rec = gid >> 3;
  const size_t r = (size_t)(mix(rec + salt) % n_rec);
  a[8 * r + (gid & 7)] = 3.0;
}

int main(int argc, char** argv) {
  const size_t big = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30);  // bytes of the big array (default 4 GiB)
  const size_t hot = 16ull << 20;
  double *a, *sink;
  CK(hipMalloc(&a, big));
  CK(hipMalloc(&sink, 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, big / 8);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, double bytes, const char* what, auto launch) {
    launch();  // warm-up (also a dispatch the counters see: the calibration script averages over the launches of a kernel)
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("PROBE %-16s bytes_by_construction %.0f  best_ms %.4f  GB/s_by_construction %.1f  launches 4  # %s\n", name, bytes, best, bytes / best / 1e6, what);
  };
  const int nwg = 256 * 16;
  timed("k_stream16", (double)big, "16 B per lane, coalesced, whole array once", [&] { hipLaunchKernelGGL(k_stream16, dim3(nwg), dim3(256), 0, 0, (const double2*)a, big / 16, sink); });
  const int per = 16;
  const size_t lanes = (size_t)nwg * 256;
  uint64_t salt = 1;
  timed("k_gather32<0, 0>", (double)lanes * per * 32, "random 32-B records, 32-B aligned, array >> L2 + MALL: 32 useful bytes per request", [&] {
    hipLaunchKernelGGL((k_gather32<0, 0>), dim3(nwg), dim3(256), 0, 0, (const double2*)a, big / 32, per, salt, sink); salt += 977; });
  timed("k_gather32<1, 0>", (double)lanes * per * 32, "the same out of 16 MB (resident in L2 / MALL after the first touches)", [&] {
    hipLaunchKernelGGL((k_gather32<1, 0>), dim3(nwg), dim3(256), 0, 0, (const double2*)a, hot / 32, per, salt, sink); salt += 977; });
  timed("k_gather32<2, 32>", (double)lanes * per * 64, "two 32-B records, the second 32 B behind the first (same 64 B)", [&] {
    hipLaunchKernelGGL((k_gather32<2, 32>), dim3(nwg), dim3(256), 0, 0, (const double2*)a, big / 32, per, salt, sink); salt += 977; });
  timed("k_gather32<3, 64>", (double)lanes * per * 64, "two 32-B records, the second 64 B behind the first (other half of the same 128-B line)", [&] {
    hipLaunchKernelGGL((k_gather32<3, 64>), dim3(nwg), dim3(256), 0, 0, (const double2*)a, big / 32, per, salt, sink); salt += 977; });
  timed("k_gather32<4, 128>", (double)lanes * per * 64, "two 32-B records, the second 128 B behind the first (next 128-B line)", [&] {
    hipLaunchKernelGGL((k_gather32<4, 128>), dim3(nwg), dim3(256), 0, 0, (const double2*)a, big / 32, per, salt, sink); salt += 977; });
  const size_t wbytes = 1ull << 30;
  timed("k_store8", (double)wbytes, "8-B stores, coalesced, 1 GiB", [&] { hipLaunchKernelGGL(k_store8, dim3(nwg), dim3(256), 0, 0, a, wbytes / 8); });
  timed("k_store8_far", (double)(wbytes / 64) * 8, "one 8-B store per 64-B line over 1 GiB: 8 useful bytes per line", [&] { hipLaunchKernelGGL(k_store8_far, dim3(nwg), dim3(256), 0, 0, a, wbytes / 64); });
  timed("k_store64", (double)lanes * 8, "whole 64-B records at random places of the big array, 8 lanes x 8 B each", [&] {
    hipLaunchKernelGGL(k_store64, dim3(nwg), dim3(256), 0, 0, a, big / 64, salt); salt += 977; });
  return 0;
}
