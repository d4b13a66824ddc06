// latency of dependent random 64-byte reads over buffers of growing size, 1 / 1024 / 5120 waves:
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/chase tools/chase_bench.hip && tools/bin/chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
// one wave, each lane chases its own random cycle through a buffer of `n` 64-byte lines
__global__ void chase(const uint32_t *buf, uint32_t n, int steps, uint32_t *out, unsigned long long *cyc) {
  uint32_t i = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) % n;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; s++) i = buf[(size_t)i * 16];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = i;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  for (size_t mb : {16, 256, 1024, 4096}) {
    uint32_t n = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<uint32_t> h((size_t)n * 16);
    std::mt19937 rng(1);
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; i++) perm[i] = i;
    for (uint32_t i = n - 1; i > 0; i--) { uint32_t j = rng() % (i + 1); std::swap(perm[i], perm[j]); }
    for (uint32_t i = 0; i < n; i++) h[(size_t)perm[i] * 16] = perm[(i + 1) % n];
    uint32_t *d, *out; unsigned long long *cyc;
    if (hipMalloc(&d, h.size() * 4) != hipSuccess) return 1;
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 1 << 16);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {1, 1024, 5120}) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      int steps = 200;
      chase<<<blocks, 64>>>(d, n, steps, out, cyc); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); chase<<<blocks, 64>>>(d, n, steps, out, cyc); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%5zu MB  %5d waves: %.0f ns per dependent 64-lane random load (%.2f G loads/s)\n", mb, blocks, ms * 1e6 / steps,
             (double)blocks * 64 * steps / (ms * 1e-3) / 1e9);
    }
    (void)hipFree(d); (void)hipFree(out); (void)hipFree(cyc);
  }
  return 0;
}
