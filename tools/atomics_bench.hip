// same-address / same-line atomic throughput (why results leave the workgroups through partial slots):
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/atom tools/atomics_bench.hip && tools/bin/atom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_same(unsigned long long *p, int naddr, int per_block_cells) {
  // every block: per_block_cells atomics, cell i by thread i, to address i % naddr
  if ((int)threadIdx.x < per_block_cells)
    atomicAdd(p + (threadIdx.x % naddr) * 16, 1ull);
}
__global__ void k_wave(unsigned long long *p, int n) {
  // lane 0 of every wave: n atomics to n addresses
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < n; i++) atomicAdd(p + i * 16, 1ull);
}
int main() {
  unsigned long long *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto f, double natom_per_addr) {
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-50s %.1f us  (%.1f ns per atomic per address)\n", name, ms * 1e3, ms * 1e6 / natom_per_addr);
  };
  run("1280 blocks x 256 cells (256 addr)", [&] { k_same<<<1280, 256>>>(d, 256, 256); }, 1280);
  run("256 blocks x 256 cells (256 addr)", [&] { k_same<<<256, 256>>>(d, 256, 256); }, 256);
  run("1280 blocks x 256 threads -> 1 addr", [&] { k_same<<<1280, 256>>>(d, 1, 256); }, 1280 * 4);
  run("4096 waves x 5 addr (256 blk x 1024)", [&] { k_wave<<<256, 1024>>>(d, 5); }, 4096);
  run("4096 waves x 1 addr", [&] { k_wave<<<256, 1024>>>(d, 1); }, 4096);
  run("empty-ish 256 blocks", [&] { k_same<<<256, 256>>>(d, 256, 0); }, 1);
  return 0;
}
