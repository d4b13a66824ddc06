/* tools/lds_conflicts.hip -- are the LDS bank conflicts of probe_pairs2_kernel (65 % of its LDS
 * cycles, profiles/r0[56]/cfg5_pmc_summary.json) a property of ITS layout, or of 64 lanes reading
 * 32-byte words at uniformly random offsets?  
 *
 * The kernel reads a filter word as two ds_read_b128 at wo and wo + 16, wo = 32 * (hash bits scaled
 * to the slice's word count): kernels_pairs2.h word_lds / woff_of, six words in flight per lane.
 * This program issues exactly that pattern from a slice-sized LDS image, and the alternatives the
 * verdict names, and is run twice: bare (time per wave-read from s_memtime) and under
 *   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
 * (tools/r06_lds_conflicts.sh), so the conflict share of every form is counted by the hardware.
 *
 *   MODE 0  conflict-free: lane l reads 16 bytes at 16 l and 16 l + half the slice -- the floor
 *   MODE 7  consecutive words in the kernel's layout (lane l: word l + rotation): stride 32 bytes
 *   MODE 1  uniformly random word, halves at wo and wo + 16              -- the kernel's form
 *   MODE 2  halves of word w at 16 w and 16 w + half the slice           -- "w and w ^ (slice_words/2)"
 *   MODE 3  MODE 1 with a 5-bit swizzle of the word index (w ^ (w >> 5) & 31)
 *   MODE 4  MODE 1, but the two halves read by lanes in opposite order (odd lanes: high half first)
 *   MODE 5  one ds_read_b128 per word only (a 16-byte word)              -- what half the bytes would cost
 *   MODE 6  the 32 bytes as four ds_read_b64
 * A bijection of a uniformly random index is a uniformly random index: MODE 2/3/4 can only differ
 * from MODE 1 if the hardware pairs the two instructions of a lane, which the counters then show.
 *
 * Build: make tools  (hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_conflicts tools/lds_conflicts.hip) */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint32_t g_slice_bytes = 58368;      /* a slice of 1824 words of 32 bytes (argv[1]: another size) */

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) lds_words_kernel(uint32_t *out, uint64_t *cyc, int iters, const uint32_t SLICE_BYTES)
{
  extern __shared__ __align__(16) unsigned char tab[];
  const uint32_t NWORDS = SLICE_BYTES / 32u;
  for (uint32_t i = threadIdx.x; i < SLICE_BYTES / 4; i += blockDim.x)
    ((uint32_t *)tab)[i] = i * 2654435761u;
  __syncthreads();
  uint64_t s = (uint64_t)(blockIdx.x * 1024u + threadIdx.x + 1u) * 0x9e3779b97f4a7c15ull;
  uint32_t acc = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    uint32_t wo[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      uint32_t w = __umul24((uint32_t)(s >> 48), NWORDS) >> 16;       /* woff_of */
      if (MODE == 0 || MODE == 7)
        w = (lane + (uint32_t)(i * 6 + k) * 64u) % NWORDS;
      if (MODE == 3)
        w = (w & ~31u) | ((w ^ (w >> 5)) & 31u);
      wo[k] = w;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const uint32_t w = wo[k];
      if (MODE == 2 || MODE == 0) {
        const u32x4 a = *(const u32x4 *)(tab + w * 16u);
        const u32x4 b = *(const u32x4 *)(tab + w * 16u + SLICE_BYTES / 2);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
      } else if (MODE == 4) {
        const uint32_t f = (lane & 1u) * 16u;
        const u32x4 a = *(const u32x4 *)(tab + w * 32u + f);
        const u32x4 b = *(const u32x4 *)(tab + w * 32u + (16u - f));
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
      } else if (MODE == 5) {
        const u32x4 a = *(const u32x4 *)(tab + w * 32u);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
      } else if (MODE == 6) {
        const uint64_t *p = (const uint64_t *)(tab + w * 32u);
        const uint64_t a = p[0], b = p[1], c = p[2], d = p[3];
        const uint64_t x = a ^ b ^ c ^ d;
        acc ^= (uint32_t)x ^ (uint32_t)(x >> 32);
      } else {
        const u32x4 a = *(const u32x4 *)(tab + w * 32u);
        const u32x4 b = *(const u32x4 *)(tab + w * 32u + 16u);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 1024u + threadIdx.x] = acc;
  if (lane == 0)
    cyc[blockIdx.x * 16u + threadIdx.x / 64u] = t1 - t0;
}

template <int MODE>
static int run(const char *what, int cus, uint32_t *d_out, uint64_t *d_cyc, bool first)
{
  const int iters = 4000, grid = cus * 2;          /* two workgroups of 16 waves per CU: 8 waves per SIMD */
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void *)lds_words_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_slice_bytes));
  hipLaunchKernelGGL(lds_words_kernel<MODE>, dim3(grid), dim3(1024), g_slice_bytes, 0, d_out, d_cyc, iters, g_slice_bytes);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(lds_words_kernel<MODE>, dim3(grid), dim3(1024), g_slice_bytes, 0, d_out, d_cyc, iters, g_slice_bytes);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint64_t> cyc((size_t)grid * 16);
  CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  double ticks = 0;
  for (uint64_t c : cyc)
    ticks += (double)c;
  ticks /= (double)cyc.size();
  /* words per CU = 32 waves x iters x 6; a CU's LDS serves them one after the other */
  const double words_per_cu = 32.0 * iters * 6.0;
  printf("%s{\"mode\": %d, \"what\": \"%s\", \"ms\": %.4f, \"memtime_ticks_per_word_per_cu\": %.2f, "
         "\"ns_per_word_per_cu\": %.3f}",
         first ? "" : ",\n ", MODE, what, ms, ticks / words_per_cu, ms * 1e6 / words_per_cu);
  return 0;
}

int main(int argc, char **argv)
{
  if (argc > 1)
    g_slice_bytes = (uint32_t)atoi(argv[1]) / 64u * 64u;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t *d_out;
  uint64_t *d_cyc;
  CHECK(hipMalloc(&d_out, (size_t)cus * 2 * 1024 * sizeof(uint32_t)));
  CHECK(hipMalloc(&d_cyc, (size_t)cus * 2 * 16 * sizeof(uint64_t)));
  printf("{\"device\": \"%s\", \"cus\": %d, \"slice_bytes\": %u, \"forms\": [\n ", prop.gcnArchName, cus, g_slice_bytes);
  if (run<0>("conflict-free: 16 l and 16 l + slice/2 (2 x b128)", cus, d_out, d_cyc, true)) return 1;
  if (run<7>("consecutive 32-byte words, halves at wo, wo+16", cus, d_out, d_cyc, false)) return 1;
  if (run<1>("uniformly random word, halves at wo, wo+16 (the kernel)", cus, d_out, d_cyc, false)) return 1;
  if (run<2>("halves at 16w and 16w + slice/2", cus, d_out, d_cyc, false)) return 1;
  if (run<3>("5-bit swizzle of the word index", cus, d_out, d_cyc, false)) return 1;
  if (run<4>("odd lanes read the high half first", cus, d_out, d_cyc, false)) return 1;
  if (run<5>("one b128 per word (16-byte words)", cus, d_out, d_cyc, false)) return 1;
  if (run<6>("four b64 per word", cus, d_out, d_cyc, false)) return 1;
  printf("]}\n");
  return 0;
}
