/*
 * calib.hip -- calibration of the units the probe kernels are priced against
 * (gfx950): cycles per wave64 VALU instruction at 1 / 2 / 4 / 8 waves per SIMD for
 * several instruction classes (the integer VOP3 mix of the probe kernels, plain
 * VOP2 v_xor_b32, v_fma_f32, v_pk_fma_f32, the v_alignbit/v_and mix of row_bits),
 * latency of a dependent ds_read_b64 chain, and throughput of conflict-free vs
 * random-address ds_read_b64 / ds_read_b128.
 *
 *   hipcc -O3 --offload-arch=gfx950 -o tools/calib tools/calib.hip && tools/calib
 *
 * Prints one JSON object (committed as profiles/rNN/calibration.json).
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

/* 8 independent chains of v_xor / v_and_or: 64 VALU instructions per iteration */
__global__ void __launch_bounds__(256) valu_kernel(uint32_t *out, uint64_t *cyc, int iters)
{
  uint32_t a[8];
  for (int k = 0; k < 8; k++)
    a[k] = threadIdx.x * 2654435761u + k;
  const uint32_t m = out[0], c = out[1];
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        a[k] ^= m;
        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(m), "v"(c));
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t x = 0;
  for (int k = 0; k < 8; k++)
    x ^= a[k];
  out[2 + blockIdx.x * 256 + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}


/* Instruction classes of the VALU stream (round 3: MI355X_MICROARCH.md lists
   v_fma_f32 at 2 cycles per wave64 with >= 2 waves per SIMD; which classes issue at
   which rate on this chip is measured here, not assumed).  16 independent registers,
   every instruction depends on the one 16 instructions earlier: no wave ever waits for
   its own result.  64 VALU instructions per iteration in every class.  Occupancy is
   pinned by the dynamic LDS size of the launch (W workgroups of 4 waves per CU = W
   waves per SIMD) and the grid holds 8 x as many workgroups as fit at once, so that
   an uneven first placement evens out. */
enum { C_XOR = 0, C_AND_OR, C_ALIGNBIT, C_FMA, C_PK_FMA, C_MUL_HI, C_MUL_U24, C_LSHL64, C_CNDMASK,
       C_MAD_U24, C_BFE, C_MIX_XOR_ANDOR, C_MIX_ROWBITS, C_ADD3, C_LSHR, C_MOV, C_CMP_CNDMASK, C_BITOP3,
       C_LSHL_ADD, C_MUL_U24_SDWA, C_COUNT };
static const char *const class_names[C_COUNT] = {
    "v_xor_b32 (VOP2)", "v_and_or_b32 (VOP3)", "v_alignbit_b32 (VOP3)", "v_fma_f32 (VOP3)", "v_pk_fma_f32",
    "v_mul_hi_u32", "v_mul_u32_u24 (VOP2)", "v_lshlrev_b64", "v_cndmask_b32 (VOP2, vcc)", "v_mad_u32_u24 (VOP3)",
    "v_bfe_u32 (VOP3)", "mix: v_xor_b32 + v_and_or_b32", "mix: 8 v_alignbit_b32 + 7 v_and_b32 + v_xor_b32 (row_bits)",
    "v_add3_u32 (VOP3)", "v_lshrrev_b32 (VOP2)", "v_mov_b32 (VOP1)",
    "pair: v_cmp_lt_u32_e64 + v_cndmask_b32_e64 (SGPR mask)", "v_bitop3_b32 (VOP3)", "v_lshl_add_u32 (VOP3)",
    "v_mul_u32_u24_sdwa (WORD_1)"};

template <int CLS>
__global__ void __launch_bounds__(256) valu_class_kernel(uint32_t *out, uint64_t *cyc, int iters)
{
  extern __shared__ uint32_t pin[];            /* (only pins the occupancy) */
  uint32_t a[16];
  float f[16];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 g[16];
  uint64_t w[16];
  for (int k = 0; k < 16; k++) {
    a[k] = threadIdx.x * 2654435761u + k;
    f[k] = (float)(threadIdx.x + k) * 1e-3f;
    g[k] = f2{f[k], f[k] + 1.0f};
    w[k] = ((uint64_t)a[k] << 32) | (a[k] * 40503u);
  }
  const uint32_t m = out[0], c = out[1];
  const float fm = 1.0f + (float)(m & 1u) * 1e-7f, fc = (float)(c & 1u) * 1e-7f;
  const f2 gm = f2{fm, fm}, gc = f2{fc, fc};
  const uint32_t sh = (m & 3u) + 1u;
  if (threadIdx.x == 0xffffu)
    pin[0] = m;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (CLS == C_XOR)
          asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(a[k]) : "v"(m), "v"(a[k]));
        else if (CLS == C_AND_OR)
          asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(m), "v"(c));
        else if (CLS == C_ALIGNBIT)
          asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(a[k]) : "v"(a[k]), "v"(m));
        else if (CLS == C_FMA)
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[k]) : "v"(f[k]), "v"(fm), "v"(fc));
        else if (CLS == C_PK_FMA)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(g[k]) : "v"(g[k]), "v"(gm), "v"(gc));
        else if (CLS == C_MUL_HI)
          asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a[k]) : "v"(a[k]), "v"(m));
        else if (CLS == C_MUL_U24)
          asm volatile("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(a[k]) : "v"(m), "v"(a[k]));
        else if (CLS == C_LSHL64)
          asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(w[k]) : "v"(sh), "v"(w[k]));
        else if (CLS == C_CNDMASK)
          asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(m), "v"(a[k]) : );
        else if (CLS == C_MAD_U24)
          asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(m), "v"(c));
        else if (CLS == C_BFE)
          asm volatile("v_bfe_u32 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(sh), "v"(c));
        else if (CLS == C_ADD3)
          asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(m), "v"(c));
        else if (CLS == C_LSHR)
          asm volatile("v_lshrrev_b32_e32 %0, 1, %1" : "=v"(a[k]) : "v"(a[k]));
        else if (CLS == C_MOV)
          asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[k]) : "v"(a[(k + 1) & 15]));
        else if (CLS == C_CMP_CNDMASK) {
          unsigned long long sm;
          if (k & 1)
            asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(sm) : "v"(a[k]), "v"(m));
          else
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(m), "s"((unsigned long long)c));
          if (k & 1)
            asm volatile("" :: "s"(sm));
        } else if (CLS == C_BITOP3)
          asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x80" : "=v"(a[k]) : "v"(a[k]), "v"(m), "v"(c));
        else if (CLS == C_LSHL_ADD)
          asm volatile("v_lshl_add_u32 %0, %1, 5, %2" : "=v"(a[k]) : "v"(a[k]), "v"(m));
        else if (CLS == C_MUL_U24_SDWA)
          asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD "
                       "src1_sel:WORD_1" : "=v"(a[k]) : "v"(m), "v"(a[k]));
        else if (CLS == C_MIX_XOR_ANDOR) {
          if (k & 1)
            asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(a[k]) : "v"(m), "v"(a[k]));
          else
            asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(m), "v"(c));
        } else {                              /* C_MIX_ROWBITS */
          if (k < 8)
            asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(a[k]) : "v"(a[k]), "v"(m));
          else if (k < 15)
            asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(a[k]) : "v"(c), "v"(a[k]));
          else
            asm volatile("v_xor_b32_e32 %0, %1, %2" : "=v"(a[k]) : "v"(m), "v"(a[k]));
        }
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t x = 0;
  for (int k = 0; k < 16; k++)
    x ^= a[k] ^ __float_as_uint(f[k]) ^ __float_as_uint(g[k].x) ^ __float_as_uint(g[k].y) ^ (uint32_t)w[k] ^
         (uint32_t)(w[k] >> 32);
  out[2 + (blockIdx.x % 2048) * 256 + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0)
    atomicAdd((unsigned long long *)(cyc + (threadIdx.x / 64)), (unsigned long long)(t1 - t0));
}

template <int CLS>
static void launch_class(int grid, size_t lds, uint32_t *out, uint64_t *cyc, int iters)
{
  static bool attr = false;
  if (!attr) {
    CHECK(hipFuncSetAttribute((const void *)valu_class_kernel<CLS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(valu_class_kernel<CLS>, dim3(grid), dim3(256), lds, 0, out, cyc, iters);
}

/* one wave: idx = lds[idx], a dependent chain of ds_read_b64 */
__global__ void __launch_bounds__(64) lds_chain_kernel(uint32_t *out, uint64_t *cyc, int iters)
{
  __shared__ uint64_t tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 64)
    tab[i] = (uint64_t)((i * 1237 + 71) & 4095);
  __syncthreads();
  uint64_t idx = threadIdx.x;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++)
    idx = tab[idx & 4095];
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)idx;
  if (threadIdx.x == 0)
    cyc[blockIdx.x] = t1 - t0;
}

/* every wave: 8 reads in flight per lane, addresses conflict-free (MODE 0) or
   pseudo-random (MODE 1); WIDTH 8 or 16 bytes; 32 KiB table */
template <int WIDTH, int MODE>
__global__ void __launch_bounds__(256) lds_tput_kernel(uint32_t *out, uint64_t *cyc, int iters)
{
  __shared__ __align__(16) unsigned char tab[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += 256)
    ((uint32_t *)tab)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t s = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
  uint32_t acc = 0;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    uint32_t addr[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (MODE == 0) {
        addr[k] = ((lane * WIDTH) + (uint32_t)(i * 8 + k) * 1024u) & 32767u;
      } else {
        s = s * 1664525u + 1013904223u;
        addr[k] = (s >> 12) & (32768u - WIDTH);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (WIDTH == 8) {
        const uint64_t v = *(const uint64_t *)(tab + addr[k]);
        acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
      } else {
        const uint4 v = *(const uint4 *)(tab + addr[k]);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

static double mean(const std::vector<uint64_t> &v, size_t n)
{
  double s = 0;
  for (size_t i = 0; i < n; i++)
    s += (double)v[i];
  return s / (double)n;
}

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t *d_out;
  uint64_t *d_cyc;
  const size_t max_blocks = (size_t)cus * 8;      /* up to 8 waves per SIMD */
  CHECK(hipMalloc(&d_out, (max_blocks * 256 + 16) * sizeof(uint32_t)));
  CHECK(hipMalloc(&d_cyc, max_blocks * 4 * sizeof(uint64_t)));
  CHECK(hipMemset(d_out, 0x5a, (max_blocks * 256 + 16) * sizeof(uint32_t)));
  std::vector<uint64_t> cyc(max_blocks * 4);
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.gcnArchName, cus,
         prop.clockRate / 1000);

  /* two clocks per figure: the wall time of the launch x the nominal shader clock,
     and the s_memtime ticks a wave counted around its loop (shader cycles) */
  auto timed = [&](auto launch) -> double {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();                                  /* warm-up */
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  const double ghz = prop.clockRate / 1e6;

  /* VALU: W blocks of 4 waves per CU = W waves per SIMD */
  printf(", \"valu\": [");
  for (int W = 1; W <= 4; W *= 2) {
    const int iters = 20000;
    const int grid = cus * W;
    const double ms = timed([&] { hipLaunchKernelGGL(valu_kernel, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, iters); });
    CHECK(hipMemcpy(cyc.data(), d_cyc, (size_t)grid * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    const double instr = (double)iters * 64.0;           /* per wave */
    printf("%s{\"waves_per_simd\": %d, \"ms\": %.4f, \"memtime_ticks_per_wave\": %.0f, "
           "\"cycles_per_instr_per_simd_at_nominal_clock\": %.3f}",
           W == 1 ? "" : ", ", W, ms, mean(cyc, (size_t)grid * 4),
           ms * 1e-3 * ghz * 1e9 / (instr * W));
  }
  printf("]");


  /* VALU by instruction class, 1 / 2 / 4 / 8 waves per SIMD.  Two clocks per figure:
     wall time x nominal shader clock, and the s_memtime ticks the waves counted
     (= shader cycles at the clock the chip actually held). */
  {
    printf(", \"valu_classes\": [");
    for (int cls = 0; cls < C_COUNT; cls++) {
      printf("%s{\"class\": \"%s\", \"waves_per_simd\": {", cls ? ", " : "", class_names[cls]);
      for (int W = 1; W <= 8; W *= 2) {
        const int rounds = 8;                       /* workgroups per resident slot */
        const int iters = 4000;
        const int grid = cus * W * rounds;
        const size_t lds = (size_t)(152 * 1024) / W;   /* W workgroups per CU fit, W + 1 do not */
        auto launch = [&] {
          CHECK(hipMemsetAsync(d_cyc, 0, 4 * sizeof(uint64_t), 0));
          switch (cls) {
#define CASE(C) case C: launch_class<C>(grid, lds, d_out, d_cyc, iters); break;
          CASE(C_XOR) CASE(C_AND_OR) CASE(C_ALIGNBIT) CASE(C_FMA) CASE(C_PK_FMA) CASE(C_MUL_HI) CASE(C_MUL_U24)
          CASE(C_LSHL64) CASE(C_CNDMASK) CASE(C_MAD_U24) CASE(C_BFE) CASE(C_MIX_XOR_ANDOR) CASE(C_MIX_ROWBITS)
          CASE(C_ADD3) CASE(C_LSHR) CASE(C_MOV) CASE(C_CMP_CNDMASK) CASE(C_BITOP3) CASE(C_LSHL_ADD) CASE(C_MUL_U24_SDWA)
#undef CASE
          }
        };
        const double ms = timed(launch);
        CHECK(hipMemcpy(cyc.data(), d_cyc, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        const double instr_per_simd = (double)iters * 64.0 * W * rounds;   /* W waves at a time, `rounds` of them */
        double ticks = 0;
        for (int q = 0; q < 4; q++)
          ticks += (double)cyc[q];
        ticks /= (double)grid * 4;                   /* per wave */
        printf("%s\"%d\": {\"cycles_per_instr_at_nominal_clock\": %.3f, \"memtime_ticks_per_instr\": %.3f, \"ms\": %.3f}",
               W == 1 ? "" : ", ", W, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ticks / ((double)iters * 64.0 * W), ms);
      }
      printf("}}");
    }
    printf("]");
  }

  /* dependent LDS chain */
  {
    const int iters = 200000;
    const double ms = timed([&] { hipLaunchKernelGGL(lds_chain_kernel, dim3(1), dim3(64), 0, 0, d_out, d_cyc, iters); });
    CHECK(hipMemcpy(cyc.data(), d_cyc, sizeof(uint64_t), hipMemcpyDeviceToHost));
    printf(", \"lds_dependent_read_b64\": {\"ms\": %.4f, \"cycles_per_read_at_nominal_clock\": %.1f, "
           "\"memtime_ticks_per_read\": %.1f}",
           ms, ms * 1e-3 * ghz * 1e9 / iters, (double)cyc[0] / iters);
  }

  /* LDS throughput, 4 blocks of 4 waves per CU */
  printf(", \"lds_throughput\": [");
  const int iters = 4000;
  const int grid = cus * 4;
  const char *names[4] = {"b64_conflict_free", "b64_random", "b128_conflict_free", "b128_random"};
  for (int k = 0; k < 4; k++) {
    double ms = 0;
    if (k == 0) ms = timed([&] { hipLaunchKernelGGL((lds_tput_kernel<8, 0>), dim3(grid), dim3(256), 0, 0, d_out, d_cyc, iters); });
    if (k == 1) ms = timed([&] { hipLaunchKernelGGL((lds_tput_kernel<8, 1>), dim3(grid), dim3(256), 0, 0, d_out, d_cyc, iters); });
    if (k == 2) ms = timed([&] { hipLaunchKernelGGL((lds_tput_kernel<16, 0>), dim3(grid), dim3(256), 0, 0, d_out, d_cyc, iters); });
    if (k == 3) ms = timed([&] { hipLaunchKernelGGL((lds_tput_kernel<16, 1>), dim3(grid), dim3(256), 0, 0, d_out, d_cyc, iters); });
    const double reads_per_cu = (double)iters * 8.0 * 16.0;      /* wave-instructions per CU */
    CHECK(hipMemcpy(cyc.data(), d_cyc, (size_t)grid * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    printf("%s{\"access\": \"%s\", \"ms\": %.4f, \"cycles_per_wave_read_per_cu_at_nominal_clock\": %.2f, "
           "\"memtime_ticks_per_wave_read_per_cu\": %.2f}",
           k ? ", " : "", names[k], ms, ms * 1e-3 * ghz * 1e9 / reads_per_cu,
           mean(cyc, (size_t)grid * 4) / reads_per_cu);
  }
  printf("]}\n");
  return 0;
}
