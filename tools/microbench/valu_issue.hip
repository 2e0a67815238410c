// valu_issue.hip -- what one gfx950 SIMD really issues: wave64 instructions per cycle for the integer
// VALU forms the codec kernels are made of, at 1/2/4/8 waves per SIMD, and LDS look-up / atomic rates.
// Build: hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue ; run on the MI355X box.
// Prints one line per (instruction, waves per SIMD): cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITERS 2048
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 64 independent-ish instructions per iteration over 8 accumulators (dependency distance 8)
#define DEF_KERNEL(NAME, ASM)                                                                     \
__global__ void NAME(uint32_t *out, uint64_t *clk, uint32_t seed)                                 \
{ uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u,     \
           a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u, b = seed | 5u;                            \
  const uint64_t t0 = __builtin_amdgcn_s_memtime();                                               \
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                           \
  for (int i = 0; i < ITERS; i++)                                                                 \
    {                                                                                             \
      _Pragma("unroll")                                                                           \
      for (int k = 0; k < 8; k++)                                                                 \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
    }                                                                                             \
  const uint64_t t1 = __builtin_amdgcn_s_memtime();                                               \
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                           \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                \
}

#define A_ADD(n)    "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_ALIGN(n)  "v_alignbit_b32 %" #n ", %" #n ", %8, %8\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %8\n"
#define A_ANDOR(n)  "v_and_or_b32 %" #n ", %" #n ", %8, %8\n"
#define A_BFE(n)    "v_bfe_u32 %" #n ", %" #n ", 3, 8\n"
#define A_PERM(n)   "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
#define A_MUL24(n)  "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_MULLO(n)  "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_DPP(n)    "v_add_u32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_SDWA(n)   "v_lshlrev_b32_sdwa %" #n ", 2, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define A_XOR3(n)   "v_xor3_b32 %" #n ", %" #n ", %8, %8\n"
#define A_ADD3(n)   "v_add3_u32 %" #n ", %" #n ", %8, %8\n"
#define A_BFI(n)    "v_bfi_b32 %" #n ", %8, %" #n ", %8\n"
#define A_BCNT(n)   "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define A_FFBH(n)   "v_ffbh_u32 %" #n ", %" #n "\n"
#define A_CNDM(n)   "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_AND(n)    "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_OR(n)     "v_or_b32 %" #n ", %" #n ", %8\n"
#define A_SHL(n)    "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define A_SHR(n)    "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_MIN(n)    "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_SUB(n)    "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_CMP(n)    "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define A_CNDS(n)   "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define A_CMPCND(n) "v_cmp_lt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_ADDLSH(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 2\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"

DEF_KERNEL(k_add, A_ADD)
DEF_KERNEL(k_alignbit, A_ALIGN)
DEF_KERNEL(k_lshl_or, A_LSHLOR)
DEF_KERNEL(k_and_or, A_ANDOR)
DEF_KERNEL(k_bfe, A_BFE)
DEF_KERNEL(k_perm, A_PERM)
DEF_KERNEL(k_mul24, A_MUL24)
DEF_KERNEL(k_mullo, A_MULLO)
DEF_KERNEL(k_dpp_add, A_DPP)
DEF_KERNEL(k_sdwa_shl, A_SDWA)

DEF_KERNEL(k_add3, A_ADD3)
DEF_KERNEL(k_bfi, A_BFI)
DEF_KERNEL(k_bcnt, A_BCNT)
DEF_KERNEL(k_ffbh, A_FFBH)
DEF_KERNEL(k_cndmask, A_CNDM)
DEF_KERNEL(k_add_lshl, A_ADDLSH)
DEF_KERNEL(k_and, A_AND)
DEF_KERNEL(k_or, A_OR)
DEF_KERNEL(k_shl, A_SHL)
DEF_KERNEL(k_shr, A_SHR)
DEF_KERNEL(k_min, A_MIN)
DEF_KERNEL(k_sub, A_SUB)
DEF_KERNEL(k_cmp, A_CMP)
DEF_KERNEL(k_cnds, A_CNDS)
DEF_KERNEL(k_cmpcnd, A_CMPCND)
DEF_KERNEL(k_lshl_add, A_LSHLADD)

// 64-bit shift: 4 register pairs
__global__ void k_lshr64(uint32_t *out, uint64_t *clk, uint32_t seed)
{ uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint32_t b = seed | 1u;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ITERS; i++)
    {
      #pragma unroll
      for (int k = 0; k < 8; k++)
        asm volatile("v_lshrrev_b64 %0, %8, %0\nv_lshrrev_b64 %1, %8, %1\nv_lshrrev_b64 %2, %8, %2\nv_lshrrev_b64 %3, %8, %3\n"
                     "v_lshrrev_b64 %4, %8, %4\nv_lshrrev_b64 %5, %8, %5\nv_lshrrev_b64 %6, %8, %6\nv_lshrrev_b64 %7, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) (a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// LDS: table look-ups (ds_read_b32, 256-entry table, index pattern chosen by `mode`) and ds_or_b32
// mode 0: 21 distinct consecutive dwords (ins-like, conflict-free), 1: 61 distinct (mrg-like), 2: all 256 random
template <int ATOMIC>
__global__ void k_lds(uint32_t *out, uint64_t *clk, uint32_t seed, int mode)
{ __shared__ uint32_t tab[4][1024];
  const int w = threadIdx.x >> 6;
  for (int k = threadIdx.x & 63; k < 1024; k += 64) tab[w & 3][k] = k * 2654435761u;
  __syncthreads();
  uint32_t x = (threadIdx.x * 2654435761u + seed) >> 7, acc = 0;
  const uint32_t range = mode == 0 ? 21u : (mode == 1 ? 61u : 256u);
  uint32_t idx[8];
  for (int k = 0; k < 8; k++) { x = x * 1664525u + 1013904223u; idx[k] = ((x >> 8) % range) * 4u; }
  uint32_t *t = tab[w & 3];
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ITERS; i++)
    {
      #pragma unroll
      for (int k = 0; k < 8; k++)
        { if (ATOMIC)
            __hip_atomic_fetch_or((uint32_t *) ((char *) t + idx[k]), 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else
            acc += *(volatile uint32_t *) ((char *) t + idx[k]);
        }
    }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + t[threadIdx.x & 255];
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t);

int main()
{ int dev = 0;
  hipDeviceProp_t p;
  CHECK(hipSetDevice(dev));
  CHECK(hipGetDeviceProperties(&p, dev));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
  uint32_t *out; uint64_t *clk;
  CHECK(hipMalloc((void **) &out, (size_t) cus * 8 * 256 * 4 * 4));
  CHECK(hipMalloc((void **) &clk, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct { const char *name; kern_t k; } ks[] = {
    { "v_add_u32", k_add }, { "v_alignbit_b32", k_alignbit }, { "v_lshl_or_b32", k_lshl_or }, { "v_and_or_b32", k_and_or },
    { "v_bfe_u32", k_bfe }, { "v_perm_b32", k_perm }, { "v_mul_u32_u24", k_mul24 }, { "v_mul_lo_u32", k_mullo },
    { "v_add_u32_dpp row_shr:1", k_dpp_add }, { "v_lshlrev_b32_sdwa", k_sdwa_shl }, { "v_add3_u32", k_add3 },
    { "v_bfi_b32", k_bfi }, { "v_bcnt_u32_b32", k_bcnt }, { "v_ffbh_u32", k_ffbh }, { "v_cndmask_b32", k_cndmask },
    { "v_add_lshl_u32", k_add_lshl }, { "v_and_b32", k_and }, { "v_or_b32", k_or }, { "v_lshlrev_b32", k_shl }, { "v_lshrrev_b32", k_shr },
    { "v_min_u32", k_min }, { "v_sub_u32", k_sub }, { "v_cmp_lt_u32 (vcc)", k_cmp }, { "v_cndmask_b32_e64 (sgpr pair)", k_cnds }, { "v_cmp + v_cndmask (pair = 2)", k_cmpcnd }, { "v_lshl_add_u32", k_lshl_add }, { "v_lshrrev_b64", k_lshr64 } };
  const int wps[] = { 1, 2, 4, 8 };
  for (auto &kk : ks)
    for (int w : wps)
      { // w waves per SIMD: one block of 256 threads puts one wave on each SIMD; w blocks per CU
        const int grid = cus * w;
        hipLaunchKernelGGL(kk.k, dim3(grid), dim3(256), 0, 0, out, clk, 1u);   // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kk.k, dim3(grid), dim3(256), 0, 0, out, clk, 2u);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t c[2]; CHECK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
        const double instr = (double) ITERS * 64.0;                 // per wave
        const double ghz = (double) c[0] / ((double) c[1] * 10.0);  // memrealtime ticks at 100 MHz
        printf("%-28s waves/SIMD %d: %7.3f ms  in-kernel %6.2f cycles per wave-instr per wave; per SIMD %5.2f cycles/instr  (clock %.2f GHz)\n",
               kk.name, w, ms, (double) c[0] / instr, (double) c[0] / (instr * w), ghz);
      }
  for (int atomic = 0; atomic < 2; atomic++)
    for (int mode = 0; mode < 3; mode++)
      for (int w : wps)
        { const int grid = cus * w;
          if (atomic) hipLaunchKernelGGL(k_lds<1>, dim3(grid), dim3(256), 0, 0, out, clk, 1u, mode);
          else        hipLaunchKernelGGL(k_lds<0>, dim3(grid), dim3(256), 0, 0, out, clk, 1u, mode);
          CHECK(hipDeviceSynchronize());
          if (atomic) hipLaunchKernelGGL(k_lds<1>, dim3(grid), dim3(256), 0, 0, out, clk, 2u, mode);
          else        hipLaunchKernelGGL(k_lds<0>, dim3(grid), dim3(256), 0, 0, out, clk, 2u, mode);
          CHECK(hipDeviceSynchronize());
          uint64_t c[2]; CHECK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
          const double instr = (double) ITERS * 8.0;
          printf("%-14s mode %d (%3d distinct) waves/SIMD %d: %6.2f cycles per wave-instr per wave; per CU %5.2f cycles/instr\n",
                 atomic ? "ds_or_b32" : "ds_read_b32", mode, mode == 0 ? 21 : (mode == 1 ? 61 : 256), w,
                 (double) c[0] / instr, (double) c[0] / (instr * w * 4));
        }
  return 0;
}
