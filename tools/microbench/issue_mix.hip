// issue_mix.hip -- how a gfx950 SIMD shares its issue slots between instruction kinds: a loop of 32 VALU instructions alone,
// with 32 SALU instructions / 32 s_nop / 8 LDS atomics interleaved, and the SALU instructions alone, at 1..8 co-resident waves
// per SIMD (ONE workgroup of 256 * W threads per CU, so the waves are co-resident by construction).
// Build: hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix ; prints cycles per loop iteration per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4096
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define V4 "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n"
#define S4 "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\ns_add_u32 s22, s22, 1\ns_add_u32 s23, s23, 1\n"
#define N4 "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n"
#define VS4 "v_add_u32 %0, %0, %4\ns_add_u32 s20, s20, 1\nv_add_u32 %1, %1, %4\ns_add_u32 s21, s21, 1\nv_add_u32 %2, %2, %4\ns_add_u32 s22, s22, 1\nv_add_u32 %3, %3, %4\ns_add_u32 s23, s23, 1\n"
#define VN4 "v_add_u32 %0, %0, %4\ns_nop 0\nv_add_u32 %1, %1, %4\ns_nop 0\nv_add_u32 %2, %2, %4\ns_nop 0\nv_add_u32 %3, %3, %4\ns_nop 0\n"
#define VL4 "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\nds_add_u32 %5, %4\n"
#define R8(X) X X X X X X X X

template <int MODE>
__global__ void k_mix(uint32_t *out, uint64_t *clk, uint32_t seed)
{ __shared__ uint32_t lds[64 * 32];
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b = seed | 5u;
  uint32_t la = (threadIdx.x & 63) * 4u + ((threadIdx.x >> 6) & 31) * 256u;          // conflict-free, a row per wave
  for (int k = threadIdx.x; k < 64 * 32; k += blockDim.x) lds[k] = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; i++)
    { if (MODE == 0) asm volatile(R8(V4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(la) : "s20", "s21", "s22", "s23", "scc");
      if (MODE == 1) asm volatile(R8(VS4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(la) : "s20", "s21", "s22", "s23", "scc");
      if (MODE == 2) asm volatile(R8(S4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(la) : "s20", "s21", "s22", "s23", "scc");
      if (MODE == 3) asm volatile(R8(VN4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(la) : "s20", "s21", "s22", "s23", "scc");
      if (MODE == 4) asm volatile(R8(VL4) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(la) : "s20", "s21", "s22", "s23", "scc", "memory");
    }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ lds[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t);
int main()
{ hipDeviceProp_t p; CHECK(hipSetDevice(0)); CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  uint32_t *out; uint64_t *clk;
  CHECK(hipMalloc((void **) &out, (size_t) cus * 1024 * 4)); CHECK(hipMalloc((void **) &clk, 16));
  struct { const char *name; kern_t k; int valu, other; } ks[] = {
    { "32 VALU", k_mix<0>, 32, 0 }, { "32 VALU + 32 SALU interleaved", k_mix<1>, 32, 32 }, { "32 SALU", k_mix<2>, 0, 32 },
    { "32 VALU + 32 s_nop interleaved", k_mix<3>, 32, 32 }, { "32 VALU + 8 ds_add_u32", k_mix<4>, 32, 8 } };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (auto &kk : ks)
    for (int w = 1; w <= 4; w *= 2)                      // 1024 threads per workgroup at most: 4 waves per SIMD
      { hipLaunchKernelGGL(kk.k, dim3(cus), dim3(256 * w), 0, 0, out, clk, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kk.k, dim3(cus), dim3(256 * w), 0, 0, out, clk, 2u);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t c; CHECK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
        printf("%-32s waves/SIMD %d: %7.3f ms, %7.1f cycles per iteration per wave, %6.1f per SIMD and wave-iteration (%d VALU + %d other)\n",
               kk.name, w, ms, (double) c / ITERS, (double) c / ITERS / w, kk.valu, kk.other);
        fflush(stdout);
      }
  return 0;
}
