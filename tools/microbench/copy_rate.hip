// copy_rate.hip -- what this GPU's memory system gives plain streaming kernels: read-only, write-only and copy, 16 bytes per
// lane and access, hand-written (tools/microbench/hbm_rates.py measures torch's kernels).  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{ const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * U)
    { u32x4 v[U];
      #pragma unroll
      for (int k = 0; k < U; k++) if (i + k * stride < n16) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
      #pragma unroll
      for (int k = 0; k < U; k++) if (i + k * stride < n16) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ src, uint32_t *out, size_t n16)
{ const size_t stride = (size_t) gridDim.x * blockDim.x;
  u32x4 acc = { 0, 0, 0, 0 };
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * U)
    {
      #pragma unroll
      for (int k = 0; k < U; k++) if (i + k * stride < n16) acc ^= src[i + k * stride];
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
template <int U>
__global__ __launch_bounds__(256) void k_fill(u32x4 *__restrict__ dst, size_t n16)
{ const size_t stride = (size_t) gridDim.x * blockDim.x;
  const u32x4 v = { 1, 2, 3, 4 };
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * U)
    {
      #pragma unroll
      for (int k = 0; k < U; k++) if (i + k * stride < n16) dst[i + k * stride] = v;
    }
}
int main()
{ hipDeviceProp_t p; CHECK(hipSetDevice(0)); CHECK(hipGetDeviceProperties(&p, 0));
  const size_t bytes = (size_t) 16 << 30, n16 = bytes / 16;
  u32x4 *a, *b; uint32_t *o;
  CHECK(hipMalloc((void **) &a, bytes)); CHECK(hipMalloc((void **) &b, bytes)); CHECK(hipMalloc((void **) &o, 64));
  CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int cus = p.multiProcessorCount;
#define RUN(NAME, GB, ...) for (int per = 4; per <= 32; per *= 2) { float best = 1e9; for (int rep = 0; rep < 3; rep++) { CHECK(hipEventRecord(e0, 0)); \
      hipLaunchKernelGGL(__VA_ARGS__); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
      printf("%-28s %2d workgroups of 256 per CU: %7.3f ms  %6.2f TB/s\n", NAME, per, best, (GB) / best / 1e9); fflush(stdout); }
  RUN("read 16 GiB (x4)", (double) bytes, k_read<4>, dim3(cus * per), dim3(256), 0, 0, a, o, n16)
  RUN("fill 16 GiB (x4)", (double) bytes, k_fill<4>, dim3(cus * per), dim3(256), 0, 0, b, n16)
  RUN("copy 16 GiB (x1), r + w", 2.0 * bytes, (k_copy<1, false>), dim3(cus * per), dim3(256), 0, 0, a, b, n16)
  RUN("copy 16 GiB (x4), r + w", 2.0 * bytes, (k_copy<4, false>), dim3(cus * per), dim3(256), 0, 0, a, b, n16)
  RUN("copy 16 GiB (x4, nt), r + w", 2.0 * bytes, (k_copy<4, true>), dim3(cus * per), dim3(256), 0, 0, a, b, n16)
  return 0;
}
