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
// ... through 16-byte stores that begin `off` bytes past a 16-byte boundary (what k_pack2_decode's lanes do: a read's text begins anywhere)
typedef u32x4 u32x4_un __attribute__((aligned(1)));
template <int U>
__global__ __launch_bounds__(256) void k_fill_off(uint8_t *__restrict__ dst, size_t n16, unsigned off)
{ const size_t stride = (size_t) gridDim.x * blockDim.x;
  const u32x4 v = { 1, 2, 3, 4 };
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n16; i += stride * U)
    {
      #pragma unroll
      for (int k = 0; k < U; k++) if (i + k * stride + 1 < n16) *(u32x4_un *) (dst + 16 * (i + k * stride) + off) = v;
    }
}
// the plainest float4 copy there is: one element a thread, as many workgroups as there are elements (what a guide's "float4 copy"
// usually is); BS threads a workgroup
template <int BS>
__global__ __launch_bounds__(BS) void k_copy_flat(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{ const size_t i = (size_t) blockIdx.x * BS + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
// ... and with every workgroup on a contiguous tile of its own (a wave's 16 loads in a row, then its 16 stores)
template <int U>
__global__ __launch_bounds__(256) void k_copy_tile(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{ const size_t base = ((size_t) blockIdx.x * U) * 256 + threadIdx.x;
  u32x4 v[U];
  #pragma unroll
  for (int k = 0; k < U; k++) if (base + (size_t) k * 256 < n16) v[k] = src[base + (size_t) k * 256];
  #pragma unroll
  for (int k = 0; k < U; k++) if (base + (size_t) k * 256 < n16) dst[base + (size_t) k * 256] = v[k];
}
// ... the way the library's kernels take their work: resident waves drawing tickets (an atomic on one word) of UNITS KiB each
template <int UNITS>
__global__ __launch_bounds__(256) void k_copy_ticket(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16, unsigned *ticket)
{ const unsigned lane = threadIdx.x & 63u;
  for (;;)
    { unsigned t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      const size_t base = (size_t) t * UNITS * 64u;
      if (base >= n16) break;
      #pragma unroll 1
      for (int k = 0; k < UNITS; k++)
        { const size_t i = base + (size_t) k * 64u + lane;
          if (i < n16) dst[i] = src[i];
        }
    }
}
// ... and the way k_qv_hist reads an entry: a wave per 50 KB "entry", its five 10 KB "lines" a KiB of each per step (read only)
__global__ __launch_bounds__(256) void k_read_entries(const u32x4 *__restrict__ src, uint32_t *out, size_t n16, unsigned *ticket)
{ const unsigned lane = threadIdx.x & 63u;
  const size_t entry16 = 50u * 1024u / 16u, line16 = entry16 / 5u, entries = n16 / entry16;
  u32x4 acc = { 0, 0, 0, 0 };
  for (;;)
    { unsigned t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      if ((size_t) 2 * t >= entries) break;
      for (int e = 0; e < 2; e++)
        { const size_t base = ((size_t) 2 * t + e) * entry16;
          for (size_t st = 0; st < line16; st += 64u)
            {
              #pragma unroll
              for (int l = 0; l < 5; l++)
                if (st + lane < line16) acc ^= src[base + (size_t) l * line16 + st + lane];
            }
        }
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
// the same bytes read front to back by the same resident waves (tickets of 2 x 50 KB)
__global__ __launch_bounds__(256) void k_read_flat_tickets(const u32x4 *__restrict__ src, uint32_t *out, size_t n16, unsigned *ticket)
{ const unsigned lane = threadIdx.x & 63u;
  const size_t unit16 = 2u * 50u * 1024u / 16u;
  u32x4 acc = { 0, 0, 0, 0 };
  for (;;)
    { unsigned t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      const size_t base = (size_t) t * unit16;
      if (base >= n16) break;
      for (size_t st = 0; st < unit16; st += 64u)
        if (base + st + lane < n16) acc ^= src[base + st + lane];
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
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
  RUN("fill 16 GiB (x4), stores 4 bytes off", (double) bytes, k_fill_off<4>, dim3(cus * per), dim3(256), 0, 0, (uint8_t *) b, n16, 4u)
  RUN("fill 16 GiB (x4), stores 5 bytes off", (double) bytes, k_fill_off<4>, dim3(cus * per), dim3(256), 0, 0, (uint8_t *) b, n16, 5u)
  RUN("copy 16 GiB (x1), r + w", 2.0 * bytes, (k_copy<1, false>), dim3(cus * per), dim3(256), 0, 0, a, b, n16)
  RUN("copy 16 GiB (x4), r + w", 2.0 * bytes, (k_copy<4, false>), dim3(cus * per), dim3(256), 0, 0, a, b, n16)
  RUN("copy 16 GiB (x4, nt), r + w", 2.0 * bytes, (k_copy<4, true>), dim3(cus * per), dim3(256), 0, 0, a, b, n16)
  // the guide's figure (MI355X_MICROARCH.md: "6.29 TB/s measured (float4 copy, 79 %)"): the shapes such a copy is usually written in,
  // over 1, 4 and 16 GiB, and the runtime's own device-to-device copy
#define ONE(NAME, GB, ...) { float best = 1e9; for (int rep = 0; rep < 5; rep++) { CHECK(hipEventRecord(e0, 0)); __VA_ARGS__; CHECK(hipEventRecord(e1, 0)); \
      CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
      printf("%-44s %8.3f ms  %6.2f TB/s\n", NAME, best, (GB) / best / 1e9); fflush(stdout); }
  for (int sh = 30; sh <= 34; sh += 2)
    { const size_t by = (size_t) 1 << sh, m16 = by / 16;
      char nm[96];
      snprintf(nm, sizeof(nm), "flat copy %2zu GiB, 256 threads, r + w", by >> 30);
      ONE(nm, 2.0 * by, hipLaunchKernelGGL((k_copy_flat<256>), dim3((unsigned) (m16 / 256)), dim3(256), 0, 0, a, b, m16))
      snprintf(nm, sizeof(nm), "flat copy %2zu GiB, 1024 threads, r + w", by >> 30);
      ONE(nm, 2.0 * by, hipLaunchKernelGGL((k_copy_flat<1024>), dim3((unsigned) (m16 / 1024)), dim3(1024), 0, 0, a, b, m16))
      snprintf(nm, sizeof(nm), "tile copy %2zu GiB, 4 x 16 B a lane, r + w", by >> 30);
      ONE(nm, 2.0 * by, hipLaunchKernelGGL((k_copy_tile<4>), dim3((unsigned) (m16 / 1024)), dim3(256), 0, 0, a, b, m16))
      snprintf(nm, sizeof(nm), "tile copy %2zu GiB, 16 x 16 B a lane, r + w", by >> 30);
      ONE(nm, 2.0 * by, hipLaunchKernelGGL((k_copy_tile<16>), dim3((unsigned) (m16 / 4096)), dim3(256), 0, 0, a, b, m16))
      snprintf(nm, sizeof(nm), "hipMemcpyAsync D2D %2zu GiB, r + w", by >> 30);
      ONE(nm, 2.0 * by, CHECK(hipMemcpyAsync(b, a, by, hipMemcpyDeviceToDevice, 0)))
    }
  { unsigned *tk; CHECK(hipMalloc((void **) &tk, 64));
    const size_t by = (size_t) 16 << 30, m16 = by / 16;
#define TICKETS(NAME, GB, WPC, ...) { float best = 1e9; for (int rep = 0; rep < 3; rep++) { CHECK(hipMemset(tk, 0, 64)); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0, 0)); \
      hipLaunchKernelGGL(__VA_ARGS__); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
      printf("%-60s %8.3f ms  %6.2f TB/s\n", NAME, best, (GB) / best / 1e9); fflush(stdout); }
    TICKETS("ticket copy 16 GiB, 1 KiB a ticket, 16 waves a CU, r + w", 2.0 * by, 16, (k_copy_ticket<1>), dim3(cus * 4), dim3(256), 0, 0, a, b, m16, tk)
    TICKETS("ticket copy 16 GiB, 4 KiB a ticket, 16 waves a CU, r + w", 2.0 * by, 16, (k_copy_ticket<4>), dim3(cus * 4), dim3(256), 0, 0, a, b, m16, tk)
    TICKETS("ticket copy 16 GiB, 16 KiB a ticket, 16 waves a CU, r + w", 2.0 * by, 16, (k_copy_ticket<16>), dim3(cus * 4), dim3(256), 0, 0, a, b, m16, tk)
    TICKETS("ticket copy 16 GiB, 16 KiB a ticket, 32 waves a CU, r + w", 2.0 * by, 32, (k_copy_ticket<16>), dim3(cus * 8), dim3(256), 0, 0, a, b, m16, tk)
    TICKETS("read 16 GiB as entries of 5 lines, 16 waves a CU", (double) by, 16, k_read_entries, dim3(cus * 4), dim3(256), 0, 0, a, o, m16, tk)
    TICKETS("read 16 GiB as entries of 5 lines, 32 waves a CU", (double) by, 32, k_read_entries, dim3(cus * 8), dim3(256), 0, 0, a, o, m16, tk)
    TICKETS("read 16 GiB front to back, 100 KB a ticket, 16 waves a CU", (double) by, 16, k_read_flat_tickets, dim3(cus * 4), dim3(256), 0, 0, a, o, m16, tk)
    TICKETS("read 16 GiB front to back, 100 KB a ticket, 32 waves a CU", (double) by, 32, k_read_flat_tickets, dim3(cus * 8), dim3(256), 0, 0, a, o, m16, tk)
  }
  { int clk = 0, mclk = 0;
    (void) hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); (void) hipDeviceGetAttribute(&mclk, hipDeviceAttributeMemoryClockRate, 0);
    printf("device: %s, %d CUs, clock %d kHz, memory clock %d kHz, bus %d bits\n", p.name, cus, clk, mclk, p.memoryBusWidth);
  }
  return 0;
}
