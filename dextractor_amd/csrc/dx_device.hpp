// dx_device.hpp -- wave-level device helpers for gfx950 (wave64).
//
// All kernels in libdexgpu give one 64-lane wavefront one unit of work (a read, a .quiva entry)
// and step over its byte streams 1 KiB at a time: 16 consecutive bytes per lane, loaded with a
// single (possibly unaligned) global_load_dwordx4.  gfx950 runs with unaligned access mode
// enabled, so byte-aligned 16-byte loads and 4-byte stores are single instructions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4    u32x4_u __attribute__((aligned(1)));
typedef uint32_t u32_u   __attribute__((aligned(1)));
typedef uint64_t u64_u   __attribute__((aligned(1)));

#define DX_STEP 1024            // bytes of one stream a wave consumes per step (16 per lane)

__device__ __forceinline__ int      lane_id()  { return (int) (threadIdx.x & 63); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{ uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (v >> 32));
  return ((uint64_t) hi << 32) | lo;
}

// Work distribution: the waves of a kernel draw their units (entries, reads) from a counter, so
// units of very different length spread evenly.  The counter only grows, so every wave sees a
// value >= n and leaves; the host zeroes it before the launch.
__device__ __forceinline__ uint64_t next_unit(uint32_t *counter, uint32_t batch = 1u)
{ uint32_t t = 0;
  if (lane_id() == 0)
    t = atomicAdd(counter, batch);
  return (uint64_t) uniform(t);
}

// Units per ticket where only the device knows how many bytes a batch holds: one thread, in front of the kernel, leaves
// the number in the word BEHIND the kernel's ticket counter (ticket[1]; the host zeroes only ticket[0]) -- about `target`
// bytes' worth per ticket, `least` units at least.  bytes = *hi + *extra - *lo.  (Every draw is an atomic on one address,
// ~11 ns each chip-wide: a fixed few units per ticket bind a batch of many short units to its counter -- 50 M reads of
// 1 kb, 16 per ticket: 34 ms of draws for 13 ms of work.)
static __global__ void k_ticket_units(const uint64_t *lo, const uint64_t *hi, const uint32_t *extra, uint64_t n,
                                      uint32_t target, uint32_t least, uint32_t *ticket)
{ const uint64_t bytes = *hi + (extra ? (uint64_t) *extra : 0ull) - *lo;
  const uint64_t mean  = n ? bytes / n + 1u : 1u;
  uint64_t u = ((uint64_t) target + mean - 1u) / mean;
  ticket[1] = (uint32_t) (u < least ? least : (u > 4096u ? 4096u : u));
}

// what the kernel makes of that word (never less than `least`, whatever is there)
__device__ __forceinline__ uint32_t ticket_units_of(const uint32_t *ticket, uint32_t least)
{ const uint32_t u = uniform(ticket[1]);
  return u < least || u > 4096u ? least : u;
}

// Orders this wave's LDS traffic: everything before is complete and visible to the other lanes
// of the wave before anything after starts.  (LDS instructions of one wave execute in order;
// this pins the compiler and waits for outstanding returns.)
__device__ __forceinline__ void wave_sync()
{ __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum over the 64 lanes, in registers (DPP row shifts + row broadcasts).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{ v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// inclusive running maximum over the 64 lanes (values >= 0)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{ v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true));
  v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true));
  v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true));
  v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true));
  v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false));
  v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false));
  return v;
}

__device__ __forceinline__ uint32_t wave_total(uint32_t incl)
{ return __builtin_amdgcn_readlane(incl, 63); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{ return wave_total(wave_incl_scan(v)); }

// 16 bytes of a stream for this lane: bytes [0,valid) are real, the rest read as zero.  A full
// chunk is one unaligned 16-byte load; a partial one (only in the last step of a stream) is
// assembled from byte loads so that nothing past the stream's end is ever touched.
__device__ __forceinline__ u32x4 load_chunk(const uint8_t *p, int valid)
{ u32x4 v = { 0u, 0u, 0u, 0u };
  if (valid >= 16)
    v = *(const u32x4_u *) p;
  else if (valid > 0)
    { uint32_t w[4] = { 0u, 0u, 0u, 0u };
      for (int b = 0; b < valid; b++)
        w[b >> 2] |= (uint32_t) p[b] << (8 * (b & 3));
      v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    }
  return v;
}

__device__ __forceinline__ uint32_t chunk_word(const u32x4 &v, int i)
{ return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// byte b (0..15, dynamic) of a chunk
__device__ __forceinline__ uint32_t chunk_byte(const u32x4 &v, int b)
{ uint32_t w = (b & 8) ? ((b & 4) ? v.w : v.z) : ((b & 4) ? v.y : v.x);
  return (w >> (8 * (b & 3))) & 0xffu;
}

// 16-bit mask: bit b set iff byte b of the chunk equals c
__device__ __forceinline__ uint32_t chunk_eq_mask(const u32x4 &v, uint32_t c)
{ uint32_t m = 0;
  #pragma unroll
  for (int i = 3; i >= 0; i--)
    { uint32_t w = chunk_word(v, i) ^ (c * 0x01010101u);
      // zero-byte detector: exact per byte (no cross-byte borrow)
      uint32_t z = ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu);   // 0x80 where byte == 0
      // flags of bytes 0..2 sit at bits 7, 15, 23: a 24-bit multiply by 1 + 2^7 + 2^14 lines them
      // up at bits 14..16 (all partial products fall on distinct bits, so nothing carries)
      const uint32_t low3 = (__umul24(z >> 7, 0x4081u) >> 14) & 7u;
      m = (m << 4) | ((z >> 31) << 3) | low3;
    }
  return m;
}

__device__ __forceinline__ void store32_u(uint8_t *p, uint32_t v) { *(u32_u *) p = v; }

// group index of the plain QV lines (dx_qv.hip writes it, dx_qv_decode.hip reads it): one byte per group of 16 symbols =
// the group's code bits minus its symbols; sub_words(L) 32-bit words per line -- and of the run-coded lines, after the four
// plain shares of the entry: three header words (the deletion line's token count or RUN_NONE, the same for the substitution
// line, the passes reserved for the deletion line), then one word per group of <= 8 tokens (the tokens a lane of
// k_qv_encode_fast codes in a pass of 512): bits | span << 16 (span = positions the group's runs and symbols cover), 64 per
// pass, deletion line first.  The layout is defined ONCE, in dx_layout.h (the host walk of a bare file writes it too):
#include "dx_layout.h"
#define SUB_NONE     DXL_SUB_NONE                           // first byte of a line: no index (a symbol without a code)
#define RUN_NONE     DXL_RUN_NONE
#define RUN_STRETCH  DXL_RUN_STRETCH                        // positions of a pass the decoder stages in LDS (a longer pass goes byte by byte)
#define RUN_PASSBITS DXL_RUN_PASSBITS                       // bits the 64 groups of a pass may take together (416 words of the decoder's window; else: no index)
__host__ __device__ __forceinline__ uint32_t sub_groups(uint32_t L) { return dxl_sub_groups(L); }
__host__ __device__ __forceinline__ uint32_t sub_words(uint32_t L)  { return dxl_sub_words(L); }
__host__ __device__ __forceinline__ uint32_t run_passes(uint32_t tokens) { return dxl_run_passes(tokens); }
__host__ __device__ __forceinline__ uint32_t run_base(uint32_t L) { return dxl_run_base(L); }    // words before the three header words

// ---------------------------------------------------------------------------------------------
//  per-wave output window: bits are ORed into a zeroed LDS word window (MSB-first within 32-bit
//  words) and leave as coalesced, possibly unaligned, dword stores at the segment's byte address
// ---------------------------------------------------------------------------------------------
struct wave_out
{ uint32_t *win;          // this wave's LDS window (zero outside [0, winbits))
  uint8_t  *seg;          // byte address of the current segment's first word
  uint32_t  wordbase;     // words of this segment already stored
  uint32_t  winbits;      // bits in the window
};

// The same in 16-byte units, for the periodic drain of a well-filled window (16-byte aligned):
// whole groups of four words leave with one ds_read_b128 + one (unaligned) 16-byte store per
// lane; the up to three completed words left over stay in the window with the partial one.
__device__ __forceinline__ void flush_quads(wave_out &o, bool swap)
{ const int      lane = lane_id();
  const uint32_t nq   = o.winbits >> 7;
  u32x4         *win4 = (u32x4 *) o.win;
  wave_sync();
  for (uint32_t j = lane; j < nq; j += 64)
    { u32x4 v = win4[j];
      if (swap)
        { v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
          v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
        }
      *(u32x4_u *) (o.seg + 4ull * o.wordbase + 16ull * j) = v;
    }
  const u32x4 rest = win4[nq];
  const u32x4 zero = { 0u, 0u, 0u, 0u };
  wave_sync();
  for (uint32_t j = lane; j <= nq; j += 64)
    win4[j] = (j == 0) ? rest : zero;
  o.wordbase += 4u * nq;
  o.winbits  &= 127u;
  wave_sync();
}

// store the window's completed words and slide the partial word to win[0]
__device__ __forceinline__ void flush_words(wave_out &o, bool swap)
{ const int      lane  = lane_id();
  const uint32_t nfull = o.winbits >> 5;
  wave_sync();
  for (uint32_t j = lane; j < nfull; j += 64)
    { const uint32_t w = o.win[j];
      store32_u(o.seg + 4ull * (o.wordbase + j), swap ? __builtin_bswap32(w) : w);
    }
  const uint32_t part = o.win[nfull];
  wave_sync();
  for (uint32_t j = lane; j <= nfull; j += 64)
    o.win[j] = (j == 0) ? part : 0u;
  o.wordbase += nfull;
  o.winbits  &= 31u;
  wave_sync();
}

