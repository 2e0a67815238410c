// dx_qv.hip -- the 5-stream Quiver QV coder behind dexqv (QV.c), as gfx950 kernels.
//
// Reference behaviour reproduced (bit-exact):
//   QVcoding_Scan           QV.c:922-1023  -> k_qv_prescan_del / k_qv_prescan_sub / k_qv_hist
//   Histogram_Seqs/_Runs    QV.c:702-724
//   Encode / Encode_Run     QV.c:386-506   -> k_qv_sizes (bit totals + pad rule) and k_qv_encode
//   Pack_Tag + Number_Read + Compress_Read  QV.c:810-819, 1402-1404; DB.c:319-338, 393-416
//   Compress_Next_QVentry   QV.c:1381-1426 (segment order del, tag, ins, mrg, sub; lossy mask)
//
// Roofline: HBM.  Algorithmic bytes per base: histogram pass 4 read; encode pass 5 read + C
// written (C = output bytes per base, ~1.44).  The size pass re-reads 4 and the tag segment
// re-reads the deletion line (both counted against `achieved`, not as algorithmic bytes).
//
// Layout: one 64-lane wavefront per .quiva entry, grid-stride over entries.  A wave walks each
// stream 1 KiB per step (16 bytes per lane, one unaligned global_load_dwordx4).  Code tables
// live in LDS as packed tokens; a DPP inclusive prefix sum over the lanes' bit counts places
// every lane's bits in a per-wave LDS word window (ds_or_b32); completed 32-bit words leave
// with coalesced dword stores at the segment's (byte-granular) file offset.
#include "dx_internal.hpp"
#include "dx_device.hpp"

// Packed token: bits [0,24) code bits (for an escaped symbol: code<<8 | literal), [24,30) length
// in bits, bit 31 = escape flag.  For run schemes the entry holds the bare run code; the 16-bit
// literal of an escaped run is appended at emission time (QV.c:486-487).
#define TOK_LEN(e)  (((e) >> 24) & 0x3fu)
#define TOK_BITS(e) ((e) & 0xffffffu)
#define TOK_ESC(e)  ((e) >> 31)

#define QV_WIN_WORDS 512                     // per-wave LDS window (2 KiB)
#define QV_WIN_BITS  (32u * (QV_WIN_WORDS - 32))   // usable bits: leaves room for one lane's worst case (896 bits)

struct qv_args
{ const uint8_t  *text;
  const uint64_t *off;
  const uint32_t *len;
  uint64_t        n;
  uint32_t        pad;          // line_pad
  int             delChar, subChar;
  int             lossy;
};

__device__ __forceinline__ const uint8_t *line_ptr(const qv_args &a, uint64_t r, uint32_t L, int k)
{ return a.text + a.off[r] + (uint64_t) k * ((uint64_t) L + a.pad); }

// ---------------------------------------------------------------------------------------------
//  run-length bookkeeping shared by the histogram, size and encode kernels
// ---------------------------------------------------------------------------------------------
// For one 1-KiB step of a run-coded stream: `nr` is the lane's 16-bit mask of NON-run symbols.
// Returns the number of run characters immediately preceding this lane's first byte (runs
// continue across lanes and across steps through C) and updates C for the next step.
__device__ __forceinline__ uint32_t run_carry(uint32_t nr, int valid, uint32_t step_valid, uint32_t &C)
{ const int      lane  = lane_id();
  const uint32_t trail = nr ? (uint32_t) valid - 1u - (31u - (uint32_t) __clz(nr)) : 0u;
  const uint64_t Z     = __ballot(nr != 0);
  const uint64_t lower = Z & ((1ull << lane) - 1ull);
  const int      j     = lower ? 63 - __clzll(lower) : 0;
  const uint32_t tj    = __shfl(trail, j);
  const uint32_t carry = lower ? 16u * (uint32_t) (lane - j - 1) + tj : C + 16u * (uint32_t) lane;
  if (Z == 0)
    C += step_valid;
  else
    { const int      jl = 63 - __clzll(Z);                       // wave-uniform
      const uint32_t tl = __builtin_amdgcn_readlane(trail, jl);
      const uint32_t vl = step_valid - 16u * jl >= 16u ? 16u : step_valid - 16u * jl;
      C = tl + step_valid - 16u * jl - vl;
    }
  return carry;
}

// =============================================================================================
//  prescan: delChar / subChar discovery (QV.c:993-1015)
// =============================================================================================

// key = (global entry index << 8) | deletion QV under the first n/N tag of that entry;
// atomicMin keeps the lowest entry.  Waves take entries in ascending order and stop as soon as a
// lower entry has been found, so the common case (an 'N' within the first entry) costs nothing.
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_prescan_del(qv_args a, uint64_t entry0, unsigned long long *key)
{ const int      lane  = lane_id();
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;

  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const unsigned long long cur = __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (entry0 + r >= (cur >> 8))
        break;
      const uint32_t L   = a.len[r];
      const uint8_t *del = line_ptr(a, r, L, 0);
      const uint8_t *tag = line_ptr(a, r, L, 1);
      bool found = false, done = false;
      for (uint32_t base = 0; base < L && !done; base += DX_STEP)
        { const uint32_t pos   = base + 16u * lane;
          const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
          const u32x4    c     = load_chunk(tag + pos, valid);
          const uint32_t m     = (chunk_eq_mask(c, 'n') | chunk_eq_mask(c, 'N')) & ((1u << valid) - 1u);
          const uint64_t any   = __ballot(m != 0);
          if (any)                                  // first n/N of this entry: QV.c:997-1001
            { const int      f = __ffsll((unsigned long long) any) - 1;
              const uint32_t k = pos + (uint32_t) (m ? __ffs(m) - 1 : 0);
              const uint32_t d = __shfl((uint32_t) (lane == f ? del[k] : 0u), f);
              done  = true;
              found = d < 128u;                     // `delChar = Read[k]` goes through a signed char
              if (found && lane == f)
                atomicMin(key, ((unsigned long long) (entry0 + r) << 8) | d);
            }
        }
      if (found)
        break;                                      // this wave's later entries are all higher
    }
}

// Single workgroup: finds the entry at which the running symbol count first reaches 100000 and
// takes the argmax (ties -> smallest value) of the substitution histogram of entries 0..that.
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_prescan_sub(qv_args a, long long *out /* [0]=entry index or -1, [1]=subChar */)
{ __shared__ uint32_t s_hist[256];
  __shared__ uint64_t s_part[DX_BLOCK];
  __shared__ long long s_first;
  const int tid = threadIdx.x;

  s_hist[tid] = 0;
  if (tid == 0) s_first = -1;
  __syncthreads();

  uint64_t running = 0;
  for (uint64_t c0 = 0; c0 < a.n; c0 += DX_BLOCK)
    { const uint64_t i = c0 + tid;
      s_part[tid] = i < a.n ? a.len[i] : 0;
      __syncthreads();
      if (tid == 0)
        for (int k = 0; k < DX_BLOCK && c0 + k < a.n; k++)
          { running += s_part[k];
            if (running >= 100000ull)
              { s_first = (long long) (c0 + k);
                break;
              }
          }
      __syncthreads();
      if (s_first >= 0)
        break;
    }
  const long long first = s_first;
  if (first < 0)
    { if (tid == 0) { out[0] = -1; out[1] = -1; }
      return;
    }

  const int lane = lane_id(), wid = tid >> 6;
  for (long long r = wid; r <= first; r += DX_WAVES_PER_BLK)
    { const uint32_t L   = a.len[r];
      const uint8_t *sub = line_ptr(a, (uint64_t) r, L, 4);
      for (uint32_t pos = lane; pos < L; pos += 64)
        atomicAdd(&s_hist[sub[pos]], 1u);
    }
  __syncthreads();
  if (tid == 0)
    { int best = 0;
      for (int k = 1; k < 256; k++)
        if (s_hist[k] > s_hist[best])
          best = k;
      out[0] = first;
      out[1] = best;
    }
}

// =============================================================================================
//  histogram pass (QV.c:702-724, 988-1017)
// =============================================================================================

__device__ __forceinline__ void hist_plain(const uint8_t *p, uint32_t L, uint32_t *h)
{ const int lane = lane_id();
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos = base + 16u * lane;
      if (L - base >= DX_STEP)                         // full step: no lane is partial
        { const u32x4 c = *(const u32x4_u *) (p + pos);
          #pragma unroll
          for (int b = 0; b < 16; b++)
            atomicAdd(&h[(chunk_word(c, b >> 2) >> (8 * (b & 3))) & 0xffu], 1u);
        }
      else
        { const int   valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
          const u32x4 c     = load_chunk(p + pos, valid);
          for (int b = 0; b < valid; b++)
            atomicAdd(&h[chunk_byte(c, b)], 1u);
        }
    }
}

// symbol histogram + run-length histogram of a run-coded stream; the run character itself is
// counted with popcounts instead of LDS atomics (it is 80-85 % of the stream)
__device__ __forceinline__ void hist_runs(const uint8_t *p, uint32_t L, uint32_t rc, uint32_t *hs, uint32_t *hr)
{ const int lane = lane_id();
  uint32_t  C = 0, nrun = 0;
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos   = base + 16u * lane;
      const uint32_t sv    = L - base >= DX_STEP ? DX_STEP : L - base;
      const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
      const u32x4    c     = load_chunk(p + pos, valid);
      const uint32_t vm    = (1u << valid) - 1u;
      uint32_t       nr    = ~chunk_eq_mask(c, rc) & vm;
      nrun += (uint32_t) valid - __popc(nr);
      uint32_t run  = run_carry(nr, valid, sv, C);
      int      prev = -1;
      while (nr)
        { const int b = __ffs(nr) - 1;
          run += (uint32_t) (b - prev - 1);
          atomicAdd(&hr[run > 255u ? 255u : run], 1u);             // QV.c:717-720
          atomicAdd(&hs[chunk_byte(c, b)], 1u);
          run  = 0;
          prev = b;
          nr  &= nr - 1u;
        }
    }
  if (C > 0 && lane == 0)                                          // stream ends in the run char
    atomicAdd(&hr[C > 255u ? 255u : C], 1u);
  const uint32_t tot = wave_sum(nrun);
  if (lane == 0 && tot)
    atomicAdd(&hs[rc], tot);
}

__global__ __launch_bounds__(DX_BLOCK)
void k_qv_hist(qv_args a, uint64_t entry0, long long del_first, long long sub_first,
               unsigned long long *g_hist /* 6*256 */, unsigned long long *g_tot)
{ __shared__ uint32_t s_hist[6][256];
  const int      lane  = lane_id();
  const int      tid   = threadIdx.x;
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (tid >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;

  for (int k = tid; k < 6 * 256; k += DX_BLOCK)
    (&s_hist[0][0])[k] = 0;
  __syncthreads();

  uint64_t tot = 0, since = 0;
  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const uint32_t  L = a.len[r];
      const long long g = (long long) (entry0 + r);
      if (a.delChar >= 0 && g >= del_first)
        hist_runs(line_ptr(a, r, L, 0), L, (uint32_t) a.delChar, s_hist[DX_DEL], s_hist[DX_DRUN]);
      else
        hist_plain(line_ptr(a, r, L, 0), L, s_hist[DX_DEL]);
      hist_plain(line_ptr(a, r, L, 2), L, s_hist[DX_INS]);
      hist_plain(line_ptr(a, r, L, 3), L, s_hist[DX_MRG]);
      if (a.subChar >= 0 && g >= sub_first)
        hist_runs(line_ptr(a, r, L, 4), L, (uint32_t) a.subChar, s_hist[DX_SUB], s_hist[DX_SRUN]);
      else
        hist_plain(line_ptr(a, r, L, 4), L, s_hist[DX_SUB]);
      tot   += L;
      since += L;
      if (since >= (1ull << 26))                   // keep the 32-bit LDS bins far from overflow
        { for (int k = lane; k < 6 * 256; k += 64)
            { const uint32_t v = atomicExch(&(&s_hist[0][0])[k], 0u);
              if (v) atomicAdd(&g_hist[k], (unsigned long long) v);
            }
          since = 0;
        }
    }
  if (lane == 0 && tot)
    atomicAdd(g_tot, (unsigned long long) tot);
  __syncthreads();
  for (int k = tid; k < 6 * 256; k += DX_BLOCK)
    { const uint32_t v = (&s_hist[0][0])[k];
      if (v) atomicAdd(&g_hist[k], (unsigned long long) v);
    }
}

// =============================================================================================
//  token tables in LDS
// =============================================================================================
__device__ __forceinline__ void load_tables(uint32_t (*s_tok)[256], const uint32_t *g_tok)
{ for (int k = threadIdx.x; k < DX_TOK_WORDS; k += DX_BLOCK)
    (&s_tok[0][0])[k] = g_tok[k];
  __syncthreads();
}

// words Encode/Encode_Run write for T bits whose final OCODE piece had `last` bits
// (QV.c:436-442): the partial word, plus one more when the decoder's 16-bit look-ahead would
// otherwise run past it.
__device__ __forceinline__ uint32_t pad_extra(uint64_t T, uint32_t last)
{ const uint32_t olen = (uint32_t) T & 31u;
  const uint32_t llen = (uint32_t) (T - last) & 31u;
  if (olen > 0)
    return (llen > 16u && olen > llen) ? 1u : 0u;
  return (T > 0 && llen > 16u) ? 1u : 0u;
}

__device__ __forceinline__ uint32_t last_piece_plain(const uint32_t *tab, const uint8_t *p, uint32_t L, uint32_t mask)
{ if (L == 0) return 0;
  const uint32_t e = tab[p[L - 1] & mask];
  return TOK_ESC(e) ? 8u : TOK_LEN(e);
}

// =============================================================================================
//  size pass: bit totals only
// =============================================================================================
__device__ __forceinline__ uint64_t bits_plain(const uint8_t *p, uint32_t L, const uint32_t *tab, uint32_t mask)
{ const int lane = lane_id();
  uint32_t  acc  = 0;
  uint64_t  tot  = 0;
  const uint32_t m4 = mask * 0x01010101u;
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos = base + 16u * lane;
      if (L - base >= DX_STEP)
        { const u32x4 c = *(const u32x4_u *) (p + pos);
          #pragma unroll
          for (int b = 0; b < 16; b++)
            acc += TOK_LEN(tab[((chunk_word(c, b >> 2) & m4) >> (8 * (b & 3))) & 0xffu]);
        }
      else
        { const int   valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
          const u32x4 c     = load_chunk(p + pos, valid);
          for (int b = 0; b < valid; b++)
            acc += TOK_LEN(tab[chunk_byte(c, b) & mask]);
        }
      if ((base & 0x3ffffffu) == 0x3fffc00u)           // fold long before a 32-bit lane sum can wrap
        { tot += wave_sum(acc);
          acc  = 0;
        }
    }
  return tot + wave_sum(acc);
}

__device__ __forceinline__ uint32_t run_token_len(const uint32_t *rtab, uint32_t run)
{ const uint32_t e = rtab[run > 255u ? 255u : run];               // QV.c:479-487
  return TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
}

// bit total of Encode_Run; also returns the number of non-run symbols (Pack_Tag's clen) and the
// length of the final piece
__device__ __forceinline__ uint64_t bits_runs(const uint8_t *p, uint32_t L, uint32_t rc,
                                              const uint32_t *ntab, const uint32_t *rtab,
                                              uint32_t &nonrun, uint32_t &last)
{ const int lane = lane_id();
  uint32_t  C = 0, acc = 0, nn = 0;
  uint64_t  tot = 0;
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos   = base + 16u * lane;
      const uint32_t sv    = L - base >= DX_STEP ? DX_STEP : L - base;
      const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
      const u32x4    c     = load_chunk(p + pos, valid);
      uint32_t       nr    = ~chunk_eq_mask(c, rc) & ((1u << valid) - 1u);
      nn += __popc(nr);
      uint32_t run  = run_carry(nr, valid, sv, C);
      int      prev = -1;
      while (nr)
        { const int b = __ffs(nr) - 1;
          run += (uint32_t) (b - prev - 1);
          acc += run_token_len(rtab, run) + TOK_LEN(ntab[chunk_byte(c, b)]);
          run  = 0;
          prev = b;
          nr  &= nr - 1u;
        }
      if ((base & 0x3ffffffu) == 0x3fffc00u)
        { tot += wave_sum(acc);
          acc  = 0;
        }
    }
  tot   += wave_sum(acc);
  nonrun = wave_sum(nn);
  if (C > 0)                                                      // trailing run token
    { const uint32_t e = rtab[C > 255u ? 255u : C];
      tot += TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
      last = TOK_ESC(e) ? 16u : TOK_LEN(e);
    }
  else
    last = last_piece_plain(ntab, p, L, 0xffu);
  return tot;
}

__device__ __forceinline__ uint32_t seg_bytes(uint64_t T, uint32_t last)
{ return 4u * ((uint32_t) (T >> 5) + (((uint32_t) T & 31u) ? 1u : 0u) + pad_extra(T, last)); }

__global__ __launch_bounds__(DX_BLOCK)
void k_qv_sizes(qv_args a, const uint32_t *g_tok, const uint64_t *hdr_off, uint32_t *rec_size)
{ __shared__ uint32_t s_tok[6][256];
  load_tables(s_tok, g_tok);
  const int      lane  = lane_id();
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  const uint32_t imask = a.lossy ? 0xfeu : 0xffu, mmask = a.lossy ? 0xfcu : 0xffu;

  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const uint32_t L = a.len[r];
      uint32_t sz = hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u;
      uint32_t clen = L, last;
      uint64_t T;

      const uint8_t *del = line_ptr(a, r, L, 0);
      if (a.delChar >= 0)
        T = bits_runs(del, L, (uint32_t) a.delChar, s_tok[DX_DEL], s_tok[DX_DRUN], clen, last);
      else
        { T = bits_plain(del, L, s_tok[DX_DEL], 0xffu);
          last = last_piece_plain(s_tok[DX_DEL], del, L, 0xffu);
        }
      sz += seg_bytes(T, last) + ((clen + 3u) >> 2);

      const uint8_t *ins = line_ptr(a, r, L, 2);
      T   = bits_plain(ins, L, s_tok[DX_INS], imask);
      sz += seg_bytes(T, last_piece_plain(s_tok[DX_INS], ins, L, imask));

      const uint8_t *mrg = line_ptr(a, r, L, 3);
      T   = bits_plain(mrg, L, s_tok[DX_MRG], mmask);
      sz += seg_bytes(T, last_piece_plain(s_tok[DX_MRG], mrg, L, mmask));

      const uint8_t *sub = line_ptr(a, r, L, 4);
      if (a.subChar >= 0)
        { uint32_t nn;
          T = bits_runs(sub, L, (uint32_t) a.subChar, s_tok[DX_SUB], s_tok[DX_SRUN], nn, last);
        }
      else
        { T = bits_plain(sub, L, s_tok[DX_SUB], 0xffu);
          last = last_piece_plain(s_tok[DX_SUB], sub, L, 0xffu);
        }
      sz += seg_bytes(T, last);

      if (lane == 0)
        rec_size[r] = sz;
    }
}

// =============================================================================================
//  exclusive scan of the record sizes (file order) -> record offsets
// =============================================================================================
#define SCAN_ITEMS 16                                   // items per thread
#define SCAN_TILE  (DX_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t *s_wave, uint64_t &total)
{ // inclusive scan of 64-bit values within the wave via two 32-bit halves with carry
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  uint64_t x = v;
  for (int d = 1; d < 64; d <<= 1)
    { const uint64_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
  if (lane == 63) s_wave[wid] = x;
  __syncthreads();
  uint64_t pre = 0, tot = 0;
  for (int w = 0; w < DX_WAVES_PER_BLK; w++)
    { if (w < wid) pre += s_wave[w];
      tot += s_wave[w];
    }
  __syncthreads();
  total = tot;
  return pre + x - v;
}

__global__ __launch_bounds__(DX_BLOCK)
void k_scan_tiles(const uint32_t *in, uint64_t n, uint64_t *tile_sum)
{ __shared__ uint64_t s_wave[DX_WAVES_PER_BLK];
  const uint64_t t0 = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (t0 + k < n) s += in[t0 + k];
  uint64_t tot;
  block_excl_scan(s, s_wave, tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(DX_BLOCK)
void k_scan_sums(uint64_t *tile_sum, uint64_t ntiles, uint64_t *grand)
{ __shared__ uint64_t s_wave[DX_WAVES_PER_BLK];
  uint64_t running = 0;
  for (uint64_t c0 = 0; c0 < ntiles; c0 += DX_BLOCK)
    { const uint64_t i = c0 + threadIdx.x;
      const uint64_t v = i < ntiles ? tile_sum[i] : 0;
      uint64_t tot;
      const uint64_t ex = block_excl_scan(v, s_wave, tot);
      if (i < ntiles) tile_sum[i] = running + ex;
      running += tot;
    }
  if (threadIdx.x == 0) *grand = running;
}

__global__ __launch_bounds__(DX_BLOCK)
void k_scan_apply(const uint32_t *in, uint64_t n, const uint64_t *tile_sum, uint64_t *out /* n+1 */,
                  const uint64_t *grand)
{ __shared__ uint64_t s_wave[DX_WAVES_PER_BLK];
  const uint64_t t0 = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
  for (int k = 0; k < SCAN_ITEMS; k++)
    { v[k] = t0 + k < n ? in[t0 + k] : 0u;
      s   += v[k];
    }
  uint64_t tot;
  uint64_t at = tile_sum[blockIdx.x] + block_excl_scan(s, s_wave, tot);
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (t0 + k < n)
      { out[t0 + k] = at;
        at += v[k];
      }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    out[n] = *grand;
}

// =============================================================================================
//  encode pass
// =============================================================================================
struct wave_out
{ uint32_t *win;          // this wave's LDS window (zero outside [0, winbits))
  uint8_t  *seg;          // byte address of the current segment's first word
  uint32_t  wordbase;     // words of this segment already stored
  uint32_t  winbits;      // bits in the window
};

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

// lane-local MSB-first bit accumulator feeding the window
struct bit_acc { uint64_t acc; uint32_t fill, w; };

__device__ __forceinline__ void acc_begin(bit_acc &s, uint32_t bit) { s.acc = 0; s.fill = bit & 31u; s.w = bit >> 5; }

__device__ __forceinline__ void acc_put(bit_acc &s, uint32_t *win, uint32_t bits, uint32_t len)
{ s.acc  |= (uint64_t) bits << ((64u - s.fill - len) & 63u);
  s.fill += len;
  if (s.fill >= 32u)
    { atomicOr(&win[s.w], (uint32_t) (s.acc >> 32));
      s.w    += 1;
      s.acc <<= 32;
      s.fill -= 32u;
    }
}

__device__ __forceinline__ void acc_end(bit_acc &s, uint32_t *win)
{ if (s.fill && (uint32_t) (s.acc >> 32))
    atomicOr(&win[s.w], (uint32_t) (s.acc >> 32));
}

// The lanes' bit strings of one step go into the window in lane order; when the step does not
// fit (only with pathological code tables) it is split at lane boundaries into several rounds.
// `incl` is the inclusive prefix sum of the lanes' bit counts `nb`.  EMIT(lane_bit_offset) writes
// one lane's bits.
#define FOR_EACH_ROUND(o, incl, nb, ...)                                                         \
  { uint32_t done_ = 0, lo_ = 0;                                                                \
    const int lane_ = lane_id();                                                                \
    while (lo_ < 64u)                                                                           \
      { const uint32_t cap_ = QV_WIN_BITS - (o).winbits;                                        \
        uint32_t hi_        = (uint32_t) __popcll(__ballot((incl) <= done_ + cap_));            \
        if (hi_ <= lo_) hi_ = 64u;   /* cannot happen (one lane's bits always fit); never spin */ \
        if ((uint32_t) lane_ >= lo_ && (uint32_t) lane_ < hi_ && (nb))                          \
          { const uint32_t bit_ = (o).winbits + ((incl) - (nb)) - done_;                        \
            __VA_ARGS__                                                                         \
          }                                                                                     \
        const uint32_t upto_ = __builtin_amdgcn_readlane((incl), (int) hi_ - 1);                \
        (o).winbits += upto_ - done_;                                                           \
        done_ = upto_;                                                                          \
        lo_   = hi_;                                                                            \
        flush_words((o), false);                                                                \
      }                                                                                         \
  }

// Encode (QV.c:386-443) of one stream into the segment at o.seg; returns the segment's bytes
__device__ __forceinline__ uint32_t encode_plain(wave_out &o, const uint8_t *p, uint32_t L,
                                                 const uint32_t *tab, uint32_t mask)
{ const int lane = lane_id();
  o.wordbase = 0;
  o.winbits  = 0;
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos   = base + 16u * lane;
      const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
      const u32x4    c     = load_chunk(p + pos, valid);
      uint32_t tok[16];
      uint32_t nb = 0;
      #pragma unroll
      for (int b = 0; b < 16; b++)
        { const uint32_t x = (chunk_word(c, b >> 2) >> (8 * (b & 3))) & mask;
          tok[b] = b < valid ? tab[x] : 0u;
          nb    += TOK_LEN(tok[b]);
        }
      const uint32_t incl = wave_incl_scan(nb);
      FOR_EACH_ROUND(o, incl, nb,
        { bit_acc s;
          acc_begin(s, bit_);
          _Pragma("unroll")
          for (int b = 0; b < 16; b++)
            acc_put(s, o.win, TOK_BITS(tok[b]), TOK_LEN(tok[b]));
          acc_end(s, o.win);
        })
    }
  // tail: partial word and the pad word of QV.c:436-442
  const uint64_t T     = 32ull * o.wordbase + o.winbits;
  const uint32_t last  = last_piece_plain(tab, p, L, mask);
  const uint32_t extra = pad_extra(T, last);
  const uint32_t tailw = (o.winbits ? 1u : 0u) + extra;
  if (lane == 0)
    { const uint32_t w = o.win[0];
      for (uint32_t k = 0; k < tailw; k++)
        store32_u(o.seg + 4ull * (o.wordbase + k), w);
      o.win[0] = 0;
    }
  wave_sync();
  return 4u * (o.wordbase + tailw);
}

// Encode_Run (QV.c:448-506)
__device__ __forceinline__ uint32_t encode_runs(wave_out &o, const uint8_t *p, uint32_t L, uint32_t rc,
                                                const uint32_t *ntab, const uint32_t *rtab)
{ const int lane = lane_id();
  uint32_t  C = 0;
  o.wordbase = 0;
  o.winbits  = 0;
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos   = base + 16u * lane;
      const uint32_t sv    = L - base >= DX_STEP ? DX_STEP : L - base;
      const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
      const u32x4    c     = load_chunk(p + pos, valid);
      const uint32_t nr0   = ~chunk_eq_mask(c, rc) & ((1u << valid) - 1u);
      const uint32_t carry = run_carry(nr0, valid, sv, C);

      uint32_t nb = 0;
      { uint32_t nr = nr0, run = carry;
        int      prev = -1;
        while (nr)
          { const int b = __ffs(nr) - 1;
            run += (uint32_t) (b - prev - 1);
            nb  += run_token_len(rtab, run) + TOK_LEN(ntab[chunk_byte(c, b)]);
            run  = 0;
            prev = b;
            nr  &= nr - 1u;
          }
      }
      const uint32_t incl = wave_incl_scan(nb);
      FOR_EACH_ROUND(o, incl, nb,
        { bit_acc  s;
          uint32_t nr = nr0, run = carry;
          int      prev = -1;
          acc_begin(s, bit_);
          while (nr)
            { const int b = __ffs(nr) - 1;
              run += (uint32_t) (b - prev - 1);
              const uint32_t re = rtab[run > 255u ? 255u : run];
              if (TOK_ESC(re))
                acc_put(s, o.win, (TOK_BITS(re) << 16) | run, TOK_LEN(re) + 16u);   // QV.c:486-487
              else
                acc_put(s, o.win, TOK_BITS(re), TOK_LEN(re));
              const uint32_t se = ntab[chunk_byte(c, b)];
              acc_put(s, o.win, TOK_BITS(se), TOK_LEN(se));
              run  = 0;
              prev = b;
              nr  &= nr - 1u;
            }
          acc_end(s, o.win);
        })
    }
  uint32_t last;
  if (C > 0)                                                       // trailing run: one run-only token
    { const uint32_t re = rtab[C > 255u ? 255u : C];
      const uint32_t tl = TOK_LEN(re) + (TOK_ESC(re) ? 16u : 0u);
      if (lane == 0 && tl)
        { bit_acc s;
          acc_begin(s, o.winbits);
          acc_put(s, o.win, TOK_ESC(re) ? ((TOK_BITS(re) << 16) | C) : TOK_BITS(re), tl);
          acc_end(s, o.win);
        }
      o.winbits += tl;
      flush_words(o, false);
      last = TOK_ESC(re) ? 16u : TOK_LEN(re);
    }
  else
    last = last_piece_plain(ntab, p, L, 0xffu);

  const uint64_t T     = 32ull * o.wordbase + o.winbits;
  const uint32_t extra = pad_extra(T, last);
  const uint32_t tailw = (o.winbits ? 1u : 0u) + extra;
  if (lane == 0)
    { const uint32_t w = o.win[0];
      for (uint32_t k = 0; k < tailw; k++)
        store32_u(o.seg + 4ull * (o.wordbase + k), w);
      o.win[0] = 0;
    }
  wave_sync();
  return 4u * (o.wordbase + tailw);
}

// Pack_Tag + Number_Read + Compress_Read (QV.c:810-819, 1402-1404): tags at positions where
// del != delChar (all positions when rc < 0), 2 bits each
__device__ __forceinline__ uint32_t encode_tags(wave_out &o, const uint8_t *del, const uint8_t *tag,
                                                uint32_t L, int rc)
{ const int lane = lane_id();
  uint32_t  G = 0;
  o.wordbase = 0;
  o.winbits  = 0;
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const uint32_t pos   = base + 16u * lane;
      const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
      const u32x4    t     = load_chunk(tag + pos, valid);
      uint32_t keep = (1u << valid) - 1u;
      if (rc >= 0)
        keep &= ~chunk_eq_mask(load_chunk(del + pos, valid), (uint32_t) rc);
      const uint32_t cnt = __popc(keep);
      uint32_t acc = 0;
      int      sh  = 30;
      #pragma unroll
      for (int b = 0; b < 16; b++)
        if ((keep >> b) & 1u)
          { const uint32_t u = ((chunk_word(t, b >> 2) >> (8 * (b & 3))) & 0xffu) & 0xdfu;
            acc |= ((u == 'C') ? 1u : (u == 'G') ? 2u : (u == 'T') ? 3u : 0u) << sh;
            sh  -= 2;
          }
      const uint32_t incl = wave_incl_scan(cnt);
      if (cnt)
        { const uint32_t bit = o.winbits + 2u * (incl - cnt);
          const uint32_t w = bit >> 5, s = bit & 31u;
          atomicOr(&o.win[w], acc >> s);
          if (s && 2u * cnt + s > 32u)
            atomicOr(&o.win[w + 1], acc << (32u - s));
        }
      const uint32_t total = wave_total(incl);
      o.winbits += 2u * total;
      G         += total;
      flush_words(o, true);
    }
  const uint32_t clen = (G + 3u) >> 2;
  const uint32_t done = 4u * o.wordbase;
  if (lane == 0)
    { const uint32_t w = __builtin_bswap32(o.win[0]);
      for (uint32_t k = done; k < clen; k++)
        o.seg[k] = (uint8_t) (w >> (8 * (k - done)));
      o.win[0] = 0;
    }
  wave_sync();
  return clen;
}

__global__ __launch_bounds__(DX_BLOCK)
void k_qv_encode(qv_args a, const uint32_t *g_tok, const uint8_t *hdr, const uint64_t *hdr_off,
                 const uint64_t *rec_off, uint8_t *out, uint32_t *seg_out)
{ __shared__ uint32_t s_tok[6][256];
  __shared__ uint32_t s_win[DX_WAVES_PER_BLK][QV_WIN_WORDS];
  load_tables(s_tok, g_tok);
  const int      lane  = lane_id();
  const int      wid   = threadIdx.x >> 6;
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + wid;
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  const uint32_t imask = a.lossy ? 0xfeu : 0xffu, mmask = a.lossy ? 0xfcu : 0xffu;

  wave_out o;
  o.win = s_win[wid];
  for (int j = lane; j < QV_WIN_WORDS; j += 64)
    o.win[j] = 0;
  wave_sync();

  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const uint32_t L   = a.len[r];
      uint8_t       *dst = out + rec_off[r];
      if (hdr != NULL)                                   // record framing (dexqv.c:128-139)
        { const uint64_t h0 = hdr_off[r];
          const uint32_t hl = (uint32_t) (hdr_off[r + 1] - h0);
          for (uint32_t k = lane; k < hl; k += 64)
            dst[k] = hdr[h0 + k];
          dst += hl;
        }
      const uint8_t *del = line_ptr(a, r, L, 0);
      uint32_t sz[5];

      o.seg = dst;                                       // QV.c:1393-1401
      sz[0] = (a.delChar >= 0)
                ? encode_runs(o, del, L, (uint32_t) a.delChar, s_tok[DX_DEL], s_tok[DX_DRUN])
                : encode_plain(o, del, L, s_tok[DX_DEL], 0xffu);
      o.seg += sz[0];                                    // QV.c:1400-1404
      sz[1]  = encode_tags(o, del, line_ptr(a, r, L, 1), L, a.delChar);
      o.seg += sz[1];                                    // QV.c:1406-1418
      sz[2]  = encode_plain(o, line_ptr(a, r, L, 2), L, s_tok[DX_INS], imask);
      o.seg += sz[2];
      sz[3]  = encode_plain(o, line_ptr(a, r, L, 3), L, s_tok[DX_MRG], mmask);
      o.seg += sz[3];                                    // QV.c:1419-1423
      sz[4]  = (a.subChar >= 0)
                ? encode_runs(o, line_ptr(a, r, L, 4), L, (uint32_t) a.subChar, s_tok[DX_SUB], s_tok[DX_SRUN])
                : encode_plain(o, line_ptr(a, r, L, 4), L, s_tok[DX_SUB], 0xffu);

      if (seg_out != NULL && lane < 5)
        seg_out[5 * r + lane] = sz[lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? 2 : lane == 3 ? 3 : 4];
    }
}

// =============================================================================================
//  C-ABI
// =============================================================================================
static int check_batch(dx_ctx *ctx, const dx_qv_batch *b, const char *who)
{ if (ctx == NULL) return DX_E_ARG;
  if (b == NULL) return dx_fail(ctx, DX_E_ARG, "%s: NULL batch", who);
  if (b->n && (!b->d_text || !b->d_off || !b->d_len))
    return dx_fail(ctx, DX_E_ARG, "%s: NULL device pointer in batch", who);
  return DX_OK;
}

static qv_args make_args(const dx_qv_batch *b, int delChar, int subChar, int lossy)
{ qv_args a;
  a.text = b->d_text; a.off = b->d_off; a.len = b->d_len; a.n = b->n; a.pad = b->line_pad;
  a.delChar = delChar; a.subChar = subChar; a.lossy = lossy;
  return a;
}

extern "C" int dx_qv_prescan(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, dx_qv_params *p)
{ int e = check_batch(ctx, b, "dx_qv_prescan");
  if (e) return e;
  if (p == NULL) return dx_fail(ctx, DX_E_ARG, "dx_qv_prescan: NULL params");
  if (b->n == 0) return DX_OK;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  qv_args a = make_args(b, -1, -1, 0);
  unsigned long long *d_key = (unsigned long long *) ctx->d_u64;
  long long          *d_sub = (long long *) (ctx->d_u64 + 2);

  const bool want_del = p->delChar < 0;
  const bool want_sub = p->subChar < 0 && entry0 == 0;
  if (want_del)
    { DX_HIP(ctx, hipMemsetAsync(d_key, 0xff, 8, ctx->stream));
      DX_LAUNCH(ctx, DX_K_QV_PRESCAN, k_qv_prescan_del, dx_grid_waves(ctx, b->n, 8), DX_BLOCK, a, entry0, d_key);
    }
  if (want_sub)
    DX_LAUNCH(ctx, DX_K_QV_PRESCAN, k_qv_prescan_sub, 1, DX_BLOCK, a, d_sub);
  unsigned long long key = ~0ull;
  long long sub[2] = { -1, -1 };
  if (want_del)
    DX_HIP(ctx, hipMemcpyAsync(&key, d_key, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (want_sub)
    DX_HIP(ctx, hipMemcpyAsync(sub, d_sub, 16, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (want_del && key != ~0ull)
    { p->delChar   = (int32_t) (key & 0xff);
      p->del_first = (int64_t) (key >> 8);
    }
  if (want_sub && sub[0] >= 0)
    { p->subChar   = (int32_t) sub[1];
      p->sub_first = sub[0];
    }
  return DX_OK;
}

extern "C" int dx_qv_hist(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, const dx_qv_params *p,
                          uint64_t hist[6][256], uint64_t *totChar)
{ int e = check_batch(ctx, b, "dx_qv_hist");
  if (e) return e;
  if (p == NULL || hist == NULL || totChar == NULL)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_hist: NULL argument");
  if (b->n == 0) return DX_OK;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long *d_hist;
  if ((e = dx_scratch(ctx, (6 * 256 + 1) * 8, (void **) &d_hist))) return e;
  DX_HIP(ctx, hipMemsetAsync(d_hist, 0, (6 * 256 + 1) * 8, ctx->stream));
  qv_args a = make_args(b, p->delChar, p->subChar, 0);
  DX_LAUNCH(ctx, DX_K_QV_HIST, k_qv_hist, dx_grid_waves(ctx, b->n, 16), DX_BLOCK,
            a, entry0, (long long) p->del_first, (long long) p->sub_first, d_hist, d_hist + 6 * 256);
  static_assert(sizeof(unsigned long long) == 8, "u64");
  uint64_t host[6 * 256 + 1];
  DX_HIP(ctx, hipMemcpyAsync(host, d_hist, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < 6; s++)
    for (int k = 0; k < 256; k++)
      hist[s][k] += host[s * 256 + k];
  *totChar += host[6 * 256];
  return DX_OK;
}

static uint32_t pack_sym(const dx_scheme *s, int x)
{ const uint32_t len = (uint32_t) s->lens[x], bits = s->bits[x];
  if (len == 0) return 0;
  const bool esc = s->type == 2 && bits == s->bits[255] && s->lens[x] == s->lens[255];   // QV.c:432
  return esc ? (((bits << 8) | (uint32_t) x) & 0xffffffu) | ((len + 8u) << 24) | 0x80000000u
             : (bits & 0xffffffu) | (len << 24);
}

static uint32_t pack_run(const dx_scheme *s, int x)
{ const uint32_t len = (uint32_t) s->lens[x], bits = s->bits[x];
  const bool esc = bits == s->bits[255] && s->lens[x] == s->lens[255];                   // QV.c:468-469, 486
  return (bits & 0xffffffu) | (len << 24) | (esc ? 0x80000000u : 0u);
}

extern "C" int dx_qv_set_coding(dx_ctx *ctx, const dx_qv_coding *c, int lossy)
{ if (ctx == NULL) return DX_E_ARG;
  if (c == NULL) return dx_fail(ctx, DX_E_ARG, "dx_qv_set_coding: NULL coding");
  uint32_t tok[DX_TOK_WORDS];
  memset(tok, 0, sizeof(tok));
  for (int s = 0; s < 6; s++)
    { const bool run = s >= DX_DRUN;
      if ((s == DX_DRUN && c->delChar < 0) || (s == DX_SRUN && c->subChar < 0))
        continue;
      for (int x = 0; x < 256; x++)
        { if (c->s[s].lens[x] > 16)
            return dx_fail(ctx, DX_E_UNSUPPORTED, "scheme %d has a %d-bit code for symbol %d "
                           "(> 16: not decodable by the reference either)", s, c->s[s].lens[x], x);
          tok[s * 256 + x] = run ? pack_run(&c->s[s], x) : pack_sym(&c->s[s], x);
        }
    }
  DX_HIP(ctx, hipSetDevice(ctx->device));
  DX_HIP(ctx, hipMemcpyAsync(ctx->d_tok, tok, sizeof(tok), hipMemcpyHostToDevice, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->coding_set = 1;
  ctx->lossy   = lossy != 0;
  ctx->delChar = c->delChar;
  ctx->subChar = c->subChar;
  return DX_OK;
}

extern "C" int dx_qv_sizes(dx_ctx *ctx, const dx_qv_batch *b, const uint64_t *d_hdr_off,
                           uint64_t *d_rec_off, uint64_t *total)
{ int e = check_batch(ctx, b, "dx_qv_sizes");
  if (e) return e;
  if (!ctx->coding_set) return dx_fail(ctx, DX_E_ARG, "dx_qv_sizes: call dx_qv_set_coding first");
  if (d_rec_off == NULL) return dx_fail(ctx, DX_E_ARG, "dx_qv_sizes: NULL d_rec_off");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n      = b->n;
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint8_t *scr;
  if ((e = dx_scratch(ctx, n * 4 + 64 + (ntiles + 2) * 8, (void **) &scr))) return e;
  uint32_t *d_size = (uint32_t *) scr;
  uint64_t *d_tile = (uint64_t *) (scr + ((n * 4 + 63) & ~(size_t) 63));
  uint64_t *d_gran = d_tile + ntiles;
  if (n == 0)
    { uint64_t z = 0;
      DX_HIP(ctx, hipMemcpyAsync(d_rec_off, &z, 8, hipMemcpyHostToDevice, ctx->stream));
      DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (total) *total = 0;
      return DX_OK;
    }
  qv_args a = make_args(b, ctx->delChar, ctx->subChar, ctx->lossy);
  DX_LAUNCH(ctx, DX_K_QV_SIZES, k_qv_sizes, dx_grid_waves(ctx, n, 16), DX_BLOCK,
            a, (const uint32_t *) ctx->d_tok, d_hdr_off, d_size);
  DX_LAUNCH(ctx, DX_K_SCAN, k_scan_tiles, (int) ntiles, DX_BLOCK, (const uint32_t *) d_size, n, d_tile);
  DX_LAUNCH(ctx, DX_K_SCAN, k_scan_sums, 1, DX_BLOCK, d_tile, ntiles, d_gran);
  DX_LAUNCH(ctx, DX_K_SCAN, k_scan_apply, (int) ntiles, DX_BLOCK, (const uint32_t *) d_size, n,
            (const uint64_t *) d_tile, d_rec_off, (const uint64_t *) d_gran);
  if (total)
    { DX_HIP(ctx, hipMemcpyAsync(total, d_gran, 8, hipMemcpyDeviceToHost, ctx->stream));
      DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  return DX_OK;
}

extern "C" int dx_qv_encode(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                            const uint64_t *d_rec_off, uint8_t *d_out, uint32_t *d_seg)
{ int e = check_batch(ctx, b, "dx_qv_encode");
  if (e) return e;
  if (!ctx->coding_set) return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: call dx_qv_set_coding first");
  if ((d_hdr == NULL) != (d_hdr_off == NULL))
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: d_hdr and d_hdr_off must be given together");
  if (b->n == 0) return DX_OK;
  if (!d_rec_off || !d_out) return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: NULL device pointer");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  qv_args a = make_args(b, ctx->delChar, ctx->subChar, ctx->lossy);
  DX_LAUNCH(ctx, DX_K_QV_ENCODE, k_qv_encode, dx_grid_waves(ctx, b->n, 16), DX_BLOCK,
            a, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off, d_rec_off, d_out, d_seg);
  return DX_OK;
}
