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
// written (C = output bytes per base, ~1.44).  The size pass re-reads 4 B/base (counted against
// `achieved`, not as algorithmic bytes).
//
// Layout: one 64-lane wavefront per .quiva entry; the waves draw entries from a ticket counter
// (next_unit), so they neither march in step nor wait for the longest entry.  A wave walks the
// streams 1 KiB per step (16 bytes per lane, one unaligned global_load_dwordx4 each), with the
// next step's chunks already in flight (register prefetch) while the current ones are processed.
// Code tables live in LDS as packed tokens; a DPP inclusive prefix sum over the lanes' bit counts
// places every lane's bits in a per-wave LDS word window (ds_or_b32); completed words leave in
// 16-byte units at the segment's (byte-granular) file offset.  All three kernels are bound by
// VALU issue, not by HBM: what counts is instructions per byte.
#include "dx_internal.hpp"
#include "dx_device.hpp"

// the text's chunks as k_qv_hist asks for them: read once, never again by this kernel -- HIST_NT: marked so (the lines they would
// push out of the L2 include the token lines under way: a step's tokens end in the middle of a cache line the next step fills up)
#ifndef HIST_NT
#define HIST_NT 0
#endif
#if HIST_NT
#define HIST_LOAD(p) __builtin_nontemporal_load((const u32x4_u *) (p))
#else
#define HIST_LOAD(p) (*(const u32x4_u *) (p))
#endif

// Packed token: bits [0,24) code bits (for an escaped symbol: code<<8 | literal), bit 25 = escape
// flag, [26,32) length in bits (a single shift extracts it; a symbol without a code is entry 0).
// For run schemes the entry holds the bare run code; the 16-bit literal of an escaped run is
// appended at emission time (QV.c:486-487).
#define TOK_LEN(e)  ((e) >> 26)
#define TOK_BITS(e) ((e) & 0xffffffu)
#define TOK_ESC(e)  (((e) >> 25) & 1u)
#define TOK_PACK(bits, len, esc) (((bits) & 0xffffffu) | ((uint32_t) (len) << 26) | ((esc) ? 0x2000000u : 0u))

#define QV_WIN_WORDS  512                          // per-wave LDS bit window (2 KiB)
#define QV_WIN_BITS   (32u * (QV_WIN_WORDS - 32))  // usable: one lane's worst case (896 bits) always fits
#define QV_WIN_PAD    4                            // words in front of a window that place_bits128 may name (with a zero)
#ifndef QV_FLUSH_BITS
#define QV_FLUSH_BITS 8192u                        // drain the window once it holds this much
#endif
#define TAG_WIN_WORDS 128                          // per-wave window of the tag segment
#define TAG_FLUSH_BITS 1024u

// minimum waves per SIMD the register allocator must leave room for (caps VGPRs: 512 / waves)
#ifndef SIZES_WAVES
#define SIZES_WAVES 4
#endif
#ifndef ENC_WAVES
#define ENC_WAVES 4
#endif
// perturbation experiments (tools/microbench/hist_time.py): parts of k_qv_hist compiled out -- wrong results, a kernel
// time that says what the part costs.  1: no token rounds, 2: no plain lines, 4: no position lists, 8: no tokens (no tag
// look-up, no assembly, no store), 16: no run-coded lines at all, 32: tokens made but not stored to memory
#ifndef HIST_SKIP
#define HIST_SKIP 0
#endif
#ifndef HIST_TRIM
#define HIST_TRIM 0
#endif

struct qv_args
{ const uint8_t  *text;
  const uint64_t *off;
  const uint32_t *len;
  uint64_t        n;
  uint64_t        text_bytes;   // readable bytes at text (0: unknown -> never read past a line)
  uint32_t        pad;          // line_pad
  int             delChar, subChar;
  int             lossy;
  uint32_t        units;        // entries per ticket (ticket_units on the host: 2 of 10 kb, more of shorter ones)
};

// Units per ticket for a batch whose n units hold `bytes` bytes in all: about `target` bytes' worth, `least` at least.
// Every draw is an atomic on ONE address, ~11 ns each chip-wide: with a fixed few units per ticket a batch of many short
// entries or reads is bound by its counter (4 M entries of 300, two per ticket: 22 ms of draws under every kernel).
static uint32_t ticket_units(uint64_t bytes, uint64_t n, uint32_t target, uint32_t least)
{ if (n == 0 || bytes == 0) return least;
  const uint64_t mean = bytes / n + 1u;
  uint64_t u = ((uint64_t) target + mean - 1u) / mean;
  return (uint32_t) (u < least ? least : (u > 4096u ? 4096u : u));
}

__device__ __forceinline__ const uint8_t *line_ptr(const qv_args &a, uint64_t r, uint32_t L, int k)
{ return a.text + a.off[r] + (uint64_t) k * ((uint64_t) L + a.pad); }

// may the last (partial) lane of this line be read with a full 16-byte load?
__device__ __forceinline__ bool can_overread(const qv_args &a, const uint8_t *p, uint32_t L)
{ return (uint64_t) (p - a.text) + L + 16u <= a.text_bytes; }

// this lane's 16 bytes of a stream at p+pos: bytes [0,valid) real, the rest zero
__device__ __forceinline__ u32x4 fetch(const uint8_t *p, uint32_t pos, uint32_t L, bool over)
{ u32x4 v = { 0u, 0u, 0u, 0u };
  if (pos < L)
    { const uint32_t left = L - pos;
      if (left >= 16u)
        v = *(const u32x4_u *) (p + pos);
      else if (over)
        { v = *(const u32x4_u *) (p + pos);
          const uint32_t m0 = left >= 4u  ? ~0u : ~(~0u << (8u * left));
          const uint32_t m1 = left >= 8u  ? ~0u : (left > 4u  ? ~(~0u << (8u * (left - 4u)))  : 0u);
          const uint32_t m2 = left >= 12u ? ~0u : (left > 8u  ? ~(~0u << (8u * (left - 8u)))  : 0u);
          const uint32_t m3 =                     (left > 12u ? ~(~0u << (8u * (left - 12u))) : 0u);
          v.x &= m0; v.y &= m1; v.z &= m2; v.w &= m3;
        }
      else
        v = load_chunk(p + pos, (int) left);
    }
  return v;
}

__device__ __forceinline__ int valid_of(uint32_t pos, uint32_t L)
{ return pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos)); }

// the chunks of the step that starts at `base` (wave-uniform): when the whole step lies inside the line -- every step
// but a line's last -- one scalar test sends all 64 lanes to a bare 16-byte load; fetch()'s per-lane bounds tests
// (three branches on the execution mask and the clearing of the result registers, per line and step) are then paid
// only at the line's end
__device__ __forceinline__ u32x4 fetch_step(const uint8_t *p, uint32_t base, uint32_t L, bool over)
{ const uint32_t pos = base + 16u * (uint32_t) lane_id();
  if (base + DX_STEP <= L)
    return *(const u32x4_u *) (p + pos);
  return fetch(p, pos, L, over);
}

#define BYTE_OF(c, b) ((chunk_word(c, (b) >> 2) >> (8 * ((b) & 3))) & 0xffu)

// ---------------------------------------------------------------------------------------------
//  run-coded streams: dense token processing (shared by the histogram, size and encode kernels)
// ---------------------------------------------------------------------------------------------
// Only the non-run symbols of a run-coded stream (15-20 % of it) produce tokens.  Letting every
// lane loop over the non-run symbols of its own 16 bytes runs to the worst lane's count with a
// third of the lanes busy.  Instead each step first compacts the positions of its non-run symbols
// into an LDS list (wave prefix sum of the per-lane counts) next to a copy of the 1-KiB chunk;
// the tokens are then processed 64 at a time, one per lane: the run before token i is simply
// list[i] - list[i-1] - 1, and C carries the run that is open at the step's start.
struct run_lds
{ uint8_t  *chunk;        // this wave's 1 KiB copy of the step's bytes (16-byte aligned)
  uint16_t *list;         // ascending step-relative positions of the non-run symbols
};

__device__ __forceinline__ uint32_t run_collect(const run_lds &R, const u32x4 &c, int valid, uint32_t rc)
{ const int lane = lane_id();
  uint32_t       nr   = ~chunk_eq_mask(c, rc) & ((1u << valid) - 1u);
  const uint32_t cnt  = __popc(nr);
  const uint32_t incl = wave_incl_scan(cnt);
  uint32_t       idx  = incl - cnt;
  *(u32x4 *) (R.chunk + 16 * lane) = c;
  const uint32_t p16 = 16u * (uint32_t) lane;
  if (HIST_SKIP & 4) nr = 0;
  while (nr)
    { R.list[idx++] = (uint16_t) (p16 + (uint32_t) __builtin_ctz(nr));
      nr &= nr - 1u;
    }
  wave_sync();
  return wave_total(incl);
}

// run open at the end of a step of sv bytes with `total` tokens
__device__ __forceinline__ uint32_t run_after(const run_lds &R, uint32_t total, uint32_t sv, uint32_t C)
{ return total ? sv - 1u - (uint32_t) R.list[total - 1] : C + sv; }

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
          const int      valid = valid_of(pos, L);
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
  for (long long r = wid; r <= first; r += DX_WAVES_PER_BLK)      // (a dozen entries: 16 bytes per lane and load, as everywhere)
    { const uint32_t L    = a.len[r];
      const uint8_t *sub  = line_ptr(a, (uint64_t) r, L, 4);
      const bool     over = can_overread(a, sub, L);
      for (uint32_t base = 0; base < L; base += DX_STEP)
        { const uint32_t pos   = base + 16u * (uint32_t) lane;
          const u32x4    c     = fetch(sub, pos, L, over);
          const int      valid = valid_of(pos, L);
          for (int b = 0; b < valid; b++)
            atomicAdd(&s_hist[chunk_byte(c, b)], 1u);
        }
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
// ds_add_u32 is serviced in two groups of 32 lanes over 32 banks of 4 bytes, and lanes that hit
// one bank (same bin or not) take one cycle each.  So every bin is kept 32 times, copy = lane & 31
// in consecutive dwords: within a group every lane then owns a bank and no histogram update ever
// conflicts, whatever the symbol distribution.  That costs 32 x 4 B per bin, affordable for the
// symbols 0..127 (all of a .quiva file's printable QVs) and run lengths 0..63 of the six
// histograms with ONE workgroup of 16 waves per CU (80 KB); bytes >= 128 and runs >= 64 go to
// ordinary 256-bin tables.
#ifndef TICKET_BATCH
#define TICKET_BATCH 2u                  // entries a wave of the histogram / size pass draws at a time
#endif
#ifndef HIST_BLOCK
#define HIST_BLOCK   1024                // threads of a histogram workgroup ...
#endif
#ifndef HIST_PER_CU
#define HIST_PER_CU  1                   // ... and how many of them a CU holds (LDS: the tables below + 4 KB per wave)
#endif
#define HIST_NWAVE   (HIST_BLOCK / 64)
// copies of every fast bin (copy = lane & (copies - 1)): 32 = no two lanes of a 32-lane group ever on one bank.  The plain
// lines (ins, mrg; del and sub when they have no run character) take 16 updates per lane and step; the run-coded lines'
// symbol and run bins only see the tokens (a sixth of the symbols, two per lane and round), so fewer copies cost little.
#ifndef HC_PLAIN
#define HC_PLAIN     32
#endif
#ifndef HC_RSYM
#define HC_RSYM      8
#endif
#ifndef HC_RUN
#define HC_RUN       8
#endif
#define HSYM_FAST    128
#define HRUN_FAST    128                 // runs 0..126, and 127 for every longer one until hist_runs_step has looked again
#ifndef HIST_BARE_FETCH
#define HIST_BARE_FETCH 1
#endif

struct hist_lds
{ uint32_t plain[2][HSYM_FAST][HC_PLAIN];   // ins, mrg
  uint32_t rsym[2][HSYM_FAST][HC_RSYM];     // del, sub
  uint32_t run[2][HRUN_FAST][HC_RUN];       // deletion runs, substitution runs
  uint32_t slow[6][256];                    // bytes >= 128, runs >= 64, trailing runs, the run characters' own counts
};
#define HIST_W_PLAIN (2 * HSYM_FAST * HC_PLAIN)
#define HIST_W_RSYM  (2 * HSYM_FAST * HC_RSYM)
#define HIST_W_RUN   (2 * HRUN_FAST * HC_RUN)
#define HIST_FAST_WORDS (HIST_W_PLAIN + HIST_W_RSYM + HIST_W_RUN)

template <int COLS>
__device__ __forceinline__ void hist_plain_step(const u32x4 &c, int valid, bool full, uint32_t (*h)[COLS], uint32_t *slow)
{ const uint32_t col = (uint32_t) lane_id() & (COLS - 1);
  const bool wide = __any((int) ((c.x | c.y | c.z | c.w) & 0x80808080u));
  if (full && !wide)
    {
      #pragma unroll
      for (int b = 0; b < 16; b++)
        atomicAdd(&h[BYTE_OF(c, b)][col], 1u);
    }
  else
    for (int b = 0; b < valid; b++)
      { const uint32_t x = chunk_byte(c, b);
        if (x < HSYM_FAST) atomicAdd(&h[x][col], 1u);
        else               atomicAdd(&slow[x], 1u);
      }
}

// ---------------------------------------------------------------------------------------------
//  token hand-over: the histogram pass tokenises, the encoder only codes
// ---------------------------------------------------------------------------------------------
// Histogram_Runs (QV.c:709-724) and Encode_Run (QV.c:475-497) cut a run-coded line into the same
// (run length, symbol) tokens.  The kernels are bound by vector-instruction issue, not by HBM, so the
// cut is made once: while it histograms a run-coded line, k_qv_hist also stores its tokens, 16 bits
// each -- bits [0,2) the 2-bit code of the deletion tag under the symbol (Pack_Tag + Number_Read,
// QV.c:810-819; 0 for the substitution line), [2,9) the symbol, [9,16) the run in front of it -- into a
// slot of len/2 + 64 tokens per entry, with the token count and the run left open at the line's end.
// k_qv_encode_fast then walks dense token arrays instead of the text.  A run of 127 or more does not fit
// the token's 7 bits: the field then holds 127 and the run's length goes into the line's EXCEPTION LIST,
// (token index, run length) pairs of 32-bit words in token order, growing backwards from the end of the
// line's slot (record j at words [-2j - 2], [-2j - 1] from the slot's end).  At most 0.3 % of a line's
// symbols can be such tokens (density d: (1 - d) d^127), so the list is short and searched by bisection.
// An entry whose line has a symbol >= 128 or more tokens (and exceptions) than its slot holds is marked
// unusable and encoded from the text by the generic kernel, as is everything when the tokens were made
// for another batch or scan state.
#define TOK_RUN_MAX  127u                      // run field 127: the run's length is in the exception list
#define TOK_BAD      0x80000000u               // info word: the stream's tokens are unusable
#define TOK_INFO     8u                        // info words per entry: tokens of del | TOK_BAD, of sub | TOK_BAD, run open at the end
                                               // of del, of sub, exceptions of del, of sub, "no counters" (hist_wave_own), 1 spare
#define TOK_XMARGIN  40u                       // token slots kept free for the exceptions one step can add (<= 9 runs of >= 127 in 1 KiB)

struct tok_sink
{ uint16_t       *del, *sub;                   // NULL del: no tokens wanted
  const uint64_t *off;                         // n + 1 slot offsets (tokens)
  uint32_t       *info;                        // n x 4
  unsigned long long *unusable;                // count of entries with an unusable stream ...
  uint32_t       *list;                        // ... and their indices in the batch (any order)
};

// tokens an entry's slot holds: the share fr8 / 256 of its symbols (see k_qv_density: what the batch's run density asks for;
// 128 = half, the fixed share of rounds 2 and 3), 64 on top, the margin of the exception records
__host__ __device__ __forceinline__ uint32_t tok_room(uint32_t L, uint32_t fr8)
{ return (((uint32_t) (((uint64_t) L * fr8) >> 8) + 64u + TOK_XMARGIN) + 7u) & ~7u; }

// run length of exception token `idx` of a line: bisection over its nx records (ascending token index)
__device__ __forceinline__ uint32_t tok_exception_at(const uint32_t *slot_end, uint32_t nx, uint32_t idx)      // which record
{ uint32_t lo = 0, hi = nx;
  while (lo < hi)
    { const uint32_t mid = (lo + hi) >> 1;
      if (*(slot_end - 2 * (int) mid - 2) < idx) lo = mid + 1; else hi = mid;
    }
  return lo;
}
__device__ __forceinline__ uint32_t tok_exception(const uint32_t *slot_end, uint32_t nx, uint32_t idx)
{ return *(slot_end - 2 * (int) tok_exception_at(slot_end, nx, idx) - 1); }

// does this entry have to be encoded from the text (generic kernel)?  info: the n x 4 words k_qv_hist left
__device__ __forceinline__ bool tok_unusable(const uint32_t *info, uint64_t r, int delChar, int subChar)
{ return (delChar >= 0 && (info[TOK_INFO * r] & TOK_BAD)) || (subChar >= 0 && (info[TOK_INFO * r + 1] & TOK_BAD)); }

// 16-bit mask: bit b set iff byte b of the chunk differs from c.  Per word: the nonzero-byte flags of w ^ cccc at bits 7, 15,
// 23, 31 (exact per byte: the add cannot carry out of a byte), lined up at bits 28..31 by one multiplication (the partial
// products fall on distinct bits: 7 14 21 28 | 15 22 29 36 | 23 30 37 44 | 31 38 45 52), so the product's top byte is the
// word's nibble << 4 over four zero bits; two byte permutes and a shift put the four nibbles side by side.
__device__ __forceinline__ uint32_t chunk_ne_mask(const u32x4 &v, uint32_t c4 /* c * 0x01010101 */)
{ uint32_t p[4];
  #pragma unroll
  for (int i = 0; i < 4; i++)
    { const uint32_t w = chunk_word(v, i) ^ c4;
      p[i] = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u) * 0x00204081u;
    }
  const uint32_t even = __builtin_amdgcn_perm(p[2], p[0], 0x0c0c0703u);     // n0 << 4 | n2 << 12
  const uint32_t odd  = __builtin_amdgcn_perm(p[3], p[1], 0x0c0c0703u);     // n1 << 4 | n3 << 12
  return (even >> 4) | odd;
}

// byte b (0..15) of a chunk by three byte permutes, no compare and no select: selA = (b & 7) | 0x0c0c0c00 picks the byte
// out of each half of the chunk (the other result bytes zero), selB = ((b >> 1) & 4) | 0x0c0c0c00 picks the half
__device__ __forceinline__ uint32_t chunk_byte_sel(const u32x4 &v, uint32_t selA, uint32_t selB)
{ return __builtin_amdgcn_perm(__builtin_amdgcn_perm(v.w, v.z, selA), __builtin_amdgcn_perm(v.y, v.x, selA), selB); }

// the value of lane - 1 (lane 0: 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)
{ return (uint32_t) __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true); }      // wave_shr:1

// One step of a run-coded line: its non-run symbols and the run before each (QV.c:709-724); the run character itself is
// counted with popcounts instead of LDS atomics (it is 80-85 % of the line).
//
// Every lane walks the non-run symbols of ITS OWN 16 bytes, everything in registers: the symbol (and the deletion tag
// under it) by byte permutes of the chunk, the run in front of it as the distance to the lane's previous one -- for a
// lane's first, to the end of the last token of any lane before it (an exclusive running maximum over the lanes, the run
// open at the step's start folded in) -- one conflict-poor ds_add for the symbol, one for the run, the finished 16-bit
// token into the wave's LDS list at the place a prefix sum of the lanes' counts gives it; the list then leaves in 8-byte
// pieces.  No look-up hangs on another (round 3: position list -> symbol and tag -> tag code, three LDS round trips per
// 128 tokens at four waves per SIMD; and 150 KB of LDS: chunk copies, position lists, 32 copies of every bin).
//
// The fast bins take symbol & 127 and min(run, 127); a step in which some token has a symbol >= 127 or a run >= 127
// (never, in a .quiva file's printable QVs at usual run densities) is gone over once more: those tokens' counts move to
// the 256-bin tables, their runs get exception records, a symbol >= 128 makes the entry's tokens unusable.
// inc: 1 when the run histogram takes part (entries from del_first / sub_first on, QV.c:1003, 1016), else 0.
// `tok` != NULL: the step's tokens are stored at tok[ntok ...] (t: this step's deletion tags where TAGS).
// A step's tokens on their way out: lane l holds tokens 4 l .. 4 l + 3 of the n that belong at tok[at ...].  They are
// stored at the head of the NEXT step, in front of its requests for chunks: a step ends with a wait for the chunks
// requested at its head, vmcnt counts loads and stores alike, in order, and a store issued in the middle of the step is
// far from complete at its end (k_qv_hist without the stores: - 1.5 ms of 12) -- issued at the head, it has the whole
// step, like the chunks behind it.
struct tok_pend { uint64_t v; uint32_t n, at; };

__device__ __forceinline__ void tok_flush(tok_pend &p, uint16_t *tok)
{ if (4u * (uint32_t) lane_id() < p.n)                           // (up to three slots past the last token are written too: inside
    *(u64_u *) (tok + p.at + 4u * (uint32_t) lane_id()) = p.v;   //  TOK_XMARGIN, and the next step's tokens go over them)
  p.n = 0;
}

// Number_Read's letter -> 2-bit code (DB.c:393-416: a c g t, either case; anything else 0) without a table: the letters
// fold to 'A' + {0, 2, 6, 19}, and a 64-bit constant holds the code of 'A' + i at bits 2 i, 2 i + 1 for i < 32
__device__ __forceinline__ uint32_t tag_code(uint32_t letter)
{ const uint32_t i = min((letter & 0xdfu) - 0x41u, 31u);
  return (uint32_t) (((1ull << 4) | (2ull << 12) | (3ull << 38)) >> (2u * i)) & 3u;
}

// LDS of one wave.  !FAST (everything decided at run time): a copy of the step's chunk and of its deletion tags (a lane
// reads back the bytes under ITS OWN tokens: one address and two byte reads per token) and the step's token list; the
// histograms are the workgroup's (hist_lds).  FAST: the wave's OWN histograms of the entry it is working on --
//   pp: the two plain lines, every counter kept 4 times (copy = lane & 3): [line][symbol][copy];
//   ps, pr: the token lines: symbols of del, of sub (two copies), runs of del, of sub, 128 counters each --
// which at the entry's end go to the workgroup's 256-bin tables AND, as 768 counters of 16 bits, to memory: with the code
// lengths known (after the tables have been built) an entry's five segment sizes are then a dot product away
// (k_qv_sizes_hist), the encoder writes every record where it belongs, and the scratch slots and their compaction -- 30 of
// the 133 GB a step moved in round 3 -- are gone.
struct hist_wave_lds
{ uint8_t  chunk[DX_STEP], tags[DX_STEP];
  uint16_t list[DX_STEP];
};
// copies of every counter of the two plain lines (copy = lane & (copies - 1); 4 or 8): lanes of a 32-lane group that hold the same
// symbol add to the same address, one LDS cycle each -- SQ_LDS_BANK_CONFLICT is 64 % of the kernel's LDS cycles, the waves stand
// a fifth of their time at the LDS's door (SQ_WAIT_INST_LDS; profiles/r06_hist_lds.txt)
#ifndef PPC_INS
#define PPC_INS 8
#endif
#ifndef PPC_MRG
#define PPC_MRG 4
#endif
// tokens a wave's list holds: a step's (1024 symbols) all, or half that -- a step with more (a run density below a half; the
// instance with the wave's own token counters is launched for sampled densities from ~0.72 up) then leaves no tokens, and its
// entry goes to the text-reading kernels like every other entry whose tokens cannot be used
#ifndef HIST_LIST
#define HIST_LIST 512
#endif
struct hist_wave_own
{ uint32_t pi[128][PPC_INS], pm[128][PPC_MRG];          // (16 bits packed two to a word, eight copies: the same time, 3 instructions more per byte)
#ifndef PSC
#define PSC 1
#endif
  uint32_t ps[2][128][PSC];                             // token symbols of del, of sub: PSC copies (copy = lane & (PSC - 1))
#ifndef PRC
#define PRC 1
#endif
  uint32_t pr[2][128][PRC];                             // runs of del, of sub (PRC copies, copy = lane & (PRC - 1))
  uint16_t list[HIST_LIST];
};
#define EH_WORDS 384u                                   // an entry's counters in memory: 768 x 16 bits: ins, mrg, del, sub, del runs, sub runs

// OWN: the fast bins are the wave's own (the symbols' in two copies, the runs' in one; the run bins count whatever `inc` says
// for the workgroup's tables -- the entry's sizes need every token); LDSX: symbol and tag are read back from the LDS copy of
// the chunk (else they come out of the chunk in registers: no room for the copies beside a wave's own counters).
template <bool TAGS, bool OWN, bool LDSX>
__device__ __forceinline__ void hist_runs_step(hist_wave_lds *W, uint16_t *wlist, const u32x4 &c, const u32x4 &t, uint32_t vmask, uint32_t sv, uint32_t rc4,
                                               uint32_t &C, uint32_t &nrun, uint32_t *hs, uint32_t *slow_s,
                                               uint32_t *hr, uint32_t *slow_r, uint32_t inc,
                                               uint16_t *tok, uint32_t &ntok, uint32_t room, uint32_t &bad, uint32_t &nexc,
                                               tok_pend &pend_out)
{ constexpr uint32_t HCS = OWN ? (uint32_t) PSC : HC_RSYM, HCR = OWN ? (uint32_t) PRC : HC_RUN;
  const uint32_t lane  = (uint32_t) lane_id();
  const uint32_t nr0   = chunk_ne_mask(c, rc4) & vmask;
  const uint32_t cnt   = __popc(nr0);
  const uint32_t incl  = wave_incl_scan(cnt);
  const uint32_t total = wave_total(incl);
  nrun += sv - total;                                            // wave-uniform
  if (LDSX)
    { *(u32x4 *) (W->chunk + 16u * lane) = c;
      if (TAGS) *(u32x4 *) (W->tags + 16u * lane) = t;
    }
  // room: what the slot holds less TOK_XMARGIN (0: no tokens wanted)
  const bool fits = OWN ? total <= (uint32_t) HIST_LIST : true;  // (the list of the instances with the wave's own counters may be the shorter one)
  const bool emit = !(HIST_SKIP & 8) && !bad && fits && ntok + total + 4u * nexc <= room;
  if (!emit) bad = 1;                                            // more tokens (and exceptions) than the slot holds, or than the list
  // where the last token of this lane ends (position + 1), the run open at the step's start added; 0: the lane has none
  const uint32_t tend  = cnt ? 16u * lane + 32u - (uint32_t) __clz(nr0) + C : 0u;
  const uint32_t imax  = wave_incl_max(tend);
  const uint32_t gmax  = __builtin_amdgcn_readlane(imax, 63);
  const uint32_t s0    = wave_shr1(imax) - (16u * lane + C);     // the run in front of the lane's token at byte b: b - s
  const uint32_t li0   = incl - cnt;                             // the lane's first token in the step's list
  uint32_t nr = (HIST_SKIP & 1) ? 0u : nr0, s = s0, acc = 0;
  uint16_t      *lp = wlist + (fits ? li0 : 0u);                 // (a step that does not fit: every lane's tokens over the list's first sixteen)
  const uint8_t *cp = LDSX ? W->chunk + 16u * lane : (const uint8_t *) NULL;
  const uint32_t pinc = OWN ? 1u : inc;                          // what the fast run bins count
  while (nr)
    { const uint32_t b   = (uint32_t) __builtin_ctz(nr);
      nr &= nr - 1u;
      uint32_t x, tg = 0;
      if (!LDSX)
        { const uint32_t selA = (b & 7u) | 0x0c0c0c00u, selB = ((b >> 1) & 4u) | 0x0c0c0c00u;
          x = chunk_byte_sel(c, selA, selB);
          if (TAGS) tg = chunk_byte_sel(t, selA, selB);
        }
      else
        { x = cp[b];
          if (TAGS) tg = cp[b + DX_STEP];                        // (both reads in front of the atomics: one wait for the two)
        }
      const uint32_t run = b - s;
      const uint32_t r7  = run < TOK_RUN_MAX ? run : TOK_RUN_MAX;
      s   = b + 1u;
      acc = max(acc, max(run, x));
#if HIST_TRIM
      // the symbol shifted once, for the token and for its counter's address (a counter is HCS words: (x & 127) * 4 HCS bytes)
      const uint32_t x4 = x << 2;
      atomicAdd((uint32_t *) ((uint8_t *) hs + (x4 & (4u * (HSYM_FAST - 1))) * HCS), 1u);
      atomicAdd(&hr[r7 * HCR], pinc);
      uint32_t tk = (r7 << 9) | x4;
      if (TAGS) tk |= tag_code(tg);
#else
      atomicAdd(&hs[(x & (HSYM_FAST - 1)) * HCS], 1u);
      atomicAdd(&hr[r7 * HCR], pinc);
      uint32_t tk = x | (r7 << 7);
      if (TAGS) tk = (tk << 2) | tag_code(tg);
      else      tk <<= 2;
#endif
      *lp++ = (uint16_t) tk;
    }
  uint32_t odd = 0;
  if (__any((int) (acc >= TOK_RUN_MAX)))                        // rare (see above): once more over the step
    { uint32_t m = nr0, ne = 0;
      s = s0;
      while (m)
        { const uint32_t b   = (uint32_t) __builtin_ctz(m);
          m &= m - 1u;
          const uint32_t x   = LDSX ? (uint32_t) cp[b] : chunk_byte(c, (int) b);
          const uint32_t run = b - s;
          s = b + 1u;
          if (run >= TOK_RUN_MAX)
            { atomicSub(&hr[TOK_RUN_MAX * HCR], pinc);
              atomicAdd(&slow_r[run > 255u ? 255u : run], inc);                                   // QV.c:717-720
              ne++;
            }
          if (x >= HSYM_FAST)
            { atomicSub(&hs[(x & (HSYM_FAST - 1)) * HCS], 1u);
              atomicAdd(&slow_s[x], 1u);
              odd = 1;
            }
        }
      if (emit && __any((int) ne))                              // exception records, in token order
        { const uint32_t ince = wave_incl_scan(ne);
          uint32_t  j    = nexc + ince - ne, i = ntok + li0;
          uint32_t *xend = (uint32_t *) (tok + room + TOK_XMARGIN);
          m = nr0; s = s0;
          while (m)
            { const uint32_t b   = (uint32_t) __builtin_ctz(m);
              m &= m - 1u;
              const uint32_t run = b - s;
              s = b + 1u;
              if (run >= TOK_RUN_MAX)
                { *(xend - 2 * (int) j - 2) = i;
                  *(xend - 2 * (int) j - 1) = run;
                  j++;
                }
              i++;
            }
          nexc += wave_total(ince);
        }
      else if (OWN && __any((int) ne))                          // (no record of these runs: the entry's sizes cannot be had from its counters)
        bad = 1;
    }
  if (emit)
    { __builtin_amdgcn_wave_barrier();                           // the list is complete: a wave's LDS instructions execute in order,
      if (HIST_SKIP & 32) { }                                    //   only the compiler has to be kept from moving them
      else if (total <= 256u)
        { pend_out.v = *(const uint64_t *) (wlist + 4u * lane); pend_out.n = total; pend_out.at = ntok; }
      else                                                       // (more than a quarter of the step's symbols: straight out)
        for (uint32_t i = 4u * lane; i < total; i += 256u)
          *(u64_u *) (tok + ntok + i) = *(const uint64_t *) (wlist + i);
      ntok += total;
      if (__any((int) odd)) bad = 1;
    }
  C = sv + C - gmax;                                             // total ? sv - (end of the last token) : C + sv
  __builtin_amdgcn_wave_barrier();
}

// the two lines that are always plain (ins, mrg), one test of "a whole step and no byte >= 128" for both
__device__ __forceinline__ void hist_plain_pair(const u32x4 &c2, const u32x4 &c3, int valid, bool full, hist_lds &H)
{ const uint32_t col = (uint32_t) lane_id() & (HC_PLAIN - 1);
  if (full && !__any((int) ((c2.x | c2.y | c2.z | c2.w | c3.x | c3.y | c3.z | c3.w) & 0x80808080u)))
    {
      #pragma unroll
      for (int b = 0; b < 16; b++)
        atomicAdd(&H.plain[0][BYTE_OF(c2, b)][col], 1u);
      #pragma unroll
      for (int b = 0; b < 16; b++)
        atomicAdd(&H.plain[1][BYTE_OF(c3, b)][col], 1u);
    }
  else
    { hist_plain_step<HC_PLAIN>(c2, valid, full, H.plain[0], H.slow[DX_INS]);
      hist_plain_step<HC_PLAIN>(c3, valid, full, H.plain[1], H.slow[DX_MRG]);
    }
}

// ... and into the wave's own counters (hist_wave_own.pp); a byte >= 128 goes to the workgroup's 256-bin table and is
// reported (wave-uniform: the entry's counters then do not hold all of it).  A line's last step is no special case: a lane's
// bytes behind the line's end are zero (fetch), the lane counts all sixteen and takes the surplus off bin 0 again -- the
// byte-by-byte loop that used to do the last step (a variable byte index, four branches a byte) cost more than a whole
// step's unrolled adds: an entry's last step took 2.2 times a full one, a 2 kb entry's two steps the time of three
// (k_qv_hist 13.0 -> 12.6 ms at 10 kb, 7.6 -> 6.5 at 2 kb, 10.8 -> 9.3 at 300 symbols; same box).
__device__ __forceinline__ uint32_t hist_plain_pair_own(const u32x4 &c2, const u32x4 &c3, int valid, bool full, uint32_t *pi, uint32_t *pm,
                                                        uint32_t *slow_ins, uint32_t *slow_mrg)
{ uint32_t *const q0 = pi + ((uint32_t) lane_id() & (PPC_INS - 1u)), *const q1 = pm + ((uint32_t) lane_id() & (PPC_MRG - 1u));
#define OWN_ADD(q, x) atomicAdd(&(q)[((q) == q0 ? (uint32_t) PPC_INS : (uint32_t) PPC_MRG) * (x)], 1u)
  if (!__any((int) ((c2.x | c2.y | c2.z | c2.w | c3.x | c3.y | c3.z | c3.w) & 0x80808080u)))
    { if (full || valid > 0)
        {
          #pragma unroll
          for (int b = 0; b < 16; b++)
            OWN_ADD(q0, BYTE_OF(c2, b));
          #pragma unroll
          for (int b = 0; b < 16; b++)
            OWN_ADD(q1, BYTE_OF(c3, b));
          if (!full && valid < 16)
            { atomicSub(&q0[0], (uint32_t) (16 - valid));
              atomicSub(&q1[0], (uint32_t) (16 - valid));
            }
        }
      return 0;
    }
  for (int b = 0; b < valid; b++)
    { const uint32_t x = chunk_byte(c2, b), y = chunk_byte(c3, b);
      if (x < 128u) OWN_ADD(q0, x);
      else          atomicAdd(&slow_ins[x], 1u);
      if (y < 128u) OWN_ADD(q1, y);
      else          atomicAdd(&slow_mrg[y], 1u);
    }
#undef OWN_ADD
  return 1;
}

// bin of the histogram g_hist[6*256] that LDS word k (of the fast tables, then the slow ones) counts
__device__ __forceinline__ uint32_t hist_bin_of(uint32_t k)
{ if (k < HIST_W_PLAIN)                                          // ins, mrg
    return (DX_INS + k / (HSYM_FAST * HC_PLAIN)) * 256u + (k / HC_PLAIN) % HSYM_FAST;
  k -= HIST_W_PLAIN;
  if (k < HIST_W_RSYM)                                           // del, sub
    return (k / (HSYM_FAST * HC_RSYM) ? DX_SUB : DX_DEL) * 256u + (k / HC_RSYM) % HSYM_FAST;
  k -= HIST_W_RSYM;
  if (k < HIST_W_RUN)
    return (4u + k / (HRUN_FAST * HC_RUN)) * 256u + (k / HC_RUN) % HRUN_FAST;
  return k - HIST_W_RUN;
}

// how many copies the fast bin holding LDS word k has (the words of one bin are consecutive)
__device__ __forceinline__ uint32_t hist_copies_of(uint32_t k)
{ return k < HIST_W_PLAIN ? HC_PLAIN : (k < HIST_W_PLAIN + HIST_W_RSYM ? HC_RSYM : HC_RUN); }

// ---------------------------------------------------------------------------------------------
//  an entry's words, asked for in one go and an entry ahead
// ---------------------------------------------------------------------------------------------
// What a wave has to know of an entry before it can ask for a byte of it (its length, where its text and its tokens lie, ...) sits
// in half a dozen arrays.  Asked for one after the other where they are needed, those words are a chain of dependent waits at every
// entry's start (see k_qv_encode_fast); instead lane k asks for word k -- ONE load instruction, the arrays' bases and strides from a
// small table in LDS -- and asks an entry ahead; each word is read where it is needed by v_readlane.
struct meta_row { const uint8_t *base; uint32_t stride, pad; };

__device__ __forceinline__ uint32_t meta_load(const meta_row *s_meta, uint32_t words, uint64_t r)
{ uint32_t v = 0;
  const uint32_t lane = (uint32_t) lane_id();
  if (lane < words)
    { const meta_row m = s_meta[lane];
      v = *(const uint32_t *) (m.base + r * m.stride);
    }
  return v;
}
#define META(m, k)   ((uint32_t) __builtin_amdgcn_readlane((int) (m), (int) (k)))
#define META64(m, k) ((uint64_t) META(m, k) | ((uint64_t) META(m, (k) + 1) << 32))

// ---------------------------------------------------------------------------------------------
//  the scan state on the device (dx_qv_scan)
// ---------------------------------------------------------------------------------------------
// dx_qv_prescan hands delChar / subChar to the host, which hands them to dx_qv_hist's kernels as arguments: a round trip,
// and a second one for the token total (what the slots need) and the slot share (which instance of k_qv_hist) -- 0.25 ms of
// a 25 ms step with nothing on the device.  dx_qv_scan leaves the state where it is made: k_scan_state folds what the two
// prescan kernels found into this record, k_qv_density / k_tok_rooms / k_qv_hist read it there, and the host sees it with the
// histograms, in the one copy it waits for.  What the host has to decide before it knows -- which instance of k_qv_hist to
// launch, whether the token buffers it has are large enough -- it GUESSES from the context's last scan, and the device
// checks: a wrong guess leaves SCAN_MISS here, k_qv_hist returns at once, and the host goes the old way (dx_qv_hist) with
// the state it now knows.  No result is carried over from one batch to the next, only the guess.
struct scan_dev
{ int32_t   delChar, subChar;
  long long del_first, sub_first;
  uint32_t  inst;                                        // the instance of k_qv_hist this batch wants (SCAN_*), | SCAN_MISS
  uint32_t  share8;                                      // k_tok_rooms' slot share
};
#define SCAN_G     0u                                    // k_qv_hist<false, false> with tokens
#define SCAN_A     1u                                    // <true, true>
#define SCAN_B     2u                                    // <true, false>
#define SCAN_NOTOK 3u                                    // no run character at all: no tokens (never guessed)
#define SCAN_MISS  0x100u

// FAST: tokens wanted and both run characters known -- the usual launch: both lines are run-coded in every entry, the tag
// line travels with them, none of that is asked per entry and step, and the histograms are first the wave's own, per
// entry (hist_wave_own); !FAST: everything decided at run time, the workgroup's replicated bins.
// OWNTOK (with FAST): the token lines' counters are the wave's own too -- at the usual run densities.  Where more than about a
// quarter of a run-coded line's symbols are tokens (k_qv_density), half of them have the same run in front (run 0) and the
// lanes of a wave queue up at one counter: those batches take the instance whose token lines count in the workgroup's
// replicated bins, and the size kernel reads their tokens instead of counters.
template <bool FAST, bool OWNTOK> struct hist_smem;
template <> struct hist_smem<false, false> { hist_lds H; hist_wave_lds W[HIST_NWAVE]; };
template <> struct hist_smem<true, true>   { uint32_t slow[6][256]; hist_wave_own W[HIST_NWAVE]; };
struct hist_wave_plain { uint32_t pi[128][PPC_INS], pm[128][PPC_MRG]; uint16_t list[DX_STEP]; };     // (the instance of the dense batches: a whole step's room)
template <> struct hist_smem<true, false>
{ uint32_t slow[6][256];
  uint32_t rsym[2][HSYM_FAST][HC_RSYM], run[2][HRUN_FAST][HC_RUN];
  hist_wave_plain W[HIST_NWAVE];
};

template <bool FAST, bool OWNTOK>
__global__ __launch_bounds__(HIST_BLOCK, (HIST_NWAVE * HIST_PER_CU + 3) / 4)
void k_qv_hist(qv_args a, uint64_t entry0, long long del_first, long long sub_first,
               unsigned long long *g_hist /* 6*256 */, unsigned long long *g_tot, uint32_t *ticket, tok_sink ts, uint32_t *eh /* n x EH_WORDS (FAST) */,
               const scan_dev *sd /* dx_qv_scan: the scan state, and the verdict on the host's guesses */, uint64_t tok_cap /* tokens the buffers hold */,
               const uint32_t *orig /* NULL, or: entry r of this batch is entry orig[r] of the file's batch (the long entries of a mixed batch) */)
{ __shared__ __attribute__((aligned(16))) hist_smem<FAST, OWNTOK> S;
  if (sd != NULL)
    { if (sd->inst != (FAST ? (OWNTOK ? SCAN_A : SCAN_B) : SCAN_G) || ts.off[a.n] > tok_cap)
        return;                                          // not the instance the batch wants, or slots beyond the buffers: the host sees it (dx_qv_scan)
      a.delChar = sd->delChar; a.subChar = sd->subChar;
      del_first = sd->del_first; sub_first = sd->sub_first;
    }
  typedef hist_smem<false, false> smem_g;
  typedef hist_smem<true, true>   smem_a;
  typedef hist_smem<true, false>  smem_b;
  const int      lane  = lane_id();
  const int      tid   = threadIdx.x;
  const int      wid   = tid >> 6;
  uint32_t (*const slow)[256] = !FAST ? ((smem_g *) (void *) &S)->H.slow : (OWNTOK ? ((smem_a *) (void *) &S)->slow : ((smem_b *) (void *) &S)->slow);
  hist_lds      *const Hp = FAST ? (hist_lds *) NULL : &((smem_g *) (void *) &S)->H;
  hist_wave_lds *const Wl = FAST ? (hist_wave_lds *) NULL : &((smem_g *) (void *) &S)->W[wid];
  hist_wave_own *const Wo = FAST && OWNTOK ? &((smem_a *) (void *) &S)->W[wid] : (hist_wave_own *) NULL;
  smem_b        *const Sb = FAST && !OWNTOK ? (smem_b *) (void *) &S : (smem_b *) NULL;
  uint32_t *const piw = !FAST ? (uint32_t *) NULL : (OWNTOK ? &Wo->pi[0][0] : &Sb->W[wid].pi[0][0]);
  uint32_t *const pmw = !FAST ? (uint32_t *) NULL : (OWNTOK ? &Wo->pm[0][0] : &Sb->W[wid].pm[0][0]);
  uint16_t      *const wlist = !FAST ? Wl->list : (OWNTOK ? Wo->list : Sb->W[wid].list);
  uint32_t *const words = (uint32_t *) (void *) &S;              // the whole of S as words (zeroed at the start)
  const uint32_t  nwords = sizeof(S) / 4;
  const bool      toks  = FAST || ts.del != NULL;
  // the run-coded lines' fast bins: this lane's copy of the workgroup's, or the wave's own
  constexpr bool OT = FAST && OWNTOK;
  uint32_t (*const rsymp)[HSYM_FAST][HC_RSYM] = FAST ? (OWNTOK ? (uint32_t (*)[HSYM_FAST][HC_RSYM]) NULL : Sb->rsym) : Hp->rsym;
  uint32_t (*const runp)[HRUN_FAST][HC_RUN]   = FAST ? (OWNTOK ? (uint32_t (*)[HRUN_FAST][HC_RUN]) NULL : Sb->run) : Hp->run;
  uint32_t *const hs0 = OT ? &Wo->ps[0][0][(uint32_t) lane & (PSC - 1)] : &rsymp[0][0][(uint32_t) lane & (HC_RSYM - 1)];
  uint32_t *const hs4 = OT ? &Wo->ps[1][0][(uint32_t) lane & (PSC - 1)] : &rsymp[1][0][(uint32_t) lane & (HC_RSYM - 1)];
  uint32_t *const hr0 = OT ? &Wo->pr[0][0][(uint32_t) lane & (PRC - 1)] : &runp[0][0][(uint32_t) lane & (HC_RUN - 1)];
  uint32_t *const hr4 = OT ? &Wo->pr[1][0][(uint32_t) lane & (PRC - 1)] : &runp[1][0][(uint32_t) lane & (HC_RUN - 1)];
  const uint32_t  rc0 = (uint32_t) (a.delChar & 0xff) * 0x01010101u, rc4 = (uint32_t) (a.subChar & 0xff) * 0x01010101u;

  // the entry's words (meta_load): 0 its length, 1 2 its text's offset, 3 .. 6 its token slot's start and end, 7 its index in the file's batch
#define HM_WORDS 8u
  __shared__ __attribute__((aligned(16))) meta_row s_meta[HM_WORDS];
  if (tid < (int) HM_WORDS)
    { meta_row m;
      if      (tid == 0) { m.base = (const uint8_t *) a.len;                   m.stride = 4u; }
      else if (tid < 3)  { m.base = (const uint8_t *) a.off + 4 * (tid - 1);   m.stride = 8u; }
      else if (tid < 7)  { m.base = (const uint8_t *) ts.off + 4 * (tid - 3);  m.stride = 8u; }
      else               { m.base = (const uint8_t *) orig;                    m.stride = 4u; }
      if ((tid >= 3 && tid < 7 && !(FAST || ts.del != NULL)) || (tid == 7 && orig == NULL))
        { m.base = (const uint8_t *) a.len; m.stride = 0u; }      // (no such array in this launch: any word)
      m.pad = 0;
      s_meta[tid] = m;
    }
  for (uint32_t k = tid; k < nwords; k += HIST_BLOCK) words[k] = 0;
  __syncthreads();

  uint64_t tot = 0, since = 0;
  uint32_t mpre = 0;                                   // the words of entry mfor
  uint64_t mfor = ~0ull;
  for (uint64_t r0 = next_unit(ticket, a.units), nxt; r0 < a.n; r0 = nxt)
  { nxt = next_unit(ticket, a.units);                  // drawn early: the atomic's latency hides behind these entries
    for (uint64_t r = r0; r < r0 + a.units && r < a.n; r++)
    { const uint32_t mw = mfor == r ? mpre : meta_load(s_meta, HM_WORDS, r);      // (not asked for ahead: a wave's first entry)
      { const uint64_t rn = r + 1 < r0 + a.units && r + 1 < a.n ? r + 1 : nxt;
        mfor = rn;
        if (rn < a.n) mpre = meta_load(s_meta, HM_WORDS, rn);
      }
      const uint32_t  L = META(mw, 0);
      const long long g = (long long) (entry0 + (orig ? (uint64_t) META(mw, 7) : r));
      const bool drun = FAST || (a.delChar >= 0 && (toks || g >= del_first));      // tokenised (and, from del_first on, run-histogrammed)
      const bool srun = FAST || (a.subChar >= 0 && (toks || g >= sub_first));
      const uint32_t dinc = (a.delChar >= 0 && g >= del_first) ? 1u : 0u, sinc = (a.subChar >= 0 && g >= sub_first) ? 1u : 0u;
      const uint8_t *p0 = a.text + META64(mw, 1);     // line_ptr(a, r, L, 0 .. 4)
      const uint64_t pitch = (uint64_t) L + a.pad;
      const uint8_t *p1 = p0 + pitch, *p2 = p1 + pitch, *p3 = p2 + pitch, *p4 = p3 + pitch;
      const bool over = can_overread(a, p4, L);       // p4 is the last line of the entry
      uint32_t C0 = 0, C4 = 0, n0 = 0, n4 = 0;
      uint16_t *tk0 = NULL, *tk4 = NULL;
      uint32_t  nt0 = 0, nt4 = 0, room = 0, bad0 = 0, bad4 = 0, nx0 = 0, nx4 = 0;
      tok_pend  pd0 = { 0ull, 0u, 0u }, pd4 = { 0ull, 0u, 0u };
      if (toks)
        { const uint64_t t0 = META64(mw, 3);
          room = (uint32_t) (META64(mw, 5) - t0) - TOK_XMARGIN;
          if (drun) tk0 = ts.del + t0;
          if (srun) tk4 = ts.sub + t0;
        }
      const bool tags = FAST || tk0 != NULL;

      uint32_t pos = 16u * lane;
      u32x4 c0 = fetch(p0, pos, L, over), c4 = fetch(p4, pos, L, over);
      u32x4 t1 = c0;
      if (tags) t1 = fetch(p1, pos, L, over);          // the deletion tags travel with the step's other chunks
      for (uint32_t base = 0; base < L; base += DX_STEP)
        { const uint32_t np = pos + DX_STEP;
          // The last step's tokens leave first (tok_pend), then come the requests: the two plain lines' chunks of THIS step
          // (they are looked at last, behind the run-coded lines: most of a step away; a register set of their own for the
          // next step's would cost the kernel its fourth wave per SIMD) and the run-coded lines' chunks of the NEXT step.
          // The step ends with the wait for those -- the copies c = d below; with token stores in the loop every wait the
          // compiler places is vmcnt(0) -- and by then they have had the whole step.  The explicit wait tells the compiler
          // that nothing older is outstanding here (true but for an entry's first step, whose chunks were requested just
          // before the loop): without it it waits for "everything" at the first use of a chunk in the step -- i.e. for the
          // requests made a moment ago.
          __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
          if (toks)
            { tok_flush(pd0, tk0);
              tok_flush(pd4, tk4);
            }
          u32x4 c2, c3, d0, d4, u1 = c0;               // (without tags u1 is never looked at; a copy of d0 would wait for d0's load)
          if (HIST_BARE_FETCH && base + DX_STEP <= L)  // this whole step is inside the lines: bare loads (see fetch_step)
            { c2 = HIST_LOAD(p2 + pos); c3 = HIST_LOAD(p3 + pos); }
          else
            { c2 = fetch(p2, pos, L, over); c3 = fetch(p3, pos, L, over); }
          if (HIST_BARE_FETCH && base + 2u * DX_STEP <= L)   // ... and so is the whole next step
            { d0 = HIST_LOAD(p0 + np); d4 = HIST_LOAD(p4 + np);
              if (tags) u1 = HIST_LOAD(p1 + np);
            }
          else
            { d0 = fetch(p0, np, L, over); d4 = fetch(p4, np, L, over);
              if (tags) u1 = fetch(p1, np, L, over);
            }
          const uint32_t sv    = L - base >= DX_STEP ? DX_STEP : L - base;
          const bool     full  = sv == DX_STEP;
          const int      valid = valid_of(pos, L);
          const uint32_t vmask = (1u << valid) - 1u;
          if (HIST_SKIP & 16) { }
          else if (drun) hist_runs_step<true, OT, !FAST>(Wl, wlist, c0, t1, vmask, sv, rc0, C0, n0, hs0, slow[DX_DEL], hr0, slow[DX_DRUN],
                                                    dinc, tk0, nt0, tk0 != NULL ? room : 0u, bad0, nx0, pd0);   // (no tags wanted: t1 = c0, never stored)
          else      hist_plain_step<HC_RSYM>(c0, valid, full, Hp->rsym[0], slow[DX_DEL]);
          if (HIST_SKIP & 16) { }
          else if (srun) hist_runs_step<false, OT, !FAST>(Wl, wlist, c4, c4, vmask, sv, rc4, C4, n4, hs4, slow[DX_SUB], hr4, slow[DX_SRUN],
                                                     sinc, tk4, nt4, tk4 != NULL ? room : 0u, bad4, nx4, pd4);
          else      hist_plain_step<HC_RSYM>(c4, valid, full, Hp->rsym[1], slow[DX_SUB]);
          if (HIST_SKIP & 2) { }
          else if (FAST) bad0 |= hist_plain_pair_own(c2, c3, valid, full, piw, pmw, slow[DX_INS], slow[DX_MRG]);
          else           hist_plain_pair(c2, c3, valid, full, *Hp);
          c0 = d0; c4 = d4; t1 = u1;
          pos = np;
        }
      tok_flush(pd0, tk0);
      tok_flush(pd4, tk4);
      if (FAST)
        { // The entry's counters: to the workgroup's tables (the run counters only from del_first / sub_first on) and, 16 bits
          // each, to memory.  Lane l holds the plain lines' symbols 2 l, 2 l + 1 (four copies each) and the same two bins of each of
          // the four token tables.  (An entry of 2^16 symbols and more has no use for its 16-bit counters: it is on the list of the
          // text-reading kernels; the workgroup's tables get the full counts.)
          __builtin_amdgcn_wave_barrier();
          uint32_t *e32 = eh + (uint64_t) r * EH_WORDS;
          const u32x4 zero = { 0u, 0u, 0u, 0u };
          #pragma unroll
          for (int k = 0; k < 2; k++)
            { const int cp = k == 0 ? PPC_INS : PPC_MRG;                              // copies a counter has (2, 4 or 8)
              u32x4 *w = (u32x4 *) ((k == 0 ? piw : pmw) + 2 * lane * cp);            // this lane's two counters: 2 cp words side by side
              uint32_t ce = 0, co = 0;
              if (cp == 2)
                { const u32x4 v = w[0];
                  w[0] = zero;
                  ce = v.x + v.y; co = v.z + v.w;
                }
              else
                {
                  #pragma unroll
                  for (int j = 0; j < cp / 4; j++)
                    { const u32x4 v0 = w[j], v1 = w[cp / 4 + j];
                      w[j] = zero; w[cp / 4 + j] = zero;
                      ce += v0.x + v0.y + v0.z + v0.w; co += v1.x + v1.y + v1.z + v1.w;
                    }
                }
              e32[64 * k + lane] = (ce & 0xffffu) | (co << 16);
              if (ce | co)
                { atomicAdd(&slow[DX_INS + k][2 * lane], ce);
                  atomicAdd(&slow[DX_INS + k][2 * lane + 1], co);
                }
            }
          #pragma unroll
          for (int k = 0; k < (OT ? 4 : 0); k++)
            { uint32_t v0, v1;
              if (k < 2)
                { uint32_t *w = &Wo->ps[k][2 * lane][0];
                  v0 = 0; v1 = 0;
                  #pragma unroll
                  for (int j = 0; j < PSC; j++)
                    { v0 += w[j]; v1 += w[PSC + j];
                      w[j] = 0; w[PSC + j] = 0;
                    }
                }
              else
                { uint32_t *w = &Wo->pr[k - 2][2 * lane][0];
                  v0 = 0; v1 = 0;
                  #pragma unroll
                  for (int j = 0; j < PRC; j++)
                    { v0 += w[j]; v1 += w[PRC + j];
                      w[j] = 0; w[PRC + j] = 0;
                    }
                }
              e32[128 + 64 * k + lane] = (v0 & 0xffffu) | (v1 << 16);
              const int      tab = k == 0 ? DX_DEL : (k == 1 ? DX_SUB : (k == 2 ? DX_DRUN : DX_SRUN));
              const uint32_t on  = k == 2 ? dinc : (k == 3 ? sinc : 1u);
              if (on && (v0 | v1))
                { atomicAdd(&slow[tab][2 * lane], v0);
                  atomicAdd(&slow[tab][2 * lane + 1], v1);
                }
            }
          __builtin_amdgcn_wave_barrier();
        }
      if (drun)                                        // trailing run + the run character's own count
        { if (dinc && C0 > 0 && lane == 0) atomicAdd(&slow[DX_DRUN][C0 > 255u ? 255u : C0], 1u);
          if (lane == 0 && n0) atomicAdd(&slow[DX_DEL][a.delChar], n0);
        }
      if (srun)
        { if (sinc && C4 > 0 && lane == 0) atomicAdd(&slow[DX_SRUN][C4 > 255u ? 255u : C4], 1u);
          if (lane == 0 && n4) atomicAdd(&slow[DX_SUB][a.subChar], n4);
        }
      if (toks)
        { if (FAST && (bad0 || bad4)) { bad0 = 1; bad4 = 1; }      // (one line's counters not to be had: the whole entry by the text-reading kernels)
          if (lane == 0)
            { uint32_t *w = ts.info + TOK_INFO * r;
              w[0] = nt0 | ((bad0 || !drun) ? TOK_BAD : 0u);
              w[1] = nt4 | ((bad4 || !srun) ? TOK_BAD : 0u);
              w[2] = C0;
              w[3] = C4;
              w[4] = nx0;
              w[5] = nx4;
              w[6] = (FAST && L > 65535u ? 1u : 0u) |  // the entry's 16-bit counters do not hold it: its sizes from its tokens and text
                     (FAST && !OWNTOK ? 2u : 0u);      // no counters of the token lines: those lines' sizes from their tokens
            }
          if (((drun && bad0) || (srun && bad4)) && lane == 0)      // rare: the generic kernel works through this list
            ts.list[atomicAdd(ts.unusable, 1ull)] = (uint32_t) r;
        }
      tot   += L;
      since += L;
      if (since >= (1ull << 26))                       // keep the 32-bit LDS bins far from overflow
        { if (FAST)
            { for (uint32_t k = lane; k < 6u * 256u; k += 64)
                { const uint32_t v = atomicExch(&slow[0][k], 0u);
                  if (v) atomicAdd(&g_hist[k], (unsigned long long) v);
                }
              if (!OWNTOK)
                { for (uint32_t k = lane; k < 2u * HSYM_FAST * HC_RSYM; k += 64)
                    { const uint32_t v = atomicExch(&(&Sb->rsym[0][0][0])[k], 0u);
                      if (v) atomicAdd(&g_hist[(k / (HSYM_FAST * HC_RSYM) ? DX_SUB : DX_DEL) * 256u + (k / HC_RSYM) % HSYM_FAST], (unsigned long long) v);
                    }
                  for (uint32_t k = lane; k < 2u * HRUN_FAST * HC_RUN; k += 64)
                    { const uint32_t v = atomicExch(&(&Sb->run[0][0][0])[k], 0u);
                      if (v) atomicAdd(&g_hist[(4u + k / (HRUN_FAST * HC_RUN)) * 256u + (k / HC_RUN) % HRUN_FAST], (unsigned long long) v);
                    }
                }
            }
          else
            for (uint32_t k = lane; k < sizeof(hist_lds) / 4; k += 64)
              { const uint32_t v = atomicExch(&words[k], 0u);
                if (v) atomicAdd(&g_hist[hist_bin_of(k)], (unsigned long long) v);
              }
          since = 0;
        }
    }
  }
  if (lane == 0 && tot)
    atomicAdd(g_tot, (unsigned long long) tot);
  __syncthreads();
  // fold the copies of every fast bin (rotated start: the lanes of a wave read distinct banks); one thread per 8 words
  // (8 divides every copy count), several threads per bin where it has more copies
  if (FAST && !OWNTOK)                                   // the token lines' replicated bins of this instance
    { for (uint32_t k = tid; k < 2u * HSYM_FAST; k += HIST_BLOCK)
        { uint32_t v = 0;
          for (uint32_t j = 0; j < HC_RSYM; j++) v += (&Sb->rsym[0][0][0])[k * HC_RSYM + j];
          if (v) atomicAdd(&g_hist[(k / HSYM_FAST ? DX_SUB : DX_DEL) * 256u + k % HSYM_FAST], (unsigned long long) v);
        }
      for (uint32_t k = tid; k < 2u * HRUN_FAST; k += HIST_BLOCK)
        { uint32_t v = 0;
          for (uint32_t j = 0; j < HC_RUN; j++) v += (&Sb->run[0][0][0])[k * HC_RUN + j];
          if (v) atomicAdd(&g_hist[(4u + k / HRUN_FAST) * 256u + k % HRUN_FAST], (unsigned long long) v);
        }
    }
  if (!FAST)
    for (uint32_t w8 = tid; w8 < HIST_FAST_WORDS / 8u; w8 += HIST_BLOCK)
      { uint32_t v = 0;
        for (uint32_t j = 0; j < 8u; j++)
          v += words[w8 * 8u + ((j + (uint32_t) lane) & 7u)];
        if (v) atomicAdd(&g_hist[hist_bin_of(w8 * 8u)], (unsigned long long) v);
      }
  for (uint32_t k = tid; k < 6 * 256; k += HIST_BLOCK)
    { const uint32_t v = slow[0][k];
      if (v) atomicAdd(&g_hist[k], (unsigned long long) v);
    }
}

__global__ void k_scan_state(const unsigned long long *key, const long long *sub, dx_qv_params in, int want_del, int want_sub, scan_dev *sd)
{ if (threadIdx.x != 0 || blockIdx.x != 0) return;
  scan_dev d;
  d.delChar = in.delChar; d.subChar = in.subChar; d.del_first = in.del_first; d.sub_first = in.sub_first;
  if (want_del && *key != ~0ull)
    { d.delChar   = (int32_t) (*key & 0xff);
      d.del_first = (long long) (*key >> 8);
    }
  if (want_sub && sub[0] >= 0)
    { d.subChar   = (int32_t) sub[1];
      d.sub_first = sub[0];
    }
  d.inst = 0; d.share8 = 0;
  *sd = d;
}

// How many of a run-coded line's symbols are tokens (not the run character)?  Counted on a sample -- up to 1024 entries spread
// evenly over the batch, the first 4 KiB of their deletion and substitution lines -- so that the token slots can be sized for
// the batch at hand: half a slot per symbol (rounds 2 and 3) sends every entry of a batch with run density 0.3 to the
// text-reading encoder and wastes two thirds of the slots at 0.85.  cnt[0..3]: symbols seen / tokens among them, del then sub.
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_density(qv_args a, uint64_t stride, unsigned long long *cnt, const scan_dev *sd /* NULL: the run characters are a's */)
{ if (sd != NULL) { a.delChar = sd->delChar; a.subChar = sd->subChar; }
  const int      lane = lane_id();
  const uint64_t w    = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t r    = w * stride;
  if (r >= a.n) return;
  const uint32_t L = a.len[r], S = L < 4096u ? L : 4096u;
  #pragma unroll 1
  for (int k = 0; k < 2; k++)
    { const int rc = k == 0 ? a.delChar : a.subChar;
      if (rc < 0) continue;
      const uint8_t *p = line_ptr(a, r, L, k == 0 ? 0 : 4);
      uint32_t tokens = 0;
      for (uint32_t j = (uint32_t) lane; j < S; j += 64u)
        tokens += p[j] != (uint8_t) rc ? 1u : 0u;
      tokens = wave_sum(tokens);
      if (lane == 0)
        { atomicAdd(&cnt[2 * k], (unsigned long long) S);
          atomicAdd(&cnt[2 * k + 1], (unsigned long long) tokens);
        }
    }
}

// the slot share (in 256ths) the sample asks for: one and a half times the denser line's token share and a twentieth on top,
// a quarter at least (entries differ), everything at most
__device__ __forceinline__ uint32_t tok_share_of(const unsigned long long *cnt)
{ const double fd = cnt[0] ? (double) cnt[1] / (double) cnt[0] : 0.0, fs = cnt[2] ? (double) cnt[3] / (double) cnt[2] : 0.0;
  const double f  = 1.5 * (fd > fs ? fd : fs) + 0.05;
  const uint32_t fr8 = (uint32_t) (256.0 * f + 0.999);
  return fr8 < 64u ? 64u : (fr8 > 256u ? 256u : fr8);
}

// slot sizes of the token hand-over (tokens per entry); the share it used goes to cnt[4] for the host's records
// sd (dx_qv_scan): the share and the instance of k_qv_hist the batch wants go to the record too -- SCAN_MISS with them when the
// instance is not the one the host has launched (guess)
#define HIST_SHARE_A 110u                                // tokens at most ~28 % of the denser line (run densities from ~0.72 up): instance A
__global__ __launch_bounds__(DX_BLOCK)
void k_tok_rooms(const uint32_t *len, uint64_t n, unsigned long long *cnt, uint32_t *room, scan_dev *sd, uint32_t guess, int shared_tokens)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  const uint32_t fr8 = tok_share_of(cnt);
  if (i == 0)
    { cnt[4] = fr8;
      if (sd != NULL)
        { const uint32_t inst = sd->delChar < 0 && sd->subChar < 0 ? SCAN_NOTOK
                              : (sd->delChar >= 0 && sd->subChar >= 0 ? (fr8 <= HIST_SHARE_A && !shared_tokens ? SCAN_A : SCAN_B) : SCAN_G);
          sd->share8 = fr8;
          sd->inst   = inst | (inst != guess ? SCAN_MISS : 0u);
        }
    }
  if (i < n) room[i] = tok_room(len[i], fr8);
}

// =============================================================================================
//  token tables in LDS
// =============================================================================================
__device__ __forceinline__ void load_tables(uint32_t (*s_tok)[256], const uint32_t *g_tok)
{ for (int k = threadIdx.x; k < DX_TOK_WORDS; k += (int) blockDim.x)
    (&s_tok[0][0])[k] = g_tok[k];
  __syncthreads();
}

// words Encode/Encode_Run write beyond the data for T bits whose final OCODE piece had `last`
// bits (QV.c:436-442): one more word when the decoder's 16-bit look-ahead would otherwise run
// past the stream.
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

__device__ __forceinline__ uint32_t last_piece_byte(const uint32_t *tab, uint32_t last_byte, uint32_t L, uint32_t mask)
{ if (L == 0) return 0;
  const uint32_t e = tab[last_byte & mask];
  return TOK_ESC(e) ? 8u : TOK_LEN(e);
}

__device__ __forceinline__ uint32_t seg_bytes(uint64_t T, uint32_t last)
{ return 4u * ((uint32_t) (T >> 5) + (((uint32_t) T & 31u) ? 1u : 0u) + pad_extra(T, last)); }

__device__ __forceinline__ uint32_t run_token_len(const uint32_t *rtab, uint32_t run)
{ const uint32_t e = rtab[run > 255u ? 255u : run];               // QV.c:479-487
  return TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
}

// =============================================================================================
//  size pass: bit totals only (4 streams in one sweep)
// =============================================================================================
// No tokens are formed here, only their lengths are summed, which needs far less than the dense
// token list of the other two passes:
//  * symbol codes: a byte-wide length table per stream (256 B: one LDS bank per dword, no bank
//    conflicts); for a run-coded stream the run character's entry is 0, so all 16 bytes of a lane
//    are simply looked up and added;
//  * run codes (Encode_Run, QV.c:475-487): the non-run mask of a lane's 16 bytes is split into two
//    bytes; the runs strictly inside a mask byte are summed by a 256-entry table built from the
//    run scheme at kernel start; what remains per lane is the run in front of its first token
//    (which needs the end of the previous token: a max-scan over the lanes, carried over steps in
//    C) and the run across the middle of the mask.
struct size_tabs
{ uint8_t  len[6][256];     // token length per symbol / per run value (run: + 16 if escaped)
  uint16_t inner[2][256];   // per 8-bit non-run mask: sum of run-token lengths of the runs inside it
};

__device__ __forceinline__ void load_size_tables(size_tabs &t, const uint32_t *g_tok, int delChar, int subChar)
{ for (int k = threadIdx.x; k < 6 * 256; k += (int) blockDim.x)
    { const uint32_t e = g_tok[k];
      uint32_t l = TOK_LEN(e) + ((k >= 4 * 256 && TOK_ESC(e)) ? 16u : 0u);
      if ((delChar >= 0 && k == DX_DEL * 256 + delChar) || (subChar >= 0 && k == DX_SUB * 256 + subChar))
        l = 0;
      (&t.len[0][0])[k] = (uint8_t) l;
    }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * 256; k += (int) blockDim.x)
    { const uint8_t *rl = t.len[4 + (k >> 8)];
      uint32_t m = (uint32_t) k & 0xffu, sum = 0;
      int prev = -1;
      for (int b = 0; b < 8; b++)
        if ((m >> b) & 1u)
          { if (prev >= 0) sum += rl[b - prev - 1];
            prev = b;
          }
      (&t.inner[0][0])[k] = (uint16_t) sum;
    }
  __syncthreads();
}

// sum of the symbol-code lengths of a lane's bytes [0, valid)
__device__ __forceinline__ uint32_t bits_syms_step(const u32x4 &c, int valid, const uint8_t *len, uint32_t m4)
{ uint32_t acc = 0;
  #pragma unroll
  for (int b = 0; b < 16; b++)
    acc += len[((chunk_word(c, b >> 2) & m4) >> (8 * (b & 3))) & 0xffu];
  return acc - (16u - (uint32_t) valid) * len[0];                // missing bytes were read as 0
}

// sum of the run-code lengths of the tokens that start in this lane's bytes; C = run open at the
// step's start (in), at its end (out); nonrun += this lane's token count
__device__ __forceinline__ uint32_t bits_runs_step(const u32x4 &c, int valid, uint32_t sv, uint32_t rc, uint32_t &C,
                                                   uint32_t &nonrun, const uint8_t *rlen, const uint16_t *inner)
{ const uint32_t nr = ~chunk_eq_mask(c, rc) & ((1u << valid) - 1u);
  const uint32_t lo = nr & 0xffu, hi = nr >> 8;
  const uint32_t p0 = 16u * (uint32_t) lane_id();
  // end (position + 1) of the last token at or before each lane; 0: none yet in this step
  const uint32_t incl = wave_incl_max(nr ? p0 + 32u - (uint32_t) __clz(nr) : 0u);
  const uint32_t prev = __builtin_amdgcn_update_dpp(0u, incl, 0x138, 0xf, 0xf, true);   // wave_shr:1
  uint32_t acc = (uint32_t) inner[lo] + (uint32_t) inner[hi];
  if (lo && hi)
    acc += rlen[(uint32_t) __clz(lo) - 24u + (uint32_t) __ffs(hi) - 1u];      // zeros above lo's top bit + below hi's lowest
  if (nr)
    { const uint32_t run = p0 + (uint32_t) __ffs(nr) - 1u - (prev ? prev : 0u - C);
      acc += rlen[run > 255u ? 255u : run];                                    // QV.c:479-482
    }
  nonrun += __popc(nr);
  const uint32_t end = __builtin_amdgcn_readlane(incl, 63);
  C = end ? sv - end : C + sv;
  return acc;
}

__global__ __launch_bounds__(DX_BLOCK, SIZES_WAVES)
void k_qv_sizes(qv_args a, const uint32_t *g_tok, const uint64_t *hdr_off, uint32_t *seg /* n x 5 */,
                uint32_t *rec_size, uint32_t *ticket,
                const uint32_t *only_list, const unsigned long long *only_count, uint64_t first_entry, const uint32_t *only_info)
{ __shared__ uint32_t  s_tok[6][256];
  __shared__ size_tabs s_t;
  load_tables(s_tok, g_tok);
  load_size_tables(s_t, g_tok, a.delChar, a.subChar);
  const int      lane  = lane_id();
  const uint32_t imask = a.lossy ? 0xfeu : 0xffu, mmask = a.lossy ? 0xfcu : 0xffu;
  const uint32_t im4 = imask * 0x01010101u, mm4 = mmask * 0x01010101u;
  const bool drun = a.delChar >= 0, srun = a.subChar >= 0;

  // all entries of the batch in pairs from the ticket counter -- or (beside k_qv_sizes_fast) just the listed ones whose
  // tokens are unusable, as in k_qv_encode
  const uint64_t listed = only_list ? (uint64_t) *only_count : 0;
  uint64_t pend = 0;
  uint32_t mine = 0;
  for (uint64_t r0 = only_list ? 0 : next_unit(ticket, a.units), nxt = 0; ; r0 = nxt)
  { uint64_t rlo, rhi;
    if (only_list == NULL)
      { if (r0 >= a.n) break;
        nxt = next_unit(ticket, a.units);
        rlo = r0; rhi = r0 + a.units < a.n ? r0 + a.units : a.n;
      }
    else
      { while (pend == 0)
          { const uint64_t t = next_unit(ticket, 64u);
            if (t >= listed) break;
            mine = t + (uint64_t) lane < listed ? only_list[t + (uint64_t) lane] : 0xffffffffu;
            pend = __ballot(mine != 0xffffffffu && (uint64_t) mine >= first_entry && (uint64_t) mine - first_entry < a.n);
          }
        if (pend == 0) break;
        const int l = __ffsll((unsigned long long) pend) - 1;
        pend &= pend - 1;
        rlo = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) mine, l) - first_entry;
        rhi = rlo + 1;
        if (!tok_unusable(only_info, rlo, a.delChar, a.subChar))
          continue;
      }
    for (uint64_t r = rlo; r < rhi; r++)
    { const uint32_t L = a.len[r];
      const uint8_t *p0 = line_ptr(a, r, L, 0), *p2 = line_ptr(a, r, L, 2);
      const uint8_t *p3 = line_ptr(a, r, L, 3), *p4 = line_ptr(a, r, L, 4);
      const bool over = can_overread(a, p4, L);
      uint32_t C0 = 0, C4 = 0, nn0 = 0, nn4 = 0;
      uint32_t a0 = 0, a2 = 0, a3 = 0, a4 = 0;          // per-lane bit sums of the four streams
      uint64_t T0 = 0, T2 = 0, T3 = 0, T4 = 0;

      uint32_t pos = 16u * lane;
      u32x4 c0 = fetch(p0, pos, L, over), c2 = fetch(p2, pos, L, over);
      u32x4 c3 = fetch(p3, pos, L, over), c4 = fetch(p4, pos, L, over);
      for (uint32_t base = 0; base < L; base += DX_STEP)
        { const uint32_t np = pos + DX_STEP;
          const u32x4 d0 = fetch(p0, np, L, over), d2 = fetch(p2, np, L, over);
          const u32x4 d3 = fetch(p3, np, L, over), d4 = fetch(p4, np, L, over);
          const uint32_t sv    = L - base >= DX_STEP ? DX_STEP : L - base;
          const int      valid = valid_of(pos, L);
          a0 += bits_syms_step(c0, valid, s_t.len[DX_DEL], ~0u);
          a2 += bits_syms_step(c2, valid, s_t.len[DX_INS], im4);
          a3 += bits_syms_step(c3, valid, s_t.len[DX_MRG], mm4);
          a4 += bits_syms_step(c4, valid, s_t.len[DX_SUB], ~0u);
          if (drun) a0 += bits_runs_step(c0, valid, sv, (uint32_t) a.delChar, C0, nn0, s_t.len[DX_DRUN], s_t.inner[0]);
          if (srun) a4 += bits_runs_step(c4, valid, sv, (uint32_t) a.subChar, C4, nn4, s_t.len[DX_SRUN], s_t.inner[1]);
          c0 = d0; c2 = d2; c3 = d3; c4 = d4;
          pos = np;
          if ((base & 0x3ffffffu) == 0x3fffc00u)       // fold long before a 32-bit lane sum can wrap
            { T0 += wave_sum(a0); T2 += wave_sum(a2); T3 += wave_sum(a3); T4 += wave_sum(a4);
              a0 = a2 = a3 = a4 = 0;
            }
        }
      T0 += wave_sum(a0); T2 += wave_sum(a2); T3 += wave_sum(a3); T4 += wave_sum(a4);

      uint32_t last0, last4, clen = L;
      if (drun)
        { clen = wave_sum(nn0);                          // Pack_Tag's count, QV.c:810-819
          if (C0 > 0)                                    // trailing run token
            { const uint32_t e = s_tok[DX_DRUN][C0 > 255u ? 255u : C0];
              T0   += TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
              last0 = TOK_ESC(e) ? 16u : TOK_LEN(e);
            }
          else
            last0 = last_piece_plain(s_tok[DX_DEL], p0, L, 0xffu);
        }
      else
        last0 = last_piece_plain(s_tok[DX_DEL], p0, L, 0xffu);
      if (srun)
        { if (C4 > 0)
            { const uint32_t e = s_tok[DX_SRUN][C4 > 255u ? 255u : C4];
              T4   += TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
              last4 = TOK_ESC(e) ? 16u : TOK_LEN(e);
            }
          else
            last4 = last_piece_plain(s_tok[DX_SUB], p4, L, 0xffu);
        }
      else
        last4 = last_piece_plain(s_tok[DX_SUB], p4, L, 0xffu);

      const uint32_t s0 = seg_bytes(T0, last0);
      const uint32_t s1 = (clen + 3u) >> 2;
      const uint32_t s2 = seg_bytes(T2, last_piece_plain(s_tok[DX_INS], p2, L, imask));
      const uint32_t s3 = seg_bytes(T3, last_piece_plain(s_tok[DX_MRG], p3, L, mmask));
      const uint32_t s4 = seg_bytes(T4, last4);
      if (lane == 0)
        { const uint32_t hl = hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u;
          uint32_t *sg = seg + 5 * r;
          sg[0] = s0; sg[1] = s1; sg[2] = s2; sg[3] = s3; sg[4] = s4;
          rec_size[r] = hl + s0 + s1 + s2 + s3 + s4;
        }
    }
  }
}

// =============================================================================================
//  exclusive scan of the record sizes (file order) -> record offsets
// =============================================================================================
#define SCAN_ITEMS 16                                   // items per thread
#define SCAN_TILE  (DX_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t *s_wave, uint64_t &total)
{ const int lane = lane_id(), wid = threadIdx.x >> 6;
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

// the same for one group of a longer array: offsets start at base_in[0]; base_out[0] = where the next group starts
__global__ __launch_bounds__(DX_BLOCK)
void k_scan_apply_base(const uint32_t *in, uint64_t n, const uint64_t *tile_sum, uint64_t *out /* n+1 */,
                       const uint64_t *grand, const uint64_t *base_in, uint64_t *base_out)
{ __shared__ uint64_t s_wave[DX_WAVES_PER_BLK];
  const uint64_t t0 = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
  for (int k = 0; k < SCAN_ITEMS; k++)
    { v[k] = t0 + k < n ? in[t0 + k] : 0u;
      s   += v[k];
    }
  uint64_t tot;
  uint64_t at = *base_in + tile_sum[blockIdx.x] + block_excl_scan(s, s_wave, tot);
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (t0 + k < n)
      { out[t0 + k] = at;
        at += v[k];
      }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    { out[n]    = *base_in + *grand;
      *base_out = *base_in + *grand;
    }
}

// =============================================================================================
//  encode pass
// =============================================================================================
// lane-local MSB-first bit accumulator feeding the window (generic path: any token lengths)
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

// funnel: low 32 bits of (hi:lo) >> s, 1 <= s <= 32
__device__ __forceinline__ uint32_t fsr(uint32_t hi, uint32_t lo, uint32_t s)
{ return (uint32_t) ((((uint64_t) hi << 32) | lo) >> s); }

// The lanes' bit strings of one step go into the window in lane order.  Normally the whole step
// fits (one round); if the window is too full it is drained first, and with pathological code
// tables a step is split at lane boundaries into several rounds.  `incl` = inclusive prefix sum
// of the lanes' bit counts `nb`; the statement block receives `bit_`, this lane's bit offset.
// FOR_EACH_ROUND drains a well-filled window behind the step; FOR_EACH_ROUND_LATE leaves that to the caller's loop,
// which does it at the START of its next step, behind the request for the chunk after: the drain's stores and that
// load then have a whole step to complete in before anything waits for them (vmcnt counts both, and a wait the
// compiler places is always for everything outstanding).
#define FOR_EACH_ROUND(o, incl, nb, ...)       FOR_EACH_ROUND_(o, incl, nb, true, __VA_ARGS__)
#define FOR_EACH_ROUND_LATE(o, incl, nb, ...)  FOR_EACH_ROUND_(o, incl, nb, false, __VA_ARGS__)
#define FOR_EACH_ROUND_(o, incl, nb, DRAIN, ...)                                                 \
  { uint32_t done_ = 0, lo_ = 0;                                                                \
    const int lane_ = lane_id();                                                                \
    while (lo_ < 64u)                                                                           \
      { const uint32_t cap_ = QV_WIN_BITS - (o).winbits;                                        \
        uint32_t hi_ = (uint32_t) __popcll(__ballot((incl) <= done_ + cap_));                   \
        if (hi_ <= lo_)                                                                         \
          { if ((o).winbits >= 32u) { flush_words((o), false); continue; }                      \
            hi_ = 64u;              /* cannot happen (one lane always fits); never spin */      \
          }                                                                                     \
        if ((uint32_t) lane_ >= lo_ && (uint32_t) lane_ < hi_ && (nb))                          \
          { const uint32_t bit_ = (o).winbits + ((incl) - (nb)) - done_;                        \
            __VA_ARGS__                                                                         \
          }                                                                                     \
        const uint32_t upto_ = __builtin_amdgcn_readlane((incl), (int) hi_ - 1);                \
        (o).winbits += upto_ - done_;                                                           \
        done_ = upto_;                                                                          \
        lo_   = hi_;                                                                            \
      }                                                                                         \
    if ((DRAIN) && (o).winbits >= QV_FLUSH_BITS)                                                \
      flush_quads((o), false);                                                                  \
  }

// The same for steps in which every lane's string has at most 128 bits (the branch-free packing paths): a step is then at most
// 8192 bits, which a drained window (< 32 bits left) always holds -- one round, no lane ranges, no ballots.  The drain,
// when one is needed, is of whole words (flush_words); the periodic 16-byte drains stay with the caller as in _LATE.
#define FOR_ONE_ROUND(o, incl, nb, ...)                                                         \
  { const uint32_t all_ = wave_total(incl);                                                     \
    if ((o).winbits + all_ > QV_WIN_BITS) flush_words((o), false);                              \
    if (nb)                                                                                     \
      { const uint32_t bit_ = (o).winbits + ((incl) - (nb));                                    \
        __VA_ARGS__                                                                             \
      }                                                                                         \
    (o).winbits += all_;                                                                        \
  }

// write the partial word and the pad word (QV.c:436-442); returns the segment's byte size
// (a segment's end finds up to QV_FLUSH_BITS + a step's bits in the window: out in 16-byte pieces, then the up to three whole
// words behind them, the partial word and the pad word by a lane each -- word by word, two loops of four-byte stores, this
// took a tenth of the encoder's time at 10 kb and a fifth at 2 kb)
__device__ __forceinline__ uint32_t finish_words(wave_out &o, uint32_t last)
{ const uint32_t lane = (uint32_t) lane_id();
  const uint32_t nq   = o.winbits >> 7;
  u32x4         *win4 = (u32x4 *) o.win;
  wave_sync();
  for (uint32_t j = lane; j < nq; j += 64u)
    *(u32x4_u *) (o.seg + 4ull * o.wordbase + 16ull * j) = win4[j];
  const u32x4    rest  = win4[nq];
  const uint32_t wb    = o.wordbase + 4u * nq;                   // words stored so far
  const uint32_t rbits = o.winbits & 127u, nfull = rbits >> 5;
  const uint64_t T     = 32ull * wb + rbits;
  const uint32_t tailw = ((rbits & 31u) ? 1u : 0u) + pad_extra(T, last);
  if (lane < nfull + tailw)                                      // (the pad word repeats the partial one, as before)
    { const uint32_t k = lane < nfull ? lane : nfull;
      store32_u(o.seg + 4ull * (wb + lane), k == 0 ? rest.x : (k == 1 ? rest.y : (k == 2 ? rest.z : rest.w)));
    }
  const u32x4 zero = { 0u, 0u, 0u, 0u };
  wave_sync();
  for (uint32_t j = lane; j <= nq; j += 64u)
    win4[j] = zero;
  o.wordbase = wb + nfull;
  o.winbits  = rbits & 31u;
  wave_sync();
  return 4u * (o.wordbase + tailw);
}

// Shift tokens for the plain streams' packing chain: code bits left-aligned in the word, low byte
// s = 32 - length (8..31; 32 for a symbol without a code).  v_alignbit_b32 takes its shift from
// the low 5 bits of an operand, so the token itself is both the shift and the low word of every
// funnel shift of the chain, and the low byte sums to 32*16 - bits.
#define STOK_DUMMY 31u                                  // one zero bit; stands in for a missing byte
__device__ __forceinline__ uint32_t shift_token(uint32_t t)
{ const uint32_t l = TOK_LEN(t);
  return l ? ((TOK_BITS(t) << (32u - l)) | (32u - l)) : 32u;
}

// tables 0..3: symbol schemes; 4, 5: run schemes, bare run code with bit 7 = escape (the 16-bit
// literal follows, QV.c:486-487); tagcode: Number_Read's letter -> 2-bit code (DB.c:319-338)
__device__ __forceinline__ void load_shift_tables(uint32_t (*s_stok)[256], uint8_t *s_tagcode, const uint32_t *g_tok)
{ for (int k = threadIdx.x; k < 6 * 256; k += (int) blockDim.x)
    { const uint32_t t = g_tok[k];
      (&s_stok[0][0])[k] = shift_token(t) | ((k >= 4 * 256 && TOK_ESC(t)) ? 0x80u : 0u);
    }
  for (int k = threadIdx.x; k < 256; k += (int) blockDim.x)
    { const int u = k & 0xdf;
      s_tagcode[k] = (uint8_t) (u == 'C' ? 1 : (u == 'G' ? 2 : (u == 'T' ? 3 : 0)));
    }
  __syncthreads();
}

// append the token's bits to the 128-bit string w3:w0
#define STOK_APPEND(t)                                                                          \
  { w3 = __builtin_amdgcn_alignbit(w3, w2, (t));                                                \
    w2 = __builtin_amdgcn_alignbit(w2, w1, (t));                                                \
    w1 = __builtin_amdgcn_alignbit(w1, w0, (t));                                                \
    w0 = __builtin_amdgcn_alignbit(w0, (t), (t));                                               \
  }

// ... and to a 96-bit string w2:w0, for the steps in which no lane of the wave has more (nearly all: sixteen symbols of a plain line are
// 70 bits on average, eight tokens 64): three funnel shifts a token instead of four, four words placed instead of five
#define STOK_APPEND3(t)                                                                         \
  { w2 = __builtin_amdgcn_alignbit(w2, w1, (t));                                                \
    w1 = __builtin_amdgcn_alignbit(w1, w0, (t));                                                \
    w0 = __builtin_amdgcn_alignbit(w0, (t), (t));                                               \
  }
#ifndef FAST_CHAIN96
#define FAST_CHAIN96 1
#endif
__device__ __forceinline__ void place_bits96(uint32_t *win, uint32_t bit, uint32_t nb, uint32_t w0, uint32_t w1, uint32_t w2)
{ const uint32_t e  = bit + nb;
  const uint32_t sl = (32u - (e & 31u)) & 31u, sr = 32u - sl;
  const uint32_t we = (e - 1u) >> 5;
  const uint32_t x0 = w0 << sl;
  const uint32_t x1 = fsr(w1, w0, sr), x2 = fsr(w2, w1, sr), x3 = fsr(0u, w2, sr);
  atomicOr(&win[we], x0);
  atomicOr(&win[(int) we - 1], x1);
  atomicOr(&win[(int) we - 2], x2);
  atomicOr(&win[(int) we - 3], x3);
}

// OR the nb (1..128) bits right-aligned in w3:w0 into the window at bit offset `bit`
__device__ __forceinline__ void place_bits128(uint32_t *win, uint32_t bit, uint32_t nb,
                                              uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{ const uint32_t e  = bit + nb;                                // end bit (exclusive)
  const uint32_t sl = (32u - (e & 31u)) & 31u, sr = 32u - sl;   // left-align to the last word
  const uint32_t we = (e - 1u) >> 5;
  const uint32_t x0 = w0 << sl;
  const uint32_t x1 = fsr(w1, w0, sr), x2 = fsr(w2, w1, sr), x3 = fsr(w3, w2, sr), x4 = fsr(0u, w3, sr);
  // x1 .. x4 are zero where the string does not reach: ORed all the same -- five LDS operations instead of up to four
  // branches on the execution mask (the kernels are bound by instruction issue, not by the LDS).  A word in front of the
  // window (we < 4) may be named with a zero: that is what the QV_WIN_PAD words in front of every window are for.
  atomicOr(&win[we], x0);
  atomicOr(&win[(int) we - 1], x1);
  atomicOr(&win[(int) we - 2], x2);
  atomicOr(&win[(int) we - 3], x3);
  atomicOr(&win[(int) we - 4], x4);
}

// ---- group index of the plain lines (decoder side: k_qv_decode_sub, dx_qv_decode.hip) -----------------------
// A Huffman stream decodes front to back only; the encoder, though, knows where every code starts.  On request
// (dx_qv_subindex) it leaves, for each plain line, one byte per group of 16 symbols -- the 16 a lane codes in a
// step -- holding the group's code bits minus its symbols (16 codes of 1..16 bits: 0..240).  The prefix sum of 64
// such bytes gives the decoder the bit at which each of 64 consecutive groups starts.  A line with a symbol that
// has no code (hand-made tables) gets SUB_NONE in its first byte: that line is decoded group after group.
// Per entry 4 * sub_words(L) words, at sub_off[r].
struct sub_mark
{ uint8_t *at;                // this line's bytes, NULL: no index wanted
  uint32_t g;                 // the group this lane codes in this step
};

__device__ __forceinline__ void sub_begin(sub_mark &m, uint32_t *at)
{ const uint64_t A = (uint64_t) at;                             // wave-uniform: keep it in scalar registers
  m.at = (uint8_t *) (((uint64_t) uniform((uint32_t) (A >> 32)) << 32) | uniform((uint32_t) A));
  m.g  = (uint32_t) lane_id();
}

// nb = code bits of this lane's `valid` symbols of the step; nocode = one of them has no code
__device__ __forceinline__ void sub_step(sub_mark &m, uint32_t nb, uint32_t valid, bool nocode)
{ if (m.at == NULL) return;
  if (valid) m.at[m.g] = (uint8_t) (nb - valid);
  if (__any((int) nocode)) { wave_sync(); if (lane_id() == 0) m.at[0] = (uint8_t) SUB_NONE; m.at = NULL; }
  m.g += 64u;
}

struct sub_sink { uint32_t *idx; const uint64_t *off; uint32_t *none; };      // idx == NULL: none wanted; none: counts the RUN_NONE lines

// one step of Encode (QV.c:427-434): 16 table look-ups per lane, prefix sum, bits into the window
// (LATE: the caller drains the window, see FOR_EACH_ROUND_LATE)
template <bool LATE = false>
__device__ __forceinline__ void encode_plain_step(wave_out &o, const u32x4 &c, int valid, bool full,
                                                  const uint32_t *tab, const uint32_t *stab, uint32_t m4, sub_mark &sm)
{ uint32_t tok[16];
  uint32_t ssum = 0, zor = 0;
  if (full)
    {
      #pragma unroll
      for (int b = 0; b < 16; b++)
        { tok[b] = stab[((chunk_word(c, b >> 2) & m4) >> (8 * (b & 3))) & 0xffu];
          ssum += tok[b] & 0xffu;
          zor  |= tok[b];
        }
    }
  else
    {
      #pragma unroll
      for (int b = 0; b < 16; b++)
        { const uint32_t t = stab[((chunk_word(c, b >> 2) & m4) >> (8 * (b & 3))) & 0xffu];
          tok[b] = b < valid ? t : STOK_DUMMY;
          ssum  += b < valid ? (t & 0xffu) : 32u;
          zor   |= tok[b];
        }
    }
  const uint32_t nb   = 512u - ssum;
  const uint32_t k    = 16u - (uint32_t) valid;                        // dummies (0 unless ragged)
  const uint32_t incl = wave_incl_scan(nb);
  sub_step(sm, nb, (uint32_t) valid, (zor & 32u) != 0u);
  const bool fast = !__any((int) ((zor & 32u) | (nb + k > 128u)));
  if (fast)
    { // every token has 1..24 bits and the lane's string fits 128 bits: branch-free packing.  The
      // dummies of a ragged last chunk append one zero bit each, shifted out again at the end.
      FOR_ONE_ROUND(o, incl, nb,
        { uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
          _Pragma("unroll")
          for (int b = 0; b < 16; b++)
            STOK_APPEND(tok[b])
          w0 = __builtin_amdgcn_alignbit(w1, w0, k);
          w1 = __builtin_amdgcn_alignbit(w2, w1, k);
          w2 = __builtin_amdgcn_alignbit(w3, w2, k);
          w3 >>= k;
          place_bits128(o.win, bit_, nb, w0, w1, w2, w3);
        })
      if (!LATE && o.winbits >= QV_FLUSH_BITS) flush_quads(o, false);
    }
  else
    { FOR_EACH_ROUND_(o, incl, nb, !LATE,
        { bit_acc s;
          acc_begin(s, bit_);
          _Pragma("unroll 1")
          for (int b = 0; b < valid; b++)
            { const uint32_t t = tab[chunk_byte(c, b) & (m4 & 0xffu)];
              acc_put(s, o.win, TOK_BITS(t), TOK_LEN(t));
            }
          acc_end(s, o.win);
        })
    }
}

// one step of Encode_Run (QV.c:475-497).  The step's non-run symbols (dense list from run_collect)
// are handled in passes of up to 64*RUN_TP tokens: every lane takes T <= RUN_TP consecutive tokens,
// chains their (run code [+ 16-bit literal] + symbol code) pieces into one string of <= 128 bits,
// and a single prefix sum + placement per pass puts the strings into the window.  With TAGS the
// same lanes also emit the 2-bit codes of the deletion tags under their symbols (Pack_Tag +
// Number_Read + Compress_Read, QV.c:810-819, 1402-1404) into the tag window.
#define RUN_TP 6u
__device__ __forceinline__ void encode_runs_step(wave_out &o, wave_out &ot, const run_lds &R, uint8_t *tagchunk,
                                                 const uint8_t *tagcode, const bool TAGS, const u32x4 &c, const u32x4 &t,
                                                 int valid, uint32_t sv, uint32_t rc, uint32_t &C,
                                                 const uint32_t *ntab, const uint32_t *rtab,
                                                 const uint32_t *nstab, const uint32_t *rstab)
{ const int lane = lane_id();
  if (TAGS)
    *(u32x4 *) (tagchunk + 16 * lane) = t;
  const uint32_t total = run_collect(R, c, valid, rc);
  for (uint32_t k0 = 0; k0 < total; k0 += 64u * RUN_TP)
    { const uint32_t m     = total - k0 < 64u * RUN_TP ? total - k0 : 64u * RUN_TP;
      const uint32_t T     = (m + 63u) >> 6;
      const uint32_t first = k0 + (uint32_t) lane * T;
      const uint32_t cnt   = first < k0 + m ? (k0 + m - first < T ? k0 + m - first : T) : 0u;
      // run before a token = its position - base; base = previous non-run position + 1, or -C
      // (the run open at the step's start) for the step's first token
      uint32_t base0 = 0;
      if (cnt)
        base0 = first ? (uint32_t) R.list[first - 1] + 1u : 0u - C;
      uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, nb = 0, zor = 0, tacc = 0, base = base0;
      // three tokens at a time, each look-up level for all three before the next level: the
      // dependent LDS reads (position -> symbol -> code) cost three waits per group, not per token
      #pragma unroll 1
      for (uint32_t j0 = 0; j0 < T; j0 += 3u)
        { uint32_t pos[3], rt[3], st[3], tg[3];
          #pragma unroll
          for (int k = 0; k < 3; k++)
            pos[k] = j0 + k < cnt ? (uint32_t) R.list[first + j0 + k] : 0u;
          #pragma unroll
          for (int k = 0; k < 3; k++)
            { st[k] = R.chunk[pos[k]];
              if (TAGS) tg[k] = tagchunk[pos[k]];
            }
          #pragma unroll
          for (int k = 0; k < 3; k++)
            { const uint32_t run = pos[k] - (k ? pos[k - 1] + 1u : base);
              rt[k] = rstab[run > 255u ? 255u : run];                 // QV.c:479-487
              st[k] = nstab[st[k]];
              if (TAGS) tg[k] = tagcode[tg[k]];
            }
          #pragma unroll
          for (int k = 0; k < 3; k++)
            if (j0 + k < cnt)
              { STOK_APPEND(rt[k])
                if (rt[k] & 0x80u)
                  { const uint32_t run = pos[k] - (k ? pos[k - 1] + 1u : base);
                    const uint32_t lit = (run << 16) | 16u;
                    STOK_APPEND(lit)
                    w0 |= run & 0xffff0000u;                          // OCODE(16,run) with run >= 2^16 (QV.c:411,420)
                    nb += 16u;
                  }
                STOK_APPEND(st[k])
                nb  += 64u - (rt[k] & 0x3fu) - (st[k] & 0xffu);
                zor |= rt[k] | st[k];
                if (TAGS)
                  tacc = (tacc << 2) | tg[k];
              }
          base = pos[2] + 1u;                                         // only used when the lane has a next group
        }
      const uint32_t incl = wave_incl_scan(nb);
      if (!__any((int) ((zor & 32u) | (nb > 128u))))
        { FOR_EACH_ROUND(o, incl, nb,
            { place_bits128(o.win, bit_, nb, w0, w1, w2, w3); })
        }
      else                                    // a symbol or run without a code, or a string > 128 bits
        { FOR_EACH_ROUND(o, incl, nb,
            { bit_acc s;
              acc_begin(s, bit_);
              uint32_t b2 = base0;
              _Pragma("unroll 1")
              for (uint32_t j = 0; j < cnt; j++)
                { const uint32_t pos = R.list[first + j];
                  const uint32_t run = pos - b2;
                  b2 = pos + 1u;
                  const uint32_t re = rtab[run > 255u ? 255u : run];
                  const uint32_t se = ntab[R.chunk[pos]];
                  acc_put(s, o.win, TOK_ESC(re) ? ((TOK_BITS(re) << 16) | run) : TOK_BITS(re),
                          TOK_LEN(re) + (TOK_ESC(re) ? 16u : 0u));
                  acc_put(s, o.win, TOK_BITS(se), TOK_LEN(se));
                }
              acc_end(s, o.win);
            })
        }
      if (TAGS)
        { if (cnt)
            { const uint32_t bit = ot.winbits + 2u * (first - k0);
              const uint32_t w = bit >> 5, sh = bit & 31u;
              const uint32_t v = tacc << (32u - 2u * cnt);
              atomicOr(&ot.win[w], v >> sh);
              if (sh + 2u * cnt > 32u)
                atomicOr(&ot.win[w + 1], v << (32u - sh));
            }
          ot.winbits += 2u * m;
          if (ot.winbits >= TAG_FLUSH_BITS)
            flush_quads(ot, true);
        }
    }
  C = run_after(R, total, sv, C);
  wave_sync();
}

// the run-only token that ends a stream whose last symbol is the run character (QV.c:476-487);
// returns the length of the final piece for the pad rule
__device__ __forceinline__ uint32_t encode_trailing_run(wave_out &o, uint32_t C, const uint32_t *rtab)
{ const uint32_t re = rtab[C > 255u ? 255u : C];
  const uint32_t tl = TOK_LEN(re) + (TOK_ESC(re) ? 16u : 0u);
  if (o.winbits + tl > QV_WIN_BITS)
    flush_words(o, false);
  if (lane_id() == 0 && tl)
    { bit_acc s;
      acc_begin(s, o.winbits);
      acc_put(s, o.win, TOK_ESC(re) ? ((TOK_BITS(re) << 16) | C) : TOK_BITS(re), tl);
      acc_end(s, o.win);
    }
  o.winbits += tl;
  return TOK_ESC(re) ? 16u : TOK_LEN(re);
}

// one step of Pack_Tag + Number_Read + Compress_Read (QV.c:810-819, 1402-1404): the tags at the
// positions in `keep`, 2 bits each, into the tag window
__device__ __forceinline__ void encode_tags_step(wave_out &o, const u32x4 &t, uint32_t keep)
{ const uint32_t cnt = __popc(keep);
  uint32_t acc = 0;
  int      sh  = 30;
  #pragma unroll
  for (int b = 0; b < 16; b++)
    if ((keep >> b) & 1u)
      { const uint32_t u = BYTE_OF(t, b) & 0xdfu;
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
  o.winbits += 2u * wave_total(incl);
  if (o.winbits >= TAG_FLUSH_BITS)
    flush_quads(o, true);
}

__device__ __forceinline__ uint32_t finish_tags(wave_out &o)
{ flush_words(o, true);
  const uint32_t clen_bytes = 4u * o.wordbase + ((o.winbits + 7u) >> 3);      // (clen + 3) >> 2
  if (lane_id() == 0)
    { const uint32_t w = __builtin_bswap32(o.win[0]);
      for (uint32_t k = 0; 8u * k < o.winbits; k++)
        o.seg[4ull * o.wordbase + k] = (uint8_t) (w >> (8 * k));
      o.win[0] = 0;
    }
  wave_sync();
  return clen_bytes;
}

// tags of an entry whose deletion stream is NOT run coded (no delChar): every position is kept
__device__ __forceinline__ uint32_t encode_all_tags(wave_out &ot, const uint8_t *p1, uint32_t L, bool over)
{ const int lane = lane_id();
  uint32_t pos = 16u * lane;
  u32x4 t = fetch(p1, pos, L, over);
  for (uint32_t base = 0; base < L; base += DX_STEP)
    { const u32x4 u = fetch(p1, pos + DX_STEP, L, over);
      encode_tags_step(ot, t, (1u << valid_of(pos, L)) - 1u);
      t = u;
      pos += DX_STEP;
    }
  return finish_tags(ot);
}

// The encoders form addresses from index arrays (entry offsets and lengths, slot offsets, token slots).  An index that
// does not hold together is reported (status bit 6) and its entry skipped -- never followed into memory: a kernel that
// faults takes the process, and possibly the node's GPUs, with it.  Wave-uniform tests on values the wave holds anyway.
#define DX_ST_INDEX 64u
__device__ __forceinline__ bool entry_sane(const qv_args &a, uint64_t r, uint32_t L)
{ if (a.text_bytes == 0) return true;                    // extent of the text unknown to the library
  const uint64_t o = a.off[r], span = 4ull * ((uint64_t) L + a.pad) + L;
  return o <= a.text_bytes && span <= a.text_bytes - o;
}

__global__ __launch_bounds__(DX_BLOCK, ENC_WAVES)
void k_qv_encode(qv_args a, const uint32_t *g_tok, const uint8_t *hdr, const uint64_t *hdr_off,
                 const uint64_t *rec_off, const uint32_t *seg, uint8_t *out, uint32_t *status, uint32_t *ticket,
                 const uint32_t *only_list, const unsigned long long *only_count, uint64_t first_entry,
                 const uint32_t *only_info, uint64_t out_cap, sub_sink sx)
{ __shared__ uint32_t s_tok[6][256];
  __shared__ uint32_t s_stok[6][256];
  __shared__ uint8_t  s_tagcode[256];
  __shared__ __attribute__((aligned(16))) uint32_t s_win[DX_WAVES_PER_BLK][QV_WIN_PAD + QV_WIN_WORDS];   // (the pad: see place_bits128)
  __shared__ __attribute__((aligned(16))) uint32_t s_tag[DX_WAVES_PER_BLK][TAG_WIN_WORDS];
  __shared__ __attribute__((aligned(16))) uint8_t s_chunk[DX_WAVES_PER_BLK][DX_STEP];
  __shared__ __attribute__((aligned(16))) uint8_t s_tchunk[DX_WAVES_PER_BLK][DX_STEP];
  __shared__ uint16_t s_list[DX_WAVES_PER_BLK][DX_STEP];
  load_tables(s_tok, g_tok);
  load_shift_tables(s_stok, s_tagcode, g_tok);
  const int      lane  = lane_id();
  const int      wid   = threadIdx.x >> 6;

  const run_lds R = { s_chunk[wid], s_list[wid] };
  wave_out o, ot;
  o.win  = s_win[wid] + QV_WIN_PAD;
  ot.win = s_tag[wid];
  for (int j = lane; j < QV_WIN_WORDS; j += 64)  o.win[j]  = 0;
  for (int j = lane; j < TAG_WIN_WORDS; j += 64) ot.win[j] = 0;
  wave_sync();

  // Which entries: all of the batch, drawn one by one from the ticket counter -- or, beside
  // k_qv_encode_fast, just those of the list k_qv_hist made of the entries whose tokens are unusable
  // (batch-wide indices, any order; this launch takes the ones in [first_entry, first_entry + n)).  The
  // list is drawn 64 indices at a time, a lane each.
  const uint64_t listed = only_list ? (uint64_t) *only_count : 0;
  uint64_t pend = 0, nxt = only_list ? 0 : next_unit(ticket, a.units), cur = 0, cur_end = 0;
  uint32_t mine = 0;
  for (;;)
    { uint64_t r;
      if (only_list == NULL)                             // (a.units entries per ticket, like everywhere)
        { if (cur >= cur_end)
            { cur = nxt; cur_end = nxt + a.units;
              if (cur >= a.n) break;
              nxt = next_unit(ticket, a.units);
            }
          r = cur++;
          if (r >= a.n) break;
        }
      else
        { while (pend == 0)
            { const uint64_t t = next_unit(ticket, 64u);
              if (t >= listed) break;
              mine = t + (uint64_t) lane < listed ? only_list[t + (uint64_t) lane] : 0xffffffffu;
              pend = __ballot(mine != 0xffffffffu && (uint64_t) mine >= first_entry && (uint64_t) mine - first_entry < a.n);
            }
          if (pend == 0) break;
          const int l = __ffsll((unsigned long long) pend) - 1;
          pend &= pend - 1;
          r = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) mine, l) - first_entry;
          if (!tok_unusable(only_info, r, a.delChar, a.subChar))
            continue;                                    // (listed for a run character the coding dropped: the fast kernel has it)
        }
      const uint32_t  L   = a.len[r];
      const uint32_t *sg  = seg + 5 * r;
      uint8_t        *dst;
      if (!entry_sane(a, r, L))
        { if (lane == 0) atomicOr(status, DX_ST_INDEX);
          continue;
        }
      if (rec_off[r + 1] > out_cap)                      // (dx_qv_encode_onepass: d_out too small) report, never overrun
        { if (lane == 0) atomicOr(status, 8u);
          continue;
        }
      dst = out + rec_off[r];
      if (hdr != NULL)                                   // record framing (dexqv.c:128-139)
        { const uint64_t h0 = hdr_off[r];
          const uint32_t hl = (uint32_t) (hdr_off[r + 1] - h0);
          for (uint32_t k = lane; k < hl; k += 64)
            dst[k] = hdr[h0 + k];
          dst += hl;
        }
      const uint8_t *p1   = line_ptr(a, r, L, 1);
      const bool     over = can_overread(a, line_ptr(a, r, L, 4), L);
      uint32_t bad = 0;

      // The four QV streams in file order: del (with its tag segment right behind it), ins, mrg,
      // sub (QV.c:1393-1423).  One loop, so that each step body exists once in the code.
      #pragma unroll 1
      for (int q = 0; q < 4; q++)
        {
          const int       line = q ? q + 1 : 0;
          const uint8_t  *p    = line_ptr(a, r, L, line);
          const int       rci  = q == 0 ? a.delChar : (q == 3 ? a.subChar : -1);
          const uint32_t *tab  = s_tok[q];
          const uint32_t  want = sg[line];
          const uint32_t  mask = !a.lossy ? 0xffu : (q == 1 ? 0xfeu : (q == 2 ? 0xfcu : 0xffu));   // QV.c:1406-1415
          o.seg = dst; o.wordbase = 0; o.winbits = 0;
          uint32_t got, pos = 16u * lane;

          if (rci >= 0)                                  // Encode_Run; for del also Pack_Tag & co.
            { const uint32_t *rtab = s_tok[q == 0 ? DX_DRUN : DX_SRUN];
              const bool      tags = q == 0;
              uint32_t C = 0;
              if (sx.idx && lane == 0)                   // no group index from this encoder: the lane-per-line decoder takes the line
                { sx.idx[sx.off[r] + run_base(L) + (q == 0 ? 0u : 1u)] = RUN_NONE;
                  atomicAdd(sx.none, 1u);
                }
              ot.seg = dst + want; ot.wordbase = 0; ot.winbits = 0;
              u32x4 c = fetch(p, pos, L, over), t = c;
              if (tags) t = fetch(p1, pos, L, over);
              for (uint32_t base = 0; base < L; base += DX_STEP)
                { const u32x4 d = fetch(p, pos + DX_STEP, L, over);
                  u32x4 u = d;
                  if (tags) u = fetch(p1, pos + DX_STEP, L, over);
                  const uint32_t sv = L - base >= DX_STEP ? DX_STEP : L - base;
                  encode_runs_step(o, ot, R, s_tchunk[wid], s_tagcode, tags, c, t, valid_of(pos, L), sv, (uint32_t) rci, C, tab, rtab,
                                   s_stok[q], s_stok[q == 0 ? DX_DRUN : DX_SRUN]);
                  c = d; t = u;
                  pos += DX_STEP;
                }
              const uint32_t last = C > 0 ? encode_trailing_run(o, C, rtab) : last_piece_plain(tab, p, L, 0xffu);
              got = finish_words(o, last);
              if (tags)
                { const uint32_t tb = finish_tags(ot);
                  bad |= tb ^ sg[1]; dst += sg[1];
                }
            }
          else                                           // Encode
            { const uint32_t m4 = mask * 0x01010101u;
              u32x4 c = fetch(p, pos, L, over);
              // ins and mrg are always plain: with the table at a compile-time LDS address a look-up
              // address is one SDWA shift of the byte (no base to add)
#define PLAIN_LOOP(STAB)                                                                        \
              for (uint32_t base = 0; base < L; base += DX_STEP)                                 \
                { const u32x4 d = fetch(p, pos + DX_STEP, L, over);                              \
                  encode_plain_step(o, c, valid_of(pos, L), L - base >= DX_STEP, tab, STAB, m4, sm); \
                  c = d;                                                                         \
                  pos += DX_STEP;                                                                \
                }
              sub_mark sm;
              sub_begin(sm, sx.idx ? sx.idx + sx.off[r] + (uint64_t) q * sub_words(L) : (uint32_t *) NULL);
              if (q == 1)      { PLAIN_LOOP(s_stok[1]) }
              else if (q == 2) { PLAIN_LOOP(s_stok[2]) }
              else             { PLAIN_LOOP(s_stok[q]) }
#undef PLAIN_LOOP
              got = finish_words(o, last_piece_plain(tab, p, L, mask));
              if (q == 0)                                // no delChar: the whole tag line is packed
                { ot.seg = dst + want; ot.wordbase = 0; ot.winbits = 0;
                  const uint32_t tb = encode_all_tags(ot, p1, L, over);
                  bad |= tb ^ sg[1]; dst += sg[1];
                }
            }
          bad |= got ^ want;
          dst += want;
        }
      if (bad && lane == 0)
        atomicOr(status, 2u);                            // sizes disagree with the size kernel's
    }
}

#include "dx_qv_fast.hpp"
static constexpr auto FAST_K    = &k_qv_encode_fast<false>;
static constexpr auto FAST_K_IX = &k_qv_encode_fast<true>;

// =============================================================================================
//  C-ABI
// =============================================================================================
static int check_batch(dx_ctx *ctx, const dx_qv_batch *b, const char *who)
{ if (ctx == NULL) return DX_E_ARG;
  if (b == NULL) return dx_fail(ctx, DX_E_ARG, "%s: NULL batch", who);
  if (b->n && (!b->d_text || !b->d_off || !b->d_len))
    return dx_fail(ctx, DX_E_ARG, "%s: NULL device pointer in batch", who);
  if (b->n >= (1ull << 31))                              // the kernels hand out entries through a 32-bit counter
    return dx_fail(ctx, DX_E_ARG, "%s: more than 2^31 - 1 entries in one batch", who);
  return DX_OK;
}

static qv_args make_args(const dx_qv_batch *b, int delChar, int subChar, int lossy)
{ qv_args a;
  a.text = b->d_text; a.off = b->d_off; a.len = b->d_len; a.n = b->n; a.pad = b->line_pad;
  a.text_bytes = b->text_bytes;
  a.delChar = delChar; a.subChar = subChar; a.lossy = lossy;
  a.units = ticket_units(b->text_bytes, b->n, TICKET_BATCH * 50000u, TICKET_BATCH);
  return a;
}

int dx_scan_u32(dx_ctx *ctx, const uint32_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *total);
static int fast_grid(dx_ctx *ctx, uint64_t entries);
#include "dx_qv_short.hpp"

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

// The lossy rounding of Compress_Next_QVentry (QV.c:1355-1372: insertion QVs to even values, merge QVs to multiples of four)
// applied in place to the text of a batch: what a lossy .dexqv decodes back to.  One wave per entry, 16 bytes per lane.
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_lossy_text(qv_args a, uint8_t *text)
{ const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const uint32_t L = a.len[r];
      for (int k = 2; k <= 3; k++)
        { uint8_t *p = text + (line_ptr(a, r, L, k) - a.text);
          const uint8_t m = k == 2 ? 0xfeu : 0xfcu;
          for (uint32_t j = (uint32_t) lane_id(); j < L; j += 64u)
            p[j] &= m;
        }
    }
}

extern "C" int dx_qv_lossy_text(dx_ctx *ctx, const dx_qv_batch *b)
{ int e = check_batch(ctx, b, "dx_qv_lossy_text");
  if (e) return e;
  if (b->n == 0) return DX_OK;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  const qv_args a = make_args(b, -1, -1, 1);
  hipLaunchKernelGGL(k_qv_lossy_text, dim3(dx_grid_waves(ctx, b->n, 32)), dim3(DX_BLOCK), 0, ctx->stream, a, (uint8_t *) b->d_text);
  DX_HIP(ctx, hipGetLastError());
  return DX_OK;
}

// Token slots for the batch: offsets by a scan of the per-entry rooms, buffers grown as needed.  Returns
// false (and leaves the hand-over off) when tokens are not wanted or the memory is not to be had.
// sd (dx_qv_scan): the run characters are the device's (p is not looked at), the share and the wanted instance of k_qv_hist go
// to *sd, `guess` is the instance the host is about to launch, and nothing comes back to the host here: the buffers are
// taken as they are (dx_qv_scan has seen to the entries' arrays; whether the tokens fit is k_qv_hist's to say).
static bool tokens_off()
{ return dx_test_on("no_tokens") != 0; }

static bool tokens_prepare(dx_ctx *ctx, const dx_qv_batch *b, const dx_qv_params *p, uint8_t *scr, size_t scr_at,
                           scan_dev *sd = NULL, uint32_t guess = 0)
{ ctx->tk.valid = 0;
  if (sd == NULL && (tokens_off() || (p->delChar < 0 && p->subChar < 0)))
    return false;
  const uint64_t n      = b->n;
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t *d_room = (uint32_t *) (scr + scr_at);
  uint64_t *d_tile = (uint64_t *) (scr + scr_at + ((n * 4 + 63) & ~(size_t) 63));
  uint64_t *d_gran = d_tile + ntiles;
  if (ctx->tk.cap_entries < n)
    { if (sd != NULL) return false;
      (void) hipFree(ctx->tk.off); (void) hipFree(ctx->tk.info); (void) hipFree(ctx->tk.count);
      ctx->tk.off = NULL; ctx->tk.info = NULL; ctx->tk.count = NULL; ctx->tk.list = NULL; ctx->tk.cap_entries = 0;
      if (hipMalloc((void **) &ctx->tk.off, (n + 1) * 8) != hipSuccess ||
          hipMalloc((void **) &ctx->tk.info, n * 4 * TOK_INFO) != hipSuccess ||
          hipMalloc((void **) &ctx->tk.count, 8 + n * 4) != hipSuccess)
        { (void) hipGetLastError(); return false; }
      ctx->tk.list = (uint32_t *) (ctx->tk.count + 1);
      ctx->tk.cap_entries = n;
    }
  { unsigned long long *d_cnt = (unsigned long long *) (ctx->d_u64 + 48);          // (5 words: k_qv_density's four, the share chosen)
    const uint64_t sample = n < 1024 ? n : 1024, stride = n / sample;
    if (hipMemsetAsync(d_cnt, 0, 40, ctx->stream) != hipSuccess) { (void) hipGetLastError(); return false; }
    qv_args a = make_args(b, sd ? -1 : p->delChar, sd ? -1 : p->subChar, 0);
    hipLaunchKernelGGL(k_qv_density, dim3((unsigned) ((sample + DX_WAVES_PER_BLK - 1) / DX_WAVES_PER_BLK)), dim3(DX_BLOCK), 0, ctx->stream,
                       a, stride, d_cnt, (const scan_dev *) sd);
    hipLaunchKernelGGL(k_tok_rooms, dim3((unsigned) ((n + DX_BLOCK - 1) / DX_BLOCK)), dim3(DX_BLOCK), 0, ctx->stream,
                       (const uint32_t *) b->d_len, n, d_cnt, d_room, sd, guess, dx_test_on("hist_shared_tokens"));
  }
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned) ntiles), dim3(DX_BLOCK), 0, ctx->stream, (const uint32_t *) d_room, n, d_tile);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(DX_BLOCK), 0, ctx->stream, d_tile, ntiles, d_gran);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned) ntiles), dim3(DX_BLOCK), 0, ctx->stream, (const uint32_t *) d_room, n,
                     (const uint64_t *) d_tile, ctx->tk.off, (const uint64_t *) d_gran);
  if (sd != NULL)                                        // (share and total: with the histograms)
    return hipGetLastError() == hipSuccess;
  uint64_t total = 0, share = 128;
  if (hipMemcpyAsync(&share, (unsigned long long *) (ctx->d_u64 + 48) + 4, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&total, d_gran, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    { (void) hipGetLastError(); return false; }
  ctx->tk.share8 = (uint32_t) share;
  if (ctx->tk.cap_tokens < total)
    { (void) hipFree(ctx->tk.del); (void) hipFree(ctx->tk.sub);
      ctx->tk.del = NULL; ctx->tk.sub = NULL; ctx->tk.cap_tokens = 0;
      if (hipMalloc((void **) &ctx->tk.del, total * 2 + 64) != hipSuccess ||
          hipMalloc((void **) &ctx->tk.sub, total * 2 + 64) != hipSuccess)
        { (void) hipGetLastError();                   // not enough memory for the hand-over: the generic encoder does it all
          (void) hipFree(ctx->tk.del); ctx->tk.del = NULL;
          return false;
        }
      ctx->tk.cap_tokens = total;
    }
  return true;
}

extern "C" int dx_qv_hist(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, const dx_qv_params *p,
                          uint64_t hist[6][256], uint64_t *totChar)
{ int e = check_batch(ctx, b, "dx_qv_hist");
  if (e) return e;
  if (p == NULL || hist == NULL || totChar == NULL)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_hist: NULL argument");
  if (b->n == 0) return DX_OK;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n      = b->n;
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  const size_t   hbytes = ((6 * 256 + 2) * 8 + 255) & ~(size_t) 255;
  uint8_t *scr;
  { const size_t want = hbytes + ((n * 4 + 63) & ~(size_t) 63) + (ntiles + 2) * 8 + 64;
    if (want > ctx->hscr_bytes)                          // (its own buffer: see dx_ctx.op)
      { DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void) hipFree(ctx->d_hscr);
        ctx->d_hscr = NULL; ctx->hscr_bytes = 0;
        if (hipMalloc(&ctx->d_hscr, want + want / 4) != hipSuccess)
          { (void) hipGetLastError();
            return dx_fail(ctx, DX_E_NOMEM, "dx_qv_hist: no memory for %zu bytes of scratch", want);
          }
        ctx->hscr_bytes = want + want / 4;
      }
    scr = (uint8_t *) ctx->d_hscr;
  }
  unsigned long long *d_hist = (unsigned long long *) scr;
  DX_HIP(ctx, hipMemsetAsync(d_hist, 0, (6 * 256 + 2) * 8, ctx->stream));
  qv_args a = make_args(b, p->delChar, p->subChar, 0);
  qs_verdict qv;
  if ((e = qs_short(ctx, b, true, &qv))) return e;
  if (qv.brief)                                          // short entries: a lane each (dx_qv_short.hpp); no tokens, no counters per entry
    { ctx->tk.valid = 0; ctx->tk.eh_valid = 0;
      DX_LAUNCH(ctx, DX_K_QV_HIST, k_qs_hist, qs_grid(ctx, n), QS_BLOCK, a, qv.deal, entry0, (long long) p->del_first, (long long) p->sub_first,
                d_hist, d_hist + 6 * 256);
      if (!qv.mixed)
        { uint64_t host[6 * 256 + 1];
          DX_HIP(ctx, hipMemcpyAsync(host, d_hist, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
          DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
          for (int s = 0; s < 6; s++)
            for (int k = 0; k < 256; k++)
              hist[s][k] += host[s * 256 + k];
          *totChar += host[6 * 256];
          return DX_OK;
        }
      // ... and the long ones among them (the lanes have skipped them): a batch of their own for the kernels below, which add to the
      // same counters; their tokens, and their own histograms, are that batch's (dx_qv_encode_onepass finds them under its arrays)
      b = &qv.sub;
      a = make_args(b, p->delChar, p->subChar, 0);
    }
  const uint32_t *orig = qv.mixed ? qv.list : (const uint32_t *) NULL;
  const uint64_t  nw   = b->n;                           // the entries the wave-per-entry kernels take
  tok_sink ts = { NULL, NULL, NULL, NULL, d_hist + 6 * 256 + 1, NULL };
  if (tokens_prepare(ctx, b, p, scr, hbytes))
    { ts.del = ctx->tk.del; ts.sub = ctx->tk.sub; ts.off = ctx->tk.off; ts.info = ctx->tk.info; ts.list = ctx->tk.list; }
  uint32_t *d_ticket = (uint32_t *) (ctx->d_u64 + 17);
  DX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
  const uint64_t hist_blocks = (nw + HIST_NWAVE - 1) / HIST_NWAVE, hist_room = (uint64_t) ctx->num_cu * HIST_PER_CU;   // HIST_PER_CU workgroups per CU
  // the FAST instance also leaves every entry's own histograms (for k_qv_sizes_hist): 1.5 KB per entry
  bool fast_hist = ts.del != NULL && p->delChar >= 0 && p->subChar >= 0;
  ctx->tk.eh_valid = 0;
  if (fast_hist && ctx->tk.cap_eh < nw)
    { (void) hipFree(ctx->tk.eh);
      ctx->tk.eh = NULL; ctx->tk.cap_eh = 0;
      if (hipMalloc((void **) &ctx->tk.eh, nw * EH_WORDS * 4 + 64) != hipSuccess)
        { (void) hipGetLastError();                        // (no memory for them: the instance without, and the slot encoder after it)
          fast_hist = false;
        }
      else
        ctx->tk.cap_eh = nw;
    }
  uint32_t inst = SCAN_G;
  if (fast_hist && ctx->tk.share8 <= HIST_SHARE_A && !dx_test_on("hist_shared_tokens"))   // (tokens at most ~28 % of the denser line: run densities from ~0.72 up)
    { inst = SCAN_A;
      DX_LAUNCH(ctx, DX_K_QV_HIST, (k_qv_hist<true, true>), (int) (hist_blocks < hist_room ? hist_blocks : hist_room), HIST_BLOCK,
                a, entry0, (long long) p->del_first, (long long) p->sub_first, d_hist, d_hist + 6 * 256, d_ticket, ts, ctx->tk.eh,
                (const scan_dev *) NULL, (uint64_t) 0, orig);
    }
  else if (fast_hist)
    { inst = SCAN_B;
      DX_LAUNCH(ctx, DX_K_QV_HIST, (k_qv_hist<true, false>), (int) (hist_blocks < hist_room ? hist_blocks : hist_room), HIST_BLOCK,
                a, entry0, (long long) p->del_first, (long long) p->sub_first, d_hist, d_hist + 6 * 256, d_ticket, ts, ctx->tk.eh,
                (const scan_dev *) NULL, (uint64_t) 0, orig);
    }
  else
    DX_LAUNCH(ctx, DX_K_QV_HIST, (k_qv_hist<false, false>), (int) (hist_blocks < hist_room ? hist_blocks : hist_room), HIST_BLOCK,
              a, entry0, (long long) p->del_first, (long long) p->sub_first, d_hist, d_hist + 6 * 256, d_ticket, ts, (uint32_t *) NULL,
              (const scan_dev *) NULL, (uint64_t) 0, orig);
  // what dx_qv_scan may guess for this context's next batch (only instances that leave tokens are guessed)
  ctx->scan.valid = !qv.mixed && ts.del != NULL && (inst != SCAN_G || (p->delChar >= 0) != (p->subChar >= 0));
  ctx->scan.inst  = inst;
  uint64_t host[6 * 256 + 2];
  DX_HIP(ctx, hipMemcpyAsync(host, d_hist, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < 6; s++)
    for (int k = 0; k < 256; k++)
      hist[s][k] += host[s * 256 + k];
  *totChar += host[6 * 256];
  if (ts.del != NULL)                                   // the tokens belong to exactly this batch and scan state
    { ctx->tk.text = b->d_text; ctx->tk.boff = b->d_off; ctx->tk.blen = b->d_len;
      ctx->tk.n = b->n; ctx->tk.text_bytes = b->text_bytes; ctx->tk.pad = b->line_pad;
      ctx->tk.delChar = p->delChar; ctx->tk.subChar = p->subChar;
      ctx->tk.unusable = host[6 * 256 + 1];
      // the generic kernel reads the count on the device: kept in front of the list (the scratch it was counted in is reused)
      DX_HIP(ctx, hipMemcpyAsync(ctx->tk.count, d_hist + 6 * 256 + 1, 8, hipMemcpyDeviceToDevice, ctx->stream));
      ctx->tk.valid = 1;
      ctx->tk.eh_valid = fast_hist ? 1 : 0;
    }
  return DX_OK;
}

// the context's pinned host words (dx_internal.hpp: h_pin)
static uint64_t *scan_pin(dx_ctx *ctx)
{ if (ctx->h_pin == NULL && hipHostMalloc((void **) &ctx->h_pin, (2048 + 768) * 8, hipHostMallocDefault) != hipSuccess)
    { (void) hipGetLastError();
      ctx->h_pin = NULL;
    }
  return ctx->h_pin;
}

// dx_qv_prescan + dx_qv_hist with ONE wait: the scan state stays on the device (scan_dev), the host's two decisions are
// guesses the device checks.  Same results as the two calls, to which it falls back whenever there is nothing to guess from
// (a context's first batch), a guess fails, or the batch may be one for the lane-per-entry kernels.
extern "C" int dx_qv_scan(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, dx_qv_params *p, uint64_t hist[6][256], uint64_t *totChar)
{ int e = check_batch(ctx, b, "dx_qv_scan");
  if (e) return e;
  if (p == NULL || hist == NULL || totChar == NULL)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_scan: NULL argument");
  const uint64_t n      = b->n;
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  const size_t   hbytes = ((6 * 256 + 2) * 8 + 255) & ~(size_t) 255;
  const size_t   want   = hbytes + ((n * 4 + 63) & ~(size_t) 63) + (ntiles + 2) * 8 + 64;
  const uint32_t guess  = ctx->scan.inst;
  bool spec = n > 0 && ctx->scan.valid && !tokens_off() && !dx_test_on("no_scan_guess") &&
              ctx->tk.cap_entries >= n && ctx->tk.cap_tokens > 0 && ctx->tk.del != NULL && ctx->tk.sub != NULL &&
              (guess == SCAN_G || ctx->tk.cap_eh >= n) && want <= ctx->hscr_bytes && scan_pin(ctx) != NULL;
  if (spec && n >= 4096 && b->text_bytes && b->text_bytes / n <= 5ull * (QS_MAXLEN + 1u) + 64u && !dx_test_on("no_short"))
    spec = false;                                        // (qs_short's to look at)
  if (!spec)
    { if ((e = dx_qv_prescan(ctx, b, entry0, p))) return e;
      return dx_qv_hist(ctx, b, entry0, p, hist, totChar);
    }
  DX_HIP(ctx, hipSetDevice(ctx->device));
  uint8_t *scr = (uint8_t *) ctx->d_hscr;
  unsigned long long *d_hist = (unsigned long long *) scr;
  unsigned long long *d_key  = (unsigned long long *) ctx->d_u64;
  long long          *d_sub  = (long long *) (ctx->d_u64 + 2);
  scan_dev           *sd     = (scan_dev *) (ctx->d_u64 + 56);
  uint64_t           *d_gran = (uint64_t *) (scr + hbytes + ((n * 4 + 63) & ~(size_t) 63)) + ntiles;
  const bool want_del = p->delChar < 0, want_sub = p->subChar < 0 && entry0 == 0;
  qv_args a = make_args(b, -1, -1, 0);
  if (want_del)
    { DX_HIP(ctx, hipMemsetAsync(d_key, 0xff, 8, ctx->stream));
      DX_LAUNCH(ctx, DX_K_QV_PRESCAN, k_qv_prescan_del, dx_grid_waves(ctx, n, 8), DX_BLOCK, a, entry0, d_key);
    }
  if (want_sub)
    DX_LAUNCH(ctx, DX_K_QV_PRESCAN, k_qv_prescan_sub, 1, DX_BLOCK, a, d_sub);
  hipLaunchKernelGGL(k_scan_state, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long *) d_key, (const long long *) d_sub, *p,
                     want_del ? 1 : 0, want_sub ? 1 : 0, sd);
  DX_HIP(ctx, hipMemsetAsync(d_hist, 0, (6 * 256 + 2) * 8, ctx->stream));
  ctx->tk.eh_valid = 0;
  if (!tokens_prepare(ctx, b, p, scr, hbytes, sd, guess))
    { (void) hipStreamSynchronize(ctx->stream);
      return dx_fail(ctx, DX_E_HIP, "dx_qv_scan: a launch failed (%s)", hipGetErrorString(hipGetLastError()));
    }
  tok_sink ts = { ctx->tk.del, ctx->tk.sub, ctx->tk.off, ctx->tk.info, d_hist + 6 * 256 + 1, ctx->tk.list };
  uint32_t *d_ticket = (uint32_t *) (ctx->d_u64 + 17);
  DX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
  const uint64_t hist_blocks = (n + HIST_NWAVE - 1) / HIST_NWAVE, hist_room = (uint64_t) ctx->num_cu * HIST_PER_CU;
  const int      grid = (int) (hist_blocks < hist_room ? hist_blocks : hist_room);
  if (guess == SCAN_A)
    DX_LAUNCH(ctx, DX_K_QV_HIST, (k_qv_hist<true, true>), grid, HIST_BLOCK, a, entry0, 0ll, 0ll, d_hist, d_hist + 6 * 256, d_ticket, ts,
              ctx->tk.eh, (const scan_dev *) sd, (uint64_t) ctx->tk.cap_tokens, (const uint32_t *) NULL);
  else if (guess == SCAN_B)
    DX_LAUNCH(ctx, DX_K_QV_HIST, (k_qv_hist<true, false>), grid, HIST_BLOCK, a, entry0, 0ll, 0ll, d_hist, d_hist + 6 * 256, d_ticket, ts,
              ctx->tk.eh, (const scan_dev *) sd, (uint64_t) ctx->tk.cap_tokens, (const uint32_t *) NULL);
  else
    DX_LAUNCH(ctx, DX_K_QV_HIST, (k_qv_hist<false, false>), grid, HIST_BLOCK, a, entry0, 0ll, 0ll, d_hist, d_hist + 6 * 256, d_ticket, ts,
              (uint32_t *) NULL, (const scan_dev *) sd, (uint64_t) ctx->tk.cap_tokens, (const uint32_t *) NULL);
  uint64_t *host = ctx->h_pin;                           // [0, 1538): histograms, total, unusable; [1600): scan_dev; [1610): tokens the slots want
  DX_HIP(ctx, hipMemcpyAsync(host, d_hist, (6 * 256 + 2) * 8, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipMemcpyAsync(host + 1600, sd, sizeof(scan_dev), hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipMemcpyAsync(host + 1610, d_gran, 8, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  scan_dev got;
  memcpy(&got, host + 1600, sizeof(got));
  p->delChar = got.delChar; p->subChar = got.subChar; p->del_first = got.del_first; p->sub_first = got.sub_first;
  if ((got.inst & SCAN_MISS) || host[1610] > ctx->tk.cap_tokens)
    return dx_qv_hist(ctx, b, entry0, p, hist, totChar);  // a guess failed (k_qv_hist has counted nothing): the plain way, the state now known
  for (int s = 0; s < 6; s++)
    for (int k = 0; k < 256; k++)
      hist[s][k] += host[s * 256 + k];
  *totChar += host[6 * 256];
  ctx->tk.share8 = got.share8;
  ctx->tk.text = b->d_text; ctx->tk.boff = b->d_off; ctx->tk.blen = b->d_len;
  ctx->tk.n = b->n; ctx->tk.text_bytes = b->text_bytes; ctx->tk.pad = b->line_pad;
  ctx->tk.delChar = p->delChar; ctx->tk.subChar = p->subChar;
  ctx->tk.unusable = host[6 * 256 + 1];
  DX_HIP(ctx, hipMemcpyAsync(ctx->tk.count, d_hist + 6 * 256 + 1, 8, hipMemcpyDeviceToDevice, ctx->stream));
  ctx->tk.valid = 1;
  ctx->tk.eh_valid = guess != SCAN_G ? 1 : 0;
  return DX_OK;
}

static uint32_t pack_sym(const dx_scheme *s, int x)
{ const uint32_t len = (uint32_t) s->lens[x], bits = s->bits[x];
  if (len == 0) return 0;
  const bool esc = s->type == 2 && bits == s->bits[255] && s->lens[x] == s->lens[255];   // QV.c:432
  return esc ? TOK_PACK((bits << 8) | (uint32_t) x, len + 8u, true) : TOK_PACK(bits, len, false);
}

static uint32_t pack_run(const dx_scheme *s, int x)
{ const uint32_t len = (uint32_t) s->lens[x], bits = s->bits[x];
  const bool esc = bits == s->bits[255] && s->lens[x] == s->lens[255];                   // QV.c:468-469, 486
  return TOK_PACK(bits, len, esc);
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
  // decode side (Read_Scheme's look-up table, QV.c:365-372, in two levels): ascending symbol order
  // so that for codes shared through the escape the last writer, 255, wins
  ctx->h_dec.assign(6 * DX_DEC_SIZE, 0);                          // the context's own: contexts run on their own threads
  ctx->h_lng.assign(6 * (1 + DX_LONG_MAX), 0);
  uint16_t *dec = ctx->h_dec.data();
  uint32_t *lng = ctx->h_lng.data();
  for (int s = 0; s < 6; s++)
    { if ((s == DX_DRUN && c->delChar < 0) || (s == DX_SRUN && c->subChar < 0))
        continue;
      uint32_t *L = lng + s * (1 + DX_LONG_MAX);
      for (int x = 0; x < 256; x++)
        { const int len = c->s[s].lens[x];
          const uint32_t bits = c->s[s].bits[x];
          if (len <= 0) continue;
          if (len <= DX_DEC_BITS)
            { const uint32_t base = bits << (DX_DEC_BITS - len), cnt = 1u << (DX_DEC_BITS - len);
              for (uint32_t j = 0; j < cnt; j++)
                dec[s * DX_DEC_SIZE + ((base + j) & (DX_DEC_SIZE - 1))] = (uint16_t) ((len << 8) | x);
            }
          else
            { const uint32_t pre = (bits << (16 - len)) & 0xffffu;
              uint32_t k;
              for (k = 1; k <= L[0]; k++)                        // same code again: later symbol wins
                if ((L[k] >> 16) == pre && ((L[k] >> 8) & 0xff) == (uint32_t) len)
                  break;
              if (k > L[0]) L[0] = k;
              L[k] = (pre << 16) | ((uint32_t) len << 8) | (uint32_t) x;
            }
        }
    }
  // The encoders read d_tok only: it goes up now, through pinned words and without a wait (the stream orders it in front of
  // whatever uses it; ev[17] says when the words may be written again).  The decode tables go up when a decode wants them
  // (dx_dec_tables): 30 KB and a wait that an encode step -- scan, tables, encode, scan, ... -- never needed.
  DX_HIP(ctx, hipSetDevice(ctx->device));
  if (scan_pin(ctx) != NULL)
    { uint32_t *up = (uint32_t *) (ctx->h_pin + 2048);
      DX_HIP(ctx, hipEventSynchronize(ctx->ev[17]));
      memcpy(up, tok, sizeof(tok));
      DX_HIP(ctx, hipMemcpyAsync(ctx->d_tok, up, sizeof(tok), hipMemcpyHostToDevice, ctx->stream));
      DX_HIP(ctx, hipEventRecord(ctx->ev[17], ctx->stream));
    }
  else
    { DX_HIP(ctx, hipMemcpyAsync(ctx->d_tok, tok, sizeof(tok), hipMemcpyHostToDevice, ctx->stream));
      DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  ctx->dec_stale = 1;
  for (int s = 0; s < 4; s++)
    { ctx->sym_type[s] = c->s[s].type;
      // most bits one symbol position can cost: its own code (with the 8-bit literal of an escape), and
      // in a run-coded line the dearest (run token + symbol)/(run + 1)
      uint32_t maxsym = 0, bound;
      for (int x = 0; x < 256; x++)
        if (TOK_LEN(tok[s * 256 + x]) > maxsym) maxsym = TOK_LEN(tok[s * 256 + x]);
      bound = maxsym;
      const int rs = s == DX_DEL ? (c->delChar >= 0 ? DX_DRUN : -1) : (s == DX_SUB ? (c->subChar >= 0 ? DX_SRUN : -1) : -1);
      if (rs >= 0)
        { bound = 0;
          for (uint32_t r = 0; r < 256; r++)
            { const uint32_t t = tok[rs * 256 + r], bits = TOK_LEN(t) + (TOK_ESC(t) ? 16u : 0u) + maxsym;
              const uint32_t per = (bits + r) / (r + 1);
              if (per > bound) bound = per;
            }
        }
      ctx->bps[s] = bound;
    }
  ctx->tok_wide = 0;                                     // a symbol token of more than 16 bits (an escape and its literal)?
  for (int i = 0; i < 4 * 256; i++)
    if (TOK_LEN(tok[i]) > 16u) ctx->tok_wide = 1;
  for (int s = 0; s < 2; s++)                            // ins, mrg: band of coded byte values for the pair tables
    { int lo = -1, hi = -1;                                // (k_qv_encode_fast: two symbols per look-up)
      for (int x = 0; x < 256; x++)
        if (c->s[DX_INS + s].lens[x] > 0)
          { if (lo < 0) lo = x;
            hi = x;
          }
      ctx->pair_lo[s] = (lo >= 0 && lo <= 192 && hi - lo < 64 && !dx_test_on("no_pairs")) ? (uint32_t) lo : 0xffffffffu;
    }
  ctx->coding_set = 1;
  ctx->lossy   = lossy != 0;
  ctx->delChar = c->delChar;
  ctx->subChar = c->subChar;
  ctx->sx.valid = 0;                                     // (a group index belongs to a stream of the tables before)
  if (ctx->op.pending) ctx->op.sx_idx = NULL;            // ... also the one an encode that has begun would arm at its end
  return DX_OK;
}

int dx_dec_tables(dx_ctx *ctx)
{ if (!ctx->dec_stale) return DX_OK;
  DX_HIP(ctx, hipMemcpyAsync(ctx->d_dec, ctx->h_dec.data(), ctx->h_dec.size() * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
  DX_HIP(ctx, hipMemcpyAsync(ctx->d_long, ctx->h_lng.data(), ctx->h_lng.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));        // (pageable sources: the context may rebuild them at once)
  ctx->dec_stale = 0;
  return DX_OK;
}

// exclusive scan of n uint32 values into n+1 uint64 offsets (d_out[n] = total)
int dx_scan_u32(dx_ctx *ctx, const uint32_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *total)
{ if (n == 0)
    { uint64_t z = 0;
      DX_HIP(ctx, hipMemcpyAsync(d_out, &z, 8, hipMemcpyHostToDevice, ctx->stream));
      DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (total) *total = 0;
      return DX_OK;
    }
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (ntiles + 2 > ctx->scan_words)                      // grow-only: no allocation per call
    { (void) hipFree(ctx->d_scan);
      ctx->d_scan = NULL; ctx->scan_words = 0;
      DX_HIP(ctx, hipMalloc((void **) &ctx->d_scan, (ntiles + 2) * 2 * 8));
      ctx->scan_words = (ntiles + 2) * 2;
    }
  uint64_t *d_tile = ctx->d_scan, *d_gran = d_tile + ntiles;
  DX_LAUNCH(ctx, DX_K_SCAN, k_scan_tiles, (int) ntiles, DX_BLOCK, d_in, n, d_tile);
  DX_LAUNCH(ctx, DX_K_SCAN, k_scan_sums, 1, DX_BLOCK, d_tile, ntiles, d_gran);
  DX_LAUNCH(ctx, DX_K_SCAN, k_scan_apply, (int) ntiles, DX_BLOCK, d_in, n, (const uint64_t *) d_tile, d_out,
            (const uint64_t *) d_gran);
  uint64_t t = 0;
  DX_HIP(ctx, hipMemcpyAsync(&t, d_gran, 8, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (total) *total = t;
  return DX_OK;
}

extern "C" int dx_qv_sizes(dx_ctx *ctx, const dx_qv_batch *b, const uint64_t *d_hdr_off,
                           uint32_t *d_seg, uint64_t *d_rec_off, uint64_t *total)
{ int e = check_batch(ctx, b, "dx_qv_sizes");
  if (e) return e;
  if (!ctx->coding_set) return dx_fail(ctx, DX_E_ARG, "dx_qv_sizes: call dx_qv_set_coding first");
  if (ctx->op.pending)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_sizes: an encode has begun in this context: end it first (dx_qv_encode_onepass_end)");
  if (d_rec_off == NULL || (b->n && d_seg == NULL))
    return dx_fail(ctx, DX_E_ARG, "dx_qv_sizes: NULL d_seg / d_rec_off");
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
  uint32_t *d_ticket = (uint32_t *) (ctx->d_u64 + 18);
  qs_verdict qv;
  if ((e = qs_short(ctx, b, false, &qv))) return e;
  const bool brief = qv.brief && !qv.mixed;              // (a mixed batch through this two-pass API: the wave-per-entry kernel for all of it)
  const qs_deal perm = qv.deal;
  DX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
  if (brief)
    DX_LAUNCH(ctx, DX_K_QV_SIZES, (k_qs_entries<false, false>), qs_grid(ctx, n), QS_BLOCK, a, perm, (const uint32_t *) ctx->d_tok, (const uint8_t *) NULL, d_hdr_off,
              (const uint64_t *) NULL, d_seg, d_size, (uint8_t *) NULL, ctx->d_status);
  else
  DX_LAUNCH(ctx, DX_K_QV_SIZES, k_qv_sizes, dx_grid_waves(ctx, n, 4 * SIZES_WAVES), DX_BLOCK,
            a, (const uint32_t *) ctx->d_tok, d_hdr_off, d_seg, d_size, d_ticket,
            (const uint32_t *) NULL, (const unsigned long long *) NULL, (uint64_t) 0, (const uint32_t *) NULL);
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

// every record of the batch from the text, in place (sizes and offsets given): the lane-per-entry kernels for a batch of short
// entries, else k_qv_encode -- with the plain lines' group index when sx_idx is given (the wave-per-entry kernel only)
static int encode_text(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                       const uint64_t *d_rec_off, const uint32_t *d_seg, uint8_t *d_out, uint64_t out_cap, uint32_t *sx_idx)
{ int e;
  DX_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
  qv_args a = make_args(b, ctx->delChar, ctx->subChar, ctx->lossy);
  uint32_t *d_ticket = (uint32_t *) (ctx->d_u64 + 19);
  qs_verdict qv;
  if ((e = qs_short(ctx, b, false, &qv))) return e;
  const bool brief = qv.brief && !qv.mixed && sx_idx == NULL;      // (the lane-per-entry kernels leave no index, and a mixed batch's long entries are not theirs)
  const qs_deal perm = qv.deal;
  DX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
  if (brief && ctx->tok_wide)
    DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, (k_qs_entries<true, true>), qs_grid(ctx, b->n), QS_BLOCK, a, perm, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
              d_rec_off, (uint32_t *) d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  else if (brief)
    DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, (k_qs_entries<true, false>), qs_grid(ctx, b->n), QS_BLOCK, a, perm, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
              d_rec_off, (uint32_t *) d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  else
  DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, k_qv_encode, dx_grid_waves(ctx, b->n, 4 * ENC_WAVES), DX_BLOCK,
            a, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off, d_rec_off, d_seg, d_out, ctx->d_status, d_ticket,
            (const uint32_t *) NULL, (const unsigned long long *) NULL, (uint64_t) 0, (const uint32_t *) NULL, out_cap,
            sub_sink{ sx_idx, sx_idx ? (const uint64_t *) ctx->sx.off : (const uint64_t *) NULL, ctx->sx.none });
  uint32_t st = 0;
  DX_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st & DX_ST_INDEX)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode: an entry's offset and length reach beyond text_bytes");
  if (st & 8u)
    return dx_fail(ctx, DX_E_SPACE, "dx_qv_encode: the record stream does not fit d_out (%llu bytes)", (unsigned long long) out_cap);
  if (st & 2u)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode: a segment's size differs from what dx_qv_sizes "
                                       "computed (d_seg / coding do not belong to this batch?)");
  return DX_OK;
}

extern "C" int dx_qv_encode(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                            const uint64_t *d_rec_off, const uint32_t *d_seg, uint8_t *d_out)
{ int e = check_batch(ctx, b, "dx_qv_encode");
  if (e) return e;
  if (!ctx->coding_set) return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: call dx_qv_set_coding first");
  if (ctx->op.pending)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: an encode has begun in this context: end it first (dx_qv_encode_onepass_end)");
  if ((d_hdr == NULL) != (d_hdr_off == NULL))
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: d_hdr and d_hdr_off must be given together");
  if (b->n == 0) return DX_OK;
  if (!d_rec_off || !d_seg || !d_out) return dx_fail(ctx, DX_E_ARG, "dx_qv_encode: NULL device pointer");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->sx.out == (const void *) d_out) ctx->sx.valid = 0;            // (the two-pass encoder leaves no group index)
  return encode_text(ctx, b, d_hdr, d_hdr_off, d_rec_off, d_seg, d_out, ~(uint64_t) 0, (uint32_t *) NULL);
}

// grid of k_qv_encode_fast: FAST_BLOCK-thread workgroups, 4 * FAST_WAVES waves per CU, no more waves than entries
static int fast_grid(dx_ctx *ctx, uint64_t entries)
{ const uint64_t per_cu = (uint64_t) (4 * FAST_WAVES * 64 / FAST_BLOCK);
  uint64_t g = (uint64_t) ctx->num_cu * (per_cu ? per_cu : 1);
  const uint64_t need = (entries + FAST_NWAVE - 1) / FAST_NWAVE;
  if (need < g) g = need;
  return (int) (g ? g : 1);
}

#define ONEPASS_MAX_GROUPS 64
// ---- group index: room for it -----------------------------------------------------------------------------
__global__ __launch_bounds__(DX_BLOCK)
void k_sub_rooms(const uint32_t *len, uint64_t n, const uint32_t *info /* token counts (k_qv_hist), or NULL */, uint32_t *room)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t passes = 0;                                   // group words only where k_qv_encode_fast will write them
  if (info != NULL)
    { const uint32_t d = info[TOK_INFO * i], s = info[TOK_INFO * i + 1];
      passes = ((d & TOK_BAD) ? 0u : run_passes(d)) + ((s & TOK_BAD) ? 0u : run_passes(s));
    }
  room[i] = run_base(len[i]) + 3u + 64u * passes;
}

extern "C" int dx_qv_subindex(dx_ctx *ctx, int on)
{ if (ctx == NULL) return DX_E_ARG;
  ctx->sx.want = on != 0;
  if (!on) ctx->sx.valid = 0;
  return DX_OK;
}

// offsets (a scan of the rooms) and the index buffer for this batch; *idx = NULL when no index is wanted
static bool onepass_tokens_ok(const dx_ctx *ctx, const dx_qv_batch *b);
static int subindex_prepare(dx_ctx *ctx, const dx_qv_batch *b, const void *d_out, const void *d_seg, uint32_t **idx)
{ *idx = NULL;
  dx_sx_drop_external(ctx);                              // (a caller's index belongs to another stream)
  ctx->sx.valid = 0;
  if (!ctx->sx.want || b->n == 0) return DX_OK;
  const uint64_t n = b->n;
  if (n > ctx->sx.cap_entries)
    { (void) hipFree(ctx->sx.off); (void) hipFree(ctx->sx.room);
      ctx->sx.off = NULL; ctx->sx.room = NULL; ctx->sx.cap_entries = 0;
      if (hipMalloc((void **) &ctx->sx.off, (n + 1) * 8) != hipSuccess || hipMalloc((void **) &ctx->sx.room, n * 4) != hipSuccess)
        { (void) hipGetLastError();
          return dx_fail(ctx, DX_E_NOMEM, "dx_qv_subindex: no memory for the offsets of %llu entries", (unsigned long long) n);
        }
      ctx->sx.cap_entries = n;
    }
  DX_LAUNCH(ctx, DX_K_SCAN, k_sub_rooms, (int) ((n + DX_BLOCK - 1) / DX_BLOCK), DX_BLOCK, (const uint32_t *) b->d_len, n,
            onepass_tokens_ok(ctx, b) ? (const uint32_t *) ctx->tk.info : (const uint32_t *) NULL, ctx->sx.room);
  uint64_t words = 0;
  int e = dx_scan_u32(ctx, ctx->sx.room, n, ctx->sx.off, &words);
  if (e) return e;
  if (words + 4 > ctx->sx.cap_idx)
    { (void) hipFree(ctx->sx.idx);
      ctx->sx.idx = NULL; ctx->sx.cap_idx = 0;
      if (hipMalloc((void **) &ctx->sx.idx, (words + 4) * 4) != hipSuccess)
        { (void) hipGetLastError();
          return dx_fail(ctx, DX_E_NOMEM, "dx_qv_subindex: no memory for %llu index words", (unsigned long long) words);
        }
      ctx->sx.cap_idx = words + 4;
    }
  if (ctx->sx.none == NULL && hipMalloc((void **) &ctx->sx.none, 64) != hipSuccess)
    { (void) hipGetLastError();
      return dx_fail(ctx, DX_E_NOMEM, "dx_qv_subindex: no memory");
    }
  DX_HIP(ctx, hipMemsetAsync(ctx->sx.none, 0, 4, ctx->stream));
  ctx->sx.out = d_out; ctx->sx.seg = d_seg; ctx->sx.n = n;
  *idx = ctx->sx.idx;
  return DX_OK;
}

// The token hand-over applies when k_qv_hist made its tokens for exactly this batch under the run characters now in
// force (a substitution run character dropped by Create_QVcoding, QV.c:1044, just leaves its tokens unused).
static bool onepass_tokens_ok(const dx_ctx *ctx, const dx_qv_batch *b)
{ return ctx->tk.valid && ctx->tk.text == (const void *) b->d_text && ctx->tk.boff == (const void *) b->d_off &&
         ctx->tk.blen == (const void *) b->d_len && ctx->tk.n == b->n && ctx->tk.text_bytes == b->text_bytes &&
         ctx->tk.pad == b->line_pad && ctx->tk.delChar == ctx->delChar &&
         (ctx->subChar < 0 || ctx->subChar == ctx->tk.subChar) && !tokens_off();
}

// dx_qv_encode_onepass with the token hand-over and no scratch slots (DEXGPU_DIRECT_ENCODE, or no memory for the
// slots).  The entries go in a few groups of growing size; on the side stream k_qv_sizes_fast (+ the scan) runs
// ahead through all groups; on the context's stream k_qv_encode_fast writes group g's records in place as soon
// as that group's offsets exist.  Only the first, small group's sizes are waited for with nothing to do.
// Entries with unusable tokens take the generic kernels (sizes and encode from the text).
static int onepass_direct(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                          uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap, uint64_t *total, bool by_hist)
{ const uint64_t n = b->n;
  const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  const size_t   a4 = (n * 4 + 255) & ~(size_t) 255;
  uint8_t *scr;
  int e;
  if ((e = dx_scratch(ctx, a4 + (ntiles + 2) * 8 + 512, (void **) &scr))) return e;
  uint32_t *d_size = (uint32_t *) scr;
  uint64_t *d_tile = (uint64_t *) (scr + a4), *d_gran = d_tile + ntiles;

  uint64_t gb[ONEPASS_MAX_GROUPS + 1];
  int      G = 0;
  gb[0] = 0;
  if (dx_test_str("onepass_groups") != NULL)             // (experiments and tests: that many equal groups)
    { int k = (int) dx_test_num("onepass_groups", 1);
      if (k < 1) k = 1;
      if (k > ONEPASS_MAX_GROUPS) k = ONEPASS_MAX_GROUPS;
      const uint64_t gs = (n + (uint64_t) k - 1) / (uint64_t) k;
      for (uint64_t at = 0; at < n; at += gs)
        gb[++G] = at + gs < n ? at + gs : n;
    }
  else if (n < 160000 || by_hist)                        // (sizes from the entries' own histograms take a moment: one group -- measured in
    gb[++G] = n;                                         //  round 6: a sixteenth first and the rest's sizes beside its encoder, 12.3 against 12.2 ms)
  else                                                   // 1/16 of the batch, then 2.5 x the one before: a group's sizes
    { uint64_t size = n / 16, at = 0;                    // are ready before the encoder has finished the group before it
      if (size < 40000) size = 40000;
      while (G < ONEPASS_MAX_GROUPS - 1 && n - at > size + size / 2)
        { at += size; gb[++G] = at;
          size = size * 5 / 2;
        }
      gb[++G] = n;
    }

  uint32_t *sx_idx = NULL;
  if ((e = subindex_prepare(ctx, b, d_out, d_seg, &sx_idx))) return e;
  hipStream_t A = ctx->stream, B = ctx->side;
  hipEvent_t *sz_done = ctx->ev, fork = ctx->ev[16], join = ctx->ev[15];
  uint64_t   *d_base = ctx->d_u64 + 24;                  // [0], [1]: running record offset, ping-pong
  uint32_t   *d_tick_enc = (uint32_t *) (ctx->d_u64 + 19), *d_tick_sz = (uint32_t *) (ctx->d_u64 + 22);
  const qv_args a = make_args(b, ctx->delChar, ctx->subChar, ctx->lossy);
  const bool    odd = ctx->tk.unusable > 0;              // some entries need the generic kernels
  DX_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, A));
  DX_HIP(ctx, hipMemsetAsync(d_base, 0, 16, A));
  DX_HIP(ctx, hipEventRecord(fork, A));
  DX_HIP(ctx, hipStreamWaitEvent(B, fork, 0));
  int rc = DX_OK;
  for (int g = 0; g < G && rc == DX_OK; g++)             // side stream: sizes and record offsets of every group, in order
    { const uint64_t g0 = gb[g], m = gb[g + 1] - g0, mt = (m + SCAN_TILE - 1) / SCAN_TILE;
      qv_args ag = a;
      ag.off = a.off + g0; ag.len = a.len + g0; ag.n = m;
      const uint64_t *hoff_g = d_hdr_off ? d_hdr_off + g0 : NULL;
      const sub_sink  sx_g   = { sx_idx, sx_idx ? (const uint64_t *) (ctx->sx.off + g0) : (const uint64_t *) NULL, ctx->sx.none };
      const tok_src   tg = { ctx->tk.del, ctx->tk.sub, ctx->tk.off + g0, ctx->tk.info + TOK_INFO * g0 };
      rc = DX_E_HIP;
      if (hipMemsetAsync(d_tick_sz, 0, 4, B) != hipSuccess) break;
      dx_prof_begin_on(ctx, DX_K_QV_SIZES, B);
      if (by_hist)
        hipLaunchKernelGGL(k_qv_sizes_hist, dim3(dx_grid_waves(ctx, (m + 7) / 8, 32)), dim3(DX_BLOCK), 0, B,
                           ag, (const uint32_t *) ctx->d_tok, hoff_g, d_seg + 5 * g0, d_size + g0, tg,
                           (const uint32_t *) (ctx->tk.eh + g0 * EH_WORDS), ctx->tk.subChar);
      else
        hipLaunchKernelGGL(k_qv_sizes_fast, dim3(fast_grid(ctx, (m + TICKET_BATCH - 1) / TICKET_BATCH)), dim3(FAST_BLOCK), 0, B,
                           ag, (const uint32_t *) ctx->d_tok, hoff_g, d_seg + 5 * g0, d_size + g0, d_tick_sz, tg);
      dx_prof_end_on(ctx, B);
      if (odd)
        { if (hipMemsetAsync(d_tick_sz, 0, 4, B) != hipSuccess) break;
          dx_prof_begin_on(ctx, DX_K_QV_SIZES, B);
          hipLaunchKernelGGL(k_qv_sizes, dim3(dx_grid_waves(ctx, ctx->tk.unusable < m ? ctx->tk.unusable : m, 4 * SIZES_WAVES)),
                             dim3(DX_BLOCK), 0, B, ag, (const uint32_t *) ctx->d_tok, hoff_g, d_seg + 5 * g0, d_size + g0, d_tick_sz,
                             (const uint32_t *) ctx->tk.list, (const unsigned long long *) ctx->tk.count, g0,
                             (const uint32_t *) (ctx->tk.info + TOK_INFO * g0));
          dx_prof_end_on(ctx, B);
        }
      dx_prof_begin_on(ctx, DX_K_SCAN, B);
      hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned) mt), dim3(DX_BLOCK), 0, B, (const uint32_t *) (d_size + g0), m, d_tile);
      hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(DX_BLOCK), 0, B, d_tile, mt, d_gran);
      hipLaunchKernelGGL(k_scan_apply_base, dim3((unsigned) mt), dim3(DX_BLOCK), 0, B, (const uint32_t *) (d_size + g0), m,
                         (const uint64_t *) d_tile, d_rec_off + g0, (const uint64_t *) d_gran,
                         (const uint64_t *) (d_base + (g & 1)), d_base + ((g + 1) & 1));
      dx_prof_end_on(ctx, B);
      if (hipGetLastError() != hipSuccess || hipEventRecord(sz_done[g & 7], B) != hipSuccess) break;
      // context's stream: the group's records, in place
      if (hipStreamWaitEvent(A, sz_done[g & 7], 0) != hipSuccess || hipMemsetAsync(d_tick_enc, 0, 4, A) != hipSuccess) break;
      dx_prof_begin_on(ctx, DX_K_QV_ENCODE, A);
      if (sx_idx)
        hipLaunchKernelGGL(FAST_K_IX, dim3(fast_grid(ctx, m)), dim3(FAST_BLOCK), 0, A,
                           ag, (const uint32_t *) ctx->d_tok, hoff_g, ctx->d_status, d_tick_enc, tg,
                           ctx->pair_lo[0], ctx->pair_lo[1], d_hdr, (const uint64_t *) (d_rec_off + g0),
                           (const uint32_t *) (d_seg + 5 * g0), d_out, out_cap, sx_g);
      else
        hipLaunchKernelGGL(FAST_K, dim3(fast_grid(ctx, m)), dim3(FAST_BLOCK), 0, A,
                           ag, (const uint32_t *) ctx->d_tok, hoff_g, ctx->d_status, d_tick_enc, tg,
                           ctx->pair_lo[0], ctx->pair_lo[1], d_hdr, (const uint64_t *) (d_rec_off + g0),
                           (const uint32_t *) (d_seg + 5 * g0), d_out, out_cap, sx_g);
      dx_prof_end_on(ctx, A);
      if (odd)
        { if (hipMemsetAsync(d_tick_enc, 0, 4, A) != hipSuccess) break;
          dx_prof_begin_on(ctx, DX_K_QV_ENCODE_TEXT, A);
          hipLaunchKernelGGL(k_qv_encode, dim3(dx_grid_waves(ctx, ctx->tk.unusable < m ? ctx->tk.unusable : m, 4 * ENC_WAVES)),
                             dim3(DX_BLOCK), 0, A, ag, (const uint32_t *) ctx->d_tok, d_hdr, hoff_g,
                             (const uint64_t *) (d_rec_off + g0), (const uint32_t *) (d_seg + 5 * g0), d_out, ctx->d_status,
                             d_tick_enc, (const uint32_t *) ctx->tk.list,
                             (const unsigned long long *) ctx->tk.count, g0, (const uint32_t *) (ctx->tk.info + TOK_INFO * g0), out_cap, sx_g);
          dx_prof_end_on(ctx, A);
        }
      if (hipGetLastError() != hipSuccess) break;
      rc = DX_OK;
    }
  (void) hipEventRecord(join, B);
  (void) hipStreamWaitEvent(A, join, 0);                 // the caller's stream sees everything finished
  uint64_t tot = 0;
  uint32_t st  = 0;
  if (rc != DX_OK)
    { (void) hipStreamSynchronize(A);
      return dx_fail(ctx, DX_E_HIP, "dx_qv_encode_onepass: a launch failed (%s)", hipGetErrorString(hipGetLastError()));
    }
  if (hipMemcpyAsync(&tot, d_base + (G & 1), 8, hipMemcpyDeviceToHost, A) != hipSuccess ||
      hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, A) != hipSuccess ||
      hipStreamSynchronize(A) != hipSuccess)
    return dx_fail(ctx, DX_E_HIP, "dx_qv_encode_onepass: reading back the totals failed");
  if (total) *total = tot;
  ctx->route.groups = 0; ctx->route.direct = by_hist ? 2 : 1; ctx->route.tokens = 1; ctx->route.region_bytes = 0;
  ctx->route.scratch_bytes = ctx->scratch_bytes; ctx->route.token_bytes = 4ull * ctx->tk.cap_tokens;
  ctx->route.text_entries = ctx->tk.unusable;
  if (tot > out_cap || (st & 8u))
    return dx_fail(ctx, DX_E_SPACE, "dx_qv_encode_onepass: the record stream needs %llu bytes, d_out holds %llu",
                   (unsigned long long) tot, (unsigned long long) out_cap);
  if (st & 2u)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode_onepass: an encoded segment differs in size from what the size kernel computed");
  ctx->sx.valid = sx_idx != NULL;
  return DX_OK;
}

static int onepass_end(dx_ctx *ctx, uint64_t *total);

// wait = false: everything is queued and the function returns; onepass_end collects the total and the status
static int onepass_impl(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                        uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap, uint64_t *total, bool wait)
{ int e = check_batch(ctx, b, "dx_qv_encode_onepass");
  if (e) return e;
  if (ctx->op.pending)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode_onepass: an encode has begun in this context: end it first (dx_qv_encode_onepass_end)");
  if (!ctx->coding_set) return dx_fail(ctx, DX_E_ARG, "dx_qv_encode_onepass: call dx_qv_set_coding first");
  if ((d_hdr == NULL) != (d_hdr_off == NULL))
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode_onepass: d_hdr and d_hdr_off must be given together");
  if (d_rec_off == NULL || (b->n && (!d_seg || !d_out)))
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode_onepass: NULL device pointer");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n = b->n;
  if (n == 0)
    { uint64_t z = 0;
      DX_HIP(ctx, hipMemcpyAsync(d_rec_off, &z, 8, hipMemcpyHostToDevice, ctx->stream));
      DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (total) *total = 0;
      if (!wait) { ctx->op.pending = 1; ctx->op.direct = 1; ctx->op.rc = DX_OK; ctx->op.total = 0; }
      return DX_OK;
    }
  // Three routes, all of which write every record where it belongs.  A batch of short entries: the lane-per-entry kernels
  // (dx_qv_short.hpp).  A batch whose tokens this context's histogram pass has left: k_qv_encode_fast, its sizes from the
  // entries' own histograms (k_qv_sizes_hist: a dot product) or, where there are none (DEXGPU_TEST=sizes_from_tokens; one
  // run character only), from tokens and plain lines (k_qv_sizes_fast).  Anything else: sizes and records from the text.
  { qs_verdict qv;
    if ((e = qs_short(ctx, b, false, &qv))) return e;
    if (qv.brief)
      { uint64_t t = 0;
        const int rc = qv.mixed ? onepass_mixed(ctx, b, &qv, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap, &t)
                                : onepass_short(ctx, b, qv.deal, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap, &t);
        if (total) *total = t;
        if (!wait) { ctx->op.pending = 1; ctx->op.direct = 1; ctx->op.rc = rc; ctx->op.total = t; return DX_OK; }   // (this route does not pipeline)
        return rc;
      }
  }
  if (onepass_tokens_ok(ctx, b))
    { uint64_t t = 0;
      const int rc = onepass_direct(ctx, b, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap, &t, ctx->tk.eh_valid && !dx_test_on("sizes_from_tokens"));
      if (total) *total = t;
      if (!wait) { ctx->op.pending = 1; ctx->op.direct = 1; ctx->op.rc = rc; ctx->op.total = t; return DX_OK; }   // (this route does not pipeline)
      return rc;
    }

  // No tokens of this batch (its histogram pass was another context's, or its scan state has changed since): sizes and
  // records from the text, in place, by the generic kernels -- dx_qv_sizes + dx_qv_encode.  (Rounds 2-4 had scratch slots
  // and a compaction kernel here, and for every batch before the entries' own histograms made the sizes a dot product:
  // taken out in round 6.)
  { uint64_t t = 0;
    uint32_t *sx_idx = NULL;
    int rc = dx_qv_sizes(ctx, b, d_hdr_off, d_seg, d_rec_off, &t);
    if (rc == DX_OK && t > out_cap)
      rc = dx_fail(ctx, DX_E_SPACE, "dx_qv_encode_onepass: the record stream needs %llu bytes, d_out holds %llu",
                   (unsigned long long) t, (unsigned long long) out_cap);
    if (rc == DX_OK) rc = subindex_prepare(ctx, b, d_out, d_seg, &sx_idx);
    if (rc == DX_OK) rc = encode_text(ctx, b, d_hdr, d_hdr_off, d_rec_off, d_seg, d_out, out_cap, sx_idx);
    if (total) *total = t;
    ctx->route.groups = 0; ctx->route.direct = 4; ctx->route.tokens = 0; ctx->route.region_bytes = 0;
    ctx->route.scratch_bytes = ctx->scratch_bytes; ctx->route.token_bytes = 0; ctx->route.text_entries = n;
    if (rc == DX_OK) ctx->sx.valid = sx_idx != NULL;
    if (!wait) { ctx->op.pending = 1; ctx->op.direct = 1; ctx->op.rc = rc; ctx->op.total = t; return DX_OK; }
    return rc;
  }
}

// the other half of dx_qv_encode_onepass_begin: the total and the verdict (every route has waited for its records: what
// "begin" leaves for "end" is the answer)
static int onepass_end(dx_ctx *ctx, uint64_t *total)
{ if (!ctx->op.pending)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_encode_onepass_end: no encode has begun in this context");
  ctx->op.pending = 0;
  if (total) *total = ctx->op.total;
  return ctx->op.rc;
}

extern "C" int dx_qv_onepass_info(const dx_ctx *ctx, dx_onepass_info *out)
{ if (ctx == NULL || out == NULL) return DX_E_ARG;
  *out = ctx->route;
  return DX_OK;
}

extern "C" int dx_qv_encode_onepass(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                                    uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap, uint64_t *total)
{ return onepass_impl(ctx, b, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap, total, true); }

extern "C" int dx_qv_encode_onepass_begin(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                                          uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap)
{ return onepass_impl(ctx, b, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap, NULL, false); }

extern "C" int dx_qv_encode_onepass_end(dx_ctx *ctx, uint64_t *total)
{ if (ctx == NULL) return DX_E_ARG;
  return onepass_end(ctx, total);
}
