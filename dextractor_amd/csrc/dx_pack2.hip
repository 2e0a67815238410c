// dx_pack2.hip -- 2-bit packers behind dexta/undexta and dexar/undexar.
//
// Reference behaviour reproduced (bit-exact):
//   Number_Read / Number_Arrow   DB.c:393-441   ASCII -> 0..3
//   Compress_Read                DB.c:319-338   4 symbols per byte, first in the top two bits
//   Uncompress_Read              DB.c:342-363
//   Lower_Read/Upper_Read/Letter_Arrow DB.c:367-389
//   sequence-line gathering      dexta.c:161-183   (newlines dropped on the device)
//   line wrapping                undexta.c:263-270
//
// Roofline: HBM.  Algorithmic bytes per symbol: encode 1 (+1/80 for the newlines of 80-column
// text) read + 0.25 written; decode 0.25 read + 1 (+1/80) written.
//
// Layout: one 64-lane wavefront per read; the waves draw reads from a ticket counter, 16 at a
// time.  A wave walks the read's text 2 KiB per step (32 bytes per lane, two unaligned
// global_load_dwordx4), maps the bytes through a 256-byte LDS table, deletes the 2-bit slots of
// the '\n' bytes, ORs each lane's <=32 code bits into a small LDS word window at the bit position
// a wave prefix sum of the kept counts gives, and drains the window in 16-byte units.
#include "dx_internal.hpp"
#include "dx_device.hpp"

#define P2_STEP       2048u              // text bytes a wave consumes per step: two 16-byte chunks per lane
#define P2_WIN_WORDS  448                // per-wave LDS window: flush threshold + one step (128 words) + slack
#define P2_FLUSH_BITS 8192u
#define P2_BATCH      16u                // reads a wave draws at a time (one same-address atomic costs ~11 ns chip-wide)
#ifndef P2_FLAT
#define P2_FLAT       0                  // > 0: no ticket counter -- as many waves as there are P2_FLAT reads, each takes its own and leaves
#endif                                   //      (the hardware's dispatcher deals the workgroups as the CUs fall free)
#if P2_FLAT
#define P2_FIRST(TICKET, TB) (((uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6)) * (TB))
#define P2_NEXT(TICKET, TB)  (~0ull)
#define P2_TB(TICKET)        ((uint32_t) P2_FLAT)
#else
#define P2_FIRST(TICKET, TB) next_unit(TICKET, TB)
#define P2_NEXT(TICKET, TB)  next_unit(TICKET, TB)
#define P2_TB(TICKET)        ticket_units_of(TICKET, P2_BATCH)
#endif
#ifndef P2D_UNROLL
#define P2D_UNROLL    1u                 // k_pack2_decode: 1 KiB steps whose loads go out together (measured, 10 M x 10 kb, ms per launch:
                                         // 1: 32.8-35.5, 2: 33.8-35.9, 4: 39.0-39.3, 8: 41-42, 4 + non-temporal stores: 37.7-39.1 --
                                         // profiles/r03_ab_pack2.txt; more stores in flight per wave do not help this write-bound kernel)
#endif
#ifndef P2D_PIPE
#define P2D_PIPE      3                  // k_pack2_decode: hand-pipelined main loop, the requests this many steps ahead (0: every step through the checked loop)
#endif
#ifndef P2E_OVER
#define P2E_OVER      1                  // k_pack2_encode: a read's last, partial chunk by one 16-byte load when text follows it in the buffer
#endif
#ifndef P2D_ENDS
#define P2D_ENDS      1                  // a read's last chunks through the 16-letters-per-lane code as well (0: byte by byte, divisions and all)
#endif
#ifndef P2D_SKIP
#define P2D_SKIP      0                  // timing experiments: 1 the pipelined steps store nothing, 2 no per-byte path (wrong results on purpose).
                                         // 10 M x 10 kb (profiles/r03d_perturb_pack2_decode.txt): 32.3 ms; 1: 15.2; 2: 29.0; 3: 9.5 -- the parts ADD UP: the
                                         // loads and the letters of one wave do not run beside the stores of another.  Loads issued three steps ahead
                                         // (three stores in flight per wave, vmcnt(5)): 31.7-32.4 against 32.1-33.6, and 73 registers -- not kept.
#endif
#if P2D_PIPE >= 2 && (P2D_SKIP & 1)
#error "the pipelined loops count their memory operations by hand (one load, one store a step): no P2D_SKIP & 1 with P2D_PIPE >= 2"
#endif
#ifndef P2D_ALIGN
#define P2D_ALIGN     1                  // k_pack2_decode: a read's text as if it began (its address mod 16) bytes earlier, so that every 16-byte store of the
#endif                                   // wave stands on a 16-byte boundary (stores 4 bytes off one: 3.5 TB/s against 5.3, profiles/r05e_copy_rate.txt: what the kernel ran at)
#ifndef P2D_NT
#define P2D_NT        0                  // k_pack2_decode: non-temporal stores for the text
#endif

// ---------------------------------------------------------------------------------------------
//  alphabet maps (computed, not tabulated)
// ---------------------------------------------------------------------------------------------
template <int ALPHA>
__device__ __forceinline__ uint32_t sym_code(uint32_t x)
{ if (ALPHA == DX_ALPHA_BASES)
    { uint32_t u = x & 0xdfu;                        // fold case; bytes >= 128 keep bit 7 and miss
      return (u == 'C') ? 1u : (u == 'G') ? 2u : (u == 'T') ? 3u : 0u;
    }
  else if (ALPHA == DX_ALPHA_ARROW)
    return (x == '1') ? 0u : (x == '2') ? 1u : (x == '3' || x == 'G') ? 2u : 3u;
  else
    return x & 3u;                                   // DX_ALPHA_NUMBERS
}

#define BYTE_AT(c, b) ((chunk_word(c, (b) >> 2) >> (8 * ((b) & 3))) & 0xffu)

// this lane's 16 bytes of a read's text at p + pos: bytes past the text's end read as 0.  over: 16 bytes may be read
// at any position of this read (what follows it in the buffer is more text) -- the read's last, partial chunk is then one
// load and four masks instead of up to 15 byte loads by one lane, which the whole wave waits for
__device__ __forceinline__ u32x4 p2_fetch(const uint8_t *p, uint32_t pos, uint32_t T, bool over)
{ if (!over || pos >= T || T - pos >= 16u)
    return load_chunk(p + pos, pos >= T ? 0 : (T - pos >= 16u ? 16 : (int) (T - pos)));
  const uint32_t left = T - pos;
  u32x4 v = *(const u32x4_u *) (p + pos);
  const uint32_t m0 = left >= 4u  ? ~0u : ~(~0u << (8u * left));
  const uint32_t m1 = left >= 8u  ? ~0u : (left > 4u  ? ~(~0u << (8u * (left - 4u)))  : 0u);
  const uint32_t m2 = left >= 12u ? ~0u : (left > 8u  ? ~(~0u << (8u * (left - 8u)))  : 0u);
  const uint32_t m3 =                     (left > 12u ? ~(~0u << (8u * (left - 12u))) : 0u);
  v.x &= m0; v.y &= m1; v.z &= m2; v.w &= m3;
  return v;
}

template <int ALPHA>
__global__ __launch_bounds__(DX_BLOCK)
void k_pack2_encode(const uint8_t *__restrict__ text, const uint64_t *__restrict__ off,
                    const uint32_t *__restrict__ tlen, const uint32_t *__restrict__ nsym, uint64_t n,
                    const uint8_t *__restrict__ hdr, const uint64_t *__restrict__ hdr_off,
                    uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
                    uint32_t *__restrict__ status, uint32_t *__restrict__ ticket)
{ __shared__ __attribute__((aligned(16))) uint32_t s_win[DX_WAVES_PER_BLK][P2_WIN_WORDS];
  __shared__ uint8_t  s_code[256];       // Number_Read / Number_Arrow as a table: 256 B = one LDS
                                         // bank per dword, so the 64 look-ups of a wave never conflict
  const int       lane  = lane_id();
  const int       wid   = threadIdx.x >> 6;

  for (int k = threadIdx.x; k < 256; k += DX_BLOCK)
    s_code[k] = (uint8_t) sym_code<ALPHA>((uint32_t) k);
  wave_out o;
  o.win = s_win[wid];
  for (int j = lane; j < P2_WIN_WORDS; j += 64)
    o.win[j] = 0;
  __syncthreads();

  const uint64_t text_end = off[n - 1] + tlen[n - 1];    // (the reads lie one behind the other: the last one's end is the buffer's)
  const uint32_t TB = P2_TB(ticket);                     // (k_ticket_units in front of the launch: 16 reads of 10 kb, more of shorter ones)
  for (uint64_t r0 = P2_FIRST(ticket, TB), nxt; r0 < n; r0 = nxt)
  { nxt = P2_NEXT(ticket, TB);                           // drawn early: hidden behind these reads
    for (uint64_t r = r0; r < r0 + TB && r < n; r++)
    { const uint8_t *src = text + off[r];
      const uint32_t T   = tlen[r];
      const bool     over = P2E_OVER && off[r] + T + 16u <= text_end;
      uint8_t       *dst = out + out_off[r];

      if (hdr != NULL)                                   // record framing bytes (dexta.c:187-198)
        { const uint64_t h0 = hdr_off[r];
          const uint32_t hl = (uint32_t) (hdr_off[r + 1] - h0);
          for (uint32_t k = lane; k < hl; k += 64)
            dst[k] = hdr[h0 + k];
          dst += hl;
        }
      o.seg = dst; o.wordbase = 0; o.winbits = 0;

      // 32 consecutive bytes per lane and step (two chunks): the prefix sum, the window bookkeeping
      // and the drain are paid once per 2 KiB
      uint32_t pos = 32u * lane;
      u32x4 cA = p2_fetch(src, pos, T, over), cB = p2_fetch(src, pos + 16u, T, over);
      for (uint32_t base = 0; base < T; base += P2_STEP)
        { const u32x4 dA = p2_fetch(src, pos + P2_STEP, T, over), dB = p2_fetch(src, pos + P2_STEP + 16u, T, over);   // next step in flight
          const uint32_t left   = pos >= T ? 0u : T - pos;
          const uint32_t validA = left >= 16u ? 16u : left, validB = left >= 32u ? 16u : (left > 16u ? left - 16u : 0u);
          // the bytes' codes, first one in the top bits, as if there were no line ends
          uint32_t accA = 0, accB = 0;
          #pragma unroll
          for (int b = 0; b < 16; b++)
            { accA = (accA << 2) | (uint32_t) s_code[BYTE_AT(cA, b)];
              accB = (accB << 2) | (uint32_t) s_code[BYTE_AT(cB, b)];
            }
          // drop the slots of the line ends, last one first so that the earlier slots stay put
          uint32_t nlA = chunk_eq_mask(cA, '\n'), nlB = chunk_eq_mask(cB, '\n');   // missing bytes read as 0: never set
          const uint32_t cntA = validA - __popc(nlA), cntB = validB - __popc(nlB), cnt = cntA + cntB;
          while (nlA | nlB)
            { if (nlA)
                { const uint32_t p = 31u - (uint32_t) __clz(nlA);  // byte position; its slot: bits 31-2p, 30-2p
                  const uint32_t K = 30u - 2u * p;
                  const uint32_t below = (1u << K) - 1u, upto = (4u << K) - 1u;
                  accA = (accA & ~upto) | ((accA & below) << 2);
                  nlA ^= 1u << p;
                }
              if (nlB)
                { const uint32_t p = 31u - (uint32_t) __clz(nlB);
                  const uint32_t K = 30u - 2u * p;
                  const uint32_t below = (1u << K) - 1u, upto = (4u << K) - 1u;
                  accB = (accB & ~upto) | ((accB & below) << 2);
                  nlB ^= 1u << p;
                }
            }
          accA = cntA ? accA & (~0u << (32u - 2u * cntA)) : 0u;    // slots past the kept symbols (missing bytes)
          accB = cntB ? accB & (~0u << (32u - 2u * cntB)) : 0u;
          // the lane's string, left-aligned in 64 bits: A's codes, then B's
          const uint64_t V  = ((uint64_t) accA << 32) | ((uint64_t) accB << (32u - 2u * cntA));
          const uint32_t hi = (uint32_t) (V >> 32), lo = (uint32_t) V;
          const uint32_t incl = wave_incl_scan(cnt);
          if (cnt)
            { const uint32_t bit = o.winbits + 2u * (incl - cnt);
              const uint32_t w = bit >> 5, sh = bit & 31u;
              const uint32_t x0 = hi >> sh, x1 = __builtin_amdgcn_alignbit(hi, lo, sh), x2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
              atomicOr(&o.win[w], x0);
              if (x1) atomicOr(&o.win[w + 1], x1);
              if (x2) atomicOr(&o.win[w + 2], x2);
            }
          o.winbits += 2u * wave_total(incl);
          if (o.winbits >= P2_FLUSH_BITS)                // (drained at the START of the next step, behind its requests -- what pays
            flush_quads(o, true);                        //  in k_qv_encode_fast -- this kernel is slower: 29.5 ms against 28.3)
          cA = dA; cB = dB;
          pos += P2_STEP;
        }

      flush_words(o, true);
      const uint32_t G    = 16u * o.wordbase + (o.winbits >> 1);  // symbols seen
      const uint32_t clen = (G + 3u) >> 2;                        // COMPRESSED_LEN, DB.h:255
      if (lane == 0)
        { const uint32_t w = __builtin_bswap32(o.win[0]);
          for (uint32_t k = 4u * o.wordbase; k < clen; k++)
            dst[k] = (uint8_t) (w >> (8 * (k - 4u * o.wordbase)));
          o.win[0] = 0;
          if (G != nsym[r])
            atomicOr(status, 1u);
        }
      wave_sync();
    }
  }
}

// ---------------------------------------------------------------------------------------------
//  decode: output driven, 16 text bytes per lane
// ---------------------------------------------------------------------------------------------
template <int LETTERS>
__device__ __forceinline__ uint32_t sym_letter(uint32_t code)
{ if (LETTERS == DX_LETTERS_LOWER) return (0x74676361u >> (8 * code)) & 0xffu;       // "acgt"
  if (LETTERS == DX_LETTERS_UPPER) return (0x54474341u >> (8 * code)) & 0xffu;       // "ACGT"
  if (LETTERS == DX_LETTERS_ARROW) return '1' + code;
  return code;                                       // DX_LETTERS_NUMBERS
}

// 16 text bytes of the generic case: any width, chunks at the end of the text
template <int LETTERS>
__device__ __forceinline__ void decode_chunk_generic(const uint8_t *src, uint8_t *dst, uint32_t q0, uint32_t T,
                                                     uint32_t clen, uint32_t width, uint32_t limit = 16u)
{ const uint32_t W1    = width + 1u;
  const int      valid = (T - q0 >= limit) ? (int) limit : (int) (T - q0);
  const uint32_t line  = q0 / W1;
  uint32_t       col   = q0 - line * W1;
  const uint32_t i0    = q0 - line;                              // symbol index of text byte q0 (if a letter)
  const uint32_t b0    = i0 >> 2;                                // first packed byte needed
  uint64_t bits = 0;                                             // packed bytes b0.., first in the low byte
  if (b0 + 8u <= clen)
    bits = *(const u64_u *) (src + b0);
  else
    for (uint32_t k = b0; k < clen; k++)
      bits |= (uint64_t) src[k] << (8 * (k - b0));

  uint32_t w[4] = { 0u, 0u, 0u, 0u };
  uint32_t idx  = i0;
  #pragma unroll
  for (int b = 0; b < 16; b++)
    { uint32_t ch;
      if (col == width || q0 + b == T - 1u)
        { ch = '\n'; col = 0; }
      else
        { const uint32_t rel = idx - 4u * b0;                    // 0 .. 18
          const uint32_t code = (uint32_t) (bits >> (8u * (rel >> 2) + 6u - 2u * (rel & 3u))) & 3u;
          ch = sym_letter<LETTERS>(code);
          idx += 1; col += 1;
        }
      w[b >> 2] |= ch << (8 * (b & 3));
    }
  if (valid == 16)
    { u32x4 v = { w[0], w[1], w[2], w[3] };
      *(u32x4_u *) (dst + q0) = v;
    }
  else
    for (int b = 0; b < valid; b++)
      dst[q0 + b] = (uint8_t) (w[b >> 2] >> (8 * (b & 3)));
}

template <int LETTERS>
__global__ __launch_bounds__(DX_BLOCK)
void k_pack2_decode(const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
                    const uint32_t *__restrict__ nsym, uint64_t n, uint32_t width,
                    uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off, uint32_t *__restrict__ ticket)
{ __shared__ uint32_t s_quad[256];       // packed byte -> its four letters (Lower_Read & co., DB.c:367-389)
  const int      lane  = lane_id();
  for (uint32_t k = threadIdx.x; k < 256u; k += DX_BLOCK)
    s_quad[k] = sym_letter<LETTERS>(k >> 6) | (sym_letter<LETTERS>((k >> 4) & 3u) << 8)
              | (sym_letter<LETTERS>((k >> 2) & 3u) << 16) | (sym_letter<LETTERS>(k & 3u) << 24);
  __syncthreads();

  // text position 16*lane + 1024*step as (line, column): advanced incrementally, no division per step
  const uint32_t W1     = width + 1u;
  const uint32_t line0  = (16u * (uint32_t) lane) / W1, col0 = 16u * (uint32_t) lane - line0 * W1;
  const uint32_t dline  = DX_STEP / W1, dcol = DX_STEP - dline * W1;
  const bool     narrow = width < 16u;   // several line ends may fall into 16 bytes: generic path only

  const uint32_t TB = P2_TB(ticket);
  for (uint64_t r0 = P2_FIRST(ticket, TB), nxt; r0 < n; r0 = nxt)
  { nxt = P2_NEXT(ticket, TB);
    for (uint64_t r = r0; r < r0 + TB && r < n; r++)
    { const uint8_t *src  = in + in_off[r];
      uint8_t       *dst  = out + out_off[r];
      const uint32_t L    = nsym[r];
      const uint32_t clen = (L + 3u) >> 2;
      const uint32_t T    = L + (L + width - 1u) / width;        // letters + newlines
      // The text as if it began adj bytes in front of dst (dstS, a 16-byte boundary): chunk q of that frame is text byte q - adj,
      // every lane's 16-byte store is aligned, and only the frame's first chunk (lane 0 of the first step: the text's first
      // 16 - adj bytes) is not whole.  (line, col) are the text position's: 16 lane - adj, one borrow at most (width >= 16).
      const uint32_t adj  = P2D_ALIGN && !narrow ? (uint32_t) ((uintptr_t) dst & 15u) : 0u;
      uint8_t       *dstS = dst - adj;
      const uint32_t TS   = T + adj;
      (void) dstS;
      uint32_t line = line0, col = col0 - adj;
      if (col0 < adj) { col += W1; line -= 1u; }                 // (lane 0: position -adj, as line -1: the advances below stay consistent)

      // Main loop: the steps in which every lane takes the fast path (16 letters from one 8-byte load; all but a read's
      // last one or two), software-pipelined by hand.  A wave's loads and stores are counted together (vmcnt) and retire
      // in issue order, and left to itself the compiler follows each 8-byte load with a wait for EVERYTHING outstanding:
      // the load's whole latency and the acknowledgement of the step before's store, every step.  Here the issue order
      // per step is [wait vmcnt(1): this step's load is back, the last step's store may still be on its way] [request
      // the next step's 8 bytes] [16 letters] [store]; two steps per iteration with the two loads in registers of their
      // own (a copy of a requested register would wait for it).
      uint32_t base = 0;
#if P2D_PIPE
#define P2D_ALLFAST(B) (!narrow && (B) + DX_STEP < TS && (((B) + DX_STEP - 16u) >> 2) + 8u <= clen)
#define P2D_ADVANCE(LN, CL) { LN += dline; CL += dcol; if (CL >= W1) { CL -= W1; LN += 1u; } }
#define P2D_PRE(B) ((B) + 16u * (uint32_t) lane < adj)        /* the frame's first chunk (lane 0 of the first step): the text's first 16 bytes instead, */ \
                                                              /* stored where they stand -- the one store of the read off a boundary; lane 1's writes the bytes they share again */
#define P2D_LOAD(B, LN) (*(const u64_u *) (src + ((P2D_PRE(B) ? 0u : (B) + 16u * (uint32_t) lane - adj - (LN)) >> 2)))
#define P2D_STEP_FAST(RAW, B, LN, CL)                                                                        \
          { const bool     pre = P2D_PRE(B);                                                                 \
            const uint32_t q0 = pre ? 0u : (B) + 16u * (uint32_t) lane - adj, i0 = pre ? 0u : q0 - (LN), nlpos = pre ? width : width - (CL); \
            const uint64_t be = ((uint64_t) __builtin_bswap32((uint32_t) (RAW)) << 32) | __builtin_bswap32((uint32_t) ((RAW) >> 32)); \
            uint32_t cw = (uint32_t) ((be << (2u * (i0 & 3u))) >> 32);                                       \
            if (nlpos < 16u)                                                                                 \
              { const uint32_t K = 30u - 2u * nlpos;           /* open a 2-bit hole at slot nlpos */         \
                const uint32_t keep = ~((4u << K) - 1u);       /* slots before it */                         \
                cw = (cw & keep) | ((cw & ~keep) >> 2);                                                      \
              }                                                                                              \
            u32x4 v;                                                                                         \
            v.x = s_quad[cw >> 24];                                                                          \
            v.y = s_quad[(cw >> 16) & 0xffu];                                                                \
            v.z = s_quad[(cw >> 8) & 0xffu];                                                                 \
            v.w = s_quad[cw & 0xffu];                                                                        \
            if (nlpos < 16u)                                                                                 \
              { const uint32_t m = 0xffu << (8u * (nlpos & 3u)), j = nlpos >> 2;                             \
                const uint32_t nl4 = 0x0a0a0a0au;                                                            \
                v.x = j == 0u ? (v.x & ~m) | (nl4 & m) : v.x;                                                \
                v.y = j == 1u ? (v.y & ~m) | (nl4 & m) : v.y;                                                \
                v.z = j == 2u ? (v.z & ~m) | (nl4 & m) : v.z;                                                \
                v.w = j == 3u ? (v.w & ~m) | (nl4 & m) : v.w;                                                \
              }                                                                                              \
            if (!(P2D_SKIP & 1)) *(u32x4_u *) (dst + q0) = v;                                                \
            else if (v.x == 0x12345678u) dst[q0] = 1;                                                        \
          }
#if P2D_PIPE == 3
      // Three steps ahead (see the two-steps-ahead loop below for the scheme): what a wave may have in flight is what bounds it -- a wait
      // for a load is a wait for every older store as well (one counter, in order), and the stores of a kernel that writes 100 GB take
      // 4 us to be acknowledged: d steps ahead is d stores in flight.  A turn waits for all but the 2 d youngest operations.
      if (P2D_ALLFAST(0u) && P2D_ALLFAST(DX_STEP) && P2D_ALLFAST(2u * DX_STEP))
        { uint32_t lineA = line, colA = col, lineB = line, colB = col, lineC, colC, lineD, colD;
          P2D_ADVANCE(lineB, colB)
          lineC = lineB; colC = colB;
          P2D_ADVANCE(lineC, colC)
#define P2D_ASK(R, B, LN) { const uint8_t *a_ = src + ((P2D_PRE(B) ? 0u : (B) + 16u * (uint32_t) lane - adj - (LN)) >> 2); \
                            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(R) : "v"(a_) : "memory"); }
#define P2D_ARRIVED(R)    asm volatile("" : "+v"(R) : : "memory");
          uint64_t rawA, rawB, rawC, rawD = 0;
          P2D_ASK(rawA, 0u, lineA)
          P2D_ASK(rawB, DX_STEP, lineB)
          P2D_ASK(rawC, 2u * DX_STEP, lineC)
          int      left = 0;
          // (X: the step at base; Y: the newest requested, base + 2 steps; Z: takes the request for base + 3 steps)
#define P2D_TURN(X, Y, Z, WAIT, OUT)                                                                              \
          { const bool ok = P2D_ALLFAST(base + 3u * DX_STEP);        /* wave-uniform */                          \
            line##Z = line##Y; col##Z = col##Y;                                                                 \
            P2D_ADVANCE(line##Z, col##Z)                                                                        \
            if (!ok) { left = OUT; break; }                                                                     \
            P2D_ASK(raw##Z, base + 3u * DX_STEP, line##Z)                                                       \
            __builtin_amdgcn_s_waitcnt(WAIT);                                                                   \
            P2D_ARRIVED(raw##X)                                                                                 \
            P2D_STEP_FAST(raw##X, base, line##X, col##X)                                                        \
            base += DX_STEP;                                                                                    \
          }
          do
            { P2D_TURN(A, C, D, 0x0F73, 0)                   // vmcnt(3): younger than A's load are B's, C's and D's
              P2D_TURN(B, D, A, 0x0F74, 1)                   // vmcnt(4): ... and A's store
              P2D_TURN(C, A, B, 0x0F75, 2)
              for (;;)
                { P2D_TURN(D, B, C, 0x0F76, 3)               // vmcnt(6): three loads and three stores
                  P2D_TURN(A, C, D, 0x0F76, 0)
                  P2D_TURN(B, D, A, 0x0F76, 1)
                  P2D_TURN(C, A, B, 0x0F76, 2)
                }
            }
          while (0);
#undef P2D_TURN
#undef P2D_ASK
          // three steps are in flight or back: the one at base and the two behind it
          __builtin_amdgcn_s_waitcnt(0x0F70);
          P2D_ARRIVED(rawA) P2D_ARRIVED(rawB) P2D_ARRIVED(rawC) P2D_ARRIVED(rawD)
#define P2D_DRAIN(X, Y, Z, W) { P2D_STEP_FAST(raw##X, base, line##X, col##X) base += DX_STEP; P2D_STEP_FAST(raw##Y, base, line##Y, col##Y) base += DX_STEP; \
                                P2D_STEP_FAST(raw##Z, base, line##Z, col##Z) base += DX_STEP; line = line##W; col = col##W; }
          if (left == 0)      P2D_DRAIN(A, B, C, D)
          else if (left == 1) P2D_DRAIN(B, C, D, A)
          else if (left == 2) P2D_DRAIN(C, D, A, B)
          else                P2D_DRAIN(D, A, B, C)
#undef P2D_DRAIN
#undef P2D_ARRIVED
        }
#elif P2D_PIPE == 2
      // Two steps ahead: a wave's step is as long as a request's way to memory and back under this kernel's load (2 us), so with the
      // next step's request alone in flight every step ends in a wait.  Three registers in turn; a turn: [request step k + 2] [wait
      // until all but the four youngest memory operations are back -- the store of step k - 2, the load of k + 1, the store of k - 1,
      // the load of k + 2: the load of step k is the fifth] [16 letters] [store].  The first two turns have fewer stores in front
      // (vmcnt 2, 3); when the steps that all lanes take the fast way run out, what is in flight is finished with plain waits.
      if (P2D_ALLFAST(0u) && P2D_ALLFAST(DX_STEP))
        { uint32_t lineA = line, colA = col, lineB = line, colB = col, lineC, colC;
          P2D_ADVANCE(lineB, colB)
          // (The requests are written in assembly: left to count them, the compiler loses the order where the loop's paths meet and
          //  puts a wait for all but two in front of every wait written here -- one step ahead again.  It cannot see these loads, so it
          //  waits for nothing; P2D_ARRIVED ties the register to the wait in front of it, so that nothing that reads it moves above.)
#define P2D_ASK(R, B, LN) { const uint8_t *a_ = src + ((P2D_PRE(B) ? 0u : (B) + 16u * (uint32_t) lane - adj - (LN)) >> 2); \
                            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(R) : "v"(a_) : "memory"); }
#define P2D_ARRIVED(R)    asm volatile("" : "+v"(R) : : "memory");
          uint64_t rawA, rawB, rawC = 0;
          P2D_ASK(rawA, 0u, lineA)
          P2D_ASK(rawB, DX_STEP, lineB)
          int      left = 0;                                 // how the loop was left: which register holds the step at `base`
          // (X: the step at base; Y: the one behind it; Z: takes the request for base + 2 steps)
#define P2D_TURN(X, Y, Z, WAIT, OUT)                                                                              \
          { const bool ok = P2D_ALLFAST(base + 2u * DX_STEP);        /* wave-uniform */                          \
            line##Z = line##Y; col##Z = col##Y;                                                                 \
            P2D_ADVANCE(line##Z, col##Z)                                                                        \
            if (!ok) { left = OUT; break; }                                                                     \
            P2D_ASK(raw##Z, base + 2u * DX_STEP, line##Z)                                                       \
            __builtin_amdgcn_s_waitcnt(WAIT);                                                                   \
            P2D_ARRIVED(raw##X)                                                                                 \
            P2D_STEP_FAST(raw##X, base, line##X, col##X)                                                        \
            base += DX_STEP;                                                                                    \
          }
          do
            { P2D_TURN(A, B, C, 0x0F72, 0)                   // vmcnt(2): younger than A's load are B's and C's
              P2D_TURN(B, C, A, 0x0F73, 1)                   // vmcnt(3): C's load, A's store, A's new load
              for (;;)
                { P2D_TURN(C, A, B, 0x0F74, 2)
                  P2D_TURN(A, B, C, 0x0F74, 0)
                  P2D_TURN(B, C, A, 0x0F74, 1)
                }
            }
          while (0);
#undef P2D_TURN
#undef P2D_ASK
          // two steps are in flight or back: the one at base and the one behind it
          __builtin_amdgcn_s_waitcnt(0x0F70);
          P2D_ARRIVED(rawA) P2D_ARRIVED(rawB) P2D_ARRIVED(rawC)
          if (left == 0)      { P2D_STEP_FAST(rawA, base, lineA, colA) base += DX_STEP; P2D_STEP_FAST(rawB, base, lineB, colB) base += DX_STEP; line = lineC; col = colC; }
          else if (left == 1) { P2D_STEP_FAST(rawB, base, lineB, colB) base += DX_STEP; P2D_STEP_FAST(rawC, base, lineC, colC) base += DX_STEP; line = lineA; col = colA; }
          else                { P2D_STEP_FAST(rawC, base, lineC, colC) base += DX_STEP; P2D_STEP_FAST(rawA, base, lineA, colA) base += DX_STEP; line = lineB; col = colB; }
#undef P2D_ARRIVED
        }
#else
      if (P2D_ALLFAST(0u))
        { uint32_t lineA = line, colA = col, lineB = line, colB = col;
          uint64_t rawA = P2D_LOAD(0u, lineA), rawB = 0;
          __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the first step has no store in front of it
          for (;;)
            { // (the waits stand at the head of each half, on every path into it: a wait on one branch only does not
              // count for the compiler where the branches meet, and it adds one of its own, for everything)
              __builtin_amdgcn_s_waitcnt(0x0F71);          // vmcnt(1): rawA is back (requested before the last store)
              bool ok = P2D_ALLFAST(base + DX_STEP);       // wave-uniform
              lineB = lineA; colB = colA;
              P2D_ADVANCE(lineB, colB)
              if (ok) rawB = P2D_LOAD(base + DX_STEP, lineB);
              P2D_STEP_FAST(rawA, base, lineA, colA)
              base += DX_STEP;
              if (!ok) { line = lineB; col = colB; break; }
              __builtin_amdgcn_s_waitcnt(0x0F71);          // vmcnt(1): rawB is back
              ok = P2D_ALLFAST(base + DX_STEP);
              lineA = lineB; colA = colB;
              P2D_ADVANCE(lineA, colA)
              if (ok) rawA = P2D_LOAD(base + DX_STEP, lineA);
              P2D_STEP_FAST(rawB, base, lineB, colB)
              base += DX_STEP;
              if (!ok) { line = lineA; col = colA; break; }
            }
        }
#endif
#undef P2D_STEP_FAST
#undef P2D_LOAD
#undef P2D_ADVANCE
#undef P2D_ALLFAST
#endif
      // the rest (and everything, for narrow lines): P2D_UNROLL steps of 1 KiB per iteration, checked lane by lane
      for (; base < TS; base += P2D_UNROLL * DX_STEP)
        { uint64_t raw[P2D_UNROLL];
          uint32_t i0k[P2D_UNROLL], nlk[P2D_UNROLL];
          bool     fastk[P2D_UNROLL];
          #pragma unroll
          for (int k = 0; k < (int) P2D_UNROLL; k++)
            { const uint32_t qs = base + (uint32_t) k * DX_STEP + 16u * lane;
              const bool     pre = qs < adj;                     // the frame's first chunk (lane 0 of the first step): the text's first 16 bytes instead
              const uint32_t q0 = pre ? 0u : qs - adj;           // the text byte the chunk begins with
              const uint32_t i0 = pre ? 0u : q0 - line;          // symbols before text byte q0
              const uint32_t b0 = i0 >> 2;
              i0k[k]   = i0;
              nlk[k]   = pre ? width : width - col;              // offset of the line end in this chunk (>= 16: none)
              // every chunk of the text but those of narrow lines and of reads shorter than 8 packed bytes: the read's last
              // chunks too (P2D_ENDS) -- their 8 bytes taken from the last 8 of the read and shifted when they would reach
              // past its end, a second hole for the text's last line end, a byte-wise store when the chunk is not whole
              fastk[k] = P2D_ENDS ? (q0 < T && clen >= 8u && !narrow) : (q0 + 16u < T && b0 + 8u <= clen && !narrow);
              raw[k]   = 0;
              if (fastk[k])
                { if (b0 + 8u <= clen)
                    raw[k] = *(const u64_u *) (src + b0);
                  else
                    { const uint32_t sh = 8u * (b0 + 8u - clen);
                      const uint64_t ld = *(const u64_u *) (src + clen - 8u);
                      raw[k] = sh < 64u ? ld >> sh : 0ull;
                    }
                }
              line += dline; col += dcol;
              if (col >= W1) { col -= W1; line += 1u; }
            }
          #pragma unroll
          for (int k = 0; k < (int) P2D_UNROLL; k++)
            { const uint32_t qs = base + (uint32_t) k * DX_STEP + 16u * lane;
              const uint32_t q0 = qs < adj ? 0u : qs - adj;
              if (fastk[k])
                { // 16 (or 15 + line end) letters from 8 packed bytes: symbol j of the chunk in bits 31-2j, 30-2j
                  const uint64_t be = ((uint64_t) __builtin_bswap32((uint32_t) raw[k]) << 32) | __builtin_bswap32((uint32_t) (raw[k] >> 32));
                  uint32_t cw = (uint32_t) ((be << (2u * (i0k[k] & 3u))) >> 32);
                  const uint32_t nlpos = nlk[k];
                  const uint32_t fpos  = P2D_ENDS ? T - 1u - q0 : 16u;   // the text's last byte (a line end) in this chunk (>= 16: not here)
                  if (nlpos < 16u)
                    { const uint32_t K = 30u - 2u * nlpos;       // open a 2-bit hole at slot nlpos
                      const uint32_t keep = ~((4u << K) - 1u);   // slots before it
                      cw = (cw & keep) | ((cw & ~keep) >> 2);
                    }
                  if (fpos < 16u && fpos != nlpos)               // (a short last line: its end is not where the width puts one)
                    { const uint32_t K = 30u - 2u * fpos;
                      const uint32_t keep = ~((4u << K) - 1u);
                      cw = (cw & keep) | ((cw & ~keep) >> 2);
                    }
                  u32x4 v;
                  v.x = s_quad[cw >> 24];
                  v.y = s_quad[(cw >> 16) & 0xffu];
                  v.z = s_quad[(cw >> 8) & 0xffu];
                  v.w = s_quad[cw & 0xffu];
                  if (nlpos < 16u)
                    { const uint32_t m = 0xffu << (8u * (nlpos & 3u)), j = nlpos >> 2;
                      const uint32_t nl4 = 0x0a0a0a0au;
                      v.x = j == 0u ? (v.x & ~m) | (nl4 & m) : v.x;
                      v.y = j == 1u ? (v.y & ~m) | (nl4 & m) : v.y;
                      v.z = j == 2u ? (v.z & ~m) | (nl4 & m) : v.z;
                      v.w = j == 3u ? (v.w & ~m) | (nl4 & m) : v.w;
                    }
                  if (fpos < 16u && fpos != nlpos)
                    { const uint32_t m = 0xffu << (8u * (fpos & 3u)), j = fpos >> 2;
                      const uint32_t nl4 = 0x0a0a0a0au;
                      v.x = j == 0u ? (v.x & ~m) | (nl4 & m) : v.x;
                      v.y = j == 1u ? (v.y & ~m) | (nl4 & m) : v.y;
                      v.z = j == 2u ? (v.z & ~m) | (nl4 & m) : v.z;
                      v.w = j == 3u ? (v.w & ~m) | (nl4 & m) : v.w;
                    }
                  if (fpos < 15u)                                // the text ends inside this chunk: its bytes one by one
                    { const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
                      for (uint32_t b = 0; b <= fpos; b++)
                        dst[q0 + b] = (uint8_t) ((b < 4u ? w4[0] : b < 8u ? w4[1] : b < 12u ? w4[2] : w4[3]) >> (8u * (b & 3u)));
                    }
                  else
#if P2D_NT
                  __builtin_nontemporal_store(v, (u32x4_u *) (dst + q0));
#else
                  *(u32x4_u *) (dst + q0) = v;
#endif
                }
              else if (q0 < T && !(P2D_SKIP & 2))
                decode_chunk_generic<LETTERS>(src, dst, q0, T, clen, width);
            }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
//  C-ABI
// ---------------------------------------------------------------------------------------------
extern "C" int dx_pack2_encode(dx_ctx *ctx, int alphabet,
                               const uint8_t *d_text, const uint64_t *d_off, const uint32_t *d_tlen,
                               const uint32_t *d_nsym, uint64_t n,
                               const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                               uint8_t *d_out, const uint64_t *d_out_off)
{ if (ctx == NULL) return DX_E_ARG;
  if (alphabet != DX_ALPHA_BASES && alphabet != DX_ALPHA_ARROW && alphabet != DX_ALPHA_NUMBERS)
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_encode: unknown alphabet %d", alphabet);
  if ((d_hdr == NULL) != (d_hdr_off == NULL))
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_encode: d_hdr and d_hdr_off must be given together");
  if (n == 0) return DX_OK;
  if (n >= (1ull << 31))
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_encode: more than 2^31 - 1 reads in one batch");
  if (!d_text || !d_off || !d_tlen || !d_nsym || !d_out || !d_out_off)
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_encode: NULL device pointer");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  DX_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
  uint32_t *d_ticket = (uint32_t *) (ctx->d_u64 + 20);
  DX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_ticket_units, dim3(1), dim3(1), 0, ctx->stream, d_off, d_off + (n - 1), d_tlen + (n - 1), n,
                     P2_BATCH * 10000u, P2_BATCH, d_ticket);
  const int grid = P2_FLAT ? (int) ((n + (uint64_t) P2_FLAT * DX_WAVES_PER_BLK - 1) / ((uint64_t) P2_FLAT * DX_WAVES_PER_BLK)) : dx_grid_waves(ctx, n, 32);
  if (alphabet == DX_ALPHA_BASES)
    DX_LAUNCH(ctx, DX_K_PACK2_ENC, k_pack2_encode<DX_ALPHA_BASES>, grid, DX_BLOCK,
              d_text, d_off, d_tlen, d_nsym, n, d_hdr, d_hdr_off, d_out, d_out_off, ctx->d_status, d_ticket);
  else if (alphabet == DX_ALPHA_ARROW)
    DX_LAUNCH(ctx, DX_K_PACK2_ENC, k_pack2_encode<DX_ALPHA_ARROW>, grid, DX_BLOCK,
              d_text, d_off, d_tlen, d_nsym, n, d_hdr, d_hdr_off, d_out, d_out_off, ctx->d_status, d_ticket);
  else
    DX_LAUNCH(ctx, DX_K_PACK2_ENC, k_pack2_encode<DX_ALPHA_NUMBERS>, grid, DX_BLOCK,
              d_text, d_off, d_tlen, d_nsym, n, d_hdr, d_hdr_off, d_out, d_out_off, ctx->d_status, d_ticket);
  uint32_t st = 0;
  DX_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st & 1u)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_pack2_encode: a read's symbol count differs from d_nsym");
  return DX_OK;
}

extern "C" int dx_pack2_decode(dx_ctx *ctx, int letters,
                               const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_nsym,
                               uint64_t n, uint32_t width, uint8_t *d_out, const uint64_t *d_out_off)
{ if (ctx == NULL) return DX_E_ARG;
  if (letters < DX_LETTERS_LOWER || letters > DX_LETTERS_NUMBERS)
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_decode: unknown letter set %d", letters);
  if (width == 0)
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_decode: line width must be >= 1 "
                                  "(the reference loops forever on -w0, undexta.c:265)");
  if (n == 0) return DX_OK;
  if (n >= (1ull << 31))
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_decode: more than 2^31 - 1 reads in one batch");
  if (!d_in || !d_in_off || !d_nsym || !d_out || !d_out_off)
    return dx_fail(ctx, DX_E_ARG, "dx_pack2_decode: NULL device pointer");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t *d_ticket = (uint32_t *) (ctx->d_u64 + 21);
  DX_HIP(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_ticket_units, dim3(1), dim3(1), 0, ctx->stream, d_out_off, d_out_off + (n - 1), d_nsym + (n - 1), n,
                     P2_BATCH * 10000u, P2_BATCH, d_ticket);
  const int grid = P2_FLAT ? (int) ((n + (uint64_t) P2_FLAT * DX_WAVES_PER_BLK - 1) / ((uint64_t) P2_FLAT * DX_WAVES_PER_BLK)) : dx_grid_waves(ctx, n, 32);
  switch (letters)
    { case DX_LETTERS_LOWER:
        DX_LAUNCH(ctx, DX_K_PACK2_DEC, k_pack2_decode<DX_LETTERS_LOWER>, grid, DX_BLOCK,
                  d_in, d_in_off, d_nsym, n, width, d_out, d_out_off, d_ticket);
        break;
      case DX_LETTERS_UPPER:
        DX_LAUNCH(ctx, DX_K_PACK2_DEC, k_pack2_decode<DX_LETTERS_UPPER>, grid, DX_BLOCK,
                  d_in, d_in_off, d_nsym, n, width, d_out, d_out_off, d_ticket);
        break;
      case DX_LETTERS_ARROW:
        DX_LAUNCH(ctx, DX_K_PACK2_DEC, k_pack2_decode<DX_LETTERS_ARROW>, grid, DX_BLOCK,
                  d_in, d_in_off, d_nsym, n, width, d_out, d_out_off, d_ticket);
        break;
      default:
        DX_LAUNCH(ctx, DX_K_PACK2_DEC, k_pack2_decode<DX_LETTERS_NUMBERS>, grid, DX_BLOCK,
                  d_in, d_in_off, d_nsym, n, width, d_out, d_out_off, d_ticket);
        break;
    }
  return DX_OK;
}
