// dx_qv_decode.hip -- Uncompress_Next_QVentry (QV.c:1428-1481) on the device.
//
// Reference behaviour reproduced (bit-exact):
//   Decode       QV.c:510-599   plain Huffman stream, 8-bit literal after the escape code (type 2)
//   Decode_Run   QV.c:604-691   run code (16-bit literal after run code 255) then one symbol
//   Packed_Length / Uncompress_Read / Lower_Read / Unpack_Tag   QV.c:823-847, DB.c:342-373
//   undexqv -U   undexqv.c:198-204
//
// A Huffman bit stream can only be decoded front to back, and the .dexqv format stores no index,
// so the parallelism is across (entry, stream) pairs: k_qv_decode gives every LANE one QV stream of
// one entry (a wavefront = the same stream kind of 64 entries) and walks it sequentially through a
// two-level table (11-bit primary look-up in LDS, linear list for the rare longer codes).  The
// segment starts come from the index k_qv_sizes produces (or, for a bare file, from the host
// walk dx_qv_walk).  k_qv_decode_tags then rebuilds the tag line of each entry from the decoded
// deletion line, one wavefront per entry, coalesced.
//
// Roofline: HBM (reads C, writes 5 bytes per base).  With a lane per stream the 64 lanes of a wave
// touch 64 different cache lines per access, so what matters is bytes moved per request: each lane
// reads its stream 16 bytes at a time into registers and stages its output in a private 64-byte LDS
// row that leaves as one contiguous burst (4-byte reads / 8-byte stores moved 13x / 4.8x the
// algorithmic bytes: 75 GB of HBM traffic per 10 GB decoded).
#include "dx_internal.hpp"
#include "dx_device.hpp"

struct dec_args
{ const uint8_t  *in;
  const uint64_t *rec_off;      // n+1
  const uint64_t *hdr_off;      // n+1 or NULL
  const uint32_t *seg;          // n x 5 segment byte sizes
  const uint32_t *len;
  uint64_t        n;
  uint8_t        *out;
  const uint64_t *out_off;
  int             delChar, subChar;
  int             type[4];      // scheme type of del/ins/mrg/sub
  int             upper;
  int             flip;         // words were written by a host of the other endianness (GETFLIP, QV.c:553-568)
};

// A code that is in no table -- a corrupt stream, or a line whose scheme holds a single symbol (its code has no bits:
// the reference writes such a file and its own undexqv then stops with "Could not read more bits (Decode)") -- cannot be
// decoded.  The readers keep moving (one bit) so that every loop ends, and leave a mark: one LDS word per workgroup,
// raised into the status word when the workgroup is done (DEC_ST_NOCODE), so that dx_qv_decode fails like the reference.
#define DEC_ST_NOCODE 16u
__shared__ uint32_t s_nocode;
__device__ __forceinline__ void nocode_begin() { if (threadIdx.x == 0) s_nocode = 0u; }             // (in front of the kernel's first barrier)
__device__ __forceinline__ void nocode_end(uint32_t *status)
{ __syncthreads();
  if (threadIdx.x == 0 && s_nocode) atomicOr(status, DEC_ST_NOCODE);
}

// MSB-first bit reader over little-endian 32-bit words; never reads past `end`.
//
// A wave waits on ALL of its outstanding global loads at once (one vmcnt per wave), so a lane that
// fetches the next piece of its own stream whenever it happens to run dry stalls the other 63 for
// a full memory latency -- and with 64 independent streams some lane runs dry in almost every
// iteration.  Here the lanes fetch together: each owns a 16-word ring in LDS plus one 16-byte chunk
// in flight in registers; br_pump (wave-uniform, once per decode iteration) fires when any lane is
// down to 4 unread words, commits every lane's chunk in flight to its ring (issued one pump ago:
// long since arrived) and issues the next.  Words then reach the bit buffer from the ring through a
// one-word prefetch register, never from memory.
#define DEC_RING_WORDS  16
#define DEC_RING_STRIDE 18                                 // words per row: 8-byte aligned, rows spread over the banks
struct bitrd
{ const uint8_t *p;              // next byte to fetch
  uint32_t       left;           // words of the segment not fetched yet
  uint32_t      *ring;           // this lane's LDS row
  uint32_t       rp, wp;         // words taken from / committed to the ring
  u32x4          pend;           // chunk in flight
  uint32_t       pend_words;     // its size (0: none)
  uint64_t       buf;            // bit buffer, next bit in bit 63
  int            nb;             // valid bits in buf
  uint32_t       nw;             // ring[rp - 1], prefetched
  bool           flip;
};

__device__ __forceinline__ void br_issue(bitrd &r)
{ if (r.left >= 4u)
    { r.pend = *(const u32x4_u *) r.p;
      r.pend_words = 4;
    }
  else                                                     // segments are whole words (QV.c:436-442)
    { r.pend.x = *(const u32_u *) r.p;
      r.pend.y = r.left >= 2u ? *(const u32_u *) (r.p + 4) : 0u;
      r.pend.z = r.left >= 3u ? *(const u32_u *) (r.p + 8) : 0u;
      r.pend.w = 0u;
      r.pend_words = r.left;
    }
  r.p    += 4 * r.pend_words;
  r.left -= r.pend_words;
}

__device__ __forceinline__ void br_commit(bitrd &r)
{ u32x4 v = r.pend;
  if (r.flip)
    { v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
      v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
    }
  uint64_t *slot = (uint64_t *) (r.ring + (r.wp & (DEC_RING_WORDS - 1)));
  slot[0] = ((uint64_t) v.y << 32) | v.x;
  slot[1] = ((uint64_t) v.w << 32) | v.z;
  r.wp += r.pend_words;
  r.pend_words = 0;
}

// once per decode iteration, all lanes of the wave together
__device__ __forceinline__ void br_pump(bitrd &r)
{ const int avail = (int) (r.wp - r.rp);
  if (__any(avail <= 4 && (r.pend_words | r.left) != 0u))
    { if (r.pend_words && avail <= DEC_RING_WORDS - 4)
        br_commit(r);
      if (!r.pend_words && r.left)
        br_issue(r);
    }
}

__device__ __forceinline__ void br_init(bitrd &r, const uint8_t *p, const uint8_t *end, bool flip, uint32_t *ring)
{ r.p = p; r.left = (uint32_t) ((end - p) >> 2); r.ring = ring; r.rp = 0; r.wp = 0; r.pend_words = 0;
  r.buf = 0; r.nb = 0; r.flip = flip;
  r.pend.x = r.pend.y = r.pend.z = r.pend.w = 0u;
  if (r.left) { br_issue(r); br_commit(r); }               // prime: one chunk in the ring ...
  if (r.left) br_issue(r);                                 // ... and one in flight
  r.nw = r.ring[0];
  r.rp = 1;
}

__device__ __forceinline__ void br_fill(bitrd &r)
{ if (r.nb <= 32)
    { r.buf |= (uint64_t) r.nw << (32 - r.nb);
      r.nb  += 32;
      r.nw   = r.ring[r.rp & (DEC_RING_WORDS - 1)];        // stale words past the segment's end are never decoded
      r.rp  += 1;
    }
}
__device__ __forceinline__ uint32_t br_peek16(const bitrd &r) { return (uint32_t) (r.buf >> 48); }
__device__ __forceinline__ void     br_skip(bitrd &r, int n)  { r.buf <<= n; r.nb -= n; }

// next code of scheme s: returns symbol, consumes its bits
__device__ __forceinline__ uint32_t dec_symbol_nofill(bitrd &r, const uint16_t *prim, const uint32_t *lng)
{ const uint32_t w = br_peek16(r);
  uint32_t e = prim[w >> (16 - DX_DEC_BITS)];
  if ((e >> 8) == 0)                                       // code longer than the primary index
    { const uint32_t cnt = lng[0];
      for (uint32_t k = 1; k <= cnt; k++)
        { const uint32_t t = lng[k], l = (t >> 8) & 0xffu;
          if ((w >> (16u - l)) == ((t >> 16) >> (16u - l)))
            { e = (l << 8) | (t & 0xffu);
              break;
            }
        }
      if ((e >> 8) == 0)                                   // no such code (see DEC_ST_NOCODE): keep moving, leave a mark
        { e |= 0x100u;
          s_nocode = 1u;
        }
    }
  br_skip(r, (int) (e >> 8));
  return e & 0xffu;
}

__device__ __forceinline__ uint32_t dec_symbol(bitrd &r, const uint16_t *prim, const uint32_t *lng)
{ br_fill(r);
  return dec_symbol_nofill(r, prim, lng);
}

// byte sink: 8 bytes gathered in a register go to the lane's LDS row (4 slots); a full row leaves
// as 32 contiguous bytes (two unaligned 16-byte stores back to back)
#define DEC_ROW_SLOTS 4
#define DEC_ROW_BYTES 40                                   // 32 used; a 10-dword stride spreads the rows over the banks
struct bsink { uint8_t *p; uint64_t acc; int cnt; uint64_t *row; int slot; };

__device__ __forceinline__ void bs_push8(bsink &o)
{ o.row[o.slot] = o.acc;
  o.acc = 0; o.cnt = 0;
  if (++o.slot == DEC_ROW_SLOTS)
    { const uint64_t a0 = o.row[0], a1 = o.row[1], a2 = o.row[2], a3 = o.row[3];
      const u32x4 v0 = { (uint32_t) a0, (uint32_t) (a0 >> 32), (uint32_t) a1, (uint32_t) (a1 >> 32) };
      const u32x4 v1 = { (uint32_t) a2, (uint32_t) (a2 >> 32), (uint32_t) a3, (uint32_t) (a3 >> 32) };
      u32x4_u *g = (u32x4_u *) o.p;
      g[0] = v0; g[1] = v1;
      o.p += 8 * DEC_ROW_SLOTS; o.slot = 0;
    }
}

__device__ __forceinline__ void bs_put(bsink &o, uint32_t b)
{ o.acc |= (uint64_t) b << (8 * o.cnt);
  if (++o.cnt == 8)
    bs_push8(o);
}

__device__ __forceinline__ void bs_fill(bsink &o, uint32_t b, uint32_t count)
{ const uint64_t pat = (uint64_t) b * 0x0101010101010101ull;
  while (count)
    { const uint32_t take = count < (uint32_t) (8 - o.cnt) ? count : (uint32_t) (8 - o.cnt);
      const uint64_t piece = take == 8u ? pat : (pat & ((1ull << (8u * take)) - 1ull));
      o.acc |= piece << (8 * o.cnt);
      o.cnt += (int) take;
      count -= take;
      if (o.cnt == 8)
        bs_push8(o);
    }
}

__device__ __forceinline__ void bs_end(bsink &o)
{ for (int k = 0; k < o.slot; k++)
    { *(u64_u *) o.p = o.row[k];
      o.p += 8;
    }
  for (int k = 0; k < o.cnt; k++)
    o.p[k] = (uint8_t) (o.acc >> (8 * k));
  o.p += o.cnt;
  o.slot = 0;
}

// The same sink for a run-coded line, cheaper: the 8-byte register starts out as eight run characters, so a run only
// moves the position (whole registers of run characters leave as they are) and a symbol is one XOR into its byte.
struct rsink { uint8_t *p; uint64_t acc, pat; uint32_t sh /* bit position in acc: 8 * bytes held */; uint64_t *row; int slot; uint32_t rc; };

__device__ __forceinline__ void rs_push8(rsink &o)
{ o.row[o.slot] = o.acc;
  o.acc = o.pat;
  if (++o.slot == DEC_ROW_SLOTS)
    { const uint64_t a0 = o.row[0], a1 = o.row[1], a2 = o.row[2], a3 = o.row[3];
      const u32x4 v0 = { (uint32_t) a0, (uint32_t) (a0 >> 32), (uint32_t) a1, (uint32_t) (a1 >> 32) };
      const u32x4 v1 = { (uint32_t) a2, (uint32_t) (a2 >> 32), (uint32_t) a3, (uint32_t) (a3 >> 32) };
      u32x4_u *g = (u32x4_u *) o.p;
      g[0] = v0; g[1] = v1;
      o.p += 8 * DEC_ROW_SLOTS; o.slot = 0;
    }
}

__device__ __forceinline__ void rs_run(rsink &o, uint32_t count)
{ o.sh += 8u * count;
  while (o.sh >= 64u)
    { rs_push8(o);
      o.sh -= 64u;
    }
}

__device__ __forceinline__ void rs_sym(rsink &o, uint32_t b)
{ o.acc ^= (uint64_t) (b ^ o.rc) << o.sh;
  o.sh  += 8u;
  if (o.sh == 64u)
    { rs_push8(o);
      o.sh = 0;
    }
}

__device__ __forceinline__ void rs_end(rsink &o)
{ for (int k = 0; k < o.slot; k++)
    { *(u64_u *) o.p = o.row[k];
      o.p += 8;
    }
  for (uint32_t k = 0; k < (o.sh >> 3); k++)
    o.p[k] = (uint8_t) (o.acc >> (8 * k));
}

#define DEC_BLOCK 1024                                     // one 16-wave workgroup per CU (142 KB of LDS)
#define DEC_NWAVE (DEC_BLOCK / 64)

__global__ __launch_bounds__(DEC_BLOCK)
void k_qv_decode(dec_args a, const uint16_t *g_dec, const uint32_t *g_long, uint32_t *status, uint32_t *next_task,
                 uint32_t kinds /* bit q set: this launch decodes stream kind q (0 del, 1 ins, 2 mrg, 3 sub) */,
                 const uint32_t *skip_idx, const uint64_t *skip_off, uint32_t skip_kinds /* run-coded kinds whose indexed
                 lines k_qv_decode_runs has decoded: only the lines without an index (RUN_NONE) are left for this kernel */,
                 const uint32_t *none_count /* how many such lines the batch has */,
                 uint32_t first_task /* tasks before this one belong to kinds not in `kinds` (the plain lines come first) */)
{ __shared__ uint16_t s_dec[6][DX_DEC_SIZE];               // 24 KB
  __shared__ uint32_t s_long[6][1 + DX_LONG_MAX];          // 6 KB
  __shared__ __attribute__((aligned(8))) uint8_t s_row[DEC_BLOCK][DEC_ROW_BYTES];    // 40 KB
  __shared__ __attribute__((aligned(8))) uint32_t s_ring[DEC_BLOCK][DEC_RING_STRIDE]; // 72 KB
  if (skip_idx != NULL && (kinds & ~skip_kinds) == 0u && *none_count == 0u)
    return;                                                // every line of this launch had its index: nothing left
  nocode_begin();
  for (int k = threadIdx.x; k < 6 * DX_DEC_SIZE; k += DEC_BLOCK)          (&s_dec[0][0])[k]  = g_dec[k];
  for (int k = threadIdx.x; k < 6 * (1 + DX_LONG_MAX); k += DEC_BLOCK)    (&s_long[0][0])[k] = g_long[k];
  __syncthreads();

  // A task = ONE stream kind (q) of 64 consecutive entries, one entry per lane: all lanes of the
  // wavefront then run the same loop (plain or run-coded) and only the trip counts differ.  Tasks
  // cost very different amounts (a run-coded line has a fifth of the tokens of a plain one), so
  // the waves draw them from a counter, the expensive plain lines first.
  const uint64_t ngroup = (a.n + 63) / 64;                 // groups of 64 entries
  for (;;)
    { uint32_t t = 0;
      if (lane_id() == 0)
        t = atomicAdd(next_task, 1u) + first_task;
      t = uniform(t);
      if ((uint64_t) t >= 4 * ngroup) break;               // every wave gets here: the counter only grows
      // order: ins and mrg of every group, then del and sub
      const uint64_t g = (uint64_t) t < 2 * ngroup ? t >> 1 : ((uint64_t) t - 2 * ngroup) >> 1;
      const int      q = (uint64_t) t < 2 * ngroup ? 1 + (int) (t & 1u) : ((t & 1u) ? 3 : 0);
      const uint64_t r = g * 64 + (uint64_t) lane_id();
      if (!((kinds >> q) & 1u))
        continue;                                          // k_qv_decode_plain has this stream kind
      bool mine = r < a.n;                                 // lanes past the last entry idle through this task
      if (mine && skip_idx != NULL && ((skip_kinds >> q) & 1u))
        { const uint32_t Lr = a.len[r];
          mine = skip_idx[skip_off[r] + run_base(Lr) + (q == 0 ? 0u : 1u)] == RUN_NONE;
        }
      if (mine)
      {
      const int      line = q == 0 ? 0 : q + 1;            // output line / segment index
      const uint32_t L  = a.len[r];
      const uint32_t *sg = a.seg + 5 * r;
      uint64_t at = a.rec_off[r] + (a.hdr_off ? a.hdr_off[r + 1] - a.hdr_off[r] : 0);
      for (int k = 0; k < line; k++)
        at += sg[k];

      bitrd rd;
      br_init(rd, a.in + at, a.in + at + sg[line], a.flip != 0, s_ring[threadIdx.x]);
      bsink o  = { a.out + a.out_off[r] + (uint64_t) line * ((uint64_t) L + 1u), 0, 0, (uint64_t *) s_row[threadIdx.x], 0 };
      const int rc = q == 0 ? a.delChar : (q == 3 ? a.subChar : -1);
      const uint16_t *prim = s_dec[q];
      const uint32_t *lng  = s_long[q];
      const bool esc = a.type[q] == 2;
      uint32_t j = 0, bad = 0;

      if (rc < 0 && !esc)                                  // Decode, QV.c:586-596; codes <= 16 bits: two per refill
        { while (j + 2u <= L)
            { br_pump(rd);
              br_fill(rd);                                 // >= 33 bits: enough for two codes
              const uint32_t c0 = dec_symbol_nofill(rd, prim, lng);
              const uint32_t c1 = dec_symbol_nofill(rd, prim, lng);
              bs_put(o, c0);
              bs_put(o, c1);
              j += 2;
            }
          if (j < L)
            { br_pump(rd);
              bs_put(o, dec_symbol(rd, prim, lng));
              j += 1;
            }
        }
      else if (rc < 0)
        while (j < L)
          { br_pump(rd);
            uint32_t c = dec_symbol(rd, prim, lng);
            if (esc && c == 255u)
              { br_fill(rd);
                c = br_peek16(rd) >> 8;
                br_skip(rd, 8);
              }
            bs_put(o, c);
            j += 1;
          }
      else if (!esc)                                       // Decode_Run, QV.c:665-688, symbols without an escape code:
        { const uint16_t *rprim = s_dec[q == 0 ? DX_DRUN : DX_SRUN];       // one refill covers run code + symbol code
          const uint32_t *rlng  = s_long[q == 0 ? DX_DRUN : DX_SRUN];
          rsink ro;
          ro.p = o.p; ro.rc = (uint32_t) rc; ro.pat = (uint64_t) (uint32_t) rc * 0x0101010101010101ull; ro.acc = ro.pat;
          ro.sh = 0; ro.row = o.row; ro.slot = 0;
          while (j < L)
            { br_pump(rd);
              br_fill(rd);                                 // >= 33 bits
              uint32_t c = dec_symbol_nofill(rd, rprim, rlng);
              if (c == 255u)                               // run of 255 or more: 16-bit literal (QV.c:670-676)
                { br_fill(rd);
                  c = br_peek16(rd);
                  br_skip(rd, 16);
                  br_fill(rd);
                }
              if (c > L - j) { bad = 1; c = L - j; }       // corrupt stream: never write past the line
              rs_run(ro, c);
              j += c;
              if (j < L)
                { rs_sym(ro, dec_symbol_nofill(rd, prim, lng));
                  j += 1;
                }
            }
          rs_sym(ro, '\n');
          rs_end(ro);
          if (bad) atomicOr(status, 4u);
          continue;                                        // (the line is complete: its end is written above)
        }
      else                                                 // ... with one (scheme type 2)
        { const uint16_t *rprim = s_dec[q == 0 ? DX_DRUN : DX_SRUN];
          const uint32_t *rlng  = s_long[q == 0 ? DX_DRUN : DX_SRUN];
          while (j < L)
            { br_pump(rd);
              uint32_t c = dec_symbol(rd, rprim, rlng);
              if (c == 255u)
                { br_fill(rd);
                  c = br_peek16(rd);
                  br_skip(rd, 16);
                }
              if (c > L - j) { bad = 1; c = L - j; }       // corrupt stream: never write past the line
              bs_fill(o, (uint32_t) rc, c);
              j += c;
              if (j < L)
                { uint32_t x = dec_symbol(rd, prim, lng);
                  if (esc && x == 255u)
                    { br_fill(rd);
                      x = br_peek16(rd) >> 8;
                      br_skip(rd, 8);
                    }
                  bs_put(o, x);
                  j += 1;
                }
            }
        }
      bs_put(o, '\n');
      bs_end(o);
      if (bad) atomicOr(status, 4u);
      }
    }
  nocode_end(status);
}

// ---------------------------------------------------------------------------------------------
//  plain lines (Decode, QV.c:510-599, schemes without escape): k_qv_decode_plain
// ---------------------------------------------------------------------------------------------
// Same division of the work (a lane per entry, a wavefront = one stream kind of 64 consecutive entries),
// a much shorter path per code -- the plain lines are three quarters of all codes:
//  * input by whole aligned 64-byte lines: a lane fetches the next line of its stream with four 16-byte loads
//    (nothing is fetched twice: the generic kernel's 16-byte reads cost 3.9 x the stream's bytes in memory
//    requests) into a two-line ring in LDS.  A lane whose ring has room for a line issues the loads before a
//    block of 8 codes and writes them to the ring behind it: the load's latency runs under the block, and no
//    load is in flight across the loop's back edge (the compiler would wait there for EVERY outstanding memory
//    operation, the block's own output store included -- that wait once cost 10 000 cycles per 16 codes).  The
//    stream's 32-bit words (the segment may start at any byte) come out of the ring by v_alignbyte;
//  * a 12-bit primary table (codes beyond it are rarer than 1 in 4096 symbols by construction of a Huffman
//    code) whose 16-bit entries carry what the steps need where they need it: the low bits = 32 - length
//    (v_alignbit takes its shift from there, for both halves of the bit buffer), byte 1 = symbol (v_perm
//    drops it into the output word): 8 vector instructions and one look-up per code;
//  * all lanes decode code j at the same time, so the output position is a compile-time quantity: 16 symbols
//    make four registers that leave as one 16-byte store; a refill (one word) is needed about every 9th code;
//  * a block of 16 symbols in which any lane meets a code beyond the primary index is decoded again, symbol
//    by symbol, from the saved reader state.
#ifndef DP_BLOCK
#define DP_BLOCK   1024                                    // 16 waves: rings 100 KB + tables 36 KB of LDS
#endif
#define DP_NWAVE   (DP_BLOCK / 64)
#ifndef DP_RING
#define DP_RING    24                                      // dwords per lane: three half-lines
#endif
#define DP_WG_PER_CU ((160 * 1024) / ((DP_BLOCK * (DP_RING + 1) * 4) + 37 * 1024))   // rings + 36.6 KB of tables
#define DP_STRIDE  (DP_RING + 1)                           // row stride in dwords (rows spread over the banks)

struct linerd
{ const uint8_t *next;           // next aligned half-line (32 bytes) to fetch
  uint32_t       halves;         // half-lines not yet fetched
  uint32_t      *ring;           // this lane's LDS row
  uint32_t       avail;          // dwords in the ring not yet read
  uint32_t       ri, wi;         // ring positions (0 .. DP_RING - 1) of the next read / the next write
  uint32_t       prev, nxt;      // the next word's two halves, prefetched from the ring
  uint32_t       o;              // byte offset of the stream's words in the aligned dwords (0..3)
  uint32_t       hi, lo;         // bit buffer: next bit in bit 31 of hi
  int            nb;             // valid bits
};

struct half32 { u32x4 a, b; };

__device__ __forceinline__ half32 lr_load(linerd &r)
{ const u32x4 *g = (const u32x4 *) r.next;
  half32 v = { g[0], g[1] };
  r.next   += 32;
  r.halves -= 1;
  return v;
}

__device__ __forceinline__ void lr_store(linerd &r, const half32 &v)
{ uint32_t *w = r.ring + r.wi;                             // (wi is a multiple of 8: a half-line never wraps)
  w[0] = v.a.x; w[1] = v.a.y; w[2] = v.a.z; w[3] = v.a.w;
  w[4] = v.b.x; w[5] = v.b.y; w[6] = v.b.z; w[7] = v.b.w;
  r.wi     = r.wi == DP_RING - 8 ? 0u : r.wi + 8u;
  r.avail += 8;
}

// Ring discipline (blocks of 8 codes = at most 4 words + 2 of look-ahead): a lane with <= 16 unread dwords
// fetches a half-line (two 16-byte loads) before the block and stores it behind the block, where at most 16 are
// unread for sure (the ring holds 24); one with more does not, and has >= 13 left behind the block.  So a block
// starts with >= 13 unread dwords, always.
__device__ __forceinline__ bool lr_wants(const linerd &r) { return r.halves != 0u && r.avail <= DP_RING - 8u; }

__device__ __forceinline__ uint32_t lr_next(linerd &r)      // next dword of the ring
{ const uint32_t v = r.ring[r.ri];
  r.ri     = r.ri == DP_RING - 1 ? 0u : r.ri + 1u;
  r.avail -= 1;
  return v;
}

__device__ __forceinline__ void lr_init(linerd &r, const uint8_t *seg, uint32_t bytes, uint32_t *ring)
{ const uintptr_t A = (uintptr_t) seg;
  r.next   = seg - (A & 31u);                              // (pointer arithmetic, not a cast: the loads stay global_load)
  r.halves = bytes ? (uint32_t) (((A + bytes - 1) >> 5) - (A >> 5)) + 1u : 0u;
  r.ring   = ring; r.avail = 0; r.ri = 0; r.wi = 0;
  r.o      = (uint32_t) (A & 3u);
  r.hi = r.lo = 0; r.nb = 0;
  #pragma unroll
  for (int k = 0; k < DP_RING / 8; k++)                    // the ring starts full
    if (r.halves) { const half32 v = lr_load(r); lr_store(r, v); }
  r.ri    = (uint32_t) ((A & 31u) >> 2);                   // dword of the first half-line the stream starts in
  r.avail = r.avail > r.ri ? r.avail - r.ri : 0u;
  r.prev  = lr_next(r);
  r.nxt   = lr_next(r);
}

// next 32-bit word of the stream (MSB-first bit order within little-endian words, QV.c:553-568)
__device__ __forceinline__ uint32_t lr_word(linerd &r, bool flip)
{ uint32_t w = __builtin_amdgcn_alignbyte(r.nxt, r.prev, r.o);
  r.prev = r.nxt;
  r.nxt  = lr_next(r);
  if (flip) w = __builtin_bswap32(w);
  return w;
}

// at least 32 valid bits afterwards (a refill adds a word behind the valid bits)
__device__ __forceinline__ void lr_fill(linerd &r, bool flip)
{ if (r.nb < 32)
    { const uint32_t w = lr_word(r, flip);
      r.hi |= w >> r.nb;
      r.lo  = __builtin_amdgcn_alignbit(w, 0u, (uint32_t) r.nb);       // w << (32 - nb); 0 for nb == 0
      r.nb += 32;
    }
}

#define DP_BITS 12                                         // primary index of k_qv_decode_plain
#define DP_SIZE (1 << DP_BITS)

// code by code (blocks with a long code, the last symbols of a line): symbol, bits consumed
__device__ __forceinline__ uint32_t lr_symbol(linerd &r, const uint16_t *tab, const uint32_t *lng, bool flip)
{ lr_fill(r, flip);
  const uint32_t w = r.hi >> 16;
  const uint32_t e = tab[w >> (16 - DP_BITS)];
  uint32_t len = (e & 31u) ? 32u - (e & 31u) : 0u, sym = e >> 8;
  if (len == 0)                                            // code longer than the primary index
    { const uint32_t cnt = lng[0];
      for (uint32_t k = 1; k <= cnt; k++)
        { const uint32_t t = lng[k], l = (t >> 8) & 0xffu;
          if ((w >> (16u - l)) == ((t >> 16) >> (16u - l)))
            { len = l; sym = t & 0xffu;
              break;
            }
        }
      if (len == 0) { len = 1; s_nocode = 1u; }            // no such code (see DEC_ST_NOCODE): keep moving, leave a mark
    }
  r.hi  = __builtin_amdgcn_alignbit(r.hi, r.lo, 32u - len);
  r.lo <<= len;
  r.nb -= (int) len;
  return sym;
}

// one block: 8 codes into two output words, with the block's own half-line fetch around it
__device__ __forceinline__ void dp_block8(linerd &rd, const uint16_t *tab, const uint32_t *lng, bool flip, uint32_t &o0, uint32_t &o1)
{ const uint32_t s_hi = rd.hi, s_lo = rd.lo, s_ri = rd.ri, s_av = rd.avail, s_prev = rd.prev, s_nxt = rd.nxt;
  const int      s_nb = rd.nb;                             // (the block touches nothing else of the reader)
  const bool     fetch = lr_wants(rd);
  half32 ln = { { 0u, 0u, 0u, 0u }, { 0u, 0u, 0u, 0u } };
  if (fetch) ln = lr_load(rd);
  uint32_t w[2] = { 0u, 0u }, zand = 31u;
  #pragma unroll
  for (int k = 0; k < 8; k += 2)
    { lr_fill(rd, flip);                                   // >= 32 bits: enough for two codes of <= 16
      #pragma unroll
      for (int h = 0; h < 2; h++)
        { const uint32_t e = tab[rd.hi >> (32 - DP_BITS)];
          zand &= e;                                       // 32 - len is 16..31 (bit 4 set) unless the code is longer than the index
          rd.hi = __builtin_amdgcn_alignbit(rd.hi, rd.lo, e);          // << len: the shift 32 - len sits in e's low bits
          rd.lo = __builtin_amdgcn_alignbit(rd.lo, 0u, e);
          rd.nb += (int) (e & 31u) - 32;
          // symbol (byte 1 of e) into byte (k + h) & 3 of the output word
          w[(k + h) >> 2] = __builtin_amdgcn_perm(e, w[(k + h) >> 2],
                                                  ((k + h) & 3) == 0 ? 0x03020105u : ((k + h) & 3) == 1 ? 0x03020500u :
                                                  ((k + h) & 3) == 2 ? 0x03050100u : 0x05020100u);
        }
    }
  if (__any((int) (~zand & 16u)))                          // a long code somewhere: this block again, code by code
    { rd.hi = s_hi; rd.lo = s_lo; rd.ri = s_ri; rd.avail = s_av; rd.prev = s_prev; rd.nxt = s_nxt; rd.nb = s_nb;
      #pragma unroll 1
      for (int k = 0; k < 8; k++)
        { const uint32_t c = lr_symbol(rd, tab, lng, flip);
          w[k >> 2] = (k & 3) ? (w[k >> 2] | (c << (8 * (k & 3)))) : c;
        }
    }
  if (fetch) lr_store(rd, ln);                             // at most 16 dwords are unread now: 8 of the ring's 24 are free
  o0 = w[0]; o1 = w[1];
}

// the 12-bit tables of the four symbol schemes from the library's 11-bit ones and their lists of longer codes
// (QV.c:365-372 in two levels); every thread of the workgroup takes part
__device__ __forceinline__ void dp_build_tables(uint16_t (*s_tab)[DP_SIZE], uint32_t (*s_long)[1 + DX_LONG_MAX],
                                                const uint16_t *g_dec, const uint32_t *g_long)
{ for (int k = threadIdx.x; k < 4 * (1 + DX_LONG_MAX); k += (int) blockDim.x) (&s_long[0][0])[k] = g_long[k];
  __syncthreads();
  for (int k = threadIdx.x; k < 4 * DP_SIZE; k += (int) blockDim.x)
    { const int      sc = k >> DP_BITS;
      const uint32_t i  = (uint32_t) k & (DP_SIZE - 1), e = g_dec[sc * DX_DEC_SIZE + (i >> (DP_BITS - DX_DEC_BITS))];
      uint32_t len = e >> 8, sym = e & 0xffu;
      if (len == 0)                                        // longer than 11 bits: exactly DP_BITS long?
        { const uint32_t *lg = s_long[sc], cnt = lg[0], pre = i << (16 - DP_BITS);
          for (uint32_t j = 1; j <= cnt; j++)
            { const uint32_t t = lg[j], l = (t >> 8) & 0xffu;
              if (l <= DP_BITS && (pre >> (16u - l)) == ((t >> 16) >> (16u - l)))
                { len = l; sym = t & 0xffu; }              // (ascending symbol order: the last match wins, as in the list)
            }
        }
      (&s_tab[0][0])[k] = (uint16_t) ((len ? 32u - len : 0u) | (sym << 8));
    }
  __syncthreads();
}

__global__ __launch_bounds__(DP_BLOCK)
void k_qv_decode_plain(dec_args a, const uint16_t *g_dec, const uint32_t *g_long, uint32_t *next_task, uint32_t kinds,
                       uint32_t *status, const uint32_t *sync_idx /* the device walk's index, or NULL: of the kinds in sync_kinds only the
                       lines without their words are this kernel's (k_qv_decode_sync has decoded the others) */,
                       const uint64_t *sync_off, uint32_t sync_kinds)
{ __shared__ uint16_t s_tab[4][DP_SIZE];                   // 32 KB: 32 - len | symbol << 8 (low bits 0: longer than DP_BITS)
  __shared__ uint32_t s_long[4][1 + DX_LONG_MAX];          //  4 KB
  __shared__ uint32_t s_ring[DP_BLOCK][DP_STRIDE];         // 100 KB
  nocode_begin();
  dp_build_tables(s_tab, s_long, g_dec, g_long);

  const uint64_t ngroup = (a.n + 63) / 64;
  const bool     flip   = a.flip != 0;
  for (;;)
    { uint32_t t = 0;
      if (lane_id() == 0)
        t = atomicAdd(next_task, 1u);
      t = uniform(t);
      if ((uint64_t) t >= 4 * ngroup) break;               // every wave gets here: the counter only grows
      const uint64_t g = (uint64_t) t < 2 * ngroup ? t >> 1 : ((uint64_t) t - 2 * ngroup) >> 1;   // same order as k_qv_decode
      const int      q = (uint64_t) t < 2 * ngroup ? 1 + (int) (t & 1u) : ((t & 1u) ? 3 : 0);
      if (!((kinds >> q) & 1u))
        continue;
      const uint64_t r    = g * 64 + (uint64_t) lane_id();
      bool           live = r < a.n;
      const int      line = q == 0 ? 0 : q + 1;            // output line / segment index
      uint32_t L = 0, sbytes = 0;
      const uint8_t *seg = a.in;
      uint8_t *out = a.out;
      if (live && sync_idx != NULL && ((sync_kinds >> q) & 1u))
        { const uint32_t Lr = a.len[r];
          live = Lr != 0u && sync_idx[sync_off[r] + (uint64_t) q * sub_words(Lr)] == DXL_SYNC_NONE;
        }
      if (live)
        { const uint32_t *sg = a.seg + 5 * r;
          uint64_t at = a.rec_off[r] + (a.hdr_off ? a.hdr_off[r + 1] - a.hdr_off[r] : 0);
          for (int k = 0; k < line; k++)
            at += sg[k];
          L      = a.len[r];
          sbytes = sg[line];
          seg    = a.in + at;
          out    = a.out + a.out_off[r] + (uint64_t) line * ((uint64_t) L + 1u);
        }
      linerd rd;
      lr_init(rd, seg, live ? sbytes : 0u, s_ring[threadIdx.x]);
      const uint16_t *tab = s_tab[q];
      const uint32_t *lng = s_long[q];
      uint32_t j = 0;

      // 64 symbols per round, all lanes at the same symbol: 16 registers that leave as four 16-byte stores to
      // consecutive addresses (a whole line's worth at once: single 16-byte stores per round cost as much as the decoding)
      while (__any(j + 64u <= L))
        { if (j + 64u <= L)
            { uint32_t w[16];
              #pragma unroll
              for (int b = 0; b < 8; b++)
                dp_block8(rd, tab, lng, flip, w[2 * b], w[2 * b + 1]);
              u32x4_u *g = (u32x4_u *) (out + j);
              const u32x4 v0 = { w[0], w[1], w[2], w[3] },    v1 = { w[4], w[5], w[6], w[7] };
              const u32x4 v2 = { w[8], w[9], w[10], w[11] },  v3 = { w[12], w[13], w[14], w[15] };
              g[0] = v0; g[1] = v1; g[2] = v2; g[3] = v3;
              j += 64;
            }
        }
      while (__any(j + 8u <= L))                           // the blocks left
        { if (j + 8u <= L)
            { uint32_t w0, w1;
              dp_block8(rd, tab, lng, flip, w0, w1);
              *(u32_u *) (out + j)      = w0;
              *(u32_u *) (out + j + 4u) = w1;
              j += 8;
            }
        }
      if (live)                                            // the last symbols of the line, and its end
        { while (j < L)                                    // (< 8 codes = at most 4 words: the ring holds >= 13)
            { out[j] = (uint8_t) lr_symbol(rd, tab, lng, flip);
              j += 1;
            }
          out[L] = '\n';
        }
    }
  nocode_end(status);
}

// ---------------------------------------------------------------------------------------------
//  an entry's whereabouts in one load (the wave-per-line kernels)
// ---------------------------------------------------------------------------------------------
// What says where an entry's lines are -- its length, record, segment sizes, place in the output, share of the index:
// 18 words from six arrays -- is ONE load whose lanes fetch a word each (DR_DESC), requested while the entry before it
// is decoded and taken apart with v_readlane when its turn comes (DR_TAKE).  The macros use the kernel's a, sub_off, lane.
struct dr_entry { uint32_t L, hl, sg[5]; uint64_t so, so1, rec, oo; };
// Entries are drawn DEC_TICKET at a time: an atomic on one address completes every ~11 ns chip-wide, so a ticket per
// entry is an 11 ms floor under a kernel over a million entries, whatever else it does.  A wave knows the entry after
// the current one a whole entry ahead: the next of its ticket, or the first of the ticket it drew when this one began.
#ifndef DEC_TICKET
#define DEC_TICKET 4u
#endif
#define DT_FIRST(R, R_END, TKV)                                                                                         \
  const uint32_t dec_ticket = ticket_units_of(next_task, DEC_TICKET);       /* (k_ticket_units in front of the launch) */ \
  { uint32_t t0_ = 0;                                                                                                   \
    if (lane == 0) { t0_ = atomicAdd(next_task, 1u); TKV = atomicAdd(next_task, 1u); }                                  \
    R = (uint64_t) uniform(t0_) * dec_ticket; R_END = R + dec_ticket;                                                   \
  }
#define DT_NEXT(R1, FRESH, R, R_END, TKV)                                                                               \
  { FRESH = (R) + 1u >= (R_END);                                                                                        \
    R1    = (R) + 1u;                                                                                                   \
    if (FRESH)                                                                                                          \
      { R1 = (uint64_t) uniform(TKV) * dec_ticket;                                                                      \
        if (lane == 0) TKV = atomicAdd(next_task, 1u);                                                                  \
      }                                                                                                                 \
  }
#define DR_RL(V, K)  ((uint32_t) __builtin_amdgcn_readlane((int) (V), K))
#define DR_DESC(V, R)                                                                                                   \
  { const uint32_t *p_ = a.len + (R);                                                                                   \
    if (lane >= 1  && lane <= 4)  p_ = (const uint32_t *) (sub_off + (R)) + (lane - 1);                                 \
    if (lane >= 5  && lane <= 6)  p_ = (const uint32_t *) (a.rec_off + (R)) + (lane - 5);                               \
    if (lane >= 7  && lane <= 8)  p_ = (const uint32_t *) (a.out_off + (R)) + (lane - 7);                               \
    if (lane >= 9  && lane <= 12 && a.hdr_off) p_ = (const uint32_t *) (a.hdr_off + (R)) + (lane - 9);                  \
    if (lane >= 13 && lane <= 17) p_ = a.seg + 5 * (R) + (lane - 13);                                                   \
    V = *p_;                                                                                                            \
  }
#define DR_TAKE(E, V)                                                                                                   \
  { E.L   = DR_RL(V, 0);                                                                                                \
    E.so  = (uint64_t) DR_RL(V, 1) | ((uint64_t) DR_RL(V, 2) << 32);                                                    \
    E.so1 = (uint64_t) DR_RL(V, 3) | ((uint64_t) DR_RL(V, 4) << 32);                                                    \
    E.rec = (uint64_t) DR_RL(V, 5) | ((uint64_t) DR_RL(V, 6) << 32);                                                    \
    E.oo  = (uint64_t) DR_RL(V, 7) | ((uint64_t) DR_RL(V, 8) << 32);                                                    \
    E.hl  = a.hdr_off ? DR_RL(V, 11) - DR_RL(V, 9) : 0u;         /* (a header is far below 4 GB: low words do) */       \
    E.sg[0] = DR_RL(V, 13); E.sg[1] = DR_RL(V, 14); E.sg[2] = DR_RL(V, 15); E.sg[3] = DR_RL(V, 16); E.sg[4] = DR_RL(V, 17); \
  }

// ---------------------------------------------------------------------------------------------
//  plain lines with the encoder's group index (dx_qv_subindex): k_qv_decode_sub
// ---------------------------------------------------------------------------------------------
// A wavefront per (entry, plain line).  The index holds the code bits of every group of 16 symbols; per round the
// wave takes the next DS_STEPS x 64 groups, turns the bit counts into start offsets with prefix sums, puts the
// words they span into its LDS window (coalesced dwords, the stream's word alignment restored on the way) and
// lane b decodes groups b, 64 + b, ... of the round: the 64 lanes' 16-byte stores of one step are 1 KiB of
// consecutive addresses -- whole lines leave the L2 once -- where a lane per line touches 64 different lines per
// access.  A whole group is decoded by positioned reads of the window, two codes per read (ds_block8_pos); a group
// with a code beyond the tables' 12-bit index, and what the positioned path cannot take, by the code sequence of
// k_qv_decode_plain's blocks (bit buffer, refills that are plain LDS reads).  A line marked SUB_NONE (a symbol
// without a code: hand-made tables only) is decoded group after group by one lane.  Every round requests what the
// NEXT round needs (see the kernel).
// LDS: only the tables of the NK plain kinds present (9 KB each) + a 5 KB window per wave; with the usual two
// plain lines (ins, mrg) two 12-wave workgroups share a CU (78 KB each), 6 waves per SIMD.
#define DS_BLOCK 768
#define DS_NWAVE (DS_BLOCK / 64)
#define DS_WIN   1280                                       // words per wave: 5 KB
#ifndef DS_WG_PER_CU
#define DS_WG_PER_CU 6                                      // (waves per SIMD, HIP's second launch bound: two 12-wave workgroups per CU with the usual two tables, <= 80 VGPRs)
#endif
#ifndef DS_NPRE
#define DS_NPRE  5                                          // words per lane of the coming round's window requested a round ahead
#endif
#ifndef DS_POS
#define DS_POS   1                                         // whole groups decoded by positioned reads, no bit buffer (ds_block8_pos)
#endif
#ifndef DS_DUAL
#define DS_DUAL  0                                         // two groups per lane decoded side by side (two look-up chains in flight)
#endif
#ifndef DS_STEPS
#define DS_STEPS 2                                         // steps of 64 groups per round (one step always fits: <= 515 words)
#endif

struct winrd
{ const uint32_t *win;
  uint32_t wi;                   // next word of the window
  uint32_t hi, lo;               // bit buffer: next bit in bit 31 of hi
  int      nb;                   // valid bits
};

__device__ __forceinline__ void wr_fill(winrd &r)           // at least 32 valid bits afterwards
{ if (r.nb < 32)
    { const uint32_t w = r.win[r.wi];
      r.wi += 1;
      r.hi |= w >> r.nb;
      r.lo  = __builtin_amdgcn_alignbit(w, 0u, (uint32_t) r.nb);        // w << (32 - nb); 0 for nb == 0
      r.nb += 32;
    }
}

__device__ __forceinline__ uint32_t wr_symbol(winrd &r, const uint16_t *tab, const uint32_t *lng)
{ wr_fill(r);
  const uint32_t w = r.hi >> 16;
  const uint32_t e = tab[w >> (16 - DP_BITS)];
  uint32_t len = (e & 31u) ? 32u - (e & 31u) : 0u, sym = e >> 8;
  if (len == 0)                                            // code longer than the primary index
    { const uint32_t cnt = lng[0];
      for (uint32_t k = 1; k <= cnt; k++)
        { const uint32_t t = lng[k], l = (t >> 8) & 0xffu;
          if ((w >> (16u - l)) == ((t >> 16) >> (16u - l)))
            { len = l; sym = t & 0xffu;
              break;
            }
        }
      if (len == 0) { len = 1; s_nocode = 1u; }            // no such code (see DEC_ST_NOCODE): keep moving, leave a mark
    }
  r.hi  = __builtin_amdgcn_alignbit(r.hi, r.lo, 32u - len);
  r.lo <<= len;
  r.nb -= (int) len;
  return sym;
}

// 8 codes into two output words
__device__ __forceinline__ void ds_block8(winrd &rd, const uint16_t *tab, const uint32_t *lng, uint32_t &o0, uint32_t &o1)
{ const winrd saved = rd;
  uint32_t w[2] = { 0u, 0u }, zand = 31u;
  #pragma unroll
  for (int k = 0; k < 8; k += 2)
    { wr_fill(rd);                                         // >= 32 bits: enough for two codes of <= 16
      #pragma unroll
      for (int h = 0; h < 2; h++)
        { const uint32_t e = tab[rd.hi >> (32 - DP_BITS)];
          zand &= e;                                       // 32 - len is 16..31 (bit 4 set) unless the code is longer than the index
          rd.hi = __builtin_amdgcn_alignbit(rd.hi, rd.lo, e);
          rd.lo = __builtin_amdgcn_alignbit(rd.lo, 0u, e);
          rd.nb += (int) (e & 31u) - 32;
          w[(k + h) >> 2] = __builtin_amdgcn_perm(e, w[(k + h) >> 2],
                                                  ((k + h) & 3) == 0 ? 0x03020105u : ((k + h) & 3) == 1 ? 0x03020500u :
                                                  ((k + h) & 3) == 2 ? 0x03050100u : 0x05020100u);
        }
    }
  if (__any((int) (~zand & 16u)))                          // a long code somewhere: this block again, code by code
    { rd = saved;
      #pragma unroll 1
      for (int k = 0; k < 8; k++)
        { const uint32_t c = wr_symbol(rd, tab, lng);
          w[k >> 2] = (k & 3) ? (w[k >> 2] | (c << (8 * (k & 3)))) : c;
        }
    }
  o0 = w[0]; o1 = w[1];
}

// 8 codes into two output words WITHOUT a bit buffer (DS_POS): per pair of codes one positioned 32-bit read of the window
// -- two words and a v_alignbit; two codes within the tables' index take <= 24 bits --, the second code looked up in the
// first one's word shifted.  The entries' low bytes (32 - length each, <= 31) are summed as they come: eight of them stay
// below 256, so the bits used so far are 32 x codes - (sum & 255).  No refill test, no branch: ~9 instructions per code
// against ~12.  p: the bit (from win[0]'s top bit) at which the block starts; win[-1] must exist.  Returns bit 4 clear if
// some code was longer than the index (the caller decodes the group again, code by code).
__device__ __forceinline__ uint32_t ds_block8_pos(const uint32_t *win, uint32_t &p, const uint16_t *tab, uint32_t &o0, uint32_t &o1)
{ uint32_t w[2] = { 0u, 0u }, zand = 31u, acc = 0u;
  const uint32_t e0 = p + 31u;
  #pragma unroll
  for (int k = 0; k < 8; k += 2)
    { const uint32_t  e  = e0 + 32u * k - (acc & 255u);    // the last of the 32 bits from the pair's first on
      const uint32_t *wp = win + (e >> 5);
      const uint32_t  x  = __builtin_amdgcn_alignbit(wp[-1], wp[0], ~e);
      const uint32_t  ea = tab[x >> (32 - DP_BITS)];
      const uint32_t  y  = __builtin_amdgcn_alignbit(x, 0u, ea);         // x << the first code's bits
      const uint32_t  eb = tab[y >> (32 - DP_BITS)];
      zand &= ea & eb;
      acc  += ea + eb;
      w[k >> 2] = __builtin_amdgcn_perm(ea, w[k >> 2], (k & 3) == 0 ? 0x03020105u : 0x03050100u);
      w[k >> 2] = __builtin_amdgcn_perm(eb, w[k >> 2], (k & 3) == 0 ? 0x03020500u : 0x05020100u);
    }
  p += 256u - (acc & 255u);
  o0 = w[0]; o1 = w[1];
  return zand;
}

// the same for two groups at once: the look-up -> shift -> look-up chains of the two are independent, so a lane has two
// LDS reads in flight where one group alone leaves it waiting for each (DS_DUAL)
__device__ __forceinline__ void ds_block8_dual(winrd &ra, winrd &rb, const uint16_t *tab, const uint32_t *lng,
                                               uint32_t &a0, uint32_t &a1, uint32_t &b0, uint32_t &b1)
{ const winrd sa = ra, sb = rb;
  uint32_t wa[2] = { 0u, 0u }, wb[2] = { 0u, 0u }, zand = 31u;
  #pragma unroll
  for (int k = 0; k < 8; k += 2)
    { wr_fill(ra);
      wr_fill(rb);
      #pragma unroll
      for (int h = 0; h < 2; h++)
        { const uint32_t ea = tab[ra.hi >> (32 - DP_BITS)], eb = tab[rb.hi >> (32 - DP_BITS)];
          const uint32_t sel = ((k + h) & 3) == 0 ? 0x03020105u : ((k + h) & 3) == 1 ? 0x03020500u :
                               ((k + h) & 3) == 2 ? 0x03050100u : 0x05020100u;
          zand &= ea & eb;
          ra.hi = __builtin_amdgcn_alignbit(ra.hi, ra.lo, ea);
          ra.lo = __builtin_amdgcn_alignbit(ra.lo, 0u, ea);
          ra.nb += (int) (ea & 31u) - 32;
          rb.hi = __builtin_amdgcn_alignbit(rb.hi, rb.lo, eb);
          rb.lo = __builtin_amdgcn_alignbit(rb.lo, 0u, eb);
          rb.nb += (int) (eb & 31u) - 32;
          wa[(k + h) >> 2] = __builtin_amdgcn_perm(ea, wa[(k + h) >> 2], sel);
          wb[(k + h) >> 2] = __builtin_amdgcn_perm(eb, wb[(k + h) >> 2], sel);
        }
    }
  if (__any((int) (~zand & 16u)))                          // a long code somewhere: both blocks again, code by code
    { ra = sa; rb = sb;
      #pragma unroll 1
      for (int k = 0; k < 8; k++)
        { const uint32_t c = wr_symbol(ra, tab, lng);
          wa[k >> 2] = (k & 3) ? (wa[k >> 2] | (c << (8 * (k & 3)))) : c;
        }
      #pragma unroll 1
      for (int k = 0; k < 8; k++)
        { const uint32_t c = wr_symbol(rb, tab, lng);
          wb[k >> 2] = (k & 3) ? (wb[k >> 2] | (c << (8 * (k & 3)))) : c;
        }
    }
  a0 = wa[0]; a1 = wa[1]; b0 = wb[0]; b1 = wb[1];
}

struct ds_line { const uint8_t *seg, *at8; uint8_t *out; uint32_t sbytes, G, L; };

__device__ __forceinline__ ds_line ds_line_of(const dec_args &a, const dr_entry &e, uint32_t q, const uint32_t *sub_idx)
{ ds_line ln;
  const uint32_t line = q == 0 ? 0u : q + 1u;              // output line / segment index
  uint64_t at = e.rec + e.hl;
  for (uint32_t k = 0; k < 5; k++)
    if (k < line) at += e.sg[k];
  ln.seg    = a.in + at;
  ln.sbytes = line == 0 ? e.sg[0] : line == 2 ? e.sg[2] : line == 3 ? e.sg[3] : e.sg[4];
  ln.out    = a.out + e.oo + (uint64_t) line * ((uint64_t) e.L + 1u);
  ln.L      = e.L;
  ln.G      = sub_groups(e.L);
  ln.at8    = (const uint8_t *) (sub_idx + e.so + (uint64_t) q * sub_words(e.L));
  return ln;
}

// a round's requests: the index bytes of its DS_STEPS x 64 groups and the first 64 x DS_NPRE words of its window
#define DS_ASK(LN, G0, BASE)                                                                                            \
  { _Pragma("unroll")                                                                                                   \
    for (int k_ = 0; k_ < DS_STEPS; k_++)                                                                               \
      { const uint32_t g_ = (G0) + 64u * k_ + (uint32_t) lane;                                                          \
        draw[k_] = g_ < (LN).G ? (uint32_t) (LN).at8[g_] : 0u;                                                          \
      }                                                                                                                 \
    _Pragma("unroll")                                                                                                   \
    for (int j_ = 0; j_ < DS_NPRE; j_++)                                                                                \
      { const uint64_t b_ = 4ull * (((BASE) >> 5) + 64u * j_ + (uint32_t) lane);                                        \
        pre[j_] = b_ + 4u <= (LN).sbytes ? *(const u32_u *) ((LN).seg + b_) : 0u;                                       \
      }                                                                                                                 \
  }

template <int NK>
__global__ __launch_bounds__(DS_BLOCK, NK <= 2 ? DS_WG_PER_CU : 3)
void k_qv_decode_sub(dec_args a, const uint16_t *g_dec, const uint32_t *g_long, uint32_t *next_task, uint32_t kinds,
                     const uint32_t *sub_idx, const uint64_t *sub_off, uint32_t *status)
{ __shared__ uint16_t s_tab[NK][DP_SIZE];                  // 8 KB each
  __shared__ uint32_t s_long[NK][1 + DX_LONG_MAX];         // 1 KB each
  __shared__ uint32_t s_win[DS_NWAVE][DS_WIN];             // 60 KB
  // tables of the kinds present, in the order of their bits (dp_build_tables for a subset)
  nocode_begin();
  { int slot_of[4], nk = 0;
    for (int q = 0; q < 4; q++) slot_of[q] = ((kinds >> q) & 1u) ? nk++ : -1;
    for (int q = 0; q < 4; q++)
      if (slot_of[q] >= 0)
        for (int k = threadIdx.x; k < 1 + DX_LONG_MAX; k += DS_BLOCK) s_long[slot_of[q]][k] = g_long[q * (1 + DX_LONG_MAX) + k];
    __syncthreads();
    for (int q = 0; q < 4; q++)
      if (slot_of[q] >= 0)
        for (int i_ = threadIdx.x; i_ < DP_SIZE; i_ += DS_BLOCK)
          { const uint32_t i = (uint32_t) i_, e = g_dec[q * DX_DEC_SIZE + (i >> (DP_BITS - DX_DEC_BITS))];
            uint32_t len = e >> 8, sym = e & 0xffu;
            if (len == 0)                                  // longer than 11 bits: exactly DP_BITS long?
              { const uint32_t *lg = s_long[slot_of[q]], cnt = lg[0], pre = i << (16 - DP_BITS);
                for (uint32_t j = 1; j <= cnt; j++)
                  { const uint32_t t = lg[j], l = (t >> 8) & 0xffu;
                    if (l <= DP_BITS && (pre >> (16u - l)) == ((t >> 16) >> (16u - l)))
                      { len = l; sym = t & 0xffu; }
                  }
              }
            s_tab[slot_of[q]][i] = (uint16_t) ((len ? 32u - len : 0u) | (sym << 8));
          }
    __syncthreads();
  }
  uint32_t *const win  = s_win[threadIdx.x >> 6] + 1;      // ([-1]: the positioned reads look one word back)
  const int       lane = lane_id();
  if (lane == 0) win[-1] = 0u;
  const uint32_t  first_kind = (uint32_t) __builtin_ctz(kinds | 16u);

  // A task = the plain lines of one entry (n < 2^31), a round = DS_STEPS x 64 groups of one line.  Every round asks for
  // what the round AFTER it will need -- the next groups of the line, the first of the entry's next line or of the next
  // entry's first line -- before it decodes: the index bytes and (from the bit position the prefix sums have just
  // given) the window's words stay in registers until that round begins.  Without this a round starts with two
  // dependent memory round trips (index, then window) and an entry with three more (ticket, length and offsets).
  uint64_t r, r_end, r1;
  uint32_t tkv = 0, dv = 0, dv_nx = 0;
  uint32_t draw[DS_STEPS], pre[DS_NPRE];
  bool     ready = false, fresh;                           // draw / pre hold the coming round's requests
  DT_FIRST(r, r_end, tkv)
  if (r < a.n) DR_DESC(dv, r)
  for (; r < a.n; r = r1, r_end = fresh ? r1 + dec_ticket : r_end, dv = dv_nx)       // (every wave gets past the end: the counter only grows)
    { dr_entry cur;
      DR_TAKE(cur, dv)
      DT_NEXT(r1, fresh, r, r_end, tkv)
      const bool more = r1 < a.n;
      if (more) DR_DESC(dv_nx, r1)
      int slot = -1;
      #pragma unroll 1
      for (uint32_t q = 0; q < 4; q++)
      { if (!((kinds >> q) & 1u)) continue;
        slot += 1;
      const ds_line   ln     = ds_line_of(a, cur, q, sub_idx);
      const uint32_t  L      = ln.L, G = ln.G, sbytes = ln.sbytes;
      const uint8_t  *seg    = ln.seg;
      uint8_t        *out    = ln.out;
      const uint16_t *tab    = s_tab[slot];
      const uint32_t *lng    = s_long[slot];
      bool            serial = false;                      // SUB_NONE: a symbol without a code in the line (hand-made tables only)
      uint32_t base = 0;                                   // bit at which the round's first group starts

      for (uint32_t g0 = 0; g0 < G; )
        { if (!ready) DS_ASK(ln, g0, base)
          ready = false;
          if (g0 == 0) serial = DR_RL(draw[0], 0) == SUB_NONE;
          // bit counts of up to DS_STEPS x 64 groups, their starts, and how many steps the window holds
          uint32_t d[DS_STEPS], st[DS_STEPS], upto[DS_STEPS];
          uint32_t run = base, steps = 0;
          #pragma unroll
          for (int k = 0; k < DS_STEPS; k++)
            { const uint32_t g = g0 + 64u * k + (uint32_t) lane;
              const uint32_t valid = g < G ? (L - 16u * g < 16u ? L - 16u * g : 16u) : 0u;
              d[k] = valid ? draw[k] + valid : 0u;
              if (serial) d[k] = valid && lane == 0 && k == 0 ? 16u * valid : 0u;      // (an upper bound: one group at a time)
            }
          #pragma unroll
          for (int k = 0; k < DS_STEPS; k++)
            { const uint32_t incl = wave_incl_scan(d[k]);
              st[k]   = run + incl - d[k];
              run    += wave_total(incl);
              upto[k] = run;
              if (((upto[k] + 31u) >> 5) + 2u - (base >> 5) <= DS_WIN - 1u && g0 + 64u * k < G && (!serial || k == 0)) steps = k + 1;
            }
          const uint32_t w0 = base >> 5;
          const uint32_t nw = ((upto[steps - 1] + 31u) >> 5) + 2u - w0;
          #pragma unroll
          for (int j = 0; j < DS_NPRE; j++)
            win[64 * j + lane] = pre[j];
          for (uint32_t i = 64u * DS_NPRE + (uint32_t) lane; i < nw; i += 64)
            { const uint64_t byte = 4ull * (w0 + i);
              win[i] = byte + 4u <= sbytes ? *(const u32_u *) (seg + byte) : 0u;     // (segments are whole words, QV.c:436-442)
            }
          // the round after this one
          if (!serial)
            { if (g0 + 64u * steps < G)
                { DS_ASK(ln, g0 + 64u * steps, upto[steps - 1])
                  ready = true;
                }
              else
                { uint32_t q2 = q + 1u;
                  while (q2 < 4u && !((kinds >> q2) & 1u)) q2 += 1u;
                  if (q2 < 4u)
                    { const ds_line l2 = ds_line_of(a, cur, q2, sub_idx);
                      DS_ASK(l2, 0u, 0u)
                      ready = true;
                    }
                  else if (more)
                    { dr_entry nx;
                      DR_TAKE(nx, dv_nx)
                      if (nx.L)
                        { const ds_line l2 = ds_line_of(a, nx, first_kind, sub_idx);
                          DS_ASK(l2, 0u, 0u)
                          ready = true;
                        }
                    }
                }
            }
          wave_sync();
          uint32_t used = 0;                               // serial mode: bits the group really took
          int kdone = 0;
#if DS_DUAL
          // pairs of steps whose 128 groups are all whole (every round but a line's last): two groups per lane side by side
          #pragma unroll
          for (int k = 0; k + 1 < DS_STEPS; k += 2)
            if (kdone == k && (uint32_t) k + 2u <= steps && !serial && 16u * (g0 + 64u * (k + 2)) <= L)
              { const uint32_t ga = g0 + 64u * k + (uint32_t) lane, gb = ga + 64u;
                winrd ra, rb;
                ra.win = win; rb.win = win;
                { const uint32_t sb = st[k] - 32u * w0, off = sb & 31u;
                  ra.wi = (sb >> 5) + 1u;
                  ra.hi = win[sb >> 5] << off; ra.lo = 0u; ra.nb = 32 - (int) off;
                }
                { const uint32_t sb = st[k + 1] - 32u * w0, off = sb & 31u;
                  rb.wi = (sb >> 5) + 1u;
                  rb.hi = win[sb >> 5] << off; rb.lo = 0u; rb.nb = 32 - (int) off;
                }
                uint32_t x0, x1, x2, x3, y0, y1, y2, y3;
                ds_block8_dual(ra, rb, tab, lng, x0, x1, y0, y1);
                ds_block8_dual(ra, rb, tab, lng, x2, x3, y2, y3);
                const u32x4 va = { x0, x1, x2, x3 }, vb = { y0, y1, y2, y3 };
                *(u32x4_u *) (out + 16ull * ga) = va;
                *(u32x4_u *) (out + 16ull * gb) = vb;
                kdone = k + 2;
              }
#endif
          #pragma unroll
          for (int k = 0; k < DS_STEPS; k++)
            if ((uint32_t) k < steps && k >= kdone)
              { const uint32_t g = g0 + 64u * k + (uint32_t) lane;
                const uint32_t valid = g < G && (!serial || lane == 0) ? (L - 16u * g < 16u ? L - 16u * g : 16u) : 0u;
                if (valid)
                  { winrd rd;
                    rd.win = win;
                    { const uint32_t sb = st[k] - 32u * w0, off = sb & 31u;
                      rd.wi = (sb >> 5) + 1u;
                      rd.hi = win[sb >> 5] << off; rd.lo = 0u; rd.nb = 32 - (int) off;
                    }
                    uint8_t *o = out + 16ull * g;
                    if (valid == 16u)
                      { uint32_t x0, x1, x2, x3;
#if DS_POS
                        uint32_t pp = st[k] - 32u * w0;
                        uint32_t z  = ds_block8_pos(win, pp, tab, x0, x1);
                        z &= ds_block8_pos(win, pp, tab, x2, x3);
                        if (__any((int) (~z & 16u)))       // a long code somewhere in the wave's groups: this group again, code by code
#endif
                        { ds_block8(rd, tab, lng, x0, x1);
                          ds_block8(rd, tab, lng, x2, x3);
                        }
                        const u32x4 v = { x0, x1, x2, x3 };
                        *(u32x4_u *) o = v;
                      }
                    else                                   // the ragged end of the line (one lane of the wave, but the whole wave's time:
                      {                                    //  sixteen codes at once here too, those past the end thrown away)
#if DS_POS
                        uint32_t x0, x1, x2, x3, pp = st[k] - 32u * w0;
                        uint32_t z = ds_block8_pos(win, pp, tab, x0, x1);
                        z &= ds_block8_pos(win, pp, tab, x2, x3);
                        if (!serial && (z & 16u))          // (a code beyond the index, real or among those thrown away: code by code)
                          { if (valid >= 4u)  *(u32_u *) o        = x0;
                            if (valid >= 8u)  *(u32_u *) (o + 4)  = x1;
                            if (valid >= 12u) *(u32_u *) (o + 8)  = x2;
                            const uint32_t w = valid >= 12u ? x3 : valid >= 8u ? x2 : valid >= 4u ? x1 : x0;
                            for (uint32_t j = 0; j < (valid & 3u); j++)
                              o[(valid & ~3u) + j] = (uint8_t) (w >> (8u * j));
                          }
                        else
#endif
                        for (uint32_t j = 0; j < valid; j++)
                          o[j] = (uint8_t) wr_symbol(rd, tab, lng);
                      }
                    used = 32u * (rd.wi - 1u - ((st[k] - 32u * w0) >> 5)) + (32u - ((st[k] - 32u * w0) & 31u)) - (uint32_t) rd.nb;
                  }
              }
          wave_sync();
          if (serial) { base += uniform(used); g0 += 1; }
          else        { base  = upto[steps - 1]; g0 += 64u * steps; }
        }
      if (lane == 0)
        out[L] = '\n';
      }
    }
  nocode_end(status);
}
#undef DS_ASK

// ---------------------------------------------------------------------------------------------
//  plain lines of a bare stream, with what the device walk noted on its way: k_qv_decode_sync
// ---------------------------------------------------------------------------------------------
// The walk of a bare stream (dx_qv_walk.hip) passes a plain line several codes a look-up and cannot say where every 16th symbol
// is (k_qv_decode_sub's index) without walking it code by code; what it can note for nothing is where the look-up BEGAN that
// reached every 64th symbol: a word per 64 symbols, the bits passed | the symbols from there to the 64th << 28 (at most 12: a
// look-up's codes), word g of the line's share of the index (word 0: 0, or DXL_SYNC_NONE: a line without -- k_qv_decode_plain
// takes it).  Here a wavefront per (entry, plain line), a LANE per 64 symbols: the words the round's lanes span go into the wave's
// LDS window (coalesced), a lane passes the few codes in front of its 64 symbols one by one and then decodes the 64 by positioned
// reads (ds_block8_pos: two codes a read, no bit buffer) into 16 registers = four 16-byte stores, 64 consecutive bytes a lane, 4 KiB
// a wave.  A round takes as many lanes as its window holds (a lane's codes take 28 bytes at the bench's 3.5 bits a code; the
// window holds 5 KB).  Decode QV.c:510-599.
#define DY_BLOCK 768
#define DY_NWAVE (DY_BLOCK / 64)
#define DY_WIN   1280                                       // words per wave: 5 KB
#ifndef DY_PER
#define DY_PER   3u                                         // stretches of 64 symbols a lane takes in a round
#endif
#define DY_SLACK 288u                                       // bits a lane may look at behind the next lane's start: <= 12 codes of 16 bits in front
                                                            // of ITS 64 symbols, the positioned reads' 32 bits, and the last pair's second look
template <int NK>
__global__ __launch_bounds__(DY_BLOCK, NK <= 2 ? 6 : 3)
void k_qv_decode_sync(dec_args a, const uint16_t *g_dec, const uint32_t *g_long, uint32_t *next_task, uint32_t kinds,
                      const uint32_t *sub_idx, const uint64_t *sub_off, uint32_t *status)
{ __shared__ uint16_t s_tab[NK][DP_SIZE];                  // 8 KB each
  __shared__ uint32_t s_long[NK][1 + DX_LONG_MAX];         // 1 KB each
  __shared__ __attribute__((aligned(16))) uint32_t s_win[DY_NWAVE][DY_WIN];   // 60 KB
  nocode_begin();
  { int slot_of[4], nk = 0;                                // tables of the kinds present, in the order of their bits (as k_qv_decode_sub)
    for (int q = 0; q < 4; q++) slot_of[q] = ((kinds >> q) & 1u) ? nk++ : -1;
    for (int q = 0; q < 4; q++)
      if (slot_of[q] >= 0)
        for (int k = threadIdx.x; k < 1 + DX_LONG_MAX; k += DY_BLOCK) s_long[slot_of[q]][k] = g_long[q * (1 + DX_LONG_MAX) + k];
    __syncthreads();
    for (int q = 0; q < 4; q++)
      if (slot_of[q] >= 0)
        for (int i_ = threadIdx.x; i_ < DP_SIZE; i_ += DY_BLOCK)
          { const uint32_t i = (uint32_t) i_, e = g_dec[q * DX_DEC_SIZE + (i >> (DP_BITS - DX_DEC_BITS))];
            uint32_t len = e >> 8, sym = e & 0xffu;
            if (len == 0)
              { const uint32_t *lg = s_long[slot_of[q]], cnt = lg[0], pre = i << (16 - DP_BITS);
                for (uint32_t j = 1; j <= cnt; j++)
                  { const uint32_t t = lg[j], l = (t >> 8) & 0xffu;
                    if (l <= DP_BITS && (pre >> (16u - l)) == ((t >> 16) >> (16u - l)))
                      { len = l; sym = t & 0xffu; }
                  }
              }
            s_tab[slot_of[q]][i] = (uint16_t) ((len ? 32u - len : 0u) | (sym << 8));
          }
    __syncthreads();
  }
  uint32_t *const win  = s_win[threadIdx.x >> 6] + 1;      // ([-1]: the positioned reads look one word back)
  const int       lane = lane_id();
  if (lane == 0) win[-1] = 0u;

  uint64_t r, r_end, r1;
  uint32_t tkv = 0, dv = 0, dv_nx = 0;
  bool     fresh;
  DT_FIRST(r, r_end, tkv)
  if (r < a.n) DR_DESC(dv, r)
  for (; r < a.n; r = r1, r_end = fresh ? r1 + dec_ticket : r_end, dv = dv_nx)       // (every wave gets past the end: the counter only grows)
    { dr_entry cur;
      DR_TAKE(cur, dv)
      DT_NEXT(r1, fresh, r, r_end, tkv)
      if (r1 < a.n) DR_DESC(dv_nx, r1)
      int slot = -1;
      #pragma unroll 1
      for (uint32_t q = 0; q < 4; q++)
      { if (!((kinds >> q) & 1u)) continue;
        slot += 1;
        const ds_line   ln     = ds_line_of(a, cur, q, sub_idx);
        const uint32_t  L      = ln.L, SG = (L + 63u) >> 6, sbytes = ln.sbytes;
        const uint8_t  *seg    = ln.seg;
        uint8_t        *out    = ln.out;
        const uint16_t *tab    = s_tab[slot];
        const uint32_t *lng    = s_long[slot];
        const uint32_t *sy     = (const uint32_t *) (const void *) ln.at8;         // the line's share: a word per 64 symbols
        if (SG && uniform(sy[0]) == DXL_SYNC_NONE) continue;                       // a line without: k_qv_decode_plain's

        for (uint32_t g0 = 0; g0 < SG; )
          { const uint32_t g  = g0 + (uint32_t) lane;
            const uint32_t w  = g < SG && g ? sy[g] : 0u;
            const uint32_t wn = g + 1u < SG ? sy[g + 1u] : 0u;
            const uint32_t T  = w & 0x0fffffffu, dl = w >> 28;
            const uint32_t w0 = uniform(T) >> 5;                                   // the window's first word: lane 0's
            // the bits a lane may look at end DY_SLACK behind the next lane's start (the line's last lane: behind the segment)
            const uint32_t need = g + 1u < SG ? (wn & 0x0fffffffu) + DY_SLACK : 8u * sbytes + 64u;
            const bool     fits = g < SG && ((need + 31u) >> 5) + 1u - w0 <= DY_WIN - 2u;
            const uint64_t miss = __ballot(!fits);
            const uint32_t cntl = miss ? (uint32_t) __ffsll((unsigned long long) miss) - 1u : 64u;      // (>= 1: one lane's bits are < 50 words)
            if (cntl == 0u)                                // words that are no index of this stream (dx_qv_use_dindex takes the caller's arrays):
              { if (lane == 0) atomicOr(status, 4u);       //   a corrupt stream, and never a round that advances by nothing
                break;
              }
            const bool     mine = (uint32_t) lane < cntl;
            uint32_t nw = mine ? ((need + 31u) >> 5) + 1u - w0 : 0u;
            nw = wave_total(wave_incl_max(nw));
            for (uint32_t i = (uint32_t) lane; i < nw; i += 64u)
              { const uint64_t byte = 4ull * (w0 + i);
                win[i] = byte + 4u <= sbytes ? *(const u32_u *) (seg + byte) : 0u;   // (segments are whole words, QV.c:436-442)
              }
            wave_sync();
            u32x4 v0 = { 0u, 0u, 0u, 0u }, v1 = v0, v2 = v0, v3 = v0;
            bool  whole = false;                           // the lane holds a whole stretch's 64 letters in v0 .. v3
            if (mine)
              { const uint32_t p0 = T - 32u * w0;
                uint32_t p = p0, x[16], z = 31u;
                // the codes in front of the lane's 64 symbols (the look-up that reached them began here)
                for (uint32_t k = 0; k < dl; k++)
                  { const uint32_t  e  = p + 31u;
                    const uint32_t *wp = win + (e >> 5);
                    const uint32_t  ea = tab[__builtin_amdgcn_alignbit(wp[-1], wp[0], ~e) >> (32 - DP_BITS)];
                    z &= ea;
                    p += 32u - (ea & 31u);
                  }
                #pragma unroll
                for (int b = 0; b < 8; b++)
                  z &= ds_block8_pos(win, p, tab, x[2 * b], x[2 * b + 1]);
                if (!(z & 16u))                            // a code beyond the tables' index somewhere: the lane's symbols again, code by code
                  { winrd rd;
                    rd.win = win; rd.wi = (p0 >> 5) + 1u;
                    rd.hi = win[p0 >> 5] << (p0 & 31u); rd.lo = 0u; rd.nb = 32 - (int) (p0 & 31u);
                    for (uint32_t k = 0; k < dl; k++) (void) wr_symbol(rd, tab, lng);
                    #pragma unroll 1
                    for (int k = 0; k < 64; k++)
                      { const uint32_t c = wr_symbol(rd, tab, lng);
                        #pragma unroll
                        for (int i = 0; i < 16; i++)
                          if (i == (k >> 2)) x[i] = (k & 3) ? (x[i] | (c << (8 * (k & 3)))) : c;
                      }
                  }
                uint8_t *o = out + 64ull * g;
                const uint32_t valid = L - 64u * g < 64u ? L - 64u * g : 64u;
                if (valid == 64u)
                  { v0 = u32x4{ x[0], x[1], x[2], x[3] };   v1 = u32x4{ x[4], x[5], x[6], x[7] };
                    v2 = u32x4{ x[8], x[9], x[10], x[11] }; v3 = u32x4{ x[12], x[13], x[14], x[15] };
                    whole = true;
                  }
                else                                       // the ragged end of the line (its codes past the end were zeros or the next segment's: thrown away)
                  {
                    #pragma unroll
                    for (int i = 0; i < 16; i++)
                      { if (4u * (uint32_t) i + 4u <= valid) *(u32_u *) (o + 4 * i) = x[i];
                        else
                          for (uint32_t j = 4u * (uint32_t) i; j < valid; j++) o[j] = (uint8_t) (x[i] >> (8u * (j & 3u)));
                      }
                  }
              }
            // A lane's 64 letters are 64 consecutive bytes of the text: stored as they are, a wave's store instruction is 64 requests of
            // 16 bytes 64 bytes apart, four times the requests of a store whose lanes write side by side -- the kernel then waits
            // on its stores (13.2 ms; k_qv_decode_sub, whose lanes interleave 16 letters each, 9).  So the round's 4 KiB change
            // hands in the window, which the round no longer needs: lane l leaves its 16-byte piece c at row l, place c ^ (l / 2 & 3)
            // (eight lanes in a row fill the LDS's eight 16-byte bank groups: no conflict, writing or reading), and takes piece
            // l & 3 of row 16 k + l / 4 for the k-th KiB of the round.
            wave_sync();
            const uint64_t wholes = __ballot(whole);
            if (wholes)
              { u32x4 *stage = (u32x4 *) (void *) (win + 3);                     // (16-byte aligned: win is s_win + 1)
                const uint32_t sw = ((uint32_t) lane >> 1) & 3u;
                if (whole)
                  { stage[4u * (uint32_t) lane + (0u ^ sw)] = v0; stage[4u * (uint32_t) lane + (1u ^ sw)] = v1;
                    stage[4u * (uint32_t) lane + (2u ^ sw)] = v2; stage[4u * (uint32_t) lane + (3u ^ sw)] = v3;
                  }
                wave_sync();
                #pragma unroll
                for (int k = 0; k < 4; k++)
                  { const uint32_t row = 16u * (uint32_t) k + ((uint32_t) lane >> 2), c = (uint32_t) lane & 3u;
                    if ((wholes >> row) & 1ull)
                      { const u32x4 v = stage[4u * row + (c ^ ((row >> 1) & 3u))];
                        *(u32x4_u *) (out + 64ull * (g0 + row) + 16u * c) = v;
                      }
                  }
                wave_sync();
              }
            g0 += cntl;
          }
        if (lane == 0)
          out[L] = '\n';
      }
    }
  nocode_end(status);
}

// ---------------------------------------------------------------------------------------------
//  run-coded lines with the encoder's group index: k_qv_decode_runs
// ---------------------------------------------------------------------------------------------
// A wavefront per (entry, run-coded line).  The index holds, for every group of <= 8 consecutive (run, symbol)
// tokens -- what a lane of the encoder coded in a pass -- the bits they take and the positions they cover.  The
// wave goes through the passes of 512 tokens the way the encoder did: two prefix sums turn the groups' bits and
// spans into where each lane starts reading and writing, the pass's words are staged in the LDS window, every lane
// decodes its tokens -- run code (+ 16-bit literal), symbol code -- into registers (one past the symbol's place |
// symbol << 16), and the pass's piece of the line then goes through a staging buffer RUN_STRETCH positions at a
// time: run characters everywhere, the symbols of the tokens that fall into it at their places, out in 16-byte
// stores; for the deletion line the same stretch of the TAG line follows.  All lanes hold the same number of tokens
// (the line's last pass aside), so the wave runs in step.  What is left of the line behind its last token is run
// characters.  Everything a pass, a line and an entry begin with is requested ahead (see the kernel).
#ifndef DR_BLOCK
#define DR_BLOCK 1024                                      // 16 waves: tables 36 KB + 16 x (1.75 + 5) KB = 144 KB
#endif
#ifndef DR_POS
#define DR_POS   RUN_STRETCH                               // positions of the staging buffer
#endif
#ifndef DR_WAVES
#define DR_WAVES 4                                         // waves per SIMD the kernel is compiled for (two 10-wave workgroups with half the
                                                           // staging buffer each -- 5 per SIMD, 96 registers -- measured 19.5 ms against 14.2)
#endif
#ifndef DR_WG_PER_CU
#define DR_WG_PER_CU 1
#endif
#define DR_NWAVE (DR_BLOCK / 64)
#define DR_WIN   448                                       // words per wave: a pass's bits (<= RUN_PASSBITS, the encoder saw to it) + slack
#define DR_MAXPIECE 65534u                                 // positions of a pass whose places (+ 1) still fit 16 bits beside the symbol
#ifndef DR_SKIP
#define DR_SKIP  0                                         // timing experiments: 1 no tag line, 2 no tokens, 4 the line's piece stays in LDS
#endif
#ifndef DR_FAST
#define DR_FAST  1                                         // the sound pass without a bit buffer (see the kernel)
#endif
#define DR_STRETCH (DR_POS / 4)                       // words per wave for a pass's piece of the line (the encoder saw to it that it fits)

// where a run-coded line of an entry is, and whether its share of the index is one the entry can have
struct dr_line { const uint8_t *seg; const uint32_t *g16; uint32_t sbytes, cnt; bool ok; };

__device__ __forceinline__ dr_line dr_line_of(const dec_args &a, const dr_entry &e, const uint32_t head[3], uint32_t q, const uint32_t *sub_idx)
{ dr_line ln;
  const uint32_t L = e.L;
  ln.cnt = head[q == 0 ? 0 : 1];                           // tokens (RUN_NONE: not indexed)
  const uint64_t share = e.so1 - e.so;                     // words of this entry in the index
  const uint32_t dpass = head[2] & ~DXL_RUN_EIGHTS;          // (the flag: how the last pass's tokens are dealt, k_qv_decode_runs)
  const uint64_t need  = (uint64_t) run_base(L) + 3u + 64ull * ((q == 0 ? 0u : dpass) + run_passes(ln.cnt));
  ln.ok     = ln.cnt != RUN_NONE && ln.cnt <= dxl_tok_limit(L) && need <= share;
  ln.seg    = a.in + e.rec + e.hl + (q == 0 ? 0ull : (uint64_t) e.sg[0] + e.sg[1] + e.sg[2] + e.sg[3]);
  ln.sbytes = q == 0 ? e.sg[0] : e.sg[4];
  ln.g16    = sub_idx + e.so + run_base(L) + 3u + (q == 0 ? 0u : 64u * dpass);        // three header words, then the groups
  return ln;
}

template <int NK>                                          // run-coded kinds in the launch: 1 or 2 (del, sub)
__global__ __launch_bounds__(DR_BLOCK, DR_WAVES)
void k_qv_decode_runs(dec_args a, const uint16_t *g_dec, const uint32_t *g_long, uint32_t *status, uint32_t *next_task,
                      uint32_t kinds, const uint32_t *sub_idx, const uint64_t *sub_off)
{ __shared__ uint16_t s_tab[2 * NK][DP_SIZE];              // per kind: symbols, runs (8 KB each)
  __shared__ uint32_t s_long[2 * NK][1 + DX_LONG_MAX];
  __shared__ uint32_t s_win[DR_NWAVE][DR_WIN];             // 52 KB; [0] is a lead word (the positioned reads look one word back)
  __shared__ __attribute__((aligned(16))) uint32_t s_str[DR_NWAVE][DR_STRETCH];       // (see DR_BLOCK)
  nocode_begin();
  { int nk = 0;
    for (int q = 0; q < 4; q++)
      if ((kinds >> q) & 1u)
        { const int src[2] = { q, q == 0 ? DX_DRUN : DX_SRUN };
          for (int h = 0; h < 2; h++)
            for (int k = threadIdx.x; k < 1 + DX_LONG_MAX; k += DR_BLOCK) s_long[2 * nk + h][k] = g_long[src[h] * (1 + DX_LONG_MAX) + k];
          nk += 1;
        }
    __syncthreads();
    nk = 0;
    for (int q = 0; q < 4; q++)
      if ((kinds >> q) & 1u)
        { const int src[2] = { q, q == 0 ? DX_DRUN : DX_SRUN };
          for (int h = 0; h < 2; h++)
            for (int i_ = threadIdx.x; i_ < DP_SIZE; i_ += DR_BLOCK)
              { const uint32_t i = (uint32_t) i_, e = g_dec[src[h] * DX_DEC_SIZE + (i >> (DP_BITS - DX_DEC_BITS))];
                uint32_t len = e >> 8, sym = e & 0xffu;
                if (len == 0)
                  { const uint32_t *lg = s_long[2 * nk + h], cnt = lg[0], pre = i << (16 - DP_BITS);
                    for (uint32_t j = 1; j <= cnt; j++)
                      { const uint32_t t = lg[j], l = (t >> 8) & 0xffu;
                        if (l <= DP_BITS && (pre >> (16u - l)) == ((t >> 16) >> (16u - l)))
                          { len = l; sym = t & 0xffu; }
                      }
                  }
                s_tab[2 * nk + h][i] = (uint16_t) ((len ? 32u - len : 0u) | (sym << 8));
              }
          nk += 1;
        }
    __syncthreads();
  }
  uint32_t *const win  = s_win[threadIdx.x >> 6] + 1, *const stretch = s_str[threadIdx.x >> 6];
  const int       lane = lane_id();
  if (lane == 0) win[-1] = 0u;

  // A task = the run-coded lines of one entry.  What says where an entry's lines are -- its length, record, segment
  // sizes, share of the index: 18 words from six arrays -- is ONE load whose lanes fetch a word each, requested while the
  // entry before it is decoded and taken apart with v_readlane when its turn comes; so are the three header words of its
  // share (asked for between the two lines) and the ticket (drawn two entries ahead).  Otherwise an entry begins with five
  // dependent memory round trips per line, at four waves per SIMD.
#define DR_HEADV(V, E) { V = 0u; if (lane < 3) V = (sub_idx + E.so + run_base(E.L))[lane]; }
  uint64_t r, r_end, r1;
  uint32_t tkv = 0, dv = 0, dv_nx = 0, hv = 0, hv_nx = 0;
  uint32_t gw_nx = 0, pre0 = 0, pre1 = 0;                  // a pass's group word and the first 128 words of its window ...
  bool     ready = false, fresh;                           // ... requested for the next line that has passes (by the line before it)
  DT_FIRST(r, r_end, tkv)
  if (r < a.n)
    { dr_entry e0;
      DR_DESC(dv, r)
      DR_TAKE(e0, dv)
      DR_HEADV(hv, e0)
    }
  for (; r < a.n; r = r1, r_end = fresh ? r1 + dec_ticket : r_end, dv = dv_nx, hv = hv_nx)       // (every wave gets past the end: the counter only grows)
    { dr_entry cur, nx = {};
      uint32_t head[3];
      DR_TAKE(cur, dv)
      head[0] = DR_RL(hv, 0); head[1] = DR_RL(hv, 1); head[2] = DR_RL(hv, 2);
      DT_NEXT(r1, fresh, r, r_end, tkv)
      const bool more = r1 < a.n;
      bool asked = false;                                  // the next entry's header words are on their way
      if (more) DR_DESC(dv_nx, r1)
      int slot = -1;
      #pragma unroll 1
      for (uint32_t q = 0; q < 4; q += 3)                  // del (0), sub (3)
      { if (!((kinds >> q) & 1u)) continue;
        slot += 1;
        const uint32_t  L   = cur.L;
        const dr_line   ln  = dr_line_of(a, cur, head, q, sub_idx);
        const uint32_t  cnt = ln.cnt;
        if (cnt == RUN_NONE) continue;                     // not indexed: k_qv_decode takes this line
        if (!ln.ok)                                        // not an index this entry can have: never follow it
          { if (lane == 0) atomicOr(status, 4u);
            continue;
          }
        const int       line = q == 0 ? 0 : 4;
        const uint32_t *sg   = cur.sg;
        const uint8_t  *seg    = ln.seg;
        const uint32_t  sbytes = ln.sbytes;
        uint8_t        *out    = a.out + cur.oo + (uint64_t) line * ((uint64_t) L + 1u);
        const uint32_t  rc     = (uint32_t) (q == 0 ? a.delChar : a.subChar);
        const uint16_t *stab   = s_tab[2 * slot], *rtab = s_tab[2 * slot + 1];
        const uint32_t *slng   = s_long[2 * slot], *rlng = s_long[2 * slot + 1];
        const uint32_t *g16    = ln.g16;

        uint32_t base_bit = 0, base_pos = 0, bad = 0;
        const uint32_t pat = rc * 0x01010101u;
        // the deletion line's tokens also say where the tag line has letters (Unpack_Tag + Lower_Read, QV.c:1437-1461):
        // token i's tag is the i-th 2-bit code of the tag segment; every other position is 'n'
        const bool     tags  = q == 0;
        const uint8_t *tsrc  = seg + sbytes;               // (the tag segment follows the deletion segment)
        const uint32_t tbytes = sg[1];
        uint8_t       *tout  = out + (uint64_t) L + 1u;
        const uint32_t fold  = a.upper ? 32u : 0u, tpat = ('n' - fold) * 0x01010101u;
        // requested a pass ahead, so that a pass does not begin with two memory round trips in a row: its group word and
        // the first 128 words of its window (a pass of the bench's lines takes ~115)
#define DR_WORD(W) (4ull * (W) + 4u <= sbytes ? *(const u32_u *) (seg + 4ull * (W)) : 0u)
        if (!ready && cnt)
          { gw_nx = g16[lane];
            pre0  = DR_WORD((uint64_t) lane);
            pre1  = DR_WORD(64ull + (uint64_t) lane);
          }
        if (cnt) ready = false;                            // (this line's first pass takes them)
        for (uint32_t k0 = 0; k0 < ((DR_SKIP & 32) ? 0u : cnt); k0 += 512u)
          { const uint32_t m     = cnt - k0 < 512u ? cnt - k0 : 512u;
            const uint32_t T     = (head[2] & DXL_RUN_EIGHTS) ? 8u : (m + 63u) >> 6;   // tokens per lane in this pass (as the encoder cut
                                                           // them; the device walk's index: 8 in the last pass as in every other)
            const uint32_t first = (uint32_t) lane * T;
            const uint32_t c     = first < m ? (m - first < T ? m - first : T) : 0u;
            const uint32_t gw    = gw_nx;
            uint32_t tg0 = 0, tg1 = 0, tg2 = 0;            // the tag bytes of this lane's tokens: wanted late, asked for now
            if (tags && c)
              { const uint32_t b0 = (k0 + first) >> 2;
                tg0 = b0      < tbytes ? (uint32_t) tsrc[b0]      : 0u;
                tg1 = b0 + 1u < tbytes ? (uint32_t) tsrc[b0 + 1u] : 0u;
                tg2 = b0 + 2u < tbytes ? (uint32_t) tsrc[b0 + 2u] : 0u;
              }
            const uint32_t bits  = c ? gw & 0xffffu : 0u, span = c ? gw >> 16 : 0u;
            const uint32_t ib = wave_incl_scan(bits), ip = wave_incl_scan(span);
            const uint32_t sb = base_bit + ib - bits;
            uint32_t       pos = ip - span;                // relative to the pass's first position
            const uint32_t tb = wave_total(ib);
            uint32_t       tp = wave_total(ip);
            if (base_pos + tp > L) { tp = L - base_pos; bad = 1; }        // corrupt index: never past the line
            const bool staged = tp <= DR_MAXPIECE;         // the pass's piece of the line goes through the staging buffer, RUN_STRETCH
                                                           // positions at a time (else, runs of thousands: straight to memory, byte by byte)
            const uint32_t w0 = base_bit >> 5;
            uint32_t nw = ((base_bit + tb + 31u) >> 5) + 2u - w0;
            if (nw > DR_WIN - 1u) { nw = DR_WIN - 1u; bad = 1; }            // (cannot happen with a sound index)
            if (!(DR_SKIP & 16))
            { win[lane] = pre0; win[64 + lane] = pre1;
            for (uint32_t i = 128u + (uint32_t) lane; i < nw; i += 64)
              win[i] = DR_WORD((uint64_t) w0 + i);
            }
            if (more && !asked)                            // (the next entry's numbers are here by now: a pass has gone by)
              { DR_TAKE(nx, dv_nx)
                DR_HEADV(hv_nx, nx)
                asked = true;
              }
            if (k0 + 512u < cnt)
              { const uint64_t w0n = (base_bit + tb) >> 5;
                gw_nx = g16[((k0 + 512u) >> 3) + (uint32_t) lane];
                pre0  = DR_WORD(w0n + (uint64_t) lane);
                pre1  = DR_WORD(w0n + 64ull + (uint64_t) lane);
              }
            else                                           // the line's last pass: the first pass of the line that comes next
              { dr_line nl = { NULL, NULL, 0u, 0u, false };
                if (q == 0 && ((kinds >> 3) & 1u)) nl = dr_line_of(a, cur, head, 3u, sub_idx);
                if (!(nl.ok && nl.cnt) && more)
                  { uint32_t hn[3];
                    hn[0] = DR_RL(hv_nx, 0); hn[1] = DR_RL(hv_nx, 1); hn[2] = DR_RL(hv_nx, 2);
                    if (kinds & 1u) nl = dr_line_of(a, nx, hn, 0u, sub_idx);
                    if (!(nl.ok && nl.cnt) && ((kinds >> 3) & 1u)) nl = dr_line_of(a, nx, hn, 3u, sub_idx);
                  }
                if (nl.ok && nl.cnt)
                  { gw_nx = nl.g16[lane];
                    pre0  = 4ull * (uint64_t) lane + 4u <= nl.sbytes ? *(const u32_u *) (nl.seg + 4ull * (uint64_t) lane) : 0u;
                    pre1  = 4ull * (64ull + (uint64_t) lane) + 4u <= nl.sbytes ? *(const u32_u *) (nl.seg + 4ull * (64ull + (uint64_t) lane)) : 0u;
                    ready = true;
                  }
              }
            if (!staged)                                   // the piece as run characters in memory, and there before the symbols go over them
              { uint8_t *o = out + base_pos;
                for (uint32_t k = 16u * (uint32_t) lane; k < tp; k += 1024u)
                  if (k + 16u <= tp) { const u32x4 v = { pat, pat, pat, pat }; *(u32x4_u *) (o + k) = v; }
                  else for (uint32_t j = k; j < tp; j++) o[j] = (uint8_t) rc;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_s_waitcnt(0);
              }
            else if (!(DR_SKIP & 8))                       // the first stretch of the piece as run characters: under way while the tokens are decoded
              { const uint32_t wl0 = tp < DR_POS ? tp : DR_POS;
                const u32x4 v = { pat, pat, pat, pat };
                for (uint32_t i = (uint32_t) lane; i < (wl0 + 15u) >> 4; i += 64)
                  ((u32x4 *) stretch)[i] = v;
              }
            wave_sync();
            // Tokens first, bytes afterwards: every lane works out where its <= 8 symbols go (one past the place | symbol << 16)
            // and only then are they put into the piece, RUN_STRETCH positions at a time.
            // The pass as it nearly always is (no code longer than the tables' index, no literal run): per token one
            // positioned 32-bit read of the window -- a token's two codes take <= 24 bits --, two look-ups, no bit buffer and
            // no branch.  Anything else: code by code.
            uint32_t posk[8];
            bool fast = false;
#if DR_FAST
            if (staged && tp && !(DR_SKIP & 2))
              { uint32_t p = sb - 32u * w0, at1 = pos, zand = 31u, rmax = 0u;
                #pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                  { posk[k] = 0u;
                    if (k < c)
                    { const uint32_t  e  = p + 31u;                   // the last of the 32 bits from p on
                      const uint32_t *wp = win + (e >> 5);
                      const uint32_t  x  = __builtin_amdgcn_alignbit(wp[-1], wp[0], ~e);
                      const uint32_t  er = rtab[x >> (32 - DP_BITS)];
                      const uint32_t  y  = __builtin_amdgcn_alignbit(x, 0u, er);          // x << the run code's bits
                      const uint32_t  es = stab[y >> (32 - DP_BITS)];
                      zand &= er & es;                                 // bit 4: set in every entry of a code within the index
                      rmax  = rmax > (er >> 8) ? rmax : er >> 8;
                      p    += 64u - (er & 31u) - (es & 31u);
                      at1  += (er >> 8) + 1u;                          // one past the symbol's place
                      posk[k] = __builtin_amdgcn_perm(es, at1, 0x0c050100u);               // place + 1 | symbol << 16
                    }
                  }
                fast = !__any((int) ((~zand & 16u) | (uint32_t) (rmax >= 255u) | (uint32_t) (c && at1 > tp)));
              }
#endif
            if (c && !fast && !(DR_SKIP & 2))
              { winrd rd;
                rd.win = win;
                { const uint32_t s0 = sb - 32u * w0, off = s0 & 31u;
                  rd.wi = (s0 >> 5) + 1u;
                  rd.hi = win[s0 >> 5] << off; rd.lo = 0u; rd.nb = 32 - (int) off;
                }
                #pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                  { posk[k] = 0u;
                    if (k < c)
                    { uint32_t run = wr_symbol(rd, rtab, rlng);      // (fills first: >= 32 bits)
                      if (run == 255u)                               // 16-bit literal, QV.c:670-676
                        { wr_fill(rd);
                          run = rd.hi >> 16;
                          rd.hi = __builtin_amdgcn_alignbit(rd.hi, rd.lo, 16u);
                          rd.lo <<= 16;
                          rd.nb -= 16;
                        }
                      pos += run;
                      const uint32_t x = wr_symbol(rd, stab, slng);
                      if (pos >= tp)   { bad = 1; pos = tp ? tp - 1u : 0u; }   // corrupt stream / index: stay inside
                      if (staged) posk[k] = (pos + 1u) | (x << 16);           // (pos < tp <= DR_MAXPIECE: 16 bits do)
                      else
                        { if (tp) out[base_pos + pos] = (uint8_t) x;
                          posk[k] = pos + 1u;
                        }
                      pos += 1;
                    }
                  }
              }
            else if (!fast)
              {
                #pragma unroll
                for (uint32_t k = 0; k < 8; k++) posk[k] = 0u;
              }
            const uint32_t Wtag = ((tg0 << 24) | (tg1 << 16) | (tg2 << 8)) << (2u * ((k0 + first) & 3u));   // this lane's tag codes, first one on top
            if (staged)
              for (uint32_t w = 0; w < tp; w += DR_POS)
                { const uint32_t wl = tp - w < DR_POS ? tp - w : DR_POS;
                  uint8_t *const s8 = (uint8_t *) stretch;
                  if (w)                                       // (the first stretch was filled before the tokens were decoded)
                    { if (!(DR_SKIP & 8))
                        { const u32x4 v = { pat, pat, pat, pat };
                          for (uint32_t i = (uint32_t) lane; i < (wl + 15u) >> 4; i += 64)    // this stretch of the line: run characters
                            ((u32x4 *) stretch)[i] = v;
                        }
                      wave_sync();
                    }
                  #pragma unroll
                  for (uint32_t k = 0; k < 8; k++)
                    { const uint32_t pl = (posk[k] & 0xffffu) - 1u - w;                  // (no token: a huge number)
                      if (k < c && pl < wl) s8[pl] = (uint8_t) (posk[k] >> 16);
                    }
                  wave_sync();
                  if (!(DR_SKIP & 4))
                    { uint8_t *o = out + base_pos + w;         // the stretch leaves in 16-byte pieces, its last bytes one by one
                      for (uint32_t i = (uint32_t) lane; 16u * i < wl; i += 64)
                        if (16u * i + 16u <= wl)
                          { const u32x4 v = ((const u32x4 *) stretch)[i];
                            *(u32x4_u *) (o + 16u * i) = v;
                          }
                        else
                          for (uint32_t j = 16u * i; j < wl; j++) o[j] = s8[j];
                    }
                  wave_sync();
                  if (tags && !(DR_SKIP & 1))                  // the same stretch of the tag line: 'n', letters at the tokens' places
                    { uint8_t *o = tout + base_pos + w;
                      const u32x4 v = { tpat, tpat, tpat, tpat };
                      for (uint32_t i = (uint32_t) lane; i < (wl + 15u) >> 4; i += 64)
                        ((u32x4 *) stretch)[i] = v;
                      wave_sync();
                      uint32_t W = Wtag;
                      #pragma unroll
                      for (uint32_t k = 0; k < 8; k++)
                        { const uint32_t ch = ((0x74676361u >> (8u * (W >> 30))) & 0xffu) - fold;   // "acgt", Lower_Read DB.c:367
                          const uint32_t pl = (posk[k] & 0xffffu) - 1u - w;
                          W <<= 2;
                          if (k < c && pl < wl) s8[pl] = (uint8_t) ch;
                        }
                      wave_sync();
                      for (uint32_t i = (uint32_t) lane; 16u * i < wl; i += 64)
                        if (16u * i + 16u <= wl)
                          { const u32x4 v2 = ((const u32x4 *) stretch)[i];
                            *(u32x4_u *) (o + 16u * i) = v2;
                          }
                        else
                          for (uint32_t j = 16u * i; j < wl; j++) o[j] = s8[j];
                      wave_sync();
                    }
                }
            else if (tags && !(DR_SKIP & 1))               // (a piece too long for 16-bit places: the tag line's straight to memory as well)
              { uint8_t *o = tout + base_pos;
                for (uint32_t k = 16u * (uint32_t) lane; k < tp; k += 1024u)
                  if (k + 16u <= tp) { const u32x4 v = { tpat, tpat, tpat, tpat }; *(u32x4_u *) (o + k) = v; }
                  else for (uint32_t j = k; j < tp; j++) o[j] = (uint8_t) tpat;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_s_waitcnt(0);
                wave_sync();
                if (c && tp)
                  { uint32_t W = Wtag;
                    #pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                      if (k < c)
                        { const uint32_t ch = ((0x74676361u >> (8u * (W >> 30))) & 0xffu) - fold;
                          W <<= 2;
                          (o - 1)[posk[k]] = (uint8_t) ch;
                        }
                  }
                wave_sync();
              }
            base_bit += tb;
            base_pos += tp;
          }
        // behind the last token: the run left open at the line's end (QV.c:497-504), and the line's end
        for (uint32_t k = base_pos + 16u * (uint32_t) lane; k < L; k += 1024u)
          if (k + 16u <= L) { const u32x4 v = { pat, pat, pat, pat }; *(u32x4_u *) (out + k) = v; }
          else for (uint32_t j = k; j < L; j++) out[j] = (uint8_t) rc;
        if (lane == 0) out[L] = '\n';
        if (tags)
          { for (uint32_t k = base_pos + 16u * (uint32_t) lane; k < L; k += 1024u)
              if (k + 16u <= L) { const u32x4 v = { tpat, tpat, tpat, tpat }; *(u32x4_u *) (tout + k) = v; }
              else for (uint32_t j = k; j < L; j++) tout[j] = (uint8_t) tpat;
            if (lane == 0) tout[L] = '\n';
          }
        if (__any((int) bad) && lane == 0) atomicOr(status, 4u);
      }
      if (more && !asked)                                  // (an entry without a pass)
        { DR_TAKE(nx, dv_nx)
          DR_HEADV(hv_nx, nx)
        }
    }
  nocode_end(status);
}

// Tag line of each entry (QV.c:1437-1461): tag[p] = 'n' where del[p] is the run character, else
// the next 2-bit code of the tag segment as a letter; one wavefront per entry.
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_decode_tags(dec_args a, const uint32_t *skip_idx, const uint64_t *skip_off)
{ // four tag bytes per look-up: index = (which of the 4 positions hold a non-run symbol) << 8 | the next
  // four 2-bit codes; the entry has letters at those positions (codes consumed in order), 'n' elsewhere
  __shared__ uint32_t s_quad[16 * 256];                    // 16 KB
  for (uint32_t k = threadIdx.x; k < 16u * 256u; k += DX_BLOCK)
    { const uint32_t m = k >> 8, fold_ = a.upper ? 32u : 0u;
      uint32_t c = k & 0xffu, v = 0;
      for (int j = 0; j < 4; j++)
        { uint32_t ch = 'n';
          if ((m >> j) & 1u)
            { ch = (0x74676361u >> (8 * (c >> 6))) & 0xffu;       // "acgt", Lower_Read DB.c:367
              c  = (c << 2) & 0xffu;
            }
          v |= (ch - fold_) << (8 * j);
        }
      s_quad[k] = v;
    }
  __syncthreads();
  const int      lane  = lane_id();
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;

  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const uint32_t  L   = a.len[r];
      if (skip_idx != NULL && skip_idx[skip_off[r] + run_base(L)] != RUN_NONE)
        continue;                                          // k_qv_decode_runs wrote this entry's tag line with its deletion line
      const uint32_t *sg  = a.seg + 5 * r;
      const uint8_t  *src = a.in + a.rec_off[r] + (a.hdr_off ? a.hdr_off[r + 1] - a.hdr_off[r] : 0) + sg[0];
      const uint32_t  tb  = sg[1];                         // packed tag bytes
      uint8_t        *del = a.out + a.out_off[r];
      uint8_t        *tag = del + (uint64_t) L + 1u;
      uint32_t G = 0;                                      // non-run symbols so far
      for (uint32_t base = 0; base < L; base += DX_STEP)
        { const uint32_t pos   = base + 16u * lane;
          const int      valid = pos >= L ? 0 : (L - pos >= 16u ? 16 : (int) (L - pos));
          const uint32_t vm    = (1u << valid) - 1u;
          uint32_t nr = vm;
          if (a.delChar >= 0)
            nr &= ~chunk_eq_mask(load_chunk(del + pos, valid), (uint32_t) a.delChar);
          const uint32_t cnt  = __popc(nr);
          const uint32_t incl = wave_incl_scan(cnt);
          uint32_t       idx  = G + incl - cnt;            // rank of this lane's first non-run symbol
          const uint32_t b0   = idx >> 2;
          uint64_t bits = 0;
          if (cnt)
            { if (b0 + 8u <= tb) bits = *(const u64_u *) (src + b0);
              else for (uint32_t k = b0; k < tb; k++) bits |= (uint64_t) src[k] << (8 * (k - b0));
            }
          // the lane's codes, first one in the top bits (symbol idx sits at bit pair idx & 3 of byte b0)
          const uint64_t be = ((uint64_t) __builtin_bswap32((uint32_t) bits) << 32) | __builtin_bswap32((uint32_t) (bits >> 32));
          uint32_t cw = (uint32_t) ((be << (2u * (idx & 3u))) >> 32);
          uint32_t w[4];
          #pragma unroll
          for (int k = 0; k < 4; k++)
            { const uint32_t m4 = (nr >> (4 * k)) & 15u;
              w[k] = s_quad[(m4 << 8) | (cw >> 24)];
              cw <<= 2u * (uint32_t) __popc(m4);
            }
          if (valid == 16)
            { u32x4 v = { w[0], w[1], w[2], w[3] };
              *(u32x4_u *) (tag + pos) = v;
            }
          else
            for (int b = 0; b < valid; b++)
              tag[pos + b] = (uint8_t) (w[b >> 2] >> (8 * (b & 3)));
          G += wave_total(incl);
        }
      if (lane == 0)
        tag[L] = '\n';
    }
}

// A group index made elsewhere (the host walk of a bare file, dx_qv_walk_indexed) for the stream at d_in / d_seg.
extern "C" int dx_qv_use_index(dx_ctx *ctx, const uint8_t *d_in, const uint32_t *d_seg, uint64_t n,
                               const uint32_t *d_gidx, const uint64_t *d_gidx_off, uint64_t none)
{ if (ctx == NULL) return DX_E_ARG;
  if (ctx->op.pending)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_use_index: an encode has begun in this context: end it first (dx_qv_encode_onepass_end)");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  if (d_gidx == NULL)                                    // take it back
    { dx_sx_drop_external(ctx);
      return DX_OK;
    }
  if (!d_in || !d_seg || !d_gidx_off || n == 0)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_use_index: NULL device pointer or no entries");
  if (!ctx->sx.external)                                 // the context's own index buffers go: one index at a time
    { DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void) hipFree(ctx->sx.idx); (void) hipFree(ctx->sx.off); (void) hipFree(ctx->sx.room);
      ctx->sx.idx = NULL; ctx->sx.off = NULL; ctx->sx.room = NULL; ctx->sx.cap_idx = 0; ctx->sx.cap_entries = 0;
    }
  if (ctx->sx.none == NULL && hipMalloc((void **) &ctx->sx.none, 64) != hipSuccess)
    { (void) hipGetLastError();
      return dx_fail(ctx, DX_E_NOMEM, "dx_qv_use_index: no memory");
    }
  const uint32_t none32 = none > 0xffffffffull ? 0xffffffffu : (uint32_t) none;
  DX_HIP(ctx, hipMemcpyAsync(ctx->sx.none, &none32, 4, hipMemcpyHostToDevice, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->sx.idx = (uint32_t *) d_gidx; ctx->sx.off = (uint64_t *) d_gidx_off;
  ctx->sx.out = d_in; ctx->sx.seg = d_seg; ctx->sx.n = n;
  ctx->sx.external = 1; ctx->sx.valid = 1; ctx->sx.walk = 0;
  return DX_OK;
}

// ... and the one a device walk has left (dx_qv_walk_device): the run-coded lines' groups only
extern "C" int dx_qv_use_dindex(dx_ctx *ctx, const uint8_t *d_in, const dx_qv_dindex *x)
{ if (ctx == NULL) return DX_E_ARG;
  if (x == NULL || x->d_gidx == NULL || x->n == 0)
    return dx_qv_use_index(ctx, NULL, NULL, 0, NULL, NULL, 0);
  const int rc = dx_qv_use_index(ctx, d_in, x->d_seg, x->n, x->d_gidx, x->d_gidx_off, x->gidx_none);
  if (rc == DX_OK) { ctx->sx.walk = 1; ctx->sx.sync_kinds = x->sync_kinds; ctx->sx.nosync = x->gidx_nosync; }
  return rc;
}

extern "C" int dx_qv_decode(dx_ctx *ctx, const uint8_t *d_in, const uint64_t *d_rec_off, const uint64_t *d_hdr_off,
                            const uint32_t *d_seg, const uint32_t *d_len, uint64_t n, int flags,
                            uint8_t *d_out, const uint64_t *d_out_off)
{ if (ctx == NULL) return DX_E_ARG;
  if (!ctx->coding_set) return dx_fail(ctx, DX_E_ARG, "dx_qv_decode: call dx_qv_set_coding first");
  if (ctx->op.pending)                                   // (its verdict sits in d_status, its output may be d_in)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_decode: an encode has begun in this context: end it first (dx_qv_encode_onepass_end)");
  if (n == 0) return DX_OK;
  if (n >= (1ull << 31))
    return dx_fail(ctx, DX_E_ARG, "dx_qv_decode: more than 2^31 - 1 entries in one batch");
  if (!d_in || !d_rec_off || !d_seg || !d_len || !d_out || !d_out_off)
    return dx_fail(ctx, DX_E_ARG, "dx_qv_decode: NULL device pointer");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  { const int e = dx_dec_tables(ctx);                    // (dx_qv_set_coding leaves them on the host until a decode asks)
    if (e) return e;
  }
  DX_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
  dec_args a;
  a.in = d_in; a.rec_off = d_rec_off; a.hdr_off = d_hdr_off; a.seg = d_seg; a.len = d_len; a.n = n;
  a.out = d_out; a.out_off = d_out_off; a.delChar = ctx->delChar; a.subChar = ctx->subChar; a.upper = (flags & DX_DECODE_UPPER) != 0; a.flip = (flags & DX_DECODE_FLIP) != 0;
  for (int s = 0; s < 4; s++) a.type[s] = ctx->sym_type[s];
  uint64_t blocks = (4 * ((n + 63) / 64) + DEC_NWAVE - 1) / DEC_NWAVE;
  const uint64_t cap = (uint64_t) ctx->num_cu;
  if (blocks > cap) blocks = cap;
  // plain lines without escape codes go to k_qv_decode_plain, the rest (run-coded lines, schemes with 8-bit
  // escapes) to the generic kernel; DEXGPU_GENERIC_DECODE keeps everything on the generic one
  uint32_t plain = 0;
  for (int q = 0; q < 4; q++)
    { const int rc = q == 0 ? ctx->delChar : (q == 3 ? ctx->subChar : -1);
      if (rc < 0 && ctx->sym_type[q] != 2) plain |= 1u << q;
    }
  if (dx_test_on("generic_decode")) plain = 0;
  uint32_t *d_next = (uint32_t *) (ctx->d_u64 + 16), *d_next2 = (uint32_t *) (ctx->d_u64 + 31);   // task counters
  DX_HIP(ctx, hipMemsetAsync(d_next, 0, 4, ctx->stream));
  DX_HIP(ctx, hipMemsetAsync(d_next2, 0, 4, ctx->stream));
  const uint32_t plain_kinds = plain;
  const uint32_t *skip_idx = NULL;
  const uint64_t *skip_off = NULL;
  uint32_t        skip_kinds = 0;
  // the encoder's group index for this very stream (dx_qv_subindex): a wavefront per line instead of a lane
  const uint32_t *sync_idx = NULL;                       // (the device walk's index: what is left for k_qv_decode_plain)
  const uint64_t *sync_off = NULL;
  uint32_t        sync_kinds = 0;
  if ((plain || ctx->sx.walk) && ctx->sx.valid && d_in == ctx->sx.out && !(flags & DX_DECODE_FLIP) && !dx_test_on("no_subindex") &&
      (const uint32_t *) d_seg >= (const uint32_t *) ctx->sx.seg &&
      ((const uint32_t *) d_seg - (const uint32_t *) ctx->sx.seg) % 5 == 0)
    { const uint64_t first = (uint64_t) ((const uint32_t *) d_seg - (const uint32_t *) ctx->sx.seg) / 5;
      if (first + n <= ctx->sx.n)
        { uint32_t *d_next3 = (uint32_t *) (ctx->d_u64 + 30);
          DX_HIP(ctx, hipMemsetAsync(d_next3, 0, 4, ctx->stream));
          hipLaunchKernelGGL(k_ticket_units, dim3(1), dim3(1), 0, ctx->stream, a.rec_off, a.rec_off + n, (const uint32_t *) NULL, n,
                             DEC_TICKET * 14000u, DEC_TICKET, d_next3);         // (4 entries of 10 kb per ticket; more of shorter ones)
          if (ctx->sx.walk)                                 // the device walk's index (dx_qv_use_dindex): a word per 64 symbols of a plain line
            { const uint32_t sy = plain & ctx->sx.sync_kinds & (!dx_test_on("no_syncindex") ? 15u : 0u);
              const int nk = __builtin_popcount(sy);
              uint64_t sb = (n + DY_NWAVE - 1) / DY_NWAVE;
              if (sb > cap * (nk <= 2 ? 2 : 1)) sb = cap * (nk <= 2 ? 2 : 1);
#define SYNC_LAUNCH(NK)                                                                                       \
              DX_LAUNCH(ctx, DX_K_QV_DEC_SUB, k_qv_decode_sync<NK>, (int) sb, DY_BLOCK, a, (const uint16_t *) ctx->d_dec,  \
                        (const uint32_t *) ctx->d_long, d_next3, sy, (const uint32_t *) ctx->sx.idx,           \
                        (const uint64_t *) (ctx->sx.off + first), ctx->d_status)
              if (nk == 1)      SYNC_LAUNCH(1);
              else if (nk == 2) SYNC_LAUNCH(2);
              else if (nk == 3) SYNC_LAUNCH(3);
              else if (nk == 4) SYNC_LAUNCH(4);
#undef SYNC_LAUNCH
              if (sy)
                { if (ctx->sx.nosync == 0) plain &= ~sy;   // every such line had its words
                  else { sync_idx = ctx->sx.idx; sync_off = ctx->sx.off + first; sync_kinds = sy; }
                }
            }
          else
          {
          const int nk = __builtin_popcount(plain);
          uint64_t sb = (n + DS_NWAVE - 1) / DS_NWAVE;
          if (sb > cap * (nk <= 2 ? 2 : 1)) sb = cap * (nk <= 2 ? 2 : 1);       // two workgroups per CU fit with <= 2 tables
#define SUB_LAUNCH(NK)                                                                                        \
          DX_LAUNCH(ctx, DX_K_QV_DEC_SUB, k_qv_decode_sub<NK>, (int) sb, DS_BLOCK, a, (const uint16_t *) ctx->d_dec,   \
                    (const uint32_t *) ctx->d_long, d_next3, plain, (const uint32_t *) ctx->sx.idx,            \
                    (const uint64_t *) (ctx->sx.off + first), ctx->d_status)
          if (nk == 1)      SUB_LAUNCH(1);
          else if (nk == 2) SUB_LAUNCH(2);
          else if (nk == 3) SUB_LAUNCH(3);
          else              SUB_LAUNCH(4);
#undef SUB_LAUNCH
          plain = 0;                                       // done
          }
          // the run-coded lines whose symbols have no escape code, by their token groups; what has no index
          // (entries the generic encoder took) is left to k_qv_decode below
          uint32_t runs = 0;
          if (ctx->delChar >= 0 && ctx->sym_type[0] != 2) runs |= 1u;
          if (ctx->subChar >= 0 && ctx->sym_type[3] != 2) runs |= 8u;
          if (runs && !dx_test_on("no_runindex"))
            { uint32_t *d_next4 = (uint32_t *) (ctx->d_u64 + 29);
              DX_HIP(ctx, hipMemsetAsync(d_next4, 0, 4, ctx->stream));
              hipLaunchKernelGGL(k_ticket_units, dim3(1), dim3(1), 0, ctx->stream, a.rec_off, a.rec_off + n, (const uint32_t *) NULL, n,
                                 DEC_TICKET * 14000u, DEC_TICKET, d_next4);
              uint64_t rb = (n + DR_NWAVE - 1) / DR_NWAVE;
              if (rb > cap * DR_WG_PER_CU) rb = cap * DR_WG_PER_CU;
              if (runs == 9u)
                DX_LAUNCH(ctx, DX_K_QV_DEC_RUNS, k_qv_decode_runs<2>, (int) rb, DR_BLOCK, a, (const uint16_t *) ctx->d_dec,
                          (const uint32_t *) ctx->d_long, ctx->d_status, d_next4, runs, (const uint32_t *) ctx->sx.idx,
                          (const uint64_t *) (ctx->sx.off + first));
              else
                DX_LAUNCH(ctx, DX_K_QV_DEC_RUNS, k_qv_decode_runs<1>, (int) rb, DR_BLOCK, a, (const uint16_t *) ctx->d_dec,
                          (const uint32_t *) ctx->d_long, ctx->d_status, d_next4, runs, (const uint32_t *) ctx->sx.idx,
                          (const uint64_t *) (ctx->sx.off + first));
              skip_idx = ctx->sx.idx; skip_off = ctx->sx.off + first; skip_kinds = runs;
            }
        }
    }
  if (plain)
    { uint64_t pb = (4 * ((n + 63) / 64) + DP_NWAVE - 1) / DP_NWAVE;
      if (pb > cap * DP_WG_PER_CU) pb = cap * DP_WG_PER_CU;
      DX_LAUNCH(ctx, DX_K_QV_DEC_PLAIN, k_qv_decode_plain, (int) pb, DP_BLOCK, a, (const uint16_t *) ctx->d_dec,
                (const uint32_t *) ctx->d_long, d_next2, plain, ctx->d_status, sync_idx, sync_off, sync_kinds);
    }
  if (plain_kinds != 15u)
    DX_LAUNCH(ctx, DX_K_QV_DECODE, k_qv_decode, (int) blocks, DEC_BLOCK, a, (const uint16_t *) ctx->d_dec,
              (const uint32_t *) ctx->d_long, ctx->d_status, d_next, 15u & ~plain_kinds,
              (const uint32_t *) skip_idx, (const uint64_t *) skip_off, skip_kinds, (const uint32_t *) ctx->sx.none,
              ((15u & ~plain_kinds) & 6u) == 0u ? (uint32_t) (2 * ((n + 63) / 64)) : 0u);
  DX_LAUNCH(ctx, DX_K_QV_DEC_TAGS, k_qv_decode_tags, dx_grid_waves(ctx, n, 16), DX_BLOCK, a,
            (skip_kinds & 1u) ? (const uint32_t *) skip_idx : (const uint32_t *) NULL, (const uint64_t *) skip_off);
  uint32_t st = 0;
  DX_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st & 4u)
    return dx_fail(ctx, DX_E_FORMAT, "dx_qv_decode: a run overruns its entry (corrupt stream or wrong index)");
  if (st & DEC_ST_NOCODE)
    return dx_fail(ctx, DX_E_FORMAT, "dx_qv_decode: a code that is in no table (a corrupt stream, or a line whose scheme holds a single "
                                     "symbol, which has no bits: the reference stops with \"Could not read more bits (Decode)\")");
  return DX_OK;
}
