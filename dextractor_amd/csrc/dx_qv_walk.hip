// dx_qv_walk.hip -- the record walk of a bare .dexqv stream on the device.
//
// The format stores no record or segment lengths (QV.c:1428-1481, undexqv.c:119-208): where a segment ends is known only
// after every code of it has been passed, so the index dx_qv_decode wants -- where every record starts, how long its five
// segments are -- takes a walk over the whole stream, code by code, in the order of the file.  The host does that on up to
// 32 threads (dx_host.c: walk_parallel; 7.5 s for the 14 GB of records of the 1 M x 10 kb batch); a GPU has no fast single
// thread, but it has half a million slow ones, and the host walk's idea carries over lane for lane:
//
//   * the stream is cut into pieces of 32 KiB or more (as many as the device holds lanes at once: 1536 a CU); k_walk_find, a wave per piece, notes
//     the first offsets in each piece at which a plausible record header stands (header_plausible, as on the host: 1.4e-8 of
//     all offsets pass by chance -- and the byte or two in front of every true header, see there);
//   * k_walk_pieces, ONE LANE per piece, tries those offsets in turn -- a start is taken when the record walks cleanly from
//     it (within a budget of two pieces' bytes: garbage may claim any length) and another plausible header, or the stream's
//     end, stands behind it -- and then walks record after record until it has left its piece, noting every record on the
//     way.  The walk needs code LENGTHS only; the look-up tables are the host walk's, 16 bits an entry, in LDS (dx_walk.h);
//   * the host looks at 48 bytes per piece: arriving exactly where the next piece's lane started proves both (a walk from a
//     wrong offset does not fall back onto record boundaries: framing fields and pad words are not self-delimiting).  A
//     piece the chain does not arrive at the start of is walked again from where the chain does arrive, a lane of its own,
//     until the chain is whole; a record that does not walk, or a chain that will not close in a few dozen rounds, is
//     DX_E_MISMATCH and the caller's business (the host walk, which also knows what to say about a damaged file);
//   * k_walk_gather, a wave per piece, puts the records of the pieces on the chain side by side.
//
// Only offsets on an unbroken chain of walks from the first record are kept: the result is the host walk's by construction
// (tests/test_gpu_walk.py: word for word, on pieces from 4 KiB up).
// Roofline: none of the usual ones -- 64 independent chains of dependent look-ups per wave, a third of the instructions
// scalar (divergent control flow), waves waiting two thirds of their time (SQ_WAIT_ANY) on LDS and load latency: 53 ms for
// those 14 GB (270 GB/s), a hundred times the host's 32 threads.  profiles/r04_device_walk.txt.
#include "dx_internal.hpp"
#include "dx_device.hpp"
#include "dx_walk.h"
#include "dx_layout.h"

#include <stdlib.h>

int dx_scan_u32(dx_ctx *ctx, const uint32_t *d_in, uint64_t n, uint64_t *d_out /* n+1 */, uint64_t *total);     // dx_qv.hip

#ifndef WALK_BLOCK
#define WALK_BLOCK     1024             // one workgroup a CU: 80 KB of tables and 72 KB of rings (a lane's next 16 words) in LDS
#endif
#define WALK_WGS_PER_CU 1
#define WALK_CAND      4u
#ifndef WALK_PIECE_KB
#define WALK_PIECE_KB 16
#endif
#define WALK_PIECE_MIN ((uint64_t) WALK_PIECE_KB << 10)
#ifndef WALK_SERVE
#define WALK_SERVE 4u                    // a power of two: every how many turns lanes end and begin pieces
#endif
#ifndef WALK_PER_LANE
#define WALK_PER_LANE 2u                 // pieces a lane, on average (a lane takes the next piece nobody has taken)
#endif
#define WALK_LANES_MAX ((uint64_t) 512 * 1024)
#define WALK_ROUNDS    48
// What one lane will walk of its own accord: a damaged stream can claim an entry of 2^31 symbols or a gigabyte of 255s, and a lane
// is a slow walker (minutes for that; every other lane of the kernel long done).  Entries beyond 4 M symbols (the bound of the
// guesses too) and headers with more than 64 K leading 255s (16 M wells) are left to the host walk: DX_E_MISMATCH.
#define WALK_RLEN_MAX  (1 << 22)
#define WALK_LEAD_MAX  65536u
// A header that is no header (1.4e-8 of all offsets read plausibly) claims whatever its bytes say -- two million symbols on average --
// and any bits walk as codes (a Huffman code is complete): the lane that tries it walks garbage until its budget is spent, alone,
// long after every other wave has left (one such guess in 250 000 pieces made a 30 ms kernel a 52 ms one).  So a GUESS is taken
// only when it claims at most WALK_TRIAL_RLEN symbols, and is given 32 bits a claimed symbol (no stream of a true record
// averages 8 bits a code), two pieces' bytes at most; a piece whose first record is longer is left to the chain, which walks from
// where it arrives without guessing.
#define WALK_TRIAL_RLEN 65536
#define WALK_RECLUSTER 8u               // clusters of guesses a lane looks for by itself before it leaves its piece to the chain

#define WP_NONE     1u            // no start in this piece
#define WP_BAD      2u            // a record on this lane's chain did not walk
#define WP_OVERFLOW 4u            // more records than the lane's scratch holds

struct walk_args
{ const uint8_t  *img;
  uint64_t        n, first, piece, pieces;
  const uint16_t *w16, *mw, *rw, *r1;     // the blob's tables in device memory (dx_walk.h)
  const uint16_t *one;
  const uint8_t  *tail;                   // the image's last bytes from tail_at on, zeros behind them: 256 bytes
  uint64_t        tail_at;
  int             delChar, subChar, flip;
  int             esc[4];
};

struct walk_rec_d   { uint64_t off; uint32_t hdr_bytes, len, seg[5]; int32_t dwell, beg, end, qv; uint32_t pad;
                      uint32_t gat[4], gtok[4];   /* what the lane noted for the group index, line by line (del, ins, mrg, sub): where it
                                                     starts in the lane's share of the group words, and -- a run-coded line -- the line's
                                                     tokens (a word per 8 of them), -- a plain line -- the words noted (one per 64 symbols
                                                     but the first); WG_NONE: nothing usable */ };
#define WG_NONE 0xffffffffu
struct walk_piece_d { uint64_t start, end, dwell_sum; uint32_t count, flags, hdr_sum, first_hdr /* framing bytes of the first record */; uint32_t tried /* guesses that did not hold */, lead255 /* bytes of 255 right in front of start (up to 4096) */; };

__device__ __forceinline__ uint32_t bswap_if(uint32_t v, int flip) { return flip ? __builtin_bswap32(v) : v; }

__device__ __forceinline__ uint32_t load32_at(const uint8_t *p) { return *(const u32_u *) p; }

// header_plausible of dx_host.c, word for word
__device__ __forceinline__ bool header_plausible_d(const walk_args &a, uint64_t at, int32_t maxlen = (1 << 22))
{ int k = 0;
  while (at < a.n && a.img[at] == 255 && k < 16) { at += 1; k += 1; }
  if (at + 13 > a.n) return false;
  at += 1;
  const int32_t beg = (int32_t) bswap_if(load32_at(a.img + at), a.flip), end_ = (int32_t) bswap_if(load32_at(a.img + at + 4), a.flip);
  const int32_t qv  = (int32_t) bswap_if(load32_at(a.img + at + 8), a.flip);
  return beg >= 0 && beg < (1 << 28) && end_ >= beg && end_ - beg <= maxlen && qv >= 0 && qv < 1000000 &&
         (uint64_t) (end_ - beg) <= 8u * (uint64_t) (a.n - at);
}

// ---------------------------------------------------------------------------------------------
//  k_walk_find: the first plausible offsets of every piece (a wave per piece, 256 offsets a round)
// ---------------------------------------------------------------------------------------------
// An offset whose byte is 255 is never a guess: a header may begin with bytes of 255 (255 wells each, undexqv.c:124-133), but
// then the offset behind them reads as a header too (fewer wells) and walks the same -- that one is the guess, k_walk_pieces
// notes how many bytes of 255 stand in front of the start it took, and the chain, which knows where the record before ended,
// gives them back to the header (a negative trim).  So every test is the thirteen bytes of one 16-byte load.
// The scan stops behind the FIRST cluster of guesses (guesses within 16 bytes of each other: a true header and the offsets
// one, two and nine bytes in front of it, whose fields read plausibly as well; WALK_CAND of them at most): half a record's
// bytes on average.  Should none of them hold (one piece in ten thousand: a chance hit in front of the first header), the
// piece's lane looks for the next cluster itself (k_walk_pieces).
__device__ __forceinline__ bool header_fast_d(const walk_args &a, uint64_t p, const u32x4 &v)
{ const int32_t beg  = (int32_t) bswap_if((v.x >> 8) | (v.y << 24), a.flip);
  const int32_t end_ = (int32_t) bswap_if((v.y >> 8) | (v.z << 24), a.flip);
  const int32_t qv   = (int32_t) bswap_if((v.z >> 8) | (v.w << 24), a.flip);
  return (v.x & 0xffu) != 255u && beg >= 0 && beg < (1 << 28) && end_ >= beg && end_ - beg <= WALK_TRIAL_RLEN && qv >= 0 && qv < 1000000 &&
         (uint64_t) (end_ - beg) <= 8u * (uint64_t) (a.n - (p + 1));
}

__global__ __launch_bounds__(DX_BLOCK)
void k_walk_find(walk_args a, uint64_t *cand, uint32_t *ncand)
{ const uint32_t lane = (uint32_t) lane_id();
  const uint64_t k = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  if (k >= a.pieces) return;
  if (k == 0) { if (lane == 0) ncand[0] = 0; return; }            // (the first piece starts at the first record)
  const uint64_t lo = a.first + k * a.piece, hi = k == a.pieces - 1 ? a.n : lo + a.piece;
  uint32_t found = 0, clusters = 0;
  uint64_t last = 0;
  bool     stop = false;
  // Every offset is a possible header, but not every offset wants a load of its own: 64 lanes reading 16 bytes each at offsets a byte
  // apart are 64 requests to the cache for 79 bytes (k_walk_find spent its 4.4 ms there, whatever it did with the bytes: fewer
  // instructions, requests a round ahead, 16 of them in flight changed nothing).  A lane takes 16 CONSECUTIVE offsets: two 16-byte
  // loads, lane after lane (the wave's requests are a KiB in a row), and looks at the quality value of each of its 16 offsets in
  // registers -- bytes 9 .. 12 below a million: one offset in 4 000 of a stream of code bits passes; only those are tested in full
  // (the wave together, one offset at a time, in the order of the file).
#define FIND_KIB 4
  for (uint64_t base = lo; base < hi && !stop; base += 1024u * FIND_KIB)
    { u32x4 A[FIND_KIB], B[FIND_KIB];
      #pragma unroll
      for (int u = 0; u < FIND_KIB; u++)
        { const uint64_t p = base + 1024u * u + 16u * lane;
          A[u] = u32x4{ 0u, 0u, 0u, 0u }; B[u] = A[u];
          if (p < hi && p + 32 <= a.n) { A[u] = *(const u32x4_u *) (a.img + p); B[u] = *(const u32x4_u *) (a.img + p + 16); }
        }
      #pragma unroll
      for (int u = 0; u < FIND_KIB; u++)
        { const uint64_t p = base + 1024u * u + 16u * lane;
          const bool have = p < hi && p + 32 <= a.n;
          const uint32_t w[8] = { A[u].x, A[u].y, A[u].z, A[u].w, B[u].x, B[u].y, B[u].z, B[u].w };
          uint32_t mask = 0;
          #pragma unroll
          for (int o = 0; o < 16; o++)
            { const uint32_t x = __builtin_amdgcn_alignbyte(w[((o + 9) >> 2) + 1], w[(o + 9) >> 2], (uint32_t) ((o + 9) & 3));
              mask |= (bswap_if(x, a.flip) < 1000000u ? 1u : 0u) << o;
            }
          if (!have) mask = 0;
          uint64_t hits = __ballot(mask != 0u);
          while (hits && !stop)
            { const uint32_t l = (uint32_t) __ffsll((unsigned long long) hits) - 1u;
              uint32_t ml = (uint32_t) __builtin_amdgcn_readlane((int) mask, (int) l);
              hits &= hits - 1;
              while (ml && !stop)
                { const uint32_t o = (uint32_t) __ffs((int) ml) - 1u;
                  const uint64_t q = base + 1024u * u + 16u * l + o;
                  ml &= ml - 1;
                  if (q >= hi) { hits = 0; break; }
                  const u32x4 v = *(const u32x4_u *) (a.img + q);        // (q + 16 <= a.n: the lane had its 32 bytes)
                  if (!header_fast_d(a, q, v)) continue;
                  if (found == 0 || q - last > 16u) clusters += 1;
                  if (clusters > 1u || found == WALK_CAND) { stop = true; break; }
                  if (lane == 0) cand[k * WALK_CAND + found] = q;
                  found += 1; last = q;
                }
            }
          if (found && base + 1024u * (u + 1) > last + 16u) stop = true;                                 // (the cluster is complete)
        }
    }
  // the image's last bytes (offsets whose 32 bytes reach behind it): byte by byte
  if (!stop && hi == a.n && a.n >= lo)
    { const uint64_t from = a.n > lo + 48u ? a.n - 48u : lo;
      const uint64_t q0 = from + lane;
      bool ok = false;
      if (lane < 48u && q0 < a.n && ((q0 - lo) & ~(uint64_t) 15) + lo + 32 > a.n && a.img[q0] != 255) ok = header_plausible_d(a, q0, WALK_TRIAL_RLEN);
      uint64_t m = __ballot(ok);
      while (m && !stop)
        { const uint32_t l = (uint32_t) __ffsll((unsigned long long) m) - 1u;
          const uint64_t q = from + l;
          m &= m - 1;
          if (found == 0 || q - last > 16u) clusters += 1;
          if (clusters > 1u || found == WALK_CAND) { stop = true; break; }
          if (lane == 0) cand[k * WALK_CAND + found] = q;
          found += 1; last = q;
        }
    }
  // A true header at p makes p - 1 plausible too wherever the quality value is below 3906 and the entry short of 16 k
  // symbols (its fields read one byte early: 256 times the length, 256 times the quality value) -- a guess that costs a
  // walk of its garbage length, the whole budget -- and a lane that walks garbage for the length of two pieces holds up the
  // other 63 of its wave (one guess in eighty was such: half the waves had one).  Of guesses within 16 bytes of each other
  // the last goes first: if it holds the earlier ones cannot (no record has fewer than 13 bytes).
  if (lane == 0)
    { uint64_t *c = cand + k * WALK_CAND;
      for (uint32_t i = 0; i < found; )                      // (every cluster back to front: two bytes early happens too, and nine --
        { uint32_t e = i;                                      //  the quality value read from the well byte and a small `beg`)
          while (e + 1 < found && c[e + 1] - c[e] <= 16u) e++;
          for (uint32_t x = i, y = e; x < y; x++, y--) { const uint64_t t = c[x]; c[x] = c[y]; c[y] = t; }
          i = e + 1;
        }
      ncand[k] = found;
    }
}

// ---------------------------------------------------------------------------------------------
//  the walk of one lane
// ---------------------------------------------------------------------------------------------
// The 64 lanes of a wave walk 64 different pieces, each somewhere else in its record: one in the insertion line, one in the
// deletion line's runs, one reading a header.  Were every segment's walk a function of its own (as on the host), a wave
// would run them one after the other for the lanes that happen to be in each -- so the whole walk of a piece is ONE loop
// whose body is one look-up of whatever segment the lane is in, and the five segments, the header and the record's end are
// states of the lane (ph).  The loop's common path: refill, a 16-bit look-up in LDS, a shift.
//
// MSB-first bit reader over the segment's 32-bit words (w_peek / w_skip of dx_host.c), by POSITION: the lane keeps the number
// of bits it has passed (T) and a ring of the segment's next 16 words in LDS; a look-up reads the two words its position falls in
// and shifts -- no bit buffer, no refill inside the look-up loop.  The ring is filled by the wave as a whole, once per turn of the
// lane loop (w_pump_d): what was asked for a turn ago goes into the ring, and up to two 16-byte requests go out for the room that
// is free -- so the one wait for memory of a turn is for requests a whole burst old, and no look-up ever waits on a load (round
// 4's reader asked and waited lane by lane: with 64 lanes in 64 phases that was a `s_waitcnt vmcnt(0)` in nearly every look-up,
// and 57 instructions a look-up of refill and select logic; the kernel was bound by instruction issue, profiles/r05_device_walk.txt).
// A turn passes at most WALK_BURST codes of <= 12 bits and one step of another kind (<= 16 + 16 + 16 + 8 bits): <= 152 bits, i.e.
// at most 5 words further on, and a look-up reads the word after its own: a turn that starts with >= 7 words in the ring never
// reads past them.  w_pump_d keeps >= 8: with <= 8 in the ring it asks for 8 more, with <= 12 for 4.
// Bytes behind the image's end read as zero; a segment's end compares the bytes used with what is there.
#define WRING_WORDS  16u
#define WRING_STRIDE 17u        // words a lane: the ring and its first word once more behind it (a look-up's two words never wrap); odd: the
                                // 64 rows of a wave start in 32 different banks
struct wrd_d
{ const uint8_t *seg;           // the segment's first byte
  uint32_t *ring;               // the lane's ring
  uint32_t T;                   // bits of the segment passed
  uint32_t have;                // words of the segment that have reached the ring (a multiple of 4)
  uint32_t ua, ub;              // 16-byte requests under way in either set: 0, 1, 2
  u32x4    a0, a1, b0, b1;      // what they bring: set a asked for by the even pumps, set b by the odd ones
};
// words in the ring from the one the lane stands in: a burst of WALK_BURST look-ups wants WRING_BURST of them (<= 12 bits a
// look-up: 3 words further on at most, and a look-up reads the word after its own), a step of another kind WRING_STEP (<= 56 bits)
#define WRING_BURST 5u
#define WRING_STEP  4u
__device__ __forceinline__ uint32_t w_words_d(const wrd_d &r) { return r.have - (r.T >> 5); }

// 16 bytes of the image at q, zeros behind its end: the image's last bytes come from a padded copy (walk_args.tail), by a select
// of the address -- no branch, and nothing is done to the bytes here (a byte swap is the taker's business, w_commit_d).
__device__ __forceinline__ const uint8_t *within(const walk_args &a, const uint8_t *q)
{ const uint64_t off = (uint64_t) (q - a.img);
  const uint64_t t   = off - a.tail_at;
  return off < a.tail_at ? q : a.tail + (t < 240u ? t : 240u);
}
__device__ __forceinline__ u32x4 load16_within(const walk_args &a, const uint8_t *q) { return *(const u32x4_u *) within(a, q); }

__device__ __forceinline__ void w_commit_d(wrd_d &r, const u32x4 &v, int flip)
{ const uint32_t s = r.have & (WRING_WORDS - 1u);
  const uint32_t x = bswap_if(v.x, flip), y = bswap_if(v.y, flip), z = bswap_if(v.z, flip), w = bswap_if(v.w, flip);
  uint32_t *d = r.ring + s;
  d[0] = x; d[1] = y; d[2] = z; d[3] = w;
  if (s == 0u) r.ring[WRING_WORDS] = x;
  r.have += 4u;
}
// The rings are filled by the wave as a whole, in front of every burst: what was asked for TWO pumps ago goes into the ring, and up
// to two 16-byte requests go out for the room that will be free.  Every pump issues exactly TWO loads, whatever its lanes want (a
// lane with nothing to ask for reads a.tail, a cache hit for the whole wave): a wave has one counter for its loads and they retire
// in order, so the compiler, which counts instructions, can wait in front of the commit for all but the last pump's two -- were
// the requests conditional it would have to wait for everything outstanding, a full memory round trip per burst.  How much a lane
// may pass before its ring runs dry is not left to arithmetic over what the pumps keep in stock: a lane whose ring holds fewer
// words than a burst (a step) can use sits the burst (the step) out, w_words_d.
template <int SET>
__device__ __forceinline__ void w_pump_d(wrd_d &r, const walk_args &a, bool want)
{ u32x4 &c0 = SET ? r.b0 : r.a0, &c1 = SET ? r.b1 : r.a1;
  uint32_t &u = SET ? r.ub : r.ua;
  const uint32_t other = SET ? r.ua : r.ub;
  if (u >= 1u) w_commit_d(r, c0, a.flip);
  if (u == 2u) w_commit_d(r, c1, a.flip);
  const uint32_t ahead = r.have + 4u * other, occ = ahead - (r.T >> 5);
  u = want && occ <= 12u ? (occ <= 8u ? 2u : 1u) : 0u;
  const uint8_t *q  = r.seg + 4ull * ahead;
  const uint8_t *p0 = u >= 1u ? within(a, q) : a.tail, *p1 = u == 2u ? within(a, q + 16) : a.tail;
  c0 = *(const u32x4_u *) p0;
  c1 = *(const u32x4_u *) p1;
}
__device__ __forceinline__ void w_open_d(wrd_d &r, const walk_args &a, const uint8_t *p)
{ r.seg = p; r.T = 0u; r.have = 0u; r.ua = 0u; r.ub = 0u;       // (what is under way for the segment before is dropped)
  const u32x4 c0 = load16_within(a, p), c1 = load16_within(a, p + 16);
  w_commit_d(r, c0, a.flip);
  w_commit_d(r, c1, a.flip);                                    // 8 words: a burst's worth and a step's
}
// the next 32 bits
__device__ __forceinline__ uint32_t w_win_d(const wrd_d &r)
{ const uint32_t *p = r.ring + ((r.T >> 5) & (WRING_WORDS - 1u));
  const uint64_t two = ((uint64_t) p[0] << 32) | p[1];
  return (uint32_t) ((two << (r.T & 31u)) >> 32);
}
__device__ __forceinline__ uint32_t w_peek_d(const wrd_d &r) { return w_win_d(r) >> 16; }
__device__ __forceinline__ void w_skip_d(wrd_d &r, uint32_t n) { r.T += n; }

__device__ __forceinline__ uint32_t pad_words_d(uint64_t T, uint32_t last)      // QV.c:436-442
{ const uint32_t olen = (uint32_t) T & 31u, llen = (uint32_t) (T - last) & 31u;
  const uint32_t w = (uint32_t) (T >> 5) + (olen ? 1u : 0u);
  if (olen > 0) return w + ((llen > 16u && olen > llen) ? 1u : 0u);
  return w + ((T > 0 && llen > 16u) ? 1u : 0u);
}

// LDS: the tables of dx_walk.h a walk looks into -- t[0..3] the line's own (del, ins, mrg, sub: the pair table of a run-coded
// line, else the several-codes table), r1 the run codes alone (del, sub), one the first code of a window alone (the four symbol
// schemes); 80 KB.  All read alike: bits to skip | the last code's length << 4 | symbols covered << 8, 0: not this way.
struct walk_lds { uint16_t t[4][4096]; uint16_t r1[2][4096]; uint16_t one[4][4096]; uint32_t ring[WALK_BLOCK][WRING_STRIDE]; };

#define PH_HEAD 0u              // at a record's first byte
#define PH_DEL  1u
#define PH_INS  2u
#define PH_MRG  3u
#define PH_SUB  4u
#define PH_DONE 5u              // behind the record's last segment
#ifndef WALK_BRANCHY
#define WALK_BRANCHY 0
#endif
#ifndef WALK_BURST
#ifndef WALK_SUB
#define WALK_SUB 4              // bursts between two looks at the lanes that need something else (an even number: the pumps' two sets)
#endif
#define WALK_BURST 8            // look-ups in a row before the lanes that need something else are seen to (and the rings: w_pump_d's
#endif                          // arithmetic is for at most 8)

// One lane per piece (todo == NULL: piece = thread index; else the listed pieces, each from start[piece], which the chain has
// arrived at): the piece's records into recs[piece * rcap ...], what became of it into pc[piece].
//
// The loop: a burst of up to WALK_BURST look-ups of the common kind -- the window holds whole codes (a whole run-and-symbol
// pair) that stay inside the line: skip their bits, count their symbols; the same few instructions whichever line the lane
// is in -- then, for the lanes the burst left waiting, one step of the other kinds: a single code (a line's last few, a pair
// that does not fit the window), a code of more than 12 bits or an escape (from the 16-bit table in memory), a run's literal,
// a segment's end, a record's header, a record's end.  (Measured: the single-code steps inside the burst as well, the lane's
// mode choosing among three tables of one form, cost the burst more than the trips outside save: 119 against 106 ms.)
__global__ __launch_bounds__(WALK_BLOCK, WALK_WGS_PER_CU * WALK_BLOCK / 256)     // (waves per SIMD: two workgroups a CU)
void k_walk_pieces(walk_args a, uint64_t *cand, const uint32_t *ncand, const uint32_t *todo, uint32_t ntodo,
                   const uint64_t *start, walk_piece_d *pc, walk_rec_d *recs, uint32_t rcap, uint32_t *gwords, uint32_t gcap,
                   uint32_t *queue /* the next piece nobody has taken (starts at the number of lanes launched); NULL: a lane, a piece */)
{ __shared__ walk_lds S;
#define WALK_STAGE(tab, from, words) { const uint32_t *s_ = (const uint32_t *) (const void *) (from); uint32_t *d_ = (uint32_t *) (void *) (tab); \
                                       for (uint32_t i = threadIdx.x; i < (words); i += blockDim.x) d_[i] = s_[i]; }
  WALK_STAGE(S.t[0], a.delChar < 0 ? a.mw + DX_DEL * 4096u : a.rw, 2048u)
  WALK_STAGE(S.t[1], a.mw + DX_INS * 4096u, 2048u)
  WALK_STAGE(S.t[2], a.mw + DX_MRG * 4096u, 2048u)
  WALK_STAGE(S.t[3], a.subChar < 0 ? a.mw + DX_SUB * 4096u : a.rw + 4096u, 2048u)
  WALK_STAGE(S.r1, a.r1, 4096u)
  WALK_STAGE(S.one, a.one, 8192u)
#undef WALK_STAGE
  __syncthreads();
  // A lane takes a piece, walks it, and takes the next one nobody has taken (queue): pieces are smaller than a lane's share of the
  // stream, so that a lane with a long record in its piece -- it walks the record to its end, wherever that is -- does not leave
  // the others waiting: with a piece a lane the kernel took as long as its longest lane (lognormal lengths: 45 ms for work of 24).
  const uint64_t idx = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t k, hi = 0;
  bool have_piece;
  if (todo) { have_piece = idx < ntodo; k = have_piece ? todo[idx] : 0; }
  else      { have_piece = idx < a.pieces; k = have_piece ? idx : 0; }
  walk_rec_d *my = recs;
  walk_piece_d out = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  uint32_t rounds_ = 0;

  // the lane's state
  uint64_t at;                         // PH_HEAD: the record's first byte; in a segment: the segment's first byte
  uint32_t ph = PH_HEAD, rlen = 0, j = 0, last = 0, nn = 0, clen = 0;
  uint32_t reclusters = 0;
  // What the lane notes for the group index on its way (dx_layout.h), into its share of gwords (gcap words; a line the lane has no
  // room for, or that does not fit the format, is left without: WG_NONE, a lane-per-line decoder takes it):
  //  * a run-coded line (what k_qv_decode_runs wants to decode it a wavefront at a time): a word per 8 tokens, the bits they take |
  //    the positions they cover << 16.  gT / gj: where the open group began; gP: where the open pass of 64 groups began;
  //  * a plain line without escape codes (k_qv_decode_sync): a word per 64 symbols but the first 64 -- the bits passed where the
  //    look-up began that reached symbol 64 g | the symbols from there to 64 g << 28 (<= 12: a look-up's codes).
  // gi: the next word; gline: where the line's words began.  A burst holds at most one of either (8 look-ups: 8 tokens; <= 96
  // symbols, and a line whose burst does pass two marks -- codes of a bit -- is left without).
  uint32_t *myg = NULL;
  uint32_t gi = 0, gT = 0, gj = 0, gP = 0, gline = 0, ghead = 0;
  bool gbad = false, trailing = false;
#define WALK_GROUP_OUT(Tx, jx) \
  { const uint32_t b_ = (Tx) - gT, s_ = (jx) - gj; \
    if (b_ > 0xffffu || s_ > 0xffffu || (gi | 3u) >= gcap) gbad = true; \
    if ((gi | 3u) < gcap) WALK_STORE(b_ | (s_ << 16))       /* (a word of a line that is left without is stored all the same: the four go together) */ \
    gi += 1u; gT = (Tx); gj = (jx); \
    if (((gi - gline) & 63u) == 0u) { if ((Tx) - gP > DXL_RUN_PASSBITS) gbad = true; gP = (Tx); } \
  }
  /* Bx: the multiple of 64 reached; Tx: the bits passed dx symbols in front of it */ \
#define WALK_SYNC_AT(Tx, dx, Bx) \
  { if ((Bx) < rlen) \
      { if ((Tx) >= (1u << 28) || (dx) > 15u || (gi | 3u) >= gcap || (Bx) != 64u * (gi - gline + 1u)) gbad = true; \
        if ((gi | 3u) < gcap) WALK_STORE((Tx) | ((dx) << 28)) \
        gi += 1u; \
      } \
  }
  /* (Tx, jx): where the look-up began that passed a multiple of 64 */ \
#define WALK_SYNC_OUT(Tx, jx) { const uint32_t B_ = ((jx) | 63u) + 1u; WALK_SYNC_AT(Tx, B_ - (jx), B_) }
  uint32_t trial = 0, ci = 0, nc = 0;  // trial: the record being walked is a guess (candidate ci of the piece's nc)
  uint64_t budget = ~0ull;
  // the record being walked goes to its place in the lane's records field by field as it becomes known (a record that does not
  // hold -- a guess -- is written over by the next; only what the lane adds up over its records stays in registers)
  uint64_t r_off = 0;
  uint32_t r_hdr = 0;
  int32_t  r_dwell = 0;
#define R_SLOT (my + (out.count < rcap ? out.count : rcap - 1u))      /* (more records than rcap: WP_OVERFLOW, the walk is the host's) */
  // the words for the group index: four at a time to the lane's share (a lane's words one by one are 700 M four-byte writes to
  // 250 000 different lines: 10 ms of a 32 ms kernel)
  u32x4 gbuf = { 0u, 0u, 0u, 0u };
#define WALK_STORE(word) \
  { const uint32_t q_ = gi & 3u; \
    gbuf.x = q_ == 0u ? (word) : gbuf.x; gbuf.y = q_ == 1u ? (word) : gbuf.y; gbuf.z = q_ == 2u ? (word) : gbuf.z; gbuf.w = q_ == 3u ? (word) : gbuf.w; \
    if (q_ == 3u) *(u32x4 *) (myg + (gi - 3u)) = gbuf; \
  }
  /* what is left in the buffer, behind the lane's last record */ \
#define WALK_STORE_FLUSH() \
  { if (myg != NULL && (gi & 3u) && (gi | 3u) < gcap) *(u32x4 *) (myg + (gi & ~3u)) = gbuf; }
  wrd_d rd;
  bool live = false, fresh = have_piece;                      // fresh: a piece has been taken and not begun
  rd.seg = a.img; rd.ring = S.ring[threadIdx.x]; rd.T = 0; rd.have = 0; rd.ua = 0; rd.ub = 0;
  rd.a0 = u32x4{ 0u, 0u, 0u, 0u }; rd.a1 = rd.a0; rd.b0 = rd.a0; rd.b1 = rd.a0;
  at = 0;

  // (a lane that is done stays in the loop, doing nothing, until its wave is: w_pump_d and the burst are the wave's)
  for (uint32_t turn = 0; ; turn++)
    { // (pieces end and begin every WALK_SERVE-th turn only: what that takes -- the piece's verdict stored, the queue, the next
      //  piece's guesses fetched -- is memory round trips for the whole wave, and with 64 lanes some lane is between pieces in
      //  nearly every turn; a lane waits two turns on average, of the 150 a piece takes)
      const bool serve = (turn & (WALK_SERVE - 1u)) == 0u;
      if (serve && !live && !fresh && have_piece)               // the lane's piece is walked: what became of it, and the next piece
        { out.tried = ci;
          if (!(out.flags & WP_NONE) && out.count)               // (see k_walk_find: bytes of 255 in front of the start may be the header's)
            { uint32_t c = 0;
              while (c < 4096u && out.start > a.first + c && a.img[out.start - 1 - c] == 255) c++;
              out.lead255 = c;
            }
          WALK_STORE_FLUSH()
          pc[k] = out;
          have_piece = false;
          if (queue != NULL)
            { const uint64_t nk = atomicAdd(queue, 1u);
              if (nk < a.pieces) { k = nk; have_piece = true; fresh = true; }
            }
        }
      if (serve && fresh)                                       // begin the piece
        { fresh = false; live = true;
          hi  = k == a.pieces - 1 ? a.n : a.first + (k + 1) * a.piece;
          my  = recs + k * (uint64_t) rcap;
          myg = gwords ? gwords + k * (uint64_t) gcap : (uint32_t *) NULL;
          out = walk_piece_d{ 0, 0, 0, 0, 0, 0, 0, 0, 0 };
          ph = PH_HEAD; rlen = 0; j = 0; last = 0; nn = 0; clen = 0; reclusters = 0;
          gi = 0; gT = 0; gj = 0; gP = 0; gline = 0; ghead = 0; gbad = false; trailing = false;
          trial = 0; ci = 0; nc = 0; budget = ~0ull;
          if (todo || k == 0)
            { out.start = todo ? start[k] : a.first; at = out.start; }
          else if (ncand[k] == 0)
            { out.flags = WP_NONE; live = false; at = 0; }
          else
            { trial = 1; budget = 16u * a.piece; at = cand[k * WALK_CAND]; nc = ncand[k]; }          // (two pieces' bytes)
          if (live && !trial && !(at < hi && at < a.n)) { live = false; out.end = at; }     // (nothing of this piece left to walk)
        }
      if (!__any(live || have_piece)) break;
      bool fail = false;
      rounds_ += 1;
      const bool inseg = live && ph >= PH_DEL && ph < PH_DONE;
      const uint32_t line = ph - PH_DEL;                        // 0 del, 1 ins, 2 mrg, 3 sub (in a segment)
      bool more = inseg;
      // WALK_SUB bursts, the rings seen to in front of each (the wave together: lanes outside a segment, or left behind by a
      // burst, ask for nothing and pass nothing).  Without branches: a lane left behind shifts by nothing and counts nothing.
      const uint16_t *tab = S.t[inseg ? line : 0u];
      const bool runs_ = inseg && myg != NULL && ((line == 0 && a.delChar >= 0) || (line == 3 && a.subChar >= 0));
      const bool sync_ = inseg && myg != NULL && !runs_ && !a.esc[inseg ? line : 0u];
      // The burst, written so that nothing of it is decided by the scalar unit (one to a CU: a look-up whose conditions are
      // and-ed and or-ed as lane masks spends more scalar than vector instructions): a lane may pass `room` more symbols (none:
      // the burst without it); a look-up that is none (cnt = 0) or would pass more changes nothing, and neither does any after
      // it in the burst -- the lane stands where it stood --, so no lane needs to be told that it has stopped.  A mark of the
      // group index (the 8th token of a group, x = the tokens; a multiple of 64 symbols, x = the symbols) is passed when x
      // changes above bit gsh; where, is kept: behind the look-up (a group's end), in front of it (gpm: the plain lines' words).
      const uint32_t gsh = runs_ ? 3u : sync_ ? 6u : 31u, gpm = runs_ ? 0u : ~0u;
#define WALK_BURST_ONCE(SET) \
        { w_pump_d<SET>(rd, a, more); \
          const bool ready = w_words_d(rd) >= WRING_BURST;      /* (a ring that is short: this burst without the lane) */ \
          uint32_t room = more && ready ? rlen - j : 0u, fT = 0u, fj = 0u, marks = 0u, x0 = runs_ ? nn : j; \
          bool took = false; \
          _Pragma("unroll 1") \
          for (int it = 0; it < WALK_BURST; it++) \
            { const uint32_t g = tab[w_win_d(rd) >> (32 - WALK_WIN)], cnt = g >> 8; \
              took = cnt - 1u < room; \
              const uint32_t nb = took ? g & 15u : 0u, c = took ? cnt : 0u; \
              rd.T += nb; j += c; room -= c; nn += took ? 1u : 0u; last = took ? (g >> 4) & 15u : last; \
              const uint32_t x1 = runs_ ? nn : j; \
              const bool mark = ((x0 ^ x1) >> gsh) != 0u; \
              marks += mark ? 1u : 0u; \
              fT = mark ? rd.T - (nb & gpm) : fT; fj = mark ? j - (c & gpm) : fj; \
              x0 = x1; \
              if (!__any(took)) break; \
            } \
          if (marks) \
            { if (marks > 1u) gbad = true; \
              if (runs_) WALK_GROUP_OUT(fT, fj) else WALK_SYNC_OUT(fT, fj) \
            } \
          more = more && (took || !ready); \
        }
      #pragma unroll 1
      for (int sub = 0; sub < WALK_SUB; sub += 2)
        { WALK_BURST_ONCE(0)
          WALK_BURST_ONCE(1)
          if (!__any(more)) break;
        }
#undef WALK_BURST_ONCE
      if (inseg && !more && j < rlen && w_words_d(rd) < WRING_STEP) more = true;     // (the step waits for the ring too)
      if (inseg)
        {
          if ((uint64_t) rd.T > budget)        // a guess that walks on and on (garbage may claim any length, and zeros
            { fail = true; more = true; }                       // behind the image read as codes): every burst, whatever became of it
          if (!more)
            { const bool runs = (line == 0 && a.delChar >= 0) || (line == 3 && a.subChar >= 0);
              if (j < rlen)                                     // one step of another kind
                { uint32_t w = w_peek_d(rd);
                  bool sym = true;                              // a symbol's code is to be passed
                  const uint32_t T0 = rd.T, j0 = j;
                  if (runs)                                     // walk_runs: the run code alone first
                    { const uint32_t e1 = S.r1[line ? 1u : 0u][w >> (16 - WALK_WIN)];
                      uint32_t c;
                      if (e1) { last = e1 & 15u; c = e1 >> 8; w_skip_d(rd, last); }
                      else                                      // more than 12 bits, the code of 255 (a 16-bit literal follows), none
                        { const uint32_t e = a.w16[(DX_DRUN + (line ? 1u : 0u)) * 65536u + w];
                          last = e >> 8; c = e & 0xffu;
                          if (last == 0) fail = true;
                          w_skip_d(rd, last);
                          if (c == 255u)
                            { c = w_peek_d(rd); w_skip_d(rd, 16u); last = 16; }
                        }
                      if (c > rlen - j) fail = true;
                      j  += c;
                      sym = !fail && j < rlen;
                      if (sym) w = w_peek_d(rd);
                      else if (runs_ && (nn & 7u))              // the line ends in a run: no token, and none of the open group's bits
                        { WALK_GROUP_OUT(T0, j0) trailing = true; }
                    }
                  if (sym)                                      // walk_plain's single code; walk_runs' symbol behind the run
                    { const uint32_t f = S.one[line][w >> (16 - WALK_WIN)] & 15u;
                      if (f) { last = f; w_skip_d(rd, f); }
                      else                                      // more than 12 bits, an escape, no code at all
                        { const uint32_t e = a.w16[line * 65536u + w];
                          last = e >> 8;
                          if (last == 0) fail = true;
                          w_skip_d(rd, last);
                          if (a.esc[line] && (e & 0xffu) == 255u)
                            { w_skip_d(rd, 8u); last = 8; }
                        }
                      j  += 1;
                      nn += 1;
                      if (runs_ && (nn & 7u) == 0u) WALK_GROUP_OUT(rd.T, j)
                      if (sync_ && (j & 63u) == 0u) WALK_SYNC_AT(rd.T, 0u, j)
                    }
                }
              const uint64_t T = rd.T;                                     // bits of the segment passed
              if (T > budget) fail = true;
              if (!fail && j >= rlen)                           // the segment's end: its bytes, and on to the next one
                { const uint64_t bytes = 4ull * pad_words_d(T, last);
                  if (at + bytes > a.n) fail = true;
                  else
                    { walk_rec_d *rs = R_SLOT;
                      rs->seg[line ? line + 1u : 0u] = (uint32_t) bytes;
                      at += bytes;
                      // what the lane has noted of the line for the group index: where it is, how much of it
                      if (runs_ && (nn & 7u) && !trailing) WALK_GROUP_OUT(rd.T, j)
                      { const uint32_t gt = gbad ? WG_NONE : runs_ ? nn : sync_ && gi - gline + 1u == ((rlen + 63u) >> 6) ? gi - gline : WG_NONE;
                        rs->gat[line] = gline; rs->gtok[line] = gt;
                      }
                      if (line == 0)                            // the tags (Pack_Tag's count, QV.c:810-819): no codes, just bytes
                        { if (runs) clen = nn;
                          const uint32_t tb = (clen + 3u) >> 2;
                          rs->seg[1] = tb;
                          if (at + tb > a.n) fail = true;
                          at += tb;
                        }
                      ph += 1; j = 0; last = 0; nn = 0;
                      gT = 0; gj = 0; gP = 0; gline = gi; gbad = false; trailing = false;
                      if (ph < PH_DONE && !fail) w_open_d(rd, a, a.img + at);
                    }
                }
            }
        }
      else if (!live)
        { }
      else if (ph == PH_HEAD)                                   // walk_framing of dx_host.c (0x55aa-keyed: 32-bit fields)
        { const uint64_t h0 = at;
          int32_t dw = 0;
          while (at < a.n && a.img[at] == 255 && at - h0 < WALK_LEAD_MAX) { dw += 255; at += 1; }
          if (at + 13 > a.n || at - h0 >= WALK_LEAD_MAX) fail = true;
          else
            { dw += a.img[at];
              const int32_t beg = (int32_t) bswap_if(load32_at(a.img + at + 1), a.flip), end_ = (int32_t) bswap_if(load32_at(a.img + at + 5), a.flip);
              const int32_t qv  = (int32_t) bswap_if(load32_at(a.img + at + 9), a.flip);
              at += 13;
              rlen = (uint32_t) ((int64_t) end_ - (int64_t) beg);
              if (end_ < beg || (int64_t) end_ - (int64_t) beg > WALK_RLEN_MAX || (uint64_t) rlen > 65536u * 8u * (uint64_t) (a.n - at) + 64u)
                fail = true;
              r_off = h0; r_hdr = (uint32_t) (at - h0); r_dwell = dw;
              { walk_rec_d *rs = R_SLOT;
                rs->off = h0; rs->hdr_bytes = r_hdr; rs->len = rlen; rs->dwell = dw; rs->beg = beg; rs->end = end_; rs->qv = qv; rs->pad = 0;
              }
              if (trial)                                        // (see WALK_TRIAL_RLEN)
                { if (rlen > (uint32_t) WALK_TRIAL_RLEN) fail = true;
                  budget = 32ull * rlen + 8192u < 16u * a.piece ? 32ull * rlen + 8192u : 16u * a.piece;
                }
              ph = PH_DEL; j = 0; last = 0; nn = 0; clen = rlen;
              ghead = gi; gT = 0; gj = 0; gP = 0; gline = gi; gbad = false; trailing = false;
              if (!fail) w_open_d(rd, a, a.img + at);
            }
        }
      else                                                      // PH_DONE: `at` is behind the record
        { if (trial)
            { if (at == a.n || header_plausible_d(a, at))       // the guess holds: the lane's first record
                { out.start = r_off; trial = 0; budget = ~0ull; }
              else
                fail = true;
            }
          if (!fail)
            { if (out.count >= rcap) out.flags |= WP_OVERFLOW;
              if (out.count == 0) out.first_hdr = r_hdr;
              out.count += 1; out.hdr_sum += r_hdr; out.dwell_sum += (uint64_t) (uint32_t) r_dwell;
              ph = PH_HEAD;
              if (!(at < hi && at < a.n)) { out.end = at; live = false; }
            }
        }
      if (fail)
        { if (trial)                                            // a wrong guess: the piece's next one
            { ci += 1;
              gi = ghead;                                       // (its groups are nobody's)
              if (ci < nc) { at = cand[k * WALK_CAND + ci]; ph = PH_HEAD; }
              else                                              // none of the cluster held (rare): the next cluster, found by the lane itself
                { uint64_t *c = cand + k * WALK_CAND, last = 0;   // (through memory: no registers for what happens once in ten thousand pieces)
                  uint32_t m = 0;
                  reclusters += 1;                              // (WALK_RECLUSTER of them at most: a crafted piece may hold thousands)
                  #pragma unroll 1
                  for (uint64_t p = c[0] + 1; reclusters <= WALK_RECLUSTER && p < hi && (m == 0 || p <= last + 16u) && m < WALK_CAND; p++)
                    if (a.img[p] != 255 && header_plausible_d(a, p, WALK_TRIAL_RLEN)) { c[m++] = p; last = p; }
                  if (m == 0) { out.flags = WP_NONE; live = false; }
                  else
                    { for (uint32_t x = 0, y = m - 1; x < y; x++, y--) { const uint64_t t = c[x]; c[x] = c[y]; c[y] = t; }   // (last first)
                      nc = m; ci = 0; at = c[0]; ph = PH_HEAD;
                    }
                }
            }
          else                                                  // a record on the lane's chain does not walk
            { out.flags |= WP_BAD; out.end = at; live = false; }
        }
    }
  (void) rounds_;
}

// the records of the pieces on the chain, side by side: a wave per piece; dst[k] = its first record's index (~0: not on the
// chain), hbase[k] / wbase[k] = the framing bytes / wells before it
// (trim[k], signed: bytes of 255 in front of the piece's first record that are its header's (< 0) -- see the chain)
__global__ __launch_bounds__(DX_BLOCK)
void k_walk_gather(uint64_t pieces, const walk_piece_d *pc, const walk_rec_d *recs, uint32_t rcap, const uint64_t *dst,
                   const uint64_t *hbase, const uint64_t *wbase, const uint64_t *trim,
                   uint64_t *rec_off, uint64_t *hdr_off, uint32_t *seg, uint32_t *len, int32_t *hdr4,
                   uint32_t gcap, uint64_t *gsrc /* 4 a record: where its lines' words lie in the lanes' group words */, uint32_t *gtok)
{ const uint32_t lane = (uint32_t) lane_id();
  const uint64_t k = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  if (k >= pieces || dst[k] == ~0ull) return;
  const walk_rec_d *my = recs + k * (uint64_t) rcap;
  const uint32_t cnt = pc[k].count;
  uint64_t hb = hbase[k], wb = wbase[k];
  for (uint32_t i0 = 0; i0 < cnt; i0 += 64u)
    { const uint32_t i = i0 + lane;
      walk_rec_d r;
      uint32_t h = 0, d = 0;
      if (i < cnt)
        { r = my[i];
          if (i == 0) { const int64_t t = (int64_t) trim[k]; r.off = (uint64_t) ((int64_t) r.off + t); r.hdr_bytes = (uint32_t) ((int64_t) r.hdr_bytes - t); r.dwell -= 255 * (int32_t) t; }
          h = r.hdr_bytes; d = (uint32_t) r.dwell;
        }
      const uint32_t hi_ = wave_incl_scan(h), di_ = wave_incl_scan(d);
      if (i < cnt)
        { const uint64_t o = dst[k] + i;
          rec_off[o] = r.off;
          hdr_off[o] = hb + hi_ - h;
          len[o]     = r.len;
          seg[5 * o] = r.seg[0]; seg[5 * o + 1] = r.seg[1]; seg[5 * o + 2] = r.seg[2]; seg[5 * o + 3] = r.seg[3]; seg[5 * o + 4] = r.seg[4];
          hdr4[4 * o] = (int32_t) (wb + di_); hdr4[4 * o + 1] = r.beg; hdr4[4 * o + 2] = r.end; hdr4[4 * o + 3] = r.qv;
          if (gsrc)
            for (int x = 0; x < 4; x++)
              { gsrc[4 * o + x] = k * (uint64_t) gcap + r.gat[x];
                gtok[4 * o + x] = r.gtok[x];
              }
        }
      hb += wave_total(hi_); wb += wave_total(di_);
    }
}

// ---------------------------------------------------------------------------------------------
//  the run-coded lines' share of the group index (dx_layout.h), from the lanes' group words
// ---------------------------------------------------------------------------------------------
// An entry's share: room for the four plain lines' bytes (left as they are: nobody reads them, DXL_RUNS_ONLY), three header words
// -- the deletion line's tokens or DXL_RUN_NONE, the substitution line's, the deletion line's passes | DXL_RUN_EIGHTS --, then 64
// words per pass of 512 tokens of either line.  The lanes noted a word per 8 tokens, the last group of a line whatever was
// left: pass p's lane l decodes tokens 512 p + 8 l ... (k_qv_encode_fast deals the last pass's m tokens (m + 63) / 64 a lane:
// DXL_RUN_EIGHTS says which).
__global__ __launch_bounds__(DX_BLOCK)
void k_walk_rooms(const uint32_t *len, const uint32_t *gtok, uint64_t n, int del_runs, int sub_runs, uint32_t *room)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t d = del_runs ? gtok[4 * i] : WG_NONE, s_ = sub_runs ? gtok[4 * i + 3] : WG_NONE;
  room[i] = dxl_run_base(len[i]) + 3u + 64u * ((d == WG_NONE ? 0u : dxl_run_passes(d)) + (s_ == WG_NONE ? 0u : dxl_run_passes(s_)));
}

// a wave per entry: the header words, the run-coded lines' groups (zeros behind a line's last, to the pass's end), and the plain
// lines' words: word g of a line's share = what the lane noted for symbol 64 g (word 0: 0, or DXL_SYNC_NONE: no words for this line)
__global__ __launch_bounds__(DX_BLOCK)
void k_walk_index(const uint32_t *len, const uint32_t *gtok, const uint64_t *gsrc, const uint32_t *gwords, uint64_t n,
                  const uint64_t *goff, uint32_t *gidx, int del_runs, int sub_runs, uint32_t sync_kinds, uint32_t *none /* [0] run-coded, [1] plain lines without */)
{ const uint32_t lane = (uint32_t) lane_id();
  const uint64_t i = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint32_t L = len[i];
  const uint32_t d = del_runs ? gtok[4 * i] : WG_NONE, s_ = sub_runs ? gtok[4 * i + 3] : WG_NONE;
  const uint32_t pd = d == WG_NONE ? 0u : dxl_run_passes(d), ps = s_ == WG_NONE ? 0u : dxl_run_passes(s_);
  uint32_t *share = gidx + goff[i], *base = share + dxl_run_base(L);
  if (lane == 0)
    { base[0] = d == WG_NONE ? DXL_RUN_NONE : d;
      base[1] = s_ == WG_NONE ? DXL_RUN_NONE : s_;
      base[2] = pd | DXL_RUN_EIGHTS;
      const uint32_t lost = (del_runs && d == WG_NONE ? 1u : 0u) + (sub_runs && s_ == WG_NONE ? 1u : 0u);
      if (lost) atomicAdd(none, lost);
    }
  for (int x = 0; x < 4; x += 3)
    { const uint32_t tokens = x ? s_ : d, passes = x ? ps : pd;
      if (tokens == WG_NONE) continue;
      const uint32_t groups = (tokens + 7u) >> 3;
      const uint32_t *from = gwords + gsrc[4 * i + x];
      uint32_t *to = base + 3u + (x ? 64u * pd : 0u);
      for (uint32_t g = lane; g < 64u * passes; g += 64u) to[g] = g < groups ? from[g] : 0u;
    }
  const uint32_t sw = dxl_sub_words(L);
  for (uint32_t q = 0; q < 4; q++)
    if (((sync_kinds >> q) & 1u) && sw)
      { const uint32_t cnt = gtok[4 * i + q];
        const bool ok = cnt != WG_NONE && cnt + 1u == sw;
        const uint32_t *from = gwords + gsrc[4 * i + q];
        uint32_t *to = share + (uint64_t) q * sw;
        if (lane == 0) { to[0] = ok ? 0u : DXL_SYNC_NONE; if (!ok) atomicAdd(none + 1, 1u); }
        if (ok) for (uint32_t g = 1u + lane; g < sw; g += 64u) to[g] = from[g - 1u];
      }
}

// ---------------------------------------------------------------------------------------------
//  host
// ---------------------------------------------------------------------------------------------
void dx_qv_dindex_free(dx_ctx *ctx, dx_qv_dindex *x)
{ if (x == NULL) return;
  (void) ctx;
  (void) hipFree(x->d_rec_off); (void) hipFree(x->d_hdr_off); (void) hipFree(x->d_seg); (void) hipFree(x->d_len); (void) hipFree(x->d_hdr4);
  (void) hipFree(x->d_gidx); (void) hipFree(x->d_gidx_off);
  memset(x, 0, sizeof(*x));
}

int dx_qv_walk_device(dx_ctx *ctx, const uint8_t *d_img, uint64_t n, uint64_t first, const dx_qv_coding *cd, int newv, int flip,
                      dx_qv_dindex *out)
{ if (ctx == NULL || d_img == NULL || cd == NULL || out == NULL || first > n) return DX_E_ARG;
  memset(out, 0, sizeof(*out));
  if (!newv)                       // 16-bit framing fields are too easily plausible (dx_host.c): the host walk's
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_walk_device: a stream with 16-bit framing fields is walked on the host");
  DX_HIP(ctx, hipSetDevice(ctx->device));         // (the scratch and the index go where the context's stream runs)
  int rc = dx_after_pending(ctx);
  if (rc != DX_OK) return rc;

  walk_args a;
  memset(&a, 0, sizeof(a));
  a.img = d_img; a.n = n; a.first = first; a.delChar = cd->delChar; a.subChar = cd->subChar; a.flip = flip;
  uint64_t lanes;
  { // As many lanes as the device holds at once (a workgroup a CU), WALK_PER_LANE pieces a lane on average: a lane takes the next piece
    // nobody has taken when it is done with its own (k_walk_pieces), so what a long record costs its lane the others make up for.
    // (Smaller pieces cost k_walk_find: every piece is scanned to its first header, half a record.)
    uint64_t per_lane = WALK_PER_LANE;
    { const uint64_t v = (uint64_t) dx_test_num("walk_per_lane", 0); if (v >= 1 && v <= 64) per_lane = v; }
    lanes = (uint64_t) ctx->num_cu * WALK_WGS_PER_CU * WALK_BLOCK < WALK_LANES_MAX ? (uint64_t) ctx->num_cu * WALK_WGS_PER_CU * WALK_BLOCK : WALK_LANES_MAX;
    uint64_t piece = (n - first + lanes * per_lane - 1) / (lanes * per_lane);
    if (piece < WALK_PIECE_MIN) piece = WALK_PIECE_MIN;
    { const uint64_t v = (uint64_t) dx_test_num("walk_piece", 0); if (v >= 4096) piece = v; }   // (tests)
    piece = (piece + 4095u) & ~(uint64_t) 4095u;
    a.piece = piece;
    a.pieces = n > first ? (n - first + piece - 1) / piece : 1;
  }
  const uint64_t P = a.pieces;
  const uint32_t rcap = (uint32_t) (a.piece / 128u);
  // the lanes' words for the group index: 0.05 words a byte at the bench's 690 words a 14.4 KB record; a share of piece / 8 + 4096 words; lines
  // of a lane that runs out are left without (DEXGPU_WALK_NOGROUPS: none at all -- the index is then the round-4 one)
  const bool     groups = !flip && !dx_test_on("walk_nogroups");
  const uint32_t gcap = groups ? (uint32_t) (a.piece / 8u) + 4096u : 0u;     // (+ what the record takes that the lane walks beyond its piece: 80 KB's worth)
  { // does the scratch fit (56 bytes a possible record, 0.44 of the stream) with room for the index behind it?  Asked first: a
    // failed allocation half way costs the allocations before it, and the caller has another way (the host walk).
    uint64_t fr = 0, all = 0;
    const double need = (double) P * ((double) rcap * sizeof(walk_rec_d) + 4.0 * gcap + sizeof(walk_piece_d) + WALK_CAND * 8 + 48) + 0.2 * (double) (n - first) + (64 << 20);
    if (dx_mem_info(ctx, &fr, &all) == DX_OK && fr > 0 && need > (double) fr)
      return dx_fail(ctx, DX_E_NOMEM, "dx_qv_walk_device: %.1f GB of scratch do not fit the device's free %.1f GB", need / 1e9, (double) fr / 1e9);
  }

  uint8_t  *blob = (uint8_t *) malloc(WALK_BLOB_BYTES);
  uint8_t  *d_blob = NULL, *d_tail = NULL;
  uint64_t *d_cand = NULL, *d_start = NULL, *d_dst = NULL;
  uint32_t *d_ncand = NULL, *d_todo = NULL;
  walk_piece_d *d_pc = NULL, *pc = (walk_piece_d *) malloc(P * sizeof(walk_piece_d));
  walk_rec_d   *d_recs = NULL;
  uint32_t *d_gwords = NULL, *d_gtok = NULL, *d_room = NULL, *d_none = NULL, *d_queue = NULL;
  uint64_t *d_gsrc = NULL;
  uint64_t *dst = (uint64_t *) malloc(4 * P * 8), *trim = dst ? dst + 3 * P : NULL;
  uint8_t  *onchain = (uint8_t *) calloc(P, 1);
  uint64_t N = 0;
#define WALK_FAIL(code, ...) do { rc = dx_fail(ctx, code, __VA_ARGS__); goto done; } while (0)
#define WALK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void) hipGetLastError(); WALK_FAIL(DX_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } } while (0)
  if (!blob || !pc || !dst || !onchain) WALK_FAIL(DX_E_NOMEM, "dx_qv_walk_device: out of host memory");
  memset(dst, 0, 4 * P * 8);
  rc = dx_walk_luts_build(cd, blob, a.esc);
  if (rc != DX_OK) WALK_FAIL(rc, "dx_qv_walk_device: the look-up tables could not be built");
  WALK_HIP(hipMalloc(&d_blob, WALK_BLOB_BYTES));
  WALK_HIP(hipMalloc(&d_cand, P * WALK_CAND * 8));
  WALK_HIP(hipMalloc(&d_ncand, P * 4));
  WALK_HIP(hipMalloc(&d_pc, P * sizeof(walk_piece_d)));
  WALK_HIP(hipMalloc(&d_recs, P * (uint64_t) rcap * sizeof(walk_rec_d)));
  if (groups) WALK_HIP(hipMalloc(&d_gwords, P * (uint64_t) gcap * 4u));
  WALK_HIP(hipMemcpyAsync(d_blob, blob, WALK_BLOB_BYTES, hipMemcpyHostToDevice, ctx->stream));
  a.w16 = (const uint16_t *) (d_blob + WALK_W16_OFF); a.mw = (const uint16_t *) (d_blob + WALK_MW_OFF);
  a.rw  = (const uint16_t *) (d_blob + WALK_RW_OFF);  a.r1 = (const uint16_t *) (d_blob + WALK_R1_OFF);
  a.one = (const uint16_t *) (d_blob + WALK_ONE_OFF);
  a.tail_at = n >= 64 ? (n - 64) & ~(uint64_t) 15 : 0;
  WALK_HIP(hipMalloc(&d_tail, 256));
  WALK_HIP(hipMemsetAsync(d_tail, 0, 256, ctx->stream));
  if (n > a.tail_at) WALK_HIP(hipMemcpyAsync(d_tail, d_img + a.tail_at, n - a.tail_at, hipMemcpyDeviceToDevice, ctx->stream));
  a.tail = d_tail;

  WALK_HIP(hipMalloc(&d_queue, 64));
  { // a workgroup per CU while the lanes allow, every CU busy; the pieces beyond the lanes launched wait in the queue
    const uint64_t L0 = P < lanes ? P : lanes;
    const uint64_t waves = (L0 + 63u) / 64u, per_cu = (waves + ctx->num_cu - 1) / ctx->num_cu;
    const uint32_t bs = per_cu >= WALK_BLOCK / 64u ? WALK_BLOCK : (uint32_t) (per_cu ? per_cu * 64u : 64u);
    const uint32_t q0 = (uint32_t) (((L0 + bs - 1) / bs) * bs);             // (the lanes launched take pieces 0 .. q0 - 1 themselves)
    WALK_HIP(hipMemcpyAsync(d_queue, &q0, 4, hipMemcpyHostToDevice, ctx->stream));
    hipEvent_t ev[3] = { NULL, NULL, NULL };                   // DEXGPU_WALK_DEBUG: the two kernels' times
    const bool timed = dx_test_on("walk_debug") && hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess && hipEventCreate(&ev[2]) == hipSuccess;
    dx_prof_begin(ctx, DX_K_QV_WALK);
    if (timed) (void) hipEventRecord(ev[0], ctx->stream);
    hipLaunchKernelGGL(k_walk_find, dim3((unsigned) ((P + DX_WAVES_PER_BLK - 1) / DX_WAVES_PER_BLK)), dim3(DX_BLOCK), 0, ctx->stream, a, d_cand, d_ncand);
    if (timed) (void) hipEventRecord(ev[1], ctx->stream);
    hipLaunchKernelGGL(k_walk_pieces, dim3((unsigned) ((L0 + bs - 1) / bs)), dim3(bs), 0, ctx->stream, a, d_cand,
                       (const uint32_t *) d_ncand, (const uint32_t *) NULL, 0u, (const uint64_t *) NULL, d_pc, d_recs, rcap, d_gwords, gcap, d_queue);
    if (timed) (void) hipEventRecord(ev[2], ctx->stream);
    dx_prof_end(ctx);
    if (timed)
      { float t1 = 0, t2 = 0;
        (void) hipEventSynchronize(ev[2]);
        (void) hipEventElapsedTime(&t1, ev[0], ev[1]); (void) hipEventElapsedTime(&t2, ev[1], ev[2]);
        fprintf(stderr, "[walk] k_walk_find %.2f ms, k_walk_pieces %.2f ms (%u threads a workgroup)\n", t1, t2, bs);
      }
    for (int i = 0; i < 3; i++) if (ev[i]) (void) hipEventDestroy(ev[i]);
  }
  WALK_HIP(hipGetLastError());
  WALK_HIP(hipMemcpyAsync(pc, d_pc, P * sizeof(walk_piece_d), hipMemcpyDeviceToHost, ctx->stream));
  WALK_HIP(hipStreamSynchronize(ctx->stream));

  if (dx_test_on("walk_debug"))
    { uint64_t tried = 0, none = 0;
      for (uint64_t k = 0; k < P; k++) { tried += pc[k].tried; none += (pc[k].flags & WP_NONE) != 0; }
      fprintf(stderr, "[walk] %llu pieces of %llu bytes: %llu guesses did not hold, %llu pieces without a start\n",
              (unsigned long long) P, (unsigned long long) a.piece, (unsigned long long) tried, (unsigned long long) none);
    }
  // the chain, from the first record on
  { uint64_t pos = first, k = 0;
    int rounds = 0;
    while (pos < n)
      { // A header may begin with bytes of 255 (255 wells each).  The lane of a piece never starts on one (k_walk_find): it takes
        // the offset behind them, where the same record reads with fewer wells, and says how many bytes of 255 stand in front
        // of its start.  The chain knows where the record before ended: the bytes of 255 from there on are this header's --
        // first record's offset, framing bytes and wells put right in the gather (trim < 0), no second walk.  (trim > 0, a start
        // in front of where the chain arrives with only bytes of 255 between: cannot happen any more, handled all the same.)
        trim[k] = 0;
        if (!(pc[k].flags & WP_NONE) && pc[k].count > 0)
          { if (pc[k].start > pos && pc[k].start - pos <= (uint64_t) pc[k].lead255)
              trim[k] = (uint64_t) 0 - (pc[k].start - pos);
            else if (pc[k].start < pos && pos - pc[k].start <= (uint64_t) pc[k].first_hdr - 13u)
              trim[k] = pos - pc[k].start;
            if (trim[k])
              { const int64_t t = (int64_t) trim[k];
                pc[k].start = pos; pc[k].hdr_sum = (uint32_t) ((int64_t) pc[k].hdr_sum - t); pc[k].dwell_sum = (uint64_t) ((int64_t) pc[k].dwell_sum - 255 * t);
              }
          }
        if (pc[k].flags & WP_NONE || pc[k].start != pos)      // the chain arrives elsewhere than this piece's lane started: once more, from here
          { if (dx_test_on("walk_debug")) fprintf(stderr, "[walk] piece %llu: chain arrives at %llu, lane started at %llu (flags %u, %u records, %u bytes of 255 in front): walked again\n", (unsigned long long) k, (unsigned long long) pos, (unsigned long long) pc[k].start, pc[k].flags, pc[k].count, pc[k].lead255);
            if (++rounds > WALK_ROUNDS) WALK_FAIL(DX_E_MISMATCH, "dx_qv_walk_device: the pieces' walks do not chain up");
            if (d_start == NULL) { WALK_HIP(hipMalloc(&d_start, P * 8)); WALK_HIP(hipMalloc(&d_todo, 4)); }
            const uint32_t kk = (uint32_t) k;
            WALK_HIP(hipMemcpyAsync(d_start + k, &pos, 8, hipMemcpyHostToDevice, ctx->stream));
            WALK_HIP(hipMemcpyAsync(d_todo, &kk, 4, hipMemcpyHostToDevice, ctx->stream));
            dx_prof_begin(ctx, DX_K_QV_WALK);
            hipLaunchKernelGGL(k_walk_pieces, dim3(1), dim3(64), 0, ctx->stream, a, d_cand, (const uint32_t *) d_ncand,
                               (const uint32_t *) d_todo, 1u, (const uint64_t *) d_start, d_pc, d_recs, rcap, d_gwords, gcap, (uint32_t *) NULL);
            dx_prof_end(ctx);
            WALK_HIP(hipGetLastError());
            WALK_HIP(hipMemcpyAsync(pc + k, d_pc + k, sizeof(walk_piece_d), hipMemcpyDeviceToHost, ctx->stream));
            WALK_HIP(hipStreamSynchronize(ctx->stream));
          }
        if (pc[k].flags & (WP_BAD | WP_OVERFLOW))
          WALK_FAIL(DX_E_MISMATCH, (pc[k].flags & WP_BAD) ? "dx_qv_walk_device: a record does not walk (damaged stream?)"
                                                          : "dx_qv_walk_device: more records in a piece than its scratch holds");
        onchain[k] = 1;
        N  += pc[k].count;
        pos = pc[k].end;
        if (pos >= n) break;
        { const uint64_t k2 = (pos - first) / a.piece;
          if (k2 <= k || k2 >= P) WALK_FAIL(DX_E_MISMATCH, "dx_qv_walk_device: a walk ended inside its own piece");
          k = k2;
        }
      }
    if (pos != n && n > first) WALK_FAIL(DX_E_MISMATCH, "dx_qv_walk_device: the last record ends behind the stream");
  }

  // side by side
  { uint64_t at = 0, hb = 0, wb = 0;
    for (uint64_t k = 0; k < P; k++)
      { dst[k] = onchain[k] ? at : ~0ull; dst[P + k] = hb; dst[2 * P + k] = wb;
        if (onchain[k]) { at += pc[k].count; hb += pc[k].hdr_sum; wb += pc[k].dwell_sum; }
      }
    WALK_HIP(hipMalloc(&d_dst, 4 * P * 8));
    WALK_HIP(hipMemcpyAsync(d_dst, dst, 4 * P * 8, hipMemcpyHostToDevice, ctx->stream));
    WALK_HIP(hipMalloc(&out->d_rec_off, (N + 1) * 8));
    WALK_HIP(hipMalloc(&out->d_hdr_off, (N + 1) * 8));
    WALK_HIP(hipMalloc(&out->d_seg, (N + 1) * 20));
    WALK_HIP(hipMalloc(&out->d_len, (N + 1) * 4));
    WALK_HIP(hipMalloc(&out->d_hdr4, (N + 1) * 16));
    if (groups && N > 0)
      { WALK_HIP(hipMalloc(&d_gsrc, N * 32)); WALK_HIP(hipMalloc(&d_gtok, N * 16)); }
    dx_prof_begin(ctx, DX_K_QV_WALK);
    hipLaunchKernelGGL(k_walk_gather, dim3((unsigned) ((P + DX_WAVES_PER_BLK - 1) / DX_WAVES_PER_BLK)), dim3(DX_BLOCK), 0, ctx->stream,
                       P, (const walk_piece_d *) d_pc, (const walk_rec_d *) d_recs, rcap, (const uint64_t *) d_dst,
                       (const uint64_t *) (d_dst + P), (const uint64_t *) (d_dst + 2 * P), (const uint64_t *) (d_dst + 3 * P),
                       out->d_rec_off, out->d_hdr_off, out->d_seg, out->d_len, out->d_hdr4, gcap, d_gsrc, d_gtok);
    dx_prof_end(ctx);
    WALK_HIP(hipGetLastError());
    const uint64_t ends[2] = { n, hb };
    WALK_HIP(hipMemcpyAsync(out->d_rec_off + N, &ends[0], 8, hipMemcpyHostToDevice, ctx->stream));
    WALK_HIP(hipMemcpyAsync(out->d_hdr_off + N, &ends[1], 8, hipMemcpyHostToDevice, ctx->stream));
    WALK_HIP(hipStreamSynchronize(ctx->stream));
    out->n = N; out->pieces = P; out->piece_bytes = a.piece;
  }
  // the run-coded lines' share of the group index, laid out as the decoders take it (an entry's room, a scan, the words)
  if (groups && N > 0)
    { uint64_t words = 0;
      uint32_t lost[2] = { 0, 0 }, sync_kinds = 0;
      for (int q = 0; q < 4; q++)                          // the plain lines without escape codes (dx_qv_decode's `plain`)
        if ((q == 0 ? cd->delChar : q == 3 ? cd->subChar : -1) < 0 && !a.esc[q]) sync_kinds |= 1u << q;
      WALK_HIP(hipMalloc(&d_room, N * 4));
      WALK_HIP(hipMalloc(&d_none, 64));
      WALK_HIP(hipMalloc(&out->d_gidx_off, (N + 1) * 8));
      WALK_HIP(hipMemsetAsync(d_none, 0, 8, ctx->stream));
      dx_prof_begin(ctx, DX_K_QV_WALK);
      hipLaunchKernelGGL(k_walk_rooms, dim3((unsigned) ((N + DX_BLOCK - 1) / DX_BLOCK)), dim3(DX_BLOCK), 0, ctx->stream,
                         (const uint32_t *) out->d_len, (const uint32_t *) d_gtok, N, cd->delChar >= 0 ? 1 : 0, cd->subChar >= 0 ? 1 : 0, d_room);
      dx_prof_end(ctx);
      WALK_HIP(hipGetLastError());
      rc = dx_scan_u32(ctx, d_room, N, out->d_gidx_off, &words);
      if (rc != DX_OK) goto done;
      WALK_HIP(hipMalloc(&out->d_gidx, (words + 4) * 4));
      dx_prof_begin(ctx, DX_K_QV_WALK);
      hipLaunchKernelGGL(k_walk_index, dim3((unsigned) ((N + DX_WAVES_PER_BLK - 1) / DX_WAVES_PER_BLK)), dim3(DX_BLOCK), 0, ctx->stream,
                         (const uint32_t *) out->d_len, (const uint32_t *) d_gtok, (const uint64_t *) d_gsrc, (const uint32_t *) d_gwords, N,
                         (const uint64_t *) out->d_gidx_off, out->d_gidx, cd->delChar >= 0 ? 1 : 0, cd->subChar >= 0 ? 1 : 0, sync_kinds, d_none);
      dx_prof_end(ctx);
      WALK_HIP(hipGetLastError());
      WALK_HIP(hipMemcpyAsync(lost, d_none, 8, hipMemcpyDeviceToHost, ctx->stream));
      WALK_HIP(hipStreamSynchronize(ctx->stream));
      out->gidx_words = words; out->gidx_none = lost[0]; out->gidx_nosync = lost[1]; out->sync_kinds = sync_kinds;
    }
  rc = DX_OK;
done:
  (void) hipFree(d_blob); (void) hipFree(d_tail); (void) hipFree(d_cand); (void) hipFree(d_ncand); (void) hipFree(d_pc); (void) hipFree(d_recs);
  (void) hipFree(d_start); (void) hipFree(d_todo); (void) hipFree(d_dst);
  (void) hipFree(d_queue); (void) hipFree(d_gwords); (void) hipFree(d_gtok); (void) hipFree(d_room); (void) hipFree(d_none); (void) hipFree(d_gsrc);
  free(blob); free(pc); free(dst); free(onchain);
  if (rc != DX_OK) dx_qv_dindex_free(ctx, out);
  return rc;
#undef WALK_HIP
#undef WALK_FAIL
}
