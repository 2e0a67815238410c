// dx_qv_short.hpp -- dexqv for batches of SHORT entries: a lane per entry (included by dx_qv.hip, whose helpers it uses).
//
// The wave-per-entry kernels give an entry a whole wavefront: a step is 1 KiB of each of its lines, and an entry pays its fixed
// costs -- a ticket, its whereabouts, five first requests, 1.5 KB of counters out, five segment ends -- whatever its length.  At
// 300 symbols an entry fills a third of a step and the batch runs at 0.06 of the HBM peak (4 M x 300: 27 ms for 6 GB); the
// round-4 verdict asked for several entries per wave.  Here: 64 entries per wave, a lane each, through the reference's own loops
// (QVcoding_Scan QV.c:922-1023 with Histogram_Seqs / Histogram_Runs :702-724; Compress_Next_QVentry QV.c:1381-1426 with
// Encode :386-443, Encode_Run :448-506, Pack_Tag :810-819 + Number_Read + Compress_Read) -- a lane reads its lines 16 bytes at
// a time and goes through them byte by byte: ~20 instructions a symbol where the wave-per-entry kernels spend 2, but no step is
// ever a third full and nothing is per entry but the entry.  Three passes over the text (histograms; sizes; records written where
// the sizes' prefix sums put them), as dx_qv_sizes / dx_qv_encode have it.  Taken for batches whose entries average at most
// QS_MEAN symbols (dx_qv_hist, dx_qv_encode_onepass, dx_qv_sizes, dx_qv_encode); DEXGPU_NO_SHORT: never.
//
// Roofline: HBM in name only -- a lane's 16-byte loads are a request each; the kernels are bound by their instructions.
#ifndef QS_MEAN
#define QS_MEAN   1200u
#endif
#define QS_BLOCK  256
#define QS_COPIES 4u                    // copies of every histogram bin in a workgroup's LDS (copy = lane & 3)

#define QS_W      8                     // 16-byte chunks a lane asks for at a time: one cache line of its line, whole and aligned (its neighbours'
                                        // lines are 1.5 KB away: a lane's requests share nothing with theirs, and a cache line asked for in two goes
                                        // is fetched twice -- 28 waves x 64 lanes x 128 bytes a CU outlive neither the 32 KB L1 nor a share of the L2)
#define QS_MAXLEN 4096u                 // ... and no entry longer than this (a lane is alone with its entry)

#define QS_BYTE(v, b) ((((b) < 4 ? (v).x : (b) < 8 ? (v).y : (b) < 12 ? (v).z : (v).w) >> (8 * ((b) & 3))) & 0xffu)

typedef __attribute__((address_space(3))) uint32_t qs_lds;        // (the tables and counters of a workgroup)
#define QS_LDS(p) ((qs_lds *) (p))
#define QS_ADD(p, v) ((void) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))

typedef __attribute__((address_space(1))) const u32x4_u qs_g128;  // (global_load / global_store, not flat)
typedef __attribute__((address_space(1))) u32_u qs_g32;
typedef __attribute__((address_space(1))) uint8_t qs_g8;
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));

// The cache line around p + pos: its 32 words (one vector: the chunk a uniform loop counter names is read out of it through the
// register index, not by selects), and which of its bytes are the line's from pos on: [skip, skip + len).
struct qs_span { u32x32 v; uint32_t skip, len; };
// 16 bytes from q of which some lie outside the image [lo, end): those are zero (rare: the image's first and last entries)
__device__ __noinline__ u32x4 qs_edge(const uint8_t *q, const uint8_t *lo, const uint8_t *end)
{ uint32_t w[4] = { 0u, 0u, 0u, 0u };
  for (int i = 0; i < 16; i++)
    if (q + i >= lo && q + i < end) w[i >> 2] |= (uint32_t) *(const qs_g8 *) (q + i) << (8 * (i & 3));
  return u32x4{ w[0], w[1], w[2], w[3] };
}
// the 128 bytes from q0 on in eight requests that go out together.  lo, end16: the first and the last address of the image 16
// bytes may be read from: a chunk outside is asked for at the nearest address inside, and put together byte by byte afterwards.
__device__ __forceinline__ u32x32 qs_fetch_raw(const uint8_t *q0, const uint8_t *lo, const uint8_t *end16)
{ u32x4 c[QS_W];
  _Pragma("unroll")
  for (int k = 0; k < QS_W; k++)
    { const uint8_t *qk = q0 + 16 * k;
      qk = qk < lo ? lo : qk;
      c[k] = *(qs_g128 *) (qk < end16 ? qk : end16);
    }
  if (q0 < lo || q0 + 16 * (QS_W - 1) > end16)
    { _Pragma("unroll")
      for (int k = 0; k < QS_W; k++)
        { const uint8_t *qk = q0 + 16 * k;
          if (qk < lo || qk > end16) c[k] = qs_edge(qk, lo, end16 + 16);
        }
    }
  u32x32 v;
  _Pragma("unroll")
  for (int k = 0; k < QS_W; k++)
    { v[4 * k] = c[k].x; v[4 * k + 1] = c[k].y; v[4 * k + 2] = c[k].z; v[4 * k + 3] = c[k].w; }
  return v;
}
// the cache line around p + pos
__device__ __forceinline__ qs_span qs_fetch(const uint8_t *p, uint32_t pos, uint32_t L, const uint8_t *lo, const uint8_t *end16)
{ qs_span s;
  const uint8_t *q = p + pos;
  s.skip = (uint32_t) ((uintptr_t) q & 127u);
  s.len  = L - pos < 128u - s.skip ? L - pos : 128u - s.skip;
  s.v    = qs_fetch_raw(q - s.skip, lo, end16);
  return s;
}
// a line's spans, a span's chunks (k uniform), a chunk's bytes
#define QS_SPANS(s, p, L, lo, end16) for (uint32_t pos = 0, step_ = 0; pos < (L); pos += step_) { qs_span s = qs_fetch(p, pos, L, lo, end16); step_ = (s).len;
#define QS_CHUNKS(s, k, c) _Pragma("nounroll") for (int k = 0; k < QS_W; k++) \
                             { if (16u * k + 16u <= (s).skip || 16u * k >= (s).skip + (s).len) continue; \
                               const u32x4 c = { (s).v[4 * k], (s).v[4 * k + 1], (s).v[4 * k + 2], (s).v[4 * k + 3] };
#define QS_BYTES(b)        _Pragma("unroll") for (int b = 0; b < 16; b++)
#define QS_LIVE(s, k, b)   ((uint32_t) (16 * (k) + (b)) - (s).skip < (s).len)

__global__ __launch_bounds__(256)
void k_qs_maxlen(const uint32_t *len, uint64_t n, uint32_t *out)
{ uint32_t m = 0;
  for (uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256u)
    m = len[i] > m ? len[i] : m;
  for (int d = 32; d >= 1; d >>= 1)
    { const uint32_t o = (uint32_t) __shfl_xor((int) m, d);
      m = o > m ? o : m;
    }
  if (lane_id() == 0 && m) atomicMax(out, m);
}

// ---------------------------------------------------------------------------------------------
//  histograms (QVcoding_Scan, QV.c:988-1017)
// ---------------------------------------------------------------------------------------------
// one line's symbols counted (Histogram_Seqs, QV.c:702-708)
__device__ __forceinline__ void qs_count(const uint8_t *p, uint32_t L, const uint8_t *lo, const uint8_t *end16, qs_lds *bins /* + the lane's copy */)
{ QS_SPANS(s, p, L, lo, end16)
      QS_CHUNKS(s, k, c)
        QS_BYTES(b) if (QS_LIVE(s, k, b)) QS_ADD(bins + QS_COPIES * QS_BYTE(c, b), 1u);
      }
    }
}
// ... of a line with a run character: the symbols, and the run lengths before every other symbol and at the line's end
// (Histogram_Runs, QV.c:710-724) when `runs`
__device__ __forceinline__ void qs_count_runs(const uint8_t *p, uint32_t L, const uint8_t *lo, const uint8_t *end16, uint32_t rc, bool runs, qs_lds *bins, qs_lds *rbins)
{ uint32_t nrc = 0, run = 0;                              // the run characters themselves are counted in a register
  QS_SPANS(s, p, L, lo, end16)
      QS_CHUNKS(s, k, c)
        QS_BYTES(b)
          { const uint32_t x = QS_BYTE(c, b);
            const bool live = QS_LIVE(s, k, b);
            if (x == rc) { nrc += live; run += live; }
            else if (live)
              { QS_ADD(bins + QS_COPIES * x, 1u);
                if (runs) QS_ADD(rbins + QS_COPIES * (run > 255u ? 255u : run), 1u);
                run = 0;
              }
          }
      }
    }
  if (nrc) QS_ADD(bins + QS_COPIES * rc, nrc);
  if (runs && run) QS_ADD(rbins + QS_COPIES * (run > 255u ? 255u : run), 1u);
}

__global__ __launch_bounds__(QS_BLOCK)
void k_qs_hist(qv_args a, uint64_t entry0, long long del_first, long long sub_first, unsigned long long *hist, unsigned long long *tot)
{ __shared__ uint32_t s_h[6][256][QS_COPIES];             // 24 KB
  for (uint32_t i = threadIdx.x; i < 6u * 256u * QS_COPIES; i += QS_BLOCK) (&s_h[0][0][0])[i] = 0u;
  __syncthreads();
  const uint32_t cp = threadIdx.x & (QS_COPIES - 1u);
  #define H(s) (QS_LDS(&s_h[s][0][0]) + cp)
  uint64_t chars = 0;
  for (uint64_t r = (uint64_t) blockIdx.x * QS_BLOCK + threadIdx.x; r < a.n; r += (uint64_t) gridDim.x * QS_BLOCK)
    { const uint32_t L = a.len[r];
      const uint8_t *p0 = line_ptr(a, r, L, 0), *p2 = line_ptr(a, r, L, 2), *p3 = line_ptr(a, r, L, 3), *p4 = line_ptr(a, r, L, 4);
      const uint8_t *end16 = a.text + a.text_bytes - 16u;    // (text_bytes known and large: qs_short)
      chars += L;
      // the run histograms take part from the entry on in which the run character was found (QV.c:1003, 1016)
      if (a.delChar >= 0)
        qs_count_runs(p0, L, a.text, end16, (uint32_t) a.delChar & 0xffu, (long long) (entry0 + r) >= del_first, H(DX_DEL), H(DX_DRUN));
      else
        qs_count(p0, L, a.text, end16, H(DX_DEL));
      qs_count(p2, L, a.text, end16, H(DX_INS));
      qs_count(p3, L, a.text, end16, H(DX_MRG));
      if (a.subChar >= 0)
        qs_count_runs(p4, L, a.text, end16, (uint32_t) a.subChar & 0xffu, (long long) (entry0 + r) >= sub_first, H(DX_SUB), H(DX_SRUN));
      else
        qs_count(p4, L, a.text, end16, H(DX_SUB));
    }
  #undef H
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 6u * 256u; i += QS_BLOCK)
    { uint32_t v = 0;
      for (uint32_t c = 0; c < QS_COPIES; c++) v += (&s_h[0][0][0])[i * QS_COPIES + c];
      if (v) atomicAdd(hist + i, (unsigned long long) v);
    }
  { const uint32_t lo = (uint32_t) chars, hi = (uint32_t) (chars >> 32);          // (summed in pieces a 32-bit wave_sum cannot overflow)
    const uint64_t s = (uint64_t) wave_sum(lo & 0xffffu) + ((uint64_t) wave_sum(lo >> 16) << 16) + ((uint64_t) wave_sum(hi) << 32);
    if (lane_id() == 0 && s) atomicAdd(tot, (unsigned long long) s);
  }
}

// ---------------------------------------------------------------------------------------------
//  a lane's walk over one entry: sizes only (EMIT = false) or the record written (EMIT = true)
// ---------------------------------------------------------------------------------------------
// MSB-first into 32-bit words as OCODE does (QV.c:404-422); T: bits so far, last: the length of the last code appended (an
// escape's literal counts as a code of its own: the pad rule QV.c:436-442 looks at where the last OCODE began).  Finished words
// wait in q0..q2 for a fourth: a lane's stores are 16 bytes (its neighbours' records are elsewhere: nothing coalesces).
#ifndef QS_QUAD
#define QS_QUAD 0
#endif
struct qs_bits { uint8_t *p; uint64_t acc; uint32_t fill, T, last, q0, q1, q2, nq; };

__device__ __forceinline__ void qs_word(qs_bits &w, uint32_t word)
{
#if QS_QUAD
  if (w.nq == 3u)
    { *(__attribute__((address_space(1))) u32x4_u *) w.p = u32x4{ w.q0, w.q1, w.q2, word };
      w.p += 16; w.nq = 0u;
    }
  else
    { w.q0 = w.nq == 0u ? word : w.q0; w.q1 = w.nq == 1u ? word : w.q1; w.q2 = w.nq == 2u ? word : w.q2;
      w.nq += 1u;
    }
#else
  *(qs_g32 *) w.p = word; w.p += 4;
#endif
}
__device__ __forceinline__ void qs_drain(qs_bits &w)
{
#if QS_QUAD
  if (w.nq > 0u) { *(qs_g32 *) w.p = w.q0; w.p += 4; }
  if (w.nq > 1u) { *(qs_g32 *) w.p = w.q1; w.p += 4; }
  if (w.nq > 2u) { *(qs_g32 *) w.p = w.q2; w.p += 4; }
  w.nq = 0u;
#endif
}

// len bits (<= 32; 0: nothing), right-aligned in bits
template <bool EMIT>
__device__ __forceinline__ void qs_put(qs_bits &w, uint32_t len, uint32_t bits)
{ if (!EMIT) w.T += len;
  if (EMIT)
    { w.acc |= ((uint64_t) bits << (32u - len)) << (32u - w.fill);      // (fill < 32)
      w.fill += len;
      if (w.fill >= 32u)
        { qs_word(w, (uint32_t) (w.acc >> 32));
          w.acc <<= 32; w.fill -= 32u;
        }
    }
}
// a run's code and, behind the longest code, its 16-bit literal (QV.c:478-488)
template <bool EMIT>
__device__ __forceinline__ void qs_run(qs_bits &w, const qs_lds *rtab, uint32_t run)
{ const uint32_t t = rtab[run > 255u ? 255u : run];
  qs_put<EMIT>(w, TOK_LEN(t), TOK_BITS(t));
  w.last = TOK_LEN(t);
  if (TOK_ESC(t)) { qs_put<EMIT>(w, 16u, run & 0xffffu); w.last = 16u; }
}
// the segment's end (QV.c:436-442): its bytes
template <bool EMIT>
__device__ __forceinline__ uint32_t qs_finish(qs_bits &w, const uint8_t *dst)
{ if (EMIT) w.T = (uint32_t) (w.p - dst) * 8u + 32u * w.nq + w.fill;      // (the sizes count their bits, the records their words)
  const uint32_t olen = w.T & 31u, llen = (w.T - w.last) & 31u;
  const uint32_t words = (w.T >> 5) + (olen ? 1u : 0u);
  const bool again = olen ? (llen > 16u && olen > llen) : (w.T > 0u && llen > 16u);
  if (EMIT)
    { const uint32_t part = (uint32_t) (w.acc >> 32);       // (0 when the last word was whole)
      qs_drain(w);
      if (olen) { *(qs_g32 *) w.p = part; w.p += 4; }
      if (again) { *(qs_g32 *) w.p = part; w.p += 4; }
    }
  return 4u * (words + (again ? 1u : 0u));
}

struct qs_ret { uint32_t bytes, n; };                   // the segment's bytes; the symbols under which a tag stands

// one plain line (Encode, QV.c:386-443); mask: the lossy rounding of the insertion / merge QVs (QV.c:1406-1415), in all four bytes
// of a word.  A token holds a symbol's code and, for an escape, its 8-bit literal behind it (one OCODE each, QV.c:430-433: the
// last code is then the literal); a byte beyond the line is a token of no bits.
template <bool EMIT>
__device__ __forceinline__ qs_ret qs_plain(const uint8_t *p, uint32_t L, const uint8_t *lo, const uint8_t *end16, const qs_lds *tab, uint32_t mask4, uint8_t *dst)
{ qs_bits w = { dst, 0ull, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
  uint32_t tl = 0;                                        // the line's last token
  QS_SPANS(s, p, L, lo, end16)
      s.v &= mask4;
      QS_CHUNKS(s, k, c)
        uint32_t t[16];
        QS_BYTES(b) t[b] = tab[QS_BYTE(c, b)];            // (whatever byte: an address in the table; the sixteen look-ups go out together)
        QS_BYTES(b)
          { const bool live = QS_LIVE(s, k, b);
            t[b] = live ? t[b] : 0u;
            tl   = live ? t[b] : tl;
            qs_put<EMIT>(w, TOK_LEN(t[b]), TOK_BITS(t[b]));
          }
      }
    }
  w.last = TOK_ESC(tl) ? 8u : TOK_LEN(tl);
  return qs_ret{ qs_finish<EMIT>(w, dst), L };
}

// 2-bit codes, four to a byte, the first in the top bits (Compress_Read DB.c:319-338): sixteen to a stored word
struct qs_tags { uint8_t *p; uint32_t acc, n; };
__device__ __forceinline__ void qs_tag(qs_tags &g, uint32_t letter)
{ g.acc = (g.acc << 2) | tag_code(letter);
  g.n += 1u;
  if ((g.n & 15u) == 0u)
    { *(qs_g32 *) g.p = __builtin_bswap32(g.acc); g.p += 4; }
}
__device__ __forceinline__ void qs_tag_end(qs_tags &g)
{ const uint32_t left = g.n & 15u;
  if (left)
    { const uint32_t v = g.acc << (2u * (16u - left));
      for (uint32_t k = 0; 4u * k < left; k++) *(qs_g8 *) (g.p + k) = (uint8_t) (v >> (24u - 8u * k));
    }
}

// one run-coded line (Encode_Run, QV.c:448-506); TAGS: the deletion line -- the tags under its non-run symbols are packed on the
// way (Pack_Tag QV.c:810-819, Number_Read DB.c:393-416), .n = how many
template <bool EMIT, bool TAGS>
__device__ __forceinline__ qs_ret qs_runs(const uint8_t *p, const uint8_t *ptag, uint32_t L, const uint8_t *lo, const uint8_t *end16, const qs_lds *tab,
                                       const qs_lds *rtab, uint32_t rc, uint8_t *dst, uint8_t *tdst)
{ qs_bits w = { dst, 0ull, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
  qs_tags g = { tdst, 0u, 0u };
  uint32_t run = 0;
  QS_SPANS(s, p, L, lo, end16)
      u32x32 gv = s.v;                                     // the tags under the span's bytes: the same 128 positions of the tag line
      if (TAGS && EMIT) gv = qs_fetch_raw(ptag + pos - s.skip, lo, end16);
      QS_CHUNKS(s, k, c)
        const u32x4 gc = { gv[4 * k], gv[4 * k + 1], gv[4 * k + 2], gv[4 * k + 3] };
        uint32_t ts[16];
        QS_BYTES(b) ts[b] = tab[QS_BYTE(c, b)];
        QS_BYTES(b)
          { const uint32_t x = QS_BYTE(c, b);
            const bool live = QS_LIVE(s, k, b);
            if (x == rc) run += live;
            else if (live)
              { const uint32_t tr = rtab[run > 255u ? 255u : run], t = ts[b];
                if (!TOK_ESC(tr | t))                        // the run's code and the symbol's in one piece of <= 32 bits
                  { qs_put<EMIT>(w, TOK_LEN(tr) + TOK_LEN(t), (TOK_BITS(tr) << TOK_LEN(t)) | TOK_BITS(t));
                    w.last = TOK_LEN(t);
                  }
                else                                         // (a literal behind one of them)
                  { qs_run<EMIT>(w, rtab, run);
                    qs_put<EMIT>(w, TOK_LEN(t), TOK_BITS(t));
                    w.last = TOK_ESC(t) ? 8u : TOK_LEN(t);
                  }
                run = 0;
                if (TAGS && EMIT) qs_tag(g, QS_BYTE(gc, b));
                else g.n += 1u;
              }
          }
      }
    }
  if (run) qs_run<EMIT>(w, rtab, run);                      // (a line that ends in a run: its code, no symbol behind it)
  if (TAGS && EMIT) qs_tag_end(g);
  return qs_ret{ qs_finish<EMIT>(w, dst), g.n };
}

// the whole tag line packed (no deletion run character: QV.c:1393-1396)
__device__ __forceinline__ void qs_tags_all(const uint8_t *ptag, uint32_t L, const uint8_t *lo, const uint8_t *end16, uint8_t *tdst)
{ qs_tags g = { tdst, 0u, 0u };
  QS_SPANS(s, ptag, L, lo, end16)
      QS_CHUNKS(s, k, c)
        QS_BYTES(b) if (QS_LIVE(s, k, b)) qs_tag(g, QS_BYTE(c, b));
      }
    }
  qs_tag_end(g);
}

// EMIT = false: seg[5 r ..] and rec_size[r] (dx_qv_sizes' contract); EMIT = true: the record at rec_off[r], its segment sizes
// compared with seg (status bit 1: they differ).  A symbol the tables have
// no code for costs no bits, as in the reference (Encode's OCODE of length 0) and in k_qv_encode
template <bool EMIT>
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_entries(qv_args a, const uint32_t *g_tok, const uint8_t *hdr, const uint64_t *hdr_off, const uint64_t *rec_off,
                  uint32_t *seg, uint32_t *rec_size, uint8_t *out, uint32_t *status)
{ __shared__ uint32_t s_tok[6][256];
  load_tables(s_tok, g_tok);
  const uint32_t imask = a.lossy ? 0xfefefefeu : ~0u, mmask = a.lossy ? 0xfcfcfcfcu : ~0u;
  uint32_t differ = 0;
  for (uint64_t r = (uint64_t) blockIdx.x * QS_BLOCK + threadIdx.x; r < a.n; r += (uint64_t) gridDim.x * QS_BLOCK)
    { const uint32_t L = a.len[r];
      const uint8_t *p0 = line_ptr(a, r, L, 0), *p1 = line_ptr(a, r, L, 1), *p2 = line_ptr(a, r, L, 2);
      const uint8_t *p3 = line_ptr(a, r, L, 3), *p4 = line_ptr(a, r, L, 4);
      const uint8_t *end16 = a.text + a.text_bytes - 16u;    // (text_bytes known and large: qs_short)
      const uint32_t hl = hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u;
      uint32_t sg[5] = { 0u, 0u, 0u, 0u, 0u }, want[5] = { 0u, 0u, 0u, 0u, 0u }, clen = L;
      uint8_t *dst = out;
      if (EMIT)
        { dst = out + rec_off[r];
          for (uint32_t k = 0; k < hl; k++) *(qs_g8 *) (dst + k) = hdr[hdr_off[r] + k];           // the record's framing bytes (dexqv.c:128-139)
          dst += hl;
          want[0] = seg[5 * r]; want[1] = seg[5 * r + 1]; want[2] = seg[5 * r + 2]; want[3] = seg[5 * r + 3]; want[4] = seg[5 * r + 4];
        }
      uint8_t *d0 = dst, *d1 = d0 + want[0], *d2 = d1 + want[1], *d3 = d2 + want[2], *d4 = d3 + want[3];
      qs_ret q;
      if (a.delChar >= 0)
        { q = qs_runs<EMIT, true>(p0, p1, L, a.text, end16, QS_LDS(s_tok[DX_DEL]), QS_LDS(s_tok[DX_DRUN]), (uint32_t) a.delChar & 0xffu, d0, d1);
          clen = q.n;
        }
      else
        { q = qs_plain<EMIT>(p0, L, a.text, end16, QS_LDS(s_tok[DX_DEL]), ~0u, d0);
          if (EMIT) qs_tags_all(p1, L, a.text, end16, d1);
        }
      sg[0] = q.bytes;
      sg[1] = (clen + 3u) >> 2;
      q = qs_plain<EMIT>(p2, L, a.text, end16, QS_LDS(s_tok[DX_INS]), imask, d2);
      sg[2] = q.bytes;
      q = qs_plain<EMIT>(p3, L, a.text, end16, QS_LDS(s_tok[DX_MRG]), mmask, d3);
      sg[3] = q.bytes;
      if (a.subChar >= 0)
        q = qs_runs<EMIT, false>(p4, p4, L, a.text, end16, QS_LDS(s_tok[DX_SUB]), QS_LDS(s_tok[DX_SRUN]), (uint32_t) a.subChar & 0xffu, d4, d4);
      else
        q = qs_plain<EMIT>(p4, L, a.text, end16, QS_LDS(s_tok[DX_SUB]), ~0u, d4);
      sg[4] = q.bytes;
      if (EMIT)
        { if (sg[0] != want[0] || sg[1] != want[1] || sg[2] != want[2] || sg[3] != want[3] || sg[4] != want[4]) differ = 1u; }
      else
        { seg[5 * r] = sg[0]; seg[5 * r + 1] = sg[1]; seg[5 * r + 2] = sg[2]; seg[5 * r + 3] = sg[3]; seg[5 * r + 4] = sg[4];
          rec_size[r] = hl + sg[0] + sg[1] + sg[2] + sg[3] + sg[4];
        }
    }
  if (differ) atomicOr(status, 2u);
}

// ---------------------------------------------------------------------------------------------
//  host side
// ---------------------------------------------------------------------------------------------
static int qs_grid(const dx_ctx *ctx, uint64_t n)
{ const uint64_t need = (n + QS_BLOCK - 1) / QS_BLOCK, room = (uint64_t) ctx->num_cu * 32u;
  return (int) (need < room ? need : room);
}

// is this a batch for the lane-per-entry kernels?  Mean length from the image's size, the longest entry from the device.
// (decided anew by every call from the same inputs: dx_qv_hist, dx_qv_sizes, dx_qv_encode and the one-pass encoder agree)
static int qs_short(dx_ctx *ctx, const dx_qv_batch *b, bool *yes)
{ *yes = false;
  const char *off = getenv("DEXGPU_NO_SHORT");
  if ((off != NULL && off[0] != '\0' && off[0] != '0') || b->n < 4096 || b->text_bytes == 0) return DX_OK;
  if (b->text_bytes / b->n > 5ull * (QS_MEAN + 1u) + 64u) return DX_OK;
  uint32_t *d_max = (uint32_t *) (ctx->d_u64 + 40), longest = 0;
  DX_HIP(ctx, hipMemsetAsync(d_max, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_qs_maxlen, dim3((unsigned) ctx->num_cu * 4u), dim3(256), 0, ctx->stream, (const uint32_t *) b->d_len, b->n, d_max);
  DX_HIP(ctx, hipMemcpyAsync(&longest, d_max, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *yes = longest <= QS_MAXLEN;
  return DX_OK;
}

// sizes, offsets, records: dx_qv_encode_onepass's contract
static int onepass_short(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                         uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap, uint64_t *total)
{ const uint64_t n = b->n;
  int e;
  uint8_t *scr;
  if ((e = dx_scratch(ctx, n * 4 + 256, (void **) &scr))) return e;
  uint32_t *d_size = (uint32_t *) scr;
  dx_sx_drop_external(ctx);
  ctx->sx.valid = 0;                                     // (these kernels leave no group index)
  const qv_args a = make_args(b, ctx->delChar, ctx->subChar, ctx->lossy);
  DX_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
  DX_LAUNCH(ctx, DX_K_QV_SIZES, k_qs_entries<false>, qs_grid(ctx, n), QS_BLOCK, a, (const uint32_t *) ctx->d_tok, (const uint8_t *) NULL, d_hdr_off,
            (const uint64_t *) NULL, d_seg, d_size, (uint8_t *) NULL, ctx->d_status);
  uint64_t tot = 0;
  if ((e = dx_scan_u32(ctx, d_size, n, d_rec_off, &tot))) return e;
  if (total) *total = tot;
  ctx->route.groups = 0; ctx->route.direct = 3; ctx->route.tokens = 0; ctx->route.region_bytes = 0;
  ctx->route.scratch_bytes = ctx->scratch_bytes; ctx->route.token_bytes = 0; ctx->route.text_entries = n;
  if (tot > out_cap)
    return dx_fail(ctx, DX_E_SPACE, "dx_qv_encode_onepass: the record stream needs %llu bytes, d_out holds %llu",
                   (unsigned long long) tot, (unsigned long long) out_cap);
  DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, k_qs_entries<true>, qs_grid(ctx, n), QS_BLOCK, a, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
            (const uint64_t *) d_rec_off, d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  uint32_t st = 0;
  DX_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st & 2u)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode_onepass: an encoded segment differs in size from what the size kernel computed");
  return DX_OK;
}
