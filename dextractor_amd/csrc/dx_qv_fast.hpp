// dx_qv_fast.hpp -- k_qv_encode_fast: Compress_Next_QVentry (QV.c:1381-1426) from the token hand-over.
// (Textually included by dx_qv.hip after the encoder's building blocks; not a stand-alone header.)
//
// The run-coded lines of an entry arrive as dense 16-bit tokens made by k_qv_hist (see "token hand-over"
// in dx_qv.hip): Encode_Run (QV.c:475-497) is then a walk over an array -- no equality masks, no
// position lists, no second read of the deletion, tag and substitution lines -- and Pack_Tag +
// Number_Read + Compress_Read (QV.c:810-819, 1402-1404) is the 2-bit field already sitting in each
// deletion token.  The insertion and merge lines (always plain, QV.c:1417-1418) are read from the text.
// Every record is written where it belongs (sizes and offsets are known: k_qv_sizes_hist / k_qv_sizes_fast + the scan), and
// the sizes its segments turn out to have are compared with the given ones; entries whose tokens are marked unusable are
// left to the generic kernel.
// perturbation experiments (tools/microbench/hist_time.py --encode): parts of k_qv_encode_fast compiled out -- wrong
// output, a kernel time that says what the part costs.  1: no run-coded lines, 2: no plain lines, 4: bits are not placed
// in the window (plain lines), 8: one-symbol steps instead of pair tables
#ifndef FAST_SKIP
#define FAST_SKIP 0
#endif
#define TOK_TP 8u                                        // tokens a lane takes per pass (one 16-byte load)

struct tok_src
{ const uint16_t *del, *sub;
  const uint64_t *off;                                   // slot offsets (tokens), n + 1
  const uint32_t *info;                                  // n x 4 (see k_qv_hist)
};

// Encode_Run over the line's tokens.  Passes of up to 64 * TOK_TP tokens: a lane loads TOK_TP
// consecutive tokens with one 16-byte load, looks up run and symbol codes (shift tokens), chains them
// into one string of <= 128 bits, and a single prefix sum + placement per pass puts the strings into
// the window; with TAGS the lanes' 2-bit tag fields go into the tag window the same way.
// gix (group index, dx_qv_subindex): the line's header word; groups: one word per lane and pass.
// xend / nx: the line's exception list (runs of 127 and more: see "token hand-over"); XC = the line has exceptions
// (nx is wave-uniform: a line without any runs the instance without a single instruction for them).  With exceptions a lane finds the
// record of its first exception token by counting the exception tokens in front of it (the list is in token order); the records of the
// others among its tokens follow it in the list, and each run is swapped in for its field's 127.  (Round 5 sent a pass with two exceptions in one lane to the
// token-by-token path: rare at a run density of 0.85, three passes in four at 0.99, where 28 % of the tokens have a run of >= 127
// in front -- that batch's encoder took 15.7 ms instead of 10.)
// per: the tokens of a pass -- 64 * TOK_TP, or fewer (a multiple of 64) for a line whose codes are long: a lane's string has to
// fit 128 bits or the whole pass is placed token by token, and at 13 bits a token eight of them do not, in one lane of 64 or another.
template <bool TAGS, bool XC>
__device__ __forceinline__ void encode_token_line(wave_out &o, wave_out &ot, const uint16_t *tok, uint32_t cnt,
                                                  const uint32_t *ntab, const uint32_t *rtab,
                                                  const uint32_t *nstab, const uint32_t *rstab, uint32_t *gix, uint32_t *groups,
                                                  uint32_t *none_count, const uint32_t *xend, uint32_t nx, uint32_t per)
{ const uint32_t lane = (uint32_t) lane_id();
  uint32_t *g16 = gix ? groups + lane : (uint32_t *) NULL;
  uint32_t  wide = 0;                                              // a group that does not fit its 16 bits
  // a pass's tokens are requested a pass ahead, and the windows are drained at the start of a pass behind that request
  // (see FOR_EACH_ROUND_LATE): loads and stores then have a pass to complete in
#define TOK_PASS(K0, M_, T_, FIRST_, C_)                                                                   \
      const uint32_t M_     = cnt - (K0) < per ? cnt - (K0) : per;                                          \
      const uint32_t T_     = (M_ + 63u) >> 6;                     /* tokens per lane in this pass (wave-uniform) */ \
      const uint32_t FIRST_ = (K0) + lane * T_;                                                             \
      const uint32_t C_     = FIRST_ < (K0) + M_ ? ((K0) + M_ - FIRST_ < T_ ? (K0) + M_ - FIRST_ : T_) : 0u;
  u32x4 tw = { 0u, 0u, 0u, 0u };
  uint32_t xbase = 0;                                              // exception tokens of the passes before
  if (cnt)
    { TOK_PASS(0u, m0, T0, first0, c0)
      if (c0) tw = *(const u32x4_u *) (tok + first0);              // (may read up to 7 tokens past the count: inside the padded buffer)
    }
  for (uint32_t k0 = 0; k0 < cnt; k0 += per)
    { TOK_PASS(k0, m, T, first, c)
      u32x4 twn = { 0u, 0u, 0u, 0u };
      if (k0 + per < cnt)
        { TOK_PASS(k0 + per, mn, Tn, firstn, cn)
          if (cn) twn = *(const u32x4_u *) (tok + firstn);
        }
      if (o.winbits >= QV_FLUSH_BITS) flush_quads(o, false);
      if (TAGS && ot.winbits >= TAG_FLUSH_BITS) flush_quads(ot, true);
      uint32_t rt[TOK_TP], st[TOK_TP];
      #pragma unroll
      for (int k = 0; k < (int) TOK_TP; k++)                       // all look-ups first: one round of LDS waits per pass
        { const uint32_t t16 = (k & 1) ? chunk_word(tw, k >> 1) >> 16 : chunk_word(tw, k >> 1) & 0xffffu;
          rt[k] = rstab[t16 >> 9];                                 // QV.c:479-487 (runs below TOK_RUN_MAX: no clamp needed)
          st[k] = *(const uint32_t *) ((const uint8_t *) nstab + (t16 & 0x1fcu));
        }
      uint32_t xmask = 0, xi0 = 0, xtra = 0;                       // this lane's exception tokens: which; the first one's record; their runs beyond 127
#define TOK_XRUN(k) (*(xend - 2 * (int) (xi0 + (uint32_t) __builtin_popcount(xmask & ((1u << (k)) - 1u)) < nx ? xi0 + (uint32_t) __builtin_popcount(xmask & ((1u << (k)) - 1u)) : nx - 1u) - 1))
      if (XC)
        {
          #pragma unroll
          for (int k = 0; k < (int) TOK_TP; k++)
            { const uint32_t t16 = (k & 1) ? chunk_word(tw, k >> 1) >> 16 : chunk_word(tw, k >> 1) & 0xffffu;
              if ((uint32_t) k < c && (t16 >> 9) == TOK_RUN_MAX) xmask |= 1u << k;
            }
          // (the list is in token order and so are the lanes: a lane's first record is the number of exception tokens in front of it --
          //  a prefix sum over the wave, not a bisection's seven dependent loads)
          { const uint32_t xc = (uint32_t) __builtin_popcount(xmask), xin = wave_incl_scan(xc);
            xi0 = xbase + xin - xc;
            xbase += wave_total(xin);
          }
          if (xmask)
            {
              #pragma unroll
              for (int k = 0; k < (int) TOK_TP; k++)
                if ((xmask >> k) & 1u)
                  { const uint32_t xr = TOK_XRUN(k);
                    rt[k] = rstab[xr > 255u ? 255u : xr];          // QV.c:479-482
                    xtra += xr - TOK_RUN_MAX;
                  }
            }
        }
      // the lengths first: a wave none of whose lanes has more than 96 bits (and no run with a literal) chains into three words, not four
      uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, nb = 0, zor = 0, tacc = 0, span = 0;
      #pragma unroll
      for (int k = 0; k < (int) TOK_TP; k++)
        if ((uint32_t) k < c)
          { const uint32_t t16 = (k & 1) ? chunk_word(tw, k >> 1) >> 16 : chunk_word(tw, k >> 1) & 0xffffu;
            span += (t16 >> 9) + 1u;
            nb   += 64u - (rt[k] & 0x3fu) - (st[k] & 0xffu) + ((rt[k] & 0x80u) ? 16u : 0u);
            zor  |= rt[k] | st[k];
            if (TAGS)
              tacc = (tacc << 2) | (t16 & 3u);
          }
      span += xtra;
      const bool narrow96 = FAST_CHAIN96 && !__any((int) ((zor & 0x80u) | (nb > 96u)));
      if (narrow96)
        {
          #pragma unroll
          for (int k = 0; k < (int) TOK_TP; k++)
            if ((uint32_t) k < c)
              { STOK_APPEND3(rt[k])
                STOK_APPEND3(st[k])
              }
        }
      else
        {
          #pragma unroll
          for (int k = 0; k < (int) TOK_TP; k++)
            if ((uint32_t) k < c)
              { const uint32_t t16 = (k & 1) ? chunk_word(tw, k >> 1) >> 16 : chunk_word(tw, k >> 1) & 0xffffu;
                STOK_APPEND(rt[k])
                if (rt[k] & 0x80u)                                 // escaped run: its 16-bit literal follows (QV.c:486-487)
                  { const uint32_t run = XC && ((xmask >> k) & 1u) ? TOK_XRUN(k) : t16 >> 9;
                    const uint32_t lit = (run << 16) | 16u;
                    STOK_APPEND(lit)
                    w0 |= run & 0xffff0000u;                       // OCODE(16, run) with run >= 2^16 (QV.c:411, 420)
                  }
                STOK_APPEND(st[k])
              }
        }
      const uint32_t incl = wave_incl_scan(nb);
      if (g16)
        { *g16 = nb | (span << 16);
          g16 += 64;
          wide |= ((zor & 32u) || span > 0xffffu || wave_total(incl) > RUN_PASSBITS) ? 1u : 0u;
        }
      if (!__any((int) ((zor & 96u) | (nb > 128u))))
        { if (narrow96)
            { FOR_ONE_ROUND(o, incl, nb,
                { place_bits96(o.win, bit_, nb, w0, w1, w2); })
            }
          else
            { FOR_ONE_ROUND(o, incl, nb,
                { place_bits128(o.win, bit_, nb, w0, w1, w2, w3); })
            }
        }
      else                                    // a symbol or run without a code, or a string > 128 bits
        { FOR_EACH_ROUND_LATE(o, incl, nb,
            { bit_acc s;
              acc_begin(s, bit_);
              _Pragma("unroll 1")
              for (uint32_t j = 0; j < c; j++)
                { const uint32_t t16 = tok[first + j];
                  uint32_t run = t16 >> 9;
                  if (XC && run == TOK_RUN_MAX) run = tok_exception(xend, nx, first + j);
                  const uint32_t re  = rtab[run > 255u ? 255u : run];
                  const uint32_t se  = ntab[(t16 >> 2) & 0x7fu];
                  acc_put(s, o.win, TOK_ESC(re) ? ((TOK_BITS(re) << 16) | run) : TOK_BITS(re),
                          TOK_LEN(re) + (TOK_ESC(re) ? 16u : 0u));
                  acc_put(s, o.win, TOK_BITS(se), TOK_LEN(se));
                }
              acc_end(s, o.win);
            })
        }
      if (TAGS)
        { if (c)
            { const uint32_t bit = ot.winbits + 2u * (first - k0);
              const uint32_t w = bit >> 5, sh = bit & 31u;
              const uint32_t v = tacc << (32u - 2u * c);
              atomicOr(&ot.win[w], v >> sh);
              if (sh + 2u * c > 32u)
                atomicOr(&ot.win[w + 1], v << (32u - sh));
            }
          ot.winbits += 2u * m;
        }
      tw = twn;
    }
#undef TOK_PASS
#undef TOK_XRUN
  if (gix)
    { const bool none = __any((int) wide) != 0;
      if (lane == 0)
        { gix[0] = none ? RUN_NONE : cnt;
          if (none) atomicAdd(none_count, 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  plain lines two symbols per look-up
// ---------------------------------------------------------------------------------------------
// The insertion and merge QVs of a file sit in a narrow band of byte values (a few dozen symbols).  When a
// line's coded symbols span at most 64 values starting at `lo` (<= 192), a 4096-entry table per line holds,
// for every ordered pair of them, the two codes already chained into one shift token (if they have <= 24
// bits together): half the look-ups and half the chain steps of Encode (QV.c:427-434) per byte.  The table
// is built from the shift tokens by every workgroup at its start.  Pair index of bytes (b0, b1) in stream
// order: (b0 - lo) + 64 * (b1 - lo).  A step with a byte outside the band, a pair without a token or a
// lane string beyond 128 bits is handed to the one-symbol step.
// LDS banks: a look-up is serviced 32 lanes at a time over 32 banks of 4 bytes, one cycle per distinct address on a
// bank.  Laid out as (b0 - lo) + 64 * (b1 - lo), an entry's bank is (b0 - lo) mod 32 whatever b1 is: with the
// insertion QVs' geometric distribution a quarter of the lanes have the same b0 and three or four different b1, and
// nearly every look-up took three or four passes (round 2: 279 M conflict cycles on 159 M cycles of LDS work).  With a
// row stride of 69 the bank is (b0 + 5 b1) mod 32: the thirty most frequent pairs sit in thirty different banks, and
// lanes with the SAME pair read one address, which is a broadcast.  (69 * 63 + 63 < 65536: both 16-bit halves of a word
// still index independently, by one multiply-add instead of the shift and the OR.)
#define PAIR_NONE 0xffffffffu
#ifndef PAIR_STRIDE
#define PAIR_STRIDE 69u
#endif
#define PAIR_SIZE (((63u * PAIR_STRIDE + 64u) + 15u) & ~15u)

__device__ __forceinline__ void build_pair_tables(uint32_t (*s_pair)[PAIR_SIZE], const uint32_t (*s_stok)[256],
                                                  uint32_t lo_ins, uint32_t lo_mrg)
{ for (uint32_t k = threadIdx.x; k < 2u * 4096u; k += blockDim.x)
    { const uint32_t q = 1u + (k >> 12), lo = q == 1u ? lo_ins : lo_mrg;
      const uint32_t ia = k & 63u, ib = (k >> 6) & 63u, idx = (q - 1u) * PAIR_SIZE + ia + PAIR_STRIDE * ib;
      const uint32_t a = lo + ia, b = lo + ib;
      uint32_t tok = 32u;                                            // "no code": the step goes to the one-symbol path
      if (lo != PAIR_NONE && a < 256u && b < 256u)
        { const uint32_t t1 = s_stok[q][a], t2 = s_stok[q][b];
          const uint32_t s1 = t1 & 0xffu, s2 = t2 & 0xffu;           // 32 - length each
          if (s1 < 32u && s2 < 32u && s1 + s2 >= 40u)                // both coded, <= 24 bits together
            tok = (t1 & 0xffffff00u) | ((t2 & 0xffffff00u) >> (32u - s1)) | (s1 + s2 - 32u);
        }
      (&s_pair[0][0])[idx] = tok;
    }
  __syncthreads();
}

// one full step (16 bytes per lane) of Encode through the pair table; false: not applicable to this step
__device__ __forceinline__ bool encode_plain_step_pair(wave_out &o, const u32x4 &c, const uint32_t *ptab, uint32_t lo4, uint32_t m4,
                                                       sub_mark &sm)
{ uint32_t tok[8];
  uint32_t ssum = 0, zor = 0, bad = 0;
  #pragma unroll
  for (int w = 0; w < 4; w++)
    { const uint32_t x = (chunk_word(c, w) & m4) - lo4;              // bytes - lo: all < 64 in the band (no borrow then)
#if PAIR_STRIDE == 64
      const uint32_t m = (x & 0x003f003fu) | ((x >> 2) & 0x0fc00fc0u);   // two 12-bit pair indices, one per half word
#else
      const uint32_t m = __umul24((x >> 8) & 0x003f003fu, PAIR_STRIDE) + (x & 0x003f003fu);   // two pair indices b0 + 69 b1, one per half word
#endif
      bad |= x;
      tok[2 * w]     = ptab[m & 0xffffu];
      tok[2 * w + 1] = ptab[m >> 16];
    }
  #pragma unroll
  for (int k = 0; k < 8; k++)
    { ssum += tok[k] & 0xffu;
      zor  |= tok[k];
    }
  const uint32_t nb = 256u - ssum;
  if (__any((int) ((bad & 0xc0c0c0c0u) | (zor & 32u) | (nb > 128u))))
    return false;
  const uint32_t incl = wave_incl_scan(nb);
  sub_step(sm, nb, 16u, false);
  if (FAST_CHAIN96 && !__any((int) (nb > 96u)))
    { FOR_ONE_ROUND(o, incl, nb,
        { uint32_t w0 = 0, w1 = 0, w2 = 0;
          _Pragma("unroll")
          for (int k = 0; k < 8; k++)
            STOK_APPEND3(tok[k])
          place_bits96(o.win, bit_, nb, w0, w1, w2);
        })
      return true;
    }
  FOR_ONE_ROUND(o, incl, nb,
    { uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
      _Pragma("unroll")
      for (int k = 0; k < 8; k++)
        STOK_APPEND(tok[k])
      if (!(FAST_SKIP & 4) || w3 == 0x12345u)
        place_bits128(o.win, bit_, nb, w0, w1, w2, w3);
    })
  return true;
}

// ... and a line's last step, `valid` (0..16) bytes per lane: whole pairs through the pair table, an odd last byte through the
// one-symbol table, a dummy of one zero bit for every pair behind the line's end (shifted out again at the string's end, as
// in encode_plain_step).  The last step used to go to the one-symbol step altogether: 1.7 times a pair step's time -- per
// line, which at 2 kb is every other step.
__device__ __forceinline__ bool encode_plain_step_pair_last(wave_out &o, const u32x4 &c, int valid, const uint32_t *ptab, const uint32_t *stab,
                                                            uint32_t lo4, uint32_t m4, sub_mark &sm)
{ uint32_t tok[8];
  uint32_t ssum = 0, zor = 0, bad = 0;
  #pragma unroll
  for (int w = 0; w < 4; w++)
    { const int      n  = valid - 4 * w;                                  // this word's bytes inside the line
      const uint32_t vm = n >= 4 ? ~0u : (n > 0 ? (1u << (8 * n)) - 1u : 0u);
      const uint32_t cw = chunk_word(c, w) & m4;
      const uint32_t x  = (cw - lo4) & vm;                                // (bytes behind the end: index 0)
      const uint32_t m  = __umul24((x >> 8) & 0x003f003fu, PAIR_STRIDE) + (x & 0x003f003fu);
      bad |= x;
      const uint32_t t0 = ptab[m & 0xffffu], t1 = ptab[m >> 16];
      const uint32_t s0 = stab[cw & 0xffu], s1 = stab[(cw >> 16) & 0xffu];    // the pair's first byte alone
      tok[2 * w]     = n >= 2 ? t0 : (n == 1 ? s0 : STOK_DUMMY);
      tok[2 * w + 1] = n >= 4 ? t1 : (n == 3 ? s1 : STOK_DUMMY);
    }
  #pragma unroll
  for (int k = 0; k < 8; k++)
    { ssum += 2 * k < valid ? tok[k] & 0xffu : 32u;                       // (a dummy: no bit of the line's)
      zor  |= tok[k];                                                    // (bit 5: a pair or a lone byte without a code)
    }
  const uint32_t k  = 8u - (((uint32_t) valid + 1u) >> 1);                // dummies
  const uint32_t nb = 256u - ssum;
  if (__any((int) ((bad & 0xc0c0c0c0u) | (zor & 32u) | (nb + k > 128u))))
    return false;
  const uint32_t incl = wave_incl_scan(nb);
  sub_step(sm, nb, (uint32_t) valid, false);
  FOR_ONE_ROUND(o, incl, nb,
    { uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
      _Pragma("unroll")
      for (int j = 0; j < 8; j++)
        STOK_APPEND(tok[j])
      w0 = __builtin_amdgcn_alignbit(w1, w0, k);
      w1 = __builtin_amdgcn_alignbit(w2, w1, k);
      w2 = __builtin_amdgcn_alignbit(w3, w2, k);
      w3 >>= k;
      place_bits128(o.win, bit_, nb, w0, w1, w2, w3);
    })
  return true;
}

// a 64-bit value of lane j (uniform j).  The builtin returns int: the low half must not be sign-extended into the high one.
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t j)
{ const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, (int) j);
  const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), (int) j);
  return ((uint64_t) hi << 32) | lo;
}

__device__ __forceinline__ uint32_t token_bits(const uint16_t *tok, uint32_t cnt, const uint8_t *slen, const uint8_t *rlen,
                                               const uint32_t *xend, uint32_t nx)
{ const uint32_t lane = (uint32_t) lane_id();
  uint32_t acc = 0;
  // exception tokens (runs of 127 and more) were priced as run 127 below: swap in the real run's price
  for (uint32_t j = lane; j < nx; j += 64u)
    { const uint32_t run = *(xend - 2 * (int) j - 1);
      acc += (uint32_t) rlen[run > 255u ? 255u : run] - (uint32_t) rlen[TOK_RUN_MAX];
    }
  // (two passes' worth of tokens requested before either is looked at: with one request under way per wave the kernel ran at
  //  the memory's latency -- 10 ms for the 28 GB of tokens of a batch of run density 0.3)
  for (uint32_t k0 = 0; k0 < cnt; k0 += 2u * 64u * TOK_TP)
    { const uint32_t fa = k0 + lane * TOK_TP, fb = fa + 64u * TOK_TP;
      const uint32_t ca = fa < cnt ? (cnt - fa < TOK_TP ? cnt - fa : TOK_TP) : 0u;
      const uint32_t cb = fb < cnt ? (cnt - fb < TOK_TP ? cnt - fb : TOK_TP) : 0u;
      u32x4 ta = { 0u, 0u, 0u, 0u }, tb = ta;
      if (ca) ta = *(const u32x4_u *) (tok + fa);
      if (cb) tb = *(const u32x4_u *) (tok + fb);
      #pragma unroll
      for (int k = 0; k < (int) TOK_TP; k++)
        { const uint32_t t16 = (k & 1) ? chunk_word(ta, k >> 1) >> 16 : chunk_word(ta, k >> 1) & 0xffffu;
          const uint32_t l   = (uint32_t) rlen[t16 >> 9] + (uint32_t) slen[(t16 >> 2) & 0x7fu];
          acc += (uint32_t) k < ca ? l : 0u;
        }
      #pragma unroll
      for (int k = 0; k < (int) TOK_TP; k++)
        { const uint32_t t16 = (k & 1) ? chunk_word(tb, k >> 1) >> 16 : chunk_word(tb, k >> 1) & 0xffffu;
          const uint32_t l   = (uint32_t) rlen[t16 >> 9] + (uint32_t) slen[(t16 >> 2) & 0x7fu];
          acc += (uint32_t) k < cb ? l : 0u;
        }
    }
  return acc;                                            // (per lane; < 2^32 for entries of < 2^27 symbols)
}


// the five segment sizes of one entry from its tokens and its plain lines (what k_qv_sizes_fast stores): sz[0..4] = del words, tag bytes, ins, mrg, sub words
__device__ __forceinline__ void entry_sizes_fast(const qv_args &a, uint64_t r, uint32_t L, const uint32_t *inf, uint64_t toff, uint64_t tend,
                                                 const tok_src &tk, const uint32_t (*s_tok)[256], const uint8_t (*s_len)[256],
                                                 bool over, uint32_t *sz)
{ const int lane = lane_id();
  sz[0] = sz[2] = sz[3] = sz[4] = 0;
  sz[1] = (L + 3u) >> 2;                                 // all tags kept unless the deletion line is run-coded
  #pragma unroll 1
  for (int q = 0; q < 4; q++)
    { const int       line = q ? q + 1 : 0;
      const int       rci  = q == 0 ? a.delChar : (q == 3 ? a.subChar : -1);
      const uint32_t *tab  = s_tok[q];
      uint64_t T;
      uint32_t last;
      if (rci >= 0)                                      // Encode_Run: token lengths (QV.c:475-497)
        { const int       rs  = q == 0 ? DX_DRUN : DX_SRUN;
          const uint16_t *tok = (q == 0 ? tk.del : tk.sub) + toff;
          const uint32_t  cnt = inf[q == 0 ? 0 : 1] & ~TOK_BAD, C = inf[q == 0 ? 2 : 3];
          T = wave_sum(token_bits(tok, cnt, s_len[q], s_len[rs],
                                  (const uint32_t *) ((q == 0 ? tk.del : tk.sub) + tend), inf[q == 0 ? 4 : 5]));
          if (C > 0)                                     // run-only token at the line's end
            { const uint32_t e = s_tok[rs][C > 255u ? 255u : C];
              T   += TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
              last = TOK_ESC(e) ? 16u : TOK_LEN(e);
            }
          else if (cnt > 0)
            { const uint32_t e = tab[((uint32_t) tok[cnt - 1] >> 2) & 0x7fu];
              last = TOK_ESC(e) ? 8u : TOK_LEN(e);
            }
          else
            last = 0;
          if (q == 0) sz[1] = (cnt + 3u) >> 2;           // Pack_Tag's count, QV.c:810-819
        }
      else                                               // Encode: code lengths of the line's bytes (QV.c:427-434)
        { const uint8_t *p    = line_ptr(a, r, L, line);
          const uint32_t mask = !a.lossy ? 0xffu : (q == 1 ? 0xfeu : (q == 2 ? 0xfcu : 0xffu));
          const uint32_t m4   = mask * 0x01010101u;
          uint32_t pos = 16u * lane, acc = 0;
          u32x4 c = fetch(p, pos, L, over);
          T = 0;
          for (uint32_t base = 0; base < L; base += DX_STEP)
            { const u32x4 d = fetch_step(p, base + DX_STEP, L, over);
              acc += bits_syms_step(c, valid_of(pos, L), s_len[q], m4);
              c = d;
              pos += DX_STEP;
              if ((base & 0x3ffffffu) == 0x3fffc00u)     // fold long before a 32-bit lane sum can wrap
                { T += wave_sum(acc); acc = 0; }
            }
          T   += wave_sum(acc);
          last = last_piece_plain(tab, p, L, mask);
        }
      sz[line] = seg_bytes(T, last);
    }
}

#ifndef FAST_GUARDS
#define FAST_GUARDS 0
#endif
#ifndef FAST_TICKET
#define FAST_TICKET 2u                                   // entries a wave draws at a time
#endif
#ifndef FAST_WAVES
#define FAST_WAVES 4                                     // waves per SIMD the register allocation leaves room for
#endif
#ifndef FAST_BLOCK
#define FAST_BLOCK 512                                   // 8 waves share the 44 KB of tables: two workgroups per CU
#endif
#define FAST_NWAVE (FAST_BLOCK / 64)

// SUB: with the group index (dx_qv_subindex).  Two instances because of registers: the compaction of the group
// before runs BESIDE this kernel (k_qv_compact: 52 VGPRs -> 56 allocated), and its waves only find room on a SIMD
// whose four encoder waves leave 64 of the 512 registers, i.e. at <= 112 each.  The plain instance has 110; with the
// index code compiled in it had 116 (-> 120 allocated) even when no index was asked for, the two kernels ran one
// after the other, and a step took 32.7 ms instead of 31.0.
#ifndef FAST_VGPR_CAP
#define FAST_VGPR_CAP __attribute__((amdgpu_num_vgpr(112)))
#endif
// What the encoder has to know of an entry before it can ask for a byte of it -- its length, where its text, its tokens, its record,
// its header lie, its five sizes, the hist pass's six words about its tokens: 26 words from six arrays.  Asked
// for one after the other where they are needed they were a chain of six dependent waits at every entry's start (tokens usable? ->
// record inside the output? -> length and offsets -> header offsets -> header bytes ...), 3 us of an entry's 11 (1 M entries of 10 kb:
// 2.7 of the kernel's 11.2 ms; 2 M of 2 kb: 5.4 of 9.5).  Now lane k < 26 asks for word k -- ONE load instruction, the arrays' bases
// and strides from a table in LDS -- and asks an entry ahead: by the time the wave gets there the words have long arrived, and each is
// read where it is needed by v_readlane.
#define META_INFO  0      // .. 5: the info words 0 .. 5
#define META_LEN   6
#define META_OFF   7      // 7, 8: a.off[r]
#define META_TOFF  9      // 9 .. 12: tk.off[r], tk.off[r + 1]
#define META_SEG  13      // .. 17
#define META_REC  18      // 18 .. 21: rec_off[r], rec_off[r + 1]
#define META_HDR  22      // 22 .. 25: hdr_off[r], hdr_off[r + 1]
#define META_WORDS 26
template <bool SUB>
__global__ __launch_bounds__(FAST_BLOCK, FAST_WAVES) FAST_VGPR_CAP
void k_qv_encode_fast(qv_args a, const uint32_t *g_tok, const uint64_t *hdr_off, uint32_t *status, uint32_t *ticket,
                      tok_src tk, uint32_t pair_lo_ins, uint32_t pair_lo_mrg,
                      const uint8_t *hdr, const uint64_t *rec_off, const uint32_t *seg, uint8_t *out, uint64_t out_cap,
                      sub_sink sx)
{ __shared__ uint32_t s_tok[6][256];
  __shared__ uint32_t s_stok[6][256];
  __shared__ uint32_t s_pair[2][PAIR_SIZE];
  __shared__ uint8_t  s_tagcode[256];
  __shared__ __attribute__((aligned(16))) uint32_t s_win[FAST_NWAVE][QV_WIN_PAD + QV_WIN_WORDS];   // (the pad: see place_bits128)
  __shared__ __attribute__((aligned(16))) uint32_t s_tag[FAST_NWAVE][TAG_WIN_WORDS];
  __shared__ __attribute__((aligned(16))) meta_row s_meta[META_WORDS];
  if (threadIdx.x < META_WORDS)
    { const uint32_t k = threadIdx.x;
      meta_row m;
      if      (k < META_LEN)  { m.base = (const uint8_t *) tk.info + 4u * k;               m.stride = 4u * TOK_INFO; }
      else if (k == META_LEN) { m.base = (const uint8_t *) a.len;                         m.stride = 4u; }
      else if (k < META_TOFF) { m.base = (const uint8_t *) a.off + 4u * (k - META_OFF);   m.stride = 8u; }
      else if (k < META_SEG)  { m.base = (const uint8_t *) tk.off + 4u * (k - META_TOFF); m.stride = 8u; }
      else if (k < META_REC)  { m.base = (const uint8_t *) seg + 4u * (k - META_SEG);     m.stride = 20u; }
      else if (k < META_HDR)  { m.base = (const uint8_t *) rec_off + 4u * (k - META_REC); m.stride = 8u; }
      else                    { m.base = (const uint8_t *) hdr_off + 4u * (k - META_HDR); m.stride = 8u; }
      if (k >= META_HDR && (hdr == NULL || hdr_off == NULL))
        { m.base = (const uint8_t *) a.len; m.stride = 0u; }          // (no such array in this launch: any word)
      m.pad = 0;
      s_meta[k] = m;
    }
  load_tables(s_tok, g_tok);
  load_shift_tables(s_stok, s_tagcode, g_tok);
  build_pair_tables(s_pair, s_stok, pair_lo_ins, pair_lo_mrg);
  const int lane = lane_id();
  const int wid  = threadIdx.x >> 6;

  wave_out o, ot;
  o.win  = s_win[wid] + QV_WIN_PAD;
  ot.win = s_tag[wid];
  for (int j = lane; j < QV_WIN_WORDS; j += 64)  o.win[j]  = 0;
  for (int j = lane; j < TAG_WIN_WORDS; j += 64) ot.win[j] = 0;
  wave_sync();

  // Entries are drawn a few at a time (a.units): every draw is an atomic on ONE address, ~11 ns each chip-wide whoever asks
  // (a kernel doing nothing but drawing its 500 k tickets takes 5.8 ms), so a million single draws are 11 ms of the
  // counter's time inside a 13.7 ms kernel.
  const uint32_t TB = a.units;
  uint32_t mpre = 0;                                     // the words of entry mfor (see meta_load)
  uint64_t mfor = ~0ull;
  for (uint64_t r0 = next_unit(ticket, TB), nxt = 0; r0 < a.n; r0 = nxt)
  { nxt = next_unit(ticket, TB);
  for (uint64_t r = r0; r < r0 + TB && r < a.n; r++)
    { // (SUB, the instance that also writes the group index, has no register left for any of this: it asks as it goes, as before)
      uint32_t mw = 0;
      if (!SUB)
        { mw = mfor == r ? mpre : meta_load(s_meta, META_WORDS, r);   // (not asked for ahead: a wave's first entry)
          const uint64_t rn = r + 1 < r0 + TB && r + 1 < a.n ? r + 1 : nxt;
          mfor = rn;
          if (rn < a.n) mpre = meta_load(s_meta, META_WORDS, rn);
        }
#define INF(k)  (SUB ? tk.info[TOK_INFO * r + (k)] : META(mw, META_INFO + (k)))
#define SG(k)   (SUB ? seg[5u * r + (k)] : META(mw, META_SEG + (k)))
      if ((a.delChar >= 0 && (INF(0) & TOK_BAD)) || (a.subChar >= 0 && (INF(1) & TOK_BAD)))       // tok_unusable
        continue;                                        // the generic kernel encodes this entry from the text
      // The sizes are known (k_qv_sizes_hist, k_qv_sizes_fast): the record is written where it belongs -- framing bytes, del
      // words, tag bytes, ins, mrg, sub words (QV.c:1393-1423) -- and every size is checked.
      const uint32_t  L      = SUB ? a.len[r] : META(mw, META_LEN);
      const uint64_t  toff   = SUB ? tk.off[r] : META64(mw, META_TOFF);
      const uint8_t  *ebase  = a.text + (SUB ? a.off[r] : META64(mw, META_OFF));
#define tend (SUB ? tk.off[r + 1] : META64(mw, META_TOFF + 2))
#define ELINE(k) (ebase + (uint64_t) (k) * ((uint64_t) L + a.pad))            /* line_ptr(a, r, L, k) */
#if FAST_GUARDS
      { // the entry inside the text, its token slot as k_tok_rooms laid it out, the counts inside the slot: else report and
        // skip (see entry_sane).  Off by default: the branch makes the wave wait for ALL of an entry's index words before its
        // first data load goes out -- one more memory round trip per entry, 0.6 ms of the 13.7 a 1 M-entry batch takes
        // (measured); the generic kernel (the odd entries) has them
        const uint64_t room = tend - toff;
        const bool ok = entry_sane(a, r, L) && room >= 64u + TOK_XMARGIN &&
                        (INF(0) & ~TOK_BAD) <= room && (INF(1) & ~TOK_BAD) <= room && INF(4) <= room / 4u && INF(5) <= room / 4u;
        if (!ok)
          { if (lane == 0) atomicOr(status, DX_ST_INDEX);
            continue;
          }
      }
#endif
      uint8_t        *dst, *tag_at;
      if ((SUB ? rec_off[r + 1] : META64(mw, META_REC + 2)) > out_cap)            // d_out is too small: report, never overrun
        { if (lane == 0) atomicOr(status, 8u);
          continue;
        }
      dst = out + (SUB ? rec_off[r] : META64(mw, META_REC));
      if (hdr != NULL)                                   // record framing (dexqv.c:128-139)
        { const uint64_t h0 = SUB ? hdr_off[r] : META64(mw, META_HDR);
          const uint32_t hl = (uint32_t) ((SUB ? hdr_off[r + 1] : META64(mw, META_HDR + 2)) - h0);
          for (uint32_t k = (uint32_t) lane; k < hl; k += 64)
            dst[k] = hdr[h0 + k];
          dst += hl;
        }
      tag_at = dst + SG(0);
      const uint8_t  *p1     = ELINE(1);
      const bool      over   = can_overread(a, ELINE(4), L);
      uint32_t        bad    = 0;

      // The four QV streams in file order: del (its tag segment goes to the slot's end), ins, mrg, sub
      // (QV.c:1393-1423).
      #pragma unroll 1
      for (int q = 0; q < 4; q++)
        { const int       line = q ? q + 1 : 0;
          const int       rci  = q == 0 ? a.delChar : (q == 3 ? a.subChar : -1);
          const uint32_t *tab  = s_tok[q];
          o.seg = dst; o.wordbase = 0; o.winbits = 0;
          uint32_t got;
          if ((FAST_SKIP & 1) && rci >= 0) { got = 0; }
          else if ((FAST_SKIP & 2) && rci < 0) { got = 0; }
          else
          if (rci >= 0)                                  // Encode_Run from the tokens; for del also the tags
            { const int       rs   = q == 0 ? DX_DRUN : DX_SRUN;
              const uint32_t *rtab = s_tok[rs];
              const uint16_t *tok  = (q == 0 ? tk.del : tk.sub) + toff;
              const uint32_t  cnt  = INF(q == 0 ? 0 : 1) & ~TOK_BAD;
              const uint32_t  C    = INF(q == 0 ? 2 : 3);           // run left open at the line's end
              ot.seg = tag_at; ot.wordbase = 0; ot.winbits = 0;
              uint32_t *gix = NULL, *grp = NULL;           // group index: header word and group words of this line
              if (SUB && sx.idx)
                { uint32_t *base = sx.idx + sx.off[r] + run_base(L);
                  const uint32_t pd = (INF(0) & TOK_BAD) ? 0u : run_passes(INF(0));     // (as k_sub_rooms laid it out)
                  gix = base + (q == 0 ? 0u : 1u);
                  grp = base + 3u + (q == 0 ? 0u : 64u * pd);
                  if (lane == 0) base[2] = pd;
                }
              const uint32_t *xend = (const uint32_t *) ((q == 0 ? tk.del : tk.sub) + tend);      // the slot's end
              const uint32_t  nx   = INF(q == 0 ? 4 : 5);
              // tokens a pass: eight a lane while the line's codes average <= 10 bits a token (its size is known), four up to 20, else
              // two (see encode_token_line); the group index is laid out for eight
              const uint32_t  lb   = SG(line);
              const uint32_t  per  = gix != NULL || 4u * lb <= 5u * cnt ? 64u * TOK_TP : (2u * lb <= 5u * cnt ? 32u * TOK_TP : 16u * TOK_TP);
              if (nx == 0)
                { if (q == 0) encode_token_line<true,  false>(o, ot, tok, cnt, tab, rtab, s_stok[q], s_stok[rs], gix, grp, sx.none, xend, 0u, per);
                  else        encode_token_line<false, false>(o, ot, tok, cnt, tab, rtab, s_stok[q], s_stok[rs], gix, grp, sx.none, xend, 0u, per);
                }
              else
                { if (q == 0) encode_token_line<true,  true >(o, ot, tok, cnt, tab, rtab, s_stok[q], s_stok[rs], gix, grp, sx.none, xend, nx, per);
                  else        encode_token_line<false, true >(o, ot, tok, cnt, tab, rtab, s_stok[q], s_stok[rs], gix, grp, sx.none, xend, nx, per);
                }
              uint32_t last;
              if (C > 0)
                last = encode_trailing_run(o, C, rtab);
              else if (cnt > 0)                          // the line ends in its last token's symbol
                { const uint32_t e = tab[((uint32_t) tok[cnt - 1] >> 2) & 0x7fu];
                  last = TOK_ESC(e) ? 8u : TOK_LEN(e);
                }
              else
                last = 0;                                // empty line
              got = finish_words(o, last);
              if (q == 0)
                { const uint32_t tb = finish_tags(ot);
                  const uint32_t w1 = SG(1);
                  bad |= tb ^ w1; dst += w1;
                }
            }
          else                                           // Encode
            { const uint8_t *p    = ELINE(line);
              const uint32_t mask = !a.lossy ? 0xffu : (q == 1 ? 0xfeu : (q == 2 ? 0xfcu : 0xffu));   // QV.c:1406-1415
              const uint32_t m4   = mask * 0x01010101u;
              uint32_t pos = 16u * lane;
              u32x4 c = fetch(p, pos, L, over);
              // (wanted at the line's end, requested now: a memory round trip less per line)
              const uint32_t lastb = L ? (uint32_t) p[L - 1] : 0u;
#define PLAIN_LOOP(STAB)                                                                        \
              for (uint32_t base = 0; base < L; base += DX_STEP)                                 \
                { const u32x4 d = fetch_step(p, base + DX_STEP, L, over);                        \
                  if (o.winbits >= QV_FLUSH_BITS) flush_quads(o, false);   /* behind the load: see FOR_EACH_ROUND_LATE */ \
                  encode_plain_step<true>(o, c, valid_of(pos, L), L - base >= DX_STEP, tab, STAB, m4, sm); \
                  c = d;                                                                         \
                  pos += DX_STEP;                                                                \
                }
#define PAIR_LOOP(STAB, PTAB, LO)                                                               \
              { const uint32_t lo4 = (LO) * 0x01010101u;                                         \
                for (uint32_t base = 0; base < L; base += DX_STEP)                               \
                  { const u32x4 d = fetch_step(p, base + DX_STEP, L, over);                      \
                    const bool full = L - base >= DX_STEP;                                       \
                    if (o.winbits >= QV_FLUSH_BITS) flush_quads(o, false);   /* behind the load */ \
                    if ((FAST_SKIP & 32) && !full) { } else                                   \
                    if (full ? !encode_plain_step_pair(o, c, PTAB, lo4, m4, sm)                  \
                             : !encode_plain_step_pair_last(o, c, valid_of(pos, L), PTAB, STAB, lo4, m4, sm)) \
                      encode_plain_step<true>(o, c, valid_of(pos, L), full, tab, STAB, m4, sm);  \
                    c = d;                                                                       \
                    pos += DX_STEP;                                                              \
                  }                                                                              \
              }
              sub_mark sm;
              sub_begin(sm, SUB && sx.idx ? sx.idx + sx.off[r] + (uint64_t) q * sub_words(L) : (uint32_t *) NULL);
              if (q == 1)
                { if (!(FAST_SKIP & 8) && pair_lo_ins != PAIR_NONE) PAIR_LOOP(s_stok[1], s_pair[0], pair_lo_ins)
                  else                          { PLAIN_LOOP(s_stok[1]) }
                }
              else if (q == 2)
                { if (!(FAST_SKIP & 8) && pair_lo_mrg != PAIR_NONE) PAIR_LOOP(s_stok[2], s_pair[1], pair_lo_mrg)
                  else                          { PLAIN_LOOP(s_stok[2]) }
                }
              else             { PLAIN_LOOP(s_stok[q]) }
#undef PAIR_LOOP
#undef PLAIN_LOOP
              if (FAST_SKIP & 16) got = 4u * o.wordbase;
              else
              got = finish_words(o, last_piece_byte(tab, lastb, L, mask));
              if (q == 0)                                // no delChar: the whole tag line is packed
                { ot.seg = tag_at; ot.wordbase = 0; ot.winbits = 0;
                  const uint32_t tb = encode_all_tags(ot, p1, L, over);
                  const uint32_t w1 = SG(1);
                  bad |= tb ^ w1; dst += w1;
                }
            }
          bad |= got ^ SG(line);
          dst += got;
        }
      if (bad && lane == 0 && !FAST_SKIP)
        atomicOr(status, 2u);                            // a size differs from what the size kernel computed
#undef SG
#undef ELINE
#undef INF
#undef tend
    }
  }
}


// ---------------------------------------------------------------------------------------------
//  k_qv_sizes_fast: the five segment sizes of every entry, from the tokens and the two plain lines
// ---------------------------------------------------------------------------------------------
// With the sizes known up front the encoder writes every record where it belongs and the scratch round trip
// (slots + compaction: 2 x the output in extra HBM traffic, and a compaction kernel competing with the encoder)
// is gone.  Sizes need code LENGTHS only: per token two byte-wide look-ups (run length table, symbol length
// table; both 256 B = conflict-free), per byte of the insertion / merge lines one; the pad rule QV.c:436-442 in
// closed form.  Reads the tokens (~0.7 B per base) and the two plain lines (2 B per base): a memory-bound kernel
// with a quarter of the encoder's instructions, run for group g + 1 beside the encoder of group g.
__global__ __launch_bounds__(FAST_BLOCK)
void k_qv_sizes_fast(qv_args a, const uint32_t *g_tok, const uint64_t *hdr_off, uint32_t *seg /* n x 5 */, uint32_t *rec_size,
                     uint32_t *ticket, tok_src tk)
{ __shared__ uint32_t  s_tok[6][256];
  __shared__ size_tabs s_t;
  load_tables(s_tok, g_tok);
  load_size_tables(s_t, g_tok, a.delChar, a.subChar);
  const int lane = lane_id();

  for (uint64_t r0 = next_unit(ticket, a.units), nxt; r0 < a.n; r0 = nxt)
  { nxt = next_unit(ticket, a.units);
    for (uint64_t r = r0; r < r0 + a.units && r < a.n; r++)
    { if (tok_unusable(tk.info, r, a.delChar, a.subChar))
        continue;                                        // k_qv_sizes (generic) has this entry
      const uint32_t L = a.len[r];
      uint32_t sz[5];
      entry_sizes_fast(a, r, L, tk.info + TOK_INFO * r, tk.off[r], tk.off[r + 1], tk, s_tok, s_t.len,
                       can_overread(a, line_ptr(a, r, L, 4), L), sz);
      if (lane == 0)
        { const uint32_t hl = hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u;
          uint32_t *sg = seg + 5 * r;
          sg[0] = sz[0]; sg[1] = sz[1]; sg[2] = sz[2]; sg[3] = sz[3]; sg[4] = sz[4];
          rec_size[r] = hl + sz[0] + sz[1] + sz[2] + sz[3] + sz[4];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
//  k_qv_sizes_hist: the five segment sizes of every entry from the entry's own histograms
// ---------------------------------------------------------------------------------------------
// k_qv_hist<true> leaves, per entry, how often every symbol occurs in its insertion and merge lines and every symbol and
// run length among the tokens of its deletion and substitution lines (768 counters of 16 bits).  Once the tables exist a
// line's bits are the dot product of its counters with the code lengths -- a wave per entry, two bins per lane, six sums --
// plus what the counters do not hold: the runs of 127 and more (the line's exception list), the run open at a line's end,
// and the pad rule (QV.c:436-442), which wants the last code's length.  1.5 KB read per entry where k_qv_sizes_fast read
// the entry's tokens and plain lines again (28 GB for the 1 M x 10 kb batch: 7 ms) -- and with the sizes known up front the
// encoder writes every record in place: no scratch slots, no compaction.
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_sizes_hist(qv_args a, const uint32_t *g_tok, const uint64_t *hdr_off, uint32_t *seg /* n x 5 */, uint32_t *rec_size,
                     tok_src tk, const uint32_t *eh, int sub_made /* the substitution run character the counters were made under */)
{ __shared__ uint32_t  s_tok[6][256];
  __shared__ size_tabs s_t;
  load_tables(s_tok, g_tok);
  load_size_tables(s_t, g_tok, a.delChar, a.subChar);
  const uint32_t lane = (uint32_t) lane_id();
  const uint32_t im = !a.lossy ? 0xffu : 0xfeu, mm = !a.lossy ? 0xffu : 0xfcu;         // QV.c:1406-1415
  const uint32_t s0 = 2u * lane, s1 = s0 + 1u;
  // this lane's two bins of every table: their code lengths (constant over the entries)
  const uint32_t l_ins0 = s_t.len[DX_INS][s0 & im], l_ins1 = s_t.len[DX_INS][s1 & im];
  const uint32_t l_mrg0 = s_t.len[DX_MRG][s0 & mm], l_mrg1 = s_t.len[DX_MRG][s1 & mm];
  const uint32_t l_del0 = s_t.len[DX_DEL][s0], l_del1 = s_t.len[DX_DEL][s1], l_sub0 = s_t.len[DX_SUB][s0], l_sub1 = s_t.len[DX_SUB][s1];
  const uint32_t l_dr0 = s_t.len[DX_DRUN][s0], l_dr1 = s_t.len[DX_DRUN][s1], l_sr0 = s_t.len[DX_SRUN][s0], l_sr1 = s_t.len[DX_SRUN][s1];

  // Create_QVcoding may drop the substitution run character the scan chose (QV.c:1044): the line is then coded plain, from the
  // text, and its bits are its token symbols' plus its run characters' -- as many as the line has symbols that are no tokens.
  const bool     sub_plain = a.subChar < 0 && sub_made >= 0;
  const uint32_t l_subrc   = sub_plain ? (uint32_t) s_t.len[DX_SUB][sub_made & 0xff] : 0u;
  const uint32_t need = ((l_ins0 | l_ins1) ? 1u : 0u) | ((l_mrg0 | l_mrg1) ? 2u : 0u) | ((l_del0 | l_del1) ? 4u : 0u) | ((l_sub0 | l_sub1) ? 8u : 0u) |
                        ((l_dr0 | l_dr1) ? 16u : 0u) | ((!sub_plain && (l_sr0 | l_sr1)) ? 32u : 0u);      // the tables of which this lane's counters count

  // Eight entries a time, dealt to the waves in turn (every entry costs the same: no ticket counter, whose draws -- 11 ns
  // each, chip-wide -- were nine tenths of this kernel's time).  What an entry's sizes need beside its counters -- length, token counts, open runs, the last
  // token of either run-coded line, the last byte of either plain one -- lane j fetches for entry j: eight entries' worth
  // in two dependent rounds of loads instead of sixteen; the dot products then go entry by entry over the whole wave, and
  // lane j works its entry's five sizes out of the sums.
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6), nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  for (uint64_t r0 = 8u * wave0; r0 < a.n; r0 += 8u * nwave)
  { const uint64_t r    = r0 + lane;
    const bool     mine = lane < 8u && r < a.n;
    uint32_t L = 0, i0 = TOK_BAD, i1 = TOK_BAD, C0 = 0, C4 = 0, nx0 = 0, nx4 = 0, hl = 0, nocnt = 0;
    uint64_t toff = 0, tend = 0;
    if (mine)
      { const uint32_t *inf = tk.info + TOK_INFO * r;
        L = a.len[r]; toff = tk.off[r]; tend = tk.off[r + 1];
        i0 = inf[0]; i1 = inf[1]; C0 = inf[2]; C4 = inf[3]; nx0 = inf[4]; nx4 = inf[5]; nocnt = inf[6];
        hl = hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u;
      }
    const bool     tokd = mine && !((a.delChar >= 0 && (i0 & TOK_BAD)) || (a.subChar >= 0 && (i1 & TOK_BAD)));   // (else: k_qv_sizes has this entry)
    const bool     use  = tokd && !(nocnt & 1u);
    const uint64_t longs = __ballot(tokd && (nocnt & 1u));   // entries of 2^16 symbols and more: from their tokens and text, below
    const uint64_t bytok = __ballot(use && (nocnt & 2u));    // no counters of the token lines (k_qv_hist<true, false>): from the tokens
    const uint32_t cnt0 = i0 & ~TOK_BAD, cnt4 = i1 & ~TOK_BAD;
    uint32_t lt0 = 0, lt4 = 0, lb2 = 0, lb3 = 0, lb4 = 0;
    if (use)
      { if (cnt0) lt0 = tk.del[toff + cnt0 - 1];
        if (cnt4) lt4 = tk.sub[toff + cnt4 - 1];
        if (L)    { lb2 = line_ptr(a, r, L, 2)[L - 1]; lb3 = line_ptr(a, r, L, 3)[L - 1]; lb4 = line_ptr(a, r, L, 4)[L - 1]; }
      }
    const uint64_t usable = __ballot(use);
    uint32_t T0 = 0, T2 = 0, T3 = 0, T4 = 0;             // this lane's entry: bits of del, ins, mrg, sub
    for (uint32_t j = 0; j < 8u && r0 + j < a.n; j++)
      { if (!((usable >> j) & 1ull)) continue;
        const uint32_t *e32 = eh + (r0 + j) * EH_WORDS;
        // (a lane whose two symbols have no code -- they occur nowhere in the file -- asks for nothing: of the twelve cache lines of
        //  an entry's counters the symbols of a .quiva file touch six or seven)
        uint32_t v[6];
        #pragma unroll
        for (int k = 0; k < 6; k++) v[k] = (need >> k) & 1u ? e32[64 * k + lane] : 0u;
        uint32_t b_del = (v[2] & 0xffffu) * l_del0 + (v[2] >> 16) * l_del1 + (v[4] & 0xffffu) * l_dr0 + (v[4] >> 16) * l_dr1;
        uint32_t b_sub = (v[3] & 0xffffu) * l_sub0 + (v[3] >> 16) * l_sub1;
        if (!sub_plain) b_sub += (v[5] & 0xffffu) * l_sr0 + (v[5] >> 16) * l_sr1;
        const uint32_t b_ins = (v[0] & 0xffffu) * l_ins0 + (v[0] >> 16) * l_ins1;
        const uint32_t b_mrg = (v[1] & 0xffffu) * l_mrg0 + (v[1] >> 16) * l_mrg1;
        const uint32_t xj0 = __builtin_amdgcn_readlane(nx0, (int) j), xj4 = __builtin_amdgcn_readlane(nx4, (int) j);
        if ((bytok >> j) & 1ull)                         // the token lines from their tokens (lengths by table, exceptions included)
          { const uint64_t to = readlane64(toff, j), te = readlane64(tend, j);
            const uint32_t cj0 = __builtin_amdgcn_readlane(cnt0, (int) j), cj4 = __builtin_amdgcn_readlane(cnt4, (int) j);
            b_del = token_bits(tk.del + to, cj0, s_t.len[DX_DEL], s_t.len[DX_DRUN], (const uint32_t *) (tk.del + te), xj0);
            if (!sub_plain)
              b_sub = token_bits(tk.sub + to, cj4, s_t.len[DX_SUB], s_t.len[DX_SRUN], (const uint32_t *) (tk.sub + te), xj4);
            else                                         // (a dropped run character: the symbols of the tokens, plain)
              { b_sub = 0;
                for (uint32_t k = lane; k < cj4; k += 64u)
                  b_sub += s_t.len[DX_SUB][((uint32_t) tk.sub[to + k] >> 2) & 0x7fu];
              }
          }
        else if (xj0 | xj4)                              // runs of 127 and more: in no counter (hist_runs_step took them out again)
          { const uint64_t te = readlane64(tend, j);
            const uint32_t *xd = (const uint32_t *) (tk.del + te), *xs = (const uint32_t *) (tk.sub + te);
            for (uint32_t k = lane; k < xj0; k += 64u)
              { const uint32_t run = *(xd - 2 * (int) k - 1);
                b_del += s_t.len[DX_DRUN][run > 255u ? 255u : run];
              }
            for (uint32_t k = lane; k < (sub_plain ? 0u : xj4); k += 64u)
              { const uint32_t run = *(xs - 2 * (int) k - 1);
                b_sub += s_t.len[DX_SRUN][run > 255u ? 255u : run];
              }
          }
        const uint32_t t0 = wave_sum(b_del), t2 = wave_sum(b_ins), t3 = wave_sum(b_mrg), t4 = wave_sum(b_sub);
        if (lane == j) { T0 = t0; T2 = t2; T3 = t3; T4 = t4; }
      }
    if (use)
      { uint32_t sz[5];
        uint64_t Td = T0, Ts = T4;
        uint32_t lastd, lasts;
        if (C0 > 0)                                      // run-only token at the line's end (QV.c:490-497)
          { const uint32_t e = s_tok[DX_DRUN][C0 > 255u ? 255u : C0];
            Td += TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
            lastd = TOK_ESC(e) ? 16u : TOK_LEN(e);
          }
        else if (cnt0 > 0)
          { const uint32_t e = s_tok[DX_DEL][(lt0 >> 2) & 0x7fu];
            lastd = TOK_ESC(e) ? 8u : TOK_LEN(e);
          }
        else
          lastd = 0;
        if (sub_plain)
          { Ts   += (uint64_t) (L - cnt4) * l_subrc;
            lasts = last_piece_byte(s_tok[DX_SUB], lb4, L, 0xffu);
          }
        else if (C4 > 0)
          { const uint32_t e = s_tok[DX_SRUN][C4 > 255u ? 255u : C4];
            Ts += TOK_LEN(e) + (TOK_ESC(e) ? 16u : 0u);
            lasts = TOK_ESC(e) ? 16u : TOK_LEN(e);
          }
        else if (cnt4 > 0)
          { const uint32_t e = s_tok[DX_SUB][(lt4 >> 2) & 0x7fu];
            lasts = TOK_ESC(e) ? 8u : TOK_LEN(e);
          }
        else
          lasts = 0;
        sz[0] = seg_bytes(Td, lastd);
        sz[1] = (cnt0 + 3u) >> 2;                        // Pack_Tag's count, QV.c:810-819
        sz[2] = seg_bytes(T2, last_piece_byte(s_tok[DX_INS], lb2, L, im));
        sz[3] = seg_bytes(T3, last_piece_byte(s_tok[DX_MRG], lb3, L, mm));
        sz[4] = seg_bytes(Ts, lasts);
        uint32_t *sg = seg + 5 * r;
        sg[0] = sz[0]; sg[1] = sz[1]; sg[2] = sz[2]; sg[3] = sz[3]; sg[4] = sz[4];
        rec_size[r] = hl + sz[0] + sz[1] + sz[2] + sz[3] + sz[4];
      }
    for (uint32_t j = 0; longs && j < 8u; j++)           // (rare: a .quiva entry seldom has 65536 symbols)
      if ((longs >> j) & 1ull)
        { const uint64_t rj = r0 + j;
          const uint32_t Lj = a.len[rj];
          uint32_t sz[5];
          entry_sizes_fast(a, rj, Lj, tk.info + TOK_INFO * rj, tk.off[rj], tk.off[rj + 1], tk, s_tok, s_t.len,
                           can_overread(a, line_ptr(a, rj, Lj, 4), Lj), sz);
          if (lane == 0)
            { const uint32_t hj = hdr_off ? (uint32_t) (hdr_off[rj + 1] - hdr_off[rj]) : 0u;
              uint32_t *sg = seg + 5 * rj;
              sg[0] = sz[0]; sg[1] = sz[1]; sg[2] = sz[2]; sg[3] = sz[3]; sg[4] = sz[4];
              rec_size[rj] = hj + sz[0] + sz[1] + sz[2] + sz[3] + sz[4];
            }
        }
  }
}
