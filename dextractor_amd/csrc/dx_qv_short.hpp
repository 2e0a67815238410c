// dx_qv_short.hpp -- dexqv for batches of SHORT entries: a lane per entry (included by dx_qv.hip, whose helpers it uses).
//
// The wave-per-entry kernels give an entry a whole wavefront: a step is 1 KiB of each of its lines, and an entry pays its fixed
// costs -- a ticket, its whereabouts, five first requests, 1.5 KB of counters out, five segment ends -- whatever its length.  At
// 300 symbols an entry fills a third of a step and the batch ran at 216 GB/s (4 M x 300: 27.8 ms for 6 GB); the round-4 verdict
// asked for several entries per wave.  Here: 64 entries per wave, a lane each, through the reference's own loops (QVcoding_Scan
// QV.c:922-1023 with Histogram_Seqs / Histogram_Runs :702-724; Compress_Next_QVentry QV.c:1381-1426 with Encode :386-443,
// Encode_Run :448-506, Pack_Tag :810-819 + Number_Read + Compress_Read).  Three passes over the text (histograms; sizes;
// records written where the sizes' prefix sums put them), as dx_qv_sizes / dx_qv_encode have it.
//
// What shapes the kernels (profiles/r05_short_entries.txt has the measurements, 4 M x 300, ms per step):
//   * a lane's line is 1.5 KB from its neighbour's: nothing coalesces, and a cache line asked for in two goes is fetched twice
//     (a CU's lanes hold 16 waves x 64 x 128 bytes: more than its L1, and with the other CUs' more than the L2).  A lane asks for the
//     whole aligned 128-byte line around its position in eight requests that go out together (15.8 -> 12.8);
//   * they are bound by their vector instructions -- a wave64 instruction takes a SIMD four cycles, 6.1e11 a second on the chip,
//     and the three kernels issue 4.6e9 -- so: the span is one 32-word vector and the piece at work is read out of it through
//     the register index (a loop counter, uniform), not unrolled sixteen times or chosen by selects; the look-ups of a piece go
//     out together; two symbols of a plain line are one piece of <= 32 bits for the bit buffer, a run's code and its symbol's
//     too; in a run-coded line only the symbols are walked (qs_others finds them eight bytes at a time);
//   * finished words leave through a ring in LDS sixteen bytes at a time (4-byte stores: 2.8 of the encoder's 7.4 ms);
//   * the lanes of a wave are at different places in their cache lines: of the byte steps a wave takes 69 % are some lane's
//     (300-symbol lines; 88 % at 1000).
//   * a wave takes as long as its longest entry: the 256 consecutive entries of a round are dealt to a workgroup's lanes in the order
//     of their lengths (k_qs_survey sorts once a batch and leaves a byte an entry), and what decides whether a batch is these kernels'
//     is what an entry COSTS them -- the longest entry of its wave, averaged over the batch -- not the mean length.
// 4 M x 300: 10.8 ms, 555 GB/s (k_qs_hist 1.9, k_qs_entries<sizes> 2.0, <records> 6.1).  Against the wave-per-entry kernels by
// mean length (2 M entries, GB/s): fixed lengths 600: 632 / 407, 1000: 689 / 647, 1200: 699 / 735; lognormal (sigma 0.35, a wave's
// longest 1.45 x the mean) 600: 494 / 431, 800: 497 / 533 -- taken for batches of >= 4096 entries that cost at most QS_MEAN symbols
// an entry and have none longer than QS_MAXLEN (a lane is alone with its entry); DEXGPU_NO_SHORT: never.  No tokens, no group
// index: the decoder reads such a stream with its generic kernel.
//
// Roofline: the vector ALUs, not the HBM: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs accounts for 1.5 of k_qs_hist's 1.9 ms, 2.1 of the
// sizes' 2.1 and 3.8 of the records' 6.1 (the rest of those: waits for the look-ups and for the next cache line, 4 waves a SIMD).
#ifndef QS_MEAN
#define QS_MEAN   1000u
#endif
#ifndef QS_BLOCK
#define QS_BLOCK  256
#endif
#define QS_COPIES 4u                    // copies of every histogram bin in a workgroup's LDS (copy = lane & 3)

#ifndef QS_W
#define QS_W      8
#endif
//                ^                    // 16-byte chunks a lane asks for at a time: one cache line of its line, whole and aligned (its neighbours'
                                        // lines are 1.5 KB away: a lane's requests share nothing with theirs, and a cache line asked for in two goes
                                        // is fetched twice -- 28 waves x 64 lanes x 128 bytes a CU outlive neither the 32 KB L1 nor a share of the L2)
#ifndef QS_MAXLEN
#define QS_MAXLEN 4096u
#endif
//                    ^                // ... and no entry longer than this (a lane is alone with its entry)


typedef __attribute__((address_space(3))) uint32_t qs_lds;        // (the tables and counters of a workgroup)
#define QS_LDS(p) ((qs_lds *) (p))
#define QS_ADD(p, v) ((void) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))

typedef __attribute__((address_space(1))) const u32x4_u qs_g128;  // (global_load / global_store, not flat)
typedef __attribute__((address_space(1))) u32_u qs_g32;
typedef __attribute__((address_space(1))) uint8_t qs_g8;
typedef uint32_t u32x32 __attribute__((ext_vector_type(4 * QS_W)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));

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
__device__ __forceinline__ u32x2 qs_fetch8(const uint8_t *q, const uint8_t *lo, const uint8_t *end16)
{ const uint8_t *qc = q < lo ? lo : q;
  u32x2 c = *(__attribute__((address_space(1))) const u32x2_u *) (qc < end16 ? qc : end16);
  if (q < lo || q > end16) { const u32x4 e = qs_edge(q, lo, end16 + 16); c = u32x2{ e.x, e.y }; }
  return c;
}
// the cache line around p + pos
__device__ __forceinline__ qs_span qs_fetch(const uint8_t *p, uint32_t pos, uint32_t L, const uint8_t *lo, const uint8_t *end16)
{ qs_span s;
  const uint8_t *q = p + pos;
  s.skip = (uint32_t) ((uintptr_t) q & (16u * QS_W - 1u));
  s.len  = L - pos < 16u * QS_W - s.skip ? L - pos : 16u * QS_W - s.skip;
  s.v    = qs_fetch_raw(q - s.skip, lo, end16);
  return s;
}
// a line's spans; a span's sixteen pieces of 8 bytes (k uniform: the piece is read out of the vector through the register index);
// a piece's bytes
#define QS_PIECES (2 * QS_W)
#define QS_SPANS(s, p, L, lo, end16) for (uint32_t pos = 0, step_ = 0; pos < (L); pos += step_) { qs_span s = qs_fetch(p, pos, L, lo, end16); step_ = (s).len;
#define QS_DEAD(s, k)      (8u * (k) + 8u <= (s).skip || 8u * (k) >= (s).skip + (s).len)
#define QS_CHUNKS(s, k, c) _Pragma("nounroll") for (int k = 0; k < QS_PIECES; k++) \
                             { if (QS_DEAD(s, k)) continue; \
                               const u32x2 c = { (s).v[2 * k], (s).v[2 * k + 1] };
#define QS_BYTES(b)        _Pragma("unroll") for (int b = 0; b < 8; b++)
#define QS_BYTE(v, b)      ((((b) < 4 ? (v).x : (v).y) >> (8 * ((b) & 3))) & 0xffu)
#define QS_LIVE(s, k, b)   ((uint32_t) (8 * (k) + (b)) - (s).skip < (s).len)

// The 64 lanes of a wave go through their entries in step: a wave takes as long as its longest entry.  The QS_BLOCK consecutive entries
// of a round are therefore dealt to a workgroup's lanes in the order of their lengths (a bitonic sort of length << 8 | place in LDS, once
// a batch: k_qs_survey leaves every entry's place, a byte, for the three kernels): a wave's 64 entries are then of a length (lognormal
// lengths, 2 M x 800: 450 -> 497 GB/s), and they are still neighbours in the file -- dealing the whole batch's entries by length and
// alignment costs the cache lines two entries share and the DRAM rows more than it gains (profiles/r05_short_entries.txt).
// Returns this thread's entry's place in the round (0xffffffff: the round has fewer entries); a round of one length: as it comes.
__device__ __forceinline__ uint32_t qs_order(qs_lds *s_key /* [QS_BLOCK] */, const uint32_t *len, uint64_t base, uint64_t n)
{ static_assert(QS_BLOCK <= 256, "qs_order: the place in the round has 8 bits");
  const uint32_t t = threadIdx.x;
  const uint32_t mine = base + t < n ? len[base + t] : 0xffffffffu;
  if (!__syncthreads_or((int) (mine != 0xffffffffu && mine != len[base])))     // (a round of one length: as it comes)
    return mine == 0xffffffffu ? mine : t;
  s_key[t] = mine == 0xffffffffu ? mine : (mine << 8) | t;
  __syncthreads();
  for (uint32_t k = 2; k <= QS_BLOCK; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1)
      { const uint32_t p = t ^ j;
        if (p > t)
          { const uint32_t x = s_key[t], y = s_key[p];
            if ((x > y) == ((t & k) == 0u)) { s_key[t] = y; s_key[p] = x; }
          }
        __syncthreads();
      }
  const uint32_t key = s_key[t];
  __syncthreads();                                         // (the next round writes the keys again)
  return key == 0xffffffffu ? key : key & 0xffu;
}

// Every entry's place among the QS_BLOCK of its round when they are taken by length (perm, a byte an entry: the three kernels read it
// instead of sorting again -- and without the sort's code they keep the registers of their fifth wave a SIMD), and what the batch
// would cost these kernels: work[0] = its longest entry, work[1] = its shortest, work[2..3] = the sum over the waves of the wave's
// longest entry (a wave's 64 lanes go through their entries in step): x 64 / n it is the length an entry costs.
// How the lanes are dealt a batch: perm (a byte an entry: its place among the 256 of its round when they are taken by length), order (the
// rounds, those with the longest entries first: a lane is alone with its entry, and the longest entry's lane should be the first to
// start, not the last), cut (entries longer than that are not the lanes': the wave-per-entry kernels take them as a batch of their own).
struct qs_deal { const uint8_t *perm; const uint32_t *order; uint32_t cut; };

// entries and symbols by length: bucket len / 64 (64 buckets up to QS_MAXLEN, one for everything longer): cnt[0..64], sym[0..64] (units of 64 symbols)
// (every `every`-th entry, counted `every` times: an estimate is all the cut wants)
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_lenhist(const uint32_t *len, uint64_t n, unsigned long long *cnt /* 65 + 65 */, uint32_t every)
{ __shared__ uint32_t s_c[65], s_s[65];
  if (threadIdx.x < 65) { s_c[threadIdx.x] = 0; s_s[threadIdx.x] = 0; }
  __syncthreads();
  for (uint64_t i = ((uint64_t) blockIdx.x * QS_BLOCK + threadIdx.x) * every; i < n; i += (uint64_t) gridDim.x * QS_BLOCK * every)
    { const uint32_t L = len[i], k = L > QS_MAXLEN ? 64u : (L ? (L - 1u) >> 6 : 0u);
      atomicAdd(&s_c[k], every);
      atomicAdd(&s_s[k], every * ((L + 63u) >> 6));
    }
  __syncthreads();
  if (threadIdx.x < 65)
    { if (s_c[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long) s_c[threadIdx.x]);
      if (s_s[threadIdx.x]) atomicAdd(&cnt[65 + threadIdx.x], (unsigned long long) s_s[threadIdx.x]);
    }
}

// Where the lanes stop: the longest entry they take is one lane's work from its first symbol to its last (~1.1 us a symbol when that lane
// is alone on its SIMD), and it must not outlast what all the lanes together take for the entries up to that length (the text at
// 2.5 TB/s in the fastest of the three kernels; the rounds with the longest entries go first) -- the largest multiple of 64 that does not.
__host__ __device__ static uint32_t qs_cut(const unsigned long long *cnt /* 65 entries + 65 x 64 symbols, by length / 64 */, uint64_t *entries_le)
{ uint64_t S = 0, N = 0;                                  // symbols and entries of the entries of at most 64 k symbols
  for (int k = 0; k < 64; k++) { N += cnt[k]; S += 64ull * cnt[65 + k]; }
  for (int k = 64; k >= 1; k--)
    { const double lanes_s = 5.0 * (double) S / 2.5e12, lone_s = 64.0 * k * 1.1e-6;
      if (N >= 4096 && lone_s <= (lanes_s > 1e-3 ? lanes_s : 1e-3)) { *entries_le = N; return 64u * (uint32_t) k; }      // (a millisecond at least: batches that small take no time either way)
      N -= cnt[k - 1]; S -= 64ull * cnt[65 + k - 1];
    }
  *entries_le = 0;
  return 0;
}

// ... decided where the counts are: *cut = the cut (forced: that value instead, the tests' way)
__global__ void k_qs_cut(const unsigned long long *cnt, uint32_t forced, uint32_t *cut)
{ if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t le;
  const uint32_t c = forced ? forced : qs_cut(cnt, &le);
  *cut = c > QS_MAXLEN ? QS_MAXLEN : c;
}

// Entries of more than `cut` symbols are none of these kernels' (a lane is alone with its entry): they go on a list -- work[4] counts
// them, list[] holds their indices, in no order -- and the wave-per-entry kernels take them as a batch of their own (a batch of short
// entries with some long ones among them: what real subread sets look like); the lanes they fall to stand idle, and none of the figures
// above counts them (work[0], the longest, is the longest SHORT entry; work[5] = their symbols in all, in units of 1024).
// rmax[round] = the round's longest short entry, work[6..7] their sum (k_qs_round_order).
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_survey(const uint32_t *len, uint64_t n, uint8_t *perm, uint32_t every /* one round in `every` counts for the sum, `every` times */, uint32_t *work,
                 uint32_t *list, uint32_t list_cap, const uint32_t *cut_at, uint32_t *rmax)
{ __shared__ uint32_t s_key[QS_BLOCK];
  const uint32_t cut = *cut_at;                            // (k_qs_cut's; 0: none of these entries is the lanes')
  if (cut == 0) return;
  __shared__ uint32_t s_max;
  if (threadIdx.x == 0) s_max = 0;
  const uint64_t base = (uint64_t) blockIdx.x * QS_BLOCK;
  const uint32_t place = qs_order(QS_LDS(s_key), len, base, n);      // (the t-th shortest of the round is this thread's)
  const bool     have  = place != 0xffffffffu;
  const uint32_t Lall  = have ? len[base + place] : 0u;
  const bool     big   = Lall > cut;
  const uint32_t L     = big ? 0u : Lall;
  if (have) perm[base + threadIdx.x] = (uint8_t) place;
  { const uint64_t m = __ballot(big);                      // (one draw a wave: atomics on one word are 10 ns each, whoever asks)
    if (m)
      { const uint32_t kib = wave_sum(big ? (Lall + 1023u) >> 10 : 0u);
        uint32_t at0 = 0;
        if (lane_id() == 0)
          { at0 = atomicAdd(work + 4, (uint32_t) __popcll(m));
            atomicAdd(work + 5, kib);
          }
        at0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) at0);
        const uint32_t at = at0 + (uint32_t) __popcll(m & ((1ull << lane_id()) - 1ull));
        if (big && at < list_cap) list[at] = (uint32_t) (base + place);
      }
  }
  const uint32_t wmax = wave_total(wave_incl_max(L));      // the wave's longest short entry: what its 64 lanes take
  if (blockIdx.x % every == 0u && (threadIdx.x & 63u) == 0u && wmax)
    atomicAdd((unsigned long long *) (work + 2), (unsigned long long) wmax * every);   // (62 500 atomics on one word are 0.6 ms: an estimate from a sixteenth)
  if ((threadIdx.x & 63u) == 0u && wmax > __hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(work, wmax);                                 // (checked first: atomics on one word are 10 ns each)
  if (threadIdx.x == 0u && have && !big && L < __hip_atomic_load(work + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMin(work + 1, L);
  if ((threadIdx.x & 63u) == 0u && wmax) atomicMax(&s_max, wmax);
  __syncthreads();
  if (threadIdx.x == 0u)
    { rmax[blockIdx.x] = s_max;
      if (blockIdx.x % every == 0u && s_max) atomicAdd((unsigned long long *) (work + 6), (unsigned long long) s_max * every);
    }
}

// The rounds as the lanes take them: those whose longest entry is well beyond the usual (more than one and a half times the rounds'
// average: work[6..7] holds their sum) first, the others behind them in the file's order (a lane is alone with its entry: the longest
// ones must be among the first to start; everything else reads the file front to back).  One workgroup.
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_round_order(const uint32_t *rmax, uint64_t nblk, const uint32_t *work, uint32_t *order)
{ __shared__ uint32_t s_heavy, s_wave[QS_BLOCK / 64];
  const unsigned long long sum = ((unsigned long long) work[7] << 32) | work[6];
  const uint32_t thr = (uint32_t) (3ull * sum / (2ull * (nblk ? nblk : 1)));
  if (threadIdx.x == 0) s_heavy = 0;
  __syncthreads();
  for (uint64_t k = threadIdx.x; k < nblk; k += QS_BLOCK)
    if (rmax[k] > thr) order[atomicAdd(&s_heavy, 1u)] = (uint32_t) k;
  __syncthreads();
  uint32_t running = s_heavy;
  for (uint64_t c0 = 0; c0 < nblk; c0 += QS_BLOCK)
    { const uint64_t k = c0 + threadIdx.x;
      const bool light = k < nblk && rmax[k] <= thr;
      const uint64_t m = __ballot(light);
      if (lane_id() == 0) s_wave[threadIdx.x >> 6] = (uint32_t) __popcll(m);
      __syncthreads();
      uint32_t before = 0, all = 0;
      for (uint32_t w = 0; w < QS_BLOCK / 64; w++) { if (w < (threadIdx.x >> 6)) before += s_wave[w]; all += s_wave[w]; }
      if (light) order[running + before + (uint32_t) __popcll(m & ((1ull << lane_id()) - 1ull))] = (uint32_t) k;
      running += all;
      __syncthreads();
    }
}

// the long entries as a batch of their own: their offsets and lengths side by side (the wave-per-entry kernels index a batch by r)
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_sub_batch(const uint64_t *off, const uint32_t *len, const uint32_t *list, uint64_t nl, uint64_t *off2, uint32_t *len2)
{ const uint64_t j = (uint64_t) blockIdx.x * QS_BLOCK + threadIdx.x;
  if (j < nl)
    { const uint32_t r = list[j];
      off2[j] = off[r]; len2[j] = len[r];
    }
}

// what the long entries' own kernels found, to the whole batch's arrays: the five segment sizes, and the record's size with its framing
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_scatter_sizes(const uint32_t *list, uint64_t nl, const uint32_t *seg2, const uint32_t *size2, const uint64_t *hdr_off,
                        uint32_t *seg, uint32_t *rec_size)
{ const uint64_t j = (uint64_t) blockIdx.x * QS_BLOCK + threadIdx.x;
  if (j >= nl) return;
  const uint32_t r = list[j];
  for (int k = 0; k < 5; k++) seg[5ull * r + k] = seg2[5 * j + k];
  rec_size[r] = size2[j] + (hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u);
}

// ... and back: where each long entry's payload goes (behind its framing bytes, which are written here)
__global__ __launch_bounds__(QS_BLOCK)
void k_qs_gather_places(const uint32_t *list, uint64_t nl, const uint64_t *rec_off, const uint8_t *hdr, const uint64_t *hdr_off,
                        uint8_t *out, uint64_t *rec2 /* nl + 1 */)
{ const uint64_t j = (uint64_t) blockIdx.x * QS_BLOCK + threadIdx.x;
  if (j > nl) return;
  if (j == nl) { rec2[j] = 0; return; }                  // (read by the encoder's bound test only, which is off: out_cap = ~0)
  const uint32_t r = list[j];
  uint64_t at = rec_off[r];
  if (hdr != NULL)
    { const uint64_t h0 = hdr_off[r], hl = hdr_off[r + 1] - h0;
      for (uint64_t k = 0; k < hl; k++) out[at + k] = hdr[h0 + k];
      at += hl;
    }
  rec2[j] = at;
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
void k_qs_hist(qv_args a, qs_deal deal, uint64_t entry0, long long del_first, long long sub_first, unsigned long long *hist,
               unsigned long long *tot)
{ __shared__ uint32_t s_h[6][256][QS_COPIES];             // 24 KB
  __shared__ uint32_t s_key[QS_BLOCK];
  for (uint32_t i = threadIdx.x; i < 6u * 256u * QS_COPIES; i += QS_BLOCK) (&s_h[0][0][0])[i] = 0u;
  __syncthreads();
  const uint32_t cp = threadIdx.x & (QS_COPIES - 1u);
  #define H(s) (QS_LDS(&s_h[s][0][0]) + cp)
  uint64_t chars = 0;
  for (uint64_t rk = blockIdx.x; rk * QS_BLOCK < a.n; rk += gridDim.x)
    { const uint64_t base = (deal.order ? (uint64_t) deal.order[rk] : rk) * QS_BLOCK;        // (the rounds with the longest entries first)
      if (base + threadIdx.x >= a.n) continue;
      const uint64_t r = base + (deal.perm ? (uint32_t) deal.perm[base + threadIdx.x] : threadIdx.x);   // (by length within the round: k_qs_survey)
      const uint32_t L = a.len[r];
      if (L > deal.cut) continue;                          // (on the long entries' list: the wave-per-entry kernels')
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
// MSB-first into 32-bit words as OCODE does (QV.c:404-422).  Sizes: T counts the bits.  Records: finished words go to the lane's
// column of a ring in LDS (QS_RING words; word j at ring[(j % QS_RING) * QS_BLOCK]: the lanes of a wave write side by side), and
// leave it sixteen bytes at a time at the end of every chunk -- a lane's records are its own (its neighbours' are elsewhere,
// nothing coalesces), and 4-byte stores cost the encoder 2.8 of its 7.4 ms.  A piece of 8 bytes adds at most 11 words to the at most 3
// that stayed: a run of 255 and more, which has a literal, can only end at a piece's first symbol (32 bits), the other runs in
// it are of 6 at most (7 codes of 16 bits), and eight symbols have 24 bits at most.
#ifndef QS_RING
#define QS_RING 16u
#endif
struct qs_bits { uint8_t *p; qs_lds *ring; uint64_t acc; uint32_t fill, T, wr, rd; };

__device__ __forceinline__ qs_bits qs_begin(uint8_t *dst, qs_lds *ring)
{ return qs_bits{ dst, ring, 0ull, 0u, 0u, 0u, 0u }; }
__device__ __forceinline__ void qs_word(qs_bits &w, uint32_t word)
{ w.ring[(w.wr & (QS_RING - 1u)) * QS_BLOCK] = word;
  w.wr += 1u;
}
__device__ __forceinline__ void qs_quads(qs_bits &w)
{ while (w.wr - w.rd >= 4u)
    { const uint32_t j = w.rd & (QS_RING - 1u);            // (a multiple of 4: the ring is read in fours only, until the line's end)
      const u32x4 v = { w.ring[j * QS_BLOCK], w.ring[(j + 1u) * QS_BLOCK], w.ring[(j + 2u) * QS_BLOCK], w.ring[(j + 3u) * QS_BLOCK] };
#ifndef QS_NOSTORE
      *(__attribute__((address_space(1))) u32x4_u *) w.p = v;
#endif
      w.p += 16; w.rd += 4u;
    }
}
// len bits (<= 32; 0: nothing), right-aligned in bits
template <bool EMIT>
__device__ __forceinline__ void qs_put(qs_bits &w, uint32_t len, uint32_t bits)
{ if (!EMIT) w.T += len;
  if (EMIT)
    { w.acc |= (uint64_t) bits << ((64u - len - w.fill) & 63u);         // (fill < 32: a shift of 1..63, or bits of no length)
      w.fill += len;
      if (w.fill >= 32u)
        { qs_word(w, (uint32_t) (w.acc >> 32));
          w.acc <<= 32; w.fill -= 32u;
        }
    }
}
// the segment's end (QV.c:436-442): its bytes.  last: the length of the last code appended (a literal counts as a code of its
// own: the pad rule looks at where the last OCODE began)
template <bool EMIT>
__device__ __forceinline__ uint32_t qs_finish(qs_bits &w, const uint8_t *dst, uint32_t last)
{ if (EMIT)
    { qs_quads(w);
      while (w.wr != w.rd)
        { *(qs_g32 *) w.p = w.ring[(w.rd & (QS_RING - 1u)) * QS_BLOCK];
          w.p += 4; w.rd += 1u;
        }
      w.T = (uint32_t) (w.p - dst) * 8u + w.fill;          // (the sizes count their bits, the records their words)
    }
  const uint32_t olen = w.T & 31u, llen = (w.T - last) & 31u;
  const uint32_t words = (w.T >> 5) + (olen ? 1u : 0u);
  const bool again = olen ? (llen > 16u && olen > llen) : (w.T > 0u && llen > 16u);
  if (EMIT)
    { const uint32_t part = (uint32_t) (w.acc >> 32);       // (0 when the last word was whole)
      if (olen) { *(qs_g32 *) w.p = part; w.p += 4; }
      if (again) { *(qs_g32 *) w.p = part; w.p += 4; }
    }
  return 4u * (words + (again ? 1u : 0u));
}

struct qs_ret { uint32_t bytes, n; };                   // the segment's bytes; the symbols under which a tag stands

// one plain line (Encode, QV.c:386-443); mask4: the lossy rounding of the insertion / merge QVs (QV.c:1406-1415), in all four
// bytes of a word.  A token holds a symbol's code and, for an escape, its 8-bit literal behind it (one OCODE each, QV.c:430-433:
// the last code is then the literal); a byte beyond the line is a token of no bits.
template <bool EMIT, bool WIDE>
__device__ __forceinline__ qs_ret qs_plain(const uint8_t *p, uint32_t L, const uint8_t *lo, const uint8_t *end16, const qs_lds *tab, uint32_t mask4,
                                           uint8_t *dst, qs_lds *ring)
{ qs_bits w = qs_begin(dst, ring);
  uint32_t tl = 0;                                        // the line's last token
  QS_SPANS(s, p, L, lo, end16)
      s.v &= mask4;
      QS_CHUNKS(s, k, c)
        uint32_t t[8];
        QS_BYTES(b) t[b] = tab[QS_BYTE(c, b)];            // (whatever byte: an address in the table; the eight look-ups go out together)
        QS_BYTES(b)
          { const bool live = QS_LIVE(s, k, b);
            t[b] = live ? t[b] : 0u;
            tl   = live ? t[b] : tl;
          }
        if (!EMIT)
          { QS_BYTES(b) w.T += TOK_LEN(t[b]); }
        else if (WIDE)                                    // (a table with escapes: a token of up to 24 bits at a time)
          { QS_BYTES(b) qs_put<EMIT>(w, TOK_LEN(t[b]), TOK_BITS(t[b])); }
        else                                              // codes of at most 16 bits: two symbols at a time
          { _Pragma("unroll")
            for (int b = 0; b < 8; b += 2)
              qs_put<EMIT>(w, TOK_LEN(t[b]) + TOK_LEN(t[b + 1]), (TOK_BITS(t[b]) << TOK_LEN(t[b + 1])) | TOK_BITS(t[b + 1]));
          }
        if (EMIT) qs_quads(w);
      }
    }
  return qs_ret{ qs_finish<EMIT>(w, dst, TOK_ESC(tl) ? 8u : TOK_LEN(tl)), L };
}

// 2-bit codes, four to a byte, the first in the top bits (Compress_Read DB.c:319-338): sixteen to a stored word
struct qs_tags { uint8_t *p; uint32_t acc, n; };
__device__ __forceinline__ void qs_tag(qs_tags &g, uint32_t letter, bool on)
{ g.acc = on ? (g.acc << 2) | tag_code(letter) : g.acc;
  g.n  += on;
  if (on && (g.n & 15u) == 0u)
    {
#ifndef QS_NOSTORE
      *(qs_g32 *) g.p = __builtin_bswap32(g.acc);
#endif
      g.p += 4;
    }
}
__device__ __forceinline__ void qs_tag_end(qs_tags &g)
{ const uint32_t left = g.n & 15u;
  if (left)
    { const uint32_t v = g.acc << (2u * (16u - left));
      for (uint32_t k = 0; 4u * k < left; k++) *(qs_g8 *) (g.p + k) = (uint8_t) (v >> (24u - 8u * k));
    }
}

// bit b: byte b of the piece is not the byte rc4 holds four times
__device__ __forceinline__ uint32_t qs_others(u32x2 c, uint32_t rc4)
{ const uint32_t t0 = c.x ^ rc4, t1 = c.y ^ rc4;
  const uint32_t n0 = ((((t0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t0) & 0x80808080u) >> 7;     // bits 0, 8, 16, 24: the byte is not zero
  const uint32_t n1 = ((((t1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t1) & 0x80808080u) >> 7;
  return (((n0 * 0x00204081u) >> 21) & 0xfu) | (((n1 * 0x00204081u) >> 17) & 0xf0u);
}
__device__ __forceinline__ uint32_t qs_byte_at(u32x2 c, uint32_t b)      // (b a lane's own: 0..7)
{ return ((b < 4u ? c.x : c.y) >> (8u * (b & 3u))) & 0xffu; }

// one run-coded line (Encode_Run, QV.c:448-506): before every symbol that is not the run character the code of the run that ended
// there (of 0 too), the longest code with a 16-bit literal behind it (QV.c:478-488), then the symbol's; a run at the line's end:
// its code alone.  TAGS: the deletion line -- the tags under its symbols are packed on the way (Pack_Tag QV.c:810-819,
// Number_Read DB.c:393-416), .n = how many.  Most bytes are the run character: a piece's symbols are found eight bytes at a time
// and only they are walked (the wave goes round as often as its lane with the most has symbols: 4 to 5 times in 8 bytes at the
// reference's run densities, not 8); the run's code and the symbol's are one piece of at most 32 bits unless a literal stands
// behind one of them.
template <bool EMIT, bool TAGS>
__device__ __forceinline__ qs_ret qs_runs(const uint8_t *p, const uint8_t *ptag, uint32_t L, const uint8_t *lo, const uint8_t *end16, const qs_lds *tab,
                                          const qs_lds *rtab, uint32_t rc, uint8_t *dst, uint8_t *tdst, qs_lds *ring)
{ qs_bits w = qs_begin(dst, ring);
  qs_tags g = { tdst, 0u, 0u };
  const u32x2 c_zero = { 0u, 0u };
  const uint32_t rc4 = rc * 0x01010101u;
  uint32_t run = 0, nsym = 0, tl = 0;                      // the open run; symbols so far; the last symbol's token
  QS_SPANS(s, p, L, lo, end16)
      // the tags under the span's bytes: the same positions of the tag line, a piece ahead of the piece at work
      const uint8_t *tq = ptag + pos - s.skip;
      u32x2 gnext = c_zero;
      if (TAGS && EMIT) gnext = qs_fetch8(tq, lo, end16);
      _Pragma("nounroll")
      for (int k = 0; k < QS_PIECES; k++)
        { const u32x2 gc = gnext;
          if (TAGS && EMIT && k + 1 < QS_PIECES) gnext = qs_fetch8(tq + 8 * (k + 1), lo, end16);
          if (QS_DEAD(s, k)) continue;
          const u32x2 c = { s.v[2 * k], s.v[2 * k + 1] };
          const uint32_t b0 = s.skip > 8u * k ? s.skip - 8u * k : 0u;                          // the line's bytes of the piece: [b0, b1)
          const uint32_t b1 = s.skip + s.len - 8u * k < 8u ? s.skip + s.len - 8u * k : 8u;
          uint32_t syms = qs_others(c, rc4) & ((1u << b1) - (1u << b0)), cur = b0;
          nsym += __builtin_popcount(syms);
          while (syms)
            { const uint32_t b = __builtin_ctz(syms);
              syms &= syms - 1u;
              run += b - cur; cur = b + 1u;
              const uint32_t tr = rtab[run > 255u ? 255u : run], t = tab[qs_byte_at(c, b)];
              if (!TOK_ESC(tr | t))
                qs_put<EMIT>(w, TOK_LEN(tr) + TOK_LEN(t), (TOK_BITS(tr) << TOK_LEN(t)) | TOK_BITS(t));
              else                                           // (rare: a run of 255 and more, or tables with escapes)
                { qs_put<EMIT>(w, TOK_LEN(tr), TOK_BITS(tr));
                  if (TOK_ESC(tr)) qs_put<EMIT>(w, 16u, run & 0xffffu);
                  qs_put<EMIT>(w, TOK_LEN(t), TOK_BITS(t));
                }
              tl = t; run = 0u;
              if (TAGS && EMIT) qs_tag(g, qs_byte_at(gc, b), true);
            }
          run += b1 - cur;
          if (EMIT) qs_quads(w);
        }
    }
  uint32_t last = TOK_ESC(tl) ? 8u : TOK_LEN(tl);
  if (run)                                                  // (a line that ends in a run: its code, no symbol behind it)
    { const uint32_t t = rtab[run > 255u ? 255u : run];
      qs_put<EMIT>(w, TOK_LEN(t), TOK_BITS(t));
      last = TOK_LEN(t);
      if (TOK_ESC(t)) { qs_put<EMIT>(w, 16u, run & 0xffffu); last = 16u; }
    }
  if (TAGS && EMIT) qs_tag_end(g);
  return qs_ret{ qs_finish<EMIT>(w, dst, last), nsym };
}

// the whole tag line packed (no deletion run character: QV.c:1393-1396)
__device__ __forceinline__ void qs_tags_all(const uint8_t *ptag, uint32_t L, const uint8_t *lo, const uint8_t *end16, uint8_t *tdst)
{ qs_tags g = { tdst, 0u, 0u };
  QS_SPANS(s, ptag, L, lo, end16)
      QS_CHUNKS(s, k, c)
        QS_BYTES(b) qs_tag(g, QS_BYTE(c, b), QS_LIVE(s, k, b));
      }
    }
  qs_tag_end(g);
}

// EMIT = false: seg[5 r ..] and rec_size[r] (dx_qv_sizes' contract); EMIT = true: the record at rec_off[r], its segment sizes
// compared with seg (status bit 1: they differ).  A symbol the tables have
// no code for costs no bits, as in the reference (Encode's OCODE of length 0) and in k_qv_encode
#ifdef QS_OCC
#define QS_WAVES __attribute__((amdgpu_waves_per_eu(QS_OCC)))
#else
#define QS_WAVES
#endif
template <bool EMIT, bool WIDE>
__global__ __launch_bounds__(QS_BLOCK) QS_WAVES
void k_qs_entries(qv_args a, qs_deal deal, const uint32_t *g_tok, const uint8_t *hdr, const uint64_t *hdr_off, const uint64_t *rec_off,
                  uint32_t *seg, uint32_t *rec_size, uint8_t *out, uint32_t *status)
{ __shared__ uint32_t s_tok[6][256];
  __shared__ uint32_t s_ring[EMIT ? QS_RING : 1u][QS_BLOCK];
  load_tables(s_tok, g_tok);
  qs_lds *ring = QS_LDS(&s_ring[0][threadIdx.x]);
  const uint32_t imask = a.lossy ? 0xfefefefeu : ~0u, mmask = a.lossy ? 0xfcfcfcfcu : ~0u;
  uint32_t differ = 0;
  for (uint64_t rk = blockIdx.x; rk * QS_BLOCK < a.n; rk += gridDim.x)
    { const uint64_t base = (deal.order ? (uint64_t) deal.order[rk] : rk) * QS_BLOCK;        // (the rounds with the longest entries first)
      if (base + threadIdx.x >= a.n) continue;
      const uint64_t r = base + (deal.perm ? (uint32_t) deal.perm[base + threadIdx.x] : threadIdx.x);   // (by length within the round: k_qs_survey)
      const uint32_t L = a.len[r];
      if (L > deal.cut) continue;                          // (on the long entries' list: the wave-per-entry kernels')
      const uint8_t *p0 = line_ptr(a, r, L, 0);              // (the other lines' addresses where they are wanted: five pointers held are ten registers)
#define p1 (p0 + ((uint64_t) L + a.pad))
#define p2 (p0 + 2u * ((uint64_t) L + a.pad))
#define p3 (p0 + 3u * ((uint64_t) L + a.pad))
#define p4 (p0 + 4u * ((uint64_t) L + a.pad))
      const uint8_t *end16 = a.text + a.text_bytes - 16u;    // (text_bytes known and large: qs_short)
      const uint32_t hl = hdr_off ? (uint32_t) (hdr_off[r + 1] - hdr_off[r]) : 0u;
      uint32_t want[5] = { 0u, 0u, 0u, 0u, 0u }, clen = L;
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
        { q = qs_runs<EMIT, true>(p0, p1, L, a.text, end16, QS_LDS(s_tok[DX_DEL]), QS_LDS(s_tok[DX_DRUN]), (uint32_t) a.delChar & 0xffu, d0, d1, ring);
          clen = q.n;
        }
      else
        { q = qs_plain<EMIT, WIDE>(p0, L, a.text, end16, QS_LDS(s_tok[DX_DEL]), ~0u, d0, ring);
          if (EMIT) qs_tags_all(p1, L, a.text, end16, d1);
        }
      // a size is noted (sizes) or compared (records) as soon as it is known: five of them held to the entry's end are five registers
      uint32_t sum = hl;
#define QS_SEG(k, v) { const uint32_t v_ = (v); if (EMIT) differ |= v_ ^ want[k]; else { seg[5 * r + (k)] = v_; sum += v_; } }
      QS_SEG(0, q.bytes)
      QS_SEG(1, (clen + 3u) >> 2)
      q = qs_plain<EMIT, WIDE>(p2, L, a.text, end16, QS_LDS(s_tok[DX_INS]), imask, d2, ring);
      QS_SEG(2, q.bytes)
      q = qs_plain<EMIT, WIDE>(p3, L, a.text, end16, QS_LDS(s_tok[DX_MRG]), mmask, d3, ring);
      QS_SEG(3, q.bytes)
      if (a.subChar >= 0)
        q = qs_runs<EMIT, false>(p4, p4, L, a.text, end16, QS_LDS(s_tok[DX_SUB]), QS_LDS(s_tok[DX_SRUN]), (uint32_t) a.subChar & 0xffu, d4, d4, ring);
      else
        q = qs_plain<EMIT, WIDE>(p4, L, a.text, end16, QS_LDS(s_tok[DX_SUB]), ~0u, d4, ring);
      QS_SEG(4, q.bytes)
#undef QS_SEG
      if (!EMIT) rec_size[r] = sum;
    }
#undef p1
#undef p2
#undef p3
#undef p4
  if (differ) atomicOr(status, 2u);
}

// ---------------------------------------------------------------------------------------------
//  host side
// ---------------------------------------------------------------------------------------------
static int qs_grid(const dx_ctx *ctx, uint64_t n)
{ const uint64_t need = (n + QS_BLOCK - 1) / QS_BLOCK, room = (uint64_t) ctx->num_cu * 32u;
  return (int) (need < room ? need : room);
}

// Is this a batch for the lane-per-entry kernels?  What an entry costs them is the length of the longest entry of its wave
// (k_qs_survey): a batch is theirs when that, averaged over the entries, is at most QS_MEAN -- fixed lengths: the mean itself,
// 1100 symbols being where the wave-per-entry kernels take over; lognormal lengths (sigma 0.35): 1.45 x the mean, and measured
// (2 M entries, GB/s with / without): mean 600: 493 / 431, mean 800: 497 / 533.  Entries of more than QS_MAXLEN symbols (a lane
// takes ~2.4 us a symbol through the three kernels: one entry of 4096 is 10 ms of one lane) are not the lanes': a batch that has
// some -- `mixed` -- is dealt, the short entries to the lanes (which skip the long ones), the long ones as a batch of their own
// (sub) to the wave-per-entry kernels, when at least 4096 short ones are left and they cost what a batch of short ones may.
// fresh (dx_qv_hist, the first to see a batch): looked at anew; the others take what the context remembers of a batch of these
// arrays and sizes -- were the arrays' contents changed in between, the verdict is the old contents': slow at worst, not wrong.
struct qs_verdict { bool brief, mixed; qs_deal deal; dx_qv_batch sub; const uint32_t *list; };

static void qs_free(dx_ctx *ctx)
{ (void) hipFree(ctx->qs.perm); (void) hipFree(ctx->qs.list); (void) hipFree(ctx->qs.off2); (void) hipFree(ctx->qs.len2);
  (void) hipFree(ctx->qs.order); (void) hipFree(ctx->qs.rmax); (void) hipFree(ctx->qs.aux);
  ctx->qs.perm = NULL; ctx->qs.list = NULL; ctx->qs.off2 = NULL; ctx->qs.len2 = NULL; ctx->qs.order = NULL; ctx->qs.rmax = NULL; ctx->qs.aux = NULL;
  ctx->qs.cap = 0;
}

static int qs_short(dx_ctx *ctx, const dx_qv_batch *b, bool fresh, qs_verdict *v)
{ v->brief = v->mixed = false; v->deal = qs_deal{ NULL, NULL, QS_MAXLEN }; v->list = NULL;
  memset(&v->sub, 0, sizeof(v->sub));
  if (dx_test_on("no_short") || b->n < 4096 || b->text_bytes == 0) return DX_OK;
  if (b->text_bytes / b->n > 5ull * (QS_MAXLEN + 1u) + 64u) return DX_OK;          // (a mean beyond what a lane takes at all: nothing to look at)
  if (!(!fresh && ctx->qs.valid && ctx->qs.off == (const void *) b->d_off && ctx->qs.len == (const void *) b->d_len && ctx->qs.n == b->n &&
        ctx->qs.text_bytes == b->text_bytes))
    { ctx->qs.valid = 0;
      const uint64_t nblk = (b->n + QS_BLOCK - 1) / QS_BLOCK;
      if (ctx->qs.cap < b->n)
        { qs_free(ctx);
          if (hipMalloc((void **) &ctx->qs.perm, b->n + 256) != hipSuccess || hipMalloc((void **) &ctx->qs.list, b->n * 4 + 64) != hipSuccess ||
              hipMalloc((void **) &ctx->qs.off2, b->n * 8 + 64) != hipSuccess || hipMalloc((void **) &ctx->qs.len2, b->n * 4 + 64) != hipSuccess ||
              hipMalloc((void **) &ctx->qs.order, nblk * 4 + 64) != hipSuccess || hipMalloc((void **) &ctx->qs.rmax, nblk * 4 + 64) != hipSuccess ||
              hipMalloc((void **) &ctx->qs.aux, 256 * 8) != hipSuccess)
            { (void) hipGetLastError();
              qs_free(ctx);
              return DX_OK;                                // (no memory for a few bytes an entry: the wave-per-entry kernels)
            }
          ctx->qs.cap = b->n;
        }
      // the lengths first (a sample of them): where the lanes stop -- decided on the device, read back with the survey's figures
      uint32_t *d_work = (uint32_t *) (ctx->d_u64 + 40), *d_cut = (uint32_t *) (ctx->qs.aux + 200);
      uint32_t  work[8] = { 0u, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u }, cut = 0;
      const uint32_t every = b->n >= (1u << 18) ? 16u : 1u;
      const uint64_t hb = (b->n / every + QS_BLOCK - 1) / QS_BLOCK;
      DX_HIP(ctx, hipMemsetAsync(ctx->qs.aux, 0, 256 * 8, ctx->stream));
      DX_HIP(ctx, hipMemcpyAsync(d_work, work, 32, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_qs_lenhist, dim3((unsigned) (hb < 1024 ? (hb ? hb : 1) : 1024)), dim3(QS_BLOCK), 0, ctx->stream, (const uint32_t *) b->d_len, b->n, ctx->qs.aux, every);
      hipLaunchKernelGGL(k_qs_cut, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long *) ctx->qs.aux,
                         dx_test_on("short_force") ? (uint32_t) dx_test_num("short_cut", QS_MAXLEN) : 0u, d_cut);
      hipLaunchKernelGGL(k_qs_survey, dim3((unsigned) nblk), dim3(QS_BLOCK), 0, ctx->stream, (const uint32_t *) b->d_len, b->n,
                         ctx->qs.perm, nblk >= 1024 ? 16u : 1u, d_work, ctx->qs.list, (uint32_t) b->n, (const uint32_t *) d_cut, ctx->qs.rmax);
      hipLaunchKernelGGL(k_qs_round_order, dim3(1), dim3(QS_BLOCK), 0, ctx->stream, (const uint32_t *) ctx->qs.rmax, nblk, (const uint32_t *) d_work, ctx->qs.order);
      DX_HIP(ctx, hipMemcpyAsync(work, d_work, 24, hipMemcpyDeviceToHost, ctx->stream));
      DX_HIP(ctx, hipMemcpyAsync(&cut, d_cut, 4, hipMemcpyDeviceToHost, ctx->stream));
      DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ctx->qs.off = b->d_off; ctx->qs.len = b->d_len; ctx->qs.n = b->n; ctx->qs.text_bytes = b->text_bytes;
      ctx->qs.brief = 0; ctx->qs.mixed = 0; ctx->qs.nl = 0; ctx->qs.cut = cut;
      if (cut > 0)
        {
          const uint64_t cost = ((uint64_t) work[3] << 32) | work[2], nl = work[4], ns = b->n - nl;
          const uint64_t long_bytes = 5ull * 1024u * work[5];                            // (the long entries' share of the text, rounded up)
          const uint64_t short_bytes = b->text_bytes > long_bytes ? b->text_bytes - long_bytes : 0;
          // theirs when what an entry costs them -- the longest entry of its wave, averaged -- is at most QS_MEAN symbols
          bool yes = ns >= 4096 && 64u * cost <= (uint64_t) QS_MEAN * ns && short_bytes / ns <= 5ull * (QS_MEAN + 1u) + 64u;
          if (dx_test_on("short_force")) yes = ns >= 1;
          if (dx_test_on("no_mixed") && nl) yes = false;
          ctx->qs.ordered = work[0] != work[1] && !dx_test_on("short_file_order");  // (entries of one length: as they come)
          ctx->qs.brief = yes ? 1 : 0;
          ctx->qs.mixed = yes && nl ? 1 : 0;
          ctx->qs.nl = nl;
          if (ctx->qs.mixed)
            { hipLaunchKernelGGL(k_qs_sub_batch, dim3((unsigned) ((nl + QS_BLOCK - 1) / QS_BLOCK)), dim3(QS_BLOCK), 0, ctx->stream,
                                 (const uint64_t *) b->d_off, (const uint32_t *) b->d_len, (const uint32_t *) ctx->qs.list, nl, ctx->qs.off2, ctx->qs.len2);
              DX_HIP(ctx, hipGetLastError());
            }
        }
      ctx->qs.valid = 1;
    }
  v->brief = ctx->qs.brief != 0;
  v->mixed = ctx->qs.mixed != 0;
  v->deal  = qs_deal{ v->brief && ctx->qs.ordered ? ctx->qs.perm : (const uint8_t *) NULL,
                      v->brief && ctx->qs.ordered ? ctx->qs.order : (const uint32_t *) NULL, ctx->qs.cut };
  if (v->mixed)
    { v->sub = *b;
      v->sub.d_off = ctx->qs.off2; v->sub.d_len = ctx->qs.len2; v->sub.n = ctx->qs.nl;
      v->list = ctx->qs.list;
    }
  return DX_OK;
}

// sizes, offsets, records: dx_qv_encode_onepass's contract
static int onepass_short(dx_ctx *ctx, const dx_qv_batch *b, qs_deal perm, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
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
  DX_LAUNCH(ctx, DX_K_QV_SIZES, (k_qs_entries<false, false>), qs_grid(ctx, n), QS_BLOCK, a, perm, (const uint32_t *) ctx->d_tok, (const uint8_t *) NULL, d_hdr_off,
            (const uint64_t *) NULL, d_seg, d_size, (uint8_t *) NULL, ctx->d_status);
  uint64_t tot = 0;
  if ((e = dx_scan_u32(ctx, d_size, n, d_rec_off, &tot))) return e;
  if (total) *total = tot;
  ctx->route.groups = 0; ctx->route.direct = 3; ctx->route.tokens = 0; ctx->route.region_bytes = 0;
  ctx->route.scratch_bytes = ctx->scratch_bytes; ctx->route.token_bytes = 0; ctx->route.text_entries = n;
  if (tot > out_cap)
    return dx_fail(ctx, DX_E_SPACE, "dx_qv_encode_onepass: the record stream needs %llu bytes, d_out holds %llu",
                   (unsigned long long) tot, (unsigned long long) out_cap);
  if (ctx->tok_wide)
    DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, (k_qs_entries<true, true>), qs_grid(ctx, n), QS_BLOCK, a, perm, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
              (const uint64_t *) d_rec_off, d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  else
    DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, (k_qs_entries<true, false>), qs_grid(ctx, n), QS_BLOCK, a, perm, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
              (const uint64_t *) d_rec_off, d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  uint32_t st = 0;
  DX_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st & 2u)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode_onepass: an encoded segment differs in size from what the size kernel computed");
  return DX_OK;
}

// A batch of short entries with long ones among them (qs_verdict.mixed): the short ones as above (the lanes skip the long ones), the
// long ones -- a batch of their own under the arrays k_qs_sub_batch made, with the tokens and histograms dx_qv_hist left for THAT batch --
// through the wave-per-entry kernels; between sizes and records the long entries' sizes go to their places among all, one scan lays the
// records out, and the long ones' places (behind their framing bytes, written on the way) come back to their own batch.
static bool onepass_tokens_ok(const dx_ctx *ctx, const dx_qv_batch *b);
static int onepass_mixed(dx_ctx *ctx, const dx_qv_batch *b, const qs_verdict *qv, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                         uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap, uint64_t *total)
{ const uint64_t n = b->n, nl = qv->sub.n;
  const dx_qv_batch *b2 = &qv->sub;
  int e;
  uint8_t *scr;
  const size_t a4 = (n * 4 + 255) & ~(size_t) 255, s4 = (nl * 4 + 255) & ~(size_t) 255, s20 = (nl * 20 + 255) & ~(size_t) 255;
  if ((e = dx_scratch(ctx, a4 + s4 + s20 + (nl + 1) * 8 + 512, (void **) &scr))) return e;
  uint32_t *d_size = (uint32_t *) scr, *size2 = (uint32_t *) (scr + a4), *seg2 = (uint32_t *) (scr + a4 + s4);
  uint64_t *rec2 = (uint64_t *) (scr + a4 + s4 + s20);
  dx_sx_drop_external(ctx);
  ctx->sx.valid = 0;                                     // (no group index from this route)
  const qv_args a  = make_args(b,  ctx->delChar, ctx->subChar, ctx->lossy);
  const qv_args a2 = make_args(b2, ctx->delChar, ctx->subChar, ctx->lossy);
  const bool toks = onepass_tokens_ok(ctx, b2);          // the long entries' tokens (k_qv_hist over the batch of their own) ...
  const bool by_hist = toks && ctx->tk.eh_valid && !dx_test_on("sizes_from_tokens");
  const bool odd  = toks && ctx->tk.unusable > 0;        // ... some of which may be unusable: those entries from the text
  const tok_src tg = { ctx->tk.del, ctx->tk.sub, ctx->tk.off, ctx->tk.info };
  uint32_t *d_tick = (uint32_t *) (ctx->d_u64 + 19), *d_tick2 = (uint32_t *) (ctx->d_u64 + 22);
  const unsigned lb = (unsigned) ((nl + QS_BLOCK - 1) / QS_BLOCK), lb1 = (unsigned) ((nl + 1 + QS_BLOCK - 1) / QS_BLOCK);
  DX_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
  // sizes: the short entries' by the lanes, the long ones' by their own kernels
  DX_LAUNCH(ctx, DX_K_QV_SIZES, (k_qs_entries<false, false>), qs_grid(ctx, n), QS_BLOCK, a, qv->deal, (const uint32_t *) ctx->d_tok, (const uint8_t *) NULL, d_hdr_off,
            (const uint64_t *) NULL, d_seg, d_size, (uint8_t *) NULL, ctx->d_status);
  DX_HIP(ctx, hipMemsetAsync(d_tick2, 0, 4, ctx->stream));
  if (by_hist)
    DX_LAUNCH(ctx, DX_K_QV_SIZES, k_qv_sizes_hist, dx_grid_waves(ctx, (nl + 7) / 8, 32), DX_BLOCK, a2, (const uint32_t *) ctx->d_tok, (const uint64_t *) NULL,
              seg2, size2, tg, (const uint32_t *) ctx->tk.eh, ctx->tk.subChar);
  else if (toks)
    DX_LAUNCH(ctx, DX_K_QV_SIZES, k_qv_sizes_fast, fast_grid(ctx, (nl + TICKET_BATCH - 1) / TICKET_BATCH), FAST_BLOCK, a2, (const uint32_t *) ctx->d_tok,
              (const uint64_t *) NULL, seg2, size2, d_tick2, tg);
  if (!toks || odd)
    { if (toks) DX_HIP(ctx, hipMemsetAsync(d_tick2, 0, 4, ctx->stream));
      DX_LAUNCH(ctx, DX_K_QV_SIZES, k_qv_sizes, dx_grid_waves(ctx, toks ? (ctx->tk.unusable < nl ? ctx->tk.unusable : nl) : nl, 4 * SIZES_WAVES), DX_BLOCK,
                a2, (const uint32_t *) ctx->d_tok, (const uint64_t *) NULL, seg2, size2, d_tick2,
                toks ? (const uint32_t *) ctx->tk.list : (const uint32_t *) NULL, toks ? (const unsigned long long *) ctx->tk.count : (const unsigned long long *) NULL,
                (uint64_t) 0, toks ? (const uint32_t *) ctx->tk.info : (const uint32_t *) NULL);
    }
  hipLaunchKernelGGL(k_qs_scatter_sizes, dim3(lb), dim3(QS_BLOCK), 0, ctx->stream, qv->list, nl, (const uint32_t *) seg2, (const uint32_t *) size2,
                     d_hdr_off, d_seg, d_size);
  DX_HIP(ctx, hipGetLastError());
  uint64_t tot = 0;
  if ((e = dx_scan_u32(ctx, d_size, n, d_rec_off, &tot))) return e;
  if (total) *total = tot;
  ctx->route.groups = 0; ctx->route.direct = 5; ctx->route.tokens = toks ? 1 : 0; ctx->route.region_bytes = 0;
  ctx->route.scratch_bytes = ctx->scratch_bytes; ctx->route.token_bytes = toks ? 4ull * ctx->tk.cap_tokens : 0;
  ctx->route.text_entries = n - nl + (toks ? ctx->tk.unusable : nl);
  if (tot > out_cap)
    return dx_fail(ctx, DX_E_SPACE, "dx_qv_encode_onepass: the record stream needs %llu bytes, d_out holds %llu",
                   (unsigned long long) tot, (unsigned long long) out_cap);
  // records: the long entries' places and framing bytes, the short entries by the lanes, the long ones by their own kernels
  hipLaunchKernelGGL(k_qs_gather_places, dim3(lb1), dim3(QS_BLOCK), 0, ctx->stream, qv->list, nl, (const uint64_t *) d_rec_off, d_hdr, d_hdr_off, d_out, rec2);
  DX_HIP(ctx, hipGetLastError());
  if (ctx->tok_wide)
    DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, (k_qs_entries<true, true>), qs_grid(ctx, n), QS_BLOCK, a, qv->deal, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
              (const uint64_t *) d_rec_off, d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  else
    DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, (k_qs_entries<true, false>), qs_grid(ctx, n), QS_BLOCK, a, qv->deal, (const uint32_t *) ctx->d_tok, d_hdr, d_hdr_off,
              (const uint64_t *) d_rec_off, d_seg, (uint32_t *) NULL, d_out, ctx->d_status);
  DX_HIP(ctx, hipMemsetAsync(d_tick, 0, 4, ctx->stream));
  if (toks)
    DX_LAUNCH(ctx, DX_K_QV_ENCODE, FAST_K, fast_grid(ctx, nl), FAST_BLOCK, a2, (const uint32_t *) ctx->d_tok, (const uint64_t *) NULL, ctx->d_status, d_tick, tg,
              ctx->pair_lo[0], ctx->pair_lo[1], (const uint8_t *) NULL, (const uint64_t *) rec2, (const uint32_t *) seg2, d_out, ~(uint64_t) 0,
              sub_sink{ NULL, NULL, NULL });
  if (!toks || odd)
    { if (toks) DX_HIP(ctx, hipMemsetAsync(d_tick, 0, 4, ctx->stream));
      DX_LAUNCH(ctx, DX_K_QV_ENCODE_TEXT, k_qv_encode, dx_grid_waves(ctx, toks ? (ctx->tk.unusable < nl ? ctx->tk.unusable : nl) : nl, 4 * ENC_WAVES), DX_BLOCK,
                a2, (const uint32_t *) ctx->d_tok, (const uint8_t *) NULL, (const uint64_t *) NULL, (const uint64_t *) rec2, (const uint32_t *) seg2, d_out,
                ctx->d_status, d_tick, toks ? (const uint32_t *) ctx->tk.list : (const uint32_t *) NULL,
                toks ? (const unsigned long long *) ctx->tk.count : (const unsigned long long *) NULL, (uint64_t) 0,
                toks ? (const uint32_t *) ctx->tk.info : (const uint32_t *) NULL, ~(uint64_t) 0, sub_sink{ NULL, NULL, NULL });
    }
  uint32_t st = 0;
  DX_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st & DX_ST_INDEX)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode_onepass: an entry's offset and length reach beyond text_bytes");
  if (st & 2u)
    return dx_fail(ctx, DX_E_MISMATCH, "dx_qv_encode_onepass: an encoded segment differs in size from what the size kernel computed");
  return DX_OK;
}

