// dx_internal.hpp -- shared host-side definitions of libdexgpu (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "dexgpu.h"
#include "dx_env.h"

#define DX_WAVE          64
#define DX_BLOCK         256                 // 4 waves per workgroup
#define DX_WAVES_PER_BLK (DX_BLOCK / DX_WAVE)
#define DX_TOK_WORDS     (6 * 256)           // device token table: 6 schemes x 256 entries
#define DX_DEC_BITS      11                  // primary decode table: indexed by the next 11 bits
#define DX_DEC_SIZE      (1 << DX_DEC_BITS)
#define DX_LONG_MAX      256                 // codes longer than DX_DEC_BITS, per scheme

struct dx_pending { hipEvent_t a, b; int kernel; };

struct dx_ctx
{ int          device;
  hipStream_t  own, stream;
  hipStream_t  side;             // second stream: compaction beside the next group's encode (dx_qv_encode_onepass)
  hipEvent_t   ev[19];           // ordering between the two streams (no timing): 8 + 8 + 1, + 2 of the hybrid route
  int          num_cu;
  char         err[512];

  // profiling
  bool                    profiling;
  std::vector<dx_pending> pend;
  double                  ms[DX_K_COUNT];
  uint64_t                launches[DX_K_COUNT];

  // QV coder state (dx_qv_set_coding)
  uint32_t *d_tok;             // DX_TOK_WORDS packed tokens (see dx_qv.hip)
  uint16_t *d_dec;             // 6 x DX_DEC_SIZE primary decode entries (len << 8 | symbol; 0: long code)
  uint32_t *d_long;            // 6 x (1 + DX_LONG_MAX): count, then prefix16 << 16 | len << 8 | symbol
  int       sym_type[4];       // scheme types of del/ins/mrg/sub (2: escape code present)
  int       coding_set;
  int       lossy;
  int       delChar, subChar;
  uint32_t  bps[4];            // upper bound of encoded bits per symbol of del/ins/mrg/sub (dx_qv_encode_onepass)
  int       tok_wide;          // a symbol token of more than 16 bits in the tables (k_qs_entries: one code at a time then)
  struct                       // batches of short entries (dx_qv_short.hpp): the verdict on the last batch looked at, and every entry's place
    { uint8_t  *perm;          // among the 256 of its round when they are taken by length (k_qs_survey; NULL while none was wanted)
      uint64_t  cap, n, text_bytes;
      const void *off, *len;
      int       valid, brief, ordered;
      // a batch of short entries with long ones among them: the long ones' indices (k_qs_survey), and their offsets and lengths side by
      // side -- the batch of their own as which the wave-per-entry kernels take them
      int       mixed;
      uint64_t  nl;            // how many
      uint32_t *list;          // cap of them
      uint64_t *off2;
      uint32_t *len2;
      uint32_t  cut;           // entries longer than this are the long ones (chosen per batch: the longest short entry's lane must be done when the lanes are)
      uint32_t *order, *rmax;  // the rounds, longest entries first, and each round's longest (cap / 256 + 1 of them)
      unsigned long long *aux; // 256 words of counters for the survey (length histogram, round buckets)
    } qs;
  uint32_t  pair_lo[2];        // ins, mrg: lowest coded byte value when the coded values span <= 64 (pair tables), else ~0
  int       onepass_min_groups;// dx_qv_encode_onepass: fewest groups whose scratch regions have fitted the device so far
  uint64_t  scratch_budget;    // dx_set_scratch_budget (0: by hipMemGetInfo)
  dx_onepass_info route;       // what the last dx_qv_encode_onepass did (dx_qv_onepass_info)

  // run-length tokens the histogram pass leaves for the encoder (dx_qv.hip: "token hand-over")
  struct
  { uint16_t *del, *sub;       // token slots of the deletion / substitution stream, tok_off[r] .. tok_off[r+1]
    uint64_t *off;             // n + 1 slot offsets, in tokens
    uint32_t *info;            // n x 4: tokens of del | bit 31 unusable, of sub | bit 31, open run at the end of del, of sub
    uint32_t *eh;              // n x 384 words: every entry's own histograms as 768 counters of 16 bits (k_qv_hist<true>; see hist_wave_own)
    size_t    cap_eh;          // entries eh holds
    int       eh_valid;        // eh belongs to the tokens (the FAST instance made them)
    uint32_t  share8;          // the slot share k_tok_rooms chose for the batch (256ths of an entry's symbols; from the sampled run density)
    unsigned long long *count; // device copy of `unusable`, followed in the same allocation by ...
    uint32_t *list;            // ... the indices of those entries (any order)
    size_t    cap_tokens, cap_entries;        // what the buffers above hold
    // the batch and scan state they were made for (the encoder uses them only for exactly this batch)
    const void *text, *boff, *blen;
    uint64_t    n, text_bytes;
    uint32_t    pad;
    int         delChar, subChar;
    uint64_t    unusable;      // entries of the batch whose tokens cannot be used (encoded by the generic kernel)
    int         valid;
  } tk;

  // group index of the plain lines, left by dx_qv_encode_onepass for dx_qv_decode when dx_qv_subindex is on
  struct
  { int       want, valid;
    int       external;          // idx / off are the caller's (dx_qv_use_index): never freed here, dropped before the context makes its own
    int       walk;              // ... and are the device walk's (dx_qv_use_dindex): a plain line's share holds a word per 64 symbols
    uint32_t  sync_kinds;        //     (k_qv_decode_sync) of the kinds in sync_kinds, `nosync` lines without
    uint64_t  nosync;
    uint32_t *idx;               // one byte per group of 16 symbols, 4 * sub_words(len) words per entry
    uint64_t *off;               // n + 1: where each entry's words start
    uint32_t *room;              // n: scratch of the offsets' scan
    uint32_t *none;              // device counter: run-coded lines the encoders left without an index (RUN_NONE)
    size_t    cap_idx, cap_entries;
    const void *out, *seg;       // the record stream and segment index it belongs to
    uint64_t    n;
  } sx;

  // scratch owned by the context
  uint32_t *d_status;          // device error flags (bit 0: symbol count mismatch)
  uint64_t *d_u64;             // small device scalars (prescan keys, totals, ...)
  void     *d_scratch;         // grow-only scratch (scan partials, histograms, sizes)
  size_t    scratch_bytes;
  uint32_t  compact_units;     // records per ticket of k_qv_compact for the batch being encoded (onepass_impl sets it)
  uint64_t  scratch_gen;       // counts its re-allocations: a new block may come back at the OLD address with other contents,
                               //   so "did it move" must never be asked of the pointer
  uint64_t *d_scan;            // grow-only tile sums of dx_scan_u32 (its callers hold d_scratch)
  size_t    scan_words;
  uint8_t  *h_stage[2];        // pinned staging of dx_d2h_stream (made at its first call)
  uint8_t  *h_up;              // pinned staging of a large dx_h2d (made at its first such call)
  uint8_t  *h_down;            // pinned staging of dx_d2h_stream with several sink threads (dx_set_sink_threads)
  int       sink_threads;
  // dx_qv_scan: what the last scan of this context launched (a guess for the next, checked on the device) and pinned host words --
  // [0, 2048): what a scan brings back (histograms, totals, the device's scan state), [2048, 2048 + 768): the code tables on their way up
  struct { int valid; uint32_t inst; } scan;
  uint64_t *h_pin;
  std::vector<uint16_t> h_dec; // the decode tables dx_qv_set_coding built (Read_Scheme's look-up, two levels), uploaded by the first
  std::vector<uint32_t> h_lng; //   dx_qv_decode that wants them (dec_stale): an encode step never reads them
  int       dec_stale;
  void     *d_hscr;            // grow-only scratch of dx_qv_hist (its own: d_scratch may still be read by the compaction
  size_t    hscr_bytes;        //   of an encode that has begun, dx_qv_encode_onepass_begin)
  // an encode that has begun and not ended (dx_qv_encode_onepass_begin / _end)
  struct { int pending, direct, rc; uint64_t total; uint64_t *d_total; uint64_t out_cap; uint32_t *sx_idx; } op;
};

int  dx_fail(dx_ctx *ctx, int code, const char *fmt, ...);
// Every device allocation of the library goes through dx_hip_malloc: with DEXGPU_POISON=<byte value> in the environment
// the new block is filled with that byte, so that a kernel reading memory nothing has written yet does so reproducibly
// (a debugging aid; fresh device memory usually reads as zeros and hides such reads).
hipError_t dx_hip_malloc(void **p, size_t bytes);
#define hipMalloc(p, n) dx_hip_malloc((void **) (p), (n))
int  dx_scratch(dx_ctx *ctx, size_t bytes, void **p);
int  dx_dec_tables(dx_ctx *ctx);             // the decode tables on the device (uploads them when dx_qv_set_coding has left them stale)
void dx_sx_drop_external(dx_ctx *ctx);      // forget a caller's group index (dx_qv_use_index) before the context's own buffers are touched
uint64_t dx_budget(const dx_ctx *ctx);
int  dx_after_pending(dx_ctx *ctx);
void dx_prof_begin_on(dx_ctx *ctx, int kernel, hipStream_t stream);
void dx_prof_end_on(dx_ctx *ctx, hipStream_t stream);
static inline void dx_prof_begin(dx_ctx *ctx, int kernel) { dx_prof_begin_on(ctx, kernel, ctx->stream); }
static inline void dx_prof_end(dx_ctx *ctx)               { dx_prof_end_on(ctx, ctx->stream); }
int  dx_grid_waves(dx_ctx *ctx, uint64_t n_units, int waves_per_cu);

#define DX_HIP(ctx, call)                                                                   \
  do { hipError_t e_ = (call);                                                              \
       if (e_ != hipSuccess)                                                                \
         return dx_fail(ctx, DX_E_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                                \
     } while (0)

// Launch `kern<<<grid, block, 0, ctx->stream>>>(...)` bracketed by profiling events.
#define DX_LAUNCH_ON(ctx, strm, id, kern, grid, block, ...)                                 \
  do { dx_prof_begin_on(ctx, id, strm);                                                     \
       hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, strm, __VA_ARGS__);             \
       dx_prof_end_on(ctx, strm);                                                           \
       DX_HIP(ctx, hipGetLastError());                                                      \
     } while (0)
#define DX_LAUNCH(ctx, id, kern, grid, block, ...) DX_LAUNCH_ON(ctx, (ctx)->stream, id, kern, grid, block, __VA_ARGS__)
