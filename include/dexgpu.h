/*
 * dexgpu.h -- C-ABI of libdexgpu: MI355X (gfx950) kernels for DEXTRACTOR's three codecs.
 *
 * The reference (thegenemyers/DEXTRACTOR) has no plugin/FFI interface; its boundary is the C API
 * of QV.h:48-97 + DB.h:255-267 (per-entry, FILE*-streaming, static state) under six CLI tools.
 * A GPU cannot be fed one entry at a time through FILE*, so this library exposes the SAME
 * operations batch-wise over device-resident buffers.  Each entry point names the reference
 * function(s) it replaces.  Plain pointers and sizes only; no HIP or torch types.
 *
 * Conventions
 *   - `d_` pointers are DEVICE pointers (from dx_malloc, hipMalloc, or a torch tensor's
 *     data_ptr()); everything else is host memory.
 *   - every function returns DX_OK (0) or a negative DX_E_* code; dx_last_error() gives a
 *     message.  Nothing here ever calls exit() (the reference's batch error model, DB.h:45-47,
 *     is reproduced by the CLI front-ends, not the library).
 *   - one dx_ctx per GPU and per host thread; all work of a context is issued on one HIP
 *     stream (dx_set_stream lets a caller supply its own, e.g. torch's current stream).
 *   - multi-byte integers in all produced file images are little-endian, as the reference
 *     writes them on x86-64 hosts (endian key 0x55aa / 0x33cc).
 */
#ifndef DEXGPU_H
#define DEXGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DX_OK              0
#define DX_E_ARG         (-1)   /* bad argument */
#define DX_E_HIP         (-2)   /* HIP runtime error (message has hipGetErrorString) */
#define DX_E_FORMAT      (-3)   /* malformed input image */
#define DX_E_DEGENERATE  (-4)   /* a stream that needs a scheme has an empty histogram (reference: UB, QV.c:201) */
#define DX_E_UNSUPPORTED (-5)   /* a final code longer than 16 bits (the reference's own decoder cannot read it) */
#define DX_E_NOMEM       (-6)
#define DX_E_MISMATCH    (-7)   /* device-side consistency check failed (e.g. symbol count != expected) */
#define DX_E_SPACE       (-8)   /* output buffer too small */
#define DX_E_IO          (-9)   /* a caller's sink refused data (dx_d2h_stream); a file could not be read (dx_h2d_fd) */
#define DX_E_AGAIN       (-10)  /* not this way: an entry point that takes a file descriptor wants the image in memory after all
                                   (a small or malformed file, one that does not fit the device): call its in-memory twin */

typedef struct dx_ctx dx_ctx;

/* ------------------------------------------------------------------------------------------
 *  context, memory, profiling
 * ------------------------------------------------------------------------------------------ */
int         dx_device_count(void);
int         dx_open(int device, dx_ctx **ctx);
void        dx_close(dx_ctx *ctx);
const char *dx_last_error(const dx_ctx *ctx);      /* ctx may be NULL: last error of dx_open */
int         dx_set_stream(dx_ctx *ctx, void *hip_stream);   /* issue on this hipStream_t; NULL = the default (null) stream,
                                                               e.g. torch's current stream when it is the default one */
int         dx_reset_stream(dx_ctx *ctx);                   /* back to the context's own non-blocking stream */
int         dx_sync(dx_ctx *ctx);

int dx_malloc(dx_ctx *ctx, size_t bytes, void **d_ptr);
int dx_free  (dx_ctx *ctx, void *d_ptr);
int dx_h2d   (dx_ctx *ctx, void *d_dst, const void *src, size_t bytes);   /* synchronous */
int dx_d2h   (dx_ctx *ctx, void *dst, const void *d_src, size_t bytes);   /* synchronous */
/* bytes [foff, foff + bytes) of the file behind fd to device memory, read (pread) by several threads straight into pinned buffers
 * that leave as they fill: no mapping of the file in the caller's address space (a large dx_h2d from memory goes through the same
 * threads).  DX_E_IO: the file could not be read.                                                                       */
int dx_h2d_fd(dx_ctx *ctx, void *d_dst, int fd, uint64_t foff, size_t bytes);
/* Device memory to a consumer on the host, in order, in chunks (32 MiB) through two pinned staging buffers the
 * context owns: sink(user, data, len, at) is called on a helper thread for the bytes [at, at + len) while the
 * next chunk is in flight; it may modify data[0 .. len) and returns 0, or nonzero to stop (DX_E_IO comes back).
 * For outputs headed to a file: a pwrite per chunk costs no page fault in the caller's address space, where
 * dx_d2h into fresh memory pays one per 4 KiB (0.25 s per GiB, single-threaded).                             */
typedef int (*dx_sink_fn)(void *user, uint8_t *data, size_t len, size_t at);
int dx_d2h_stream(dx_ctx *ctx, const void *d_src, size_t bytes, dx_sink_fn sink, void *user);
/* A sink that is positional -- it takes chunk [at, at + len) whatever came before, from whichever thread (a pwrite per chunk) --
 * may be fed by several threads: from 128 MiB on dx_d2h_stream (and every file driver that delivers through a sink) then keeps up
 * to eight 16 MiB copies in flight and calls the sink from `threads` helper threads, chunks in any order (one thread writing into
 * a file's pages is slower than the link brings them).  Default 1: in order, one thread.  Returns the value in force before.     */
int dx_set_sink_threads(dx_ctx *ctx, int threads);
int dx_memset(dx_ctx *ctx, void *d_dst, int value, size_t bytes);

/* Per-kernel device time, measured with HIP events on the context's stream around every launch
 * while profiling is enabled.  Kernel ids: */
enum { DX_K_PACK2_ENC = 0, DX_K_PACK2_DEC, DX_K_QV_PRESCAN, DX_K_QV_HIST, DX_K_QV_SIZES, DX_K_SCAN,
       DX_K_QV_ENCODE, DX_K_QV_DECODE, DX_K_SYNTH, DX_K_INDEX, DX_K_QV_COMPACT,
       DX_K_QV_ENCODE_TEXT,                       /* the encoder that reads the text (two-pass path; entries without usable tokens) */
       DX_K_QV_DEC_SUB, DX_K_QV_DEC_RUNS, DX_K_QV_DEC_PLAIN, DX_K_QV_DEC_TAGS,   /* DX_K_QV_DECODE: the generic lane-per-line decoder */
       DX_K_QV_WALK,                              /* the record walk of a bare stream (dx_qv_walk_device) */
       DX_K_COUNT };
int         dx_profile(dx_ctx *ctx, int enable);                  /* enabling resets the counters */
int         dx_profile_get(dx_ctx *ctx, int kernel, double *ms_total, uint64_t *launches);  /* syncs */
const char *dx_kernel_name(int kernel);

/* ------------------------------------------------------------------------------------------
 *  2-bit packers: dexta/undexta, dexar/undexar
 * ------------------------------------------------------------------------------------------ */
enum { DX_ALPHA_BASES = 0,    /* Number_Read  DB.c:393: c/C->1 g/G->2 t/T->3, everything else 0   */
       DX_ALPHA_ARROW = 1,    /* Number_Arrow DB.c:418: '1'->0 '2'->1 '3','G'->2, everything else 3 */
       DX_ALPHA_NUMBERS = 2 };/* already numbers 0..3 (what Compress_Read DB.c:319 takes): the low two bits */
enum { DX_LETTERS_LOWER = 0,  /* Lower_Read  DB.c:367 acgt */
       DX_LETTERS_UPPER = 1,  /* Upper_Read  DB.c:375 ACGT */
       DX_LETTERS_ARROW = 2,  /* Letter_Arrow DB.c:383 1234 */
       DX_LETTERS_NUMBERS = 3 };/* the numbers 0..3 themselves (what Uncompress_Read DB.c:342 leaves) */

/* Replaces, for n reads at once: the line-gathering loop dexta.c:161-183 (newlines are skipped on
 * the device), Number_Read/Number_Arrow (DB.c:393-441), Compress_Read (DB.c:319-338) and the
 * record writes dexta.c:187-204 / dexar.c:193-210.
 *   read i's sequence text = d_text[d_off[i] .. d_off[i]+d_tlen[i]) : any number of lines, '\n'
 *   bytes are dropped, all other bytes are symbols; it must hold exactly d_nsym[i] symbols.
 *   Output record i is written at d_out + d_out_off[i]: first the framing bytes
 *   d_hdr[d_hdr_off[i] .. d_hdr_off[i+1]) (may be NULL: no framing), then (nsym+3)>>2 packed
 *   bytes, first symbol in the top two bits, missing symbols 0.
 * Returns DX_E_MISMATCH if some read's symbol count differs from d_nsym[i].                     */
int dx_pack2_encode(dx_ctx *ctx, int alphabet,
                    const uint8_t *d_text, const uint64_t *d_off, const uint32_t *d_tlen,
                    const uint32_t *d_nsym, uint64_t n,
                    const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                    uint8_t *d_out, const uint64_t *d_out_off);

/* Replaces Uncompress_Read (DB.c:342-363) + Lower_Read/Upper_Read/Letter_Arrow (DB.c:367-389)
 * + the line wrapping of undexta.c:263-270 / undexar.c:221-228.
 *   read i's packed bytes start at d_in + d_in_off[i]; its text (d_nsym[i] letters, a '\n'
 *   after every `width` letters and after the last partial line; nothing for an empty read) is
 *   written at d_out + d_out_off[i].  width >= 1.                                               */
int dx_pack2_decode(dx_ctx *ctx, int letters,
                    const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_nsym, uint64_t n,
                    uint32_t width, uint8_t *d_out, const uint64_t *d_out_off);

/* Host helpers for the record framing of all three formats (dexta.c:187-198, dexar.c:159-163 +
 * 193-204, dexqv.c:128-139): well-delta chain, int32 beg, end, then int32 qv (kind 0) or four
 * uint16 cnr (kind 1, values already converted with dx_snr_to_cnr).  `field[i]` points at the
 * 4 int32 {well,beg,end,qv} (kind 0) or {well,beg,end,-} followed in `cnr` by 4 uint16 (kind 1).
 * blob must hold dx_frame_bound(...) bytes; off gets n+1 entries.  *lwell carries the previous
 * well across calls (starts at 0, dexta.c:137).                                                 */
size_t   dx_frame_bound(const int32_t *hdr4, uint64_t n, int32_t lwell, int kind);
int      dx_frame_headers(const int32_t *hdr4, const uint16_t *cnr4, uint64_t n, int kind,
                          int32_t *lwell, uint8_t *blob, uint64_t *off);
uint16_t dx_snr_to_cnr(float snr);                 /* dexar.c:159-163 */

/* ------------------------------------------------------------------------------------------
 *  text front end (host): index a file image so the kernels can read it in place
 * ------------------------------------------------------------------------------------------ */
/* why an image was rejected (*errcode), with the 1-based line number in *errline */
enum { DX_IDX_NO_NEWLINE = 1,   /* last line does not end with a newline   (QV.c:779) */
       DX_IDX_NO_HEADER  = 2,   /* header line missing                     (QV.c:954, dexta.c:113) */
       DX_IDX_BAD_HEADER = 3,   /* header line incorrectly formatted       (QV.c:958-968, dexta.c:146-155) */
       DX_IDX_INCOMPLETE = 4,   /* incomplete last entry of .quiva file    (QV.c:789) */
       DX_IDX_RAGGED     = 5,   /* lines for an entry are not the same length (QV.c:793) */
       DX_IDX_TOO_LONG   = 6,   /* fasta/arrow line longer than 99998 chars (dexta.c:165-172) */
       DX_IDX_EMPTY      = 7 }; /* empty input (the reference reads an uninitialised buffer, dexta.c:108) */

/* .quiva: replaces the Read_Lines loops and header validation of QVcoding_Scan (QV.c:751-798,
 * 948-978).  Fills (up to cap entries of) off[i] = offset of entry i's first data line, len[i] =
 * symbols per line, hdr4[4i..] = well, beg, end, qv; *count = entries found (call with cap 0 to
 * count); *prefix_len = bytes of the first header before its first '/' after position 0.      */
int dx_index_quiva(const uint8_t *text, size_t n, uint64_t cap,
                   uint64_t *off, uint32_t *len, int32_t *hdr4,
                   uint64_t *count, size_t *prefix_len, uint64_t *errline, int *errcode);

/* .fasta (arrow = 0) / .arrow (arrow = 1): replaces dexta.c:104-183 / dexar.c:103-188.  off[i] =
 * offset of read i's first sequence line, tlen[i] = its text bytes up to the next header or the
 * end (newlines included), nsym[i] = symbols (tlen - lines), hdr4 = well, beg, end, qv (qv = 0
 * when "RQ=" is missing), cnr4 (arrow) = the four SN values converted by dx_snr_to_cnr.        */
int dx_index_seq(int arrow, const uint8_t *text, size_t n, uint64_t cap,
                 uint64_t *off, uint32_t *tlen, uint32_t *nsym, int32_t *hdr4, uint16_t *cnr4,
                 uint64_t *count, size_t *prefix_len, uint64_t *errline, int *errcode);

/* GPU text front end for a .quiva image already in device memory: newline scan, structure checks
 * and entry index on the device; only the header lines come back to the host, where sscanf parses
 * them as QV.c:964 does.  Same results as dx_index_quiva.  *d_off / *d_len are device arrays
 * (release with dx_free), *hdr4 a host array (free()).  On DX_E_FORMAT *errline / *errcode name an
 * offending line; callers that need the reference's exact first message re-run dx_index_quiva
 * on the host copy (the error path is not performance relevant).                               */
int dx_index_quiva_device(dx_ctx *ctx, const uint8_t *d_text, uint64_t nbytes,
                          uint64_t **d_off, uint32_t **d_len, uint64_t *count,
                          int32_t **hdr4, size_t *prefix_len, uint64_t *errline, int *errcode);
int dx_parse_quiva_headers(const uint8_t *blob, const uint64_t *pos, uint64_t n, int32_t *hdr4,
                           size_t *prefix_len, uint64_t *bad_entry);

/* The same for a .fasta (arrow = 0) / .arrow (arrow = 1) image: records are delimited on the device by
 * their header lines ('>' first); results as dx_index_seq.  On DX_E_FORMAT re-run dx_index_seq on the
 * host copy for the reference's exact first message.                                             */
int dx_index_seq_device(dx_ctx *ctx, int arrow, const uint8_t *d_text, uint64_t nbytes,
                        uint64_t **d_off, uint32_t **d_tlen, uint32_t **d_nsym, uint64_t *count,
                        int32_t **hdr4, uint16_t **cnr4, size_t *prefix_len,
                        uint64_t *errline, int *errcode);
int dx_parse_seq_headers(int arrow, const uint8_t *blob, const uint64_t *pos, uint64_t n,
                         int32_t *hdr4, uint16_t *cnr4, size_t *prefix_len, uint64_t *bad_entry);

/* ------------------------------------------------------------------------------------------
 *  5-stream QV coder: dexqv/undexqv (QV.c)
 * ------------------------------------------------------------------------------------------ */
enum { DX_DEL = 0, DX_INS = 1, DX_MRG = 2, DX_SUB = 3, DX_DRUN = 4, DX_SRUN = 5 };

/* A batch of .quiva entries resident on the device.  Entry i's five streams (deletion QV,
 * deletion tag, insertion QV, merge QV, substitution QV -- the line order of QV.c:973-991) are
 * d_len[i] bytes each; stream k starts at d_text + d_off[i] + k*(d_len[i] + line_pad).  For a
 * .quiva file image line_pad = 1 (the '\n').                                                    */
typedef struct
  { const uint8_t  *d_text;
    const uint64_t *d_off;
    const uint32_t *d_len;
    uint64_t        n;
    uint64_t        text_bytes;   /* readable bytes at d_text (the whole image).  With it the kernels
                                     may load a line's last, partial 16-byte chunk in one piece;
                                     0 = unknown: nothing beyond a line's end is ever read        */
    uint32_t        line_pad;
  } dx_qv_batch;

/* The order-dependent scan state of QVcoding_Scan (QV.c:993-1017): run characters (-1 none) and
 * the GLOBAL entry index from which each run-length histogram starts counting.                  */
typedef struct
  { int32_t delChar, subChar;
    int64_t del_first, sub_first;
  } dx_qv_params;

/* The lossy rounding of Compress_Next_QVentry (QV.c:1355-1372: insertion QVs >> 1 << 1, merge QVs >> 2 << 2) applied IN PLACE
 * to the batch's text on the device: the text a lossy .dexqv decodes back to (what a lossy round trip is compared with). */
int dx_qv_lossy_text(dx_ctx *ctx, const dx_qv_batch *b);

/* Finds delChar (deletion QV under the first 'n'/'N' tag, QV.c:993-1002) and the provisional
 * subChar (argmax of the substitution histogram of the entries up to the one where the running
 * symbol count first reaches 100000, ties to the smallest value, QV.c:1006-1015), on the device.
 * entry0 = global index of the batch's first entry.  Fields of *p that are already set (>= 0)
 * are kept; initialise *p to {-1,-1,-1,-1}.  The subChar search needs the file's first entries,
 * so it only runs for the batch with entry0 == 0 (a shard holding the first 100000 symbols).   */
int dx_qv_prescan(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, dx_qv_params *p);

/* Histogram_Seqs x4 + Histogram_Runs x2 (QV.c:702-724, 988-1017) over the batch: ADDS the
 * batch's counts into hist (host, 6x256, order DX_DEL..DX_SRUN) and its symbol count into
 * *totChar.  Run histograms are raw counts: the reference's start value of 1 per bin
 * (QV.c:934-935) is added once by dx_qv_build.  Shards of one file: call per shard (any GPU),
 * add the arrays on the host.                                                                   */
int dx_qv_hist(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, const dx_qv_params *p,
               uint64_t hist[6][256], uint64_t *totChar);

/* QVcoding_Scan (QV.c:922-1023) over one batch in one call: dx_qv_prescan followed by dx_qv_hist -- the same *p, hist and
 * *totChar as the two calls leave -- with ONE wait on the device instead of three.  The scan state stays on the device
 * between the kernels; the two things the host has to decide before it knows that state (which of its histogram kernels
 * fits the batch's run density, whether the token buffers it holds are large enough) it takes from the context's last
 * scan, and the device checks them: a context's first batch, or a batch unlike the last one, simply goes through the
 * two calls.  Nothing of one batch's RESULTS is carried to the next.  DEXGPU_TEST=no_scan_guess: always the two calls.   */
int dx_qv_scan(dx_ctx *ctx, const dx_qv_batch *b, uint64_t entry0, dx_qv_params *p,
               uint64_t hist[6][256], uint64_t *totChar);

typedef struct
  { int32_t  type;              /* 0 normal, 2 truncated with escape code (QV.c:76-81) */
    uint32_t bits[256];
    int32_t  lens[256];
  } dx_scheme;

typedef struct
  { dx_scheme s[6];             /* DX_DEL..DX_SRUN; run schemes valid iff the run char is >= 0 */
    int32_t   delChar, subChar; /* final values (subChar may have been dropped, QV.c:1044-1045) */
  } dx_qv_coding;

/* Create_QVcoding (QV.c:1029-1169) incl. Huffman/Reheap/Build_Table (QV.c:91-220): host only. */
int dx_qv_build(const uint64_t hist[6][256], uint64_t totChar, const dx_qv_params *p, int lossy,
                dx_qv_coding *out);

/* Write_QVcoding (QV.c:1173-1210) / Read_QVcoding (QV.c:1214-1320) on memory images.  The
 * image starts at the 0x33cc key (dexqv's own 0x55aa key, dexqv.c:105, is not included).       */
int dx_qv_write_coding(const dx_qv_coding *c, const char *prefix, size_t plen,
                       uint8_t *buf, size_t cap, size_t *written);
int dx_qv_read_coding(const uint8_t *buf, size_t n, dx_qv_coding *c, int *flip,
                      char *prefix, size_t pcap, size_t *consumed);

/* Uploads the code tables for the kernels below (and remembers `lossy`, QV.c:1406-1415). */
int dx_qv_set_coding(dx_ctx *ctx, const dx_qv_coding *c, int lossy);

/* Record sizes: for each entry the bytes Compress_Next_QVentry (QV.c:1381-1426) would write
 * (bit totals, pad rule QV.c:436-442 / 499-505, Pack_Tag length): d_seg[5*i+k] = bytes of entry
 * i's del / tag / ins / mrg / sub segment -- the index the format itself does not store and a
 * parallel decoder needs.  Adding the framing bytes d_hdr_off[i+1]-d_hdr_off[i] (NULL: none) and
 * an exclusive scan in file order gives d_rec_off[0..n] (device, n+1 entries); *total =
 * d_rec_off[n].                                                                                */
int dx_qv_sizes(dx_ctx *ctx, const dx_qv_batch *b, const uint64_t *d_hdr_off,
                uint32_t *d_seg, uint64_t *d_rec_off, uint64_t *total);

/* Compress_Next_QVentry for the whole batch: record i = framing bytes, then the del words, tag
 * bytes, ins words, mrg words, sub words (Encode / Encode_Run / Pack_Tag+Number_Read+
 * Compress_Read, QV.c:386-506, 810-819, 1393-1423) at d_out + d_rec_off[i].  d_seg and
 * d_rec_off are the outputs of dx_qv_sizes for the same batch and coding (the deletion QVs and
 * their tags are written in one sweep, which needs the deletion segment's size up front); the
 * kernel re-derives every segment size and returns DX_E_MISMATCH if one disagrees.              */
int dx_qv_encode(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                 const uint64_t *d_rec_off, const uint32_t *d_seg, uint8_t *d_out);

/* Compress_Next_QVentry x n (QV.c:1381-1426) in one call: the five segment sizes of every entry (d_seg, n x 5: the same index
 * dx_qv_sizes produces), a scan that turns them into d_rec_off (n + 1), and every record -- framing bytes and segments -- written
 * where it belongs in d_out.  *total receives the stream's size; DX_E_SPACE if it exceeds out_cap (nothing useful is in d_out
 * then).  The bytes are those of dx_qv_sizes + dx_qv_encode.  Which kernels run depends on what this context's dx_qv_hist (or
 * dx_qv_scan) of the same batch has left: tokens and every entry's own histograms (the usual case: sizes by a dot product,
 * k_qv_encode_fast), tokens alone (sizes from tokens and plain lines), nothing (sizes and records from the text); a batch of
 * short entries goes a lane an entry, one with long entries among the short ones is dealt between the two kinds of kernel
 * (dx_qv_onepass_info tells).  No scratch beyond O(n) bookkeeping: rounds 2-3's slots and their compaction are gone (round 6). */
int dx_qv_encode_onepass(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                         uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap, uint64_t *total);
/* The same in two halves (kept from the rounds in which a compaction ran behind the encoder: a caller written for them works
 * unchanged): _begin does the encode and keeps the answer, _end hands back the stream's size and the verdict (DX_E_SPACE, ...).
 * Between the two, the NEXT batch's dx_qv_prescan, dx_qv_hist, dx_qv_build and dx_qv_set_coding may run; dx_qv_sizes,
 * dx_qv_encode, dx_qv_decode and a second dx_qv_encode_onepass[_begin] are refused (DX_E_ARG) until _end has been called.  A
 * dx_qv_set_coding in between leaves the group index (dx_qv_subindex) of the ended encode unarmed.  One encode at a time per
 * context.                                                                                                              */
int dx_qv_encode_onepass_begin(dx_ctx *ctx, const dx_qv_batch *b, const uint8_t *d_hdr, const uint64_t *d_hdr_off,
                               uint32_t *d_seg, uint64_t *d_rec_off, uint8_t *d_out, uint64_t out_cap);
int dx_qv_encode_onepass_end(dx_ctx *ctx, uint64_t *total);

/* Which way the last dx_qv_encode_onepass / _begin of this context took, for logs and benchmarks: direct (below), tokens = 1 when
 * the histogram pass's tokens fed the encoder, scratch_bytes = the context's scratch allocation after the call, token_bytes =
 * the token slots, text_entries = entries the text-reading kernels took (no tokens, or tokens that could not be used: a byte
 * >= 128 in a run-coded line, more tokens than the slot or the list holds; the short entries of routes 3 and 5).  groups,
 * region_bytes and avail_bytes are of the scratch-slot route, which is gone: always 0 (kept for the layout).            */
typedef struct
  { int32_t  groups, direct, tokens, reserved;        /* direct: 1 sizes by k_qv_sizes_fast (tokens and plain lines read again), 2 sizes from the
                                                         entries' own histograms (k_qv_sizes_hist): the product route, 3 a batch of short entries: a
                                                         lane per entry (k_qs_entries: sizes, then records), 4 no tokens of this batch: sizes and records
                                                         from the text, 5 short entries by the lanes and the long ones among them as a batch of their
                                                         own through the wave-per-entry kernels */
    uint64_t region_bytes, scratch_bytes, avail_bytes, token_bytes, text_entries;
    uint64_t chain_waits[3];                          /* always 0 (the routes that reported here are gone; kept for the layout) */
  } dx_onepass_info;
int dx_qv_onepass_info(const dx_ctx *ctx, dx_onepass_info *out);

/* An upper bound on one scratch request of the context (0, the default: none; requests of up to 64 MB always pass).  Rounds
 * 2-3's encoder sized its scratch regions by it; since round 6 no route needs scratch beyond O(n) bookkeeping, and the call
 * only bounds that.  The environment variable DEXGPU_SCRATCH_BUDGET (bytes) overrides it.                              */
int dx_set_scratch_budget(dx_ctx *ctx, uint64_t bytes);

/* Device memory free / in all on the context's GPU right now (hipMemGetInfo). */
int dx_mem_info(dx_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* Gives device memory the context keeps between calls back to the device (it is allocated again when next needed):
 * DX_TRIM_SCRATCH the scratch regions of dx_qv_encode_onepass and dx_qv_hist, DX_TRIM_TOKENS the token slots the
 * histogram pass leaves for the encoder (the next encode without a fresh dx_qv_hist then reads the text),
 * DX_TRIM_INDEX the group index of dx_qv_subindex (the next decode then takes the lane-per-line kernels).       */
#define DX_TRIM_SCRATCH 1
#define DX_TRIM_TOKENS  2
#define DX_TRIM_INDEX   4
#define DX_TRIM_ALL     7
int dx_trim(dx_ctx *ctx, int what);

/* Host helper sizing d_out for dx_qv_encode_onepass: an upper bound of the bytes the batch's n entries
 * encode to (framing bytes not included), from the batch's own raw histograms (what dx_qv_hist added
 * for it) and the coding in force -- every symbol priced at its code (QV.c:427-434), run tokens at
 * theirs (QV.c:475-497), partial / pad words (QV.c:436-442) and tag bytes (QV.c:810-819) per entry.
 * Within a few percent of the real size once the file is much longer than the 100000 symbols after which
 * the run histograms start (the runs before are priced at the dearest run token).                    */
uint64_t dx_qv_out_bound(const uint64_t hist[6][256], uint64_t n, const dx_qv_coding *c, int lossy);

/* Uncompress_Next_QVentry (QV.c:1428-1481: Decode, Decode_Run, Unpack_Tag) for n records whose
 * segment starts are known: record i starts at d_in + d_rec_off[i] with d_hdr_off[i+1] -
 * d_hdr_off[i] framing bytes (NULL: none) followed by segments of d_seg[5*i+k] bytes -- exactly
 * the arrays dx_qv_sizes produces, or dx_qv_walk for a bare file.  The five decoded lines
 * (d_len[i] symbols each, every line followed by '\n') are written at d_out + d_out_off[i].
 * flags: DX_DECODE_UPPER applies undexqv's -U (undexqv.c:198-204); DX_DECODE_FLIP reads the code
 * words byte-swapped (a file written on a host of the other endianness: GETFLIP, QV.c:553-568).
 * Uses the tables of dx_qv_set_coding.                                                          */
#define DX_DECODE_UPPER 1
#define DX_DECODE_FLIP  2
int dx_qv_decode(dx_ctx *ctx, const uint8_t *d_in, const uint64_t *d_rec_off, const uint64_t *d_hdr_off,
                 const uint32_t *d_seg, const uint32_t *d_len, uint64_t n, int flags,
                 uint8_t *d_out, const uint64_t *d_out_off);

/* Host walk of a bare .dexqv image (undexqv.c:101-208 and the bit-level structure of QV.c:510-691):
 * the format stores no lengths, so the start of every segment is only known after walking the one
 * before it, code by code.  Fills the index dx_qv_decode needs (arrays are malloc'd; release with
 * dx_qv_index_free).  Sequential by nature of the format.                                        */
typedef struct
  { uint64_t      n;            /* records */
    uint64_t     *rec_off;      /* n+1: offset of each record in the image */
    uint64_t     *hdr_off;      /* n+1: running sum of the records' framing bytes */
    uint32_t     *seg;          /* n x 5 segment byte sizes */
    uint32_t     *len;          /* n: symbols per entry (end - beg, undexqv.c:186) */
    int32_t      *hdr4;         /* n x 4: well, beg, end, qv */
    dx_qv_coding  coding;
    char         *prefix;       /* header prefix (QV.c:1256-1265) */
    int           newv, flip;   /* 0x55aa-keyed file (int32 fields) / byte-swapped writer */
    /* dx_qv_walk_indexed(want_index != 0): the group index of dx_qv_subindex, made by the walk (it passes every code anyway) */
    uint32_t     *gidx;         /* gidx_words words: per entry the plain lines' group bytes, three header words, the run lines' group words */
    uint64_t     *gidx_off;     /* n+1: where each entry's words start */
    uint64_t      gidx_words;
    uint64_t      gidx_none;    /* run-coded lines left without an index (a group that does not fit its word) */
  } dx_qv_index;
int  dx_qv_walk(const uint8_t *img, size_t n, dx_qv_index *idx);
/* The same, and with want_index != 0 also the group index (layout: csrc/dx_layout.h) that lets dx_qv_decode take a
 * wavefront per line instead of a lane: hand it over with dx_qv_use_index.  The walk then takes one table look-up per
 * symbol instead of one per several.                                                                         */
int  dx_qv_walk_indexed(const uint8_t *img, size_t n, dx_qv_index *idx, int want_index);
/* A group index for the stream at d_in with the segment index d_seg (n entries), uploaded by the caller (d_gidx: the
 * words, d_gidx_off: n + 1 offsets; none: dx_qv_index.gidx_none).  dx_qv_decode of that stream -- all of it or a
 * contiguous part, same d_in and d_seg -- then decodes with it.  The memory stays the caller's and must outlive the
 * decodes; d_gidx == NULL takes the index back.  An index the context made itself (dx_qv_subindex) is dropped.   */
int  dx_qv_use_index(dx_ctx *ctx, const uint8_t *d_in, const uint32_t *d_seg, uint64_t n,
                     const uint32_t *d_gidx, const uint64_t *d_gidx_off, uint64_t none);
void dx_qv_index_free(dx_qv_index *idx);

/* The same walk on the device, for a record stream already there (undexqv.c:119-208, QV.c:1428-1481): d_img[0, n) is the
 * file image, `first` the offset of its first record (behind the 0x55aa key and the coding, which the caller has read on
 * the host: dx_qv_read_coding), cd that coding, newv / flip as dx_qv_index has them.  The stream is cut into pieces of
 * 32 KiB (more of a stream beyond 16 GB); a lane per piece finds a record start in its piece, checks it by walking, and walks on to the next piece's
 * start; only offsets on an unbroken chain of such walks from the first record are kept, so the index is the host
 * walk's (csrc/dx_qv_walk.hip).  The arrays are device memory of the library's (dx_qv_dindex_free): what dx_qv_decode
 * takes, and d_hdr4 (n x 4: well, beg, end, qv) for the header lines.  DX_E_MISMATCH: the walks did not chain up, a record
 * did not walk, 16-bit framing fields -- the caller then walks on the host (dx_qv_walk), which also names what is wrong
 * with a damaged file.  The run-coded lines' groups of the group index are noted on the way (dx_qv_use_dindex).               */
typedef struct
  { uint64_t  n;                /* records */
    uint64_t *d_rec_off;        /* n + 1 */
    uint64_t *d_hdr_off;        /* n + 1 */
    uint32_t *d_seg;            /* n x 5 */
    uint32_t *d_len;            /* n */
    int32_t  *d_hdr4;           /* n x 4 */
    uint64_t  pieces, piece_bytes;      /* how the stream was cut (for logs) */
    /* the run-coded lines' share of the group index (NULL: none made -- neither line run-coded, a byte-swapped stream):
       gidx_words words, gidx_off[n + 1] where each entry's begin, gidx_none run-coded lines without (dx_qv_use_dindex) */
    uint32_t *d_gidx;
    uint64_t *d_gidx_off;
    uint64_t  gidx_words, gidx_none;
    uint64_t  gidx_nosync;      /* plain lines without their words (where every 64th symbol is) */
    uint32_t  sync_kinds;       /* bit q: the plain lines of kind q (0 del, 1 ins, 2 mrg, 3 sub) have such words */
  } dx_qv_dindex;
int  dx_qv_walk_device(dx_ctx *ctx, const uint8_t *d_img, uint64_t n, uint64_t first, const dx_qv_coding *cd, int newv, int flip,
                       dx_qv_dindex *out);
void dx_qv_dindex_free(dx_ctx *ctx, dx_qv_dindex *x);
/* Hands the index a device walk has left to the decoder: a dx_qv_decode of the stream at d_in with x's arrays (the whole of
 * it or a contiguous part) then decodes every run-coded line -- and the tag line with the deletion line -- a wavefront at a
 * time (k_qv_decode_runs; QV.c:604-691, 823-847) instead of a lane; the plain lines have no groups in this index and stay with
 * the lane-per-line kernel.  x == NULL or x->d_gidx == NULL: takes it back.  x must outlive the decodes.                  */
int  dx_qv_use_dindex(dx_ctx *ctx, const uint8_t *d_in, const dx_qv_dindex *x);

/* Group index (on = 1): dx_qv_encode_onepass also leaves, in the context, where the codes are: one byte per group of
 * 16 symbols of each plain line (the group's code bits), one word per group of <= 8 (run, symbol) tokens of each
 * run-coded line (bits and positions covered) -- about 4.5 KB per 10 kb entry.  A dx_qv_decode of that record stream
 * (same d_in / d_seg, the whole batch or a contiguous part of it) in the same context then decodes each line with a
 * whole wavefront, 64 consecutive groups at a time, instead of a lane (k_qv_decode_sub, k_qv_decode_runs).  The .dexqv
 * bytes do not change; a stream decoded without the index (a bare file, another context) takes the lane-per-line
 * kernels, as do the lines of entries the fast encoder does not handle (runs of 127 and more, bytes >= 128).  Off by
 * default.                                                                                                       */
int  dx_qv_subindex(dx_ctx *ctx, int on);

/* ------------------------------------------------------------------------------------------
 *  whole-file drivers (host images in, host images out): what the six CLI tools call
 * ------------------------------------------------------------------------------------------ */
/* Each takes a complete input file image and returns the complete output image (malloc'd, free
 * with dx_file_free), byte-identical to the file the reference tool writes.  On DX_E_FORMAT from
 * the text front end, *errline / *errcode say where and why (DX_IDX_*).
 *   dx_file_pack2    dexta.c:104-205 (arrow = 0) / dexar.c:103-211 (arrow = 1)
 *   dx_file_unpack2  undexta.c:131-271 (mode DX_LETTERS_LOWER / _UPPER) / undexar.c:129-229 (_ARROW)
 *   dx_file_dexqv    dexqv.c:79-143                                                             */
int  dx_file_pack2  (dx_ctx *ctx, int arrow, const uint8_t *text, size_t n,
                     uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode);
/* The same for a text that arrives in pieces (a pipe: dexta -i, dexta.c:55-58; a file too large to hold): rd(user, buf, want)
 * returns the bytes it put into buf (fewer than want only at the end, 0 at the end, < 0: an error); `chunk` bytes are read at a time
 * (0: 256 MiB), whole records of them packed, the image's bytes handed to the sink in file order (at = their place in the
 * image).  The same bytes as dx_file_pack2 of the whole text; memory: about 1.3 x chunk whatever the text's size.           */
typedef long (*dx_read_fn)(void *user, void *buf, size_t want);
int  dx_file_pack2_stream(dx_ctx *ctx, int arrow, dx_read_fn rd, void *ruser, size_t chunk,
                          dx_sink_fn sink, void *suser, size_t *out_len, uint64_t *errline, int *errcode);
/* ... and the other way (undexta -i / undexar -i, undexta.c:175-271): the image in pieces of `chunk` bytes (0: 128 MiB), the text of
 * the whole records among them to the sink in file order; mode and width as for dx_file_unpack2.                              */
int  dx_file_unpack2_stream(dx_ctx *ctx, int mode, dx_read_fn rd, void *ruser, size_t chunk, uint32_t width,
                            dx_sink_fn sink, void *suser, size_t *out_len);
int  dx_file_unpack2(dx_ctx *ctx, int mode, const uint8_t *img, size_t n, uint32_t width,
                     uint8_t **out, size_t *out_len);
int  dx_file_dexqv  (dx_ctx *ctx, const uint8_t *text, size_t n, int lossy,
                     uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode);
/* dx_file_unpack2 with the text delivered through a sink, in order (dx_d2h_stream), header lines in place */
int  dx_file_unpack2_to(dx_ctx *ctx, int mode, const uint8_t *img, size_t n, uint32_t width,
                        dx_sink_fn sink, void *user, size_t *out_len);
/* dx_file_dexqv with the .dexqv image delivered through a sink instead (the file's head, then the record stream
 * in chunks: dx_d2h_stream), for a caller that writes it straight to a file.  The sink sees nothing unless the
 * whole input was valid and encoded.                                                                       */
/* (dx_file_dexqv / _to: a .quiva image that does not fit the device beside its tokens, scratch and output -- or is larger than
 *  DEXGPU_TEXT_BUDGET bytes when that is set -- is worked through in SLICES of whole entries, like the reference streams a
 *  file of any size (dexqv.c:112-143): the scan pass over every slice (histograms added up on the host, the scan state
 *  carried along), the tables, then slice by slice again -- upload, tokens, encode, records out.  Same bytes.)          */
/* ... and with the .quiva read from a file descriptor (dx_h2d_fd: no image of it in the caller's memory at all): n bytes from
 * offset 0.  DX_E_AGAIN -- nothing has been passed to the sink -- when the file has to be handed over in memory after all (below
 * 1 MiB, larger than the device takes at once, or malformed: the in-memory driver has the reference's words for that).   */
int  dx_file_dexqv_fd_to(dx_ctx *ctx, int fd, size_t n, int lossy, dx_sink_fn sink, void *user,
                         size_t *out_len, uint64_t *errline, int *errcode);
int  dx_file_dexqv_to(dx_ctx *ctx, const uint8_t *text, size_t n, int lossy, dx_sink_fn sink, void *user,
                      size_t *out_len, uint64_t *errline, int *errcode);
/* dexqv of ONE file on several GPUs (one context each; contiguous entry ranges, one host thread per
 * context, scan state and 12 KB histograms merged on the host, outputs concatenated): identical
 * bytes to dx_file_dexqv.  nctx == 1 is dx_file_dexqv.                                          */
int  dx_file_dexqv_sharded(dx_ctx **ctxs, int nctx, const uint8_t *text, size_t n, int lossy,
                           uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode);
/* dexta / dexar of ONE file on several GPUs: contiguous read ranges balanced by text bytes, one host
 * thread per context, no exchange at all; identical bytes to dx_file_pack2.        */
int  dx_file_pack2_sharded(dx_ctx **ctxs, int nctx, int arrow, const uint8_t *text, size_t n,
                           uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode);
int  dx_file_undexqv(dx_ctx *ctx, const uint8_t *img, size_t n, int upper,
                     uint8_t **out, size_t *out_len);                       /* undexqv.c:101-208 */
/* The same in two steps, for a caller that wants the text somewhere else than in a malloc'd image (the CLI
 * streams it into the output file).  The plan -- the host walk over the record stream (dx_qv_walk) and the
 * header lines, undexqv.c:182 -- needs no GPU (it can run while a context is still being opened) and tells
 * the output's size; the run decodes on the GPU and delivers the text through the sink, in order
 * (dx_d2h_stream).  img must stay valid until the run is over.                                           */
typedef struct dx_undexqv_plan dx_undexqv_plan;
int  dx_file_undexqv_plan(const uint8_t *img, size_t n, dx_undexqv_plan **plan, size_t *out_len);
/* The plan with the GPU at hand: a large 0x55aa-keyed image is uploaded and its records are walked on the device
 * (dx_qv_walk_device), anything else -- and anything the device walk turns down -- is planned on the host as above.  Such a
 * plan holds device memory of ctx until it is freed, and runs on ctx only.  dx_file_undexqv plans this way.
 * A text that does not fit the device beside the image (or DEXGPU_TEXT_BUDGET bytes) comes out in slices of whole entries. */
int  dx_file_undexqv_plan_on(dx_ctx *ctx, const uint8_t *img, size_t n, dx_undexqv_plan **plan, size_t *out_len);
int  dx_file_undexqv_run (dx_ctx *ctx, const dx_undexqv_plan *plan, int upper, dx_sink_fn sink, void *user);
/* The record index the plan holds, as host arrays of the caller's (release with dx_qv_index_free; no group index): where every
 * record and segment of the image stands -- what the walk found, wherever it ran.                                       */
int  dx_file_undexqv_plan_index(const dx_undexqv_plan *plan, dx_qv_index *idx);
void dx_file_undexqv_plan_free(dx_undexqv_plan *plan);
void dx_file_free(void *p);

/* ------------------------------------------------------------------------------------------
 *  in-memory entry API: QVcoding_Scan1 / Compress_Next_QVentry1 (QV.c:866-920, 1343-1379) as a batch
 * ------------------------------------------------------------------------------------------ */
/* dex2DB.c:511-643 feeds entries one at a time through the *1 functions and writes the compressed
 * bytes of each to a .qvs file, remembering its offset (DAZZ_READ.coff).  Here the entries are
 * gathered first (dx_entries_add has the *1 functions' argument list), then scanned and compressed
 * together on the GPU: *records = the bare record stream (no framing bytes), coff[i] = offset of
 * entry i in it (n+1 values), *coding = the tables to write with dx_qv_write_coding.  records / coff
 * are malloc'd (dx_file_free).  The streams are copied; the caller's buffers are not modified
 * (the reference mutates them in place).                                                        */
typedef struct dx_entries dx_entries;
dx_entries *dx_entries_new(void);
void        dx_entries_free(dx_entries *e);
int         dx_entries_add(dx_entries *e, int rlen, const char *del, const char *tag, const char *ins,
                           const char *mrg, const char *sub);
int         dx_entries_compress(dx_ctx *ctx, const dx_entries *e, int lossy, dx_qv_coding *coding,
                                uint8_t **records, size_t *nbytes, uint64_t **coff);

/* ------------------------------------------------------------------------------------------
 *  seeded synthetic corpora on the device (benchmark/test plumbing; mirrors dextractor_amd/synth.py)
 * ------------------------------------------------------------------------------------------ */
/* Writes entries [entry0, entry0+n) of the .quiva corpus `seed`: for each entry the header line
 * "@<movie>/WWWWWWWW/BBBBBBB_EEEEEEE RQ=0.QQQ\n" (fixed width, from d_hdr4 = well,beg,end,qv)
 * ending right before d_off[i], then the five lines.  d_lut = 5 x 4096 symbol tables (del, tag,
 * ins, mrg, sub); del_run = value under which the tag is 'N' (-1 never).                        */
int dx_synth_quiva(dx_ctx *ctx, uint32_t seed, uint64_t entry0, uint64_t n,
                   const uint64_t *d_off, const uint32_t *d_len, const int32_t *d_hdr4,
                   const uint8_t *d_lut, int del_run, const char *movie, uint8_t *d_text);

#ifdef __cplusplus
}
#endif
#endif
