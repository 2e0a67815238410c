/*
 * dx_files.c -- whole-file drivers: the host side of the six tools, above the kernel C-ABI.
 *
 * Each function takes a complete input file image in host memory and returns the complete output
 * image (malloc'd; release with dx_file_free), byte-identical to what the reference tool writes:
 *
 *     dx_file_pack2    dexta.c:104-205 / dexar.c:103-211
 *     dx_file_unpack2  undexta.c:131-271 / undexar.c:129-229
 *     dx_file_dexqv    dexqv.c:79-143 (QVcoding_Scan, Create_QVcoding, Write_QVcoding,
 *                      Compress_Next_QVentry per entry)
 *     dx_file_undexqv  undexqv.c:101-208
 *
 * The host does what is O(records) or pure text parsing (indexing lines, sscanf of the header
 * fields, sprintf of decoded headers, Huffman table construction); every per-symbol loop runs on
 * the GPU through the kernels of libdexgpu.  Plain C: only the public C-ABI is used.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dexgpu.h"
#include "dx_env.h"

void dx_file_free(void *p) { free(p); }

/* DEXGPU_TIMING=1: where a file driver spends its time (stderr; the tools print their own marks beside these) */
#include <time.h>
#include <unistd.h>
static void fmark(const char *what)
{ static double t0 = -1.0;
  struct timespec ts;
  double now;
  if (getenv("DEXGPU_TIMING") == NULL) return;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  now = ts.tv_sec + 1e-9 * ts.tv_nsec;
  if (t0 < 0) t0 = now;
  fprintf(stderr, "[dx_file %8.1f ms] %s\n", (now - t0) * 1e3, what);
}

#define TRY(x) do { rc = (x); if (rc != DX_OK) goto done; } while (0)

#define DX_GPU_INDEX_MIN (1u << 20)      /* .quiva images from 1 MiB on are indexed on the GPU */

typedef struct { void *p[24]; int n; dx_ctx *ctx; } dpool;

static int dalloc(dpool *pool, size_t bytes, void **out)
{ int rc;
  if (pool->n >= (int) (sizeof(pool->p) / sizeof(pool->p[0])) - 2) return DX_E_NOMEM;   /* pool slots exhausted */
  rc = dx_malloc(pool->ctx, bytes + 64, out);
  if (rc == DX_OK) pool->p[pool->n++] = *out;
  return rc;
}

static int dupload(dpool *pool, const void *src, size_t bytes, void **out)
{ int rc = dalloc(pool, bytes, out);
  if (rc == DX_OK && bytes) rc = dx_h2d(pool->ctx, *out, src, bytes);
  return rc;
}

/* The encoder of the file drivers is dx_qv_encode_onepass (no size pass; the same bytes).  DEXGPU_TEST=twopass
 * selects dx_qv_sizes + dx_qv_encode instead (the two-pass API, kept as a cross-check). */
static int two_pass(void)
{ return dx_test_on("twopass"); }

static void dfree_all(dpool *pool)
{ int i;
  for (i = 0; i < pool->n; i++)
    dx_free(pool->ctx, pool->p[i]);
  pool->n = 0;
}

/* ==========================================================================================
 *  dexta / dexar
 * ========================================================================================== */

/* A header line with the end of the file right behind it is an error for the reference unless it is the file's only line
   (dx_index_seq has the story): such a text goes to the host index, which knows; the device front end takes the empty read. */
static int ends_with_a_header(const uint8_t *text, size_t n)
{ size_t at;
  if (n < 2 || text[n - 1] != '\n') return 0;
  for (at = n - 1; at > 0 && text[at - 1] != '\n'; at--) ;
  return at > 0 && text[at] == '>';
}

/* how much text the device packs at once: DEXGPU_TEXT_BUDGET (bytes) when set, else all of it (0) unless the text, its packed
   image and the index do not fit what is free */
static size_t pack2_cap(dx_ctx *ctx, size_t n)
{ const char *e = getenv("DEXGPU_TEXT_BUDGET");
  uint64_t fr = 0, all = 0;
  if (e != NULL && *e)
    { const unsigned long long v = strtoull(e, NULL, 10);
      return v && v < n ? (size_t) (v < 65536u ? 65536u : v) : 0;
    }
  if (dx_mem_info(ctx, &fr, &all) != DX_OK || fr == 0) return 0;
  return 1.35 * (double) n > 0.9 * (double) fr ? (size_t) (0.6 * (double) fr) : 0;
}

/* One piece of a .fasta / .arrow text -- whole records, the first of them the file's first (`first`: the image then begins with the
   key and the name prefix, dexta.c:124-129) or a later one; *well: the last record's well before and after (the record framing
   codes differences, dexta.c:187-193). */
static int pack2_piece(dx_ctx *ctx, int arrow, const uint8_t *text, size_t n, int first, int32_t *well,
                       uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ dpool     pool = { {0}, 0, ctx };
  uint64_t  cnt = 0, i, *off = NULL, *hoff = NULL, *ooff = NULL;
  uint32_t *tlen = NULL, *nsym = NULL;
  int32_t  *hdr4 = NULL, lwell = *well;
  uint16_t *cnr4 = NULL;
  uint8_t  *blob = NULL, *img = NULL;
  size_t    plen = 0, at, total;
  void     *d_text = NULL, *d_off = NULL, *d_tlen = NULL, *d_nsym = NULL, *d_hdr, *d_hoff, *d_out, *d_ooff;
  size_t    sliced;
  int       rc;

  if (ctx == NULL || out == NULL || out_len == NULL) return DX_E_ARG;
  *out = NULL; *out_len = 0;

  /* A text that does not fit the device beside its packed image (or DEXGPU_TEXT_BUDGET): indexed on the host, then slices of
     whole reads -- upload, pack, the slice's records into the image (the reference reads record after record,
     dexta.c:104-205). */
  sliced = pack2_cap(ctx, n);
  /* index: on the GPU for large images (newline scan, record extents there; only header lines come
     back), on the host for small ones and for anything the GPU front end rejects (exact message) */
  if (n > 0 && !sliced) TRY(dupload(&pool, text, n, &d_text));
  if (!sliced && n >= DX_GPU_INDEX_MIN && !dx_test_on("host_index") && !ends_with_a_header(text, n))
    { uint64_t *go = NULL; uint32_t *gt = NULL, *gs = NULL;
      rc = dx_index_seq_device(ctx, arrow, d_text, n, &go, &gt, &gs, &cnt, &hdr4, &cnr4, &plen, errline, errcode);
      if (rc == DX_OK)
        { d_off = go; d_tlen = gt; d_nsym = gs;
          pool.p[pool.n++] = go; pool.p[pool.n++] = gt; pool.p[pool.n++] = gs;
          nsym = malloc((cnt + 1) * sizeof(*nsym));
          if (!nsym) { rc = DX_E_NOMEM; goto done; }
          TRY(dx_d2h(ctx, nsym, d_nsym, cnt * 4));
        }
      else if (rc != DX_E_FORMAT)
        goto done;
      rc = DX_OK;
    }
  if (d_off == NULL)
    { TRY(dx_index_seq(arrow, text, n, 0, NULL, NULL, NULL, NULL, NULL, &cnt, &plen, errline, errcode));
      off  = malloc((cnt + 1) * sizeof(*off));
      tlen = malloc((cnt + 1) * sizeof(*tlen));
      nsym = malloc((cnt + 1) * sizeof(*nsym));
      hdr4 = malloc((cnt + 1) * 4 * sizeof(*hdr4));
      cnr4 = malloc((cnt + 1) * 4 * sizeof(*cnr4));
      if (!off || !tlen || !nsym || !hdr4 || !cnr4) { rc = DX_E_NOMEM; goto done; }
      TRY(dx_index_seq(arrow, text, n, cnt, off, tlen, nsym, hdr4, cnr4, &cnt, &plen, errline, errcode));
      if (!sliced)
        { TRY(dupload(&pool, off,  cnt * 8, &d_off));
          TRY(dupload(&pool, tlen, cnt * 4, &d_tlen));
          TRY(dupload(&pool, nsym, cnt * 4, &d_nsym));
        }
    }
  hoff = malloc((cnt + 1) * sizeof(*hoff));
  ooff = malloc((cnt + 1) * sizeof(*ooff));
  if (!hoff || !ooff) { rc = DX_E_NOMEM; goto done; }

  blob = malloc(dx_frame_bound(hdr4, cnt, 0, arrow) + 16);
  if (!blob) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_frame_headers(hdr4, cnr4, cnt, arrow, &lwell, blob, hoff));

  if (!first) plen = 0;
  at = first ? 2 + 4 + plen : 0;                       /* key, prefix length, prefix: dexta.c:124-129 */
  for (i = 0; i < cnt; i++)
    { ooff[i] = at;
      at += (size_t) (hoff[i+1] - hoff[i]) + (((size_t) nsym[i] + 3) >> 2);
    }
  total = at;

  img = malloc(total + 16);
  if (!img) { rc = DX_E_NOMEM; goto done; }
  if (first)
    { uint16_t key = 0x55aa;
      int32_t  pl  = (int32_t) plen;
      memcpy(img, &key, 2);
      memcpy(img + 2, &pl, 4);
      memcpy(img + 6, text, plen);
    }

  if (cnt > 0 && sliced)
    { uint64_t *rel = NULL, i0, i1, most = 0;
      size_t    tmax = 0, omax = 0;
#define READ_END(i) ((size_t) off[i] + tlen[i])
      for (i0 = 0; i0 < cnt; i0 = i1)
        { i1 = i0 + 1;
          while (i1 < cnt && READ_END(i1) - (size_t) off[i0] <= sliced) i1++;
          if (READ_END(i1 - 1) - (size_t) off[i0] > tmax) tmax = READ_END(i1 - 1) - (size_t) off[i0];
          if ((i1 < cnt ? ooff[i1] : total) - ooff[i0] > omax) omax = (size_t) ((i1 < cnt ? ooff[i1] : total) - ooff[i0]);
          if (i1 - i0 > most) most = i1 - i0;
        }
      rel = malloc(3 * (most + 1) * sizeof(*rel));
      if (rel == NULL) { rc = DX_E_NOMEM; goto done; }
      rc = dalloc(&pool, tmax, &d_text);
      if (rc == DX_OK) rc = dalloc(&pool, (most + 1) * 8, &d_off);
      if (rc == DX_OK) rc = dupload(&pool, tlen, cnt * 4, &d_tlen);
      if (rc == DX_OK) rc = dupload(&pool, nsym, cnt * 4, &d_nsym);
      if (rc == DX_OK) rc = dupload(&pool, blob, (size_t) hoff[cnt], &d_hdr);
      if (rc == DX_OK) rc = dalloc(&pool, (most + 1) * 8, &d_hoff);
      if (rc == DX_OK) rc = dalloc(&pool, (most + 1) * 8, &d_ooff);
      if (rc == DX_OK) rc = dalloc(&pool, omax, &d_out);
      for (i0 = 0; i0 < cnt && rc == DX_OK; i0 = i1)
        { const size_t b0 = (size_t) off[i0];
          const uint64_t o0 = ooff[i0];
          i1 = i0 + 1;
          while (i1 < cnt && READ_END(i1) - b0 <= sliced) i1++;
          for (i = i0; i <= i1; i++)
            { if (i < i1) { rel[i - i0] = off[i] - b0; rel[2 * (most + 1) + (i - i0)] = ooff[i] - o0; }
              rel[(most + 1) + (i - i0)] = hoff[i] - hoff[i0];
            }
          rc = dx_h2d(ctx, d_text, text + b0, READ_END(i1 - 1) - b0);
          if (rc == DX_OK) rc = dx_h2d(ctx, d_off, rel, (i1 - i0) * 8);
          if (rc == DX_OK) rc = dx_h2d(ctx, d_hoff, rel + (most + 1), (i1 - i0 + 1) * 8);
          if (rc == DX_OK) rc = dx_h2d(ctx, d_ooff, rel + 2 * (most + 1), (i1 - i0) * 8);
          if (rc == DX_OK)
            rc = dx_pack2_encode(ctx, arrow ? DX_ALPHA_ARROW : DX_ALPHA_BASES, d_text, d_off, (const uint32_t *) d_tlen + i0, (const uint32_t *) d_nsym + i0,
                                 i1 - i0, (const uint8_t *) d_hdr + hoff[i0], d_hoff, d_out, d_ooff);
          if (rc == DX_OK) rc = dx_d2h(ctx, img + o0, d_out, (size_t) ((i1 < cnt ? ooff[i1] : total) - o0));
        }
#undef READ_END
      free(rel);
      if (rc != DX_OK) goto done;
    }
  else if (cnt > 0)
    { TRY(dupload(&pool, blob, (size_t) hoff[cnt], &d_hdr));
      TRY(dupload(&pool, hoff, (cnt + 1) * 8, &d_hoff));
      TRY(dupload(&pool, ooff, cnt * 8, &d_ooff));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_pack2_encode(ctx, arrow ? DX_ALPHA_ARROW : DX_ALPHA_BASES, d_text, d_off, d_tlen, d_nsym, cnt,
                          d_hdr, d_hoff, d_out, d_ooff));
      TRY(dx_d2h(ctx, img + ooff[0], (uint8_t *) d_out + ooff[0], total - (size_t) ooff[0]));
    }
  *out = img; *out_len = total; img = NULL;
  *well = lwell;
  rc = DX_OK;

done:
  dfree_all(&pool);
  free(off); free(hoff); free(ooff); free(tlen); free(nsym); free(hdr4); free(cnr4); free(blob); free(img);
  return rc;
}

int dx_file_pack2(dx_ctx *ctx, int arrow, const uint8_t *text, size_t n,
                  uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ int32_t well = 0;
  if (ctx == NULL || out == NULL || out_len == NULL) return DX_E_ARG;
  return pack2_piece(ctx, arrow, text, n, 1, &well, out, out_len, errline, errcode);
}

/* dexta / dexar of a text that arrives in pieces -- a pipe, or a file too large to hold: the reference reads record after record
   (dexta.c:104-205, dexar.c:103-211) and never holds more than one.  Here: `chunk` bytes at a time from rd(); a piece is cut in
   front of the buffer's last header but one (so that what stays behind begins with a header and holds another: the last piece,
   which the end of the input makes, then tells a lone last header -- the reference's "too long" -- from a file of one header), the
   piece's records packed on the device like a whole file's, its bytes handed to the sink in file order, the rest moved to the
   buffer's front.  Memory: the buffer (chunk + a record or two) and a piece's image. */
/* line ends in n bytes, eight at a time (a text of gigabytes byte by byte is seconds) */
static uint64_t count_newlines(const uint8_t *p, size_t n)
{ uint64_t c = 0;
  size_t   i = 0;
  for (; i + 8 <= n; i += 8)
    { uint64_t w, x, t;
      memcpy(&w, p + i, 8);
      x = w ^ 0x0a0a0a0a0a0a0a0aull;
      t = (((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & 0x8080808080808080ull;     /* 0x80 in every byte that is no line end */
      c += 8u - (uint64_t) __builtin_popcountll(t);
    }
  for (; i < n; i++) c += p[i] == '\n';
  return c;
}

/* Reading ahead: a helper thread takes the next block from the caller's read function while the last one is on the device -- a pipe
   hands over 2 GB/s at best, and a chunk's packing is no faster than that: one after the other they add up (dexta -i of 4 GB: 3.1 s),
   side by side the slower one counts.  Two blocks of RA_BLOCK bytes; ra_read() gives their bytes out in order. */
#define RA_BLOCK ((size_t) 32 << 20)
typedef struct
  { dx_read_fn      rd;
    void           *user;
    uint8_t        *blk[2];
    size_t          len[2], pos;
    int             full[2], cur, eof, err, stop, threaded;
    pthread_mutex_t mx;
    pthread_cond_t  cv;
    pthread_t       th;
  } readahead;

static void *ra_main(void *arg)
{ readahead *r = arg;
  int slot = 0;
  for (;;)
    { size_t n = 0;
      pthread_mutex_lock(&r->mx);
      while (r->full[slot] && !r->stop) pthread_cond_wait(&r->cv, &r->mx);
      if (r->stop) { pthread_mutex_unlock(&r->mx); break; }
      pthread_mutex_unlock(&r->mx);
      while (n < RA_BLOCK)
        { const long k = r->rd(r->user, r->blk[slot] + n, RA_BLOCK - n);
          if (k < 0) { r->err = 1; break; }
          if (k == 0) break;
          n += (size_t) k;
        }
      pthread_mutex_lock(&r->mx);
      r->len[slot] = n; r->full[slot] = 1;
      if (n < RA_BLOCK) r->eof = 1;
      pthread_cond_broadcast(&r->cv);
      pthread_mutex_unlock(&r->mx);
      if (n < RA_BLOCK) break;
      slot ^= 1;
    }
  return NULL;
}

static void ra_begin(readahead *r, dx_read_fn rd, void *user)
{ memset(r, 0, sizeof(*r));
  r->rd = rd; r->user = user;
  r->blk[0] = malloc(RA_BLOCK); r->blk[1] = malloc(RA_BLOCK);
  if (r->blk[0] != NULL && r->blk[1] != NULL && !dx_test_on("no_readahead"))
    { pthread_mutex_init(&r->mx, NULL);
      pthread_cond_init(&r->cv, NULL);
      r->threaded = pthread_create(&r->th, NULL, ra_main, r) == 0;
    }
}

static long ra_read(void *arg, void *buf, size_t want)
{ readahead *r = arg;
  size_t got = 0;
  if (!r->threaded) return r->rd(r->user, buf, want);
  while (got < want)
    { int have;
      pthread_mutex_lock(&r->mx);
      while (!r->full[r->cur] && !r->eof && !r->err) pthread_cond_wait(&r->cv, &r->mx);
      have = r->full[r->cur];
      pthread_mutex_unlock(&r->mx);
      if (r->err) return -1;
      if (!have) break;                                /* the input's end, and nothing left in this block (blocks come in turn) */
      { const size_t k = r->len[r->cur] - r->pos < want - got ? r->len[r->cur] - r->pos : want - got;
        memcpy((uint8_t *) buf + got, r->blk[r->cur] + r->pos, k);
        got += k; r->pos += k;
      }
      if (r->pos == r->len[r->cur])
        { const int last = r->len[r->cur] < RA_BLOCK;
          pthread_mutex_lock(&r->mx);
          r->full[r->cur] = 0;
          pthread_cond_broadcast(&r->cv);
          pthread_mutex_unlock(&r->mx);
          r->cur ^= 1; r->pos = 0;
          if (last) break;
        }
    }
  return (long) got;
}

static void ra_end(readahead *r)
{ if (r->threaded)
    { pthread_mutex_lock(&r->mx);
      r->stop = 1;
      pthread_cond_broadcast(&r->cv);
      pthread_mutex_unlock(&r->mx);
      pthread_join(r->th, NULL);
      pthread_cond_destroy(&r->cv);
      pthread_mutex_destroy(&r->mx);
    }
  free(r->blk[0]); free(r->blk[1]);
}

int dx_file_pack2_stream(dx_ctx *ctx, int arrow, dx_read_fn rd_, void *ruser_, size_t chunk,
                         dx_sink_fn sink, void *suser, size_t *out_len, uint64_t *errline, int *errcode)
{ uint8_t *buf = NULL;
  readahead ra;
  dx_read_fn rd = ra_read;
  void      *ruser = &ra;
  size_t   cap, have = 0, total = 0;
  uint64_t lines = 0;
  int32_t  well = 0;
  int      eof = 0, first = 1, rc = DX_OK;

  if (ctx == NULL || rd_ == NULL || sink == NULL) return DX_E_ARG;
  if (chunk == 0) chunk = (size_t) dx_test_num("stream_chunk", (long long) 256 << 20);
  if (chunk < 4096) chunk = 4096;
  cap = chunk + 65536;
  buf = malloc(cap);
  if (buf == NULL) return DX_E_NOMEM;
  ra_begin(&ra, rd_, ruser_);
  if (out_len) *out_len = 0;
  for (;;)
    { size_t cut, k, heads = 0;
      while (!eof && have < chunk)
        { const long got = rd(ruser, buf + have, chunk - have);
          if (got < 0) { rc = DX_E_IO; goto done; }
          if (got == 0) eof = 1;
          have += (size_t) got;
        }
      cut = have;
      if (!eof)                                        /* the last header line but one that is not the buffer's first line -- nor stands */
        { int good = 0;                                /* behind another header line: a piece that ENDS in a header reads like a file that does */
          for (k = have; k > 1 && !good; k--)
            if (buf[k - 1] == '>' && buf[k - 2] == '\n')
              { size_t q = k - 2;                       /* the line in front of this header begins at q */
                while (q > 0 && buf[q - 1] != '\n') q--;
                heads++;
                if (heads >= 2 && buf[q] != '>') { cut = k - 1; good = 1; }
              }
          if (!good)                                   /* a record (or two) larger than the chunk: more of it */
            { uint8_t *nb;
              chunk += chunk;
              nb = realloc(buf, chunk + 65536);
              if (nb == NULL) { rc = DX_E_NOMEM; goto done; }
              buf = nb; cap = chunk + 65536;
              continue;
            }
        }
      if (cut > 0 || first)
        { uint8_t *img = NULL;
          size_t   il = 0;
          uint64_t el = 0;
          rc = pack2_piece(ctx, arrow, buf, cut, first, &well, &img, &il, &el, errcode);
          if (rc != DX_OK)
            { if (errline) *errline = el ? lines + el : 0;
              goto done;
            }
          if (il > 0 && sink(suser, img, il, total)) { free(img); rc = DX_E_IO; goto done; }
          free(img);
          total += il;
          lines += count_newlines(buf, cut);
          first = 0;
        }
      memmove(buf, buf + cut, have - cut);
      have -= cut;
      if (eof && have == 0) break;
    }
  if (out_len) *out_len = total;
done:
  ra_end(&ra);
  free(buf);
  return rc;
}

/* ==========================================================================================
 *  dexta / dexar of one file on several GPUs: reads are independent, so contiguous read ranges
 *  (balanced by text bytes) go to one host thread per context; nothing is exchanged -- the only
 *  cross-record datum, the previous well of a range's first read, is known from the host index.
 * ========================================================================================== */
typedef struct
  { dx_ctx         *ctx;
    int             arrow, rc;
    const uint8_t  *text;
    const uint64_t *off, *ooff, *hoff;       /* per read: text offset, output offset, header-blob offset */
    const uint32_t *tlen, *nsym;
    const uint8_t  *blob;
    uint8_t        *img;
    uint64_t        lo, hi;                   /* reads [lo, hi) */
  } p2_job;

static void *p2_main(void *arg)
{ p2_job   *j = (p2_job *) arg;
  dpool     pool = { {0}, 0, j->ctx };
  uint64_t  m = j->hi - j->lo, i, *roff = NULL, *rhoff = NULL, *rooff = NULL;
  void     *d_text, *d_off, *d_tlen, *d_nsym, *d_hdr, *d_hoff, *d_out, *d_ooff;
  int       rc = DX_OK;
  if (m == 0) { j->rc = DX_OK; return NULL; }
  { const uint64_t t0 = j->off[j->lo], t1 = j->off[j->hi - 1] + j->tlen[j->hi - 1];
    const uint64_t h0 = j->hoff[j->lo], o0 = j->ooff[j->lo];
    const uint64_t obytes = j->ooff[j->hi] - o0;
    roff  = malloc(m * sizeof(*roff));
    rhoff = malloc((m + 1) * sizeof(*rhoff));
    rooff = malloc(m * sizeof(*rooff));
    if (!roff || !rhoff || !rooff) { rc = DX_E_NOMEM; goto done; }
    for (i = 0; i < m; i++)
      { roff[i]  = j->off[j->lo + i] - t0;
        rhoff[i] = j->hoff[j->lo + i] - h0;
        rooff[i] = j->ooff[j->lo + i] - o0;
      }
    rhoff[m] = j->hoff[j->hi] - h0;
    TRY(dupload(&pool, j->text + t0, (size_t) (t1 - t0), &d_text));
    TRY(dupload(&pool, roff, m * 8, &d_off));
    TRY(dupload(&pool, j->tlen + j->lo, m * 4, &d_tlen));
    TRY(dupload(&pool, j->nsym + j->lo, m * 4, &d_nsym));
    TRY(dupload(&pool, j->blob + h0, (size_t) rhoff[m], &d_hdr));
    TRY(dupload(&pool, rhoff, (m + 1) * 8, &d_hoff));
    TRY(dupload(&pool, rooff, m * 8, &d_ooff));
    TRY(dalloc(&pool, (size_t) obytes, &d_out));
    TRY(dx_pack2_encode(j->ctx, j->arrow ? DX_ALPHA_ARROW : DX_ALPHA_BASES, d_text, d_off, d_tlen, d_nsym, m,
                        d_hdr, d_hoff, d_out, d_ooff));
    TRY(dx_d2h(j->ctx, j->img + o0, d_out, (size_t) obytes));
  }
done:
  dfree_all(&pool);
  free(roff); free(rhoff); free(rooff);
  j->rc = rc;
  return NULL;
}

int dx_file_pack2_sharded(dx_ctx **ctxs, int nctx, int arrow, const uint8_t *text, size_t n,
                          uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ uint64_t  cnt = 0, i, *off = NULL, *hoff = NULL, *ooff = NULL;
  uint32_t *tlen = NULL, *nsym = NULL;
  int32_t  *hdr4 = NULL, lwell = 0;
  uint16_t *cnr4 = NULL;
  uint8_t  *blob = NULL, *img = NULL;
  size_t    plen = 0, at;
  p2_job   *jobs = NULL;
  pthread_t *th = NULL;
  int       rc, k, started = 0;

  if (ctxs == NULL || nctx < 1 || out == NULL || out_len == NULL) return DX_E_ARG;
  if (nctx == 1) return dx_file_pack2(ctxs[0], arrow, text, n, out, out_len, errline, errcode);
  *out = NULL; *out_len = 0;

  TRY(dx_index_seq(arrow, text, n, 0, NULL, NULL, NULL, NULL, NULL, &cnt, &plen, errline, errcode));
  off  = malloc((cnt + 1) * sizeof(*off));
  tlen = malloc((cnt + 1) * sizeof(*tlen));
  nsym = malloc((cnt + 1) * sizeof(*nsym));
  hdr4 = malloc((cnt + 1) * 4 * sizeof(*hdr4));
  cnr4 = malloc((cnt + 1) * 4 * sizeof(*cnr4));
  hoff = malloc((cnt + 1) * sizeof(*hoff));
  ooff = malloc((cnt + 1) * sizeof(*ooff));
  if (!off || !tlen || !nsym || !hdr4 || !cnr4 || !hoff || !ooff) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_index_seq(arrow, text, n, cnt, off, tlen, nsym, hdr4, cnr4, &cnt, &plen, errline, errcode));
  blob = malloc(dx_frame_bound(hdr4, cnt, 0, arrow) + 16);
  if (!blob) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_frame_headers(hdr4, cnr4, cnt, arrow, &lwell, blob, hoff));     /* one pass: well deltas chain over the whole file */

  at = 2 + 4 + plen;
  for (i = 0; i < cnt; i++)
    { ooff[i] = at;
      at += (size_t) (hoff[i+1] - hoff[i]) + (((size_t) nsym[i] + 3) >> 2);
    }
  ooff[cnt] = at;
  img = malloc(at + 16);
  if (!img) { rc = DX_E_NOMEM; goto done; }
  { uint16_t key = 0x55aa;
    int32_t  pl  = (int32_t) plen;
    memcpy(img, &key, 2);
    memcpy(img + 2, &pl, 4);
    memcpy(img + 6, text, plen);
  }

  jobs = calloc((size_t) nctx, sizeof(*jobs));
  th   = calloc((size_t) nctx, sizeof(*th));
  if (!jobs || !th) { rc = DX_E_NOMEM; goto done; }
  { uint64_t lo = 0;
    const uint64_t tbytes = cnt ? off[cnt - 1] + tlen[cnt - 1] - off[0] : 0;
    for (k = 0; k < nctx; k++)
      { uint64_t hi = lo;
        const uint64_t want = off[0] + tbytes / (uint64_t) nctx * (uint64_t) (k + 1);
        if (k == nctx - 1) hi = cnt;
        else while (hi < cnt && off[hi] < want) hi++;
        jobs[k].ctx = ctxs[k]; jobs[k].arrow = arrow; jobs[k].text = text; jobs[k].off = off; jobs[k].ooff = ooff;
        jobs[k].hoff = hoff; jobs[k].tlen = tlen; jobs[k].nsym = nsym; jobs[k].blob = blob; jobs[k].img = img;
        jobs[k].lo = lo; jobs[k].hi = hi;
        lo = hi;
      }
  }
  for (k = 0; k < nctx; k++)
    { if (pthread_create(&th[k], NULL, p2_main, &jobs[k]) != 0) { rc = DX_E_NOMEM; break; }
      started++;
    }
  for (k = 0; k < started; k++)
    pthread_join(th[k], NULL);
  if (started < nctx) goto done;
  rc = DX_OK;
  for (k = 0; k < nctx; k++)
    if (jobs[k].rc != DX_OK) { rc = jobs[k].rc; break; }
  if (rc == DX_OK)
    { *out = img; *out_len = at; img = NULL; }

done:
  free(off); free(hoff); free(ooff); free(tlen); free(nsym); free(hdr4); free(cnr4); free(blob); free(img);
  free(jobs); free(th);
  return rc;
}

/* ==========================================================================================
 *  undexta / undexar
 * ========================================================================================== */
typedef struct { const uint8_t *p; size_t n, at; int bad; } rsrc;

static void rd(rsrc *r, void *dst, size_t k)
{ if (r->at + k > r->n) { r->bad = 1; memset(dst, 0, k); r->at = r->n; return; }
  memcpy(dst, r->p + r->at, k);
  r->at += k;
}
static uint16_t sw16(uint16_t v) { return (uint16_t) ((v << 8) | (v >> 8)); }
static uint32_t sw32(uint32_t v) { return (v << 24) | ((v & 0xff00u) << 8) | ((v >> 8) & 0xff00u) | (v >> 24); }
static int32_t  rd_i32(rsrc *r, int flip) { uint32_t v; rd(r, &v, 4); return (int32_t) (flip ? sw32(v) : v); }
static uint16_t rd_u16(rsrc *r, int flip) { uint16_t v; rd(r, &v, 2); return flip ? sw16(v) : v; }

typedef struct { char *p; size_t len, cap; } tbuf;

static int tb_room(tbuf *b, size_t more)
{ if (b->len + more > b->cap)
    { size_t nc = (b->len + more) * 2 + 4096;
      char  *np = realloc(b->p, nc);
      if (np == NULL) return DX_E_NOMEM;
      b->p = np; b->cap = nc;
    }
  return DX_OK;
}

/* A chunk of decoded text on its way out (dx_d2h_stream): the header lines that fall into it are laid over it.
   Entry i's text starts at ooff[i]; its header line, hd[hat[i] .. hat[i+1]), ends there.                 */
typedef struct
  { uint64_t n; const uint64_t *ooff, *hat; const char *hd;
    dx_sink_fn sink; void *user;
    size_t base;                  /* where in the text the streamed buffer starts (a slice of the entries; else 0) */
  } hdr_patch;

static int patch_and_pass(void *arg, uint8_t *data, size_t len, size_t at0)
{ hdr_patch *h = arg;
  const size_t at = at0 + h->base;
  uint64_t lo = 0, hi = h->n, i;
  while (lo < hi)                                         /* first entry whose text starts beyond `at` */
    { uint64_t mid = (lo + hi) / 2;
      if (h->ooff[mid] > at) hi = mid; else lo = mid + 1;
    }
  for (i = lo; i < h->n; i++)
    { const size_t hl = (size_t) (h->hat[i+1] - h->hat[i]), h0 = (size_t) h->ooff[i] - hl, h1 = (size_t) h->ooff[i];
      const size_t c0 = h0 > at ? h0 : at, c1 = h1 < at + len ? h1 : at + len;
      if (h0 >= at + len) break;
      if (c0 < c1)
        memcpy(data + (c0 - at), h->hd + h->hat[i] + (c0 - h0), c1 - c0);
    }
  return h->sink(h->user, data, len, at);
}

/* how much of an output of `total` bytes the device makes at once beside an input of n bytes (and 48 bytes of index a unit): 0 = all
   of it; DEXGPU_TEXT_BUDGET (bytes) when set, else what is free decides */
static size_t out_cap(dx_ctx *ctx, size_t n, size_t total, uint64_t units)
{ const char *e = getenv("DEXGPU_TEXT_BUDGET");
  uint64_t fr = 0, all = 0;
  if (e != NULL && *e)
    { const unsigned long long v = strtoull(e, NULL, 10);
      return v && v < total ? (size_t) (v < 65536u ? 65536u : v) : 0;
    }
  if (dx_mem_info(ctx, &fr, &all) != DX_OK || fr == 0) return 0;
  if ((double) n + (double) total + 48.0 * (double) units <= 0.9 * (double) fr) return 0;
  { const double room = 0.9 * (double) fr - (double) n - 48.0 * (double) units;
    return room > (double) ((size_t) 4 << 20) ? (size_t) room : (size_t) 4 << 20;
  }
}

/* an image that arrives in pieces (dx_file_unpack2_stream): what the file's head said, the well the last record stood at, and
   how far into this piece the whole records reached (a piece may end inside a record: `more` says that more is coming) */
typedef struct { int started, flip, newv, well, more; int32_t plen; char *name; size_t consumed; } u2_state;

/* mode: DX_LETTERS_LOWER / _UPPER (dexta images) or _ARROW (dexar images); out != NULL: the text in memory,
   else through the sink */
static int unpack2_core(dx_ctx *ctx, int mode, const uint8_t *img, size_t n, uint32_t width,
                        uint8_t **out, dx_sink_fn sink, void *user, size_t *out_len, u2_state *st)
{ dpool     pool = { {0}, 0, ctx };
  rsrc      r = { img, n, 0, 0 };
  tbuf      hd = { NULL, 0, 0 };               /* all header lines, concatenated */
  uint64_t  cnt = 0, cap = 0, i, *ioff = NULL, *ooff = NULL, *hat = NULL;
  uint32_t *nsym = NULL;
  uint16_t  key;
  int       flip, newv, well = 0, rc, arrow = (mode == DX_LETTERS_ARROW);
  int32_t   plen;
  char     *name = NULL;
  uint8_t  *res = NULL;
  size_t    total = 0;
  void     *d_in, *d_ioff, *d_nsym, *d_out, *d_ooff;

  if (ctx == NULL || (out == NULL && sink == NULL) || out_len == NULL || img == NULL) return DX_E_ARG;
  if (width == 0) return DX_E_ARG;
  if (out) *out = NULL;
  *out_len = 0;

  if (st != NULL) st->consumed = 0;
  if (st != NULL && st->started)                          /* a later piece: records from its first byte on */
    { flip = st->flip; newv = st->newv; plen = st->plen; well = st->well;
      name = malloc((size_t) plen + 1);
      if (!name) return DX_E_NOMEM;
      memcpy(name, st->name, (size_t) plen + 1);
    }
  else
    { rd(&r, &key, 2);                                    /* undexta.c:138-159, undexar.c:136-145 */
      if (r.bad) return st != NULL && st->more ? DX_OK : DX_E_FORMAT;
      if (key == 0x55aa)               { flip = 0; newv = 1; }
      else if (key == 0xaa55)          { flip = 1; newv = 1; }
      else if (!arrow && key == 0x33cc) { flip = 0; newv = 0; }
      else if (!arrow && key == 0xcc33) { flip = 1; newv = 0; }
      else return DX_E_FORMAT;

      plen = rd_i32(&r, flip);                            /* undexta.c:161-169 */
      if (r.bad) return st != NULL && st->more ? DX_OK : DX_E_FORMAT;            /* (DX_OK, nothing consumed: the head is not all here yet) */
      if (plen < 0) return DX_E_FORMAT;
      if ((size_t) plen > n - r.at) return st != NULL && st->more && plen < (1 << 24) ? DX_OK : DX_E_FORMAT;
      name = malloc((size_t) plen + 1);
      if (!name) return DX_E_NOMEM;
      rd(&r, name, (size_t) plen);
      name[plen] = '\0';
      if (st != NULL)
        { st->name = malloc((size_t) plen + 1);
          if (st->name == NULL) { free(name); return DX_E_NOMEM; }
          memcpy(st->name, name, (size_t) plen + 1);
          st->started = 1; st->flip = flip; st->newv = newv; st->plen = plen;
          st->consumed = r.at;
        }
    }

  while (r.at < r.n)                                      /* undexta.c:175-271: walk the records */
    { uint8_t  byte;
      int      beg, end, qv = 0, k;
      uint16_t cnr[4] = { 0, 0, 0, 0 };
      uint32_t rlen;
      size_t   clen;
      const size_t rec_at = r.at;
      const int    well_was = well;

      rd(&r, &byte, 1);
      while (byte == 255 && !r.bad)
        { well += 255;
          rd(&r, &byte, 1);
        }
      well += byte;
      if (newv)
        { beg = rd_i32(&r, flip);
          end = rd_i32(&r, flip);
          if (arrow) for (k = 0; k < 4; k++) cnr[k] = rd_u16(&r, flip);
          else       qv = rd_i32(&r, flip);
        }
      else
        { beg = rd_u16(&r, flip); end = rd_u16(&r, flip); qv = rd_u16(&r, flip); }
      if (r.bad && st != NULL && st->more)                /* the piece ends inside this record's head: the next piece has it whole */
        { r.at = rec_at; r.bad = 0; well = well_was; break; }
      if (r.bad || end < beg || (int64_t) end - (int64_t) beg > 0x7fffffff)   /* (hostile headers: no int overflow) */
        { rc = DX_E_FORMAT; goto done; }
      rlen = (uint32_t) ((int64_t) end - (int64_t) beg);
      clen = ((size_t) rlen + 3) >> 2;
      if (r.at + clen > r.n)
        { if (st != NULL && st->more) { r.at = rec_at; well = well_was; break; }     /* ... or inside its bases */
          rc = DX_E_FORMAT; goto done;
        }

      if (cnt == cap)
        { void *t;                                        /* a failed realloc leaves the old block to `done` */
          cap  = cap ? 2 * cap : 1024;
          if ((t = realloc(ioff, cap * sizeof(*ioff))) == NULL) { rc = DX_E_NOMEM; goto done; }
          ioff = t;
          if ((t = realloc(ooff, cap * sizeof(*ooff))) == NULL) { rc = DX_E_NOMEM; goto done; }
          ooff = t;
          if ((t = realloc(hat, (cap + 1) * sizeof(*hat))) == NULL) { rc = DX_E_NOMEM; goto done; }
          hat = t;
          if ((t = realloc(nsym, cap * sizeof(*nsym))) == NULL) { rc = DX_E_NOMEM; goto done; }
          nsym = t;
        }
      if ((rc = tb_room(&hd, (size_t) plen + 160)) != DX_OK) goto done;
      hat[cnt] = hd.len;
      if (arrow)                                          /* undexar.c:199-203 */
        { float snr[4];
          for (k = 0; k < 4; k++) snr[k] = (float) (cnr[k] / 100.);
          hd.len += (size_t) sprintf(hd.p + hd.len, "%s/%d/%d_%d SN=%.2f,%.2f,%.2f,%.2f\n", name, well, beg, end,
                                     snr[0], snr[1], snr[2], snr[3]);
        }
      else                                                /* undexta.c:242 */
        hd.len += (size_t) sprintf(hd.p + hd.len, "%s/%d/%d_%d RQ=0.%d\n", name, well, beg, end, qv);

      ioff[cnt] = r.at;
      nsym[cnt] = rlen;
      r.at += clen;
      cnt  += 1;
    }
  if (cnt) hat[cnt] = hd.len;

  for (i = 0; i < cnt; i++)                               /* output layout: header line, wrapped text */
    { size_t L = nsym[i];
      total  += (size_t) (hat[i+1] - hat[i]);
      ooff[i] = total;
      total  += L + (L + width - 1) / width;
    }
  if (out)
    { res = malloc(total + 16);
      if (!res) { rc = DX_E_NOMEM; goto done; }
    }

  { /* A text that does not fit the device beside the image (or DEXGPU_TEXT_BUDGET): slices of whole reads, the image resident
       (a quarter of the text), every slice's text out before the next one's is made (the reference writes read after read,
       undexta.c:175-271). */
    const size_t cap = out_cap(ctx, n, total, cnt);
    if (cnt > 0 && cap)
      { uint64_t *rel = NULL, i0, i1, most = 0;
        size_t    tmax = 0;
        hdr_patch h = { cnt, ooff, hat, hd.p, sink, user, 0 };
#define TEXT_AT(i) ((i) < cnt ? (size_t) ooff[i] - (size_t) (hat[(i) + 1] - hat[i]) : total)
        for (i0 = 0; i0 < cnt; i0 = i1)
          { i1 = i0 + 1;
            while (i1 < cnt && TEXT_AT(i1 + 1) - TEXT_AT(i0) <= cap) i1++;
            if (TEXT_AT(i1) - TEXT_AT(i0) > tmax) tmax = TEXT_AT(i1) - TEXT_AT(i0);
            if (i1 - i0 > most) most = i1 - i0;
          }
        rel = malloc((most + 1) * sizeof(*rel));
        if (rel == NULL) { rc = DX_E_NOMEM; goto done; }
        rc = dupload(&pool, img, n, &d_in);
        if (rc == DX_OK) rc = dupload(&pool, ioff, cnt * 8, &d_ioff);
        if (rc == DX_OK) rc = dupload(&pool, nsym, cnt * 4, &d_nsym);
        if (rc == DX_OK) rc = dalloc(&pool, (most + 1) * 8, &d_ooff);
        if (rc == DX_OK) rc = dalloc(&pool, tmax, &d_out);
        for (i0 = 0; i0 < cnt && rc == DX_OK; i0 = i1)
          { const size_t t0 = TEXT_AT(i0);
            i1 = i0 + 1;
            while (i1 < cnt && TEXT_AT(i1 + 1) - t0 <= cap) i1++;
            for (i = i0; i < i1; i++) rel[i - i0] = ooff[i] - t0;
            rc = dx_h2d(ctx, d_ooff, rel, (i1 - i0) * 8);
            if (rc == DX_OK)
              rc = dx_pack2_decode(ctx, mode, d_in, (const uint64_t *) d_ioff + i0, (const uint32_t *) d_nsym + i0, i1 - i0, width, d_out, d_ooff);
            if (rc == DX_OK && out)
              { rc = dx_d2h(ctx, res + t0, d_out, TEXT_AT(i1) - t0);
                for (i = i0; i < i1 && rc == DX_OK; i++)
                  memcpy(res + ooff[i] - (hat[i+1] - hat[i]), hd.p + hat[i], (size_t) (hat[i+1] - hat[i]));
              }
            else if (rc == DX_OK)
              { h.base = t0;
                rc = dx_d2h_stream(ctx, d_out, TEXT_AT(i1) - t0, patch_and_pass, &h);
              }
          }
#undef TEXT_AT
        free(rel);
        if (rc != DX_OK) goto done;
        cnt = 0;                                          /* (done: nothing left for the one-shot path below) */
      }
  }
  if (cnt > 0)
    { TRY(dupload(&pool, img, n, &d_in));
      TRY(dupload(&pool, ioff, cnt * 8, &d_ioff));
      TRY(dupload(&pool, nsym, cnt * 4, &d_nsym));
      TRY(dupload(&pool, ooff, cnt * 8, &d_ooff));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_pack2_decode(ctx, mode, d_in, d_ioff, d_nsym, cnt, width, d_out, d_ooff));
      if (out)
        { TRY(dx_d2h(ctx, res, d_out, total));
          for (i = 0; i < cnt; i++)
            memcpy(res + ooff[i] - (hat[i+1] - hat[i]), hd.p + hat[i], (size_t) (hat[i+1] - hat[i]));
        }
      else
        { hdr_patch h = { cnt, ooff, hat, hd.p, sink, user, 0 };
          TRY(dx_d2h_stream(ctx, d_out, total, patch_and_pass, &h));
        }
    }
  if (out) { *out = res; res = NULL; }
  *out_len = total;
  if (st != NULL) { st->well = well; st->consumed = r.at; }
  rc = DX_OK;

done:
  dfree_all(&pool);
  free(name); free(hd.p); free(ioff); free(ooff); free(hat); free(nsym); free(res);
  return rc;
}

int dx_file_unpack2(dx_ctx *ctx, int mode, const uint8_t *img, size_t n, uint32_t width, uint8_t **out, size_t *out_len)
{ if (out == NULL) return DX_E_ARG;
  return unpack2_core(ctx, mode, img, n, width, out, NULL, NULL, out_len, NULL);
}

int dx_file_unpack2_to(dx_ctx *ctx, int mode, const uint8_t *img, size_t n, uint32_t width,
                       dx_sink_fn sink, void *user, size_t *out_len)
{ if (sink == NULL) return DX_E_ARG;
  return unpack2_core(ctx, mode, img, n, width, NULL, sink, user, out_len, NULL);
}

/* undexta / undexar of an image that arrives in pieces (a pipe: undexta -i, undexta.c:175-271 reads record after record): `chunk`
   bytes at a time from rd(), the whole records among them unpacked on the device, their text handed to the sink in file order, the
   rest (a record the chunk cuts) moved to the buffer's front.  The same bytes as dx_file_unpack2 of the whole image. */
typedef struct { dx_sink_fn sink; void *user; size_t shift; } u2_shift;
static int u2_pass(void *arg, uint8_t *data, size_t len, size_t at)
{ u2_shift *h = arg;
  return h->sink(h->user, data, len, at + h->shift);
}

int dx_file_unpack2_stream(dx_ctx *ctx, int mode, dx_read_fn rd_, void *ruser, size_t chunk, uint32_t width,
                           dx_sink_fn sink, void *suser, size_t *out_len)
{ uint8_t *buf = NULL;
  size_t   have = 0, total = 0;
  u2_state st;
  int      eof = 0, rc = DX_OK;

  if (ctx == NULL || rd_ == NULL || sink == NULL || width == 0) return DX_E_ARG;
  memset(&st, 0, sizeof(st));
  if (chunk == 0) chunk = (size_t) dx_test_num("stream_chunk", (long long) 128 << 20);
  if (chunk < 4096) chunk = 4096;
  buf = malloc(chunk + 16);
  if (buf == NULL) return DX_E_NOMEM;
  if (out_len) *out_len = 0;
  for (;;)
    { size_t piece = 0;
      u2_shift h = { sink, suser, total };
      while (!eof && have < chunk)
        { const long got = rd_(ruser, buf + have, chunk - have);
          if (got < 0) { rc = DX_E_IO; goto done; }
          if (got == 0) eof = 1;
          have += (size_t) got;
        }
      st.more = !eof;
      rc = unpack2_core(ctx, mode, buf, have, width, NULL, u2_pass, &h, &piece, &st);
      if (rc != DX_OK) goto done;
      total += piece;
      if (eof) break;                                    /* (the last piece: whole, or the core has said DX_E_FORMAT) */
      if (st.consumed == 0)                              /* not one whole record in the buffer: a larger one */
        { uint8_t *nb;
          chunk += chunk;
          nb = realloc(buf, chunk + 16);
          if (nb == NULL) { rc = DX_E_NOMEM; goto done; }
          buf = nb;
          continue;
        }
      memmove(buf, buf + st.consumed, have - st.consumed);
      have -= st.consumed;
    }
  if (out_len) *out_len = total;
done:
  free(st.name);
  free(buf);
  return rc;
}

/* ==========================================================================================
 *  dexqv
 * ========================================================================================== */
/* a sink that sees its chunks `shift` bytes further on (the record stream follows the file's head) */
typedef struct { dx_sink_fn sink; void *user; size_t shift; } shifted_sink;
static int pass_shifted(void *arg, uint8_t *data, size_t len, size_t at)
{ shifted_sink *h = arg;
  return h->sink(h->user, data, len, at + h->shift);
}


/* ---- a .quiva image larger than the device (or than DEXGPU_TEXT_BUDGET): slices of whole entries -------------------
 * The reference streams a file of any size through two passes (dexqv.c:81-82, 112-143).  Here: the host index of the
 * whole image (line structure, header fields), then per slice of at most `cap` bytes of text
 *   pass 1: upload, dx_qv_prescan (the scan state carried from slice to slice, entry0 = the slice's first entry),
 *           dx_qv_hist (adds into the file's histograms);
 *   tables, the file's head (key + coding) out;
 *   pass 2: upload again, dx_qv_hist once more (for the tokens of THIS slice under the final scan state; its counts go
 *           nowhere), dx_qv_encode_onepass, the slice's records out behind the last slice's.
 * A slice is bound by the host link (two uploads of the text at ~50 GB/s against kernels at ~1.7 TB/s), so the second
 * histogram pass costs nothing that shows.  The well chain of the framing bytes runs through the slices.          */
typedef struct { uint8_t *p; size_t n, cap; } grow_sink;
static int grow_take(void *arg, uint8_t *data, size_t len, size_t at)
{ grow_sink *g = arg;
  if (at + len > g->cap)
    { size_t nc = 2 * g->cap + at + len + 4096;
      uint8_t *t = realloc(g->p, nc);
      if (t == NULL) return 1;
      g->p = t; g->cap = nc;
    }
  memcpy(g->p + at, data, len);
  if (at + len > g->n) g->n = at + len;
  return 0;
}

static int dexqv_sliced(dx_ctx *ctx, const uint8_t *text, size_t n, int lossy, size_t cap, uint8_t **out, dx_sink_fn sink, void *user,
                        size_t *out_len, uint64_t *errline, int *errcode)
{ dpool        pool = { {0}, 0, ctx };
  uint64_t     cnt = 0, *off = NULL, *hoff = NULL, *rel = NULL, tot = 0, e0, at;
  uint32_t    *len = NULL;
  int32_t     *hdr4 = NULL, lwell = 0;
  uint8_t     *blob = NULL, *head_img = NULL;
  size_t       plen = 0, clen = 0, head = 0, maxent = 0, slice_bytes = 0;
  dx_qv_params p = { -1, -1, -1, -1 };
  dx_qv_coding *cd = NULL;
  uint64_t   (*hist)[256] = NULL, (*junk)[256] = NULL;
  void        *d_text = NULL, *d_off = NULL, *d_len = NULL, *d_hdr = NULL, *d_hoff = NULL, *d_rec = NULL, *d_seg = NULL, *d_out = NULL;
  size_t       out_cap = 0;
  grow_sink    grow = { NULL, 0, 0 };
  int          rc, pass, was_threads = 0;

  if (out) { sink = grow_take; user = &grow; was_threads = dx_set_sink_threads(ctx, 1); }     /* (grow_take wants its chunks in order) */
  cd = malloc(sizeof(*cd)); hist = calloc(6, sizeof(*hist)); junk = calloc(6, sizeof(*junk));
  if (!cd || !hist || !junk) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_index_quiva(text, n, 0, NULL, NULL, NULL, &cnt, &plen, errline, errcode));
  if (cnt == 0) { rc = DX_E_DEGENERATE; goto done; }
  off = malloc((cnt + 1) * sizeof(*off)); len = malloc((cnt + 1) * sizeof(*len)); hdr4 = malloc((cnt + 1) * 4 * sizeof(*hdr4));
  if (!off || !len || !hdr4) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_index_quiva(text, n, cnt, off, len, hdr4, &cnt, &plen, errline, errcode));
  /* the widest slice in entries and bytes under the cap (an entry larger than the cap is a slice of its own) */
  for (e0 = 0; e0 < cnt; )
    { const uint64_t s0 = e0 ? off[e0 - 1] + 5 * ((uint64_t) len[e0 - 1] + 1) : 0;
      uint64_t e1 = e0 + 1;
      while (e1 < cnt && off[e1] + 5 * ((uint64_t) len[e1] + 1) - s0 <= cap) e1++;
      if (e1 - e0 > maxent) maxent = (size_t) (e1 - e0);
      if (off[e1 - 1] + 5 * ((uint64_t) len[e1 - 1] + 1) - s0 > slice_bytes) slice_bytes = (size_t) (off[e1 - 1] + 5 * ((uint64_t) len[e1 - 1] + 1) - s0);
      e0 = e1;
    }
  hoff = malloc((maxent + 1) * sizeof(*hoff)); rel = malloc((maxent + 1) * sizeof(*rel));
  if (!hoff || !rel) { rc = DX_E_NOMEM; goto done; }
  TRY(dalloc(&pool, slice_bytes, &d_text));
  TRY(dalloc(&pool, (maxent + 1) * 8, &d_off));
  TRY(dalloc(&pool, (maxent + 1) * 4, &d_len));
  TRY(dalloc(&pool, (maxent + 1) * 8, &d_hoff));
  TRY(dalloc(&pool, (maxent + 1) * 8, &d_rec));
  TRY(dalloc(&pool, maxent * 5 * 4 + 64, &d_seg));

  at = 0;
  for (pass = 1; pass <= 2; pass++)
    { if (pass == 2)
        { TRY(dx_qv_build((const uint64_t (*)[256]) hist, tot, &p, lossy, cd));          /* Create_QVcoding, dexqv.c:86 */
          TRY(dx_qv_set_coding(ctx, cd, lossy));
          rc = dx_qv_write_coding(cd, (const char *) text, plen, NULL, 0, &clen);
          if (rc != DX_OK && rc != DX_E_SPACE) goto done;
          head = 2 + clen;
          head_img = malloc(head + 16);
          if (!head_img) { rc = DX_E_NOMEM; goto done; }
          { uint16_t key = 0x55aa;                                                       /* dexqv.c:105-108 */
            memcpy(head_img, &key, 2);
            TRY(dx_qv_write_coding(cd, (const char *) text, plen, head_img + 2, clen, &clen));
          }
          if (sink(user, head_img, head, 0)) { rc = DX_E_IO; goto done; }
          at = head;
        }
      for (e0 = 0; e0 < cnt; )
        { const uint64_t s0 = e0 ? off[e0 - 1] + 5 * ((uint64_t) len[e0 - 1] + 1) : 0;
          uint64_t e1 = e0 + 1, s1, k, m, total = 0;
          dx_qv_batch b;
          while (e1 < cnt && off[e1] + 5 * ((uint64_t) len[e1] + 1) - s0 <= cap) e1++;
          s1 = off[e1 - 1] + 5 * ((uint64_t) len[e1 - 1] + 1);
          m  = e1 - e0;
          for (k = 0; k < m; k++) rel[k] = off[e0 + k] - s0;
          TRY(dx_h2d(ctx, d_text, text + s0, (size_t) (s1 - s0)));
          TRY(dx_h2d(ctx, d_off, rel, (size_t) m * 8));
          TRY(dx_h2d(ctx, d_len, len + e0, (size_t) m * 4));
          b.d_text = d_text; b.d_off = d_off; b.d_len = d_len; b.n = m; b.line_pad = 1; b.text_bytes = s1 - s0;
          if (pass == 1)
            TRY(dx_qv_scan(ctx, &b, e0, &p, hist, &tot));                               /* QV.c:993-1017, state carried along */
          else
            { uint64_t t2 = 0, bound;
              size_t   bb;
              memset(junk, 0, 6 * sizeof(*junk));
              TRY(dx_qv_hist(ctx, &b, e0, &p, junk, &t2));                               /* this slice's tokens (and its own counts, for the bound) */
              bb = dx_frame_bound(hdr4 + 4 * e0, m, lwell, 0) + 16;
              { uint8_t *nb = realloc(blob, bb);
                if (nb == NULL) { rc = DX_E_NOMEM; goto done; }
                blob = nb;
              }
              TRY(dx_frame_headers(hdr4 + 4 * e0, NULL, m, 0, &lwell, blob, hoff));
              if (d_hdr) { dx_free(ctx, d_hdr); d_hdr = NULL; }
              TRY(dx_malloc(ctx, (size_t) hoff[m] + 64, &d_hdr));
              TRY(dx_h2d(ctx, d_hdr, blob, (size_t) hoff[m]));
              TRY(dx_h2d(ctx, d_hoff, hoff, (size_t) (m + 1) * 8));
              bound = hoff[m] + dx_qv_out_bound((const uint64_t (*)[256]) junk, m, cd, lossy);
              if (bound > out_cap)
                { if (d_out) { dx_free(ctx, d_out); d_out = NULL; }
                  TRY(dx_malloc(ctx, (size_t) bound + 64, &d_out));
                  out_cap = (size_t) bound;
                }
              TRY(dx_qv_encode_onepass(ctx, &b, d_hdr, d_hoff, d_seg, d_rec, d_out, out_cap, &total));
              { shifted_sink h = { sink, user, (size_t) at };
                TRY(dx_d2h_stream(ctx, d_out, total, pass_shifted, &h));
              }
              at += total;
            }
          e0 = e1;
        }
    }
  *out_len = (size_t) at;
  if (out) { *out = grow.p; grow.p = NULL; }
  rc = DX_OK;

done:
  if (was_threads) (void) dx_set_sink_threads(ctx, was_threads);
  if (d_hdr) dx_free(ctx, d_hdr);
  if (d_out) dx_free(ctx, d_out);
  dfree_all(&pool);
  (void) dx_trim(ctx, DX_TRIM_TOKENS);                     /* (a slice's tokens must not meet another batch that looks like it) */
  free(off); free(hoff); free(rel); free(len); free(hdr4); free(blob); free(cd); free(hist); free(junk); free(head_img); free(grow.p);
  return rc;
}

/* how much text the device takes at once: DEXGPU_TEXT_BUDGET (bytes) when set, else what fits beside the tokens, the scratch
   regions and the output (about 2.5 bytes of device memory per byte of text), 0 = all of it */
static size_t text_cap(dx_ctx *ctx, size_t n)
{ const char *e = getenv("DEXGPU_TEXT_BUDGET");
  uint64_t fr = 0, all = 0;
  if (e != NULL && *e)
    { const unsigned long long v = strtoull(e, NULL, 10);
      return v && v < n ? (size_t) (v < (4u << 20) ? (4u << 20) : v) : 0;
    }
  if (dx_mem_info(ctx, &fr, &all) != DX_OK || fr == 0) return 0;
  return (double) n * 2.5 > (double) fr ? (size_t) (fr / 3) : 0;
}

/* out != NULL: the image in memory; else through the sink, in order, nothing before all of it is known to exist */
/* text == NULL: the image is the first n bytes of the file behind fd (dx_file_dexqv_fd_to): uploaded by dx_h2d_fd, and whatever
   wants it in memory -- a small file, slices, the host indexer's words for a malformed one -- is DX_E_AGAIN */
static int dexqv_core(dx_ctx *ctx, const uint8_t *text, int fd, size_t n, int lossy, uint8_t **out, dx_sink_fn sink, void *user,
                      size_t *out_len, uint64_t *errline, int *errcode)
{ dpool        pool = { {0}, 0, ctx };
  uint8_t      headbuf[4096];
  uint64_t     cnt = 0, *off = NULL, *hoff = NULL, total = 0, tot = 0;
  uint32_t    *len = NULL;
  int32_t     *hdr4 = NULL, lwell = 0;
  uint8_t     *blob = NULL, *img = NULL;
  size_t       plen = 0, clen = 0, head;
  dx_qv_batch  b;
  dx_qv_params p = { -1, -1, -1, -1 };
  dx_qv_coding *cd = NULL;
  uint64_t   (*hist)[256] = NULL;
  void        *d_text, *d_off = NULL, *d_len = NULL, *d_hdr, *d_hoff, *d_rec, *d_seg, *d_out;
  int          rc;

  if (ctx == NULL || (out == NULL && sink == NULL) || out_len == NULL) return DX_E_ARG;
  if (out) *out = NULL;
  *out_len = 0;
  { const size_t cap = text_cap(ctx, n);
    if (cap)
      return text == NULL ? DX_E_AGAIN : dexqv_sliced(ctx, text, n, lossy, cap, out, sink, user, out_len, errline, errcode);
  }
  if (text == NULL && (n < DX_GPU_INDEX_MIN || dx_test_on("host_index"))) return DX_E_AGAIN;

  /* pass 1 of the reference (QVcoding_Scan, dexqv.c:81-82): validate + index.  Large images are
   * indexed on the GPU (newline scan + structure checks there, only the header lines come back);
   * small ones, and any image the GPU front end rejects (so that the message is exactly the
   * reference's first one), by the host indexer.                                               */
  cd   = malloc(sizeof(*cd));
  hist = calloc(6, sizeof(*hist));
  if (!cd || !hist) { rc = DX_E_NOMEM; goto done; }
  fmark("dexqv: begin");
  if (text != NULL) TRY(dupload(&pool, text, n, &d_text));
  else
    { size_t got = 0, want = n < sizeof(headbuf) ? n : sizeof(headbuf);
      TRY(dalloc(&pool, n, &d_text));
      TRY(dx_h2d_fd(ctx, d_text, fd, 0, n));
      while (got < want)                                  /* (the first header line, for the coding's prefix) */
        { const ssize_t k = pread(fd, headbuf + got, want - got, (off_t) got);
          if (k <= 0) { rc = DX_E_IO; goto done; }
          got += (size_t) k;
        }
    }
  fmark("dexqv: text on the device");
  if (n >= DX_GPU_INDEX_MIN && !dx_test_on("host_index"))
    { uint64_t *go = NULL; uint32_t *gl = NULL;
      rc = dx_index_quiva_device(ctx, d_text, n, &go, &gl, &cnt, &hdr4, &plen, errline, errcode);
      if (rc == DX_OK && cnt > 0)
        { d_off = go; d_len = gl;
          pool.p[pool.n++] = go; pool.p[pool.n++] = gl;
        }
      else if (rc != DX_OK && rc != DX_E_FORMAT)
        goto done;
      else if (text == NULL)                              /* (malformed, or empty: the in-memory driver says what is wrong) */
        { rc = DX_E_AGAIN; goto done; }
      rc = DX_OK;
    }
  if (text == NULL)
    { if (plen >= sizeof(headbuf)) { rc = DX_E_AGAIN; goto done; }
      text = headbuf;                                     /* (from here on only the prefix is looked at) */
    }
  if (d_off == NULL)
    { TRY(dx_index_quiva(text, n, 0, NULL, NULL, NULL, &cnt, &plen, errline, errcode));
      off  = malloc((cnt + 1) * sizeof(*off));
      len  = malloc((cnt + 1) * sizeof(*len));
      free(hdr4);
      hdr4 = malloc((cnt + 1) * 4 * sizeof(*hdr4));
      if (!off || !len || !hdr4) { rc = DX_E_NOMEM; goto done; }
      TRY(dx_index_quiva(text, n, cnt, off, len, hdr4, &cnt, &plen, errline, errcode));
      if (cnt > 0)
        { TRY(dupload(&pool, off, cnt * 8, &d_off));
          TRY(dupload(&pool, len, cnt * 4, &d_len));
        }
    }
  if (cnt == 0)
    { rc = DX_E_DEGENERATE;     /* empty file: the reference dereferences a NULL header (dexqv.c:94) */
      goto done;
    }
  fmark("dexqv: indexed");
  hoff = malloc((cnt + 1) * sizeof(*hoff));
  if (!hoff) { rc = DX_E_NOMEM; goto done; }

  blob = malloc(dx_frame_bound(hdr4, cnt, 0, 0) + 16);
  if (!blob) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_frame_headers(hdr4, NULL, cnt, 0, &lwell, blob, hoff));

  TRY(dupload(&pool, blob, (size_t) hoff[cnt], &d_hdr));
  TRY(dupload(&pool, hoff, (cnt + 1) * 8, &d_hoff));
  TRY(dalloc(&pool, (cnt + 1) * 8, &d_rec));
  TRY(dalloc(&pool, cnt * 5 * 4, &d_seg));
  b.d_text = d_text; b.d_off = d_off; b.d_len = d_len; b.n = cnt; b.line_pad = 1;
  b.text_bytes = n;

  /* ... and histogram on the device (QV.c:988-1017) */
  TRY(dx_qv_scan(ctx, &b, 0, &p, hist, &tot));
  TRY(dx_qv_build((const uint64_t (*)[256]) hist, tot, &p, lossy, cd));   /* Create_QVcoding, dexqv.c:86 */
  TRY(dx_qv_set_coding(ctx, cd, lossy));
  fmark("dexqv: scanned, tables built");

  rc = dx_qv_write_coding(cd, (const char *) text, plen, NULL, 0, &clen);  /* size of Write_QVcoding */
  if (rc != DX_OK && rc != DX_E_SPACE) goto done;
  head = 2 + clen;

  /* pass 2, dexqv.c:112-143: Compress_Next_QVentry for every entry */
  if (!two_pass())
    { const uint64_t cap = hoff[cnt] + dx_qv_out_bound((const uint64_t (*)[256]) hist, cnt, cd, lossy);
      TRY(dalloc(&pool, cap, &d_out));
      TRY(dx_qv_encode_onepass(ctx, &b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap, &total));
    }
  else
    { TRY(dx_qv_sizes(ctx, &b, d_hoff, d_seg, d_rec, &total));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_qv_encode(ctx, &b, d_hdr, d_hoff, d_rec, d_seg, d_out));
    }
  fmark("dexqv: encoded");
  img = malloc(head + (out ? total : 0) + 16);
  if (!img) { rc = DX_E_NOMEM; goto done; }
  { uint16_t key = 0x55aa;                                                 /* dexqv.c:105-108 */
    memcpy(img, &key, 2);
    TRY(dx_qv_write_coding(cd, (const char *) text, plen, img + 2, clen, &clen));
  }
  if (out)
    { TRY(dx_d2h(ctx, img + head, d_out, total));
      *out = img; img = NULL;
    }
  else
    { shifted_sink h = { sink, user, head };
      if (sink(user, img, head, 0)) { rc = DX_E_IO; goto done; }
      TRY(dx_d2h_stream(ctx, d_out, total, pass_shifted, &h));
    }
  *out_len = head + total;
  rc = DX_OK;
  fmark("dexqv: output passed on");

done:
  dfree_all(&pool);
  free(off); free(hoff); free(len); free(hdr4); free(blob); free(cd); free(hist); free(img);
  fmark("dexqv: device memory released");
  return rc;
}

int dx_file_dexqv(dx_ctx *ctx, const uint8_t *text, size_t n, int lossy,
                  uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ if (out == NULL || text == NULL) return DX_E_ARG;
  return dexqv_core(ctx, text, -1, n, lossy, out, NULL, NULL, out_len, errline, errcode);
}

int dx_file_dexqv_fd_to(dx_ctx *ctx, int fd, size_t n, int lossy, dx_sink_fn sink, void *user,
                        size_t *out_len, uint64_t *errline, int *errcode)
{ if (sink == NULL || fd < 0) return DX_E_ARG;
  return dexqv_core(ctx, NULL, fd, n, lossy, NULL, sink, user, out_len, errline, errcode);
}

int dx_file_dexqv_to(dx_ctx *ctx, const uint8_t *text, size_t n, int lossy, dx_sink_fn sink, void *user,
                     size_t *out_len, uint64_t *errline, int *errcode)
{ if (sink == NULL || text == NULL) return DX_E_ARG;
  return dexqv_core(ctx, text, -1, n, lossy, NULL, sink, user, out_len, errline, errcode);
}

/* ==========================================================================================
 *  undexqv
 * ========================================================================================== */
/* undexqv in two steps (dexgpu.h): the plan is host work only, the run is the GPU's */
struct dx_undexqv_plan
  { const uint8_t *img;
    size_t         n, total;
    dx_qv_index    x;
    tbuf           hd;            /* the header lines, one after the other */
    uint64_t      *ooff, *hat;    /* per entry: where its five data lines start in the text; where its header line starts in hd */
    /* a plan made on the device (dx_file_undexqv_plan_on): the image is there already, and so is the index */
    dx_ctx        *ctx;
    void          *d_in;          /* the image, when it is on ctx's device already */
    dx_qv_dindex   dix;           /* the index, when it was made there (d_rec_off != NULL) */
  };
#define PLAN_HAS_IMAGE(p) ((p)->ctx != NULL && (p)->d_in != NULL)
#define PLAN_HAS_INDEX(p) ((p)->ctx != NULL && (p)->dix.d_rec_off != NULL)

void dx_file_undexqv_plan_free(dx_undexqv_plan *p)
{ if (p == NULL) return;
  dx_qv_index_free(&p->x);
  if (p->ctx != NULL)
    { dx_qv_dindex_free(p->ctx, &p->dix);
      if (p->d_in != NULL) (void) dx_free(p->ctx, p->d_in);
    }
  free(p->ooff); free(p->hat); free(p->hd.p);
  free(p);
}

/* header lines (undexqv.c:182) and where every entry's lines go in the text, from p->x.n / len / hdr4 / prefix */
static int plan_layout(dx_undexqv_plan *p)
{ const size_t plen = strlen(p->x.prefix);
  size_t   total = 0;
  uint64_t i;
  int      rc;
  p->ooff = malloc((p->x.n + 1) * sizeof(*p->ooff));
  p->hat  = malloc((p->x.n + 1) * sizeof(*p->hat));
  if (!p->ooff || !p->hat) return DX_E_NOMEM;
  for (i = 0; i < p->x.n; i++)
    { const int32_t *h = p->x.hdr4 + 4*i;
      if ((rc = tb_room(&p->hd, plen + 80)) != DX_OK) return rc;
      p->hat[i]  = p->hd.len;
      p->hd.len += (size_t) sprintf(p->hd.p + p->hd.len, "%s/%d/%d_%d RQ=0.%d\n", p->x.prefix, h[0], h[1], h[2], h[3]);
      total     += p->hd.len - (size_t) p->hat[i];
      p->ooff[i] = total;
      total     += 5 * ((size_t) p->x.len[i] + 1);        /* undexqv.c:206-207 */
    }
  p->hat[p->x.n] = p->hd.len;
  p->ooff[p->x.n] = total;
  p->total = total;
  return DX_OK;
}

/* The plan of a large 0x55aa-keyed image with the GPU at hand: the image goes to the device (where the run wants it anyway),
   the records are walked THERE (dx_qv_walk_device: a lane per 32 KiB piece; 0.1 s for 14 GB of records where 32 host
   threads take 7.5 s), and only the entries' lengths and header fields come back for the header lines.  Whatever the device
   walk does not take -- small images (the host walk is over before the device's tables are up), 16-bit framing fields,
   walks that do not chain up, a damaged stream -- is planned on the host as before (dx_file_undexqv_plan), which also
   has the words for what is wrong with a file.  DEXGPU_HOST_WALK=1: always on the host.                             */
#define DX_DEVICE_WALK_MIN ((size_t) 256 << 20)
int dx_file_undexqv_plan_on(dx_ctx *ctx, const uint8_t *img, size_t n, dx_undexqv_plan **plan, size_t *out_len)
{ dx_undexqv_plan *p;
  uint16_t key;
  size_t   at = 2, used = 0;
  int      rc, keep = 0;
  const size_t least = (size_t) dx_test_num("device_walk_min", (long long) DX_DEVICE_WALK_MIN);

  if (img == NULL || plan == NULL || out_len == NULL) return DX_E_ARG;
  if (ctx == NULL || n < least || n < 16 || dx_test_on("host_walk"))
    return dx_file_undexqv_plan(img, n, plan, out_len);
  memcpy(&key, img, 2);
  if (key != 0x55aa && key != 0xaa55)
    return dx_file_undexqv_plan(img, n, plan, out_len);
  *plan = NULL; *out_len = 0;
  p = calloc(1, sizeof(*p));
  if (p == NULL) return DX_E_NOMEM;
  p->x.newv = 1;
  { uint16_t k2 = 0;                                      /* the coding, as dx_qv_walk reads it (QV.c:1222-1256) */
    uint32_t pl = 0;
    memcpy(&k2, img + at, 2);
    memcpy(&pl, img + at + 6, 4);
    if (k2 != 0x33cc) pl = ((pl & 0xffu) << 24) | ((pl & 0xff00u) << 8) | ((pl >> 8) & 0xff00u) | (pl >> 24);
    rc = (uint64_t) pl > (uint64_t) (n - at - 10) ? DX_E_FORMAT : DX_OK;
    if (rc == DX_OK && (p->x.prefix = malloc((size_t) pl + 1)) == NULL) rc = DX_E_NOMEM;
    if (rc == DX_OK) rc = dx_qv_read_coding(img + at, n - at, &p->x.coding, &p->x.flip, p->x.prefix, (size_t) pl + 1, &used);
  }
  if (rc != DX_OK) goto host;
  at += used;
  { /* image, walk scratch (records 0.7, the lanes' words for the group index 1.1 of the image) and index (0.3) must fit together;
       asked before anything goes up (dx_qv_walk_device asks again, to the byte) */
    uint64_t fr = 0, all = 0;
    if (dx_mem_info(ctx, &fr, &all) == DX_OK && fr > 0 && 3.2 * (double) n + (double) (128 << 20) > 0.95 * (double) fr)
      goto host;
  }
  p->ctx = ctx;
  if ((rc = dx_malloc(ctx, n + 64, &p->d_in)) != DX_OK) { p->d_in = NULL; goto host; }
  if ((rc = dx_h2d(ctx, p->d_in, img, n)) != DX_OK) goto host;
  rc = dx_qv_walk_device(ctx, p->d_in, n, at, &p->x.coding, 1, p->x.flip, &p->dix);
  if (rc != DX_OK) { keep = rc != DX_E_NOMEM && rc != DX_E_HIP; goto host; }
  p->x.n    = p->dix.n;
  p->x.len  = malloc((p->x.n + 1) * sizeof(uint32_t));
  p->x.hdr4 = malloc((p->x.n + 1) * 4 * sizeof(int32_t));
  if (!p->x.len || !p->x.hdr4) { rc = DX_E_NOMEM; goto fail; }
  if (p->x.n > 0 && ((rc = dx_d2h(ctx, p->x.len, p->dix.d_len, p->x.n * 4)) != DX_OK ||
                     (rc = dx_d2h(ctx, p->x.hdr4, p->dix.d_hdr4, p->x.n * 16)) != DX_OK)) goto fail;
  p->img = img; p->n = n;
  if ((rc = plan_layout(p)) != DX_OK) goto fail;
  *plan = p; *out_len = p->total;
  return DX_OK;

host:                                                     /* not the device's: the host walk (and its verdict) */
  { void *d_in = keep ? p->d_in : NULL;                   /* an image that is up stays up: the run wants it there */
    if (d_in != NULL) p->d_in = NULL;
    dx_file_undexqv_plan_free(p);
    rc = dx_file_undexqv_plan(img, n, plan, out_len);
    if (d_in != NULL)
      { if (rc == DX_OK) { (*plan)->ctx = ctx; (*plan)->d_in = d_in; }
        else             (void) dx_free(ctx, d_in);
      }
    return rc;
  }
fail:
  dx_file_undexqv_plan_free(p);
  return rc;
}

int dx_file_undexqv_plan(const uint8_t *img, size_t n, dx_undexqv_plan **plan, size_t *out_len)
{ dx_undexqv_plan *p;
  int      rc;

  if (img == NULL || plan == NULL || out_len == NULL) return DX_E_ARG;
  *plan = NULL; *out_len = 0;
  p = calloc(1, sizeof(*p));
  if (p == NULL) return DX_E_NOMEM;
  /* boundary walk (host).  With DEXGPU_WALK_INDEX set it also leaves the group index the wave-per-line decoders take
     (dx_qv_use_index below): 31 instead of 50 ms of kernels per 14 GB of records -- but the walk is 45 % longer with it
     and the index is another 30 % to upload, and from file to file that costs more than it saves (undexqv of a 1 GB
     .quiva: 0.54-0.59 s with, 0.44-0.48 s without; profiles/r03c_cli_timing.txt), so it is off unless asked for */
  rc = dx_qv_walk_indexed(img, n, &p->x, dx_test_on("walk_index"));
  if (rc != DX_OK) { free(p); return rc; }
  p->img = img; p->n = n;
  if ((rc = plan_layout(p)) != DX_OK) goto fail;
  *plan = p; *out_len = p->total;
  return DX_OK;

fail:
  dx_file_undexqv_plan_free(p);
  return rc;
}

/* The record index a plan holds, as host arrays of the caller's (dx_qv_index_free): n, rec_off, hdr_off, seg, len, hdr4, the
   coding, prefix, newv / flip -- copied from the host walk's, or downloaded when the plan was made on the device. */
int dx_file_undexqv_plan_index(const dx_undexqv_plan *p, dx_qv_index *x)
{ const uint64_t n = p ? p->x.n : 0;
  int rc = DX_OK;
  if (p == NULL || x == NULL) return DX_E_ARG;
  memset(x, 0, sizeof(*x));
  x->n = n; x->coding = p->x.coding; x->newv = p->x.newv; x->flip = p->x.flip;
  x->rec_off = malloc((n + 1) * sizeof(uint64_t));
  x->hdr_off = malloc((n + 1) * sizeof(uint64_t));
  x->seg     = malloc((n + 1) * 5 * sizeof(uint32_t));
  x->len     = malloc((n + 1) * sizeof(uint32_t));
  x->hdr4    = malloc((n + 1) * 4 * sizeof(int32_t));
  x->prefix  = malloc(strlen(p->x.prefix) + 1);
  if (!x->rec_off || !x->hdr_off || !x->seg || !x->len || !x->hdr4 || !x->prefix) { dx_qv_index_free(x); return DX_E_NOMEM; }
  strcpy(x->prefix, p->x.prefix);
  memcpy(x->len, p->x.len, n * sizeof(uint32_t));
  memcpy(x->hdr4, p->x.hdr4, n * 4 * sizeof(int32_t));
  if (PLAN_HAS_INDEX(p))
    { if ((rc = dx_d2h(p->ctx, x->rec_off, p->dix.d_rec_off, (n + 1) * 8)) == DX_OK &&
          (rc = dx_d2h(p->ctx, x->hdr_off, p->dix.d_hdr_off, (n + 1) * 8)) == DX_OK && n > 0)
        rc = dx_d2h(p->ctx, x->seg, p->dix.d_seg, n * 20);
    }
  else
    { memcpy(x->rec_off, p->x.rec_off, (n + 1) * 8);
      memcpy(x->hdr_off, p->x.hdr_off, (n + 1) * 8);
      memcpy(x->seg, p->x.seg, n * 20);
    }
  if (rc != DX_OK) dx_qv_index_free(x);
  return rc;
}

/* ---- a text larger than the device (or than DEXGPU_TEXT_BUDGET): slices of whole entries ------------------------------
 * The reference writes entry after entry (undexqv.c:182-207).  Here: per slice of at most `cap` bytes of text, the slice's
 * records -- the whole image stays on the device when it is there already (a plan made there) or fits beside a slice's text,
 * else the slice's bytes are uploaded -- are decoded into one buffer that goes out before the next slice comes in.
 * Same text; such a file is bound by the host link.                                                               */
static int undexqv_sliced(dx_ctx *ctx, const dx_undexqv_plan *p, int upper, dx_sink_fn sink, void *user, size_t cap, int whole_in_)
{ const int whole_in = whole_in_ || PLAN_HAS_IMAGE(p);    /* (an image that is there is there whole) */
  dpool     pool = { {0}, 0, ctx };
  const uint64_t n = p->x.n;
  void     *d_in = NULL, *d_rec = NULL, *d_hoff = NULL, *d_seg = NULL, *d_len = NULL, *d_out = NULL, *d_ooff = NULL;
  uint64_t *rel = NULL, i0, i1, i, most = 0;
  size_t    tmax = 0, imax = 0;
  hdr_patch h;
  int       rc = DX_OK, indexed = 0;
#define TEXT_AT(i) ((i) < n ? (size_t) p->ooff[i] - (size_t) (p->hat[(i) + 1] - p->hat[i]) : p->total)      /* where entry i's header line starts */
  h.n = n; h.ooff = p->ooff; h.hat = p->hat; h.hd = p->hd.p; h.sink = sink; h.user = user; h.base = 0;
  for (i0 = 0; i0 < n; i0 = i1)                           /* the largest slice: one allocation serves them all */
    { i1 = i0 + 1;
      while (i1 < n && TEXT_AT(i1 + 1) - TEXT_AT(i0) <= cap) i1++;
      if (TEXT_AT(i1) - TEXT_AT(i0) > tmax) tmax = TEXT_AT(i1) - TEXT_AT(i0);
      if (i1 - i0 > most) most = i1 - i0;
      if (!whole_in && p->x.rec_off[i1] - p->x.rec_off[i0] > imax) imax = (size_t) (p->x.rec_off[i1] - p->x.rec_off[i0]);
    }
  rel = malloc((most + 1) * 2 * sizeof(*rel));
  if (rel == NULL) return DX_E_NOMEM;
  TRY(dx_qv_set_coding(ctx, &p->x.coding, 0));
  if (PLAN_HAS_IMAGE(p))  d_in = p->d_in;
  else if (whole_in)      TRY(dupload(&pool, p->img, p->n, &d_in));
  else                    TRY(dalloc(&pool, imax, &d_in));
  if (PLAN_HAS_INDEX(p))
    { d_rec = p->dix.d_rec_off; d_hoff = p->dix.d_hdr_off; d_seg = p->dix.d_seg; d_len = p->dix.d_len; }
  else
    { if (whole_in) TRY(dupload(&pool, p->x.rec_off, (n + 1) * 8, &d_rec));
      else          TRY(dalloc(&pool, (most + 1) * 8, &d_rec));
      TRY(dupload(&pool, p->x.hdr_off, (n + 1) * 8, &d_hoff));
      TRY(dupload(&pool, p->x.seg, n * 5 * 4, &d_seg));
      TRY(dupload(&pool, p->x.len, n * 4, &d_len));
    }
  TRY(dalloc(&pool, (most + 1) * 8, &d_ooff));
  TRY(dalloc(&pool, tmax, &d_out));
  if (PLAN_HAS_IMAGE(p) && PLAN_HAS_INDEX(p) && p->dix.d_gidx != NULL && !p->x.flip)     /* (a slice is a contiguous part of the walk's index) */
    { TRY(dx_qv_use_dindex(ctx, d_in, &p->dix));
      indexed = 1;
    }
  for (i0 = 0; i0 < n; i0 = i1)
    { const size_t t0 = TEXT_AT(i0);
      const uint64_t *rec = d_rec;
      i1 = i0 + 1;
      while (i1 < n && TEXT_AT(i1 + 1) - t0 <= cap) i1++;
      for (i = i0; i < i1; i++) rel[i - i0] = p->ooff[i] - t0;
      TRY(dx_h2d(ctx, d_ooff, rel, (i1 - i0) * 8));
      if (whole_in)
        rec = (const uint64_t *) d_rec + i0;
      else                                                /* this slice's records, their offsets from the slice's first byte */
        { const uint64_t b0 = p->x.rec_off[i0];
          for (i = i0; i <= i1; i++) rel[most + 1 + (i - i0)] = p->x.rec_off[i] - b0;
          TRY(dx_h2d(ctx, d_in, p->img + b0, (size_t) (p->x.rec_off[i1] - b0)));
          TRY(dx_h2d(ctx, d_rec, rel + most + 1, (i1 - i0 + 1) * 8));
        }
      TRY(dx_qv_decode(ctx, d_in, rec, (const uint64_t *) d_hoff + i0, (const uint32_t *) d_seg + 5 * i0, (const uint32_t *) d_len + i0, i1 - i0,
                       (upper ? DX_DECODE_UPPER : 0) | (p->x.flip ? DX_DECODE_FLIP : 0), d_out, d_ooff));
      h.base = t0;
      TRY(dx_d2h_stream(ctx, d_out, TEXT_AT(i1) - t0, patch_and_pass, &h));
    }
#undef TEXT_AT
done:
  if (indexed) (void) dx_qv_use_index(ctx, NULL, NULL, 0, NULL, NULL, 0);
  dfree_all(&pool);
  free(rel);
  return rc;
}

int dx_file_undexqv_run(dx_ctx *ctx, const dx_undexqv_plan *p, int upper, dx_sink_fn sink, void *user)
{ dpool     pool = { {0}, 0, ctx };
  void     *d_in, *d_rec, *d_hoff, *d_seg, *d_len, *d_out, *d_ooff;
  hdr_patch h;
  int       rc = DX_OK, indexed = 0;

  if (ctx == NULL || p == NULL || sink == NULL) return DX_E_ARG;
  h.n = p->x.n; h.ooff = p->ooff; h.hat = p->hat; h.hd = p->hd.p; h.sink = sink; h.user = user; h.base = 0;
  if (p->ctx != NULL && p->ctx != ctx) return DX_E_ARG;   /* (a plan made on a device runs there) */
  if (p->x.n > 0)
    { /* does the text fit beside the image?  DEXGPU_TEXT_BUDGET (bytes) says how much text the device takes at once; else what
         is free decides: the image (unless it is there already), the index and the text, and a tenth to spare */
      const char *e = getenv("DEXGPU_TEXT_BUDGET");
      uint64_t fr = 0, all = 0;
      size_t   cap = 0;
      int      whole_in = 1;
      if (e != NULL && *e)
        { const unsigned long long v = strtoull(e, NULL, 10);
          if (v && v < p->total) cap = (size_t) (v < 65536u ? 65536u : v);
        }
      else if (dx_mem_info(ctx, &fr, &all) == DX_OK && fr > 0)
        { const double in = PLAN_HAS_IMAGE(p) ? 0.0 : (double) p->n;
          if (in + (double) p->total + 48.0 * (double) p->x.n > 0.9 * (double) fr)
            { whole_in = in <= 0.4 * (double) fr;
              cap = (size_t) ((0.9 * (double) fr - (whole_in ? in : 0.0) - 48.0 * (double) p->x.n) / (whole_in ? 1.0 : 1.4));
              if (cap < ((size_t) 4 << 20)) cap = (size_t) 4 << 20;
            }
        }
      if (cap)
        return undexqv_sliced(ctx, p, upper, sink, user, cap, dx_test_on("slice_input") && !PLAN_HAS_IMAGE(p) ? 0 : whole_in);   /* (DEXGPU_TEST=slice_input) */
    }
  if (p->x.n > 0)
    { TRY(dx_qv_set_coding(ctx, &p->x.coding, 0));
      if (PLAN_HAS_IMAGE(p)) d_in = p->d_in;              /* (the image is on the device already) */
      else                   TRY(dupload(&pool, p->img, p->n, &d_in));
      if (PLAN_HAS_INDEX(p))                              /* (and so is the index) */
        { d_rec = p->dix.d_rec_off; d_hoff = p->dix.d_hdr_off; d_seg = p->dix.d_seg; d_len = p->dix.d_len; }
      else
        { TRY(dupload(&pool, p->x.rec_off, (p->x.n + 1) * 8, &d_rec));
          TRY(dupload(&pool, p->x.hdr_off, (p->x.n + 1) * 8, &d_hoff));
          TRY(dupload(&pool, p->x.seg, p->x.n * 5 * 4, &d_seg));
          TRY(dupload(&pool, p->x.len, p->x.n * 4, &d_len));
        }
      TRY(dupload(&pool, p->ooff, p->x.n * 8, &d_ooff));
      TRY(dalloc(&pool, p->total, &d_out));
      if (p->x.gidx != NULL && !p->x.flip)               /* the walk's group index: a wavefront per line (dx_qv_use_index) */
        { void *d_gidx, *d_goff;
          TRY(dupload(&pool, p->x.gidx, (size_t) p->x.gidx_words * 4, &d_gidx));
          TRY(dupload(&pool, p->x.gidx_off, (p->x.n + 1) * 8, &d_goff));
          TRY(dx_qv_use_index(ctx, d_in, d_seg, p->x.n, d_gidx, d_goff, p->x.gidx_none));
          indexed = 1;
        }
      else if (PLAN_HAS_INDEX(p) && p->dix.d_gidx != NULL && !p->x.flip)      /* the device walk's: the run-coded lines' groups */
        { TRY(dx_qv_use_dindex(ctx, d_in, &p->dix));
          indexed = 1;
        }
      fmark("undexqv: buffers ready");
      TRY(dx_qv_decode(ctx, d_in, d_rec, d_hoff, d_seg, d_len, p->x.n,
                       (upper ? DX_DECODE_UPPER : 0) | (p->x.flip ? DX_DECODE_FLIP : 0), d_out, d_ooff));
      fmark("undexqv: decoded");
      TRY(dx_d2h_stream(ctx, d_out, p->total, patch_and_pass, &h));
      fmark("undexqv: text passed on");
    }
done:
  if (indexed) (void) dx_qv_use_index(ctx, NULL, NULL, 0, NULL, NULL, 0);     /* (the index lives in the pool freed below) */
  dfree_all(&pool);
  return rc;
}

typedef struct { uint8_t *res; } mem_sink;
static int to_memory(void *user, uint8_t *data, size_t len, size_t at)
{ memcpy(((mem_sink *) user)->res + at, data, len);
  return 0;
}

int dx_file_undexqv(dx_ctx *ctx, const uint8_t *img, size_t n, int upper, uint8_t **out, size_t *out_len)
{ dx_undexqv_plan *p = NULL;
  mem_sink m = { NULL };
  size_t   total = 0;
  int      rc;

  if (ctx == NULL || out == NULL || out_len == NULL || img == NULL) return DX_E_ARG;
  *out = NULL; *out_len = 0;
  rc = dx_file_undexqv_plan_on(ctx, img, n, &p, &total);
  if (rc != DX_OK) return rc;
  m.res = malloc(total + 16);
  if (m.res == NULL) rc = DX_E_NOMEM;
  else               rc = dx_file_undexqv_run(ctx, p, upper, to_memory, &m);
  if (rc == DX_OK) { *out = m.res; *out_len = total; }
  else             free(m.res);
  dx_file_undexqv_plan_free(p);
  return rc;
}

/* ==========================================================================================
 *  dexqv of one file on several GPUs: contiguous entry ranges, one host thread
 *  per context; the only exchange is on the host -- the merged scan state (32 bytes) and the sum
 *  of the 12 KB histograms -- after which every shard is encoded with identical tables and the
 *  record streams are concatenated in order.  No RCCL.
 * ========================================================================================== */

typedef struct shard_job shard_job;

typedef struct
  { int               nsh, lossy, rc;
    int               ok;                    /* written by shard 0 in its merge steps only, read by all after the next barrier */
    int               go;                    /* start gate: 0 wait, 1 run, -1 a thread could not be created: leave */
    pthread_mutex_t   gate_mx;
    pthread_cond_t    gate_cv;
    pthread_barrier_t bar;
    const uint8_t    *text;
    const uint64_t   *off;
    const uint32_t   *len;
    const int32_t    *hdr4;
    uint64_t          cnt, cut;              /* cut: entry at which the running symbol count reaches 100000 */
    dx_qv_params      p;
    dx_qv_coding      cd;
    uint64_t          hist[6][256], tot;
    uint8_t          *img;
    size_t            head, total;
    shard_job        *jobs;
    /* by bytes (large files): no index of the whole file exists; every shard finds and indexes its own records (shard_slice) */
    int               by_bytes, again;       /* again: something is not as it should be -- the whole file once more, the serial way */
    size_t            n, plen;
  } shard_all;

struct shard_job
  { shard_all   *all;
    dx_ctx      *ctx;
    int          id, rc;
    uint64_t     lo, hi;                      /* entries [lo, hi) */
    dx_qv_params p;
    uint64_t     hist[6][256], tot, bytes, at;
    /* by bytes: the shard's byte range as dealt, the newlines in it, where its first record begins and the line that is, its own
       index (hdr4 / len: host, the shard's entries; the offsets stay on the device) */
    size_t       p0, p1, start;
    uint64_t     nl, line0;
    int32_t     *hdr4;
    uint32_t    *len;
  };

/* Steps alternate between "every shard works and sets its own rc" and "shard 0 folds the results",
 * with a barrier after each: shard 0 reads the others' rc only in its folding steps (nobody writes
 * then) and publishes the verdict in a->ok, which the working steps read (nobody writes it then).   */
static int all_ok(shard_all *a)
{ int k;
  for (k = 0; k < a->nsh; k++)
    if (a->jobs[k].rc != DX_OK) return 0;
  return a->rc == DX_OK;
}

static int shard_slice(shard_job *j, dpool *pool, void **d_text, void **d_off, void **d_len, uint64_t *span);

static void *shard_main(void *arg)
{ shard_job  *j = arg;
  shard_all  *a = j->all;
  dpool       pool = { {0}, 0, j->ctx };
  uint64_t    m = j->hi - j->lo, i, *roff = NULL, *hoff = NULL, base = 0, span = 0, total = 0;
  uint8_t    *blob = NULL;
  void       *d_text = NULL, *d_off = NULL, *d_len = NULL, *d_hdr = NULL, *d_hoff = NULL, *d_rec = NULL, *d_seg = NULL, *d_out = NULL;
  dx_qv_batch b;
  int         rc = DX_OK, k;

  pthread_mutex_lock(&a->gate_mx);                        /* all threads exist, or none runs */
  while (a->go == 0) pthread_cond_wait(&a->gate_cv, &a->gate_mx);
  k = a->go;
  pthread_mutex_unlock(&a->gate_mx);
  if (k < 0) return NULL;

  memset(&b, 0, sizeof(b));
  j->p.delChar = j->p.subChar = -1; j->p.del_first = j->p.sub_first = -1;
  memset(j->hist, 0, sizeof(j->hist)); j->tot = 0; j->bytes = 0;

  if (a->by_bytes)
    { rc = shard_slice(j, &pool, &d_text, &d_off, &d_len, &span);        /* (five barriers inside, whatever becomes of it) */
      m = j->hi - j->lo;
      if (rc == DX_OK && m > 0)
        { int32_t lwell = j->id ? a->jobs[j->id - 1].hdr4[4 * (a->jobs[j->id - 1].hi - a->jobs[j->id - 1].lo - 1)] : 0;
          hoff = malloc((m + 1) * sizeof(*hoff));
          blob = malloc(dx_frame_bound(j->hdr4, m, lwell, 0) + 16);
          if (!hoff || !blob) rc = DX_E_NOMEM;
          if (rc == DX_OK) rc = dx_frame_headers(j->hdr4, NULL, m, 0, &lwell, blob, hoff);
          if (rc == DX_OK) rc = dupload(&pool, blob, (size_t) hoff[m], &d_hdr);
          if (rc == DX_OK) rc = dupload(&pool, hoff, (m + 1) * 8, &d_hoff);
          if (rc == DX_OK) rc = dalloc(&pool, (m + 1) * 8, &d_rec);
          if (rc == DX_OK) rc = dalloc(&pool, m * 20, &d_seg);
          b.d_text = d_text; b.d_off = d_off; b.d_len = d_len; b.n = m; b.line_pad = 1; b.text_bytes = span;
          if (rc == DX_OK) rc = dx_qv_prescan(j->ctx, &b, j->lo, &j->p);
        }
    }
  else if (m > 0)                                        /* this shard's slice of the text image */
    { int32_t lwell = j->lo ? a->hdr4[4*(j->lo-1)] : 0;
      base = a->off[j->lo];
      span = a->off[j->hi-1] + 5 * ((uint64_t) a->len[j->hi-1] + 1) - base;
      roff = malloc(m * sizeof(*roff));
      hoff = malloc((m + 1) * sizeof(*hoff));
      blob = malloc(dx_frame_bound(a->hdr4 + 4*j->lo, m, lwell, 0) + 16);
      if (!roff || !hoff || !blob) rc = DX_E_NOMEM;
      for (i = 0; rc == DX_OK && i < m; i++) roff[i] = a->off[j->lo + i] - base;
      if (rc == DX_OK) rc = dx_frame_headers(a->hdr4 + 4*j->lo, NULL, m, 0, &lwell, blob, hoff);
      if (rc == DX_OK) rc = dupload(&pool, a->text + base, span, &d_text);
      if (rc == DX_OK) rc = dupload(&pool, roff, m * 8, &d_off);
      if (rc == DX_OK) rc = dupload(&pool, a->len + j->lo, m * 4, &d_len);
      if (rc == DX_OK) rc = dupload(&pool, blob, (size_t) hoff[m], &d_hdr);
      if (rc == DX_OK) rc = dupload(&pool, hoff, (m + 1) * 8, &d_hoff);
      if (rc == DX_OK) rc = dalloc(&pool, (m + 1) * 8, &d_rec);
      if (rc == DX_OK) rc = dalloc(&pool, m * 20, &d_seg);
      b.d_text = d_text; b.d_off = d_off; b.d_len = d_len; b.n = m; b.line_pad = 1; b.text_bytes = span;
      if (rc == DX_OK) rc = dx_qv_prescan(j->ctx, &b, j->lo, &j->p);       /* QV.c:993-1015, per shard */
    }
  if (rc == DX_OK && j->id == 0 && a->cut >= j->hi && !a->by_bytes)      /* (by bytes: shard_slice has seen to it that this is not so) */
    { /* the file's first 100000 symbols (QV.c:1006-1015) reach beyond shard 0: find the provisional
         subChar on a prefix batch of entries [0, cut] instead */
      uint64_t     mp = a->cut + 1, sp = a->off[a->cut] + 5 * ((uint64_t) a->len[a->cut] + 1) - a->off[0];
      uint64_t    *po = malloc(mp * sizeof(*po));
      void        *pt = NULL, *pd_off = NULL, *pd_len = NULL;
      dx_qv_batch  pb;
      dx_qv_params pp = { 0, -1, 0, -1 };                 /* delChar "set": only the sub search runs */
      if (po == NULL) rc = DX_E_NOMEM;
      for (i = 0; rc == DX_OK && i < mp; i++) po[i] = a->off[i] - a->off[0];
      if (rc == DX_OK) rc = dupload(&pool, a->text + a->off[0], sp, &pt);
      if (rc == DX_OK) rc = dupload(&pool, po, mp * 8, &pd_off);
      if (rc == DX_OK) rc = dupload(&pool, a->len, mp * 4, &pd_len);
      pb.d_text = pt; pb.d_off = pd_off; pb.d_len = pd_len; pb.n = mp; pb.line_pad = 1; pb.text_bytes = sp;
      if (rc == DX_OK) rc = dx_qv_prescan(j->ctx, &pb, 0, &pp);
      j->p.subChar = pp.subChar; j->p.sub_first = pp.sub_first;
      free(po);
    }
  j->rc = rc;
  pthread_barrier_wait(&a->bar);

  if (j->id == 0 && (a->ok = all_ok(a)))                 /* merge the scan state (lowest entry wins) */
    { a->p.delChar = a->p.subChar = -1; a->p.del_first = a->p.sub_first = -1;
      for (k = 0; k < a->nsh; k++)
        if (a->jobs[k].p.delChar >= 0 && (a->p.delChar < 0 || a->jobs[k].p.del_first < a->p.del_first))
          { a->p.delChar = a->jobs[k].p.delChar; a->p.del_first = a->jobs[k].p.del_first; }
      for (k = 0; k < a->nsh; k++)
        if (a->jobs[k].lo == 0 && a->jobs[k].hi > 0)
          { a->p.subChar = a->jobs[k].p.subChar; a->p.sub_first = a->jobs[k].p.sub_first; }
    }
  pthread_barrier_wait(&a->bar);

  if (a->ok && m > 0)
    j->rc = dx_qv_hist(j->ctx, &b, j->lo, &a->p, j->hist, &j->tot);        /* QV.c:988-1017, per shard */
  pthread_barrier_wait(&a->bar);

  if (j->id == 0 && (a->ok = all_ok(a)))                 /* host-side sum + Create_QVcoding */
    { int s, x;
      memset(a->hist, 0, sizeof(a->hist)); a->tot = 0;
      for (k = 0; k < a->nsh; k++)
        { for (s = 0; s < 6; s++)
            for (x = 0; x < 256; x++)
              a->hist[s][x] += a->jobs[k].hist[s][x];
          a->tot += a->jobs[k].tot;
        }
      a->rc = dx_qv_build((const uint64_t (*)[256]) a->hist, a->tot, &a->p, a->lossy, &a->cd);
      a->ok = a->rc == DX_OK;
    }
  pthread_barrier_wait(&a->bar);

  if (a->ok && m > 0)                                    /* Compress_Next_QVentry for the shard's entries */
    { rc = dx_qv_set_coding(j->ctx, &a->cd, a->lossy);
      if (rc == DX_OK && !two_pass())
        { const uint64_t cap = hoff[m] + dx_qv_out_bound((const uint64_t (*)[256]) j->hist, m, &a->cd, a->lossy);
          rc = dalloc(&pool, cap, &d_out);
          if (rc == DX_OK) rc = dx_qv_encode_onepass(j->ctx, &b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap, &total);
        }
      else if (rc == DX_OK)
        { rc = dx_qv_sizes(j->ctx, &b, d_hoff, d_seg, d_rec, &total);
          if (rc == DX_OK) rc = dalloc(&pool, total, &d_out);
          if (rc == DX_OK) rc = dx_qv_encode(j->ctx, &b, d_hdr, d_hoff, d_rec, d_seg, d_out);
        }
      j->bytes = total;
      j->rc = rc;
    }
  pthread_barrier_wait(&a->bar);

  if (j->id == 0 && (a->ok = all_ok(a)))                 /* layout of the final image */
    { size_t clen = 0, plen = 0;
      if (a->by_bytes) plen = a->plen;
      else
        { const uint8_t *h = a->text, *slash = memchr(h + 1, '/', (size_t) (a->off[0] - 1));
          plen = slash ? (size_t) (slash - h) : 0;
        }
      dx_qv_write_coding(&a->cd, (const char *) a->text, plen, NULL, 0, &clen);
      a->head = 2 + clen;
      a->total = a->head;
      for (k = 0; k < a->nsh; k++)
        { a->jobs[k].at = a->total;
          a->total += a->jobs[k].bytes;
        }
      a->img = malloc(a->total + 16);
      if (a->img == NULL) a->rc = DX_E_NOMEM;
      else
        { uint16_t key = 0x55aa;
          memcpy(a->img, &key, 2);
          a->rc = dx_qv_write_coding(&a->cd, (const char *) a->text, plen, a->img + 2, clen, &clen);
        }
      a->ok = a->rc == DX_OK;
    }
  pthread_barrier_wait(&a->bar);

  if (a->ok && m > 0)
    j->rc = dx_d2h(j->ctx, a->img + j->at, d_out, total);
  dfree_all(&pool);
  free(roff); free(hoff); free(blob);
  return NULL;
}

/* A file too large to be indexed by one thread first (a terabyte over eight GPUs): the bytes are dealt evenly, and
 * every shard finds the records that BEGIN in its range -- a record is six lines (QV.c:948-978), so all it needs of the others is
 * how many newlines stand in front of its range --, uploads exactly those and has its own device index them (dx_index_quiva_device:
 * structure checks and all).  Anything out of the ordinary (a line count that is no multiple of six, an indexer that says no, the
 * first 100000 symbols reaching beyond shard 0) sets a->again: dx_file_dexqv_sharded then does the file the serial way, which also has
 * the reference's words for a malformed file.  Every thread passes the same five barriers.                                     */
static int shard_slice(shard_job *j, dpool *pool, void **d_text, void **d_off, void **d_len, uint64_t *span)
{ shard_all *a = j->all;
  int rc = DX_OK, k;
  { const uint8_t *q = a->text + j->p0, *e = a->text + j->p1;            /* 1: the newlines of the range as dealt */
    uint64_t c = 0;
    while (q < e && (q = memchr(q, '\n', (size_t) (e - q))) != NULL) { c += 1; q += 1; }
    j->nl = c;
  }
  pthread_barrier_wait(&a->bar);
  if (j->id == 0)                                        /* 2: the lines in front of every range; six lines a record, the last one whole */
    { uint64_t before = 0;
      for (k = 0; k < a->nsh; k++) { a->jobs[k].line0 = before; before += a->jobs[k].nl; }
      if (before % 6 != 0 || before == 0 || a->text[a->n - 1] != '\n') a->again = 1;
      a->cnt = before / 6;
    }
  pthread_barrier_wait(&a->bar);
  if (!a->again)                                         /* 3: the first record that begins in the range */
    { const uint8_t *q = a->text + j->p0, *e = a->text + a->n;
      uint64_t line = j->line0;                            /* (the line p0 stands in) */
      if (j->p0 > 0 && q[-1] != '\n')                       /* ... which began in front of the range: the next one */
        { q = memchr(q, '\n', (size_t) (e - q)); q = q ? q + 1 : e; line += 1; }
      while (line % 6 != 0 && q < e)
        { q = memchr(q, '\n', (size_t) (e - q)); q = q ? q + 1 : e; line += 1; }
      j->start = (size_t) (q - a->text);
      j->lo = line / 6;
    }
  pthread_barrier_wait(&a->bar);
  if (!a->again)                                         /* 4: the shard's records, to its device, indexed there */
    { const size_t end = j->id + 1 < a->nsh ? a->jobs[j->id + 1].start : a->n;
      uint64_t cnt = 0, el = 0;
      int      ec = 0;
      j->hi = j->id + 1 < a->nsh ? a->jobs[j->id + 1].lo : a->cnt;
      *span = end - j->start;
      if (j->hi > j->lo)
        { uint64_t *go = NULL; uint32_t *gl = NULL;
          size_t plen = 0;
          rc = dupload(pool, a->text + j->start, (size_t) *span, d_text);
          if (rc == DX_OK) rc = dx_index_quiva_device(j->ctx, *d_text, *span, &go, &gl, &cnt, &j->hdr4, &plen, &el, &ec);
          if (rc == DX_OK && cnt > 0) { pool->p[pool->n++] = go; pool->p[pool->n++] = gl; *d_off = go; *d_len = gl; }
          if (rc == DX_OK && cnt != j->hi - j->lo) rc = DX_E_FORMAT;
          if (rc == DX_OK && (j->len = malloc((size_t) cnt * 4)) == NULL) rc = DX_E_NOMEM;
          if (rc == DX_OK) rc = dx_d2h(j->ctx, j->len, gl, (size_t) cnt * 4);
          if (j->id == 0) a->plen = plen;
        }
      else if (j->hi < j->lo) rc = DX_E_FORMAT;
      if (rc != DX_OK) { j->rc = rc; }
    }
  pthread_barrier_wait(&a->bar);
  if (j->id == 0 && !a->again)                           /* the entry at which the running symbol count reaches 100000 (QV.c:1006-1015) */
    { uint64_t run = 0, e2 = 0, m0 = a->jobs[0].hi - a->jobs[0].lo;
      for (k = 0; k < a->nsh; k++) if (a->jobs[k].rc != DX_OK || a->jobs[k].hi <= a->jobs[k].lo) a->again = 1;
      for (e2 = 0; !a->again && e2 < m0; e2++)
        { run += a->jobs[0].len[e2];
          if (run >= 100000) break;
        }
      if (!a->again && e2 >= m0) a->again = 1;             /* (not within shard 0: a small file, the serial way knows what to do) */
      a->cut = e2;
    }
  pthread_barrier_wait(&a->bar);
  if (a->again) { j->hi = j->lo; return DX_E_FORMAT; }
  return rc;
}

#define DX_SHARD_BYTES_MIN ((size_t) 64 << 20)           /* per shard: from here on the shards index their own byte ranges */
int dx_file_dexqv_sharded(dx_ctx **ctxs, int nctx, const uint8_t *text, size_t n, int lossy,
                          uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ shard_all  a;
  pthread_t *th = NULL;
  uint64_t   cnt = 0, *off = NULL;
  uint32_t  *len = NULL;
  int32_t   *hdr4 = NULL;
  size_t     plen = 0;
  int        rc, k, started = 0, by_bytes;

  if (ctxs == NULL || nctx < 1 || out == NULL || out_len == NULL) return DX_E_ARG;
  if (nctx == 1) return dx_file_dexqv(ctxs[0], text, n, lossy, out, out_len, errline, errcode);
  *out = NULL; *out_len = 0;
  { const size_t least = (size_t) dx_test_num("shard_bytes_min", (long long) DX_SHARD_BYTES_MIN);     /* (tests: the by-bytes way on small files) */
    by_bytes = n / (size_t) nctx >= least && n / (size_t) nctx >= 4096 && !dx_test_on("host_index");
  }
again:
  memset(&a, 0, sizeof(a));
  started = 0;
  a.jobs = calloc((size_t) nctx, sizeof(*a.jobs));
  th = calloc((size_t) nctx, sizeof(*th));
  if (!a.jobs || !th) { rc = DX_E_NOMEM; goto done; }
  a.nsh = nctx; a.lossy = lossy; a.text = text; a.n = n; a.rc = DX_OK; a.by_bytes = by_bytes;

  if (!by_bytes)                                          /* the whole file indexed here first (small files; what the shards turn down) */
    { rc = dx_index_quiva(text, n, 0, NULL, NULL, NULL, &cnt, &plen, errline, errcode);
      if (rc != DX_OK) goto done;
      if (cnt == 0) { rc = DX_E_DEGENERATE; goto done; }
      off  = malloc((cnt + 1) * sizeof(*off));
      len  = malloc((cnt + 1) * sizeof(*len));
      hdr4 = malloc((cnt + 1) * 4 * sizeof(*hdr4));
      if (!off || !len || !hdr4) { rc = DX_E_NOMEM; goto done; }
      TRY(dx_index_quiva(text, n, cnt, off, len, hdr4, &cnt, &plen, errline, errcode));
      a.off = off; a.len = len; a.hdr4 = hdr4; a.cnt = cnt;
      { uint64_t run = 0, e;
        a.cut = 0;
        for (e = 0; e < cnt; e++)
          { run += len[e];
            if (run >= 100000) break;
          }
        a.cut = e < cnt ? e : 0;             /* never reached: no subChar at all, shard 0 finds that too */
      }
    }
  pthread_barrier_init(&a.bar, NULL, (unsigned) nctx);
  pthread_mutex_init(&a.gate_mx, NULL);
  pthread_cond_init(&a.gate_cv, NULL);
  a.go = 0; a.ok = 1;
  { uint64_t per = cnt / (uint64_t) nctx, extra = cnt % (uint64_t) nctx, lo = 0;
    for (k = 0; k < nctx; k++)
      { uint64_t m = per + ((uint64_t) k < extra ? 1 : 0);
        a.jobs[k].all = &a; a.jobs[k].ctx = ctxs[k]; a.jobs[k].id = k;
        a.jobs[k].lo = lo; a.jobs[k].hi = lo + m; a.jobs[k].rc = DX_OK;
        lo += m;
        a.jobs[k].p0 = (size_t) ((unsigned __int128) n * (unsigned) k / (unsigned) nctx);        /* (by bytes: the range as dealt) */
        a.jobs[k].p1 = (size_t) ((unsigned __int128) n * (unsigned) (k + 1) / (unsigned) nctx);
      }
  }
  for (k = 0; k < nctx; k++)                              /* the barriers count nctx threads: all of them or none */
    { if (pthread_create(&th[k], NULL, shard_main, &a.jobs[k]) != 0) break;
      started += 1;
    }
  pthread_mutex_lock(&a.gate_mx);
  a.go = started == nctx ? 1 : -1;
  pthread_cond_broadcast(&a.gate_cv);
  pthread_mutex_unlock(&a.gate_mx);
  for (k = 0; k < started; k++)
    pthread_join(th[k], NULL);
  if (started < nctx)
    rc = DX_E_NOMEM;
  else
    { rc = a.rc;
      for (k = 0; k < nctx && rc == DX_OK; k++)
        rc = a.jobs[k].rc;
    }
  if (rc == DX_OK && !a.again)
    { *out = a.img; *out_len = a.total; a.img = NULL; }

  pthread_barrier_destroy(&a.bar);
  pthread_mutex_destroy(&a.gate_mx);
  pthread_cond_destroy(&a.gate_cv);
done:
  for (k = 0; a.jobs != NULL && k < nctx; k++) { free(a.jobs[k].hdr4); free(a.jobs[k].len); }
  free(off); free(len); free(hdr4); free(a.jobs); free(th); free(a.img);
  off = NULL; len = NULL; hdr4 = NULL; th = NULL;
  if (by_bytes && a.again && started == nctx)            /* the shards turned the file down: the serial way (and its words for what is wrong) */
    { by_bytes = 0;
      goto again;
    }
  return rc;
}

/* ==========================================================================================
 *  in-memory entry API: the shape of QVcoding_Scan1 /
 *  Compress_Next_QVentry1 (QV.c:866-920, 1343-1379), i.e. what dex2DB.c:511-643 calls per entry
 *  to write a .qvs track -- as a batch: entries are gathered on the host, then scanned and
 *  compressed together on the GPU.  The output is the bare record stream (no framing bytes) and
 *  the offset of every entry in it (DAZZ_READ.coff, dex2DB.c:617-621).
 * ========================================================================================== */
struct dx_entries
  { uint8_t  *text;  size_t tlen, tcap;      /* five lines back to back per entry (line_pad 0) */
    uint64_t *off;   uint32_t *len;
    uint64_t  n, cap;
  };

dx_entries *dx_entries_new(void) { return calloc(1, sizeof(dx_entries)); }

void dx_entries_free(dx_entries *e)
{ if (e == NULL) return;
  free(e->text); free(e->off); free(e->len); free(e);
}

/* QVcoding_Scan1's / Compress_Next_QVentry1's argument list: one entry, five streams of rlen bytes */
int dx_entries_add(dx_entries *e, int rlen, const char *del, const char *tag, const char *ins,
                   const char *mrg, const char *sub)
{ const char *s[5];
  int k;
  if (e == NULL || rlen < 0 || (rlen > 0 && (!del || !tag || !ins || !mrg || !sub))) return DX_E_ARG;
  s[0] = del; s[1] = tag; s[2] = ins; s[3] = mrg; s[4] = sub;
  if (e->n == e->cap)
    { uint64_t nc = e->cap ? 2 * e->cap : 1024;
      uint64_t *no = realloc(e->off, nc * sizeof(*no));
      uint32_t *nl = realloc(e->len, nc * sizeof(*nl));
      if (no) e->off = no;
      if (nl) e->len = nl;
      if (!no || !nl) return DX_E_NOMEM;
      e->cap = nc;
    }
  if (e->tlen + 5 * (size_t) rlen + 16 > e->tcap)
    { size_t nc = 2 * e->tcap + 5 * (size_t) rlen + 4096;
      uint8_t *nt = realloc(e->text, nc);
      if (!nt) return DX_E_NOMEM;
      e->text = nt; e->tcap = nc;
    }
  e->off[e->n] = e->tlen;
  e->len[e->n] = (uint32_t) rlen;
  for (k = 0; k < 5; k++)
    { memcpy(e->text + e->tlen, s[k], (size_t) rlen);
      e->tlen += (size_t) rlen;
    }
  e->n += 1;
  return DX_OK;
}

int dx_entries_compress(dx_ctx *ctx, const dx_entries *e, int lossy, dx_qv_coding *coding,
                        uint8_t **records, size_t *nbytes, uint64_t **coff)
{ dpool        pool = { {0}, 0, ctx };
  dx_qv_batch  b;
  dx_qv_params p = { -1, -1, -1, -1 };
  uint64_t   (*hist)[256] = NULL, tot = 0, total = 0;
  void        *d_text, *d_off, *d_len, *d_rec, *d_seg, *d_out;
  uint8_t     *res = NULL;
  uint64_t    *ro = NULL;
  int          rc;

  if (ctx == NULL || e == NULL || coding == NULL || records == NULL || nbytes == NULL) return DX_E_ARG;
  *records = NULL; *nbytes = 0;
  if (coff) *coff = NULL;
  if (e->n == 0) return DX_E_DEGENERATE;
  hist = calloc(6, sizeof(*hist));
  if (!hist) return DX_E_NOMEM;
  TRY(dupload(&pool, e->text, e->tlen, &d_text));
  TRY(dupload(&pool, e->off, e->n * 8, &d_off));
  TRY(dupload(&pool, e->len, e->n * 4, &d_len));
  TRY(dalloc(&pool, (e->n + 1) * 8, &d_rec));
  TRY(dalloc(&pool, e->n * 20, &d_seg));
  b.d_text = d_text; b.d_off = d_off; b.d_len = d_len; b.n = e->n; b.line_pad = 0; b.text_bytes = e->tlen;
  TRY(dx_qv_scan(ctx, &b, 0, &p, hist, &tot));             /* QVcoding_Scan1 over all entries */
  TRY(dx_qv_build((const uint64_t (*)[256]) hist, tot, &p, lossy, coding));   /* Create_QVcoding */
  TRY(dx_qv_set_coding(ctx, coding, lossy));
  if (!two_pass())                                              /* Compress_Next_QVentry1 x n */
    { const uint64_t cap = dx_qv_out_bound((const uint64_t (*)[256]) hist, e->n, coding, lossy);
      TRY(dalloc(&pool, cap, &d_out));
      TRY(dx_qv_encode_onepass(ctx, &b, NULL, NULL, d_seg, d_rec, d_out, cap, &total));
    }
  else
    { TRY(dx_qv_sizes(ctx, &b, NULL, d_seg, d_rec, &total));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_qv_encode(ctx, &b, NULL, NULL, d_rec, d_seg, d_out));
    }
  res = malloc(total + 16);
  ro  = malloc((e->n + 1) * sizeof(*ro));
  if (!res || !ro) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_d2h(ctx, res, d_out, total));
  TRY(dx_d2h(ctx, ro, d_rec, (e->n + 1) * 8));
  *records = res; *nbytes = total; res = NULL;
  if (coff) { *coff = ro; ro = NULL; }
  rc = DX_OK;

done:
  dfree_all(&pool);
  free(hist); free(res); free(ro);
  return rc;
}
