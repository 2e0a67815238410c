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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dexgpu.h"

void dx_file_free(void *p) { free(p); }

#define TRY(x) do { rc = (x); if (rc != DX_OK) goto done; } while (0)

#define DX_GPU_INDEX_MIN (1u << 20)      /* .quiva images from 1 MiB on are indexed on the GPU */

typedef struct { void *p[16]; int n; dx_ctx *ctx; } dpool;

static int dalloc(dpool *pool, size_t bytes, void **out)
{ int rc = dx_malloc(pool->ctx, bytes + 64, out);
  if (rc == DX_OK) pool->p[pool->n++] = *out;
  return rc;
}

static int dupload(dpool *pool, const void *src, size_t bytes, void **out)
{ int rc = dalloc(pool, bytes, out);
  if (rc == DX_OK && bytes) rc = dx_h2d(pool->ctx, *out, src, bytes);
  return rc;
}

static void dfree_all(dpool *pool)
{ int i;
  for (i = 0; i < pool->n; i++)
    dx_free(pool->ctx, pool->p[i]);
  pool->n = 0;
}

/* ==========================================================================================
 *  dexta / dexar
 * ========================================================================================== */
int dx_file_pack2(dx_ctx *ctx, int arrow, const uint8_t *text, size_t n,
                  uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ dpool     pool = { {0}, 0, ctx };
  uint64_t  cnt = 0, i, *off = NULL, *hoff = NULL, *ooff = NULL;
  uint32_t *tlen = NULL, *nsym = NULL;
  int32_t  *hdr4 = NULL, lwell = 0;
  uint16_t *cnr4 = NULL;
  uint8_t  *blob = NULL, *img = NULL;
  size_t    plen = 0, at, total;
  void     *d_text, *d_off, *d_tlen, *d_nsym, *d_hdr, *d_hoff, *d_out, *d_ooff;
  int       rc;

  if (ctx == NULL || out == NULL || out_len == NULL) return DX_E_ARG;
  *out = NULL; *out_len = 0;

  TRY(dx_index_seq(arrow, text, n, 0, NULL, NULL, NULL, NULL, NULL, &cnt, &plen, errline, errcode));
  off  = malloc((cnt + 1) * sizeof(*off));
  hoff = malloc((cnt + 1) * sizeof(*hoff));
  ooff = malloc((cnt + 1) * sizeof(*ooff));
  tlen = malloc((cnt + 1) * sizeof(*tlen));
  nsym = malloc((cnt + 1) * sizeof(*nsym));
  hdr4 = malloc((cnt + 1) * 4 * sizeof(*hdr4));
  cnr4 = malloc((cnt + 1) * 4 * sizeof(*cnr4));
  if (!off || !hoff || !ooff || !tlen || !nsym || !hdr4 || !cnr4) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_index_seq(arrow, text, n, cnt, off, tlen, nsym, hdr4, cnr4, &cnt, &plen, errline, errcode));

  blob = malloc(dx_frame_bound(hdr4, cnt, 0, arrow) + 16);
  if (!blob) { rc = DX_E_NOMEM; goto done; }
  TRY(dx_frame_headers(hdr4, cnr4, cnt, arrow, &lwell, blob, hoff));

  at = 2 + 4 + plen;                                   /* key, prefix length, prefix: dexta.c:124-129 */
  for (i = 0; i < cnt; i++)
    { ooff[i] = at;
      at += (size_t) (hoff[i+1] - hoff[i]) + (((size_t) nsym[i] + 3) >> 2);
    }
  total = at;

  img = malloc(total + 16);
  if (!img) { rc = DX_E_NOMEM; goto done; }
  { uint16_t key = 0x55aa;
    int32_t  pl  = (int32_t) plen;
    memcpy(img, &key, 2);
    memcpy(img + 2, &pl, 4);
    memcpy(img + 6, text, plen);
  }

  if (cnt > 0)
    { TRY(dupload(&pool, text, n, &d_text));
      TRY(dupload(&pool, off,  cnt * 8, &d_off));
      TRY(dupload(&pool, tlen, cnt * 4, &d_tlen));
      TRY(dupload(&pool, nsym, cnt * 4, &d_nsym));
      TRY(dupload(&pool, blob, (size_t) hoff[cnt], &d_hdr));
      TRY(dupload(&pool, hoff, (cnt + 1) * 8, &d_hoff));
      TRY(dupload(&pool, ooff, cnt * 8, &d_ooff));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_pack2_encode(ctx, arrow ? DX_ALPHA_ARROW : DX_ALPHA_BASES, d_text, d_off, d_tlen, d_nsym, cnt,
                          d_hdr, d_hoff, d_out, d_ooff));
      TRY(dx_d2h(ctx, img + ooff[0], (uint8_t *) d_out + ooff[0], total - (size_t) ooff[0]));
    }
  *out = img; *out_len = total; img = NULL;
  rc = DX_OK;

done:
  dfree_all(&pool);
  free(off); free(hoff); free(ooff); free(tlen); free(nsym); free(hdr4); free(cnr4); free(blob); free(img);
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

/* mode: DX_LETTERS_LOWER / _UPPER (dexta images) or _ARROW (dexar images) */
int dx_file_unpack2(dx_ctx *ctx, int mode, const uint8_t *img, size_t n, uint32_t width,
                    uint8_t **out, size_t *out_len)
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

  if (ctx == NULL || out == NULL || out_len == NULL || img == NULL) return DX_E_ARG;
  if (width == 0) return DX_E_ARG;
  *out = NULL; *out_len = 0;

  rd(&r, &key, 2);                                        /* undexta.c:138-159, undexar.c:136-145 */
  if (r.bad) return DX_E_FORMAT;
  if (key == 0x55aa)               { flip = 0; newv = 1; }
  else if (key == 0xaa55)          { flip = 1; newv = 1; }
  else if (!arrow && key == 0x33cc) { flip = 0; newv = 0; }
  else if (!arrow && key == 0xcc33) { flip = 1; newv = 0; }
  else return DX_E_FORMAT;

  plen = rd_i32(&r, flip);                                /* undexta.c:161-169 */
  if (r.bad || plen < 0 || (size_t) plen > n) return DX_E_FORMAT;
  name = malloc((size_t) plen + 1);
  if (!name) return DX_E_NOMEM;
  rd(&r, name, (size_t) plen);
  name[plen] = '\0';

  while (r.at < r.n)                                      /* undexta.c:175-271: walk the records */
    { uint8_t  byte;
      int      beg, end, qv = 0, k;
      uint16_t cnr[4] = { 0, 0, 0, 0 };
      size_t   clen;

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
      if (r.bad || end < beg) { rc = DX_E_FORMAT; goto done; }
      clen = ((size_t) (end - beg) + 3) >> 2;
      if (r.at + clen > r.n) { rc = DX_E_FORMAT; goto done; }

      if (cnt == cap)
        { cap  = cap ? 2 * cap : 1024;
          ioff = realloc(ioff, cap * sizeof(*ioff));
          ooff = realloc(ooff, cap * sizeof(*ooff));
          hat  = realloc(hat,  (cap + 1) * sizeof(*hat));
          nsym = realloc(nsym, cap * sizeof(*nsym));
          if (!ioff || !ooff || !hat || !nsym) { rc = DX_E_NOMEM; goto done; }
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
      nsym[cnt] = (uint32_t) (end - beg);
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
  res = malloc(total + 16);
  if (!res) { rc = DX_E_NOMEM; goto done; }

  if (cnt > 0)
    { TRY(dupload(&pool, img, n, &d_in));
      TRY(dupload(&pool, ioff, cnt * 8, &d_ioff));
      TRY(dupload(&pool, nsym, cnt * 4, &d_nsym));
      TRY(dupload(&pool, ooff, cnt * 8, &d_ooff));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_pack2_decode(ctx, mode, d_in, d_ioff, d_nsym, cnt, width, d_out, d_ooff));
      TRY(dx_d2h(ctx, res, d_out, total));
      for (i = 0; i < cnt; i++)
        memcpy(res + ooff[i] - (hat[i+1] - hat[i]), hd.p + hat[i], (size_t) (hat[i+1] - hat[i]));
    }
  *out = res; *out_len = total; res = NULL;
  rc = DX_OK;

done:
  dfree_all(&pool);
  free(name); free(hd.p); free(ioff); free(ooff); free(hat); free(nsym); free(res);
  return rc;
}

/* ==========================================================================================
 *  dexqv
 * ========================================================================================== */
int dx_file_dexqv(dx_ctx *ctx, const uint8_t *text, size_t n, int lossy,
                  uint8_t **out, size_t *out_len, uint64_t *errline, int *errcode)
{ dpool        pool = { {0}, 0, ctx };
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

  if (ctx == NULL || out == NULL || out_len == NULL) return DX_E_ARG;
  *out = NULL; *out_len = 0;

  /* pass 1 of the reference (QVcoding_Scan, dexqv.c:81-82): validate + index.  Large images are
   * indexed on the GPU (newline scan + structure checks there, only the header lines come back);
   * small ones, and any image the GPU front end rejects (so that the message is exactly the
   * reference's first one), by the host indexer.                                               */
  cd   = malloc(sizeof(*cd));
  hist = calloc(6, sizeof(*hist));
  if (!cd || !hist) { rc = DX_E_NOMEM; goto done; }
  TRY(dupload(&pool, text, n, &d_text));
  if (n >= DX_GPU_INDEX_MIN && getenv("DEXGPU_HOST_INDEX") == NULL)
    { uint64_t *go = NULL; uint32_t *gl = NULL;
      rc = dx_index_quiva_device(ctx, d_text, n, &go, &gl, &cnt, &hdr4, &plen, errline, errcode);
      if (rc == DX_OK && cnt > 0)
        { d_off = go; d_len = gl;
          pool.p[pool.n++] = go; pool.p[pool.n++] = gl;
        }
      else if (rc != DX_OK && rc != DX_E_FORMAT)
        goto done;
      rc = DX_OK;
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
  TRY(dx_qv_prescan(ctx, &b, 0, &p));
  TRY(dx_qv_hist(ctx, &b, 0, &p, hist, &tot));
  TRY(dx_qv_build((const uint64_t (*)[256]) hist, tot, &p, lossy, cd));   /* Create_QVcoding, dexqv.c:86 */
  TRY(dx_qv_set_coding(ctx, cd, lossy));

  rc = dx_qv_write_coding(cd, (const char *) text, plen, NULL, 0, &clen);  /* size of Write_QVcoding */
  if (rc != DX_OK && rc != DX_E_SPACE) goto done;
  head = 2 + clen;

  TRY(dx_qv_sizes(ctx, &b, d_hoff, d_seg, d_rec, &total));
  img = malloc(head + total + 16);
  if (!img) { rc = DX_E_NOMEM; goto done; }
  { uint16_t key = 0x55aa;                                                 /* dexqv.c:105-108 */
    memcpy(img, &key, 2);
    TRY(dx_qv_write_coding(cd, (const char *) text, plen, img + 2, clen, &clen));
  }
  TRY(dalloc(&pool, total, &d_out));
  TRY(dx_qv_encode(ctx, &b, d_hdr, d_hoff, d_rec, d_seg, d_out));          /* pass 2, dexqv.c:112-143 */
  TRY(dx_d2h(ctx, img + head, d_out, total));
  *out = img; *out_len = head + total; img = NULL;
  rc = DX_OK;

done:
  dfree_all(&pool);
  free(off); free(hoff); free(len); free(hdr4); free(blob); free(cd); free(hist); free(img);
  return rc;
}

/* ==========================================================================================
 *  undexqv
 * ========================================================================================== */
int dx_file_undexqv(dx_ctx *ctx, const uint8_t *img, size_t n, int upper, uint8_t **out, size_t *out_len)
{ dpool       pool = { {0}, 0, ctx };
  dx_qv_index x;
  tbuf        hd = { NULL, 0, 0 };
  uint64_t   *ooff = NULL, *hat = NULL, i;
  uint8_t    *res = NULL;
  size_t      total = 0, plen;
  void       *d_in, *d_rec, *d_hoff, *d_seg, *d_len, *d_out, *d_ooff;
  int         rc;

  if (ctx == NULL || out == NULL || out_len == NULL || img == NULL) return DX_E_ARG;
  *out = NULL; *out_len = 0;
  rc = dx_qv_walk(img, n, &x);                            /* sequential boundary walk (host) */
  if (rc != DX_OK) return rc;
  plen = strlen(x.prefix);

  ooff = malloc((x.n + 1) * sizeof(*ooff));
  hat  = malloc((x.n + 1) * sizeof(*hat));
  if (!ooff || !hat) { rc = DX_E_NOMEM; goto done; }
  for (i = 0; i < x.n; i++)                               /* header lines, undexqv.c:182 */
    { const int32_t *h = x.hdr4 + 4*i;
      if ((rc = tb_room(&hd, plen + 80)) != DX_OK) goto done;
      hat[i]  = hd.len;
      hd.len += (size_t) sprintf(hd.p + hd.len, "%s/%d/%d_%d RQ=0.%d\n", x.prefix, h[0], h[1], h[2], h[3]);
      total  += hd.len - (size_t) hat[i];
      ooff[i] = total;
      total  += 5 * ((size_t) x.len[i] + 1);              /* undexqv.c:206-207 */
    }
  hat[x.n] = hd.len;
  res = malloc(total + 16);
  if (!res) { rc = DX_E_NOMEM; goto done; }

  if (x.n > 0)
    { TRY(dx_qv_set_coding(ctx, &x.coding, 0));
      TRY(dupload(&pool, img, n, &d_in));
      TRY(dupload(&pool, x.rec_off, (x.n + 1) * 8, &d_rec));
      TRY(dupload(&pool, x.hdr_off, (x.n + 1) * 8, &d_hoff));
      TRY(dupload(&pool, x.seg, x.n * 5 * 4, &d_seg));
      TRY(dupload(&pool, x.len, x.n * 4, &d_len));
      TRY(dupload(&pool, ooff, x.n * 8, &d_ooff));
      TRY(dalloc(&pool, total, &d_out));
      TRY(dx_qv_decode(ctx, d_in, d_rec, d_hoff, d_seg, d_len, x.n,
                       (upper ? DX_DECODE_UPPER : 0) | (x.flip ? DX_DECODE_FLIP : 0), d_out, d_ooff));
      TRY(dx_d2h(ctx, res, d_out, total));
      for (i = 0; i < x.n; i++)
        memcpy(res + ooff[i] - (hat[i+1] - hat[i]), hd.p + hat[i], (size_t) (hat[i+1] - hat[i]));
    }
  *out = res; *out_len = total; res = NULL;
  rc = DX_OK;

done:
  dfree_all(&pool);
  dx_qv_index_free(&x);
  free(ooff); free(hat); free(hd.p); free(res);
  return rc;
}
