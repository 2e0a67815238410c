/*
 * dx_host.c -- host-only parts of libdexgpu (plain C, as the reference's host code is C):
 *   record framing of the three formats, the Huffman scheme builder and the (de)serialisation
 *   of the .dexqv coding header.  O(records) / O(256 log 256) work: it stays on the CPU.
 */
#include <stdlib.h>
#include <string.h>

#include "dexgpu.h"
#include "dx_env.h"

/* ---- DEXGPU_TEST (dx_env.h) ------------------------------------------------------------------------------------------- */
const char *dx_test_str(const char *key)
{ static __thread char val[160];
  const char *e = getenv("DEXGPU_TEST");
  const size_t kl = strlen(key);
  if (e == NULL) return NULL;
  while (*e)
    { const char *t = e;
      size_t n;
      while (*e && *e != ',' && *e != ' ') e++;
      n = (size_t) (e - t);
      if (n >= kl && memcmp(t, key, kl) == 0 && (n == kl || t[kl] == '='))
        { size_t vl = n == kl ? 0 : n - kl - 1;
          if (vl >= sizeof(val)) vl = sizeof(val) - 1;
          memcpy(val, t + kl + (n == kl ? 0 : 1), vl);
          val[vl] = '\0';
          return val;
        }
      while (*e == ',' || *e == ' ') e++;
    }
  return NULL;
}

int dx_test_on(const char *key)
{ const char *v = dx_test_str(key);
  return v != NULL && v[0] != '0';
}

long long dx_test_num(const char *key, long long dflt)
{ const char *v = dx_test_str(key);
  return v != NULL && v[0] != '\0' ? strtoll(v, NULL, 0) : dflt;
}

/* ==========================================================================================
 *  record framing (dexta.c:187-198, dexar.c:159-163 + 193-204, dexqv.c:128-139)
 * ========================================================================================== */

uint16_t dx_snr_to_cnr(float snr)                         /* dexar.c:159-163 */
{ if (snr > 99.99)
    return 9999;
  return (uint16_t) ((uint32_t) (snr * 100.));
}

/* bytes of the well-delta chain: one 0xff per whole 255 of a non-negative delta, then one byte */
static size_t well_chain(int32_t well, int32_t lwell)
{ int64_t d = (int64_t) well - (int64_t) lwell;
  return (d >= 255) ? (size_t) (d / 255) + 1 : 1;
}

size_t dx_frame_bound(const int32_t *hdr4, uint64_t n, int32_t lwell, int kind)
{ size_t tot = 0;
  uint64_t i;
  for (i = 0; i < n; i++)
    { tot  += well_chain(hdr4[4*i], lwell) + 8 + (kind ? 8 : 4);
      lwell = hdr4[4*i];
    }
  return tot;
}

int dx_frame_headers(const int32_t *hdr4, const uint16_t *cnr4, uint64_t n, int kind,
                     int32_t *lwell, uint8_t *blob, uint64_t *off)
{ uint64_t i, at = 0;
  int32_t  last;
  if (hdr4 == NULL || lwell == NULL || blob == NULL || off == NULL || (kind && cnr4 == NULL))
    return DX_E_ARG;
  last = *lwell;
  for (i = 0; i < n; i++)
    { const int32_t *h = hdr4 + 4*i;
      int32_t well = h[0];
      off[i] = at;
      while (well - last >= 255)                          /* dexta.c:187-191 */
        { blob[at++] = 0xff;
          last += 255;
        }
      blob[at++] = (uint8_t) (well - last);
      last = well;
      memcpy(blob + at, h + 1, 4); at += 4;               /* beg */
      memcpy(blob + at, h + 2, 4); at += 4;               /* end */
      if (kind)
        { memcpy(blob + at, cnr4 + 4*i, 8); at += 8; }    /* dexar.c:204 */
      else
        { memcpy(blob + at, h + 3, 4); at += 4; }         /* qv, dexta.c:198 */
    }
  off[n] = at;
  *lwell = last;
  return DX_OK;
}

/* ==========================================================================================
 *  Huffman schemes (QV.c:91-220, 1029-1169)
 *
 *  The tree shape is part of the file format (the .dexqv header stores the code bits), so the
 *  construction order is reproduced exactly: leaves enter the heap in ascending symbol order
 *  (the escape pseudo-leaf first), the heap is built bottom-up, and the sift-down breaks count
 *  ties towards the RIGHT child while moving a child up only if it is strictly smaller.
 * ========================================================================================== */

typedef struct
  { uint64_t cnt[512];
    int16_t  kid[512][2];      /* kid[v][0] < 0: leaf, symbol in kid[v][1] */
    int      heap[260];
    int      hsize;
  } hwork;

static void sift(hwork *w, int s)
{ int held = w->heap[s], at = s;
  for (;;)
    { int l = 2*at, pick;
      if (l > w->hsize)
        break;
      pick = l;
      if (l + 1 <= w->hsize && !(w->cnt[w->heap[l+1]] > w->cnt[w->heap[l]]))
        pick = l + 1;                                     /* right child unless it is larger */
      if (!(w->cnt[held] > w->cnt[w->heap[pick]]))
        break;
      w->heap[at] = w->heap[pick];
      at = pick;
    }
  w->heap[at] = held;
}

static int huffman(const uint64_t *hist, const dx_scheme *prev, dx_scheme *out)
{ hwork w;
  int   nleaf = 0, nnode, i;

  w.hsize = 0;
  if (prev != NULL)                                       /* QV.c:162-167 */
    { w.cnt[0] = 0;
      w.kid[0][0] = -1;
      w.kid[0][1] = 255;
      w.heap[++w.hsize] = nleaf++;
    }
  for (i = 0; i < 256; i++)                               /* QV.c:168-178 */
    { if (hist[i] == 0)
        continue;
      if (prev != NULL && (prev->lens[i] > 16 || i == 255))
        w.cnt[0] += hist[i];
      else
        { w.cnt[nleaf] = hist[i];
          w.kid[nleaf][0] = -1;
          w.kid[nleaf][1] = (int16_t) i;
          w.heap[++w.hsize] = nleaf++;
        }
    }
  if (nleaf == 0)
    return DX_E_DEGENERATE;

  for (i = w.hsize / 2; i >= 1; i--)                      /* QV.c:180-181 */
    sift(&w, i);

  nnode = nleaf;
  for (i = 1; i < nleaf; i++)                             /* QV.c:183-194 */
    { int a = w.heap[1], b;
      w.heap[1] = w.heap[w.hsize--];
      sift(&w, 1);
      b = w.heap[1];
      w.cnt[nnode]    = w.cnt[a] + w.cnt[b];
      w.kid[nnode][0] = (int16_t) a;
      w.kid[nnode][1] = (int16_t) b;
      w.heap[1] = nnode++;
      sift(&w, 1);
    }

  memset(out->bits, 0, sizeof(out->bits));
  memset(out->lens, 0, sizeof(out->lens));
  { /* QV.c:125-137, iteratively: (node, code, len) stack; left edge 0, right edge 1 */
    int      sn[512], sl[512], sp = 0;
    uint32_t sc[512];
    sn[0] = nnode - 1; sc[0] = 0; sl[0] = 0; sp = 1;
    while (sp > 0)
      { int v = sn[--sp], len = sl[sp];
        uint32_t code = sc[sp];
        if (w.kid[v][0] < 0)
          { out->bits[w.kid[v][1]] = code;
            out->lens[w.kid[v][1]] = len;
          }
        else
          { sn[sp] = w.kid[v][1]; sc[sp] = (code << 1) | 1u; sl[sp] = len + 1; sp++;
            sn[sp] = w.kid[v][0]; sc[sp] = (code << 1);      sl[sp] = len + 1; sp++;
          }
      }
  }

  if (prev != NULL)                                       /* QV.c:203-210 */
    { out->type = 2;
      for (i = 0; i < 255; i++)
        if (prev->lens[i] > 16 || out->lens[i] > 16)
          { out->lens[i] = out->lens[255];
            out->bits[i] = out->bits[255];
          }
    }
  else                                                    /* QV.c:211-217 */
    { out->type = 0;
      for (i = 0; i < 256; i++)
        if (out->lens[i] > 16)
          out->type = 1;
    }
  return DX_OK;
}

static int scheme_for(const uint64_t *hist, dx_scheme *out)          /* QV.c:1069-1078 */
{ dx_scheme first;
  int e = huffman(hist, NULL, &first), i;
  if (e) return e;
  if (first.type)
    { e = huffman(hist, &first, out);
      if (e) return e;
    }
  else
    *out = first;
  for (i = 0; i < 256; i++)
    if (out->lens[i] > 16)
      return DX_E_UNSUPPORTED;
  return DX_OK;
}

int dx_qv_build(const uint64_t hist[6][256], uint64_t totChar, const dx_qv_params *p, int lossy,
                dx_qv_coding *out)
{ uint64_t h[6][256];
  int      k, e, delChar, subChar;

  if (hist == NULL || p == NULL || out == NULL)
    return DX_E_ARG;
  memcpy(h, hist, sizeof(h));
  memset(out, 0, sizeof(*out));
  delChar = p->delChar;
  subChar = p->subChar;
  if (delChar > 255 || subChar > 255)
    return DX_E_ARG;

  for (k = 0; k < 256; k++)                               /* run bins start at 1, QV.c:934-935 */
    { h[DX_DRUN][k] += 1;
      h[DX_SRUN][k] += 1;
    }

  if (totChar < 200000 || (subChar >= 0 && (double) h[DX_SUB][subChar] < .5 * (double) totChar))
    subChar = -1;                                         /* QV.c:1044-1045 */

  if (lossy)                                              /* QV.c:1049-1065 */
    { for (k = 0; k < 256; k += 2)
        { h[DX_INS][k]  += h[DX_INS][k+1];
          h[DX_INS][k+1] = 0;
        }
      for (k = 0; k < 256; k += 4)
        { h[DX_MRG][k]  += h[DX_MRG][k+1] + h[DX_MRG][k+2] + h[DX_MRG][k+3];
          h[DX_MRG][k+1] = h[DX_MRG][k+2] = h[DX_MRG][k+3] = 0;
        }
    }

  if (delChar >= 0)                                       /* QV.c:1097-1108 */
    { h[DX_DEL][delChar] = 0;
      if ((e = scheme_for(h[DX_DRUN], &out->s[DX_DRUN]))) return e;
    }
  if ((e = scheme_for(h[DX_DEL], &out->s[DX_DEL]))) return e;
  if ((e = scheme_for(h[DX_INS], &out->s[DX_INS]))) return e;   /* QV.c:1121-1122 */
  if ((e = scheme_for(h[DX_MRG], &out->s[DX_MRG]))) return e;
  if (subChar >= 0)                                       /* QV.c:1124-1135 */
    { h[DX_SUB][subChar] = 0;
      if ((e = scheme_for(h[DX_SRUN], &out->s[DX_SRUN]))) return e;
    }
  if ((e = scheme_for(h[DX_SUB], &out->s[DX_SUB]))) return e;

  out->delChar = delChar;
  out->subChar = subChar;
  return DX_OK;
}

/* Upper bound, from the raw histograms of a batch (dx_qv_hist) and the coding built from them or from
 * the whole file, of the bytes Compress_Next_QVentry writes for the batch's `n` entries (without framing
 * bytes): every symbol costs exactly its code (QV.c:427-434, 489-497), every non-run symbol of a
 * run-coded line is preceded by one run token and an entry ends with at most one more (QV.c:475-487):
 * the runs the histogram counted (entries from del_first / sub_first on) are priced exactly, the others
 * at the dearest run token; per entry at most one partial and one pad word for each of the four code
 * segments (QV.c:436-442) and a last partial tag byte.  Sizes d_out for dx_qv_encode_onepass.        */
uint64_t dx_qv_out_bound(const uint64_t hist[6][256], uint64_t n, const dx_qv_coding *c, int lossy)
{ uint64_t bits = 0, tags = 0;
  int      s, x;
  if (hist == NULL || c == NULL) return 0;
  for (s = 0; s < 4; s++)
    { const int      rc  = s == DX_DEL ? c->delChar : (s == DX_SUB ? c->subChar : -1);
      const int      rs  = s == DX_DEL ? DX_DRUN : DX_SRUN;
      const int      msk = !lossy ? 0xff : (s == DX_INS ? 0xfe : (s == DX_MRG ? 0xfc : 0xff));   /* QV.c:1406-1415 */
      const dx_scheme *sc = &c->s[s];
      uint64_t nonrun = 0;
      for (x = 0; x < 256; x++)
        { const int y = x & msk;
          uint64_t  l = (uint64_t) sc->lens[y];
          if (x == rc) continue;
          if (sc->type == 2 && sc->bits[y] == sc->bits[255] && sc->lens[y] == sc->lens[255]) l += 8;   /* QV.c:432 */
          bits   += hist[s][x] * l;
          nonrun += hist[s][x];
        }
      if (s == DX_DEL) tags = nonrun;                     /* Pack_Tag keeps the tags of the non-run positions */
      if (rc >= 0)
        { const dx_scheme *rn = &c->s[rs];
          uint64_t counted = 0, dearest = 0;
          for (x = 0; x < 256; x++)
            { uint64_t l = (uint64_t) rn->lens[x];
              if (rn->bits[x] == rn->bits[255] && rn->lens[x] == rn->lens[255]) l += 16;               /* QV.c:486 */
              if (l > dearest) dearest = l;
              bits    += hist[rs][x] * l;
              counted += hist[rs][x];
            }
          if (nonrun + n > counted)
            bits += (nonrun + n - counted) * dearest;
        }
    }
  return (bits + 7) / 8 + tags / 4 + n * (4 * 8 + 1) + 64;
}

/* ==========================================================================================
 *  coding header image (QV.c:300-375, 1173-1320)
 * ========================================================================================== */

typedef struct { uint8_t *p; size_t at, cap; } wbuf;

static void wput(wbuf *b, const void *src, size_t n)
{ if (b->at + n <= b->cap)
    memcpy(b->p + b->at, src, n);
  b->at += n;
}

static void put_scheme(wbuf *b, const dx_scheme *s)       /* QV.c:300-318 */
{ int i;
  uint8_t x = (uint8_t) s->type;
  wput(b, &x, 1);
  for (i = 0; i < 256; i++)
    { x = (uint8_t) s->lens[i];
      wput(b, &x, 1);
      if (x > 0)
        wput(b, &s->bits[i], 4);
    }
}

int dx_qv_write_coding(const dx_qv_coding *c, const char *prefix, size_t plen,
                       uint8_t *buf, size_t cap, size_t *written)
{ wbuf     b;
  uint16_t half;
  int32_t  len = (int32_t) plen;

  if (c == NULL || (plen && prefix == NULL) || written == NULL)
    return DX_E_ARG;
  b.p = buf; b.at = 0; b.cap = buf ? cap : 0;

  half = 0x33cc;                                           wput(&b, &half, 2);   /* QV.c:1180 */
  half = c->delChar < 0 ? 256 : (uint16_t) c->delChar;     wput(&b, &half, 2);
  half = c->subChar < 0 ? 256 : (uint16_t) c->subChar;     wput(&b, &half, 2);
  wput(&b, &len, 4);
  wput(&b, prefix, plen);
  put_scheme(&b, &c->s[DX_DEL]);                                                 /* QV.c:1202-1209 */
  if (c->delChar >= 0) put_scheme(&b, &c->s[DX_DRUN]);
  put_scheme(&b, &c->s[DX_INS]);
  put_scheme(&b, &c->s[DX_MRG]);
  put_scheme(&b, &c->s[DX_SUB]);
  if (c->subChar >= 0) put_scheme(&b, &c->s[DX_SRUN]);

  *written = b.at;
  return (b.at <= b.cap) ? DX_OK : DX_E_SPACE;
}

typedef struct { const uint8_t *p; size_t at, n; int bad; } rbuf;

static void rget(rbuf *b, void *dst, size_t k)
{ if (b->at + k > b->n) { b->bad = 1; memset(dst, 0, k); b->at = b->n; return; }
  memcpy(dst, b->p + b->at, k);
  b->at += k;
}

static uint16_t flip16(uint16_t v) { return (uint16_t) ((v << 8) | (v >> 8)); }
static uint32_t flip32(uint32_t v)
{ return (v << 24) | ((v & 0xff00u) << 8) | ((v >> 8) & 0xff00u) | (v >> 24); }

static int get_scheme(rbuf *b, int flip, dx_scheme *s)    /* QV.c:322-363 */
{ int i;
  uint8_t x;
  rget(b, &x, 1);
  s->type = x;
  for (i = 0; i < 256; i++)
    { rget(b, &x, 1);
      s->lens[i] = x;
      s->bits[i] = 0;
      if (x > 0)
        { uint32_t v;
          rget(b, &v, 4);
          s->bits[i] = flip ? flip32(v) : v;
        }
      if (x > 16)
        return DX_E_UNSUPPORTED;
    }
  return b->bad ? DX_E_FORMAT : DX_OK;
}

int dx_qv_read_coding(const uint8_t *buf, size_t n, dx_qv_coding *c, int *flip,
                      char *prefix, size_t pcap, size_t *consumed)
{ rbuf     b;
  uint16_t half;
  uint32_t len;
  int      fl, e;

  if (buf == NULL || c == NULL)
    return DX_E_ARG;
  b.p = buf; b.at = 0; b.n = n; b.bad = 0;
  memset(c, 0, sizeof(*c));

  rget(&b, &half, 2);                                      /* QV.c:1222-1226 */
  fl = (half != 0x33cc);
  rget(&b, &half, 2); if (fl) half = flip16(half);
  c->delChar = half >= 256 ? -1 : half;
  rget(&b, &half, 2); if (fl) half = flip16(half);
  c->subChar = half >= 256 ? -1 : half;
  rget(&b, &len, 4);  if (fl) len = flip32(len);
  if (b.bad || len > n)
    return DX_E_FORMAT;
  if (prefix != NULL)
    { if ((size_t) len + 1 > pcap)
        return DX_E_SPACE;
      rget(&b, prefix, len);
      prefix[len] = '\0';
    }
  else
    b.at += len;

  if ((e = get_scheme(&b, fl, &c->s[DX_DEL]))) return e;                          /* QV.c:1281-1302 */
  if (c->delChar >= 0 && (e = get_scheme(&b, fl, &c->s[DX_DRUN]))) return e;
  if ((e = get_scheme(&b, fl, &c->s[DX_INS]))) return e;
  if ((e = get_scheme(&b, fl, &c->s[DX_MRG]))) return e;
  if ((e = get_scheme(&b, fl, &c->s[DX_SUB]))) return e;
  if (c->subChar >= 0 && (e = get_scheme(&b, fl, &c->s[DX_SRUN]))) return e;
  if (b.bad)
    return DX_E_FORMAT;
  if (flip) *flip = fl;
  if (consumed) *consumed = b.at;
  return DX_OK;
}

/* ==========================================================================================
 *  text front end: index .quiva / .fasta / .arrow images (host, O(file) memchr work)
 *
 *  Replaces the fgets loops of QV.c:751-798 + the header checks of QV.c:954-968 (quiva) and
 *  dexta.c:104-183 / dexar.c:103-188 (fasta/arrow).  The kernels then read the streams straight
 *  from the file image through the offsets produced here.
 * ========================================================================================== */
#include <stdio.h>

#define DX_LINE_LIMIT 100000            /* MAX_BUFFER, dexta.c:21 */

typedef struct { const uint8_t *p; size_t n, at; uint64_t line; } tsrc;

/* 1 = line read, 0 = end of input, -1 = last line has no newline */
static int get_line(tsrc *t, const uint8_t **s, size_t *len)
{ const uint8_t *nl;
  if (t->at >= t->n)
    return 0;
  t->line += 1;
  nl = memchr(t->p + t->at, '\n', t->n - t->at);
  if (nl == NULL)                                        /* the text's last line, unterminated: all of it */
    { *s   = t->p + t->at;
      *len = t->n - t->at;
      t->at = t->n;
      return -1;
    }
  *s   = t->p + t->at;
  *len = (size_t) (nl - *s);
  t->at += *len + 1;
  return 1;
}

/* sscanf needs a terminated string: copy the (bounded) header tail */
static int scan_tail(const uint8_t *s, size_t n, const char *fmt, int32_t *f, float *snr)
{ char tmp[400];
  int  w, b, e, q, k;
  if (n > sizeof(tmp) - 2) n = sizeof(tmp) - 2;
  memcpy(tmp, s, n);
  tmp[n] = '\n';
  tmp[n+1] = '\0';
  if (snr != NULL)
    { k = sscanf(tmp, fmt, &w, &b, &e, snr, snr+1, snr+2, snr+3);
      q = 0;
    }
  else
    { q = 0;
      k = sscanf(tmp, fmt, &w, &b, &e, &q);
    }
  if (k >= 1) f[0] = w;
  if (k >= 2) f[1] = b;
  if (k >= 3) f[2] = e;
  f[3] = (snr == NULL && k >= 4) ? q : 0;
  return k;
}

static int idx_fail(uint64_t line, int code, uint64_t *errline, int *errcode)
{ if (errline) *errline = line;
  if (errcode) *errcode = code;
  return DX_E_FORMAT;
}

int dx_index_quiva(const uint8_t *text, size_t n, uint64_t cap,
                   uint64_t *off, uint32_t *len, int32_t *hdr4,
                   uint64_t *count, size_t *prefix_len, uint64_t *errline, int *errcode)
{ tsrc     t;
  uint64_t k = 0;
  if (text == NULL && n) return DX_E_ARG;
  t.p = text; t.n = n; t.at = 0; t.line = 0;
  if (prefix_len) *prefix_len = 0;
  for (;;)
    { const uint8_t *h, *s, *slash;
      size_t  hl, sl = 0, first = 0;
      int32_t f[4];
      int     r, j;

      r = get_line(&t, &h, &hl);                          /* QV.c:948-952 */
      if (r == 0) break;
      if (r < 0) return idx_fail(t.line, DX_IDX_NO_NEWLINE, errline, errcode);
      if (hl == 0 || h[0] != '@')                         /* QV.c:954-957 */
        return idx_fail(t.line, DX_IDX_NO_HEADER, errline, errcode);
      slash = hl > 1 ? memchr(h + 1, '/', hl - 1) : NULL; /* QV.c:958 */
      if (slash == NULL)
        return idx_fail(t.line, DX_IDX_BAD_HEADER, errline, errcode);
      if (scan_tail(slash + 1, hl - (size_t) (slash + 1 - h), "%d/%d_%d RQ=0.%d\n", f, NULL) != 4)
        return idx_fail(t.line, DX_IDX_BAD_HEADER, errline, errcode);     /* QV.c:964-968 */
      if (k == 0 && prefix_len) *prefix_len = (size_t) (slash - h);      /* dexqv.c:94-102 */

      for (j = 0; j < 5; j++)                             /* QV.c:973-978, 785-796 */
        { r = get_line(&t, &s, &sl);
          if (r == 0) return idx_fail(t.line + 1, DX_IDX_INCOMPLETE, errline, errcode);
          if (r < 0)                                      /* the file's last line has no newline.  The reference grows its buffer
                                                             only for an entry's FIRST line and says so (QV.c:771-781); lines 2-5 it
                                                             reads with fgets and compares strlen -- newline included -- with the first
                                                             line's (QV.c:786-795): one character more than the others passes (the
                                                             last character stands where the newline would), anything else is ragged */
            { if (j == 0) return idx_fail(t.line, DX_IDX_NO_NEWLINE, errline, errcode);
              if (sl != first + 1) return idx_fail(t.line, DX_IDX_RAGGED, errline, errcode);
              sl = first;
            }
          if (j == 0)
            { first = sl;
              if (off != NULL && k < cap) off[k] = (uint64_t) (s - text);
            }
          else if (sl != first)
            return idx_fail(t.line, DX_IDX_RAGGED, errline, errcode);
        }
      if (first > 0x7fffffffu)
        return idx_fail(t.line, DX_IDX_TOO_LONG, errline, errcode);
      if (k < cap)
        { if (len  != NULL) len[k] = (uint32_t) first;
          if (hdr4 != NULL) memcpy(hdr4 + 4*k, f, sizeof(f));
        }
      k += 1;
    }
  if (count) *count = k;
  return DX_OK;
}

int dx_index_seq(int arrow, const uint8_t *text, size_t n, uint64_t cap,
                 uint64_t *off, uint32_t *tlen, uint32_t *nsym, int32_t *hdr4, uint16_t *cnr4,
                 uint64_t *count, size_t *prefix_len, uint64_t *errline, int *errcode)
{ tsrc     t;
  uint64_t k = 0;
  const uint8_t *h, *slash;
  size_t   hl;
  int      r;

  if (text == NULL && n) return DX_E_ARG;
  t.p = text; t.n = n; t.at = 0; t.line = 0;
  if (count) *count = 0;
  if (prefix_len) *prefix_len = 0;

  r = get_line(&t, &h, &hl);                              /* dexta.c:108-122 */
  if (r == 0) return idx_fail(1, DX_IDX_EMPTY, errline, errcode);
  if (r < 0 || hl + 1 >= DX_LINE_LIMIT) return idx_fail(1, DX_IDX_TOO_LONG, errline, errcode);
  if (hl == 0 || h[0] != '>') return idx_fail(1, DX_IDX_NO_HEADER, errline, errcode);
  slash = memchr(h, '/', hl);
  if (slash == NULL) return idx_fail(1, DX_IDX_BAD_HEADER, errline, errcode);
  if (prefix_len) *prefix_len = (size_t) (slash - h);

  for (;;)                                                /* dexta.c:139-205 */
    { int32_t  f[4];
      float    snr[4];
      const uint8_t *s;
      size_t   sl, start, end, lines = 0;
      int      more = 0, x;

      slash = hl > 1 ? memchr(h + 1, '/', hl - 1) : NULL; /* dexta.c:146 */
      if (slash == NULL) return idx_fail(t.line, DX_IDX_BAD_HEADER, errline, errcode);
      if (arrow)
        { x = scan_tail(slash + 1, hl - (size_t) (slash + 1 - h), "%d/%d_%d SN=%f,%f,%f,%f\n", f, snr);
          if (x != 7) return idx_fail(t.line, DX_IDX_BAD_HEADER, errline, errcode);   /* dexar.c:152-157 */
        }
      else
        { x = scan_tail(slash + 1, hl - (size_t) (slash + 1 - h), "%d/%d_%d RQ=0.%d\n", f, NULL);
          if (x < 3) return idx_fail(t.line, DX_IDX_BAD_HEADER, errline, errcode);    /* dexta.c:151-157 */
        }

      start = t.at;                                       /* dexta.c:161-183: lines up to the next '>' */
      end   = t.at;
      for (;;)
        { r = get_line(&t, &s, &sl);
          if (r == 0 && k > 0 && t.at == start)           /* a header -- not the file's first -- with the end of the file right behind
                                                             it: the reference's fgets leaves its buffer as the header's parse left it,
                                                             finds no newline where it looks for one and says the NEXT line is too
                                                             long (dexta.c:163-172; a file of one header alone passes, as do empty
                                                             lines behind the header: checked against the binaries, tools/stress_cli.py) */
            return idx_fail(t.line + 1, DX_IDX_TOO_LONG, errline, errcode);
          if (r == 0) break;
          if (r < 0 || sl + 1 >= DX_LINE_LIMIT) return idx_fail(t.line, DX_IDX_TOO_LONG, errline, errcode);
          if (sl > 0 && s[0] == '>')
            { h = s; hl = sl; more = 1;
              break;
            }
          lines += 1;
          end = t.at;
        }
      if (end - start - lines > 0x7fffffffu) return idx_fail(t.line, DX_IDX_TOO_LONG, errline, errcode);
      if (k < cap)
        { if (off  != NULL) off[k]  = (uint64_t) start;
          if (tlen != NULL) tlen[k] = (uint32_t) (end - start);
          if (nsym != NULL) nsym[k] = (uint32_t) (end - start - lines);
          if (hdr4 != NULL) memcpy(hdr4 + 4*k, f, sizeof(f));
          if (arrow && cnr4 != NULL)
            { int j;
              for (j = 0; j < 4; j++) cnr4[4*k + j] = dx_snr_to_cnr(snr[j]);
            }
        }
      k += 1;
      if (!more) break;
    }
  if (count) *count = k;
  return DX_OK;
}

/* ==========================================================================================
 *  bare-file index: walk a .dexqv image front to back (host)
 *
 *  The format stores no record or segment lengths (QV.c:1428-1481; undexqv.c:119-208), so the
 *  start of every segment is known only after the previous one has been walked code by code.
 *  This is that walk: it decodes code LENGTHS only (plus what it needs to count symbols) and
 *  yields the index dx_qv_decode takes.  Inherently sequential; everything that produces
 *  symbols runs on the GPU afterwards.
 * ========================================================================================== */

typedef struct { uint16_t e[0x10000]; } wlut;          /* len << 8 | symbol, by 16-bit window (QV.c:365-372) */

static void build_wlut(const dx_scheme *s, wlut *t)
{ int i;
  memset(t->e, 0, sizeof(t->e));
  for (i = 0; i < 256; i++)                             /* ascending: 255 wins shared escape codes */
    if (s->lens[i] > 0 && s->lens[i] <= 16)
      { uint32_t base = (s->bits[i] << (16 - s->lens[i])) & 0xffffu, cnt = 1u << (16 - s->lens[i]), j;
        for (j = 0; j < cnt; j++)
          t->e[(base + j) & 0xffffu] = (uint16_t) ((s->lens[i] << 8) | i);
      }
}

typedef struct { const uint8_t *p, *end; uint64_t buf; int nb; uint64_t T; int flip; } wrd;

static void w_fill(wrd *r)
{ if (r->p + 4 <= r->end)                               /* (well predicted; the data-dependent part is branch-free) */
    { uint32_t w;
      const uint64_t take = (uint64_t) 0 - (uint64_t) (r->nb <= 32);      /* all ones: the buffer has room for a word */
      memcpy(&w, r->p, 4);
      if (r->flip) w = flip32(w);
      r->buf |= ((uint64_t) w << ((32 - r->nb) & 63)) & take;
      r->nb  += (int) (32 & take);
      r->p   += 4 & take;
    }
}
static uint32_t w_peek(wrd *r) { w_fill(r); return (uint32_t) (r->buf >> 48); }
static void w_skip(wrd *r, int n) { r->buf <<= n; r->nb -= n; r->T += (uint64_t) n; }

static uint32_t pad_words(uint64_t T, uint32_t last)    /* QV.c:436-442 */
{ uint32_t olen = (uint32_t) T & 31u, llen = (uint32_t) (T - last) & 31u;
  uint32_t w = (uint32_t) (T >> 5) + (olen ? 1u : 0u);
  if (olen > 0) return w + ((llen > 16u && olen > llen) ? 1u : 0u);
  return w + ((T > 0 && llen > 16u) ? 1u : 0u);
}

/* Several codes per look-up for the walk, which needs lengths only: for every 12-bit window, how
   many whole codes it holds, their total length and the length of the last one (escape codes end
   a group: the 8-bit literal that follows is not a code).  8 KB per scheme: stays in L1. */
#define MW_BITS 12
typedef struct { uint8_t nbits, nsym, last, pad; } mwent;
typedef struct { mwent e[1 << MW_BITS]; } mwlut;

static void build_mwlut(const wlut *t, int esc, mwlut *m)
{ uint32_t x;
  for (x = 0; x < (1u << MW_BITS); x++)
    { uint32_t pos = 0, cnt = 0, last = 0;
      for (;;)
        { uint32_t e = t->e[((x << (16 - MW_BITS)) << pos) & 0xffffu], l = e >> 8;
          if (l == 0 || pos + l > MW_BITS || (esc && (e & 0xff) == 255)) break;
          pos += l; cnt += 1; last = l;
        }
      m->e[x].nbits = (uint8_t) pos; m->e[x].nsym = (uint8_t) cnt; m->e[x].last = (uint8_t) last; m->e[x].pad = 0;
    }
}

/* bytes of a plain-coded segment of rlen symbols starting at p (QV.c:510-599) */
static int64_t walk_plain(const uint8_t *p, const uint8_t *end, uint32_t rlen, const wlut *t, const mwlut *m,
                          int esc, int flip)
{ wrd r = { p, end, 0, 0, 0, flip };
  uint32_t j = 0, last = 0;
  int64_t bytes;
  while (j < rlen)
    { uint32_t w = w_peek(&r);
      const mwent g = m->e[w >> (16 - MW_BITS)];
      if (r.nb < 0) return -1;                           /* more bits consumed than the image holds */
      if (g.nsym && j + g.nsym <= rlen)
        { w_skip(&r, g.nbits);
          j   += g.nsym;
          last = g.last;
          continue;
        }
      { uint32_t e = t->e[w];
        last = e >> 8;
        if (last == 0) return -1;                          /* no such code */
        w_skip(&r, (int) last);
        if (esc && (e & 0xff) == 255)
          { w_peek(&r); w_skip(&r, 8); last = 8; }
        j += 1;
      }
    }
  bytes = 4 * (int64_t) pad_words(r.T, last);
  return (p + bytes <= end) ? bytes : -1;
}

/* The same for a run-coded stream: a (run code, symbol code) pair that fits the window whole and
   needs no literal (run < 255, symbol not escaped). */
typedef struct { uint8_t nbits, run, last, ok; } rwent;
typedef struct { rwent e[1 << MW_BITS]; } rwlut;

static void build_rwlut(const wlut *rt, const wlut *nt, int esc, rwlut *m)
{ uint32_t x;
  for (x = 0; x < (1u << MW_BITS); x++)
    { const uint32_t w  = x << (16 - MW_BITS);
      const uint32_t e1 = rt->e[w], l1 = e1 >> 8;
      rwent g = { 0, 0, 0, 0 };
      if (l1 > 0 && l1 < MW_BITS && (e1 & 0xff) != 255)
        { const uint32_t e2 = nt->e[(w << l1) & 0xffffu], l2 = e2 >> 8;
          if (l2 > 0 && l1 + l2 <= MW_BITS && !(esc && (e2 & 0xff) == 255))
            { g.nbits = (uint8_t) (l1 + l2); g.run = (uint8_t) (e1 & 0xff); g.last = (uint8_t) l2; g.ok = 1; }
        }
      m->e[x] = g;
    }
}

/* The same tables for the device walk (dx_qv_walk.hip), 16 bits an entry: layout in dx_walk.h. */
#include "dx_walk.h"
int dx_walk_luts_build(const dx_qv_coding *cd, uint8_t *blob, int esc[4])
{ wlut  *lut[6] = { NULL, NULL, NULL, NULL, NULL, NULL };
  mwlut *ml = malloc(sizeof(mwlut));
  rwlut *rl = malloc(sizeof(rwlut));
  int    s, rc = DX_OK;
  uint32_t x;
  if (cd == NULL || blob == NULL || esc == NULL || ml == NULL || rl == NULL) { free(ml); free(rl); return cd && blob && esc ? DX_E_NOMEM : DX_E_ARG; }
  memset(blob, 0, WALK_BLOB_BYTES);
  for (s = 0; s < 6; s++)
    { if ((s == DX_DRUN && cd->delChar < 0) || (s == DX_SRUN && cd->subChar < 0)) continue;
      lut[s] = malloc(sizeof(wlut));
      if (lut[s] == NULL) { rc = DX_E_NOMEM; goto done; }
      build_wlut(&cd->s[s], lut[s]);
      memcpy(blob + WALK_W16_OFF + (size_t) s * 65536u * 2u, lut[s]->e, 65536u * 2u);
    }
  for (s = 0; s < 4; s++)
    { uint16_t *m16 = (uint16_t *) (blob + WALK_MW_OFF) + (size_t) s * 4096u;
      esc[s] = cd->s[s].type == 2;
      build_mwlut(lut[s], esc[s], ml);
      for (x = 0; x < 4096u; x++)
        { const uint32_t e = lut[s]->e[x << 4], l = e >> 8;
          const uint32_t first = (l > 0 && l <= WALK_WIN && !(esc[s] && (e & 0xff) == 255)) ? l : 0u;
          m16[x] = ml->e[x].nsym ? (uint16_t) (ml->e[x].nbits | (ml->e[x].last << 4) | (ml->e[x].nsym << 8)) : 0;
          ((uint16_t *) (blob + WALK_ONE_OFF))[(size_t) s * 4096u + x] = first ? (uint16_t) (first | (first << 4) | (1u << 8)) : 0;
        }
    }
  for (s = 0; s < 2; s++)
    { const int sym = s ? DX_SUB : DX_DEL, run = s ? DX_SRUN : DX_DRUN;
      uint16_t *r16 = (uint16_t *) (blob + WALK_RW_OFF) + (size_t) s * 4096u;
      uint16_t *o16 = (uint16_t *) (blob + WALK_R1_OFF) + (size_t) s * 4096u;
      if (lut[run] == NULL) continue;
      build_rwlut(lut[run], lut[sym], esc[sym], rl);
      for (x = 0; x < 4096u; x++)
        { const uint32_t e1 = lut[run]->e[x << 4], l1 = e1 >> 8;
          r16[x] = rl->e[x].ok ? (uint16_t) (rl->e[x].nbits | (rl->e[x].last << 4) | (((uint32_t) rl->e[x].run + 1u) << 8)) : 0;
          o16[x] = (l1 > 0 && l1 <= WALK_WIN && (e1 & 0xff) != 255) ? (uint16_t) (l1 | (l1 << 4) | ((e1 & 0xff) << 8)) : 0;
        }
    }
done:
  for (s = 0; s < 6; s++) free(lut[s]);
  free(ml); free(rl);
  return rc;
}

/* run-coded segment (QV.c:604-691); *nonrun receives the number of non-run symbols */
static int64_t walk_runs(const uint8_t *p, const uint8_t *end, uint32_t rlen, const wlut *nt, int esc,
                         const wlut *rt, const rwlut *pair, uint32_t *nonrun, int flip)
{ wrd r = { p, end, 0, 0, 0, flip };
  uint32_t j = 0, last = 0, nn = 0;
  int64_t bytes;
  while (j < rlen)
    { uint32_t w = w_peek(&r), e, c;
      const rwent g = pair->e[w >> (16 - MW_BITS)];
      if (r.nb < 0) return -1;                           /* more bits consumed than the image holds */
      if (g.ok && j + g.run < rlen)                      /* run, then a symbol that exists */
        { w_skip(&r, g.nbits);
          j   += (uint32_t) g.run + 1u;
          nn  += 1;
          last = g.last;
          continue;
        }
      e = rt->e[w]; c = e & 0xff;
      last = e >> 8;
      if (last == 0) return -1;                            /* no such code */
      w_skip(&r, (int) last);
      if (c == 255)
        { c = w_peek(&r); w_skip(&r, 16); last = 16; }
      if (c > rlen - j) return -1;
      j += c;
      if (j < rlen)
        { e = nt->e[w_peek(&r)];
          last = e >> 8;
          if (last == 0) return -1;
          w_skip(&r, (int) last);
          if (esc && (e & 0xff) == 255)
            { w_peek(&r); w_skip(&r, 8); last = 8; }
          j  += 1;
          nn += 1;
        }
    }
  *nonrun = nn;
  bytes = 4 * (int64_t) pad_words(r.T, last);
  return (p + bytes <= end) ? bytes : -1;
}

/* ---- the same walks, leaving the GROUP INDEX the wave-per-line decoders take (dx_layout.h; device side: dx_device.hpp,
 * k_qv_encode_fast writes it, k_qv_decode_sub / k_qv_decode_runs read it).  The walk passes every code anyway; with the
 * index a bare file decodes on the kernels that otherwise only serve a stream the same context has just encoded.
 * One look-up per symbol here (the several-codes-per-look-up tables above do not stop at group boundaries). ---------- */
#include "dx_layout.h"

/* plain line: one byte per group of 16 symbols = the group's code bits minus its symbols.  Several codes per look-up
   (mwlut) while they stay inside the group, one code at a time across its boundary. */
static int64_t walk_plain_ix(const uint8_t *p, const uint8_t *end, uint32_t rlen, const wlut *t, const mwlut *m, int esc, int flip,
                             uint8_t *share)
{ wrd r = { p, end, 0, 0, 0, flip };
  uint32_t j = 0, last = 0, gbits = 0, none = 0;
  int64_t bytes;
  while (j < rlen)
    { uint32_t w = w_peek(&r);
      const mwent g = m->e[w >> (16 - MW_BITS)];
      if (r.nb < 0) return -1;
      if (g.nsym && (j & 15u) + g.nsym <= 16u && j + g.nsym <= rlen)
        { w_skip(&r, g.nbits);
          j += g.nsym; gbits += g.nbits; last = g.last;
        }
      else
        { uint32_t e = t->e[w];
          last = e >> 8;
          if (last == 0) return -1;
          w_skip(&r, (int) last);
          gbits += last;
          if (esc && (e & 0xff) == 255)
            { w_peek(&r); w_skip(&r, 8); last = 8; gbits += 8; }
          j += 1;
        }
      if ((j & 15u) == 0 || j == rlen)
        { const uint32_t valid = (j & 15u) ? (j & 15u) : 16u;
          if (gbits - valid > 254u) none = 1;               /* does not fit the byte (escape schemes): no index for this line */
          share[(j - 1) >> 4] = (uint8_t) (gbits - valid);
          gbits = 0;
        }
    }
  if ((none || esc) && rlen) share[0] = (uint8_t) DXL_SUB_NONE;
  bytes = 4 * (int64_t) pad_words(r.T, last);
  return (p + bytes <= end) ? bytes : -1;
}

/* run-coded line: per token its bits and the positions it covers (run + 1), into tb / ts (room for rlen tokens) */
static int64_t walk_runs_ix(const uint8_t *p, const uint8_t *end, uint32_t rlen, const wlut *nt, int esc,
                            const wlut *rt, const rwlut *pair, uint32_t *nonrun, int flip, uint16_t *tb, uint32_t *ts)
{ wrd r = { p, end, 0, 0, 0, flip };
  uint32_t j = 0, last = 0, nn = 0;
  int64_t bytes;
  while (j < rlen)
    { uint32_t w = w_peek(&r), e, c, bits;
      const rwent g = pair->e[w >> (16 - MW_BITS)];
      if (r.nb < 0) return -1;
      if (g.ok && j + g.run < rlen)
        { w_skip(&r, g.nbits);
          tb[nn] = g.nbits; ts[nn] = (uint32_t) g.run + 1u;
          j   += (uint32_t) g.run + 1u;
          nn  += 1;
          last = g.last;
          continue;
        }
      e = rt->e[w]; c = e & 0xff;
      last = e >> 8;
      if (last == 0) return -1;
      w_skip(&r, (int) last);
      bits = last;
      if (c == 255)
        { c = w_peek(&r); w_skip(&r, 16); last = 16; bits += 16; }
      if (c > rlen - j) return -1;
      j += c;
      if (j < rlen)
        { e = nt->e[w_peek(&r)];
          last = e >> 8;
          if (last == 0) return -1;
          w_skip(&r, (int) last);
          bits += last;
          if (esc && (e & 0xff) == 255)
            { w_peek(&r); w_skip(&r, 8); last = 8; bits += 8; }
          tb[nn] = (uint16_t) bits; ts[nn] = c + 1u;
          j  += 1;
          nn += 1;
        }
    }
  *nonrun = nn;
  bytes = 4 * (int64_t) pad_words(r.T, last);
  return (p + bytes <= end) ? bytes : -1;
}

/* the group words of a run-coded line from its tokens, as k_qv_encode_fast cuts them: passes of 512 tokens, in a pass
   of m tokens lane l holds the (m + 63) / 64 tokens from l times that on; word = bits | positions << 16.  Returns the
   header word: the token count, or DXL_RUN_NONE when a group does not fit (positions > 65535, a pass > RUN_PASSBITS). */
static uint32_t run_groups(const uint16_t *tb, const uint32_t *ts, uint32_t cnt, uint32_t L, uint32_t *grp)
{ uint32_t k0, none = 0;
  if (cnt > dxl_tok_limit(L)) none = 1;    /* more tokens than the encoder's token slots hold: such a line never has an
                                                             index (k_qv_decode_runs refuses one), the lane-per-line kernel takes it */
  for (k0 = 0; k0 < cnt; k0 += DXL_RUN_PASS, grp += 64)
    { const uint32_t m = cnt - k0 < DXL_RUN_PASS ? cnt - k0 : DXL_RUN_PASS, T = (m + 63u) >> 6;
      uint32_t lane, total = 0;
      for (lane = 0; lane < 64; lane++)
        { const uint32_t first = lane * T, c = first < m ? (m - first < T ? m - first : T) : 0u;
          uint32_t nb = 0, span = 0, k;
          for (k = 0; k < c; k++) { nb += tb[k0 + first + k]; span += ts[k0 + first + k]; }
          if (span > 0xffffu || nb > 0xffffu) none = 1;
          grp[lane] = nb | (span << 16);
          total += nb;
        }
      if (total > DXL_RUN_PASSBITS) none = 1;
    }
  return none ? DXL_RUN_NONE : cnt;
}

void dx_qv_index_free(dx_qv_index *x)
{ if (x == NULL) return;
  free(x->rec_off); free(x->hdr_off); free(x->seg); free(x->len); free(x->hdr4); free(x->prefix);
  free(x->gidx); free(x->gidx_off);
  memset(x, 0, sizeof(*x));
}

/* everything a walk needs besides the image */
typedef struct
  { wlut  *lut[6];
    mwlut *mlut[4];
    rwlut *rlut[2];
    const dx_qv_coding *cd;
    int    newv, flip;
    int    want_index;                /* leave the group index too (dx_qv_walk_indexed) */
  } walk_tabs;

typedef struct { uint32_t hdr_bytes, len, seg[5]; int32_t dwell, beg, end, qv; uint64_t gx_at, gx_words; uint32_t gx_none; } walk_rec;

/* index words of the records a thread walks (in the order it walks them) + its token scratch */
typedef struct { uint32_t *w; uint64_t n, cap; uint16_t *tb; uint32_t *ts; uint32_t tcap; } gx_buf;

static uint32_t *gx_room(gx_buf *g, uint64_t words)        /* `words` more zeroed words; NULL: out of memory */
{ if (g->n + words > g->cap)
    { uint64_t nc = g->cap ? g->cap : (1u << 16);
      uint32_t *q;
      while (nc < g->n + words) nc *= 2;
      q = realloc(g->w, nc * sizeof(*q));
      if (q == NULL) return NULL;
      g->w = q; g->cap = nc;
    }
  memset(g->w + g->n, 0, words * sizeof(uint32_t));
  g->n += words;
  return g->w + g->n - words;
}

/* One record at img + at (undexqv.c:119-208): the framing fields, then the five segments walked code by
   code.  Returns the offset behind the record; 0 if there is no well-formed record here.            */
static size_t walk_record_ix(const walk_tabs *t, const uint8_t *img, size_t n, size_t at, walk_rec *r, gx_buf *g);

/* the framing fields of a record (undexqv.c:119-180); returns the offset of its first segment, 0: no plausible record here */
static size_t walk_framing(const walk_tabs *t, const uint8_t *img, size_t n, size_t at, walk_rec *r)
{ const size_t h0 = at;
  int32_t  beg, end_, qv, dw = 0;
  uint32_t rlen;

  while (at < n && img[at] == 255) { dw += 255; at += 1; }
  if (at >= n) return 0;
  dw += img[at++];
  if (t->newv)
    { if (at + 12 > n) return 0;
      memcpy(&beg, img + at, 4); memcpy(&end_, img + at + 4, 4); memcpy(&qv, img + at + 8, 4);
      if (t->flip)                                        /* undexqv.c:140-148 */
        { beg = (int32_t) flip32((uint32_t) beg); end_ = (int32_t) flip32((uint32_t) end_); qv = (int32_t) flip32((uint32_t) qv); }
      at += 12;
    }
  else
    { uint16_t h[3];
      if (at + 6 > n) return 0;
      memcpy(h, img + at, 6);
      if (t->flip) { h[0] = flip16(h[0]); h[1] = flip16(h[1]); h[2] = flip16(h[2]); }
      beg = h[0]; end_ = h[1]; qv = h[2];
      at += 6;
    }
  if (end_ < beg || (int64_t) end_ - (int64_t) beg > 0x7fffffff) return 0;
  rlen = (uint32_t) ((int64_t) end_ - (int64_t) beg);
  if ((uint64_t) rlen > 65536u * 8u * (uint64_t) (n - at) + 64u)      /* a token has >= 1 bit and covers <= 65536 symbols */
    return 0;
  r->hdr_bytes = (uint32_t) (at - h0);
  r->len = rlen; r->dwell = dw; r->beg = beg; r->end = end_; r->qv = qv;
  return at;
}

static size_t walk_record(const walk_tabs *t, const uint8_t *img, size_t n, size_t at, walk_rec *r)
{ const uint8_t *end = img + n;
  const dx_qv_coding *cd = t->cd;
  uint32_t rlen, clen;
  int64_t  b;

  at = walk_framing(t, img, n, at, r);
  if (at == 0) return 0;
  rlen = r->len;

  clen = rlen;                                            /* QV.c:1433-1462 */
  if (cd->delChar < 0)
    b = walk_plain(img + at, end, rlen, t->lut[DX_DEL], t->mlut[DX_DEL], cd->s[DX_DEL].type == 2, t->flip);
  else
    b = walk_runs(img + at, end, rlen, t->lut[DX_DEL], cd->s[DX_DEL].type == 2, t->lut[DX_DRUN], t->rlut[0], &clen, t->flip);
  if (b < 0) return 0;
  r->seg[0] = (uint32_t) b; at += (size_t) b;
  r->seg[1] = (clen + 3) >> 2;
  if (at + r->seg[1] > n) return 0;
  at += r->seg[1];
  b = walk_plain(img + at, end, rlen, t->lut[DX_INS], t->mlut[DX_INS], cd->s[DX_INS].type == 2, t->flip);   /* QV.c:1464 */
  if (b < 0) return 0;
  r->seg[2] = (uint32_t) b; at += (size_t) b;
  b = walk_plain(img + at, end, rlen, t->lut[DX_MRG], t->mlut[DX_MRG], cd->s[DX_MRG].type == 2, t->flip);   /* QV.c:1467 */
  if (b < 0) return 0;
  r->seg[3] = (uint32_t) b; at += (size_t) b;
  if (cd->subChar < 0)                                                          /* QV.c:1470-1478 */
    b = walk_plain(img + at, end, rlen, t->lut[DX_SUB], t->mlut[DX_SUB], cd->s[DX_SUB].type == 2, t->flip);
  else
    { uint32_t nn;
      b = walk_runs(img + at, end, rlen, t->lut[DX_SUB], cd->s[DX_SUB].type == 2, t->lut[DX_SRUN], t->rlut[1], &nn, t->flip);
    }
  if (b < 0) return 0;
  r->seg[4] = (uint32_t) b; at += (size_t) b;
  return at;
}

/* the same record, with its share of the group index appended to g (walk_tabs.want_index); on failure g is as it was */
static size_t walk_record_ix(const walk_tabs *t, const uint8_t *img, size_t n, size_t at, walk_rec *r, gx_buf *g)
{ const uint8_t *end = img + n;
  const dx_qv_coding *cd = t->cd;
  const uint64_t at0_words = g->n;
  uint32_t rlen, clen, nd = 0, ns = 0, pd, sw, rb;
  uint32_t *blk;
  int64_t  b;

  at = walk_framing(t, img, n, at, r);
  if (at == 0) return 0;
  rlen = r->len;
  if (rlen > g->tcap)
    { uint16_t *tb = realloc(g->tb, ((size_t) rlen + 1) * sizeof(*tb));
      uint32_t *ts;
      if (tb == NULL) return 0;
      g->tb = tb;
      ts = realloc(g->ts, ((size_t) rlen + 1) * sizeof(*ts));
      if (ts == NULL) return 0;
      g->ts = ts; g->tcap = rlen;
    }
  sw = dxl_sub_words(rlen); rb = dxl_run_base(rlen);
  if (gx_room(g, (uint64_t) rb + 3u) == NULL) return 0;
  r->gx_at = at0_words; r->gx_none = 0;
#define BLK (g->w + at0_words)
#define FAIL { g->n = at0_words; return 0; }
  BLK[rb + 0] = DXL_RUN_NONE; BLK[rb + 1] = DXL_RUN_NONE;
  clen = rlen;
  if (cd->delChar < 0)
    b = walk_plain_ix(img + at, end, rlen, t->lut[DX_DEL], t->mlut[DX_DEL], cd->s[DX_DEL].type == 2, t->flip, (uint8_t *) BLK);
  else
    { b = walk_runs_ix(img + at, end, rlen, t->lut[DX_DEL], cd->s[DX_DEL].type == 2, t->lut[DX_DRUN], t->rlut[0], &clen, t->flip, g->tb, g->ts);
      nd = clen;
    }
  if (b < 0) FAIL
  r->seg[0] = (uint32_t) b; at += (size_t) b;
  pd = cd->delChar < 0 ? 0u : dxl_run_passes(nd);
  if (pd)
    { if ((blk = gx_room(g, 64ull * pd)) == NULL) FAIL
      BLK[rb + 0] = run_groups(g->tb, g->ts, nd, rlen, blk);
    }
  else if (cd->delChar >= 0)
    BLK[rb + 0] = 0;                                      /* a line of run characters only: no tokens, indexed */
  BLK[rb + 2] = pd;
  if (cd->delChar >= 0 && BLK[rb + 0] == DXL_RUN_NONE) r->gx_none += 1;
  r->seg[1] = (clen + 3) >> 2;
  if (at + r->seg[1] > n) FAIL
  at += r->seg[1];
  b = walk_plain_ix(img + at, end, rlen, t->lut[DX_INS], t->mlut[DX_INS], cd->s[DX_INS].type == 2, t->flip, (uint8_t *) (BLK + sw));
  if (b < 0) FAIL
  r->seg[2] = (uint32_t) b; at += (size_t) b;
  b = walk_plain_ix(img + at, end, rlen, t->lut[DX_MRG], t->mlut[DX_MRG], cd->s[DX_MRG].type == 2, t->flip, (uint8_t *) (BLK + 2 * sw));
  if (b < 0) FAIL
  r->seg[3] = (uint32_t) b; at += (size_t) b;
  if (cd->subChar < 0)
    b = walk_plain_ix(img + at, end, rlen, t->lut[DX_SUB], t->mlut[DX_SUB], cd->s[DX_SUB].type == 2, t->flip, (uint8_t *) (BLK + 3 * sw));
  else
    { b = walk_runs_ix(img + at, end, rlen, t->lut[DX_SUB], cd->s[DX_SUB].type == 2, t->lut[DX_SRUN], t->rlut[1], &ns, t->flip, g->tb, g->ts);
      if (b >= 0)
        { const uint32_t ps = dxl_run_passes(ns);
          if (ps)
            { if ((blk = gx_room(g, 64ull * ps)) == NULL) FAIL
              BLK[rb + 1] = run_groups(g->tb, g->ts, ns, rlen, blk);
            }
          else
            BLK[rb + 1] = 0;
          if (BLK[rb + 1] == DXL_RUN_NONE) r->gx_none += 1;
        }
    }
  if (b < 0) FAIL
  r->seg[4] = (uint32_t) b; at += (size_t) b;
#undef FAIL
#undef BLK
  r->gx_words = g->n - at0_words;
  return at;
}

/* records of img[from, to) appended to a growing list; stops at `to` exactly (returns it), behind it
   (a record straddles `to`: returns that offset) or 0 on a malformed record / out of memory (*rc says which) */
typedef struct { walk_rec *r; uint64_t n, cap; gx_buf gx; } rec_list;

static size_t walk_span(const walk_tabs *t, const uint8_t *img, size_t n, size_t from, size_t to, rec_list *L, int *rc)
{ size_t at = from;
  while (at < to)
    { size_t nx;
      if (L->n == L->cap)
        { uint64_t  nc = L->cap ? 2 * L->cap : 1024;
          walk_rec *q  = realloc(L->r, nc * sizeof(*q));
          if (q == NULL) { *rc = DX_E_NOMEM; return 0; }
          L->r = q; L->cap = nc;
        }
      nx = t->want_index ? walk_record_ix(t, img, n, at, &L->r[L->n], &L->gx) : walk_record(t, img, n, at, &L->r[L->n]);
      if (nx == 0) { *rc = DX_E_FORMAT; return 0; }
      L->n += 1;
      at = nx;
    }
  return at;
}

/* ---- the walk on several host threads ---------------------------------------------------------
 * Where a record starts is only known by walking from the file's first record -- but a guessed start can be
 * CHECKED: the image is cut into pieces, a thread per piece looks for the first offset in its piece at which
 * a plausible record header stands (0 <= beg <= end, a sane quality value) AND from which two consecutive
 * records walk cleanly, and then walks from there to the start the next thread found.  Arriving there
 * exactly proves both guesses (a walk from a wrong offset does not re-synchronise onto record boundaries:
 * framing fields and pad words are not self-delimiting); any thread that overshoots its neighbour's start
 * condemns the attempt, and the file is walked front to back as before.  Results are identical by
 * construction: only offsets verified by an unbroken chain of walks from the first record are kept.   */
typedef struct
  { const walk_tabs *t;
    const uint8_t   *img;
    size_t           n, lo, hi;       /* piece [lo, hi) */
    size_t           start, stop;     /* where this thread's records begin / must end */
    size_t           landed;
    rec_list         L;
    int              rc;
  } walk_job;

static int header_plausible(const walk_tabs *t, const uint8_t *img, size_t n, size_t at)
{ int32_t beg, end_, qv;
  int k = 0;
  while (at < n && img[at] == 255 && k < 16) { at += 1; k += 1; }
  if (at + 13 > n) return 0;
  at += 1;
  memcpy(&beg, img + at, 4); memcpy(&end_, img + at + 4, 4); memcpy(&qv, img + at + 8, 4);
  if (t->flip)
    { beg = (int32_t) flip32((uint32_t) beg); end_ = (int32_t) flip32((uint32_t) end_); qv = (int32_t) flip32((uint32_t) qv); }
  /* A guess that passes costs a walk of its (garbage) length: entries beyond 4 M symbols are left to be
     reached by the neighbouring thread's walk rather than guessed at (1 in ~10^7 offsets passes by chance). */
  return beg >= 0 && beg < (1 << 28) && end_ >= beg && end_ - beg <= (1 << 22) && qv >= 0 && qv < 1000000 &&
         (uint64_t) (end_ - beg) <= 8u * (uint64_t) (n - at);
}

static void *walk_find(void *arg)                       /* first checked record start in the piece (0: none) */
{ walk_job *j = arg;
  size_t p;
  j->start = 0;
  for (p = j->lo; p < j->hi; p++)
    if (header_plausible(j->t, j->img, j->n, p))
      { walk_rec r;
        size_t a = walk_record(j->t, j->img, j->n, p, &r), b;
        if (a == 0) continue;
        if (a == j->n) { j->start = p; break; }         /* the file's last record */
        if (!header_plausible(j->t, j->img, j->n, a)) continue;
        b = walk_record(j->t, j->img, j->n, a, &r);
        if (b == 0) continue;
        j->start = p;
        break;
      }
  return NULL;
}

static void *walk_piece(void *arg)
{ walk_job *j = arg;
  j->rc = DX_OK;
  j->landed = walk_span(j->t, j->img, j->n, j->start, j->stop, &j->L, &j->rc);
  return NULL;
}

#include <pthread.h>
#include <unistd.h>

#define WALK_PIECE_MIN ((size_t) 2 << 20)               /* bytes of image a thread should at least have */
#define WALK_THREADS_MAX 64

/* records of img[first, n) into *out on up to `threads` threads; DX_E_MISMATCH: the guesses did not chain up */
#include <time.h>
static void walk_mark(const char *what)                  /* DEXGPU_TIMING=1: where the walk's time goes */
{ static double t0 = 0;
  struct timespec ts;
  double now;
  if (getenv("DEXGPU_TIMING") == NULL) return;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  now = (double) ts.tv_sec * 1e3 + (double) ts.tv_nsec / 1e6;
  if (what == NULL) { t0 = now; return; }
  fprintf(stderr, "[walk %8.1f ms] %s\n", now - t0, what);
}

static int walk_parallel(const walk_tabs *t, const uint8_t *img, size_t n, size_t first, int threads, rec_list *out)
{ walk_job  job[WALK_THREADS_MAX];
  pthread_t th[WALK_THREADS_MAX];
  int       T = threads, k, m, rc = DX_OK, made;
  size_t    piece;
  if (T > WALK_THREADS_MAX) T = WALK_THREADS_MAX;
  if ((size_t) T > (n - first) / WALK_PIECE_MIN) T = (int) ((n - first) / WALK_PIECE_MIN);
  if (T < 2) return DX_E_MISMATCH;
  piece = (n - first) / (size_t) T;
  memset(job, 0, sizeof(job));
  for (k = 0; k < T; k++)
    { job[k].t = t; job[k].img = img; job[k].n = n;
      job[k].lo = first + (size_t) k * piece;
      job[k].hi = k == T - 1 ? n : first + (size_t) (k + 1) * piece;
    }
  job[0].start = first;
  made = 0;                                             /* guesses: pieces 1 .. T-1 */
  for (k = 1; k < T; k++)
    { if (pthread_create(&th[k], NULL, walk_find, &job[k]) != 0) break;
      made = k;
    }
  for (k = 1; k <= made; k++) pthread_join(th[k], NULL);
  walk_mark("record starts guessed and checked");
  if (made < T - 1) return DX_E_MISMATCH;
  m = 0;                                                /* pieces without a start are walked by their predecessor */
  for (k = 1; k < T; k++)
    if (job[k].start != 0)
      { job[m].stop = job[k].start;
        m += 1;
        if (m != k) job[m] = job[k];
      }
  job[m].stop = n;
  T = m + 1;
  made = -1;
  for (k = 0; k < T; k++)
    { if (pthread_create(&th[k], NULL, walk_piece, &job[k]) != 0) break;
      made = k;
    }
  for (k = 0; k <= made; k++) pthread_join(th[k], NULL);
  walk_mark("pieces walked");
  if (made < T - 1) rc = DX_E_MISMATCH;
  for (k = 0; k < T && rc == DX_OK; k++)
    if (job[k].rc == DX_E_NOMEM) rc = DX_E_NOMEM;
    else if (job[k].rc != DX_OK || job[k].landed != job[k].stop) rc = DX_E_MISMATCH;   /* a wrong guess (or a damaged file): walk it front to back */
  if (rc == DX_OK)
    { uint64_t tot = 0, at = 0, gw = 0, gat = 0, i;
      for (k = 0; k < T; k++) { tot += job[k].L.n; gw += job[k].L.gx.n; }
      out->r = malloc((tot + 1) * sizeof(walk_rec));
      if (t->want_index) out->gx.w = malloc((gw + 1) * sizeof(uint32_t));
      if (out->r == NULL || (t->want_index && out->gx.w == NULL)) rc = DX_E_NOMEM;
      else
        { for (k = 0; k < T; k++)
            { memcpy(out->r + at, job[k].L.r, job[k].L.n * sizeof(walk_rec));
              if (t->want_index)
                { memcpy(out->gx.w + gat, job[k].L.gx.w, job[k].L.gx.n * sizeof(uint32_t));
                  for (i = 0; i < job[k].L.n; i++) out->r[at + i].gx_at += gat;    /* (offsets were into the thread's own words) */
                  gat += job[k].L.gx.n;
                }
              at += job[k].L.n;
            }
          out->n = out->cap = tot;
          out->gx.n = out->gx.cap = gw;
        }
    }
  for (k = 0; k < T; k++) { free(job[k].L.r); free(job[k].L.gx.w); free(job[k].L.gx.tb); free(job[k].L.gx.ts); }
  return rc;
}

int dx_qv_walk(const uint8_t *img, size_t n, dx_qv_index *x) { return dx_qv_walk_indexed(img, n, x, 0); }

int dx_qv_walk_indexed(const uint8_t *img, size_t n, dx_qv_index *x, int want_index)
{ walk_tabs t;
  rec_list  L;
  size_t    at = 0, used = 0;
  uint64_t  hat = 0, i;
  uint16_t  key;
  int       rc = DX_OK, well = 0, s, threads;

  if (img == NULL || x == NULL) return DX_E_ARG;
  memset(x, 0, sizeof(*x));
  memset(&t, 0, sizeof(t));
  memset(&L, 0, sizeof(L));
  t.want_index = want_index != 0;
  if (n < 2) return DX_E_FORMAT;
  memcpy(&key, img, 2);                                   /* undexqv.c:103-110 */
  if (key == 0x55aa || key == 0xaa55) { x->newv = 1; at = 2; }
  { uint16_t k2 = 0;                                      /* prefix length first (QV.c:1222-1256): key, two run chars, int32 */
    uint32_t pl = 0;
    if (n - at < 10) return DX_E_FORMAT;
    memcpy(&k2, img + at, 2);
    memcpy(&pl, img + at + 6, 4);
    if (k2 != 0x33cc) pl = flip32(pl);
    if ((uint64_t) pl > (uint64_t) (n - at - 10)) return DX_E_FORMAT;
    x->prefix = malloc((size_t) pl + 1);
    if (x->prefix == NULL) return DX_E_NOMEM;
    rc = dx_qv_read_coding(img + at, n - at, &x->coding, &x->flip, x->prefix, (size_t) pl + 1, &used);
  }
  if (rc != DX_OK) goto fail;
  at += used;
  t.cd = &x->coding; t.newv = x->newv; t.flip = x->flip;

  for (s = 0; s < 6; s++)
    { if ((s == DX_DRUN && x->coding.delChar < 0) || (s == DX_SRUN && x->coding.subChar < 0)) continue;
      t.lut[s] = malloc(sizeof(wlut));
      if (t.lut[s] == NULL) { rc = DX_E_NOMEM; goto fail; }
      build_wlut(&x->coding.s[s], t.lut[s]);
      if (s < 4)
        { t.mlut[s] = malloc(sizeof(mwlut));
          if (t.mlut[s] == NULL) { rc = DX_E_NOMEM; goto fail; }
          build_mwlut(t.lut[s], x->coding.s[s].type == 2, t.mlut[s]);
        }
    }
  for (s = 0; s < 2; s++)
    { const int sym = s ? DX_SUB : DX_DEL, run = s ? DX_SRUN : DX_DRUN;
      if (t.lut[run] == NULL) continue;
      t.rlut[s] = malloc(sizeof(rwlut));
      if (t.rlut[s] == NULL) { rc = DX_E_NOMEM; goto fail; }
      build_rwlut(t.lut[run], t.lut[sym], x->coding.s[sym].type == 2, t.rlut[s]);
    }

  /* the records: on several threads when the image is large (32-bit framing fields only: the older
     16-bit ones are too easily plausible), else -- and whenever the guesses do not chain up -- front to back */
  { const char *e = getenv("DEXGPU_WALK_THREADS");
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    threads = e ? atoi(e) : (int) (cores > 32 ? 32 : cores);
  }
  walk_mark(NULL);
  rc = DX_E_MISMATCH;
  if (threads > 1 && x->newv && n - at >= 4 * WALK_PIECE_MIN)
    rc = walk_parallel(&t, img, n, at, threads, &L);
  if (rc == DX_E_MISMATCH && dx_test_on("walk_require_parallel"))
    goto fail;                                            /* (tests: no silent front-to-back walk) */
  if (rc == DX_E_MISMATCH)
    { free(L.r); free(L.gx.w); free(L.gx.tb); free(L.gx.ts);
      memset(&L, 0, sizeof(L));
      rc = DX_OK;
      if (walk_span(&t, img, n, at, n, &L, &rc) == 0 && rc == DX_OK && at < n) rc = DX_E_FORMAT;
    }
  if (rc != DX_OK) goto fail;

  x->n       = L.n;
  x->rec_off = malloc((L.n + 1) * sizeof(uint64_t));
  x->hdr_off = malloc((L.n + 1) * sizeof(uint64_t));
  x->seg     = malloc((L.n + 1) * 5 * sizeof(uint32_t));
  x->len     = malloc((L.n + 1) * sizeof(uint32_t));
  x->hdr4    = malloc((L.n + 1) * 4 * sizeof(int32_t));
  if (!x->rec_off || !x->hdr_off || !x->seg || !x->len || !x->hdr4) { rc = DX_E_NOMEM; goto fail; }
  if (t.want_index)                                       /* the group index: the threads' words are already in record order */
    { x->gidx_off = malloc((L.n + 1) * sizeof(uint64_t));
      if (!x->gidx_off) { rc = DX_E_NOMEM; goto fail; }
      x->gidx = L.gx.w; L.gx.w = NULL;
      for (i = 0; i < L.n; i++)
        { x->gidx_off[i] = L.r[i].gx_at;
          x->gidx_none  += L.r[i].gx_none;
        }
      x->gidx_off[L.n] = L.gx.n;
      x->gidx_words    = L.gx.n;
    }
  for (i = 0; i < L.n; i++)
    { const walk_rec *r = &L.r[i];
      x->rec_off[i] = at;
      x->hdr_off[i] = hat;
      hat  += r->hdr_bytes;
      well += r->dwell;                                   /* undexqv.c:124-133: wells are a running sum */
      x->len[i] = r->len;
      x->hdr4[4*i] = well; x->hdr4[4*i+1] = r->beg; x->hdr4[4*i+2] = r->end; x->hdr4[4*i+3] = r->qv;
      memcpy(x->seg + 5*i, r->seg, sizeof(r->seg));
      at += (size_t) r->hdr_bytes + r->seg[0] + r->seg[1] + r->seg[2] + r->seg[3] + r->seg[4];
    }
  x->rec_off[L.n] = at;
  x->hdr_off[L.n] = hat;
  walk_mark("index assembled");
  free(L.r); free(L.gx.w); free(L.gx.tb); free(L.gx.ts);
  for (s = 0; s < 6; s++) free(t.lut[s]);
  for (s = 0; s < 4; s++) free(t.mlut[s]);
  free(t.rlut[0]); free(t.rlut[1]);
  return DX_OK;

fail:
  free(L.r); free(L.gx.w); free(L.gx.tb); free(L.gx.ts);
  for (s = 0; s < 6; s++) free(t.lut[s]);
  for (s = 0; s < 4; s++) free(t.mlut[s]);
  free(t.rlut[0]); free(t.rlut[1]);
  dx_qv_index_free(x);
  return rc;
}

/* Header lines gathered by the GPU text front end (dx_index_quiva_device): blob holds line i at
 * [pos[i], pos[i+1]) including its newline.  The same checks as dx_index_quiva / QV.c:958-968. */
int dx_parse_quiva_headers(const uint8_t *blob, const uint64_t *pos, uint64_t n, int32_t *hdr4,
                           size_t *prefix_len, uint64_t *bad_entry)
{ uint64_t i;
  for (i = 0; i < n; i++)
    { const uint8_t *h = blob + pos[i], *slash;
      size_t hl = (size_t) (pos[i+1] - pos[i]) - 1;
      int32_t f[4];
      slash = hl > 1 ? memchr(h + 1, '/', hl - 1) : NULL;
      if (slash == NULL || scan_tail(slash + 1, hl - (size_t) (slash + 1 - h), "%d/%d_%d RQ=0.%d\n", f, NULL) != 4)
        { if (bad_entry) *bad_entry = i;
          return DX_E_FORMAT;
        }
      memcpy(hdr4 + 4*i, f, sizeof(f));
      if (i == 0 && prefix_len) *prefix_len = (size_t) (slash - h);
    }
  return DX_OK;
}

/* Header lines of .fasta / .arrow records gathered by dx_index_seq_device: the checks and field
 * conversions of dexta.c:118-157 / dexar.c:118-163.                                            */
int dx_parse_seq_headers(int arrow, const uint8_t *blob, const uint64_t *pos, uint64_t n,
                         int32_t *hdr4, uint16_t *cnr4, size_t *prefix_len, uint64_t *bad_entry)
{ uint64_t i;
  for (i = 0; i < n; i++)
    { const uint8_t *h = blob + pos[i], *slash;
      size_t  hl = (size_t) (pos[i+1] - pos[i]) - 1;
      int32_t f[4];
      float   snr[4];
      int     x, j;
      if (i == 0)
        { slash = memchr(h, '/', hl);                           /* dexta.c:118 */
          if (slash == NULL) goto bad;
          if (prefix_len) *prefix_len = (size_t) (slash - h);
        }
      slash = hl > 1 ? memchr(h + 1, '/', hl - 1) : NULL;       /* dexta.c:146 */
      if (slash == NULL) goto bad;
      if (arrow)
        { x = scan_tail(slash + 1, hl - (size_t) (slash + 1 - h), "%d/%d_%d SN=%f,%f,%f,%f\n", f, snr);
          if (x != 7) goto bad;
          for (j = 0; j < 4; j++) cnr4[4*i + j] = dx_snr_to_cnr(snr[j]);
        }
      else
        { x = scan_tail(slash + 1, hl - (size_t) (slash + 1 - h), "%d/%d_%d RQ=0.%d\n", f, NULL);
          if (x < 3) goto bad;
          for (j = 0; j < 4; j++) cnr4[4*i + j] = 0;
        }
      memcpy(hdr4 + 4*i, f, sizeof(f));
      continue;
    bad:
      if (bad_entry) *bad_entry = i;
      return DX_E_FORMAT;
    }
  return DX_OK;
}
