/*
 * dexref.c -- CPU restatement (ORACLE) of DEXTRACTOR's dexta/dexar/dexqv codecs.
 *
 * TEST INFRASTRUCTURE ONLY: see dexref.h.  Every function cites the reference file:line whose
 * behaviour it restates.  Written from the behaviour of the reference (SURVEY.md section 8a /
 * Appendix A), operating on memory images instead of FILE* streams.
 *
 * Parity status: pinned against the compiled reference (oracle/_ref) -- see
 * tests/test_oracle_golden.py and tests/golden/make_golden.py.
 */
#include "dexref.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LINE_LIMIT 100000          /* MAX_BUFFER, dexta.c:21: fgets(...,MAX_BUFFER,...) */

/* =========================================================================================
 *  byte sink
 * ========================================================================================= */

typedef struct { uint8_t *p; size_t len, cap; int over; } sink_t;

static void sink_init(sink_t *s, uint8_t *p, size_t cap) { s->p = p; s->len = 0; s->cap = cap; s->over = 0; }

static void sink_put(sink_t *s, const void *src, size_t n)
{ if (s->len + n > s->cap) { s->over = 1; s->len += n; return; }
  if (!s->over) memcpy(s->p + s->len, src, n);
  s->len += n;
}

static void sink_u8 (sink_t *s, uint8_t v)  { sink_put(s, &v, 1); }
static void sink_u16(sink_t *s, uint16_t v) { sink_put(s, &v, 2); }
static void sink_i32(sink_t *s, int32_t v)  { sink_put(s, &v, 4); }
static void sink_u32(sink_t *s, uint32_t v) { sink_put(s, &v, 4); }

/* =========================================================================================
 *  2-bit packer (DB.c:319-441)
 * ========================================================================================= */

/* DB.c:393-416: table is 0 everywhere except c/C=1 g/G=2 t/T=3.  The reference indexes a
 * 128-entry table with a signed char (UB for bytes >= 128); the restatement maps those to 0. */
void ref_number_read(uint8_t *s, size_t n)
{ size_t i;
  for (i = 0; i < n; i++)
    switch (s[i])
      { case 'c': case 'C': s[i] = 1; break;
        case 'g': case 'G': s[i] = 2; break;
        case 't': case 'T': s[i] = 3; break;
        default:            s[i] = 0; break;
      }
}

/* DB.c:418-441: table is 3 everywhere except '1'=0 '2'=1 '3'=2 and the quirk 'G'(71)=2. */
void ref_number_arrow(uint8_t *s, size_t n)
{ size_t i;
  for (i = 0; i < n; i++)
    switch (s[i])
      { case '1': s[i] = 0; break;
        case '2': s[i] = 1; break;
        case '3': case 'G': s[i] = 2; break;
        default:  s[i] = 3; break;
      }
}

/* DB.c:319-338: four symbols per byte, first symbol in the top two bits; symbols past len
 * count as 0 (the reference zeroes s[len..len+2] around the loop).  Like the reference's
 * `(char)((s0<<6)|(s1<<4)|(s2<<2)|s3)` the OR is taken on the full values and truncated. */
size_t ref_compress_read(size_t len, const uint8_t *num, uint8_t *out)
{ size_t i, j = 0;
  for (i = 0; i < len; i += 4)
    { unsigned a = num[i];
      unsigned b = (i+1 < len) ? num[i+1] : 0;
      unsigned c = (i+2 < len) ? num[i+2] : 0;
      unsigned d = (i+3 < len) ? num[i+3] : 0;
      out[j++] = (uint8_t) ((a << 6) | (b << 4) | (c << 2) | d);
    }
  return j;
}

/* DB.c:342-363 */
void ref_uncompress_read(size_t len, const uint8_t *in, uint8_t *num)
{ size_t i;
  for (i = 0; i < len; i++)
    num[i] = (uint8_t) ((in[i >> 2] >> (6 - 2*(i & 3))) & 3);
}

/* =========================================================================================
 *  line reader over a memory image (stands in for fgets)
 * ========================================================================================= */

typedef struct { const uint8_t *p; size_t n, pos; } text_t;

/* Next line [*beg, *beg+*len) without its '\n'.  Returns 1 ok, 0 eof, -1 last line lacks '\n'. */
static int next_line(text_t *t, const uint8_t **beg, size_t *len)
{ const uint8_t *nl;
  if (t->pos >= t->n) return 0;
  nl = memchr(t->p + t->pos, '\n', t->n - t->pos);
  if (nl == NULL) return -1;
  *beg = t->p + t->pos;
  *len = (size_t) (nl - *beg);
  t->pos += *len + 1;
  return 1;
}

static const uint8_t *find_byte(const uint8_t *s, size_t n, int c) { return n ? memchr(s, c, n) : NULL; }

/* sscanf over a header tail that is not NUL terminated: copy the tail into a bounded buffer */
static int scan_header(const uint8_t *s, size_t n, const char *fmt, void *a, void *b, void *c,
                       void *d, void *e, void *f, void *g)
{ char tmp[512];
  if (n > sizeof(tmp) - 2) n = sizeof(tmp) - 2;
  memcpy(tmp, s, n);
  tmp[n] = '\n';
  tmp[n+1] = '\0';
  return sscanf(tmp, fmt, a, b, c, d, e, f, g);
}

/* well delta bytes, dexta.c:187-194 / dexar.c:193-200 / dexqv.c:128-135 */
static void put_well(sink_t *o, int well, int *lwell)
{ while (well - *lwell >= 255)
    { sink_u8(o, 0xff);
      *lwell += 255;
    }
  sink_u8(o, (uint8_t) (well - *lwell));
  *lwell = well;
}

/* =========================================================================================
 *  dexta / dexar  (dexta.c:104-205, dexar.c:103-211)
 * ========================================================================================= */

static long pack2_file(const uint8_t *txt, size_t n, uint8_t *out, size_t cap, int arrow)
{ text_t   t = { txt, n, 0 };
  sink_t   o;
  const uint8_t *line, *slash;
  size_t   llen;
  uint8_t *seq = NULL, *pk = NULL;
  size_t   smax = 0;
  int      r, lwell = 0, have_hdr, entries = 0, first_get;
  long     ret;

  sink_init(&o, out, cap);

  r = next_line(&t, &line, &llen);                       /* dexta.c:108-116 */
  if (r <= 0 || llen + 1 >= LINE_LIMIT) return REF_E_FORMAT;
  if (llen == 0 || line[0] != '>') return REF_E_FORMAT;
  slash = find_byte(line, llen, '/');                    /* dexta.c:118: index(read,'/') */
  if (slash == NULL) return REF_E_FORMAT;

  sink_u16(&o, 0x55aa);                                  /* dexta.c:124-129 */
  sink_i32(&o, (int32_t) (slash - line));
  sink_put(&o, line, (size_t) (slash - line));

  have_hdr = 1;
  while (have_hdr)                                       /* dexta.c:139 */
    { int     well, beg, end, qv = 0, x;
      float   snr[4];
      size_t  rlen = 0, clen;

      slash = (llen > 1) ? find_byte(line + 1, llen - 1, '/') : NULL;   /* dexta.c:146 */
      if (slash == NULL) { ret = REF_E_FORMAT; goto done; }
      if (arrow)
        { x = scan_header(slash + 1, llen - (size_t) (slash + 1 - line), "%d/%d_%d SN=%f,%f,%f,%f\n",
                          &well, &beg, &end, snr, snr+1, snr+2, snr+3);                /* dexar.c:152 */
          if (x != 7) { ret = REF_E_FORMAT; goto done; }
        }
      else
        { x = scan_header(slash + 1, llen - (size_t) (slash + 1 - line), "%d/%d_%d RQ=0.%d\n",
                          &well, &beg, &end, &qv, NULL, NULL, NULL);                   /* dexta.c:151 */
          if (x < 3) { ret = REF_E_FORMAT; goto done; }
          if (x == 3) qv = 0;
        }

      have_hdr = 0;                                      /* dexta.c:161-183: gather sequence lines */
      entries += 1;
      first_get = 1;
      while (1)
        { r = next_line(&t, &line, &llen);
          if (r < 0 || (r > 0 && llen + 1 >= LINE_LIMIT)) { ret = REF_E_FORMAT; goto done; }
          /* The end of the file right behind a header that is not the file's first: the reference's fgets returns NULL and
             leaves in its buffer what the header's parse left there, in which dexta.c:166-167 finds no newline where it looks
             for one: "Line %d: Fasta line is too long", exit 1 (observed with the compiled reference -- tools/stress_cli.py,
             tests/test_cli.py --; a file of ONE header alone passes, so do empty lines behind a header) */
          if (r == 0 && first_get && entries > 1) { ret = REF_E_FORMAT; goto done; }
          first_get = 0;
          if (r == 0) break;
          if (llen > 0 && line[0] == '>') { have_hdr = 1; break; }
          if (rlen + llen + 4 > smax)
            { smax = (size_t) (1.2 * (double) (rlen + llen)) + 1000;
              seq  = realloc(seq, smax);
              pk   = realloc(pk, smax/4 + 8);
              if (seq == NULL || pk == NULL) { ret = REF_E_SPACE; goto done; }
            }
          memcpy(seq + rlen, line, llen);
          rlen += llen;
        }

      put_well(&o, well, &lwell);                        /* dexta.c:187-198 */
      sink_i32(&o, beg);
      sink_i32(&o, end);
      if (arrow)
        { int k;
          for (k = 0; k < 4; k++)                        /* dexar.c:159-163 */
            { uint16_t cnr;
              if (snr[k] > 99.99)
                cnr = 9999;
              else
                cnr = (uint16_t) ((uint32_t) (snr[k] * 100.));
              sink_u16(&o, cnr);
            }
        }
      else
        sink_i32(&o, qv);

      if (rlen > 0)                                      /* dexta.c:202-204 */
        { if (arrow) ref_number_arrow(seq, rlen); else ref_number_read(seq, rlen);
          clen = ref_compress_read(rlen, seq, pk);
          sink_put(&o, pk, clen);
        }
    }
  ret = o.over ? REF_E_SPACE : (long) o.len;

done:
  free(seq);
  free(pk);
  return ret;
}

long ref_dexta(const uint8_t *fasta, size_t n, uint8_t *out, size_t cap) { return pack2_file(fasta, n, out, cap, 0); }
long ref_dexar(const uint8_t *arrow, size_t n, uint8_t *out, size_t cap) { return pack2_file(arrow, n, out, cap, 1); }

/* =========================================================================================
 *  undexta / undexar  (undexta.c:131-271, undexar.c:129-229)
 * ========================================================================================= */

typedef struct { const uint8_t *p; size_t n, pos; int fail; } bin_t;

static void bin_get(bin_t *b, void *dst, size_t k)
{ if (b->pos + k > b->n) { b->fail = 1; memset(dst, 0, k); b->pos = b->n; return; }
  memcpy(dst, b->p + b->pos, k);
  b->pos += k;
}

static uint16_t swap16(uint16_t v) { return (uint16_t) ((v >> 8) | (v << 8)); }
static uint32_t swap32(uint32_t v)
{ return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

static int32_t get_i32(bin_t *b, int flip) { uint32_t v; bin_get(b, &v, 4); return (int32_t) (flip ? swap32(v) : v); }
static uint16_t get_u16(bin_t *b, int flip) { uint16_t v; bin_get(b, &v, 2); return flip ? swap16(v) : v; }

static void put_text(sink_t *o, const char *fmt, ...);
#include <stdarg.h>
static void put_text(sink_t *o, const char *fmt, ...)
{ char tmp[1200];
  va_list ap;
  int k;
  va_start(ap, fmt);
  k = vsnprintf(tmp, sizeof(tmp), fmt, ap);
  va_end(ap);
  if (k > 0) sink_put(o, tmp, (size_t) k);
}

static long unpack2_file(const uint8_t *img, size_t n, int mode /*0 lower 1 upper 2 arrow*/, int width,
                         uint8_t *out, size_t cap)
{ bin_t    b = { img, n, 0, 0 };
  sink_t   o;
  uint16_t key;
  int      flip, newv, well;
  int32_t  plen;
  char    *name;
  uint8_t *num = NULL;
  size_t   nmax = 0;
  long     ret;
  static const char lower[4] = { 'a','c','g','t' }, upper[4] = { 'A','C','G','T' }, pw[4] = { '1','2','3','4' };
  const char *alpha = (mode == 0) ? lower : (mode == 1 ? upper : pw);

  if (width <= 0) return REF_E_FORMAT;        /* -w0 loops forever in the reference (undexta.c:265) */
  sink_init(&o, out, cap);

  bin_get(&b, &key, 2);                        /* undexta.c:138-159 / undexar.c:136-145 */
  if (b.fail) return REF_E_TRUNC;
  if (key == 0x55aa)      { flip = 0; newv = 1; }
  else if (key == 0xaa55) { flip = 1; newv = 1; }
  else if (mode != 2 && key == 0x33cc) { flip = 0; newv = 0; }
  else if (mode != 2 && key == 0xcc33) { flip = 1; newv = 0; }
  else return REF_E_FORMAT;

  plen = get_i32(&b, flip);                    /* undexta.c:161-169 */
  if (b.fail || plen < 0 || (size_t) plen > n) return REF_E_TRUNC;
  name = malloc((size_t) plen + 1);
  bin_get(&b, name, (size_t) plen);
  name[plen] = '\0';
  if (b.fail) { free(name); return REF_E_TRUNC; }

  well = 0;
  while (b.pos < b.n)                          /* undexta.c:175-271 */
    { uint8_t  byte;
      int      beg, end, qv = 0, rlen, j;
      uint16_t cnr[4] = { 0, 0, 0, 0 };
      size_t   clen;

      bin_get(&b, &byte, 1);
      while (byte == 255)
        { well += 255;
          bin_get(&b, &byte, 1);
          if (b.fail) { ret = REF_E_TRUNC; goto done; }
        }
      well += byte;

      if (newv)
        { beg = get_i32(&b, flip);
          end = get_i32(&b, flip);
          if (mode == 2)
            for (j = 0; j < 4; j++) cnr[j] = get_u16(&b, flip);
          else
            qv = get_i32(&b, flip);
        }
      else
        { beg = get_u16(&b, flip);
          end = get_u16(&b, flip);
          qv  = get_u16(&b, flip);
        }
      if (b.fail) { ret = REF_E_TRUNC; goto done; }

      if (mode == 2)                           /* undexar.c:199-203 */
        { float snr[4];
          for (j = 0; j < 4; j++) snr[j] = (float) (cnr[j] / 100.);
          put_text(&o, "%s/%d/%d_%d SN=%.2f,%.2f,%.2f,%.2f\n", name, well, beg, end,
                   snr[0], snr[1], snr[2], snr[3]);
        }
      else
        put_text(&o, "%s/%d/%d_%d RQ=0.%d\n", name, well, beg, end, qv);   /* undexta.c:242 */

      rlen = end - beg;                        /* undexta.c:247 */
      if (rlen < 0) { ret = REF_E_FORMAT; goto done; }
      clen = ((size_t) rlen + 3) >> 2;
      if (b.pos + clen > b.n) { ret = REF_E_TRUNC; goto done; }
      if ((size_t) rlen + 1 > nmax)
        { nmax = (size_t) rlen + 1024;
          num  = realloc(num, nmax);
          if (num == NULL) { ret = REF_E_SPACE; goto done; }
        }
      ref_uncompress_read((size_t) rlen, b.p + b.pos, num);
      b.pos += clen;
      for (j = 0; j < rlen; j++) num[j] = (uint8_t) alpha[num[j]];

      for (j = 0; j < rlen; j += width)        /* undexta.c:263-270 */
        { int w = (j + width > rlen) ? rlen - j : width;
          sink_put(&o, num + j, (size_t) w);
          sink_u8(&o, '\n');
        }
    }
  ret = o.over ? REF_E_SPACE : (long) o.len;

done:
  free(name);
  free(num);
  return ret;
}

long ref_undexta(const uint8_t *dexta, size_t n, int upper, int width, uint8_t *out, size_t cap)
{ return unpack2_file(dexta, n, upper ? 1 : 0, width, out, cap); }

long ref_undexar(const uint8_t *dexar, size_t n, int width, uint8_t *out, size_t cap)
{ return unpack2_file(dexar, n, 2, width, out, cap); }

/* =========================================================================================
 *  Huffman scheme construction (QV.c:91-220)
 * ========================================================================================= */

typedef struct { int lft, rgt; uint64_t count; } hnode;    /* leaf: rgt < 0, symbol in lft (QV.c:83-86) */

/* QV.c:91-120.  Sift heap[s] down.  Children 2c, 2c+1.  The LEFT child is taken when there is no
 * right child or right.count > left.count (so on a tie the RIGHT child is taken); the child moves
 * up only if it is strictly smaller than the sifted node.                                       */
static void reheap(int s, int *heap, int hsize, const hnode *node)
{ int c = s, l, r, hs = heap[s];
  while ((l = 2*c) <= hsize)
    { int pick;
      r = l + 1;
      if (r > hsize || node[heap[r]].count > node[heap[l]].count)
        pick = l;
      else
        pick = r;
      if (node[hs].count > node[heap[pick]].count)
        { heap[c] = heap[pick];
          c = pick;
        }
      else
        break;
    }
  if (c != s)
    heap[c] = hs;
}

/* QV.c:125-137: left edge appends 0, right edge appends 1. */
static void build_table(const hnode *node, int v, uint32_t code, int len, uint32_t *bits, int32_t *lens)
{ if (node[v].rgt < 0)
    { bits[node[v].lft] = code;
      lens[node[v].lft] = len;
    }
  else
    { build_table(node, node[v].lft, code << 1, len + 1, bits, lens);
      build_table(node, node[v].rgt, (code << 1) + 1, len + 1, bits, lens);
    }
}

int ref_huffman(const uint64_t hist[256], const ref_scheme *in, ref_scheme *out)
{ hnode node[512];
  int   heap[260];
  int   hsize = 0, value = 0, range, i;

  if (in != NULL)                                      /* QV.c:162-167: escape leaf "255" first */
    { node[0].count = 0;
      node[0].lft   = 255;
      node[0].rgt   = -1;
      heap[++hsize] = value++;
    }
  for (i = 0; i < 256; i++)                            /* QV.c:168-178 */
    if (hist[i] > 0)
      { if (in != NULL && (in->lens[i] > 16 || i == 255))
          node[0].count += hist[i];
        else
          { node[value].count = hist[i];
            node[value].lft   = i;
            node[value].rgt   = -1;
            heap[++hsize] = value++;
          }
      }
  if (value == 0)
    return REF_E_DEGEN;                                /* reference reads node[-1] here */

  for (i = hsize/2; i >= 1; i--)                       /* QV.c:180-181 */
    reheap(i, heap, hsize, node);

  range = value;                                       /* QV.c:183-194 */
  for (i = 1; i < value; i++)
    { int lft = heap[1], rgt;
      heap[1] = heap[hsize--];
      reheap(1, heap, hsize, node);
      rgt = heap[1];
      node[range].lft   = lft;
      node[range].rgt   = rgt;
      node[range].count = node[lft].count + node[rgt].count;
      heap[1] = range++;
      reheap(1, heap, hsize, node);
    }

  memset(out->bits, 0, sizeof(out->bits));
  memset(out->lens, 0, sizeof(out->lens));
  build_table(node, range - 1, 0, 0, out->bits, out->lens);

  if (in != NULL)                                      /* QV.c:203-210 */
    { out->type = 2;
      for (i = 0; i < 255; i++)
        if (in->lens[i] > 16 || out->lens[i] > 16)
          { out->lens[i] = out->lens[255];
            out->bits[i] = out->bits[255];
          }
    }
  else                                                 /* QV.c:211-217 */
    { out->type = 0;
      for (i = 0; i < 256; i++)
        if (out->lens[i] > 16)
          out->type = 1;
    }
  return 0;
}

/* QV.c:1069-1078: two-pass construction when the first tree has a code longer than 16 bits */
static int make_scheme(const uint64_t hist[256], ref_scheme *out)
{ ref_scheme first;
  int e = ref_huffman(hist, NULL, &first);
  if (e) return e;
  if (first.type)
    return ref_huffman(hist, &first, out);
  *out = first;
  return 0;
}

/* =========================================================================================
 *  statistics scan (QV.c:702-724, 922-1023)
 * ========================================================================================= */

static void histogram_seqs(uint64_t *hist, const uint8_t *s, size_t rlen)      /* QV.c:702-707 */
{ size_t k;
  for (k = 0; k < rlen; k++)
    hist[s[k]] += 1;
}

static void histogram_runs(uint64_t *run, const uint8_t *s, size_t rlen, int rc)  /* QV.c:709-724 */
{ size_t k = 0, h;
  while (k < rlen)
    { h = k;
      while (k < rlen && s[k] == rc)
        k += 1;
      if (k - h >= 256)
        run[255] += 1;
      else
        run[k-h] += 1;
      if (k < rlen)
        k += 1;
    }
}

/* One .quiva entry: header line + five equal-length lines (QV.c:751-798, 948-978). */
typedef struct { const uint8_t *hdr; size_t hlen; const uint8_t *line[5]; size_t rlen; } qentry;

/* returns 1 entry read, 0 clean eof, REF_E_FORMAT on error */
static int next_entry(text_t *t, qentry *e, int validate)
{ int r, j;
  size_t len;
  r = next_line(t, &e->hdr, &e->hlen);
  if (r == 0) return 0;
  if (r < 0) return REF_E_FORMAT;
  if (validate)
    { const uint8_t *slash;
      int well, beg, end, qv;
      if (e->hlen == 0 || e->hdr[0] != '@') return REF_E_FORMAT;              /* QV.c:954 */
      slash = (e->hlen > 1) ? find_byte(e->hdr + 1, e->hlen - 1, '/') : NULL; /* QV.c:958 */
      if (slash == NULL) return REF_E_FORMAT;
      if (scan_header(slash + 1, e->hlen - (size_t) (slash + 1 - e->hdr), "%d/%d_%d RQ=0.%d\n",
                      &well, &beg, &end, &qv, NULL, NULL, NULL) != 4)         /* QV.c:964 */
        return REF_E_FORMAT;
    }
  else if (e->hlen == 0)
    return 0;                                   /* dexqv.c:118: while (Read_Lines(input,1) > 0) */
  for (j = 0; j < 5; j++)
    { r = next_line(t, &e->line[j], &len);
      if (r < 0 && j > 0)                        /* the file's last line, without a newline, as line 2-5 of an entry: the reference's
                                                    fgets takes it and QV.c:792 compares its strlen with the first line's, newline
                                                    included: ONE character more than the other lines passes (the last character
                                                    stands where the newline would), anything else is "not the same length" */
        { const size_t left = t->n - t->pos;
          if (left != e->rlen + 1) return REF_E_FORMAT;
          e->line[j] = t->p + t->pos;
          t->pos = t->n;
          continue;
        }
      if (r <= 0) return REF_E_FORMAT;                                        /* QV.c:771-781, 788-791 */
      if (j == 0) e->rlen = len;
      else if (len != e->rlen) return REF_E_FORMAT;                           /* QV.c:792-795 */
    }
  return 1;
}

int ref_qv_scan(const uint8_t *quiva, size_t n, ref_qvstats *st)
{ text_t t = { quiva, n, 0 };
  qentry e;
  int    r, i;

  memset(st, 0, sizeof(*st));
  for (i = 0; i < 256; i++)                                                   /* QV.c:934-935 */
    st->hist[REF_DRUN][i] = st->hist[REF_SRUN][i] = 1;
  st->delChar = st->subChar = -1;
  st->del_first = st->sub_first = -1;

  while ((r = next_entry(&t, &e, 1)) > 0)
    { histogram_seqs(st->hist[REF_DEL], e.line[0], e.rlen);                   /* QV.c:988-991 */
      histogram_seqs(st->hist[REF_INS], e.line[2], e.rlen);
      histogram_seqs(st->hist[REF_MRG], e.line[3], e.rlen);
      histogram_seqs(st->hist[REF_SUB], e.line[4], e.rlen);

      if (st->delChar < 0)                                                    /* QV.c:993-1002 */
        { size_t k;
          for (k = 0; k < e.rlen; k++)
            if (e.line[1][k] == 'n' || e.line[1][k] == 'N')
              { st->delChar   = (int8_t) e.line[0][k];   /* `delChar = Read[k]` through a (signed) char */
                st->del_first = st->nentries;
                break;
              }
        }
      if (st->delChar >= 0)                                                   /* QV.c:1003-1004 */
        histogram_runs(st->hist[REF_DRUN], e.line[0], e.rlen, st->delChar);
      st->totChar += e.rlen;
      if (st->subChar < 0 && st->totChar >= 100000)                           /* QV.c:1006-1015 */
        { int k;
          st->subChar = 0;
          for (k = 1; k < 256; k++)
            if (st->hist[REF_SUB][k] > st->hist[REF_SUB][st->subChar])
              st->subChar = k;
          st->sub_first = st->nentries;
        }
      if (st->subChar >= 0)                                                   /* QV.c:1016-1017 */
        histogram_runs(st->hist[REF_SRUN], e.line[4], e.rlen, st->subChar);
      st->nentries += 1;
    }
  return r;
}

/* QV.c:1029-1169 */
int ref_qv_create(const ref_qvstats *st, int lossy, ref_coding *c)
{ uint64_t h[6][256];
  int      k, e;
  int      delChar = st->delChar, subChar = st->subChar;

  memcpy(h, st->hist, sizeof(h));
  memset(c, 0, sizeof(*c));

  if (st->totChar < 200000 || (double) h[REF_SUB][subChar < 0 ? 0 : subChar] < .5 * (double) st->totChar)
    subChar = -1;                                                             /* QV.c:1044-1045 */

  if (lossy)                                                                  /* QV.c:1049-1065 */
    { for (k = 0; k < 256; k += 2)
        { h[REF_INS][k] += h[REF_INS][k+1];
          h[REF_INS][k+1] = 0;
        }
      for (k = 0; k < 256; k += 4)
        { h[REF_MRG][k] += h[REF_MRG][k+1] + h[REF_MRG][k+2] + h[REF_MRG][k+3];
          h[REF_MRG][k+1] = h[REF_MRG][k+2] = h[REF_MRG][k+3] = 0;
        }
    }

  if (delChar >= 0)                                                           /* QV.c:1097-1108 */
    { h[REF_DEL][delChar] = 0;
      if ((e = make_scheme(h[REF_DEL],  &c->s[REF_DEL])))  return e;
      if ((e = make_scheme(h[REF_DRUN], &c->s[REF_DRUN]))) return e;
    }
  else if ((e = make_scheme(h[REF_DEL], &c->s[REF_DEL]))) return e;

  if ((e = make_scheme(h[REF_INS], &c->s[REF_INS]))) return e;               /* QV.c:1121-1122 */
  if ((e = make_scheme(h[REF_MRG], &c->s[REF_MRG]))) return e;

  if (subChar >= 0)                                                           /* QV.c:1124-1135 */
    { h[REF_SUB][subChar] = 0;
      if ((e = make_scheme(h[REF_SUB],  &c->s[REF_SUB])))  return e;
      if ((e = make_scheme(h[REF_SRUN], &c->s[REF_SRUN]))) return e;
    }
  else if ((e = make_scheme(h[REF_SUB], &c->s[REF_SUB]))) return e;

  c->delChar = delChar;
  c->subChar = subChar;
  return 0;
}

static void write_scheme(sink_t *o, const ref_scheme *s)                      /* QV.c:300-318 */
{ int i;
  sink_u8(o, (uint8_t) s->type);
  for (i = 0; i < 256; i++)
    { uint8_t x = (uint8_t) s->lens[i];
      sink_u8(o, x);
      if (x > 0)
        sink_u32(o, s->bits[i]);
    }
}

static void write_coding(sink_t *o, const ref_coding *c, const char *prefix, size_t plen)  /* QV.c:1173-1210 */
{ sink_u16(o, 0x33cc);
  sink_u16(o, c->delChar < 0 ? 256 : (uint16_t) c->delChar);
  sink_u16(o, c->subChar < 0 ? 256 : (uint16_t) c->subChar);
  sink_i32(o, (int32_t) plen);
  sink_put(o, prefix, plen);
  write_scheme(o, &c->s[REF_DEL]);
  if (c->delChar >= 0) write_scheme(o, &c->s[REF_DRUN]);
  write_scheme(o, &c->s[REF_INS]);
  write_scheme(o, &c->s[REF_MRG]);
  write_scheme(o, &c->s[REF_SUB]);
  if (c->subChar >= 0) write_scheme(o, &c->s[REF_SRUN]);
}

long ref_qv_write_coding(const ref_coding *c, const char *prefix, uint8_t *out, size_t cap)
{ sink_t o;
  sink_init(&o, out, cap);
  write_coding(&o, c, prefix, strlen(prefix));
  return o.over ? REF_E_SPACE : (long) o.len;
}

/* =========================================================================================
 *  bit packer (QV.c:386-506)
 * ========================================================================================= */

typedef struct { sink_t *o; uint32_t word; int olen, llen; } bitw;

/* OCODE, QV.c:404-422: append the low `len` bits of `code`, MSB first, flushing whole uint32s */
static void ocode(bitw *w, int len, uint32_t code)
{ int tot = w->olen + len;
  w->llen = w->olen;
  if (tot >= 32)
    { w->olen  = tot - 32;
      w->word |= (w->olen < 32) ? (code >> w->olen) : 0;
      sink_u32(w->o, w->word);
      w->word = (w->olen > 0) ? (code << (32 - w->olen)) : 0;
    }
  else
    { w->olen = tot;
      if (tot > 0)                       /* reference shifts by 32 when tot==0 (code is 0 then) */
        w->word |= code << (32 - tot);
    }
}

/* QV.c:436-442: the "tricky" tail that keeps the decoder's 16-bit look-ahead inside the stream */
static void bit_finish(bitw *w)
{ if (w->olen > 0)
    { sink_u32(w->o, w->word);
      if (w->llen > 16 && w->olen > w->llen)
        sink_u32(w->o, w->word);
    }
  else if (w->llen > 16)
    sink_u32(w->o, w->word);
}

static void encode_plain(const ref_scheme *s, sink_t *o, const uint8_t *rd, int rlen, int mask)   /* QV.c:386-443 */
{ bitw     w = { o, 0, 0, 0 };
  uint32_t nspec = 0x7fffffff;
  int      nslen = 0x7fffffff, k;
  if (s->type == 2) { nspec = s->bits[255]; nslen = s->lens[255]; }
  for (k = 0; k < rlen; k++)
    { uint32_t x = rd[k] & (uint32_t) mask;
      int      n = s->lens[x];
      uint32_t c = s->bits[x];
      ocode(&w, n, c);
      if (c == nspec && n == nslen)
        ocode(&w, 8, x);
    }
  bit_finish(&w);
}

static void encode_run(const ref_scheme *ns, const ref_scheme *rs, sink_t *o,                      /* QV.c:448-506 */
                       const uint8_t *rd, int rlen, int rchar)
{ bitw     w = { o, 0, 0, 0 };
  uint32_t nspec = 0x7fffffff, rspec = rs->bits[255];
  int      nslen = 0x7fffffff, rslen = rs->lens[255], k = 0;
  if (ns->type == 2) { nspec = ns->bits[255]; nslen = ns->lens[255]; }
  while (k < rlen)
    { int      h = k, n;
      uint32_t x, c;
      while (k < rlen && rd[k] == rchar)
        k += 1;
      x = (k - h >= 255) ? 255 : (uint32_t) (k - h);
      n = rs->lens[x];
      c = rs->bits[x];
      ocode(&w, n, c);
      if (c == rspec && n == rslen)
        ocode(&w, 16, (uint32_t) (k - h));
      if (k < rlen)
        { x = rd[k];
          n = ns->lens[x];
          c = ns->bits[x];
          ocode(&w, n, c);
          if (c == nspec && n == nslen)
            ocode(&w, 8, x);
          k += 1;
        }
    }
  bit_finish(&w);
}

/* QV.c:1381-1426 (and its in-memory twin :1343-1379) */
static void encode_entry(const ref_coding *c, int lossy, int rlen, const uint8_t *const line[5],
                         sink_t *o, uint32_t seg[5], uint8_t *scratch)
{ size_t  at = o->len;
  int     clen, k;

  if (c->delChar < 0)                                                         /* QV.c:1393-1401 */
    { encode_plain(&c->s[REF_DEL], o, line[0], rlen, 0xff);
      clen = rlen;
      memcpy(scratch, line[1], (size_t) rlen);
    }
  else
    { encode_run(&c->s[REF_DEL], &c->s[REF_DRUN], o, line[0], rlen, c->delChar);
      clen = 0;                                                               /* Pack_Tag, QV.c:810-819 */
      for (k = 0; k < rlen; k++)
        if ((int8_t) line[0][k] != c->delChar)        /* `qvs[k] != rchar` with qvs a (signed) char* */
          scratch[clen++] = line[1][k];
    }
  seg[0] = (uint32_t) (o->len - at); at = o->len;

  ref_number_read(scratch, (size_t) clen);                                    /* QV.c:1402-1404 */
  { uint8_t *pk = scratch + rlen + 4;
    size_t   pl = ref_compress_read((size_t) clen, scratch, pk);
    sink_put(o, pk, pl);
  }
  seg[1] = (uint32_t) (o->len - at); at = o->len;

  encode_plain(&c->s[REF_INS], o, line[2], rlen, lossy ? 0xfe : 0xff);        /* QV.c:1406-1418 */
  seg[2] = (uint32_t) (o->len - at); at = o->len;
  encode_plain(&c->s[REF_MRG], o, line[3], rlen, lossy ? 0xfc : 0xff);
  seg[3] = (uint32_t) (o->len - at); at = o->len;

  if (c->subChar < 0)                                                         /* QV.c:1419-1423 */
    encode_plain(&c->s[REF_SUB], o, line[4], rlen, 0xff);
  else
    encode_run(&c->s[REF_SUB], &c->s[REF_SRUN], o, line[4], rlen, c->subChar);
  seg[4] = (uint32_t) (o->len - at);
}

long ref_qv_encode_entry(const ref_coding *c, int lossy, int rlen,
                         const uint8_t *del, const uint8_t *tag, const uint8_t *ins,
                         const uint8_t *mrg, const uint8_t *sub,
                         uint8_t *out, size_t cap, uint32_t seg[5])
{ const uint8_t *line[5] = { del, tag, ins, mrg, sub };
  uint8_t *scratch = malloc(2 * (size_t) rlen + 16);
  sink_t   o;
  sink_init(&o, out, cap);
  encode_entry(c, lossy, rlen, line, &o, seg, scratch);
  free(scratch);
  return o.over ? REF_E_SPACE : (long) o.len;
}

/* dexqv.c:79-143 */
long ref_dexqv(const uint8_t *quiva, size_t n, int lossy, uint8_t *out, size_t cap)
{ ref_qvstats *st;
  ref_coding  *c;
  text_t   t = { quiva, n, 0 };
  qentry   e;
  sink_t   o;
  uint8_t *scratch = NULL;
  size_t   smax = 0;
  int      r, lwell = 0;
  long     ret;

  st = malloc(sizeof(*st));
  c  = malloc(sizeof(*c));
  sink_init(&o, out, cap);

  r = ref_qv_scan(quiva, n, st);                                              /* dexqv.c:81-82 */
  if (r < 0) { ret = r; goto done; }
  r = ref_qv_create(st, lossy, c);                                            /* dexqv.c:86 */
  if (r < 0) { ret = r; goto done; }

  { const uint8_t *hdr, *slash;                                               /* dexqv.c:90-103 */
    size_t hlen;
    text_t t0 = { quiva, n, 0 };
    if (next_line(&t0, &hdr, &hlen) <= 0 || hlen < 2) { ret = REF_E_FORMAT; goto done; }
    slash = find_byte(hdr + 1, hlen - 1, '/');
    if (slash == NULL) { ret = REF_E_FORMAT; goto done; }
    sink_u16(&o, 0x55aa);                                                     /* dexqv.c:105-108 */
    write_coding(&o, c, (const char *) hdr, (size_t) (slash - hdr));
  }

  while ((r = next_entry(&t, &e, 0)) > 0)                                     /* dexqv.c:118-142 */
    { const uint8_t *slash = find_byte(e.hdr, e.hlen, '/');
      int       well = 0, beg = 0, end = 0, qv = 0;
      uint32_t  seg[5];
      if (slash == NULL) { ret = REF_E_FORMAT; goto done; }
      scan_header(slash + 1, e.hlen - (size_t) (slash + 1 - e.hdr), "%d/%d_%d RQ=0.%d\n",
                  &well, &beg, &end, &qv, NULL, NULL, NULL);
      put_well(&o, well, &lwell);
      sink_i32(&o, beg);
      sink_i32(&o, end);
      sink_i32(&o, qv);
      if (2 * e.rlen + 16 > smax)
        { smax    = 2 * e.rlen + 4096;
          scratch = realloc(scratch, smax);
        }
      encode_entry(c, lossy, (int) e.rlen, e.line, &o, seg, scratch);
    }
  ret = (r < 0) ? r : (o.over ? REF_E_SPACE : (long) o.len);

done:
  free(scratch);
  free(st);
  free(c);
  return ret;
}

/* =========================================================================================
 *  decoder (QV.c:322-375, 510-691, 823-847, 1214-1320, 1428-1481; undexqv.c:101-208)
 * ========================================================================================= */

typedef struct
  { int32_t  type;
    int32_t  lens[256];
    uint32_t bits[256];
    uint8_t  look[0x10000];
  } dscheme;

static int read_scheme(bin_t *b, int flip, dscheme *s)                        /* QV.c:322-375 */
{ int i;
  uint8_t x;
  bin_get(b, &x, 1);
  s->type = x;
  for (i = 0; i < 256; i++)
    { bin_get(b, &x, 1);
      s->lens[i] = x;
      if (x > 0)
        { uint32_t v;
          bin_get(b, &v, 4);
          s->bits[i] = flip ? swap32(v) : v;
        }
      else
        s->bits[i] = 0;
    }
  if (b->fail) return REF_E_TRUNC;
  memset(s->look, 0, sizeof(s->look));
  for (i = 0; i < 256; i++)                 /* ascending i: for shared escape codes 255 wins */
    if (s->lens[i] > 0)
      { uint32_t base, powr, j;
        if (s->lens[i] > 16) return REF_E_FORMAT;
        base = s->bits[i] << (16 - s->lens[i]);
        powr = 1u << (16 - s->lens[i]);
        for (j = 0; j < powr; j++)
          s->look[(base + j) & 0xffff] = (uint8_t) i;
      }
  return 0;
}

/* The reference's 64-bit shift register (QV.c:537-551): the 16-bit window sits at bits 32..47;
 * a new uint32 is pulled in whenever the bits to drop exceed the bits left below the window.  */
typedef struct { bin_t *b; int flip; uint64_t icode; int ilen, n; } bitr;

static int bit_get(bitr *r)
{ if (r->n > r->ilen)
    { uint32_t w;
      r->icode <<= r->ilen;
      bin_get(r->b, &w, 4);
      if (r->b->fail) return REF_E_TRUNC;
      if (r->flip) w = swap32(w);
      r->icode = (r->icode & 0xffffffff00000000ull) | w;
      r->ilen   = r->n - r->ilen;
      r->icode <<= r->ilen;
      r->ilen   = 32 - r->ilen;
    }
  else
    { r->icode <<= r->n;
      r->ilen   -= r->n;
    }
  return 0;
}

#define WIN16(r) ((uint32_t) (((r)->icode >> 32) & 0xffff))
#define WIN8(r)  ((uint32_t) (((r)->icode >> 40) & 0xff))

static int decode_plain(const dscheme *s, bin_t *b, int flip, uint8_t *rd, int rlen)     /* QV.c:510-599 */
{ bitr r = { b, flip, 0, 0, 16 };
  int  signal = (s->type == 2) ? 255 : 256, j;
  for (j = 0; j < rlen; j++)
    { int c;
      if (bit_get(&r)) return REF_E_TRUNC;
      c   = s->look[WIN16(&r)];
      r.n = s->lens[c];
      if (c == signal)
        { if (bit_get(&r)) return REF_E_TRUNC;
          c   = (int) WIN8(&r);
          r.n = 8;
        }
      rd[j] = (uint8_t) c;
    }
  return 0;
}

static int decode_run(const dscheme *ns, const dscheme *rs, bin_t *b, int flip,          /* QV.c:604-691 */
                      uint8_t *rd, int rlen, int rchar)
{ bitr r = { b, flip, 0, 0, 16 };
  int  nsignal = (ns->type == 2) ? 255 : 256, j;
  for (j = 0; j < rlen; j++)
    { int c, k;
      if (bit_get(&r)) return REF_E_TRUNC;
      c   = rs->look[WIN16(&r)];
      r.n = rs->lens[c];
      if (c == 255)
        { if (bit_get(&r)) return REF_E_TRUNC;
          c   = (int) WIN16(&r);
          r.n = 16;
        }
      for (k = 0; k < c; k++)
        { if (j >= rlen) return REF_E_FORMAT;       /* the reference would overrun its buffer */
          rd[j++] = (uint8_t) rchar;
        }
      if (j < rlen)
        { if (bit_get(&r)) return REF_E_TRUNC;
          c   = ns->look[WIN16(&r)];
          r.n = ns->lens[c];
          if (c == nsignal)
            { if (bit_get(&r)) return REF_E_TRUNC;
              c   = (int) WIN8(&r);
              r.n = 8;
            }
          rd[j] = (uint8_t) c;
        }
    }
  return 0;
}

long ref_undexqv(const uint8_t *img, size_t n, int upper, uint8_t *out, size_t cap)
{ bin_t    b = { img, n, 0, 0 };
  sink_t   o;
  uint16_t half;
  int      newv, flip, delChar, subChar, well = 0, e;
  int32_t  plen;
  char    *prefix = NULL;
  dscheme *sc[6] = { NULL, NULL, NULL, NULL, NULL, NULL };
  uint8_t *ent = NULL;
  size_t   emax = 0;
  long     ret;
  static const char lower[4] = { 'a','c','g','t' };

  sink_init(&o, out, cap);

  bin_get(&b, &half, 2);                                                      /* undexqv.c:103-110 */
  if (b.fail) return REF_E_TRUNC;
  if (half == 0x55aa || half == 0xaa55)
    newv = 1;
  else
    { newv = 0;
      b.pos = 0;
    }

  bin_get(&b, &half, 2);                                                      /* QV.c:1222-1266 */
  if (b.fail) return REF_E_TRUNC;
  flip = (half != 0x33cc);
  delChar = get_u16(&b, flip); if (delChar >= 256) delChar = -1;
  subChar = get_u16(&b, flip); if (subChar >= 256) subChar = -1;
  plen = get_i32(&b, flip);
  if (b.fail || plen < 0 || (size_t) plen > n) return REF_E_TRUNC;
  prefix = malloc((size_t) plen + 1);
  bin_get(&b, prefix, (size_t) plen);
  prefix[plen] = '\0';

  for (e = 0; e < 6; e++)                                                     /* QV.c:1281-1302 */
    { static const int order[6] = { REF_DEL, REF_DRUN, REF_INS, REF_MRG, REF_SUB, REF_SRUN };
      int id = order[e];
      if ((id == REF_DRUN && delChar < 0) || (id == REF_SRUN && subChar < 0))
        continue;
      sc[id] = malloc(sizeof(dscheme));
      if ((ret = read_scheme(&b, flip, sc[id])) < 0) goto done;
    }

  while (b.pos < b.n)                                                         /* undexqv.c:119-208 */
    { uint8_t byte;
      int     beg, end, qv, rlen, clen, k;
      uint8_t *line[5];

      bin_get(&b, &byte, 1);
      while (byte == 255)
        { well += 255;
          bin_get(&b, &byte, 1);
          if (b.fail) { ret = REF_E_TRUNC; goto done; }
        }
      well += byte;
      if (newv)
        { beg = get_i32(&b, flip); end = get_i32(&b, flip); qv = get_i32(&b, flip); }
      else
        { beg = get_u16(&b, flip); end = get_u16(&b, flip); qv = get_u16(&b, flip); }
      if (b.fail) { ret = REF_E_TRUNC; goto done; }

      put_text(&o, "%s/%d/%d_%d RQ=0.%d\n", prefix, well, beg, end, qv);      /* undexqv.c:182 */

      rlen = end - beg;                                                       /* undexqv.c:186 */
      if (rlen < 0) { ret = REF_E_FORMAT; goto done; }
      if (5 * ((size_t) rlen + 8) > emax)
        { emax = 5 * ((size_t) rlen + 8) + 4096;
          ent  = realloc(ent, emax);
        }
      for (k = 0; k < 5; k++)
        line[k] = ent + (size_t) k * ((size_t) rlen + 8);

      if (delChar < 0)                                                        /* QV.c:1433-1462 */
        { if ((ret = decode_plain(sc[REF_DEL], &b, flip, line[0], rlen)) < 0) goto done;
          clen = rlen;
        }
      else
        { if ((ret = decode_run(sc[REF_DEL], sc[REF_DRUN], &b, flip, line[0], rlen, delChar)) < 0) goto done;
          clen = 0;
          for (k = 0; k < rlen; k++)
            if ((int8_t) line[0][k] != delChar)
              clen += 1;
        }
      { size_t tlen = ((size_t) clen + 3) >> 2;
        int    j;
        if (b.pos + tlen > b.n) { ret = REF_E_TRUNC; goto done; }
        ref_uncompress_read((size_t) clen, b.p + b.pos, line[1]);
        b.pos += tlen;
        for (k = 0; k < clen; k++) line[1][k] = (uint8_t) lower[line[1][k]];
        if (delChar >= 0)                                                     /* Unpack_Tag, QV.c:837-847 */
          { j = clen - 1;
            for (k = rlen - 1; k >= 0; k--)
              if ((int8_t) line[0][k] == delChar)
                line[1][k] = 'n';
              else
                line[1][k] = line[1][j--];
          }
      }

      if ((ret = decode_plain(sc[REF_INS], &b, flip, line[2], rlen)) < 0) goto done;   /* QV.c:1464-1468 */
      if ((ret = decode_plain(sc[REF_MRG], &b, flip, line[3], rlen)) < 0) goto done;
      if (subChar < 0)                                                        /* QV.c:1470-1478 */
        { if ((ret = decode_plain(sc[REF_SUB], &b, flip, line[4], rlen)) < 0) goto done; }
      else
        { if ((ret = decode_run(sc[REF_SUB], sc[REF_SRUN], &b, flip, line[4], rlen, subChar)) < 0) goto done; }

      if (upper)                                                              /* undexqv.c:198-204 */
        for (k = 0; k < rlen; k++)
          line[1][k] = (uint8_t) (line[1][k] - 32);

      for (k = 0; k < 5; k++)                                                 /* undexqv.c:206-207 */
        { sink_put(&o, line[k], (size_t) rlen);
          sink_u8(&o, '\n');
        }
    }
  ret = o.over ? REF_E_SPACE : (long) o.len;

done:
  for (e = 0; e < 6; e++) free(sc[e]);
  free(prefix);
  free(ent);
  return ret;
}
