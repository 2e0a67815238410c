/* Host-side C of libdexgpu (dx_host.c) under AddressSanitizer + UBSan: walks, indexers, table
 * builder on files given on the command line and on truncated / corrupted copies of them.
 * Built and run by tests/test_host_asan.py; test infrastructure only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dexgpu.h"

static uint8_t *slurp(const char *path, size_t *n)
{ FILE *f = fopen(path, "rb");
  uint8_t *b;
  long sz;
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET);
  b = malloc((size_t) sz ? (size_t) sz : 1);            /* exact size: any over-read is caught */
  if (sz && fread(b, 1, (size_t) sz, f) != (size_t) sz) exit(2);
  fclose(f);
  *n = (size_t) sz;
  return b;
}

static unsigned long long checksum = 0;

static void try_walk(const uint8_t *img, size_t n)
{ dx_qv_index x, y;
  int a = dx_qv_walk(img, n, &x), b = dx_qv_walk_indexed(img, n, &y, 1);     /* the same records with and without the group index */
  if ((a == DX_OK) != (b == DX_OK)) { fprintf(stderr, "walk %d but indexed walk %d\n", a, b); exit(3); }
  if (a == DX_OK)
    { uint64_t i, w = 0;
      if (x.n != y.n || x.rec_off[x.n] != y.rec_off[y.n] || y.gidx_off[y.n] != y.gidx_words) { fprintf(stderr, "indexed walk differs\n"); exit(3); }
      for (i = 0; i < y.n; i++)
        { if (x.len[i] != y.len[i] || x.seg[5*i] != y.seg[5*i] || x.seg[5*i+4] != y.seg[5*i+4]) { fprintf(stderr, "indexed walk differs at %llu\n", (unsigned long long) i); exit(3); }
          w += y.gidx_off[i+1] - y.gidx_off[i];
        }
      for (i = 0; i < y.gidx_words; i++) checksum += y.gidx[i];           /* every word of the index is read: all of it initialised */
      checksum += x.n + x.rec_off[x.n] + w;
      dx_qv_index_free(&x); dx_qv_index_free(&y);
    }
}

static void try_index(const uint8_t *txt, size_t n, int kind)
{ uint64_t cnt = 0, line = 0;
  size_t   pl = 0;
  int      code = 0;
  if (kind == 0)
    { if (dx_index_quiva(txt, n, 0, NULL, NULL, NULL, &cnt, &pl, &line, &code) == DX_OK && cnt)
        { uint64_t *off = malloc(cnt * 8); uint32_t *len = malloc(cnt * 4); int32_t *h = malloc(cnt * 16);
          if (dx_index_quiva(txt, n, cnt, off, len, h, &cnt, &pl, &line, &code) == DX_OK) checksum += off[cnt - 1] + len[0];
          free(off); free(len); free(h);
        }
    }
  else
    { if (dx_index_seq(kind == 2, txt, n, 0, NULL, NULL, NULL, NULL, NULL, &cnt, &pl, &line, &code) == DX_OK && cnt)
        { uint64_t *off = malloc(cnt * 8); uint32_t *tl = malloc(cnt * 4), *ns = malloc(cnt * 4);
          int32_t *h = malloc(cnt * 16); uint16_t *c4 = malloc(cnt * 8);
          if (dx_index_seq(kind == 2, txt, n, cnt, off, tl, ns, h, c4, &cnt, &pl, &line, &code) == DX_OK) checksum += off[cnt - 1] + ns[0];
          free(off); free(tl); free(ns); free(h); free(c4);
        }
    }
}

int main(int argc, char **argv)
{ int i;
  for (i = 1; i < argc; i++)
    { size_t n, k;
      uint8_t *b = slurp(argv[i], &n), *c;
      const char *ext = strrchr(argv[i], '.');
      const int kind = ext && !strcmp(ext, ".quiva") ? 0 : (ext && !strcmp(ext, ".fasta") ? 1 : (ext && !strcmp(ext, ".arrow") ? 2 : 3));
      unsigned rng = 12345u + (unsigned) i;
      for (k = 0; k < 40; k++)                           /* whole file, then ever shorter prefixes, each in its own exact-size block */
        { const size_t m = k == 0 ? n : (size_t) ((double) n * (40 - k) / 40.0) + (k & 3);
          const size_t mm = m > n ? n : m;
          c = malloc(mm ? mm : 1);
          memcpy(c, b, mm);
          if (kind == 3) try_walk(c, mm); else try_index(c, mm, kind);
          if (mm > 64)                                   /* and with a few bytes damaged */
            { int z;
              for (z = 0; z < 8; z++)
                { rng = rng * 1664525u + 1013904223u;
                  c[rng % mm] ^= (uint8_t) (1u << (rng >> 29));
                }
              if (kind == 3) try_walk(c, mm); else try_index(c, mm, kind);
            }
          free(c);
        }
      free(b);
    }
  { uint64_t hist[6][256];                               /* table builder on a few shapes */
    dx_qv_params p = { -1, -1, -1, -1 };
    dx_qv_coding cd;
    int s, k2;
    for (s = 0; s < 6; s++) for (k2 = 0; k2 < 256; k2++) hist[s][k2] = (uint64_t) ((k2 * 2654435761u + s) % 97u) * (k2 % 7 == 0);
    for (s = 0; s < 4; s++) hist[s][33] = 1000, hist[s][34] = 1;
    if (dx_qv_build(hist, 300000, &p, 0, &cd) == DX_OK)
      { uint8_t buf[8192]; size_t len = 0;
        if (dx_qv_write_coding(&cd, "@m", 2, buf, sizeof(buf), &len) == DX_OK) checksum += len;
      }
  }
  printf("ok %llu\n", checksum);
  return 0;
}
