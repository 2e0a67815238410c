/*
 * dexref.h -- CPU restatement of DEXTRACTOR's three codecs (TEST INFRASTRUCTURE ONLY).
 *
 * This is the ORACLE: a scalar, single-threaded C restatement of the reference's algorithm for
 * the dexta/undexta, dexar/undexar and dexqv/undexqv path.  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * path (libdexgpu + the CLI tools) never links or calls anything in this directory.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * outputs of the real reference tools (oracle/_ref/, compiled from /root/reference by
 * oracle/Makefile) on the committed fixtures in tests/golden/, and -- when oracle/_ref is
 * present -- on freshly generated seeded corpora in every table regime of SURVEY.md 8(c).
 *
 * All functions work on memory images of the files (no FILE*), so that the same bytes can be
 * handed to the GPU path.  Multi-byte integers are written in native (little-endian) order,
 * exactly as the reference's fwrite calls do on this host.
 */
#ifndef DEXREF_H
#define DEXREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (negative returns) */
#define REF_E_FORMAT   (-1)   /* malformed input (the reference would print a message and exit 1) */
#define REF_E_SPACE    (-2)   /* output buffer too small */
#define REF_E_TRUNC    (-3)   /* input ended early (reference: SYSTEM_READ_ERROR / "Could not read") */
#define REF_E_DEGEN    (-4)   /* empty histogram: reference behaviour undefined (QV.c:201 node[-1]) */

/* ---- 2-bit packer primitives (DB.c:319-441) ---- */
void   ref_number_read (uint8_t *s, size_t n);               /* DB.c:393  c/C->1 g/G->2 t/T->3 else 0 */
void   ref_number_arrow(uint8_t *s, size_t n);               /* DB.c:418  '1'->0 '2'->1 '3','G'->2 else 3 */
size_t ref_compress_read  (size_t len, const uint8_t *num, uint8_t *out);   /* DB.c:319 */
void   ref_uncompress_read(size_t len, const uint8_t *in,  uint8_t *num);   /* DB.c:342 */

/* ---- whole-file codecs on memory images; return output length or REF_E_* ---- */
long ref_dexta  (const uint8_t *fasta, size_t n, uint8_t *out, size_t cap);                 /* dexta.c:104-205 */
long ref_undexta(const uint8_t *dexta, size_t n, int upper, int width, uint8_t *out, size_t cap); /* undexta.c:131-271 */
long ref_dexar  (const uint8_t *arrow, size_t n, uint8_t *out, size_t cap);                 /* dexar.c:103-211 */
long ref_undexar(const uint8_t *dexar, size_t n, int width, uint8_t *out, size_t cap);      /* undexar.c:129-229 */
long ref_dexqv  (const uint8_t *quiva, size_t n, int lossy, uint8_t *out, size_t cap);      /* dexqv.c:79-143 */
long ref_undexqv(const uint8_t *dexqv, size_t n, int upper, uint8_t *out, size_t cap);      /* undexqv.c:101-208 */

/* ---- QV coder pieces, for kernel-level parity ---- */
enum { REF_DEL = 0, REF_INS = 1, REF_MRG = 2, REF_SUB = 3, REF_DRUN = 4, REF_SRUN = 5 };

typedef struct
  { uint64_t hist[6][256];     /* del, ins, mrg, sub symbol counts; dRun, sRun run-length counts
                                  (run bins start at 1, QV.c:934-935)                              */
    uint64_t totChar;          /* QV.c:1005 */
    int32_t  delChar, subChar; /* -1 = none (QV.c:938-939) */
    int64_t  del_first;        /* index of the entry in which delChar was discovered (-1 none)    */
    int64_t  sub_first;        /* index of the entry at which subChar was chosen (-1 none)        */
    int64_t  nentries;
  } ref_qvstats;

typedef struct
  { int32_t  type;             /* 0 normal, 2 truncated with escape (QV.c:76-81) */
    uint32_t bits[256];
    int32_t  lens[256];
  } ref_scheme;

typedef struct
  { ref_scheme s[6];           /* indexed by REF_DEL..REF_SRUN */
    int32_t    delChar, subChar;
  } ref_coding;

int  ref_qv_scan  (const uint8_t *quiva, size_t n, ref_qvstats *st);              /* QV.c:922-1023 */
int  ref_huffman  (const uint64_t hist[256], const ref_scheme *in, ref_scheme *out);  /* QV.c:147-220 */
int  ref_qv_create(const ref_qvstats *st, int lossy, ref_coding *c);              /* QV.c:1029-1169 */
long ref_qv_write_coding(const ref_coding *c, const char *prefix, uint8_t *out, size_t cap); /* QV.c:1173-1210 */

/* One entry, five segments (QV.c:1381-1426).  Streams are NOT modified (the reference mutates
 * them in place; the lossy mask is applied on the fly).  seg[0..4] receive the byte size of
 * the del / tag / ins / mrg / sub segments.  Returns total bytes or REF_E_SPACE.              */
long ref_qv_encode_entry(const ref_coding *c, int lossy, int rlen,
                         const uint8_t *del, const uint8_t *tag, const uint8_t *ins,
                         const uint8_t *mrg, const uint8_t *sub,
                         uint8_t *out, size_t cap, uint32_t seg[5]);

#ifdef __cplusplus
}
#endif
#endif
