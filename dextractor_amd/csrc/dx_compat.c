/*
 * dx_compat.c -- QVcoding_Scan1 / Create_QVcoding / Write_QVcoding / Compress_Next_QVentry1 /
 * Free_QVcoding (QV.h:61-87) for a dex2DB-style caller, over the batch API of libdexgpu.
 * See include/dexcompat.h.  Static state, as in QV.c (which keeps its histograms and the coding
 * in file-scope variables: QV.c:860-862, 1030).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dexgpu.h"
#include "dexcompat.h"

static dx_ctx       *Ctx     = NULL;
static dx_entries   *Batch   = NULL;      /* entries gathered by QVcoding_Scan1 */
static uint32_t     *Lens    = NULL;      /* their lengths: the second pass is checked against them */
static uint64_t     *Sums    = NULL;      /* ... and a checksum of each entry's five lines: the second pass hands the same bytes, or dies */
static uint64_t      Nent = 0, LensCap = 0;
static dx_qv_coding  Tables;              /* what Create_QVcoding built */
static uint8_t      *Records = NULL;      /* every entry already compressed (Create_QVcoding) */
static uint64_t     *Coff    = NULL;      /* n + 1 offsets into Records */
static uint64_t      Next    = 0;         /* entry the next Compress_Next_QVentry1 call writes */
static QVcoding      Coding;              /* the object handed to the caller (QV.c:1030: static there too) */

static void die(const char *msg)
{ fprintf(stderr, "libdexgpu: %s\n", msg);          /* batch error convention, DB.h:45-47 */
  exit(1);
}

/* FNV-1a over the five lines of an entry (64 bits: a second pass that hands other bytes of the same length is caught) */
static uint64_t entry_sum(int rlen, const char *const line[5])
{ uint64_t h = 0xcbf29ce484222325ull;
  int k, j;
  for (k = 0; k < 5; k++)
    for (j = 0; j < rlen; j++)
      { h ^= (uint8_t) line[k][j];
        h *= 0x100000001b3ull;
      }
  return h;
}

static void drop_results(void)
{ dx_file_free(Records); dx_file_free(Coff);
  Records = NULL; Coff = NULL; Next = 0;
}

void QVcoding_Scan1(int rlen, char *del, char *tag, char *ins, char *mrg, char *sub)
{ if (rlen == 0)                                    /* reset: QV.c:871-886 */
    { dx_entries_free(Batch);
      Batch = dx_entries_new();
      if (Batch == NULL) die("Out of memory (QVcoding_Scan1)");
      Nent = 0;
      drop_results();
      return;
    }
  if (Batch == NULL)
    { Batch = dx_entries_new();
      if (Batch == NULL) die("Out of memory (QVcoding_Scan1)");
    }
  if (dx_entries_add(Batch, rlen, del, tag, ins, mrg, sub) != DX_OK)
    die("Out of memory (QVcoding_Scan1)");
  if (Nent == LensCap)
    { uint64_t  nc = LensCap ? 2 * LensCap : 1024;
      uint32_t *t  = realloc(Lens, nc * sizeof(*t));
      uint64_t *u;
      if (t == NULL) die("Out of memory (QVcoding_Scan1)");
      Lens = t;
      u = realloc(Sums, nc * sizeof(*u));
      if (u == NULL) die("Out of memory (QVcoding_Scan1)");
      Sums = u; LensCap = nc;
    }
  { const char *const line[5] = { del, tag, ins, mrg, sub };
    Sums[Nent] = entry_sum(rlen, line);
  }
  Lens[Nent++] = (uint32_t) rlen;
}

QVcoding *Create_QVcoding(int lossy)
{ size_t nbytes = 0;
  int    rc;
  if (Batch == NULL || Nent == 0)
    die("Create_QVcoding: no entries were scanned");
  if (Ctx == NULL)
    { const char *d = getenv("DEXGPU_DEVICE");
      if (dx_open(d ? atoi(d) : 0, &Ctx) != DX_OK)
        { fprintf(stderr, "libdexgpu: cannot open a GPU (%s)\n", dx_last_error(NULL));
          exit(1);
        }
    }
  drop_results();
  rc = dx_entries_compress(Ctx, Batch, lossy, &Tables, &Records, &nbytes, &Coff);
  if (rc != DX_OK)
    { fprintf(stderr, "libdexgpu: Create_QVcoding: %s\n", dx_last_error(Ctx));
      exit(1);
    }
  dx_entries_free(Batch);                           /* the streams are no longer needed: the records exist */
  Batch = NULL;
  memset(&Coding, 0, sizeof(Coding));
  Coding.delScheme = &Tables;                       /* opaque to the caller, as in the reference */
  Coding.delChar   = Tables.delChar;
  Coding.subChar   = Tables.subChar;
  Coding.flip      = 0;
  Coding.prefix    = NULL;                          /* set by the caller before Write_QVcoding (dex2DB.c:561-565) */
  return &Coding;
}

void Write_QVcoding(FILE *output, QVcoding *coding)
{ size_t   need = 0, plen;
  uint8_t *buf;
  if (coding == NULL || coding->delScheme != (void *) &Tables)
    die("Write_QVcoding: not the coding Create_QVcoding returned");
  plen = coding->prefix ? strlen(coding->prefix) : 0;
  dx_qv_write_coding(&Tables, coding->prefix, plen, NULL, 0, &need);
  buf = malloc(need + 1);
  if (buf == NULL) die("Out of memory (Write_QVcoding)");
  if (dx_qv_write_coding(&Tables, coding->prefix, plen, buf, need, &need) != DX_OK ||
      fwrite(buf, 1, need, output) != need)
    die("Write_QVcoding: write failed");
  free(buf);
}

void Compress_Next_QVentry1(int rlen, char *del, char *tag, char *ins, char *mrg, char *sub,
                            FILE *output, QVcoding *coding, int lossy)
{ size_t k;
  (void) lossy;                                     /* (the records were made with Create_QVcoding's `lossy`, as in dex2DB.c) */
  if (coding == NULL || coding->delScheme != (void *) &Tables || Records == NULL)
    die("Compress_Next_QVentry1: call Create_QVcoding first");
  if (Next >= Nent || (uint32_t) rlen != Lens[Next])
    die("Compress_Next_QVentry1: the entries must come in the order they were scanned in");
  { const char *const line[5] = { del, tag, ins, mrg, sub };      /* the reference encodes what it is handed here (QV.c:1343-1379): */
    if (entry_sum(rlen, line) != Sums[Next])                       /* other bytes than were scanned must not pass silently */
      die("Compress_Next_QVentry1: this entry's lines are not the ones QVcoding_Scan1 was given for it");
  }
  k = (size_t) (Coff[Next + 1] - Coff[Next]);
  if (k && fwrite(Records + Coff[Next], 1, k, output) != k)
    die("Compress_Next_QVentry1: write failed");
  Next += 1;
}

/* ---- decode side (undexqv.c:112-208) ------------------------------------------------------------------------ */
static uint8_t     *DImg   = NULL;        /* the whole .dexqv file */
static size_t       DImgN  = 0;
static dx_qv_index  DWalk;                /* its records (host walk) */
static int          DHave  = 0;
static uint8_t     *DText  = NULL;        /* every entry decoded: the .quiva text dx_file_undexqv makes */
static size_t       DTextN = 0, DAt = 0;  /* DAt: where the next entry's header line starts in DText */
static uint64_t     DNext  = 0;
static QVcoding     DCoding;

static void drop_decode(void)
{ free(DImg); DImg = NULL; DImgN = 0;
  if (DHave) dx_qv_index_free(&DWalk);
  DHave = 0;
  dx_file_free(DText); DText = NULL; DTextN = 0; DAt = 0; DNext = 0;
}

static void open_gpu(const char *who)
{ if (Ctx == NULL)
    { const char *d = getenv("DEXGPU_DEVICE");
      if (dx_open(d ? atoi(d) : 0, &Ctx) != DX_OK)
        { fprintf(stderr, "libdexgpu: %s: cannot open a GPU (%s)\n", who, dx_last_error(NULL));
          exit(1);
        }
    }
}

QVcoding *Read_QVcoding(FILE *input)
{ long   end;
  size_t plen;
  drop_decode();
  if (input == NULL || fseek(input, 0, SEEK_END) != 0 || (end = ftell(input)) < 0 || fseek(input, 0, SEEK_SET) != 0)
    die("Read_QVcoding: the input must be a seekable file (the whole of it is decoded at once)");
  DImgN = (size_t) end;
  DImg  = malloc(DImgN ? DImgN : 1);
  if (DImg == NULL) die("Out of memory (Read_QVcoding)");
  if (DImgN && fread(DImg, 1, DImgN, input) != DImgN) die("Read_QVcoding: read failed");
  if (dx_qv_walk(DImg, DImgN, &DWalk) != DX_OK) die("Read_QVcoding: not a .dexqv file, or a damaged one");
  DHave = 1;
  open_gpu("Read_QVcoding");
  if (dx_file_undexqv(Ctx, DImg, DImgN, /*upper*/ 0, &DText, &DTextN) != DX_OK)
    { fprintf(stderr, "libdexgpu: Read_QVcoding: %s\n", dx_last_error(Ctx));
      exit(1);
    }
  if (fseek(input, (long) DWalk.rec_off[0], SEEK_SET) != 0) die("Read_QVcoding: seek failed");   /* behind the coding, as QV.c:1214-1320 leaves it */
  memset(&DCoding, 0, sizeof(DCoding));
  DCoding.delScheme = &DWalk;                           /* opaque to the caller */
  DCoding.delChar   = DWalk.coding.delChar;
  DCoding.subChar   = DWalk.coding.subChar;
  DCoding.flip      = DWalk.flip;
  plen = strlen(DWalk.prefix);
  DCoding.prefix = malloc(plen + 1);                    /* the caller's to keep until Free_QVcoding, QV.c:1256-1265 */
  if (DCoding.prefix == NULL) die("Out of memory (Read_QVcoding)");
  memcpy(DCoding.prefix, DWalk.prefix, plen + 1);
  return &DCoding;
}

int Uncompress_Next_QVentry(FILE *input, char **entry, QVcoding *coding, int rlen)
{ const uint64_t k = DNext;
  const uint8_t *nl;
  size_t at;
  long   pos;
  int    e;
  if (coding == NULL || coding->delScheme != (void *) &DWalk || !DHave)
    die("Uncompress_Next_QVentry: call Read_QVcoding first");
  if (k >= DWalk.n) die("Uncompress_Next_QVentry: no more entries in this file");
  if ((uint32_t) rlen != DWalk.len[k])
    die("Uncompress_Next_QVentry: rlen is not the length of the entry that stands here");
  pos = ftell(input);                                   /* the caller has read this record's framing bytes, no more, no less */
  if (pos < 0 || (uint64_t) pos != DWalk.rec_off[k] + (DWalk.hdr_off[k + 1] - DWalk.hdr_off[k]))
    die("Uncompress_Next_QVentry: the stream does not stand at the start of the next entry's segments");
  nl = memchr(DText + DAt, '\n', DTextN - DAt);          /* the entry's header line, then its five lines */
  if (nl == NULL) die("Uncompress_Next_QVentry: internal: decoded text ends early");
  at = (size_t) (nl - DText) + 1;
  if (at + 5 * ((size_t) rlen + 1) > DTextN) die("Uncompress_Next_QVentry: internal: decoded text ends early");
  for (e = 0; e < 5; e++)
    memcpy(entry[e], DText + at + (size_t) e * ((size_t) rlen + 1), (size_t) rlen);
  DAt   = at + 5 * ((size_t) rlen + 1);
  DNext = k + 1;
  if (fseek(input, (long) DWalk.rec_off[k + 1], SEEK_SET) != 0) die("Uncompress_Next_QVentry: seek failed");
  return 0;
}

void Free_QVcoding(QVcoding *coding)                /* QV.c:1324-1334: the auxiliary storage, not the object */
{ if (coding != NULL)
    { free(coding->prefix);
      coding->prefix = NULL;
    }
  if (coding == &DCoding)                           /* a coding Read_QVcoding made: the decode side's state goes */
    { drop_decode();
      return;
    }
  drop_results();
  free(Lens); Lens = NULL; free(Sums); Sums = NULL; LensCap = 0; Nent = 0;
}
