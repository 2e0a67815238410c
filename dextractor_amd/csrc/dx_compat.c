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

static void scan_reset(void)
{ dx_entries_free(Batch);
  Batch = dx_entries_new();
  if (Batch == NULL) die("Out of memory (QVcoding_Scan1)");
  Nent = 0;
  drop_results();
}

static void scan_add(int rlen, const char *del, const char *tag, const char *ins, const char *mrg, const char *sub);

void QVcoding_Scan1(int rlen, char *del, char *tag, char *ins, char *mrg, char *sub)
{ if (rlen == 0)                                    /* reset: QV.c:871-886 */
    scan_reset();
  else
    scan_add(rlen, del, tag, ins, mrg, sub);
}

static void scan_add(int rlen, const char *del, const char *tag, const char *ins, const char *mrg, const char *sub)
{ if (Batch == NULL)
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

/* ---- the FILE * entry points (dexqv.c:81-141) ------------------------------------------------------------------
 * Read_Lines / QVentry / Set_QV_Line / Get_QV_Line (QV.c:733-798): a line reader with the reference's contract -- line j
 * of a call at QVentry() + j * (its row stride), the common length returned without the line end, -1 at the end of the
 * input, -2 (and the reference's message) on an error.  QVcoding_Scan reads entries with it and GATHERS them exactly as
 * QVcoding_Scan1 does; Compress_Next_QVentry reads the next five lines and writes the record Create_QVcoding made of
 * them.  Host code only: text intake, as in cli/.                                                                */
static char *RLbuf = NULL;               /* five rows of RLmax bytes */
static int   RLmax = 0;
static int   RLline = 0;

char *QVentry(void) { return RLbuf; }
void  Set_QV_Line(int line) { RLline = line; }
int   Get_QV_Line(void) { return RLline; }

static int rl_grow(void)                 /* rows wider; row 0 keeps what it holds */
{ int   nmax = RLmax + RLmax / 2 + 10000;
  char *t    = realloc(RLbuf, 5 * (size_t) nmax);
  if (t == NULL)
    { fprintf(stderr, "libdexgpu: Out of memory (Reallocating QV entry read buffer)\n");
      return -1;
    }
  RLbuf = t; RLmax = nmax;
  return 0;
}

int Read_Lines(FILE *input, int nlines)
{ int rlen, i;
  if (RLbuf == NULL)
    { RLmax = 0;
      if (rl_grow()) return -2;
    }
  RLline += 1;
  if (fgets(RLbuf, RLmax, input) == NULL)
    return -1;
  rlen = (int) strlen(RLbuf);
  while (RLbuf[rlen - 1] != '\n')
    { if (rl_grow()) return -2;
      if (fgets(RLbuf + rlen, RLmax - rlen, input) == NULL)
        { fprintf(stderr, "Line %d: Last line does not end with a newline !\n", RLline);
          return -2;
        }
      rlen += (int) strlen(RLbuf + rlen);
    }
  for (i = 1; i < nlines; i++)
    { char *other = RLbuf + (size_t) i * RLmax;
      RLline += 1;
      if (fgets(other, RLmax, input) == NULL)
        { fprintf(stderr, "Line %d: incomplete last entry of .quiv file\n", RLline);
          return -2;
        }
      if (rlen != (int) strlen(other))
        { fprintf(stderr, "Line %d: Lines for an entry are not the same length\n", RLline);
          return -2;
        }
    }
  return rlen - 1;
}

int QVcoding_Scan(FILE *input, int num, FILE *temp)
{ int i, r = 0;
  scan_reset();                                             /* a scan starts afresh: QV.c:931-942 zeroes its histograms */
  for (i = 0; i < num; i++)
    { int   well, beg, end, qv, rlen, k;
      char *slash;
      rlen = Read_Lines(input, 1);
      if (rlen == -2) return -1;
      if (rlen < 0) break;
      if (rlen == 0 || RLbuf[0] != '@')
        { fprintf(stderr, "Line %d: Header in quiva file is missing\n", RLline);
          return -1;
        }
      slash = strchr(RLbuf + 1, '/');
      if (slash == NULL || sscanf(slash + 1, "%d/%d_%d RQ=0.%d\n", &well, &beg, &end, &qv) != 4)
        { fprintf(stderr, "libdexgpu: Line %d: Header line incorrectly formatted ?\n", RLline);
          return -1;
        }
      if (temp != NULL) fputs(RLbuf, temp);
      rlen = Read_Lines(input, 5);
      if (rlen < 0)
        { if (rlen == -1) fprintf(stderr, "Line %d: incomplete last entry of .quiv file\n", RLline);
          return -1;
        }
      if (temp != NULL)
        for (k = 0; k < 5; k++) fputs(RLbuf + (size_t) k * RLmax, temp);
      scan_add(rlen, RLbuf, RLbuf + RLmax, RLbuf + 2 * (size_t) RLmax, RLbuf + 3 * (size_t) RLmax, RLbuf + 4 * (size_t) RLmax);
      r += 1;
    }
  return r;
}

void Compress_Next_QVentry1(int rlen, char *del, char *tag, char *ins, char *mrg, char *sub,
                            FILE *output, QVcoding *coding, int lossy);

int Compress_Next_QVentry(FILE *input, FILE *output, QVcoding *coding, int lossy)
{ const int rlen = Read_Lines(input, 5);
  if (rlen < 0)
    { if (rlen == -1) fprintf(stderr, "Line %d: incomplete last entry of .quiv file\n", RLline);
      exit(1);                                              /* (the reference's batch convention: EXIT(-1), DB.h:45-47) */
    }
  Compress_Next_QVentry1(rlen, RLbuf, RLbuf + RLmax, RLbuf + 2 * (size_t) RLmax, RLbuf + 3 * (size_t) RLmax, RLbuf + 4 * (size_t) RLmax,
                         output, coding, lossy);
  return rlen;
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
{ long   end, at;
  size_t plen;
  /* The reference reads ONE coding at the stream's position and keeps one static coding object (QV.c:1214-1320); a caller
   * that reads codings from the middle of a file (DB.c:2450-2502 does, for .qvs tracks) or keeps two alive must not get
   * another file's state silently: the stream must stand at the start of a .dexqv file -- at 0, or at 2 behind the
   * 0x55aa key undexqv.c:112-118 has read -- and a second coding while one is live is refused. */
  if (DHave)
    die("Read_QVcoding: a coding of this process is still live (Free_QVcoding it first): one file at a time");
  if (input == NULL || (at = ftell(input)) < 0)
    die("Read_QVcoding: the input must be a seekable file (the whole of it is decoded at once)");
  if (at != 0 && at != 2)
    die("Read_QVcoding: the stream must stand at the start of a .dexqv file (offset 0, or 2 behind its key); codings inside other files are not supported");
  drop_decode();
  if (fseek(input, 0, SEEK_END) != 0 || (end = ftell(input)) < 0 || fseek(input, 0, SEEK_SET) != 0)
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
