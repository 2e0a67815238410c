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
  free(DText); DText = NULL; DTextN = 0; DAt = 0; DNext = 0;
}

static int text_to_memory(void *user, uint8_t *data, size_t len, size_t at) { memcpy((uint8_t *) user + at, data, len); return 0; }

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
  open_gpu("Read_QVcoding");
  { /* the records walked where dx_file_undexqv walks them (on the device for a large file), once: the plan's index is the shim's */
    dx_undexqv_plan *plan = NULL;
    size_t total = 0;
    if (dx_file_undexqv_plan_on(Ctx, DImg, DImgN, &plan, &total) != DX_OK) die("Read_QVcoding: not a .dexqv file, or a damaged one");
    if (dx_file_undexqv_plan_index(plan, &DWalk) != DX_OK) die("Out of memory (Read_QVcoding)");
    DHave = 1;
    DText = malloc(total + 16);
    DTextN = total;
    if (DText == NULL) die("Out of memory (Read_QVcoding)");
    if (dx_file_undexqv_run(Ctx, plan, /*upper*/ 0, text_to_memory, DText) != DX_OK)
      { fprintf(stderr, "libdexgpu: Read_QVcoding: %s\n", dx_last_error(Ctx));
        exit(1);
      }
    dx_file_undexqv_plan_free(plan);
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

/* ==========================================================================================
 *  DB.h:257-267: the per-read helpers, one in-memory string at a time
 * ==========================================================================================
 * Compress_Read / Uncompress_Read / Lower_Read / Upper_Read / Number_Read / Letter_Arrow / Number_Arrow under their own
 * names, over the 2-bit kernels (dx_pack2_encode / dx_pack2_decode with no framing; the alphabets DX_ALPHA_NUMBERS and
 * DX_LETTERS_NUMBERS exist for these: a string that is already numbers, numbers wanted back).  Every call is a round trip to
 * the GPU for ONE read -- a caller with many reads wants the batch calls --, but a program written against
 * DB.h links and gets the reference's bytes.  A letters-to-letters call is two kernels (pack, unpack).  Not what the
 * reference does: bytes of '\n' inside a string given to Number_Read / Number_Arrow (the packer drops line ends; the
 * reference maps them to 0 / 3): such a call dies.  Print_Read and Change_Read are host formatting, not provided.       */
static void  *RdDev[6];                   /* text, its three index words, packed, its offsets, letters out, their offsets */
static size_t RdCap = 0;

static void read_room(const char *who, size_t len)
{ int k;
  open_gpu(who);
  if (len + 64 <= RdCap) return;
  for (k = 0; k < 6; k++) if (RdDev[k]) { dx_free(Ctx, RdDev[k]); RdDev[k] = NULL; }
  RdCap = 2 * len + 4096;
  if (dx_malloc(Ctx, RdCap, &RdDev[0]) != DX_OK || dx_malloc(Ctx, 64, &RdDev[1]) != DX_OK ||
      dx_malloc(Ctx, RdCap / 4 + 64, &RdDev[2]) != DX_OK || dx_malloc(Ctx, 64, &RdDev[3]) != DX_OK ||
      dx_malloc(Ctx, RdCap + 64, &RdDev[4]) != DX_OK || dx_malloc(Ctx, 64, &RdDev[5]) != DX_OK)
    die("out of device memory (a read)");
}

/* s[0, len) through the packer (alpha >= 0: s holds text / numbers; < 0: s holds the packed bytes already) and, letters >= 0,
   back through the unpacker; the result into s (packed bytes when letters < 0)                                         */
static void read_through(const char *who, char *s, size_t len, int alpha, int letters)
{ const size_t clen = (len + 3) >> 2;
  uint64_t off[2], coff[2], toff[2];
  uint32_t idx[2];
  if (len == 0) return;
  if (len >= (1u << 31)) die("a read of 2^31 symbols and more");
  read_room(who, len);
  off[0] = 0; off[1] = len; coff[0] = 0; coff[1] = clen; toff[0] = 0; toff[1] = len + 1;
  idx[0] = (uint32_t) len; idx[1] = (uint32_t) len;
  if (dx_h2d(Ctx, RdDev[1], off, 16) != DX_OK || dx_h2d(Ctx, (char *) RdDev[1] + 16, idx, 8) != DX_OK ||
      dx_h2d(Ctx, RdDev[3], coff, 16) != DX_OK || dx_h2d(Ctx, RdDev[5], toff, 16) != DX_OK)
    die(dx_last_error(Ctx));
  if (alpha >= 0)
    { if (dx_h2d(Ctx, RdDev[0], s, len) != DX_OK ||
          dx_pack2_encode(Ctx, alpha, RdDev[0], RdDev[1], (const uint32_t *) ((char *) RdDev[1] + 16), (const uint32_t *) ((char *) RdDev[1] + 20), 1,
                          NULL, NULL, RdDev[2], RdDev[3]) != DX_OK)
        die(dx_last_error(Ctx));
    }
  else if (dx_h2d(Ctx, RdDev[2], s, clen) != DX_OK)
    die(dx_last_error(Ctx));
  if (letters < 0)
    { if (dx_d2h(Ctx, s, RdDev[2], clen) != DX_OK) die(dx_last_error(Ctx));
      return;
    }
  if (dx_pack2_decode(Ctx, letters, RdDev[2], RdDev[3], (const uint32_t *) ((char *) RdDev[1] + 20), 1, (uint32_t) len, RdDev[4], RdDev[5]) != DX_OK ||
      dx_d2h(Ctx, s, RdDev[4], len) != DX_OK)
    die(dx_last_error(Ctx));
}

void Compress_Read(int len, char *s)   { if (len > 0) read_through("Compress_Read", s, (size_t) len, DX_ALPHA_NUMBERS, -1); }            /* DB.c:319-338 */
void Uncompress_Read(int len, char *s) { if (len > 0) read_through("Uncompress_Read", s, (size_t) len, -1, DX_LETTERS_NUMBERS); s[len > 0 ? len : 0] = 4; }   /* DB.c:342-363 */

static size_t numbers_len(const char *s) { size_t n = 0; while (s[n] != 4) n++; return n; }

void Lower_Read(char *s)   { const size_t n = numbers_len(s); read_through("Lower_Read", s, n, DX_ALPHA_NUMBERS, DX_LETTERS_LOWER); s[n] = '\0'; }     /* DB.c:367 */
void Upper_Read(char *s)   { const size_t n = numbers_len(s); read_through("Upper_Read", s, n, DX_ALPHA_NUMBERS, DX_LETTERS_UPPER); s[n] = '\0'; }     /* DB.c:375 */
void Letter_Arrow(char *s) { const size_t n = numbers_len(s); read_through("Letter_Arrow", s, n, DX_ALPHA_NUMBERS, DX_LETTERS_ARROW); s[n] = '\0'; }   /* DB.c:383 */
void Number_Read(char *s)  { const size_t n = strlen(s); read_through("Number_Read", s, n, DX_ALPHA_BASES, DX_LETTERS_NUMBERS); s[n] = 4; }             /* DB.c:393 */
void Number_Arrow(char *s) { const size_t n = strlen(s); read_through("Number_Arrow", s, n, DX_ALPHA_ARROW, DX_LETTERS_NUMBERS); s[n] = 4; }            /* DB.c:418 */
