/*
 * dexcompat.h -- the reference's per-entry QV entry points under their own names, over libdexgpu.
 *
 * dex2DB.c:511-643, 732-839 builds a .qvs track by calling, per read,
 *     QVcoding_Scan1 (QV.h:61)  ->  Create_QVcoding (QV.h:68)  ->  Write_QVcoding (QV.h:75)  ->
 *     Compress_Next_QVentry1 (QV.h:87) per read again  ->  Free_QVcoding (QV.h:79).
 * A GPU cannot be fed one entry at a time, so these shims gather: QVcoding_Scan1 copies the entry into a
 * batch (dx_entries_add), Create_QVcoding scans AND compresses the whole batch on the GPU
 * (dx_entries_compress: dx_qv_prescan / dx_qv_hist / dx_qv_build / dx_qv_encode_onepass), and
 * Compress_Next_QVentry1 writes the already finished bytes of the next entry.  The caller contract is the
 * reference's own use (dex2DB.c): the second pass presents the entries of the first, in the same order.
 * Same static-state model as QV.c (one scan at a time per process); errors follow the reference's batch
 * convention (message on stderr, exit(1); DB.h:45-47).  The GPU is device DEXGPU_DEVICE (default 0).
 */
#ifndef DEXCOMPAT_H
#define DEXCOMPAT_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct                 /* QV.h:31-42, field for field (the scheme pointers are opaque there too) */
  { void    *delScheme;
    void    *insScheme;
    void    *mrgScheme;
    void    *subScheme;
    void    *dRunScheme;
    void    *sRunScheme;
    int      delChar;
    int      subChar;
    int      flip;
    char    *prefix;
  } QVcoding;

/* The FILE * entry points as dexqv.c:81-141 calls them: QVcoding_Scan reads up to num entries (copying them to temp if
 * given) and gathers them the way QVcoding_Scan1 does; Compress_Next_QVentry reads the next entry's five lines and writes
 * the record Create_QVcoding made of them (the second pass presents the entries of the first, in the same order: else
 * it dies).  Read_Lines / QVentry / Set_QV_Line / Get_QV_Line: the reference's line reader, line j of a call at
 * QVentry() + j * stride; -1 at the end of the input, -2 (message on stderr) on an error.                          */
int       Read_Lines(FILE *input, int nlines);                                               /* QV.c:751-798 */
char     *QVentry(void);                                                                     /* QV.c:737 */
void      Set_QV_Line(int line);                                                             /* QV.c:740 */
int       Get_QV_Line(void);                                                                 /* QV.c:743 */
int       QVcoding_Scan(FILE *input, int num, FILE *temp);                                   /* QV.c:922-1023 */
int       Compress_Next_QVentry(FILE *input, FILE *output, QVcoding *coding, int lossy);     /* QV.c:1381-1426 */

/* Differences in side effects (the reference's own callers do not depend on them): the reference's compress calls rewrite the
 * caller's buffers in place -- tag becomes the packed 2-bit codes (Number_Read / Compress_Read), ins / mrg are rounded when
 * lossy (QV.c:1355-1372) -- these leave them untouched; Read_QVcoding wants the stream at offset 0 or 2 of a .dexqv file
 * and refuses a second coding while one is live (the reference reads a coding wherever the stream stands).           */
void      QVcoding_Scan1(int rlen, char *del, char *tag, char *ins, char *mrg, char *sub);   /* QV.c:866-920; rlen == 0: reset */
QVcoding *Create_QVcoding(int lossy);                                                        /* QV.c:1029-1169 */
void      Write_QVcoding(FILE *output, QVcoding *coding);                                    /* QV.c:1173-1210 */
void      Compress_Next_QVentry1(int rlen, char *del, char *tag, char *ins, char *mrg, char *sub,
                                 FILE *output, QVcoding *coding, int lossy);                 /* QV.c:1343-1379 */
void      Free_QVcoding(QVcoding *coding);                                                   /* QV.c:1324-1334 */

/* The decode side, as undexqv.c:112-208 uses it: Read_QVcoding after the caller has read (or not found) the 0x55aa key,
 * then per entry the caller's own reads of the framing bytes and one Uncompress_Next_QVentry.  A GPU cannot be fed an
 * entry at a time here either: Read_QVcoding takes the WHOLE file from `input` (which must be seekable), walks it and
 * decodes every entry on the GPU at once (dx_qv_walk + the decode kernels); Uncompress_Next_QVentry hands out the next
 * entry's five lines (tags in lower case, as the reference leaves them) and leaves `input` at the next record's framing
 * bytes, exactly where the reference's reads would have left it.  It checks that the stream stands where this entry's
 * segments start and that rlen is the entry's length; one file at a time per process.                          */
QVcoding *Read_QVcoding(FILE *input);                                                        /* QV.c:1214-1320 */
int       Uncompress_Next_QVentry(FILE *input, char **entry, QVcoding *coding, int rlen);    /* QV.c:1428-1481 */

/* DB.h:257-267, the per-read helpers: one in-memory string at a time through the 2-bit kernels (a GPU round trip per call: a
 * caller with many reads wants dx_pack2_encode / dx_pack2_decode on a batch; these are for programs written against DB.h).
 * Same results as the reference's for strings of its own alphabets; a '\n' inside a string given to Number_Read / Number_Arrow
 * dies (the packer drops line ends).  Print_Read and Change_Read (formatting) are not provided.                        */
#define COMPRESSED_LEN(len)  (((len)+3) >> 2)
void   Compress_Read(int len, char *s);   /* DB.c:319-338: numbers 0..3 -> 2-bit form, in place */
void Uncompress_Read(int len, char *s);   /* DB.c:342-363: 2-bit form -> numbers, s[len] = 4 */
void Lower_Read(char *s);                 /* DB.c:367: numbers (terminated by 4) -> acgt */
void Upper_Read(char *s);                 /* DB.c:375: -> ACGT */
void Number_Read(char *s);                /* DB.c:393: letters -> numbers, terminated by 4 */
void Letter_Arrow(char *s);               /* DB.c:383: numbers -> 1234 */
void Number_Arrow(char *s);               /* DB.c:418: 1234 -> numbers */

#ifdef __cplusplus
}
#endif
#endif
