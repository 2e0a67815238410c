/* dexqv_like.c -- a caller of the reference's FILE * QV entry points, written for this test (not the reference's dexqv.c):
 *     dexqv_like [-l] in.quiva out.dexqv
 * scans the file (QVcoding_Scan), makes and writes the coding with the header prefix of the first entry
 * (Create_QVcoding, Write_QVcoding), then per entry writes the well / beg / end / qv framing and calls
 * Compress_Next_QVentry -- the call order of dexqv.c:81-141, through include/dexcompat.h over libdexgpu.        */
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dexcompat.h"

int main(int argc, char **argv)
{ int lossy = 0, a = 1;
  if (argc > 1 && strcmp(argv[1], "-l") == 0) { lossy = 1; a = 2; }
  if (argc - a != 2) { fprintf(stderr, "usage: dexqv_like [-l] in.quiva out.dexqv\n"); return 2; }
  FILE *in = fopen(argv[a], "r"), *out = fopen(argv[a + 1], "w");
  if (in == NULL || out == NULL) { perror("dexqv_like"); return 1; }

  Set_QV_Line(0);
  if (QVcoding_Scan(in, INT_MAX, NULL) < 0) return 1;
  QVcoding *coding = Create_QVcoding(lossy);

  rewind(in);
  Read_Lines(in, 1);                                /* the first header line: its text up to the first '/' is the prefix */
  { char *line = QVentry(), *slash = strchr(line + 1, '/');
    size_t n = (size_t) (slash - line);
    coding->prefix = malloc(n + 1);
    memcpy(coding->prefix, line, n);
    coding->prefix[n] = '\0';
  }
  const uint16_t key = 0x55aa;
  fwrite(&key, 2, 1, out);
  Write_QVcoding(out, coding);

  rewind(in);
  Set_QV_Line(0);
  int last_well = 0;
  while (Read_Lines(in, 1) > 0)
    { int well, beg, end, qv;
      sscanf(strchr(QVentry(), '/') + 1, "%d/%d_%d RQ=0.%d\n", &well, &beg, &end, &qv);
      for (; well - last_well >= 255; last_well += 255) fputc(0xff, out);
      fputc(well - last_well, out);
      last_well = well;
      fwrite(&beg, sizeof(int), 1, out);
      fwrite(&end, sizeof(int), 1, out);
      fwrite(&qv, sizeof(int), 1, out);
      Compress_Next_QVentry(in, out, coding, lossy);
    }
  Free_QVcoding(coding);
  fclose(in);
  return fclose(out) == 0 ? 0 : 1;
}
