#define _GNU_SOURCE
/*
 * cli_common.c -- command-line surface of the reference's six codec tools, on top of libdexgpu.
 *
 * Same flags, usage text, file naming (<dir>/<root><.ext>), -i pipe mode, -v messages, source
 * removal unless -k, error messages and exit codes as dexta.c / undexta.c / dexar.c / undexar.c /
 * dexqv.c / undexqv.c (argument macros DB.h:79-123, path helpers DB.c:112-181).  The file is
 * read whole, handed to the GPU through the whole-file drivers of libdexgpu, and the result is
 * written whole.  There is no CPU codec here: without a HIP device the tools fail loudly.
 */
#include <pthread.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

#include "cli_common.h"
#include "dexgpu.h"
#include "../dx_env.h"

typedef struct
  { const char *name, *flags, *src_ext, *dst_ext, *what, *usage;
    int         pipe_ok;
    const char *help[4];
  } tool_t;

static const tool_t TOOLS[6] =
  { { "dexta",   "vki",  ".fasta", ".dexta", "Fasta", "[-vk] ( -i | <path:fasta> ... )", 1,
      { "      -i: source is on standard input.\n",
        "      -k: do *not* remove the .fasta file on completion.\n",
        "      -w: line width for sequence lines.\n", NULL } },
    { "undexta", "vkiU", ".dexta", ".fasta", "dexta", "[-vkU] [-w<int(80)>] ( -i | <path:dexta> ... )", 1,
      { "      -i: source is on standard input.\n",
        "      -k: do *not* remove the .dexta file on completion.\n",
        "      -U: use uppercase letters (default is lower case).\n",
        "      -w: line width for sequence lines.\n" } },
    { "dexar",   "vki",  ".arrow", ".dexar", "Arrow", "[-vk] ( -i | <path:arrow> ... )", 1,
      { "      -i: source is on standard input.\n",
        "      -k: do *not* remove the .arrow file on completion.\n", NULL, NULL } },
    { "undexar", "vki",  ".dexar", ".arrow", "dexar", "[-vk] [-w<int(80)>] ( -i | <path:dexar> ... )", 1,
      { "      -i: source is on standard input.\n",
        "      -k: do *not* remove the .dexar file on completion.\n",
        "      -w: line width for arrow lines.\n", NULL } },
    { "dexqv",   "vkl",  ".quiva", ".dexqv", "quiva", "[-vkl] <path:quiva> ...", 0,
      { "      -k: do *not* remove the .quiva file on completion.\n",
        "      -l: use lossy compression (not recommended).\n", NULL, NULL } },
    { "undexqv", "vkU",  ".dexqv", ".quiva", "dexqv", "[-vkU] <path:dexqv> ...", 0,
      { "      -k: do *not* remove the .dexqv file on completion.\n",
        "      -U: use uppercase letters (default is lower case).\n", NULL, NULL } } };

static const char *Prog;

/* ---- whole-file I/O ---------------------------------------------------------------------- */

/* The whole input as one buffer.  A regular file is mapped (no copy through a read buffer: the
   upload to the GPU reads the page cache directly); a pipe is read to its end.  *mapped tells
   unslurp which. */
static uint8_t *slurp(FILE *f, size_t *n, int *mapped)
{ size_t cap = 1 << 20, len = 0, k;
  uint8_t *buf;
  struct stat st;
  *mapped = 0;
  if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0)
    { void *m = mmap(NULL, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
      if (m != MAP_FAILED)
        { /* pages brought in 16 MiB at a time rather than with MAP_POPULATE: one long populate holds the address
             space's lock against the GPU runtime that is starting up on the other thread (its own mmaps then
             wait for the whole read: 100 -> 185 ms of context creation beside a 1 GB file) */
#ifdef MADV_POPULATE_READ
          size_t at;
          for (at = 0; at < (size_t) st.st_size; at += (size_t) 16 << 20)
            { size_t len = (size_t) st.st_size - at < ((size_t) 16 << 20) ? (size_t) st.st_size - at : (size_t) 16 << 20;
              if (madvise((uint8_t *) m + at, len, MADV_POPULATE_READ) != 0)
                break;                                   /* (older kernel: the pages fault in when they are read) */
            }
#endif
          *mapped = 1;
          *n = (size_t) st.st_size;
          return (uint8_t *) m;
        }
      cap = (size_t) st.st_size + 1;
    }
  buf = malloc(cap);
  if (buf == NULL) return NULL;
  while ((k = fread(buf + len, 1, cap - len, f)) > 0)
    { len += k;
      if (len == cap)
        { uint8_t *nb = realloc(buf, cap *= 2);
          if (nb == NULL) { free(buf); return NULL; }
          buf = nb;
        }
    }
  *n = len;
  return buf;
}

/* The whole output image to a regular file: large images through a shared mapping of the file that several
   threads fill, a slice each (write() calls on one file queue up behind its inode lock, page faults on a
   mapping do not).  The file's pages are allocated first, in one posix_fallocate call: threads that fault
   fresh pages into one file contend for its page-cache lock (1 GiB to tmpfs, 8 threads: 2.6 s; allocated
   first: 0.09 s + 0.12 s of copying), and a full file system is an error return instead of a SIGBUS.
   Pipes, small outputs and anything that cannot be mapped go by fwrite.  0 on success.                    */
typedef struct { uint8_t *dst; const uint8_t *src; size_t n; } wjob;

static void *copy_slice(void *arg)
{ wjob *j = arg;
  memcpy(j->dst, j->src, j->n);
  return NULL;
}

/* May the output be laid out directly in the file behind f (posix_fallocate / ftruncate / pwrite at absolute
   offsets / a mapping)?  Only when the file is ours to lay out: a regular file that is empty, positioned at its
   start and not opened for appending.  `tool -i <in >>all` (O_APPEND: pwrite appends whatever the offset, ftruncate
   would cut what is there) and `1<>file` (existing bytes behind offset 0) go through fwrite like a pipe.           */
static int file_is_ours(FILE *f)
{ struct stat st;
  int fd = fileno(f), fl;
  if (fflush(f) != 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size != 0) return 0;
  if ((fl = fcntl(fd, F_GETFL)) < 0 || (fl & O_APPEND)) return 0;
  return lseek(fd, 0, SEEK_CUR) == 0;
}

static int write_image(FILE *f, const uint8_t *buf, size_t n)
{ long cores = sysconf(_SC_NPROCESSORS_ONLN);
  int  T = cores > 16 ? 16 : (int) cores, k, made = 0;
  uint8_t *map;
  if (n < ((size_t) 32 << 20) || T < 2 || !file_is_ours(f) || posix_fallocate(fileno(f), 0, (off_t) n) != 0 ||
      ftruncate(fileno(f), (off_t) n) != 0)
    return (n > 0 && fwrite(buf, 1, n, f) != n) ? -1 : 0;
  map = mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fileno(f), 0);
  if (map == MAP_FAILED)
    return (fwrite(buf, 1, n, f) != n) ? -1 : 0;
  { wjob      job[16];
    pthread_t th[16];
    size_t    slice = (n / (size_t) T + 4095) & ~(size_t) 4095;
    for (k = 0; k < T; k++)
      { size_t lo = (size_t) k * slice, hi = lo + slice < n ? lo + slice : n;
        if (lo >= n) break;
        job[k].dst = map + lo; job[k].src = buf + lo; job[k].n = hi - lo;
        if (k > 0 && pthread_create(&th[k], NULL, copy_slice, &job[k]) != 0)
          { copy_slice(&job[k]);                          /* no thread to be had: this one does the slice */
            th[k] = pthread_self();
          }
        made = k + 1;
      }
    copy_slice(&job[0]);
    for (k = 1; k < made; k++)
      if (!pthread_equal(th[k], pthread_self())) pthread_join(th[k], NULL);
  }
  if (munmap(map, n) != 0) return -1;
  return lseek(fileno(f), (off_t) n, SEEK_SET) < 0 ? -1 : 0;
}

static void unslurp(uint8_t *buf, size_t n, int mapped)
{ if (mapped) munmap(buf, n);
  else        free(buf);
}

/* directory part / root name, as PathTo and Root do (DB.c:112-160) */
static char *path_to(const char *name)
{ const char *s = strrchr(name, '/');
  char *p;
  if (s == NULL) return strdup(".");
  p = malloc((size_t) (s - name) + 1);
  memcpy(p, name, (size_t) (s - name));
  p[s - name] = '\0';
  return p;
}

static char *root_of(const char *name, const char *suffix)
{ const char *f = strrchr(name, '/');
  size_t fl, sl = strlen(suffix);
  char  *r;
  f  = f ? f + 1 : name;
  fl = strlen(f);
  r  = strdup(f);
  if (fl > sl && strcasecmp(f + (fl - sl), suffix) == 0)
    r[fl - sl] = '\0';
  return r;
}

static char *catenate(const char *dir, const char *root, const char *ext)
{ char *p = malloc(strlen(dir) + strlen(root) + strlen(ext) + 2);
  sprintf(p, "%s/%s%s", dir, root, ext);
  return p;
}

/* ---- error reporting in the reference's words ------------------------------------------------ */

static void report_text_error(int tool, uint64_t line, int code)
{ const tool_t *t = &TOOLS[tool];
  if (tool == TOOL_DEXQV)
    switch (code)
      { case DX_IDX_NO_NEWLINE: fprintf(stderr, "Line %llu: Last line does not end with a newline !\n", (unsigned long long) line); break;   /* QV.c:779 */
        case DX_IDX_NO_HEADER:  fprintf(stderr, "Line %llu: Header in quiva file is missing\n", (unsigned long long) line); break;          /* QV.c:955 */
        case DX_IDX_INCOMPLETE: fprintf(stderr, "Line %llu: incomplete last entry of .quiv file\n", (unsigned long long) line); break;      /* QV.c:789 */
        case DX_IDX_RAGGED:     fprintf(stderr, "Line %llu: Lines for an entry are not the same length\n", (unsigned long long) line); break; /* QV.c:793 */
        default:                fprintf(stderr, "%s: Line %llu: Header line incorrectly formatted ?\n", Prog, (unsigned long long) line); break; /* QV.c:960 */
      }
  else
    switch (code)
      { case DX_IDX_TOO_LONG:
        case DX_IDX_NO_NEWLINE:
        case DX_IDX_EMPTY:     fprintf(stderr, "Line %llu: %s line is too long (> %d chars)\n", (unsigned long long) line,
                                       tool == TOOL_DEXAR && line != 1 ? "Fasta" : t->what, 99998); break;   /* dexta.c:110,168; dexar.c says "Arrow" for line 1
                                                                                                                 (dexar.c:109) and "Fasta" for every other (dexar.c:175) */
        case DX_IDX_NO_HEADER: fprintf(stderr, "Line 1: First header in %s file is missing\n", tool == TOOL_DEXTA ? "fasta" : "arrow"); break;   /* dexta.c:114 */
        default:               fprintf(stderr, "%s: Header line incorrectly formatted ?\n", Prog); break;                                        /* dexta.c:120,148,154 */
      }
}

/* ---- one file --------------------------------------------------------------------------------- */

static dx_ctx *Ctxs[64];     /* DEXGPU_DEVICES: one context per listed GPU; a file's entries are sharded over them */
static int     Nctx = 0;

static int report_failure(dx_ctx *ctx, int tool, const uint8_t *in, size_t n, int rc, uint64_t line, int code);

static int convert(dx_ctx *ctx, int tool, const uint8_t *in, size_t n, int opt_U, int opt_l, int width,
                   uint8_t **out, size_t *out_len)
{ uint64_t line = 0;
  int      code = 0, rc;
  if (tool == TOOL_DEXQV && Nctx > 1)
    rc = dx_file_dexqv_sharded(Ctxs, Nctx, in, n, opt_l, out, out_len, &line, &code);
  else if ((tool == TOOL_DEXTA || tool == TOOL_DEXAR) && Nctx > 1)
    rc = dx_file_pack2_sharded(Ctxs, Nctx, tool == TOOL_DEXAR, in, n, out, out_len, &line, &code);
  else
  switch (tool)
    { case TOOL_DEXTA:   rc = dx_file_pack2(ctx, 0, in, n, out, out_len, &line, &code); break;
      case TOOL_DEXAR:   rc = dx_file_pack2(ctx, 1, in, n, out, out_len, &line, &code); break;
      case TOOL_UNDEXTA: rc = dx_file_unpack2(ctx, opt_U ? DX_LETTERS_UPPER : DX_LETTERS_LOWER, in, n, (uint32_t) width, out, out_len); break;
      case TOOL_UNDEXAR: rc = dx_file_unpack2(ctx, DX_LETTERS_ARROW, in, n, (uint32_t) width, out, out_len); break;
      case TOOL_DEXQV:   rc = dx_file_dexqv(ctx, in, n, opt_l, out, out_len, &line, &code); break;
      default:           rc = dx_file_undexqv(ctx, in, n, opt_U, out, out_len); break;
    }
  return rc == DX_OK ? 0 : report_failure(ctx, tool, in, n, rc, line, code);
}

/* sinks of dx_file_undexqv_run: the output file at its offset / a buffer */
static int sink_pwrite(void *user, uint8_t *data, size_t len, size_t at)
{ const int fd = *(int *) user;
  size_t done = 0;
  while (done < len)
    { ssize_t k = pwrite(fd, data + done, len - done, (off_t) (at + done));
      if (k <= 0) return 1;
      done += (size_t) k;
    }
  return 0;
}

/* a pipe: the chunks come in file order (dx_file_pack2_stream) */
static int sink_stream(void *user, uint8_t *data, size_t len, size_t at)
{ (void) at;
  return fwrite(data, 1, len, (FILE *) user) != len;
}

/* dx_read_fn over a descriptor: a pipe gives what it has, so ask until the want is met or the input ends */
static long read_fd(void *user, void *buf, size_t want)
{ const int fd = *(int *) user;
  size_t got = 0;
  while (got < want)
    { const ssize_t k = read(fd, (uint8_t *) buf + got, want - got);
      if (k < 0) return -1;
      if (k == 0) break;
      got += (size_t) k;
    }
  return (long) got;
}

/* ... with the input's first bytes looked at beforehand (an image's endian key, for the reference's words about a wrong one) */
typedef struct { int fd; uint8_t pre[2]; size_t npre, given; } peek_fd;
static long read_peeked(void *user, void *buf, size_t want)
{ peek_fd *p = user;
  size_t got = 0;
  while (p->given < p->npre && got < want) ((uint8_t *) buf)[got++] = p->pre[p->given++];
  if (got < want)
    { const long k = read_fd(&p->fd, (uint8_t *) buf + got, want - got);
      if (k < 0) return -1;
      got += (size_t) k;
    }
  return (long) got;
}

static int sink_memory(void *user, uint8_t *data, size_t len, size_t at)
{ memcpy((uint8_t *) user + at, data, len);
  return 0;
}

/* ... and a large output file whose size is known (undexqv: the plan's) or can be guessed (dexqv: three tenths of the text):
   a helper thread allocates its pages a stretch at a time while the text is still on its way -- one posix_fallocate call for
   20 GB takes a second during which nothing else goes on --, and the sink writes a chunk (pwrite, in order, one thread) as soon
   as the pages under it are there; what the guess leaves out the write allocates itself, what it has too much ftruncate takes
   back.  For dexqv, whose output is ready while its pages can still be laid out beside the upload (5.7 GB: 0.93 -> 0.49 s).
   (Measured and not kept for undexqv's 20 GB of text: the chunks copied into a shared mapping of the file by six threads -- every
   page of the mapping faults once: 3.1 s where one thread's pwrite takes 2.0, and several threads' pwrite queue up behind the
   inode's lock; the allocator beside the writer -- the two take turns at that lock: 2.1 - 4.5 s; the file allocated whole, then
   written, is what undexqv does: 0.9 + 2.0 s.)  A full file system is an error return of posix_fallocate or pwrite.       */
#define OUT_STRETCH ((size_t) 256 << 20)
typedef struct
  { size_t n; int fd, failed; size_t upto;
    pthread_mutex_t mx; pthread_cond_t cv; pthread_t th; int threaded;
  } outfile;

static void *outfile_alloc(void *arg)
{ outfile *o = arg;
  size_t off;
  for (off = 0; off < o->n; off += OUT_STRETCH)
    { const size_t len = o->n - off < OUT_STRETCH ? o->n - off : OUT_STRETCH;
      /* The size is a guess and laying pages out ahead is a convenience: when the file system will not (ENOSPC on an
         over-estimate, EOPNOTSUPP / EINVAL where there is no fallocate) the writer simply goes on without -- pwrite
         reports what is really wrong with the output, if anything is.                                             */
      const int bad = posix_fallocate(o->fd, (off_t) off, (off_t) len) != 0;
      pthread_mutex_lock(&o->mx);
      o->upto = bad ? o->n : off + len;
      pthread_cond_broadcast(&o->cv);
      pthread_mutex_unlock(&o->mx);
      if (bad) break;
    }
  return NULL;
}

static int outfile_begin(outfile *o, FILE *f, size_t expect)
{ memset(o, 0, sizeof(*o));
  o->fd = fileno(f); o->n = expect;
  { const size_t least = (size_t) dx_test_num("outfile_min", (long long) 1 << 30);          /* (tests: that way from this size on) */
    if (expect < least || expect == 0 || !file_is_ours(f)) return 0;
  }
  pthread_mutex_init(&o->mx, NULL);
  pthread_cond_init(&o->cv, NULL);
  o->threaded = pthread_create(&o->th, NULL, outfile_alloc, o) == 0;
  if (!o->threaded) (void) outfile_alloc(o);
  return 1;
}

static int sink_outfile(void *user, uint8_t *data, size_t len, size_t at)
{ outfile *o = user;
  const size_t want = at + len < o->n ? at + len : o->n;    /* (behind the expected size: the write allocates) */
  int bad;
  pthread_mutex_lock(&o->mx);
  while (o->upto < want && !o->failed) pthread_cond_wait(&o->cv, &o->mx);
  bad = o->failed;
  pthread_mutex_unlock(&o->mx);
  return bad ? 1 : sink_pwrite(&o->fd, data, len, at);
}

static int outfile_end(outfile *o, size_t size)             /* 0: the file is complete, `size` bytes long */
{ int bad;
  if (o->threaded) pthread_join(o->th, NULL);
  bad = o->failed;
  pthread_cond_destroy(&o->cv);
  pthread_mutex_destroy(&o->mx);
  return bad || ftruncate(o->fd, (off_t) size) != 0 || lseek(o->fd, (off_t) size, SEEK_SET) < 0;
}

static int report_failure(dx_ctx *ctx, int tool, const uint8_t *in, size_t n, int rc, uint64_t line, int code)
{ if (rc == DX_E_FORMAT && (tool == TOOL_DEXTA || tool == TOOL_DEXAR || tool == TOOL_DEXQV))
    { report_text_error(tool, line, code);
      return 1;
    }
  if (rc == DX_E_FORMAT && n >= 2 && (tool == TOOL_UNDEXTA || tool == TOOL_UNDEXAR))
    { uint16_t key;
      memcpy(&key, in, 2);
      if (key != 0x55aa && key != 0xaa55 && !(tool == TOOL_UNDEXTA && (key == 0x33cc || key == 0xcc33)))
        { fprintf(stderr, "%s: Not a .%s file, endian key invalid\n", Prog, TOOLS[tool].what);   /* undexta.c:156 */
          return 1;
        }
    }
  if (rc == DX_E_FORMAT)
    { fprintf(stderr, "%s: System error, read failed!\n", Prog);                                 /* DB.h:136-139 */
      return 2;
    }
  { const char *why = dx_last_error(ctx);
    if (why == NULL || why[0] == '\0')                    /* (host-side failures carry a code only) */
      why = rc == DX_E_DEGENERATE ? "a stream that needs a Huffman scheme holds no symbols (e.g. a deletion line of nothing but its run "
                                    "character, or an empty file): the reference reads out of bounds there (QV.c:201), nothing is written"
          : rc == DX_E_NOMEM      ? "out of memory"
          : rc == DX_E_UNSUPPORTED ? "a code longer than 16 bits: the reference would write a file its own decoder cannot read"
          : "failed";
    fprintf(stderr, "%s: %s (libdexgpu error %d)\n", Prog, why, rc);
  }
  return 1;
}

/* The GPU contexts are opened on a second thread while the first input file is being read
   (HIP initialisation and a large read each take a few hundred milliseconds). */
static dx_ctx *Ctx0 = NULL;
static void *open_contexts(void *arg)
{ dx_ctx *ctx = NULL;
  int     k;
  const char *dev = getenv("DEXGPU_DEVICE"), *devs = getenv("DEXGPU_DEVICES");
  (void) arg;
  if (devs != NULL && *devs != '\0')              /* "all" or a comma list, e.g. 0,1,2,3 */
    { if (strcmp(devs, "all") == 0)
        { int nd = dx_device_count();
          for (k = 0; k < nd && k < 64; k++)
            if (dx_open(k, &Ctxs[Nctx]) == DX_OK) Nctx += 1;
        }
      else
        { const char *q = devs;
          while (*q != '\0' && Nctx < 64)
            { char *e;
              long  d = strtol(q, &e, 10);
              if (e == q) break;
              if (dx_open((int) d, &Ctxs[Nctx]) != DX_OK)
                { fprintf(stderr, "%s: cannot open GPU %ld: %s\n", Prog, d, dx_last_error(NULL));
                  exit(1);
                }
              Nctx += 1;
              q = (*e == ',') ? e + 1 : e;
            }
        }
      if (Nctx > 0) ctx = Ctxs[0];
    }
  if (ctx == NULL && dx_open(dev ? atoi(dev) : 0, &ctx) != DX_OK)
    { fprintf(stderr, "%s: cannot open a GPU: %s\n", Prog, dx_last_error(NULL));
      exit(1);
    }
  Ctx0 = ctx;
  return NULL;
}

static pthread_t Opener;
static int       Opening = 0;

/* leave only after the opener thread is done: exit() while HIP initialises on another thread is unsafe */
static void leave(int code)
{ if (Opening)
    { pthread_join(Opener, NULL);
      Opening = 0;
    }
  exit(code);
}

/* DEXGPU_TIMING=1: wall-clock marks on stderr (where an end-to-end run spends its time) */
#include <time.h>
static void tmark(const char *what)
{ static double t0 = -1.0;
  struct timespec ts;
  double now;
  if (getenv("DEXGPU_TIMING") == NULL) return;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  now = ts.tv_sec + 1e-9 * ts.tv_nsec;
  if (t0 < 0) t0 = now;
  fprintf(stderr, "[%s %8.1f ms] %s\n", Prog, (now - t0) * 1e3, what);
}

int dex_tool_main(int tool, int argc, char *argv[])
{ const tool_t *t = &TOOLS[tool];
  int     flags[128], i, j, k, width = 80;
  int     VERBOSE, KEEP, PIPE, UPPER, LOSSY;
  dx_ctx *ctx = NULL;

  Prog = t->name;
  memset(flags, 0, sizeof(flags));

  j = 1;                                                   /* ARG_INIT / ARG_FLAGS, DB.h:79-91 */
  for (i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      { if (argv[i][1] == 'w' && (tool == TOOL_UNDEXTA || tool == TOOL_UNDEXAR))
          { char *eptr;                                    /* ARG_NON_NEGATIVE, DB.h:105-115 */
            width = (int) strtol(argv[i] + 2, &eptr, 10);
            if (*eptr != '\0' || argv[i][2] == '\0')
              { fprintf(stderr, "%s: -%c '%s' argument is not an integer\n", Prog, argv[i][1], argv[i] + 2);
                exit(1);
              }
            if (width < 0)
              { fprintf(stderr, "%s: %s must be non-negative (%d)\n", Prog, "Line width", width);
                exit(1);
              }
            continue;
          }
        for (k = 1; argv[i][k] != '\0'; k++)
          { if (strchr(t->flags, argv[i][k]) == NULL)
              { fprintf(stderr, "%s: -%c is an illegal option\n", Prog, argv[i][k]);
                exit(1);
              }
            flags[(int) argv[i][k]] = 1;
          }
      }
    else
      argv[j++] = argv[i];
  argc = j;

  VERBOSE = flags['v'];
  KEEP    = flags['k'];
  PIPE    = flags['i'];
  UPPER   = flags['U'];
  LOSSY   = flags['l'];

  if ((t->pipe_ok && ((PIPE && argc > 1) || (!PIPE && argc <= 1))) || (!t->pipe_ok && argc == 1))
    { fprintf(stderr, "Usage: %s %s\n", Prog, t->usage);   /* e.g. dexta.c:47-54 */
      fprintf(stderr, "\n");
      for (k = 0; k < 4; k++)
        if (t->help[k] != NULL)
          fprintf(stderr, "%s", t->help[k]);
      exit(1);
    }
  if (PIPE)
    { KEEP = 1;
      argc = 2;
    }
  if (width == 0)
    { fprintf(stderr, "%s: Line width must be positive (the reference never terminates on -w0)\n", Prog);
      exit(1);
    }

  tmark("start");
#ifdef F_SETPIPE_SZ
  if (PIPE) (void) fcntl(0, F_SETPIPE_SZ, 1 << 20);        /* (a pipe's 64 KB are 2 GB/s at best; no harm where stdin is none or the size is refused) */
#endif
  Opening = pthread_create(&Opener, NULL, open_contexts, NULL) == 0;
  if (!Opening) open_contexts(NULL);
  for (i = 1; i < argc; i++)
    { char    *pwd = NULL, *root, *src = NULL, *dst = NULL;
      FILE    *input, *output;
      uint8_t *in, *out = NULL;
      size_t   n = 0, out_len = 0;
      int      st, mapped = 0;

      if (PIPE)
        { input  = stdin;
          output = stdout;
          root   = strdup("Standard Input");
        }
      else
        { pwd  = path_to(argv[i]);                         /* dexta.c:87-94 */
          root = root_of(argv[i], t->src_ext);
          src  = catenate(pwd, root, t->src_ext);
          dst  = catenate(pwd, root, t->dst_ext);
          if ((input = fopen(src, "r")) == NULL)
            { fprintf(stderr, "%s: Cannot open %s for 'r'\n", Prog, src);   /* Fopen, DB.c:103-110 */
              leave(1);
            }
          if ((output = fopen(dst, "w+")) == NULL)           /* (readable too: a large output is written through a shared mapping, which wants that) */
            { fprintf(stderr, "%s: Cannot open %s for 'w'\n", Prog, dst);
              leave(1);
            }
        }

      if (VERBOSE)
        { fprintf(stderr, "Processing '%s' ...\n", root);
          fflush(stderr);
        }

      if (tool == TOOL_DEXQV && Nctx <= 1 && !PIPE)
        { /* a large .quiva is read from its file straight into the buffers that go to the GPU (dx_file_dexqv_fd_to): no image
             of it in this process, whose pages a mapping brings in one by one and gives back one by one (a second and a half
             of the two and a half a 20 GB file took) */
          struct stat st;
          const off_t least = (off_t) dx_test_num("fd_min", (long long) 256 << 20);         /* (tests: that way from this size on) */
          if (fstat(fileno(input), &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= least && st.st_size > 0 && file_is_ours(output))
            { uint64_t line = 0;
              int      code = 0, rc, fd = fileno(output);
              if (Opening)
                { pthread_join(Opener, NULL);
                  Opening = 0;
                }
              ctx = Ctx0;
              tmark("GPU context open");
              { outfile of;                                 /* (a .dexqv is about three tenths of its .quiva) */
                if (outfile_begin(&of, output, (size_t) st.st_size / 10 * 3))
                  { rc = dx_file_dexqv_fd_to(ctx, fileno(input), (size_t) st.st_size, LOSSY, sink_outfile, &of, &out_len, &line, &code);
                    if (outfile_end(&of, rc == DX_OK ? out_len : 0) && rc == DX_OK) rc = DX_E_IO;
                  }
                else
                  { rc = dx_file_dexqv_fd_to(ctx, fileno(input), (size_t) st.st_size, LOSSY, sink_pwrite, &fd, &out_len, &line, &code);
                    if (rc == DX_OK && lseek(fd, (off_t) out_len, SEEK_SET) < 0) rc = DX_E_IO;
                  }
              }
              if (rc == DX_E_IO)
                { fprintf(stderr, "%s: System error, write failed!\n", Prog);
                  leave(2);
                }
              if (rc == DX_OK)
                { tmark("output written");
                  goto written;
                }
              if (rc != DX_E_AGAIN)
                leave(report_failure(ctx, tool, NULL, 0, rc, line, code));
            }                                              /* (DX_E_AGAIN: through memory, below) */
        }
      if ((tool == TOOL_DEXTA || tool == TOOL_DEXAR) && Nctx <= 1)
        { /* dexta -i / dexar -i, and a file too large to hold beside its image (8 GiB and more): the text goes through the device a
             chunk of whole records at a time (dx_file_pack2_stream) -- as the reference reads record after record (dexta.c:104-205),
             with a chunk's memory whatever the input's size.  (Smaller files stay whole: a mapping, one upload, one pass -- 0.9 s
             for 4 GB where the pieces, one after the other, take 1.6.) */
          struct stat st;
          const off_t least = (off_t) dx_test_num("fd_min", (long long) 8 << 30);
          const int   isfile = fstat(fileno(input), &st) == 0 && S_ISREG(st.st_mode);
          if (PIPE || (isfile && st.st_size >= least && file_is_ours(output)))
            { uint64_t line = 0;
              int      code = 0, rc, fdin = fileno(input), fdout = fileno(output);
              if (Opening)
                { pthread_join(Opener, NULL);
                  Opening = 0;
                }
              ctx = Ctx0;
              tmark("GPU context open");
              if (PIPE || !file_is_ours(output))
                rc = dx_file_pack2_stream(ctx, tool == TOOL_DEXAR, read_fd, &fdin, (size_t) dx_test_num("stream_chunk", (long long) 64 << 20),
                                          sink_stream, output, &out_len, &line, &code);          /* (a pipe: 64 MiB at a time -- the memory stays small, the pipe sets the pace) */
              else
                { rc = dx_file_pack2_stream(ctx, tool == TOOL_DEXAR, read_fd, &fdin, 0, sink_pwrite, &fdout, &out_len, &line, &code);
                  if (rc == DX_OK && lseek(fdout, (off_t) out_len, SEEK_SET) < 0) rc = DX_E_IO;
                }
              if (rc == DX_E_IO)
                { fprintf(stderr, "%s: System error, write failed!\n", Prog);
                  leave(2);
                }
              if (rc != DX_OK)
                leave(report_failure(ctx, tool, NULL, 0, rc, line, code));
              tmark("output written");
              goto written;
            }
        }
      if ((tool == TOOL_UNDEXTA || tool == TOOL_UNDEXAR) && PIPE)
        { /* undexta -i / undexar -i: the image through the device a chunk of whole records at a time, their text out in order
             (the reference reads and writes record after record, undexta.c:175-271) */
          int     rc;
          peek_fd pk = { fileno(input), { 0, 0 }, 0, 0 };
          { const long k = read_fd(&pk.fd, pk.pre, 2);
            pk.npre = k > 0 ? (size_t) k : 0;
          }
          if (Opening)
            { pthread_join(Opener, NULL);
              Opening = 0;
            }
          ctx = Ctx0;
          tmark("GPU context open");
          rc = dx_file_unpack2_stream(ctx, tool == TOOL_UNDEXAR ? DX_LETTERS_ARROW : (UPPER ? DX_LETTERS_UPPER : DX_LETTERS_LOWER),
                                      read_peeked, &pk, 0, (uint32_t) width, sink_stream, output, &out_len);
          if (rc == DX_E_IO)
            { fprintf(stderr, "%s: System error, write failed!\n", Prog);
              leave(2);
            }
          if (rc != DX_OK)
            leave(report_failure(ctx, tool, pk.pre, pk.npre, rc, 0, 0));
          tmark("output written");
          goto written;
        }
      in = slurp(input, &n, &mapped);
      tmark("input read");
      if (tool == TOOL_UNDEXQV && in != NULL && Nctx <= 1)
        { /* The text goes from the GPU into the output file chunk by chunk (no image of it in this process):
             its size comes from the host walk over the record stream, which runs while the GPU context is
             still being opened, and so does the allocation of the file's pages. */
          dx_undexqv_plan *plan = NULL;
          int              rc, direct, fd = fileno(output);
          if (n >= ((size_t) 256 << 20))                  /* a large file: its records are walked on the GPU (dx_file_undexqv_plan_on) */
            { if (Opening)
                { pthread_join(Opener, NULL);
                  Opening = 0;
                }
              rc = dx_file_undexqv_plan_on(Ctx0, in, n, &plan, &out_len);
            }
          else
            rc = dx_file_undexqv_plan(in, n, &plan, &out_len);
          tmark("records walked");
          if (rc == DX_OK)
            { direct = file_is_ours(output) &&
                       (out_len == 0 || posix_fallocate(fd, 0, (off_t) out_len) == 0) && ftruncate(fd, (off_t) out_len) == 0;
              tmark("output file allocated");
              if (Opening)
                { pthread_join(Opener, NULL);
                  Opening = 0;
                }
              ctx = Ctx0;
              tmark("GPU context open");
              if (direct)
                { rc = dx_file_undexqv_run(ctx, plan, UPPER, sink_pwrite, &fd);
                  if (rc == DX_OK && lseek(fd, (off_t) out_len, SEEK_SET) < 0) rc = DX_E_IO;
                }
              else                                        /* a pipe: through memory */
                { out = malloc(out_len + 16);
                  rc  = out == NULL ? DX_E_NOMEM : dx_file_undexqv_run(ctx, plan, UPPER, sink_memory, out);
                  if (rc == DX_OK && out_len > 0 && fwrite(out, 1, out_len, output) != out_len) rc = DX_E_IO;
                  free(out);
                }
              dx_file_undexqv_plan_free(plan);
            }
          if (rc == DX_E_IO)
            { fprintf(stderr, "%s: System error, write failed!\n", Prog);
              leave(2);
            }
          if (rc != DX_OK)
            leave(report_failure(Ctx0, tool, in, n, rc, 0, 0));
          unslurp(in, n, mapped);
          tmark("output written");
          goto written;
        }
      if (Opening)
        { pthread_join(Opener, NULL);
          Opening = 0;
        }
      ctx = Ctx0;
      tmark("GPU context open");
      if (in == NULL)
        { fprintf(stderr, "%s: Out of memory (Allocating read buffer)\n", Prog);
          leave(1);
        }
      if ((tool == TOOL_DEXQV && Nctx <= 1) || tool == TOOL_UNDEXTA || tool == TOOL_UNDEXAR)
        { /* the output goes from the GPU into the output file chunk by chunk, if that is a regular file of ours */
          int         fd = fileno(output);
          if (file_is_ours(output))
            { uint64_t line = 0;
              int      code = 0, rc;
              if (tool == TOOL_DEXQV)
                rc = dx_file_dexqv_to(ctx, in, n, LOSSY, sink_pwrite, &fd, &out_len, &line, &code);
              else
                rc = dx_file_unpack2_to(ctx, tool == TOOL_UNDEXAR ? DX_LETTERS_ARROW : (UPPER ? DX_LETTERS_UPPER : DX_LETTERS_LOWER),
                                        in, n, (uint32_t) width, sink_pwrite, &fd, &out_len);
              if (rc == DX_OK && lseek(fd, (off_t) out_len, SEEK_SET) < 0) rc = DX_E_IO;
              if (rc == DX_E_IO)
                { fprintf(stderr, "%s: System error, write failed!\n", Prog);
                  leave(2);
                }
              if (rc != DX_OK)
                leave(report_failure(ctx, tool, in, n, rc, line, code));
              unslurp(in, n, mapped);
              tmark("output written");
              goto written;
            }
        }
      st = convert(ctx, tool, in, n, UPPER, LOSSY, width, &out, &out_len);
      if (st != 0)
        leave(st);
      tmark("converted (index, copies, kernels)");
      if (write_image(output, out, out_len) != 0)
        { fprintf(stderr, "%s: System error, write failed!\n", Prog);
          leave(2);
        }
      dx_file_free(out);
      unslurp(in, n, mapped);
      tmark("output written");
written:

      if (!PIPE)
        { fclose(input);
          if (fclose(output) != 0)                         /* a deferred write error (ENOSPC, quota, NFS) surfaces here: */
            { fprintf(stderr, "%s: System error, write failed!\n", Prog);   /* the source must survive it */
              leave(2);
            }
          if (!KEEP)
            unlink(src);
        }
      else if (fflush(output) != 0)
        { fprintf(stderr, "%s: System error, write failed!\n", Prog);
          leave(2);
        }
      free(root); free(pwd); free(src); free(dst);

      if (VERBOSE)
        { fprintf(stderr, "Done\n");
          fflush(stderr);
        }
    }

  /* Every output is closed (fclose has reported what there was to report), every input is released: nothing is left but to give
     back what the process holds on the device and in the HIP runtime -- 0.2 s of a 0.45 s run on a 1 GB file (r03c_cli_timing),
     which the system does by itself when the process ends.  DEXGPU_TEARDOWN=1: the orderly way (leak checkers).               */
  if (getenv("DEXGPU_TEARDOWN") == NULL && getenv("LD_PRELOAD") == NULL && getenv("ROCP_TOOL_LIBRARIES") == NULL &&
      getenv("ROCPROFILER_REGISTER_ROOT") == NULL)        /* (a profiler, a sanitizer or a coverage run writes its output at exit) */
    { if (Opening) { pthread_join(Opener, NULL); Opening = 0; }      /* (never while HIP comes up on another thread) */
      tmark("leaving");
      fflush(NULL);
      _exit(0);
    }
  if (Nctx > 0)
    for (k = 0; k < Nctx; k++) dx_close(Ctxs[k]);
  else
    dx_close(ctx);
  exit(0);
}
