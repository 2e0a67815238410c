// dx_ctx.hip -- context, device memory and per-kernel timing plumbing of libdexgpu.
#include "dx_internal.hpp"
#include <pthread.h>
#include <unistd.h>
#include <time.h>
#include <string.h>

#include <stdlib.h>

static char g_open_err[512] = "";

#undef hipMalloc
hipError_t dx_hip_malloc(void **p, size_t bytes)
{ static int poison = -2;                                   // -2: not looked up yet, -1: off
  if (poison == -2)
    { const long long v = dx_test_num("poison", -1);
      poison = v >= 0 ? (int) (v & 0xff) : -1;
    }
  // DEXGPU_TEST=fail_malloc_over=<bytes>[,fail_malloc_under=<bytes>] (tests): allocations beyond that size (and below the other) fail as
  // if the device were full (looked up every time: a test sets it around one call)
  long long fail_over = dx_test_num("fail_malloc_over", -1);
  { const long long under = dx_test_num("fail_malloc_under", -1);
    if (fail_over >= 0 && under >= 0 && (long long) bytes >= under) fail_over = -1;
  }
  const hipError_t rc = (fail_over >= 0 && bytes > (size_t) fail_over) ? hipErrorOutOfMemory : hipMalloc(p, bytes);
  if (rc != hipSuccess)                                    // the runtime keeps a failed call's error until it is read: the next
    { (void) hipGetLastError();                            // launch's hipGetLastError() would report THIS one (a caller that
      *p = NULL;                                           // goes on another way after a failed allocation must find none)
    }
  if (rc == hipSuccess && poison >= 0 && bytes)
    { (void) hipMemset(*p, poison, bytes);
      (void) hipDeviceSynchronize();
    }
  return rc;
}
#define hipMalloc(p, n) dx_hip_malloc((void **) (p), (n))

void dx_sx_drop_external(dx_ctx *ctx)
{ if (!ctx->sx.external) return;
  ctx->sx.idx = NULL; ctx->sx.off = NULL; ctx->sx.cap_idx = 0;
  (void) hipFree(ctx->sx.room); ctx->sx.room = NULL; ctx->sx.cap_entries = 0;
  ctx->sx.external = 0; ctx->sx.valid = 0; ctx->sx.walk = 0;
}

int dx_fail(dx_ctx *ctx, int code, const char *fmt, ...)
{ char *dst = ctx ? ctx->err : g_open_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *dx_last_error(const dx_ctx *ctx) { return ctx ? ctx->err : g_open_err; }

extern "C" int dx_device_count(void)
{ int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

// DEXGPU_TIMING: where opening a context spends its time (stderr)
static void open_mark(const char *what)
{ static double t0 = -1.0;
  if (getenv("DEXGPU_TIMING") == NULL) return;
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  const double now = ts.tv_sec + 1e-9 * ts.tv_nsec;
  if (t0 < 0) t0 = now;
  fprintf(stderr, "[dx_open %8.1f ms] %s\n", (now - t0) * 1e3, what);
}

extern "C" int dx_open(int device, dx_ctx **out)
{ if (out == NULL)
    return dx_fail(NULL, DX_E_ARG, "dx_open: NULL result pointer");
  *out = NULL;
  open_mark("start");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return dx_fail(NULL, DX_E_HIP, "dx_open: no HIP device available (%s)",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n)
    return dx_fail(NULL, DX_E_ARG, "dx_open: device %d out of range (0..%d)", device, n - 1);

  open_mark("device count");
  dx_ctx *ctx = new dx_ctx();
  ctx->device = device;
  ctx->err[0] = '\0';
  ctx->profiling = false;
  memset(ctx->ms, 0, sizeof(ctx->ms));
  memset(ctx->launches, 0, sizeof(ctx->launches));
  ctx->d_tok = NULL;
  ctx->d_dec = NULL;
  ctx->d_long = NULL;
  ctx->coding_set = 0;
  ctx->d_scratch = NULL;
  ctx->scratch_bytes = 0;
  ctx->d_scan = NULL;
  ctx->scan_words = 0;
  ctx->onepass_min_groups = 0;
  ctx->scratch_budget = 0;
  memset(&ctx->route, 0, sizeof(ctx->route));
  memset(&ctx->sx, 0, sizeof(ctx->sx));
  ctx->h_stage[0] = ctx->h_stage[1] = NULL; ctx->h_up = NULL; ctx->h_down = NULL; ctx->sink_threads = 1;
  ctx->d_hscr = NULL; ctx->hscr_bytes = 0;
  memset(&ctx->op, 0, sizeof(ctx->op));
  memset(&ctx->tk, 0, sizeof(ctx->tk));

#define OPEN_HIP(call)                                                                       \
  do { hipError_t e_ = (call);                                                               \
       if (e_ != hipSuccess)                                                                 \
         { dx_fail(NULL, DX_E_HIP, "dx_open: %s: %s", #call, hipGetErrorString(e_));         \
           delete ctx;                                                                       \
           return DX_E_HIP;                                                                  \
         }                                                                                   \
     } while (0)

  OPEN_HIP(hipSetDevice(device));
  open_mark("set device");
  hipDeviceProp_t prop;
  OPEN_HIP(hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  open_mark("device properties");
  OPEN_HIP(hipStreamCreateWithFlags(&ctx->own, hipStreamNonBlocking));
  OPEN_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  open_mark("streams");
  for (int k = 0; k < 19; k++)
    OPEN_HIP(hipEventCreateWithFlags(&ctx->ev[k], hipEventDisableTiming));
  open_mark("events");
  ctx->stream = ctx->own;
  OPEN_HIP(hipMalloc((void **) &ctx->d_tok, DX_TOK_WORDS * sizeof(uint32_t)));
  OPEN_HIP(hipMalloc((void **) &ctx->d_dec, 6 * DX_DEC_SIZE * sizeof(uint16_t)));
  OPEN_HIP(hipMalloc((void **) &ctx->d_long, 6 * (1 + DX_LONG_MAX) * sizeof(uint32_t)));
  OPEN_HIP(hipMalloc((void **) &ctx->d_status, 64));
  OPEN_HIP(hipMalloc((void **) &ctx->d_u64, 64 * sizeof(uint64_t)));
  OPEN_HIP(hipMemset(ctx->d_status, 0, 64));
  open_mark("allocations");
#undef OPEN_HIP
  *out = ctx;
  return DX_OK;
}

extern "C" void dx_close(dx_ctx *ctx)
{ if (ctx == NULL)
    return;
  (void) hipSetDevice(ctx->device);
  (void) hipStreamSynchronize(ctx->side);
  (void) hipStreamSynchronize(ctx->stream);
  for (auto &p : ctx->pend)
    { (void) hipEventDestroy(p.a);
      (void) hipEventDestroy(p.b);
    }
  (void) hipFree(ctx->d_tok);
  (void) hipFree(ctx->d_dec);
  (void) hipFree(ctx->d_long);
  (void) hipFree(ctx->d_status);
  (void) hipFree(ctx->d_u64);
  (void) hipFree(ctx->d_scratch);
  (void) hipFree(ctx->d_scan);
  (void) hipFree(ctx->d_hscr);
  dx_sx_drop_external(ctx);
  (void) hipFree(ctx->sx.idx); (void) hipFree(ctx->sx.off); (void) hipFree(ctx->sx.room); (void) hipFree(ctx->sx.none);
  if (ctx->h_stage[0]) (void) hipHostFree(ctx->h_stage[0]);
  if (ctx->h_up) (void) hipHostFree(ctx->h_up);
  if (ctx->h_down) (void) hipHostFree(ctx->h_down);
  if (ctx->h_pin) (void) hipHostFree(ctx->h_pin);
  (void) hipFree(ctx->tk.del); (void) hipFree(ctx->tk.sub); (void) hipFree(ctx->tk.off); (void) hipFree(ctx->tk.info); (void) hipFree(ctx->tk.count);
  (void) hipFree(ctx->tk.eh);
  (void) hipFree(ctx->qs.perm); (void) hipFree(ctx->qs.list); (void) hipFree(ctx->qs.off2); (void) hipFree(ctx->qs.len2);
  (void) hipFree(ctx->qs.order); (void) hipFree(ctx->qs.rmax); (void) hipFree(ctx->qs.aux);
  (void) hipStreamDestroy(ctx->own);
  (void) hipStreamDestroy(ctx->side);
  for (int k = 0; k < 19; k++) (void) hipEventDestroy(ctx->ev[k]);
  delete ctx;
}

extern "C" int dx_mem_info(dx_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{ if (ctx == NULL) return DX_E_ARG;
  size_t f = 0, t = 0;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  DX_HIP(ctx, hipMemGetInfo(&f, &t));
  if (free_bytes)  *free_bytes  = f;
  if (total_bytes) *total_bytes = t;
  return DX_OK;
}

// Gives back what the context keeps between calls to save allocations: the scratch regions of the one-pass encoder,
// the token slots of the histogram pass, the group index.  (The code tables and everything a later call needs to be
// correct stay; the next call that wants one of the buffers allocates it again.)
extern "C" int dx_trim(dx_ctx *ctx, int what)
{ if (ctx == NULL) return DX_E_ARG;
  if (ctx->op.pending)
    return dx_fail(ctx, DX_E_ARG, "dx_trim: an encode has begun in this context: end it first (dx_qv_encode_onepass_end)");
  DX_HIP(ctx, hipSetDevice(ctx->device));
  DX_HIP(ctx, hipStreamSynchronize(ctx->side));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (what & DX_TRIM_SCRATCH)
    { (void) hipFree(ctx->d_scratch); ctx->d_scratch = NULL; ctx->scratch_bytes = 0;
      (void) hipFree(ctx->d_hscr);    ctx->d_hscr = NULL;    ctx->hscr_bytes = 0;
    }
  if (what & DX_TRIM_TOKENS)
    { (void) hipFree(ctx->tk.del); (void) hipFree(ctx->tk.sub); (void) hipFree(ctx->tk.off); (void) hipFree(ctx->tk.info);
      (void) hipFree(ctx->tk.count); (void) hipFree(ctx->tk.eh);
      memset(&ctx->tk, 0, sizeof(ctx->tk));
      (void) hipFree(ctx->qs.perm);                      /* the short-entry survey's dealing of the batch (k_qs_survey) */
      (void) hipFree(ctx->qs.list); (void) hipFree(ctx->qs.off2); (void) hipFree(ctx->qs.len2);
      (void) hipFree(ctx->qs.order); (void) hipFree(ctx->qs.rmax); (void) hipFree(ctx->qs.aux);
      memset(&ctx->qs, 0, sizeof(ctx->qs));
    }
  if (what & DX_TRIM_INDEX)
    { const int want = ctx->sx.want;
      dx_sx_drop_external(ctx);
      (void) hipFree(ctx->sx.idx); (void) hipFree(ctx->sx.off); (void) hipFree(ctx->sx.room); (void) hipFree(ctx->sx.none);
      memset(&ctx->sx, 0, sizeof(ctx->sx));
      ctx->sx.want = want;
    }
  return DX_OK;
}

extern "C" int dx_set_stream(dx_ctx *ctx, void *hip_stream)
{ if (ctx == NULL) return DX_E_ARG;
  ctx->stream = (hipStream_t) hip_stream;
  return DX_OK;
}

extern "C" int dx_reset_stream(dx_ctx *ctx)
{ if (ctx == NULL) return DX_E_ARG;
  ctx->stream = ctx->own;
  return DX_OK;
}

// An encode that has begun (dx_qv_encode_onepass_begin) still has its last compaction on the side stream: a call that
// reads what it writes makes the context's stream wait for it first (the contract of dexgpu.h).
int dx_after_pending(dx_ctx *ctx)
{ if (ctx->op.pending && !ctx->op.direct)
    DX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[16], 0));
  return DX_OK;
}

extern "C" int dx_sync(dx_ctx *ctx)
{ if (ctx == NULL) return DX_E_ARG;
  int e = dx_after_pending(ctx);
  if (e) return e;
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DX_OK;
}

extern "C" int dx_malloc(dx_ctx *ctx, size_t bytes, void **d_ptr)
{ if (ctx == NULL || d_ptr == NULL) return DX_E_ARG;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
  if (e != hipSuccess)
    return dx_fail(ctx, DX_E_NOMEM, "dx_malloc(%zu): %s", bytes, hipGetErrorString(e));
  return DX_OK;
}

extern "C" int dx_free(dx_ctx *ctx, void *d_ptr)
{ if (ctx == NULL) return DX_E_ARG;
  DX_HIP(ctx, hipFree(d_ptr));
  return DX_OK;
}

// ---- dx_h2d of a large image: several threads, each through two pinned buffers of its own ------------------------------
// A copy from pageable memory (a mapped file: what the tools hand over) goes through the runtime's own staging on ONE thread: 9 GB/s
// on this host, a fifth of what the link carries -- and the upload is most of what a tool spends on a large file.  From
// DX_UP_MIN bytes on the image is dealt in chunks of DX_UP_CHUNK to DX_UP_THREADS threads; each copies its chunks into one of its
// two pinned buffers (touching the file's pages on the way: they fault in side by side) and sends the buffer off on a stream of
// its own while it fills the other.  DEXGPU_PLAIN_H2D=1: the one-call copy.
#define DX_UP_MIN     ((size_t) 64 << 20)
#define DX_UP_CHUNK   ((size_t) 8 << 20)
#define DX_UP_THREADS 6
struct up_job { dx_ctx *ctx; uint8_t *dst; const uint8_t *src; size_t bytes; int id, nth; uint8_t *buf[2]; hipError_t err;
                int fd; uint64_t foff; int ioerr; };    // fd >= 0: the bytes come from the file at foff (pread), not from src

static void *up_main(void *arg)
{ up_job *j = (up_job *) arg;
  hipStream_t st = NULL;
  hipEvent_t  ev[2] = { NULL, NULL };
  j->err = hipSetDevice(j->ctx->device);
  if (j->err == hipSuccess) j->err = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (j->err == hipSuccess) j->err = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
  if (j->err == hipSuccess) j->err = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
  size_t k = 0;
  for (size_t at = (size_t) j->id * DX_UP_CHUNK; j->err == hipSuccess && at < j->bytes; at += (size_t) j->nth * DX_UP_CHUNK, k++)
    { const size_t len = j->bytes - at < DX_UP_CHUNK ? j->bytes - at : DX_UP_CHUNK;
      const int    s = (int) (k & 1);
      if (k >= 2) j->err = hipEventSynchronize(ev[s]);     // (the copy that last left from this buffer)
      if (j->err != hipSuccess) break;
      if (j->fd < 0) memcpy(j->buf[s], j->src + at, len);
      else
        { size_t got = 0;
          while (got < len)
            { const ssize_t k2 = pread(j->fd, j->buf[s] + got, len - got, (off_t) (j->foff + at + got));
              if (k2 <= 0) { j->ioerr = 1; break; }
              got += (size_t) k2;
            }
          if (j->ioerr) break;
        }
      j->err = hipMemcpyAsync(j->dst + at, j->buf[s], len, hipMemcpyHostToDevice, st);
      if (j->err == hipSuccess) j->err = hipEventRecord(ev[s], st);
    }
  if (st != NULL) { const hipError_t e = hipStreamSynchronize(st); if (j->err == hipSuccess) j->err = e; }
  for (int s = 0; s < 2; s++) if (ev[s]) (void) hipEventDestroy(ev[s]);
  if (st) (void) hipStreamDestroy(st);
  return NULL;
}

static int h2d_from(dx_ctx *ctx, void *d_dst, const void *src, int fd, uint64_t foff, size_t bytes);

extern "C" int dx_h2d(dx_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{ if (ctx == NULL) return DX_E_ARG;
  if (bytes == 0) return DX_OK;
  return h2d_from(ctx, d_dst, src, -1, 0, bytes);
}

// bytes [foff, foff + bytes) of the file behind fd to the device: read (pread) straight into the pinned buffers by the copying
// threads -- no mapping of the file, whose pages a copy from a mapping has to fault in one by one and the process to give back
// one by one when it unmaps or ends (a 20 GB input: 0.5 s to bring in, 1 s to unmap, of a 2.5 s dexqv)
extern "C" int dx_h2d_fd(dx_ctx *ctx, void *d_dst, int fd, uint64_t foff, size_t bytes)
{ if (ctx == NULL || fd < 0) return DX_E_ARG;
  if (bytes == 0) return DX_OK;
  return h2d_from(ctx, d_dst, NULL, fd, foff, bytes);
}

static int h2d_from(dx_ctx *ctx, void *d_dst, const void *src, int fd, uint64_t foff, size_t bytes)
{ if ((fd >= 0 || bytes >= DX_UP_MIN) && (fd >= 0 || !dx_test_on("plain_h2d")))
    { DX_HIP(ctx, hipSetDevice(ctx->device));
      if (ctx->h_up == NULL && hipHostMalloc((void **) &ctx->h_up, 2 * DX_UP_THREADS * DX_UP_CHUNK, hipHostMallocDefault) != hipSuccess)
        { (void) hipGetLastError(); ctx->h_up = NULL; }
      if (ctx->h_up != NULL)
        { DX_HIP(ctx, hipStreamSynchronize(ctx->stream));         // (what the stream still does with the destination comes first)
          up_job    job[DX_UP_THREADS];
          pthread_t th[DX_UP_THREADS];
          int       made[DX_UP_THREADS];
          for (int t = 0; t < DX_UP_THREADS; t++)
            { job[t].ctx = ctx; job[t].dst = (uint8_t *) d_dst; job[t].src = (const uint8_t *) src; job[t].bytes = bytes;
              job[t].id = t; job[t].nth = DX_UP_THREADS; job[t].err = hipSuccess; job[t].fd = fd; job[t].foff = foff; job[t].ioerr = 0;
              job[t].buf[0] = ctx->h_up + (size_t) (2 * t) * DX_UP_CHUNK; job[t].buf[1] = job[t].buf[0] + DX_UP_CHUNK;
              made[t] = pthread_create(&th[t], NULL, up_main, &job[t]) == 0;
              if (!made[t]) (void) up_main(&job[t]);                // (no thread to be had: in line)
            }
          hipError_t bad = hipSuccess;
          int ioerr = 0;
          for (int t = 0; t < DX_UP_THREADS; t++)
            { if (made[t]) pthread_join(th[t], NULL);
              if (job[t].err != hipSuccess) bad = job[t].err;
              ioerr |= job[t].ioerr;
            }
          if (ioerr) return dx_fail(ctx, DX_E_IO, "dx_h2d_fd: the file could not be read");
          if (bad != hipSuccess)
            { (void) hipGetLastError();
              return dx_fail(ctx, DX_E_HIP, "dx_h2d: copy failed (%s)", hipGetErrorString(bad));
            }
          return DX_OK;
        }
    }
  if (fd >= 0) return dx_fail(ctx, DX_E_NOMEM, "dx_h2d_fd: no pinned staging memory");
  DX_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DX_OK;
}

extern "C" int dx_d2h(dx_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{ if (ctx == NULL) return DX_E_ARG;
  if (bytes == 0) return DX_OK;
  int e = dx_after_pending(ctx);
  if (e) return e;
  DX_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DX_OK;
}

// ---- dx_d2h_stream: chunks through two pinned buffers, a helper thread feeding the sink ----------
#define DX_STAGE_BYTES ((size_t) 32 << 20)

struct stream_job
{ pthread_mutex_t mx;
  pthread_cond_t  cv;
  uint8_t        *buf[2];
  size_t          len[2], at[2];
  int             full[2];        // slot holds a chunk the sink has not taken yet
  int             done, failed;   // producer finished; the sink said stop
  dx_sink_fn      sink;
  void           *user;
};

static void *stream_consumer(void *arg)
{ stream_job *j = (stream_job *) arg;
  for (int slot = 0; ; slot ^= 1)
    { pthread_mutex_lock(&j->mx);
      while (!j->full[slot] && !j->done) pthread_cond_wait(&j->cv, &j->mx);
      const bool have = j->full[slot] != 0;
      pthread_mutex_unlock(&j->mx);
      if (!have) break;                                    // done, and nothing left in this slot: chunks come in slot order
      const int bad = j->failed ? 0 : j->sink(j->user, j->buf[slot], j->len[slot], j->at[slot]);
      pthread_mutex_lock(&j->mx);
      if (bad) j->failed = 1;
      j->full[slot] = 0;
      pthread_cond_broadcast(&j->cv);
      pthread_mutex_unlock(&j->mx);
    }
  return NULL;
}

// ---- ... and for a sink that takes chunks from several threads, in any order (dx_set_sink_threads: a pwrite per chunk) ----------
// One thread copying into a file's pages (tmpfs: 8 GB/s) is slower than the link brings the text (20 GB of text: 2.6 of a 4.1 s
// undexqv).  DX_MT_BUFS pinned buffers of DX_MT_CHUNK; the caller's thread issues the copies, one after the other, never waiting
// for one to land; T helper threads take the chunks in the order they were issued, wait for the chunk's event and hand it to the sink.
#define DX_MT_CHUNK ((size_t) 16 << 20)
#define DX_MT_BUFS  8
#define DX_MT_MIN   ((size_t) 128 << 20)
struct mt_job
{ pthread_mutex_t mx;
  pthread_cond_t  cv;
  uint8_t        *buf[DX_MT_BUFS];
  hipEvent_t      ev[DX_MT_BUFS];
  size_t          len[DX_MT_BUFS], at[DX_MT_BUFS];
  int             busy[DX_MT_BUFS];   // issued and not yet given back by the sink
  size_t          issued, taken;
  int             done, failed, device;
  dx_sink_fn      sink;
  void           *user;
};

static void *mt_consumer(void *arg)
{ mt_job *j = (mt_job *) arg;
  (void) hipSetDevice(j->device);
  for (;;)
    { pthread_mutex_lock(&j->mx);
      while (j->taken >= j->issued && !j->done) pthread_cond_wait(&j->cv, &j->mx);
      if (j->taken >= j->issued) { pthread_mutex_unlock(&j->mx); break; }
      const size_t c = j->taken++;
      const int failed = j->failed;
      pthread_mutex_unlock(&j->mx);
      const int slot = (int) (c % DX_MT_BUFS);
      int bad = hipEventSynchronize(j->ev[slot]) != hipSuccess;
      if (!bad && !failed) bad = j->sink(j->user, j->buf[slot], j->len[slot], j->at[slot]) != 0;
      pthread_mutex_lock(&j->mx);
      if (bad) j->failed = 1;
      j->busy[slot] = 0;
      pthread_cond_broadcast(&j->cv);
      pthread_mutex_unlock(&j->mx);
    }
  return NULL;
}

static int d2h_stream_mt(dx_ctx *ctx, const void *d_src, size_t bytes, dx_sink_fn sink, void *user, int T)
{ if (ctx->h_down == NULL && hipHostMalloc((void **) &ctx->h_down, DX_MT_BUFS * DX_MT_CHUNK, hipHostMallocDefault) != hipSuccess)
    { (void) hipGetLastError(); ctx->h_down = NULL; return 1; }          // (no pinned memory to be had: the caller goes the old way)
  mt_job j;
  pthread_mutex_init(&j.mx, NULL);
  pthread_cond_init(&j.cv, NULL);
  j.issued = j.taken = 0; j.done = 0; j.failed = 0; j.sink = sink; j.user = user; j.device = ctx->device;
  int nev = 0;
  for (int b = 0; b < DX_MT_BUFS; b++)
    { j.buf[b] = ctx->h_down + (size_t) b * DX_MT_CHUNK; j.busy[b] = 0; j.ev[b] = NULL;
      if (hipEventCreateWithFlags(&j.ev[b], hipEventDisableTiming) == hipSuccess) nev++;
    }
  pthread_t th[16];
  int made = 0;
  if (T > 16) T = 16;
  if (nev == DX_MT_BUFS)
    for (int t = 0; t < T; t++)
      if (pthread_create(&th[made], NULL, mt_consumer, &j) == 0) made++;
  int rc = DX_OK;
  if (made == 0) rc = 1;
  size_t c = 0;
  for (size_t at = 0; at < bytes && rc == DX_OK; at += DX_MT_CHUNK, c++)
    { const size_t len = bytes - at < DX_MT_CHUNK ? bytes - at : DX_MT_CHUNK;
      const int    slot = (int) (c % DX_MT_BUFS);
      pthread_mutex_lock(&j.mx);
      while (j.busy[slot] && !j.failed) pthread_cond_wait(&j.cv, &j.mx);
      const int failed = j.failed;
      pthread_mutex_unlock(&j.mx);
      if (failed) break;
      if (hipMemcpyAsync(j.buf[slot], (const uint8_t *) d_src + at, len, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipEventRecord(j.ev[slot], ctx->stream) != hipSuccess)
        { rc = dx_fail(ctx, DX_E_HIP, "dx_d2h_stream: copy failed (%s)", hipGetErrorString(hipGetLastError()));
          break;
        }
      pthread_mutex_lock(&j.mx);
      j.len[slot] = len; j.at[slot] = at; j.busy[slot] = 1; j.issued += 1;
      pthread_cond_broadcast(&j.cv);
      pthread_mutex_unlock(&j.mx);
    }
  pthread_mutex_lock(&j.mx);
  j.done = 1;
  pthread_cond_broadcast(&j.cv);
  pthread_mutex_unlock(&j.mx);
  for (int t = 0; t < made; t++) pthread_join(th[t], NULL);
  (void) hipStreamSynchronize(ctx->stream);
  for (int b = 0; b < DX_MT_BUFS; b++) if (j.ev[b]) (void) hipEventDestroy(j.ev[b]);
  pthread_cond_destroy(&j.cv);
  pthread_mutex_destroy(&j.mx);
  if (rc == DX_OK && j.failed)
    rc = dx_fail(ctx, DX_E_IO, "dx_d2h_stream: the sink refused data");
  return rc;
}

extern "C" int dx_set_sink_threads(dx_ctx *ctx, int threads)
{ if (ctx == NULL) return 1;
  const int was = ctx->sink_threads > 0 ? ctx->sink_threads : 1;
  ctx->sink_threads = threads > 0 ? threads : 1;
  return was;
}

extern "C" int dx_d2h_stream(dx_ctx *ctx, const void *d_src, size_t bytes, dx_sink_fn sink, void *user)
{ if (ctx == NULL || sink == NULL || (bytes && d_src == NULL)) return DX_E_ARG;
  if (bytes == 0) return DX_OK;
  DX_HIP(ctx, hipSetDevice(ctx->device));
  { int e = dx_after_pending(ctx);
    if (e) return e;
  }
  if (ctx->sink_threads > 1 && bytes >= DX_MT_MIN && !dx_test_on("plain_d2h"))
    { const int rc = d2h_stream_mt(ctx, d_src, bytes, sink, user, ctx->sink_threads);
      if (rc <= 0) return rc;                              // (1: no pinned memory or no thread -- nothing has been passed on: the one-thread way)
    }
  if (ctx->h_stage[0] == NULL)
    { void *h = NULL;
      if (hipHostMalloc(&h, 2 * DX_STAGE_BYTES, hipHostMallocDefault) != hipSuccess)
        return dx_fail(ctx, DX_E_NOMEM, "dx_d2h_stream: no pinned staging memory");
      ctx->h_stage[0] = (uint8_t *) h;
      ctx->h_stage[1] = (uint8_t *) h + DX_STAGE_BYTES;
    }
  stream_job j;
  pthread_mutex_init(&j.mx, NULL);
  pthread_cond_init(&j.cv, NULL);
  j.buf[0] = ctx->h_stage[0]; j.buf[1] = ctx->h_stage[1];
  j.full[0] = j.full[1] = 0; j.done = 0; j.failed = 0; j.sink = sink; j.user = user;
  pthread_t th;
  const bool threaded = pthread_create(&th, NULL, stream_consumer, &j) == 0;
  int    rc = DX_OK, slot = 0;
  for (size_t at = 0; at < bytes && rc == DX_OK; at += DX_STAGE_BYTES, slot ^= 1)
    { const size_t len = bytes - at < DX_STAGE_BYTES ? bytes - at : DX_STAGE_BYTES;
      pthread_mutex_lock(&j.mx);
      while (j.full[slot]) pthread_cond_wait(&j.cv, &j.mx);          // the sink is done with this buffer
      const int failed = j.failed;
      pthread_mutex_unlock(&j.mx);
      if (failed) break;
      if (hipMemcpyAsync(j.buf[slot], (const uint8_t *) d_src + at, len, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        { rc = dx_fail(ctx, DX_E_HIP, "dx_d2h_stream: copy failed (%s)", hipGetErrorString(hipGetLastError()));
          break;
        }
      if (!threaded)                                       // no helper thread to be had: in line
        { if (sink(user, j.buf[slot], len, at)) j.failed = 1;
          continue;
        }
      pthread_mutex_lock(&j.mx);
      j.len[slot] = len; j.at[slot] = at; j.full[slot] = 1;
      pthread_cond_broadcast(&j.cv);
      pthread_mutex_unlock(&j.mx);
    }
  pthread_mutex_lock(&j.mx);
  j.done = 1;
  pthread_cond_broadcast(&j.cv);
  pthread_mutex_unlock(&j.mx);
  if (threaded) pthread_join(th, NULL);
  pthread_cond_destroy(&j.cv);
  pthread_mutex_destroy(&j.mx);
  if (rc == DX_OK && j.failed)
    rc = dx_fail(ctx, DX_E_IO, "dx_d2h_stream: the sink refused data");
  return rc;
}

extern "C" int dx_memset(dx_ctx *ctx, void *d_dst, int value, size_t bytes)
{ if (ctx == NULL) return DX_E_ARG;
  if (bytes == 0) return DX_OK;
  DX_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
  return DX_OK;
}

// the scratch budget in force: DEXGPU_SCRATCH_BUDGET, else dx_set_scratch_budget, else 0 (none)
uint64_t dx_budget(const dx_ctx *ctx)
{ const char *e = getenv("DEXGPU_SCRATCH_BUDGET");
  if (e != NULL && *e) return strtoull(e, NULL, 10);
  return ctx->scratch_budget;
}

extern "C" int dx_set_scratch_budget(dx_ctx *ctx, uint64_t bytes)
{ if (ctx == NULL) return DX_E_ARG;
  ctx->scratch_budget = bytes;
  ctx->onepass_min_groups = 0;                             // (what an earlier budget forced no longer binds)
  return DX_OK;
}

int dx_scratch(dx_ctx *ctx, size_t bytes, void **p)
{ if (ctx->op.pending && !ctx->op.direct)                  // an encode has begun: its last compaction reads this buffer
    DX_HIP(ctx, hipStreamSynchronize(ctx->side));
  const uint64_t budget = dx_budget(ctx);
  if (budget && bytes > budget && bytes > ((size_t) 64 << 20))          // (small bookkeeping requests always pass)
    return dx_fail(ctx, DX_E_NOMEM, "scratch request of %zu bytes exceeds the budget of %llu", bytes, (unsigned long long) budget);
  if (bytes > ctx->scratch_bytes)
    { DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_scratch)
        DX_HIP(ctx, hipFree(ctx->d_scratch));
      ctx->d_scratch = NULL;
      ctx->scratch_bytes = 0;
      ctx->scratch_gen += 1;
      size_t want = bytes + (bytes < ((size_t) 1 << 30) ? bytes / 4 : 0) + 4096;     // (head room only while it is cheap)
      if (budget && want > budget && bytes <= budget) want = bytes;
      hipError_t e = hipMalloc(&ctx->d_scratch, want);
      if (e != hipSuccess)
        return dx_fail(ctx, DX_E_NOMEM, "scratch allocation of %zu bytes failed: %s", want,
                       hipGetErrorString(e));
      ctx->scratch_bytes = want;
    }
  *p = ctx->d_scratch;
  return DX_OK;
}

// Number of workgroups for a one-wave-per-unit kernel: enough waves to fill the
// chip (waves_per_cu resident waves on each of the CUs) but never more than the units.
int dx_grid_waves(dx_ctx *ctx, uint64_t n_units, int waves_per_cu)
{ uint64_t waves = (uint64_t) ctx->num_cu * (uint64_t) waves_per_cu;
  if (waves > n_units) waves = n_units;
  uint64_t blocks = (waves + DX_WAVES_PER_BLK - 1) / DX_WAVES_PER_BLK;
  return blocks ? (int) blocks : 1;
}

// ---- per-kernel timing --------------------------------------------------------------------

static const char *k_names[DX_K_COUNT] =
  { "k_pack2_encode", "k_pack2_decode", "k_qv_prescan", "k_qv_hist", "k_qv_sizes", "k_scan",
    "k_qv_encode", "k_qv_decode", "k_synth", "k_index", "k_qv_compact", "k_qv_encode_text",
    "k_qv_decode_sub", "k_qv_decode_runs", "k_qv_decode_plain", "k_qv_decode_tags", "k_qv_walk" };

extern "C" const char *dx_kernel_name(int kernel)
{ return (kernel >= 0 && kernel < DX_K_COUNT) ? k_names[kernel] : "?"; }

void dx_prof_begin_on(dx_ctx *ctx, int kernel, hipStream_t stream)
{ if (!ctx->profiling) return;
  dx_pending p;
  p.kernel = kernel;
  if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess)
    return;
  (void) hipEventRecord(p.a, stream);
  ctx->pend.push_back(p);
}

void dx_prof_end_on(dx_ctx *ctx, hipStream_t stream)
{ if (!ctx->profiling || ctx->pend.empty()) return;
  (void) hipEventRecord(ctx->pend.back().b, stream);
}

static int prof_collect(dx_ctx *ctx)
{ if (ctx->pend.empty()) return DX_OK;
  DX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &p : ctx->pend)
    { float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
        { ctx->ms[p.kernel] += ms;
          ctx->launches[p.kernel] += 1;
        }
      (void) hipEventDestroy(p.a);
      (void) hipEventDestroy(p.b);
    }
  ctx->pend.clear();
  return DX_OK;
}

extern "C" int dx_profile(dx_ctx *ctx, int enable)
{ if (ctx == NULL) return DX_E_ARG;
  int r = prof_collect(ctx);
  if (r) return r;
  if (enable)
    { memset(ctx->ms, 0, sizeof(ctx->ms));
      memset(ctx->launches, 0, sizeof(ctx->launches));
    }
  ctx->profiling = enable != 0;
  return DX_OK;
}

extern "C" int dx_profile_get(dx_ctx *ctx, int kernel, double *ms_total, uint64_t *launches)
{ if (ctx == NULL || kernel < 0 || kernel >= DX_K_COUNT) return DX_E_ARG;
  int r = prof_collect(ctx);
  if (r) return r;
  if (ms_total) *ms_total = ctx->ms[kernel];
  if (launches) *launches = ctx->launches[kernel];
  return DX_OK;
}
