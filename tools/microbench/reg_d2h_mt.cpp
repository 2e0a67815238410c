// tools/microbench/reg_d2h_mt.cpp -- reg_d2h with the registering spread over T threads: each takes slices of S MiB of the shared
// mapping of the (sparse or laid-out) tmpfs file in turn -- hipHostRegister, copy on a stream of its own, hipHostUnregister.
// build: hipcc -O2 -o tools/microbench/reg_d2h_mt tools/microbench/reg_d2h_mt.cpp -lpthread ; ./reg_d2h_mt [GB] [threads] [slice MiB] [fallocate 0|1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static size_t N, SL; static int T; static char *M; static char *D; static volatile int bad = 0;
static size_t next_ = 0; static pthread_mutex_t mx = PTHREAD_MUTEX_INITIALIZER;
static double t_reg[64], t_cp[64], t_un[64];
static void *worker(void *arg)
{ const int id = (int) (size_t) arg;
  hipStream_t st;
  hipSetDevice(0);
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { bad = 1; return NULL; }
  for (;;)
    { pthread_mutex_lock(&mx); const size_t at = next_; next_ += SL; pthread_mutex_unlock(&mx);
      if (at >= N) break;
      const size_t len = N - at < SL ? N - at : SL;
      double a = now();
      if (hipHostRegister(M + at, len, hipHostRegisterDefault) != hipSuccess) { bad = 2; break; }
      double b = now();
      if (hipMemcpyAsync(M + at, D + at, len, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { bad = 3; break; }
      double c = now();
      if (hipHostUnregister(M + at) != hipSuccess) { bad = 4; break; }
      double d = now();
      t_reg[id] += b - a; t_cp[id] += c - b; t_un[id] += d - c;
    }
  hipStreamDestroy(st);
  return NULL;
}
int main(int argc, char **argv)
{ const size_t G = argc > 1 ? (size_t) atoi(argv[1]) : 8;
  T = argc > 2 ? atoi(argv[2]) : 8; SL = (size_t) (argc > 3 ? atoi(argv[3]) : 256) << 20;
  const int fa = argc > 4 ? atoi(argv[4]) : 0;
  N = G << 30;
  if (hipMalloc((void **) &D, N) != hipSuccess || hipMemset(D, 0x5a, N) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("no device memory\n"); return 1; }
  char path[] = "/dev/shm/reg_d2h.XXXXXX";
  int fd = mkstemp(path);
  unlink(path);
  double t0 = now();
  if (fa) { if (posix_fallocate(fd, 0, (off_t) N) != 0) { perror("fallocate"); return 1; } }
  else if (ftruncate(fd, (off_t) N) != 0) { perror("ftruncate"); return 1; }
  double t1 = now();
  M = (char *) mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (M == MAP_FAILED) { perror("mmap"); return 1; }
  pthread_t th[64];
  for (int i = 0; i < T; i++) pthread_create(&th[i], NULL, worker, (void *) (size_t) i);
  for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
  double t2 = now();
  double r = 0, c = 0, u = 0; for (int i = 0; i < T; i++) { r += t_reg[i]; c += t_cp[i]; u += t_un[i]; }
  printf("%zu GB, %d threads, slices of %zu MiB, %s: layout %.3f s, register+copy+unregister %.3f s (%.1f GB/s; a thread's sums: register %.3f copy %.3f unregister %.3f), bad %d, bytes %02x..%02x\n",
         G, T, SL >> 20, fa ? "laid out first" : "sparse", t1 - t0, t2 - t1, N / 1e9 / (t2 - t1), r / T, c / T, u / T, bad,
         (unsigned char) M[0], (unsigned char) M[N - 1]);
  double t3 = now();
  munmap(M, N);
  printf("  munmap %.3f s\n", now() - t3);
  close(fd);
  return 0;
}
