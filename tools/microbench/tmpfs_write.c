/* tmpfs_write.c -- how to land 1 GiB in a tmpfs file fastest (what the CLI's output stage does).
 * usage: tmpfs_write <path>      prints one line per strategy
 * strategies: mmap + T threads faulting fresh pages; posix_fallocate first, then the same; posix_fallocate +
 * pwrite (1 and T threads); plain fwrite. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
typedef struct { int fd; char *d; const char *s; size_t n; off_t at; } job;
static void *cp(void *a) { job *j = a; memcpy(j->d, j->s, j->n); return 0; }
static void *wr(void *a)
{ job *j = a; size_t d = 0;
  while (d < j->n)
    { ssize_t k = pwrite(j->fd, j->s + d, j->n - d > (8 << 20) ? (8 << 20) : j->n - d, j->at + d);
      if (k <= 0) break;
      d += k;
    }
  return 0;
}
int main(int argc, char **argv)
{ size_t n = (size_t) 1 << 30;
  const char *path = argc > 1 ? argv[1] : "/dev/shm/tmpfs_write.bin";
  char *src = malloc(n);
  int   mode, T, rep;
  memset(src, 7, n);
  for (rep = 0; rep < 2; rep++)
  for (mode = 0; mode < 4; mode++)
    for (T = 1; T <= 16; T *= 4)
      { int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        pthread_t th[16]; job jb[16]; size_t sl = n / T; int k;
        double t0 = now(), t1, t2;
        if (mode & 1) { if (posix_fallocate(fd, 0, n)) return 1; }
        else if (ftruncate(fd, n)) return 1;
        t1 = now();
        if (mode & 2)
          { for (k = 0; k < T; k++) { jb[k].fd = fd; jb[k].s = src + k * sl; jb[k].n = sl; jb[k].at = k * sl; pthread_create(&th[k], 0, wr, &jb[k]); }
            for (k = 0; k < T; k++) pthread_join(th[k], 0);
          }
        else
          { char *m = mmap(0, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            for (k = 0; k < T; k++) { jb[k].d = m + k * sl; jb[k].s = src + k * sl; jb[k].n = sl; pthread_create(&th[k], 0, cp, &jb[k]); }
            for (k = 0; k < T; k++) pthread_join(th[k], 0);
            munmap(m, n);
          }
        t2 = now();
        close(fd);
        printf("%-10s %-7s T=%-2d  alloc %.3f  copy %.3f  total %.3f s\n", (mode & 1) ? "fallocate" : "ftruncate",
               (mode & 2) ? "pwrite" : "mmap", T, t1 - t0, t2 - t1, t2 - t0);
        unlink(path);
      }
  return 0;
}
