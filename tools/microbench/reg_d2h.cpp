// tools/microbench/reg_d2h.cpp -- could a device-to-host copy land in the output file's own pages?  A tmpfs file of G GB is laid out
// (posix_fallocate), mapped shared, the mapping registered with the runtime (hipHostRegister: its pages pinned) and the device
// buffer copied into it; against that: the same bytes through pinned staging buffers and pwrite (what the tools do).
// build: hipcc -O2 -o tools/microbench/reg_d2h tools/microbench/reg_d2h.cpp ; run on the GPU box: ./tools/microbench/reg_d2h [GB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{ const size_t G = argc > 1 ? (size_t) atoi(argv[1]) : 8, n = G << 30;
  void *d = NULL;
  CK(hipMalloc(&d, n));
  CK(hipMemset(d, 0x5a, n));
  CK(hipDeviceSynchronize());
  char path[] = "/dev/shm/reg_d2h.XXXXXX";
  int fd = mkstemp(path);
  if (fd < 0) { perror("mkstemp"); return 1; }
  unlink(path);
  double t0 = now();
  if (posix_fallocate(fd, 0, (off_t) n) != 0) { perror("fallocate"); return 1; }
  double t1 = now();
  void *m = mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (m == MAP_FAILED) { perror("mmap"); return 1; }
  double t2 = now();
  hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
  double t3 = now();
  printf("%zu GB: fallocate %.3f s, mmap %.3f s, hipHostRegister %.3f s (%s)\n", G, t1 - t0, t2 - t1, t3 - t2, hipGetErrorString(e));
  if (e == hipSuccess)
    { CK(hipMemcpy(m, d, n, hipMemcpyDeviceToHost));
      double t4 = now();
      CK(hipHostUnregister(m));
      double t5 = now();
      printf("  copy into the file's pages %.3f s (%.1f GB/s), unregister %.3f s; all of it %.3f s\n", t4 - t3, n / 1e9 / (t4 - t3), t5 - t4, t5 - t0);
      printf("  first bytes %02x %02x, last %02x\n", ((unsigned char *) m)[0], ((unsigned char *) m)[1], ((unsigned char *) m)[n - 1]);
    }
  double t6 = now();
  munmap(m, n);
  printf("  munmap %.3f s\n", now() - t6);
  // without fallocate: register a fresh sparse file's mapping
  { ftruncate(fd, 0); ftruncate(fd, (off_t) n);
    double a0 = now();
    void *m2 = mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    hipError_t e2 = hipHostRegister(m2, n, hipHostRegisterDefault);
    double a1 = now();
    printf("sparse file: mmap + hipHostRegister %.3f s (%s)\n", a1 - a0, hipGetErrorString(e2));
    if (e2 == hipSuccess)
      { CK(hipMemcpy(m2, d, n, hipMemcpyDeviceToHost));
        double a2 = now();
        CK(hipHostUnregister(m2));
        printf("  copy %.3f s (%.1f GB/s); all of it %.3f s\n", a2 - a1, n / 1e9 / (a2 - a1), now() - a0);
      }
    munmap(m2, n);
  }
  close(fd);
  return 0;
}
