/*
 * dx_env.h -- the library's switches.
 *
 * What a user of the tools may set stands in the environment under a name of its own (README.md): DEXGPU_DEVICE, DEXGPU_DEVICES,
 * DEXGPU_TIMING, DEXGPU_TEARDOWN, DEXGPU_TEXT_BUDGET, DEXGPU_SCRATCH_BUDGET, DEXGPU_WALK_THREADS.  Everything else -- a route
 * forced for a test, a threshold lowered so that a small file takes the large files' path, a failure injected -- is a key of ONE
 * variable, DEXGPU_TEST: keys and key=value pairs separated by commas or blanks, e.g.
 *     DEXGPU_TEST="no_tokens,onepass_groups=3,device_walk_min=1"
 * looked up where the decision is made (never inside a kernel's launch loop); tests/ and tools/ set nothing else.
 */
#ifndef DX_ENV_H
#define DX_ENV_H
#ifdef __cplusplus
extern "C" {
#endif
/* the value of `key` in DEXGPU_TEST ("" for a key that stands alone), NULL when it is not there; the pointer is the calling thread's until its next call */
const char *dx_test_str(const char *key);
/* is the key there, with a value that does not begin with '0'? */
int         dx_test_on(const char *key);
/* the key's value as a number (strtoll, base 0), `dflt` when the key is not there or has no value */
long long   dx_test_num(const char *key, long long dflt);
#ifdef __cplusplus
}
#endif
#endif
