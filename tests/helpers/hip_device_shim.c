/* LD_PRELOAD shim for tests/test_gpu_dist.py::test_multi_context_call_selects_every_device: pretends the
 * box has MPSFR_SHIM_DEVICES GPUs (all of them the one real device 0) and logs, per host thread, the
 * device id every hipSetDevice asked for -- so that the per-device worker threads of
 * mpsfr_reconstruct_multi can be checked on a one-GPU box.  gcc -shared -fPIC -o shim.so shim.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

typedef int (*fn_set)(int);
typedef int (*fn_count)(int*);
typedef int (*fn_attr)(int*, int, int);

static int ndev(void) {
    const char* e = getenv("MPSFR_SHIM_DEVICES");
    return e ? atoi(e) : 4;
}

int hipGetDeviceCount(int* n) {
    fn_count real = (fn_count)dlsym(RTLD_NEXT, "hipGetDeviceCount");
    int rc = real(n);
    if (rc == 0 && *n >= 1) *n = ndev();
    return rc;
}

int hipSetDevice(int id) {
    fn_set real = (fn_set)dlsym(RTLD_NEXT, "hipSetDevice");
    const char* log = getenv("MPSFR_SHIM_LOG");
    if (log) {
        FILE* f = fopen(log, "a");
        if (f) {
            fprintf(f, "%lu %d\n", (unsigned long)pthread_self(), id);
            fclose(f);
        }
    }
    return real(0);
}

int hipDeviceGetAttribute(int* v, int attr, int id) {
    fn_attr real = (fn_attr)dlsym(RTLD_NEXT, "hipDeviceGetAttribute");
    (void)id;
    return real(v, attr, 0);
}
