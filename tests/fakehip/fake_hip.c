/*
 * fake_hip.c -- TEST INFRASTRUCTURE: a do-nothing HIP runtime for running the HOST half of
 * liblut_ldpc_amd.so under AddressSanitizer / UBSan on a machine without a GPU
 * (tests/test_host_dryrun_asan.py, tools/asan_host.sh).
 *
 * "Device" memory is host malloc (so ASan sees every hipMemcpy/hipMemset that runs past an allocation,
 * every double free and every use after hipFree), kernel launches are recorded and skipped.  The shim checks
 * what a launch can be checked for without running it: non-zero grid/block, block <= 1024 threads, grid
 * within the HIP limits, no launch with a NULL function.  Nothing in lut_ldpc_amd/ knows about this file:
 * the ASan build of the library is linked against it instead of libamdhip64 (hiprtc stays the real one).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
typedef struct { uint32_t x, y, z; } dim3_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };

static hipError_t g_last = 0;
static long g_launches = 0, g_allocs = 0, g_frees = 0, g_live_bytes = 0, g_graph_launches = 0, g_captures = 0;
static __thread int g_capturing = 0;            /* a capture belongs to the stream of one host thread (hipStreamCaptureModeThreadLocal) */
/* several host threads drive "devices" at once in the multi-device ber_sim (ber_sim_multi.cpp): the shim's tables are shared */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
#define LOCK() pthread_mutex_lock(&g_mu)
#define UNLOCK() pthread_mutex_unlock(&g_mu)
static int fake_device_count(void) { const char *e = getenv("FAKEHIP_DEVICES"); int n = e ? atoi(e) : 1; return n < 1 ? 1 : n > 16 ? 16 : n; }

#define MAX_FUNCS 4096
static struct { const void *host; char name[200]; } g_funcs[MAX_FUNCS];
static int g_nfuncs = 0;
static long g_by_func[MAX_FUNCS];

/* live allocations: so that hipFree of an unknown pointer and leaks at exit are reported */
#define MAX_LIVE 65536
static struct { void *p; size_t n; } g_live[MAX_LIVE];
static int g_nlive = 0;

static hipError_t fail(const char *what)
{
    fprintf(stderr, "fake_hip: %s\n", what);
    g_last = hipErrorInvalidValue;
    abort();            /* under ASan this prints a stack */
    return hipErrorInvalidValue;
}

/* ---- introspection for the tests */
long fakehip_launches(void) { return g_launches; }
long fakehip_graph_launches(void) { return g_graph_launches; }
long fakehip_captures(void) { return g_captures; }
long fakehip_live_allocations(void) { return g_nlive; }
long fakehip_live_bytes(void) { return g_live_bytes; }
long fakehip_launches_of(const char *substr)
{
    long n = 0;
    for (int i = 0; i < g_nfuncs; i++) if (strstr(g_funcs[i].name, substr)) n += g_by_func[i];
    return n;
}
void fakehip_reset_counters(void) { g_launches = 0; g_graph_launches = 0; g_captures = 0; memset(g_by_func, 0, sizeof g_by_func); }

/* ---- device / stream */
hipError_t hipGetDeviceCount(int *n) { *n = fake_device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d) { return (d >= 0 && d < fake_device_count()) ? hipSuccess : fail("hipSetDevice: no such device"); }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = 0; return e; }
const char *hipGetErrorString(hipError_t e) { return e ? "fake_hip error" : "no error"; }
hipError_t hipStreamCreateWithFlags(void **s, unsigned flags) { (void)flags; *s = malloc(16); return hipSuccess; }
hipError_t hipStreamDestroy(void *s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(void *s) { (void)s; if (g_capturing) return fail("hipStreamSynchronize during capture"); return hipSuccess; }

/* ---- memory */
hipError_t hipMalloc(void **p, size_t n)
{
    if (g_capturing) return fail("hipMalloc during stream capture");
    if (n == 0) { *p = NULL; return hipSuccess; }
    *p = malloc(n);
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0xA5, n);                 /* device memory is NOT zeroed by hipMalloc: poison it */
    LOCK();
    if (g_nlive >= MAX_LIVE) { UNLOCK(); return fail("fake_hip: live table full"); }
    g_live[g_nlive].p = *p; g_live[g_nlive].n = n; g_nlive++;
    g_allocs++; g_live_bytes += (long)n;
    UNLOCK();
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    if (g_capturing) return fail("hipFree during stream capture");
    LOCK();
    for (int i = 0; i < g_nlive; i++)
        if (g_live[i].p == p) {
            g_live_bytes -= (long)g_live[i].n;
            g_live[i] = g_live[--g_nlive];
            g_frees++;
            UNLOCK();
            free(p);
            return hipSuccess;
        }
    UNLOCK();
    return fail("hipFree of a pointer hipMalloc never returned (or double free)");
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, int kind)
{
    (void)kind;
    if (g_capturing) return fail("synchronous hipMemcpy during stream capture");
    if (n) memmove(dst, src, n);        /* instrumented: both ranges are checked by ASan */
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, int kind, void *stream)
{
    (void)kind; (void)stream;
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, void *stream)
{
    (void)stream;
    if (n) memset(dst, v, n);
    return hipSuccess;
}

/* ---- events */
hipError_t hipEventCreate(void **e) { *e = malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(void *e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(void *e, void *s) { (void)e; (void)s; return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, void *a, void *b) { (void)a; (void)b; *ms = 0.001f; return hipSuccess; }
hipError_t hipEventSynchronize(void *e) { (void)e; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)16 << 30; return hipSuccess; }     /* (placement search of large batches) */

/* ---- kernel registration + launches */
void **__hipRegisterFatBinary(const void *data) { (void)data; static void *h; return &h; }
void __hipUnregisterFatBinary(void **m) { (void)m; }
void __hipRegisterFunction(void **m, const void *host, char *dev, const char *name, unsigned limit, void *a, void *b, void *c, void *d, int *w)
{
    (void)m; (void)dev; (void)limit; (void)a; (void)b; (void)c; (void)d; (void)w;
    if (g_nfuncs < MAX_FUNCS) { g_funcs[g_nfuncs].host = host; snprintf(g_funcs[g_nfuncs].name, sizeof g_funcs[0].name, "%s", name); g_nfuncs++; }
}
void __hipRegisterVar(void **m, void *var, char *a, const char *b, int ext, size_t size, int constant, int global)
{ (void)m; (void)var; (void)a; (void)b; (void)ext; (void)size; (void)constant; (void)global; }

static __thread struct { dim3_t g, b; size_t shmem; void *stream; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3_t g, dim3_t b, size_t shmem, void *stream) { g_cfg.g = g; g_cfg.b = b; g_cfg.shmem = shmem; g_cfg.stream = stream; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3_t *g, dim3_t *b, size_t *shmem, void **stream) { *g = g_cfg.g; *b = g_cfg.b; *shmem = g_cfg.shmem; *stream = g_cfg.stream; return hipSuccess; }

static hipError_t check_geometry(uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by, uint32_t bz, size_t shmem)
{
    if (!gx || !gy || !gz) return fail("launch with an empty grid");
    if (!bx || !by || !bz || (uint64_t)bx * by * bz > 1024) return fail("launch with an empty or oversized block");
    if ((uint64_t)gx * bx > 0xFFFFFFFFull || gy > 65535 || gz > 65535) return fail("grid beyond the HIP limits");
    if (shmem > 160 * 1024) return fail("more than 160 KiB of dynamic LDS");
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *func, dim3_t g, dim3_t b, void **args, size_t shmem, void *stream)
{
    (void)stream;
    if (!func || !args) return fail("hipLaunchKernel: NULL function or argument array");
    hipError_t e = check_geometry(g.x, g.y, g.z, b.x, b.y, b.z, shmem);
    if (e) return e;
    int known = 0;
    LOCK();
    for (int i = 0; i < g_nfuncs; i++) if (g_funcs[i].host == func) { g_by_func[i]++; known = 1; if (getenv("FAKEHIP_TRACE")) fprintf(stderr, "launch %s\n", g_funcs[i].name); break; }
    g_launches += known;
    UNLOCK();
    if (!known) return fail("hipLaunchKernel: function was never registered");
    return hipSuccess;
}
hipError_t hipFuncGetAttributes(void *attr, const void *func)
{
    (void)attr;
    for (int i = 0; i < g_nfuncs; i++) if (g_funcs[i].host == func) return hipSuccess;
    return fail("hipFuncGetAttributes: function was never registered");
}
hipError_t hipFuncSetAttribute(const void *func, int attr, int value)
{
    (void)attr;
    if (value < 0 || value > 160 * 1024) return fail("hipFuncSetAttribute: dynamic LDS beyond 160 KB");
    for (int i = 0; i < g_nfuncs; i++) if (g_funcs[i].host == func) return hipSuccess;
    return fail("hipFuncSetAttribute: function was never registered");
}
hipError_t hipDeviceSynchronize(void) { if (g_capturing) return fail("hipDeviceSynchronize during capture"); return hipSuccess; }
hipError_t hipModuleLoadData(void **mod, const void *image) { if (!image) return fail("hipModuleLoadData(NULL)"); *mod = malloc(8); return hipSuccess; }
hipError_t hipModuleUnload(void *mod) { free(mod); return hipSuccess; }
hipError_t hipModuleGetFunction(void **f, void *mod, const char *name) { (void)mod; (void)name; static int fn; *f = &fn; return hipSuccess; }
hipError_t hipModuleLaunchKernel(void *f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shmem, void *stream, void **params, void **extra)
{
    (void)stream; (void)extra;
    if (!f || !params) return fail("hipModuleLaunchKernel: NULL function or parameters");
    hipError_t e = check_geometry(gx, gy, gz, bx, by, bz, shmem);
    if (e) return e;
    LOCK(); g_launches++; UNLOCK();
    return hipSuccess;
}

/* ---- graphs */
hipError_t hipStreamBeginCapture(void *s, int mode) { (void)s; (void)mode; if (g_capturing) return fail("nested capture"); g_capturing = 1; LOCK(); g_captures++; UNLOCK(); return hipSuccess; }
hipError_t hipStreamEndCapture(void *s, void **graph) { (void)s; if (!g_capturing) return fail("EndCapture without BeginCapture"); g_capturing = 0; *graph = malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(void **exec, void *graph, void *a, void *b, size_t c) { (void)a; (void)b; (void)c; if (!graph) return fail("instantiate NULL graph"); *exec = malloc(8); return hipSuccess; }
hipError_t hipGraphDestroy(void *g) { free(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(void *e) { free(e); return hipSuccess; }
hipError_t hipGraphLaunch(void *e, void *s) { (void)s; if (!e) return fail("hipGraphLaunch(NULL)"); LOCK(); g_graph_launches++; UNLOCK(); return hipSuccess; }
