/*
 * fake_rccl.c -- TEST INFRASTRUCTURE: the five RCCL entry points ber_sim_multi.cpp loads with dlopen, for the sanitizer build
 * on a machine without GPUs (LUTLDPC_RCCL_LIB points here).  "Device" buffers are host memory (fake_hip.c), a communicator
 * group is a rendezvous of host threads: every collective blocks until all ranks of the group have called it, then each rank
 * computes its own result from the posted send buffers -- the semantics of ncclAllGather / ncclAllReduce(sum) on int64.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct ihipStream_t *hipStream_t;      /* the pointer type ber_sim_multi.cpp calls through */

typedef struct Group {
    int n;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int arrived, gen, left;
    const void *send[64];
} Group;
typedef struct Comm { Group *g; int rank; } Comm;

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    (void)devlist;
    if (ndev < 1 || ndev > 64) return 4;   /* ncclInvalidArgument */
    Group *g = calloc(1, sizeof *g);
    g->n = ndev; g->left = ndev;
    pthread_mutex_init(&g->mu, NULL); pthread_cond_init(&g->cv, NULL);
    for (int i = 0; i < ndev; i++) { Comm *c = malloc(sizeof *c); c->g = g; c->rank = i; comms[i] = c; }
    return 0;
}
int ncclCommDestroy(void *comm)
{
    Comm *c = comm;
    Group *g = c->g;
    pthread_mutex_lock(&g->mu);
    const int last = --g->left == 0;
    pthread_mutex_unlock(&g->mu);
    if (last) { pthread_mutex_destroy(&g->mu); pthread_cond_destroy(&g->cv); free(g); }
    free(c);
    return 0;
}
const char *ncclGetErrorString(int rc) { return rc ? "fake_rccl error" : "no error"; }

/* post `send`, wait for everybody, run body (reads all posted buffers), wait again so that nobody re-posts early */
static void rendezvous(Comm *c, const void *send)
{
    Group *g = c->g;
    pthread_mutex_lock(&g->mu);
    g->send[c->rank] = send;
    const int gen = g->gen;
    if (++g->arrived == g->n) { g->arrived = 0; g->gen++; pthread_cond_broadcast(&g->cv); }
    else while (gen == g->gen) pthread_cond_wait(&g->cv, &g->mu);
    pthread_mutex_unlock(&g->mu);
}
int ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t stream)
{
    (void)stream;
    if (dtype != 4) return 4;              /* ncclInt64 only */
    Comm *c = comm;
    rendezvous(c, send);
    int64_t *out = recv;
    for (int r = 0; r < c->g->n; r++) memcpy(out + (size_t)r * sendcount, c->g->send[r], sendcount * sizeof(int64_t));
    rendezvous(c, send);
    return 0;
}
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream)
{
    (void)stream;
    if (dtype != 4 || op != 0) return 4;   /* ncclInt64, ncclSum */
    Comm *c = comm;
    rendezvous(c, send);
    int64_t acc[64];
    if (count > 64) return 4;
    memset(acc, 0, sizeof acc);
    for (int r = 0; r < c->g->n; r++) for (size_t k = 0; k < count; k++) acc[k] += ((const int64_t *)c->g->send[r])[k];
    rendezvous(c, send);                   /* (send == recv is allowed: write only after everybody has read) */
    memcpy(recv, acc, count * sizeof(int64_t));
    return 0;
}
