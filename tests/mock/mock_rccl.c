/* mock_rccl.c -- a stand-in for librccl that runs the collectives between THREADS of one process on HOST memory.
 *
 * Test infrastructure only (tests/test_host_comm.py): loaded through O2V_RCCL_LIB, it lets the library's RCCL code path -
 * dlopen and symbol lookup, ncclGetUniqueId, ncclCommInitRank from one thread per rank, the five collectives' argument
 * marshalling (element counts, datatypes, reduction operators, the in-place all-gather, the broadcast root), the error
 * string path and ncclCommDestroy - execute for N = 8 on a machine without any GPU.  It implements the subset of the RCCL API
 * obj2voxel_amd/csrc/o2v_comm.cpp binds, with RCCL's enum values (rccl.h: ncclSum 0, ncclMax 2, ncclMin 3; ncclUint8 1,
 * ncclUint32 3, ncclUint64 5).  Pointers are taken as host memory, streams are ignored, every collective is synchronous.
 * A communicator group is keyed by its unique id; ncclCommInitRank blocks until every rank of the group has joined, as
 * RCCL's does. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t; /* 0 = ncclSuccess */
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
typedef void *hipStream_t;

#define MAX_RANKS 64
typedef struct Group {
    char id[128];
    int world, joined, left;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    /* one collective at a time: the ranks deposit their pointers, the last to arrive performs it */
    int arrived, generation;
    const void *send[MAX_RANKS];
    void *recv[MAX_RANKS];
    struct Group *next;
} Group;
typedef struct ncclComm { Group *g; int rank; } *ncclComm_t;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static Group *g_groups = NULL;
static int g_ids = 0;
static int g_calls[8]; /* init, destroy, allreduce, allgather, broadcast, id */

int mock_rccl_calls(int which) { return g_calls[which & 7]; }

const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "mock rccl: invalid argument"; }
ncclResult_t ncclGroupStart(void) { return 0; }
ncclResult_t ncclGroupEnd(void) { return 0; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return 4;
    pthread_mutex_lock(&g_mu);
    memset(id->internal, 0, sizeof(id->internal));
    const int n = ++g_ids;
    memcpy(id->internal, "mock-rccl", 9);
    memcpy(id->internal + 16, &n, sizeof(n));
    g_calls[5]++;
    pthread_mutex_unlock(&g_mu);
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;
    pthread_mutex_lock(&g_mu);
    Group *g = g_groups;
    while (g && memcmp(g->id, id.internal, 128) != 0) g = g->next;
    if (!g) {
        g = (Group *) calloc(1, sizeof(Group));
        memcpy(g->id, id.internal, 128);
        g->world = nranks;
        pthread_mutex_init(&g->mu, NULL);
        pthread_cond_init(&g->cv, NULL);
        g->next = g_groups;
        g_groups = g;
    }
    g_calls[0]++;
    pthread_mutex_unlock(&g_mu);
    if (g->world != nranks) return 4;
    pthread_mutex_lock(&g->mu);
    g->joined++;
    pthread_cond_broadcast(&g->cv);
    while (g->joined < g->world) pthread_cond_wait(&g->cv, &g->mu); /* (RCCL blocks here until every rank has joined) */
    pthread_mutex_unlock(&g->mu);
    *comm = (ncclComm_t) calloc(1, sizeof(**comm));
    (*comm)->g = g;
    (*comm)->rank = rank;
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return 4;
    pthread_mutex_lock(&g_mu);
    g_calls[1]++;
    pthread_mutex_unlock(&g_mu);
    free(comm);
    return 0;
}

static size_t type_size(ncclDataType_t t) { return t == 1 ? 1 : t == 3 ? 4 : t == 5 ? 8 : 0; }

/* every rank calls with its buffers; `op_fn` runs once, on the last rank to arrive, with everybody's pointers */
typedef void (*coll_fn)(Group *g, size_t count, ncclDataType_t type, int op_or_root);
static ncclResult_t rendezvous(ncclComm_t c, const void *send, void *recv, size_t count, ncclDataType_t type, int arg, coll_fn fn)
{
    if (!c || !type_size(type)) return 4;
    Group *g = c->g;
    pthread_mutex_lock(&g->mu);
    const int gen = g->generation;
    g->send[c->rank] = send;
    g->recv[c->rank] = recv;
    if (++g->arrived == g->world) {
        fn(g, count, type, arg);
        g->arrived = 0;
        g->generation++;
        pthread_cond_broadcast(&g->cv);
    }
    else {
        while (g->generation == gen) pthread_cond_wait(&g->cv, &g->mu);
    }
    pthread_mutex_unlock(&g->mu);
    return 0;
}

static void do_allreduce(Group *g, size_t count, ncclDataType_t type, int op)
{
    const size_t sz = type_size(type);
    unsigned char *acc = (unsigned char *) malloc(count * sz);
    memcpy(acc, g->send[0], count * sz);
    for (int r = 1; r < g->world; ++r)
        for (size_t i = 0; i < count; ++i) {
            if (type == 3) {
                uint32_t a, b;
                memcpy(&a, acc + i * 4, 4);
                memcpy(&b, (const unsigned char *) g->send[r] + i * 4, 4);
                a = op == 0 ? a + b : op == 2 ? (a > b ? a : b) : (a < b ? a : b);
                memcpy(acc + i * 4, &a, 4);
            }
            else if (type == 5) {
                uint64_t a, b;
                memcpy(&a, acc + i * 8, 8);
                memcpy(&b, (const unsigned char *) g->send[r] + i * 8, 8);
                a = op == 0 ? a + b : op == 2 ? (a > b ? a : b) : (a < b ? a : b);
                memcpy(acc + i * 8, &a, 8);
            }
        }
    for (int r = 0; r < g->world; ++r) memcpy(g->recv[r], acc, count * sz);
    free(acc);
}
static void do_allgather(Group *g, size_t count, ncclDataType_t type, int unused)
{
    (void) unused;
    const size_t bytes = count * type_size(type);
    unsigned char *all = (unsigned char *) malloc(bytes * (size_t) g->world);
    for (int r = 0; r < g->world; ++r) memcpy(all + (size_t) r * bytes, g->send[r], bytes); /* (gathered first: the calls are in place) */
    for (int r = 0; r < g->world; ++r) memcpy(g->recv[r], all, bytes * (size_t) g->world);
    free(all);
}
static void do_broadcast(Group *g, size_t count, ncclDataType_t type, int root)
{
    const size_t bytes = count * type_size(type);
    unsigned char *src = (unsigned char *) malloc(bytes);
    memcpy(src, g->send[root], bytes);
    for (int r = 0; r < g->world; ++r) memcpy(g->recv[r], src, bytes);
    free(src);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t s)
{
    (void) s;
    if (op != 0 && op != 2 && op != 3) return 4;
    if (type != 3 && type != 5) return 4;
    __sync_fetch_and_add(&g_calls[2], 1);
    return rendezvous(comm, send, recv, count, type, op, do_allreduce);
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t type, ncclComm_t comm, hipStream_t s)
{
    (void) s;
    __sync_fetch_and_add(&g_calls[3], 1);
    return rendezvous(comm, send, recv, sendcount, type, 0, do_allgather);
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t s)
{
    (void) s;
    if (!comm || root < 0 || root >= comm->g->world) return 4;
    __sync_fetch_and_add(&g_calls[4], 1);
    return rendezvous(comm, send, recv, count, type, root, do_broadcast);
}
