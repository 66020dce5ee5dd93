/* tests/rccl_stub.c -- a HOST stand-in for librccl.so (test infrastructure).
 *
 * Exports the eight symbols sm_gather.hip binds, with RCCL's group semantics restated on host memory: between
 * ncclGroupStart and ncclGroupEnd every ncclSend / ncclRecv / ncclBroadcast is only RECORDED; ncclGroupEnd pairs
 * each receive (rank d from peer s) with the oldest unmatched send (rank s to peer d) of the same size and copies,
 * and fails -- as RCCL would hang -- if anything stays unpaired or sizes differ.  Outside a group an operation is a
 * group of its own (a lone send can never complete: error).  A communicator is a record {rank, n}.  The call
 * sequence is kept as text (stub_log) so that a test can check the ORDER the library issues its calls in.
 * The symbol sm_rccl_host_stand_in tells libstereo_hip.so that buffers are host memory (include/stereo_hip.h).   */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int rank, n; } comm_t;
typedef struct { int kind, rank, peer; const void *src; void *dst; size_t bytes; int done; } op_t;   /* kind: 0 send, 1 recv, 2 bcast */

int sm_rccl_host_stand_in = 1;
static op_t ops[4096];
static int n_ops, depth, poisoned;
static char log_text[1 << 16];

static void logf_(const char *fmt, int a, int b, size_t c)
{
    const size_t l = strlen(log_text);
    if (l + 96 < sizeof log_text) snprintf(log_text + l, sizeof log_text - l, fmt, a, b, c);
}

const char *stub_log(void) { return log_text; }
void stub_log_reset(void) { log_text[0] = 0; }
void stub_poison_next_group(void) { poisoned = 1; }          /* the next ncclGroupEnd fails (error paths) */

static int run_group(void)
{
    int rc = 0;
    for (int i = 0; i < n_ops; i++) {
        if (ops[i].kind != 1) continue;
        int m = -1;
        for (int j = 0; j < n_ops && m < 0; j++)
            if (ops[j].kind == 0 && !ops[j].done && ops[j].rank == ops[i].peer && ops[j].peer == ops[i].rank) m = j;
        if (m < 0 || ops[m].bytes != ops[i].bytes) { rc = 3; continue; }
        memcpy(ops[i].dst, ops[m].src, ops[i].bytes);
        ops[m].done = ops[i].done = 1;
    }
    for (int i = 0; i < n_ops; i++) {
        if (ops[i].kind == 2) { if (ops[i].dst != ops[i].src) memcpy(ops[i].dst, ops[i].src, ops[i].bytes); ops[i].done = 1; }
        if (!ops[i].done) rc = 3;             /* a send nobody receives, a receive nobody feeds: RCCL would hang */
    }
    n_ops = 0;
    if (poisoned) { poisoned = 0; rc = 3; }
    return rc;
}

int ncclCommInitAll(void **comms, int n, const int *devs)
{
    (void)devs;
    for (int r = 0; r < n; r++) {
        comm_t *c = malloc(sizeof *c);
        c->rank = r; c->n = n;
        comms[r] = c;
    }
    logf_("init(%d) ", n, 0, 0);
    return 0;
}
int ncclCommDestroy(void *c) { free(c); return 0; }
int ncclGroupStart(void) { depth++; logf_("start ", 0, 0, 0); return 0; }
int ncclGroupEnd(void)
{
    logf_("end ", 0, 0, 0);
    if (depth <= 0) return 3;
    return --depth == 0 ? run_group() : 0;
}
static int add(int kind, int rank, int peer, const void *src, void *dst, size_t bytes)
{
    if (n_ops >= 4096) return 3;
    ops[n_ops++] = (op_t){kind, rank, peer, src, dst, bytes, 0};
    return depth ? 0 : run_group();
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream)
{
    (void)stream;
    const comm_t *c = comm;
    if (dtype != 1 || peer < 0 || peer >= c->n) return 4;        /* ncclUint8 = 1 */
    logf_("send(%d>%d,%zu) ", c->rank, peer, count);
    return add(0, c->rank, peer, buf, NULL, count);
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *stream)
{
    (void)stream;
    const comm_t *c = comm;
    if (dtype != 1 || peer < 0 || peer >= c->n) return 4;
    logf_("recv(%d<%d,%zu) ", c->rank, peer, count);
    return add(1, c->rank, peer, NULL, buf, count);
}
int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, void *stream)
{
    (void)stream;
    const comm_t *c = comm;
    if (dtype != 1 || root < 0 || root >= c->n) return 4;
    logf_("bcast(%d<%d,%zu) ", c->rank, root, count);
    return add(2, c->rank, root, send, recv, count);
}
const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 4 ? "invalid argument (stub)" : "unpaired or failed group (stub)"; }
