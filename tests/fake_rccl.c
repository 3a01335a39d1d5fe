/* Test infrastructure, not part of the product: a stand-in for the six RCCL entry points the in-library GPU group
 * uses (pt_group_create / pt_group_render in cuda-pathtracer_amd/csrc/pt_capi.hip), so that the group's host threads,
 * frame split, accumulator clears and grouped reduce can run on a machine with one GPU. Selected through the
 * environment variable PTAMD_RCCL_LIBRARY. The reduce goes through host memory: at ncclGroupEnd every rank's stream is
 * synchronised, the send buffers are summed in rank order in fp32, and the sum is copied to the root's receive buffer. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

typedef struct fake_world { int n; int refs; } fake_world;
typedef struct fake_comm { int rank, n, device; fake_world* world; } fake_comm;
typedef fake_comm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;

typedef struct pending { const void* send; void* recv; size_t count; int root; fake_comm* comm; hipStream_t stream; } pending;
static pending g_pending[64];
static int g_n_pending = 0, g_group_depth = 0;
static int g_calls_reduce = 0, g_calls_group = 0;

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devices)
{
    if (!comms || n < 1 || n > 64 || !devices) return ncclInvalidArgument;
    fake_world* w = (fake_world*)calloc(1, sizeof(fake_world));
    if (!w) return ncclSystemError;
    w->n = n; w->refs = n;
    for (int i = 0; i < n; i++) {
        fake_comm* c = (fake_comm*)calloc(1, sizeof(fake_comm));
        if (!c) return ncclSystemError;
        c->rank = i; c->n = n; c->device = devices[i]; c->world = w;
        comms[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    if (--c->world->refs == 0) free(c->world);
    free(c);
    return ncclSuccess;
}

static ncclResult_t run_pending(void)
{
    if (g_n_pending == 0) return ncclSuccess;
    const int n = g_pending[0].comm->n;
    const size_t count = g_pending[0].count;
    const int root = g_pending[0].root;
    if (g_n_pending != n) return ncclInvalidUsage;           /* every rank of the communicator must take part */
    pending* by_rank[64] = {0};
    for (int i = 0; i < n; i++) {
        pending* p = &g_pending[i];
        if (p->comm->world != g_pending[0].comm->world || p->count != count || p->root != root || by_rank[p->comm->rank]) return ncclInvalidUsage;
        by_rank[p->comm->rank] = p;
    }
    float* sum = (float*)malloc(count * sizeof(float));
    float* part = (float*)malloc(count * sizeof(float));
    if (!sum || !part) return ncclSystemError;
    for (int r = 0; r < n; r++) {
        pending* p = by_rank[r];
        if (hipSetDevice(p->comm->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(p->stream) != hipSuccess) return ncclUnhandledCudaError;      /* the rank's render is queued on this stream */
        if (hipMemcpy(r == 0 ? sum : part, p->send, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (r > 0) for (size_t k = 0; k < count; k++) sum[k] = sum[k] + part[k];
    }
    pending* pr = by_rank[root];
    if (hipSetDevice(pr->comm->device) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(pr->recv, sum, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    free(sum); free(part);
    g_n_pending = 0;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { g_group_depth++; g_calls_group++; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    return run_pending();
}

ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream)
{
    if (!send || !comm || type != ncclFloat || op != ncclSum || root < 0 || root >= comm->n || g_n_pending >= 64) return ncclInvalidArgument;
    if (comm->rank == root && !recv) return ncclInvalidArgument;
    g_calls_reduce++;
    pending p = {send, recv, count, root, comm, stream};
    g_pending[g_n_pending++] = p;
    if (g_group_depth == 0) return comm->n == 1 ? run_pending() : ncclInvalidUsage;       /* one thread drives several ranks: only inside a group */
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (stand-in)";
        case ncclInvalidArgument: return "invalid argument (stand-in)";
        case ncclInvalidUsage: return "invalid usage (stand-in)";
        default: return "error (stand-in)";
    }
}

/* how often the library came through here, for the test */
void fake_rccl_calls(int* reduces, int* groups) { if (reduces) *reduces = g_calls_reduce; if (groups) *groups = g_calls_group; }
