// comm.hip -- RCCL below the C ABI: halo exchange of a lattice slab with its neighbouring ranks and the all-reduce of the
// observables, without PyTorch.  (The reference has no communication of any kind: SURVEY.md section 2; this serves
// BASELINE configs[3], the 16384^2 lattice cut into row slabs over the GPUs of one node.)
//
// RCCL is loaded at run time (dlopen: librccl.so.1, then librccl.so), so libtsu_hip.so carries no link-time dependency on
// it and a process that already has an RCCL loaded (PyTorch ships its own) shares that one.  The communicator's unique id is
// created on one rank (tsu_comm_unique_id) and handed to the others by the caller through whatever channel it has (a file, an
// environment variable, MPI, torch.distributed ...): 128 opaque bytes.
//
// Exchange of one refresh period: rank r sends its first `ghost` owned rows to the rank above and its last `ghost` owned rows to
// the rank below, and receives its ghost rows from them -- four point-to-point operations in ONE RCCL group on the context's
// stream (ordered with the sweep launches on that stream).  Order inside the group: send up, send down, receive from below,
// receive from above, so that with two ranks (up == down) the first send pairs with the peer's first receive.
#include <dlfcn.h>

#include <chrono>
#include <thread>
#include <rccl/rccl.h>

#include "ising2d.h"

struct tsu_comm {
    tsu_ctx* ctx;
    ncclComm_t comm;
    int rank, nranks;
    int64_t* d_red;  // device scratch of the all-reduce
    hipEvent_t done;  // recorded behind the last group issued (tsu_comm_wait)
    uint64_t n_exchanges;
};

// wait for everything issued on the context's stream so far, but never for ever: an RCCL group whose peer never arrives would hold
// hipStreamSynchronize until somebody kills the process.  Event query with a deadline; on expiry the caller gets TSU_E_RCCL and is
// expected to leave (the communicator is not usable afterwards).
static int tsu_comm_bounded_wait(tsu_comm* c, double timeout_s, const char* what) {
    tsu_ctx* ctx = c->ctx;
    TSU_HIP_TRY(ctx, hipEventRecord(c->done, ctx->stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(c->done);
        if (q == hipSuccess) return TSU_OK;
        if (q != hipErrorNotReady) return tsu_fail(ctx, TSU_E_HIP, "%s: %s", what, hipGetErrorString(q));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > timeout_s)
            return tsu_fail(ctx, TSU_E_RCCL, "%s: rank %d of %d waited %.0f s for its peers (exchange %llu): giving up", what, c->rank, c->nranks,
                            timeout_s, (unsigned long long)c->n_exchanges);
        if (spins > 1000) std::this_thread::sleep_for(std::chrono::microseconds(spins > 100000 ? 1000 : 20));
    }
}

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";
};

Rccl& rccl() {
    static Rccl R;
    static bool tried = false;
    if (tried) return R;
    tried = true;
    const char* names[] = {getenv("TSU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (R.handle) break;
    }
    if (!R.handle) {
        snprintf(R.why, sizeof R.why, "cannot load RCCL (%s)", dlerror());
        return R;
    }
#define TSU_SYM(field, name)                                                        \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, name));           \
    if (!R.field) {                                                                 \
        snprintf(R.why, sizeof R.why, "RCCL library has no symbol %s", name);       \
        return R;                                                                   \
    }
    TSU_SYM(GetUniqueId, "ncclGetUniqueId")
    TSU_SYM(CommInitRank, "ncclCommInitRank")
    TSU_SYM(CommDestroy, "ncclCommDestroy")
    TSU_SYM(GroupStart, "ncclGroupStart")
    TSU_SYM(GroupEnd, "ncclGroupEnd")
    TSU_SYM(Send, "ncclSend")
    TSU_SYM(Recv, "ncclRecv")
    TSU_SYM(AllReduce, "ncclAllReduce")
    TSU_SYM(GetErrorString, "ncclGetErrorString")
#undef TSU_SYM
    R.ok = true;
    return R;
}

#define TSU_RCCL_TRY(ctx, expr)                                                                                   \
    do {                                                                                                          \
        ncclResult_t r__ = (expr);                                                                                \
        if (r__ != ncclSuccess) return tsu_fail((ctx), TSU_E_RCCL, "%s: %s (%s:%d)", #expr, rccl().GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)
}  // namespace

extern "C" {

int tsu_comm_unique_id(uint8_t id[128]) {
    if (!id) return TSU_E_INVALID;
    Rccl& R = rccl();
    if (!R.ok) return tsu_fail(nullptr, TSU_E_RCCL, "tsu_comm_unique_id: %s", R.why);
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    ncclUniqueId u;
    ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) return tsu_fail(nullptr, TSU_E_RCCL, "ncclGetUniqueId: %s", R.GetErrorString(r));
    memcpy(id, &u, 128);
    return TSU_OK;
}

int tsu_comm_create(tsu_ctx* ctx, int nranks, int rank, const uint8_t id[128], tsu_comm** out) {
    TSU_ENTER(ctx);
    if (!ctx) return TSU_E_INVALID;
    TSU_REQUIRE(ctx, out && id && nranks >= 1 && rank >= 0 && rank < nranks, "tsu_comm_create: bad arguments");
    *out = nullptr;
    Rccl& R = rccl();
    if (!R.ok) return tsu_fail(ctx, TSU_E_RCCL, "tsu_comm_create: %s", R.why);
    tsu_comm* c = new (std::nothrow) tsu_comm();
    if (!c) return tsu_fail(ctx, TSU_E_NOMEM, "tsu_comm_create: host allocation failed");
    c->ctx = ctx;
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = R.CommInitRank(&c->comm, nranks, u, rank);  // collective over the ranks: every rank calls it
    if (r != ncclSuccess) {
        delete c;
        return tsu_fail(ctx, TSU_E_RCCL, "ncclCommInitRank(%d of %d): %s", rank, nranks, R.GetErrorString(r));
    }
    c->n_exchanges = 0;
    if (hipMalloc(&c->d_red, 8 * sizeof(int64_t)) != hipSuccess || hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        (void)R.CommDestroy(c->comm);
        delete c;
        return tsu_fail(ctx, TSU_E_NOMEM, "tsu_comm_create: hipMalloc failed");
    }
    *out = c;
    return TSU_OK;
}

int tsu_comm_destroy(tsu_comm* c) {
    TSU_ENTER(c ? c->ctx : nullptr);
    if (!c) return TSU_OK;
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)rccl().CommDestroy(c->comm);
    (void)hipFree(c->d_red);
    (void)hipEventDestroy(c->done);
    delete c;
    return TSU_OK;
}

int tsu_comm_wait(tsu_comm* c, double timeout_s, uint64_t* n_exchanges) {
    TSU_ENTER(c ? c->ctx : nullptr);
    if (!c) return TSU_E_INVALID;
    TSU_REQUIRE(c->ctx, timeout_s > 0.0, "tsu_comm_wait: timeout must be positive");
    if (n_exchanges) *n_exchanges = c->n_exchanges;
    return tsu_comm_bounded_wait(c, timeout_s, "tsu_comm_wait");
}

int tsu_ising2d_halo_exchange(tsu_ising2d* L, tsu_comm* c) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L || !c) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, c->ctx == ctx, "ising2d_halo_exchange: the communicator belongs to another context");
    TSU_REQUIRE(ctx, L->ghost > 0 && L->ghost <= L->rows, "ising2d_halo_exchange: the lattice is not a slab with ghost rows");
    TSU_REQUIRE(ctx, (long long)L->rows * c->nranks == L->total_rows && L->row0 == (long long)L->rows * c->rank,
                "ising2d_halo_exchange: slab %d of %d does not match rows [%lld, %lld) of %lld", c->rank, c->nranks, (long long)L->row0,
                (long long)(L->row0 + L->rows), (long long)L->total_rows);
    Rccl& R = rccl();
    const int G = L->ghost, up = c->rank > 0 ? c->rank - 1 : (L->periodic ? c->nranks - 1 : -1),
              down = c->rank < c->nranks - 1 ? c->rank + 1 : (L->periodic ? 0 : -1);
    int8_t* base = L->alloc[L->cur];  // buffer rows: [ghost][owned][ghost]
    const size_t bytes = (size_t)G * L->pitch;
    int8_t* gtop = base;
    int8_t* top = base + (size_t)G * L->pitch;
    int8_t* bot = base + (size_t)L->rows * L->pitch;           // last G owned rows start at buffer row rows
    int8_t* gbot = base + (size_t)(G + L->rows) * L->pitch;
    TSU_RCCL_TRY(ctx, R.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (up >= 0 && r == ncclSuccess) r = R.Send(top, bytes, ncclInt8, up, c->comm, ctx->stream);
    if (down >= 0 && r == ncclSuccess) r = R.Send(bot, bytes, ncclInt8, down, c->comm, ctx->stream);
    if (down >= 0 && r == ncclSuccess) r = R.Recv(gbot, bytes, ncclInt8, down, c->comm, ctx->stream);
    if (up >= 0 && r == ncclSuccess) r = R.Recv(gtop, bytes, ncclInt8, up, c->comm, ctx->stream);
    const ncclResult_t re = R.GroupEnd();
    if (r != ncclSuccess) return tsu_fail(ctx, TSU_E_RCCL, "ising2d_halo_exchange: %s", R.GetErrorString(r));
    TSU_RCCL_TRY(ctx, re);
    c->n_exchanges += 1;
    return TSU_OK;
}

int tsu_comm_allreduce_i64(tsu_comm* c, int64_t* values, int n) {
    TSU_ENTER(c ? c->ctx : nullptr);
    if (!c) return TSU_E_INVALID;
    tsu_ctx* ctx = c->ctx;
    TSU_REQUIRE(ctx, values && n >= 1 && n <= 8, "tsu_comm_allreduce_i64: 1..8 values");
    TSU_HIP_TRY(ctx, hipMemcpyAsync(c->d_red, values, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    TSU_RCCL_TRY(ctx, rccl().AllReduce(c->d_red, c->d_red, (size_t)n, ncclInt64, ncclSum, c->comm, ctx->stream));
    TSU_HIP_TRY(ctx, hipMemcpyAsync(values, c->d_red, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    static const double timeout_s = getenv("TSU_COMM_TIMEOUT_S") ? atof(getenv("TSU_COMM_TIMEOUT_S")) : 120.0;
    return tsu_comm_bounded_wait(c, timeout_s > 0 ? timeout_s : 120.0, "tsu_comm_allreduce_i64");
}

}  // extern "C"
