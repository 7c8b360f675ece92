// nz_comm.cpp -- the multi-GPU half of libnoize_hip.so: one process per GPU, a grid cut into row stripes, RCCL
// neighbour halo exchange (ncclSend / ncclRecv with rank +- 1 on a communicator stream of its own), the path's one
// collective (ncclAllGather for GetMapRangeJob), and the sharded launch plan of the stock stage list.  New-framework
// feature: the reference has independent clamped tiles only (Scripts/MeshTileGenerator.cs:166-192), requested
// one by one through BasePipeline.Schedule (Pipeline/Executable/Pipeline.cs:104-128).  See include/noize_hip.h.
//
// RCCL is opened with dlopen when the first communicator entry is called -- a host that never shards never maps it, and a
// process that already holds a copy (PyTorch bundles one under the same SONAME) shares that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the table below

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "nz_internal.hpp"

namespace {

struct rccl_api {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

std::mutex g_rccl_mx;
rccl_api g_rccl;

int32_t rccl_load(const rccl_api **out) {
    std::lock_guard<std::mutex> lk(g_rccl_mx);
    if (!g_rccl.so) {
        const char *env = getenv("NZ_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        void *so = nullptr;
        char tried[768] = "";
        for (const char *n : names) {
            if (!n || !*n) continue;
            so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
            const char *why = dlerror();  // every candidate's reason, not only the last one's
            size_t used = strlen(tried);
            snprintf(tried + used, sizeof tried - used, "%s%s: %s", used ? "; " : "", n, why ? why : "?");
        }
        if (!so) {
            nz_set_error("RCCL is not available (NZ_RCCL_LIB names another path): %s", tried);
            return NZ_ERR_COMM;
        }
        rccl_api a;
        a.so = so;
#define NZ_SYM(field, name)                                                  \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(so, name));          \
    if (!a.field) {                                                          \
        nz_set_error("librccl: symbol %s is missing", name);                 \
        dlclose(so);                                                         \
        return NZ_ERR_COMM;                                                  \
    }
        NZ_SYM(GetUniqueId, "ncclGetUniqueId")
        NZ_SYM(CommInitRank, "ncclCommInitRank")
        NZ_SYM(CommDestroy, "ncclCommDestroy")
        NZ_SYM(GroupStart, "ncclGroupStart")
        NZ_SYM(GroupEnd, "ncclGroupEnd")
        NZ_SYM(Send, "ncclSend")
        NZ_SYM(Recv, "ncclRecv")
        NZ_SYM(AllGather, "ncclAllGather")
        NZ_SYM(GetErrorString, "ncclGetErrorString")
        NZ_SYM(GetVersion, "ncclGetVersion")
#undef NZ_SYM
        g_rccl = a;
    }
    *out = &g_rccl;
    return NZ_OK;
}

#define NZ_NCCL(api, expr)                                                                                   \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) {                                                                             \
            nz_set_error("%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__);     \
            return NZ_ERR_COMM;                                                                              \
        }                                                                                                    \
    } while (0)

#define NZ_TRY(expr)          \
    do {                      \
        int32_t rc_ = (expr); \
        if (rc_) return rc_;  \
    } while (0)

// `floats` contiguous floats from `send` on rank `src` to `recv` on rank `dst`; a rank fills in the end(s) it holds.
// Both ends of every pair of ranks walk their lists in the same order (RCCL matches the k-th send to a peer with that
// peer's k-th receive from us), and a transfer between two stripes of ONE rank posts its send and its receive back to back.
struct xfer {
    const float *send;
    float *recv;
    size_t floats;
    int src, dst;
};

}  // namespace

struct nz_comm {
    nz_ctx *ctx = nullptr;
    const rccl_api *api = nullptr;
    int device = 0, rank = 0, world = 1;
    int users = 0;  // nz_sharded objects that hold this communicator
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;                 // the exchanges' own stream
    hipEvent_t ev_go = nullptr, ev_done = nullptr;  // ctx stream -> comm stream, comm stream -> ctx stream
    bool pending = false;                         // a batch is in flight that ctx's stream has not waited for
};

namespace {

// Posts one batch behind everything enqueued on ctx's stream so far.  comm == NULL (one rank, no RCCL): device copies on
// the context's own stream.
// on_ctx_stream: the batch is enqueued on ctx's own stream (no event hand-off, nothing to finish): the launch that follows
// simply follows.
int32_t post_batch(nz_ctx *ctx, nz_comm *comm, const xfer *x, size_t n, bool on_ctx_stream = false) {
    if (n == 0) return NZ_OK;
    if (!comm) {
        for (size_t i = 0; i < n; i++) {
            NZ_REQUIRE(x[i].src == 0 && x[i].dst == 0, "a transfer between ranks needs a communicator");
            NZ_HIP(hipMemcpyAsync(x[i].recv, x[i].send, x[i].floats * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        }
        return NZ_OK;
    }
    NZ_REQUIRE(!comm->pending, "nz_halo_exchange_begin: the previous exchange has not been finished");
    const rccl_api *api = comm->api;
    hipStream_t xs = on_ctx_stream ? ctx->stream : comm->stream;
    if (!on_ctx_stream) {
        NZ_HIP(hipEventRecord(comm->ev_go, ctx->stream));
        NZ_HIP(hipStreamWaitEvent(comm->stream, comm->ev_go, 0));
    }
    NZ_NCCL(api, api->GroupStart());
    for (size_t i = 0; i < n; i++) {
        if (x[i].src == comm->rank) {
            ncclResult_t r = api->Send(x[i].send, x[i].floats, ncclFloat, x[i].dst, comm->comm, xs);
            if (r != ncclSuccess) {
                (void)api->GroupEnd();
                nz_set_error("ncclSend to rank %d failed: %s", x[i].dst, api->GetErrorString(r));
                return NZ_ERR_COMM;
            }
        }
        if (x[i].dst == comm->rank) {
            ncclResult_t r = api->Recv(x[i].recv, x[i].floats, ncclFloat, x[i].src, comm->comm, xs);
            if (r != ncclSuccess) {
                (void)api->GroupEnd();
                nz_set_error("ncclRecv from rank %d failed: %s", x[i].src, api->GetErrorString(r));
                return NZ_ERR_COMM;
            }
        }
    }
    NZ_NCCL(api, api->GroupEnd());
    if (on_ctx_stream) return NZ_OK;
    NZ_HIP(hipEventRecord(comm->ev_done, comm->stream));
    comm->pending = true;
    return NZ_OK;
}

int32_t finish_batch(nz_ctx *ctx, nz_comm *comm) {
    if (!comm || !comm->pending) return NZ_OK;
    NZ_HIP(hipStreamWaitEvent(ctx->stream, comm->ev_done, 0));
    comm->pending = false;
    return NZ_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nz_comm_rccl_version(int32_t *version) {
    NZ_REQUIRE(version, "version is NULL");
    const rccl_api *api = nullptr;
    NZ_TRY(rccl_load(&api));
    int v = 0;
    NZ_NCCL(api, api->GetVersion(&v));
    *version = v;
    return NZ_OK;
}

extern "C" int32_t nz_comm_unique_id(uint8_t *id_out) {
    NZ_REQUIRE(id_out, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == NZ_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    const rccl_api *api = nullptr;
    NZ_TRY(rccl_load(&api));
    ncclUniqueId id;
    NZ_NCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return NZ_OK;
}

extern "C" int32_t nz_comm_init(nz_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world, nz_comm **out) {
    NZ_REQUIRE(ctx && id && out, "ctx/id/out is NULL");
    *out = nullptr;
    NZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside [0, %d)", rank, world);
    const rccl_api *api = nullptr;
    NZ_TRY(rccl_load(&api));
    NZ_HIP(hipSetDevice(ctx->device));
    nz_comm *c = new nz_comm();
    c->ctx = ctx;
    c->api = api;
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = api->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        nz_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, api->GetErrorString(r));
        delete c;
        return NZ_ERR_COMM;
    }
    // highest priority: a hardware queue of its own (streams of one priority share a few), and the short transfer
    // kernels are dispatched ahead of the long stencil launches they run beside
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_go, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        nz_set_error("communicator stream / events: %s", hipGetErrorString(e));
        (void)nz_comm_destroy(c);
        return NZ_ERR_HIP;
    }
    *out = c;
    return NZ_OK;
}

extern "C" int32_t nz_comm_destroy(nz_comm *c) {
    if (!c) return NZ_OK;
    // a sharded grid keeps a pointer to its communicator: destroy the grids first
    NZ_REQUIRE(c->users == 0, "nz_comm_destroy: %d nz_sharded object(s) still use this communicator (nz_sharded_destroy them first)",
               c->users);
    (void)hipSetDevice(c->device);
    // with overlap 0 (the default) the transfers and the one collective run on the CONTEXT's stream: both streams drain
    // before ncclCommDestroy
    if (c->ctx && c->ctx->stream) (void)hipStreamSynchronize(c->ctx->stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)c->api->CommDestroy(c->comm);
    if (c->ev_go) (void)hipEventDestroy(c->ev_go);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return NZ_OK;
}

extern "C" int32_t nz_comm_rank(const nz_comm *c) { return c ? c->rank : 0; }
extern "C" int32_t nz_comm_world(const nz_comm *c) { return c ? c->world : 1; }

// ---------------------------------------------------------------------------------------------
// neighbour halo exchange of one stripe per rank
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nz_halo_exchange_begin(nz_ctx *ctx, nz_comm *comm, float *const *planes, int32_t n_planes,
                                          const nz_stripe *st, int32_t up_rows, int32_t down_rows, nz_handle dep) {
    NZ_TRY(nz_ctx_begin(ctx, dep));
    NZ_REQUIRE(comm, "comm is NULL");
    NZ_REQUIRE(comm->ctx == ctx, "the communicator belongs to another context");
    NZ_REQUIRE(planes && n_planes >= 1 && n_planes <= 64, "planes / n_planes invalid");
    NZ_TRY(nz_check_stripe(st, 0));
    NZ_REQUIRE(st->pitch == 0 || st->pitch == st->cols, "halo exchange needs contiguous rows (pitch == cols)");
    NZ_REQUIRE(up_rows >= 0 && down_rows >= 0, "negative row count");
    const int nown = st->own1 - st->own0;
    const bool has_up = comm->rank > 0, has_down = comm->rank + 1 < comm->world;
    NZ_REQUIRE(nown >= up_rows && nown >= down_rows, "stripe thinner than the halo: ghost rows come from the adjacent rank only");
    NZ_REQUIRE(!has_up || st->own0 - up_rows >= 0, "no room for %d ghost rows above", up_rows);
    NZ_REQUIRE(!has_down || st->own1 + down_rows <= st->rows, "no room for %d ghost rows below", down_rows);
    const size_t W = (size_t)st->cols;
    std::vector<xfer> x;
    for (int p = 0; p < n_planes; p++) {
        float *t = planes[p];
        NZ_REQUIRE(t, "planes[%d] is NULL", p);
        if (up_rows > 0) {  // my top ghost rows <- the rows just above, owned by rank - 1
            if (has_down) x.push_back({t + (size_t)(st->own1 - up_rows) * W, nullptr, up_rows * W, comm->rank, comm->rank + 1});
            if (has_up) x.push_back({nullptr, t + (size_t)(st->own0 - up_rows) * W, up_rows * W, comm->rank - 1, comm->rank});
        }
        if (down_rows > 0) {  // my bottom ghost rows <- the rows just below, owned by rank + 1
            if (has_up) x.push_back({t + (size_t)st->own0 * W, nullptr, down_rows * W, comm->rank, comm->rank - 1});
            if (has_down) x.push_back({nullptr, t + (size_t)st->own1 * W, down_rows * W, comm->rank + 1, comm->rank});
        }
    }
    return post_batch(ctx, comm, x.data(), x.size());
}

extern "C" int32_t nz_halo_exchange_finish(nz_ctx *ctx, nz_comm *comm, nz_handle *out) {
    NZ_REQUIRE(ctx && comm, "ctx/comm is NULL");
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_TRY(finish_batch(ctx, comm));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_halo_exchange(nz_ctx *ctx, nz_comm *comm, float *const *planes, int32_t n_planes, const nz_stripe *st,
                                    int32_t up_rows, int32_t down_rows, nz_handle dep, nz_handle *out) {
    NZ_TRY(nz_halo_exchange_begin(ctx, comm, planes, n_planes, st, up_rows, down_rows, dep));
    return nz_halo_exchange_finish(ctx, comm, out);
}

// ---------------------------------------------------------------------------------------------
// GetMapRangeJob over the ranks
// ---------------------------------------------------------------------------------------------
namespace {
// triples: `nloc` local {min, max, range} triples (device); work: 5 * n + 6 floats with n = nloc * world
int32_t gather_and_fold(nz_ctx *ctx, nz_comm *comm, const float *triples, int nloc, float *work, float *res, float lim_min,
                        float lim_max) {
    const int world = comm ? comm->world : 1, n = nloc * world;
    float *gathered = work, *mins = work + 3 * n, *maxs = work + 4 * n, *lo = work + 5 * n, *hi = work + 5 * n + 3;
    if (comm) {
        NZ_NCCL(comm->api, comm->api->AllGather(triples, gathered, (size_t)3 * nloc, ncclFloat, comm->comm, ctx->stream));
    } else {
        NZ_HIP(hipMemcpyAsync(gathered, triples, (size_t)3 * nloc * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    NZ_TRY(nz_launch_range_split(ctx->stream, gathered, n, mins, maxs));
    float *scratch = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, nz_map_range_scratch_floats(), &scratch));
    NZ_TRY(nz_launch_map_range(ctx->stream, mins, (size_t)n, lim_min, -__builtin_inff(), lo, scratch));
    NZ_TRY(nz_launch_map_range(ctx->stream, maxs, (size_t)n, __builtin_inff(), lim_max, hi, scratch));
    return nz_launch_range_compose(ctx->stream, lo, hi, res);
}
}  // namespace

extern "C" int32_t nz_comm_allgather_range(nz_ctx *ctx, nz_comm *comm, const float *map, size_t n_floats, float *res,
                                           float lim_min, float lim_max, nz_handle dep, nz_handle *out) {
    NZ_TRY(nz_ctx_begin(ctx, dep));
    NZ_REQUIRE(map && res, "map/res is NULL");
    NZ_REQUIRE(n_floats >= 1, "empty map");
    NZ_REQUIRE(!(lim_min != lim_min) && !(lim_max != lim_max), "the limits must not be NaN");
    NZ_REQUIRE(!comm || comm->ctx == ctx, "the communicator belongs to another context");
    const int world = comm ? comm->world : 1;
    // scratch: the partials of nz_launch_map_range, then {local triple, gathered, mins, maxs, lo, hi}
    const size_t part = nz_map_range_scratch_floats(), need = part + 3 + 5 * (size_t)world + 6;
    float *scratch = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, need, &scratch));
    float *triple = scratch + part, *work = triple + 3;
    NZ_TRY(nz_launch_map_range(ctx->stream, map, n_floats, __builtin_inff(), -__builtin_inff(), triple, scratch));
    NZ_TRY(gather_and_fold(ctx, comm, triple, 1, work, res, lim_min, lim_max));
    return nz_ctx_finish(ctx, out);
}

// ---------------------------------------------------------------------------------------------
// the sharded stage list
// ---------------------------------------------------------------------------------------------
namespace {
enum { OP_NOISE = 1, OP_XBEGIN = 2, OP_XFINISH = 3, OP_FILTER = 4, OP_FLOW = 5, OP_EROSION = 6, OP_MARK = 7 };

struct sh_op {
    int kind, stripe;  // stripe -1: all
    int n, a, b;
    int own0, own1;
    int src, dst, sin, sout;  // plane ids: 0 = A, 1 = B; state sets 0 / 1
    int batch;                // OP_XBEGIN: index into nz_sharded::batches
};

struct sh_stripe {
    int v;                      // index among all stripes of the grid
    int g0, nown;               // first owned global row, owned rows
    int rows, own0, own1, grow0;
    float *A = nullptr, *B = nullptr;
    float *S[2] = {nullptr, nullptr};  // two sets of {water, fN, fS, fE, fW}, plane-major
};

struct launch_rad { int kind, n, up, down; };  // kind: 1 filter, 2 flow, 3 erosion

struct win { int own0, own1; };
}  // namespace

struct nz_sharded {
    nz_ctx *ctx = nullptr;
    nz_comm *comm = nullptr;
    nz_sharded_desc d{};
    nz_terrain_params p{};
    nz_kernel_taps taps{};
    int rank = 0, world = 1;  // of the GEOMETRY (asRank / asWorld in a rehearsal)
    int S = 1, V = 1, halo = 1;
    int result_plane = 0;
    bool need_state = false;
    bool dry = false;  // plan only (created without a context): no planes, no transfers, cannot run
    std::vector<sh_stripe> st;
    std::vector<sh_op> prog;
    std::vector<std::vector<xfer>> batches;
    size_t bytes_sent = 0;
    float *range_work = nullptr;  // S triples + 5 V + 6
    // timing of the compute stream's waits
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;

    float *plane(int j, int id) const {
        const sh_stripe &s = st[j];
        if (id == 0) return s.A;
        if (id == 1) return s.B;
        const int set = (id - 2) / 5, k = (id - 2) % 5;
        return s.S[set] + (size_t)k * s.rows * d.cols;
    }
    nz_stripe stripe_view(int j, int own0, int own1) const {
        const sh_stripe &s = st[j];
        return nz_stripe{d.cols, s.rows, s.grow0, d.grows, own0, own1, 0};
    }
};

namespace {

std::vector<int> split_iterations(int n, int cap) {
    const int launches = (n + cap - 1) / cap, base = n / launches, rem = n % launches;
    std::vector<int> v;
    for (int i = 0; i < launches; i++) v.push_back(base + (i < rem ? 1 : 0));
    return v;
}

int32_t launch_radii(const nz_terrain_params &p, nz_kernel_taps *taps, std::vector<launch_rad> *out) {
    out->clear();
    NZ_REQUIRE(p.filterIterations >= 0 && p.flowIterations >= 0 && p.erosionIterations >= 0, "negative iteration count");
    if (p.filterIterations > 0) {
        NZ_TRY(nz_filter_taps(p.filter, taps));
        const int cap = (taps->ksize & 1) ? nz_conv_tcap(taps->ksize) : 0;
        if (cap <= 0) {
            nz_set_error("KernelFilterType %d has no fused stripe kernel", p.filter);
            return NZ_ERR_UNSUPPORTED;
        }
        const int O = (taps->ksize - 1) / 2;
        for (int T : split_iterations(p.filterIterations, cap)) out->push_back({1, T, T * O, T * O});
    }
    if (p.flowIterations > 0)
        for (int n : split_iterations(p.flowIterations, nz_flow_fused_max())) out->push_back({2, n, 2 * n, 2 * n});
    for (int left = p.erosionIterations; left > 0;) {
        const int E = std::min(left, nz_erosion_max_fused());
        out->push_back({3, E, E, 0});  // the min window reaches upwards only
        left -= E;
    }
    return NZ_OK;
}

// StripePlan.widened / rows_window (noize_job_amd/sharded.py)
win widened(const nz_sharded &sh, const sh_stripe &s, int up, int down) {
    return win{std::max(s.own0 - up, -s.grow0), std::min(s.own1 + down, sh.d.grows - s.grow0)};
}
win rows_window(win w, int a, int b) { return win{std::max(a, w.own0), std::min(b, w.own1)}; }

// one exchange of `planes` (ids, the same for every stripe): every stripe's `up` ghost rows above from the stripe above,
// `down` ghost rows below from the stripe below.  Walks the edges between adjacent stripes top to bottom.
int add_batch(nz_sharded &sh, const std::vector<int> &planes, int up, int down) {
    std::vector<xfer> x;
    const size_t W = (size_t)sh.d.cols;
    // plan only (no context): the lists of the REAL rank asRank of asWorld, without addresses -- what nz_sharded_transfers
    // hands out, so that a test can lay the lists of all ranks side by side
    const int me = sh.dry ? sh.rank : (sh.comm ? sh.comm->rank : 0);
    const bool rehearsal = sh.d.asWorld > 0 && !sh.dry;
    auto local = [&](int v) { return v >= sh.rank * sh.S && v < (sh.rank + 1) * sh.S; };
    auto owner = [&](int v) { return rehearsal ? me : v / sh.S; };
    auto at = [&](int j, int id, size_t row) -> float * { return sh.dry ? nullptr : sh.plane(j, id) + row * W; };
    auto edge = [&](int ju, int jl, bool lu, bool ll, int ru, int rl) {
        // upper stripe (local index ju if lu) above lower stripe (jl if ll)
        for (int id : planes) {
            if (up > 0) {  // the lower stripe's top ghost rows <- the upper stripe's last owned rows
                xfer t{nullptr, nullptr, up * W, ru, rl};
                if (lu) t.send = at(ju, id, (size_t)(sh.st[ju].own1 - up));
                if (ll) t.recv = at(jl, id, (size_t)(sh.st[jl].own0 - up));
                x.push_back(t);
                if (lu) sh.bytes_sent += t.floats * sizeof(float);
            }
            if (down > 0) {  // the upper stripe's bottom ghost rows <- the lower stripe's first owned rows
                xfer t{nullptr, nullptr, down * W, rl, ru};
                if (ll) t.send = at(jl, id, (size_t)sh.st[jl].own0);
                if (lu) t.recv = at(ju, id, (size_t)sh.st[ju].own1);
                x.push_back(t);
                if (ll) sh.bytes_sent += t.floats * sizeof(float);
            }
        }
    };
    const int v0 = sh.rank * sh.S, v1 = v0 + sh.S;  // my stripes [v0, v1)
    if (rehearsal) {
        // an interior rank whose neighbours are played by its own stripes: the edge above my first stripe and the edge
        // below my last one close into a ring -- as many sends and receives per plane as the real rank posts
        for (int v = v0 + 1; v < v1; v++) edge(v - 1 - v0, v - v0, true, true, me, me);
        edge(sh.S - 1, 0, true, true, me, me);
    } else {
        for (int v = std::max(v0, 1); v <= std::min(v1, sh.V - 1); v++)
            edge(v - 1 - v0, v - v0, local(v - 1), local(v), owner(v - 1), owner(v));
    }
    sh.batches.push_back(std::move(x));
    return (int)sh.batches.size() - 1;
}

struct sh_layout {
    std::vector<launch_rad> radii;
    int sum_up = 0, sum_down = 0;
    int flow_first = -1, flow_last = -1, flow_widest = 0;
};

// radii of the launches, ghost rows per plane, geometry of my stripes (StripePlan of sharded.py)
int32_t plan_geometry(nz_sharded &sh, sh_layout &L) {
    NZ_TRY(launch_radii(sh.p, &sh.taps, &L.radii));
    const int mode = sh.d.haloMode;
    const bool recompute = mode != NZ_HALO_EXCHANGE;
    int widest = 1;
    for (const launch_rad &r : L.radii) {
        L.sum_up += r.up;
        L.sum_down += r.down;
        widest = std::max(widest, std::max(r.up, r.down));
    }
    sh.halo = recompute ? std::max(1, std::max(L.sum_up, L.sum_down)) : widest;
    for (size_t i = 0; i < L.radii.size(); i++)
        if (L.radii[i].kind == 2) {
            if (L.flow_first < 0) L.flow_first = (int)i;
            L.flow_last = (int)i;
            L.flow_widest = std::max(L.flow_widest, L.radii[i].up);
        }
    sh.need_state = L.flow_first >= 0 && L.flow_last > L.flow_first;
    const int base = sh.d.grows / sh.V, rem = sh.d.grows % sh.V;
    NZ_REQUIRE(base >= 1, "more stripes (%d) than rows (%d)", sh.V, sh.d.grows);
    sh.st.resize(sh.S);
    for (int j = 0; j < sh.S; j++) {
        sh_stripe &s = sh.st[j];
        s.v = sh.rank * sh.S + j;
        s.g0 = s.v * base + std::min(s.v, rem);
        s.nown = base + (s.v < rem ? 1 : 0);
        s.rows = s.nown + 2 * sh.halo;
        s.own0 = sh.halo;
        s.own1 = sh.halo + s.nown;
        s.grow0 = s.g0 - sh.halo;
        // an exchange takes ghost rows from the adjacent stripe only; recomputed ghost rows may reach further
        NZ_REQUIRE(mode == NZ_HALO_RECOMPUTE || s.nown >= sh.halo || sh.V == 1,
                   "stripe of %d rows is thinner than the %d ghost rows it must hand to its neighbour", s.nown, sh.halo);
    }
    NZ_REQUIRE((size_t)sh.st[0].rows * sh.d.cols < ((size_t)1 << 31), "stripe plane too large");
    NZ_REQUIRE(!sh.d.externalSource || mode != NZ_HALO_RECOMPUTE,
               "an external source plane cannot be recomputed: use an exchange mode");
    return NZ_OK;
}

// the launch plan (pipeline_steps of sharded.py, all local stripes in lockstep); the planes must exist
int32_t build_program(nz_sharded &sh, const sh_layout &L) {
    const std::vector<launch_rad> &radii = L.radii;
    const int mode = sh.d.haloMode;
    const bool recompute = mode != NZ_HALO_EXCHANGE;
    const int sum_up = L.sum_up, sum_down = L.sum_down, flow_first = L.flow_first, flow_last = L.flow_last,
              flow_widest = L.flow_widest;
    auto emit = [&](sh_op o) { sh.prog.push_back(o); };
    auto mark = [&](int i) { emit(sh_op{OP_MARK, -1, i, 0, 0, 0, 0, 0, 0, 0, 0, -1}); };
    auto exchange = [&](std::vector<int> planes, int up, int down) {
        const int b = add_batch(sh, planes, up, down);
        emit(sh_op{OP_XBEGIN, -1, (int)planes.size(), up, down, 0, 0, planes[0], 0, 0, 0, b});
    };
    auto finish = [&]() { emit(sh_op{OP_XFINISH, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1}); };

    int cur = 0, nxt = 1, s_cur = 0, s_nxt = 1;
    int need_up = recompute ? sum_up : 0, need_down = recompute ? sum_down : 0;
    mark(0);
    if (!sh.d.externalSource) {
        for (int j = 0; j < sh.S; j++) {
            const win w = mode == NZ_HALO_EXCHANGE_ONCE ? win{sh.st[j].own0, sh.st[j].own1}
                                                        : widened(sh, sh.st[j], need_up, need_down);
            emit(sh_op{OP_NOISE, j, 0, 0, 0, w.own0, w.own1, cur, cur, 0, 0, -1});
        }
    }
    if (mode == NZ_HALO_EXCHANGE_ONCE && (need_up > 0 || need_down > 0)) {
        exchange({cur}, need_up, need_down);  // the source plane's ghost rows for the whole pipeline, once
        finish();
    }
    int current = 0;
    // the exchange launch i needs before it runs, with the planes as they are assigned when it starts
    struct xspec { std::vector<int> planes; int up, down; };
    auto spec = [&](size_t i, int cur_, int s_cur_) {
        const launch_rad &r = radii[i];
        if (r.kind == 2 && (int)i == flow_first) return xspec{{cur_}, flow_widest, flow_widest};  // height: once, read by every flow launch
        if (r.kind == 2) return xspec{{2 + 5 * s_cur_, 3 + 5 * s_cur_, 4 + 5 * s_cur_, 5 + 5 * s_cur_, 6 + 5 * s_cur_}, r.up, r.down};
        return xspec{{cur_}, r.up, r.down};
    };
    // overlap 2 (border first): a launch produces the rows its neighbours need FIRST, the exchange for the NEXT launch
    // starts behind them and travels while the launch's interior rows run -- a launch never waits for rows it has just
    // asked for.  The noise stage is split the same way for the first launch's exchange.
    const bool border_first = !recompute && sh.d.overlap == 2;
    bool in_flight = false;  // border_first: the exchange for the launch about to be emitted has been posted
    if (border_first && !radii.empty() && !sh.prog.empty()) {
        // redo the noise ops: rows the first exchange sends, then the exchange, then the rest
        const xspec x0 = spec(0, cur, s_cur);
        std::vector<sh_op> noise;
        while (!sh.prog.empty() && sh.prog.back().kind == OP_NOISE) {
            noise.insert(noise.begin(), sh.prog.back());
            sh.prog.pop_back();
        }
        bool ok = !noise.empty();
        for (const sh_op &o : noise) ok = ok && (o.own1 - o.own0 > x0.up + x0.down);
        if (ok) {
            for (const sh_op &o : noise) {
                sh_op t = o, b = o;
                t.own1 = o.own0 + x0.down;
                b.own0 = o.own1 - x0.up;
                if (t.own1 > t.own0) emit(t);
                if (b.own1 > b.own0) emit(b);
            }
            exchange(x0.planes, x0.up, x0.down);
            in_flight = true;
            for (const sh_op &o : noise) {
                sh_op m = o;
                m.own0 = o.own0 + x0.down;
                m.own1 = o.own1 - x0.up;
                emit(m);
            }
        } else {
            for (const sh_op &o : noise) emit(o);
        }
    }
    for (size_t i = 0; i < radii.size(); i++) {
        const launch_rad &r = radii[i];
        while (current < r.kind) mark(++current);  // a stage left out: an empty interval
        if (recompute) {
            need_up -= r.up;
            need_down -= r.down;
        }
        const bool first = (int)i == flow_first, last = (int)i == flow_last;
        bool async = false;
        if (!recompute && !in_flight) {
            const xspec x = spec(i, cur, s_cur);
            exchange(x.planes, x.up, x.down);
            async = sh.d.overlap == 1;
        }
        // the launch on rows [own0, own1) of stripe j
        auto launch = [&](int j, win w) {
            if (w.own1 <= w.own0) return;
            if (r.kind == 1) emit(sh_op{OP_FILTER, j, r.n, 0, 0, w.own0, w.own1, cur, nxt, 0, 0, -1});
            else if (r.kind == 2) emit(sh_op{OP_FLOW, j, r.n, first, last, w.own0, w.own1, cur, nxt, s_cur, s_nxt, -1});
            else emit(sh_op{OP_EROSION, j, r.n, 0, 0, w.own0, w.own1, cur, nxt, 0, 0, -1});
        };
        std::vector<win> wins(sh.S);
        for (int j = 0; j < sh.S; j++) wins[j] = widened(sh, sh.st[j], need_up, need_down);
        // planes as launch i + 1 will find them
        int cur2 = cur, nxt2 = nxt, s_cur2 = s_cur, s_nxt2 = s_nxt;
        if (r.kind == 2) {
            std::swap(s_cur2, s_nxt2);
            if (last) std::swap(cur2, nxt2);
        } else {
            std::swap(cur2, nxt2);
        }
        if (border_first) {
            finish();  // the rows this launch reads have arrived (posted a launch ago)
            in_flight = false;
            bool split = i + 1 < radii.size();
            xspec xn{{}, 0, 0};
            if (split) {
                xn = spec(i + 1, cur2, s_cur2);
                // the flow stage's later launches exchange state planes a non-last launch writes, its first one the heights,
                // which no flow launch writes: either way the rows come out of THIS launch, unless it is a flow launch that
                // hands heights on unchanged
                for (int j = 0; j < sh.S; j++)
                    if (wins[j].own1 - wins[j].own0 <= xn.up + xn.down) split = false;
            }
            if (split) {
                for (int j = 0; j < sh.S; j++) {
                    if (xn.down > 0) launch(j, rows_window(wins[j], wins[j].own0, wins[j].own0 + xn.down));
                    if (xn.up > 0) launch(j, rows_window(wins[j], wins[j].own1 - xn.up, wins[j].own1));
                }
                exchange(xn.planes, xn.up, xn.down);
                in_flight = true;
                for (int j = 0; j < sh.S; j++) launch(j, rows_window(wins[j], wins[j].own0 + xn.down, wins[j].own1 - xn.up));
            } else {
                for (int j = 0; j < sh.S; j++) launch(j, wins[j]);
            }
        } else {
            // split_launch (sharded.py): interior rows while the ghost rows travel, border rows after the wait
            bool split = async;
            for (int j = 0; j < sh.S; j++)
                if (wins[j].own1 - r.down <= wins[j].own0 + r.up) split = false;
            if (!split) {
                if (!recompute) finish();
                for (int j = 0; j < sh.S; j++) launch(j, wins[j]);
            } else {
                for (int j = 0; j < sh.S; j++) launch(j, rows_window(wins[j], wins[j].own0 + r.up, wins[j].own1 - r.down));
                finish();
                for (int j = 0; j < sh.S; j++) {
                    if (r.up > 0) launch(j, rows_window(wins[j], wins[j].own0, wins[j].own0 + r.up));
                    if (r.down > 0) launch(j, rows_window(wins[j], wins[j].own1 - r.down, wins[j].own1));
                }
            }
        }
        cur = cur2; nxt = nxt2; s_cur = s_cur2; s_nxt = s_nxt2;
    }
    for (int k = current + 1; k <= 3; k++) mark(k);  // stages left out: empty intervals
    mark(4);
    sh.result_plane = cur;
    return NZ_OK;
}

int32_t run_op(nz_sharded &sh, const sh_op &o, nz_handle *marks) {
    nz_ctx *ctx = sh.ctx;
    const nz_terrain_params &p = sh.p;
    switch (o.kind) {
        case OP_MARK:
            return marks ? nz_ctx_finish(ctx, &marks[o.n]) : NZ_OK;
        case OP_NOISE: {
            const sh_stripe &s = sh.st[o.stripe];
            return nz_fractal_rows(ctx, ctx->stream, p.noiseType, sh.plane(o.stripe, o.src) + (size_t)o.own0 * sh.d.cols,
                                   o.own1 - o.own0, sh.d.cols, sh.d.cols, p.hurst, p.startingAmplitude, p.stepdown, p.detuneRate,
                                   p.octaves, sh.d.xpos, sh.d.zpos + s.grow0 + o.own0, p.noiseSize);
        }
        case OP_XBEGIN: {
            const std::vector<xfer> &b = sh.batches[o.batch];
            // overlap 0: the transfers run on the compute stream itself, between the launch that produced the rows and the
            // launch that reads them -- no second stream, no event hand-off
            const bool inl = sh.d.overlap == 0;
            if (inl && sh.comm && !b.empty() && sh.timing && sh.ev_used + 2 <= sh.ev_pool.size()) {
                NZ_HIP(hipEventRecord(sh.ev_pool[sh.ev_used], ctx->stream));
                NZ_TRY(post_batch(ctx, sh.comm, b.data(), b.size(), true));
                NZ_HIP(hipEventRecord(sh.ev_pool[sh.ev_used + 1], ctx->stream));
                sh.ev_used += 2;
                return NZ_OK;
            }
            return post_batch(ctx, sh.comm, b.data(), b.size(), inl);
        }
        case OP_XFINISH: {
            if (!sh.comm || !sh.comm->pending) return NZ_OK;
            if (sh.timing && sh.ev_used + 2 <= sh.ev_pool.size()) {
                NZ_HIP(hipEventRecord(sh.ev_pool[sh.ev_used], ctx->stream));
                NZ_TRY(finish_batch(ctx, sh.comm));
                NZ_HIP(hipEventRecord(sh.ev_pool[sh.ev_used + 1], ctx->stream));
                sh.ev_used += 2;
                return NZ_OK;
            }
            return finish_batch(ctx, sh.comm);
        }
        case OP_FILTER: {
            const nz_stripe v = sh.stripe_view(o.stripe, o.own0, o.own1);
            return nz_launch_conv_fused(ctx->stream, sh.plane(o.stripe, o.src), sh.plane(o.stripe, o.dst), nz_geom_from_stripe(v),
                                        sh.taps, o.n);
        }
        case OP_EROSION: {
            const nz_stripe v = sh.stripe_view(o.stripe, o.own0, o.own1);
            return nz_launch_erosion_fused(ctx->stream, sh.plane(o.stripe, o.src), sh.plane(o.stripe, o.dst),
                                           nz_geom_from_stripe(v), o.n);
        }
        case OP_FLOW: {
            const nz_stripe v = sh.stripe_view(o.stripe, o.own0, o.own1);
            const float *in[5];
            float *out[5];
            const bool first = o.a != 0, last = o.b != 0;
            for (int k = 0; k < 5; k++) {
                in[k] = first ? nullptr : sh.plane(o.stripe, 2 + 5 * o.sin + k);
                out[k] = last ? nullptr : sh.plane(o.stripe, 2 + 5 * o.sout + k);
            }
            return nz_launch_flow_fused(ctx->stream, sh.plane(o.stripe, o.src), first ? nullptr : in, last ? nullptr : out,
                                        last ? sh.plane(o.stripe, o.dst) : nullptr, nullptr, nz_geom_from_stripe(v), o.n, first,
                                        last, p.normMin, p.normMax - p.normMin);
        }
    }
    nz_set_error("sharded plan: unknown op %d", o.kind);
    return NZ_ERR_INVALID;
}

}  // namespace

extern "C" int32_t nz_sharded_create(nz_ctx *ctx, nz_comm *comm, const nz_sharded_desc *desc, const nz_terrain_params *params,
                                     nz_sharded **out) {
    NZ_REQUIRE(desc && params && out, "desc/params/out is NULL");
    *out = nullptr;
    NZ_REQUIRE(ctx || !comm, "a communicator without its context");
    NZ_REQUIRE(!comm || comm->ctx == ctx, "the communicator belongs to another context");
    NZ_REQUIRE(desc->grows >= 1 && desc->cols >= 1 && desc->cols <= 46340 * 4, "grid %d x %d out of range", desc->grows,
               desc->cols);
    NZ_REQUIRE(desc->haloMode >= NZ_HALO_RECOMPUTE && desc->haloMode <= NZ_HALO_EXCHANGE_ONCE, "unknown haloMode %d",
               desc->haloMode);
    NZ_REQUIRE(params->octaves >= 0 && (desc->externalSource || params->noiseSize != 0), "octaves < 0 or noiseSize == 0");
    NZ_REQUIRE(desc->externalSource ||
                   (params->noiseType >= 0 && params->noiseType <= NZ_NOISE_DOMAIN_ROTATED_SIMPLEX),
               "unknown noise type %d", params->noiseType);
    const int cworld = comm ? comm->world : 1, crank = comm ? comm->rank : 0;
    nz_sharded *sh = new nz_sharded();
    sh->ctx = ctx;
    sh->comm = comm;
    if (comm) comm->users++;
    sh->d = *desc;
    sh->p = *params;
    sh->rank = crank;
    sh->world = cworld;
    sh->dry = ctx == nullptr;  // plan only: nz_sharded_plan / nz_sharded_stripe work, nz_sharded_pipeline does not
    auto fail = [&](int32_t rc) {
        (void)nz_sharded_destroy(sh);
        return rc;
    };
    if (desc->asWorld > 0) {
        if (cworld != 1 || desc->asRank < 0 || desc->asRank >= desc->asWorld) {
            nz_set_error("a rehearsal (asRank %d of %d) runs on ONE rank", desc->asRank, desc->asWorld);
            return fail(NZ_ERR_INVALID);
        }
        if (!sh->dry && desc->haloMode != NZ_HALO_RECOMPUTE && !(desc->asRank > 0 && desc->asRank + 1 < desc->asWorld)) {
            nz_set_error("an exchange rehearsal plays an INTERIOR rank (0 < asRank < asWorld - 1)");
            return fail(NZ_ERR_INVALID);
        }
        sh->rank = desc->asRank;
        sh->world = desc->asWorld;
    }
    if (desc->stripes < sh->world || desc->stripes % sh->world != 0) {
        nz_set_error("stripes (%d) must be a positive multiple of the world size (%d)", desc->stripes, sh->world);
        return fail(NZ_ERR_INVALID);
    }
    sh->V = desc->stripes;
    sh->S = sh->V / sh->world;
    sh_layout L;
    int32_t rc = plan_geometry(*sh, L);
    if (rc) return fail(rc);
    if (sh->dry) {
        rc = build_program(*sh, L);
        if (rc) return fail(rc);
        *out = sh;
        return NZ_OK;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(NZ_ERR_HIP);
    // planes: A, B and -- only when the flow stage needs more than one launch -- two sets of five state planes
    for (sh_stripe &s : sh->st) {
        const size_t n = (size_t)s.rows * desc->cols;
        const size_t total = n * (2 + (sh->need_state ? 10 : 0));
        float *base = nullptr;
        hipError_t e = hipMalloc((void **)&base, total * sizeof(float));
        if (e != hipSuccess) {
            nz_set_error("hipMalloc(%zu floats) for a stripe: %s", total, hipGetErrorString(e));
            return fail(NZ_ERR_NOMEM);
        }
        s.A = base;
        s.B = base + n;
        if (sh->need_state) {
            s.S[0] = base + 2 * n;
            s.S[1] = base + 7 * n;
        }
        if (hipMemsetAsync(base, 0, total * sizeof(float), ctx->stream) != hipSuccess) return fail(NZ_ERR_HIP);
    }
    {
        const size_t n = 3 * (size_t)sh->S + 5 * (size_t)sh->V + 6;
        if (hipMalloc((void **)&sh->range_work, n * sizeof(float)) != hipSuccess) return fail(NZ_ERR_NOMEM);
    }
    rc = build_program(*sh, L);
    if (rc) return fail(rc);
    *out = sh;
    return NZ_OK;
}

extern "C" int32_t nz_sharded_destroy(nz_sharded *sh) {
    if (!sh) return NZ_OK;
    if (sh->ctx) {
        (void)hipSetDevice(sh->ctx->device);
        (void)hipStreamSynchronize(sh->ctx->stream);
        if (sh->comm && sh->comm->stream) (void)hipStreamSynchronize(sh->comm->stream);
    }
    for (sh_stripe &s : sh->st)
        if (s.A) (void)hipFree(s.A);
    if (sh->range_work) (void)hipFree(sh->range_work);
    for (hipEvent_t e : sh->ev_pool) (void)hipEventDestroy(e);
    if (sh->comm) sh->comm->users--;
    delete sh;
    return NZ_OK;
}

extern "C" int32_t nz_sharded_local_stripes(const nz_sharded *sh) { return sh ? sh->S : 0; }

extern "C" int32_t nz_sharded_stripe(const nz_sharded *sh, int32_t i, nz_stripe *st, float **source, float **result) {
    NZ_REQUIRE(sh, "sharded is NULL");
    NZ_REQUIRE(i >= 0 && i < sh->S, "local stripe %d outside [0, %d)", i, sh->S);
    if (st) *st = sh->stripe_view(i, sh->st[i].own0, sh->st[i].own1);
    if (source) *source = sh->st[i].A;
    if (result) *result = sh->plane(i, sh->result_plane);
    return NZ_OK;
}

extern "C" int32_t nz_sharded_plan(const nz_sharded *sh, int32_t *records, int32_t max_records, int32_t *count) {
    NZ_REQUIRE(sh && count, "sharded/count is NULL");
    *count = (int32_t)sh->prog.size();
    if (!records) return NZ_OK;
    NZ_REQUIRE(max_records >= *count, "room for %d records, the plan holds %d", max_records, *count);
    for (size_t i = 0; i < sh->prog.size(); i++) {
        const sh_op &o = sh->prog[i];
        int32_t *r = records + 8 * i;
        r[0] = o.kind; r[1] = o.stripe; r[2] = o.n; r[3] = o.a; r[4] = o.b; r[5] = o.own0; r[6] = o.own1;
        r[7] = o.src | (o.dst << 8) | (o.sin << 16) | (o.sout << 24);
    }
    return NZ_OK;
}

extern "C" int32_t nz_sharded_pipeline(nz_ctx *ctx, nz_sharded *sh, nz_handle *marks, nz_handle dep, nz_handle *out) {
    NZ_TRY(nz_ctx_begin(ctx, dep));
    NZ_REQUIRE(sh, "sharded is NULL");
    NZ_REQUIRE(!sh->dry, "a plan-only object (created without a context) cannot run");
    NZ_REQUIRE(sh->ctx == ctx, "the sharded grid belongs to another context");
    for (const sh_op &o : sh->prog) NZ_TRY(run_op(*sh, o, marks));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_sharded_transfers(const nz_sharded *sh, int32_t *records, int32_t max_records, int32_t *count) {
    NZ_REQUIRE(sh && count, "sharded/count is NULL");
    int n = 0;
    for (const std::vector<xfer> &b : sh->batches) n += (int)b.size();
    *count = n;
    if (!records) return NZ_OK;
    NZ_REQUIRE(max_records >= n, "room for %d records, the plan holds %d", max_records, n);
    int i = 0;
    for (size_t b = 0; b < sh->batches.size(); b++)
        for (const xfer &t : sh->batches[b]) {
            int32_t *r = records + 4 * i++;
            r[0] = (int32_t)b; r[1] = t.src; r[2] = t.dst; r[3] = (int32_t)t.floats;
        }
    return NZ_OK;
}

extern "C" int32_t nz_sharded_traffic(const nz_sharded *sh, int32_t *exchanges, size_t *bytes_sent) {
    NZ_REQUIRE(sh, "sharded is NULL");
    if (exchanges) *exchanges = (int32_t)sh->batches.size();
    if (bytes_sent) *bytes_sent = sh->bytes_sent;
    return NZ_OK;
}

extern "C" int32_t nz_sharded_set_timing(nz_sharded *sh, int32_t on) {
    NZ_REQUIRE(sh && !sh->dry, "sharded is NULL or plan-only");
    NZ_HIP(hipSetDevice(sh->ctx->device));
    if (on && sh->ev_pool.empty()) {
        sh->ev_pool.resize(2048, nullptr);
        for (hipEvent_t &e : sh->ev_pool) NZ_HIP(hipEventCreate(&e));
    }
    sh->timing = on != 0;
    return NZ_OK;
}

extern "C" int32_t nz_sharded_exchange_ms(nz_sharded *sh, float *ms) {
    NZ_REQUIRE(sh && ms && !sh->dry, "sharded/ms is NULL or plan-only");
    NZ_HIP(hipSetDevice(sh->ctx->device));
    float total = 0.0f;
    for (size_t i = 0; i + 1 < sh->ev_used; i += 2) {
        NZ_HIP(hipEventSynchronize(sh->ev_pool[i + 1]));
        float t = 0.0f;
        NZ_HIP(hipEventElapsedTime(&t, sh->ev_pool[i], sh->ev_pool[i + 1]));
        total += t;
    }
    sh->ev_used = 0;
    *ms = total;
    return NZ_OK;
}

extern "C" int32_t nz_sharded_map_range(nz_ctx *ctx, nz_sharded *sh, float *res, float lim_min, float lim_max, nz_handle dep,
                                        nz_handle *out) {
    NZ_TRY(nz_ctx_begin(ctx, dep));
    NZ_REQUIRE(sh && res, "sharded/res is NULL");
    NZ_REQUIRE(sh->ctx == ctx, "the sharded grid belongs to another context");
    NZ_REQUIRE(!(lim_min != lim_min) && !(lim_max != lim_max), "the limits must not be NaN");
    NZ_REQUIRE(sh->d.asWorld == 0, "a rehearsal holds one rank's stripes only");
    float *scratch = nullptr;
    NZ_TRY(nz_ctx_scratch(ctx, nz_map_range_scratch_floats(), &scratch));
    float *triples = sh->range_work, *work = triples + 3 * sh->S;
    for (int j = 0; j < sh->S; j++) {
        const sh_stripe &s = sh->st[j];
        NZ_TRY(nz_launch_map_range(ctx->stream, sh->plane(j, sh->result_plane) + (size_t)s.own0 * sh->d.cols,
                                   (size_t)s.nown * sh->d.cols, __builtin_inff(), -__builtin_inff(), triples + 3 * j, scratch));
    }
    NZ_TRY(gather_and_fold(ctx, sh->comm, triples, sh->S, work, res, lim_min, lim_max));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_sharded_normalize(nz_ctx *ctx, nz_sharded *sh, const float *args, nz_handle dep, nz_handle *out) {
    NZ_TRY(nz_ctx_begin(ctx, dep));
    NZ_REQUIRE(sh && args, "sharded/args is NULL");
    NZ_REQUIRE(sh->ctx == ctx, "the sharded grid belongs to another context");
    for (int j = 0; j < sh->S; j++) {
        const sh_stripe &s = sh->st[j];
        NZ_TRY(nz_launch_normalize_args(ctx->stream, sh->plane(j, sh->result_plane) + (size_t)s.own0 * sh->d.cols,
                                        (size_t)s.nown * sh->d.cols, args));
    }
    return nz_ctx_finish(ctx, out);
}
