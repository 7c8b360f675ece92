// nz_runtime.cpp -- the runtime half of libnoize_hip.so: errors, contexts (one HIP stream each), the host-built
// noise tables, JobHandle markers and device tiles.  The stage entry points are in nz_stages.cpp; see
// include/noize_hip.h for the reference interface each entry replaces.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "nz_internal.hpp"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void nz_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *nz_last_error(void) { return g_err; }
extern "C" int32_t nz_version(void) { return NZ_VERSION; }

extern "C" int32_t nz_device_count(int32_t *count) {
    NZ_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        nz_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return NZ_ERR_NO_DEVICE;
    }
    *count = n;
    return NZ_OK;
}


// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
static constexpr size_t NZ_EVENT_RING = 4096;
static int32_t ctx_sync_all(nz_ctx *ctx);
static void registry_add(nz_ctx *ctx);
static void registry_remove(nz_ctx *ctx);
static bool registry_full();
static int32_t ctx_chain_check(nz_ctx *ctx, uint64_t waited_seq);
static constexpr uint64_t NZ_WAIT_ALL = ~0ull;  // nz_ctx_synchronize: everything issued so far


static float h_mod289(float x);
static float h_permute(float x);

static int32_t build_rgrad_table(nz_ctx *ctx) {
    // noise.psrnoise.  Every hash argument is an integer-valued float, so both
    // permutes are pure functions of small integers and are tabulated with the reference's own fp32
    // operations (the first one overflows 2^24 and rounds; the table reproduces that rounding because it is
    // computed the same way).  (cos u, sin u) of rgrad2 come from the host libm the CPU restatement calls.
    std::vector<int32_t> buf(NZ_PSR_T1 + 2 * NZ_PSR_T2 * 2);
    for (int i = 0; i < NZ_PSR_T1; i++) buf[i] = 8 * ((int32_t)h_permute((float)(i - NZ_PSR_O1)) + NZ_PSR_O2);
    float *t2 = reinterpret_cast<float *>(buf.data() + NZ_PSR_T1);
    const float rots[2] = {0.0f, 0.62f};  // PeriodicPerlinGetter / RotatedSimplexGetter, Fractal.cs:184,201
    for (int t = 0; t < 2; t++) {
        for (int j = 0; j < NZ_PSR_T2; j++) {
            float h = h_permute((float)(j - NZ_PSR_O2));
            float u = h * 0.0243902439f + rots[t];
            u = (u - floorf(u)) * 6.28318530718f;
            t2[(t * NZ_PSR_T2 + j) * 2 + 0] = cosf(u);
            t2[(t * NZ_PSR_T2 + j) * 2 + 1] = sinf(u);
        }
    }
    NZ_HIP(hipMalloc((void **)&ctx->d_rgrad, buf.size() * sizeof(int32_t)));
    NZ_HIP(hipMemcpy(ctx->d_rgrad, buf.data(), buf.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return NZ_OK;
}

// Host copies of the reference's hash helpers; plain IEEE fp32, no contraction.
static float h_mod289(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
static float h_permute(float x) { return h_mod289((34.0f * x + 1.0f) * x); }

static float h_mod7(float x) { return x - floorf(x * (1.0f / 7.0f)) * 7.0f; }

static int32_t build_simplex_tables(nz_ctx *ctx) {
    // Lattice tables of nz_fractal.hip (snoise2_tab, cnoise2_tab, cellular_rect_tab), every entry computed
    // with exactly the operations of the corresponding Unity.Mathematics.noise function:
    //   simplex : T1[292] int = 16*permute(i);      T2[580] float4 = {a0, h, 1.79284291400159 - 0.85373472095314*(a0*a0+h*h), 0}
    //   perlin  : P1[292] int = 8*permute(i);       P2[584] float2 = {gx*norm, gy*norm} of permute(j)
    //   cellular: C1[292] int = 8*permute(i-1);     C2[584] float2 = {ox, oy} of permute(a-1)
    //   3-D     : P3[580] int = permute(j);           G3c[292] / G3s[292] float4 = normalised corner gradient of
    //             noise.cnoise(float3) / noise.snoise(float3) for the final hash value 0..288
    constexpr int T1 = 292, T2 = 580, B1 = 292, B2 = 584, P3 = 580, G3 = 292;
    std::vector<int32_t> buf(T1 + T2 * 4 + 2 * (B1 + B2 * 2) + P3 + 2 * G3 * 4);
    for (int i = 0; i < T1; i++) buf[i] = 16 * (int32_t)h_permute((float)i);
    float *t2 = reinterpret_cast<float *>(buf.data() + T1);
    for (int j = 0; j < T2; j++) {
        float p = h_permute((float)j);
        float y = p * 0.024390243902439f;
        float x = 2.0f * (y - floorf(y)) - 1.0f;
        float h = fabsf(x) - 0.5f;
        float ox = floorf(x + 0.5f);
        float a0 = x - ox;
        float nrm = 1.79284291400159f - 0.85373472095314f * (a0 * a0 + h * h);
        t2[4 * j + 0] = a0;
        t2[4 * j + 1] = h;
        t2[4 * j + 2] = nrm;
        t2[4 * j + 3] = 0.0f;
    }
    int32_t *p1 = buf.data() + T1 + T2 * 4;
    float *p2 = reinterpret_cast<float *>(p1 + B1);
    for (int i = 0; i < B1; i++) p1[i] = 8 * (int32_t)h_permute((float)i);
    for (int j = 0; j < B2; j++) {  // noise.cnoise: gradient of i = permute(permute(ix) + iy), normalised
        float i = h_permute((float)j);
        float y = i * (1.0f / 41.0f);
        float g = (y - floorf(y)) * 2.0f - 1.0f;
        float gy = fabsf(g) - 0.5f;
        float tx = floorf(g + 0.5f);
        float gx = g - tx;
        float nrm = 1.79284291400159f - 0.85373472095314f * (gx * gx + gy * gy);
        p2[2 * j + 0] = gx * nrm;
        p2[2 * j + 1] = gy * nrm;
    }
    int32_t *c1 = p1 + B1 + B2 * 2;
    float *c2 = reinterpret_cast<float *>(c1 + B1);
    const float K = 0.142857142857f, Ko = 0.428571428571f;
    for (int i = 0; i < B1; i++) c1[i] = 8 * (int32_t)h_permute((float)(i - 1));
    for (int a = 0; a < B2; a++) {  // noise.cellular: ox = frac(p*K) - Ko; oy = mod7(floor(p*K))*K - Ko
        float p = h_permute((float)(a - 1));
        float pk = p * K;
        c2[2 * a + 0] = (pk - floorf(pk)) - Ko;
        c2[2 * a + 1] = h_mod7(floorf(pk)) * K - Ko;
    }
    int32_t *p3 = c1 + B1 + B2 * 2;
    for (int j = 0; j < P3; j++) p3[j] = (int32_t)h_permute((float)j);
    float *g3c = reinterpret_cast<float *>(p3 + P3), *g3s = g3c + G3 * 4;
    auto step = [](float y, float x) { return x >= y ? 1.0f : 0.0f; };  // math.step(y, x)
    for (int h = 0; h < G3; h++) {
        {  // noise.cnoise(float3): gradient decode of ixy0 / ixy1
            float gx = (float)h * (1.0f / 7.0f);
            float t = floorf(gx) * (1.0f / 7.0f);
            float gy = (t - floorf(t)) - 0.5f;
            gx = gx - floorf(gx);
            float gz = 0.5f - fabsf(gx) - fabsf(gy);
            float sz = step(gz, 0.0f);
            gx -= sz * (step(0.0f, gx) - 0.5f);
            gy -= sz * (step(0.0f, gy) - 0.5f);
            float nr = 1.79284291400159f - 0.85373472095314f * (gx * gx + gy * gy + gz * gz);
            g3c[4 * h + 0] = gx * nr;
            g3c[4 * h + 1] = gy * nr;
            g3c[4 * h + 2] = gz * nr;
            g3c[4 * h + 3] = 0.0f;
        }
        {  // noise.snoise(float3): p -> (x, y, h) on the 7x7 grid, octahedron fold, normalisation
            const float n_ = 0.142857142857f;
            const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
            float pp = (float)h;
            float j = pp - 49.0f * floorf(pp * nsz * nsz);
            float x_ = floorf(j * nsz);
            float y_ = floorf(j - 7.0f * x_);
            float X = x_ * nsx + nsy, Y = y_ * nsx + nsy;
            float H = 1.0f - fabsf(X) - fabsf(Y);
            float sx = floorf(X) * 2.0f + 1.0f, sy = floorf(Y) * 2.0f + 1.0f;
            float sh = -step(H, 0.0f);
            float ax = X + sx * sh, ay = Y + sy * sh;
            float nr = 1.79284291400159f - 0.85373472095314f * (ax * ax + ay * ay + H * H);
            g3s[4 * h + 0] = ax * nr;
            g3s[4 * h + 1] = ay * nr;
            g3s[4 * h + 2] = H * nr;
            g3s[4 * h + 3] = 0.0f;
        }
    }
    NZ_HIP(hipMalloc(&ctx->d_simplex, buf.size() * sizeof(int32_t)));
    NZ_HIP(hipMemcpy(ctx->d_simplex, buf.data(), buf.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return NZ_OK;
}

static int32_t ctx_create(int32_t device, hipStream_t stream, bool own, nz_ctx **out) {
    NZ_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        nz_set_error("no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "count 0");
        return NZ_ERR_NO_DEVICE;
    }
    NZ_REQUIRE(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
    NZ_REQUIRE(!registry_full(), "this process has created 2^24 contexts: a handle cannot name another one");
    NZ_HIP(hipSetDevice(device));
    nz_ctx *ctx = new nz_ctx();
    ctx->device = device;
    if (own) {
        hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete ctx;
            nz_set_error("hipStreamCreate: %s", hipGetErrorString(se));
            return NZ_ERR_HIP;
        }
        ctx->owns_stream = true;
    } else {
        ctx->stream = stream;
    }
    // The chained filter launches rest on workgroups being started in index order, round-robin over the XCDs; a CU mask
    // takes that away (a bounded wait would time out and cost a tile): such a process runs separate launches from the start
    for (const char *v : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"})
        if (getenv(v) && *getenv(v)) ctx->chain_off = true;
    int32_t rc = build_rgrad_table(ctx);
    if (rc == NZ_OK) rc = build_simplex_tables(ctx);
    if (rc != NZ_OK) {
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return rc;
    }
    registry_add(ctx);
    *out = ctx;
    return NZ_OK;
}

int nz_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            return 256;
        return v;
    }();
    return n;
}

extern "C" int32_t nz_ctx_create(int32_t device, nz_ctx **out) { return ctx_create(device, nullptr, true, out); }

extern "C" int32_t nz_ctx_create_on_stream(int32_t device, void *hip_stream, nz_ctx **out) {
    return ctx_create(device, (hipStream_t)hip_stream, false, out);
}

extern "C" int32_t nz_ctx_destroy(nz_ctx *ctx) {
    if (!ctx) return NZ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    registry_remove(ctx);  // from here on its handles read as completed
    for (hipEvent_t ev : ctx->events)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->d_rgrad) (void)hipFree(ctx->d_rgrad);
    if (ctx->d_simplex) (void)hipFree(ctx->d_simplex);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->chain_flags) (void)hipFree(ctx->chain_flags);
    if (ctx->chain_err) (void)hipHostFree(ctx->chain_err);
    if (ctx->chain_err_epoch) (void)hipFree(ctx->chain_err_epoch);
    if (ctx->pool_ctl) (void)hipFree(ctx->pool_ctl);
    if (ctx->pool_hint) (void)hipHostFree(ctx->pool_hint);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return NZ_OK;
}

extern "C" int32_t nz_ctx_synchronize(nz_ctx *ctx) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_TRY_(ctx_sync_all(ctx));
    return ctx_chain_check(ctx, NZ_WAIT_ALL);
}

extern "C" void *nz_ctx_stream(nz_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int32_t nz_ctx_device(nz_ctx *ctx) { return ctx ? ctx->device : -1; }

static int32_t ctx_sync_all(nz_ctx *ctx) {
    NZ_HIP(hipStreamSynchronize(ctx->stream));
    return NZ_OK;
}
// ---- context registry: a handle value carries the id of the context that issued it ------------------------------
static std::mutex g_reg_mx;
static std::vector<nz_ctx *> g_reg;  // index = ctx id - 1; nullptr once destroyed

static void registry_add(nz_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_reg_mx);
    // ids are never reused (24 bits of a handle: 16 M contexts per process): a stale handle of a destroyed context keeps
    // naming that context, whose slot stays empty, and so reads as completed -- it can never alias a later context's marker
    g_reg.push_back(ctx);
    ctx->id = (uint32_t)g_reg.size();
}
static_assert(NZ_HANDLE_SEQ_BITS == 40, "a handle holds 24 bits of context id above 40 bits of sequence number");
static bool registry_full() {
    std::lock_guard<std::mutex> lk(g_reg_mx);
    return g_reg.size() >= ((size_t)1 << (64 - NZ_HANDLE_SEQ_BITS)) - 1;
}

static void registry_remove(nz_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_reg_mx);
    if (ctx->id && ctx->id <= g_reg.size() && g_reg[ctx->id - 1] == ctx) g_reg[ctx->id - 1] = nullptr;
}

static inline uint32_t handle_ctx_id(nz_handle h) { return (uint32_t)(h >> NZ_HANDLE_SEQ_BITS); }
static inline uint64_t handle_seq(nz_handle h) { return h & NZ_HANDLE_SEQ_MASK; }

static bool seq_live(nz_ctx *c, uint64_t q) { return q != 0 && q <= c->last_seq && q + NZ_EVENT_RING > c->last_seq; }

// Runs fn(owner, seq) with the registry locked (the owner cannot be destroyed meanwhile) and the owner's ring
// locked.  owner == nullptr: the issuing context no longer exists -- nz_ctx_destroy synchronised its stream, so
// everything it ever issued has completed.
template <class F>
static int32_t with_owner(nz_handle h, F fn) {
    std::lock_guard<std::mutex> lk(g_reg_mx);
    uint32_t id = handle_ctx_id(h);
    NZ_REQUIRE(id >= 1 && id <= g_reg.size(), "handle %llu was never issued by a context of this process",
               (unsigned long long)h);
    nz_ctx *owner = g_reg[id - 1];
    if (!owner) return fn((nz_ctx *)nullptr, handle_seq(h));
    std::lock_guard<std::mutex> lk2(owner->hmx);
    return fn(owner, handle_seq(h));
}

// The event that stands for `q` of `owner`: its own while it is in the ring, otherwise the owner's newest marker
// (later in stream order, so waiting on it implies `q`).
static hipEvent_t event_for(nz_ctx *owner, uint64_t q) {
    if (seq_live(owner, q)) return owner->events[q % NZ_EVENT_RING];
    return owner->last_seq ? owner->events[owner->last_seq % NZ_EVENT_RING] : nullptr;
}

thread_local hipEvent_t nz_tls_stop_event = nullptr;
thread_local int nz_tls_float_mode = NZ_FLOAT_STRICT;

extern "C" int32_t nz_ctx_set_float_mode(nz_ctx *ctx, int32_t mode) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    NZ_REQUIRE(mode >= NZ_FLOAT_STRICT && mode <= NZ_FLOAT_RELAXED, "unknown float mode %d", mode);
    ctx->float_mode = mode;
    return NZ_OK;
}
extern "C" int32_t nz_ctx_float_mode(nz_ctx *ctx) { return ctx ? ctx->float_mode : -1; }

void nz_ctx_handle_rides(nz_ctx *ctx, bool wanted) {
    ctx->handle_rides = wanted;
}

void nz_ctx_arm_last_launch(nz_ctx *ctx) {
    if (!ctx->handle_rides) return;
    std::lock_guard<std::mutex> lk(ctx->hmx);
    if (ctx->events.empty()) ctx->events.assign(NZ_EVENT_RING, nullptr);
    const uint64_t q = ctx->last_seq + 1;
    if (q >= NZ_HANDLE_SEQ_MASK) return;  // nz_ctx_finish reports it
    hipEvent_t *ev = &ctx->events[q % NZ_EVENT_RING];
    if (!*ev && hipEventCreate(ev) != hipSuccess) {
        *ev = nullptr;
        return;  // nz_ctx_finish records the old way and reports what fails
    }
    nz_tls_stop_event = *ev;
    ctx->armed_seq = q;
}

int32_t nz_ctx_begin(nz_ctx *ctx, nz_handle dep) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    nz_tls_stop_event = nullptr;  // (an entry that failed between arming and finishing)
    ctx->armed_seq = 0;
    ctx->handle_rides = false;
    nz_tls_float_mode = ctx->float_mode;
    NZ_HIP(hipSetDevice(ctx->device));
    if (dep == 0) return NZ_OK;  // default(JobHandle)
    if (handle_ctx_id(dep) == ctx->id) {
        // all work of a ctx is ordered on its stream: a dependency on one of its own handles is already satisfied
        NZ_REQUIRE(handle_seq(dep) <= ctx->last_seq, "dependency handle %llu was never issued",
                   (unsigned long long)dep);
        return NZ_OK;
    }
    // a handle of ANOTHER context (another HIP stream): this stream waits for its marker on the device, the host
    // does not block (JobHandle dependencies between pipelines, ReducePipeline.cs:82-148, PipelineStateLock.cs:12-39)
    return with_owner(dep, [&](nz_ctx *owner, uint64_t q) -> int32_t {
        if (!owner) return NZ_OK;
        NZ_REQUIRE(q <= owner->last_seq, "dependency handle %llu was never issued", (unsigned long long)dep);
        hipEvent_t ev = event_for(owner, q);
        if (ev) NZ_HIP(hipStreamWaitEvent(ctx->stream, ev, 0));
        return NZ_OK;
    });
}

int32_t nz_ctx_finish(nz_ctx *ctx, nz_handle *out) {
    // the entry's last launch took the event with it: the handle is that launch's completion
    const uint64_t armed = ctx->armed_seq;
    const bool rode = armed != 0 && nz_tls_stop_event == nullptr;
    nz_tls_stop_event = nullptr;
    ctx->armed_seq = 0;
    ctx->handle_rides = false;
    if (!out) return NZ_OK;
    std::lock_guard<std::mutex> lk(ctx->hmx);
    if (rode && armed == ctx->last_seq + 1) {
        ctx->last_seq = armed;
        *out = ((uint64_t)ctx->id << NZ_HANDLE_SEQ_BITS) | armed;
        return NZ_OK;
    }
    if (ctx->events.empty()) ctx->events.assign(NZ_EVENT_RING, nullptr);
    uint64_t q = ctx->last_seq + 1;
    NZ_REQUIRE(q < NZ_HANDLE_SEQ_MASK, "handle sequence exhausted");
    hipEvent_t *ev = &ctx->events[q % NZ_EVENT_RING];
    if (!*ev) NZ_HIP(hipEventCreate(ev));
    NZ_HIP(hipEventRecord(*ev, ctx->stream));
    ctx->last_seq = q;
    *out = ((uint64_t)ctx->id << NZ_HANDLE_SEQ_BITS) | q;
    return NZ_OK;
}

int32_t nz_ctx_pool_state(nz_ctx *ctx) {
    if (!ctx->pool_ctl) {
        NZ_HIP(hipMalloc((void **)&ctx->pool_ctl, 64));
        NZ_HIP(hipMemsetAsync(ctx->pool_ctl, 0, 64, ctx->stream));
    }
    if (!ctx->pool_hint) {
        NZ_HIP(hipHostMalloc((void **)&ctx->pool_hint, 64, hipHostMallocMapped));
        ctx->pool_hint[0] = ctx->pool_hint[1] = 0;  // job 0: nothing known yet
        NZ_HIP(hipHostGetDevicePointer((void **)&ctx->pool_hint_dev, ctx->pool_hint, 0));
    }
    return NZ_OK;
}

int32_t nz_ctx_scratch(nz_ctx *ctx, size_t floats, float **out) {
    if (floats > ctx->scratch_floats) {
        if (ctx->scratch) {
            NZ_TRY_(ctx_sync_all(ctx));
            NZ_HIP(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_floats = 0;
        }
        hipError_t e = hipMalloc((void **)&ctx->scratch, floats * sizeof(float));
        if (e != hipSuccess) {
            nz_set_error("hipMalloc(%zu floats): %s", floats, hipGetErrorString(e));
            return NZ_ERR_NOMEM;
        }
        ctx->scratch_floats = floats;
    }
    *out = ctx->scratch;
    return NZ_OK;
}

// The flags of the chained launches.  Flags are compared with an epoch that grows by one per launch, so stale contents never
// match; a (re)allocated array is zeroed and the epoch restarts above zero.
// The context's error words: mapped host memory a kernel whose bounded wait gives up stores to (system scope) and the host
// reads with a load at its next synchronisation.  [0] = a chained filter launch (NZ_ERR_RETRY), [1] = the pile solver's ticket
// kernel (an internal error: its wait cannot time out unless the protocol is broken).
int32_t nz_ctx_error_word(nz_ctx *ctx, unsigned **err_host) {
    if (!ctx->chain_err) {
        NZ_HIP(hipHostMalloc((void **)&ctx->chain_err, 64, hipHostMallocMapped));
        ctx->chain_err[0] = ctx->chain_err[1] = 0;
        NZ_HIP(hipHostGetDevicePointer((void **)&ctx->chain_err_dev, ctx->chain_err, 0));
    }
    *err_host = ctx->chain_err_dev;
    return NZ_OK;
}

int32_t nz_ctx_chain_state(nz_ctx *ctx, size_t items, int **flags, unsigned *epoch, unsigned **err_host, unsigned **err_epoch) {
    NZ_TRY_(nz_ctx_error_word(ctx, err_host));
    if (!ctx->chain_err_epoch) {
        NZ_HIP(hipMalloc((void **)&ctx->chain_err_epoch, 4));
        NZ_HIP(hipMemsetAsync(ctx->chain_err_epoch, 0xff, 4, ctx->stream));
    }
    if (items > ctx->chain_flags_n) {
        if (ctx->chain_flags) {
            NZ_TRY_(ctx_sync_all(ctx));
            NZ_HIP(hipFree(ctx->chain_flags));
            ctx->chain_flags = nullptr;
            ctx->chain_flags_n = 0;
        }
        size_t n = items + items / 2 + 1024;
        NZ_HIP(hipMalloc((void **)&ctx->chain_flags, n * sizeof(int)));
        NZ_HIP(hipMemsetAsync(ctx->chain_flags, 0, n * sizeof(int), ctx->stream));
        ctx->chain_flags_n = n;
    }
    if (++ctx->chain_epoch == 0) ctx->chain_epoch = 1;  // 0 is what a fresh flag holds
    // the handle sequence number this launch is issued at: the next handle of the context is the launching entry's own or a
    // later one
    ctx->chain_marks[ctx->chain_epoch % 64] = {ctx->chain_epoch, ctx->last_seq + 1};
    *flags = ctx->chain_flags;
    *epoch = ctx->chain_epoch;
    *err_epoch = ctx->chain_err_epoch;
    return NZ_OK;
}

// A kernel whose bounded wait gave up has raised one of the context's error words (mapped host memory: no device-to-host
// copy on the host's wait path): reported wherever the host waits for work the failure can have touched.  `waited_seq`: the
// sequence number of the handle the host has just waited for (NZ_WAIT_ALL: the whole stream).
static int32_t ctx_chain_check(nz_ctx *ctx, uint64_t waited_seq) {
    if (!ctx->chain_err) return NZ_OK;
    volatile unsigned *w = reinterpret_cast<volatile unsigned *>(ctx->chain_err);
    if (w[1]) {
        w[1] = 0;
        ctx->pile_ticket_off = true;  // this context runs a launch per colour from now on
        nz_set_error("internal: a block of the pile solver gave up waiting for a neighbouring block (nz_erode_height_maps); the "
                     "height plane is invalid.  The context runs the four colour launches from now on; nz_ctx_set_pile_safe(ctx, 1) "
                     "makes the job keep a copy of the plane and run again by itself");
        return NZ_ERR_HIP;
    }
    if (w[0]) {
        w[0] = 0;
        if (!ctx->retry_hi) {
            // first sight of it.  The wait of a chained launch terminates whatever happens (bounded poll), but it only makes
            // PROGRESS while the hardware starts the grid's workgroups in index order, round-robin over the XCDs -- observed
            // behaviour, not a contract (CU masking or a partitioned mode could break it): after one time-out the context
            // falls back to separate launches for good.
            ctx->chain_off = true;
            unsigned e = 0xffffffffu;
            (void)hipMemcpy(&e, ctx->chain_err_epoch, 4, hipMemcpyDeviceToHost);  // (slow path: once per context at most)
            uint64_t lo = 1;
            if (e != 0xffffffffu && ctx->chain_marks[e % 64].epoch == e) lo = ctx->chain_marks[e % 64].seq;
            ctx->retry_lo = lo;
            ctx->retry_hi = ctx->last_seq > lo ? ctx->last_seq : lo;
            ctx->retry_sync_pending = true;
            // The host may notice on a wait that is NOT for the failed work (an older handle: clean, below).  Until the
            // failure has actually been REPORTED, whatever it enqueues -- a stage that takes an in-window handle as its
            // dependency, a pipeline on the same plane -- is computed from the invalid plane too: the window stays open
            // (its upper end follows last_seq) until the first report.
            ctx->retry_open = true;
        }
    }
    if (!ctx->retry_hi) return NZ_OK;
    if (ctx->retry_open && ctx->last_seq > ctx->retry_hi) ctx->retry_hi = ctx->last_seq;
    bool hit;
    if (waited_seq == NZ_WAIT_ALL) {
        hit = ctx->retry_sync_pending;
        ctx->retry_sync_pending = false;
    } else {
        hit = waited_seq >= ctx->retry_lo && waited_seq <= ctx->retry_hi;
        if (hit) ctx->retry_sync_pending = false;
    }
    if (hit) ctx->retry_open = false;  // the host knows now: handles it issues from here on are its own decision
    if (!hit) return NZ_OK;
    nz_set_error("a chained kernel-filter launch timed out waiting for a producer tile: the plane that stage left (and whatever was "
                 "computed from it) is invalid; this context now runs filter stages as separate launches -- schedule the work item "
                 "again (NZ_ERR_RETRY)");
    return NZ_ERR_RETRY;
}

extern "C" int32_t nz_handle_record(nz_ctx *ctx, nz_handle *out) {
    NZ_REQUIRE(ctx && out, "ctx/out is NULL");
    NZ_HIP(hipSetDevice(ctx->device));
    return nz_ctx_finish(ctx, out);
}

// JobHandle.CombineDependencies(h0, h1, ...): a marker on ctx's stream that completes after every one of them
extern "C" int32_t nz_handle_combine(nz_ctx *ctx, const nz_handle *handles, int32_t count, nz_handle *out) {
    NZ_REQUIRE(ctx && out && (handles || count == 0) && count >= 0, "ctx/out/handles invalid");
    NZ_HIP(hipSetDevice(ctx->device));
    for (int32_t i = 0; i < count; i++) NZ_TRY_(nz_ctx_begin(ctx, handles[i]));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_handle_context_id(nz_handle h) { return (int32_t)handle_ctx_id(h); }
extern "C" int32_t nz_ctx_id(nz_ctx *ctx) { return ctx ? (int32_t)ctx->id : 0; }

// `ctx` may be any live context (kept for the signature's sake): the handle names its owner
extern "C" int32_t nz_handle_query(nz_ctx *ctx, nz_handle h, int32_t *is_completed) {
    NZ_REQUIRE(ctx && is_completed, "ctx/is_completed is NULL");
    if (h == 0) {
        *is_completed = 1;  // default(JobHandle).IsCompleted == true
        return NZ_OK;
    }
    return with_owner(h, [&](nz_ctx *owner, uint64_t q) -> int32_t {
        if (!owner) {
            *is_completed = 1;
            return NZ_OK;
        }
        NZ_REQUIRE(q <= owner->last_seq, "unknown handle");
        NZ_HIP(hipSetDevice(owner->device));
        hipError_t e = seq_live(owner, q) ? hipEventQuery(owner->events[q % NZ_EVENT_RING]) : hipStreamQuery(owner->stream);
        if (e == hipSuccess) {
            *is_completed = 1;
        } else if (e == hipErrorNotReady) {
            *is_completed = 0;
            (void)hipGetLastError();
        } else {
            nz_set_error("handle query: %s", hipGetErrorString(e));
            return NZ_ERR_HIP;
        }
        return NZ_OK;
    });
}

extern "C" int32_t nz_handle_wait(nz_ctx *ctx, nz_handle h) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    if (h == 0) return NZ_OK;
    hipEvent_t ev = nullptr;
    int dev = 0;
    int32_t rc = with_owner(h, [&](nz_ctx *owner, uint64_t q) -> int32_t {
        if (!owner) return NZ_OK;
        NZ_REQUIRE(q <= owner->last_seq, "unknown handle");
        ev = event_for(owner, q);
        dev = owner->device;
        return NZ_OK;
    });
    if (rc || !ev) return rc;
    // outside the locks: the host blocks here.  (An event of the ring is only destroyed by nz_ctx_destroy, which the
    // caller must not run concurrently with a wait on that context's handles.)
    NZ_HIP(hipSetDevice(dev));
    NZ_HIP(hipEventSynchronize(ev));
    // a failure is reported to whoever waits for work at or after the launch that failed -- on the context that ISSUED the
    // handle, whichever context the caller passed
    return with_owner(h, [&](nz_ctx *owner, uint64_t q) -> int32_t { return owner ? ctx_chain_check(owner, q) : NZ_OK; });
}

extern "C" int32_t nz_handle_elapsed_ms(nz_ctx *ctx, nz_handle start, nz_handle stop, float *ms) {
    NZ_REQUIRE(ctx && ms, "ctx/ms is NULL");
    NZ_REQUIRE(handle_ctx_id(start) == ctx->id && handle_ctx_id(stop) == ctx->id, "handles of another context");
    NZ_REQUIRE(seq_live(ctx, handle_seq(start)) && seq_live(ctx, handle_seq(stop)), "handle expired or unknown");
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_HIP(hipEventElapsedTime(ms, ctx->events[handle_seq(start) % NZ_EVENT_RING],
                               ctx->events[handle_seq(stop) % NZ_EVENT_RING]));
    return NZ_OK;
}

// ---------------------------------------------------------------------------------------------
// tiles
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nz_tile_alloc(nz_ctx *ctx, size_t n_floats, float **out_dev) {
    NZ_REQUIRE(ctx && out_dev, "ctx/out is NULL");
    NZ_HIP(hipSetDevice(ctx->device));
    *out_dev = nullptr;
    hipError_t e = hipMalloc((void **)out_dev, (n_floats ? n_floats : 1) * sizeof(float));
    if (e != hipSuccess) {
        nz_set_error("hipMalloc(%zu floats): %s", n_floats, hipGetErrorString(e));
        return NZ_ERR_NOMEM;
    }
    return NZ_OK;
}

extern "C" int32_t nz_tile_free(nz_ctx *ctx, float *dev) {
    NZ_REQUIRE(ctx, "ctx is NULL");
    if (!dev) return NZ_OK;
    NZ_HIP(hipSetDevice(ctx->device));
    NZ_TRY_(ctx_sync_all(ctx));  // Dispose(handle): free after the work that uses it
    NZ_HIP(hipFree(dev));
    return NZ_OK;
}

extern "C" int32_t nz_tile_upload(nz_ctx *ctx, float *dev, const float *host, size_t n_floats, nz_handle dep,
                                  nz_handle *out) {
    int32_t rc = nz_ctx_begin(ctx, dep);
    if (rc) return rc;
    NZ_REQUIRE(dev && host, "dev/host is NULL");
    NZ_HIP(hipMemcpyAsync(dev, host, n_floats * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_bytes_download(nz_ctx *ctx, const void *dev, void *host, size_t n_bytes, nz_handle dep,
                                     nz_handle *out) {
    int32_t rc = nz_ctx_begin(ctx, dep);
    if (rc) return rc;
    NZ_REQUIRE(dev && host, "dev/host is NULL");
    NZ_HIP(hipMemcpyAsync(host, dev, n_bytes, hipMemcpyDeviceToHost, ctx->stream));
    return nz_ctx_finish(ctx, out);
}

extern "C" int32_t nz_tile_download(nz_ctx *ctx, const float *dev, float *host, size_t n_floats, nz_handle dep,
                                    nz_handle *out) {
    return nz_bytes_download(ctx, dev, host, n_floats * sizeof(float), dep, out);
}
