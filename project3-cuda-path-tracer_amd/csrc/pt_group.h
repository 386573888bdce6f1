// pt_group.h -- renderer contexts and device groups of the C ABI (include/pt_amd.h: pt_ctx_*, pt_group_*).
//
// A CONTEXT is one renderer instance (State): the C ABI's functions act on the calling thread's current one.  A GROUP is N contexts that
// render the row shards y % N of ONE frame -- one context per device of a node (the north star's tile-sharded 8 x MI355X), or
// several on one device -- behind one call each: the native multi-device host path the reference's own C++ host can sit on
// (host/pathtrace_shim.cpp honours PT_AMD_DEVICES; the reference is hard-wired to device 0, src/preview.cpp:107, src/pathtrace.cu:70-71).
// Pixels are independent and seeded by their GLOBAL index, so the shards' rows, put together, are the one-device frame bit for bit.
//
// The frame is assembled at readback, by the collective SURVEY 8e names where it exists:
//   "rccl reduce"  every member accumulates into a zero-padded FULL frame on its own device (own rows in place), ncclReduce(sum, root 0) of
//                  the N frames (x + 0 is exact: disjoint rows) on one stream per device, one D2H copy from device 0.  Taken when the
//                  members sit on N DISTINCT devices and librccl.so loads (dlopen: the product library does not link against RCCL);
//                  single process, ncclCommInitAll.
//   "host gather"  every member accumulates its own rows, packed (PT_FLAG_ACCUM_SHARD_ROWS); readback copies each shard to the host and
//                  interleaves the rows.  Members that share a device (this is what a one-GPU box can run), no RCCL, or PT_AMD_COLLECTIVE=host.
// The RCCL leg has run on ONE device only (a one-member group, tests/test_gpu_contexts.py): no multi-GPU node was available to the build
// rounds -- it is written to SURVEY 8e's letter and UNMEASURED across devices.
// Included by pt_api.hip inside its extern "C" block.
#pragma once
#include <dlfcn.h>

struct PtContext { State st; };

PtContext *pt_ctx_create(void) {
    PtContext *c = new (std::nothrow) PtContext();
    if (!c) { fail(PT_ERR_INVALID, "pt_ctx_create: out of memory"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_ctxMutex);
    g_contexts.push_back(&c->st);
    return c;
}
PtContext *pt_ctx_current(void) { return t_ctx == &g_default ? nullptr : reinterpret_cast<PtContext *>(t_ctx); }
int pt_ctx_make_current(PtContext *ctx) {
    if (ctx) {
        std::lock_guard<std::mutex> lock(g_ctxMutex);
        if (std::find(g_contexts.begin(), g_contexts.end(), &ctx->st) == g_contexts.end()) return fail(PT_ERR_INVALID, "pt_ctx_make_current: not a live context");
    }
    t_ctx = ctx ? &ctx->st : &g_default;
    // an initialised renderer lives on ONE device: its calls allocate, launch and copy there
    if (R().init || R().nslots > 0) HIPCHECK(hipSetDevice(R().device));
    return PT_OK;
}
int pt_ctx_destroy(PtContext *ctx) {
    if (!ctx) return PT_OK;
    {
        std::lock_guard<std::mutex> lock(g_ctxMutex);
        auto it = std::find(g_contexts.begin(), g_contexts.end(), &ctx->st);
        if (it == g_contexts.end()) return fail(PT_ERR_INVALID, "pt_ctx_destroy: not a live context");
        g_contexts.erase(it);
    }
    State *const prev = t_ctx;
    t_ctx = &ctx->st;
    free_renderer();
    t_ctx = prev == &ctx->st ? &g_default : prev;
    delete ctx;
    return PT_OK;
}

// ---- groups ---------------------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (CommInitAll && CommDestroy && Reduce && GroupStart && GroupEnd && GetErrorString) return true;
        dlclose(lib);
        lib = nullptr;
        return false;
    }
} g_rccl;
constexpr int kNcclFloat = 7, kNcclSum = 0;      // rccl.h: ncclFloat32, ncclSum
// the calling thread's current context, put back when a group call returns
struct CurrentGuard {
    State *saved = t_ctx;
    int dev = -1;
    CurrentGuard() { (void)hipGetDevice(&dev); }
    ~CurrentGuard() {
        t_ctx = saved;
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};
}  // namespace

struct PtGroup {
    std::vector<PtContext *> ctx;
    std::vector<int> device;
    bool rccl = false;                     // the frame is assembled by ncclReduce (else: on the host)
    std::vector<void *> comm;              // [n] ncclComm_t
    std::vector<hipStream_t> stream;       // [n] one stream per device for the collective
    std::vector<float *> full;             // [n] rccl: the member's zero-padded full-frame accumulator (pt_init's accum_dev)
    float *reduced = nullptr;              // rccl: the reduce's result on device[0]
    int W = 0, H = 0;
    std::vector<float> stage;              // host gather: one shard's frame
    std::string how = "host gather";
};

static void group_release_buffers(PtGroup *g) {
    for (size_t i = 0; i < g->full.size(); ++i)
        if (g->full[i]) { (void)hipSetDevice(g->device[i]); (void)hipFree(g->full[i]); g->full[i] = nullptr; }
    if (g->reduced) { (void)hipSetDevice(g->device[0]); (void)hipFree(g->reduced); g->reduced = nullptr; }
}

int pt_group_create(PtGroup **out, int n, const int32_t *devices) {
    if (!out || n < 1 || n > 64) return fail(PT_ERR_INVALID, "pt_group_create: 1..64 members");
    const int ndev = count_devices();
    if (ndev < 1) return fail(PT_ERR_NO_GPU, "pt_group_create: no HIP device (this library has no CPU fallback)");
    register_exit_handler();
    PtGroup *g = new (std::nothrow) PtGroup();
    if (!g) return fail(PT_ERR_INVALID, "pt_group_create: out of memory");
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        const int d = devices ? devices[i] : i % ndev;
        if (d < 0 || d >= ndev) { delete g; return fail(PT_ERR_INVALID, "pt_group_create: device %d of %d", d, ndev); }
        for (int q : g->device) distinct = distinct && q != d;
        g->device.push_back(d);
    }
    for (int i = 0; i < n; ++i) {
        PtContext *c = pt_ctx_create();
        if (!c) { pt_group_destroy(g); return PT_ERR_INVALID; }
        g->ctx.push_back(c);
    }
    // RCCL: one rank per DISTINCT device, one process (ncclCommInitAll)
    const char *want = getenv("PT_AMD_COLLECTIVE");
    const bool forbid = want && !strcmp(want, "host");
    if (distinct && !forbid && (n > 1 || (want && !strcmp(want, "rccl"))) && g_rccl.load()) {
        CurrentGuard guard;
        g->comm.assign(n, nullptr);
        const int r = g_rccl.CommInitAll(g->comm.data(), n, g->device.data());
        if (r != 0) {
            g->comm.clear();
            fprintf(stderr, "pt_group_create: ncclCommInitAll failed (%s): the frame is assembled on the host\n", g_rccl.GetErrorString(r));
        } else {
            g->rccl = true;
            g->how = "rccl reduce";
            g->stream.assign(n, nullptr);
            for (int i = 0; i < n; ++i) {
                if (hipSetDevice(g->device[i]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking) != hipSuccess) {
                    pt_group_destroy(g);
                    return fail(PT_ERR_HIP, "pt_group_create: no stream on device %d", g->device[i]);
                }
            }
        }
    }
    *out = g;
    return PT_OK;
}

void pt_group_destroy(PtGroup *g) {
    if (!g) return;
    CurrentGuard guard;
    for (PtContext *c : g->ctx) (void)pt_ctx_destroy(c);
    group_release_buffers(g);
    for (size_t i = 0; i < g->stream.size(); ++i)
        if (g->stream[i]) { (void)hipSetDevice(g->device[i]); (void)hipStreamDestroy(g->stream[i]); }
    for (void *c : g->comm)
        if (c) (void)g_rccl.CommDestroy(c);
    delete g;
}

int pt_group_size(const PtGroup *g) { return g ? (int)g->ctx.size() : 0; }
const char *pt_group_collective(const PtGroup *g) { return g ? g->how.c_str() : ""; }

int pt_group_set_meshes(PtGroup *g, const PtMesh *meshes, int nmeshes) {
    if (!g) return fail(PT_ERR_INVALID, "pt_group_set_meshes: null group");
    CurrentGuard guard;
    for (PtContext *c : g->ctx) {
        t_ctx = &c->st;
        int rc = pt_set_meshes(meshes, nmeshes);
        if (rc) return rc;
    }
    return PT_OK;
}

int pt_group_init(PtGroup *g, const PtCamera *cam, const PtGeom *geoms, int ngeoms, const PtMaterial *mats, int nmats, int traceDepth,
                  const PtOptions *opts) {
    if (!g || !cam) return fail(PT_ERR_INVALID, "pt_group_init: null argument");
    CurrentGuard guard;
    const int n = (int)g->ctx.size();
    group_release_buffers(g);
    g->W = cam->resolution[0];
    g->H = cam->resolution[1];
    const size_t frameFloats = (size_t)std::max(g->W, 0) * (size_t)std::max(g->H, 0) * 3;
    if (g->rccl) {
        g->full.assign(n, nullptr);
        for (int i = 0; i < n; ++i) {
            HIPCHECK(hipSetDevice(g->device[i]));
            HIPCHECK(hipMalloc(&g->full[i], std::max<size_t>(frameFloats, 1) * sizeof(float)));
            HIPCHECK(hipMemset(g->full[i], 0, std::max<size_t>(frameFloats, 1) * sizeof(float)));
        }
        HIPCHECK(hipSetDevice(g->device[0]));
        HIPCHECK(hipMalloc(&g->reduced, std::max<size_t>(frameFloats, 1) * sizeof(float)));
    }
    for (int i = 0; i < n; ++i) {
        PtOptions o;
        memset(&o, 0, sizeof o);
        if (opts) o = *opts;
        o.shard_rank = i;
        o.shard_count = n;
        o.device = g->device[i];
        o.stream = nullptr;
        o.flags &= ~PT_FLAG_ACCUM_SHARD_ROWS;
        if (g->rccl) o.accum_dev = g->full[i];            // own rows in place, zeros elsewhere: what the reduce sums
        else { o.accum_dev = nullptr; o.flags |= PT_FLAG_ACCUM_SHARD_ROWS; }
        t_ctx = &g->ctx[i]->st;
        int rc = pt_init(cam, geoms, ngeoms, mats, nmats, traceDepth, &o);
        if (rc) return rc;
    }
    return PT_OK;
}

// every member enqueues its shard's wavefront batch (asynchronous: the devices run side by side)
int pt_group_iterate_batch(PtGroup *g, int frame, int first_iter, int count) {
    if (!g) return fail(PT_ERR_INVALID, "pt_group_iterate_batch: null group");
    CurrentGuard guard;
    for (PtContext *c : g->ctx) {
        int rc = pt_ctx_make_current(c);
        if (rc) return rc;
        if ((rc = pt_iterate_batch(frame, first_iter, count, nullptr))) return rc;
    }
    return PT_OK;
}

int pt_group_sync(PtGroup *g) {
    if (!g) return fail(PT_ERR_INVALID, "pt_group_sync: null group");
    CurrentGuard guard;
    for (PtContext *c : g->ctx) {
        int rc = pt_ctx_make_current(c);
        if (rc) return rc;
        if ((rc = pt_sync())) return rc;
    }
    return PT_OK;
}

// the whole frame's un-normalised running sum (W * H * 3 floats), as pt_readback delivers it for one device
int pt_group_readback(PtGroup *g, float *rgb_sum_host) {
    if (!g || !rgb_sum_host) return fail(PT_ERR_INVALID, "pt_group_readback: null argument");
    CurrentGuard guard;
    const int n = (int)g->ctx.size();
    const size_t frameFloats = (size_t)g->W * g->H * 3;
    int rc = pt_group_sync(g);                               // (reports a member's device fault)
    if (rc) return rc;
    if (g->rccl) {
        int r = g_rccl.GroupStart();
        for (int i = 0; i < n && r == 0; ++i) {
            HIPCHECK(hipSetDevice(g->device[i]));
            r = g_rccl.Reduce(g->full[i], i == 0 ? g->reduced : nullptr, frameFloats, kNcclFloat, kNcclSum, 0, g->comm[i], g->stream[i]);
        }
        const int r2 = g_rccl.GroupEnd();
        if (r != 0 || r2 != 0) return fail(PT_ERR_HIP, "pt_group_readback: ncclReduce failed: %s", g_rccl.GetErrorString(r != 0 ? r : r2));
        for (int i = 0; i < n; ++i) {
            HIPCHECK(hipSetDevice(g->device[i]));
            HIPCHECK(hipStreamSynchronize(g->stream[i]));
        }
        HIPCHECK(hipSetDevice(g->device[0]));
        HIPCHECK(hipMemcpy(rgb_sum_host, g->reduced, frameFloats * sizeof(float), hipMemcpyDeviceToHost));
        return PT_OK;
    }
    // host gather: pt_readback of a row shard delivers a full frame with the other shards' rows zero
    if (n == 1) {
        if ((rc = pt_ctx_make_current(g->ctx[0]))) return rc;
        return pt_readback(rgb_sum_host);
    }
    g->stage.resize(frameFloats);
    const size_t rowFloats = (size_t)g->W * 3;
    for (int i = 0; i < n; ++i) {
        if ((rc = pt_ctx_make_current(g->ctx[i]))) return rc;
        if ((rc = pt_readback(g->stage.data()))) return rc;
        for (int y = i; y < g->H; y += n) memcpy(rgb_sum_host + (size_t)y * rowFloats, g->stage.data() + (size_t)y * rowFloats, rowFloats * sizeof(float));
    }
    return PT_OK;
}

// the members' tallies, summed (every member counts the paths of its own rows)
int pt_group_counters(PtGroup *g, PtCounters *out) {
    if (!g || !out) return fail(PT_ERR_INVALID, "pt_group_counters: null argument");
    CurrentGuard guard;
    memset(out, 0, sizeof *out);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        int rc = pt_ctx_make_current(g->ctx[i]);
        if (rc) return rc;
        PtCounters c;
        if ((rc = pt_counters(&c))) return rc;
        for (int d = 0; d < PT_MAX_DEPTH + 2; ++d) { out->live[d] += c.live[d]; out->ended_early[d] += c.ended_early[d]; }
        out->light_hits += c.light_hits;
        out->misses += c.misses;
        out->iterations = i == 0 ? c.iterations : std::min(out->iterations, c.iterations);
        out->bounce_launches += c.bounce_launches;
        out->bounce_kernel_ms += c.bounce_kernel_ms;
    }
    return PT_OK;
}
