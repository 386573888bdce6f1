// pt_group.h -- renderer contexts and device groups of the C ABI (include/pt_amd.h: pt_ctx_*, pt_group_*).
//
// A CONTEXT is one renderer instance (State): the C ABI's functions act on the calling thread's current one.  A GROUP is N contexts that
// render the row shards y % N of ONE frame -- one context per device of a node (the north star's tile-sharded 8 x MI355X), or
// several on one device -- behind one call each: the native multi-device host path the reference's own C++ host can sit on
// (host/pathtrace_shim.cpp honours PT_AMD_DEVICES; the reference is hard-wired to device 0, src/preview.cpp:107, src/pathtrace.cu:70-71).
// Pixels are independent and seeded by their GLOBAL index, so the shards' rows, put together, are the one-device frame bit for bit.
//
// Round 6 -- how a group is laid out and driven:
//   * ONE zero-padded full-frame accumulator per DISTINCT device (GroupDevice::full).  The members on a device commit their own rows in
//     place into it (k_commit indexes by the global pixel; rows are disjoint), so members that share a device need no collective at all.
//   * ONE issuing THREAD per member (Worker): a member's launches are enqueued by its own host thread, so eight devices are fed side by
//     side instead of one after the other (a C2 shard of 1/8 is ~0.24 ms of GPU time per batch of 64 -- less than one thread needs to
//     enqueue eight members' launches).  PT_AMD_GROUP_THREADS=0: the calling thread issues everything (experiments).
//   * The frame is ASSEMBLED asynchronously (assemble(); pt_group_iterate / pt_group_reduce / the first pt_group_readback after a batch):
//       "rccl reduce"         distinct devices, librccl.so loaded (dlopen; single process, ncclCommInitAll): every member's commit
//                             ALSO writes its rows' new values into one of two SNAPSHOT frames of its device (k_commit's `snap`: the
//                             accumulator is read once, for the addition and the copy), then ONE ncclReduce(sum, float32, 3 W H, root =
//                             member 0's device) of the snapshots, in place at the root, on the device's collective stream behind the
//                             members' commit events -- x + 0 is exact: disjoint rows.  Nothing waits on the host: the reduce of
//                             iteration i runs while iteration i + 1 is committed into the OTHER snapshot and the batches traced ahead keep
//                             tracing; a snapshot is written again two calls later, behind its reduce's event.  pt_group_readback = one
//                             D2H copy of the latest reduced snapshot on the root's collective stream + a wait for THAT stream only.
//       "shared accumulator"  every member on ONE device: the accumulator is the frame; readback copies it behind the members' commit events.
//       "host gather"         several devices without RCCL (PT_AMD_COLLECTIVE=host, no librccl.so): every device's frame is copied to the host
//                             and the rows are taken from their owners' (synchronous; a fallback).
//   * ncclGroupStart / ncclGroupEnd are balanced on EVERY path: the loop between them records the first error and goes on to GroupEnd.
// Across DISTINCT devices the RCCL leg is UNMEASURED: no multi-GPU node was available to any build round (pt_group_collective() says so);
// what has run is the same pipeline in a one-rank communicator (PT_AMD_COLLECTIVE=rccl), bit-identical after every call.
// Included by pt_api.hip inside its extern "C" block.
#pragma once
#include <dlfcn.h>      // (the C++ headers this file needs -- <atomic>, <condition_variable>, <functional>, <thread> -- are included by pt_api.hip)

struct PtContext {
    State st;                       // (first: pt_ctx_current hands &st out as the context)
    std::atomic<int> users{0};      // threads this context is current on (pt_ctx_make_current)
};

namespace {
// the context pt_ctx_make_current bound to this thread: released when the thread rebinds or ends, so that pt_ctx_destroy can tell
// whether ANOTHER thread still acts on the context (its t_ctx would dangle)
struct CurrentRef {
    PtContext *held = nullptr;
    ~CurrentRef() {
        if (held) held->users.fetch_sub(1);
    }
};
thread_local CurrentRef t_ref;
}  // namespace

PtContext *pt_ctx_create(void) {
    PtContext *c = new (std::nothrow) PtContext();
    if (!c) { fail(PT_ERR_INVALID, "pt_ctx_create: out of memory"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_ctxMutex);
    g_contexts.push_back(&c->st);
    return c;
}
PtContext *pt_ctx_current(void) { return t_ctx == &g_default ? nullptr : reinterpret_cast<PtContext *>(t_ctx); }
int pt_ctx_make_current(PtContext *ctx) {
    {
        std::lock_guard<std::mutex> lock(g_ctxMutex);
        if (ctx && std::find(g_contexts.begin(), g_contexts.end(), &ctx->st) == g_contexts.end()) return fail(PT_ERR_INVALID, "pt_ctx_make_current: not a live context");
        if (t_ref.held != ctx) {
            if (t_ref.held) t_ref.held->users.fetch_sub(1);
            t_ref.held = ctx;
            if (ctx) ctx->users.fetch_add(1);
        }
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
        // current on ANOTHER thread: that thread's next pt_* call would act on freed memory
        if (ctx->users.load() > (t_ref.held == ctx ? 1 : 0))
            return fail(PT_ERR_INVALID, "pt_ctx_destroy: the context is current on another thread (pt_ctx_make_current(NULL) there first)");
        g_contexts.erase(it);
        if (t_ref.held == ctx) { t_ref.held = nullptr; ctx->users.fetch_sub(1); }
    }
    int dev = -1;
    (void)hipGetDevice(&dev);
    State *const prev = t_ctx;
    t_ctx = &ctx->st;
    free_renderer();
    t_ctx = prev == &ctx->st ? &g_default : prev;
    if (dev >= 0) (void)hipSetDevice(dev);          // (free_renderer switched to the renderer's device)
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
};
Rccl *rcclPtr() {
    static Rccl r;
    return &r;
}
#define rccl() (*rcclPtr())
std::mutex g_rcclMutex;                          // (loading; the calls themselves go through one group's host thread)
constexpr int kNcclFloat = 7, kNcclSum = 0;      // rccl.h: ncclFloat32, ncclSum
// the calling thread's current context and device, put back when a group call returns
struct CurrentGuard {
    State *saved = t_ctx;
    int dev = -1;
    CurrentGuard() { (void)hipGetDevice(&dev); }
    ~CurrentGuard() {
        t_ctx = saved;
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};

// One host thread per member: it owns the member's context (t_ctx) and device for its lifetime and enqueues what the group's calls post.
// (Spinning on an atomic word before sleeping on the condition variable was tried for the per-iteration protocol and measured no
// different -- 0.097 ms per pt_group_iterate with 8 members on one device either way: the members' commits, not the wake-ups, bound it.)
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool pending = false, quit = false;
    int rc = PT_OK;
    std::string err;
    void loop(State *st) {
        t_ctx = st;
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return pending || quit; });
                if (!pending && quit) return;
                f = job;
            }
            g_err.clear();
            const int r = f();
            {
                std::lock_guard<std::mutex> lk(m);
                rc = r;
                err = g_err;
                pending = false;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> f) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(f);
            pending = true;
        }
        cv.notify_all();
    }
    int wait(std::string *e) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !pending; });
        if (rc != PT_OK && e) *e = err;
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// what a group holds per DISTINCT device
struct GroupDevice {
    int device = 0;
    std::vector<int> members;              // indices into PtGroup::ctx
    float *full = nullptr;                 // the zero-padded full-frame accumulator the members on this device commit into
    float *snap[2] = {nullptr, nullptr};   // rccl: the members' commits copy their rows here (k_commit's `snap`); the reduce reads -- at the root: in place
    hipStream_t coll = nullptr;            // the collective's stream (and the read-back's)
    hipEvent_t evRed[2] = {nullptr, nullptr};     // the reduce that read snapshot k is through (collective stream)
    void *comm = nullptr;                  // ncclComm_t
};
}  // namespace

struct PtGroup {
    std::vector<PtContext *> ctx;
    std::vector<int> device;               // [n] member -> HIP device
    std::vector<int> devIndex;             // [n] member -> index into dev
    std::vector<hipStream_t> stream;       // [n] the member's own "caller's stream": commits, in call order
    std::vector<hipEvent_t> evCommit;      // [n] the member's commits so far (recorded behind every call's enqueue)
    std::vector<std::unique_ptr<Worker>> worker;   // [n] or empty (PT_AMD_GROUP_THREADS=0, one member)
    std::vector<GroupDevice> dev;          // distinct devices, dev[0] = member 0's = the reduce's root
    bool rccl = false;                     // the frame is assembled by ncclReduce (else: shared accumulator / on the host)
    std::vector<char> waitRed;             // [n] rccl: the member's next commit writes a snapshot a reduce may still be reading: wait for evRed first
    int W = 0, H = 0;
    bool inited = false;
    bool dirty = false;                    // commits were enqueued since the frame was last assembled
    int k = 0;                             // the snapshot / result buffer the NEXT assembly takes
    int last = -1;                         // ... and the one that holds the latest assembled frame
    int failReduces = 0;                   // tests (pt_test_group_fail_next_reduce): the next reduces are issued with a null communicator
    std::vector<float> stage;              // host gather: one device's frame
    std::string how = "shared accumulator";
};

namespace {
// runs f(i) for every member -- on the members' own threads when the group has them, else on the calling thread, member by member --
// and returns the first failure (its message becomes the calling thread's pt_last_error)
int for_members(PtGroup *g, const std::function<int(int)> &f) {
    const int n = (int)g->ctx.size();
    int first = PT_OK;
    if (!g->worker.empty()) {
        for (int i = 0; i < n; ++i) g->worker[i]->post([&f, i] { return f(i); });
        for (int i = 0; i < n; ++i) {
            std::string e;
            const int rc = g->worker[i]->wait(&e);
            if (rc != PT_OK && first == PT_OK) { first = rc; g_err = e; }
        }
        return first;
    }
    for (int i = 0; i < n; ++i) {
        t_ctx = &g->ctx[i]->st;
        (void)hipSetDevice(g->device[i]);
        const int rc = f(i);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

void group_release_buffers(PtGroup *g) {
    for (GroupDevice &d : g->dev) {
        (void)hipSetDevice(d.device);
        if (d.coll) (void)hipStreamSynchronize(d.coll);
        if (d.full) { (void)hipFree(d.full); d.full = nullptr; }
        for (int q = 0; q < 2; ++q)
            if (d.snap[q]) { (void)hipFree(d.snap[q]); d.snap[q] = nullptr; }
    }
    g->inited = false;
    g->dirty = false;
    g->last = -1;
}

// Assemble the frame from what the members have committed so far -- asynchronously: nothing here waits on the host.
int assemble(PtGroup *g) {
    const int k = g->k;
    if (!g->rccl) {
        // shared accumulator / host gather: the accumulators ARE the frame's parts, and the collective streams -- which the read-back copies
        // on -- already wait for the members' commits (pt_group_iterate_batch)
        g->dirty = false;
        return PT_OK;
    }
    if (!g->dirty && g->last >= 0) return PT_OK;                 // nothing was committed since the last assembly: its frame stands
    // 1. (per device, the collective stream already waits for the commits that wrote snapshot k: every member's last call wrote its rows
    //    there and queued the wait -- pt_group_iterate_batch)
    // 2. ONE ncclReduce(sum) of the snapshots to dev[0].  No early return between GroupStart and GroupEnd: the first error is
    //    kept, the group is always closed, then the call fails -- an open RCCL group would swallow every later collective of the process.
    Rccl &rc_ = rccl();
    int r = rc_.GroupStart(), rHip = 0;
    if (r == 0) {
        for (size_t di = 0; di < g->dev.size(); ++di) {
            GroupDevice &d = g->dev[di];
            if (hipSetDevice(d.device) != hipSuccess) { rHip = 1; continue; }
            void *comm = g->failReduces > 0 ? nullptr : d.comm;
            const int ri = rc_.Reduce(d.snap[k], di == 0 ? d.snap[k] : nullptr, (size_t)g->W * g->H * 3, kNcclFloat, kNcclSum, 0, comm, d.coll);
            if (ri != 0 && r == 0) r = ri;
        }
        const int r2 = rc_.GroupEnd();
        if (r == 0) r = r2;
    }
    if (g->failReduces > 0) --g->failReduces;
    // (an error leaves the snapshot as the commits wrote it and nothing but waits on the collective streams: the next call tries again)
    if (r != 0 || rHip) return fail(PT_ERR_HIP, "pt_group: ncclReduce failed: %s", rHip ? "hipSetDevice" : rc_.GetErrorString(r));
    for (GroupDevice &d : g->dev) {
        HIPCHECK(hipSetDevice(d.device));
        HIPCHECK(hipEventRecord(d.evRed[k], d.coll));
    }
    g->last = k;
    g->k = k ^ 1;
    g->dirty = false;
    std::fill(g->waitRed.begin(), g->waitRed.end(), 1);      // snapshot k ^ 1 was last read by the reduce before this one
    return PT_OK;
}
}  // namespace

namespace {
// the collective's streams, events and RCCL's communicator: once per group, behind the members' first pt_init (see pt_group_create)
int ensure_collective(PtGroup *g) {
    for (GroupDevice &d : g->dev) {
        if (d.coll) continue;
        HIPCHECK(hipSetDevice(d.device));
        HIPCHECK(hipStreamCreateWithFlags(&d.coll, hipStreamNonBlocking));
        for (int q = 0; q < 2; ++q) HIPCHECK(hipEventCreateWithFlags(&d.evRed[q], hipEventDisableTiming));
    }
    if (g->rccl && !g->dev[0].comm) {
        const int nd = (int)g->dev.size();
        std::vector<void *> comms(nd, nullptr);
        std::vector<int> devs;
        for (const GroupDevice &d : g->dev) devs.push_back(d.device);
        std::lock_guard<std::mutex> lock(g_rcclMutex);
        const int r = rccl().CommInitAll(comms.data(), nd, devs.data());
        if (r != 0) {
            fprintf(stderr, "pt_group_init: ncclCommInitAll failed (%s): the frame is assembled %s\n", rccl().GetErrorString(r), nd > 1 ? "on the host" : "in place");
            g->rccl = false;
            g->how = nd > 1 ? "host gather" : "shared accumulator";
        } else {
            for (int q = 0; q < nd; ++q) g->dev[q].comm = comms[q];
        }
    }
    return PT_OK;
}
}  // namespace

int pt_group_create(PtGroup **out, int n, const int32_t *devices) {
    if (!out || n < 1 || n > 64) return fail(PT_ERR_INVALID, "pt_group_create: 1..64 members");
    const int ndev = count_devices();
    if (ndev < 1) return fail(PT_ERR_NO_GPU, "pt_group_create: no HIP device (this library has no CPU fallback)");
    register_exit_handler();
    CurrentGuard guard;
    PtGroup *g = new (std::nothrow) PtGroup();
    if (!g) return fail(PT_ERR_INVALID, "pt_group_create: out of memory");
    for (int i = 0; i < n; ++i) {
        const int d = devices ? devices[i] : i % ndev;
        if (d < 0 || d >= ndev) { delete g; return fail(PT_ERR_INVALID, "pt_group_create: device %d of %d", d, ndev); }
        g->device.push_back(d);
        int di = -1;
        for (size_t q = 0; q < g->dev.size(); ++q)
            if (g->dev[q].device == d) di = (int)q;
        if (di < 0) {
            GroupDevice gd;
            gd.device = d;
            g->dev.push_back(gd);
            di = (int)g->dev.size() - 1;
        }
        g->dev[di].members.push_back(i);
        g->devIndex.push_back(di);
    }
    for (int i = 0; i < n; ++i) {
        PtContext *c = pt_ctx_create();
        if (!c) { pt_group_destroy(g); return PT_ERR_INVALID; }
        g->ctx.push_back(c);
    }
    // Streams.  A process gets FOUR hardware queues per device by default (GPU_MAX_HW_QUEUES), handed to streams as they are created; a
    // fifth stream shares a queue with an earlier one, and work that shares a queue runs in queue order whatever the streams say: commits
    // that land behind a tracing stream's bounce launches wait for whole launches (config C3 as written, one member: 0.046 against 0.040 ms
    // per iteration; 0.053 against 0.0445 with the collective -- profiles/r06_group_experiments.txt).  So: a member that has its device to
    // itself commits on that device's NULL stream (the renderer's other streams are non-blocking: nothing synchronises with it implicitly),
    // and everything the collective needs -- its stream, RCCL's communicator with the streams IT creates -- is made AFTER the members'
    // first pt_init has created their tracing streams (ensure_collective, from pt_group_init): null, two tracing streams, collective.
    // Members that share a device (the one-GPU rehearsal) get a commit stream each.
    g->stream.assign(n, nullptr);
    g->evCommit.assign(n, nullptr);
    bool ok = true;
    const char *ownStreams = getenv("PT_AMD_GROUP_OWN_STREAMS");      // experiments only
    for (int i = 0; i < n && ok; ++i) {
        const bool alone = g->dev[g->devIndex[i]].members.size() == 1 && !(ownStreams && atoi(ownStreams));
        ok = hipSetDevice(g->device[i]) == hipSuccess && (alone || hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking) == hipSuccess) &&
             hipEventCreateWithFlags(&g->evCommit[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        pt_group_destroy(g);
        return fail(PT_ERR_HIP, "pt_group_create: no stream / event on a member's device");
    }
    // RCCL: one rank per DISTINCT device, one process (ncclCommInitAll).  Members that all share one device need no collective
    // (PT_AMD_COLLECTIVE=rccl sends their frame through a one-rank communicator all the same: the call path on a one-GPU box).
    // The library is loaded here -- pt_group_collective() answers from now on --, the communicator is made by the first pt_group_init.
    const char *want = getenv("PT_AMD_COLLECTIVE");
    const bool forbid = want && !strcmp(want, "host");
    const bool force = want && !strcmp(want, "rccl");
    const int nd = (int)g->dev.size();
    if (nd > 1) g->how = "host gather";
    if (!forbid && (nd > 1 || force)) {
        std::lock_guard<std::mutex> lock(g_rcclMutex);
        if (rccl().load()) {
            g->rccl = true;
            g->how = nd > 1 ? "rccl reduce (unmeasured across devices: no multi-GPU node was available to the build)" : "rccl reduce (one-rank communicator)";
        }
    }
    // the members' issuing threads
    const char *thr = getenv("PT_AMD_GROUP_THREADS");
    if (n > 1 && !(thr && atoi(thr) == 0)) {
        for (int i = 0; i < n; ++i) {
            g->worker.emplace_back(new Worker());
            Worker *w = g->worker.back().get();
            State *st = &g->ctx[i]->st;
            const int d = g->device[i];
            w->th = std::thread([w, st, d] {
                (void)hipSetDevice(d);
                w->loop(st);
            });
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_groupMutex);      // (a group the host forgets is destroyed by the process's exit handler)
        g_groups.push_back(g);
    }
    *out = g;
    return PT_OK;
}

void pt_group_destroy(PtGroup *g) {
    if (!g) return;
    CurrentGuard guard;
    // the members' renderers first (they synchronise their own streams), on their own threads' behalf; then the threads
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        t_ctx = &g->ctx[i]->st;
        free_renderer();
    }
    t_ctx = guard.saved;
    for (auto &w : g->worker) w->stop();
    g->worker.clear();
    for (PtContext *c : g->ctx) (void)pt_ctx_destroy(c);
    group_release_buffers(g);
    for (size_t i = 0; i < g->stream.size(); ++i) {
        (void)hipSetDevice(g->device[i]);
        if (g->evCommit[i]) (void)hipEventDestroy(g->evCommit[i]);
        (void)hipStreamSynchronize(g->stream[i]);
        if (g->stream[i]) (void)hipStreamDestroy(g->stream[i]);
    }
    for (GroupDevice &d : g->dev) {
        (void)hipSetDevice(d.device);
        for (int q = 0; q < 2; ++q)
            if (d.evRed[q]) (void)hipEventDestroy(d.evRed[q]);
        if (d.coll) (void)hipStreamDestroy(d.coll);
        if (d.comm) (void)rccl().CommDestroy(d.comm);
    }
    {
        std::lock_guard<std::mutex> lock(g_groupMutex);
        g_groups.erase(std::remove(g_groups.begin(), g_groups.end(), g), g_groups.end());
    }
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
    if (cam->resolution[0] <= 0 || cam->resolution[1] <= 0) return fail(PT_ERR_INVALID, "pt_group_init: bad resolution");
    CurrentGuard guard;
    const int n = (int)g->ctx.size();
    {   // a re-init (the reference's Free -> Init restart): the old renderers go first, then the buffers they accumulate into
        int rc = for_members(g, [](int) -> int { free_renderer(); return PT_OK; });
        if (rc) return rc;
    }
    group_release_buffers(g);
    g->W = cam->resolution[0];
    g->H = cam->resolution[1];
    const size_t frameBytes = (size_t)g->W * (size_t)g->H * 3 * sizeof(float);
    for (GroupDevice &d : g->dev) {
        HIPCHECK(hipSetDevice(d.device));
        HIPCHECK(hipMalloc(&d.full, frameBytes));
        HIPCHECK(hipMemset(d.full, 0, frameBytes));
        if (g->rccl)
            for (int q = 0; q < 2; ++q) {       // (zeroed: the rows of the OTHER devices' members stay zero, which is what the sum needs)
                HIPCHECK(hipMalloc(&d.snap[q], frameBytes));
                HIPCHECK(hipMemset(d.snap[q], 0, frameBytes));
            }
        HIPCHECK(hipDeviceSynchronize());
    }
    g->waitRed.assign(n, 0);
    PtOptions base;
    memset(&base, 0, sizeof base);
    if (opts) base = *opts;
    int rc = for_members(g, [&](int i) -> int {
        PtOptions o = base;
        o.shard_rank = i;
        o.shard_count = n;
        o.device = g->device[i];
        o.stream = g->stream[i];
        o.flags &= ~PT_FLAG_ACCUM_SHARD_ROWS;
        o.accum_dev = g->dev[g->devIndex[i]].full;          // own rows in place, the co-members' beside them, zeros for the other devices' rows
        return pt_init(cam, geoms, ngeoms, mats, nmats, traceDepth, &o);
    });
    if (rc) return rc;
    if ((rc = ensure_collective(g))) return rc;      // (behind the members' tracing streams: see pt_group_create)
    g->inited = true;
    g->dirty = false;
    g->k = 0;
    g->last = -1;
    return PT_OK;
}

// every member enqueues its shard's wavefront batch (asynchronous: the devices run side by side, fed by a host thread each)
int pt_group_iterate_batch(PtGroup *g, int frame, int first_iter, int count) {
    if (!g) return fail(PT_ERR_INVALID, "pt_group_iterate_batch: null group");
    if (!g->inited) return fail(PT_ERR_NOT_INIT, "pt_group_iterate_batch before pt_group_init");
    CurrentGuard guard;
    int rc = for_members(g, [&](int i) -> int {
        if (g->rccl) {      // this call's commit copies the member's rows into snapshot k as well
            GroupDevice &d = g->dev[g->devIndex[i]];
            R().snapTarget = d.snap[g->k];
            R().snapWait = g->waitRed[i] ? d.evRed[g->k] : nullptr;
            g->waitRed[i] = 0;
        }
        int r = pt_iterate_batch(frame, first_iter, count, nullptr);
        if (r) return r;
        // the member's commits so far, and the device's collective stream behind them (whatever reads the frame next -- a reduce, a read-back --
        // is enqueued there after every member is through with this: the members' threads do it side by side)
        HIPCHECK(hipEventRecord(g->evCommit[i], g->stream[i]));
        HIPCHECK(hipStreamWaitEvent(g->dev[g->devIndex[i]].coll, g->evCommit[i], 0));
        return PT_OK;
    });
    if (rc) return rc;
    g->dirty = true;
    return PT_OK;
}

// config C3 as written: one iteration on every member (a commit out of the batches traced ahead with PT_FLAG_TRACE_AHEAD), then the
// frame's assembly -- the reference's per-iteration full-frame transfer (src/pathtrace.cu:170-171) with the reduce in its place
int pt_group_iterate(PtGroup *g, int frame, int iter) {
    int rc = pt_group_iterate_batch(g, frame, iter, 1);
    if (rc) return rc;
    return pt_group_reduce(g);
}

int pt_group_reduce(PtGroup *g) {
    if (!g) return fail(PT_ERR_INVALID, "pt_group_reduce: null group");
    if (!g->inited) return fail(PT_ERR_NOT_INIT, "pt_group_reduce before pt_group_init");
    CurrentGuard guard;
    return assemble(g);
}

int pt_group_sync(PtGroup *g) {
    if (!g) return fail(PT_ERR_INVALID, "pt_group_sync: null group");
    if (!g->inited) return fail(PT_ERR_NOT_INIT, "pt_group_sync before pt_group_init");
    CurrentGuard guard;
    int rc = for_members(g, [](int) -> int { return pt_sync(); });      // (reports a member's device fault)
    if (rc) return rc;
    for (GroupDevice &d : g->dev) {
        HIPCHECK(hipSetDevice(d.device));
        HIPCHECK(hipStreamSynchronize(d.coll));
    }
    return PT_OK;
}

// the whole frame's un-normalised running sum (W * H * 3 floats), as pt_readback delivers it for one device.  Waits for the frame's
// assembly and the copy only -- not for the batches the members trace ahead.
int pt_group_readback(PtGroup *g, float *rgb_sum_host) {
    if (!g || !rgb_sum_host) return fail(PT_ERR_INVALID, "pt_group_readback: null argument");
    if (!g->inited) return fail(PT_ERR_NOT_INIT, "pt_group_readback before pt_group_init");
    CurrentGuard guard;
    const size_t frameFloats = (size_t)g->W * g->H * 3;
    int rc;
    if (g->dirty || (g->rccl && g->last < 0)) {
        if ((rc = assemble(g))) return rc;
    }
    if (g->rccl) {
        HIPCHECK(hipSetDevice(g->dev[0].device));
        HIPCHECK(hipMemcpyAsync(rgb_sum_host, g->dev[0].snap[g->last], frameFloats * sizeof(float), hipMemcpyDeviceToHost, g->dev[0].coll));
        HIPCHECK(hipStreamSynchronize(g->dev[0].coll));
    } else if (g->dev.size() == 1) {
        HIPCHECK(hipSetDevice(g->dev[0].device));
        HIPCHECK(hipMemcpyAsync(rgb_sum_host, g->dev[0].full, frameFloats * sizeof(float), hipMemcpyDeviceToHost, g->dev[0].coll));
        HIPCHECK(hipStreamSynchronize(g->dev[0].coll));
    } else {
        // host gather: row y belongs to member y % n, i.e. to that member's device
        g->stage.resize(frameFloats);
        const size_t rowFloats = (size_t)g->W * 3;
        const int n = (int)g->ctx.size();
        for (size_t di = 0; di < g->dev.size(); ++di) {
            GroupDevice &d = g->dev[di];
            HIPCHECK(hipSetDevice(d.device));
            HIPCHECK(hipMemcpyAsync(g->stage.data(), d.full, frameFloats * sizeof(float), hipMemcpyDeviceToHost, d.coll));
            HIPCHECK(hipStreamSynchronize(d.coll));
            for (int y = 0; y < g->H; ++y)
                if (g->devIndex[y % n] == (int)di) memcpy(rgb_sum_host + (size_t)y * rowFloats, g->stage.data() + (size_t)y * rowFloats, rowFloats * sizeof(float));
        }
    }
    // a member's device fault: the kernels write the sticky fault word to page-locked host memory too (readback_fault), so the good path
    // costs no copy and no wait for the members' streams
    for (PtContext *c : g->ctx) {
        t_ctx = &c->st;
        if (R().hostFault && *(volatile uint32_t *)R().hostFault != 0u) {
            (void)hipSetDevice(R().device);
            if ((rc = readback_fault())) return rc;
        }
    }
    return PT_OK;
}

// the members' tallies, summed (every member counts the paths of its own rows)
int pt_group_counters(PtGroup *g, PtCounters *out) {
    if (!g || !out) return fail(PT_ERR_INVALID, "pt_group_counters: null argument");
    if (!g->inited) return fail(PT_ERR_NOT_INIT, "pt_group_counters before pt_group_init");
    CurrentGuard guard;
    memset(out, 0, sizeof *out);
    std::vector<PtCounters> cs(g->ctx.size());
    int rc = for_members(g, [&](int i) -> int { return pt_counters(&cs[i]); });
    if (rc) return rc;
    for (size_t i = 0; i < cs.size(); ++i) {
        const PtCounters &c = cs[i];
        for (int d = 0; d < PT_MAX_DEPTH + 2; ++d) { out->live[d] += c.live[d]; out->ended_early[d] += c.ended_early[d]; }
        out->light_hits += c.light_hits;
        out->misses += c.misses;
        out->iterations = i == 0 ? c.iterations : std::min(out->iterations, c.iterations);
        out->bounce_launches += c.bounce_launches;
        out->bounce_kernel_ms += c.bounce_kernel_ms;
    }
    return PT_OK;
}
