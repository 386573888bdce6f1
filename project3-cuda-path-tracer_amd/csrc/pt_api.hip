// pt_api.hip -- MI355X (gfx950 / CDNA4) path-tracing hot path + its C ABI (include/pt_amd.h).
//
// One iteration = `traceDepth` launches of ONE fused persistent kernel per bounce (the first builds its camera rays in
// registers): nearest-hit over the scene (geometry through the scalar path, materials in LDS) -> shade/scatter -> park
// emitter radiance -> stream compaction of the survivors straight into the next bounce's SoA path pool (wave64
// ballot/mbcnt ranks, LDS wave totals = workgroup-level exclusive scan per class; the tile's output runs are reserved
// with ONE atomic instruction on sharded position counters, chunks of the pool are handed out on demand).  No host
// round trip inside an iteration: live counts stay on the device.  The multi-workgroup ORDERED scan (two-level decoupled
// look-back) is the stream-compaction library in pt_compaction.h (pt_scan_exclusive_i32 / pt_compact_nonzero_i32).
// Files: pt_device.h (math), pt_trace.h (render kernels), pt_compaction.h, pt_test_kernels.h (primitives for the parity
// tests), this file (host side + C ABI).
//
// Replaces the unsolved pipeline of reference src/pathtrace.cu:133-167 (spec: SURVEY.md 3.4 S0-S9).
// HBM layout, kernels, rooflines: DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <mutex>

#include "../../include/pt_amd.h"
#include "pt_compaction.h"
#include "pt_trace.h"
#include "pt_mesh_walk.h"
#include "pt_mesh.h"
// The entry points of include/pt_amd_test.h (device primitives one by one, soundness sweeps, the fault-word hook) exist only in
// the second link target of this source, libpt_amd_test.so (-DPT_TEST_API): the product library exports none of them.
#ifdef PT_TEST_API
#include "../../include/pt_amd_test.h"
#include "pt_test_kernels.h"
#endif

using namespace ptd;
using namespace ptk;

static_assert(sizeof(PtGeom) == 236 && sizeof(PtMaterial) == 44 && sizeof(PtCamera) == 52,
              "layout must equal reference src/sceneStructs.h:18-47");
namespace {

// =====================================================================================================
// host side
// =====================================================================================================
thread_local std::string g_err = "";     // pt_last_error(): the calling thread's last failure

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(PT_ERR_HIP, "HIP error (%s:%d): %s: %s", "pt_api.hip", __LINE__, #expr, \
                        hipGetErrorString(e_));                                                     \
    } while (0)

constexpr int kMaxSlots = 4;

// One in-flight iteration: its own stream, path buffers, counters and deferred-radiance buffer.
struct Slot {
    hipStream_t stream = nullptr;
    float *pathbuf[2] = {nullptr, nullptr};      // two path pools (ping-pong by bounce), poolChunks * kChunk paths per array
    unsigned long long *chunkList[2] = {nullptr, nullptr}; // per pool: [kSeg][poolChunks] chunk lists of the queue it holds
    uint32_t gen[2] = {0, 0};      // per pool: serial number of the launch that filled it (tags the chunk-list entries)
    Ctrl *ctrl = nullptr;
    float *contrib = nullptr;      // maxBatch x W*H*3: an entry is valid while its bit of `hitMask` is set
    uint32_t *hitMask = nullptr;   // ceil(maxBatch / 32) x W*H: which iterations of the batch wrote a pixel's `contrib`, zero between batches
    unsigned long long *meshHit = nullptr;   // scenes with meshes: the walks' result per path of the bounce being launched (BounceArgs::meshHit)
    hipEvent_t evDone = nullptr;       // all bounce launches of the slot's current iteration finished
    hipEvent_t evCommitted = nullptr;  // k_commit consumed `contrib` (cleared the masks' bits)
    int parity = 0;                // which half of Ctrl::cursor the slot's next batch uses
};

struct State {
    bool init = false;
    int device = 0;
    hipStream_t stream = nullptr;   // the caller's stream: commits, PBO conversion, readback
    PtCamera cam;
    KParams prm;
    int P = 0;              // W*H
    int nLocal = 0;
    int flags = 0;
    float *image = nullptr;
    bool ownImage = false;
    int nslots = 0;
    int maxBatch = 1;       // iterations that may share one wavefront (pt_iterate_batch)
    Slot slot[kMaxSlots];
    GeomDev *dgeoms = nullptr;
    float4 *dGeomHit = nullptr;     // GeomHitDev[ngeoms], as the kernels stage it in LDS (scenes that are not sphere-heavy)
    float *dRows = nullptr;         // sphere-heavy scenes of hundreds of primitives: the matrix rows that do not go to LDS (BounceArgs::rows)
    MaterialDev *dmats = nullptr;
    WallBox *dwalls = nullptr;
    SphereCull *dSphCull = nullptr; // sphere-heavy scenes: packed culling data of the spheres, and ...
    SphereCull *dSphGroups = nullptr;   // ... scenes of hundreds of them: the bounding balls of the table's groups (BounceArgs::sphGroups)
    bool grouped = false;           // ... whose later bounces take the k_bounce<..., GROUPS> instantiations
    int *dRowOff = nullptr, *dRowIdx = nullptr;   // camera-ray bounce: per image row, the primitives whose pixel rectangle covers it
    int *dClassIdx = nullptr;       // later bounces: per queue class, the primitives to look at (KParams::classOff)
    float4 *dMeshRecs = nullptr;    // ptd::MeshUnit[]: triangles and inner nodes of every mesh of the scene (k_bounce<., ., ., true>)
    bool mesh = false;      // the scene holds triangle meshes: the k_bounce<., false, ., true> variants
    // ... whose walks run ahead of every bounce launch (k_mesh_walk): the meshes alone per queue class / in all / per image row
    int *dWalkIdx = nullptr, *dWalkRowOff = nullptr;
    float4 *dWalkMeshRows = nullptr;   // ptk::WalkMesh per mesh (BounceArgs::walkMeshRows); walkMeshLds: how many of them a workgroup stages in LDS (all, or none)
    int walkMeshLds = 0;
    int walkClassOff[kClsMax + 1] = {0}, walkAll0 = 0, walkAll1 = 0;
    int gridWalk = 0, gridWalkFirst = 0;
    size_t ldsWalk = 0;
    int numTilesMax = 0;    // upper bound of tiles in one bounce queue (incl. one partial tile per segment)
    int poolChunks = 0;     // chunks per path pool (incl. the trash chunk 0); a pool holds poolChunks * kChunk paths per array
    int grid = 0;           // persistent grid of k_bounce<false>
    int gridFirst = 0;      // ... and of k_bounce<true> (its own register budget, hence its own residency)
    bool many = false;      // more than kBinMax small primitives (spheres, cubes that are neither walls nor binned): the k_bounce<., true> variants
    bool sweptCubes = false;  // ... some of them cubes (TileArgs::hot: kHotSweptCubes)
    bool dof = false;       // thin-lens camera: the k_bounce<true, ., true> variants for the camera-ray bounce
    bool plain = false;     // no refractive material, no specular exponent on a reflective one, no direct lighting: k_bounce<..., PLAIN>
    size_t ldsBytes = 0, ldsBytesNext = 0;   // dynamic LDS of the camera-ray launch / of the later ones
    long long iterations = 0;
    long long seq = 0;      // batches enqueued since pt_init: slot = seq % nslots
    // PT_FLAG_TRACE_AHEAD: batches traced ahead of the pt_iterate calls that will ask for their iterations, oldest first.
    // A parked batch occupies its slot (radiance buffers, iteration masks) until its last iteration is committed or it
    // is discarded; the parked slots are the `ahead.size()` slots before seq % nslots in the rotation.
    struct Parked { int slot, first, count, next; bool waited; };   // iterations first .. first + count - 1, the next one to commit = first + next;
                                                                    // waited: the caller's stream already waits for the batch's launches
    std::deque<Parked> ahead;
    uint32_t launchSerial = 0;   // bounce launches since pt_init
    // host buffer of pt_readback, page-locked on first use so the per-iteration D2H copy of the reference protocol
    // (src/pathtrace.cu:170-171) runs at PCIe rate instead of through a pageable staging copy
    void *pinnedHost = nullptr;
    size_t pinnedBytes = 0;
    // kernel timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evBounce;
    std::vector<hipEvent_t> evFree;   // resolved timing events, reused (creating two per launch cost 1 % of a timed step)
    double msBounce = 0;
    long long nBounce = 0;
    // The sticky fault word once more, in page-locked HOST memory the kernels can write (a fault path stores it there too): pt_readback
    // and pt_readback_rgba8, which synchronise anyway, then report a faulted render without a device-to-host copy of their own.
    uint32_t *hostFault = nullptr;       // host address
    uint32_t *hostFaultDev = nullptr;    // the same word as the kernels address it
    // device groups (pt_group.h): the full frame the NEXT commit also copies this shard's pixels into (k_commit's `snap`: the snapshot a
    // frame reduce reads), and the event -- the reduce that last read that buffer -- the commit has to wait for first.  Set per call.
    float *snapTarget = nullptr;
    hipEvent_t snapWait = nullptr;
    // triangle soups registered by pt_set_meshes, consumed by the next pt_init (kept across pt_free: the reference's
    // Free -> Init restart protocol re-initialises the same scene)
    std::vector<ptm::HostMesh> meshes;
};

// Renderer instances.  The reference keeps its renderer in file-static globals (src/pathtrace.cu:70-71: one per process, not
// re-entrant); so did rounds 1-4 here.  Now a renderer is a CONTEXT: the C ABI's functions act on the calling thread's CURRENT
// context -- the process-wide default one unless pt_ctx_make_current named another (include/pt_amd.h) -- so that one host process
// drives several renderers: one per device of a node (row shards of one frame, pt_group_*), or several on one device.
State g_default;
thread_local State *t_ctx = &g_default;
inline State &R() { return *t_ctx; }
std::mutex g_ctxMutex;                    // guards g_contexts
std::vector<State *> g_contexts;          // every context pt_ctx_create made and pt_ctx_destroy has not released (the exit handler frees their renderers)
std::mutex g_groupMutex;                  // guards g_groups
std::vector<PtGroup *> g_groups;          // every group pt_group_create made and pt_group_destroy has not released (the exit handler destroys them:
                                          // their issuing threads, communicators, streams and frame buffers -- pt_group.h)

const ptm::HostMesh *mesh_of(int geom) {
    for (const ptm::HostMesh &m : R().meshes)
        if (m.geom == geom) return &m;
    return nullptr;
}

int count_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int reset_ctrl(Ctrl *dev, hipStream_t st) {
    HIPCHECK(hipMemsetAsync(dev, 0, sizeof(Ctrl), st));
    HIPCHECK(hipStreamSynchronize(st));
    return PT_OK;
}

PathPool pool(const Slot &sl, int which) {
    PathPool p;
    p.base = sl.pathbuf[which];
    p.list = sl.chunkList[which];
    p.cap = (uint32_t)R().poolChunks << R().prm.chunkShift;
    return p;
}

#include "pt_host_scene.h"

int resolve_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &ms, long long &n) {
    for (auto &pr : v) {
        float t = 0;
        HIPCHECK(hipEventSynchronize(pr.second));
        HIPCHECK(hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        n += 1;
        R().evFree.push_back(pr.first);
        R().evFree.push_back(pr.second);
    }
    v.clear();
    return PT_OK;
}

// The instantiation of k_bounce a launch takes: FIRST (camera rays), MANY (per-lane sphere lists: scenes with more than
// kBinMax spheres), DOF (thin lens: the camera-ray launch only), MESH (scenes with triangle meshes).
template <bool F, bool M, bool D, bool ME, bool PL = false, bool CU = false, bool GR = false>
const void *kb() { return reinterpret_cast<const void *>(k_bounce<F, M, D, ME, PL, CU, GR>); }
const void *bounce_kernel(bool first, bool dof) {
    // (plain scenes -- diffuse / emissive / perfect-mirror materials, no README extra: the instantiations without the rarer branches)
    if (R().plain && !R().mesh && !R().many && !dof) return first ? kb<true, false, false, false, true>() : kb<false, false, false, false, true>();
    if (R().grouped && !first)            // hundreds of swept primitives: the two-level sweep, nothing of the scene's tables in LDS
        return R().sweptCubes ? kb<false, true, false, false, false, true, true>() : kb<false, true, false, false, false, false, true>();
    if (R().grouped && !dof)              // ... and their (pinhole) camera-ray bounce: hit records, frames and rows from global memory too
        return R().sweptCubes ? kb<true, true, false, false, false, true, true>() : kb<true, true, false, false, false, false, true>();
    if (R().many && R().sweptCubes) {     // many small primitives, cubes among them: the per-lane tests take either type
        if (R().mesh) return first ? (dof ? kb<true, true, true, true, false, true>() : kb<true, true, false, true, false, true>()) : kb<false, true, false, true, false, true>();
        return first ? (dof ? kb<true, true, true, false, false, true>() : kb<true, true, false, false, false, true>()) : kb<false, true, false, false, false, true>();
    }
    if (R().mesh && R().many) return first ? (dof ? kb<true, true, true, true>() : kb<true, true, false, true>()) : kb<false, true, false, true>();
    if (R().mesh) return first ? (dof ? kb<true, false, true, true>() : kb<true, false, false, true>()) : kb<false, false, false, true>();
    if (first && dof) return R().many ? kb<true, true, true, false>() : kb<true, false, true, false>();
    if (first) return R().many ? kb<true, true, false, false>() : kb<true, false, false, false>();
    return R().many ? kb<false, true, false, false>() : kb<false, false, false, false>();
}

const void *walk_kernel(bool first, bool dof) {
    if (R().walkMeshLds != 0) {      // the meshes' rows in LDS (scenes of at most kWalkMeshLdsMax meshes)
        if (first) return dof ? reinterpret_cast<const void *>(k_mesh_walk<true, true, true>) : reinterpret_cast<const void *>(k_mesh_walk<true, false, true>);
        return reinterpret_cast<const void *>(k_mesh_walk<false, false, true>);
    }
    if (first) return dof ? reinterpret_cast<const void *>(k_mesh_walk<true, true, false>) : reinterpret_cast<const void *>(k_mesh_walk<true, false, false>);
    return reinterpret_cast<const void *>(k_mesh_walk<false, false, false>);
}

int launch_bounce(Slot &sl, int iter, int batch, int depth, bool lastBounce, float *contrib, bool nextIsLast = false) {
    const PathPool in = pool(sl, (depth - 1) & 1);
    const PathPool out = pool(sl, depth & 1);
    // chunk-list entries carry the serial number of the launch that wrote them (never 0)
    const uint32_t genIn = sl.gen[(depth - 1) & 1];
    if (++R().launchSerial == 0u) ++R().launchSerial;
    const uint32_t genOut = sl.gen[depth & 1] = R().launchSerial;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (R().flags & PT_FLAG_KERNEL_TIMING) {
        for (hipEvent_t *e : {&e0, &e1}) {
            if (!R().evFree.empty()) { *e = R().evFree.back(); R().evFree.pop_back(); }
            else HIPCHECK(hipEventCreate(e));
        }
        HIPCHECK(hipEventRecord(e0, sl.stream));
    }
    BounceArgs ba;
    ba.prm = R().prm;
    ba.iter = iter; ba.batch = batch; ba.depth = depth; ba.lastBounce = lastBounce ? 1 : 0; ba.parity = sl.parity;
    ba.genIn = genIn; ba.genOut = genOut;
    ba.in = in; ba.out = out;
    ba.tile.inBase = in.base; ba.tile.inList = in.list; ba.tile.inCap = in.cap;
    ba.tile.genIn = genIn; ba.tile.poolChunks = (uint32_t)R().prm.poolChunks; ba.tile.chunkShift = (uint32_t)R().prm.chunkShift;
    ba.tile.skipNonCandidates = (lastBounce && R().prm.emittersBinned) ? 1u : 0u;
    ba.tile.hot = (lastBounce ? kHotLast : 0u) | (R().prm.allClassified ? kHotAllClassified : 0u) | (contrib ? kHotContrib : 0u) |
                  ((R().prm.directDepth != 0 && depth == R().prm.directDepth && R().prm.nEmit > 0) ? kHotToLight : 0u) |
                  (R().prm.contribLocal ? kHotContribLocal : 0u) | ((R().flags & PT_FLAG_MIXTURE_WEIGHTED) ? kHotMixWeighted : 0u) | ((uint32_t)R().prm.nWalls << 8) | ((uint32_t)R().prm.nSlotWalls << 11) | ((uint32_t)R().prm.nPlaneWalls << 17) |
                  ((uint32_t)R().prm.nBinned << 14) | ((uint32_t)R().prm.nmats << 20);
    // sphere clusters: the queue that ENTERS the last bounce carries other candidate bits (k_bounce: kHotWritesLastBits) -- that launch only asks
    // whether a path ends on an emitter, and with every emitter binned it visits the tiles of the binned primitives' candidates alone
    const bool lastBits = R().many && !R().mesh && R().prm.sphOMax > 0.0f && R().prm.emittersBinned;
    if (lastBits && nextIsLast) ba.tile.hot |= kHotWritesLastBits;
    if (lastBits && lastBounce && depth > 1) ba.tile.hot |= kHotReadsLastBits;
    ba.ctrl = sl.ctrl; ba.ggeoms = R().dgeoms; ba.gmats = R().dmats; ba.ghit = R().dGeomHit; ba.contrib = contrib; ba.hitMask = sl.hitMask;
    ba.sphCull = R().dSphCull; ba.classIdx = R().dClassIdx;
    ba.rowOff = R().dRowOff; ba.rowIdx = R().dRowIdx;
    ba.walls = R().dwalls;
    ba.meshRecs = R().dMeshRecs;
    ba.hostFault = R().hostFaultDev;
    ba.rows = reinterpret_cast<const float4 *>(R().dRows);
    ba.meshHit = sl.meshHit; ba.walkIdx = R().dWalkIdx; ba.walkRowOff = R().dWalkRowOff;
    memcpy(ba.walkClassOff, R().walkClassOff, sizeof ba.walkClassOff);
    ba.walkAll0 = R().walkAll0; ba.walkAll1 = R().walkAll1;
    ba.walkMeshRows = R().dWalkMeshRows; ba.walkMeshLds = R().walkMeshLds;
    ba.sphGroups = R().dSphGroups;
    void *kargs[] = {&ba};
    const bool first = depth == 1;
    // scenes with meshes: the walks of this bounce's rays, ahead of it (pt_mesh_walk.h)
    if (R().mesh)
        HIPCHECK(hipLaunchKernel(walk_kernel(first, first && R().dof), dim3(first ? R().gridWalkFirst : R().gridWalk), dim3(kBlock), kargs, R().ldsWalk, sl.stream));
    HIPCHECK(hipLaunchKernel(bounce_kernel(first, first && R().dof), dim3(first ? R().gridFirst : R().grid), dim3(kBlock), kargs, first ? R().ldsBytes : R().ldsBytesNext, sl.stream));
    if (e0) {
        HIPCHECK(hipEventRecord(e1, sl.stream));
        R().evBounce.emplace_back(e0, e1);
        if (R().evBounce.size() > 8192) {
            int rc = resolve_events(R().evBounce, R().msBounce, R().nBounce);
            if (rc) return rc;
        }
    }
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

// scan library workspace: the chunk totals, one buffer per stream (calls on different streams may overlap; calls on one
// stream are ordered, so they share it)
struct ScanWs {
    int device = -1;                 // the device the buffers live on
    uint32_t *partial = nullptr;     // [kScanChunksMax + 1]
    unsigned long long *chained = nullptr;   // k_scan_chained: {ticket, sums[kScanChunksMax]}
    uint32_t gen = 0;                // calls of the chained scan on this stream (tags the sums; never 0)
};
std::map<std::pair<int, hipStream_t>, ScanWs> g_scan;   // per (device, stream): the NULL stream -- or an equal handle -- of another device is another workspace
std::mutex g_scanMutex;              // the scan library may be called from several host threads (one stream each)

// The caller HOLDS g_scanMutex from here until its launches that use the workspace are enqueued: pt_free, which releases the
// workspaces, takes the same lock first and then waits for the device -- so a workspace is never freed between its look-up and the
// kernels that use it (round 3 handed the pointer out of the lock: a second host thread could enqueue on freed memory).
int scan_ws(hipStream_t st, ScanWs **out) {
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));    // the device current at the call: where the caller's buffers live and the kernels will run
    ScanWs &W = g_scan[std::make_pair(dev, st)];          // (std::map: the reference stays valid while other streams are added)
    W.device = dev;
    if (!W.partial) HIPCHECK(hipMalloc(&W.partial, (size_t)(kScanChunksMax + 1) * sizeof(uint32_t)));
    if (!W.chained) {
        HIPCHECK(hipMalloc(&W.chained, (size_t)(kScanChunksMax + 1) * sizeof(unsigned long long)));
        HIPCHECK(hipMemset(W.chained, 0, (size_t)(kScanChunksMax + 1) * sizeof(unsigned long long)));
    }
    *out = &W;
    return PT_OK;
}
// releases every stream's workspace: the PUBLIC pt_free and the process's exit only -- not the pt_free inside pt_init (the
// reference's Free -> Init restart), which other host threads' scans must survive.  Under the lock: no scan call is between its
// look-up and its launches; then everything enqueued is waited for, then freed.
void scan_release() {
    std::lock_guard<std::mutex> lock(g_scanMutex);
    if (g_scan.empty()) return;
    int cur = -1, synced = -1;
    (void)hipGetDevice(&cur);
    for (auto &kv : g_scan) {        // (ordered by device: every device that owns a workspace is waited for once, then its buffers go)
        if (kv.second.device != synced) {
            (void)hipSetDevice(kv.second.device);
            (void)hipDeviceSynchronize();
            synced = kv.second.device;
        }
        if (kv.second.partial) (void)hipFree(kv.second.partial);
        if (kv.second.chained) (void)hipFree(kv.second.chained);
    }
    g_scan.clear();
    if (cur >= 0) (void)hipSetDevice(cur);
}
bool scan_in_use() {
    std::lock_guard<std::mutex> lock(g_scanMutex);
    return !g_scan.empty();
}
// the array as chunks of whole tiles: at most kScanChunksMax of them
void scan_chunks(long long n, long long *tilesPerChunk, int *chunks) {
    const long long tiles = (n + kScanTile - 1) / kScanTile;
    const long long per = (tiles + kScanChunksMax - 1) / kScanChunksMax;
    *tilesPerChunk = per < 1 ? 1 : per;
    *chunks = (int)((tiles + *tilesPerChunk - 1) / *tilesPerChunk);
}

// wait for every stream the renderer uses
int sync_all() {
    for (int i = 0; i < R().nslots; ++i) HIPCHECK(hipStreamSynchronize(R().slot[i].stream));
    HIPCHECK(hipStreamSynchronize(R().stream));
    return PT_OK;
}

int check_device_fault() {
    int rc = sync_all();
    if (rc) return rc;
    for (int i = 0; i < R().nslots; ++i) {
        uint32_t err = 0;
        HIPCHECK(hipMemcpy(&err, &R().slot[i].ctrl->error, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(PT_ERR_DEVICE, "device fault 0x%x:%s%s (results of this render are void; re-init)", err,
                             (err & kFaultPoolExhausted) ? " path pool exhausted" : "", (err & kFaultReserveTimeout) ? " chunk reservation timed out" : "");
    }
    return PT_OK;
}

// pt_readback / pt_readback_rgba8 have just synchronised the caller's stream: every launch whose radiance the image holds has
// finished, and a fault any of them raised is in the host-visible copy of the fault word -- no device-to-host copy on the
// good path.  A faulted render is then reported like pt_sync does (the image has been copied all the same; it is void).
int readback_fault() {
    if (R().hostFault && *(volatile uint32_t *)R().hostFault != 0u) {
        int rc = check_device_fault();
        if (rc) return rc;
        return fail(PT_ERR_DEVICE, "device fault (results of this render are void; re-init)");
    }
    return PT_OK;
}

int persistent_grid(const void *kernel, size_t lds, int *grid) {
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, dev));
    int perCU = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, kBlock, lds));
    if (perCU < 1) perCU = 1;
    // All four instantiations fit eight workgroups per CU (<= 80 SGPRs, <= 64 VGPRs; the sphere-list variants seven).  A
    // launch that has the GPU to itself (pipeline_depth 1) is fastest with all of them (0.253 ms against 0.274 with six);
    // with batches in flight on neighbouring streams six per launch is better (131.2 G paths/s against 128.9 with eight):
    // the two free wave slots per SIMD go to the neighbouring batch's launches, which fill this launch's tail.
    int cap = R().nslots > 1 ? 6 : 8;
    // PT_FLAG_TRACE_AHEAD (one call and one small commit per iteration while batches are traced ahead): four, so that the commit -- and the
    // copies and collectives a caller puts behind it every iteration -- find free wave slots next to two batches' persistent workgroups
    // instead of waiting for one of them to end (config C3 as written: 0.0538 -> 0.0492 ms per iteration, profiles/r05_c3_experiments.txt)
    if ((R().flags & PT_FLAG_TRACE_AHEAD) && R().nslots > 1) cap = 4;
    if (const char *e = getenv("PT_AMD_BLOCKS_PER_CU")) cap = atoi(e);   // experiments only
    if (perCU > cap) perCU = cap;
    *grid = prop.multiProcessorCount * perCU;
    if (const char *e = getenv("PT_AMD_MAX_GRID")) *grid = std::max(1, std::min(*grid, atoi(e)));   // tests: a small or partitioned device's grid
    return PT_OK;
}

constexpr int kIterEnd = 1 << 22;     // iterations are 1 .. kIterEnd - 1 (seed bits, pathtrace.cu:43)

// the bounce launches of iterations first_iter .. first_iter + count - 1 as one wavefront batch on the slot's stream; the
// radiance they find is parked in the slot's buffers until a commit consumes it
int trace_batch(Slot &sl, int first_iter, int count) {
    // the slot's radiance buffers must have been consumed by the commit of its previous batch
    HIPCHECK(hipStreamWaitEvent(sl.stream, sl.evCommitted, 0));
    const int D = R().prm.traceDepth;
    for (int d = 1; d <= D; ++d) {
        int rc = launch_bounce(sl, first_iter, count, d, d == D, sl.contrib, d + 1 == D);
        if (rc) {
            // a launch failed with part of the batch enqueued: counters, parity and radiance buffers are half-updated, so the
            // renderer refuses further work until it is re-initialised (pt_free still releases everything)
            if (d > 1) R().init = false;
            return rc;
        }
    }
    sl.parity ^= 1;   // the last launch re-armed the other half of the slot's counters
    HIPCHECK(hipEventRecord(sl.evDone, sl.stream));
    return PT_OK;
}

// iterations [b0, b1) of the slot's batch of `count` into the accumulator (or nowhere: `discard`), on the caller's stream
int commit_range(Slot &sl, int count, int b0, int b1, bool discard) {
    if (R().nLocal > 0) {
        float *snap = discard ? nullptr : R().snapTarget;
        if (snap && R().snapWait) {
            HIPCHECK(hipStreamWaitEvent(R().stream, R().snapWait, 0));
            R().snapWait = nullptr;
        }
        const int compact = (R().flags & PT_FLAG_ACCUM_SHARD_ROWS) ? 1 : 0;
        if (b1 == b0 + 1)      // one iteration of the batch (the reference's protocol over a batch traced ahead): the light kernel
            hipLaunchKernelGGL(k_commit_one, dim3((R().nLocal + kBlock - 1) / kBlock), dim3(kBlock), 0, R().stream, R().prm, R().image, sl.contrib, sl.hitMask,
                               compact, b0, discard ? 1 : 0, snap);
        else
            hipLaunchKernelGGL(k_commit, dim3((R().nLocal + kBlock - 1) / kBlock), dim3(kBlock), 0, R().stream, R().prm, R().image, sl.contrib, sl.hitMask,
                               count, compact, b0, b1, discard ? 1 : 0, snap);
        HIPCHECK(hipGetLastError());
    }
    return PT_OK;
}

// PT_FLAG_TRACE_AHEAD: trace the batch that starts at iteration `first` into the next slot of the rotation and park it
int trace_ahead(int first) {
    const int count = std::min(R().maxBatch, kIterEnd - first);
    const int slot = (int)(R().seq % R().nslots);
    int rc = trace_batch(R().slot[slot], first, count);
    if (rc) return rc;
    R().ahead.push_back({slot, first, count, 0, false});
    R().seq += 1;
    return PT_OK;
}

// ... and drop what is parked: the caller asked for something else.  The iterations not yet committed are consumed without
// being added, which leaves the slots' buffers zeroed as every batch expects to find them.
int discard_ahead() {
    for (const State::Parked &p : R().ahead) {
        Slot &sl = R().slot[p.slot];
        HIPCHECK(hipStreamWaitEvent(R().stream, sl.evDone, 0));
        int rc = commit_range(sl, p.count, p.next, p.count, true);
        if (rc) return rc;
        HIPCHECK(hipEventRecord(sl.evCommitted, R().stream));
    }
    R().ahead.clear();
    return PT_OK;
}

// Process exit with work in flight (an exception in the host between pt_iterate and pt_free, an interpreter that is torn down
// with a live renderer): the streams are drained and everything is released BEFORE the HIP runtime's own exit handlers run --
// this handler is registered after the library's first HIP call, and exit handlers run in reverse order of registration --
// so that no launch of this process is still executing when its queues, its code object and its memory go away.
extern "C" void pt_free(void);
extern "C" void pt_group_destroy(PtGroup *g);
void free_renderer();
void exit_handler() {
    for (;;) {                                    // groups the host forgot: their members' renderers, threads, RCCL communicators, buffers
        PtGroup *g = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_groupMutex);
            if (!g_groups.empty()) g = g_groups.back();
        }
        if (!g) break;
        pt_group_destroy(g);
    }
    pt_free();                                    // the calling thread's current context (normally the default one) + the scan library
    std::vector<State *> all;
    {
        std::lock_guard<std::mutex> lock(g_ctxMutex);
        all = g_contexts;
    }
    all.push_back(&g_default);
    for (State *st : all) {                       // ... and every other renderer the process still holds
        t_ctx = st;
        free_renderer();
    }
    t_ctx = &g_default;
}
void register_exit_handler() {
    static std::once_flag once;                   // (several host threads may drive contexts of their own)
    std::call_once(once, [] { atexit(exit_handler); });
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        HIPCHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
        return PT_OK;
    }
};

}  // namespace

// =====================================================================================================
// C ABI (the library is built with -fvisibility=hidden: these entry points are all it exports)
// =====================================================================================================
#pragma GCC visibility push(default)
extern "C" {

const char *pt_last_error(void) { return g_err.c_str(); }
int pt_device_count(void) { return count_devices(); }
int pt_abi_version(void) { return PT_AMD_ABI_VERSION; }

void pt_free(void) {
    // the scan library's per-stream workspaces (allocated on first use, with or without a renderer)
    scan_release();
    free_renderer();
}

}  // extern "C"
#pragma GCC visibility pop
namespace {
// everything pt_init allocated (pt_free, and pt_init's own restart)
void free_renderer() {
    // pathtraceFree before the first Init (src/main.cpp:91-94) must be a no-op
    if (!R().init && !R().image && !R().dgeoms && R().nslots == 0 && !R().hostFault && !R().pinnedHost) return;
    if (R().device >= 0 && R().nslots > 0) (void)hipSetDevice(R().device);     // (a host that switched devices in between)
    for (int i = 0; i < kMaxSlots; ++i)
        if (R().slot[i].stream) (void)hipStreamSynchronize(R().slot[i].stream);
    (void)hipStreamSynchronize(R().stream);
    for (auto &pr : R().evBounce) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    R().evBounce.clear();
    for (hipEvent_t e : R().evFree) (void)hipEventDestroy(e);
    R().evFree.clear();
    for (int i = 0; i < kMaxSlots; ++i) {
        Slot &sl = R().slot[i];
        for (int k = 0; k < 2; ++k) {
            if (sl.pathbuf[k]) (void)hipFree(sl.pathbuf[k]);
            if (sl.chunkList[k]) (void)hipFree(sl.chunkList[k]);
        }
        if (sl.ctrl) (void)hipFree(sl.ctrl);
        if (sl.contrib) (void)hipFree(sl.contrib);
        if (sl.hitMask) (void)hipFree(sl.hitMask);
        if (sl.meshHit) (void)hipFree(sl.meshHit);
        if (sl.evDone) (void)hipEventDestroy(sl.evDone);
        if (sl.evCommitted) (void)hipEventDestroy(sl.evCommitted);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (R().pinnedHost) (void)hipHostUnregister(R().pinnedHost);
    if (R().hostFault) (void)hipHostFree(R().hostFault);
    if (R().ownImage && R().image) (void)hipFree(R().image);
    if (R().dgeoms) (void)hipFree(R().dgeoms);
    if (R().dGeomHit) (void)hipFree(R().dGeomHit);
    if (R().dRows) (void)hipFree(R().dRows);
    if (R().dmats) (void)hipFree(R().dmats);
    if (R().dwalls) (void)hipFree(R().dwalls);
    if (R().dSphCull) (void)hipFree(R().dSphCull);
    if (R().dSphGroups) (void)hipFree(R().dSphGroups);
    if (R().dClassIdx) (void)hipFree(R().dClassIdx);
    if (R().dRowOff) (void)hipFree(R().dRowOff);
    if (R().dRowIdx) (void)hipFree(R().dRowIdx);
    if (R().dMeshRecs) (void)hipFree(R().dMeshRecs);
    if (R().dWalkIdx) (void)hipFree(R().dWalkIdx);
    if (R().dWalkRowOff) (void)hipFree(R().dWalkRowOff);
    if (R().dWalkMeshRows) (void)hipFree(R().dWalkMeshRows);
    {   // (the registered meshes outlive the renderer: see State::meshes)
        std::vector<ptm::HostMesh> keep = std::move(R().meshes);
        R() = State();
        R().meshes = std::move(keep);
    }
}
}  // namespace
#pragma GCC visibility push(default)
extern "C" {

int pt_set_meshes_sized(const PtMesh *meshes, int nmeshes, size_t mesh_struct_bytes) {
    if (mesh_struct_bytes != sizeof(PtMesh))
        return fail(PT_ERR_INVALID, "pt_set_meshes: the caller's PtMesh is %zu bytes, this library's %zu (built against another pt_amd.h: ABI version %d here)",
                    mesh_struct_bytes, sizeof(PtMesh), PT_AMD_ABI_VERSION);
    return pt_set_meshes(meshes, nmeshes);
}

int pt_set_meshes(const PtMesh *meshes, int nmeshes) {
    if (nmeshes < 0 || (nmeshes && !meshes)) return fail(PT_ERR_INVALID, "pt_set_meshes: null argument");
    for (int i = 0; i < nmeshes; ++i) {
        if (meshes[i].geom < 0 || meshes[i].ntris < 1 || !meshes[i].tris) return fail(PT_ERR_INVALID, "pt_set_meshes: mesh %d is empty", i);
        for (int j = 0; j < i; ++j)
            if (meshes[j].geom == meshes[i].geom) return fail(PT_ERR_INVALID, "pt_set_meshes: geom %d given twice", meshes[i].geom);
        for (size_t q = 0; q < 9 * (size_t)meshes[i].ntris; ++q)
            if (!std::isfinite(meshes[i].tris[q])) return fail(PT_ERR_INVALID, "pt_set_meshes: mesh %d holds a non-finite coordinate", i);
        if (meshes[i].normals)
            for (size_t q = 0; q < 9 * (size_t)meshes[i].ntris; ++q)
                if (!std::isfinite(meshes[i].normals[q])) return fail(PT_ERR_INVALID, "pt_set_meshes: mesh %d holds a non-finite normal", i);
    }
    R().meshes.clear();
    for (int i = 0; i < nmeshes; ++i) {
        ptm::HostMesh m;
        m.geom = meshes[i].geom;
        m.tris.assign(meshes[i].tris, meshes[i].tris + 9 * (size_t)meshes[i].ntris);
        if (meshes[i].normals) m.normals.assign(meshes[i].normals, meshes[i].normals + 9 * (size_t)meshes[i].ntris);
        if (meshes[i].materials) m.mats.assign(meshes[i].materials, meshes[i].materials + (size_t)meshes[i].ntris);
        R().meshes.push_back(std::move(m));
    }
    return PT_OK;
}

int pt_init(const PtCamera *cam, const PtGeom *geoms, int ngeoms, const PtMaterial *mats, int nmats, int traceDepth,
            const PtOptions *opts) {
    if (!cam || ngeoms < 0 || nmats < 0 || (ngeoms && !geoms) || (nmats && !mats))
        return fail(PT_ERR_INVALID, "pt_init: null argument");
    if (cam->resolution[0] <= 0 || cam->resolution[1] <= 0) return fail(PT_ERR_INVALID, "pt_init: bad resolution");
    const bool direct = opts && (opts->flags & PT_FLAG_DIRECT_LIGHTING);
    if (traceDepth < 1 || traceDepth + (direct ? 1 : 0) > PT_MAX_DEPTH)
        return fail(PT_ERR_INVALID, "pt_init: traceDepth must be 1..%d", PT_MAX_DEPTH - (direct ? 1 : 0));
    if (opts && (!(opts->lens_radius >= 0.0f) || (opts->lens_radius > 0.0f && !(opts->focal_distance > 0.0f))))
        return fail(PT_ERR_INVALID, "pt_init: lens_radius must be >= 0 and focal_distance > 0 with a lens");
    if ((long long)cam->resolution[0] * cam->resolution[1] > (1ll << 30)) return fail(PT_ERR_INVALID, "pt_init: frame too large");
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_SPHERE && geoms[i].type != PT_CUBE && geoms[i].type != PT_MESH) return fail(PT_ERR_INVALID, "pt_init: geom %d has unknown type", i);
        if (geoms[i].type == PT_MESH && !mesh_of(i)) return fail(PT_ERR_INVALID, "pt_init: geom %d is a mesh without triangles (pt_set_meshes)", i);
        if (geoms[i].materialid < 0 || geoms[i].materialid >= nmats) return fail(PT_ERR_INVALID, "pt_init: geom %d references material %d", i, geoms[i].materialid);
    }
    for (const ptm::HostMesh &m : R().meshes)
        if (m.geom < 0 || m.geom >= ngeoms || geoms[m.geom].type != PT_MESH)
            return fail(PT_ERR_INVALID, "pt_init: triangles registered for geom %d, which is not a mesh of this scene (pt_set_meshes)", m.geom);
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "pt_init: no HIP device (this library has no CPU fallback)");
    register_exit_handler();
    free_renderer();

    PtOptions o;
    memset(&o, 0, sizeof o);
    o.shard_count = 1;
    o.device = -1;
    if (opts) o = *opts;
    if (o.shard_count < 1 || o.shard_rank < 0 || o.shard_rank >= o.shard_count) return fail(PT_ERR_INVALID, "pt_init: bad shard %d/%d", o.shard_rank, o.shard_count);
    if (o.pipeline_depth < 0 || o.pipeline_depth > kMaxSlots) return fail(PT_ERR_INVALID, "pt_init: pipeline_depth must be 0..%d", kMaxSlots);
    if (o.max_batch < 0 || o.max_batch > PT_MAX_BATCH) return fail(PT_ERR_INVALID, "pt_init: max_batch must be 0..%d", PT_MAX_BATCH);
    if (o.device >= 0) HIPCHECK(hipSetDevice(o.device));
    HIPCHECK(hipGetDevice(&R().device));
    R().stream = (hipStream_t)o.stream;
    R().flags = o.flags;
    R().cam = *cam;

    HIPCHECK(hipHostMalloc((void **)&R().hostFault, sizeof(uint32_t), hipHostMallocMapped));
    *R().hostFault = 0u;
    HIPCHECK(hipHostGetDevicePointer((void **)&R().hostFaultDev, R().hostFault, 0));

    const int Wd = cam->resolution[0], H = cam->resolution[1];
    R().P = Wd * H;
    const int rows = H > o.shard_rank ? (H - o.shard_rank + o.shard_count - 1) / o.shard_count : 0;
    R().nLocal = rows * Wd;

    KParams &k = R().prm;
    memset(&k, 0, sizeof k);
    camera_params(*cam, k);
    const H3 view{cam->view.x, cam->view.y, cam->view.z};
    k.shardRank = o.shard_rank; k.shardCount = o.shard_count;
    k.nLocal = R().nLocal;
    magic_divisor((uint32_t)Wd, k.magicW, k.shiftW);
    magic_divisor((uint32_t)o.shard_count, k.magicS, k.shiftS);
    k.contribLocal = (o.shard_count > 1 && (long long)Wd * H < (1ll << 27)) ? 1 : 0;   // (the multiply-shift divisions hold below 2^27)
    // camera-ray tiles lie on rows padded to a multiple of the tile size (KParams::Wp)
    k.Wp = (Wd + kBlock - 1) / kBlock * kBlock;
    if ((long long)rows * k.Wp >= (1ll << 30)) return fail(PT_ERR_INVALID, "pt_init: frame too large (rows x padded width must stay below 2^30)");
    k.nLocalPad = rows * k.Wp;
    magic_divisor((uint32_t)k.Wp, k.magicWp, k.shiftWp);
    magic_divisor((uint32_t)std::max(k.nLocalPad, 1), k.magicN, k.shiftN);
    for (uint32_t d : {(uint32_t)Wd, (uint32_t)k.Wp, (uint32_t)std::max(k.nLocalPad, 1)}) {       // self-check on the edges of every quotient range
        uint32_t m, sh;
        magic_divisor(d, m, sh);
        for (uint64_t q = 0; q * d < (1ull << 30); q = q < 64 ? q + 1 : q * 2 + 1)
            for (uint64_t n : {q * d, q * d + d - 1, (uint64_t)((1ull << 30) - 1) - q})
                if (n < (1ull << 30) && (uint32_t)((n * m) >> sh) != (uint32_t)(n / d))
                    return fail(PT_ERR_INVALID, "pt_init: magic division self-check failed for d=%u n=%llu", d, (unsigned long long)n);
    }
    k.ngeoms = ngeoms; k.nmats = nmats;
    // direct lighting: bounce `traceDepth` aims its diffuse scatter at a light and one more launch collects
    k.traceDepth = traceDepth + (direct ? 1 : 0);
    k.directDepth = direct ? traceDepth : 0;
    // the emitters of the direct-lighting bounce, file order: primitives with an emissive material, sampled through their unit cube --
    // and meshes (round 5) whose own or any of whose faces' materials emits, through the object-space bounds of their vertices
    // (the CPU oracle restates this loop operation for operation: its rebuild_emitters)
    k.nEmit = 0;
    for (int i = 0; i < ngeoms && k.nEmit < kEmitMax; ++i) {
        bool emits = mats[geoms[i].materialid].emittance > 0.0f;
        float c[3] = {0.0f, 0.0f, 0.0f}, e[3] = {1.0f, 1.0f, 1.0f};
        if (geoms[i].type == PT_MESH) {
            const ptm::HostMesh *hm_ = mesh_of(i);
            if (!hm_ || hm_->tris.size() < 9) continue;
            for (int fm : hm_->mats) emits = emits || (fm >= 0 && fm < nmats && mats[fm].emittance > 0.0f);
            float lo[3] = {hm_->tris[0], hm_->tris[1], hm_->tris[2]}, hi[3] = {lo[0], lo[1], lo[2]};
            for (size_t q = 0; q + 2 < hm_->tris.size(); q += 3)
                for (int a = 0; a < 3; ++a) {
                    lo[a] = lo[a] < hm_->tris[q + a] ? lo[a] : hm_->tris[q + a];
                    hi[a] = hi[a] < hm_->tris[q + a] ? hm_->tris[q + a] : hi[a];
                }
            for (int a = 0; a < 3; ++a) { c[a] = (lo[a] + hi[a]) * 0.5f; e[a] = hi[a] - lo[a]; }
        }
        if (!emits) continue;
        const PtVec3 sc = geoms[i].scale;
        const float sx = sc.x * e[0], sy = sc.y * e[1], sz = sc.z * e[2];
        k.emitRho2[k.nEmit] = ((sx * sx + sy * sy) + sz * sz) * 0.25f;
        for (int a = 0; a < 3; ++a) { k.emitBox[k.nEmit][a] = c[a]; k.emitBox[k.nEmit][3 + a] = e[a]; }
        k.emitGeom[k.nEmit++] = i;
    }
    k.lensRadius = o.lens_radius;
    k.focalDistance = o.focal_distance;
    {
        const H3 vn = hnormalize(view);
        k.viewN[0] = vn.x; k.viewN[1] = vn.y; k.viewN[2] = vn.z;
    }
    R().dof = o.lens_radius > 0.0f;
    // plain: nothing in the scene takes the scatter's rarer branches (PT_AMD_NO_PLAIN: experiments / tests only)
    R().plain = !direct && !(o.flags & PT_FLAG_MIXTURE_WEIGHTED) && !(getenv("PT_AMD_NO_PLAIN") && atoi(getenv("PT_AMD_NO_PLAIN")));
    for (int i = 0; i < nmats; ++i)
        if (mats[i].hasRefractive > 0.0f || (mats[i].hasReflective > 0.0f && mats[i].specularExponent > 0.0f)) R().plain = false;

    if (o.accum_dev) {
        R().image = o.accum_dev;
        R().ownImage = false;
    } else {
        const size_t n = (R().flags & PT_FLAG_ACCUM_SHARD_ROWS) ? (size_t)(R().nLocal > 0 ? R().nLocal : 1) : (size_t)R().P;
        HIPCHECK(hipMalloc(&R().image, n * 3 * sizeof(float)));
        R().ownImage = true;
        HIPCHECK(hipMemsetAsync(R().image, 0, n * 3 * sizeof(float), R().stream));
    }
    // Path pools: a bounce's queue is kSeg = kCls x kSub segments, each a list of chunks handed out on demand, one ahead of
    // their use (ptk::reserveRun).  At most nLocal * maxBatch paths are alive; every segment may end in a partly filled
    // chunk and holds one chunk installed ahead: ceil(paths / chunk) + 2 kSeg chunks always suffice, whatever the
    // distribution over the classes (+ the trash chunk 0).  Chunk size: a power of two, at least 2048 (rounds 1-3: ~1/1024 of the
    // paths, so that the slack stayed around 10 % while a chunk outlasts the appends of one memory round trip).
    R().maxBatch = o.max_batch > 0 ? o.max_batch : 1;
    {   // a path carries pixelIndex | batch index << pixBits in ONE word (ptk::PathC)
        int pixBits = 1, batchBits = 0;
        while (((long long)k.W * k.H - 1) >> pixBits) ++pixBits;
        while ((R().maxBatch - 1) >> batchBits) ++batchBits;
        if (pixBits + batchBits > 32)
            return fail(PT_ERR_INVALID, "pt_init: %d x %d pixels and max_batch %d need %d + %d bits of a path's 32-bit index word: lower max_batch", k.W, k.H,
                        R().maxBatch, pixBits, batchBits);
        k.pixBits = pixBits;
    }
    if (R().maxBatch == 1) R().flags &= ~PT_FLAG_TRACE_AHEAD;   // nothing to trace ahead with: every call traces its own iteration
    // slots are 32-bit element indices with 32-bit byte offsets: paths per pool must stay below 2^30
    const long long maxPaths = (long long)R().nLocal * R().maxBatch;
    if (maxPaths > (1ll << 29)) return fail(PT_ERR_INVALID, "pt_init: max_batch x pixels too large (limit 2^29 paths per batch)");
    if ((long long)k.nLocalPad * R().maxBatch >= (1ll << 30))     // (the camera-ray tiles' index space: rows padded to the tile size)
        return fail(PT_ERR_INVALID, "pt_init: max_batch x rows x padded width too large (limit 2^30)");
    R().numTilesMax = (int)((maxPaths + kBlock - 1) / kBlock) + kSeg;
    // (round 4: ~1/256 of the paths, at most 2^18, where it was 1/1024 and 2^17 -- a run that opens a new chunk pays a dependent
    // look-up of the chunk list INSIDE the reservation's window, and four times fewer of them are +2 % on C2, +4 % on the closed box
    // (profiles/r04_chunk_size_sweep.txt); the slack of 2 kSeg chunks then doubles a mid-sized pool, which 288 GB shrug off)
    k.chunkShift = kMinChunkShift;
    while (k.chunkShift < 18 && (maxPaths >> k.chunkShift) > 256) ++k.chunkShift;
    if (const char *e = getenv("PT_AMD_CHUNK_SHIFT")) {      // experiments only
        const int v = atoi(e);
        if (v >= kMinChunkShift && v <= 20) k.chunkShift = v;
    }
    const long long chunkPaths = 1ll << k.chunkShift;
    R().poolChunks = (int)((maxPaths + chunkPaths - 1) / chunkPaths) + 2 * kSeg + 1;
    if (const char *e = getenv("PT_AMD_POOL_CHUNKS")) {      // tests only: an undersized pool must fail loudly (PT_ERR_DEVICE)
        const int v = atoi(e);
        if (v >= kSeg + 2) R().poolChunks = v;
    }
    k.poolChunks = R().poolChunks;
    const size_t cap = (size_t)R().poolChunks << k.chunkShift;
    // scenes with meshes: a record per path of a bounce's input queue (camera rays: per pixel of the tiles' padded index space) -- BounceArgs::meshHit
    bool anyMesh = false;
    for (int i = 0; i < ngeoms; ++i) anyMesh = anyMesh || geoms[i].type == PT_MESH;
    const size_t meshHitWords = anyMesh ? std::max(cap, (size_t)k.nLocalPad * (size_t)R().maxBatch) : 0;
    // Iterations are independent (RNG keyed on pixel/iteration/depth), so up to `nslots` of them are in flight
    // on their own streams; the small late-bounce launches of one overlap the big early launches of the next.
    R().nslots = o.pipeline_depth > 0 ? o.pipeline_depth : 3;
    {   // the memory budget BEFORE the first large allocation: a batch that does not fit fails here, with nothing to undo
        const size_t cpx = k.contribLocal ? (size_t)(R().nLocal > 0 ? R().nLocal : 1) : (size_t)R().P;
        const size_t perSlot = 2 * (cap * kNumArrays * sizeof(float) + (size_t)kSeg * R().poolChunks * sizeof(unsigned long long)) + sizeof(Ctrl) +
                               (size_t)R().maxBatch * cpx * 3 * sizeof(float) + (size_t)((R().maxBatch + 31) / 32) * cpx * sizeof(uint32_t) +
                               meshHitWords * sizeof(unsigned long long);
        const size_t need = perSlot * (size_t)R().nslots;
        size_t freeB = 0, totalB = 0;
        HIPCHECK(hipMemGetInfo(&freeB, &totalB));
        if (need > freeB)
            return fail(PT_ERR_HIP, "pt_init: %.2f GB of path pools and radiance buffers (max_batch %d x pipeline_depth %d) exceed the %.2f GB of free "
                        "device memory: lower max_batch or pipeline_depth", need / 1e9, R().maxBatch, R().nslots, freeB / 1e9);
    }
    for (int i = 0; i < R().nslots; ++i) {
        Slot &sl = R().slot[i];
        HIPCHECK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIPCHECK(hipMalloc(&sl.pathbuf[b], cap * kNumArrays * sizeof(float)));
            HIPCHECK(hipMalloc(&sl.chunkList[b], (size_t)kSeg * R().poolChunks * sizeof(unsigned long long)));
            HIPCHECK(hipMemset(sl.chunkList[b], 0, (size_t)kSeg * R().poolChunks * sizeof(unsigned long long)));
        }
        HIPCHECK(hipMalloc(&sl.ctrl, sizeof(Ctrl)));
        int rcc = reset_ctrl(sl.ctrl, nullptr);
        if (rcc) return rcc;
        // radiance buffers and iteration masks: the frame's pixels, or only this shard's (KParams::contribLocal)
        const size_t cpx = k.contribLocal ? (size_t)(R().nLocal > 0 ? R().nLocal : 1) : (size_t)R().P;
        HIPCHECK(hipMalloc(&sl.contrib, (size_t)R().maxBatch * cpx * 3 * sizeof(float)));
        HIPCHECK(hipMemset(sl.contrib, 0, (size_t)R().maxBatch * cpx * 3 * sizeof(float)));
        HIPCHECK(hipMalloc(&sl.hitMask, (size_t)((R().maxBatch + 31) / 32) * cpx * sizeof(uint32_t)));
        HIPCHECK(hipMemset(sl.hitMask, 0, (size_t)((R().maxBatch + 31) / 32) * cpx * sizeof(uint32_t)));
        if (meshHitWords) HIPCHECK(hipMalloc(&sl.meshHit, meshHitWords * sizeof(unsigned long long)));
        HIPCHECK(hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&sl.evCommitted, hipEventDisableTiming));
    }

    std::vector<GeomDev> hg(ngeoms ? ngeoms : 1);
    std::vector<MaterialDev> hm(nmats ? nmats : 1);
    // triangle meshes: one record array for the scene, a hierarchy per mesh (pt_mesh.h)
    std::vector<ptd::MeshUnit> meshRecs;
    int meshStackNeed = 0;
    const bool flatMeshes = getenv("PT_AMD_MESH_FLAT") && atoi(getenv("PT_AMD_MESH_FLAT"));   // tests only: no hierarchy
    std::vector<std::array<float, 6>> meshBox(ngeoms ? ngeoms : 1);
    std::vector<const float *> boxes(ngeoms ? ngeoms : 1, nullptr);
    for (int i = 0; i < ngeoms; ++i) {
        float *box = meshBox[i].data();
        const bool isMesh = geoms[i].type == PT_MESH;
        if (isMesh) boxes[i] = box;
        uint32_t root = ptd::kMeshEnd, stride = 0;
        const uint32_t unit0 = (uint32_t)meshRecs.size();        // (the mesh's units: its triangles -- and normals -- lie in [unit0, root))
        if (isMesh) {
            const ptm::HostMesh *hm_ = mesh_of(i);
            for (int fm : hm_->mats)
                if (fm >= nmats) return fail(PT_ERR_INVALID, "pt_init: a face of mesh geom %d names material %d of %d", i, fm, nmats);
            const ptm::MeshLayout lay = ptm::appendMesh(hm_->tris.data(), (int)(hm_->tris.size() / 9), flatMeshes, meshRecs, box,
                                                        hm_->normals.empty() ? nullptr : hm_->normals.data(), hm_->mats.empty() ? nullptr : hm_->mats.data());
            root = lay.root;
            stride = lay.stride;
            meshStackNeed = std::max(meshStackNeed, lay.stackNeed);
            if (meshRecs.size() >= (1ull << 31)) return fail(PT_ERR_INVALID, "pt_init: too many triangles");
        }
        pack_geom(geoms[i], hg[i], k.pos, isMesh ? box : nullptr);
        hg[i].meshRoot = root;
        if (isMesh) { hg[i].meshStride = stride; hg[i].meshUnit0 = unit0; hg[i].meshUnit1 = root; }
        if (geoms[i].type == PT_CUBE) {
            if (k.nCubes >= 32767) return fail(PT_ERR_INVALID, "pt_init: more than 32767 cubes");
            hg[i].frameSlot = (short)k.nCubes++;
        }
    }
    // camera rays: pixel rectangles, their union and the per-row lists (thin lens: none -- rays start anywhere on the lens)
    CameraCull cc;
    const bool cullOff = R().dof || (getenv("PT_AMD_NO_CAMERA_CULL") && atoi(getenv("PT_AMD_NO_CAMERA_CULL")));   // (the variable: tests only)
    build_camera_cull(geoms, ngeoms, k, cullOff, boxes, hg, cc);
    for (int a = 0; a < 4; ++a) k.sceneRect[a] = cc.sceneRect[a];
    {   // The camera-ray tiles' index space covers only the column bands (of kBlock pixels) and the rows of this shard that meet the
        // scene rectangle: at 16:9 two of Cornell's five bands lie outside it, and a workgroup spent a tenth of the launch
        // stepping over their tiles one by one.  The pixels never visited are misses whatever their jitter: tallied at once.
        const int perRow = k.Wp / kBlock;
        int c0 = 0, c1 = perRow - 1, r0 = 0, r1 = rows - 1;
        if (cc.sceneRect[0] > cc.sceneRect[2] || cc.sceneRect[1] > cc.sceneRect[3]) {      // nothing can be hit
            c1 = -1; r1 = -1;
        } else {
            c0 = std::max(cc.sceneRect[0], 0) / kBlock;
            c1 = std::min(std::min(cc.sceneRect[2], Wd - 1) / kBlock, perRow - 1);
            // rows y = lr * shard_count + shard_rank inside [sceneRect[1], sceneRect[3]]
            const int y0 = std::max(cc.sceneRect[1], 0), y1 = std::min(cc.sceneRect[3], H - 1);
            r0 = y0 <= o.shard_rank ? 0 : (y0 - o.shard_rank + o.shard_count - 1) / o.shard_count;
            r1 = y1 < o.shard_rank ? -1 : std::min((y1 - o.shard_rank) / o.shard_count, rows - 1);
        }
        const int nCols = std::max(c1 - c0 + 1, 0), nRows = std::max(r1 - r0 + 1, 0);
        const long long visited = nCols > 0 && nRows > 0 ? (long long)nRows * (std::min(Wd, (c1 + 1) * kBlock) - c0 * kBlock) : 0;
        k.firstY0 = (nRows > 0 ? r0 : 0) * o.shard_count + o.shard_rank;      // (the first column: the band of sceneRect[0], k_bounce)
        k.firstSkipped = (int)((long long)R().nLocal - visited);
        k.Wp = std::max(nCols, 1) * kBlock;                       // (>= one band: the divisions below stay defined)
        k.nLocalPad = nCols > 0 ? nRows * k.Wp : 0;
        magic_divisor((uint32_t)k.Wp, k.magicWp, k.shiftWp);
        magic_divisor((uint32_t)std::max(k.nLocalPad, 1), k.magicN, k.shiftN);
        for (uint32_t d : {(uint32_t)k.Wp, (uint32_t)std::max(k.nLocalPad, 1)}) {
            uint32_t m, sh;
            magic_divisor(d, m, sh);
            for (uint64_t q = 0; q * d < (1ull << 30); q = q < 64 ? q + 1 : q * 2 + 1)
                for (uint64_t n : {q * d, q * d + d - 1, (uint64_t)((1ull << 30) - 1) - q})
                    if (n < (1ull << 30) && (uint32_t)((n * m) >> sh) != (uint32_t)(n / d))
                        return fail(PT_ERR_INVALID, "pt_init: magic division self-check failed for d=%u n=%llu", d, (unsigned long long)n);
        }
    }
    for (int i = 0; i < nmats; ++i) pack_material(mats[i], hm[i]);
    // Small primitives the queue is binned by (k_bounce): the spheres when there are at most kBinMax of them, then the
    // cubes whose bounding ball is small against the scene's (<= 0.3 of its radius), smallest first.  A choice that only
    // steers which tiles skip which tests; results never depend on it.
    {
        double cm[3] = {0, 0, 0}, sceneR = 0;
        for (int i = 0; i < ngeoms; ++i)
            for (int a = 0; a < 3; ++a) cm[a] += hg[i].centre[a] / std::max(ngeoms, 1);
        for (int i = 0; i < ngeoms; ++i) {
            const double dx = hg[i].centre[0] - cm[0], dy = hg[i].centre[1] - cm[1], dz = hg[i].centre[2] - cm[2];
            const double r = std::sqrt(dx * dx + dy * dy + dz * dz) + hg[i].boundR;
            if (std::isfinite(r)) sceneR = std::max(sceneR, r);
        }
        int nsph = 0;
        for (int i = 0; i < ngeoms; ++i) nsph += geoms[i].type == PT_SPHERE;
        std::vector<std::pair<double, int>> cand;
        for (int i = 0; i < ngeoms; ++i) {
            if (!std::isfinite(hg[i].cullR2)) continue;                       // never culled: cannot take part
            const double r = hg[i].boundR;
            if (geoms[i].type == PT_SPHERE) { if (nsph <= kBinMax) cand.emplace_back(-1.0, i); }   // spheres first
            else if (r <= 0.3 * sceneR) cand.emplace_back(r, i);
        }
        std::sort(cand.begin(), cand.end());
        k.nBinned = 0;
        for (size_t c = 0; c < cand.size() && k.nBinned < kBinMax; ++c) {
            k.binGeom[k.nBinned++] = cand[c].second;
            hg[cand[c].second].binned = 1;
            hg[cand[c].second].flags |= 2;
            hg[cand[c].second].cullFlags |= 2;
        }
    }
    // Mesh scenes bin by two candidate bits (pt_trace.h: kClsMax): the costliest binned mesh (most triangles) alone in group 1 when there is
    // another binned primitive beside it, everything else in group 0 -- a tile of the next bounce then walks that mesh only when its paths
    // can hit it, with all its lanes, instead of every cand tile walking every mesh with some.
    int binGroup[kBinMax] = {0, 0, 0, 0};
    if (!meshRecs.empty() && k.nBinned > 1) {
        int bestB = -1;
        size_t bestTris = 0;
        for (int b = 0; b < k.nBinned; ++b)
            if (geoms[k.binGeom[b]].type == PT_MESH) {
                const size_t nt = mesh_of(k.binGeom[b])->tris.size() / 9;
                if (nt > bestTris) { bestTris = nt; bestB = b; }
            }
        if (bestB >= 0) binGroup[bestB] = 1;
    }
    for (int b = 0; b < kBinMax; ++b) {                      // (KParams::binCull: the binned primitives' culling groups, inline)
        for (int q = 0; q < 8; ++q) k.binCull[b][q] = 0.0f;
        k.binCull[b][3] = -INFINITY;                          // beyond nBinned: certified for everybody
        if (b < k.nBinned) {
            const GeomDev &G = hg[k.binGeom[b]];
            k.binCull[b][0] = G.centre[0]; k.binCull[b][1] = G.centre[1]; k.binCull[b][2] = G.centre[2];
            k.binCull[b][3] = G.cullR2; k.binCull[b][4] = G.cullK;
            // word 5: the primitive's candidate bit in a survivor's class (k_bounce<..., MESH>): 1 = group 0, 2 = group 1
            const uint32_t bit = 1u << binGroup[b];
            memcpy(&k.binCull[b][5], &bit, sizeof bit);
        }
    }
    // Walls: the large cubes -- not binned, finite -- at most kWallMax of them, the largest first.  Survivors are classed by
    // the one wall they can still hit (ptd::wallCertainMiss against the inflated world boxes computed here), so a tile of
    // the next bounce tests one wall instead of all of them, and a survivor that can hit nothing at all ends at once.
    // A choice that only steers which tiles skip which tests; results never depend on it.
    std::vector<WallBox> hw(kWallMax);
    {
        std::vector<int> wallGeom;
        choose_walls(geoms, ngeoms, hg, k, hw, wallGeom);
        for (int w = 0; w < k.nWalls; ++w) {
            hg[wallGeom[w]].flags |= (w + 1) << 2;
            hg[wallGeom[w]].cullFlags |= (w + 1) << 2;
        }
        if (const char *e = getenv("PT_AMD_NO_WALLS")) { if (atoi(e)) { for (int i = 0; i < ngeoms; ++i) { hg[i].flags &= 3; hg[i].cullFlags &= 3; } k.nWalls = 0; k.wallOMax = 0.0f; k.nSlotWalls = 0; k.nPlaneWalls = 0; } }   // experiments only
        if (k.nPlaneWalls > 0) R().plain = false;      // (the rotated walls' certificate lives in the general instantiations only: k_bounce, wallPlanesOriented)
        k.allClassified = k.nWalls > 0 ? 1 : 0;
        for (int i = 0; i < ngeoms; ++i)
            if (!hg[i].binned && (hg[i].flags & 28) == 0) k.allClassified = 0;
    }
    k.emittersBinned = k.nBinned > 0 ? 1 : 0;
    for (int i = 0; i < ngeoms; ++i) {
        bool emits = mats[geoms[i].materialid].emittance > 0.0f;
        if (geoms[i].type == PT_MESH)                   // (a mesh emits when any of its faces' own materials does)
            if (const ptm::HostMesh *hm_ = mesh_of(i))
                for (int fm : hm_->mats) emits = emits || (fm >= 0 && mats[fm].emittance > 0.0f);
        if (emits && !hg[i].binned) k.emittersBinned = 0;
    }
    HIPCHECK(hipMalloc(&R().dgeoms, hg.size() * sizeof(GeomDev)));
    HIPCHECK(hipMalloc(&R().dmats, hm.size() * sizeof(MaterialDev)));
    HIPCHECK(hipMemcpy(R().dgeoms, hg.data(), hg.size() * sizeof(GeomDev), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(R().dmats, hm.data(), hm.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));
    {   // the per-primitive hit records, ready-made: a workgroup's prologue copies them to LDS in one round trip instead of following
        // primitive -> material index -> material on the device (every workgroup of every launch did)
        std::vector<GeomHitDev> hh(hg.size());
        for (size_t i = 0; i < hg.size(); ++i) {
            GeomHitDev &h = hh[i];
            memset(&h, 0, sizeof h);
            const GeomDev &G = hg[i];
            const int mi = (int)i < ngeoms && G.material >= 0 && G.material < nmats ? G.material : 0;
            const MaterialDev &M = hm[(size_t)mi];
            h.type = G.type;
            h.emittance = M.emittance; h.hasReflective = M.hasReflective; h.hasRefractive = M.hasRefractive;
            for (int a = 0; a < 3; ++a) h.color[a] = M.color[a];
            h.material = G.material;
            memcpy(h.nm, G.invT, sizeof h.nm);
            memcpy(h.cubeFrame, G.cubeFrame, sizeof h.cubeFrame);
        }
        HIPCHECK(hipMalloc(&R().dGeomHit, hh.size() * sizeof(GeomHitDev)));
        HIPCHECK(hipMemcpy(R().dGeomHit, hh.data(), hh.size() * sizeof(GeomHitDev), hipMemcpyHostToDevice));
    }
    HIPCHECK(hipMalloc(&R().dwalls, hw.size() * sizeof(WallBox)));
    HIPCHECK(hipMemcpy(R().dwalls, hw.data(), hw.size() * sizeof(WallBox), hipMemcpyHostToDevice));
    R().mesh = !meshRecs.empty();
    if (R().mesh) {
        HIPCHECK(hipMalloc(&R().dMeshRecs, (meshRecs.size() + 4) * sizeof(ptd::MeshUnit)));     // (+ 4: a walk may read the record behind the last one)
        HIPCHECK(hipMemcpy(R().dMeshRecs, meshRecs.data(), meshRecs.size() * sizeof(ptd::MeshUnit), hipMemcpyHostToDevice));
    }

    // The SWEPT primitives of a scene with many small ones (round 5: cubes too -- rounds 2-4 swept spheres only, and 64 small cubes cost
    // 4.7 x what 64 spheres did, profiles/r05_generality.txt): every sphere, and every cube that is neither a wall nor binned.  Their
    // bounding balls are swept per lane from a packed table (ptk::SphereCull) instead of being visited one by one by the whole wave.
    std::vector<char> swept(ngeoms, 0);
    int nswept = 0, nsweptCubes = 0;
    for (int i = 0; i < ngeoms; ++i) {
        const bool smallCube = geoms[i].type == PT_CUBE && !hg[i].binned && (hg[i].flags & 28) == 0 && std::isfinite(hg[i].cullR2);
        swept[i] = geoms[i].type == PT_SPHERE || smallCube;
        nswept += swept[i];
    }
    R().many = nswept > kBinMax;
    if (R().many && ngeoms > 65535) return fail(PT_ERR_INVALID, "pt_init: more than 65535 primitives");
    if (!R().many) std::fill(swept.begin(), swept.end(), 0);
    for (int i = 0; i < ngeoms; ++i)
        if (swept[i] && geoms[i].type == PT_CUBE) {
            ++nsweptCubes;
            hg[i].flags |= 64;                        // (bit 6: a swept cube -- the camera-ray bounce lists it like a sphere)
            hg[i].cullFlags |= 64;
        }
    R().sweptCubes = nsweptCubes > 0;
    if (nsweptCubes) HIPCHECK(hipMemcpy(R().dgeoms, hg.data(), hg.size() * sizeof(GeomDev), hipMemcpyHostToDevice));
    if (R().many) {        // the later bounces take the swept primitives from a packed copy of their culling data (ptk::SphereCull)
        std::vector<SphereCull> sc;
        for (int i = 0; i < ngeoms; ++i)
            if (swept[i]) {
                SphereCull e;
                memset(&e, 0, sizeof e);
                for (int a = 0; a < 3; ++a) e.centre[a] = hg[i].centre[a];
                e.cullR2 = hg[i].cullR2;
                e.cullK = hg[i].cullK + kUnitDirSlack;      // (the sweep's direction is normalised approximately: sphereHalfLineExcess)
                e.geom = i;
                sc.push_back(e);
            }
        // the K |oc|^2 term of the certificate folded into the sweep's direction (ptd::sphereHalfLineExcessScaled): one factor for the
        // scene, from its largest K, and every threshold multiplied by its square -- both rounded upwards (the conservative side)
        double kmax = 0.0;
        for (const SphereCull &e : sc) kmax = std::max(kmax, (double)e.cullK);
        const float sdir = std::nextafter((float)std::sqrt(1.0 / (1.0 - kmax)), INFINITY);
        k.sphDirScale = sdir;
        for (SphereCull &e : sc)
            if (std::isfinite(e.cullR2)) e.cullR2 = std::nextafter((float)((double)e.cullR2 * (double)sdir * (double)sdir), INFINITY);
        // two spatial CLUSTERS (scenes without meshes, whose second candidate bit is free): build_sphere_clusters
        k.sphN0 = 0; k.sphOMax = 0.0f;
        for (int g = 0; g < 2; ++g) for (int q = 0; q < 8; ++q) k.sphBox[g][q] = 0.0f;
        if (meshRecs.empty()) {
            std::vector<int> binned(k.binGeom, k.binGeom + k.nBinned);
            build_sphere_clusters(geoms, ngeoms, hg, binned, sc, k.sphN0, k.sphOMax, k.sphBox);
        }
        if (k.sphOMax <= 0.0f) { k.sphN0 = 0; k.sphOMax = -1.0f; }      // no clusters: no certificate is issued, every tile sweeps the whole table
        else if (k.nWalls > 0) {
            // with the spheres behind candidate bits too, a survivor whose certificates leave no wall, no binned primitive and no cluster has
            // nothing left to hit (KParams::allClassified) -- when there is no primitive of another kind
            k.allClassified = 1;
            for (int i = 0; i < ngeoms; ++i)
                if (!hg[i].binned && (hg[i].flags & 28) == 0 && !swept[i]) k.allClassified = 0;
        }
        if (sc.size() % 2) sc.push_back(sc.back());      // (two per scalar load; testing a sphere twice changes nothing)
        // Hundreds of swept primitives (round 6): the flat sweep of the later bounces is linear in their number (512 spheres: 3.5 x the time of 64).
        // The table then comes in spatial groups of kSphGroupSize with a bounding ball each, and the later bounces take instantiations of their
        // own (k_bounce<..., GROUPS>: two-level sweep, no scene table in LDS).  PT_AMD_GROUPS=0 / 1: never / whenever possible (experiments, tests).
        k.nSphGroups = 0; k.grpN0 = 0; k.grpOMax = 0.0f; k.grpLds = 0;
        std::vector<SphereCull> groups;
        {
            const char *ge = getenv("PT_AMD_GROUPS");
            R().grouped = meshRecs.empty() && (ge ? atoi(ge) != 0 : nswept >= kGroupedMin);
            if (R().grouped) {
                int n0 = k.sphN0;
                const double ob = scene_origin_bound(geoms, ngeoms, hg);
                build_sphere_groups(sc, n0, ob, sdir, groups, k.grpN0);
                k.sphN0 = n0;
                k.nSphGroups = (int)(sc.size() / (size_t)kSphGroupSize);
                k.grpOMax = std::nextafter((float)ob, 0.0f);
                k.grpLds = (int)sc.size() <= kGroupLdsMax ? 1 : 0;
                HIPCHECK(hipMalloc(&R().dSphGroups, groups.size() * sizeof(SphereCull)));
                HIPCHECK(hipMemcpy(R().dSphGroups, groups.data(), groups.size() * sizeof(SphereCull), hipMemcpyHostToDevice));
            }
        }
        k.nSphCull = (int)sc.size();
        HIPCHECK(hipMalloc(&R().dSphCull, sc.size() * sizeof(SphereCull)));
        HIPCHECK(hipMemcpy(R().dSphCull, sc.data(), sc.size() * sizeof(SphereCull), hipMemcpyHostToDevice));
        // The tables a sphere-heavy workgroup stages in LDS -- the compact hit records, the cubes' face frames, the spheres' matrix rows,
        // the sweep's entry -> primitive map -- as ONE image in the kernel's own layout (k_bounce: S_GEOMHIT_SMALL .. behind S_SPH), so that
        // the prologue is a straight copy of 16-byte words: gathering them field by field from the primitives took ~30 dependent
        // round trips, 28 us at the head of every launch of C5 (profiles/timeline_phases.py: 62 k cycles against Cornell's 10 k).
        {
            const size_t hitB = manyHitBytes(ngeoms), frameB = (size_t)k.nCubes * 54 * sizeof(float) + manyFramePad(k.nCubes);
            // (scenes of hundreds of primitives: the matrix rows -- 112 B per primitive -- stay in global memory, KParams::ldsRowFloats = 0; with
            // them a workgroup of the 518-primitive scene took 103 KB of LDS, one per CU.  The limit: what four workgroups per CU leave each.)
            const size_t rowsAll = (size_t)ngeoms * kSphRowFloats * sizeof(float);
            const size_t ldsWithRows = sizeof(MaterialDev) * nmats + (size_t)miscWords(kClsMax) * sizeof(uint32_t) + hitB + frameB + rowsAll + (size_t)kListMax * kBlock * sizeof(uint16_t);
            const bool rowsInLds = ldsWithRows <= 40 * 1024 && !R().grouped && !(getenv("PT_AMD_ROWS_GLOBAL") && atoi(getenv("PT_AMD_ROWS_GLOBAL")));   // (the variable: tests only)
            k.ldsRowFloats = rowsInLds ? ngeoms * kSphRowFloats : 0;
            const size_t rowB = rowsInLds ? rowsAll : 0, mapB = ((size_t)k.nSphCull + 7) / 8 * 8 * sizeof(uint16_t);
            // (+ 64 bytes: behind the last cube's frames a row of NaNs -- what k_bounce<..., GROUPS>, which reads the frames from this image in
            // global memory, selects for a cube hit without an exit slab, as the other kernels select their NaN row in LDS)
            std::vector<unsigned char> blob(hitB + frameB + rowB + mapB + 64, 0);
            std::vector<float> rowsGlobal(rowsInLds ? 0 : (size_t)ngeoms * kSphRowFloats, 0.0f);
            GeomHitSmall *hs = reinterpret_cast<GeomHitSmall *>(blob.data());
            float *fr = reinterpret_cast<float *>(blob.data() + hitB);
            float *rows = rowsInLds ? reinterpret_cast<float *>(blob.data() + hitB + frameB) : rowsGlobal.data();
            uint16_t *map = reinterpret_cast<uint16_t *>(blob.data() + hitB + frameB + rowB);
            for (int g = 0; g < ngeoms; ++g) {
                const GeomDev &G = hg[g];
                memcpy(hs[g].nm, G.invT, sizeof hs[g].nm);
                hs[g].material = G.material; hs[g].type = G.type; hs[g].frame = G.type == 1 ? (int)G.frameSlot : 0;
                if (G.type == 1) memcpy(fr + (size_t)G.frameSlot * 54, G.cubeFrame, 54 * sizeof(float));
                float *r = rows + (size_t)g * kSphRowFloats;
                memcpy(r, G.inv, 12 * sizeof(float)); memcpy(r + 12, G.xf, 12 * sizeof(float)); memcpy(r + 24, G.invZ, 3 * sizeof(float));
            }
            for (int i = 0; i < k.nSphCull; ++i) map[i] = (uint16_t)sc[i].geom;
            if (R().grouped) {
                const float qnan = std::nanf("");
                for (int q = 0; q < 9; ++q) memcpy(blob.data() + hitB + (size_t)k.nCubes * 54 * sizeof(float) + q * sizeof(float), &qnan, sizeof qnan);
            }
            if (R().dGeomHit) (void)hipFree(R().dGeomHit);
            R().dGeomHit = nullptr;
            HIPCHECK(hipMalloc(&R().dGeomHit, blob.size()));
            HIPCHECK(hipMemcpy(R().dGeomHit, blob.data(), blob.size(), hipMemcpyHostToDevice));
            if (!rowsInLds) {
                HIPCHECK(hipMalloc(&R().dRows, rowsGlobal.size() * sizeof(float)));
                HIPCHECK(hipMemcpy(R().dRows, rowsGlobal.data(), rowsGlobal.size() * sizeof(float), hipMemcpyHostToDevice));
            }
        }
    }
    {   // Later bounces: which primitives a tile of queue class c looks at.  Class bit 3 = its paths may hit a binned primitive;
        // bits 0-2 in a scene with walls = the one wall they can still hit (6: any, 7: none), else the direction octant.
        std::vector<int> idx;
        const int ncls = (R().mesh || R().many) ? kClsMax : kCls;                 // (mesh and sphere-heavy scenes: two candidate bits, 32 classes)
        for (int c = 0; c < kClsMax; ++c) {
            k.classOff[c] = (int)idx.size();
            if (c >= ncls) continue;
            const int small = c >> 3;                                          // candidate bits: which groups of binned primitives
            const int wall = k.nWalls > 0 ? (c & 7) : 6;
            for (int i = 0; i < ngeoms; ++i) {
                if (swept[i]) continue;                                          // swept from their packed culling data
                if (hg[i].binned) {
                    int grp = 0;
                    for (int b = 0; b < k.nBinned; ++b)
                        if (k.binGeom[b] == i) grp = binGroup[b];
                    if (!((small >> grp) & 1)) continue;
                }
                const int w = (hg[i].flags >> 2) & 7;                          // 1 + index among the walls, 0: not one
                if (w != 0 && wall != 6 && w != wall + 1) continue;
                idx.push_back(i);
            }
        }
        k.classOff[kClsMax] = (int)idx.size();
        // camera rays: the per-row primitive lists (build_camera_cull)
        if (!cc.rowOff.empty()) {
            HIPCHECK(hipMalloc(&R().dRowOff, cc.rowOff.size() * sizeof(int)));
            HIPCHECK(hipMemcpy(R().dRowOff, cc.rowOff.data(), cc.rowOff.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHECK(hipMalloc(&R().dRowIdx, cc.rowIdx.size() * sizeof(int)));
            HIPCHECK(hipMemcpy(R().dRowIdx, cc.rowIdx.data(), cc.rowIdx.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        if (!meshRecs.empty()) {
            // the mesh walks (k_mesh_walk) look at the meshes alone: the classes' lists, one list of all, the rows' lists (pairs as rowIdx's)
            std::vector<int> w;
            for (int c = 0; c < kClsMax; ++c) {
                R().walkClassOff[c] = (int)w.size();
                for (int e = k.classOff[c]; e < (c + 1 < kClsMax ? k.classOff[c + 1] : (int)idx.size()); ++e)
                    if (hg[idx[e]].flags & 32) w.push_back(idx[e]);
            }
            R().walkClassOff[kClsMax] = (int)w.size();
            R().walkAll0 = (int)w.size();
            for (int i = 0; i < ngeoms; ++i)
                if (hg[i].flags & 32) {
                    if (w.size() - (size_t)R().walkAll0 >= 32767) return fail(PT_ERR_INVALID, "pt_init: more than 32767 meshes");
                    hg[i].frameSlot = (short)(w.size() - (size_t)R().walkAll0);     // (a mesh's ordinal: its row of the walk's LDS table)
                    w.push_back(i);
                }
            R().walkAll1 = (int)w.size();
            HIPCHECK(hipMemcpy(R().dgeoms, hg.data(), hg.size() * sizeof(GeomDev), hipMemcpyHostToDevice));
            {   // the walk's rows, one per mesh in the order of their ordinals (ptk::WalkMesh)
                const int nm = R().walkAll1 - R().walkAll0;
                std::vector<WalkMesh> rowsW((size_t)nm);
                for (int q = 0; q < nm; ++q) {
                    const GeomDev &G = hg[(size_t)w[(size_t)R().walkAll0 + q]];
                    WalkMesh &r = rowsW[(size_t)q];
                    memcpy(r.inv, G.inv, sizeof r.inv); memcpy(r.invZ, G.invZ, sizeof r.invZ);
                    r.root = G.meshRoot;
                    memcpy(r.xf, G.xf, sizeof r.xf); memcpy(r.camObj, G.camObj, sizeof r.camObj);
                    r.stride = G.meshStride;
                }
                HIPCHECK(hipMalloc(&R().dWalkMeshRows, std::max<size_t>(rowsW.size(), 1) * sizeof(WalkMesh)));
                HIPCHECK(hipMemcpy(R().dWalkMeshRows, rowsW.data(), rowsW.size() * sizeof(WalkMesh), hipMemcpyHostToDevice));
                const bool forceGlobal = getenv("PT_AMD_WALK_ROWS_GLOBAL") && atoi(getenv("PT_AMD_WALK_ROWS_GLOBAL"));      // tests only
                R().walkMeshLds = (nm <= kWalkMeshLdsMax && !forceGlobal) ? nm : 0;
            }
            if (!cc.rowOff.empty()) {
                if (w.size() % 2) w.push_back(0);                  // (the rows' entries are pairs: offsets count pairs from the array's start)
                std::vector<int> ro(cc.rowOff.size());
                for (size_t y = 0; y + 1 < cc.rowOff.size(); ++y) {
                    ro[y] = (int)(w.size() / 2);
                    for (int e = cc.rowOff[y]; e < cc.rowOff[y + 1]; ++e)
                        if (hg[cc.rowIdx[2 * e]].flags & 32) { w.push_back(cc.rowIdx[2 * e]); w.push_back(cc.rowIdx[2 * e + 1]); }
                }
                ro[cc.rowOff.size() - 1] = (int)(w.size() / 2);
                HIPCHECK(hipMalloc(&R().dWalkRowOff, ro.size() * sizeof(int)));
                HIPCHECK(hipMemcpy(R().dWalkRowOff, ro.data(), ro.size() * sizeof(int), hipMemcpyHostToDevice));
            }
            HIPCHECK(hipMalloc(&R().dWalkIdx, w.size() * sizeof(int)));
            HIPCHECK(hipMemcpy(R().dWalkIdx, w.data(), w.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        if (idx.empty()) idx.push_back(0);
        HIPCHECK(hipMalloc(&R().dClassIdx, idx.size() * sizeof(int)));
        HIPCHECK(hipMemcpy(R().dClassIdx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    const size_t ldsFixed = sizeof(MaterialDev) * nmats + (size_t)miscWords((R().mesh || R().many) ? kClsMax : kCls) * sizeof(uint32_t) +
                            (R().many ? manyHitBytes(ngeoms) + (size_t)k.nCubes * 54 * sizeof(float) + manyFramePad(k.nCubes) +
                                          (size_t)k.ldsRowFloats * sizeof(float)
                                    : sizeof(GeomHitDev) * ngeoms);
    // (sphere-heavy scenes: the camera-ray launch keeps the lanes' candidate lists behind the tables, the later ones only the sweep's
    // entry -> primitive map -- 4 KB less, which is what their seventh workgroup per CU needs)
    // (... and, behind the map, the pooled pass's pair descriptors: [kWaves][64] words)
    const size_t sphMapBytes = ((size_t)k.nSphCull + 7) / 8 * 8 * sizeof(uint16_t), pairBytes = (size_t)kBlock * sizeof(uint32_t);
    k.pairOff = (int)(ldsFixed + sphMapBytes);
    R().ldsBytes = ldsFixed + (R().many ? std::max((size_t)kListMax * kBlock * sizeof(uint16_t), sphMapBytes + pairBytes) : 0);
    R().ldsBytesNext = (R().many && !R().mesh) ? ldsFixed + sphMapBytes + pairBytes : 0;
    if (R().grouped && !R().dof)      // (the camera-ray bounce of a grouped scene: the materials and the lanes' candidate lists)
        R().ldsBytes = sizeof(MaterialDev) * nmats + (size_t)miscWords(kClsMax) * sizeof(uint32_t) + (size_t)kListMax * kBlock * sizeof(uint16_t) + 16;
    if (R().grouped)       // (the later bounces stage the materials and nothing else of the scene)
    {
        // (... and, behind them, the lanes' parked candidates: KParams::pairOff, [kCandPairs][kBlock] words)
        const size_t members = sizeof(MaterialDev) * nmats + (size_t)miscWords(kClsMax) * sizeof(uint32_t) + (k.grpLds ? ((size_t)k.nSphCull * 18 + 15) / 16 * 16 : 0);
        k.pairOff = (int)members;
        R().ldsBytesNext = members + (size_t)kCandPairs * kBlock * sizeof(uint32_t) + 16;
    }
    k.meshStackOff = 0;
    if (R().mesh) {        // (the lanes' stacks of far children belong to the walk's own launches: k_mesh_walk)
        R().ldsWalk = walkLdsBytes(meshStackNeed, R().walkMeshLds);
        if (R().ldsWalk > 160 * 1024) return fail(PT_ERR_INVALID, "pt_init: a mesh's hierarchy needs %d stack levels (%zu B of LDS for the lanes' stacks)", meshStackNeed, R().ldsWalk);
    }
    if (R().ldsBytesNext == 0) R().ldsBytesNext = R().ldsBytes;
    if (R().ldsBytes > 160 * 1024) return fail(PT_ERR_INVALID, "pt_init: scene does not fit the 160 KiB LDS (%zu B)", R().ldsBytes);
    if (nmats >= 4096) return fail(PT_ERR_INVALID, "pt_init: more than 4095 materials");      // (TileArgs::hot holds nmats in 12 bits)
    const void *kFirst = bounce_kernel(true, R().dof);
    const void *kNext = bounce_kernel(false, false);
    if (R().ldsBytes > 64 * 1024) {
        HIPCHECK(hipFuncSetAttribute(kFirst, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R().ldsBytes));
        HIPCHECK(hipFuncSetAttribute(kNext, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R().ldsBytes));
    }
    if (R().mesh)
        for (int first = 0; first < 2; ++first) {
            const void *kw = walk_kernel(first != 0, first && R().dof);
            if (R().ldsWalk > 64 * 1024) HIPCHECK(hipFuncSetAttribute(kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R().ldsWalk));
            int dev = 0, perCU = 0;
            HIPCHECK(hipGetDevice(&dev));
            hipDeviceProp_t prop;
            HIPCHECK(hipGetDeviceProperties(&prop, dev));
            HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kw, kBlock, R().ldsWalk));
            if (perCU < 1) perCU = 1;
            if (const char *e = getenv("PT_AMD_WALK_BLOCKS_PER_CU")) perCU = std::max(1, atoi(e));   // experiments only
            int &grid = first ? R().gridWalkFirst : R().gridWalk;
            grid = prop.multiProcessorCount * perCU;
            if (const char *e = getenv("PT_AMD_MAX_GRID")) grid = std::max(1, std::min(grid, atoi(e)));
        }
    for (int first = 0; first < 2; ++first) {
        int &grid = first ? R().gridFirst : R().grid;
        int rc = persistent_grid(first ? kFirst : kNext, first ? R().ldsBytes : R().ldsBytesNext, &grid);
        if (rc) return rc;
        if (grid > R().numTilesMax) grid = R().numTilesMax;
        grid = (grid / kSub) * kSub;      // T % kSub == blockIdx % kSub for every tile T of a workgroup (kSub: a multiple of the mesh scenes' 4 too)
        if (grid < kSub) grid = kSub;
    }
    // Camera-ray bounce: tile T covers pixels 256 T ... of the row-major frame and a workgroup owns the tiles b, b + grid,
    // ...  When the grid shares a factor with the tiles per row (1280 workgroups, 5 tiles per 1280-pixel row) a workgroup
    // stays in few column bands of the frame, and the bands outside the scene rectangle finish 15x earlier than the
    // others.  Either the grid can be made coprime to the tiles per row (both must stay multiples of kSub, so only for an
    // odd tile count per row), or the kernel rotates the k-th tile of a workgroup k bands to the right inside its row,
    // which needs a grid that is a multiple of the tiles per row (k_bounce<true, .>).
    R().prm.tilesPerRow = 0;
    if (R().prm.Wp / kBlock > 1) {
        const int perRow = R().prm.Wp / kBlock;
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        if (gcd(perRow, kSub) == 1) {
            for (int tries = 0; tries < 64 && R().gridFirst > kSub && gcd(R().gridFirst, perRow) != 1; ++tries) R().gridFirst -= kSub;
        } else {
            const int unit = perRow / gcd(perRow, kSub) * kSub;        // lcm(perRow, kSub)
            if (R().gridFirst >= 4 * unit) {
                R().gridFirst = R().gridFirst / unit * unit;
                R().prm.tilesPerRow = perRow;
            }
        }
    }
    if (getenv("PT_AMD_VERBOSE") && atoi(getenv("PT_AMD_VERBOSE")))       // experiments: what pt_init decided
        fprintf(stderr, "pt_init: lds %zu / %zu B, grid %d / %d, mesh %d many %d plain %d, binned %d walls %d (slots %d, planes %d) allClassified %d, sphCull %d (cluster 0: %d) omax %g\n",
                R().ldsBytes, R().ldsBytesNext, R().gridFirst, R().grid, (int)R().mesh, (int)R().many, (int)R().plain, k.nBinned, k.nWalls, k.nSlotWalls, k.nPlaneWalls, k.allClassified, k.nSphCull,
                k.sphN0, (double)k.sphOMax);
    HIPCHECK(hipDeviceSynchronize());
    R().init = true;
    g_err.clear();
    return PT_OK;
}

int pt_iterate_batch(int frame, int first_iter, int count, void *rgba8_dev) {
    (void)frame;  // always 0 in the reference (src/main.cpp:102)
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_iterate before pt_init");
    if (count < 1 || count > R().maxBatch) return fail(PT_ERR_INVALID, "pt_iterate_batch: count must be 1..max_batch (%d)", R().maxBatch);
    if (first_iter < 1 || first_iter + count - 1 >= kIterEnd)
        return fail(PT_ERR_INVALID, "pt_iterate: iter must be 1..4194303 (seed bits, pathtrace.cu:43)");
    // every argument is checked BEFORE anything is enqueued: a rejected call leaves the image and the counters untouched
    if (rgba8_dev && (R().flags & PT_FLAG_ACCUM_SHARD_ROWS))
        return fail(PT_ERR_INVALID, "pt_iterate: no PBO conversion from a row-sharded accumulator");
    int rc;
    if ((R().flags & PT_FLAG_TRACE_AHEAD) && count == 1) {
        // the reference's protocol, one call per iteration: the iteration comes out of a batch that was traced ahead
        if (!R().ahead.empty() && R().ahead.front().first + R().ahead.front().next != first_iter) {
            rc = discard_ahead();                    // not the iteration the parked batches continue with
            if (rc) return rc;
        }
        if (R().ahead.empty()) {
            rc = trace_ahead(first_iter);
            if (rc) return rc;
        }
        State::Parked &p = R().ahead.front();
        Slot &sl = R().slot[p.slot];
        if (!p.waited) {      // (once per batch: the commits that follow on the caller's stream are ordered behind this one)
            HIPCHECK(hipStreamWaitEvent(R().stream, sl.evDone, 0));
            p.waited = true;
        }
        rc = commit_range(sl, p.count, p.next, p.next + 1, false);
        if (rc) return rc;
        int after = p.first + p.count;               // first iteration behind the parked batches
        if (++p.next == p.count) {
            HIPCHECK(hipEventRecord(sl.evCommitted, R().stream));
            R().ahead.pop_front();
        }
        // every free slot traces on: the GPU stays ahead of the caller by at least a batch
        if (!R().ahead.empty()) after = R().ahead.back().first + R().ahead.back().count;
        while ((int)R().ahead.size() < R().nslots && after < kIterEnd) {
            rc = trace_ahead(after);
            if (rc) return rc;
            after = R().ahead.back().first + R().ahead.back().count;
        }
    } else {
        if (!R().ahead.empty()) {
            rc = discard_ahead();
            if (rc) return rc;
        }
        Slot &sl = R().slot[R().seq % R().nslots];
        rc = trace_batch(sl, first_iter, count);
        if (rc) return rc;
        // commit on the caller's stream: commits are therefore ordered like the pt_iterate calls
        HIPCHECK(hipStreamWaitEvent(R().stream, sl.evDone, 0));
        rc = commit_range(sl, count, 0, count, false);
        if (rc) return rc;
        HIPCHECK(hipEventRecord(sl.evCommitted, R().stream));
        R().seq += 1;
    }
    if (rgba8_dev) {
        hipLaunchKernelGGL(k_to_rgba8, dim3((R().P + kBlock - 1) / kBlock), dim3(kBlock), 0, R().stream, R().image, R().P,
                           first_iter + count - 1, reinterpret_cast<uchar4 *>(rgba8_dev));
        HIPCHECK(hipGetLastError());
    }
    R().iterations += count;
    return PT_OK;
}

int pt_iterate(int frame, int iter, void *rgba8_dev) { return pt_iterate_batch(frame, iter, 1, rgba8_dev); }

int pt_sync(void) {
    if (!R().init) {
        if (!scan_in_use()) return fail(PT_ERR_NOT_INIT, "pt_sync before pt_init");
        HIPCHECK(hipDeviceSynchronize());          // the scan library alone (the current device's calls)
        return PT_OK;
    }
    return check_device_fault();
}

int pt_readback(float *rgb_sum_host) {
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_readback before pt_init");
    if (!rgb_sum_host) return fail(PT_ERR_INVALID, "pt_readback: null");
    // every commit so far is already ordered before this copy on the caller's stream
    if (R().flags & PT_FLAG_ACCUM_SHARD_ROWS) {   // scatter this shard's rows into a zeroed full frame
        std::vector<float> rows((size_t)R().nLocal * 3);
        if (R().nLocal) HIPCHECK(hipMemcpyAsync(rows.data(), R().image, rows.size() * sizeof(float), hipMemcpyDeviceToHost, R().stream));
        HIPCHECK(hipStreamSynchronize(R().stream));
        memset(rgb_sum_host, 0, (size_t)R().P * 3 * sizeof(float));
        const size_t rowFloats = (size_t)R().prm.W * 3;
        for (int lr = 0; lr * R().prm.W < R().nLocal; ++lr)
            memcpy(rgb_sum_host + (size_t)(lr * R().prm.shardCount + R().prm.shardRank) * rowFloats, rows.data() + lr * rowFloats,
                   rowFloats * sizeof(float));
        return readback_fault();
    }
    // a plain copy: at PCIe rate into a buffer the caller has page-locked with pt_pin_host, through the runtime's
    // pageable staging path otherwise.  The library never registers memory it does not own on its own initiative.
    const size_t bytes = (size_t)R().P * 3 * sizeof(float);
    HIPCHECK(hipMemcpyAsync(rgb_sum_host, R().image, bytes, hipMemcpyDeviceToHost, R().stream));
    HIPCHECK(hipStreamSynchronize(R().stream));
    return readback_fault();
}

int pt_pin_host(void *host, size_t bytes) {
    if (!host || bytes == 0) return fail(PT_ERR_INVALID, "pt_pin_host: bad argument");
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (R().pinnedHost) {
        if (R().pinnedHost == host && R().pinnedBytes == bytes) return PT_OK;
        (void)hipHostUnregister(R().pinnedHost);
        R().pinnedHost = nullptr;
    }
    HIPCHECK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    R().pinnedHost = host;
    R().pinnedBytes = bytes;
    return PT_OK;
}

int pt_unpin_host(void) {
    if (R().pinnedHost) {
        (void)hipStreamSynchronize(R().stream);
        HIPCHECK(hipHostUnregister(R().pinnedHost));
        R().pinnedHost = nullptr;
        R().pinnedBytes = 0;
    }
    return PT_OK;
}

int pt_readback_rgba8(int iter, uint8_t *rgba_host) {
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_readback_rgba8 before pt_init");
    if (!rgba_host || iter < 1) return fail(PT_ERR_INVALID, "pt_readback_rgba8: bad argument");
    if (R().flags & PT_FLAG_ACCUM_SHARD_ROWS) return fail(PT_ERR_INVALID, "pt_readback_rgba8: accumulator is row-sharded");
    DevBuf<uchar4> tmp;
    int rc = tmp.alloc(R().P);
    if (rc) return rc;
    hipLaunchKernelGGL(k_to_rgba8, dim3((R().P + kBlock - 1) / kBlock), dim3(kBlock), 0, R().stream, R().image, R().P, iter, tmp.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(rgba_host, tmp.p, (size_t)R().P * 4, hipMemcpyDeviceToHost, R().stream));
    HIPCHECK(hipStreamSynchronize(R().stream));
    return readback_fault();
}

int pt_counters(PtCounters *out) {
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_counters before pt_init");
    if (!out) return fail(PT_ERR_INVALID, "pt_counters: null");
    int rc = sync_all();
    if (rc) return rc;
    rc = resolve_events(R().evBounce, R().msBounce, R().nBounce);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    std::unique_ptr<Ctrl> hp(new Ctrl);   // 2 MB: off the stack, and not shared by the host threads that drive other contexts
    Ctrl &h = *hp;
    uint32_t faultBits = 0;
    for (int i = 0; i < R().nslots; ++i) {
        HIPCHECK(hipMemcpy(&h, R().slot[i].ctrl, sizeof h, hipMemcpyDeviceToHost));
        for (int d = 0; d < kMaxDepthSlots; ++d) {
            int64_t early = 0;
            for (int sg = 0; sg < kTallyShards; ++sg) early += (int64_t)h.early[d][sg][0];
            out->live[d] += (int64_t)h.sum_live[d] + early;   // they did enter bounce d
            out->ended_early[d] += early;                      // ... without being moved through memory
        }
        for (int sg = 0; sg < kTallyShards; ++sg) {
            out->light_hits += (int64_t)h.light_hits[sg][0];
            out->misses += (int64_t)h.misses[sg][0];
        }
        faultBits |= h.error;
    }
    out->iterations = R().iterations;
    out->bounce_launches = R().nBounce;
    out->bounce_kernel_ms = R().msBounce;
    out->raygen_kernel_ms = 0.0;   // camera rays are generated inside the first bounce launch
    out->raygen_launches = 0;
    if (faultBits) return fail(PT_ERR_DEVICE, "device fault 0x%x:%s%s (results of this render are void; re-init)", faultBits,
                               (faultBits & kFaultPoolExhausted) ? " path pool exhausted" : "", (faultBits & kFaultReserveTimeout) ? " chunk reservation timed out" : "");
    return PT_OK;
}

int pt_counters_reset(void) {
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_counters_reset before pt_init");
    int rc = sync_all();
    if (rc) return rc;
    rc = resolve_events(R().evBounce, R().msBounce, R().nBounce);
    if (rc) return rc;
    R().msBounce = 0;
    R().nBounce = 0;
    R().iterations = 0;
    for (int i = 0; i < R().nslots; ++i) {
        // the sticky fault word survives a counter reset
        uint32_t err = 0;
        HIPCHECK(hipMemcpy(&err, &R().slot[i].ctrl->error, sizeof err, hipMemcpyDeviceToHost));
        rc = reset_ctrl(R().slot[i].ctrl, nullptr);
        if (rc) return rc;
        R().slot[i].parity = 0;
        if (err) HIPCHECK(hipMemcpy(&R().slot[i].ctrl->error, &err, sizeof err, hipMemcpyHostToDevice));
    }
    return PT_OK;
}

// ---- stream compaction library -------------------------------------------------------------------------
int pt_scan_exclusive_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, void *stream) {
    if (n < 0 || (n > 0 && (!in_dev || !out_dev))) return fail(PT_ERR_INVALID, "pt_scan_exclusive_i32: bad argument");
    if (n == 0) return PT_OK;
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (n > (1ll << 42)) return fail(PT_ERR_INVALID, "pt_scan_exclusive_i32: n too large");
    hipStream_t st = (hipStream_t)stream;
    register_exit_handler();
    std::lock_guard<std::mutex> lock(g_scanMutex);   // (look-up AND launches: see scan_ws)
    ScanWs *wp = nullptr;
    int rc = scan_ws(st, &wp);
    if (rc) return rc;
    long long per;
    int chunks;
    scan_chunks(n, &per, &chunks);
    // Two launches (12 bytes of HBM traffic per element).  PT_AMD_SCAN=1: the ONE-launch form (k_scan_chained: ticketed chunks, chained
    // prefix; 8 bytes per element when a chunk's second read comes out of the caches) -- measured SLOWER on MI355X, 0.214 against 0.172 ms at
    // 2^26 (0.81 against 0.66 at 2^28): the 2048 resident workgroups' chunks (128 KB each) do not survive in the 4 MB L2 of an XCD between
    // their two reads, so it moves the same 12 bytes and adds the ticket and the wait; kept selectable, not the default.
    const char *mode = getenv("PT_AMD_SCAN");
    if (!(mode && atoi(mode) == 1)) {
        // (chunks of four tiles or more take the kernels that load a tile ahead; the apply adds up the totals before its chunk itself)
        if (per >= 4) hipLaunchKernelGGL((k_scan_reduce<false, true>), dim3(chunks), dim3(kBlock), 0, st, in_dev, (long long)n, per, wp->partial);
        else hipLaunchKernelGGL((k_scan_reduce<false, false>), dim3(chunks), dim3(kBlock), 0, st, in_dev, (long long)n, per, wp->partial);
        if (per >= 4) hipLaunchKernelGGL((k_scan_apply<true>), dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->partial);
        else hipLaunchKernelGGL((k_scan_apply<false>), dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->partial);
    } else {
        if (++wp->gen == 0u) ++wp->gen;
        HIPCHECK(hipMemsetAsync(wp->chained, 0, sizeof(uint32_t), st));          // the ticket (the sums are tagged with the call's generation)
        hipLaunchKernelGGL(k_scan_chained, dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->chained, wp->gen);
    }
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

int pt_compact_nonzero_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, int64_t *count_dev, void *stream) {
    if (n < 0 || !count_dev || (n > 0 && (!in_dev || !out_dev))) return fail(PT_ERR_INVALID, "pt_compact_nonzero_i32: bad argument");
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (n > 0xffffffffll) return fail(PT_ERR_INVALID, "pt_compact_nonzero_i32: n too large (positions are 32-bit)");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        HIPCHECK(hipMemsetAsync(count_dev, 0, sizeof(int64_t), st));
        return PT_OK;
    }
    register_exit_handler();
    std::lock_guard<std::mutex> lock(g_scanMutex);   // (look-up AND launches: see scan_ws)
    ScanWs *wp = nullptr;
    int rc = scan_ws(st, &wp);
    if (rc) return rc;
    long long per;
    int chunks;
    scan_chunks(n, &per, &chunks);
    if (per >= 4) hipLaunchKernelGGL((k_scan_reduce<true, true>), dim3(chunks), dim3(kBlock), 0, st, in_dev, (long long)n, per, wp->partial);
    else hipLaunchKernelGGL((k_scan_reduce<true, false>), dim3(chunks), dim3(kBlock), 0, st, in_dev, (long long)n, per, wp->partial);
    if (per >= 4)
        hipLaunchKernelGGL((k_compact_apply<true>), dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->partial, reinterpret_cast<long long *>(count_dev));
    else
        hipLaunchKernelGGL((k_compact_apply<false>), dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->partial, reinterpret_cast<long long *>(count_dev));
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

#include "pt_group.h"

#ifdef PT_TEST_API
#include "pt_test_api.h"
#endif  // PT_TEST_API

}  // extern "C"
#pragma GCC visibility pop
