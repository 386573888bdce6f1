// pt_mesh_walk.h -- the walks of the triangle meshes' hierarchies, AHEAD of the bounce that uses their results (round 5).
// Header-only part of the single translation unit pt_api.hip (namespace ptk).
//
// Rounds 2-4 walked a mesh inside k_bounce's loop over a tile's primitives, every lane its own ray.  The walks of a wave are of very
// unequal length -- on scenes/cornell_mesh.txt a ray takes 13 inner steps on average, the longest of a wave's 53 rays 69 -- and a wave
// steps until its last lane is through: ten lanes of 64 were at work in an average step (profiles/r05_mesh_walk_experiments.txt), and a tile cannot
// take new rays in, because everything behind the walk (shading, scatter, compaction) needs the whole tile's hits.  Handing parts of a
// long walk to the wave's idle lanes was built and measured (bit-identical, 2 % slower: the rounds of a walk are as long as before).
//
// So the walks left the tile.  k_mesh_walk runs BEFORE the bounce launch over the same queue: persistent WAVES, each on its own, draw
// quarter tiles (tickets), test their rays against the bounding balls of the meshes its class lists (the camera-ray bounce: its row's),
// and queue a JOB (path, mesh) in LDS for every ray that may hit; a lane that is through with a walk takes the next job, whatever
// tile it came from.  A job's result is folded into the path's record with one 64-bit atomic minimum,
//     meshHit[path] = bits of the world distance << 32 | winning triangle's unit << 1 | front side        (all ones: no mesh is hit),
// which IS the rule of the bounce's loop over the meshes -- nearest first, of two equally near the one earlier in the list: the lists
// are in file order, and so are the meshes' units in the record array.  The bounce then evaluates the winner alone (ptd::meshWinner).
// Same rays, same tests (ptd::meshPlanesPass / meshBoxPass / meshTriangle), same winner: the frame is the one of rounds 2-4, bit for bit.
#pragma once
#include "pt_trace.h"

namespace ptk {

constexpr int kWalkQueue = 128;          // jobs a wave holds at most (8 bytes each)
#ifndef PT_WALK_IDLE_MIN
#define PT_WALK_IDLE_MIN 24
#endif
#ifndef PT_WALK_LEAF_MIN
#define PT_WALK_LEAF_MIN 16
#endif
#ifndef PT_WALK_TWO
#define PT_WALK_TWO 1
#endif
#ifndef PT_WALK_QUARTERS_MIN
#define PT_WALK_QUARTERS_MIN 8
#endif
constexpr int kWalkQuartersMin = PT_WALK_QUARTERS_MIN;   // quarter tiles (64 paths each) per wave of a launch, at least (see the tickets)
constexpr int kWalkIdleMin = PT_WALK_IDLE_MIN;   // idle lanes a wave counts before it hands out jobs (and queues more)
constexpr int kWalkLeafMin = PT_WALK_LEAF_MIN;   // lanes that hold a triangle before the wave tests triangles
constexpr int kWalkLdsFixedWords = kSeg + (kSeg + 2) + 2 * kWalkQueue * kWaves;     // segment counts and prefix, the waves' queues
static_assert((kSeg + kSeg + 2) % 2 == 0, "the queues hold 64-bit jobs");
// what a lane needs of the mesh whose hierarchy it walks, staged in LDS by every workgroup (a gather of it from the primitives' 448-byte
// records, per lane and job, was a fifth of the kernel's vector-memory instructions: profiles/r05_mesh_walk_experiments.txt, 2h)
struct WalkMesh {
    float inv[12], invZ[3];
    uint32_t root;
    float xf[12], camObj[3];
    uint32_t stride;
};
static_assert(sizeof(WalkMesh) == 128, "eight 16-byte words");
constexpr int kWalkMeshWords = (int)(sizeof(WalkMesh) / sizeof(uint32_t));
constexpr int kWalkMeshLdsMax = 32;      // meshes whose rows go to LDS (4 KB); a scene with more reads them from global memory, per job
// dynamic LDS of a launch: the fixed part, the table of the scene's meshes, then the lanes' stacks of waiting far children, [levels][kBlock] words
inline size_t walkLdsBytes(int levels, int nmeshes) {
    // (+ 2 levels: the sentinel below a lane's first entry, and the slot above a full stack, which an inner step writes whether it keeps it or not)
    return ((size_t)kWalkLdsFixedWords + (size_t)nmeshes * kWalkMeshWords + (size_t)((levels < 1 ? 1 : levels) + 2) * kBlock) * sizeof(uint32_t);
}

// ROWS_LDS: the per-mesh rows (WalkMesh) are staged in LDS -- scenes of at most kWalkMeshLdsMax meshes; else every job reads its mesh's row from
// global memory (an instantiation of its own: both forms in one kernel cost the common one its registers)
// (the pinhole camera-ray walk carries the pixel's coordinates through its queueing and needs a 65th register: seven workgroups per CU for it --
// with eight it parked the triangle test's `front` flag in scratch inside the leaf step; it is a twentieth of a mesh scene's GPU time)
template <bool FIRST, bool DOF, bool ROWS_LDS>
__global__ __launch_bounds__(kBlock, (FIRST && !DOF) ? 7 : 8) void k_mesh_walk(BounceArgs A) {
    static_assert(FIRST || !DOF, "the lens only concerns the camera rays");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CLS = kClsMax, SUB = kSeg / CLS;             // (scenes with meshes bin by two candidate bits: 32 classes)
    uint32_t *const s_segcnt = reinterpret_cast<uint32_t *>(smem);      // [kSeg]   paths per input segment (0: a class that lists no mesh)
    uint32_t *const s_segpre = s_segcnt + kSeg;                          // [kSeg+2] tile prefix per input segment
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long *const s_queue = reinterpret_cast<unsigned long long *>(s_segpre + kSeg + 2) + wave * kWalkQueue;
    // (a job names its mesh by its ordinal among the scene's meshes, GeomDev::frameSlot = its row of the table: the host's copy
    // A.walkMeshRows, staged here when it is small)
    const int nMeshes = ROWS_LDS ? A.walkMeshLds : 0;
    WalkMesh *const s_mesh = reinterpret_cast<WalkMesh *>(reinterpret_cast<uint32_t *>(smem) + kWalkLdsFixedWords);
    const WalkMesh *const g_mesh = reinterpret_cast<const WalkMesh *>(A.walkMeshRows);
    auto withMesh = [&](uint32_t g, auto &&f) {                 // f(the row of mesh g)
        if (ROWS_LDS) f(s_mesh[g]);
        else f(g_mesh[g]);
    };
    // (a lane's stack: slot 0 holds kDone for good -- popping an empty stack yields "through" without a test --, its entries start at slot 1)
    uint32_t *const stack = reinterpret_cast<uint32_t *>(smem) + kWalkLdsFixedWords + nMeshes * kWalkMeshWords + threadIdx.x + kBlock;
    stack[-kBlock] = 0xffffffffu;
    for (int i = threadIdx.x; i < nMeshes * kWalkMeshWords; i += kBlock) reinterpret_cast<uint32_t *>(s_mesh)[i] = reinterpret_cast<const uint32_t *>(g_mesh)[i];
    __syncthreads();
    const KParams &prm = A.prm;
    Ctrl *const ctrl = A.ctrl;
    const int depth = A.depth, parity = A.parity;

    // the tiles of this launch (as k_bounce's prologue counts them), without those of the classes that list no mesh
    uint32_t numTiles;
    if (FIRST) {
        numTiles = (uint32_t)prm.nLocalPad / kBlock * (uint32_t)A.batch;
    } else {
        const bool skipNonCand = A.tile.skipNonCandidates != 0u;
        if (threadIdx.x < 64) {
            constexpr int kPerLane = (kSeg + 63) / 64;
            uint32_t cs[kPerLane], ts[kPerLane];
            uint32_t inc = 0u;
#pragma unroll
            for (int q = 0; q < kPerLane; ++q) {
                const int sgi = (int)threadIdx.x * kPerLane + q;
                const int cls = sgi / SUB;
                cs[q] = sgi < kSeg ? ctrl->pos[parity][depth][sgi][0] : 0u;
                if (sgi >= kSeg || A.walkClassOff[cls + 1] == A.walkClassOff[cls]) cs[q] = 0u;
                if (skipNonCand && ((uint32_t)cls & 24u) == 0u) cs[q] = 0u;      // (the bounce skips these tiles: k_bounce, kEmitBits)
                ts[q] = (cs[q] + kBlock - 1) / kBlock;
                inc += ts[q];
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(inc, o, 64);
                if ((int)threadIdx.x >= o) inc += up;
            }
            uint32_t run = inc;
#pragma unroll
            for (int q = kPerLane - 1; q >= 0; --q) {
                const int sgi = (int)threadIdx.x * kPerLane + q;
                if (sgi < kSeg) { s_segcnt[sgi] = cs[q]; s_segpre[sgi + 1] = run; }
                run -= ts[q];
            }
            if (threadIdx.x == 0) s_segpre[0] = 0;
        }
        __syncthreads();
        numTiles = s_segpre[kSeg];
    }
    // (re-arm the tickets of the slot's next batch; nobody draws from the other parity's now)
    if (blockIdx.x == 0 && threadIdx.x < kTicketShards) ctrl->walkTicket[parity ^ 1][depth][threadIdx.x][0] = 0u;

    // ---- tickets: wave w of the launch draws the QUARTER tiles t n + s (quarter q of tile T: 4 T + q) of shard s = w % n,
    // n = min(kTicketShards, waves), in increasing order -- a quarter, ~50 jobs, is a tenth of a wave's share of a late bounce
    // (a wave wants kWalkQuartersMin quarters or more to its name -- with fewer, most of its time is the tail of its last jobs: the late bounces of
    // a batch hold a few jobs per LANE of a full grid -- so the waves beyond that many end here; at least one per CU-sized group stays)
    uint32_t nWaves = gridDim.x * kWaves;
    {
        const uint32_t want = 4u * numTiles / (uint32_t)kWalkQuartersMin;      // (camera rays: far more tiles than waves)
        const uint32_t floor = nWaves < 256u ? nWaves : 256u;
        const uint32_t cap = want > floor ? want : floor;
        if (cap < nWaves) nWaves = cap;
        if (blockIdx.x * kWaves + wave >= nWaves) return;
    }
    const uint32_t nShards = nWaves < (uint32_t)kTicketShards ? nWaves : (uint32_t)kTicketShards;
    const uint32_t shard = (blockIdx.x * kWaves + wave) % nShards;
    uint32_t *const ticket = &ctrl->walkTicket[parity][depth][shard][0];
    // (later bounces: a ticket is a QUARTER tile, 4 T + q -- a wave's share of a late bounce is a handful of tiles; camera rays: a whole tile --
    // a batch holds hundreds of thousands of them, most outside the meshes' rows and spans, and a ticket per quarter was 14 000 atomics per word)
    constexpr uint32_t kPerTile = FIRST ? 1u : 4u;
    bool exhausted = false;
    auto nextTile = [&]() -> uint32_t {
        uint32_t t = 0u;
        if (lane == 0) t = atomicAdd(ticket, 1u);
        const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) * nShards + shard;
        return T < kPerTile * numTiles ? T : 0xffffffffu;
    };

    // ---- the ray of record index i (a path's slot in the input pool; camera rays: i = 256 tile + lane in the padded pixel space)
    auto fetchRay = [&](uint32_t i, F3 &org, F3 &dir) {
        if (FIRST) {
            const uint32_t itb = fastDiv(i, prm.magicN, prm.shiftN);
            const uint32_t j = i - itb * (uint32_t)prm.nLocalPad;
            const int lr = (int)fastDiv(j, prm.magicWp, prm.shiftWp);
            const int sr0 = prm.sceneRect[0];
            const int x = ((int)j - lr * prm.Wp) + ((sr0 > 0 ? sr0 : 0) & ~(kBlock - 1));
            const int y = lr * prm.shardCount + prm.firstY0;
            cameraRayAt(prm, iterationHash(A.iter + (int)itb, 0), x + y * prm.W, x, y, org, dir);
        } else {
            const float4 a = *reinterpret_cast<const float4 *>(A.in.arrA(i));
            const float4 b = *reinterpret_cast<const float4 *>(A.in.arrB(i));
            org = f3(a.x, a.y, a.z);
            dir = f3(a.w, b.x, b.y);
        }
    };

    // ---- queueing: the quarter tile being turned into jobs, a mesh of its list at a time, so that the queue never has to take more
    // than 64 jobs in one go
    uint32_t curT = 0xffffffffu, curQ = 0u, curK = 0u;
    uint32_t qn = 0u;                                           // jobs in the queue
    auto refill = [&]() {
        while (qn + 64u <= (uint32_t)kWalkQueue) {
            probeCount(27, true);
            if (curT == 0xffffffffu) {
                if (exhausted) break;
                const uint32_t t4 = nextTile();
                if (t4 == 0xffffffffu) { exhausted = true; break; }
                curT = FIRST ? t4 : t4 >> 2; curQ = FIRST ? 0u : t4 & 3u; curK = 0u;
            }
            // the quarter's rays and its list of meshes
            bool valid;
            uint32_t idx;
            int px = 0, py = 0;
            int l0, l1;
            bool rows = false;
            if (FIRST) {
                const uint32_t idx0 = curT * kBlock;
                const uint32_t itb0 = fastDiv(idx0, prm.magicN, prm.shiftN);
                const uint32_t j0 = idx0 - itb0 * (uint32_t)prm.nLocalPad;
                const int lr0 = (int)fastDiv(j0, prm.magicWp, prm.shiftWp);
                const int sr0 = prm.sceneRect[0], sr1 = prm.sceneRect[1], sr2 = prm.sceneRect[2], sr3 = prm.sceneRect[3];
                const int x0 = ((int)j0 - lr0 * prm.Wp) + ((sr0 > 0 ? sr0 : 0) & ~(kBlock - 1));
                const int y0 = lr0 * prm.shardCount + prm.firstY0;
                if ((y0 < sr1) | (y0 > sr3) | (x0 + (kBlock - 1) < sr0) | (x0 > sr2)) {      // (the bounce skips the tile as a whole)
                    curT = 0xffffffffu;
                    continue;
                }
                px = x0 + (int)(curQ * 64u + lane);
                py = y0;
                idx = idx0 + curQ * 64u + lane;
                valid = px < prm.W && px >= sr0 && px <= sr2;        // (inScene: the rows were tested above)
                if (!DOF && A.walkRowOff != nullptr) {
                    rows = true;
                    l0 = A.walkRowOff[y0]; l1 = A.walkRowOff[y0 + 1];
                } else {
                    l0 = A.walkAll0; l1 = A.walkAll1;
                }
            } else {
                uint32_t cnt = 0u;
#pragma unroll
                for (int q = 0; q < kSeg / 64; ++q) cnt += (uint32_t)__popcll(__ballot(s_segpre[1 + 64 * q + lane] <= curT));
                const uint32_t sg = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
                const uint32_t segFirst = s_segpre[sg];
                const uint32_t local = (curT - segFirst) * kBlock + curQ * 64u + lane;
                valid = local < s_segcnt[sg];
                const uint32_t shift = (uint32_t)prm.chunkShift, poolChunks = (uint32_t)prm.poolChunks;
                const uint32_t j = ((curT - segFirst) * kBlock) >> shift;
                uint32_t chunk = 1u + sg;
                if (j != 0u) {
                    const unsigned long long e = j < poolChunks ? A.in.list[(size_t)sg * poolChunks + j] : 0ull;
                    chunk = (uint32_t)(e >> 32) == A.genIn ? (uint32_t)e : 0u;
                    chunk = chunk < poolChunks ? chunk : 0u;
                }
                idx = (chunk << shift) + (local - (j << shift));
                const uint32_t cls = sg / SUB;
                l0 = A.walkClassOff[cls]; l1 = A.walkClassOff[cls + 1];
            }
            if (l1 - l0 == 0) {                                  // (camera rays: a row no mesh reaches -- the bounce asks for no record)
                curT = 0xffffffffu;
                continue;
            }
            // camera rays: only the pixels inside a mesh's span (its rectangle) can reach it -- the bounce asks for the records of those alone --, and a
            // quarter that holds none of them is done (a mesh covers a fraction of its rows: most of their rays were built for nothing)
            bool reach = valid;
            if (FIRST && !DOF) {
                reach = false;
                for (int k = 0; k < l1 - l0; ++k) {
                    if (rows) {
                        const int span = A.walkIdx[2 * (l0 + k) + 1];
                        reach = reach | ((px >= (span & 0xffff)) & (px <= (span >> 16)));
                    } else {
                        const GeomDev &G = A.ggeoms[A.walkIdx[l0 + k]];
                        reach = reach | ((px >= G.rect[0]) & (px <= G.rect[2]) & (py >= G.rect[1]) & (py <= G.rect[3]));
                    }
                }
                reach = reach & valid;
                if (__ballot(reach) == 0ull) {
                    curK = 0u;
                    if (++curQ == 4u) curT = 0xffffffffu;
                    continue;
                }
            }
            F3 org = f3(0, 0, 0), dir = f3(0, 0, 1);
            if (reach) fetchRay(idx, org, dir);
            if (curK == 0u) {
                // no mesh is hit, unless a walk says otherwise; the store is COMPLETE before this wave issues an atomic on the word
                if (FIRST || valid) A.meshHit[idx] = ~0ull;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const float dd = dot(dir, dir);
            bool full = false;
            for (int k = (int)curK; k < l1 - l0; ++k) {
                if (qn + 64u > (uint32_t)kWalkQueue) {           // (no room for a mesh's worth of jobs: go on from here next time)
                    curK = (uint32_t)k;
                    full = true;
                    break;
                }
                int g, span = 0;
                if (rows) { g = A.walkIdx[2 * (l0 + k)]; span = A.walkIdx[2 * (l0 + k) + 1]; }
                else g = A.walkIdx[l0 + k];
                const GeomDev &G = A.ggeoms[g];
                bool want = reach;
                if (FIRST && !DOF) {
                    if (rows) want = want & (px >= (span & 0xffff)) & (px <= (span >> 16));
                    else want = want & (px >= G.rect[0]) & (px <= G.rect[2]) & (py >= G.rect[1]) & (py <= G.rect[3]);
                }
                want = want && !certainMiss(G, org, dir, dd);
                probeCount(28, want);
                const unsigned long long b = __ballot(want);
                if (b != 0ull) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                    // (bit 63: the tile lists this ONE mesh -- its record has no rival to be compared with: see the fold)
                    if (want) s_queue[qn + rank] = ((unsigned long long)(((uint32_t)G.frameSlot & 0x7fffu) | (l1 - l0 == 1 ? 0x80000000u : 0u)) << 32) | idx;
                    qn += (uint32_t)__popcll(b);
                }
            }
            if (full) break;
            curK = 0u;
            if (!FIRST || ++curQ == 4u) curT = 0xffffffffu;      // (camera rays: the tile's next quarter)
        }
    };

    // ---- the walks: a lane's state
    constexpr uint32_t kDone = 0xffffffffu;                     // (reads as a triangle's ref: the loop over inner nodes stops on it)
    // (a lane is idle exactly when ref == kDone; one that is idle WITH a job -- jobIdx != kNoJob -- is through with it and its result not
    // folded in yet: no flags beside the two words, and nothing to update at a turn's end)
    constexpr uint32_t kNoJob = 0xffffffffu;
    uint32_t jobIdx = kNoJob, jobGeom = 0u;
    F3 ro = f3(0, 0, 0), rd = f3(0, 0, 1), inv = f3(1, 1, 1), rc = f3(0, 0, 0);
    uint32_t keyT = 0xffffffffu, keyI = 0xffffffffu;            // the best hit so far: bits of t + 0.0f, unit << 1 | front (all ones: none; t then reads as a NaN)
    uint32_t ref = kDone;
    uint32_t *sp = stack;
    auto pop = [&]() -> uint32_t {                              // (the sentinel below the first entry: an empty stack pops kDone, and the walk ends there)
        sp -= kBlock;
        return *sp;
    };
    const float4 *const recs = A.meshRecs;
    // ONE loop, three kinds of step; the wave votes on the next one:
    //   hand-out   when kWalkIdleMin lanes stand idle and there is work to give them (or nobody is at work at all): what the lanes that
    //              are through have found goes into their paths' records, the queue is topped up, the idle lanes take jobs;
    //   triangles  when kWalkLeafMin lanes hold one (or no lane holds an inner node): the triangle's own box, then the triangle;
    //   inner node otherwise: the lanes that hold one test its two children.
    // (Rounds 3-4 -- and this kernel's first version -- ran "inner nodes until EVERY lane holds a triangle, then the triangles": a round was
    // as long as the longest descent of its lanes, 13 steps where a lane's own took 4.7 -- profiles/r05_mesh_walk_experiments.txt.)
    for (;;) {
        probe(40);                                              // (marks of the ISA listing: the vote)
        // (two ballots: kDone reads as a triangle's ref, so "at an inner node" is the leaf bit alone, and the lanes at a triangle are the rest --
        // every lane of the wave is in this loop)
        const bool busy = ref != kDone;
        const bool atInner = !(ref & kMeshLeaf), atLeaf = busy && !atInner;
        const unsigned long long idleMask = __ballot(!busy);
        const uint32_t nInner = (uint32_t)__popcll(__ballot(atInner)), nIdle = (uint32_t)__popcll(idleMask);
        const uint32_t nLeaf = 64u - nInner - nIdle;
        const bool more = qn != 0u || !exhausted;
        probeCount(29, atInner); probeCount(30, atLeaf); probeCount(31, !busy); probeCount(23, !more);
        if ((more && nIdle >= (uint32_t)kWalkIdleMin) || nInner + nLeaf == 0u) {
            probe(41);                                          // (hand-out: fold)
            // what the lanes that are through have found: the winner's distance in the world, and into the path's record with it.
            // (t: the bits of t + 0.0f serve -- a winner at -0 differs from +0 in the signs of zeros of P alone, which the length squares away)
            const bool pending = !busy && jobIdx != kNoJob;
            probeCount(21, pending);                            // (instrumented build: the walk's wave steps and their lanes)
            if (pending) {
                if (keyT != 0xffffffffu && (jobGeom & 0x80000000u) != 0u) {
                    // the tile lists one mesh: nothing to take a minimum with -- the bounce evaluates the winner's distance itself (meshWinner: the
                    // same operations on the same operands) and drops a hit at distance <= 0 exactly as the comparison below would have.  A plain
                    // store of the winner in place of the ray's re-read, the point, its transform, the length and the atomic.
                    A.meshHit[jobIdx] = (unsigned long long)keyI;
                } else if (keyT != 0xffffffffu) {
                    // (the ray's origin in the world: the eye for pinhole camera rays, else the first of the path's two 16-byte words)
                    F3 org, dir;
                    if (FIRST && !DOF) org = f3(prm.pos[0], prm.pos[1], prm.pos[2]);
                    else if (FIRST) fetchRay(jobIdx, org, dir);
                    else {
                        const float4 a = *reinterpret_cast<const float4 *>(A.in.arrA(jobIdx));
                        org = f3(a.x, a.y, a.z);
                    }
                    withMesh(jobGeom & 0x7fffu, [&](const WalkMesh &G) {
                        const F3 P = mulMV(G.xf, getPointOnRay(ro, rd, __uint_as_float(keyT)), 1.0f);
                        const float t = length(org - P);
                        if (t > 0.0f)
                            __hip_atomic_fetch_min(A.meshHit + jobIdx, ((unsigned long long)__float_as_uint(t) << 32) | keyI, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    });
                }
                jobIdx = kNoJob;
            }
            probe(42);                                          // (queueing)
            if (qn < nIdle) refill();
            probe(43);                                          // (jobs to lanes)
            // the idle lanes take the youngest jobs
            const uint32_t take = nIdle < qn ? nIdle : qn;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0u));
            probeCount(26, !busy && rank < take);
            if (!busy && rank < take) {
                const unsigned long long job = s_queue[qn - 1u - rank];
                jobIdx = (uint32_t)job;
                jobGeom = (uint32_t)(job >> 32);
                F3 org, dir;
                fetchRay(jobIdx, org, dir);
                withMesh(jobGeom & 0x7fffu, [&](const WalkMesh &G) {
                    ro = (FIRST && !DOF) ? f3(G.camObj[0], G.camObj[1], G.camObj[2]) : mulMV(G.inv, org, 1.0f);
                    rd = normalize(mulMV0(G.inv, G.invZ, dir));
                    inv = f3(guardedReciprocal(rd.x), guardedReciprocal(rd.y), guardedReciprocal(rd.z));
                    rc = f3(-(ro.x * inv.x), -(ro.y * inv.y), -(ro.z * inv.z));
                    const uint32_t octant = (__float_as_uint(inv.x) >> 31) | ((__float_as_uint(inv.y) >> 31) << 1) | ((__float_as_uint(inv.z) >> 31) << 2);
                    ref = G.root + octant * G.stride;
                });
                sp = stack;
                keyT = keyI = 0xffffffffu;
            }
            qn -= take;
            if (take == 0u && nInner + nLeaf == 0u) break;      // (no job left anywhere: the queue is empty and the tiles are drawn)
        } else if (nLeaf >= (uint32_t)kWalkLeafMin || nInner == 0u) {
            probe(44);                                          // (triangles)
            probeCount(24, atLeaf);
            if (atLeaf) {
                const float4 *r = recs + (size_t)(ref & ~kMeshLeaf);
                const float4 q0 = r[0], q1 = r[1];
                const float2 q2 = *reinterpret_cast<const float2 *>(r + 2);
                const F3 v0 = f3(q0.x, q0.y, q0.z), v1 = f3(q0.w, q1.x, q1.y), v2 = f3(q1.z, q1.w, q2.x);
                const float m = q2.y;
                // the triangle's box as pt_mesh.h states it: min / max of the vertices, moved outwards by the mesh's margin
                const F3 lo = f3(__builtin_fminf(__builtin_fminf(v0.x, v1.x), v2.x) - m, __builtin_fminf(__builtin_fminf(v0.y, v1.y), v2.y) - m,
                                 __builtin_fminf(__builtin_fminf(v0.z, v1.z), v2.z) - m);
                const F3 hi = f3(__builtin_fmaxf(__builtin_fmaxf(v0.x, v1.x), v2.x) + m, __builtin_fmaxf(__builtin_fmaxf(v0.y, v1.y), v2.y) + m,
                                 __builtin_fmaxf(__builtin_fmaxf(v0.z, v1.z), v2.z) + m);
                float tmin;
                if (meshBoxPass(lo, hi, inv, rc, true, __uint_as_float(keyT), tmin)) {
                    float t;
                    bool front;
                    if (meshTriangle(ro, rd, v0, v1 - v0, v2 - v0, t, front)) {
                        // (accepted t are >= 0: the order of the bits of t + 0.0f is the order of the values, -0 = +0 included)
                        const uint32_t kt = __float_as_uint(t + 0.0f), ki = ((ref & ~kMeshLeaf) << 1) | (front ? 1u : 0u);
                        if ((t >= tmin) & ((kt < keyT) | ((kt == keyT) & (ki < keyI)))) {
                            keyT = kt;
                            keyI = ki;
                        }
                    }
                }
                ref = pop();
            }
        } else {
            probe(45);                                          // (inner node)
            probeCount(22, atInner);
            if (atInner) {
                const float4 *r = recs + (size_t)ref;
                const float4 q0 = r[0], q1 = r[1];
#if PT_WALK_TWO
                // (the record BEHIND the node -- in the hierarchy's depth-first order its first inner child -- comes in the same round trip: a lane
                // that goes there next takes two levels in one turn.  +2.6 % on the mesh scene; a third record: -10 %.  Inside the tile, where
                // the texture addresser was the bound, the same idea cost 7 %: profiles/r04_mesh_walk_experiments.txt)
                const float4 q2 = r[2], q3 = r[3];
                const uint32_t behind = ref + (uint32_t)kMeshNodeUnits;
#endif
                const float tb = __uint_as_float(keyT);
                auto step = [&](float4 a, float4 b) {
                    const bool passN = meshPlanesPass(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), inv, rc, true, tb);
                    const bool passF = meshPlanesPass(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), inv, rc, true, tb);
                    const uint32_t refN = __float_as_uint(a.w), refF = __float_as_uint(b.w);
                    // (the far child goes to the slot above the top whether it waits there or not -- the slot is free --, and the top moves when
                    // both pass: no branch around the store)
                    *sp = refF;
                    sp += (passN & passF) ? kBlock : 0;
                    ref = passN ? refN : (passF ? refF : pop());
                };
                step(q0, q1);
#if PT_WALK_TWO
                if (ref == behind) step(q2, q3);
#endif
            }
        }
        probe(46);                                              // (the turn's end)
    }
}

}  // namespace ptk
