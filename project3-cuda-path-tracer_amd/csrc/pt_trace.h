// pt_trace.h -- render kernels of the MI355X path tracer (gfx950): fused per-bounce kernel, ordered radiance
// commit, PBO conversion, and the device-side control block / path-state layout they share.
// Header-only part of the single translation unit pt_api.hip (namespace ptk).
#pragma once
#include "../../include/pt_amd.h"
#include <type_traits>

#include "pt_common.h"
#include "pt_device.h"

namespace ptk {
using namespace ptd;

constexpr int kNumArrays = 11;       // PathSegment: origin3, dir3, throughput3, the pixel's hash, pixelIndex | batch index = 11 dwords = 44 bytes per path
constexpr int kMaxDepthSlots = PT_MAX_DEPTH + 2;

// ---- device control block ------------------------------------------------------------------------
constexpr int kOct = 8;              // direction octants: paths are binned by the signs of their new direction ...
constexpr int kCls = 2 * kOct;       // ... and by "may hit a small primitive" (bit 3): 16 classes
constexpr int kSub = 8;              // append-counter shards per class (workgroup blockIdx % kSub)
constexpr int kSeg = kCls * kSub;    // path buffers are split into kSeg segments with one append counter each
// Scenes with triangle meshes bin by TWO candidate bits instead of one (class bits 3 and 4: "may hit a binned primitive of group 0 /
// of group 1" -- pt_init puts the costliest mesh alone into group 1): 32 classes of kSeg / 32 shards each in the same kSeg segments.
// A tile then walks only the mesh its paths can hit instead of every mesh one after the other with part of its lanes (round 4).
constexpr int kClsMax = 2 * kCls;
static_assert(kSeg % 64 == 0, "setupTile compares kSeg / 64 prefix entries per lane");
constexpr int kBinMax = 4;           // at most this many small primitives take part in the binning
constexpr int kEmitMax = 8;          // emissive primitives the direct-lighting bounce chooses from
constexpr int kWallMax = 6;          // large cubes ("walls") whose world-space boxes class the survivors (wallCertainMiss)
// Path state lives in POOLS of fixed-size chunks (2^chunkShift paths each, chosen by pt_init): a segment of a bounce's
// queue is a list of chunks, handed out by an atomic bump counter while the segment is filled, so a pool is sized for the
// paths that can be alive (pixels x batch, plus two chunks of slack per segment) and not for the worst case of every path
// in one class.
constexpr int kMinChunkShift = 11;   // chunks hold at least 2048 paths: a multiple of the tile size, so a tile never straddles two
// k_bounce<., MANY> (scenes with more than kBinMax spheres): LDS words of the fixed scratch, floats per staged sphere
// record (inverseTransform rows, transform rows, GeomDev::invZ, 4 B of padding), spheres a lane can record per tile
constexpr int kNanWords = 12;        // nine NaNs (+ padding): the "face frame" of a cube hit without an exit slab (cubeFace: a ray of NaNs)
constexpr int kTicketWords = 4;      // the workgroup's next ticket (+ padding)
// LDS words of the fixed scratch of an instantiation with `cls` queue classes (16, or the mesh scenes' 32)
constexpr int miscWords(int cls) { return 2 * kWaves * cls + 5 * cls + kSeg + (kSeg + 2) + 2 * PT_MAX_BATCH + 2 + kNanWords + kTicketWords; }
static_assert(miscWords(kCls) % 4 == 0 && miscWords(kClsMax) % 4 == 0, "the sphere records that follow are read as float4");
constexpr int kSphRowFloats = 28;
constexpr int kListMax = 6;          // (8 until the 32 queue classes of round 4 took 832 B more: with C5's scene the camera-ray launch then needed 23 200 B of
                                     // LDS, one 512-byte granule more than seven workgroups per CU leave each other -- 2.39 -> 2.86 ms; a camera ray rarely
                                     // meets more than three bounding balls, and a lane whose list is full tests in place)
constexpr int kCtrPad = 32;          // one counter per 128-byte line: same-line atomics serialise at the memory side
// Every workgroup of a launch ends with one atomic per tally, and atomics on one address serialise at ~12 ns each (round 2 measured a
// 0.1 ms tail with 8192 of them on one word): 64 shards leave 32 per word for a 2048-workgroup launch (rounds 2-3: 8 shards, 256 per
// word = the last 3 us of every launch).
constexpr int kTallyShards = 64;
constexpr int kTicketShards = 64;    // tile tickets (k_bounce): 32 workgroups of a 2048-workgroup launch share a counter

struct Ctrl {
    // pos[p][d][s][0] = paths appended to segment s of the queue ENTERING bounce d of a batch with parity p (a run of
    // survivors reserves its place with ONE atomic add, see reserveRun), bump[p][d][0] = chunks handed out for that queue.
    // The last bounce launch of a batch zeroes the OTHER parity, i.e. re-arms the next batch of the slot, so a batch
    // needs neither a memset nor a separate re-arm launch.
    uint32_t pos[2][kMaxDepthSlots][kSeg][kCtrPad];
    uint32_t bump[2][kMaxDepthSlots][kCtrPad];
    // ticket[p][d][0] = tiles handed out beyond the two static ones per workgroup by the launch of bounce d (k_bounce: "tickets")
    // (sharded: ticket t of shard s = blockIdx % n stands for tile 2 grid + t n + s, n = min(kTicketShards, grid) -- one word for a whole
    // launch's tiles serialised its 20 000 atomics at ~12 ns each, twice the launch's own length)
    uint32_t ticket[2][kMaxDepthSlots][kTicketShards][kCtrPad];
    // ... and the tiles handed out by the mesh walk that runs ahead of bounce d (k_mesh_walk, pt_mesh_walk.h: all of its tiles are drawn;
    // the walk of (p, d) zeroes the words of (p ^ 1, d), its slot's next batch)
    uint32_t walkTicket[2][kMaxDepthSlots][kTicketShards][kCtrPad];
    // never zeroed by an iteration
    uint32_t error;                    // sticky device fault: kFault* bits (pool exhausted, chunk-list poll timeout)
    uint32_t pad[kCtrPad - 1];
    unsigned long long sum_live[kMaxDepthSlots];
    // paths that ended at the scatter of bounce d - 1 (they enter bounce d and miss, but are never enqueued), sharded like
    // the tallies below: every workgroup adds to them when it ends, and 2048 atomics on ONE address take ~100 us
    unsigned long long early[kMaxDepthSlots][kTallyShards][kCtrPad / 2];
    unsigned long long light_hits[kTallyShards][kCtrPad / 2], misses[kTallyShards][kCtrPad / 2];
};
constexpr uint32_t kFaultPoolExhausted = 1u, kFaultReserveTimeout = 2u;
constexpr int kReservePollLimit = 1 << 14;    // polls (each a memory round trip) for a chunk-list entry; a real wait is a few

// Camera constants derived once on the host (spec S2)
struct KParams {
    float view[3], up[3], right[3], pos[3];
    float pixLenX, pixLenY, halfW, halfH;
    int   W, H;
    int   shardRank, shardCount;
    int   nLocal;       // pixels rendered by this shard
    int   ngeoms, nmats;
    int   traceDepth;
    int   pixBits;      // bits of a global pixel index (W * H - 1): PathC::pk = pixelIndex | batch index << pixBits
    int   poolChunks;   // chunks of a path pool (chunk 0 is the trash chunk: never handed out, written only after a fault)
    int   chunkShift;   // log2(paths per chunk), >= kMinChunkShift
    int   sceneRect[4]; // union of the primitives' pixel rectangles (GeomDev::rect): camera rays outside miss everything
    // n / W and n / nLocal for n < 2^30 as (n * magic) >> shift (exact, see pt_init: magic_divisor): the two divisions of the
    // camera-ray bounce cost ~50 instructions each when the compiler expands them
    uint32_t magicW, shiftW, magicN, shiftN;
    int   contribLocal; // the radiance buffers and iteration masks hold only this shard's pixels, indexed x + (y / shardCount) * W
                        // (row shards of frames below 2^27 pixels): a rank of N then touches 1/N of the memory, not all of it
    uint32_t magicS, shiftS;   // n / shardCount
    // Camera-ray tiles are laid over rows PADDED to a multiple of the tile size (Wp = ceil(W / 256) * 256; lanes beyond W idle), so a
    // tile is always 256 pixels of ONE row whatever the frame's width: its pixels follow from its wave-uniform position, it is
    // skipped as a whole outside the scene rectangle, and it walks its row's own list of primitives (rowOff / rowIdx).
    int   Wp;           // width of the camera-ray tiles' index space: the column bands (of kBlock pixels) that meet the scene rectangle
    int   nLocalPad;    // this shard's rows that meet the scene rectangle x Wp: the tile index space of one iteration; magicN / shiftN divide by it
    int   firstY0;              // image row of that index space's first row (its first column: the band of sceneRect[0])
    int   firstSkipped;         // this shard's pixels outside it: camera rays that miss whatever their jitter (tallied once per iteration)
    uint32_t magicWp, shiftWp;   // n / Wp
    int   tilesPerRow;  // Wp / 256 when the camera-ray grid is a multiple of it (see k_bounce), else 0
    int   emittersBinned; // every primitive with an emissive material is one of binGeom[]
    int   nBinned;      // 1..kBinMax small primitives (spheres, small cubes): survivors are binned by whether they can
    int   binGeom[kBinMax];   // hit one of them (certainMiss of each); 0: off, every path counts as a candidate
    // ... and their culling data (GeomDev::centre, cullR2, cullK) once more, here: the scatter's two bounding-ball certificates then
    // cost ONE scalar load from the argument block instead of a chain of three (index -> primitive -> its culling group) each.
    // Rows beyond nBinned hold cullR2 = -inf: a ball nobody can miss being certified for.
    float binCull[kBinMax][8];
    // ---- walls: the scene's large cubes (non-binned, at most kWallMax).  With walls the three low class bits of a survivor
    // are not its direction octant but WHICH wall it can still hit: 0..5 = that wall only (every other wall certified
    // missed by wallCertainMiss), 6 = several or uncertified, 7 = none
    int   nWalls;
    int   allClassified;  // every primitive is a wall or binned: a survivor that certainly misses all of them is a miss, now
    float wallOMax;       // certificates are only issued for ray origins with |x| + |y| + |z| <= wallOMax
    // Walls 0 .. nSlotWalls - 1 are certified by ONE plane each (ptd::wallPlanesPossible): six slots -- the plane is x = th with the wall
    // on the low side, on the high side, then y, then z -- hold at most one wall each; the walls behind them (none in a box-shaped
    // room) keep the slab certificate against their inflated box (ptd::wallCertainMiss).
    int   nSlotWalls;
    float slotTh[6];          // the slot's plane, moved towards the interior by the slack that covers the exit point's rounding
    uint32_t slotBit[6];      // 1 << (the slot's wall), or 0: no wall in this slot
    float outerLo[3], outerHi[3];   // box around all the walls' inflated boxes
    // Walls nSlotWalls .. nSlotWalls + nPlaneWalls - 1 are ROTATED cubes: certified by the plane of the face that looks at the scene's
    // interior, whatever its direction (ptd::wallPlanesOriented; round 5): {unit normal n towards the interior, threshold moved that way,
    // `far`: n . x of the half-space n . x >= far that holds EVERY wall's inflated cube -- the ray's segment ends where it leaves it}
    int   nPlaneWalls;
    float planeN[kWallMax][8];
    // ---- README extras, all off by default
    float lensRadius, focalDistance;   // thin lens (depth of field, README.md:100-101); radius 0 = pinhole
    float viewN[3];                    // normalize(view)
    int   directDepth;                 // direct lighting (README.md:107-108): the bounce whose diffuse scatter aims at a light
                                       // (= the scene's trace depth; traceDepth is then one more: the bounce that collects); 0 = off
    int   nEmit;                       // emissive primitives the direct-lighting bounce samples, at most kEmitMax (file order)
    int   nCubes;                      // cubes of the scene (sphere-heavy scenes: rows of the LDS frame table)
    int   nSphCull;                    // sphere-heavy scenes: entries of BounceArgs::sphCull (even: padded with a copy of the last one)
    float sphDirScale;                 // ... and the factor s >= 1 / sqrt(1 - K) on the sweep's unit direction (ptd::sphereHalfLineExcessScaled)
    int   sphN0;                       // ... the first sphN0 entries (even) are the spheres of CLUSTER 0, the rest those of cluster 1: a survivor's class
                                       // bits 3 / 4 say which clusters its ray can hit at all, and a tile sweeps only those (k_bounce: CLUSTER)
    float sphOMax;                     // ... largest |x| + |y| + |z| of a ray origin the clusters' box certificates are issued for
    float sphBox[2][8];                // ... the clusters' inflated world boxes {lo, hi, -, -} (ptd::wallCertainMiss); neither cluster is empty (pt_init)
    int   ldsRowFloats;                // sphere-heavy scenes: floats of the primitives' matrix rows in LDS, ngeoms x kSphRowFloats -- or 0: the rows stay in global
                                       // memory (BounceArgs::rows; scenes of hundreds of primitives, whose rows would leave one workgroup per CU)
    int   pairOff;                     // sphere-heavy scenes, later bounces: byte offset of the pooled pass's pair descriptors in the dynamic LDS ([kWaves][64] words)
    // scenes of HUNDREDS of swept primitives (round 6; k_bounce<..., GROUPS>): the table's entries come in spatial GROUPS of kSphGroupSize
    // consecutive ones, each with a bounding ball of its members' certificate balls (BounceArgs::sphGroups, entries like the table's own); a
    // later tile tests the groups' balls first and sweeps, lane by lane, only the groups its ray may reach
    int   nSphGroups;                  // ... groups in all (0: none -- the flat sweep)
    int   grpN0;                       // ... the first grpN0 of them hold cluster 0's entries [0, sphN0)
    float grpOMax;                     // ... largest |x| + |y| + |z| of a ray origin the groups' certificates are issued for
    int   grpLds;                      // ... 1: the entries' {centre, threshold} and primitive indices are staged in LDS behind the materials (nSphCull <= kGroupLdsMax)
    int   meshStackOff;                // scenes with meshes: byte offset of the lanes' traversal stacks in the dynamic LDS ([levels][kBlock] words)
    int   classOff[kClsMax + 1];          // later bounces: the primitives a tile of class c has to look at are classIdx[classOff[c] ..
                                       // classOff[c + 1]) (BounceArgs::classIdx): not the binned ones unless the class says so, of the walls only
                                       // the class's own, in sphere-heavy scenes no sphere (those come from sphCull)
    int   emitGeom[kEmitMax];
    float emitRho2[kEmitMax];          // |scale x extent|^2 / 4 of each: squared radius of the ball around the box it is sampled through
    float emitBox[kEmitMax][6];        // ... that object-space box as centre, extent: 0, 1 = the unit cube of a sphere or cube; a mesh: its vertices' bounds
};

// SoA PathSegment pool: THREE arrays of `cap` = poolChunks << chunkShift elements -- A: float4 {origin, direction.x} at base,
// B: float4 {direction.yz, throughput.xy} at base + 16 cap bytes, C: three dwords {throughput.z, utilhash(pixelIndex), pixelIndex | batch
// index << pixBits} at base + 32 cap bytes -- the same 44 bytes per path as eleven dword arrays (rounds 1-2), moved by 3 + 3 vector-memory
// instructions per path and bounce instead of 11 + 11: round 3's in-kernel timeline showed a fifth of a later tile's time going into
// ISSUING those instructions (64 lanes x 4 B each), not into waiting for their data.  A wave's access is 1 KiB (768 B) contiguous.
// Chunk c owns [c << chunkShift, (c + 1) << chunkShift) of every array.
// list[s * poolChunks + j] = {generation : 32 | chunk : 32} of the j-th chunk (j >= 1) of segment s of the queue the pool
// holds; the generation is the serial number of the launch that filled the queue, so entries of earlier launches read as
// "not there yet" without any clearing.  The 0-th chunk of segment s is always chunk 1 + s.
struct PathPool {
    float              *base;
    unsigned long long *list;
    uint32_t            cap;
    // byte address of element `slot` of array A / B / C
    __host__ __device__ __forceinline__ char *arrA(size_t slot) const { return reinterpret_cast<char *>(base) + 16 * slot; }
    __host__ __device__ __forceinline__ char *arrB(size_t slot) const { return reinterpret_cast<char *>(base) + 16 * (size_t)cap + 16 * slot; }
    __host__ __device__ __forceinline__ char *arrC(size_t slot) const { return reinterpret_cast<char *>(base) + 32 * (size_t)cap + 12 * slot; }
};
// array C's element (12 bytes: one global_load / store_dwordx3).  pixHash = utilhash(pixelIndex): the pixel's half of every bounce's RNG seed
// (src/pathtrace.cu:41-45), computed ONCE, by the camera-ray bounce (rounds 1-4 stored remainingBounces here -- the same number for every path
// of a launch, traceDepth - depth -- and hashed the pixel index again at every scatter: ~20 vector instructions per path and bounce);
// pk = pixelIndex | batch index << KParams::pixBits (pt_init: the two fit 32 bits).
struct PathC { float cz; uint32_t pixHash, pk; };

// element `slot` of an array whose (wave-uniform) base pointer is `arr`: uniform 64-bit base + 32-bit byte offset, which
// is the addressing form of global_load/store with an SGPR base (no 64-bit vector arithmetic per access)
__device__ __forceinline__ float ldSlot(const float *arr, uint32_t byteOff) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(arr) + byteOff);
}
__device__ __forceinline__ void stSlot(float *arr, uint32_t byteOff, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(arr) + byteOff) = v;
}

// Reserve room for a run of `total` (0..256) paths at the end of segment `oseg`: ONE atomic add on the segment's position
// counter -- a reservation never closes the segment for the others.  Position p lies in the segment's (p >> chunkShift)-th
// chunk.  Chunks are installed ONE AHEAD of their use: the run that contains the first slot of the segment's k-th chunk
// takes a chunk from the pool's bump counter and publishes it as the (k+1)-th (the run at position 0 does so for the 1st;
// the 0-th is static), i.e. a whole chunk's worth of appends before anybody needs it.  A run then looks up the one or two
// chunks it lies in (`cacheK`/`cacheC`: the lane's last lookup, which the next tiles mostly repeat); should an entry not be
// there yet -- the installer's own add is less than a memory round trip old -- it is polled.  Installing comes before
// looking up and waits for nothing, so every poll ends; it is bounded all the same: a timeout or an exhausted pool sets
// the sticky fault word and directs the run to the trash chunk 0 (in bounds; results void; the host reports PT_ERR_DEVICE).
// The loop is WAVE-UNIFORM with one poll per lane and trip: a lane never spins inside a trip, whatever the compiler
// makes of the branches.
// The run occupies  [base0, base0 + split)  and  [base1, base1 + total - split).
__device__ __forceinline__ void reserveRun(uint32_t *pos, uint32_t *bump, unsigned long long *segList, uint32_t oseg,
                                           uint32_t poolChunks, uint32_t shift, uint32_t gen, uint32_t total, uint32_t *fault,
                                           uint32_t &cacheK, uint32_t &cacheC, uint32_t &base0, uint32_t &split, uint32_t &base1) {
    base0 = base1 = 0u;
    split = total;
    uint32_t p = total ? atomicAdd(pos, total) : 0u;
    PT_EXP_RESERVE(p, pos, total)        // (experiment builds only: pt_experiments.h)
    const uint32_t k0 = p >> shift, k1 = (p + (total ? total - 1u : 0u)) >> shift;
    if (total && ((p & ((1u << shift) - 1u)) == 0u || k1 != k0)) { // this run holds the first slot of chunk k1: install the one after it
        const uint32_t kNew = k1 + 1u;
        uint32_t x = (uint32_t)kSeg + 1u + atomicAdd(bump, 1u);
        if (x >= poolChunks) {
            atomicOr(fault, kFaultPoolExhausted);
            x = 0u;
        }
        if (kNew < poolChunks)
            __hip_atomic_store(&segList[kNew], ((unsigned long long)gen << 32) | x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t c0 = 1u + oseg, c1 = 0u;                              // chunk of k0 (k0 == 0: static), chunk of k1
    bool need0 = total && k0 != 0u, need1 = total && k1 != k0;
    if (need0 && cacheK == k0) { c0 = cacheC; need0 = false; }
    for (int polls = 0; __ballot(need0 || need1) != 0ull; ++polls) {
        if (need0) {
            const unsigned long long e = k0 < poolChunks ? __hip_atomic_load(&segList[k0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                         : ((unsigned long long)gen << 32);
            if ((uint32_t)(e >> 32) == gen) { c0 = (uint32_t)e; need0 = false; }
        }
        if (need1) {
            const unsigned long long e = k1 < poolChunks ? __hip_atomic_load(&segList[k1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                         : ((unsigned long long)gen << 32);
            if ((uint32_t)(e >> 32) == gen) { c1 = (uint32_t)e; need1 = false; }
        }
        if (polls > kReservePollLimit && (need0 || need1)) {
            atomicOr(fault, kFaultReserveTimeout);
            if (need0) c0 = 0u;
            if (need1) c1 = 0u;
            need0 = need1 = false;
        }
    }
    if (total) {
        const uint32_t mask = (1u << shift) - 1u;
        base0 = (c0 << shift) + (p & mask);
        if (k1 != k0) {
            split = ((k0 + 1u) << shift) - p;
            base1 = c1 << shift;
            cacheK = k1; cacheC = c1;
        } else {
            cacheK = k0; cacheC = c0;
        }
    }
}

__device__ __forceinline__ uint32_t fastDiv(uint32_t n, uint32_t magic, uint32_t shift) {
    return (uint32_t)(((unsigned long long)n * magic) >> shift);
}

// The camera ray of pixel (x, y) = index pix (spec S2) as k_bounce<true, ...> builds it in place: jitter from the depth-0 stream of
// (iteration, pixel) -- iterHash0 = iterationHash(iter, 0) --, then the thin lens if the camera has one.  (The mesh walk and the test
// library's kernels call this; the camera-ray bounce keeps its own inlined copy, whose register allocation is measured.)
__device__ __forceinline__ void cameraRayAt(const KParams &prm, uint32_t iterHash0, int pix, int x, int y, F3 &org, F3 &dir) {
    Rng rng = makeSeededRandomEngineHashed(iterHash0, pix);
    const float jx = u01(rng);
    const float jy = u01(rng);
    const float sx = ((float)x + jx) - prm.halfW;
    const float sy = ((float)y + jy) - prm.halfH;
    const float a = prm.pixLenX * sx;
    const float b = prm.pixLenY * sy;
    const F3 view = f3(prm.view[0], prm.view[1], prm.view[2]);
    const F3 up = f3(prm.up[0], prm.up[1], prm.up[2]);
    const F3 right = f3(prm.right[0], prm.right[1], prm.right[2]);
    org = f3(prm.pos[0], prm.pos[1], prm.pos[2]);
    dir = normalize((view - right * a) - up * b);
    if (prm.lensRadius > 0.0f) {     // thin lens, as in k_bounce<true, ., true>
        const float lr = prm.lensRadius * __builtin_sqrtf(u01(rng));
        const float phi = u01(rng) * kTwoPi;
        float s, c;
        sincosPoly(phi, s, c);
        const float ft = prm.focalDistance / dot(dir, f3(prm.viewN[0], prm.viewN[1], prm.viewN[2]));
        const F3 focus = org + dir * ft;
        org = (org + right * (lr * c)) + up * (lr * s);
        dir = normalize(focus - org);
    }
}

// ---- one bounce: intersect + shade + accumulate + compact (spec S3-S8) -----------------------------
// Persistent workgroups walk the 256-path tiles of the bounce's queue (the kSeg input segments laid
// end to end), blockIdx-strided.  Survivors are BINNED BY CLASS while they are compacted:
//   class     = bits 0-2 | candidate << 3.  Bits 0-2: in a scene with walls, WHICH wall the new ray can still hit
//               (wallCertainMiss of every wall, see KParams::nWalls; 6 = several, 7 = none), else the octant of its
//               direction; candidate = the new ray is not a certain miss (certainMiss: bounding ball with a 50x safety
//               margin) of every SMALL primitive of the scene -- its spheres and the cubes much smaller than the scene,
//               at most kBinMax, chosen by pt_init,
//   segment   = class * kSub + blockIdx % kSub,
//   rank      = position among the wave's lanes of the same class (four bit ballots -> same-class mask -> mbcnt) plus
//               the earlier waves' totals through LDS = workgroup-level exclusive scan per class,
//   base      = ONE atomic add per non-empty class of the tile on that segment's position counter (16 lanes, one
//               instruction; reserveRun).
// A tile of the next bounce therefore holds paths that test ONE wall (or, without walls, rays of a single direction octant,
// which turns the exact early-miss of the box test (pt_device.h) into a wave-uniform branch), and either candidates only --
// whose tests of the small primitives then run with full waves instead of a few lanes -- or paths that skip the small
// primitives altogether (both are sufficient conditions for the reference's own miss, evaluated on the very ray that is
// stored): it loops over its class's own list of primitives (KParams::classOff).  A survivor that can hit nothing at all
// (scenes whose primitives are all walls or binned) ends at its scatter.
// Queue order never influences results: RNG and accumulator are keyed on the pixel index.
// No workgroup ever waits for another one's work, so there is no residency / dispatch-order assumption (the one
// cross-workgroup wait is reserveRun's bounded poll for a chunk-list entry published a chunk's worth of appends earlier).
// A segment receives survivors of the tiles of the workgroups with one value of blockIdx % kSub only; its paths live in
// chunks of the output pool handed out on demand (reserveRun), so any distribution over the classes fits.
// (Deriving the shard from the tile index instead, T % kSub, measured 3 % slower; 64-path tiles that a wave loads, traces
// and compacts alone, without any barrier in the loop, 18 % slower: four times the atomics, runs a quarter as long.)
//
// FIRST = true is bounce 1 fused with camera-ray generation (spec S2): tile T holds the paths
// j = 256 T + lane of this shard's pixel list and the ray is built in registers, so the first bounce
// reads no path state at all.
//
// Registers: 78 SGPRs and at most 64 VGPRs without a spill in the four main instantiations, i.e. eight workgroups per
// CU (DESIGN.md section 4 lists what it took: laundered constant-address-space access to the kernel's single by-value
// argument and to the primitives, flags that cross divergent regions kept as ints, the bitwise early miss).  Built
// without the SLP vectoriser (packed fp32 runs at half rate on gfx950: profiles/valu_issue_rate.json).
//
// MANY (scenes with more than kBinMax spheres, e.g. the 64-sphere configuration): the wave-uniform loop over the
// spheres only runs the cheap bounding-ball test and RECORDS the spheres a lane may hit (a handful out of 64, and a
// different handful for every lane of a wave of incoherent rays: testing them in place ran the full sphere test 21
// times per wave with 1.8 active lanes).  After the loop, pass k tests every lane's k-th recorded sphere, each lane with
// its own matrices from LDS -- 3-4 passes with tens of active lanes.  Same tests on the same operands; the nearest hit is
// chosen by (distance, then file order), which is what the in-order loop with its strict `<` computes.
// some value, for free (see the tile loop)
__device__ __forceinline__ float anyFloat() {
    float x;
    asm volatile("" : "=v"(x));
    return x;
}
__device__ __forceinline__ F3 anyF3() { return f3(anyFloat(), anyFloat(), anyFloat()); }
__device__ __forceinline__ uint32_t waveSum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Everything a launch needs, passed BY VALUE as the kernel's single argument and read back, phase by phase, from the
// kernarg segment through a laundered constant-address-space pointer (see launder() in pt_device.h): a field then lives
// in SGPRs from the s_load of the phase that uses it to its last use there, instead of from the kernel's entry to its
// end (the compiler hoists kernel arguments and everything derived from them out of the tile loop and then spills).
// Sphere-heavy scenes, later bounces: the bounding-ball data of every sphere, packed (two per 64-byte scalar load instead of
// one 32-byte load out of a 448-byte GeomDev per loop iteration -- the scalar cache's round trip per primitive was the
// critical path of the loop over seventy of them).
struct SphereCull {
    float centre[3];
    float cullR2, cullK;   // cullR2: the threshold of the SCALED certificate, rho^2 smax^2 (1 + 1e-3) s^2, rounded up (sphDirScale); cullK: as GeomDev's + slack
    int   geom;          // index of the sphere among the scene's primitives; -1: padding
    int   pad[2];
};
static_assert(sizeof(SphereCull) == 32, "two per s_load_dwordx16");
constexpr int kSphGroupSize = 8;         // entries per group (k_bounce<..., GROUPS>): an 8-bit outcome mask per lane and group
constexpr int kCandPairs = 12;           // words a lane parks its candidates in between two rounds of passes (k_bounce<..., GROUPS>: [kCandPairs][kBlock] in LDS)
constexpr int kGroupLdsMax = 1280;       // entries whose level-2 data -- {centre, threshold} 16 B + the primitive's index 2 B -- a grouped workgroup stages in LDS
                                         // (23 KB: six workgroups per CU); a longer table is read from global memory, lane by lane
constexpr int kGroupedMin = 128;         // swept primitives from which a scene's later bounces take the grouped instantiations (pt_init)
typedef int int16v __attribute__((ext_vector_type(16)));

// LDS layout of the sphere-heavy variants: the 68-byte hit records are padded to a multiple of 16 B, and so is the frame table
// (the sphere matrices behind them are read as float4)
__host__ __device__ constexpr size_t manyHitBytes(int ngeoms) { return ((size_t)ngeoms * sizeof(GeomHitSmall) + 15) / 16 * 16; }
__host__ __device__ constexpr size_t manyFramePad(int ncubes) { return (16 - ((size_t)ncubes * 54 * sizeof(float)) % 16) % 16; }

// What the set-up of a later bounce's tile reads from the argument block, side by side at its very start: ONE scalar load (the
// fields used to be fetched where they were needed -- five scalar-cache round trips one after the other on the way to a tile's loads).
struct TileArgs {
    const float *inBase;                // PathPool `in`: base, chunk lists, capacity
    const unsigned long long *inList;
    uint32_t inCap;
    uint32_t genIn;
    uint32_t poolChunks, chunkShift;
    uint32_t skipNonCandidates;         // lastBounce && emittersBinned: tiles whose class says "no binned primitive" have nothing to add
    // The launch's small wave-uniform facts in ONE word, fetched once per workgroup and kept in a scalar register: each of them used to
    // be its own scalar load in the phase that asks for it, and round 3's experiments price a dependent scalar-cache round trip at
    // half a percent of a tile's time (profiles/r03_sensitivity_experiments.txt).
    //   bit 0 lastBounce, 1 allClassified, 2 radiance is parked (contrib != null), 3 this bounce aims at a light (direct lighting),
    //   4 contribLocal, 5 / 6 the output / input queue carries the last bounce's candidate bits; bits 8-10 nWalls, 11-13 nSlotWalls, 14-16 nBinned, 17-19 nPlaneWalls, 20-31 nmats
    uint32_t hot;
};
constexpr uint32_t kHotLast = 1u, kHotAllClassified = 2u, kHotContrib = 4u, kHotToLight = 8u, kHotContribLocal = 16u;
// Sphere clusters (k_bounce: CLUSTER), scenes whose emitters are all binned: the last bounce only asks whether a path ends on an emitter and
// visits the tiles whose class has bit 3 -- which, between the other bounces, stands for "a binned primitive OR a sphere of cluster 0", five
// times as many paths as the binned primitives' candidates alone.  So the queue that enters the last bounce is classed differently: bit 3 =
// a binned primitive's candidate, bit 4 = a candidate of EITHER cluster (a tile with it sweeps all the spheres: they may stand in front of
// the emitter).  kHotWritesLastBits: the launch that fills that queue; kHotReadsLastBits: the launch that reads it.
constexpr uint32_t kHotWritesLastBits = 32u, kHotReadsLastBits = 64u;
// PT_FLAG_MIXTURE_WEIGHTED: a REFL > 0 material's branch carries its 1 / p weight (src/interactions.h:54-58 read to the letter)
constexpr uint32_t kHotMixWeighted = 128u;
__host__ __device__ constexpr uint32_t hotWalls(uint32_t h) { return (h >> 8) & 7u; }
__host__ __device__ constexpr uint32_t hotSlotWalls(uint32_t h) { return (h >> 11) & 7u; }
__host__ __device__ constexpr uint32_t hotPlaneWalls(uint32_t h) { return (h >> 17) & 7u; }
__host__ __device__ constexpr uint32_t hotBinned(uint32_t h) { return (h >> 14) & 7u; }
__host__ __device__ constexpr uint32_t hotMats(uint32_t h) { return h >> 20; }
static_assert(sizeof(TileArgs) == 40, "ten dwords");
struct BounceArgs {
    TileArgs tile;                      // (first: offset 0)
    KParams prm;
    int iter, batch, depth, lastBounce, parity;
    uint32_t genIn, genOut;             // serial numbers of the launches that filled / fill the input / output pool
    PathPool in, out;
    Ctrl *ctrl;
    const GeomDev *ggeoms;
    const MaterialDev *gmats;
    const float4 *ghit;                 // GeomHitDev[ngeoms] (19 x 16 B each), staged in LDS by the prologue
    float *contrib;
    const WallBox *walls;               // [prm.nWalls] inflated world-space boxes of the walls
    const float4 *meshRecs;             // ptd::MeshRec[] of every mesh of the scene (k_bounce<., ., ., true>), or nullptr
    uint32_t *hitMask;                  // [ceil(max_batch / 32)][W * H]: bit b of word w set = contrib[32 w + b][pix] was written
    const SphereCull *sphCull;          // sphere-heavy scenes (k_bounce<false, true, ...>): the spheres' culling data, packed
    const int *classIdx;                // later bounces: per queue class, the indices of the primitives to look at, file order (KParams::classOff)
    const int *rowOff;                  // camera rays (a tile is 256 pixels of one row): the primitives that can be reached from image row y are the entries
    const int *rowIdx;                  //   rowOff[y] .. rowOff[y + 1] of rowIdx, file order: pairs {primitive, x0 | x1 << 16} = the row's pixels
                                        //   inside the hull of the primitive's projected corners (pt_init); rowOff == nullptr: every primitive
    uint32_t *hostFault;                // the sticky fault word's copy in page-locked host memory (written by a batch's last launch), or nullptr
    // ---- scenes with meshes: the walks run AHEAD of the bounce (k_mesh_walk) and leave, per path of a tile that lists a mesh, the nearest
    // mesh hit: meshHit[i] = bits of its distance << 32 | the winning triangle's unit << 1 | front side (all ones: none), i = the path's slot
    // (the distance only orders the hits of SEVERAL meshes -- the walk's atomic minimum; a tile that lists one mesh leaves it zero)
    // in the input pool (camera rays: its index in the tiles' padded pixel space)
    const float4 *rows;                 // sphere-heavy scenes: [ngeoms][7] the primitives' matrix rows (inverseTransform, transform, GeomDev::invZ), as the LDS copy holds them
    unsigned long long *meshHit;
    const int *walkIdx;                 // the meshes alone: per queue class (walkClassOff), all of them ([walkAll0, walkAll1)), per image row pairs as rowIdx (walkRowOff)
    const int *walkRowOff;              // ... nullptr where rowOff is
    int walkClassOff[kClsMax + 1];
    int walkAll0, walkAll1;
    // the walk's per-mesh rows (ptk::WalkMesh, 128 B each, made by pt_init): staged in LDS by every workgroup when the scene holds at most
    // kWalkMeshLdsMax meshes (walkMeshLds = their number), else read from here per job (walkMeshLds = 0: a table of a thousand meshes would
    // not fit the LDS, and one of a hundred would cost the kernel its residency)
    const float4 *walkMeshRows;
    int walkMeshLds;
    const SphereCull *sphGroups;        // scenes of hundreds of swept primitives: the groups' bounding balls (KParams::nSphGroups), or nullptr
};
typedef const PT_CAS BounceArgs *ArgsPtr;
typedef const PT_CAS GeomDev *GeomPtr;
typedef const PT_CAS WallBox *WallPtr;
// the culling group of a GeomDev (offset 0x60), as one 32-byte scalar load
struct CullGroup {
    float centre[3];
    float cullR2, cullK, boundR;
    int   cullFlags, material;
};
// ONE scalar load (inline asm: the compiler would split the group again, fetch the flags word first and the rest behind
// the branch on it -- two serialised scalar-memory round trips per sphere)
typedef int int8v __attribute__((ext_vector_type(8)));
template <bool ONE_LOAD>
__device__ __forceinline__ CullGroup loadCull(const PT_CAS GeomDev *g) {
    CullGroup c;
    if (!ONE_LOAD) {   // small scenes: field by field, the compiler places the loads (no 8-aligned SGPR tuple to find)
        c.centre[0] = g->centre[0]; c.centre[1] = g->centre[1]; c.centre[2] = g->centre[2];
        c.cullR2 = g->cullR2; c.cullK = g->cullK; c.boundR = g->boundR;
        c.cullFlags = g->cullFlags; c.material = g->material;
        return c;
    }
    int8v v;
    asm volatile("s_load_dwordx8 %0, %1, 0x60\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(g) : "memory");
    c.centre[0] = __int_as_float(v[0]); c.centre[1] = __int_as_float(v[1]); c.centre[2] = __int_as_float(v[2]);
    c.cullR2 = __int_as_float(v[3]); c.cullK = __int_as_float(v[4]); c.boundR = __int_as_float(v[5]);
    c.cullFlags = v[6]; c.material = v[7];
    return c;
}
static_assert(offsetof(GeomDev, centre) == 0x60, "loadCull's offset");
static_assert(sizeof(CullGroup) == 32 && offsetof(GeomDev, cullFlags) - offsetof(GeomDev, centre) == offsetof(CullGroup, cullFlags), "CullGroup mirrors GeomDev");

// DOF (with FIRST only): camera rays start on a thin lens (README.md:100-101), so they share no origin (no precomputed
// object-space camera position) and the pixel rectangles, which project the primitives through a pinhole, are not used.
// PLAIN: the scene's materials are diffuse, emissive or 50/50 perfect mirrors only and no README extra is on -- no refraction, no
// specular lobe, no direct lighting: their code (a third of the scatter's static instructions: Schlick + refract, the lobe's pow, the
// emitter pick) is not even compiled in.  Same results (the branches are never taken in such a scene); pt_init picks the instantiation.
// CUBES (with MANY only): the swept small primitives include cubes -- the per-lane tests then look the primitive's type up and run the box
// test for one; an instantiation of its own (four workgroups per CU: both tests inlined in every pass need the registers), so that a scene of
// spheres runs exactly the code of rounds 2-4.
// GROUPS (with MANY, no meshes): scenes of hundreds of swept primitives -- the later bounces' sweep is two-level (the groups' bounding balls,
// then per LANE the members of the groups its ray may reach), and neither hit records nor matrix rows nor face frames are staged in LDS,
// in the camera-ray bounce either (518 primitives' records left it three workgroups per CU): instantiations of their own, so that the
// 64-sphere configuration runs the code it ran before.
template <bool FIRST, bool MANY, bool DOF = false, bool MESH = false, bool PLAIN = false, bool CUBES = false, bool GROUPS = false>
#ifndef PT_MESH_WG_FIRST
#define PT_MESH_WG_FIRST 7
#define PT_MESH_WG_NEXT 7
#endif
#ifndef PT_GROUPS_WG
#define PT_GROUPS_WG 5
#endif
#ifndef PT_CUBES_WG
#define PT_CUBES_WG 4
#endif
__global__ __launch_bounds__(kBlock, (MANY && CUBES) ? (GROUPS ? 4 : PT_CUBES_WG) : (MESH ? (MANY ? 4 : (FIRST ? PT_MESH_WG_FIRST : PT_MESH_WG_NEXT)) : (DOF ? 5 : (MANY ? (FIRST ? 7 : (GROUPS ? PT_GROUPS_WG : 6)) : 8)))) void k_bounce(BounceArgs argsByValue) {
    static_assert(MANY || !CUBES, "swept cubes only exist where primitives are swept");
    static_assert(!GROUPS || (MANY && !MESH && !DOF), "groups: sphere-heavy scenes without meshes (the camera-ray bounce: its pinhole form)");
    static_assert(FIRST || !DOF, "the lens only concerns the camera-ray bounce");
    (void)argsByValue;
    const ArgsPtr kargs = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    censusEnter();                       // (instrumented build only)
    probe(FIRST ? 31 : 30);              // (prologue; timeline builds: the first stamp of the launch)
    // LDS: the material table and the per-geom hit records (normal matrix, material, type: indexed per lane by
    // the nearest hit), and the compaction scratch.  Geometry itself is wave-uniform in the nearest-hit loop, so it is fetched through the
    // scalar path (s_load into SGPRs, used directly as VALU operands): measured against an LDS-staged copy
    // read back with ds_read_b128 broadcasts this is 5 % faster on Cornell (7 geoms) and 11 % on the 70-geom
    // scene, and it frees ~40 VGPRs.
    // Layout: the fixed-size scratch first (constant offsets), then the tables whose sizes depend on the scene.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *const s_misc = reinterpret_cast<uint32_t *>(smem);
    // sphere-heavy scenes without meshes: the spheres in two spatial clusters, a survivor's candidate bits name the clusters it can hit
    constexpr bool CLUSTER = MANY && !MESH;
    constexpr bool WIDE = MESH || CLUSTER;                 // two candidate bits: 32 queue classes
    constexpr int kMiscWords = miscWords(WIDE ? kClsMax : kCls);
    MaterialDev *const smats = reinterpret_cast<MaterialDev *>(smem + kMiscWords * sizeof(uint32_t));
#define S_GEOMHIT(nmats_) (reinterpret_cast<GeomHitDev *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * (nmats_)))
// sphere-heavy scenes (MANY): compact hit records, then the cubes' face frames, then the per-primitive matrices and the lanes' lists
#define S_GEOMHIT_SMALL(nmats_) (reinterpret_cast<GeomHitSmall *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * (nmats_)))
#define S_FRAMES(nmats_, ngeoms_) (reinterpret_cast<float *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * (nmats_) + manyHitBytes(ngeoms_)))
#define S_SPH(nmats_, ngeoms_, ncubes_) (reinterpret_cast<float *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * (nmats_) + \
                                         (MANY ? manyHitBytes(ngeoms_) + (size_t)(ncubes_) * 54 * sizeof(float) + manyFramePad(ncubes_) : sizeof(GeomHitDev) * (ngeoms_))))
    // queue classes of this instantiation and append-counter shards per class (see kClsMax)
    constexpr int CLS = WIDE ? kClsMax : kCls, SUB = kSeg / CLS, CLSBITS = WIDE ? 5 : 4;
    // the class bits behind which a binned primitive -- an emitter, where a tile is skipped for not reaching one -- can stand: the candidate
    // bits; CLUSTER: bit 3 alone (the binned primitives are all of group 0, bit 4 stands for spheres of cluster 1 only)
    constexpr uint32_t kEmitBits = CLUSTER ? 8u : (WIDE ? 24u : 8u);
    uint32_t *const s_wave = s_misc;                       // [2][kWaves][CLS] survivors per wave and class (zero between tiles)
    uint32_t *const s_base = s_wave + 2 * kWaves * CLS;        // [5][CLS] this tile's output run per class: first slot, paths before the
                                                           //           chunk boundary, first slot behind it; the class's last chunk lookup (reserveRun)
    uint32_t *const s_segcnt = s_base + 5 * CLS;           // [kSeg]   paths per input segment
    uint32_t *const s_segpre = s_segcnt + kSeg;            // [kSeg+2] tile prefix per input segment, [kSeg+1] = live paths
    uint32_t *const s_iterHash = s_segpre + kSeg + 2;      // [2][PT_MAX_BATCH] iterationHash(iter + b, depth) and (iter + b, 0)
    const float *const s_nan = reinterpret_cast<const float *>(s_iterHash + 2 * PT_MAX_BATCH + 2);   // [kNanWords] NaNs: see cubeFace
    uint32_t *const s_ticket = s_iterHash + 2 * PT_MAX_BATCH + 2 + kNanWords;   // [1] the tile after the next one (tickets, below)

    uint32_t nLive, numTiles;
    {
        const ArgsPtr A = launder(kargs);
        Ctrl *const ctrl = A->ctrl;
        const int depth = A->depth, parity = A->parity;
        // The prologue's global reads ALL go out here, before anything waits: the first 256 x 16 bytes of the materials and of the
        // ready-made hit records (Cornell: 20 + 152) stay in registers across the scan of the segment counts, whose own loads follow.
        // (Round 3's timeline: 29 k cycles of prologue per workgroup and launch -- kernel arguments, counts, barrier, materials,
        // primitive, ITS material, barrier: five dependent round trips, a third of a last bounce's 33 us.)
        const int m16 = A->prm.nmats * (int)(sizeof(MaterialDev) / 16);
        const int h16 = MANY ? 0 : A->prm.ngeoms * (int)(sizeof(GeomHitDev) / 16);
        const float4 *const msrc = reinterpret_cast<const float4 *>(A->gmats);
        const float4 *const hsrc = A->ghit;
        float4 stageM = make_float4(0, 0, 0, 0), stageH = make_float4(0, 0, 0, 0);
        if ((int)threadIdx.x < m16) stageM = msrc[threadIdx.x];
        if ((int)threadIdx.x < h16) stageH = hsrc[threadIdx.x];
        if (A->lastBounce) {   // re-arm the slot's next batch: nobody touches the other parity's counters now
            uint32_t *other = &ctrl->pos[parity ^ 1][0][0][0];
            const int nwords = (A->prm.traceDepth + 2) * kSeg;
            for (int i = blockIdx.x * kBlock + threadIdx.x; i < nwords; i += gridDim.x * kBlock) other[i * kCtrPad] = 0u;
            if (blockIdx.x == 0 && threadIdx.x < kMaxDepthSlots) ctrl->bump[parity ^ 1][threadIdx.x][0] = 0u;
            if (blockIdx.x == 1 % gridDim.x)
                for (int i = threadIdx.x; i < kMaxDepthSlots * kTicketShards; i += kBlock) (&ctrl->ticket[parity ^ 1][0][0][0])[i * kCtrPad] = 0u;
            // ... and the batch's last launch (which compacts nothing, hence raises no fault itself) hands a fault word the earlier
            // launches may have set to the host: its copy in page-locked memory is what pt_readback looks at after its synchronisation,
            // without a device-to-host copy of its own.  Here, in the prologue, it costs the tile loop no register.
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                const uint32_t e = __hip_atomic_load(&ctrl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t *const hostFault = A->hostFault;
                if (e != 0u && hostFault != nullptr) __hip_atomic_store(hostFault, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        // input queue: segment s holds s_cnt[s] paths = tiles [s_pre[s], s_pre[s+1]) of the global tile index
        if (FIRST) {
            nLive = (uint32_t)A->prm.nLocal * (uint32_t)A->batch;     // `batch` consecutive iterations share one wavefront
            numTiles = (uint32_t)A->prm.nLocalPad / kBlock * (uint32_t)A->batch;   // (tiles lie on padded rows)
        } else {
            const bool skipNonCand = A->tile.skipNonCandidates != 0u;
            if (threadIdx.x < 64) {          // wave 0: exclusive scan of the kSeg tile counts, kSeg / 64 consecutive segments per lane
                constexpr int kPerLane = (kSeg + 63) / 64;
                uint32_t cs[kPerLane], ts[kPerLane];
                uint32_t inc = 0u, sum = 0u;
#pragma unroll
                for (int q = 0; q < kPerLane; ++q) {
                    const int sgi = (int)threadIdx.x * kPerLane + q;
                    cs[q] = sgi < kSeg ? ctrl->pos[parity][depth][sgi][0] : 0u;
                    ts[q] = (cs[q] + kBlock - 1) / kBlock;
                    // (the last bounce of a scene whose emitters are all binned: the tiles of the classes that cannot reach one have
                    // nothing to add -- they get no tile index at all instead of being stepped over one by one)
                    if (skipNonCand && ((uint32_t)(sgi / SUB) & kEmitBits) == 0u) ts[q] = 0u;
                    inc += ts[q];
                    sum += cs[q];
                }
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(inc, o, 64), us = __shfl_up(sum, o, 64);
                    if ((int)threadIdx.x >= o) { inc += up; sum += us; }
                }
                uint32_t run = inc;                                // inclusive over this lane's segments; walk them backwards
#pragma unroll
                for (int q = kPerLane - 1; q >= 0; --q) {
                    const int sgi = (int)threadIdx.x * kPerLane + q;
                    if (sgi < kSeg) { s_segcnt[sgi] = cs[q]; s_segpre[sgi + 1] = run; }
                    run -= ts[q];
                }
                if (threadIdx.x == 0) s_segpre[0] = 0;
                if (threadIdx.x == 63) s_segpre[kSeg + 1] = sum;   // total live paths
            }
            __syncthreads();
            numTiles = s_segpre[kSeg];
            nLive = s_segpre[kSeg + 1];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctrl->sum_live[depth], (unsigned long long)nLive);
        // (camera rays: the pixels outside the tiles' index space are misses; workgroup 0 tallies them, tiles or no tiles)
        if (blockIdx.x >= numTiles && !(FIRST && blockIdx.x == 0)) return;
        if (threadIdx.x < 2 * kWaves * CLS) s_wave[threadIdx.x] = 0u;
        if (threadIdx.x < CLS) s_base[3 * CLS + threadIdx.x] = 0xffffffffu;   // no chunk looked up yet
        if (threadIdx.x < kNanWords) s_iterHash[2 * PT_MAX_BATCH + 2 + threadIdx.x] = 0x7fc00000u;
        // (only the batch's own iterations: a batch of 32 needs 64 of the 512 entries -- every workgroup of every launch fills this table)
        {
            const int nb = A->batch;
            for (int i = threadIdx.x; i < 2 * nb; i += kBlock) {
                const int b = i < nb ? i : i - nb;
                s_iterHash[i < nb ? b : PT_MAX_BATCH + b] = iterationHash(A->iter + b, i < nb ? depth : 0);
            }
        }

        // stage the materials in LDS once per (persistent) workgroup, 16 B per lane per step
        const int ngeoms = A->prm.ngeoms;
        GeomHitDev *const s_geomHit = S_GEOMHIT(A->prm.nmats);
        float4 *mdst = reinterpret_cast<float4 *>(smats);
        if ((int)threadIdx.x < m16) mdst[threadIdx.x] = stageM;
        for (int i = threadIdx.x + kBlock; i < m16; i += kBlock) mdst[i] = msrc[i];
        if (MANY && !GROUPS) {
            // the tables of a sphere-heavy scene -- hit records, face frames, matrix rows, (later bounces) the sweep's entry -> primitive
            // map -- arrive as ONE host-built image in this very layout (pt_init): a straight copy, four 16-byte loads per lane in flight
            const int n16 = (int)((manyHitBytes(ngeoms) + (size_t)A->prm.nCubes * 54 * sizeof(float) + manyFramePad(A->prm.nCubes) +
                                   (size_t)A->prm.ldsRowFloats * sizeof(float) +
                                   (FIRST ? 0 : ((size_t)A->prm.nSphCull + 7) / 8 * 8 * sizeof(uint16_t))) / 16);
            float4 *const dst = reinterpret_cast<float4 *>(S_GEOMHIT_SMALL(A->prm.nmats));
            for (int i0 = 0; i0 < n16; i0 += 4 * kBlock) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + q * kBlock + (int)threadIdx.x;
                    v[q] = i < n16 ? hsrc[i] : make_float4(0, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + q * kBlock + (int)threadIdx.x;
                    if (i < n16) dst[i] = v[q];
                }
            }
        }
        if (GROUPS && !FIRST && A->prm.grpLds) {
            // the sweep's level-2 data: per entry {centre, threshold} (the first 16 bytes of its SphereCull) and the primitive's index -- what a
            // lane reads for the members of ITS candidate groups (from global memory a group's sixteen-byte loads were a chain of round trips)
            const int nS = A->prm.nSphCull;
            float4 *const e16 = reinterpret_cast<float4 *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * A->prm.nmats);
            uint16_t *const emap = reinterpret_cast<uint16_t *>(e16 + nS);
            const float4 *const src = reinterpret_cast<const float4 *>(A->sphCull);
            for (int i = threadIdx.x; i < nS; i += kBlock) {
                const float4 a = src[2 * i], b = src[2 * i + 1];
                e16[i] = a;
                emap[i] = (uint16_t)__float_as_int(b.y);
            }
        }
        if (!MANY) {
            float4 *const hdst = reinterpret_cast<float4 *>(s_geomHit);
            if ((int)threadIdx.x < h16) hdst[threadIdx.x] = stageH;
            for (int i = threadIdx.x + kBlock; i < h16; i += kBlock) hdst[i] = hsrc[i];
        }
    }
    // (the barrier that publishes the staged tables stands further down, behind the first tile's loads: they only need the
    // segment prefix, which the barrier above published, and fly while the workgroup meets here)

    // (see TileArgs::hot; opaque to the optimiser so that it stays ONE register instead of being re-derived from re-loaded fields)
    uint32_t hotWord = launder(kargs)->tile.hot;
    asm volatile("" : "+s"(hotWord));
    // (every use goes through an empty asm of its own: what is derived from the word -- a mask, a count, an LDS offset -- is then
    // derived where it is used, not hoisted out of the tile loop into registers that live, and spill, across the whole tile)
    auto hotNow = [&]() -> uint32_t {
        uint32_t h = hotWord;
        asm volatile("" : "+s"(h));
        return h;
    };
    // tallies of the WAVE, in scalar registers (a ballot's population per tile; rounds 1-4 kept them per lane: three vector registers across
    // the whole tile loop, which round 5's box test needed back), flushed once at the end
    uint32_t sLight = 0, sMiss = 0;
    uint32_t wvSel = 0;                     // which half of s_wave the current tile counts in (0 or kWaves * kCls)
    uint32_t sEarly = 0;                    // survivors that certainly miss everything: ended at the scatter
    uint32_t sgIn = 0;                      // input segment of the tile being set up (tiles are visited in increasing order)
    // FIRST: the k-th tile of a workgroup is rotated k column bands to the right inside its row (see below).  The grid is a multiple
    // of the tiles per row (pt_init), so every tile of a workgroup has the same band c0 = blockIdx % tilesPerRow: ONE division, here.
    // The loop then carries the FIRST tile of the row (T - c0, which steps by the grid like T and stays below numTiles exactly as
    // long as T does: both are multiples of the tiles per row apart from c0 < tilesPerRow) and the rotated band.
    uint32_t rot = 0;                       // (c0 + k) % tilesPerRow of the tile about to be processed
    uint32_t rowShift = 0;                  // c0
    // (a camera-ray launch that draws tickets -- see TICKETS below -- takes its tiles in ticket order: no rotation to set up)
    constexpr bool kTickets = true;
    const bool ticketed = kTickets && (hotWord & kHotLast) == 0u;
    if (FIRST && !ticketed) {
        const uint32_t tpr = (uint32_t)launder(kargs)->prm.tilesPerRow;
        if (tpr > 1) rowShift = rot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x % tpr));
    }

    // ---- later bounces: a tile's paths are requested ONE TILE AHEAD.  A workgroup's tiles form a chain of dependent
    // latencies -- load 44 B per path, trace, reserve the output runs (an atomic round trip between two barriers), store --
    // and with eight workgroups per CU all in the same chain the vector units idled half of the time (round 2 profile:
    // 174 us for 15 k tiles at bounce 7 against 291 us for 53 k at bounce 2).  The loads of tile k + 1 are issued when
    // tile k has been traced, just before its compaction: by then the tracing registers are dead (no extra VGPRs), and
    // the loads fly while the workgroup waits at the barriers and for the atomics.
    struct TileMeta {
        bool valid;                         // this lane holds a path
        uint32_t idx;                       // its slot in the input pool
        uint32_t cls;                       // wave-uniform: the tile's queue class: bit 3 = its paths may hit a binned primitive; bits 0-2 in a
                                            // scene with walls = the one wall they can still hit (6: any, 7: none), else the direction octant
        const float *base;                  // the input pool's arrays (from setupTile's one scalar load to loadTile; dead afterwards)
        uint32_t cap;
    };
    struct PathRegs { F3 org, dir, col; uint32_t pixHash, pk; };
    // tile T of the queue -> its segment, class and the lane's slot; false: the tile needs no work at all
    auto setupTile = [&](uint32_t T, uint32_t tid, TileMeta &m) -> bool {
        // everything the set-up needs from the argument block: one scalar load, in flight while the segment is looked up in LDS
        const PT_CAS TileArgs &ta = launder(kargs)->tile;
        const uint32_t shift = ta.chunkShift, poolChunks = ta.poolChunks, genIn = ta.genIn, skipNonCand = ta.skipNonCandidates;
        const PT_CAS unsigned long long *const inList = (const PT_CAS unsigned long long *)ta.inList;
        m.base = ta.inBase;
        m.cap = ta.inCap;
        // global tile -> (segment, local tile): the segment is the number of prefix entries s_segpre[1 .. kSeg] that do not exceed T.
        // Every lane compares its kSeg / 64 entries and the ballots are counted -- ONE LDS round trip.  (Rounds 1-3 walked the prefix
        // from the previous tile's segment, a chain of dependent LDS reads: ~10 per tile, and up to kSeg of them in front of a
        // workgroup's very first loads -- round 3's timeline has 1067 cycles per tile in it, and the tail of the prologue.)
        {
            const uint32_t ln = tid & 63u;
            uint32_t cnt = 0u;
#pragma unroll
            for (int q = 0; q < kSeg / 64; ++q) cnt += (uint32_t)__popcll(__ballot(s_segpre[1 + 64 * q + ln] <= T));
            sgIn = cnt;
        }
        probe(26);                                              // (segment found)
        sgIn = (uint32_t)__builtin_amdgcn_readfirstlane((int)sgIn);
        const uint32_t segFirst = s_segpre[sgIn];
        const uint32_t local = (T - segFirst) * kBlock + tid;
        m.valid = local < s_segcnt[sgIn];
        m.cls = sgIn / SUB;
        // The last bounce only asks whether a path ends on an emitter (S7: no scatter).  When every emitter of the scene
        // is a binned small primitive, the paths of a non-candidate tile certainly miss all of them: nothing to add.
        if (skipNonCand != 0u && (m.cls & kEmitBits) == 0u) return false;
        // the tile's chunk (a chunk is a multiple of the tile size): j-th chunk of the segment, wave-uniform lookup through the
        // scalar cache (entries were written by the previous launch; the 0-th chunk of a segment is static)
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((T - segFirst) * kBlock));
        const uint32_t j = q0 >> shift;
        uint32_t chunk = 1u + sgIn;
        if (j != 0u) {
            const unsigned long long e = j < poolChunks ? inList[(size_t)sgIn * poolChunks + j] : 0ull;
            chunk = (uint32_t)(e >> 32) == genIn ? (uint32_t)e : 0u;
            chunk = chunk < poolChunks ? chunk : 0u;
        }
        m.idx = (chunk << shift) + (local - (j << shift));
        probe(27);                                              // (chunk looked up: the loads can go)
        return true;
    };
    auto loadTile = [&](const TileMeta &m, PathRegs &r) {
        if (m.valid) {
            // three loads: 16 + 16 + 12 bytes (uniform base of the array + the lane's element)
            const char *const src = reinterpret_cast<const char *>(m.base);
            const size_t cap = (size_t)m.cap;
            const float4 a = *reinterpret_cast<const float4 *>(src + 16 * (size_t)m.idx);
            const float4 b = *reinterpret_cast<const float4 *>(src + 16 * cap + 16 * (size_t)m.idx);
            const PathC c = *reinterpret_cast<const PathC *>(src + 32 * cap + 12 * (size_t)m.idx);
            r.org = f3(a.x, a.y, a.z);
            r.dir = f3(a.w, b.x, b.y);
            r.col = f3(b.z, b.w, c.cz);
            r.pixHash = c.pixHash;
            r.pk = c.pk;                                        // pixelIndex | batch index << pixBits
        }
    };
    // the first tile of this workgroup that needs work, and its paths
    uint32_t T = blockIdx.x - rowShift;      // (FIRST with rotated bands: the first tile of the row; else the tile itself)
    TileMeta nextMeta = {false, 0u, 0u, nullptr, 0u};
    PathRegs nextRegs = {f3(0, 0, 0), f3(0, 0, 1), f3(0, 0, 0), 0, 0};
    if (!FIRST) {
        while (T < numTiles && !setupTile(T, threadIdx.x, nextMeta)) T += gridDim.x;
        if (T < numTiles) loadTile(nextMeta, nextRegs);
    }
    __syncthreads();                         // (materials, hit records, iteration hashes, the cleared scratch: staged by the prologue)
    // ---- TICKETS (later bounces that compact): a workgroup's first two tiles are static (blockIdx, blockIdx + grid), every further one
    // is 2 grid + a ticket from the launch's counter.  Tiles differ in cost by a factor of three (one wall's box test against every
    // primitive of a candidate tile), and a launch is only 7-26 tiles per workgroup long: with a fixed stride the slowest workgroup's
    // surplus was the tail of every launch.  The ticket for the tile after the next one is drawn by lane 16 of wave 0 next to the
    // reservation's atomics (same round trip, between the same two barriers) and handed to the other waves through LDS, so the next
    // tile -- whose loads are requested before the compaction -- is always known: no new latency in the chain.  A shard's tickets are
    // drawn in increasing order, so a workgroup whose next tile lies beyond the queue's end holds no valid later one.
    uint32_t Tn1 = kTickets ? T + gridDim.x : 0u;   // the tile after T (carried only where tickets can be drawn; else always T + grid)
    // (a grid smaller than kTicketShards -- a partitioned or small device, PT_AMD_BLOCKS_PER_CU, one workgroup per CU -- has one shard per
    // workgroup: a shard nobody draws from would leave its tiles unvisited)
    auto ticketShards = [&]() -> uint32_t { return gridDim.x < (uint32_t)kTicketShards ? gridDim.x : (uint32_t)kTicketShards; };
    while (T < numTiles) {
        // the lane id, opaque to the optimiser: the lane masks derived from it (tid < 16, wave > k, ...) are then
        // recomputed where a tile needs them -- one v_cmp each -- instead of being hoisted out of the loop into SGPR pairs
        // that live, spilled, across the whole tile
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        // the tile index is wave-uniform; said so explicitly, everything derived from it -- the tile's iteration, row and first
        // column, the scene-rectangle test, the row's primitive list -- stays in scalar registers and on the scalar unit (left to
        // itself the compiler carried T in a VGPR and expanded the divisions below into ~60 vector instructions per tile)
        T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
        probe(14);                                              // (a tile starts)
        uint32_t Tnext = kTickets ? Tn1 : T + gridDim.x;   // (last bounce: T + grid as well)
        bool valid;
        uint32_t tileCls = 0u;              // wave-uniform: the tile's queue class (later bounces)
        PathRegs cur = nextRegs;
        int itb = 0;                                            // which iteration of the batch this path belongs to
        int px = 0, py = 0;                                     // pixel coordinates (FIRST only)
        uint32_t recIdx0 = 0u;                                  // FIRST, MESH: the tile's first index in the padded pixel space (BounceArgs::meshHit)
        if (FIRST) {
            const ArgsPtr A = launder(kargs);
            const PT_CAS KParams &prm = A->prm;
            // Tile T = b + k grid covers 256 pixels of one (padded) row.  A row is `perRow` tiles; when the grid is a multiple
            // of perRow (pt_init) a workgroup would stay in ONE column band of the frame -- outside the scene rectangle
            // its tiles cost 15x less than inside.  So the k-th tile of a workgroup is rotated k bands to the right inside its
            // row: a bijection on the row's tiles (they share k), which walks every workgroup through all bands.
            uint32_t pixTile = T;
            const uint32_t tilesPerRow = (uint32_t)prm.tilesPerRow;
            if (tilesPerRow > 1 && !ticketed) {
                pixTile = T + rot;                              // (T: the first tile of this tile's row)
                rot = rot + 1u == tilesPerRow ? 0u : rot + 1u;
            }
            // the tile's (wave-uniform) position: iteration of the batch, row of the shard, first column -- no per-lane divisions
            const uint32_t idx0 = pixTile * kBlock;
            if (MESH) recIdx0 = idx0;
            const uint32_t itb0 = fastDiv(idx0, prm.magicN, prm.shiftN);
            const uint32_t j0 = idx0 - itb0 * (uint32_t)prm.nLocalPad;
            const int lr0 = (int)fastDiv(j0, prm.magicWp, prm.shiftWp);
            // (the index space starts at the column band and the row that meet the scene rectangle first: the band from the
            // rectangle's own bound, which the test below loads anyway, the row folded into the shard's offset by the host)
            const int sr0 = prm.sceneRect[0], sr1 = prm.sceneRect[1], sr2 = prm.sceneRect[2], sr3 = prm.sceneRect[3];
            const int x0 = ((int)j0 - lr0 * prm.Wp) + ((sr0 > 0 ? sr0 : 0) & ~(kBlock - 1));
            const int y0 = lr0 * prm.shardCount + prm.firstY0;
            px = x0 + (int)tid;
            py = y0;
            itb = (int)itb0;
            valid = px < prm.W;                                 // (lanes in the padding hold no pixel)
            // When the tile lies outside the scene rectangle altogether, all its camera rays are misses -- tally them and take
            // the next tile (no rays, no compaction, no barrier; the test is the same for the four waves of the workgroup).
            // (the four bounds loaded together and compared without short-circuit: one scalar load, no chain of dependent ones)
            if ((y0 < sr1) | (y0 > sr3) | (x0 + (kBlock - 1) < sr0) | (x0 > sr2)) {
                sMiss += (uint32_t)__popcll(__ballot(valid));
                T = Tnext;
                continue;
            }
        } else {
            valid = nextMeta.valid;
            tileCls = nextMeta.cls;
        }

        // (per-lane flags that are set deep inside the divergent code and read after it are ints: as bools they would live in
        // SGPR pairs and cost three mask operations at every join on the way out)
        // bit 0: the path goes on (a survivor), 1: it ended on an emitter, 2: it missed everything, 3: it ended at its scatter (certain to miss everything)
        uint32_t fl = 0u;
        uint32_t smallCandI = 1u;                               // class bit 3 of a survivor
        uint32_t wallSel = 8u;                                  // class bits 0-2 of a survivor in a scene with walls (8: no walls: the octant)
        // (a path's state is only ever read for the lanes that hold one -- and the record of the nearest hit only after a hit: whatever the
        // registers happen to hold will do for the others.  Spelled as an empty asm's output, which costs no instruction and, unlike an
        // uninitialised variable, is a definite value to the optimiser; the zeros they used to start from were ~25 v_mov per tile.)
        F3 org = anyF3(), dir = anyF3(), col = anyF3();
        int pix = __float_as_int(anyFloat());                   // (FIRST; later bounces unpack it where it is needed: an emitter hit)
        uint32_t pkCur = __float_as_uint(anyFloat());           // later bounces: pixelIndex | batch index << pixBits, as loaded (ONE register
                                                                // across the tile; the iteration is shifted out where it is needed)
        auto iterOf = [&]() -> int { return FIRST ? itb : (int)(pkCur >> (uint32_t)launder(kargs)->prm.pixBits); };
        uint32_t pixHash = __float_as_uint(anyFloat());         // utilhash(pix): FIRST computes it (the camera jitter's and the scatter's engines share it), later bounces load it
        if (valid) {
            // Camera rays (FIRST): a primitive is reachable only from the pixels inside the projection of its bounding
            // cube (GeomDev::rect, 2-pixel margin >> any rounding of the reference's tests), and nothing is reachable
            // outside the union of those rectangles -- at 16:9 more than half of Cornell's camera rays, which are not
            // even generated: they are misses whatever their jitter.
            bool inScene = true;
            if (FIRST) {
                const ArgsPtr A = launder(kargs);
                const PT_CAS KParams &prm = A->prm;
                pix = px + py * prm.W;
                inScene = px >= prm.sceneRect[0] && px <= prm.sceneRect[2] && py >= prm.sceneRect[1] && py <= prm.sceneRect[3];
                if (inScene) {
                    // camera ray (spec S2), jitter from the depth-0 stream of (iteration, pixel)
                    pixHash = utilhash((uint32_t)pix);              // (once: the scatter's engine below is keyed on the same pixel)
                    Rng rng = seedEngine(s_iterHash[PT_MAX_BATCH + itb] ^ pixHash);
                    const float jx = u01(rng);
                    const float jy = u01(rng);
                    const float sx = ((float)px + jx) - prm.halfW;
                    const float sy = ((float)py + jy) - prm.halfH;
                    const float a = prm.pixLenX * sx;
                    const float b = prm.pixLenY * sy;
                    const F3 view = f3(prm.view[0], prm.view[1], prm.view[2]);
                    const F3 up = f3(prm.up[0], prm.up[1], prm.up[2]);
                    const F3 right = f3(prm.right[0], prm.right[1], prm.right[2]);
                    org = f3(prm.pos[0], prm.pos[1], prm.pos[2]);
                    dir = normalize((view - right * a) - up * b);
                    if (DOF) {
                        // the pinhole ray fixes the point in focus; the ray starts at a uniformly sampled point of the lens
                        // disc (two more draws of the depth-0 stream) and aims at it
                        const float lr = prm.lensRadius * __builtin_sqrtf(u01(rng));
                        const float phi = u01(rng) * kTwoPi;
                        float s, c;
                        sincosPoly(phi, s, c);
                        const float ft = prm.focalDistance / dot(dir, f3(prm.viewN[0], prm.viewN[1], prm.viewN[2]));
                        const F3 focus = org + dir * ft;
                        org = (org + right * (lr * c)) + up * (lr * s);
                        dir = normalize(focus - org);
                    }
                }
                col = f3(1.0f, 1.0f, 1.0f);
            } else {
                org = cur.org; dir = cur.dir; col = cur.col;        // requested one tile ahead (loadTile)
                // The pixel index and the packed word outlive the next tile's loads (the stores need them), so they cannot stay in the
                // register tuple array C's element is loaded into: their copies out of it are made HERE, where the data has long
                // arrived.  (Left to itself the register allocator copied the NEXT tile's pixel index right behind its load -- v_mov
                // behind s_waitcnt vmcnt(2) -- and every wave sat out a memory round trip BEFORE the compaction's barriers instead of
                // under them: the "3 loads issued: 7 %" of round 3's timeline.)
                uint32_t chash = cur.pixHash, cpk = cur.pk;
                if (!MANY) asm volatile("" : "+v"(chash), "+v"(cpk));      // (sphere-heavy variants: no register to spare for the copies)
                pixHash = chash;
                pkCur = cpk;
            }

            if (FIRST) { if (inScene) probe(8); } else probe(7);                                   // tiles (waves with at least one valid path) and valid paths
            // nearest hit, geoms in file order, strict '<' so the first geom wins ties (S3)
            probe(15);                                          // (nearest-hit loop)
            float tbest = anyFloat();
            int hit = -1;
            F3 P = anyF3(), nsrc = anyF3();
            // (an int, not a bool: a loop-carried per-lane flag would live in an SGPR pair and cost three mask operations at
            // every merge point of the loop over the primitives; as a VGPR it costs one select per update)
            int outsideI = __float_as_int(anyFloat());
            const float dd = dot(dir, dir);
            int nCand = 0;                                       // MANY: spheres recorded by this lane
            float *s_sph = nullptr;                              // MANY: [ngeoms][kSphRowFloats], then [kListMax][kBlock] lists
            uint16_t *s_list = nullptr;
            if (MANY && !GROUPS) {
                const ArgsPtr A = launder(kargs);
                s_sph = S_SPH(A->prm.nmats, A->prm.ngeoms, A->prm.nCubes);
                s_list = reinterpret_cast<uint16_t *>(s_sph + (size_t)A->prm.ldsRowFloats);
            } else if (GROUPS && FIRST) {                        // (no scene table in LDS: the lanes' candidate lists follow the materials)
                s_list = reinterpret_cast<uint16_t *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * launder(kargs)->prm.nmats);
            }
            // the matrix row of primitive g for a per-lane test: from the LDS table -- or, in a scene of hundreds of primitives whose rows
            // would leave one workgroup per CU, from their copy in global memory (KParams::ldsRowFloats; wave-uniform)
            // (spheres only -- !CUBES: the 16-byte words 0-2, inverseTransform, and 6, GeomDev::invZ, are what a test needs before it knows that it
            // hits; words 3-5, the transform, are read by the lanes that do: loadRowXf.  Four per-lane 16-byte reads per candidate instead of seven)
            auto loadRow = [&](int g, float (&m)[28]) {
                const ArgsPtr A = launder(kargs);
                if (!GROUPS && A->prm.ldsRowFloats != 0) {
                    const float4 *row = reinterpret_cast<const float4 *>(s_sph + g * kSphRowFloats);
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
                        if (!CUBES && q >= 3 && q < 6) continue;
                        const float4 v = row[q];
                        m[4 * q] = v.x; m[4 * q + 1] = v.y; m[4 * q + 2] = v.z; m[4 * q + 3] = v.w;
                    }
                } else {
                    const float4 *row = A->rows + (size_t)g * 7;
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
                        if (!CUBES && q >= 3 && q < 6) continue;
                        const float4 v = row[q];
                        m[4 * q] = v.x; m[4 * q + 1] = v.y; m[4 * q + 2] = v.z; m[4 * q + 3] = v.w;
                    }
                }
            };
            auto loadRowXf = [&](int g, float (&x)[12]) {
                const ArgsPtr A = launder(kargs);
                if (!GROUPS && A->prm.ldsRowFloats != 0) {
                    const float4 *row = reinterpret_cast<const float4 *>(s_sph + g * kSphRowFloats) + 3;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const float4 v = row[q];
                        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                    }
                } else {
                    const float4 *row = A->rows + (size_t)g * 7 + 3;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const float4 v = row[q];
                        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                    }
                }
            };
            // Sphere-heavy scenes, later bounces: the spheres come from their packed culling data (order does not matter: the
            // nearest hit is taken by (distance, file order)), AFTER the loop over the primitives that are not spheres.
            constexpr bool PACKED = MANY && !FIRST;
            if (inScene) {
                const ArgsPtr A = launder(kargs);
                // a later tile looks at the list of its class; a camera-ray tile (256 pixels of one image row) at the list of
                // its row -- the primitives whose pixel rectangle covers that row -- and their rectangles sort the lanes out
                int gk0, gk1;
                const PT_CAS int *classIdx;
                typedef int int2v __attribute__((ext_vector_type(2)));
                const PT_CAS int2v *rowList = nullptr;               // camera rays of a whole-tile row: {primitive, its pixel span in this row}
                bool listed = !FIRST;
                if (FIRST) {
                    gk0 = 0; gk1 = A->prm.ngeoms;
                    classIdx = (const PT_CAS int *)(A->rowIdx);
                    if (!DOF && A->rowOff != nullptr) {
                        const PT_CAS int *rowOff = (const PT_CAS int *)(A->rowOff);
                        const int row = __builtin_amdgcn_readfirstlane(py);
                        gk0 = rowOff[row]; gk1 = rowOff[row + 1];
                        rowList = (const PT_CAS int2v *)(A->rowIdx);
                        listed = true;
                    }
                } else {
                    gk0 = A->prm.classOff[tileCls]; gk1 = A->prm.classOff[tileCls + 1u];
                    classIdx = (const PT_CAS int *)(A->classIdx);
                }
                const GeomPtr geoms = (GeomPtr)(A->ggeoms);
                const bool earlyMiss = FIRST || hotWalls(hotNow()) == 0u || (tileCls & 7u) >= 6u;    // (wave-uniform)
                // a later tile whose class lists ONE primitive (its paths' one wall): a hit's distance is compared with nothing (PACKED: the spheres follow)
                const bool onlyOne = !FIRST && !PACKED && gk1 - gk0 == 1;
                for (int gk = gk0; gk < gk1; ++gk) {
                    int g, span = 0;
                    if (FIRST && listed) {
                        const int2v e = rowList[gk];
                        g = e.x; span = e.y;
                    } else {
                        g = listed ? classIdx[gk] : gk;
                    }
                    // (sphere-heavy scenes: no laundering per primitive -- the camera-ray bounce still walks all seventy of them,
                    // and the compiler's own scheduling of the scalar loads across iterations is worth more than the registers
                    // it costs; measured on C5)
                    const PT_CAS GeomDev &G = *((MANY ? geoms : launder(geoms)) + g);
                    const CullGroup cg = loadCull<MANY>(&G);
                    const int flags = cg.cullFlags;
                    F3 p, n;
                    bool o = false;
                    int fm = 0;                                      // MESH: 1 + the material of the face that was hit, if it has its own
                    float t = -1.0f;
                    // camera rays: only the lanes whose pixel lies in the primitive's rectangle take the test.  A predicate and a
                    // wave-uniform skip, not a per-lane `continue`: the loop over the primitives stays a scalar loop
                    bool inRect = true;
                    // (testing the WAVE's pixel span against the rectangle in scalar registers instead -- a wave of a camera-ray tile
                    // is 64 consecutive pixels of one row -- saved 2 % of the vector instructions and cost 20 % more scalar ones:
                    // 1.5 % slower, not kept)
                    if (FIRST && !DOF) {
                        if (listed) inRect = (px >= (span & 0xffff)) & (px <= (span >> 16));   // (the list is this row's)
                        else inRect = (px >= G.rect[0]) & (px <= G.rect[2]) & (py >= G.rect[1]) & (py <= G.rect[3]);
                        if (__ballot(inRect) == 0ull) continue;
                    }
                    // (later bounces: the class's list holds no binned primitive its paths certainly miss, and of the walls only
                    // the one they can still hit)
                    if (!inRect) {
                        // (not reachable from this pixel: t stays -1)
                    } else if (MESH && (flags & 32) != 0) {
                        // a triangle mesh: the walks ran AHEAD of this launch (k_mesh_walk: rays from many tiles side by side, a lane that is
                        // through with one takes the next), and the path's record names the winning triangle among the meshes its tile lists
                        const ArgsPtr A2 = launder(kargs);
                        const unsigned long long rec = A2->meshHit[FIRST ? recIdx0 + tid : nextMeta.idx];
                        const uint32_t unit = (uint32_t)rec >> 1;
                        if (rec != ~0ull && unit >= G.meshUnit0 && unit < G.meshUnit1)
                            t = meshWinner<FIRST && !DOF>(G, A2->meshRecs, unit, ((uint32_t)rec & 1u) != 0u, org, dir, p, n, o, fm);
                    } else if (!PACKED && ((flags & 1) == 0 || (MANY && (flags & 64) != 0))) {    // (PACKED: no swept primitive comes this way)
                        probe(3);
                        // a sphere -- or, in a scene with many small primitives, a small cube (flag bit 6): the bounding ball first;
                        // MANY: recorded in the lane's list, tested after the loop with the lane's own matrices
                        if (!certainMiss(cg, org, dir, dd)) {
                            if (MANY && nCand < kListMax) {
                                s_list[nCand * kBlock + tid] = (uint16_t)g;
                                ++nCand;
                            } else if (MANY && (flags & 1) != 0) {   // (MANY: the lane's list is full -- test in place)
                                t = boxIntersectionTest<false, FIRST && !DOF>(G, org, dir, p, n, o);
                            } else {
                                t = sphereIntersectionTest<FIRST && !DOF>(G, org, dir, p, n, o);
                            }
                        }
                    } else {
                        // (the exact early miss pays where whole waves take it: not for camera rays inside a primitive's hull span -- 97 %
                        // of their box tests reach the hit phase -- nor in a tile whose class names the ONE wall its paths can still hit)
                        t = boxIntersectionTest<!FIRST, FIRST && !DOF>(G, org, dir, p, n, o, earlyMiss, onlyOne);
                    }
                    // (PACKED: a sphere tested in place above may hold the record with a higher index: file order decides a tie)
                    if (t > 0.0f && (hit < 0 || t < tbest || (PACKED && t == tbest && g < hit))) {
                        // (bit 0: the hit is on the outside; MESH: the bits above it carry the face's own material, 1 + its index)
                        tbest = t; hit = g; P = p; nsrc = n; outsideI = (o ? 1 : 0) | (MESH ? fm << 1 : 0);
                    }
                }
            }
            // ONE swept primitive with the lane's own matrices (an LDS row: inverseTransform 12, transform 12, GeomDev::invZ 3): the sphere
            // test -- or, where the scene's swept primitives include cubes (kHotSweptCubes) and this one is one, the box test
            auto sweptTest = [&](int g, const float (&m)[28], F3 ro, F3 rdir, F3 &p, F3 &n, bool &o) -> float {
                if (CUBES) {
                    const ArgsPtr A = launder(kargs);
                    // (GROUPS: the hit records are not staged -- the image pt_init made of them in global memory, BounceArgs::ghit)
                    const int type = GROUPS ? reinterpret_cast<const GeomHitSmall *>(A->ghit)[g].type : S_GEOMHIT_SMALL(A->prm.nmats)[g].type;
                    if (type == 1) {
                        struct Rows { const float *inv, *invZ, *xf, *camObj; } rows = {m, m + 24, m + 12, m};
                        return boxIntersectionTest<false, false>(rows, ro, rdir, p, n, o);
                    }
                }
                if (CUBES) return sphereIntersectionTestM<false>(m, m + 24, m + 12, m, ro, rdir, p, n, o);
                return sphereIntersectionTestLazy<false>(m, m + 24, [&](float (&x)[12]) { loadRowXf(g, x); }, m, ro, rdir, p, n, o);
            };
            // the primitives a lane recorded: pass k tests every lane's k-th one with that lane's own matrices from LDS
            auto candidatePass = [&]() {
                for (int k = 0; __ballot(k < nCand) != 0ull; ++k) {          // wave-uniform trip count
                    if (k < nCand) {
                        const int g = s_list[k * kBlock + tid];
                        float m[28];
                        loadRow(g, m);
                        F3 p, n;
                        bool o = false;
                        probe(4);
                        const float t = sweptTest(g, m, org, dir, p, n, o);
                        // a recorded sphere may precede, in file order, the primitive that holds the record so far
                        if (t > 0.0f && (hit < 0 || t < tbest || (t == tbest && g < hit))) {
                            tbest = t; hit = g; P = p; nsrc = n; outsideI = o ? 1 : 0;
                        }
                    }
                }
            };
            if (MANY && !PACKED) candidatePass();
            if (PACKED && GROUPS) {
                // TWO-LEVEL sweep (scenes of hundreds of swept primitives: the flat sweep below is linear in their number -- 512 spheres cost
                // 3.5 x what 64 did).  Level 1: the groups' bounding balls, wave-uniform through the scalar path like the flat sweep's entries,
                // the outcomes shifted into per-lane masks.  Level 2: every lane takes ITS next candidate group and runs the members' own
                // certificates -- the flat sweep's, on the same operands: entries read per lane from the table in global memory -- then the
                // reference's test on the members that remain.  A group's ball holds every member's certificate ball for origins within
                // grpOMax (pt_init: build_sphere_groups), so a half-line that misses it passes every member's certificate: the members
                // skipped are certified misses, the candidates the flat sweep's own -- same hits, same (distance, file order) winner.
                probe(24);
                const ArgsPtr A = launder(kargs);
                int gN = A->prm.nSphGroups, gG0 = 0;
                if (CLUSTER) {                                       // (the classes' cluster bits, as for the flat sweep)
                    if (hotNow() & kHotReadsLastBits) {
                        gN = (tileCls & 16u) ? gN : 0;
                    } else {
                        gN = (tileCls & 16u) ? gN : A->prm.grpN0;
                        gG0 = (tileCls & 8u) ? 0 : A->prm.grpN0;
                    }
                }
                const PT_CAS SphereCull *gt = (const PT_CAS SphereCull *)(A->sphGroups);
                const float4 *const ent = reinterpret_cast<const float4 *>(A->sphCull);    // entry e: ent[2 e] = {centre, cullR2}, ent[2 e + 1] = {cullK, geom, -, -}
                const F3 dhat = unitDirectionScaled(dir, dd, A->prm.sphDirScale);
                // (an origin beyond the bound the groups' balls were built for -- none of a scattered ray's -- takes every group)
                const bool nearO = (__builtin_fabsf(org.x) + __builtin_fabsf(org.y)) + __builtin_fabsf(org.z) <= A->prm.grpOMax;
                for (int base = gG0; base < gN; base += 64) {         // (wave-uniform: rounds of 64 groups)
                    uint32_t gHi = 0u, gLo = 0u;                      // group base + j: bit 31 - j of gHi (j < 32) / of gLo
                    if (inScene) {
                        auto sweepG = [&](int k0, uint32_t &m) {
                            const int n = min(32, gN - k0);
                            if (n <= 0) return;
                            for (int k = 0; k < n; k += 2) {           // (two per 64-byte scalar load; the table is padded to an even count)
                                int16v v;
                                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(gt + k0 + k) : "memory");
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    if (h == 1 && k + 1 >= n) break;
                                    const float x = sphereHalfLineExcessScaled(f3(__int_as_float(v[8 * h]), __int_as_float(v[8 * h + 1]), __int_as_float(v[8 * h + 2])), org, dhat);
                                    asm("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "s"(__int_as_float(v[8 * h + 3])), "v"(x) : "vcc");
                                }
                            }
                            m <<= (32 - n);
                            if (!nearO) m = 0xffffffffu << (32 - n);
                        };
                        sweepG(base, gHi);
                        sweepG(base + 32, gLo);
                    }
                    probe(25);
                    // level 2, lane by lane; `E16(e)` = entry e's {centre, threshold}, `GEOM(e)` = its primitive: from LDS (the common case) or global memory.
                    // A lane's candidates are PARKED -- one word per group that has any: group << 8 | its members' outcome bits, in the lane's
                    // column of an LDS list -- and tested when every lane is through with its groups: the passes then number the busiest lane's
                    // candidates in all (3-4), where testing group by group cost a pass per group turn with a handful of lanes at work.
                    auto level2 = [&](auto E16, auto GEOM) {
                        uint32_t *const s_cand = reinterpret_cast<uint32_t *>(smem + A->prm.pairOff) + tid;      // [kCandPairs][kBlock]
                        uint32_t nP = 0u;                             // words parked by this lane
                        auto passes = [&]() {
                            uint32_t rd = 0u, cur = 0u;               // the word being consumed: group << 8 | bits left
                            for (;;) {
                                if ((cur & 0xffu) == 0u && rd < nP) { cur = s_cand[rd * kBlock]; ++rd; }
                                const bool has = (cur & 0xffu) != 0u;
                                if (__ballot(has) == 0ull) break;     // (wave-uniform: the busiest lane's candidates)
                                if (has) {
                                    const int j = __builtin_clz(cur << 24);           // member 0 at bit 7
                                    cur &= ~(0x80u >> j);
                                    const int g = GEOM((int)(cur >> 8) * kSphGroupSize + j);
                                    float mr[28];
                                    loadRow(g, mr);
                                    F3 p, n;
                                    bool o = false;
                                    probe(4);
                                    const float t = sweptTest(g, mr, org, dir, p, n, o);
                                    if (t > 0.0f && (hit < 0 || t < tbest || (t == tbest && g < hit))) {
                                        tbest = t; hit = g; P = p; nsrc = n; outsideI = o ? 1 : 0;
                                    }
                                }
                            }
                            nP = 0u;
                        };
                        while (__ballot((gHi | gLo) != 0u) != 0ull) { // wave-uniform trip count: the busiest lane's candidate groups
                            if (__ballot(nP >= (uint32_t)kCandPairs) != 0ull) passes();      // (a lane's column is full: everybody tests what is parked)
                            if ((gHi | gLo) != 0u) {
                                const bool hi = gHi != 0u;
                                const uint32_t mm = hi ? gHi : gLo;
                                const int j = __builtin_clz(mm);
                                const uint32_t rest = mm & ~(0x80000000u >> j);
                                gHi = hi ? rest : gHi;
                                gLo = hi ? gLo : rest;
                                const int grp = base + (hi ? 0 : 32) + j;
                                const int e0 = grp * kSphGroupSize;
                                uint32_t m = 0u;                      // member i: bit 7 - i
#pragma unroll
                                for (int q = 0; q < kSphGroupSize; q += 4) {       // (four loads in flight: sixteen registers, not thirty-two)
                                    float4 c[4];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) c[i] = E16(e0 + q + i);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        probe(3);
                                        const float x = sphereHalfLineExcessScaled(f3(c[i].x, c[i].y, c[i].z), org, dhat);
                                        m = (m << 1) | (!(c[i].w < x) ? 1u : 0u);  // candidate = !(cullR2 < x) (NaN: candidate)
                                    }
                                }
                                if (m != 0u) { s_cand[nP * kBlock] = ((uint32_t)grp << 8) | m; ++nP; }
                            }
                        }
                        passes();
                    };
                    if (A->prm.grpLds) {
                        const float4 *const e16 = reinterpret_cast<const float4 *>(smem + kMiscWords * sizeof(uint32_t) + sizeof(MaterialDev) * A->prm.nmats);
                        const uint16_t *const emap = reinterpret_cast<const uint16_t *>(e16 + A->prm.nSphCull);
                        level2([&](int e) { return e16[e]; }, [&](int e) { return (int)emap[e]; });
                    } else {
                        level2([&](int e) { return ent[2 * e]; }, [&](int e) { return __float_as_int(ent[2 * e + 1].y); });
                    }
                }
            }
            if (PACKED && !GROUPS) {
                // The sweep keeps no list: the outcome of a sphere's bounding-ball test (sphereHalfLineExcess, pt_device.h) is SHIFTED into a per-lane bit mask (compare,
                // then add-with-carry m = m + m + vcc: two instructions, no branch, no exec-mask change, no LDS), 64 spheres per
                // round; the candidate passes then take each lane's set bits from the top.  (Rounds 1-3 recorded up to eight
                // candidates per lane in LDS lists: ~8 vector instructions, two exec-mask regions and two branches per sphere on top
                // of the 15 of the test.)  Two spheres per 64-byte scalar load.
                probe(24);                                          // (the sweep of the packed spheres)
                const ArgsPtr A = launder(kargs);
                // (even: the host pads with a copy of the last sphere.)  CLUSTER: entries [0, sphN0) are cluster 0, the tile's class bit 3,
                // [sphN0, nSphCull) cluster 1, bit 4 -- a tile sweeps the clusters its paths can hit (wave-uniform bounds)
                // (the queue that enters the last bounce: bit 4 = either cluster, bit 3 = the binned primitives alone -- kHotReadsLastBits)
                int nS = A->prm.nSphCull, kS0 = 0;
                if (CLUSTER) {
                    if (hotNow() & kHotReadsLastBits) {
                        nS = (tileCls & 16u) ? nS : 0;
                    } else {
                        nS = (tileCls & 16u) ? nS : A->prm.sphN0;
                        kS0 = (tileCls & 8u) ? 0 : A->prm.sphN0;
                    }
                }
                const PT_CAS SphereCull *sc = (const PT_CAS SphereCull *)(A->sphCull);
                const uint16_t *const sphGeom = s_list;               // [nS]: the primitive behind entry k (prologue)
                const F3 dhat = unitDirectionScaled(dir, dd, A->prm.sphDirScale);
                for (int base = kS0; base < nS; base += 64) {         // (wave-uniform)
                    uint32_t mHi = 0u, mLo = 0u;                      // entry base + j: bit 31 - j of mHi (j < 32) / of mLo
                    if (inScene) {
                        auto sweep32 = [&](int k0, uint32_t &m) {
                            const int n = min(32, nS - k0);           // (even)
                            if (n <= 0) return;
                            for (int k = 0; k < n; k += 2) {
                                int16v v;
                                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(sc + k0 + k) : "memory");
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    probe(3);
                                    const float x = sphereHalfLineExcessScaled(f3(__int_as_float(v[8 * h]), __int_as_float(v[8 * h + 1]), __int_as_float(v[8 * h + 2])), org, dhat);
                                    // candidate = !(cullR2 < x) (NaN: candidate), shifted in from below
                                    asm("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "s"(__int_as_float(v[8 * h + 3])), "v"(x) : "vcc");
                                }
                            }
                            m <<= (32 - n);                           // (wave-uniform shift: entry k0 + j at bit 31 - j whatever n)
                        };
                        sweep32(base, mHi);
                        sweep32(base + 32, mLo);
                        PT_EXP_SWEEP_TWICE(base, mHi, mLo, org)         // (experiment builds only)
                    }
                    probe(25);                                      // (the candidates' passes)
                    // POOLED pass.  A pass of the loop below runs for the lanes that still hold a candidate -- a quarter of a wave in C5
                    // (profiles/probe_phases.py: 13.9 lanes per pass, 1.6 passes per tile) -- and a wave takes as many passes as its
                    // busiest lane has candidates.  When some lane has two or more and the wave's candidates are no more than its running lanes, ALL of
                    // them are tested in ONE pass instead: every (ray, sphere) pair is handed to a lane of its own -- the pairs numbered
                    // level by level (a ballot per level: the lanes with at least l candidates), the descriptor {owner lane, entry} through
                    // LDS, the owner's ray through ds_bpermute -- and each owner collects its pairs' outcomes the same way, nearest first,
                    // file order on a tie, exactly as the passes would have.  Same arithmetic on another lane: bit-identical.
                    {
                        const uint32_t cnt = (uint32_t)(__popc(mHi) + __popc(mLo));
                        if (__ballot(cnt >= 2u) != 0ull) {              // (wave-uniform)
                            const ArgsPtr A2 = launder(kargs);
                            uint32_t *const s_pair = reinterpret_cast<uint32_t *>(smem + A2->prm.pairOff) + (tid & ~63u);   // [kWaves][64]
                            const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                            uint32_t total = 0u, hi = mHi, lo = mLo;
                            for (uint32_t l = 1u;; ++l) {               // (wave-uniform trip count: the busiest lane's candidates)
                                const unsigned long long b = __ballot(cnt >= l);
                                if (b == 0ull) break;
                                if (cnt >= l) {
                                    const bool useHi = hi != 0u;
                                    const uint32_t mm = useHi ? hi : lo;
                                    const int j = __builtin_clz(mm);
                                    const uint32_t rest = mm & ~(0x80000000u >> j);
                                    hi = useHi ? rest : hi;
                                    lo = useHi ? lo : rest;
                                    const uint32_t slot = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                                    if (slot < 64u) s_pair[slot] = ln | ((uint32_t)((useHi ? 0 : 32) + j) << 8);
                                }
                                total += (uint32_t)__popcll(b);
                            }
                            // (a pair's lane must be one that runs: a tile's valid lanes are a prefix of it, so of every wave)
                            if (total <= (uint32_t)__popcll(__ballot(true))) {   // (wave-uniform)
                                const uint32_t desc = ln < total ? s_pair[ln] : ln;
                                const int src = (int)((desc & 63u) << 2);
                                const F3 oorg = f3(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(org.x))),
                                                   __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(org.y))),
                                                   __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(org.z))));
                                const F3 odir = f3(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dir.x))),
                                                   __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dir.y))),
                                                   __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dir.z))));
                                float te = -1.0f;
                                int ge = 0, oe = 0;
                                F3 pe = f3(0, 0, 0), ne = f3(0, 0, 0);
                                if (ln < total) {
                                    ge = sphGeom[base + (int)(desc >> 8)];
                                    float m[28];
                                    loadRow(ge, m);
                                    bool o = false;
                                    probe(4);
                                    te = sweptTest(ge, m, oorg, odir, pe, ne, o);
                                    oe = o ? 1 : 0;
                                }
                                // every owner collects its pairs: the same numbering, level by level
                                float tb = -1.0f;
                                int gb = 0;
                                uint32_t sb = ln, off = 0u;
                                for (uint32_t l = 1u;; ++l) {
                                    const unsigned long long b = __ballot(cnt >= l);
                                    if (b == 0ull) break;
                                    const uint32_t slot = cnt >= l ? off + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u)) : ln;
                                    const float tt = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(slot << 2), __float_as_int(te)));
                                    const int gg = __builtin_amdgcn_ds_bpermute((int)(slot << 2), ge);
                                    if (cnt >= l && tt > 0.0f && (!(tb > 0.0f) || tt < tb || (tt == tb && gg < gb))) { tb = tt; gb = gg; sb = slot; }
                                    off += (uint32_t)__popcll(b);
                                }
                                const int ws = (int)(sb << 2);
                                const F3 pw = f3(__int_as_float(__builtin_amdgcn_ds_bpermute(ws, __float_as_int(pe.x))),
                                                 __int_as_float(__builtin_amdgcn_ds_bpermute(ws, __float_as_int(pe.y))),
                                                 __int_as_float(__builtin_amdgcn_ds_bpermute(ws, __float_as_int(pe.z))));
                                const F3 nw = f3(__int_as_float(__builtin_amdgcn_ds_bpermute(ws, __float_as_int(ne.x))),
                                                 __int_as_float(__builtin_amdgcn_ds_bpermute(ws, __float_as_int(ne.y))),
                                                 __int_as_float(__builtin_amdgcn_ds_bpermute(ws, __float_as_int(ne.z))));
                                const int ow = __builtin_amdgcn_ds_bpermute(ws, oe);
                                if (tb > 0.0f && (hit < 0 || tb < tbest || (tb == tbest && gb < hit))) {
                                    tbest = tb; hit = gb; P = pw; nsrc = nw; outsideI = ow;
                                }
                                mHi = 0u; mLo = 0u;
                            }
                        }
                    }
                    // pass k tests every lane's k-th candidate with that lane's own matrices from LDS
                    while (__ballot((mHi | mLo) != 0u) != 0ull) {     // wave-uniform trip count
                        if ((mHi | mLo) != 0u) {
                            const bool hi = mHi != 0u;
                            const uint32_t mm = hi ? mHi : mLo;
                            const int j = __builtin_clz(mm);
                            const uint32_t rest = mm & ~(0x80000000u >> j);
                            mHi = hi ? rest : mHi;
                            mLo = hi ? mLo : rest;
                            const int g = sphGeom[base + (hi ? 0 : 32) + j];
                            float m[28];
                            loadRow(g, m);
                            F3 p, n;
                            bool o = false;
                            probe(4);
                            const float t = sweptTest(g, m, org, dir, p, n, o);
                            // nearest by (distance, file order): a sphere may precede the primitive that holds the record so far
                            if (t > 0.0f && (hit < 0 || t < tbest || (t == tbest && g < hit))) {
                                tbest = t; hit = g; P = p; nsrc = n; outsideI = o ? 1 : 0;
                            }
                        }
                    }
                }
            }
            probe(16);                                          // (shading)
            if (hit < 0) {
                fl = 4u;                                         // S4: background is black
            } else {
                probe(9);
                // per-lane primitive: LDS lookup of its hit record (sphere-heavy scenes: the compact record + the frame table)
                const int faceMat = MESH ? (outsideI >> 1) : 0;    // a mesh face with a material of its own (`usemtl`): 1 + its index
                int ghType, ghMaterial;
                const float *ghNm, *ghFrame;
                float mEmit, mRefl, mRefr;                       // the material's hot fields
                F3 mcolMany = f3(0, 0, 0);
                const float4 *hrec1 = nullptr;                   // small scenes: the record's second 16 bytes {material colour, material index}
                if (MANY) {
                    const ArgsPtr A = launder(kargs);
                    const GeomHitSmall &h = GROUPS ? reinterpret_cast<const GeomHitSmall *>(A->ghit)[hit] : S_GEOMHIT_SMALL(A->prm.nmats)[hit];
                    ghType = h.type; ghMaterial = h.material; ghNm = h.nm;
                    ghFrame = (GROUPS ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(A->ghit) + manyHitBytes(A->prm.ngeoms))
                                      : S_FRAMES(A->prm.nmats, A->prm.ngeoms)) + h.frame * 54;
                    if (MESH && faceMat != 0) ghMaterial = faceMat - 1;
                    const MaterialDev &Mm = smats[ghMaterial];
                    mEmit = Mm.emittance; mRefl = Mm.hasReflective; mRefr = Mm.hasRefractive;
                    mcolMany = f3(Mm.color[0], Mm.color[1], Mm.color[2]);
                } else {
                    // the record's header: type and the material's three switches now; its colour and index are read where they are
                    // used (fetched here with the rest they sat in four registers across the normal's evaluation -- one too many: the
                    // allocator parked them in scratch -- while a second LDS read costs one instruction and hides behind the engine's hash)
                    const GeomHitDev &h = S_GEOMHIT(hotMats(hotNow()))[hit];
                    const float4 h0 = reinterpret_cast<const float4 *>(&h)[0];
                    hrec1 = reinterpret_cast<const float4 *>(&h) + 1;
                    ghType = __float_as_int(h0.x); mEmit = h0.y; mRefl = h0.z; mRefr = h0.w;
                    ghMaterial = 0;
                    if (MESH && faceMat != 0) {                      // (the face's material instead of the record's copy of the object's)
                        const MaterialDev &Mf = smats[faceMat - 1];
                        mEmit = Mf.emittance; mRefl = Mf.hasReflective; mRefr = Mf.hasRefractive;
                    }
                    ghNm = h.nm; ghFrame = h.cubeFrame;
                }
                const bool isSphere = ghType == 0;
                bool faceOk = true;
                const int face = isSphere ? 0 : cubeFace(nsrc, faceOk);
                const bool outside = (outsideI & 1) != 0;
                // a cube face's frame (normal + the sampler's two tangents, nine floats): ONE select on the address -- the face's
                // row of the table, or the row of NaNs for a hit without an exit slab -- instead of nine on the values
                // (GROUPS: the frames live in global memory, and so does their row of NaNs -- behind the last cube's frames, pt_init)
                const float *const nanRow = GROUPS ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(launder(kargs)->ghit) +
                                                                                     manyHitBytes(launder(kargs)->prm.ngeoms)) + launder(kargs)->prm.nCubes * 54
                                                   : s_nan;
                const float *const fv = faceOk ? ghFrame + 9 * face : nanRow;
                const F3 N = isSphere ? hitNormalSphere(ghNm, nsrc, outside) : f3(fv[0], fv[1], fv[2]);
                F3 mcol = mcolMany;
                if (!MANY) {
                    uint32_t off = (uint32_t)(reinterpret_cast<const unsigned char *>(hrec1) - smem);
                    asm volatile("" : "+v"(off));                // (read HERE, not hoisted to the header's read; the LDS offset, so that the
                                                                 // read stays a ds_read)
                    const float4 h1 = *reinterpret_cast<const float4 *>(smem + off);
                    mcol = f3(h1.x, h1.y, h1.z); ghMaterial = __float_as_int(h1.w);
                    if (MESH && faceMat != 0) {
                        ghMaterial = faceMat - 1;
                        const MaterialDev &Mf = smats[ghMaterial];
                        mcol = f3(Mf.color[0], Mf.color[1], Mf.color[2]);
                    }
                }
                const MaterialDev &M = smats[ghMaterial];       // (the fields of the rarer branches)
                if (mEmit > 0.0f) {                              // S5: emitter ends the path
                    fl = 2u;
                    const ArgsPtr A = launder(kargs);
                    float *const contrib = A->contrib;
                    if (hotNow() & kHotContrib) {
                        // Deferred accumulation: iterations overlap on several streams, so the radiance
                        // is parked in this iteration's own buffer (one path per pixel: race-free, no
                        // read) and k_commit adds it to the accumulator in iteration order.
                        const F3 c = (col * mcol) * mEmit;
                        // (index of the pixel in the radiance buffers: the frame's, or, for a row shard, the shard's own)
                        size_t frame = (size_t)A->prm.W * A->prm.H;
                        const uint32_t gpix = FIRST ? (uint32_t)pix : (pkCur & ((1u << (uint32_t)A->prm.pixBits) - 1u));
                        uint32_t cpix = gpix;
                        if (hotNow() & kHotContribLocal) {
                            const uint32_t y = fastDiv(gpix, A->prm.magicW, A->prm.shiftW);
                            const uint32_t lr = fastDiv(y, A->prm.magicS, A->prm.shiftS);
                            cpix = gpix - (y - lr) * (uint32_t)A->prm.W;               // x + lr * W
                            frame = (size_t)A->prm.nLocal;
                        }
                        const int itq = iterOf();
                        float *dst = contrib + 3 * ((size_t)itq * frame + (size_t)cpix);
                        dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
                        // ... and the pixel's mask says which iterations of the batch left something: k_commit then reads
                        // 4 B per pixel and word instead of 12 B per pixel and iteration (one path per pixel and iteration: the
                        // bits of a word come from different launches or lanes, hence the atomic; nobody waits for it)
                        (void)__hip_atomic_fetch_or(A->hitMask + (size_t)(itq >> 5) * frame + (size_t)cpix, 1u << (itq & 31), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else if (!(hotNow() & kHotLast)) {                  // S6 scatter (S7: skipped on the last bounce)
                    probe(10);
                    Rng rng = seedEngine(s_iterHash[iterOf()] ^ pixHash);   // = makeSeededRandomEngineHashed(., pix)
                    const F3 scol = f3(M.specColor[0], M.specColor[1], M.specColor[2]);
                    F3 ndir = dir, norg;
                    bool diffuse = false;                        // the hemisphere is sampled at one place, after the branches
                    if (!PLAIN && mRefr > 0.0f) {
                        const float eta = outside ? M.invIor : M.ior;
                        const float c = dot(N, dir);
                        const float k = 1.0f - eta * eta * (1.0f - c * c);
                        const float u = u01(rng);
                        bool doReflect = true;
                        if (k >= 0.0f) {
                            const float r0 = M.r0;
                            const float cosx = outside ? -c : __builtin_sqrtf(k);
                            const float w = 1.0f - cosx;
                            const float w2 = w * w;
                            const float w5 = w2 * w2 * w;
                            const float fres = r0 + (1.0f - r0) * w5;
                            doReflect = u < fres;
                        }
                        if (doReflect) {
                            ndir = reflect(dir, N);
                            norg = P + N * 0.001f;
                            col = col * scol;
                        } else {
                            ndir = refract(dir, N, eta);
                            norg = P - N * 0.001f;
                            col = col * mcol;
                        }
                    } else if (mRefl > 0.0f) {
                        // (the documented variant: 50 / 50 "divided by the probability" -- x 2 whichever branch is taken; exact, so it
                        // commutes with the colour products below.  Off by default: see PT_FLAG_MIXTURE_WEIGHTED)
                        if (!PLAIN && (hotNow() & kHotMixWeighted)) col = col * 2.0f;
                        const float u = u01(rng);
                        if (u < 0.5f) {
                            ndir = reflect(dir, N);
                            if (!PLAIN && M.invSpecExp1 > 0.0f) ndir = specularLobeDirection(ndir, N, M.invSpecExp1, rng);   // SPECEX > 0
                            col = col * scol;
                        } else {
                            diffuse = true;
                        }
                        norg = P + N * 0.001f;
                    } else {
                        diffuse = true;
                        norg = P + N * 0.001f;
                    }
                    bool toLight = false;
                    if (diffuse) {
                        // direct lighting (README.md:107-108): at the scene's last bounce the diffuse scatter is a ray to a uniformly
                        // chosen point of the (transformed) unit cube of a uniformly chosen emissive primitive, weighted by the
                        // cosine at the surface; the launch after this one collects what it hits
                        const ArgsPtr A = launder(kargs);
                        toLight = !PLAIN && (hotNow() & kHotToLight) != 0u;
                        if (toLight) {
                            const int ne = A->prm.nEmit;
                            int pick = (int)(u01(rng) * (float)ne);
                            pick = pick > ne - 1 ? ne - 1 : pick;
                            int e = A->prm.emitGeom[0];
                            float rho2 = A->prm.emitRho2[0];
#pragma unroll
                            for (int q = 1; q < kEmitMax; ++q) {
                                e = pick == q ? A->prm.emitGeom[q] : e;
                                rho2 = pick == q ? A->prm.emitRho2[q] : rho2;
                            }
                            const float ux = u01(rng) - 0.5f;
                            const float uy = u01(rng) - 0.5f;
                            const float uz = u01(rng) - 0.5f;
                            const float *xf = A->ggeoms[e].xf;              // per-lane primitive: vector loads
                            float m[12];
#pragma unroll
                            for (int q = 0; q < 12; ++q) m[q] = xf[q];
                            // a point of the emitter's object-space box, centre + u x extent (the unit cube: 0 + u x 1 = u); per-lane emitter:
                            // vector loads from the argument block
                            const PT_CAS float *bx = &A->prm.emitBox[0][0] + 6 * pick;
                            const F3 target = mulMV(m, f3(bx[0] + ux * bx[3], bx[1] + uy * bx[4], bx[2] + uz * bx[5]), 1.0f);
                            const F3 toward = target - norg;
                            ndir = normalize(toward);
                            float w = dot(N, ndir);
                            w = w > 0.0f ? w : 0.0f;
                            // ... and by the share of the hemisphere the emitter's bounding ball covers, min(1, rho^2 / r^2)
                            float cover = rho2 / dot(toward, toward);
                            cover = cover < 1.0f ? cover : 1.0f;
                            col = (col * mcol) * (w * cover);
                            // (the engine's state has been DEAD since the third draw: a ray to a light samples no hemisphere.  Said out loud, at the
                            // block's end, because the register allocator cannot see that `toLight` excludes the hemisphere below and parked the state
                            // in scratch across this block -- the scratch traffic inside the tile loop of the non-PLAIN later-bounce kernel,
                            // VERDICT round 5 weak #4; C4 itself never enters this block)
                            rng.x = __float_as_uint(anyFloat());
                        } else {
                            col = col * mcol;
                        }
                    }
                    if (diffuse && !toLight) {
                        probe(11);
                        float up, cOver, sOver;
                        hemisphereDraws(rng, up, cOver, sOver);
                        F3 p1, p2;                                // the sampler's tangent frame: computed for a sphere, looked up for a cube face
                        if (isSphere) {
                            hemisphereFrame(N, p1, p2);
                        } else {
                            p1 = f3(fv[3], fv[4], fv[5]);
                            p2 = f3(fv[6], fv[7], fv[8]);
                        }
                        ndir = hemisphereCombine(N, p1, p2, up, cOver, sOver);
                    }
                    org = norg;
                    dir = ndir;
                    fl = 1u;
                    {                                            // class bit 3: can the new ray hit a small primitive at all?
                        const int nBinned = (int)hotBinned(hotNow());
                        if (nBinned > 0) {
                            const float ndd = dot(ndir, ndir);
                            uint32_t cand = 0u;
                            for (int sI = 0; sI < nBinned; sI += 2) { probe(12);          // two at a time: one 64-byte scalar load
                                int16v v;
                                asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(launder(kargs)), "s"((int)(offsetof(BounceArgs, prm) + offsetof(KParams, binCull)) + sI * 32) : "memory");
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    // (sphere-heavy scenes bin one primitive, the light: the row behind it stands for nobody -- a scalar
                                    // branch instead of fourteen vector instructions per scatter)
                                    if (MANY && h == 1 && sI + 1 >= nBinned) break;
                                    CullGroup cg;
                                    cg.centre[0] = __int_as_float(v[8 * h]); cg.centre[1] = __int_as_float(v[8 * h + 1]); cg.centre[2] = __int_as_float(v[8 * h + 2]);
                                    cg.cullR2 = __int_as_float(v[8 * h + 3]); cg.cullK = __int_as_float(v[8 * h + 4]);
                                    // (the primitive's candidate bit: 1, or 2 for the mesh scenes' group 1 -- word 5 of its row)
                                    cand |= certainMiss(cg, norg, ndir, ndd) ? 0u : (WIDE ? (uint32_t)v[8 * h + 5] : 1u);
                                }
                            }
                            smallCandI = cand;
                        }
                    }
                    {                                            // class bits 0-2 with walls: which of them can the new ray still hit?
                        const ArgsPtr A = launder(kargs);
                        const int nWalls = (int)hotWalls(hotNow());
                        // CLUSTER: the sphere clusters too -- a slab certificate against each cluster's inflated box (the box holds every sphere's
                        // half-line ball, sqrt(cullR2 + K |oc|^2) around its centre for the origins certificates are issued for: a half-line
                        // that misses the box passes every sphere's own certificate with room to spare)
                        uint32_t cb = CLUSTER ? 3u : 0u;
                        if (nWalls > 0 || CLUSTER) {
                            wallSel = 6u;
                            const float l1 = (__builtin_fabsf(norg.x) + __builtin_fabsf(norg.y)) + __builtin_fabsf(norg.z);
                            const bool wallsOk = nWalls > 0 && l1 <= A->prm.wallOMax;          // (NaN fails)
                            const bool sphOk = CLUSTER && l1 <= A->prm.sphOMax;
                            if (wallsOk || sphOk) {
                                const F3 inv = f3(__builtin_amdgcn_rcpf(ndir.x), __builtin_amdgcn_rcpf(ndir.y), __builtin_amdgcn_rcpf(ndir.z));
                                if (sphOk) {
                                    int16v v;
                                    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(launder(kargs)), "s"((int)(offsetof(BounceArgs, prm) + offsetof(KParams, sphBox))) : "memory");
                                    cb = 0u;
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        WallBox wb;
                                        wb.lo[0] = __int_as_float(v[8 * h]); wb.lo[1] = __int_as_float(v[8 * h + 1]); wb.lo[2] = __int_as_float(v[8 * h + 2]);
                                        wb.hi[0] = __int_as_float(v[8 * h + 3]); wb.hi[1] = __int_as_float(v[8 * h + 4]); wb.hi[2] = __int_as_float(v[8 * h + 5]);
                                        cb |= wallCertainMiss(wb, norg, inv) ? 0u : (1u << h);
                                    }
                                }
                                if (CLUSTER) smallCandI = (hotBinned(hotNow()) > 0u ? smallCandI : 0u) | ((hotNow() & kHotWritesLastBits) ? (cb != 0u ? 2u : 0u) : cb);
                                if (wallsOk) {
                                    const WallPtr walls = (WallPtr)(A->walls);
                                    probe(13);
                                    uint32_t possible = wallPlanesPossible(A->prm, norg, ndir, inv);     // walls 0 .. nSlotWalls - 1
                                    // (rotated walls: the plane of their inner face, whatever its direction -- never in the PLAIN instantiations,
                                    // which pt_init does not pick for a scene that has one)
                                    const int nPl = PLAIN ? 0 : (int)hotPlaneWalls(hotNow());
                                    if (!PLAIN && nPl > 0) possible |= wallPlanesOriented(A->prm, norg, ndir, inv, (int)hotSlotWalls(hotNow()), nPl);
                                    for (int w = (int)hotSlotWalls(hotNow()) + nPl; w < nWalls; ++w) { probe(13);
                                        possible |= wallCertainMiss(*(launder(walls) + w), norg, inv) ? 0u : (1u << w); }
                                    const int cnt = __popc(possible);
                                    wallSel = cnt == 1 ? (uint32_t)(__ffs((int)possible) - 1) : (cnt == 0 ? 7u : 6u);
                                    // nothing left to hit: the reference's nearest-hit loop would come back empty at the next bounce.
                                    // The path ends here and is tallied as what it is, a path that entered that bounce and missed.
                                    // (Not under pt_debug_trace_paths, which shows the queue as the oracle lists it.)
                                    if (cnt == 0 && smallCandI == 0u && (hotNow() & (kHotAllClassified | kHotContrib)) == (kHotAllClassified | kHotContrib)) {
                                        fl = 8u;
                                    }
                                }
                            } else if (CLUSTER) {
                                smallCandI = (hotBinned(hotNow()) > 0u ? smallCandI : 0u) | ((hotNow() & kHotWritesLastBits) ? 2u : cb);
                            }
                        }
                    }
                }
            }
        }
        PT_EXP_TILE_LOAD(org, dir, pixHash, fl, kargs)          // (experiment builds only)
        probe(17);                                              // (next tile's loads)
        sLight += (uint32_t)__popcll(__ballot((fl & 2u) != 0u));
        sMiss += (uint32_t)__popcll(__ballot((fl & 4u) != 0u));
        sEarly += (uint32_t)__popcll(__ballot((fl & 8u) != 0u));
        if (!FIRST) {       // the next tile of this workgroup that needs work: its loads fly during the compaction below
            while (Tnext < numTiles && !setupTile(Tnext, tid, nextMeta)) Tnext += gridDim.x;
            if (Tnext < numTiles) loadTile(nextMeta, nextRegs);
        }
        probe(18);                                              // (compaction)
        const bool alive = (fl & 1u) != 0u;

        if (!(hotNow() & kHotLast)) {                                 // S8: compaction into `out`, binned by class
            // The compaction is the tile's latency chain (barrier, reservation round trip, barrier, stores): its waves issue
            // ahead of the ones that are tracing, so the chain is not stretched by instruction arbitration.  Measured
            // (profiles/r02_priority_experiments.txt): a launch on its own 5-6 % shorter, the pipelined rate +0.5 % (Cornell)
            // / +1.6 % (glass, depth 16) / +-0 (64 spheres); any level above 0 does; the reserving wave alone, or the window
            // opened at the next tile's loads already, gain nothing.
            __builtin_amdgcn_s_setprio(3);
            const int wave = (int)(tid >> 6), lane = (int)(tid & 63u);
            uint32_t *wv = s_wave + wvSel;                       // this tile's half of the double-buffered counts
            // same-class mask of this lane from four bit ballots (the sign compares already are the ballots)
            const unsigned long long ba = __ballot(alive);
            // class bits 0-2: the wall the ray can still hit (scenes with walls) or the octant of its direction
            uint32_t cls;
            if (hotWalls(hotNow()) > 0u) {
                cls = wallSel & 7u;
            } else {
                cls = (dir.x < 0.0f ? 1u : 0u) | (dir.y < 0.0f ? 2u : 0u) | (dir.z < 0.0f ? 4u : 0u);
            }
            cls |= smallCandI << 3;                             // (bit 3; mesh scenes: bits 3 and 4)
            // the lanes of this lane's class: per class bit k the ballot of the bit, taken as it is where the lane's own bit is set and
            // complemented where it is clear (bit - 1 = 0 or ~0) -- 32-bit halves, three instructions per bit and half
            uint32_t sameLo = (uint32_t)ba, sameHi = (uint32_t)(ba >> 32);
#pragma unroll
            for (int k = 0; k < CLSBITS; ++k) {
                const uint32_t bit = (cls >> k) & 1u;
                const unsigned long long bk = __ballot(bit != 0u);
                const uint32_t flip = bit - 1u;
                sameLo &= (uint32_t)bk ^ flip;
                sameHi &= (uint32_t)(bk >> 32) ^ flip;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(sameHi, __builtin_amdgcn_mbcnt_lo(sameLo, 0u));
            if (alive && rank == 0u) wv[wave * CLS + cls] = (uint32_t)(__popc(sameLo) + __popc(sameHi));   // the class's first lane
            probe(21);                                          // (first barrier)
            __syncthreads();
            probe(22);                                          // (reservation)
            uint32_t tk = 0u;
            if (ticketed && tid == (uint32_t)CLS) {             // (the lane behind the reserving ones, wave 0: in flight together with the reservation's atomics)
                const ArgsPtr A = launder(kargs);
                tk = atomicAdd(&A->ctrl->ticket[A->parity][A->depth][blockIdx.x % ticketShards()][0], 1u);
            }
            if (tid < CLS) {
                const ArgsPtr A = launder(kargs);
                Ctrl *const ctrl = A->ctrl;
                const uint32_t poolChunks = (uint32_t)A->prm.poolChunks;
                const int parity = A->parity, dnext = A->depth + 1;
                uint32_t total = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) total += wv[w * CLS + tid];
                const uint32_t oseg = tid * SUB + (blockIdx.x % SUB);
                uint32_t r0, sp, r1;
                reserveRun(&ctrl->pos[parity][dnext][oseg][0], &ctrl->bump[parity][dnext][0], A->out.list + (size_t)oseg * poolChunks, oseg,
                           poolChunks, (uint32_t)A->prm.chunkShift, A->genOut, total, &ctrl->error, s_base[3 * CLS + tid],
                           s_base[4 * CLS + tid], r0, sp, r1);
                s_base[tid] = r0;
                s_base[CLS + tid] = sp;
                s_base[2 * CLS + tid] = r1;
            }
            if (ticketed && tid == (uint32_t)CLS) s_ticket[0] = 2u * gridDim.x + tk * ticketShards() + blockIdx.x % ticketShards();
            probe(23);                                          // (second barrier)
            __syncthreads();
            if (kTickets) Tn1 = ticketed ? s_ticket[0] : Tnext + gridDim.x;
            PT_EXP_EXTRA_BARRIER()                                // (experiment builds only)
            probe(19);                                          // (stores)
            if (alive) {
                const ArgsPtr A = launder(kargs);
                // earlier waves' survivors of this class (kWaves = 4: three conditional terms, no loop)
                uint32_t waveOff = 0u;
#pragma unroll
                for (int w = 0; w < kWaves - 1; ++w) {
                    const uint32_t wq = wv[w * CLS + cls];
                    waveOff += wave > w ? wq : 0u;
                }
                const uint32_t r = waveOff + rank, sp = s_base[CLS + cls];
                const uint32_t slot = r < sp ? s_base[cls] + r : s_base[2 * CLS + cls] + (r - sp);
                char *const dst = reinterpret_cast<char *>(A->out.base);
                const size_t ocap = (size_t)A->out.cap;
                *reinterpret_cast<float4 *>(dst + 16 * (size_t)slot) = make_float4(org.x, org.y, org.z, dir.x);
                *reinterpret_cast<float4 *>(dst + 16 * ocap + 16 * (size_t)slot) = make_float4(dir.y, dir.z, col.x, col.y);
                PathC c;
                c.cz = col.z; c.pixHash = pixHash; c.pk = FIRST ? ((uint32_t)pix | ((uint32_t)itb << (uint32_t)launder(kargs)->prm.pixBits)) : pkCur;
                *reinterpret_cast<PathC *>(dst + 32 * ocap + 12 * (size_t)slot) = c;
            }
            // No third barrier: the counts are double-buffered.  The other half was last read in the previous tile, and every
            // wave finished those reads before it arrived at THIS tile's first barrier, so each wave may now clear its own
            // row of it for the next tile (its own next writes follow in program order; the other waves' next reads of that
            // row come after the next tile's first barrier).  s_base is rewritten only after the next tile's first barrier.
            wvSel ^= (uint32_t)(kWaves * CLS);
            if (lane < CLS) s_wave[wvSel + wave * CLS + lane] = 0u;
            __builtin_amdgcn_s_setprio(0);
        } else if (kTickets) {
            Tn1 = Tnext + gridDim.x;
        }
        probe(20);                                              // (tile done)
        T = Tnext;
    }
    censusLeave();
    probe(29);                           // (timeline builds: the launch's sums go out)
    // tallies: lanes -> wave (shuffles) -> workgroup (LDS) -> ONE atomic per workgroup and tally on counters sharded kTallyShards ways
    // (every workgroup of a launch ends with these: unsharded, or one per wave, they serialise at the memory side)
    const uint32_t waveLight = sLight, waveEarly = sEarly, waveMiss = sMiss + sEarly;
    __syncthreads();                                   // (every wave is done with the scratch of its last tile)
    {   // (the lane id laundered: the compiler had kept the prologue's `threadIdx.x * 4` LDS address alive for this one store and parked it
        // in scratch across the whole tile loop -- the last 8 B of scratch of the non-PLAIN later-bounce kernel)
        uint32_t tidE = threadIdx.x;
        asm volatile("" : "+v"(tidE));
        if (tidE < 3u) s_wave[tidE] = 0u;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        if (waveLight) atomicAdd(&s_wave[0], waveLight);
        if (waveMiss) atomicAdd(&s_wave[1], waveMiss);
        if (waveEarly) atomicAdd(&s_wave[2], waveEarly);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const ArgsPtr A = launder(kargs);
        Ctrl *const ctrl = A->ctrl;
        const int shard = blockIdx.x % kTallyShards;
        const uint32_t wgLight = s_wave[0], wgEarly = s_wave[2];
        unsigned long long wgMiss = s_wave[1];
        // (the camera rays of the pixels outside the tiles' index space, KParams::firstSkipped: misses whatever their jitter)
        if (FIRST && blockIdx.x == 0) wgMiss += (unsigned long long)A->prm.firstSkipped * (unsigned long long)A->batch;
        if (wgLight) atomicAdd(&ctrl->light_hits[shard][0], (unsigned long long)wgLight);
        if (wgMiss) atomicAdd(&ctrl->misses[shard][0], wgMiss);
        if (wgEarly) atomicAdd(&ctrl->early[A->depth + 1][shard][0], (unsigned long long)wgEarly);
    }
}

#undef S_GEOMHIT
#undef S_SPH

// ---- commit one iteration's radiance: image[pix] += contrib[pix]; the pixel's mask bit cleared -------------------
// Runs on the caller's stream, one launch per iteration in iteration order, so every pixel receives its
// samples in exactly the order a sequential renderer adds them (fp32 addition is not associative).
// Skipping an all-zero contribution equals adding +0 (the accumulator is never -0).
// `compactRows`: the accumulator holds only this shard's rows (PT_FLAG_ACCUM_SHARD_ROWS), pixel j of the shard
// at image[3j]; otherwise it is the full frame indexed by the global pixel index.
// `batch` iterations were traced together; their radiance buffers are consumed in iteration order.
// [b0, b1) = the iterations of the batch this launch consumes (the whole batch: 0, batch); the others stay parked -- their
// radiance entries and mask bits untouched -- for a later launch (PT_FLAG_TRACE_AHEAD: pt_iterate commits ONE iteration of a
// batch that was traced ahead).  `discard`: consume without adding (a traced-ahead batch the caller did not come back for).
// `snap` (round 6; device groups, pt_group.h): a full frame that receives the value of EVERY pixel of this shard after the commit --
// the snapshot a frame reduce reads while the next iteration is committed -- in the same pass: the accumulator is read once, for the
// addition and for the copy (a separate 11 MB device-to-device copy per iteration cost config C3 as written ~7 us each).
__global__ __launch_bounds__(kBlock) void k_commit(KParams prm, float *image, float *contrib, uint32_t *hitMask, int batch, int compactRows,
                                                   int b0, int b1, int discard, float *snap) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= prm.nLocal) return;
    const int lr = j / prm.W;
    const int x = j - lr * prm.W;
    const size_t gpix = (size_t)x + (size_t)(lr * prm.shardCount + prm.shardRank) * prm.W;
    const size_t pix = prm.contribLocal ? (size_t)j : gpix;                      // index in the radiance buffers / masks
    const size_t frame = prm.contribLocal ? (size_t)prm.nLocal : (size_t)prm.W * prm.H;
    float *px = image + 3 * (compactRows ? (size_t)j : gpix);
    // the pixel's mask words first (independent loads), then its entries sixteen at a time -- their loads are in flight
    // together, the additions stay in iteration order.  (A rank of 8 traces 256 iterations per batch: ~75 entries per lit pixel,
    // one memory round trip each if taken one by one.)
    constexpr int kWordsMax = (PT_MAX_BATCH + 31) / 32;
    uint32_t mw[kWordsMax], keep[kWordsMax];
    uint32_t any = 0u;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (snap) { ax = px[0]; ay = px[1]; az = px[2]; }            // (with a snapshot every pixel is read: issued beside the mask words)
#pragma unroll
    for (int w = 0; w < kWordsMax; ++w) {
        // bits of word w inside [b0, b1)
        const int lo = b0 - 32 * w, hi = b1 - 32 * w;
        const uint32_t below = lo <= 0 ? 0u : (lo >= 32 ? 0xffffffffu : (1u << lo) - 1u);      // bits < lo
        const uint32_t upto = hi <= 0 ? 0u : (hi >= 32 ? 0xffffffffu : (1u << hi) - 1u);        // bits < hi
        const uint32_t range = upto & ~below;
        const uint32_t word = (w * 32 < batch && range != 0u) ? hitMask[(size_t)w * frame + pix] : 0u;
        mw[w] = word & range;
        keep[w] = word & ~range;
        any |= mw[w];
    }
    float *const sp = snap ? snap + 3 * gpix : nullptr;
    if (any == 0u) {
        if (sp) { sp[0] = ax; sp[1] = ay; sp[2] = az; }
        return;
    }
    if (!snap) { ax = px[0]; ay = px[1]; az = px[2]; }
#pragma unroll
    for (int w = 0; w < kWordsMax; ++w) {
        uint32_t m = mw[w];
        if (m == 0u) continue;
        hitMask[(size_t)w * frame + pix] = keep[w];
        float *const base = contrib + 3 * ((size_t)(w * 32) * frame + pix);
        while (m) {                                            // ascending bits = iteration order
            int b[16];
            float v[16][3];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                b[q] = -1;
                if (m) {
                    b[q] = __builtin_ctz(m);
                    m &= m - 1u;
                    const float *c = base + 3 * (size_t)b[q] * frame;
                    v[q][0] = c[0]; v[q][1] = c[1]; v[q][2] = c[2];
                }
            }
            // (the entries are NOT re-zeroed: the pixel's mask bits say which of them hold something -- a bounce launch writes an entry with
            // plain stores, once, and sets its bit -- so a consumed entry is dead until the bit is set again; rounds 1-5 wrote 12 bytes of
            // zeros per consumed entry, 141 MB per batch of 64 on C2)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (b[q] >= 0 && !discard) { ax += v[q][0]; ay += v[q][1]; az += v[q][2]; }
        }
    }
    if (!discard) { px[0] = ax; px[1] = ay; px[2] = az; }
    if (sp) { sp[0] = ax; sp[1] = ay; sp[2] = az; }
}

// ... the same for ONE iteration of the batch (b1 == b0 + 1: every pt_iterate of a PT_FLAG_TRACE_AHEAD renderer, i.e. the reference's own
// protocol, and config C3 as written): one mask word, one bit, one entry per pixel.  The general kernel above carries sixteen entries and
// eight mask words per lane -- 100 VGPRs, four waves per SIMD -- through this case too, and a per-iteration commit costs the tracing
// kernels next to it its whole duration (they are bound by vector issue: profiles/r06_group_experiments.txt).  Same loads, same single
// addition per channel, same stores: the accumulator is bit-identical.
__global__ __launch_bounds__(kBlock) void k_commit_one(KParams prm, float *image, float *contrib, uint32_t *hitMask, int compactRows, int b, int discard,
                                                       float *snap) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= prm.nLocal) return;
    const int lr = j / prm.W;
    const int x = j - lr * prm.W;
    const size_t gpix = (size_t)x + (size_t)(lr * prm.shardCount + prm.shardRank) * prm.W;
    const size_t pix = prm.contribLocal ? (size_t)j : gpix;
    const size_t frame = prm.contribLocal ? (size_t)prm.nLocal : (size_t)prm.W * prm.H;
    float *px = image + 3 * (compactRows ? (size_t)j : gpix);
    uint32_t *const mp = hitMask + (size_t)(b >> 5) * frame + pix;
    const uint32_t bit = 1u << (b & 31);
    const uint32_t word = *mp;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (snap) { ax = px[0]; ay = px[1]; az = px[2]; }
    float *const sp = snap ? snap + 3 * gpix : nullptr;
    if (!(word & bit)) {
        if (sp) { sp[0] = ax; sp[1] = ay; sp[2] = az; }
        return;
    }
    float *const c = contrib + 3 * ((size_t)b * frame + pix);
    const float cx = c[0], cy = c[1], cz = c[2];
    if (!snap) { ax = px[0]; ay = px[1]; az = px[2]; }
    *mp = word & ~bit;                                           // (the entry itself is dead until its bit is set again: no re-zeroing)
    if (!discard) {
        ax += cx; ay += cy; az += cz;
        px[0] = ax; px[1] = ay; px[2] = az;
    }
    if (sp) { sp[0] = ax; sp[1] = ay; sp[2] = az; }
}

// ---- sendImageToPBO (reference src/pathtrace.cu:48-68) ---------------------------------------------
__global__ __launch_bounds__(kBlock) void k_to_rgba8(const float *image, int npix, int iter, uchar4 *pbo) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    const float *p = image + 3 * (size_t)i;
    int r = (int)(p[0] / iter * 255.0);
    int g = (int)(p[1] / iter * 255.0);
    int b = (int)(p[2] / iter * 255.0);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);   // glm::clamp = min(max(x, lo), hi), func_common.inl:451-456
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    uchar4 o;
    o.w = 0; o.x = (unsigned char)r; o.y = (unsigned char)g; o.z = (unsigned char)b;
    pbo[i] = o;
}

}  // namespace ptk
