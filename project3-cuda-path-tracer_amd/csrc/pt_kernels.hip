// pt_kernels.hip -- MI355X (gfx950 / CDNA4) path-tracing hot path + its C ABI (include/pt_amd.h).
//
// One iteration = camera-ray generation, then `traceDepth` launches of ONE fused persistent kernel
// per bounce: nearest-hit over the scene (geometry through the scalar path, materials in LDS) -> shade/scatter
// -> park emitter radiance ->
// stream compaction of the survivors straight into the next bounce's SoA buffers (wave64
// ballot/mbcnt ranks, LDS wave totals = workgroup-level exclusive scan; the workgroup's output range
// is reserved with ONE atomic on one of 8 sharded segment counters).  No host round trip inside an
// iteration: live counts stay on the device.  The multi-workgroup ORDERED scan (two-level decoupled
// look-back) is the stream-compaction library at the end of this file (pt_scan_exclusive_i32 /
// pt_compact_nonzero_i32).
//
// Replaces the unsolved pipeline of reference src/pathtrace.cu:133-167 (spec: SURVEY.md 3.4 S0-S9).
// HBM layout, kernels, rooflines: DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_device.h"

using namespace ptd;

static_assert(sizeof(PtGeom) == 236 && sizeof(PtMaterial) == 44 && sizeof(PtCamera) == 52,
              "layout must equal reference src/sceneStructs.h:18-47");

namespace {

constexpr int kBlock = 256;          // threads per workgroup = paths per tile (4 wave64)
constexpr int kWaves = kBlock / 64;
constexpr int kNumArrays = 11;       // SoA PathSegment: origin3, dir3, throughput3, pixelIndex, remainingBounces
constexpr int kMaxDepthSlots = PT_MAX_DEPTH + 2;

// ---- device control block ------------------------------------------------------------------------
constexpr int kOct = 8;              // direction octants: paths are binned by the signs of their new direction
constexpr int kSub = 4;              // append-counter shards per octant (workgroup blockIdx % kSub)
constexpr int kSeg = kOct * kSub;    // path buffers are split into kSeg segments with one append counter each
constexpr int kCtrPad = 32;          // one counter per 128-byte line: same-line atomics serialise at the memory side

struct Ctrl {
    // seg_count[p][d][s][0] = paths in segment s entering bounce d of an iteration with parity p.
    // The last bounce launch of an iteration zeroes the OTHER parity, i.e. re-arms the next iteration,
    // so an iteration needs neither a memset nor a separate re-arm launch.
    uint32_t seg_count[2][kMaxDepthSlots][kSeg][kCtrPad];
    // never zeroed by an iteration
    uint32_t error;                    // sticky device fault (scan-library look-back timeout)
    uint32_t pad[kCtrPad - 1];
    unsigned long long sum_live[kMaxDepthSlots];
    unsigned long long light_hits[kOct][kCtrPad / 2], misses[kOct][kCtrPad / 2];
};

// Camera constants derived once on the host (spec S2)
struct KParams {
    float view[3], up[3], right[3], pos[3];
    float pixLenX, pixLenY, halfW, halfH;
    int   W, H;
    int   shardRank, shardCount;
    int   nLocal;       // pixels rendered by this shard
    int   ngeoms, nmats;
    int   traceDepth;
    int   segCap;       // capacity (paths) of one segment of a path buffer, multiple of kBlock
};

// SoA PathSegment buffer: 11 arrays of `cap` = kSeg * segCap 4-byte elements, array k at base + k*cap
// (0-2 origin, 3-5 direction, 6-8 throughput, 9 pixelIndex, 10 remainingBounces); inside every array
// segment s owns [s*segCap, (s+1)*segCap) and is filled from its start.
struct PathSoA {
    float *base;
    int    cap;
    __host__ __device__ __forceinline__ float *a(int k) const { return base + (size_t)k * cap; }
    __host__ __device__ __forceinline__ int *pix() const { return reinterpret_cast<int *>(base + (size_t)9 * cap); }
    __host__ __device__ __forceinline__ int *rem() const { return reinterpret_cast<int *>(base + (size_t)10 * cap); }
};

// ---- cross-workgroup ordered prefix: two-level look-back ----------------------------------------------
// Tiles are handed out by an atomic ticket, so every predecessor of a tile is owned by a workgroup that
// is already running: no residency or dispatch-order assumption.  Each tile publishes
//   * its aggregate as ONE 8-byte agent-scope granule  desc[tile] = {status = 1 (hi), value (lo)}
//   * and adds it to its 64-tile group's word          grp[tile/64] += {value (hi), 1 (lo)}   (count in the
//     low half so that the wrapping sum can never carry into it).
// The exclusive prefix of tile t = sum of the full groups before it (one probe per 64 groups = 4096 tiles)
// + sum of the aggregates of its own group's earlier tiles (one probe).  Both probes are issued together,
// so the dependent latency is ~one memory round trip instead of the (tiles in flight)/64 serial probes of a
// flat decoupled look-back.  The value IS the flag in both words, so no fence is needed (the payload
// travels inside the granule).  Spins are bounded; a timeout sets the sticky error word.
constexpr int kSpinLimit = 1 << 22;
constexpr int kGroup = 64;

__device__ __forceinline__ unsigned long long word_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Called by all 64 lanes of wave 0 of the workgroup that owns `tile`.
__device__ __forceinline__ uint32_t lookback_exclusive(unsigned long long *desc, unsigned long long *grp, int tile,
                                                       uint32_t block_total, uint32_t *error_word) {
    const int lane = threadIdx.x & 63;
    const int g = tile / kGroup, r = tile - g * kGroup;
    if (lane == 0) {
        __hip_atomic_store(&desc[tile], (1ull << 32) | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&grp[g], ((unsigned long long)block_total << 32) | 1ull, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t excl = 0;
    // own group: aggregates of tiles 64g .. tile-1; first window of previous groups probed in the same trip
    {
        const int jg = g - 1 - lane;
        unsigned long long dt = 1ull << 32, dg = (unsigned long long)kGroup;   // "ready, value 0"
        int spins = 0;
        for (;;) {
            if (lane < r) dt = word_load(&desc[g * kGroup + lane]);
            if (jg >= 0) dg = word_load(&grp[jg]);
            if (__all((uint32_t)(dt >> 32) != 0u && (uint32_t)dg == (uint32_t)kGroup)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicExch(error_word, 1u);
                return 0u;
            }
        }
        excl = wave_sum((lane < r ? (uint32_t)dt : 0u) + (jg >= 0 ? (uint32_t)(dg >> 32) : 0u));
    }
    // more than 64 previous groups (> 4096 tiles ahead of this one)
    for (int base = g - 1 - 64; base >= 0; base -= 64) {
        const int jg = base - lane;
        unsigned long long dg = (unsigned long long)kGroup;
        int spins = 0;
        for (;;) {
            if (jg >= 0) dg = word_load(&grp[jg]);
            if (__all((uint32_t)dg == (uint32_t)kGroup)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicExch(error_word, 1u);
                return 0u;
            }
        }
        excl += wave_sum(jg >= 0 ? (uint32_t)(dg >> 32) : 0u);
    }
    return excl;
}

// Workgroup-level stable compaction rank of a 0/1 flag: ballot + mbcnt inside each wave, wave
// totals through LDS, cross-tile base from the look-back.  Returns the destination slot of this
// thread (valid when flag) and the tile's inclusive end in *tile_end (valid in every thread).
__device__ __forceinline__ uint32_t compact_slot(bool flag, int tile, unsigned long long *desc, unsigned long long *grp,
                                                 uint32_t *s_wave, uint32_t *s_excl, uint32_t *error_word,
                                                 uint32_t *tile_end) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long ballot = __ballot(flag);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
    if ((threadIdx.x & 63) == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        uint32_t c = s_wave[w];
        wave_off += w < wave ? c : 0u;
        total += c;
    }
    if (wave == 0) {
        uint32_t e = lookback_exclusive(desc, grp, tile, total, error_word);
        if (threadIdx.x == 0) *s_excl = e;
    }
    __syncthreads();
    const uint32_t excl = *s_excl;
    *tile_end = excl + total;
    return excl + wave_off + rank;
}

// ---- camera ray of the j-th pixel of this shard (spec S2) ------------------------------------------
__device__ __forceinline__ void cameraRay(const KParams &prm, int iter, int j, int &pix, F3 &org, F3 &dir) {
    const int lr = j / prm.W;
    const int x = j - lr * prm.W;
    const int y = lr * prm.shardCount + prm.shardRank;
    pix = x + y * prm.W;
    Rng rng = makeSeededRandomEngine(iter, pix, 0);
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
}

// camera rays alone, for pt_debug_trace_paths(bounces = 0)
__global__ __launch_bounds__(kBlock) void k_debug_camera_rays(KParams prm, int iter, float *o3, float *d3, int *pixOut) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= prm.nLocal) return;
    int pix;
    F3 org, dir;
    cameraRay(prm, iter, j, pix, org, dir);
    o3[3 * j] = org.x; o3[3 * j + 1] = org.y; o3[3 * j + 2] = org.z;
    d3[3 * j] = dir.x; d3[3 * j + 1] = dir.y; d3[3 * j + 2] = dir.z;
    pixOut[j] = pix;
}

// ---- one bounce: intersect + shade + accumulate + compact (spec S3-S8) -----------------------------
// Persistent workgroups walk the 256-path tiles of the bounce's queue (the kSeg input segments laid
// end to end), blockIdx-strided.  Survivors are BINNED BY DIRECTION OCTANT while they are compacted:
//   segment   = octant(new direction) * kSub + blockIdx % kSub,
//   rank      = exclusive scan of the lane's octant flag inside the wave (ballot + mbcnt) plus the earlier
//               waves' totals through LDS = workgroup-level exclusive scan per octant,
//   base      = ONE atomicAdd per non-empty octant of the tile on that segment's counter (8 lanes, one
//               instruction).
// A tile of the next bounce therefore holds rays of a single direction octant, which turns the exact
// early-miss of the box test (pt_device.h) into a wave-uniform branch for axis-aligned boxes.  Queue order
// never influences results: RNG and accumulator are keyed on the pixel index.
// No workgroup ever waits for another one, so there is no residency / dispatch-order assumption.
// A segment receives survivors of the tiles of the workgroups with one value of blockIdx % kSub only,
// i.e. at most ceil(tiles / kSub) * 256 <= segCap paths (see pt_init).
//
// FIRST = true is bounce 1 fused with camera-ray generation (spec S2): tile T holds the paths
// j = 256 T + lane of this shard's pixel list and the ray is built in registers, so the first bounce
// reads no path state at all.
template <bool FIRST>
__global__ __launch_bounds__(kBlock, 5) void k_bounce(KParams prm, int iter, int batch, int depth, int lastBounce, int parity,
                                                   PathSoA in, PathSoA out, Ctrl *ctrl,
                                                   const GeomDev *__restrict__ ggeoms,
                                                   const MaterialDev *__restrict__ gmats, float *contrib) {
    // LDS: the material table and the geom -> material map (indexed per lane by the nearest hit), and the
    // compaction scratch.  Geometry itself is wave-uniform in the nearest-hit loop, so it is fetched through the
    // scalar path (s_load into SGPRs, used directly as VALU operands): measured against an LDS-staged copy
    // read back with ds_read_b128 broadcasts this is 5 % faster on Cornell (7 geoms) and 11 % on the 70-geom
    // scene, and it frees ~40 VGPRs.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MaterialDev *smats = reinterpret_cast<MaterialDev *>(smem);
    int *s_geomMat = reinterpret_cast<int *>(smem + sizeof(MaterialDev) * prm.nmats);
    uint32_t *s_misc = reinterpret_cast<uint32_t *>(smem + sizeof(MaterialDev) * prm.nmats + sizeof(int) * ((prm.ngeoms + 3) & ~3));
    uint32_t *s_wave = s_misc;                       // [kWaves][kOct] alive count per wave and octant
    uint32_t *s_base = s_wave + kWaves * kOct;       // [kOct]   first output slot of this tile per octant
    uint32_t *s_segcnt = s_base + kOct;              // [kSeg]   paths per input segment
    uint32_t *s_segpre = s_segcnt + kSeg;            // [kSeg+2] tile prefix per input segment, [kSeg+1] = live paths

    if (lastBounce) {   // re-arm the next iteration: nobody touches the other parity's counters now
        uint32_t *other = &ctrl->seg_count[parity ^ 1][0][0][0];
        const int nwords = (prm.traceDepth + 2) * kSeg * kCtrPad;
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < nwords; i += gridDim.x * kBlock) other[i] = 0u;
    }
    // input queue: segment s holds s_cnt[s] paths = tiles [s_pre[s], s_pre[s+1]) of the global tile index
    uint32_t nLive, numTiles;
    if (FIRST) {
        nLive = (uint32_t)prm.nLocal * (uint32_t)batch;     // `batch` consecutive iterations share one wavefront
        numTiles = (nLive + kBlock - 1) / kBlock;
    } else {
        if (threadIdx.x < 64) {          // wave 0: exclusive scan of the kSeg tile counts
            const uint32_t c = threadIdx.x < kSeg ? ctrl->seg_count[parity][depth][threadIdx.x][0] : 0u;
            const uint32_t t = (c + kBlock - 1) / kBlock;
            uint32_t inc = t, sum = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(inc, o, 64), us = __shfl_up(sum, o, 64);
                if ((int)threadIdx.x >= o) { inc += up; sum += us; }
            }
            if (threadIdx.x < kSeg) { s_segcnt[threadIdx.x] = c; s_segpre[threadIdx.x + 1] = inc; }
            if (threadIdx.x == 0) s_segpre[0] = 0;
            if (threadIdx.x == 63) s_segpre[kSeg + 1] = sum;   // total live paths
        }
        __syncthreads();
        numTiles = s_segpre[kSeg];
        nLive = s_segpre[kSeg + 1];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctrl->sum_live[depth], (unsigned long long)nLive);
    if (blockIdx.x >= numTiles) return;

    // stage the materials in LDS once per (persistent) workgroup, 16 B per lane per step
    {
        const float4 *msrc = reinterpret_cast<const float4 *>(gmats);
        float4 *mdst = reinterpret_cast<float4 *>(smats);
        const int m16 = prm.nmats * (int)(sizeof(MaterialDev) / 16);
        for (int i = threadIdx.x; i < m16; i += kBlock) mdst[i] = msrc[i];
        for (int i = threadIdx.x; i < prm.ngeoms; i += kBlock) s_geomMat[i] = ggeoms[i].material;
    }
    __syncthreads();

    uint32_t waveLight = 0, waveMiss = 0;   // wave-uniform tallies, flushed once at the end
    uint32_t sgIn = 0;                      // input segment of the current tile (tiles are visited in increasing order)
    for (uint32_t T = blockIdx.x; T < numTiles; T += gridDim.x) {
        bool valid;
        uint32_t idx = 0;
        if (FIRST) {
            idx = T * kBlock + threadIdx.x;                     // position in this shard's pixel list
            valid = idx < nLive;
        } else {
            // global tile -> (segment, local tile)
            while (T >= s_segpre[sgIn + 1]) ++sgIn;
            const uint32_t local = (T - s_segpre[sgIn]) * kBlock + threadIdx.x;
            valid = local < s_segcnt[sgIn];
            idx = sgIn * (uint32_t)prm.segCap + local;
        }

        bool alive = false;
        bool lightHit = false, missed = false;
        F3 org = f3(0, 0, 0), dir = f3(0, 0, 1), col = f3(0, 0, 0);
        int pix = 0, rem = 0;
        int itb = 0;                                            // which iteration of the batch this path belongs to
        if (valid) {
            if (FIRST) {
                itb = (int)(idx / (uint32_t)prm.nLocal);
                cameraRay(prm, iter + itb, (int)(idx - (uint32_t)itb * (uint32_t)prm.nLocal), pix, org, dir);
                col = f3(1.0f, 1.0f, 1.0f);
                rem = prm.traceDepth;
            } else {
                org = f3(in.a(0)[idx], in.a(1)[idx], in.a(2)[idx]);
                dir = f3(in.a(3)[idx], in.a(4)[idx], in.a(5)[idx]);
                col = f3(in.a(6)[idx], in.a(7)[idx], in.a(8)[idx]);
                pix = in.pix()[idx];
                const int packed = in.rem()[idx];               // remainingBounces | batch index << 8
                rem = packed & 0xff;
                itb = packed >> 8;
            }

            // nearest hit, geoms in file order, strict '<' so the first geom wins ties (S3)
            float tbest = 0.0f;
            int hit = -1;
            F3 P = f3(0, 0, 0), N = f3(0, 0, 0);
            bool outside = false;
            const float dd = dot(dir, dir);
            for (int g = 0; g < prm.ngeoms; ++g) {
                const GeomDev &G = ggeoms[g];
                const int type = G.type;
                F3 p, n;
                bool o = false;
                float t = -1.0f;
                if (type == 0) {
                    if (!sphereCertainMiss(G, org, dir, dd)) t = sphereIntersectionTest<FIRST>(G, org, dir, p, n, o);
                } else {
                    t = boxIntersectionTest<true, FIRST>(G, org, dir, p, n, o);
                }
                if (t > 0.0f && (hit < 0 || t < tbest)) {
                    tbest = t; hit = g; P = p; N = n; outside = o;
                }
            }
            if (hit < 0) {
                missed = true;                                   // S4: background is black
            } else {
                const MaterialDev &M = smats[s_geomMat[hit]];
                const F3 mcol = f3(M.color[0], M.color[1], M.color[2]);
                if (M.emittance > 0.0f) {                        // S5: emitter ends the path
                    lightHit = true;
                    if (contrib) {
                        // Deferred accumulation: iterations overlap on several streams, so the radiance
                        // is parked in this iteration's own buffer (one path per pixel: race-free, no
                        // read) and k_commit adds it to the accumulator in iteration order.
                        const F3 c = (col * mcol) * M.emittance;
                        float *px = contrib + 3 * ((size_t)itb * ((size_t)prm.W * prm.H) + (size_t)pix);
                        px[0] = c.x; px[1] = c.y; px[2] = c.z;
                    }
                } else if (!lastBounce) {                        // S6 scatter (S7: skipped on the last bounce)
                    Rng rng = makeSeededRandomEngine(iter + itb, pix, depth);
                    const F3 scol = f3(M.specColor[0], M.specColor[1], M.specColor[2]);
                    F3 ndir, norg;
                    if (M.hasRefractive > 0.0f) {
                        const float ior = M.ior;
                        const float eta = outside ? 1.0f / ior : ior;
                        const float c = dot(N, dir);
                        const float k = 1.0f - eta * eta * (1.0f - c * c);
                        const float u = u01(rng);
                        bool doReflect = true;
                        if (k >= 0.0f) {
                            float r0 = (1.0f - ior) / (1.0f + ior);
                            r0 = r0 * r0;
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
                    } else if (M.hasReflective > 0.0f) {
                        const float u = u01(rng);
                        if (u < 0.5f) {
                            ndir = reflect(dir, N);
                            col = col * scol;
                        } else {
                            ndir = calculateRandomDirectionInHemisphere(N, rng);
                            col = col * mcol;
                        }
                        norg = P + N * 0.001f;
                    } else {
                        ndir = calculateRandomDirectionInHemisphere(N, rng);
                        col = col * mcol;
                        norg = P + N * 0.001f;
                    }
                    org = norg;
                    dir = ndir;
                    alive = true;
                }
            }
        }
        waveLight += (uint32_t)__popcll(__ballot(lightHit));
        waveMiss += (uint32_t)__popcll(__ballot(missed));

        if (!lastBounce) {                                       // S8: compaction into `out`, binned by octant
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const int oct = (dir.x < 0.0f ? 1 : 0) | (dir.y < 0.0f ? 2 : 0) | (dir.z < 0.0f ? 4 : 0);
            uint32_t rank = 0, myCount = 0;
#pragma unroll
            for (int k = 0; k < kOct; ++k) {
                const unsigned long long m = __ballot(alive && oct == k);
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (oct == k) rank = r;
                if (lane == k) myCount = (uint32_t)__popcll(m);
            }
            if (lane < kOct) s_wave[wave * kOct + lane] = myCount;
            __syncthreads();
            if (threadIdx.x < kOct) {
                uint32_t total = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) total += s_wave[w * kOct + threadIdx.x];
                const uint32_t oseg = threadIdx.x * kSub + (blockIdx.x % kSub);
                s_base[threadIdx.x] = total ? atomicAdd(&ctrl->seg_count[parity][depth + 1][oseg][0], total) : 0u;
            }
            __syncthreads();
            if (alive) {
                uint32_t waveOff = 0;
                for (int w = 0; w < wave; ++w) waveOff += s_wave[w * kOct + oct];
                const uint32_t oseg = (uint32_t)oct * kSub + (blockIdx.x % kSub);
                const uint32_t slot = oseg * (uint32_t)prm.segCap + s_base[oct] + waveOff + rank;
                out.a(0)[slot] = org.x; out.a(1)[slot] = org.y; out.a(2)[slot] = org.z;
                out.a(3)[slot] = dir.x; out.a(4)[slot] = dir.y; out.a(5)[slot] = dir.z;
                out.a(6)[slot] = col.x; out.a(7)[slot] = col.y; out.a(8)[slot] = col.z;
                out.pix()[slot] = pix;
                out.rem()[slot] = (rem - 1) | (itb << 8);
            }
            __syncthreads();   // s_wave / s_base are rewritten by the next tile
        }
    }
    if ((threadIdx.x & 63) == 0) {
        const int shard = blockIdx.x % kOct;
        if (waveLight) atomicAdd(&ctrl->light_hits[shard][0], (unsigned long long)waveLight);
        if (waveMiss) atomicAdd(&ctrl->misses[shard][0], (unsigned long long)waveMiss);
    }
}

// ---- commit one iteration's radiance: image[pix] += contrib[pix]; contrib[pix] = 0 -------------------
// Runs on the caller's stream, one launch per iteration in iteration order, so every pixel receives its
// samples in exactly the order a sequential renderer adds them (fp32 addition is not associative).
// Skipping an all-zero contribution equals adding +0 (the accumulator is never -0).
// `compactRows`: the accumulator holds only this shard's rows (PT_FLAG_ACCUM_SHARD_ROWS), pixel j of the shard
// at image[3j]; otherwise it is the full frame indexed by the global pixel index.
// `batch` iterations were traced together; their radiance buffers are consumed in iteration order.
__global__ __launch_bounds__(kBlock) void k_commit(KParams prm, float *image, float *contrib, int batch, int compactRows) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= prm.nLocal) return;
    const int lr = j / prm.W;
    const int x = j - lr * prm.W;
    const size_t pix = (size_t)x + (size_t)(lr * prm.shardCount + prm.shardRank) * prm.W;
    const size_t frame = (size_t)prm.W * prm.H;
    float *px = image + 3 * (compactRows ? (size_t)j : pix);
    float ax = px[0], ay = px[1], az = px[2];
    bool dirty = false;
    for (int b = 0; b < batch; ++b) {
        float *c = contrib + 3 * ((size_t)b * frame + pix);
        const float cx = c[0], cy = c[1], cz = c[2];
        if (cx != 0.0f || cy != 0.0f || cz != 0.0f) {
            ax += cx; ay += cy; az += cz;
            c[0] = 0.0f; c[1] = 0.0f; c[2] = 0.0f;
            dirty = true;
        }
    }
    if (dirty) { px[0] = ax; px[1] = ay; px[2] = az; }
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

// ---- stream-compaction library kernels ---------------------------------------------------------------
struct ScanCtrl {
    uint32_t ticket;
    uint32_t error;
};
constexpr int kScanItems = 4;                 // int4 per thread
constexpr int kScanTile = kBlock * kScanItems;

__global__ __launch_bounds__(kBlock) void k_scan_exclusive(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                           long long n, ScanCtrl *sc, unsigned long long *desc,
                                                           unsigned long long *grp) {
    __shared__ uint32_t s_wave[kWaves];
    __shared__ uint32_t s_excl, s_tile;
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(&sc->ticket, 1u);
        __syncthreads();
        const long long tile = s_tile;
        if (tile >= numTiles) break;
        const long long base = tile * kScanTile + (long long)threadIdx.x * kScanItems;
        int32_t v[kScanItems];
        if (base + kScanItems <= n && ((reinterpret_cast<uintptr_t>(in + base) & 15) == 0)) {
            const int4 q = *reinterpret_cast<const int4 *>(in + base);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : 0;
        }
        const uint32_t tsum = (uint32_t)v[0] + (uint32_t)v[1] + (uint32_t)v[2] + (uint32_t)v[3];
        // wave-level inclusive scan of the per-thread sums
        uint32_t inc = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            uint32_t c = s_wave[w];
            wave_off += w < wave ? c : 0u;
            total += c;
        }
        if (wave == 0) {
            uint32_t e = lookback_exclusive(desc, grp, (int)tile, total, &sc->error);
            if (threadIdx.x == 0) s_excl = e;
        }
        __syncthreads();
        uint32_t run = s_excl + wave_off + (inc - tsum);
        int32_t o4[kScanItems];
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            o4[k] = (int32_t)run;
            run += (uint32_t)v[k];
        }
        if (base + kScanItems <= n && ((reinterpret_cast<uintptr_t>(out + base) & 15) == 0)) {
            *reinterpret_cast<int4 *>(out + base) = make_int4(o4[0], o4[1], o4[2], o4[3]);
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k)
                if (base + k < n) out[base + k] = o4[k];
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_compact_nonzero(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                            long long n, ScanCtrl *sc, unsigned long long *desc,
                                                            unsigned long long *grp, long long *count_out) {
    __shared__ uint32_t s_wave[kWaves];
    __shared__ uint32_t s_excl, s_tile;
    const long long numTiles = (n + kBlock - 1) / kBlock;
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) *count_out = 0;
    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(&sc->ticket, 1u);
        __syncthreads();
        const long long tile = s_tile;
        if (tile >= numTiles) break;
        const long long i = tile * kBlock + threadIdx.x;
        const int32_t v = i < n ? in[i] : 0;
        uint32_t tileEnd;
        const uint32_t slot = compact_slot(v != 0, (int)tile, desc, grp, s_wave, &s_excl, &sc->error, &tileEnd);
        if (v != 0) out[slot] = v;
        if (tile == numTiles - 1 && threadIdx.x == 0) *count_out = (long long)tileEnd;
    }
}

// ---- primitive test kernels (device functions exactly as the render kernels use them) -------------------
__global__ void k_test_utilhash(const uint32_t *in, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = utilhash(in[i]);
}
__global__ void k_test_rng(const uint32_t *seeds, int nseeds, int ndraws, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseeds) return;
    Rng r = seedEngine(seeds[i]);
    for (int k = 0; k < ndraws; ++k) out[(size_t)i * ndraws + k] = u01(r);
}
__global__ void k_test_intersect(const GeomDev *geoms, const int *gidx, const float *rays, int n, float *t, float *p3,
                                 float *n3, int *outside) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GeomDev G = geoms[gidx[i]];
    F3 ro = f3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    F3 rd = f3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    F3 P = f3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]);
    F3 N = f3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
    bool o = outside[i] != 0;
    // odd lanes take the early-miss variant so both instantiations are checked against the golden vectors
    // the certain-miss shortcut must agree with the full test on every golden vector
    const bool cull = G.type == 0 && sphereCertainMiss(G, ro, rd, dot(rd, rd));
    if (cull && sphereIntersectionTest(G, ro, rd, P, N, o) != -1.0f) { t[i] = __builtin_nanf(""); return; }
    t[i] = G.type == 0 ? (cull ? -1.0f : sphereIntersectionTest(G, ro, rd, P, N, o))
         : ((i & 1) ? boxIntersectionTest<true>(G, ro, rd, P, N, o) : boxIntersectionTest<false>(G, ro, rd, P, N, o));
    p3[3 * i] = P.x; p3[3 * i + 1] = P.y; p3[3 * i + 2] = P.z;
    n3[3 * i] = N.x; n3[3 * i + 1] = N.y; n3[3 * i + 2] = N.z;
    outside[i] = o ? 1 : 0;
}
// sphereCertainMiss soundness sweep: pseudo-random rays (origins up to ~60 units away, aimed near the sphere
// so that grazing cases are dense) against every sphere of `geoms`; counts culled rays and VIOLATIONS
// (culled although the full test returns a hit).
__global__ void k_sweep_sphere_cull(const GeomDev *geoms, int ngeoms, unsigned long long seed, int per_thread,
                                    unsigned long long *culled, unsigned long long *violations) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nv = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[8];
        for (int j = 0; j < 8; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const GeomDev G = geoms[(blockIdx.x + k) % ngeoms];
        if (G.type != 0) continue;
        const F3 c = f3(G.centre[0], G.centre[1], G.centre[2]);
        const float dist = __builtin_exp2f(u[0] * 12.0f - 6.0f);                 // 1/64 .. 64 units
        const F3 od = normalize(f3(u[1] - 0.5f, u[2] - 0.5f, u[3] - 0.5f));
        const F3 org = c + od * dist;
        // aim at a point within ~1.3 bounding radii of the centre: hits, grazes and near misses
        const float R = __builtin_sqrtf(G.cullR2 * 4.0f) * 0.5f;
        const F3 tgt = c + f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f) * (2.6f * R);
        F3 dir = normalize(tgt - org);
        if (u[7] < 0.1f) dir = -dir;
        if (sphereCertainMiss(G, org, dir, dot(dir, dir))) {
            ++nc;
            F3 P, N;
            bool o;
            if (sphereIntersectionTest(G, org, dir, P, N, o) != -1.0f) ++nv;
        }
    }
    if (nc) atomicAdd(culled, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// slabQuotients vs the compiler's correctly rounded division; counts mismatching lanes
__global__ void k_test_slab_quotients(const float *o, const float *d, int n, float *t1, float *t2, float *r1, float *r2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    slabQuotients(o[i], d[i], t1[i], t2[i]);
    r1[i] = (-0.5f - o[i]) / d[i];
    r2[i] = (+0.5f - o[i]) / d[i];
}
// pseudo-random sweep entirely on the device: returns the number of bit mismatches (NaN == NaN)
__global__ void k_sweep_slab_quotients(unsigned long long seed, int per_thread, unsigned long long *mismatches) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int bad = 0;
    for (int k = 0; k < per_thread; ++k) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;                     // xorshift64
        const uint32_t ob = (uint32_t)x, db = (uint32_t)(x >> 32);
        float o, d;
        if ((k & 3) == 0) {            // raw bit patterns: every exponent, denormals, inf, NaN
            o = __uint_as_float(ob); d = __uint_as_float(db);
        } else {                       // the range the tracer lives in: |o| < ~4000, |d| <= 1
            o = ((int)(ob >> 8) - (1 << 23)) * (1.0f / 2048.0f) * ((k & 4) ? 1.0f : 1e-3f);
            d = __uint_as_float((db & 0x807fffffu) | ((uint32_t)(127 - (db >> 23 & 31)) << 23));
            if ((k & 15) == 5) o = (ob & 1) ? 0.5f : -0.5f;           // numerator exactly +0
        }
        float t1, t2;
        slabQuotients(o, d, t1, t2);
        const float r1 = (-0.5f - o) / d, r2 = (+0.5f - o) / d;
        const bool e1 = __float_as_uint(t1) == __float_as_uint(r1) || (t1 != t1 && r1 != r1);
        const bool e2 = __float_as_uint(t2) == __float_as_uint(r2) || (t2 != t2 && r2 != r2);
        bad += (e1 ? 0u : 1u) + (e2 ? 0u : 1u);
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
__global__ void k_test_hemisphere(const float *nrm, const int *iid, int n, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r = makeSeededRandomEngine(iid[3 * i], iid[3 * i + 1], iid[3 * i + 2]);
    F3 d = calculateRandomDirectionInHemisphere(f3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), r);
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
}
__global__ void k_test_sincos(const float *x, int n, float *s, float *c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincosPoly(x[i], s[i], c[i]);
}
__global__ void k_test_reflect_refract(const float *I, const float *N, const float *eta, int n, float *rl, float *rr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 a = f3(I[3 * i], I[3 * i + 1], I[3 * i + 2]), b = f3(N[3 * i], N[3 * i + 1], N[3 * i + 2]);
    F3 r1 = reflect(a, b), r2 = refract(a, b, eta[i]);
    rl[3 * i] = r1.x; rl[3 * i + 1] = r1.y; rl[3 * i + 2] = r1.z;
    rr[3 * i] = r2.x; rr[3 * i + 1] = r2.y; rr[3 * i + 2] = r2.z;
}

// =====================================================================================================
// host side
// =====================================================================================================
std::string g_err = "";

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
            return fail(PT_ERR_HIP, "HIP error (%s:%d): %s: %s", "pt_kernels.hip", __LINE__, #expr, \
                        hipGetErrorString(e_));                                                     \
    } while (0)

constexpr int kMaxSlots = 4;

// One in-flight iteration: its own stream, path buffers, counters and deferred-radiance buffer.
struct Slot {
    hipStream_t stream = nullptr;
    float *pathbuf[2] = {nullptr, nullptr};
    Ctrl *ctrl = nullptr;
    float *contrib = nullptr;      // maxBatch x W*H*3, zero between batches
    hipEvent_t evDone = nullptr;       // all bounce launches of the slot's current iteration finished
    hipEvent_t evCommitted = nullptr;  // k_commit consumed (and re-zeroed) `contrib`
    int parity = 0;                // which half of Ctrl::seg_count the slot's next iteration uses
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
    MaterialDev *dmats = nullptr;
    int numTilesMax = 0;    // upper bound of tiles in one bounce queue (incl. one partial tile per segment)
    int segCap = 0;         // paths per segment; a path buffer holds kSeg * segCap paths per array
    int grid = 0;
    size_t ldsBytes = 0;
    long long iterations = 0;
    long long seq = 0;      // iterations enqueued since pt_init: slot = seq % nslots
    // kernel timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evBounce;
    double msBounce = 0;
    long long nBounce = 0;
} S;

int count_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

PathSoA soa(float *base, int cap) {
    PathSoA s;
    s.base = base;
    s.cap = cap > 0 ? cap : 1;
    return s;
}

void pack_geom(const PtGeom &g, GeomDev &d, const float *eye = nullptr) {
    memset(&d, 0, sizeof d);
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) {
            d.inv[c * 3 + r] = g.inverseTransform[c * 4 + r];
            d.xf[c * 3 + r] = g.transform[c * 4 + r];
            d.invT[c * 3 + r] = g.invTranspose[c * 4 + r];
        }
    d.type = g.type;
    d.material = g.materialid;
    // sphere culling data (sphereCertainMiss): bounds smax >= sigma_max, smin <= sigma_min of the 3x3 part
    double A[3][3], Ai[3][3];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) { A[c][r] = g.transform[c * 4 + r]; Ai[c][r] = g.inverseTransform[c * 4 + r]; }
    double len[3], fro = 0, froi = 0;
    bool orth = true;
    for (int c = 0; c < 3; ++c) {
        len[c] = std::sqrt(A[c][0] * A[c][0] + A[c][1] * A[c][1] + A[c][2] * A[c][2]);
        for (int r = 0; r < 3; ++r) { fro += A[c][r] * A[c][r]; froi += Ai[c][r] * Ai[c][r]; }
    }
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b) {
            const double dp = A[a][0] * A[b][0] + A[a][1] * A[b][1] + A[a][2] * A[b][2];
            if (!(std::fabs(dp) <= 1e-5 * len[a] * len[b])) orth = false;
        }
    double smax, smin;
    if (orth) {            // rotation x scale: the singular values are the column lengths
        smax = std::max(len[0], std::max(len[1], len[2])) * (1 + 1e-5);
        smin = std::min(len[0], std::min(len[1], len[2])) * (1 - 1e-5);
    } else {               // any matrix: Frobenius bounds
        smax = std::sqrt(fro);
        smin = froi > 0 ? 1.0 / std::sqrt(froi) : 0.0;
    }
    d.centre[0] = g.transform[12]; d.centre[1] = g.transform[13]; d.centre[2] = g.transform[14];
    const double r2 = 0.25 * smax * smax * (1 + 1e-3), kk = smin > 0 ? 1e-4 * (smax / smin) * (smax / smin) : INFINITY;
    const bool ok = std::isfinite(r2) && std::isfinite(kk) && kk < 0.5 && smin > 0;
    d.cullR2 = ok ? (float)r2 : INFINITY;      // infinite radius: never culled
    d.cullK = ok ? (float)kk : 0.0f;
    if (eye) {   // ptd::mulMV(inv, eye, 1) in the same operation order (this file is built with -ffp-contract=off)
        const float *m = d.inv;
        for (int r = 0; r < 3; ++r) {
            const float a0 = m[0 + r] * eye[0], a1 = m[3 + r] * eye[1], a2 = m[6 + r] * eye[2], a3 = m[9 + r] * 1.0f;
            const float s01 = a0 + a1, s23 = a2 + a3;
            d.camObj[r] = s01 + s23;
        }
    }
}
void pack_material(const PtMaterial &m, MaterialDev &d) {
    memset(&d, 0, sizeof d);
    d.color[0] = m.color.x; d.color[1] = m.color.y; d.color[2] = m.color.z;
    d.specColor[0] = m.specularColor.x; d.specColor[1] = m.specularColor.y; d.specColor[2] = m.specularColor.z;
    d.hasReflective = m.hasReflective;
    d.hasRefractive = m.hasRefractive;
    d.ior = m.indexOfRefraction;
    d.emittance = m.emittance;
}

// host mirrors of the glm ops used for the camera basis (same op order as ptd::)
struct H3 { float x, y, z; };
H3 hcross(H3 x, H3 y) { return H3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
H3 hnormalize(H3 a) {
    float d = a.x * a.x + a.y * a.y + a.z * a.z;
    float s = 1.0f / std::sqrt(d);
    return H3{a.x * s, a.y * s, a.z * s};
}

int resolve_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &ms, long long &n) {
    for (auto &pr : v) {
        float t = 0;
        HIPCHECK(hipEventSynchronize(pr.second));
        HIPCHECK(hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        n += 1;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    v.clear();
    return PT_OK;
}

int launch_bounce(Slot &sl, int iter, int batch, int depth, bool lastBounce, float *contrib) {
    const PathSoA in = soa(sl.pathbuf[(depth - 1) & 1], kSeg * S.segCap);
    const PathSoA out = soa(sl.pathbuf[depth & 1], kSeg * S.segCap);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (S.flags & PT_FLAG_KERNEL_TIMING) {
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventCreate(&e1));
        HIPCHECK(hipEventRecord(e0, sl.stream));
    }
    if (depth == 1)
        hipLaunchKernelGGL(k_bounce<true>, dim3(S.grid), dim3(kBlock), S.ldsBytes, sl.stream, S.prm, iter, batch, depth,
                           lastBounce ? 1 : 0, sl.parity, in, out, sl.ctrl, S.dgeoms, S.dmats, contrib);
    else
        hipLaunchKernelGGL(k_bounce<false>, dim3(S.grid), dim3(kBlock), S.ldsBytes, sl.stream, S.prm, iter, batch, depth,
                           lastBounce ? 1 : 0, sl.parity, in, out, sl.ctrl, S.dgeoms, S.dmats, contrib);
    if (e0) {
        HIPCHECK(hipEventRecord(e1, sl.stream));
        S.evBounce.emplace_back(e0, e1);
        if (S.evBounce.size() > 8192) {
            int rc = resolve_events(S.evBounce, S.msBounce, S.nBounce);
            if (rc) return rc;
        }
    }
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

// wait for every stream the renderer uses
int sync_all() {
    for (int i = 0; i < S.nslots; ++i) HIPCHECK(hipStreamSynchronize(S.slot[i].stream));
    HIPCHECK(hipStreamSynchronize(S.stream));
    return PT_OK;
}

int check_device_fault() {
    int rc = sync_all();
    if (rc) return rc;
    for (int i = 0; i < S.nslots; ++i) {
        uint32_t err = 0;
        HIPCHECK(hipMemcpy(&err, &S.slot[i].ctrl->error, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(PT_ERR_DEVICE, "device fault flag set");
    }
    return PT_OK;
}

// scan library workspace
struct ScanWs {
    ScanCtrl *ctrl = nullptr;
    unsigned long long *desc = nullptr;
    long long tiles = 0;
} W;

int scan_ws(long long tiles) {
    if (!W.ctrl) HIPCHECK(hipMalloc(&W.ctrl, sizeof(ScanCtrl)));
    if (tiles > W.tiles) {
        if (W.desc) HIPCHECK(hipFree(W.desc));
        W.desc = nullptr;
        long long cap = tiles < 1024 ? 1024 : tiles;
        HIPCHECK(hipMalloc(&W.desc, (size_t)(cap + (cap + kGroup - 1) / kGroup) * 8));
        W.tiles = cap;
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
    if (perCU > 8) perCU = 8;
    *grid = prop.multiProcessorCount * perCU;
    return PT_OK;
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
// C ABI
// =====================================================================================================
extern "C" {

const char *pt_last_error(void) { return g_err.c_str(); }
int pt_device_count(void) { return count_devices(); }

void pt_free(void) {
    // pathtraceFree before the first Init (src/main.cpp:91-94) must be a no-op
    if (!S.init && !S.image && !S.dgeoms && S.nslots == 0) return;
    for (int i = 0; i < kMaxSlots; ++i)
        if (S.slot[i].stream) (void)hipStreamSynchronize(S.slot[i].stream);
    (void)hipStreamSynchronize(S.stream);
    for (auto &pr : S.evBounce) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    S.evBounce.clear();
    for (int i = 0; i < kMaxSlots; ++i) {
        Slot &sl = S.slot[i];
        for (int k = 0; k < 2; ++k)
            if (sl.pathbuf[k]) (void)hipFree(sl.pathbuf[k]);
        if (sl.ctrl) (void)hipFree(sl.ctrl);
        if (sl.contrib) (void)hipFree(sl.contrib);
        if (sl.evDone) (void)hipEventDestroy(sl.evDone);
        if (sl.evCommitted) (void)hipEventDestroy(sl.evCommitted);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (S.ownImage && S.image) (void)hipFree(S.image);
    if (S.dgeoms) (void)hipFree(S.dgeoms);
    if (S.dmats) (void)hipFree(S.dmats);
    S = State();
}

int pt_init(const PtCamera *cam, const PtGeom *geoms, int ngeoms, const PtMaterial *mats, int nmats, int traceDepth,
            const PtOptions *opts) {
    if (!cam || ngeoms < 0 || nmats < 0 || (ngeoms && !geoms) || (nmats && !mats))
        return fail(PT_ERR_INVALID, "pt_init: null argument");
    if (cam->resolution[0] <= 0 || cam->resolution[1] <= 0) return fail(PT_ERR_INVALID, "pt_init: bad resolution");
    if (traceDepth < 1 || traceDepth > PT_MAX_DEPTH) return fail(PT_ERR_INVALID, "pt_init: traceDepth must be 1..%d", PT_MAX_DEPTH);
    if ((long long)cam->resolution[0] * cam->resolution[1] > (1ll << 30)) return fail(PT_ERR_INVALID, "pt_init: frame too large");
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_SPHERE && geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_init: geom %d has unknown type", i);
        if (geoms[i].materialid < 0 || geoms[i].materialid >= nmats) return fail(PT_ERR_INVALID, "pt_init: geom %d references material %d", i, geoms[i].materialid);
    }
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "pt_init: no HIP device (this library has no CPU fallback)");
    pt_free();

    PtOptions o;
    memset(&o, 0, sizeof o);
    o.shard_count = 1;
    o.device = -1;
    if (opts) o = *opts;
    if (o.shard_count < 1 || o.shard_rank < 0 || o.shard_rank >= o.shard_count) return fail(PT_ERR_INVALID, "pt_init: bad shard %d/%d", o.shard_rank, o.shard_count);
    if (o.pipeline_depth < 0 || o.pipeline_depth > kMaxSlots) return fail(PT_ERR_INVALID, "pt_init: pipeline_depth must be 0..%d", kMaxSlots);
    if (o.max_batch < 0 || o.max_batch > PT_MAX_BATCH) return fail(PT_ERR_INVALID, "pt_init: max_batch must be 0..%d", PT_MAX_BATCH);
    if (o.device >= 0) HIPCHECK(hipSetDevice(o.device));
    HIPCHECK(hipGetDevice(&S.device));
    S.stream = (hipStream_t)o.stream;
    S.flags = o.flags;
    S.cam = *cam;

    const int Wd = cam->resolution[0], H = cam->resolution[1];
    S.P = Wd * H;
    const int rows = H > o.shard_rank ? (H - o.shard_rank + o.shard_count - 1) / o.shard_count : 0;
    S.nLocal = rows * Wd;

    KParams &k = S.prm;
    memset(&k, 0, sizeof k);
    const H3 view{cam->view.x, cam->view.y, cam->view.z}, up{cam->up.x, cam->up.y, cam->up.z};
    const H3 right = hnormalize(hcross(view, up));
    k.view[0] = view.x; k.view[1] = view.y; k.view[2] = view.z;
    k.up[0] = up.x; k.up[1] = up.y; k.up[2] = up.z;
    k.right[0] = right.x; k.right[1] = right.y; k.right[2] = right.z;
    k.pos[0] = cam->position.x; k.pos[1] = cam->position.y; k.pos[2] = cam->position.z;
    const float kPI = 3.1415926535897932384626422832795028841971f;   // src/utilities.h:12
    const float ys = std::tan(cam->fov[1] * (kPI / 180));            // src/scene.cpp:133 convention
    const float xs = (ys * Wd) / H;
    k.pixLenX = (2.0f * xs) / (float)Wd;
    k.pixLenY = (2.0f * ys) / (float)H;
    k.halfW = (float)Wd * 0.5f;
    k.halfH = (float)H * 0.5f;
    k.W = Wd; k.H = H;
    k.shardRank = o.shard_rank; k.shardCount = o.shard_count;
    k.nLocal = S.nLocal;
    k.ngeoms = ngeoms; k.nmats = nmats;
    k.traceDepth = traceDepth;

    if (o.accum_dev) {
        S.image = o.accum_dev;
        S.ownImage = false;
    } else {
        const size_t n = (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) ? (size_t)(S.nLocal > 0 ? S.nLocal : 1) : (size_t)S.P;
        HIPCHECK(hipMalloc(&S.image, n * 3 * sizeof(float)));
        S.ownImage = true;
        HIPCHECK(hipMemsetAsync(S.image, 0, n * 3 * sizeof(float), S.stream));
    }
    // Path buffers: kSeg = kOct x kSub segments.  A segment receives survivors only from the workgroups with one
    // value of blockIdx % kSub; tiles are blockIdx-strided and the grid is a multiple of kSub, so those workgroups
    // process at most ceil(tiles / kSub) tiles, and tiles <= ceil(nLocal/256) + kSeg (one partial tile per input
    // segment).  Worst case (every ray in one octant) is provisioned: 8x the live paths, 325 MB per buffer at 720p.
    S.maxBatch = o.max_batch > 0 ? o.max_batch : 1;
    if ((long long)S.nLocal * S.maxBatch > (1ll << 30)) return fail(PT_ERR_INVALID, "pt_init: max_batch x pixels too large");
    S.numTilesMax = (int)(((long long)S.nLocal * S.maxBatch + kBlock - 1) / kBlock) + kSeg;
    S.segCap = ((S.numTilesMax + kSub - 1) / kSub) * kBlock;
    k.segCap = S.segCap;
    const size_t cap = (size_t)kSeg * S.segCap;
    // Iterations are independent (RNG keyed on pixel/iteration/depth), so up to `nslots` of them are in flight
    // on their own streams; the small late-bounce launches of one overlap the big early launches of the next.
    S.nslots = o.pipeline_depth > 0 ? o.pipeline_depth : 3;
    for (int i = 0; i < S.nslots; ++i) {
        Slot &sl = S.slot[i];
        HIPCHECK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) HIPCHECK(hipMalloc(&sl.pathbuf[b], cap * kNumArrays * sizeof(float)));
        HIPCHECK(hipMalloc(&sl.ctrl, sizeof(Ctrl)));
        HIPCHECK(hipMemset(sl.ctrl, 0, sizeof(Ctrl)));
        HIPCHECK(hipMalloc(&sl.contrib, (size_t)S.maxBatch * S.P * 3 * sizeof(float)));
        HIPCHECK(hipMemset(sl.contrib, 0, (size_t)S.maxBatch * S.P * 3 * sizeof(float)));
        HIPCHECK(hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&sl.evCommitted, hipEventDisableTiming));
    }

    std::vector<GeomDev> hg(ngeoms ? ngeoms : 1);
    std::vector<MaterialDev> hm(nmats ? nmats : 1);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i], k.pos);
    for (int i = 0; i < nmats; ++i) pack_material(mats[i], hm[i]);
    HIPCHECK(hipMalloc(&S.dgeoms, hg.size() * sizeof(GeomDev)));
    HIPCHECK(hipMalloc(&S.dmats, hm.size() * sizeof(MaterialDev)));
    HIPCHECK(hipMemcpy(S.dgeoms, hg.data(), hg.size() * sizeof(GeomDev), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(S.dmats, hm.data(), hm.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));

    S.ldsBytes = sizeof(MaterialDev) * nmats + sizeof(int) * ((ngeoms + 3) & ~3) +
                 (kWaves * kOct + kOct + kSeg + kSeg + 2 + 2) * sizeof(uint32_t);
    if (S.ldsBytes > 160 * 1024) return fail(PT_ERR_INVALID, "pt_init: scene does not fit the 160 KiB LDS (%zu B)", S.ldsBytes);
    if (S.ldsBytes > 64 * 1024) {
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bounce<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.ldsBytes));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bounce<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.ldsBytes));
    }
    int rc = persistent_grid(reinterpret_cast<const void *>(k_bounce<false>), S.ldsBytes, &S.grid);
    if (rc) return rc;
    if (S.grid > S.numTilesMax) S.grid = S.numTilesMax;
    S.grid = (S.grid / kSub) * kSub;      // T % kSub == blockIdx % kSub for every tile T of a workgroup
    if (S.grid < kSub) S.grid = kSub;
    HIPCHECK(hipDeviceSynchronize());
    S.init = true;
    g_err.clear();
    return PT_OK;
}

int pt_iterate_batch(int frame, int first_iter, int count, void *rgba8_dev) {
    (void)frame;  // always 0 in the reference (src/main.cpp:102)
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_iterate before pt_init");
    if (count < 1 || count > S.maxBatch) return fail(PT_ERR_INVALID, "pt_iterate_batch: count must be 1..max_batch (%d)", S.maxBatch);
    if (first_iter < 1 || first_iter + count - 1 >= (1 << 22))
        return fail(PT_ERR_INVALID, "pt_iterate: iter must be 1..4194303 (seed bits, pathtrace.cu:43)");
    Slot &sl = S.slot[S.seq % S.nslots];
    // the slot's radiance buffers must have been consumed by the commit of its previous batch
    HIPCHECK(hipStreamWaitEvent(sl.stream, sl.evCommitted, 0));
    const int D = S.prm.traceDepth;
    for (int d = 1; d <= D; ++d) {
        int rc = launch_bounce(sl, first_iter, count, d, d == D, sl.contrib);
        if (rc) return rc;
    }
    sl.parity ^= 1;   // the last launch re-armed the other half of the slot's counters
    HIPCHECK(hipEventRecord(sl.evDone, sl.stream));
    // commit on the caller's stream: commits are therefore ordered like the pt_iterate calls
    HIPCHECK(hipStreamWaitEvent(S.stream, sl.evDone, 0));
    if (S.nLocal > 0) {
        hipLaunchKernelGGL(k_commit, dim3((S.nLocal + kBlock - 1) / kBlock), dim3(kBlock), 0, S.stream, S.prm, S.image, sl.contrib,
                           count, (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) ? 1 : 0);
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipEventRecord(sl.evCommitted, S.stream));
    if (rgba8_dev) {
        if (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) return fail(PT_ERR_INVALID, "pt_iterate: no PBO conversion from a row-sharded accumulator");
        hipLaunchKernelGGL(k_to_rgba8, dim3((S.P + kBlock - 1) / kBlock), dim3(kBlock), 0, S.stream, S.image, S.P,
                           first_iter + count - 1, reinterpret_cast<uchar4 *>(rgba8_dev));
        HIPCHECK(hipGetLastError());
    }
    S.seq += 1;
    S.iterations += count;
    return PT_OK;
}

int pt_iterate(int frame, int iter, void *rgba8_dev) { return pt_iterate_batch(frame, iter, 1, rgba8_dev); }

int pt_sync(void) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_sync before pt_init");
    return check_device_fault();
}

int pt_readback(float *rgb_sum_host) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_readback before pt_init");
    if (!rgb_sum_host) return fail(PT_ERR_INVALID, "pt_readback: null");
    // every commit so far is already ordered before this copy on the caller's stream
    if (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) {   // scatter this shard's rows into a zeroed full frame
        std::vector<float> rows((size_t)S.nLocal * 3);
        if (S.nLocal) HIPCHECK(hipMemcpyAsync(rows.data(), S.image, rows.size() * sizeof(float), hipMemcpyDeviceToHost, S.stream));
        HIPCHECK(hipStreamSynchronize(S.stream));
        memset(rgb_sum_host, 0, (size_t)S.P * 3 * sizeof(float));
        const size_t rowFloats = (size_t)S.prm.W * 3;
        for (int lr = 0; lr * S.prm.W < S.nLocal; ++lr)
            memcpy(rgb_sum_host + (size_t)(lr * S.prm.shardCount + S.prm.shardRank) * rowFloats, rows.data() + lr * rowFloats,
                   rowFloats * sizeof(float));
        return PT_OK;
    }
    HIPCHECK(hipMemcpyAsync(rgb_sum_host, S.image, (size_t)S.P * 3 * sizeof(float), hipMemcpyDeviceToHost, S.stream));
    HIPCHECK(hipStreamSynchronize(S.stream));
    return PT_OK;
}

int pt_readback_rgba8(int iter, uint8_t *rgba_host) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_readback_rgba8 before pt_init");
    if (!rgba_host || iter < 1) return fail(PT_ERR_INVALID, "pt_readback_rgba8: bad argument");
    if (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) return fail(PT_ERR_INVALID, "pt_readback_rgba8: accumulator is row-sharded");
    DevBuf<uchar4> tmp;
    int rc = tmp.alloc(S.P);
    if (rc) return rc;
    hipLaunchKernelGGL(k_to_rgba8, dim3((S.P + kBlock - 1) / kBlock), dim3(kBlock), 0, S.stream, S.image, S.P, iter, tmp.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(rgba_host, tmp.p, (size_t)S.P * 4, hipMemcpyDeviceToHost, S.stream));
    HIPCHECK(hipStreamSynchronize(S.stream));
    return PT_OK;
}

int pt_counters(PtCounters *out) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_counters before pt_init");
    if (!out) return fail(PT_ERR_INVALID, "pt_counters: null");
    int rc = sync_all();
    if (rc) return rc;
    rc = resolve_events(S.evBounce, S.msBounce, S.nBounce);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    static Ctrl h;   // 0.5 MB: keep it off the stack
    bool fault = false;
    for (int i = 0; i < S.nslots; ++i) {
        HIPCHECK(hipMemcpy(&h, S.slot[i].ctrl, sizeof h, hipMemcpyDeviceToHost));
        for (int d = 0; d < kMaxDepthSlots; ++d) out->live[d] += (int64_t)h.sum_live[d];
        for (int sg = 0; sg < kOct; ++sg) {
            out->light_hits += (int64_t)h.light_hits[sg][0];
            out->misses += (int64_t)h.misses[sg][0];
        }
        fault = fault || h.error != 0;
    }
    out->iterations = S.iterations;
    out->bounce_launches = S.nBounce;
    out->bounce_kernel_ms = S.msBounce;
    out->raygen_kernel_ms = 0.0;   // camera rays are generated inside the first bounce launch
    out->raygen_launches = 0;
    if (fault) return fail(PT_ERR_DEVICE, "device fault flag set");
    return PT_OK;
}

int pt_counters_reset(void) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_counters_reset before pt_init");
    int rc = sync_all();
    if (rc) return rc;
    rc = resolve_events(S.evBounce, S.msBounce, S.nBounce);
    if (rc) return rc;
    S.msBounce = 0;
    S.nBounce = 0;
    S.iterations = 0;
    for (int i = 0; i < S.nslots; ++i) HIPCHECK(hipMemset(S.slot[i].ctrl, 0, sizeof(Ctrl)));
    return PT_OK;
}

int pt_debug_trace_paths(int iter, int bounces, float *origin3, float *dir3, float *color3, int32_t *pixelIndex,
                         int32_t *count) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_debug_trace_paths before pt_init");
    if (bounces < 0 || bounces > PT_MAX_DEPTH || !count) return fail(PT_ERR_INVALID, "pt_debug_trace_paths: bad argument");
    if (bounces > S.prm.traceDepth) return fail(PT_ERR_INVALID, "pt_debug_trace_paths: bounces > traceDepth");
    int rc = sync_all();
    if (rc) return rc;
    Slot &sl = S.slot[0];
    if (bounces == 0) {   // camera rays only (they never exist in HBM: generation is fused into bounce 1)
        const int nl = S.nLocal;
        *count = nl;
        if (nl == 0) return PT_OK;
        DevBuf<float> o, d;
        DevBuf<int> px;
        if ((rc = o.alloc((size_t)nl * 3)) || (rc = d.alloc((size_t)nl * 3)) || (rc = px.alloc(nl))) return rc;
        hipLaunchKernelGGL(k_debug_camera_rays, dim3((nl + kBlock - 1) / kBlock), dim3(kBlock), 0, sl.stream, S.prm, iter,
                           o.p, d.p, px.p);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(sl.stream));
        if (origin3) HIPCHECK(hipMemcpy(origin3, o.p, (size_t)nl * 12, hipMemcpyDeviceToHost));
        if (dir3) HIPCHECK(hipMemcpy(dir3, d.p, (size_t)nl * 12, hipMemcpyDeviceToHost));
        if (pixelIndex) HIPCHECK(hipMemcpy(pixelIndex, px.p, (size_t)nl * 4, hipMemcpyDeviceToHost));
        if (color3)
            for (size_t i = 0; i < (size_t)nl * 3; ++i) color3[i] = 1.0f;
        return PT_OK;
    }
    HIPCHECK(hipMemsetAsync(&sl.ctrl->seg_count[0][0][0][0], 0, sizeof(sl.ctrl->seg_count), sl.stream));
    for (int d = 1; d <= bounces; ++d) {
        rc = launch_bounce(sl, iter, 1, d, false, nullptr);  // no radiance, survivors always written
        if (rc) return rc;
    }
    // gather the kSeg segments of the queue entering bounce `bounces + 1`, then sort by pixel index
    uint32_t segn[kSeg];
    for (int sg = 0; sg < kSeg; ++sg)
        HIPCHECK(hipMemcpyAsync(&segn[sg], &sl.ctrl->seg_count[sl.parity][bounces + 1][sg][0], 4, hipMemcpyDeviceToHost, sl.stream));
    HIPCHECK(hipMemsetAsync(&sl.ctrl->seg_count[0][0][0][0], 0, sizeof(sl.ctrl->seg_count), sl.stream));
    HIPCHECK(hipStreamSynchronize(sl.stream));
    size_t n = 0;
    for (int sg = 0; sg < kSeg; ++sg) n += segn[sg];
    *count = (int32_t)n;
    if (n == 0) return PT_OK;
    const PathSoA sb = soa(sl.pathbuf[bounces & 1], kSeg * S.segCap);
    std::vector<float> cols[kNumArrays];
    for (int k = 0; k < kNumArrays; ++k) {
        cols[k].resize(n);
        size_t off = 0;
        for (int sg = 0; sg < kSeg; ++sg) {
            if (segn[sg])
                HIPCHECK(hipMemcpy(cols[k].data() + off, sb.a(k) + (size_t)sg * S.segCap, (size_t)segn[sg] * 4, hipMemcpyDeviceToHost));
            off += segn[sg];
        }
    }
    const int *pixcol = reinterpret_cast<const int *>(cols[9].data());
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return pixcol[x] < pixcol[y]; });
    float *dst[3] = {origin3, dir3, color3};
    for (size_t i = 0; i < n; ++i) {
        const size_t src = order[i];
        for (int grp = 0; grp < 3; ++grp)
            if (dst[grp])
                for (int c = 0; c < 3; ++c) dst[grp][3 * i + c] = cols[grp * 3 + c][src];
        if (pixelIndex) pixelIndex[i] = pixcol[src];
    }
    return PT_OK;
}

// ---- stream compaction library -------------------------------------------------------------------------
int pt_scan_exclusive_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, void *stream) {
    if (n < 0 || (n > 0 && (!in_dev || !out_dev))) return fail(PT_ERR_INVALID, "pt_scan_exclusive_i32: bad argument");
    if (n == 0) return PT_OK;
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    const long long tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles > 0x7fffffffll) return fail(PT_ERR_INVALID, "pt_scan_exclusive_i32: n too large");
    int rc = scan_ws(tiles);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHECK(hipMemsetAsync(W.ctrl, 0, sizeof(ScanCtrl), st));
    const long long words = tiles + (tiles + kGroup - 1) / kGroup;
    HIPCHECK(hipMemsetAsync(W.desc, 0, (size_t)words * 8, st));
    int grid = 0;
    rc = persistent_grid(reinterpret_cast<const void *>(k_scan_exclusive), 0, &grid);
    if (rc) return rc;
    if (grid > tiles) grid = (int)tiles;
    hipLaunchKernelGGL(k_scan_exclusive, dim3(grid), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, W.ctrl, W.desc,
                       W.desc + tiles);
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

int pt_compact_nonzero_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, int64_t *count_dev, void *stream) {
    if (n < 0 || !count_dev || (n > 0 && (!in_dev || !out_dev))) return fail(PT_ERR_INVALID, "pt_compact_nonzero_i32: bad argument");
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    const long long tiles = (n + kBlock - 1) / kBlock;
    if (tiles > 0x7fffffffll) return fail(PT_ERR_INVALID, "pt_compact_nonzero_i32: n too large");
    int rc = scan_ws(tiles ? tiles : 1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHECK(hipMemsetAsync(W.ctrl, 0, sizeof(ScanCtrl), st));
    const long long words = tiles + (tiles + kGroup - 1) / kGroup;
    HIPCHECK(hipMemsetAsync(W.desc, 0, (size_t)(words ? words : 1) * 8, st));
    int grid = 0;
    rc = persistent_grid(reinterpret_cast<const void *>(k_compact_nonzero), 0, &grid);
    if (rc) return rc;
    if (grid > tiles) grid = (int)tiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_compact_nonzero, dim3(grid), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, W.ctrl, W.desc,
                       W.desc + tiles, reinterpret_cast<long long *>(count_dev));
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

// ---- primitive tests over host arrays ------------------------------------------------------------------
#define NEED_GPU() do { if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device"); } while (0)
#define UP(buf, host, count) do { int rc_ = buf.alloc(count); if (rc_) return rc_; \
    HIPCHECK(hipMemcpy(buf.p, host, (size_t)(count) * sizeof(*buf.p), hipMemcpyHostToDevice)); } while (0)
#define DOWN(host, buf, count) HIPCHECK(hipMemcpy(host, buf.p, (size_t)(count) * sizeof(*buf.p), hipMemcpyDeviceToHost))
#define GRID(n) dim3(((n) + 255) / 256), dim3(256), 0, 0

int pt_test_utilhash(const uint32_t *in, uint32_t *out, int n) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<uint32_t> a, b;
    UP(a, in, n);
    int rc = b.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_utilhash, GRID(n), a.p, b.p, n);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(out, b, n);
    return PT_OK;
}

int pt_test_rng(const uint32_t *seeds, int nseeds, int ndraws, float *u01_out) {
    NEED_GPU();
    if (nseeds <= 0 || ndraws <= 0) return PT_OK;
    DevBuf<uint32_t> a;
    DevBuf<float> b;
    UP(a, seeds, nseeds);
    int rc = b.alloc((size_t)nseeds * ndraws); if (rc) return rc;
    hipLaunchKernelGGL(k_test_rng, GRID(nseeds), a.p, nseeds, ndraws, b.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(u01_out, b, (size_t)nseeds * ndraws);
    return PT_OK;
}

int pt_test_intersect(const PtGeom *geoms, int ngeoms, const int32_t *geom_index, const float *rays, int n, float *t,
                      float *p3, float *n3, int32_t *outside) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    for (int i = 0; i < n; ++i)
        if (geom_index[i] < 0 || geom_index[i] >= ngeoms) return fail(PT_ERR_INVALID, "pt_test_intersect: geom index out of range");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i]);
    DevBuf<GeomDev> dg;
    DevBuf<int> di, dout;
    DevBuf<float> dr, dt, dp, dn;
    UP(dg, hg.data(), ngeoms);
    UP(di, geom_index, n);
    UP(dr, rays, (size_t)n * 6);
    UP(dp, p3, (size_t)n * 3);
    UP(dn, n3, (size_t)n * 3);
    UP(dout, outside, n);
    int rc = dt.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_intersect, GRID(n), dg.p, di.p, dr.p, n, dt.p, dp.p, dn.p, dout.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(t, dt, n);
    DOWN(p3, dp, (size_t)n * 3);
    DOWN(n3, dn, (size_t)n * 3);
    DOWN(outside, dout, n);
    return PT_OK;
}

int pt_test_sphere_cull_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled,
                              uint64_t *violations) {
    NEED_GPU();
    if (!geoms || ngeoms < 1 || !culled || !violations || rays < 0) return fail(PT_ERR_INVALID, "pt_test_sphere_cull_sweep: bad argument");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i]);
    DevBuf<GeomDev> dg;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    int rc = cnt.alloc(2);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 16));
    const int per_thread = 256, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_sphere_cull, dim3((unsigned)blocks), dim3(threads), 0, 0, dg.p, ngeoms, (unsigned long long)seed,
                       per_thread, cnt.p, cnt.p + 1);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 16, hipMemcpyDeviceToHost));
    *culled = h[0];
    *violations = h[1];
    return PT_OK;
}

int pt_test_slab_quotients(const float *o, const float *d, int n, float *t1, float *t2, float *ref1, float *ref2) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, b, q1, q2, r1, r2;
    UP(a, o, n);
    UP(b, d, n);
    int rc;
    if ((rc = q1.alloc(n)) || (rc = q2.alloc(n)) || (rc = r1.alloc(n)) || (rc = r2.alloc(n))) return rc;
    hipLaunchKernelGGL(k_test_slab_quotients, GRID(n), a.p, b.p, n, q1.p, q2.p, r1.p, r2.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(t1, q1, n);
    DOWN(t2, q2, n);
    DOWN(ref1, r1, n);
    DOWN(ref2, r2, n);
    return PT_OK;
}

int pt_test_slab_quotients_sweep(uint64_t seed, int64_t pairs, uint64_t *mismatches) {
    NEED_GPU();
    if (!mismatches || pairs < 0) return fail(PT_ERR_INVALID, "pt_test_slab_quotients_sweep: bad argument");
    DevBuf<unsigned long long> m;
    int rc = m.alloc(1);
    if (rc) return rc;
    HIPCHECK(hipMemset(m.p, 0, 8));
    const int per_thread = 1024, threads = 256;
    long long blocks = (pairs + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_slab_quotients, dim3((unsigned)blocks), dim3(threads), 0, 0, (unsigned long long)seed, per_thread, m.p);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    HIPCHECK(hipMemcpy(&h, m.p, 8, hipMemcpyDeviceToHost));
    *mismatches = h;
    return PT_OK;
}

int pt_test_hemisphere(const float *normals3, const int32_t *iid3, int n, float *out3) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, o;
    DevBuf<int> b;
    UP(a, normals3, (size_t)n * 3);
    UP(b, iid3, (size_t)n * 3);
    int rc = o.alloc((size_t)n * 3); if (rc) return rc;
    hipLaunchKernelGGL(k_test_hemisphere, GRID(n), a.p, b.p, n, o.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(out3, o, (size_t)n * 3);
    return PT_OK;
}

int pt_test_sincos(const float *x, int n, float *s, float *c) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, ds, dc;
    UP(a, x, n);
    int rc = ds.alloc(n); if (rc) return rc;
    rc = dc.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_sincos, GRID(n), a.p, n, ds.p, dc.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(s, ds, n);
    DOWN(c, dc, n);
    return PT_OK;
}

int pt_test_reflect_refract(const float *I3, const float *N3, const float *eta, int n, float *refl3, float *refr3) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, b, e, r1, r2;
    UP(a, I3, (size_t)n * 3);
    UP(b, N3, (size_t)n * 3);
    UP(e, eta, n);
    int rc = r1.alloc((size_t)n * 3); if (rc) return rc;
    rc = r2.alloc((size_t)n * 3); if (rc) return rc;
    hipLaunchKernelGGL(k_test_reflect_refract, GRID(n), a.p, b.p, e.p, n, r1.p, r2.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(refl3, r1, (size_t)n * 3);
    DOWN(refr3, r2, (size_t)n * 3);
    return PT_OK;
}

}  // extern "C"
