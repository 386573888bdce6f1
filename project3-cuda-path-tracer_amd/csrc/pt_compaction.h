// pt_compaction.h -- the stream-compaction library (the reference's empty stream_compaction/ stub, README.md:83-86):
// work-efficient multi-workgroup exclusive scan and stable compaction on gfx950.
// Included by pt_api.hip only.
//
// Reduce-then-scan over CHUNKS: the array is cut into at most 2048 chunks of whole 4096-element tiles;
//   1. k_scan_reduce    one workgroup per chunk: its sum (scan) or its count of non-zero elements (compaction)
//   2. k_scan_apply / k_compact_apply   one workgroup per chunk: adds up the totals of the chunks before its own (at most 2047 words), re-reads
//      the chunk and writes its part of the result from that prefix (chunks newest first: the re-read comes out of the memory-side cache where
//      it still holds them); the workgroup of the last chunk delivers the compaction's count
// i.e. 12 bytes of traffic per element for the scan (8 is the minimum) and 8 + 4 per kept element for the compaction,
// TWO launches on the caller's stream, and NO workgroup ever waits for another one: no tickets, no look-back, no spinning.
// (Rounds 3-5 ran a third launch between the two -- one workgroup scanning the chunk totals: ~5 us, a quarter of a call at 2^22 elements.)
// (Rounds 1-2 shipped a single-pass scan with decoupled look-back: one atomic ticket per tile on ONE address bounded it at
// 25 / 180 GB/s with 256- / 1024-element tiles and at 500 GB/s with 4096-element ones.)
#pragma once
#include "pt_common.h"

namespace ptk {

constexpr int kScanItems = 16;                     // per thread: four 16-byte loads
constexpr int kScanTile = kBlock * kScanItems;     // 4096 elements
constexpr int kScanChunksMax = 2048;               // <= kScanTile: the chunk totals are scanned as one tile

// loads the thread's 16 consecutive elements (zeros beyond n); `fast`: whole tile inside the array and 16-byte aligned
__device__ __forceinline__ void load_items(const int32_t *__restrict__ in, long long base, long long n, bool fast, int32_t (&v)[kScanItems]) {
    if (fast) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            const int4 x = *reinterpret_cast<const int4 *>(in + base + 4 * q);
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : 0;
    }
}
// the same tile for a SUM (any order will do): thread t takes the 16-byte words t, t + 256, t + 512, t + 768 of the tile -- a wave's load is one
// run of 1 KB instead of 64 pieces of 16 bytes 64 bytes apart (round 6)
__device__ __forceinline__ void load_items_striped(const int32_t *__restrict__ in, long long tileBase, long long n, bool fast, int32_t (&v)[kScanItems]) {
    if (fast) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            const int4 x = *reinterpret_cast<const int4 *>(in + tileBase + 4 * (q * kBlock + (long long)threadIdx.x));
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
        load_items(in, tileBase + (long long)threadIdx.x * kScanItems, n, false, v);
    }
}
// Inclusive scan of one value per lane inside the wave: six DPP steps, no LDS crossbar -- row_shr:1 / 2 / 4 / 8 inside each row of 16 lanes
// (a source lane outside the row reads as 0), then row_bcast:15 (lane 15 of a row into the next row: rows 1 and 3) and row_bcast:31
// (lane 31 into rows 2 and 3).  (Rounds 2-4 used __shfl_up: a ds_bpermute round trip, a compare and a select per step.)
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t x) {
#define PT_DPP_ADD(ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, false)
    PT_DPP_ADD(0x111, 0xf);     // row_shr:1
    PT_DPP_ADD(0x112, 0xf);     // row_shr:2
    PT_DPP_ADD(0x114, 0xf);     // row_shr:4
    PT_DPP_ADD(0x118, 0xf);     // row_shr:8
    PT_DPP_ADD(0x142, 0xa);     // row_bcast:15 -> rows 1, 3
    PT_DPP_ADD(0x143, 0xc);     // row_bcast:31 -> rows 2, 3
#undef PT_DPP_ADD
    return x;
}
// the wave's sum: lane 63 of its inclusive scan (a scalar)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive(v), 63); }
// the thread's exclusive offset inside the tile and the tile's total (one workgroup barrier; the caller puts another one
// before s_wave is written again)
__device__ __forceinline__ uint32_t tile_offsets(uint32_t mine, uint32_t *s_wave, uint32_t *tile_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive(mine);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t c = s_wave[w];
        wave_off += w < wave ? c : 0u;
        total += c;
    }
    *tile_total = total;
    return wave_off + (inc - mine);
}

// 1. chunk totals: COUNT = false: sum of the elements (mod 2^32); true: number of non-zero elements
// PIPE (chunks of four tiles or more): the next tile's loads are issued before the current tile is worked on -- +4 .. 5 % at 2^26 and 2^28
// elements, -4 % at 2^24, where a chunk is two tiles: the host picks (profiles/r06_scan_summary.txt)
template <bool COUNT, bool PIPE>
__global__ __launch_bounds__(kBlock) void k_scan_reduce(const int32_t *__restrict__ in, long long n, long long tilesPerChunk,
                                                        uint32_t *__restrict__ partial) {
    __shared__ uint32_t s_wave[kWaves];
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const long long t0 = (long long)blockIdx.x * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    uint32_t acc = 0;
    // (the NEXT tile's loads are issued before this tile is summed: two tiles' worth of loads in flight per workgroup -- round 6)
    int32_t v[kScanItems], nx[kScanItems];
    auto load = [&](long long tile, int32_t (&w)[kScanItems]) { load_items_striped(in, tile * kScanTile, n, aligned && (tile + 1) * kScanTile <= n, w); };
    if (PIPE && t0 < t1) load(t0, v);
    for (long long tile = t0; tile < t1; ++tile) {
        if (!PIPE) load(tile, v);
        if (PIPE && tile + 1 < t1) load(tile + 1, nx);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) acc += COUNT ? (v[k] != 0 ? 1u : 0u) : (uint32_t)v[k];
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k) v[k] = nx[k];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) total += s_wave[w];
        partial[blockIdx.x] = total;
    }
}

// 2a. the scan of a chunk, starting from its prefix
// `partial` holds the chunks' TOTALS as k_scan_reduce left them: the workgroup adds up the ones before its chunk itself (at most 2047 words, eight
// per thread) -- no launch in between that scans the totals (one workgroup, ~5 us of launch and round trips: round 6)
template <bool PIPE>
__global__ __launch_bounds__(kBlock) void k_scan_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, long long n,
                                                       long long tilesPerChunk, const uint32_t *__restrict__ partial) {
    __shared__ uint32_t s_wave[2 * kWaves];      // (the waves' totals of two consecutive tiles: ONE barrier per tile instead of two -- round 6)
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    // The chunks are taken NEWEST FIRST (workgroup 0 = the last chunk): k_scan_reduce has just streamed the whole array through the
    // 256 MB memory-side cache front to back, so its END is what the cache still holds -- a second front-to-back pass would evict
    // every line just before it needs it, the reverse pass reads most of the array from the cache instead of from HBM.
    const long long chunk = (long long)gridDim.x - 1 - blockIdx.x;
    const long long t0 = chunk * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    int32_t v[kScanItems], nx[kScanItems];
    if (PIPE && t0 < t1) load_items(in, t0 * kScanTile + (long long)threadIdx.x * kScanItems, n, aligned && (t0 + 1) * kScanTile <= n, v);
    // (the totals before the chunk are loaded behind the first tile's elements and added up inside the first tile's own barrier: every
    // workgroup of the launch is resident at once, so a prologue of its own is paid in full -- measured: +4 us at 2^26 elements.  Eight
    // INDEPENDENT loads: a loop of `chunk / 256` trips waits for each load before it issues the next one, +7 us)
    __shared__ uint32_t s_pre[kWaves];
    uint32_t carry = 0, pacc = 0;
    {
        uint32_t pv[kScanChunksMax / kBlock];
#pragma unroll
        for (int k = 0; k < kScanChunksMax / kBlock; ++k) {
            const long long i = (long long)k * kBlock + threadIdx.x;
            pv[k] = i < chunk ? partial[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kScanChunksMax / kBlock; ++k) pacc += pv[k];
    }
    // (the NEXT tile's loads are issued before this tile's scan -- its barriers and its stores -- so that a workgroup's memory round trips
    // overlap its own work, not only its neighbours': round 6)
    for (long long tile = t0; tile < t1; ++tile) {
        const long long base = tile * kScanTile + (long long)threadIdx.x * kScanItems;
        const bool fast = aligned && (tile + 1) * kScanTile <= n;
        if (!PIPE) load_items(in, base, n, fast, v);
        if (PIPE && tile + 1 < t1) load_items(in, (tile + 1) * kScanTile + (long long)threadIdx.x * kScanItems, n, aligned && (tile + 2) * kScanTile <= n, nx);
        uint32_t tsum = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) tsum += (uint32_t)v[k];
        uint32_t total;
        if (tile == t0) {
            pacc = wave_sum(pacc);
            if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = pacc;
        }
        // (this tile's half of s_wave was last read before the previous tile's barrier: nobody is still in it)
        const uint32_t offs = tile_offsets(tsum, s_wave + ((tile - t0) & 1) * kWaves, &total);
        if (tile == t0) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) carry += s_pre[w];
        }
        uint32_t run = carry + offs;
        int32_t o[kScanItems];
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            o[k] = (int32_t)run;
            run += (uint32_t)v[k];
        }
        if (fast) {
#pragma unroll
            for (int q = 0; q < kScanItems / 4; ++q)
                *reinterpret_cast<int4 *>(out + base + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k)
                if (base + k < n) out[base + k] = o[k];
        }
        carry += total;
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k) v[k] = nx[k];
        }
    }
}

// ---- the scan in ONE launch (round 5): reduce, chained prefix, apply -- 8 bytes of HBM traffic per element ------------------------------------
// A workgroup draws a chunk by ticket (so that every chunk with a lower index is already running), sums it (the first read: HBM), publishes
// the sum tagged with the call's generation, adds up the sums of the chunks before it -- wave 0, 64 at a time, polling the few that are not
// there yet: their owners hold earlier tickets, are running and wait for nobody before they publish, so every poll ends -- and then scans
// the chunk from that prefix (the second read comes out of the caches: a chunk is 128 KB at 2^26 elements, and the 2048 of them together fit
// the 256 MB memory-side cache).  `state` = {ticket, -, agg[kScanChunksMax] as (sum << 32 | generation)}: the ticket is zeroed by the host
// before the launch (4 bytes), the sums need no clearing (a stale generation reads as "not there yet").
// Measured (profiles/r05_scan_summary.txt): 0.214 ms at 2^26 against 0.172 of the three-launch form -- the chunks of 2048 resident
// workgroups do not survive in the caches between their two reads; opt-in (PT_AMD_SCAN=1), not the default.
__global__ __launch_bounds__(kBlock) void k_scan_chained(const int32_t *__restrict__ in, int32_t *__restrict__ out, long long n,
                                                         long long tilesPerChunk, unsigned long long *state, uint32_t gen) {
    __shared__ uint32_t s_wave[kWaves];
    __shared__ uint32_t s_bcast[2];
    if (threadIdx.x == 0) s_bcast[0] = atomicAdd(reinterpret_cast<uint32_t *>(state), 1u);
    __syncthreads();
    const long long chunk = (long long)s_bcast[0];
    unsigned long long *const agg = state + 1;
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const long long t0 = chunk * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    // 1. the chunk's sum
    uint32_t acc = 0;
    for (long long tile = t0; tile < t1; ++tile) {
        const long long base = tile * kScanTile + (long long)threadIdx.x * kScanItems;
        int32_t v[kScanItems];
        load_items(in, base, n, aligned && (tile + 1) * kScanTile <= n, v);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) acc += (uint32_t)v[k];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    // 2. publish it; the prefix = the sums of every chunk before this one
    if (threadIdx.x < 64) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) total += s_wave[w];
        if (threadIdx.x == 0)
            __hip_atomic_store(&agg[chunk], ((unsigned long long)total << 32) | gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (relaxed: sum and tag travel in ONE word; a release would write the whole L2 back)
        uint32_t pre = 0;
        for (long long j0 = 0; j0 < chunk; j0 += 64) {
            const long long j = j0 + (long long)threadIdx.x;
            uint32_t val = 0;
            if (j < chunk) {
                unsigned long long e;
                do {        // (bounded by the predecessors' own reduce: they run, and publish before they wait for anything)
                    e = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((uint32_t)e != gen);
                val = (uint32_t)(e >> 32);
            }
            pre += wave_sum(val);
        }
        if (threadIdx.x == 0) s_bcast[1] = pre;
    }
    __syncthreads();
    uint32_t carry = s_bcast[1];
    // 3. the chunk once more (from the caches), scanned from its prefix
    for (long long tile = t0; tile < t1; ++tile) {
        const long long base = tile * kScanTile + (long long)threadIdx.x * kScanItems;
        const bool fast = aligned && (tile + 1) * kScanTile <= n;
        int32_t v[kScanItems];
        load_items(in, base, n, fast, v);
        uint32_t tsum = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) tsum += (uint32_t)v[k];
        uint32_t total;
        uint32_t run = carry + tile_offsets(tsum, s_wave, &total);
        int32_t o[kScanItems];
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            o[k] = (int32_t)run;
            run += (uint32_t)v[k];
        }
        if (fast) {
#pragma unroll
            for (int q = 0; q < kScanItems / 4; ++q)
                *reinterpret_cast<int4 *>(out + base + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k)
                if (base + k < n) out[base + k] = o[k];
        }
        carry += total;
        __syncthreads();                       // s_wave is free again
    }
}

// 2b. stable compaction of a chunk's non-zero elements by wave-level ballot / mbcnt (the north star's primitives; round 5).  A tile is read
// STRIPED -- step k of a wave holds 64 CONSECUTIVE elements, k * 256 + its lanes -- so that a step's survivors are ranked by one ballot and
// one mbcnt and leave for LDS as one run of consecutive words: no bank conflict by construction.  (Rounds 2-4: every thread scattered the
// survivors of its own 16 consecutive elements to s_stage[off++] -- lanes of a wave wrote to unrelated banks: 31 % of the LDS-active cycles
// were conflicts, profiles/r02_scan_summary.txt.)  The 64 (step, wave) counts of a tile are scanned by wave 0 (DPP) through LDS; the
// staged run leaves with coalesced stores at the chunk's running position.
// `partial` holds the chunks' COUNTS as k_scan_reduce<true> left them; the workgroup adds up the ones before its chunk itself, inside its first
// tile's barrier (see k_scan_apply), and the one that takes the last chunk delivers the grand total: two launches (round 6)
template <bool PIPE>
__global__ __launch_bounds__(kBlock) void k_compact_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, long long n,
                                                          long long tilesPerChunk, const uint32_t *__restrict__ partial, long long *count_out) {
    __shared__ uint32_t s_cnt[kScanItems * kWaves + 1];      // [step][wave] survivors, then their exclusive prefix; [64] = the tile's total
    __shared__ int32_t s_stage[kScanTile];
    __shared__ uint32_t s_pre[kWaves];
    static_assert(kScanItems * kWaves == 64, "wave 0 scans the (step, wave) counts with one value per lane");
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const long long chunk = (long long)gridDim.x - 1 - blockIdx.x;           // (newest first: see k_scan_apply)
    const long long t0 = chunk * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto load = [&](long long tile, int32_t (&w)[kScanItems]) {
        const long long tb = tile * kScanTile + threadIdx.x;
        const bool whole = (tile + 1) * kScanTile <= n;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) w[k] = (whole || tb + (long long)k * kBlock < n) ? in[tb + (long long)k * kBlock] : 0;
    };
    int32_t v[kScanItems], nx[kScanItems];
    if (PIPE && t0 < t1) load(t0, v);
    uint32_t pacc = 0;
    {
        uint32_t pv[kScanChunksMax / kBlock];
#pragma unroll
        for (int k = 0; k < kScanChunksMax / kBlock; ++k) {
            const long long i = (long long)k * kBlock + threadIdx.x;
            pv[k] = i < chunk ? partial[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kScanChunksMax / kBlock; ++k) pacc += pv[k];
    }
    const uint32_t own = partial[chunk];
    long long dst = 0;
    for (long long tile = t0; tile < t1; ++tile) {
        if (!PIPE) load(tile, v);
        if (PIPE && tile + 1 < t1) load(tile + 1, nx);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            const unsigned long long b = __ballot(v[k] != 0);
            if (lane == 0) s_cnt[k * kWaves + wave] = (uint32_t)__popcll(b);
        }
        if (tile == t0) {
            pacc = wave_sum(pacc);
            if (lane == 0) s_pre[wave] = pacc;
        }
        __syncthreads();
        if (tile == t0) {
            uint32_t pre = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) pre += s_pre[w];
            dst = (long long)pre;
            if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = (long long)pre + (long long)own;      // (workgroup 0 takes the LAST chunk)
        }
        if (wave == 0) {
            const uint32_t c = s_cnt[lane], inc = wave_inclusive(c);
            s_cnt[lane] = inc - c;
            if (lane == 63) s_cnt[64] = inc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            const unsigned long long b = __ballot(v[k] != 0);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (v[k] != 0) s_stage[s_cnt[k * kWaves + wave] + rank] = v[k];
        }
        const uint32_t total = s_cnt[64];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) out[dst + i] = s_stage[i];
        dst += (long long)total;
        __syncthreads();                       // s_stage and s_cnt are free again
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < kScanItems; ++k) v[k] = nx[k];
        }
    }
}

}  // namespace ptk
