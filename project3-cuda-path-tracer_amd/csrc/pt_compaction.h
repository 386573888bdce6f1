// pt_compaction.h -- the stream-compaction library (the reference's empty stream_compaction/ stub, README.md:83-86):
// work-efficient multi-workgroup exclusive scan and stable compaction on gfx950.
// Included by pt_api.hip only.
//
// Reduce-then-scan over CHUNKS: the array is cut into at most 2048 chunks of whole 4096-element tiles;
//   1. k_scan_reduce    one workgroup per chunk: its sum (scan) or its count of non-zero elements (compaction)
//   2. k_scan_partials  one workgroup: exclusive scan of the chunk totals (they fit one tile)
//   3. k_scan_apply / k_compact_apply   one workgroup per chunk: re-reads the chunk and writes its part of the result,
//      starting from the chunk's prefix
// i.e. 12 bytes of traffic per element for the scan (8 is the minimum) and 8 + 4 per kept element for the compaction,
// three launches on the caller's stream, and NO workgroup ever waits for another one: no tickets, no look-back, no spinning.
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
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan of one value per lane inside the wave
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(x, o, 64);
        if (lane >= o) x += up;
    }
    return x;
}
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
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_scan_reduce(const int32_t *__restrict__ in, long long n, long long tilesPerChunk,
                                                        uint32_t *__restrict__ partial) {
    __shared__ uint32_t s_wave[kWaves];
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const long long t0 = (long long)blockIdx.x * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    uint32_t acc = 0;
    for (long long tile = t0; tile < t1; ++tile) {
        const long long base = tile * kScanTile + (long long)threadIdx.x * kScanItems;
        int32_t v[kScanItems];
        load_items(in, base, n, aligned && (tile + 1) * kScanTile <= n, v);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) acc += COUNT ? (v[k] != 0 ? 1u : 0u) : (uint32_t)v[k];
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

// 2. exclusive scan of the chunk totals in place (chunks <= kScanTile); the grand total goes to partial[chunks] and, for the
// compaction, to *count_out
__global__ __launch_bounds__(kBlock) void k_scan_partials(uint32_t *partial, int chunks, long long *count_out) {
    __shared__ uint32_t s_wave[kWaves];
    uint32_t v[kScanItems], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = threadIdx.x * kScanItems + k;
        v[k] = i < chunks ? partial[i] : 0u;
        mine += v[k];
    }
    uint32_t total;
    uint32_t run = tile_offsets(mine, s_wave, &total);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = threadIdx.x * kScanItems + k;
        if (i < chunks) partial[i] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) {
        partial[chunks] = total;
        if (count_out) *count_out = (long long)total;
    }
}

// 3a. the scan of a chunk, starting from its prefix
__global__ __launch_bounds__(kBlock) void k_scan_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, long long n,
                                                       long long tilesPerChunk, const uint32_t *__restrict__ partial) {
    __shared__ uint32_t s_wave[kWaves];
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const long long t0 = (long long)blockIdx.x * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    uint32_t carry = partial[blockIdx.x];
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

// 3b. stable compaction of a chunk's non-zero elements: a tile's kept elements are gathered in LDS (in order) and leave
// with coalesced stores at the chunk's running position
__global__ __launch_bounds__(kBlock) void k_compact_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, long long n,
                                                          long long tilesPerChunk, const uint32_t *__restrict__ partial) {
    __shared__ uint32_t s_wave[kWaves];
    __shared__ int32_t s_stage[kScanTile];
    const long long numTiles = (n + kScanTile - 1) / kScanTile;
    const long long t0 = (long long)blockIdx.x * tilesPerChunk;
    const long long t1 = t0 + tilesPerChunk < numTiles ? t0 + tilesPerChunk : numTiles;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    long long dst = (long long)partial[blockIdx.x];
    for (long long tile = t0; tile < t1; ++tile) {
        const long long base = tile * kScanTile + (long long)threadIdx.x * kScanItems;
        int32_t v[kScanItems];
        load_items(in, base, n, aligned && (tile + 1) * kScanTile <= n, v);
        uint32_t kept = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) kept += v[k] != 0 ? 1u : 0u;
        uint32_t total;
        uint32_t off = tile_offsets(kept, s_wave, &total);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if (v[k] != 0) s_stage[off++] = v[k];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) out[dst + i] = s_stage[i];
        dst += (long long)total;
        __syncthreads();                       // s_stage and s_wave are free again
    }
}

}  // namespace ptk
