// pt_compaction.h -- the stream-compaction library (the reference's empty stream_compaction/ stub, README.md:83-86):
// work-efficient single-pass multi-workgroup exclusive scan and stable compaction on gfx950.
// Included by pt_api.hip only.
#pragma once
#include "pt_common.h"

namespace ptk {

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

}  // namespace ptk
