// pt_experiments.h -- everything the DIAGNOSTIC builds of the render kernels add (never included by the product build, see pt_device.h):
//   make probe     (-DPT_PROBE)           phase execution counters + residency census   -> libpt_amd_probe.so     profiles/probe_phases.py, census.py
//   make timeline  (-DPT_PROBE_TIMELINE)  in-kernel timeline of a tile (s_memtime stamps) -> libpt_amd_timeline.so  profiles/timeline_phases.py
//   make marks     (-DPT_MARK)            static phase marks in the ISA listing           -> /tmp/pt_marks/*.s       profiles/phase_instructions.py, phase_cycles.py
//   make exp EXP=n (-DPT_EXP=n)           timing experiments (what is a tile's time sensitive to?) -> libpt_amd_exp<n>.so  profiles/exp_*.sh
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {
// Instrumentation for profiles/probe_phases.py (built with -DPT_PROBE only): wave-level executions and active lanes
// of the phases of the two intersection tests.  g_probe[2k] += 1 per wave that enters phase k, g_probe[2k+1] += lanes.
#ifdef PT_PROBE_TIMELINE
// A timeline (MI355X_MICROARCH.md: in-kernel stamps, diagnostic build only, make timeline; nothing else is instrumented in it): the shader cycles that wave 0 of every workgroup spends
// between two consecutive marks, summed per phase = the mark the interval starts at (g_phaseT[k], intervals counted in g_phaseN[k]).
// The stamps cost ~10 % and go nowhere but these two arrays.
// Marks 30 / 31 open a launch (later bounces / the camera-ray bounce, whose sums are kept apart: index + 32).
__device__ unsigned long long g_phaseT[64], g_phaseN[64];
// (sums are kept in LDS and flushed once, at mark 29 = the end of the kernel: global atomics at every mark would queue the wave's own
// loads and stores behind them and inflate exactly the phases that touch memory)
__device__ __forceinline__ void phaseStamp(int k) {
    __shared__ unsigned long long s_phaseLast[3];            // [0] the last stamp, [1] the mark it was taken at, [2] 32 in the camera-ray launch
    __shared__ unsigned long long s_phaseT[32];
    __shared__ unsigned int s_phaseN[32];
    if (threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (k >= 30) {
            s_phaseLast[2] = k == 31 ? 32ull : 0ull;
            for (int q = 0; q < 32; ++q) { s_phaseT[q] = 0ull; s_phaseN[q] = 0u; }
            k = 30;
        } else {
            const unsigned int idx = (unsigned int)(s_phaseLast[1] & 31ull);
            s_phaseT[idx] += t - s_phaseLast[0];
            s_phaseN[idx] += 1u;
            if (k == 29)
                for (int q = 0; q < 32; ++q)
                    if (s_phaseN[q]) {
                        atomicAdd(&g_phaseT[q + s_phaseLast[2]], s_phaseT[q]);
                        atomicAdd(&g_phaseN[q + s_phaseLast[2]], (unsigned long long)s_phaseN[q]);
                    }
        }
        s_phaseLast[0] = __builtin_amdgcn_s_memtime();
        s_phaseLast[1] = (unsigned long long)k;
    }
}
__device__ __forceinline__ void probe(int k) {
    if (k >= 9 && k < 32) phaseStamp(k);        // (the tile-level marks only; 40 and up: the mesh walk's listing marks: the marks inside the intersection tests would dominate what they measure)
}
__device__ __forceinline__ void probeCount(int, bool) {}
__device__ __forceinline__ void censusEnter() {}
__device__ __forceinline__ void censusLeave() {}
#elif defined(PT_PROBE)
__device__ unsigned long long g_probe[64];
// residency census (MI355X_MICROARCH.md: "verify with a census kernel"): workgroups of k_bounce resident on each CU right
// now and the most there ever were, keyed by (XCC, SE, SH, CU) from the hardware id registers
__device__ unsigned int g_censusNow[4096], g_censusMax[4096];
__device__ __forceinline__ unsigned censusKey() {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hwid >> 8) & 15u, sh = (hwid >> 12) & 1u, se = (hwid >> 13) & 7u;
    return ((xcc & 15u) << 8) | (se << 5) | (sh << 4) | cu;
}
__device__ __forceinline__ void censusEnter() {
    if (threadIdx.x == 0) {
        const unsigned k = censusKey();
        atomicMax(&g_censusMax[k], atomicAdd(&g_censusNow[k], 1u) + 1u);
    }
}
__device__ __forceinline__ void censusLeave() {
    if (threadIdx.x == 0) atomicSub(&g_censusNow[censusKey()], 1u);
}
__device__ __forceinline__ void probe(int k) {
    if (k >= 14) return;      // (14-23 are marks of the static listing and of the timeline: g_probe holds counters 0-13)
    const unsigned long long m = __ballot(1);
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0) {
        atomicAdd(&g_probe[2 * k], 1ull);
        atomicAdd(&g_probe[2 * k + 1], (unsigned long long)__popcll(m));
    }
}
// (counters 16 .. 31 of the same array: events of the lanes for which `cond` holds, e.g. the box test's fast path giving up, by cause)
__device__ __forceinline__ void probeCount(int k, bool cond) {
    const unsigned long long m = __ballot(cond);
    if (m != 0ull && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)__builtin_ctzll(__ballot(1))) {
        atomicAdd(&g_probe[2 * k], 1ull);
        atomicAdd(&g_probe[2 * k + 1], (unsigned long long)__popcll(m));
    }
}
#elif defined(PT_MARK)
// static phase marks in the ISA listing (make marks; profiles/phase_instructions.py counts the instructions between them)
#define probe(k) asm volatile("; PTMARK " #k)
__device__ __forceinline__ void probeCount(int, bool) {}
__device__ __forceinline__ void censusEnter() {}
__device__ __forceinline__ void censusLeave() {}
#else
__device__ __forceinline__ void probe(int) {}
__device__ __forceinline__ void probeCount(int, bool) {}
__device__ __forceinline__ void censusEnter() {}
__device__ __forceinline__ void censusLeave() {}
#endif
}  // namespace ptd

// ---- timing experiments (PT_EXP: a bit set; the hooks below stand in k_bounce's tile loop; results never change) ---------------------------
#ifndef PT_EXP
#define PT_EXP 0
#endif
// bit 0: one more dependent memory round trip per reservation (how exposed is it?)
#if PT_EXP & 1
#define PT_EXP_RESERVE(p, pos, total) if (total) p += atomicAdd(pos + 1 + (p & 7u), 1u) >> 31;
#else
#define PT_EXP_RESERVE(p, pos, total)
#endif
// bit 7: what does the sweep of the packed spheres cost?  (run it twice; the masks are the same)
#if PT_EXP & 128
#define PT_EXP_SWEEP_TWICE(base, mHi, mLo, org)      \
    {                                                \
        uint32_t xHi = 0u, xLo = 0u;                 \
        asm volatile("" : "+v"(org.x));              \
        sweep32(base, xHi);                          \
        sweep32(base + 32, xLo);                     \
        mHi |= xHi; mLo |= xLo;                      \
    }
#else
#define PT_EXP_SWEEP_TWICE(base, mHi, mLo, org)
#endif
// bit 1: one more workgroup barrier per tile
#if PT_EXP & 2
#define PT_EXP_EXTRA_BARRIER() { asm volatile("" ::: "memory"); __syncthreads(); }
#else
#define PT_EXP_EXTRA_BARRIER()
#endif
// bits 2-6: 100 more instructions of one class per wave and tile (the result feeds a flag that is never set, so nothing is optimised away and
// no result changes): 4 vector with VGPR operands only, 8 vector with one SGPR operand, 16 scalar ALU, 64 compare into an SGPR pair + select on it,
// 32: 20 dependent scalar loads (a latency chain through the scalar cache)
#if PT_EXP & 0x7c
#define PT_EXP_TILE_LOAD(org, dir, pixHash, fl, kargs)                                                                                             \
    {                                                                                                                                              \
        float xa = org.x, xb = dir.y;                                                                                                              \
        uint32_t sa = (uint32_t)__builtin_amdgcn_readfirstlane((int)pixHash), sb = sa ^ 0x55u;                                                     \
        _Pragma("unroll 1") for (int q = 0; q < 25; ++q) {                                                                                         \
            if (PT_EXP & 4) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0" : "+v"(xa), "+v"(xb)); \
            if (PT_EXP & 8) asm volatile("v_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %2, %1\n\tv_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %2, %1" : "+v"(xa), "+v"(xb) : "s"(sa)); \
            if (PT_EXP & 16) asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0" : "+s"(sa), "+s"(sb)); \
            if (PT_EXP & 64) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %1, %1, %0, vcc" : "+v"(xa), "+v"(xb) : : "vcc"); \
        }                                                                                                                                          \
        if (PT_EXP & 32) {                                                                                                                         \
            const PT_CAS uint32_t *pp = (const PT_CAS uint32_t *)launder(kargs);                                                                   \
            uint32_t acc = 0;                                                                                                                      \
            _Pragma("unroll 1") for (int q = 0; q < 20; ++q) {                                                                                     \
                uint32_t v;                                                                                                                        \
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pp + ((acc & 3u))) : "memory");                   \
                acc = (acc + v) & 0xffu;                                                                                                           \
            }                                                                                                                                      \
            sa += acc;                                                                                                                             \
        }                                                                                                                                          \
        if (__float_as_uint(xa) + __float_as_uint(xb) + sa + sb == 0x12345677u && pixHash == 0x12345u && sa == 77u) fl |= 4u;                      \
    }
#else
#define PT_EXP_TILE_LOAD(org, dir, pixHash, fl, kargs)
#endif
