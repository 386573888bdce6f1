// pt_device.h -- device-side math of the MI355X path tracer (gfx950 only).
//
// Every function is a restatement, in the reference's operation order, of what its kernels
// would inline from src/intersections.h, src/interactions.h and the vendored glm 0.9.6.3
// (citations relative to the reference root).  The translation unit is compiled with
// -ffp-contract=off and correctly-rounded fp32 divide/sqrt, so each source-level operation is
// exactly one IEEE-754 fp32 operation and results are bit-identical to the reference's host
// evaluation of the same functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// Phase marks of the render kernels: probe(k) / probeCount(k, cond) / censusEnter() / censusLeave() compile to NOTHING in the product.  The
// instrumented builds (make probe / timeline / marks / exp: phase execution counters, in-kernel timelines, static marks in the ISA listing,
// timing experiments -- diagnostic libraries that are never loaded by the product) define them in pt_experiments.h, which only those
// builds include.
#if defined(PT_PROBE) || defined(PT_PROBE_TIMELINE) || defined(PT_MARK) || defined(PT_EXP)
#include "pt_experiments.h"
#else
namespace ptd {
__device__ __forceinline__ void probe(int) {}
__device__ __forceinline__ void probeCount(int, bool) {}
__device__ __forceinline__ void censusEnter() {}
__device__ __forceinline__ void censusLeave() {}
}  // namespace ptd
#define PT_EXP_RESERVE(p, pos, total)
#define PT_EXP_SWEEP_TWICE(base, mHi, mLo, org)
#define PT_EXP_TILE_LOAD(org, dir, pixHash, fl, kargs)
#define PT_EXP_EXTRA_BARRIER()
#endif

namespace ptd {

struct F3 {
    float x, y, z;
};

__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }

// glm/detail/func_geometric.inl:64-83 (compute_dot<tvec3>): tmp = x*y; tmp.x + tmp.y + tmp.z
__device__ __forceinline__ float dot(F3 a, F3 b) {
    F3 t = a * b;
    return t.x + t.y + t.z;
}
// glm/detail/func_geometric.inl:134-143
__device__ __forceinline__ F3 cross(F3 x, F3 y) {
    return f3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// glm/detail/func_geometric.inl:154-159 with inversesqrt = 1/sqrt (func_exponential.inl:150-153)
__device__ __forceinline__ F3 normalize(F3 a) { return a * (1.0f / __builtin_sqrtf(dot(a, a))); }
// glm/detail/func_geometric.inl:95-100
__device__ __forceinline__ float length(F3 a) { return __builtin_sqrtf(dot(a, a)); }

// ---- correctly rounded sqrt without its range handling ------------------------------------------------
// hipcc expands a correctly rounded fp32 sqrt into 16 VALU instructions; seven of them only act near the exponent
// limits (operands below 2^-96 are pre-scaled by 2^32 and the root by 2^-16; +-0 / +inf are passed through).
// sqrtUnscaled issues the remaining nine -- v_sqrt_f32, its two neighbours and their fma residuals -- and is
// bit-identical to __builtin_sqrtf for  x == +-0  or  2^-96 <= x < inf
// (tests/test_gpu_parity.py::test_unscaled_sqrt_exhaustive compares EVERY float of that range).  It is used where
// the operand is inside that range by construction; behind a run-time range test the compare + branch + copies cost
// what the shorter sequence saves (measured: no change in SQ_INSTS_VALU), so normalize / length keep the plain form.
__device__ __forceinline__ float sqrtUnscaled(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = 0.0f >= rd ? sd : s;
    r = 0.0f < ru ? su : r;
    return r;
}
// glm/detail/func_geometric.inl:176-179 : I - N * dot(N, I) * 2
__device__ __forceinline__ F3 reflect(F3 I, F3 N) { return I - (N * dot(N, I)) * 2.0f; }
// glm/detail/func_geometric.inl:193-200.  NaN when k < 0 (sqrt of a negative times 0): callers test k.
__device__ __forceinline__ F3 refract(F3 I, F3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    F3 r = I * eta - N * (eta * d + __builtin_sqrtf(k));
    return r * (k >= 0.0f ? 1.0f : 0.0f);
}

// src/utilities.h:12-15
constexpr float kTwoPi = 6.2831853071795864769252867665590057683943f;
constexpr float kSqrtOneThird = 0.5773502691896257645091487805019574556476f;

// ---- scene data ------------------------------------------------------------------------------------
// Scalar data of the kernels is read through CONSTANT-address-space pointers (always s_load -> SGPR operands), and the
// pointers are "laundered" through an empty asm where a phase of the kernel starts: the loads then cannot be hoisted
// out of the tile / primitive loops, whose live ranges would otherwise pile up in SGPRs (round 1: 106 SGPRs and up to
// 51 spilled to VGPR lanes, six workgroups per CU instead of eight).
#define PT_CAS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const PT_CAS T *launder(const PT_CAS T *p) {
    asm volatile("" : "+s"(p));
    return p;
}

// Rows 0..2 of the three column-major mat4 of a Geom (the w row is never used: multiplyMV clips
// to vec3, src/intersections.h:33-35).  m[col*3 + row].  Laid out by phase of the nearest-hit loop, each group
// 16- or 8-dword aligned so that it arrives in one scalar load.
struct GeomDev {
    // ---- 0x00: the test (one s_load_dwordx16)
    float inv[12];   // inverseTransform
    // inverseTransform's translation column times 0.0f (a signed zero each, or NaN): the w = 0 products of a direction
    // transform, multiplyMV(inverseTransform, (d, 0)), evaluated once instead of once per ray
    float invZ[3];
    int   flags;     // bit 0: type (0 sphere, 1 cube), bit 1: binned, bits 2-4: 1 + index among the scene's walls (0: not one),
                     // bit 5: a triangle mesh (bit 0 clear: shaded like a sphere, from an object-space normal vector)
    // ---- 0x40: camera rays (first bounce only)
    // object-space camera position multiplyMV(inverseTransform, (eye, 1)), evaluated once on the host with the
    // same operation order: every camera ray of the first bounce shares it
    float camObj[3];
    uint32_t meshRoot;   // a mesh's first node in the scene's node array (MeshNode)
    // Pixels whose camera rays can reach this primitive: inclusive bounds [x0, y0, x1, y1] of the projection of its
    // object-space unit cube, widened by 2 pixels (host, double precision); the whole frame when a corner is not in
    // front of the eye.  Camera rays of other pixels skip the primitive (first bounce only).
    int   rect[4];
    // ---- 0x60: conservative world-space culling (certainMiss) against the primitive's bounding ball: centre, rho^2 smax^2 (1 + 1e-3)
    // with rho^2 = 1/4 (sphere) or 3/4 (cube: half its diagonal), 1e-4 (smax / smin)^2 with smax / smin bounds of the
    // transform's singular values
    float centre[3];
    float cullR2, cullK;
    float boundR;    // radius of the bounding ball, rho smax (host bookkeeping: which primitives are small)
    int   cullFlags; // = flags: the loop over the primitives reads THIS group first (type and binned bit, and for a sphere
                     // everything the bounding-ball test needs: one scalar load per sphere), the test group only for a cube
    int   material;
    // ---- 0x80: a hit
    float xf[12];    // transform
    float invT[12];  // invTranspose
    // cube: for each of its six faces (entry axis * 2 + (sign > 0)) the surface normal
    // normalize(multiplyMV(transform, (+-e_axis, 0))) (src/intersections.h:85) and the two tangent directions the
    // hemisphere sampler derives from a normal (src/interactions.h:22-35), 9 floats per face, evaluated once on the host
    // with the operations the kernels would issue per hit
    union {
        float cubeFrame[54];
        struct {
            uint32_t meshStride;   // a mesh: nodes per copy of its hierarchy; the copy for direction octant k starts at meshRoot + k * meshStride
            uint32_t meshUnit0, meshUnit1;   // ... and the units of the scene's record array that hold its triangles: [meshUnit0, meshUnit1)
        };
    };
    // 1: a small primitive the queue is binned by (KParams::binGeom): tiles of paths that certainly miss all of them
    // skip it (mirrored in flags)
    short binned;
    short frameSlot;   // a cube's ordinal among the scene's cubes (sphere-heavy scenes: its row in the LDS frame table)
    int   type;      // 0 sphere, 1 cube (src/sceneStructs.h:8-11); a mesh is 0 here (its normal is made like a sphere's) and flags bit 5
};
static_assert(sizeof(GeomDev) == 448, "GeomDev is 28 x 16 B");
static_assert(offsetof(GeomDev, camObj) == 64 && offsetof(GeomDev, centre) == 96 && offsetof(GeomDev, xf) == 128, "scalar-load groups");

struct MaterialDev {
    float color[3];
    float specColor[3];
    float hasReflective, hasRefractive, ior, emittance;
    // per-material constants of the dielectric branch, evaluated once on the host with the same IEEE operations the
    // shader would issue per path: 1.0f / ior and Schlick's r0 = ((1 - ior) / (1 + ior))^2
    float invIor, r0;
    // imperfect specular (SPECEX > 0, reference README.md:171-185): 1.0f / (exponent + 1.0f), evaluated on the host; 0 = perfect mirror
    float invSpecExp1;
    float pad[3];
};
static_assert(sizeof(MaterialDev) == 64, "MaterialDev is 4 x 16 B");

// multiplyMV (src/intersections.h:33-35) = vec3(m * v), glm/detail/type_mat4x4.inl:617-628:
// (m0*v0 + m1*v1) + (m2*v2 + m3*v3).  The products with w = 0 / w = 1 are kept as IEEE ops
// (x*1 folds exactly; x*0 keeps its signed zero), so signed zeros match the reference too.
template <typename P>
__device__ __forceinline__ F3 mulMV(P m, F3 v, float w) {
    F3 r;
    r.x = (m[0] * v.x + m[3] * v.y) + (m[6] * v.z + m[9] * w);
    r.y = (m[1] * v.x + m[4] * v.y) + (m[7] * v.z + m[10] * w);
    r.z = (m[2] * v.x + m[5] * v.y) + (m[8] * v.z + m[11] * w);
    return r;
}

// mulMV(m, v, 0) with the three products m[9 + r] * 0.0f supplied (z0): same sums, three multiplications less
template <typename P, typename Q>
__device__ __forceinline__ F3 mulMV0(P m, Q z0, F3 v) {
    F3 r;
    r.x = (m[0] * v.x + m[3] * v.y) + (m[6] * v.z + z0[0]);
    r.y = (m[1] * v.x + m[4] * v.y) + (m[7] * v.z + z0[1]);
    r.z = (m[2] * v.x + m[5] * v.y) + (m[8] * v.z + z0[2]);
    return r;
}

// src/intersections.h:11-19
__device__ __forceinline__ uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

// thrust::default_random_engine = minstd_rand: x' = 48271 x mod (2^31 - 1); seed s -> s mod m, 0 -> 1
// (thrust/random/detail/linear_congruential_engine.inl).  The Mersenne-prime fold replaces
// thrust's Schrage form (thrust/random/detail/mod.h); both equal (48271 * x) mod m.
struct Rng {
    uint32_t x;
};
// p mod (2^31 - 1) for a 32-bit p: one fold, (p & m) + (p >> 31) <= m + 1, and one conditional subtraction
__device__ __forceinline__ uint32_t mod_m31(uint32_t p) {
    const uint32_t m = 2147483647u;
    const uint32_t q = (p & m) + (p >> 31);
    return q >= m ? q - m : q;
}
// 48271 x mod (2^31 - 1) for x < 2^31: the product is below 2^47, so its fold (p & m) + (p >> 31) stays below
// 2^31 + 2^16 < 2 m: one 32 x 32 -> 64 multiply, 32-bit arithmetic from there (p >> 31 is one v_alignbit_b32)
__device__ __forceinline__ uint32_t lcgStep(uint32_t x) {
    const uint32_t m = 2147483647u;
    const uint64_t p = (uint64_t)x * 48271ull;
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32);
    const uint32_t q = (lo & m) + __builtin_amdgcn_alignbit(hi, lo, 31);
    return q >= m ? q - m : q;
}
// src/pathtrace.cu:41-45
// the iteration/depth half of the seed: the same word for every pixel of a launch (per iteration of its batch), so the
// render kernel hashes it once per workgroup into LDS instead of once per path
__device__ __forceinline__ uint32_t iterationHash(int iter, int depth) {
    return utilhash((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter);
}
__device__ __forceinline__ Rng makeSeededRandomEngineHashed(uint32_t iterHash, int index) {
    uint32_t x = mod_m31(iterHash ^ utilhash((uint32_t)index));
    Rng r;
    r.x = x == 0u ? 1u : x;
    return r;
}
__device__ __forceinline__ Rng makeSeededRandomEngine(int iter, int index, int depth) {
    return makeSeededRandomEngineHashed(iterationHash(iter, depth), index);
}
__device__ __forceinline__ Rng seedEngine(uint32_t h) {
    uint32_t x = mod_m31(h);
    Rng r;
    r.x = x == 0u ? 1u : x;
    return r;
}
// thrust::uniform_real_distribution<float>(0,1) (uniform_real_distribution.inl:71-79):
// float(x - min) / (1.0f + float(max - min)) = float(x - 1) / 2^31 (both roundings give 2^31).
__device__ __forceinline__ float u01(Rng &r) {
    r.x = lcgStep(r.x);
    return (float)(r.x - 1u) * 4.656612873077392578125e-10f;  // exact power-of-two scaling
}

// Build-defined sin/cos, identical to the CPU oracle's polynomial (the reference calls the
// platform libm, src/interactions.h:40-41): Cody-Waite reduction by pi/2 + minimax polynomials.
__device__ __forceinline__ void sincosPoly(float x, float &s, float &c) {
    float kf = __builtin_rintf(x * 0.636619747f);
    int k = (int)kf;
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r;
    float sr = ps + r;
    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    float cr = (pc - 0.5f * z) + 1.0f;
    float s0 = (k & 1) ? cr : sr;
    float c0 = (k & 1) ? sr : cr;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
}

// Build-defined x^e for 0 <= x <= 1, 0 < e <= 1 (imperfect specular: cos(theta) = xi^(1/(n+1)), GPU Gems 3 ch. 20 eq. 7-9,
// named by the reference's README.md:171-185): exp2(e * log2(x)) with the cephes logf / exp2f polynomials, operation for
// operation the CPU oracle's pow_poly (relative error ~2e-6).
__device__ __forceinline__ float powPoly(float x, float e) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    const uint32_t bits = __float_as_uint(x);
    if (bits < 0x00800000u) return 0.0f;
    int k = (int)(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; k += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = p * f - 1.1514610310e-1f;
    p = p * f + 1.1676998740e-1f;
    p = p * f - 1.2420140846e-1f;
    p = p * f + 1.4249322787e-1f;
    p = p * f - 1.6668057665e-1f;
    p = p * f + 2.0000714765e-1f;
    p = p * f - 2.4999993993e-1f;
    p = p * f + 3.3333331174e-1f;
    float y = (f * z) * p;
    y = y - 0.5f * z;
    const float ln = f + y;
    const float l2 = ln * 1.44269504088896341f + (float)k;
    const float t = e * l2;
    if (t < -126.0f) return 0.0f;
    const float nf = __builtin_rintf(t);
    const float g = t - nf;
    float q = 1.535336188319500e-4f;
    q = q * g + 1.339887440266574e-3f;
    q = q * g + 9.618437357674640e-3f;
    q = q * g + 5.550332471162809e-2f;
    q = q * g + 2.402264791363012e-1f;
    q = q * g + 6.931472028550421e-1f;
    const float r = q * g + 1.0f;
    return __uint_as_float(__float_as_uint(r) + (uint32_t)((int)nf << 23));
}

// 1.0f / sqrtf(x), both correctly rounded, for x within 256 ulps of 1 without the square root and the division.
// getPointOnRay re-normalises a direction both intersection tests have just normalised, so its dot product is
// 1 - k 2^-24 or 1 + k 2^-23 with k <= 6 in practice, and then
//   x = 1 + k 2^-23:  sqrt rounds to 1 + floor(k/2) 2^-23 =: 1 + j 2^-23,  its reciprocal to 1 - 2j 2^-24
//   x = 1 - k 2^-24:  sqrt rounds to 1 - ceil(k/2) 2^-24  =: 1 - j 2^-24,  its reciprocal to 1 + ceil(j/2) 2^-23
// (the neglected second-order terms stay below half an ulp up to k ~ 1400; every other x takes the two operations).
// Integer arithmetic on the bit pattern, 13 instructions instead of 27 with two quarter-rate ones;
// tests/test_gpu_parity.py::test_unscaled_sqrt_exhaustive compares it with 1.0f / sqrtf(x) on every fp32 bit pattern.
__device__ __forceinline__ float inverseSqrtNearOne(float x) {
    const int b = (int)__float_as_uint(x) - 0x3f800000;           // ulps above (spacing 2^-23) or below (2^-24) one
    // the closed form is evaluated for every lane (integer arithmetic, cannot trap); the general form is a one-sided,
    // practically never taken fix-up -- one divergent region instead of an if / else pair
    const int above = 0x3f800000 - (b & ~1);                      // 1 - 2 floor(b/2) 2^-24
    const int below = 0x3f800000 + ((((1 - b) >> 1) + 1) >> 1);   // 1 + ceil(ceil(k/2)/2) 2^-23, k = -b
    float r = __uint_as_float((uint32_t)(b >= 0 ? above : below));
    if (!((unsigned)(b + 256) <= 512u)) r = 1.0f / __builtin_sqrtf(x);
    return r;
}

// src/intersections.h:26-28 : origin + (t - .0001f) * normalize(direction)
__device__ __forceinline__ F3 getPointOnRay(F3 origin, F3 direction, float t) {
    return origin + (direction * inverseSqrtNearOne(dot(direction, direction))) * (t - .0001f);   // = normalize(direction) * ...
}

// ---- the two slab quotients of one axis: t1 = (-.5 - o) / d, t2 = (+.5 - o) / d ---------------------
// Correctly rounded fp32 division is ~11 VALU instructions on gfx950 (v_div_scale x2, v_rcp, 4 fma, mul,
// v_div_fmas, v_div_fixup) and hipcc does not share anything between two quotients of one divisor.
// Inside the range where v_div_scale_f32 leaves both operands unscaled and v_div_fixup_f32 returns its
// first operand (both operands normal and far from the exponent limits), that sequence is exactly
//     r = rcp(d); r += fma(-d, r, 1) * r;  q = a * r;  q += fma(-d, q, a) * r;  q += fma(-d, q, a) * r
// The same instructions are issued here ONCE for the shared reciprocal and as packed (v_pk_*) pairs for
// the two numerators; outside the guarded range the plain `/` is used.  Bit-identical to `/` by
// construction; tests/test_gpu_parity.py::test_slab_quotients_equal_ieee_division checks it on 10^9 pairs.
// Numerators: o is a float, so a = +-.5 - o is either +0 (o = +-.5) or at least 2^-25 in magnitude, never
// -0 and never tiny; the guard therefore only bounds |o| and |d|.
typedef float float2v __attribute__((ext_vector_type(2)));

// The guarded sequence, evaluated whatever the operands (none of its instructions can trap); returns whether the guard
// holds, i.e. whether t1, t2 ARE the quotients.
__device__ __forceinline__ bool slabQuotientsFast(float o, float d, float &t1, float &t2) {
    const float a1 = -0.5f - o, a2 = +0.5f - o;
    const float ad = __builtin_fabsf(d);
    const bool fast = (__builtin_fabsf(o) <= 0x1p+39f) & (ad >= 0x1p-40f) & (ad <= 0x1p+40f);   // NaN fails all three
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    const float2v nd = {-d, -d}, rr = {r, r}, a = {a1, a2};
    float2v q = a * rr;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q, a), rr, q);
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q, a), rr, q);
    t1 = q.x;
    t2 = q.y;
    return fast;
}
__device__ __forceinline__ void slabQuotients(float o, float d, float &t1, float &t2) {
    if (!slabQuotientsFast(o, d, t1, t2)) {
        t1 = (-0.5f - o) / d;
        t2 = (+0.5f - o) / d;
    }
}

// ---- the slab phase of the box test (src/intersections.h:51-77), twice ---------------------------------
// Both forms leave: hit = `tmax >= tmin && tmax > 0`, and on a hit tmin = the parameter the reference goes on with (tmax when the origin
// is inside), outs = `outside`, face = the slab normal tmin_n = +-e_axis as axis * 2 + (sign > 0) -- or -1: no slab ever updated it (a
// ray of NaNs "hits" with the zero vector for a normal, which the reference then normalises: every derived vector is NaN).
//
// boxSlabsExact: the reference's loop as it stands -- normalize, six correctly rounded quotients, the sequential min / max updates.
__device__ __forceinline__ void boxSlabsExact(F3 qo, F3 qdu, F3 &qd, bool &hit, bool &outs, float &tminOut, int &face) {
    qd = normalize(qdu);
    float tmin = -1e38f, tmax = 1e38f;
    // the slab normal n (zero except n[xyz] = +-1) is tracked as its face index instead of a vector
    int tmin_face = -1, tmax_face = -1;
    const float qoa[3] = {qo.x, qo.y, qo.z};
    const float qda[3] = {qd.x, qd.y, qd.z};
    // the three axes' quotient pairs by the guarded sequence, and ONE practically never taken region that redoes the
    // axes whose guard failed with the plain division (instead of an if / else per axis)
    float t1a[3], t2a[3];
    const bool ok0 = slabQuotientsFast(qoa[0], qda[0], t1a[0], t2a[0]);
    const bool ok1 = slabQuotientsFast(qoa[1], qda[1], t1a[1], t2a[1]);
    const bool ok2 = slabQuotientsFast(qoa[2], qda[2], t1a[2], t2a[2]);
    if (!(ok0 & ok1 & ok2)) {
        const bool ok[3] = {ok0, ok1, ok2};
#pragma unroll
        for (int xyz = 0; xyz < 3; ++xyz) {
            const float p1 = (-0.5f - qoa[xyz]) / qda[xyz], p2 = (+0.5f - qoa[xyz]) / qda[xyz];
            t1a[xyz] = ok[xyz] ? t1a[xyz] : p1;
            t2a[xyz] = ok[xyz] ? t2a[xyz] : p2;
        }
    }
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz) {
        const float t1 = t1a[xyz], t2 = t2a[xyz];
        const float ta = t1 < t2 ? t1 : t2;  // glm::min, func_common.inl:409-414
        const float tb = t1 > t2 ? t1 : t2;  // glm::max, func_common.inl:430-435
        const int nf = 2 * xyz + (t2 < t1 ? 1 : 0);      // n[xyz] = t2 < t1 ? +1 : -1
        if (ta > 0 && ta > tmin) {
            tmin = ta;
            tmin_face = nf;
        }
        if (tb < tmax) {
            tmax = tb;
            tmax_face = nf;
        }
    }
    hit = tmax >= tmin && tmax > 0;
    outs = true;
    if (tmin <= 0) {
        tmin = tmax;
        tmin_face = tmax_face;
        outs = false;
    }
    tminOut = tmin;
    face = tmin_face;
}

// a / d for operands inside the range where the compiler's correctly rounded division neither scales nor fixes anything up (see
// slabQuotientsFast: the very instructions of that expansion, minus v_div_scale / v_div_fixup); `r0` = v_rcp_f32(d)
__device__ __forceinline__ float divUnscaled(float a, float d, float r0) {
    const float r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
    return q;
}

// boxSlabsFastDecide + boxSlabsFastFinish (round 5): the same outputs, bit for bit, for half of the instructions -- or `false`: no
// statement, run boxSlabsExact.
// The reference's loop needs SIX correctly rounded quotients (~11 instructions each) to take ONE of them as its result; the other
// five only take part in comparisons.  Here the comparisons are decided on approximate quotients a * v_rcp_f32(d) wherever their
// outcome is beyond doubt -- the operands differ by more than 2^-19 relative, against an error below 2^-21.5 (v_rcp_f32: one ulp;
// the product and the reference's own quotient: half an ulp each) -- and only the winner is divided exactly.  Any doubt (operands
// within the margin of each other: a ray through an edge of the cube) and any operand outside the guarded range (NaN, inf, a
// direction component below 2^-40 of its length, an object-space origin beyond 2^20, a direction transform outside [2^-40, 2^40])
// returns `false`.
// Under the guards (|o| <= 2^20, 2^-40 <= |d| <= ~1, all finite):
//   a1 = RN(-.5 - o) < a2 = RN(.5 - o) strictly (they differ by 1, an ulp is at most 2^-3), hence t1 = RN(a1 / d) and t2 = RN(a2 / d)
//   differ as well (relative distance >= 2^-21.2) and are ordered by the sign of d:  d > 0: ta = t1, tb = t2, n = -1;  d < 0: ta = t2,
//   tb = t1, n = +1.  So ta = RN(an / d), tb = RN(af / d) with the NEAR numerator an = -h - o, the FAR one af = h - o, h =
//   copysign(.5, d), and n = -sign(d) for both.
//   The loop leaves tmin = the largest positive ta (-1e38 when none is), tmax = the smallest tb, each with the FIRST axis that
//   attains it; signs of quotients are exact in either arithmetic (no underflow: |a| is 0 or >= 2^-25), so "some ta > 0" and
//   "tmax > 0" (= every tb > 0) are decided exactly; `tmax >= tmin` by the margin; the axis is the unique one whose approximate
//   quotient stands clear of the other two by the margin -- else `false`.
// Its normalize is the reference's, correctly rounded sqrt and division, without their range handling (sqrtUnscaled, divUnscaled):
// the squared length is inside [2^-80, 2^80) or the function returns `false`.
// Two parts, so that the axis and the exact quotient are evaluated for the hits only.  `axis` (Decide -> Finish): the axis K whose
// quotient is the result -- the entry axis, or the exit axis of a ray that starts inside.
// tests/test_gpu_parity.py::test_box_fast_path_equals_the_exact_one sweeps 2^28 rays dense in edges, corners and grazes; in a render
// of cornell.txt one wave in 1400 meets a lane that falls back (profiles/probe_phases.py).
__device__ __forceinline__ bool boxSlabsFastDecide(F3 qo, F3 qdu, F3 &qd, bool &hit, bool &outs, int &axis) {
    const float x = dot(qdu, qdu);
    bool ok = (__float_as_uint(x) - 0x17800000u) < 0x50000000u;            // 2^-80 <= x < 2^80 (NaN, inf, 0, negative: no)
    const float s = sqrtUnscaled(x);
    const float inv = divUnscaled(1.0f, s, __builtin_amdgcn_rcpf(s));      // = 1.0f / s: glm's inversesqrt
    qd = qdu * inv;
    ok &= (__builtin_fabsf(qo.x) + __builtin_fabsf(qo.y)) + __builtin_fabsf(qo.z) <= 0x1p+20f;     // (NaN fails)
    ok &= __builtin_fminf(__builtin_fminf(__builtin_fabsf(qd.x), __builtin_fabsf(qd.y)), __builtin_fabsf(qd.z)) >= 0x1p-40f;   // (finite: x is)
    const F3 h = f3(__builtin_copysignf(0.5f, qd.x), __builtin_copysignf(0.5f, qd.y), __builtin_copysignf(0.5f, qd.z));
    const F3 an = (-h) - qo, af = h - qo;                                 // (-.5 - o, not -(.5 + o): the zero of o = -.5 is +0)
    const F3 r0 = f3(__builtin_amdgcn_rcpf(qd.x), __builtin_amdgcn_rcpf(qd.y), __builtin_amdgcn_rcpf(qd.z));
    const F3 Ta = an * r0, Tb = af * r0;
    const float m = __builtin_fmaxf(__builtin_fmaxf(Ta.x, Ta.y), Ta.z), md = __builtin_amdgcn_fmed3f(Ta.x, Ta.y, Ta.z);
    const float M = __builtin_fminf(__builtin_fminf(Tb.x, Tb.y), Tb.z), Md = __builtin_amdgcn_fmed3f(Tb.x, Tb.y, Tb.z);
    outs = m > 0.0f;                                                      // some ta > 0 (exact)
    const bool exitPos = M > 0.0f;                                         // tmax > 0 (exact)
    const float diff = M - m, tol = 0x1p-19f * (__builtin_fabsf(M) + __builtin_fabsf(m));
    const bool sureHit = exitPos & (!outs | (diff > tol)), sureMiss = !exitPos | (outs & (-diff > tol));
    constexpr float kBelowOne = 1.0f - 0x1p-19f;
    const bool unique = (outs & (md < m * kBelowOne)) | (!outs & (Md * kBelowOne > M));   // (hits only: M > 0 there)
    probeCount(16, true);                                                 // (instrumented build: why the fast path gives up)
    probeCount(17, !ok);
    probeCount(18, !(sureHit | sureMiss));
    probeCount(19, sureHit & !unique);
    ok &= sureMiss | (sureHit & unique);
    probeCount(20, !ok);
    hit = sureHit;
    const float key0 = outs ? Ta.x : Tb.x, key1 = outs ? Ta.y : Tb.y, ref = outs ? m : M;
    axis = key0 == ref ? 0 : (key1 == ref ? 1 : 2);
    return ok;
}
// the one exact quotient of a hit
__device__ __forceinline__ void boxSlabsFastFinish(F3 qo, F3 qd, bool outs, int axis, float &tminOut, int &face) {
    const bool e0 = axis == 0, e1 = axis == 1;
    const float oK = e0 ? qo.x : (e1 ? qo.y : qo.z), dK = e0 ? qd.x : (e1 ? qd.y : qd.z);
    const float hK = __builtin_copysignf(0.5f, dK);
    const float aK = (outs ? -hK : hK) - oK;
    tminOut = divUnscaled(aK, dK, __builtin_amdgcn_rcpf(dK));
    face = 2 * axis + (int)(__float_as_uint(dK) >> 31);                  // n = -sign(d): + (face bit 1) for d < 0
}

// src/intersections.h:47-89
//
// Early miss (exact, not a heuristic): if on some axis the object-space origin lies beyond a face
// (o > .5 or o < -.5) and the ray moves further away (d has the sign of o), then in the reference's
// slab loop both quotients t1, t2 of that axis are <= -0 (a non-zero numerator over a divisor of the
// opposite sign; +-0 and inf divisors included, NaN impossible), so tb <= -0 updates tmax to a value
// <= 0 and the final `tmax > 0` fails whatever the other axes do: the test returns -1.  The sign of
// every component of normalize(v) equals the sign of v's component (the scale 1/sqrt(dot) is >= 0 or
// NaN, and NaN fails the comparisons below), so the decision is taken BEFORE the normalisation and
// the six correctly rounded divisions.  It only pays when whole waves take it, i.e. for coherent
// rays (EARLY_MISS is set for the camera-ray bounce); results are identical either way.
// CAM_ORIGIN: the ray starts at the camera, whose object-space position is precomputed (GeomDev::camObj).
// `early` (wave-uniform, with EARLY_MISS): take the early miss at all -- callers that expect nearly every ray to hit skip its compares.
// `onlyOne` (wave-uniform): the return value is only compared with 0 (see below): a hit returns the SQUARED distance.
// EXACT_ONLY: the slab phase by the reference's loop alone (the parity sweep's other side; experiments)
#ifndef PT_BOX_FAST
#define PT_BOX_FAST 1
#endif
template <bool EARLY_MISS, bool CAM_ORIGIN = false, bool EXACT_ONLY = !PT_BOX_FAST, typename GD>
__device__ __forceinline__ float boxIntersectionTest(const GD &g, F3 ro, F3 rd, F3 &P, F3 &nsrc, bool &outside, bool early = true, bool onlyOne = false) {
    probe(0);
    const F3 qo = CAM_ORIGIN ? f3(g.camObj[0], g.camObj[1], g.camObj[2]) : mulMV(g.inv, ro, 1.0f);
    const F3 qdu = mulMV0(g.inv, g.invZ, rd);
    if (EARLY_MISS && early) {
        // (bitwise on purpose: twelve compares and eleven mask operations, no nest of divergent branches)
        const bool away = ((qo.x > 0.5f) & (qdu.x > 0.0f)) | ((qo.x < -0.5f) & (qdu.x < 0.0f)) |
                          ((qo.y > 0.5f) & (qdu.y > 0.0f)) | ((qo.y < -0.5f) & (qdu.y < 0.0f)) |
                          ((qo.z > 0.5f) & (qdu.z > 0.0f)) | ((qo.z < -0.5f) & (qdu.z < 0.0f));
        if (away) return -1.0f;
    }
    probe(1);
    F3 qd;
    float tmin;
    int face, axis;
    bool hit, outs;
    if (EXACT_ONLY || !boxSlabsFastDecide(qo, qdu, qd, hit, outs, axis)) boxSlabsExact(qo, qdu, qd, hit, outs, tmin, face);
    else if (hit) boxSlabsFastFinish(qo, qd, outs, axis, tmin, face);
    if (hit) {
        probe(2);
        outside = outs;
        P = mulMV(g.xf, getPointOnRay(qo, qd, tmin), 1.0f);
        // the slab normal's face index, as bits, in nsrc.x: normal = normalize(transform * (tmin_n, 0)) is looked up by it for the nearest hit only (cubeFace)
        nsrc = f3(__int_as_float(face), 0.0f, 0.0f);
        // `onlyOne` (wave-uniform): this primitive is the only one the caller looks at, so the distance is compared with nothing but 0 --
        // and sqrt(x) > 0 exactly when x > 0: the squared distance stands in for it (a correctly rounded sqrt is 16 instructions)
        const F3 dP = ro - P;
        const float d2 = dot(dP, dP);
        return onlyOne ? d2 : __builtin_sqrtf(d2);
    }
    return -1.0f;
}

// Certain miss of a primitive, decided in WORLD space against its bounding ball for ~20 instructions.  Not an
// approximation of the result but a sufficient condition for the reference's own miss:
//   the image of the object-space ball of radius rho (1/2: the sphere itself; sqrt(3)/2: the cube's corners) lies inside
//   the world ball of radius rho smax around `centre`; if the ray's squared distance from the centre exceeds
//   rho^2 smax^2 (1 + 1e-3) + 1e-4 (smax/smin)^2 |o - c|^2  then in object space  perp^2 - rho^2 > 1e-4 |ro|^2.
//   Sphere: the exact radicand is below -1e-4 |ro|^2, while every rounding in the reference's evaluation (transform,
//   normalize, two dots) perturbs its radicand by less than ~2e-6 |ro|^2 -- `radicand < 0` exits (intersections.h:114).
//   Cube: the line clears the cube by c >= 5e-5 |ro| object units, so the entry parameter of one slab exceeds the exit
//   parameter of another by >= 2c, a relative gap >= 1e-4 of quotients the reference evaluates to ~1e-6 -- its
//   `tmax >= tmin` fails (intersections.h:70).  The margin is 50x either way, and tests sweep 2^28 rays dense in grazes.
// dd = dot(dir, dir) (the world direction is only approximately unit) is hoisted out of the geom loop.
// NaN / inf operands fail the comparison, i.e. fall through to the full test.
template <typename GD>
__device__ __forceinline__ bool certainMiss(const GD &g, F3 org, F3 dir, float dd) {
    // (fused multiply-adds on purpose: this is a sufficient condition with margins of 1e-3 / 1e-4, not one of the reference's
    // operations -- 14 instructions instead of 20, and sphere-heavy scenes run it seventy times per tile)
    const F3 oc = org - f3(g.centre[0], g.centre[1], g.centre[2]);
    const float oo = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x));
    const float od = __builtin_fmaf(oc.z, dir.z, __builtin_fmaf(oc.y, dir.y, oc.x * dir.x));
    return __builtin_fmaf(oo, dd, -(od * od)) > __builtin_fmaf(g.cullK, oo, g.cullR2) * dd;
}

// Certain miss of a SPHERE by the HALF-line a ray is (sphere-heavy scenes, later bounces: k_bounce's packed sweep), against its
// bounding ball with a direction the caller has normalised: dhat = dir * rsq(|dir|^2), unit to 5e-7.
//   d2 = |oc|^2 - max(-oc.dhat, 0)^2 is the squared distance of the centre from the half-line { o + t dhat, t >= 0 } (the origin itself
//   when the centre lies behind it); the certificate is  d2 > rho^2 smax^2 (1 + 1e-3) + K |oc|^2  with K = cullK + kUnitDirSlack:
//   the slack pays for |dhat|^2 = 1 +- 5e-7 (the product form of certainMiss carried |dir|^2 on both sides instead).
//   Centre ahead: the LINE's distance, i.e. certainMiss's own condition and argument (`radicand < 0`, intersections.h:114).
//   Centre behind (-oc.dhat < 0): the origin lies outside the inflated ball, |ro|^2 - 1/4 > 1e-4 |ro|^2 in object space, and the
//   object-space centre is behind the object-space ray as well (an affine map keeps the order of points along a line).  Either the
//   radicand is negative, or both roots are: the larger one, t1 = -b + sqrt(b^2 - (|ro|^2 - 1/4)) with b = ro.rd > 0, is at most
//   -(|ro|^2 - 1/4) / 2b <= -5e-5 |ro|, five hundred times the ~1e-7 |ro| its evaluation can be off by -- `t1 < 0 && t2 < 0`
//   returns the miss (intersections.h:121-123).  Half of the spheres a LINE through a scene meets lie behind the ray's origin, the
//   one a scattered ray leaves among them: the passes that run the full test on a lane's candidates see half as many.
// Three instructions fewer than certainMiss as well (no |dir|^2 on either side, the threshold compared from a scalar register).
// NaN / inf operands fail the comparison, i.e. fall through to the full test.
constexpr float kUnitDirSlack = 2e-6f;
__device__ __forceinline__ float sphereHalfLineExcess(F3 centre, float K, F3 org, F3 dhat) {
    const F3 oc = org - centre;
    const float oo = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x));
    const float od = __builtin_fmaf(oc.z, dhat.z, __builtin_fmaf(oc.y, dhat.y, oc.x * dhat.x));
    const float t = __builtin_fmaxf(-od, 0.0f);
    return __builtin_fmaf(-K, oo, __builtin_fmaf(-t, t, oo));       // certain miss  <=>  this > cullR2
}
__device__ __forceinline__ F3 unitDirection(F3 dir, float dd) { return dir * __builtin_amdgcn_rsqf(dd); }
// The same certificate with the K |oc|^2 term FOLDED into the direction (round 4: 13 instead of 14 instructions per sphere, and the
// sweep runs 64 of them per lane and bounce): with dhat' = s dhat, s^2 >= 1 / (1 - K),
//     |oc|^2 - max(-oc.dhat', 0)^2 > R2 s^2    ==>    |oc|^2 / s^2 - t^2 > R2    ==>    (1 - K) |oc|^2 - t^2 > R2,
// i.e. sphereHalfLineExcess(centre, K, org, dhat) > R2 -- the left side only shrinks when 1 / s^2 is replaced by the larger 1 - K
// (|oc|^2 >= 0).  One s serves every sphere of a scene when it is taken for the LARGEST K among them (a smaller K is implied a
// fortiori); the host rounds s and the thresholds R2 s^2 upwards (pt_init: SphereCull::cullR2, KParams::sphDirScale).
__device__ __forceinline__ F3 unitDirectionScaled(F3 dir, float dd, float s) { return dir * (__builtin_amdgcn_rsqf(dd) * s); }
__device__ __forceinline__ float sphereHalfLineExcessScaled(F3 centre, F3 org, F3 dhatS) {
    const F3 oc = org - centre;
    const float oo = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x));
    const float od = __builtin_fmaf(oc.z, dhatS.z, __builtin_fmaf(oc.y, dhatS.y, oc.x * dhatS.x));
    const float t = __builtin_fmaxf(-od, 0.0f);
    return __builtin_fmaf(-t, t, oo);                               // certain miss  <=>  this > cullR2 s^2
}

// Certain miss of a LARGE cube (a "wall"), decided in world space against its axis-aligned bounding box for ~25
// instructions: the classic slab test on the box INFLATED by delta = 4e-5 of its extent (wall_box in pt_api.hip, double
// precision, rounded outwards), with approximate arithmetic (v_rcp_f32 reciprocals of the direction, supplied by the
// caller).  Like certainMiss it is not an approximation of the result but a sufficient condition for the reference's own
// miss (intersections.h:70: `tmax >= tmin && tmax > 0` fails):
//   every computed t carries a relative error below 3e-7 (one subtraction, one reciprocal, one product), so
//   `tmin - tmax > 1e-5 (|tmin| + |tmax|)` or `tmax < 0` implies that the exact line misses the inflated box, i.e. stays
//   delta away from the cube itself; the reference's evaluation (transform to object space, normalisation, three
//   divisions) moves that decision by less than ~2e-7 of the coordinates involved -- bounded by the caller, which
//   only certifies rays whose origin lies within KParams::wallOMax of the world origin -- a margin of 40x and more
//   (tests/test_gpu_parity.py::test_wall_boxes_never_reject_a_hit sweeps 2^28 rays dense in grazes, 0 violations).
// Infinities (a direction component of 0) and NaNs fail the comparisons, i.e. fall through to the full test.
struct WallBox {
    float lo[3];
    float hi[3];
    float pad[2];
};
static_assert(sizeof(WallBox) == 32, "one s_load_dwordx8");
template <typename WB>
__device__ __forceinline__ bool wallCertainMiss(const WB &w, F3 o, F3 inv) {
    // (subtract, then multiply -- NOT fma(plane, inv, -(o * inv)): with a direction component of 0 the reciprocal is infinite
    // and that form is inf - inf = NaN on EVERY plane of the axis, which min / max silently drop: the sweep below found false
    // certificates for axis-parallel rays at once.  Here such an axis gives +-inf, or NaN only for an origin on the plane.)
    const float tx1 = (w.lo[0] - o.x) * inv.x, tx2 = (w.hi[0] - o.x) * inv.x;
    const float ty1 = (w.lo[1] - o.y) * inv.y, ty2 = (w.hi[1] - o.y) * inv.y;
    const float tz1 = (w.lo[2] - o.z) * inv.z, tz2 = (w.hi[2] - o.z) * inv.z;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx1, tx2), __builtin_fminf(ty1, ty2)), __builtin_fminf(tz1, tz2));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx1, tx2), __builtin_fmaxf(ty1, ty2)), __builtin_fmaxf(tz1, tz2));
    return (tmin - tmax > 1e-5f * (__builtin_fabsf(tmin) + __builtin_fabsf(tmax))) | (tmax < 0.0f);
}

// Certain miss of walls by ONE plane each, for ~5 instructions per wall instead of ~27.  Every wall's inflated box lies inside the
// box `outer` around all of them, a convex set: the part of the ray that can touch any wall is the SEGMENT from its origin to the
// point where it leaves `outer`.  A segment whose two end points lie strictly on the outer side of one plane of a wall's box -- the
// plane that faces the scene's interior, chosen on the host -- lies on that side as a whole: it misses the box.  Like
// wallCertainMiss this is a sufficient condition for the reference's own miss with the inflation (4e-5 of the scene) as its
// margin: the exit point carries a relative error below 4e-7 of (|origin| + the diagonal of `outer`), the thresholds are moved
// towards the interior by five times that (pt_init), and origins beyond KParams::wallOMax get no certificate (caller).  A
// scattered ray starts 1e-3 off the surface it leaves, beyond that wall's plane: the wall it leaves is certified with the others,
// and what remains is, nearly always, exactly the one wall it will hit.
// The planes sit in six SLOTS -- x low side, x high side, y low, y high, z low, z high; at most one wall each (KParams::slotTh,
// slotBit: the wall's bit, 0 for an empty slot) -- so the test is straight-line code: one scalar load, no loop, no branch.
// Returns the slot walls it does NOT certify (their bits); all of them when the ray does not leave `outer` at a positive finite
// parameter (origin outside, NaN).
// (tests/test_gpu_parity.py::test_wall_planes_never_reject_a_hit: 2^28 rays from the surfaces, the interior and the corners.)
template <typename KP>
__device__ __forceinline__ uint32_t wallPlanesPossible(const KP &prm, F3 o, F3 d, F3 inv) {
    // the eighteen wave-uniform constants, fetched together and pinned to scalar registers by ONE empty asm (which also keeps a
    // select of two of them from becoming a vector load from a selected address)
    float lo0 = prm.outerLo[0], lo1 = prm.outerLo[1], lo2 = prm.outerLo[2], hi0 = prm.outerHi[0], hi1 = prm.outerHi[1], hi2 = prm.outerHi[2];
    float th0 = prm.slotTh[0], th1 = prm.slotTh[1], th2 = prm.slotTh[2], th3 = prm.slotTh[3], th4 = prm.slotTh[4], th5 = prm.slotTh[5];
    uint32_t b0 = prm.slotBit[0], b1 = prm.slotBit[1], b2 = prm.slotBit[2], b3 = prm.slotBit[3], b4 = prm.slotBit[4], b5 = prm.slotBit[5];
    asm volatile("" : "+s"(lo0), "+s"(lo1), "+s"(lo2), "+s"(hi0), "+s"(hi1), "+s"(hi2), "+s"(th0), "+s"(th1), "+s"(th2), "+s"(th3), "+s"(th4),
                 "+s"(th5), "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3), "+s"(b4), "+s"(b5));
    // (the bound a component leaves through follows the sign of ITS reciprocal: a zero component gives +-inf, i.e. never)
    const float bx = inv.x > 0.0f ? hi0 : lo0;
    const float by = inv.y > 0.0f ? hi1 : lo1;
    const float bz = inv.z > 0.0f ? hi2 : lo2;
    const float tOut = __builtin_fminf(__builtin_fminf((bx - o.x) * inv.x, (by - o.y) * inv.y), (bz - o.z) * inv.z);
    const F3 e = f3(__builtin_fmaf(d.x, tOut, o.x), __builtin_fmaf(d.y, tOut, o.y), __builtin_fmaf(d.z, tOut, o.z));
    const bool leaves = (tOut > 0.0f) & (tOut < 3.0e38f);          // (NaN fails both)
    uint32_t possible = 0u;
    // slot: the wall lies on the low side of the axis -- certified when both end points are above its plane -- or on the high side
    possible |= ((o.x > th0) & (e.x > th0) & leaves) ? 0u : b0;
    possible |= ((o.x < th1) & (e.x < th1) & leaves) ? 0u : b1;
    possible |= ((o.y > th2) & (e.y > th2) & leaves) ? 0u : b2;
    possible |= ((o.y < th3) & (e.y < th3) & leaves) ? 0u : b3;
    possible |= ((o.z > th4) & (e.z > th4) & leaves) ? 0u : b4;
    possible |= ((o.z < th5) & (e.z < th5) & leaves) ? 0u : b5;
    return possible;
}

// The same certificate for ROTATED walls (round 5: a room none of whose walls is axis-aligned ran at a third of Cornell's rate -- the world
// boxes of tilted slabs fill the room, and no certificate was ever issued).  A cube lies on ONE side of the plane of each of its faces,
// whichever way that face looks: with n the unit normal of the face that looks at the scene's interior, pointing there, and th the largest
// n . corner of the inflated cube moved further that way by the slack, a segment whose end points both have n . x > th misses the cube.
// The segment is the one of wallPlanesPossible -- the ray from its origin to where it leaves `outer` --, the margins are its margins
// plus the rounding of the two dot products (host: choose_walls).  planeN[w] = {n, th} of wall first + w.
// (tests/test_gpu_parity.py::test_wall_planes_never_reject_a_hit sweeps rotated rooms too.)
template <typename KP>
__device__ __forceinline__ uint32_t wallPlanesOriented(const KP &prm, F3 o, F3 d, F3 inv, int first, int count) {
    const float bx = inv.x > 0.0f ? prm.outerHi[0] : prm.outerLo[0];
    const float by = inv.y > 0.0f ? prm.outerHi[1] : prm.outerLo[1];
    const float bz = inv.z > 0.0f ? prm.outerHi[2] : prm.outerLo[2];
    float tOut = __builtin_fminf(__builtin_fminf((bx - o.x) * inv.x, (by - o.y) * inv.y), (bz - o.z) * inv.z);
    // ... and where it leaves the half-spaces n . x >= far of the rotated walls (each holds every wall's cube: the convex set the segment
    // lives in is `outer` cut by them -- the world box of a room of tilted slabs reaches far behind its walls, where their planes cross)
    for (int w = 0; w < count; ++w) {
        const float nx = prm.planeN[w][0], ny = prm.planeN[w][1], nz = prm.planeN[w][2], far = prm.planeN[w][4];
        const float so = __builtin_fmaf(nz, o.z, __builtin_fmaf(ny, o.y, nx * o.x)), sd = __builtin_fmaf(nz, d.z, __builtin_fmaf(ny, d.y, nx * d.x));
        const float t = (far - so) * __builtin_amdgcn_rcpf(sd);
        tOut = ((sd < 0.0f) & (t > 0.0f)) ? __builtin_fminf(tOut, t) : tOut;     // (an origin beyond `far`, a NaN: this half-space cuts nothing)
    }
    const F3 e = f3(__builtin_fmaf(d.x, tOut, o.x), __builtin_fmaf(d.y, tOut, o.y), __builtin_fmaf(d.z, tOut, o.z));
    const bool leaves = (tOut > 0.0f) & (tOut < 3.0e38f);          // (NaN fails both)
    uint32_t possible = 0u;
    for (int w = 0; w < count; ++w) {
        const float nx = prm.planeN[w][0], ny = prm.planeN[w][1], nz = prm.planeN[w][2], th = prm.planeN[w][3];
        const float so = __builtin_fmaf(nz, o.z, __builtin_fmaf(ny, o.y, nx * o.x)), se = __builtin_fmaf(nz, e.z, __builtin_fmaf(ny, e.y, nx * e.x));
        possible |= ((so > th) & (se > th) & leaves) ? 0u : (1u << (first + w));
    }
    return possible;
}

// src/intersections.h:101-143 (pow(radius, 2) == 0.25f in the float overload nvcc selects).
// `inv`, `invZ`, `xf`: rows 0-2 of inverseTransform (and its w = 0 products, GeomDev::invZ) / transform as mulMV expects them -- SGPR operands when the sphere is
// wave-uniform (GeomDev through the scalar path), registers when every lane tests its own sphere (k_bounce<., MANY>).
// `camObj`: the precomputed object-space origin of a camera ray, or nullptr.
template <bool CAM_ORIGIN, typename P1, typename P2, typename P3, typename P4>
__device__ __forceinline__ float sphereIntersectionTestLazy(P1 inv, P2 invZ, P3 loadXf, P4 camObj,
                                                            F3 ro_w, F3 rd_w, F3 &P, F3 &nsrc, bool &outside) {
    F3 ro = CAM_ORIGIN ? f3(camObj[0], camObj[1], camObj[2]) : mulMV(inv, ro_w, 1.0f);
    F3 rd = normalize(mulMV0(inv, invZ, rd_w));
    float vDotDirection = dot(ro, rd);
    float radicand = vDotDirection * vDotDirection - (dot(ro, ro) - 0.25f);
    if (radicand < 0) return -1.0f;
    probe(5);
    float squareRoot = __builtin_sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;
    float t;
    if (t1 < 0 && t2 < 0) {
        return -1.0f;
    } else if (t1 > 0 && t2 > 0) {
        t = t2 < t1 ? t2 : t1;  // min(t1, t2)
        outside = true;
    } else {
        t = t1 < t2 ? t2 : t1;  // max(t1, t2)
        outside = false;
    }
    probe(6);
    F3 obj = getPointOnRay(ro, rd, t);
    float xf[12];
    loadXf(xf);      // (the transform's rows are fetched HERE, by the lanes that hit: sphere-heavy scenes read them per lane from LDS / global memory)
    P = mulMV(xf, obj, 1.0f);
    nsrc = obj;      // normal = +-normalize(invTranspose * (obj, 0)): hitNormal(), evaluated for the nearest hit only
    return length(ro_w - P);
}
template <bool CAM_ORIGIN = false, typename P1, typename P2, typename P3, typename P4>
__device__ __forceinline__ float sphereIntersectionTestM(P1 inv, P2 invZ, P3 xf, P4 camObj,
                                                         F3 ro_w, F3 rd_w, F3 &P, F3 &nsrc, bool &outside) {
    return sphereIntersectionTestLazy<CAM_ORIGIN>(inv, invZ, [&](float (&x)[12]) {
#pragma unroll
        for (int i = 0; i < 12; ++i) x[i] = xf[i];
    }, camObj, ro_w, rd_w, P, nsrc, outside);
}
template <bool CAM_ORIGIN = false, typename GD>
__device__ __forceinline__ float sphereIntersectionTest(const GD &g, F3 ro_w, F3 rd_w, F3 &P, F3 &nsrc,
                                                        bool &outside) {
    probe(4);
    return sphereIntersectionTestM<CAM_ORIGIN>(g.inv, g.invZ, g.xf, g.camObj, ro_w, rd_w, P, nsrc, outside);
}

// ---- triangle meshes (README.md:112-116, 236: object type "mesh"; build-defined, the reference holds no mesh code) ----
// Semantics (include/pt_amd.h; the CPU oracle states them as a loop over every triangle): the ray goes to object space like the sphere test's; a triangle
// is tested -- glm::intersectRayTriangle (glm/gtx/intersect.inl:36-72) made two-sided -- only when the ray passes the slab
// test of the triangle's own bounding box (inflated by a per-mesh margin), and its hit counts only at or beyond that box's
// entry parameter; nearest = smallest object-space t, ties to the lower triangle index.  Every part of that rule is local
// to (ray, triangle), and fp32 subtraction, multiplication by a common factor and comparison are monotone: a ray that
// passes a box's test passes the test of every box that contains it.  So the bounding-volume hierarchy below
// (node boxes = exact unions of their triangles' boxes) visits every triangle the brute-force rule accepts, and a node
// whose entry parameter lies beyond the best hit so far cannot hold a better one: bit-identical results in any order.
//
// ONE array per scene, addressed in units of 16 bytes by a 31-bit index (`ref`, bit 31 = the record is a triangle), two kinds of record:
//   inner node (two units):  the boxes of BOTH its children and their refs -- a visit decides about two subtrees.  The boxes are
//                            stored as the planes a ray enters / leaves through (below) in HALF precision, rounded outwards: an inner
//                            box only has to CONTAIN what lies below it (a ray that passes a box passes every box around it), and a
//                            visit is what the walk pays for: the texture addresser is busy 16 cycles per 16-byte-per-lane load,
//                            however few lanes take part, and it is what bounds a mesh scene (profiles/r03_mesh_walk_experiments.txt:
//                            TA_BUSY 80 % of a launch);
//   triangle (three units):  its vertices and the mesh's box margin, 40 bytes of the 48 fetched (16 + 16 + 8): the edges e1 = v1 - v0,
//                            e2 = v2 - v0 and the triangle's own inflated box -- the box the semantics test, min / max of the
//                            vertices -+ the margin -- are the same fp32 operations wherever they are evaluated, so they are
//                            evaluated here, for 18 instructions, instead of being fetched, for 24 more bytes per lane.
// Rounds 1-2 walked 32-byte full-precision nodes in depth-first order with skip links, no stack: every child of a visited node, hit
// or missed, was a visit of its own (2 I + 1 for a ray that passes I inner nodes), and a leaf two fetches (its node, then its triangle).
// Here the far child of a node whose children both pass waits on a short per-lane stack in LDS (slots kBlock words apart: lanes never
// share a bank; its depth is the scene's, computed by pt_init), and a ray keeps the texture addresser busy for 32 I + 40 T cycles
// instead of 64 I + 32 + 80 T.
// Inner nodes are stored once per sign octant of the ray direction, nearer child first (pt_mesh.h): a ray walks its octant's copy
// front to back, so the first hits prune most of what lies behind them; the triangles, in file order, once.
struct MeshUnit {
    uint32_t w[4];
};
// inner node: w[0..2] the near child's planes as six halves -- entry x, y, z, exit x, y, z --, w[3] its ref, w[4..6] / w[7] the far child's.
//             The copy of an octant knows the sign of every direction component, hence which of a box's two planes per axis a ray of
//             that octant meets first (lo where the component is positive, hi where it is negative): min(fma(lo, inv, rc),
//             fma(hi, inv, rc)) IS the entry plane's parameter (fma is monotone in its first operand) -- the hierarchy stores the
//             planes in that order and the six min / max per box of the slab test are gone.  lo is rounded down, hi up.
// triangle:   floats 0..2 v0, 3..5 v1, 6..8 v2, 9 the mesh's margin (pt_mesh.h: meshMargin); word 10 = 1 + the face's own material (0: the
//             object's), word 11 = ref of the triangle's vertex normals -- three units: n0, n1, n2 -- or 0: flat shading (PtMesh::normals /
//             materials; read for the winning triangle of a walk only)
static_assert(sizeof(MeshUnit) == 16, "one float4 load");
constexpr int kMeshNodeUnits = 2, kMeshTriUnits = 3;
constexpr uint32_t kMeshEnd = 0xffffffffu;                 // GeomDev::meshRoot of a primitive that is not a mesh
constexpr uint32_t kMeshLeaf = 0x80000000u;
constexpr float kMeshEps = 1.1920928955078125e-07f;        // std::numeric_limits<float>::epsilon(), intersect.inl:50
constexpr float kMeshUp = 1.00001f, kMeshDn = 0.99999f;    // relative slack of the slab comparison

__device__ __forceinline__ float guardedReciprocal(float d) {
    const float g = __builtin_fabsf(d) < 1e-30f ? __builtin_copysignf(1e-30f, d) : d;
    return 1.0f / g;
}

// The triangle test; returns t (>= 0) or -1, `front` = the counter-clockwise side faces the ray.
__device__ __forceinline__ bool meshTriangle(F3 o, F3 d, F3 v0, F3 e1, F3 e2, float &t, bool &front) {
    const F3 p = cross(d, e2);
    const float a = dot(e1, p);
    if (__builtin_fabsf(a) < kMeshEps) return false;
    const float f = 1.0f / a;
    const F3 s = o - v0;
    const float u = f * dot(s, p);
    if (u < 0.0f) return false;
    if (u > 1.0f) return false;
    const F3 q = cross(s, e1);
    const float v = f * dot(d, q);
    if (v < 0.0f) return false;
    if (v + u > 1.0f) return false;
    t = f * dot(e2, q);
    front = a > 0.0f;
    return t >= 0.0f;
}

// the slab test of one box, the semantics' own (fma form, see below): entry parameter scaled down, pass = not certainly missed
// and not certainly behind the best hit so far
__device__ __forceinline__ bool meshBoxPass(F3 lo, F3 hi, F3 inv, F3 rc, bool haveBest, float tbest, float &tmin) {
    const float ax = __builtin_fmaf(lo.x, inv.x, rc.x), bx = __builtin_fmaf(hi.x, inv.x, rc.x);
    const float ay = __builtin_fmaf(lo.y, inv.y, rc.y), by = __builtin_fmaf(hi.y, inv.y, rc.y);
    const float az = __builtin_fmaf(lo.z, inv.z, rc.z), bz = __builtin_fmaf(hi.z, inv.z, rc.z);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    tmin = tn * kMeshDn;
    // (a box whose entry parameter lies beyond the best hit cannot improve on it: accepted hits have t >= tmin)
    return (tf * kMeshUp >= tmin) & (tf >= 0.0f) & (!haveBest | !(tmin > tbest));
}

// ... and of a child's box inside an inner node of the ray's octant copy: entry / exit planes as stored (see MeshUnit), three words of
// two halves each.  The parameters meshBoxPass would compute from the rounded (lo, hi); an operand that is NaN (inf - inf: coordinates
// beyond 1e8) drops out of max3 / min3, i.e. leaves the test more permissive, which an inner node may always be.
__device__ __forceinline__ float halfLo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float halfHi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
__device__ __forceinline__ bool meshPlanesPass(uint32_t w0, uint32_t w1, uint32_t w2, F3 inv, F3 rc, bool haveBest, float tbest) {
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(halfLo(w0), inv.x, rc.x), __builtin_fmaf(halfHi(w0), inv.y, rc.y)),
                                     __builtin_fmaf(halfLo(w1), inv.z, rc.z));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(halfHi(w1), inv.x, rc.x), __builtin_fmaf(halfLo(w2), inv.y, rc.y)),
                                     __builtin_fmaf(halfHi(w2), inv.z, rc.z));
    const float tmin = tn * kMeshDn;
    return (tf * kMeshUp >= tmin) & (tf >= 0.0f) & (!haveBest | !(tmin > tbest));
}

// One ray against one mesh.  `recs`: the scene's record array (per-lane loads: every lane walks its own way through the
// hierarchy), `root`: the ref of the mesh's root node in the copy of octant 0, `stride`: inner nodes per copy.  `stack`: this lane's
// first stack slot (LDS), the next ones STRIDE words apart.  Outputs as the sphere test's (P world point, nsrc the
// object-space vector the normal is made of -- here the unit face normal -- and `outside` = front side).
// (NaN operands: every triangle test fails whatever the slab tests say -- a, or s, is NaN -- like in the oracle's loop.)
// `faceMat`: 1 + the scene material of the triangle that was hit when it has one of its own (`usemtl`), else 0.
template <bool CAM_ORIGIN = false, int STRIDE = 256, typename GD>
__device__ __forceinline__ float meshIntersectionTest(const GD &g, const float4 *recs, uint32_t root, uint32_t stride, uint32_t *stack,
                                                      F3 ro_w, F3 rd_w, F3 &P, F3 &nsrc, bool &outside, int &faceMat) {
    const F3 ro = CAM_ORIGIN ? f3(g.camObj[0], g.camObj[1], g.camObj[2]) : mulMV(g.inv, ro_w, 1.0f);
    const F3 rd = normalize(mulMV0(g.inv, g.invZ, rd_w));
    const F3 inv = f3(guardedReciprocal(rd.x), guardedReciprocal(rd.y), guardedReciprocal(rd.z));
    // the parameter of a box plane x = lo is ONE fused multiply-add, fma(lo, inv, rc) with rc = -(ro * inv): a single rounding of
    // lo * inv - fl(ro * inv), monotone in lo like the subtract-then-multiply form, at a third fewer instructions per box
    const F3 rc = f3(-(ro.x * inv.x), -(ro.y * inv.y), -(ro.z * inv.z));
    int best = -1;                                          // unit index of the best triangle so far (file order within a mesh)
    float tbest = 0.0f;
    uint32_t bestFront = 0u;
    // (the octant is read off the SIGN BITS of the reciprocals, the operands of the plane parameters: a component of -0 counts as negative)
    const uint32_t octant = (__float_as_uint(inv.x) >> 31) | ((__float_as_uint(inv.y) >> 31) << 1) | ((__float_as_uint(inv.z) >> 31) << 2);
    uint32_t ref = root + octant * stride;
    uint32_t *sp = stack;
    constexpr uint32_t kDone = 0xffffffffu;                 // (reads as a triangle's ref: the loop over inner nodes stops on it)
    auto pop = [&]() -> uint32_t {
        if (sp == stack) return kDone;
        sp -= STRIDE;
        return *sp;
    };
    // Inner nodes until the lane holds a triangle (or nothing), THEN the triangle: the lanes of a wave leave the inner loop together, so
    // the triangle code -- twice an inner step's instructions -- runs once for all the triangles they found instead of at every step
    // at which some lane happens to hold one (one loop over both kinds of record ran both halves at four steps out of five).
    for (;;) {
        while (!(ref & kMeshLeaf)) {
            const float4 *r = recs + (size_t)ref;
            const float4 q0 = r[0], q1 = r[1];
            const bool passN = meshPlanesPass(__float_as_uint(q0.x), __float_as_uint(q0.y), __float_as_uint(q0.z), inv, rc, best >= 0, tbest);
            const bool passF = meshPlanesPass(__float_as_uint(q1.x), __float_as_uint(q1.y), __float_as_uint(q1.z), inv, rc, best >= 0, tbest);
            const uint32_t refN = __float_as_uint(q0.w), refF = __float_as_uint(q1.w);
            if (passN & passF) {
                *sp = refF;
                sp += STRIDE;
            }
            ref = passN ? refN : (passF ? refF : pop());
        }
        if (ref == kDone) break;
        {
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
            if (meshBoxPass(lo, hi, inv, rc, best >= 0, tbest, tmin)) {
                float t;
                bool front;
                if (meshTriangle(ro, rd, v0, v1 - v0, v2 - v0, t, front)) {
                    const int tri = (int)(ref & ~kMeshLeaf);
                    if ((t >= tmin) & ((best < 0) | (t < tbest) | ((t == tbest) & (tri < best)))) {
                        best = tri;
                        tbest = t;
                        bestFront = front ? 1u : 0u;
                    }
                }
            }
        }
        ref = pop();
    }
    if (best < 0) return -1.0f;
    const float4 a = recs[(size_t)best], b = recs[(size_t)best + 1], c = recs[(size_t)best + 2];
    const F3 w0 = f3(a.x, a.y, a.z);
    const F3 e1 = f3(a.w, b.x, b.y) - w0, e2 = f3(b.z, b.w, c.x) - w0;
    const F3 nface = cross(e1, e2);
    F3 nobj = normalize(nface);
    faceMat = (int)__float_as_uint(c.z);
    const uint32_t nref = __float_as_uint(c.w);
    if (nref != 0u) {
        // vertex normals: the blend n0 (1 - u - v) + n1 u + n2 v with the hit's own (u, v) -- the triangle test's, evaluated once more
        // for the winner (same operands, same bits) --, turned to the face normal's side; a blend of length zero keeps the face normal
        const F3 p = cross(rd, e2);
        const float f = 1.0f / dot(e1, p);
        const F3 sv = ro - w0;
        const float u = f * dot(sv, p);
        const F3 q = cross(sv, e1);
        const float v = f * dot(rd, q);
        const float4 n0 = recs[(size_t)nref], n1 = recs[(size_t)nref + 1], n2 = recs[(size_t)nref + 2];
        const float w = (1.0f - u) - v;
        F3 ns = (f3(n0.x, n0.y, n0.z) * w + f3(n0.w, n1.x, n1.y) * u) + f3(n1.z, n1.w, n2.x) * v;
        if (dot(ns, nface) < 0.0f) ns = -ns;
        if (dot(ns, ns) > 0.0f) nobj = normalize(ns);
    }
    const F3 obj = getPointOnRay(ro, rd, tbest);
    P = mulMV(g.xf, obj, 1.0f);
    nsrc = nobj;         // normal = +-normalize(invTranspose * (nobj, 0)): hitNormal(), evaluated for the nearest hit only
    outside = bestFront != 0u;
    return length(ro_w - P);
}

// What a walk's WINNER -- triangle record `best` -- means for the ray: the outputs of meshIntersectionTest, from the triangle alone.  The
// render kernels' walks run ahead of the bounce, in a kernel of their own (pt_mesh_walk.h), and leave the winner's unit per path; the bounce
// evaluates it here: t, and (u, v) for the vertex normals, are the triangle test's, evaluated once more (same operands, same bits).
template <bool CAM_ORIGIN = false, typename GD>
__device__ __forceinline__ float meshWinner(const GD &g, const float4 *recs, uint32_t best, bool front, F3 ro_w, F3 rd_w, F3 &P, F3 &nsrc,
                                            bool &outside, int &faceMat) {
    const F3 ro = CAM_ORIGIN ? f3(g.camObj[0], g.camObj[1], g.camObj[2]) : mulMV(g.inv, ro_w, 1.0f);
    const F3 rd = normalize(mulMV0(g.inv, g.invZ, rd_w));
    const float4 a = recs[(size_t)best], b = recs[(size_t)best + 1], c = recs[(size_t)best + 2];
    const F3 w0 = f3(a.x, a.y, a.z);
    const F3 e1 = f3(a.w, b.x, b.y) - w0, e2 = f3(b.z, b.w, c.x) - w0;
    const F3 nface = cross(e1, e2);
    F3 nobj = normalize(nface);
    faceMat = (int)__float_as_uint(c.z);
    const uint32_t nref = __float_as_uint(c.w);
    const F3 p = cross(rd, e2);
    const float f = 1.0f / dot(e1, p);
    const F3 sv = ro - w0;
    const F3 q = cross(sv, e1);
    const float tbest = f * dot(e2, q);
    if (nref != 0u) {
        const float u = f * dot(sv, p);
        const float v = f * dot(rd, q);
        const float4 n0 = recs[(size_t)nref], n1 = recs[(size_t)nref + 1], n2 = recs[(size_t)nref + 2];
        const float w = (1.0f - u) - v;
        F3 ns = (f3(n0.x, n0.y, n0.z) * w + f3(n0.w, n1.x, n1.y) * u) + f3(n1.z, n1.w, n2.x) * v;
        if (dot(ns, nface) < 0.0f) ns = -ns;
        if (dot(ns, ns) > 0.0f) nobj = normalize(ns);
    }
    const F3 obj = getPointOnRay(ro, rd, tbest);
    P = mulMV(g.xf, obj, 1.0f);
    nsrc = nobj;
    outside = front;
    return length(ro_w - P);
}

template <bool CAM_ORIGIN = false, int STRIDE = 256, typename GD>
__device__ __forceinline__ float meshIntersectionTest(const GD &g, const float4 *recs, uint32_t root, uint32_t stride, uint32_t *stack,
                                                      F3 ro_w, F3 rd_w, F3 &P, F3 &nsrc, bool &outside) {
    int faceMat = 0;
    return meshIntersectionTest<CAM_ORIGIN, STRIDE>(g, recs, root, stride, stack, ro_w, rd_w, P, nsrc, outside, faceMat);
}

// The surface normal of a hit, from what the two tests leave in `nsrc` (src/intersections.h:85 and :137-140).
// The reference computes it for every primitive a ray hits; only the nearest hit's normal is ever used, so the
// kernels evaluate it once, after the nearest-hit loop -- same inputs, same operations, same bits.
// Sphere: `m` = rows 0-2 of invTranspose as mulMV expects them, nsrc = the object-space hit point.
// Cube: nsrc = +-e_axis, so the normal is one of six vectors per cube, which pack_geom (pt_api.hip) evaluates with the
// very operations of normalize(mulMV(transform, nsrc, 0)): a table lookup instead of 52 instructions per hit.  The
// table also holds the hemisphere sampler's two tangent directions of each face (functions of the normal alone).
__device__ __forceinline__ F3 hitNormalSphere(const float *invT, F3 nsrc, bool outside) {
    const F3 n = normalize(mulMV(invT, nsrc, 0.0f));
    return outside ? n : -n;
}
// face index of a cube hit (axis * 2 + (sign > 0)), which boxIntersectionTest leaves as bits in nsrc.x; ok = false (face -1) for a hit
// without an exit slab -- a ray of NaNs -- whose slab normal is the zero vector: the reference then normalises it, i.e. every derived vector is NaN
__device__ __forceinline__ int cubeFace(F3 nsrc, bool &ok) {
    const int face = __float_as_int(nsrc.x);
    ok = face >= 0;
    return face;
}
// entry `which` (0 normal, 1 and 2 the sampler's tangents) of a face of the table (selects: adding 0 would turn -0 into +0)
__device__ __forceinline__ F3 cubeFrameVector(const float *cubeFrame, int face, int which, bool ok) {
    const float *n = cubeFrame + 9 * face + 3 * which;
    const float nan = __builtin_nanf("");
    return f3(ok ? n[0] : nan, ok ? n[1] : nan, ok ? n[2] : nan);
}
__device__ __forceinline__ F3 hitNormal(const GeomDev &g, F3 nsrc, bool outside) {
    if (g.type == 0) return hitNormalSphere(g.invT, nsrc, outside);
    bool ok;
    const int face = cubeFace(nsrc, ok);
    return cubeFrameVector(g.cubeFrame, face, 0, ok);
}

// Per-geom record staged in LDS for the per-lane lookups that follow the nearest-hit loop: the sphere's normal matrix
// (12 floats), the material index, the type and the cube's six face frames.  Lanes of a wave index different geoms, so
// the row stride is 76 words: consecutive rows start 12 banks apart and 8 different rows are conflict-free
// (64-B rows collided two ways: SQ_LDS_BANK_CONFLICT 20 % of LDS cycles).
// It opens with what EVERY hit needs -- the primitive's type and the hot fields of ITS material, copied in -- as two 16-byte reads
// issued together: one LDS round trip where the chain hit -> record -> material index -> material table took two (round 3: a
// wave's dependent waits on the scalar cache and on LDS, not its instruction count, turned out to bound the later bounces; the
// fields of the rarer branches -- specular colour, index of refraction, Schlick's r0, the lobe exponent -- stay in the table).
struct GeomHitDev {
    int   type;
    float emittance, hasReflective, hasRefractive;
    float color[3];
    int   material;
    float nm[12];
    float cubeFrame[54];
    int   pad[2];
};
static_assert(sizeof(GeomHitDev) == 304 && offsetof(GeomHitDev, nm) == 32, "GeomHitDev is 19 x 16 B, its hot header 2 x 16 B");
// Sphere-heavy scenes: 304 B for each of seventy primitives (21 KB, together with the lanes' sphere lists and matrices 33 KB of
// LDS per workgroup) admitted only FOUR workgroups per CU where the registers allow seven (residency census,
// profiles/census.py).  There the record is 68 B -- 17 words: an odd stride, so lanes that index different primitives do not
// collide -- and the face frames of the (few) cubes sit in a table of their own, addressed by `frame`.
struct GeomHitSmall {
    float nm[12];
    int   material;
    int   type;
    int   frame;         // cubes: row of the frame table (54 floats each); others: 0
    int   pad[2];
};
static_assert(sizeof(GeomHitSmall) == 68, "17 words");

// src/interactions.h:10-42, in three parts: the tangent frame is a function of the normal alone (for a cube face it
// comes from GeomDev::cubeFrame), the draws are two random numbers, the combination is the reference's last line.
__device__ __forceinline__ void hemisphereFrame(F3 normal, F3 &p1, F3 &p2) {
    F3 notNormal;
    if (__builtin_fabsf(normal.x) < kSqrtOneThird) {
        notNormal = f3(1, 0, 0);
    } else if (__builtin_fabsf(normal.y) < kSqrtOneThird) {
        notNormal = f3(0, 1, 0);
    } else {
        notNormal = f3(0, 0, 1);
    }
    p1 = normalize(cross(normal, notNormal));
    p2 = normalize(cross(normal, p1));
}
// the sample's coordinates in that frame: `up` along the normal, (c, s) * over across it (two draws: up, then the angle)
__device__ __forceinline__ void hemisphereDraws(Rng &rng, float &up, float &cOver, float &sOver) {
    // u01 is 0 or at least 2^-31, and 1 - up*up is 0 or at least 2^-24 (up*up <= 1 is a float): both operands are
    // inside sqrtUnscaled's range by construction
    up = sqrtUnscaled(u01(rng));
    const float over = sqrtUnscaled(1 - up * up);
    const float around = u01(rng) * kTwoPi;
    float s, c;
    sincosPoly(around, s, c);
    cOver = c * over;
    sOver = s * over;
}
__device__ __forceinline__ F3 hemisphereCombine(F3 normal, F3 p1, F3 p2, float up, float cOver, float sOver) {
    return (normal * up + p1 * cOver) + p2 * sOver;
}
// Imperfect specular: a direction in the Phong lobe (1 / invExp1 - 1 = exponent) around the mirror direction R, in the tangent
// frame the hemisphere sampler builds around a vector; a sample below the surface falls back to R (the oracle's
// random_direction_in_specular_lobe, operation for operation).
__device__ __forceinline__ F3 specularLobeDirection(F3 R, F3 normal, float invExp1, Rng &rng) {
    const float cosT = powPoly(u01(rng), invExp1);
    const float sinT = __builtin_sqrtf(1 - cosT * cosT);
    const float around = u01(rng) * kTwoPi;
    F3 p1, p2;
    hemisphereFrame(R, p1, p2);
    float s, c;
    sincosPoly(around, s, c);
    const F3 d = hemisphereCombine(R, p1, p2, cosT, c * sinT, s * sinT);
    return dot(d, normal) > 0.0f ? d : R;
}
__device__ __forceinline__ F3 calculateRandomDirectionInHemisphere(F3 normal, Rng &rng) {
    float up, cOver, sOver;
    hemisphereDraws(rng, up, cOver, sOver);
    F3 p1, p2;
    hemisphereFrame(normal, p1, p2);
    return hemisphereCombine(normal, p1, p2, up, cOver, sOver);
}

}  // namespace ptd
