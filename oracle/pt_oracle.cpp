/*
 * pt_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See pt_oracle.h.
 *
 * Plain single-thread C++.  Build: g++ -O2 -ffp-contract=off (no FMA contraction, no
 * fast-math) so that every fp32 operation below is one IEEE-754 round-to-nearest op,
 * in the order the reference (glm 0.9.6.3 + src/intersections.h + src/interactions.h)
 * performs it.  Citations are relative to /root/reference.
 */
#include "pt_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

/* src/utilities.h:12-15 */
const float kPI = 3.1415926535897932384626422832795028841971f;
const float kTWO_PI = 6.2831853071795864769252867665590057683943f;
const float kSQRT_OF_ONE_THIRD = 0.5773502691896257645091487805019574556476f;

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};

inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V4 v4(float x, float y, float z, float w) { V4 r = {x, y, z, w}; return r; }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 muls(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V4 add4(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline V4 sub4(V4 a, V4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline V4 mul4(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline V4 muls4(V4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }

/* glm/detail/func_geometric.inl:64-83 : tmp = x*y ; tmp.x + tmp.y + tmp.z */
inline float dot3(V3 a, V3 b) {
    V3 t = mul(a, b);
    return t.x + t.y + t.z;
}
/* glm/detail/func_geometric.inl:134-143 */
inline V3 cross3(V3 x, V3 y) {
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm/detail/func_exponential.inl:150-153 : 1 / sqrt(x) */
inline float inversesqrt1(float x) { return 1.0f / std::sqrt(x); }
/* glm/detail/func_geometric.inl:154-159 : x * inversesqrt(dot(x,x)) */
inline V3 normalize3(V3 a) { return muls(a, inversesqrt1(dot3(a, a))); }
/* glm/detail/func_geometric.inl:95-100 */
inline float length3(V3 a) { return std::sqrt(dot3(a, a)); }
/* glm/detail/func_geometric.inl:176-179 : I - N * dot(N, I) * 2 */
inline V3 reflect3(V3 I, V3 N) { return sub(I, muls(muls(N, dot3(N, I)), 2.0f)); }
/* glm/detail/func_geometric.inl:193-200 (vector overload) */
inline V3 refract3(V3 I, V3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    V3 r = sub(muls(I, eta), muls(N, eta * d + std::sqrt(k)));
    return muls(r, (float)(k >= 0.0f));
}

/* column-major mat4: m[col*4 + row] (glm tmat4x4: m[col][row]) */
struct M4 {
    V4 c[4];
};
inline M4 m4_from(const float *p) {
    M4 m;
    for (int i = 0; i < 4; ++i) m.c[i] = v4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
    return m;
}
inline void m4_to(const M4 &m, float *p) {
    for (int i = 0; i < 4; ++i) {
        p[4 * i] = m.c[i].x; p[4 * i + 1] = m.c[i].y; p[4 * i + 2] = m.c[i].z; p[4 * i + 3] = m.c[i].w;
    }
}
inline M4 m4_identity() {
    M4 m;
    m.c[0] = v4(1, 0, 0, 0); m.c[1] = v4(0, 1, 0, 0); m.c[2] = v4(0, 0, 1, 0); m.c[3] = v4(0, 0, 0, 1);
    return m;
}
/* glm/detail/type_mat4x4.inl:617-628 : (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
inline V4 m4_mulv(const M4 &m, V4 v) {
    V4 mul0 = muls4(m.c[0], v.x);
    V4 mul1 = muls4(m.c[1], v.y);
    V4 add0 = add4(mul0, mul1);
    V4 mul2 = muls4(m.c[2], v.z);
    V4 mul3 = muls4(m.c[3], v.w);
    V4 add1 = add4(mul2, mul3);
    return add4(add0, add1);
}
/* src/intersections.h:33-35 */
inline V3 multiplyMV(const M4 &m, V4 v) {
    V4 r = m4_mulv(m, v);
    return v3(r.x, r.y, r.z);
}
/* glm/detail/type_mat4x4.inl:686-704 : ((A0*b0 + A1*b1) + A2*b2) + A3*b3 */
inline M4 m4_mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int j = 0; j < 4; ++j) {
        V4 bj = b.c[j];
        r.c[j] = add4(add4(add4(muls4(a.c[0], bj.x), muls4(a.c[1], bj.y)), muls4(a.c[2], bj.z)),
                      muls4(a.c[3], bj.w));
    }
    return r;
}
/* glm/gtc/matrix_transform.inl:40-50 */
inline M4 m4_translate(const M4 &m, V3 v) {
    M4 r = m;
    r.c[3] = add4(add4(add4(muls4(m.c[0], v.x), muls4(m.c[1], v.y)), muls4(m.c[2], v.z)), m.c[3]);
    return r;
}
/* glm/gtc/matrix_transform.inl:52-86 */
inline M4 m4_rotate(const M4 &m, float angle, V3 v) {
    const float a = angle;
    const float c = std::cos(a);
    const float s = std::sin(a);
    V3 axis = normalize3(v);
    V3 temp = v3((1.0f - c) * axis.x, (1.0f - c) * axis.y, (1.0f - c) * axis.z);
    float R00 = c + temp.x * axis.x;
    float R01 = 0 + temp.x * axis.y + s * axis.z;
    float R02 = 0 + temp.x * axis.z - s * axis.y;
    float R10 = 0 + temp.y * axis.x - s * axis.z;
    float R11 = c + temp.y * axis.y;
    float R12 = 0 + temp.y * axis.z + s * axis.x;
    float R20 = 0 + temp.z * axis.x + s * axis.y;
    float R21 = 0 + temp.z * axis.y - s * axis.x;
    float R22 = c + temp.z * axis.z;
    M4 r;
    r.c[0] = add4(add4(muls4(m.c[0], R00), muls4(m.c[1], R01)), muls4(m.c[2], R02));
    r.c[1] = add4(add4(muls4(m.c[0], R10), muls4(m.c[1], R11)), muls4(m.c[2], R12));
    r.c[2] = add4(add4(muls4(m.c[0], R20), muls4(m.c[1], R21)), muls4(m.c[2], R22));
    r.c[3] = m.c[3];
    return r;
}
/* glm/gtc/matrix_transform.inl:122-134 */
inline M4 m4_scale(const M4 &m, V3 v) {
    M4 r;
    r.c[0] = muls4(m.c[0], v.x);
    r.c[1] = muls4(m.c[1], v.y);
    r.c[2] = muls4(m.c[2], v.z);
    r.c[3] = m.c[3];
    return r;
}
inline float E(const M4 &m, int col, int row) {
    const V4 &c = m.c[col];
    return row == 0 ? c.x : row == 1 ? c.y : row == 2 ? c.z : c.w;
}
/* glm/detail/type_mat4x4.inl:37-91 */
M4 m4_inverse(const M4 &m) {
#define M(c, r) E(m, c, r)
    float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    V4 Fac0 = v4(Coef00, Coef00, Coef02, Coef03);
    V4 Fac1 = v4(Coef04, Coef04, Coef06, Coef07);
    V4 Fac2 = v4(Coef08, Coef08, Coef10, Coef11);
    V4 Fac3 = v4(Coef12, Coef12, Coef14, Coef15);
    V4 Fac4 = v4(Coef16, Coef16, Coef18, Coef19);
    V4 Fac5 = v4(Coef20, Coef20, Coef22, Coef23);
    V4 Vec0 = v4(M(1, 0), M(0, 0), M(0, 0), M(0, 0));
    V4 Vec1 = v4(M(1, 1), M(0, 1), M(0, 1), M(0, 1));
    V4 Vec2 = v4(M(1, 2), M(0, 2), M(0, 2), M(0, 2));
    V4 Vec3 = v4(M(1, 3), M(0, 3), M(0, 3), M(0, 3));
    V4 Inv0 = add4(sub4(mul4(Vec1, Fac0), mul4(Vec2, Fac1)), mul4(Vec3, Fac2));
    V4 Inv1 = add4(sub4(mul4(Vec0, Fac0), mul4(Vec2, Fac3)), mul4(Vec3, Fac4));
    V4 Inv2 = add4(sub4(mul4(Vec0, Fac1), mul4(Vec1, Fac3)), mul4(Vec3, Fac5));
    V4 Inv3 = add4(sub4(mul4(Vec0, Fac2), mul4(Vec1, Fac4)), mul4(Vec2, Fac5));
    V4 SignA = v4(+1, -1, +1, -1);
    V4 SignB = v4(-1, +1, -1, +1);
    M4 Inverse;
    Inverse.c[0] = mul4(Inv0, SignA);
    Inverse.c[1] = mul4(Inv1, SignB);
    Inverse.c[2] = mul4(Inv2, SignA);
    Inverse.c[3] = mul4(Inv3, SignB);
    V4 Row0 = v4(Inverse.c[0].x, Inverse.c[1].x, Inverse.c[2].x, Inverse.c[3].x);
    V4 Dot0 = mul4(m.c[0], Row0);
    float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    float OneOverDeterminant = 1.0f / Dot1;
    M4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = muls4(Inverse.c[i], OneOverDeterminant);
    return r;
#undef M
}
/* glm/gtc/matrix_inverse.inl:95-158 */
M4 m4_inverse_transpose(const M4 &m) {
#define M(c, r) E(m, c, r)
    float S00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    float S01 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    float S02 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    float S03 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    float S04 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    float S05 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    float S06 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    float S07 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float S08 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    float S09 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    float S10 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    float S11 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float S12 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    float S13 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float S14 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float S15 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float S16 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float S17 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float S18 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    float I[4][4];
    I[0][0] = +(M(1, 1) * S00 - M(1, 2) * S01 + M(1, 3) * S02);
    I[0][1] = -(M(1, 0) * S00 - M(1, 2) * S03 + M(1, 3) * S04);
    I[0][2] = +(M(1, 0) * S01 - M(1, 1) * S03 + M(1, 3) * S05);
    I[0][3] = -(M(1, 0) * S02 - M(1, 1) * S04 + M(1, 2) * S05);
    I[1][0] = -(M(0, 1) * S00 - M(0, 2) * S01 + M(0, 3) * S02);
    I[1][1] = +(M(0, 0) * S00 - M(0, 2) * S03 + M(0, 3) * S04);
    I[1][2] = -(M(0, 0) * S01 - M(0, 1) * S03 + M(0, 3) * S05);
    I[1][3] = +(M(0, 0) * S02 - M(0, 1) * S04 + M(0, 2) * S05);
    I[2][0] = +(M(0, 1) * S06 - M(0, 2) * S07 + M(0, 3) * S08);
    I[2][1] = -(M(0, 0) * S06 - M(0, 2) * S09 + M(0, 3) * S10);
    I[2][2] = +(M(0, 0) * S11 - M(0, 1) * S09 + M(0, 3) * S12);
    I[2][3] = -(M(0, 0) * S08 - M(0, 1) * S10 + M(0, 2) * S12);
    I[3][0] = -(M(0, 1) * S13 - M(0, 2) * S14 + M(0, 3) * S15);
    I[3][1] = +(M(0, 0) * S13 - M(0, 2) * S16 + M(0, 3) * S17);
    I[3][2] = -(M(0, 0) * S14 - M(0, 1) * S16 + M(0, 3) * S18);
    I[3][3] = +(M(0, 0) * S15 - M(0, 1) * S17 + M(0, 2) * S18);
    float Determinant = +M(0, 0) * I[0][0] + M(0, 1) * I[0][1] + M(0, 2) * I[0][2] + M(0, 3) * I[0][3];
    M4 r;
    for (int c = 0; c < 4; ++c)
        r.c[c] = v4(I[c][0] / Determinant, I[c][1] / Determinant, I[c][2] / Determinant,
                    I[c][3] / Determinant);
    return r;
#undef M
}

/* src/utilities.cpp:65-72 */
M4 build_transformation_matrix(V3 translation, V3 rotation, V3 scale) {
    M4 translationMat = m4_translate(m4_identity(), translation);
    M4 rotationMat = m4_rotate(m4_identity(), rotation.x * kPI / 180, v3(1, 0, 0));
    rotationMat = m4_mul(rotationMat, m4_rotate(m4_identity(), rotation.y * kPI / 180, v3(0, 1, 0)));
    rotationMat = m4_mul(rotationMat, m4_rotate(m4_identity(), rotation.z * kPI / 180, v3(0, 0, 1)));
    M4 scaleMat = m4_scale(m4_identity(), scale);
    return m4_mul(m4_mul(translationMat, rotationMat), scaleMat);
}

/* ---- RNG: thrust::default_random_engine == minstd_rand (a=48271, c=0, m=2^31-1);
 *      thrust/random/detail/linear_congruential_engine.inl (seed, operator()),
 *      thrust/random/detail/uniform_real_distribution.inl:71-79 (u01).          */
const uint32_t kM = 2147483647u;
inline uint32_t rng_seed(uint32_t s) {
    uint32_t x = s % kM;
    return x == 0 ? 1u : x;
}
inline uint32_t rng_next(uint32_t &x) {
    x = (uint32_t)(((uint64_t)x * 48271ull) % (uint64_t)kM);
    return x;
}
inline float rng_u01(uint32_t &x) {
    float result = (float)(rng_next(x) - 1u);               /* urng() - min, min = 1 */
    result /= (1.0f + (float)(2147483646u - 1u));           /* 1 + float(max - min) == 2^31 */
    return (result * (1.0f - 0.0f)) + 0.0f;
}

inline uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}
/* src/pathtrace.cu:41-45 */
inline uint32_t make_seed(int iter, int index, int depth) {
    return utilhash((uint32_t)((1 << 31) | (depth << 22) | iter)) ^ utilhash((uint32_t)index);
}

/* ---- build-defined sin/cos (the reference calls the platform libm; CUDA's sinf/cosf and
 * glibc's differ by ulps, so the build fixes ONE polynomial, used bit-identically by the
 * HIP kernels).  Cody-Waite reduction by pi/2 (3 constants), cephes-style minimax
 * polynomials on [-pi/4, pi/4]; max error measured in tests/test_oracle_math.py. */
inline void sincos_poly(float x, float *s, float *c) {
    float kf = std::rint(x * 0.636619747f);  /* 2/pi, round-to-nearest-even */
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
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

/* ---- build-defined x^e for 0 <= x <= 1, 0 < e <= 1 (imperfect specular: cos(theta) = xi^(1/(n+1)), GPU Gems 3 ch. 20
 * eq. 7-9; the reference names the formula, README.md:171-185, and implements nothing).  exp2(e * log2(x)) with the
 * cephes logf / exp2f polynomials; like sincos_poly, ONE sequence of fp32 operations used bit-identically by the HIP
 * kernels (ptd::powPoly).  Relative error ~2e-6. */
inline float pow_poly(float x, float e) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    /* x = m 2^k, m in [sqrt(1/2), sqrt(2)) */
    uint32_t bits;
    memcpy(&bits, &x, 4);
    if (bits < 0x00800000u) return 0.0f;               /* subnormal xi (p ~ 1e-38 per draw): treated as 0 */
    int k = (int)(bits >> 23) - 127;
    uint32_t mb = (bits & 0x007fffffu) | 0x3f800000u;   /* m in [1, 2) */
    float m;
    memcpy(&m, &mb, 4);
    if (m > 1.41421356f) { m = m * 0.5f; k += 1; }
    float f = m - 1.0f;
    float z = f * f;
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
    float ln = f + y;                                   /* ln(m) */
    float l2 = ln * 1.44269504088896341f + (float)k;    /* log2(x) <= 0 */
    float t = e * l2;
    if (t < -126.0f) return 0.0f;
    float nf = std::rint(t);
    float g = t - nf;                                   /* [-0.5, 0.5] */
    float q = 1.535336188319500e-4f;
    q = q * g + 1.339887440266574e-3f;
    q = q * g + 9.618437357674640e-3f;
    q = q * g + 5.550332471162809e-2f;
    q = q * g + 2.402264791363012e-1f;
    q = q * g + 6.931472028550421e-1f;
    float r = q * g + 1.0f;                             /* 2^g */
    uint32_t rb;
    memcpy(&rb, &r, 4);
    rb += (uint32_t)((int)nf << 23);                    /* * 2^n, n in [-126, 0]: stays normal (r >= 0.70) or flushes below */
    float out;
    memcpy(&out, &rb, 4);
    return out;
}

struct Ray {
    V3 origin, direction;
};

/* src/intersections.h:26-28 */
inline V3 get_point_on_ray(const Ray &r, float t) {
    return add(r.origin, muls(normalize3(r.direction), t - .0001f));
}

inline float comp(V3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
inline void setcomp(V3 &v, int i, float f) {
    if (i == 0) v.x = f; else if (i == 1) v.y = f; else v.z = f;
}

/* src/intersections.h:47-89 */
float box_intersection_test(const OGeom &box, const Ray &r, V3 &intersectionPoint, V3 &normal,
                            bool &outside) {
    M4 inv = m4_from(box.inverseTransform);
    M4 xf = m4_from(box.transform);
    Ray q;
    q.origin = multiplyMV(inv, v4(r.origin.x, r.origin.y, r.origin.z, 1.0f));
    q.direction = normalize3(multiplyMV(inv, v4(r.direction.x, r.direction.y, r.direction.z, 0.0f)));

    float tmin = -1e38f;
    float tmax = 1e38f;
    V3 tmin_n = v3(0, 0, 0);
    V3 tmax_n = v3(0, 0, 0);
    for (int xyz = 0; xyz < 3; ++xyz) {
        float qdxyz = comp(q.direction, xyz);
        {
            float t1 = (-0.5f - comp(q.origin, xyz)) / qdxyz;
            float t2 = (+0.5f - comp(q.origin, xyz)) / qdxyz;
            float ta = t1 < t2 ? t1 : t2; /* glm::min, func_common.inl:409-414 */
            float tb = t1 > t2 ? t1 : t2; /* glm::max, func_common.inl:430-435 */
            V3 n = v3(0, 0, 0);
            setcomp(n, xyz, t2 < t1 ? +1.0f : -1.0f);
            if (ta > 0 && ta > tmin) {
                tmin = ta;
                tmin_n = n;
            }
            if (tb < tmax) {
                tmax = tb;
                tmax_n = n;
            }
        }
    }
    if (tmax >= tmin && tmax > 0) {
        outside = true;
        if (tmin <= 0) {
            tmin = tmax;
            tmin_n = tmax_n;
            outside = false;
        }
        V3 op = get_point_on_ray(q, tmin);
        intersectionPoint = multiplyMV(xf, v4(op.x, op.y, op.z, 1.0f));
        normal = normalize3(multiplyMV(xf, v4(tmin_n.x, tmin_n.y, tmin_n.z, 0.0f)));
        return length3(sub(r.origin, intersectionPoint));
    }
    return -1;
}

/* src/intersections.h:101-143.  pow(radius, 2) is exactly 0.25f in the float overload nvcc
 * selects; all arithmetic fp32. */
float sphere_intersection_test(const OGeom &sphere, const Ray &r, V3 &intersectionPoint, V3 &normal,
                               bool &outside) {
    M4 inv = m4_from(sphere.inverseTransform);
    M4 xf = m4_from(sphere.transform);
    M4 invT = m4_from(sphere.invTranspose);
    V3 ro = multiplyMV(inv, v4(r.origin.x, r.origin.y, r.origin.z, 1.0f));
    V3 rd = normalize3(multiplyMV(inv, v4(r.direction.x, r.direction.y, r.direction.z, 0.0f)));
    Ray rt;
    rt.origin = ro;
    rt.direction = rd;

    float vDotDirection = dot3(rt.origin, rt.direction);
    float radicand = vDotDirection * vDotDirection - (dot3(rt.origin, rt.origin) - 0.25f);
    if (radicand < 0) return -1;

    float squareRoot = std::sqrt(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;

    float t = 0;
    if (t1 < 0 && t2 < 0) {
        return -1;
    } else if (t1 > 0 && t2 > 0) {
        t = t2 < t1 ? t2 : t1; /* min(t1, t2) */
        outside = true;
    } else {
        t = t1 < t2 ? t2 : t1; /* max(t1, t2) */
        outside = false;
    }
    V3 obj = get_point_on_ray(rt, t);
    intersectionPoint = multiplyMV(xf, v4(obj.x, obj.y, obj.z, 1.f));
    normal = normalize3(multiplyMV(invT, v4(obj.x, obj.y, obj.z, 0.f)));
    if (!outside) normal = neg(normal);
    return length3(sub(r.origin, intersectionPoint));
}

/* ===================================================================== */
/* triangle meshes (README.md:112-116, 236: object type "mesh")            */
/* ===================================================================== */
/* The reference names meshes and `glm::intersectRayTriangle` but holds no mesh code: the semantics below are build-defined.
 *   - triangles live in OBJECT space (the OBJ file's coordinates); the ray is taken there like the sphere test does
 *     (intersections.h:104-110): ro = multiplyMV(inverseTransform, (o, 1)), rd = normalize(multiplyMV(inverseTransform, (d, 0)));
 *   - the triangle test is glm::intersectRayTriangle (glm/gtx/intersect.inl:36-72) op for op, made two-sided:
 *     `a < epsilon` becomes `|a| < epsilon`, and a > 0 is the front (counter-clockwise) side, reported as `outside`;
 *   - a triangle is only tested when the ray passes the slab test of the triangle's bounding box, inflated by a per-mesh
 *     margin, and its hit only counts at or beyond that box's entry parameter.  Both hold for every geometrically true hit;
 *     they make the rule local to (ray, triangle), so a bounding-volume hierarchy whose node boxes contain their triangles'
 *     boxes visits exactly the triangles this loop accepts (fp32 subtraction, multiplication by a common factor and
 *     comparison are monotone, so a ray that passes a box's test passes the test of every box containing it), and may skip
 *     nodes whose entry parameter lies beyond the best hit so far: any traversal order gives THIS function's result;
 *   - nearest = smallest object-space parameter t, ties to the lower triangle index; point and distance as the sphere test
 *     does (getPointOnRay, transform, world distance); flat normal normalize(invTranspose * normalize(cross(e1, e2))),
 *     negated on the back side. */
const float kMeshEps = 1.1920928955078125e-07f;     /* std::numeric_limits<float>::epsilon(), intersect.inl:50 */
const float kMeshUp = 1.00001f, kMeshDn = 0.99999f; /* relative slack of the slab comparison */

struct OMesh {
    int geom;
    std::vector<float> tris;    /* 9 floats per triangle: v0, v1, v2 */
    float margin;
    /* README.md:112-116 leftovers, build-defined (include/pt_amd.h states the semantics): */
    std::vector<float> normals; /* 9 floats per triangle: the vertex normals n0, n1, n2 (object space) -- or EMPTY: flat shading */
    std::vector<int> mats;      /* one per triangle: the scene material of that face, -1 = the object's own -- or EMPTY */
};

float mesh_margin(const float *tris, int ntris) {
    float maxAbs = 0.0f;
    for (int i = 0; i < 9 * ntris; ++i) {
        float a = std::fabs(tris[i]);
        if (a > maxAbs) maxAbs = a;
    }
    float m = 1e-5f * maxAbs;
    if (!(m >= 1e-30f)) m = 1e-30f;
    return m;
}

inline float min2(float a, float b) { return a < b ? a : b; }
inline float max2(float a, float b) { return a < b ? b : a; }

void tri_box(const float *t, float m, V3 &lo, V3 &hi) {
    lo = v3(min2(min2(t[0], t[3]), t[6]) - m, min2(min2(t[1], t[4]), t[7]) - m, min2(min2(t[2], t[5]), t[8]) - m);
    hi = v3(max2(max2(t[0], t[3]), t[6]) + m, max2(max2(t[1], t[4]), t[7]) + m, max2(max2(t[2], t[5]), t[8]) + m);
}

inline float guarded_reciprocal(float d) {
    float g = std::fabs(d) < 1e-30f ? std::copysign(1e-30f, d) : d;
    return 1.0f / g;
}

/* the slab test of an axis-aligned box; inv = guarded reciprocals of the direction, c = -(o * inv) per axis: the parameter of
 * a plane x = lo is ONE fused multiply-add, fma(lo, inv, c) (a single rounding of lo * inv - fl(o * inv): still monotone in
 * lo, which is what the hierarchy's equivalence rests on) */
bool mesh_slab(V3 lo, V3 hi, V3 inv, V3 c, float &tmin) {
    float ax = std::fmaf(lo.x, inv.x, c.x), bx = std::fmaf(hi.x, inv.x, c.x);
    float ay = std::fmaf(lo.y, inv.y, c.y), by = std::fmaf(hi.y, inv.y, c.y);
    float az = std::fmaf(lo.z, inv.z, c.z), bz = std::fmaf(hi.z, inv.z, c.z);
    float tn = max2(max2(min2(ax, bx), min2(ay, by)), min2(az, bz));
    float tf = min2(min2(max2(ax, bx), max2(ay, by)), max2(az, bz));
    tmin = tn * kMeshDn;
    return tf * kMeshUp >= tmin && tf >= 0.0f;
}

/* glm::intersectRayTriangle, two-sided; e1 = v1 - v0, e2 = v2 - v0 */
bool mesh_triangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float &t, bool &front, float *uv = NULL) {
    V3 p = cross3(d, e2);
    float a = dot3(e1, p);
    if (std::fabs(a) < kMeshEps) return false;
    float f = 1.0f / a;
    V3 s = sub(o, v0);
    float u = f * dot3(s, p);
    if (uv) uv[0] = u;
    if (u < 0.0f) return false;
    if (u > 1.0f) return false;
    V3 q = cross3(s, e1);
    float v = f * dot3(d, q);
    if (uv) uv[1] = v;
    if (v < 0.0f) return false;
    if (v + u > 1.0f) return false;
    t = f * dot3(e2, q);
    front = a > 0.0f;
    return t >= 0.0f;
}

float mesh_intersection_test(const OGeom &g, const OMesh &mesh, const Ray &r, V3 &intersectionPoint, V3 &normal,
                             bool &outside, int *triOut = NULL, int *matOut = NULL) {
    M4 inv = m4_from(g.inverseTransform);
    M4 xf = m4_from(g.transform);
    M4 invT = m4_from(g.invTranspose);
    Ray rt;
    rt.origin = multiplyMV(inv, v4(r.origin.x, r.origin.y, r.origin.z, 1.0f));
    rt.direction = normalize3(multiplyMV(inv, v4(r.direction.x, r.direction.y, r.direction.z, 0.0f)));
    V3 rinv = v3(guarded_reciprocal(rt.direction.x), guarded_reciprocal(rt.direction.y), guarded_reciprocal(rt.direction.z));
    V3 rc = v3(-(rt.origin.x * rinv.x), -(rt.origin.y * rinv.y), -(rt.origin.z * rinv.z));
    int best = -1;
    float tbest = 0.0f;
    bool bestFront = false;
    int ntris = (int)(mesh.tris.size() / 9);
    for (int i = 0; i < ntris; ++i) {
        const float *tv = &mesh.tris[9 * (size_t)i];
        V3 lo, hi;
        tri_box(tv, mesh.margin, lo, hi);
        float tmin;
        if (!mesh_slab(lo, hi, rinv, rc, tmin)) continue;
        V3 v0 = v3(tv[0], tv[1], tv[2]);
        V3 e1 = sub(v3(tv[3], tv[4], tv[5]), v0), e2 = sub(v3(tv[6], tv[7], tv[8]), v0);
        float t;
        bool front;
        if (!mesh_triangle(rt.origin, rt.direction, v0, e1, e2, t, front)) continue;
        if (!(t >= tmin)) continue;
        if (best < 0 || t < tbest) {   /* (index order: an equal t keeps the lower index) */
            best = i;
            tbest = t;
            bestFront = front;
        }
    }
    if (triOut) *triOut = best;
    if (best < 0) return -1;
    if (matOut) *matOut = mesh.mats.empty() ? -1 : mesh.mats[(size_t)best];
    const float *tv = &mesh.tris[9 * (size_t)best];
    V3 v0 = v3(tv[0], tv[1], tv[2]);
    V3 e1 = sub(v3(tv[3], tv[4], tv[5]), v0), e2 = sub(v3(tv[6], tv[7], tv[8]), v0);
    V3 nface = cross3(e1, e2);
    V3 nobj = normalize3(nface);
    if (!mesh.normals.empty()) {
        /* vertex normals (`vn`): the object-space normal is the barycentric blend n0 (1 - u - v) + n1 u + n2 v of the triangle's three,
         * (u, v) from the very triangle test that found the hit (evaluated once more for the winner: same operands, same bits); turned
         * to the side the counter-clockwise face normal points to; a blend of length zero (or NaN) keeps the face normal */
        float t2, uv[2] = {0.0f, 0.0f};
        bool f2;
        (void)mesh_triangle(rt.origin, rt.direction, v0, e1, e2, t2, f2, uv);
        const float *nn = &mesh.normals[9 * (size_t)best];
        float w = (1.0f - uv[0]) - uv[1];
        V3 ns = add(add(muls(v3(nn[0], nn[1], nn[2]), w), muls(v3(nn[3], nn[4], nn[5]), uv[0])), muls(v3(nn[6], nn[7], nn[8]), uv[1]));
        if (dot3(ns, nface) < 0.0f) ns = neg(ns);
        if (dot3(ns, ns) > 0.0f) nobj = normalize3(ns);
    }
    V3 obj = get_point_on_ray(rt, tbest);
    intersectionPoint = multiplyMV(xf, v4(obj.x, obj.y, obj.z, 1.f));
    normal = normalize3(multiplyMV(invT, v4(nobj.x, nobj.y, nobj.z, 0.f)));
    outside = bestFront;
    if (!outside) normal = neg(normal);
    return length3(sub(r.origin, intersectionPoint));
}

/* Wavefront OBJ: `v x y z` and `f a b c ...` (a = i, i/j, i//k or i/j/k; 1-based, negative = relative to the vertices read
 * so far); polygons are fanned from their first vertex; everything else is ignored.  false when the file cannot be read. */
bool load_obj(const std::string &path, std::vector<float> &tris, std::vector<float> *normalsOut = NULL, std::vector<int> *matsOut = NULL) {
    std::ifstream fp(path.c_str());
    if (!fp.is_open()) return false;
    std::vector<float> verts, vnorm;
    std::vector<float> normals;         /* 9 per triangle; allNormals says whether every corner of every face named one */
    std::vector<int> mats;
    bool allNormals = true, anyMat = false;
    int curMat = -1;
    std::string line;
    while (std::getline(fp, line)) {
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key)) continue;
        if (key == "v" || key == "vn") {
            double x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            std::vector<float> &dst = key == "v" ? verts : vnorm;
            dst.push_back((float)x); dst.push_back((float)y); dst.push_back((float)z);
        } else if (key == "usemtl") {
            /* `usemtl <k>`: the faces that follow take scene material k (an integer; anything else: back to the object's material) */
            std::string tok;
            curMat = -1;
            if (ss >> tok) {
                char *end = NULL;
                long k = strtol(tok.c_str(), &end, 10);
                if (end && *end == 0 && k >= 0 && k < 1000000) curMat = (int)k;
            }
        } else if (key == "f") {
            std::vector<int> idx, nidx;
            std::string tok;
            while (ss >> tok) {
                int i = atoi(tok.c_str());      /* stops at the first '/' */
                int nv = (int)(verts.size() / 3);
                int k = i > 0 ? i - 1 : nv + i;
                if (i == 0 || k < 0 || k >= nv) { idx.clear(); break; }
                idx.push_back(k);
                /* the normal index: the field behind the second '/' (i//k or i/j/k) */
                int nk = -1;
                size_t s1 = tok.find('/');
                size_t s2 = s1 == std::string::npos ? std::string::npos : tok.find('/', s1 + 1);
                if (s2 != std::string::npos && s2 + 1 < tok.size()) {
                    int j = atoi(tok.c_str() + s2 + 1);
                    int nn = (int)(vnorm.size() / 3);
                    int q = j > 0 ? j - 1 : nn + j;
                    if (j != 0 && q >= 0 && q < nn) nk = q;
                }
                nidx.push_back(nk);
            }
            for (size_t k = 2; k < idx.size(); ++k) {
                const int tri[3] = {idx[0], idx[k - 1], idx[k]};
                const int ntri[3] = {nidx[0], nidx[k - 1], nidx[k]};
                for (int c = 0; c < 3; ++c)
                    for (int a = 0; a < 3; ++a) {
                        tris.push_back(verts[3 * (size_t)tri[c] + a]);
                        normals.push_back(ntri[c] >= 0 ? vnorm[3 * (size_t)ntri[c] + a] : 0.0f);
                    }
                if (ntri[0] < 0 || ntri[1] < 0 || ntri[2] < 0) allNormals = false;
                mats.push_back(curMat);
                if (curMat >= 0) anyMat = true;
            }
        }
    }
    if (normalsOut) { if (allNormals && !tris.empty()) *normalsOut = normals; else normalsOut->clear(); }
    if (matsOut) { if (anyMat) *matsOut = mats; else matsOut->clear(); }
    return true;
}

/* src/interactions.h:10-42 */
V3 random_direction_in_hemisphere(V3 normal, uint32_t &rng) {
    float up = std::sqrt(rng_u01(rng));     /* cos(theta) */
    float over = std::sqrt(1 - up * up);    /* sin(theta) */
    float around = rng_u01(rng) * kTWO_PI;

    V3 directionNotNormal;
    if (std::fabs(normal.x) < kSQRT_OF_ONE_THIRD) {
        directionNotNormal = v3(1, 0, 0);
    } else if (std::fabs(normal.y) < kSQRT_OF_ONE_THIRD) {
        directionNotNormal = v3(0, 1, 0);
    } else {
        directionNotNormal = v3(0, 0, 1);
    }
    V3 p1 = normalize3(cross3(normal, directionNotNormal));
    V3 p2 = normalize3(cross3(normal, p1));
    float s, c;
    sincos_poly(around, &s, &c);
    return add(add(muls(normal, up), muls(p1, c * over)), muls(p2, s * over));
}

/* Imperfect specular (README.md:171-185 -> GPU Gems 3 ch. 20 eq. 7-9): a direction in the Phong lobe of exponent n around the
 * mirror direction R: cos(theta) = xi1^(1/(n+1)), phi = 2 pi xi2, in the tangent frame the hemisphere sampler builds
 * around a vector.  Build-defined detail: a sample below the surface falls back to R itself. */
V3 random_direction_in_specular_lobe(V3 R, V3 normal, float invExpPlus1, uint32_t &rng) {
    float cosT = pow_poly(rng_u01(rng), invExpPlus1);
    float sinT = std::sqrt(1 - cosT * cosT);
    float around = rng_u01(rng) * kTWO_PI;
    V3 directionNotR;
    if (std::fabs(R.x) < kSQRT_OF_ONE_THIRD) {
        directionNotR = v3(1, 0, 0);
    } else if (std::fabs(R.y) < kSQRT_OF_ONE_THIRD) {
        directionNotR = v3(0, 1, 0);
    } else {
        directionNotR = v3(0, 0, 1);
    }
    V3 p1 = normalize3(cross3(R, directionNotR));
    V3 p2 = normalize3(cross3(R, p1));
    float s, c;
    sincos_poly(around, &s, &c);
    V3 d = add(add(muls(R, cosT), muls(p1, c * sinT)), muls(p2, s * sinT));
    return dot3(d, normal) > 0.0f ? d : R;
}

}  // namespace

/* ===================================================================== */
/* scene + renderer objects                                               */
/* ===================================================================== */
struct OScene {
    std::vector<OGeom> geoms;
    std::vector<OMaterial> materials;
    OCamera camera;
    int iterations;
    int traceDepth;
    std::string imageName;
    std::string dir;                   /* directory of the scene file: `mesh <file>` paths are relative to it */
    std::vector<OMesh> meshes;         /* geoms of type 2 */
    bool failed = false;               /* a `mesh <file>` object whose file cannot be read: the load fails (like an unreadable scene file) */
};

struct ORender {
    OCamera cam;
    std::vector<OGeom> geoms;
    std::vector<OMaterial> mats;
    int traceDepth;
    /* derived camera constants (spec S2) */
    V3 view, up, right, position;
    float pixLenX, pixLenY, halfW, halfH;
    /* README extras, all off by default = the behaviour every other test pins */
    float lensRadius, focalDistance;   /* thin lens (README.md:100-101); radius 0 = pinhole */
    V3 viewN;                          /* normalize(view) */
    int directLighting;                /* README.md:107-108: a final ray to a random point of an emissive object */
    /* STUDY VARIANTS (orc_render_set_variant; never used by a parity test): where the reference is silent the pipeline is build-defined
     * (spec S6), and the one external anchor -- the staff render img/REFERENCE_cornell.5000samp.png -- differs from it by a systematic
     * 2 % on the back wall.  These switches render the candidate explanations in the oracle (DESIGN.md section 2 tabulates them). */
    float scatterOffset = 0.001f;      /* new origin = hit +- offset * normal */
    int mirrorMode = 0;                /* REFL > 0 materials: 0 = 50/50 mirror / diffuse, energy conserving (the build's choice); 1 = 50/50 with
                                          the 1 / p weights (either branch x 2); 2 = a pure mirror; 3 = the split "based on the intensity of
                                          each material color", either branch divided by its probability (src/interactions.h:56-59) */
    int emitColorMode = 0;             /* 0 = an emitter hit contributes throughput x m.color x emittance (the build's choice: `color *= m.color`
                                          BEFORE the emitter test); 1 = throughput x emittance (m.color applied only to paths that go on) */
    /* what the direct-lighting bounce samples (file order): every primitive with an emissive material -- a mesh: also one whose faces name
     * an emissive material of their own -- with the object-space box it is sampled through: the unit cube [-.5, .5]^3 of a sphere or cube
     * (centre 0, extent 1), the bounds of a mesh's vertices (round 5: README.md:107-108 x :112-116, emissive meshes) */
    struct Emitter { int geom; V3 c, e; };
    std::vector<Emitter> emitters;
    std::vector<OMesh> meshes;         /* triangle data of the geoms of type 2 */
    std::vector<int> meshOf;           /* geom -> index into meshes, -1 */
};

namespace {

inline V3 from(const OVec3 &v) { return v3(v.x, v.y, v.z); }

/* src/utilities.cpp:82-112 */
std::istream &safe_getline(std::istream &is, std::string &t) {
    t.clear();
    std::istream::sentry se(is, true);
    std::streambuf *sb = is.rdbuf();
    for (;;) {
        int c = sb->sbumpc();
        switch (c) {
            case '\n': return is;
            case '\r':
                if (sb->sgetc() == '\n') sb->sbumpc();
                return is;
            case EOF:
                if (t.empty()) is.setstate(std::ios::eofbit);
                return is;
            default: t += (char)c;
        }
    }
}
/* src/utilities.cpp:74-80 */
std::vector<std::string> tokenize(const std::string &s) {
    std::stringstream ss(s);
    std::vector<std::string> out;
    std::string tok;
    while (ss >> tok) out.push_back(tok);
    return out;
}
inline float atoff(const std::string &s) { return (float)atof(s.c_str()); }
/* the k-th token, "" when the line is too short (the reference indexes unchecked, src/scene.cpp:103-111,160-172: undefined there) */
inline const std::string &tokk(const std::vector<std::string> &t, size_t k) {
    static const std::string empty;
    return k < t.size() ? t[k] : empty;
}
inline bool is(const std::vector<std::string> &t, const char *key) {
    return !t.empty() && strcmp(t[0].c_str(), key) == 0;
}
inline OVec3 tok3(const std::vector<std::string> &t) {
    OVec3 v = {0, 0, 0};
    if (t.size() >= 4) { v.x = atoff(t[1]); v.y = atoff(t[2]); v.z = atoff(t[3]); }
    return v;
}

/* src/scene.cpp:132-136 */
void compute_fov(OCamera &cam, float fovy) {
    float yscaled = std::tan(fovy * (kPI / 180));
    float xscaled = (yscaled * cam.resX) / cam.resY;
    float fovx = (std::atan(xscaled) * 180) / kPI;
    cam.fovX = fovx;
    cam.fovY = fovy;
}

/* src/scene.cpp:147-182 */
void load_material(OScene &sc, std::ifstream &fp, const std::string &idtok) {
    int id = atoi(idtok.c_str());
    if (id != (int)sc.materials.size()) return; /* :149-151 prints ERROR and skips */
    OMaterial m;
    memset(&m, 0, sizeof m);
    for (int i = 0; i < 7; ++i) {
        std::string line;
        safe_getline(fp, line);
        std::vector<std::string> t = tokenize(line);
        if (is(t, "RGB")) m.color = tok3(t);
        else if (is(t, "SPECEX")) m.specExponent = atoff(tokk(t, 1));
        else if (is(t, "SPECRGB")) m.specColor = tok3(t);
        else if (is(t, "REFL")) m.hasReflective = atoff(tokk(t, 1));
        else if (is(t, "REFR")) m.hasRefractive = atoff(tokk(t, 1));
        else if (is(t, "REFRIOR")) m.indexOfRefraction = atoff(tokk(t, 1));
        else if (is(t, "EMITTANCE")) m.emittance = atoff(tokk(t, 1));
    }
    sc.materials.push_back(m);
}

/* src/scene.cpp:92-145 */
void load_camera(OScene &sc, std::ifstream &fp) {
    OCamera &cam = sc.camera;
    float fovy = 0;
    for (int i = 0; i < 5; ++i) {
        std::string line;
        safe_getline(fp, line);
        std::vector<std::string> t = tokenize(line);
        if (is(t, "RES")) { cam.resX = atoi(tokk(t, 1).c_str()); cam.resY = atoi(tokk(t, 2).c_str()); }
        else if (is(t, "FOVY")) fovy = atoff(tokk(t, 1));
        else if (is(t, "ITERATIONS")) sc.iterations = atoi(tokk(t, 1).c_str());
        else if (is(t, "DEPTH")) sc.traceDepth = atoi(tokk(t, 1).c_str());
        else if (is(t, "FILE")) sc.imageName = tokk(t, 1);
    }
    std::string line;
    safe_getline(fp, line);
    while (!line.empty() && fp.good()) {
        std::vector<std::string> t = tokenize(line);
        if (is(t, "EYE")) cam.position = tok3(t);
        else if (is(t, "VIEW")) cam.view = tok3(t);
        else if (is(t, "UP")) cam.up = tok3(t);
        safe_getline(fp, line);
    }
    compute_fov(cam, fovy);
}

/* src/scene.cpp:35-90 */
void load_geom(OScene &sc, std::ifstream &fp, const std::string &idtok) {
    int id = atoi(idtok.c_str());
    if (id != (int)sc.geoms.size()) return; /* :37-39 */
    OGeom g;
    memset(&g, 0, sizeof g);
    std::string line;
    safe_getline(fp, line);
    if (!line.empty() && fp.good()) {
        if (strcmp(line.c_str(), "sphere") == 0) g.type = 0;
        else if (strcmp(line.c_str(), "cube") == 0) g.type = 1;
        else {
            /* README.md:236 names a third type, "mesh"; the file is build-defined: `mesh <path.obj>`, relative to the scene */
            std::vector<std::string> t = tokenize(line);
            if (t.size() >= 2 && t[0] == "mesh") {
                OMesh m;
                m.geom = id;
                std::string path = t[1][0] == '/' ? t[1] : sc.dir + t[1];
                if (load_obj(path, m.tris, &m.normals, &m.mats) && !m.tris.empty()) {
                    m.margin = mesh_margin(m.tris.data(), (int)(m.tris.size() / 9));
                    sc.meshes.push_back(m);
                    g.type = 2;
                } else {
                    sc.failed = true;
                }
            }
        }
    }
    safe_getline(fp, line);
    if (!line.empty() && fp.good()) {
        std::vector<std::string> t = tokenize(line);
        if (t.size() >= 2) g.materialid = atoi(t[1].c_str());
    }
    safe_getline(fp, line);
    while (!line.empty() && fp.good()) {
        std::vector<std::string> t = tokenize(line);
        if (is(t, "TRANS")) g.translation = tok3(t);
        else if (is(t, "ROTAT")) g.rotation = tok3(t);
        else if (is(t, "SCALE")) g.scale = tok3(t);
        safe_getline(fp, line);
    }
    M4 xf = build_transformation_matrix(from(g.translation), from(g.rotation), from(g.scale));
    m4_to(xf, g.transform);
    m4_to(m4_inverse(xf), g.inverseTransform);
    m4_to(m4_inverse_transpose(xf), g.invTranspose);
    sc.geoms.push_back(g);
}

/* Nearest hit over all geoms in file order (spec S3).  Returns geom index or -1. */
int nearest_hit(const ORender &R, const Ray &ray, V3 &p, V3 &n, bool &outside, int *faceMat = NULL) {
    float t_min = 0;
    int hit = -1;
    if (faceMat) *faceMat = -1;
    for (int i = 0; i < (int)R.geoms.size(); ++i) {
        V3 tp = v3(0, 0, 0), tn = v3(0, 0, 0);
        bool to = false;
        float t;
        int fm = -1;
        if (R.geoms[i].type == 2) {
            /* a mesh geom without triangle data is never hit */
            t = i < (int)R.meshOf.size() && R.meshOf[i] >= 0 ? mesh_intersection_test(R.geoms[i], R.meshes[R.meshOf[i]], ray, tp, tn, to, NULL, &fm) : -1.0f;
        } else {
            t = R.geoms[i].type == 0 ? sphere_intersection_test(R.geoms[i], ray, tp, tn, to)
                                     : box_intersection_test(R.geoms[i], ray, tp, tn, to);
        }
        if (t > 0.0f && (hit < 0 || t < t_min)) {
            t_min = t;
            hit = i;
            p = tp;
            n = tn;
            outside = to;
            if (faceMat) *faceMat = fm;
        }
    }
    return hit;
}

/* spec S2 */
Ray camera_ray(const ORender &R, int iter, int index) {
    int W = R.cam.resX;
    int x = index % W, y = index / W;
    uint32_t rng = rng_seed(make_seed(iter, index, 0));
    float jx = rng_u01(rng);
    float jy = rng_u01(rng);
    float sx = ((float)x + jx) - R.halfW;
    float sy = ((float)y + jy) - R.halfH;
    float a = R.pixLenX * sx;
    float b = R.pixLenY * sy;
    Ray r;
    r.origin = R.position;
    r.direction = normalize3(sub(sub(R.view, muls(R.right, a)), muls(R.up, b)));
    if (R.lensRadius > 0.0f) {
        /* depth of field by jittering rays within an aperture (README.md:100-101, PBRT 6.2.3): the pinhole ray fixes the
         * point in focus, at distance focalDistance along the view axis; the ray starts at a uniformly sampled point of
         * the lens disc (two more draws of the depth-0 stream) and aims at it */
        float lr = R.lensRadius * std::sqrt(rng_u01(rng));
        float phi = rng_u01(rng) * kTWO_PI;
        float s, c;
        sincos_poly(phi, &s, &c);
        float ft = R.focalDistance / dot3(r.direction, R.viewN);
        V3 focus = add(R.position, muls(r.direction, ft));
        r.origin = add(add(R.position, muls(R.right, lr * c)), muls(R.up, lr * s));
        r.direction = normalize3(sub(focus, r.origin));
    }
    return r;
}

enum Fate { ALIVE = 0, MISS = 1, LIGHT = 2 };

/* One bounce of one path (spec S3-S6).  On LIGHT, `contrib` holds the radiance to add. */
/* diffuse scatter of the direct-lighting bounce: instead of a hemisphere sample, a ray to a uniformly chosen point of the
 * (transformed) unit cube of a uniformly chosen emissive primitive, weighted by the cosine at the surface and by the
 * share of the hemisphere the emitter's bounding ball covers, min(1, rho^2 / r^2) with rho^2 = |scale|^2 / 4 (a hemisphere
 * sample would have found the light with about that probability) */
static void scatter_to_light(const ORender &R, V3 n, V3 norg, V3 mcol, uint32_t &rng, V3 &ndir, V3 &color) {
    int ne = (int)R.emitters.size();
    int pick = (int)(rng_u01(rng) * (float)ne);
    if (pick > ne - 1) pick = ne - 1;
    const ORender::Emitter &E = R.emitters[pick];
    const OGeom &L = R.geoms[E.geom];
    float ux = rng_u01(rng) - 0.5f;
    float uy = rng_u01(rng) - 0.5f;
    float uz = rng_u01(rng) - 0.5f;
    /* a point of the emitter's object-space box: centre + u x extent (the unit cube: 0 + u x 1 = u) */
    V3 target = multiplyMV(m4_from(L.transform), v4(E.c.x + ux * E.e.x, E.c.y + uy * E.e.y, E.c.z + uz * E.e.z, 1.0f));
    V3 toward = sub(target, norg);
    ndir = normalize3(toward);
    float w = dot3(n, ndir);
    w = w > 0.0f ? w : 0.0f;
    /* rho^2 = |scale x extent|^2 / 4: the squared radius of the ball around the box */
    const float sx = L.scale.x * E.e.x, sy = L.scale.y * E.e.y, sz = L.scale.z * E.e.z;
    float rho2 = ((sx * sx + sy * sy) + sz * sz) * 0.25f;
    float cover = rho2 / dot3(toward, toward);
    cover = cover < 1.0f ? cover : 1.0f;
    color = muls(mul(color, mcol), w * cover);
}

Fate bounce(const ORender &R, int iter, int index, int depth, Ray &ray, V3 &color, V3 &contrib, bool direct = false) {
    V3 p = v3(0, 0, 0), n = v3(0, 0, 0);
    bool outside = false;
    int faceMat = -1;
    int g = nearest_hit(R, ray, p, n, outside, &faceMat);
    if (g < 0) return MISS;
    /* (a mesh face with a material of its own, `usemtl`; ids beyond the scene's materials fall back to the object's) */
    const OMaterial &m = R.mats[faceMat >= 0 && faceMat < (int)R.mats.size() ? faceMat : R.geoms[g].materialid];
    V3 mcol = from(m.color);
    if (m.emittance > 0.0f) {
        contrib = R.emitColorMode == 1 ? muls(color, m.emittance) : muls(mul(color, mcol), m.emittance);
        return LIGHT;
    }
    uint32_t rng = rng_seed(make_seed(iter, index, depth));
    V3 scol = from(m.specColor);
    V3 ndir;
    V3 norg;
    if (m.hasRefractive > 0.0f) {
        /* dielectric, Schlick Fresnel (README.md:96-99); normal always faces the incoming ray */
        float ior = m.indexOfRefraction;
        float eta = outside ? 1.0f / ior : ior;
        float c = dot3(n, ray.direction);
        float k = 1.0f - eta * eta * (1.0f - c * c);
        float u = rng_u01(rng);
        bool doReflect = true;
        if (k >= 0.0f) {
            float r0 = (1.0f - ior) / (1.0f + ior);
            r0 = r0 * r0;
            float cosx = outside ? -c : std::sqrt(k);
            float w = 1.0f - cosx;
            float w2 = w * w;
            float w5 = w2 * w2 * w;
            float fres = r0 + (1.0f - r0) * w5;
            doReflect = u < fres;
        }
        if (doReflect) {
            ndir = reflect3(ray.direction, n);
            norg = add(p, muls(n, R.scatterOffset));
            color = mul(color, scol);
        } else {
            ndir = refract3(ray.direction, n, eta);
            norg = sub(p, muls(n, R.scatterOffset));
            color = mul(color, mcol);
        }
    } else if (m.hasReflective > 0.0f) {
        /* energy-conserving 50/50 mirror/diffuse mixture (spec S6) */
        float u = rng_u01(rng);
        norg = add(p, muls(n, R.scatterOffset));
        /* (study variant 3: p(mirror) = I(specular colour) / (I(specular colour) + I(diffuse colour)), I = r + g + b) */
        float is = (scol.x + scol.y) + scol.z, id = (mcol.x + mcol.y) + mcol.z;
        float pSpec = R.mirrorMode == 3 && is + id > 0.0f ? is / (is + id) : 0.5f;
        bool mirror = u < pSpec || R.mirrorMode == 2;
        if (mirror) {
            ndir = reflect3(ray.direction, n);
            if (m.specExponent > 0.0f)      /* SPECEX > 0: imperfect specular (README.md:171-185); 0 = the perfect mirror */
                ndir = random_direction_in_specular_lobe(ndir, n, 1.0f / (m.specExponent + 1.0f), rng);
            color = mul(color, scol);
        } else if (direct && !R.emitters.empty()) {
            scatter_to_light(R, n, norg, mcol, rng, ndir, color);
        } else {
            ndir = random_direction_in_hemisphere(n, rng);
            color = mul(color, mcol);
        }
        if (R.mirrorMode == 1) color = muls(color, 2.0f);      /* (study variant: the 1 / p weight of the branch taken) */
        if (R.mirrorMode == 3) color = muls(color, 1.0f / (mirror ? pSpec : 1.0f - pSpec));
    } else {
        norg = add(p, muls(n, R.scatterOffset));
        if (direct && !R.emitters.empty()) {
            scatter_to_light(R, n, norg, mcol, rng, ndir, color);
        } else {
            ndir = random_direction_in_hemisphere(n, rng);
            color = mul(color, mcol);
        }
    }
    ray.origin = norg;
    ray.direction = ndir;
    return ALIVE;
}

inline int shard_rows(int H, int rank, int count) { return (H - rank + count - 1) / count; }

}  // namespace

extern "C" {

uint32_t orc_utilhash(uint32_t a) { return utilhash(a); }
uint32_t orc_seed(int iter, int index, int depth) { return make_seed(iter, index, depth); }

void orc_rng_stream_from_seed(uint32_t seed, int n, float *u01_out, uint32_t *state_out) {
    uint32_t x = rng_seed(seed);
    for (int i = 0; i < n; ++i) {
        float u = rng_u01(x);
        if (u01_out) u01_out[i] = u;
        if (state_out) state_out[i] = x;
    }
}
void orc_rng_stream(int iter, int index, int depth, int n, float *u01_out, uint32_t *state_out) {
    orc_rng_stream_from_seed(make_seed(iter, index, depth), n, u01_out, state_out);
}
void orc_sincos(float x, float *s, float *c) { sincos_poly(x, s, c); }
void orc_normalize(const float v[3], float out[3]) {
    V3 r = normalize3(v3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_reflect(const float I[3], const float N[3], float out[3]) {
    V3 r = reflect3(v3(I[0], I[1], I[2]), v3(N[0], N[1], N[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float I[3], const float N[3], float eta, float out[3]) {
    V3 r = refract3(v3(I[0], I[1], I[2]), v3(N[0], N[1], N[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_mulmv(const float m[16], const float v[4], float out[3]) {
    V3 r = multiplyMV(m4_from(m), v4(v[0], v[1], v[2], v[3]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_point_on_ray(const float ray[6], float t, float out[3]) {
    Ray r;
    r.origin = v3(ray[0], ray[1], ray[2]);
    r.direction = v3(ray[3], ray[4], ray[5]);
    V3 p = get_point_on_ray(r, t);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
void orc_build_transform(const float t[3], const float r[3], const float s[3], float transform[16],
                         float inverse[16], float invTranspose[16]) {
    M4 xf = build_transformation_matrix(v3(t[0], t[1], t[2]), v3(r[0], r[1], r[2]), v3(s[0], s[1], s[2]));
    m4_to(xf, transform);
    m4_to(m4_inverse(xf), inverse);
    m4_to(m4_inverse_transpose(xf), invTranspose);
}
static float isect(const OGeom *g, const float ray[6], float p[3], float n[3], int *outside, bool sphere) {
    Ray r;
    r.origin = v3(ray[0], ray[1], ray[2]);
    r.direction = v3(ray[3], ray[4], ray[5]);
    V3 P = v3(p[0], p[1], p[2]), N = v3(n[0], n[1], n[2]);
    bool o = *outside != 0;
    float t = sphere ? sphere_intersection_test(*g, r, P, N, o) : box_intersection_test(*g, r, P, N, o);
    p[0] = P.x; p[1] = P.y; p[2] = P.z;
    n[0] = N.x; n[1] = N.y; n[2] = N.z;
    *outside = o ? 1 : 0;
    return t;
}
float orc_box_intersect(const OGeom *g, const float ray[6], float p[3], float n[3], int *outside) {
    return isect(g, ray, p, n, outside, false);
}
float orc_sphere_intersect(const OGeom *g, const float ray[6], float p[3], float n[3], int *outside) {
    return isect(g, ray, p, n, outside, true);
}
void orc_hemisphere(const float n[3], uint32_t *rng_state, float out[3]) {
    V3 r = random_direction_in_hemisphere(v3(n[0], n[1], n[2]), *rng_state);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_hemisphere_seeded(const float n[3], int iter, int index, int depth, float out[3]) {
    uint32_t x = rng_seed(make_seed(iter, index, depth));
    orc_hemisphere(n, &x, out);
}

/* ---- scene -------------------------------------------------------------- */
OScene *orc_scene_load(const char *path) {
    std::ifstream fp(path);
    if (!fp.is_open()) return NULL; /* scene.cpp:12-15 aborts */
    OScene *sc = new OScene();
    memset(&sc->camera, 0, sizeof(OCamera));
    sc->iterations = 0;
    sc->traceDepth = 0;
    {
        std::string sp(path);
        size_t slash = sp.find_last_of('/');
        sc->dir = slash == std::string::npos ? std::string() : sp.substr(0, slash + 1);
    }
    while (fp.good()) { /* scene.cpp:16-32 */
        std::string line;
        safe_getline(fp, line);
        if (!line.empty()) {
            std::vector<std::string> t = tokenize(line);
            if (t.size() >= 2 && is(t, "MATERIAL")) load_material(*sc, fp, t[1]);
            else if (t.size() >= 2 && is(t, "OBJECT")) load_geom(*sc, fp, t[1]);
            else if (is(t, "CAMERA")) load_camera(*sc, fp);
        }
    }
    if (sc->failed) {
        delete sc;
        return NULL;
    }
    return sc;
}
void orc_scene_free(OScene *s) { delete s; }
int orc_scene_num_geoms(const OScene *s) { return (int)s->geoms.size(); }
int orc_scene_num_materials(const OScene *s) { return (int)s->materials.size(); }
const OGeom *orc_scene_geoms(const OScene *s) { return s->geoms.data(); }
const OMaterial *orc_scene_materials(const OScene *s) { return s->materials.data(); }
const OCamera *orc_scene_camera(const OScene *s) { return &s->camera; }
int orc_scene_iterations(const OScene *s) { return s->iterations; }
int orc_scene_depth(const OScene *s) { return s->traceDepth; }
const char *orc_scene_image_name(const OScene *s) { return s->imageName.c_str(); }
int orc_scene_num_meshes(const OScene *s) { return (int)s->meshes.size(); }
int orc_scene_mesh_geom(const OScene *s, int i) { return s->meshes[i].geom; }
int orc_scene_mesh_ntris(const OScene *s, int i) { return (int)(s->meshes[i].tris.size() / 9); }
const float *orc_scene_mesh_tris(const OScene *s, int i) { return s->meshes[i].tris.data(); }
const float *orc_scene_mesh_normals(const OScene *s, int i) { return s->meshes[i].normals.empty() ? NULL : s->meshes[i].normals.data(); }
const int *orc_scene_mesh_materials(const OScene *s, int i) { return s->meshes[i].mats.empty() ? NULL : s->meshes[i].mats.data(); }
void orc_camera_set_resolution(OCamera *cam, int w, int h) {
    cam->resX = w;
    cam->resY = h;
    compute_fov(*cam, cam->fovY);
}

/* ---- renderer ------------------------------------------------------------ */
/* the emitters of the direct-lighting bounce (ORender::emitters), from the primitives, their materials and the meshes set so far */
static void rebuild_emitters(ORender *R) {
    R->emitters.clear();
    for (int i = 0; i < (int)R->geoms.size(); ++i) {
        const OGeom &g = R->geoms[i];
        bool emits = g.materialid >= 0 && g.materialid < (int)R->mats.size() && R->mats[g.materialid].emittance > 0.0f;
        ORender::Emitter E;
        E.geom = i;
        E.c = v3(0, 0, 0);
        E.e = v3(1, 1, 1);
        if (g.type == 2) {
            const int mi = i < (int)R->meshOf.size() ? R->meshOf[i] : -1;
            if (mi < 0) continue;
            const OMesh &m = R->meshes[(size_t)mi];
            for (size_t f = 0; f < m.mats.size(); ++f)
                emits = emits || (m.mats[f] >= 0 && m.mats[f] < (int)R->mats.size() && R->mats[m.mats[f]].emittance > 0.0f);
            V3 lo = v3(m.tris[0], m.tris[1], m.tris[2]), hi = lo;
            for (size_t q = 0; q + 2 < m.tris.size(); q += 3) {
                lo = v3(min2(lo.x, m.tris[q]), min2(lo.y, m.tris[q + 1]), min2(lo.z, m.tris[q + 2]));
                hi = v3(max2(hi.x, m.tris[q]), max2(hi.y, m.tris[q + 1]), max2(hi.z, m.tris[q + 2]));
            }
            E.c = muls(add(lo, hi), 0.5f);
            E.e = sub(hi, lo);
        }
        if (emits) R->emitters.push_back(E);
    }
}

ORender *orc_render_create(const OCamera *cam, const OGeom *geoms, int ngeoms, const OMaterial *mats,
                           int nmats, int traceDepth) {
    ORender *R = new ORender();
    R->cam = *cam;
    R->geoms.assign(geoms, geoms + ngeoms);
    R->mats.assign(mats, mats + nmats);
    R->traceDepth = traceDepth;
    R->view = from(cam->view);
    R->up = from(cam->up);
    R->position = from(cam->position);
    R->right = normalize3(cross3(R->view, R->up));
    float ys = std::tan(cam->fovY * (kPI / 180));
    float xs = (ys * cam->resX) / cam->resY;
    R->pixLenX = (2.0f * xs) / (float)cam->resX;
    R->pixLenY = (2.0f * ys) / (float)cam->resY;
    R->halfW = (float)cam->resX * 0.5f;
    R->halfH = (float)cam->resY * 0.5f;
    R->lensRadius = 0.0f;
    R->focalDistance = 0.0f;
    R->viewN = normalize3(R->view);
    R->directLighting = 0;
    R->meshOf.assign(ngeoms, -1);
    rebuild_emitters(R);
    return R;
}
void orc_render_set_mesh(ORender *R, int geom, const float *tris, int ntris) {
    if (geom < 0 || geom >= (int)R->geoms.size() || ntris <= 0) return;
    OMesh m;
    m.geom = geom;
    m.tris.assign(tris, tris + 9 * (size_t)ntris);
    m.margin = mesh_margin(tris, ntris);
    R->meshOf[geom] = (int)R->meshes.size();
    R->meshes.push_back(m);
    rebuild_emitters(R);
}
/* vertex normals (ntris x 9, or NULL: flat) and face materials (ntris, or NULL: the object's) of a mesh set before */
void orc_render_set_mesh_attributes(ORender *R, int geom, const float *normals, const int *mats) {
    if (geom < 0 || geom >= (int)R->meshOf.size() || R->meshOf[geom] < 0) return;
    OMesh &m = R->meshes[(size_t)R->meshOf[geom]];
    const size_t nt = m.tris.size() / 9;
    if (normals) m.normals.assign(normals, normals + 9 * nt); else m.normals.clear();
    if (mats) m.mats.assign(mats, mats + nt); else m.mats.clear();
    rebuild_emitters(R);
}
float orc_mesh_margin(const float *tris, int ntris) { return mesh_margin(tris, ntris); }
/* the two-sided triangle test alone: returns hit; tuv = (t, u, v) as far as they were evaluated (glm's baryPosition is (u, v, t)) */
int orc_mesh_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3], float tuv[3],
                      int *front) {
    V3 a = v3(v0[0], v0[1], v0[2]);
    V3 e1 = sub(v3(v1[0], v1[1], v1[2]), a), e2 = sub(v3(v2[0], v2[1], v2[2]), a);
    float t = tuv[0];
    bool f = *front != 0;
    bool hit = mesh_triangle(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), a, e1, e2, t, f, tuv + 1);
    tuv[0] = t;
    *front = f ? 1 : 0;
    return hit ? 1 : 0;
}
float orc_mesh_intersect_attr(const OGeom *g, const float *tris, int ntris, const float *normals, const float ray[6], float p[3], float n[3],
                              int *outside, int *tri) {
    OMesh m;
    m.geom = 0;
    m.tris.assign(tris, tris + 9 * (size_t)ntris);
    m.margin = mesh_margin(tris, ntris);
    if (normals) m.normals.assign(normals, normals + 9 * (size_t)ntris);
    Ray r;
    r.origin = v3(ray[0], ray[1], ray[2]);
    r.direction = v3(ray[3], ray[4], ray[5]);
    V3 tp = v3(p[0], p[1], p[2]), tn = v3(n[0], n[1], n[2]);
    bool to = *outside != 0;
    float t = mesh_intersection_test(*g, m, r, tp, tn, to, tri);
    p[0] = tp.x; p[1] = tp.y; p[2] = tp.z;
    n[0] = tn.x; n[1] = tn.y; n[2] = tn.z;
    *outside = to ? 1 : 0;
    return t;
}
float orc_mesh_intersect(const OGeom *g, const float *tris, int ntris, const float ray[6], float p[3], float n[3], int *outside,
                         int *tri) {
    OMesh m;
    m.geom = 0;
    m.tris.assign(tris, tris + 9 * (size_t)ntris);
    m.margin = mesh_margin(tris, ntris);
    Ray r;
    r.origin = v3(ray[0], ray[1], ray[2]);
    r.direction = v3(ray[3], ray[4], ray[5]);
    V3 tp = v3(p[0], p[1], p[2]), tn = v3(n[0], n[1], n[2]);
    bool to = *outside != 0;
    float t = mesh_intersection_test(*g, m, r, tp, tn, to, tri);
    p[0] = tp.x; p[1] = tp.y; p[2] = tp.z;
    n[0] = tn.x; n[1] = tn.y; n[2] = tn.z;
    *outside = to ? 1 : 0;
    return t;
}
void orc_render_set_extras(ORender *R, float lensRadius, float focalDistance, int directLighting) {
    R->lensRadius = lensRadius;
    R->focalDistance = focalDistance;
    R->directLighting = directLighting;
}
/* study variants (see ORender): offset of the scattered ray's origin, treatment of REFL > 0 materials */
void orc_render_set_variant(ORender *R, float scatterOffset, int mirrorMode) {
    R->scatterOffset = scatterOffset;
    R->mirrorMode = mirrorMode;
}
void orc_render_set_emit_variant(ORender *R, int emitColorMode) { R->emitColorMode = emitColorMode; }
float orc_pow(float x, float e) { return pow_poly(x, e); }
void orc_render_free(ORender *R) { delete R; }

void orc_camera_ray(ORender *R, int iter, int index, float ray[6]) {
    Ray r = camera_ray(*R, iter, index);
    ray[0] = r.origin.x; ray[1] = r.origin.y; ray[2] = r.origin.z;
    ray[3] = r.direction.x; ray[4] = r.direction.y; ray[5] = r.direction.z;
}

void orc_render_iterate(ORender *R, int iter, float *image, int shardRank, int shardCount,
                        OCounters *counters) {
    const int W = R->cam.resX, H = R->cam.resY;
    OCounters local;
    memset(&local, 0, sizeof local);
    for (int y = shardRank; y < H; y += shardCount) {
        for (int x = 0; x < W; ++x) {
            int index = x + y * W;
            Ray ray = camera_ray(*R, iter, index);
            V3 color = v3(1, 1, 1);
            bool alive = true;
            /* direct lighting: the last bounce aims its diffuse scatter at a light and one more bounce collects it */
            const int nb = R->traceDepth + (R->directLighting ? 1 : 0);
            for (int d = 1; d <= nb && alive; ++d) {
                if (d < 64) local.live[d]++;
                V3 contrib = v3(0, 0, 0);
                Fate f = bounce(*R, iter, index, d, ray, color, contrib, R->directLighting && d == R->traceDepth);
                if (f == MISS) {
                    local.misses++;
                    alive = false;
                } else if (f == LIGHT) {
                    image[3 * index + 0] += contrib.x;
                    image[3 * index + 1] += contrib.y;
                    image[3 * index + 2] += contrib.z;
                    local.lightHits++;
                    alive = false;
                }
            }
            if (alive) local.depthKilled++;
        }
    }
    if (counters) *counters = local;
}

int orc_render_dump_paths(ORender *R, int iter, int bounces, int shardRank, int shardCount,
                          float *origin3, float *dir3, float *color3, int *pixelIndex) {
    const int W = R->cam.resX, H = R->cam.resY;
    int n = 0;
    for (int y = shardRank; y < H; y += shardCount) {
        for (int x = 0; x < W; ++x) {
            int index = x + y * W;
            Ray ray = camera_ray(*R, iter, index);
            V3 color = v3(1, 1, 1);
            bool alive = true;
            for (int d = 1; d <= bounces && alive; ++d) {
                V3 contrib;
                alive = bounce(*R, iter, index, d, ray, color, contrib, R->directLighting && d == R->traceDepth) == ALIVE;
            }
            if (!alive) continue;
            origin3[3 * n] = ray.origin.x; origin3[3 * n + 1] = ray.origin.y; origin3[3 * n + 2] = ray.origin.z;
            dir3[3 * n] = ray.direction.x; dir3[3 * n + 1] = ray.direction.y; dir3[3 * n + 2] = ray.direction.z;
            color3[3 * n] = color.x; color3[3 * n + 1] = color.y; color3[3 * n + 2] = color.z;
            pixelIndex[n] = index;
            ++n;
        }
    }
    (void)shard_rows;
    return n;
}

/* src/pathtrace.cu:48-68 : clamp((int)(pix / iter * 255.0), 0, 255), fp64 multiply */
void orc_to_rgba8(const float *image, int npixels, int iter, uint8_t *rgba) {
    for (int i = 0; i < npixels; ++i) {
        for (int c = 0; c < 3; ++c) {
            int v = (int)(image[3 * i + c] / iter * 255.0);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            rgba[4 * i + c] = (uint8_t)v;
        }
        rgba[4 * i + 3] = 0;
    }
}

void orc_scan_exclusive_i32(const int32_t *in, int32_t *out, int64_t n) {
    int32_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t v = in[i];
        out[i] = acc;
        acc += v;
    }
}
int64_t orc_compact_nonzero_i32(const int32_t *in, int32_t *out, int64_t n) {
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i)
        if (in[i] != 0) out[k++] = in[i];
    return k;
}

}  // extern "C"
