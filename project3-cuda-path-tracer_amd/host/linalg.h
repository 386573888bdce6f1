// linalg.h -- the handful of glm 0.9.6.3 operations the scene loader needs, with glm's exact
// operation order (so matrices are bit-identical to what the reference's loader builds):
//   mat4 * mat4            glm/detail/type_mat4x4.inl:686-704
//   translate/rotate/scale glm/gtc/matrix_transform.inl:40-50, 52-86, 122-134
//   inverse                glm/detail/type_mat4x4.inl:37-91
//   inverseTranspose       glm/gtc/matrix_inverse.inl:95-158
//   normalize              glm/detail/func_geometric.inl:154-159
// Compile with -ffp-contract=off.
#pragma once
#include <cmath>

namespace lin {

struct ivec2 { int x, y; };
struct vec2 { float x, y; };

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 operator*(float s, const vec3 &v) { return vec3(s * v.x, s * v.y, s * v.z); }
inline vec3 operator*(const vec3 &v, float s) { return vec3(v.x * s, v.y * s, v.z * s); }
inline vec3 operator/(const vec3 &v, float s) { return vec3(v.x / s, v.y / s, v.z / s); }
inline float dot(const vec3 &a, const vec3 &b) { vec3 t(a.x * b.x, a.y * b.y, a.z * b.z); return t.x + t.y + t.z; }
inline vec3 normalize(const vec3 &v) { return v * (1.0f / std::sqrt(dot(v, v))); }

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    float operator[](int i) const { return (&x)[i]; }
    float &operator[](int i) { return (&x)[i]; }
};
inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4 &a, const vec4 &b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator/(const vec4 &a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }

struct mat4 {
    vec4 col[4];   // column-major like glm: m[c][r]
    mat4() { col[0] = vec4(1, 0, 0, 0); col[1] = vec4(0, 1, 0, 0); col[2] = vec4(0, 0, 1, 0); col[3] = vec4(0, 0, 0, 1); }
    const vec4 &operator[](int c) const { return col[c]; }
    vec4 &operator[](int c) { return col[c]; }
};
static_assert(sizeof(mat4) == 64 && sizeof(vec3) == 12, "glm-compatible layout");

inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int j = 0; j < 4; ++j)
        r[j] = a[0] * b[j][0] + a[1] * b[j][1] + a[2] * b[j][2] + a[3] * b[j][3];
    return r;
}

inline mat4 translate(const mat4 &m, const vec3 &v) {
    mat4 r(m);
    r[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3];
    return r;
}

inline mat4 rotate(const mat4 &m, float angle, const vec3 &v) {
    const float a = angle;
    const float c = std::cos(a);
    const float s = std::sin(a);
    vec3 axis(normalize(v));
    vec3 temp((1.0f - c) * axis);
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    mat4 r;
    r[0] = m[0] * R[0][0] + m[1] * R[0][1] + m[2] * R[0][2];
    r[1] = m[0] * R[1][0] + m[1] * R[1][1] + m[2] * R[1][2];
    r[2] = m[0] * R[2][0] + m[1] * R[2][1] + m[2] * R[2][2];
    r[3] = m[3];
    return r;
}

inline mat4 scale(const mat4 &m, const vec3 &v) {
    mat4 r;
    r[0] = m[0] * v[0];
    r[1] = m[1] * v[1];
    r[2] = m[2] * v[2];
    r[3] = m[3];
    return r;
}

inline mat4 inverse(const mat4 &m) {
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
    vec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]);
    vec4 v1(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]);
    vec4 v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    vec4 i0(v1 * f0 - v2 * f1 + v3 * f2);
    vec4 i1(v0 * f0 - v2 * f3 + v3 * f4);
    vec4 i2(v0 * f1 - v1 * f3 + v3 * f5);
    vec4 i3(v0 * f2 - v1 * f4 + v2 * f5);
    vec4 sA(+1, -1, +1, -1), sB(-1, +1, -1, +1);
    mat4 inv;
    inv[0] = i0 * sA; inv[1] = i1 * sB; inv[2] = i2 * sA; inv[3] = i3 * sB;
    vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    vec4 d0(m[0] * row0);
    float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    float ood = 1.0f / d1;
    mat4 r;
    for (int c = 0; c < 4; ++c) r[c] = inv[c] * ood;
    return r;
}

inline mat4 inverseTranspose(const mat4 &m) {
    float s00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float s01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float s02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float s03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float s04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float s05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float s06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float s07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float s08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float s09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float s10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float s11 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float s12 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float s13 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float s14 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float s15 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float s16 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float s17 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float s18 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    mat4 inv;
    inv[0][0] = +(m[1][1] * s00 - m[1][2] * s01 + m[1][3] * s02);
    inv[0][1] = -(m[1][0] * s00 - m[1][2] * s03 + m[1][3] * s04);
    inv[0][2] = +(m[1][0] * s01 - m[1][1] * s03 + m[1][3] * s05);
    inv[0][3] = -(m[1][0] * s02 - m[1][1] * s04 + m[1][2] * s05);
    inv[1][0] = -(m[0][1] * s00 - m[0][2] * s01 + m[0][3] * s02);
    inv[1][1] = +(m[0][0] * s00 - m[0][2] * s03 + m[0][3] * s04);
    inv[1][2] = -(m[0][0] * s01 - m[0][1] * s03 + m[0][3] * s05);
    inv[1][3] = +(m[0][0] * s02 - m[0][1] * s04 + m[0][2] * s05);
    inv[2][0] = +(m[0][1] * s06 - m[0][2] * s07 + m[0][3] * s08);
    inv[2][1] = -(m[0][0] * s06 - m[0][2] * s09 + m[0][3] * s10);
    inv[2][2] = +(m[0][0] * s11 - m[0][1] * s09 + m[0][3] * s12);
    inv[2][3] = -(m[0][0] * s08 - m[0][1] * s10 + m[0][2] * s12);
    inv[3][0] = -(m[0][1] * s13 - m[0][2] * s14 + m[0][3] * s15);
    inv[3][1] = +(m[0][0] * s13 - m[0][2] * s16 + m[0][3] * s17);
    inv[3][2] = -(m[0][0] * s14 - m[0][1] * s16 + m[0][3] * s18);
    inv[3][3] = +(m[0][0] * s15 - m[0][1] * s17 + m[0][2] * s18);
    float det = +m[0][0] * inv[0][0] + m[0][1] * inv[0][1] + m[0][2] * inv[0][2] + m[0][3] * inv[0][3];
    for (int c = 0; c < 4; ++c) inv[c] = inv[c] / det;
    return inv;
}

}  // namespace lin
