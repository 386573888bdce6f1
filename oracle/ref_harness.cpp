/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" wrapper around the REFERENCE's own
 * sources, compiled where they lie under /root/reference (never copied):
 *     src/intersections.h, src/sceneStructs.h, src/utilities.{h,cpp}, src/scene.{h,cpp},
 *     external/include/glm (vendored glm 0.9.6.3; glm/gtx/intersect.inl for the triangle test README.md:116 names)
 * Built only in the authoring container by oracle/Makefile (target _ref) with plain g++;
 * <cuda_runtime.h> (src/sceneStructs.h:5) resolves to the real CUDA runtime header that ships
 * in this image with triton (no stand-in headers are written).  Output: oracle/_ref/libptref.so.
 *
 * Not buildable here (recorded in DESIGN.md): src/interactions.h (needs thrust, which this image
 * only has as rocThrust = hipcc-only, and that cannot be combined with the CUDA runtime header),
 * src/pathtrace.cu (nvcc).  Those are pinned by SURVEY section 8a KATs and rocThrust (rng).
 *
 * The `using` declarations restore the float overloads nvcc would pick for the unqualified
 * min/max/sqrt/pow calls in src/intersections.h:113,118,127,130 ("overload hazards").
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
using std::abs;
using std::max;
using std::min;
using std::sqrt;
/* nvcc resolves pow(float, int) to a float overload; C++11 std::pow(float,int) promotes to double.
 * Keep the reference's float semantics for src/intersections.h:113. */
#ifndef REF_HOST_DOUBLE_POW
static inline float pow(float a, int b) { return b == 2 ? a * a : (float)std::pow((double)a, (double)b); }
#else
using std::pow; /* variant build: C++11 std::pow(float,int) -> double (what a non-nvcc host compiler does) */
#endif

#include "intersections.h"
#include "scene.h"
#include <glm/gtc/matrix_inverse.hpp>
#include <glm/gtx/intersect.hpp>

static_assert(sizeof(Ray) == 24 && sizeof(Geom) == 236 && sizeof(Material) == 44 && sizeof(Camera) == 52,
              "reference layout (SURVEY section 7 step 1)");

static Ray make_ray(const float *r) {
    Ray q;
    q.origin = glm::vec3(r[0], r[1], r[2]);
    q.direction = glm::vec3(r[3], r[4], r[5]);
    return q;
}
static void put3(float *o, const glm::vec3 &v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

extern "C" {

unsigned ref_utilhash(unsigned a) { return utilhash(a); }

void ref_point_on_ray(const float *ray, float t, float *out) { put3(out, getPointOnRay(make_ray(ray), t)); }

void ref_mulmv(const float *m, const float *v, float *out) {
    glm::mat4 M;
    memcpy(&M, m, 64);
    put3(out, multiplyMV(M, glm::vec4(v[0], v[1], v[2], v[3])));
}

float ref_box(const void *geom236, const float *ray, float *p, float *n, int *outside) {
    Geom g;
    memcpy(&g, geom236, sizeof g);
    glm::vec3 P(p[0], p[1], p[2]), N(n[0], n[1], n[2]);
    bool o = *outside != 0;
    float t = boxIntersectionTest(g, make_ray(ray), P, N, o);
    put3(p, P); put3(n, N);
    *outside = o;
    return t;
}

float ref_sphere(const void *geom236, const float *ray, float *p, float *n, int *outside) {
    Geom g;
    memcpy(&g, geom236, sizeof g);
    glm::vec3 P(p[0], p[1], p[2]), N(n[0], n[1], n[2]);
    bool o = *outside != 0;
    float t = sphereIntersectionTest(g, make_ray(ray), P, N, o);
    put3(p, P); put3(n, N);
    *outside = o;
    return t;
}

void ref_normalize(const float *v, float *out) { put3(out, glm::normalize(glm::vec3(v[0], v[1], v[2]))); }
void ref_cross(const float *a, const float *b, float *out) {
    put3(out, glm::cross(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2])));
}
float ref_dot(const float *a, const float *b) {
    return glm::dot(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}
float ref_length(const float *a) { return glm::length(glm::vec3(a[0], a[1], a[2])); }
void ref_reflect(const float *I, const float *N, float *out) {
    put3(out, glm::reflect(glm::vec3(I[0], I[1], I[2]), glm::vec3(N[0], N[1], N[2])));
}
void ref_refract(const float *I, const float *N, float eta, float *out) {
    put3(out, glm::refract(glm::vec3(I[0], I[1], I[2]), glm::vec3(N[0], N[1], N[2]), eta));
}

/* glm/gtx/intersect.inl:36-72, the triangle test README.md:116 points mesh loaders to */
int ref_intersect_ray_triangle(const float *o, const float *d, const float *v0, const float *v1, const float *v2, float *bary) {
    glm::vec3 b(bary[0], bary[1], bary[2]);
    bool hit = glm::intersectRayTriangle(glm::vec3(o[0], o[1], o[2]), glm::vec3(d[0], d[1], d[2]), glm::vec3(v0[0], v0[1], v0[2]),
                                         glm::vec3(v1[0], v1[1], v1[2]), glm::vec3(v2[0], v2[1], v2[2]), b);
    put3(bary, b);
    return hit ? 1 : 0;
}

/* src/utilities.cpp:65-72 + src/scene.cpp:82-85 */
void ref_build_transform(const float *t, const float *r, const float *s, float *transform, float *inverse,
                         float *invTranspose) {
    glm::mat4 xf = utilityCore::buildTransformationMatrix(glm::vec3(t[0], t[1], t[2]), glm::vec3(r[0], r[1], r[2]),
                                                          glm::vec3(s[0], s[1], s[2]));
    glm::mat4 inv = glm::inverse(xf);
    glm::mat4 it = glm::inverseTranspose(xf);
    memcpy(transform, &xf, 64);
    memcpy(inverse, &inv, 64);
    memcpy(invTranspose, &it, 64);
}

/* src/scene.cpp:7-182.  Scene::~Scene is declared (src/scene.h:21) but defined nowhere in the
 * reference, so the object is intentionally never destroyed. */
void *ref_scene_load(const char *path) {
    std::streambuf *old = std::cout.rdbuf();
    std::ostringstream sink;
    std::cout.rdbuf(sink.rdbuf()); /* the loader is chatty (scene.cpp:8,41,49,...) */
    Scene *s = new Scene(path);
    std::cout.rdbuf(old);
    return s;
}
int ref_scene_num_geoms(void *s) { return (int)((Scene *)s)->geoms.size(); }
int ref_scene_num_materials(void *s) { return (int)((Scene *)s)->materials.size(); }
void ref_scene_copy_geoms(void *s, void *out) {
    Scene *sc = (Scene *)s;
    memcpy(out, sc->geoms.data(), sc->geoms.size() * sizeof(Geom));
}
void ref_scene_copy_materials(void *s, void *out) {
    Scene *sc = (Scene *)s;
    memcpy(out, sc->materials.data(), sc->materials.size() * sizeof(Material));
}
void ref_scene_copy_camera(void *s, void *out) { memcpy(out, &((Scene *)s)->state.camera, sizeof(Camera)); }
int ref_scene_iterations(void *s) { return (int)((Scene *)s)->state.iterations; }
int ref_scene_depth(void *s) { return ((Scene *)s)->state.traceDepth; }
int ref_scene_image_len(void *s) { return (int)((Scene *)s)->state.image.size(); }
const char *ref_scene_image_name(void *s) { return ((Scene *)s)->state.imageName.c_str(); }

}  // extern "C"
