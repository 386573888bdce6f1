// sceneStructs.h -- host-side scene PODs of the MI355X build.
//
// Binary contract: every struct below has exactly the size and field offsets of its namesake in the
// reference's src/sceneStructs.h:8-55 (Ray 24 B, Geom 236 B, Material 44 B, Camera 52 B; mat4 is
// column-major), so `scene->geoms.data()`, `scene->materials.data()` and `&scene->state.camera` go
// straight into the C ABI (include/pt_amd.h: PtGeom / PtMaterial / PtCamera) without a copy.
// The byte offsets are written next to the fields and checked at compile time at the end of the file.
// Vector types come from the mini library in linalg.h instead of glm; no GPU header is needed on the host.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

#include "linalg.h"

enum GeomType {
    SPHERE,   // unit-diameter sphere at the origin of object space
    CUBE,     // unit cube [-0.5, 0.5]^3 in object space
    MESH,     // README.md:236 "mesh": a triangle soup in object space (Scene::meshes), not in the reference's enum
};

struct Ray {
    lin::vec3 origin;                 // @0
    lin::vec3 direction;              // @12
};

struct Geom {
    enum GeomType type;               // @0   (int-sized)
    int materialid;                   // @4   index into Scene::materials
    lin::vec3 translation;            // @8
    lin::vec3 rotation;               // @20  degrees, applied x then y then z
    lin::vec3 scale;                  // @32
    lin::mat4 transform;              // @44  object -> world
    lin::mat4 inverseTransform;       // @108 world -> object
    lin::mat4 invTranspose;           // @172 for normals
};

struct Material {
    lin::vec3 color;                  // @0   diffuse / transmission tint
    struct {
        float exponent;               // @12  parsed (SPECEX), unused by the renderer
        lin::vec3 color;              // @16  mirror tint
    } specular;
    float hasReflective;              // @28  > 0: mirror/diffuse mixture
    float hasRefractive;              // @32  > 0: dielectric
    float indexOfRefraction;          // @36
    float emittance;                  // @40  > 0: light source
};

struct Camera {
    lin::ivec2 resolution;            // @0
    lin::vec3 position;               // @8
    lin::vec3 view;                   // @20
    lin::vec3 up;                     // @32
    lin::vec2 fov;                    // @44  degrees; fov.y is the vertical HALF angle
};

struct RenderState {
    Camera camera;
    unsigned int iterations;          // samples per pixel to render
    int traceDepth;                   // bounces per path
    std::vector<lin::vec3> image;     // un-normalised running sum, index x + y * width
    std::string imageName;
};

static_assert(sizeof(Ray) == 24 && sizeof(Geom) == 236 && sizeof(Material) == 44 && sizeof(Camera) == 52,
              "sizes must stay identical to the reference's sceneStructs.h");
static_assert(offsetof(Geom, translation) == 8 && offsetof(Geom, transform) == 44 &&
              offsetof(Geom, inverseTransform) == 108 && offsetof(Geom, invTranspose) == 172,
              "Geom field offsets");
static_assert(offsetof(Material, specular) == 12 && offsetof(Material, hasReflective) == 28 &&
              offsetof(Material, emittance) == 40, "Material field offsets");
static_assert(offsetof(Camera, position) == 8 && offsetof(Camera, fov) == 44, "Camera field offsets");
