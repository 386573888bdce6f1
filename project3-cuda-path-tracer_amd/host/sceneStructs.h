// sceneStructs.h -- host-side scene PODs, layout-compatible with the reference's
// src/sceneStructs.h:8-55 (Ray 24 B, Geom 236 B, Material 44 B, Camera 52 B; mat4 column-major)
// so `scene->geoms.data()` etc. pass straight into the C ABI (include/pt_amd.h).
// Uses the mini vector library in linalg.h instead of glm; no CUDA/HIP header is needed on the host.
#pragma once

#include <string>
#include <vector>

#include "linalg.h"

enum GeomType {
    SPHERE,
    CUBE,
};

struct Ray {
    lin::vec3 origin;
    lin::vec3 direction;
};

struct Geom {
    enum GeomType type;
    int materialid;
    lin::vec3 translation;
    lin::vec3 rotation;
    lin::vec3 scale;
    lin::mat4 transform;
    lin::mat4 inverseTransform;
    lin::mat4 invTranspose;
};

struct Material {
    lin::vec3 color;
    struct {
        float exponent;
        lin::vec3 color;
    } specular;
    float hasReflective;
    float hasRefractive;
    float indexOfRefraction;
    float emittance;
};

struct Camera {
    lin::ivec2 resolution;
    lin::vec3 position;
    lin::vec3 view;
    lin::vec3 up;
    lin::vec2 fov;
};

struct RenderState {
    Camera camera;
    unsigned int iterations;
    int traceDepth;
    std::vector<lin::vec3> image;
    std::string imageName;
};

static_assert(sizeof(Ray) == 24 && sizeof(Geom) == 236 && sizeof(Material) == 44 && sizeof(Camera) == 52,
              "must stay byte-identical to the reference's sceneStructs.h");
