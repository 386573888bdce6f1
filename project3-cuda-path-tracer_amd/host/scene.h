// scene.h -- the reference's `Scene` (src/scene.h:13-26): text scene file -> geoms, materials, state.
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "sceneStructs.h"
#include "utilities.h"

// (tells pathtrace_shim.cpp that this Scene has `meshes`; the reference's own scene.h, which the shim also builds against, has not)
#define PT_SCENE_HAS_MESHES 1

// Triangles of one `mesh` object (README.md:112-116, 236), in the OBJ file's coordinates = the geom's object space.
struct Mesh {
    int geom;                   // index into Scene::geoms (a Geom of type MESH)
    std::vector<float> tris;    // 9 floats per triangle: v0, v1, v2
    std::vector<float> normals; // 9 floats per triangle: the vertex normals n0, n1, n2 (`vn`, smooth shading) -- or EMPTY: flat shading
    std::vector<int> mats;      // one per triangle: the scene material of that face (`usemtl <k>`), -1 = the object's own -- or EMPTY
};

class Scene {
private:
    std::ifstream fp_in;                          // the scene file while it is being parsed
    int loadMaterial(std::string materialid);     // MATERIAL block: exactly 7 keyword lines
    int loadGeom(std::string objectid);           // OBJECT block: type, material, TRANS/ROTAT/SCALE up to a blank line
    int loadCamera();                             // CAMERA block: 5 keyword lines + EYE/VIEW/UP up to a blank line
    bool verbose;
    std::string dir;                              // directory of the scene file ("" or ending in '/'): mesh paths are relative to it

public:
    // Throws std::runtime_error when the file cannot be opened (the reference prints
    // "Error reading from file - aborting!" and terminates, src/scene.cpp:12-15).
    explicit Scene(std::string filename, bool verbose = false);
    ~Scene();

    // RES override used by the headless driver; recomputes fov.x like src/scene.cpp:133-136
    void setResolution(int w, int h);

    std::vector<Geom> geoms;            // in file order = intersection order (first geom wins distance ties)
    std::vector<Material> materials;    // indexed by Geom::materialid
    std::vector<Mesh> meshes;           // one per Geom of type MESH, file order
    RenderState state;                  // camera, iteration count, depth, output name, host image
};
