// scene.h -- the reference's `Scene` (src/scene.h:13-26): text scene file -> geoms, materials, state.
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "sceneStructs.h"
#include "utilities.h"

class Scene {
private:
    std::ifstream fp_in;                          // the scene file while it is being parsed
    int loadMaterial(std::string materialid);     // MATERIAL block: exactly 7 keyword lines
    int loadGeom(std::string objectid);           // OBJECT block: type, material, TRANS/ROTAT/SCALE up to a blank line
    int loadCamera();                             // CAMERA block: 5 keyword lines + EYE/VIEW/UP up to a blank line
    bool verbose;

public:
    // Throws std::runtime_error when the file cannot be opened (the reference prints
    // "Error reading from file - aborting!" and terminates, src/scene.cpp:12-15).
    explicit Scene(std::string filename, bool verbose = false);
    ~Scene();

    // RES override used by the headless driver; recomputes fov.x like src/scene.cpp:133-136
    void setResolution(int w, int h);

    std::vector<Geom> geoms;            // in file order = intersection order (first geom wins distance ties)
    std::vector<Material> materials;    // indexed by Geom::materialid
    RenderState state;                  // camera, iteration count, depth, output name, host image
};
