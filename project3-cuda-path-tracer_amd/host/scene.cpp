// scene.cpp -- parser for the reference's scene-file format (reference src/scene.cpp:7-182), quirks kept:
//   * MATERIAL / OBJECT ids must be sequential, otherwise the block is skipped with an ERROR line (:37,:149)
//   * a MATERIAL block is exactly 7 lines, a CAMERA block 5 lines + EYE/VIEW/UP lines up to a blank line
//   * the object type line must be exactly "sphere" or "cube" (:48-53); TRANS/ROTAT/SCALE up to a blank line
//   * README.md:236 names a third object type, "mesh", which the reference's loader does not know: here the type line
//     `mesh <file.obj>` (path relative to the scene file) loads a Wavefront OBJ into Scene::meshes (loadObj below)
//   * the specular exponent keyword is SPECEX (:164) although the README says SPECX
//   * CRLF / CR / LF line ends; '//' comment lines are simply unknown keywords
// Unlike the reference, fields whose keyword is missing are zero instead of uninitialised.
#include "scene.h"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>

using utilityCore::safeGetline;
using utilityCore::tokenizeString;

namespace {
typedef std::vector<std::string> Tokens;
bool key(const Tokens &t, const char *k) { return !t.empty() && t[0] == k; }
float num(const Tokens &t, size_t i) { return i < t.size() ? (float)atof(t[i].c_str()) : 0.0f; }
lin::vec3 triple(const Tokens &t) { return lin::vec3(num(t, 1), num(t, 2), num(t, 3)); }

// Wavefront OBJ -> triangle soup.  `v x y z` and `f a b c ...` with a = i, i/j, i//k or i/j/k (1-based; negative = relative
// to the vertices read so far); polygons are fanned from their first vertex.  `vn x y z` + the third field of a face corner give
// vertex normals: kept only when EVERY corner of EVERY face names one (else the mesh is shaded flat).  `usemtl <k>` (k an integer
// = a MATERIAL of the scene file) gives the faces that follow a material of their own; any other name returns to the object's.
// Every other statement is ignored.
bool loadObj(const std::string &path, std::vector<float> &tris, std::vector<float> &normalsOut, std::vector<int> &matsOut) {
    std::ifstream fp(path.c_str());
    if (!fp.is_open()) return false;
    std::vector<float> verts, vnorm, normals;
    std::vector<int> mats;
    bool allNormals = true, anyMat = false;
    int curMat = -1;
    std::string line;
    while (std::getline(fp, line)) {
        std::istringstream ss(line);
        std::string keyword;
        if (!(ss >> keyword)) continue;
        if (keyword == "v" || keyword == "vn") {
            double c[3] = {0, 0, 0};
            ss >> c[0] >> c[1] >> c[2];
            std::vector<float> &dst = keyword == "v" ? verts : vnorm;
            for (int a = 0; a < 3; ++a) dst.push_back((float)c[a]);
        } else if (keyword == "usemtl") {
            std::string tok;
            curMat = -1;
            if (ss >> tok) {
                char *end = nullptr;
                const long k = strtol(tok.c_str(), &end, 10);
                if (end && *end == 0 && k >= 0 && k < 1000000) curMat = (int)k;
            }
        } else if (keyword == "f") {
            const int nv = (int)(verts.size() / 3), nn = (int)(vnorm.size() / 3);
            std::vector<int> corner, ncorner;
            std::string ref;
            bool ok = true;
            while (ss >> ref) {
                const int i = atoi(ref.c_str());              // (reads up to the first '/')
                const int k = i > 0 ? i - 1 : nv + i;
                if (i == 0 || k < 0 || k >= nv) { ok = false; break; }
                corner.push_back(k);
                int nk = -1;                                   // the normal: the field behind the second '/'
                const size_t s1 = ref.find('/');
                const size_t s2 = s1 == std::string::npos ? std::string::npos : ref.find('/', s1 + 1);
                if (s2 != std::string::npos && s2 + 1 < ref.size()) {
                    const int j = atoi(ref.c_str() + s2 + 1);
                    const int q = j > 0 ? j - 1 : nn + j;
                    if (j != 0 && q >= 0 && q < nn) nk = q;
                }
                ncorner.push_back(nk);
            }
            for (size_t k = 2; ok && k < corner.size(); ++k) {
                const int tri[3] = {corner[0], corner[k - 1], corner[k]};
                const int ntri[3] = {ncorner[0], ncorner[k - 1], ncorner[k]};
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
    if (allNormals && !tris.empty()) normalsOut = normals; else normalsOut.clear();
    if (anyMat) matsOut = mats; else matsOut.clear();
    return true;
}

void deriveFov(Camera &camera, float fovy) {
    // reference src/scene.cpp:133-136
    float yscaled = std::tan(fovy * (PI / 180));
    float xscaled = (yscaled * camera.resolution.x) / camera.resolution.y;
    float fovx = (std::atan(xscaled) * 180) / PI;
    camera.fov.x = fovx;
    camera.fov.y = fovy;
}
}  // namespace

Scene::Scene(std::string filename, bool verbose_) : verbose(verbose_) {
    if (verbose) std::cout << "Reading scene from " << filename << " ..." << std::endl << " " << std::endl;
    state.iterations = 0;
    state.traceDepth = 0;
    state.camera.resolution.x = state.camera.resolution.y = 0;
    state.camera.fov.x = state.camera.fov.y = 0.0f;
    {
        const size_t slash = filename.find_last_of('/');
        dir = slash == std::string::npos ? std::string() : filename.substr(0, slash + 1);
    }
    fp_in.open(filename.c_str());
    if (!fp_in.is_open()) {
        std::cout << "Error reading from file - aborting!" << std::endl;
        throw std::runtime_error("Error reading from file - aborting!");
    }
    std::string line;
    while (fp_in.good()) {
        safeGetline(fp_in, line);
        if (line.empty()) continue;
        const Tokens tokens = tokenizeString(line);
        bool block = true;
        if (tokens.size() >= 2 && key(tokens, "MATERIAL")) loadMaterial(tokens[1]);
        else if (tokens.size() >= 2 && key(tokens, "OBJECT")) loadGeom(tokens[1]);
        else if (key(tokens, "CAMERA")) loadCamera();
        else block = false;
        if (block && verbose) std::cout << " " << std::endl;
    }
    // `usemtl <k>` names a MATERIAL of THIS file.  An OBJ from elsewhere may carry numeric material names of its own: an index the
    // scene does not have falls back to the object's material, loudly (it used to surface only later, as pt_init's PT_ERR_INVALID).
    for (size_t i = 0; i < meshes.size(); ++i) {
        size_t bad = 0;
        for (size_t f = 0; f < meshes[i].mats.size(); ++f)
            if (meshes[i].mats[f] >= (int)materials.size()) {
                meshes[i].mats[f] = -1;
                ++bad;
            }
        if (bad)
            std::cout << "WARNING: mesh of object " << meshes[i].geom << ": " << bad << " faces name a material (usemtl <k>) the scene has "
                      << "not (" << materials.size() << " MATERIAL blocks): they take the object's material" << std::endl;
    }
}

Scene::~Scene() {}

int Scene::loadMaterial(std::string materialid) {
    if (atoi(materialid.c_str()) != (int)materials.size()) {
        std::cout << "ERROR: MATERIAL ID does not match expected number of materials" << std::endl;
        return -1;
    }
    if (verbose) std::cout << "Loading Material " << materials.size() << "..." << std::endl;
    Material m;
    m.specular.exponent = m.hasReflective = m.hasRefractive = m.indexOfRefraction = m.emittance = 0.0f;
    std::string line;
    for (int i = 0; i < 7; ++i) {
        safeGetline(fp_in, line);
        const Tokens t = tokenizeString(line);
        if (key(t, "RGB")) m.color = triple(t);
        else if (key(t, "SPECEX")) m.specular.exponent = num(t, 1);
        else if (key(t, "SPECRGB")) m.specular.color = triple(t);
        else if (key(t, "REFL")) m.hasReflective = num(t, 1);
        else if (key(t, "REFR")) m.hasRefractive = num(t, 1);
        else if (key(t, "REFRIOR")) m.indexOfRefraction = num(t, 1);
        else if (key(t, "EMITTANCE")) m.emittance = num(t, 1);
    }
    materials.push_back(m);
    return 1;
}

int Scene::loadCamera() {
    if (verbose) std::cout << "Loading Camera ..." << std::endl;
    Camera &camera = state.camera;
    float fovy = 0;
    std::string line;
    for (int i = 0; i < 5; ++i) {
        safeGetline(fp_in, line);
        const Tokens t = tokenizeString(line);
        if (key(t, "RES") && t.size() >= 3) {
            camera.resolution.x = atoi(t[1].c_str());
            camera.resolution.y = atoi(t[2].c_str());
        } else if (key(t, "FOVY")) fovy = num(t, 1);
        else if (key(t, "ITERATIONS") && t.size() >= 2) state.iterations = atoi(t[1].c_str());
        else if (key(t, "DEPTH") && t.size() >= 2) state.traceDepth = atoi(t[1].c_str());
        else if (key(t, "FILE") && t.size() >= 2) state.imageName = t[1];
    }
    for (safeGetline(fp_in, line); !line.empty() && fp_in.good(); safeGetline(fp_in, line)) {
        const Tokens t = tokenizeString(line);
        if (key(t, "EYE")) camera.position = triple(t);
        else if (key(t, "VIEW")) camera.view = triple(t);
        else if (key(t, "UP")) camera.up = triple(t);
    }
    deriveFov(camera, fovy);
    state.image.assign((size_t)camera.resolution.x * camera.resolution.y, lin::vec3());
    if (verbose) std::cout << "Loaded camera!" << std::endl;
    return 1;
}

int Scene::loadGeom(std::string objectid) {
    if (atoi(objectid.c_str()) != (int)geoms.size()) {
        std::cout << "ERROR: OBJECT ID does not match expected number of geoms" << std::endl;
        return -1;
    }
    if (verbose) std::cout << "Loading Geom " << geoms.size() << "..." << std::endl;
    Geom g;
    g.type = SPHERE;
    g.materialid = 0;
    std::string line;
    safeGetline(fp_in, line);
    if (!line.empty() && fp_in.good()) {
        if (line == "sphere") {
            if (verbose) std::cout << "Creating new sphere..." << std::endl;
            g.type = SPHERE;
        } else if (line == "cube") {
            if (verbose) std::cout << "Creating new cube..." << std::endl;
            g.type = CUBE;
        } else {
            const Tokens t = tokenizeString(line);
            if (t.size() >= 2 && t[0] == "mesh") {
                Mesh m;
                m.geom = (int)geoms.size();
                const std::string path = t[1][0] == '/' ? t[1] : dir + t[1];
                if (loadObj(path, m.tris, m.normals, m.mats) && !m.tris.empty()) {
                    if (verbose) std::cout << "Creating new mesh (" << m.tris.size() / 9 << " triangles)..." << std::endl;
                    g.type = MESH;
                    meshes.push_back(m);
                } else {
                    // a mesh that cannot be read is a scene that cannot be rendered -- not a unit sphere with the mesh's
                    // transform and material: fail the load like an unreadable scene file does
                    std::cout << "ERROR: cannot read triangles from " << path << " - aborting!" << std::endl;
                    throw std::runtime_error("cannot read triangles from " + path);
                }
            }
        }
    }
    safeGetline(fp_in, line);
    if (!line.empty() && fp_in.good()) {
        const Tokens t = tokenizeString(line);
        if (t.size() >= 2) g.materialid = atoi(t[1].c_str());
        if (verbose) std::cout << "Connecting Geom " << objectid << " to Material " << g.materialid << "..." << std::endl;
    }
    for (safeGetline(fp_in, line); !line.empty() && fp_in.good(); safeGetline(fp_in, line)) {
        const Tokens t = tokenizeString(line);
        if (key(t, "TRANS")) g.translation = triple(t);
        else if (key(t, "ROTAT")) g.rotation = triple(t);
        else if (key(t, "SCALE")) g.scale = triple(t);
    }
    g.transform = utilityCore::buildTransformationMatrix(g.translation, g.rotation, g.scale);
    g.inverseTransform = lin::inverse(g.transform);
    g.invTranspose = lin::inverseTranspose(g.transform);
    geoms.push_back(g);
    return 1;
}

void Scene::setResolution(int w, int h) {
    state.camera.resolution.x = w;
    state.camera.resolution.y = h;
    deriveFov(state.camera, state.camera.fov.y);
    state.image.assign((size_t)w * h, lin::vec3());
}
