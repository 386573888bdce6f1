// main.cpp -- headless driver: the reference's main()/runCuda()/saveImage() without GLFW/GL
// (reference src/main.cpp:21-113).  Same protocol: Free -> Init when iteration == 0, 1-based
// iteration passed to pathtrace(pbo, 0, iteration), save + Free when the sample count is reached,
// output name <FILE>.<start time>.<N>samp.png, X mirrored, divided by the sample count.
//
//   pt_render SCENEFILE.txt [--res W H] [--iterations N] [--depth D] [--out BASENAME] [--hdr] [--batch B]
//                           [--lens RADIUS FOCALDISTANCE] [--direct]
// --lens / --direct switch on the README extras (depth of field, README.md:100-101; direct lighting, :107-108);
// imperfect specular needs no switch, it is a material's SPECEX > 0 in the scene file (README.md:171-185).
//
// --batch B (B > 1) leaves the reference protocol where nothing can observe it: iterations are traced B at a time
// through the C ABI (pt_iterate_batch) and the running sum is copied to the host once, before the image is saved,
// instead of after every iteration.  Same pixels (bit for bit), an order of magnitude less wall time.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <sstream>
#include <stdexcept>

#include "image.h"
#include "pathtrace.h"
#include "pt_amd.h"

static std::string startTimeString;
static Scene *scene;
static RenderState *renderState;
static int iteration;
static int width, height;
static std::string outBase;
static bool writeHdr = false;
static int batch = 1;
static float lensRadius = 0.0f, focalDistance = 0.0f;
static bool directLighting = false;
void pathtraceExtras(float lensRadius, float focalDistance, bool directLighting);   // pathtrace_shim.cpp

static std::string currentTimeString() {
    time_t now;
    time(&now);
    char buf[sizeof "0000-00-00_00-00-00z"];
    strftime(buf, sizeof buf, "%Y-%m-%d_%H-%M-%Sz", gmtime(&now));
    return std::string(buf);
}

static void saveImage() {
    float samples = iteration;
    image img(width, height);
    for (int x = 0; x < width; x++) {
        for (int y = 0; y < height; y++) {
            int index = x + (y * width);
            lin::vec3 pix = renderState->image[index];
            img.setPixel(width - 1 - x, y, pix / samples);   // X mirror, src/main.cpp:58
        }
    }
    std::ostringstream ss;
    if (outBase.empty()) ss << renderState->imageName << "." << startTimeString << "." << samples << "samp";
    else ss << outBase;
    img.savePNG(ss.str());
    if (writeHdr) img.saveHDR(ss.str());   // the reference keeps this behind a comment, src/main.cpp:69
}

// one trip through the reference's per-frame function; returns false when rendering is complete
static bool runHip() {
    if (iteration == 0) {
        pathtraceFree();
        pathtraceInit(scene);
    }
    if (iteration < (int)renderState->iterations) {
        iteration++;
        pathtrace(NULL, 0, iteration);   // headless: no PBO
        return true;
    }
    saveImage();
    pathtraceFree();
    return false;
}

// --batch: pt_init / pt_iterate_batch / pt_readback directly; exits like checkCUDAError on a failure
static void check(int status, const char *what) {
    if (status == PT_OK) return;
    fprintf(stderr, "HIP error (main.cpp): %s: %s\n", what, pt_last_error());
    exit(EXIT_FAILURE);
}
static void renderBatched() {
    PtOptions opt;
    memset(&opt, 0, sizeof opt);
    opt.shard_count = 1;
    opt.device = -1;
    opt.max_batch = batch;
    opt.pipeline_depth = 2;        // two batches in flight are as fast as three and provision a third less memory
    opt.lens_radius = lensRadius;
    opt.focal_distance = focalDistance;
    if (directLighting) opt.flags |= PT_FLAG_DIRECT_LIGHTING;
    std::vector<PtMesh> meshes;        // `mesh` objects: their triangles first (like pathtrace_shim.cpp)
    for (size_t i = 0; i < scene->meshes.size(); ++i) {
        PtMesh m;
        m.geom = scene->meshes[i].geom;
        m.ntris = (int)(scene->meshes[i].tris.size() / 9);
        m.tris = scene->meshes[i].tris.data();
        m.normals = scene->meshes[i].normals.empty() ? NULL : scene->meshes[i].normals.data();      // `vn`: smooth shading
        m.materials = scene->meshes[i].mats.empty() ? NULL : scene->meshes[i].mats.data();          // `usemtl <k>`: a material per face
        meshes.push_back(m);
    }
    check(pt_set_meshes(meshes.empty() ? NULL : meshes.data(), (int)meshes.size()), "pt_set_meshes");
    check(pt_init((const PtCamera *)&renderState->camera, (const PtGeom *)scene->geoms.data(), (int)scene->geoms.size(),
                  (const PtMaterial *)scene->materials.data(), (int)scene->materials.size(), renderState->traceDepth, &opt),
          "pt_init");
    const int total = (int)renderState->iterations;
    while (iteration < total) {
        const int n = total - iteration < batch ? total - iteration : batch;
        check(pt_iterate_batch(0, iteration + 1, n, NULL), "pt_iterate_batch");
        iteration += n;
    }
    check(pt_readback((float *)renderState->image.data()), "pt_readback");
    saveImage();
    pt_free();
}

int main(int argc, char **argv) {
    startTimeString = currentTimeString();
    if (argc < 2) {
        printf("Usage: %s SCENEFILE.txt [--res W H] [--iterations N] [--depth D] [--out BASENAME] [--hdr] [--batch B] [--lens R F] [--direct]\n", argv[0]);
        return 1;
    }
    try {
        scene = new Scene(argv[1], true);
    } catch (const std::exception &e) {          // unreadable scene file, or a mesh object whose OBJ cannot be read
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    renderState = &scene->state;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--res") && i + 2 < argc) { scene->setResolution(atoi(argv[i + 1]), atoi(argv[i + 2])); i += 2; }
        else if (!strcmp(argv[i], "--iterations") && i + 1 < argc) renderState->iterations = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) renderState->traceDepth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outBase = argv[++i];
        else if (!strcmp(argv[i], "--hdr")) writeHdr = true;
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--lens") && i + 2 < argc) { lensRadius = (float)atof(argv[i + 1]); focalDistance = (float)atof(argv[i + 2]); i += 2; }
        else if (!strcmp(argv[i], "--direct")) directLighting = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 1; }
    }
    pathtraceExtras(lensRadius, focalDistance, directLighting);
    iteration = 0;
    width = renderState->camera.resolution.x;
    height = renderState->camera.resolution.y;
    const auto t0 = std::chrono::steady_clock::now();
    if (batch > 1) renderBatched();
    else while (runHip()) {}
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d iterations of %dx%d, depth %d: %.3f s wall (init, %s and PNG included)\n", iteration, width, height,
           renderState->traceDepth, s, batch > 1 ? "one D2H copy" : "per-iteration D2H copy");
    delete scene;
    return 0;
}
