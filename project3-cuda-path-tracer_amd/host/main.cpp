// main.cpp -- headless driver: the reference's main()/runCuda()/saveImage() without GLFW/GL
// (reference src/main.cpp:21-113).  Same protocol: Free -> Init when iteration == 0, 1-based
// iteration passed to pathtrace(pbo, 0, iteration), save + Free when the sample count is reached,
// output name <FILE>.<start time>.<N>samp.png, X mirrored, divided by the sample count.
//
//   pt_render SCENEFILE.txt [--res W H] [--iterations N] [--depth D] [--out BASENAME] [--hdr]
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <sstream>

#include "image.h"
#include "pathtrace.h"

static std::string startTimeString;
static Scene *scene;
static RenderState *renderState;
static int iteration;
static int width, height;
static std::string outBase;
static bool writeHdr = false;

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

int main(int argc, char **argv) {
    startTimeString = currentTimeString();
    if (argc < 2) {
        printf("Usage: %s SCENEFILE.txt [--res W H] [--iterations N] [--depth D] [--out BASENAME] [--hdr]\n", argv[0]);
        return 1;
    }
    scene = new Scene(argv[1], true);
    renderState = &scene->state;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--res") && i + 2 < argc) { scene->setResolution(atoi(argv[i + 1]), atoi(argv[i + 2])); i += 2; }
        else if (!strcmp(argv[i], "--iterations") && i + 1 < argc) renderState->iterations = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) renderState->traceDepth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outBase = argv[++i];
        else if (!strcmp(argv[i], "--hdr")) writeHdr = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 1; }
    }
    iteration = 0;
    width = renderState->camera.resolution.x;
    height = renderState->camera.resolution.y;
    const auto t0 = std::chrono::steady_clock::now();
    while (runHip()) {}
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d iterations of %dx%d, depth %d: %.3f s wall (init, per-iteration D2H copy and PNG included)\n", iteration, width,
           height, renderState->traceDepth, s);
    delete scene;
    return 0;
}
