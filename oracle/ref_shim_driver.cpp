/*
 * ref_shim_driver.cpp -- TEST INFRASTRUCTURE: proof of the drop-in claim at compile, link and run level.
 *
 * This translation unit and the product's shim (project3-cuda-path-tracer_amd/host/pathtrace_shim.cpp, UNCHANGED,
 * reached through a symlink so that its `#include "pathtrace.h"` resolves to the reference's header) are compiled
 * against the REFERENCE's own headers where they lie under /root/reference -- src/pathtrace.h:1-8, src/scene.h:13-26,
 * src/sceneStructs.h, the vendored glm -- together with the reference's own loader (src/scene.cpp, src/utilities.cpp),
 * and linked with libpt_amd.so.  Built only in the authoring container (oracle/Makefile, target `ref`); the binary
 * oracle/_ref/ref_shim_driver travels to the GPU box, the sources never do.
 *
 * main() follows runCuda() of src/main.cpp:72-113 call for call: the camera-change block (:73-86, the reference's own
 * glm arithmetic), Free -> Init whenever iteration == 0 (:91-94), pathtrace(pbo, 0, ++iteration) (:96-106), and at
 * the end saveImage's input -- here written raw -- then pathtraceFree() (:107-112).
 *
 *   ref_shim_driver <scene> <iterations> <out.bin> [<move_after> <dx> <dy> <dz> <theta> <phi>]
 * out.bin = Camera (52 B) + W*H glm::vec3 (the un-normalised running sum of scene->state.image).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pathtrace.h"            /* the reference's: /root/reference/src/pathtrace.h */
#include <glm/gtx/transform.hpp>

Scene::~Scene() {}                /* declared at src/scene.h:21, defined nowhere in the reference */

static Scene *scene;
static RenderState *renderState;
static int iteration;
static bool camchanged = false;
static float theta, phi;
static glm::vec3 cammove;

static bool runCuda(uchar4 *pbo_dptr) {
    if (camchanged) {             /* src/main.cpp:73-86 */
        iteration = 0;
        Camera &cam = renderState->camera;
        glm::vec3 v = cam.view;
        glm::vec3 u = cam.up;
        glm::vec3 r = glm::cross(v, u);
        glm::mat4 rotmat = glm::rotate(theta, r) * glm::rotate(phi, u);
        cam.view = glm::vec3(rotmat * glm::vec4(v, 0.f));
        cam.up = glm::vec3(rotmat * glm::vec4(u, 0.f));
        cam.position += cammove.x * r + cammove.y * u + cammove.z * v;
        theta = phi = 0;
        cammove = glm::vec3();
        camchanged = false;
    }
    if (iteration == 0) {         /* :91-94 */
        pathtraceFree();
        pathtraceInit(scene);
    }
    if (iteration < (int)renderState->iterations) {   /* :96-106 */
        iteration++;
        int frame = 0;
        pathtrace(pbo_dptr, frame, iteration);
        return true;
    }
    return false;                 /* :107-112: saveImage(); pathtraceFree(); exit */
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s scene iterations out.bin [move_after dx dy dz theta phi]\n", argv[0]);
        return 2;
    }
    scene = new Scene(argv[1]);   /* the reference's loader, src/scene.cpp:7-33 */
    renderState = &scene->state;
    renderState->iterations = (unsigned)atoi(argv[2]);
    const int moveAfter = argc >= 10 ? atoi(argv[4]) : -1;
    iteration = 0;
    int calls = 0;
    while (runCuda(NULL)) {
        ++calls;
        if (calls == moveAfter) { /* what keyCallback + mousePositionCallback set, src/main.cpp:115-137 */
            cammove = glm::vec3((float)atof(argv[5]), (float)atof(argv[6]), (float)atof(argv[7]));
            theta = (float)atof(argv[8]);
            phi = (float)atof(argv[9]);
            camchanged = true;
        }
    }
    FILE *f = fopen(argv[3], "wb");
    if (!f) return 3;
    fwrite(&renderState->camera, sizeof(Camera), 1, f);
    fwrite(renderState->image.data(), sizeof(glm::vec3), renderState->image.size(), f);
    fclose(f);
    printf("ref_shim_driver: %d pathtrace calls, %d x %d, last iteration %d\n", calls, renderState->camera.resolution.x,
           renderState->camera.resolution.y, iteration);
    pathtraceFree();
    return 0;
}
