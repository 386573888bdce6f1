// pathtrace_shim.cpp -- the three reference symbols (src/pathtrace.cu:75-92,123-174) over the C ABI.
// Error behaviour follows checkCUDAError (src/pathtrace.cu:21-39): message on stderr, exit(EXIT_FAILURE).
#include "pathtrace.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pt_amd.h"

static Scene *hst_scene = NULL;
// PT_AMD_DEVICES=<n> (n >= 2): the frame's rows are rendered by n renderers side by side -- one per HIP device of the node, member i
// on device i % pt_device_count() -- behind the same three symbols (include/pt_amd.h: pt_group_*; the reference is hard-wired to
// device 0, src/preview.cpp:107).  The image after every call is the one-device image bit for bit.  Headless hosts only: a device
// `pbo` belongs to one device's GL context, which a group has none of.
static PtGroup *group = NULL;
static float extraLens = 0.0f, extraFocal = 0.0f;   // README extras (pathtraceExtras), off = the reference's renderer
static bool extraDirect = false;

static void checkPtError(int status, const char *msg) {
    if (status == PT_OK) return;
    fprintf(stderr, "HIP error (pathtrace_shim.cpp): %s: %s\n", msg, pt_last_error());
    exit(EXIT_FAILURE);
}

void pathtraceInit(Scene *scene) {
    hst_scene = scene;
    static_assert(sizeof(Geom) == sizeof(PtGeom) && sizeof(Material) == sizeof(PtMaterial) &&
                  sizeof(Camera) == sizeof(PtCamera), "layout contract of include/pt_amd.h");
    PtOptions opt = PtOptions();        // all defaults = the reference's behaviour ...
    opt.shard_count = 1;
    opt.device = -1;
    opt.lens_radius = extraLens;        // ... unless pathtraceExtras() switched a README extra on
    opt.focal_distance = extraFocal;
    if (extraDirect) opt.flags |= PT_FLAG_DIRECT_LIGHTING;
    // The reference's host calls pathtrace(pbo, frame, iter) once per iteration with iter = 1, 2, 3, ... (src/main.cpp:97-103):
    // the library traces such a sequence in wavefront batches AHEAD of the calls (PT_FLAG_TRACE_AHEAD; same image after every
    // call, bit for bit), so a call costs a commit instead of eight small dependent launches.  A camera move goes through
    // pathtraceFree / pathtraceInit (src/main.cpp:91-95), which drops whatever was traced ahead.
    // PT_AMD_TRACE_AHEAD=<iterations per batch> overrides (0 or 1: off, every call traces its own iteration).
    int ahead = 32;
    if (const char *e = getenv("PT_AMD_TRACE_AHEAD")) ahead = atoi(e);
    if (scene->state.iterations < (unsigned)(ahead > 0 ? ahead : 0)) ahead = (int)scene->state.iterations;
    if (ahead > PT_MAX_BATCH) ahead = PT_MAX_BATCH;
    // A host written against the reference's pathtraceInit knows nothing of batches, so tracing ahead must never be the
    // reason an Init fails: the batch is clamped to the library's limits for this frame (pixels x batch <= 2^29 paths,
    // rows x padded width x batch < 2^30 tile slots: an 8192 x 8192 frame still traces 8 iterations ahead), and should the
    // pools of the batch not fit the device's free memory -- pt_init then fails before it has allocated any -- it is halved
    // until they do; the last resort is the plain protocol, one iteration per call, which fits frames up to 2^30 pixels.
    {
        const long long W = scene->state.camera.resolution.x, H = scene->state.camera.resolution.y;
        const long long padded = (W + 255) / 256 * 256 * H;
        while (ahead > 1 && (W * H * ahead > (1ll << 29) || padded * ahead >= (1ll << 30))) ahead /= 2;
    }
    // `mesh` objects (README.md:236): their triangles go in before the geoms that refer to them.  (Built against the
    // reference's own scene.h, whose loader knows no meshes, the shim registers none.)
    std::vector<PtMesh> meshes;
#ifdef PT_SCENE_HAS_MESHES
    for (size_t i = 0; i < scene->meshes.size(); ++i) {
        PtMesh m;
        m.geom = scene->meshes[i].geom;
        m.ntris = (int)(scene->meshes[i].tris.size() / 9);
        m.tris = scene->meshes[i].tris.data();
        m.normals = scene->meshes[i].normals.empty() ? NULL : scene->meshes[i].normals.data();      // `vn`: smooth shading
        m.materials = scene->meshes[i].mats.empty() ? NULL : scene->meshes[i].mats.data();          // `usemtl <k>`: a material per face
        meshes.push_back(m);
    }
#endif
    int members = 1;
    if (const char *e = getenv("PT_AMD_DEVICES")) members = atoi(e);
    if (members >= 2) {
        if (group && pt_group_size(group) != members) {
            pt_group_destroy(group);
            group = NULL;
        }
        if (!group) checkPtError(pt_group_create(&group, members, NULL), "pathtraceInit (PT_AMD_DEVICES)");
        checkPtError(pt_group_set_meshes(group, meshes.empty() ? NULL : meshes.data(), (int)meshes.size()), "pathtraceInit");
    } else {
        checkPtError(pt_set_meshes_sized(meshes.empty() ? NULL : meshes.data(), (int)meshes.size(), sizeof(PtMesh)), "pathtraceInit");
    }
    int status;
    for (;;) {
        opt.flags &= ~PT_FLAG_TRACE_AHEAD;
        opt.max_batch = 0;
        opt.pipeline_depth = 0;
        if (ahead > 1) {
            opt.flags |= PT_FLAG_TRACE_AHEAD;
            opt.max_batch = ahead;
            opt.pipeline_depth = 2;     // one batch being consumed, one being traced (a third in flight only competes with the copy)
        }
        const PtCamera *cam = reinterpret_cast<const PtCamera *>(&scene->state.camera);
        const PtGeom *geoms = reinterpret_cast<const PtGeom *>(scene->geoms.data());
        const PtMaterial *mats = reinterpret_cast<const PtMaterial *>(scene->materials.data());
        status = group ? pt_group_init(group, cam, geoms, (int)scene->geoms.size(), mats, (int)scene->materials.size(), scene->state.traceDepth, &opt)
                       : pt_init(cam, geoms, (int)scene->geoms.size(), mats, (int)scene->materials.size(), scene->state.traceDepth, &opt);
        // (PT_ERR_INVALID / PT_ERR_HIP: a limit or an allocation that a smaller batch may satisfy; anything else is final)
        if (status == PT_OK || ahead <= 1 || (status != PT_ERR_INVALID && status != PT_ERR_HIP)) break;
        ahead /= 2;
    }
    checkPtError(status, "pathtraceInit");
    // state.image is owned by the Scene and lives from Init to Free: page-lock it for the per-iteration copy below
    // (an optimisation only; failure to register is not an error of the renderer)
    // (a group's read-back copies into the same buffer: the registration is process-wide, made through the default context)
    if (!scene->state.image.empty())
        (void)pt_pin_host(scene->state.image.data(), scene->state.image.size() * sizeof(scene->state.image[0]));
}

// Not a reference symbol: depth of field and direct lighting (README.md:100-101, :107-108) for the next pathtraceInit.
// A host that never calls it gets exactly the reference's renderer.
void pathtraceExtras(float lensRadius, float focalDistance, bool directLighting) {
    extraLens = lensRadius;
    extraFocal = focalDistance;
    extraDirect = directLighting;
}

void pathtraceFree() {
    if (group) {
        pt_group_destroy(group);
        group = NULL;
    }
    pt_free();  // no-op when nothing was initialised (src/main.cpp:91-94 calls Free before the first Init)
}

void pathtrace(uchar4 *pbo, int frame, int iter) {
    if (group) {
        if (pbo) {
            fprintf(stderr, "HIP error (pathtrace_shim.cpp): pathtrace: PT_AMD_DEVICES renders headless (pbo must be NULL)\n");
            exit(EXIT_FAILURE);
        }
        // one iteration on every member and the frame's reduce (asynchronous: the members go on tracing ahead), then the copy of the
        // reduced frame, which waits for the collective stream alone
        checkPtError(pt_group_iterate(group, frame, iter), "pathtrace");
        checkPtError(pt_group_readback(group, reinterpret_cast<float *>(hst_scene->state.image.data())), "pathtrace");
        return;
    }
    checkPtError(pt_iterate(frame, iter, pbo), "pathtrace");
    // Retrieve image from GPU: the un-normalised running sum (src/pathtrace.cu:170-171)
    checkPtError(pt_readback(reinterpret_cast<float *>(hst_scene->state.image.data())), "pathtrace");
}
