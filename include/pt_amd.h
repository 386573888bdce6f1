/*
 * pt_amd.h -- C ABI of the MI355X-native path-tracing hot path (libpt_amd.so).
 *
 * Drop-in boundary for CIS565-Fall-2015/Project3-CUDA-Path-Tracer's renderer API
 *     void pathtraceInit(Scene *scene);                       reference src/pathtrace.h:6,  src/pathtrace.cu:75-85
 *     void pathtraceFree();                                   reference src/pathtrace.h:7,  src/pathtrace.cu:87-92
 *     void pathtrace(uchar4 *pbo, int frame, int iteration);  reference src/pathtrace.h:8,  src/pathtrace.cu:123-174
 * Those three symbols are C++-mangled and take a libstdc++/glm-dependent `Scene*`, so the FFI-able
 * core below takes plain pointers and sizes; the ~40-line C++ shim that provides the three reference
 * symbols on top of it is project3-cuda-path-tracer_amd/host/pathtrace_shim.cpp.
 *
 * PtGeom / PtMaterial / PtCamera are byte-identical to Geom / Material / Camera of reference
 * src/sceneStructs.h:18-47, so `scene->geoms.data()`, `scene->materials.data()` and
 * `&scene->state.camera` pass straight through.
 *
 * All functions return PT_OK (0) or a negative PtStatus; pt_last_error() describes the calling thread's last failure.
 * A renderer instance is a CONTEXT.  The functions below act on the calling thread's CURRENT context: the process-wide default one --
 * one renderer per process, like the reference's file-static state (src/pathtrace.cu:70-71) -- unless pt_ctx_make_current named another
 * (section "contexts and device groups" at the end: several renderers in one process, one per device of a node).  One context is driven by
 * one host thread at a time (the reference is single-threaded); different threads may drive different contexts.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtVec3 { float x, y, z; } PtVec3;

/* reference src/sceneStructs.h:8-11; PT_MESH: the third object type of the scene format (README.md:236, "mesh"), which the
 * reference's enum and loader do not have -- see pt_set_meshes */
enum { PT_SPHERE = 0, PT_CUBE = 1, PT_MESH = 2 };

/* reference src/sceneStructs.h:18-27 -- 236 bytes, mat4 column-major (m[col*4+row]) */
typedef struct PtGeom {
    int32_t type;
    int32_t materialid;
    PtVec3  translation, rotation, scale;
    float   transform[16];
    float   inverseTransform[16];
    float   invTranspose[16];
} PtGeom;

/* reference src/sceneStructs.h:29-39 -- 44 bytes (flags are floats) */
typedef struct PtMaterial {
    PtVec3 color;
    float  specularExponent;
    PtVec3 specularColor;
    float  hasReflective, hasRefractive, indexOfRefraction, emittance;
} PtMaterial;

/* reference src/sceneStructs.h:41-47 -- 52 bytes; fov in degrees, fov.y = vertical HALF angle */
typedef struct PtCamera {
    int32_t resolution[2];
    PtVec3  position, view, up;
    float   fov[2];
} PtCamera;

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID = -1,      /* bad argument */
    PT_ERR_NOT_INIT = -2,     /* pt_iterate & co. before pt_init */
    PT_ERR_HIP = -3,          /* a HIP runtime call failed (replaces checkCUDAError, pathtrace.cu:21-39) */
    PT_ERR_DEVICE = -4,       /* a kernel reported an internal fault (path pool exhausted, chunk-list timeout): sticky,
                                 reported by pt_sync / pt_counters; results void */
    PT_ERR_NO_GPU = -5        /* no HIP device: there is NO CPU fallback */
} PtStatus;

enum {
    PT_FLAG_KERNEL_TIMING = 1,   /* bracket every bounce-kernel launch with HIP events (roofline measurement) */
    PT_FLAG_ACCUM_SHARD_ROWS = 2,/* the accumulator holds ONLY this shard's rows, packed (local row lr = global row
                                    lr * shard_count + shard_rank; nLocal * 3 floats): what a multi-GPU run gathers
                                    at rank 0 instead of reducing zero-padded full frames.  pt_readback still returns
                                    a full frame (other rows zero). */
    PT_FLAG_TRACE_AHEAD = 8,     /* for callers that keep the reference's protocol -- pathtrace(pbo, frame, iter) once per iteration
                                    with iter = 1, 2, 3, ... (src/main.cpp:97-103): pt_iterate(frame, iter) traces iterations
                                    iter .. iter + max_batch - 1 as ONE wavefront batch (iterations are independent: RNG keyed on
                                    iteration / pixel / depth), commits iteration `iter` alone and parks the radiance of the others;
                                    the calls for iter + 1, iter + 2, ... then only commit theirs, while the free slots of the
                                    pipeline trace the batches that follow.  The accumulator after every call is bit-identical to
                                    the one without the flag.  A call that does not continue the sequence (another iter, a
                                    pt_iterate_batch with count > 1) discards what is parked and starts over; pt_init / pt_free
                                    discard as well (the reference's camera-move restart, src/main.cpp:91-95).  Needs max_batch > 1
                                    to have any effect.  PtCounters: `iterations` counts committed iterations; live / light_hits /
                                    misses also cover the iterations traced ahead. */
    PT_FLAG_MIXTURE_WEIGHTED = 16,/* A material with REFL > 0 scatters 50 / 50 between the mirror and the diffuse lobe.  The reference's comment asks
                                    for "a 50/50 split ... divided by the probability" (src/interactions.h:54-58); the build's default
                                    takes SURVEY S6's energy-conserving reading WITHOUT the 1 / p weight, because that is what the
                                    staff render shows (the weighted form is 1.32 x on Cornell's back wall).  This flag renders the
                                    documented reading: the branch taken carries its weight 2 (the oracle's mirror mode 1). */
    PT_FLAG_DIRECT_LIGHTING = 4  /* README.md:107-108: "a final ray directly to a random point on an emissive object": at the
                                    last of the traceDepth bounces a diffuse scatter aims at a uniformly chosen point of a
                                    uniformly chosen emissive primitive (cosine-weighted), and ONE more bounce collects what
                                    that ray hits (traceDepth + 1 launches; traceDepth <= PT_MAX_DEPTH - 1) */
};

typedef struct PtOptions {
    int32_t shard_rank;       /* this process renders image rows y with y % shard_count == shard_rank */
    int32_t shard_count;      /* 1 = whole frame (reference behaviour) */
    int32_t device;           /* HIP device ordinal; -1 = current device */
    int32_t flags;            /* PT_FLAG_* */
    int32_t pipeline_depth;   /* iterations kept in flight on internal streams: 0 = default (3), 1 = none, max 4.
                                 Results do not depend on it: radiance is committed in iteration order. */
    int32_t max_batch;        /* largest `count` pt_iterate_batch will be given (sizes the path pools: two pools of
                                 44 B x pixels x max_batch per slot plus ~10 % of chunk slack -- 81 MB per iteration of a
                                 1280x720 frame and slot -- and pixels x max_batch must stay below 2^29);
                                 0 = 1, max PT_MAX_BATCH */
    void   *stream;           /* hipStream_t to enqueue on; NULL = the default stream */
    float  *accum_dev;        /* optional caller-owned device accumulator, W*H*3 floats (or the shard's rows only
                                 with PT_FLAG_ACCUM_SHARD_ROWS), zeroed by the caller (e.g. a torch tensor that
                                 RCCL exchanges); NULL = owned by the library like dev_image (pathtrace.cu:71,80-81) */
    /* README extras the reference names but does not implement; 0 = off = the reference's pinhole camera.
     * Depth of field by jittering rays within an aperture (README.md:100-101): camera rays start at a uniformly sampled
     * point of a lens disc of this radius around the eye and pass through the point their pinhole ray reaches at
     * focal_distance along the view axis.  (Imperfect specular, README.md:171-185, needs no option: a material with
     * REFL > 0 and SPECEX > 0 scatters into the Phong lobe of that exponent around the mirror direction.) */
    float   lens_radius;
    float   focal_distance;
} PtOptions;

#define PT_MAX_DEPTH 62
#define PT_MAX_BATCH 256
/* Bumped whenever a struct of this header changes size or meaning (6: round 6 -- pt_group_iterate / pt_group_reduce, the group's asynchronous
 * assembly; 5: contexts and groups; PtMesh has carried `normals` and `materials` since 4).  A host built against another header finds out with
 * pt_abi_version() != PT_AMD_ABI_VERSION before it passes structs. */
#define PT_AMD_ABI_VERSION 6
int pt_abi_version(void);

typedef struct PtCounters {
    int64_t live[PT_MAX_DEPTH + 2]; /* live[d] = paths entering bounce d (d = 1..depth), summed over   */
    int64_t light_hits;             /*           every iteration since pt_init / pt_counters_reset      */
    int64_t misses;                 /* traced paths that hit nothing (diagnostic: paths of the last bounce that certainly
                                       cannot reach an emitter are not traced and appear in neither tally)          */
    int64_t iterations;
    int64_t bounce_launches;        /* bounce-kernel launches covered by bounce_kernel_ms               */
    double  bounce_kernel_ms;       /* sum of HIP-event durations (PT_FLAG_KERNEL_TIMING only)          */
    double  raygen_kernel_ms;
    int64_t raygen_launches;
    int64_t ended_early[PT_MAX_DEPTH + 2]; /* of live[d]: paths whose scatter at bounce d - 1 certainly misses every primitive (scenes
                                              whose primitives are all walls or binned): tallied as entering bounce d and missing,
                                              which is what they do, but never written to or read from the path pools */
} PtCounters;

/* Triangle meshes (README.md:112-116 "arbitrary mesh loading and rendering", README.md:236 object type "mesh"; the reference
 * names them and `glm::intersectRayTriangle` and holds no mesh code, so the semantics are build-defined -- oracle/pt_oracle.cpp,
 * mesh_intersection_test):
 *   - a PtGeom of type PT_MESH carries its transform like any primitive; its triangles are in OBJECT space (the OBJ file's
 *     coordinates), 9 floats each (v0, v1, v2), counter-clockwise = front ("outside");
 *   - the ray is taken to object space like the sphere test does (src/intersections.h:104-110); the triangle test is
 *     glm::intersectRayTriangle (glm/gtx/intersect.inl:36-72) made two-sided; a triangle is tested when the ray passes the
 *     slab test of the triangle's bounding box (inflated by 1e-5 of the mesh's largest |coordinate|; plane parameters
 *     fma(plane, 1/d, -(o * 1/d)), compared with a relative slack of 1e-5) and its hit counts at or
 *     beyond that box's entry -- true of every geometric hit, and what lets a bounding-volume hierarchy return exactly the
 *     brute-force result; nearest = smallest object-space t, ties to the lower triangle index; flat shading, the normal
 *     negated on the back side; hit point and distance as the sphere test's (getPointOnRay, transform, world distance).
 * pt_set_meshes registers the triangle soups of the scene's mesh geoms for the NEXT pt_init (copied; kept across pt_free, so the
 * reference's Free -> Init restart re-initialises the same scene; pt_set_meshes(NULL, 0) clears them).  pt_init fails with
 * PT_ERR_INVALID when a PT_MESH geom has no triangles or triangles are registered for a geom that is not a mesh. */
typedef struct PtMesh {
    int32_t geom;           /* index into pt_init's geoms */
    int32_t ntris;
    const float *tris;      /* ntris x 9 floats, finite */
    /* README.md:112-116 leftovers (round 4), both optional:
     *   normals    ntris x 9 floats: the vertex normals n0, n1, n2 of every triangle, object space (an OBJ's `vn`).  The shading normal
     *              of a hit is then the barycentric blend n0 (1 - u - v) + n1 u + n2 v with the hit's own (u, v), turned to the side the
     *              counter-clockwise face normal points to and normalised (a blend of length zero keeps the face normal); everything
     *              else -- which side is "outside", the hit point, the invTranspose map to world space -- as for flat shading.
     *   materials  ntris scene materials, one per face (an OBJ's `usemtl <k>`); -1 = the object's own.  pt_init fails with
     *              PT_ERR_INVALID on an id >= nmats. */
    const float *normals;   /* or NULL: flat shading */
    const int32_t *materials;   /* or NULL: every face takes the object's material */
} PtMesh;
int pt_set_meshes(const PtMesh *meshes, int nmeshes);
/* ... the same with the caller's sizeof(PtMesh): PT_ERR_INVALID instead of strided garbage when host and library disagree about the struct */
int pt_set_meshes_sized(const PtMesh *meshes, int nmeshes, size_t mesh_struct_bytes);

/* pathtraceInit: upload scene, allocate the accumulator and the SoA path-state buffers.
 * Replaces reference src/pathtrace.cu:75-85.  Calling it twice without pt_free re-initialises. */
int pt_init(const PtCamera *cam, const PtGeom *geoms, int ngeoms, const PtMaterial *mats, int nmats,
            int traceDepth, const PtOptions *opts /* may be NULL */);

/* pathtrace(pbo, frame, iter): one iteration (1 spp), iter is 1-based like src/main.cpp:97-103.
 * Enqueues camera-ray generation, traceDepth fused intersect+shade+compact launches and, when
 * rgba8_dev != NULL, the sendImageToPBO conversion (src/pathtrace.cu:48-68) into that DEVICE
 * buffer of W*H uchar4.  Asynchronous on the configured stream.  Replaces src/pathtrace.cu:123-167. */
int pt_iterate(int frame, int iter, void *rgba8_dev /* may be NULL (headless) */);

/* `count` consecutive iterations first_iter .. first_iter+count-1 traced as ONE wavefront (their paths share the
 * bounce launches, which makes each launch `count` times larger: fewer, fatter launches for small frames or
 * small multi-GPU shards).  The result is identical to `count` pt_iterate calls: every pixel still receives its
 * samples in iteration order.  rgba8_dev, if given, is converted with iter = first_iter+count-1.
 * pt_iterate(frame, iter, pbo) == pt_iterate_batch(frame, iter, 1, pbo). */
int pt_iterate_batch(int frame, int first_iter, int count, void *rgba8_dev);

/* Wait for every stream the renderer uses; reports device-side faults (PT_ERR_DEVICE).  (checkCUDAError's sync,
 * pathtrace.cu:23.)  Without a renderer it waits for the scan library's work on every stream. */
int pt_sync(void);

/* Copy the un-normalised running sum (W*H*3 floats, index = x + y*W) to host: the D2H copy of
 * src/pathtrace.cu:170-171 into scene->state.image.  Synchronises. */
int pt_readback(float *rgb_sum_host);

/* Page-lock a caller-owned host buffer (e.g. scene->state.image) so that pt_readback into it runs at PCIe rate instead of
 * through the runtime's pageable staging copy: the per-iteration D2H copy of the reference protocol (pathtrace.cu:170-171).
 * The caller keeps the memory alive until pt_unpin_host / pt_free, which release the registration; one buffer at a time.
 * Purely an optimisation: pt_readback works with any host pointer. */
int pt_pin_host(void *host, size_t bytes);
int pt_unpin_host(void);

/* sendImageToPBO into a HOST buffer (W*H*4 bytes). Synchronises. */
int pt_readback_rgba8(int iter, uint8_t *rgba_host);

int pt_counters(PtCounters *out);     /* synchronises */
int pt_counters_reset(void);

/* pathtraceFree: tolerant of the never-initialised state (src/main.cpp:91-94 calls it first). */
void pt_free(void);

const char *pt_last_error(void);

/* Number of HIP devices visible (0 = none; the library then refuses to run). */
int pt_device_count(void);

/* ---- stream compaction library (the reference's empty stream_compaction/ stub, README.md:83-86):
 * work-efficient exclusive scan / compaction over DEVICE buffers, multi-block, any n >= 0 (compaction: n < 2^32).
 * Reduce-then-scan over at most 2048 chunks: two launches on `stream`, asynchronous, no workgroup waits for another;
 * calls on different streams use separate workspaces and may overlap.  Sums wrap modulo 2^32 like int32 arithmetic. ---------- */
int pt_scan_exclusive_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, void *stream);
/* keeps the non-zero elements in order; *count_dev (device) receives how many were kept */
int pt_compact_nonzero_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, int64_t *count_dev,
                           void *stream);

/* ---- contexts and device groups (round 5) -----------------------------------------------------------------------------------------
 * The reference's renderer is one set of file-static globals bound to device 0 (src/pathtrace.cu:70-71, src/preview.cpp:107).  Here a
 * renderer is a context: pt_ctx_create makes one, pt_ctx_make_current(ctx) makes every function above act on it for the calling thread
 * (NULL: back to the default context), pt_ctx_destroy frees its renderer and the context.  A context initialised on a device makes that
 * device current when it is made current.  pt_set_meshes registers meshes per context. */
typedef struct PtContext PtContext;
PtContext *pt_ctx_create(void);               /* NULL: out of memory */
int        pt_ctx_make_current(PtContext *ctx /* NULL = the default context */);
PtContext *pt_ctx_current(void);              /* NULL = the default context */
int        pt_ctx_destroy(PtContext *ctx);    /* PT_ERR_INVALID while the context is current on ANOTHER thread (that thread's next call
                                                 would act on freed memory); the caller's own current context may be destroyed: the
                                                 thread falls back to the default one.  The current HIP device is left as it was. */

/* A GROUP: n contexts that render the row shards y % n of ONE frame, member i on devices[i] (NULL: device i % pt_device_count()) -- the
 * node's GPUs side by side in one host process ("single process, ncclCommInitAll, one stream per device"; the reference is
 * hard-wired to device 0, src/preview.cpp:107), or several renderers on one device.  Pixels are seeded by their global index, so the shards
 * together are the one-device frame bit for bit.  Every member's launches are enqueued by a host thread of its own.
 *   pt_group_init            = pt_init on every member (opts' shard_rank / shard_count / device / stream / accum_dev are the group's to set;
 *                              max_batch, pipeline_depth, flags and the lens apply to each member).  The members on one device commit their rows
 *                              into ONE zero-padded full-frame accumulator of that device.
 *   pt_group_iterate_batch   enqueues every member's wavefront batch (asynchronous: the devices run concurrently).
 *   pt_group_reduce          assembles the frame from what has been committed so far, asynchronously: members on DISTINCT devices (and
 *                              librccl.so loadable: dlopen, never linked) by ONE ncclReduce(sum, float32, 3 W H, root = member 0's device) over
 *                              xGMI of double-buffered snapshots of the devices' accumulators on a collective stream per device -- the next
 *                              iteration's commits wait on the device for the snapshot only, never for the reduce, and nothing waits on the host;
 *                              members that all share ONE device need no collective (their accumulator is the frame).
 *   pt_group_iterate         = pt_group_iterate_batch(frame, iter, 1) + pt_group_reduce: BASELINE config C3 as written -- the reference's
 *                              per-iteration full-frame transfer (src/pathtrace.cu:170-171, src/main.cpp:97-106) with the reduce in its place.
 *                              With PT_FLAG_TRACE_AHEAD in opts->flags the iteration comes out of batches traced ahead (one small commit per call).
 *   pt_group_readback        the latest assembled frame's running sum on the host (assembling first if commits were enqueued since): one D2H copy
 *                              on the root's collective stream and a wait for THAT stream -- not for the batches traced ahead.
 *   pt_group_sync            waits for every member's streams and the collective streams; reports a member's device fault.
 * pt_group_collective() says how the frame is assembled: "rccl reduce ..." (with "unmeasured across devices": NO multi-GPU node was
 * available to any build round -- the leg has run in a one-rank communicator only), "shared accumulator" (one device) or "host gather"
 * (several devices without RCCL: every device's frame copied to the host, rows taken from their owners; synchronous).
 * PT_AMD_COLLECTIVE=host|rccl overrides (rccl: a one-rank communicator even when every member shares one device);
 * PT_AMD_GROUP_THREADS=0 issues every member's work from the calling thread (experiments).
 * A failed ncclReduce leaves RCCL's call group closed (ncclGroupEnd runs on every path) and the group usable: a later call tries again.
 * A group the host forgets to destroy is destroyed by the library's exit handler. */
typedef struct PtGroup PtGroup;
int  pt_group_create(PtGroup **out, int n, const int32_t *devices /* may be NULL */);
void pt_group_destroy(PtGroup *g);
int  pt_group_size(const PtGroup *g);
const char *pt_group_collective(const PtGroup *g);
int  pt_group_set_meshes(PtGroup *g, const PtMesh *meshes, int nmeshes);
int  pt_group_init(PtGroup *g, const PtCamera *cam, const PtGeom *geoms, int ngeoms, const PtMaterial *mats, int nmats, int traceDepth,
                   const PtOptions *opts /* may be NULL */);
int  pt_group_iterate_batch(PtGroup *g, int frame, int first_iter, int count);
int  pt_group_iterate(PtGroup *g, int frame, int iter);
int  pt_group_reduce(PtGroup *g);
int  pt_group_sync(PtGroup *g);
int  pt_group_readback(PtGroup *g, float *rgb_sum_host);
int  pt_group_counters(PtGroup *g, PtCounters *out);      /* summed over the members; iterations: the least any member has committed */

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
