// pt_api.hip -- MI355X (gfx950 / CDNA4) path-tracing hot path + its C ABI (include/pt_amd.h).
//
// One iteration = `traceDepth` launches of ONE fused persistent kernel per bounce (the first builds its camera rays in
// registers): nearest-hit over the scene (geometry through the scalar path, materials in LDS) -> shade/scatter -> park
// emitter radiance -> stream compaction of the survivors straight into the next bounce's SoA path pool (wave64
// ballot/mbcnt ranks, LDS wave totals = workgroup-level exclusive scan per class; the tile's output runs are reserved
// with ONE atomic instruction on sharded position counters, chunks of the pool are handed out on demand).  No host
// round trip inside an iteration: live counts stay on the device.  The multi-workgroup ORDERED scan (two-level decoupled
// look-back) is the stream-compaction library in pt_compaction.h (pt_scan_exclusive_i32 / pt_compact_nonzero_i32).
// Files: pt_device.h (math), pt_trace.h (render kernels), pt_compaction.h, pt_test_kernels.h (primitives for the parity
// tests), this file (host side + C ABI).
//
// Replaces the unsolved pipeline of reference src/pathtrace.cu:133-167 (spec: SURVEY.md 3.4 S0-S9).
// HBM layout, kernels, rooflines: DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include <mutex>

#include "../../include/pt_amd.h"
#include "pt_compaction.h"
#include "pt_trace.h"
#include "pt_mesh.h"
// The entry points of include/pt_amd_test.h (device primitives one by one, soundness sweeps, the fault-word hook) exist only in
// the second link target of this source, libpt_amd_test.so (-DPT_TEST_API): the product library exports none of them.
#ifdef PT_TEST_API
#include "../../include/pt_amd_test.h"
#include "pt_test_kernels.h"
#endif

using namespace ptd;
using namespace ptk;

static_assert(sizeof(PtGeom) == 236 && sizeof(PtMaterial) == 44 && sizeof(PtCamera) == 52,
              "layout must equal reference src/sceneStructs.h:18-47");
namespace {

// =====================================================================================================
// host side
// =====================================================================================================
std::string g_err = "";

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(PT_ERR_HIP, "HIP error (%s:%d): %s: %s", "pt_api.hip", __LINE__, #expr, \
                        hipGetErrorString(e_));                                                     \
    } while (0)

constexpr int kMaxSlots = 4;

// One in-flight iteration: its own stream, path buffers, counters and deferred-radiance buffer.
struct Slot {
    hipStream_t stream = nullptr;
    float *pathbuf[2] = {nullptr, nullptr};      // two path pools (ping-pong by bounce), poolChunks * kChunk paths per array
    unsigned long long *chunkList[2] = {nullptr, nullptr}; // per pool: [kSeg][poolChunks] chunk lists of the queue it holds
    uint32_t gen[2] = {0, 0};      // per pool: serial number of the launch that filled it (tags the chunk-list entries)
    Ctrl *ctrl = nullptr;
    float *contrib = nullptr;      // maxBatch x W*H*3, zero between batches
    uint32_t *hitMask = nullptr;   // ceil(maxBatch / 32) x W*H: which iterations of the batch wrote a pixel's `contrib`, zero between batches
    hipEvent_t evDone = nullptr;       // all bounce launches of the slot's current iteration finished
    hipEvent_t evCommitted = nullptr;  // k_commit consumed (and re-zeroed) `contrib`
    int parity = 0;                // which half of Ctrl::cursor the slot's next batch uses
};

struct State {
    bool init = false;
    int device = 0;
    hipStream_t stream = nullptr;   // the caller's stream: commits, PBO conversion, readback
    PtCamera cam;
    KParams prm;
    int P = 0;              // W*H
    int nLocal = 0;
    int flags = 0;
    float *image = nullptr;
    bool ownImage = false;
    int nslots = 0;
    int maxBatch = 1;       // iterations that may share one wavefront (pt_iterate_batch)
    Slot slot[kMaxSlots];
    GeomDev *dgeoms = nullptr;
    float4 *dGeomHit = nullptr;     // GeomHitDev[ngeoms], as the kernels stage it in LDS (scenes that are not sphere-heavy)
    MaterialDev *dmats = nullptr;
    WallBox *dwalls = nullptr;
    SphereCull *dSphCull = nullptr; // sphere-heavy scenes: packed culling data of the spheres, and ...
    int *dRowOff = nullptr, *dRowIdx = nullptr;   // camera-ray bounce: per image row, the primitives whose pixel rectangle covers it
    int *dClassIdx = nullptr;       // later bounces: per queue class, the primitives to look at (KParams::classOff)
    float4 *dMeshRecs = nullptr;    // ptd::MeshUnit[]: triangles and inner nodes of every mesh of the scene (k_bounce<., ., ., true>)
    bool mesh = false;      // the scene holds triangle meshes: the k_bounce<., false, ., true> variants
    int numTilesMax = 0;    // upper bound of tiles in one bounce queue (incl. one partial tile per segment)
    int poolChunks = 0;     // chunks per path pool (incl. the trash chunk 0); a pool holds poolChunks * kChunk paths per array
    int grid = 0;           // persistent grid of k_bounce<false>
    int gridFirst = 0;      // ... and of k_bounce<true> (its own register budget, hence its own residency)
    bool many = false;      // more than kBinMax spheres: the k_bounce<., true> variants (per-lane sphere lists)
    bool dof = false;       // thin-lens camera: the k_bounce<true, ., true> variants for the camera-ray bounce
    bool plain = false;     // no refractive material, no specular exponent on a reflective one, no direct lighting: k_bounce<..., PLAIN>
    size_t ldsBytes = 0, ldsBytesNext = 0;   // dynamic LDS of the camera-ray launch / of the later ones
    long long iterations = 0;
    long long seq = 0;      // batches enqueued since pt_init: slot = seq % nslots
    // PT_FLAG_TRACE_AHEAD: batches traced ahead of the pt_iterate calls that will ask for their iterations, oldest first.
    // A parked batch occupies its slot (radiance buffers, iteration masks) until its last iteration is committed or it
    // is discarded; the parked slots are the `ahead.size()` slots before seq % nslots in the rotation.
    struct Parked { int slot, first, count, next; bool waited; };   // iterations first .. first + count - 1, the next one to commit = first + next;
                                                                    // waited: the caller's stream already waits for the batch's launches
    std::deque<Parked> ahead;
    uint32_t launchSerial = 0;   // bounce launches since pt_init
    // host buffer of pt_readback, page-locked on first use so the per-iteration D2H copy of the reference protocol
    // (src/pathtrace.cu:170-171) runs at PCIe rate instead of through a pageable staging copy
    void *pinnedHost = nullptr;
    size_t pinnedBytes = 0;
    // kernel timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evBounce;
    std::vector<hipEvent_t> evFree;   // resolved timing events, reused (creating two per launch cost 1 % of a timed step)
    double msBounce = 0;
    long long nBounce = 0;
    // The sticky fault word once more, in page-locked HOST memory the kernels can write (a fault path stores it there too): pt_readback
    // and pt_readback_rgba8, which synchronise anyway, then report a faulted render without a device-to-host copy of their own.
    uint32_t *hostFault = nullptr;       // host address
    uint32_t *hostFaultDev = nullptr;    // the same word as the kernels address it
} S;

// triangle soups registered by pt_set_meshes, consumed by the next pt_init (kept across pt_free: the reference's
// Free -> Init restart protocol re-initialises the same scene)
std::vector<ptm::HostMesh> g_meshes;
const ptm::HostMesh *mesh_of(int geom) {
    for (const ptm::HostMesh &m : g_meshes)
        if (m.geom == geom) return &m;
    return nullptr;
}

int count_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int reset_ctrl(Ctrl *dev, hipStream_t st) {
    HIPCHECK(hipMemsetAsync(dev, 0, sizeof(Ctrl), st));
    HIPCHECK(hipStreamSynchronize(st));
    return PT_OK;
}

PathPool pool(const Slot &sl, int which) {
    PathPool p;
    p.base = sl.pathbuf[which];
    p.list = sl.chunkList[which];
    p.cap = (uint32_t)S.poolChunks << S.prm.chunkShift;
    return p;
}

// `box`: the object-space bounds lo[3], hi[3] of the primitive -- nullptr = the unit cube [-0.5, 0.5]^3 of a sphere or cube,
// a mesh passes the union of its (inflated) triangle boxes.
void pack_geom(const PtGeom &g, GeomDev &d, const float *eye = nullptr, const float *box = nullptr) {
    memset(&d, 0, sizeof d);
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) {
            d.inv[c * 3 + r] = g.inverseTransform[c * 4 + r];
            d.xf[c * 3 + r] = g.transform[c * 4 + r];
            d.invT[c * 3 + r] = g.invTranspose[c * 4 + r];
        }
    for (int r = 0; r < 3; ++r) d.invZ[r] = d.inv[9 + r] * 0.0f;
    d.type = g.type == PT_CUBE ? 1 : 0;                   // (a mesh's normal is made like a sphere's: type 0, flags bit 5)
    d.flags = d.cullFlags = g.type == PT_CUBE ? 1 : (g.type == PT_MESH ? 32 : 0);    // bit 1 (binned) is set by pt_init
    d.meshRoot = ptd::kMeshEnd;                           // set by pt_init / the mesh tests
    d.material = g.materialid;
    // bounding-ball culling data (ptd::certainMiss): bounds smax >= sigma_max, smin <= sigma_min of the 3x3 part
    double A[3][3], Ai[3][3];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) { A[c][r] = g.transform[c * 4 + r]; Ai[c][r] = g.inverseTransform[c * 4 + r]; }
    double len[3], fro = 0, froi = 0;
    bool orth = true;
    for (int c = 0; c < 3; ++c) {
        len[c] = std::sqrt(A[c][0] * A[c][0] + A[c][1] * A[c][1] + A[c][2] * A[c][2]);
        for (int r = 0; r < 3; ++r) { fro += A[c][r] * A[c][r]; froi += Ai[c][r] * Ai[c][r]; }
    }
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b) {
            const double dp = A[a][0] * A[b][0] + A[a][1] * A[b][1] + A[a][2] * A[b][2];
            if (!(std::fabs(dp) <= 1e-5 * len[a] * len[b])) orth = false;
        }
    double smax, smin;
    if (orth) {            // rotation x scale: the singular values are the column lengths
        smax = std::max(len[0], std::max(len[1], len[2])) * (1 + 1e-5);
        smin = std::min(len[0], std::min(len[1], len[2])) * (1 - 1e-5);
    } else {               // any matrix: Frobenius bounds
        smax = std::sqrt(fro);
        smin = froi > 0 ? 1.0 / std::sqrt(froi) : 0.0;
    }
    d.centre[0] = g.transform[12]; d.centre[1] = g.transform[13]; d.centre[2] = g.transform[14];
    double rho2 = g.type == PT_SPHERE ? 0.25 : 0.75;       // object-space bounding ball: the sphere / the cube's corners
    bool boxOk = true;
    if (box) {
        // a mesh: the ball around the centre of its box.  The margins of certainMiss are relative to the ball; they cover the
        // rounding of the object-space evaluation (relative to the distance from the object-space ORIGIN) only while the
        // mesh is not far off its own origin: otherwise it is never culled.
        double c[3], far = 0;
        rho2 = 0;
        for (int a = 0; a < 3; ++a) {
            c[a] = 0.5 * ((double)box[a] + box[3 + a]);
            const double h = 0.5 * ((double)box[3 + a] - box[a]);
            rho2 += h * h;
            far = std::max(far, std::max(std::fabs((double)box[a]), std::fabs((double)box[3 + a])));
        }
        rho2 *= 1 + 1e-6;                                  // (the centre is rounded to float below)
        for (int r = 0; r < 3; ++r)
            d.centre[r] = (float)((double)g.transform[0 + r] * c[0] + (double)g.transform[4 + r] * c[1] + (double)g.transform[8 + r] * c[2] +
                                  (double)g.transform[12 + r]);
        boxOk = std::isfinite(rho2) && rho2 > 0 && far <= 100.0 * std::sqrt(rho2);
    }
    const double r2 = rho2 * smax * smax * (1 + 1e-3), kk = smin > 0 ? 1e-4 * (smax / smin) * (smax / smin) : INFINITY;
    const bool ok = boxOk && std::isfinite(r2) && std::isfinite(kk) && kk < 0.5 && smin > 0;
    d.boundR = (float)(std::sqrt(rho2) * smax);
    d.cullR2 = ok ? (float)r2 : INFINITY;      // infinite radius: never culled
    d.cullK = ok ? (float)kk : 0.0f;
    d.rect[0] = d.rect[1] = 0;                 // whole frame until pt_init projects the primitive (project_geom)
    d.rect[2] = d.rect[3] = 0x7fffffff;
    if (g.type == PT_CUBE) {
        // Per face: ptd::normalize(ptd::mulMV(xf, +-e_axis, 0)) and ptd::hemisphereFrame of that normal, operation by
        // operation (this file is built with -ffp-contract=off; host sqrt and division are correctly rounded like the
        // device's): what cubeFrameVector() looks up.
        struct V { float x, y, z; };
        auto normalize = [](V a) {
            const float xx = a.x * a.x, yy = a.y * a.y, zz = a.z * a.z;
            const float xy = xx + yy;
            const float dt = xy + zz;                                 // glm dot: (x*x + y*y) + z*z
            const float inv = 1.0f / std::sqrt(dt);                   // glm::inversesqrt
            return V{a.x * inv, a.y * inv, a.z * inv};
        };
        auto cross = [](V x, V y) {                                   // glm/detail/func_geometric.inl:134-143
            const float a0 = x.y * y.z, a1 = y.y * x.z, b0 = x.z * y.x, b1 = y.z * x.x, c0 = x.x * y.y, c1 = y.x * x.y;
            return V{a0 - a1, b0 - b1, c0 - c1};
        };
        const float kSqrtOneThird = 0.5773502691896257645091487805019574556476f;   // src/utilities.h:15
        const float *m = d.xf;
        for (int axis = 0; axis < 3; ++axis)
            for (int pos = 0; pos < 2; ++pos) {
                float v[3] = {0.0f, 0.0f, 0.0f};
                v[axis] = pos ? 1.0f : -1.0f;
                float r[3];
                for (int c = 0; c < 3; ++c) {
                    const float a0 = m[0 + c] * v[0], a1 = m[3 + c] * v[1], a2 = m[6 + c] * v[2], a3 = m[9 + c] * 0.0f;
                    const float s01 = a0 + a1, s23 = a2 + a3;
                    r[c] = s01 + s23;
                }
                const V n = normalize(V{r[0], r[1], r[2]});
                V notNormal;
                if (std::fabs(n.x) < kSqrtOneThird) notNormal = V{1, 0, 0};
                else if (std::fabs(n.y) < kSqrtOneThird) notNormal = V{0, 1, 0};
                else notNormal = V{0, 0, 1};
                const V p1 = normalize(cross(n, notNormal));
                const V p2 = normalize(cross(n, p1));
                float *out = d.cubeFrame + 9 * (2 * axis + pos);
                out[0] = n.x; out[1] = n.y; out[2] = n.z;
                out[3] = p1.x; out[4] = p1.y; out[5] = p1.z;
                out[6] = p2.x; out[7] = p2.y; out[8] = p2.z;
            }
    }
    if (eye) {   // ptd::mulMV(inv, eye, 1) in the same operation order (this file is built with -ffp-contract=off)
        const float *m = d.inv;
        for (int r = 0; r < 3; ++r) {
            const float a0 = m[0 + r] * eye[0], a1 = m[3 + r] * eye[1], a2 = m[6 + r] * eye[2], a3 = m[9 + r] * 1.0f;
            const float s01 = a0 + a1, s23 = a2 + a3;
            d.camObj[r] = s01 + s23;
        }
    }
}
// World-space box of a cube, INFLATED for ptd::wallCertainMiss: the 8 corners of the unit cube through `transform` in double
// precision, widened by delta = 4e-5 S, S = max(diagonal of the box, largest |coordinate|), and rounded outwards to float.
// Why 4e-5: a scattered ray starts 1e-3 off the surface it leaves (spec S6), and the certificate has to be able to tell
// that it leaves -- delta must stay below that offset for a scene of Cornell's size (S = 14: delta = 5.7e-4) -- while the
// reference's own evaluation moves the boundary by ~2e-7 (|o| + S) (object-space transform, thin axis: products of
// magnitude 100 |o| rounded to 2^-24, scaled back by 1/100), i.e. 1.2e-5 for |o| + S <= 60 = *omax: a margin of 47x.
// Returns S (a negative value when the cube is not finite); *omax receives the largest |x| + |y| + |z| of a ray origin
// for which that margin holds, 5 S - (largest |coordinate|).
double wall_box(const PtGeom &g, WallBox &w, double *omax = nullptr) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int corner = 0; corner < 8; ++corner) {
        const double o[3] = {(corner & 1) ? 0.5 : -0.5, (corner & 2) ? 0.5 : -0.5, (corner & 4) ? 0.5 : -0.5};
        for (int r = 0; r < 3; ++r) {
            const double q = (double)g.transform[0 + r] * o[0] + (double)g.transform[4 + r] * o[1] + (double)g.transform[8 + r] * o[2] +
                             (double)g.transform[12 + r];
            if (!std::isfinite(q)) return -1.0;
            lo[r] = std::min(lo[r], q);
            hi[r] = std::max(hi[r], q);
        }
    }
    double diag = 0, big = 0;
    for (int r = 0; r < 3; ++r) {
        diag += (hi[r] - lo[r]) * (hi[r] - lo[r]);
        big = std::max(big, std::max(std::fabs(lo[r]), std::fabs(hi[r])));
    }
    const double S_ = std::max(std::sqrt(diag), big);
    const double delta = 4e-5 * S_;
    if (omax) *omax = 5.0 * S_ - big;
    memset(&w, 0, sizeof w);
    for (int r = 0; r < 3; ++r) {
        w.lo[r] = std::nextafter((float)(lo[r] - delta), -INFINITY);
        w.hi[r] = std::nextafter((float)(hi[r] + delta), INFINITY);
        if (!std::isfinite(w.lo[r]) || !std::isfinite(w.hi[r])) return -1.0;
    }
    return S_;
}

// Sphere-heavy scenes: the spheres in TWO SPATIAL CLUSTERS.  `sc` (every sphere's packed culling data, thresholds already scaled) is split at
// the median centre along one axis and reordered, cluster 0 first (n0 entries, even: padded with a copy of its last one).  A survivor's class bits
// 3 / 4 say which clusters its ray can hit (k_bounce: a slab certificate, ptd::wallCertainMiss, against each cluster's box), and a tile of the
// next bounce sweeps only those.  The binned primitives (group 0) share bit 3: axis and order of the halves are the ones with the smallest sum of
// (surface area of what a bit stands for) x (spheres behind it).  A choice that only steers which tiles skip which tests; results never depend on it.
// What the certificate rests on: box g holds, for every sphere of cluster g, the ball of radius sqrt(cullR2 + K ocMax^2) (1 + 1e-6) around its
// centre -- the sphere's own half-line certificate (ptd::sphereHalfLineExcess: distance^2 of the centre from the half-line > cullR2 + K |oc|^2, with
// |oc| <= ocMax = omax + |centre| for every origin a certificate is issued for, |x| + |y| + |z| <= omax) holds for every half-line that misses
// that ball -- and is inflated like a wall's box (wall_box: delta = 4e-5 S against ~2e-7 (|o| + S) of rounding in the slab test, |o| <= 5 S).
// Returns false (and leaves everything as it was) when no clusters can be built: fewer than two spheres, or one that is never culled.
// (tests/test_gpu_parity.py::test_sphere_cluster_boxes_never_reject_a_hit: pt_test_sphere_cluster_sweep, 2^28 rays, 0 violations.)
bool build_sphere_clusters(const PtGeom *geoms, int ngeoms, const std::vector<GeomDev> &hg, const std::vector<int> &binned, std::vector<SphereCull> &sc,
                           int &n0, float &omaxOut, float box[2][8]) {
    // a sphere's half-line ball for the origins certificates are issued for: radius^2 = cullR2 + K ocMax^2 (scaled thresholds: the larger)
    auto build = [&](const std::vector<SphereCull> &v, double omax, double lo[3], double hi[3]) {
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (const SphereCull &e : v) {
            const double cn = std::sqrt((double)e.centre[0] * e.centre[0] + (double)e.centre[1] * e.centre[1] + (double)e.centre[2] * e.centre[2]);
            const double ocMax = omax + cn;
            const double r = std::sqrt((double)e.cullR2 + (double)e.cullK * ocMax * ocMax) * (1.0 + 1e-6);
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)e.centre[a] - r); hi[a] = std::max(hi[a], (double)e.centre[a] + r); }
        }
    };
    auto area = [](const double lo[3], const double hi[3]) {
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    };
    bool finite = true;
    for (const SphereCull &e : sc) finite = finite && std::isfinite(e.cullR2) && std::isfinite(e.cullK);
    if (!finite || sc.size() < 2) return false;
    double bestCost = INFINITY;
    std::vector<SphereCull> best0, best1;
    for (int axis = 0; axis < 3; ++axis)
        for (int swap = 0; swap < 2; ++swap) {
            std::vector<SphereCull> v = sc;
            std::stable_sort(v.begin(), v.end(), [&](const SphereCull &a, const SphereCull &b) { return a.centre[axis] < b.centre[axis]; });
            const size_t h = v.size() / 2;
            std::vector<SphereCull> c0(v.begin(), v.begin() + h), c1(v.begin() + h, v.end());
            if (swap) std::swap(c0, c1);
            double lo0[3], hi0[3], lo1[3], hi1[3];
            build(c0, 0.0, lo0, hi0);
            build(c1, 0.0, lo1, hi1);
            for (int i : binned) {                                // bit 3 also stands for the binned primitives
                const GeomDev &G = hg[i];
                for (int a = 0; a < 3; ++a) { lo0[a] = std::min(lo0[a], (double)G.centre[a] - G.boundR); hi0[a] = std::max(hi0[a], (double)G.centre[a] + G.boundR); }
            }
            const double cost = area(lo0, hi0) * (double)c0.size() + area(lo1, hi1) * (double)c1.size();
            if (cost < bestCost) { bestCost = cost; best0 = c0; best1 = c1; }
        }
    // (a scattered ray starts on a primitive: the scene's own extent, |x| + |y| + |z| over its bounding box, with a quarter to spare,
    // bounds the origins worth a certificate -- and K |oc|^2 grows with the bound)
    double omax = 0.0, S_[2], big_[2];
    {
        double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < ngeoms; ++i) {
            WallBox wb;
            if (geoms[i].type == PT_CUBE && wall_box(geoms[i], wb) >= 0)
                for (int a = 0; a < 3; ++a) { slo[a] = std::min(slo[a], (double)wb.lo[a]); shi[a] = std::max(shi[a], (double)wb.hi[a]); }
            else if (std::isfinite(hg[i].boundR))
                for (int a = 0; a < 3; ++a) { slo[a] = std::min(slo[a], (double)hg[i].centre[a] - hg[i].boundR); shi[a] = std::max(shi[a], (double)hg[i].centre[a] + hg[i].boundR); }
        }
        for (int a = 0; a < 3; ++a) omax += std::max(std::fabs(slo[a]), std::fabs(shi[a]));
        omax *= 1.25;
        if (!std::isfinite(omax)) omax = 0.0;
    }
    float bx[2][8];
    for (int g = 0; g < 2; ++g) for (int q = 0; q < 8; ++q) bx[g][q] = 0.0f;
    for (int pass = 0; pass < 2; ++pass)          // pass 0: the boxes' sizes with |oc| = |c|, for the bound; pass 1: the boxes for the bound that gave
        for (int g = 0; g < 2; ++g) {
            double lo[3], hi[3];
            build(g ? best1 : best0, pass ? omax : 0.0, lo, hi);
            double diag = 0, big = 0;
            for (int a = 0; a < 3; ++a) {
                diag += (hi[a] - lo[a]) * (hi[a] - lo[a]);
                big = std::max(big, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
            }
            S_[g] = std::max(std::sqrt(diag), big); big_[g] = big;
            if (!pass) omax = std::min(omax, 5.0 * S_[g] - big);
            else {
                const double delta = 4e-5 * S_[g];
                for (int a = 0; a < 3; ++a) {
                    bx[g][a] = std::nextafter((float)(lo[a] - delta), -INFINITY);
                    bx[g][3 + a] = std::nextafter((float)(hi[a] + delta), INFINITY);
                    finite = finite && std::isfinite(bx[g][a]) && std::isfinite(bx[g][3 + a]);
                }
                // (the boxes only grew since pass 0 -- and a box built for a larger bound than the final one is the conservative side)
                omax = std::min(omax, 5.0 * S_[g] - big_[g]);
            }
        }
    if (!finite || !(omax > 0.0) || !std::isfinite(omax)) return false;
    if (best0.size() % 2) best0.push_back(best0.back());      // (two per scalar load; testing a sphere twice changes nothing)
    sc = best0;
    sc.insert(sc.end(), best1.begin(), best1.end());
    n0 = (int)best0.size();
    omaxOut = std::nextafter((float)omax, 0.0f);
    memcpy(box, bx, sizeof bx);
    return true;
}

// The walls of a scene -- its large cubes: not binned, finite, at most kWallMax of them, the largest first -- with what the survivors'
// certificates need (k_bounce: which wall can a scattered ray still hit?): the inflated world boxes (wall_box), the bound on the ray
// origins the margins hold for, and for the walls that have one the PLANE of their box that faces the scene's interior
// (ptd::wallPlanesPossible): with C the centre of the box `outer` around all the walls' boxes, a face of a wall's box whose whole box
// lies beyond C on that axis.  Six slots (axis x side) hold one wall each: a wall takes the free slot in which it lies farthest out
// (in units of the scene's extent), the largest walls choose first; walls without a slot (a box across the middle of the scene, a
// second wall on the same side) are numbered behind the others and keep the slab test.
// Thresholds: the plane moved towards the interior by slack = 2e-6 (wallOMax + |diagonal of outer|), five times the rounding of the
// ray's exit point.  A choice that only steers which tiles skip which tests; results never depend on it.
// wallGeom[w] = the primitive that is wall w.
void choose_walls(const PtGeom *geoms, int ngeoms, const std::vector<GeomDev> &hg, KParams &k, std::vector<WallBox> &hw, std::vector<int> &wallGeom) {
    std::vector<std::pair<double, int>> cand;
    for (int i = 0; i < ngeoms; ++i)
        if (geoms[i].type == PT_CUBE && !hg[i].binned) cand.emplace_back(-(double)hg[i].boundR, i);
    std::sort(cand.begin(), cand.end());
    std::vector<WallBox> boxes;
    std::vector<int> which;
    double omaxAll = INFINITY;
    for (size_t c = 0; c < cand.size() && (int)boxes.size() < kWallMax; ++c) {
        WallBox wb;
        double om = 0;
        if (wall_box(geoms[cand[c].second], wb, &om) < 0) continue;
        omaxAll = std::min(omaxAll, om);
        boxes.push_back(wb);
        which.push_back(cand[c].second);
    }
    const int n = (int)boxes.size();
    k.nWalls = n;
    k.wallOMax = n > 0 ? (float)omaxAll : 0.0f;      // the margin must hold for every wall
    k.nSlotWalls = 0;
    for (int sl = 0; sl < 6; ++sl) { k.slotTh[sl] = 0.0f; k.slotBit[sl] = 0u; }
    for (int a = 0; a < 3; ++a) k.outerLo[a] = k.outerHi[a] = 0.0f;
    wallGeom.clear();
    if (n == 0) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const WallBox &b : boxes)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)b.lo[a]); hi[a] = std::max(hi[a], (double)b.hi[a]); }
    double diag = 0;
    for (int a = 0; a < 3; ++a) diag += (hi[a] - lo[a]) * (hi[a] - lo[a]);
    diag = std::sqrt(diag);
    const double slack = 2e-6 * (omaxAll + diag);
    // slot of every wall: 2 axis + (high side), or -1
    std::vector<int> slot(n, -1);
    std::vector<double> th(n, 0.0);
    bool taken[6] = {false, false, false, false, false, false};
    for (int w = 0; w < n && std::isfinite(slack); ++w) {            // (largest walls first)
        double best = 0;
        for (int a = 0; a < 3; ++a) {
            const double C = 0.5 * (lo[a] + hi[a]), ext = std::max(hi[a] - lo[a], 1e-30);
            const double gLow = (C - boxes[w].hi[a]) / ext, gHigh = (boxes[w].lo[a] - C) / ext;
            if (gLow > best && !taken[2 * a] && std::isfinite((double)boxes[w].hi[a] + slack)) { best = gLow; slot[w] = 2 * a; th[w] = (double)boxes[w].hi[a] + slack; }
            if (gHigh > best && !taken[2 * a + 1] && std::isfinite((double)boxes[w].lo[a] - slack)) { best = gHigh; slot[w] = 2 * a + 1; th[w] = (double)boxes[w].lo[a] - slack; }
        }
        if (slot[w] >= 0) taken[slot[w]] = true;
    }
    for (int pass = 0; pass < 2; ++pass)                              // walls with a slot first
        for (int w = 0; w < n; ++w)
            if ((slot[w] >= 0) == (pass == 0)) {
                const int idx = (int)wallGeom.size();
                hw[idx] = boxes[w];
                if (slot[w] >= 0) {
                    // rounded towards the interior: a threshold may only make the certificate rarer
                    const float t = (float)th[w];
                    k.slotTh[slot[w]] = (slot[w] & 1) ? ((double)t > th[w] ? std::nextafter(t, -INFINITY) : t) : ((double)t < th[w] ? std::nextafter(t, INFINITY) : t);
                    k.slotBit[slot[w]] = 1u << idx;
                    k.nSlotWalls = idx + 1;
                }
                wallGeom.push_back(which[w]);
            }
    for (int a = 0; a < 3; ++a) {       // rounded outwards
        k.outerLo[a] = std::nextafter((float)lo[a], -INFINITY);
        k.outerHi[a] = std::nextafter((float)hi[a], INFINITY);
    }
}

// n / d for every n < 2^30 as (n * magic) >> shift: with s = ceil(log2 d), shift = 30 + s and magic = ceil(2^shift / d)
// (< 2^31) the error e = magic * d - 2^shift is below d <= 2^s, so n * e < 2^(30 + s) = 2^shift and the quotient is exact
// (Granlund-Montgomery); n * magic < 2^61 fits the 64-bit product.  (The camera-ray bounce divides path indices up to
// pixels x max_batch <= 2^29 by the shard's pixel count; rounds 1-2 used shift = 28 + s, exact only below 2^28.)
void magic_divisor(uint32_t d, uint32_t &magic, uint32_t &shift) {
    if (d <= 1) { magic = 1; shift = 0; return; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    shift = 30 + s;
    magic = (uint32_t)(((1ull << shift) + d - 1) / d);
}

void pack_material(const PtMaterial &m, MaterialDev &d) {
    memset(&d, 0, sizeof d);
    d.color[0] = m.color.x; d.color[1] = m.color.y; d.color[2] = m.color.z;
    d.specColor[0] = m.specularColor.x; d.specColor[1] = m.specularColor.y; d.specColor[2] = m.specularColor.z;
    d.hasReflective = m.hasReflective;
    d.hasRefractive = m.hasRefractive;
    d.ior = m.indexOfRefraction;
    d.emittance = m.emittance;
    d.invIor = 1.0f / d.ior;
    const float q = (1.0f - d.ior) / (1.0f + d.ior);
    d.r0 = q * q;
    d.invSpecExp1 = m.specularExponent > 0.0f ? 1.0f / (m.specularExponent + 1.0f) : 0.0f;
}

// Pixel rectangle from which camera rays can reach a primitive: project the 8 corners of its object-space unit cube
// (which contains the unit-diameter sphere as well; `box`: a mesh's object-space bounds instead) in double precision.  A camera ray is
//     eye + lambda * (view - right * pixLenX * (px - W/2) - up * pixLenY * (py - H/2)),   px in [x, x+1], py in [y, y+1],
// so a world point Q lies on the ray through continuous pixel (px, py) iff  Q - eye = M * (lambda, lambda sx, lambda sy)
// with M = [view | -pixLenX right | -pixLenY up].  The convex hull of the projected corners contains the projection of
// the primitive; its bounding rectangle is widened by 2 pixels.  Any corner not strictly in front of the eye, or a
// singular M, disables the culling for this primitive (whole frame).
// `hull` (optional): the eight projected corners (continuous pixel coordinates) when the rectangle is a real one, else empty
// What the rectangle must contain is not the primitive but every pixel whose camera ray the REFERENCE's test can report as a hit,
// and that test works in fp32 in object space: seen from R object units away (R large for a small, a distant or a flat
// primitive -- the inverse transform magnifies the eye's coordinates by 1 / scale), the object-space origin carries an absolute error
// ~eps R, the normalised object-space direction ~eps (row sums of the inverse transform x the transform's largest singular value), and the sphere's
// radicand (ro . rd)^2 - (ro . ro - 0.25) ~eps R^2.  A ray that misses the exact primitive by less than that can come back as a hit
// (the device sweep found them at once: a 100 : 1 ellipsoid seen from 20 000 object units through a 1.5-degree lens "hit" from
// pixels 60 columns off its projection).  So the box whose corners are projected is the object-space box INFLATED by those errors, with
// factors on first-order bounds (eps = 2^-24; k = 8 for a cube, 128 for a sphere or a mesh's box):
//     A_i  = sum_j |inv_ij| |eye_j| + |inv_i3|        magnitude of the sums behind ro_i          (error of ro_i   <= 3 eps A_i)
//     B_i  = sum_j |inv_ij|                           ... behind (inverseTransform d)_i, |d| <= 1 (error          <= 3 eps B_i)
//     R    = |A| + 1                                  object-space distance over which a direction error acts
//     D_i  = k eps A_i + R (k eps B_i smax + k eps)    displacement of the computed line along axis i (smax >= the transform's largest
//                                                     singular value: |inverseTransform d| >= |d| / smax)
//     cube / mesh box: half extent + 2 D_i (+ 2e-5 R for a mesh: the relative slack of its slab comparisons), all x (1 + 1e-5)
//     sphere:          the cube of half extent  sqrt(1/4 + 512 eps R^2) + 2 |D|  on every axis
// THE MARGIN, stated like certainMiss's (what is bounded, by which factor, what the sweep saw): the bound is on the distance, in object
// space, by which the EXACT half-line of a camera ray may miss the primitive while the reference's fp32 test still reports a hit.
// pt_test_camera_cull_margin measures it per hit -- the exact half-line in double precision against the primitive grown by a
// fraction s of the inflation -- over the 10 500 (camera, primitive set) pairs of the soundness sweep
// (tests/test_gpu_camera_cull.py::test_inflation_margin_of_the_culling_tables, per primitive type).  Round 3's factors (8 eps, 32 eps R^2:
// "safety factors 2 - 3") turned out to leave the worst SPHERE hit of those cases at s ~ 0.85 of the inflation -- a margin of 1.2 x where
// every other shortcut has 40 - 500 x (0.445 over the first 3000 cases; 0.212 over all of them with the terms x 4, measured on the way).
// Round 4 multiplies the sphere's error terms by SIXTEEN: worst observed fraction 0.105 = a margin of 9.5 x in distance (the case is a
// sphere seen from ~10^4 object units, where the inflation is the radicand's term sqrt(512 eps) R: in that term's factor the margin is
// the square, ~90 x).  CUBES keep round 3's factors: of ~10^9 cube hits in the sweep 18 needed any inflation at all, the worst 0.003 of
// it -- a margin of 300 x -- and the x 16 terms, tried first for every type, made Cornell's thin walls (inverse scale 100 on one axis)
// 0.33 units thick in the tables, ten pixels per side: 7 % of the headline throughput for nothing (profiles/exp_r4l.sh).
// For Cornell's walls that is a fraction of a pixel at 1280 x 720; for the ellipsoid above a hundred pixels; when the inflated box
// reaches the eye, a corner is no longer in front of it and the primitive is not culled at all.
void inflated_object_box(const PtGeom &g, const float *eye, const float *box, double lo[3], double hi[3]) {
    const double eps = 5.9604644775390625e-08;                // 2^-24
    double A[3], B[3], smax2 = 0;
    for (int i = 0; i < 3; ++i) {
        A[i] = std::fabs((double)g.inverseTransform[12 + i]);
        B[i] = 0;
        for (int j = 0; j < 3; ++j) {
            const double m = std::fabs((double)g.inverseTransform[j * 4 + i]);
            A[i] += m * std::fabs((double)eye[j]);
            B[i] += m;
            smax2 += (double)g.transform[j * 4 + i] * (double)g.transform[j * 4 + i];       // Frobenius norm >= largest singular value
        }
    }
    const double smax = std::sqrt(smax2);
    const double R = std::sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]) + 1.0;
    // (the factor on the first-order terms: 8 for a cube -- measured margin 300 x, below --, 128 for a sphere and for a mesh's box)
    const double kf = (g.type == PT_CUBE && !box) ? 8.0 : 128.0;
    double D[3], Dn = 0;
    for (int i = 0; i < 3; ++i) {
        D[i] = kf * eps * A[i] + R * (kf * eps * B[i] * smax + kf * eps);
        Dn += D[i] * D[i];
    }
    Dn = std::sqrt(Dn);
    for (int i = 0; i < 3; ++i) {
        double l = box ? box[i] : -0.5, h = box ? box[3 + i] : 0.5;
        if (g.type == PT_SPHERE) {
            const double r = std::sqrt(0.25 + 512 * eps * R * R) + 2 * Dn;
            l = -r; h = r;
        } else {
            const double d = 2 * D[i] + (box ? 2e-5 * R : 0.0);
            l -= d; h += d;
        }
        const double c = 0.5 * (l + h), e = 0.5 * (h - l) * (1 + 1e-5);
        lo[i] = c - e;
        hi[i] = c + e;
    }
}

void project_geom(const PtGeom &g, const KParams &k, int rect[4], const float *box = nullptr, std::vector<std::pair<double, double>> *hull = nullptr) {
    if (hull) hull->clear();
    std::vector<std::pair<double, double>> pts;
    rect[0] = rect[1] = 0;
    rect[2] = k.W - 1;
    rect[3] = k.H - 1;
    double blo[3], bhi[3];
    inflated_object_box(g, k.pos, box, blo, bhi);
    for (int a = 0; a < 3; ++a)
        if (!std::isfinite(blo[a]) || !std::isfinite(bhi[a])) return;
    const double M[3][3] = {{k.view[0], -(double)k.pixLenX * k.right[0], -(double)k.pixLenY * k.up[0]},
                            {k.view[1], -(double)k.pixLenX * k.right[1], -(double)k.pixLenY * k.up[1]},
                            {k.view[2], -(double)k.pixLenX * k.right[2], -(double)k.pixLenY * k.up[2]}};
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    double scale = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(M[r][c]));
    if (!(std::fabs(det) > 1e-12 * scale * scale * scale) || !std::isfinite(det)) return;
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int corner = 0; corner < 8; ++corner) {
        double o[3];
        for (int a = 0; a < 3; ++a) o[a] = ((corner >> a) & 1) ? bhi[a] : blo[a];       // (the inflated unit cube, or a mesh's inflated box)
        double q[3];
        for (int r = 0; r < 3; ++r)
            q[r] = (double)g.transform[0 + r] * o[0] + (double)g.transform[4 + r] * o[1] + (double)g.transform[8 + r] * o[2] +
                   (double)g.transform[12 + r] - (double)k.pos[r];
        // Cramer: (lambda, lambda sx, lambda sy) = M^-1 q
        double c[3];
        for (int col = 0; col < 3; ++col) {
            double A[3][3];
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) A[r][cc] = cc == col ? q[r] : M[r][cc];
            c[col] = (A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                      A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0])) / det;
        }
        const double dist = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        const double vlen = std::sqrt((double)k.view[0] * k.view[0] + (double)k.view[1] * k.view[1] + (double)k.view[2] * k.view[2]);
        if (!(c[0] * vlen > 1e-3 * dist) || !std::isfinite(c[0])) return;     // corner not clearly in front of the eye
        const double px = c[1] / c[0] + k.halfW, py = c[2] / c[0] + k.halfH;
        if (!std::isfinite(px) || !std::isfinite(py)) return;
        xmin = std::min(xmin, px); xmax = std::max(xmax, px);
        ymin = std::min(ymin, py); ymax = std::max(ymax, py);
        pts.emplace_back(px, py);
    }
    if (hull) *hull = pts;
    auto clampi = [](double v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); };
    rect[0] = clampi(std::floor(xmin) - 2, 0, k.W);
    rect[1] = clampi(std::floor(ymin) - 2, 0, k.H);
    rect[2] = clampi(std::ceil(xmax) + 2, -1, k.W - 1);
    rect[3] = clampi(std::ceil(ymax) + 2, -1, k.H - 1);
}

// Pixels of image row y from which camera rays can reach a primitive whose projected corners are `pts`: the projection of
// the primitive lies in the convex hull of the points, a ray of row y passes the image plane at py in [y, y + 1], so the
// candidates are the x-extent of  hull /\ {y - 2 <= py <= y + 3}  widened by 2 pixels (the rectangle's margins, per row).
// The hull's extent inside a horizontal strip is attained at a vertex inside the strip or where an edge -- of the hull, but
// taking every segment between two of the points only adds points of the hull -- crosses one of the strip's two borders.
// false: the strip misses the hull.
bool hull_row_span(const std::vector<std::pair<double, double>> &pts, int y, double &xmin, double &xmax) {
    const double lo = (double)y - 2.0, hi = (double)y + 3.0;
    xmin = INFINITY; xmax = -INFINITY;
    for (size_t i = 0; i < pts.size(); ++i) {
        if (pts[i].second >= lo && pts[i].second <= hi) { xmin = std::min(xmin, pts[i].first); xmax = std::max(xmax, pts[i].first); }
        for (size_t j = i + 1; j < pts.size(); ++j)
            for (double border : {lo, hi}) {
                const double y0 = pts[i].second, y1 = pts[j].second;
                if ((y0 < border) != (y1 < border)) {
                    const double x = pts[i].first + (pts[j].first - pts[i].first) * ((border - y0) / (y1 - y0));
                    xmin = std::min(xmin, x); xmax = std::max(xmax, x);
                }
            }
    }
    return xmin <= xmax;
}

// Screen-space culling of camera rays, everything pt_init derives from the camera and the primitives' transforms:
//   hg[i].rect   pixel rectangle of primitive i (project_geom),
//   sceneRect    their union: camera rays of pixels outside miss everything (whole tiles are skipped there),
//   rowOff/rowIdx per image row y the primitives whose rectangle covers it, each with the pixels of that row inside the convex
//                hull of its projected corners (hull_row_span): entries {primitive, x0 | x1 << 16}, file order; empty when the
//                frame is too large for the tables (the kernels then use the rectangles).
// Soundness: a primitive lies inside its object-space box, the box inside the convex hull of its eight corners, and a camera ray
// of pixel (x, y) passes the image plane at continuous coordinates in [x, x + 1] x [y, y + 1]; every hit of the reference's tests
// (src/intersections.h:47-143) is a geometric hit of the primitive up to their ~1e-6 relative rounding, which the two pixels of
// margin on every side exceed by orders of magnitude at any supported width (2 px of a 32768-px row is still 6e-5 of the
// image plane).  Whatever cannot be bounded -- a corner not clearly in front of the eye, a singular camera basis -- disables
// the culling for that primitive (whole frame).  `boxes[i]`: object-space bounds of a mesh (6 floats), nullptr otherwise.
// `off`: no culling at all (PT_AMD_NO_CAMERA_CULL, tests only: the reference semantics the culled render must reproduce).
struct CameraCull {
    int sceneRect[4];
    std::vector<int> rowOff, rowIdx;
};
void build_camera_cull(const PtGeom *geoms, int ngeoms, const KParams &k, bool off, const std::vector<const float *> &boxes,
                       std::vector<GeomDev> &hg, CameraCull &cc) {
    const int Wd = k.W, H = k.H;
    cc.sceneRect[0] = cc.sceneRect[1] = 0x7fffffff;   // empty union: a scene without primitives is never entered
    cc.sceneRect[2] = cc.sceneRect[3] = -1;
    cc.rowOff.clear();
    cc.rowIdx.clear();
    std::vector<std::vector<std::pair<double, double>>> hulls(ngeoms ? ngeoms : 1);   // projected corners per primitive
    for (int i = 0; i < ngeoms; ++i) {
        project_geom(geoms[i], k, hg[i].rect, boxes[i], &hulls[i]);
        if (off) {        // (thin lens: rays start anywhere on the lens, the pinhole projection bounds nothing)
            hg[i].rect[0] = hg[i].rect[1] = 0;
            hg[i].rect[2] = Wd - 1;
            hg[i].rect[3] = H - 1;
            hulls[i].clear();
        }
        cc.sceneRect[0] = std::min(cc.sceneRect[0], hg[i].rect[0]);
        cc.sceneRect[1] = std::min(cc.sceneRect[1], hg[i].rect[1]);
        cc.sceneRect[2] = std::max(cc.sceneRect[2], hg[i].rect[2]);
        cc.sceneRect[3] = std::max(cc.sceneRect[3], hg[i].rect[3]);
    }
    if (off || !((long long)H * ngeoms < (1ll << 26) && Wd <= 32768)) return;
    cc.rowOff.resize(H + 1);
    for (int y = 0; y < H; ++y) {
        cc.rowOff[y] = (int)(cc.rowIdx.size() / 2);
        for (int i = 0; i < ngeoms; ++i) {
            if (!(y >= hg[i].rect[1] && y <= hg[i].rect[3] && hg[i].rect[0] <= hg[i].rect[2])) continue;
            int x0 = hg[i].rect[0], x1 = hg[i].rect[2];
            if (!hulls[i].empty()) {
                double xmin, xmax;
                if (!hull_row_span(hulls[i], y, xmin, xmax)) continue;
                x0 = std::max(x0, (int)std::max(std::floor(xmin) - 2.0, -1.0e9));
                x1 = std::min(x1, (int)std::min(std::ceil(xmax) + 2.0, 1.0e9));
                if (x0 > x1) continue;
            }
            cc.rowIdx.push_back(i);
            cc.rowIdx.push_back(x0 | (x1 << 16));
        }
    }
    cc.rowOff[H] = (int)(cc.rowIdx.size() / 2);
    if (cc.rowIdx.empty()) { cc.rowIdx.push_back(0); cc.rowIdx.push_back(0); }
}

// host mirrors of the glm ops used for the camera basis (same op order as ptd::)
struct H3 { float x, y, z; };
H3 hcross(H3 x, H3 y) { return H3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
H3 hnormalize(H3 a) {
    float d = a.x * a.x + a.y * a.y + a.z * a.z;
    float s = 1.0f / std::sqrt(d);
    return H3{a.x * s, a.y * s, a.z * s};
}

// the camera constants of spec S2 (KParams: basis, pixel lengths, frame size), derived once on the host
void camera_params(const PtCamera &cam, KParams &k) {
    const int Wd = cam.resolution[0], H = cam.resolution[1];
    const H3 view{cam.view.x, cam.view.y, cam.view.z}, up{cam.up.x, cam.up.y, cam.up.z};
    const H3 right = hnormalize(hcross(view, up));
    k.view[0] = view.x; k.view[1] = view.y; k.view[2] = view.z;
    k.up[0] = up.x; k.up[1] = up.y; k.up[2] = up.z;
    k.right[0] = right.x; k.right[1] = right.y; k.right[2] = right.z;
    k.pos[0] = cam.position.x; k.pos[1] = cam.position.y; k.pos[2] = cam.position.z;
    const float kPI = 3.1415926535897932384626422832795028841971f;   // src/utilities.h:12
    const float ys = std::tan(cam.fov[1] * (kPI / 180));             // src/scene.cpp:133 convention
    const float xs = (ys * Wd) / H;
    k.pixLenX = (2.0f * xs) / (float)Wd;
    k.pixLenY = (2.0f * ys) / (float)H;
    k.halfW = (float)Wd * 0.5f;
    k.halfH = (float)H * 0.5f;
    k.W = Wd; k.H = H;
    k.shardRank = 0; k.shardCount = 1;
}

int resolve_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &ms, long long &n) {
    for (auto &pr : v) {
        float t = 0;
        HIPCHECK(hipEventSynchronize(pr.second));
        HIPCHECK(hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        n += 1;
        S.evFree.push_back(pr.first);
        S.evFree.push_back(pr.second);
    }
    v.clear();
    return PT_OK;
}

// The instantiation of k_bounce a launch takes: FIRST (camera rays), MANY (per-lane sphere lists: scenes with more than
// kBinMax spheres), DOF (thin lens: the camera-ray launch only), MESH (scenes with triangle meshes).
template <bool F, bool M, bool D, bool ME, bool PL = false>
const void *kb() { return reinterpret_cast<const void *>(k_bounce<F, M, D, ME, PL>); }
const void *bounce_kernel(bool first, bool dof) {
    // (plain scenes -- diffuse / emissive / perfect-mirror materials, no README extra: the instantiations without the rarer branches)
    if (S.plain && !S.mesh && !S.many && !dof) return first ? kb<true, false, false, false, true>() : kb<false, false, false, false, true>();
    if (S.mesh && S.many) return first ? (dof ? kb<true, true, true, true>() : kb<true, true, false, true>()) : kb<false, true, false, true>();
    if (S.mesh) return first ? (dof ? kb<true, false, true, true>() : kb<true, false, false, true>()) : kb<false, false, false, true>();
    if (first && dof) return S.many ? kb<true, true, true, false>() : kb<true, false, true, false>();
    if (first) return S.many ? kb<true, true, false, false>() : kb<true, false, false, false>();
    return S.many ? kb<false, true, false, false>() : kb<false, false, false, false>();
}

int launch_bounce(Slot &sl, int iter, int batch, int depth, bool lastBounce, float *contrib, bool nextIsLast = false) {
    const PathPool in = pool(sl, (depth - 1) & 1);
    const PathPool out = pool(sl, depth & 1);
    // chunk-list entries carry the serial number of the launch that wrote them (never 0)
    const uint32_t genIn = sl.gen[(depth - 1) & 1];
    if (++S.launchSerial == 0u) ++S.launchSerial;
    const uint32_t genOut = sl.gen[depth & 1] = S.launchSerial;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (S.flags & PT_FLAG_KERNEL_TIMING) {
        for (hipEvent_t *e : {&e0, &e1}) {
            if (!S.evFree.empty()) { *e = S.evFree.back(); S.evFree.pop_back(); }
            else HIPCHECK(hipEventCreate(e));
        }
        HIPCHECK(hipEventRecord(e0, sl.stream));
    }
    BounceArgs ba;
    ba.prm = S.prm;
    ba.iter = iter; ba.batch = batch; ba.depth = depth; ba.lastBounce = lastBounce ? 1 : 0; ba.parity = sl.parity;
    ba.genIn = genIn; ba.genOut = genOut;
    ba.in = in; ba.out = out;
    ba.tile.inBase = in.base; ba.tile.inList = in.list; ba.tile.inCap = in.cap;
    ba.tile.genIn = genIn; ba.tile.poolChunks = (uint32_t)S.prm.poolChunks; ba.tile.chunkShift = (uint32_t)S.prm.chunkShift;
    ba.tile.skipNonCandidates = (lastBounce && S.prm.emittersBinned) ? 1u : 0u;
    ba.tile.hot = (lastBounce ? kHotLast : 0u) | (S.prm.allClassified ? kHotAllClassified : 0u) | (contrib ? kHotContrib : 0u) |
                  ((S.prm.directDepth != 0 && depth == S.prm.directDepth && S.prm.nEmit > 0) ? kHotToLight : 0u) |
                  (S.prm.contribLocal ? kHotContribLocal : 0u) | ((uint32_t)S.prm.nWalls << 8) | ((uint32_t)S.prm.nSlotWalls << 11) |
                  ((uint32_t)S.prm.nBinned << 14) | ((uint32_t)S.prm.nmats << 20);
    // sphere clusters: the queue that ENTERS the last bounce carries other candidate bits (k_bounce: kHotWritesLastBits) -- that launch only asks
    // whether a path ends on an emitter, and with every emitter binned it visits the tiles of the binned primitives' candidates alone
    const bool lastBits = S.many && !S.mesh && S.prm.sphOMax > 0.0f && S.prm.emittersBinned;
    if (lastBits && nextIsLast) ba.tile.hot |= kHotWritesLastBits;
    if (lastBits && lastBounce && depth > 1) ba.tile.hot |= kHotReadsLastBits;
    ba.ctrl = sl.ctrl; ba.ggeoms = S.dgeoms; ba.gmats = S.dmats; ba.ghit = S.dGeomHit; ba.contrib = contrib; ba.hitMask = sl.hitMask;
    ba.sphCull = S.dSphCull; ba.classIdx = S.dClassIdx;
    ba.rowOff = S.dRowOff; ba.rowIdx = S.dRowIdx;
    ba.walls = S.dwalls;
    ba.meshRecs = S.dMeshRecs;
    ba.hostFault = S.hostFaultDev;
    void *kargs[] = {&ba};
    const bool first = depth == 1;
    HIPCHECK(hipLaunchKernel(bounce_kernel(first, first && S.dof), dim3(first ? S.gridFirst : S.grid), dim3(kBlock), kargs, first ? S.ldsBytes : S.ldsBytesNext, sl.stream));
    if (e0) {
        HIPCHECK(hipEventRecord(e1, sl.stream));
        S.evBounce.emplace_back(e0, e1);
        if (S.evBounce.size() > 8192) {
            int rc = resolve_events(S.evBounce, S.msBounce, S.nBounce);
            if (rc) return rc;
        }
    }
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

// scan library workspace: the chunk totals, one buffer per stream (calls on different streams may overlap; calls on one
// stream are ordered, so they share it)
struct ScanWs {
    uint32_t *partial = nullptr;     // [kScanChunksMax + 1]
};
std::map<hipStream_t, ScanWs> g_scan;
std::mutex g_scanMutex;              // the scan library may be called from several host threads (one stream each)

// The caller HOLDS g_scanMutex from here until its launches that use the workspace are enqueued: pt_free, which releases the
// workspaces, takes the same lock first and then waits for the device -- so a workspace is never freed between its look-up and the
// kernels that use it (round 3 handed the pointer out of the lock: a second host thread could enqueue on freed memory).
int scan_ws(hipStream_t st, ScanWs **out) {
    ScanWs &W = g_scan[st];          // (std::map: the reference stays valid while other streams are added)
    if (!W.partial) HIPCHECK(hipMalloc(&W.partial, (size_t)(kScanChunksMax + 1) * sizeof(uint32_t)));
    *out = &W;
    return PT_OK;
}
// releases every stream's workspace: the PUBLIC pt_free and the process's exit only -- not the pt_free inside pt_init (the
// reference's Free -> Init restart), which other host threads' scans must survive.  Under the lock: no scan call is between its
// look-up and its launches; then everything enqueued is waited for, then freed.
void scan_release() {
    std::lock_guard<std::mutex> lock(g_scanMutex);
    if (g_scan.empty()) return;
    (void)hipDeviceSynchronize();
    for (auto &kv : g_scan)
        if (kv.second.partial) (void)hipFree(kv.second.partial);
    g_scan.clear();
}
bool scan_in_use() {
    std::lock_guard<std::mutex> lock(g_scanMutex);
    return !g_scan.empty();
}
// the array as chunks of whole tiles: at most kScanChunksMax of them
void scan_chunks(long long n, long long *tilesPerChunk, int *chunks) {
    const long long tiles = (n + kScanTile - 1) / kScanTile;
    const long long per = (tiles + kScanChunksMax - 1) / kScanChunksMax;
    *tilesPerChunk = per < 1 ? 1 : per;
    *chunks = (int)((tiles + *tilesPerChunk - 1) / *tilesPerChunk);
}

// wait for every stream the renderer uses
int sync_all() {
    for (int i = 0; i < S.nslots; ++i) HIPCHECK(hipStreamSynchronize(S.slot[i].stream));
    HIPCHECK(hipStreamSynchronize(S.stream));
    return PT_OK;
}

int check_device_fault() {
    int rc = sync_all();
    if (rc) return rc;
    for (int i = 0; i < S.nslots; ++i) {
        uint32_t err = 0;
        HIPCHECK(hipMemcpy(&err, &S.slot[i].ctrl->error, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(PT_ERR_DEVICE, "device fault 0x%x:%s%s (results of this render are void; re-init)", err,
                             (err & kFaultPoolExhausted) ? " path pool exhausted" : "", (err & kFaultReserveTimeout) ? " chunk reservation timed out" : "");
    }
    return PT_OK;
}

// pt_readback / pt_readback_rgba8 have just synchronised the caller's stream: every launch whose radiance the image holds has
// finished, and a fault any of them raised is in the host-visible copy of the fault word -- no device-to-host copy on the
// good path.  A faulted render is then reported like pt_sync does (the image has been copied all the same; it is void).
int readback_fault() {
    if (S.hostFault && *(volatile uint32_t *)S.hostFault != 0u) {
        int rc = check_device_fault();
        if (rc) return rc;
        return fail(PT_ERR_DEVICE, "device fault (results of this render are void; re-init)");
    }
    return PT_OK;
}

int persistent_grid(const void *kernel, size_t lds, int *grid) {
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, dev));
    int perCU = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, kBlock, lds));
    if (perCU < 1) perCU = 1;
    // All four instantiations fit eight workgroups per CU (<= 80 SGPRs, <= 64 VGPRs; the sphere-list variants seven).  A
    // launch that has the GPU to itself (pipeline_depth 1) is fastest with all of them (0.253 ms against 0.274 with six);
    // with batches in flight on neighbouring streams six per launch is better (131.2 G paths/s against 128.9 with eight):
    // the two free wave slots per SIMD go to the neighbouring batch's launches, which fill this launch's tail.
    int cap = S.nslots > 1 ? 6 : 8;
    if (const char *e = getenv("PT_AMD_BLOCKS_PER_CU")) cap = atoi(e);   // experiments only
    if (perCU > cap) perCU = cap;
    *grid = prop.multiProcessorCount * perCU;
    if (const char *e = getenv("PT_AMD_MAX_GRID")) *grid = std::max(1, std::min(*grid, atoi(e)));   // tests: a small or partitioned device's grid
    return PT_OK;
}

constexpr int kIterEnd = 1 << 22;     // iterations are 1 .. kIterEnd - 1 (seed bits, pathtrace.cu:43)

// the bounce launches of iterations first_iter .. first_iter + count - 1 as one wavefront batch on the slot's stream; the
// radiance they find is parked in the slot's buffers until a commit consumes it
int trace_batch(Slot &sl, int first_iter, int count) {
    // the slot's radiance buffers must have been consumed by the commit of its previous batch
    HIPCHECK(hipStreamWaitEvent(sl.stream, sl.evCommitted, 0));
    const int D = S.prm.traceDepth;
    for (int d = 1; d <= D; ++d) {
        int rc = launch_bounce(sl, first_iter, count, d, d == D, sl.contrib, d + 1 == D);
        if (rc) {
            // a launch failed with part of the batch enqueued: counters, parity and radiance buffers are half-updated, so the
            // renderer refuses further work until it is re-initialised (pt_free still releases everything)
            if (d > 1) S.init = false;
            return rc;
        }
    }
    sl.parity ^= 1;   // the last launch re-armed the other half of the slot's counters
    HIPCHECK(hipEventRecord(sl.evDone, sl.stream));
    return PT_OK;
}

// iterations [b0, b1) of the slot's batch of `count` into the accumulator (or nowhere: `discard`), on the caller's stream
int commit_range(Slot &sl, int count, int b0, int b1, bool discard) {
    if (S.nLocal > 0) {
        hipLaunchKernelGGL(k_commit, dim3((S.nLocal + kBlock - 1) / kBlock), dim3(kBlock), 0, S.stream, S.prm, S.image, sl.contrib, sl.hitMask,
                           count, (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) ? 1 : 0, b0, b1, discard ? 1 : 0);
        HIPCHECK(hipGetLastError());
    }
    return PT_OK;
}

// PT_FLAG_TRACE_AHEAD: trace the batch that starts at iteration `first` into the next slot of the rotation and park it
int trace_ahead(int first) {
    const int count = std::min(S.maxBatch, kIterEnd - first);
    const int slot = (int)(S.seq % S.nslots);
    int rc = trace_batch(S.slot[slot], first, count);
    if (rc) return rc;
    S.ahead.push_back({slot, first, count, 0, false});
    S.seq += 1;
    return PT_OK;
}

// ... and drop what is parked: the caller asked for something else.  The iterations not yet committed are consumed without
// being added, which leaves the slots' buffers zeroed as every batch expects to find them.
int discard_ahead() {
    for (const State::Parked &p : S.ahead) {
        Slot &sl = S.slot[p.slot];
        HIPCHECK(hipStreamWaitEvent(S.stream, sl.evDone, 0));
        int rc = commit_range(sl, p.count, p.next, p.count, true);
        if (rc) return rc;
        HIPCHECK(hipEventRecord(sl.evCommitted, S.stream));
    }
    S.ahead.clear();
    return PT_OK;
}

// Process exit with work in flight (an exception in the host between pt_iterate and pt_free, an interpreter that is torn down
// with a live renderer): the streams are drained and everything is released BEFORE the HIP runtime's own exit handlers run --
// this handler is registered after the library's first HIP call, and exit handlers run in reverse order of registration --
// so that no launch of this process is still executing when its queues, its code object and its memory go away.
extern "C" void pt_free(void);
void free_renderer();
void exit_handler() { pt_free(); }
void register_exit_handler() {
    static bool done = false;
    if (!done) {
        done = true;
        atexit(exit_handler);
    }
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        HIPCHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
        return PT_OK;
    }
};

}  // namespace

// =====================================================================================================
// C ABI (the library is built with -fvisibility=hidden: these entry points are all it exports)
// =====================================================================================================
#pragma GCC visibility push(default)
extern "C" {

const char *pt_last_error(void) { return g_err.c_str(); }
int pt_device_count(void) { return count_devices(); }

void pt_free(void) {
    // the scan library's per-stream workspaces (allocated on first use, with or without a renderer)
    scan_release();
    free_renderer();
}

}  // extern "C"
#pragma GCC visibility pop
namespace {
// everything pt_init allocated (pt_free, and pt_init's own restart)
void free_renderer() {
    // pathtraceFree before the first Init (src/main.cpp:91-94) must be a no-op
    if (!S.init && !S.image && !S.dgeoms && S.nslots == 0 && !S.hostFault) return;
    if (S.device >= 0 && S.nslots > 0) (void)hipSetDevice(S.device);     // (a host that switched devices in between)
    for (int i = 0; i < kMaxSlots; ++i)
        if (S.slot[i].stream) (void)hipStreamSynchronize(S.slot[i].stream);
    (void)hipStreamSynchronize(S.stream);
    for (auto &pr : S.evBounce) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    S.evBounce.clear();
    for (hipEvent_t e : S.evFree) (void)hipEventDestroy(e);
    S.evFree.clear();
    for (int i = 0; i < kMaxSlots; ++i) {
        Slot &sl = S.slot[i];
        for (int k = 0; k < 2; ++k) {
            if (sl.pathbuf[k]) (void)hipFree(sl.pathbuf[k]);
            if (sl.chunkList[k]) (void)hipFree(sl.chunkList[k]);
        }
        if (sl.ctrl) (void)hipFree(sl.ctrl);
        if (sl.contrib) (void)hipFree(sl.contrib);
        if (sl.hitMask) (void)hipFree(sl.hitMask);
        if (sl.evDone) (void)hipEventDestroy(sl.evDone);
        if (sl.evCommitted) (void)hipEventDestroy(sl.evCommitted);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (S.pinnedHost) (void)hipHostUnregister(S.pinnedHost);
    if (S.hostFault) (void)hipHostFree(S.hostFault);
    if (S.ownImage && S.image) (void)hipFree(S.image);
    if (S.dgeoms) (void)hipFree(S.dgeoms);
    if (S.dGeomHit) (void)hipFree(S.dGeomHit);
    if (S.dmats) (void)hipFree(S.dmats);
    if (S.dwalls) (void)hipFree(S.dwalls);
    if (S.dSphCull) (void)hipFree(S.dSphCull);
    if (S.dClassIdx) (void)hipFree(S.dClassIdx);
    if (S.dRowOff) (void)hipFree(S.dRowOff);
    if (S.dRowIdx) (void)hipFree(S.dRowIdx);
    if (S.dMeshRecs) (void)hipFree(S.dMeshRecs);
    S = State();
}
}  // namespace
#pragma GCC visibility push(default)
extern "C" {

int pt_set_meshes(const PtMesh *meshes, int nmeshes) {
    if (nmeshes < 0 || (nmeshes && !meshes)) return fail(PT_ERR_INVALID, "pt_set_meshes: null argument");
    for (int i = 0; i < nmeshes; ++i) {
        if (meshes[i].geom < 0 || meshes[i].ntris < 1 || !meshes[i].tris) return fail(PT_ERR_INVALID, "pt_set_meshes: mesh %d is empty", i);
        for (int j = 0; j < i; ++j)
            if (meshes[j].geom == meshes[i].geom) return fail(PT_ERR_INVALID, "pt_set_meshes: geom %d given twice", meshes[i].geom);
        for (size_t q = 0; q < 9 * (size_t)meshes[i].ntris; ++q)
            if (!std::isfinite(meshes[i].tris[q])) return fail(PT_ERR_INVALID, "pt_set_meshes: mesh %d holds a non-finite coordinate", i);
        if (meshes[i].normals)
            for (size_t q = 0; q < 9 * (size_t)meshes[i].ntris; ++q)
                if (!std::isfinite(meshes[i].normals[q])) return fail(PT_ERR_INVALID, "pt_set_meshes: mesh %d holds a non-finite normal", i);
    }
    g_meshes.clear();
    for (int i = 0; i < nmeshes; ++i) {
        ptm::HostMesh m;
        m.geom = meshes[i].geom;
        m.tris.assign(meshes[i].tris, meshes[i].tris + 9 * (size_t)meshes[i].ntris);
        if (meshes[i].normals) m.normals.assign(meshes[i].normals, meshes[i].normals + 9 * (size_t)meshes[i].ntris);
        if (meshes[i].materials) m.mats.assign(meshes[i].materials, meshes[i].materials + (size_t)meshes[i].ntris);
        g_meshes.push_back(std::move(m));
    }
    return PT_OK;
}

int pt_init(const PtCamera *cam, const PtGeom *geoms, int ngeoms, const PtMaterial *mats, int nmats, int traceDepth,
            const PtOptions *opts) {
    if (!cam || ngeoms < 0 || nmats < 0 || (ngeoms && !geoms) || (nmats && !mats))
        return fail(PT_ERR_INVALID, "pt_init: null argument");
    if (cam->resolution[0] <= 0 || cam->resolution[1] <= 0) return fail(PT_ERR_INVALID, "pt_init: bad resolution");
    const bool direct = opts && (opts->flags & PT_FLAG_DIRECT_LIGHTING);
    if (traceDepth < 1 || traceDepth + (direct ? 1 : 0) > PT_MAX_DEPTH)
        return fail(PT_ERR_INVALID, "pt_init: traceDepth must be 1..%d", PT_MAX_DEPTH - (direct ? 1 : 0));
    if (opts && (!(opts->lens_radius >= 0.0f) || (opts->lens_radius > 0.0f && !(opts->focal_distance > 0.0f))))
        return fail(PT_ERR_INVALID, "pt_init: lens_radius must be >= 0 and focal_distance > 0 with a lens");
    if ((long long)cam->resolution[0] * cam->resolution[1] > (1ll << 30)) return fail(PT_ERR_INVALID, "pt_init: frame too large");
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_SPHERE && geoms[i].type != PT_CUBE && geoms[i].type != PT_MESH) return fail(PT_ERR_INVALID, "pt_init: geom %d has unknown type", i);
        if (geoms[i].type == PT_MESH && !mesh_of(i)) return fail(PT_ERR_INVALID, "pt_init: geom %d is a mesh without triangles (pt_set_meshes)", i);
        if (geoms[i].materialid < 0 || geoms[i].materialid >= nmats) return fail(PT_ERR_INVALID, "pt_init: geom %d references material %d", i, geoms[i].materialid);
    }
    for (const ptm::HostMesh &m : g_meshes)
        if (m.geom < 0 || m.geom >= ngeoms || geoms[m.geom].type != PT_MESH)
            return fail(PT_ERR_INVALID, "pt_init: triangles registered for geom %d, which is not a mesh of this scene (pt_set_meshes)", m.geom);
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "pt_init: no HIP device (this library has no CPU fallback)");
    register_exit_handler();
    free_renderer();

    PtOptions o;
    memset(&o, 0, sizeof o);
    o.shard_count = 1;
    o.device = -1;
    if (opts) o = *opts;
    if (o.shard_count < 1 || o.shard_rank < 0 || o.shard_rank >= o.shard_count) return fail(PT_ERR_INVALID, "pt_init: bad shard %d/%d", o.shard_rank, o.shard_count);
    if (o.pipeline_depth < 0 || o.pipeline_depth > kMaxSlots) return fail(PT_ERR_INVALID, "pt_init: pipeline_depth must be 0..%d", kMaxSlots);
    if (o.max_batch < 0 || o.max_batch > PT_MAX_BATCH) return fail(PT_ERR_INVALID, "pt_init: max_batch must be 0..%d", PT_MAX_BATCH);
    if (o.device >= 0) HIPCHECK(hipSetDevice(o.device));
    HIPCHECK(hipGetDevice(&S.device));
    S.stream = (hipStream_t)o.stream;
    S.flags = o.flags;
    S.cam = *cam;

    HIPCHECK(hipHostMalloc((void **)&S.hostFault, sizeof(uint32_t), hipHostMallocMapped));
    *S.hostFault = 0u;
    HIPCHECK(hipHostGetDevicePointer((void **)&S.hostFaultDev, S.hostFault, 0));

    const int Wd = cam->resolution[0], H = cam->resolution[1];
    S.P = Wd * H;
    const int rows = H > o.shard_rank ? (H - o.shard_rank + o.shard_count - 1) / o.shard_count : 0;
    S.nLocal = rows * Wd;

    KParams &k = S.prm;
    memset(&k, 0, sizeof k);
    camera_params(*cam, k);
    const H3 view{cam->view.x, cam->view.y, cam->view.z};
    k.shardRank = o.shard_rank; k.shardCount = o.shard_count;
    k.nLocal = S.nLocal;
    magic_divisor((uint32_t)Wd, k.magicW, k.shiftW);
    magic_divisor((uint32_t)o.shard_count, k.magicS, k.shiftS);
    k.contribLocal = (o.shard_count > 1 && (long long)Wd * H < (1ll << 27)) ? 1 : 0;   // (the multiply-shift divisions hold below 2^27)
    // camera-ray tiles lie on rows padded to a multiple of the tile size (KParams::Wp)
    k.Wp = (Wd + kBlock - 1) / kBlock * kBlock;
    if ((long long)rows * k.Wp >= (1ll << 30)) return fail(PT_ERR_INVALID, "pt_init: frame too large (rows x padded width must stay below 2^30)");
    k.nLocalPad = rows * k.Wp;
    magic_divisor((uint32_t)k.Wp, k.magicWp, k.shiftWp);
    magic_divisor((uint32_t)std::max(k.nLocalPad, 1), k.magicN, k.shiftN);
    for (uint32_t d : {(uint32_t)Wd, (uint32_t)k.Wp, (uint32_t)std::max(k.nLocalPad, 1)}) {       // self-check on the edges of every quotient range
        uint32_t m, sh;
        magic_divisor(d, m, sh);
        for (uint64_t q = 0; q * d < (1ull << 30); q = q < 64 ? q + 1 : q * 2 + 1)
            for (uint64_t n : {q * d, q * d + d - 1, (uint64_t)((1ull << 30) - 1) - q})
                if (n < (1ull << 30) && (uint32_t)((n * m) >> sh) != (uint32_t)(n / d))
                    return fail(PT_ERR_INVALID, "pt_init: magic division self-check failed for d=%u n=%llu", d, (unsigned long long)n);
    }
    k.ngeoms = ngeoms; k.nmats = nmats;
    // direct lighting: bounce `traceDepth` aims its diffuse scatter at a light and one more launch collects
    k.traceDepth = traceDepth + (direct ? 1 : 0);
    k.directDepth = direct ? traceDepth : 0;
    k.nEmit = 0;
    for (int i = 0; i < ngeoms && k.nEmit < kEmitMax; ++i)
        if (geoms[i].type != PT_MESH && mats[geoms[i].materialid].emittance > 0.0f) {   // (direct lighting samples unit cubes: not meshes)
            const PtVec3 sc = geoms[i].scale;
            const float xx = sc.x * sc.x, yy = sc.y * sc.y, zz = sc.z * sc.z;
            const float xy = xx + yy;
            k.emitRho2[k.nEmit] = (xy + zz) * 0.25f;
            k.emitGeom[k.nEmit++] = i;
        }
    k.lensRadius = o.lens_radius;
    k.focalDistance = o.focal_distance;
    {
        const H3 vn = hnormalize(view);
        k.viewN[0] = vn.x; k.viewN[1] = vn.y; k.viewN[2] = vn.z;
    }
    S.dof = o.lens_radius > 0.0f;
    // plain: nothing in the scene takes the scatter's rarer branches (PT_AMD_NO_PLAIN: experiments / tests only)
    S.plain = !direct && !(getenv("PT_AMD_NO_PLAIN") && atoi(getenv("PT_AMD_NO_PLAIN")));
    for (int i = 0; i < nmats; ++i)
        if (mats[i].hasRefractive > 0.0f || (mats[i].hasReflective > 0.0f && mats[i].specularExponent > 0.0f)) S.plain = false;

    if (o.accum_dev) {
        S.image = o.accum_dev;
        S.ownImage = false;
    } else {
        const size_t n = (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) ? (size_t)(S.nLocal > 0 ? S.nLocal : 1) : (size_t)S.P;
        HIPCHECK(hipMalloc(&S.image, n * 3 * sizeof(float)));
        S.ownImage = true;
        HIPCHECK(hipMemsetAsync(S.image, 0, n * 3 * sizeof(float), S.stream));
    }
    // Path pools: a bounce's queue is kSeg = kCls x kSub segments, each a list of chunks handed out on demand, one ahead of
    // their use (ptk::reserveRun).  At most nLocal * maxBatch paths are alive; every segment may end in a partly filled
    // chunk and holds one chunk installed ahead: ceil(paths / chunk) + 2 kSeg chunks always suffice, whatever the
    // distribution over the classes (+ the trash chunk 0).  Chunk size: a power of two, at least 2048 (rounds 1-3: ~1/1024 of the
    // paths, so that the slack stayed around 10 % while a chunk outlasts the appends of one memory round trip).
    S.maxBatch = o.max_batch > 0 ? o.max_batch : 1;
    {   // a path carries pixelIndex | batch index << pixBits in ONE word (ptk::PathC)
        int pixBits = 1, batchBits = 0;
        while (((long long)k.W * k.H - 1) >> pixBits) ++pixBits;
        while ((S.maxBatch - 1) >> batchBits) ++batchBits;
        if (pixBits + batchBits > 32)
            return fail(PT_ERR_INVALID, "pt_init: %d x %d pixels and max_batch %d need %d + %d bits of a path's 32-bit index word: lower max_batch", k.W, k.H,
                        S.maxBatch, pixBits, batchBits);
        k.pixBits = pixBits;
    }
    if (S.maxBatch == 1) S.flags &= ~PT_FLAG_TRACE_AHEAD;   // nothing to trace ahead with: every call traces its own iteration
    // slots are 32-bit element indices with 32-bit byte offsets: paths per pool must stay below 2^30
    const long long maxPaths = (long long)S.nLocal * S.maxBatch;
    if (maxPaths > (1ll << 29)) return fail(PT_ERR_INVALID, "pt_init: max_batch x pixels too large (limit 2^29 paths per batch)");
    if ((long long)k.nLocalPad * S.maxBatch >= (1ll << 30))     // (the camera-ray tiles' index space: rows padded to the tile size)
        return fail(PT_ERR_INVALID, "pt_init: max_batch x rows x padded width too large (limit 2^30)");
    S.numTilesMax = (int)((maxPaths + kBlock - 1) / kBlock) + kSeg;
    // (round 4: ~1/256 of the paths, at most 2^18, where it was 1/1024 and 2^17 -- a run that opens a new chunk pays a dependent
    // look-up of the chunk list INSIDE the reservation's window, and four times fewer of them are +2 % on C2, +4 % on the closed box
    // (profiles/r04_chunk_size_sweep.txt); the slack of 2 kSeg chunks then doubles a mid-sized pool, which 288 GB shrug off)
    k.chunkShift = kMinChunkShift;
    while (k.chunkShift < 18 && (maxPaths >> k.chunkShift) > 256) ++k.chunkShift;
    if (const char *e = getenv("PT_AMD_CHUNK_SHIFT")) {      // experiments only
        const int v = atoi(e);
        if (v >= kMinChunkShift && v <= 20) k.chunkShift = v;
    }
    const long long chunkPaths = 1ll << k.chunkShift;
    S.poolChunks = (int)((maxPaths + chunkPaths - 1) / chunkPaths) + 2 * kSeg + 1;
    if (const char *e = getenv("PT_AMD_POOL_CHUNKS")) {      // tests only: an undersized pool must fail loudly (PT_ERR_DEVICE)
        const int v = atoi(e);
        if (v >= kSeg + 2) S.poolChunks = v;
    }
    k.poolChunks = S.poolChunks;
    const size_t cap = (size_t)S.poolChunks << k.chunkShift;
    // Iterations are independent (RNG keyed on pixel/iteration/depth), so up to `nslots` of them are in flight
    // on their own streams; the small late-bounce launches of one overlap the big early launches of the next.
    S.nslots = o.pipeline_depth > 0 ? o.pipeline_depth : 3;
    {   // the memory budget BEFORE the first large allocation: a batch that does not fit fails here, with nothing to undo
        const size_t cpx = k.contribLocal ? (size_t)(S.nLocal > 0 ? S.nLocal : 1) : (size_t)S.P;
        const size_t perSlot = 2 * (cap * kNumArrays * sizeof(float) + (size_t)kSeg * S.poolChunks * sizeof(unsigned long long)) + sizeof(Ctrl) +
                               (size_t)S.maxBatch * cpx * 3 * sizeof(float) + (size_t)((S.maxBatch + 31) / 32) * cpx * sizeof(uint32_t);
        const size_t need = perSlot * (size_t)S.nslots;
        size_t freeB = 0, totalB = 0;
        HIPCHECK(hipMemGetInfo(&freeB, &totalB));
        if (need > freeB)
            return fail(PT_ERR_HIP, "pt_init: %.2f GB of path pools and radiance buffers (max_batch %d x pipeline_depth %d) exceed the %.2f GB of free "
                        "device memory: lower max_batch or pipeline_depth", need / 1e9, S.maxBatch, S.nslots, freeB / 1e9);
    }
    for (int i = 0; i < S.nslots; ++i) {
        Slot &sl = S.slot[i];
        HIPCHECK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIPCHECK(hipMalloc(&sl.pathbuf[b], cap * kNumArrays * sizeof(float)));
            HIPCHECK(hipMalloc(&sl.chunkList[b], (size_t)kSeg * S.poolChunks * sizeof(unsigned long long)));
            HIPCHECK(hipMemset(sl.chunkList[b], 0, (size_t)kSeg * S.poolChunks * sizeof(unsigned long long)));
        }
        HIPCHECK(hipMalloc(&sl.ctrl, sizeof(Ctrl)));
        int rcc = reset_ctrl(sl.ctrl, nullptr);
        if (rcc) return rcc;
        // radiance buffers and iteration masks: the frame's pixels, or only this shard's (KParams::contribLocal)
        const size_t cpx = k.contribLocal ? (size_t)(S.nLocal > 0 ? S.nLocal : 1) : (size_t)S.P;
        HIPCHECK(hipMalloc(&sl.contrib, (size_t)S.maxBatch * cpx * 3 * sizeof(float)));
        HIPCHECK(hipMemset(sl.contrib, 0, (size_t)S.maxBatch * cpx * 3 * sizeof(float)));
        HIPCHECK(hipMalloc(&sl.hitMask, (size_t)((S.maxBatch + 31) / 32) * cpx * sizeof(uint32_t)));
        HIPCHECK(hipMemset(sl.hitMask, 0, (size_t)((S.maxBatch + 31) / 32) * cpx * sizeof(uint32_t)));
        HIPCHECK(hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&sl.evCommitted, hipEventDisableTiming));
    }

    std::vector<GeomDev> hg(ngeoms ? ngeoms : 1);
    std::vector<MaterialDev> hm(nmats ? nmats : 1);
    // triangle meshes: one record array for the scene, a hierarchy per mesh (pt_mesh.h)
    std::vector<ptd::MeshUnit> meshRecs;
    int meshStackNeed = 0;
    const bool flatMeshes = getenv("PT_AMD_MESH_FLAT") && atoi(getenv("PT_AMD_MESH_FLAT"));   // tests only: no hierarchy
    std::vector<std::array<float, 6>> meshBox(ngeoms ? ngeoms : 1);
    std::vector<const float *> boxes(ngeoms ? ngeoms : 1, nullptr);
    for (int i = 0; i < ngeoms; ++i) {
        float *box = meshBox[i].data();
        const bool isMesh = geoms[i].type == PT_MESH;
        if (isMesh) boxes[i] = box;
        uint32_t root = ptd::kMeshEnd, stride = 0;
        if (isMesh) {
            const ptm::HostMesh *hm_ = mesh_of(i);
            for (int fm : hm_->mats)
                if (fm >= nmats) return fail(PT_ERR_INVALID, "pt_init: a face of mesh geom %d names material %d of %d", i, fm, nmats);
            const ptm::MeshLayout lay = ptm::appendMesh(hm_->tris.data(), (int)(hm_->tris.size() / 9), flatMeshes, meshRecs, box,
                                                        hm_->normals.empty() ? nullptr : hm_->normals.data(), hm_->mats.empty() ? nullptr : hm_->mats.data());
            root = lay.root;
            stride = lay.stride;
            meshStackNeed = std::max(meshStackNeed, lay.stackNeed);
            if (meshRecs.size() >= (1ull << 31)) return fail(PT_ERR_INVALID, "pt_init: too many triangles");
        }
        pack_geom(geoms[i], hg[i], k.pos, isMesh ? box : nullptr);
        hg[i].meshRoot = root;
        if (isMesh) hg[i].meshStride = stride;
        if (geoms[i].type == PT_CUBE) {
            if (k.nCubes >= 32767) return fail(PT_ERR_INVALID, "pt_init: more than 32767 cubes");
            hg[i].frameSlot = (short)k.nCubes++;
        }
    }
    // camera rays: pixel rectangles, their union and the per-row lists (thin lens: none -- rays start anywhere on the lens)
    CameraCull cc;
    const bool cullOff = S.dof || (getenv("PT_AMD_NO_CAMERA_CULL") && atoi(getenv("PT_AMD_NO_CAMERA_CULL")));   // (the variable: tests only)
    build_camera_cull(geoms, ngeoms, k, cullOff, boxes, hg, cc);
    for (int a = 0; a < 4; ++a) k.sceneRect[a] = cc.sceneRect[a];
    {   // The camera-ray tiles' index space covers only the column bands (of kBlock pixels) and the rows of this shard that meet the
        // scene rectangle: at 16:9 two of Cornell's five bands lie outside it, and a workgroup spent a tenth of the launch
        // stepping over their tiles one by one.  The pixels never visited are misses whatever their jitter: tallied at once.
        const int perRow = k.Wp / kBlock;
        int c0 = 0, c1 = perRow - 1, r0 = 0, r1 = rows - 1;
        if (cc.sceneRect[0] > cc.sceneRect[2] || cc.sceneRect[1] > cc.sceneRect[3]) {      // nothing can be hit
            c1 = -1; r1 = -1;
        } else {
            c0 = std::max(cc.sceneRect[0], 0) / kBlock;
            c1 = std::min(std::min(cc.sceneRect[2], Wd - 1) / kBlock, perRow - 1);
            // rows y = lr * shard_count + shard_rank inside [sceneRect[1], sceneRect[3]]
            const int y0 = std::max(cc.sceneRect[1], 0), y1 = std::min(cc.sceneRect[3], H - 1);
            r0 = y0 <= o.shard_rank ? 0 : (y0 - o.shard_rank + o.shard_count - 1) / o.shard_count;
            r1 = y1 < o.shard_rank ? -1 : std::min((y1 - o.shard_rank) / o.shard_count, rows - 1);
        }
        const int nCols = std::max(c1 - c0 + 1, 0), nRows = std::max(r1 - r0 + 1, 0);
        const long long visited = nCols > 0 && nRows > 0 ? (long long)nRows * (std::min(Wd, (c1 + 1) * kBlock) - c0 * kBlock) : 0;
        k.firstY0 = (nRows > 0 ? r0 : 0) * o.shard_count + o.shard_rank;      // (the first column: the band of sceneRect[0], k_bounce)
        k.firstSkipped = (int)((long long)S.nLocal - visited);
        k.Wp = std::max(nCols, 1) * kBlock;                       // (>= one band: the divisions below stay defined)
        k.nLocalPad = nCols > 0 ? nRows * k.Wp : 0;
        magic_divisor((uint32_t)k.Wp, k.magicWp, k.shiftWp);
        magic_divisor((uint32_t)std::max(k.nLocalPad, 1), k.magicN, k.shiftN);
        for (uint32_t d : {(uint32_t)k.Wp, (uint32_t)std::max(k.nLocalPad, 1)}) {
            uint32_t m, sh;
            magic_divisor(d, m, sh);
            for (uint64_t q = 0; q * d < (1ull << 30); q = q < 64 ? q + 1 : q * 2 + 1)
                for (uint64_t n : {q * d, q * d + d - 1, (uint64_t)((1ull << 30) - 1) - q})
                    if (n < (1ull << 30) && (uint32_t)((n * m) >> sh) != (uint32_t)(n / d))
                        return fail(PT_ERR_INVALID, "pt_init: magic division self-check failed for d=%u n=%llu", d, (unsigned long long)n);
        }
    }
    for (int i = 0; i < nmats; ++i) pack_material(mats[i], hm[i]);
    // Small primitives the queue is binned by (k_bounce): the spheres when there are at most kBinMax of them, then the
    // cubes whose bounding ball is small against the scene's (<= 0.3 of its radius), smallest first.  A choice that only
    // steers which tiles skip which tests; results never depend on it.
    {
        double cm[3] = {0, 0, 0}, sceneR = 0;
        for (int i = 0; i < ngeoms; ++i)
            for (int a = 0; a < 3; ++a) cm[a] += hg[i].centre[a] / std::max(ngeoms, 1);
        for (int i = 0; i < ngeoms; ++i) {
            const double dx = hg[i].centre[0] - cm[0], dy = hg[i].centre[1] - cm[1], dz = hg[i].centre[2] - cm[2];
            const double r = std::sqrt(dx * dx + dy * dy + dz * dz) + hg[i].boundR;
            if (std::isfinite(r)) sceneR = std::max(sceneR, r);
        }
        int nsph = 0;
        for (int i = 0; i < ngeoms; ++i) nsph += geoms[i].type == PT_SPHERE;
        std::vector<std::pair<double, int>> cand;
        for (int i = 0; i < ngeoms; ++i) {
            if (!std::isfinite(hg[i].cullR2)) continue;                       // never culled: cannot take part
            const double r = hg[i].boundR;
            if (geoms[i].type == PT_SPHERE) { if (nsph <= kBinMax) cand.emplace_back(-1.0, i); }   // spheres first
            else if (r <= 0.3 * sceneR) cand.emplace_back(r, i);
        }
        std::sort(cand.begin(), cand.end());
        k.nBinned = 0;
        for (size_t c = 0; c < cand.size() && k.nBinned < kBinMax; ++c) {
            k.binGeom[k.nBinned++] = cand[c].second;
            hg[cand[c].second].binned = 1;
            hg[cand[c].second].flags |= 2;
            hg[cand[c].second].cullFlags |= 2;
        }
    }
    // Mesh scenes bin by two candidate bits (pt_trace.h: kClsMax): the costliest binned mesh (most triangles) alone in group 1 when there is
    // another binned primitive beside it, everything else in group 0 -- a tile of the next bounce then walks that mesh only when its paths
    // can hit it, with all its lanes, instead of every cand tile walking every mesh with some.
    int binGroup[kBinMax] = {0, 0, 0, 0};
    if (!meshRecs.empty() && k.nBinned > 1) {
        int bestB = -1;
        size_t bestTris = 0;
        for (int b = 0; b < k.nBinned; ++b)
            if (geoms[k.binGeom[b]].type == PT_MESH) {
                const size_t nt = mesh_of(k.binGeom[b])->tris.size() / 9;
                if (nt > bestTris) { bestTris = nt; bestB = b; }
            }
        if (bestB >= 0) binGroup[bestB] = 1;
    }
    for (int b = 0; b < kBinMax; ++b) {                      // (KParams::binCull: the binned primitives' culling groups, inline)
        for (int q = 0; q < 8; ++q) k.binCull[b][q] = 0.0f;
        k.binCull[b][3] = -INFINITY;                          // beyond nBinned: certified for everybody
        if (b < k.nBinned) {
            const GeomDev &G = hg[k.binGeom[b]];
            k.binCull[b][0] = G.centre[0]; k.binCull[b][1] = G.centre[1]; k.binCull[b][2] = G.centre[2];
            k.binCull[b][3] = G.cullR2; k.binCull[b][4] = G.cullK;
            // word 5: the primitive's candidate bit in a survivor's class (k_bounce<..., MESH>): 1 = group 0, 2 = group 1
            const uint32_t bit = 1u << binGroup[b];
            memcpy(&k.binCull[b][5], &bit, sizeof bit);
        }
    }
    // Walls: the large cubes -- not binned, finite -- at most kWallMax of them, the largest first.  Survivors are classed by
    // the one wall they can still hit (ptd::wallCertainMiss against the inflated world boxes computed here), so a tile of
    // the next bounce tests one wall instead of all of them, and a survivor that can hit nothing at all ends at once.
    // A choice that only steers which tiles skip which tests; results never depend on it.
    std::vector<WallBox> hw(kWallMax);
    {
        std::vector<int> wallGeom;
        choose_walls(geoms, ngeoms, hg, k, hw, wallGeom);
        for (int w = 0; w < k.nWalls; ++w) {
            hg[wallGeom[w]].flags |= (w + 1) << 2;
            hg[wallGeom[w]].cullFlags |= (w + 1) << 2;
        }
        if (const char *e = getenv("PT_AMD_NO_WALLS")) { if (atoi(e)) { for (int i = 0; i < ngeoms; ++i) { hg[i].flags &= 3; hg[i].cullFlags &= 3; } k.nWalls = 0; k.wallOMax = 0.0f; k.nSlotWalls = 0; } }   // experiments only
        k.allClassified = k.nWalls > 0 ? 1 : 0;
        for (int i = 0; i < ngeoms; ++i)
            if (!hg[i].binned && (hg[i].flags & 28) == 0) k.allClassified = 0;
    }
    k.emittersBinned = k.nBinned > 0 ? 1 : 0;
    for (int i = 0; i < ngeoms; ++i) {
        bool emits = mats[geoms[i].materialid].emittance > 0.0f;
        if (geoms[i].type == PT_MESH)                   // (a mesh emits when any of its faces' own materials does)
            if (const ptm::HostMesh *hm_ = mesh_of(i))
                for (int fm : hm_->mats) emits = emits || (fm >= 0 && mats[fm].emittance > 0.0f);
        if (emits && !hg[i].binned) k.emittersBinned = 0;
    }
    HIPCHECK(hipMalloc(&S.dgeoms, hg.size() * sizeof(GeomDev)));
    HIPCHECK(hipMalloc(&S.dmats, hm.size() * sizeof(MaterialDev)));
    HIPCHECK(hipMemcpy(S.dgeoms, hg.data(), hg.size() * sizeof(GeomDev), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(S.dmats, hm.data(), hm.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));
    {   // the per-primitive hit records, ready-made: a workgroup's prologue copies them to LDS in one round trip instead of following
        // primitive -> material index -> material on the device (every workgroup of every launch did)
        std::vector<GeomHitDev> hh(hg.size());
        for (size_t i = 0; i < hg.size(); ++i) {
            GeomHitDev &h = hh[i];
            memset(&h, 0, sizeof h);
            const GeomDev &G = hg[i];
            const int mi = (int)i < ngeoms && G.material >= 0 && G.material < nmats ? G.material : 0;
            const MaterialDev &M = hm[(size_t)mi];
            h.type = G.type;
            h.emittance = M.emittance; h.hasReflective = M.hasReflective; h.hasRefractive = M.hasRefractive;
            for (int a = 0; a < 3; ++a) h.color[a] = M.color[a];
            h.material = G.material;
            memcpy(h.nm, G.invT, sizeof h.nm);
            memcpy(h.cubeFrame, G.cubeFrame, sizeof h.cubeFrame);
        }
        HIPCHECK(hipMalloc(&S.dGeomHit, hh.size() * sizeof(GeomHitDev)));
        HIPCHECK(hipMemcpy(S.dGeomHit, hh.data(), hh.size() * sizeof(GeomHitDev), hipMemcpyHostToDevice));
    }
    HIPCHECK(hipMalloc(&S.dwalls, hw.size() * sizeof(WallBox)));
    HIPCHECK(hipMemcpy(S.dwalls, hw.data(), hw.size() * sizeof(WallBox), hipMemcpyHostToDevice));
    S.mesh = !meshRecs.empty();
    if (S.mesh) {
        HIPCHECK(hipMalloc(&S.dMeshRecs, meshRecs.size() * sizeof(ptd::MeshUnit)));
        HIPCHECK(hipMemcpy(S.dMeshRecs, meshRecs.data(), meshRecs.size() * sizeof(ptd::MeshUnit), hipMemcpyHostToDevice));
    }

    int nspheres = 0;
    for (int i = 0; i < ngeoms; ++i) nspheres += geoms[i].type == PT_SPHERE;
    S.many = nspheres > kBinMax;
    if (S.many && ngeoms > 65535) return fail(PT_ERR_INVALID, "pt_init: more than 65535 primitives");
    if (S.many) {        // the later bounces take the spheres from a packed copy of their culling data (ptk::SphereCull)
        std::vector<SphereCull> sc;
        for (int i = 0; i < ngeoms; ++i)
            if (geoms[i].type == PT_SPHERE) {
                SphereCull e;
                memset(&e, 0, sizeof e);
                for (int a = 0; a < 3; ++a) e.centre[a] = hg[i].centre[a];
                e.cullR2 = hg[i].cullR2;
                e.cullK = hg[i].cullK + kUnitDirSlack;      // (the sweep's direction is normalised approximately: sphereHalfLineExcess)
                e.geom = i;
                sc.push_back(e);
            }
        // the K |oc|^2 term of the certificate folded into the sweep's direction (ptd::sphereHalfLineExcessScaled): one factor for the
        // scene, from its largest K, and every threshold multiplied by its square -- both rounded upwards (the conservative side)
        double kmax = 0.0;
        for (const SphereCull &e : sc) kmax = std::max(kmax, (double)e.cullK);
        const float sdir = std::nextafter((float)std::sqrt(1.0 / (1.0 - kmax)), INFINITY);
        k.sphDirScale = sdir;
        for (SphereCull &e : sc)
            if (std::isfinite(e.cullR2)) e.cullR2 = std::nextafter((float)((double)e.cullR2 * (double)sdir * (double)sdir), INFINITY);
        // two spatial CLUSTERS (scenes without meshes, whose second candidate bit is free): build_sphere_clusters
        k.sphN0 = 0; k.sphOMax = 0.0f;
        for (int g = 0; g < 2; ++g) for (int q = 0; q < 8; ++q) k.sphBox[g][q] = 0.0f;
        if (meshRecs.empty()) {
            std::vector<int> binned(k.binGeom, k.binGeom + k.nBinned);
            build_sphere_clusters(geoms, ngeoms, hg, binned, sc, k.sphN0, k.sphOMax, k.sphBox);
        }
        if (k.sphOMax <= 0.0f) { k.sphN0 = 0; k.sphOMax = -1.0f; }      // no clusters: no certificate is issued, every tile sweeps the whole table
        else if (k.nWalls > 0) {
            // with the spheres behind candidate bits too, a survivor whose certificates leave no wall, no binned primitive and no cluster has
            // nothing left to hit (KParams::allClassified) -- when there is no primitive of another kind
            k.allClassified = 1;
            for (int i = 0; i < ngeoms; ++i)
                if (!hg[i].binned && (hg[i].flags & 28) == 0 && geoms[i].type != PT_SPHERE) k.allClassified = 0;
        }
        if (sc.size() % 2) sc.push_back(sc.back());      // (two per scalar load; testing a sphere twice changes nothing)
        k.nSphCull = (int)sc.size();
        HIPCHECK(hipMalloc(&S.dSphCull, sc.size() * sizeof(SphereCull)));
        HIPCHECK(hipMemcpy(S.dSphCull, sc.data(), sc.size() * sizeof(SphereCull), hipMemcpyHostToDevice));
        // The tables a sphere-heavy workgroup stages in LDS -- the compact hit records, the cubes' face frames, the spheres' matrix rows,
        // the sweep's entry -> primitive map -- as ONE image in the kernel's own layout (k_bounce: S_GEOMHIT_SMALL .. behind S_SPH), so that
        // the prologue is a straight copy of 16-byte words: gathering them field by field from the primitives took ~30 dependent
        // round trips, 28 us at the head of every launch of C5 (profiles/timeline_phases.py: 62 k cycles against Cornell's 10 k).
        {
            const size_t hitB = manyHitBytes(ngeoms), frameB = (size_t)k.nCubes * 54 * sizeof(float) + manyFramePad(k.nCubes);
            const size_t rowB = (size_t)ngeoms * kSphRowFloats * sizeof(float), mapB = ((size_t)k.nSphCull + 7) / 8 * 8 * sizeof(uint16_t);
            std::vector<unsigned char> blob(hitB + frameB + rowB + mapB, 0);
            GeomHitSmall *hs = reinterpret_cast<GeomHitSmall *>(blob.data());
            float *fr = reinterpret_cast<float *>(blob.data() + hitB);
            float *rows = reinterpret_cast<float *>(blob.data() + hitB + frameB);
            uint16_t *map = reinterpret_cast<uint16_t *>(blob.data() + hitB + frameB + rowB);
            for (int g = 0; g < ngeoms; ++g) {
                const GeomDev &G = hg[g];
                memcpy(hs[g].nm, G.invT, sizeof hs[g].nm);
                hs[g].material = G.material; hs[g].type = G.type; hs[g].frame = G.type == 1 ? (int)G.frameSlot : 0;
                if (G.type == 1) memcpy(fr + (size_t)G.frameSlot * 54, G.cubeFrame, 54 * sizeof(float));
                float *r = rows + (size_t)g * kSphRowFloats;
                memcpy(r, G.inv, 12 * sizeof(float)); memcpy(r + 12, G.xf, 12 * sizeof(float)); memcpy(r + 24, G.invZ, 3 * sizeof(float));
            }
            for (int i = 0; i < k.nSphCull; ++i) map[i] = (uint16_t)sc[i].geom;
            if (S.dGeomHit) (void)hipFree(S.dGeomHit);
            S.dGeomHit = nullptr;
            HIPCHECK(hipMalloc(&S.dGeomHit, blob.size()));
            HIPCHECK(hipMemcpy(S.dGeomHit, blob.data(), blob.size(), hipMemcpyHostToDevice));
        }
    }
    {   // Later bounces: which primitives a tile of queue class c looks at.  Class bit 3 = its paths may hit a binned primitive;
        // bits 0-2 in a scene with walls = the one wall they can still hit (6: any, 7: none), else the direction octant.
        std::vector<int> idx;
        const int ncls = (S.mesh || S.many) ? kClsMax : kCls;                 // (mesh and sphere-heavy scenes: two candidate bits, 32 classes)
        for (int c = 0; c < kClsMax; ++c) {
            k.classOff[c] = (int)idx.size();
            if (c >= ncls) continue;
            const int small = c >> 3;                                          // candidate bits: which groups of binned primitives
            const int wall = k.nWalls > 0 ? (c & 7) : 6;
            for (int i = 0; i < ngeoms; ++i) {
                if (S.many && geoms[i].type == PT_SPHERE) continue;            // swept from their packed culling data
                if (hg[i].binned) {
                    int grp = 0;
                    for (int b = 0; b < k.nBinned; ++b)
                        if (k.binGeom[b] == i) grp = binGroup[b];
                    if (!((small >> grp) & 1)) continue;
                }
                const int w = (hg[i].flags >> 2) & 7;                          // 1 + index among the walls, 0: not one
                if (w != 0 && wall != 6 && w != wall + 1) continue;
                idx.push_back(i);
            }
        }
        k.classOff[kClsMax] = (int)idx.size();
        // camera rays: the per-row primitive lists (build_camera_cull)
        if (!cc.rowOff.empty()) {
            HIPCHECK(hipMalloc(&S.dRowOff, cc.rowOff.size() * sizeof(int)));
            HIPCHECK(hipMemcpy(S.dRowOff, cc.rowOff.data(), cc.rowOff.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHECK(hipMalloc(&S.dRowIdx, cc.rowIdx.size() * sizeof(int)));
            HIPCHECK(hipMemcpy(S.dRowIdx, cc.rowIdx.data(), cc.rowIdx.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        if (idx.empty()) idx.push_back(0);
        HIPCHECK(hipMalloc(&S.dClassIdx, idx.size() * sizeof(int)));
        HIPCHECK(hipMemcpy(S.dClassIdx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    const size_t ldsFixed = sizeof(MaterialDev) * nmats + (size_t)miscWords((S.mesh || S.many) ? kClsMax : kCls) * sizeof(uint32_t) +
                            (S.many ? manyHitBytes(ngeoms) + (size_t)k.nCubes * 54 * sizeof(float) + manyFramePad(k.nCubes) +
                                          (size_t)ngeoms * kSphRowFloats * sizeof(float)
                                    : sizeof(GeomHitDev) * ngeoms);
    // (sphere-heavy scenes: the camera-ray launch keeps the lanes' candidate lists behind the tables, the later ones only the sweep's
    // entry -> primitive map -- 4 KB less, which is what their seventh workgroup per CU needs)
    // (... and, behind the map, the pooled pass's pair descriptors: [kWaves][64] words)
    const size_t sphMapBytes = ((size_t)k.nSphCull + 7) / 8 * 8 * sizeof(uint16_t), pairBytes = (size_t)kBlock * sizeof(uint32_t);
    k.pairOff = (int)(ldsFixed + sphMapBytes);
    S.ldsBytes = ldsFixed + (S.many ? std::max((size_t)kListMax * kBlock * sizeof(uint16_t), sphMapBytes + pairBytes) : 0);
    S.ldsBytesNext = (S.many && !S.mesh) ? ldsFixed + sphMapBytes + pairBytes : 0;
    if (S.mesh) {        // the lanes' stacks of far children (ptd::meshIntersectionTest): kBlock words per level, behind everything else
        S.ldsBytes = (S.ldsBytes + 15) / 16 * 16;
        k.meshStackOff = (int)S.ldsBytes;
        S.ldsBytes += (size_t)std::max(meshStackNeed, 1) * kBlock * sizeof(uint32_t);
    }
    if (S.ldsBytesNext == 0) S.ldsBytesNext = S.ldsBytes;
    if (S.ldsBytes > 160 * 1024) return fail(PT_ERR_INVALID, "pt_init: scene does not fit the 160 KiB LDS (%zu B)", S.ldsBytes);
    if (nmats >= 4096) return fail(PT_ERR_INVALID, "pt_init: more than 4095 materials");      // (TileArgs::hot holds nmats in 12 bits)
    const void *kFirst = bounce_kernel(true, S.dof);
    const void *kNext = bounce_kernel(false, false);
    if (S.ldsBytes > 64 * 1024) {
        HIPCHECK(hipFuncSetAttribute(kFirst, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.ldsBytes));
        HIPCHECK(hipFuncSetAttribute(kNext, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.ldsBytes));
    }
    for (int first = 0; first < 2; ++first) {
        int &grid = first ? S.gridFirst : S.grid;
        int rc = persistent_grid(first ? kFirst : kNext, first ? S.ldsBytes : S.ldsBytesNext, &grid);
        if (rc) return rc;
        if (grid > S.numTilesMax) grid = S.numTilesMax;
        grid = (grid / kSub) * kSub;      // T % kSub == blockIdx % kSub for every tile T of a workgroup (kSub: a multiple of the mesh scenes' 4 too)
        if (grid < kSub) grid = kSub;
    }
    // Camera-ray bounce: tile T covers pixels 256 T ... of the row-major frame and a workgroup owns the tiles b, b + grid,
    // ...  When the grid shares a factor with the tiles per row (1280 workgroups, 5 tiles per 1280-pixel row) a workgroup
    // stays in few column bands of the frame, and the bands outside the scene rectangle finish 15x earlier than the
    // others.  Either the grid can be made coprime to the tiles per row (both must stay multiples of kSub, so only for an
    // odd tile count per row), or the kernel rotates the k-th tile of a workgroup k bands to the right inside its row,
    // which needs a grid that is a multiple of the tiles per row (k_bounce<true, .>).
    S.prm.tilesPerRow = 0;
    if (S.prm.Wp / kBlock > 1) {
        const int perRow = S.prm.Wp / kBlock;
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        if (gcd(perRow, kSub) == 1) {
            for (int tries = 0; tries < 64 && S.gridFirst > kSub && gcd(S.gridFirst, perRow) != 1; ++tries) S.gridFirst -= kSub;
        } else {
            const int unit = perRow / gcd(perRow, kSub) * kSub;        // lcm(perRow, kSub)
            if (S.gridFirst >= 4 * unit) {
                S.gridFirst = S.gridFirst / unit * unit;
                S.prm.tilesPerRow = perRow;
            }
        }
    }
    if (getenv("PT_AMD_VERBOSE") && atoi(getenv("PT_AMD_VERBOSE")))       // experiments: what pt_init decided
        fprintf(stderr, "pt_init: lds %zu / %zu B, grid %d / %d, mesh %d many %d plain %d, binned %d walls %d allClassified %d, sphCull %d (cluster 0: %d) omax %g\n",
                S.ldsBytes, S.ldsBytesNext, S.gridFirst, S.grid, (int)S.mesh, (int)S.many, (int)S.plain, k.nBinned, k.nWalls, k.allClassified, k.nSphCull,
                k.sphN0, (double)k.sphOMax);
    HIPCHECK(hipDeviceSynchronize());
    S.init = true;
    g_err.clear();
    return PT_OK;
}

int pt_iterate_batch(int frame, int first_iter, int count, void *rgba8_dev) {
    (void)frame;  // always 0 in the reference (src/main.cpp:102)
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_iterate before pt_init");
    if (count < 1 || count > S.maxBatch) return fail(PT_ERR_INVALID, "pt_iterate_batch: count must be 1..max_batch (%d)", S.maxBatch);
    if (first_iter < 1 || first_iter + count - 1 >= kIterEnd)
        return fail(PT_ERR_INVALID, "pt_iterate: iter must be 1..4194303 (seed bits, pathtrace.cu:43)");
    // every argument is checked BEFORE anything is enqueued: a rejected call leaves the image and the counters untouched
    if (rgba8_dev && (S.flags & PT_FLAG_ACCUM_SHARD_ROWS))
        return fail(PT_ERR_INVALID, "pt_iterate: no PBO conversion from a row-sharded accumulator");
    int rc;
    if ((S.flags & PT_FLAG_TRACE_AHEAD) && count == 1) {
        // the reference's protocol, one call per iteration: the iteration comes out of a batch that was traced ahead
        if (!S.ahead.empty() && S.ahead.front().first + S.ahead.front().next != first_iter) {
            rc = discard_ahead();                    // not the iteration the parked batches continue with
            if (rc) return rc;
        }
        if (S.ahead.empty()) {
            rc = trace_ahead(first_iter);
            if (rc) return rc;
        }
        State::Parked &p = S.ahead.front();
        Slot &sl = S.slot[p.slot];
        if (!p.waited) {      // (once per batch: the commits that follow on the caller's stream are ordered behind this one)
            HIPCHECK(hipStreamWaitEvent(S.stream, sl.evDone, 0));
            p.waited = true;
        }
        rc = commit_range(sl, p.count, p.next, p.next + 1, false);
        if (rc) return rc;
        int after = p.first + p.count;               // first iteration behind the parked batches
        if (++p.next == p.count) {
            HIPCHECK(hipEventRecord(sl.evCommitted, S.stream));
            S.ahead.pop_front();
        }
        // every free slot traces on: the GPU stays ahead of the caller by at least a batch
        if (!S.ahead.empty()) after = S.ahead.back().first + S.ahead.back().count;
        while ((int)S.ahead.size() < S.nslots && after < kIterEnd) {
            rc = trace_ahead(after);
            if (rc) return rc;
            after = S.ahead.back().first + S.ahead.back().count;
        }
    } else {
        if (!S.ahead.empty()) {
            rc = discard_ahead();
            if (rc) return rc;
        }
        Slot &sl = S.slot[S.seq % S.nslots];
        rc = trace_batch(sl, first_iter, count);
        if (rc) return rc;
        // commit on the caller's stream: commits are therefore ordered like the pt_iterate calls
        HIPCHECK(hipStreamWaitEvent(S.stream, sl.evDone, 0));
        rc = commit_range(sl, count, 0, count, false);
        if (rc) return rc;
        HIPCHECK(hipEventRecord(sl.evCommitted, S.stream));
        S.seq += 1;
    }
    if (rgba8_dev) {
        hipLaunchKernelGGL(k_to_rgba8, dim3((S.P + kBlock - 1) / kBlock), dim3(kBlock), 0, S.stream, S.image, S.P,
                           first_iter + count - 1, reinterpret_cast<uchar4 *>(rgba8_dev));
        HIPCHECK(hipGetLastError());
    }
    S.iterations += count;
    return PT_OK;
}

int pt_iterate(int frame, int iter, void *rgba8_dev) { return pt_iterate_batch(frame, iter, 1, rgba8_dev); }

int pt_sync(void) {
    if (!S.init) {
        if (!scan_in_use()) return fail(PT_ERR_NOT_INIT, "pt_sync before pt_init");
        HIPCHECK(hipDeviceSynchronize());          // the scan library alone
        return PT_OK;
    }
    return check_device_fault();
}

int pt_readback(float *rgb_sum_host) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_readback before pt_init");
    if (!rgb_sum_host) return fail(PT_ERR_INVALID, "pt_readback: null");
    // every commit so far is already ordered before this copy on the caller's stream
    if (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) {   // scatter this shard's rows into a zeroed full frame
        std::vector<float> rows((size_t)S.nLocal * 3);
        if (S.nLocal) HIPCHECK(hipMemcpyAsync(rows.data(), S.image, rows.size() * sizeof(float), hipMemcpyDeviceToHost, S.stream));
        HIPCHECK(hipStreamSynchronize(S.stream));
        memset(rgb_sum_host, 0, (size_t)S.P * 3 * sizeof(float));
        const size_t rowFloats = (size_t)S.prm.W * 3;
        for (int lr = 0; lr * S.prm.W < S.nLocal; ++lr)
            memcpy(rgb_sum_host + (size_t)(lr * S.prm.shardCount + S.prm.shardRank) * rowFloats, rows.data() + lr * rowFloats,
                   rowFloats * sizeof(float));
        return readback_fault();
    }
    // a plain copy: at PCIe rate into a buffer the caller has page-locked with pt_pin_host, through the runtime's
    // pageable staging path otherwise.  The library never registers memory it does not own on its own initiative.
    const size_t bytes = (size_t)S.P * 3 * sizeof(float);
    HIPCHECK(hipMemcpyAsync(rgb_sum_host, S.image, bytes, hipMemcpyDeviceToHost, S.stream));
    HIPCHECK(hipStreamSynchronize(S.stream));
    return readback_fault();
}

int pt_pin_host(void *host, size_t bytes) {
    if (!host || bytes == 0) return fail(PT_ERR_INVALID, "pt_pin_host: bad argument");
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (S.pinnedHost) {
        if (S.pinnedHost == host && S.pinnedBytes == bytes) return PT_OK;
        (void)hipHostUnregister(S.pinnedHost);
        S.pinnedHost = nullptr;
    }
    HIPCHECK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    S.pinnedHost = host;
    S.pinnedBytes = bytes;
    return PT_OK;
}

int pt_unpin_host(void) {
    if (S.pinnedHost) {
        (void)hipStreamSynchronize(S.stream);
        HIPCHECK(hipHostUnregister(S.pinnedHost));
        S.pinnedHost = nullptr;
        S.pinnedBytes = 0;
    }
    return PT_OK;
}

int pt_readback_rgba8(int iter, uint8_t *rgba_host) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_readback_rgba8 before pt_init");
    if (!rgba_host || iter < 1) return fail(PT_ERR_INVALID, "pt_readback_rgba8: bad argument");
    if (S.flags & PT_FLAG_ACCUM_SHARD_ROWS) return fail(PT_ERR_INVALID, "pt_readback_rgba8: accumulator is row-sharded");
    DevBuf<uchar4> tmp;
    int rc = tmp.alloc(S.P);
    if (rc) return rc;
    hipLaunchKernelGGL(k_to_rgba8, dim3((S.P + kBlock - 1) / kBlock), dim3(kBlock), 0, S.stream, S.image, S.P, iter, tmp.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(rgba_host, tmp.p, (size_t)S.P * 4, hipMemcpyDeviceToHost, S.stream));
    HIPCHECK(hipStreamSynchronize(S.stream));
    return readback_fault();
}

int pt_counters(PtCounters *out) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_counters before pt_init");
    if (!out) return fail(PT_ERR_INVALID, "pt_counters: null");
    int rc = sync_all();
    if (rc) return rc;
    rc = resolve_events(S.evBounce, S.msBounce, S.nBounce);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    static Ctrl h;   // 0.5 MB: keep it off the stack
    uint32_t faultBits = 0;
    for (int i = 0; i < S.nslots; ++i) {
        HIPCHECK(hipMemcpy(&h, S.slot[i].ctrl, sizeof h, hipMemcpyDeviceToHost));
        for (int d = 0; d < kMaxDepthSlots; ++d) {
            int64_t early = 0;
            for (int sg = 0; sg < kTallyShards; ++sg) early += (int64_t)h.early[d][sg][0];
            out->live[d] += (int64_t)h.sum_live[d] + early;   // they did enter bounce d
            out->ended_early[d] += early;                      // ... without being moved through memory
        }
        for (int sg = 0; sg < kTallyShards; ++sg) {
            out->light_hits += (int64_t)h.light_hits[sg][0];
            out->misses += (int64_t)h.misses[sg][0];
        }
        faultBits |= h.error;
    }
    out->iterations = S.iterations;
    out->bounce_launches = S.nBounce;
    out->bounce_kernel_ms = S.msBounce;
    out->raygen_kernel_ms = 0.0;   // camera rays are generated inside the first bounce launch
    out->raygen_launches = 0;
    if (faultBits) return fail(PT_ERR_DEVICE, "device fault 0x%x:%s%s (results of this render are void; re-init)", faultBits,
                               (faultBits & kFaultPoolExhausted) ? " path pool exhausted" : "", (faultBits & kFaultReserveTimeout) ? " chunk reservation timed out" : "");
    return PT_OK;
}

int pt_counters_reset(void) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_counters_reset before pt_init");
    int rc = sync_all();
    if (rc) return rc;
    rc = resolve_events(S.evBounce, S.msBounce, S.nBounce);
    if (rc) return rc;
    S.msBounce = 0;
    S.nBounce = 0;
    S.iterations = 0;
    for (int i = 0; i < S.nslots; ++i) {
        // the sticky fault word survives a counter reset
        uint32_t err = 0;
        HIPCHECK(hipMemcpy(&err, &S.slot[i].ctrl->error, sizeof err, hipMemcpyDeviceToHost));
        rc = reset_ctrl(S.slot[i].ctrl, nullptr);
        if (rc) return rc;
        S.slot[i].parity = 0;
        if (err) HIPCHECK(hipMemcpy(&S.slot[i].ctrl->error, &err, sizeof err, hipMemcpyHostToDevice));
    }
    return PT_OK;
}

int pt_debug_trace_paths(int iter, int bounces, float *origin3, float *dir3, float *color3, int32_t *pixelIndex,
                         int32_t *count) {
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_debug_trace_paths before pt_init");
    if (bounces < 0 || bounces > PT_MAX_DEPTH || !count) return fail(PT_ERR_INVALID, "pt_debug_trace_paths: bad argument");
    if (bounces > S.prm.traceDepth) return fail(PT_ERR_INVALID, "pt_debug_trace_paths: bounces > traceDepth");
    int rc = sync_all();
    if (rc) return rc;
    Slot &sl = S.slot[0];
    if (bounces == 0) {   // camera rays only (they never exist in HBM: generation is fused into bounce 1)
        const int nl = S.nLocal;
        *count = nl;
        if (nl == 0) return PT_OK;
        DevBuf<float> o, d;
        DevBuf<int> px;
        if ((rc = o.alloc((size_t)nl * 3)) || (rc = d.alloc((size_t)nl * 3)) || (rc = px.alloc(nl))) return rc;
        hipLaunchKernelGGL(k_debug_camera_rays, dim3((nl + kBlock - 1) / kBlock), dim3(kBlock), 0, sl.stream, S.prm, iter,
                           o.p, d.p, px.p);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(sl.stream));
        if (origin3) HIPCHECK(hipMemcpy(origin3, o.p, (size_t)nl * 12, hipMemcpyDeviceToHost));
        if (dir3) HIPCHECK(hipMemcpy(dir3, d.p, (size_t)nl * 12, hipMemcpyDeviceToHost));
        if (pixelIndex) HIPCHECK(hipMemcpy(pixelIndex, px.p, (size_t)nl * 4, hipMemcpyDeviceToHost));
        if (color3)
            for (size_t i = 0; i < (size_t)nl * 3; ++i) color3[i] = 1.0f;
        return PT_OK;
    }
    // private run on slot 0: re-arm its cursors, trace, read the queue entering bounce `bounces + 1`, re-arm again
    uint32_t err = 0;
    HIPCHECK(hipMemcpy(&err, &sl.ctrl->error, sizeof err, hipMemcpyDeviceToHost));
    rc = reset_ctrl(sl.ctrl, sl.stream);
    if (rc) return rc;
    for (int d = 1; d <= bounces; ++d) {
        rc = launch_bounce(sl, iter, 1, d, false, nullptr);  // no radiance, survivors always written
        if (rc) return rc;
    }
    HIPCHECK(hipStreamSynchronize(sl.stream));
    // gather the kSeg segments (chunk lists) of that queue, then sort by pixel index
    static Ctrl h;
    HIPCHECK(hipMemcpy(&h, sl.ctrl, sizeof h, hipMemcpyDeviceToHost));
    rc = reset_ctrl(sl.ctrl, sl.stream);
    if (rc) return rc;
    err |= h.error;
    if (err) HIPCHECK(hipMemcpy(&sl.ctrl->error, &err, sizeof err, hipMemcpyHostToDevice));
    if (h.error) return fail(PT_ERR_DEVICE, "pt_debug_trace_paths: device fault 0x%x", h.error);
    const PathPool pb = pool(sl, bounces & 1);
    const uint32_t gen = sl.gen[bounces & 1];
    std::vector<unsigned long long> lists((size_t)kSeg * S.poolChunks);
    HIPCHECK(hipMemcpy(lists.data(), pb.list, lists.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint32_t segn[kSeg];
    size_t n = 0;
    for (int sg = 0; sg < kSeg; ++sg) {
        segn[sg] = h.pos[sl.parity][bounces + 1][sg][0];
        n += segn[sg];
    }
    *count = (int32_t)n;
    if (n == 0) return PT_OK;
    const uint32_t chunkPaths = 1u << S.prm.chunkShift;
    // the pool's three arrays (A: 16 B, B: 16 B, C: 12 B per path), chunk by chunk, into the eleven columns
    std::vector<float> cols[kNumArrays];
    for (int k = 0; k < kNumArrays; ++k) cols[k].resize(n);
    {
        std::vector<float> bufA((size_t)chunkPaths * 4), bufB((size_t)chunkPaths * 4), bufC((size_t)chunkPaths * 3);
        size_t off = 0;
        for (int sg = 0; sg < kSeg; ++sg)
            for (uint32_t done = 0, j = 0; done < segn[sg]; done += chunkPaths, ++j) {
                const uint32_t m = std::min<uint32_t>(chunkPaths, segn[sg] - done);
                const unsigned long long e = lists[(size_t)sg * S.poolChunks + j];
                const uint32_t c = j == 0 ? 1u + (uint32_t)sg : (uint32_t)e;
                if (c == 0 || c >= (uint32_t)S.poolChunks || (j != 0 && (uint32_t)(e >> 32) != gen))
                    return fail(PT_ERR_DEVICE, "pt_debug_trace_paths: corrupt chunk list");
                const size_t first = (size_t)c << S.prm.chunkShift;
                HIPCHECK(hipMemcpy(bufA.data(), pb.arrA(first), (size_t)m * 16, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(bufB.data(), pb.arrB(first), (size_t)m * 16, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(bufC.data(), pb.arrC(first), (size_t)m * 12, hipMemcpyDeviceToHost));
                for (uint32_t i = 0; i < m; ++i) {
                    const float *a = &bufA[4 * (size_t)i], *b = &bufB[4 * (size_t)i], *cc = &bufC[3 * (size_t)i];
                    const float v[kNumArrays] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], cc[0], cc[1], cc[2]};
                    for (int k = 0; k < kNumArrays; ++k) cols[k][off + i] = v[k];
                }
                off += m;
            }
    }
    // (array C's third word: pixelIndex | batch index << pixBits; this private run traces ONE iteration: the batch index is 0)
    const int *pixcol = reinterpret_cast<const int *>(cols[10].data());
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return pixcol[x] < pixcol[y]; });
    float *dst[3] = {origin3, dir3, color3};
    for (size_t i = 0; i < n; ++i) {
        const size_t src = order[i];
        for (int grp = 0; grp < 3; ++grp)
            if (dst[grp])
                for (int c = 0; c < 3; ++c) dst[grp][3 * i + c] = cols[grp * 3 + c][src];
        if (pixelIndex) pixelIndex[i] = pixcol[src];
    }
    return PT_OK;
}

// ---- stream compaction library -------------------------------------------------------------------------
int pt_scan_exclusive_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, void *stream) {
    if (n < 0 || (n > 0 && (!in_dev || !out_dev))) return fail(PT_ERR_INVALID, "pt_scan_exclusive_i32: bad argument");
    if (n == 0) return PT_OK;
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (n > (1ll << 42)) return fail(PT_ERR_INVALID, "pt_scan_exclusive_i32: n too large");
    hipStream_t st = (hipStream_t)stream;
    register_exit_handler();
    std::lock_guard<std::mutex> lock(g_scanMutex);   // (look-up AND launches: see scan_ws)
    ScanWs *wp = nullptr;
    int rc = scan_ws(st, &wp);
    if (rc) return rc;
    long long per;
    int chunks;
    scan_chunks(n, &per, &chunks);
    hipLaunchKernelGGL((k_scan_reduce<false>), dim3(chunks), dim3(kBlock), 0, st, in_dev, (long long)n, per, wp->partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(kBlock), 0, st, wp->partial, chunks, (long long *)nullptr);
    hipLaunchKernelGGL(k_scan_apply, dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->partial);
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

int pt_compact_nonzero_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, int64_t *count_dev, void *stream) {
    if (n < 0 || !count_dev || (n > 0 && (!in_dev || !out_dev))) return fail(PT_ERR_INVALID, "pt_compact_nonzero_i32: bad argument");
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (n > 0xffffffffll) return fail(PT_ERR_INVALID, "pt_compact_nonzero_i32: n too large (positions are 32-bit)");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        HIPCHECK(hipMemsetAsync(count_dev, 0, sizeof(int64_t), st));
        return PT_OK;
    }
    register_exit_handler();
    std::lock_guard<std::mutex> lock(g_scanMutex);   // (look-up AND launches: see scan_ws)
    ScanWs *wp = nullptr;
    int rc = scan_ws(st, &wp);
    if (rc) return rc;
    long long per;
    int chunks;
    scan_chunks(n, &per, &chunks);
    hipLaunchKernelGGL((k_scan_reduce<true>), dim3(chunks), dim3(kBlock), 0, st, in_dev, (long long)n, per, wp->partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(kBlock), 0, st, wp->partial, chunks, reinterpret_cast<long long *>(count_dev));
    hipLaunchKernelGGL(k_compact_apply, dim3(chunks), dim3(kBlock), 0, st, in_dev, out_dev, (long long)n, per, wp->partial);
    HIPCHECK(hipGetLastError());
    return PT_OK;
}

#ifdef PT_TEST_API
// ---- primitive tests over host arrays ------------------------------------------------------------------
int pt_test_force_fault(int which) {
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (which != 0 && which != 2) return fail(PT_ERR_INVALID, "pt_test_force_fault: which must be 0 (clear) or 2 (the renderer's fault word)");
    if (!S.init) return fail(PT_ERR_NOT_INIT, "pt_test_force_fault before pt_init");
    HIPCHECK(hipDeviceSynchronize());
    const uint32_t word = which == 2 ? 1u : 0u;
    for (int i = 0; i < (which == 2 ? 1 : S.nslots); ++i) HIPCHECK(hipMemcpy(&S.slot[i].ctrl->error, &word, sizeof word, hipMemcpyHostToDevice));
    if (S.hostFault) *S.hostFault = word;
    return PT_OK;
}

#define NEED_GPU() do { if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device"); } while (0)
#define UP(buf, host, count) do { int rc_ = buf.alloc(count); if (rc_) return rc_; \
    HIPCHECK(hipMemcpy(buf.p, host, (size_t)(count) * sizeof(*buf.p), hipMemcpyHostToDevice)); } while (0)
#define DOWN(host, buf, count) HIPCHECK(hipMemcpy(host, buf.p, (size_t)(count) * sizeof(*buf.p), hipMemcpyDeviceToHost))
#define GRID(n) dim3(((n) + 255) / 256), dim3(256), 0, 0

int pt_test_utilhash(const uint32_t *in, uint32_t *out, int n) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<uint32_t> a, b;
    UP(a, in, n);
    int rc = b.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_utilhash, GRID(n), a.p, b.p, n);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(out, b, n);
    return PT_OK;
}

int pt_test_rng(const uint32_t *seeds, int nseeds, int ndraws, float *u01_out) {
    NEED_GPU();
    if (nseeds <= 0 || ndraws <= 0) return PT_OK;
    DevBuf<uint32_t> a;
    DevBuf<float> b;
    UP(a, seeds, nseeds);
    int rc = b.alloc((size_t)nseeds * ndraws); if (rc) return rc;
    hipLaunchKernelGGL(k_test_rng, GRID(nseeds), a.p, nseeds, ndraws, b.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(u01_out, b, (size_t)nseeds * ndraws);
    return PT_OK;
}

int pt_test_intersect(const PtGeom *geoms, int ngeoms, const int32_t *geom_index, const float *rays, int n, float *t,
                      float *p3, float *n3, int32_t *outside) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    for (int i = 0; i < n; ++i)
        if (geom_index[i] < 0 || geom_index[i] >= ngeoms) return fail(PT_ERR_INVALID, "pt_test_intersect: geom index out of range");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i]);
    DevBuf<GeomDev> dg;
    DevBuf<int> di, dout;
    DevBuf<float> dr, dt, dp, dn;
    UP(dg, hg.data(), ngeoms);
    UP(di, geom_index, n);
    UP(dr, rays, (size_t)n * 6);
    UP(dp, p3, (size_t)n * 3);
    UP(dn, n3, (size_t)n * 3);
    UP(dout, outside, n);
    int rc = dt.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_intersect, GRID(n), dg.p, di.p, dr.p, n, dt.p, dp.p, dn.p, dout.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(t, dt, n);
    DOWN(p3, dp, (size_t)n * 3);
    DOWN(n3, dn, (size_t)n * 3);
    DOWN(outside, dout, n);
    return PT_OK;
}

int pt_test_mesh_intersect(const PtGeom *geom, const float *tris, int ntris, int flat, const float *rays, int n, float *t,
                           float *p3, float *n3, int32_t *outside, int32_t *culled) {
    NEED_GPU();
    if (!geom || !tris || ntris < 1 || geom->type != PT_MESH) return fail(PT_ERR_INVALID, "pt_test_mesh_intersect: bad argument");
    if (n <= 0) return PT_OK;
    std::vector<ptd::MeshUnit> recs;
    float box[6];
    const ptm::MeshLayout lay = ptm::appendMesh(tris, ntris, flat != 0, recs, box);
    GeomDev hg;
    pack_geom(*geom, hg, nullptr, box);
    hg.meshRoot = lay.root;
    hg.meshStride = lay.stride;
    DevBuf<GeomDev> dg;
    DevBuf<ptd::MeshUnit> drec;
    DevBuf<int> dout, dcull;
    DevBuf<float> dr, dt, dp, dn;
    UP(dg, &hg, 1);
    UP(drec, recs.data(), recs.size());
    UP(dr, rays, (size_t)n * 6);
    UP(dp, p3, (size_t)n * 3);
    UP(dn, n3, (size_t)n * 3);
    UP(dout, outside, n);
    int rc = dt.alloc(n); if (rc) return rc;
    rc = dcull.alloc(n); if (rc) return rc;
    const size_t stackBytes = (size_t)std::max(lay.stackNeed, 1) * 256 * sizeof(uint32_t);
    if (stackBytes > 64 * 1024) return fail(PT_ERR_INVALID, "pt_test_mesh_intersect: the hierarchy needs %d stack levels", lay.stackNeed);
    hipLaunchKernelGGL(k_test_mesh, dim3((unsigned)((n + 255) / 256)), dim3(256), stackBytes, 0, dg.p, reinterpret_cast<const float4 *>(drec.p),
                       dr.p, n, dt.p, dp.p, dn.p, dout.p, dcull.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(t, dt, n);
    DOWN(p3, dp, (size_t)n * 3);
    DOWN(n3, dn, (size_t)n * 3);
    DOWN(outside, dout, n);
    DOWN(culled, dcull, n);
    return PT_OK;
}

int pt_test_mesh_cull_sweep(const PtGeom *geom, const float *tris, int ntris, uint64_t seed, int64_t rays, uint64_t *culled,
                            uint64_t *violations, uint64_t *hits) {
    NEED_GPU();
    if (!geom || !tris || ntris < 1 || geom->type != PT_MESH || !culled || !violations || !hits || rays < 0)
        return fail(PT_ERR_INVALID, "pt_test_mesh_cull_sweep: bad argument");
    std::vector<ptd::MeshUnit> recs;
    float box[6];
    const ptm::MeshLayout lay = ptm::appendMesh(tris, ntris, false, recs, box);
    GeomDev hg;
    pack_geom(*geom, hg, nullptr, box);
    hg.meshRoot = lay.root;
    hg.meshStride = lay.stride;
    if (!std::isfinite(hg.cullR2)) return fail(PT_ERR_INVALID, "pt_test_mesh_cull_sweep: this mesh is never culled");
    DevBuf<GeomDev> dg;
    DevBuf<ptd::MeshUnit> drec;
    DevBuf<unsigned long long> cnt;
    UP(dg, &hg, 1);
    UP(drec, recs.data(), recs.size());
    int rc = cnt.alloc(3);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 24));
    const int per_thread = 64, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    const size_t stackBytes = (size_t)std::max(lay.stackNeed, 1) * 256 * sizeof(uint32_t);
    if (stackBytes > 64 * 1024) return fail(PT_ERR_INVALID, "pt_test_mesh_cull_sweep: the hierarchy needs %d stack levels", lay.stackNeed);
    hipLaunchKernelGGL(k_sweep_mesh_cull, dim3((unsigned)blocks), dim3(threads), stackBytes, 0, dg.p, reinterpret_cast<const float4 *>(drec.p),
                       (unsigned long long)seed, per_thread, cnt.p, cnt.p + 1, cnt.p + 2);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[3] = {0, 0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 24, hipMemcpyDeviceToHost));
    *culled = h[0];
    *violations = h[1];
    *hits = h[2];
    return PT_OK;
}

// host only: no GPU is touched
int pt_test_mesh_bvh(const float *tris, int ntris, int octant, uint32_t *units4, int *nrecs, int *stack_need) {
    if (!tris || ntris < 1 || !units4 || !nrecs || !stack_need || octant < 0 || octant > 7) return fail(PT_ERR_INVALID, "pt_test_mesh_bvh: bad argument");
    std::vector<ptd::MeshUnit> recs;
    float box[6];
    const ptm::MeshLayout lay = ptm::appendMesh(tris, ntris, false, recs, box);
    // units of 16 bytes: the triangles (three each), [one unit of padding when their count is odd,] this octant's inner nodes (two each)
    const uint32_t triUnits = (uint32_t)ptd::kMeshTriUnits * (uint32_t)ntris, pad = triUnits % 2u;
    const uint32_t total = triUnits + pad + lay.stride;
    if ((int)total > *nrecs) return fail(PT_ERR_INVALID, "pt_test_mesh_bvh: %u units do not fit %d", total, *nrecs);
    // refs rebased to this array: triangle i -> kMeshLeaf | 3 i, inner node j of the copy -> triUnits + pad + 2 j
    const uint32_t innerBase = triUnits + pad + (uint32_t)octant * lay.stride;
    auto rebase = [&](uint32_t r) { return (r & ptd::kMeshLeaf) ? r : r - innerBase + triUnits + pad; };
    memcpy(units4, recs.data(), (size_t)(triUnits + pad) * sizeof(ptd::MeshUnit));
    for (uint32_t j = 0; j < lay.stride; ++j) {
        ptd::MeshUnit n = recs[(size_t)innerBase + j];
        n.w[3] = rebase(n.w[3]);
        memcpy(units4 + 4 * ((size_t)triUnits + pad + j), &n, sizeof n);
    }
    *nrecs = (int)total;
    *stack_need = lay.stackNeed;
    return PT_OK;
}

int pt_test_sphere_cull_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled,
                              uint64_t *violations) {
    NEED_GPU();
    if (!geoms || ngeoms < 1 || !culled || !violations || rays < 0) return fail(PT_ERR_INVALID, "pt_test_sphere_cull_sweep: bad argument");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i]);
    DevBuf<GeomDev> dg;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    int rc = cnt.alloc(2);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 16));
    const int per_thread = 256, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_sphere_cull, dim3((unsigned)blocks), dim3(threads), 0, 0, dg.p, ngeoms, (unsigned long long)seed,
                       per_thread, cnt.p, cnt.p + 1);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 16, hipMemcpyDeviceToHost));
    *culled = h[0];
    *violations = h[1];
    return PT_OK;
}

int pt_test_sphere_halfline_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled, uint64_t *behind,
                                  uint64_t *violations) {
    NEED_GPU();
    if (!geoms || ngeoms < 1 || !culled || !behind || !violations || rays < 0) return fail(PT_ERR_INVALID, "pt_test_sphere_halfline_sweep: bad argument");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_SPHERE) return fail(PT_ERR_INVALID, "pt_test_sphere_halfline_sweep: spheres only");
        pack_geom(geoms[i], hg[i]);
    }
    DevBuf<GeomDev> dg;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    int rc = cnt.alloc(3);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 24));
    const int per_thread = 256, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_sphere_halfline, dim3((unsigned)blocks), dim3(threads), 0, 0, dg.p, ngeoms, (unsigned long long)seed,
                       per_thread, cnt.p, cnt.p + 1, cnt.p + 2);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[3] = {0, 0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 24, hipMemcpyDeviceToHost));
    *culled = h[0];
    *behind = h[1];
    *violations = h[2];
    return PT_OK;
}

// the sweep's table: every sphere of `geoms` with its culling data as pt_init packs it (thresholds scaled for the folded K |oc|^2 term)
static void pack_sphere_cull(const PtGeom *geoms, int ngeoms, std::vector<GeomDev> &hg, std::vector<SphereCull> &sc) {
    hg.resize(ngeoms);
    sc.clear();
    for (int i = 0; i < ngeoms; ++i) {
        pack_geom(geoms[i], hg[i]);
        if (geoms[i].type != PT_SPHERE) continue;
        SphereCull e;
        memset(&e, 0, sizeof e);
        for (int a = 0; a < 3; ++a) e.centre[a] = hg[i].centre[a];
        e.cullR2 = hg[i].cullR2;
        e.cullK = hg[i].cullK + kUnitDirSlack;
        e.geom = i;
        sc.push_back(e);
    }
    double kmax = 0.0;
    for (const SphereCull &e : sc) kmax = std::max(kmax, (double)e.cullK);
    const float sdir = std::nextafter((float)std::sqrt(1.0 / (1.0 - kmax)), INFINITY);
    for (SphereCull &e : sc)
        if (std::isfinite(e.cullR2)) e.cullR2 = std::nextafter((float)((double)e.cullR2 * (double)sdir * (double)sdir), INFINITY);
}

int pt_test_sphere_clusters(const PtGeom *geoms, int ngeoms, float *info18, int32_t *table, int32_t table_cap, int32_t *ntable) {
    if (!geoms || ngeoms < 2 || !info18 || !table || !ntable) return fail(PT_ERR_INVALID, "pt_test_sphere_clusters: bad argument");
    for (int i = 0; i < ngeoms; ++i)
        if (geoms[i].type == PT_MESH) return fail(PT_ERR_INVALID, "pt_test_sphere_clusters: spheres and cubes only");
    std::vector<GeomDev> hg;
    std::vector<SphereCull> sc;
    pack_sphere_cull(geoms, ngeoms, hg, sc);
    int n0 = 0;
    float omax = 0.0f, box[2][8];
    if (!build_sphere_clusters(geoms, ngeoms, hg, std::vector<int>(), sc, n0, omax, box))
        return fail(PT_ERR_INVALID, "pt_test_sphere_clusters: no clusters for this scene");
    if ((int)sc.size() > table_cap) return fail(PT_ERR_INVALID, "pt_test_sphere_clusters: table_cap too small (%d entries)", (int)sc.size());
    info18[0] = omax; info18[1] = (float)n0;
    for (int g = 0; g < 2; ++g) for (int q = 0; q < 8; ++q) info18[2 + 8 * g + q] = box[g][q];
    for (size_t i = 0; i < sc.size(); ++i) table[i] = sc[i].geom;
    *ntable = (int)sc.size();
    return PT_OK;
}

int pt_test_sphere_cluster_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *certified2, uint64_t *violations, float *info18) {
    NEED_GPU();
    if (!geoms || ngeoms < 2 || !certified2 || !violations || rays < 0) return fail(PT_ERR_INVALID, "pt_test_sphere_cluster_sweep: bad argument");
    // the clusters exactly as pt_init builds them for this scene (no primitive binned: binning only moves the choice of the split)
    for (int i = 0; i < ngeoms; ++i)
        if (geoms[i].type == PT_MESH) return fail(PT_ERR_INVALID, "pt_test_sphere_cluster_sweep: spheres and cubes only");
    std::vector<GeomDev> hg;
    std::vector<SphereCull> sc;
    pack_sphere_cull(geoms, ngeoms, hg, sc);
    int n0 = 0;
    float omax = 0.0f, box[2][8];
    if (!build_sphere_clusters(geoms, ngeoms, hg, std::vector<int>(), sc, n0, omax, box))
        return fail(PT_ERR_INVALID, "pt_test_sphere_cluster_sweep: no clusters for this scene");
    std::vector<GeomDev> hs;
    for (const SphereCull &e : sc) hs.push_back(hg[e.geom]);
    WallBox hb[2];
    memset(hb, 0, sizeof hb);
    for (int g = 0; g < 2; ++g)
        for (int a = 0; a < 3; ++a) { hb[g].lo[a] = box[g][a]; hb[g].hi[a] = box[g][3 + a]; }
    F3 slo = F3{INFINITY, INFINITY, INFINITY}, shi = F3{-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < ngeoms; ++i) {
        const float r = hg[i].boundR;
        if (!std::isfinite(r)) continue;
        slo = F3{std::min(slo.x, hg[i].centre[0] - r), std::min(slo.y, hg[i].centre[1] - r), std::min(slo.z, hg[i].centre[2] - r)};
        shi = F3{std::max(shi.x, hg[i].centre[0] + r), std::max(shi.y, hg[i].centre[1] + r), std::max(shi.z, hg[i].centre[2] + r)};
    }
    if (info18) {
        info18[0] = omax; info18[1] = (float)n0;
        for (int g = 0; g < 2; ++g) for (int q = 0; q < 8; ++q) info18[2 + 8 * g + q] = box[g][q];
    }
    DevBuf<GeomDev> ds;
    DevBuf<WallBox> db;
    DevBuf<unsigned long long> cnt;
    UP(ds, hs.data(), (int)hs.size());
    UP(db, hb, 2);
    int rc = cnt.alloc(3);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 24));
    const int per_thread = 64, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 22)) blocks = 1 << 22;
    hipLaunchKernelGGL(k_sweep_sphere_clusters, dim3((unsigned)blocks), dim3(threads), 0, 0, ds.p, (int)hs.size(), n0, db.p, omax, slo, shi,
                       (unsigned long long)seed, per_thread, cnt.p, cnt.p + 2);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long hc[3] = {0, 0, 0};
    HIPCHECK(hipMemcpy(hc, cnt.p, 24, hipMemcpyDeviceToHost));
    certified2[0] = hc[0];
    certified2[1] = hc[1];
    *violations = hc[2];
    return PT_OK;
}

int pt_test_wall_box_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled, uint64_t *violations) {
    NEED_GPU();
    if (!geoms || ngeoms < 1 || !culled || !violations || rays < 0) return fail(PT_ERR_INVALID, "pt_test_wall_box_sweep: bad argument");
    std::vector<GeomDev> hg(ngeoms);
    std::vector<WallBox> hw(ngeoms);
    std::vector<float> omax(ngeoms);
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_wall_box_sweep: cubes only");
        pack_geom(geoms[i], hg[i]);
        double om = 0;
        const double b = wall_box(geoms[i], hw[i], &om);
        if (b < 0) return fail(PT_ERR_INVALID, "pt_test_wall_box_sweep: cube %d is not finite", i);
        omax[i] = (float)om;                      // pt_init's bound, for a scene that consists of this wall alone
    }
    DevBuf<GeomDev> dg;
    DevBuf<WallBox> dw;
    DevBuf<float> dm;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    UP(dw, hw.data(), ngeoms);
    UP(dm, omax.data(), ngeoms);
    int rc = cnt.alloc(2);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 16));
    const int per_thread = 256, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_wall_box, dim3((unsigned)blocks), dim3(threads), 0, 0, dg.p, dw.p, dm.p, ngeoms, (unsigned long long)seed,
                       per_thread, cnt.p, cnt.p + 1);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 16, hipMemcpyDeviceToHost));
    *culled = h[0];
    *violations = h[1];
    return PT_OK;
}

// host only: no GPU is touched
int pt_test_camera_cull_tables(const PtCamera *cam, const PtGeom *geoms, int ngeoms, int32_t *rects4, int32_t *scene_rect4, int32_t *spans2) {
    if (!cam || !geoms || ngeoms < 1 || !rects4 || !scene_rect4 || !spans2) return fail(PT_ERR_INVALID, "pt_test_camera_cull_tables: bad argument");
    if (cam->resolution[0] < 1 || cam->resolution[1] < 1) return fail(PT_ERR_INVALID, "pt_test_camera_cull_tables: bad resolution");
    KParams k;
    memset(&k, 0, sizeof k);
    camera_params(*cam, k);
    k.ngeoms = ngeoms;
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i], k.pos);
    CameraCull cc;
    build_camera_cull(geoms, ngeoms, k, false, std::vector<const float *>(ngeoms, nullptr), hg, cc);
    for (int i = 0; i < ngeoms; ++i)
        for (int a = 0; a < 4; ++a) rects4[4 * i + a] = hg[i].rect[a];
    for (int a = 0; a < 4; ++a) scene_rect4[a] = cc.sceneRect[a];
    for (int y = 0; y < k.H; ++y)
        for (int i = 0; i < ngeoms; ++i) {
            int x0 = 1, x1 = 0;                                // (empty: the row's list does not hold the primitive)
            if (cc.rowOff.empty()) { x0 = hg[i].rect[0]; x1 = hg[i].rect[2]; }
            else
                for (int e = cc.rowOff[y]; e < cc.rowOff[y + 1]; ++e)
                    if (cc.rowIdx[2 * e] == i) { x0 = cc.rowIdx[2 * e + 1] & 0xffff; x1 = cc.rowIdx[2 * e + 1] >> 16; }
            spans2[2 * ((size_t)y * ngeoms + i)] = x0;
            spans2[2 * ((size_t)y * ngeoms + i) + 1] = x1;
        }
    return PT_OK;
}

int pt_test_camera_cull_sweep(const PtCamera *cam, const PtGeom *geoms, int ngeoms, int samples, uint64_t *hits, uint64_t *culled,
                              uint64_t *violations) {
    NEED_GPU();
    if (!cam || !geoms || ngeoms < 1 || samples < 1 || !hits || !culled || !violations) return fail(PT_ERR_INVALID, "pt_test_camera_cull_sweep: bad argument");
    if (cam->resolution[0] < 1 || cam->resolution[1] < 1 || (long long)cam->resolution[0] * cam->resolution[1] > (1ll << 26))
        return fail(PT_ERR_INVALID, "pt_test_camera_cull_sweep: bad resolution");
    for (int i = 0; i < ngeoms; ++i)
        if (geoms[i].type != PT_SPHERE && geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_camera_cull_sweep: spheres and cubes only");
    // exactly what pt_init derives: camera constants, packed primitives (object-space eye), rectangles, union, row lists
    KParams k;
    memset(&k, 0, sizeof k);
    camera_params(*cam, k);
    k.ngeoms = ngeoms;
    magic_divisor((uint32_t)k.W, k.magicW, k.shiftW);
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) pack_geom(geoms[i], hg[i], k.pos);
    CameraCull cc;
    build_camera_cull(geoms, ngeoms, k, false, std::vector<const float *>(ngeoms, nullptr), hg, cc);
    for (int a = 0; a < 4; ++a) k.sceneRect[a] = cc.sceneRect[a];
    DevBuf<GeomDev> dg;
    DevBuf<int> doff, didx;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    if (!cc.rowOff.empty()) {
        UP(doff, cc.rowOff.data(), cc.rowOff.size());
        UP(didx, cc.rowIdx.data(), cc.rowIdx.size());
    }
    int rc = cnt.alloc(3);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 24));
    const int npix = k.W * k.H;
    hipLaunchKernelGGL(k_sweep_camera_cull, GRID(npix), k, dg.p, doff.p, didx.p, samples, cnt.p, cnt.p + 1, cnt.p + 2);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[3] = {0, 0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 24, hipMemcpyDeviceToHost));
    *hits = h[0];
    *culled = h[1];
    *violations = h[2];
    return PT_OK;
}

int pt_test_camera_cull_margin(const PtCamera *cam, const PtGeom *geoms, int ngeoms, int samples, double *worst_fraction, uint64_t *needed) {
    NEED_GPU();
    if (!cam || !geoms || ngeoms < 1 || samples < 1 || !worst_fraction || !needed) return fail(PT_ERR_INVALID, "pt_test_camera_cull_margin: bad argument");
    if (cam->resolution[0] < 1 || cam->resolution[1] < 1 || (long long)cam->resolution[0] * cam->resolution[1] > (1ll << 26))
        return fail(PT_ERR_INVALID, "pt_test_camera_cull_margin: bad resolution");
    for (int i = 0; i < ngeoms; ++i)
        if (geoms[i].type != PT_SPHERE && geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_camera_cull_margin: spheres and cubes only");
    KParams k;
    memset(&k, 0, sizeof k);
    camera_params(*cam, k);
    k.ngeoms = ngeoms;
    magic_divisor((uint32_t)k.W, k.magicW, k.shiftW);
    std::vector<GeomDev> hg(ngeoms);
    std::vector<double> infl(4 * (size_t)ngeoms, 0.0);
    for (int i = 0; i < ngeoms; ++i) {
        pack_geom(geoms[i], hg[i], k.pos);
        double lo[3], hi[3];
        inflated_object_box(geoms[i], k.pos, nullptr, lo, hi);
        int rect[4];
        std::vector<std::pair<double, double>> hull;
        project_geom(geoms[i], k, rect, nullptr, &hull);
        for (int a = 0; a < 3; ++a) infl[4 * i + a] = geoms[i].type == PT_CUBE ? hi[a] - 0.5 : hi[0];
        infl[4 * i + 3] = hull.empty() ? 0.0 : 1.0;          // (culling switched off for this primitive: nothing to measure)
    }
    DevBuf<GeomDev> dg;
    DevBuf<double> di;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    UP(di, infl.data(), infl.size());
    int rc = cnt.alloc(2);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 16));
    const int npix = k.W * k.H;
    hipLaunchKernelGGL(k_sweep_camera_cull_margin, GRID(npix), k, dg.p, di.p, samples, cnt.p, cnt.p + 1);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 16, hipMemcpyDeviceToHost));
    memcpy(worst_fraction, &h[0], sizeof(double));
    *needed = h[1];
    return PT_OK;
}

int pt_test_wall_plane_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, int32_t *nplane, uint64_t *certified,
                             uint64_t *violations, uint64_t *single) {
    NEED_GPU();
    if (!geoms || ngeoms < 1 || !nplane || !certified || !violations || !single || rays < 0) return fail(PT_ERR_INVALID, "pt_test_wall_plane_sweep: bad argument");
    // the walls exactly as pt_init chooses and numbers them (no primitive of the set is binned here)
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_wall_plane_sweep: cubes only");
        pack_geom(geoms[i], hg[i]);
    }
    KParams k;
    memset(&k, 0, sizeof k);
    std::vector<WallBox> hw(kWallMax);
    std::vector<int> wallGeom;
    choose_walls(geoms, ngeoms, hg, k, hw, wallGeom);
    *nplane = k.nSlotWalls;
    *certified = *violations = *single = 0;
    if (k.nWalls < 1 || k.nSlotWalls < 1) return PT_OK;
    std::vector<GeomDev> wg(k.nWalls);
    for (int w = 0; w < k.nWalls; ++w) wg[w] = hg[wallGeom[w]];
    DevBuf<GeomDev> dg;
    DevBuf<WallBox> dw;
    DevBuf<unsigned long long> cnt;
    UP(dg, wg.data(), k.nWalls);
    UP(dw, hw.data(), k.nWalls);
    int rc = cnt.alloc(3);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 24));
    const int per_thread = 256, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_wall_planes, dim3((unsigned)blocks), dim3(threads), 0, 0, k, dg.p, dw.p, (unsigned long long)seed, per_thread,
                       cnt.p, cnt.p + 1, cnt.p + 2);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[3] = {0, 0, 0};
    HIPCHECK(hipMemcpy(h, cnt.p, 24, hipMemcpyDeviceToHost));
    *certified = h[0];
    *violations = h[1];
    *single = h[2];
    return PT_OK;
}

int pt_test_box_fast_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t counts[4], uint64_t div_mismatches[2]) {
    NEED_GPU();
    if (!geoms || ngeoms < 1 || !counts || !div_mismatches || rays < 0) return fail(PT_ERR_INVALID, "pt_test_box_fast_sweep: bad argument");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_box_fast_sweep: cubes only");
        pack_geom(geoms[i], hg[i]);
    }
    DevBuf<GeomDev> dg;
    DevBuf<unsigned long long> cnt;
    UP(dg, hg.data(), ngeoms);
    int rc = cnt.alloc(8);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 64));
    DevBuf<float> dump;
    const bool verbose = getenv("PT_AMD_VERBOSE") && atoi(getenv("PT_AMD_VERBOSE"));
    if (verbose && (rc = dump.alloc(8 * 24))) return rc;
    const int per_thread = 256, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_box_fast, dim3((unsigned)blocks), dim3(threads), 0, 0, dg.p, ngeoms, (unsigned long long)seed, per_thread, cnt.p,
                       verbose ? dump.p : nullptr);
    hipLaunchKernelGGL(k_sweep_div_unscaled, dim3(1 << 12), dim3(256), 0, 0, (unsigned long long)seed, 1024, cnt.p + 4);   // 2^20 threads
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[8];
    HIPCHECK(hipMemcpy(h, cnt.p, 64, hipMemcpyDeviceToHost));
    if (verbose && h[6]) {                                // (experiments: the first mismatching rays)
        float r[8 * 24];
        HIPCHECK(hipMemcpy(r, dump.p, sizeof r, hipMemcpyDeviceToHost));
        for (int i = 0; i < (int)std::min<unsigned long long>(h[6], 8); ++i) {
            const float *q = r + 24 * i;
            fprintf(stderr, "box sweep mismatch: geom %d org %.9g %.9g %.9g dir %.9g %.9g %.9g | t fast %.9g exact %.9g early %.9g outside %g %g | P %.9g %.9g %.9g / %.9g %.9g %.9g | n %a %a %a / %a %a %a\n",
                    (int)q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12], q[13], q[14], q[15], q[16], q[17], q[18], q[19], q[20], q[21], q[22], q[23]);
        }
    }
    for (int i = 0; i < 4; ++i) counts[i] = h[i];
    div_mismatches[0] = h[4];
    div_mismatches[1] = h[5];
    return PT_OK;
}

int pt_test_slab_quotients(const float *o, const float *d, int n, float *t1, float *t2, float *ref1, float *ref2) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, b, q1, q2, r1, r2;
    UP(a, o, n);
    UP(b, d, n);
    int rc;
    if ((rc = q1.alloc(n)) || (rc = q2.alloc(n)) || (rc = r1.alloc(n)) || (rc = r2.alloc(n))) return rc;
    hipLaunchKernelGGL(k_test_slab_quotients, GRID(n), a.p, b.p, n, q1.p, q2.p, r1.p, r2.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(t1, q1, n);
    DOWN(t2, q2, n);
    DOWN(ref1, r1, n);
    DOWN(ref2, r2, n);
    return PT_OK;
}

int pt_test_slab_quotients_sweep(uint64_t seed, int64_t pairs, uint64_t *mismatches) {
    NEED_GPU();
    if (!mismatches || pairs < 0) return fail(PT_ERR_INVALID, "pt_test_slab_quotients_sweep: bad argument");
    DevBuf<unsigned long long> m;
    int rc = m.alloc(1);
    if (rc) return rc;
    HIPCHECK(hipMemset(m.p, 0, 8));
    const int per_thread = 1024, threads = 256;
    long long blocks = (pairs + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(k_sweep_slab_quotients, dim3((unsigned)blocks), dim3(threads), 0, 0, (unsigned long long)seed, per_thread, m.p);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    HIPCHECK(hipMemcpy(&h, m.p, 8, hipMemcpyDeviceToHost));
    *mismatches = h;
    return PT_OK;
}

int pt_test_unscaled_sqrt_sweep(uint64_t mismatches[4]) {
    NEED_GPU();
    if (!mismatches) return fail(PT_ERR_INVALID, "pt_test_unscaled_sqrt_sweep: bad argument");
    DevBuf<unsigned long long> m;
    int rc = m.alloc(4);
    if (rc) return rc;
    HIPCHECK(hipMemset(m.p, 0, 32));
    hipLaunchKernelGGL(k_sweep_unscaled_sqrt, dim3(1 << 14), dim3(256), 0, 0, m.p);   // 2^22 threads x 2^10 patterns
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    HIPCHECK(hipMemcpy(h, m.p, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) mismatches[i] = h[i];
    return PT_OK;
}

#ifdef PT_PROBE_TIMELINE
// instrumented build only (make timeline): reads and clears the per-phase cycle sums of pt_device.h
extern "C" int pt_probe_timeline(uint64_t out[128]) {      // [0, 64) cycles (later bounces, then + 32 the camera-ray bounce), [64, 128) intervals
    NEED_GPU();
    unsigned long long h[128], z[64] = {0};
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ptd::g_phaseT), 64 * sizeof(unsigned long long)));
    HIPCHECK(hipMemcpyFromSymbol(h + 64, HIP_SYMBOL(ptd::g_phaseN), 64 * sizeof(unsigned long long)));
    HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_phaseT), z, sizeof z));
    HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_phaseN), z, sizeof z));
    for (int i = 0; i < 128; ++i) out[i] = h[i];
    return PT_OK;
}
#elif defined(PT_PROBE)
// instrumented build only (make probe): the residency census of k_bounce -- out[k] = number of CUs on which at most k
// workgroups of it were ever resident together (k = 0..15); cleared by the call
extern "C" int pt_probe_census(uint32_t out[16]) {
    NEED_GPU();
    static unsigned int h[4096], z[4096];
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ptd::g_censusMax), sizeof h));
    HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_censusMax), z, sizeof z));
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (int i = 0; i < 4096; ++i)
        if (h[i]) out[h[i] < 15 ? h[i] : 15]++;
    return PT_OK;
}
// instrumented build only (make probe): reads and clears the phase counters of pt_device.h
extern "C" int pt_probe_read(uint64_t out[64]) {
    NEED_GPU();
    unsigned long long h[64], z[64] = {0};
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ptd::g_probe), sizeof h));
    HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_probe), z, sizeof z));
    for (int i = 0; i < 64; ++i) out[i] = h[i];
    return PT_OK;
}
#endif

int pt_test_hemisphere(const float *normals3, const int32_t *iid3, int n, float *out3) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, o;
    DevBuf<int> b;
    UP(a, normals3, (size_t)n * 3);
    UP(b, iid3, (size_t)n * 3);
    int rc = o.alloc((size_t)n * 3); if (rc) return rc;
    hipLaunchKernelGGL(k_test_hemisphere, GRID(n), a.p, b.p, n, o.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(out3, o, (size_t)n * 3);
    return PT_OK;
}

int pt_test_pow(const float *x, const float *e, int n, float *out) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, b, o;
    UP(a, x, n);
    UP(b, e, n);
    int rc = o.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_pow, GRID(n), a.p, b.p, n, o.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(out, o, n);
    return PT_OK;
}

int pt_test_sincos(const float *x, int n, float *s, float *c) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, ds, dc;
    UP(a, x, n);
    int rc = ds.alloc(n); if (rc) return rc;
    rc = dc.alloc(n); if (rc) return rc;
    hipLaunchKernelGGL(k_test_sincos, GRID(n), a.p, n, ds.p, dc.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(s, ds, n);
    DOWN(c, dc, n);
    return PT_OK;
}

int pt_test_reflect_refract(const float *I3, const float *N3, const float *eta, int n, float *refl3, float *refr3) {
    NEED_GPU();
    if (n <= 0) return PT_OK;
    DevBuf<float> a, b, e, r1, r2;
    UP(a, I3, (size_t)n * 3);
    UP(b, N3, (size_t)n * 3);
    UP(e, eta, n);
    int rc = r1.alloc((size_t)n * 3); if (rc) return rc;
    rc = r2.alloc((size_t)n * 3); if (rc) return rc;
    hipLaunchKernelGGL(k_test_reflect_refract, GRID(n), a.p, b.p, e.p, n, r1.p, r2.p);
    HIPCHECK(hipDeviceSynchronize());
    DOWN(refl3, r1, (size_t)n * 3);
    DOWN(refr3, r2, (size_t)n * 3);
    return PT_OK;
}

#endif  // PT_TEST_API

}  // extern "C"
#pragma GCC visibility pop
