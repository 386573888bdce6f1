// pt_test_kernels.h -- device primitives exposed one by one for the parity tests (include/pt_amd.h, pt_test_*):
// the same __device__ functions the render kernels call.  Included by pt_api.hip only.
#pragma once
#include "pt_trace.h"

namespace ptk {

// (camera rays as standalone functions: the render kernel builds them in registers inside k_bounce<true, ...>, from a tile's wave-uniform
// position; these restate spec S2 per pixel for pt_debug_trace_paths(bounces = 0) and the screen-space culling sweeps)
// ---- camera ray of the j-th pixel of this shard (spec S2) ------------------------------------------
// pixel coordinates and global pixel index of the shard's j-th pixel
__device__ __forceinline__ void shardPixel(const KParams &prm, int j, int &pix, int &x, int &y) {
    const int lr = (int)fastDiv((uint32_t)j, prm.magicW, prm.shiftW);
    x = j - lr * prm.W;
    y = lr * prm.shardCount + prm.shardRank;
    pix = x + y * prm.W;
}
__device__ __forceinline__ void cameraRay(const KParams &prm, int iter, int j, int &pix, int &x, int &y, F3 &org, F3 &dir) {
    shardPixel(prm, j, pix, x, y);
    cameraRayAt(prm, iterationHash(iter, 0), pix, x, y, org, dir);
}

// camera rays alone, for pt_debug_trace_paths(bounces = 0)
__global__ __launch_bounds__(kBlock) void k_debug_camera_rays(KParams prm, int iter, float *o3, float *d3, int *pixOut) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= prm.nLocal) return;
    int pix, x, y;
    F3 org, dir;
    cameraRay(prm, iter, j, pix, x, y, org, dir);
    o3[3 * j] = org.x; o3[3 * j + 1] = org.y; o3[3 * j + 2] = org.z;
    d3[3 * j] = dir.x; d3[3 * j + 1] = dir.y; d3[3 * j + 2] = dir.z;
    pixOut[j] = pix;
}


// ---- primitive test kernels (device functions exactly as the render kernels use them) -------------------
__global__ void k_test_pow(const float *x, const float *e, int n, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = powPoly(x[i], e[i]);
}
__global__ void k_test_utilhash(const uint32_t *in, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = utilhash(in[i]);
}
__global__ void k_test_rng(const uint32_t *seeds, int nseeds, int ndraws, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseeds) return;
    Rng r = seedEngine(seeds[i]);
    for (int k = 0; k < ndraws; ++k) out[(size_t)i * ndraws + k] = u01(r);
}
__global__ void k_test_intersect(const GeomDev *geoms, const int *gidx, const float *rays, int n, float *t, float *p3,
                                 float *n3, int *outside) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GeomDev G = geoms[gidx[i]];
    F3 ro = f3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    F3 rd = f3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    F3 P = f3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]);
    F3 N = f3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
    bool o = outside[i] != 0;
    // odd lanes take the early-miss variant so both instantiations are checked against the golden vectors
    // the certain-miss shortcut must agree with the full test on every golden vector
    F3 nsrc = f3(0, 0, 0);
    const bool cull = certainMiss(G, ro, rd, dot(rd, rd));
    if (cull && (G.type == 0 ? sphereIntersectionTest(G, ro, rd, P, nsrc, o) : boxIntersectionTest<false>(G, ro, rd, P, nsrc, o)) != -1.0f) {
        t[i] = __builtin_nanf("");
        return;
    }
    t[i] = G.type == 0 ? (cull ? -1.0f : sphereIntersectionTest(G, ro, rd, P, nsrc, o))
         : ((i & 1) ? boxIntersectionTest<true>(G, ro, rd, P, nsrc, o) : boxIntersectionTest<false>(G, ro, rd, P, nsrc, o));
    if (t[i] != -1.0f) N = hitNormal(G, nsrc, o);   // the normal is an output of the reference's tests: same values here
    p3[3 * i] = P.x; p3[3 * i + 1] = P.y; p3[3 * i + 2] = P.z;
    n3[3 * i] = N.x; n3[3 * i + 1] = N.y; n3[3 * i + 2] = N.z;
    outside[i] = o ? 1 : 0;
}
// rays against one triangle mesh: the render kernels' test, plus the verdict of the bounding-ball test
// (blocks of 256 threads; dynamic LDS = the lanes' traversal stacks, [levels][256] words)
__global__ void k_test_mesh(const GeomDev *geom, const float4 *recs, const float *rays, int n, float *t,
                            float *p3, float *n3, int *outside, int *culled) {
    extern __shared__ uint32_t s_meshStack[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GeomDev G = *geom;
    F3 ro = f3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    F3 rd = f3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    F3 P = f3(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]);
    F3 N = f3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]);
    bool o = outside[i] != 0;
    F3 nsrc = f3(0, 0, 0);
    const bool cull = certainMiss(G, ro, rd, dot(rd, rd));
    const float tt = meshIntersectionTest<false, 256>(G, recs, G.meshRoot, G.meshStride, s_meshStack + threadIdx.x, ro, rd, P, nsrc, o);
    culled[i] = cull ? 1 : 0;
    t[i] = cull && tt != -1.0f ? __builtin_nanf("") : tt;
    if (tt != -1.0f) N = hitNormal(G, nsrc, o);
    p3[3 * i] = P.x; p3[3 * i + 1] = P.y; p3[3 * i + 2] = P.z;
    n3[3 * i] = N.x; n3[3 * i + 1] = N.y; n3[3 * i + 2] = N.z;
    outside[i] = o ? 1 : 0;
}
// certainMiss soundness sweep for ONE mesh geom: rays as in k_sweep_sphere_cull (origins 1/64 .. 64 units from the bounding
// ball's centre, aimed within ~1.3 radii of it: hits, grazes, near misses); a culled ray that the full walk hits is a violation
__global__ void k_sweep_mesh_cull(const GeomDev *geom, const float4 *recs, unsigned long long seed, int per_thread,
                                  unsigned long long *culled, unsigned long long *violations, unsigned long long *hits) {
    extern __shared__ uint32_t s_meshStack[];
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nv = 0, nh = 0;
    const GeomDev G = *geom;
    const F3 c = f3(G.centre[0], G.centre[1], G.centre[2]);
    const float R = __builtin_sqrtf(G.cullR2);
    for (int k = 0; k < per_thread; ++k) {
        float u[8];
        for (int j = 0; j < 8; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const float dist = R * __builtin_exp2f(u[0] * 12.0f - 6.0f);             // 1/64 .. 64 bounding radii
        const F3 od = normalize(f3(u[1] - 0.5f, u[2] - 0.5f, u[3] - 0.5f));
        const F3 org = c + od * dist;
        const F3 tgt = c + f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f) * (2.6f * R);
        F3 dir = normalize(tgt - org);
        if (u[7] < 0.1f) dir = -dir;
        const bool cull = certainMiss(G, org, dir, dot(dir, dir));
        F3 P, N;
        bool o;
        // (every ray takes the walk: the hit count shows that the sweep does probe the mesh)
        const float t = meshIntersectionTest<false, 256>(G, recs, G.meshRoot, G.meshStride, s_meshStack + threadIdx.x, org, dir, P, N, o);
        nc += cull ? 1u : 0u;
        nh += t != -1.0f ? 1u : 0u;
        nv += cull && t != -1.0f ? 1u : 0u;
    }
    if (nc) atomicAdd(culled, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
    if (nh) atomicAdd(hits, (unsigned long long)nh);
}
// certainMiss soundness sweep: pseudo-random rays (origins up to ~60 units away, aimed near the primitive's bounding
// ball so that grazing cases are dense) against every primitive of `geoms`; counts culled rays and VIOLATIONS
// (culled although the full test returns a hit).
__global__ void k_sweep_sphere_cull(const GeomDev *geoms, int ngeoms, unsigned long long seed, int per_thread,
                                    unsigned long long *culled, unsigned long long *violations) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nv = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[8];
        for (int j = 0; j < 8; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const GeomDev G = geoms[(blockIdx.x + k) % ngeoms];
        const F3 c = f3(G.centre[0], G.centre[1], G.centre[2]);
        const float dist = __builtin_exp2f(u[0] * 12.0f - 6.0f);                 // 1/64 .. 64 units
        const F3 od = normalize(f3(u[1] - 0.5f, u[2] - 0.5f, u[3] - 0.5f));
        const F3 org = c + od * dist;
        // aim at a point within ~1.3 bounding radii of the centre: hits, grazes and near misses
        const float R = __builtin_sqrtf(G.cullR2);
        const F3 tgt = c + f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f) * (2.6f * R);
        F3 dir = normalize(tgt - org);
        if (u[7] < 0.1f) dir = -dir;
        if (certainMiss(G, org, dir, dot(dir, dir))) {
            ++nc;
            F3 P, N;
            bool o;
            const float t = G.type == 0 ? sphereIntersectionTest(G, org, dir, P, N, o) : boxIntersectionTest<false>(G, org, dir, P, N, o);
            if (t != -1.0f) ++nv;   // (N receives the normal source here)
        }
    }
    if (nc) atomicAdd(culled, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// sphereHalfLineExcess soundness sweep (the certificate of k_bounce's packed sphere sweep, with the kernel's own approximate
// normalisation of the direction): three families of rays per sphere -- (a) k_sweep_sphere_cull's (origins 1/64 .. 64 units away, aimed
// within ~1.3 bounding radii of the centre; one in ten reversed), (b) origins ON the sphere's surface moved 1e-3 along the normal, as a
// scatter leaves them, any direction (the sphere just left lies behind half of them, by next to nothing), (c) origins within 2 % of the
// bounding ball's surface, inside and outside, any direction, directions not unit (|dir| in 0.999 .. 1.001 and 0.25 .. 4).
// Counts certified misses, those certified with the centre BEHIND the origin, and VIOLATIONS (certified although the full test hits).
__global__ void k_sweep_sphere_halfline(const GeomDev *geoms, int ngeoms, unsigned long long seed, int per_thread,
                                        unsigned long long *culled, unsigned long long *behind, unsigned long long *violations) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nb = 0, nv = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[10];
        for (int j = 0; j < 10; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const GeomDev G = geoms[(blockIdx.x + k) % ngeoms];
        const F3 c = f3(G.centre[0], G.centre[1], G.centre[2]);
        const float R = __builtin_sqrtf(G.cullR2);
        const F3 od = normalize(f3(u[1] - 0.5f, u[2] - 0.5f, u[3] - 0.5f));
        const int family = k % 3;
        F3 org, dir;
        if (family == 0) {
            org = c + od * __builtin_exp2f(u[0] * 12.0f - 6.0f);
            dir = normalize(c + f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f) * (2.6f * R) - org);
            if (u[7] < 0.1f) dir = -dir;
        } else if (family == 1) {
            // a point of the primitive itself: the image of a unit-diameter object-space point (a cube: pushed out to its surface),
            // then 1e-3 along +-the world normal (a cube: the direction away from its centre)
            F3 pobj = od * 0.5f;
            if (G.type == 1) pobj = pobj * (0.5f / __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(pobj.x), __builtin_fabsf(pobj.y)), __builtin_fabsf(pobj.z)));
            const F3 P = mulMV(G.xf, pobj, 1.0f);
            const F3 N = G.type == 1 ? normalize(P - c) : normalize(mulMV(G.invT, pobj, 0.0f));
            org = P + N * (u[0] < 0.5f ? 0.001f : -0.001f);
            dir = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f));
        } else {
            org = c + od * (R * (0.98f + 0.04f * u[0]));
            dir = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f));
        }
        dir = dir * (u[8] < 0.5f ? 1.0f : (u[8] < 0.75f ? 0.999f + 0.002f * u[9] : __builtin_exp2f(4.0f * u[9] - 2.0f)));
        const float dd = dot(dir, dir);
        // the kernel's own form (round 4): the K |oc|^2 term folded into the direction, factor and threshold rounded upwards as pt_init does
        const float K = G.cullK + kUnitDirSlack;
        const float sdir = __uint_as_float(__float_as_uint(__builtin_sqrtf(1.0f / (1.0f - K)) * 1.0000002f) + 1u);     // (positive: one ulp up)
        const float R2s = __uint_as_float(__float_as_uint(G.cullR2 * sdir * sdir * 1.0000002f) + 1u);
        const F3 dhat = unitDirectionScaled(dir, dd, sdir);
        if (sphereHalfLineExcessScaled(c, org, dhat) > R2s) {
            ++nc;
            if (dot(org - c, dhat) > 0.0f) ++nb;
            F3 P, N;
            bool o;
            const float t = G.type == 1 ? boxIntersectionTest<false>(G, org, dir, P, N, o) : sphereIntersectionTest(G, org, dir, P, N, o);
            if (t != -1.0f) ++nv;
        }
    }
    if (nc) atomicAdd(culled, (unsigned long long)nc);
    if (nb) atomicAdd(behind, (unsigned long long)nb);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// wallCertainMiss soundness sweep: pseudo-random rays against every cube of `geoms` and its inflated world box `walls`
// (origins inside the |x| + |y| + |z| bound the render kernel certifies under, from touching the cube to far away; aimed at
// points on and around the cube so that grazes, edge-on plates and corner passes are dense; directions with exact zeros
// mixed in); counts certified misses and VIOLATIONS (certified although the full test returns a hit).
__global__ void k_sweep_wall_box(const GeomDev *geoms, const WallBox *walls, const float *omax, int ngeoms, unsigned long long seed,
                                 int per_thread, unsigned long long *culled, unsigned long long *violations) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nv = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[10];
        for (int j = 0; j < 10; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const int gi = (blockIdx.x + k) % ngeoms;
        const GeomDev G = geoms[gi];
        const WallBox W = walls[gi];
        const F3 lo = f3(W.lo[0], W.lo[1], W.lo[2]), hi = f3(W.hi[0], W.hi[1], W.hi[2]);
        const F3 c = (lo + hi) * 0.5f, h = (hi - lo) * 0.5f;
        // a target on / near the box: inside it, or on a shell 0.9 .. 1.2 of its half extents (where the margin matters)
        const float shell = u[9] < 0.5f ? 0.9f + 0.3f * u[8] : u[8];
        const F3 tgt = c + f3((2 * u[0] - 1) * h.x, (2 * u[1] - 1) * h.y, (2 * u[2] - 1) * h.z) * shell;
        const float dist = __builtin_exp2f(u[3] * 14.0f - 8.0f) * (h.x + h.y + h.z);     // 2^-8 .. 2^6 box sizes away
        F3 od = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f));
        if (u[7] < 0.15f) od = f3(u[7] < 0.05f ? 1.0f : 0.0f, (u[7] >= 0.05f && u[7] < 0.1f) ? 1.0f : 0.0f, u[7] >= 0.1f ? 1.0f : 0.0f);   // axis-parallel
        const F3 org = tgt + od * dist;
        F3 dir = normalize(tgt - org);
        if (u[7] > 0.9f) dir = -dir;
        const float l1 = (__builtin_fabsf(org.x) + __builtin_fabsf(org.y)) + __builtin_fabsf(org.z);
        if (!(l1 <= omax[gi])) continue;
        const F3 inv = f3(__builtin_amdgcn_rcpf(dir.x), __builtin_amdgcn_rcpf(dir.y), __builtin_amdgcn_rcpf(dir.z));
        if (wallCertainMiss(W, org, inv)) {
            ++nc;
            F3 P, N;
            bool o;
            if (boxIntersectionTest<false>(G, org, dir, P, N, o) != -1.0f) ++nv;
        }
    }
    if (nc) atomicAdd(culled, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// Soundness sweep of the sphere CLUSTERS' box certificates (k_bounce: CLUSTER; pt_api.hip: build_sphere_clusters): `spheres` in table
// order, the first n0 of them cluster 0; box[g] = cluster g's inflated world box; certificates are issued for origins with
// |x| + |y| + |z| <= omax.  Rays in four families -- leaving a sphere's own surface as a scatter does (+-1e-3 along the normal); from a point
// of the scene's extent (the walls, the interior) towards a cluster's box, its shell 0.9 .. 1.2 of the half extents included, where the
// inflation is what decides; from close to a box outwards and along it; axis-parallel directions with exact zeros mixed in.  A box
// certified as missed sends the ray through the FULL test of every sphere of that cluster: a hit is a VIOLATION (must be 0).
__global__ void k_sweep_sphere_clusters(const GeomDev *spheres, int nspheres, int n0, const WallBox *box, float omax, F3 sceneLo, F3 sceneHi,
                                        unsigned long long seed, int per_thread, unsigned long long *certified, unsigned long long *violations) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc0 = 0, nc1 = 0, nv = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[12];
        for (int j = 0; j < 12; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const int gsel = u[10] < 0.5f ? 0 : 1;
        const WallBox W = box[gsel];
        const F3 lo = f3(W.lo[0], W.lo[1], W.lo[2]), hi = f3(W.hi[0], W.hi[1], W.hi[2]);
        const F3 c = (lo + hi) * 0.5f, h = (hi - lo) * 0.5f;
        const float shell = u[9] < 0.5f ? 0.9f + 0.3f * u[8] : u[8];
        const F3 tgt = c + f3((2 * u[0] - 1) * h.x, (2 * u[1] - 1) * h.y, (2 * u[2] - 1) * h.z) * shell;
        const int family = k & 3;
        F3 org, dir;
        if (family == 0) {              // a scatter off a sphere (any of them), any direction
            const GeomDev &G = spheres[(int)(u[3] * (float)nspheres) % nspheres];
            const F3 pobj = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f)) * 0.5f;
            const F3 P = mulMV(G.xf, pobj, 1.0f);
            const F3 N = normalize(mulMV(G.invT, pobj, 0.0f));
            org = P + N * (u[7] < 0.5f ? 0.001f : -0.001f);
            dir = u[11] < 0.5f ? normalize(f3(u[0] - 0.5f, u[1] - 0.5f, u[2] - 0.5f)) : normalize(tgt - org);
        } else if (family == 1) {       // from the scene's extent (a wall, the interior) towards the box and its shell
            org = f3(sceneLo.x + u[4] * (sceneHi.x - sceneLo.x), sceneLo.y + u[5] * (sceneHi.y - sceneLo.y), sceneLo.z + u[6] * (sceneHi.z - sceneLo.z));
            if (u[7] < 0.5f) {          // ... on one of the extent's six faces
                const int face = (int)(u[7] * 12.0f);
                if (face == 0) org.x = sceneLo.x; else if (face == 1) org.x = sceneHi.x; else if (face == 2) org.y = sceneLo.y;
                else if (face == 3) org.y = sceneHi.y; else if (face == 4) org.z = sceneLo.z; else org.z = sceneHi.z;
            }
            dir = normalize(tgt - org);
        } else if (family == 2) {       // from close to the box (2^-8 .. 2^2 of its size away from the target), towards it or away
            const float dist = __builtin_exp2f(u[3] * 10.0f - 8.0f) * (h.x + h.y + h.z);
            const F3 od = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f));
            org = tgt + od * dist;
            dir = normalize(tgt - org);
            if (u[7] > 0.8f) dir = -dir;
        } else {                        // axis-parallel and plane-parallel directions (exact zeros), past the box's faces
            const float dist = __builtin_exp2f(u[3] * 10.0f - 8.0f) * (h.x + h.y + h.z);
            const int ax = (int)(u[7] * 3.0f) % 3;
            F3 od = f3(ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f);
            if (u[11] < 0.5f) od = normalize(f3(ax == 0 ? 0.0f : u[4] - 0.5f, ax == 1 ? 0.0f : u[5] - 0.5f, ax == 2 ? 0.0f : u[6] - 0.5f));
            org = tgt + od * dist;
            dir = -od;
            if (u[8] > 0.8f) dir = od;
        }
        // (directions of a scatter are unit to a few ulp; lengths far from 1 as well: the certificate must not depend on it)
        dir = dir * (u[9] < 0.7f ? 1.0f : __builtin_exp2f(4.0f * u[8] - 2.0f));
        const float l1 = (__builtin_fabsf(org.x) + __builtin_fabsf(org.y)) + __builtin_fabsf(org.z);
        if (!(l1 <= omax)) continue;
        const F3 inv = f3(__builtin_amdgcn_rcpf(dir.x), __builtin_amdgcn_rcpf(dir.y), __builtin_amdgcn_rcpf(dir.z));
        for (int g = 0; g < 2; ++g) {
            if (!wallCertainMiss(box[g], org, inv)) continue;
            if (g) ++nc1; else ++nc0;
            const int s0 = g ? n0 : 0, s1 = g ? nspheres : n0;
            for (int si = s0; si < s1; ++si) {
                F3 P, N;
                bool o;
                if (sphereIntersectionTest(spheres[si], org, dir, P, N, o) != -1.0f) ++nv;
            }
        }
    }
    if (nc0) atomicAdd(certified, (unsigned long long)nc0);
    if (nc1) atomicAdd(certified + 1, (unsigned long long)nc1);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// Soundness sweep of the sphere GROUPS' bounding balls (k_bounce<..., GROUPS>; pt_host_scene.h: build_sphere_groups): `spheres` in table order,
// group g = entries [16 g, 16 g + 16), groups[g] = its ball {centre, threshold of the scaled certificate}; certificates are issued for origins
// with |x| + |y| + |z| <= omax.  Rays: scatters off a sphere's own surface; from the scene's extent towards a group's ball and its shell (0.9 ..
// 1.2 of the radius, where the margins decide); from close by, towards it or away; axis- and plane-parallel directions with exact zeros; unit
// and non-unit lengths.  A group certified as missed sends the ray through the FULL test of each of its members: a hit is a VIOLATION (0).
__global__ void k_sweep_sphere_groups(const GeomDev *spheres, int nspheres, const SphereCull *groups, int ngroups, float sdir, float omax, F3 sceneLo,
                                      F3 sceneHi, unsigned long long seed, int per_thread, unsigned long long *certified, unsigned long long *violations) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nv = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[12];
        for (int j = 0; j < 12; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const SphereCull Gs = groups[(int)(u[10] * (float)ngroups) % ngroups];
        const F3 c = f3(Gs.centre[0], Gs.centre[1], Gs.centre[2]);
        const float R = __builtin_sqrtf(Gs.cullR2) / sdir;
        const float shell = u[9] < 0.5f ? 0.9f + 0.3f * u[8] : u[8];
        const F3 tgt = c + normalize(f3(u[0] - 0.5f, u[1] - 0.5f, u[2] - 0.5f)) * (R * shell);
        const int family = k & 3;
        F3 org, dir;
        if (family == 0) {              // a scatter off a sphere (any of them), any direction or at the group's shell
            const GeomDev &G = spheres[(int)(u[3] * (float)nspheres) % nspheres];
            const F3 pobj = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f)) * 0.5f;
            const F3 P = mulMV(G.xf, pobj, 1.0f);
            const F3 N = normalize(mulMV(G.invT, pobj, 0.0f));
            org = P + N * (u[7] < 0.5f ? 0.001f : -0.001f);
            dir = u[11] < 0.5f ? normalize(f3(u[0] - 0.5f, u[1] - 0.5f, u[2] - 0.5f)) : normalize(tgt - org);
        } else if (family == 1) {       // from the scene's extent towards the ball and its shell
            org = f3(sceneLo.x + u[4] * (sceneHi.x - sceneLo.x), sceneLo.y + u[5] * (sceneHi.y - sceneLo.y), sceneLo.z + u[6] * (sceneHi.z - sceneLo.z));
            dir = normalize(tgt - org);
        } else if (family == 2) {       // from close to the ball (2^-8 .. 2^2 of its radius away from the target), towards it or away
            const float dist = __builtin_exp2f(u[3] * 10.0f - 8.0f) * R;
            const F3 od = normalize(f3(u[4] - 0.5f, u[5] - 0.5f, u[6] - 0.5f));
            org = tgt + od * dist;
            dir = normalize(tgt - org);
            if (u[7] > 0.8f) dir = -dir;
        } else {                        // axis-parallel and plane-parallel directions (exact zeros)
            const float dist = __builtin_exp2f(u[3] * 10.0f - 8.0f) * R;
            const int ax = (int)(u[7] * 3.0f) % 3;
            F3 od = f3(ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f);
            if (u[11] < 0.5f) od = normalize(f3(ax == 0 ? 0.0f : u[4] - 0.5f, ax == 1 ? 0.0f : u[5] - 0.5f, ax == 2 ? 0.0f : u[6] - 0.5f));
            org = tgt + od * dist;
            dir = -od;
            if (u[8] > 0.8f) dir = od;
        }
        dir = dir * (u[9] < 0.7f ? 1.0f : __builtin_exp2f(4.0f * u[8] - 2.0f));
        const float l1 = (__builtin_fabsf(org.x) + __builtin_fabsf(org.y)) + __builtin_fabsf(org.z);
        if (!(l1 <= omax)) continue;
        const F3 dhat = unitDirectionScaled(dir, dot(dir, dir), sdir);
        for (int g = 0; g < ngroups; ++g) {
            const SphereCull E = groups[g];
            const float xe = sphereHalfLineExcessScaled(f3(E.centre[0], E.centre[1], E.centre[2]), org, dhat);
            if (!(E.cullR2 < xe)) continue;                     // (the kernel's own comparison: candidate = !(cullR2 < x))
            ++nc;
            for (int si = g * kSphGroupSize; si < (g + 1) * kSphGroupSize && si < nspheres; ++si) {
                F3 P, N;
                bool o;
                if (sphereIntersectionTest(spheres[si], org, dir, P, N, o) != -1.0f) ++nv;
            }
        }
    }
    if (nc) atomicAdd(certified, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// Soundness sweep of the camera-ray culling (GeomDev::rect, KParams::sceneRect, the per-row lists with their hull spans): every
// pixel of the frame sends `samples` camera rays (the render kernel's own cameraRayAt: iterations 1 .. samples of the pixel's
// depth-0 stream) through the FULL test of EVERY primitive, exactly the instantiations the camera-ray bounce runs.  A hit from a
// pixel the culling would have skipped -- outside the scene rectangle, outside the primitive's rectangle, or outside its span in
// the row's list -- is a VIOLATION (must be 0); `culled` counts the (ray, primitive) pairs the culling skips, `hits` the hits.
__global__ void k_sweep_camera_cull(KParams prm, const GeomDev *geoms, const int *rowOff, const int *rowIdx, int samples,
                                    unsigned long long *hits, unsigned long long *culled, unsigned long long *violations) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= prm.W * prm.H) return;
    const int y = j / prm.W, x = j - y * prm.W;
    const bool inScene = x >= prm.sceneRect[0] && x <= prm.sceneRect[2] && y >= prm.sceneRect[1] && y <= prm.sceneRect[3];
    unsigned int nh = 0, nc = 0, nv = 0;
    for (int g = 0; g < prm.ngeoms; ++g) {
        const GeomDev &G = geoms[g];
        bool reach = inScene && x >= G.rect[0] && x <= G.rect[2] && y >= G.rect[1] && y <= G.rect[3];
        if (reach && rowOff) {            // ... and listed for this row with a span that holds the pixel
            bool listed = false;
            for (int e = rowOff[y]; e < rowOff[y + 1]; ++e)
                if (rowIdx[2 * e] == g) listed = x >= (rowIdx[2 * e + 1] & 0xffff) && x <= (rowIdx[2 * e + 1] >> 16);
            reach = listed;
        }
        for (int sI = 1; sI <= samples; ++sI) {
            F3 org, dir, P, N;
            bool o;
            cameraRayAt(prm, iterationHash(sI, 0), j, x, y, org, dir);
            const float t = (G.flags & 1) ? boxIntersectionTest<true, true>(G, org, dir, P, N, o) : sphereIntersectionTest<true>(G, org, dir, P, N, o);
            nh += t > 0.0f ? 1u : 0u;
            nc += reach ? 0u : 1u;
            nv += (!reach && t > 0.0f) ? 1u : 0u;
        }
    }
    if (nh) atomicAdd(hits, (unsigned long long)nh);
    if (nc) atomicAdd(culled, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
}

// How much of the culling tables' object-space inflation (pt_api.hip: inflated_object_box) do the reference's hits actually NEED?  For
// every camera ray the full test reports as a hit, the ray as the test received it (fp32 origin and direction) is taken through the
// primitive's inverse transform in DOUBLE precision -- the exact line the fp32 test approximates -- and the smallest fraction s of the
// inflation is found at which that half-line meets the primitive grown by s x the inflation: s = 0 for a hit that is a geometric
// hit, 0 < s <= 1 for one the rounding of the fp32 test created (the hits the inflation exists for), s > 1 would be a hit outside the
// inflated box.  `infl`: per primitive {dx, dy, dz, active} for a cube (growth of the half extents) and {r_inflated, -, -, active}
// for a sphere; primitives whose culling is switched off (a corner not in front of the eye) are skipped.  Outputs: the largest s
// (bit pattern of a non-negative double: ordered like an unsigned integer) and the number of hits with s > 0.
__device__ __forceinline__ bool halfLineMeetsBoxD(const double ro[3], const double rd[3], const double h[3]) {
    double t0 = 0.0, t1 = 1e300;
    for (int a = 0; a < 3; ++a) {
        if (rd[a] == 0.0) {
            if (ro[a] < -h[a] || ro[a] > h[a]) return false;
            continue;
        }
        double ta = (-h[a] - ro[a]) / rd[a], tb = (h[a] - ro[a]) / rd[a];
        if (ta > tb) { const double q = ta; ta = tb; tb = q; }
        t0 = ta > t0 ? ta : t0;
        t1 = tb < t1 ? tb : t1;
    }
    return t0 <= t1;
}
__global__ void k_sweep_camera_cull_margin(KParams prm, const GeomDev *geoms, const double *infl, int samples, unsigned long long *worstBits,
                                           unsigned long long *needed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= prm.W * prm.H) return;
    const int y = j / prm.W, x = j - y * prm.W;
    double worst = 0.0;
    unsigned int nn = 0;
    for (int g = 0; g < prm.ngeoms; ++g) {
        if (infl[4 * g + 3] == 0.0) continue;
        const GeomDev &G = geoms[g];
        for (int sI = 1; sI <= samples; ++sI) {
            F3 org, dir, P, N;
            bool o;
            cameraRayAt(prm, iterationHash(sI, 0), j, x, y, org, dir);
            const float t = (G.flags & 1) ? boxIntersectionTest<true, true>(G, org, dir, P, N, o) : sphereIntersectionTest<true>(G, org, dir, P, N, o);
            if (!(t > 0.0f)) continue;
            const double ow[3] = {org.x, org.y, org.z}, dw[3] = {dir.x, dir.y, dir.z};
            double ro[3], rd[3];
            for (int r = 0; r < 3; ++r) {
                ro[r] = (double)G.inv[r] * ow[0] + (double)G.inv[3 + r] * ow[1] + (double)G.inv[6 + r] * ow[2] + (double)G.inv[9 + r];
                rd[r] = (double)G.inv[r] * dw[0] + (double)G.inv[3 + r] * dw[1] + (double)G.inv[6 + r] * dw[2];
            }
            double s = 0.0;
            if (G.flags & 1) {
                double h[3] = {0.5, 0.5, 0.5};
                if (!halfLineMeetsBoxD(ro, rd, h)) {
                    double lo = 0.0, hi = 1.0;
                    for (int a = 0; a < 3; ++a) h[a] = 0.5 + infl[4 * g + a];
                    if (!halfLineMeetsBoxD(ro, rd, h)) {
                        s = 2.0;                              // (outside the inflated box: reported as 2)
                    } else {
                        for (int it = 0; it < 30; ++it) {
                            const double mid = 0.5 * (lo + hi);
                            for (int a = 0; a < 3; ++a) h[a] = 0.5 + mid * infl[4 * g + a];
                            if (halfLineMeetsBoxD(ro, rd, h)) hi = mid; else lo = mid;
                        }
                        s = hi;
                    }
                }
            } else {
                const double a = rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2], b = ro[0] * rd[0] + ro[1] * rd[1] + ro[2] * rd[2];
                double ts = a > 0.0 ? -b / a : 0.0;
                ts = ts > 0.0 ? ts : 0.0;
                const double qx = ro[0] + ts * rd[0], qy = ro[1] + ts * rd[1], qz = ro[2] + ts * rd[2];
                const double dist = sqrt(qx * qx + qy * qy + qz * qz);
                if (dist > 0.5) s = (dist - 0.5) / (infl[4 * g] - 0.5);
            }
            if (s > 0.0) ++nn;
            worst = s > worst ? s : worst;
        }
    }
    if (nn) atomicAdd(needed, (unsigned long long)nn);
    if (worst > 0.0) atomicMax(worstBits, (unsigned long long)__double_as_longlong(worst));
}

// wallPlanesPossible soundness sweep: pseudo-random rays against the walls of a scene as pt_init numbers them (`wallGeoms[w]` = the
// cube that is wall w).  Origins as the render kernel meets them and worse: on a wall's inner face pushed 1e-3 (or 0 .. 4e-3) into
// the room, in the corners where two and three walls meet, anywhere inside the box around the walls, outside it; directions random,
// aimed at the shell of a wall's box (grazes), axis-parallel, with exact zeros.  A wall the planes certify as missed that the full
// test hits is a VIOLATION.  `certified` counts (ray, wall) certificates, `single` the rays left with exactly one possible wall.
__global__ void k_sweep_wall_planes(KParams prm, const GeomDev *wallGeoms, const WallBox *walls, unsigned long long seed, int per_thread,
                                    unsigned long long *certified, unsigned long long *violations, unsigned long long *single) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nc = 0, nv = 0, ns = 0;
    const int nPlane = prm.nSlotWalls + prm.nPlaneWalls;      // (walls certified by an axis slot's plane, then by the plane of a rotated cube's inner face)
    const F3 olo = f3(prm.outerLo[0], prm.outerLo[1], prm.outerLo[2]), ohi = f3(prm.outerHi[0], prm.outerHi[1], prm.outerHi[2]);
    const F3 oc = (olo + ohi) * 0.5f, oh = (ohi - olo) * 0.5f;
    for (int k = 0; k < per_thread; ++k) {
        float u[12];
        for (int j = 0; j < 12; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const int wi = (int)(u[0] * (float)prm.nWalls) % prm.nWalls;
        const WallBox W = walls[wi];
        const F3 wlo = f3(W.lo[0], W.lo[1], W.lo[2]), whi = f3(W.hi[0], W.hi[1], W.hi[2]);
        const F3 wc = (wlo + whi) * 0.5f, wh = (whi - wlo) * 0.5f;
        F3 org;
        if (u[1] < 0.55f) {
            // on wall wi's box, pulled towards the middle of the room by 0 .. 4e-3 (1e-3 in half of the cases: the scatter offset)
            org = wc + f3((2 * u[2] - 1) * wh.x, (2 * u[3] - 1) * wh.y, (2 * u[4] - 1) * wh.z);
            const float off = u[5] < 0.5f ? 1e-3f : u[5] * 8e-3f - 4e-3f;
            // towards the room's centre along the wall's thinnest axis, from the face on that side
            const int a = wh.x <= wh.y && wh.x <= wh.z ? 0 : (wh.y <= wh.z ? 1 : 2);
            const float ca = a == 0 ? oc.x : (a == 1 ? oc.y : oc.z), wa = a == 0 ? wc.x : (a == 1 ? wc.y : wc.z), ha = a == 0 ? wh.x : (a == 1 ? wh.y : wh.z);
            const float sgn = ca >= wa ? 1.0f : -1.0f;
            const float pos = wa + sgn * (ha * 0.9999f + off);      // (0.9999: the real face lies inside the inflated box)
            if (a == 0) org.x = pos; else if (a == 1) org.y = pos; else org.z = pos;
            if (u[6] < 0.3f) {                                      // ... into a corner: clamp another coordinate to the room's border
                const float m = u[6] < 0.15f ? 1.0f : -1.0f;
                if (a != 0 && u[7] < 0.5f) org.x = oc.x + m * oh.x * (1.0f - 2e-3f * u[8]);
                else if (a != 1) org.y = oc.y + m * oh.y * (1.0f - 2e-3f * u[8]);
                else org.z = oc.z + m * oh.z * (1.0f - 2e-3f * u[8]);
            }
            if (wi >= prm.nSlotWalls && wi < nPlane && u[7] < 0.6f) {
                // a ROTATED wall: on the cube's own face that looks at the room's middle, 0 .. 4e-3 off it (1e-3 in half of the cases)
                const GeomDev &G = wallGeoms[wi];
                const F3 c0 = f3(G.xf[0], G.xf[1], G.xf[2]), c1 = f3(G.xf[3], G.xf[4], G.xf[5]), c2 = f3(G.xf[6], G.xf[7], G.xf[8]);
                const float l0 = dot(c0, c0), l1 = dot(c1, c1), l2 = dot(c2, c2);
                const int ta = l0 <= l1 && l0 <= l2 ? 0 : (l1 <= l2 ? 1 : 2);
                const F3 ca = ta == 0 ? c0 : (ta == 1 ? c1 : c2);
                const float len = __builtin_sqrtf(ta == 0 ? l0 : (ta == 1 ? l1 : l2));
                const F3 centre = f3(G.xf[9], G.xf[10], G.xf[11]);
                const float sgn = dot(ca, oc - centre) >= 0.0f ? 1.0f : -1.0f;
                F3 po = f3((u[2] - 0.5f) * 0.999f, (u[3] - 0.5f) * 0.999f, (u[4] - 0.5f) * 0.999f);
                const float h = sgn * (0.5f + off / len);
                if (ta == 0) po.x = h; else if (ta == 1) po.y = h; else po.z = h;
                org = mulMV(G.xf, po, 1.0f);
            }
        } else if (u[1] < 0.85f) {
            org = oc + f3((2 * u[2] - 1) * oh.x, (2 * u[3] - 1) * oh.y, (2 * u[4] - 1) * oh.z);
        } else {
            org = oc + f3((2 * u[2] - 1) * oh.x, (2 * u[3] - 1) * oh.y, (2 * u[4] - 1) * oh.z) * (1.0f + 3.0f * u[5]);
        }
        F3 dir;
        if (u[9] < 0.5f) {
            dir = normalize(f3(u[10] - 0.5f, u[11] - 0.5f, u[8] - 0.5f));
        } else {                                                    // aimed at the shell of a wall's box: hits, grazes, near misses
            const int wj = (int)(u[8] * (float)prm.nWalls) % prm.nWalls;
            const WallBox V = walls[wj];
            const F3 vc = f3(V.lo[0] + V.hi[0], V.lo[1] + V.hi[1], V.lo[2] + V.hi[2]) * 0.5f, vh = f3(V.hi[0] - V.lo[0], V.hi[1] - V.lo[1], V.hi[2] - V.lo[2]) * 0.5f;
            const float shell = 0.95f + 0.1f * u[7];
            dir = normalize(vc + f3((2 * u[10] - 1) * vh.x, (2 * u[11] - 1) * vh.y, (2 * u[6] - 1) * vh.z) * shell - org);
        }
        if (u[9] > 0.92f) dir = f3(u[9] < 0.95f ? 1.0f : 0.0f, (u[9] >= 0.95f && u[9] < 0.98f) ? -1.0f : 0.0f, u[9] >= 0.98f ? 1.0f : 0.0f);   // axis-parallel
        else if (u[9] > 0.85f) dir = normalize(f3(dir.x, 0.0f, dir.z));                                      // an exact zero
        const float l1 = (__builtin_fabsf(org.x) + __builtin_fabsf(org.y)) + __builtin_fabsf(org.z);
        if (!(l1 <= prm.wallOMax)) continue;
        const F3 inv = f3(__builtin_amdgcn_rcpf(dir.x), __builtin_amdgcn_rcpf(dir.y), __builtin_amdgcn_rcpf(dir.z));
        uint32_t possible = wallPlanesPossible(prm, org, dir, inv);
        if (prm.nPlaneWalls > 0) possible |= wallPlanesOriented(prm, org, dir, inv, prm.nSlotWalls, prm.nPlaneWalls);
        for (int w = 0; w < nPlane; ++w)
            if (!((possible >> w) & 1u)) {
                ++nc;
                F3 P, N;
                bool o;
                if (boxIntersectionTest<false>(wallGeoms[w], org, dir, P, N, o) != -1.0f) ++nv;
            }
        ns += __popc(possible) == 1 ? 1u : 0u;
    }
    if (nc) atomicAdd(certified, (unsigned long long)nc);
    if (nv) atomicAdd(violations, (unsigned long long)nv);
    if (ns) atomicAdd(single, (unsigned long long)ns);
}

// boxIntersectionTest with the fast slab phase (boxSlabsFast, falling back on the reference's loop) next to the test with the
// reference's loop alone, on rays dense in what the fast path has to be careful about: edges and corners of the cube (the slabs'
// parameters within the margin of each other), grazes, origins on, just off and inside the surface, directions with tiny, zero,
// infinite and NaN components, axis-parallel ones.  cnt[0] = rays, [1] = rays the fast path decided, [2] = hits among those,
// [3] = bit mismatches (t, and on a hit P, nsrc, outside; NaN == NaN) -- must be 0.
__global__ void k_sweep_box_fast(const GeomDev *geoms, int ngeoms, unsigned long long seed, int per_thread, unsigned long long *cnt, float *dump) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int nr = 0, nf = 0, nh = 0, bad = 0;
    for (int k = 0; k < per_thread; ++k) {
        float u[16];
        for (int j = 0; j < 16; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u[j] = (float)(x >> 40) * (1.0f / 16777216.0f);
        }
        const GeomDev &G = geoms[(int)(u[0] * (float)ngeoms) % ngeoms];
        // a point of the cube in object space: anywhere in it, on a face, on an edge, at a corner (coordinates snapped to +-.5)
        auto snap = [](float v, float w) { return w < 0.5f ? v - 0.5f : (w < 0.75f ? -0.5f : 0.5f); };
        const F3 onCube = f3(snap(u[1], u[4]), snap(u[2], u[5]), snap(u[3], u[6] * 0.75f));
        // the origin: around the cube (up to 3 of its sizes away), a multiple of it (far), on / just off its surface, inside
        F3 oobj;
        const int ok = (int)(u[7] * 8.0f);
        if (ok < 3) oobj = f3(6 * u[8] - 3, 6 * u[9] - 3, 6 * u[10] - 3);
        else if (ok == 3) oobj = f3(600 * u[8] - 300, 600 * u[9] - 300, 600 * u[10] - 300);
        else if (ok == 4) oobj = f3(u[8] - 0.5f, u[9] - 0.5f, u[10] - 0.5f);
        else {
            const F3 q = f3(snap(u[8], u[11]), snap(u[9], 0.2f), snap(u[10], 0.2f));
            const float off = ok == 5 ? 0.0f : (ok == 6 ? 1e-3f : 1e-6f) * (u[12] < 0.5f ? 1.0f : -1.0f);
            oobj = f3(q.x * (1.0f + off), q.y, q.z);
            if (u[12] > 0.66f) oobj = f3(q.y, q.x * (1.0f + off), q.z); else if (u[12] > 0.33f) oobj = f3(q.z, q.y, q.x * (1.0f + off));
        }
        F3 org = mulMV(G.xf, oobj, 1.0f);
        // the direction: at the point of the cube (hits through faces, edges, corners), nudged by a few ulps, or anywhere
        const F3 target = mulMV(G.xf, onCube, 1.0f);
        F3 dir = target - org;
        const int dk = (int)(u[13] * 16.0f);
        if (dk < 3) dir = f3(u[1] - 0.5f, u[2] - 0.5f, u[3] - 0.5f);
        else if (dk < 6) dir = dir + f3(u[14] - 0.5f, u[15] - 0.5f, u[11] - 0.5f) * (1e-6f * (__builtin_fabsf(dir.x) + __builtin_fabsf(dir.y) + __builtin_fabsf(dir.z)));
        else if (dk == 6) dir = mulMV(G.xf, f3(u[14] < 0.5f ? 1.0f : -1.0f, 0, 0), 0.0f);       // along an object axis
        else if (dk == 7) dir = mulMV(G.xf, f3(0, u[14] - 0.5f, u[15] - 0.5f), 0.0f);            // in an object plane
        else if (dk == 8) dir = f3(dir.x, 0.0f, dir.z);
        else if (dk == 9) dir = f3(dir.x, dir.y * 1e-13f, dir.z * (u[14] < 0.5f ? 1e-20f : 1.0f));
        dir = normalize(dir);
        if (dk == 10) {                                  // raw bit patterns now and then: NaN, inf, denormals, huge
            const uint32_t b = (uint32_t)x;
            if (u[14] < 0.3f) dir.x = __uint_as_float(b); else if (u[14] < 0.6f) org.y = __uint_as_float(b); else dir = dir * __uint_as_float(b & 0x7fffffffu);
        }
        F3 P1 = f3(1, 2, 3), N1 = f3(4, 5, 6), P2 = P1, N2 = N1;
        bool o1 = false, o2 = false;
        const float t1 = boxIntersectionTest<false, false, false>(G, org, dir, P1, N1, o1);
        const float t2 = boxIntersectionTest<false, false, true>(G, org, dir, P2, N2, o2);
        auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); };
        bool eq = same(t1, t2) && o1 == o2 && same(P1.x, P2.x) && same(P1.y, P2.y) && same(P1.z, P2.z) && same(N1.x, N2.x) && same(N1.y, N2.y) && same(N1.z, N2.z);
        // (the early-miss variant as well: it returns before the slab phase or not at all)
        F3 P3 = f3(1, 2, 3), N3 = f3(4, 5, 6);
        bool o3 = false;
        const float t3 = boxIntersectionTest<true, false, false>(G, org, dir, P3, N3, o3);
        // (its argument -- both quotients of an axis the ray leaves are <= -0 -- takes finite products: a direction of 1e38 makes the
        // transform overflow, inf - inf = NaN in the reference's loop and a plain miss here; the renderer's directions are unit vectors)
        const bool tame = __builtin_fabsf(dir.x) + __builtin_fabsf(dir.y) + __builtin_fabsf(dir.z) < 1e30f;
        if (tame) eq = eq && same(t3, t2) && o3 == o2 && same(P3.x, P2.x) && same(P3.y, P2.y) && same(P3.z, P2.z) && same(N3.x, N2.x);
        bad += eq ? 0u : 1u;
        if (!eq && dump) {                               // (the first few mismatching rays, for the host to print: PT_AMD_VERBOSE)
            const unsigned long long slot = atomicAdd(&cnt[6], 1ull);
            if (slot < 8) {
                float *r = dump + 24 * slot;
                r[0] = (float)(&G - geoms); r[1] = org.x; r[2] = org.y; r[3] = org.z; r[4] = dir.x; r[5] = dir.y; r[6] = dir.z;
                r[7] = t1; r[8] = t2; r[9] = t3; r[10] = o1; r[11] = o2; r[12] = P1.x; r[13] = P1.y; r[14] = P1.z; r[15] = P2.x; r[16] = P2.y; r[17] = P2.z;
                r[18] = N1.x; r[19] = N1.y; r[20] = N1.z; r[21] = N2.x; r[22] = N2.y; r[23] = N2.z;
            }
        }
        ++nr;
        {   // what the fast path says about this ray
            const F3 qo = mulMV(G.inv, org, 1.0f), qdu = mulMV0(G.inv, G.invZ, dir);
            F3 qd;
            bool hit, outs;
            int axis;
            if (boxSlabsFastDecide(qo, qdu, qd, hit, outs, axis)) { ++nf; nh += hit ? 1u : 0u; }
        }
    }
    atomicAdd(&cnt[0], (unsigned long long)nr);
    if (nf) atomicAdd(&cnt[1], (unsigned long long)nf);
    if (nh) atomicAdd(&cnt[2], (unsigned long long)nh);
    if (bad) atomicAdd(&cnt[3], (unsigned long long)bad);
}
// divUnscaled(1, s) = 1.0f / s on EVERY float of [2^-40, 2^40] (the fast normalize's reciprocal), and divUnscaled(a, d) = a / d on
// pseudo-random pairs of the box test's range (|a| <= 2^20 + 1, 2^-40 <= |d| <= 2): bad[0] / bad[1] count bit mismatches
__global__ void k_sweep_div_unscaled(unsigned long long seed, int per_thread, unsigned long long *bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    unsigned int b0 = 0, b1 = 0;
    for (unsigned long long bits = 0x2b800000ull + tid; bits <= 0x53800000ull; bits += stride) {
        const float s = __uint_as_float((uint32_t)bits);
        const float q = divUnscaled(1.0f, s, __builtin_amdgcn_rcpf(s)), r = 1.0f / s;
        b0 += __float_as_uint(q) == __float_as_uint(r) ? 0u : 1u;
    }
    unsigned long long x = seed + tid * 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < per_thread; ++k) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint32_t ab = (uint32_t)x, db = (uint32_t)(x >> 32);
        // a: sign, exponent 2^-25 .. 2^20, any mantissa (or exactly 0); d: sign, exponent 2^-40 .. 2^0, any mantissa
        float a = __uint_as_float((ab & 0x807fffffu) | ((uint32_t)(127 - 25 + (ab >> 23 & 63) % 46) << 23));
        if ((k & 63) == 7) a = 0.0f;
        const float d = __uint_as_float((db & 0x807fffffu) | ((uint32_t)(127 - 40 + (db >> 23 & 63) % 41) << 23));
        const float q = divUnscaled(a, d, __builtin_amdgcn_rcpf(d)), r = a / d;
        b1 += __float_as_uint(q) == __float_as_uint(r) ? 0u : 1u;
    }
    if (b0) atomicAdd(&bad[0], (unsigned long long)b0);
    if (b1) atomicAdd(&bad[1], (unsigned long long)b1);
}
// slabQuotients vs the compiler's correctly rounded division; counts mismatching lanes
__global__ void k_test_slab_quotients(const float *o, const float *d, int n, float *t1, float *t2, float *r1, float *r2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    slabQuotients(o[i], d[i], t1[i], t2[i]);
    r1[i] = (-0.5f - o[i]) / d[i];
    r2[i] = (+0.5f - o[i]) / d[i];
}
// pseudo-random sweep entirely on the device: returns the number of bit mismatches (NaN == NaN)
__global__ void k_sweep_slab_quotients(unsigned long long seed, int per_thread, unsigned long long *mismatches) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned int bad = 0;
    for (int k = 0; k < per_thread; ++k) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;                     // xorshift64
        const uint32_t ob = (uint32_t)x, db = (uint32_t)(x >> 32);
        float o, d;
        if ((k & 3) == 0) {            // raw bit patterns: every exponent, denormals, inf, NaN
            o = __uint_as_float(ob); d = __uint_as_float(db);
        } else {                       // the range the tracer lives in: |o| < ~4000, |d| <= 1
            o = ((int)(ob >> 8) - (1 << 23)) * (1.0f / 2048.0f) * ((k & 4) ? 1.0f : 1e-3f);
            d = __uint_as_float((db & 0x807fffffu) | ((uint32_t)(127 - (db >> 23 & 31)) << 23));
            if ((k & 15) == 5) o = (ob & 1) ? 0.5f : -0.5f;           // numerator exactly +0
        }
        float t1, t2;
        slabQuotients(o, d, t1, t2);
        const float r1 = (-0.5f - o) / d, r2 = (+0.5f - o) / d;
        const bool e1 = __float_as_uint(t1) == __float_as_uint(r1) || (t1 != t1 && r1 != r1);
        const bool e2 = __float_as_uint(t2) == __float_as_uint(r2) || (t2 != t2 && r2 != r2);
        bad += (e1 ? 0u : 1u) + (e2 ? 0u : 1u);
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
// sqrtUnscaled next to the compiler's correctly rounded sqrt on EVERY fp32 bit pattern of its range, and
// inverseSqrtNearOne next to 1.0f / sqrtf on every bit pattern at all (thread t checks patterns t, t + stride, ...):
// bad[0] / bad[2] count bit mismatches (NaN == NaN), bad[1] the patterns inside sqrtUnscaled's range, bad[3] the
// patterns that took inverseSqrtNearOne's short path.
__global__ void k_sweep_unscaled_sqrt(unsigned long long *bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    unsigned int b0 = 0, n = 0, b2 = 0, n3 = 0;
    for (unsigned long long bits = tid; bits < (1ull << 32); bits += stride) {
        const float x = __uint_as_float((uint32_t)bits);
        const float r = __builtin_sqrtf(x);
        const float q = inverseSqrtNearOne(x), qr = 1.0f / r;
        b2 += (__float_as_uint(q) == __float_as_uint(qr) || (q != q && qr != qr)) ? 0u : 1u;
        n3 += (unsigned)((int)(uint32_t)bits - 0x3f800000 + 256) <= 512u ? 1u : 0u;
        if (!(x == 0.0f || (x >= 0x1p-96f && x < __builtin_inff()))) continue;
        const float a = sqrtUnscaled(x);
        b0 += (__float_as_uint(a) == __float_as_uint(r) || (a != a && r != r)) ? 0u : 1u;
        ++n;
    }
    if (b0) atomicAdd(&bad[0], (unsigned long long)b0);
    atomicAdd(&bad[1], (unsigned long long)n);
    if (b2) atomicAdd(&bad[2], (unsigned long long)b2);
    if (n3) atomicAdd(&bad[3], (unsigned long long)n3);
}
__global__ void k_test_hemisphere(const float *nrm, const int *iid, int n, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r = makeSeededRandomEngine(iid[3 * i], iid[3 * i + 1], iid[3 * i + 2]);
    F3 d = calculateRandomDirectionInHemisphere(f3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), r);
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
}
__global__ void k_test_sincos(const float *x, int n, float *s, float *c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincosPoly(x[i], s[i], c[i]);
}
__global__ void k_test_reflect_refract(const float *I, const float *N, const float *eta, int n, float *rl, float *rr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 a = f3(I[3 * i], I[3 * i + 1], I[3 * i + 2]), b = f3(N[3 * i], N[3 * i + 1], N[3 * i + 2]);
    F3 r1 = reflect(a, b), r2 = refract(a, b, eta[i]);
    rl[3 * i] = r1.x; rl[3 * i + 1] = r1.y; rl[3 * i + 2] = r1.z;
    rr[3 * i] = r2.x; rr[3 * i + 1] = r2.y; rr[3 * i + 2] = r2.z;
}

}  // namespace ptk
