/*
 * pt_amd_test.h -- TEST-ONLY entry points of the MI355X path-tracing hot path (libpt_amd_test.so).
 *
 * Not part of the drop-in boundary: the product library libpt_amd.so (include/pt_amd.h) exports none of these.  The same
 * single source (csrc/pt_api.hip) is linked a second time with -DPT_TEST_API into libpt_amd_test.so, which adds the entry
 * points below -- the __device__ functions of the render kernels evaluated one by one over host arrays (parity tests of
 * SURVEY 8 rows a6-a12, a19), the device-side soundness sweeps of every culling shortcut, and a hook that sets the
 * renderer's fault word by hand.  Only tests/ and profiles/ load that library.
 */
#ifndef PT_AMD_TEST_H
#define PT_AMD_TEST_H

#include "pt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostics of a renderer: the instance that lives in THIS library (initialise it with this library's pt_init) ----
 * State of the paths still alive after `bounces` bounces of iteration `iter`, sorted by pixel index (the device queue order is
 * arrival order): arrays of capacity W*H (x3).  Does not touch the accumulator.  (Rounds 1-4 exported it from the product.) */
int pt_debug_trace_paths(int iter, int bounces, float *origin3, float *dir3, float *color3,
                         int32_t *pixelIndex, int32_t *count);

/* ---- device primitives evaluated on the GPU over HOST arrays (parity tests of rows a6-a12, a19):
 * the same __device__ functions the render kernels call. ------------------------------------------ */
int pt_test_utilhash(const uint32_t *in, uint32_t *out, int n);
int pt_test_rng(const uint32_t *seeds, int nseeds, int ndraws, float *u01_out /* nseeds*ndraws */);
/* rays: n x 6; each ray against ONE geom (geoms[geom_index[i]]); outputs keep their input values
 * on a miss like the reference's out-parameters (intersections.h:86-88,114-126). */
int pt_test_intersect(const PtGeom *geoms, int ngeoms, const int32_t *geom_index, const float *rays,
                      int n, float *t, float *p3, float *n3, int32_t *outside);
/* sphereCertainMiss (world-space culling of spheres) soundness: `rays` pseudo-random rays against the spheres
 * of `geoms`; *violations = rays culled although the full test hits (must be 0), *culled = rays culled. */
int pt_test_sphere_cull_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled,
                              uint64_t *violations);
/* sphereHalfLineExcess (pt_device.h: the certificate of the sphere-heavy sweep of the later bounces -- the squared distance of a sphere's
 * centre from the HALF-line, direction normalised approximately) swept like certainMiss: `rays` rays in three families (aimed near the
 * ball from 1/64 .. 64 units; leaving the sphere's own surface as a scatter does; from within 2 % of the bounding ball's surface),
 * directions unit and not.  *culled = certified misses, *behind = those with the centre behind the origin, *violations = certified
 * although src/intersections.h:101-143 returns a hit (must be 0). */
int pt_test_sphere_halfline_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled, uint64_t *behind,
                                  uint64_t *violations);
/* The sphere CLUSTERS' box certificates (sphere-heavy scenes without meshes: a survivor's class bits 3 / 4 say which of the two clusters of
 * spheres its ray can hit; pt_api.hip: build_sphere_clusters builds them for `geoms` -- a whole scene, spheres and cubes -- exactly as
 * pt_init does) soundness: `rays` rays in four families (scatters off the spheres; from the scene's extent towards a cluster's box and
 * its shell; from close to a box; axis- and plane-parallel directions).  certified2[g] = certificates issued for cluster g, *violations
 * = spheres of a cluster certified as missed that src/intersections.h:101-143 hits (must be 0).  info18 (may be null): the bound on
 * the origins, the size of cluster 0 in the table, the two boxes {lo, hi, -, -}. */
int pt_test_sphere_cluster_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *certified2, uint64_t *violations,
                                 float *info18);
/* ... and of the sphere GROUPS' bounding balls (k_bounce<..., GROUPS>: scenes of hundreds of swept primitives, pt_init: build_sphere_groups): a group
 * certified as missed sends the ray through the full test of each of its 16 members; a hit is a violation (must be 0) */
int pt_test_sphere_group_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *certified, uint64_t *violations, int32_t *ngroups);
/* ... and the clusters themselves for inspection (host only, no GPU needed): info18 as above; table[k] = the primitive behind entry k of the
 * sweep's table (cluster 0 = entries 0 .. info18[1] - 1, padded to an even count with a copy of its last sphere; then cluster 1 -- whose end pt_init
 * pads likewise), *ntable entries (<= table_cap). */
int pt_test_sphere_clusters(const PtGeom *geoms, int ngeoms, float *info18, int32_t *table, int32_t table_cap, int32_t *ntable);
/* wallCertainMiss (world-space culling of large cubes against their inflated bounding boxes, which classes the queue by
 * the wall a path can still hit) soundness: as above, for the cubes of `geoms`. */
int pt_test_wall_box_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *culled,
                           uint64_t *violations);
/* wallPlanesPossible / wallPlanesOriented (the one-plane-per-wall certificates of the survivors' queue classes; pt_init's choose_walls numbers the cubes of
 * `geoms` as walls exactly as a render would) soundness: `rays` pseudo-random rays from the walls' inner faces, the corners, the
 * interior and beyond; *violations = (ray, wall) pairs certified as missed although the full test hits (must be 0), *certified =
 * certificates issued, *single = rays left with exactly one possible wall, *nplane = walls that have a plane. */
int pt_test_wall_plane_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, int32_t *nplane, uint64_t *certified,
                             uint64_t *violations, uint64_t *single);
/* (host only, no GPU needed) the planes pt_init's choose_walls keeps for ROTATED walls (ptd::wallPlanesOriented, round 5): planes[w] =
 * {unit normal towards the scene's interior (3), threshold, far} of plane wall w, wall_geom[i] = the primitive that is wall i (walls with
 * an axis slot first, then the plane walls, then the rest), *nslot / *nplane / *nwalls their counts.  planes: 6 x 5 floats, wall_geom: 6 ints. */
int pt_test_wall_planes(const PtGeom *geoms, int ngeoms, float *planes, int32_t *wall_geom, int32_t *nslot, int32_t *nplane, int32_t *nwalls);
/* Screen-space culling of camera rays (per-primitive pixel rectangles, their union, the per-row primitive lists with their hull
 * spans -- everything pt_init derives from the camera, built here by the very same host functions) soundness: every pixel of
 * `cam`'s frame sends `samples` camera rays through the full tests of every primitive; *violations = hits from a pixel the culling
 * would have skipped (must be 0), *culled = (ray, primitive) pairs the culling skips, *hits = hits.  Spheres and cubes. */
int pt_test_camera_cull_sweep(const PtCamera *cam, const PtGeom *geoms, int ngeoms, int samples, uint64_t *hits, uint64_t *culled,
                              uint64_t *violations);
/* ... and the MARGIN of the tables' object-space inflation (pt_api.hip: inflated_object_box): for every hit of the same sweep, the
 * fraction of the inflation the hit needs -- 0 for a geometric hit, up to 1 for a hit the fp32 test's rounding created, above 1 for a
 * hit outside the inflated box -- evaluated in double precision on the exact object-space half-line.  *worst_fraction = the largest
 * (its reciprocal is the safety margin), *needed = hits with a fraction above 0.  Primitives whose culling is off are skipped. */
int pt_test_camera_cull_margin(const PtCamera *cam, const PtGeom *geoms, int ngeoms, int samples, double *worst_fraction, uint64_t *needed);
/* The same tables for inspection (host only, no GPU needed): rects4[4 i ..] = primitive i's pixel rectangle, scene_rect4 = their
 * union, spans2[2 (y ngeoms + i) ..] = the pixels x0 .. x1 of row y from which primitive i is reachable (x0 > x1: none). */
int pt_test_camera_cull_tables(const PtCamera *cam, const PtGeom *geoms, int ngeoms, int32_t *rects4, int32_t *scene_rect4, int32_t *spans2);
/* Triangle meshes.  pt_test_mesh_intersect: `n` rays against ONE mesh geom on the GPU, through the hierarchy (flat = 0) or a
 * plain list of its triangles (flat = 1: the brute-force rule); outputs keep their input values on a miss; culled[i] = 1 when
 * the bounding-ball test (certainMiss) rejected the ray -- t[i] is NaN if the full test hits nevertheless (must not happen).
 * pt_test_mesh_bvh (host only, no GPU needed): the records pt_init would build, in units of 4 words (pt_device.h: MeshUnit) --
 * the triangles in file order, three units each (v0, v1, v2, the mesh's box margin as floats, two words unused), one unit of
 * padding when that is an odd number of units, then the inner nodes of the copy laid out for rays of direction octant `octant`
 * (bit a set: the direction's component a is negative), two units each, nearer child first: six half-precision planes in three
 * words (entry x, y, z, exit x, y, z; entry = lo where the direction is positive, hi where negative; lo rounded down, hi up)
 * and the child's ref, then the same for the far child; refs in units relative to this array, bit 31 = a triangle;
 * *nrecs in = capacity in units, out = unit count; *stack_need = far children that can wait at once on a lane's stack, over
 * all eight copies. */
int pt_test_mesh_intersect(const PtGeom *geom, const float *tris, int ntris, int flat, const float *rays, int n, float *t,
                           float *p3, float *n3, int32_t *outside, int32_t *culled);
int pt_test_mesh_bvh(const float *tris, int ntris, int octant, uint32_t *units4, int *nrecs, int *stack_need);
/* certainMiss soundness for a mesh geom: `rays` pseudo-random rays dense in grazes of its bounding ball (origins 1/64 .. 64
 * radii away); *violations = rays the bounding-ball test rejected although the walk hits (must be 0), *hits = rays that hit */
int pt_test_mesh_cull_sweep(const PtGeom *geom, const float *tris, int ntris, uint64_t seed, int64_t rays, uint64_t *culled,
                            uint64_t *violations, uint64_t *hits);
/* slabQuotients (shared-reciprocal packed division of the box test) next to the compiler's correctly
 * rounded `/`: per-element outputs, and a device-side pseudo-random sweep that returns the number of
 * bit mismatches over `pairs` (o, d) pairs (must be 0). */
int pt_test_slab_quotients(const float *o, const float *d, int n, float *t1, float *t2, float *ref1, float *ref2);
int pt_test_slab_quotients_sweep(uint64_t seed, int64_t pairs, uint64_t *mismatches);
/* The box test's fast slab phase (pt_device.h: boxSlabsFast -- comparisons on approximate quotients wherever their outcome is beyond
 * doubt, ONE exact quotient, the reference's loop as the fallback) against the reference's loop alone: `rays` pseudo-random rays per
 * call against the cubes of `geoms`, dense in edges, corners, grazes, surface origins, degenerate directions.  counts[0] = rays,
 * [1] = rays the fast path decided, [2] = hits among those, [3] = bit mismatches of (t, P, normal source, outside) -- must be 0.
 * div_mismatches[0]: the fast normalize's reciprocal against 1.0f / s on every float of [2^-40, 2^40]; [1]: the one exact quotient
 * against a / d on 2^30 pairs of the box test's range -- both must be 0. */
int pt_test_box_fast_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t counts[4], uint64_t div_mismatches[2]);
/* sqrtUnscaled (the correctly rounded sqrt without its exponent-range handling, used by the hemisphere sampler)
 * next to the compiler's sqrt on every fp32 bit pattern of its range (+-0 and [2^-96, inf)), and inverseSqrtNearOne
 * (the re-normalisation of getPointOnRay) next to 1.0f / sqrtf on every bit pattern: mismatches[0] and [2] must be 0,
 * [1] = patterns inside sqrtUnscaled's range, [3] = patterns on inverseSqrtNearOne's short path (513). */
int pt_test_unscaled_sqrt_sweep(uint64_t mismatches[4]);
/* sets the renderer's device fault word by hand (2; 0 clears it in every slot), so that the reporting path can be tested */
int pt_test_force_fault(int which);
/* the next `count` assemblies of the group issue their ncclReduce with a NULL communicator (ncclInvalidArgument): the error path of
 * pt_group_reduce / pt_group_iterate / pt_group_readback -- ncclGroupEnd must still run and a later call must succeed */
int pt_test_group_fail_next_reduce(PtGroup *g, int count);
int pt_test_hemisphere(const float *normals3, const int32_t *iter_index_depth3, int n, float *out3);
int pt_test_sincos(const float *x, int n, float *s, float *c);
int pt_test_pow(const float *x, const float *e, int n, float *out);   /* build-defined x^e of the imperfect-specular sampler */
int pt_test_reflect_refract(const float *I3, const float *N3, const float *eta, int n, float *refl3,
                            float *refr3);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_TEST_H */
