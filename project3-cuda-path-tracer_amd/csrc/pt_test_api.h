// pt_test_api.h -- the entry points of include/pt_amd_test.h (device primitives one by one over host arrays, the soundness sweeps of the
// certificates, the fault-word hook, the instrumented builds' read-outs).  They exist only in libpt_amd_test.so (-DPT_TEST_API), the second
// link target of pt_api.hip, which includes this file inside its extern "C" block; the product library exports none of them.
#pragma once

// ---- diagnostics of a renderer (this library's own instance: the product exports none) ----------------------
int pt_debug_trace_paths(int iter, int bounces, float *origin3, float *dir3, float *color3, int32_t *pixelIndex,
                         int32_t *count) {
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_debug_trace_paths before pt_init");
    if (bounces < 0 || bounces > PT_MAX_DEPTH || !count) return fail(PT_ERR_INVALID, "pt_debug_trace_paths: bad argument");
    if (bounces > R().prm.traceDepth) return fail(PT_ERR_INVALID, "pt_debug_trace_paths: bounces > traceDepth");
    int rc = sync_all();
    if (rc) return rc;
    Slot &sl = R().slot[0];
    if (bounces == 0) {   // camera rays only (they never exist in HBM: generation is fused into bounce 1)
        const int nl = R().nLocal;
        *count = nl;
        if (nl == 0) return PT_OK;
        DevBuf<float> o, d;
        DevBuf<int> px;
        if ((rc = o.alloc((size_t)nl * 3)) || (rc = d.alloc((size_t)nl * 3)) || (rc = px.alloc(nl))) return rc;
        hipLaunchKernelGGL(k_debug_camera_rays, dim3((nl + kBlock - 1) / kBlock), dim3(kBlock), 0, sl.stream, R().prm, iter,
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
    std::vector<unsigned long long> lists((size_t)kSeg * R().poolChunks);
    HIPCHECK(hipMemcpy(lists.data(), pb.list, lists.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint32_t segn[kSeg];
    size_t n = 0;
    for (int sg = 0; sg < kSeg; ++sg) {
        segn[sg] = h.pos[sl.parity][bounces + 1][sg][0];
        n += segn[sg];
    }
    *count = (int32_t)n;
    if (n == 0) return PT_OK;
    const uint32_t chunkPaths = 1u << R().prm.chunkShift;
    // the pool's three arrays (A: 16 B, B: 16 B, C: 12 B per path), chunk by chunk, into the eleven columns
    std::vector<float> cols[kNumArrays];
    for (int k = 0; k < kNumArrays; ++k) cols[k].resize(n);
    {
        std::vector<float> bufA((size_t)chunkPaths * 4), bufB((size_t)chunkPaths * 4), bufC((size_t)chunkPaths * 3);
        size_t off = 0;
        for (int sg = 0; sg < kSeg; ++sg)
            for (uint32_t done = 0, j = 0; done < segn[sg]; done += chunkPaths, ++j) {
                const uint32_t m = std::min<uint32_t>(chunkPaths, segn[sg] - done);
                const unsigned long long e = lists[(size_t)sg * R().poolChunks + j];
                const uint32_t c = j == 0 ? 1u + (uint32_t)sg : (uint32_t)e;
                if (c == 0 || c >= (uint32_t)R().poolChunks || (j != 0 && (uint32_t)(e >> 32) != gen))
                    return fail(PT_ERR_DEVICE, "pt_debug_trace_paths: corrupt chunk list");
                const size_t first = (size_t)c << R().prm.chunkShift;
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

// ---- primitive tests over host arrays ------------------------------------------------------------------
int pt_test_force_fault(int which) {
    if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device");
    if (which != 0 && which != 2) return fail(PT_ERR_INVALID, "pt_test_force_fault: which must be 0 (clear) or 2 (the renderer's fault word)");
    if (!R().init) return fail(PT_ERR_NOT_INIT, "pt_test_force_fault before pt_init");
    HIPCHECK(hipDeviceSynchronize());
    const uint32_t word = which == 2 ? 1u : 0u;
    for (int i = 0; i < (which == 2 ? 1 : R().nslots); ++i) HIPCHECK(hipMemcpy(&R().slot[i].ctrl->error, &word, sizeof word, hipMemcpyHostToDevice));
    if (R().hostFault) *R().hostFault = word;
    return PT_OK;
}

#define NEED_GPU() do { if (count_devices() < 1) return fail(PT_ERR_NO_GPU, "no HIP device"); } while (0)
#define UP(buf, host, count) do { int rc_ = buf.alloc(count); if (rc_) return rc_; \
    HIPCHECK(hipMemcpy(buf.p, host, (size_t)(count) * sizeof(*buf.p), hipMemcpyHostToDevice)); } while (0)
#define DOWN(host, buf, count) HIPCHECK(hipMemcpy(host, buf.p, (size_t)(count) * sizeof(*buf.p), hipMemcpyDeviceToHost))
#define GRID(n) dim3(((n) + 255) / 256), dim3(256), 0, 0

int pt_test_group_fail_next_reduce(PtGroup *g, int count) {
    if (!g || count < 0) return fail(PT_ERR_INVALID, "pt_test_group_fail_next_reduce: bad argument");
    g->failReduces = count;
    return PT_OK;
}

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

// host only: no GPU is touched
int pt_test_wall_planes(const PtGeom *geoms, int ngeoms, float *planes, int32_t *wall_geom, int32_t *nslot, int32_t *nplane, int32_t *nwalls) {
    if (!geoms || ngeoms < 1 || !planes || !wall_geom || !nslot || !nplane || !nwalls) return fail(PT_ERR_INVALID, "pt_test_wall_planes: bad argument");
    std::vector<GeomDev> hg(ngeoms);
    for (int i = 0; i < ngeoms; ++i) {
        if (geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_wall_planes: cubes only");
        pack_geom(geoms[i], hg[i]);
    }
    KParams k;
    memset(&k, 0, sizeof k);
    std::vector<WallBox> hw(kWallMax);
    std::vector<int> wallGeom;
    choose_walls(geoms, ngeoms, hg, k, hw, wallGeom);
    *nslot = k.nSlotWalls; *nplane = k.nPlaneWalls; *nwalls = k.nWalls;
    for (int w = 0; w < kWallMax; ++w) {
        for (int q = 0; q < 5; ++q) planes[5 * w + q] = w < k.nPlaneWalls ? k.planeN[w][q] : 0.0f;
        wall_geom[w] = w < (int)wallGeom.size() ? wallGeom[w] : -1;
    }
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
        if (geoms[i].type != PT_SPHERE && geoms[i].type != PT_CUBE) return fail(PT_ERR_INVALID, "pt_test_sphere_halfline_sweep: spheres and cubes only");
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

int pt_test_sphere_group_sweep(const PtGeom *geoms, int ngeoms, uint64_t seed, int64_t rays, uint64_t *certified, uint64_t *violations, int32_t *ngroups) {
    NEED_GPU();
    if (!geoms || ngeoms < 2 || !certified || !violations || rays < 0) return fail(PT_ERR_INVALID, "pt_test_sphere_group_sweep: bad argument");
    for (int i = 0; i < ngeoms; ++i)
        if (geoms[i].type == PT_MESH) return fail(PT_ERR_INVALID, "pt_test_sphere_group_sweep: spheres and cubes only");
    // the groups exactly as pt_init builds them for this scene's spheres (one cluster: clusters only split the table in two before the grouping)
    std::vector<GeomDev> hg;
    std::vector<SphereCull> sc;
    pack_sphere_cull(geoms, ngeoms, hg, sc);
    if (sc.size() < 2) return fail(PT_ERR_INVALID, "pt_test_sphere_group_sweep: fewer than two spheres");
    double kmax = 0.0;
    for (const SphereCull &e : sc) kmax = std::max(kmax, (double)e.cullK);
    const float sdir = std::nextafter((float)std::sqrt(1.0 / (1.0 - kmax)), INFINITY);
    const double ob = scene_origin_bound(geoms, ngeoms, hg);
    std::vector<SphereCull> groups;
    int n0 = 0, grpN0 = 0;
    build_sphere_groups(sc, n0, ob, sdir, groups, grpN0);
    const int ng = (int)(sc.size() / (size_t)kSphGroupSize);
    if (ngroups) *ngroups = ng;
    std::vector<GeomDev> hs;
    for (const SphereCull &e : sc) hs.push_back(hg[e.geom]);
    F3 slo = F3{INFINITY, INFINITY, INFINITY}, shi = F3{-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < ngeoms; ++i) {
        const float r = hg[i].boundR;
        if (!std::isfinite(r)) continue;
        slo = F3{std::min(slo.x, hg[i].centre[0] - r), std::min(slo.y, hg[i].centre[1] - r), std::min(slo.z, hg[i].centre[2] - r)};
        shi = F3{std::max(shi.x, hg[i].centre[0] + r), std::max(shi.y, hg[i].centre[1] + r), std::max(shi.z, hg[i].centre[2] + r)};
    }
    DevBuf<GeomDev> ds;
    DevBuf<SphereCull> dg;
    DevBuf<unsigned long long> cnt;
    UP(ds, hs.data(), (int)hs.size());
    UP(dg, groups.data(), (int)groups.size());
    int rc = cnt.alloc(2);
    if (rc) return rc;
    HIPCHECK(hipMemset(cnt.p, 0, 16));
    const int per_thread = 64, threads = 256;
    long long blocks = (rays + (long long)per_thread * threads - 1) / ((long long)per_thread * threads);
    if (blocks < 1) blocks = 1;
    if (blocks > (1 << 22)) blocks = 1 << 22;
    hipLaunchKernelGGL(k_sweep_sphere_groups, dim3((unsigned)blocks), dim3(threads), 0, 0, ds.p, (int)hs.size(), dg.p, ng, sdir, std::nextafter((float)ob, 0.0f), slo, shi,
                       (unsigned long long)seed, per_thread, cnt.p, cnt.p + 1);
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long hc[2] = {0, 0};
    HIPCHECK(hipMemcpy(hc, cnt.p, 16, hipMemcpyDeviceToHost));
    *certified = hc[0];
    *violations = hc[1];
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
    *nplane = k.nSlotWalls + k.nPlaneWalls;
    *certified = *violations = *single = 0;
    if (k.nWalls < 1 || k.nSlotWalls + k.nPlaneWalls < 1) return PT_OK;
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

