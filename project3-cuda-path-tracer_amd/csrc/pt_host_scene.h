// pt_host_scene.h -- host side of pt_init: what the kernels need to know about a scene, derived once per pt_init in double precision or with
// the very fp32 operations the kernels would issue -- packed primitives and materials (GeomDev, MaterialDev), the walls' inflated boxes and
// one-plane certificates, the sphere clusters of sphere-heavy scenes, the camera constants and the screen-space culling tables of the
// camera-ray bounce.  Pure host code: no renderer state, no HIP call.  Header-only part of the single translation unit pt_api.hip
// (inside its anonymous namespace).
#pragma once

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

// Scenes of HUNDREDS of swept primitives (k_bounce<..., GROUPS>): the packed table in spatial GROUPS of kSphGroupSize consecutive entries, each
// with a bounding ball.  `sc` comes in as build_sphere_clusters left it (cluster 0 = [0, n0), cluster 1 = the rest; n0 = 0: no clusters) and
// leaves with every cluster's entries ordered by recursive median splits of their centres (the widest axis; the left side a multiple of the
// group size, so only a cluster's last group is short: padded with copies of its last entry -- testing a primitive twice changes nothing),
// cluster 0 padded to an EVEN number of groups; `groups` receives one entry per group in the table's own format (geom = -1).
// What a group's certificate rests on: its ball holds, for every member i, the ball of radius r_i = sqrt(cullR2_i + K_i ocMax_i^2) (1 + 1e-6)
// around the member's centre -- exactly the balls build_sphere_clusters' boxes hold --, ocMax_i = omax + |centre_i| for the origins with
// |x| + |y| + |z| <= omax.  The device tests the group like a member (ptd::sphereHalfLineExcessScaled, the scene's one scaled direction):
// x = |oc|^2 - s^2 p^2 > T = R^2 s^2 (1 + 1e-6) implies d^2 - K_max |oc|^2 >= x / s^2 > R^2 up to fp32 rounding of x (4 ulp of |oc|^2 = 2.4e-7
// |oc|^2, against K_max >= 1e-4), i.e. the exact half-line passes the group's ball at a distance -- and so every member's ball: each member's
// own certificate would hold, with the room its ball was given.  Origins beyond omax take every group (k_bounce).
// (tests/test_gpu_parity.py::test_sphere_group_balls_never_reject_a_hit: pt_test_sphere_group_sweep.)
void build_sphere_groups(std::vector<SphereCull> &sc, int &n0, double omax, float sdir, std::vector<SphereCull> &groups, int &grpN0) {
    const int S = kSphGroupSize;
    auto order = [&](std::vector<SphereCull> &v) {
        struct Rec {
            std::vector<SphereCull> &v;
            int S;
            void run(int lo, int hi) {
                if (hi - lo <= S) return;
                double cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int i = lo; i < hi; ++i)
                    for (int a = 0; a < 3; ++a) { cmin[a] = std::min(cmin[a], (double)v[(size_t)i].centre[a]); cmax[a] = std::max(cmax[a], (double)v[(size_t)i].centre[a]); }
                int axis = 0;
                for (int a = 1; a < 3; ++a)
                    if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
                int mid = lo + ((hi - lo) / 2 + S - 1) / S * S;
                if (mid >= hi) mid = lo + (hi - lo) / 2 / S * S;
                if (mid <= lo) return;
                std::nth_element(v.begin() + lo, v.begin() + mid, v.begin() + hi, [axis](const SphereCull &x, const SphereCull &y) {
                    return x.centre[axis] < y.centre[axis] || (x.centre[axis] == y.centre[axis] && x.geom < y.geom);
                });
                run(lo, mid);
                run(mid, hi);
            }
        } r{v, S};
        r.run(0, (int)v.size());
        while (v.size() % (size_t)S) v.push_back(v.back());
    };
    std::vector<SphereCull> c0(sc.begin(), sc.begin() + n0), c1(sc.begin() + n0, sc.end());
    if (!c0.empty()) {
        order(c0);
        if ((c0.size() / (size_t)S) % 2) c0.insert(c0.end(), c0.end() - S, c0.end());      // an even number of groups: the last one once more
    }
    if (!c1.empty()) order(c1);
    sc = c0;
    sc.insert(sc.end(), c1.begin(), c1.end());
    n0 = (int)c0.size();
    grpN0 = n0 / S;
    groups.clear();
    for (size_t g = 0; g * (size_t)S < sc.size(); ++g) {
        double c[3] = {0, 0, 0};
        for (int i = 0; i < S; ++i)
            for (int a = 0; a < 3; ++a) c[a] += (double)sc[g * S + i].centre[a] / S;
        SphereCull e;
        memset(&e, 0, sizeof e);
        for (int a = 0; a < 3; ++a) e.centre[a] = (float)c[a];
        double R = 0;
        for (int i = 0; i < S; ++i) {
            const SphereCull &m = sc[g * S + i];
            const double cn = std::sqrt((double)m.centre[0] * m.centre[0] + (double)m.centre[1] * m.centre[1] + (double)m.centre[2] * m.centre[2]);
            const double ocMax = omax + cn;
            const double r = std::sqrt((double)m.cullR2 + (double)m.cullK * ocMax * ocMax) * (1.0 + 1e-6);
            const double dx = (double)m.centre[0] - e.centre[0], dy = (double)m.centre[1] - e.centre[1], dz = (double)m.centre[2] - e.centre[2];
            R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + r);
        }
        R *= 1.0 + 1e-6;
        e.cullR2 = std::nextafter((float)(R * R * (double)sdir * (double)sdir * (1.0 + 1e-6)), INFINITY);
        e.cullK = 0.0f;
        e.geom = -1;
        groups.push_back(e);
    }
    if (groups.size() % 2) groups.push_back(groups.back());      // (the table is read two entries per scalar load; the pad is never evaluated)
}

// the bound on the ray origins worth a certificate: a scattered ray starts on a primitive -- the scene's own extent, |x| + |y| + |z| over its
// bounding box, with a quarter to spare (as build_sphere_clusters takes it)
double scene_origin_bound(const PtGeom *geoms, int ngeoms, const std::vector<GeomDev> &hg) {
    double omax = 0.0, slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < ngeoms; ++i) {
        WallBox wb;
        if (geoms[i].type == PT_CUBE && wall_box(geoms[i], wb) >= 0)
            for (int a = 0; a < 3; ++a) { slo[a] = std::min(slo[a], (double)wb.lo[a]); shi[a] = std::max(shi[a], (double)wb.hi[a]); }
        else if (std::isfinite(hg[i].boundR))
            for (int a = 0; a < 3; ++a) { slo[a] = std::min(slo[a], (double)hg[i].centre[a] - hg[i].boundR); shi[a] = std::max(shi[a], (double)hg[i].centre[a] + hg[i].boundR); }
    }
    for (int a = 0; a < 3; ++a) omax += std::max(std::fabs(slo[a]), std::fabs(shi[a]));
    omax *= 1.25;
    return std::isfinite(omax) ? omax : 0.0;
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
    k.nPlaneWalls = 0;
    memset(k.planeN, 0, sizeof k.planeN);
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
    // ROTATED walls (an edge of the cube that is not parallel to a world axis): their world boxes are far larger than they are, so they
    // take no slot; the plane of the face that looks at C, the middle of `outer`, certifies them instead (ptd::wallPlanesOriented):
    // n = that face's unit normal towards C, th = the largest n . corner + the box's inflation (wall_box: 4e-5 S) + the slack (the
    // exit point's rounding as above, the two dot products', and n's own rounding to float: 3e-6 (wallOMax + |diagonal of outer|)).
    std::vector<int> rotated(n, 0);
    std::vector<std::array<double, 4>> plane(n);
    std::vector<double> planeS(n, 0.0);
    double Smax = 0;            // the largest inflation scale among the walls (wall_box)
    std::vector<std::array<double, 3>> allCorners;   // every wall's cube, corner by corner
    for (int w = 0; w < n; ++w) {
        const PtGeom &g = geoms[which[w]];
        WallBox tmp;
        Smax = std::max(Smax, wall_box(g, tmp));
        for (int q = 0; q < 8; ++q) {
            const double o[3] = {(q & 1) ? 0.5 : -0.5, (q & 2) ? 0.5 : -0.5, (q & 4) ? 0.5 : -0.5};
            std::array<double, 3> c;
            for (int r = 0; r < 3; ++r)
                c[r] = (double)g.transform[0 + r] * o[0] + (double)g.transform[4 + r] * o[1] + (double)g.transform[8 + r] * o[2] + (double)g.transform[12 + r];
            allCorners.push_back(c);
        }
    }
    const bool noOriented = getenv("PT_AMD_NO_ORIENTED_WALLS") && atoi(getenv("PT_AMD_NO_ORIENTED_WALLS"));     // (experiments: round 4's behaviour)
    for (int w = 0; w < n && std::isfinite(slack) && !noOriented; ++w) {
        const PtGeom &g = geoms[which[w]];
        bool aligned = true;
        for (int c = 0; c < 3; ++c) {           // column c of the transform: the world direction of the cube's edge c
            double m = 0;
            int big = 0;
            for (int r = 0; r < 3; ++r) m = std::max(m, std::fabs((double)g.transform[4 * c + r]));
            for (int r = 0; r < 3; ++r) big += std::fabs((double)g.transform[4 * c + r]) > 1e-7 * m ? 1 : 0;
            aligned = aligned && big <= 1;
        }
        if (aligned) continue;
        // (a SMALL rotated cube -- a tilted light under the ceiling -- keeps the slab test against its world box: the plane of one face says
        // little about a box that spans a fraction of the room, the box around it a lot)
        if ((double)hg[which[w]].boundR < 0.2 * diag) { rotated[w] = -1; continue; }
        double corner[8][3];
        for (int q = 0; q < 8; ++q) {
            const double o[3] = {(q & 1) ? 0.5 : -0.5, (q & 2) ? 0.5 : -0.5, (q & 4) ? 0.5 : -0.5};
            for (int r = 0; r < 3; ++r)
                corner[q][r] = (double)g.transform[0 + r] * o[0] + (double)g.transform[4 + r] * o[1] + (double)g.transform[8 + r] * o[2] + (double)g.transform[12 + r];
        }
        WallBox tmp;
        const double Sw = wall_box(g, tmp);
        const double C[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
        double bestGap = 1e-3 * diag;           // (C clearly on the far side of the face, or no plane)
        for (int a = 0; a < 3; ++a)
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                // outward normal of the face (sgn e_a in the cube's space): sgn x row a of the inverse transform; towards C: its negative
                double nn[3] = {-sgn * (double)g.inverseTransform[0 + a], -sgn * (double)g.inverseTransform[4 + a], -sgn * (double)g.inverseTransform[8 + a]};
                const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                if (!(len > 0) || !std::isfinite(len)) continue;
                for (double &v : nn) v /= len;
                double th = -INFINITY;
                for (int q = 0; q < 8; ++q) th = std::max(th, nn[0] * corner[q][0] + nn[1] * corner[q][1] + nn[2] * corner[q][2]);
                const double gap = (nn[0] * C[0] + nn[1] * C[1] + nn[2] * C[2]) - th;
                if (gap > bestGap && std::isfinite(th)) {
                    bestGap = gap;
                    rotated[w] = 1;
                    plane[w] = {nn[0], nn[1], nn[2], th + 4e-5 * Sw + 3e-6 * (omaxAll + diag)};
                    planeS[w] = Sw;
                }
            }
        if (!rotated[w]) rotated[w] = -1;       // (rotated, no usable face: the slab test)
    }
    for (int w = 0; w < n && std::isfinite(slack); ++w) {            // (largest walls first)
        if (rotated[w] != 0) continue;
        double best = 0;
        for (int a = 0; a < 3; ++a) {
            const double C = 0.5 * (lo[a] + hi[a]), ext = std::max(hi[a] - lo[a], 1e-30);
            const double gLow = (C - boxes[w].hi[a]) / ext, gHigh = (boxes[w].lo[a] - C) / ext;
            if (gLow > best && !taken[2 * a] && std::isfinite((double)boxes[w].hi[a] + slack)) { best = gLow; slot[w] = 2 * a; th[w] = (double)boxes[w].hi[a] + slack; }
            if (gHigh > best && !taken[2 * a + 1] && std::isfinite((double)boxes[w].lo[a] - slack)) { best = gHigh; slot[w] = 2 * a + 1; th[w] = (double)boxes[w].lo[a] - slack; }
        }
        if (slot[w] >= 0) taken[slot[w]] = true;
    }
    for (int pass = 0; pass < 3; ++pass)                              // walls with a slot first, then those with a plane of their own, then the rest
        for (int w = 0; w < n; ++w)
            if ((slot[w] >= 0 ? 0 : (rotated[w] == 1 ? 1 : 2)) == pass) {
                const int idx = (int)wallGeom.size();
                hw[idx] = boxes[w];
                if (pass == 1) {
                    float *pn = k.planeN[k.nPlaneWalls++];
                    for (int q = 0; q < 3; ++q) pn[q] = (float)plane[w][q];
                    const float t = (float)plane[w][3];
                    pn[3] = (double)t < plane[w][3] ? std::nextafter(t, INFINITY) : t;          // (towards the interior: the certificate may only get rarer)
                    // `far`: the half-space n . x >= far holds every wall's inflated cube (the segment's end may only move OUTWARDS)
                    double farD = INFINITY;
                    for (const auto &c : allCorners) farD = std::min(farD, (double)pn[0] * c[0] + (double)pn[1] * c[1] + (double)pn[2] * c[2]);
                    farD -= 4e-5 * Smax + 3e-6 * (omaxAll + diag);
                    const float ff = (float)farD;
                    pn[4] = (double)ff > farD ? std::nextafter(ff, -INFINITY) : ff;
                }
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

