// pt_mesh.h -- host side of the triangle meshes: the bounding-volume hierarchy ptd::meshIntersectionTest walks.
// (README.md:112-116, 236 name the object type "mesh"; the reference holds no mesh code -- semantics in pt_device.h.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pt_device.h"

namespace ptm {

using ptd::MeshUnit;
using ptd::kMeshEnd;
using ptd::kMeshLeaf;
using ptd::kMeshNodeUnits;
using ptd::kMeshTriUnits;

// what a GeomDev needs to know about its mesh, and what pt_init needs for the LDS budget
struct MeshLayout {
    uint32_t root;      // ref of the root node in the copy of octant 0
    uint32_t stride;    // units per octant copy: 2 max(ntris - 1, 1)
    int stackNeed;      // far children that can wait at once on a lane's stack (ptd::meshIntersectionTest)
};
// a hierarchy that would need more stack than this is rebuilt by median splits alone (need <= ceil(log2 ntris))
constexpr int kMeshStackSoftMax = 24;

// triangle soup registered by pt_set_meshes for the next pt_init
struct HostMesh {
    int geom;
    std::vector<float> tris;    // 9 floats per triangle: v0, v1, v2 (object space)
    std::vector<float> normals; // 9 floats per triangle: vertex normals (PtMesh::normals), or empty: flat shading
    std::vector<int> mats;      // one per triangle: the face's scene material (PtMesh::materials; -1 = the object's), or empty
};

// The per-mesh margin every triangle box is inflated by: 1e-5 of the largest |coordinate| (fp32, as the oracle's mesh_margin).
inline float meshMargin(const float *tris, int ntris) {
    float maxAbs = 0.0f;
    for (size_t i = 0; i < 9 * (size_t)ntris; ++i) {
        const float a = std::fabs(tris[i]);
        if (a > maxAbs) maxAbs = a;
    }
    float m = 1e-5f * maxAbs;
    if (!(m >= 1e-30f)) m = 1e-30f;
    return m;
}

inline float min2(float a, float b) { return a < b ? a : b; }
inline float max2(float a, float b) { return a < b ? b : a; }

// fp32 -> fp16 bits, rounded DOWN (towards -inf) or UP: what a box plane may do when the box only has to contain its content.
// (values beyond the half range go to the largest finite half or to the infinity on the permitted side; NaN does not occur)
inline uint16_t halfBitsDirected(float x, bool up) {
    if (x == 0.0f) return 0;
    const bool neg = x < 0.0f;
    const double a = std::fabs((double)x);
    // magnitude rounds away from zero when the value moves in its own direction (up for positives, down for negatives)
    const bool away = up != neg;
    uint32_t bits;
    if (std::isinf(a)) bits = 0x7c00u;
    else if (a >= 65504.0) bits = a > 65504.0 && away ? 0x7c00u : 0x7bffu;
    else {
        int e;
        (void)std::frexp(a, &e);                       // a = m 2^e, m in [0.5, 1)
        int exp = e - 1;                               // a = 1.f 2^exp
        double q;                                      // spacing of halves at this magnitude
        if (exp < -14) { exp = -15; q = std::ldexp(1.0, -24); }      // subnormal halves
        else q = std::ldexp(1.0, exp - 10);
        const double n = a / q;                        // exact: a is a float, q a power of two
        double k = away ? std::ceil(n) : std::floor(n);
        if (exp == -15) bits = (uint32_t)k;            // 0 .. 1024 (1024 = the smallest normal)
        else {
            // k in [1024, 2048]; 2048 carries into the next exponent
            bits = ((uint32_t)(exp + 15) << 10) + ((uint32_t)k - 1024u);
        }
        if (bits > 0x7c00u) bits = 0x7c00u;
    }
    return (uint16_t)(bits | (neg ? 0x8000u : 0u));
}

// Appends one mesh to the scene's record array: its triangles in file order (a triangle's record index is its rank in the tie rule),
// then its inner nodes.  A triangle's box is its bounding box inflated by the margin -- the very box the semantics test -- and an
// inner node carries the boxes of its two children, each the exact union of what lies below (min / max of floats: no rounding), so
// box inclusion holds exactly, which is all the traversal's equivalence with the brute-force rule needs.  Split by a binned
// surface-area heuristic with a median fallback (below).
//
// The inner nodes are laid out EIGHT times, once per sign octant of the ray direction, each in depth-first order with the child on
// the ray's near side of the split FIRST in its parent's record: a ray walks the copy of its octant (root + octant * stride) front
// to back, so the first hits prune most of what lies behind them.  (The order of the visits never changes the result, only how much
// is pruned.)  `flat`: no hierarchy -- a chain of nodes whose near child is triangle i and whose far child, in a box that holds
// everything, is the rest of the chain: the brute-force rule on the device, for the tests.  bbox receives the union of all
// triangle boxes (lo[3], hi[3]).
// `normals` (9 floats per triangle, or nullptr) / `mats` (one int per triangle, or nullptr): the triangle records' two spare words --
// word 10 = 1 + the face's material (0: the object's), word 11 = the ref of the triangle's three normal units (0: flat shading), which
// follow the mesh's triangle records: only the WINNING triangle of a walk ever reads them.
inline MeshLayout appendMesh(const float *tris, int ntris, bool flat, std::vector<MeshUnit> &recs, float bbox[6], const float *normals = nullptr,
                             const int *mats = nullptr) {
    const float m = meshMargin(tris, ntris);
    while (recs.size() % 4) recs.push_back(MeshUnit());       // (64-byte alignment of a mesh's first record)
    const uint32_t triBase = (uint32_t)recs.size();
    const uint32_t normBase = triBase + (uint32_t)kMeshTriUnits * (uint32_t)ntris;       // (behind the triangle records)
    struct Leaf { float lo[3], hi[3], c[3]; int idx; };
    std::vector<Leaf> leaves((size_t)ntris);
    for (int a = 0; a < 3; ++a) { bbox[a] = INFINITY; bbox[3 + a] = -INFINITY; }
    for (int i = 0; i < ntris; ++i) {
        const float *t = tris + 9 * (size_t)i;
        float mt[12];
        memset(mt, 0, sizeof mt);
        Leaf &L = leaves[(size_t)i];
        L.idx = i;
        for (int a = 0; a < 3; ++a) {
            mt[a] = t[a];
            mt[3 + a] = t[3 + a];
            mt[6 + a] = t[6 + a];
            L.lo[a] = min2(min2(t[a], t[3 + a]), t[6 + a]) - m;
            L.hi[a] = max2(max2(t[a], t[3 + a]), t[6 + a]) + m;
            L.c[a] = 0.5f * (L.lo[a] + L.hi[a]);
            bbox[a] = min2(bbox[a], L.lo[a]);
            bbox[3 + a] = max2(bbox[3 + a], L.hi[a]);
        }
        mt[9] = m;
        MeshUnit u[3];
        memcpy(u, mt, sizeof mt);
        u[2].w[2] = mats && mats[i] >= 0 ? (uint32_t)mats[i] + 1u : 0u;
        u[2].w[3] = normals ? normBase + 3u * (uint32_t)i : 0u;
        recs.push_back(u[0]);
        recs.push_back(u[1]);
        recs.push_back(u[2]);
    }
    if (normals)
        for (int i = 0; i < ntris; ++i) {
            float nt[12];
            memset(nt, 0, sizeof nt);
            memcpy(nt, normals + 9 * (size_t)i, 9 * sizeof(float));
            MeshUnit u[3];
            memcpy(u, nt, sizeof nt);
            recs.push_back(u[0]);
            recs.push_back(u[1]);
            recs.push_back(u[2]);
        }
    MeshLayout lay;
    lay.stackNeed = 0;
    // an inner node's record for octant `oct`: child h's box as entry / exit planes (entry = lo where the rays of the octant run
    // towards +axis, hi where they run towards -axis), refs in f[3] and f[7]
    struct Node { MeshUnit u[2]; };
    auto innerRec = [](int oct, const float *lo0, const float *hi0, uint32_t ref0, const float *lo1, const float *hi1, uint32_t ref1) {
        Node n;
        const float *lo[2] = {lo0, lo1}, *hi[2] = {hi0, hi1};
        const uint32_t ref[2] = {ref0, ref1};
        for (int h = 0; h < 2; ++h) {
            uint16_t pl[6];                      // entry x, y, z, exit x, y, z; lo rounded down, hi up
            for (int a = 0; a < 3; ++a) {
                const bool neg = ((oct >> a) & 1) != 0;
                const uint16_t l = halfBitsDirected(lo[h][a], false), u = halfBitsDirected(hi[h][a], true);
                pl[a] = neg ? u : l;
                pl[3 + a] = neg ? l : u;
            }
            for (int k = 0; k < 3; ++k) n.u[h].w[k] = (uint32_t)pl[2 * k] | ((uint32_t)pl[2 * k + 1] << 16);
            n.u[h].w[3] = ref[h];
        }
        return n;
    };
    auto pushNode = [&recs](const Node &n) { recs.push_back(n.u[0]); recs.push_back(n.u[1]); };
    // a box no ray passes (entry planes at +inf of the ray's parameter, exit planes at -inf) / every ray passes
    const float kNone[2][3] = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    const float kAll[2][3] = {{-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
    if (recs.size() % 2) recs.push_back(MeshUnit());          // (an inner node's two units share a 32-byte sector)
    const uint32_t innerBase = (uint32_t)recs.size();
    if (ntris == 1 || flat) {
        // one triangle: a root whose near child is the triangle and whose far child is nothing.  flat: a chain -- near child =
        // triangle i, far child = the rest of the chain in a box that holds everything (the last node: triangle ntris - 1)
        const uint32_t stride = (uint32_t)std::max(ntris - 1, 1) * kMeshNodeUnits;
        for (int oct = 0; oct < 8; ++oct) {
            const uint32_t base = innerBase + (uint32_t)oct * stride;
            for (uint32_t i = 0; i * kMeshNodeUnits < stride; ++i) {
                const Leaf &L = leaves[(size_t)i];
                const uint32_t ref0 = kMeshLeaf | (triBase + kMeshTriUnits * i);
                if (ntris == 1) pushNode(innerRec(oct, L.lo, L.hi, ref0, kNone[0], kNone[1], ref0));
                else if (i + 2 == (uint32_t)ntris) pushNode(innerRec(oct, L.lo, L.hi, ref0, leaves[(size_t)i + 1].lo, leaves[(size_t)i + 1].hi, kMeshLeaf | (triBase + kMeshTriUnits * (i + 1u))));
                else pushNode(innerRec(oct, L.lo, L.hi, ref0, kAll[0], kAll[1], base + kMeshNodeUnits * (i + 1u)));
            }
        }
        lay.root = innerBase;
        lay.stride = stride;
        lay.stackNeed = ntris == 1 ? 0 : 1;
        return lay;
    }
    // the tree: node k has a box and either a triangle or two children (lower / upper half along `axis`)
    struct TreeNode { float lo[3], hi[3]; int tri, left, right, axis, size; };
    std::vector<TreeNode> tree;
    tree.reserve(2 * (size_t)ntris);
    struct Builder {
        std::vector<Leaf> &lv;
        std::vector<TreeNode> &tree;
        bool medianOnly;
        int build(int lo, int hi) {                       // recursion depth: ceil(log2 ntris) + 1
            const int me = (int)tree.size();
            tree.push_back(TreeNode());
            if (hi - lo == 1) {
                TreeNode &n = tree[(size_t)me];
                for (int a = 0; a < 3; ++a) { n.lo[a] = lv[(size_t)lo].lo[a]; n.hi[a] = lv[(size_t)lo].hi[a]; }
                n.tri = lv[(size_t)lo].idx;
                n.left = n.right = -1;
                n.axis = 0;
                n.size = 1;
                return me;
            }
            float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = lo; i < hi; ++i)
                for (int a = 0; a < 3; ++a) { cmin[a] = min2(cmin[a], lv[(size_t)i].c[a]); cmax[a] = max2(cmax[a], lv[(size_t)i].c[a]); }
            // Split by the surface-area heuristic over 16 bins of the centroids per axis (cost = area x count of the two
            // sides; double precision: the choice only shapes the tree, it never reaches a result).  A split that leaves less
            // than an eighth on one side, or no usable split, falls back to the median along the widest axis, which keeps the
            // depth logarithmic.
            const int count = hi - lo;
            int axis = 0, mid = lo + count / 2;
            bool byBins = false;
            constexpr int kBins = 16;
            auto binOf = [&](const Leaf &L, int a) {
                const double w = (double)cmax[a] - (double)cmin[a];
                const int b = (int)(((double)L.c[a] - (double)cmin[a]) / w * kBins);
                return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            };
            if (count > 4 && !medianOnly) {
                double bestCost = INFINITY;
                int bestBin = 0;
                for (int a = 0; a < 3; ++a) {
                    if (!((double)cmax[a] - (double)cmin[a] > 0)) continue;
                    int cnt[kBins] = {0};
                    double blo[kBins][3], bhi[kBins][3];
                    for (int k = 0; k < kBins; ++k)
                        for (int q = 0; q < 3; ++q) { blo[k][q] = INFINITY; bhi[k][q] = -INFINITY; }
                    for (int i = lo; i < hi; ++i) {
                        const Leaf &L = lv[(size_t)i];
                        const int k = binOf(L, a);
                        ++cnt[k];
                        for (int q = 0; q < 3; ++q) { blo[k][q] = std::min(blo[k][q], (double)L.lo[q]); bhi[k][q] = std::max(bhi[k][q], (double)L.hi[q]); }
                    }
                    auto area = [](const double *l, const double *h) {
                        const double x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2];
                        return x * y + y * z + z * x;
                    };
                    double rl[kBins][3], rh[kBins][3];      // boxes and counts of the bins k .. kBins - 1
                    int rn[kBins];
                    double cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
                    int cn = 0;
                    for (int k = kBins - 1; k >= 0; --k) {
                        for (int q = 0; q < 3; ++q) { cl[q] = std::min(cl[q], blo[k][q]); ch[q] = std::max(ch[q], bhi[k][q]); }
                        cn += cnt[k];
                        for (int q = 0; q < 3; ++q) { rl[k][q] = cl[q]; rh[k][q] = ch[q]; }
                        rn[k] = cn;
                    }
                    double ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY};
                    int ln = 0;
                    for (int k = 0; k + 1 < kBins; ++k) {          // left = bins 0 .. k
                        for (int q = 0; q < 3; ++q) { ll[q] = std::min(ll[q], blo[k][q]); lh[q] = std::max(lh[q], bhi[k][q]); }
                        ln += cnt[k];
                        const int rnn = rn[k + 1];
                        if (ln * 8 < count || rnn * 8 < count) continue;
                        const double cost = area(ll, lh) * ln + area(rl[k + 1], rh[k + 1]) * rnn;
                        if (cost < bestCost) {
                            bestCost = cost;
                            axis = a;
                            bestBin = k;
                            byBins = true;
                        }
                    }
                }
                if (byBins) {
                    const int a = axis, kb = bestBin;
                    const auto it = std::stable_partition(lv.begin() + lo, lv.begin() + hi, [&](const Leaf &L) { return binOf(L, a) <= kb; });
                    mid = (int)(it - lv.begin());
                }
            }
            if (!byBins) {
                axis = 0;
                for (int a = 1; a < 3; ++a)
                    if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
                mid = lo + count / 2;
                std::nth_element(lv.begin() + lo, lv.begin() + mid, lv.begin() + hi, [axis](const Leaf &x, const Leaf &y) {
                    return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.idx < y.idx);
                });
            }
            const int l = build(lo, mid), r = build(mid, hi);
            TreeNode &n = tree[(size_t)me];
            for (int a = 0; a < 3; ++a) {
                n.lo[a] = min2(tree[(size_t)l].lo[a], tree[(size_t)r].lo[a]);
                n.hi[a] = max2(tree[(size_t)l].hi[a], tree[(size_t)r].hi[a]);
            }
            n.tri = -1;
            n.left = l; n.right = r;
            n.axis = axis;
            n.size = 1 + tree[(size_t)l].size + tree[(size_t)r].size;
            return me;
        }
    } builder{leaves, tree, false};
    builder.build(0, ntris);
    // what a lane's stack must hold: the far child waits while the near one is walked (the near child depends on the octant)
    auto stackNeed = [&](int oct) {
        struct Rec {
            const std::vector<TreeNode> &tree;
            int oct;
            int need(int k) const {
                const TreeNode &t = tree[(size_t)k];
                if (t.tri >= 0) return 0;
                const bool upperFirst = ((oct >> t.axis) & 1) != 0;
                const int nearC = upperFirst ? t.right : t.left, farC = upperFirst ? t.left : t.right;
                return std::max(1 + need(nearC), need(farC));
            }
        } r{tree, oct};
        return r.need(0);
    };
    auto needAll = [&]() {
        int n = 0;
        for (int oct = 0; oct < 8; ++oct) n = std::max(n, stackNeed(oct));
        return n;
    };
    lay.stackNeed = needAll();
    // (tests only: PT_AMD_MESH_STACK_MAX lowers the threshold so that ordinary meshes take the rebuild)
    const char *envMax = getenv("PT_AMD_MESH_STACK_MAX");
    const int softMax = envMax && atoi(envMax) > 0 ? atoi(envMax) : kMeshStackSoftMax;
    if (lay.stackNeed > softMax) {                           // (lopsided splits all the way down: median splits bound the depth by log2)
        tree.clear();
        std::sort(leaves.begin(), leaves.end(), [](const Leaf &x, const Leaf &y) { return x.idx < y.idx; });
        builder.medianOnly = true;
        builder.build(0, ntris);
        lay.stackNeed = needAll();
    }
    const uint32_t stride = ((uint32_t)ntris - 1u) * kMeshNodeUnits;     // (a binary tree with ntris leaves has ntris - 1 inner nodes)
    for (int oct = 0; oct < 8; ++oct) {
        // depth-first emission of the inner nodes; a child's ref is known once its position is: inner children are numbered as
        // they are met (near child = the next record, far child = after the near child's inner nodes)
        struct Emit {
            const std::vector<TreeNode> &tree;
            std::vector<MeshUnit> &recs;
            uint32_t triBase;
            int oct;
            decltype(innerRec) &innerRec;
            static int inner(const std::vector<TreeNode> &tree, int k) { return tree[(size_t)k].tri >= 0 ? 0 : (tree[(size_t)k].size - 1) / 2; }
            void emit(int k) {
                const TreeNode &t = tree[(size_t)k];
                const bool upperFirst = ((oct >> t.axis) & 1) != 0;   // the ray runs towards -axis: the upper half is nearer
                const int c[2] = {upperFirst ? t.right : t.left, upperFirst ? t.left : t.right};
                const uint32_t me = (uint32_t)recs.size();
                recs.push_back(MeshUnit());
                recs.push_back(MeshUnit());
                uint32_t ref[2];
                ref[0] = tree[(size_t)c[0]].tri >= 0 ? (kMeshLeaf | (triBase + kMeshTriUnits * (uint32_t)tree[(size_t)c[0]].tri)) : me + kMeshNodeUnits;
                ref[1] = tree[(size_t)c[1]].tri >= 0 ? (kMeshLeaf | (triBase + kMeshTriUnits * (uint32_t)tree[(size_t)c[1]].tri))
                                                      : me + kMeshNodeUnits * (1u + (uint32_t)inner(tree, c[0]));
                const auto n = innerRec(oct, tree[(size_t)c[0]].lo, tree[(size_t)c[0]].hi, ref[0], tree[(size_t)c[1]].lo, tree[(size_t)c[1]].hi, ref[1]);
                recs[(size_t)me] = n.u[0];
                recs[(size_t)me + 1] = n.u[1];
                if (tree[(size_t)c[0]].tri < 0) emit(c[0]);
                if (tree[(size_t)c[1]].tri < 0) emit(c[1]);
            }
        } e{tree, recs, triBase, oct, innerRec};
        e.emit(0);
    }
    lay.root = innerBase;
    lay.stride = stride;
    return lay;
}

}  // namespace ptm
