// pt_mesh.h -- host side of the triangle meshes: the bounding-volume hierarchy ptd::meshIntersectionTest walks.
// (README.md:112-116, 236 name the object type "mesh"; the reference holds no mesh code -- semantics in pt_device.h.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pt_device.h"

namespace ptm {

using ptd::MeshNode;
using ptd::MeshTri;
using ptd::kMeshEnd;

// triangle soup registered by pt_set_meshes for the next pt_init
struct HostMesh {
    int geom;
    std::vector<float> tris;    // 9 floats per triangle: v0, v1, v2 (object space)
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

// Appends one mesh to the scene's node and triangle arrays.  Leaves are single triangles whose box is the triangle's
// bounding box inflated by the margin -- the very box the semantics tests -- and an inner node's box is the exact union
// of its children's (min / max of floats: no rounding), so box inclusion holds exactly, which is all the traversal's
// equivalence with the brute-force rule needs.  Median split of the centroids along their widest axis.
//
// The tree is laid out EIGHT times, once per sign octant of the ray direction, each in depth-first order with skip links
// and with the child on the ray's near side of the split first: a ray walks the copy of its octant (root + octant *
// stride) front to back, so the first hits prune most of what lies behind them, and the walk itself stays a loop over
// consecutive nodes without a stack.  (The order of the visits never changes the result, only how much is pruned.)
// Returns the stride = nodes per copy (2 ntris - 1).  `flat`: no hierarchy, ONE list of leaves in index order and
// stride 0 (tests: the brute-force rule on the device).  bbox receives the union of all leaf boxes (lo[3], hi[3]).
inline uint32_t appendMesh(const float *tris, int ntris, bool flat, std::vector<MeshNode> &nodes, std::vector<MeshTri> &out,
                           float bbox[6]) {
    const float m = meshMargin(tris, ntris);
    const int triBase = (int)out.size();
    struct Leaf { float lo[3], hi[3], c[3]; int idx; };
    std::vector<Leaf> leaves((size_t)ntris);
    for (int a = 0; a < 3; ++a) { bbox[a] = INFINITY; bbox[3 + a] = -INFINITY; }
    for (int i = 0; i < ntris; ++i) {
        const float *t = tris + 9 * (size_t)i;
        MeshTri mt;
        memset(&mt, 0, sizeof mt);
        Leaf &L = leaves[(size_t)i];
        L.idx = i;
        for (int a = 0; a < 3; ++a) {
            mt.v0[a] = t[a];
            mt.e1[a] = t[3 + a] - t[a];
            mt.e2[a] = t[6 + a] - t[a];
            L.lo[a] = min2(min2(t[a], t[3 + a]), t[6 + a]) - m;
            L.hi[a] = max2(max2(t[a], t[3 + a]), t[6 + a]) + m;
            L.c[a] = 0.5f * (L.lo[a] + L.hi[a]);
            bbox[a] = min2(bbox[a], L.lo[a]);
            bbox[3 + a] = max2(bbox[3 + a], L.hi[a]);
        }
        out.push_back(mt);
    }
    if (flat) {
        const uint32_t nodeBase = (uint32_t)nodes.size();
        for (int i = 0; i < ntris; ++i) {
            MeshNode n;
            for (int a = 0; a < 3; ++a) { n.lo[a] = leaves[(size_t)i].lo[a]; n.hi[a] = leaves[(size_t)i].hi[a]; }
            n.tri = triBase + i;
            n.skip = i + 1 < ntris ? nodeBase + (uint32_t)i + 1u : kMeshEnd;
            nodes.push_back(n);
        }
        return 0u;
    }
    // the tree: node k has a box and either a triangle or two children (lower / upper half along `axis`)
    struct TreeNode { float lo[3], hi[3]; int tri, left, right, axis, size; };
    std::vector<TreeNode> tree;
    tree.reserve(2 * (size_t)ntris);
    struct Builder {
        std::vector<Leaf> &lv;
        std::vector<TreeNode> &tree;
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
            if (count > 4) {
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
    } builder{leaves, tree};
    builder.build(0, ntris);
    const uint32_t stride = (uint32_t)tree.size();
    for (int oct = 0; oct < 8; ++oct) {
        const uint32_t base = (uint32_t)nodes.size(), end = base + stride;
        // depth-first emission with an explicit stack of (tree node); a node's skip link = its position + its subtree size
        std::vector<int> stack(1, 0);
        while (!stack.empty()) {
            const TreeNode &t = tree[(size_t)stack.back()];
            stack.pop_back();
            MeshNode n;
            for (int a = 0; a < 3; ++a) { n.lo[a] = t.lo[a]; n.hi[a] = t.hi[a]; }
            n.tri = t.tri >= 0 ? triBase + t.tri : -1;
            const uint32_t next = (uint32_t)nodes.size() + (uint32_t)t.size;
            n.skip = next < end ? next : kMeshEnd;
            nodes.push_back(n);
            if (t.tri < 0) {
                const bool upperFirst = ((oct >> t.axis) & 1) != 0;       // the ray runs towards -axis: the upper half is nearer
                const int first = upperFirst ? t.right : t.left, second = upperFirst ? t.left : t.right;
                stack.push_back(second);
                stack.push_back(first);
            }
        }
    }
    return stride;
}

}  // namespace ptm
