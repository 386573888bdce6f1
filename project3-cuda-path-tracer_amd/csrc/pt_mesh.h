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
// equivalence with the brute-force rule needs.  Median split of the centroids along their widest axis; children in
// that order; depth-first layout with skip links.  `flat`: no hierarchy, one leaf per triangle in index order (tests:
// the brute-force rule on the device).  bbox receives the union of all leaf boxes (lo[3], hi[3]).
inline void appendMesh(const float *tris, int ntris, bool flat, std::vector<MeshNode> &nodes, std::vector<MeshTri> &out,
                       float bbox[6]) {
    const float m = meshMargin(tris, ntris);
    const uint32_t nodeBase = (uint32_t)nodes.size();
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
    auto leafNode = [&](const Leaf &L) {
        MeshNode n;
        for (int a = 0; a < 3; ++a) { n.lo[a] = L.lo[a]; n.hi[a] = L.hi[a]; }
        n.skip = 0;
        n.tri = triBase + L.idx;
        return n;
    };
    if (flat) {
        for (int i = 0; i < ntris; ++i) {
            MeshNode n = leafNode(leaves[(size_t)i]);
            n.skip = i + 1 < ntris ? nodeBase + (uint32_t)i + 1u : kMeshEnd;
            nodes.push_back(n);
        }
        return;
    }
    // depth-first build over index ranges of `leaves` (recursion depth: ceil(log2 ntris) + 1)
    struct Rec {
        std::vector<Leaf> &lv;
        std::vector<MeshNode> &nodes;
        decltype(leafNode) &mk;
        void build(int lo, int hi) {
            if (hi - lo == 1) { nodes.push_back(mk(lv[(size_t)lo])); return; }
            float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = lo; i < hi; ++i)
                for (int a = 0; a < 3; ++a) { cmin[a] = min2(cmin[a], lv[(size_t)i].c[a]); cmax[a] = max2(cmax[a], lv[(size_t)i].c[a]); }
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
            const int mid = lo + (hi - lo) / 2;
            std::nth_element(lv.begin() + lo, lv.begin() + mid, lv.begin() + hi, [axis](const Leaf &x, const Leaf &y) {
                return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.idx < y.idx);
            });
            const size_t me = nodes.size();
            MeshNode n;
            memset(&n, 0, sizeof n);
            n.tri = -1;
            nodes.push_back(n);
            const size_t left = nodes.size();
            build(lo, mid);
            const size_t right = nodes.size();
            build(mid, hi);
            for (int a = 0; a < 3; ++a) {
                nodes[me].lo[a] = min2(nodes[left].lo[a], nodes[right].lo[a]);
                nodes[me].hi[a] = max2(nodes[left].hi[a], nodes[right].hi[a]);
            }
        }
    } rec{leaves, nodes, leafNode};
    rec.build(0, ntris);
    // skip links: the node after a node's subtree.  Subtree sizes follow from the layout: a leaf is 1 node, an inner node
    // 1 + its two children's subtrees; one backwards pass.
    const uint32_t end = (uint32_t)nodes.size();
    std::vector<uint32_t> size(end - nodeBase);
    for (uint32_t i = end; i-- > nodeBase;) {
        if (nodes[i].tri >= 0) size[i - nodeBase] = 1;
        else {
            const uint32_t l = i + 1, r = l + size[l - nodeBase];
            size[i - nodeBase] = 1 + size[l - nodeBase] + size[r - nodeBase];
        }
        const uint32_t next = i + size[i - nodeBase];
        nodes[i].skip = next < end ? next : kMeshEnd;
    }
}

}  // namespace ptm
