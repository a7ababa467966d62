// okrt_fast_build.cpp — host-side derivation of the MI355X traversal structure from the
// reference's uploaded tables (okrt_fast.h).
//
// The reference traverses octree -> instance -> per-mesh BVH with no front-to-back order
// (RaytracerCS.hlsl:256-334). The closest hit it returns is min over {triangles reachable that
// way} of the world-space hit distance, so any structure over the same triangle set returns the
// same hit (ties aside). Here:
//   1. walk the octree from node 0 and collect the instances it reaches, walk each instance's BVH
//      and collect its leaf triangles (exactly the set the shader can test);
//   2. bake each instance's triangles into world space (identity instances bit for bit; others as
//      M*p — quirk Q3/Q22: the reference's local-space culling is only exact for unit scale);
//   3. build one binned-SAH BVH2 over all of them (16 bins x 3 axes, leaves <= 4 triangles),
//      in parallel over the top subtrees;
//   4. collapse to 4-wide nodes by repeatedly opening the child with the largest surface area and
//      lay nodes and triangles out in depth-first order (subtrees and their leaves contiguous).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>
#include <vector>

#include "okrt_fast.h"

namespace okrt {

namespace {

struct Box {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], p[k]);
            mx[k] = std::max(mx[k], p[k]);
        }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], b.mn[k]);
            mx[k] = std::max(mx[k], b.mx[k]);
        }
    }
    float area() const {
        float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        if (!(ex >= 0.f) || !(ey >= 0.f) || !(ez >= 0.f)) return 0.f;
        return ex * ey + ey * ez + ez * ex;
    }
};

struct Prim {
    Box box;
    float c[3];
    uint32_t srcTri, instance;
    float p[9];  // world-space p0,p1,p2
};

struct Node2 {
    Box box;
    int32_t left = -1, right = -1;  // children, or -1 for a leaf
    uint32_t first = 0, count = 0;  // leaf range in the index array
};

constexpr int kBins = 16;
// Leaves: at most kLeafSize triangles; below that a range becomes a leaf as soon as testing all its triangles is no
// dearer than the best split (SAH with a triangle test = 1, a BVH2 node = kNodeCost: the traversal kernel spends one
// iteration on either, and a 4-wide node absorbs two to three BVH2 nodes). Developer knobs.
static const uint32_t kLeafSize = getenv("OKRT_BUILD_LEAF") ? (uint32_t)atoi(getenv("OKRT_BUILD_LEAF")) : 6u;
static const float kNodeCost = getenv("OKRT_BUILD_NODE_COST") ? (float)atof(getenv("OKRT_BUILD_NODE_COST")) : 1.f;  // < 0: leaves whenever count <= kLeafSize

struct Builder {
    std::vector<Prim>& prims;
    std::vector<uint32_t> idx;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> nodeCount{0};

    explicit Builder(std::vector<Prim>& p) : prims(p) {
        idx.resize(p.size());
        for (uint32_t i = 0; i < p.size(); ++i) idx[i] = i;
        nodes.resize(std::max<size_t>(2 * p.size(), 1));
    }
    uint32_t alloc() { return nodeCount.fetch_add(1); }

    void build(uint32_t nodeIdx, uint32_t lo, uint32_t hi, int depth) {
        Node2& n = nodes[nodeIdx];
        Box bb, cb;
        for (uint32_t i = lo; i < hi; ++i) {
            bb.grow(prims[idx[i]].box);
            cb.grow(prims[idx[i]].c);
        }
        n.box = bb;
        uint32_t cnt = hi - lo;
        if (cnt <= 1 || (cnt <= kLeafSize && kNodeCost < 0.f)) {
            n.first = lo;
            n.count = cnt;
            return;
        }
        int bestAxis = -1, bestBin = -1;
        float bestCost = FLT_MAX;
        for (int axis = 0; axis < 3; ++axis) {
            float cmin = cb.mn[axis], cext = cb.mx[axis] - cb.mn[axis];
            if (!(cext > 0.f)) continue;
            float scale = kBins / cext;
            Box binBox[kBins];
            uint32_t binCnt[kBins] = {};
            for (uint32_t i = lo; i < hi; ++i) {
                const Prim& p = prims[idx[i]];
                int b = std::min(kBins - 1, std::max(0, (int)((p.c[axis] - cmin) * scale)));
                binCnt[b]++;
                binBox[b].grow(p.box);
            }
            float rightArea[kBins];
            uint32_t rightCnt[kBins];
            Box acc;
            uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(binBox[b]);
                c += binCnt[b];
                rightArea[b] = acc.area();
                rightCnt[b] = c;
            }
            Box accL;
            uint32_t cl = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                accL.grow(binBox[b]);
                cl += binCnt[b];
                if (!cl || !rightCnt[b + 1]) continue;
                float cost = accL.area() * cl + rightArea[b + 1] * rightCnt[b + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestBin = b;
                }
            }
        }
        if (cnt <= kLeafSize && (bestAxis < 0 || (float)cnt * bb.area() <= kNodeCost * bb.area() + bestCost)) {
            n.first = lo;
            n.count = cnt;
            return;
        }
        uint32_t mid;
        if (bestAxis < 0) {  // all centroids coincide: split the list in half
            mid = lo + cnt / 2;
        } else {
            float cmin = cb.mn[bestAxis], scale = kBins / (cb.mx[bestAxis] - cb.mn[bestAxis]);
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t t) {
                int b = std::min(kBins - 1, std::max(0, (int)((prims[t].c[bestAxis] - cmin) * scale)));
                return b <= bestBin;
            });
            mid = (uint32_t)(it - idx.begin());
            if (mid == lo || mid == hi) mid = lo + cnt / 2;
        }
        uint32_t l = alloc(), r = alloc();
        nodes[nodeIdx].left = (int32_t)l;
        nodes[nodeIdx].right = (int32_t)r;
        if (depth < 4 && cnt > 20000) {  // top of a large tree: build the halves on separate threads
            auto fut = std::async(std::launch::async, [this, l, lo, mid, depth] { build(l, lo, mid, depth + 1); });
            build(r, mid, hi, depth + 1);
            fut.get();
        } else {
            build(l, lo, mid, depth + 1);
            build(r, mid, hi, depth + 1);
        }
    }
};

bool isIdentity(const float* m) {
    static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 16; ++i)
        if (m[i] != I[i]) return false;
    return true;
}


// Quantise up to four child boxes onto a per-node grid. The grid origin is the union's lower corner
// minus a pad, the step a power of two with 255 steps covering the padded extent; planes are
// floor/ceil'ed, so every quantised box contains its child (checked in double below). The pad keeps
// planes that coincide with the node's own faces strictly inside the grid, which absorbs the few-ulp
// rounding of the kernel's fused slab arithmetic t = fma(q, step/d, (origin - o)/d).
Node4Q makeNode4Q(const Box* boxes, const uint32_t* refs, int n) {
    Node4Q nd;
    std::memset(&nd, 0, sizeof(nd));
    Box u;
    for (int i = 0; i < n; ++i) u.grow(boxes[i]);
    uint32_t exps = 0;
    double org[3], step[3];
    for (int a = 0; a < 3; ++a) {
        double lo = u.mn[a], hi = u.mx[a];
        double pad = std::max({(hi - lo) * (1.0 / 512.0), std::max(std::fabs(lo), std::fabs(hi)) * 4e-6, 1e-30});
        float o = (float)(lo - pad);
        if ((double)o > lo - pad) o = std::nextafterf(o, -INFINITY);
        double ext = (hi + pad) - (double)o;
        int k;
        std::frexp(ext / 255.0, &k);  // ext/255 = m * 2^k, m in [0.5, 1)  ->  2^k >= ext/255
        k = std::min(std::max(k, -126), 127);
        while ((double)o + 255.0 * std::ldexp(1.0, k) < hi && k < 127) ++k;
        nd.origin[a] = o;
        org[a] = o;
        step[a] = std::ldexp(1.0, k);
        exps |= (uint32_t)(k + 127) << (8 * a);
    }
    nd.exps = exps;
    for (int i = 0; i < 4; ++i) {
        if (i >= n) {
            nd.child[i] = kEmptyRef;
            for (int a = 0; a < 3; ++a) {  // inverted: can never be entered
                nd.qlo[a] |= 255u << (8 * i);
            }
            continue;
        }
        nd.child[i] = refs[i];
        for (int a = 0; a < 3; ++a) {
            double ql = std::floor((boxes[i].mn[a] - org[a]) / step[a]);
            double qh = std::ceil((boxes[i].mx[a] - org[a]) / step[a]);
            ql = std::min(std::max(ql, 0.0), 255.0);
            qh = std::min(std::max(qh, 0.0), 255.0);
            nd.qlo[a] |= (uint32_t)ql << (8 * i);
            nd.qhi[a] |= (uint32_t)qh << (8 * i);
        }
    }
    return nd;
}

struct Emitter {
    const Builder& b;
    FastHostScene& out;
    uint32_t maxDepth = 0;

    uint32_t emitLeaf(const Node2& n) {
        uint32_t first = (uint32_t)out.tris.size();
        for (uint32_t i = 0; i < n.count; ++i) {
            const Prim& p = b.prims[b.idx[n.first + i]];
            Tri48 t{};
            for (int k = 0; k < 3; ++k) {
                t.p2[k] = p.p[6 + k];
                t.e1[k] = p.p[k] - p.p[6 + k];      // E1 = p0 - p2   (GPU-Utilities.hlsli:124)
                t.e2[k] = p.p[3 + k] - p.p[6 + k];  // E2 = p1 - p2
            }
            t.srcTri = p.srcTri;
            t.instance = p.instance;
            out.tris.push_back(t);
        }
        return kLeafBit | ((n.count - 1) << 27) | first;
    }

    // ---- which BVH2 nodes become the children of a wide node -----------------------------------------------------------
    // Greedy (rounds 1-2): open the inner child with the largest area until there are four. SAH-optimal (round 3, after
    // Ylitie, Karras, Laine 2017, for 4-wide nodes): cost[n][i] = the cheapest way to represent BVH2 subtree n as a FOREST of
    // at most i wide-node roots / leaves, with a wide node costing area x kWideNodeCost and a leaf area x triangles x
    // kWideTriCost; cost[n][1] roots a wide node at n whose (at most four) children are the forests of its two halves.
    static constexpr float kInfCost = 3.0e38f;
    std::vector<float> cost;       // [node][i], i = 1..3 at [4 * node + i]
    std::vector<uint8_t> choice;   // [4 * node + i]: i = 1: roots the left half gets (1..3); i > 1: 0 = as for i - 1, else roots for the left half
    bool optimal = false;
    mutable bool broken = false;  // forest() met a collapse table that would have dropped a subtree: the build is refused (fastBuild)

    void solve(int32_t root) {
        static const float kWideNodeCost = getenv("OKRT_BUILD_WIDE_NODE_COST") ? (float)atof(getenv("OKRT_BUILD_WIDE_NODE_COST")) : 1.0f;
        static const float kWideTriCost = getenv("OKRT_BUILD_WIDE_TRI_COST") ? (float)atof(getenv("OKRT_BUILD_WIDE_TRI_COST")) : 0.6f;
        cost.assign(4 * b.nodes.size(), kInfCost);
        choice.assign(4 * b.nodes.size(), 0);
        // post-order without recursion: a node is finished when both children are
        std::vector<std::pair<int32_t, bool>> stack{{root, false}};
        while (!stack.empty()) {
            const auto [n, expanded] = stack.back();
            stack.pop_back();
            const Node2& nd = b.nodes[n];
            float* c = &cost[4 * (size_t)n];
            uint8_t* ch = &choice[4 * (size_t)n];
            if (nd.left < 0) {
                c[1] = c[2] = c[3] = nd.box.area() * (float)nd.count * kWideTriCost;
                continue;
            }
            if (!expanded) {
                stack.push_back({n, true});
                stack.push_back({nd.left, false});
                stack.push_back({nd.right, false});
                continue;
            }
            const float* l = &cost[4 * (size_t)nd.left];
            const float* r = &cost[4 * (size_t)nd.right];
            float best = kInfCost;
            ch[1] = 2;  // (boxes with extents around 1e19 overflow every sum to inf: no candidate is < best, and forest() must still get a split in 1..3)
            for (int i = 1; i <= 3; ++i) {  // a wide node here: i children from the left half, 4 - i from the right
                const float v = std::min(l[i] + r[4 - i], kInfCost);
                if (v < best) {
                    best = v;
                    ch[1] = (uint8_t)i;
                }
            }
            c[1] = nd.box.area() * kWideNodeCost + best;
            for (int i = 2; i <= 3; ++i) {  // a forest of at most i roots: no better than i - 1, or split between the halves
                c[i] = c[i - 1];
                ch[i] = 0;
                for (int k = 1; k < i; ++k) {
                    const float v = l[k] + r[i - k];
                    if (v < c[i]) {
                        c[i] = v;
                        ch[i] = (uint8_t)k;
                    }
                }
            }
        }
        optimal = true;
    }
    // the roots of the cheapest forest of at most i roots for subtree n
    void forest(int32_t n, int i, int32_t* out, int& count) const {
        const Node2& nd = b.nodes[n];
        if (nd.left < 0 || i == 1) {
            out[count++] = n;
            return;
        }
        // cannot happen with the tables solve() fills (i in 2..3 here, 0 <= choice < i, at most four roots in all): a malformed table
        // must neither index outside its node's four entries, nor recurse for ever, nor overrun kids[4] — and must not LOSE a subtree
        // either: the whole of n then goes out as one root while there is room, and the build is marked broken when there is none
        if (i < 2 || i > 3 || count >= 4) {
            if (count < 4) out[count++] = n;
            else broken = true;
            return;
        }
        const uint8_t k = choice[4 * (size_t)n + i];
        if (k >= i) {
            out[count++] = n;
            return;
        }
        if (k == 0) forest(n, i - 1, out, count);
        else {
            forest(nd.left, k, out, count);
            forest(nd.right, i - k, out, count);
        }
    }

    // Emits the Node4 for BVH2 node `root` at out.nodes[slot].
    void emitNode(uint32_t slot, int32_t root, uint32_t depth) {
        maxDepth = std::max(maxDepth, depth);
        int32_t kids[4] = {b.nodes[root].left, b.nodes[root].right, -1, -1};
        int n = 2;
        if (optimal) {
            n = 0;
            const int k = choice[4 * (size_t)root + 1];
            forest(b.nodes[root].left, k, kids, n);
            forest(b.nodes[root].right, 4 - k, kids, n);
        } else
        while (n < 4) {  // open the inner child with the largest area
            int pick = -1;
            float best = -1.f;
            for (int i = 0; i < n; ++i) {
                const Node2& c = b.nodes[kids[i]];
                if (c.left >= 0 && c.box.area() > best) {
                    best = c.box.area();
                    pick = i;
                }
            }
            if (pick < 0) break;
            const Node2& c = b.nodes[kids[pick]];
            kids[pick] = c.left;
            kids[n++] = c.right;
        }
        Box boxes[4];
        uint32_t refs[4] = {kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef};
        uint32_t childSlot[4];
        for (int i = 0; i < n; ++i) {
            const Node2& c = b.nodes[kids[i]];
            boxes[i] = c.box;
            if (c.left < 0) {
                refs[i] = emitLeaf(c);
                maxDepth = std::max(maxDepth, depth + 1);
            } else {
                childSlot[i] = (uint32_t)out.nodes.size();
                out.nodes.emplace_back();
                refs[i] = childSlot[i];
            }
        }
        Node4Q nd = makeNode4Q(boxes, refs, n);
        out.nodes[slot] = nd;
        for (int i = 0; i < n; ++i)
            if (b.nodes[kids[i]].left >= 0) emitNode(childSlot[i], kids[i], depth + 1);
    }
};

}  // namespace

void buildFastHost(const FastInputs& in, FastHostScene& out) {
    out.nodes.clear();
    out.tris.clear();
    out.maxDepth = 0;
    out.allIdentity = true;

    // 1. instances reachable from octree node 0, in first-visit order
    std::vector<uint32_t> instances;
    if (in.numOct) {
        std::vector<uint8_t> seen(in.numMeshes, 0);
        std::vector<uint32_t> stack{0};
        while (!stack.empty()) {
            const OctNodeRec& n = in.oct[stack.back()];
            stack.pop_back();
            for (uint32_t i = n.meshesStart; i < n.meshesEnd && i < in.numMeshes; ++i)
                if (!seen[i]) {
                    seen[i] = 1;
                    instances.push_back(i);
                }
            for (uint32_t c = 0; c < n.numChildren; ++c) stack.push_back(n.firstChild + c);
        }
    }

    // 2. their leaf triangles, baked to world space
    std::vector<Prim> prims;
    for (uint32_t inst : instances) {
        const MeshRec& m = in.meshes[inst];
        const bool ident = isIdentity(m.M);
        out.allIdentity = out.allIdentity && ident;
        std::vector<uint32_t> stack{m.bvhNodeStartIdx};
        while (!stack.empty()) {
            const BvhNodeRec& n = in.bvh[stack.back()];
            stack.pop_back();
            if (n.firstChild != kInvalid) {
                stack.push_back(n.firstChild);
                stack.push_back(n.firstChild + 1);
                continue;
            }
            if (n.triStart == kInvalid) continue;
            for (uint32_t j = n.triStart; j < n.triEnd; ++j) {
                Prim p;
                const float* src = in.triPos[j].p;
                for (int v = 0; v < 3; ++v) {
                    const float* q = src + 3 * v;
                    if (ident) {
                        p.p[3 * v] = q[0]; p.p[3 * v + 1] = q[1]; p.p[3 * v + 2] = q[2];
                    } else {  // mul(float4(q,1), M): component r runs along memory row r
                        for (int r = 0; r < 3; ++r)
                            p.p[3 * v + r] = std::fmaf(m.M[4 * r + 2], q[2], std::fmaf(m.M[4 * r + 1], q[1], m.M[4 * r] * q[0])) + m.M[4 * r + 3];
                    }
                    p.box.grow(&p.p[3 * v]);
                }
                for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.box.mn[k] + p.box.mx[k]);
                bool finite = true;
                for (float f : p.p) finite = finite && std::isfinite(f);
                if (!finite) continue;  // can never be hit (every comparison with NaN/inf fails)
                p.srcTri = j;
                p.instance = inst;
                prims.push_back(p);
            }
        }
    }

    out.nodes.emplace_back();
    out.nodes[0] = makeNode4Q(nullptr, nullptr, 0);  // all-empty root
    if (prims.empty()) return;

    // 3. BVH2
    Builder b(prims);
    uint32_t r = b.alloc();
    b.build(r, 0, (uint32_t)prims.size(), 0);

    // 4. collapse + layout
    Emitter e{b, out};
    out.tris.reserve(prims.size());
    if (b.nodes[r].left < 0) {  // a single leaf: hang it under the root
        uint32_t ref = e.emitLeaf(b.nodes[r]);
        out.nodes[0] = makeNode4Q(&b.nodes[r].box, &ref, 1);
        out.maxDepth = 1;
        return;
    }
    static const bool kOptimalCollapse = !getenv("OKRT_BUILD_COLLAPSE") || atoi(getenv("OKRT_BUILD_COLLAPSE")) != 0;  // 0 = greedy (rounds 1-2)
    if (kOptimalCollapse) e.solve((int32_t)r);
    e.emitNode(0, (int32_t)r, 1);
    out.maxDepth = e.maxDepth;
    out.consistent = !e.broken;
}

// ---- top of the tree for the device builder ---------------------------------------------------------------------------
// PLOC clusters bottom-up inside a Morton window and never sees the large occluders of an architectural scene; a
// top-down SAH does. The device builder therefore stops clustering at a few thousand clusters and hands their boxes
// and triangle counts to this routine, which builds the upper levels by binned SAH weighted with the counts.
namespace {
struct TopItem { Box box; float c[3]; uint32_t weight, id; };
uint32_t buildTopRange(std::vector<TopItem>& items, uint32_t lo, uint32_t hi, uint32_t n, std::vector<TopNode>& out) {
    if (hi - lo == 1) return items[lo].id;  // a cluster
    Box bb, cb;
    uint32_t total = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        bb.grow(items[i].box);
        cb.grow(items[i].c);
        total += items[i].weight;
    }
    int bestAxis = -1, bestBin = -1;
    float bestCost = FLT_MAX;
    for (int axis = 0; axis < 3; ++axis) {
        const float cmin = cb.mn[axis], cext = cb.mx[axis] - cb.mn[axis];
        if (!(cext > 0.f)) continue;
        const float scale = kBins / cext;
        Box binBox[kBins];
        uint32_t binW[kBins] = {}, binN[kBins] = {};
        for (uint32_t i = lo; i < hi; ++i) {
            const int b = std::min(kBins - 1, std::max(0, (int)((items[i].c[axis] - cmin) * scale)));
            binW[b] += items[i].weight;
            binN[b]++;
            binBox[b].grow(items[i].box);
        }
        float rightArea[kBins];
        uint32_t rightW[kBins], rightN[kBins];
        Box acc;
        uint32_t w = 0, c = 0;
        for (int b = kBins - 1; b > 0; --b) {
            acc.grow(binBox[b]);
            w += binW[b];
            c += binN[b];
            rightArea[b] = acc.area();
            rightW[b] = w;
            rightN[b] = c;
        }
        Box accL;
        uint32_t wl = 0, cl = 0;
        for (int b = 0; b < kBins - 1; ++b) {
            accL.grow(binBox[b]);
            wl += binW[b];
            cl += binN[b];
            if (!cl || !rightN[b + 1]) continue;
            const float cost = accL.area() * (float)wl + rightArea[b + 1] * (float)rightW[b + 1];
            if (cost < bestCost) {
                bestCost = cost;
                bestAxis = axis;
                bestBin = b;
            }
        }
    }
    uint32_t mid = lo + (hi - lo) / 2;
    if (bestAxis >= 0) {
        const float cmin = cb.mn[bestAxis], scale = kBins / (cb.mx[bestAxis] - cb.mn[bestAxis]);
        auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const TopItem& t) {
            return std::min(kBins - 1, std::max(0, (int)((t.c[bestAxis] - cmin) * scale))) <= bestBin;
        });
        const uint32_t m = (uint32_t)(it - items.begin());
        if (m != lo && m != hi) mid = m;
    }
    const uint32_t self = (uint32_t)out.size();
    out.emplace_back();
    const uint32_t l = buildTopRange(items, lo, mid, n, out), r = buildTopRange(items, mid, hi, n, out);
    TopNode& t = out[self];
    for (int k = 0; k < 3; ++k) { t.mn[k] = bb.mn[k]; t.mx[k] = bb.mx[k]; }
    t.left = l;
    t.right = r;
    t.count = total;
    return n + self;
}
}  // namespace

uint32_t buildTopSAH(const float* boxes6, const uint32_t* weights, uint32_t n, std::vector<TopNode>& out) {
    std::vector<TopItem> items(n);
    for (uint32_t i = 0; i < n; ++i) {
        TopItem& t = items[i];
        for (int k = 0; k < 3; ++k) {
            t.box.mn[k] = boxes6[6 * i + k];
            t.box.mx[k] = boxes6[6 * i + 3 + k];
            t.c[k] = 0.5f * t.box.mn[k] + 0.5f * t.box.mx[k];
        }
        t.weight = weights[i];
        t.id = i;
    }
    out.clear();
    out.reserve(n);
    return buildTopRange(items, 0, n, n, out);
}

}  // namespace okrt

// ---- host-side check of the structure (no GPU): used by tests/test_fast_structure.py ----------------
// Builds the structure from raw reference tables and traces rays through it on the CPU with the
// same node/leaf decoding as the kernel. Not used by any render path.
extern "C" int okrt_debug_fast_closest_hit(const void* triPos, uint32_t numTris, const void* bvh, uint32_t numNodes, const void* oct,
                                           uint32_t numOct, const void* meshes, uint32_t numMeshes, uint32_t numRays,
                                           const float* origins, const float* dirs, float* outDist, uint32_t* outTri,
                                           uint32_t* outInstance, uint32_t* outStats) {
    using namespace okrt;
    FastInputs in{(const TriangleRec*)triPos, numTris, (const BvhNodeRec*)bvh, numNodes, (const OctNodeRec*)oct, numOct, (const MeshRec*)meshes, numMeshes};
    FastHostScene fs;
    buildFastHost(in, fs);
    if (outStats) {
        outStats[0] = (uint32_t)fs.nodes.size();
        outStats[1] = (uint32_t)fs.tris.size();
        outStats[2] = fs.maxDepth;
        outStats[3] = fs.allIdentity ? 1u : 0u;
    }
    auto dot = [](const float* a, const float* b) { return std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])); };
    auto cross = [](const float* a, const float* b, float* r) {
        r[0] = std::fmaf(a[1], b[2], -(a[2] * b[1]));
        r[1] = std::fmaf(a[2], b[0], -(a[0] * b[2]));
        r[2] = std::fmaf(a[0], b[1], -(a[1] * b[0]));
    };
    for (uint32_t r = 0; r < numRays; ++r) {
        const float* o = origins + 3 * r;
        float d[3] = {dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]};
        float rs = 1.0f / std::sqrt(dot(d, d));
        for (float& x : d) x *= rs;
        float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
        float best = FLT_MAX;
        uint32_t bt = kInvalid, bi = kInvalid;
        std::vector<uint32_t> stack{0};
        while (!stack.empty()) {
            uint32_t cur = stack.back();
            stack.pop_back();
            if (cur == kEmptyRef) continue;
            if (cur & kLeafBit) {
                uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 15u) + 1u;
                for (uint32_t j = 0; j < count; ++j) {
                    const Tri48& t = fs.tris[first + j];
                    float c1[3], c2[3], mv[3] = {o[0] - t.p2[0], o[1] - t.p2[1], o[2] - t.p2[2]};
                    cross(d, t.e2, c1);
                    float det = dot(t.e1, c1);
                    if (det < 0.000001f && det > -0.000001f) continue;
                    float id = 1.f / det;
                    float u = dot(mv, c1) * id;
                    if (u < -0.000001f) continue;
                    cross(mv, t.e1, c2);
                    float v = dot(d, c2) * id;
                    if (v < -0.000001f || u + v > 1.0f) continue;
                    float tt = dot(t.e2, c2) * id;
                    if (tt < -0.000001f || !(tt > 0.f)) continue;
                    float w[3] = {d[0] * tt, d[1] * tt, d[2] * tt};
                    float dist = std::sqrt(dot(w, w));
                    if (dist < best) {
                        best = dist;
                        bt = t.srcTri;
                        bi = t.instance;
                    }
                }
                continue;
            }
            const Node4Q& n = fs.nodes[cur];
            for (int k = 0; k < 4; ++k) {
                if (n.child[k] == kEmptyRef) continue;
                float tn = -FLT_MAX, tf = FLT_MAX;
                for (int a = 0; a < 3; ++a) {  // the kernel's fused slab arithmetic
                    uint32_t eb = (n.exps >> (8 * a)) & 255u, bits = eb << 23;
                    float sc;
                    std::memcpy(&sc, &bits, 4);
                    float iq = std::fmin(std::fmax(inv[a], -1e30f), 1e30f);
                    float A = sc * iq, B = (n.origin[a] - o[a]) * iq;
                    float ql = (float)((n.qlo[a] >> (8 * k)) & 255u), qh = (float)((n.qhi[a] >> (8 * k)) & 255u);
                    float tnear = std::fmaf(iq >= 0.f ? ql : qh, A, B), tfar = std::fmaf(iq >= 0.f ? qh : ql, A, B);
                    tn = std::fmax(tn, tnear);
                    tf = std::fmin(tf, tfar);
                }
                if (tf >= tn && tf > 0.f && tn < best) stack.push_back(n.child[k]);
            }
        }
        outDist[r] = best;
        outTri[r] = bt;
        outInstance[r] = bi;
    }
    return 0;
}
