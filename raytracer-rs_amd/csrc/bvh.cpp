// bvh.cpp — binned-SAH BVH2 builder, see bvh.hpp.
#include "bvh.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <queue>
#include <cstdlib>

namespace mi355rt {
namespace {

struct Box {
    float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    float half_area() const
    {
        float d[3] = { mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2] };
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0f;
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};

struct TmpNode { Box box; int32_t left = -1, right = -1; uint32_t first = 0, count = 0; uint32_t depth = 0; };

static uint32_t g_max_leaf = kBvhMaxLeaf;
static float g_trav_cost = 0.5f;

struct Builder {
    const float* verts;
    std::vector<Box> tri_box;
    std::vector<float> centroid;     // ntri*3
    std::vector<uint32_t> order;
    std::vector<TmpNode> tmp;

    static uint32_t levels_needed(uint32_t n)
    {
        uint32_t l = 0;
        while ((g_max_leaf << l) < n) ++l;
        return l;
    }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth)
    {
        TmpNode node;
        node.first = first; node.count = count; node.depth = depth;
        Box cbox;
        for (uint32_t i = first; i < first + count; ++i) { node.box.grow(tri_box[order[i]]); cbox.grow(&centroid[3 * order[i]]); }
        int32_t idx = (int32_t)tmp.size();
        tmp.push_back(node);
        if (count <= 1) return idx;

        constexpr int NB = 16;
        float best_cost = 3.4e38f; int best_axis = -1, best_bin = -1;
        bool force_balanced = depth + levels_needed(count) + 1 >= kBvhMaxDepth;
        if (!force_balanced) {
            for (int a = 0; a < 3; ++a) {
                float lo = cbox.mn[a], hi = cbox.mx[a];
                if (!(hi > lo)) continue;
                float scale = NB / (hi - lo);
                Box bb[NB]; uint32_t bc[NB] = { 0 };
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = std::min(NB - 1, (int)((centroid[3 * order[i] + a] - lo) * scale));
                    bb[b].grow(tri_box[order[i]]); bc[b]++;
                }
                float right_area[NB]; uint32_t right_cnt[NB];
                Box acc; uint32_t c = 0;
                for (int b = NB - 1; b > 0; --b) { acc.grow(bb[b]); c += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = c; }
                Box lacc; uint32_t lc = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    lacc.grow(bb[b]); lc += bc[b];
                    if (lc == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = lacc.half_area() * lc + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        if (count <= g_max_leaf) {
            // leaf unless the split is clearly cheaper (traversal step ~ 1.5 triangle tests)
            float leaf_cost = node.box.half_area() * (float)count;
            float split_cost = best_axis >= 0 ? best_cost + g_trav_cost * node.box.half_area() : 3.4e38f;
            if (!(split_cost < leaf_cost)) return idx;
        }
        uint32_t mid;
        if (best_axis >= 0) {
            float lo = cbox.mn[best_axis], hi = cbox.mx[best_axis];
            float scale = NB / (hi - lo);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = std::min(NB - 1, (int)((centroid[3 * t + best_axis] - lo) * scale));
                return b <= best_bin;
            });
            mid = (uint32_t)(it - order.begin());
        } else {
            // no usable SAH split (coincident centroids or depth budget): median along the widest axis
            int a = 0;
            float d[3] = { cbox.mx[0] - cbox.mn[0], cbox.mx[1] - cbox.mn[1], cbox.mx[2] - cbox.mn[2] };
            if (d[1] > d[a]) a = 1;
            if (d[2] > d[a]) a = 2;
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](uint32_t x, uint32_t y) { return centroid[3 * x + a] < centroid[3 * y + a]; });
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        int32_t l = build(first, mid - first, depth + 1);
        int32_t r = build(mid, first + count - mid, depth + 1);
        tmp[idx].left = l; tmp[idx].right = r;
        return idx;
    }
};

// ---- insertion-based optimisation of the finished binary tree (after Bittner, Hapala, Havran: "Fast insertion-based optimization of
// bounding volume hierarchies", 2013).  The trace kernels are bound by the cache lines their node fetches touch (profiles/r03_notes.md),
// i.e. by the number of nodes a ray visits, and the top-down binned build leaves overlap that a local search removes: every subtree N is
// cut out (its parent is spliced out) and put back where the inner-node surface of the whole tree grows least — a branch-and-bound search
// from the root, bounded by the surface the insertion costs the ancestors.  Putting N back beside its old sibling is always a candidate,
// so a move never makes the tree worse.  The deepest leaf may not get deeper than it was (the traversal stack is sized for it).
struct TreeOptimizer {
    std::vector<TmpNode>& t;
    std::vector<int32_t> parent;
    std::vector<uint8_t> height;          // of the subtree: 0 for a leaf
    int32_t root;
    uint32_t dmax;                        // deepest leaf allowed

    static float area(const Box& b) { return b.half_area(); }
    static Box merged(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }

    TreeOptimizer(std::vector<TmpNode>& tmp, int32_t r) : t(tmp), parent(tmp.size(), -1), height(tmp.size(), 0), root(r), dmax(0)
    {
        // children follow their parent in the array (the builder appends depth-first): one backward sweep gives the heights
        for (size_t i = 0; i < t.size(); ++i) if (t[i].left >= 0) { parent[t[i].left] = (int32_t)i; parent[t[i].right] = (int32_t)i; }
        for (size_t i = t.size(); i-- > 0;) {
            if (t[i].left >= 0) height[i] = (uint8_t)(1 + std::max(height[t[i].left], height[t[i].right]));
            else dmax = std::max(dmax, t[i].depth);
        }
    }
    void refit_up(int32_t n)
    {
        for (; n >= 0; n = parent[n]) {
            t[n].box = merged(t[t[n].left].box, t[t[n].right].box);
            height[n] = (uint8_t)(1 + std::max(height[t[n].left], height[t[n].right]));
        }
    }
    double inner_area() const { double a = 0; for (const TmpNode& n : t) if (n.left >= 0) a += area(n.box); return a; }

    bool reinsert(int32_t N)
    {
        const int32_t P = parent[N];
        if (P < 0) return false;
        const int32_t G = parent[P];
        if (G < 0) return false;                                  // children of the root stay
        const int32_t S = t[P].left == N ? t[P].right : t[P].left;
        // cut N out: the sibling takes the parent's place
        (t[G].left == P ? t[G].left : t[G].right) = S; parent[S] = G;
        refit_up(G);
        // branch and bound for the node X beside which N costs least
        struct Item { float induced; int32_t node; uint32_t depth; bool operator<(const Item& o) const { return induced > o.induced; } };
        std::priority_queue<Item> q;
        q.push(Item{ 0.0f, root, 0u });
        const float area_n = area(t[N].box);
        float best = 3.4e38f; int32_t best_x = S;
        while (!q.empty()) {
            const Item it = q.top(); q.pop();
            if (it.induced + area_n >= best) break;               // nothing left in the queue can beat the best
            const int32_t X = it.node;
            const float direct = area(merged(t[X].box, t[N].box));
            const float total = it.induced + direct;
            if (total < best && it.depth + 1u + std::max(height[X], height[N]) <= dmax) { best = total; best_x = X; }
            if (t[X].left >= 0) {
                const float child_induced = total - area(t[X].box);          // what X itself grows by if N goes below it
                if (child_induced + area_n < best) { q.push(Item{ child_induced, t[X].left, it.depth + 1u }); q.push(Item{ child_induced, t[X].right, it.depth + 1u }); }
            }
        }
        // put N back beside best_x, re-using P as their parent
        const int32_t X = best_x, Q = parent[X];
        t[P].left = X; t[P].right = N; parent[X] = P; parent[N] = P; parent[P] = Q;
        if (Q >= 0) (t[Q].left == X ? t[Q].left : t[Q].right) = P; else root = P;
        refit_up(P);
        return X != S;
    }
    // depth of every node again (the flattening reads TmpNode::depth of the leaves)
    void assign_depths()
    {
        std::vector<int32_t> stack(1, root);
        t[root].depth = 0;
        while (!stack.empty()) {
            const int32_t n = stack.back(); stack.pop_back();
            if (t[n].left < 0) continue;
            t[t[n].left].depth = t[n].depth + 1; t[t[n].right].depth = t[n].depth + 1;
            stack.push_back(t[n].left); stack.push_back(t[n].right);
        }
    }
    int32_t run(int passes)
    {
        std::vector<int32_t> cand;
        for (int pass = 0; pass < passes; ++pass) {
            cand.clear();
            for (size_t i = 0; i < t.size(); ++i) if ((int32_t)i != root && parent[i] >= 0 && parent[parent[i]] >= 0) cand.push_back((int32_t)i);
            std::sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { const float x = area(t[a].box), y = area(t[b].box); return x != y ? x > y : a < b; });   // large subtrees first
            size_t moved = 0;
            for (int32_t n : cand) if (parent[n] >= 0 && parent[parent[n]] >= 0) moved += reinsert(n) ? 1u : 0u;
            if (moved * 200 < cand.size()) break;                  // < 0.5 % of the nodes moved: converged
        }
        assign_depths();
        return root;
    }
};

int32_t leaf_code(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

// float -> binary16 bit pattern, rounded toward -inf (up == false) or +inf (up == true)
uint32_t half_bits_directed(float x, bool up)
{
    if (x != x) return up ? 0x7C00u : 0xFC00u;                 // NaN: widest possible bound
    _Float16 h = (_Float16)x;                                   // round to nearest
    uint16_t b;
    std::memcpy(&b, &h, 2);
    const float back = (float)h;
    if (up ? back < x : back > x) {
        // step one representable value in the wanted direction
        if ((b & 0x7FFFu) == 0) b = up ? 0x0001u : 0x8001u;    // +-0 -> smallest denormal of the right sign
        else if (((b & 0x8000u) == 0) == up) b += 1;           // moving away from zero
        else b -= 1;                                           // moving toward zero
    }
    return b;
}
// experiment knob (profiles/r02_notes.md): MI355RT_BOX_EXTRA_ULPS=n moves every bound n more half-precision steps outward —
// how much do the node visits per ray depend on the quantisation of the boxes?
uint32_t step_outward(uint32_t b, bool up, int n)
{
    for (int i = 0; i < n; ++i) {
        if ((b & 0x7FFFu) >= 0x7BFFu) break;
        if ((b & 0x7FFFu) == 0) b = up ? 0x0001u : 0x8001u;
        else if (((b & 0x8000u) == 0) == up) b += 1;
        else b -= 1;
    }
    return b;
}
void pack_box(const Box& box, float pad, uint32_t out[3])
{
    static const int extra = [] { const char* e = getenv("MI355RT_BOX_EXTRA_ULPS"); return e ? atoi(e) : 0; }();
    for (int a = 0; a < 3; ++a)
        out[a] = step_outward(half_bits_directed(box.mn[a] - pad, false), false, extra) | (step_outward(half_bits_directed(box.mx[a] + pad, true), true, extra) << 16);
}

}  // namespace

void build_bvh(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri, Bvh& out)
{
    out = Bvh();
    g_max_leaf = kBvhMaxLeaf; g_trav_cost = 0.5f;   // measured: smaller leaves win (profiles/r01_notes.md)
    if (const char* e = std::getenv("MI355RT_MAX_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= 8) g_max_leaf = (uint32_t)v; }
    if (const char* e = std::getenv("MI355RT_TRAV_COST")) { float v = (float)std::atof(e); if (v >= 0.0f) g_trav_cost = v; }
    if (ntri == 0) {
        // one empty leaf is not representable (count >= 1): a degenerate triangle that can never
        // be hit (|det| < EPSILON) keeps the kernels free of an "empty scene" special case
        BvhTri t; std::memset(&t, 0, sizeof t); t.prim = 0xFFFFFFFFu;
        out.tris.push_back(t);
        out.root = leaf_code(0, 1);
        out.leaves = 1; out.max_leaf = 1;
        out.nodes.resize(1);
        std::memset(out.nodes.data(), 0, sizeof(BvhNode));
        return;
    }
    Builder b;
    b.verts = tri_verts;
    b.tri_box.resize(ntri); b.centroid.resize((size_t)ntri * 3); b.order.resize(ntri);
    Box scene;
    for (uint32_t t = 0; t < ntri; ++t) {
        for (int v = 0; v < 3; ++v) b.tri_box[t].grow(&tri_verts[9 * t + 3 * v]);
        for (int a = 0; a < 3; ++a) b.centroid[3 * t + a] = 0.5f * (b.tri_box[t].mn[a] + b.tri_box[t].mx[a]);
        b.order[t] = t;
        scene.grow(b.tri_box[t]);
    }
    std::memcpy(out.scene_min, scene.mn, 12); std::memcpy(out.scene_max, scene.mx, 12);
    b.tmp.reserve(2 * (size_t)ntri);
    int32_t root = b.build(0, ntri, 0);
    {   // MI355RT_BVH_OPT = passes of the insertion-based optimisation (0: the tree as built).  Measured on thai2 (profiles/r03_notes.md): 0 / 1 / 2 / 4
        // passes -> inner-node surface 100 / 94.3 / 92.8 / 92.3 %, 18.93 / 18.90 / 18.70 / 18.42 node visits per traced ray, 21.68 / 21.52 / 21.24 /
        // 21.05 ms per frame, 10 / 46 / 82 / 151 ms of build time inside create.  A pass costs ~1 us per node on one host thread, so large scenes get fewer.
        int passes = ntri <= 32768u ? 4 : ntri <= 65536u ? 2 : ntri <= 131072u ? 1 : 0;
        if (const char* e = std::getenv("MI355RT_BVH_OPT")) passes = std::max(0, std::min(8, std::atoi(e)));
        if (passes > 0 && b.tmp[root].left >= 0) {
            TreeOptimizer opt(b.tmp, root);
            const double before = opt.inner_area();
            root = opt.run(passes);
            if (std::getenv("MI355RT_DEBUG_BVH")) fprintf(stderr, "[mi355rt] BVH optimisation: inner-node surface %.6g -> %.6g (%.1f %%), deepest leaf <= %u\n", before, opt.inner_area(), 100.0 * opt.inner_area() / before, opt.dmax);
        }
    }

    // Conservative padding: the exact (unfused f32) triangle test decides hits; a box test must
    // never reject a ray the triangle test would accept, so every box is grown by `pad`.
    float diag = std::sqrt((scene.mx[0] - scene.mn[0]) * (scene.mx[0] - scene.mn[0]) + (scene.mx[1] - scene.mn[1]) * (scene.mx[1] - scene.mn[1])
                           + (scene.mx[2] - scene.mn[2]) * (scene.mx[2] - scene.mn[2]));
    float pad = std::max(diag * 2e-5f, 1e-6f);

    // Numbering of the inner nodes = their place in memory: breadth-first (the top of the tree has the lowest indices).
    // MI355RT_NODE_LAYOUT=area numbers them by descending box surface instead (a priority queue from the root: the first N nodes are
    // the N a ray most probably visits).  Measured for the trace kernels (profiles/r03_notes.md): by surface +0.2 ms per frame, a treelet
    // layout (parent / children / grandchild in one 128-byte line) +-0 — which lines a fetch touches is not what bounds them.
    std::vector<int32_t> inner_index(b.tmp.size(), -1);
    std::vector<int32_t> bfs;                       // layout order: tmp-node per slot
    const char* layout_env = std::getenv("MI355RT_NODE_LAYOUT");
    const bool by_area = layout_env && std::strcmp(layout_env, "area") == 0;
    if (b.tmp[root].left >= 0 && !by_area) {
        std::queue<int32_t> q; q.push(root);
        while (!q.empty()) {
            int32_t n = q.front(); q.pop();
            inner_index[n] = (int32_t)bfs.size(); bfs.push_back(n);
            for (int32_t c : { b.tmp[n].left, b.tmp[n].right }) if (b.tmp[c].left >= 0) q.push(c);
        }
    } else if (b.tmp[root].left >= 0) {
        auto surface = [&](int32_t n) { const Box& x = b.tmp[n].box; const float dx = x.mx[0] - x.mn[0], dy = x.mx[1] - x.mn[1], dz = x.mx[2] - x.mn[2]; return dx * dy + dy * dz + dz * dx; };
        typedef std::pair<float, int32_t> Item;                           // (surface, -node): ties in creation order
        std::priority_queue<Item> q;
        q.push(Item(surface(root), -root));
        while (!q.empty()) {
            const int32_t n = -q.top().second; q.pop();
            inner_index[n] = (int32_t)bfs.size(); bfs.push_back(n);
            for (int32_t c : { b.tmp[n].left, b.tmp[n].right }) if (b.tmp[c].left >= 0) q.push(Item(surface(c), -c));
        }
    }
    auto emit_leaf = [&](const TmpNode& n) -> int32_t {
        uint32_t first = (uint32_t)out.tris.size();
        for (uint32_t i = n.first; i < n.first + n.count; ++i) {
            uint32_t t = b.order[i];
            const float* v = &tri_verts[9 * t];
            BvhTri r; std::memset(&r, 0, sizeof r);
            for (int a = 0; a < 3; ++a) { r.v0[a] = v[a]; r.e1[a] = v[3 + a] - v[a]; r.e2[a] = v[6 + a] - v[a]; }
            r.prim = t; r.geom = tri_geom[t];
            out.tris.push_back(r);
        }
        out.leaves++; out.max_leaf = std::max(out.max_leaf, n.count); out.max_depth = std::max(out.max_depth, n.depth);
        return leaf_code(first, n.count);
    };
    if (bfs.empty()) {
        out.root = emit_leaf(b.tmp[root]);
        out.nodes.resize(1);                      // never visited: keeps predicated node fetches in bounds
        std::memset(out.nodes.data(), 0, sizeof(BvhNode));
        return;
    }
    out.nodes.resize(bfs.size());
    for (size_t i = 0; i < bfs.size(); ++i) {
        const TmpNode& n = b.tmp[bfs[i]];
        const TmpNode& c0 = b.tmp[n.left];
        const TmpNode& c1 = b.tmp[n.right];
        BvhNode& o = out.nodes[i];
        pack_box(c0.box, pad, o.h0);
        pack_box(c1.box, pad, o.h1);
        o.child0 = c0.left >= 0 ? inner_index[n.left] : emit_leaf(c0);
        o.child1 = c1.left >= 0 ? inner_index[n.right] : emit_leaf(c1);
    }
    out.root = 0;
}

// ---- the binary tree collapsed to 4-wide nodes of 48 bytes (bvh.hpp, BvhNode4) ---------------------------------------
namespace {
struct WideChild { int32_t ref; float mn[3], mx[3]; };
float half_to_f32(uint32_t h) { _Float16 v; const uint16_t b = (uint16_t)h; std::memcpy(&v, &b, 2); return (float)v; }
WideChild wide_child(const BvhNode& n, int c)
{
    WideChild w; const uint32_t* h = c ? n.h1 : n.h0;
    w.ref = c ? n.child1 : n.child0;
    for (int a = 0; a < 3; ++a) { w.mn[a] = half_to_f32(h[a] & 0xFFFFu); w.mx[a] = half_to_f32(h[a] >> 16); }
    return w;
}
double wide_surface(const WideChild& w)
{
    const double dx = (double)w.mx[0] - w.mn[0], dy = (double)w.mx[1] - w.mn[1], dz = (double)w.mx[2] - w.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
}  // namespace

void build_wide(Bvh& b)
{
    b.nodes4.clear(); b.stack_need4 = 0;
    if (b.root < 0 || b.nodes.empty() || b.max_leaf > 4 || b.tris.size() >= (1u << 20)) return;
    std::vector<int32_t> bin_of(1, b.root);          // wide node -> the binary node it is rooted at; grows while it is walked (breadth-first)
    std::vector<uint32_t> deferred(1, 0u);           // deferred children on the stack when the walk enters the node, at most
    std::vector<BvhNode4> out;
    std::vector<BvhTri> tris; tris.reserve(b.tris.size());
    std::vector<uint32_t> new_first(b.tris.size(), 0xFFFFFFFFu);
    uint32_t need = 0;
    for (size_t w = 0; w < bin_of.size(); ++w) {
        const BvhNode& n = b.nodes[bin_of[w]];
        WideChild c[4]; int k = 2;
        c[0] = wide_child(n, 0); c[1] = wide_child(n, 1);
        while (k < 4) {                               // open the inner child with the largest surface
            int best = -1; double bs = -1.0;
            for (int i = 0; i < k; ++i) if (c[i].ref >= 0 && wide_surface(c[i]) > bs) { bs = wide_surface(c[i]); best = i; }
            if (best < 0) break;
            const BvhNode& m = b.nodes[c[best].ref];
            c[best] = wide_child(m, 0); c[k++] = wide_child(m, 1);
        }
        BvhNode4 o; std::memset(&o, 0, sizeof o);
        const uint32_t child_base = (uint32_t)bin_of.size(), tri_base = (uint32_t)tris.size();
        double org[3], scale[3]; uint32_t eb[3];
        for (int a = 0; a < 3; ++a) {
            float lo = c[0].mn[a], hi = c[0].mx[a];
            for (int i = 1; i < k; ++i) { lo = std::min(lo, c[i].mn[a]); hi = std::max(hi, c[i].mx[a]); }
            if (!(std::fabs(lo) < 3e38f) || !(std::fabs(hi) < 3e38f)) return;                 // a coordinate beyond the half range (+-inf): binary nodes only
            const double ext = (double)hi - (double)lo;
            int e = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -100;
            e = std::max(e, -100);
            while (std::ldexp(255.0, e) < ext) ++e;
            if (e > 100) return;
            o.org[a] = lo; org[a] = lo; scale[a] = std::ldexp(1.0, e); eb[a] = (uint32_t)(e + 127);
        }
        uint32_t lo_w[3] = { 0, 0, 0 }, hi_w[3] = { 0, 0, 0 }, meta = 0, ninner = 0;
        for (int i = 0; i < 4; ++i) {
            uint32_t mb = 0;
            for (int a = 0; a < 3; ++a) {
                uint32_t ql = 255u, qh = 0u;                                                  // unused slot: inverted box
                if (i < k) {
                    const double fl = std::floor(((double)c[i].mn[a] - org[a]) / scale[a]), fh = std::ceil(((double)c[i].mx[a] - org[a]) / scale[a]);
                    ql = (uint32_t)std::min(255.0, std::max(0.0, fl)); qh = (uint32_t)std::min(255.0, std::max(0.0, fh));
                    if (org[a] + ql * scale[a] > (double)c[i].mn[a] || org[a] + qh * scale[a] < (double)c[i].mx[a]) return;   // cannot happen (e is chosen so that 255 steps span the node)
                }
                lo_w[a] |= ql << (8 * i); hi_w[a] |= qh << (8 * i);
            }
            if (i < k && c[i].ref >= 0) {
                mb = 0x80u | ninner++;
                bin_of.push_back(c[i].ref); deferred.push_back(deferred[w] + (uint32_t)k - 1u);
            } else if (i < k) {
                const uint32_t code = ~(uint32_t)c[i].ref, first = code >> 3, cnt = (code & 7u) + 1u;
                const uint32_t at = (uint32_t)tris.size();
                for (uint32_t t = 0; t < cnt; ++t) tris.push_back(b.tris[first + t]);
                new_first[first] = at;
                mb = ((at - tri_base) << 3) | (cnt - 1u);                                      // <= 12 << 3 | 3
            }
            meta |= mb << (8 * i);
        }
        need = std::max(need, deferred[w] + (uint32_t)k - 1u);
        if (child_base + ninner >= (1u << 20) || tris.size() >= (1u << 20)) return;
        o.ew = eb[0] | eb[1] << 8 | eb[2] << 16 | ((tri_base >> 12) & 0xFFu) << 24;
        o.lox = lo_w[0]; o.hix = hi_w[0]; o.loy = lo_w[1]; o.hiy = hi_w[1]; o.loz = lo_w[2]; o.hiz = hi_w[2];
        o.meta = meta; o.bases = child_base | (tri_base & 0xFFFu) << 20;
        out.push_back(o);
        {   // what the kernels decode from these words (traverse.hpp, wide_children) must be the children collected above
            const uint32_t cbm = (o.bases & 0xFFFFFu) - 128u, nb = ~(((o.bases >> 20) | ((o.ew >> 24) << 12)) << 3);
            uint32_t inner_seen = 0;
            for (int i = 0; i < k; ++i) {
                const uint32_t t = (o.meta >> (8 * i)) & 0xFFu;
                const int32_t ref = (t & 0x80u) ? (int32_t)(cbm + t) : (int32_t)(nb - t);
                if (c[i].ref >= 0) { if (ref != (int32_t)(child_base + inner_seen) || bin_of[child_base + inner_seen] != c[i].ref) return; ++inner_seen; }
                else {
                    const uint32_t code = ~(uint32_t)c[i].ref, cnt = (code & 7u) + 1u, ncode = ~(uint32_t)ref;
                    if (ref >= 0 || (ncode & 7u) + 1u != cnt || (ncode >> 3) + cnt > tris.size() || new_first[code >> 3] != (ncode >> 3)) return;
                }
            }
        }
    }
    if (tris.size() != b.tris.size()) return;                                                 // every leaf hangs under exactly one wide node
    // commit: the triangles in the wide tree's order, the binary leaves re-pointed at them
    for (BvhNode& n : b.nodes)
        for (int32_t* ch : { &n.child0, &n.child1 })
            if (*ch < 0) { const uint32_t code = ~(uint32_t)*ch, first = code >> 3, cnt = (code & 7u) + 1u; if (first < new_first.size() && new_first[first] != 0xFFFFFFFFu) *ch = leaf_code(new_first[first], cnt); }
    // (the leaf byte of an unused slot points at triangle tri_base, which exists: only a ray with a null direction passes the inverted box, and it can hit nothing)
    b.tris.swap(tris);
    b.nodes4.swap(out);
    b.stack_need4 = need;
    if (std::getenv("MI355RT_DEBUG_BVH")) fprintf(stderr, "[mi355rt] wide BVH: %zu nodes of 4 (binary: %zu), stack need %u (binary depth %u)\n", b.nodes4.size(), b.nodes.size(), b.stack_need4, b.max_depth);
}

}  // namespace mi355rt
