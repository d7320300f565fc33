// rt_bvh.h - host-side builder of the optional acceleration structure (no counterpart in the
// reference, whose scene query is the brute-force loop of 03 Utils.h:255-271 / 04 Utils.h:362-390).
// A binary BVH over the BOUNDED slots of a path scene (spheres, rectangles incl. box faces, triangles); planes stay
// in a short list that every ray tests.  The kernel that walks it (rt_bvh_device.h) returns exactly what the brute-force
// loop returns - same distance bits, same winning element, including exact ties and the sphere tMax rule - so the
// structure only has to be CONSERVATIVE: boxes are padded so that float rounding in the slab test can never cull a
// primitive whose own test would have accepted the ray.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>
#include <hip/hip_runtime.h>
#include "rt_args.h"

namespace rt {

#define RT_BVH_LEAF  2       /* primitives per leaf at most */

struct BvhItem {
    float lo[3], hi[3];
    uint32_t ref;            // slot byte offset inside the slot section | device type (offsets are multiples of 64)
    bool bounded;
};

struct Bvh {
    // node = 4 float4 holding the boxes of BOTH children, so one fetch decides a level of the walk:
    //   (L.lo.xyz, refL) (L.hi.xyz, refR) (R.lo.xyz, -) (R.hi.xyz, -)
    // ref < 2^31: inner node index.  ref >= 2^31: leaf, refs[(ref & 0x7fffffff) >> 3 ...] holds (ref & 7) + 1 entries.
    // Node 0 is the root; a scene with a single leaf gets a root whose children are both that leaf.
    std::vector<float4> nodes;
    std::vector<uint32_t> refs;        // leaf entries: slot byte offset | device type
    std::vector<uint32_t> unbounded;   // planes: tested by every ray, in scene order
    int depth = 0;                     // inner levels on the longest root-to-leaf path = most entries the walk's stack can hold
};

namespace bvh_detail {
struct Box { float lo[3], hi[3]; };
inline Box empty_box() { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } return b; }
inline void grow(Box& b, const float* lo, const float* hi) { for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], lo[a]); b.hi[a] = std::max(b.hi[a], hi[a]); } }
inline float half_area(const Box& b) {
    float d[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
    if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0f;
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}

struct Builder {
    std::vector<BvhItem>& items;
    Bvh& out;
    // builds items[begin, end), returns its ref and box; depth = inner nodes above it
    uint32_t build(int begin, int end, int depth, Box& b) {
        b = empty_box(); Box cb = empty_box();
        for (int i = begin; i < end; ++i) {
            grow(b, items[i].lo, items[i].hi);
            float c[3] = {0.5f * (items[i].lo[0] + items[i].hi[0]), 0.5f * (items[i].lo[1] + items[i].hi[1]), 0.5f * (items[i].lo[2] + items[i].hi[2])};
            grow(cb, c, c);
        }
        const int n = end - begin;
        if (n <= RT_BVH_LEAF) {
            out.depth = std::max(out.depth, depth);
            uint32_t first = (uint32_t)out.refs.size();
            // scene order inside a leaf (not needed for the result - the kernel's tie rule is order-free)
            std::sort(items.begin() + begin, items.begin() + end, [](const BvhItem& x, const BvhItem& y) { return x.ref < y.ref; });
            for (int i = begin; i < end; ++i) out.refs.push_back(items[i].ref);
            return 0x80000000U | (first << 3) | (uint32_t)(n - 1);
        }
        // binned SAH over the three axes (16 bins); object median when the centroids coincide or the tree gets deep
        int best_axis = -1, best_bin = -1; float best_cost = INFINITY;
        const int NB = 16;
        if (depth < 20) {
            for (int a = 0; a < 3; ++a) {
                float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.0f)) continue;
                Box bb[NB]; int cnt[NB];
                for (int k = 0; k < NB; ++k) { bb[k] = empty_box(); cnt[k] = 0; }
                for (int i = begin; i < end; ++i) {
                    float c = 0.5f * (items[i].lo[a] + items[i].hi[a]);
                    int k = std::min(NB - 1, std::max(0, (int)((c - lo) / ext * NB)));
                    grow(bb[k], items[i].lo, items[i].hi); cnt[k]++;
                }
                float right_area[NB]; int right_cnt[NB];
                Box acc = empty_box(); int ac = 0;
                for (int k = NB - 1; k > 0; --k) { grow(acc, bb[k].lo, bb[k].hi); ac += cnt[k]; right_area[k] = half_area(acc); right_cnt[k] = ac; }
                acc = empty_box(); ac = 0;
                for (int k = 0; k < NB - 1; ++k) {
                    grow(acc, bb[k].lo, bb[k].hi); ac += cnt[k];
                    if (ac == 0 || right_cnt[k + 1] == 0) continue;
                    float cost = half_area(acc) * ac + right_area[k + 1] * right_cnt[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
                }
            }
        }
        int mid;
        if (best_axis >= 0) {
            const int a = best_axis; const float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
            auto it = std::partition(items.begin() + begin, items.begin() + end, [&](const BvhItem& x) {
                float c = 0.5f * (x.lo[a] + x.hi[a]);
                int k = std::min(NB - 1, std::max(0, (int)((c - lo) / ext * NB)));
                return k <= best_bin; });
            mid = (int)(it - items.begin());
        } else {
            int a = 0; float e = -1.0f;
            for (int k = 0; k < 3; ++k) if (b.hi[k] - b.lo[k] > e) { e = b.hi[k] - b.lo[k]; a = k; }
            mid = begin + n / 2;
            std::nth_element(items.begin() + begin, items.begin() + mid, items.begin() + end,
                             [a](const BvhItem& x, const BvhItem& y) { return x.lo[a] + x.hi[a] < y.lo[a] + y.hi[a]; });
        }
        if (mid <= begin || mid >= end) mid = begin + n / 2;
        const uint32_t idx = (uint32_t)(out.nodes.size() / 4);
        out.nodes.resize(out.nodes.size() + 4);
        Box bl, br;
        const uint32_t rl = build(begin, mid, depth + 1, bl), rr = build(mid, end, depth + 1, br);
        write(idx, bl, rl, br, rr);
        return idx;
    }
    void write(uint32_t idx, const Box& bl, uint32_t rl, const Box& br, uint32_t rr) {
        out.nodes[4 * idx] = make_float4(bl.lo[0], bl.lo[1], bl.lo[2], __builtin_bit_cast(float, rl));
        out.nodes[4 * idx + 1] = make_float4(bl.hi[0], bl.hi[1], bl.hi[2], __builtin_bit_cast(float, rr));
        out.nodes[4 * idx + 2] = make_float4(br.lo[0], br.lo[1], br.lo[2], 0.0f);
        out.nodes[4 * idx + 3] = make_float4(br.hi[0], br.hi[1], br.hi[2], 0.0f);
    }
};
} // namespace bvh_detail

// Pads every bounded item (absolute 1e-3 plus 1e-4 of its largest coordinate: far above the rounding of p = o + t*d and of
// the slab arithmetic for |t| up to 1e3 times the scene scale) and builds the tree.  items is reordered.
inline void build_bvh(std::vector<BvhItem>& items, Bvh& out) {
    out = Bvh();
    std::vector<BvhItem> bounded;
    for (const BvhItem& it : items) {
        if (!it.bounded) { out.unbounded.push_back(it.ref); continue; }
        BvhItem b = it;
        float m = 0.0f;
        for (int a = 0; a < 3; ++a) m = std::max(m, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
        const float pad = 1e-3f + 1e-4f * m;
        for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
        bounded.push_back(b);
    }
    if (bounded.empty()) return;   // the kernel skips the walk (DevScene.n_bvh_refs = 0)
    bvh_detail::Builder B{bounded, out};
    bvh_detail::Box box;
    const uint32_t root = B.build(0, (int)bounded.size(), 0, box);
    if (root & 0x80000000U) {      // a single leaf: the root's two children are both that leaf (testing a slot twice changes nothing)
        out.nodes.resize(4);
        B.write(0, box, root, box, root);
        out.depth = 1;
    }
}

} // namespace rt
