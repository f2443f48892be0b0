#include "bvh_build.h"

#include <algorithm>
#include <limits>

namespace cpt {
namespace {

struct Box {
    dvec3 lo, hi;
};
inline void grow(Box& b, const PrimBounds& p) {  // glm::min / glm::max as getBox applies them (bvh.cpp:141-142)
    b.lo = dv((p.lo.x < b.lo.x) ? p.lo.x : b.lo.x, (p.lo.y < b.lo.y) ? p.lo.y : b.lo.y, (p.lo.z < b.lo.z) ? p.lo.z : b.lo.z);
    b.hi = dv((b.hi.x < p.hi.x) ? p.hi.x : b.hi.x, (b.hi.y < p.hi.y) ? p.hi.y : b.hi.y, (b.hi.z < p.hi.z) ? p.hi.z : b.hi.z);
}
inline double surface(const Box& b) {  // src/scene/bbox.cc:87-89
    return 2.0 * ((b.hi.x - b.lo.x) * (b.hi.y - b.lo.y) + (b.hi.y - b.lo.y) * (b.hi.z - b.lo.z) +
                  (b.hi.z - b.lo.z) * (b.hi.x - b.lo.x));
}

class Builder {
public:
    explicit Builder(const std::vector<PrimBounds>& prims) : prims_(prims), order_(prims.size()) {
        for (size_t i = 0; i < prims.size(); ++i) order_[i] = (int)i;
        prefix_.resize(prims.size() + 1);
        suffix_.resize(prims.size() + 1);
    }
    Bvh run() {
        Bvh out;
        if (!prims_.empty()) out.height = range(out, 0, (int)prims_.size());
        return out;
    }

private:
    // Window of candidate split positions around the median and the cost of each, as
    // split() in the reference (bvh.cpp:148-176).  prefix_[i] = box of [st, i), accumulated in
    // the reference's left-to-right order; suffix_[i] = box of [i, end).
    int choose_split(int st, int end, const Box& total) {
        const int samples = 500;
        const double trimeshCost = 1, bbCost = 1;
        int mid = st + (end - st) / 2;
        int lo = std::max(mid - samples, st + 1), hi = std::min(mid + samples, end - 1);
        int best = mid;
        if (lo >= hi) return best;
        Box acc{prims_[order_[st]].lo, prims_[order_[st]].hi};
        for (int i = st + 1; i < hi; ++i) {
            prefix_[i] = acc;  // box of [st, i)
            grow(acc, prims_[order_[i]]);
        }
        Box back{prims_[order_[end - 1]].lo, prims_[order_[end - 1]].hi};
        suffix_[end - 1] = back;
        for (int i = end - 2; i >= lo; --i) {
            grow(back, prims_[order_[i]]);
            suffix_[i] = back;  // box of [i, end)
        }
        double cost = std::numeric_limits<double>::max();
        const double total_area = surface(total);
        for (int i = lo; i < hi; ++i) {
            double my = bbCost + (surface(prefix_[i]) / total_area) * (i - st) * trimeshCost +
                        (surface(suffix_[i]) / total_area) * (end - i) * trimeshCost;
            if (my < cost) { cost = my; best = i; }
        }
        return best;
    }

    int range(Bvh& out, int st, int end) {
        int id = (int)out.nodes.size();
        out.nodes.emplace_back();
        Box b{prims_[order_[st]].lo, prims_[order_[st]].hi};
        for (int i = st + 1; i < end; ++i) grow(b, prims_[order_[i]]);
        out.nodes[id].lo = b.lo;
        out.nodes[id].hi = b.hi;
        if (end - st <= 1) {
            out.nodes[id].prim = order_[st];
            return 1;
        }
        dvec3 diff = b.hi - b.lo;
        int ax;
        if (diff.x >= diff.y && diff.x >= diff.z) ax = 0;
        else if (diff.y >= diff.x && diff.y >= diff.z) ax = 1;
        else ax = 2;
        std::sort(order_.begin() + st, order_.begin() + end,
                  [&](int p, int q) { return axis(prims_[p].center, ax) < axis(prims_[q].center, ax); });
        int mid = choose_split(st, end, b);
        int l = (int)out.nodes.size();
        int hl = range(out, st, mid);
        int r = (int)out.nodes.size();
        int hr = range(out, mid, end);
        out.nodes[id].left = l;
        out.nodes[id].right = r;
        return 1 + std::max(hl, hr);
    }

    const std::vector<PrimBounds>& prims_;
    std::vector<int> order_;
    std::vector<Box> prefix_, suffix_;
};

// Full-sweep surface-area build (build_bvh_sah).  Ranges are kept in one index array; a node's range is re-sorted
// along each axis in turn, swept from both ends, and left in the order of the winning axis.
class SahBuilder {
public:
    SahBuilder(const std::vector<PrimBounds>& prims, int max_height) : prims_(prims), order_(prims.size()), max_height_(max_height) {
        for (size_t i = 0; i < prims.size(); ++i) order_[i] = (int)i;
        suffix_area_.resize(prims.size() + 1);
        centre_.resize(prims.size());
        for (size_t i = 0; i < prims.size(); ++i)
            centre_[i] = dv(0.5 * (prims[i].lo.x + prims[i].hi.x), 0.5 * (prims[i].lo.y + prims[i].hi.y), 0.5 * (prims[i].lo.z + prims[i].hi.z));
    }
    Bvh run() {
        Bvh out;
        if (prims_.empty()) return out;
        out.nodes.reserve(2 * prims_.size());
        // explicit work list (depth-first, left first) so that node ids come out in preorder and a deep branch costs no call stack
        struct Item { int st, end, depth, parent; bool right; };
        std::vector<Item> todo;
        todo.push_back({0, (int)prims_.size(), 1, -1, false});
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            const int id = (int)out.nodes.size();
            out.nodes.emplace_back();
            if (it.parent >= 0) (it.right ? out.nodes[it.parent].right : out.nodes[it.parent].left) = id;
            Box b{prims_[order_[it.st]].lo, prims_[order_[it.st]].hi};
            for (int i = it.st + 1; i < it.end; ++i) grow(b, prims_[order_[i]]);
            out.nodes[id].lo = b.lo;
            out.nodes[id].hi = b.hi;
            out.height = std::max(out.height, it.depth);
            if (it.end - it.st <= 1) {
                out.nodes[id].prim = order_[it.st];
                continue;
            }
            const int cut = it.depth >= max_height_ - balanced_levels(it.end - it.st) ? median(it.st, it.end, b) : best_cut(it.st, it.end);
            todo.push_back({cut, it.end, it.depth + 1, id, true});   // popped second: the right subtree follows the whole left one
            todo.push_back({it.st, cut, it.depth + 1, id, false});
        }
        return out;
    }

private:
    static int balanced_levels(int n) {  // levels a median-split subtree over n primitives takes below its root
        int l = 0;
        while ((1 << l) < n) ++l;
        return l;
    }
    void sort_axis(int st, int end, int ax) {
        std::sort(order_.begin() + st, order_.begin() + end, [&](int p, int q) {
            const double a = axis(centre_[p], ax), b = axis(centre_[q], ax);
            return a < b || (a == b && p < q);
        });
    }
    int median(int st, int end, const Box& b) {
        const dvec3 diff = b.hi - b.lo;
        const int ax = (diff.x >= diff.y && diff.x >= diff.z) ? 0 : (diff.y >= diff.z ? 1 : 2);
        sort_axis(st, end, ax);
        return st + (end - st) / 2;
    }
    int best_cut(int st, int end) {
        double best = std::numeric_limits<double>::infinity();
        int best_ax = 0, best_i = st + (end - st) / 2;
        for (int ax = 0; ax < 3; ++ax) {
            sort_axis(st, end, ax);
            Box back{prims_[order_[end - 1]].lo, prims_[order_[end - 1]].hi};
            suffix_area_[end - 1] = surface(back);
            for (int i = end - 2; i > st; --i) {
                grow(back, prims_[order_[i]]);
                suffix_area_[i] = surface(back);  // area of the box of [i, end)
            }
            Box front{prims_[order_[st]].lo, prims_[order_[st]].hi};
            for (int i = st + 1; i < end; ++i) {  // cut before position i
                const double cost = surface(front) * (i - st) + suffix_area_[i] * (end - i);
                if (cost < best) { best = cost; best_ax = ax; best_i = i; }
                grow(front, prims_[order_[i]]);
            }
        }
        if (best_ax != 2) sort_axis(st, end, best_ax);  // the range is in z order after the loop
        return best_i;
    }

    const std::vector<PrimBounds>& prims_;
    std::vector<int> order_;
    int max_height_;
    std::vector<double> suffix_area_;
    std::vector<dvec3> centre_;
};

}  // namespace

Bvh build_bvh(const std::vector<PrimBounds>& prims) {
    Builder b(prims);
    return b.run();
}

Bvh build_bvh_sah(const std::vector<PrimBounds>& prims, int max_height) {
    SahBuilder b(prims, max_height);
    return b.run();
}

std::vector<uint32_t> test_ranks(const Bvh& bvh, size_t n_prims) {
    std::vector<uint32_t> rank(n_prims, 0u);
    if (bvh.nodes.empty() || bvh.nodes[0].prim >= 0) return rank;
    std::vector<int> todo{0};
    uint32_t order = 0;
    while (!todo.empty()) {
        const BvhNode& n = bvh.nodes[(size_t)todo.back()];
        todo.pop_back();
        const BvhNode &l = bvh.nodes[(size_t)n.left], &r = bvh.nodes[(size_t)n.right];
        if (l.prim >= 0) rank[(size_t)l.prim] = 2u * order;
        if (r.prim >= 0) rank[(size_t)r.prim] = 2u * order + 1u;
        if (r.prim < 0) todo.push_back(n.right);
        if (l.prim < 0) todo.push_back(n.left);
        ++order;
    }
    return rank;
}

}  // namespace cpt
