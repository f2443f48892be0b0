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

}  // namespace

Bvh build_bvh(const std::vector<PrimBounds>& prims) {
    Builder b(prims);
    return b.run();
}

}  // namespace cpt
