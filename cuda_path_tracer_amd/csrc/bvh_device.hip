// Device-side BVH construction: a Morton-order tree (Karras, "Maximizing parallelism in the
// construction of BVHs, octrees, and k-d trees", 2012) built with one thread per triangle / node, written
// straight into the node records the traversal kernels read.
//
// This is the fast alternative to the host build (reference: src/scene/bvh.cpp:101-220, which sorts and
// sweeps on one CPU thread, and src/gpu/GPUBVH.cu:9-31, which then copies the pointer tree node by
// node).  The tree is a different tree than the reference's: boxes are exact bounds, every triangle is
// in exactly one leaf, so the set of triangles a ray hits is the same up to rays that graze a box
// plane; visit sets, step counts and the order in which equal-t hits are found are this tree's own.
//
// Pipeline (n triangles, m = n - 1 internal nodes):
//   k_centroid_bounds   bounds of the triangle centres (order-preserving integer min/max atomics)
//   k_morton            key = morton(centre) << 32 | triangle
//   k_radix_*           stable LSD radix sort on the Morton bits, 8 bits a pass
//   k_hierarchy         Karras' range/split search: children, parents, first leaf of every node
//   k_fit               bottom-up: each leaf climbs, the second arrival at a node joins the two boxes
//   k_number            preorder numbers, depth and expansion order of every node by walking up
//   k_radix_* (1 pass)  nodes into level order (a prefix of the record array is the top of the tree,
//                       which is what the extend kernel keeps in LDS)
//   k_emit              final records with renumbered child references; tie-break ranks of triangles
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"

#define LB_THREADS 256
#define LB_SORT_TILE 1024  // keys per one-wave workgroup of the sort

namespace {

// Monotone map from double to uint64 (and back), so min/max run as integer atomics.
__device__ inline unsigned long long ordered(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double unordered(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

struct Box { double lo[3], hi[3]; };

__device__ inline Box tri_box(const cpt_tri& t) {
    Box b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = fmin(fmin(t.a[k], t.b[k]), t.c[k]);
        b.hi[k] = fmax(fmax(t.a[k], t.b[k]), t.c[k]);
    }
    return b;
}

// bounds[0..2] = min of centres, bounds[3..5] = max, as ordered integers
__global__ void k_centroid_bounds(const cpt_tri* __restrict__ tris, int n, unsigned long long* __restrict__ bounds) {
    __shared__ unsigned long long s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = threadIdx.x < 3 ? ~0ull : 0ull;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Box b = tri_box(tris[i]);
        for (int k = 0; k < 3; ++k) {
            unsigned long long c = ordered(0.5 * (b.lo[k] + b.hi[k]));
            atomicMin(&s[k], c);
            atomicMax(&s[3 + k], c);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s[threadIdx.x]);
}

// Spreads the low 10 bits of v so that two zero bits follow each.
__device__ inline uint32_t spread3(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// `bits` Morton bits per axis (<= 10); fewer bits give a lower, more balanced tree.
__global__ void k_morton(const cpt_tri* __restrict__ tris, int n, const unsigned long long* __restrict__ bounds, int bits,
                         unsigned long long* __restrict__ keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Box b = tri_box(tris[i]);
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        double lo = unordered(bounds[k]), hi = unordered(bounds[3 + k]);
        double c = 0.5 * (b.lo[k] + b.hi[k]);
        double w = hi - lo;
        double u = w > 0.0 ? (c - lo) / w : 0.0;
        int cell = (int)(u * 1024.0);
        cell = cell < 0 ? 0 : (cell > 1023 ? 1023 : cell);
        q[k] = ((uint32_t)cell >> (10 - bits)) << (10 - bits);
    }
    uint32_t code = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    keys[i] = ((unsigned long long)code << 32) | (uint32_t)i;
}

// ---- stable radix sort, one 8-bit digit per pass ------------------------------------------------------
// A workgroup is one wave and owns LB_SORT_TILE consecutive keys.  Pass = histogram, scan, scatter.
// hist is digit-major: hist[d * n_tiles + tile], so an exclusive scan of the whole array gives each
// (digit, tile) its first output position.

__global__ void __launch_bounds__(64) k_radix_hist(const unsigned long long* __restrict__ keys, int n, int shift,
                                                   uint32_t* __restrict__ hist, int n_tiles) {
    __shared__ uint32_t s[256];
    for (int d = threadIdx.x; d < 256; d += 64) s[d] = 0;
    __syncthreads();
    int base = blockIdx.x * LB_SORT_TILE;
    for (int j = threadIdx.x; j < LB_SORT_TILE; j += 64) {
        int i = base + j;
        if (i < n) atomicAdd(&s[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += 64) hist[(size_t)d * n_tiles + blockIdx.x] = s[d];
}

// Exclusive scan of `count` words by one workgroup of 1024 threads.
__global__ void __launch_bounds__(1024) k_radix_scan(uint32_t* __restrict__ v, int count) {
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < count; base += 1024) {
        int i = base + (int)threadIdx.x;
        uint32_t x = i < count ? v[i] : 0u;
        s_part[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            uint32_t y = threadIdx.x >= (unsigned)off ? s_part[threadIdx.x - off] : 0u;
            __syncthreads();
            s_part[threadIdx.x] += y;
            __syncthreads();
        }
        uint32_t incl = s_part[threadIdx.x];
        uint32_t carry = s_carry;
        if (i < count) v[i] = carry + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) k_radix_scatter(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                      int n, int shift, const uint32_t* __restrict__ hist, int n_tiles) {
    __shared__ uint32_t s_next[256];  // next output position of each digit for this tile
    for (int d = threadIdx.x; d < 256; d += 64) s_next[d] = hist[(size_t)d * n_tiles + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = blockIdx.x * LB_SORT_TILE;
    for (int j = 0; j < LB_SORT_TILE; j += 64) {  // 64 keys at a time, in order: stability
        int i = base + j + lane;
        bool live = i < n;
        unsigned long long key = live ? in[i] : 0ull;
        uint32_t d = (uint32_t)(key >> shift) & 255u;
        // lanes holding the same digit
        unsigned long long peers = __ballot(live);
        for (int b = 0; b < 8; ++b) {
            unsigned long long m = __ballot(live && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t rank = (uint32_t)__popcll(peers & below);
        uint32_t pos = 0;
        if (live) pos = s_next[d] + rank;
        __syncthreads();
        if (live && rank == 0) s_next[d] += (uint32_t)__popcll(peers);
        __syncthreads();
        if (live) out[pos] = key;
    }
}

// ---- hierarchy ------------------------------------------------------------------------------------------

// Length of the common prefix of the keys at sorted positions i and j; the position breaks ties between
// equal Morton codes, which splits runs of duplicates evenly.
__device__ inline int prefix(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = (keys[i] & 0xffffffff00000000ull) | (uint32_t)i;
    unsigned long long b = (keys[j] & 0xffffffff00000000ull) | (uint32_t)j;
    return __clzll((long long)(a ^ b));
}

// Child references in the working arrays: >= 0 internal node, < 0 leaf at sorted position ~ref.
__global__ void k_hierarchy(const unsigned long long* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                            int* __restrict__ parent_int, int* __restrict__ parent_leaf, int* __restrict__ first) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = prefix(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int g = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int lref = (lo == g) ? ~g : g;
    int rref = (hi == g + 1) ? ~(g + 1) : g + 1;
    left[i] = lref;
    right[i] = rref;
    first[i] = lo;
    if (lref >= 0) parent_int[lref] = i; else parent_leaf[~lref] = i;
    if (rref >= 0) parent_int[rref] = i; else parent_leaf[~rref] = i;
    if (i == 0) parent_int[0] = -1;
}

// ---- boxes ----------------------------------------------------------------------------------------------

// One thread per leaf.  A thread writes its box into its parent's record and bumps the parent's arrival
// counter; the first to arrive stops, the second (which then sees both boxes: the counter is an
// acquire-release operation at device scope) joins them and carries the result one level up.
__global__ void k_fit(const cpt_tri* __restrict__ tris, const unsigned long long* __restrict__ keys, int n,
                      const int* __restrict__ right, const int* __restrict__ parent_int, const int* __restrict__ parent_leaf,
                      uint32_t* __restrict__ arrived, cpt_node* __restrict__ work) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    Box b = tri_box(tris[(uint32_t)keys[p]]);
    int node = parent_leaf[p];
    bool from_right = right[node] == ~p;
    for (;;) {
        double* dst = from_right ? work[node].rbox : work[node].lbox;
        for (int k = 0; k < 3; ++k) { dst[k] = b.lo[k]; dst[3 + k] = b.hi[k]; }
        uint32_t before = __hip_atomic_fetch_add(&arrived[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == 0u) return;
        if (node == 0) return;  // the root's own box is never tested
        const double* other = from_right ? work[node].lbox : work[node].rbox;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = fmin(b.lo[k], other[k]);
            b.hi[k] = fmax(b.hi[k], other[3 + k]);
        }
        int up = parent_int[node];
        from_right = right[up] == node;
        node = up;
    }
}

// ---- numbering --------------------------------------------------------------------------------------------

// For a node whose first leaf is at sorted position l, with `depth` ancestors of which `turns` were
// entered through their right child: preorder number over all nodes = depth + 2l - turns, preorder
// number over internal nodes only (the order in which the traversal expands them) = depth + l - turns.
struct Climb { int depth, turns; };

__device__ inline Climb climb(const int* __restrict__ right, const int* __restrict__ parent_int, int node, Climb c) {
    while (node != 0) {
        int up = parent_int[node];
        c.depth += 1;
        c.turns += right[up] == node ? 1 : 0;
        node = up;
    }
    return c;
}

// Threads [0, m): internal nodes; [m, m + n): leaves.  level_keys[order] = depth << 32 | node, i.e. the
// internal nodes listed in expansion order, ready for one stable sort pass on the depth.
__global__ void k_number(int n, const int* __restrict__ right, const int* __restrict__ parent_int, const int* __restrict__ parent_leaf,
                         const int* __restrict__ first, int* __restrict__ node_id, int* __restrict__ node_order,
                         int* __restrict__ leaf_id, unsigned long long* __restrict__ level_keys, uint32_t* __restrict__ max_depth) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int m = n - 1;
    if (i < m) {
        Climb c = climb(right, parent_int, i, Climb{0, 0});
        int l = first[i];
        node_id[i] = c.depth + 2 * l - c.turns;
        int order = c.depth + l - c.turns;
        node_order[i] = order;
        level_keys[order] = ((unsigned long long)(uint32_t)c.depth << 32) | (uint32_t)i;
    } else if (i < m + n) {
        int p = i - m;
        int up = parent_leaf[p];
        Climb c = climb(right, parent_int, up, Climb{1, right[up] == ~p ? 1 : 0});
        leaf_id[p] = c.depth + 2 * p - c.turns;
        atomicMax(max_depth, (uint32_t)c.depth);
    }
}

__global__ void k_new_index(const unsigned long long* __restrict__ level_keys, int m, int* __restrict__ new_index) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) new_index[(uint32_t)level_keys[i]] = i;
}

__global__ void k_emit(const unsigned long long* __restrict__ keys, int n, const int* __restrict__ left, const int* __restrict__ right,
                       const int* __restrict__ parent_leaf, const int* __restrict__ node_id, const int* __restrict__ node_order,
                       const int* __restrict__ leaf_id, const int* __restrict__ new_index, const cpt_node* __restrict__ work,
                       cpt_node* __restrict__ nodes, cpt_tri* __restrict__ tris) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int m = n - 1;
    if (i < m) {
        cpt_node r = work[i];
        int lr = left[i], rr = right[i];
        r.lref = lr >= 0 ? new_index[lr] : ~(int)(uint32_t)keys[~lr];
        r.rref = rr >= 0 ? new_index[rr] : ~(int)(uint32_t)keys[~rr];
        r.lid = lr >= 0 ? node_id[lr] : leaf_id[~lr];
        r.rid = rr >= 0 ? node_id[rr] : leaf_id[~rr];
        r.self_id = node_id[i];
        r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
        nodes[new_index[i]] = r;
    } else if (i < m + n) {
        int p = i - m;
        int up = parent_leaf[p];
        tris[(uint32_t)keys[p]].test_rank = 2u * (uint32_t)node_order[up] + (right[up] == ~p ? 1u : 0u);
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Work {
    unsigned long long *keys_a, *keys_b, *keys_c, *bounds;
    uint32_t *hist, *arrived, *max_depth;
    int *left, *right, *parent_int, *parent_leaf, *first, *node_id, *node_order, *leaf_id, *new_index;
    cpt_node* recs;
    size_t bytes;
};

Work carve(char* base, int n) {
    Work w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    size_t tiles = ((size_t)n + LB_SORT_TILE - 1) / LB_SORT_TILE;
    w.keys_a = (unsigned long long*)take(8 * (size_t)n);
    w.keys_b = (unsigned long long*)take(8 * (size_t)n);
    w.keys_c = (unsigned long long*)take(8 * (size_t)n);
    w.bounds = (unsigned long long*)take(8 * 6);
    w.hist = (uint32_t*)take(4 * 256 * tiles);
    w.arrived = (uint32_t*)take(4 * (size_t)n);
    w.max_depth = (uint32_t*)take(4);
    int** ints[] = {&w.left, &w.right, &w.parent_int, &w.parent_leaf, &w.first, &w.node_id, &w.node_order, &w.leaf_id, &w.new_index};
    for (int** q : ints) *q = (int*)take(4 * (size_t)n);
    w.recs = (cpt_node*)take(sizeof(cpt_node) * (size_t)n);
    w.bytes = off;
    return w;
}

void radix_pass(hipStream_t st, const unsigned long long* in, unsigned long long* out, int n, int shift, uint32_t* hist) {
    int tiles = (n + LB_SORT_TILE - 1) / LB_SORT_TILE;
    k_radix_hist<<<tiles, 64, 0, st>>>(in, n, shift, hist, tiles);
    k_radix_scan<<<1, 1024, 0, st>>>(hist, 256 * tiles);
    k_radix_scatter<<<tiles, 64, 0, st>>>(in, out, n, shift, hist, tiles);
}

}  // namespace

size_t lbvh_work_bytes(int n_tris) { return carve(nullptr, n_tris).bytes; }

// Builds the tree over d_tris[0, n_tris) into d_nodes[0, n_tris - 1) and fills the triangles' test_rank.
// `morton_bits` per axis in [0, 10].  *h_height receives the tree's height in levels (root = 1) after a
// synchronisation of `st`.  n_tris >= 2.
hipError_t launch_lbvh_build(hipStream_t st, cpt_tri* d_tris, int n_tris, cpt_node* d_nodes, void* d_work, int morton_bits,
                             int* h_height) {
    const int n = n_tris, m = n - 1;
    Work w = carve((char*)d_work, n);
    hipError_t e;
    if ((e = hipMemsetAsync(w.bounds, 0xff, 8 * 3, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.bounds + 3, 0, 8 * 3, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.arrived, 0, 4 * (size_t)n, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.max_depth, 0, 4, st)) != hipSuccess) return e;
    const int gn = (n + LB_THREADS - 1) / LB_THREADS, gall = (n + m + LB_THREADS - 1) / LB_THREADS;
    k_centroid_bounds<<<gn < 1024 ? gn : 1024, LB_THREADS, 0, st>>>(d_tris, n, w.bounds);
    k_morton<<<gn, LB_THREADS, 0, st>>>(d_tris, n, w.bounds, morton_bits, w.keys_a);
    unsigned long long *src = w.keys_a, *dst = w.keys_b;
    for (int shift = 32; shift < 32 + 30; shift += 8) {
        radix_pass(st, src, dst, n, shift, w.hist);
        unsigned long long* t = src; src = dst; dst = t;
    }
    // src: sorted keys; dst: free, reused for the level keys
    k_hierarchy<<<gn, LB_THREADS, 0, st>>>(src, n, w.left, w.right, w.parent_int, w.parent_leaf, w.first);
    k_fit<<<gn, LB_THREADS, 0, st>>>(d_tris, src, n, w.right, w.parent_int, w.parent_leaf, w.arrived, w.recs);
    k_number<<<gall, LB_THREADS, 0, st>>>(n, w.right, w.parent_int, w.parent_leaf, w.first, w.node_id, w.node_order, w.leaf_id, dst,
                                          w.max_depth);
    // Level order: the keys are listed in expansion order, and one stable pass on the depth (< 256) keeps that
    // order inside a level.  Children are deeper than their parents, so they come after them.
    if (m > 0) {
        radix_pass(st, dst, w.keys_c, m, 32, w.hist);
        k_new_index<<<(m + LB_THREADS - 1) / LB_THREADS, LB_THREADS, 0, st>>>(w.keys_c, m, w.new_index);
    }
    k_emit<<<gall, LB_THREADS, 0, st>>>(src, n, w.left, w.right, w.parent_leaf, w.node_id, w.node_order, w.leaf_id, w.new_index,
                                        w.recs, d_nodes, d_tris);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    uint32_t depth = 0;
    if ((e = hipMemcpyAsync(&depth, w.max_depth, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *h_height = (int)depth + 1;
    return hipSuccess;
}
