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
//   k_number            preorder numbers, depth and expansion order of every node by walking up
//   k_radix_* (1 pass)  nodes into level order (a prefix of the record array is the top of the tree,
//                       which is what the extend kernel keeps in LDS)
//   k_level_starts      where each level begins in that order
//   k_fit_level         one launch per level, deepest first: a node's record = its children's boxes (a
//                       leaf's bounds, or the union of the two boxes in the child's finished record),
//                       renumbered child references and ids.  The usual single-launch bottom-up pass
//                       (second thread to arrive at a node carries on) needs a device-scope release and
//                       acquire per node, which on this chip means L2 write-backs and invalidations
//                       across eight L2s; kernel boundaries give the same ordering for a few launches
//   k_ranks             tie-break ranks of the triangles
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"
#include "dvec.h"

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
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Box b = tri_box(tris[i]);
        for (int k = 0; k < 3; ++k) {
            unsigned long long c = ordered(0.5 * (b.lo[k] + b.hi[k]));
            lo[k] = c < lo[k] ? c : lo[k];
            hi[k] = c > hi[k] ? c : hi[k];
        }
    }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off >= 1; off >>= 1) {
            unsigned long long a = __shfl_xor(lo[k], off), c = __shfl_xor(hi[k], off);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = c > hi[k] ? c : hi[k];
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s[k], lo[k]); atomicMax(&s[3 + k], hi[k]); }
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
    double wmax = 0.0;  // one cell size for all axes: cubical cells
    for (int k = 0; k < 3; ++k) wmax = fmax(wmax, unordered(bounds[3 + k]) - unordered(bounds[k]));
    for (int k = 0; k < 3; ++k) {
        double lo = unordered(bounds[k]);
        double c = 0.5 * (b.lo[k] + b.hi[k]);
        double w = wmax;
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

// Exclusive scan of `count` words by one workgroup of 1024 threads: every thread sums a contiguous run, the
// 1024 sums are scanned through LDS (wave scans by lane shuffles, then the 16 wave totals), and every thread
// walks its run again writing prefixes.
__global__ void __launch_bounds__(1024) k_radix_scan(uint32_t* __restrict__ v, int count) {
    __shared__ uint32_t s_wave[16];
    const int run = (count + 1023) / 1024;
    const int begin = (int)threadIdx.x * run;
    const int end = begin + run < count ? begin + run : count;
    uint32_t sum = 0;
    for (int i = begin; i < end; ++i) sum += v[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    uint32_t prefix = base + incl - sum;
    for (int i = begin; i < end; ++i) {
        uint32_t x = v[i];
        v[i] = prefix;
        prefix += x;
    }
}

// Scan of a long array in three steps: sums of 4096-word blocks, the scan above over those sums, then each
// block scans its own words (four per thread, wave scans by lane shuffles) on top of its block's prefix.
#define LB_SCAN_BLOCK 4096

__device__ inline uint4 load4_guarded(const uint32_t* __restrict__ v, int i, int count) {
    uint4 x = {0u, 0u, 0u, 0u};
    if (i + 3 < count) return *(const uint4*)(v + i);
    if (i < count) x.x = v[i];
    if (i + 1 < count) x.y = v[i + 1];
    if (i + 2 < count) x.z = v[i + 2];
    return x;
}

__global__ void __launch_bounds__(1024) k_scan_block_sums(const uint32_t* __restrict__ v, int count, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_wave[16];
    uint4 x = load4_guarded(v, blockIdx.x * LB_SCAN_BLOCK + (int)threadIdx.x * 4, count);
    uint32_t sum = x.x + x.y + x.z + x.w;
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w) t += s_wave[w];
        sums[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(1024) k_scan_apply(uint32_t* __restrict__ v, int count, const uint32_t* __restrict__ block_prefix) {
    __shared__ uint32_t s_wave[16];
    const int i = blockIdx.x * LB_SCAN_BLOCK + (int)threadIdx.x * 4;
    uint4 x = load4_guarded(v, i, count);
    const uint32_t sum = x.x + x.y + x.z + x.w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t p = block_prefix[blockIdx.x] + incl - sum;
    for (int w = 0; w < wave; ++w) p += s_wave[w];
    uint4 out = {p, p + x.x, p + x.x + x.y, p + x.x + x.y + x.z};
    if (i + 3 < count) *(uint4*)(v + i) = out;
    else {
        if (i < count) v[i] = out.x;
        if (i + 1 < count) v[i + 1] = out.y;
        if (i + 2 < count) v[i + 2] = out.z;
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

// level_sorted: internal nodes sorted by depth (key = depth << 32 | node).  new_index[node] = position in
// that order = final record index; level_start[d] = first position of depth d, level_start[max depth + 1] = m.
__global__ void k_level_starts(const unsigned long long* __restrict__ level_sorted, int m, int* __restrict__ new_index,
                               int* __restrict__ level_start) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long key = level_sorted[i];
    new_index[(uint32_t)key] = i;
    int d = (int)(key >> 32);
    if (i == 0 || (int)(level_sorted[i - 1] >> 32) != d) level_start[d] = i;
    if (i == m - 1) level_start[d + 1] = m;
}

// Final records of one level.  Children are one level deeper or leaves, so their records are complete.
__global__ void k_fit_level(const unsigned long long* __restrict__ level_sorted, const int* __restrict__ level_start, int level,
                            const unsigned long long* __restrict__ keys, const cpt_tri* __restrict__ tris,
                            const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ node_id,
                            const int* __restrict__ leaf_id, const int* __restrict__ new_index, cpt_node* __restrict__ nodes) {
    const int begin = level_start[level], end = level_start[level + 1];
    for (int i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        const int node = (int)(uint32_t)level_sorted[i];
        cpt_node r;
        const int refs[2] = {left[node], right[node]};
        int out_ref[2], out_id[2];
        for (int side = 0; side < 2; ++side) {
            double* dst = side ? r.rbox : r.lbox;
            const int ref = refs[side];
            if (ref < 0) {
                const int tri = (int)(uint32_t)keys[~ref];
                Box b = tri_box(tris[tri]);
                for (int k = 0; k < 3; ++k) { dst[k] = b.lo[k]; dst[3 + k] = b.hi[k]; }
                out_ref[side] = ~tri;
                out_id[side] = leaf_id[~ref];
            } else {
                const cpt_node& c = nodes[new_index[ref]];
                for (int k = 0; k < 3; ++k) {
                    dst[k] = fmin(c.lbox[k], c.rbox[k]);
                    dst[3 + k] = fmax(c.lbox[3 + k], c.rbox[3 + k]);
                }
                out_ref[side] = new_index[ref];
                out_id[side] = node_id[ref];
            }
        }
        r.lref = out_ref[0]; r.rref = out_ref[1];
        r.lid = out_id[0]; r.rid = out_id[1];
        r.self_id = node_id[node];
        r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
        nodes[i] = r;
    }
}

__global__ void k_ranks(const unsigned long long* __restrict__ keys, int n, const int* __restrict__ right,
                        const int* __restrict__ parent_leaf, const int* __restrict__ node_order, cpt_tri* __restrict__ tris) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int up = parent_leaf[p];
    tris[(uint32_t)keys[p]].test_rank = 2u * (uint32_t)node_order[up] + (right[up] == ~p ? 1u : 0u);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Work {
    unsigned long long *keys_a, *keys_b, *keys_c, *bounds;
    uint32_t *hist, *max_depth;
    int *left, *right, *parent_int, *parent_leaf, *first, *node_id, *node_order, *leaf_id, *new_index, *level_start;
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
    w.hist = (uint32_t*)take(4 * (256 * tiles + (256 * tiles + LB_SCAN_BLOCK - 1) / LB_SCAN_BLOCK + 4));
    w.max_depth = (uint32_t*)take(4);
    int** ints[] = {&w.left, &w.right, &w.parent_int, &w.parent_leaf, &w.first, &w.node_id, &w.node_order, &w.leaf_id, &w.new_index};
    for (int** q : ints) *q = (int*)take(4 * (size_t)n);
    w.level_start = (int*)take(4 * 258);
    w.bytes = off;
    return w;
}

void radix_pass(hipStream_t st, const unsigned long long* in, unsigned long long* out, int n, int shift, uint32_t* hist) {
    int tiles = (n + LB_SORT_TILE - 1) / LB_SORT_TILE;
    k_radix_hist<<<tiles, 64, 0, st>>>(in, n, shift, hist, tiles);
    const int count = 256 * tiles;
    if (count <= 4 * LB_SCAN_BLOCK) {
        k_radix_scan<<<1, 1024, 0, st>>>(hist, count);
    } else {
        const int blocks = (count + LB_SCAN_BLOCK - 1) / LB_SCAN_BLOCK;
        uint32_t* sums = hist + count;  // room for it is part of the histogram allocation
        k_scan_block_sums<<<blocks, 1024, 0, st>>>(hist, count, sums);
        k_radix_scan<<<1, 1024, 0, st>>>(sums, blocks);
        k_scan_apply<<<blocks, 1024, 0, st>>>(hist, count, sums);
    }
    k_radix_scatter<<<tiles, 64, 0, st>>>(in, out, n, shift, hist, tiles);
}

}  // namespace

// Shading records: the material's record (filled on the host, one per material) with the triangle's own
// geometry and vertex normals put in.  One thread per triangle.
__global__ void k_build_shade(const cpt_tri* __restrict__ tris, const cpt_tri_normals* __restrict__ normals,
                              const ShadeRec* __restrict__ by_material, int n, ShadeRec* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cpt_tri& t = tris[i];
    const cpt_tri_normals& tn = normals[i];
    ShadeRec r = by_material[t.material];  // diff_w holds kd and inv_p_trans holds p_trans on arrival
    r.tri_flags = t.flags;
    for (int k = 0; k < 3; ++k) {
        r.a[k] = t.a[k]; r.b[k] = t.b[k]; r.c[k] = t.c[k];
        r.na[k] = tn.na[k]; r.nb[k] = tn.nb[k]; r.nc[k] = tn.nc[k];
    }
    // per-hit expressions of the shading code whose operands are the same at every hit (same operations, same order)
    const dvec3 n1 = normalize(dv(t.n)), n2 = normalize(n1);
    const double pdf = 1 / CPT_PI, theta = 1;
    const dvec3 brdf = dv(r.diff_w) / CPT_PI;
    const dvec3 dw = brdf * theta / (pdf * r.p_diff);
    const dvec3 sw = dv(r.ks) * 1.0 / r.p_spec;
    r.n1[0] = n1.x; r.n1[1] = n1.y; r.n1[2] = n1.z;
    r.n2[0] = n2.x; r.n2[1] = n2.y; r.n2[2] = n2.z;
    r.diff_w[0] = dw.x; r.diff_w[1] = dw.y; r.diff_w[2] = dw.z;
    r.spec_w[0] = sw.x; r.spec_w[1] = sw.y; r.spec_w[2] = sw.z;
    r.inv_p_trans = 1 / r.inv_p_trans;
    double q = (1.0 - r.index) / (1.0 + r.index);
    r.r0 = q * q;
    const dvec3 ca = dv(t.c) - dv(t.a);
    r.ca[0] = ca.x; r.ca[1] = ca.y; r.ca[2] = ca.z;
    r.denom = length(cross(dv(t.b) - dv(t.a), ca));
    out[i] = r;
}

void launch_build_shade(hipStream_t st, const cpt_tri* d_tris, const cpt_tri_normals* d_normals, const ShadeRec* d_by_material,
                        int n_tris, ShadeRec* d_out) {
    k_build_shade<<<(n_tris + LB_THREADS - 1) / LB_THREADS, LB_THREADS, 0, st>>>(d_tris, d_normals, d_by_material, n_tris, d_out);
}

// ---- refit: new vertices, same tree ------------------------------------------------------------------------

// Take the geometry, material and flags of the new triangles, keep the tie-break ranks the tree gave the old ones.
__global__ void k_update_tris(const cpt_tri* __restrict__ fresh, int n, cpt_tri* __restrict__ tris) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cpt_tri t = fresh[i];
    t.test_rank = tris[i].test_rank;
    tris[i] = t;
}

// Boxes of the records order[begin, end), all of one level: a leaf child's box is its triangle's bounds, an
// internal child's the union of the two boxes in its record, which an earlier launch (a deeper level) has redone.
__global__ void k_refit_level(const int* __restrict__ order, int begin, int end, const cpt_tri* __restrict__ tris,
                              cpt_node* __restrict__ nodes) {
    for (int i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        cpt_node& r = nodes[order[i]];
        for (int side = 0; side < 2; ++side) {
            double* dst = side ? r.rbox : r.lbox;
            const int ref = side ? r.rref : r.lref;
            if (ref < 0) {
                Box b = tri_box(tris[~ref]);
                for (int k = 0; k < 3; ++k) { dst[k] = b.lo[k]; dst[3 + k] = b.hi[k]; }
            } else {
                const cpt_node& c = nodes[ref];
                for (int k = 0; k < 3; ++k) {
                    dst[k] = fmin(c.lbox[k], c.rbox[k]);
                    dst[3 + k] = fmax(c.lbox[3 + k], c.rbox[3 + k]);
                }
            }
        }
    }
}

void launch_update_tris(hipStream_t st, const cpt_tri* d_fresh, int n_tris, cpt_tri* d_tris) {
    k_update_tris<<<(n_tris + LB_THREADS - 1) / LB_THREADS, LB_THREADS, 0, st>>>(d_fresh, n_tris, d_tris);
}

// d_order: record indices sorted by level; level_start[l] .. level_start[l + 1]: the records of level l (host array,
// n_levels + 1 entries).  Deepest level first.
void launch_refit(hipStream_t st, const int* d_order, const int* level_start, int n_levels, const cpt_tri* d_tris, cpt_node* d_nodes) {
    for (int level = n_levels - 1; level >= 0; --level) {
        const int begin = level_start[level], end = level_start[level + 1];
        if (end <= begin) continue;
        int blocks = (end - begin + LB_THREADS - 1) / LB_THREADS;
        if (blocks > 4096) blocks = 4096;
        k_refit_level<<<blocks, LB_THREADS, 0, st>>>(d_order, begin, end, d_tris, d_nodes);
    }
}

size_t lbvh_work_bytes(int n_tris) { return carve(nullptr, n_tris).bytes; }

// Builds the tree over d_tris[0, n_tris) into d_nodes[0, n_tris - 1) and fills the triangles' test_rank.
// `morton_bits` per axis in [0, 10].  *h_height receives the tree's height in levels (root = 1) after a
// synchronisation of `st`.  n_tris >= 2.
hipError_t launch_lbvh_build(hipStream_t st, cpt_tri* d_tris, int n_tris, cpt_node* d_nodes, void* d_work, int morton_bits,
                             int* h_height, bool keep_ranks) {
    const int n = n_tris, m = n - 1;
    Work w = carve((char*)d_work, n);
    hipError_t e;
    if ((e = hipMemsetAsync(w.bounds, 0xff, 8 * 3, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.bounds + 3, 0, 8 * 3, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.max_depth, 0, 4, st)) != hipSuccess) return e;
    const int gn = (n + LB_THREADS - 1) / LB_THREADS, gm = (m + LB_THREADS - 1) / LB_THREADS, gall = (n + m + LB_THREADS - 1) / LB_THREADS;
    k_centroid_bounds<<<gn < 1024 ? gn : 1024, LB_THREADS, 0, st>>>(d_tris, n, w.bounds);
    k_morton<<<gn, LB_THREADS, 0, st>>>(d_tris, n, w.bounds, morton_bits, w.keys_a);
    unsigned long long *src = w.keys_a, *dst = w.keys_b;
    for (int shift = 32; shift < 32 + 30; shift += 8) {
        radix_pass(st, src, dst, n, shift, w.hist);
        unsigned long long* t = src; src = dst; dst = t;
    }
    // src: sorted keys; dst: free, reused for the level keys
    k_hierarchy<<<gn, LB_THREADS, 0, st>>>(src, n, w.left, w.right, w.parent_int, w.parent_leaf, w.first);
    k_number<<<gall, LB_THREADS, 0, st>>>(n, w.right, w.parent_int, w.parent_leaf, w.first, w.node_id, w.node_order, w.leaf_id, dst,
                                          w.max_depth);
    // Level order: the keys are listed in expansion order, and one stable pass on the depth (< 256) keeps that
    // order inside a level.  Children are deeper than their parents, so they come after them.
    radix_pass(st, dst, w.keys_c, m, 32, w.hist);
    k_level_starts<<<gm, LB_THREADS, 0, st>>>(w.keys_c, m, w.new_index, w.level_start);
    uint32_t leaf_depth = 0;  // deepest leaf; internal nodes sit at depths 0 .. leaf_depth - 1
    if ((e = hipMemcpyAsync(&leaf_depth, w.max_depth, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (leaf_depth < 1 || leaf_depth > 255) return hipErrorUnknown;
    for (int level = (int)leaf_depth - 1; level >= 0; --level) {
        // a level of a binary tree holds at most 2^level nodes, and at most m
        long long cap = level < 30 ? (1ll << level) : (long long)m;
        if (cap > m) cap = m;
        int blocks = (int)((cap + LB_THREADS - 1) / LB_THREADS);
        if (blocks > 4096) blocks = 4096;
        k_fit_level<<<blocks, LB_THREADS, 0, st>>>(w.keys_c, w.level_start, level, src, d_tris, w.left, w.right, w.node_id, w.leaf_id,
                                                   w.new_index, d_nodes);
    }
    // keep_ranks: the triangle records came with the tie-break ranks of another tree (the reference's) and keep them
    if (!keep_ranks) k_ranks<<<gn, LB_THREADS, 0, st>>>(src, n, w.right, w.parent_leaf, w.node_order, d_tris);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *h_height = (int)leaf_depth + 1;
    return hipSuccess;
}
