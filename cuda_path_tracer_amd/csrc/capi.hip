// Implementation of the C ABI declared in include/cpt_amd.h: context, scene upload, the render
// loop of the wavefront pipeline, and the parity hooks.  Host code only; the kernels live in
// kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <cmath>
#include <vector>

#include "cpt_amd.h"
#include "launch.h"

#define CPT_STACK_ENTRIES_HOST 32  // must equal CPT_STACK_ENTRIES in device_core.h
#define CPT_MAX_DEPTH 1022
#define CPT_MAX_LANES 4

static_assert(sizeof(cpt_node) == 128, "cpt_node must be 128 bytes");
static_assert(sizeof(cpt_tri) == 128, "cpt_tri must be 128 bytes");
static_assert(sizeof(cpt_tri_normals) == 72, "cpt_tri_normals must be 72 bytes");
static_assert(sizeof(cpt_material) == 136, "cpt_material must be 136 bytes");

// Working memory of cpt_render_noalloc: ray streams, hit records, per-path radiance, segment counters, claim
// cursors and stack-spill columns of one batch, carved out of this array instead of out of hipMalloc'ed blocks.
// Every GPU the library runs on holds its own copy, at its own address, from the moment the code object is
// loaded until the process ends.  The name is a host-side handle only: the address kernels and memsets use is
// the one hipGetSymbolAddress returns with that GPU current (static_scratch_for, called by cpt_create).
static __device__ __align__(256) unsigned char scratch[CPT_NOALLOC_SCRATCH_BYTES];
#define CPT_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[CPT_MAX_DEVICES];  // per device: where that GPU keeps `scratch`

// Address of `scratch` on `device`, which must be the current device.  Looked up once per device.
static hipError_t static_scratch_for(int device, unsigned char** out) {
    *out = nullptr;
    if (device < 0 || device >= CPT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

struct cpt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    void *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_nodes32 = nullptr;
    void* d_nodes_pw = nullptr;  // node records with children in stack-minimal order (or d_nodes itself)
    // d_nodes_pw and d_tris restated for rays that start at the camera's eye (launch_eye_records), allocated with the scene
    // and filled in by the first pin-hole render that uses them; eye_of is the eye they were made for
    void *d_nodes_eye = nullptr, *d_tris_eye = nullptr;
    double eye_of[3] = {0, 0, 0};
    bool eye_current = false;
    // Whether the tree on the device has the properties the binary32 extend variants rely on: every child box lies inside
    // its parent's box and every leaf box is exactly the bounds of its triangle's vertices (then a leaf's box passing the
    // exact test implies every box above it passes, and the root's two child boxes bound every plane of the tree).  The
    // library's own trees have them by construction; an uploaded tree is checked.
    bool tree_nested = false;
    // Triangles are kept on the device in an order of the library's own when the scene does not fit the extend kernel's
    // record cache whole (largest bounding box first: the cached prefix then holds the triangles most rays test);
    // tri_ext[internal] = the caller's index, tri_int the inverse.  Empty: the caller's order.
    std::vector<int32_t> tri_ext, tri_int;
    DScene sc{};
    bool has_scene = false;
    int n_tree_nodes = 0, tree_height = 0, measured_height = 0;
    int stack_need = 0;  // most node-stack entries the reference walk can hold at once on the tree in place (see minimise_stack)

    // A lane is everything one batch of paths needs while it is in flight.  Consecutive batches of a
    // render go to alternating lanes, each on its own HIP stream, so that the arithmetic-bound extend
    // kernel of one batch runs under the memory-bound shade kernel of the other.
    struct Lane {
        hipStream_t own_stream = nullptr;  // lane 0 runs on the context's stream instead
        void* d_paths = nullptr;           // one allocation: two ray streams, hit buffer, per-slot radiance
        size_t path_capacity = 0;          // in path slots
        RayStream R[2]{};
        HitBuf H{};
        Radiance C{};
        uint32_t* d_counters = nullptr;    // per bounce the lengths of the stream's segments (CPT_PW_CURSOR_WORDS words)
        uint32_t* d_cursors = nullptr;     // per bounce CPT_PW_CURSOR_WORDS claim counters of the persistent extend kernel
        uint32_t* d_spill = nullptr;       // stack overflow columns of the persistent-wave extend kernel
        int spill_blocks = 0;
        hipEvent_t resolved = nullptr;     // recorded after the lane's latest k_resolve
    };
    Lane lanes[CPT_MAX_LANES];
    int n_lanes = 2;
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_tail = nullptr;          // recorded on the context's stream after the latest asynchronous call
    bool tail_recorded = false;
    unsigned long long* d_work = nullptr;  // paths, segments, hits, node steps, triangle tests
    unsigned char* d_static_scratch = nullptr;  // this GPU's copy of the file-scope `scratch` array (never freed)

    bool time_kernels = false, count_work = false;
    cpt_stats stats{};
    std::vector<hipEvent_t> events;

    int max_blocks = 0;            // grid cap of the grid-stride kernels
    // Path slots of one batch of the wavefront pipeline, at most (209 bytes each: 7 GB per batch in flight).  Larger
    // batches render a few per cent faster (shorter launch tails: dragon +6 % at 2^27), but a process that allocates
    // tens of GB and exits leaves the driver seconds of reclaiming that the next process waits for, which matters
    // more to a command line that renders in half a second; long-lived callers raise it with cpt_set_option.
    long long batch_paths = 1 << 25;
    int pw_blocks = 0;             // persistent workgroups of the extend kernel (one per CU)
    int pw_chunk = 0;              // rays a persistent wave claims per atomic; 0: chosen from the scene, see chunk_for
    int pw_refill = 12, pw_leaf_go = 0, pw_full_go = 1;  // pw_leaf_go 0: chosen from the walk, see leaf_go_for
    int pw_box_repeat = 0;  // 0: chosen from the tree, see box_repeat_for
    int pw_eye = 1;         // first rays of a pin-hole render through the shared-origin instance of the persistent extend kernel
    bool caller_ranks = false;  // the scene in place was uploaded with CPT_SCENE_KEEP_RANKS (a tree other than the reference's)
    int extend_variant = 1;        // 0: one ray per thread; 1: persistent waves with refill; 2: as 1, nearest-first with culling
};

static thread_local std::string g_create_error;

static int fail(cpt_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}
#define HIP_TRY(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail((c), CPT_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Leave a marker behind the asynchronous work a call has queued on the context's stream (see cpt_set_stream).
static int mark_tail(cpt_ctx* c) {
    HIP_TRY(c, hipEventRecord(c->ev_tail, c->stream));
    c->tail_recorded = true;
    return CPT_OK;
}

static void free_scene(cpt_ctx* c) {
    if (c->d_nodes_pw && c->d_nodes_pw != c->d_nodes) (void)hipFree(c->d_nodes_pw);
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_tris) (void)hipFree(c->d_tris);
    if (c->d_shade) (void)hipFree(c->d_shade);
    if (c->d_nodes32) (void)hipFree(c->d_nodes32);
    if (c->d_nodes_eye) (void)hipFree(c->d_nodes_eye);
    if (c->d_tris_eye) (void)hipFree(c->d_tris_eye);
    c->d_nodes = c->d_tris = c->d_shade = c->d_nodes32 = c->d_nodes_pw = c->d_nodes_eye = c->d_tris_eye = nullptr;
    c->eye_current = false;
    c->has_scene = false;
    c->tri_ext.clear();  // the index translation belongs to the scene that is gone
    c->tri_int.clear();
}
static void free_paths(cpt_ctx::Lane& l) {
    if (l.d_paths) (void)hipFree(l.d_paths);
    l.d_paths = nullptr;
    l.path_capacity = 0;
}
static hipStream_t lane_stream(cpt_ctx* c, int k) { return k == 0 ? c->stream : c->lanes[k].own_stream; }

extern "C" {

int cpt_create(int device, cpt_ctx** out) {
    if (!out) return fail(nullptr, CPT_EINVAL, "cpt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(nullptr, CPT_EINVAL, "cpt_create: device %d not in [0, %d)", device, n);
    HIP_TRY(nullptr, hipSetDevice(device));
    cpt_ctx* c = new cpt_ctx;
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete c; return fail(nullptr, CPT_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); }
    c->max_blocks = prop.multiProcessorCount * 8;
    c->pw_blocks = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamDefault);
    if (e != hipSuccess) { delete c; return fail(nullptr, CPT_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    for (int k = 0; k < CPT_MAX_LANES && e == hipSuccess; ++k) {
        e = hipMalloc((void**)&c->lanes[k].d_counters, (size_t)(CPT_MAX_DEPTH + 2) * CPT_PW_CURSOR_WORDS * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->lanes[k].d_cursors, (size_t)(CPT_MAX_DEPTH + 2) * CPT_PW_CURSOR_WORDS * sizeof(uint32_t));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->lanes[k].resolved, hipEventDisableTiming);
        if (e == hipSuccess && k > 0) e = hipStreamCreateWithFlags(&c->lanes[k].own_stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming);
    if (e == hipSuccess) e = static_scratch_for(device, &c->d_static_scratch);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_work, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->d_work, 0, 8 * sizeof(unsigned long long));
    if (e != hipSuccess) { cpt_destroy(c); return fail(nullptr, CPT_EHIP, "context set-up: %s", hipGetErrorString(e)); }
    *out = c;
    return CPT_OK;
}

void cpt_destroy(cpt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < CPT_MAX_LANES; ++k) {
        cpt_ctx::Lane& l = c->lanes[k];
        if (l.own_stream) (void)hipStreamSynchronize(l.own_stream);
        free_paths(l);
        if (l.d_counters) (void)hipFree(l.d_counters);
        if (l.d_cursors) (void)hipFree(l.d_cursors);
        if (l.d_spill) (void)hipFree(l.d_spill);
        if (l.resolved) (void)hipEventDestroy(l.resolved);
        if (l.own_stream) (void)hipStreamDestroy(l.own_stream);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    free_scene(c);
    if (c->d_work) (void)hipFree(c->d_work);
    for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* cpt_last_error(const cpt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int cpt_set_stream(cpt_ctx* c, void* s) {
    if (!c) return CPT_EINVAL;
    hipStream_t next = s == CPT_OWN_STREAM ? c->own_stream : (hipStream_t)s;
    if (next == c->stream) return CPT_OK;
    // The context's buffers (lane 0's streams and counters, the scene arrays) are ordered by the stream they were
    // last used on only: work queued on the new stream must not overtake what is still running on the old one.  The
    // new stream waits for the event the latest asynchronous call left behind; the old stream's handle is not
    // touched (its owner may have destroyed it already), and the context moves to the new stream even when the
    // wait could not be queued, so that it stays usable.
    c->stream = next;
    if (c->tail_recorded) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamWaitEvent(next, c->ev_tail, 0));
    }
    return CPT_OK;
}

int cpt_set_instrumentation(cpt_ctx* c, int time_kernels, int count_work) {
    if (!c) return CPT_EINVAL;
    c->time_kernels = time_kernels != 0;
    c->count_work = count_work != 0;
    return CPT_OK;
}

int cpt_set_option(cpt_ctx* c, const char* key, long long value) {
    if (!c || !key) return CPT_EINVAL;
    if (!strcmp(key, "batch_paths")) {
        if (value < 64 || value > (1ll << 28)) return fail(c, CPT_EINVAL, "batch_paths must be in [64, 2^28]");
        c->batch_paths = value;
    } else if (!strcmp(key, "streams")) {
        if (value < 1 || value > CPT_MAX_LANES) return fail(c, CPT_EINVAL, "streams must be in [1, %d]", CPT_MAX_LANES);
        c->n_lanes = (int)value;
    } else if (!strcmp(key, "max_blocks")) {
        if (value < 1) return fail(c, CPT_EINVAL, "max_blocks must be >= 1");
        c->max_blocks = (int)value;
    } else if (!strcmp(key, "pw_blocks")) {
        if (value < 1) return fail(c, CPT_EINVAL, "pw_blocks must be >= 1");
        c->pw_blocks = (int)value;
    } else if (!strcmp(key, "pw_chunk")) {
        if (value < 0 || value > (1 << 20)) return fail(c, CPT_EINVAL, "pw_chunk must be in [0, 2^20]");
        c->pw_chunk = (int)value;
    } else if (!strcmp(key, "pw_refill")) {
        if (value < 1 || value > 64) return fail(c, CPT_EINVAL, "pw_refill must be in [1, 64]");
        c->pw_refill = (int)value;
    } else if (!strcmp(key, "pw_full_go")) {
        if (value < 1 || value > 64) return fail(c, CPT_EINVAL, "pw_full_go must be in [1, 64]");
        c->pw_full_go = (int)value;
    } else if (!strcmp(key, "pw_box_repeat")) {
        if (value != 0 && value != 2 && value != 3) return fail(c, CPT_EINVAL, "pw_box_repeat must be 0 (automatic), 2 or 3");
        c->pw_box_repeat = (int)value;
    } else if (!strcmp(key, "pw_eye")) {
        if (value != 0 && value != 1) return fail(c, CPT_EINVAL, "pw_eye must be 0 or 1");
        c->pw_eye = (int)value;
    } else if (!strcmp(key, "pw_leaf_go")) {
        if (value < 0 || value > 64) return fail(c, CPT_EINVAL, "pw_leaf_go must be in [0, 64]");
        c->pw_leaf_go = (int)value;
    } else if (!strcmp(key, "extend_variant")) {
        if (value < 0 || value > 4) return fail(c, CPT_EINVAL, "extend_variant must be 0 .. 4");
        c->extend_variant = (int)value;
    } else {
        return fail(c, CPT_EINVAL, "unknown option '%s'", key);
    }
    return CPT_OK;
}

// Which instance of the persistent extend kernel the scene in place takes: 2 when it runs without any global-memory
// record or stack traffic, 1 when at least the node stack stays in LDS, 0 otherwise.  The reference walk postpones at
// most stack_need nodes; the nearest-first walks choose their own order, for which only the height bounds the stack.
// The binary32 test (3, 4) and the culling (2, 4) need a nested tree whose leaf boxes are their triangles' exact bounds: the first
// so that a leaf's own box decides, the second so that no hit lies before the point where the ray enters a box above it or
// the leaf's own.  On any other tree those variants run as variant 1, whose hits they would have returned.
static bool variant_f32(const cpt_ctx* c) { return c->extend_variant >= 3 && c->tree_nested; }
static bool variant_cull(const cpt_ctx* c) { return (c->extend_variant == 2 || c->extend_variant == 4) && c->tree_nested; }
static int scene_mode(const cpt_ctx* c) {
    if (!c->has_scene) return 0;
    const int stack = variant_cull(c) ? c->measured_height - 2 : std::min(c->stack_need, c->measured_height - 2);
    if (stack > CPT_PW_STACK_LDS_HOST) return 0;
    if (variant_f32(c)) {  // 128-byte slots the scene takes in the record cache (binary32 node records are half a slot each)
        const long long slots = (c->sc.n_nodes + 1) / 2 + (long long)c->sc.n_tris;
        return slots <= CPT_PW_CACHE_RECORDS_HOST ? 2 : 1;
    }
    // the exact variants' resident instance has a record layout, a stack and a cache size of its own
    const bool cull = variant_cull(c);
    return stack <= cpt_pw_resident_stack(cull) && cpt_pw_resident_bytes(c->sc.n_nodes, c->sc.n_tris) <= cpt_pw_resident_cache_bytes(cull) ? 2 : 1;
}

// Rays a wave claims at a time.  A claim stalls the wave for an atomic's round trip; a chunk is also what a wave may be
// left holding when every other wave has run dry, so the kernel itself shortens it for short streams (at least eight
// claims per wave).  Measured with launches of 2^25 .. 2^28 rays: 256 is as good as or better than 128 on every bundled
// scene (teapot +1.7 % at 2^28), 64 loses 3-13 %, 512 is the same as 256.
static int chunk_for(const cpt_ctx* c) {
    return c->pw_chunk > 0 ? c->pw_chunk : 256;
}

// Structural validation of the flattened scene: every kernel indexes these arrays without bounds
// checks, so a malformed scene must be refused here.
static int validate_triangles(cpt_ctx* c, const cpt_scene_desc* s) {
    if (s->n_tris < 1) return fail(c, CPT_EINVAL, "scene has no triangles");
    if (s->n_materials < 1) return fail(c, CPT_EINVAL, "scene has no materials");
    if (!s->tris || !s->tri_normals || !s->materials) return fail(c, CPT_EINVAL, "scene arrays missing");
    for (int i = 0; i < s->n_tris; ++i)
        if (s->tris[i].material < 0 || s->tris[i].material >= s->n_materials) return fail(c, CPT_EINVAL, "triangle %d: bad material", i);
    return CPT_OK;
}

// The scene as the device will hold it.  When triangles and nodes do not all fit the persistent extend kernel's LDS
// record cache, the cache holds a prefix of each array; node records already come top of the tree first, and the
// triangles are put largest bounding box first here (a ray tests a leaf's triangle when it pierces the leaf's box, so
// the few triangles with scene-sized boxes -- walls, floors -- take most of the tests: on the dragon scene the eight
// largest take over 90 % of them).  Hit records, shading records and tie-break ranks use the internal index; the entry
// points that report triangle indices translate back.
struct OrderedScene {
    std::vector<int32_t> tri_ext, tri_int;  // a freshly chosen order; moved into the context once the new scene is in place
    std::vector<cpt_tri> tris;
    std::vector<cpt_tri_normals> normals;
    std::vector<cpt_node> nodes;
    cpt_scene_desc desc;
};
static double box_area_of(const cpt_tri& t) {
    double e[3];
    for (int k = 0; k < 3; ++k) e[k] = std::max(t.a[k], std::max(t.b[k], t.c[k])) - std::min(t.a[k], std::min(t.b[k], t.c[k]));
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
}
// new_order: choose the order afresh (upload, build); otherwise apply the one in place (refit).  `out` receives the
// description to use from here on, which points into `os` or is *s itself when the caller's order stands.
static const cpt_scene_desc* order_scene(cpt_ctx* c, const cpt_scene_desc* s, bool new_order, bool with_nodes, OrderedScene& os) {
    const int n = s->n_tris;
    if (new_order) {  // into `os`: the context keeps the old scene's tables until the old scene is gone (commit_order)
        const long long n_nodes = with_nodes ? s->n_nodes : (long long)n - 1;
        if (cpt_pw_resident_bytes(n_nodes, n) > cpt_pw_resident_cache_bytes(false)) {  // it cannot live in LDS whole
            std::vector<double> area((size_t)n);
            for (int i = 0; i < n; ++i) area[(size_t)i] = box_area_of(s->tris[i]);
            os.tri_ext.resize((size_t)n);
            for (int i = 0; i < n; ++i) os.tri_ext[(size_t)i] = i;
            std::stable_sort(os.tri_ext.begin(), os.tri_ext.end(), [&](int32_t a, int32_t b) {
                const double x = area[(size_t)a], y = area[(size_t)b];
                return (x > y) || (x != x && y == y);  // NaN areas (non-finite vertices are refused elsewhere) first, deterministically
            });
            os.tri_int.resize((size_t)n);
            for (int i = 0; i < n; ++i) os.tri_int[(size_t)os.tri_ext[(size_t)i]] = i;
        }
    }
    const std::vector<int32_t>& ext = new_order ? os.tri_ext : c->tri_ext;
    const std::vector<int32_t>& inv = new_order ? os.tri_int : c->tri_int;
    if (ext.empty() || (int)ext.size() != n) return s;
    os.tris.resize((size_t)n);
    os.normals.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        os.tris[(size_t)i] = s->tris[ext[(size_t)i]];
        os.normals[(size_t)i] = s->tri_normals[ext[(size_t)i]];
    }
    os.desc = *s;
    os.desc.tris = os.tris.data();
    os.desc.tri_normals = os.normals.data();
    if (with_nodes && s->n_nodes > 0) {
        os.nodes.assign(s->nodes, s->nodes + s->n_nodes);
        for (cpt_node& nd : os.nodes) {
            if (nd.lref < 0) nd.lref = ~inv[(size_t)(~nd.lref)];
            if (nd.rref < 0) nd.rref = ~inv[(size_t)(~nd.rref)];
        }
        os.desc.nodes = os.nodes.data();
    }
    return &os.desc;
}
// The new scene's index translation takes effect together with the new scene (free_scene has cleared the old one's).
static void commit_order(cpt_ctx* c, OrderedScene& os) {
    c->tri_ext = std::move(os.tri_ext);
    c->tri_int = std::move(os.tri_int);
}

// The reference's walk tests both children of a node, goes on with the first that is an inner node whose box was hit and
// postpones the other on its stack (src/gpu/GPUBVH.cu:80-97); which of the two it takes first has no influence on what
// is tested.  While the walk is below the first child the second one occupies a stack entry, below the second nothing
// of this node does: need(node) = max(need(first) + 1, need(second)) when both children are inner nodes, smallest when
// the child whose subtree can itself postpone less goes first (the register-numbering argument of Sethi and Ullman,
// mirrored).  The records are rewritten with that child in the left slot -- boxes, references and canonical ids swapped
// together; tie-break ranks were taken from the caller's order beforehand -- and the root's need is returned: the most
// entries the walk's stack can ever hold.  On the bundled scenes: 4 instead of 9 (teapot), 9 instead of 14 (dragon),
// 10 instead of 15 (three dragons), against tree heights of 19 to 25.  Records come after their parents.
static int minimise_stack(std::vector<cpt_node>& nodes) {
    std::vector<int> need(nodes.size(), 0);
    for (size_t k = nodes.size(); k-- > 0;) {
        cpt_node& nd = nodes[k];
        const bool li = nd.lref >= 0, ri = nd.rref >= 0;
        const int nl = li ? need[(size_t)nd.lref] : 0, nr = ri ? need[(size_t)nd.rref] : 0;
        if (li && ri) {
            if (nl > nr) {
                for (int q = 0; q < 6; ++q) std::swap(nd.lbox[q], nd.rbox[q]);
                std::swap(nd.lref, nd.rref);
                std::swap(nd.lid, nd.rid);
            }
            need[k] = std::max(std::max(nl, nr), std::min(nl, nr) + 1);
        } else {
            need[k] = std::max(nl, nr);
        }
    }
    return nodes.empty() ? 0 : need[0];
}

// Whether every child box of the tree lies inside its parent's box and every leaf box equals the bounds of its
// triangle's three vertices (see cpt_ctx::tree_nested).  The root has no box of its own.
static bool tree_is_nested(const cpt_scene_desc* s) {
    auto inside = [](const double* in, const double* out) {
        for (int k = 0; k < 3; ++k)
            if (!(in[k] >= out[k]) || !(in[k + 3] <= out[k + 3])) return false;
        return true;
    };
    for (int k = 0; k < s->n_nodes; ++k) {
        const cpt_node& nd = s->nodes[k];
        const int refs[2] = {nd.lref, nd.rref};
        const double* boxes[2] = {nd.lbox, nd.rbox};
        for (int q = 0; q < 2; ++q) {
            if (refs[q] >= 0) {
                const cpt_node& ch = s->nodes[refs[q]];
                if (!inside(ch.lbox, boxes[q]) || !inside(ch.rbox, boxes[q])) return false;
            } else {
                const cpt_tri& t = s->tris[~refs[q]];
                for (int a = 0; a < 3; ++a) {
                    const double lo = std::min(t.a[a], std::min(t.b[a], t.c[a])), hi = std::max(t.a[a], std::max(t.b[a], t.c[a]));
                    if (boxes[q][a] != lo || boxes[q][a + 3] != hi) return false;
                }
            }
        }
    }
    return true;
}

static int validate_scene(cpt_ctx* c, const cpt_scene_desc* s, int* measured_height) {
    int rc = validate_triangles(c, s);
    if (rc != CPT_OK) return rc;
    if (s->n_nodes != s->n_tris - 1) return fail(c, CPT_EINVAL, "n_nodes (%d) != n_tris - 1 (%d)", s->n_nodes, s->n_tris - 1);
    if (s->n_nodes > 0 && !s->nodes) return fail(c, CPT_EINVAL, "nodes missing");
    if (s->n_tree_nodes != 2 * s->n_tris - 1) return fail(c, CPT_EINVAL, "n_tree_nodes must be 2*n_tris-1");
    std::vector<unsigned char> seen_tri((size_t)s->n_tris, 0), seen_node((size_t)std::max(s->n_nodes, 1), 0);
    for (int k = 0; k < s->n_nodes; ++k) {
        for (int q = 0; q < 6; ++q)  // the division-free box test assumes finite planes
            if (!std::isfinite(s->nodes[k].lbox[q]) || !std::isfinite(s->nodes[k].rbox[q]))
                return fail(c, CPT_EINVAL, "node %d: box plane is not finite", k);
        const int refs[2] = {s->nodes[k].lref, s->nodes[k].rref};
        const int ids[2] = {s->nodes[k].lid, s->nodes[k].rid};
        for (int q = 0; q < 2; ++q) {
            if (ids[q] < 0 || ids[q] >= s->n_tree_nodes) return fail(c, CPT_EINVAL, "node %d: canonical id out of range", k);
            if (refs[q] >= 0) {
                if (refs[q] <= k || refs[q] >= s->n_nodes) return fail(c, CPT_EINVAL, "node %d: child record %d not after its parent", k, refs[q]);
                if (seen_node[refs[q]]++) return fail(c, CPT_EINVAL, "node record %d has two parents", refs[q]);
            } else {
                int t = ~refs[q];
                if (t < 0 || t >= s->n_tris) return fail(c, CPT_EINVAL, "node %d: leaf triangle out of range", k);
                if (seen_tri[t]++) return fail(c, CPT_EINVAL, "triangle %d is in two leaves", t);
            }
        }
    }
    // The traversal stacks are sized for a maximum tree height; measure it rather than trust the
    // caller's figure (records come after their parents, so one forward pass finds every depth).
    std::vector<int> level((size_t)std::max(s->n_nodes, 1), 1);
    int levels = s->n_nodes > 0 ? 1 : 0;  // levels of internal nodes
    for (int k = 0; k < s->n_nodes; ++k) {
        const int refs[2] = {s->nodes[k].lref, s->nodes[k].rref};
        for (int q = 0; q < 2; ++q)
            if (refs[q] >= 0) { level[(size_t)refs[q]] = level[(size_t)k] + 1; levels = std::max(levels, level[(size_t)k] + 1); }
    }
    const int height = levels + 1;  // plus the leaf level
    *measured_height = height;
    if (height - 2 > CPT_STACK_ENTRIES_HOST || s->tree_height - 2 > CPT_STACK_ENTRIES_HOST)
        return fail(c, CPT_ELIMIT, "tree height %d needs a %d-entry stack; limit is %d", std::max(height, s->tree_height),
                    std::max(height, s->tree_height) - 2, CPT_STACK_ENTRIES_HOST);
    return CPT_OK;
}

// One shading record per triangle: geometry, vertex normals and a copy of its material.  The material part
// (lobe probabilities, log kt) is computed here, once per material; a kernel then writes the per-triangle
// records from the triangles already on the device and the vertex normals.
static int upload_shading(cpt_ctx* c, const cpt_scene_desc* s) {
    std::vector<ShadeRec> by_material((size_t)s->n_materials);
    for (int i = 0; i < s->n_materials; ++i) {
        const cpt_material& m = s->materials[i];
        ShadeRec& r = by_material[(size_t)i];
        memset(&r, 0, sizeof r);
        r.mat_flags = m.flags;
        const LobeProb pr = lobe_probabilities(m.kd, m.ks, m.kt);
        r.p_diff = pr.diff; r.p_spec = pr.spec; r.inv_p_trans = pr.trans;  // the kernel turns p_trans into its reciprocal
        r.index = m.index;
        for (int k = 0; k < 3; ++k) {
            r.ke[k] = m.ke[k]; r.diff_w[k] = m.kd[k]; r.ks[k] = m.ks[k]; r.kr[k] = m.kr[k]; r.kt[k] = m.kt[k];  // diff_w: kd for now
            r.log_kt[k] = std::log(std::fabs(m.kt[k]));  // -inf for kt = 0 (kt^t = exp(-inf) = 0); the kernel sees to the sign of a negative kt
        }
    }
    void *d_mat = nullptr, *d_normals = nullptr;
    hipError_t e = hipMalloc(&d_mat, sizeof(ShadeRec) * by_material.size());
    if (e == hipSuccess) e = hipMalloc(&d_normals, sizeof(cpt_tri_normals) * (size_t)s->n_tris);
    if (e == hipSuccess) e = hipMemcpy(d_mat, by_material.data(), sizeof(ShadeRec) * by_material.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_normals, s->tri_normals, sizeof(cpt_tri_normals) * (size_t)s->n_tris, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_build_shade(c->stream, (const cpt_tri*)c->d_tris, (const cpt_tri_normals*)d_normals, (const ShadeRec*)d_mat, s->n_tris,
                           (ShadeRec*)c->d_shade);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_mat);
    (void)hipFree(d_normals);
    if (e != hipSuccess) return fail(c, CPT_EHIP, "shading records: %s", hipGetErrorString(e));
    return CPT_OK;
}

// Binary32 copies of the node records (boxes rounded outwards) for the extend variants that test inner nodes in
// binary32, derived on the device from the records in place.  Also (re)allocates the shared-origin copies, whose
// contents wait for a render to name the eye: every scene-changing call ends here.
static int make_nodes32(cpt_ctx* c, int n_nodes) {
    c->eye_current = false;
    if (!c->d_nodes_eye) HIP_TRY(c, hipMalloc(&c->d_nodes_eye, sizeof(cpt_node) * (size_t)std::max(n_nodes, 1)));
    if (!c->d_tris_eye) HIP_TRY(c, hipMalloc(&c->d_tris_eye, sizeof(cpt_tri) * (size_t)std::max(c->sc.n_tris, 1)));
    c->sc.nodes_eye = (const cpt_node*)c->d_nodes_eye;
    c->sc.tris_eye = (const cpt_tri*)c->d_tris_eye;
    if (c->d_nodes32) { (void)hipFree(c->d_nodes32); c->d_nodes32 = nullptr; }
    HIP_TRY(c, hipMalloc(&c->d_nodes32, 64 * (size_t)std::max(n_nodes, 1)));
    launch_nodes32(c->stream, (const cpt_node*)c->d_nodes_pw, n_nodes, c->d_nodes32);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sc.nodes32 = c->d_nodes32;
    return CPT_OK;
}

int cpt_upload_scene(cpt_ctx* c, const cpt_scene_desc* s) {
    if (!c || !s) return CPT_EINVAL;
    int measured = 0;
    int rc = validate_scene(c, s, &measured);
    if (rc != CPT_OK) return rc;
    const bool nested = tree_is_nested(s);  // references were validated above
    OrderedScene ordered;
    s = order_scene(c, s, true, true, ordered);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    size_t nb = sizeof(cpt_node) * (size_t)std::max(s->n_nodes, 1);
    HIP_TRY(c, hipMalloc(&c->d_nodes, nb));
    HIP_TRY(c, hipMalloc(&c->d_tris, sizeof(cpt_tri) * (size_t)s->n_tris));
    HIP_TRY(c, hipMalloc(&c->d_shade, sizeof(ShadeRec) * (size_t)s->n_tris));
    int stack_need = 0;
    HIP_TRY(c, hipMalloc(&c->d_nodes_pw, nb));
    if (s->n_nodes > 0) {
        HIP_TRY(c, hipMemcpy(c->d_nodes, s->nodes, sizeof(cpt_node) * (size_t)s->n_nodes, hipMemcpyHostToDevice));
        std::vector<cpt_node> nodes(s->nodes, s->nodes + s->n_nodes);
        stack_need = minimise_stack(nodes);
        HIP_TRY(c, hipMemcpy(c->d_nodes_pw, nodes.data(), sizeof(cpt_node) * (size_t)s->n_nodes, hipMemcpyHostToDevice));
    }
    {   // The tie-break rank of a triangle is the position of its leaf's box test in the reference's
        // traversal order: 2 * (expansion order of the parent) + side, where nodes are expanded in
        // depth-first, left-first order.  Derived here from the node table, whatever order the records
        // are stored in.
        std::vector<cpt_tri> tris(s->tris, s->tris + s->n_tris);
        const bool keep_ranks = (s->flags & CPT_SCENE_KEEP_RANKS) != 0;  // the caller's ranks belong to another tree on purpose
        if (!keep_ranks) for (cpt_tri& t : tris) t.test_rank = 0;
        std::vector<int> todo;
        if (s->n_nodes > 0 && !keep_ranks) todo.push_back(0);
        uint32_t order = 0;
        while (!todo.empty()) {
            int k = todo.back();
            todo.pop_back();
            const cpt_node& nd = s->nodes[k];
            if (nd.lref < 0) tris[(size_t)(~nd.lref)].test_rank = 2u * order;
            if (nd.rref < 0) tris[(size_t)(~nd.rref)].test_rank = 2u * order + 1u;
            if (nd.rref >= 0) todo.push_back(nd.rref);
            if (nd.lref >= 0) todo.push_back(nd.lref);
            ++order;
        }
        HIP_TRY(c, hipMemcpy(c->d_tris, tris.data(), sizeof(cpt_tri) * (size_t)s->n_tris, hipMemcpyHostToDevice));
    }
    rc = upload_shading(c, s);
    if (rc != CPT_OK) return rc;
    c->sc.nodes = (const cpt_node*)c->d_nodes;
    c->sc.nodes_pw = (const cpt_node*)c->d_nodes_pw;
    c->sc.tris = (const cpt_tri*)c->d_tris;
    c->sc.shade = (const ShadeRec*)c->d_shade;
    c->sc.cam = s->camera;
    c->sc.n_nodes = s->n_nodes;
    c->sc.n_tris = s->n_tris;
    c->n_tree_nodes = s->n_tree_nodes;
    c->tree_height = s->tree_height;
    c->measured_height = measured;
    c->stack_need = stack_need;
    rc = make_nodes32(c, s->n_nodes);
    if (rc != CPT_OK) return rc;
    commit_order(c, ordered);
    c->tree_nested = nested;
    c->caller_ranks = (s->flags & CPT_SCENE_KEEP_RANKS) != 0;
    c->has_scene = true;
    return CPT_OK;
}

// Height limits shared by every traversal kernel: the per-lane LDS stack of the hooks and the per-pixel kernel.
#define CPT_MAX_TREE_HEIGHT (CPT_STACK_ENTRIES_HOST + 2)

int cpt_build_scene(cpt_ctx* c, const cpt_scene_desc* s) {
    if (!c || !s) return CPT_EINVAL;
    int rc = validate_triangles(c, s);
    if (rc != CPT_OK) return rc;
    if (s->n_tris > (1 << 28)) return fail(c, CPT_ELIMIT, "device BVH build takes at most 2^28 triangles, got %d", s->n_tris);
    for (int i = 0; i < s->n_tris; ++i)  // Morton codes and box unions need finite vertices
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(s->tris[i].a[k]) || !std::isfinite(s->tris[i].b[k]) || !std::isfinite(s->tris[i].c[k]))
                return fail(c, CPT_EINVAL, "triangle %d: vertex is not finite", i);
    OrderedScene ordered;
    s = order_scene(c, s, true, false, ordered);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    const int n = s->n_tris;
    HIP_TRY(c, hipMalloc(&c->d_nodes, sizeof(cpt_node) * (size_t)std::max(n - 1, 1)));
    HIP_TRY(c, hipMalloc(&c->d_tris, sizeof(cpt_tri) * (size_t)n));
    HIP_TRY(c, hipMalloc(&c->d_shade, sizeof(ShadeRec) * (size_t)n));
    // test_rank is written by the build; with one triangle there are no ties to break
    HIP_TRY(c, hipMemcpy(c->d_tris, s->tris, sizeof(cpt_tri) * (size_t)n, hipMemcpyHostToDevice));
    rc = upload_shading(c, s);
    if (rc != CPT_OK) return rc;
    int height = 1;
    if (n >= 2) {
        void* work = nullptr;
        HIP_TRY(c, hipMalloc(&work, lbvh_work_bytes(n)));
        // Ten Morton bits per axis first; a tree too tall for the traversal stacks (long runs of nearly
        // coincident triangles) is rebuilt with coarser codes, which trade depth for balance: with no Morton bits
        // at all the tree is the balanced one over the input order.
        hipError_t e = hipSuccess;
        for (int bits = 10; bits >= 0; bits -= 2) {
            e = launch_lbvh_build(c->stream, (cpt_tri*)c->d_tris, n, (cpt_node*)c->d_nodes, work, bits, &height,
                                  (s->flags & CPT_SCENE_KEEP_RANKS) != 0);
            if (e != hipSuccess || height <= CPT_MAX_TREE_HEIGHT) break;
        }
        (void)hipFree(work);
        if (e != hipSuccess) return fail(c, CPT_EHIP, "device BVH build: %s", hipGetErrorString(e));
        if (height > CPT_MAX_TREE_HEIGHT)
            return fail(c, CPT_ELIMIT, "tree height %d needs a %d-entry stack; limit is %d", height, height - 2, CPT_STACK_ENTRIES_HOST);
    }
    // The persistent kernel's copy with the children in stack-minimal order is made on the host (one pass over the
    // records); for very large trees the round trip is not worth it and the tree is walked as built, its height bounding the stack.
    c->d_nodes_pw = c->d_nodes;
    int stack_need = height;
    if (n - 1 > 0 && n - 1 <= (1 << 18)) {
        std::vector<cpt_node> built((size_t)(n - 1));
        HIP_TRY(c, hipMemcpy(built.data(), c->d_nodes, sizeof(cpt_node) * built.size(), hipMemcpyDeviceToHost));
        stack_need = minimise_stack(built);
        void* pw = nullptr;
        HIP_TRY(c, hipMalloc(&pw, sizeof(cpt_node) * built.size()));
        c->d_nodes_pw = pw;
        HIP_TRY(c, hipMemcpy(c->d_nodes_pw, built.data(), sizeof(cpt_node) * built.size(), hipMemcpyHostToDevice));
    }
    c->sc.nodes = (const cpt_node*)c->d_nodes;
    c->sc.nodes_pw = (const cpt_node*)c->d_nodes_pw;
    c->sc.tris = (const cpt_tri*)c->d_tris;
    c->sc.shade = (const ShadeRec*)c->d_shade;
    c->sc.cam = s->camera;
    c->sc.n_nodes = n - 1;
    c->sc.n_tris = n;
    c->n_tree_nodes = 2 * n - 1;
    c->tree_height = height;
    c->measured_height = height;
    c->stack_need = stack_need;
    rc = make_nodes32(c, n - 1);
    if (rc != CPT_OK) return rc;
    commit_order(c, ordered);
    c->tree_nested = true;  // the build's boxes are exact unions, leaf boxes exact bounds
    c->caller_ranks = (s->flags & CPT_SCENE_KEEP_RANKS) != 0;
    c->has_scene = true;
    return CPT_OK;
}

int cpt_refit_scene(cpt_ctx* c, const cpt_scene_desc* s) {
    if (!c || !s) return CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "cpt_refit_scene: no scene uploaded");
    int rc = validate_triangles(c, s);
    if (rc != CPT_OK) return rc;
    if (s->n_tris != c->sc.n_tris) return fail(c, CPT_EINVAL, "refit needs the same %d triangles, got %d", c->sc.n_tris, s->n_tris);
    const int n = s->n_tris, m = c->sc.n_nodes;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(s->tris[i].a[k]) || !std::isfinite(s->tris[i].b[k]) || !std::isfinite(s->tris[i].c[k]))
                return fail(c, CPT_EINVAL, "triangle %d: vertex is not finite", i);
    OrderedScene ordered;
    s = order_scene(c, s, false, false, ordered);  // the order chosen when the tree in place was made
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // levels of the records of the tree in place (children are stored after their parents, so one forward pass
    // finds every depth), then the records listed level by level
    std::vector<int> order((size_t)m), level_start;
    if (m > 0) {
        std::vector<cpt_node> nodes((size_t)m);
        HIP_TRY(c, hipMemcpy(nodes.data(), c->d_nodes, sizeof(cpt_node) * (size_t)m, hipMemcpyDeviceToHost));
        std::vector<int> level((size_t)m, 0);
        int n_levels = 1;
        for (int k = 0; k < m; ++k) {
            const int refs[2] = {nodes[(size_t)k].lref, nodes[(size_t)k].rref};
            for (int q = 0; q < 2; ++q)
                if (refs[q] >= 0) { level[(size_t)refs[q]] = level[(size_t)k] + 1; n_levels = std::max(n_levels, level[(size_t)k] + 2); }
        }
        level_start.assign((size_t)n_levels + 1, 0);
        for (int k = 0; k < m; ++k) level_start[(size_t)level[(size_t)k] + 1]++;
        for (int l = 0; l < n_levels; ++l) level_start[(size_t)l + 1] += level_start[(size_t)l];
        std::vector<int> at(level_start.begin(), level_start.end() - 1);
        for (int k = 0; k < m; ++k) order[(size_t)at[(size_t)level[(size_t)k]]++] = k;
    }
    void *d_fresh = nullptr, *d_order = nullptr;
    hipError_t e = hipMalloc(&d_fresh, sizeof(cpt_tri) * (size_t)n);
    if (e == hipSuccess && m > 0) e = hipMalloc(&d_order, sizeof(int) * (size_t)m);
    if (e == hipSuccess) e = hipMemcpy(d_fresh, s->tris, sizeof(cpt_tri) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess && m > 0) e = hipMemcpy(d_order, order.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_update_tris(c->stream, (const cpt_tri*)d_fresh, n, (cpt_tri*)c->d_tris);
        if (m > 0) launch_refit(c->stream, (const int*)d_order, level_start.data(), (int)level_start.size() - 1, (const cpt_tri*)c->d_tris, (cpt_node*)c->d_nodes);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_fresh);
    (void)hipFree(d_order);
    if (e != hipSuccess) return fail(c, CPT_EHIP, "refit: %s", hipGetErrorString(e));
    rc = upload_shading(c, s);
    if (rc != CPT_OK) return rc;
    if (m > 0 && c->d_nodes_pw != c->d_nodes) {  // the walk's copy of the records: same child order as before, new boxes
        std::vector<cpt_node> fresh((size_t)m);
        HIP_TRY(c, hipMemcpy(fresh.data(), c->d_nodes, sizeof(cpt_node) * (size_t)m, hipMemcpyDeviceToHost));
        c->stack_need = minimise_stack(fresh);  // the need depends on the topology alone, so the order comes out the same
        HIP_TRY(c, hipMemcpy(c->d_nodes_pw, fresh.data(), sizeof(cpt_node) * (size_t)m, hipMemcpyHostToDevice));
    }
    rc = make_nodes32(c, m);  // the boxes changed
    if (rc != CPT_OK) return rc;
    c->sc.cam = s->camera;
    c->tree_nested = true;  // every box is now the exact union of what lies below it
    return CPT_OK;
}

int cpt_download_tree(cpt_ctx* c, cpt_node* h_nodes, int32_t node_capacity, uint32_t* h_test_rank, int32_t tri_capacity,
                      int32_t* n_nodes, int32_t* height) {
    if (!c) return CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "no scene uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_nodes) *n_nodes = c->sc.n_nodes;
    if (height) *height = c->measured_height;
    if (h_nodes) {
        if (node_capacity < c->sc.n_nodes) return fail(c, CPT_EINVAL, "node buffer holds %d records, the tree has %d", node_capacity, c->sc.n_nodes);
        if (c->sc.n_nodes > 0)
            HIP_TRY(c, hipMemcpy(h_nodes, c->d_nodes, sizeof(cpt_node) * (size_t)c->sc.n_nodes, hipMemcpyDeviceToHost));
        if (!c->tri_ext.empty())  // leaves name triangles by the caller's indices
            for (int k = 0; k < c->sc.n_nodes; ++k) {
                if (h_nodes[k].lref < 0) h_nodes[k].lref = ~c->tri_ext[(size_t)(~h_nodes[k].lref)];
                if (h_nodes[k].rref < 0) h_nodes[k].rref = ~c->tri_ext[(size_t)(~h_nodes[k].rref)];
            }
    }
    if (h_test_rank) {
        if (tri_capacity < c->sc.n_tris) return fail(c, CPT_EINVAL, "rank buffer holds %d entries, the scene has %d triangles", tri_capacity, c->sc.n_tris);
        std::vector<cpt_tri> tris((size_t)c->sc.n_tris);
        HIP_TRY(c, hipMemcpy(tris.data(), c->d_tris, sizeof(cpt_tri) * tris.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tris.size(); ++i) h_test_rank[c->tri_ext.empty() ? i : (size_t)c->tri_ext[i]] = tris[i].test_rank;
    }
    return CPT_OK;
}

// Lay the arrays of one batch (two ray streams, hit records, per-slot radiance) out in `base`, which holds
// slots * CPT_SLOT_BYTES bytes and is 256-byte aligned; slots is a multiple of 64, so every array stays 256-byte aligned.
static void carve_paths(cpt_ctx::Lane& l, void* base, size_t slots) {
    double* p = (double*)base;
    for (int b = 0; b < 2; ++b) {
        cpt_v2d** f[CPT_STREAM_DOUBLES / 2] = {&l.R[b].oxy, &l.R[b].ozdx, &l.R[b].dyz, &l.R[b].mdax, &l.R[b].ayz};
        for (int k = 0; k < CPT_STREAM_DOUBLES / 2; ++k) { *f[k] = (cpt_v2d*)p; p += 2 * slots; }
    }
    l.H.rec = (HitRec*)p; p += 2 * slots;
    l.C.cx = p; p += slots;
    l.C.cy = p; p += slots;
    l.C.cz = p; p += slots;
    uint32_t* u = (uint32_t*)p;
    l.R[0].slot = u; u += slots;
    l.R[1].slot = u; u += slots;
    l.C.lit = (uint8_t*)u;
    l.path_capacity = slots;
}

static int ensure_paths(cpt_ctx* c, cpt_ctx::Lane& l, hipStream_t st, size_t slots) {
    if (slots <= l.path_capacity) return CPT_OK;
    HIP_TRY(c, hipStreamSynchronize(st));
    free_paths(l);
    slots = (slots + 63) / 64 * 64;
    HIP_TRY(c, hipMalloc(&l.d_paths, slots * CPT_SLOT_BYTES));
    carve_paths(l, l.d_paths, slots);
    return CPT_OK;
}

static int ensure_spill(cpt_ctx* c, cpt_ctx::Lane& l, hipStream_t st) {
    if (c->extend_variant < 1 || l.spill_blocks >= c->pw_blocks) return CPT_OK;
    HIP_TRY(c, hipStreamSynchronize(st));
    if (l.d_spill) (void)hipFree(l.d_spill);
    l.d_spill = nullptr;
    HIP_TRY(c, hipMalloc((void**)&l.d_spill, (size_t)(CPT_PW_STACK_MAX - CPT_PW_STACK_LDS_HOST) * (size_t)c->pw_blocks * CPT_PW_THREADS_HOST * sizeof(uint32_t)));
    l.spill_blocks = c->pw_blocks;
    return CPT_OK;
}

// One image-size rule for every entry point that takes a width and a height: at most 2^28 pixels, also counted in
// whole 8x8 tiles (what path slots enumerate, and what fast_div is exact for).
static int check_image_size(cpt_ctx* c, int w, int h) {
    if (w < 1 || h < 1) return fail(c, CPT_EINVAL, "bad image size %dx%d", w, h);
    if ((long long)w * h > (1ll << 28) || (long long)((w + 7) / 8) * ((h + 7) / 8) * 64 > (1ll << 28))
        return fail(c, CPT_ELIMIT, "image %dx%d too large: at most 2^28 pixels", w, h);
    return CPT_OK;
}

static int make_params(cpt_ctx* c, const cpt_render_params* p, DParams& P) {
    int size_rc = check_image_size(c, p->width, p->height);
    if (size_rc != CPT_OK) return size_rc;
    if (p->sample_begin < 0 || p->sample_end < p->sample_begin) return fail(c, CPT_EINVAL, "bad sample range [%d, %d)", p->sample_begin, p->sample_end);
    if (p->depth < 0 || p->depth > CPT_MAX_DEPTH) return fail(c, CPT_EINVAL, "depth %d not in [0, %d]", p->depth, CPT_MAX_DEPTH);
    if (p->mode != CPT_MODE_WAVEFRONT && p->mode != CPT_MODE_MEGAKERNEL) return fail(c, CPT_EINVAL, "unknown mode %d", p->mode);
    P.w = p->width; P.h = p->height; P.s0 = p->sample_begin; P.s1 = p->sample_end; P.depth = p->depth;
    P.dof = p->dof ? 1 : 0; P.rr = p->russian_roulette ? 1 : 0;
    P.seed_lo = (uint32_t)p->seed; P.seed_hi = (uint32_t)(p->seed >> 32);
    P.focal_distance = p->focal_distance; P.aperture = p->aperture;
    P.tiles_x = (p->width + 7) / 8; P.tiles_y = (p->height + 7) / 8;
    const int n_tiles = P.tiles_x * P.tiles_y;
    if (p->tile_count < 0 || p->tile_begin < 0 || (p->tile_count == 0 && p->tile_begin != 0) ||
        (long long)p->tile_begin + p->tile_count > n_tiles)
        return fail(c, CPT_EINVAL, "bad tile window [%d, +%d) of %d tiles", p->tile_begin, p->tile_count, n_tiles);
    P.tile0 = p->tile_begin;
    P.n_pix_slots = (p->tile_count > 0 ? p->tile_count : n_tiles) * 64;
    P.by_pix_slots = fast_div_for((uint32_t)P.n_pix_slots);
    P.by_tiles_x = fast_div_for((uint32_t)P.tiles_x);
    return CPT_OK;
}

// Pixels of the image that lie in the tile window of P (edge tiles are partly outside the image).
static long long window_pixels(const DParams& P) {
    const int n_tiles = P.n_pix_slots / 64;
    if (P.tile0 == 0 && n_tiles == P.tiles_x * P.tiles_y) return (long long)P.w * P.h;
    long long n = 0;
    for (int t = P.tile0; t < P.tile0 + n_tiles; ++t) {
        const int tx = t % P.tiles_x, ty = t / P.tiles_x;
        n += (long long)std::min(8, P.w - tx * 8) * std::min(8, P.h - ty * 8);
    }
    return n;
}

static int need_events(cpt_ctx* c, size_t n) {
    while (c->events.size() < n) {
        hipEvent_t ev;
        HIP_TRY(c, hipEventCreate(&ev));
        c->events.push_back(ev);
    }
    return CPT_OK;
}

static int collect_work(cpt_ctx* c) {
    unsigned long long w[8];
    HIP_TRY(c, hipMemcpyAsync(w, c->d_work, sizeof w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_work, 0, sizeof w, c->stream));
    c->stats.segments += w[1];
    c->stats.hits += w[2];
    c->stats.node_steps += w[3];
    c->stats.tri_tests += w[4];
    c->stats.exact_node_steps += w[5];
    c->stats.box_passes += w[6];
    c->stats.leaf_passes += w[7];
    return CPT_OK;
}

// The wavefront pipeline over samples [P.s0, P.s1) in batches of `batch` whole samples, dealt in turn to the
// n_lanes prepared lanes (lane k on stream st[k]; st[0] is the caller's stream).  Allocates nothing unless
// `timed` needs more events than the context holds.
static int run_wavefront(cpt_ctx* c, const DParams& P, LaunchCfg L, cpt_ctx::Lane* const* lanes, const hipStream_t* st, int n_lanes,
                         int batch, bool timed, unsigned long long* work, double* d_accum) {
    const long long per_sample = P.n_pix_slots;
    int rc;
    const size_t n_ev = 2 * (size_t)(P.depth + 1) + 3;
    if (timed) { rc = need_events(c, n_ev); if (rc) return rc; }
    // First rays of a pin-hole render all start at the eye: their extend launch takes the records restated for that origin,
    // which are brought up to date here when the scene or the eye has changed since they were made.
    const double* eye = c->sc.cam.eye;
    const bool use_eye = c->pw_eye && !P.dof && c->extend_variant >= 1 && !L.pw_f32 && c->sc.n_nodes > 0 &&
                         std::fabs(eye[0]) < 1e300 && std::fabs(eye[1]) < 1e300 && std::fabs(eye[2]) < 1e300;
    if (use_eye && !(c->eye_current && eye[0] == c->eye_of[0] && eye[1] == c->eye_of[1] && eye[2] == c->eye_of[2])) {
        launch_eye_records(st[0], c->sc, (cpt_node*)c->d_nodes_eye, (cpt_tri*)c->d_tris_eye);
        HIP_TRY(c, hipGetLastError());
        for (int k = 0; k < 3; ++k) c->eye_of[k] = eye[k];
        c->eye_current = true;
    }
    if (n_lanes > 1) {  // the other lanes start after whatever the caller queued before this render
        HIP_TRY(c, hipEventRecord(c->ev_fork, st[0]));
        for (int k = 1; k < n_lanes; ++k) HIP_TRY(c, hipStreamWaitEvent(st[k], c->ev_fork, 0));
    }
    int last_lane = 0;
    for (int i = 0, bs = P.s0; bs < P.s1; bs += batch, ++i) {
        const int k = i % n_lanes;
        cpt_ctx::Lane& ln = *lanes[k];
        L.stream = st[k];
        int bn = std::min(batch, P.s1 - bs);
        long long slots = (long long)bn * per_sample;
        size_t ev = 0;
        uint32_t* cursors = ln.d_cursors;
        const uint32_t cap = cpt_segment_cap((size_t)slots);
        HIP_TRY(c, hipMemsetAsync(ln.d_counters, 0, (size_t)(P.depth + 2) * CPT_PW_CURSOR_WORDS * sizeof(uint32_t), L.stream));
        if (c->extend_variant >= 1) HIP_TRY(c, hipMemsetAsync(cursors, 0, (size_t)(P.depth + 1) * CPT_PW_CURSOR_WORDS * sizeof(uint32_t), L.stream));
        if (timed) HIP_TRY(c, hipEventRecord(c->events[ev++], L.stream));
        launch_dense_counts(L, ln.d_counters, (uint32_t)slots, cap);  // the first rays sit at positions 0 .. slots-1
        launch_generate(L, c->sc, P, ln.R[0], ln.C, bs, bn);
        if (timed) HIP_TRY(c, hipEventRecord(c->events[ev++], L.stream));
        for (int b = 0; b <= P.depth; ++b) {
            const RayStream& in = ln.R[b & 1];
            uint32_t* count_in = ln.d_counters + (size_t)b * CPT_PW_CURSOR_WORDS;
            L.pw_eye = b == 0 && use_eye;
            if (c->extend_variant >= 1) launch_extend_pw(L, c->sc, in, ln.H, count_in, cap, cursors + (size_t)b * CPT_PW_CURSOR_WORDS, ln.d_spill, work, ExtendObserve{nullptr, nullptr, 0});
            else launch_extend(L, c->sc, in, ln.H, count_in, cap, slots, work, ExtendObserve{nullptr, nullptr, 0});
            if (timed) HIP_TRY(c, hipEventRecord(c->events[ev++], L.stream));
            launch_shade(L, c->sc, P, in, ln.H, ln.R[(b + 1) & 1], ln.C, bs, b, count_in, count_in + CPT_PW_CURSOR_WORDS, cap, slots, work);
            if (timed) HIP_TRY(c, hipEventRecord(c->events[ev++], L.stream));
        }
        // a pixel's samples are added in sample order: this batch's resolve follows the previous batch's
        if (i > 0 && n_lanes > 1) HIP_TRY(c, hipStreamWaitEvent(L.stream, lanes[last_lane]->resolved, 0));
        launch_resolve(L, P, ln.C, bn, d_accum);
        HIP_TRY(c, hipGetLastError());
        if (n_lanes > 1) HIP_TRY(c, hipEventRecord(ln.resolved, L.stream));
        last_lane = k;
        c->stats.launches_generate++;
        c->stats.launches_extend += (uint64_t)(P.depth + 1);
        c->stats.launches_shade += (uint64_t)(P.depth + 1);
        c->stats.iterations += (uint64_t)(P.depth + 1);
        if (timed) {
            HIP_TRY(c, hipEventRecord(c->events[ev++], L.stream));
            HIP_TRY(c, hipEventSynchronize(c->events[ev - 1]));
            float ms = 0;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->events[0], c->events[1]));
            c->stats.ms_generate += ms;
            for (int b = 0; b <= P.depth; ++b) {
                HIP_TRY(c, hipEventElapsedTime(&ms, c->events[1 + 2 * b], c->events[2 + 2 * b]));
                c->stats.ms_extend += ms;
                HIP_TRY(c, hipEventElapsedTime(&ms, c->events[2 + 2 * b], c->events[3 + 2 * b]));
                c->stats.ms_shade += ms;
            }
            HIP_TRY(c, hipEventElapsedTime(&ms, c->events[0], c->events[ev - 1]));
            c->stats.ms_total += ms;
        }
    }
    // the resolves are chained, so the last one completes the render; the caller's stream waits for it
    if (n_lanes > 1 && last_lane != 0) HIP_TRY(c, hipStreamWaitEvent(st[0], lanes[last_lane]->resolved, 0));
    return CPT_OK;
}

// Lanes with a parked leaf that make a wave run its triangle pass.  The reference walk's work is fixed, so it waits for
// half the wave (fuller passes); the culling walks want a closest hit early, because every box beyond it is then skipped
// (measured on the 10 000-triangle scene over the surface-area tree: 16 is 3 % faster than 32, 8 the same as 16).
static int leaf_go_for(const cpt_ctx* c) { return c->pw_leaf_go > 0 ? c->pw_leaf_go : (variant_cull(c) ? 16 : 32); }

// Node expansions per lane between two looks at the triangle phase.  A turn of the kernel's loop runs at most one triangle
// pass, so the ratio fixes how many triangle tests per node step the loop can absorb before lanes wait with parked leaves: on
// the reference's trees (4 node steps per triangle test on the bundled scenes) three expansions per turn are best, on the
// surface-area tree's shorter walks (3 per test) two are (cfg2 +3.5 %, teapot +2 % there; 1.5-3 % worse on the reference tree).
static int box_repeat_for(const cpt_ctx* c) { return c->pw_box_repeat ? c->pw_box_repeat : (c->caller_ranks ? 2 : 3); }

static LaunchCfg launch_cfg(const cpt_ctx* c) {
    return LaunchCfg{c->stream, c->max_blocks, c->pw_blocks, chunk_for(c), c->pw_refill, scene_mode(c), variant_cull(c), variant_f32(c), c->pw_full_go, leaf_go_for(c), box_repeat_for(c), 0};
}

int cpt_render(cpt_ctx* c, const cpt_render_params* p, double* d_accum) {
    if (!c || !p || !d_accum) return c ? fail(c, CPT_EINVAL, "cpt_render: NULL argument") : CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "cpt_render: no scene uploaded");
    DParams P;
    int rc = make_params(c, p, P);
    if (rc != CPT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (P.s1 == P.s0) return CPT_OK;
    LaunchCfg L = launch_cfg(c);
    unsigned long long* work = c->count_work ? c->d_work : nullptr;
    const bool timed = c->time_kernels;
    long long n_paths = window_pixels(P) * (P.s1 - P.s0);

    if (p->mode == CPT_MODE_MEGAKERNEL) {
        if (timed) { rc = need_events(c, 2); if (rc) return rc; HIP_TRY(c, hipEventRecord(c->events[0], c->stream)); }
        launch_megakernel(L, c->sc, P, d_accum, work);
        HIP_TRY(c, hipGetLastError());
        if (timed) {
            HIP_TRY(c, hipEventRecord(c->events[1], c->stream));
            HIP_TRY(c, hipEventSynchronize(c->events[1]));
            float ms = 0;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->events[0], c->events[1]));
            c->stats.ms_mega += ms;
            c->stats.ms_total += ms;
        }
        c->stats.launches_mega++;
        c->stats.paths += (uint64_t)n_paths;
        if (work) return collect_work(c);
        return mark_tail(c);
    }

    // wavefront: batches of whole samples, dealt to the lanes in turn
    const long long per_sample = P.n_pix_slots;
    const int total = P.s1 - P.s0;
    int batch = (int)std::max(1ll, std::min((long long)total, c->batch_paths / per_sample));
    int n_lanes = timed ? 1 : c->n_lanes;  // per-kernel times are only meaningful when nothing else runs beside the kernel
    if (n_lanes > 1) {  // a render that fits one batch is split so that there is something to overlap, unless it is tiny
        const int share = (total + n_lanes - 1) / n_lanes;
        if ((long long)share * per_sample >= (1ll << 20)) batch = std::min(batch, share);
    }
    const int n_batches = (total + batch - 1) / batch;
    n_lanes = std::min(n_lanes, n_batches);
    cpt_ctx::Lane* lanes[CPT_MAX_LANES];
    hipStream_t st[CPT_MAX_LANES];
    for (int k = 0; k < n_lanes; ++k) {
        lanes[k] = &c->lanes[k];
        st[k] = lane_stream(c, k);
        rc = ensure_spill(c, c->lanes[k], st[k]);
        if (rc) return rc;
        rc = ensure_paths(c, c->lanes[k], st[k], cpt_stream_positions((size_t)batch * (size_t)per_sample));
        if (rc) return rc;
    }
    rc = run_wavefront(c, P, L, lanes, st, n_lanes, batch, timed, work, d_accum);
    if (rc) return rc;
    c->stats.paths += (uint64_t)n_paths;
    if (work) return collect_work(c);
    return mark_tail(c);
}

// Stack-spill columns the persistent extend kernel can touch on this context's scene: one per stack entry beyond
// the LDS part, for every thread of every persistent workgroup.
static size_t spill_bytes_needed(const cpt_ctx* c) {
    if (c->extend_variant < 1) return 0;
    const int entries = std::max(c->measured_height, c->tree_height) - 2 - CPT_PW_STACK_LDS_HOST;
    return entries > 0 ? (size_t)entries * (size_t)c->pw_blocks * CPT_PW_THREADS_HOST * sizeof(uint32_t) : 0;
}

int cpt_render_noalloc(cpt_ctx* c, const cpt_render_params* p, double* d_accum) {
    if (!c || !p || !d_accum) return c ? fail(c, CPT_EINVAL, "cpt_render_noalloc: NULL argument") : CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "cpt_render_noalloc: no scene uploaded");
    if (p->mode != CPT_MODE_WAVEFRONT) return fail(c, CPT_EINVAL, "cpt_render_noalloc: wavefront mode only");
    DParams P;
    int rc = make_params(c, p, P);
    if (rc != CPT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (P.s1 == P.s0) return CPT_OK;
    // carve this GPU's copy of the static array: segment counters | claim cursors | spill columns | one batch of paths
    const size_t align = 256;
    auto up = [&](size_t n) { return (n + align - 1) / align * align; };
    const size_t counter_bytes = up((size_t)(P.depth + 2) * CPT_PW_CURSOR_WORDS * sizeof(uint32_t));
    const size_t spill_bytes = up(spill_bytes_needed(c));
    const size_t fixed = 2 * counter_bytes + spill_bytes;
    const size_t per_sample = cpt_stream_positions((size_t)P.n_pix_slots);  // slots one sample of the window needs
    if (fixed >= CPT_NOALLOC_SCRATCH_BYTES || (CPT_NOALLOC_SCRATCH_BYTES - fixed) / CPT_SLOT_BYTES < per_sample)
        return fail(c, CPT_ELIMIT,
                    "cpt_render_noalloc: one sample of this window needs %zu bytes of working memory (%zu path slots of %d bytes, "
                    "%zu bytes of counters and stack spill); the static scratch holds %zu: render a smaller tile window or use cpt_render",
                    fixed + per_sample * CPT_SLOT_BYTES, per_sample, (int)CPT_SLOT_BYTES, fixed, (size_t)CPT_NOALLOC_SCRATCH_BYTES);
    // whole samples per batch: what fits (a batch of n samples needs cpt_stream_positions(n * n_pix_slots) slots)
    const int total = P.s1 - P.s0;
    const size_t room = (CPT_NOALLOC_SCRATCH_BYTES - fixed) / CPT_SLOT_BYTES;
    int batch = (int)std::min<size_t>((size_t)total, room / (size_t)P.n_pix_slots);
    while (batch > 1 && cpt_stream_positions((size_t)batch * (size_t)P.n_pix_slots) > room) --batch;
    if (batch < 1) batch = 1;
    const size_t slots = cpt_stream_positions((size_t)batch * (size_t)P.n_pix_slots);
    unsigned char* base = c->d_static_scratch;  // the device address, never the array's name
    cpt_ctx::Lane ln;
    ln.d_counters = (uint32_t*)base;
    ln.d_cursors = (uint32_t*)(base + counter_bytes);
    ln.d_spill = (uint32_t*)(base + 2 * counter_bytes);
    carve_paths(ln, base + fixed, slots);
    cpt_ctx::Lane* lanes[1] = {&ln};
    hipStream_t st[1] = {c->stream};
    // no per-launch timing (it would create events) and no work counting (it would synchronise): one lane, stream-ordered
    rc = run_wavefront(c, P, launch_cfg(c), lanes, st, 1, batch, false, nullptr, d_accum);
    if (rc) return rc;
    c->stats.paths += (uint64_t)(window_pixels(P) * (P.s1 - P.s0));
    return mark_tail(c);
}

int cpt_tonemap(cpt_ctx* c, const double* d_accum, int32_t w, int32_t h, int32_t total_spp, uint8_t* d_rgb) {
    if (!c || !d_accum || !d_rgb) return c ? fail(c, CPT_EINVAL, "cpt_tonemap: NULL argument") : CPT_EINVAL;
    int size_rc = check_image_size(c, w, h);
    if (size_rc != CPT_OK) return size_rc;
    if (total_spp < 1) return fail(c, CPT_EINVAL, "cpt_tonemap: bad sample count %d", total_spp);
    HIP_TRY(c, hipSetDevice(c->device));
    LaunchCfg L = launch_cfg(c);
    launch_tonemap(L, d_accum, (long long)w * h * 3, total_spp, d_rgb);
    HIP_TRY(c, hipGetLastError());
    return mark_tail(c);
}

int cpt_render_image(cpt_ctx* c, const cpt_render_params* p, uint8_t* h_rgb) {
    if (!c || !p || !h_rgb) return c ? fail(c, CPT_EINVAL, "cpt_render_image: NULL argument") : CPT_EINVAL;
    if (p->sample_end <= p->sample_begin) return fail(c, CPT_EINVAL, "cpt_render_image: empty sample range");
    {   // every argument check before any buffer is allocated
        DParams P;
        int prc = make_params(c, p, P);
        if (prc != CPT_OK) return prc;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)p->width * p->height * 3;
    double* d_accum = nullptr;
    uint8_t* d_rgb = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_accum, n * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_rgb, n);
    if (e != hipSuccess) { (void)hipFree(d_accum); return fail(c, CPT_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
    int rc = CPT_OK;
    e = hipMemsetAsync(d_accum, 0, n * sizeof(double), c->stream);
    if (e != hipSuccess) rc = fail(c, CPT_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    if (rc == CPT_OK) rc = cpt_render(c, p, d_accum);
    if (rc == CPT_OK) rc = cpt_tonemap(c, d_accum, p->width, p->height, p->sample_end - p->sample_begin, d_rgb);
    if (rc == CPT_OK) {
        e = hipMemcpyAsync(h_rgb, d_rgb, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, CPT_EHIP, "copy back: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_accum);
    (void)hipFree(d_rgb);
    return rc;
}

// device buffer that lives for one blocking call
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};

int cpt_render_progressive(cpt_ctx* c, const cpt_render_params* p, int32_t chunk_spp, double* h_accum, int32_t accum_samples,
                           uint8_t* h_rgb, cpt_progress_fn progress, void* user, int32_t* samples_done) {
    if (samples_done) *samples_done = 0;
    if (!c || !p || !h_rgb) return c ? fail(c, CPT_EINVAL, "cpt_render_progressive: NULL argument") : CPT_EINVAL;
    if (p->sample_end <= p->sample_begin) return fail(c, CPT_EINVAL, "cpt_render_progressive: empty sample range");
    {   // every argument check before any buffer is allocated
        DParams P;
        int prc = make_params(c, p, P);
        if (prc != CPT_OK) return prc;
    }
    if (chunk_spp < 1) return fail(c, CPT_EINVAL, "cpt_render_progressive: chunk_spp must be >= 1");
    if (accum_samples < 0 || (accum_samples > 0 && !h_accum)) return fail(c, CPT_EINVAL, "cpt_render_progressive: accum_samples without h_accum");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)p->width * p->height * 3;
    DevBuf accum, rgb;
    HIP_TRY(c, accum.alloc(n * sizeof(double)));
    HIP_TRY(c, rgb.alloc(n));
    if (accum_samples > 0) HIP_TRY(c, hipMemcpyAsync(accum.p, h_accum, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    else HIP_TRY(c, hipMemsetAsync(accum.p, 0, n * sizeof(double), c->stream));
    int rc = CPT_OK;
    int done = 0;
    for (int s = p->sample_begin; s < p->sample_end && rc == CPT_OK; s += chunk_spp) {
        cpt_render_params q = *p;
        q.sample_begin = s;
        q.sample_end = std::min(p->sample_end, s + chunk_spp);
        rc = cpt_render(c, &q, (double*)accum.p);
        if (rc != CPT_OK) break;
        done += q.sample_end - q.sample_begin;
        rc = cpt_tonemap(c, (const double*)accum.p, p->width, p->height, accum_samples + done, (uint8_t*)rgb.p);
        if (rc != CPT_OK) break;
        hipError_t e = hipMemcpyAsync(h_rgb, rgb.p, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(c, CPT_EHIP, "copy back: %s", hipGetErrorString(e)); break; }
        if (progress && progress(user, accum_samples + done, h_rgb) != 0) break;
    }
    if (rc == CPT_OK && h_accum) {
        hipError_t e = hipMemcpyAsync(h_accum, accum.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(c, CPT_EHIP, "copy back: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(c->stream);
    if (samples_done) *samples_done = done;
    return rc;
}

// One render on several GPUs of this process.  Context k renders a contiguous share of the sample indices of every pixel
// (the first total % n contexts one more) from a host thread of its own into a buffer on its own GPU; the sums are then
// brought to the first context's GPU one after the other, device to device, and added there in list order, so the result
// is ((sum 0 + sum 1) + sum 2) + ... whatever the GPUs' timing.
int cpt_render_image_multi(cpt_ctx* const* ctxs, int32_t n_ctx, const cpt_render_params* p, uint8_t* h_rgb, int32_t* samples_done) {
    if (samples_done) *samples_done = 0;
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return CPT_EINVAL;
    cpt_ctx* c0 = ctxs[0];
    if (!p || !h_rgb) return fail(c0, CPT_EINVAL, "cpt_render_image_multi: NULL argument");
    if (p->sample_end <= p->sample_begin) return fail(c0, CPT_EINVAL, "cpt_render_image_multi: empty sample range");
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k]) return fail(c0, CPT_EINVAL, "cpt_render_image_multi: context %d is NULL", k);
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(c0, CPT_EINVAL, "cpt_render_image_multi: context %d is listed twice", k);
        if (!ctxs[k]->has_scene) return fail(c0, CPT_ENOSCENE, "cpt_render_image_multi: context %d has no scene", k);
        DParams P;
        int prc = make_params(ctxs[k], p, P);
        if (prc != CPT_OK) { if (k) c0->err = ctxs[k]->err; return prc; }
    }
    const size_t n = (size_t)p->width * p->height * 3;
    const int total = p->sample_end - p->sample_begin, base = total / n_ctx, extra = total % n_ctx;
    std::vector<DevBuf> sums((size_t)n_ctx);
    std::vector<int> rcs((size_t)n_ctx, CPT_OK), share((size_t)n_ctx, 0);
    auto work = [&](int k) {
        cpt_ctx* c = ctxs[k];
        cpt_render_params q = *p;
        q.sample_begin = p->sample_begin + k * base + std::min(k, extra);
        q.sample_end = q.sample_begin + base + (k < extra ? 1 : 0);
        share[(size_t)k] = q.sample_end - q.sample_begin;
        if (share[(size_t)k] == 0 && k > 0) return;
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = sums[(size_t)k].alloc(n * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(sums[(size_t)k].p, 0, n * sizeof(double), c->stream);
        if (e != hipSuccess) { rcs[(size_t)k] = fail(c, CPT_EHIP, "radiance sums: %s", hipGetErrorString(e)); return; }
        int rc = share[(size_t)k] > 0 ? cpt_render(c, &q, (double*)sums[(size_t)k].p) : CPT_OK;
        if (rc == CPT_OK) rc = cpt_synchronize(c);
        rcs[(size_t)k] = rc;
    };
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < n_ctx; ++k) pool.emplace_back(work, k);
        work(0);
        for (std::thread& t : pool) t.join();
    }
    for (int k = 0; k < n_ctx; ++k)
        if (rcs[(size_t)k] != CPT_OK) {
            if (k) c0->err = "context " + std::to_string(k) + " (GPU " + std::to_string(ctxs[k]->device) + "): " + ctxs[k]->err;
            return rcs[(size_t)k];
        }
    HIP_TRY(c0, hipSetDevice(c0->device));
    DevBuf incoming, rgb;
    if (n_ctx > 1) HIP_TRY(c0, incoming.alloc(n * sizeof(double)));
    HIP_TRY(c0, rgb.alloc(n));
    for (int k = 1; k < n_ctx; ++k) {
        if (share[(size_t)k] == 0) continue;
        // every context was synchronised above, so the source is complete; the copy and the add queue up on the first context's stream
        HIP_TRY(c0, hipMemcpyPeerAsync(incoming.p, c0->device, sums[(size_t)k].p, ctxs[k]->device, n * sizeof(double), c0->stream));
        launch_add_values(c0->stream, (double*)sums[0].p, (const double*)incoming.p, (long long)n, c0->max_blocks);
        HIP_TRY(c0, hipGetLastError());
    }
    int rc = cpt_tonemap(c0, (const double*)sums[0].p, p->width, p->height, total, (uint8_t*)rgb.p);
    if (rc != CPT_OK) return rc;
    HIP_TRY(c0, hipMemcpyAsync(h_rgb, rgb.p, n, hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(c0, hipStreamSynchronize(c0->stream));
    if (samples_done) *samples_done = total;
    return CPT_OK;
}

int cpt_synchronize(cpt_ctx* c) {
    if (!c) return CPT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CPT_OK;
}

int cpt_get_stats(cpt_ctx* c, cpt_stats* out) {
    if (!c || !out) return CPT_EINVAL;
    *out = c->stats;
    return CPT_OK;
}
int cpt_reset_stats(cpt_ctx* c) {
    if (!c) return CPT_EINVAL;
    c->stats = cpt_stats{};
    return CPT_OK;
}


// ---- parity hooks: blocking, host arrays in and out ---------------------------------------------
int cpt_trace_rays(cpt_ctx* c, int32_t n, const double* h_o, const double* h_d, int32_t* h_tri, double* h_t,
                   double* h_normal, uint64_t* h_vhash, int32_t* h_counts, uint8_t* h_bitmap, int32_t stride) {
    if (!c || !h_o || !h_d || !h_tri || !h_t || n < 0) return c ? fail(c, CPT_EINVAL, "cpt_trace_rays: bad argument") : CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "cpt_trace_rays: no scene uploaded");
    if (h_bitmap && stride * 8 < c->n_tree_nodes) return fail(c, CPT_EINVAL, "bitmap_stride %d too small for %d nodes", stride, c->n_tree_nodes);
    if (n == 0) return CPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf o, d, tri, t, nrm, vh, cnt, bm;
    size_t N = (size_t)n;
    HIP_TRY(c, o.alloc(N * 24)); HIP_TRY(c, d.alloc(N * 24)); HIP_TRY(c, tri.alloc(N * 4)); HIP_TRY(c, t.alloc(N * 8));
    HIP_TRY(c, nrm.alloc(N * 24)); HIP_TRY(c, vh.alloc(N * 8)); HIP_TRY(c, cnt.alloc(N * 16));
    if (h_bitmap) { HIP_TRY(c, bm.alloc(N * (size_t)stride)); HIP_TRY(c, hipMemsetAsync(bm.p, 0, N * (size_t)stride, c->stream)); }
    HIP_TRY(c, hipMemcpyAsync(o.p, h_o, N * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d.p, h_d, N * 24, hipMemcpyHostToDevice, c->stream));
    LaunchCfg L = launch_cfg(c);
    launch_trace_rays(L, c->sc, n, (const double*)o.p, (const double*)d.p, (int*)tri.p, (double*)t.p, (double*)nrm.p,
                      (unsigned long long*)vh.p, (int*)cnt.p, h_bitmap ? (uint8_t*)bm.p : nullptr, stride);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(h_tri, tri.p, N * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_t, t.p, N * 8, hipMemcpyDeviceToHost, c->stream));
    if (h_normal) HIP_TRY(c, hipMemcpyAsync(h_normal, nrm.p, N * 24, hipMemcpyDeviceToHost, c->stream));
    if (h_vhash) HIP_TRY(c, hipMemcpyAsync(h_vhash, vh.p, N * 8, hipMemcpyDeviceToHost, c->stream));
    if (h_counts) HIP_TRY(c, hipMemcpyAsync(h_counts, cnt.p, N * 16, hipMemcpyDeviceToHost, c->stream));
    if (h_bitmap) HIP_TRY(c, hipMemcpyAsync(h_bitmap, bm.p, N * (size_t)stride, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!c->tri_ext.empty())
        for (size_t i = 0; i < N; ++i)
            if (h_tri[i] >= 0) h_tri[i] = c->tri_ext[(size_t)h_tri[i]];
    return CPT_OK;
}

// cpt_extend_rays and cpt_extend_rays_observed: the rays go into lane 0's first stream as a dense run and through
// one launch of the extend kernel the renderer would use; with `observe` it is the kernel's counting instance, which
// also reports what each ray's walk touched.
static int extend_rays_impl(cpt_ctx* c, const char* who, int32_t n, const double* h_o, const double* h_d, int32_t* h_tri, double* h_t,
                            bool observe, uint64_t* h_vhash, uint8_t* h_bitmap, int32_t stride) {
    if (!c || !h_o || !h_d || !h_tri || !h_t || n < 0) return c ? fail(c, CPT_EINVAL, "%s: bad argument", who) : CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "%s: no scene uploaded", who);
    if (h_bitmap && (stride < 1 || (long long)stride * 8 < c->n_tree_nodes))
        return fail(c, CPT_EINVAL, "%s: bitmap_stride %d too small for %d nodes", who, stride, c->n_tree_nodes);
    if (n == 0) return CPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    cpt_ctx::Lane& ln = c->lanes[0];
    int rc = ensure_spill(c, ln, c->stream);
    if (rc) return rc;
    rc = ensure_paths(c, ln, c->stream, cpt_stream_positions((size_t)n));
    if (rc) return rc;
    const size_t N = (size_t)n;
    DevBuf vh, bm;
    ExtendObserve obs{nullptr, nullptr, 0};
    if (observe && h_vhash) { HIP_TRY(c, vh.alloc(N * 8)); HIP_TRY(c, hipMemsetAsync(vh.p, 0, N * 8, c->stream)); obs.vhash = (unsigned long long*)vh.p; }
    if (observe && h_bitmap) {
        HIP_TRY(c, bm.alloc(N * (size_t)stride));
        HIP_TRY(c, hipMemsetAsync(bm.p, 0, N * (size_t)stride, c->stream));
        obs.bitmap = (uint8_t*)bm.p;
        obs.bitmap_stride = stride;
    }
    std::vector<double> packed(6 * N);  // the stream keeps components in pairs: (ox, oy) | (oz, dx) | (dy, dz)
    for (size_t i = 0; i < N; ++i) {
        const double v[6] = {h_o[3 * i], h_o[3 * i + 1], h_o[3 * i + 2], h_d[3 * i], h_d[3 * i + 1], h_d[3 * i + 2]};
        for (int k = 0; k < 3; ++k) { packed[(size_t)k * 2 * N + 2 * i] = v[2 * k]; packed[(size_t)k * 2 * N + 2 * i + 1] = v[2 * k + 1]; }
    }
    const RayStream& in = ln.R[0];
    cpt_v2d* dst[3] = {in.oxy, in.ozdx, in.dyz};
    for (int k = 0; k < 3; ++k) HIP_TRY(c, hipMemcpyAsync(dst[k], packed.data() + (size_t)k * 2 * N, 2 * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    uint32_t* cursors = ln.d_cursors;
    HIP_TRY(c, hipMemsetAsync(cursors, 0, CPT_PW_CURSOR_WORDS * sizeof(uint32_t), c->stream));
    const uint32_t cap = cpt_segment_cap((size_t)n);
    LaunchCfg L = launch_cfg(c);
    unsigned long long* work = (c->count_work || observe) ? c->d_work : nullptr;
    launch_dense_counts(L, ln.d_counters, (uint32_t)n, cap);
    if (c->extend_variant >= 1) launch_extend_pw(L, c->sc, in, ln.H, ln.d_counters, cap, cursors, ln.d_spill, work, obs);
    else launch_extend(L, c->sc, in, ln.H, ln.d_counters, cap, (long long)n, work, obs);
    HIP_TRY(c, hipGetLastError());
    c->stats.launches_extend++;
    std::vector<HitRec> recs(N);
    HIP_TRY(c, hipMemcpyAsync(recs.data(), ln.H.rec, N * sizeof(HitRec), hipMemcpyDeviceToHost, c->stream));
    if (obs.vhash) HIP_TRY(c, hipMemcpyAsync(h_vhash, vh.p, N * 8, hipMemcpyDeviceToHost, c->stream));
    if (obs.bitmap) HIP_TRY(c, hipMemcpyAsync(h_bitmap, bm.p, N * (size_t)stride, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < N; ++i) {
        h_t[i] = recs[i].t;
        h_tri[i] = (recs[i].tri >= 0 && !c->tri_ext.empty()) ? c->tri_ext[(size_t)recs[i].tri] : recs[i].tri;
    }
    if (c->count_work) return collect_work(c);
    if (work) HIP_TRY(c, hipMemsetAsync(c->d_work, 0, 8 * sizeof(unsigned long long), c->stream));  // counted only because the instance does
    return CPT_OK;
}

int cpt_extend_rays(cpt_ctx* c, int32_t n, const double* h_o, const double* h_d, int32_t* h_tri, double* h_t) {
    return extend_rays_impl(c, "cpt_extend_rays", n, h_o, h_d, h_tri, h_t, false, nullptr, nullptr, 0);
}

int cpt_extend_rays_observed(cpt_ctx* c, int32_t n, const double* h_o, const double* h_d, int32_t* h_tri, double* h_t,
                             uint64_t* h_vhash, uint8_t* h_bitmap, int32_t stride) {
    return extend_rays_impl(c, "cpt_extend_rays_observed", n, h_o, h_d, h_tri, h_t, true, h_vhash, h_bitmap, stride);
}

int cpt_camera_rays(cpt_ctx* c, int32_t w, int32_t h, double jx, double jy, double* h_o, double* h_d) {
    if (!c || !h_o || !h_d || w < 1 || h < 1) return c ? fail(c, CPT_EINVAL, "cpt_camera_rays: bad argument") : CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "cpt_camera_rays: no scene uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t N = (size_t)w * h;
    DevBuf o, d;
    HIP_TRY(c, o.alloc(N * 24)); HIP_TRY(c, d.alloc(N * 24));
    LaunchCfg L = launch_cfg(c);
    launch_camera_rays(L, c->sc, w, h, jx, jy, (double*)o.p, (double*)d.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(h_o, o.p, N * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_d, d.p, N * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CPT_OK;
}

int cpt_path_radiance(cpt_ctx* c, const cpt_render_params* p, int32_t n, const int32_t* h_ijs, double* h_rad) {
    if (!c || !p || !h_ijs || !h_rad || n < 0) return c ? fail(c, CPT_EINVAL, "cpt_path_radiance: bad argument") : CPT_EINVAL;
    if (!c->has_scene) return fail(c, CPT_ENOSCENE, "cpt_path_radiance: no scene uploaded");
    DParams P;
    int rc = make_params(c, p, P);
    if (rc) return rc;
    for (int k = 0; k < n; ++k)
        if (h_ijs[3 * k] < 0 || h_ijs[3 * k] >= P.w || h_ijs[3 * k + 1] < 0 || h_ijs[3 * k + 1] >= P.h || h_ijs[3 * k + 2] < 0)
            return fail(c, CPT_EINVAL, "path %d: pixel or sample out of range", k);
    if (n == 0) return CPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf ij, rad;
    size_t N = (size_t)n;
    HIP_TRY(c, ij.alloc(N * 12)); HIP_TRY(c, rad.alloc(N * 24));
    HIP_TRY(c, hipMemcpyAsync(ij.p, h_ijs, N * 12, hipMemcpyHostToDevice, c->stream));
    LaunchCfg L = launch_cfg(c);
    launch_path_radiance(L, c->sc, P, n, (const int*)ij.p, (double*)rad.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(h_rad, rad.p, N * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CPT_OK;
}

int cpt_probe_math(cpt_ctx* c, int32_t n, const double* h_a, const double* h_b, double* h_sqrt, double* h_div) {
    if (!c || !h_a || !h_sqrt || !h_div || n < 0) return c ? fail(c, CPT_EINVAL, "cpt_probe_math: bad argument") : CPT_EINVAL;
    if (n == 0) return CPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t N = (size_t)n;
    DevBuf a, b, s, q;
    HIP_TRY(c, a.alloc(N * 8)); HIP_TRY(c, b.alloc(N * 8)); HIP_TRY(c, s.alloc(N * 8)); HIP_TRY(c, q.alloc(N * 8));
    HIP_TRY(c, hipMemcpyAsync(a.p, h_a, N * 8, hipMemcpyHostToDevice, c->stream));
    if (h_b) HIP_TRY(c, hipMemcpyAsync(b.p, h_b, N * 8, hipMemcpyHostToDevice, c->stream));
    LaunchCfg L = launch_cfg(c);
    launch_probe_math(L, n, (const double*)a.p, h_b ? (const double*)b.p : nullptr, (double*)s.p, (double*)q.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(h_sqrt, s.p, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_div, q.p, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CPT_OK;
}

int cpt_probe_rng(cpt_ctx* c, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], double out_u[2]) {
    if (!c || !ctr || !key || !out || !out_u) return c ? fail(c, CPT_EINVAL, "cpt_probe_rng: bad argument") : CPT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf o, u;
    HIP_TRY(c, o.alloc(16)); HIP_TRY(c, u.alloc(16));
    LaunchCfg L = launch_cfg(c);
    launch_probe_rng(L, ctr, key, (uint32_t*)o.p, (double*)u.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, o.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_u, u.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CPT_OK;
}

}  // extern "C"
