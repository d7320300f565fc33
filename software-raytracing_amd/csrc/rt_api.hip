// rt_api.hip - host side of librt_mi355.so: the extern "C" entry points of include/rt_mi355.h.
// Flattens and precomputes the scene (what the reference recomputes per ray: rotation matrices
// of Rectangle, r*r, n.p0, the hemisphere cone cosine), uploads it, launches the kernels of
// rt_kernels.hip on a HIP stream and reads results back.  No CPU tracing path exists here:
// without a gfx950 device every render call fails with RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>
#include "rt_device.h"
#include "rt_kernels.h"
#include "rt_jit.h"
#include "rt_bvh.h"
#include "../../include/rt_mi355_debug.h"

using namespace rt;

// ---- scratch of the allocation-free entry point (rt_path_accumulate_device_noalloc) ---------------------------------------
// One fixed array per GPU, part of the code object: every GPU the library runs on holds it from module load until the process
// ends.  Layout: [scene blob | ray counters | per-slice partial sums].  In host code the NAME 'scratch' is only a stand-in for
// the symbol, not a device address: rt_create() asks hipGetSymbolAddress for the address on ITS device (each GPU has its own
// copy at its own address) and keeps it in g_scratch_ptr[device]; host code only ever uses that pointer.
#define RT_NOALLOC_SCENE_BYTES   (64u * 1024u)                 /* flattened scene incl. materials: about 700 spheres/planes */
#define RT_NOALLOC_COUNTER_BYTES (RT_COUNTER_ALLOC_WORDS * 8u)
#define RT_NOALLOC_PARTIAL_BYTES (32u * 1024u * 1024u)         /* per-slice partial images (width*height*16 B each) */
#define RT_NOALLOC_SCRATCH_BYTES (RT_NOALLOC_SCENE_BYTES + RT_NOALLOC_COUNTER_BYTES + RT_NOALLOC_PARTIAL_BYTES)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int RT_MAX_DEVICES = 64;
unsigned char* g_scratch_ptr[RT_MAX_DEVICES] = {};       // device address of 'scratch' per HIP device, filled by rt_create
hipEvent_t g_scratch_ev[RT_MAX_DEVICES] = {};            // per device, like the array it guards: recorded behind the last kernel that used this GPU's scratch, by whichever context and stream
std::mutex g_scratch_mu;

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = n + n / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostScene {
    std::vector<int4> runs, info;         // Whitted scenes: the IEEE-exact layout of rt_strict.h
    std::vector<int4> sruns;              // slot layout: runs over 'slots' (all path queries; 02's shadow samples)
    std::vector<float4> geom, mats, lights;
    bool path = false;
    bool whitted_branches = false;
    bool has_strict = false;              // path scenes: some slot is PRIM_*_STRICT (coplanar elements of different arithmetic): hit points are rounded as the reference rounds them
    bool giant_sphere = false;            // path scenes: a sphere of radius > RT_GIANT_SPHERE_RATIO x tMin (rt_device.h): scattered directions are normalised in the reference's order
    std::vector<float4> slots;            // path scenes (variant 3/4): the slot section of rt_device.h
    std::vector<size_t> slot_mat_fixups;  // indices into slots of the B records; B.w holds the material index until upload
    std::vector<int4> chunks;             // path scenes: {first slot byte offset (in the slot section), bytes, first run, n runs}
    int has_aa = 0;
    std::vector<BvhItem> items;           // one per slot, in scene order: bounds for the optional BVH (rt_bvh.h)
    Bvh bvh;
    // Appends one slot of n float4 already pushed to 'slots' to the run/chunk tables: runs never straddle a chunk, so a
    // scene too large for LDS can be walked chunk by chunk (k_path_tiled).
    // pts/n_pts: world-space corners (triangles; strict rectangles, whose slot holds the object frame instead of plane rows)
    void add_slot(int dtype, size_t n_f4, int prim, const float* pts = nullptr, int n_pts = 3) {
        const int bytes = (int)(n_f4 * 16);
        const int start = (int)(slots.size() * 16) - bytes;
        {
            BvhItem it{}; it.ref = (uint32_t)start | (uint32_t)dtype; it.bounded = true;
            const float4* S = slots.data() + slots.size() - n_f4;
            if (is_sphere_type(dtype)) {
                const float c[3] = {S[0].x, S[0].y, S[0].z}, r = sqrtf(S[0].w) * 1.000001f;
                for (int a = 0; a < 3; ++a) { it.lo[a] = c[a] - r; it.hi[a] = c[a] + r; }
            } else if (dtype == PRIM_RECT) {
                // the rows x, y, z of the slot are orthonormal, so the points with in-plane coordinates (u, v) are
                // p = x*(u - G1.w) + y*(v - G2.w) + z*G.w, |u| <= G3.x, |v| <= G3.y
                const float4 &G = S[0], &G1 = S[4], &G2 = S[5], &G3 = S[6];
                for (int a = 0; a < 3; ++a) { it.lo[a] = INFINITY; it.hi[a] = -INFINITY; }
                for (int k = 0; k < 4; ++k) {
                    const float u = ((k & 1) ? G3.x : -G3.x) - G1.w, v = ((k & 2) ? G3.y : -G3.y) - G2.w;
                    const float p[3] = {G1.x * u + G2.x * v + G.x * G.w, G1.y * u + G2.y * v + G.y * G.w, G1.z * u + G2.z * v + G.z * G.w};
                    for (int a = 0; a < 3; ++a) { it.lo[a] = std::min(it.lo[a], p[a]); it.hi[a] = std::max(it.hi[a], p[a]); }
                }
            } else if ((dtype == PRIM_TRI || dtype == PRIM_TRI_STRICT || dtype == PRIM_RECT_STRICT) && pts) {
                for (int a = 0; a < 3; ++a) { it.lo[a] = INFINITY; it.hi[a] = -INFINITY; }
                for (int k = 0; k < n_pts; ++k) for (int a = 0; a < 3; ++a) { it.lo[a] = std::min(it.lo[a], pts[3 * k + a]); it.hi[a] = std::max(it.hi[a], pts[3 * k + a]); }
            } else it.bounded = false;       // planes
            if (n_f4 == 8) {
                // wide slots carry a bounding sphere (centre, R^2) in their last float4: k_path_tiled skips a rectangle / triangle
                // whose sphere no lane of the wave can hit.  Padded like the BVH boxes (rounding of p = o + t*d and of the inclusive
                // edge tests); a primitive without finite bounds gets an infinite sphere (never skipped).
                float c[3] = {0, 0, 0}, r2 = INFINITY;
                bool ok = true;
                for (int a = 0; a < 3; ++a) ok = ok && std::isfinite(it.lo[a]) && std::isfinite(it.hi[a]);
                if (ok && it.bounded) {
                    float m = 0.0f, h2 = 0.0f;
                    for (int a = 0; a < 3; ++a) {
                        c[a] = 0.5f * (it.lo[a] + it.hi[a]);
                        const float h = 0.5f * (it.hi[a] - it.lo[a]); h2 += h * h;      // the box's half diagonal bounds every corner / vertex
                        m = std::max(m, std::max(std::fabs(it.lo[a]), std::fabs(it.hi[a])));
                    }
                    const float r = sqrtf(h2) * 1.0001f + 1e-3f + 1e-4f * m;
                    r2 = r * r;
                }
                slots[slots.size() - 1] = make_float4(c[0], c[1], c[2], r2);
            }
            for (int a = 0; a < 3; ++a) if (!std::isfinite(it.lo[a]) || !std::isfinite(it.hi[a])) it.bounded = false;
            items.push_back(it);
        }
        bool new_chunk = chunks.empty() || chunks.back().y + bytes > RT_TILE_BYTES;
        if (new_chunk) chunks.push_back(make_int4(start, 0, (int)sruns.size(), 0));
        if (!new_chunk && !sruns.empty() && sruns.back().x == dtype) sruns.back().z++;
        else { sruns.push_back(make_int4(dtype, 0, 1, prim)); chunks.back().w++; }
        chunks.back().y += bytes;
    }
};

} // namespace

struct rt_ctx {
    int device = 0;
    int n_cus = 256;                              // compute units: sizes the persistent grid of the item kernels
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_scene_use = nullptr;            // recorded behind the last work that reads scene_buf / h_stage, on whichever stream it ran
    // allocation-free entry point: this device's copy of 'scratch', a fixed pinned staging buffer for its scene blob (both
    // set up once by rt_create) and the event behind the last kernel that read them
    unsigned char* scratch_dev = nullptr; void* h_stage_na = nullptr; hipEvent_t ev_scene_use_na = nullptr;      // the event is the DEVICE's (g_scratch_ev), not owned
    bool timing_pending = false;
    DevBuf scene_buf, partial, accum, bgra, rgb;
    DevBuf wf_req, wf_words;                      // wavefront form of the Whitted kernels: the shadow requests (48 B each) and their prev / occ words + one head word per pixel
    unsigned long long* d_counters = nullptr;     // RT_COUNTER_BANKS x {rays, shadow rays}
    unsigned long long* h_counters = nullptr;     // pinned
    std::string jit_note;                         // why the last path render was not scene-specialised ("" = it was)
    int last_path_kernel = -1;                    // 3 BVH walk, 2 specialised, 1 generic LDS-resident, 0 tiled / global
    int last_path_form = -1, last_unit_log2 = 0;  // RT_FORM_* of the last path render; item form: log2 of the pixels per unit
    int last_whitted_kernel = -1;                 // 2 specialised (02 only), 1 generic
    int last_whitted_form = -1;                   // 1 wavefront form (walk / shadow rays / combine), 0 monolithic kernel
    void* h_stage = nullptr; size_t h_stage_cap = 0;   // pinned scene staging
    uint64_t pending_samples = 0;
    rt_stats stats{};
    std::string err;
    int regen_threshold_override = -1;
    int debug_flags = 0;
    // device group (rt_create_multi): this context renders on device_ids[0] and owns one child context per further entry;
    // a path render is split into sample ranges over all of them and the accumulators are summed (see render_path_group)
    std::vector<rt_ctx*> peers;
    std::vector<hipEvent_t> peer_done;            // per peer, on its device: its accumulator is complete
    hipEvent_t ar0 = nullptr, ar1 = nullptr;      // timing of the sum on this device's stream
    void* rccl_lib = nullptr; std::vector<void*> comms;   // RCCL communicators, one per device of the group (all devices distinct)
    std::string group_note;                       // how the group sums its accumulators, and why (rt_debug_group_mode)
    DevBuf gather;                                // root-gather form: a peer's accumulator copied to this device
    // progressive frame (rt_progressive_*): the running sums stay in device memory between passes; the session keeps its own
    // copy of the scene records so that the caller's may go away between passes
    struct Progressive {
        bool active = false; int W = 0, H = 0, begin = 0, end = 0, done = 0;
        rt_path_params params{}; rt_camera cam{};
        std::vector<rt_prim> prims; std::vector<rt_material> mats; std::vector<rt_light> lights;
    } prog;
    DevBuf prog_sum;
    int accel = 0;                                // RT_OPT_ACCEL: 0 brute force (the reference's algorithm), 1 BVH
    int arith = 0;                                // RT_OPT_ARITHMETIC: 0 fast, 1 the reference's operation sequence for rays and spheres
    // what scene_buf holds: a render of the same scene (same bytes, variant, BVH or not, stream) skips compile, build and upload
    bool cache_valid = false; uint64_t cache_key = 0; hipStream_t cache_stream = nullptr;
    std::vector<unsigned char> cache_bytes;       // the rt_scene records the cache was built from: a hit is confirmed by comparing them, not by the hash alone
    HostScene cache_H; DevScene cache_D{};
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// math::GetRotationMat (Common/Math.hpp:913-930): Euler degrees -> 3x3, row-major out[9].
void rotation_mat(const float deg[3], float out[9]) {
    const double k = 3.14159265358979323846 / 180.0f;
    float ax = (float)(deg[0] * k), ay = (float)(deg[1] * k), az = (float)(deg[2] * k);
    float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay), cz = cosf(az), sz = sinf(az);
    // columns i, j, k of the reference constructor; storage is row-major (Math.hpp:401-406)
    float i[3] = {cz * cy, sz * cy, -sy};
    float j[3] = {cz * sy * sx - sz * cx, sz * sy * sx + cz * cx, cy * sx};
    float kk[3] = {cz * sy * cx + sz * sx, sz * sy * cx - cz * sx, cy * cx};
    for (int r = 0; r < 3; ++r) { out[3 * r] = i[r]; out[3 * r + 1] = j[r]; out[3 * r + 2] = kk[r]; }
}

float4 f4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }

// Ray::intersectsTriangle (Common/Ray.hpp:150-212) precomputed: n = normalize(cross(e2, e1)) (:159) and rows 0, 1 of the
// inverse of the matrix with COLUMNS e2, e1, n (:190, Math.hpp:635-654).  det == 0: Inverse returns zeros, which passes
// (0,0) in the reference; 'degenerate' lets the callers keep such a triangle out instead.
struct TriBasis { float n[3], r0[3], r1[3]; bool degenerate; };
TriBasis tri_basis(const float* f) {
    TriBasis T{};
    float e1[3] = {f[3] - f[0], f[4] - f[1], f[5] - f[2]}, e2[3] = {f[6] - f[0], f[7] - f[1], f[8] - f[2]};
    float n[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
    float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (float& v : n) v = len > 0 ? v / len : 0.0f;
    float m[9] = {e2[0], e1[0], n[0], e2[1], e1[1], n[1], e2[2], e1[2], n[2]};
    float det = m[0] * m[4] * m[8] + m[1] * m[5] * m[6] + m[2] * m[3] * m[7] - m[2] * m[4] * m[6] - m[0] * m[5] * m[7] - m[1] * m[3] * m[8];
    float di = det != 0 ? 1 / det : 0.0f;
    float r0[3] = {(m[4] * m[8] - m[7] * m[5]) * di, (m[2] * m[7] - m[1] * m[8]) * di, (m[1] * m[5] - m[2] * m[4]) * di};
    float r1[3] = {(m[5] * m[6] - m[3] * m[8]) * di, (m[0] * m[8] - m[2] * m[6]) * di, (m[3] * m[2] - m[0] * m[5]) * di};
    for (int k = 0; k < 3; ++k) { T.n[k] = n[k]; T.r0[k] = r0[k]; T.r1[k] = r1[k]; }
    T.degenerate = det == 0;
    return T;
}

// ---- coplanar planar elements of different arithmetic (rt_device.h PRIM_*_STRICT) ----------------------------------------------
// Every planar element of a path scene (plane, rectangle, each of a box's six faces, triangle) as its supporting plane in
// double precision plus its outline; two of them that lie in one plane (normal within 1e-5, offset within 1e-5 of the scene's
// scale), overlap in more than a line and are NOT evaluated by identical arithmetic in the reference (same class and the same
// numbers feeding t) are flagged: their slots take the reference's own operation sequence.  flags[prim * 6 + face].
struct BoxFace { int ax; float off; int u; float ulo, uhi; int v; float vlo, vhi; float sgn; };
void box_faces(float hw, float hh, float hl, BoxFace out[6]) {
    // scene::Box::intersectsRay (04 Scene/Box.hpp:84-131): the face order of its six tests; the two Z faces take y up to +halfLength as written (:88, :95)
    const BoxFace f[6] = {{2, -hl, 0, -hw, hw, 1, -hh, hl, -1.0f}, {2, hl, 0, -hw, hw, 1, -hh, hl, 1.0f},
                          {0, -hw, 1, -hh, hh, 2, -hl, hl, -1.0f}, {0, hw, 1, -hh, hh, 2, -hl, hl, 1.0f},
                          {1, -hh, 0, -hw, hw, 2, -hl, hl, -1.0f}, {1, hh, 0, -hw, hw, 2, -hl, hl, 1.0f}};
    for (int k = 0; k < 6; ++k) out[k] = f[k];
}
// world corners of the rectangle 'coordinate ax = off, u in [ulo, uhi], v in [vlo, vhi]' of the frame with rows ri at pos
void frame_rect_corners(const float ri[9], const float pos[3], int ax, float off, int u, float ulo, float uhi, int v, float vlo, float vhi, float out[12]) {
    const float us[4] = {ulo, uhi, uhi, ulo}, vs[4] = {vlo, vlo, vhi, vhi};
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 3; ++a) out[3 * k + a] = pos[a] + off * ri[3 * ax + a] + us[k] * ri[3 * u + a] + vs[k] * ri[3 * v + a];
}
struct Planar {
    int prim, face, cls;            // cls: 0 plane, 1 rectangle / box face, 2 triangle
    double N[3], W;                 // canonical unit normal (first significant component positive), N.p = W
    double P[4][3]; int np;         // outline (np = 0: unbounded)
    float same[8]; int n_same;      // the numbers that feed t in the reference: equal class + equal numbers = identical arithmetic, a true tie there too
};
bool convex_overlap(const Planar& a, const Planar& b) {
    if (a.np == 0 || b.np == 0) return true;
    double scale = 0.0;
    for (int k = 0; k < a.np; ++k) for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(a.P[k][c] - a.P[0][c]));
    const Planar* pp[2] = {&a, &b};
    for (int w = 0; w < 2; ++w) {   // separating axes: the in-plane edge normals of both outlines
        const Planar& e = *pp[w];
        for (int k = 0; k < e.np; ++k) {
            const double* p0 = e.P[k]; const double* p1 = e.P[(k + 1) % e.np];
            const double ed[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
            double ax[3] = {ed[1] * a.N[2] - ed[2] * a.N[1], ed[2] * a.N[0] - ed[0] * a.N[2], ed[0] * a.N[1] - ed[1] * a.N[0]};
            const double l = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            if (!(l > 0.0)) continue;
            for (double& v : ax) v /= l;
            double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
            for (int q = 0; q < 2; ++q) for (int j = 0; j < pp[q]->np; ++j) {
                const double* P = pp[q]->P[j]; const double x = P[0] * ax[0] + P[1] * ax[1] + P[2] * ax[2];
                lo[q] = std::min(lo[q], x); hi[q] = std::max(hi[q], x);
            }
            if (std::min(hi[0], hi[1]) - std::max(lo[0], lo[1]) <= 1e-7 * std::max(scale, 1e-3)) return false;      // apart, or touching along a line only
        }
    }
    return true;
}
std::vector<unsigned char> find_coplanar(const rt_scene* s) {
    std::vector<unsigned char> flags((size_t)s->n_prims * 6, 0);
    std::vector<Planar> items;
    auto finish = [&](Planar& q, const double n_in[3], const double p[3]) {
        double n[3] = {n_in[0], n_in[1], n_in[2]};
        const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(l > 0.0) || !std::isfinite(l)) return;
        for (double& v : n) v /= l;
        double sgn = 1.0;
        for (int a = 0; a < 3; ++a) if (std::fabs(n[a]) > 1e-6) { sgn = n[a] < 0.0 ? -1.0 : 1.0; break; }
        for (int a = 0; a < 3; ++a) q.N[a] = sgn * n[a];
        q.W = q.N[0] * p[0] + q.N[1] * p[1] + q.N[2] * p[2];
        if (std::isfinite(q.W)) items.push_back(q);
    };
    for (int i = 0; i < s->n_prims; ++i) {
        const rt_prim& P = s->prims[i];
        const float* f = P.f;
        if (P.type == RT_PRIM_PLANE) {
            Planar q{}; q.prim = i; q.face = 0; q.cls = 0; q.np = 0; q.n_same = 6;
            for (int a = 0; a < 6; ++a) q.same[a] = f[a];
            const double n[3] = {f[3], f[4], f[5]}, p[3] = {f[0], f[1], f[2]};
            finish(q, n, p);
        } else if (P.type == RT_PRIM_TRIANGLE) {
            const TriBasis T = tri_basis(f);
            if (T.degenerate) continue;
            Planar q{}; q.prim = i; q.face = 0; q.cls = 2; q.np = 3; q.n_same = 6;
            for (int a = 0; a < 3; ++a) { q.same[a] = f[a]; q.same[3 + a] = T.n[a]; }
            for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) q.P[k][a] = f[3 * k + a];
            const double n[3] = {T.n[0], T.n[1], T.n[2]}, p[3] = {f[0], f[1], f[2]};
            finish(q, n, p);
        } else if (P.type == RT_PRIM_RECT || P.type == RT_PRIM_BOX) {
            const bool box = P.type == RT_PRIM_BOX;
            const float* orient = box ? f + 6 : f + 5;
            float neg[3] = {-orient[0], -orient[1], -orient[2]}, ri[9];
            rotation_mat(neg, ri);
            BoxFace F[6]; int nf = 1;
            if (box) { box_faces(f[3] / 2.0f, f[4] / 2.0f, f[5] / 2.0f, F); nf = 6; }
            else F[0] = BoxFace{2, 0.0f, 0, -f[3] / 2.0f, f[3] / 2.0f, 1, -f[4] / 2.0f, f[4] / 2.0f, 1.0f};
            for (int k = 0; k < nf; ++k) {
                Planar q{}; q.prim = i; q.face = k; q.cls = 1; q.np = 4; q.n_same = 8;
                for (int a = 0; a < 3; ++a) { q.same[a] = f[a]; q.same[3 + a] = orient[a]; }
                q.same[6] = (float)F[k].ax; q.same[7] = F[k].off;
                float c[12]; frame_rect_corners(ri, f, F[k].ax, F[k].off, F[k].u, F[k].ulo, F[k].uhi, F[k].v, F[k].vlo, F[k].vhi, c);
                for (int j = 0; j < 4; ++j) for (int a = 0; a < 3; ++a) q.P[j][a] = c[3 * j + a];
                const double n[3] = {ri[3 * F[k].ax], ri[3 * F[k].ax + 1], ri[3 * F[k].ax + 2]}, p[3] = {c[0], c[1], c[2]};
                finish(q, n, p);
            }
        }
    }
    if (items.size() < 2) return flags;
    std::vector<int> order(items.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return items[a].W < items[b].W; });
    long long budget = 4000000;      // pair checks: a scene of thousands of coplanar elements (a tessellated floor) is cut off, not quadratic
    for (size_t x = 0; x < order.size() && budget > 0; ++x) {
        const Planar& a = items[order[x]];
        const double tol = 1e-5 * std::max(1.0, std::fabs(a.W));
        for (size_t y = x + 1; y < order.size() && items[order[y]].W - a.W <= tol && budget > 0; ++y, --budget) {
            const Planar& b = items[order[y]];
            if (a.prim == b.prim) continue;
            if (std::fabs(a.N[0] - b.N[0]) > 1e-5 || std::fabs(a.N[1] - b.N[1]) > 1e-5 || std::fabs(a.N[2] - b.N[2]) > 1e-5) continue;
            if (a.cls == b.cls && a.n_same == b.n_same && !memcmp(a.same, b.same, sizeof(float) * (size_t)a.n_same)) continue;   // identical arithmetic: a tie in the reference as well
            if (!convex_overlap(a, b)) continue;
            flags[(size_t)a.prim * 6 + a.face] = 1; flags[(size_t)b.prim * 6 + b.face] = 1;
        }
    }
    return flags;
}

// Validates rt_scene and builds the device layout documented in rt_device.h.
// tmin: the path render's tMin (spheres of radius > RT_BIG_SPHERE_RATIO * tmin get the refined root, rt_device.h); unused for Whitted scenes
int compile_scene(rt_ctx* c, const rt_scene* s, int variant /*0 whitted, 3, 4*/, float tmin, HostScene& H) {
    if (!s || s->n_prims < 0 || s->n_materials < 0 || s->n_lights < 0) return fail(c, RT_ERR_BAD_ARG, "scene: null or negative count");
    if ((s->n_prims && !s->prims) || (s->n_materials && !s->materials) || (s->n_lights && !s->lights))
        return fail(c, RT_ERR_BAD_ARG, "scene: null array with non-zero count");
    for (int m = 0; m < s->n_materials; ++m) {
        const rt_material& M = s->materials[m];
        float4 m0 = f4(M.albedo[0], M.albedo[1], M.albedo[2], 0), m1 = f4(M.specular_intensity, M.shininess, M.primary_to_secondary, M.reflect_to_refract);
        float4 m2 = f4(M.refraction_eta, 0, M.refraction_index, 0);
        int type;
        switch (M.type) {
        case RT_MAT_PHONG: type = MAT_PHONG; break;
        case RT_MAT_DIFFUSE: type = MAT_DIFFUSE;
            m2.y = (variant == 3) ? 90.0f / 57.2958f : cosf(90.0f / 57.2958f);                         // 03 Utils.h:82, 04 Utils.h:141
            if (variant == 4) m2.x = (1.0f - m2.y) * (1.0f / 65536.0f); break;                          // the sampler's height = cone + (1 - cone)*u, u = k/65536
        case RT_MAT_METAL: type = MAT_METAL;
            if (variant == 3) m2.y = (M.roughness > 0.0f) ? (60.0f * M.roughness) / 57.2958f : -1.0f;   // 03 Materials.hpp:152-154
            else { m2.y = cosf((60.0f * M.roughness) / 57.2958f); m2.x = (1.0f - m2.y) * (1.0f / 65536.0f); } break;   // 04 Metal.hpp:76
        case RT_MAT_DIELECTRIC: { type = MAT_DIELECTRIC;
            float ratio = 1.0f / M.refraction_index;                                                    // Schlick r0, 04 Refractive.hpp:24-25
            float r0 = (1.0f - ratio) / (1.0f + ratio);
            m2.w = r0 * r0; m0.x = 1.0f / M.refraction_index; break; }
        case RT_MAT_LIGHT: type = MAT_LIGHT; break;
        default: return fail(c, RT_ERR_UNSUPPORTED, "material " + std::to_string(m) + ": unknown type " + std::to_string(M.type));
        }
        if (variant == 0 && type != MAT_PHONG) return fail(c, RT_ERR_UNSUPPORTED, "whitted render needs RT_MAT_PHONG materials (material " + std::to_string(m) + ")");
        if (variant != 0 && type == MAT_PHONG) return fail(c, RT_ERR_UNSUPPORTED, "path render cannot shade RT_MAT_PHONG (material " + std::to_string(m) + ")");
        m0.w = __builtin_bit_cast(float, type);
        H.mats.push_back(m0); H.mats.push_back(m1); H.mats.push_back(m2);
        if (type == MAT_PHONG && M.primary_to_secondary < 1.0f && M.reflect_to_refract > 0.0f && M.reflect_to_refract < 1.0f) H.whitted_branches = true;
    }
    // material slot n_materials: "no material".  Whitted: SceneElement's default (01 Types.h:51-58); path: black (04 Main.cpp:489-492)
    {
        int type = variant == 0 ? MAT_PHONG : MAT_NONE;
        H.mats.push_back(f4(1, 1, 1, __builtin_bit_cast(float, type))); H.mats.push_back(f4(0.0f, 16.0f, 1.0f, 1.0f)); H.mats.push_back(f4(1.0f, 0, 1.0f, 0));
    }
    // path scenes: planar elements coplanar with an element of other arithmetic take the reference's operation sequence
    // (rt_device.h PRIM_*_STRICT); RT_MI355_STRICT_COPLANAR=0 keeps the fast forms (A/B switch of the parity tests)
    std::vector<unsigned char> strict_flags;
    {
        const char* e = getenv("RT_MI355_STRICT_COPLANAR");
        if (variant != 0 && !(e && e[0] == '0')) strict_flags = find_coplanar(s);
        for (unsigned char f : strict_flags) H.has_strict = H.has_strict || f != 0;
    }
    auto strict_at = [&](int prim, int face) { return !strict_flags.empty() && strict_flags[(size_t)prim * 6 + face] != 0; };
    for (int i = 0; i < s->n_prims; ++i) {
        const rt_prim& P = s->prims[i];
        if (P.material >= s->n_materials) return fail(c, RT_ERR_BAD_ARG, "prim " + std::to_string(i) + ": material index out of range");
        int mat = P.material < 0 ? s->n_materials : P.material;
        int gi = (int)H.geom.size();
        int dtype = P.type;
        const float* f = P.f;
        if (variant == 0 && (P.type == RT_PRIM_RECT || P.type == RT_PRIM_BOX))
            return fail(c, RT_ERR_UNSUPPORTED, "prim " + std::to_string(i) + ": rectangles and boxes exist only in the path-tracing examples");
        H.path = variant != 0;
        {   // slot layout (rt_device.h): every query of the path scenes; the shadow samples of 02
            float4 G = f4(0, 0, 0, 0), A = f4(0, 0, 0, 0), B = f4(0, 0, 0, 0), C = f4(0, 0, 0, 0), G1 = G, G2 = G, G3 = G;
            bool wide = false;
            float strict_pts[12]; int n_strict_pts = 0;      // a strict rectangle's corners (its slot no longer holds plane rows to bound it by)
            switch (P.type) {
            case RT_PRIM_SPHERE: {
                if (variant != 0 && (fabsf(f[3]) > RT_BIG_SPHERE_RATIO * tmin || (c && c->arith == RT_ARITH_REFERENCE))) dtype = PRIM_SPHERE_BIG;      // reference arithmetic: every sphere
                if (variant != 0 && fabsf(f[3]) > RT_GIANT_SPHERE_RATIO * tmin) H.giant_sphere = true;
                G = f4(f[0], f[1], f[2], f[3] * f[3]);
                A = f4(f[0], f[1], f[2], (P.flags & RT_PRIM_FLAG_FLIP_NORMALS) ? -1.0f : 1.0f);
                break; }
            case RT_PRIM_PLANE: {
                int axis = -1;    // unit normal exactly +-e_axis: shares 1/d per ray (bit-identical to the general form)
                for (int a = 0; a < 3; ++a)
                    if (fabsf(f[3 + a]) == 1.0f && f[3 + (a + 1) % 3] == 0.0f && f[3 + (a + 2) % 3] == 0.0f) axis = a;
                if (strict_at(i, 0)) { G = f4(f[3], f[4], f[5], 0); A = f4(f[0], f[1], f[2], 0); dtype = PRIM_PLANE_STRICT; }   // n, p0 (k = 0: A.xyz is free)
                else if (axis >= 0) { G = f4(f[axis], 0, 0, 0); dtype = PRIM_AAPLANE_X + axis; H.has_aa = 1; }
                else G = f4(f[3], f[4], f[5], f[3] * f[0] + f[4] * f[1] + f[5] * f[2]);
                B = f4(f[3], f[4], f[5], 0); C = B;
                break; }
            case RT_PRIM_RECT: {
                float neg[3] = {-f[5], -f[6], -f[7]}, pos[3] = {f[5], f[6], f[7]}, ri[9], rf[9];
                rotation_mat(neg, ri); rotation_mat(pos, rf);                                           // Rectangle.hpp:58, :92
                auto dotc = [&](int row) { return ri[3 * row] * f[0] + ri[3 * row + 1] * f[1] + ri[3 * row + 2] * f[2]; };
                G = f4(ri[6], ri[7], ri[8], dotc(2));
                G1 = f4(ri[0], ri[1], ri[2], -dotc(0)); G2 = f4(ri[3], ri[4], ri[5], -dotc(1)); G3 = f4(f[3] / 2.0f, f[4] / 2.0f, 0, 0);
                B = f4(rf[2], rf[5], rf[8], 0); C = f4(ri[6], ri[7], ri[8], 0);
                wide = true;
                if (strict_at(i, 0)) {      // object frame instead of plane rows: strict_rect (rt_device.h)
                    const float hw = f[3] / 2.0f, hh = f[4] / 2.0f;                                     // Rectangle.hpp:66-67
                    G = f4(ri[0], ri[1], ri[2], 0.0f); A = f4(f[0], f[1], f[2], 0); G1 = f4(ri[3], ri[4], ri[5], -hw); G2 = f4(ri[6], ri[7], ri[8], -hh);
                    G3 = f4(hw, hh, __builtin_bit_cast(float, 2), 0);
                    frame_rect_corners(ri, f, 2, 0.0f, 0, -hw, hw, 1, -hh, hh, strict_pts); n_strict_pts = 4;
                    dtype = PRIM_RECT_STRICT;
                }
                break; }
            case RT_PRIM_TRIANGLE: {
                const TriBasis T = tri_basis(f);
                const float *n = T.n, *r0 = T.r0, *r1 = T.r1;
                G = f4(n[0], n[1], n[2], n[0] * f[0] + n[1] * f[1] + n[2] * f[2]);
                G1 = f4(r0[0], r0[1], r0[2], -(r0[0] * f[0] + r0[1] * f[1] + r0[2] * f[2]));
                G2 = f4(r1[0], r1[1], r1[2], -(r1[0] * f[0] + r1[1] * f[1] + r1[2] * f[2]));
                if (T.degenerate) G1 = f4(0, 0, 0, -1.0f);
                B = f4(n[0], n[1], n[2], 0); C = B;
                wide = true;
                if (strict_at(i, 0) && !T.degenerate) {      // strict_tri (rt_device.h)
                    G = f4(n[0], n[1], n[2], 0); A = f4(f[0], f[1], f[2], 0); G1 = f4(r0[0], r0[1], r0[2], 0); G2 = f4(r1[0], r1[1], r1[2], 0);
                    dtype = PRIM_TRI_STRICT;
                }
                break; }
            case RT_PRIM_BOX: {
                // scene::Box (04 Scene/Box.hpp:55-158) tests its six faces in a fixed order against a shrinking tMax - exactly
                // what the element list does with consecutive elements - so a box is flattened into six rectangle slots in
                // that order.  Object frame rows x,y,z = GetRotationMat(-orientation); face normals leave through
                // GetRotationMat(+orientation) (:154), the side is decided in the object frame (:148).
                float neg[3] = {-f[6], -f[7], -f[8]}, pos[3] = {f[6], f[7], f[8]}, ri[9], rf[9];
                rotation_mat(neg, ri); rotation_mat(pos, rf);
                const float hw = f[3] / 2.0f, hh = f[4] / 2.0f, hl = f[5] / 2.0f;
                auto dotc = [&](int row) { return ri[3 * row] * f[0] + ri[3 * row + 1] * f[1] + ri[3 * row + 2] * f[2]; };
                const float flip = (P.flags & RT_PRIM_FLAG_FLIP_NORMALS) ? -1.0f : 1.0f;
                // face: plane axis, its offset, the two in-plane axes with their [lo, hi] ranges, local normal sign
                BoxFace faces[6];
                box_faces(hw, hh, hl, faces);
                for (int fk = 0; fk < 6; ++fk) {
                    const BoxFace& F = faces[fk];
                    const float s = F.sgn * flip;
                    if (strict_at(i, fk)) {      // this face in the reference's operation sequence: strict_rect (rt_device.h)
                        H.slots.push_back(f4(ri[0], ri[1], ri[2], F.off));                                                // G: row 0, the face's coordinate
                        H.slots.push_back(f4(f[0], f[1], f[2], 0));                                                       // A: pos, k = 0
                        H.slot_mat_fixups.push_back(H.slots.size());
                        H.slots.push_back(f4(s * rf[F.ax], s * rf[3 + F.ax], s * rf[6 + F.ax], __builtin_bit_cast(float, mat)));
                        H.slots.push_back(f4(s * ri[3 * F.ax], s * ri[3 * F.ax + 1], s * ri[3 * F.ax + 2], 0));
                        H.slots.push_back(f4(ri[3], ri[4], ri[5], F.ulo));                                                // G1: row 1, u range from
                        H.slots.push_back(f4(ri[6], ri[7], ri[8], F.vlo));                                                // G2: row 2, v range from
                        H.slots.push_back(f4(F.uhi, F.vhi, __builtin_bit_cast(float, F.ax), 0));                          // G3
                        H.slots.push_back(f4(0, 0, 0, 0));
                        float pts[12]; frame_rect_corners(ri, f, F.ax, F.off, F.u, F.ulo, F.uhi, F.v, F.vlo, F.vhi, pts);
                        H.add_slot(PRIM_RECT_STRICT, 8, i, pts, 4);
                        continue;
                    }
                    const float umid = 0.5f * (F.ulo + F.uhi), vmid = 0.5f * (F.vlo + F.vhi);   // asymmetric ranges: |u - mid| <= half
                    H.slots.push_back(f4(ri[3 * F.ax], ri[3 * F.ax + 1], ri[3 * F.ax + 2], dotc(F.ax) + F.off));          // G
                    H.slots.push_back(f4(0, 0, 0, 0));                                                                   // A
                    H.slot_mat_fixups.push_back(H.slots.size());
                    H.slots.push_back(f4(s * rf[F.ax], s * rf[3 + F.ax], s * rf[6 + F.ax], __builtin_bit_cast(float, mat)));  // B: R(+o) column
                    H.slots.push_back(f4(s * ri[3 * F.ax], s * ri[3 * F.ax + 1], s * ri[3 * F.ax + 2], 0));              // C: R(-o) row
                    H.slots.push_back(f4(ri[3 * F.u], ri[3 * F.u + 1], ri[3 * F.u + 2], -dotc(F.u) - umid));             // G1
                    H.slots.push_back(f4(ri[3 * F.v], ri[3 * F.v + 1], ri[3 * F.v + 2], -dotc(F.v) - vmid));             // G2
                    H.slots.push_back(f4(0.5f * (F.uhi - F.ulo), 0.5f * (F.vhi - F.vlo), 0, 0));                         // G3
                    H.slots.push_back(f4(0, 0, 0, 0));
                    H.add_slot(PRIM_RECT, 8, i);
                }
                H.info.push_back(make_int4(PRIM_RECT, 0, mat, (int)P.flags));
                continue; }   // boxes: path scenes only
            default: return fail(c, RT_ERR_UNSUPPORTED, "prim " + std::to_string(i) + ": unknown type " + std::to_string(P.type));
            }
            B.w = __builtin_bit_cast(float, mat);
            H.slots.push_back(G); H.slots.push_back(A); H.slot_mat_fixups.push_back(H.slots.size()); H.slots.push_back(B); H.slots.push_back(C);
            if (wide) { H.slots.push_back(G1); H.slots.push_back(G2); H.slots.push_back(G3); H.slots.push_back(f4(0, 0, 0, 0)); }
            // rectangles and triangles share one loop in the kernel but not one run: the run type picks the inside test
            if (n_strict_pts) H.add_slot(dtype, 8, i, strict_pts, n_strict_pts);
            else H.add_slot(dtype, wide ? 8 : 4, i, P.type == RT_PRIM_TRIANGLE ? f : nullptr);
            if (variant != 0) { H.info.push_back(make_int4(dtype, 0, mat, (int)P.flags)); continue; }
            dtype = P.type;
        }
        switch (P.type) {
        // Whitted scenes (01/02): the IEEE-exact layout of rt_strict.h
        case RT_PRIM_SPHERE: H.geom.push_back(f4(f[0], f[1], f[2], f[3])); break;
        case RT_PRIM_PLANE: H.geom.push_back(f4(f[3], f[4], f[5], 0)); H.geom.push_back(f4(f[0], f[1], f[2], 0)); break;   // n, p0
        case RT_PRIM_TRIANGLE: {
            TriBasis T = tri_basis(f);
            if (T.degenerate) T.r0[0] = T.r0[1] = T.r0[2] = -1.0f;
            H.geom.push_back(f4(T.n[0], T.n[1], T.n[2], T.n[0] * f[0] + T.n[1] * f[1] + T.n[2] * f[2]));
            H.geom.push_back(f4(f[0], f[1], f[2], 0)); H.geom.push_back(f4(T.r0[0], T.r0[1], T.r0[2], 0)); H.geom.push_back(f4(T.r1[0], T.r1[1], T.r1[2], 0));
            break; }
        default: return fail(c, RT_ERR_UNSUPPORTED, "prim " + std::to_string(i) + ": unknown type " + std::to_string(P.type));
        }
        H.info.push_back(make_int4(dtype, gi, mat, (int)P.flags));
        if (!H.runs.empty() && H.runs.back().x == dtype) H.runs.back().z++;
        else H.runs.push_back(make_int4(dtype, gi, 1, i));
    }
    for (int l = 0; l < s->n_lights; ++l) {
        const rt_light& L = s->lights[l];
        H.lights.push_back(f4(L.position[0], L.position[1], L.position[2], L.radius)); H.lights.push_back(f4(L.color[0], L.color[1], L.color[2], 0));
    }
    return RT_OK;
}

// Byte offsets of the sections of the scene blob (rt_device.h) for a HostScene.
struct BlobLayout { size_t runs, sruns, chunks, geom, info, slots, mats, lights, bnodes, brefs, bunb, total; };
BlobLayout blob_layout(const HostScene& H) {
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    BlobLayout L{};
    // one spare zero record after the runs: the path kernels prefetch runs[r + 1]
    L.runs = 0; L.sruns = pad(L.runs + (H.runs.size() + 1) * 16); L.chunks = pad(L.sruns + (H.sruns.size() + 1) * 16);
    L.geom = pad(L.chunks + H.chunks.size() * 16); L.info = pad(L.geom + H.geom.size() * 16);
    L.slots = pad(L.info + (H.path ? 0 : H.info.size()) * 16); L.mats = pad(L.slots + H.slots.size() * 16);
    L.lights = pad(L.mats + H.mats.size() * 16); L.bnodes = pad(L.lights + H.lights.size() * 16 + 16); L.brefs = pad(L.bnodes + H.bvh.nodes.size() * 16);
    L.bunb = pad(L.brefs + H.bvh.refs.size() * 4); L.total = pad(L.bunb + H.bvh.unbounded.size() * 4 + 16);
    return L;
}

// Packs the HostScene into the pinned staging buffer h (>= L.total bytes), copies it to the device blob d_blob (>= L.total
// bytes) on st and fills DevScene.  'guard' is the event recorded behind the last work that read either buffer.
int fill_and_upload_scene(rt_ctx* c, const HostScene& H, const BlobLayout& L, hipStream_t st, DevScene& D, char* h, void* d_blob, hipEvent_t guard) {
    const size_t o_runs = L.runs, o_sruns = L.sruns, o_chunks = L.chunks, o_geom = L.geom, o_info = L.info, o_slots = L.slots, o_mats = L.mats,
                 o_lights = L.lights, o_bnodes = L.bnodes, o_brefs = L.brefs, o_bunb = L.bunb, total = L.total;
    if (o_slots == 0) return fail(c, RT_ERR_HIP, "internal: slot section at blob offset 0 (offset 0 means 'no hit')");
    // the staging blob and the device blob may still be in use by the previous render - on this stream or on another one the
    // caller passed then: wait for the event recorded behind that render's last kernel (valid even if its stream is gone)
    HIPCHK(c, hipEventSynchronize(guard));
    memset(h, 0, o_geom);
    memcpy(h + o_runs, H.runs.data(), H.runs.size() * 16); memcpy(h + o_sruns, H.sruns.data(), H.sruns.size() * 16);
    for (size_t k = 0; k < H.chunks.size(); ++k) {   // slot-section offsets -> blob byte offsets
        int4 ch = H.chunks[k]; ch.x += (int)o_slots;
        memcpy(h + o_chunks + k * 16, &ch, 16);
    }
    memcpy(h + o_geom, H.geom.data(), H.geom.size() * 16);
    if (!H.path) memcpy(h + o_info, H.info.data(), H.info.size() * 16);
    memcpy(h + o_slots, H.slots.data(), H.slots.size() * 16);
    for (size_t b : H.slot_mat_fixups) {   // material index -> byte offset of its record in the blob
        float4* B = reinterpret_cast<float4*>(h + o_slots) + b;
        const int mi = __builtin_bit_cast(int, B->w);
        uint32_t off = (uint32_t)(o_mats + (size_t)mi * 48);
        B->w = __builtin_bit_cast(float, off);
        B[1].w = H.mats[(size_t)mi * 3].w;      // C.w: the material's class (the sorted form classifies a hit by one dword)
    }
    memcpy(h + o_mats, H.mats.data(), H.mats.size() * 16);
    memcpy(h + o_lights, H.lights.data(), H.lights.size() * 16);
    {   // slot-section offsets -> blob byte offsets (the low bits carry the device type)
        memcpy(h + o_bnodes, H.bvh.nodes.data(), H.bvh.nodes.size() * 16);
        uint32_t* r = reinterpret_cast<uint32_t*>(h + o_brefs);
        for (size_t k = 0; k < H.bvh.refs.size(); ++k) r[k] = H.bvh.refs[k] + (uint32_t)o_slots;
        r = reinterpret_cast<uint32_t*>(h + o_bunb);
        for (size_t k = 0; k < H.bvh.unbounded.size(); ++k) r[k] = H.bvh.unbounded[k] + (uint32_t)o_slots;
    }
    HIPCHK(c, hipMemcpyAsync(d_blob, h, total, hipMemcpyHostToDevice, st));
    D.blob = (const float4*)d_blob;
    D.off_runs = (int)(o_runs / 16); D.off_geom = (int)(o_geom / 16); D.off_info = (int)(o_info / 16);
    D.off_mats = (int)(o_mats / 16); D.off_lights = (int)(o_lights / 16); D.total_f4 = (int)(total / 16);
    D.n_runs = (int)H.runs.size(); D.n_prims = (int)H.info.size(); D.n_mats = (int)H.mats.size() / 3; D.n_lights = (int)H.lights.size() / 2;
    D.has_aa = H.has_aa; D.off_slots = (int)(o_slots / 16);
    D.off_chunks = (int)(o_chunks / 16); D.n_chunks = (int)H.chunks.size();
    D.whitted_branches = H.whitted_branches;
    D.srun_pack[0] = D.srun_pack[1] = 0ULL; D.srun_packed = H.sruns.size() <= 8;
    for (size_t r = 0; r < H.sruns.size() && r < 8; ++r) {
        if (H.sruns[r].z > 8191 || H.sruns[r].x > 7) D.srun_packed = 0;      // 3 bits of type, 13 of count
        D.srun_pack[r / 4] |= (unsigned long long)(((uint32_t)H.sruns[r].x | ((uint32_t)H.sruns[r].z << 3)) & 0xFFFFU) << ((r % 4) * 16);
    }
    D.off_sruns = (int)(o_sruns / 16); D.n_sruns = (int)H.sruns.size(); D.slot_bytes = (int)(H.slots.size() * 16);
    D.off_bvh_nodes = (int)(o_bnodes / 16); D.off_bvh_refs = (int)(o_brefs / 16); D.off_bvh_unbounded = (int)(o_bunb / 16);
    D.n_bvh_refs = (int)H.bvh.refs.size(); D.n_bvh_unbounded = (int)H.bvh.unbounded.size(); D.bvh_depth = H.bvh.depth; D.n_bvh_nodes = (int)H.bvh.nodes.size() / 4;
    return RT_OK;
}

// The allocating form: grows the context's pinned staging buffer and device blob as needed.
int upload_scene(rt_ctx* c, const HostScene& H, hipStream_t st, DevScene& D) {
    const BlobLayout L = blob_layout(H);
    if (L.total > c->h_stage_cap) {
        HIPCHK(c, hipEventSynchronize(c->ev_scene_use));
        if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
        c->h_stage = nullptr; c->h_stage_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_stage, L.total * 2, hipHostMallocDefault));
        c->h_stage_cap = L.total * 2;
    }
    HIPCHK(c, c->scene_buf.reserve(L.total));
    return fill_and_upload_scene(c, H, L, st, D, (char*)c->h_stage, c->scene_buf.p, c->ev_scene_use);
}

// compile_scene + (optional) build_bvh + upload_scene behind a one-entry cache keyed by the scene's bytes: progressive and
// repeated renders of one scene (the BVH build of the stress scene alone is 20 ms) pay for it once.
uint64_t scene_key(const rt_scene* s, int variant, int accel, float tmin) {
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p; size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 1099511628211ULL; }
        for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ULL;
    };
    int32_t tmin_bits; memcpy(&tmin_bits, &tmin, 4);
    const int32_t head[6] = {variant, accel, s->n_prims, s->n_materials, s->n_lights, tmin_bits};
    mix(head, sizeof head);
    if (s->n_prims > 0 && s->prims) mix(s->prims, (size_t)s->n_prims * sizeof(rt_prim));
    if (s->n_materials > 0 && s->materials) mix(s->materials, (size_t)s->n_materials * sizeof(rt_material));
    if (s->n_lights > 0 && s->lights) mix(s->lights, (size_t)s->n_lights * sizeof(rt_light));
    return h;
}
int prepare_scene(rt_ctx* c, const rt_scene* scene, int variant, float tmin, bool accel, hipStream_t st, const HostScene*& H, DevScene& D) {
    if (!scene || scene->n_prims < 0 || scene->n_materials < 0 || scene->n_lights < 0 || (scene->n_prims && !scene->prims) ||
        (scene->n_materials && !scene->materials) || (scene->n_lights && !scene->lights)) {
        HostScene tmp; return compile_scene(c, scene, variant, tmin, tmp);      // reports the error
    }
    const uint64_t key = scene_key(scene, variant | (c->arith << 8), accel ? 1 : 0, tmin);
    const size_t nb_p = (size_t)scene->n_prims * sizeof(rt_prim), nb_m = (size_t)scene->n_materials * sizeof(rt_material), nb_l = (size_t)scene->n_lights * sizeof(rt_light);
    bool hit = c->cache_valid && c->cache_key == key && c->cache_stream == st && c->cache_bytes.size() == nb_p + nb_m + nb_l;
    if (hit) {   // a 64-bit hash can collide: the cached blob is only reused for the very same records
        const unsigned char* b = c->cache_bytes.data();
        hit = (!nb_p || !memcmp(b, scene->prims, nb_p)) && (!nb_m || !memcmp(b + nb_p, scene->materials, nb_m)) && (!nb_l || !memcmp(b + nb_p + nb_m, scene->lights, nb_l));
    }
    if (!hit) {
        c->cache_valid = false;
        c->cache_H = HostScene();
        int rc = compile_scene(c, scene, variant, tmin, c->cache_H);
        if (rc) return rc;
        if (accel) {
            build_bvh(c->cache_H.items, c->cache_H.bvh);
            if (c->cache_H.bvh.depth > RT_BVH_STACK)
                return fail(c, RT_ERR_TOO_LARGE, "BVH deeper than the traversal stack (" + std::to_string(c->cache_H.bvh.depth) + " > " + std::to_string(RT_BVH_STACK) + ")");
        }
        if ((rc = upload_scene(c, c->cache_H, st, c->cache_D))) return rc;
        c->cache_bytes.resize(nb_p + nb_m + nb_l);
        if (nb_p) memcpy(c->cache_bytes.data(), scene->prims, nb_p);
        if (nb_m) memcpy(c->cache_bytes.data() + nb_p, scene->materials, nb_m);
        if (nb_l) memcpy(c->cache_bytes.data() + nb_p + nb_m, scene->lights, nb_l);
        c->cache_key = key; c->cache_stream = st; c->cache_valid = true;
    }
    H = &c->cache_H; D = c->cache_D;
    return RT_OK;
}

int begin_timed(rt_ctx* c, hipStream_t st, unsigned long long* d_counters = nullptr) {
    HIPCHK(c, hipMemsetAsync(d_counters ? d_counters : c->d_counters, 0, RT_COUNTER_ALLOC_WORDS * sizeof(unsigned long long), st));      // ray counters + the item kernels' work counter
    HIPCHK(c, hipEventRecord(c->ev0, st));
    return RT_OK;
}
int end_timed(rt_ctx* c, hipStream_t st, unsigned long long* d_counters = nullptr, hipEvent_t ev_use = nullptr) {
    HIPCHK(c, hipEventRecord(c->ev1, st));
    HIPCHK(c, hipEventRecord(ev_use ? ev_use : c->ev_scene_use, st));     // the scene blob is free for rewriting once this has passed
    HIPCHK(c, hipMemcpyAsync(c->h_counters, d_counters ? d_counters : c->d_counters, RT_COUNTER_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    c->timing_pending = true;
    return RT_OK;
}

int check_ctx(rt_ctx* c) {
    if (!c) return RT_ERR_BAD_ARG;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return RT_OK;
}

// Legacy forms (path_body, tiled, BVH walk: one lane per pixel): sample slices when the image alone cannot fill the chip.
int pick_slices(int width, int height, int n_samples) {
    // enough threads to fill 256 CUs x 2048 lanes several times over; one slice at 4K
    long long pixels = (long long)width * height;
    long long target = 4LL * 256 * 2048;
    int s = (int)((target + pixels - 1) / pixels);
    if (s < 1) s = 1;
    if (s > n_samples) s = n_samples;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}
// Item form (path_items_body): a unit of work is 2^unit_log2 consecutive pixels x the sample range.  Long frames are cut into
// units by shrinking the PIXEL group, not the sample range - about RT_UNIT_ITEMS paths per unit whatever the frame (16 pixels
// at 256 ... 1024 spp, 64 at <= 64) - so every pixel keeps one owner and no partial image exists.  The unit size is a
// measured compromise (tools/sweep_unit.py, profiles/r03_unit_size_sweep.txt): a wave drains a unit before it takes the next (its
// last paths run in a partly filled wave: short units cost 03 +5 % at 2048 paths, nothing on 04, whose paths end together), few
// pixels per unit put many lanes on one pixel's sums (same-address LDS atomics: 03 x2 at 2 pixels), and long units make the
// launch's tail long and its load uneven (03 at 1080p/256 spp: 6.0 ms at 16384 paths per unit, 5.4 at 4096).  There must also
// be enough units to occupy the chip; frames too small for that (784x561 at 32 spp is 0.5 ms of work) keep one lane per pixel
// and sample slices (path_body), which fill the chip with fewer paths per wave.
constexpr int RT_UNIT_ITEMS = 4096, RT_UNIT_MIN_ITEMS = 2048;
struct ItemPlan { bool ok; int unit_log2, n_groups; };
ItemPlan plan_items(int width, int height, int n_samples, int n_cus) {
    const long long pixels = (long long)width * height, waves = (long long)n_cus * 32;      // resident waves at 8 blocks per CU
    int k = 6;
    while (k > 0 && ((long long)n_samples << k) > RT_UNIT_ITEMS) --k;
    // a unit's pixels are stored as one run of 16 B x 2^k: two 128-byte lines when that keeps the unit within 4 x RT_UNIT_ITEMS (measured HBM writes per 4K frame
    // of 132.7 MB at 1024 spp: 4 pixels per unit 199.5 MB - partial lines -, 8 pixels 166.3 MB, 16 pixels 149.8 MB; same speed, 04's paths end together)
    while (k < 4 && ((long long)n_samples << (k + 1)) <= 4 * RT_UNIT_ITEMS) ++k;
    while (k > 0 && (pixels >> k) < 6 * waves && ((long long)n_samples << (k - 1)) >= RT_UNIT_MIN_ITEMS) --k;      // small frames: more, shorter units (784x561 at 256 spp: 3.5 -> 3.3 ms)
    ItemPlan p{true, k, (int)((pixels + (1 << k) - 1) >> k)};
    p.ok = ((long long)n_samples << k) >= RT_UNIT_MIN_ITEMS && p.n_groups >= waves;
    return p;
}

// Outcome-sorted form (path_sorted_body): the unit belongs to a workgroup of 256 lanes: up to 128 pixels, at least 64 paths per lane.
ItemPlan plan_sorted(int width, int height, int n_samples, int n_cus) {
    const long long pixels = (long long)width * height, blocks = (long long)n_cus * 5;
    int k = 7;
    while (k > 0 && ((long long)n_samples << k) > 4 * RT_UNIT_ITEMS) --k;
    while (k > 0 && (pixels >> k) < 2 * blocks && ((long long)n_samples << (k - 1)) >= 4 * RT_UNIT_MIN_ITEMS) --k;
    ItemPlan p{true, k, (int)((pixels + (1 << k) - 1) >> k)};
    p.ok = ((long long)n_samples << k) >= 4 * RT_UNIT_MIN_ITEMS && p.n_groups >= blocks;
    return p;
}

// ---- device group: RCCL bound at run time ------------------------------------------------------------------------------
// librccl is loaded with dlopen when a group of DISTINCT devices is created, not linked: a process that also runs
// torch.distributed already holds PyTorch's own copy of RCCL, and a single-GPU user needs none.  Signatures of rccl.h.
struct Rccl {
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;                              // g_rccl is process-wide: bound under this lock by every group that loads the library
constexpr int kNcclFloat32 = 7, kNcclSum = 0;      // rccl.h: ncclFloat32 = 7, ncclSum = 0
bool rccl_bind(void* lib) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    return g_rccl.CommInitAll && g_rccl.AllReduce && g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.CommDestroy && g_rccl.GetErrorString;
}

// Sample range [s0, s1) cut into n contiguous shares; share i (may be empty).  The reference's own split is by pixel ranges
// over 8 threads (04 Main.cpp:399-420); sample ranges balance whatever the scene.
void share_of(int s0, int s1, int i, int n, int& b, int& e) {
    const long long len = s1 - s0;
    b = s0 + (int)(len * i / n); e = s0 + (int)(len * (i + 1) / n);
}

} // namespace

extern "C" {

int32_t rt_abi_version(void) { return RT_ABI_VERSION; }

rt_ctx* rt_create(int32_t device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0"); return nullptr; }
    int dev = device_id;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= n) { g_create_error = "device id out of range"; return nullptr; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) { g_create_error = hipGetErrorString(e); return nullptr; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) { g_create_error = std::string("device is ") + prop.gcnArchName + ", library is built for gfx950 only"; return nullptr; }
    if ((e = hipSetDevice(dev)) != hipSuccess) { g_create_error = hipGetErrorString(e); return nullptr; }
    rt_ctx* c = new rt_ctx();
    c->device = dev;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&c->ev0) == hipSuccess &&
              hipEventCreate(&c->ev1) == hipSuccess && hipEventCreateWithFlags(&c->ev_scene_use, hipEventDisableTiming) == hipSuccess && hipMalloc((void**)&c->d_counters, RT_COUNTER_ALLOC_WORDS * sizeof(unsigned long long)) == hipSuccess &&
              hipHostMalloc((void**)&c->h_counters, RT_COUNTER_WORDS * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess;
    if (!ok) { g_create_error = "failed to create stream/events/counters"; rt_destroy(c); return nullptr; }
    {   // set-up of the allocation-free entry point, once per device, with that device current
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (dev < RT_MAX_DEVICES && !g_scratch_ptr[dev]) {
            unsigned char* ptr = nullptr;
            hipError_t se = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
            if (se != hipSuccess || !ptr) { g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(se); rt_destroy(c); return nullptr; }
            hipEvent_t gev = nullptr;
            if (hipEventCreateWithFlags(&gev, hipEventDisableTiming) != hipSuccess) { g_create_error = "hipEventCreate (scratch guard) failed"; rt_destroy(c); return nullptr; }
            g_scratch_ptr[dev] = ptr; g_scratch_ev[dev] = gev;      // lives until the process ends, like the array
        }
        c->scratch_dev = dev < RT_MAX_DEVICES ? g_scratch_ptr[dev] : nullptr;
        c->ev_scene_use_na = dev < RT_MAX_DEVICES ? g_scratch_ev[dev] : nullptr;
    }
    if (hipHostMalloc(&c->h_stage_na, RT_NOALLOC_SCENE_BYTES, hipHostMallocDefault) != hipSuccess) { g_create_error = "failed to create the allocation-free entry point's staging buffer"; rt_destroy(c); return nullptr; }
    memset(c->h_counters, 0, RT_COUNTER_WORDS * sizeof(unsigned long long));
    return c;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    for (void* comm : c->comms) if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
    c->comms.clear();
    if (c->rccl_lib) { (void)dlclose(c->rccl_lib); c->rccl_lib = nullptr; }      // dlopen counts references: the library stays while another group holds it
    for (size_t i = 0; i < c->peers.size(); ++i) {
        if (c->peer_done[i]) { (void)hipSetDevice(c->peers[i]->device); (void)hipEventDestroy(c->peer_done[i]); }
        rt_destroy(c->peers[i]);
    }
    c->peers.clear(); c->peer_done.clear();
    (void)hipSetDevice(c->device);
    if (c->ar0) (void)hipEventDestroy(c->ar0);
    if (c->ar1) (void)hipEventDestroy(c->ar1);
    c->gather.release();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->wf_req.release(); c->wf_words.release(); c->scene_buf.release(); c->partial.release(); c->accum.release(); c->bgra.release(); c->rgb.release(); c->prog_sum.release();
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_scene_use) (void)hipEventDestroy(c->ev_scene_use);
    if (c->h_stage_na) (void)hipHostFree(c->h_stage_na);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

rt_ctx* rt_create_multi(const int32_t* device_ids, int32_t n_devices) {
    if (!device_ids || n_devices <= 0 || n_devices > RT_MAX_DEVICES) { g_create_error = "rt_create_multi: null list or device count outside [1, 64]"; return nullptr; }
    rt_ctx* c = rt_create(device_ids[0]);
    if (!c) return nullptr;
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    for (int i = 1; i < n_devices; ++i) {
        rt_ctx* p = rt_create(device_ids[i]);
        if (!p) { rt_destroy(c); return nullptr; }      // g_create_error says which device and why
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { g_create_error = "rt_create_multi: hipEventCreate failed"; rt_destroy(p); rt_destroy(c); return nullptr; }
        c->peers.push_back(p); c->peer_done.push_back(ev);
    }
    (void)hipSetDevice(c->device);
    if (hipEventCreate(&c->ar0) != hipSuccess || hipEventCreate(&c->ar1) != hipSuccess) { g_create_error = "rt_create_multi: hipEventCreate failed"; rt_destroy(c); return nullptr; }
    // Distinct devices: one RCCL communicator per device, a single process driving all of them (ncclCommInitAll); the per-GPU
    // accumulators are then combined with ONE all-reduce over xGMI.  The same device listed twice (a one-GPU rehearsal of the
    // group code) cannot form a communicator: the sums then go through the root-gather form (copy to device_ids[0], add).
    // RCCL that cannot be loaded or initialised is not an error: the root-gather form needs none of it.
    c->group_note = n_devices == 1 ? "one device: nothing to sum" : !distinct ? "a device is listed twice: root-gather form (copies to the first device, added in a fixed order)"
                  : getenv("RT_MI355_NO_RCCL") ? "RT_MI355_NO_RCCL is set: root-gather form" : "";
    if (distinct && n_devices > 1 && !(getenv("RT_MI355_NO_RCCL"))) {
        const char* libname = getenv("RT_MI355_RCCL_LIB");      // test hook: another library name (a name that cannot be loaded exercises the fallback)
        c->rccl_lib = dlopen(libname ? libname : "librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!c->rccl_lib && !libname) c->rccl_lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!c->rccl_lib || !rccl_bind(c->rccl_lib)) {
            const char* why = dlerror();                 // one call: dlerror() clears the message it returns
            c->group_note = std::string("RCCL not usable (") + (why ? why : "missing symbols") + "): root-gather form";
            if (c->rccl_lib) { (void)dlclose(c->rccl_lib); c->rccl_lib = nullptr; }
        } else {
            c->comms.assign((size_t)n_devices, nullptr);
            std::vector<int> devs(device_ids, device_ids + n_devices);
            int r = g_rccl.CommInitAll(c->comms.data(), n_devices, devs.data());
            if (r != 0) {
                c->group_note = std::string("ncclCommInitAll failed (") + g_rccl.GetErrorString(r) + "): root-gather form";
                c->comms.clear();
            } else c->group_note = "one grouped ncclAllReduce(sum, float32) over a single-process communicator";
            (void)hipSetDevice(c->device);
        }
    }
    return c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int32_t rt_get_stats(const rt_ctx* c, rt_stats* out) {
    if (!c || !out) return RT_ERR_BAD_ARG;
    *out = c->stats;
    return RT_OK;
}

int32_t rt_whitted_params_default(int32_t example, rt_whitted_params* out) {
    if (!out || (example != 1 && example != 2)) return RT_ERR_BAD_ARG;
    out->max_depth = 4; out->shadow_samples = example == 2 ? 16 : 0; out->seed = 1; out->reserved = 0;
    return RT_OK;
}
int32_t rt_path_params_default(int32_t example, rt_path_params* out) {
    if (!out || (example != 3 && example != 4)) return RT_ERR_BAD_ARG;
    out->variant = example; out->spp = example == 3 ? 16 : 32; out->max_depth = example == 3 ? 24 : 6; out->seed = 1;
    out->sample_begin = 0; out->sample_end = 0; out->t_min = example == 3 ? 0.001f : 0.01f; out->t_max = 1000.0f;
    return RT_OK;
}

int32_t rt_set_option(rt_ctx* c, int32_t option, int32_t value) {
    if (!c) return RT_ERR_BAD_ARG;
    if (option == RT_OPT_ACCEL) {
        if (value != RT_ACCEL_NONE && value != RT_ACCEL_BVH) return fail(c, RT_ERR_BAD_ARG, "RT_OPT_ACCEL: value must be RT_ACCEL_NONE or RT_ACCEL_BVH");
        c->accel = value;
        return RT_OK;
    }
    if (option == RT_OPT_ARITHMETIC) {
        if (value != RT_ARITH_FAST && value != RT_ARITH_REFERENCE) return fail(c, RT_ERR_BAD_ARG, "RT_OPT_ARITHMETIC: value must be RT_ARITH_FAST or RT_ARITH_REFERENCE");
        c->arith = value;
        for (rt_ctx* d : c->peers) if (d) d->arith = value;
        return RT_OK;
    }
    return fail(c, RT_ERR_BAD_ARG, "rt_set_option: unknown option " + std::to_string(option));
}

int32_t rt_sync(rt_ctx* c, void* stream) {
    int rc = check_ctx(c); if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->timing_pending) {
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.kernel_ms = ms;
        c->stats.rays = c->stats.shadow_rays = 0;   // the kernels spread their atomics over RT_COUNTER_BANKS banks
        for (int b = 0; b < RT_COUNTER_BANKS; ++b) { c->stats.rays += c->h_counters[2 * b]; c->stats.shadow_rays += c->h_counters[2 * b + 1]; }
        c->stats.samples = c->pending_samples;
        c->timing_pending = false;
    }
    return RT_OK;
}

// noalloc: the body of rt_path_accumulate_device_noalloc - scene blob, ray counters and per-slice partial sums live in this
// device's 'scratch' array, the generic (precompiled) kernels run, and no HIP allocation or free happens on any path below.
static int32_t path_accumulate(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_path_params* p,
                               int32_t width, int32_t height, float* d_accum, void* stream, bool noalloc) {
    int rc = check_ctx(c); if (rc) return rc;
    if (!cam || !p || !d_accum || width <= 0 || height <= 0) return fail(c, RT_ERR_BAD_ARG, "rt_path_accumulate_device: null pointer or non-positive size");
    if ((long long)width * height > (1LL << 28)) return fail(c, RT_ERR_TOO_LARGE, "image larger than 2^28 pixels");
    if (p->variant != RT_PATH_SKY && p->variant != RT_PATH_LIGHTS) return fail(c, RT_ERR_BAD_ARG, "variant must be 3 or 4");
    int s0 = p->sample_begin, s1 = p->sample_end;
    if (s0 == 0 && s1 == 0) s1 = p->spp;
    if (p->spp <= 0 || s0 < 0 || s1 < s0 || p->max_depth < 0) return fail(c, RT_ERR_BAD_ARG, "bad spp / sample range / max_depth");
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const bool accel = c->accel == 1, ref = c->arith == RT_ARITH_REFERENCE;
    if (ref && accel) return fail(c, RT_ERR_UNSUPPORTED, "RT_ARITH_REFERENCE renders by brute force only (RT_OPT_ACCEL must be RT_ACCEL_NONE)");
    const HostScene* Hp = nullptr; DevScene D;
    HostScene na_H;
    unsigned long long* counters = c->d_counters;
    if (noalloc) {
        if (accel) return fail(c, RT_ERR_UNSUPPORTED, "rt_path_accumulate_device_noalloc: brute force only (RT_OPT_ACCEL must be RT_ACCEL_NONE)");
        if (!c->scratch_dev) return fail(c, RT_ERR_HIP, "rt_path_accumulate_device_noalloc: no scratch array on this device");
        if ((rc = compile_scene(c, scene, p->variant, p->t_min, na_H))) return rc;
        const BlobLayout L = blob_layout(na_H);
        if (L.total > RT_NOALLOC_SCENE_BYTES)
            return fail(c, RT_ERR_TOO_LARGE, "rt_path_accumulate_device_noalloc: flattened scene is " + std::to_string(L.total) + " bytes, the fixed scratch holds " +
                        std::to_string(RT_NOALLOC_SCENE_BYTES) + " (use rt_path_accumulate_device)");
        if ((rc = fill_and_upload_scene(c, na_H, L, st, D, (char*)c->h_stage_na, c->scratch_dev, c->ev_scene_use_na))) return rc;
        Hp = &na_H;
        counters = reinterpret_cast<unsigned long long*>(c->scratch_dev + RT_NOALLOC_SCENE_BYTES);
    } else if ((rc = prepare_scene(c, scene, p->variant, p->t_min, accel, st, Hp, D))) return rc;
    const HostScene& H = *Hp;

    PathArgs A{};
    A.width = width; A.height = height; A.sample_begin = s0; A.sample_end = s1; A.max_depth = p->max_depth;
    A.jitter = p->spp > 1; A.seed = p->seed; A.t_min = p->t_min; A.t_min_pos = p->t_min > 0.0f ? p->t_min : 1.401298464e-45f; A.t_max = p->t_max;
    A.debug_flags = c->debug_flags; A.ref_rays = ref ? 1 : 0; A.norm_dirs = H.giant_sphere ? 1 : 0; A.strict_point = H.has_strict ? 1 : 0;
    {   // Refract (Math.hpp:306-312) assumes |d| = 1 and multiplies what |d|^2 is off by with eta^2: harmless through a glass body (in: eta^2, out: 1/eta^2), but a path that refracts
        // the SAME way many times grows it exponentially (2.8^n at eta = 1/0.6) - and the fast sphere test takes a = d.d as 1.  That happens at the virtual surfaces of the tMax rule
        // inside a glass sphere wider than tMax (a hit AT tMax, again and again: Common/Ray.hpp:127-134), and through a stack of glass sheets.  The reference normalises every ray;
        // scenes where this can accumulate do too (fuzz family A at tMax = 12, a glass ground: +2.4 % scene queries without, -1e-6 with; profiles/r03_path_fuzz.txt 14).
        int sheets = 0;
        for (int i = 0; i < scene->n_prims && !A.norm_dirs; ++i) {
            const rt_prim& P = scene->prims[i];
            if (P.material < 0 || P.material >= scene->n_materials || scene->materials[P.material].type != RT_MAT_DIELECTRIC) continue;
            if (P.type == RT_PRIM_SPHERE) { if (2.0f * fabsf(P.f[3]) > p->t_max) A.norm_dirs = 1; }
            else if (P.type != RT_PRIM_BOX && ++sheets >= 4) A.norm_dirs = 1;
        }
    }
    // camera code costs about as much as testing 20 primitives: with few primitives wait for a batch of idle lanes,
    // with many refill every idle lane at once (measured: tools/sweep_regen.py, tools/sweep_regen_stress.py)
    A.regen_threshold = c->regen_threshold_override >= 0 ? c->regen_threshold_override : ((accel || scene->n_prims >= 128) ? 1 : 16);
    float fov_rad = (float)(cam->fov_deg / (180.0f / 3.14159265358979323846));       // 03 Main.cpp:317
    float th = tanf(fov_rad / 2.0f), w = (float)width, h = (float)height;
    A.inv_w2 = 2.0f / w; A.inv_h2 = 2.0f / h; A.kx = th * w / h; A.ky = th;
    rotation_mat(cam->orientation_deg, A.r);
    A.px = cam->position[0]; A.py = cam->position[1]; A.pz = cam->position[2];
    int n_samples = s1 - s0;
    size_t npx = (size_t)width * height;
    // Which form of the bounce loop.  LDS-resident scenes (the specialised builds included) take the ITEM form (path_items_body:
    // persistent waves, a work counter, lanes decoupled from pixels, fixed-point per-pixel sums) unless something it
    // cannot do is asked for: material colours outside [0, 1] (lights: [0, 1e6]) - its sums are unsigned 32.32 fixed point.
    // RT_MI355_PATH=r or debug flag 512 keeps per-lane regeneration (path_body), RT_MI355_PATH=q or debug flag 256 selects the
    // queue form (path_queue_body: workgroup-level ray-queue compaction through LDS; it lost its A/B).  Scenes
    // beyond LDS (k_path_tiled) and the BVH walk keep one lane per pixel.
    const char* pk = getenv("RT_MI355_PATH");
    bool colours_ok = true;
    for (int m = 0; colours_ok && m < scene->n_materials; ++m)
        for (int k = 0; k < 3; ++k) {
            const float v = scene->materials[m].albedo[k];
            colours_ok = colours_ok && v >= 0.0f && v <= (scene->materials[m].type == RT_MAT_LIGHT ? 1e6f : 1.0f);
        }
    const bool lds_fit = (size_t)D.total_f4 * 16 <= RT_LDS_SCENE_MAX_BYTES;
    A.queue = (((pk && pk[0] == 'q') || (c->debug_flags & 256)) && !noalloc && !accel && !ref && !(c->debug_flags & (2 | 4)) && lds_fit && colours_ok) ? 1 : 0;
    A.items = (!A.queue && !accel && !ref && lds_fit && colours_ok && !(pk && pk[0] == 'r') && !(c->debug_flags & 512) && n_samples > 0) ? 1 : 0;
    // The outcome-sorted form (path_sorted_body: the workgroup sorts its paths by what the query found, every wave runs one kind of
    // work): RT_MI355_PATH=s or debug flag 2048; same eligibility as the item form, blob + exchange buffers within 64 KB of LDS.
    A.sorted = 0;
    if (A.items && ((pk && pk[0] == 's') || (c->debug_flags & 2048)) && !noalloc && !(c->debug_flags & (2 | 4)) && p->max_depth <= 254 &&
        (size_t)D.total_f4 * 16 + RT_SORT_WS_BYTES <= 64 * 1024) {
        ItemPlan sp = plan_sorted(width, height, n_samples, c->n_cus);
        if (c->debug_flags & 1024) sp.ok = true;
        if (sp.ok) { A.sorted = 1; A.items = 0; A.unit_log2 = sp.unit_log2; A.n_groups = sp.n_groups; }
    }
    int slices = 1;
    if (A.items) {
        ItemPlan ip = plan_items(width, height, n_samples, c->n_cus);
        if (const char* e = getenv("RT_MI355_UNIT_LOG2")) {      // experiment switch: item form with 2^v pixels per unit, whatever the frame
            const int v = atoi(e);
            if (v >= 0 && v <= 6) { ip.ok = true; ip.unit_log2 = v; ip.n_groups = (int)((npx + ((size_t)1 << v) - 1) >> v); }
        }
        if (c->debug_flags & 1024) ip.ok = true;                 // tests: item form on small frames too
        A.items = ip.ok ? 1 : 0; A.unit_log2 = ip.unit_log2; A.n_groups = ip.n_groups;
        // Units are handed out BOTTOM-UP (last pixel group first).  The launch ends when the last unit does, so cheap units should
        // come last; what a unit costs is not known beforehand, but a camera that looks at the horizon has the sky on top (one ray
        // per sample) and the ground below (03: 3 rays per sample), and a closed room does not care.  Measured (tools/sweep_order.py,
        // profiles/r03_unit_size_sweep.txt): 03 1080p/256 spp 5.57 -> 5.37 ms, 04 within +-0.5 %; a strided order was no better.
        // RT_MI355_UNIT_ORDER=0: row-major (experiment switch).
        A.unit_order = 1;
        if (const char* e = getenv("RT_MI355_UNIT_ORDER")) A.unit_order = atoi(e) == 0 ? 0 : 1;
    }
    if (!A.items && !A.sorted && n_samples > 0) slices = pick_slices(width, height, n_samples);
    if (const char* e = getenv("RT_MI355_SLICES")) {      // experiment switch: sample slices per frame
        const int v = atoi(e);
        if (v >= 1 && v <= 64 && n_samples > 0) slices = v < n_samples ? v : n_samples;
    }
    if (noalloc) {   // the partial images must fit the fixed scratch: fewer slices, or none (the kernel then writes d_accum directly)
        const size_t fit = RT_NOALLOC_PARTIAL_BYTES / (npx * sizeof(float4));
        if ((size_t)slices > fit) slices = fit >= 2 ? (int)fit : 1;
    }
    A.slice_len = n_samples > 0 ? (n_samples + slices - 1) / slices : 1;
    slices = n_samples > 0 ? (n_samples + A.slice_len - 1) / A.slice_len : 1;
    A.n_units = (A.items || A.sorted) ? A.n_groups * slices : 0;
    float4* target = (float4*)d_accum;
    float4* partial = nullptr;
    if (slices > 1 && noalloc) partial = reinterpret_cast<float4*>(c->scratch_dev + RT_NOALLOC_SCENE_BYTES + RT_NOALLOC_COUNTER_BYTES);
    else if (slices > 1) { HIPCHK(c, c->partial.reserve(npx * slices * sizeof(float4))); partial = (float4*)c->partial.p; }
    if (slices > 1) target = partial;
    // small scenes: the scene query compiled for THIS scene (literal operands, no loops); debug flag 8 or RT_MI355_JIT=0 keeps
    // the generic kernel, debug flag 16 or RT_MI355_JIT=2 specialises at the first render (A/B and parity tests); an unavailable hiprtc is not an error, the generic kernel renders the same image
    hipFunction_t jit = nullptr;
    const char* jit_env = getenv("RT_MI355_JIT");
    c->jit_note.clear();
    if (!noalloc && !accel && !(c->debug_flags & (2 | 8)) && !(jit_env && jit_env[0] == '0') && lds_fit &&
        jit_wanted(p->variant, H.slots, H.sruns, (double)width * height * (s1 - s0), (c->debug_flags & 16) || (jit_env && jit_env[0] == '2'), c->jit_note))
        jit = jit_path_kernel(p->variant, H.slots, H.sruns, p->t_min, p->t_max,
                              (A.queue ? RT_FORM_QUEUE : A.items ? RT_FORM_ITEMS : A.sorted ? RT_FORM_SORTED : ref ? (RT_FORM_REGEN | RT_FORM_REF) : RT_FORM_REGEN) | (A.norm_dirs ? RT_FORM_NORM : 0) | (A.strict_point ? RT_FORM_POINT : 0), c->jit_note);
    if (noalloc) c->jit_note = "allocation-free entry point: generic kernel";
    if ((rc = begin_timed(c, st, counters))) return rc;      // after a possible compile: kernel_ms is GPU time of the kernels only
    c->last_path_kernel = accel ? 3 : (jit ? 2 : ((size_t)D.total_f4 * 16 <= RT_LDS_SCENE_MAX_BYTES && !(c->debug_flags & 2) ? 1 : 0));
    c->last_path_form = A.queue ? RT_FORM_QUEUE : A.items ? RT_FORM_ITEMS : A.sorted ? RT_FORM_SORTED : RT_FORM_REGEN; c->last_unit_log2 = (A.items || A.sorted) ? A.unit_log2 : 0;
    if (accel) { c->jit_note = "BVH walk (RT_OPT_ACCEL)"; launch_path_bvh(p->variant, D, A, target, slices, counters, st); }
    else launch_path(p->variant, D, A, target, slices, counters, st, jit, c->n_cus * 8);
    HIPCHK(c, hipGetLastError());
    if (slices > 1) { launch_sum_slices(partial, (int)npx, slices, (float4*)d_accum, st); HIPCHK(c, hipGetLastError()); }
    c->pending_samples = (uint64_t)npx * (uint64_t)n_samples;
    return end_timed(c, st, counters, noalloc ? c->ev_scene_use_na : nullptr);
}

int32_t rt_path_accumulate_device(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_path_params* p,
                                  int32_t width, int32_t height, float* d_accum, void* stream) {
    return path_accumulate(c, scene, cam, p, width, height, d_accum, stream, false);
}
int32_t rt_path_accumulate_device_noalloc(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_path_params* p,
                                          int32_t width, int32_t height, float* d_accum, void* stream) {
    return path_accumulate(c, scene, cam, p, width, height, d_accum, stream, true);
}
int32_t rt_noalloc_limits(int32_t* scene_bytes, int32_t* partial_bytes) {
    if (scene_bytes) *scene_bytes = (int32_t)RT_NOALLOC_SCENE_BYTES;
    if (partial_bytes) *partial_bytes = (int32_t)RT_NOALLOC_PARTIAL_BYTES;
    return RT_OK;
}

int32_t rt_whitted_device(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_whitted_params* p,
                          int32_t width, int32_t height, float* d_color, void* stream) {
    int rc = check_ctx(c); if (rc) return rc;
    if (!cam || !p || !d_color || width <= 0 || height <= 0) return fail(c, RT_ERR_BAD_ARG, "rt_whitted_device: null pointer or non-positive size");
    if ((long long)width * height > (1LL << 28)) return fail(c, RT_ERR_TOO_LARGE, "image larger than 2^28 pixels");
    if (p->max_depth < 0 || p->max_depth > RT_WHITTED_MAX_DEPTH_LIMIT) return fail(c, RT_ERR_TOO_LARGE, "max_depth outside [0, RT_WHITTED_MAX_DEPTH_LIMIT]");
    if (p->shadow_samples < 0 || p->shadow_samples > RT_MAX_SHADOW_SAMPLES) return fail(c, RT_ERR_TOO_LARGE, "shadow_samples outside [0, RT_MAX_SHADOW_SAMPLES]");
    for (int k = 0; k < 3; ++k)
        if (cam->position[k] != 0.0f || cam->orientation_deg[k] != 0.0f) return fail(c, RT_ERR_BAD_ARG, "examples 01/02 have a fixed camera: position/orientation must be zero");
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const HostScene* Hp = nullptr; DevScene D;
    if ((rc = prepare_scene(c, scene, 0, 0.0f, false, st, Hp, D))) return rc;
    const HostScene& H = *Hp;
    WhittedArgs A{};
    A.width = width; A.height = height; A.max_depth = p->max_depth; A.shadow_samples = p->shadow_samples; A.seed = p->seed;
    float fov_rad = (float)(cam->fov_deg / (180.0f / 3.14159265358979323846));       // 01 Main.cpp:333
    float th = tanf(fov_rad / 2.0f), w = (float)width, h = (float)height;
    A.w = w; A.h = h; A.tan_half = th; A.debug_flags = c->debug_flags;
    // 02: nearly every ray is a shadow sample; for small scenes their scene query is compiled for THIS scene like the path
    // kernels' (same policy, same switches: second render of a scene, a large job, debug flag 16 / RT_MI355_JIT=2; flag 8 / =0 off)
    hipFunction_t jit = nullptr, jit_stage2 = nullptr;
    const char* jit_env = getenv("RT_MI355_JIT");
    c->jit_note.clear();
    // 01 (round 3): the same for its strict queries - every ray of the tree walk and its one shadow ray per hit and light (02's tree walk too)
    const int jit_variant = p->shadow_samples > 0 ? RT_JIT_WHITTED_SOFT : RT_JIT_WHITTED_HARD;
    if (!(c->debug_flags & 8) && !(jit_env && jit_env[0] == '0') &&
        jit_wanted(jit_variant, H.slots, H.sruns, (double)width * height * (p->shadow_samples > 0 ? p->shadow_samples : 1) * (scene->n_lights > 0 ? scene->n_lights : 1),
                   (c->debug_flags & 16) || (jit_env && jit_env[0] == '2'), c->jit_note))
        jit = jit_path_kernel(jit_variant, H.slots, H.sruns, 0.01f, 1000.0f, RT_FORM_REGEN, c->jit_note, &jit_stage2, &H.runs, &H.geom);
    c->last_whitted_kernel = jit ? 2 : 1;
    // Wavefront form (rt_whitted_device.h; debug flag 8192 or RT_MI355_WHITTED=w): the tree walk emits shadow REQUESTS, a second
    // kernel traces them with 2^g threads per request, a third adds their contributions per pixel.  Request slots: one per (pixel,
    // light) for primary hits + room for deeper tree nodes (a pixel that needs more is rendered by the monolithic kernel in a last
    // pass).  Not the default: on the reference's closed room every lane of the monolithic kernel is busy through its shadow
    // loop, and the stages' extra launches and request traffic cost more than they return.
    // 02 only: with ONE shadow ray per hit (01) the stages have nothing to parallelise and three launches cost more than the frame.
    const char* wf_env = getenv("RT_MI355_WHITTED");
    // Default: the scene-specialised monolithic kernel when there is one (no scratch at 7 waves per SIMD, the fastest form measured);
    // without it - the first render of a scene, scenes beyond 64 primitives - the wavefront form, whose generic kernels need no
    // scratch either (the generic monolithic 02 kernel spills 168 B per lane); debug flag 4096 forces the monolithic kernel.
    const bool wavefront = p->shadow_samples > 0 && ((c->debug_flags & 8192) || (wf_env && wf_env[0] == 'w') || (!jit && !(c->debug_flags & 4096))) && scene->n_lights >= 1 && scene->n_lights <= 32 &&
                           (long long)width * height * scene->n_lights * 2 * 64 < (1LL << 32);
    c->last_whitted_form = wavefront ? 1 : 0;
    bool overflow_impossible = false;
    if (wavefront) {
        // deeper nodes: the worst case of a non-branching tree (max_depth further shaded nodes per pixel) when that is modest, else a quarter of the primary slots
        const size_t base = (size_t)width * height * (size_t)scene->n_lights, worst = H.whitted_branches ? (size_t)-1 : base * (size_t)p->max_depth;
        bool any_secondary = false;
        for (int m = 0; m < scene->n_materials; ++m) any_secondary = any_secondary || scene->materials[m].primary_to_secondary < 1.0f;
        size_t extra = base / 4 + 4096;
        if (!any_secondary) { extra = 64; overflow_impossible = true; }                                  // no material spawns a secondary ray: primary hits only
        else if (worst != (size_t)-1 && (base + worst) * 56 <= ((size_t)256 << 20)) { extra = worst; overflow_impossible = true; }
        const size_t slots = base + extra;
        HIPCHK(c, c->wf_req.reserve(slots * 48));
        HIPCHK(c, c->wf_words.reserve((2 * slots + (size_t)width * height) * 4));
        A.wf_req = (float4*)c->wf_req.p;
        A.wf_prev = (uint32_t*)c->wf_words.p; A.wf_occ = A.wf_prev + slots; A.wf_head = A.wf_occ + slots;
        A.wf_counter = reinterpret_cast<uint32_t*>(c->d_counters + RT_WORK_COUNTER + 1);      // cleared with the ray counters by begin_timed
        A.wf_base = (uint32_t)base; A.wf_capacity = (uint32_t)extra;
    }
    if ((rc = begin_timed(c, st))) return rc;      // after a possible compile
    if (wavefront) launch_whitted_wavefront(true, D, A, (float4*)d_color, c->d_counters, st, jit_stage2, c->n_cus * 8, overflow_impossible);
    else launch_whitted(p->shadow_samples > 0, D, A, (float4*)d_color, c->d_counters, st, jit);
    HIPCHK(c, hipGetLastError());
    c->pending_samples = (uint64_t)width * height;
    return end_timed(c, st);
}

int32_t rt_resolve_device(rt_ctx* c, const float* d_accum, int32_t width, int32_t height, float divisor, int32_t gamma,
                          uint8_t* d_bgra, int32_t stride, float* d_rgb, void* stream) {
    int rc = check_ctx(c); if (rc) return rc;
    if (!d_accum || width <= 0 || height <= 0 || (!d_bgra && !d_rgb)) return fail(c, RT_ERR_BAD_ARG, "rt_resolve_device: null pointer or non-positive size");
    if (stride == 0) stride = width * 4;
    if (stride < width * 4 || (stride & 3)) return fail(c, RT_ERR_BAD_ARG, "bgra stride must be >= width*4 and a multiple of 4");
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    launch_resolve((const float4*)d_accum, width, height, divisor, gamma, d_bgra, stride, d_rgb, st);
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

// Path render on a device group: device i renders share i of the sample range into its own full-frame accumulator (all
// devices enqueued before anything is waited for), then the accumulators are summed - one RCCL all-reduce over the group's
// communicators (distinct devices), or copies to the first device added in a fixed order (root-gather form) - and the first
// device resolves the frame.  d_accum_root: the first device's accumulator (c->accum), holding the sum on return.
static int32_t path_accumulate_group(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_path_params* pp, int W, int Hh) {
    const int n = 1 + (int)c->peers.size();
    const size_t npx = (size_t)W * Hh, bytes = npx * sizeof(float4);
    int s0 = pp->sample_begin, s1 = pp->sample_end;
    if (s0 == 0 && s1 == 0) s1 = pp->spp;
    if (pp->spp <= 0 || s0 < 0 || s1 < s0) return fail(c, RT_ERR_BAD_ARG, "bad spp / sample range");
    int rc;
    for (int i = 0; i < n; ++i) {                       // fan out
        rt_ctx* d = i ? c->peers[i - 1] : c;
        if ((rc = check_ctx(d))) return fail(c, rc, "device group: " + d->err);
        d->debug_flags = c->debug_flags; d->accel = c->accel; d->arith = c->arith; d->regen_threshold_override = c->regen_threshold_override;
        if (hipError_t e = d->accum.reserve(bytes)) return fail(c, RT_ERR_HIP, std::string("device group: accumulator: ") + hipGetErrorString(e));
        rt_path_params q = *pp;
        share_of(s0, s1, i, n, q.sample_begin, q.sample_end);
        if (q.sample_end > q.sample_begin) {
            if ((rc = rt_path_accumulate_device(d, scene, cam, &q, W, Hh, (float*)d->accum.p, nullptr))) return fail(c, rc, "device group: device " + std::to_string(d->device) + ": " + d->err);
        } else {                                        // an empty share ((0,0) would mean "the whole frame"): zeros
            HIPCHK(c, hipMemsetAsync(d->accum.p, 0, bytes, d->stream));
            d->pending_samples = 0; d->timing_pending = false; d->stats = rt_stats{};
        }
        if (i) HIPCHK(c, hipEventRecord(c->peer_done[i - 1], d->stream));
    }
    if ((rc = check_ctx(c))) return rc;
    HIPCHK(c, hipEventRecord(c->ar0, c->stream));
    if (!c->comms.empty()) {                            // one all-reduce(sum, float32) of the per-GPU accumulators, in place
        int r = g_rccl.GroupStart();
        for (int i = 0; i < n && r == 0; ++i) {
            rt_ctx* d = i ? c->peers[i - 1] : c;
            (void)hipSetDevice(d->device);
            r = g_rccl.AllReduce(d->accum.p, d->accum.p, npx * 4, kNcclFloat32, kNcclSum, c->comms[(size_t)i], d->stream);
        }
        const int r2 = g_rccl.GroupEnd();
        (void)hipSetDevice(c->device);
        if (r != 0 || r2 != 0) return fail(c, RT_ERR_HIP, std::string("device group: ncclAllReduce: ") + g_rccl.GetErrorString(r ? r : r2));
    } else {                                            // root-gather form
        for (int i = 1; i < n; ++i) {
            rt_ctx* d = c->peers[i - 1];
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->peer_done[i - 1], 0));
            const float4* src = (const float4*)d->accum.p;
            if (d->device != c->device) {
                HIPCHK(c, c->gather.reserve(bytes));
                HIPCHK(c, hipMemcpyPeerAsync(c->gather.p, c->device, d->accum.p, d->device, bytes, c->stream));
                src = (const float4*)c->gather.p;
            }
            launch_add_image((float4*)c->accum.p, src, (int)npx, c->stream);
            HIPCHK(c, hipGetLastError());
        }
    }
    HIPCHK(c, hipEventRecord(c->ar1, c->stream));
    return RT_OK;
}
// Collects the group's counters after the first device's stream has drained: rays and samples add up, the kernel time is
// the slowest device's (they run side by side).
static int32_t group_sync(rt_ctx* c) {
    int rc;
    uint64_t rays = c->stats.rays, srays = c->stats.shadow_rays, samples = c->stats.samples;
    double kms = c->stats.kernel_ms;
    for (rt_ctx* d : c->peers) {
        if ((rc = rt_sync(d, nullptr))) return fail(c, rc, "device group: " + d->err);
        rays += d->stats.rays; srays += d->stats.shadow_rays; samples += d->stats.samples;
        kms = d->stats.kernel_ms > kms ? d->stats.kernel_ms : kms;
    }
    if ((rc = check_ctx(c))) return rc;
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ar0, c->ar1));
    c->stats.rays = rays; c->stats.shadow_rays = srays; c->stats.samples = samples; c->stats.kernel_ms = kms; c->stats.allreduce_ms = ms;
    return RT_OK;
}

static int32_t render_host(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_whitted_params* wp, const rt_path_params* pp, const rt_image* out) {
    auto t0 = std::chrono::steady_clock::now();
    int rc = check_ctx(c); if (rc) return rc;
    const bool group = !wp && pp && !c->peers.empty();      // Whitted frames (one ray tree per pixel, < 1 ms) stay on the first device
    if (!out || out->width <= 0 || out->height <= 0 || (!out->bgra && !out->rgb)) return fail(c, RT_ERR_BAD_ARG, "image: null or non-positive size");
    int W = out->width, Hh = out->height;
    size_t npx = (size_t)W * Hh;
    int stride = out->bgra_stride_bytes ? out->bgra_stride_bytes : W * 4;
    if (stride < W * 4 || (stride & 3)) return fail(c, RT_ERR_BAD_ARG, "bgra stride must be >= width*4 and a multiple of 4");
    HIPCHK(c, c->accum.reserve(npx * sizeof(float4)));
    if (out->bgra) HIPCHK(c, c->bgra.reserve((size_t)stride * Hh));
    if (out->rgb) HIPCHK(c, c->rgb.reserve(npx * 3 * sizeof(float)));
    float divisor = 1.0f; int gamma = 0;
    if (wp) rc = rt_whitted_device(c, scene, cam, wp, W, Hh, (float*)c->accum.p, nullptr);
    else if (group) { rc = path_accumulate_group(c, scene, cam, pp, W, Hh); divisor = (float)pp->spp; gamma = 1; }
    else { rc = rt_path_accumulate_device(c, scene, cam, pp, W, Hh, (float*)c->accum.p, nullptr); divisor = pp ? (float)pp->spp : 1.0f; gamma = 1; }
    if (rc) return rc;
    rc = rt_resolve_device(c, (const float*)c->accum.p, W, Hh, divisor, gamma, out->bgra ? (uint8_t*)c->bgra.p : nullptr, stride,
                           out->rgb ? (float*)c->rgb.p : nullptr, nullptr);
    if (rc) return rc;
    if (out->bgra) HIPCHK(c, hipMemcpyAsync(out->bgra, c->bgra.p, (size_t)stride * Hh, hipMemcpyDeviceToHost, c->stream));
    if (out->rgb) HIPCHK(c, hipMemcpyAsync(out->rgb, c->rgb.p, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    c->stats.allreduce_ms = 0.0;
    if ((rc = rt_sync(c, nullptr))) return rc;
    if (group && (rc = group_sync(c))) return rc;
    c->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int32_t rt_render_whitted(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_whitted_params* p, const rt_image* out) {
    if (!c) return RT_ERR_BAD_ARG;
    if (!p) return fail(c, RT_ERR_BAD_ARG, "params is null");
    return render_host(c, scene, cam, p, nullptr, out);
}
int32_t rt_render_path(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_path_params* p, const rt_image* out) {
    if (!c) return RT_ERR_BAD_ARG;
    if (!p) return fail(c, RT_ERR_BAD_ARG, "params is null");
    return render_host(c, scene, cam, nullptr, p, out);
}

// ---- progressive frame: sums resident in HBM, a preview per pass ---------------------------------------------------------

int32_t rt_progressive_begin(rt_ctx* c, const rt_scene* scene, const rt_camera* cam, const rt_path_params* p, int32_t width, int32_t height) {
    int rc = check_ctx(c); if (rc) return rc;
    if (!scene || !cam || !p) return fail(c, RT_ERR_BAD_ARG, "rt_progressive_begin: null argument");
    if (width <= 0 || height <= 0 || (size_t)width * height > ((size_t)1 << 28)) return fail(c, RT_ERR_BAD_ARG, "rt_progressive_begin: bad image size");
    if (scene->n_prims < 0 || scene->n_materials < 0 || scene->n_lights < 0 || (scene->n_prims && !scene->prims) || (scene->n_materials && !scene->materials) || (scene->n_lights && !scene->lights))
        return fail(c, RT_ERR_BAD_ARG, "rt_progressive_begin: scene arrays");
    int s0 = p->sample_begin, s1 = p->sample_end;
    if (s0 == 0 && s1 == 0) s1 = p->spp;
    if (p->spp <= 0 || s0 < 0 || s1 < s0 || s1 > p->spp) return fail(c, RT_ERR_BAD_ARG, "rt_progressive_begin: bad spp / sample range");
    auto& g = c->prog;
    g.active = false;
    HIPCHK(c, c->prog_sum.reserve((size_t)width * height * sizeof(float4)));
    HIPCHK(c, hipMemsetAsync(c->prog_sum.p, 0, (size_t)width * height * sizeof(float4), c->stream));
    g.prims.assign(scene->prims, scene->prims + scene->n_prims);
    g.mats.assign(scene->materials, scene->materials + scene->n_materials);
    g.lights.assign(scene->lights, scene->lights + scene->n_lights);
    g.params = *p; g.cam = *cam; g.W = width; g.H = height; g.begin = s0; g.end = s1; g.done = 0;
    g.active = true;
    return RT_OK;
}

int32_t rt_progressive_pass(rt_ctx* c, int32_t n_samples, const rt_image* preview, int32_t* samples_done) {
    auto t0 = std::chrono::steady_clock::now();
    int rc = check_ctx(c); if (rc) return rc;
    auto& g = c->prog;
    if (!g.active) return fail(c, RT_ERR_BAD_ARG, "rt_progressive_pass: no progressive frame (rt_progressive_begin)");
    if (n_samples < 0) return fail(c, RT_ERR_BAD_ARG, "rt_progressive_pass: negative sample count");
    const int W = g.W, Hh = g.H;
    const size_t npx = (size_t)W * Hh;
    int stride = 0;
    if (preview) {
        if (preview->width != W || preview->height != Hh || (!preview->bgra && !preview->rgb)) return fail(c, RT_ERR_BAD_ARG, "rt_progressive_pass: preview image must have the frame's size and a buffer");
        stride = preview->bgra_stride_bytes ? preview->bgra_stride_bytes : W * 4;
        if (stride < W * 4 || (stride & 3)) return fail(c, RT_ERR_BAD_ARG, "bgra stride must be >= width*4 and a multiple of 4");
    }
    const int b = g.begin + g.done, e = (long long)b + n_samples < g.end ? b + n_samples : g.end;
    const bool group = !c->peers.empty();
    const bool rendered = e > b;
    c->stats.allreduce_ms = 0.0;
    if (rendered) {
        const rt_scene sc{g.prims.data(), (int32_t)g.prims.size(), g.mats.data(), (int32_t)g.mats.size(), g.lights.data(), (int32_t)g.lights.size()};
        rt_path_params q = g.params; q.sample_begin = b; q.sample_end = e;
        HIPCHK(c, c->accum.reserve(npx * sizeof(float4)));
        rc = group ? path_accumulate_group(c, &sc, &g.cam, &q, W, Hh) : rt_path_accumulate_device(c, &sc, &g.cam, &q, W, Hh, (float*)c->accum.p, nullptr);
        if (rc) return rc;
        launch_add_image((float4*)c->prog_sum.p, (const float4*)c->accum.p, (int)npx, c->stream);
        HIPCHK(c, hipGetLastError());
        g.done += e - b;
    } else { c->stats.rays = c->stats.shadow_rays = c->stats.samples = 0; c->stats.kernel_ms = 0.0; }
    if (preview) {                                          // exposed for the samples done so far: sum / done, sqrt, clamp, truncate
        if (preview->bgra) HIPCHK(c, c->bgra.reserve((size_t)stride * Hh));
        if (preview->rgb) HIPCHK(c, c->rgb.reserve(npx * 3 * sizeof(float)));
        rc = rt_resolve_device(c, (const float*)c->prog_sum.p, W, Hh, g.done ? (float)g.done : 1.0f, 1, preview->bgra ? (uint8_t*)c->bgra.p : nullptr, stride,
                               preview->rgb ? (float*)c->rgb.p : nullptr, nullptr);
        if (rc) return rc;
        if (preview->bgra) HIPCHK(c, hipMemcpyAsync(preview->bgra, c->bgra.p, (size_t)stride * Hh, hipMemcpyDeviceToHost, c->stream));
        if (preview->rgb) HIPCHK(c, hipMemcpyAsync(preview->rgb, c->rgb.p, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    if (rendered) { if ((rc = rt_sync(c, nullptr))) return rc; if (group && (rc = group_sync(c))) return rc; }
    else HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (samples_done) *samples_done = g.done;
    return RT_OK;
}

int32_t rt_progressive_end(rt_ctx* c) {
    int rc = check_ctx(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->prog = rt_ctx::Progressive{};
    c->prog_sum.release();
    return RT_OK;
}

// test/bench hook: regeneration threshold (idle lanes per wave); <0 restores the default. Not part of rt_mi355.h.
// Diagnostics (not in rt_mi355.h).  rt_debug_last_path_kernel: which kernel rendered the last path image (2 scene-specialised,
// 1 generic LDS-resident, 0 tiled / global) and, when not specialised, why.  rt_debug_jit_probe: builds the specialised
// kernel source for a scene and compiles it (no GPU needed for the compile; loading the module needs one).
int32_t rt_debug_last_path_kernel(const rt_ctx* c, char* note, int32_t cap) {
    if (!c) return -1;
    if (note && cap > 0) { strncpy(note, c->jit_note.c_str(), (size_t)cap - 1); note[cap - 1] = 0; }
    return c->last_path_kernel;
}
int32_t rt_debug_last_whitted_kernel(const rt_ctx* c) { return c ? c->last_whitted_kernel : -1; }
int32_t rt_debug_last_whitted_form(const rt_ctx* c) { return c ? c->last_whitted_form : -1; }
// which form of the bounce loop rendered the last path image: 0 per-lane regeneration (one lane per pixel), 1 queue form, 2 item
// form (persistent waves + work counter); unit_log2: item form, log2 of the pixels per unit of work
int32_t rt_debug_last_path_form(const rt_ctx* c, int32_t* unit_log2) { if (!c) return -1; if (unit_log2) *unit_log2 = c->last_unit_log2; return c->last_path_form; }
// variant 3 / 4: the path kernels; 2: the 02 kernel, 1: the 01 kernel (a Whitted scene)
int32_t rt_debug_jit_probe(const rt_scene* scene, int32_t variant, char* note, int32_t cap) {
    HostScene H;
    const float tmin = (variant & 7) == 3 ? 0.001f : 0.01f;       // the examples' constants (rt_path_params_default)
    const bool whitted = (variant & 7) == RT_JIT_WHITTED_SOFT || (variant & 7) == RT_JIT_WHITTED_HARD;
    int rc = compile_scene(nullptr, scene, whitted ? 0 : (variant & 7), tmin, H);
    if (rc) return rc;
    std::string n;
    hipFunction_t fn = jit_path_kernel(variant & 7, H.slots, H.sruns, tmin, 1000.0f, (variant & 8) ? RT_FORM_QUEUE : (variant & 16) ? RT_FORM_ITEMS : (variant & 32) ? RT_FORM_SORTED : RT_FORM_REGEN, n,
                                       nullptr, whitted ? &H.runs : nullptr, whitted ? &H.geom : nullptr);
    if (note && cap > 0) { strncpy(note, n.c_str(), (size_t)cap - 1); note[cap - 1] = 0; }
    return fn ? 1 : 0;
}
// rt_debug_bvh_probe: builds the BVH of a scene on the host (no GPU needed) and checks it: every bounded slot is referenced by
// exactly one leaf whose box contains the slot's padded box, every child box lies inside its parent's, leaves hold at most
// RT_BVH_LEAF entries.  out[0..4] = {nodes, leaf entries, unbounded slots, depth, slots}.  Returns 0, or the number of violations.
int32_t rt_debug_bvh_probe(const rt_scene* scene, int32_t variant, int32_t* out) {
    HostScene H;
    int rc = compile_scene(nullptr, scene, variant, variant == 3 ? 0.001f : 0.01f, H);
    if (rc) return rc;
    std::vector<BvhItem> items = H.items;
    build_bvh(items, H.bvh);
    const Bvh& B = H.bvh;
    int bad = 0;
    std::vector<int> seen(H.slots.size() / 4 + 1, 0);
    const int n_nodes = (int)B.nodes.size() / 4;
    auto inside = [](const float4& lo, const float4& hi, const float* ilo, const float* ihi) {
        return lo.x <= ilo[0] && lo.y <= ilo[1] && lo.z <= ilo[2] && hi.x >= ihi[0] && hi.y >= ihi[1] && hi.z >= ihi[2]; };
    std::vector<int> visits(n_nodes, 0);
    std::vector<const BvhItem*> item_of(H.slots.size() / 4 + 1, nullptr);      // by slot offset / 64
    for (const BvhItem& it : H.items) item_of[(it.ref & ~63U) / 64] = &it;
    for (int k = 0; k < n_nodes; ++k) {
        for (int ch = 0; ch < 2; ++ch) {
            const float4 lo = B.nodes[4 * k + 2 * ch], hi = B.nodes[4 * k + 2 * ch + 1];
            const uint32_t ref = __builtin_bit_cast(uint32_t, ch ? B.nodes[4 * k + 1].w : B.nodes[4 * k].w);
            if (ch == 1 && (ref & 0x80000000U) && ref == __builtin_bit_cast(uint32_t, B.nodes[4 * k].w)) continue;   // single-leaf root: both children are that leaf
            if (!(ref & 0x80000000U)) {
                if ((int)ref >= n_nodes || (int)ref <= k) { ++bad; continue; }      // children come after their parent
                visits[ref]++;
                for (int g = 0; g < 2; ++g) {             // grandchildren boxes lie inside this child's box
                    const float4 gl = B.nodes[4 * ref + 2 * g], gh = B.nodes[4 * ref + 2 * g + 1];
                    const float l[3] = {gl.x, gl.y, gl.z}, h[3] = {gh.x, gh.y, gh.z};
                    if (!inside(lo, hi, l, h)) ++bad;
                }
            } else {
                const uint32_t first = (ref & 0x7fffffffU) >> 3, n = (ref & 7U) + 1U;
                if (n > RT_BVH_LEAF || first + n > B.refs.size()) { ++bad; continue; }
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t r = B.refs[first + i];
                    seen[(r & ~63U) / 64]++;
                    const BvhItem* it = item_of[(r & ~63U) / 64];      // the unpadded box of the slot must lie inside the leaf's
                    if (!it || it->ref != r || !inside(lo, hi, it->lo, it->hi)) ++bad;
                }
            }
        }
    }
    for (int k = 1; k < n_nodes; ++k) if (visits[k] != 1) ++bad;    // a tree: every inner node but the root has one parent
    int n_bounded = 0;
    for (const BvhItem& it : H.items) {
        if (it.bounded) { ++n_bounded; if (seen[(it.ref & ~63U) / 64] != 1) ++bad; }
        else if (seen[(it.ref & ~63U) / 64] != 0) ++bad;
    }
    if ((int)B.refs.size() != n_bounded || B.unbounded.size() + B.refs.size() != H.items.size()) ++bad;
    if (out) { out[0] = n_nodes; out[1] = (int)B.refs.size(); out[2] = (int)B.unbounded.size(); out[3] = B.depth; out[4] = (int)H.items.size(); }
    return bad;
}
// rt_debug_strict_slots: compiles a path scene on the host (no GPU needed) and counts the slots that take the reference's own
// operation sequence because they are coplanar with an element of other arithmetic (rt_device.h PRIM_*_STRICT).
// out[0..2] = {strict planes, strict rectangles / box faces, strict triangles}.  Returns their total, or an error code < 0.
int32_t rt_debug_strict_slots(const rt_scene* scene, int32_t variant, int32_t* out) {
    HostScene H;
    int rc = compile_scene(nullptr, scene, variant, variant == 3 ? 0.001f : 0.01f, H);
    if (rc) return rc;
    int n[3] = {0, 0, 0};
    for (const int4& r : H.sruns) if (is_strict_type(r.x)) n[r.x - PRIM_PLANE_STRICT] += r.z;
    if (out) { out[0] = n[0]; out[1] = n[1]; out[2] = n[2]; }
    return n[0] + n[1] + n[2];
}
// how a device group sums its accumulators: 1 = RCCL all-reduce, 0 = root-gather form (or a single device); note: why
int32_t rt_debug_group_mode(const rt_ctx* c, char* note, int32_t cap) {
    if (!c) return -1;
    if (note && cap > 0) { strncpy(note, c->group_note.c_str(), (size_t)cap - 1); note[cap - 1] = 0; }
    return c->comms.empty() ? 0 : 1;
}
// the device group's partition of a sample range (rt_create_multi); needs no GPU
int32_t rt_debug_group_share(int32_t s0, int32_t s1, int32_t i, int32_t n, int32_t* b, int32_t* e) {
    if (!b || !e || n <= 0 || i < 0 || i >= n || s1 < s0) return RT_ERR_BAD_ARG;
    int bb, ee; share_of(s0, s1, i, n, bb, ee); *b = bb; *e = ee;
    return RT_OK;
}
// Self-test of the run-time RCCL binding on one device (the device-group path itself needs distinct GPUs): dlopen + symbols,
// a one-rank communicator, one grouped out-of-place all-reduce(sum, float32) of n floats, result compared with the input.
// 0 = ok; 1 = librccl not loadable / symbols missing; 2 = an RCCL call failed; 3 = HIP error; 4 = wrong result.  msg: detail.
int32_t rt_debug_rccl_selftest(int32_t device, int32_t n, char* msg, int32_t cap) {
    auto say = [&](const std::string& s) { if (msg && cap > 0) { strncpy(msg, s.c_str(), (size_t)cap - 1); msg[cap - 1] = 0; } };
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { const char* why = dlerror(); say(why ? why : "dlopen failed"); return 1; }
    if (!rccl_bind(lib)) { say("missing symbols"); return 1; }
    if (hipSetDevice(device) != hipSuccess || n <= 0) { say("hipSetDevice / n"); return 3; }
    void* comm = nullptr; int dev = device;
    int r = g_rccl.CommInitAll(&comm, 1, &dev);
    if (r != 0) { say(std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); return 2; }
    std::vector<float> h((size_t)n), back((size_t)n, -1.0f);
    for (int i = 0; i < n; ++i) h[(size_t)i] = 0.25f * (float)(i % 1021) - 3.0f;
    float *a = nullptr, *b = nullptr; hipStream_t st = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&a, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&b, (size_t)n * 4) != hipSuccess || hipStreamCreate(&st) != hipSuccess ||
        hipMemcpyAsync(a, h.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipMemsetAsync(b, 0xFF, (size_t)n * 4, st) != hipSuccess) { say("HIP set-up"); rc = 3; }
    if (!rc) {
        r = g_rccl.GroupStart();
        if (r == 0) r = g_rccl.AllReduce(a, b, (size_t)n, kNcclFloat32, kNcclSum, comm, st);
        const int r2 = g_rccl.GroupEnd();
        if (r != 0 || r2 != 0) { say(std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r ? r : r2)); rc = 2; }
    }
    if (!rc && (hipMemcpyAsync(back.data(), b, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) { say("HIP read-back"); rc = 3; }
    if (!rc && memcmp(back.data(), h.data(), (size_t)n * 4) != 0) { say("all-reduce of one rank did not reproduce its input"); rc = 4; }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    (void)g_rccl.CommDestroy(comm);
    if (!rc) say("ok");
    return rc;
}
int32_t rt_debug_set_flags(rt_ctx* c, int32_t f) { if (!c) return RT_ERR_BAD_ARG; c->debug_flags = f; return RT_OK; }
int32_t rt_debug_set_regen_threshold(rt_ctx* c, int32_t n) { if (!c) return RT_ERR_BAD_ARG; c->regen_threshold_override = n; return RT_OK; }

} // extern "C"
