// rt_jit.hip - scene-specialised build of k_path<VARIANT, true> for small scenes.
//
// On gfx950 a VALU instruction reads a 32-bit LITERAL at full rate but an SGPR operand at half rate
// (profiles/r01_valu_issue_rates_ubench.txt), and the generic closest-hit loop spends about a fifth of its VALU work on
// what a fixed scene does not need: run dispatch, cursor arithmetic, loop-carried copies, LDS reads of primitive records.
// So for a scene of up to RT_JIT_MAX_PRIMS primitives the scene query is GENERATED as straight-line code, one block per
// primitive in scene order with the primitive's numbers as literals - the same arithmetic, operation for operation, as the
// loops of closest_hit_runs (rt_device.h) - and compiled with hiprtc together with the unchanged shading / regeneration
// code (rt_path_device.h, embedded as text).  Shading still reads the winner's slot from the LDS-staged blob.
#include "rt_jit.h"
#include "rt_device.h"
#include <hip/hiprtc.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unistd.h>

namespace rt {
namespace {
const char* const kSrcArgs =
#include "rt_jit_src_args.inc"
    ;
const char* const kSrcDevice =
#include "rt_jit_src_device.inc"
    ;
const char* const kSrcPath =
#include "rt_jit_src_path.inc"
    ;
const char* const kSrcStrict =
#include "rt_jit_src_strict.inc"
    ;
const char* const kSrcWhitted =
#include "rt_jit_src_whitted.inc"
    ;

constexpr int RT_JIT_MAX_PRIMS = 64;    // measured (tools/perf_mid.py): at 94 slots the straight-line code ties with the generic loops, at 184 it loses 19 % (instruction cache)

std::string lit(float v) {   // exact: hexadecimal floating literal
    if (std::isinf(v)) return v > 0 ? "__builtin_inff()" : "(-__builtin_inff())";
    if (std::isnan(v)) return "__builtin_nanf(\"\")";
    char b[64];
    snprintf(b, sizeof b, "%af", (double)v);
    return b;
}
std::string v3(const float4& g) { return "F3(" + lit(g.x) + ", " + lit(g.y) + ", " + lit(g.z) + ")"; }
// opens a block and leaves the roots t1 <= t2 of a sphere slot g = (centre, r*r) (NaN when the line misses);
// PRIM_SPHERE_BIG: the coefficients in the reference's own operation order and a correctly rounded root (rt_device.h: big_sphere_roots; reference arithmetic: strict_sphere_roots)
thread_local bool g_gen_ref = false;      // the source being generated is for RT_ARITH_REFERENCE
std::string sphere_root_source(int type, const float4& g) {
    if (type == PRIM_SPHERE_BIG && g_gen_ref)
        return "    { float t1, t2; strict_sphere_roots(" + v3(g) + ", " + lit(g.w) + ", o, d, t1, t2);";
    if (type == PRIM_SPHERE_BIG)
        return "    { float t1, t2; big_sphere_roots(" + v3(g) + ", " + lit(g.w) + ", o, d, t1, t2);";
    return "    { f3 co = " + v3(g) + " - o; float b = dot(d, co);\n"
           "      float c = fmaf(co.z, co.z, fmaf(co.y, co.y, fmaf(co.x, co.x, " + lit(-g.w) + ")));\n"
           "      float s = sqrt_fast(fmaf(b, b, -c)); float t1 = b - s, t2 = b + s;";
}


// Literal-coefficient sums with the terms whose coefficient is exactly 0 left out: in the generic loops they are multiplied
// out (x*0 = +-0, which changes no finite sum; only an infinite or NaN x would differ, and such a t is rejected by its key /
// range test anyway), here they cost nothing.  dot3: a.x*v.x + a.y*v.y + a.z*v.z in the operation order of dot() (rt_device.h):
// fmaf(a.z, v.z, fmaf(a.y, v.y, a.x * v.x)).  affine3: fmaf(a.z, v.z, fmaf(a.y, v.y, fmaf(a.x, v.x, w))).
std::string dot3(const float4& a, const char* v) {
    std::string e;
    const float c[3] = {a.x, a.y, a.z}; const char* n[3] = {".x", ".y", ".z"};
    for (int k = 0; k < 3; ++k) {
        if (c[k] == 0.0f) continue;
        const std::string term = std::string(v) + n[k];
        e = e.empty() ? "(" + lit(c[k]) + " * " + term + ")" : "fmaf(" + lit(c[k]) + ", " + term + ", " + e + ")";
    }
    return e.empty() ? "0.0f" : e;
}
std::string affine3(const float4& a, const char* v) {
    std::string e = lit(a.w);
    const float c[3] = {a.x, a.y, a.z}; const char* n[3] = {".x", ".y", ".z"};
    for (int k = 0; k < 3; ++k) if (c[k] != 0.0f) e = "fmaf(" + lit(c[k]) + ", " + std::string(v) + n[k] + ", " + e + ")";
    return e;
}

// A PRIM_*_STRICT slot (rt_device.h): the call of its strict_* test with the slot's numbers as literals; sets 't', yields "passes for tMax = inf".
std::string strict_call(int type, const float4* g) {
    if (type == PRIM_PLANE_STRICT) return "strict_plane(" + v3(g[0]) + ", " + v3(g[1]) + ", o, ds, tminp, t)";
    if (type == PRIM_TRI_STRICT) return "strict_tri(" + v3(g[0]) + ", " + v3(g[1]) + ", " + v3(g[4]) + ", " + v3(g[5]) + ", o, ds, tminp, t)";
    int ax; memcpy(&ax, &g[6].z, 4);
    return "strict_rect(" + v3(g[0]) + ", " + v3(g[4]) + ", " + v3(g[5]) + ", " + v3(g[1]) + ", " + std::to_string(ax) + ", " + lit(g[0].w) + ", " + lit(g[4].w) + ", " +
           lit(g[6].x) + ", " + lit(g[5].w) + ", " + lit(g[6].y) + ", o, ds, tminp, t)";
}

// One block per primitive; K = byte offset of its slot in the slot section + SLOT_BYTES (0 stays "no hit").
std::string gen_closest_hit(const std::vector<float4>& slots, const std::vector<int4>& sruns) {
    std::string s;
    s += "namespace rt {\n__device__ __forceinline__ uint32_t closest_hit_jit(f3 o, f3 d, float tmin, float tminp, float tmax, float& t_out) {\n";
    s += "    uint32_t best = 0U;\n    float closest = tmax;\n    const f3 inv_d = F3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));\n    (void)inv_d; (void)tmin; (void)tminp;\n";
    for (const int4& run : sruns) if (is_strict_type(run.x)) { s += g_gen_ref ? "    const f3 ds = d;\n" : "    const f3 ds = strict_dir(d);\n"; break; }
    size_t q = 0;   // float4 index into slots
    for (const int4& run : sruns) {
        for (int i = 0; i < run.z; ++i) {
            const float4* g = &slots[q];
            const std::string K = std::to_string(q * 16 + SLOT_BYTES) + "U";
            const std::string take = "closest = hit ? t : closest; best = hit ? " + K + " : best; }\n";
            if (is_sphere_type(run.x)) {
                s += sphere_root_source(run.x, g[0]) + "\n";
                s += "      bool near_behind = t1 < tmin; float tc = near_behind ? t2 : t1; bool ge = tc >= tmin, le = tc <= closest;\n";
                s += "      closest = (ge & le) ? tc : closest; best = (ge & (near_behind | le)) ? " + K + " : best; }\n";
            } else if (is_aaplane_type(run.x)) {
                const char* ax = run.x == PRIM_AAPLANE_X ? "x" : run.x == PRIM_AAPLANE_Y ? "y" : "z";
                s += "    { float t = (" + lit(g[0].x) + " - o." + ax + ") * inv_d." + ax + "; bool hit = (t >= tminp) & (t <= closest); " + take;
            } else if (run.x == PRIM_PLANE) {
                s += "    { float t = (" + lit(g[0].w) + " - " + dot3(g[0], "o") + ") * rcp_fast(" + dot3(g[0], "d") + "); bool hit = (t >= tminp) & (t <= closest); " + take;
            } else if (is_strict_type(run.x)) {
                s += "    { float t; bool hit = " + strict_call(run.x, g) + "; hit = hit & (t <= closest); " + take;
            } else {   // rectangle / triangle: wide slot, G1..G3 behind the shading records
                const float4 &g1 = g[4], &g2 = g[5], &g3 = g[6];
                s += "    { float t = (" + lit(g[0].w) + " - " + dot3(g[0], "o") + ") * rcp_fast(" + dot3(g[0], "d") + "); f3 p = madd(d, t, o);\n";
                s += "      float x = " + affine3(g1, "p") + ";\n";
                s += "      float y = " + affine3(g2, "p") + ";\n";
                if (run.x == PRIM_RECT) s += "      bool in = (fabsf(x) <= " + lit(g3.x) + ") & (fabsf(y) <= " + lit(g3.y) + ");\n";
                else                    s += "      bool in = (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);\n";
                s += "      bool hit = (t >= tminp) & (t <= closest) & in; " + take;
            }
            q += is_wide_type(run.x) ? 8 : 4;
        }
    }
    s += "    t_out = closest;\n    return best;\n}\n";
    return s;
}
std::string gen_any_hit(const std::vector<float4>& slots, const std::vector<int4>& sruns) {
    std::string s;
    size_t q = 0;
    // The same tests against a FIXED tMax, OR-ed (02's shadow samples): "something occludes" without the chain through
    // 'closest'.  Equivalent to closest_hit_jit(...) != 0: a test that fails there only because an earlier element was closer
    // changes nothing (that element occludes), and the sphere rule "origin inside, far root beyond tMax" occludes in both.
    s += "__device__ __forceinline__ bool any_hit_jit(f3 o, f3 d, float tmin, float tminp, float tmax) {\n";
    s += "    bool occ = false;\n    const f3 inv_d = F3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));\n    (void)inv_d; (void)tmin; (void)tminp;\n";
    for (const int4& run : sruns) if (is_strict_type(run.x)) { s += g_gen_ref ? "    const f3 ds = d;\n" : "    const f3 ds = strict_dir(d);\n"; break; }
    for (const int4& run : sruns) {
        for (int i = 0; i < run.z; ++i) {
            const float4* g = &slots[q];
            if (is_sphere_type(run.x)) {
                s += sphere_root_source(run.x, g[0]) + "\n";
                s += "      bool near_behind = t1 < tmin; float tc = near_behind ? t2 : t1; occ |= (tc >= tmin) & (near_behind | (tc <= tmax)); }\n";
            } else if (is_aaplane_type(run.x)) {
                const char* ax = run.x == PRIM_AAPLANE_X ? "x" : run.x == PRIM_AAPLANE_Y ? "y" : "z";
                s += "    { float t = (" + lit(g[0].x) + " - o." + ax + ") * inv_d." + ax + "; occ |= (t >= tminp) & (t <= tmax); }\n";
            } else if (run.x == PRIM_PLANE) {
                s += "    { float t = (" + lit(g[0].w) + " - " + dot3(g[0], "o") + ") * rcp_fast(" + dot3(g[0], "d") + "); occ |= (t >= tminp) & (t <= tmax); }\n";
            } else if (is_strict_type(run.x)) {
                s += "    { float t; bool hit = " + strict_call(run.x, g) + "; occ |= hit & (t <= tmax); }\n";
            } else {
                const float4 &g1 = g[4], &g2 = g[5], &g3 = g[6];
                s += "    { float t = (" + lit(g[0].w) + " - " + dot3(g[0], "o") + ") * rcp_fast(" + dot3(g[0], "d") + "); f3 p = madd(d, t, o);\n";
                s += "      float x = " + affine3(g1, "p") + ";\n";
                s += "      float y = " + affine3(g2, "p") + ";\n";
                if (run.x == PRIM_RECT) s += "      bool in = (fabsf(x) <= " + lit(g3.x) + ") & (fabsf(y) <= " + lit(g3.y) + ");\n";
                else                    s += "      bool in = (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);\n";
                s += "      occ |= (t >= tminp) & (t <= tmax) & in; }\n";
            }
            q += is_wide_type(run.x) ? 8 : 4;
        }
    }
    s += "    return occ;\n}\n} // namespace rt\n";
    return s;
}


// ---- the Whitted kernels' strict queries (01: every ray; 02: the tree walk) ----------------------------------------------
// strict::closest_hit / strict::any_hit (rt_strict.h) walk the run table and read every primitive through scalar loads from global
// memory: per run a dependent load, per primitive another - on a 784x561 frame (less than one wave per SIMD slot) that latency IS the
// frame time.  Generated: one call of the same strict test per primitive, in scene order, its numbers as literals - the operations
// the reference's x86 build executes, bit for bit (the tests are the inline functions of rt_strict.h, compiled under its
// 'fp contract(off)'; constant folding of literal-only subexpressions rounds like the hardware does).  That this holds was NOT
// free: with -ffp-contract=fast the backend fuses multiply-add pairs whatever the pragma says, and literal operands changed
// which ones it picked - the specialised 01 frame differed from the generic one in 681 pixels until both builds moved to
// -ffp-contract=fast-honor-pragmas (Makefile; options below).  RT_MI355_JIT_NO_STRICT=1 keeps the run-table walk (A/B switch).
std::string f4lit(const float4& g) { return "make_float4(" + lit(g.x) + ", " + lit(g.y) + ", " + lit(g.z) + ", " + lit(g.w) + ")"; }
std::string gen_strict_queries(const std::vector<int4>& wruns, const std::vector<float4>& wgeom) {
    std::string c = "namespace rt {\n__device__ __forceinline__ int closest_hit_strict_jit(f3 o, f3 d, float tmin, float tmax, float& t_out) {\n    int best = -1;\n    float nearest = 3.402823466e+38f;\n";
    std::string a = "__device__ __forceinline__ bool any_hit_strict_jit(f3 o, f3 d, float tmin, float tmax) {\n    bool hit = false;\n";
    for (const int4& run : wruns) {
        const float4* g = wgeom.data() + run.y;
        for (int i = 0; i < run.z; ++i) {
            std::string call, pre;
            if (run.x == PRIM_SPHERE) call = "strict::hit_sphere(" + f4lit(g[i]) + ", o, d, tmin, tmax, t)";
            else if (run.x == PRIM_PLANE) call = "strict::hit_plane(" + v3(g[2 * i]) + ", " + v3(g[2 * i + 1]) + ", o, d, tmin, tmax, t)";
            else if (run.x == PRIM_TRI) {
                pre = "const float4 g4[4] = {" + f4lit(g[4 * i]) + ", " + f4lit(g[4 * i + 1]) + ", " + f4lit(g[4 * i + 2]) + ", " + f4lit(g[4 * i + 3]) + "}; ";
                call = "strict::hit_tri(g4, o, d, tmin, tmax, t)";
            } else continue;
            c += "    { " + pre + "float t; if (" + call + " && t < nearest) { nearest = t; best = " + std::to_string(run.w + i) + "; } }\n";
            a += "    { " + pre + "float t; hit |= " + call + "; }\n";
        }
    }
    c += "    t_out = nearest;\n    return best;\n}\n";
    a += "    return hit;\n}\n} // namespace rt\n";
    return c + a;
}

// ---- generation 2 of the generated scene query ("key" form) ------------------------------------------------------------
// What the first generation spent most of its non-arithmetic instructions on is the bookkeeping per primitive: two float
// compares (t >= tMin, t <= closest; 4.6 issue cycles each on gfx950, twice a VOP2), a scalar AND and two selects.  Here
// the distance is carried as an unsigned KEY = bits(t) - bits(tMin): for t >= tMin (both positive floats) the key is a small
// number that orders like t; for t < tMin, negative t and NaN the subtraction wraps or the sign bit makes the key larger
// than any valid one - so ONE unsigned compare "key <= closest key" is the whole range test, exactly (no rounding: the
// winner's t is key + bits(tMin), bit for bit what the loops of closest_hit_runs compute), and "<=" keeps the reference's
// "last element wins ties".  tMin/tMax are literals (they are part of the generated source, hence of the cache key).
// A sphere's two roots t1 <= t2 become k = min(k1, k2): k1 when t1 >= tMin, else k2 (k1 is then huge) - the reference's
// "a root below tMin is replaced" (Common/Ray.hpp:127-128).  Its rule "origin inside, far root beyond tMax: hit AT tMax"
// (:129-134) needs extra compares and is generated only for a sphere whose ball another primitive reaches into (or whose
// diameter exceeds tMax): only then can something be hit before the ray leaves the sphere.
struct Jit2 { uint32_t C, ck0, vmax; };
uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
std::string hex(uint32_t u) { char b[32]; snprintf(b, sizeof b, "0x%08xU", u); return b; }

// true when, for a ray that starts inside sphere 'self' (slot index q0), nothing else can be hit before it leaves the sphere
bool sphere_interior_is_empty(const std::vector<float4>& slots, const std::vector<int4>& sruns, size_t q0, float tmin, float tmax) {
    const float4 S = slots[q0];
    const double r = std::sqrt((double)S.w), slack = r * 1e-4;
    if (!(2.0 * r + tmin < tmax)) return false;
    size_t q = 0;
    for (const int4& run : sruns) {
        for (int i = 0; i < run.z; ++i) {
            const float4* g = &slots[q];
            const size_t step = is_wide_type(run.x) ? 8 : 4;
            if (q != q0) {
                double dist;      // a lower bound of the distance from the sphere's centre to the other primitive's surface
                if (is_sphere_type(run.x)) {
                    const double dx = g[0].x - S.x, dy = g[0].y - S.y, dz = g[0].z - S.z;
                    dist = std::fabs(std::sqrt(dx * dx + dy * dy + dz * dz) - std::sqrt((double)g[0].w));
                } else if (is_aaplane_type(run.x)) {
                    const double c = run.x == PRIM_AAPLANE_X ? S.x : run.x == PRIM_AAPLANE_Y ? S.y : S.z;
                    dist = std::fabs(c - g[0].x);
                } else if (run.x == PRIM_PLANE) {
                    const double nl = std::sqrt((double)g[0].x * g[0].x + (double)g[0].y * g[0].y + (double)g[0].z * g[0].z);
                    dist = nl > 0 ? std::fabs((double)g[0].x * S.x + (double)g[0].y * S.y + (double)g[0].z * S.z - g[0].w) / nl : 0.0;
                } else {          // rectangle / triangle: its bounding sphere (last float4 of the wide slot)
                    const float4 b = g[7];
                    const double dx = b.x - S.x, dy = b.y - S.y, dz = b.z - S.z;
                    dist = std::isfinite(b.w) ? std::sqrt(dx * dx + dy * dy + dz * dz) - std::sqrt((double)b.w) : 0.0;
                }
                if (!(dist >= r - slack)) return false;
            }
            q += step;
        }
    }
    return true;
}

std::string gen_closest_hit_v2(const std::vector<float4>& slots, const std::vector<int4>& sruns, float tmin, float tmax, bool plain_compare) {
    const float tminp = tmin > 0.0f ? tmin : 1.401298464e-45f;
    Jit2 J{fbits(tminp), fbits(tmax) - fbits(tminp), 0x7f800000U - fbits(tminp)};
    const std::string C = hex(J.C);
    std::string s;
    s += "namespace rt {\n__device__ __forceinline__ uint32_t closest_hit_jit(f3 o, f3 d, float, float, float, float& t_out) {\n";
    s += "    uint32_t best = 0U, ck = " + hex(J.ck0) + ";\n    const f3 inv_d = F3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));\n    (void)inv_d;\n";
    auto take = [&](const std::string& K) { return std::string(plain_compare ? " key_take3" : " key_take") + "(ck, best, k, " + K + "); }\n"; };
    size_t q = 0;
    const char* skip = getenv("RT_MI355_JIT_SKIP");      // timing experiments only: leave a primitive type out of the generated query (wrong images)
    for (const int4& run : sruns) {
        for (int i = 0; i < run.z; ++i) {
            const float4* g = &slots[q];
            const std::string K = std::to_string(q * 16 + SLOT_BYTES) + "U";
            if (skip && ((is_sphere_type(run.x) && strchr(skip, 's')) || (is_aaplane_type(run.x) && strchr(skip, 'a')) || ((run.x == PRIM_RECT || run.x == PRIM_TRI) && strchr(skip, 'r')))) {
                q += is_wide_type(run.x) ? 8 : 4; continue; }
            if (is_sphere_type(run.x)) {
                s += sphere_root_source(run.x, g[0]) + "\n";
                if (sphere_interior_is_empty(slots, sruns, q, tmin, tmax)) {
                    s += "      uint32_t k = __float_as_uint(t1 < " + lit(tmin) + " ? t2 : t1) - " + C + ";\n";
                    s += "     " + take(K);
                    q += 4; continue;
                }
                s += "      uint32_t k1 = __float_as_uint(t1) - " + C + ", k2 = __float_as_uint(t2) - " + C + "; uint32_t k = k1 < k2 ? k1 : k2;\n";
                // the tMax rule: near root behind tMin, far root ahead (finite) but not accepted: the element wins, the distance stays
                    s += "      bool a = k <= ck; bool w = a | ((k1 > " + hex(J.vmax) + ") & (k2 <= " + hex(J.vmax) + ")); ck = a ? k : ck; best = w ? " + K + " : best; }\n";
            } else if (is_aaplane_type(run.x)) {
                const char* ax = run.x == PRIM_AAPLANE_X ? "x" : run.x == PRIM_AAPLANE_Y ? "y" : "z";
                s += "    { uint32_t k = __float_as_uint((" + lit(g[0].x) + " - o." + ax + ") * inv_d." + ax + ") - " + C + ";" + take(K);
            } else if (run.x == PRIM_PLANE) {
                s += "    { uint32_t k = __float_as_uint((" + lit(g[0].w) + " - " + dot3(g[0], "o") + ") * rcp_fast(" + dot3(g[0], "d") + ")) - " + C + ";" + take(K);
            } else {
                const float4 &g1 = g[4], &g2 = g[5], &g3 = g[6];
                s += "    { float t = (" + lit(g[0].w) + " - " + dot3(g[0], "o") + ") * rcp_fast(" + dot3(g[0], "d") + "); f3 p = madd(d, t, o);\n";
                s += "      float x = " + affine3(g1, "p") + ";\n";
                s += "      float y = " + affine3(g2, "p") + ";\n";
                if (run.x == PRIM_RECT) s += "      bool in = (fabsf(x) <= " + lit(g3.x) + ") & (fabsf(y) <= " + lit(g3.y) + ");\n";
                else                    s += "      bool in = (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);\n";
                s += "      uint32_t k = in ? __float_as_uint(t) - " + C + " : 0xffffffffU;" + take(K);
            }
            q += is_wide_type(run.x) ? 8 : 4;
        }
    }
    s += "    t_out = __uint_as_float(ck + " + C + ");\n    return best;\n}\n";
    return s;
}

uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ULL;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ULL; }
    return h;
}

struct Entry { hipModule_t mod = nullptr; hipFunction_t fn = nullptr, fn2 = nullptr; std::string note; uint64_t last_use = 0; };      // fn2: the 02 module's second kernel (stage 2 of the wavefront form)
uint64_t g_tick = 0;                                   // one per lookup (under g_mu): the clock of the least-recently-used replacement
std::mutex g_mu;
std::map<std::pair<int, uint64_t>, Entry> g_cache;   // (device, source hash)

}  // namespace

// When to pay the ~0.7 s compile: a scene that is rendered once at example size would lose by it, a scene that is rendered
// again (progressive / interactive use, benchmarks) or for seconds wins.  So: compile on the SECOND render of the same
// scene in this process, or at once when the job is large or the caller forces it.  At most RT_JIT_MAX_SCENES compiled scenes are kept per process
// (RT_MI355_JIT_MAX_SCENES=<n> overrides, 1..4096).  When the table is full the least recently used kernel of this device makes room - but only if it has not been
// used for 4 x the capacity in lookups: a long-lived process that moves on to new scenes keeps specialising, a process that cycles through more scenes than fit
// does not recompile on every render (0.7 s each) but runs the newcomers on the generic kernels, as before.
constexpr size_t RT_JIT_MAX_SCENES = 256;      // ~100 KB of code object each
size_t jit_capacity() {
    static const size_t n = [] { const char* e = getenv("RT_MI355_JIT_MAX_SCENES"); const long v = e ? atol(e) : 0; return v >= 1 && v <= 4096 ? (size_t)v : RT_JIT_MAX_SCENES; }();
    return n;
}
constexpr double RT_JIT_BIG_JOB_SAMPLES = 5e10;
std::map<uint64_t, int> g_sightings;

bool jit_wanted(int variant, const std::vector<float4>& slots, const std::vector<int4>& sruns, double job_samples, bool force, std::string& note) {
    uint64_t h = 1469598103934665603ULL ^ (uint64_t)variant;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; } };
    mix(slots.data(), slots.size() * sizeof(float4)); mix(sruns.data(), sruns.size() * sizeof(int4));
    std::lock_guard<std::mutex> lock(g_mu);
    int seen = g_sightings.size() < 4096 || g_sightings.count(h) ? ++g_sightings[h] : 1;
    if (force || seen >= 2 || job_samples >= RT_JIT_BIG_JOB_SAMPLES) return true;
    note = "first render of this scene and a small job: generic kernel (specialised from the second render on)";
    return false;
}

hipFunction_t jit_path_kernel(int variant, const std::vector<float4>& slots, const std::vector<int4>& sruns, float tmin, float tmax, int form, std::string& note, hipFunction_t* second,
                              const std::vector<int4>* wruns, const std::vector<float4>* wgeom) {
    if (second) *second = nullptr;
    const bool ref = (form & RT_FORM_REF) != 0, norm = (form & RT_FORM_NORM) != 0, point = (form & RT_FORM_POINT) != 0;
    form &= 7;
    if (ref && form != RT_FORM_REGEN) { note = "reference arithmetic: per-lane regeneration only"; return nullptr; }
    const bool whitted = variant == RT_JIT_WHITTED_SOFT || variant == RT_JIT_WHITTED_HARD;
    size_t n_prims = 0;
    for (const int4& r : sruns) n_prims += (size_t)r.z;
    if (whitted && wruns) { size_t n = 0; for (const int4& r : *wruns) n += (size_t)r.z; n_prims = n > n_prims ? n : n_prims; }
    if (variant == RT_JIT_WHITTED_HARD && !(wruns && wgeom)) { note = "01 kernel: no strict layout given"; return nullptr; }
    if (n_prims == 0 || n_prims > RT_JIT_MAX_PRIMS) { note = "scene has " + std::to_string(n_prims) + " slots (specialised up to " + std::to_string(RT_JIT_MAX_PRIMS) + ")"; return nullptr; }
    std::string defs;      // RT_MI355_JIT_DEFS=NAME[,NAME...]: extra #defines for the generated source (A/B switches of the device code)
    if (const char* d = getenv("RT_MI355_JIT_DEFS")) {
        std::string cur;
        for (const char* p = d;; ++p) {
            if (*p == ',' || *p == 0) { if (!cur.empty()) defs += "#define " + cur + " 1\n"; cur.clear(); if (!*p) break; }
            else cur += *p;
        }
    }
    if (ref) defs += "#define RT_REF_RAYS true\n";
    if (!whitted) defs += norm ? "#define RT_JIT_NORM_DIRS 1\n" : "#define RT_JIT_NORM_DIRS 0\n";
    if (!whitted) defs += point ? "#define RT_JIT_STRICT_POINT 1\n" : "#define RT_JIT_STRICT_POINT 0\n";
    g_gen_ref = ref;
    std::string src = defs + std::string("#define RT_JIT_SCENE 1\ntypedef unsigned int uint32_t; typedef unsigned char uint8_t; typedef unsigned long long uint64_t; typedef unsigned long uintptr_t;\n") + kSrcArgs + "\n" + kSrcDevice + "\n";
    // generation 1 (float compares) is the default: on MI355X it measured 2 % faster than generation 2 (unsigned keys) although
    // the latter has the cheaper instruction mix on paper; RT_MI355_JIT_GEN=2 selects generation 2 (same image, bit for bit)
    const char* gen_env = getenv("RT_MI355_JIT_GEN");
    bool has_strict = false;      // slots in the reference's operation sequence (rt_device.h PRIM_*_STRICT): generation 1 only
    for (const int4& r : sruns) has_strict = has_strict || is_strict_type(r.x);
    const bool gen2 = !whitted && gen_env && (gen_env[0] == '2' || gen_env[0] == '3') && tmin >= 0.0f && tmax > tmin && std::isfinite(tmax) && !has_strict;
    src += (gen2 ? gen_closest_hit_v2(slots, sruns, tmin, tmax, gen_env[0] == '3') : gen_closest_hit(slots, sruns)) + gen_any_hit(slots, sruns);
    const char* entry = whitted ? "k_whitted_jit" : "k_path_jit";
    std::string strict_gen;
    if (whitted && wruns && wgeom && !getenv("RT_MI355_JIT_NO_STRICT")) {
        strict_gen += "#define RT_JIT_STRICT_SCENE 1\n";
        strict_gen += gen_strict_queries(*wruns, *wgeom);
    }
    if (variant == RT_JIT_WHITTED_HARD) {   // 01: whitted_body<false>; registers as needed (no scratch), like the ahead-of-time k_whitted_hard
        src += std::string(kSrcStrict) + "\n" + strict_gen + kSrcWhitted +
            "\nextern \"C\" __global__ __launch_bounds__(WBLOCK) void k_whitted_jit(rt::DevScene SD, rt::WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {\n"
            "    rt::whitted_body<false>(SD, A, out, counters);\n}\n";
    }
    else if (variant == RT_JIT_WHITTED_SOFT) {   // 02: the generated any-hit query serves the shadow samples - of the monolithic kernel (k_whitted_jit) and of stage 2 of the wavefront form (k_whitted_shadow_jit)
        // waves per SIMD of the monolithic kernel: 7 (72 VGPRs) needs no scratch; 8 (64 VGPRs) spills a few cold values (RT_MI355_WHITTED_WAVES, measured: DESIGN.md 3)
        const char* wv = getenv("RT_MI355_WHITTED_WAVES");
        const std::string waves = (wv && wv[0] >= '4' && wv[0] <= '8' && !wv[1]) ? std::string(wv) : "7";
        src += std::string(kSrcStrict) + "\n" + strict_gen + kSrcWhitted +
            "\nextern \"C\" __global__ __launch_bounds__(WBLOCK) __attribute__((amdgpu_waves_per_eu(" + waves + ", 8))) void k_whitted_jit(rt::DevScene SD, rt::WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {\n"
            "    rt::whitted_body<true>(SD, A, out, counters);\n}\n"
            "\nextern \"C\" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_whitted_shadow_jit(rt::DevScene SD, rt::WhittedArgs A) {\n"
            "    rt::whitted_shadow_body<true>(SD, A);\n}\n";
    }
    else
        src += std::string(kSrcPath) +
            "\nextern \"C\" __global__ __launch_bounds__(256) void k_path_jit(rt::DevScene SD, rt::PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {\n"
            "    rt::path_" + std::string(form == RT_FORM_QUEUE ? "queue_" : form == RT_FORM_ITEMS ? "items_" : form == RT_FORM_SORTED ? "sorted_" : "") + "body<" + std::to_string(variant) + ", true>(SD, A, out, counters);\n}\n";
    int dev = 0; (void)hipGetDevice(&dev);
    const auto key = std::make_pair(dev, fnv1a(src));
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find(key);
    ++g_tick;
    if (it != g_cache.end()) { it->second.last_use = g_tick; note = it->second.note; if (second) *second = it->second.fn2; return it->second.fn; }
    if (g_cache.size() >= jit_capacity()) {
        auto lru = g_cache.end();
        for (auto e = g_cache.begin(); e != g_cache.end(); ++e)
            if (e->first.first == dev && (lru == g_cache.end() || e->second.last_use < lru->second.last_use)) lru = e;
        if (lru == g_cache.end() || g_tick - lru->second.last_use < 4 * jit_capacity()) {
            note = "more than " + std::to_string(jit_capacity()) + " distinct scenes in use in this process: generic kernel"; return nullptr; }
        // nothing of that module can still be running once the device is idle; its last launch was >= 4 x capacity renders ago (a caller launches what it was handed
        // before it asks again, and any other caller would have to fit that many lookups - compiles among them - between being handed the kernel and launching it)
        if (lru->second.mod) { (void)hipDeviceSynchronize(); (void)hipModuleUnload(lru->second.mod); }
        g_cache.erase(lru);
    }
    Entry e;
    // RT_MI355_JIT_CACHE=<directory>: compiled code objects are kept there across processes, keyed by the hash of the generated source and the
    // hiprtc version (a serving process or a render farm pays the ~0.7 s compile once per scene and compiler, not once per process).  Off unless set:
    // the library writes nowhere it was not pointed at.  A file that fails to load is ignored and rewritten.
    std::string cache_file;
    if (const char* dir = getenv("RT_MI355_JIT_CACHE")) {
        int maj = 0, min = 0; (void)hiprtcVersion(&maj, &min);
        char name[96]; snprintf(name, sizeof name, "/rt_mi355_%016llx_hiprtc%d.%d_gfx950.co", (unsigned long long)key.second, maj, min);
        if (dir[0] && !getenv("RT_MI355_JIT_FLAGS")) cache_file = std::string(dir) + name;
    }
    bool loaded = false;
    if (!cache_file.empty()) {
        if (FILE* f = fopen(cache_file.c_str(), "rb")) {
            std::vector<char> code; char buf[65536]; size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + n);
            fclose(f);
            if (!code.empty() && hipModuleLoadData(&e.mod, code.data()) == hipSuccess && hipModuleGetFunction(&e.fn, e.mod, entry) == hipSuccess) {
                loaded = true;
                if (variant == RT_JIT_WHITTED_SOFT && hipModuleGetFunction(&e.fn2, e.mod, "k_whitted_shadow_jit") != hipSuccess) e.fn2 = nullptr;
            } else { (void)hipGetLastError(); e = Entry(); }
        }
    }
    hiprtcProgram prog = nullptr;
    if (loaded) { /* nothing to compile */ }
    else if (hiprtcCreateProgram(&prog, src.c_str(), "rt_path_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) e.note = "hiprtcCreateProgram failed";
    else {
        // same code generation as the Makefile's: contraction on except where a pragma turns it off (rt_strict.h; '=fast' would let the backend fuse there too), '/' and sqrtf correctly rounded
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast-honor-pragmas", "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt"};
        std::vector<const char*> ov(opts, opts + sizeof opts / sizeof opts[0]);
        std::vector<std::string> extra;      // RT_MI355_JIT_FLAGS="-mllvm;-amdgpu-sched-strategy=max-ilp": extra compiler options (experiments; ';'-separated)
        if (const char* xf = getenv("RT_MI355_JIT_FLAGS")) { std::string cur; for (const char* q = xf;; ++q) { if (*q == ';' || *q == 0) { if (!cur.empty()) extra.push_back(cur); cur.clear(); if (!*q) break; } else cur += *q; } }
        for (const std::string& x : extra) ov.push_back(x.c_str());
        hiprtcResult r = hiprtcCompileProgram(prog, (int)ov.size(), ov.data());
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
            std::string log(n, '\0'); if (n) hiprtcGetProgramLog(prog, &log[0]);
            e.note = "hiprtc compile failed: " + log.substr(0, 2000);
        } else {
            size_t n = 0; hiprtcGetCodeSize(prog, &n);
            std::vector<char> code(n); hiprtcGetCode(prog, code.data());
            if (const char* dump = getenv("RT_MI355_JIT_DUMP")) {   // <prefix>.hip = generated source, <prefix>.co = code object (llvm-objdump -d)
                if (FILE* f = fopen((std::string(dump) + ".hip").c_str(), "w")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
                if (FILE* f = fopen((std::string(dump) + ".co").c_str(), "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
            }
            if (!cache_file.empty()) {      // written under a temporary name and renamed: a reader never sees half a file
                const std::string tmp = cache_file + ".tmp" + std::to_string((long long)getpid());
                if (FILE* f = fopen(tmp.c_str(), "wb")) {
                    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
                    fclose(f);
                    if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) (void)remove(tmp.c_str());
                }
            }
            if (hipModuleLoadData(&e.mod, code.data()) != hipSuccess || hipModuleGetFunction(&e.fn, e.mod, entry) != hipSuccess) { e.fn = nullptr; e.note = "hipModuleLoadData/GetFunction failed"; }
            else if (variant == RT_JIT_WHITTED_SOFT && hipModuleGetFunction(&e.fn2, e.mod, "k_whitted_shadow_jit") != hipSuccess) e.fn2 = nullptr;
        }
        hiprtcDestroyProgram(&prog);
    }
    e.last_use = g_tick;
    g_cache[key] = e;
    note = e.note;
    if (second) *second = e.fn2;
    return e.fn;
}

} // namespace rt
