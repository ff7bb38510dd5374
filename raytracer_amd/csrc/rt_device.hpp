// rt_device.hpp -- device-side data layout and scalar arithmetic of librt355.
//
// Every expression that feeds a hit/miss decision is written with an explicit
// evaluation order and the library is built with -ffp-contract=off, so a
// multiply-add always rounds twice (the reference is plain Rust f32
// arithmetic, e.g. geometry/src/bounds3.rs:212-238).  Division and sqrt are
// the correctly rounded forms hipcc emits by default.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt355.h"

namespace rt {

// ---- resident scene (device pointers), passed to kernels by value --------------
struct LeafPrim {      // one entry per primitive, in leaf order (BVH::primitives, bvh.rs:35)
    uint32_t kind;     // rt_prim_kind | reverse << 8 | kLastInLeaf | kRareShade | xform << 16
    uint32_t index;    // index into the typed table (tables are packed in leaf order); see kMediumInSphere
    uint32_t prim;     // caller's primitive index
    uint32_t material;
    float4 geom;       // sphere: {c.xyz, r}; medium in a sphere: its boundary {c.xyz, r}; else unused
};

constexpr uint32_t kLastInLeaf = 0x200u; // LeafPrim.kind: no further primitive in this leaf
// LeafPrim.kind: a hit on this primitive needs the general shader (anything but a sphere,
// triangle, rectangle, box or medium without instance transform, carrying a Lambertian,
// metallic, emissive or isotropic material whose texture is a solid colour or a 3-D checker
// of solid colours)
constexpr uint32_t kRareShade = 0x400u;
// LeafPrim.kind of a medium bounded by a sphere: geom is the sphere and index holds the bits
// of -1 / density instead of a row of SceneDev::media
constexpr uint32_t kMediumInSphere = 0x800u;
// LeafPrim.kind: the primitive is one of the lights RT_RENDER_DIRECT_LIGHT samples (SceneDev::lights):
// a path that sampled the lights at its previous vertex must not add this one's emission again
constexpr uint32_t kSampledLight = 0x1000u;
constexpr uint32_t kLeafRef = 0x80000000u; // child reference: leaf starting at primitive (ref & ~kLeafRef)

struct GSphere { // derived form of rt_gsphere, 128 B
    float o2w[12];
    float w2o[12];
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    uint32_t flip, _pad;
};

struct MSphere { // 32 B
    float c0[3], r, dcdt[3], time0;
};

struct Medium { // 16 B
    uint32_t boundary_kind, boundary_index, xform;
    float neg_inv_density; // -1.0f / density, divided on the host
};

// A light of RT_RENDER_DIRECT_LIGHT: a rectangle without instance transform whose material is
// a DiffuseLight with a solid colour, scale = area * n_lights / pi (fp32, in that order); or
// (axis 3) a distant light: unit direction towards it in a0, a1, b0, radiance in rgb, k = the
// distance along that direction of the point its shadow ray is aimed at, scale = n_lights / pi.
struct LightDev { // 48 B
    uint32_t axis;
    float k, a0, a1, b0, b1;
    float scale;
    float rgb[3];
    uint32_t _pad[2];
};

// What the plain shader needs of a material and its texture, one record per material, so that a
// shading batch reads it in one load beside the geometry's instead of following material ->
// texture (-> checker cells) through three dependent loads.  Meant for the hits of leaf entries
// without kRareShade (a per-primitive table of the same records made the large scenes slower: a
// 48-byte gather from a table as large as the primitives').
struct PlainShade { // 48 B
    uint32_t material_kind; // rt_material_kind
    float fuzz, ior;
    uint32_t checker;       // 0: solid colour a; 1: 3-D checker, a where the sines' product is negative, else b
    float a[3], scale;
    float b[3], _pad;
};

// A ball: a medium bounded by a sphere that no other primitive of the tree reaches into, with or
// without a sphere primitive of exactly that geometry around it (a glass ball filled with fog).
// A ray that starts inside meets everything else only after it has left the ball, so the ball's
// own primitives can decide its closest hit without a walk through the tree (ball_resolve).
constexpr uint32_t kMaxBalls = 4;
struct BallDev { // 40 B
    float c[3], r;
    float r_in2;      // a ray starts inside when |o - c|^2 < r_in2
    uint32_t n;       // primitives of the ball: 1 (the medium) or 2
    uint32_t best[2]; // their leaf slots in leaf order, with kBestRare where a hit needs the general shader
    int32_t sphere;   // which of them is the sphere primitive, or -1
    uint32_t medium;  // which is the medium
};

struct ImageDev {
    const uint8_t *rgb8;
    uint32_t w, h;
};

struct SceneDev {
    // One 64-byte record per interior node holding the boxes of both children
    // (LinearBVHNode.bounds of node + 1 and of node child_offset, bvh.rs:83-89):
    //   {lo0.xyz, ref0} {hi0.xyz, ref1} {lo1.xyz, split axis} {hi1.xyz, -}
    // refN = index of the child's own record, or kLeafRef | first primitive of a leaf child.
    const float4 *nodes;
    // Trees in global memory also come as one 128-byte record per interior node at an even level
    // below the root, holding the boxes of its (up to four) grandchildren:
    //   {lo.xyz of entry 0, ref 0} {hi 0, ref 1} {lo 1, ref 2} {hi 1, ref 3} {lo 2, axes} {hi 2, -} {lo 3, -} {hi 3, -}
    // entries 0, 1 = the children of the node's first child, 2, 3 = those of its second child; a
    // child that is a leaf stands as the first entry of its side and the second is empty (ref
    // 0xFFFFFFFF).  ref = index of the entry's own record here, or kLeafRef | first primitive.
    // axes: bits 0-1 the node's split axis, 2-3 its first child's, 4-5 its second child's.
    // Null when the tree is in LDS or has no such records.
    const float4 *quads;
    uint32_t root_quad; // reference to the root in `quads` terms
    const LeafPrim *leaf;     // n_prims
    const float4 *spheres;    // {c.xyz, r}
    const MSphere *msph;
    const float4 *tris;       // 3 x float4 per triangle {v.xyz, -}
    const float *tri_uvs;     // 6 per triangle or null
    const float *tri_normals; // 9 per triangle or null
    const rt_rect *rects;
    const rt_box *boxes;
    const GSphere *gsph;
    const rt_xform *xforms;
    const Medium *media;
    const rt_material *materials;
    const rt_texture *textures;
    const PlainShade *plain;  // one per material, or (plain_per_entry) one per leaf entry
    const ImageDev *images;
    const float *perlin_vec;
    const int32_t *perlin_perm;
    const LightDev *lights;
    uint32_t n_lights;
    uint32_t plain_per_entry; // `plain` is indexed by leaf slot instead of by material
    uint32_t n_nodes, n_prims; // n_nodes: LinearBVHNode count of the tree (0 = empty tree)
    // Media that enclose the scene are not in the tree: their entries follow the tree's
    // n_tree_prims primitives in `leaf`, and every ray tests them before it starts traversing.
    uint32_t n_tree_prims, n_enclosing;
    uint32_t enclosing_rare; // bit e: a scatter in enclosing medium e needs the general shader
    uint32_t n_balls;
    BallDev balls[kMaxBalls];
    uint32_t n_pairs;          // interior records in `nodes`
    uint32_t root_ref;         // reference to the root (a leaf when the tree is a single node)
    uint32_t stack_lds;        // stack levels per lane kept in LDS; deeper entries use private memory
    float root_lo[3], root_hi[3];
    float background[3];
};

struct v3 {
    float x, y, z;
};

#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__

// A table entry read through the constant address space: the scene's tables do not change while a
// kernel runs, and saying so lets a read whose address is the same in every lane become one scalar
// load (no vector-memory round trip, no vector registers) instead of 64 identical gathers.
template <class T>
RT_DEV T load_const(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
#ifdef RT_NO_CONST_LOADS
    return *p;
#endif
    typedef const __attribute__((address_space(4))) uint32_t *WordPtr;
    const WordPtr q = (WordPtr)(const void *)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = q[i];
    T v;
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}

RT_DEV v3 mk(float x, float y, float z) { return v3{x, y, z}; }
RT_DEV v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
// Div<T> for Vector3 / Point3 (vector3.rs:260-264, point3.rs:271-275): one reciprocal, three products
RT_DEV v3 operator/(v3 a, float s) {
    const float inv = 1.0f / s;
    return mk(a.x * inv, a.y * inv, a.z * inv);
}
RT_DEV v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV v3 vabs(v3 a) { return mk(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
RT_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV v3 cross(v3 a, v3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV float length(v3 a) { return sqrtf(dot(a, a)); }
RT_DEV v3 normalize(v3 a) { return a / length(a); } // vector3.rs:140-142
RT_DEV float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
RT_DEV v3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

RT_DEV float sel_max(float a, float b) { return a > b ? a : b; } // number.rs:121-128
RT_DEV float sel_min(float a, float b) { return a < b ? a : b; } // number.rs:131-138

// gamma(n) with eps = f32::EPSILON, geometry/src/float.rs:44-46, number.rs:205
constexpr float kEps = 1.1920928955078125e-7f;
constexpr float kOneMinusEps = 1.0f - kEps; // ONE_MINUS_EPSILON, geometry/src/number.rs:209
__host__ __device__ constexpr float gamma_n(int n) {
    return ((float)n * kEps) / (1.0f - (float)n * kEps);
}
constexpr float G2 = gamma_n(2), G3 = gamma_n(3), G5 = gamma_n(5), G7 = gamma_n(7);
constexpr float kTwoPi = 6.28318548f, kPi = 3.14159274f;
constexpr float kInf = __builtin_huge_valf();

// next_float_up / next_float_down, geometry/src/float.rs:8-41 (positive branch
// of the downward step decrements)
RT_DEV float next_float_up(float v) {
    if (v == kInf) return v;
    if (v == 0.0f) v = 0.0f;
    uint32_t b = __float_as_uint(v);
    if (v >= 0.0f) b += 1; else b -= 1;
    return __uint_as_float(b);
}
RT_DEV float next_float_down(float v) {
    if (v == -kInf) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t b = __float_as_uint(v);
    if (v > 0.0f) b -= 1; else b += 1;
    return __uint_as_float(b);
}

// ---- transcendental functions as fixed polynomials (+ - * / only) --------------
RT_HD float p_sin(float x) {
    float k = floorf(x * 0.318309886f + 0.5f);
    float r = x - k * 3.140625f;
    r = r - k * 9.67502593994140625e-4f;
    r = r - k * 1.509957990978376432e-7f;
    float r2 = r * r;
    float p = -2.5052108e-8f;
    p = p * r2 + 2.7557319e-6f;
    p = p * r2 + -1.9841270e-4f;
    p = p * r2 + 8.3333333e-3f;
    p = p * r2 + -1.6666667e-1f;
    float s = r + r * r2 * p;
    int ki = (int)k;
    return (ki & 1) ? -s : s;
}
RT_HD float p_cos(float x) { return p_sin(x + 1.57079637f); }

RT_HD float p_log(float x) {
    uint32_t ix = __builtin_bit_cast(uint32_t, x);
    int e = (int)(ix >> 23) - 127;
    float f = __builtin_bit_cast(float, (ix & 0x007fffffu) | 0x3f800000u);
    if (f > 1.41421356f) { f = f * 0.5f; e += 1; }
    float s = (f - 1.0f) / (f + 1.0f);
    float z = s * s;
    float p = 0.181818182f;
    p = p * z + 0.222222222f;
    p = p * z + 0.285714286f;
    p = p * z + 0.4f;
    p = p * z + 0.666666667f;
    p = p * z + 2.0f;
    float lf = s * p;
    float fe = (float)e;
    return fe * 0.693359375f + (lf + fe * -2.12194440e-4f);
}

RT_HD float p_atan2(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;
    float base = 0.0f;
    if (a > 0.414213562f) { a = (a - 1.0f) / (a + 1.0f); base = 0.785398163f; }
    float z = a * a;
    float p = -0.0666666667f;
    p = p * z + 0.0769230769f;
    p = p * z + -0.0909090909f;
    p = p * z + 0.111111111f;
    p = p * z + -0.142857143f;
    p = p * z + 0.2f;
    p = p * z + -0.333333333f;
    p = p * z + 1.0f;
    float r = base + a * p;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    if (y < 0.0f) r = -r;
    return r;
}
RT_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
RT_HD float p_acos(float x) { return p_atan2(sqrtf((1.0f - x) * (1.0f + x)), x); }

// ---- PCG32, geometry/src/rng.rs:5-67 ----------------------------------------------
struct Rng {
    uint64_t state, inc;
};
RT_DEV uint32_t rng_u32(Rng &r) {
    uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t x = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (x >> rot) | (x << ((~rot + 1u) & 31u));
}
RT_DEV void rng_seq(Rng &r, uint64_t index) {
    r.state = 0;
    r.inc = (index << 1) | 1u;
    (void)rng_u32(r);
    r.state += 0x853c49e6748fea9bULL;
    (void)rng_u32(r);
}
RT_DEV float rng_f(Rng &r) { // rng.rs:63-67
    float num = (float)rng_u32(r) * 2.3283064365386963e-10f;
    float lim = 1.0f - kEps;
    return lim < num ? lim : num;
}
RT_DEV uint32_t hash32(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

// ---- affine transforms, geometry/src/transform.rs ---------------------------------
RT_DEV v3 xf_point(const float *m, v3 p) { // :335-341
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
RT_DEV v3 xf_vector(const float *m, v3 p) { // :435-442
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
              m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
RT_DEV v3 xf_normal(const float *inv, v3 n) { // :504-516
    return mk(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z,
              inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}
RT_DEV v3 xf_abs_sum(const float *m, v3 p) { // :372-376
    return mk(fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]),
              fabsf(m[4] * p.x) + fabsf(m[5] * p.y) + fabsf(m[6] * p.z) + fabsf(m[7]),
              fabsf(m[8] * p.x) + fabsf(m[9] * p.y) + fabsf(m[10] * p.z) + fabsf(m[11]));
}
// :390-429, incoming error scaled by |M| without the translation column
RT_DEV v3 xf_point_err(const float *m, v3 p, v3 e, v3 &err) {
    v3 es = mk(fabsf(m[0] * e.x) + fabsf(m[1] * e.y) + fabsf(m[2] * e.z),
               fabsf(m[4] * e.x) + fabsf(m[5] * e.y) + fabsf(m[6] * e.z),
               fabsf(m[8] * e.x) + fabsf(m[9] * e.y) + fabsf(m[10] * e.z));
    v3 as = xf_abs_sum(m, p);
    err = es * (G3 + 1.0f) + as * G3;
    return xf_point(m, p);
}

// ---- ray / hit ------------------------------------------------------------------------
struct Ray {
    v3 o, d;
    float t_max, time;
};

struct Surf { // accepted hit with everything shading and the next ray need
    float t, u, v;
    v3 n, p, perr;
    uint32_t flags;
};

// ---- quadratic + root choice: float.rs:182-204 and quadric.rs:24-35, plain fp32 -------
RT_DEV bool quadratic(float a, float b, float c, float &t0, float &t1) {
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    float root = sqrtf(disc);
    float q = (b < 0.0f) ? -0.5f * (b - root) : -0.5f * (b + root);
    float r0 = q / a;
    float r1 = c / q;
    if (r0 > r1) { float tmp = r0; r0 = r1; r1 = tmp; }
    t0 = r0;
    t1 = r1;
    return true;
}
RT_DEV bool sphere_roots(v3 o, v3 d, float r, float &t0, float &t1) {
    float a = dot(d, d);
    float b = 2.0f * dot(d, o);
    float c = dot(o, o) - r * r;
    return quadratic(a, b, c, t0, t1);
}
RT_DEV bool sphere_pick(float t0, float t1, float t_max, float &t) {
    if (t0 > t_max || t1 <= 0.0f) return false;
    float ts = t0;
    if (t0 <= 0.0f) {
        if (t1 > t_max) return false;
        ts = t1;
    }
    if (!(ts > 0.0f && ts <= t_max)) return false;
    t = ts;
    return true;
}
RT_DEV v3 msphere_centre(const MSphere &m, float time) {
    float dt = time - m.time0;
    return mk(m.c0[0] + dt * m.dcdt[0], m.c0[1] + dt * m.dcdt[1], m.c0[2] + dt * m.dcdt[2]);
}

// Entry and exit of a ray through an axis-aligned box with the faces they lie on.  An axis
// the ray runs along (d == 0) constrains nothing if the origin is between its planes and
// rules the box out otherwise.  Written with selects: the three axes run without branches.
// inv: 1 / d per axis.
RT_DEV bool box_interval_inv(v3 o, v3 d, v3 inv3, const rt_box &b, float &tn_o, float &tf_o, int &fn_o, int &ff_o) {
    float tn = -kInf, tf = kInf;
    int fn = 0, ff = 0;
    bool out = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float oi = comp(o, i), di = comp(d, i), inv = comp(inv3, i);
        const bool along = di == 0.0f;
        out = out || (along && (oi < b.lo[i] || oi > b.hi[i]));
        float t0 = (b.lo[i] - oi) * inv;
        float t1 = (b.hi[i] - oi) * inv;
        int f0 = 2 * i, f1 = 2 * i + 1;
        if (inv < 0.0f) {
            float tt = t0; t0 = t1; t1 = tt;
            int ft = f0; f0 = f1; f1 = ft;
        }
        if (!along && t0 > tn) { tn = t0; fn = f0; }
        if (!along && t1 < tf) { tf = t1; ff = f1; }
    }
    if (out || tn > tf) return false;
    tn_o = tn; tf_o = tf; fn_o = fn; ff_o = ff;
    return true;
}
RT_DEV bool box_interval(v3 o, v3 d, const rt_box &b, float &tn_o, float &tf_o, int &fn_o, int &ff_o) {
    return box_interval_inv(o, d, mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), b, tn_o, tf_o, fn_o, ff_o);
}
// Same for a ray whose reciprocal direction rd = 1.0f / d is already at hand (the traversal
// loop computes exactly that value for its slab tests): three divisions fewer.
RT_DEV bool box_interval_rd(v3 o, v3 d, v3 rd, const rt_box &b, float &tn_o, float &tf_o, int &fn_o, int &ff_o) {
    return box_interval_inv(o, d, rd, b, tn_o, tf_o, fn_o, ff_o);
}

// rt_box as two 16-byte loads (the per-axis reads of the interval tests would otherwise
// become six conditional 4-byte loads)
RT_DEV rt_box load_box(const rt_box *boxes, uint32_t i) {
    const float4 *p = reinterpret_cast<const float4 *>(boxes + i);
    const float4 a = p[0], b = p[1];
    rt_box r;
    r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z; r._p0 = 0.0f;
    r.hi[0] = b.x; r.hi[1] = b.y; r.hi[2] = b.z; r._p1 = 0.0f;
    return r;
}

RT_DEV void rect_axes(uint32_t axis, int &A, int &B) {
    if (axis == 0) { A = 1; B = 2; }
    else if (axis == 1) { A = 0; B = 2; }
    else { A = 0; B = 1; }
}

RT_DEV void to_object(const SceneDev &S, uint32_t xform, const Ray &ray, v3 &o, v3 &d) {
    o = ray.o;
    d = ray.d;
    if (xform) {
        const float *m = S.xforms[xform - 1].w2o;
        o = xf_point(m, ray.o);
        d = xf_vector(m, ray.d);
    }
}

// ---- reference Sphere (sphere.rs:66-157) ----------------------------------------------
RT_DEV void gsphere_ray(const GSphere &g, const Ray &ray, v3 &o, v3 &d) { // transform.rs:567-587
    v3 oo = xf_point(g.w2o, ray.o);
    v3 oerr = xf_abs_sum(g.w2o, ray.o) * G3;
    v3 dd = xf_vector(g.w2o, ray.d);
    float len_sq = dot(dd, dd);
    if (len_sq > 0.0f) {
        float dt = dot(vabs(dd), oerr) / len_sq;
        oo = oo + dd * dt;
    }
    o = oo;
    d = dd;
}
RT_DEV v3 gsphere_point(const GSphere &g, v3 o, v3 d, float t, float &phi) { // sphere.rs:70-79
    v3 p = o + d * t;
    p = p * (g.radius / length(p));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1e-5f * g.radius;
    float ph = p_atan2(p.y, p.x);
    if (ph < 0.0f) ph += kTwoPi;
    phi = ph;
    return p;
}
RT_DEV bool gsphere_clipped(const GSphere &g, v3 p, float phi) { // sphere.rs:82-84
    return (g.z_min > -g.radius && p.z < g.z_min) || (g.z_max < g.radius && p.z > g.z_max) ||
           phi > g.phi_max;
}
RT_DEV bool gsphere_test(const GSphere &g, const Ray &ray, float &t_out) {
    v3 o, d;
    gsphere_ray(g, ray, o, d);
    float t0, t1, ts, phi;
    if (!sphere_roots(o, d, g.radius, t0, t1)) return false;
    if (!sphere_pick(t0, t1, ray.t_max, ts)) return false;
    v3 p = gsphere_point(g, o, d, ts, phi);
    if (gsphere_clipped(g, p, phi)) { // sphere.rs:85-108
        if (ts == t1 || t1 > ray.t_max) return false;
        ts = t1;
        if (!(ts > 0.0f)) return false;
        p = gsphere_point(g, o, d, ts, phi);
        if (gsphere_clipped(g, p, phi)) return false;
    }
    t_out = ts;
    return true;
}

// ---- watertight triangle, shapes/src/triangle.rs:108-202 ------------------------------
// (axis permutation wraps with (k+1) % 3; max_dimension keeps the last maximal index)
RT_DEV bool tri_test(v3 v0, v3 v1, v3 v2, const Ray &ray, float &t_out, float &b0_o, float &b1_o,
                     float &b2_o) {
    v3 p0 = v0 - ray.o, p1 = v1 - ray.o, p2 = v2 - ray.o;
    v3 ad = vabs(ray.d);
    int kz = 0;
    float best = ad.x;
    if (ad.y >= best) { kz = 1; best = ad.y; }
    if (ad.z >= best) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    v3 dp = mk(comp(ray.d, kx), comp(ray.d, ky), comp(ray.d, kz));
    v3 a = mk(comp(p0, kx), comp(p0, ky), comp(p0, kz));
    v3 b = mk(comp(p1, kx), comp(p1, ky), comp(p1, kz));
    v3 c = mk(comp(p2, kx), comp(p2, ky), comp(p2, kz));
    float sx = -dp.x / dp.z, sy = -dp.y / dp.z, sz = 1.0f / dp.z;
    a.x += sx * a.z; a.y += sy * a.z;
    b.x += sx * b.z; b.y += sy * b.z;
    c.x += sx * c.z; c.y += sy * c.z;
    float e0 = b.x * c.y - b.y * c.x;
    float e1 = c.x * a.y - c.y * a.x;
    float e2 = a.x * b.y - a.y * b.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { // triangle.rs:145-155
        double ax = a.x, ay = a.y, bx = b.x, by = b.y, cx = c.x, cy = c.y;
        e0 = (float)(bx * cy - by * cx);
        e1 = (float)(cx * ay - cy * ax);
        e2 = (float)(ax * by - ay * bx);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    a.z *= sz; b.z *= sz; c.z *= sz;
    float scaled = e0 * a.z + e1 * b.z + e2 * c.z;
    if (det < 0.0f && (scaled >= 0.0f || scaled < ray.t_max * det)) return false;
    if (det > 0.0f && (scaled <= 0.0f || scaled > ray.t_max * det)) return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = scaled * inv_det;
    float max_zt = sel_max(sel_max(fabsf(a.z), fabsf(b.z)), fabsf(c.z)); // triangle.rs:186-202
    float delta_z = max_zt * G3;
    float max_xt = sel_max(sel_max(fabsf(a.x), fabsf(b.x)), fabsf(c.x));
    float max_yt = sel_max(sel_max(fabsf(a.y), fabsf(b.y)), fabsf(c.y));
    float delta_x = (max_xt + max_zt) * G5;
    float delta_y = (max_yt + max_zt) * G5;
    float delta_e = 2.0f * (max_xt * max_yt * G2 + delta_y * max_xt + delta_x * max_yt);
    float max_e = sel_max(sel_max(fabsf(e0), fabsf(e1)), fabsf(e2));
    float delta_t = 3.0f * (max_e * max_zt * G3 + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return false;
    if (!(t <= ray.t_max)) return false;
    t_out = t;
    b0_o = b0; b1_o = b1; b2_o = b2;
    return true;
}

// ---- constant medium: free-flight distance from a hash of the ray -----------------------
// One multiply-xor step per word of the ray and a full mixing round at the end: 32-bit
// integer multiplies run at a quarter of the ALU rate, and this is part of the most
// frequent primitive test of scenes filled with fog.
RT_DEV float medium_uniform(const Ray &ray, uint32_t prim) {
    const uint32_t k = 0x9E3779B1u;
    uint32_t h = (prim ^ __float_as_uint(ray.d.z)) * k;
    h = (h ^ __float_as_uint(ray.d.y)) * k;
    h = (h ^ __float_as_uint(ray.d.x)) * k;
    h = (h ^ __float_as_uint(ray.o.z)) * k;
    h = (h ^ __float_as_uint(ray.o.y)) * k;
    h = hash32(h ^ __float_as_uint(ray.o.x));
    return (float)(h >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
}
// The medium's test after the boundary's: the ray crosses the boundary between t_in and t_out_b
// (medium_span clips that to the ray), its direction has length len, and the medium wants to
// scatter it hit_dist further on (medium_scatter).
RT_DEV bool medium_span(float &t_in, float &t_out_b, float t_max) {
    if (t_in < 0.0f) t_in = 0.0f;
    if (t_out_b > t_max) t_out_b = t_max;
    return t_in < t_out_b;
}
RT_DEV bool medium_scatter(float t_in, float t_out_b, float t_max, float len, float hit_dist, float &t_out) {
    float inside = (t_out_b - t_in) * len;
    if (hit_dist > inside) return false;
    float t = t_in + hit_dist / len;
    if (!(t > 0.0f && t <= t_max)) return false;
    t_out = t;
    return true;
}

RT_DEV bool medium_test(const SceneDev &S, const LeafPrim &lp, const Ray &ray, v3 rd, float &t_out) {
    float t_in, t_out_b, neg_inv_density;
    if (lp.kind & kMediumInSphere) { // everything the test needs is in the leaf entry
        const float4 sp = lp.geom;
        neg_inv_density = __uint_as_float(lp.index);
        if (!sphere_roots(ray.o - mk(sp.x, sp.y, sp.z), ray.d, sp.w, t_in, t_out_b)) return false;
    } else {
        const Medium m = S.media[lp.index];
        neg_inv_density = m.neg_inv_density;
        int fn, ff;
        if (m.xform == 0) {
            if (!box_interval_rd(ray.o, ray.d, rd, load_box(S.boxes, m.boundary_index), t_in, t_out_b, fn, ff))
                return false;
        } else {
            v3 o, d;
            to_object(S, m.xform, ray, o, d);
            if (!box_interval(o, d, load_box(S.boxes, m.boundary_index), t_in, t_out_b, fn, ff)) return false;
        }
    }
    if (!medium_span(t_in, t_out_b, ray.t_max)) return false;
    const float len = length(ray.d);
    const float hit_dist = neg_inv_density * p_log(medium_uniform(ray, lp.prim));
    return medium_scatter(t_in, t_out_b, ray.t_max, len, hit_dist, t_out);
}

// ---- one primitive ---------------------------------------------------------------------
// aux: triangle barycentrics / box face.  Returns true for 0 < t <= ray.t_max.
// rd = 1.0f / ray.d, as the traversal loop holds it.
RT_DEV bool prim_test(const SceneDev &S, const LeafPrim &lp, const Ray &ray, v3 rd, float &t, float &a0,
                      float &a1, float &a2) {
    uint32_t kind = lp.kind & 0xffu;
    float t0, t1;
    switch (kind) {
    case RT_SPHERE: {
        float4 sp = lp.geom;
        if (!sphere_roots(ray.o - mk(sp.x, sp.y, sp.z), ray.d, sp.w, t0, t1)) return false;
        return sphere_pick(t0, t1, ray.t_max, t);
    }
    case RT_MOVING_SPHERE: {
        MSphere m = S.msph[lp.index];
        if (!sphere_roots(ray.o - msphere_centre(m, ray.time), ray.d, m.r, t0, t1)) return false;
        return sphere_pick(t0, t1, ray.t_max, t);
    }
    case RT_TRIANGLE: {
        const float4 *v = S.tris + 3 * (size_t)lp.index;
        float4 q0 = v[0], q1 = v[1], q2 = v[2];
        return tri_test(mk(q0.x, q0.y, q0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), ray, t, a0,
                        a1, a2);
    }
    case RT_RECT: {
        rt_rect r = S.rects[lp.index];
        v3 o, d;
        int A, B;
        to_object(S, lp.kind >> 16, ray, o, d);
        rect_axes(r.axis, A, B);
        float tt = (r.k - comp(o, (int)r.axis)) / comp(d, (int)r.axis);
        if (!(tt > 0.0f && tt <= ray.t_max)) return false;
        float a = comp(o, A) + tt * comp(d, A);
        float b = comp(o, B) + tt * comp(d, B);
        if (a < r.a0 || a > r.a1 || b < r.b0 || b > r.b1) return false;
        t = tt;
        return true;
    }
    case RT_BOX: {
        v3 o, d;
        int fn, ff;
        if ((lp.kind >> 16) == 0) {
            if (!box_interval_rd(ray.o, ray.d, rd, load_box(S.boxes, lp.index), t0, t1, fn, ff)) return false;
        } else {
            to_object(S, lp.kind >> 16, ray, o, d);
            if (!box_interval(o, d, load_box(S.boxes, lp.index), t0, t1, fn, ff)) return false;
        }
        float tt = t0;
        int face = fn;
        if (!(t0 > 0.0f)) { tt = t1; face = ff; }
        if (!(tt > 0.0f && tt <= ray.t_max)) return false;
        t = tt;
        a0 = (float)face;
        return true;
    }
    case RT_GSPHERE: return gsphere_test(S.gsph[lp.index], ray, t);
    case RT_MEDIUM: return medium_test(S, lp, ray, rd, t);
    default: return false;
    }
}

template <bool WANT_UV = true>
RT_DEV void sphere_surface(v3 oc, v3 d, v3 c, float r, float t, Surf &out) {
    float ar = fabsf(r);
    v3 p = oc + d * t;
    p = p * (ar / length(p));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1e-5f * ar;
    if (WANT_UV) {
        float phi = p_atan2(p.y, p.x);
        if (phi < 0.0f) phi += kTwoPi;
        float theta = p_acos(clampf(p.z / ar, -1.0f, 1.0f));
        out.u = phi / kTwoPi;
        out.v = (theta - kPi) / (0.0f - kPi);
    }
    out.n = p / r;
    out.p = p + c;
    v3 ap = vabs(p);
    v3 e = ap * G5;
    out.perr = e * (G3 + 1.0f) + (ap + vabs(c)) * G3;
}

RT_DEV void to_world(const SceneDev &S, uint32_t xform, v3 p, v3 e, v3 n, Surf &out) {
    if (xform) {
        const rt_xform &x = S.xforms[xform - 1];
        out.p = xf_point_err(x.o2w, p, e, out.perr);
        out.n = normalize(xf_normal(x.w2o, n));
    } else {
        out.p = p;
        out.perr = e;
        out.n = n;
    }
}

RT_DEV void set3(float *a, int i, float v) { a[i] = v; }

// Everything the hit record, the shader and the next ray need for an accepted hit.
RT_DEV void prim_finalize(const SceneDev &S, const LeafPrim &lp, const Ray &ray, float t, float a0,
                          float a1, float a2, Surf &out) {
    uint32_t kind = lp.kind & 0xffu;
    bool reverse = (lp.kind & RT_PRIM_REVERSE) != 0;
    uint32_t xform = lp.kind >> 16;
    out.t = t;
    out.flags = kind;
    out.u = 0.0f;
    out.v = 0.0f;
    switch (kind) {
    case RT_SPHERE: {
        float4 sp = lp.geom;
        v3 c = mk(sp.x, sp.y, sp.z);
        sphere_surface(ray.o - c, ray.d, c, sp.w, t, out);
        break;
    }
    case RT_MOVING_SPHERE: {
        MSphere m = S.msph[lp.index];
        v3 c = msphere_centre(m, ray.time);
        sphere_surface(ray.o - c, ray.d, c, m.r, t, out);
        break;
    }
    case RT_TRIANGLE: { // triangle.rs:204-223,253-255,275-345
        const float4 *v = S.tris + 3 * (size_t)lp.index;
        float4 q0 = v[0], q1 = v[1], q2 = v[2];
        v3 p0 = mk(q0.x, q0.y, q0.z), p1 = mk(q1.x, q1.y, q1.z), p2 = mk(q2.x, q2.y, q2.z);
        float b0 = a0, b1 = a1, b2 = a2;
        out.p = mk(p0.x * b0 + p1.x * b1 + p2.x * b2, p0.y * b0 + p1.y * b1 + p2.y * b2,
                   p0.z * b0 + p1.z * b1 + p2.z * b2);
        out.perr = mk(fabsf(p0.x * b0) + fabsf(p1.x * b1) + fabsf(p2.x * b2),
                      fabsf(p0.y * b0) + fabsf(p1.y * b1) + fabsf(p2.y * b2),
                      fabsf(p0.z * b0) + fabsf(p1.z * b1) + fabsf(p2.z * b2)) * G7;
        if (S.tri_uvs) {
            const float *uv = S.tri_uvs + 6 * (size_t)lp.index;
            out.u = uv[0] * b0 + uv[2] * b1 + uv[4] * b2;
            out.v = uv[1] * b0 + uv[3] * b1 + uv[5] * b2;
        } else {
            out.u = 0.0f * b0 + 1.0f * b1 + 0.0f * b2;
            out.v = 0.0f * b0 + 0.0f * b1 + 1.0f * b2;
        }
        v3 n = normalize(cross(p0 - p2, p1 - p2));
        if (S.tri_normals) {
            const float *nn = S.tri_normals + 9 * (size_t)lp.index;
            v3 ns = normalize(mk(nn[0] * b0 + nn[3] * b1 + nn[6] * b2, nn[1] * b0 + nn[4] * b1 + nn[7] * b2,
                                 nn[2] * b0 + nn[5] * b1 + nn[8] * b2));
            if (dot(n, ns) < 0.0f) n = -n;
        } else if (reverse) {
            n = -n;
        }
        out.n = n;
        break;
    }
    case RT_RECT: {
        rt_rect r = S.rects[lp.index];
        v3 oo, dd;
        int A, B;
        to_object(S, xform, ray, oo, dd);
        rect_axes(r.axis, A, B);
        float ta = t * comp(dd, A), tb = t * comp(dd, B);
        float pa = comp(oo, A) + ta, pb = comp(oo, B) + tb;
        float ea = (fabsf(comp(oo, A)) + fabsf(ta)) * G5;
        float eb = (fabsf(comp(oo, B)) + fabsf(tb)) * G5;
        float ek = fabsf(r.k) * G5;
        float nk = reverse ? -1.0f : 1.0f;
        v3 p, e, n;
        if (r.axis == 0) { p = mk(r.k, pa, pb); e = mk(ek, ea, eb); n = mk(nk, 0.0f, 0.0f); }
        else if (r.axis == 1) { p = mk(pa, r.k, pb); e = mk(ea, ek, eb); n = mk(0.0f, nk, 0.0f); }
        else { p = mk(pa, pb, r.k); e = mk(ea, eb, ek); n = mk(0.0f, 0.0f, nk); }
        out.u = (pa - r.a0) / (r.a1 - r.a0);
        out.v = (pb - r.b0) / (r.b1 - r.b0);
        out.flags |= r.axis << 8;
        to_world(S, xform, p, e, n, out);
        break;
    }
    case RT_BOX: {
        rt_box bx = S.boxes[lp.index];
        v3 oo, dd;
        to_object(S, xform, ray, oo, dd);
        int face = (int)a0;
        int ax = face >> 1, A, B;
        rect_axes((uint32_t)ax, A, B);
        float ta = t * comp(dd, A), tb = t * comp(dd, B);
        float pa = comp(oo, A) + ta, pb = comp(oo, B) + tb;
        float pk = (face & 1) ? bx.hi[ax] : bx.lo[ax];
        float ea = (fabsf(comp(oo, A)) + fabsf(ta)) * G5;
        float eb = (fabsf(comp(oo, B)) + fabsf(tb)) * G5;
        float ek = fabsf(pk) * G5;
        float nk = (face & 1) ? 1.0f : -1.0f;
        v3 p, e, n;
        if (ax == 0) { p = mk(pk, pa, pb); e = mk(ek, ea, eb); n = mk(nk, 0.0f, 0.0f); }
        else if (ax == 1) { p = mk(pa, pk, pb); e = mk(ea, ek, eb); n = mk(0.0f, nk, 0.0f); }
        else { p = mk(pa, pb, pk); e = mk(ea, eb, ek); n = mk(0.0f, 0.0f, nk); }
        out.u = (pa - bx.lo[A]) / (bx.hi[A] - bx.lo[A]);
        out.v = (pb - bx.lo[B]) / (bx.hi[B] - bx.lo[B]);
        out.flags |= (uint32_t)face << 8;
        to_world(S, xform, p, e, n, out);
        break;
    }
    case RT_GSPHERE: { // sphere.rs:110-154, interaction.rs:138-177,243-257
        const GSphere &g = S.gsph[lp.index];
        v3 go, gd;
        float phi;
        gsphere_ray(g, ray, go, gd);
        v3 p = gsphere_point(g, go, gd, t, phi);
        out.u = phi / g.phi_max;
        float theta = p_acos(clampf(p.z / g.radius, -1.0f, 1.0f));
        out.v = (theta - g.theta_min) / (g.theta_max - g.theta_min);
        float z_radius = sqrtf(p.x * p.x + p.y * p.y);
        float inv_zr = 1.0f / z_radius;
        float cos_phi = p.x * inv_zr, sin_phi = p.y * inv_zr;
        v3 dpdu = mk(-g.phi_max * p.y, g.phi_max * p.x, 0.0f);
        v3 dpdv = mk(p.z * cos_phi, p.z * sin_phi, -g.radius * p_sin(theta)) * (g.theta_max - g.theta_min);
        v3 n = normalize(cross(dpdu, dpdv));
        if (reverse != (g.flip != 0)) n = -n;
        v3 e = vabs(p) * G5;
        out.p = xf_point_err(g.o2w, p, e, out.perr);
        out.n = normalize(xf_normal(g.w2o, n));
        break;
    }
    default: // RT_MEDIUM
        out.p = ray.o + ray.d * t;
        out.perr = mk(0.0f, 0.0f, 0.0f);
        out.n = mk(1.0f, 0.0f, 0.0f);
        break;
    }
}

// prim_finalize for the hits the plain shader takes (no kRareShade): the four surface kinds
// without instance transform and scattering inside a medium, and no surface (u, v), which
// only image and uv-checker textures read.
// Point, error bound and normal come out of the same arithmetic as above.
RT_DEV void prim_finalize_plain(const SceneDev &S, const LeafPrim &lp, const Ray &ray, float t, float a0,
                                float a1, float a2, Surf &out) {
    uint32_t kind = lp.kind & 0xffu;
    bool reverse = (lp.kind & RT_PRIM_REVERSE) != 0;
    out.t = t;
    out.flags = kind;
    out.u = 0.0f;
    out.v = 0.0f;
    if (kind == RT_SPHERE) {
        float4 sp = lp.geom;
        v3 c = mk(sp.x, sp.y, sp.z);
        sphere_surface<false>(ray.o - c, ray.d, c, sp.w, t, out);
    } else if (kind == RT_MEDIUM) {
        out.p = ray.o + ray.d * t;
        out.perr = mk(0.0f, 0.0f, 0.0f);
        out.n = mk(1.0f, 0.0f, 0.0f);
    } else if (kind == RT_BOX) {
        rt_box bx = S.boxes[lp.index];
        int face = (int)a0;
        int ax = face >> 1, A, B;
        rect_axes((uint32_t)ax, A, B);
        float ta = t * comp(ray.d, A), tb = t * comp(ray.d, B);
        float pa = comp(ray.o, A) + ta, pb = comp(ray.o, B) + tb;
        float pk = (face & 1) ? bx.hi[ax] : bx.lo[ax];
        float ea = (fabsf(comp(ray.o, A)) + fabsf(ta)) * G5;
        float eb = (fabsf(comp(ray.o, B)) + fabsf(tb)) * G5;
        float ek = fabsf(pk) * G5;
        float nk = (face & 1) ? 1.0f : -1.0f;
        if (ax == 0) { out.p = mk(pk, pa, pb); out.perr = mk(ek, ea, eb); out.n = mk(nk, 0.0f, 0.0f); }
        else if (ax == 1) { out.p = mk(pa, pk, pb); out.perr = mk(ea, ek, eb); out.n = mk(0.0f, nk, 0.0f); }
        else { out.p = mk(pa, pb, pk); out.perr = mk(ea, eb, ek); out.n = mk(0.0f, 0.0f, nk); }
    } else if (kind == RT_TRIANGLE) {
        const float4 *v = S.tris + 3 * (size_t)lp.index;
        float4 q0 = v[0], q1 = v[1], q2 = v[2];
        v3 p0 = mk(q0.x, q0.y, q0.z), p1 = mk(q1.x, q1.y, q1.z), p2 = mk(q2.x, q2.y, q2.z);
        float b0 = a0, b1 = a1, b2 = a2;
        out.p = mk(p0.x * b0 + p1.x * b1 + p2.x * b2, p0.y * b0 + p1.y * b1 + p2.y * b2,
                   p0.z * b0 + p1.z * b1 + p2.z * b2);
        out.perr = mk(fabsf(p0.x * b0) + fabsf(p1.x * b1) + fabsf(p2.x * b2),
                      fabsf(p0.y * b0) + fabsf(p1.y * b1) + fabsf(p2.y * b2),
                      fabsf(p0.z * b0) + fabsf(p1.z * b1) + fabsf(p2.z * b2)) * G7;
        v3 n = normalize(cross(p0 - p2, p1 - p2));
        if (S.tri_normals) {
            const float *nn = S.tri_normals + 9 * (size_t)lp.index;
            v3 ns = normalize(mk(nn[0] * b0 + nn[3] * b1 + nn[6] * b2, nn[1] * b0 + nn[4] * b1 + nn[7] * b2,
                                 nn[2] * b0 + nn[5] * b1 + nn[8] * b2));
            if (dot(n, ns) < 0.0f) n = -n;
        } else if (reverse) {
            n = -n;
        }
        out.n = n;
    } else { // RT_RECT
        rt_rect r = S.rects[lp.index];
        int A, B;
        rect_axes(r.axis, A, B);
        float ta = t * comp(ray.d, A), tb = t * comp(ray.d, B);
        float pa = comp(ray.o, A) + ta, pb = comp(ray.o, B) + tb;
        float ea = (fabsf(comp(ray.o, A)) + fabsf(ta)) * G5;
        float eb = (fabsf(comp(ray.o, B)) + fabsf(tb)) * G5;
        float ek = fabsf(r.k) * G5;
        float nk = reverse ? -1.0f : 1.0f;
        if (r.axis == 0) { out.p = mk(r.k, pa, pb); out.perr = mk(ek, ea, eb); out.n = mk(nk, 0.0f, 0.0f); }
        else if (r.axis == 1) { out.p = mk(pa, r.k, pb); out.perr = mk(ea, ek, eb); out.n = mk(0.0f, nk, 0.0f); }
        else { out.p = mk(pa, pb, r.k); out.perr = mk(ea, eb, ek); out.n = mk(0.0f, 0.0f, nk); }
    }
}

// ---- slab test, geometry/src/bounds3.rs:212-238 ------------------------------------------
// Bounds3::intersect_inv, geometry/src/bounds3.rs:212-238.  t_near is the entry
// distance the final comparison `t_near < ray.t_max` used: the only term of the
// test that depends on ray.t_max, so a caller that keeps it can repeat the test
// against a shortened ray without the box.
#ifdef RT_SLAB_BRANCHY
RT_DEV bool slab(float4 n0, float4 n1, const Ray &ray, v3 inv, bool negx, bool negy, bool negz, float &t_near) {
    const float g = 1.0f + 2.0f * G3;
    float t_min = ((negx ? n1.x : n0.x) - ray.o.x) * inv.x;
    float t_max = ((negx ? n0.x : n1.x) - ray.o.x) * inv.x;
    float ty_min = ((negy ? n1.y : n0.y) - ray.o.y) * inv.y;
    float ty_max = ((negy ? n0.y : n1.y) - ray.o.y) * inv.y;
    t_max *= g;
    ty_max *= g;
    if (t_min > ty_max || ty_min > t_max) return false;
    t_min = sel_max(t_min, ty_min);
    t_max = sel_min(t_max, ty_max);
    float tz_min = ((negz ? n1.z : n0.z) - ray.o.z) * inv.z;
    float tz_max = ((negz ? n0.z : n1.z) - ray.o.z) * inv.z;
    tz_max *= g;
    if (t_min > tz_max || tz_min > t_max) return false;
    t_min = sel_max(t_min, tz_min);
    t_max = sel_min(t_max, tz_max);
    t_near = t_min;
    return t_min < ray.t_max && t_max > 0.0f;
}
#else
RT_DEV bool slab(float4 n0, float4 n1, const Ray &ray, v3 inv, bool negx, bool negy, bool negz, float &t_near) {
    // straight-line form: the reference's early returns only skip work, the outcome is the
    // conjunction of its three conditions
    const float g = 1.0f + 2.0f * G3;
    float t_min = ((negx ? n1.x : n0.x) - ray.o.x) * inv.x;
    float t_max = ((negx ? n0.x : n1.x) - ray.o.x) * inv.x;
    float ty_min = ((negy ? n1.y : n0.y) - ray.o.y) * inv.y;
    float ty_max = ((negy ? n0.y : n1.y) - ray.o.y) * inv.y;
    float tz_min = ((negz ? n1.z : n0.z) - ray.o.z) * inv.z;
    float tz_max = ((negz ? n0.z : n1.z) - ray.o.z) * inv.z;
    t_max *= g;
    ty_max *= g;
    tz_max *= g;
    const bool miss_xy = t_min > ty_max || ty_min > t_max;
    t_min = sel_max(t_min, ty_min);
    t_max = sel_min(t_max, ty_max);
    const bool miss_z = t_min > tz_max || tz_min > t_max;
    t_min = sel_max(t_min, tz_min);
    t_max = sel_min(t_max, tz_max);
    t_near = t_min;
    return !miss_xy && !miss_z && t_min < ray.t_max && t_max > 0.0f;
}

#endif

} // namespace rt
