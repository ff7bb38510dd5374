/*
 * orc_math.h -- scalar fp32 arithmetic of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This directory is a CPU restatement of the
 * reference algorithms used as the parity checker by tests/, by
 * __graft_entry__.smoke() and by the cpu_baseline leg of bench.py.  Nothing in
 * the shipped library includes, links or calls it.
 *
 * Parity status: the reference holds no golden vector for ray traversal or
 * ray/shape intersection, so that part is "parity unpinned"; the AABB helpers
 * are pinned by geometry/src/tests/bounds.rs:37-117 and the rest is
 * cross-checked by an independent fp64 brute-force intersector in tests/.
 *
 * Every expression is written with explicit evaluation order and must be
 * compiled with -ffp-contract=off: the GPU kernels evaluate the same
 * expressions in the same order, so results agree bit for bit.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
/* Div<T> for Vector3 / Point3 (vector3.rs:260-264, point3.rs:271-275): one reciprocal, three products */
static inline v3 v3_div(v3 a, float s) {
    const float inv = 1.0f / s;
    return v3_make(a.x * inv, a.y * inv, a.z * inv);
}
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_abs(v3 a) { return v3_make(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
/* dot: ((x*x + y*y) + z*z), geometry/src/vector3.rs dot() order */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_len(v3 a) { return sqrtf(v3_dot(a, a)); }
/* normalise = self / length (vector3.rs:140-142) */
static inline v3 v3_norm(v3 a) { return v3_div(a, v3_len(a)); }
static inline float v3_get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Number::max / Number::min are plain selects, geometry/src/number.rs:121-138 */
static inline float sel_max(float a, float b) { return a > b ? a : b; }
static inline float sel_min(float a, float b) { return a < b ? a : b; }

/* f32::EPSILON = 2^-23 is the reference's machine epsilon (number.rs:205). */
#define ORC_EPS 1.1920928955078125e-7f
/* gamma(n) = n*eps / (1 - n*eps), geometry/src/float.rs:44-46 */
static inline float orc_gamma(int n) {
    return ((float)n * ORC_EPS) / (1.0f - (float)n * ORC_EPS);
}

/* next_float_up / next_float_down, geometry/src/float.rs:8-41 with the
 * direction of the positive branch of next_float_down corrected. */
static inline float next_float_up(float v) {
    if (isinf(v) && v > 0.0f) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t b = f2u(v);
    if (v >= 0.0f) b += 1; else b -= 1;
    return u2f(b);
}
static inline float next_float_down(float v) {
    if (isinf(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t b = f2u(v);
    if (v > 0.0f) b -= 1; else b += 1;
    return u2f(b);
}

/* ---- transcendental functions ------------------------------------------
 * Polynomial restatements built from + - * / only, so that the CPU and the
 * GPU produce the same bits (libm and the device math library round
 * differently).  Absolute error is below 2e-7 over the ranges used. */

static inline float orc_sin(float x) {
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
static inline float orc_cos(float x) { return orc_sin(x + 1.57079637f); }

/* natural log of a positive normal float */
static inline float orc_log(float x) {
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 127;
    float f = u2f((ix & 0x007fffffu) | 0x3f800000u);
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

static inline float orc_atan2(float y, float x) {
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

static inline float orc_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* acos on [-1, 1] */
static inline float orc_acos(float x) {
    float s = sqrtf((1.0f - x) * (1.0f + x));
    return orc_atan2(s, x);
}

/* ---- PCG32, geometry/src/rng.rs:5-67 ------------------------------------ */
typedef struct { uint64_t state, inc; } orc_rng;

static inline uint32_t orc_rng_u32(orc_rng *r) {
    uint64_t old = r->state;
    r->state = old * 0x5851f42d4c957f2dULL + r->inc;
    uint32_t x = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (x >> rot) | (x << ((~rot + 1u) & 31u));
}
static inline void orc_rng_seq(orc_rng *r, uint64_t index) {
    r->state = 0;
    r->inc = (index << 1) | 1u;
    (void)orc_rng_u32(r);
    r->state += 0x853c49e6748fea9bULL;
    (void)orc_rng_u32(r);
}
/* uniform_float: min(1 - eps, (f32)u32 * 2^-32), rng.rs:63-67 */
static inline float orc_rng_f(orc_rng *r) {
    float num = (float)orc_rng_u32(r) * 2.3283064365386963e-10f;
    float lim = 1.0f - ORC_EPS;
    return lim < num ? lim : num;
}
/* uniform_u32_limit, rng.rs:51-59 */
static inline uint32_t orc_rng_limit(orc_rng *r, uint32_t limit) {
    uint32_t threshold = (~limit + 1u) % limit;
    for (;;) {
        uint32_t v = orc_rng_u32(r);
        if (v >= threshold) return v % limit;
    }
}

/* 32-bit mixing hash (PCG output permutation applied to a 32-bit LCG step) */
static inline uint32_t orc_hash32(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

#endif
