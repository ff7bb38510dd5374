// rt_probe.hpp -- measuring aids of rt_kernels.hip, kept out of the kernels' text.  None of them is
// active in the shipped library: every macro below is defined only by tools/build_variants.py builds.
//
//   -DRT_PROFILE      the counting kernels also time their phases with the shader clock and tally lane
//                     occupancy (Counters::p); rt_render_dev prints the totals.  PROF(...) keeps such
//                     statements where they measure and makes them vanish otherwise.
//   -DRT_DOUBLE_SLAB / _PRIM / _PRETEST / _RS (and _SHADE, in BounceFeed::shade_batch)
//                     cost probes: the named piece runs a second time on opaque copies of its inputs with
//                     its results thrown away, so that instruction-counter deltas between two builds
//                     measure that piece (tools/valu_count.sh).
//   -DRT_TAIL_PROBE   per-wave clock records at the end of the one-launch kernel (tools/tail_probe.py);
//                     its statements stay in LoopFeed, where the state they record lives.
#pragma once

#include "rt_device.hpp"

namespace rt {

#ifdef RT_PROFILE
#define PROF(...) __VA_ARGS__
#else
#define PROF(...)
#endif

// a value the compiler must treat as new: work done on it cannot be merged with work on the original
RT_DEV float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// defined in rt_kernels.hip
RT_DEV void pretest(const SceneDev &S, const Ray &ray, float &t_max, uint32_t &code);
RT_DEV v3 random_in_unit_sphere(Rng &rng);

namespace probe {

RT_DEV void double_slab(float4 q0, float4 q1, float4 q2, float4 q3, const Ray &ray, v3 inv, bool negx, bool negy,
                        bool negz) {
#ifdef RT_DOUBLE_SLAB
    q0.x = opaque(q0.x); q1.x = opaque(q1.x); q2.x = opaque(q2.x); q3.x = opaque(q3.x);
    float u0, u1;
    const bool g0 = slab(q0, q1, ray, inv, negx, negy, negz, u0);
    const bool g1 = slab(q2, q3, ray, inv, negx, negy, negz, u1);
    const float sink = (g0 ? u0 : 0.0f) + (g1 ? u1 : 0.0f);
    asm volatile("" ::"v"(sink));
#endif
}

RT_DEV void double_prim(const SceneDev &S, const LeafPrim &lp, const Ray &ray, v3 inv) {
#ifdef RT_DOUBLE_PRIM
    Ray r2 = ray;
    r2.o.x = opaque(r2.o.x);
    float t2 = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    const bool g = prim_test(S, lp, r2, inv, t2, c0, c1, c2);
    const float sink = g ? t2 + c0 + c1 + c2 : 0.0f;
    asm volatile("" ::"v"(sink));
#endif
}

RT_DEV void double_pretest(const SceneDev &S, const Ray &fresh) {
#ifdef RT_DOUBLE_PRETEST
    Ray again = fresh;
    again.o.x = opaque(again.o.x);
    float t2;
    uint32_t c2;
    pretest(S, again, t2, c2);
    const float sink = t2 + __uint_as_float(c2);
    asm volatile("" ::"v"(sink));
#endif
}

RT_DEV void double_unit_sphere(const Rng &rng) {
#ifdef RT_DOUBLE_RS
    Rng r2 = rng;
    r2.inc = (uint64_t)__float_as_uint(opaque(__uint_as_float((uint32_t)r2.inc))) | (r2.inc & 0xffffffff00000000ull);
    const v3 q = random_in_unit_sphere(r2);
    const float sink = q.x + q.y + q.z;
    asm volatile("" ::"v"(sink));
#endif
}

} // namespace probe
} // namespace rt
