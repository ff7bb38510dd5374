// Smallest C++ caller of librt355.so through rt355.hpp.
//   g++ -std=c++17 -I include -I bindings/cpp bindings/cpp/example.cpp
//       -L raytracer_amd -lrt355 -Wl,-rpath,$PWD/raytracer_amd -o example
// Prints the world bound and, when a GPU is present, the hit of three rays.
// Without a GPU the compute calls must fail with RT_ERR_NO_DEVICE: exit 0 then too.
#include <cstdio>

#include "rt355.hpp"

int main() {
    rt355::SceneBuilder s;
    s.texture({RT_TEX_SOLID, {0.5f, 0.5f, 0.5f}, 0, 0, 0.f, 0});
    s.material({RT_MAT_LAMBERTIAN, 0, 0.f, 1.f});
    s.add_sphere({0.f, 0.f, -5.f}, 1.f);
    s.add_sphere({3.f, 0.f, -5.f}, 1.f);
    const float tri[9] = {-10.f, -1.f, 0.f, 10.f, -1.f, 0.f, 0.f, -1.f, -20.f};
    s.add_triangle(tri);
    rt355::Bvh bvh(s, 4, rt355::SplitMethod::SAH);
    rt355::Bounds3 b = bvh.world_bound();
    std::printf("world_bound [%g %g %g] [%g %g %g] nodes %zu\n", b.min[0], b.min[1], b.min[2], b.max[0], b.max[1],
                b.max[2], bvh.nodes().size());
    std::vector<rt_ray> rays = {
        {{0.f, 0.f, 0.f}, {0.f, 0.f, -1.f}, 1e30f, 0.f},   // first sphere, t = 4
        {{0.f, 5.f, 0.f}, {0.f, 1.f, 0.f}, 1e30f, 0.f},    // miss
        {{0.f, 0.f, -2.f}, {0.f, -1.f, 0.f}, 1e30f, 0.f},  // triangle, t = 1
    };
    try {
        std::vector<rt_hit> hits = bvh.intersect(rays);
        for (const rt_hit &h : hits) std::printf("prim %d t %g n [%g %g %g]\n", int(h.prim), h.t, h.n[0], h.n[1], h.n[2]);
        bool ok = hits[0].prim == 0 && hits[0].t == 4.f && hits[1].prim == RT_MISS && hits[2].prim == 2 && hits[2].t == 1.f;
        ok = ok && bvh.does_intersect(rays[0]) && !bvh.does_intersect(rays[1]) && bvh.intersect(rays[1]) == std::nullopt;
        std::printf(ok ? "ok\n" : "MISMATCH\n");
        return ok ? 0 : 1;
    } catch (const rt355::Error &e) {
        std::printf("no device: %s\n", e.what());
        return e.code == RT_ERR_NO_DEVICE ? 0 : 1;
    }
}
