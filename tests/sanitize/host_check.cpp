// Host-side sanitizer run (AddressSanitizer + UBSan on the CPU code only): builds scenes of
// every primitive kind with each split method through librt355's host code (validation,
// bounds, tree builders on several threads, flattening; no GPU is touched) and runs the CPU
// oracle on the same description (build, closest hit, any hit, a small render with every
// sampler and a filter).  Also feeds the validator malformed descriptions.
// Built and run by tests/sanitize/run.sh.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../bindings/cpp/rt355.hpp"
extern "C" {
#include "../../oracle/orc.h"
}

static uint64_t state = 0x853c49e6748fea9bULL;
static float frand() {
    state = state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (float)((state >> 40) & 0xffffff) / 16777216.0f;
}

static void fill(rt355::SceneBuilder &b, int n) {
    rt_texture tex{};
    tex.kind = RT_TEX_SOLID;
    tex.rgb[0] = tex.rgb[1] = tex.rgb[2] = 0.7f;
    uint32_t t = b.texture(tex);
    rt_material m{};
    m.kind = RT_MAT_LAMBERTIAN;
    m.texture = t;
    uint32_t mat = b.material(m);
    for (int i = 0; i < n; i++) {
        float x = frand() * 20 - 10, y = frand() * 20 - 10, z = frand() * 20 - 10;
        switch (i % 4) {
        case 0: b.add_sphere({x, y, z}, 0.1f + frand(), mat); break;
        case 1: {
            float v[9] = {x, y, z, x + frand(), y, z + frand(), x, y + frand(), z};
            b.add_triangle(v, mat);
            break;
        }
        case 2: b.add_box({x, y, z}, {x + 0.2f + frand(), y + 0.2f + frand(), z + 0.2f + frand()}, mat); break;
        default: {
            rt_rect r{};
            r.axis = (uint32_t)(i / 4 % 3);
            r.k = x;
            r.a0 = y; r.a1 = y + 0.5f + frand();
            r.b0 = z; r.b1 = z + 0.5f + frand();
            b.add_rect(r, mat);
        }
        }
    }
    if (n >= 5) {
        // media: a thin fog around everything (kept out of the tree), a fog ball inside a glass
        // shell away from the rest (a ball: rays from inside skip the tree), one without a shell,
        // one with a neighbour too close to count, and a box of smoke
        rt_material iso{};
        iso.kind = RT_MAT_ISOTROPIC;
        iso.texture = t;
        const uint32_t fog = b.material(iso);
        rt_material glass{};
        glass.kind = RT_MAT_DIELECTRIC;
        glass.ior = 1.5f;
        const uint32_t gl = b.material(glass);
        b.add_medium_sphere({0, 0, 0}, 200.0f, 0.002f, fog);
        b.add_sphere({40, 0, 0}, 5.0f, gl);
        b.add_medium_sphere({40, 0, 0}, 5.0f, 0.5f, fog);
        b.add_medium_sphere({-40, 0, 0}, 4.0f, 0.3f, fog);
        b.add_medium_sphere({0, 40, 0}, 5.0f, 0.5f, fog);
        b.add_sphere({0, 40, 5.504f}, 0.5f, mat);
        b.add_medium_box({20, 20, 20}, {24, 23, 25}, 0.2f, fog);
    }
    b.background(0.5f, 0.6f, 0.9f);
}

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "host_check: %s failed at line %d\n", #cond, __LINE__); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main() {
    for (int n : {0, 1, 5, 300, 20000}) {
        rt355::SceneBuilder b;
        fill(b, n);
        rt_scene_desc d = b.desc();
        for (uint32_t split = 0; split < 4; split++) {
            rt_build_opts o{};
            o.max_prims_in_node = 1 + split;
            o.split_method = split;
            rt_scene *s = nullptr;
            CHECK(rt_scene_create(&d, &o, &s) == RT_OK);
            rt_scene_info info{};
            CHECK(rt_scene_get_info(s, &info) == RT_OK);
            CHECK(n < 5 || (info.n_enclosing == 1 && info.n_balls == 2));
            std::vector<rt_linear_node> nodes(info.n_nodes ? info.n_nodes : 1);
            std::vector<uint32_t> order(info.n_prims ? info.n_prims : 1);
            CHECK(rt_scene_get_nodes(s, nodes.data(), nodes.size()) >= 0);
            CHECK(rt_scene_get_prim_order(s, order.data(), order.size()) >= 0);
            orc_scene *os = nullptr;
            CHECK(orc_scene_create(&d, &o, &os) == RT_OK);
            rt_scene_info oinfo{};
            CHECK(orc_scene_get_info(os, &oinfo) == RT_OK && oinfo.n_balls == info.n_balls &&
                  oinfo.n_enclosing == info.n_enclosing);
            // same tree on both sides
            std::vector<rt_linear_node> onodes(nodes.size());
            CHECK(orc_scene_get_nodes(os, onodes.data(), onodes.size()) >= 0);
            CHECK(info.n_nodes == 0 || memcmp(nodes.data(), onodes.data(), info.n_nodes * sizeof(rt_linear_node)) == 0);
            std::vector<rt_ray> rays(2000);
            size_t k = 0;
            for (auto &r : rays) {
                r.o[0] = frand() * 30 - 15; r.o[1] = frand() * 30 - 15; r.o[2] = frand() * 30 - 15;
                if (k++ % 4 == 0) { // a quarter start inside the balls
                    r.o[0] = (k % 8 < 4 ? 40.0f : -40.0f) + frand() * 4 - 2; r.o[1] = frand() * 4 - 2; r.o[2] = frand() * 4 - 2;
                }
                r.d[0] = frand() - 0.5f; r.d[1] = frand() - 0.5f; r.d[2] = frand() - 0.5f;
                r.t_max = INFINITY;
                r.time = frand();
            }
            std::vector<rt_hit> hits(rays.size());
            std::vector<uint8_t> occ(rays.size());
            CHECK(orc_intersect(os, rays.data(), rays.size(), hits.data(), 3, nullptr) == RT_OK);
            CHECK(orc_occluded(os, rays.data(), rays.size(), occ.data(), 3) == RT_OK);
            for (size_t i = 0; i < rays.size(); i++) CHECK((hits[i].prim != RT_MISS) == (occ[i] != 0));
            if (split == 0 && n == 300) {
                rt_camera cam{};
                // pinhole at (0, 0, -40) looking along +z: raster (x, y) -> camera (ax - t, t - by, 1)
                const float t = 0.4f, w = 24.0f, h = 16.0f;
                const float r2c[16] = {2 * t / w, 0, 0, -t, 0, -2 * t / h, 0, t, 0, 0, 0, 1, 0, 0, 0, 1};
                const float c2w[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -40};
                memcpy(cam.raster_to_camera, r2c, sizeof r2c);
                memcpy(cam.camera_to_world, c2w, sizeof c2w);
                cam.focal_distance = 1e6f;
                for (uint32_t sampler = 0; sampler <= RT_SAMPLER_STRATIFIED; sampler++) {
                    rt_render_opts ro{};
                    ro.width = 24; ro.height = 16; ro.spp = 4; ro.max_depth = 6; ro.shard_count = 1;
                    ro.sampler = sampler;
                    ro.sampler_arg = 2;
                    if (sampler == RT_SAMPLER_HALTON) {
                        ro.filter.kind = RT_FILTER_GAUSSIAN;
                        ro.filter.radius[0] = ro.filter.radius[1] = 1.5f;
                        ro.filter.a = 2.0f;
                    }
                    std::vector<float> img(24 * 16 * 4);
                    rt_stats st{};
                    CHECK(orc_render(os, &cam, &ro, img.data(), &st, 2) == RT_OK);
                    for (float v : img) CHECK(std::isfinite(v));
                }
            }
            orc_scene_destroy(os);
            // no GPU here: the compute entry points must say so, not crash
            CHECK(rt_intersect(s, rays.data(), rays.size(), hits.data(), 0) == RT_ERR_NO_DEVICE || n == 0);
            rt_scene_destroy(s);
        }
    }
    // malformed descriptions are rejected
    {
        rt355::SceneBuilder b;
        fill(b, 8);
        rt_scene_desc d = b.desc();
        rt_build_opts o{};
        rt_scene *s = nullptr;
        std::vector<rt_prim_ref> bad(d.prims, d.prims + d.n_prims);
        bad[3].index = 1000;
        d.prims = bad.data();
        CHECK(rt_scene_create(&d, &o, &s) == RT_ERR_INVALID);
        bad[3].index = 0;
        bad[2].kind = 99;
        CHECK(rt_scene_create(&d, &o, &s) == RT_ERR_INVALID);
        bad[2].kind = RT_BOX;
        bad[2].material = 77;
        CHECK(rt_scene_create(&d, &o, &s) == RT_ERR_INVALID);
        o.split_method = 9;
        CHECK(rt_scene_create(&d, &o, &s) == RT_ERR_INVALID);
        CHECK(rt_scene_create(nullptr, &o, &s) == RT_ERR_INVALID);
    }
    uint32_t t32[260];
    uint64_t vdc[52], inv[52];
    CHECK(rt_sobol_tables(1920, 1080, t32, vdc, inv) == RT_OK);
    puts("host_check ok");
    return 0;
}
