// rt355.hpp -- header-only C++17 mirror of the reference's aggregate interface
// over the C ABI of librt355.so.
//
//   rt355::SceneBuilder   collects the primitive list handed to BVH::new
//                         (shapes/src/bvh.rs:108-112) as typed arrays
//   rt355::Bvh            Primitive<f32> as implemented by BVH<f32>
//                         (shapes/src/primitive.rs:8-27, bvh.rs:727-863):
//                         world_bound(), intersect(), does_intersect()
//
// Methods keep the trait's names and meaning; they take batches because one GPU
// launch per ray is meaningless.  A miss is std::nullopt / false.  Where the
// reference panics (bvh.rs:670-686) these throw rt355::Error carrying the code
// and rt_last_error().
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt355.h"

namespace rt355 {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *what) : std::runtime_error(std::string(what) + ": " + rt_last_error()), code(c) {}
};
inline void check(int rc, const char *what) {
    if (rc != RT_OK) throw Error(rc, what);
}

// SplitMethod, bvh.rs:10-22
enum class SplitMethod : uint32_t { SAH = RT_SPLIT_SAH, HLBVH = RT_SPLIT_HLBVH, Middle = RT_SPLIT_MIDDLE, EqualCounts = RT_SPLIT_EQUAL_COUNTS };

struct Bounds3 {
    std::array<float, 3> min, max;
};

class SceneBuilder {
  public:
    uint32_t material(rt_material m) { materials_.push_back(m); return uint32_t(materials_.size() - 1); }
    uint32_t texture(rt_texture t) { textures_.push_back(t); return uint32_t(textures_.size() - 1); }
    uint32_t xform(const rt_xform &x) { xforms_.push_back(x); return uint32_t(xforms_.size()); }  // 1-based
    // each add_* returns the primitive index that rt_hit.prim will report
    uint32_t add_sphere(std::array<float, 3> c, float r, uint32_t mat = 0) {
        spheres_.push_back({{c[0], c[1], c[2]}, r});
        return push(RT_SPHERE, uint32_t(spheres_.size() - 1), mat, 0);
    }
    uint32_t add_moving_sphere(const rt_moving_sphere &s, uint32_t mat = 0) {
        moving_.push_back(s);
        return push(RT_MOVING_SPHERE, uint32_t(moving_.size() - 1), mat, 0);
    }
    uint32_t add_triangle(const float v[9], uint32_t mat = 0, bool reverse_orientation = false) {
        tri_.insert(tri_.end(), v, v + 9);
        return push(RT_TRIANGLE | (reverse_orientation ? RT_PRIM_REVERSE : 0u), uint32_t(tri_.size() / 9 - 1), mat, 0);
    }
    uint32_t add_rect(const rt_rect &r, uint32_t mat = 0, uint32_t xf = 0) {
        rects_.push_back(r);
        return push(RT_RECT, uint32_t(rects_.size() - 1), mat, xf);
    }
    uint32_t add_box(std::array<float, 3> lo, std::array<float, 3> hi, uint32_t mat = 0, uint32_t xf = 0) {
        boxes_.push_back({{lo[0], lo[1], lo[2]}, 0.f, {hi[0], hi[1], hi[2]}, 0.f});
        return push(RT_BOX, uint32_t(boxes_.size() - 1), mat, xf);
    }
    // Sphere::new (shapes/src/sphere.rs:28-47)
    uint32_t add_general_sphere(const rt_gsphere &s, uint32_t mat = 0, bool reverse_orientation = false) {
        gspheres_.push_back(s);
        return push(RT_GSPHERE | (reverse_orientation ? RT_PRIM_REVERSE : 0u), uint32_t(gspheres_.size() - 1), mat, 0);
    }
    // Constant-density media (a Volume of the scene files): scattering with the phase function of
    // `mat` (RT_MAT_ISOTROPIC) inside a sphere, or inside a box with an optional instance transform
    uint32_t add_medium_sphere(std::array<float, 3> c, float r, float density, uint32_t mat = 0) {
        spheres_.push_back({{c[0], c[1], c[2]}, r});
        media_.push_back({RT_BOUNDARY_SPHERE, uint32_t(spheres_.size() - 1), 0u, density});
        return push(RT_MEDIUM, uint32_t(media_.size() - 1), mat, 0);
    }
    uint32_t add_medium_box(std::array<float, 3> lo, std::array<float, 3> hi, float density, uint32_t mat = 0,
                            uint32_t xf = 0) {
        boxes_.push_back({{lo[0], lo[1], lo[2]}, 0.f, {hi[0], hi[1], hi[2]}, 0.f});
        media_.push_back({RT_BOUNDARY_BOX, uint32_t(boxes_.size() - 1), xf, density});
        return push(RT_MEDIUM, uint32_t(media_.size() - 1), mat, 0);
    }
    void background(float r, float g, float b) { bg_ = {r, g, b}; }
    // LightSource "distant": unit vector towards the light, radiance (sampled by RT_RENDER_DIRECT_LIGHT)
    void distant_light(const float w[3], const float rgb[3]) {
        distant_.push_back({{w[0], w[1], w[2]}, {rgb[0], rgb[1], rgb[2]}, {0u, 0u}});
    }

    rt_scene_desc desc() const {
        rt_scene_desc d{};
        d.prims = prims_.data();
        d.spheres = spheres_.data();
        d.moving_spheres = moving_.data();
        d.tri_verts = tri_.data();
        d.rects = rects_.data();
        d.boxes = boxes_.data();
        d.gspheres = gspheres_.data();
        d.xforms = xforms_.data();
        d.media = media_.data();
        d.materials = materials_.data();
        d.textures = textures_.data();
        d.n_prims = uint32_t(prims_.size());
        d.n_spheres = uint32_t(spheres_.size());
        d.n_moving_spheres = uint32_t(moving_.size());
        d.n_tris = uint32_t(tri_.size() / 9);
        d.n_rects = uint32_t(rects_.size());
        d.n_boxes = uint32_t(boxes_.size());
        d.n_gspheres = uint32_t(gspheres_.size());
        d.n_xforms = uint32_t(xforms_.size());
        d.n_media = uint32_t(media_.size());
        d.n_materials = uint32_t(materials_.size());
        d.n_textures = uint32_t(textures_.size());
        d.background[0] = bg_[0], d.background[1] = bg_[1], d.background[2] = bg_[2];
        d.distant_lights = distant_.empty() ? nullptr : distant_.data();
        d.n_distant_lights = uint32_t(distant_.size());
        return d;
    }

  private:
    uint32_t push(uint32_t kind, uint32_t index, uint32_t mat, uint32_t xf) {
        prims_.push_back({kind, index, mat, xf});
        return uint32_t(prims_.size() - 1);
    }
    std::vector<rt_prim_ref> prims_;
    std::vector<rt_sphere> spheres_;
    std::vector<rt_moving_sphere> moving_;
    std::vector<float> tri_;
    std::vector<rt_rect> rects_;
    std::vector<rt_box> boxes_;
    std::vector<rt_gsphere> gspheres_;
    std::vector<rt_xform> xforms_;
    std::vector<rt_medium> media_;
    std::vector<rt_material> materials_;
    std::vector<rt_texture> textures_;
    std::array<float, 3> bg_{0.f, 0.f, 0.f};
    std::vector<rt_distant_light> distant_;
};

class Bvh {
  public:
    // BVH::new(primitives, primitives_in_node, split_method), bvh.rs:108-121
    Bvh(const SceneBuilder &prims, uint32_t primitives_in_node, SplitMethod split, int device = 0) : device_(device) {
        rt_scene_desc d = prims.desc();
        rt_build_opts o{primitives_in_node, uint32_t(split)};
        check(rt_scene_create(&d, &o, &scene_), "rt_scene_create");
    }
    Bvh(const Bvh &) = delete;
    Bvh &operator=(const Bvh &) = delete;
    Bvh(Bvh &&o) noexcept : scene_(o.scene_), device_(o.device_) { o.scene_ = nullptr; }
    ~Bvh() { rt_scene_destroy(scene_); }

    // Primitive::world_bound, bvh.rs:728-734
    Bounds3 world_bound() const {
        float b[6];
        check(rt_scene_world_bound(scene_, b), "rt_scene_world_bound");
        return {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
    }
    // Primitive::intersect, bvh.rs:736-799
    std::vector<rt_hit> intersect(const std::vector<rt_ray> &rays) const {
        std::vector<rt_hit> hits(rays.size());
        check(rt_intersect(scene_, rays.data(), rays.size(), hits.data(), device_), "rt_intersect");
        return hits;
    }
    std::optional<rt_hit> intersect(const rt_ray &ray) const {
        rt_hit h;
        check(rt_intersect(scene_, &ray, 1, &h, device_), "rt_intersect");
        return h.prim == RT_MISS ? std::nullopt : std::optional<rt_hit>(h);
    }
    // Primitive::does_intersect, bvh.rs:801-863
    std::vector<uint8_t> does_intersect(const std::vector<rt_ray> &rays) const {
        std::vector<uint8_t> out(rays.size());
        check(rt_occluded(scene_, rays.data(), rays.size(), out.data(), device_), "rt_occluded");
        return out;
    }
    bool does_intersect(const rt_ray &ray) const {
        uint8_t b;
        check(rt_occluded(scene_, &ray, 1, &b, device_), "rt_occluded");
        return b != 0;
    }
    // device-pointer forms, asynchronous on `stream` (a hipStream_t)
    void intersect_dev(const rt_ray *d_rays, size_t n, rt_hit *d_hits, void *stream = nullptr) const {
        check(rt_intersect_dev(scene_, d_rays, n, d_hits, device_, stream), "rt_intersect_dev");
    }
    void does_intersect_dev(const rt_ray *d_rays, size_t n, uint8_t *d_out, void *stream = nullptr) const {
        check(rt_occluded_dev(scene_, d_rays, n, d_out, device_, stream), "rt_occluded_dev");
    }
    std::vector<rt_linear_node> nodes() const {
        rt_scene_info info;
        check(rt_scene_get_info(scene_, &info), "rt_scene_get_info");
        std::vector<rt_linear_node> out(info.n_nodes);
        rt_scene_get_nodes(scene_, out.data(), out.size());
        return out;
    }
    rt_scene *handle() const { return scene_; }

  private:
    rt_scene *scene_ = nullptr;
    int device_;
};

}  // namespace rt355
