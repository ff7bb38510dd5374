// rt_api.cpp -- host runtime behind the C ABI of include/rt355.h: scene
// construction (BVH::new, shapes/src/bvh.rs:108-121), the one-time upload of
// the flattened tree and primitive tables to HBM, batch intersection entry
// points and the wavefront render loop.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_bvh.hpp"
#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                 \
    } while (0)

using rt::Aabb;

// ---- host arithmetic for primitive bounds (same expressions as the kernels) ----------------
struct H3 {
    float x, y, z;
};
H3 h_point(const float *m, H3 p) { // transform.rs:335-341
    return H3{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
float sel_lo(float a, float b) { return b < a ? b : a; }
float sel_hi(float a, float b) { return b > a ? b : a; }
Aabb grow(Aabb b, H3 p) {
    b.lo[0] = sel_lo(b.lo[0], p.x); b.lo[1] = sel_lo(b.lo[1], p.y); b.lo[2] = sel_lo(b.lo[2], p.z);
    b.hi[0] = sel_hi(b.hi[0], p.x); b.hi[1] = sel_hi(b.hi[1], p.y); b.hi[2] = sel_hi(b.hi[2], p.z);
    return b;
}
Aabb join(Aabb a, const Aabb &b) {
    for (int k = 0; k < 3; k++) {
        a.lo[k] = sel_lo(a.lo[k], b.lo[k]);
        a.hi[k] = sel_hi(a.hi[k], b.hi[k]);
    }
    return a;
}
Aabb xf_box(const float *m, const Aabb &b) { // Bounds3 * Transform, transform.rs:645-657
    Aabb r = rt::empty_box();
    for (int c = 0; c < 8; c++)
        r = grow(r, h_point(m, H3{(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1],
                                  (c & 4) ? b.hi[2] : b.lo[2]}));
    return r;
}
Aabb ball(const float c[3], float r) {
    r = fabsf(r);
    return Aabb{{c[0] - r, c[1] - r, c[2] - r}, {c[0] + r, c[1] + r, c[2] + r}};
}

template <class T>
std::vector<T> copy_in(const T *p, size_t n) {
    return p && n ? std::vector<T>(p, p + n) : std::vector<T>();
}

struct DeviceBuffers {
    std::vector<void *> allocs;
    rt::SceneDev dev{};
    rt::LaunchShape shape{};
    unsigned long long *counters = nullptr;
    uint8_t *bounce_scratch = nullptr; // per-wave parking and save areas of the bounce kernel
    uint64_t bytes = 0;
    // render workspace (grown on demand)
    std::mutex render_lock;
    uint32_t ws_paths = 0, ws_depth = 0;
    rt::Queue q[2]{};
    rt::HitQueue hits{}; // two-kernel bounces only (split_bounces)
    bool split_bounces = false; // a per-bounce launch is trace + shade (trees in global memory)
    float4 *radiance = nullptr;
    uint32_t *counts = nullptr;
    std::vector<void *> ws_allocs;
    // shadow rays of RT_RENDER_DIRECT_LIGHT (allocated when a frame first asks for them)
    rt::ShadowQueue shadow{};
    uint32_t *shadow_counts = nullptr;
    uint32_t shadow_paths = 0, shadow_depth = 0;
    std::vector<void *> shadow_allocs;
    // one block cursor per stream the trace entry points were called on: launches on a stream
    // run one after another, so they can share a word that is cleared in front of each
    std::mutex cursor_lock;
    std::map<hipStream_t, uint32_t *> cursors;
    // Bounce at which the last batch of a frame of this shape handed over to the one-launch
    // kernel: the next batch of such a frame switches there without asking the device
    // (queue sizes at a given bounce differ by a few per cent between batches and frames of the
    // same shape, and the threshold is not sharp).
    struct LoopHint {
        uint32_t width = 0, height = 0, paths = 0, depth = 0, shard_index = 0, shard_count = 0, bounce = 0;
        uint64_t below = 0;
        bool valid = false;
    } loop_hint;
    // tables or per-pixel samples of the camera sampler (grown on demand)
    void *sampler_mem = nullptr;
    size_t sampler_bytes = 0;
    // Test and experiment hooks (environment), read once when the scene is placed on the device:
    //   RT355_LOOP_FROM=<bounce>    the one-launch kernel takes over at this bounce
    //   RT355_LOOP_BELOW=<rays>     ... or once fewer rays than this are queued (0: never)
    //   RT355_LOOP_STEAL=0          its waves read their own sub-queue only
    //   RT355_LOOP_TURN_LIMIT=<n>   its spin guard
    //   RT355_TAIL_FILE=<path>      where a probe build writes its per-wave records
    //   RT355_SPLIT=<0|1>           per-bounce launches fused (0) or as trace + shade kernels (1)
    struct Hooks {
        uint32_t loop_from = 0xffffffffu;
        bool loop_below_set = false;
        uint64_t loop_below = 0;
        bool loop_steal = true;
        uint32_t turn_limit = rt::kLoopTurnLimit;
        std::string tail_file;
        int split = -1; // RT355_SPLIT=0/1: a bounce as one fused kernel / as trace + shade kernels (unset: by node source)
        void read() {
            if (const char *e = getenv("RT355_SPLIT")) split = atoi(e) != 0;
            if (const char *e = getenv("RT355_LOOP_FROM")) {
                loop_from = (uint32_t)atoi(e);
                loop_below_set = true;
                loop_below = 0;
            }
            if (const char *e = getenv("RT355_LOOP_BELOW")) {
                loop_below_set = true;
                loop_below = strtoull(e, nullptr, 10);
            }
            if (const char *e = getenv("RT355_LOOP_STEAL")) loop_steal = atoi(e) != 0;
            if (const char *e = getenv("RT355_LOOP_TURN_LIMIT")) turn_limit = (uint32_t)strtoul(e, nullptr, 10);
            if (const char *e = getenv("RT355_TAIL_FILE")) tail_file = e;
        }
    } hooks;
    // A frame enqueued without waiting for it (rt_render_dev with stats == NULL): the event that
    // marks its end, and the word of pinned host memory its last copy brings the one-launch
    // kernel's error flag to.  The next frame on this device waits for the event (the workspace
    // is shared) and fails if the flag is set.
    hipEvent_t pending_done = nullptr;
    uint32_t *pending_error = nullptr; // hipHostMalloc
    bool pending = false;
};

} // namespace

struct rt_scene {
    // host copies of what the description pointed to
    std::vector<rt_prim_ref> prims;
    std::vector<rt_sphere> spheres;
    std::vector<rt_moving_sphere> moving;
    std::vector<float> tri_verts, tri_uvs, tri_normals;
    std::vector<rt_rect> rects;
    std::vector<rt_box> boxes;
    std::vector<rt_gsphere> gspheres;
    std::vector<rt::GSphere> gsph_derived;
    std::vector<rt_xform> xforms;
    std::vector<rt_medium> media;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<std::vector<uint8_t>> image_data;
    std::vector<rt_image> images;
    std::vector<float> perlin_vec;
    std::vector<int32_t> perlin_perm;
    float background[3] = {0, 0, 0};
    std::vector<rt_distant_light> distant;
    // media that enclose the scene: kept out of the tree, tested once per ray before the
    // traversal (find_enclosing); caller's primitive indices in ascending order
    std::vector<uint32_t> enclosing;
    // balls (find_balls): a ray that starts inside one may be resolved by the ball's primitives alone
    struct Ball {
        uint32_t medium;    // the medium's primitive index
        int64_t sphere;     // the shell's primitive index, or -1
        float c[3], r, r_in2;
    };
    std::vector<Ball> balls;
    rt::BuiltTree tree;
    double build_ms = 0;
    mutable std::mutex lock;
    std::map<int, std::unique_ptr<DeviceBuffers>> devices;
};

namespace {

rt::GSphere derive_gsphere(const rt_gsphere &in) { // Sphere::new, shapes/src/sphere.rs:28-47
    rt::GSphere g{};
    memcpy(g.o2w, in.o2w, sizeof g.o2w);
    memcpy(g.w2o, in.w2o, sizeof g.w2o);
    float r = in.radius;
    float zlo = in.z_min < in.z_max ? in.z_min : in.z_max;
    float zhi = in.z_min < in.z_max ? in.z_max : in.z_min;
    g.radius = r;
    g.z_min = rt::clampf(zlo, -r, r);
    g.z_max = rt::clampf(zhi, -r, r);
    g.theta_min = rt::p_acos(rt::clampf(g.z_min / r, -1.0f, 1.0f));
    g.theta_max = rt::p_acos(rt::clampf(g.z_max / r, -1.0f, 1.0f));
    g.phi_max = rt::clampf(in.phi_max_deg, 0.0f, 360.0f) * 0.0174532924f;
    const float *m = in.o2w; // swaps_handedness, transform.rs:259-266
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
                m[2] * (m[4] * m[9] - m[5] * m[8]);
    g.flip = det < 0.0f ? 1u : 0u;
    return g;
}

// 0, or what is wrong with the description
const char *validate(const rt_scene_desc *d) {
    if (d->n_prims && !d->prims) return "prims is null";
    if (d->n_materials && !d->materials) return "materials is null";
    if (d->n_textures && !d->textures) return "textures is null";
    if (d->n_images && !d->images) return "images is null";
    if (d->n_distant_lights && !d->distant_lights) return "distant_lights is null";
    for (uint32_t i = 0; i < d->n_distant_lights; i++) {
        const float *w = d->distant_lights[i].w;
        const float len2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        if (!(len2 > 0.99f && len2 < 1.01f)) return "distant light direction is not a unit vector";
    }
    if (d->n_xforms && !d->xforms) return "xforms is null";
    if (d->n_media && (!d->media || (!d->spheres && !d->boxes))) return "media or their boundaries are null";
    for (uint32_t i = 0; i < d->n_images; i++)
        if (!d->images[i].rgb8 || !d->images[i].w || !d->images[i].h) return "image without pixels";
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const rt_prim_ref &p = d->prims[i];
        uint32_t kind = p.kind & 0xffu, lim = 0;
        const void *table = nullptr;
        switch (kind) {
        case RT_SPHERE: lim = d->n_spheres; table = d->spheres; break;
        case RT_MOVING_SPHERE: lim = d->n_moving_spheres; table = d->moving_spheres; break;
        case RT_TRIANGLE: lim = d->n_tris; table = d->tri_verts; break;
        case RT_RECT: lim = d->n_rects; table = d->rects; break;
        case RT_BOX: lim = d->n_boxes; table = d->boxes; break;
        case RT_GSPHERE: lim = d->n_gspheres; table = d->gspheres; break;
        case RT_MEDIUM: lim = d->n_media; table = d->media; break;
        default: return "unknown primitive kind";
        }
        if (p.index >= lim || !table) return "primitive index out of range";
        if (p.xform > d->n_xforms || p.xform > 0xffffu) return "primitive transform out of range";
        if (d->n_materials && p.material >= d->n_materials) return "primitive material out of range";
        if (kind == RT_RECT && d->rects[p.index].axis > 2) return "rectangle axis out of range";
        if (kind == RT_MEDIUM) {
            const rt_medium &m = d->media[p.index];
            if (m.boundary_kind == RT_BOUNDARY_SPHERE ? (m.boundary_index >= d->n_spheres || !d->spheres)
                                                      : (m.boundary_index >= d->n_boxes || !d->boxes))
                return "medium boundary out of range";
            if (m.xform > d->n_xforms) return "medium transform out of range";
        }
    }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->n_textures && d->materials[i].texture >= d->n_textures) return "material texture out of range";
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rt_texture &t = d->textures[i];
        if ((t.kind == RT_TEX_CHECKER || t.kind == RT_TEX_CHECKER_UV) &&
            (t.a >= d->n_textures || t.b >= d->n_textures))
            return "checker texture refers to a texture out of range";
        if (t.kind == RT_TEX_IMAGE && t.a >= d->n_images) return "texture image out of range";
        if (t.kind == RT_TEX_NOISE && (!d->perlin_vec || !d->perlin_perm)) return "noise texture without Perlin tables";
    }
    if (d->perlin_perm) // the kernels index perlin_vec[3 * (x ^ y ^ z)] with these
        for (int i = 0; i < 768; i++)
            if (d->perlin_perm[i] < 0 || d->perlin_perm[i] > 255) return "perlin_perm entry outside 0..255";
    return nullptr;
}

Aabb prim_bounds(const rt_scene &s, const rt_prim_ref &pr) {
    const uint32_t kind = pr.kind & 0xffu;
    switch (kind) {
    case RT_SPHERE: return ball(s.spheres[pr.index].c, s.spheres[pr.index].r);
    case RT_MOVING_SPHERE: {
        const rt_moving_sphere &m = s.moving[pr.index];
        return join(ball(m.c0, m.r), ball(m.c1, m.r));
    }
    case RT_TRIANGLE: { // triangle.rs:100-106
        const float *v = s.tri_verts.data() + 9 * (size_t)pr.index;
        Aabb b = rt::empty_box();
        for (int k = 0; k < 3; k++) b = grow(b, H3{v[3 * k], v[3 * k + 1], v[3 * k + 2]});
        return b;
    }
    case RT_RECT: {
        const rt_rect &r = s.rects[pr.index];
        int A = r.axis == 0 ? 1 : 0, B = r.axis == 2 ? 1 : 2;
        Aabb b;
        b.lo[r.axis] = r.k - 1e-4f; b.hi[r.axis] = r.k + 1e-4f;
        b.lo[A] = r.a0; b.hi[A] = r.a1;
        b.lo[B] = r.b0; b.hi[B] = r.b1;
        return pr.xform ? xf_box(s.xforms[pr.xform - 1].o2w, b) : b;
    }
    case RT_BOX: {
        const rt_box &x = s.boxes[pr.index];
        Aabb b{{x.lo[0], x.lo[1], x.lo[2]}, {x.hi[0], x.hi[1], x.hi[2]}};
        return pr.xform ? xf_box(s.xforms[pr.xform - 1].o2w, b) : b;
    }
    case RT_GSPHERE: { // sphere.rs:59-64 through shapes/src/lib.rs:51-53
        const rt::GSphere &g = s.gsph_derived[pr.index];
        Aabb ob{{-g.radius, -g.radius, g.z_min}, {g.radius, g.radius, g.z_max}};
        return xf_box(g.o2w, ob);
    }
    default: { // RT_MEDIUM
        const rt_medium &m = s.media[pr.index];
        if (m.boundary_kind == RT_BOUNDARY_SPHERE)
            return ball(s.spheres[m.boundary_index].c, s.spheres[m.boundary_index].r);
        const rt_box &x = s.boxes[m.boundary_index];
        Aabb b{{x.lo[0], x.lo[1], x.lo[2]}, {x.hi[0], x.hi[1], x.hi[2]}};
        return m.xform ? xf_box(s.xforms[m.xform - 1].o2w, b) : b;
    }
    }
}

// Media that enclose the scene.  The scene files put a thin fog around everything
// (test/week.toml:86-93: a Volume bounded by a sphere of radius 5000 about the origin).  As a
// primitive of the tree such a medium is a child of the root that every ray meets, tested
// inside the most divergent code of the kernels, and it stretches the root box to its own
// size.  A medium bounded by a sphere encloses the scene when the box around all primitives
// that are not sphere-bounded media lies strictly inside its sphere (all eight corners,
// distances in double precision).  The first kMaxEnclosing such media in primitive order are
// left out of the tree; every ray tests them once before its traversal starts, in that order,
// and a hit shortens the ray like any other.  The reference has no media, so this re-orders
// nothing of it.
std::vector<uint32_t> find_enclosing(const rt_scene &s, const std::vector<Aabb> &bounds) {
    std::vector<uint32_t> out;
    // test hook: the literal build, BVH::new over all primitives (shapes/src/bvh.rs:108-151);
    // hits, any-hit answers and frames must not depend on it
    if (getenv("RT355_NO_ENCLOSING")) return out;
    auto sphere_medium = [&](const rt_prim_ref &p) {
        return (p.kind & 0xffu) == RT_MEDIUM && s.media[p.index].boundary_kind == RT_BOUNDARY_SPHERE;
    };
    Aabb others = rt::empty_box();
    bool any = false;
    for (size_t i = 0; i < s.prims.size(); i++) {
        if (sphere_medium(s.prims[i])) continue;
        others = join(others, bounds[i]);
        any = true;
    }
    if (!any) return out;
    for (size_t i = 0; i < s.prims.size() && out.size() < rt::kMaxEnclosing; i++) {
        if (!sphere_medium(s.prims[i])) continue;
        const rt_sphere &sp = s.spheres[s.media[s.prims[i].index].boundary_index];
        bool inside = true;
        for (int c = 0; c < 8 && inside; c++) {
            const double dx = (double)((c & 1) ? others.hi[0] : others.lo[0]) - (double)sp.c[0];
            const double dy = (double)((c & 2) ? others.hi[1] : others.lo[1]) - (double)sp.c[1];
            const double dz = (double)((c & 4) ? others.hi[2] : others.lo[2]) - (double)sp.c[2];
            inside = dx * dx + dy * dy + dz * dz < (double)sp.r * (double)sp.r;
        }
        if (inside) out.push_back((uint32_t)i);
    }
    return out;
}

// Balls.  The scene files hold a glass sphere filled with fog (test/week.toml: a Volume bounded
// by the sphere that is also a dielectric primitive); a path that gets inside stays there for
// dozens of segments, each of which would walk the tree from the root only to find the two
// primitives it started in.  A ball is a medium bounded by a sphere (no instance transform, not
// one of the enclosing media), optionally with one sphere primitive of bit-identical centre and
// positive radius, such that no other primitive of the tree comes nearer to the sphere than a
// safety distance: for a sphere-like primitive the distance between the surfaces, for any other
// the distance from the ball's centre to the primitive's box minus the radius, must exceed
// 0.002 x (radius + half the diagonal of the other's box).  A ray that starts inside such a ball
// (|o - c|^2 < (0.998 r)^2) meets everything else only after it has left the ball, so its
// closest hit is decided by the ball's own primitives (ball_resolve in rt_kernels.hip).  The
// first kMaxBalls balls in primitive order are kept, out of the first 64 such media.  RT355_NO_BALLS=1 finds none (a test hook:
// results must not depend on it).
std::vector<rt_scene::Ball> find_balls(const rt_scene &s, const std::vector<Aabb> &bounds) {
    std::vector<rt_scene::Ball> out;
    if (getenv("RT355_NO_BALLS")) return out;
    auto is_enclosing = [&](size_t i) {
        for (uint32_t e : s.enclosing)
            if (e == i) return true;
        return false;
    };
    uint32_t examined = 0;
    for (size_t i = 0; i < s.prims.size() && out.size() < rt::kMaxBalls; i++) {
        const rt_prim_ref &p = s.prims[i];
        if ((p.kind & 0xffu) != RT_MEDIUM) continue;
        const rt_medium &m = s.media[p.index];
        if (m.boundary_kind != RT_BOUNDARY_SPHERE || m.xform != 0 || is_enclosing(i)) continue;
        const rt_sphere &sp = s.spheres[m.boundary_index];
        if (!(sp.r > 0.0f)) continue;
        if (++examined > 64) break; // every candidate is held against every primitive: keep that bounded
        rt_scene::Ball b;
        b.medium = (uint32_t)i;
        b.sphere = -1;
        memcpy(b.c, sp.c, sizeof b.c);
        b.r = sp.r;
        const double rin = (double)sp.r * 0.998;
        b.r_in2 = (float)(rin * rin);
        bool ok = true;
        for (size_t j = 0; j < s.prims.size() && ok; j++) {
            if (j == i || is_enclosing(j)) continue; // those are not in the tree; every ray tests them first anyway
            const rt_prim_ref &q = s.prims[j];
            const uint32_t kind = q.kind & 0xffu;
            const rt_sphere *other = nullptr;
            if (kind == RT_SPHERE) other = &s.spheres[q.index];
            else if (kind == RT_MEDIUM && s.media[q.index].boundary_kind == RT_BOUNDARY_SPHERE && s.media[q.index].xform == 0)
                other = &s.spheres[s.media[q.index].boundary_index];
            if (kind == RT_SPHERE && other->r > 0.0f && memcmp(other->c, sp.c, sizeof sp.c) == 0 &&
                memcmp(&other->r, &sp.r, sizeof sp.r) == 0 && b.sphere < 0) {
                b.sphere = (int64_t)j; // the shell
                continue;
            }
            const double hx = ((double)bounds[j].hi[0] - (double)bounds[j].lo[0]) * 0.5;
            const double hy = ((double)bounds[j].hi[1] - (double)bounds[j].lo[1]) * 0.5;
            const double hz = ((double)bounds[j].hi[2] - (double)bounds[j].lo[2]) * 0.5;
            const double safety = 0.002 * ((double)sp.r + sqrt(hx * hx + hy * hy + hz * hz));
            double gap;
            if (other) {
                const double dx = (double)other->c[0] - (double)sp.c[0], dy = (double)other->c[1] - (double)sp.c[1],
                             dz = (double)other->c[2] - (double)sp.c[2];
                gap = sqrt(dx * dx + dy * dy + dz * dz) - fabs((double)other->r) - (double)sp.r;
            } else {
                const double cx = sp.c[0], cy = sp.c[1], cz = sp.c[2];
                const double px = cx < bounds[j].lo[0] ? bounds[j].lo[0] : (cx > bounds[j].hi[0] ? bounds[j].hi[0] : cx);
                const double py = cy < bounds[j].lo[1] ? bounds[j].lo[1] : (cy > bounds[j].hi[1] ? bounds[j].hi[1] : cy);
                const double pz = cz < bounds[j].lo[2] ? bounds[j].lo[2] : (cz > bounds[j].hi[2] ? bounds[j].hi[2] : cz);
                gap = sqrt((px - cx) * (px - cx) + (py - cy) * (py - cy) + (pz - cz) * (pz - cz)) - (double)sp.r;
            }
            if (!(gap > safety)) ok = false;
        }
        if (ok) out.push_back(b);
    }
    return out;
}

template <class T>
int upload(DeviceBuffers &db, const std::vector<T> &host, const T **dev) {
    *dev = nullptr;
    if (host.empty()) return RT_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, host.size() * sizeof(T)));
    db.allocs.push_back(p);
    db.bytes += host.size() * sizeof(T);
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return RT_OK;
}

void release(DeviceBuffers &db) {
    for (void *p : db.allocs) (void)hipFree(p);
    for (void *p : db.ws_allocs) (void)hipFree(p);
    for (void *p : db.shadow_allocs) (void)hipFree(p);
    db.shadow_allocs.clear();
    for (auto &kv : db.cursors) (void)hipFree(kv.second);
    if (db.pending_done) (void)hipEventDestroy(db.pending_done);
    if (db.pending_error) (void)hipHostFree(db.pending_error);
    db.pending_done = nullptr;
    db.pending_error = nullptr;
    db.pending = false;
    db.allocs.clear();
    db.ws_allocs.clear();
    db.cursors.clear();
    if (db.sampler_mem) (void)hipFree(db.sampler_mem);
    db.sampler_mem = nullptr;
    db.sampler_bytes = 0;
}

// HIP events of one call; `timed` lists the pairs recorded around the bounce launches.
struct EventList {
    std::vector<hipEvent_t> all, timed;
    hipError_t make(hipEvent_t *e) {
        hipError_t rc = hipEventCreate(e);
        if (rc == hipSuccess) all.push_back(*e);
        return rc;
    }
    ~EventList() {
        for (hipEvent_t e : all) (void)hipEventDestroy(e);
    }
};

// Clears the stream's block cursor and launches behind it; both are enqueued under the lock so
// that two host threads sharing a stream cannot interleave them.
template <class Launch>
int with_cursor(DeviceBuffers &db, hipStream_t st, Launch launch) {
    std::lock_guard<std::mutex> g(db.cursor_lock);
    uint32_t *&c = db.cursors[st];
    if (!c) {
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, rt::kCursorBytes));
        c = static_cast<uint32_t *>(p);
    }
    HIP_TRY(hipMemsetAsync(c, 0, rt::kCursorBytes, st));
    HIP_TRY(launch(c));
    return RT_OK;
}

// Forgets a stream that is about to be destroyed (its work must have completed): the map is
// keyed by stream handle, and a caller that makes a stream per call would otherwise leave one
// cursor behind per handle value it ever saw.
void drop_cursor(DeviceBuffers &db, hipStream_t st) {
    std::lock_guard<std::mutex> g(db.cursor_lock);
    auto it = db.cursors.find(st);
    if (it == db.cursors.end()) return;
    (void)hipFree(it->second);
    db.cursors.erase(it);
}

// RT_OK, or why the sampler options are not accepted (Stratified::new / ZeroTwo::new arguments)
int sampler_check(const rt_render_opts &o) {
    if (o.sampler > RT_SAMPLER_STRATIFIED) return fail(RT_ERR_INVALID, "unknown sampler");
    if (o.sampler == RT_SAMPLER_SOBOL && std::max(o.width, o.height) > rt::kSobolMaxResolution)
        return fail(RT_ERR_LIMIT, "the Sobol' sampler's tables end at a resolution of 2^26 (data/src/sobol.rs)");
    if (o.sampler == RT_SAMPLER_ZERO_TWO && (o.spp & (o.spp - 1)))
        return fail(RT_ERR_INVALID,
                    "the (0,2) sampler needs a power-of-two spp (ZeroTwo::new rounds up: pass the rounded count)");
    if (o.sampler == RT_SAMPLER_STRATIFIED) {
        const uint32_t nx = o.sampler_arg & 0xffffu;
        if (!nx || o.spp % nx) return fail(RT_ERR_INVALID, "stratified sampler: sampler_arg's x strata must divide spp");
    }
    return RT_OK;
}

// Device memory the frame's camera sampler reads, filled on `stream`: the Sobol' tables, or
// every sample of n_slots pixels for the pixel samplers.  `mem`/`bytes` is a grow-only block.
int sampler_prepare(rt::FrameDev &F, uint32_t n_slots, bool by_rank, uint64_t first_pixel, void **mem, size_t *bytes,
                    hipStream_t stream) {
    size_t need = 0;
    if (F.sampler == RT_SAMPLER_SOBOL) need = sizeof(rt::SobolDev);
    else if (F.sampler == RT_SAMPLER_ZERO_TWO || F.sampler == RT_SAMPLER_STRATIFIED)
        need = (size_t)n_slots * F.spp * 5 * sizeof(float);
    if (!need) return RT_OK;
    if ((uint64_t)n_slots * F.spp >= (1ull << 32)) return fail(RT_ERR_LIMIT, "too many camera samples");
    if (need > *bytes) {
        if (*mem) (void)hipFree(*mem);
        *mem = nullptr;
        *bytes = 0;
        if (hipMalloc(mem, need) != hipSuccess) {
            *mem = nullptr;
            return fail(RT_ERR_NOMEM, "no device memory for %zu bytes of camera samples", need);
        }
        *bytes = need;
    }
    if (F.sampler == RT_SAMPLER_SOBOL) {
        rt::SobolDev t;
        F.sobol_log2 = rt::sobol_setup(F.width, F.height, t);
        HIP_TRY(hipMemcpyAsync(*mem, &t, sizeof t, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream)); // t leaves scope
        F.sobol = static_cast<const rt::SobolDev *>(*mem);
    } else {
        F.sample_stride = n_slots * F.spp;
        F.pixel_samples = static_cast<const float *>(*mem);
        HIP_TRY(rt::launch_pixel_samples(F, n_slots, by_rank, first_pixel, static_cast<float *>(*mem), stream));
    }
    return RT_OK;
}

// A light of RT_RENDER_DIRECT_LIGHT: a rectangle without instance transform whose material is a
// DiffuseLight with a solid colour.
bool is_sampled_light(const rt_scene &s, const rt_prim_ref &pr) {
    if ((pr.kind & 0xffu) != RT_RECT || pr.xform || pr.material >= s.materials.size()) return false;
    const rt_material &m = s.materials[pr.material];
    return m.kind == RT_MAT_DIFFUSE_LIGHT && m.texture < s.textures.size() &&
           s.textures[m.texture].kind == RT_TEX_SOLID;
}

// Record order for trees that are traversed from global memory.  Every interior step is a
// dependent 64-byte fetch at a random place of a table far larger than the caches; in the
// pre-order numbering only a node's first child is its neighbour.  Here records are grouped
// into clusters of up to `size` records that form a connected piece of the tree around the
// cluster's root: the piece grows by the fringe node with the largest box area (the one a
// ray that reached the piece is most likely to visit next), so the few steps a ray makes
// after entering a cluster tend to touch the same few cache lines and the same DRAM page.
// Clusters follow one another in depth-first order of their roots, larger boxes first, and
// start at multiples of `align` records (unused records are zero and never referenced).
// RT355_NODE_CLUSTER = "size[,align]" overrides the shape; size 1 keeps the pre-order.
// Returns the number of records of the table, unused ones included.
constexpr uint32_t kNodeCluster = 1, kNodeClusterAlign = 1;
uint32_t cluster_records(const std::vector<rt_linear_node> &nodes, std::vector<uint32_t> &record_of) {
    uint32_t size = kNodeCluster, align = kNodeClusterAlign;
    if (const char *e = getenv("RT355_NODE_CLUSTER")) {
        unsigned a = 0, b = 0;
        const int got = sscanf(e, "%u,%u", &a, &b);
        if (got >= 1 && a >= 1 && a <= 4096) size = a;
        align = got >= 2 && b >= 1 && b <= 4096 ? b : 1;
    }
    uint32_t n_pairs = 0;
    for (const rt_linear_node &n : nodes) n_pairs += n.primitive_count == 0;
    if (size <= 1) return n_pairs;
    auto area = [&](uint32_t i) {
        const rt_linear_node &n = nodes[i];
        const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Cand {
        float area;
        uint32_t node;
        bool operator<(const Cand &o) const { return area < o.area || (area == o.area && node > o.node); }
    };
    std::vector<uint32_t> roots; // cluster roots still to lay out (a stack: depth first)
    std::vector<Cand> fringe;    // heap
    uint32_t next = 0;
    if (nodes[0].primitive_count == 0) roots.push_back(0);
    while (!roots.empty()) {
        const uint32_t root = roots.back();
        roots.pop_back();
        next = (next + align - 1) / align * align;
        fringe.clear();
        fringe.push_back(Cand{area(root), root});
        uint32_t taken = 0;
        while (taken < size && !fringe.empty()) {
            std::pop_heap(fringe.begin(), fringe.end());
            const uint32_t x = fringe.back().node;
            fringe.pop_back();
            record_of[x] = next++;
            taken++;
            const uint32_t kids[2] = {x + 1, nodes[x].child_offset};
            for (uint32_t c : kids)
                if (nodes[c].primitive_count == 0) {
                    fringe.push_back(Cand{area(c), c});
                    std::push_heap(fringe.begin(), fringe.end());
                }
        }
        // what is left of the fringe starts clusters of its own: smallest box first onto the
        // stack, so that the largest is laid out next
        std::sort(fringe.begin(), fringe.end());
        for (const Cand &c : fringe) roots.push_back(c.node);
    }
    return next;
}

// Four-box records for trees that are traversed from global memory (SceneDev::quads): one
// 128-byte record per interior node that the walk reaches in an even number of steps from the
// root, with the boxes of that node's grandchildren -- or of a child itself where the child is a
// leaf.  Records are numbered depth first.  A random 128-byte read costs the memory system what a
// random 64-byte one does, so a ray that walks these records makes half as many dependent fetches.
// Skipping the children's own box tests is exact only if every interior node's box is the union
// of its children's boxes, bit for bit (the builders make it so; checked here): returns false
// otherwise, or when the tree is a single leaf, and the kernels then walk the two-box records.
// stack_need: the most entries a lane can have stacked (up to three per record on its way down).
bool build_quads(const std::vector<rt_linear_node> &nodes, std::vector<float4> &out, uint32_t &stack_need) {
    out.clear();
    stack_need = 0;
    if (nodes.empty() || nodes[0].primitive_count) return false;
    auto leaf = [&](uint32_t i) { return nodes[i].primitive_count != 0; };
    std::vector<uint32_t> quad_of(nodes.size(), 0xffffffffu), order, todo{0};
    while (!todo.empty()) {
        const uint32_t i = todo.back();
        todo.pop_back();
        quad_of[i] = (uint32_t)order.size();
        order.push_back(i);
        uint32_t next[4], n_next = 0;
        const uint32_t kids[2] = {i + 1, nodes[i].child_offset};
        for (uint32_t c : kids) {
            if (leaf(c)) continue;
            const uint32_t grand[2] = {c + 1, nodes[c].child_offset};
            for (uint32_t g : grand)
                if (!leaf(g)) next[n_next++] = g;
        }
        while (n_next) todo.push_back(next[--n_next]); // the first of them is numbered next
    }
    if (order.size() >= (1u << 31)) return false;
    auto bits = [](uint32_t v) {
        float f;
        memcpy(&f, &v, 4);
        return f;
    };
    out.assign(8 * order.size(), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<uint32_t> need(order.size(), 0);
    for (size_t r = order.size(); r-- > 0;) { // children's records have larger numbers
        const uint32_t i = order[r];
        const uint32_t kids[2] = {i + 1, nodes[i].child_offset};
        const rt_linear_node *box[4] = {nullptr, nullptr, nullptr, nullptr};
        uint32_t ref[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint32_t axes = nodes[i].axis & 3u, entries = 0, deepest = 0;
        for (int side = 0; side < 2; side++) {
            const uint32_t c = kids[side];
            if (leaf(c)) {
                box[2 * side] = &nodes[c];
                ref[2 * side] = rt::kLeafRef | nodes[c].child_offset;
                entries++;
                continue;
            }
            const uint32_t grand[2] = {c + 1, nodes[c].child_offset};
            for (int k = 0; k < 3; k++) { // the child's box must be the union of its children's
                const float lo = std::min(nodes[grand[0]].lo[k], nodes[grand[1]].lo[k]);
                const float hi = std::max(nodes[grand[0]].hi[k], nodes[grand[1]].hi[k]);
                if (!(lo == nodes[c].lo[k] && hi == nodes[c].hi[k])) return false;
            }
            axes |= (nodes[c].axis & 3u) << (2 + 2 * side);
            for (int k = 0; k < 2; k++) {
                const uint32_t g = grand[k];
                box[2 * side + k] = &nodes[g];
                ref[2 * side + k] = leaf(g) ? (rt::kLeafRef | nodes[g].child_offset) : quad_of[g];
                if (!leaf(g)) deepest = std::max(deepest, need[quad_of[g]]);
                entries++;
            }
        }
        need[r] = entries - 1 + deepest;
        float4 *q = out.data() + 8 * r;
        const float meta[4] = {bits(axes), 0.0f, 0.0f, 0.0f};
        for (int e = 0; e < 4; e++) {
            const rt_linear_node *b = box[e];
            // references ride in the first four quarters, the axes in the fifth
            q[2 * e] = make_float4(b ? b->lo[0] : 0.0f, b ? b->lo[1] : 0.0f, b ? b->lo[2] : 0.0f, 0.0f);
            q[2 * e + 1] = make_float4(b ? b->hi[0] : 0.0f, b ? b->hi[1] : 0.0f, b ? b->hi[2] : 0.0f, 0.0f);
        }
        for (int e = 0; e < 4; e++) q[e].w = bits(ref[e]);
        q[4].w = meta[0];
    }
    stack_need = need[0];
    return true;
}

// Lays the scene out for the kernels: one 64-byte record per interior node with
// the boxes of both children, the primitive list in leaf order, and every typed
// table re-packed in the order the leaves reference it so that a leaf's
// primitives sit next to each other.
int build_device(rt_scene &s, int device, DeviceBuffers &db) {
    HIP_TRY(hipSetDevice(device));
    db.hooks.read();
    const auto &nodes = s.tree.nodes;
    // records are numbered in the pre-order of the flattened tree (bvh.rs:661-691)
    std::vector<uint32_t> record_of(nodes.size(), 0);
    uint32_t n_pairs = 0;
    for (size_t i = 0; i < nodes.size(); i++)
        if (nodes[i].primitive_count == 0) record_of[i] = n_pairs++;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    // A lane stacks at most one far child per interior level, i.e. max_depth - 1 entries; the
    // first stack_lds of them live in LDS, the rest in private memory.  If the records and at
    // least kMinLds stack levels of a 1024-thread workgroup fit in one CU's LDS the tree is
    // traversed from LDS (one workgroup per CU, 16-bit references); otherwise 256-thread
    // workgroups read records from global memory, with as many stack levels in LDS as still
    // leave room for kGlobalWaves workgroups per CU.
    const uint32_t slots = s.tree.max_depth > 1 ? s.tree.max_depth - 1 : 1;
    const size_t lds_cu = std::min<size_t>(prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlockOptin);
    constexpr uint32_t kMinLds = 8;
    const char *force = getenv("RT355_NODES"); // "global": override for experiments
    bool lds_nodes = false;
    uint32_t stack_lds = 0;
    constexpr size_t kDump = 256; // slack (rt::lds_bytes adds it)
    if (n_pairs > 0 && n_pairs < 0x8000u && s.prims.size() < 0x8000u && (size_t)n_pairs * 64 + kDump < lds_cu) {
        const size_t room = lds_cu - (size_t)n_pairs * 64 - kDump;
        stack_lds = (uint32_t)std::min<size_t>(slots, room / (1024 * 6));
        lds_nodes = stack_lds >= std::min(slots, kMinLds);
    }
    if (force && !strcmp(force, "global")) lds_nodes = false;
    if (!lds_nodes) stack_lds = (uint32_t)std::min<size_t>(slots, (lds_cu / rt::kGlobalWaves - 512) / (256 * 8));
    // Trees that stay in global memory: records are laid out in clusters instead (see
    // cluster_records).  Record numbers are internal to the device tables, so the visiting
    // order, the results and rt_scene_get_nodes do not change.
    uint32_t n_records = n_pairs;
    if (!lds_nodes && n_pairs > 1) n_records = cluster_records(nodes, record_of);
    // primitives the plain shader does not cover by kind
    auto rare_prim = [&](const rt_prim_ref &pr) {
        const uint32_t kind = pr.kind & 0xffu;
        return pr.xform != 0 ||
               !(kind == RT_SPHERE || kind == RT_TRIANGLE || kind == RT_RECT || kind == RT_BOX || kind == RT_MEDIUM);
    };
    auto ref_of = [&](size_t i) {
        return nodes[i].primitive_count ? (rt::kLeafRef | nodes[i].child_offset) : record_of[i];
    };
    // which hits the plain shader covers: see kRareShade
    auto solid = [&](uint32_t t) { return t < s.textures.size() && s.textures[t].kind == RT_TEX_SOLID; };
    auto rare_shade = [&](const rt_prim_ref &pr) {
        if (rare_prim(pr)) return true;
        if (pr.material >= s.materials.size()) return true; // intersection-only scene: never shaded
        const rt_material &m = s.materials[pr.material];
        if (m.kind == RT_MAT_DIELECTRIC) return false; // no texture
        if (m.texture >= s.textures.size()) return true;
        if (!(m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METALLIC || m.kind == RT_MAT_DIFFUSE_LIGHT ||
              m.kind == RT_MAT_ISOTROPIC))
            return true;
        const rt_texture &t = s.textures[m.texture];
        return !(t.kind == RT_TEX_SOLID || (t.kind == RT_TEX_CHECKER && solid(t.a) && solid(t.b)));
    };
    auto bits = [](uint32_t v) {
        float f;
        memcpy(&f, &v, 4);
        return f;
    };
    std::vector<float4> dn(4 * (size_t)n_records, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (size_t i = 0; i < nodes.size(); i++) {
        const rt_linear_node &n = nodes[i];
        if (n.primitive_count) continue;
        const rt_linear_node &c0 = nodes[i + 1], &c1 = nodes[n.child_offset];
        float4 *r = dn.data() + 4 * (size_t)record_of[i];
        r[0] = make_float4(c0.lo[0], c0.lo[1], c0.lo[2], bits(ref_of(i + 1)));
        r[1] = make_float4(c0.hi[0], c0.hi[1], c0.hi[2], bits(ref_of(n.child_offset)));
        r[2] = make_float4(c1.lo[0], c1.lo[1], c1.lo[2], bits(n.axis));
        r[3] = make_float4(c1.hi[0], c1.hi[1], c1.hi[2], 0.0f);
    }
    std::vector<rt::LeafPrim> leaf(s.prims.size());
    std::vector<float4> spheres, tris;
    std::vector<float> uvs, normals;
    std::vector<rt::MSphere> msph;
    std::vector<rt_rect> rects;
    std::vector<rt_box> boxes;
    std::vector<rt::GSphere> gsph;
    std::vector<rt::Medium> media;
    auto push_sphere = [&](uint32_t idx) {
        const rt_sphere &sp = s.spheres[idx];
        spheres.push_back(make_float4(sp.c[0], sp.c[1], sp.c[2], sp.r));
        return (uint32_t)spheres.size() - 1;
    };
    auto push_box = [&](uint32_t idx) {
        boxes.push_back(s.boxes[idx]);
        return (uint32_t)boxes.size() - 1;
    };
    const size_t n_tree_prims = s.tree.order.size();
    for (size_t k = 0; k < leaf.size(); k++) {
        // the tree's primitives in leaf order, then the media that enclose the scene
        const uint32_t id = k < n_tree_prims ? s.tree.order[k] : s.enclosing[k - n_tree_prims];
        const rt_prim_ref &pr = s.prims[id];
        const uint32_t kind = pr.kind & 0xffu;
        rt::LeafPrim lp{(pr.kind & 0x1ffu) | (pr.xform << 16) | (rare_shade(pr) ? rt::kRareShade : 0u), 0, id,
                        pr.material, make_float4(0, 0, 0, 0)};
        switch (kind) {
        case RT_SPHERE:
            lp.index = push_sphere(pr.index);
            lp.geom = spheres.back();
            break;
        case RT_MOVING_SPHERE: {
            const rt_moving_sphere &m = s.moving[pr.index];
            rt::MSphere o{};
            float dt = m.time1 - m.time0;
            for (int c = 0; c < 3; c++) {
                o.c0[c] = m.c0[c];
                o.dcdt[c] = (m.c1[c] - m.c0[c]) / dt;
            }
            o.r = m.r;
            o.time0 = m.time0;
            msph.push_back(o);
            lp.index = (uint32_t)msph.size() - 1;
            break;
        }
        case RT_TRIANGLE: {
            const float *v = s.tri_verts.data() + 9 * (size_t)pr.index;
            for (int c = 0; c < 3; c++) tris.push_back(make_float4(v[3 * c], v[3 * c + 1], v[3 * c + 2], 0.0f));
            if (!s.tri_uvs.empty())
                uvs.insert(uvs.end(), s.tri_uvs.begin() + 6 * (size_t)pr.index,
                           s.tri_uvs.begin() + 6 * (size_t)pr.index + 6);
            if (!s.tri_normals.empty())
                normals.insert(normals.end(), s.tri_normals.begin() + 9 * (size_t)pr.index,
                               s.tri_normals.begin() + 9 * (size_t)pr.index + 9);
            lp.index = (uint32_t)(tris.size() / 3) - 1;
            break;
        }
        case RT_RECT:
            rects.push_back(s.rects[pr.index]);
            lp.index = (uint32_t)rects.size() - 1;
            break;
        case RT_BOX: lp.index = push_box(pr.index); break;
        case RT_GSPHERE:
            gsph.push_back(s.gsph_derived[pr.index]);
            lp.index = (uint32_t)gsph.size() - 1;
            break;
        default: { // RT_MEDIUM
            const rt_medium &m = s.media[pr.index];
            rt::Medium o{m.boundary_kind, 0, m.xform, -1.0f / m.density};
            if (m.boundary_kind == RT_BOUNDARY_SPHERE) {
                o.boundary_index = push_sphere(m.boundary_index);
                lp.geom = spheres.back();
            } else {
                o.boundary_index = push_box(m.boundary_index);
            }
            media.push_back(o);
            lp.index = (uint32_t)media.size() - 1;
            if (m.boundary_kind == RT_BOUNDARY_SPHERE) {
                lp.kind |= rt::kMediumInSphere;
                memcpy(&lp.index, &o.neg_inv_density, sizeof(float));
            }
            break;
        }
        }
        leaf[k] = lp;
    }
    for (const rt_linear_node &n : nodes)
        if (n.primitive_count) leaf[n.child_offset + n.primitive_count - 1].kind |= rt::kLastInLeaf;
    // the lights RT_RENDER_DIRECT_LIGHT samples (is_sampled_light, the rule the table below is built by)
    for (rt::LeafPrim &lp : leaf)
        if (is_sampled_light(s, s.prims[lp.prim])) lp.kind |= rt::kSampledLight;
    // What the plain shader needs of every material, in one record: its kind and parameters and
    // its texture when that is a solid colour or a checker of two solid colours (the hits of any
    // other material go to the general shader, which follows the references).
    std::vector<rt::PlainShade> plain(s.materials.size());
    for (size_t k = 0; k < plain.size(); k++) {
        rt::PlainShade &ps = plain[k];
        memset(&ps, 0, sizeof ps);
        const rt_material &m = s.materials[k];
        ps.material_kind = m.kind;
        ps.fuzz = m.fuzz;
        ps.ior = m.ior;
        if (m.kind == RT_MAT_DIELECTRIC || m.texture >= s.textures.size()) continue; // no texture read
        const rt_texture &t = s.textures[m.texture];
        if (t.kind == RT_TEX_CHECKER) {
            if (t.a >= s.textures.size() || t.b >= s.textures.size()) continue;
            ps.checker = 1;
            ps.scale = t.scale;
            memcpy(ps.a, s.textures[t.a].rgb, sizeof ps.a);
            memcpy(ps.b, s.textures[t.b].rgb, sizeof ps.b);
        } else {
            memcpy(ps.a, t.rgb, sizeof ps.a);
        }
    }
    rt::SceneDev &D = db.dev;
    int rc;
    // trees in LDS (at most 32 k primitives): the record once per leaf entry, in leaf order (RT355_PLAIN_PER_ENTRY=0/1 overrides)
    {
        const char *hook = getenv("RT355_PLAIN_PER_ENTRY");
        D.plain_per_entry = (hook ? atoi(hook) != 0 : lds_nodes) ? 1u : 0u;
        if (D.plain_per_entry) {
            std::vector<rt::PlainShade> per_entry(leaf.size());
            for (size_t k = 0; k < leaf.size(); k++) {
                memset(&per_entry[k], 0, sizeof per_entry[k]);
                if (leaf[k].material < plain.size()) per_entry[k] = plain[leaf[k].material];
            }
            plain.swap(per_entry);
        }
    }
    if ((rc = upload(db, dn, &D.nodes))) return rc;
    // four-box records beside them for trees in global memory.  RT355_QUADS=0: none, 2: walked by every
    // launch without counters instead of by the later bounces only (LaunchShape::quads; a test hook)
    std::vector<float4> dq;
    uint32_t quad_need = 0;
    const char *quads_hook = getenv("RT355_QUADS");
    const uint32_t quads_mode = quads_hook ? (uint32_t)atoi(quads_hook) : 1u;
    bool quads = !lds_nodes && quads_mode != 0 && build_quads(nodes, dq, quad_need);
    if (quads && quad_need > rt::kSpillLevels) quads = false; // deeper than a lane's stack: two-box records only
    D.quads = nullptr;
    D.root_quad = 0;
    if (quads && (rc = upload(db, dq, &D.quads))) return rc;
    dq = std::vector<float4>();
    if ((rc = upload(db, leaf, &D.leaf))) return rc;
    if ((rc = upload(db, plain, &D.plain))) return rc;
    if ((rc = upload(db, spheres, &D.spheres))) return rc;
    if ((rc = upload(db, msph, &D.msph))) return rc;
    if ((rc = upload(db, tris, &D.tris))) return rc;
    if ((rc = upload(db, uvs, &D.tri_uvs))) return rc;
    if ((rc = upload(db, normals, &D.tri_normals))) return rc;
    if ((rc = upload(db, rects, &D.rects))) return rc;
    if ((rc = upload(db, boxes, &D.boxes))) return rc;
    if ((rc = upload(db, gsph, &D.gsph))) return rc;
    if ((rc = upload(db, s.xforms, &D.xforms))) return rc;
    if ((rc = upload(db, media, &D.media))) return rc;
    if ((rc = upload(db, s.materials, &D.materials))) return rc;
    if ((rc = upload(db, s.textures, &D.textures))) return rc;
    if ((rc = upload(db, s.perlin_vec, &D.perlin_vec))) return rc;
    if ((rc = upload(db, s.perlin_perm, &D.perlin_perm))) return rc;
    // lights of RT_RENDER_DIRECT_LIGHT: rectangles without instance transform whose material is a
    // DiffuseLight with a solid colour, in primitive order
    std::vector<rt::LightDev> lights;
    for (const rt_prim_ref &pr : s.prims) {
        if (!is_sampled_light(s, pr)) continue;
        const rt_material &m = s.materials[pr.material];
        const rt_rect &r = s.rects[pr.index];
        const rt_texture &t = s.textures[m.texture];
        lights.push_back(rt::LightDev{r.axis, r.k, r.a0, r.a1, r.b0, r.b1, 0.0f, {t.rgb[0], t.rgb[1], t.rgb[2]}, {0, 0}});
    }
    const size_t n_rect_lights = lights.size(), n_all_lights = n_rect_lights + s.distant.size();
    for (rt::LightDev &l : lights) {
        const float area = (l.a1 - l.a0) * (l.b1 - l.b0);
        l.scale = area * (float)n_all_lights * 0.318309886f;
    }
    // Distant lights follow (axis 3): direction in a0, a1, b0; k = how far along it the point
    // lies that the shadow ray is aimed at -- twice the radius of the sphere around the scene's
    // box, so that it is outside the scene from wherever the ray starts (pbrt's distant light;
    // the oracle forms the same number from the same box).
    if (!s.distant.empty()) {
        float wb[6];
        if (rt_scene_world_bound(&s, wb) != RT_OK) return RT_ERR_INVALID;
        const float hx = wb[3] - (wb[0] + wb[3]) * 0.5f, hy = wb[4] - (wb[1] + wb[4]) * 0.5f,
                    hz = wb[5] - (wb[2] + wb[5]) * 0.5f;
        const float reach = nodes.empty() && s.enclosing.empty() ? 0.0f : 2.0f * sqrtf((hx * hx + hy * hy) + hz * hz);
        for (const rt_distant_light &dl : s.distant)
            lights.push_back(rt::LightDev{3u, reach, dl.w[0], dl.w[1], dl.w[2], 0.0f, (float)n_all_lights * 0.318309886f,
                                          {dl.rgb[0], dl.rgb[1], dl.rgb[2]}, {0, 0}});
    }
    if ((rc = upload(db, lights, &D.lights))) return rc;
    D.n_lights = (uint32_t)lights.size();
    std::vector<rt::ImageDev> imgs;
    for (size_t i = 0; i < s.images.size(); i++) {
        const uint8_t *px = nullptr;
        if ((rc = upload(db, s.image_data[i], &px))) return rc;
        imgs.push_back(rt::ImageDev{px, s.images[i].w, s.images[i].h});
    }
    if ((rc = upload(db, imgs, &D.images))) return rc;
    D.n_nodes = (uint32_t)nodes.size();
    D.n_prims = (uint32_t)leaf.size();
    D.n_tree_prims = (uint32_t)n_tree_prims;
    D.n_enclosing = (uint32_t)s.enclosing.size();
    // balls: the leaf slots of their primitives, in leaf order
    D.n_balls = 0;
    for (const rt_scene::Ball &b : s.balls) {
        int64_t at_medium = -1, at_sphere = -1;
        for (size_t k = 0; k < leaf.size(); k++) {
            if (leaf[k].prim == b.medium) at_medium = (int64_t)k;
            if (b.sphere >= 0 && leaf[k].prim == (uint32_t)b.sphere) at_sphere = (int64_t)k;
        }
        if (at_medium < 0 || (b.sphere >= 0 && at_sphere < 0)) continue;
        rt::BallDev &o = D.balls[D.n_balls++];
        memcpy(o.c, b.c, sizeof o.c);
        o.r = b.r;
        o.r_in2 = b.r_in2;
        auto best_of = [&](int64_t k) {
            return (uint32_t)k | ((leaf[(size_t)k].kind & rt::kRareShade) ? 0x80000000u : 0u);
        };
        if (b.sphere < 0) {
            o.n = 1;
            o.best[0] = o.best[1] = best_of(at_medium);
            o.sphere = -1;
            o.medium = 0;
        } else {
            o.n = 2;
            o.sphere = at_sphere < at_medium ? 0 : 1;
            o.medium = 1u - (uint32_t)o.sphere;
            o.best[o.sphere] = best_of(at_sphere);
            o.best[o.medium] = best_of(at_medium);
        }
    }
    D.enclosing_rare = 0;
    for (size_t e = 0; e < s.enclosing.size(); e++)
        if (leaf[n_tree_prims + e].kind & rt::kRareShade) D.enclosing_rare |= 1u << e;
    D.n_pairs = n_pairs;
    D.root_ref = nodes.empty() ? 0u : ref_of(0);
    for (int c = 0; c < 3; c++) {
        D.root_lo[c] = nodes.empty() ? 0.0f : nodes[0].lo[c];
        D.root_hi[c] = nodes.empty() ? 0.0f : nodes[0].hi[c];
    }
    memcpy(D.background, s.background, sizeof D.background);

    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, rt::kCounterWords * sizeof(unsigned long long)));
    db.allocs.push_back(p);
    db.counters = static_cast<unsigned long long *>(p);

    // Persistent grid (the node source and the LDS share of the stacks were chosen above).
    // test hook: fewer LDS levels, so that the private part of the stack is exercised
    if (const char *cap = getenv("RT355_STACK_LDS")) {
        const uint32_t v = (uint32_t)atoi(cap);
        if (v >= 1 && v < stack_lds) stack_lds = v;
    }
    if (slots - stack_lds > rt::kSpillLevels) return fail(RT_ERR_LIMIT, "traversal stack too deep for this device");
    D.stack_lds = stack_lds;
    db.shape.slots = slots;
    db.shape.lds_nodes = lds_nodes;
    db.shape.quads = quads ? (quads_mode >= 2 ? 2u : 1u) : 0u;
    // trees in global memory: a bounce is a trace launch and a shade launch (measured on the 11 M
    // primitive scene: +8 %); trees in LDS: one fused launch
    db.split_bounces = db.hooks.split >= 0 ? db.hooks.split != 0 : !lds_nodes;
    // The bounce kernel binds wave w to sub-queue w % kShards: the wave count of the grid
    // (16 waves per 1024-thread workgroup, 4 per 256-thread one) is kept a multiple of kShards.
    if (lds_nodes) {
        db.shape.grid = std::max(2u, (uint32_t)prop.multiProcessorCount / 2 * 2);
    } else {
        // as many persistent workgroups as are resident at once: one more per CU would only
        // start when another ends, and its rays would be the frame's tail
        uint32_t per_cu = 1;
        HIP_TRY(rt::resident_workgroups(rt::lds_bytes(false, n_pairs, stack_lds), &per_cu));
        if (per_cu > rt::kGlobalWaves) per_cu = rt::kGlobalWaves;
        db.shape.grid = std::max(8u, (uint32_t)prop.multiProcessorCount * per_cu / 8 * 8);
    }
    static_assert(rt::kShards == 32, "grid rounding above assumes 32 sub-queues");
    HIP_TRY(rt::static_cursor_setup(device)); // for rt_intersect_dev_noalloc, which must not set anything up
    const size_t scratch_bytes = (size_t)rt::bounce_waves(db.shape) * std::max(rt::kWaveScratchBytes, rt::kWaveLoopBytes);
    HIP_TRY(hipMalloc(&p, scratch_bytes));
    db.allocs.push_back(p);
    db.bounce_scratch = static_cast<uint8_t *>(p);
    db.bytes += scratch_bytes;
    return RT_OK;
}

int get_device(rt_scene *s, int device, DeviceBuffers **out) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device available; librt355 has no CPU path");
    if (device < 0 || device >= count)
        return fail(RT_ERR_NO_DEVICE, "device %d out of range (0..%d); librt355 has no CPU path", device,
                    count - 1);
    std::lock_guard<std::mutex> g(s->lock);
    auto it = s->devices.find(device);
    if (it == s->devices.end()) {
        auto db = std::make_unique<DeviceBuffers>();
        int rc = build_device(*s, device, *db);
        if (rc != RT_OK) {
            release(*db);
            return rc;
        }
        it = s->devices.emplace(device, std::move(db)).first;
    }
    HIP_TRY(hipSetDevice(device));
    *out = it->second.get();
    return RT_OK;
}

uint32_t stack_slots(const rt_scene *s) { return s->tree.max_depth > 1 ? s->tree.max_depth - 1 : 1; }

constexpr size_t kChunk = 1u << 28; // rays per launch (32-bit indices in the kernels)

size_t count_words(uint32_t depth) { return (size_t)(depth + 2) * rt::kShards * rt::kCountStride; }

int grow_workspace(DeviceBuffers &db, uint32_t paths, uint32_t depth) {
    if (paths <= db.ws_paths && depth <= db.ws_depth) return RT_OK;
    for (void *p : db.ws_allocs) (void)hipFree(p);
    db.ws_allocs.clear();
    db.ws_paths = db.ws_depth = 0;
    auto take = [&](size_t bytes, void **out) -> int {
        HIP_TRY(hipMalloc(out, bytes));
        db.ws_allocs.push_back(*out);
        return RT_OK;
    };
    int rc;
    // sub-queue capacity: its share of the 64-path chunks, rounded up
    const uint32_t chunks = (paths + 63) / 64;
    const uint32_t cap = (chunks + rt::kShards - 1) / rt::kShards * 64;
    const size_t entries = (size_t)cap * rt::kShards;
    for (int k = 0; k < 2; k++) {
        if ((rc = take(entries * sizeof(float4), (void **)&db.q[k].a))) return rc;
        if ((rc = take(entries * sizeof(float4), (void **)&db.q[k].b))) return rc;
        if ((rc = take(entries * sizeof(float4), (void **)&db.q[k].c))) return rc;
        if ((rc = take(entries * sizeof(uint2), (void **)&db.q[k].d))) return rc;
        db.q[k].e = nullptr;
        if ((db.dev.n_enclosing || db.dev.n_balls) && (rc = take(entries * sizeof(uint2), (void **)&db.q[k].e))) return rc;
        db.q[k].cap = cap;
    }
    db.hits = rt::HitQueue{};
    if (db.split_bounces) {
        if ((rc = take(entries * sizeof(float4), (void **)&db.hits.a))) return rc;
        if ((rc = take(entries * sizeof(float), (void **)&db.hits.b))) return rc;
    }
    if ((rc = take((size_t)paths * sizeof(float4), (void **)&db.radiance))) return rc;
    if ((rc = take(count_words(depth) * sizeof(uint32_t), (void **)&db.counts))) return rc;
    db.ws_paths = paths;
    db.ws_depth = depth;
    return RT_OK;
}

int grow_shadow_workspace(DeviceBuffers &db, uint32_t paths, uint32_t depth) {
    if (paths <= db.shadow_paths && depth <= db.shadow_depth) return RT_OK;
    for (void *p : db.shadow_allocs) (void)hipFree(p);
    db.shadow_allocs.clear();
    db.shadow_paths = db.shadow_depth = 0;
    auto take = [&](size_t bytes, void **out) -> int {
        HIP_TRY(hipMalloc(out, bytes));
        db.shadow_allocs.push_back(*out);
        return RT_OK;
    };
    int rc;
    const uint32_t chunks = (paths + 63) / 64;
    const uint32_t cap = (chunks + rt::kShards - 1) / rt::kShards * 64; // as the bounce queues'
    const size_t entries = (size_t)cap * rt::kShards;
    if ((rc = take(entries * sizeof(float4), (void **)&db.shadow.a))) return rc;
    if ((rc = take(entries * sizeof(float4), (void **)&db.shadow.b))) return rc;
    if ((rc = take(entries * sizeof(float4), (void **)&db.shadow.c))) return rc;
    db.shadow.cap = cap;
    if ((rc = take(count_words(depth) * sizeof(uint32_t), (void **)&db.shadow_counts))) return rc;
    db.shadow_paths = paths;
    db.shadow_depth = depth;
    return RT_OK;
}

} // namespace

extern "C" {

const char *rt_last_error(void) { return g_error; }

int rt_abi_sizes(uint32_t *out, size_t cap) {
    const uint32_t sizes[] = {sizeof(rt_ray),        sizeof(rt_hit),         sizeof(rt_prim_ref),
                              sizeof(rt_sphere),     sizeof(rt_moving_sphere), sizeof(rt_rect),
                              sizeof(rt_box),        sizeof(rt_gsphere),     sizeof(rt_xform),
                              sizeof(rt_medium),     sizeof(rt_material),    sizeof(rt_texture),
                              sizeof(rt_image),      sizeof(rt_scene_desc),  sizeof(rt_build_opts),
                              sizeof(rt_linear_node), sizeof(rt_scene_info), sizeof(rt_camera),
                              sizeof(rt_filter),     sizeof(rt_render_opts), sizeof(rt_stats),
                              sizeof(rt_distant_light)};
    size_t n = sizeof sizes / sizeof sizes[0];
    if (n > cap) n = cap;
    if (out) memcpy(out, sizes, n * sizeof(uint32_t));
    return (int)n;
}

int rt_scene_create(const rt_scene_desc *d, const rt_build_opts *opts, rt_scene **out) {
    if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (const char *why = validate(d)) return fail(RT_ERR_INVALID, "malformed scene description: %s", why);
    try {
        auto s = std::make_unique<rt_scene>();
        s->prims = copy_in(d->prims, d->n_prims);
        s->spheres = copy_in(d->spheres, d->n_spheres);
        s->moving = copy_in(d->moving_spheres, d->n_moving_spheres);
        s->tri_verts = copy_in(d->tri_verts, 9 * (size_t)d->n_tris);
        s->tri_uvs = copy_in(d->tri_uvs, 6 * (size_t)d->n_tris);
        s->tri_normals = copy_in(d->tri_normals, 9 * (size_t)d->n_tris);
        s->rects = copy_in(d->rects, d->n_rects);
        s->boxes = copy_in(d->boxes, d->n_boxes);
        s->gspheres = copy_in(d->gspheres, d->n_gspheres);
        s->xforms = copy_in(d->xforms, d->n_xforms);
        s->media = copy_in(d->media, d->n_media);
        s->materials = copy_in(d->materials, d->n_materials);
        s->textures = copy_in(d->textures, d->n_textures);
        s->perlin_vec = copy_in(d->perlin_vec, 768);
        s->perlin_perm = copy_in(d->perlin_perm, 768);
        for (uint32_t i = 0; i < d->n_images; i++) {
            s->images.push_back(d->images[i]);
            s->image_data.push_back(copy_in(d->images[i].rgb8, (size_t)d->images[i].w * d->images[i].h * 3));
        }
        memcpy(s->background, d->background, sizeof s->background);
        s->distant = copy_in(d->distant_lights, d->n_distant_lights);
        for (const rt_gsphere &g : s->gspheres) s->gsph_derived.push_back(derive_gsphere(g));
        auto t0 = std::chrono::steady_clock::now();
        std::vector<Aabb> bounds(s->prims.size());
        {
            // every primitive's box; large scenes share the loop between the builders' threads
            std::atomic<size_t> bad{SIZE_MAX}; // lowest primitive without a finite box
            auto some = [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; i++) {
                    bounds[i] = prim_bounds(*s, s->prims[i]);
                    // NaN or infinite geometry would reach the builders' bucket and Morton casts
                    bool ok = true;
                    for (int k = 0; k < 3; k++)
                        ok = ok && std::isfinite(bounds[i].lo[k]) && std::isfinite(bounds[i].hi[k]) &&
                             bounds[i].lo[k] <= bounds[i].hi[k];
                    if (!ok) {
                        size_t seen = bad.load();
                        while (i < seen && !bad.compare_exchange_weak(seen, i)) {}
                    }
                }
            };
            const size_t n = bounds.size();
            const unsigned threads = n < (1u << 16) ? 1u : rt::build_threads();
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < threads; t++) pool.emplace_back(some, n * t / threads, n * (t + 1) / threads);
            some(0, n / threads);
            for (auto &t : pool) t.join();
            if (bad.load() != SIZE_MAX)
                return fail(RT_ERR_INVALID, "primitive %zu has no finite bounding box", bad.load());
        }
        s->enclosing = find_enclosing(*s, bounds);
        s->balls = find_balls(*s, bounds);
        std::vector<uint32_t> in_tree; // primitives the tree is built over, when some are left out
        if (!s->enclosing.empty()) {
            size_t e = 0;
            for (uint32_t i = 0; i < (uint32_t)bounds.size(); i++) {
                if (e < s->enclosing.size() && s->enclosing[e] == i) {
                    e++;
                    continue;
                }
                bounds[in_tree.size()] = bounds[i];
                in_tree.push_back(i);
            }
            bounds.resize(in_tree.size());
        }
        std::string err;
        int rc;
        if (opts && (opts->flags & RT_BUILD_ON_DEVICE)) {
            if (opts->split_method != RT_SPLIT_HLBVH)
                return fail(RT_ERR_INVALID, "only the HLBVH split method builds on the device");
            rc = rt::build_tree_hlbvh_device(bounds, opts->max_prims_in_node, (int)opts->device, s->tree, err);
        } else {
            rc = rt::build_tree(bounds, opts ? opts->max_prims_in_node : 4,
                                opts ? opts->split_method : (uint32_t)RT_SPLIT_SAH, s->tree, err);
        }
        if (rc != RT_OK) return fail(rc, "%s", err.c_str());
        if (!in_tree.empty())
            for (uint32_t &id : s->tree.order) id = in_tree[id];
        s->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *out = s.release();
        return RT_OK;
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_NOMEM, "out of host memory building the scene");
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID, "scene construction failed: %s", e.what());
    }
}

void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    for (auto &kv : s->devices) {
        (void)hipSetDevice(kv.first);
        release(*kv.second);
    }
    delete s;
}

int rt_scene_world_bound(const rt_scene *s, float out[6]) { // bvh.rs:728-734
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    Aabb b = rt::empty_box();
    if (!s->tree.nodes.empty()) {
        memcpy(b.lo, s->tree.nodes[0].lo, 12);
        memcpy(b.hi, s->tree.nodes[0].hi, 12);
    }
    for (uint32_t id : s->enclosing) b = join(b, prim_bounds(*s, s->prims[id])); // they belong to the scene
    memcpy(out, b.lo, 12);
    memcpy(out + 3, b.hi, 12);
    return RT_OK;
}

int rt_scene_get_info(const rt_scene *s, rt_scene_info *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    out->n_prims = (uint32_t)s->prims.size();
    out->n_nodes = (uint32_t)s->tree.nodes.size();
    out->max_depth = s->tree.max_depth;
    out->stack_slots = stack_slots(s);
    {
        std::lock_guard<std::mutex> g(s->lock); // another thread may be adding a device
        for (auto &kv : s->devices) out->device_bytes = kv.second->bytes;
    }
    out->build_ms = s->build_ms;
    out->n_enclosing = (uint32_t)s->enclosing.size();
    out->n_balls = (uint32_t)s->balls.size();
    for (size_t k = 0; k < s->enclosing.size() && k < 4; k++) out->enclosing[k] = s->enclosing[k];
    return RT_OK;
}

int64_t rt_scene_get_nodes(const rt_scene *s, rt_linear_node *out, size_t cap) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    size_t n = std::min(cap, s->tree.nodes.size());
    if (out && n) memcpy(out, s->tree.nodes.data(), n * sizeof *out);
    return (int64_t)s->tree.nodes.size();
}

int64_t rt_scene_get_prim_order(const rt_scene *s, uint32_t *out, size_t cap) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    size_t n = std::min(cap, s->tree.order.size());
    if (out && n) memcpy(out, s->tree.order.data(), n * sizeof *out);
    return (int64_t)s->tree.order.size();
}

int rt_scene_upload(rt_scene *s, int device) {
    DeviceBuffers *db;
    return get_device(s, device, &db);
}

int rt_intersect_dev(rt_scene *s, const rt_ray *d_rays, size_t n, rt_hit *d_hits, int device, void *stream) {
    DeviceBuffers *db;
    int rc = get_device(s, device, &db);
    if (rc != RT_OK) return rc;
    if (n && (!d_rays || !d_hits)) return fail(RT_ERR_INVALID, "null ray or hit buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (size_t at = 0; at < n; at += kChunk) {
        uint32_t m = (uint32_t)std::min(kChunk, n - at);
        if ((rc = with_cursor(*db, st, [&](uint32_t *cursor) {
                 return rt::launch_intersect(db->dev, db->shape, d_rays + at, m, d_hits + at, nullptr, cursor, st);
             })) != RT_OK)
            return rc;
    }
    return RT_OK;
}

// rt_intersect_dev for callers that must not allocate: the scene is already resident, the
// block cursors live in a file-scope device array, one launch.
int rt_intersect_dev_noalloc(rt_scene *s, const rt_ray *d_rays, size_t n, rt_hit *d_hits, int device, void *stream) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    if (n > RT_NOALLOC_MAX_RAYS)
        return fail(RT_ERR_LIMIT, "rt_intersect_dev_noalloc takes at most %u rays per call (got %zu)",
                    (unsigned)RT_NOALLOC_MAX_RAYS, n);
    if (n && (!d_rays || !d_hits)) return fail(RT_ERR_INVALID, "null ray or hit buffer");
    DeviceBuffers *db = nullptr;
    {
        std::lock_guard<std::mutex> g(s->lock);
        auto it = s->devices.find(device);
        if (it != s->devices.end()) db = it->second.get();
    }
    uint32_t *cursor = rt::static_cursor(device);
    if (!db || !cursor)
        return fail(RT_ERR_INVALID, "scene is not resident on device %d: call rt_scene_upload first "
                                    "(rt_intersect_dev_noalloc allocates nothing)", device);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(cursor, 0, rt::kCursorBytes, st));
    HIP_TRY(rt::launch_intersect(db->dev, db->shape, d_rays, (uint32_t)n, d_hits, nullptr, cursor, st));
    return RT_OK;
}

int rt_intersect_count(rt_scene *s, const rt_ray *d_rays, size_t n, rt_hit *d_hits, int device,
                       uint64_t out[9]) {
    DeviceBuffers *db;
    int rc = get_device(s, device, &db);
    if (rc != RT_OK) return rc;
    if (!out || (n && (!d_rays || !d_hits))) return fail(RT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(db->render_lock);
    HIP_TRY(hipMemset(db->counters, 0, rt::kCounterWords * sizeof(unsigned long long)));
    for (size_t at = 0; at < n; at += kChunk) {
        uint32_t m = (uint32_t)std::min(kChunk, n - at);
        if ((rc = with_cursor(*db, nullptr, [&](uint32_t *cursor) {
                 return rt::launch_intersect(db->dev, db->shape, d_rays + at, m, d_hits + at, db->counters, cursor,
                                             nullptr);
             })) != RT_OK)
            return rc;
    }
    HIP_TRY(hipMemcpy(out, db->counters, 9 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_occluded_dev(rt_scene *s, const rt_ray *d_rays, size_t n, uint8_t *d_out, int device, void *stream) {
    DeviceBuffers *db;
    int rc = get_device(s, device, &db);
    if (rc != RT_OK) return rc;
    if (n && (!d_rays || !d_out)) return fail(RT_ERR_INVALID, "null ray or output buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (size_t at = 0; at < n; at += kChunk) {
        uint32_t m = (uint32_t)std::min(kChunk, n - at);
        if ((rc = with_cursor(*db, st, [&](uint32_t *cursor) {
                 return rt::launch_occluded(db->dev, db->shape, d_rays + at, m, d_out + at, cursor, st);
             })) != RT_OK)
            return rc;
    }
    return RT_OK;
}

// host-pointer variants: stage through device memory in pieces
static int staged(rt_scene *s, const rt_ray *rays, size_t n, rt_hit *hits, uint8_t *occ, int device) {
    DeviceBuffers *db;
    int rc = get_device(s, device, &db);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (!rays || (!hits && !occ)) return fail(RT_ERR_INVALID, "null ray or output buffer");
    const size_t piece = std::min(n, (size_t)1 << 24);
    rt_ray *d_rays = nullptr;
    void *d_out = nullptr;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreate(&st));
    hipError_t e1 = hipMalloc((void **)&d_rays, piece * sizeof(rt_ray));
    hipError_t e2 = hipMalloc(&d_out, piece * (hits ? sizeof(rt_hit) : 1));
    rc = RT_OK;
    if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(RT_ERR_HIP, "hipMalloc of staging buffers failed");
    for (size_t at = 0; at < n && rc == RT_OK; at += piece) {
        size_t m = std::min(piece, n - at);
        if (hipMemcpyAsync(d_rays, rays + at, m * sizeof(rt_ray), hipMemcpyHostToDevice, st) != hipSuccess) {
            rc = fail(RT_ERR_HIP, "ray upload failed");
            break;
        }
        rc = hits ? rt_intersect_dev(s, d_rays, m, static_cast<rt_hit *>(d_out), device, st)
                  : rt_occluded_dev(s, d_rays, m, static_cast<uint8_t *>(d_out), device, st);
        if (rc != RT_OK) break;
        hipError_t e = hits ? hipMemcpyAsync(hits + at, d_out, m * sizeof(rt_hit), hipMemcpyDeviceToHost, st)
                            : hipMemcpyAsync(occ + at, d_out, m, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, "kernel or download failed: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(st);
    drop_cursor(*db, st);
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    (void)hipStreamDestroy(st);
    return rc;
}

int rt_intersect(rt_scene *s, const rt_ray *rays, size_t n, rt_hit *hits, int device) {
    if (n && !hits) return fail(RT_ERR_INVALID, "null hit buffer");
    return staged(s, rays, n, hits, nullptr, device);
}

int rt_occluded(rt_scene *s, const rt_ray *rays, size_t n, uint8_t *out, int device) {
    if (n && !out) return fail(RT_ERR_INVALID, "null output buffer");
    return staged(s, rays, n, nullptr, out, device);
}

int rt_generate_rays(const rt_camera *cam, const rt_render_opts *opts, uint64_t first_path, size_t n,
                     rt_ray *rays, int device) {
    if (!cam || !opts || (n && !rays) || !opts->spp || !opts->width)
        return fail(RT_ERR_INVALID, "bad argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(RT_ERR_NO_DEVICE, "device %d not available; librt355 has no CPU path", device);
    if (n > kChunk) return fail(RT_ERR_LIMIT, "at most %zu rays per call", kChunk);
    if (int bad = sampler_check(*opts)) return bad;
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    rt_ray *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, n * sizeof(rt_ray)));
    rt::FrameDev F{};
    F.width = opts->width;
    F.height = opts->height;
    F.spp = opts->spp;
    F.tiles_x = (opts->width + 31) / 32;
    F.shard_count = 1;
    F.seed = opts->seed;
    F.sampler = opts->sampler;
    F.sampler_arg = opts->sampler_arg;
    if (opts->sampler == RT_SAMPLER_HALTON) rt::halton_setup(opts->width, opts->height, F.halton);
    void *smem = nullptr;
    size_t sbytes = 0;
    const uint64_t first_pixel = first_path / opts->spp, last_pixel = (first_path + n - 1) / opts->spp;
    int rc = sampler_prepare(F, (uint32_t)(last_pixel - first_pixel + 1), false, first_pixel, &smem, &sbytes, nullptr);
    if (rc != RT_OK) {
        (void)hipFree(d);
        if (smem) (void)hipFree(smem);
        return rc;
    }
    hipError_t e = rt::launch_generate_rays(F, *cam, first_path, (uint32_t)n, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(rays, d, n * sizeof(rt_ray), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (smem) (void)hipFree(smem);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "ray generation failed: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C++" {
namespace rt {

// The tables of data/src/sobol.rs for one frame, computed instead of stored.  Generator
// matrices of dimensions 0-4 come from the Joe-Kuo direction numbers the file cites
// (data/src/sobol.rs:57-63): column k of a dimension with primitive polynomial of degree s
// and coefficients a is m_k * 2^(51-k), m_k = m_(k-s) ^ 2^s m_(k-s) ^ sum a_i 2^i m_(k-i);
// MATRICES_32 keeps the top 32 of the 52 bits.  With resolution 2^m, T(index) = (x << m) | y
// where x, y are the top m bits of dimensions 0 and 1; MATRICES_INV[m-1] inverts T on index
// bits 0..2m-1 and MATRICES[m-1][c] = T(2^(2m+c)).
uint32_t sobol_setup(uint32_t width, uint32_t height, SobolDev &t) {
    struct Poly { uint32_t s, a; uint64_t m[3]; };
    static const Poly poly[5] = {{0, 0, {0, 0, 0}}, {1, 0, {1, 0, 0}}, {2, 1, {1, 3, 0}}, {3, 1, {1, 3, 1}}, {3, 2, {1, 1, 1}}};
    memset(&t, 0, sizeof t);
    for (int d = 0; d < 5; d++) {
        std::vector<uint64_t> mk(52, 1); // degree 0: van der Corput, every m_k = 1
        const Poly &p = poly[d];
        for (uint32_t k = 0; k < 52 && p.s; k++) {
            if (k < p.s) {
                mk[k] = p.m[k];
                continue;
            }
            uint64_t v = mk[k - p.s] ^ (mk[k - p.s] << p.s);
            for (uint32_t i = 1; i < p.s; i++)
                if ((p.a >> (p.s - 1 - i)) & 1u) v ^= mk[k - i] << i;
            mk[k] = v;
        }
        for (uint32_t k = 0; k < 52; k++) t.m32[d][k] = (uint32_t)((mk[k] << (51 - k)) >> 20);
    }
    uint32_t m = 0;
    const uint32_t res = std::min(std::max(width, height), kSobolMaxResolution); // callers reject larger frames
    while ((1u << m) < res) m++; // round_up_pow_2 + log2usize, sobol.rs:15-21
    if (m == 0) return 0;
    auto T = [&](uint32_t j) {
        return ((uint64_t)(t.m32[0][j] >> (32 - m)) << m) | (uint64_t)(t.m32[1][j] >> (32 - m));
    };
    // Gauss-Jordan over GF(2) on the pairs (T(2^j), 2^j), j < 2m
    const uint32_t n = 2 * m;
    std::vector<uint64_t> image(n), source(n);
    std::vector<bool> pivot(n, false);
    for (uint32_t j = 0; j < n; j++) {
        image[j] = T(j);
        source[j] = 1ull << j;
    }
    for (uint32_t c = 0; c < n; c++) {
        uint32_t r = 0;
        while (r < n && (pivot[r] || !((image[r] >> c) & 1u))) r++;
        if (r == n) return m; // cannot happen: T is invertible on these bits
        pivot[r] = true;
        for (uint32_t i = 0; i < n; i++)
            if (i != r && ((image[i] >> c) & 1u)) {
                image[i] ^= image[r];
                source[i] ^= source[r];
            }
        t.inv[c] = 0;
    }
    for (uint32_t r = 0; r < n; r++) t.inv[__builtin_ctzll(image[r])] = source[r];
    for (uint32_t c = 0; 2 * m + c < 52; c++) t.vdc[c] = T(2 * m + c);
    return m;
}

// Halton::new, colour/src/samplers/halton.rs:25-52, and the permutation table of
// data/src/lib.rs:16-41 (digits of each prime shuffled by one Rng::default() stream).
void halton_setup(uint32_t width, uint32_t height, HaltonDev &h) {
    const uint32_t res[2] = {width, height};
    for (int i = 0; i < 2; i++) {
        const uint32_t base = (uint32_t)i + 2, limit = std::min(res[i], 128u);
        uint32_t scale = 1, exponent = 0;
        while (scale < limit) {
            scale *= base;
            exponent++;
        }
        h.scale[i] = scale;
        h.exponent[i] = exponent;
    }
    h.stride = h.scale[0] * h.scale[1];
    struct Gcd { // extended_gcd / multiplicative_inverse, halton.rs:113-126
        static void run(int64_t a, int64_t b, int64_t &x, int64_t &y) {
            if (b == 0) { x = 1; y = 0; return; }
            int64_t xp, yp;
            run(b, a % b, xp, yp);
            x = yp;
            y = xp - (a / b) * yp;
        }
        static uint32_t inverse(int64_t a, int64_t n) {
            int64_t x, y;
            run(a, n, x, y);
            const int64_t r = x % n;
            return (uint32_t)(r < 0 ? r + n : r);
        }
    };
    h.mult_inv[0] = Gcd::inverse(h.scale[1], h.scale[0]);
    h.mult_inv[1] = Gcd::inverse(h.scale[0], h.scale[1]);
    uint64_t state = 0x853c49e6748fea9bULL; // Rng::default, geometry/src/rng.rs:15-22
    const uint64_t inc = 0xda3e39cb94b95bdbULL;
    auto next = [&]() { // uniform_u32, rng.rs:41-47
        const uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        const uint32_t shifted = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
        return (shifted >> rot) | (shifted << ((~rot + 1u) & 31u));
    };
    auto below = [&](uint32_t limit) { // uniform_u32_limit, rng.rs:50-58
        const uint32_t threshold = (~limit + 1u) % limit;
        for (;;) {
            const uint32_t r = next();
            if (r >= threshold) return r % limit;
        }
    };
    const uint32_t primes[5] = {2, 3, 5, 7, 11};
    uint8_t *p = h.perm;
    for (uint32_t n : primes) {
        for (uint32_t j = 0; j < n; j++) p[j] = (uint8_t)j;
        for (uint32_t j = 0; j < n; j++) std::swap(p[j], p[j + below(n - j)]); // shuffle_dims(.., 1), rng.rs:77-84
        p += n;
    }
}

} // namespace rt
} // extern "C++"

namespace {

// colour/src/filters/{box,triangle,gaussian,mitchell,lanczos_sinc}.rs, evaluated in fp32 with
// the host's libm.  Mitchell's outer cubic starts (-B - 6C) x^3 as in pbrt (mitchell.rs:33
// multiplies B by 6C).
float filter_value(const rt_filter &f, float x, float y) {
    const float rx = f.radius[0], ry = f.radius[1];
    auto pos = [](float v) { return v > 0.0f ? v : 0.0f; };
    switch (f.kind) {
    case RT_FILTER_BOX: return 1.0f;
    case RT_FILTER_TRIANGLE: return pos(rx - fabsf(x)) * pos(ry - fabsf(y));
    case RT_FILTER_GAUSSIAN: {
        auto g = [&](float d, float r) { return pos(expf(-f.a * d * d) - expf(-f.a * r * r)); };
        return g(x, rx) * g(y, ry);
    }
    case RT_FILTER_MITCHELL: {
        const float B = f.a, Cc = f.b;
        auto m = [&](float v) {
            const float t = fabsf(2.0f * v);
            if (t > 1.0f)
                return ((-B - 6.0f * Cc) * t * t * t + (6.0f * B + 30.0f * Cc) * t * t +
                        (-12.0f * B - 48.0f * Cc) * t + (8.0f * B + 24.0f * Cc)) * (1.0f / 6.0f);
            return ((12.0f - 9.0f * B - 6.0f * Cc) * t * t * t + (-18.0f + 12.0f * B + 6.0f * Cc) * t * t +
                    (6.0f - 2.0f * B)) * (1.0f / 6.0f);
        };
        return m(x * (1.0f / rx)) * m(y * (1.0f / ry));
    }
    case RT_FILTER_LANCZOS: {
        const float pi = 3.14159265358979323846f;
        auto sinc = [&](float v) { return v < 1e-5f ? 1.0f : sinf(pi * v) / (pi * v); };
        auto ws = [&](float v, float r) {
            v = fabsf(v);
            return v > r ? 0.0f : sinc(v) * sinc(v / f.a);
        };
        return ws(x, rx) * ws(y, ry);
    }
    default: return 0.0f;
    }
}

} // namespace

int rt_filter_table(const rt_filter *f, float out[256]) {
    if (!f || !out) return fail(RT_ERR_INVALID, "null argument");
    if (f->kind < RT_FILTER_BOX || f->kind > RT_FILTER_LANCZOS) return fail(RT_ERR_INVALID, "unknown filter kind");
    if (!(f->radius[0] > 0.0f && f->radius[1] > 0.0f) || f->radius[0] > RT_FILTER_MAX_RADIUS ||
        f->radius[1] > RT_FILTER_MAX_RADIUS)
        return fail(RT_ERR_INVALID, "filter radius must be in (0, %g]", (double)RT_FILTER_MAX_RADIUS);
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) // film.rs:99-113
            out[y * 16 + x] = filter_value(*f, ((float)x + 0.5f) * f->radius[0] / 16.0f,
                                           ((float)y + 0.5f) * f->radius[1] / 16.0f);
    return RT_OK;
}

int rt_sobol_tables(uint32_t width, uint32_t height, uint32_t m32[260], uint64_t vdc[52], uint64_t inv[52]) {
    if (!m32 || !vdc || !inv || !width || !height) return fail(RT_ERR_INVALID, "bad argument");
    if (std::max(width, height) > rt::kSobolMaxResolution)
        return fail(RT_ERR_LIMIT, "the Sobol' tables end at a resolution of 2^26 (data/src/sobol.rs)");
    rt::SobolDev t;
    (void)rt::sobol_setup(width, height, t);
    memcpy(m32, t.m32, sizeof t.m32);
    memcpy(vdc, t.vdc, sizeof t.vdc);
    memcpy(inv, t.inv, sizeof t.inv);
    return RT_OK;
}

int rt_film_resolve_dev(float *d_rgba, uint32_t width, uint32_t height, int device, void *stream) {
    if (!d_rgba) return fail(RT_ERR_INVALID, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(RT_ERR_NO_DEVICE, "device %d not available; librt355 has no CPU path", device);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(rt::launch_film_resolve(reinterpret_cast<float4 *>(d_rgba), (size_t)width * height,
                                    static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int rt_render_dev(rt_scene *s, const rt_camera *cam, const rt_render_opts *o, float *d_rgba, rt_stats *stats,
                  int device, void *stream) {
    DeviceBuffers *db;
    int rc = get_device(s, device, &db);
    if (rc != RT_OK) return rc;
    if (!cam || !o || !d_rgba) return fail(RT_ERR_INVALID, "null argument");
    if (!o->width || !o->height || !o->spp || !o->max_depth) return fail(RT_ERR_INVALID, "empty frame");
    if ((uint64_t)o->width * o->height * o->spp >= (1ull << 32))
        return fail(RT_ERR_LIMIT, "width * height * spp must stay below 2^32");
    if (s->materials.empty() && !s->prims.empty()) return fail(RT_ERR_INVALID, "scene has no materials");
    // the path kernel carries a segment's depth in 16 bits
    if (o->max_depth >= (1u << 16)) return fail(RT_ERR_LIMIT, "max_depth must stay below 65536");
    const uint32_t shard_count = o->shard_count ? o->shard_count : 1;
    if (o->shard_index >= shard_count) return fail(RT_ERR_INVALID, "shard index out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    rt::FilterDev film{};
    const bool filtered = o->filter.kind != RT_FILTER_NONE;
    if (filtered) {
        if ((rc = rt_filter_table(&o->filter, film.w)) != RT_OK) return rc;
        film.rx = o->filter.radius[0];
        film.ry = o->filter.radius[1];
    }
    std::lock_guard<std::mutex> g(db->render_lock);

    if (int bad = sampler_check(*o)) return bad;
    rt::FrameDev F{};
    F.width = o->width;
    F.height = o->height;
    F.spp = o->spp;
    F.tiles_x = (o->width + 31) / 32;
    F.shard_index = o->shard_index;
    F.shard_count = shard_count;
    F.seed = o->seed;
    F.sampler = o->sampler;
    F.sampler_arg = o->sampler_arg;
    if (o->sampler == RT_SAMPLER_HALTON) rt::halton_setup(o->width, o->height, F.halton);
    const uint32_t tiles = F.tiles_x * ((o->height + 31) / 32);
    const uint32_t my_tiles = tiles > o->shard_index ? (tiles - o->shard_index + shard_count - 1) / shard_count : 0;
    const uint32_t n_ranks = my_tiles * 1024;
    uint32_t owned_pixels = 0; // pixels of this shard that lie inside the image
    for (uint32_t t = o->shard_index; t < tiles; t += shard_count) {
        uint32_t tx = t % F.tiles_x, ty = t / F.tiles_x;
        owned_pixels += std::min(32u, o->width - tx * 32) * std::min(32u, o->height - ty * 32);
    }
    // Paths in flight per batch.  The library's own choice is 2^27 (two queues of 64-72 bytes per
    // path and 16 bytes of radiance: 20 GB, nothing on a 288 GB device), less where the device has
    // less to give: at most half of what is free now plus what the workspace already holds.
    uint32_t max_paths = o->max_paths;
    if (!max_paths) {
        max_paths = 1u << 27;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_path = 2 * (3 * sizeof(float4) + sizeof(uint2) + (db->dev.n_enclosing || db->dev.n_balls ? sizeof(uint2) : 0)) +
                                    sizeof(float4) + ((o->flags & RT_RENDER_DIRECT_LIGHT) ? 3 * sizeof(float4) : 0) +
                                    (db->split_bounces ? sizeof(float4) + sizeof(float) : 0);
            const size_t held = (size_t)db->ws_paths * per_path;
            const size_t fit = (free_b + held) / 2 / per_path;
            if (fit < max_paths) max_paths = (uint32_t)std::max<size_t>(fit, 1u << 16);
        }
    }
    if (max_paths < 1024) max_paths = 1024;
    // samples per batch: as many as fit, all pixels of the shard in every batch
    uint32_t s_batch = n_ranks ? max_paths / n_ranks : 1;
    if (s_batch < 1) s_batch = 1;
    if (s_batch > o->spp) s_batch = o->spp;
    const uint64_t batch_paths64 = (uint64_t)n_ranks * s_batch;
    if (batch_paths64 >= (1ull << 31)) return fail(RT_ERR_LIMIT, "shard too large for one sample per batch");
    const uint32_t batch_paths = (uint32_t)batch_paths64;
    if ((rc = grow_workspace(*db, batch_paths ? batch_paths : 1, o->max_depth)) != RT_OK) return rc;
    // direct lighting: one launch per bounce, each followed by the any-hit launch of its shadow rays
    const bool direct = (o->flags & RT_RENDER_DIRECT_LIGHT) && db->dev.n_lights != 0;
    if (direct && (rc = grow_shadow_workspace(*db, batch_paths ? batch_paths : 1, o->max_depth)) != RT_OK) return rc;

    // Once few rays are left one launch follows them to the end of their paths
    // (launch_bounce_loop): a launch per bounce is the better deal for full queues (less state
    // per ray, rays of one depth shade together) and a floor of ~0.1 ms per bounce under the
    // frame time for nearly empty ones.  The first batch of a frame shape looks at the queue size
    // before bounces 2, 4, 8, 16, ... (one small synchronising copy each) and switches below
    // `loop_below` rays; later batches and frames of the same shape switch at the bounce found
    // then, with no round trip to the host in the middle of a frame.
    // RT355_LOOP_FROM=<bounce> fixes the bounce instead, RT355_LOOP_BELOW=<rays> the threshold
    // (0: never), for tests and experiments.
    // The hooks that fix the bounce or the threshold instead were read when the scene was placed
    // on the device (DeviceBuffers::Hooks).
    const DeviceBuffers::Hooks &hooks = db->hooks;
    const uint32_t loop_from = hooks.loop_from;
    const uint64_t loop_below = hooks.loop_below_set ? hooks.loop_below : (db->shape.lds_nodes ? (12u << 20) : (2u << 20));
    const bool loop_steal = hooks.loop_steal;
    const uint32_t turn_limit = hooks.turn_limit;
    const bool want_counters = (o->flags & RT_RENDER_COUNTERS) != 0;
    const bool want_timing = (o->flags & RT_RENDER_TIMING) != 0;
    // A frame the caller does not want statistics of is enqueued and not waited for (below).  The
    // frame before it on this device may have been one of those: wait for it now, since both use
    // the same workspace, and report the error its one-launch kernel may have flagged.
    const bool no_wait = stats == nullptr;
    if (db->pending) {
        HIP_TRY(hipEventSynchronize(db->pending_done));
        db->pending = false;
        if (*db->pending_error) {
            *db->pending_error = 0;
            return fail(RT_ERR_LIMIT, "a wave of the path kernel exceeded its feed turns in the frame before this one "
                                      "(enqueued without statistics): that frame is incomplete");
        }
    }
    if (no_wait && !db->pending_done) {
        HIP_TRY(hipEventCreateWithFlags(&db->pending_done, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void **)&db->pending_error, sizeof(uint32_t), hipHostMallocDefault));
        *db->pending_error = 0;
    }
    EventList events; // destroyed on every way out
    std::vector<hipEvent_t> &ev = events.timed;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    HIP_TRY(events.make(&ev_begin));
    HIP_TRY(events.make(&ev_end));
    HIP_TRY(hipMemsetAsync(d_rgba, 0, (size_t)o->width * o->height * 4 * sizeof(float), st));
    if (want_counters) HIP_TRY(hipMemsetAsync(db->counters, 0, rt::kCounterWords * sizeof(unsigned long long), st));
    HIP_TRY(hipEventRecord(ev_begin, st));
    // the sampler's tables or per-pixel samples belong to the frame's time
    if ((rc = sampler_prepare(F, n_ranks, true, 0, &db->sampler_mem, &db->sampler_bytes, st)) != RT_OK) return rc;
    uint64_t segments = 0, paths = 0, launches = 0, occlusion = 0;
    uint32_t batches = 0, max_bounce = 0;
    std::vector<uint32_t> counts(count_words(o->max_depth)), scounts(direct ? count_words(o->max_depth) : 0);
    std::vector<uint64_t> per_bounce(o->max_depth, 0);
    for (uint32_t s0 = 0; s0 < o->spp && n_ranks; s0 += s_batch, batches++) {
        const uint32_t sc = std::min(s_batch, o->spp - s0);
        const uint32_t n_local = n_ranks * sc;
        DeviceBuffers::LoopHint &hint = db->loop_hint;
        const bool hinted = hint.valid && hint.width == o->width && hint.height == o->height && hint.paths == n_local &&
                            hint.depth == o->max_depth && hint.shard_index == o->shard_index &&
                            hint.shard_count == shard_count && hint.below == loop_below;
        uint32_t switched_at = 0xffffffffu;
        HIP_TRY(hipMemsetAsync(db->counts, 0, count_words(o->max_depth) * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(db->radiance, 0, (size_t)n_local * sizeof(float4), st));
        if (direct) HIP_TRY(hipMemsetAsync(db->shadow_counts, 0, count_words(o->max_depth) * sizeof(uint32_t), st));
        HIP_TRY(rt::launch_generate(db->dev, F, *cam, db->q[0], db->counts, n_local, s0, sc, st));
        for (uint32_t b = 0; b < o->max_depth; b++) {
            if (want_timing) {
                hipEvent_t e0, e1;
                HIP_TRY(events.make(&e0));
                HIP_TRY(events.make(&e1));
                HIP_TRY(hipEventRecord(e0, st));
                ev.push_back(e0);
                ev.push_back(e1);
            }
            bool rest = !direct && b >= loop_from && b + 1 < o->max_depth;
            if (!direct && !rest && loop_below && hinted) {
                rest = b >= hint.bounce && b + 1 < o->max_depth;
            } else if (!direct && !rest && loop_below && b >= 2 && (b & (b - 1)) == 0 && b + 1 < o->max_depth) {
                uint32_t sizes[rt::kShards * rt::kCountStride];
                HIP_TRY(hipMemcpyAsync(sizes, db->counts + (size_t)b * rt::kShards * rt::kCountStride, sizeof sizes,
                                       hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                uint64_t queued = 0;
                for (uint32_t k = 0; k < rt::kShards; k++) queued += sizes[k * rt::kCountStride];
                rest = queued < loop_below;
                if (rest) switched_at = b;
            }
            if (rest)
                HIP_TRY(rt::launch_bounce_loop(db->dev, db->shape, db->q[b & 1], db->counts, b, o->max_depth - 1,
                                               o->seed, db->radiance, db->bounce_scratch,
                                               want_counters ? db->counters : nullptr, st, loop_steal, turn_limit));
            else if (db->split_bounces) {
                HIP_TRY(rt::launch_trace(db->dev, db->shape, db->q[b & 1], db->hits, db->counts, b,
                                         want_counters ? db->counters : nullptr, st));
                HIP_TRY(rt::launch_shade(db->dev, db->shape, db->q[b & 1], db->q[(b + 1) & 1], db->hits, db->counts, b,
                                         o->max_depth - 1, o->seed, db->radiance, st,
                                         direct ? &db->shadow : nullptr, db->shadow_counts));
            } else
                HIP_TRY(rt::launch_bounce(db->dev, db->shape, db->q[b & 1], db->q[(b + 1) & 1],
                                          db->counts, b, o->max_depth - 1, o->seed, db->radiance,
                                          db->bounce_scratch, want_counters ? db->counters : nullptr, st,
                                          direct ? &db->shadow : nullptr, db->shadow_counts));
            if (direct)
                HIP_TRY(rt::launch_shadow(db->dev, db->shape, db->shadow, db->shadow_counts, b, db->radiance, st));
            if (want_timing) HIP_TRY(hipEventRecord(ev.back(), st));
            launches++;
            if (rest) break;
        }
        if (switched_at != 0xffffffffu) {
            hint.width = o->width; hint.height = o->height; hint.paths = n_local; hint.depth = o->max_depth;
            hint.shard_index = o->shard_index; hint.shard_count = shard_count; hint.bounce = switched_at;
            hint.below = loop_below; hint.valid = true;
        }
        if (filtered) {
            HIP_TRY(rt::launch_film(F, film, db->radiance, s0, sc, reinterpret_cast<float4 *>(d_rgba), st));
            if (s0 + sc >= o->spp && !(o->flags & RT_RENDER_RAW_FILM))
                HIP_TRY(rt::launch_film_resolve(reinterpret_cast<float4 *>(d_rgba), (size_t)o->width * o->height, st));
        } else {
            HIP_TRY(rt::launch_resolve(F, db->radiance, n_ranks, sc, s0 + sc >= o->spp ? 1 : 0,
                                       reinterpret_cast<float4 *>(d_rgba), st));
        }
        if (no_wait && switched_at == 0xffffffffu) {
            // nobody asked for this batch's counts: only the error flag travels, behind everything
            // else, and is looked at by the next call on this device
            HIP_TRY(hipMemcpyAsync(db->pending_error, db->counts + rt::kLoopErrorWord, sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, st));
            continue;
        }
        // queue sizes of this batch = ray segments traced per bounce
        HIP_TRY(hipMemcpyAsync(counts.data(), db->counts, counts.size() * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        if (direct)
            HIP_TRY(hipMemcpyAsync(scounts.data(), db->shadow_counts, scounts.size() * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (counts[rt::kLoopErrorWord])
            return fail(RT_ERR_LIMIT, "a wave of the path kernel exceeded %u feed turns and dropped its segments: "
                                      "the frame is incomplete", turn_limit);
        if (hinted && loop_below) {
            // Does the remembered bounce still fit?  Another camera or seed may have changed how fast the
            // queue shrinks: the kernel that follows paths to their end should take over at the first of
            // the bounces 2, 4, 8, ... whose queue is below the threshold.  If it did not, the next frame
            // of this shape looks again.
            auto queued_at = [&](uint32_t b) {
                uint64_t q = 0;
                for (uint32_t k = 0; k < rt::kShards; k++) q += counts[((size_t)b * rt::kShards + k) * rt::kCountStride];
                return q;
            };
            const uint32_t at = hint.bounce;
            if (queued_at(at) >= loop_below || (at >= 4 && queued_at(at / 2) < loop_below)) hint.valid = false;
        }
        const uint64_t live = (uint64_t)owned_pixels * sc; // entries of bounce 0 that are not padding
        paths += live;
        for (uint32_t b = 0; b < o->max_depth; b++) {
            uint64_t q = 0;
            for (uint32_t k = 0; k < rt::kShards; k++) q += counts[((size_t)b * rt::kShards + k) * rt::kCountStride];
            if (b == 0) q = live;
            per_bounce[b] += q;
            segments += q;
            if (q && b > max_bounce) max_bounce = b;
            if (direct)
                for (uint32_t k = 0; k < rt::kShards; k++)
                    occlusion += scounts[((size_t)b * rt::kShards + k) * rt::kCountStride];
        }
    }
    if (no_wait) {
        HIP_TRY(hipEventRecord(db->pending_done, st));
        db->pending = true;
        return RT_OK;
    }
    HIP_TRY(hipEventRecord(ev_end, st));
    HIP_TRY(hipEventSynchronize(ev_end));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->paths = paths;
        stats->segments = segments + occlusion;
        stats->occlusion_rays = occlusion;
        stats->trace_launches = launches;
        stats->batches = batches;
        stats->max_bounce = max_bounce;
        for (uint32_t b = 0; b < o->max_depth && b < 64; b++) stats->bounce_rays[b] = per_bounce[b];
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->total_ms = ms;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            stats->trace_ms += ms;
        }
        if (want_counters) {
            unsigned long long c[9];
            HIP_TRY(hipMemcpy(c, db->counters, sizeof c, hipMemcpyDeviceToHost));
            stats->nodes_visited = c[0];
            for (int k = 0; k < 8; k++) stats->prims_tested[k] = c[1 + k];
#ifdef RT_TAIL_PROBE
            {
                unsigned long long tp[4];
                HIP_TRY(hipMemcpy(tp, db->counters + 28, sizeof tp, hipMemcpyDeviceToHost));
                if (tp[3]) {
                    fprintf(stderr, "rt355 tail probe: loop launch %.3f ms, mean wave %.3f ms, %llu waves\n",
                            (double)(tp[1] - ~tp[0]) / 1e5, (double)tp[2] / (double)tp[3] / 1e5, tp[3]);
                    // per-wave records: start, queue read out, end (100 MHz), segments held then | turns after
                    const uint32_t waves = rt::bounce_waves(db->shape);
                    std::vector<unsigned long long> rec((size_t)waves * 5);
                    HIP_TRY(hipMemcpy2D(rec.data(), 40, db->bounce_scratch, rt::kWaveLoopBytes, 40, waves,
                                        hipMemcpyDeviceToHost));
                    if (!hooks.tail_file.empty()) {
                        if (FILE *out = fopen(hooks.tail_file.c_str(), "w")) {
                            const unsigned long long t0 = ~tp[0];
                            for (uint32_t w = 0; w < waves; w++) {
                                const unsigned long long *r = &rec[(size_t)w * 5];
                                if (r[2] < t0 || r[2] > tp[1]) continue; // wave did not run in this launch
                                fprintf(out, "%u %.4f %.4f %.4f %u %u %u %u\n", w, (double)(r[0] - t0) / 1e5,
                                        r[1] ? (double)(r[1] - t0) / 1e5 : -1.0, (double)(r[2] - t0) / 1e5,
                                        (unsigned)(r[3] >> 32), (unsigned)r[3], (unsigned)(r[4] >> 32), (unsigned)r[4]);
                            }
                            fclose(out);
                        }
                    }
                }
            }
#endif
#ifdef RT_PROFILE
            unsigned long long pr[32];
            HIP_TRY(hipMemcpy(pr, db->counters + 32, sizeof pr, hipMemcpyDeviceToHost));
            fprintf(stderr, "rt355 profile:");
            for (int k = 0; k < 32; k++) fprintf(stderr, " %llu", pr[k]);
            fprintf(stderr, "\n");
#endif
        }
    }
    return RT_OK;
}

int rt_render(rt_scene *s, const rt_camera *cam, const rt_render_opts *o, float *rgba, rt_stats *stats,
              int device) {
    DeviceBuffers *db;
    int rc = get_device(s, device, &db);
    if (rc != RT_OK) return rc;
    if (!o || !rgba) return fail(RT_ERR_INVALID, "null argument");
    const size_t bytes = (size_t)o->width * o->height * 4 * sizeof(float);
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, bytes ? bytes : 16));
    rt_stats waited; // statistics asked for = the call waits for the frame and reports its errors
    rc = rt_render_dev(s, cam, o, d, stats ? stats : &waited, device, nullptr);
    if (rc == RT_OK && hipMemcpy(rgba, d, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(RT_ERR_HIP, "framebuffer download failed");
    (void)hipFree(d);
    return rc;
}

} // extern "C"
