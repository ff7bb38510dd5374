/*
 * orc_render.c -- camera rays, textures, materials and the path loop of the
 * CPU oracle (test infrastructure only; see orc_math.h).
 *
 * Camera: colour/src/camera/perspective.rs:54-78, geometry/src/sampling.rs:4-22
 * (second branch uses x / y), geometry/src/transform.rs:530-552.
 * Next-ray origin: shapes/src/interaction.rs:86-99, geometry/src/lib.rs:129-151; shadow rays:
 * spawn_ray_to / spawn_ray_intersect, interaction.rs:102-133.
 * RNG: geometry/src/rng.rs.  Materials, textures, Perlin noise and constant
 * media have no code in the reference; they follow the schema of the scene
 * files under test/ (Lambertian / Metallic / Dielectric / DiffuseLight,
 * Solid / Checker / Image / Noise, Volume) and are specified here.
 */
#include <stdlib.h>
#include "orc_internal.h"

#define G3 orc_gamma(3)

v3 orc_offset_origin(v3 p, v3 err, v3 n, v3 w) { /* lib.rs:129-151 */
    float d = v3_dot(v3_abs(n), err);
    v3 off = v3_scale(n, d);
    if (v3_dot(n, w) < 0.0f) off = v3_neg(off);
    v3 po = v3_add(p, off);
    if (off.x > 0.0f) po.x = next_float_up(po.x); else if (off.x < 0.0f) po.x = next_float_down(po.x);
    if (off.y > 0.0f) po.y = next_float_up(po.y); else if (off.y < 0.0f) po.y = next_float_down(po.y);
    if (off.z > 0.0f) po.z = next_float_up(po.z); else if (off.z < 0.0f) po.z = next_float_down(po.z);
    return po;
}

/* Interaction::spawn_ray_to, interaction.rs:102-116 (spawn_ray_intersect, :119-133, is the same
 * with the other interaction's point as the target): the origin is offset towards the target,
 * the direction is target - origin, not normalised, and the ray ends just short of the target. */
void orc_spawn_ray_to(v3 p, v3 err, v3 n, v3 target, float time, rt_ray *out) {
    v3 o = orc_offset_origin(p, err, n, v3_sub(target, p));
    v3 d = v3_sub(target, o);
    out->o[0] = o.x; out->o[1] = o.y; out->o[2] = o.z;
    out->d[0] = d.x; out->d[1] = d.y; out->d[2] = d.z;
    out->t_max = 1.0f - 0.0001f;
    out->time = time;
}

/* concentric_sample_disk, sampling.rs:4-22 */
static void concentric_disk(float u, float v, float *dx, float *dy) {
    float ox = u * 2.0f - 1.0f, oy = v * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) { *dx = 0.0f; *dy = 0.0f; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        r = ox;
        theta = 0.785398163f * (oy / ox);
    } else {
        r = oy;
        theta = 1.57079637f - 0.785398163f * (ox / oy);
    }
    *dx = orc_cos(theta) * r;
    *dy = orc_sin(theta) * r;
}

void orc_camera_ray(const rt_camera *cam, float fx, float fy, float lu, float lv, float tu,
                    rt_ray *out) {
    const float *m = cam->raster_to_camera; /* Point3 * Transform, transform.rs:332-349 */
    float xp = m[0] * fx + m[1] * fy + m[2] * 0.0f + m[3];
    float yp = m[4] * fx + m[5] * fy + m[6] * 0.0f + m[7];
    float zp = m[8] * fx + m[9] * fy + m[10] * 0.0f + m[11];
    float wp = m[12] * fx + m[13] * fy + m[14] * 0.0f + m[15];
    v3 pc = v3_make(xp, yp, zp);
    if (wp != 1.0f) pc = v3_div(pc, wp);
    v3 o = v3_make(0.0f, 0.0f, 0.0f);
    v3 d = v3_norm(pc);
    if (cam->lens_radius > 0.0f) { /* perspective.rs:60-69 */
        float lx, ly;
        concentric_disk(lu, lv, &lx, &ly);
        lx *= cam->lens_radius;
        ly *= cam->lens_radius;
        float ft = cam->focal_distance / d.z;
        v3 focus = v3_scale(d, ft);
        o = v3_make(lx, ly, 0.0f);
        d = v3_norm(v3_sub(focus, o));
    }
    /* lerp(t, open, close), lib.rs:85-87 */
    float time = (1.0f - tu) * cam->shutter_open + tu * cam->shutter_close;
    /* ray * camera_to_world, transform.rs:530-552 */
    const float *c = cam->camera_to_world;
    v3 wo = v3_make(c[0] * o.x + c[1] * o.y + c[2] * o.z + c[3],
                    c[4] * o.x + c[5] * o.y + c[6] * o.z + c[7],
                    c[8] * o.x + c[9] * o.y + c[10] * o.z + c[11]);
    v3 oerr = v3_scale(
        v3_make(fabsf(c[0] * o.x) + fabsf(c[1] * o.y) + fabsf(c[2] * o.z) + fabsf(c[3]),
                fabsf(c[4] * o.x) + fabsf(c[5] * o.y) + fabsf(c[6] * o.z) + fabsf(c[7]),
                fabsf(c[8] * o.x) + fabsf(c[9] * o.y) + fabsf(c[10] * o.z) + fabsf(c[11])),
        G3);
    v3 wd = v3_make(c[0] * d.x + c[1] * d.y + c[2] * d.z, c[4] * d.x + c[5] * d.y + c[6] * d.z,
                    c[8] * d.x + c[9] * d.y + c[10] * d.z);
    float len_sq = v3_dot(wd, wd);
    float t_max = INFINITY;
    if (len_sq > 0.0f) {
        float dt = v3_dot(v3_abs(wd), oerr) / len_sq;
        wo = v3_add(wo, v3_scale(wd, dt));
        t_max -= dt;
    }
    out->o[0] = wo.x; out->o[1] = wo.y; out->o[2] = wo.z;
    out->d[0] = wd.x; out->d[1] = wd.y; out->d[2] = wd.z;
    out->t_max = t_max;
    out->time = time;
}

/* ---- Halton camera samples: colour/src/samplers/halton.rs + low_discrepancy.rs --------
 * Deviations from the literal text, following pbrt-v3 which it transcribes: dimension 1
 * divides the index by base_scales[1] (halton.rs:98 shifts by it).  Permutations are
 * data/src/lib.rs:16-41: the digits of each prime shuffled by one Rng::default() stream,
 * primes in order; only those of 5, 7 and 11 (dimensions 2-4) are used here. */
static uint8_t g_perm[28];
static int g_perm_ready;
static const uint32_t kPrime[5] = {2, 3, 5, 7, 11}, kPrimeSum[5] = {0, 2, 5, 10, 17};

static void halton_perms(void) {
    orc_rng rng = {0x853c49e6748fea9bULL, 0xda3e39cb94b95bdbULL}; /* Rng::default, rng.rs:15-22 */
    uint8_t perm[28];
    for (int i = 0; i < 5; i++) {
        uint8_t *p = perm + kPrimeSum[i];
        uint32_t n = kPrime[i];
        for (uint32_t j = 0; j < n; j++) p[j] = (uint8_t)j;
        for (uint32_t j = 0; j < n; j++) { /* shuffle_dims(.., 1), rng.rs:77-84 */
            uint32_t other = j + orc_rng_limit(&rng, n - j);
            uint8_t t = p[j]; p[j] = p[other]; p[other] = t;
        }
    }
    memcpy(g_perm, perm, sizeof perm);
    __atomic_store_n(&g_perm_ready, 1, __ATOMIC_RELEASE);
}

static float radical_inverse(uint32_t base, uint64_t num) { /* low_discrepancy.rs:7-21 */
    float inv = 1.0f / (float)base, inv_base_n = 1.0f;
    uint64_t digits = 0;
    while (num > 0) {
        uint64_t next = num / base, digit = num - next * base;
        digits = digits * base + digit;
        inv_base_n *= inv;
        num = next;
    }
    float v = (float)(double)digits * inv_base_n;
    return v < 1.0f - ORC_EPS ? v : 1.0f - ORC_EPS; /* T::ONE_MINUS_EPSILON, number.rs:209 */
}

static float scrambled_radical_inverse(uint32_t base, uint64_t num, const uint8_t *perm) { /* :43-62 */
    float inv = 1.0f / (float)base, inv_base_n = 1.0f;
    uint64_t digits = 0;
    while (num > 0) {
        uint64_t next = num / base, digit = num - next * base;
        digits = digits * base + perm[digit];
        inv_base_n *= inv;
        num = next;
    }
    float v = inv_base_n * ((float)(double)digits + inv * (float)perm[0] / (1.0f - inv));
    return v < 1.0f - ORC_EPS ? v : 1.0f - ORC_EPS; /* T::ONE_MINUS_EPSILON, number.rs:209 */
}

static uint64_t inverse_radical_inverse(uint32_t base, uint64_t inverse, uint32_t n_digits) { /* :31-40 */
    uint64_t index = 0;
    for (uint32_t i = 0; i < n_digits; i++) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}

static int64_t ext_gcd_x(int64_t a, int64_t b, int64_t *y) { /* halton.rs:119-126 */
    if (b == 0) { *y = 0; return 1; }
    int64_t yp, xp = ext_gcd_x(b, a % b, &yp);
    *y = xp - (a / b) * yp;
    return yp;
}
static uint64_t mult_inverse(int64_t a, int64_t n) { /* halton.rs:113-116 */
    int64_t y, x = ext_gcd_x(a, n, &y);
    int64_t r = x % n;
    return (uint64_t)(r < 0 ? r + n : r);
}

static void halton_sample(const rt_render_opts *o, uint32_t px, uint32_t py, uint32_t s, float u[5]) {
    if (!__atomic_load_n(&g_perm_ready, __ATOMIC_ACQUIRE)) halton_perms();
    uint32_t res[2] = {o->width, o->height}, scale[2], expo[2];
    for (int i = 0; i < 2; i++) { /* Halton::new, halton.rs:33-45: 2^j, 3^k >= min(res, 128) */
        uint32_t base = (uint32_t)i + 2, sc = 1, e = 0, lim = res[i] < 128 ? res[i] : 128;
        while (sc < lim) { sc *= base; e++; }
        scale[i] = sc; expo[i] = e;
    }
    uint64_t stride = (uint64_t)scale[0] * scale[1], offset = 0;
    if (stride > 1) { /* get_index_for_sample, halton.rs:62-86 */
        uint32_t pm[2] = {px % 128, py % 128};
        uint64_t mi[2] = {mult_inverse(scale[1], scale[0]), mult_inverse(scale[0], scale[1])};
        for (int i = 0; i < 2; i++) {
            uint64_t d = inverse_radical_inverse((uint32_t)i + 2, pm[i], expo[i]);
            offset += d * (stride / scale[i]) * mi[i];
        }
        offset %= stride;
    }
    uint64_t index = offset + (uint64_t)s * stride;
    u[0] = radical_inverse(2, index >> expo[0]); /* sample_dimension, halton.rs:90-106 */
    u[1] = radical_inverse(3, index / scale[1]);
    u[2] = scrambled_radical_inverse(5, index, g_perm + kPrimeSum[2]);
    u[3] = scrambled_radical_inverse(7, index, g_perm + kPrimeSum[3]);
    u[4] = scrambled_radical_inverse(11, index, g_perm + kPrimeSum[4]);
}

/* ---- Sobol' camera samples: colour/src/samplers/sobol.rs --------------------------------
 * The tables of data/src/sobol.rs are not stored here but derived: the generator matrices
 * of dimensions 0-4 from the Joe-Kuo direction numbers the file names as its source
 * (data/src/sobol.rs:57-63; 52-bit columns, MATRICES_32 keeps the top 32 bits), and the
 * two pixel-interval matrices (MATRICES / MATRICES_INV) from dimensions 0 and 1 by linear
 * algebra over GF(2): with resolution 2^m the top m bits of dimension 0 and of dimension 1
 * are the pixel (x, y); T maps an index to (x << m) | y.  MATRICES_INV[m-1] is the inverse
 * of T on the low 2m index bits and MATRICES[m-1][c] = T(1 << (2m + c)).
 * Deviations from the literal text, following pbrt-v3/v4: b = ((x << m) | y) ^ delta
 * (sobol.rs:58-59 xors delta into y only); a 1x1 frame uses the sample number as index
 * (sobol.rs:38-40 returns 0 for every sample); dimensions 0,1 / 2 / 3,4 feed film, time
 * and lens (global_sampler.rs:121-134 starts a 2D sample one dimension late). */
static const struct { uint32_t s, a, m[3]; } kJoeKuo[5] = {
    {0, 0, {0, 0, 0}}, {1, 0, {1, 0, 0}}, {2, 1, {1, 3, 0}}, {3, 1, {1, 3, 1}}, {3, 2, {1, 1, 1}}};

typedef struct {
    uint32_t m32[5][52]; /* MATRICES_32, dimensions 0-4 */
    uint64_t vdc[52];    /* MATRICES[m - 1]             */
    uint64_t inv[52];    /* MATRICES_INV[m - 1]         */
    uint32_t m;          /* log2 of the resolution      */
} sobol_tables;

static void sobol_matrices(uint32_t m32[5][52]) {
    for (int d = 0; d < 5; d++) {
        uint64_t mk[52];
        uint32_t s = kJoeKuo[d].s, a = kJoeKuo[d].a;
        for (uint32_t k = 0; k < 52; k++) {
            if (s == 0) mk[k] = 1; /* van der Corput */
            else if (k < s) mk[k] = kJoeKuo[d].m[k];
            else {
                uint64_t v = mk[k - s] ^ (mk[k - s] << s);
                for (uint32_t i = 1; i < s; i++)
                    if ((a >> (s - 1 - i)) & 1u) v ^= mk[k - i] << i;
                mk[k] = v;
            }
            m32[d][k] = (uint32_t)((mk[k] << (51 - k)) >> 20);
        }
    }
}

static uint64_t sobol_pixel_bits(const uint32_t m32[5][52], uint32_t j, uint32_t m) {
    return ((uint64_t)(m32[0][j] >> (32 - m)) << m) | (uint64_t)(m32[1][j] >> (32 - m));
}

static void sobol_setup(uint32_t width, uint32_t height, sobol_tables *t) {
    memset(t, 0, sizeof *t);
    sobol_matrices(t->m32);
    uint32_t res = width > height ? width : height, m = 0;
    while ((1u << m) < res) m++; /* round_up_pow_2 + log2usize, sobol.rs:15-21 */
    t->m = m;
    if (m == 0) return;
    uint32_t n = 2 * m;
    uint64_t img[52], pre[52];
    for (uint32_t j = 0; j < n; j++) { img[j] = sobol_pixel_bits(t->m32, j, m); pre[j] = 1ull << j; }
    int used[52] = {0};
    for (uint32_t c = 0; c < n; c++) { /* Gauss-Jordan: afterwards img[r] = 1 << c for one row r */
        uint32_t r = 0;
        while (r < n && (used[r] || !((img[r] >> c) & 1u))) r++;
        used[r] = 1; /* T is invertible, so a row always exists */
        for (uint32_t i = 0; i < n; i++)
            if (i != r && ((img[i] >> c) & 1u)) { img[i] ^= img[r]; pre[i] ^= pre[r]; }
    }
    for (uint32_t r = 0; r < n; r++)
        for (uint32_t c = 0; c < n; c++)
            if (img[r] == (1ull << c)) t->inv[c] = pre[r];
    for (uint32_t c = 0; 2 * m + c < 52; c++) t->vdc[c] = sobol_pixel_bits(t->m32, 2 * m + c, m);
}

static float sobol_f32(const sobol_tables *t, uint64_t index, int dim) { /* sobol.rs:111-129 */
    uint32_t v = 0;
    for (int i = 0; index; i++, index >>= 1)
        if (index & 1u) v ^= t->m32[dim][i];
    float f = (float)v * 2.3283064365386963e-10f;
    return f < 1.0f - ORC_EPS ? f : 1.0f - ORC_EPS;
}

static void sobol_sample(const rt_render_opts *o, uint32_t px, uint32_t py, uint32_t s, float u[5]) {
    static __thread sobol_tables t;
    static __thread uint32_t tw, th;
    if (tw != o->width || th != o->height) { sobol_setup(o->width, o->height, &t); tw = o->width; th = o->height; }
    uint64_t index = s;
    if (t.m) { /* get_index_for_sample, sobol.rs:31-74 */
        uint64_t delta = 0, frame = s;
        for (int c = 0; frame; frame >>= 1, c++)
            if (frame & 1u) delta ^= t.vdc[c];
        uint64_t b = (((uint64_t)px << t.m) | py) ^ delta;
        index = (uint64_t)s << (2 * t.m);
        for (int c = 0; b; b >>= 1, c++)
            if (b & 1u) index ^= t.inv[c];
    }
    const float res = (float)(1u << t.m), pix[2] = {(float)px, (float)py};
    for (int k = 0; k < 5; k++) u[k] = sobol_f32(&t, index, k);
    for (int k = 0; k < 2; k++) { /* sample_dimension, sobol.rs:88-92 */
        float v = u[k] * res + 0.0f - pix[k];
        v = v < 0.0f ? 0.0f : v;
        u[k] = v < 1.0f - ORC_EPS ? v : 1.0f - ORC_EPS;
    }
}

void orc_sobol_tables(uint32_t width, uint32_t height, uint32_t m32[5 * 52], uint64_t vdc[52], uint64_t inv[52]) {
    sobol_tables t;
    sobol_setup(width, height, &t);
    memcpy(m32, t.m32, sizeof t.m32);
    memcpy(vdc, t.vdc, sizeof t.vdc);
    memcpy(inv, t.inv, sizeof t.inv);
}

/* ---- pixel samplers: colour/src/samplers/{pixel_sampler,zero_two,stratified}.rs ----------
 * A PixelSampler makes all samples of a pixel in start_pixel from its own generator.  The
 * reference keeps one generator per sampler clone, so a pixel's samples depend on every
 * pixel rendered before it; here the generator is reseeded per pixel
 * (sequence = 2^63 | (seed * 2^32 + pixel)), which makes pixels independent.  Four sampled
 * dimensions (pbrt's default); the camera uses 1D dimension 0 (time) and 2D dimensions 0
 * (film) and 1 (lens), the rest only advance the generator.
 * buf: film x, film y, time, lens u, lens v, each `spp` floats. */
#define PIXEL_SAMPLER_DIMS 4

static void shuffle1(orc_rng *rng, float *a, uint32_t n) { /* shuffle_dims(.., 1), rng.rs:77-84 */
    for (uint32_t i = 0; i < n; i++) {
        uint32_t other = i + orc_rng_limit(rng, n - i);
        if (a) { float t = a[i]; a[i] = a[other]; a[other] = t; }
    }
}
static void shuffle2(orc_rng *rng, float *a, float *b, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t other = i + orc_rng_limit(rng, n - i);
        if (a) {
            float t = a[i]; a[i] = a[other]; a[other] = t;
            t = b[i]; b[i] = b[other]; b[other] = t;
        }
    }
}
static float u32_to_unit(uint32_t v) {
    float f = (float)v * 2.3283064365386963e-10f;
    return f < 1.0f - ORC_EPS ? f : 1.0f - ORC_EPS;
}

/* van_der_corput(1, n, ..), zero_two.rs:68-92; out may be NULL (generator advanced only) */
static void zero_two_1d(orc_rng *rng, float *out, uint32_t n) {
    uint32_t v = orc_rng_u32(rng);
    for (uint32_t i = 0; i < n; i++) { /* sample_gray_code, zero_two.rs:129-136 */
        if (out) out[i] = u32_to_unit(v);
        v ^= 0x80000000u >> __builtin_ctz(i + 1);
    }
    for (uint32_t i = 0; i < n; i++) (void)orc_rng_limit(rng, 1); /* one-element shuffles, :86-89 */
    shuffle1(rng, out, n);
}
/* second generator matrix of the (0,2) sequence (zero_two.rs:106-112): column c is row c
 * of Pascal's triangle mod 2, most significant bit first */
static uint32_t sobol2_column(uint32_t c) {
    uint32_t v = 0;
    for (uint32_t k = 0; k <= c; k++)
        if ((c & k) == k) v |= 0x80000000u >> k;
    return v;
}
/* sobol_2d(1, n, ..), zero_two.rs:95-126 */
static void zero_two_2d(orc_rng *rng, float *x, float *y, uint32_t n) {
    uint32_t vx = orc_rng_u32(rng), vy = orc_rng_u32(rng);
    for (uint32_t i = 0; i < n; i++) { /* sample_gray_code2, zero_two.rs:140-155 */
        if (x) { x[i] = u32_to_unit(vx); y[i] = u32_to_unit(vy); }
        uint32_t c = (uint32_t)__builtin_ctz(i + 1);
        vx ^= 0x80000000u >> c;
        vy ^= sobol2_column(c);
    }
    for (uint32_t i = 0; i < n; i++) (void)orc_rng_limit(rng, 1);
    shuffle2(rng, x, y, n);
}

/* stratified_sample_one + shuffle, stratified.rs:35-38,81-88 */
static void stratified_1d(orc_rng *rng, float *out, uint32_t n, int jitter) {
    float inv = 1.0f / (float)(int32_t)n;
    for (uint32_t i = 0; i < n; i++) {
        float delta = jitter ? orc_rng_f(rng) : 0.5f;
        float v = ((float)(int32_t)i + delta) * inv;
        if (out) out[i] = v < 1.0f - ORC_EPS ? v : 1.0f - ORC_EPS;
    }
    shuffle1(rng, out, n);
}
/* stratified_sample_two + shuffle, stratified.rs:40-49,91-116; the strata are the nx x ny
 * grid cells as in pbrt (stratified.rs:111-114 uses the linear index on both axes) */
static void stratified_2d(orc_rng *rng, float *x, float *y, uint32_t nx, uint32_t ny, int jitter) {
    float dx = 1.0f / (float)(int32_t)nx, dy = 1.0f / (float)(int32_t)ny;
    for (uint32_t iy = 0, i = 0; iy < ny; iy++)
        for (uint32_t ix = 0; ix < nx; ix++, i++) {
            float jx = jitter ? orc_rng_f(rng) : 0.5f;
            float jy = jitter ? orc_rng_f(rng) : 0.5f;
            float vx = ((float)(int32_t)ix + jx) * dx, vy = ((float)(int32_t)iy + jy) * dy;
            if (x) {
                x[i] = vx < 1.0f - ORC_EPS ? vx : 1.0f - ORC_EPS;
                y[i] = vy < 1.0f - ORC_EPS ? vy : 1.0f - ORC_EPS;
            }
        }
    shuffle2(rng, x, y, nx * ny);
}

static void pixel_samples(const rt_render_opts *o, uint64_t pixel, float *buf) {
    const uint32_t n = o->spp;
    orc_rng rng;
    orc_rng_seq(&rng, (1ull << 63) | ((o->seed << 32) + pixel));
    float *fx = buf, *fy = buf + n, *time = buf + 2 * (size_t)n, *lu = buf + 3 * (size_t)n, *lv = buf + 4 * (size_t)n;
    if (o->sampler == RT_SAMPLER_ZERO_TWO) {
        for (int d = 0; d < PIXEL_SAMPLER_DIMS; d++) zero_two_1d(&rng, d == 0 ? time : NULL, n);
        zero_two_2d(&rng, fx, fy, n);
        zero_two_2d(&rng, lu, lv, n);
    } else {
        const uint32_t nx = o->sampler_arg & 0xffffu, ny = n / nx;
        const int jitter = !(o->sampler_arg & RT_STRATIFIED_NO_JITTER);
        for (int d = 0; d < PIXEL_SAMPLER_DIMS; d++) stratified_1d(&rng, d == 0 ? time : NULL, n, jitter);
        stratified_2d(&rng, fx, fy, nx, ny, jitter);
        stratified_2d(&rng, lu, lv, nx, ny, jitter);
    }
}

/* the calling thread's most recent pixel; a worker thread releases it before it ends */
static __thread struct { float *buf; uint32_t cap; uint64_t pixel, seed; uint32_t sampler, arg, spp, width, valid; } g_pixel;
void orc_thread_release(void) {
    free(g_pixel.buf);
    memset(&g_pixel, 0, sizeof g_pixel);
}
static void pixel_sampler_sample(const rt_render_opts *o, uint32_t px, uint32_t py, uint32_t s, float u[5]) {
#define c g_pixel
    const uint64_t pixel = (uint64_t)py * o->width + px;
    if (!c.valid || c.pixel != pixel || c.seed != o->seed || c.sampler != o->sampler || c.arg != o->sampler_arg ||
        c.spp != o->spp || c.width != o->width) {
        if (c.cap < o->spp) {
            free(c.buf);
            c.buf = (float *)malloc((size_t)o->spp * 5 * sizeof(float));
            c.cap = o->spp;
        }
        pixel_samples(o, pixel, c.buf);
        c.pixel = pixel; c.seed = o->seed; c.sampler = o->sampler; c.arg = o->sampler_arg;
        c.spp = o->spp; c.width = o->width; c.valid = 1;
    }
    for (int k = 0; k < 5; k++) u[k] = c.buf[(size_t)k * o->spp + s];
}
#undef c

/* 0 when the options name a sampler and arguments the samplers accept */
int orc_sampler_check(const rt_render_opts *o) {
    if (o->sampler > RT_SAMPLER_STRATIFIED) return -1;
    if (o->sampler == RT_SAMPLER_ZERO_TWO && (o->spp & (o->spp - 1))) return -1;
    if (o->sampler == RT_SAMPLER_STRATIFIED) {
        uint32_t nx = o->sampler_arg & 0xffffu;
        if (!nx || o->spp % nx) return -1;
    }
    return 0;
}

void orc_camera_sample(const rt_render_opts *opts, uint32_t px, uint32_t py, uint32_t s, orc_rng *rng, float u[5]) {
    if (opts->sampler == RT_SAMPLER_HALTON) {
        halton_sample(opts, px, py, s, u);
    } else if (opts->sampler == RT_SAMPLER_SOBOL) {
        sobol_sample(opts, px, py, s, u);
    } else if (opts->sampler == RT_SAMPLER_ZERO_TWO || opts->sampler == RT_SAMPLER_STRATIFIED) {
        pixel_sampler_sample(opts, px, py, s, u);
    } else {
        for (int k = 0; k < 5; k++) u[k] = orc_rng_f(rng);
    }
}

/* known-answer hook for the tests: the camera sample of a sampler that draws nothing from the path's generator */
void orc_camera_sample_test(const rt_render_opts *opts, uint32_t px, uint32_t py, uint32_t s, float u[5]) {
    orc_rng rng = {0, 1};
    orc_camera_sample(opts, px, py, s, &rng, u);
}

/* path p = pixel * spp + sample; its PCG32 sequence index is seed * 2^32 + p.
 * Camera sample order: film x, film y, time, lens u, lens v. */
void orc_path_setup(const rt_camera *cam, const rt_render_opts *opts, uint64_t path, orc_rng *rng,
                    rt_ray *ray) {
    uint64_t pixel = path / opts->spp;
    uint32_t px = (uint32_t)(pixel % opts->width), py = (uint32_t)(pixel / opts->width);
    orc_rng_seq(rng, (opts->seed << 32) + path);
    float u[5];
    orc_camera_sample(opts, px, py, (uint32_t)(path % opts->spp), rng, u);
    orc_camera_ray(cam, (float)px + u[0], (float)py + u[1], u[3], u[4], u[2], ray);
}

/* ---- textures ---------------------------------------------------------------- */
static float perlin_noise(const orc_scene *s, v3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    const int32_t *px = s->perlin_perm, *py = s->perlin_perm + 256, *pz = s->perlin_perm + 512;
    float accum = 0.0f;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                int idx = px[(i + di) & 255] ^ py[(j + dj) & 255] ^ pz[(k + dk) & 255];
                const float *c = s->perlin_vec + 3 * idx;
                float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
                float dot = c[0] * wx + c[1] * wy + c[2] * wz;
                float fi = di ? uu : 1.0f - uu;
                float fj = dj ? vv : 1.0f - vv;
                float fk = dk ? ww : 1.0f - ww;
                accum += fi * fj * fk * dot;
            }
    return accum;
}
static float perlin_turb(const orc_scene *s, v3 p) {
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; i++) {
        accum += weight * perlin_noise(s, p);
        weight *= 0.5f;
        p = v3_scale(p, 2.0f);
    }
    return fabsf(accum);
}

static v3 texture_value(const orc_scene *s, uint32_t tex, float u, float v, v3 p) {
    for (int level = 0; level < 8; level++) {
        const rt_texture *t = &s->textures[tex];
        switch (t->kind) {
        case RT_TEX_CHECKER: {
            float sines = orc_sin(t->scale * p.x) * orc_sin(t->scale * p.y) * orc_sin(t->scale * p.z);
            tex = sines < 0.0f ? t->a : t->b;
            continue;
        }
        case RT_TEX_CHECKER_UV: {
            int cu = (int)floorf(u * t->scale), cv = (int)floorf(v * t->scale);
            tex = ((cu + cv) & 1) ? t->a : t->b;
            continue;
        }
        case RT_TEX_IMAGE: {
            const rt_image *im = &s->images[t->a];
            float uc = orc_clamp(u, 0.0f, 1.0f), vc = 1.0f - orc_clamp(v, 0.0f, 1.0f);
            int i = (int)(uc * (float)im->w), j = (int)(vc * (float)im->h);
            if (i >= (int)im->w) i = (int)im->w - 1;
            if (j >= (int)im->h) j = (int)im->h - 1;
            const uint8_t *px = im->rgb8 + 3 * ((size_t)j * im->w + (size_t)i);
            const float sc = 1.0f / 255.0f;
            return v3_make((float)px[0] * sc, (float)px[1] * sc, (float)px[2] * sc);
        }
        case RT_TEX_NOISE: {
            float g = 0.5f * (1.0f + orc_sin(t->scale * p.z + 10.0f * perlin_turb(s, p)));
            return v3_make(g, g, g);
        }
        default: return v3_make(t->rgb[0], t->rgb[1], t->rgb[2]);
        }
    }
    return v3_make(0.0f, 0.0f, 0.0f);
}

/* ---- scattering --------------------------------------------------------------- */
static v3 random_in_unit_sphere(orc_rng *rng) {
    for (;;) {
        float x = 2.0f * orc_rng_f(rng) - 1.0f;
        float y = 2.0f * orc_rng_f(rng) - 1.0f;
        float z = 2.0f * orc_rng_f(rng) - 1.0f;
        if (x * x + y * y + z * z < 1.0f) return v3_make(x, y, z);
    }
}
static v3 reflect(v3 v, v3 n) { return v3_sub(v, v3_scale(n, 2.0f * v3_dot(v, n))); }

/* One path.  Returns the number of closest-hit segments traced; *occlusion (may be NULL)
 * receives the number of shadow rays.
 * With RT_RENDER_DIRECT_LIGHT every hit on a Lambertian surface samples one point on one light
 * and sends spawn_ray_to(point) through the any-hit traversal; the draws follow the scatter
 * direction's (light number only when there are several lights, then the two coordinates of the
 * point), and a light the path reaches by its very next segment adds nothing (its share was
 * taken by the shadow ray). */
uint32_t orc_trace_path(const orc_scene *s, const rt_camera *cam, const rt_render_opts *opts,
                        uint64_t path, float rgb[3], uint64_t counters[9], uint32_t *occlusion) {
    orc_rng rng;
    rt_ray ray;
    orc_path_setup(cam, opts, path, &rng, &ray);
    v3 beta = v3_make(1.0f, 1.0f, 1.0f);
    v3 L = v3_make(0.0f, 0.0f, 0.0f);
    uint32_t segments = 0, shadow_rays = 0;
    const uint32_t n_all_lights = s->n_lights + s->n_distant; /* rectangles, then distant lights */
    const int direct = (opts->flags & RT_RENDER_DIRECT_LIGHT) && n_all_lights;
    int skip_emission = 0; /* the previous vertex sampled the lights */
    for (uint32_t depth = 0; depth < opts->max_depth; depth++) {
        orc_surf h;
        segments++;
        if (!orc_closest(s, &ray, &h, counters)) {
            L = v3_add(L, v3_make(beta.x * s->background[0], beta.y * s->background[1],
                                  beta.z * s->background[2]));
            break;
        }
        v3 d = v3_make(ray.d[0], ray.d[1], ray.d[2]);
        int front = v3_dot(d, h.n) < 0.0f;
        v3 nf = front ? h.n : v3_neg(h.n);
        const rt_material *m = &s->materials[s->prims[h.prim].material];
        v3 dir;
        if (m->kind == RT_MAT_DIFFUSE_LIGHT) {
            v3 e = texture_value(s, m->texture, h.u, h.v, h.p);
            /* a light the shadow ray of the vertex before has already accounted for adds nothing;
             * emitters the light list does not hold (spheres, boxes, transformed or textured
             * rectangles) are reached by scattering only and keep their emission */
            if (!(skip_emission && s->is_light[h.prim])) L = v3_add(L, v3_make(beta.x * e.x, beta.y * e.y, beta.z * e.z));
            break;
        } else if (m->kind == RT_MAT_LAMBERTIAN) {
            v3 r = random_in_unit_sphere(&rng);
            dir = v3_add(nf, r);
            if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = nf;
            dir = v3_norm(dir);
            v3 a = texture_value(s, m->texture, h.u, h.v, h.p);
            if (direct) {
                uint32_t li = 0;
                if (n_all_lights > 1) {
                    li = (uint32_t)(orc_rng_f(&rng) * (float)n_all_lights);
                    if (li >= n_all_lights) li = n_all_lights - 1;
                }
                float u1 = orc_rng_f(&rng), u2 = orc_rng_f(&rng); /* drawn for every kind of light */
                if (li >= s->n_lights) {
                    /* a distant light (test/test.pbrt:16-17): the shadow ray is spawn_ray_to the point
                     * distant_reach away in the light's direction, outside the scene; what arrives is
                     * albedo / pi * radiance * cos * n_lights */
                    const rt_distant_light *dl = &s->distant[li - s->n_lights];
                    v3 w = v3_make(dl->w[0], dl->w[1], dl->w[2]);
                    v3 q = v3_add(h.p, v3_scale(w, s->distant_reach));
                    rt_ray sr;
                    orc_spawn_ray_to(h.p, h.perr, h.n, q, ray.time, &sr);
                    v3 sd = v3_make(sr.d[0], sr.d[1], sr.d[2]);
                    float d2 = v3_dot(sd, sd), cs = v3_dot(w, nf);
                    if (cs > 0.0f && d2 > 0.0f) {
                        shadow_rays++;
                        if (!orc_any(s, &sr)) {
                            float g = cs * ((float)n_all_lights * 0.318309886f);
                            L = v3_add(L, v3_make(beta.x * a.x * dl->rgb[0] * g, beta.y * a.y * dl->rgb[1] * g,
                                                  beta.z * a.z * dl->rgb[2] * g));
                        }
                    }
                    goto sampled;
                }
                const rt_prim_ref *lp = &s->prims[s->lights[li]];
                const rt_rect *lr = &s->rects[lp->index];
                const rt_texture *lt = &s->textures[s->materials[lp->material].texture];
                float qa = lr->a0 + u1 * (lr->a1 - lr->a0), qb = lr->b0 + u2 * (lr->b1 - lr->b0);
                v3 q = lr->axis == 0 ? v3_make(lr->k, qa, qb)
                                     : (lr->axis == 1 ? v3_make(qa, lr->k, qb) : v3_make(qa, qb, lr->k));
                rt_ray sr;
                orc_spawn_ray_to(h.p, h.perr, h.n, q, ray.time, &sr);
                v3 sd = v3_make(sr.d[0], sr.d[1], sr.d[2]);
                float d2 = v3_dot(sd, sd), cs = v3_dot(sd, nf), cl = fabsf(v3_get(sd, (int)lr->axis));
                if (cs > 0.0f && cl > 0.0f && d2 > 0.0f) {
                    shadow_rays++;
                    if (!orc_any(s, &sr)) {
                        /* albedo / pi * emitted * (cs / |d|) (cl / |d|) / |d|^2 * area * n_lights */
                        float area = (lr->a1 - lr->a0) * (lr->b1 - lr->b0);
                        float scale = area * (float)n_all_lights * 0.318309886f;
                        float g = (cs * cl) / (d2 * d2) * scale;
                        L = v3_add(L, v3_make(beta.x * a.x * lt->rgb[0] * g, beta.y * a.y * lt->rgb[1] * g,
                                              beta.z * a.z * lt->rgb[2] * g));
                    }
                }
            }
            sampled:
            skip_emission = direct;
            beta = v3_make(beta.x * a.x, beta.y * a.y, beta.z * a.z);
        } else if (m->kind == RT_MAT_METALLIC) {
            v3 r = random_in_unit_sphere(&rng);
            dir = v3_add(reflect(d, nf), v3_scale(r, m->fuzz));
            if (!(v3_dot(dir, nf) > 0.0f)) break; /* absorbed */
            dir = v3_norm(dir);
            v3 a = texture_value(s, m->texture, h.u, h.v, h.p);
            beta = v3_make(beta.x * a.x, beta.y * a.y, beta.z * a.z);
            skip_emission = 0;
        } else if (m->kind == RT_MAT_DIELECTRIC) {
            float ratio = front ? 1.0f / m->ior : m->ior;
            float cos_t = sel_min(v3_dot(v3_neg(d), nf), 1.0f);
            float sin_t = sqrtf(1.0f - cos_t * cos_t);
            int reflect_it = ratio * sin_t > 1.0f;
            if (!reflect_it) { /* Schlick */
                float r0 = (1.0f - ratio) / (1.0f + ratio);
                r0 = r0 * r0;
                float c = 1.0f - cos_t;
                float refl = r0 + (1.0f - r0) * (c * c * c * c * c);
                reflect_it = refl > orc_rng_f(&rng);
            }
            if (reflect_it) {
                dir = reflect(d, nf);
            } else {
                v3 perp = v3_scale(v3_add(d, v3_scale(nf, cos_t)), ratio);
                float k = 1.0f - v3_dot(perp, perp);
                v3 par = v3_scale(nf, -sqrtf(fabsf(k)));
                dir = v3_add(perp, par);
            }
            dir = v3_norm(dir);
            skip_emission = 0;
        } else { /* RT_MAT_ISOTROPIC */
            dir = v3_norm(random_in_unit_sphere(&rng));
            v3 a = texture_value(s, m->texture, h.u, h.v, h.p);
            beta = v3_make(beta.x * a.x, beta.y * a.y, beta.z * a.z);
            skip_emission = 0;
        }
        v3 no = orc_offset_origin(h.p, h.perr, h.n, dir); /* spawn_ray, interaction.rs:86-99 */
        ray.o[0] = no.x; ray.o[1] = no.y; ray.o[2] = no.z;
        ray.d[0] = dir.x; ray.d[1] = dir.y; ray.d[2] = dir.z;
        ray.t_max = INFINITY;
    }
    rgb[0] = L.x; rgb[1] = L.y; rgb[2] = L.z;
    if (occlusion) *occlusion = shadow_rays;
    return segments;
}
