/*
 * orc_bvh.c -- BVH construction, flattening and traversal of the CPU oracle
 * (test infrastructure only; see orc_math.h for the parity statement).
 *
 * Follows shapes/src/bvh.rs: construct :124-151, recursive_build :154-238,
 * middle :261-276, equal_counts :282-295, sah :300-382, hlbvh_build :385-449,
 * emit_lvbh :452-538, build_upper_sah :541-658, flatten_tree :661-691,
 * intersect :736-799, does_intersect :801-863, partition :881-900,
 * morton_encode3 :903-919, radix_sort :923-966; the slab test is
 * geometry/src/bounds3.rs:212-238.  Where the literal text cannot run (see the
 * comment at each place) the behaviour it transcribes is used instead.
 */
#include <stdlib.h>
#include <time.h>

#include "orc_internal.h"

typedef struct {
    uint32_t prim;
    aabb b;
    v3 c;
} pinfo; /* BVHPrimitiveInfo, bvh.rs:42-51 */

typedef struct bnode {
    aabb b;
    struct bnode *child[2];
    int axis;
    uint32_t first, count;
} bnode; /* BVHBuildNode, bvh.rs:54-66 */

typedef struct {
    orc_scene *s;
    pinfo *info;
    uint32_t *ordered;
    uint32_t n_ordered;
    uint32_t total_nodes;
    uint32_t max_prims;
    int method;
    bnode *pool;
    uint32_t pool_used, pool_cap;
} builder;

static bnode *new_node(builder *B) {
    if (B->pool_used == B->pool_cap) return NULL;
    bnode *n = &B->pool[B->pool_used++];
    memset(n, 0, sizeof *n);
    return n;
}

/* corrected surface_area: 2(dx*dy + dx*dz + dy*dz) (bounds3.rs:114-117 has 2*(d.d)) */
static float surface_area(aabb b) {
    v3 d = v3_sub(b.hi, b.lo);
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}
static int maximum_extent(aabb b) { /* bounds3.rs:126-135 */
    v3 d = v3_sub(b.hi, b.lo);
    if (d.x > d.y && d.x > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}
static float offset_dim(aabb b, v3 p, int dim) { /* bounds3.rs:144-157 */
    float o = v3_get(p, dim) - v3_get(b.lo, dim);
    float hi = v3_get(b.hi, dim), lo = v3_get(b.lo, dim);
    if (hi > lo) o /= hi - lo;
    return o;
}

static bnode *make_leaf(builder *B, uint32_t start, uint32_t end, aabb bounds) {
    bnode *n = new_node(B); /* create_leaf_node, bvh.rs:241-258 with start..start+n */
    if (!n) return NULL;
    n->first = B->n_ordered;
    n->count = end - start;
    for (uint32_t i = start; i < end; i++) B->ordered[B->n_ordered++] = B->info[i].prim;
    n->b = bounds;
    B->total_nodes++;
    return n;
}

/* strict total order on (centroid[dim], primitive number): makes the
 * "unstable" selection of bvh.rs:288 deterministic */
static int key_less(const pinfo *a, const pinfo *b, int dim) {
    float ka = v3_get(a->c, dim), kb = v3_get(b->c, dim);
    if (ka < kb) return 1;
    if (kb < ka) return 0;
    return a->prim < b->prim;
}
static void swap_info(pinfo *a, pinfo *b) { pinfo t = *a; *a = *b; *b = t; }

/* select_nth: after the call a[k] holds the k-th smallest key, smaller keys
 * are left of it, larger right.  Median-of-three quickselect with Hoare
 * partition; the GPU library's host builder runs the same sequence of swaps. */
static void select_nth(pinfo *a, uint32_t n, uint32_t k, int dim) {
    uint32_t lo = 0, hi = n; /* [lo, hi) */
    while (hi - lo > 2) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (key_less(&a[mid], &a[lo], dim)) swap_info(&a[mid], &a[lo]);
        if (key_less(&a[hi - 1], &a[lo], dim)) swap_info(&a[hi - 1], &a[lo]);
        if (key_less(&a[hi - 1], &a[mid], dim)) swap_info(&a[hi - 1], &a[mid]);
        pinfo pivot = a[mid];
        uint32_t i = lo, j = hi - 1;
        for (;;) {
            while (key_less(&a[i], &pivot, dim)) i++;
            while (key_less(&pivot, &a[j], dim)) j--;
            if (i >= j) break;
            swap_info(&a[i], &a[j]);
            i++;
            j--;
        }
        /* now [lo, j] <= pivot <= [j+1, hi) */
        if (k <= j) hi = j + 1; else lo = j + 1;
    }
    if (hi - lo == 2 && key_less(&a[lo + 1], &a[lo], dim)) swap_info(&a[lo], &a[lo + 1]);
}

/* partition(), bvh.rs:881-900, same swap sequence */
typedef int (*pred_fn)(const pinfo *, const void *);
static uint32_t partition_info(pinfo *a, uint32_t len, pred_fn f, const void *ctx) {
    if (len == 0) return 0;
    uint32_t left = 0, right = len - 1;
    for (;;) {
        while (left < len && f(&a[left], ctx)) left++;
        while (right > 0 && !f(&a[right], ctx)) right--;
        if (left >= right) return left;
        swap_info(&a[left], &a[right]);
    }
}

typedef struct { int dim; float mid; } mid_ctx;
static int pred_mid(const pinfo *p, const void *c) {
    const mid_ctx *m = (const mid_ctx *)c;
    return v3_get(p->c, m->dim) < m->mid;
}
typedef struct { int dim; aabb cb; int split; } sah_ctx;
static int bucket_of(aabb cb, v3 c, int dim) {
    /* bvh.rs:325-329 with the multiplication done before the cast */
    int b = (int)(12.0f * offset_dim(cb, c, dim));
    if (b >= 12) b = 11;
    if (b < 0) b = 0;
    return b;
}
static int pred_sah(const pinfo *p, const void *c) {
    const sah_ctx *m = (const sah_ctx *)c;
    return bucket_of(m->cb, p->c, m->dim) <= m->split;
}

static uint32_t equal_counts(pinfo *a, uint32_t n, int dim) { /* bvh.rs:282-295 */
    uint32_t k = n / 2;
    select_nth(a, n, k, dim);
    return k;
}

/* Returns the split position relative to `a`, or -1 for "make a leaf". */
static int64_t sah_split(builder *B, pinfo *a, uint32_t n, aabb cb, int dim, aabb bounds) {
    if (n <= 4) return equal_counts(a, n, dim); /* bvh.rs:307-309 */
    uint32_t count[12];
    aabb bb[12];
    for (int i = 0; i < 12; i++) { count[i] = 0; bb[i] = aabb_empty(); }
    for (uint32_t i = 0; i < n; i++) {
        int b = bucket_of(cb, a[i].c, dim);
        count[b]++;
        bb[b] = aabb_union(bb[b], a[i].b);
    }
    float cost[11];
    for (int i = 0; i < 11; i++) { /* bvh.rs:335-356, counts summed on both sides */
        aabb b0 = aabb_empty(), b1 = aabb_empty();
        uint32_t c0 = 0, c1 = 0;
        for (int j = 0; j <= i; j++) { b0 = aabb_union(b0, bb[j]); c0 += count[j]; }
        for (int j = i + 1; j < 12; j++) { b1 = aabb_union(b1, bb[j]); c1 += count[j]; }
        cost[i] = 0.125f + ((float)c0 * surface_area(b0) + (float)c1 * surface_area(b1)) /
                               surface_area(bounds);
    }
    float min_cost = cost[0]; /* bvh.rs:358-366 over the 11 written slots */
    int min_idx = 0;
    for (int i = 1; i < 11; i++)
        if (cost[i] < min_cost) { min_cost = cost[i]; min_idx = i; }
    if (n > B->max_prims || min_cost < (float)n) { /* bvh.rs:368-370 compared as float */
        sah_ctx ctx = {dim, cb, min_idx};
        return partition_info(a, n, pred_sah, &ctx);
    }
    return -1;
}

static bnode *recursive_build(builder *B, uint32_t start, uint32_t end) {
    pinfo *a = B->info + start;
    uint32_t n = end - start;
    aabb bounds = aabb_empty();
    for (uint32_t i = 0; i < n; i++) bounds = aabb_union(bounds, a[i].b);
    if (n == 1) return make_leaf(B, start, end, bounds);
    aabb cb = aabb_empty();
    for (uint32_t i = 0; i < n; i++) cb = aabb_union_pt(cb, a[i].c);
    int dim = maximum_extent(cb);
    if (v3_get(cb.hi, dim) == v3_get(cb.lo, dim)) return make_leaf(B, start, end, bounds);
    int64_t mid;
    switch (B->method) {
    case RT_SPLIT_MIDDLE: { /* bvh.rs:261-276 */
        mid_ctx ctx = {dim, (v3_get(cb.lo, dim) + v3_get(cb.hi, dim)) / 2.0f};
        mid = partition_info(a, n, pred_mid, &ctx);
        if (mid == 0 || mid == (int64_t)n) mid = equal_counts(a, n, dim);
        break;
    }
    case RT_SPLIT_EQUAL_COUNTS: mid = equal_counts(a, n, dim); break;
    default:
        mid = sah_split(B, a, n, cb, dim, bounds);
        if (mid < 0) return make_leaf(B, start, end, bounds);
        if (mid == 0 || mid == (int64_t)n) mid = equal_counts(a, n, dim);
        break;
    }
    bnode *node = new_node(B);
    if (!node) return NULL;
    bnode *c0 = recursive_build(B, start, start + (uint32_t)mid);
    bnode *c1 = recursive_build(B, start + (uint32_t)mid, end);
    if (!c0 || !c1) return NULL;
    node->child[0] = c0;
    node->child[1] = c1;
    node->axis = dim;
    node->b = aabb_union(c0->b, c1->b);
    B->total_nodes++;
    return node;
}

/* ---- HLBVH ----------------------------------------------------------------- */
typedef struct { uint32_t prim_index; uint32_t code; } morton_prim;

static uint32_t left_shift3(uint32_t x) { /* bvh.rs:903-912 */
    if (x == (1u << 10)) x -= 1;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
static uint32_t morton3(float x, float y, float z) { /* bvh.rs:915-919 */
    return (left_shift3((uint32_t)z) << 2) | (left_shift3((uint32_t)y) << 1) |
           left_shift3((uint32_t)x);
}
static void radix_sort(morton_prim *v, morton_prim *tmp, uint32_t n) { /* bvh.rs:923-966 */
    morton_prim *in = v, *out = tmp;
    for (int pass = 0; pass < 5; pass++) {
        int low = pass * 6;
        uint32_t bucket[64] = {0}, index[64];
        for (uint32_t i = 0; i < n; i++) bucket[(in[i].code >> low) & 63u]++;
        index[0] = 0;
        for (int i = 1; i < 64; i++) index[i] = index[i - 1] + bucket[i - 1];
        for (uint32_t i = 0; i < n; i++) out[index[(in[i].code >> low) & 63u]++] = in[i];
        morton_prim *t = in; in = out; out = t;
    }
    if (in != v) memcpy(v, in, (size_t)n * sizeof *v);
}

static bnode *emit_lbvh(builder *B, const pinfo *by_prim, const morton_prim *mp, uint32_t n,
                        int bit) { /* bvh.rs:452-538 */
    if (bit == -1 || n < B->max_prims) {
        bnode *leaf = new_node(B);
        if (!leaf) return NULL;
        B->total_nodes++;
        aabb b = aabb_empty();
        leaf->first = B->n_ordered;
        leaf->count = n;
        for (uint32_t i = 0; i < n; i++) {
            B->ordered[B->n_ordered++] = by_prim[mp[i].prim_index].prim;
            b = aabb_union(b, by_prim[mp[i].prim_index].b);
        }
        leaf->b = b;
        return leaf;
    }
    uint32_t mask = 1u << bit;
    if ((mp[0].code & mask) == (mp[n - 1].code & mask))
        return emit_lbvh(B, by_prim, mp, n, bit - 1);
    uint32_t lo = 0, hi = n - 1; /* binary search for the bit flip */
    while (lo + 1 != hi) {
        uint32_t mid = (lo + hi) / 2;
        if ((mp[lo].code & mask) == (mp[mid].code & mask)) lo = mid; else hi = mid;
    }
    uint32_t split = hi; /* split <= n-1 is legal (bvh.rs:512 asserts < n-1) */
    bnode *node = new_node(B);
    if (!node) return NULL;
    B->total_nodes++;
    bnode *c0 = emit_lbvh(B, by_prim, mp, split, bit - 1);
    bnode *c1 = emit_lbvh(B, by_prim, mp + split, n - split, bit - 1);
    if (!c0 || !c1) return NULL;
    node->child[0] = c0;
    node->child[1] = c1;
    node->axis = bit % 3;
    node->b = aabb_union(c0->b, c1->b);
    return node;
}

static uint32_t partition_roots(bnode **a, uint32_t len, aabb cb, int dim, int split) {
    if (len == 0) return 0;
    uint32_t left = 0, right = len - 1;
    for (;;) {
#define ROOT_PRED(nd) (bucket_of(cb, v3_scale(v3_add((nd)->b.lo, (nd)->b.hi), 0.5f), dim) <= split)
        while (left < len && ROOT_PRED(a[left])) left++;
        while (right > 0 && !ROOT_PRED(a[right])) right--;
#undef ROOT_PRED
        if (left >= right) return left;
        bnode *t = a[left]; a[left] = a[right]; a[right] = t;
    }
}

static bnode *build_upper_sah(builder *B, bnode **roots, uint32_t n) { /* bvh.rs:541-658 */
    if (n == 1) return roots[0];
    bnode *node = new_node(B);
    if (!node) return NULL;
    B->total_nodes++;
    aabb bounds = aabb_empty(), cb = aabb_empty();
    for (uint32_t i = 0; i < n; i++) {
        bounds = aabb_union(bounds, roots[i]->b);
        cb = aabb_union_pt(cb, v3_scale(v3_add(roots[i]->b.lo, roots[i]->b.hi), 0.5f));
    }
    int dim = maximum_extent(cb);
    uint32_t mid;
    if (v3_get(cb.hi, dim) == v3_get(cb.lo, dim)) {
        mid = n / 2; /* the reference asserts here (bvh.rs:570) */
    } else {
        uint32_t count[12];
        aabb bb[12];
        for (int i = 0; i < 12; i++) { count[i] = 0; bb[i] = aabb_empty(); }
        for (uint32_t i = 0; i < n; i++) {
            int b = bucket_of(cb, v3_scale(v3_add(roots[i]->b.lo, roots[i]->b.hi), 0.5f), dim);
            count[b]++;
            bb[b] = aabb_union(bb[b], roots[i]->b);
        }
        float min_cost = 0.0f;
        int min_idx = 0;
        for (int i = 0; i < 11; i++) {
            aabb b0 = aabb_empty(), b1 = aabb_empty();
            uint32_t c0 = 0, c1 = 0;
            for (int j = 0; j <= i; j++) { b0 = aabb_union(b0, bb[j]); c0 += count[j]; }
            for (int j = i + 1; j < 12; j++) { b1 = aabb_union(b1, bb[j]); c1 += count[j]; }
            float cost = 0.125f + ((float)c0 * surface_area(b0) + (float)c1 * surface_area(b1)) /
                                      surface_area(bounds);
            if (i == 0 || cost < min_cost) { min_cost = cost; min_idx = i; }
        }
        mid = partition_roots(roots, n, cb, dim, min_idx);
        if (mid == 0 || mid == n) mid = n / 2;
    }
    bnode *c0 = build_upper_sah(B, roots, mid);
    bnode *c1 = build_upper_sah(B, roots + mid, n - mid);
    if (!c0 || !c1) return NULL;
    node->child[0] = c0;
    node->child[1] = c1;
    node->axis = dim;
    node->b = aabb_union(c0->b, c1->b);
    return node;
}

static bnode *hlbvh_build(builder *B, uint32_t n) { /* bvh.rs:385-449 */
    aabb cb = aabb_empty();
    for (uint32_t i = 0; i < n; i++) cb = aabb_union_pt(cb, B->info[i].c);
    morton_prim *mp = (morton_prim *)malloc((size_t)n * sizeof *mp);
    morton_prim *tmp = (morton_prim *)malloc((size_t)n * sizeof *tmp);
    bnode **roots = (bnode **)malloc((size_t)n * sizeof *roots);
    bnode *result = NULL;
    if (!mp || !tmp || !roots) goto done;
    for (uint32_t i = 0; i < n; i++) { /* scale first, then cast (bvh.rs:400-406) */
        float ox = offset_dim(cb, B->info[i].c, 0) * 1024.0f;
        float oy = offset_dim(cb, B->info[i].c, 1) * 1024.0f;
        float oz = offset_dim(cb, B->info[i].c, 2) * 1024.0f;
        mp[i].prim_index = i; /* place in info; the primitive's number is info[i].prim */
        mp[i].code = morton3(ox, oy, oz);
    }
    radix_sort(mp, tmp, n);
    uint32_t n_roots = 0;
    for (uint32_t start = 0, end = 1; end <= n; end++) { /* bvh.rs:414-431, bounded */
        const uint32_t mask = 0x3ffc0000u;
        if (end == n || ((mp[start].code & mask) != (mp[end].code & mask))) {
            bnode *r = emit_lbvh(B, B->info, mp + start, end - start, 29 - 12);
            if (!r) goto done;
            roots[n_roots++] = r;
            start = end;
        }
    }
    result = build_upper_sah(B, roots, n_roots);
done:
    free(mp);
    free(tmp);
    free(roots);
    return result;
}

/* ---- flatten, bvh.rs:661-691 ------------------------------------------------ */
static uint32_t flatten(orc_scene *s, const bnode *n) {
    uint32_t idx = s->n_nodes++;
    rt_linear_node *ln = &s->nodes[idx];
    memset(ln, 0, sizeof *ln);
    ln->lo[0] = n->b.lo.x; ln->lo[1] = n->b.lo.y; ln->lo[2] = n->b.lo.z;
    ln->hi[0] = n->b.hi.x; ln->hi[1] = n->b.hi.y; ln->hi[2] = n->b.hi.z;
    if (n->child[0]) {
        ln->axis = (uint8_t)n->axis;
        ln->primitive_count = 0;
        flatten(s, n->child[0]);
        uint32_t second = flatten(s, n->child[1]);
        s->nodes[idx].child_offset = second;
    } else {
        ln->child_offset = n->first;
        ln->primitive_count = (uint16_t)n->count;
    }
    return idx;
}

/* deepest stack the traversal loop can need (+1), over both visiting orders */
uint32_t orc_tree_depth(const rt_linear_node *nodes, uint32_t n_nodes) {
    if (n_nodes == 0) return 0;
    /* iterative pre-order walk carrying the pending-stack depth */
    uint32_t *stk = (uint32_t *)malloc((size_t)n_nodes * 2 * sizeof *stk);
    if (!stk) return 0xffffffffu;
    uint32_t top = 0, best = 0;
    stk[top++] = 0; stk[top++] = 0;
    while (top) {
        uint32_t depth = stk[--top], idx = stk[--top];
        if (depth > best) best = depth;
        if (nodes[idx].primitive_count == 0) {
            /* descending into either child leaves one more entry on the stack */
            stk[top++] = idx + 1; stk[top++] = depth + 1;
            stk[top++] = nodes[idx].child_offset; stk[top++] = depth + 1;
        }
    }
    free(stk);
    return best + 1;
}

/* Media that enclose the scene.  The scene files put a thin fog around everything
 * (test/week.toml:86-93: a Volume whose boundary is a sphere of radius 5000 about the origin).
 * As a primitive of the tree such a medium is a child of the root that every ray meets, and it
 * stretches the root box to its own size.  Rule, the same in the library: a medium bounded by a
 * sphere encloses the scene when the box around all primitives that are not sphere-bounded
 * media lies strictly inside its sphere (all eight corners, distances in double precision).
 * The first ORC_MAX_ENCLOSING such media in primitive order are left out of the tree and tested
 * once per ray before the traversal starts, in that order; a hit shortens the ray like any
 * other.  The reference has no media at all, so nothing of it is re-ordered by this; with
 * ORC_NO_ENCLOSING=1 the tree is built over every primitive as BVH::new does
 * (shapes/src/bvh.rs:108-151), and hits, any-hit answers and frames must come out the same. */
static void find_enclosing(orc_scene *s, const aabb *bounds) {
    s->n_enclosing = 0;
    /* ORC_NO_ENCLOSING: the literal build, BVH::new over all primitives (shapes/src/bvh.rs:108-151);
     * tests compare results with and without */
    if (getenv("ORC_NO_ENCLOSING")) return;
    aabb others = aabb_empty();
    int any = 0;
    for (uint32_t i = 0; i < s->n_prims; i++) {
        const rt_prim_ref *p = &s->prims[i];
        if ((p->kind & 0xffu) == RT_MEDIUM && s->media[p->index].boundary_kind == RT_BOUNDARY_SPHERE) continue;
        others = aabb_union(others, bounds[i]);
        any = 1;
    }
    if (!any) return;
    for (uint32_t i = 0; i < s->n_prims && s->n_enclosing < ORC_MAX_ENCLOSING; i++) {
        const rt_prim_ref *p = &s->prims[i];
        if ((p->kind & 0xffu) != RT_MEDIUM || s->media[p->index].boundary_kind != RT_BOUNDARY_SPHERE) continue;
        const rt_sphere *sp = &s->spheres[s->media[p->index].boundary_index];
        int inside = 1;
        for (int c = 0; c < 8 && inside; c++) {
            double dx = (double)((c & 1) ? others.hi.x : others.lo.x) - (double)sp->c[0];
            double dy = (double)((c & 2) ? others.hi.y : others.lo.y) - (double)sp->c[1];
            double dz = (double)((c & 4) ? others.hi.z : others.lo.z) - (double)sp->c[2];
            inside = dx * dx + dy * dy + dz * dz < (double)sp->r * (double)sp->r;
        }
        if (inside) s->enclosing[s->n_enclosing++] = i;
    }
}

/* Balls.  The scene files hold a glass sphere filled with fog (test/week.toml: a Volume bounded
 * by the sphere that is also a dielectric primitive); a path that gets inside stays there for
 * dozens of segments, each of which would walk the tree from the root only to find the two
 * primitives it started in.  A ball is a medium bounded by a sphere (no instance transform, not
 * one of the enclosing media), optionally with one sphere primitive of bit-identical centre and
 * positive radius, such that no other primitive of the tree comes nearer to the sphere than a
 * safety distance: for a sphere-like primitive the distance between the surfaces, for any other
 * the distance from the ball's centre to the primitive's box minus the radius, must exceed
 * 0.002 x (radius + half the diagonal of the other's box).  A ray that starts inside such a ball
 * (|o - c|^2 < (0.998 r)^2) meets everything else only after it has left the ball, so its
 * closest hit is decided by the ball's own primitives: see ball_resolve.  The first
 * ORC_MAX_BALLS balls in primitive order are kept, out of the first 64 such media.  Same rule, same arithmetic in the library. */
static void find_balls(orc_scene *s, const aabb *bounds) {
    s->n_balls = 0;
    if (getenv("ORC_NO_BALLS")) return; /* tests compare results with and without the short cut */
    uint32_t examined = 0;
    for (uint32_t i = 0; i < s->n_prims && s->n_balls < ORC_MAX_BALLS; i++) {
        const rt_prim_ref *p = &s->prims[i];
        if ((p->kind & 0xffu) != RT_MEDIUM) continue;
        const rt_medium *m = &s->media[p->index];
        if (m->boundary_kind != RT_BOUNDARY_SPHERE || m->xform != 0) continue;
        int skip = 0;
        for (uint32_t e = 0; e < s->n_enclosing; e++) skip |= s->enclosing[e] == i;
        if (skip) continue;
        const rt_sphere *sp = &s->spheres[m->boundary_index];
        if (!(sp->r > 0.0f)) continue;
        if (++examined > 64) break; /* every candidate is held against every primitive: keep that bounded */
        orc_ball b;
        memcpy(b.c, sp->c, sizeof b.c);
        b.r = sp->r;
        const double rin = (double)sp->r * 0.998;
        b.r_in2 = (float)(rin * rin);
        b.n = 1;
        b.prim[0] = i;
        b.sphere = -1;
        int ok = 1;
        for (uint32_t j = 0; j < s->n_prims && ok; j++) {
            if (j == i) continue;
            int enclosing = 0;
            for (uint32_t e = 0; e < s->n_enclosing; e++) enclosing |= s->enclosing[e] == j;
            if (enclosing) continue; /* not in the tree; every ray tests them first anyway */
            const rt_prim_ref *q = &s->prims[j];
            const uint32_t kind = q->kind & 0xffu;
            const rt_sphere *other = NULL;
            if (kind == RT_SPHERE) other = &s->spheres[q->index];
            else if (kind == RT_MEDIUM && s->media[q->index].boundary_kind == RT_BOUNDARY_SPHERE &&
                     s->media[q->index].xform == 0)
                other = &s->spheres[s->media[q->index].boundary_index];
            if (kind == RT_SPHERE && other->r > 0.0f && memcmp(other->c, sp->c, sizeof sp->c) == 0 &&
                memcmp(&other->r, &sp->r, sizeof sp->r) == 0 && b.sphere < 0) {
                b.sphere = (int)j; /* the shell: position fixed below */
                continue;
            }
            const double hx = ((double)bounds[j].hi.x - (double)bounds[j].lo.x) * 0.5;
            const double hy = ((double)bounds[j].hi.y - (double)bounds[j].lo.y) * 0.5;
            const double hz = ((double)bounds[j].hi.z - (double)bounds[j].lo.z) * 0.5;
            const double safety = 0.002 * ((double)sp->r + sqrt(hx * hx + hy * hy + hz * hz));
            double gap;
            if (other) {
                const double dx = (double)other->c[0] - (double)sp->c[0], dy = (double)other->c[1] - (double)sp->c[1],
                             dz = (double)other->c[2] - (double)sp->c[2];
                gap = sqrt(dx * dx + dy * dy + dz * dz) - fabs((double)other->r) - (double)sp->r;
            } else {
                const double cx = sp->c[0], cy = sp->c[1], cz = sp->c[2];
                const double px = cx < bounds[j].lo.x ? bounds[j].lo.x : (cx > bounds[j].hi.x ? bounds[j].hi.x : cx);
                const double py = cy < bounds[j].lo.y ? bounds[j].lo.y : (cy > bounds[j].hi.y ? bounds[j].hi.y : cy);
                const double pz = cz < bounds[j].lo.z ? bounds[j].lo.z : (cz > bounds[j].hi.z ? bounds[j].hi.z : cz);
                gap = sqrt((px - cx) * (px - cx) + (py - cy) * (py - cy) + (pz - cz) * (pz - cz)) - (double)sp->r;
            }
            if (!(gap > safety)) ok = 0;
        }
        if (!ok) continue;
        if (b.sphere >= 0) {
            b.n = 2;
            b.prim[1] = (uint32_t)b.sphere; /* order by leaf position: set after the build */
            b.sphere = 1;
        }
        s->balls[s->n_balls++] = b;
    }
}

/* the ball's primitives in the order the leaves hold them (the order the traversal would test
 * them in when they share a leaf) */
static void order_balls(orc_scene *s) {
    for (uint32_t k = 0; k < s->n_balls; k++) {
        orc_ball *b = &s->balls[k];
        if (b->n != 2) continue;
        uint32_t at[2] = {0, 0};
        for (uint32_t i = 0; i < s->n_tree_prims; i++) {
            if (s->order[i] == b->prim[0]) at[0] = i;
            if (s->order[i] == b->prim[1]) at[1] = i;
        }
        if (at[1] < at[0]) {
            const uint32_t t = b->prim[0];
            b->prim[0] = b->prim[1];
            b->prim[1] = t;
            b->sphere = 0;
        }
    }
}

int orc_build(orc_scene *s, const rt_build_opts *opts) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    s->n_nodes = 0;
    s->max_depth = 0;
    s->n_enclosing = 0;
    s->n_balls = 0;
    s->n_tree_prims = s->n_prims;
    if (s->n_prims == 0) return RT_OK;
    aabb *bounds = (aabb *)malloc((size_t)s->n_prims * sizeof *bounds);
    if (!bounds) { orc_set_error("out of memory building tree"); return RT_ERR_NOMEM; }
    for (uint32_t i = 0; i < s->n_prims; i++) bounds[i] = orc_prim_bounds(s, i);
    find_enclosing(s, bounds);
    find_balls(s, bounds);
    s->n_tree_prims = s->n_prims - s->n_enclosing;
    uint32_t n = s->n_tree_prims;
    builder B;
    memset(&B, 0, sizeof B);
    B.s = s;
    B.max_prims = opts && opts->max_prims_in_node ? opts->max_prims_in_node : 4;
    if (B.max_prims > 255) B.max_prims = 255; /* bvh.rs:114 */
    B.method = opts ? (int)opts->split_method : RT_SPLIT_SAH;
    B.info = (pinfo *)malloc((size_t)n * sizeof *B.info);
    B.ordered = (uint32_t *)malloc((size_t)n * sizeof *B.ordered);
    B.pool_cap = 2 * n + 1;
    B.pool = (bnode *)malloc((size_t)B.pool_cap * sizeof *B.pool);
    if (!B.info || !B.ordered || !B.pool) {
        free(B.info); free(B.ordered); free(B.pool); free(bounds);
        orc_set_error("out of memory building tree");
        return RT_ERR_NOMEM;
    }
    for (uint32_t i = 0, k = 0, e = 0; i < s->n_prims; i++) { /* BVHPrimitiveInfo::new, bvh.rs:694-703 */
        if (e < s->n_enclosing && s->enclosing[e] == i) { e++; continue; }
        B.info[k].prim = i;
        B.info[k].b = bounds[i];
        B.info[k].c = v3_add(v3_scale(B.info[k].b.lo, 0.5f), v3_scale(B.info[k].b.hi, 0.5f));
        k++;
    }
    free(bounds);
    bnode *root = (B.method == RT_SPLIT_HLBVH) ? hlbvh_build(&B, n) : recursive_build(&B, 0, n);
    int rc = RT_OK;
    if (!root || B.n_ordered != n) {
        orc_set_error("tree construction failed");
        rc = RT_ERR_INVALID;
    } else {
        s->nodes = (rt_linear_node *)malloc((size_t)B.total_nodes * sizeof *s->nodes);
        s->order = B.ordered;
        B.ordered = NULL;
        for (bnode *p = B.pool; p < B.pool + B.pool_used; p++)
            if (!p->child[0] && p->count > 65535) rc = RT_ERR_LIMIT;
        if (rc == RT_OK) {
            flatten(s, root);
            order_balls(s);
            s->max_depth = orc_tree_depth(s->nodes, s->n_nodes);
            if (s->max_depth > 64) { /* nodes_to_visit = [0; 64], bvh.rs:753 */
                orc_set_error("tree needs %u stack entries, the traversal stack holds 64",
                              s->max_depth);
                rc = RT_ERR_DEPTH;
            }
        } else {
            orc_set_error("leaf with more than 65535 primitives");
        }
    }
    free(B.info);
    free(B.ordered);
    free(B.pool);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    s->build_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    return rc;
}

/* ---- slab test, geometry/src/bounds3.rs:212-238 ------------------------------ */
int orc_slab(const rt_linear_node *n, const rt_ray *ray, const float inv[3], const int neg[3]) {
    const float g = 1.0f + 2.0f * orc_gamma(3);
    float t_min = ((neg[0] ? n->hi[0] : n->lo[0]) - ray->o[0]) * inv[0];
    float t_max = ((neg[0] ? n->lo[0] : n->hi[0]) - ray->o[0]) * inv[0];
    float ty_min = ((neg[1] ? n->hi[1] : n->lo[1]) - ray->o[1]) * inv[1];
    float ty_max = ((neg[1] ? n->lo[1] : n->hi[1]) - ray->o[1]) * inv[1];
    t_max *= g;
    ty_max *= g;
    if (t_min > ty_max || ty_min > t_max) return 0;
    t_min = sel_max(t_min, ty_min);
    t_max = sel_min(t_max, ty_max);
    float tz_min = ((neg[2] ? n->hi[2] : n->lo[2]) - ray->o[2]) * inv[2];
    float tz_max = ((neg[2] ? n->lo[2] : n->hi[2]) - ray->o[2]) * inv[2];
    tz_max *= g;
    if (t_min > tz_max || tz_min > t_max) return 0;
    t_min = sel_max(t_min, tz_min);
    t_max = sel_min(t_max, tz_max);
    return t_min < ray->t_max && t_max > 0.0f;
}

/* A ray that starts inside a ball (find_balls), after the enclosing media have been tested:
 * everything outside the ball lies beyond the point where the ray leaves it, so if the ball's
 * own primitives decide the hit the tree need not be walked.  They do when
 *   - with a shell: the shell is hit at all (t_max = infinity), so the ray does leave through it,
 *     and testing the ball's primitives in either order gives the same winner and distance
 *     (the traversal may meet them in either order; only an exact tie tells the orders apart);
 *   - without a shell: the medium scatters the ray (nothing else can come before that).
 * Then *ray, *found, *best_prim, best_aux are the traversal's result and 1 is returned; the
 * counters take one test per primitive of the ball and no node.  Otherwise nothing is changed,
 * 0 is returned and the traversal runs as usual. */
static int ball_resolve(const orc_scene *s, rt_ray *ray, int *found, uint32_t *best_prim, float best_aux[4],
                        uint64_t counters[9]) {
    for (uint32_t k = 0; k < s->n_balls; k++) {
        const orc_ball *b = &s->balls[k];
        const float dx = ray->o[0] - b->c[0], dy = ray->o[1] - b->c[1], dz = ray->o[2] - b->c[2];
        if (!(dx * dx + dy * dy + dz * dz < b->r_in2)) continue;
        if (b->sphere >= 0) {
            rt_ray far = *ray;
            far.t_max = INFINITY;
            float t, aux[4] = {0, 0, 0, 0};
            if (!orc_prim_test(s, b->prim[b->sphere], &far, &t, aux)) return 0;
        }
        rt_ray r[2];
        int f[2] = {0, 0};
        uint32_t bp[2] = {0, 0};
        float ba[2][4];
        for (uint32_t order = 0; order < b->n; order++) {
            r[order] = *ray;
            f[order] = *found;
            bp[order] = *best_prim;
            memcpy(ba[order], best_aux, sizeof ba[order]);
            for (uint32_t q = 0; q < b->n; q++) {
                const uint32_t prim = b->prim[order == 0 ? q : b->n - 1 - q];
                float t, aux[4] = {0, 0, 0, 0};
                if (orc_prim_test(s, prim, &r[order], &t, aux)) {
                    r[order].t_max = t;
                    f[order] = 1;
                    bp[order] = prim;
                    memcpy(ba[order], aux, sizeof aux);
                }
            }
        }
        if (b->n == 2 && (f[0] != f[1] || bp[0] != bp[1] || memcmp(&r[0].t_max, &r[1].t_max, sizeof(float)) != 0))
            return 0;
        if (b->sphere < 0 && !(f[0] && bp[0] == b->prim[0])) return 0;
        if (counters)
            for (uint32_t q = 0; q < b->n; q++) counters[1 + (s->prims[b->prim[q]].kind & 0xffu)]++;
        *ray = r[0];
        *found = f[0];
        *best_prim = bp[0];
        memcpy(best_aux, ba[0], sizeof ba[0]);
        return 1;
    }
    return 0;
}

/* closest hit, bvh.rs:736-799.  The reference keeps the last primitive hit and
 * never shortens the ray (its Primitive::intersect returns no distance); here
 * t_max shrinks on every accepted hit, ties go to the later primitive. */
int orc_closest(const orc_scene *s, const rt_ray *ray_in, orc_surf *out, uint64_t counters[9]) {
    rt_ray ray = *ray_in;
    float inv[3] = {1.0f / ray.d[0], 1.0f / ray.d[1], 1.0f / ray.d[2]};
    int neg[3] = {inv[0] < 0.0f, inv[1] < 0.0f, inv[2] < 0.0f};
    uint32_t stack[64];
    uint32_t sp = 0, cur = 0;
    int found = 0;
    uint32_t best_prim = 0;
    float best_aux[4] = {0, 0, 0, 0};
    for (uint32_t e = 0; e < s->n_enclosing; e++) { /* media around the whole scene: see find_enclosing */
        float t, aux[4] = {0, 0, 0, 0};
        if (counters) counters[1 + RT_MEDIUM]++;
        if (orc_prim_test(s, s->enclosing[e], &ray, &t, aux)) {
            ray.t_max = t;
            found = 1;
            best_prim = s->enclosing[e];
            memcpy(best_aux, aux, sizeof aux);
        }
    }
    if (ball_resolve(s, &ray, &found, &best_prim, best_aux, counters)) {
        if (found && out) orc_prim_finalize(s, best_prim, ray_in, ray.t_max, best_aux, out);
        return found;
    }
    for (; s->n_nodes != 0;) {
        const rt_linear_node *node = &s->nodes[cur];
        if (counters) counters[0]++;
        if (orc_slab(node, &ray, inv, neg)) {
            if (node->primitive_count > 0) {
                for (uint32_t i = 0; i < node->primitive_count; i++) {
                    uint32_t prim = s->order[node->child_offset + i];
                    float t, aux[4] = {0, 0, 0, 0};
                    if (counters) counters[1 + (s->prims[prim].kind & 0xffu)]++;
                    if (orc_prim_test(s, prim, &ray, &t, aux)) {
                        ray.t_max = t;
                        found = 1;
                        best_prim = prim;
                        memcpy(best_aux, aux, sizeof aux);
                    }
                }
                if (sp == 0) break;
                cur = stack[--sp];
            } else if (neg[node->axis]) {
                stack[sp++] = cur + 1;
                cur = node->child_offset;
            } else {
                stack[sp++] = node->child_offset;
                cur = cur + 1;
            }
        } else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    if (found && out) orc_prim_finalize(s, best_prim, ray_in, ray.t_max, best_aux, out);
    return found;
}

/* any hit, bvh.rs:801-863 */
int orc_any(const orc_scene *s, const rt_ray *ray) {
    for (uint32_t e = 0; e < s->n_enclosing; e++) {
        float t, aux[4];
        if (orc_prim_test(s, s->enclosing[e], ray, &t, aux)) return 1;
    }
    if (s->n_nodes == 0) return 0;
    float inv[3] = {1.0f / ray->d[0], 1.0f / ray->d[1], 1.0f / ray->d[2]};
    int neg[3] = {inv[0] < 0.0f, inv[1] < 0.0f, inv[2] < 0.0f};
    uint32_t stack[64];
    uint32_t sp = 0, cur = 0;
    for (;;) {
        const rt_linear_node *node = &s->nodes[cur];
        if (orc_slab(node, ray, inv, neg)) {
            if (node->primitive_count > 0) {
                for (uint32_t i = 0; i < node->primitive_count; i++) {
                    float t, aux[4];
                    if (orc_prim_test(s, s->order[node->child_offset + i], ray, &t, aux)) return 1;
                }
                if (sp == 0) break;
                cur = stack[--sp];
            } else if (neg[node->axis]) {
                stack[sp++] = cur + 1;
                cur = node->child_offset;
            } else {
                stack[sp++] = node->child_offset;
                cur = cur + 1;
            }
        } else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return 0;
}
