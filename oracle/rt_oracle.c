/*
 * rt_oracle.c -- CPU restatement of the reference path tracer's hot path.
 * TEST INFRASTRUCTURE ONLY (checker + CPU baseline); see rt_oracle.h for the
 * citations, the pinning status and the randomness note.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (Makefile).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F32_MAX 3.40282346638528859811704183484516925440e+38f

/* ------------------------------------------------------------------ */
/* data model: mirrors the reference's pointer-based layout            */
/* ------------------------------------------------------------------ */

/* models/triangle.go:12-31 */
typedef struct OTriangle {
    int32_t index;        /* per-group running index (context.go:118,158) */
    int32_t orig;         /* position in the input array (restatement only) */
    float v[3][3];
    float normal[3];
    float e0[3], e1[3], e2[3];
    int32_t material;
    int32_t is_light;
    float tc[3][2];       /* TextureCoords (context.go:138-156), zero when the mesh has none */
} OTriangle;

/* models/texture.go:13-19: an image.RGBA behind a material's map_Kd */
typedef struct OTexture {
    uint8_t *pix;         /* RGBA8 rows, as image.RGBA.Pix */
    int32_t w, h;
} OTexture;

/* models/bvh.go:17-29 + models/aabb.go:11-22 (only Bounds is used) */
typedef struct ONode {
    int32_t depth, index;
    struct ONode *left, *right;
    float bounds[2][3];
    int32_t start, end;
    float split[4];
} ONode;

struct orc_scene {
    int32_t n;
    OTriangle *store;     /* triangle records */
    OTriangle **tris;     /* the reference's []*Triangle, sorted by the build */
    int32_t n_mat;
    float *kd;
    uint8_t *mat_light;
    OTexture *tex;        /* per material; pix == NULL: no map_Kd / not found */
    ONode *root;
    int32_t n_nodes;
    /* build parameters */
    int32_t use_bvh, max_leaf, max_depth;
    /* sample tables (input data) */
    double *cam_tab;
    int32_t n_cam;
    double *light_tab;
    int32_t n_light;
    /* 0: the build's row index, a hash of the camera sample's stream + the
     * sample's running number along the path (what the GPU kernels use);
     * 1: the reference's own scheme, ONE index per light that advances with
     * every Sample() call for the lifetime of the worker and wraps at the
     * table's end (arealight.go:42-56).  Mode 1 is serial by nature; it exists
     * to compare image statistics of the two schemes. */
    int32_t light_index_mode;
    uint64_t *light_cursor;
};

#define COUNT(c, f) do { if (c) (c)->f++; } while (0)

/* ------------------------------------------------------------------ */
/* Go math package restatements (float64, pure-Go versions)            */
/* ------------------------------------------------------------------ */

/* math.Max / math.Min special cases: NaN propagates, Max(+0,-0) = +0,
 * Min(-0,+0) = -0 */
double orc_go_max(double x, double y)
{
    if (isinf(x) && x > 0) return x;
    if (isinf(y) && y > 0) return y;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}
double orc_go_min(double x, double y)
{
    if (isinf(x) && x < 0) return x;
    if (isinf(y) && y < 0) return y;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? x : y;
    return x < y ? x : y;
}

static const double SIN_C[6] = {
    1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
    -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1,
};
static const double COS_C[6] = {
    -1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
    2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2,
};
#define GO_PI 3.14159265358979323846264338327950288419716939937510582097494459
#define PI4A 7.85398125648498535156e-1
#define PI4B 3.77489470793079817668e-8
#define PI4C 2.69515142907905952645e-15

/* Cephes-derived kernel shared by Sin and Cos (math/sin.go).  Arguments on
 * this path are bounded (|x| <= a few), far below the Payne-Hanek threshold
 * of 2^29, which is therefore not restated. */
static double go_sincos(double x, int want_cos)
{
    if (isnan(x) || isinf(x)) return NAN;
    int sign = 0;
    if (x < 0) {
        x = -x;
        if (!want_cos) sign = 1;
    }
    uint64_t j = (uint64_t)(x * (4.0 / GO_PI));
    double y = (double)j;
    if (j & 1) { j++; y++; }
    j &= 7;
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) { j -= 4; sign = !sign; }
    if (want_cos && j > 1) sign = !sign;
    double zz = z * z;
    int use_cos_poly = want_cos ? !(j == 1 || j == 2) : (j == 1 || j == 2);
    if (use_cos_poly)
        y = 1.0 - 0.5 * zz + zz * zz * ((((((COS_C[0] * zz) + COS_C[1]) * zz + COS_C[2]) * zz + COS_C[3]) * zz + COS_C[4]) * zz + COS_C[5]);
    else
        y = z + z * zz * ((((((SIN_C[0] * zz) + SIN_C[1]) * zz + SIN_C[2]) * zz + SIN_C[3]) * zz + SIN_C[4]) * zz + SIN_C[5]);
    return sign ? -y : y;
}
double orc_go_cos(double x) { return go_sincos(x, 1); }
static double go_sin(double x) { return go_sincos(x, 0); }

#define LN2HI 6.93147180369123816490e-01
#define LN2LO 1.90821492927058770002e-10
#define LOG2E 1.44269504088896338700e+00

/* math/exp.go (FreeBSD e_exp.c derived) */
double orc_go_exp(double x)
{
    const double Overflow = 7.09782712893383973096e+02;
    const double Underflow = -7.45133219101941108420e+02;
    const double NearZero = 1.0 / (double)(1 << 28);
    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (isinf(x)) return 0;
    if (x > Overflow) return INFINITY;
    if (x < Underflow) return 0;
    if (-NearZero < x && x < NearZero) return 1 + x;
    int k = 0;
    if (x < 0) k = (int)(LOG2E * x - 0.5);
    else if (x > 0) k = (int)(LOG2E * x + 0.5);
    double hi = x - (double)k * LN2HI;
    double lo = (double)k * LN2LO;
    const double P1 = 1.66666666666666657415e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return ldexp(y, k);
}

/* math/log.go (FreeBSD e_log.c derived) */
double orc_go_log(double x)
{
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                 L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                 L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < 1.41421356237309504880168872420969807856967187537694807317667974 / 2) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * LN2HI - ((hfsq - (s * (hfsq + R) + k * LN2LO)) - f);
}

static int is_odd_int(double x)
{
    double ip;
    double fp = modf(x, &ip);
    return fp == 0 && ((int64_t)ip & 1) == 1;
}

/* math/pow.go */
double orc_go_pow(double x, double y)
{
    if (y == 0 || x == 1) return 1;
    if (y == 1) return x;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0) {
        if (y < 0) {
            if (is_odd_int(y)) return copysign(INFINITY, x);
            return INFINITY;
        }
        if (is_odd_int(y)) return x;
        return 0;
    }
    if (isinf(y)) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (isinf(x)) {
        if (x < 0) {
            if (y < 0) return is_odd_int(y) ? copysign(0.0, -1.0) : 0;
            return is_odd_int(y) ? -INFINITY : INFINITY;
        }
        return y < 0 ? 0 : INFINITY;
    }
    if (y == 0.5) return sqrt(x);
    if (y == -0.5) return 1 / sqrt(x);

    double yi;
    double yf = modf(fabs(y), &yi);
    if (yf != 0 && x < 0) return NAN;
    if (yi >= 9.223372036854775808e18) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    double a1 = 1.0;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5) {
            yf--;
            yi++;
        }
        a1 = orc_go_exp(yf * orc_go_log(x));
    }
    int xe;
    double x1 = frexp(x, &xe);
    for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 12) || (1 << 12) < xe) {
            ae += xe;
            break;
        }
        if (i & 1) {
            a1 *= x1;
            ae += xe;
        }
        x1 *= x1;
        xe <<= 1;
        if (x1 < .5) {
            x1 += x1;
            xe--;
        }
    }
    if (y < 0) {
        a1 = 1 / a1;
        ae = -ae;
    }
    return ldexp(a1, ae);
}

/* ------------------------------------------------------------------ */
/* mgl32 helpers (float32)                                             */
/* ------------------------------------------------------------------ */

static inline float dot3(const float a[3], const float b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline void cross3(float out[3], const float a[3], const float b[3])
{
    float r0 = a[1] * b[2] - a[2] * b[1];
    float r1 = a[2] * b[0] - a[0] * b[2];
    float r2 = a[0] * b[1] - a[1] * b[0];
    out[0] = r0; out[1] = r1; out[2] = r2;
}
static inline float lensqr3(const float a[3]) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
/* Vec3.Len: float32(math.Sqrt(float64(sum))) */
static inline float len3(const float a[3]) { return (float)sqrt((double)lensqr3(a)); }
/* Vec3.Normalize: multiply by the reciprocal of the length */
static inline void normalize3(float out[3], const float a[3])
{
    float l = 1.0f / len3(a);
    out[0] = a[0] * l; out[1] = a[1] * l; out[2] = a[2] * l;
}
/* mgl32.TransformCoordinate: (m * (v,1)).xyz * (1/w), column-major m */
static void transform_coordinate(float out[3], const float v[3], const float m[16])
{
    float t[4];
    for (int r = 0; r < 4; ++r)
        t[r] = m[r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * 1.0f;
    float iw = 1 / t[3];
    out[0] = t[0] * iw; out[1] = t[1] * iw; out[2] = t[2] * iw;
}

/* utility.go:34-40 */
static inline float clamp01(float c) { return (float)orc_go_min(orc_go_max((double)c, 0), 1.0); }

/* ------------------------------------------------------------------ */
/* ray / box / triangle / walk, instantiated for float and double      */
/* ------------------------------------------------------------------ */

#define REAL float
#define FN(x) x##_f32
#include "rt_walk.inc"
#undef REAL
#undef FN
#define REAL double
#define FN(x) x##_f64
#include "rt_walk.inc"
#undef REAL
#undef FN

/* ------------------------------------------------------------------ */
/* primitives exposed for the known-answer tests                       */
/* ------------------------------------------------------------------ */

void orc_aabb_intersect(const float bmin[3], const float bmax[3], const float o[3],
                        const float d[3], int mode, int *hit, double *tmin, double *tmax)
{
    float bounds[2][3];
    memcpy(bounds[0], bmin, sizeof bounds[0]);
    memcpy(bounds[1], bmax, sizeof bounds[1]);
    *tmin = 0; *tmax = 0;
    if (mode == ORC_MODE_F64) {
        ray_t_f64 r;
        double ro[3] = {o[0], o[1], o[2]}, rd[3] = {d[0], d[1], d[2]};
        new_ray_f64(&r, ro, rd);
        *hit = aabb_hit_f64(bounds, &r, tmin, tmax);
    } else {
        ray_t_f32 r;
        float t0 = 0, t1 = 0;
        new_ray_f32(&r, o, d);
        *hit = aabb_hit_f32(bounds, &r, &t0, &t1);
        *tmin = t0; *tmax = t1;
    }
}

/* models/triangle.go:45-65 */
static void make_triangle(OTriangle *t, const float v[9], int32_t material, int32_t is_light,
                          int32_t index, int32_t orig)
{
    float a[3], b[3], c[3];
    memcpy(t->v, v, sizeof t->v);
    for (int k = 0; k < 3; ++k) {
        a[k] = t->v[1][k] - t->v[0][k];
        b[k] = t->v[2][k] - t->v[0][k];
    }
    cross3(c, a, b);
    normalize3(t->normal, c);
    for (int k = 0; k < 3; ++k) {
        t->e0[k] = t->v[1][k] - t->v[0][k];
        t->e1[k] = t->v[2][k] - t->v[1][k];
        t->e2[k] = t->v[0][k] - t->v[2][k];
    }
    t->material = material;
    t->is_light = is_light;
    t->index = index;
    t->orig = orig;
}

void orc_tri_normal(const float v0[3], const float v1[3], const float v2[3], float n[3])
{
    OTriangle t;
    float v[9];
    memcpy(v, v0, 12); memcpy(v + 3, v1, 12); memcpy(v + 6, v2, 12);
    make_triangle(&t, v, 0, 0, 0, 0);
    memcpy(n, t.normal, 12);
}

void orc_tri_intersect(const float v0[3], const float v1[3], const float v2[3],
                       const float o[3], const float d[3], int mode,
                       double *t, double *u, double *v)
{
    OTriangle tri;
    float vv[9];
    memcpy(vv, v0, 12); memcpy(vv + 3, v1, 12); memcpy(vv + 6, v2, 12);
    make_triangle(&tri, vv, 0, 0, 0, 0);
    if (mode == ORC_MODE_F64) {
        ray_t_f64 r;
        double ro[3] = {o[0], o[1], o[2]}, rd[3] = {d[0], d[1], d[2]};
        new_ray_f64(&r, ro, rd);
        *t = tri_hit_f64(&tri, &r, u, v);
    } else {
        ray_t_f32 r;
        float fu, fv;
        new_ray_f32(&r, o, d);
        *t = tri_hit_f32(&tri, &r, &fu, &fv);
        *u = fu; *v = fv;
    }
}

/* ------------------------------------------------------------------ */
/* counter RNG (build-defined; the GPU path implements the same spec)  */
/* ------------------------------------------------------------------ */

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#define GOLDEN64 0x9E3779B97F4A7C15ULL

uint64_t orc_rng_stream(int64_t seed, uint64_t path_index)
{
    uint64_t a = mix64((uint64_t)seed + GOLDEN64);
    return mix64(a ^ (path_index * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
}
static inline uint64_t rng_u64(uint64_t stream, uint32_t counter)
{
    return mix64(stream + ((uint64_t)counter + 1) * GOLDEN64);
}
/* 24 random bits -> [0,1), the granularity of Go's rand.Float32 */
float orc_rng_f32(uint64_t stream, uint32_t counter)
{
    return (float)(rng_u64(stream, counter) >> 40) * 0x1p-24f;
}

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */

static void free_tree(ONode *n)
{
    if (!n) return;
    free_tree(n->left);
    free_tree(n->right);
    free(n);
}

orc_scene *orc_scene_create(const float *verts, const int32_t *material,
                            const int32_t *group_index, int32_t n,
                            const float *kd, const uint8_t *is_light, int32_t m)
{
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->n = n;
    s->store = (OTriangle *)calloc((size_t)(n > 0 ? n : 1), sizeof(OTriangle));
    s->tris = (OTriangle **)calloc((size_t)(n > 0 ? n : 1), sizeof(OTriangle *));
    s->n_mat = m;
    s->kd = (float *)malloc(sizeof(float) * 3 * (size_t)(m > 0 ? m : 1));
    s->mat_light = (uint8_t *)malloc((size_t)(m > 0 ? m : 1));
    memcpy(s->kd, kd, sizeof(float) * 3 * (size_t)m);
    memcpy(s->mat_light, is_light, (size_t)m);
    s->tex = (OTexture *)calloc((size_t)(m > 0 ? m : 1), sizeof(OTexture));
    for (int32_t i = 0; i < n; ++i) {
        int32_t mat = material ? material[i] : 0;
        make_triangle(&s->store[i], verts + 9 * (size_t)i, mat, is_light[mat],
                      group_index ? group_index[i] : i, i);
        s->tris[i] = &s->store[i];
    }
    return s;
}

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    free_tree(s->root);
    if (s->tex)
        for (int32_t i = 0; i < s->n_mat; ++i) free(s->tex[i].pix);
    free(s->tex);
    free(s->store); free(s->tris); free(s->kd); free(s->mat_light);
    free(s->cam_tab); free(s->light_tab);
    free(s->light_cursor);
    free(s);
}

/* context.go:138-156: per-vertex texture coordinates of every triangle, in
 * input order (n*6 floats: uv0, uv1, uv2). */
void orc_scene_set_texcoords(orc_scene *s, const float *uv)
{
    for (int32_t i = 0; i < s->n; ++i)
        memcpy(s->store[i].tc, uv + 6 * (size_t)i, sizeof(float) * 6);
}

/* context.go:85-103 TextureLookup[MapKd] for one material: RGBA8 pixels as
 * image.RGBA holds them (texture.go:24-46). */
int32_t orc_scene_set_texture(orc_scene *s, int32_t material, const uint8_t *rgba,
                              int32_t w, int32_t h)
{
    if (material < 0 || material >= s->n_mat || w <= 0 || h <= 0 || !rgba) return -1;
    OTexture *t = &s->tex[material];
    free(t->pix);
    t->pix = (uint8_t *)malloc((size_t)w * (size_t)h * 4);
    memcpy(t->pix, rgba, (size_t)w * (size_t)h * 4);
    t->w = w;
    t->h = h;
    return 0;
}

void orc_set_light_index_mode(orc_scene *s, int32_t mode)
{
    if (!s->light_cursor) s->light_cursor = (uint64_t *)calloc(1, sizeof(uint64_t));
    *s->light_cursor = 0;
    s->light_index_mode = mode;
}

void orc_set_sample_tables(orc_scene *s, const double *camera_tab, int32_t n_camera,
                           const double *light_tab, int32_t n_light)
{
    free(s->cam_tab); free(s->light_tab);
    s->cam_tab = (double *)malloc(sizeof(double) * 2 * (size_t)n_camera);
    s->light_tab = (double *)malloc(sizeof(double) * 2 * (size_t)n_light);
    memcpy(s->cam_tab, camera_tab, sizeof(double) * 2 * (size_t)n_camera);
    memcpy(s->light_tab, light_tab, sizeof(double) * 2 * (size_t)n_light);
    s->n_cam = n_camera;
    s->n_light = n_light;
}

/* ------------------------------------------------------------------ */
/* BVH build: literal restatement of bvh.go:31-114,146-212             */
/* ------------------------------------------------------------------ */

static void tri_center(const OTriangle *t, float c[3])
{
    /* Vertices[0].Add(Vertices[1]).Add(Vertices[2]).Mul(1.0 / 3.0) */
    const float third = (float)(1.0 / 3.0);
    for (int k = 0; k < 3; ++k)
        c[k] = ((t->v[0][k] + t->v[1][k]) + t->v[2][k]) * third;
}
static void tri_min(const OTriangle *t, float m[3])
{
    for (int k = 0; k < 3; ++k)
        m[k] = (float)orc_go_min(orc_go_min(t->v[0][k], t->v[1][k]), t->v[2][k]);
}
static void tri_max(const OTriangle *t, float m[3])
{
    for (int k = 0; k < 3; ++k)
        m[k] = (float)orc_go_max(orc_go_max(t->v[0][k], t->v[1][k]), t->v[2][k]);
}

typedef struct { float key; int32_t index; int32_t orig; OTriangle *t; } sort_item;

static int sort_cmp(const void *pa, const void *pb)
{
    const sort_item *a = (const sort_item *)pa, *b = (const sort_item *)pb;
    if (a->key == b->key) {
        if (a->index != b->index) return a->index < b->index ? -1 : 1;
        /* the reference stops here (triangle.go:71-73) and leaves equal
         * (key, Index) pairs to an unstable sort; the restatement breaks
         * the remaining ties by input position so the order is total */
        return a->orig < b->orig ? -1 : (a->orig > b->orig ? 1 : 0);
    }
    return a->key < b->key ? -1 : 1;
}

/* triangle.go:67-76 -- sort [start..end] by axis.Dot(Center()), ties by Index */
static void triangle_sorter(const float axis[3], OTriangle **tris, int32_t start, int32_t end,
                            sort_item *scratch)
{
    int32_t cnt = end - start + 1;
    if (cnt <= 1) return;
    for (int32_t i = 0; i < cnt; ++i) {
        float c[3];
        tri_center(tris[start + i], c);
        scratch[i].key = dot3(axis, c);
        scratch[i].index = tris[start + i]->index;
        scratch[i].orig = tris[start + i]->orig;
        scratch[i].t = tris[start + i];
    }
    qsort(scratch, (size_t)cnt, sizeof(sort_item), sort_cmp);
    for (int32_t i = 0; i < cnt; ++i)
        tris[start + i] = scratch[i].t;
}

static float box_area(const float mn[3], const float mx[3])
{
    float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    return 2 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
}

typedef struct {
    orc_scene *s;
    sort_item *scratch;
    float *left_sizes, *right_sizes;
} build_ctx;

/* bvh.go:146-212 */
static void split_plane_sah(build_ctx *b, OTriangle **tris, const ONode *node, float plane[4])
{
    float lowest = F32_MAX;
    plane[0] = plane[1] = plane[2] = plane[3] = 0;
    static const float axes[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    int32_t start = node->start, end = node->end;
    for (int a = 0; a < 3; ++a) {
        triangle_sorter(axes[a], tris, start, end, b->scratch);
        float lmin[3], lmax[3], rmin[3], rmax[3], m[3];
        tri_min(tris[start], lmin); tri_max(tris[start], lmax);
        tri_min(tris[end], rmin); tri_max(tris[end], rmax);
        for (int32_t i = start; i <= end; ++i) {
            tri_min(tris[i], m);
            for (int k = 0; k < 3; ++k) lmin[k] = (float)orc_go_min(lmin[k], m[k]);
            tri_max(tris[i], m);
            for (int k = 0; k < 3; ++k) lmax[k] = (float)orc_go_max(lmax[k], m[k]);
            b->left_sizes[i - start] = box_area(lmin, lmax);
        }
        for (int32_t i = end; i >= start; --i) {
            tri_min(tris[i], m);
            for (int k = 0; k < 3; ++k) rmin[k] = (float)orc_go_min(rmin[k], m[k]);
            tri_max(tris[i], m);
            for (int k = 0; k < 3; ++k) rmax[k] = (float)orc_go_max(rmax[k], m[k]);
            b->right_sizes[i - start] = box_area(rmin, rmax);
        }
        for (int32_t i = start + 1; i < end; ++i) {
            float la = b->left_sizes[i - start];
            float ra = b->right_sizes[i - start + 1];
            float cost = la * (float)(i - start) + ra * (float)(end - i + 1);
            if (cost < lowest) {
                lowest = cost;
                float c0[3], c1[3];
                tri_center(tris[i], c0);
                tri_center(tris[i + 1], c1);
                float w = (c0[a] + c1[a]) / 2.0f;
                plane[0] = axes[a][0]; plane[1] = axes[a][1]; plane[2] = axes[a][2];
                plane[3] = w;
            }
        }
    }
}

/* triangle.go:140-168 */
static void triangle_bounds(OTriangle **tris, int32_t start, int32_t end, float mn[3], float mx[3])
{
    for (int k = 0; k < 3; ++k) { mn[k] = F32_MAX; mx[k] = -F32_MAX; }
    for (int32_t i = start; i <= end; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                float c = tris[i]->v[j][k];
                if (c < mn[k]) mn[k] = c;
                if (c > mx[k]) mx[k] = c;
            }
}

/* bvh.go:56-114 */
static ONode *build_node(build_ctx *b, OTriangle **tris, int32_t start, int32_t end,
                         int32_t depth, int32_t *index)
{
    orc_scene *s = b->s;
    ONode *node = (ONode *)calloc(1, sizeof *node);
    node->depth = depth;
    node->index = (*index)++;
    node->start = start;
    node->end = end;
    triangle_bounds(tris, start, end, node->bounds[0], node->bounds[1]);
    int32_t count = end - start + 1;
    if (s->use_bvh && count > s->max_leaf && depth < s->max_depth) {
        float plane[4];
        split_plane_sah(b, tris, node, plane);
        memcpy(node->split, plane, sizeof plane);
        triangle_sorter(plane, tris, start, end, b->scratch);
        int32_t split = start;
        float c[3];
        tri_center(tris[start], c);
        int side0 = dot3(plane, c) > plane[3];
        for (int32_t i = start + 1; i < end; ++i) {
            tri_center(tris[i], c);
            if ((dot3(plane, c) > plane[3]) != side0) {
                split = i;
                break;
            }
        }
        if (split == start || split == end)
            split = start + (end - start) / 2;
        node->left = build_node(b, tris, start, split - 1, depth + 1, index);
        node->right = build_node(b, tris, split, end, depth + 1, index);
    }
    return node;
}

int32_t orc_build_bvh(orc_scene *s, int32_t use_bvh, int32_t max_leaf, int32_t max_depth)
{
    free_tree(s->root);
    s->use_bvh = use_bvh; s->max_leaf = max_leaf; s->max_depth = max_depth;
    build_ctx b;
    b.s = s;
    size_t n = (size_t)(s->n > 0 ? s->n : 1);
    b.scratch = (sort_item *)malloc(sizeof(sort_item) * n);
    b.left_sizes = (float *)malloc(sizeof(float) * n);
    b.right_sizes = (float *)malloc(sizeof(float) * n);
    int32_t index = 0;
    s->root = build_node(&b, s->tris, 0, s->n - 1, 0, &index);
    s->n_nodes = index;
    free(b.scratch); free(b.left_sizes); free(b.right_sizes);
    return index;
}

static ONode *install_node(int32_t *cursor, int32_t n_nodes, const int32_t *is_leaf,
                           const int32_t *start, const int32_t *end, const float *bmin,
                           const float *bmax, int32_t depth)
{
    if (*cursor >= n_nodes) return NULL;
    int32_t i = (*cursor)++;
    ONode *node = (ONode *)calloc(1, sizeof *node);
    node->index = i; node->depth = depth;
    node->start = start[i]; node->end = end[i];
    memcpy(node->bounds[0], bmin + 3 * (size_t)i, 12);
    memcpy(node->bounds[1], bmax + 3 * (size_t)i, 12);
    if (!is_leaf[i]) {
        node->left = install_node(cursor, n_nodes, is_leaf, start, end, bmin, bmax, depth + 1);
        node->right = install_node(cursor, n_nodes, is_leaf, start, end, bmin, bmax, depth + 1);
    }
    return node;
}

int32_t orc_set_bvh(orc_scene *s, int32_t n_nodes, const int32_t *is_leaf,
                    const int32_t *start, const int32_t *end, const float *bmin,
                    const float *bmax, const int32_t *order)
{
    free_tree(s->root);
    for (int32_t i = 0; i < s->n; ++i)
        s->tris[i] = &s->store[order[i]];
    int32_t cursor = 0;
    s->root = install_node(&cursor, n_nodes, is_leaf, start, end, bmin, bmax, 0);
    s->n_nodes = cursor;
    return cursor;
}

int32_t orc_node_count(const orc_scene *s) { return s->n_nodes; }

static void export_node(const ONode *n, int32_t *is_leaf, int32_t *start, int32_t *end,
                        int32_t *depth, float *bmin, float *bmax, float *split)
{
    if (!n) return;
    int32_t i = n->index;
    if (is_leaf) is_leaf[i] = n->left == NULL;
    if (start) start[i] = n->start;
    if (end) end[i] = n->end;
    if (depth) depth[i] = n->depth;
    if (bmin) memcpy(bmin + 3 * (size_t)i, n->bounds[0], 12);
    if (bmax) memcpy(bmax + 3 * (size_t)i, n->bounds[1], 12);
    if (split) memcpy(split + 4 * (size_t)i, n->split, 16);
    export_node(n->left, is_leaf, start, end, depth, bmin, bmax, split);
    export_node(n->right, is_leaf, start, end, depth, bmin, bmax, split);
}

void orc_export_bvh(const orc_scene *s, int32_t *is_leaf, int32_t *start, int32_t *end,
                    int32_t *depth, float *bmin, float *bmax, float *split, int32_t *order)
{
    export_node(s->root, is_leaf, start, end, depth, bmin, bmax, split);
    if (order)
        for (int32_t i = 0; i < s->n; ++i)
            order[i] = s->tris[i]->orig;
}

/* ------------------------------------------------------------------ */
/* area light discovery: context.go:172-226                            */
/* ------------------------------------------------------------------ */

/* triangle.go:105-138 */
static const float *shortest_edge(const OTriangle *t)
{
    if (len3(t->e0) < len3(t->e1)) {
        if (len3(t->e0) < len3(t->e2)) return t->e0;
        return t->e2;
    }
    if (len3(t->e1) < len3(t->e2)) return t->e1;
    return t->e2;
}
static const float *middle_edge(const OTriangle *t)
{
    if (len3(t->e0) < len3(t->e1)) {
        if (len3(t->e1) < len3(t->e2)) return t->e1;
        if (len3(t->e0) < len3(t->e2)) return t->e2;
        return t->e0;
    }
    if (len3(t->e2) < len3(t->e1)) return t->e1;
    if (len3(t->e0) < len3(t->e2)) return t->e0;
    return t->e2;
}

void orc_scene_light(const orc_scene *s, orc_light *out)
{
    float mn[3] = {F32_MAX, F32_MAX, F32_MAX}, mx[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
    float normal[3] = {0, 0, 0}, up[3] = {0, 0, 0}, shortest[3] = {0, 0, 0}, middle[3] = {0, 0, 0};
    int found = 0;
    memset(out, 0, sizeof *out);
    /* runs inside RenderContext.Initialize (context.go:180), before buildBVH /
     * loadBVH sort the array (main.go:65-90): context.Triangles is still in
     * OBJ order here, so the walk is over the input order whatever tree was
     * installed since; only the LAST light triangle's normal and edges survive */
    for (int32_t i = 0; i < s->n; ++i) {
        const OTriangle *t = &s->store[i];
        if (!t->is_light) continue;
        found = 1;
        memcpy(normal, t->normal, 12);
        memcpy(shortest, shortest_edge(t), 12);
        memcpy(middle, middle_edge(t), 12);
        float c[3];
        cross3(c, shortest, t->normal);
        normalize3(up, c);
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                float v = t->v[j][k];
                if (v < mn[k]) mn[k] = v;
                if (v > mx[k]) mx[k] = v;
            }
    }
    out->found = found;
    if (!found) return;
    float center[3], right[3];
    for (int k = 0; k < 3; ++k) center[k] = (mn[k] + mx[k]) / 2.0f;
    cross3(right, normal, up);
    /* Translate3D(center) * Mat3FromCols(normal x up, up, normal).Mat4() */
    float *m = out->transform;
    memset(m, 0, sizeof(float) * 16);
    for (int k = 0; k < 3; ++k) {
        m[k] = right[k];
        m[4 + k] = up[k];
        m[8 + k] = normal[k];
        m[12 + k] = center[k];
    }
    m[15] = 1;
    out->size[0] = len3(shortest) / 2.0f;
    out->size[1] = len3(middle) / 2.0f;
    out->emission[0] = out->emission[1] = out->emission[2] = 100;
    memcpy(out->normal, normal, 12);
}

/* ------------------------------------------------------------------ */
/* batch rayCast                                                       */
/* ------------------------------------------------------------------ */

void orc_trace_rays(const orc_scene *s, const float *rays, const float *tmax, int64_t n,
                    int mode, int32_t *tri_pos, float *t, float *u, float *v,
                    orc_counters *counters)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *o = rays + 6 * i, *d = rays + 6 * i + 3;
        float tm = tmax ? tmax[i] : F32_MAX;
        if (mode == ORC_MODE_F64) {
            hit_t_f64 h;
            ray_cast_f64(s, o, d, tm, &h, counters);
            tri_pos[i] = h.pos;
            if (t) t[i] = (float)h.tmin;
            if (u) u[i] = (float)h.umin;
            if (v) v[i] = (float)h.vmin;
        } else {
            hit_t_f32 h;
            ray_cast_f32(s, o, d, tm, &h, counters);
            tri_pos[i] = h.pos;
            if (t) t[i] = h.tmin;
            if (u) u[i] = h.umin;
            if (v) v[i] = h.vmin;
        }
    }
}

/* ------------------------------------------------------------------ */
/* camera: camera.go:43-129                                            */
/* ------------------------------------------------------------------ */

/* mgl32.QuatRotate(angle, axis).Rotate(v) */
static void quat_rotate(float out[3], float angle, const float axis[3], const float v[3])
{
    float c = (float)orc_go_cos((double)(angle / 2));
    float s = (float)go_sin((double)(angle / 2));
    float qv[3] = {axis[0] * s, axis[1] * s, axis[2] * s};
    float cr[3], q2[3], cr2[3];
    cross3(cr, qv, v);
    float w2 = 2 * c;
    for (int k = 0; k < 3; ++k) q2[k] = qv[k] * 2;
    cross3(cr2, q2, cr);
    for (int k = 0; k < 3; ++k) out[k] = (v[k] + cr[k] * w2) + cr2[k];
}

void orc_camera_setup(const orc_camera *cam, int32_t total_w, int32_t total_h,
                      float top_left[3], float steps[2])
{
    float tl[3], br[3];
    float ppd = cam->projection_plane_distance;
    if (cam->projection == 0) {
        const float pi32 = (float)GO_PI;
        float vha = pi32 * (cam->field_of_view / 2.0f) / 180.0f;
        float hha = vha * ((float)total_w / (float)total_h);
        float fwd[3] = {0, 0, ppd};
        static const float ax_y[3] = {0, 1, 0}, ax_x[3] = {1, 0, 0};
        float left[3], right[3], top[3], bottom[3];
        quat_rotate(left, -hha, ax_y, fwd);
        quat_rotate(right, hha, ax_y, fwd);
        quat_rotate(top, -vha, ax_x, fwd);
        quat_rotate(bottom, vha, ax_x, fwd);
        float sl = ppd / left[2], sr = ppd / right[2], st = ppd / top[2], sb = ppd / bottom[2];
        tl[0] = left[0] * sl; tl[1] = top[1] * st; tl[2] = ppd;
        br[0] = right[0] * sr; br[1] = bottom[1] * sb; br[2] = ppd;
    } else {
        float hw = cam->ortographic_size * ((float)total_w / (float)total_h);
        tl[0] = -hw; tl[1] = cam->ortographic_size; tl[2] = ppd;
        br[0] = hw; br[1] = -cam->ortographic_size; br[2] = ppd;
    }
    steps[1] = (tl[1] - br[1]) / (float)total_h;   /* verticalStep */
    steps[0] = (br[0] - tl[0]) / (float)total_w;   /* horizontalStep */
    memcpy(top_left, tl, 12);
}

static void camera_ray(const orc_scene *s, const orc_camera *cam, const float tl[3],
                       const float steps[2], int32_t xoff, int32_t yoff, int32_t x, int32_t y,
                       uint64_t path_index, float o[3], float d[3])
{
    float jx = 0.5f, jy = 0.5f;
    if (s->n_cam > 0) {
        uint64_t k = path_index % (uint64_t)s->n_cam;
        jx = (float)s->cam_tab[2 * k];
        jy = (float)s->cam_tab[2 * k + 1];
    }
    float lx = tl[0] + steps[0] * ((float)(xoff + x) + jx);
    float ly = tl[1] - steps[1] * ((float)(yoff + y) + jy);
    float oc[3] = {lx, ly, -cam->projection_plane_distance};
    transform_coordinate(o, oc, cam->transform);
    float diff[3];
    if (cam->projection == 0) {
        for (int k = 0; k < 3; ++k) diff[k] = o[k] - cam->transform[12 + k];
    } else {
        float f[3] = {0, 0, -1}, tf[3];
        transform_coordinate(tf, f, cam->transform);
        for (int k = 0; k < 3; ++k) diff[k] = tf[k] - cam->transform[12 + k];
    }
    normalize3(d, diff);
}

/* context.go:245-275: the effective camera transform and light of a pass */
static void pass_prepare(const orc_scene *s, const orc_pass *pass, orc_camera *cam,
                         orc_light *light)
{
    *cam = pass->camera;
    float tr = cam->transform[0] + cam->transform[5] + cam->transform[10] + cam->transform[15];
    if (fabs((double)tr) < 0.001) {
        memset(cam->transform, 0, sizeof cam->transform);
        cam->transform[0] = cam->transform[5] = cam->transform[10] = cam->transform[15] = 1;
    }
    orc_scene_light(s, light);
    const orc_settings *st = &pass->settings;
    if (!light->found || st->force_debug_light) {
        const float *m = st->debug_light_at_camera ? cam->transform : st->debug_light_transform;
        memcpy(light->transform, m, sizeof light->transform);
        float f[3] = {0, 0, -1}, tf[3];
        transform_coordinate(tf, f, m);
        for (int k = 0; k < 3; ++k) light->normal[k] = tf[k] - m[12 + k];
        light->size[0] = light->size[1] = st->debug_light_size;
        light->found = 1;
    }
    for (int k = 0; k < 3; ++k) light->emission[k] = 1.0f * st->light_intensity;
}

void orc_camera_ray(const orc_scene *s, const orc_pass *pass, int32_t x, int32_t y,
                    uint64_t path_index, float ray[6])
{
    orc_camera cam;
    orc_light light;
    float tl[3], steps[2];
    pass_prepare(s, pass, &cam, &light);
    orc_camera_setup(&cam, pass->total_width < 0 ? 0 : pass->total_width,
                     pass->total_height < 0 ? 0 : pass->total_height, tl, steps);
    camera_ray(s, &cam, tl, steps, pass->x_offset, pass->y_offset, x, y, path_index, ray, ray + 3);
}

/* ------------------------------------------------------------------ */
/* Trace: process/trace.go:20-119                                      */
/* ------------------------------------------------------------------ */

#define MAX_VERTS 257
#define HEMI_MAX_TRIES 64

typedef struct {
    const OTriangle *tri;
    float t, u, v;
    float point[3];
} cast_result;

/* trace.go:121-152; returns 0 for the reference's nil result */
static int ray_cast(const orc_scene *s, const float o[3], const float d[3], float tmax,
                    cast_result *res, orc_counters *c)
{
    hit_t_f32 h;
    ray_cast_f32(s, o, d, tmax, &h, c);
    if (!(h.tmin < F32_MAX)) return 0;
    res->tri = h.pos >= 0 ? s->tris[h.pos] : NULL;
    res->t = h.tmin; res->u = h.umin; res->v = h.vmin;
    for (int k = 0; k < 3; ++k) res->point[k] = o[k] + d[k] * h.tmin;
    return 1;
}

/* arealight.go:42-56 with the table index supplied by the caller */
static void light_sample(const orc_scene *s, const orc_light *light, uint64_t idx,
                         float out[3], float *pdf)
{
    double a = 0.5, b = 0.5;
    if (s->n_light > 0) {
        uint64_t k = idx % (uint64_t)s->n_light;
        a = s->light_tab[2 * k];
        b = s->light_tab[2 * k + 1];
    }
    float sample[3] = {(float)(a * 2 - 1) * light->size[0], (float)(b * 2 - 1) * light->size[1], 0};
    transform_coordinate(out, sample, light->transform);
    *pdf = 1.0f / (4.0f * light->size[0] * light->size[1]);
}

/* utility.go:50-70 on the counter RNG: rejection-sample the unit ball
 * (three draws per try), flip into the hemisphere.  The try count is bounded
 * so a GPU lane always terminates; the fallback (probability < 1e-20) is the
 * last candidate scaled into the ball. */
static void random_in_hemisphere(uint64_t stream, uint32_t *ctr, const float normal[3], float out[3])
{
    float p[3] = {0, 0, 0};
    for (int tries = 0; tries < HEMI_MAX_TRIES; ++tries) {
        for (int k = 0; k < 3; ++k)
            p[k] = orc_rng_f32(stream, (*ctr)++) * 2 - 1;
        if (lensqr3(p) < 1) break;
        if (tries == HEMI_MAX_TRIES - 1)
            for (int k = 0; k < 3; ++k) p[k] = p[k] * 0.5f;
    }
    if (dot3(p, normal) > 0.0f) {
        memcpy(out, p, 12);
    } else {
        for (int k = 0; k < 3; ++k) out[k] = p[k] * -1;
    }
}

/* trace.go:154-187 getMaterialParameters: Kd, times the nearest texel of map_Kd
 * when the material has one (texture.go:48-59 SampleUV / GetTexelCoordinates:
 * no V flip, repeat wrap by uv - floor(uv), image.RGBA.At widened to 16 bits). */
static void material_diffuse(const orc_scene *s, const OTriangle *t, float u, float v,
                             float diffuse[3])
{
    const float *kd = s->kd + 3 * t->material;
    diffuse[0] = kd[0]; diffuse[1] = kd[1]; diffuse[2] = kd[2];
    const OTexture *tex = &s->tex[t->material];
    if (!tex->pix) return;
    const float w0 = 1.0f - u - v;
    float tu = (t->tc[0][0] * w0 + t->tc[1][0] * u) + t->tc[2][0] * v;
    float tv = (t->tc[0][1] * w0 + t->tc[1][1] * u) + t->tc[2][1] * v;
    tu = tu - (float)floor((double)tu);
    tv = tv - (float)floor((double)tv);
    const float cx = tu * (float)tex->w, cy = tv * (float)tex->h;
    const int32_t px = (int32_t)cx, py = (int32_t)cy;
    float smp[3] = {0, 0, 0};                 /* At() outside the bounds: color.RGBA{} */
    if (px >= 0 && py >= 0 && px < tex->w && py < tex->h) {
        const uint8_t *p = tex->pix + 4 * ((size_t)py * (size_t)tex->w + (size_t)px);
        const float k = (float)(1.0 / 65535.0);
        for (int c = 0; c < 3; ++c) smp[c] = (float)((uint32_t)p[c] * 257u) * k;   /* c | c<<8 */
    }
    for (int c = 0; c < 3; ++c)
        diffuse[c] = diffuse[c] * (float)orc_go_min(orc_go_max((double)smp[c], 0.0), 1.0);
}

static void trace(const orc_scene *s, const orc_pass *pass, const orc_light *light,
                  const float ro[3], const float rd[3], uint64_t path_index, float rgb[3],
                  orc_counters *c)
{
    const int L = pass->settings.light_sample_rays;
    const int B = pass->settings.bounce_limit & 0xff;      /* uint8 in the reference */
    const float pi32 = (float)GO_PI;
    rgb[0] = rgb[1] = rgb[2] = 0.0f / 255.0f;
    cast_result res;
    if (!ray_cast(s, ro, rd, F32_MAX, &res, c)) return;

    static __thread float shading_terms[MAX_VERTS][3], brdf_terms[MAX_VERTS][3];
    int n_terms = 0;
    int indirect = 0;
    float cur_dir[3] = {rd[0], rd[1], rd[2]};
    uint64_t stream = orc_rng_stream(pass->rng_seed, path_index);
    uint32_t ctr = 0;
    uint64_t light_base = rng_u64(stream ^ 0xA5A5A5A5A5A5A5A5ULL, 0);
    uint64_t light_count = 0;

    for (;;) {
        float shading[3] = {0, 0, 0};
        float diffuse[3];
        material_diffuse(s, res.tri, res.u, res.v, diffuse);
        const float *normal = res.tri->normal;

        for (int i = 0; i < L; ++i) {
            float ls[3], pdf;
            uint64_t row = light_base + light_count++;
            if (s->light_index_mode == 1) row = (*s->light_cursor)++;
            light_sample(s, light, row, ls, &pdf);
            float sr[3] = {ls[0] - res.point[0], ls[1] - res.point[1], ls[2] - res.point[2]};
            float dist = len3(sr);
            float srn[3];
            normalize3(srn, sr);
            float li = dot3(srn, light->normal);
            if (li < 0) {
                cast_result sh;
                int ok = ray_cast(s, res.point, srn, dist, &sh, c);
                int lit = ok && (sh.t >= dist || (sh.tri != NULL && sh.tri->is_light));
                if (lit) {
                    float theta_l = (float)orc_go_max((double)(-li), 0.0);
                    float theta = (float)orc_go_max((double)dot3(srn, res.tri->normal), 0.0);
                    float r2 = lensqr3(sr);
                    float k = theta_l * theta / (r2 * pdf * pi32);
                    for (int q = 0; q < 3; ++q)
                        shading[q] = shading[q] + (diffuse[q] * light->emission[q]) * k;
                }
            }
        }
        float invl = 1 / (float)L;
        for (int q = 0; q < 3; ++q)
            shading_terms[n_terms][q] = clamp01(shading[q] * invl);

        if (indirect >= B) {
            brdf_terms[n_terms][0] = brdf_terms[n_terms][1] = brdf_terms[n_terms][2] = 0;
            n_terms++;
            break;
        }
        float hs[3], sample[3];
        random_in_hemisphere(stream, &ctr, res.tri->normal, hs);
        normalize3(sample, hs);
        float origin[3] = {res.point[0], res.point[1], res.point[2]};
        float vnormal[3] = {normal[0], normal[1], normal[2]};
        if (!ray_cast(s, origin, sample, F32_MAX, &res, c)) {
            brdf_terms[n_terms][0] = brdf_terms[n_terms][1] = brdf_terms[n_terms][2] = 0;
            n_terms++;
            break;
        }
        indirect++;
        float brdf_theta = dot3(cur_dir, sample) * -1;
        float theta = dot3(sample, vnormal);
        double pdf = orc_go_cos((double)brdf_theta) / GO_PI;
        float k = (float)(orc_go_cos((double)theta) / (GO_PI * pdf));
        for (int q = 0; q < 3; ++q)
            brdf_terms[n_terms][q] = diffuse[q] * k;
        n_terms++;
        memcpy(cur_dir, sample, 12);
        if (n_terms >= MAX_VERTS - 1) {   /* unreachable: B <= 255 */
            shading_terms[n_terms][0] = shading_terms[n_terms][1] = shading_terms[n_terms][2] = 0;
            brdf_terms[n_terms][0] = brdf_terms[n_terms][1] = brdf_terms[n_terms][2] = 0;
            n_terms++;
            break;
        }
    }
    for (int i = n_terms - 2; i >= 0; --i)
        for (int q = 0; q < 3; ++q)
            brdf_terms[i][q] = brdf_terms[i][q] * (shading_terms[i + 1][q] + brdf_terms[i + 1][q]);
    for (int q = 0; q < 3; ++q)
        rgb[q] = shading_terms[0][q] + brdf_terms[0][q];
}

/* ------------------------------------------------------------------ */
/* pixel loop + resolve: main.go:131-182 (and :250-301)                */
/* ------------------------------------------------------------------ */

static uint8_t to_u8(float c)
{
    float x = 255 * c;
    if (!(x == x)) return 0;          /* NaN: saturating conversion yields 0 */
    if (x <= 0) return 0;
    if (x >= 255) return 255;
    return (uint8_t)x;
}

void orc_resolve(const float *accum, int64_t pixels, int32_t divisor, int32_t gamma_on,
                 float gamma, uint8_t *rgba)
{
    float inv = 1.0f / (float)divisor;
    double g = (double)(1.0f / gamma);
    for (int64_t i = 0; i < pixels; ++i) {
        float c[3];
        for (int q = 0; q < 3; ++q) {
            c[q] = accum[3 * i + q] * inv;
            if (gamma_on) c[q] = (float)orc_go_pow((double)c[q], g);
            c[q] = clamp01(c[q]);
            rgba[4 * i + q] = to_u8(c[q]);
        }
        rgba[4 * i + 3] = 255;
    }
}

void orc_render_tile(orc_scene *s, const orc_pass *pass, int32_t sample_begin,
                     int32_t sample_count, int32_t incremental, float *accum,
                     uint8_t *rgba, float *radiance, orc_counters *counters)
{
    orc_camera cam;
    orc_light light;
    float tl[3], steps[2];
    pass_prepare(s, pass, &cam, &light);
    orc_camera_setup(&cam, pass->total_width < 0 ? 0 : pass->total_width,
                     pass->total_height < 0 ? 0 : pass->total_height, tl, steps);
    int64_t pixels = (int64_t)pass->width * pass->height;
    int32_t spp = cam.rays_per_pixel;
    float *acc = accum ? accum : (float *)calloc((size_t)(pixels > 0 ? pixels : 1) * 3, sizeof(float));
    for (int64_t i = 0; i < pixels; ++i) {
        int32_t x = (int32_t)(i % pass->width), y = (int32_t)(i / pass->width);
        for (int32_t j = sample_begin; j < sample_begin + sample_count; ++j) {
            uint64_t path = incremental ? (uint64_t)j * (uint64_t)pixels + (uint64_t)i
                                        : (uint64_t)i * (uint64_t)spp + (uint64_t)j;
            float o[3], d[3], rgb[3];
            camera_ray(s, &cam, tl, steps, pass->x_offset, pass->y_offset, x, y, path, o, d);
            trace(s, pass, &light, o, d, path, rgb, counters);
            if (radiance)
                memcpy(radiance + ((size_t)(j - sample_begin) * (size_t)pixels + (size_t)i) * 3, rgb, 12);
            for (int q = 0; q < 3; ++q) acc[3 * i + q] = acc[3 * i + q] + rgb[q];
        }
    }
    if (rgba)
        orc_resolve(acc, pixels, sample_begin + sample_count, pass->settings.gamma_correction,
                    pass->settings.gamma, rgba);
    if (!accum) free(acc);
}

/* main.go:117-154: the progress bookkeeping of render() around the same pixel
 * loop.  Records every ProgressUpdate(progress, "trace", ...) the reference
 * issues while tracing -- 0.0 first, one whenever the running sample index
 * passes updateIndex + updateInterval, 1.0 last -- with the value context.Rays
 * has at that moment (base_rays + rays cast so far).  Returns the number of
 * events (at most max_events are stored). */
int32_t orc_render_progress(orc_scene *s, const orc_pass *pass, uint64_t base_rays,
                            float *progress, uint64_t *rays, int32_t max_events)
{
    orc_camera cam;
    orc_light light;
    float tl[3], steps[2];
    orc_counters counters;
    memset(&counters, 0, sizeof counters);
    pass_prepare(s, pass, &cam, &light);
    orc_camera_setup(&cam, pass->total_width < 0 ? 0 : pass->total_width,
                     pass->total_height < 0 ? 0 : pass->total_height, tl, steps);
    int32_t n = 0;
#define ORC_EVENT(p) do { if (n < max_events) { progress[n] = (p); rays[n] = base_rays + counters.rays; } n++; } while (0)
    int pixel_count = pass->width * pass->height;
    int ray_count = pixel_count * cam.rays_per_pixel;
    ORC_EVENT(0.0f);
    int update_interval = (int)((float)ray_count / 10.0f);
    int update_index = 0;
    int ri = 0;
    for (int i = 0; i < pixel_count; ++i) {
        int32_t x = i % pass->width, y = i / pass->width;
        for (int j = 0; j < cam.rays_per_pixel; ++j) {
            if (ri > update_index + update_interval) {
                update_index = ri;
                ORC_EVENT((float)update_index / (float)ray_count);
            }
            ri += 1;
            uint64_t path = (uint64_t)i * (uint64_t)cam.rays_per_pixel + (uint64_t)j;
            float o[3], d[3], rgb[3];
            camera_ray(s, &cam, tl, steps, pass->x_offset, pass->y_offset, x, y, path, o, d);
            trace(s, pass, &light, o, d, path, rgb, &counters);
        }
    }
    ORC_EVENT(1.0f);
#undef ORC_EVENT
    return n;
}

