// emul.cpp -- lane-serial CPU driver around raytracer_amd/csrc/rt_core.h.
// Test-only: lets the CPU test-suite run the kernels' per-ray and per-vertex
// arithmetic (the same inline functions the HIP kernels call) against the
// oracle without a GPU.  It is never part of the shipped library.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../raytracer_amd/csrc/rt_core.h"

using namespace rtc;

extern "C" {

struct EmulScene {
    DevScene d;
};

struct EmulRay { float o[3]; float tmax; float dir[3]; float pad; };
struct EmulHit { int32_t tri_pos; float t, u, v; };

void emul_trace(const float *nodes, int32_t n_nodes, const float *tris, int32_t n_tris,
                const EmulRay *rays, int64_t n, int mode, EmulHit *hits, uint64_t *counters) {
    DevScene s{};
    s.nodes = reinterpret_cast<const f4 *>(nodes);
    s.tris = reinterpret_cast<const f4 *>(tris);
    s.n_nodes = n_nodes;
    s.n_tris = n_tris;
    Counters c;
    uint64_t tot[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const EmulRay &r = rays[i];
        c = Counters();
        if (mode == 1) {
            Ray<double> ray;
            ray_init(ray, r.o[0], r.o[1], r.o[2], r.dir[0], r.dir[1], r.dir[2]);
            Hit<double> h{(double)r.tmax, 0, 0, -1};
            traverse_ref_order<double, true>(s, ray, h, &c);
            hits[i] = EmulHit{h.pos, (float)h.t, (float)h.u, (float)h.v};
        } else {
            Ray<float> ray;
            ray_init(ray, r.o[0], r.o[1], r.o[2], r.dir[0], r.dir[1], r.dir[2]);
            Hit<float> h{r.tmax, 0, 0, -1};
            traverse_ref_order<float, true>(s, ray, h, &c);
            hits[i] = EmulHit{h.pos, h.t, h.u, h.v};
        }
        tot[0] += c.nodes_visited; tot[1] += c.box_tests; tot[2] += c.tris_fetched; tot[3] += c.tris_tested;
    }
    if (counters) {
        counters[0] = (uint64_t)n;
        for (int k = 0; k < 4; ++k) counters[1 + k] = tot[k];
    }
}

// near_form != 0: the near-origin form of the step (RT_MODE_FAST_NEAR_F32)
void emul_trace_fast(const float *wide, uint32_t fast_root, const float *tris, int32_t n_tris,
                     const EmulRay *rays, int64_t n, EmulHit *hits, uint64_t *counters, int near_form) {
    DevScene s{};
    s.wide = reinterpret_cast<const f4 *>(wide);
    s.fast_root = fast_root;
    s.tris = reinterpret_cast<const f4 *>(tris);
    s.n_tris = n_tris;
    uint64_t tot[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const EmulRay &r = rays[i];
        Counters c;
        Ray<float> ray;
        ray_init(ray, r.o[0], r.o[1], r.o[2], r.dir[0], r.dir[1], r.dir[2]);
        Hit<float> h{r.tmax, 0, 0, -1};
        if (near_form) traverse_ordered<true, true>(s, ray, h, &c);
        else traverse_ordered<true, false>(s, ray, h, &c);
        hits[i] = EmulHit{h.pos, h.t, h.u, h.v};
        tot[0] += c.nodes_visited; tot[1] += c.box_tests; tot[2] += c.tris_fetched; tot[3] += c.tris_tested;
    }
    if (counters) {
        counters[0] = (uint64_t)n;
        for (int k = 0; k < 4; ++k) counters[1 + k] = tot[k];
    }
}

// Optional texture inputs of the next emul_render calls (all NULL = untextured):
// per-triangle UVs in array order, per-material texture slot, per-slot RGBA8
// rows and (w, h).
static const float *g_tri_uv = nullptr;
static const int32_t *g_mat_tex = nullptr;
static const uint8_t *const *g_tex_data = nullptr;
static const int32_t *g_tex_dim = nullptr;
void emul_set_textures(const float *tri_uv, const int32_t *mat_tex, const uint8_t *const *tex_data,
                       const int32_t *tex_dim) {
    g_tri_uv = tri_uv;
    g_mat_tex = mat_tex;
    g_tex_data = tex_data;
    g_tex_dim = tex_dim;
}

// Serial path tracer over the device functions: the order of operations of
// k_raygen / k_shade / k_finish / k_accumulate / k_resolve for one tile.
void emul_render(const float *nodes, int32_t n_nodes, const float *tris, int32_t n_tris,
                 const int32_t *tri_mat, const float *kd, const double *cam_tab, int32_t n_cam,
                 const double *light_tab, int32_t n_light, const PassParams *pp,
                 int32_t sample_begin, int32_t sample_count, float *accum, float *radiance,
                 uint64_t *ray_count) {
    DevScene s{};
    s.nodes = reinterpret_cast<const f4 *>(nodes);
    s.tris = reinterpret_cast<const f4 *>(tris);
    std::vector<f4> shade((size_t)n_tris);   // (normal, material word) per triangle, as the library builds it
    for (int32_t i = 0; i < n_tris; ++i)
        shade[(size_t)i] = f4{tris[12 * (size_t)i], tris[12 * (size_t)i + 1], tris[12 * (size_t)i + 2], i2f(tri_mat[i])};
    s.tri_shade = shade.data();
    s.kd = kd;
    s.tri_uv = g_tri_uv;
    s.mat_tex = g_mat_tex;
    s.tex_data = g_tex_data;
    s.tex_dim = g_tex_dim;
    s.cam_tab = cam_tab;
    s.light_tab = light_tab;
    s.n_nodes = n_nodes; s.n_tris = n_tris; s.n_cam = n_cam; s.n_light = n_light;
    const PassParams &p = *pp;
    const int64_t pixels = (int64_t)p.width * p.height;
    uint64_t rays = 0;
    auto cast = [&](const v3 &o, const v3 &d, float tmax, Hit<float> &h) {
        Ray<float> ray;
        ray_init(ray, o.x, o.y, o.z, d.x, d.y, d.z);
        h = Hit<float>{tmax, 0, 0, -1};
        Counters c;
        traverse_ref_order<float, false>(s, ray, h, &c);
        rays++;
    };
    std::vector<v3> shading, brdf;
    for (int64_t pix = 0; pix < pixels; ++pix) {
        for (int32_t sm = sample_begin; sm < sample_begin + sample_count; ++sm) {
            const PathCtx ctx = locate_path(p, (uint32_t)pix, (uint32_t)sm);
            const uint64_t path = ctx.path_index;
            v3 o, d;
            camera_ray(s, p, ctx, o, d);
            shading.clear(); brdf.clear();
            const uint64_t stream = rng_stream(p.rng_seed, path);
            const uint64_t lbase = light_base(stream);
            uint32_t ctr = 0;
            Hit<float> h;
            cast(o, d, kF32Max, h);
            v3 rgb{0, 0, 0};
            if (h.t < kF32Max) {
                v3 cur = d;
                v3 ro = o, rd = d;
                for (int k = 0;; ++k) {
                    const v3 point{ro.x + rd.x * h.t, ro.y + rd.y * h.t, ro.z + rd.z * h.t};
                    const f4 q0 = s.tri_shade[h.pos];
                    const v3 normal{q0.x, q0.y, q0.z};
                    const v3 diffuse = material_diffuse(s, h.pos, f2i(q0.w), h.u, h.v);
                    v3 sh{0, 0, 0};
                    for (int i = 0; i < p.L; ++i) {
                        v3 dn, col; float dist;
                        if (shadow_candidate(s, p, point, normal, diffuse, lbase + (uint64_t)k * p.L + i, dn, dist, col)) {
                            Hit<float> shh;
                            cast(point, dn, dist, shh);
                            const bool reached = shh.t < kF32Max;
                            const bool lit = reached && (shh.t >= dist || (shh.pos >= 0 && (f2i(s.tri_shade[shh.pos].w) & kLightBit)));
                            if (lit) sh = v3{sh.x + col.x, sh.y + col.y, sh.z + col.z};
                        }
                    }
                    const float invl = 1 / (float)p.L;
                    shading.push_back(v3{gomath::clamp01(sh.x * invl), gomath::clamp01(sh.y * invl), gomath::clamp01(sh.z * invl)});
                    if (k >= p.B) { brdf.push_back(v3{0, 0, 0}); break; }
                    const v3 sample = normalize(random_in_hemisphere(stream, ctr, normal));
                    Hit<float> nh;
                    cast(point, sample, kF32Max, nh);
                    if (!(nh.t < kF32Max)) { brdf.push_back(v3{0, 0, 0}); break; }
                    const float f = brdf_factor(cur, sample, normal);
                    brdf.push_back(v3{diffuse.x * f, diffuse.y * f, diffuse.z * f});
                    cur = sample; ro = point; rd = sample; h = nh;
                }
                const int n = (int)shading.size();
                v3 x = brdf[(size_t)n - 1];
                for (int t = n - 2; t >= 0; --t) {
                    const v3 sn = shading[(size_t)t + 1], bt = brdf[(size_t)t];
                    x = v3{bt.x * (sn.x + x.x), bt.y * (sn.y + x.y), bt.z * (sn.z + x.z)};
                }
                rgb = v3{shading[0].x + x.x, shading[0].y + x.y, shading[0].z + x.z};
            }
            if (radiance) {
                float *r = radiance + ((size_t)(sm - sample_begin) * (size_t)pixels + (size_t)pix) * 3;
                r[0] = rgb.x; r[1] = rgb.y; r[2] = rgb.z;
            }
            accum[3 * pix] = accum[3 * pix] + rgb.x;
            accum[3 * pix + 1] = accum[3 * pix + 1] + rgb.y;
            accum[3 * pix + 2] = accum[3 * pix + 2] + rgb.z;
        }
    }
    if (ray_count) *ray_count = rays;
}

void emul_resolve(const float *accum, int64_t pixels, int32_t divisor, int32_t gamma_on, float gamma, uint8_t *rgba) {
    const float inv = 1.0f / (float)divisor;
    const double inv_gamma = (double)(1.0f / gamma);
    for (int64_t i = 0; i < pixels; ++i) {
        for (int q = 0; q < 3; ++q) rgba[4 * i + q] = resolve_channel(accum[3 * i + q], inv, gamma_on != 0, inv_gamma);
        rgba[4 * i + 3] = 255;
    }
}

int32_t emul_pass_params_size() { return (int32_t)sizeof(PassParams); }

}  // extern "C"
