// scene_host.cpp -- host-side set-up that feeds the device buffers:
// per-triangle precompute (models/triangle.go:45-65), area-light discovery
// (models/context.go:172-226, models/triangle.go:105-138), projection-plane
// set-up (models/camera.go:43-85) and the default sample tables.
// float32 arithmetic follows github.com/go-gl/mathgl v1.0.0 (mgl32): products
// and sums rounded one at a time, Normalize multiplies by 1/Len.
#include <cmath>
#include <cstring>

#include "gomath.h"
#include "rt_core.h"
#include "rt_internal.h"

namespace rt {
namespace {

thread_local std::string g_error;


inline void cross3(float out[3], const float a[3], const float b[3]) {
    float r0 = a[1] * b[2] - a[2] * b[1];
    float r1 = a[2] * b[0] - a[0] * b[2];
    float r2 = a[0] * b[1] - a[1] * b[0];
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
}
inline float len3(const float a[3]) {
    return (float)std::sqrt((double)(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]));
}
inline void normalize3(float out[3], const float a[3]) {
    float l = 1.0f / len3(a);
    out[0] = a[0] * l;
    out[1] = a[1] * l;
    out[2] = a[2] * l;
}

struct Edges {
    float e0[3], e1[3], e2[3];
};
Edges edges_of(const float v[9]) {
    Edges e;
    for (int k = 0; k < 3; ++k) {
        e.e0[k] = v[3 + k] - v[k];
        e.e1[k] = v[6 + k] - v[3 + k];
        e.e2[k] = v[k] - v[6 + k];
    }
    return e;
}

// Triangle.GetShortestEdge / GetMiddleEdge (triangle.go:105-138)
const float *shortest_edge(const Edges &e) {
    if (len3(e.e0) < len3(e.e1)) {
        if (len3(e.e0) < len3(e.e2)) return e.e0;
        return e.e2;
    }
    if (len3(e.e1) < len3(e.e2)) return e.e1;
    return e.e2;
}
const float *middle_edge(const Edges &e) {
    if (len3(e.e0) < len3(e.e1)) {
        if (len3(e.e1) < len3(e.e2)) return e.e1;
        if (len3(e.e0) < len3(e.e2)) return e.e2;
        return e.e0;
    }
    if (len3(e.e2) < len3(e.e1)) return e.e1;
    if (len3(e.e0) < len3(e.e2)) return e.e0;
    return e.e2;
}

// mgl32.QuatRotate(angle, axis).Rotate(v)
void quat_rotate(float out[3], float angle, const float axis[3], const float v[3]) {
    float c = (float)gomath::go_cos((double)(angle / 2));
    float s = (float)gomath::go_sin((double)(angle / 2));
    float qv[3] = {axis[0] * s, axis[1] * s, axis[2] * s};
    float cr[3], q2[3], cr2[3];
    cross3(cr, qv, v);
    float w2 = 2 * c;
    for (int k = 0; k < 3; ++k) q2[k] = qv[k] * 2;
    cross3(cr2, q2, cr);
    for (int k = 0; k < 3; ++k) out[k] = (v[k] + cr[k] * w2) + cr2[k];
}

}  // namespace

rt_status fail(rt_status code, const std::string &msg) {
    g_error = msg;
    return code;
}

const std::string &last_error_text() { return g_error; }

void make_tri_record(const float v[9], TriRecord &out) {
    float a[3], b[3], c[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = v[3 + k] - v[k];
        b[k] = v[6 + k] - v[k];
    }
    cross3(c, a, b);
    normalize3(out.normal, c);
    for (int k = 0; k < 3; ++k) {
        out.v0[k] = v[k];
        out.e0[k] = v[3 + k] - v[k];
        // Edge2 = v0 - v2, and the intersection uses Edge2.Mul(-1)
        out.v0v2[k] = (v[k] - v[6 + k]) * -1.0f;
    }
}

// The reference looks for the light inside RenderContext.Initialize
// (context.go:180), before buildBVH / loadBVH sort the triangle array
// (main.go:65-90): the walk is over the triangles in input (OBJ) order, and the
// LAST "Light" triangle in that order sets normal, up and size.
void find_light(const rt_mesh &mesh, rt_light &out) {
    std::memset(&out, 0, sizeof out);
    float mn[3] = {kF32Max, kF32Max, kF32Max}, mx[3] = {-kF32Max, -kF32Max, -kF32Max};
    float normal[3] = {0, 0, 0}, up[3] = {0, 0, 0}, shortest[3] = {0, 0, 0}, middle[3] = {0, 0, 0};
    bool found = false;
    for (int32_t id = 0; id < mesh.n_tris; ++id) {
        int32_t mat = mesh.material ? mesh.material[id] : 0;
        if (!mesh.is_light[mat]) continue;
        found = true;
        const float *v = mesh.verts + 9 * (size_t)id;
        TriRecord rec;
        make_tri_record(v, rec);
        Edges e = edges_of(v);
        std::memcpy(normal, rec.normal, 12);
        std::memcpy(shortest, shortest_edge(e), 12);
        std::memcpy(middle, middle_edge(e), 12);
        float c[3];
        cross3(c, shortest, normal);
        normalize3(up, c);
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                float x = v[3 * j + k];
                if (x < mn[k]) mn[k] = x;
                if (x > mx[k]) mx[k] = x;
            }
    }
    out.found = found ? 1 : 0;
    if (!found) return;
    float center[3], right[3];
    for (int k = 0; k < 3; ++k) center[k] = (mn[k] + mx[k]) / 2.0f;
    cross3(right, normal, up);
    // Translate3D(center).Mul4(Mat3FromCols(normal x up, up, normal).Mat4())
    for (int k = 0; k < 3; ++k) {
        out.transform[k] = right[k];
        out.transform[4 + k] = up[k];
        out.transform[8 + k] = normal[k];
        out.transform[12 + k] = center[k];
    }
    out.transform[15] = 1;
    out.size[0] = len3(shortest) / 2.0f;
    out.size[1] = len3(middle) / 2.0f;
    out.emission[0] = out.emission[1] = out.emission[2] = 100;
    std::memcpy(out.normal, normal, 12);
}

void camera_setup(const rt_camera &cam, int32_t total_w, int32_t total_h, float top_left[3],
                  float steps[2]) {
    float tl[3], br[3];
    const float ppd = cam.projection_plane_distance;
    if (cam.projection == 0) {
        const float pi32 = (float)gomath::kPi;
        float vha = pi32 * (cam.field_of_view / 2.0f) / 180.0f;
        float hha = vha * ((float)total_w / (float)total_h);
        const float fwd[3] = {0, 0, ppd};
        const float ax_y[3] = {0, 1, 0}, ax_x[3] = {1, 0, 0};
        float left[3], right[3], top[3], bottom[3];
        quat_rotate(left, -hha, ax_y, fwd);
        quat_rotate(right, hha, ax_y, fwd);
        quat_rotate(top, -vha, ax_x, fwd);
        quat_rotate(bottom, vha, ax_x, fwd);
        float sl = ppd / left[2], sr = ppd / right[2], st = ppd / top[2], sb = ppd / bottom[2];
        tl[0] = left[0] * sl;
        tl[1] = top[1] * st;
        tl[2] = ppd;
        br[0] = right[0] * sr;
        br[1] = bottom[1] * sb;
        br[2] = ppd;
    } else {
        float hw = cam.ortographic_size * ((float)total_w / (float)total_h);
        tl[0] = -hw;
        tl[1] = cam.ortographic_size;
        tl[2] = ppd;
        br[0] = hw;
        br[1] = -cam.ortographic_size;
        br[2] = ppd;
    }
    steps[1] = (tl[1] - br[1]) / (float)total_h;
    steps[0] = (br[0] - tl[0]) / (float)total_w;
    std::memcpy(top_left, tl, 12);
}

// Halton points in bases 2 and 3 with a per-digit-position random permutation
// of the digits (a plain scrambled Halton; the reference uses gonum's
// Owen-scrambled variant seeded from the global RNG, which cannot be
// reproduced, so only the table's role and shape are kept).
void default_sample_table(int32_t n, uint64_t seed, double *out) {
    auto mix = [](uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    };
    const int bases[2] = {2, 3};
    for (int d = 0; d < 2; ++d) {
        const int b = bases[d];
        // digit permutations for up to 40 digit positions; position p maps digit
        // x -> perm[p][x], with perm[p][0] left free (leading zeros scramble too)
        int perm[40][3];
        for (int p = 0; p < 40; ++p) {
            for (int x = 0; x < b; ++x) perm[p][x] = x;
            uint64_t r = mix(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(d * 64 + p + 1));
            for (int x = b - 1; x > 0; --x) {
                int j = (int)(r % (uint64_t)(x + 1));
                r /= (uint64_t)(x + 1);
                int t = perm[p][x];
                perm[p][x] = perm[p][j];
                perm[p][j] = t;
            }
        }
        const int digits = (b == 2) ? 32 : 20;
        for (int32_t i = 0; i < n; ++i) {
            uint64_t k = (uint64_t)i + 1;
            double inv = 1.0 / b, f = inv, x = 0;
            for (int p = 0; p < digits; ++p) {
                int digit = (int)(k % (uint64_t)b);
                k /= (uint64_t)b;
                x += perm[p][digit] * f;
                f *= inv;
            }
            if (x >= 1.0) x = std::nextafter(1.0, 0.0);
            out[2 * (size_t)i + d] = x;
        }
    }
}

// context.go:245-275 (RenderPass.Initialize) + camera.go:43-85, on the host
rt_status prepare_pass(const rt_light &scene_light, const rt_pass &pass, rtc::PassParams &pp) {
    using rtc::v3;
    std::memset(&pp, 0, sizeof pp);
    if (pass.width <= 0 || pass.height <= 0) return fail(RT_ERR_INVALID, "pass width/height must be positive");
    if (pass.camera.rays_per_pixel <= 0) return fail(RT_ERR_INVALID, "Camera.RaysPerPixel must be positive");
    if (pass.settings.light_sample_rays < 0) return fail(RT_ERR_INVALID, "LightSampleRays is negative");
    rt_camera cam = pass.camera;
    const float tr = cam.transform[0] + cam.transform[5] + cam.transform[10] + cam.transform[15];
    if (std::fabs((double)tr) < 0.001) {
        std::memset(cam.transform, 0, sizeof cam.transform);
        cam.transform[0] = cam.transform[5] = cam.transform[10] = cam.transform[15] = 1;
    }
    std::memcpy(pp.cam_transform, cam.transform, sizeof pp.cam_transform);
    const int32_t tw = pass.total_width < 0 ? 0 : pass.total_width;
    const int32_t th = pass.total_height < 0 ? 0 : pass.total_height;
    camera_setup(cam, tw, th, pp.top_left, pp.steps);
    pp.ppd = cam.projection_plane_distance;
    pp.projection = cam.projection;
    pp.spp = cam.rays_per_pixel;
    pp.x_offset = pass.x_offset;
    pp.y_offset = pass.y_offset;
    pp.width = pass.width;
    pp.height = pass.height;
    pp.rng_seed = pass.rng_seed;
    pp.L = pass.settings.light_sample_rays;
    pp.B = pass.settings.bounce_limit & 0xff;
    rt_light light = scene_light;
    const rt_settings &st = pass.settings;
    if (!light.found || st.force_debug_light) {
        const float *m = st.debug_light_at_camera ? cam.transform : st.debug_light_transform;
        std::memcpy(light.transform, m, sizeof light.transform);
        const v3 f = rtc::transform_coordinate(v3{0, 0, -1}, m);
        light.normal[0] = f.x - m[12];
        light.normal[1] = f.y - m[13];
        light.normal[2] = f.z - m[14];
        light.size[0] = light.size[1] = st.debug_light_size;
    }
    for (int k = 0; k < 3; ++k) light.emission[k] = 1.0f * st.light_intensity;
    std::memcpy(pp.light_transform, light.transform, sizeof pp.light_transform);
    std::memcpy(pp.light_size, light.size, sizeof pp.light_size);
    std::memcpy(pp.light_emission, light.emission, sizeof pp.light_emission);
    std::memcpy(pp.light_normal, light.normal, sizeof pp.light_normal);
    return RT_OK;
}


}  // namespace rt
