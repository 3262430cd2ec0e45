// backend.cpp -- the reference's six string-typed entry points
// (src/backend/main.go:39-344) on top of the native library: JSON in, JSON out,
// state in an rtb_context instead of package globals.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "../../include/rt_backend.h"
#include "backend.h"
#include "json.h"
#include "rt_device.h"
#include "rt_internal.h"

namespace rt {
rt_status bvh_to_json(const Bvh &bvh, std::string &out);
rt_status bvh_from_json(const rt_mesh &mesh, const char *json, size_t len, Bvh &out);
}  // namespace rt

using rt::fail;

struct rtb_context {
    int32_t device = 0;
    int32_t mode = RT_MODE_STRICT_F32;
    rtb_progress_fn progress = nullptr;
    void *progress_user = nullptr;

    // models.RenderContext (context.go:11-38)
    bool initialized = false;
    bool debug = false, use_bvh = true;
    int32_t max_leaf = 6, max_depth = 16, worker_id = 0;
    rt::ObjScene obj;
    std::vector<float> kd;
    std::vector<uint8_t> is_light;
    std::vector<int32_t> mat_tex;
    struct Texture {
        std::string name;
        std::vector<uint8_t> rgba;
        int32_t w = 0, h = 0;
    };
    std::vector<Texture> textures;
    rt_mesh mesh{};
    std::unique_ptr<rt::Bvh> bvh;          // installed by loadBVH
    rt::Scene *scene = nullptr;            // device copy, made at the first render
    uint64_t rays = 0;
    int32_t active_render_key = -1;
    uint32_t progress_deliveries = 0;      // separate moments at which the last render sent trace updates

    // incremental render state (main.go:190-194)
    bool inc_ready = false;
    rt_pass inc_pass{};
    int32_t inc_index = 0;
    float *d_inc_accum = nullptr;
    uint8_t *d_inc_rgba = nullptr;

    std::vector<uint8_t> last_image;
    int32_t last_w = 0, last_h = 0;

    ~rtb_context() { release_device(); }
    void release_device() {
        if (d_inc_accum) (void)hipFree(d_inc_accum);
        if (d_inc_rgba) (void)hipFree(d_inc_rgba);
        d_inc_accum = nullptr;
        d_inc_rgba = nullptr;
        if (scene) rt::scene_destroy(scene);
        scene = nullptr;
    }
};

namespace {

void send_progress(rtb_context *c, float progress, const char *event, int32_t task_id) {
    if (!c->progress) return;
    std::string s = "{\"progress\":";
    rtjson::append_f32(s, progress);
    s += ",\"event\":\"";
    s += event;
    s += "\",\"taskId\":" + std::to_string(task_id) + ",\"rays\":" + std::to_string(c->rays) + "}";
    c->progress(s.c_str(), c->progress_user);
}

char *dup_string(const std::string &s, size_t *len) {
    char *p = static_cast<char *>(std::malloc(s.size() + 1));
    if (!p) return nullptr;
    std::memcpy(p, s.c_str(), s.size() + 1);
    if (len) *len = s.size();
    return p;
}

// models.RenderResult.Output (result.go:8-21) around an image.RGBA
std::string result_json(int exit_code, const std::string &message, const uint8_t *pix, int32_t w, int32_t h) {
    std::string s = "{\"exitCode\":" + std::to_string(exit_code) + ",\"message\":";
    rtjson::append_escaped(s, message);
    s += ",\"imageData\":{\"Pix\":\"";
    s += rt::base64_encode(pix, (size_t)w * (size_t)h * 4);
    s += "\",\"Stride\":" + std::to_string(4 * w) + ",\"Rect\":{\"Min\":{\"X\":0,\"Y\":0},\"Max\":{\"X\":" +
         std::to_string(w) + ",\"Y\":" + std::to_string(h) + "}}}}";
    return s;
}

// main.go:308-325 handleError: exitCode -1, the message, a blank 500x500 image
rt_status error_result(rt_status st, char **out, size_t *len) {
    const std::string msg = rt::last_error_text().empty() ? "Unknown error" : rt::last_error_text();
    if (out) {
        std::vector<uint8_t> blank((size_t)500 * 500 * 4, 0);
        *out = dup_string(result_json(-1, msg, blank.data(), 500, 500), len);
    }
    return st;
}

bool read_mat16(const rtjson::Value *v, float out[16]) {
    if (!v || v->kind != rtjson::Value::Array || v->arr.size() != 16) return false;
    for (int i = 0; i < 16; ++i) {
        if (v->arr[(size_t)i].kind != rtjson::Value::Number) return false;
        out[i] = (float)v->arr[(size_t)i].num;
    }
    return true;
}

// encoding/json into models.RenderPass (context.go:40-52, camera.go:19-31,
// settings.go:5-16): absent fields keep their zero value, a value of the
// wrong JSON type is an error.
rt_status parse_pass(const char *json, rt_pass &p) {
    std::memset(&p, 0, sizeof p);
    if (!json) return fail(RT_ERR_INVALID, "pass JSON is NULL");
    rtjson::Value doc;
    std::string err;
    if (!rtjson::parse(json, std::strlen(json), doc, err)) return fail(RT_ERR_PARSE, "render pass: " + err);
    if (doc.kind != rtjson::Value::Object) return fail(RT_ERR_PARSE, "render pass: JSON document is not an object");
    bool bad = false;
    std::string bad_name;
    auto num = [&](const rtjson::Value &o, const char *name, double dflt) {
        const rtjson::Value *v = o.get(name);
        if (!v || v->is_null()) return dflt;
        if (v->kind != rtjson::Value::Number) {
            bad = true;
            bad_name = name;
            return dflt;
        }
        return v->num;
    };
    // integer struct fields: encoding/json refuses fractions and values outside the type
    auto integer = [&](const rtjson::Value &o, const char *name, double lo, double hi) -> double {
        const double v = num(o, name, 0);
        if (!(v >= lo && v <= hi) || v != std::floor(v)) {
            bad = true;
            bad_name = name;
            return 0;
        }
        return v;
    };
    auto int32 = [&](const rtjson::Value &o, const char *name) {
        return (int32_t)integer(o, name, -2147483648.0, 2147483647.0);
    };
    auto boolean = [&](const rtjson::Value &o, const char *name) {
        const rtjson::Value *v = o.get(name);
        if (!v || v->is_null()) return false;
        if (v->kind != rtjson::Value::Bool) {
            bad = true;
            bad_name = name;
            return false;
        }
        return v->b;
    };
    if (const rtjson::Value *cam = doc.get("Camera")) {
        if (cam->kind == rtjson::Value::Object) {
            const rtjson::Value *t = cam->get("Transform");
            if (t && !t->is_null() && !read_mat16(t, p.camera.transform)) {
                bad = true;
                bad_name = "Camera.Transform";
            }
            p.camera.projection_plane_distance = (float)num(*cam, "ProjectionPlaneDistance", 0);
            p.camera.rays_per_pixel = int32(*cam, "RaysPerPixel");
            p.camera.projection = int32(*cam, "Projection");
            p.camera.field_of_view = (float)num(*cam, "FieldOfView", 0);
            p.camera.ortographic_size = (float)num(*cam, "OrtographicSize", 0);
        } else if (!cam->is_null()) {
            bad = true;
            bad_name = "Camera";
        }
    }
    p.total_width = int32(doc, "TotalWidth");
    p.total_height = int32(doc, "TotalHeight");
    p.task_id = int32(doc, "TaskID");
    p.rng_seed = (int64_t)integer(doc, "RNGSeed", -9223372036854775808.0, 9223372036854774784.0);
    p.x_offset = int32(doc, "XOffset");
    p.y_offset = int32(doc, "YOffset");
    p.width = int32(doc, "Width");
    p.height = int32(doc, "Height");
    p.render_key = int32(doc, "RenderKey");
    if (const rtjson::Value *s = doc.get("Settings")) {
        if (s->kind == rtjson::Value::Object) {
            p.settings.draw_surface_normal = boolean(*s, "DrawSurfaceNormal");
            p.settings.gamma_correction = boolean(*s, "GammaCorrection");
            p.settings.gamma = (float)num(*s, "Gamma", 0);
            p.settings.light_sample_rays = int32(*s, "LightSampleRays");
            p.settings.bounce_limit = (int32_t)integer(*s, "BounceLimit", 0, 255);   // uint8 in the reference
            p.settings.light_intensity = (float)num(*s, "LightIntensity", 0);
            p.settings.debug_light_size = (float)num(*s, "DebugLightSize", 0);
            p.settings.force_debug_light = boolean(*s, "ForceDebugLight");
            p.settings.debug_light_at_camera = boolean(*s, "DebugLightAtCamera");
            const rtjson::Value *t = s->get("DebugLightTransform");
            if (t && !t->is_null() && !read_mat16(t, p.settings.debug_light_transform)) {
                bad = true;
                bad_name = "Settings.DebugLightTransform";
            }
        } else if (!s->is_null()) {
            bad = true;
            bad_name = "Settings";
        }
    }
    if (bad) return fail(RT_ERR_PARSE, "json: cannot unmarshal value into Go struct field RenderPass." + bad_name);
    return RT_OK;
}

rt_status ensure_scene(rtb_context *c) {
    if (!c->initialized) return fail(RT_ERR_STATE, "render before initialize");
    if (!c->bvh) return fail(RT_ERR_STATE, "render before loadBVH: the context has no BVH");
    if (c->scene) return RT_OK;
    rt_status st = rt::scene_create(c->mesh, *c->bvh, c->device, &c->scene);
    if (st != RT_OK) return st;
    for (size_t i = 0; i < c->textures.size(); ++i) {
        const auto &t = c->textures[i];
        if (t.rgba.empty()) continue;
        st = rt::scene_set_texture(*c->scene, (int32_t)i, t.rgba.data(), t.w, t.h);
        if (st != RT_OK) return st;
    }
    return RT_OK;
}

}  // namespace

#define RTB_GUARD_BEGIN try {
#define RTB_GUARD_END                                                            \
    }                                                                            \
    catch (const std::bad_alloc &) {                                             \
        return fail(RT_ERR_NOMEM, "out of host memory");                         \
    }                                                                            \
    catch (const std::exception &e) {                                            \
        return fail(RT_ERR_INVALID, std::string("internal error: ") + e.what()); \
    }                                                                            \
    catch (...) {                                                                \
        return fail(RT_ERR_INVALID, "internal error");                           \
    }

extern "C" {

rt_status rtb_create(int32_t device, rtb_context **out) {
    RTB_GUARD_BEGIN
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    if (device < 0) return fail(RT_ERR_INVALID, "device index is negative");
    *out = new rtb_context;
    (*out)->device = device;
    return RT_OK;
    RTB_GUARD_END
}

void rtb_destroy(rtb_context *ctx) { delete ctx; }

void rtb_set_progress_callback(rtb_context *ctx, rtb_progress_fn fn, void *user) {
    if (!ctx) return;
    ctx->progress = fn;
    ctx->progress_user = user;
}

rt_status rtb_set_mode(rtb_context *ctx, int32_t mode) {
    if (!ctx) return fail(RT_ERR_INVALID, "context is NULL");
    if (mode != RT_MODE_STRICT_F32 && mode != RT_MODE_FP64 && mode != RT_MODE_FAST_F32 && mode != RT_MODE_FAST_NEAR_F32)
        return fail(RT_ERR_INVALID, "unknown traversal mode " + std::to_string(mode));
    ctx->mode = mode;
    return RT_OK;
}

rt_status rtb_initialize(rtb_context *ctx, const char *context_json, const uint8_t *const *textures,
                         const size_t *texture_sizes, int32_t n_textures) {
    RTB_GUARD_BEGIN
    if (!ctx || !context_json) return fail(RT_ERR_INVALID, "NULL argument");
    rtjson::Value doc;
    std::string err;
    if (!rtjson::parse(context_json, std::strlen(context_json), doc, err))
        return fail(RT_ERR_PARSE, "render context: " + err);
    if (doc.kind != rtjson::Value::Object) return fail(RT_ERR_PARSE, "render context: JSON document is not an object");
    ctx->release_device();
    ctx->initialized = false;
    ctx->bvh.reset();
    ctx->inc_ready = false;
    ctx->rays = 0;   // context.go:57
    ctx->debug = doc.bool_or("Debug", false);
    ctx->use_bvh = doc.bool_or("UseBVH", false);
    auto int_field = [&](const char *name, int32_t &out) {   // encoding/json: a Go int takes whole numbers only
        const double v = doc.number_or(name, 0);
        if (!(v >= -2147483648.0 && v <= 2147483647.0) || v != std::floor(v)) return false;
        out = (int32_t)v;
        return true;
    };
    if (!int_field("BVHMaxLeafSize", ctx->max_leaf) || !int_field("BVHMaxDepth", ctx->max_depth) ||
        !int_field("WorkerId", ctx->worker_id))
        return fail(RT_ERR_PARSE, "render context: BVHMaxLeafSize, BVHMaxDepth and WorkerId must be whole numbers");
    const std::string obj = doc.string_or("ObjBuffer", ""), mtl = doc.string_or("MtlBuffer", "");
    send_progress(ctx, 0.0f, "RenderContext.Initialize", -1);
    if (obj.empty() || mtl.empty())
        return fail(RT_ERR_INVALID, "render context: ObjBuffer and MtlBuffer must both be non-empty");
    rt_status st = rt::parse_obj(obj, mtl, ctx->obj);
    if (st != RT_OK) return st;

    // RawTextures[i].Name <-> textures[i] (context.go:94-100)
    ctx->textures.clear();
    if (const rtjson::Value *raw = doc.get("RawTextures")) {
        if (raw->kind == rtjson::Value::Array) {
            for (size_t i = 0; i < raw->arr.size(); ++i) {
                rtb_context::Texture t;
                t.name = raw->arr[i].string_or("Name", "");
                if ((int32_t)i >= n_textures || !textures || !texture_sizes || !textures[i])
                    return fail(RT_ERR_INVALID, "render context: no bytes given for texture '" + t.name + "'");
                st = rt::decode_png(textures[i], texture_sizes[i], t.rgba, t.w, t.h);
                if (st != RT_OK) return st;
                ctx->textures.push_back(std::move(t));
            }
        }
    }
    const size_t nm = ctx->obj.materials.size();
    ctx->kd.resize(nm * 3);
    ctx->is_light.resize(nm);
    ctx->mat_tex.assign(nm, -1);
    for (size_t m = 0; m < nm; ++m) {
        const rt::ObjMaterial &mat = ctx->obj.materials[m];
        std::memcpy(&ctx->kd[3 * m], mat.kd, 12);
        ctx->is_light[m] = mat.name == "Light" ? 1 : 0;   // triangle.go:58
        if (!mat.map_kd.empty())   // TextureLookup[Material.MapKd] (trace.go:162-164)
            for (size_t t = 0; t < ctx->textures.size(); ++t)
                if (ctx->textures[t].name == mat.map_kd) ctx->mat_tex[m] = (int32_t)t;
    }
    rt_mesh &mesh = ctx->mesh;
    mesh.verts = ctx->obj.verts.data();
    mesh.material = ctx->obj.material.data();
    mesh.group_index = ctx->obj.group_index.data();
    mesh.texcoords = (ctx->obj.has_texcoords && !ctx->textures.empty()) ? ctx->obj.texcoords.data() : nullptr;
    mesh.n_tris = ctx->obj.n_tris();
    mesh.kd = ctx->kd.data();
    mesh.is_light = ctx->is_light.data();
    mesh.texture = ctx->textures.empty() ? nullptr : ctx->mat_tex.data();
    mesh.n_materials = (int32_t)nm;
    st = rt::check_mesh(&mesh);
    if (st != RT_OK) return st;
    ctx->initialized = true;
    send_progress(ctx, 1.0f, "RenderContext.Initialize", -1);
    return RT_OK;
    RTB_GUARD_END
}

rt_status rtb_build_bvh(rtb_context *ctx, char **bvh_json, size_t *len) {
    RTB_GUARD_BEGIN
    if (!ctx || !bvh_json) return fail(RT_ERR_INVALID, "NULL argument");
    *bvh_json = nullptr;
    if (!ctx->initialized) return fail(RT_ERR_STATE, "buildBVH before initialize");
    send_progress(ctx, 0.0f, "RenderContext.BuildBVH", -1);
    rt::Bvh bvh;
    rt_status st = rt::build_bvh(ctx->mesh, ctx->use_bvh, ctx->max_leaf, ctx->max_depth, bvh);
    if (st != RT_OK) return st;
    send_progress(ctx, 1.0f, "RenderContext.BuildBVH", -1);
    std::string s;
    st = rt::bvh_to_json(bvh, s);
    if (st != RT_OK) return st;
    *bvh_json = dup_string(s, len);
    if (!*bvh_json) return fail(RT_ERR_NOMEM, "out of host memory");
    return RT_OK;
    RTB_GUARD_END
}

rt_status rtb_load_bvh(rtb_context *ctx, const char *bvh_json, size_t len) {
    RTB_GUARD_BEGIN
    if (!ctx || !bvh_json) return fail(RT_ERR_INVALID, "NULL argument");
    if (!ctx->initialized) return fail(RT_ERR_STATE, "loadBVH before initialize");
    send_progress(ctx, 0.0f, "RenderContext.LoadBVH", -1);
    std::unique_ptr<rt::Bvh> bvh(new rt::Bvh);
    rt_status st = rt::bvh_from_json(ctx->mesh, bvh_json, len, *bvh);
    if (st != RT_OK) return st;
    ctx->release_device();   // a scene built from an older tree is stale
    ctx->inc_ready = false;
    ctx->bvh = std::move(bvh);
    send_progress(ctx, 1.0f, "RenderContext.LoadBVH", -1);
    return RT_OK;
    RTB_GUARD_END
}

rt_status rtb_render(rtb_context *ctx, const char *pass_json, char **result_json_out, size_t *len) {
    RTB_GUARD_BEGIN
    if (!ctx || !result_json_out) return fail(RT_ERR_INVALID, "NULL argument");
    *result_json_out = nullptr;
    rt_pass pass;
    rt_status st = parse_pass(pass_json, pass);
    if (st != RT_OK) return error_result(st, result_json_out, len);
    if (ctx->active_render_key != pass.render_key) {   // main.go:107-110
        ctx->active_render_key = pass.render_key;
        ctx->rays = 0;
    }
    st = ensure_scene(ctx);
    if (st != RT_OK) return error_result(st, result_json_out, len);
    send_progress(ctx, 0.0f, "trace", pass.task_id);
    if (pass.width <= 0 || pass.height <= 0) {
        st = fail(RT_ERR_INVALID, "pass width/height must be positive");
        return error_result(st, result_json_out, len);
    }
    ctx->last_image.assign((size_t)pass.width * (size_t)pass.height * 4, 0);
    // main.go:117-122: an update whenever the running sample index passes
    // updateIndex + updateInterval, each with the value the reference's ray counter has
    // at that sample index.  A small tile is traced as a whole and its events go out
    // together once it is done (a preset tile takes a millisecond); a large one (from
    // kProgressiveSamples camera samples) is traced in ten pixel ranges and the events
    // whose sample indices a range completes go out as soon as it is done, so that a
    // caller sees progress while a long pass renders, as with the reference.
    constexpr int64_t kProgressiveSamples = 1 << 22;
    rt::ProgressBins bins;
    const int64_t ray_count = (int64_t)pass.width * (int64_t)pass.height * (int64_t)pass.camera.rays_per_pixel;
    const int64_t interval = (int64_t)((float)ray_count / 10.0f);
    const bool want_progress = ctx->progress != nullptr && pass.camera.rays_per_pixel > 0 && ray_count < (1LL << 31);
    if (want_progress) {
        bins.step = (uint32_t)(interval + 1);
        int64_t events = (ray_count - 1) / (interval + 1);           // sample indices k * step <= ray_count - 1
        if (events > rt::kProgressBins - 1) events = rt::kProgressBins - 1;
        bins.n_bins = (int32_t)events + 1;
    }
    const uint64_t rays_before = ctx->rays;
    int32_t sent = 0;                      // events 1..sent have gone out
    ctx->progress_deliveries = 0;
    auto send_up_to = [&](int32_t complete) {
        // event k reports the rays of the camera samples below k * step: bins 0..k-1
        if (complete > bins.n_bins - 1) complete = bins.n_bins - 1;
        if (complete <= sent) return;
        uint64_t running = 0;
        for (int32_t k = 1; k <= complete; ++k) {
            running += bins.rays[k - 1];
            if (k <= sent) continue;
            ctx->rays = rays_before + running;
            send_progress(ctx, (float)((int64_t)k * (interval + 1)) / (float)ray_count, "trace", pass.task_id);
        }
        ctx->rays = rays_before;
        sent = complete;
        ctx->progress_deliveries++;
    };
    if (want_progress && ray_count >= kProgressiveSamples) {
        bins.chunks = 10;
        bins.on_range = send_up_to;
    }
    rt_stats stats;
    st = rt::render_tile_host(*ctx->scene, pass, ctx->mode, nullptr, ctx->last_image.data(), &stats,
                              want_progress ? &bins : nullptr);
    if (st != RT_OK) return error_result(st, result_json_out, len);
    ctx->last_w = pass.width;
    ctx->last_h = pass.height;
    if (want_progress) send_up_to(bins.n_bins - 1);      // whatever has not gone out yet
    ctx->rays = rays_before + stats.rays;
    send_progress(ctx, 1.0f, "trace", pass.task_id);
    send_progress(ctx, 0.0f, "output", pass.task_id);
    *result_json_out = dup_string(result_json(0, "", ctx->last_image.data(), pass.width, pass.height), len);
    send_progress(ctx, 1.0f, "output", pass.task_id);
    if (!*result_json_out) return fail(RT_ERR_NOMEM, "out of host memory");
    return RT_OK;
    RTB_GUARD_END
}

rt_status rtb_initialize_incremental_render(rtb_context *ctx, const char *pass_json,
                                            char **result_json_out, size_t *len) {
    RTB_GUARD_BEGIN
    if (!ctx) return fail(RT_ERR_INVALID, "context is NULL");
    if (result_json_out) *result_json_out = nullptr;
    ctx->inc_ready = false;
    ctx->inc_index = 0;
    rt_pass pass;
    rt_status st = parse_pass(pass_json, pass);
    if (st != RT_OK) return error_result(st, result_json_out, len);
    if (ctx->active_render_key != pass.render_key) {
        ctx->active_render_key = pass.render_key;
        ctx->rays = 0;
    }
    st = ensure_scene(ctx);
    if (st != RT_OK) return error_result(st, result_json_out, len);
    if (pass.width <= 0 || pass.height <= 0 || pass.camera.rays_per_pixel <= 0) {
        st = fail(RT_ERR_INVALID, "pass width/height/RaysPerPixel must be positive");
        return error_result(st, result_json_out, len);
    }
    const size_t pixels = (size_t)pass.width * (size_t)pass.height;
    if (ctx->d_inc_accum) (void)hipFree(ctx->d_inc_accum);
    if (ctx->d_inc_rgba) (void)hipFree(ctx->d_inc_rgba);
    ctx->d_inc_accum = nullptr;
    ctx->d_inc_rgba = nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->d_inc_accum), pixels * 12);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ctx->d_inc_rgba), pixels * 4);
    if (e == hipSuccess) e = hipMemset(ctx->d_inc_accum, 0, pixels * 12);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        st = fail(RT_ERR_DEVICE, std::string("incremental buffers: ") + hipGetErrorString(e));
        return error_result(st, result_json_out, len);
    }
    ctx->inc_pass = pass;
    ctx->inc_ready = true;
    return RT_OK;
    RTB_GUARD_END
}

rt_status rtb_incremental_render(rtb_context *ctx, char **result_json_out, size_t *len) {
    RTB_GUARD_BEGIN
    if (!ctx || !result_json_out) return fail(RT_ERR_INVALID, "NULL argument");
    *result_json_out = nullptr;
    if (!ctx->inc_ready) {
        rt_status st = fail(RT_ERR_STATE, "incrementalRender before initializeIncrementalRender");
        return error_result(st, result_json_out, len);
    }
    const rt_pass &pass = ctx->inc_pass;
    if (ctx->inc_index == 0) send_progress(ctx, 0.0f, "trace", pass.task_id);
    rt_stats stats;
    rt_status st = rt::render_accumulate(*ctx->scene, &pass, 1, ctx->mode, ctx->inc_index, 1, true, false,
                                         ctx->d_inc_accum, ctx->d_inc_rgba, &stats, nullptr);
    if (st != RT_OK) return error_result(st, result_json_out, len);
    const size_t pixels = (size_t)pass.width * (size_t)pass.height;
    ctx->last_image.assign(pixels * 4, 0);
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipMemcpy(ctx->last_image.data(), ctx->d_inc_rgba, pixels * 4, hipMemcpyDeviceToHost);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        st = fail(RT_ERR_DEVICE, std::string("copy-back failed: ") + hipGetErrorString(e));
        return error_result(st, result_json_out, len);
    }
    ctx->last_w = pass.width;
    ctx->last_h = pass.height;
    ctx->rays += stats.rays;
    ctx->inc_index++;
    if (ctx->inc_index == pass.camera.rays_per_pixel) send_progress(ctx, 1.0f, "trace", pass.task_id);
    *result_json_out = dup_string(result_json(0, "", ctx->last_image.data(), pass.width, pass.height), len);
    if (!*result_json_out) return fail(RT_ERR_NOMEM, "out of host memory");
    return RT_OK;
    RTB_GUARD_END
}

uint64_t rtb_rays(const rtb_context *ctx) { return ctx ? ctx->rays : 0; }
uint32_t rtb_progress_deliveries(const rtb_context *ctx) { return ctx ? ctx->progress_deliveries : 0; }
int32_t rtb_triangle_count(const rtb_context *ctx) { return (ctx && ctx->initialized) ? ctx->mesh.n_tris : 0; }

size_t rtb_last_image(const rtb_context *ctx, const uint8_t **pixels, int32_t *width, int32_t *height) {
    if (!ctx || ctx->last_image.empty()) return 0;
    if (pixels) *pixels = ctx->last_image.data();
    if (width) *width = ctx->last_w;
    if (height) *height = ctx->last_h;
    return ctx->last_image.size();
}

}  // extern "C"
