// host_fuzz.cpp -- mutation fuzzing of everything the library parses from
// caller-supplied bytes, for a host build with AddressSanitizer and
// UndefinedBehaviorSanitizer (tools/fuzz/run.sh).  Needs no GPU: the entry points
// below do their parsing on the host, and the ones that would go on to render
// stop with RT_ERR_DEVICE, which counts as a clean answer here.
//
// The contract under test: whatever the bytes, a call returns a status (and the
// render-level calls an error document, main.go:308-325) -- no crash, no read or
// write outside a buffer, no undefined arithmetic.
//
//   inputs mutated                    entry point
//   RenderContext JSON + OBJ + MTL    rtb_initialize        (context.go:54-234)
//   BVH JSON                          rtb_load_bvh, rt_bvh_from_json  (main.go:78-90)
//   binary BVH cache file             rt_bvh_load
//   PNG bytes                         rt_png_decode, rtb_initialize textures
//   RenderPass JSON                   rtb_render            (main.go:93-188, 336-344)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_backend.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() {
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static std::string json_escape(const std::string &s) {
    std::string o;
    for (char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if (c == '\n') o += "\\n";
        else if (c == '\r') o += "\\r";
        else if (c == '\t') o += "\\t";
        else o += c;
    }
    return o;
}

// A few byte-level mutations; numbers and structure characters of the text formats
// are favoured so that a mutant often still parses part of the way.
static std::string mutate(const std::string &in) {
    std::string s = in;
    const int n_mut = 1 + (int)below(4);
    static const char *tokens[] = {"0", "-1", "1e39", "-1e-46", "nan", "2147483648", "-2147483649", "99999999999999999999",
                                   "{", "}", "[", "]", ",", ":", "\"", "\\", "\\u0000", "\\ud800", "null", "true",
                                   "f 1 2 3\n", "f 1/1/1 2/2/2 3/3/3 4/4/4\n", "f -1 -2 -3\n", "f 0 0 0\n", "v 1 2\n",
                                   "usemtl Light\n", "newmtl \n", "map_Kd x.png\n", "Kd 1\n", "g\n", "o\n", "vt 1e40 0\n"};
    for (int m = 0; m < n_mut; ++m) {
        if (s.empty()) { s = tokens[below(sizeof tokens / sizeof *tokens)]; continue; }
        switch (below(7)) {
        case 0: s[below(s.size())] = (char)below(256); break;                         // byte flip
        case 1: s.resize(below(s.size() + 1)); break;                                 // truncate
        case 2: s.erase(below(s.size()), 1 + below(16)); break;                       // cut
        case 3: s.insert(below(s.size() + 1), tokens[below(sizeof tokens / sizeof *tokens)]); break;
        case 4: {                                                                     // duplicate a slice
            const size_t a = below(s.size()), len = 1 + below(64);
            s.insert(below(s.size() + 1), s.substr(a, len));
            break;
        }
        case 5: {                                                                     // overwrite a digit run
            size_t a = below(s.size());
            while (a < s.size() && !(s[a] >= '0' && s[a] <= '9')) ++a;
            size_t b = a;
            while (b < s.size() && ((s[b] >= '0' && s[b] <= '9') || s[b] == '.')) ++b;
            if (b > a) s.replace(a, b - a, tokens[below(8)]);
            break;
        }
        default: s[below(s.size())] ^= (char)(1u << below(8)); break;                // bit flip
        }
    }
    return s;
}

static void free_doc(char *p) { if (p) rt_free(p); }

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: host_fuzz <dir with cornell-box.obj/.mtl> <iterations> [seed]\n");
        return 2;
    }
    const std::string dir = argv[1];
    const long iters = std::atol(argv[2]);
    if (argc > 3) g_state = std::strtoull(argv[3], nullptr, 0);
    const std::string obj = slurp(dir + "/cornell-box.obj"), mtl = slurp(dir + "/cornell-box.mtl");
    if (obj.empty() || mtl.empty()) { std::fprintf(stderr, "scene files not found\n"); return 2; }

    auto context_json = [&](const std::string &o, const std::string &m, const std::string &tail) {
        return std::string("{\"Debug\":false,\"UseBVH\":true,\"BVHMaxLeafSize\":6,\"BVHMaxDepth\":16,\"Scene\":{},"
                           "\"ObjBuffer\":\"") + json_escape(o) + "\",\"MtlBuffer\":\"" + json_escape(m) + "\"," + tail + "\"WorkerId\":0}";
    };
    const std::string good_ctx = context_json(obj, mtl, "\"RawTextures\":[],");
    const std::string pass_json =
        "{\"Camera\":{\"Transform\":[1,0,0,0,0,1,0,0,0,0,1,0,-0.225,2.55,6,1],\"ProjectionPlaneDistance\":1,"
        "\"RaysPerPixel\":1,\"Projection\":0,\"FieldOfView\":45,\"OrtographicSize\":1},\"TotalWidth\":16,\"TotalHeight\":16,"
        "\"TaskID\":0,\"RNGSeed\":1,\"XOffset\":0,\"YOffset\":0,\"Width\":16,\"Height\":16,"
        "\"Settings\":{\"BounceLimit\":1,\"LightSampleRays\":1,\"GammaCorrection\":true,\"Gamma\":2.2,"
        "\"LightIntensity\":100,\"DebugLightSize\":1,\"DebugLightTransform\":[1,0,0,0,0,1,0,0,0,0,1,0,0,0,0,1]},\"RenderKey\":\"k\"}";

    // a good tree and its two serialised forms, from the library itself
    rtb_context *ctx = nullptr;
    if (rtb_create(0, &ctx) != RT_OK) { std::fprintf(stderr, "rtb_create failed\n"); return 2; }
    if (rtb_initialize(ctx, good_ctx.c_str(), nullptr, nullptr, 0) != RT_OK) {
        std::fprintf(stderr, "good context rejected: %s\n", rt_last_error());
        return 2;
    }
    char *bvh_text = nullptr;
    size_t bvh_len = 0;
    if (rtb_build_bvh(ctx, &bvh_text, &bvh_len) != RT_OK) { std::fprintf(stderr, "buildBVH failed\n"); return 2; }
    const std::string good_bvh(bvh_text, bvh_len);
    free_doc(bvh_text);

    // a small mesh for the array-level readers: the unit quad as two triangles, eight times
    std::vector<float> verts;
    for (int k = 0; k < 8; ++k) {
        const float z = (float)k;
        const float q[18] = {0, 0, z, 1, 0, z, 1, 1, z, 0, 0, z, 1, 1, z, 0, 1, z};
        verts.insert(verts.end(), q, q + 18);
    }
    const float kd[3] = {0.5f, 0.5f, 0.5f};
    const uint8_t is_light[1] = {0};
    rt_mesh mesh{};
    mesh.verts = verts.data();
    mesh.n_tris = (int32_t)(verts.size() / 9);
    mesh.kd = kd;
    mesh.is_light = is_light;
    mesh.n_materials = 1;
    rt_bvh *tree = nullptr;
    if (rt_bvh_build(&mesh, 1, 2, 8, &tree) != RT_OK) { std::fprintf(stderr, "rt_bvh_build failed\n"); return 2; }
    char *tree_text = nullptr;
    size_t tree_len = 0;
    rt_bvh_to_json(tree, &tree_text, &tree_len);
    const std::string good_tree(tree_text ? tree_text : "", tree_len);
    free_doc(tree_text);
    const std::string cache_path = "/tmp/rt_host_fuzz.bvh", mutant_path = "/tmp/rt_host_fuzz_mutant.bvh";
    rt_bvh_save(tree, cache_path.c_str());
    const std::string good_cache = slurp(cache_path);
    rt_bvh_destroy(tree);

    // a 2x2 RGBA PNG, stored (uncompressed) deflate: signature, IHDR, IDAT, IEND -- written by
    // tools/fuzz/run.sh with Python's zlib beside the scene files
    const std::string good_png = slurp(dir + "/fuzz_seed.png");

    long clean_errors = 0, accepted = 0;
    for (long it = 0; it < iters; ++it) {
        switch (it % 8) {
        case 0: {   // context JSON as a whole
            const std::string m = mutate(good_ctx);
            (rtb_initialize(ctx, m.c_str(), nullptr, nullptr, 0) == RT_OK ? accepted : clean_errors)++;
            break;
        }
        case 1: {   // OBJ text inside a well-formed context
            const std::string m = context_json(mutate(obj), mtl, "\"RawTextures\":[],");
            (rtb_initialize(ctx, m.c_str(), nullptr, nullptr, 0) == RT_OK ? accepted : clean_errors)++;
            break;
        }
        case 2: {   // MTL text, and a texture that may or may not decode
            const std::string png = mutate(good_png);
            const uint8_t *tex[1] = {reinterpret_cast<const uint8_t *>(png.data())};
            const size_t sizes[1] = {png.size()};
            const std::string m = context_json(obj, mutate(mtl + "map_Kd fuzz.png\n"), "\"RawTextures\":[{\"Name\":\"fuzz.png\"}],");
            (rtb_initialize(ctx, m.c_str(), tex, sizes, 1) == RT_OK ? accepted : clean_errors)++;
            break;
        }
        case 3: {   // BVH JSON through loadBVH (after a good initialize, so that there are triangles to re-sort)
            rtb_initialize(ctx, good_ctx.c_str(), nullptr, nullptr, 0);
            const std::string m = mutate(good_bvh);
            (rtb_load_bvh(ctx, m.c_str(), m.size()) == RT_OK ? accepted : clean_errors)++;
            break;
        }
        case 4: {   // BVH JSON through the array-level reader
            const std::string m = mutate(good_tree);
            rt_bvh *t = nullptr;
            if (rt_bvh_from_json(&mesh, m.c_str(), m.size(), &t) == RT_OK) { accepted++; rt_bvh_destroy(t); } else clean_errors++;
            break;
        }
        case 5: {   // binary cache
            const std::string m = mutate(good_cache);
            { std::ofstream f(mutant_path, std::ios::binary); f.write(m.data(), (std::streamsize)m.size()); }
            rt_bvh *t = nullptr;
            if (rt_bvh_load(&mesh, mutant_path.c_str(), &t) == RT_OK) { accepted++; rt_bvh_destroy(t); } else clean_errors++;
            break;
        }
        case 6: {   // PNG decoder: size query, then the decode into exactly that much
            const std::string m = mutate(good_png);
            int32_t w = 0, h = 0;
            const uint8_t *d = reinterpret_cast<const uint8_t *>(m.data());
            if (rt_png_decode(d, m.size(), nullptr, 0, &w, &h) == RT_OK && w > 0 && h > 0 && (int64_t)w * h <= (1 << 22)) {
                std::vector<uint8_t> px((size_t)w * (size_t)h * 4);
                (rt_png_decode(d, m.size(), px.data(), px.size(), &w, &h) == RT_OK ? accepted : clean_errors)++;
            } else {
                clean_errors++;
            }
            break;
        }
        default: {  // pass JSON: parse errors come back as the reference's error document
            rtb_initialize(ctx, good_ctx.c_str(), nullptr, nullptr, 0);
            rtb_load_bvh(ctx, good_bvh.c_str(), good_bvh.size());
            const std::string m = mutate(pass_json);
            char *doc = nullptr;
            size_t len = 0;
            const rt_status st = rtb_render(ctx, m.c_str(), &doc, &len);
            if (st == RT_OK) accepted++; else clean_errors++;
            if (doc && len >= 2 && doc[0] != '{') { std::fprintf(stderr, "render answered with a non-document\n"); return 1; }
            free_doc(doc);
            break;
        }
        }
    }
    rtb_destroy(ctx);
    std::printf("{\"iterations\": %ld, \"accepted\": %ld, \"clean_errors\": %ld}\n", iters, accepted, clean_errors);
    return 0;
}
