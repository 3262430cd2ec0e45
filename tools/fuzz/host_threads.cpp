// host_threads.cpp -- the library's multi-threaded host paths under
// ThreadSanitizer (tools/fuzz/run_tsan.sh): the SAH builder (presort on several
// threads, subtrees forked while threads are free), the JSON writer and the JSON
// reader (nodes of one depth re-sorted on several threads), and two threads that
// build and load trees of their own at the same time.  No GPU needed.
//
// Besides "no data race reported" it checks that the threaded build is the same
// tree every time: the JSON text of repeated builds must be identical.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"

static uint64_t mix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static float unit(uint64_t &s) { return (float)(mix(s) >> 40) * 0x1p-24f; }

static std::vector<float> soup(int n, uint64_t seed) {
    std::vector<float> v((size_t)n * 9);
    for (int i = 0; i < n; ++i) {
        const float c[3] = {unit(seed) * 2 - 1, unit(seed) * 2 - 1, unit(seed) * 2 - 1};
        for (int j = 0; j < 9; ++j) v[(size_t)i * 9 + j] = c[j % 3] + (unit(seed) - 0.5f) * 0.04f;
    }
    return v;
}

static std::string build_text(const rt_mesh &mesh, int leaf, int depth, bool *ok) {
    rt_bvh *t = nullptr;
    if (rt_bvh_build(&mesh, 1, leaf, depth, &t) != RT_OK) { *ok = false; return ""; }
    char *text = nullptr;
    size_t len = 0;
    if (rt_bvh_to_json(t, &text, &len) != RT_OK) { *ok = false; rt_bvh_destroy(t); return ""; }
    std::string s(text, len);
    rt_free(text);
    // and back through the reader, which re-sorts the triangles per node on several threads
    rt_bvh *u = nullptr;
    if (rt_bvh_from_json(&mesh, s.c_str(), s.size(), &u) != RT_OK) *ok = false;
    else {
        if (rt_bvh_node_count(u) != rt_bvh_node_count(t)) *ok = false;
        rt_bvh_destroy(u);
    }
    rt_bvh_destroy(t);
    return s;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 150000;
    const std::vector<float> va = soup(n, 1), vb = soup(n / 2, 2);
    const float kd[3] = {0.5f, 0.5f, 0.5f};
    const uint8_t is_light[1] = {0};
    rt_mesh a{}, b{};
    a.verts = va.data(); a.n_tris = n; a.kd = kd; a.is_light = is_light; a.n_materials = 1;
    b = a; b.verts = vb.data(); b.n_tris = n / 2;
    bool ok = true;
    const std::string first = build_text(a, 6, 32, &ok);
    for (int rep = 0; rep < 2 && ok; ++rep)
        if (build_text(a, 6, 32, &ok) != first) { std::fprintf(stderr, "the threaded build is not reproducible\n"); return 1; }
    // two callers at once, each with its own mesh and tree
    std::string ta, tb;
    bool oka = true, okb = true;
    std::thread x([&] { ta = build_text(a, 6, 32, &oka); });
    std::thread y([&] { tb = build_text(b, 8, 40, &okb); });
    x.join();
    y.join();
    if (!ok || !oka || !okb || ta != first) { std::fprintf(stderr, "a build or a load failed, or the concurrent build differs\n"); return 1; }
    std::printf("{\"triangles\": %d, \"json_bytes\": %zu, \"builds\": 5, \"reproducible\": true}\n", n, first.size());
    return 0;
}
