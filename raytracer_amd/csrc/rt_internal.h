// rt_internal.h -- host-side data model shared by the builder, the flattener
// and the C ABI.  Not part of the public interface.
#pragma once

#include <cstdint>
#include <exception>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"

namespace rt {

constexpr float kF32Max = 3.40282346638528859811704183484516925440e+38f;

// Sets the calling thread's error text and returns `code` (for `return fail(...)`).
rt_status fail(rt_status code, const std::string &msg);

// Runs the tasks at the same time, the first on the calling thread and the
// others on threads of their own, and returns when all have finished; a task's
// exception (out of memory, typically) is rethrown here instead of ending the
// process from inside its thread.
inline void run_together(const std::vector<std::function<void()>> &tasks) {
    std::vector<std::exception_ptr> errors(tasks.size());
    auto guarded = [&](size_t i) {
        try {
            tasks[i]();
        } catch (...) {
            errors[i] = std::current_exception();
        }
    };
    std::vector<std::thread> threads;
    for (size_t i = 1; i < tasks.size(); ++i) {
        try {
            threads.emplace_back(guarded, i);
        } catch (...) {          // no thread to be had: run it here
            guarded(i);
        }
    }
    if (!tasks.empty()) guarded(0);
    for (std::thread &t : threads) t.join();
    for (const std::exception_ptr &e : errors)
        if (e) std::rethrow_exception(e);
}

// Host copy of the BVH in preorder (BVHNode.Index order, bvh.go:59-67).
struct Bvh {
    int32_t n_tris = 0;
    int32_t n_nodes = 0;
    int32_t max_depth = 0;
    std::vector<int32_t> is_leaf, start, end, depth;
    std::vector<float> bmin, bmax, split;   // 3, 3, 4 floats per node
    std::vector<int32_t> order;             // original triangle id per sorted position
};

// Validates a mesh description (pointers, counts, finite vertices).
rt_status check_mesh(const rt_mesh *mesh);

// bvh.go:31-114,146-212
rt_status build_bvh(const rt_mesh &mesh, bool use_bvh, int32_t max_leaf, int32_t max_depth,
                    Bvh &out);

// Flat records as the kernels read them (layout: rt_core.h, DevScene).
struct FlatScene {
    std::vector<float> nodes;      // n_nodes * 8 floats (2 x 16-byte words)
    std::vector<float> tris;       // n_tris * 12 floats (3 x 16-byte words)
    std::vector<int32_t> tri_mat;  // material id | light bit
    std::vector<float> tri_uv;     // n_tris * 6 or empty
    std::vector<float> wide;       // ordered-traversal records, 16 floats each (may be empty)
    uint32_t fast_root = 0;
    bool fast_ok = false;
    // the region ray origins must lie in for the near-origin form of the ordered
    // traversal: the scene's box grown by rtc::kNearOriginDiagonals diagonals
    float near_lo[3] = {0, 0, 0}, near_hi[3] = {0, 0, 0};
};
rt_status flatten(const rt_mesh &mesh, const Bvh &bvh, FlatScene &out);
const std::string &last_error_text();

// Per-triangle precompute of models/triangle.go:45-65 in sorted-array order.
struct TriRecord {
    float v0[3], e0[3], v0v2[3], normal[3];
};
void make_tri_record(const float v[9], TriRecord &out);

// context.go:172-226 over the triangles in input order (the reference runs it
// before any BVH sort).
void find_light(const rt_mesh &mesh, rt_light &out);

// camera.go:43-85
void camera_setup(const rt_camera &cam, int32_t total_w, int32_t total_h, float top_left[3],
                  float steps[2]);

// context.go:245-275 (RenderPass.Initialize) + camera.go:43-85: everything a
// pass needs on the device, prepared on the host.
}  // namespace rt
namespace rtc { struct PassParams; }
namespace rt {
rt_status prepare_pass(const rt_light &scene_light, const rt_pass &pass, rtc::PassParams &pp);

// Scrambled Halton (bases 2 and 3) points in [0,1)^2.
void default_sample_table(int32_t n, uint64_t seed, double *out);

}  // namespace rt

// Opaque handle of the public header.
struct rt_bvh {
    rt::Bvh bvh;
};
