// rt_device.h -- device-side scene object and the launch entry points the C
// ABI layer forwards to.  Internal.
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "rt_core.h"
#include "rt_internal.h"

namespace rt {

// Ring of ray-queue cursors, one per traversal launch, each in a 64-byte line of
// its own.  A slot is used again 1024 launches later; launches on one stream
// are ordered anyway, so only that many launches in flight at once on
// different streams could ever share one.
constexpr uint32_t kCursorSlots = 1024;
constexpr uint32_t kCursorStride = 16;  // in 32-bit words

// Per-path word, read and written once per path vertex as one 16-byte access.
//   stream_lo/hi  counter-RNG stream key of the camera sample
//   packed        bits 0-1 state, bits 2-10 number of shading terms (0..256),
//                 bits 11-31 RNG draws taken so far (at most 256 x 64 x 3)
//   bounce_q      slot of the path's current ray in the ray queue being traced
struct alignas(16) PathHead {
    uint32_t stream_lo, stream_hi, packed, bounce_q;
};
struct alignas(8) Pair32 {
    float x, y;
};

// Per-batch path state of the wavefront path tracer (all device pointers into
// ONE block of device memory, see ws_layout).
struct PathBuffers {
    rt_ray *rays[2];
    rt_hit *hits[2];
    uint32_t *counts;          // queue lengths: [0] primaries, [k+1] rays emitted at vertex k
    PathHead *head;            // P
    // L*P, [i*P + path]: what light sample i adds to the vertex's shading when
    // its shadow ray reaches the light (xyz), and that ray's slot in the queue
    // (w, as bits; -1: no ray was cast)
    rtc::f4 *slots;
    // (B+1)*P each, [k*P + path], written at vertex k: terms_a = (shading term of
    // vertex k-1, x of the BRDF term of vertex k), terms_b = its y and z.  The
    // BRDF term of a path's last vertex is zero in the reference whichever way the
    // path ends (trace.go:81-84, 92-96) and is not stored.
    rtc::f4 *terms_a;
    Pair32 *terms_b;
    float *radiance;           // 3 planes of P floats (r, g, b), each [sample][pixel]
    unsigned long long *ray_total;
    unsigned long long *trav_counters;   // 8 x u64, filled when the scene counts traversal work
    uint32_t *cursor;          // ray hand-out cursor of the traversal launches (fixed-block calls only)
    // rays cast so far by each path of the batch; only kept (non-null) for calls that
    // report the reference's progress events
    uint32_t *path_rays;
    unsigned long long *progress_bins;   // kProgressBins x u64
};

// The reference reports progress while it traces (main.go:120-154): an event
// whenever the running camera-sample index passes updateIndex + updateInterval,
// each with the ray counter's value at that moment.  A call that wants those
// values gets, per bin j, the rays cast by the camera samples whose index (in
// the reference's pixel-major order) lies in [j * step, (j + 1) * step), the
// last bin taking all the rest.
constexpr int kProgressBins = 16;
struct ProgressBins {
    uint32_t step = 1;                     // updateInterval + 1
    int32_t n_bins = 1;                    // bins in use (at most kProgressBins)
    unsigned long long rays[kProgressBins] = {};
    // chunks > 1: the tile is traced in that many pixel ranges, one after the other, and
    // after each but the last `on_range` is called with the number of bins that are
    // complete by then (`rays` holds their sums): the caller can report progress WHILE
    // the tile renders, as the reference does.  Results do not depend on the batch shape.
    int32_t chunks = 1;
    std::function<void(int32_t bins_complete)> on_range;
};

struct Scene {
    int device = 0;
    int num_cus = 256;
    int32_t n_tris = 0, n_nodes = 0, max_depth = 0, n_materials = 0;
    rt_light light{};
    // flattened buffers (HBM)
    rtc::f4 *d_nodes = nullptr;
    rtc::f4 *d_tris = nullptr;
    rtc::f4 *d_linked = nullptr;      // reference-order tree with explicit links, top levels first (may be null)
    int32_t n_top = 0;
    rtc::f4 *d_wide = nullptr;        // ordered-traversal records (fast mode), may be null
    uint32_t fast_root = 0;
    int32_t n_wide = 0;               // records in d_wide
    uint64_t wide_bytes = 0;
    float near_lo[3] = {0, 0, 0}, near_hi[3] = {0, 0, 0};   // origin region of the near-origin ordered form
    rtc::f4 *d_tri_shade = nullptr;   // per triangle (normal, material word)
    float *d_kd = nullptr;
    float *d_tri_uv = nullptr;
    int32_t *d_mat_tex = nullptr;
    uint8_t **d_tex_ptrs = nullptr;
    int32_t *d_tex_dims = nullptr;
    double *d_cam_tab = nullptr;
    double *d_light_tab = nullptr;
    int32_t n_cam = 0, n_light = 0;
    uint32_t *d_cursors = nullptr;
    std::atomic<uint32_t> cursor_next{0};
    uint64_t node_bytes = 0, tri_bytes = 0;
    std::vector<int32_t> mat_tex;
    std::vector<uint8_t *> tex_ptrs;
    std::vector<int32_t> tex_dims;
    // path-tracing workspace (grow-only)
    size_t max_paths_in_flight = (size_t)1 << 26;
    // launch tuning (defaults chosen by measurement; overridable for experiments)
    int tune_stack = 8;           // LDS stack entries per lane of the fast kernel
    int tune_refill = 0;          // refill idle lanes once fewer than this many are active (0 = per-kernel default)
    int tune_blocks_per_cu = 8;   // persistent grid size
    int tune_leaf_min = 0;        // leave the leaf phase below this many lanes in a leaf (0 = per-kernel default)
    int tune_descend_min = 0;     // leave the descent phase below this many descending lanes (0 = per-kernel default)
    bool count_traversal = false;   // render calls run the counting kernel variant
    size_t ws_paths = 0;
    int ws_L = 0, ws_B = 0;
    PathBuffers ws{};
    void *ws_block = nullptr;     // the one allocation the path buffers are carved from
    size_t ws_block_bytes = 0;
    size_t ws_byte_limit = 0;     // 0: none; a batch whose workspace would be larger is halved
    rtc::TileDesc *d_tiles = nullptr;   // tile table of a multi-tile call
    size_t tile_cap = 0;
    // The workspace is shared by every render call on this scene: calls take
    // the mutex on the host, and each call's stream first waits for the event
    // the previous call recorded at its end, whatever stream that was on.
    std::mutex render_mu;
    hipEvent_t ws_done = nullptr;
};

rt_status scene_create(const rt_mesh &mesh, const Bvh &bvh, int device, Scene **out);
rt_status scene_set_tables(Scene &sc, const double *cam, int32_t n_cam, const double *lig, int32_t n_lig);
rt_status scene_set_texture(Scene &sc, int32_t slot, const uint8_t *rgba, int32_t w, int32_t h);
void scene_destroy(Scene *sc);
// Sets the byte bound of the path workspace; a workspace already larger is released.
rt_status scene_set_workspace_limit(Scene &sc, size_t max_bytes);

rt_status launch_trace(Scene &sc, const rt_ray *d_rays, size_t n_host, const uint32_t *d_n,
                       rt_hit *d_hits, int mode, unsigned long long *d_counters,
                       hipStream_t stream, uint32_t *d_cursor = nullptr);
rt_status trace_rays_host(Scene &sc, const rt_ray *rays, size_t n, rt_hit *hits, int mode,
                          bool want_counters, rt_stats *stats);
rt_status render_accumulate(Scene &sc, const rt_pass *passes, int32_t n_passes, int mode,
                            int32_t sample_begin, int32_t sample_count, bool incremental,
                            bool zero_accum, float *d_accum, uint8_t *d_rgba, rt_stats *stats,
                            hipStream_t stream);
// One tile with the path workspace taken from the library's static device
// array instead of allocated memory; calls nothing that allocates.
rt_status render_tile_static(Scene &sc, const rt_pass &pass, int mode, float *d_accum,
                             uint8_t *d_rgba, hipStream_t stream);
// Bytes of path workspace one camera sample in flight takes at these settings.
size_t workspace_bytes_per_path(int light_samples, int bounce_limit);
rt_status render_tile_host(Scene &sc, const rt_pass &pass, int mode, float *accum_rgb,
                           uint8_t *rgba, rt_stats *stats, ProgressBins *progress = nullptr);
rt_status gather_probe(int device, size_t table_bytes, double *records_per_s);
rt_status resolve_device(const float *d_accum, size_t pixels, int32_t divisor, int gamma_on,
                         float gamma, uint8_t *d_rgba, hipStream_t stream);

}  // namespace rt

struct rt_scene {
    rt::Scene *impl = nullptr;
};
