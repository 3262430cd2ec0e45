/*
 * rt_backend.h -- the reference backend's six entry points with their
 * JSON-string arguments and results, on top of rt_amd.h.
 *
 * The reference registers initialize / buildBVH / loadBVH / render /
 * initializeIncrementalRender / incrementalRender on the JS global object
 * (src/backend/main.go:28-33) and keeps its state in package globals
 * (main.go:20-21,190-194), one wasm instance per worker.  Here that state
 * lives in an rtb_context handle (one per worker / per GPU); arguments and
 * results are the same JSON documents, so a caller written against the
 * reference only changes how it reaches the functions.
 *
 * Every function returns an rt_status.  The render calls ALSO report failures
 * the way the reference does: the result document then carries exitCode -1,
 * the message and a blank 500x500 image (main.go:308-325), and the status is
 * the matching negative code.  Result strings are malloc'd; free with rt_free.
 */
#ifndef RT_BACKEND_H
#define RT_BACKEND_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtb_context rtb_context;

/* utility.ProgressUpdate (utility/utility.go:13-32): called with the JSON
 * {"progress":..,"event":"..","taskId":..,"rays":..} the reference hands to
 * the JS function progressUpdate. */
typedef void (*rtb_progress_fn)(const char *json, void *user);

/* One context per worker; `device` is the GPU its scene will live on. */
rt_status rtb_create(int32_t device, rtb_context **out);
void rtb_destroy(rtb_context *ctx);
void rtb_set_progress_callback(rtb_context *ctx, rtb_progress_fn fn, void *user);
/* Traversal mode used by the render calls (RT_MODE_*); default RT_MODE_STRICT_F32. */
rt_status rtb_set_mode(rtb_context *ctx, int32_t mode);

/* main.go:39-63 initialize(ctxJSON, ...textureBytes): RenderContext JSON with
 * Debug, UseBVH, BVHMaxLeafSize, BVHMaxDepth, Scene, ObjBuffer, MtlBuffer,
 * RawTextures[{Name}], WorkerId (models/context.go:11-38); textures[i] are the
 * PNG bytes of RawTextures[i]. */
rt_status rtb_initialize(rtb_context *ctx, const char *context_json, const uint8_t *const *textures,
                         const size_t *texture_sizes, int32_t n_textures);
/* main.go:65-76 buildBVH(): returns the BVH JSON; like the reference it does
 * NOT install the tree (loadBVH does). */
rt_status rtb_build_bvh(rtb_context *ctx, char **bvh_json, size_t *len);
/* main.go:78-90 loadBVH(bvhJSON) */
rt_status rtb_load_bvh(rtb_context *ctx, const char *bvh_json, size_t len);
/* main.go:93-188 render(passJSON) -> {"exitCode","message","imageData":{"Pix","Stride","Rect"}} */
rt_status rtb_render(rtb_context *ctx, const char *pass_json, char **result_json, size_t *len);
/* main.go:196-231 initializeIncrementalRender(passJSON): *result_json is NULL
 * on success (the reference returns nil) or the error document. */
rt_status rtb_initialize_incremental_render(rtb_context *ctx, const char *pass_json,
                                            char **result_json, size_t *len);
/* main.go:234-306 incrementalRender(): one more sample per pixel, running mean */
rt_status rtb_incremental_render(rtb_context *ctx, char **result_json, size_t *len);

/* context.Rays (process/trace.go:122; reset when RenderKey changes, main.go:107-110) */
uint64_t rtb_rays(const rtb_context *ctx);
/* At how many separate moments the last render() call sent its intermediate "trace"
 * progress updates (utility.go:13-32, main.go:120-154): 1 when they all went out
 * after the tile was done (tiles below 2^22 camera samples), more when a large tile
 * was traced range by range and reported while it rendered; 0 without a callback. */
uint32_t rtb_progress_deliveries(const rtb_context *ctx);
/* number of triangles after initialize (0 before) */
int32_t rtb_triangle_count(const rtb_context *ctx);
/* Raw RGBA8 bytes of the last render / incrementalRender result (width*height*4),
 * for callers that do not want to decode base64; returns the byte count. */
size_t rtb_last_image(const rtb_context *ctx, const uint8_t **pixels, int32_t *width, int32_t *height);

#ifdef __cplusplus
}
#endif
#endif
