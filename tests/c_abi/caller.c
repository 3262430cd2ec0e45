/* A plain C99 caller of the library, the way a cgo binding sees it: both
 * headers must compile as C, the opaque handles and PODs must be usable
 * without anything from C++, and failures come back as statuses.  No GPU is
 * touched: only the host-side entry points are called.  Prints the struct
 * sizes for the Python side to compare with its ctypes mirror. */
#include <stdio.h>
#include <string.h>

#include "rt_amd.h"
#include "rt_backend.h"

int main(void) {
    static const float verts[2 * 9] = {0, 0, 0, 1, 0, 0, 0, 1, 0,     /* z = 0 */
                                       0, 0, 1, 1, 0, 1, 0, 1, 1};    /* z = 1 */
    static const float kd[3] = {0.5f, 0.5f, 0.5f};
    static const uint8_t is_light[1] = {0};
    rt_mesh mesh;
    rt_bvh *bvh = NULL;
    char *json = NULL;
    size_t len = 0;
    rtb_context *ctx = NULL;

    memset(&mesh, 0, sizeof mesh);
    mesh.verts = verts;
    mesh.n_tris = 2;
    mesh.kd = kd;
    mesh.is_light = is_light;
    mesh.n_materials = 1;

    if (strstr(rt_version(), "gfx950") == NULL) return 1;
    if (rt_bvh_build(&mesh, 1, 1, 8, &bvh) != RT_OK) return 2;
    if (rt_bvh_node_count(bvh) < 1 || rt_bvh_tri_count(bvh) != 2) return 3;
    if (rt_bvh_to_json(bvh, &json, &len) != RT_OK || json == NULL || len != strlen(json)) return 4;
    if (strstr(json, "\"Root\"") == NULL || strstr(json, "\"SplitPlane\"") == NULL) return 5;
    rt_free(json);
    rt_bvh_destroy(bvh);

    /* errors are statuses with a message, never an abort */
    bvh = NULL;
    if (rt_bvh_from_json(&mesh, "{\"Root\":", 8, &bvh) != RT_ERR_PARSE || bvh != NULL) return 6;
    if (strlen(rt_last_error()) == 0) return 7;
    mesh.n_materials = 0;
    if (rt_bvh_build(&mesh, 1, 1, 8, &bvh) != RT_ERR_INVALID) return 8;

    /* the string-typed entry points: call order is checked before any device work */
    if (rtb_create(0, &ctx) != RT_OK || ctx == NULL) return 9;
    if (rtb_build_bvh(ctx, &json, &len) != RT_ERR_STATE) return 10;
    if (rtb_triangle_count(ctx) != 0 || rtb_rays(ctx) != 0) return 11;
    rtb_destroy(ctx);

    printf("rt_ray %u rt_hit %u rt_pass %u rt_camera %u rt_settings %u rt_light %u rt_stats %u rt_mesh %u\n",
           (unsigned)sizeof(rt_ray), (unsigned)sizeof(rt_hit), (unsigned)sizeof(rt_pass),
           (unsigned)sizeof(rt_camera), (unsigned)sizeof(rt_settings), (unsigned)sizeof(rt_light),
           (unsigned)sizeof(rt_stats), (unsigned)sizeof(rt_mesh));
    return 0;
}
