/* A plain C99 caller of the library, the way a cgo binding sees it: both
 * headers must compile as C, the opaque handles and PODs must be usable
 * without anything from C++, and failures come back as statuses.  No GPU is
 * touched: only the host-side entry points are called.  Prints the struct
 * sizes for the Python side to compare with its ctypes mirror. */
#include <stdio.h>
#include <string.h>

#include "rt_amd.h"
#include "rt_backend.h"

/* Device part (argument "gpu"): the two-triangle scene, rays whose answers are
 * known by hand.  Triangle 0 lies in z = 0, triangle 1 in z = 1, both facing
 * +z with the right angle at (0, 0, z) (triangle.go:78-103: one-sided test,
 * u along v0->v1, v along v0->v2). */
static int device_part(const rt_mesh *mesh) {
    rt_bvh *bvh = NULL;
    rt_scene *scene = NULL;
    rt_ray rays[4];
    rt_hit hits[4];
    rt_stats stats;
    int mode;
    if (rt_bvh_build(mesh, 0, 6, 16, &bvh) != RT_OK) return 20;
    if (rt_scene_create(mesh, bvh, 0, &scene) != RT_OK) {
        fprintf(stderr, "%s\n", rt_last_error());
        return 21;
    }
    memset(rays, 0, sizeof rays);
    /* 0: straight down onto triangle 1 (z = 1) at (0.25, 0.5): t = 1, u = 0.25, v = 0.5 */
    rays[0].origin[0] = 0.25f; rays[0].origin[1] = 0.5f; rays[0].origin[2] = 2.0f;
    rays[0].dir[2] = -1.0f; rays[0].tmax = 3.4028235e38f;
    /* 1: the same ray starting between the triangles: reaches triangle 0 at t = 0.5 */
    rays[1] = rays[0]; rays[1].origin[2] = 0.5f;
    /* 2: from below, along +z: both triangles are back-facing, nothing is hit */
    rays[2] = rays[0]; rays[2].origin[2] = -1.0f; rays[2].dir[2] = 1.0f;
    /* 3: the first ray again, cut short before the triangle */
    rays[3] = rays[0]; rays[3].tmax = 0.75f;
    for (mode = RT_MODE_STRICT_F32; mode <= RT_MODE_FAST_F32; ++mode) {
        memset(hits, 0xff, sizeof hits);
        if (rt_trace_rays(scene, rays, 4, hits, mode, 0, &stats) != RT_OK) {
            fprintf(stderr, "%s\n", rt_last_error());
            return 22;
        }
        if (stats.rays != 4) return 23;
        if (hits[0].tri_pos != 1 || hits[0].t != 1.0f || hits[0].u != 0.25f || hits[0].v != 0.5f) return 24;
        if (hits[1].tri_pos != 0 || hits[1].t != 0.5f || hits[1].u != 0.25f || hits[1].v != 0.5f) return 25;
        if (hits[2].tri_pos != -1 || hits[2].t != 3.4028235e38f) return 26;
        if (hits[3].tri_pos != -1 || hits[3].t != 0.75f) return 27;
    }
    if (rt_trace_rays(scene, rays, 4, hits, 99, 0, NULL) != RT_ERR_INVALID) return 28;
    rt_scene_destroy(scene);
    rt_bvh_destroy(bvh);
    return 0;
}

int main(int argc, char **argv) {
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

    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        int rc = device_part(&mesh);
        if (rc != 0) return rc;
    }

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
