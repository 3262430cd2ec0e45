// backend.h -- host-side pieces behind the string-typed entry points:
// OBJ/MTL ingestion, PNG decoding, base64.  Internal.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "rt_internal.h"

namespace rt {

// What RenderContext.Initialize builds from the OBJ/MTL buffers
// (models/context.go:54-168): triangles in file order with their material and
// per-group running index.
struct ObjMaterial {
    std::string name;
    float kd[3] = {1, 1, 1};
    std::string map_kd;
};

struct ObjScene {
    std::vector<float> verts;         // n * 9
    std::vector<float> texcoords;     // n * 6 (zeros when the file has no vt)
    std::vector<int32_t> material;    // n
    std::vector<int32_t> group_index; // n: Triangle.Index
    std::vector<ObjMaterial> materials;   // the last entry is the "Debug" fallback
    bool has_texcoords = false;
    int32_t n_tris() const { return (int32_t)material.size(); }
};

rt_status parse_obj(const std::string &obj, const std::string &mtl, ObjScene &out);

// PNG -> 8-bit RGBA rows the way NewTexture ends up holding them
// (models/texture.go:24-46): opaque images as they are, images with alpha
// premultiplied like image/draw's NRGBA -> RGBA conversion.
rt_status decode_png(const uint8_t *data, size_t size, std::vector<uint8_t> &rgba, int32_t &width,
                     int32_t &height);

std::string base64_encode(const uint8_t *data, size_t n);

}  // namespace rt
