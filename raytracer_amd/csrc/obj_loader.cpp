// obj_loader.cpp -- OBJ/MTL text -> triangle list, standing in for the
// github.com/udhos/gwob v0.0.0-20200524213453-619810f75817 parser the
// reference calls (models/context.go:61-76) plus the triangle loop over its
// groups (context.go:109-168).
//
// Behaviour kept from that pipeline: faces are triangles or quads (a quad
// becomes two triangles), indices may be negative (relative), a new group
// starts when `usemtl` names a different material or `g` a different name, the
// triangle index restarts at 0 in every group, materials are looked up by name
// and a missing one falls back to the pink "Debug" material (context.go:82-91,
// 111-114), texture coordinates are taken as written (no V flip).  The gwob
// sources are not available offline; the quad split order (0,1,2)+(2,3,0) is
// the documented behaviour as far as recalled and is not pinned by any
// reference fixture.
#include <cstdlib>
#include <cstring>
#include <map>

#include "backend.h"

namespace rt {
namespace {

struct Cursor {
    const char *p, *end;
    void skip_space() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    }
    bool at_eol() const { return p >= end || *p == '\n'; }
    void next_line() {
        while (p < end && *p != '\n') ++p;
        if (p < end) ++p;
    }
    std::string word() {
        skip_space();
        const char *s = p;
        while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
        return std::string(s, p);
    }
    std::string rest() {
        skip_space();
        const char *s = p;
        while (p < end && *p != '\n') ++p;
        const char *e = p;
        while (e > s && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
        return std::string(s, e);
    }
    bool number(float &out) {
        skip_space();
        if (at_eol()) return false;
        char *e = nullptr;
        // strtof stops at the first non-number character; the buffer ends in a
        // NUL because it comes from a std::string
        out = std::strtof(p, &e);
        if (e == p) return false;
        p = e;
        return true;
    }
};

struct Corner {
    int v = 0, t = 0;   // 1-based resolved indices, 0 = absent
};

bool parse_corner(const std::string &w, size_t n_v, size_t n_t, Corner &c) {
    const char *s = w.c_str();
    char *e = nullptr;
    long v = std::strtol(s, &e, 10);
    if (e == s) return false;
    long t = 0;
    if (*e == '/') {
        const char *s2 = e + 1;
        if (*s2 != '/' && *s2 != '\0') {
            t = std::strtol(s2, &e, 10);
            if (e == s2) return false;
        }
    }
    if (v < 0) v = (long)n_v + v + 1;
    if (t < 0) t = (long)n_t + t + 1;
    if (v < 1 || v > (long)n_v) return false;
    if (t < 0 || t > (long)n_t) return false;
    c.v = (int)v;
    c.t = (int)t;
    return true;
}

void parse_mtl(const std::string &mtl, std::vector<ObjMaterial> &out) {
    Cursor c{mtl.c_str(), mtl.c_str() + mtl.size()};
    while (c.p < c.end) {
        const std::string key = c.word();
        if (key == "newmtl") {
            ObjMaterial m;
            m.name = c.rest();
            out.push_back(m);
        } else if (!out.empty() && key == "Kd") {
            float v;
            for (int k = 0; k < 3; ++k)
                if (c.number(v)) out.back().kd[k] = v;
        } else if (!out.empty() && key == "map_Kd") {
            out.back().map_kd = c.rest();
        }
        c.next_line();
    }
}

}  // namespace

rt_status parse_obj(const std::string &obj, const std::string &mtl, ObjScene &out) {
    out = ObjScene();
    parse_mtl(mtl, out.materials);
    ObjMaterial debug;   // context.go:82-91
    debug.name = "Debug";
    debug.kd[0] = 1;
    debug.kd[1] = 0;
    debug.kd[2] = 1;
    out.materials.push_back(debug);
    const int32_t debug_id = (int32_t)out.materials.size() - 1;
    std::map<std::string, int32_t> by_name;
    for (int32_t i = 0; i < debug_id; ++i) by_name[out.materials[(size_t)i].name] = i;   // later duplicates win

    std::vector<float> pos, tex;
    std::string cur_mtl, cur_group;
    bool have_mtl = false, have_group = false;
    int32_t cur_mat = debug_id, tri_index = 0, line_no = 0;
    Cursor c{obj.c_str(), obj.c_str() + obj.size()};
    while (c.p < c.end) {
        ++line_no;
        const std::string key = c.word();
        if (key == "v") {
            float v[3];
            for (int k = 0; k < 3; ++k)
                if (!c.number(v[k]))
                    return fail(RT_ERR_PARSE, "OBJ line " + std::to_string(line_no) + ": bad vertex");
            pos.insert(pos.end(), v, v + 3);
        } else if (key == "vt") {
            float u = 0, w = 0;
            if (!c.number(u)) return fail(RT_ERR_PARSE, "OBJ line " + std::to_string(line_no) + ": bad texture coordinate");
            c.number(w);
            tex.push_back(u);
            tex.push_back(w);
            out.has_texcoords = true;
        } else if (key == "usemtl") {
            const std::string name = c.rest();
            if (!have_mtl || name != cur_mtl) {
                cur_mtl = name;
                have_mtl = true;
                auto it = by_name.find(name);
                cur_mat = it == by_name.end() ? debug_id : it->second;
                tri_index = 0;   // new group
            }
        } else if (key == "g") {
            const std::string name = c.rest();
            if (have_group && name != cur_group) tri_index = 0;
            cur_group = name;
            have_group = true;
        } else if (key == "f") {
            Corner cs[4];
            int n = 0;
            while (!c.at_eol()) {
                const std::string w = c.word();
                if (w.empty()) break;
                if (n == 4) return fail(RT_ERR_PARSE, "OBJ line " + std::to_string(line_no) + ": faces with more than 4 vertices are not supported");
                if (!parse_corner(w, pos.size() / 3, tex.size() / 2, cs[n]))
                    return fail(RT_ERR_PARSE, "OBJ line " + std::to_string(line_no) + ": bad face index '" + w + "'");
                ++n;
            }
            if (n < 3) return fail(RT_ERR_PARSE, "OBJ line " + std::to_string(line_no) + ": face with fewer than 3 vertices");
            const int order[2][3] = {{0, 1, 2}, {2, 3, 0}};
            for (int t = 0; t < (n == 4 ? 2 : 1); ++t) {
                for (int k = 0; k < 3; ++k) {
                    const Corner &cr = cs[order[t][k]];
                    const float *p = &pos[3 * (size_t)(cr.v - 1)];
                    out.verts.insert(out.verts.end(), p, p + 3);
                    if (cr.t > 0) {
                        out.texcoords.push_back(tex[2 * (size_t)(cr.t - 1)]);
                        out.texcoords.push_back(tex[2 * (size_t)(cr.t - 1) + 1]);
                    } else {
                        out.texcoords.push_back(0);
                        out.texcoords.push_back(0);
                    }
                }
                out.material.push_back(cur_mat);
                out.group_index.push_back(tri_index++);
            }
        }
        c.next_line();
    }
    if (out.material.empty()) return fail(RT_ERR_PARSE, "OBJ buffer holds no faces");
    return RT_OK;
}

std::string base64_encode(const uint8_t *data, size_t n) {
    static const char tab[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((n + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 2 < n; i += 3) {
        const uint32_t v = ((uint32_t)data[i] << 16) | ((uint32_t)data[i + 1] << 8) | data[i + 2];
        out.push_back(tab[(v >> 18) & 63]);
        out.push_back(tab[(v >> 12) & 63]);
        out.push_back(tab[(v >> 6) & 63]);
        out.push_back(tab[v & 63]);
    }
    if (i + 1 == n) {
        const uint32_t v = (uint32_t)data[i] << 16;
        out.push_back(tab[(v >> 18) & 63]);
        out.push_back(tab[(v >> 12) & 63]);
        out += "==";
    } else if (i + 2 == n) {
        const uint32_t v = ((uint32_t)data[i] << 16) | ((uint32_t)data[i + 1] << 8);
        out.push_back(tab[(v >> 18) & 63]);
        out.push_back(tab[(v >> 12) & 63]);
        out.push_back(tab[(v >> 6) & 63]);
        out.push_back('=');
    }
    return out;
}

}  // namespace rt
