// OBJ/MTL reader.  Same statements and group rules as the reference's loader
// ("OpenGL Raytracer/OBJ.cpp":9-206) over a different mechanism: the file is read once into
// memory and walked with a token cursor, so malformed input ends in an error, never in a wild read.
#include "Mesh.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

bool read_file(const std::string &path, std::string &out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    std::fclose(f);
    return true;
}

// Walks whitespace-separated tokens; knows where lines end so that a statement can be skipped
// or bounded by its line.
struct Cursor {
    const char *p, *end;
    explicit Cursor(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}

    static bool blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
    void skip_blanks() { while (p < end && blank(*p)) p++; }
    void skip_space() { while (p < end && (blank(*p) || *p == '\n')) p++; }
    void skip_line() {
        while (p < end && *p != '\n') p++;
        if (p < end) p++;
    }
    bool at_line_end() {
        skip_blanks();
        return p >= end || *p == '\n';
    }
    // Next token anywhere ahead (crosses line ends, like operator>> on a stream).
    bool token(std::string &out) {
        skip_space();
        if (p >= end) return false;
        const char *b = p;
        while (p < end && !blank(*p) && *p != '\n') p++;
        out.assign(b, p - b);
        return true;
    }
    // Next token on the current line only.
    bool line_token(std::string &out) {
        if (at_line_end()) return false;
        return token(out);
    }
    bool number(float &out) {
        skip_space();
        if (p >= end) return false;
        char tmp[64];
        size_t n = 0;
        const char *q = p;
        while (q < end && !blank(*q) && *q != '\n' && n + 1 < sizeof tmp) tmp[n++] = *q++;
        tmp[n] = 0;
        char *e = nullptr;
        out = std::strtof(tmp, &e);
        if (e == tmp) return false;
        p += (e - tmp);
        return true;
    }
    bool unsigned_number(unsigned int &out) {
        skip_space();
        if (p >= end) return false;
        char *e = nullptr;
        char tmp[32];
        size_t n = 0;
        const char *q = p;
        while (q < end && !blank(*q) && *q != '\n' && n + 1 < sizeof tmp) tmp[n++] = *q++;
        tmp[n] = 0;
        unsigned long v = std::strtoul(tmp, &e, 10);
        if (e == tmp) return false;
        out = static_cast<unsigned int>(v);
        p += (e - tmp);
        return true;
    }
};

// One face corner: "v", "v/vt", "v//vn" or "v/vt/vn"; 0 = absent.  False on a malformed corner.
bool parse_corner(const std::string &tok, long idx[3]) {
    idx[0] = idx[1] = idx[2] = 0;
    const char *s = tok.c_str();
    for (int k = 0; k < 3; k++) {
        if (*s != '/' && *s != 0) {
            char *e = nullptr;
            idx[k] = std::strtol(s, &e, 10);
            if (e == s || idx[k] == 0) return false;
            s = e;
        }
        if (*s == 0) return idx[0] != 0;
        if (*s != '/') return false;
        s++;
    }
    return false;  // more than two slashes
}

// 1-based or negative (relative to the end) -> 0-based; -1 when out of range.
long resolve(long idx, size_t count) {
    if (idx > 0) return idx <= static_cast<long>(count) ? idx - 1 : -1;
    if (idx < 0) return -idx <= static_cast<long>(count) ? static_cast<long>(count) + idx : -1;
    return -1;
}

}  // namespace

bool Mesh::LoadMaterialData(std::string filepath, std::vector<MeshMaterialData> &data) {
    std::string text;
    if (!read_file(filepath, text)) {
        error = "Can't open file: " + filepath;
        return false;
    }
    Cursor c(text);
    std::string key;
    auto three = [&](float *dst) {
        for (int i = 0; i < 3; i++)
            if (!c.number(dst[i])) return;
    };
    while (c.token(key)) {
        if (key == "newmtl") {
            MeshMaterialData mtl;
            c.line_token(mtl.Name);
            data.push_back(mtl);
        } else if (data.empty()) {
            c.skip_line();  // a property before any newmtl has nothing to belong to
        } else if (key == "illum") {
            c.unsigned_number(data.back().Illum);
        } else if (key == "Kd") {
            three(data.back().Kd);
        } else if (key == "Ka") {
            three(data.back().Ka);
        } else if (key == "Tf") {
            three(data.back().Tf);
        } else if (key == "map_Kd") {
            c.line_token(data.back().map_Kd);
        } else if (key == "Ni") {
            c.number(data.back().Ni);
        } else {
            c.skip_line();  // Ns, Ks, Ke, d, comments, ...: the reference echoes and drops them
        }
    }
    return true;
}

bool Mesh::LoadFromObjFile(std::string dir, std::string filename) {
    groups.clear();
    error.clear();
    const std::string path = dir + filename;
    std::string text;
    if (!read_file(path, text)) {
        error = "Can't open file: " + path;
        return false;
    }

    std::vector<rth::vec3> points, normals;
    std::vector<rth::vec2> texCoords;
    std::vector<MeshMaterialData> materials;

    // The reference's group bookkeeping (OBJ.cpp:16-18, 61-64, 93-96, 103-106, 125-139): a `g`
    // opens a group and arms `rename_next`; only an f / mtllib / usemtl as the very next
    // statement disarms it; an armed `g` renames the current group instead of opening one.
    long statement = 0, last_group_statement = 0;
    bool rename_next = false;
    auto disarm = [&] {
        if (last_group_statement + 1 == statement) rename_next = false;
    };

    Cursor c(text);
    std::string key;
    long line_of_error = 0;
    auto fail = [&](const std::string &what) {
        line_of_error = 1;
        for (const char *q = text.data(); q < c.p; q++) line_of_error += (*q == '\n');
        error = path + ":" + std::to_string(line_of_error) + ": " + what;
        groups.clear();
        return false;
    };

    while (c.token(key)) {
        if (key == "v") {
            rth::vec3 u;
            if (!c.number(u.x) || !c.number(u.y) || !c.number(u.z)) return fail("v needs three numbers");
            points.push_back(u);
        } else if (key == "vt") {
            rth::vec2 u;
            if (!c.number(u.x) || !c.number(u.y)) return fail("vt needs two numbers");
            texCoords.push_back(u);
            c.skip_line();  // an optional third coordinate
        } else if (key == "vn") {
            rth::vec3 n;
            if (!c.number(n.x) || !c.number(n.y) || !c.number(n.z)) return fail("vn needs three numbers");
            normals.push_back(n);
        } else if (key == "f") {
            disarm();
            if (groups.empty()) {
                VertexGroup g;
                g.name = "default";
                groups.push_back(g);
            }
            std::vector<VertexData> corners;
            std::string tok;
            while (c.line_token(tok)) {
                long idx[3];
                if (!parse_corner(tok, idx)) return fail("bad face corner '" + tok + "'");
                VertexData v;
                const long pi = resolve(idx[0], points.size());
                if (pi < 0) return fail("vertex index out of range in '" + tok + "'");
                v.point = points[pi];
                if (idx[1] != 0) {
                    const long ti = resolve(idx[1], texCoords.size());
                    if (ti < 0) return fail("texture index out of range in '" + tok + "'");
                    v.texCoord = texCoords[ti];
                }
                if (idx[2] != 0) {
                    const long ni = resolve(idx[2], normals.size());
                    if (ni < 0) return fail("normal index out of range in '" + tok + "'");
                    v.normal = normals[ni];
                }
                corners.push_back(v);
            }
            if (corners.size() < 3) return fail("a face needs at least three corners");
            std::vector<VertexData> &dst = groups.back().vertices;
            for (size_t k = 1; k + 1 < corners.size(); k++) {  // fan; one triangle for three corners
                dst.push_back(corners[0]);
                dst.push_back(corners[k]);
                dst.push_back(corners[k + 1]);
            }
        } else if (key == "mtllib") {
            disarm();
            std::string s;
            if (c.line_token(s)) {
                std::string keep = error;
                if (!LoadMaterialData(dir + s, materials)) std::fprintf(stderr, "%s\n", error.c_str());
                error = keep;  // a missing MTL is not fatal (OBJ.cpp:153-157)
            }
        } else if (key == "usemtl") {
            disarm();
            std::string name;
            c.line_token(name);
            for (const MeshMaterialData &m : materials) {
                if (m.Name != name) continue;
                if (!groups.empty()) {
                    groups.back().materialData = m;
                } else {
                    VertexGroup g;
                    g.name = "default";
                    g.materialData = m;
                    groups.push_back(g);
                }
                break;
            }
        } else if (key == "g") {
            last_group_statement = statement;
            std::string name;
            c.line_token(name);
            c.skip_line();  // further group names on the line are ignored
            if (rename_next && !groups.empty()) {
                groups.back().name = name;
            } else {
                VertexGroup g;
                g.name = name;
                groups.push_back(g);
                rename_next = true;
            }
        } else {
            c.skip_line();  // #, s, o, and anything else
        }
        statement++;
    }
    return true;
}
