// plymesh.cpp -- Shape "plymesh": read a PLY file into the arrays CreatePLYMesh hands to
// CreateTriangleMesh (src/shapes/plymesh.cpp:149-283). The reference reads PLY through the
// rply library; this is an independent reader of the PLY 1.0 format (ascii,
// binary_little_endian, binary_big_endian) that takes the same things from a file:
//   vertex  x y z (required), nx ny nz, and the first complete pair of
//           (u,v) (s,t) (texture_u,texture_v) (texture_s,texture_t)      plymesh.cpp:205-243
//   face    vertex_indices lists of 3 or 4; a quad (a,b,c,d) becomes (a,b,c),(d,a,c);
//           other lengths are skipped with a warning; an index outside
//           [0, vertexCount) is an error and the shape yields nothing       plymesh.cpp:104-147
// Values go through double like rply's callbacks ((float) / (int) of a double).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include "scene.h"

namespace mipt {
namespace {

enum PlyType { T_NONE, T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64 };
PlyType ParseType(const std::string &t) {
    if (t == "char" || t == "int8") return T_I8;
    if (t == "uchar" || t == "uint8") return T_U8;
    if (t == "short" || t == "int16") return T_I16;
    if (t == "ushort" || t == "uint16") return T_U16;
    if (t == "int" || t == "int32") return T_I32;
    if (t == "uint" || t == "uint32") return T_U32;
    if (t == "float" || t == "float32") return T_F32;
    if (t == "double" || t == "float64") return T_F64;
    return T_NONE;
}
int TypeSize(PlyType t) {
    switch (t) {
    case T_I8: case T_U8: return 1;
    case T_I16: case T_U16: return 2;
    case T_I32: case T_U32: case T_F32: return 4;
    case T_F64: return 8;
    default: return 0;
    }
}
struct PlyProperty {
    std::string name;
    bool isList = false;
    PlyType type = T_NONE, countType = T_NONE;
};
struct PlyElement {
    std::string name;
    long count = 0;
    std::vector<PlyProperty> props;
};

struct Reader {
    std::istream &in;
    int format;  // 0 ascii, 1 little endian, 2 big endian
    bool ok = true;
    double Next(PlyType t) {
        if (format == 0) {
            double v = 0;
            if (!(in >> v)) ok = false;
            return v;
        }
        unsigned char b[8] = {0};
        const int n = TypeSize(t);
        in.read((char *)b, n);
        if (in.gcount() != n) { ok = false; return 0; }
        if (format == 2) for (int i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);  // to little endian (host)
        switch (t) {
        case T_I8: { int8_t v; memcpy(&v, b, 1); return v; }
        case T_U8: { uint8_t v; memcpy(&v, b, 1); return v; }
        case T_I16: { int16_t v; memcpy(&v, b, 2); return v; }
        case T_U16: { uint16_t v; memcpy(&v, b, 2); return v; }
        case T_I32: { int32_t v; memcpy(&v, b, 4); return v; }
        case T_U32: { uint32_t v; memcpy(&v, b, 4); return v; }
        case T_F32: { float v; memcpy(&v, b, 4); return v; }
        case T_F64: { double v; memcpy(&v, b, 8); return v; }
        default: ok = false; return 0;
        }
    }
};

}  // namespace

bool ReadPLYMesh(const std::string &filename, PLYMeshData *out, std::vector<std::string> *warnings, std::string *err) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) { *err = "Couldn't open PLY file \"" + filename + "\""; return false; }
    std::string line;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") { *err = "Unable to read the header of PLY file \"" + filename + "\""; return false; }
    int format = -1;
    std::vector<PlyElement> elements;
    bool headerDone = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string kw;
        ls >> kw;
        if (kw == "format") {
            std::string f;
            ls >> f;
            format = (f == "ascii") ? 0 : (f == "binary_little_endian") ? 1 : (f == "binary_big_endian") ? 2 : -1;
        } else if (kw == "element") {
            PlyElement e;
            if (!(ls >> e.name >> e.count) || e.count < 0) { *err = "Unable to read the header of PLY file \"" + filename + "\""; return false; }
            // the arrays below are sized by the one vertex / face element: a second one would write past them
            for (const PlyElement &prev : elements)
                if (prev.name == e.name && (e.name == "vertex" || e.name == "face")) {
                    *err = filename + ": PLY file is invalid! More than one \"" + e.name + "\" element";
                    return false;
                }
            elements.push_back(e);
        } else if (kw == "property") {
            if (elements.empty()) { *err = "Unable to read the header of PLY file \"" + filename + "\""; return false; }
            PlyProperty p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it >> p.name;
                p.isList = true; p.countType = ParseType(ct); p.type = ParseType(it);
                if (p.countType == T_NONE) { *err = "Unable to read the header of PLY file \"" + filename + "\""; return false; }
            } else {
                p.type = ParseType(t);
                ls >> p.name;
            }
            if (p.type == T_NONE) { *err = "Unable to read the header of PLY file \"" + filename + "\""; return false; }
            elements.back().props.push_back(p);
        } else if (kw == "end_header") { headerDone = true; break; }
        // comment / obj_info lines are skipped
    }
    if (!headerDone || format < 0) { *err = "Unable to read the header of PLY file \"" + filename + "\""; return false; }

    long vertexCount = 0, faceCount = 0;
    for (const PlyElement &e : elements) {
        if (e.name == "vertex") vertexCount = e.count;
        else if (e.name == "face") faceCount = e.count;
    }
    if (vertexCount == 0 || faceCount == 0) { *err = filename + ": PLY file is invalid! No face/vertex elements found!"; return false; }
    {   // an element cannot hold more items than the file has bytes left (every item takes at least one: a binary
        // scalar, an ascii digit, or a list's count), so a count beyond that is a corrupt header, not an allocation size
        const std::streampos here = in.tellg();
        in.seekg(0, std::ios::end);
        const long long left = (long long)(in.tellg() - here);
        in.seekg(here);
        for (const PlyElement &e : elements)
            if (!e.props.empty() && (long long)e.count > left) {
                *err = filename + ": PLY file is invalid! Element \"" + e.name + "\" declares " + std::to_string(e.count) +
                       " items, the file holds " + std::to_string(left) + " bytes of data";
                return false;
            }
    }

    // which vertex properties feed which buffer (plymesh.cpp:189-243)
    const PlyElement *ve = nullptr;
    for (const PlyElement &e : elements) if (e.name == "vertex") ve = &e;
    auto has = [&](const char *n) { for (const PlyProperty &p : ve->props) if (p.name == n && !p.isList) return true; return false; };
    if (!(has("x") && has("y") && has("z"))) { *err = filename + ": Vertex coordinate property not found!"; return false; }
    const bool haveN = has("nx") && has("ny") && has("nz");
    const char *uName = nullptr, *vName = nullptr;
    const char *pairs[4][2] = {{"u", "v"}, {"s", "t"}, {"texture_u", "texture_v"}, {"texture_s", "texture_t"}};
    for (auto &pr : pairs) if (has(pr[0]) && has(pr[1])) { uName = pr[0]; vName = pr[1]; break; }
    out->P.assign((size_t)vertexCount, Vec3());
    if (haveN) out->N.assign((size_t)vertexCount, Vec3());
    if (uName) out->UV.assign((size_t)vertexCount, Vec2());
    out->indices.clear();
    out->indices.reserve((size_t)faceCount * 6);

    Reader rd{in, format};
    bool indexError = false;
    for (const PlyElement &e : elements) {
        const bool isVertex = e.name == "vertex", isFace = e.name == "face";
        for (long i = 0; i < e.count; ++i) {
            for (const PlyProperty &p : e.props) {
                if (!p.isList) {
                    const double v = rd.Next(p.type);
                    if (isVertex && (size_t)i < out->P.size()) {
                        const float f = (float)v;
                        if (p.name == "x") out->P[i].x = f; else if (p.name == "y") out->P[i].y = f; else if (p.name == "z") out->P[i].z = f;
                        else if (haveN && p.name == "nx") out->N[i].x = f; else if (haveN && p.name == "ny") out->N[i].y = f;
                        else if (haveN && p.name == "nz") out->N[i].z = f;
                        else if (uName && p.name == uName) out->UV[i].x = f; else if (vName && p.name == vName) out->UV[i].y = f;
                    }
                } else {
                    const long length = (long)rd.Next(p.countType);
                    const bool take = isFace && p.name == "vertex_indices";
                    int face[4] = {0, 0, 0, 0};
                    if (take && length != 3 && length != 4)
                        warnings->push_back("plymesh: Ignoring face with " + std::to_string(length) +
                                            " vertices (only triangles and quads are supported!)");
                    for (long k = 0; k < length && rd.ok; ++k) {
                        const double v = rd.Next(p.type);
                        if (take && (length == 3 || length == 4)) {
                            const int value = (int)v;
                            if (value < 0 || value >= vertexCount) {
                                if (!indexError)
                                    *err = "plymesh: Vertex reference " + std::to_string(value) + " is out of bounds! Valid range is [0.." +
                                           std::to_string(vertexCount) + ")";
                                indexError = true;
                            }
                            face[k] = value;
                        }
                    }
                    if (take && (length == 3 || length == 4)) {
                        for (int k = 0; k < 3; ++k) out->indices.push_back(face[k]);
                        if (length == 4) { out->indices.push_back(face[3]); out->indices.push_back(face[0]); out->indices.push_back(face[2]); }
                    }
                }
                if (!rd.ok) { *err = filename + ": unable to read the contents of PLY file"; return false; }
            }
        }
    }
    if (indexError) return false;
    return true;
}

}  // namespace mipt
