// cloud_io.hpp -- point-cloud file helpers shared by the command-line front-ends.
//
//   read_cloud():  ".ply" goes through the CPLYLoader mirror (ASCII, the reference's parse rules; include/PlyLoad.h),
//                  or through read_ply_binary() when the header says binary_little_endian (an extension: the reference
//                  loader only reads ASCII); anything else is read as the reference's text cloud format -- first the
//                  point count, then "x y z" per point (`loadPoints`, PS_AIS_Simplification/Main_KSS_List.cpp:65-94),
//                  which is what its .xyz results and the shipped data/registration/*.gird / *.wlop files are.
//   append_xyz():  the reference's writer (Main_KSS_ICP.cpp:49-59): count, points, blank line, APPEND mode.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "PlyLoad.h"

namespace kss_cli {

typedef std::vector<std::vector<double>> Cloud;

inline bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// binary_little_endian PLY: vertex element only, any mix of scalar properties; x, y, z may be float or double.
inline bool read_ply_binary(const std::string& path, Cloud& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line;
    if (!std::getline(f, line) || line.compare(0, 3, "ply") != 0) return false;
    bool little = false, in_vertex = false;
    long nvert = -1;
    struct Prop { std::string name; int size; bool is_double; };
    std::vector<Prop> props;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line == "end_header") break;
        char a[64] = {0}, b[64] = {0};
        long v = 0;
        if (std::sscanf(line.c_str(), "format %63s", a) == 1) little = std::strcmp(a, "binary_little_endian") == 0;
        else if (std::sscanf(line.c_str(), "element %63s %ld", a, &v) == 2) { in_vertex = std::strcmp(a, "vertex") == 0; if (in_vertex) nvert = v; }
        else if (in_vertex && std::sscanf(line.c_str(), "property %63s %63s", a, b) == 2) {
            if (std::strcmp(a, "list") == 0) return false;   // no lists inside the vertex element
            int sz = 0;
            const std::string t = a;
            if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") sz = 1;
            else if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") sz = 2;
            else if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") sz = 4;
            else if (t == "double" || t == "float64") sz = 8;
            else return false;
            props.push_back({b, sz, sz == 8});
        }
    }
    if (!little || nvert < 0) return false;
    int stride = 0, off[3] = {-1, -1, -1};
    bool dbl[3] = {false, false, false};
    for (const Prop& p : props) {
        for (int k = 0; k < 3; ++k)
            if (p.name == (k == 0 ? "x" : k == 1 ? "y" : "z") && p.size >= 4) { off[k] = stride; dbl[k] = p.is_double; }
        stride += p.size;
    }
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) return false;
    std::vector<char> rec((size_t)stride);
    out.clear();
    out.reserve((size_t)nvert);
    for (long i = 0; i < nvert; ++i) {
        if (!f.read(rec.data(), stride)) return false;
        std::vector<double> p(3);
        for (int k = 0; k < 3; ++k) {
            if (dbl[k]) { double d; std::memcpy(&d, rec.data() + off[k], 8); p[k] = d; }
            else { float v; std::memcpy(&v, rec.data() + off[k], 4); p[k] = (double)v; }   // float widened, as the ASCII loader does
        }
        out.push_back(p);
    }
    return true;
}

inline bool ply_is_binary(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::string line;
    for (int i = 0; i < 4 && std::getline(f, line); ++i)
        if (line.find("format binary") != std::string::npos) return true;
    return false;
}

// count, then rows (loadPoints): stops early if the file is short, like the reference's stream extraction would
inline Cloud read_text_cloud(const std::string& path) {
    Cloud out;
    std::ifstream fin(path);
    if (!fin) return out;
    long n = 0;
    if (!(fin >> n) || n <= 0) return out;
    out.reserve((size_t)n);
    for (long i = 0; i < n; ++i) {
        double x, y, z;
        if (!(fin >> x >> y >> z)) break;
        out.push_back({x, y, z});
    }
    return out;
}

inline Cloud read_cloud(const std::string& path) {
    if (ends_with(path, ".ply")) {
        if (ply_is_binary(path)) {
            Cloud c;
            if (!read_ply_binary(path, c)) c.clear();
            return c;
        }
        std::vector<char> name(path.begin(), path.end());
        name.push_back('\0');
        CPLYLoader loader;              // same loader class name / LoadModel(char*) contract as the reference
        loader.LoadModel(name.data());
        return loader.points;
    }
    return read_text_cloud(path);
}

inline bool append_xyz(const Cloud& cloud, const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "a");
    if (!f) return false;
    std::fprintf(f, "%zu\n", cloud.size());
    for (const std::vector<double>& p : cloud) std::fprintf(f, "%g %g %g\n", p[0], p[1], p[2]);
    std::fprintf(f, "\n");
    std::fclose(f);
    return true;
}

}  // namespace kss_cli
