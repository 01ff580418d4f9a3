// include/PlyLoad.h -- GL-free mirror of the reference `CPLYLoader` (PS_AIS_Simplification/PlyLoad.h:21-42,
// PlyLoad.cpp:10-190): same class name, LoadModel(char*) signature / return codes and public vectors, same
// ASCII parse rules (needs "element vertex", "element face", "end_header"; x y z [nx ny nz] [c r g b] per
// vertex parsed as float then widened; faces that start with '3').  Differences: no Draw() (OpenGL viewer
// remnant), and a missing header keyword returns -4 instead of spinning forever.
#pragma once
#ifndef PLYREADER_H_
#define PLYREADER_H_

#include <cstdio>
#include <cstring>
#include <vector>

class CPLYLoader {
public:
    std::vector<std::vector<double>> points;
    std::vector<std::vector<double>> normals;
    std::vector<std::vector<double>> colors;
    std::vector<std::vector<int>> vecFaceIndex;

    CPLYLoader() : m_totalConnectedPoints(0), m_totalFaces(0) {}

    int LoadModel(char* filename) {
        std::printf("Loading %s...\n", filename);
        if (std::strstr(filename, ".ply") == NULL) {
            std::printf("File does not have a .PLY extension. ");
            return 0;
        }
        FILE* file = std::fopen(filename, "r");
        if (!file) {
            std::printf("load PLY file %s failed\n", filename);
            return false;
        }
        char buffer[1000];
        if (!std::fgets(buffer, 300, file)) { std::fclose(file); return -4; }
        if (!seek_keyword(file, buffer, "element vertex")) { std::fclose(file); return -4; }
        std::sscanf(buffer + std::strlen("element vertex"), "%i", &m_totalConnectedPoints);
        std::fseek(file, 0, SEEK_SET);
        if (!seek_keyword(file, buffer, "element face")) { std::fclose(file); return -4; }
        std::sscanf(buffer + std::strlen("element face"), "%i", &m_totalFaces);
        if (!seek_keyword(file, buffer, "end_header")) { std::fclose(file); return -4; }

        for (int it = 0; it < m_totalConnectedPoints; ++it) {
            float v[3] = {0, 0, 0}, n[3] = {0, 0, 0}, c[3] = {0, 0, 0};
            char tmp[4];
            if (!std::fgets(buffer, 300, file)) break;
            std::sscanf(buffer, "%f %f %f %f %f %f %c %f %f %f", &v[0], &v[1], &v[2], &n[0], &n[1], &n[2], tmp, &c[0], &c[1], &c[2]);
            points.push_back({(double)v[0], (double)v[1], (double)v[2]});
            normals.push_back({(double)n[0], (double)n[1], (double)n[2]});
            colors.push_back({(double)c[0], (double)c[1], (double)c[2]});
        }
        for (int it = 0; it < m_totalFaces; ++it) {
            if (!std::fgets(buffer, 300, file)) break;
            if (buffer[0] == '3') {
                int v1 = 0, v2 = 0, v3 = 0;
                buffer[0] = ' ';
                std::sscanf(buffer, "%i%i%i", &v1, &v2, &v3);
                vecFaceIndex.push_back({v1, v2, v3});
            }
        }
        std::fclose(file);
        std::printf("%s Loaded!\n", filename);
        return 0;
    }

private:
    int m_totalConnectedPoints;
    int m_totalFaces;

    static bool seek_keyword(FILE* f, char* buffer, const char* kw) {
        while (std::strncmp(kw, buffer, std::strlen(kw)) != 0)
            if (!std::fgets(buffer, 300, f)) return false;
        return true;
    }
};

#endif
