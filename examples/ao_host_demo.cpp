// ao_host_demo.cpp -- compiled host code driving libmeao_hip.so through include/meao.hpp,
// the way a C++ engine (or the C# wrapper via P/Invoke) would: no Python, no torch.
//
//   ao_host_demo <width> <height> <depth.f32> <ao.out> [intensity] [thicknessModifier]
//
// Reads width*height raw float32 depth, writes width*height R8 AO texels.  Camera terms are
// the defaults of the synthetic inputs (near 0.1, far 100, fovY 60 deg, reversed Z).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "meao.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s width height depth.f32 ao.out [intensity] [thicknessModifier]\n", argv[0]);
        return 2;
    }
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    std::vector<float> depth(static_cast<size_t>(w) * h);
    std::FILE *f = std::fopen(argv[3], "rb");
    if (!f || std::fread(depth.data(), sizeof(float), depth.size(), f) != depth.size()) {
        std::fprintf(stderr, "cannot read %s\n", argv[3]);
        return 2;
    }
    std::fclose(f);
    try {
        MiniEngineAO::AmbientOcclusion ao(w, h);
        ao.nearClipPlane = 0.1f;
        ao.farClipPlane = 100.0f;
        const float aspect = static_cast<float>(w) / static_cast<float>(h);
        ao.projection00 = static_cast<float>(1.0 / (std::tan(60.0 * M_PI / 180.0 * 0.5) * aspect));
        if (argc > 5) ao.intensity = static_cast<float>(std::atof(argv[5]));
        if (argc > 6) ao.thicknessModifier = static_cast<float>(std::atof(argv[6]));
        std::vector<unsigned char> out(depth.size());
        ao.RenderHost(depth.data(), out.data());
        f = std::fopen(argv[4], "wb");
        if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) {
            std::fprintf(stderr, "cannot write %s\n", argv[4]);
            return 2;
        }
        std::fclose(f);
        double sum = 0;
        for (unsigned char v : out) sum += v;
        std::printf("ok %dx%d mean_ao %.6f\n", w, h, sum / 255.0 / out.size());
    } catch (const MiniEngineAO::Error &e) {
        std::fprintf(stderr, "meao error %d: %s\n", e.status(), e.what());
        return 1;
    }
    return 0;
}
