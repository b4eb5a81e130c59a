// example_regression — the reference's main.go flow (main.go:14-33: ExampleRegression ->
// GenerateImage -> write the image) through include/raytracer.hpp and librtgo_hip.so.
// Writes the frame as raw RGBA8 (width*height*4 bytes) so tests can compare it byte for byte
// with the Python mirror; PNG encoding is the reference's L0 CLI and stays out of scope.
//
//   example_regression <width> <height> <asset dir> <out.rgba> [spp]
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "raytracer.hpp"

int main(int argc, char** argv) {
    if (argc < 5) {
        std::cerr << "usage: " << argv[0] << " <width> <height> <asset dir> <out.rgba> [spp]\n";
        return 2;
    }
    try {
        auto [spec, scene] = raytracer::ExampleRegression(std::atoi(argv[1]), std::atoi(argv[2]), argv[3]);
        if (argc > 5) spec.AntiAliasingFactor = std::atoi(argv[5]);
        rt_stats st{};
        const raytracer::RGBA img = raytracer::GenerateImage(spec, scene, &st);
        std::ofstream out(argv[4], std::ios::binary);
        out.write(reinterpret_cast<const char*>(img.Pix.data()), (std::streamsize)img.Pix.size());
        if (!out) throw std::runtime_error(std::string("cannot write ") + argv[4]);
        std::cout << "rays " << st.rays_primary + st.rays_continuation + st.rays_shadow << "\n";
    } catch (const std::exception& e) {  // the reference panics
        std::cerr << "panic: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
