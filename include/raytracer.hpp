// raytracer.hpp — C++17 host-side mirror of the Go `raytracer` package API, above the C-ABI of
// librtgo_hip.so (include/rtgo.h).  The reference is compiled Go and Go is not in this image, so this
// header is the compiled-language twin of the cgo shim in INTEGRATION.md: same exported names, field
// names, argument meaning and error behaviour (Go panics -> C++ exceptions).
//
//   ImageSpec, Scene, GenerateImage                        raytracer/raytracer.go:18-107
//   Sphere, TrianglePlane (+ Rotate / Scale / Translate)    raytracer/shape.go:33-45,80-89,165-181,240-263
//   Standard, Metal, Dielectric, PhongBlinn                 raytracer/material.go:13-32
//   AmbientLight, PointLight, SpotLight                     raytracer/light.go:18-39
//   CheckersTexture, ImageTexture                           raytracer/texture.go:16-25
//   ExampleRegression                                       raytracer/example.go:9-224
//
// The Go interfaces Shape / Material / Light / texture are sealed (unexported methods), so they map
// to closed std::variant types and the flattening below is a total std::visit.
// Out of scope here: image.Decode (Go stdlib jpeg/png).  LoadRGBAImage reads the already-quantised
// pixels from a raw container ("RGBA8 <w> <h>\n" + w*h*4 bytes) that tools/ or the tests write with
// the Python LoadRGBAImage.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <istream>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "rtgo.h"

namespace raytracer {

struct Vec { double X = 0, Y = 0, Z = 0; };  // gonum r3.Vec

enum BoundingVolumeHierarchyTraversalAlgorithm { Dijkstra = 0, DepthFirstSearch = 1 };  // raytracer.go:20-23

struct ImageSpec {  // raytracer.go:25-33
    int Width = 0, Height = 0, AntiAliasingFactor = 0, RayTracingMaxDepth = 0, SoftShadowMonteCarloRepetitions = 0;
    int WorkerCount = 0;  // goroutines in the reference; accepted and ignored
    int BvhTraversalAlgorithm = Dijkstra;
};

struct RGBA {  // image.RGBA: Pix row-major RGBA8, row 0 = top
    int Width = 0, Height = 0;
    std::vector<uint8_t> Pix;
};

// ---- textures -------------------------------------------------------------------------------
struct CheckersTexture { Vec ColorFrac1, ColorFrac2; double CheckersWidth = 0, CheckersHeight = 0; };
struct ImageTexture { std::shared_ptr<const RGBA> Img; };
using Texture = std::variant<std::monostate, CheckersTexture, ImageTexture>;  // monostate = nil

inline std::shared_ptr<RGBA> LoadRGBAImage(std::istream& in) {
    std::string magic;
    int w = 0, h = 0;
    in >> magic >> w >> h;
    if (!in || magic != "RGBA8" || w <= 0 || h <= 0) throw std::runtime_error("LoadRGBAImage: not an RGBA8 container");
    in.get();  // the newline
    auto img = std::make_shared<RGBA>();
    img->Width = w; img->Height = h;
    img->Pix.resize((size_t)w * h * 4);
    in.read(reinterpret_cast<char*>(img->Pix.data()), (std::streamsize)img->Pix.size());
    if (in.gcount() != (std::streamsize)img->Pix.size()) throw std::runtime_error("LoadRGBAImage: truncated pixel data");
    return img;
}

// ---- materials ------------------------------------------------------------------------------
struct Standard { Vec ColorFrac; Texture Tex; };
struct Metal { Vec Albedo; double Fuzz = 0; };
struct Dielectric { double RefractiveIndex = 0; };
struct PhongBlinn { Vec ColorFrac, SpecularColorFrac; double SpecHardness = 0; Texture Tex; };
using Material = std::variant<Standard, Metal, Dielectric, PhongBlinn>;

// ---- shapes ---------------------------------------------------------------------------------
inline Vec rotatePoint(Vec p, Vec rv) {  // shape.go:240-263: around z, then x, then y; degrees
    const double k = M_PI / 180.0;
    double x = p.X * std::cos(k * rv.Z) - p.Y * std::sin(k * rv.Z);
    double y = p.X * std::sin(k * rv.Z) + p.Y * std::cos(k * rv.Z);
    p.X = x; p.Y = y;
    y = p.Y * std::cos(k * rv.X) - p.Z * std::sin(k * rv.X);
    double z = p.Y * std::sin(k * rv.X) + p.Z * std::cos(k * rv.X);
    p.Y = y; p.Z = z;
    x = p.X * std::cos(k * rv.Y) + p.Z * std::sin(k * rv.Y);
    z = -1 * p.X * std::sin(k * rv.Y) + p.Z * std::cos(k * rv.Y);
    p.X = x; p.Z = z;
    return p;
}
struct Sphere {
    Vec Center; double Radius = 0; Material Mat;
    void Translate(Vec tv) { Center = {tv.X + Center.X, tv.Y + Center.Y, tv.Z + Center.Z}; }  // shape.go:80-82
    void Scale(double c) { Radius *= c; }                                                      // :84-86
    void Rotate(Vec) {}                                                                        // :88-89
};
struct TrianglePlane {
    Vec PointA, PointB, PointC; bool SingleSided = false; Material Mat;
    void Translate(Vec tv) { for (Vec* p : {&PointA, &PointB, &PointC}) *p = {tv.X + p->X, tv.Y + p->Y, tv.Z + p->Z}; }
    void Scale(double c) { for (Vec* p : {&PointA, &PointB, &PointC}) *p = {c * p->X, c * p->Y, c * p->Z}; }
    void Rotate(Vec rv) { for (Vec* p : {&PointA, &PointB, &PointC}) *p = rotatePoint(*p, rv); }
};
using Shape = std::variant<Sphere, TrianglePlane>;

// ---- lights ---------------------------------------------------------------------------------
struct AmbientLight { Vec ColorFrac; double LightIntensity = 0; };
struct PointLight { Vec ColorFrac, Position; double LightIntensity = 0, SpecularLightIntensity = 0, InverseSquareLawDecayFactor = 0; };
struct SpotLight {
    Vec ColorFrac, Position; double LightIntensity = 0, SpecularLightIntensity = 0;
    Vec LookAt; double Angle = 0 /* half-angle, degrees */, InverseSquareLawDecayFactor = 0;
};
using Light = std::variant<AmbientLight, PointLight, SpotLight>;

struct Scene {  // raytracer.go:35-46
    Vec CameraLookFrom, CameraLookAt, CameraUp, CameraFocusPoint;  // CameraFocusPoint is never read (raytracer.go:59,67)
    double CameraAperature = 0, CameraFov = 0;
    std::vector<Shape> Shapes;
    std::vector<Light> Lights;
};

// ---- flattening to the C-ABI ------------------------------------------------------------------
namespace detail {
inline rt_vec3 v3(const Vec& v) { return rt_vec3{v.X, v.Y, v.Z}; }
inline bool same(const rt_vec3& a, const rt_vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
template <class... F> struct overloaded : F... { using F::operator()...; };
template <class... F> overloaded(F...) -> overloaded<F...>;

struct Flat {
    std::vector<rt_shape> shapes;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_light> lights;
    std::vector<std::shared_ptr<const RGBA>> keep;
    rt_scene_desc desc{};

    int texture(const Texture& t) {
        return std::visit(overloaded{
            [](std::monostate) { return -1; },
            [&](const CheckersTexture& c) {
                rt_texture r{};
                r.kind = RT_TEX_CHECKERS; r.color1 = v3(c.ColorFrac1); r.color2 = v3(c.ColorFrac2);
                r.checkers_width = c.CheckersWidth; r.checkers_height = c.CheckersHeight;
                if (!textures.empty()) {  // the triangles of one mesh carry the same texture value: keep one row
                    const rt_texture& p = textures.back();
                    if (p.kind == r.kind && same(p.color1, r.color1) && same(p.color2, r.color2) &&
                        p.checkers_width == r.checkers_width && p.checkers_height == r.checkers_height)
                        return (int)textures.size() - 1;
                }
                textures.push_back(r);
                return (int)textures.size() - 1;
            },
            [&](const ImageTexture& i) {
                if (!i.Img) throw std::runtime_error("ImageTexture.Img is nil");
                for (size_t q = 0; q < textures.size(); ++q)  // one upload per image
                    if (textures[q].kind == RT_TEX_IMAGE && textures[q].rgba == i.Img->Pix.data()) return (int)q;
                rt_texture r{};
                r.kind = RT_TEX_IMAGE; r.width = i.Img->Width; r.height = i.Img->Height; r.rgba = i.Img->Pix.data();
                keep.push_back(i.Img);
                textures.push_back(r);
                return (int)textures.size() - 1;
            }}, t);
    }
    int material(const Material& m) {
        rt_material r{};
        r.texture = -1;
        std::visit(overloaded{
            [&](const Standard& s) { r.kind = RT_MAT_STANDARD; r.color = v3(s.ColorFrac); r.texture = texture(s.Tex); },
            [&](const Metal& s) { r.kind = RT_MAT_METAL; r.color = v3(s.Albedo); r.param = s.Fuzz; },
            [&](const Dielectric& s) { r.kind = RT_MAT_DIELECTRIC; r.param = s.RefractiveIndex; },
            [&](const PhongBlinn& s) {
                r.kind = RT_MAT_PHONGBLINN; r.color = v3(s.ColorFrac); r.specular_color = v3(s.SpecularColorFrac);
                r.param = s.SpecHardness; r.texture = texture(s.Tex);
            }}, m);
        if (!materials.empty()) {  // same for materials: reuse the previous row when identical
            const rt_material& p = materials.back();
            if (p.kind == r.kind && p.texture == r.texture && p.param == r.param && same(p.color, r.color) &&
                same(p.specular_color, r.specular_color))
                return (int)materials.size() - 1;
        }
        materials.push_back(r);
        return (int)materials.size() - 1;
    }
    explicit Flat(const Scene& sc) {
        shapes.reserve(sc.Shapes.size());
        for (const Shape& s : sc.Shapes) {
            rt_shape r{};
            std::visit(overloaded{
                [&](const Sphere& x) { r.kind = RT_SHAPE_SPHERE; r.material = material(x.Mat); r.a = v3(x.Center); r.b = rt_vec3{x.Radius, 0, 0}; },
                [&](const TrianglePlane& x) {
                    r.kind = RT_SHAPE_TRIANGLE; r.material = material(x.Mat); r.single_sided = x.SingleSided ? 1 : 0;
                    r.a = v3(x.PointA); r.b = v3(x.PointB); r.c = v3(x.PointC);
                }}, s);
            shapes.push_back(r);
        }
        for (const Light& l : sc.Lights) {
            rt_light r{};
            std::visit(overloaded{
                [&](const AmbientLight& x) { r.kind = RT_LIGHT_AMBIENT; r.color = v3(x.ColorFrac); r.intensity = x.LightIntensity; },
                [&](const PointLight& x) {
                    r.kind = RT_LIGHT_POINT; r.color = v3(x.ColorFrac); r.position = v3(x.Position); r.intensity = x.LightIntensity;
                    r.specular_intensity = x.SpecularLightIntensity; r.decay = x.InverseSquareLawDecayFactor;
                },
                [&](const SpotLight& x) {
                    r.kind = RT_LIGHT_SPOT; r.color = v3(x.ColorFrac); r.position = v3(x.Position); r.look_at = v3(x.LookAt);
                    r.intensity = x.LightIntensity; r.specular_intensity = x.SpecularLightIntensity; r.angle_deg = x.Angle;
                    r.decay = x.InverseSquareLawDecayFactor;
                }}, l);
            lights.push_back(r);
        }
        desc.camera_look_from = v3(sc.CameraLookFrom); desc.camera_look_at = v3(sc.CameraLookAt); desc.camera_up = v3(sc.CameraUp);
        desc.camera_aperture = sc.CameraAperature; desc.camera_fov_deg = sc.CameraFov;
        desc.shapes = shapes.data(); desc.n_shapes = (int64_t)shapes.size();
        desc.materials = materials.data(); desc.n_materials = (int64_t)materials.size();
        desc.textures = textures.data(); desc.n_textures = (int64_t)textures.size();
        desc.lights = lights.data(); desc.n_lights = (int64_t)lights.size();
    }
};
inline rt_image_spec spec_of(const ImageSpec& is) {
    rt_image_spec s{};
    s.width = is.Width; s.height = is.Height; s.anti_aliasing_factor = is.AntiAliasingFactor;
    s.ray_tracing_max_depth = is.RayTracingMaxDepth; s.soft_shadow_monte_carlo_repetitions = is.SoftShadowMonteCarloRepetitions;
    s.worker_count = is.WorkerCount; s.bvh_traversal_algorithm = is.BvhTraversalAlgorithm;
    return s;
}
}  // namespace detail

// GenerateImage, raytracer.go:58-107.  Errors (the reference panics) are thrown as std::runtime_error.
inline RGBA GenerateImage(const ImageSpec& imageSpec, const Scene& scene, rt_stats* stats = nullptr) {
    if (rt_init(0) != RT_OK) throw std::runtime_error(rt_last_error());  // first: the runtime's start-up then runs under the flattening
    detail::Flat flat(scene);
    const rt_image_spec spec = detail::spec_of(imageSpec);
    std::printf("Building BoundingVolumeHierarchy\n");  // accelerationstructures.go:28
    rt_scene* dev = nullptr;
    if (rt_scene_create(&flat.desc, 0, &dev) != RT_OK) throw std::runtime_error(rt_last_error());
    std::printf("Finished building BoundingVolumeHierarchy\n");  // :63
    RGBA img;
    img.Width = imageSpec.Width; img.Height = imageSpec.Height;
    img.Pix.assign((size_t)(imageSpec.Width > 0 ? imageSpec.Width : 0) * (imageSpec.Height > 0 ? imageSpec.Height : 0) * 4, 0);
    rt_stats st{};
    const int rc = rt_render(dev, &spec, nullptr, img.Pix.data(), &st);
    const std::string err = rc == RT_OK ? "" : rt_last_error();
    rt_scene_destroy(dev);
    if (rc != RT_OK) throw std::runtime_error(err);
    std::printf("Finished ray tracing in %.3fms\n", st.ms_render);  // raytracer.go:105
    if (stats) *stats = st;
    return img;
}

// ExampleRegression, example.go:9-224, field for field.  repoBaseDir must hold
// samples_textures/Tiles075_1K_Color.rgba (the JPEG decoded + quantised by LoadRGBAImage); a missing
// file throws (the reference panics, example.go:36-44).
inline std::pair<ImageSpec, Scene> ExampleRegression(int width, int height, const std::string& repoBaseDir) {
    const double floorRadius = 100.0, centerPiecesRadius = 2.0;
    const double backMirrorRadius = 4 * centerPiecesRadius, backMirrorBorder = centerPiecesRadius / 2;
    const double r = centerPiecesRadius, R = backMirrorRadius, b = backMirrorBorder;

    const CheckersTexture texturePlane{{0, 1, 0}, {0, 0, 1}, 100.0, 100.0};
    const CheckersTexture textureLeftSphere{{0, 0, 0}, {1, 1, 1}, 10.0, 10.0};
    std::ifstream f(repoBaseDir + "/samples_textures/Tiles075_1K_Color.rgba", std::ios::binary);
    if (!f) throw std::runtime_error("open " + repoBaseDir + "/samples_textures/Tiles075_1K_Color.rgba: no such file");
    const ImageTexture textureRightSphere{LoadRGBAImage(f)};

    const Vec frame{150 / 255.0, 111 / 255.0, 51 / 255.0};
    const PhongBlinn floorMat{{0, 0, 0}, {1, 1, 1}, 1, texturePlane};
    Scene sc;
    sc.Shapes = {
        Sphere{{4 * r, r, 0}, r, Standard{{}, textureLeftSphere}},
        Sphere{{2 * r, r, 0}, r, Dielectric{1.52}},
        Sphere{{0, r, 0}, r, PhongBlinn{{1, 1, 1}, {1, 1, 1}, 1, {}}},
        Sphere{{-2 * r, r, 0}, r, Metal{{1.0, 1.0, 1.0}, 0}},
        Sphere{{-4 * r, r, 0}, r, PhongBlinn{{}, {1, 1, 1}, 1, textureRightSphere}},
        TrianglePlane{{-floorRadius, 0, -floorRadius}, {-floorRadius, 0, floorRadius}, {floorRadius, 0, -floorRadius}, true, floorMat},
        TrianglePlane{{floorRadius, 0, floorRadius}, {floorRadius, 0, -floorRadius}, {-floorRadius, 0, floorRadius}, true, floorMat},
        TrianglePlane{{R, R, R}, {R, 0, R}, {-R, R, R}, true, Standard{frame, {}}},
        TrianglePlane{{-R, 0, R}, {-R, R, R}, {R, 0, R}, true, Standard{frame, {}}},
        TrianglePlane{{R - b, R - b, R - b}, {R - b, b, R - b}, {-(R - b), R - b, R - b}, true, Metal{{1, 1, 1}, 0}},
        TrianglePlane{{-(R - b), b, R - b}, {-(R - b), R - b, R - b}, {R - b, b, R - b}, true, Metal{{1, 1, 1}, 0}},
    };
    sc.Lights = {
        AmbientLight{{255 / 255.0, 0 / 255.0, 0 / 255.0}, 0.2},
        SpotLight{{171 / 255.0, 137 / 255.0, 255 / 255.0}, {6 * r, 5 * r, -3 * r}, 100, 100, {0, 0, 0}, 30, 1.0},
        PointLight{{67 / 255.0, 163 / 255.0, 241 / 255.0}, {-4 * r, r, 3 * r}, 100, 10, 0.5},
    };
    sc.CameraLookFrom = {0, 3 * r, -5};
    sc.CameraLookAt = {0, 2 * r, 0};
    sc.CameraUp = {0, 1, 0};
    sc.CameraFocusPoint = sc.CameraLookAt;
    sc.CameraAperature = 0.015;
    sc.CameraFov = 60.0;
    ImageSpec is;
    is.Width = width; is.Height = height; is.AntiAliasingFactor = 32; is.RayTracingMaxDepth = 16;
    is.SoftShadowMonteCarloRepetitions = 16; is.WorkerCount = 16; is.BvhTraversalAlgorithm = Dijkstra;
    return {is, sc};
}

}  // namespace raytracer
