/* The cgo shim's call sequence (generateImageHIP) in plain C, for a box without a Go toolchain:
 * tables in C-allocated memory (calloc, as the shim's C.calloc), image-texture pixels copied into C memory (as
 * C.CBytes) after rt_init(0), then rt_scene_create(desc, 0, &scene) -> rt_render(scene, &spec, NULL, pix, &stats) ->
 * rt_scene_destroy(scene), errors read through rt_last_error() on the calling thread.
 *   shim_sequence <scene.bin> <out.rgba>
 * scene.bin (written by tests/test_cpp_mirror.py from the flattened ExampleRegression):
 *   int64 n_shapes, n_materials, n_textures, n_lights; rt_image_spec; camera {from, at, up: rt_vec3; aperture, fov: double};
 *   rt_shape[n_shapes]; rt_material[n_materials]; rt_texture[n_textures] (rgba = NULL); rt_light[n_lights];
 *   then width*height*4 bytes for every RT_TEX_IMAGE texture, in table order. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtgo.h"

static void need(int ok, const char* what) {
    if (!ok) { fprintf(stderr, "shim_sequence: %s\n", what); exit(2); }
}

int main(int argc, char** argv) {
    need(argc == 3, "usage: shim_sequence scene.bin out.rgba");
    if (rt_init(0) != RT_OK) { fprintf(stderr, "panic: %s\n", rt_last_error()); return 1; } /* first: starts the runtime's warm-up */
    FILE* f = fopen(argv[1], "rb");
    need(f != NULL, "cannot open the scene file");
    int64_t n[4];
    rt_image_spec spec;
    rt_scene_desc desc;
    memset(&desc, 0, sizeof desc);
    need(fread(n, sizeof n, 1, f) == 1 && fread(&spec, sizeof spec, 1, f) == 1, "short header");
    need(fread(&desc.camera_look_from, sizeof(rt_vec3), 3, f) == 3 && fread(&desc.camera_aperture, sizeof(double), 2, f) == 2, "short camera");
    /* one extra element each: calloc(0) may return NULL, and the shim allocates the same way */
    rt_shape* shapes = calloc((size_t)n[0] + 1, sizeof *shapes);
    rt_material* materials = calloc((size_t)n[1] + 1, sizeof *materials);
    rt_texture* textures = calloc((size_t)n[2] + 1, sizeof *textures);
    rt_light* lights = calloc((size_t)n[3] + 1, sizeof *lights);
    need(shapes && materials && textures && lights, "out of memory");
    need(fread(shapes, sizeof *shapes, (size_t)n[0], f) == (size_t)n[0] && fread(materials, sizeof *materials, (size_t)n[1], f) == (size_t)n[1] &&
             fread(textures, sizeof *textures, (size_t)n[2], f) == (size_t)n[2] && fread(lights, sizeof *lights, (size_t)n[3], f) == (size_t)n[3],
         "short tables");
    for (int64_t i = 0; i < n[2]; ++i) {
        textures[i].rgba = NULL;
        if (textures[i].kind != RT_TEX_IMAGE) continue;
        const size_t bytes = (size_t)textures[i].width * (size_t)textures[i].height * 4;
        uint8_t* px = malloc(bytes);
        need(px && fread(px, 1, bytes, f) == bytes, "short texture pixels");
        textures[i].rgba = px;
    }
    fclose(f);
    desc.shapes = shapes; desc.n_shapes = n[0];
    desc.materials = materials; desc.n_materials = n[1];
    desc.textures = textures; desc.n_textures = n[2];
    desc.lights = lights; desc.n_lights = n[3];

    printf("Building BoundingVolumeHierarchy\n"); /* accelerationstructures.go:28 */
    rt_scene* scene = NULL;
    if (rt_scene_create(&desc, 0, &scene) != RT_OK) { fprintf(stderr, "panic: %s\n", rt_last_error()); return 1; }
    printf("Finished building BoundingVolumeHierarchy\n"); /* :63 */
    uint8_t* pix = malloc((size_t)spec.width * (size_t)spec.height * 4); /* image.NewRGBA(...).Pix, raytracer.go:70 */
    need(pix != NULL, "out of memory");
    rt_stats stats;
    if (rt_render(scene, &spec, NULL, pix, &stats) != RT_OK) { fprintf(stderr, "panic: %s\n", rt_last_error()); return 1; }
    printf("Finished ray tracing in %.3fms\n", stats.ms_render); /* raytracer.go:105 */
    rt_scene_destroy(scene);
    for (int64_t i = 0; i < n[2]; ++i) free((void*)textures[i].rgba);
    free(shapes); free(materials); free(textures); free(lights);
    f = fopen(argv[2], "wb");
    need(f && fwrite(pix, 4, (size_t)spec.width * (size_t)spec.height, f) == (size_t)spec.width * (size_t)spec.height, "cannot write the frame");
    fclose(f);
    free(pix);
    return 0;
}
