/*
 * rtgo.h — C-ABI of librtgo_hip.so, the MI355X (gfx950) render backend for the
 * raytracer-go hot path:  camera ray-gen -> BVH traversal + ray/sphere +
 * ray/triangle -> material shading -> Monte-Carlo shadow rays -> pixel
 * accumulate -> RGBA8.
 *
 * The reference (btleedev/raytracer-go) has no FFI seam: its boundary is the Go
 * package API `raytracer.GenerateImage(ImageSpec, Scene) *image.RGBA`
 * (raytracer/raytracer.go:58).  The entry points below are what a cgo shim
 * inside GenerateImage binds (see INTEGRATION.md for that shim).  Everything is
 * plain C: POD structs, caller-owned pointers + sizes, int status returns.
 *
 * Conventions
 *   - Scene inputs are float64, exactly as Go's r3.Vec{X,Y,Z float64}; the
 *     library narrows to fp32 at upload (the device path computes in fp32).
 *   - All input pointers are borrowed for the duration of the call only.
 *   - Return 0 = RT_OK; negative = error, text via rt_last_error() (thread-local).
 *   - No C++ exception leaves the library: host allocation failures come back as RT_ERR_LIMIT
 *     ("out of host memory"), other host runtime errors as RT_ERR_DEVICE.
 *   - An rt_scene owns its device workspace: one call at a time per scene (different scenes,
 *     e.g. one per device, may be used from different threads concurrently).
 *   - Shape index (position in rt_scene_desc.shapes) is the primitive ID that
 *     rt_trace_rays reports — same order as Scene.Shapes (raytracer.go:44).
 */
#ifndef RTGO_H
#define RTGO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / malformed scene                       */
#define RT_ERR_DEVICE (-2)    /* HIP runtime error or no gfx950 device                */
#define RT_ERR_ALGORITHM (-3) /* unknown BvhTraversalAlgorithm: the reference panics, */
                              /* accelerationstructures.go:73                         */
#define RT_ERR_LIMIT (-4)     /* scene exceeds a documented capacity limit: fewer than */
                              /* 2^24 shapes and 2^24 child records per scene, 2^31-1  */
                              /* pixels, depth <= 2^20, repetitions <= 65536           */

typedef struct rt_vec3 { double x, y, z; } rt_vec3; /* r3.Vec */

/* ---- textures: texture.go:16-25 ------------------------------------------ */
enum { RT_TEX_CHECKERS = 1, RT_TEX_IMAGE = 2 };
typedef struct rt_texture {
    int32_t kind;            /* RT_TEX_*                                               */
    int32_t width, height;   /* image: Img.Bounds().Size()                             */
    rt_vec3 color1, color2;  /* checkers: ColorFrac1/2                                 */
    double checkers_width, checkers_height;
    const uint8_t* rgba;     /* image: Img.Pix, row-major RGBA8, width*height*4 bytes  */
} rt_texture;

/* ---- materials: material.go:13-32 ---------------------------------------- */
enum { RT_MAT_STANDARD = 1, RT_MAT_METAL = 2, RT_MAT_DIELECTRIC = 3, RT_MAT_PHONGBLINN = 4 };
typedef struct rt_material {
    int32_t kind;            /* RT_MAT_*                                               */
    int32_t texture;         /* index into textures, -1 = nil (Standard / PhongBlinn)  */
    rt_vec3 color;           /* Standard.ColorFrac / Metal.Albedo / PhongBlinn.ColorFrac */
    rt_vec3 specular_color;  /* PhongBlinn.SpecularColorFrac                           */
    double param;            /* Metal.Fuzz / Dielectric.RefractiveIndex / PhongBlinn.SpecHardness */
} rt_material;

/* ---- shapes: shape.go:33-45 ---------------------------------------------- */
enum { RT_SHAPE_SPHERE = 1, RT_SHAPE_TRIANGLE = 2 };
typedef struct rt_shape {
    int32_t kind;            /* RT_SHAPE_*                                             */
    int32_t material;        /* index into materials                                   */
    int32_t single_sided;    /* TrianglePlane.SingleSided                              */
    int32_t _pad;
    rt_vec3 a;               /* Sphere.Center      / TrianglePlane.PointA              */
    rt_vec3 b;               /* {Radius,0,0}       / TrianglePlane.PointB              */
    rt_vec3 c;               /* unused             / TrianglePlane.PointC              */
} rt_shape;

/* ---- lights: light.go:18-39 ---------------------------------------------- */
enum { RT_LIGHT_AMBIENT = 1, RT_LIGHT_POINT = 2, RT_LIGHT_SPOT = 3 };
typedef struct rt_light {
    int32_t kind;            /* RT_LIGHT_*                                             */
    int32_t _pad;
    rt_vec3 color;           /* ColorFrac                                              */
    rt_vec3 position;        /* Point/Spot                                             */
    rt_vec3 look_at;         /* Spot                                                   */
    double intensity;        /* LightIntensity                                         */
    double specular_intensity;
    double angle_deg;        /* Spot.Angle, half-angle in degrees                      */
    double decay;            /* InverseSquareLawDecayFactor                            */
} rt_light;

/* ---- Scene (raytracer.go:35-46); CameraFocusPoint is dead in the reference -- */
typedef struct rt_scene_desc {
    rt_vec3 camera_look_from, camera_look_at, camera_up;
    double camera_aperture;  /* CameraAperature                                        */
    double camera_fov_deg;   /* CameraFov                                              */
    const rt_shape* shapes;       int64_t n_shapes;
    const rt_material* materials; int64_t n_materials;
    const rt_texture* textures;   int64_t n_textures;
    const rt_light* lights;       int64_t n_lights;
} rt_scene_desc;

/* ---- ImageSpec (raytracer.go:25-33) --------------------------------------- */
enum { RT_BVH_DIJKSTRA = 0, RT_BVH_DEPTH_FIRST_SEARCH = 1 }; /* raytracer.go:20-23 */
typedef struct rt_image_spec {
    int32_t width, height;
    int32_t anti_aliasing_factor;
    int32_t ray_tracing_max_depth;
    int32_t soft_shadow_monte_carlo_repetitions;
    int32_t worker_count;             /* accepted, ignored (goroutine count)           */
    int32_t bvh_traversal_algorithm;  /* 0|1 accepted (both map to the device stack    */
                                      /* traversal); anything else = RT_ERR_ALGORITHM  */
    int32_t _pad;
} rt_image_spec;

/* ---- knobs the Go API has no field for ------------------------------------ */
enum { RT_PART_ALL = 0, RT_PART_TILES = 1, RT_PART_SAMPLES = 2 };
typedef struct rt_partition {
    uint64_t seed;            /* Philox key; the reference uses the unseeded global math/rand */
    int32_t mode;             /* RT_PART_*                                             */
    int32_t rank, world_size; /* this call renders rank's share of tiles / samples     */
    int32_t tile_size;        /* RT_PART_TILES: square tile edge in pixels (0 = 64)    */
    int64_t max_paths_per_batch; /* 0 = library default                               */
} rt_partition;

typedef struct rt_share {       /* what one rank of a partition renders (rt_partition_share)       */
    int32_t sample_begin, sample_end; /* samples [begin, end) of each of its pixels                  */
    int32_t tile_size, _pad;
    int64_t n_tiles;             /* tiles of tile_size^2 pixels it owns (all of them unless RT_PART_TILES) */
} rt_share;

typedef struct rt_stats {
    uint64_t rays_primary;       /* closest-hit traversals at bounce 0                 */
    uint64_t rays_continuation;  /* closest-hit traversals at bounce > 0               */
    uint64_t rays_shadow;        /* shadow rays whose visibility the device decided: traced one by one, or proven
                                    unoccluded together with the other repetitions of their (hit, light) pair by one
                                    conservative walk (rays_shadow_beam of them)                                  */
    uint64_t rays_shadow_reference_equivalent; /* incl. the ones the reference traces
                                    before its spot-cone test, light.go:124            */
    uint64_t phong_hits;
    uint64_t paths;              /* (pixel, sample) pairs rendered by this call        */
    double ms_build;             /* host tree build + flatten + upload (scene create)  */
    double ms_render;            /* device time of the render call (hipEvent)          */
    double ms_extend, ms_shadow, ms_shade; /* per-stage stream time, if timed: the launch durations of k_extend and of
                                    k_shadow, and the rest of each stream's span.  A render call runs its batches on
                                    up to four streams that overlap (RTGO_LANES=1: one), the stage times are summed over
                                    them, so the three can add up to more than ms_render.                        */
    uint64_t rays_shadow_beam;   /* of rays_shadow: decided by a pair's beam walk, never generated as rays */
    double ms_shadow_busy;       /* time during which at least one stream's k_shadow was running (the union of the
                                    launches' intervals; = ms_shadow on one stream), if timed                     */
    uint64_t beam_walks;         /* conservative any-hit walks launched for whole (hit, light) pairs (k_pair_beam): one
                                    traversal each, on top of the rays_shadow - rays_shadow_beam rays walked one by one */
} rt_stats;

typedef struct rt_scene_info {
    int64_t n_shapes, n_inner_nodes, n_child_records;
    int32_t tree_depth;          /* after single-child chain collapse                  */
    int32_t stack_bound;         /* worst-case traversal stack entries                 */
    int64_t device_bytes;
} rt_scene_info;

typedef struct rt_scene rt_scene; /* opaque: flattened tree + tables resident in HBM */

/* Thread-local text of the last error.  The Go shim turns it into a panic, matching
 * the reference's error behaviour (accelerationstructures.go:73, example.go:36-44). */
const char* rt_last_error(void);

/* Library / device probe.  Fails (RT_ERR_DEVICE) when no HIP device is visible.
 * Optional (rt_scene_create calls it), but worth calling FIRST, before the caller flattens its scene into an
 * rt_scene_desc: the first call for a device starts, on a thread of the library's own, the HIP runtime's
 * one-time copy-path set-up (about 0.15 s in a fresh process), which rt_scene_create otherwise has only its
 * tree build to hide behind; rt_scene_create waits for that thread before its first upload. */
int rt_init(int device);

/* What the loaded library was built from: "src=<hash of its sources> shadow_waves=7 lds_stack=11 ..." (the compile-time
 * knobs of the kernels).  No counterpart in the reference; bench.py uses it to tell whether committed profile counts
 * belong to this build. */
const char* rt_build_info(void);

/* Replaces NewBoundingVolumeHierarchy(&scene.Shapes) (accelerationstructures.go:27-65,
 * called at raytracer.go:69) + NewCamera (camera.go:18-35, raytracer.go:59-68):
 * builds the acceleration structure on the host, flattens it to child-record / primitive arrays and
 * uploads scene, materials, textures and lights to `device`.  The structure is a binned-SAH 8-wide BVH
 * by default and the reference's insertion octree with RTGO_TREE=octree: both enclose every shape, so
 * both return the reference's hits (INTEGRATION.md lists the switches).  Coordinates must be finite. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);

/* Replaces the body of GenerateImage after the BVH build (raytracer.go:70-106): the
 * computePixel workers (raytracer.go:109-133), color() (raytracer.go:135-162) and the
 * RGBA8 quantisation (raytracer.go:93-96).  rgba_out: caller-allocated host buffer,
 * width*height*4 bytes, row 0 = top (raytracer.go:125).  stats may be NULL. */
int rt_render(rt_scene* scene, const rt_image_spec* spec, const rt_partition* part,
              uint8_t* rgba_out, rt_stats* stats);

/* Same path, but stops before the divide/quantise: un-normalised fp32 RGB sums
 * (width*height*3 floats, row 0 = top) of this rank's share are ADDED into
 * d_accum_rgb, a DEVICE pointer owned by the caller (e.g. a torch tensor that is then
 * reduced over RCCL).  hip_stream: hipStream_t to enqueue on (NULL = default stream).
 * The call BLOCKS for most of the frame: the host waits once per bounce of every batch for
 * the queue lengths (on an event recorded before that bounce's shadow rays, so the stream
 * itself never drains).  With stats != NULL it returns when the frame is complete; with
 * stats == NULL it returns once the last batch is enqueued, its tail still running on the
 * stream.  The batches are dealt out over up to four streams — hip_stream and three that the
 * scene owns, each fed by a host thread that lives for the call — forked from hip_stream at
 * the start of the call and joined back to it at the end: what the caller queued on hip_stream before the call is seen by every
 * batch, what it queues afterwards comes after all of them (RTGO_LANES=1: one stream). */
int rt_render_accum_device(rt_scene* scene, const rt_image_spec* spec, const rt_partition* part,
                           float* d_accum_rgb, void* hip_stream, rt_stats* stats);

/* rt_render_accum_device for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, processes whose memory is all spoken for).  It has no counterpart in the reference.  Same arguments,
 * same result (bit-identical to rt_render_accum_device with part->max_paths_per_batch = 262144), but the batch
 * workspace — ray state, queues, counters, the traversal-stack spill region — is a 72 MiB static device array
 * that every GPU the library runs on holds until the process ends, instead of memory the scene allocates on
 * demand.  The call makes no hipMalloc / hipFree / hipHostMalloc / hipEventCreate; per-stage times
 * (ms_extend, ms_shadow) are not measured.  It is NOT capturable into a HIP graph: like rt_render_accum_device it
 * blocks on a host wait per bounce (hipEventSynchronize, queue lengths read back through pinned memory), which a
 * stream under capture does not allow.  Limits (RT_ERR_LIMIT beyond): tile_size^2 <= 262144 paths per batch,
 * at most 4 positional lights, RayTracingMaxDepth <= 64, a scene whose tree needs at most 54 traversal-stack
 * entries (rt_scene_info.stack_bound + 1).  All calls on one GPU share that GPU's array: two of them that
 * overlap in time (two host threads, two streams) race; successive calls need nothing from each other. */
int rt_render_accum_device_static(rt_scene* scene, const rt_image_spec* spec, const rt_partition* part,
                                  float* d_accum_rgb, void* hip_stream, rt_stats* stats);

/* Host-buffer convenience over rt_render_accum_device (used for RMSE checks). */
int rt_render_accum(rt_scene* scene, const rt_image_spec* spec, const rt_partition* part,
                    float* accum_rgb_out, rt_stats* stats);

/* raytracer.go:93-96 + :119 on device: c = sum/spp; uint8(min(255, c*255.99)); A=255.
 * d_accum_rgb: device, width*height*3 floats.  d_rgba_out: device, width*height*4. */
int rt_resolve_device(const float* d_accum_rgb, int32_t width, int32_t height, int32_t spp,
                      uint8_t* d_rgba_out, void* hip_stream);

/* One-shot GenerateImage(imageSpec, scene): scene create + render + destroy. */
int rt_generate_image(const rt_scene_desc* desc, const rt_image_spec* spec,
                      const rt_partition* part, uint8_t* rgba_out, rt_stats* stats);

/* ---- one frame on several GPUs of one node, behind one blocking call ----------------------
 * Replaces the goroutine fan-out inside GenerateImage (raytracer.go:73-87: WorkerCount workers fed from one
 * job channel) with one host thread per device: the scene is replicated (rt_multi_create builds the tree and the
 * tables once on the host and uploads them to all devices at the same time), share r of the frame — tiles k with k % n == r (RT_PART_TILES) or samples
 * [r*spp/n, (r+1)*spp/n) (RT_PART_SAMPLES) — is rendered on devices[r] into that device's fp32 accumulation
 * buffer, the buffers are summed onto devices[0] with one ncclReduce over xGMI (communicators from
 * ncclCommInitAll, librccl opened on first need), devices[0] divides, quantises (raytracer.go:93-96,119) and
 * copies the RGBA8 frame to rgba_out.  A sample's value does not depend on n (the Philox counter carries global
 * pixel / sample indices): RT_PART_TILES frames are bit-identical for every n, RT_PART_SAMPLES frames up to the
 * order of the float32 sums.  A device may be listed more than once (its shares are then summed on it without
 * RCCL): a rehearsal of an n-share frame on fewer GPUs.  stats: counters summed, times = slowest share. */
/* The share arithmetic on its own (no device needed): what rank `part->rank` of `part->world_size` renders. */
int rt_partition_share(const rt_image_spec* spec, const rt_partition* part, rt_share* out);

typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc* desc, const int32_t* devices, int32_t n_devices, rt_multi** out);
void rt_multi_destroy(rt_multi* multi);
int rt_multi_render(rt_multi* multi, const rt_image_spec* spec, int32_t mode, uint64_t seed,
                    uint8_t* rgba_out, rt_stats* stats);
/* A share that fails (allocation, launch) makes the whole call fail: every share's host thread is joined first, and
 * only when all of them rendered is the reduce enqueued (all ranks inside one ncclGroupStart/End), so no rank is left
 * waiting in a collective; on failure every device is synchronised before the first error is returned. */
typedef struct rt_multi_info {
    int32_t n_shares;            /* entries of `devices` in rt_multi_create                                   */
    int32_t n_distinct_devices;
    int32_t rccl_ranks;          /* communicators ncclCommInitAll made: n_shares when they are >= 2 distinct  */
                                 /* devices, 0 when the shares are summed on one device without RCCL          */
    int32_t _pad;
} rt_multi_info;
int rt_multi_get_info(const rt_multi* multi, rt_multi_info* out);
/* one shot: create + render + destroy */
int rt_generate_image_multi(const rt_scene_desc* desc, const rt_image_spec* spec, const int32_t* devices,
                            int32_t n_devices, int32_t mode, uint64_t seed, uint8_t* rgba_out, rt_stats* stats);

/* Parity entry point: nearest hit of explicit rays, as bvh.trace(r, tMin)
 * (accelerationstructures.go:81-124).  rays: n * 8 floats {ox,oy,oz, dx,dy,dz, tMin, 0};
 * prim_out[n]: shape index or -1; t_out[n]: hit t (undefined on miss).
 * any_hit != 0 runs the shadow-ray traversal instead (light.go:141-160 semantics with
 * ray slot 7 = tMax): prim_out = 1 if occluded else 0.  any_hit == 2: the same with ONE tMin for
 * the whole call (ray 0's slot 6; light.go:150 uses 0.01 for every shadow ray): the form of the
 * walk the render path itself runs, in which idle lanes of a wavefront walk subtrees for the
 * lanes that are still busy.
 * The tree's boxes are grown by 2^-19 of the largest coordinate a ray of the render can start at (the
 * shapes' box, the camera's lens), which makes the float32 box test conservative against the shapes'
 * own tests for such rays; a ray given here that starts much farther from the world's origin than
 * that walks boxes that are tighter than its own rounding. */
int rt_trace_rays(rt_scene* scene, const float* rays, int64_t n, int32_t any_hit,
                  int32_t* prim_out, float* t_out);

/* Camera rays exactly as the device generates them (camera.go:37-44 + raytracer.go:114-116)
 * for parity tests: for sample s of pixel (i, j) — j counted from the bottom as in the
 * reference — writes {ox,oy,oz,dx,dy,dz} * n.  pixels: n * 3 int32 {i, j, s}. */
int rt_camera_rays(rt_scene* scene, const rt_image_spec* spec, uint64_t seed,
                   const int32_t* pixels, int64_t n, float* rays_out);

/* Parity entry point: material.scatter (material.go:34-132) of the nearest hit of explicit rays,
 * as color() evaluates it at raytracer.go:146.  rays: n * 6 floats {o, d}; keys: n * 3 uint32
 * {pixel index, sample index, bounce} = the Philox counter words of the path the ray belongs to.
 * out: n * 20 floats {shouldTrace, attenuation.xyz, scattered origin.xyz, scattered direction.xyz,
 * terminal colour.xyz, t, hit point.xyz, normal.xyz}; a miss gives zeros and t = -1.  PhongBlinn
 * hits trace their soft-shadow rays (spec->soft_shadow_monte_carlo_repetitions per light). */
int rt_scatter_probe(rt_scene* scene, const rt_image_spec* spec, uint64_t seed, const float* rays,
                     const uint32_t* keys, int64_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* RTGO_H */
