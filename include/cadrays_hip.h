/* cadrays_hip.h -- C ABI of libcadrays_hip.so, the MI355X path-tracing backend that
 * stands where CADRays calls OCCT's ray-tracing core.
 *
 * Every entry point cites the reference call (file:line under /root/reference) it
 * replaces; see INTEGRATION.md for the shim a CADRays maintainer would add.
 *
 * Conventions: plain C, opaque handle, int status (0 = ok, negative = CRH_E_*), every float handed over must be finite
 * (coordinates, transforms, light and camera vectors additionally |x| <= 1e30) -- NaN / Inf inputs are rejected with CRH_E_INVALID,
 * caller owns all input buffers (copied during the call), the module owns device
 * memory, one host thread per context, one context per GPU.  No torch types.
 */
#ifndef CADRAYS_HIP_H
#define CADRAYS_HIP_H

#include <stdint.h>

#include "crh_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define CRH_API __attribute__((visibility("default")))
#else
#define CRH_API
#endif

#define CRH_OK            0
#define CRH_E_INVALID    -1   /* bad argument / bad state               */
#define CRH_E_DEVICE     -2   /* HIP runtime error (see crh_last_error) */
#define CRH_E_NOMEM      -3
#define CRH_E_NOTBUILT   -4   /* render/trace before crh_build          */

typedef struct crh_ctx crh_ctx;

/* The double-layer material, field-for-field Graphic3d_BSDF as the reference fills
 * and serialises it (MaterialEditor.cxx:281-338, ImportExport.cxx:164-231):
 *   Kc.rgb coat weight, Kc.w coat roughness        (MaterialEditor.cxx:890-896)
 *   Kd.rgb diffuse weight                          (:680); Kd.w = diffuse texture slot + 1 (0 = untextured)
 *                                                   [the aspect's Kd map, AisMesh.cxx:321-346, rttexture]
 *   Ks.rgb glossy weight,  Ks.w base roughness     (:707-713)
 *   Kt.rgb transmission weight                     (:811); Kt.w = texture scale S (0 = 1)  (rttexture -scale S T,
 *   Le.rgb emission                                (:1068-1089); Le.w = texture scale T (0 = 1)   ImportExportPlugin.cxx:679-727)
 *   Absorption.rgb colour, .w coefficient          (:817-823)
 *   FresnelCoat / FresnelBase: serialised Graphic3d_Fresnel (MaterialEditor.cxx:209-255):
 *       Schlick    : x,y,z = F0 rgb (x >= 0)
 *       Constant   : x = -1, z = value
 *       Conductor  : x = -2, y = n, z = k
 *       Dielectric : x = -3, y = n
 */
typedef struct crh_bsdf {
  float Kc[4];
  float Kd[4];
  float Ks[4];
  float Kt[4];
  float Le[4];
  float Absorption[4];
  float FresnelCoat[4];
  float FresnelBase[4];
} crh_bsdf;                       /* 128 B */

#define CRH_FRESNEL_CONSTANT   -1.0f
#define CRH_FRESNEL_CONDUCTOR  -2.0f
#define CRH_FRESNEL_DIELECTRIC -3.0f

/* V3d directional / positional light as the light editor sets it
 * (LightSourcesEditor.cxx:242-310): colour*intensity -> emission;
 * directional: vec = direction the light travels, smoothness = cone half-angle [rad];
 * positional : vec = position, smoothness = sphere radius. */
typedef struct crh_light {
  float vec[3];
  float is_point;                 /* 0 = directional, 1 = positional */
  float emission[3];
  float smoothness;
} crh_light;                      /* 32 B */

/* Graphic3d_Camera subset the reference drives (AppViewer.cxx:993-1042,
 * SettingsWidget.cxx:179-229). */
typedef struct crh_camera {
  float eye[3];
  float dir[3];
  float up[3];
  float fovy_deg;                 /* perspective: vertical field of view      */
  float aspect;                   /* width / height; <= 0 -> taken from params */
  int32_t is_ortho;
  float ortho_scale;              /* orthographic: half height of the view    */
  float aperture_radius;          /* CameraApertureRadius  (0 = pinhole)      */
  float focal_dist;               /* CameraFocalPlaneDist                     */
} crh_camera;

/* Graphic3d_RenderingParams subset (SettingsWidget.cxx:65-90, 263-477). */
typedef struct crh_params {
  uint32_t width, height;         /* render target (SettingsWidget.cxx:93-123)        */
  uint32_t max_depth;             /* RaytracingDepth 1..32 (:310-315)                  */
  float    radiance_clamp;        /* RadianceClampingValue (:318-325); <=0 = no clamp  */
  int32_t  two_sided;             /* TwoSidedBsdfModels (:328-333)                     */
  int32_t  coherent_rng;          /* CoherentPathTracingMode (:419-424)                */
  uint32_t seed;                  /* Bullard generator seed; OCCT restarts with 1      */
  uint32_t tile_size;             /* RT tile edge in pixels (sharding unit), e.g. 32   */
  int32_t  tonemap_mode;          /* 0 = disabled, 1 = filmic (:343-404)               */
  float    exposure;              /* stops                                              */
  float    white_point;
  float    background[3];         /* linear radiance used when no env map is set       */
  int32_t  env_as_background;     /* UseEnvironmentMapBackground (LightSourcesEditor.cxx:359-364) */
  float    scene_epsilon;         /* <= 0: auto = max(1e-6, 1e-5 * scene diagonal)      */
  int32_t  russian_roulette;      /* 1 = on (default in GI mode)                        */
} crh_params;

typedef struct crh_stats {
  uint64_t rays_nearest;          /* nearest-hit rays traced            */
  uint64_t rays_any;              /* any-hit (shadow) rays traced       */
  uint64_t nodes_nearest;         /* QBVH inner-node visits by nearest-hit rays  (N_inner = nearest + any) */
  uint64_t tris_nearest;          /* ray/triangle tests by nearest-hit rays      (N_tri   = nearest + any) */
  uint64_t nodes_any;             /* ... by any-hit rays                */
  uint64_t tris_any;
  uint64_t shaded_hits;           /* H                                  */
  uint64_t samples;               /* pixel-samples accumulated (S)      */
  double   seconds;               /* sum of the device time spans of the crh_render* calls (HIP events around each call's launches).  Frames that are
                                   * pipelined (back-to-back Redraw()s overlap on up to crh_set_pipeline_depth = 2 .. 8 streams) each contribute their own span, so the sum can
                                   * exceed wall time by up to the pipeline depth: use wall time around crh_sync for rates of a free-running loop */
} crh_stats;

/* create/destroy == new OpenGl_GraphicDriver + V3d_Viewer + CreateView + FBOCreate
 * (AppViewer.cxx:601-638) / release (AppViewer.cxx:1268). */
CRH_API crh_ctx*    crh_create(int device_ordinal);
CRH_API void        crh_destroy(crh_ctx* ctx);
CRH_API const char* crh_last_error(crh_ctx* ctx);

/* geometry == what AIS Display/SetLocation feeds OpenGl_SceneGeometry: indexed
 * triangle arrays with normals (+uv) and a material id per triangle
 * (AisMesh.cxx:357-423), per-object 3x4 row-major transforms (DataNode.cxx:239-242).
 * With tri_object + obj_xform the scene is a TWO-LEVEL BVH like OCCT's: vertices are in object space (each vertex must belong to one
 * object), an object with a transform gets its own tree and a top-level tree over the instances' world boxes carries the transforms.
 * Without them the arrays are world space and one tree is built.  STATIC / MOVED SPLIT: at crh_build EVERY object is baked into ONE world-space
 * tree with the transform it has at that moment (each vertex is transformed once, on the host; an object at the identity keeps its bits, so a
 * scene whose objects all sit at the identity is bit for bit the scene without objects) -- a loaded scene renders at the single-level rate
 * whatever locations its objects carry.  Only an object whose transform is CHANGED afterwards (crh_set_transforms) is rendered as an instance:
 * rays walk the static tree first and then the top-level tree of the moved objects (skipped when the ray does not come near one: crh_ray_near_sphere, crh_math.h). */
CRH_API int crh_set_geometry(crh_ctx* ctx,
                     const float* pos, const float* nrm, const float* uv, uint32_t n_vertices,
                     const int32_t* tri /* 4*nT: i0,i1,i2,material */, uint32_t n_triangles,
                     const int32_t* tri_object /* nT or NULL */,
                     const float* obj_xform /* 12*nO or NULL */, uint32_t n_objects);
/* == AIS_InteractiveObject::SetLocalTransformation / the manipulator moving an object (ImRaytraceControls.cxx:58-89,
 * DataNode.cxx:239-242): new 3x4 transforms for the n_objects of the two-level scene.  Only the top-level tree is rebuilt; restarts
 * accumulation.  The static tree is never rebuilt: when an object's transform differs from the one it was built with, its triangles there are
 * disabled in place (their leaves stay, the test rejects) and the object gets an object-space tree of its own -- built the first time, about a
 * millisecond per thousand triangles, and kept; when it returns to its build-time transform its triangles are restored and the instance is
 * dropped.  The result depends on the transforms given to crh_build and on the current ones, not on the calls in between. */
CRH_API int crh_set_transforms(crh_ctx* ctx, const float* obj_xform /* 12*nO */, uint32_t n_objects);
/* == AIS_InteractiveContext::Display / Erase of objects already in the scene: the eye icons of the GUI's scene tree and `rtdisplay` / `rterase`
 * (src/ImportExport/DataNode.cxx:304-344, ImportExportPlugin.cxx:373-425).  visible[i] != 0: object i is displayed.  No tree is rebuilt except the
 * top level: an erased object's triangle records in the static tree are disabled in place -- what crh_set_transforms does for a moved object -- and
 * its instance, if it is one, is dropped from the top level; displaying it again restores them.  `Remove` is Erase for good: hide the object and leave
 * it out of the arrays of the next crh_set_geometry.  Restarts accumulation.  Before crh_build the flags are kept and applied by the build (every
 * object is baked, the erased ones are then disabled: showing one later costs nothing).  The image equals, bit for bit, that of the scene built
 * without the erased objects for as long as they sat at their build-time placement AND the scene's bounds are the same without them (an erased object
 * keeps its place in the static tree: the bounds, hence the ray offset and the guard band of the box test, stay what they were) -- same triangle
 * arithmetic, another tree: only visit counters and the winner among hits at EQUAL distance can differ.  Needs a scene handed over with objects; a new crh_set_geometry displays everything again. */
CRH_API int crh_set_visibility(crh_ctx* ctx, const uint8_t* visible /* n_objects */, uint32_t n_objects);
/* == AIS_InteractiveContext::Display of a NEW object in a running viewer (`rtmeshread` into a loaded scene, ImportExportPlugin.cxx:132-354; the clone
 * button, main.cxx:117): the object's arrays (vertex indices local to it, material ids into the table of crh_set_materials -- extend that first) are
 * appended to the scene's, it gets an object-space tree and enters the top level as an instance under `xform` (3x4 row-major).  Cost: the object's
 * own tree (about a millisecond per thousand triangles) + the top level; nothing of the built scene is touched.  *object_out (may be NULL) = its index:
 * crh_set_transforms / crh_set_visibility take n_objects + 1 entries from now on.  The next full crh_build bakes it into the static tree like every
 * other object.  Restarts accumulation.  Needs a BUILT scene handed over with objects (CRH_E_NOTBUILT / CRH_E_INVALID otherwise). */
CRH_API int crh_add_object(crh_ctx* ctx, const float* pos, const float* nrm, const float* uv /* or NULL */, uint32_t n_vertices,
                   const int32_t* tri /* 4*nT: i0,i1,i2,material */, uint32_t n_triangles, const float* xform /* 12 */, uint32_t* object_out);
/* == Graphic3d_MaterialAspect::SetBSDF + SynchronizeAspects (MaterialEditor.cxx:331-337, Utils.cxx:57-93) */
CRH_API int crh_set_materials(crh_ctx* ctx, const crh_bsdf* m, uint32_t n);
/* == V3d_Viewer::SetLightOn/UpdateLights (LightSourcesEditor.cxx:47-87, 401-413) */
CRH_API int crh_set_lights(crh_ctx* ctx, const crh_light* l, uint32_t n);
/* == V3d_View::SetTextureEnv (LightSourcesEditor.cxx:339-354); rgb = W*H*3 linear float
 * lat-long, NULL = constant crh_params.background */
CRH_API int crh_set_envmap(crh_ctx* ctx, const float* rgb, uint32_t w, uint32_t h);
/* == Graphic3d_AspectFillArea3d::SetTextureMap (AisMesh.cxx:340-345, ImportExportPlugin.cxx:737-742): linear float
 * image for texture slot `slot`, `channels` = 3 (RGB) or 4 (RGBA) floats per texel (row 0 = top, v = 1); NULL clears
 * the slot.  Needs uv in crh_set_geometry.  The texel (bilinear, repeat wrap) multiplies Kd; the alpha a of an RGBA
 * image cuts the surface out: Kd *= a, Kt = (1 - a) + a * Kt. */
CRH_API int crh_set_texture(crh_ctx* ctx, uint32_t slot, const float* texels, uint32_t w, uint32_t h, uint32_t channels);
/* == Graphic3d_Camera setters (AppViewer.cxx:993-1042) */
CRH_API int crh_set_camera(crh_ctx* ctx, const crh_camera* cam);
/* == ChangeRenderingParams() field writes (SettingsWidget.cxx:263-477) */
CRH_API int crh_set_params(crh_ctx* ctx, const crh_params* p);
/* The switches of crh_spec.h: where this backend's frozen spec departs from the (recollected, unverifiable) arithmetic of OCCT's
 * shaders behind V3d_View::Redraw() (AppViewer.cxx:1047).  Defaults = the spec every golden vector was generated with.  Setting
 * them restarts accumulation.  crh_spec_order_exact() reports the one build-time switch (CRH_SPEC_ORDER_EXACT) of this library. */
CRH_API int crh_set_spec(crh_ctx* ctx, const crh_spec* spec);
/* spec_out->size is IN / OUT: set it to sizeof(crh_spec) as the caller's header defines the struct before the call; exactly that many bytes are written
 * (a caller built against round 3's 24-byte struct gets its five switches and nothing past its buffer); any other size -> CRH_E_INVALID, nothing written. */
CRH_API int crh_get_spec(crh_ctx* ctx, crh_spec* spec_out);
CRH_API int crh_spec_order_exact(void);
CRH_API int crh_spec_anyhit_slot_order(void);      /* the other build-time switch, CRH_SPEC_ANYHIT_SLOT_ORDER */
/* == OCCT updateRaytraceGeometry + uploadRaytraceData on a changed scene: BVH build,
 * QBVH collapse, upload.  Invalidates the accumulator. */
CRH_API int crh_build(crh_ctx* ctx);
/* crh_build with the tree handed over instead of built: `nodes` (n_nodes x 16 dwords) as crh_get_bvh / crh_build_bvh_host returned them for the SAME
 * geometry in another context or process, prim_order[leaf position] = triangle (the dword 3 of crh_get_bvh's leaf-ordered triangle records).  One
 * process per GPU on one host (the scene is replicated): local rank 0 builds once, the other ranks take its tree -- at 10 M triangles
 * the build is 4.5 s on 16 threads, and 8 ranks building at once get 2 threads each (bench.py --gpus N, cadrays_amd/sharding.py share_tree).  The
 * builder is deterministic, so the bytes are the ones crh_build would produce here.  Single-level scenes only; the array is validated (every child
 * reference inside it, the leaf order a permutation): CRH_E_INVALID otherwise.  Replaces OCCT's per-context BVH build behind
 * AIS_InteractiveContext::Display (AisMesh.cxx:357-423). */
CRH_API int crh_build_prebuilt(crh_ctx* ctx, const float* nodes, uint32_t n_nodes, const uint32_t* prim_order, uint32_t n_tris);
/* == accumulation restart (camera/scene/param change, AppViewer.cxx:979-984) */
CRH_API int crh_reset(crh_ctx* ctx);
/* == n x V3d_View::Redraw() (AppViewer.cxx:1047): +n samples per pixel over the whole target */
CRH_API int crh_render(crh_ctx* ctx, uint32_t n_iterations);
/* The RT tile entry point (adaptive tiles, SettingsWidget.cxx:451-476): render samples
 * [first_sample, first_sample + n_samples) of the listed tiles only.  Tiles are
 * tile_size x tile_size, numbered row-major over ceil(W/ts) x ceil(H/ts). */
CRH_API int crh_render_tiles(crh_ctx* ctx, const uint32_t* tile_ids, uint32_t n_tiles,
                     uint32_t first_sample, uint32_t n_samples);
/* Adaptive screen sampling == Graphic3d_RenderingParams::AdaptiveScreenSampling + NbRayTracingTiles
 * (SettingsWidget.cxx:427-477): with `on`, every crh_render iteration renders +1 sample on `tiles_per_iteration`
 * tiles drawn with probability proportional to their estimated error instead of on the whole target.
 * Changing it restarts accumulation. */
CRH_API int crh_set_adaptive(crh_ctx* ctx, int on, uint32_t tiles_per_iteration);
/* == Graphic3d_RenderingParams::ShowSamplingTiles ("Show distribution", SettingsWidget.cxx:443-449; debug view of the
 * adaptive sampler): with `on`, crh_read_ldr outlines in red (255,0,0) the tiles the most recent adaptive iteration
 * sampled.  Only the LDR read-out changes; the accumulator and crh_read_hdr never see the overlay. */
CRH_API int crh_set_show_tiles(crh_ctx* ctx, int on);
/* Speculative look-ahead for the +1-spp-per-Redraw() usage (AppViewer.cxx:1045-1047): with frames > 1, crh_render traces the
 * next `frames` whole-frame samples in ONE wide batch (late bounces stay wide) and keeps their radiance in the path buffer;
 * each call folds in only the samples it asked for, so the image after every call is bit-identical to frames = 1.  Any
 * change of scene / camera / parameters, crh_reset, crh_render_tiles or adaptive mode discards what is pending.  Ray
 * counters include the speculative samples.  frames = 1 (default) disables it. */
CRH_API int crh_set_lookahead(crh_ctx* ctx, uint32_t frames);
/* Look-ahead that follows the session (AppViewer.cxx:979-984 restarts the accumulation on every camera change, :1045-1047 then issues one
 * Redraw() per GUI frame): the first crh_render(1) after a restart traces ONE sample -- the frame the user sees while dragging costs what it
 * always did -- the next batch holds 4, then 16, ... up to max_frames samples, so a view left alone converges at the rate of the wide schedule
 * (a speculative batch costs ~1.3 ms per sample at 1080p on C3 where a lone frame costs 2.5).  Every restart (crh_reset and all that imply it)
 * starts again at one; what was traced ahead is dropped as with crh_set_lookahead.  Images are the same bit for bit.  max_frames <= 1 switches
 * it off (default); while on, it takes the place of crh_set_lookahead's fixed batch. */
CRH_API int crh_set_lookahead_auto(crh_ctx* ctx, uint32_t max_frames);
/* Which of the two wavefront schedules a batch takes.  CRH_SCHEDULE_AUTO (default): batches above ~12 M paths run the WIDE schedule
 * (one stream, full persistent grids, the plain traversal kernels); smaller ones -- one Redraw(), adaptive iterations, tile subsets --
 * run the SMALL one (two tile ranges on two streams or pipelined frames, grids that follow the batch, the work-donating traversal
 * kernels).  Both produce the same image bit for bit; the switch exists so that the parity tests and bench.py's parity gate can put
 * ANY workload through the exact kernel instantiations and launch order the headline number is measured on (CRH_SCHEDULE_WIDE), or
 * through the small-batch ones (CRH_SCHEDULE_SMALL: every batch that fits, whatever its size). */
#define CRH_SCHEDULE_AUTO  0
#define CRH_SCHEDULE_WIDE  1
#define CRH_SCHEDULE_SMALL 2
/* Round 5: a small batch is ONE launch of the frame kernel (cadrays_amd/csrc/k_frame.h: every workgroup streams its own paths through ray generation, traversal
 * and shading; no launch boundary between bounces) -- that is what AUTO and SMALL run for batches below 2^25 path slots.  CRH_SCHEDULE_STAGED: every batch that
 * fits takes the small-batch schedule in its STAGED form instead (one launch per stage and bounce on tile-range / frame-pipeline streams, the work-donating
 * traversal kernels): the schedule of rounds 2-4, kept as the reference the frame kernel's frames are compared with.  Same image, bit for bit, in every mode. */
#define CRH_SCHEDULE_STAGED 3
CRH_API int crh_set_schedule(crh_ctx* ctx, int mode);
/* Device-memory budget of the wavefront path state: at most `max_paths` path slots (196 B each) are in flight per batch; a render
 * that needs more is cut into tile groups / sample batches (same image, bit for bit).  Default 2^29 slots = 105 GB of the
 * 288 GB, allocated on demand (a 1080p Redraw() takes 0.4 GB, a 128-sample call at 1080p 52 GB): every launch of the schedule ends in a
 * drain phase of fixed length, so wide batches are faster -- 32 M / 64 M / 128 M / 256 M slots reach 80 / 86 / 91 / 93 % of the 512 M-slot rate on
 * the 1 M-triangle benchmark.  A call with many samples is cut into tile groups of up to 1024 samples (multiples of 64), not into sample
 * groups of all tiles: samples of one pixel that travel together share most of their walk.  A host that shares the GPU with other consumers lowers it here (the environment variable
 * CRH_MAX_PATHS sets the initial value).  1024 <= max_paths <= 2^30; buffers already larger are released. */
/* Frames in flight of free-running crh_render(ctx, 1) calls (the application's loop, AppViewer.cxx:1045-1047, when it does not read every frame
 * back): each call only enqueues, frame i runs on stream i mod `frames` with its own slice of the path state and is folded into the accumulator
 * after frame i - 1.  2 .. 8, default 3 (CRH_PIPE_DEPTH in the environment sets the default).  MORE THAN 3 NEEDS MORE HARDWARE QUEUES than the
 * HIP runtime creates by default (4): export GPU_MAX_HW_QUEUES >= frames + 2 (16 is fine) before the process's first HIP call -- with it, C3 at
 * 1080p renders 385 / 400 / 442 / 455 Redraw()/s at 3 / 4 / 6 / 8 frames in flight; without it streams share queues and 4 frames are SLOWER than 3
 * (320).  The library never touches the environment: it reads GPU_MAX_HW_QUEUES at its first use (crh_create / this query), crh_query_pipeline_capacity reports what this
 * process supports (max_frames = min(8, max(3, hw_queues - 2)); 3 on the runtime's default four queues) and a deeper request is refused with
 * CRH_E_INVALID and a message that names the variable.  Images do not depend on the depth.  Waits for the frames in flight. */
CRH_API int crh_set_pipeline_depth(crh_ctx* ctx, uint32_t frames);
CRH_API int crh_query_pipeline_capacity(uint32_t* max_frames, int* hw_queues);
/* Every environment variable the library reads, one "NAME<TAB>what it does" line each (reference-schedule selectors for tests and diagnostics;
 * images never depend on them).  INTEGRATION.md carries the same table. */
CRH_API const char* crh_env_table(void);
CRH_API int crh_set_path_budget(crh_ctx* ctx, uint64_t max_paths);
/* the budget in force (the default, CRH_MAX_PATHS or the last crh_set_path_budget): a host that lowers it for a while restores THIS value, not a constant */
CRH_API int crh_get_path_budget(crh_ctx* ctx, uint64_t* max_paths);
/* Round 6: the frame kernel's feeder count (wavefronts of a workgroup that shade and generate instead of tracing) is chosen by measurement after every crh_build
 * unless CRH_FRAME_FEED fixes it -- no image depends on it.  out[0] = tuning enabled, out[1] = the count chosen (0: still measuring), out[2] = frames measured
 * so far, out[3] / out[4] = mean frame-kernel time in microseconds with 3 / 4 feeders. */
CRH_API int crh_get_frame_tuning(crh_ctx* ctx, uint32_t out[5]);
/* Round 6: the order in which crh_render lists the image's tiles (the frame kernel claims them in that order: for a host that waits for its frames the tiles
 * that cost the most rays in the last accumulation first, else row-major; CRH_TILE_ORDER=0: always row-major).  No pixel depends on it.  order: n_tiles ids or NULL; counts (or NULL): how often the sorted list has been replaced, crh_render calls that
 * used the sorted list, calls that used the row-major one, frames whose accumulate added to the per-tile sums, the verdict of the library's own measurement
 * on this scene (0 measuring, 1 the sorted list pays, 2 it does not: row-major), and the two measured means in microseconds (sorted, row-major). */
CRH_API int crh_get_tile_order(crh_ctx* ctx, uint32_t* order, uint32_t* n_tiles, uint64_t counts[7]);
/* Per-tile error estimate (mean standard error of the pixel luminance) and per-tile sample count; pass NULL
 * arrays to query n_tiles.  Needs adaptive mode for a meaningful error. */
CRH_API int crh_get_tile_stats(crh_ctx* ctx, float* err, uint32_t* counts, uint32_t* n_tiles);
/* wait for all queued device work of this context */
CRH_API int crh_sync(crh_ctx* ctx);
/* == BufferDump(Graphic3d_BT_RGB_RayTraceHdrLeft -> ImgRGBF) (AppGui.cxx:345-349):
 * W*H*3 linear float, row 0 = top */
CRH_API int crh_read_hdr(crh_ctx* ctx, float* rgb_out);
/* == BufferDump(Graphic3d_BT_RGB) (AppViewer.cxx:1259-1261): W*H*3 uint8 after exposure,
 * tone map, gamma 2.2 */
CRH_API int crh_read_ldr(crh_ctx* ctx, uint8_t* rgb_out);
/* The same image without stalling the render loop.  The reference never reads pixels back to show them: ImGui draws the FBO's colour
 * texture (AppViewer.cxx:1099) and the GL driver pipelines that behind the next Redraw().  A host of this boundary that displays
 * every frame does the equivalent with two calls: crh_read_ldr_begin() queues tone map + device-to-host copy of the frame AS
 * SUBMITTED SO FAR on a stream of its own (pinned staging, two buffers) and returns at once; rendering calls made afterwards run
 * concurrently -- only their first accumulation waits until the tone map has read the accumulator; crh_read_ldr_end() waits for
 * the OLDEST begun read-back and copies its W*H*3 bytes out.  At most two may be in flight.  The bytes are those crh_read_ldr
 * would have returned at the moment of crh_read_ldr_begin(). */
CRH_API int crh_read_ldr_begin(crh_ctx* ctx);
CRH_API int crh_read_ldr_end(crh_ctx* ctx, uint8_t* rgb_out);
/* The same for the linear HDR image == BufferDump(Graphic3d_BT_RGB_RayTraceHdrLeft) (AppGui.cxx:345-349) without stalling the render loop: W*H*3 floats as
 * crh_read_hdr would have returned them at the moment of crh_read_hdr_begin().  LDR and HDR read-backs share the two slots in flight and complete in
 * the order they were begun: the oldest one must be ended with the call of its own kind. */
CRH_API int crh_read_hdr_begin(crh_ctx* ctx);
CRH_API int crh_read_hdr_end(crh_ctx* ctx, float* rgb_out);
/* Accumulator checkpoint / resume (SURVEY.md section 5 "checkpoint / resume", 8f rank 4; the reference only keeps the
 * image while paused, AppViewer.cxx:916-920,1045): copy out / restore the float4 accumulator (rgb running mean + per-pixel
 * sample count) together with the whole-frame iteration counter that selects the next frame seed. */
CRH_API int crh_save_accum(crh_ctx* ctx, float* rgba_out /* W*H*4 */, uint32_t* frames_done);
CRH_API int crh_load_accum(crh_ctx* ctx, const float* rgba /* W*H*4 */, uint32_t frames_done);
/* Device address of the float4 accumulator (W*H*4 floats: rgb running mean, a = number of
 * samples accumulated in that pixel) so a host process can hand it to RCCL without a copy.
 * Replaces the zero-copy GL texture id the GUI displays (AppViewer.cxx:1099). */
CRH_API int crh_accum_device_ptr(crh_ctx* ctx, void** dev_ptr, uint64_t* n_bytes);
/* The exchange step of the tile-sharded multi-GPU render (the reference is single-GPU, its frame is
 * simply the FBO of AppViewer.cxx:1099): sums the float4 accumulators of `n` contexts -- one per GPU, each holding only
 * its own tiles and zeros elsewhere, so the sum is exact -- into an assembled frame on ctxs[root].  Contexts on distinct
 * devices: one grouped ncclReduce over RCCL / xGMI (librccl is loaded on first use); contexts sharing a device, or a
 * process without RCCL: peer copies + adds on the root.  Until the next crh_render* / crh_reset / crh_load_accum of the
 * root, its crh_read_hdr / crh_read_ldr / crh_save_accum return the assembled frame; no context's own accumulator is
 * modified, so rendering continues (progressive display reduces every few iterations).  One host thread calls it after
 * the per-context render threads have returned.  (One process per GPU instead: hand crh_accum_device_ptr to the
 * process group's reduce -- bench.py does that through torch.distributed.) */
CRH_API int crh_reduce(crh_ctx* const* ctxs, uint32_t n, uint32_t root);
/* counters since the last crh_reset; collecting node/triangle counters needs
 * crh_enable_counters(ctx, 1) (slower kernels) */
CRH_API int crh_enable_counters(crh_ctx* ctx, int on);
CRH_API int crh_get_stats(crh_ctx* ctx, crh_stats* out);

/* --- kernel-level entry points (parity tests and micro-benchmarks) ------------------ */
/* Trace n rays {ox,oy,oz,tmax, dx,dy,dz,tmin-unused} (8 floats each) against the built scene.
 * nearest: out_hit = n x {t, u, v, prim-id-as-int-bits}; prim = -1 on miss, t = tmax.
 * any    : out_vis = n x uint32 (1 = unoccluded up to tmax). */
CRH_API int crh_trace_nearest(crh_ctx* ctx, const float* rays, uint32_t n, float* out_hit);
CRH_API int crh_trace_any(crh_ctx* ctx, const float* rays, uint32_t n, uint32_t* out_vis);
/* Copy out the built QBVH: nodes (16 dwords = 64 B each, layout in crh_bvh_format.h) and the leaf-ordered triangle
 * records (12 floats = 48 B each; n_tris = leaf positions in use: a two-level scene holds a second, object-tree copy of every object that
 * was dragged out of the static tree, and all-zero vertices where such an object's triangles are disabled).  Pass NULL buffers to query the counts. */
CRH_API int crh_get_bvh(crh_ctx* ctx, float* nodes, uint32_t* n_nodes, float* tris, uint32_t* n_tris);
/* Two-level scenes: index of the top-level root in the node array (0 when no object is an instance), number of objects rendered as
 * instances right now, number of nodes in front of the top-level tree (static tree + object trees built so far). */
CRH_API int crh_get_tlas(crh_ctx* ctx, uint32_t* root, uint32_t* n_instances, uint32_t* n_blas_nodes);
/* Host-only: run the BVH builder (no device needed) and copy out nodes / leaf-ordered triangles.
 * Call with NULL outputs to get the counts.  == BVH_BinnedBuilder + CollapseToQuadTree. */
CRH_API int crh_build_bvh_host(const float* pos, uint32_t n_vertices, const int32_t* tri, uint32_t n_triangles, int threads,
                       float* nodes, uint32_t* n_nodes, uint32_t* prim_order);
/* Device-resident micro-benchmark: trace the same device ray buffer `repeat` times, return
 * average kernel milliseconds measured with HIP events on the context's stream. */
CRH_API int crh_bench_trace(crh_ctx* ctx, const float* rays, uint32_t n, int any_hit, uint32_t repeat,
                    float* avg_ms);
/* Average duration (ms) of the dominant kernel (nearest-hit traversal) launches since the last
 * crh_reset, measured with HIP events on the launch stream when timing is enabled. */
/* Test hook: evaluate an elementary function of include/crh_math.h on the device, n elements
 * (fn: 0 sincos2pi, 1 exp, 2 log, 3 pow(a,b), 4 acos, 5 atan2(a,b), 6 sincos, 7 sqrt, 8 a/b, 9 rng stream
 * of seed (a-bits, b-bits)).  The parity tests require bit equality with the CPU build. */
CRH_API int crh_debug_math(crh_ctx* ctx, int fn, const float* a, const float* b, float* out, float* out2, uint32_t n);
/* Test hook: the layered BSDF functions of the shading kernel on caller-supplied directions in the local frame (z = shading
 * normal), n items, so that the analytic known-answer tests run on the gfx950 code itself (reference input contract:
 * Graphic3d_BSDF, MaterialEditor.cxx:281-338).  a = wo (3 floats per item; fn 3: a[3i] = cos theta).
 *   fn 0  b = wi;                                  out[3i..] = f(wo, wi) * cos
 *   fn 1  b = wi;                                  out[i]    = pdf(wo -> wi) with path weight (1,1,1)
 *   fn 2  b[3i] = rng state (uint bits), b[3i+1] != 0: inside a medium;
 *                                                  out[8i..] = wi.xyz, weight.xyz, flags (1 alive | 2 delta | 4 inside after), rng after
 *   fn 3  b unused;                                out[3i..] = Fresnel(a[3i], m->FresnelCoat) */
CRH_API int crh_debug_bsdf(crh_ctx* ctx, int fn, const crh_bsdf* m, const float* a, const float* b, float* out, uint32_t n, int two_sided);
/* TEST HOOK, not for hosts: crh_reduce's RCCL branch (communicator bookkeeping, one group of ncclReduce calls on the contexts' streams, the assembled
 * frame) executed on contexts that share one device -- a 1-GPU pool cannot run that branch otherwise.  Same result as crh_reduce. */
CRH_API int crh_debug_reduce_fake_devices(crh_ctx* const* ctxs, uint32_t n, uint32_t root);
CRH_API int crh_enable_kernel_timing(crh_ctx* ctx, int on);
CRH_API int crh_get_kernel_timing(crh_ctx* ctx, double* trace_ms_total, uint64_t* trace_launches,
                          double* all_ms_total);
/* Camera rays of wide batches walked as wavefront packets since the last restart, and how many of them met two triangles at EXACTLY the same distance and
 * were handed to the per-ray fall-back pass (k_packets.h).  A soup has none; tessellated CAD surfaces (shared edges, coincident faces) have some: the
 * figure bench.py's CAD1M leg reports.  Diagnostics: not part of crh_stats, never compared with the oracle (which has no packets). */
CRH_API int crh_get_packet_stats(crh_ctx* ctx, uint64_t* packet_rays, uint64_t* fallback_rays);

#ifdef __cplusplus
}
#endif
#endif /* CADRAYS_HIP_H */
