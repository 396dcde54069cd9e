/*
 * mcpt.h -- C ABI of libmcpt.so, the MI355X-native (gfx950 / HIP) replacement for the hot path of
 * Arieys/MonteCarloPathTracing:  render_scene -> generateImg -> ray_intersect/bvh_intersect -> shade.
 *
 * The reference has no FFI or plugin interface; its boundary is three free C++ functions.  Each entry
 * point below names the reference function it replaces (paths relative to the reference checkout):
 *
 *   mcpt_render_scene      <->  bool render_scene(std::string path, std::string filename, int N)   MTPC/MTPC.cpp:35-68
 *   mcpt_scene_load        <->  scene_data::read_scene + sort(compare) + BVH::BVH                 MTPC/MTPC.cpp:38-45,
 *                                                     MTPC/sceneManagement.cpp:264-274, MTPC/BVH.cpp:37-85
 *   mcpt_render[_device]   <->  void generateImg(scene_data&, BVH&, image&, int)                  MTPC/pathTracing.cpp:274-331
 *   mcpt_trace_closest[_device] <-> bool ray_intersect(Ray, scene_data&, BVH&, intersection&)      MTPC/pathTracing.cpp:382-390
 *   mcpt_quantize_rgb8 + mcpt_write_png <-> imshow(double*, W, H, filename, N) + svpng()           MTPC/MTPC.cpp:10-33, MTPC/svpng.inc:77
 *
 * Plain pointers and sizes only; the caller owns every buffer, the library owns the opaque handles.
 * All compute entry points run hand-written HIP kernels on an MI355X; there is NO CPU fallback: without a
 * HIP device they return MCPT_ERR_NO_DEVICE.  Functions return 0 on success or a negative MCPT_ERR_* code;
 * mcpt_last_error() gives the message for the calling thread.
 *
 * Semantics are the reference's (fp64 arithmetic in the reference's operation order, no FMA contraction)
 * with the documented seams D1..D8 of DESIGN.md (counter-based RNG instead of time(NULL) engines, stable
 * Morton sort, serial sample accumulation, CRLF stripping, no virtual-child aliasing, depth cap, texture
 * clamp, Ns/Ni defaults).
 */
#ifndef MCPT_H
#define MCPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCPT_VERSION 105

#define MCPT_OK             0
#define MCPT_ERR_IO        -1   /* a scene/texture/output file could not be opened */
#define MCPT_ERR_PARSE     -2   /* malformed scene (face before usemtl, index out of range, light without material ...) */
#define MCPT_ERR_ARG       -3   /* bad argument */
#define MCPT_ERR_NO_DEVICE -4   /* no HIP device / ordinal out of range */
#define MCPT_ERR_HIP       -5   /* a HIP runtime call failed */
#define MCPT_ERR_NOMEM     -6

#define MCPT_MAX_DEPTH 64       /* shade() recursion cap (D6) */

typedef struct mcpt_scene  mcpt_scene;    /* host: scene_data + Morton-sorted faces + implicit BVH */
typedef struct mcpt_device mcpt_device;   /* one GPU's resident copy of a scene (SoA/record arrays in HBM) */

/* MTPC/BVH.h:29 (t, Nv, Nr, Nc, Lc, Lv, Level) */
typedef struct { int32_t t, Lc, Lv, Nc, Nv, Nr, Level; } mcpt_bvh_info;

typedef struct {
    int32_t num_faces, num_materials, num_lights, width, height;
    double eye[3], look_at[3], up[3], fovy;          /* MTPC/sceneManagement.h:150-156 */
    mcpt_bvh_info bvh;
} mcpt_scene_info;

typedef struct {
    uint64_t rays_primary, rays_shadow, rays_bounce; /* closest-hit queries actually traced */
    uint64_t node_visits, tri_tests;                 /* box tests / triangle tests executed */
    uint64_t shade_calls, samples;
    uint64_t shadow_skipped;                         /* shadow rays the reference traces although it never uses their answer
                                                        (light behind the surface, pathTracing.cpp:217); not traced here */
    uint64_t dom_rays, dom_node_visits, dom_tri_tests; /* the share of the above done inside the dominant kernel (k_wf_trace):
                                                        numerator of its roofline */
    double   ms_trace, ms_total;                     /* device time of the dominant kernel / whole call (HIP events) */
    int32_t  launches;                               /* launches of the dominant kernel */
    int32_t  max_depth;
} mcpt_stats;

typedef struct {
    int32_t  spp;            /* N_ray_per_pixel */
    uint64_t seed;           /* RNG seam key (D1) */
    /* pixel ownership: the frame is cut into tile_w x tile_h tiles; tile (tx, ty) belongs to rank
     * (tx + shift*ty) mod world, shift = first integer >= world/2 coprime with world (round-robin along a row, every
     * row starting on another rank).  world<=1 renders everything.  tile_w/h <= 0 -> 32 x 8. */
    int32_t  rank, world, tile_w, tile_h;
    int32_t  flags;          /* MCPT_RENDER_* */
} mcpt_render_params;

#define MCPT_RENDER_DEFAULT      0
#define MCPT_RENDER_MEGAKERNEL   2   /* one lane per camera sample, whole path in one kernel, reference-shaped walk
                                       (the first implementation; kept for A/B runs).  Default: wavefront pipeline. */
/* A renderer that produces a SEQUENCE of frames (mcpt_render_device only):
 *   MCPT_RENDER_KEEP_STATS  the frame's statistics stay on the device (counters accumulate, event pairs are recorded, nothing is
 *                           read back) until mcpt_device_collect_stats: the call returns without waiting for the frame.
 *   MCPT_RENDER_PIPELINE    consecutive frames use the device's two frame slots (path state, radiance, counters) in turn.  Called
 *                           on alternating streams, the latency-bound tail of frame i (the last long paths, the fold) then overlaps
 *                           the head of frame i+1.  The caller gives every frame in flight its own d_img. */
#define MCPT_RENDER_KEEP_STATS   4
#define MCPT_RENDER_PIPELINE     8

/* ---- general ---- */
int         mcpt_version(void);
const char* mcpt_last_error(void);
int         mcpt_device_count(void);                        /* number of HIP devices (0 without a GPU) */
const char* mcpt_build_id(void);                            /* 16 hex digits: hash of the sources this library was compiled from */
/* Which HIP runtime this process runs on (since 105).  libmcpt.so's kernels are built by one hipcc; the libamdhip64.so they run on is
 * whichever the process loaded first under that soname (a Python process that imported torch has the wheel's bundled copy).  The first
 * mcpt_device_create / mcpt_multi_create compares hipRuntimeGetVersion() with the HIP_VERSION the library was compiled against and
 * returns MCPT_ERR_HIP, naming both and the runtime's file, when major.minor differ -- unless mcpt_allow_runtime_mismatch(1) was called
 * (or MCPT_ALLOW_RUNTIME_MISMATCH=1 is set), which turns the refusal into one line on stderr.
 *   mcpt_hip_runtime_info : versions encoded as HIP_VERSION (major * 10^7 + minor * 10^5 + patch); path = the file the runtime was loaded from
 *   mcpt_hip_runtime_check: the comparison itself, a pure function (0 = compatible; msg receives the refusal's text) */
/* Every environment variable the library reads (since 105), one per line: "NAME | default | meaning".  The environment is parsed when a
 * device (or multi-device) handle is created and kept in the handle -- never inside a launch -- so two handles created under different
 * settings keep them; INTEGRATION.md section 7 is this table. */
const char* mcpt_knobs_describe(void);
int         mcpt_hip_runtime_info(int32_t* compiled, int32_t* runtime, char* path, int64_t cap);
int         mcpt_hip_runtime_check(int32_t compiled, int32_t runtime, const char* runtime_path, char* msg, int64_t cap);
void        mcpt_allow_runtime_mismatch(int32_t allow);

/* ---- scene (host) ---- */
/* Reads <path><filename>.obj/.mtl/.camera exactly like read_scene; textures named by map_Kd are looked up
 * relative to <path> first, then the cwd (reference: cwd only). */
int  mcpt_scene_load(const char* path, const char* filename, mcpt_scene** out);
/* The same with opt-in departures from the reference's reader; load_flags = 0 is mcpt_scene_load.
 * The reference's read_obj (MTPC/sceneManagement.cpp:76-189) takes the 2nd index of a face corner as the normal and the
 * 3rd as the texture coordinate, reads triangles only, ignores mtllib, and its Morton domain is the fixed cube [-1,4]^3
 * (MTPC/morton code.h:6-7). */
#define MCPT_LOAD_STANDARD_OBJ   1   /* corners v, v/vt, v//vn, v/vt/vn as the OBJ format defines them; relative (negative)
                                        indices; polygons fan-triangulated; any run of blanks separates fields; a corner
                                        without vn gets the face normal, without vt (0,0) */
#define MCPT_LOAD_MTLLIB         2   /* read the .mtl files named by the .obj's mtllib lines (next to the .obj);
                                        <filename>.mtl only when it names none */
#define MCPT_LOAD_MORTON_BOUNDS  4   /* Morton keys on the scene's bounding box (changes the leaf order, i.e. which of two
                                        equidistant triangles wins a tie; everything else is unchanged) */
int  mcpt_scene_load_ex(const char* path, const char* filename, int32_t load_flags, mcpt_scene** out);
/* Lifetime: a device (or mcpt_multi) created from a scene shares ownership of it -- the scene's memory goes with the last of
 * mcpt_scene_free and the mcpt_device_free / mcpt_multi_free of everything created from it, in any order.  After mcpt_scene_free the
 * caller's handle must not be passed to the library again. */
void mcpt_scene_free(mcpt_scene*);
/* The same scene_data from arrays instead of files (generated scenes: the 10 M-triangle stress scene would be ~1 GB of
 * .obj text).  Faces are given in the order the .obj would list them; Face::norm, Morton keys, per-material face
 * lists and light tables are derived exactly as read_obj / shade do. */
typedef struct {
    int64_t num_faces;
    const double*  v;               /* [num_faces][9]  v1 v2 v3 */
    const double*  vn;              /* [num_faces][9]  vn1 vn2 vn3 */
    const double*  vt;              /* [num_faces][6]  vt1 vt2 vt3, may be NULL (zeros) */
    const int32_t* material;        /* [num_faces] */
    int32_t num_materials;
    const double*  material_rec;    /* [num_materials][8]  Kd xyz, Ks xyz, Ns, Ni */
    const char* const* material_names;   /* may be NULL */
    int32_t num_lights;
    const int32_t* light_material;  /* [num_lights] material that emits (".camera" mtlname lines, in order) */
    const double*  light_radiance;  /* [num_lights][3] */
    double eye[3], look_at[3], up[3], fovy;
    int32_t width, height;
} mcpt_scene_desc;
#define MCPT_SCENE_DEFER_BUILD 1    /* leave Morton sort + BVH to the GPU (mcpt_device_create then builds on the device) */
int  mcpt_scene_create(const mcpt_scene_desc*, int32_t flags, mcpt_scene** out);
int  mcpt_scene_set_resolution(mcpt_scene*, int32_t width, int32_t height);   /* overrides .camera width/height */
int  mcpt_scene_get_info(const mcpt_scene*, mcpt_scene_info* out);
/* faces in .obj order: 27 doubles each = v1 v2 v3 vn1 vn2 vn3 (xyz) vt1 vt2 vt3 (uv) norm; any pointer may be NULL */
int  mcpt_scene_get_faces(const mcpt_scene*, double* geom27, int32_t* material, uint32_t* morton);
int  mcpt_scene_get_leaf_order(const mcpt_scene*, int32_t* leaf_to_face);     /* sorted (leaf) index -> .obj index */
/* Nr nodes in the reference's compact level order; box6 = max_x,max_y,max_z,min_x,min_y,min_z (sceneManagement.h:165-171) */
int  mcpt_scene_get_bvh_nodes(const mcpt_scene*, double* box6, int32_t* level, int32_t* leaf_face);
int  mcpt_scene_find_index(const mcpt_scene*, int32_t i, int32_t l);          /* BVH::findIndex, MTPC/BVH.cpp:99-104 */
/* rec8 = kd xyz, ks xyz, Ns, Ni; flags4 = has_map, map_width, map_height, light index or -1 */
int  mcpt_scene_get_material(const mcpt_scene*, int32_t m, char name[64], double rec8[8], int32_t flags4[4]);
int  mcpt_scene_get_light(const mcpt_scene*, int32_t i, char name[64], double radiance[3], int32_t* material, double* total_area);
uint32_t mcpt_morton_code(float x, float y, float z);                         /* getMortonCode, MTPC/morton code.cpp:22-32 */

/* Diagnostic: builds the fast closest-hit hierarchy (accel_build.cpp) on the host and reports its shape.
 * leaf_order[num_faces] (may be NULL) = reference leaf index held by every slot of its triangle list;
 * nesting_ok = 1 when every stored child box contains everything below it (what the culling argument needs). */
int  mcpt_scene_fast_bvh_stats(const mcpt_scene*, int32_t* n_nodes, int32_t* max_depth, int32_t* leaf_order, int32_t* nesting_ok);

/* ---- device ---- */
int  mcpt_device_create(const mcpt_scene*, int32_t device_ordinal, mcpt_device** out);
/* Where sort(scene.f, compare) + BVH::BVH (MTPC/MTPC.cpp:44-45) run: on the host (bvh_build.cpp) or on the GPU
 * (build_kernels.hip: Morton kernel, stable radix sort of (key, face), leaf records, one union kernel per level).
 * Both give bit-identical arrays; mcpt_device_get_* read the device's copy back for that comparison. */
#define MCPT_BUILD_HOST   0
#define MCPT_BUILD_DEVICE 1
/* MCPT_BUILD_DEVICE plus most of the fast walk's culling hierarchy built on the GPU: triangles sorted by a 63-bit Morton code on
 * the scene's bounds, every four consecutive ones under one compressed node (each triangle with its own box); only the tree over
 * those clusters (a quarter of the primitives) is built by the host's SAH builder.  2 s instead of 4 for a 10 M-triangle scene,
 * 1.3-1.5x the node visits per ray; results are identical (the hierarchy only culls). */
#define MCPT_BUILD_DEVICE_FAST 2
/* MCPT_BUILD_DEVICE with the culling hierarchy grown on the GPU at close to the quality of the host's SAH build: parallel
 * locally-ordered clustering (each cluster merges with the neighbour in Morton order that gives the smallest joint surface area) up to
 * subtrees of 4096 triangles, of bounded height and bounded box area, each collapsed on the GPU into compressed 4-wide nodes with
 * leaves of up to four triangles; the host's SAH builder only sees the clusters' boxes (3 272 for 10 M triangles).  0.8 s instead of
 * 2.1 for a 10 M-triangle scene, 7-12 % more walk time than on the host's tree; results are identical (the hierarchy only culls). */
#define MCPT_BUILD_DEVICE_SAH 3
int  mcpt_device_create_ex(const mcpt_scene*, int32_t device_ordinal, int32_t build_mode, mcpt_device** out);
int  mcpt_device_get_bvh_nodes(mcpt_device*, double* box6 /* Nr*6, may be NULL */, int32_t* leaf_face /* Nr, may be NULL */);
int  mcpt_device_get_leaf_order(mcpt_device*, int32_t* leaf_to_face);
void mcpt_device_free(mcpt_device*);
/* Which walk the closest-hit queries use.  Both return identical results (tests/test_gpu_parity.py).
 *   MCPT_TRACE_FAST (default): SAH hierarchy over the reference's leaf boxes, conservative culling, distance pruning,
 *                              the reference's own fp64 leaf-box / triangle tests on every candidate;
 *   MCPT_TRACE_REFERENCE     : the reference's implicit Morton tree in the reference's visiting order
 *                              (bvh_intersect, MTPC/pathTracing.cpp:334-374), no pruning. */
#define MCPT_TRACE_FAST      0
#define MCPT_TRACE_REFERENCE 1
int  mcpt_device_set_trace_mode(mcpt_device*, int32_t mode);
/* Which closest-hit engine MCPT_TRACE_FAST runs (same tests on the same triangles, identical results): the voting engine (one ray per
 * lane in registers, csrc/trace_persistent.hpp) or the pool engine (the rays of a workgroup resident in LDS, csrc/trace_pool.hpp).  The
 * library picks by scene size -- the pool engine where the hierarchy stays in the caches and the walk is bound by instruction issue
 * (at most MCPT_POOL_MAX_TRIS triangles, default 131072) -- unless the environment says MCPT_TRACE_ENGINE=vote or =pool.
 * Returns what a device created for this scene now would use. */
#define MCPT_ENGINE_VOTE 0
#define MCPT_ENGINE_POOL 1
int  mcpt_scene_trace_engine(const mcpt_scene*);

/* ---- closest hit (ray_intersect) ---- */
/* rays: n x 6 doubles (origin xyz, direction xyz).  face[n] = .obj face index or -1, t[n], p[n*3], pn[n*3];
 * any output of the host-pointer form may be NULL.  It stages through HBM; the _device form takes device pointers
 * (d_pn may be NULL, the others are required) and a hipStream_t (NULL = default stream) and is asynchronous. */
int  mcpt_trace_closest(mcpt_device*, const double* rays, int64_t n, int32_t* face, double* t, double* p, double* pn, mcpt_stats* stats);
int  mcpt_trace_closest_device(mcpt_device*, const double* d_rays, int64_t n, int32_t* d_face, double* d_t, double* d_p, double* d_pn, void* stream);

/* ---- integrator (generateImg) ---- */
/* img: H*W*3 doubles, index (row*W + col)*3 + c (image::getIndex, sceneManagement.h:232-234).  Pixels this
 * rank does not own are left untouched.  stats may be NULL. */
int  mcpt_render(mcpt_device*, const mcpt_render_params*, double* img, mcpt_stats* stats);
int  mcpt_render_device(mcpt_device*, const mcpt_render_params*, double* d_img, mcpt_stats* stats, void* stream);
/* statistics of all MCPT_RENDER_KEEP_STATS frames since the last call (waits for them; counts summed, ms_trace = sum over
 * k_wf_trace launches, ms_total = sum of the frames' own durations -- pipelined frames overlap) */
int  mcpt_device_collect_stats(mcpt_device*, mcpt_stats* stats);
/* radiance of single camera samples: pix[n] = row*W+col, k[n] = sample index -> rgb[n*3] (test seam, host pointers) */
int  mcpt_sample_radiance(mcpt_device*, uint64_t seed, const int32_t* pix, const int32_t* k, int64_t n, double* rgb);
/* number of pixels owned by (rank, world) under the tile partition, and their indices (row*W+col, ascending) */
int64_t mcpt_owned_pixels(const mcpt_scene*, const mcpt_render_params*, int32_t* pixels /* may be NULL */);

/* ---- integrator over several GPUs of one node (no reference counterpart: generateImg is single-process OpenMP) ---- */
/* The frame is cut into tiles dealt to the GPUs exactly as mcpt_render_params.rank/world describe (rank r = devices[r]); the scene
 * is resident on every GPU; one host thread per GPU renders its tiles; at the end of the frame every rank's pixels travel as one
 * compact buffer into devices[0]'s HBM and are put at their frame positions there.  Every (pixel, sample) owns its RNG key, so the
 * frame is bit-identical for any number of GPUs (and to mcpt_render).  An ordinal may appear more than once in devices[] (several
 * ranks sharing a GPU: how the exchange is tested on a one-GPU box; MCPT_GATHER_PEER only). */
typedef struct mcpt_multi mcpt_multi;
#define MCPT_GATHER_PEER 0   /* hipMemcpyPeerAsync into devices[0] (over xGMI between the GPUs of a node) */
#define MCPT_GATHER_RCCL 1   /* ncclSend / ncclRecv in one group (librccl.so is loaded at mcpt_multi_create); distinct ordinals only */
/* devices == NULL or num_devices <= 0: every visible GPU.  build_mode as mcpt_device_create_ex (MCPT_BUILD_HOST needs a host build). */
int  mcpt_multi_create(const mcpt_scene*, const int32_t* devices, int32_t num_devices, int32_t build_mode, int32_t gather, mcpt_multi** out);
int  mcpt_multi_num_devices(const mcpt_multi*);
/* generateImg on all the GPUs: img = H*W*3 doubles on the host (every pixel is written); params->rank/world are ignored.
 * stats (may be NULL): counts summed over the GPUs, ms_total = slowest GPU + exchange, ms_trace = slowest GPU's. */
int  mcpt_multi_render(mcpt_multi*, const mcpt_render_params*, double* img, mcpt_stats* stats);
/* the same, leaving the frame in devices[0]'s HBM: *d_img (owned by the handle, valid until the next call) */
int  mcpt_multi_render_device(mcpt_multi*, const mcpt_render_params*, double** d_img, mcpt_stats* stats);
/* With MCPT_RENDER_KEEP_STATS in params->flags the GPUs keep their statistics (stats is left zero; nothing is read back inside the
 * frame); mcpt_multi_collect_stats sums what has gathered since the last call over the GPUs (ms_trace / ms_total: the slowest GPU's). */
int  mcpt_multi_collect_stats(mcpt_multi*, mcpt_stats* stats);
/* How the last frame's time divides (HIP events on each GPU's own stream): render_ms[num_devices] = each rank's render;
 * *gather_ms = from devices[0]'s own render being done to the last rank's pixels being in place in its HBM; *comm_ranks = the
 * number of ranks the RCCL communicator reports (0 with MCPT_GATHER_PEER).  Any pointer may be NULL. */
int  mcpt_multi_last_timing(const mcpt_multi*, double* render_ms, double* gather_ms, int32_t* comm_ranks);
void mcpt_multi_free(mcpt_multi*);

/* ---- one process per GPU (since 105; no reference counterpart) ---- */
/* The same exchange between the PROCESSES of a launch: every rank is a process that drives one GPU through mcpt_device_* (params->rank /
 * world = its place in the launch), and at the end of a frame the ranks' pixels travel over an RCCL communicator into rank 0's frame.
 * Rank 0 asks mcpt_comm_unique_id for RCCL's 128-byte id and hands it to the other ranks by whatever channel the launcher offers
 * (montecarlopathtracing_amd/procs.py: a file keyed by the launcher's process id); every rank then calls mcpt_comm_create.  No torch in
 * the process: the ranks run on the HIP runtime libmcpt.so was compiled against (see mcpt_hip_runtime_check) and /opt/rocm's librccl.
 *   mcpt_comm_gather_frame: d_frame = this rank's frame on its GPU (H*W*3 doubles, only its own pixels written); on return rank 0's holds
 *                           every pixel.  tile_w / tile_h of params select the partition; stream = the stream the frame was rendered on.
 *   mcpt_comm_allreduce   : v[n <= 64] on the host, summed (op 0) or maximised (op 1) over the ranks in place; n = 0: a barrier. */
typedef struct mcpt_comm mcpt_comm;
int  mcpt_comm_unique_id(uint8_t* id, int64_t cap /* >= 128 */);          /* returns the number of bytes written (128) or an error */
int  mcpt_comm_create(int32_t device_ordinal, int32_t rank, int32_t world, const uint8_t* id, int64_t id_bytes, mcpt_comm** out);
int  mcpt_comm_size(const mcpt_comm*);                                     /* ranks the RCCL communicator reports */
int  mcpt_comm_gather_frame(mcpt_comm*, const mcpt_scene*, const mcpt_render_params*, double* d_frame, void* stream);
int  mcpt_comm_allreduce(mcpt_comm*, double* v, int32_t n, int32_t op);
void mcpt_comm_free(mcpt_comm*);

/* ---- output (imshow + svpng) ---- */
int  mcpt_quantize_rgb8(const double* img, int64_t n, uint8_t* rgb8);        /* (unsigned char)clamp(v*255,0,255) */
int  mcpt_write_png(const char* file, const uint8_t* rgb8, int32_t width, int32_t height);
int64_t mcpt_png_encode(const uint8_t* rgb8, int32_t width, int32_t height, uint8_t* out, int64_t cap);

/* Output beyond svpng.  mcpt_png_encode_deflate / mcpt_write_png_deflate: the same 8-bit RGB picture as a
 * compressed PNG (per-row filter choice + deflate).  mcpt_write_pfm: img = double[h*w*3] as generateImg leaves it, written as
 * a little-endian fp32 Portable Float Map (bottom row first), no clamp.  Checkpoints: the fp64 frame plus the finished ones
 * of `parts` tile partitions (mcpt_render_params.rank/world = part/parts); load returns MCPT_ERR_IO when there is no file
 * and MCPT_ERR_PARSE when the file belongs to another frame (size, spp, seed, parts or scene differ). */
int64_t mcpt_png_encode_deflate(const uint8_t* rgb8, int32_t width, int32_t height, uint8_t* out, int64_t cap);
int  mcpt_write_png_deflate(const char* file, const uint8_t* rgb8, int32_t width, int32_t height);
int  mcpt_write_pfm(const char* file, const double* img, int32_t width, int32_t height);
int  mcpt_checkpoint_save(const char* file, const mcpt_scene* scene, const double* img, int32_t spp, uint64_t seed, int32_t parts, const uint8_t* done);
int  mcpt_checkpoint_load(const char* file, const mcpt_scene* scene, double* img, int32_t spp, uint64_t seed, int32_t parts, uint8_t* done);

/* Texture input (what the reference gets from cv::imread, MTPC/sceneManagement.h:137): decodes a baseline or progressive
 * JFIF file into an 8-bit BGR raster (rows x cols x 3).  With bgr == NULL only the size is returned. */
int  mcpt_decode_jpeg(const char* file, int32_t* width, int32_t* height, uint8_t* bgr, int64_t cap);

/* ---- whole program (render_scene) ---- */
/* Reads <path><filename>.*, renders with N samples per pixel on GPU 0 (or the GPUs named by the options) and writes
 * "../result/<filename>-SPP<N>.png" relative to the cwd, like the reference. */
int  mcpt_render_scene(const char* path, const char* filename, int32_t spp);
#define MCPT_OUT_PNG_DEFLATE  1      /* the .png is deflate-compressed (same pixels; the reference's svpng stores them raw) */
#define MCPT_OUT_PFM          2      /* also write <prefix>-SPP<N>.pfm: the linear fp32 radiance before imshow's clamp */
typedef struct {
    uint64_t seed;
    int32_t  device;            /* HIP ordinal */
    int32_t  width, height;     /* >0 overrides the .camera resolution */
    int32_t  quiet;             /* suppress the reference-style progress prints */
    const char* output_prefix;  /* NULL -> "../result/<filename>"; file = <prefix>-SPP<N>.png */
    /* since MCPT_VERSION 101 (all zero = the reference's behaviour): */
    int32_t  load_flags;        /* MCPT_LOAD_* */
    int32_t  output_flags;      /* MCPT_OUT_* */
    const char* checkpoint;     /* a file: the frame is rendered in checkpoint_parts tile partitions, the fp64 frame is saved
                                   after each, and a run that finds a matching file resumes after the partitions it holds */
    int32_t  checkpoint_parts;  /* 0 -> 8 */
    int32_t  reserved;
    /* since MCPT_VERSION 102: the frame on several GPUs of the node (mcpt_multi_*); all zero = one GPU (`device`) */
    int32_t  num_devices;       /* > 0: devices[0..num_devices); -1: every visible GPU */
    int32_t  gather;            /* MCPT_GATHER_* */
    const int32_t* devices;     /* NULL with num_devices > 0: ordinals 0..num_devices-1 */
} mcpt_render_scene_options;
/* The struct has grown with MCPT_VERSION and carries no size field.  mcpt_render_scene_ex -- the only entry point through version
 * 102 -- reads the struct as it stood at 102, i.e. every field above: what a caller sets through it (load_flags, checkpoint,
 * num_devices ...) is honoured, never silently dropped; a caller compiled against a 100 / 101 header must pass a zero-extended
 * struct of this size.  mcpt_render_scene_opts (since 103) takes sizeof(mcpt_render_scene_options) as the caller's header defines
 * it and reads exactly that many bytes: the entry point for any field added after 102, and the safe one for older headers. */
int  mcpt_render_scene_ex(const char* path, const char* filename, int32_t spp, const mcpt_render_scene_options*, mcpt_stats* stats);
int  mcpt_render_scene_opts(const char* path, const char* filename, int32_t spp, const mcpt_render_scene_options*, int64_t options_bytes, mcpt_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* MCPT_H */
