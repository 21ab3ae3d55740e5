/*
 * rtgs_hip.h -- C ABI of the MI355X (gfx950) ray-traced Gaussian-splat render path.
 *
 * This is the drop-in boundary for ONE hot path of fangjunzhou/rt-gaussian-splat-renderer
 * (reference @ 2025-06-29): camera ray generation -> quaternion+scale -> covariance ->
 * ray/ellipsoid intersection -> per-ray depth sort -> front-to-back alpha compositing.
 * The reference has no FFI of its own: its operator boundary is "Python host object ->
 * @ti.kernel launch".  Each entry point below names the reference kernel / @ti.func it
 * replaces (paths relative to the reference's src/rtgs/).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in the signatures
 *     (rtgs_stream_t is a hipStream_t passed as void*; NULL = the default stream).
 *   - every pointer is a DEVICE pointer on the current device unless it is a small
 *     parameter struct (RtgsScene / RtgsCamera / RtgsTraceParams: host structs, by pointer).
 *   - the caller owns every buffer; the library never allocates, copies to the host or
 *     synchronises (graph-capture safe).  Work is ordered on the caller's stream.
 *   - stateless and re-entrant; return 0 on success, a negative RTGS_E* code otherwise.
 *   - quaternions are scalar-last (x, y, z, w) (utils/quaternion.py); image buffers are
 *     indexed [i][j], i = x in [0, W), j = y in [0, H), (0,0) bottom-left, j fastest in
 *     memory (the (W, H) Taichi field of camera.py:29 / ray_tracer.py:33-37).
 *   - fp32 in, fp32 out; indices int32.
 */
#ifndef RTGS_HIP_H
#define RTGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGS_ABI_VERSION 7
#define RTGS_MAX_DEPTH 16        /* peels kept per ray in one pass (__main__.py:70 default) */
#ifndef RTGS_TILE
#define RTGS_TILE 16             /* macro-tile edge in pixels: one workgroup, the unit of sharding */
#endif
#define RTGS_RAY_FLOATS 8        /* Ray = origin3, direction3, start, end (ray.py:4-18) */
#define RTGS_SH_COEFFS 15        /* sh_10 .. sh_36 (gaussian.py:41-55) */

enum {
  RTGS_OK = 0,
  RTGS_EINVAL = -1,      /* NULL pointer, non-positive size, bad enum */
  RTGS_EDEPTH = -2,      /* depth < 1, or > RTGS_MAX_DEPTH without the per-pixel state buffers */
  RTGS_EWORKSPACE = -3,  /* workspace smaller than rtgs_workspace_bytes() */
  RTGS_ELAUNCH = -4,     /* hipGetLastError() after a launch was not hipSuccess */
  RTGS_EDEVICE = -5      /* no usable gfx950 device / runtime error */
};

typedef void* rtgs_stream_t;

/* Gaussians, structure-of-arrays in HBM.  Values are post-activation, i.e. exactly what
 * Gaussian.init takes (gaussian.py:59-84): what Scene.load_file produces at scene.py:110-114. */
/* The kernels rebuild Sigma^-1 = R S^-2 R^T in fp64 from (rotation, scale) wherever they need it:
 * an inverse covariance ROUNDED to fp32 moves delta/(4A) by ~2e-4 for a 70:1 needle Gaussian,
 * enough to flip grazing hits against the fp64 oracle (found by tools/fuzz_gpu.py). */
typedef struct RtgsScene {
  int32_t n;                  /* 1 .. 2^26 (67 M) for the render path: RTGS_EINVAL beyond */
  const float* position;      /* [n][3]  Gaussian.position                         */
  const float* rotation;      /* [n][4]  Gaussian.rotation, xyzw                   */
  const float* scale;         /* [n][3]  Gaussian.scale                            */
  const float* color_opacity; /* [n][4]  Gaussian.color rgb, Gaussian.opacity      */
  const float* sh;            /* [n][15][3] sh_10..sh_36, or NULL (all zero)       */
} RtgsScene;

/* Camera(position, rotation, buf_size, focal_length), camera.py:17-29. */
typedef struct RtgsCamera {
  float position[3];
  float rotation[4];          /* xyzw */
  int32_t width, height;      /* buf_size; sensor size == buf_size (camera.py:26) */
  float fx, fy;               /* focal_length, in pixels */
  const void* device_pose;    /* NULL, or a DEVICE buffer of rtgs_pose_bytes() bytes holding what
                                 rtgs_pose_pack() made of (position, rotation, fx, fy, width, height):
                                 the kernels then read the pose from there instead of from this
                                 struct, so that a captured hipGraph of rtgs_project + rtgs_trace can be
                                 replayed with a new pose per frame (copy the packed block in from
                                 pinned host memory inside the graph).  width/height/fx/fy of this
                                 struct must still be the real ones: they size the launch. */
} RtgsCamera;

/* flags for RtgsTraceParams.flags */
#define RTGS_TRACE_RESUME 1u      /* read per-pixel ray.start and attenuation (a partial sample
                                     in progress); otherwise start = 0, attenuation = 1      */
#define RTGS_TRACE_ACCUMULATE 2u  /* sample_buf += radiance (ray_tracer.py:95); otherwise the
                                     shard's pixels are overwritten                          */
#define RTGS_TRACE_COMPACT 4u     /* sample_buf is the shard-compact [local_tile][T][T][3], T = rtgs_tile_size(),
                                     buffer that rtgs_untile scatters, not the (W,H,3) image  */
/* (8u and 16u were RTGS_TRACE_SPLIT_ON / _OFF up to ABI v6 -- the two-waves-per-8x8-tile form of the kernel, retired
 * in round 5: the bits are ignored) */
#define RTGS_TRACE_TILE_LIST 32u  /* the shard is tile_list[0 .. tile_count) (ABI v6), not (tile_first, tile_step);
                                     tile_count = 0 (an empty share) renders nothing */
#define RTGS_TRACE_REGION 64u     /* the shard's tiles all lie inside RtgsTraceParams.region: the projection this call
                                     runs skips the Gaussians that cannot reach it (same pixels) */

/* RTGS_TRACE_BINS_OFF: this call's own projection builds no per-frame slab bins (k_bin); its tiles scan the slab bounds
 * themselves.  For experiments: a tile that scans lists its slabs in another order than its bin holds them, and the
 * arrival order decides which of two EXACTLY tied entries a ray composites -- a shard rendered with the flag and one
 * without may differ in such a pixel (everywhere else bit for bit the same).  Without the flag every projection of a
 * scene large enough for bins builds them, whatever the shard (up to round 4 shards below a sixth of the frame did not;
 * RTGS_TRACE_BINS_ON, which forced them, is accepted and ignored). */
#define RTGS_TRACE_BINS_ON 128u
#define RTGS_TRACE_BINS_OFF 256u

typedef struct RtgsTraceParams {
  int32_t depth;        /* peels performed by this call, >= 1.  Above RTGS_MAX_DEPTH the call  */
                        /* is a chain of ceil(depth/16) launches through attenuation_buf and  */
                        /* ray_start (both required then; one projection serves the chain)   */
  float t_cutoff;       /* transmittance early-out of the compositing loop; 0 = disabled  */
                        /* (the reference has none)                                       */
  int32_t tile_first;   /* this shard renders macro-tiles tile_first, +tile_step, ...     */
  int32_t tile_step;    /* (rank, world_size); (0, 1) = whole frame                       */
  uint32_t flags;
  const int32_t* tile_order; /* optional DEVICE array, a permutation of 0..local_tiles-1: workgroup b
                           renders the shard's local tile tile_order[b] (NULL = b).  Scheduling
                           only: the image and the compact layout do not depend on it.  Work-
                           groups b and b + 256 share a CU on MI355X, so an order that sorts the
                           tiles by expected cost gives every CU a mix of heavy and light tiles. */
  const int32_t* tile_list;  /* with RTGS_TRACE_TILE_LIST (ABI v6): DEVICE array of tile_count macro-tile ids, the shard
                           is exactly these tiles -- any partition of the frame, e.g. unequal shares per rank
                           (rtgs.parallel.Partition) -- and local tile l (compact slot l, tile_order's index) is
                           tile_list[l]; tile_first / tile_step are ignored.  Must be NULL without the flag. */
  int32_t tile_count;
  float region[4];      /* with RTGS_TRACE_REGION (ABI v6): a rectangle of pixel-centre offsets from the image centre,
                           (X lo, Y lo, X hi, Y hi) with X = i + 0.5 - W/2, Y = j + 0.5 - H/2, that contains every pixel
                           of the shard -- one compact part of the frame per rank (rtgs.parallel.Partition.wedges).
                           The per-frame projection of this call (projected = 0, rtgs_shard_frame, rtgs_shard_frames)
                           then treats Gaussians whose conservative screen bound misses it as culled: at 1 M
                           Gaussians an eighth-of-the-frame sector skips most of k_project's arithmetic and writes. */
} RtgsTraceParams;

int rtgs_abi_version(void);
const char* rtgs_strerror(int code);

/* Gaussian.cov (gaussian.py:86-102) with quat.as_rotation_mat3 (utils/quaternion.py:99-121):
 * Sigma = R S S^T R^T per Gaussian, and its inverse (the ti.math.inverse of gaussian.py:196,215,
 * evaluated as R S^-2 R^T), rounded to fp32 for the caller.  cov6 may be NULL.  Layout
 * xx,xy,xz,yy,yz,zz.  (The render kernels do not consume this output: see RtgsScene.) */
int rtgs_covariance(int32_t n, const float* rotation /*[n][4]*/, const float* scale /*[n][3]*/,
                    float* cov6 /*[n][6] or NULL*/, float* cov_inv6 /*[n][6]*/,
                    rtgs_stream_t stream);

/* The same under the name and argument list SURVEY.md 8(b) gives it: sigma_inv6 [n][6] only
 * (position is accepted and unused: the inverse covariance does not depend on it). */
int rtgs_precompute(int32_t n, const float* position /*[n][3]*/, const float* rotation /*[n][4]*/,
                    const float* scale /*[n][3]*/, float* sigma_inv6 /*[n][6]*/, rtgs_stream_t stream);

/* Camera.generate_ray_field (camera.py:57-71): rays[(i*H + j)*8 ..] = {o, d, 0, +inf}. */
int rtgs_generate_rays(const RtgsCamera* cam, float* rays /*[W][H][8]*/, rtgs_stream_t stream);

/* Camera pose block for RtgsCamera.device_pose: its size, and the packing of a camera into a HOST
 * buffer of that size (plain host function: no device work).  The viewer loop of the reference
 * re-generates its ray field from a new pose every frame (__main__.py:236-250, camera.py:57-71);
 * this is that pose on a path that needs no new kernel arguments. */
size_t rtgs_pose_bytes(void);
int rtgs_pose_pack(const RtgsCamera* cam, void* host_out);

/* Bytes of scratch rtgs_trace needs for n Gaussians (per-frame projected records, slab bounds, bins):
 * rtgs_workspace_bytes is enough for a frame of ANY size (it provides 64 MiB for the macro-tiles' own
 * slab lists), rtgs_workspace_bytes_frame (ABI v6) is what a width x height frame needs (8 MB of
 * lists at 1080p, 33 MB at 4K).  The calls check against the frame's figure. */
size_t rtgs_workspace_bytes(int32_t n);
size_t rtgs_workspace_bytes_frame(int32_t n, int32_t width, int32_t height);
/* (ABI v7) A workspace for `frames` consecutive frames of one render stream of rtgs_shard_frames, which then renders
 * them in ONE launch per kernel (grid.y = frames, at most rtgs_max_frames_per_launch() = 8): per frame the workspace
 * above plus a pose block, rounded to 256 B.  frames = 1 is rtgs_workspace_bytes_frame. */
size_t rtgs_workspace_bytes_frames(int32_t n, int32_t width, int32_t height, int32_t frames);
int32_t rtgs_max_frames_per_launch(void);

/* Macro-tile edge the library was built with, the number of macro-tiles of a W x H image, and
 * how many of them a shard owns. */
int32_t rtgs_tile_size(void);
int32_t rtgs_num_tiles(int32_t width, int32_t height);
int32_t rtgs_num_local_tiles(int32_t width, int32_t height, int32_t tile_first, int32_t tile_step);

/* Host only: fills order[0 .. rtgs_num_local_tiles()) (HOST memory) with a launch order for
 * RtgsTraceParams.tile_order (copy it to the device once per (size, shard)); returns the count or
 * RTGS_EINVAL.  Scheduling only -- the pixels do not depend on it.
 *   kind 0  most central tile first: workgroups b, b+256, ... share a CU, so every CU gets one tile
 *           of each cost class (perspective piles depth complexity up in the centre);
 *   kind 1  the same inside 8 equal-count image wedges, dealt out so that workgroup b -- which the
 *           dispatcher hands to XCD b % 8 -- renders a tile of wedge b % 8: each XCD's 4 MiB L2 then
 *           serves the Gaussians of one eighth of the screen (k_trace c3 0.139 -> 0.135 ms, with SH
 *           0.310 -> 0.289).  What rtgs.RayTracer and the tile-parallel renderer use. */
int32_t rtgs_launch_order(int32_t width, int32_t height, int32_t tile_first, int32_t tile_step, int32_t kind,
                          int32_t* order);
/* the same for an explicit tile list (HOST array, what RtgsTraceParams.tile_list will hold on the device) */
int32_t rtgs_launch_order_list(int32_t width, int32_t height, const int32_t* tiles, int32_t n_tiles, int32_t kind,
                               int32_t* order);

/* RayTracer.sample_step x depth (ray_tracer.py:39-54, 79-104) with Scene.hit's closest-hit
 * semantics (scene.py:433-437), Gaussian.hit (gaussian.py:203-230), Gaussian.eval / eval_sh
 * (gaussian.py:140-201) and the clear_attenuation + generate_ray_field of step 0, in one pass:
 * per pixel, the `depth` nearest entry distances t1 > ray.start are found, sorted and
 * composited front to back.
 *   sample_buf       [W][H][3]  (or compact, see RTGS_TRACE_COMPACT)
 *   attenuation_buf  [W][H]     in (RESUME) / out; may be NULL without RESUME
 *   ray_start        &rays[0][0][6] with stride 8 floats, i.e. cam_ray_field[i,j].start,
 *                    in (RESUME) / out (t1 of the last peel + 1e-8, or +inf after a miss);
 *                    may be NULL without RESUME and depth <= RTGS_MAX_DEPTH.  Exact fp32 ties in
 *                    t1: one of the tied Gaussians is composited, as in the reference (scene.py:433
 *                    strict `>` after ray_tracer.py:101), and `depth` DISTINCT entry distances
 *                    are composited whatever the ties -- a tied twin takes one of the 16 list slots
 *                    of a launch, and the (rare) ray whose list came up short that way peels on
 *                    inside the kernel (ABI v6; up to v5 such a pixel lost its last peel).
 *                    ray_start_stride in floats (8 for a Ray field, 1 for a plain [W][H] plane).
 * Runs the per-frame projection (rtgs_project) first unless `projected` is non-zero. */
int rtgs_trace(const RtgsScene* scene, const RtgsCamera* cam, const RtgsTraceParams* params,
               float* sample_buf, float* attenuation_buf, float* ray_start,
               int32_t ray_start_stride, void* workspace, size_t workspace_bytes,
               int32_t projected, rtgs_stream_t stream);

/* The per-frame half of rtgs_trace on its own: camera-relative conic records and screen
 * bounds of every Gaussian into the workspace (quaternion -> covariance -> ray/ellipsoid
 * quadratic of gaussian.py:215-219, factored per Gaussian instead of per ray). */
int rtgs_project(const RtgsScene* scene, const RtgsCamera* cam, void* workspace,
                 size_t workspace_bytes, rtgs_stream_t stream);

/* rtgs_project exactly as rtgs_trace(projected = 0) would run it for the shard (tile_first,
 * tile_step) -- since round 5 the same as rtgs_project (whole-frame bins whatever the shard); kept for callers that
 * run a shard's projection apart from its trace (rtgs_trace with projected = 1). */
int rtgs_project_shard(const RtgsScene* scene, const RtgsCamera* cam, int32_t tile_first,
                       int32_t tile_step, void* workspace, size_t workspace_bytes, rtgs_stream_t stream);

/* One-call form of a whole sample (SURVEY.md 8(b) signature): rtgs_project + rtgs_trace of the
 * macro-tile ROWS [tile_y0, tile_y1) (rows of rtgs_tile_size() pixels in j; all rows = 0 ..
 * ceil(H / tile)) into the (W,H,3) image sample_out, overwriting those rows' pixels; the other
 * pixels are not touched.  transmittance_out: optional (W,H) plane. */
int rtgs_render(const RtgsScene* scene, const RtgsCamera* cam, int32_t depth, float t_cutoff,
                int32_t tile_y0, int32_t tile_y1, float* sample_out, float* transmittance_out,
                void* workspace, size_t workspace_bytes, rtgs_stream_t stream);

/* ---- peel lists (ABI v6; v7: written at their real lengths) ----
 * The reference's viewer takes a sample as `depth` calls of RayTracer.sample (__main__.py:253-256), each
 * one peel (ray_tracer.py:39-54) that re-walks the scene for every pixel.  These three serve that call
 * pattern from ONE traversal: rtgs_trace_lists runs rtgs_trace's pass for the whole frame but, instead
 * of compositing, stores every ray's sorted DISTINCT entries (up to RTGS_MAX_DEPTH of them: a count per
 * pixel, then Gaussian index, exponent and t1 per entry) in `lists` (rtgs_peel_list_bytes(W, H) bytes,
 * 16-B aligned, caller-owned: 196 B per pixel of capacity, of which 4 B + 12 B per entry the pixel
 * really has are written -- ABI v6 wrote a terminator into every unused slot); rtgs_peel composites
 * entry k of every ray -- sample_buf += attenuation * alpha * colour, attenuation *= 1 - alpha, exactly
 * sample_step's update -- reading colours / opacities / SH from the scene at call time; rtgs_peel_start
 * writes the ray.start the first `peels_done` peels imply (t1 of the last + 1e-8, +inf after a miss, 0 for
 * none).  The lists depend on the Gaussians' geometry and the camera only: they stay valid across samples
 * while neither changes.
 *   flags of rtgs_peel: RTGS_PEEL_FRESH = the sample starts with this peel (attenuation 1 instead of the
 *   buffer's contents: the clear_attenuation of ray_tracer.py:46). */
#define RTGS_PEEL_FRESH 1u
size_t rtgs_peel_list_bytes(int32_t width, int32_t height);
int rtgs_trace_lists(const RtgsScene* scene, const RtgsCamera* cam, const int32_t* tile_order /*optional, as RtgsTraceParams*/,
                     void* lists, size_t lists_bytes, void* workspace, size_t workspace_bytes, int32_t projected,
                     rtgs_stream_t stream);
int rtgs_peel(const RtgsScene* scene, const RtgsCamera* cam, const void* lists, size_t lists_bytes, int32_t k,
              uint32_t flags, float* sample_buf /*[W][H][3] in/out*/, float* attenuation_buf /*[W][H] in/out*/,
              rtgs_stream_t stream);
int rtgs_peel_start(const RtgsCamera* cam, const void* lists, size_t lists_bytes, int32_t peels_done,
                    float* ray_start, int32_t ray_start_stride, rtgs_stream_t stream);

/* Tile-parallel epilogue: scatter gathered shard-compact buffers
 * compact[rank][local_tile][T][T][3], T = rtgs_tile_size(), (rank-major, `tiles_per_rank` tiles each, as produced
 * by RTGS_TRACE_COMPACT with (tile_first, tile_step) = (rank, world)) into image [W][H][3]. */
int rtgs_untile(const float* compact, int32_t world, int32_t tiles_per_rank, int32_t width,
                int32_t height, float* image, rtgs_stream_t stream);

/* ---- stream plumbing of the tile-parallel frame loop (rtgs.parallel.TileParallelRenderer) ----
 * The reference has no multi-device path; these keep the per-frame HOST cost of the
 * loop at two library calls + the collective.  Events are hipEvent_t (timing disabled) behind void*.
 * rtgs_event_query: 1 = complete or never recorded, 0 = pending. */
int rtgs_event_create(void** event);
int rtgs_event_destroy(void* event);
int rtgs_event_query(void* event);

/* One shard of one frame, render side: [render_stream waits for reuse_event unless it is complete
 * or NULL -- the previous reader of `out`] -> rtgs_project + rtgs_trace of the shard on render_stream
 * (as rtgs_trace with projected = 0, ray_start_stride 1; attenuation_buf / ray_start only needed for
 * depth > RTGS_MAX_DEPTH) -> traced_event recorded on render_stream -> [consumer_stream made to wait
 * for it when consumer_waits != 0; it is an argument of its own because the NULL stream is a stream].
 * The projection this call (and rtgs_shard_frames) leaves in `workspace` serves THIS shard only: its slab bins are
 * built for the shard's own tiles (and, with RTGS_TRACE_REGION, its Gaussians culled to the region).  Tracing other
 * tiles from it (rtgs_trace with projected = 1) is undefined; rtgs_project / rtgs_trace(projected = 0) keep
 * whole-frame bins for that. */
int rtgs_shard_frame(const RtgsScene* scene, const RtgsCamera* cam, const RtgsTraceParams* params,
                     float* out, float* attenuation_buf, float* ray_start, void* workspace, size_t workspace_bytes,
                     rtgs_stream_t render_stream, void* reuse_event, void* traced_event,
                     rtgs_stream_t consumer_stream, int32_t consumer_waits);

/* B frames of one shard, render side, in ONE call (ABI v7): for frame k = 0 .. n_frames-1, rtgs_project + rtgs_trace of
 * the shard seen from cams[k] (HOST array of poses) into outs[k] (HOST array of device pointers) on
 * render_streams[(first_stream + k) % n_streams], with workspaces[that stream] (and attenuation_bufs / ray_starts
 * [that stream] for depth > RTGS_MAX_DEPTH; both arrays may be NULL otherwise).  When workspace_bytes (the size of
 * EACH stream's workspace) has room for G >= 2 frames (rtgs_workspace_bytes_frames), depth <= RTGS_MAX_DEPTH and the
 * two state arrays are NULL, runs of up to G consecutive frames whose outs[] are equally spaced share ONE launch per
 * kernel (grid.y = frames: a small shard's frames fill the chip together instead of queueing as latency-bound
 * launches), group g of the call on render_streams[(first_stream + g) % n_streams]; same pixels either way.
 * Every stream used first waits for
 * reuse_event (the previous reader of the buffers behind outs[]; NULL or complete: no wait), records
 * done_events[stream] after its last frame of the call, and consumer_stream is made to wait for those events when
 * consumer_waits != 0.  The reference renders one pose per loop iteration (__main__.py:236-257); a turntable or a
 * benchmark that knows its next B poses pays one library call -- and, in rtgs.parallel, one collective -- per B frames. */
int rtgs_shard_frames(const RtgsScene* scene, const RtgsCamera* cams, int32_t n_frames, const RtgsTraceParams* params,
                      float* const* outs, float* const* attenuation_bufs, float* const* ray_starts,
                      void* const* workspaces, size_t workspace_bytes, const rtgs_stream_t* render_streams,
                      int32_t n_streams, int32_t first_stream, void* reuse_event, void* const* done_events,
                      rtgs_stream_t consumer_stream, int32_t consumer_waits);

/* Destination side, after the gather was queued on `stream`: records sent_event there (the senders'
 * buffers may be reused once it completes; may be NULL), then scatters the first n_land of `batch`
 * frames -- gathered[rank][frame][local_tile][T][T][3], every rank's block holding `batch` frames'
 * shards -- into images[0 .. n_land) (HOST array of device pointers, [W][H][3] each).  batch = 1 is
 * rtgs_untile.  n_land = 0 (ranks that are not the destination): only the event. */
int rtgs_frames_land(const float* gathered, int32_t world, int32_t tiles_per_rank, int32_t batch, int32_t n_land,
                     int32_t width, int32_t height, float* const* images, rtgs_stream_t stream, void* sent_event);

/* The landing side for ANY partition of the frame (ABI v6; RtgsTraceParams.tile_list on the rendering side):
 * rank r's buffer rank_bufs[r] (HOST array of `world` <= 16 device pointers; the destination's own shard
 * is read where it was traced) holds `batch` frames of rank_tiles[r] (HOST array) compact tiles each,
 * [frame][slot][T][T][3]; tile_map (DEVICE, one int32 per macro-tile of the frame) = rank << 24 | slot.
 * rtgs_frames_land_map is rtgs_frames_land (event, then the first n_land frames scattered into images),
 * rtgs_untile_map one frame without the event. */
int rtgs_frames_land_map(const float* const* rank_bufs, const int32_t* rank_tiles, int32_t world, const int32_t* tile_map,
                         int32_t batch, int32_t n_land, int32_t width, int32_t height, float* const* images,
                         rtgs_stream_t stream, void* sent_event);
int rtgs_untile_map(const float* const* rank_bufs, const int32_t* rank_tiles, int32_t world, const int32_t* tile_map,
                    int32_t width, int32_t height, float* image, rtgs_stream_t stream);

/* Scene.hit (scene.py:406-450) for a batch of arbitrary rays: brute-force closest hit,
 * argmin t1 subject to ray.start < t1 < ray.end; lowest index wins exact ties.
 *   rays [n_rays][8]; hit_idx [n_rays] (-1 = miss); hit_t [n_rays][2] = (t1, t2), +inf on miss. */
int rtgs_closest_hit(const RtgsScene* scene, int32_t n_rays, const float* rays,
                     int32_t* hit_idx, float* hit_t, rtgs_stream_t stream);

/* Acceleration structure for the two arbitrary-ray queries (what the reference's BVH is to its
 * Scene.hit, scene.py:162-450): per-Gaussian fp64 inverse covariances, exact world-space boxes of
 * every 64 consecutive Gaussians and of every 16 such slabs, the squared radius of each Gaussian's
 * bounding ball.  Boxes are visited nearest first per wave of 64 rays.  It helps when consecutive Gaussians
 * are spatially close (rtgs.Scene stores them in Morton order); answers are identical to the
 * brute-force calls for any order.  Build once per scene (caller-owned buffer of
 * rtgs_accel_bytes(n) bytes, 16-B aligned), rebuild when the Gaussians change.
 * accel == NULL in the *_accel calls means brute force. */
size_t rtgs_accel_bytes(int32_t n);
int rtgs_accel_build(const RtgsScene* scene, void* accel, size_t accel_bytes, rtgs_stream_t stream);
int rtgs_closest_hit_accel(const RtgsScene* scene, const void* accel, size_t accel_bytes, int32_t n_rays,
                           const float* rays, int32_t* hit_idx, float* hit_t, rtgs_stream_t stream);
int rtgs_sample_step_rays_accel(const RtgsScene* scene, const void* accel, size_t accel_bytes,
                                int32_t n_rays, float* rays, float* sample_buf, float* attenuation_buf,
                                rtgs_stream_t stream);

/* RayTracer.sample_step (ray_tracer.py:79-104) for an ARBITRARY ray field, literally: Scene.hit by
 * brute force, Gaussian.eval at the chord midpoint (t1+t2)/2 along ray.direction, sample +=
 * attenuation * alpha * rgb, attenuation *= 1 - alpha, ray.start = t1 + 1e-8 (+inf on a miss).
 * The camera-ray fast path is rtgs_trace; this serves ray fields the caller has edited.
 *   rays [n_rays][8] in/out (.start); sample_buf [n_rays][3] and attenuation_buf [n_rays] in/out. */
int rtgs_sample_step_rays(const RtgsScene* scene, int32_t n_rays, float* rays, float* sample_buf,
                          float* attenuation_buf, rtgs_stream_t stream);

/* Gaussian.hit (gaussian.py:203-230), elementwise: ray k against Gaussian k. */
int rtgs_gaussian_hit(int32_t n, const float* rays /*[n][8]*/, const float* position /*[n][3]*/,
                      const float* rotation /*[n][4]*/, const float* scale /*[n][3]*/,
                      float* hit_t /*[n][2]*/, rtgs_stream_t stream);

/* Gaussian.eval (gaussian.py:183-201) incl. eval_sh (gaussian.py:140-181), elementwise:
 * out[k] = (rgb, alpha) of Gaussian k at pos[k] seen along dir[k].  sh may be NULL. */
int rtgs_gaussian_eval(int32_t n, const float* pos /*[n][3]*/, const float* dir /*[n][3]*/,
                       const float* position, const float* rotation, const float* scale,
                       const float* color_opacity, const float* sh, float* out /*[n][4]*/,
                       rtgs_stream_t stream);

/* quat.mul / conj / rot_vec3 / as_rotation_mat3 (utils/quaternion.py:8-35, 84-121), batched.
 * op: 0 mul(p,q)->[n][4]; 1 conj(p)->[n][4]; 2 rot_vec3(p, q.xyz)->[n][3];
 *     3 as_rotation_mat3(p)->[n][9] row-major.  q is [n][4] (ignored for 1 and 3). */
int rtgs_quat_op(int32_t op, int32_t n, const float* p, const float* q, float* out,
                 rtgs_stream_t stream);

/* Ray.get (ray.py:43-52): out[k] = o_k + t_k d_k. */
int rtgs_ray_get(int32_t n, const float* rays /*[n][8]*/, const float* t /*[n]*/,
                 float* out /*[n][3]*/, rtgs_stream_t stream);

/* RayTracer.generate_disp_buffer (ray_tracer.py:68-77): disp = sample / (ns + nst/nd). */
int rtgs_disp_buffer(int32_t n_pixels, const float* sample_buf, int32_t num_samples,
                     int32_t num_steps, int32_t num_depth, float* disp_buf, rtgs_stream_t stream);

/* RayTracer.clear_sample / clear_attenuation (ray_tracer.py:56-66): fill n floats. */
int rtgs_fill(float* dst, size_t n, float value, rtgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RTGS_HIP_H */
