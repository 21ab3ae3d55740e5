// rtgs_trace.hip -- the hot path: per-frame projection + one-pass trace/sort/composite.
//
// Reference semantics (paths relative to the reference's src/rtgs/):
//   ray generation        camera.py:31-71
//   ray/ellipsoid quadratic  gaussian.py:203-230  (A, B, C, delta, t1, t2; threshold 3)
//   closest-hit criteria  scene.py:433-437         (start < t1 < end, argmin t1)
//   peel + composite      ray_tracer.py:39-54, 79-104
//   shading               gaussian.py:140-201      (alpha = opacity*exp(-d^T S^-1 d), SH)
//
// MI355X design (not the reference's: it walks a BVH per ray per peel with fp32 3x3 inverses):
//   * k_project (once per frame, one thread per Gaussian, fp64 inside): the quadratic of
//     gaussian.py:215-219 is factored per Gaussian.  With E = (X, Y, -1), X/Y the pixel-centre
//     offsets from the image centre, delta/4 = E^T Q E is a conic in the image plane, and A and
//     b = B/2 are a quadratic and a linear form in (X, Y).  All three are stored as their
//     (exact) Taylor polynomials about an expansion point c -- the on-screen point where delta
//     is largest, rounded to fp32 -- so that the fp32 per-ray evaluation works with small
//     offsets (x, y) = (X, Y) - c and never forms the literal b^2 - A*C, which loses ~1e-3
//     relative for a 0.01-unit Gaussian 3 units away; the quadratic part of delta/4 is a sum
//     of squares (LDL^T), so sub-pixel needles do not cancel either.  For an ellipsoid in front
//     of the camera the conic is an ellipse and yields a screen bounding box; Gaussians that
//     contain the camera or lie behind it are culled here (they can never satisfy t1 > 0);
//     ellipsoids crossing the camera plane keep no bound and are decided per pixel by b < 0.
//     Gaussians are streamed in SLABS of 64 consecutive records (one wave of k_project); the
//     wave also reduces the slab's union screen bound and a depth key.  With the scene stored in
//     Morton order (rtgs.Scene does that at load) a slab is spatially compact: its bound is tight.
//   * k_trace: one workgroup per RTGS_TILE x RTGS_TILE macro-tile, one wave64 per 8x8 pixel
//     tile, one lane per ray.  The workgroup scans the slab
//     bounds (16 B per 64 Gaussians; in two levels for big scenes), keeps the slabs that touch
//     the macro-tile (deterministic ballot/prefix compaction into LDS), orders them nearest
//     first, expands them -- one wave per slab, one lane per Gaussian, 16 B bounds, SoA in HBM
//     -- and compacts the surviving Gaussians into an LDS list.  Each wave then re-tests the
//     survivors against its own 8x8 tile (one survivor per lane, ballot) and, for every set
//     bit, all 64 rays evaluate the exact conic; the record's address is wave-uniform, so it
//     arrives by scalar loads.  Hits go into a per-lane sorted register list of the
//     RTGS_MAX_DEPTH nearest (t1, index) -- the reference's `depth` peels in one pass -- and are
//     composited front to back at the end; a wave ballot ends the compositing loop early when
//     every lane is below the transmittance cut-off.  No MFMA: there is no contraction anywhere
//     on this path.
#include "rtgs_common.hpp"

#include <string.h>

#include <type_traits>
#include <vector>
#include <algorithm>
#include <cmath>

namespace rtgs {
namespace {

// Tuning switches; defaults = the fastest measured on MI355X (c3, tools/variants.sh, round 1).
// Other things tried and dropped, same scene: record broadcast with v_readlane from a lane-held
// copy 0.426 ms vs scalar loads 0.377; scalar loads prefetched one candidate ahead 0.406 (SGPR
// pressure); IEEE division/sqrt +8 %; 32x32 macro-tiles 0.475, 8x8 single-wave workgroups 0.473;
// a second slab level (1024-Gaussian groups) -12 % at 100 k Gaussians (it is what Tune<true> uses
// above 262 k); XCD-contiguous tile chunks 0.444 (centre-heavy scene: imbalance), row-striped
// XCD map no gain (round 2: one equal-count image WEDGE per XCD, most central tile first inside
// it, is a gain -- rtgs_launch_order below); software-pipelining insertion(k) with the test of
// k+1 0.329.
// Round 3, measured and dropped (A/B builds of the same tree, tools/ab.sh; k_trace ms at c3 / c5):
//  * candidate records double-buffered in SGPRs (candidate k+1's s_load_dwordx16 in flight while k is
//    evaluated; 8 s_mov_b64 per candidate to rotate the buffers): 0.1346 -> 0.1440 / 1.305 -> 1.361;
//  * a key-only K = 16 list -- ONE register per entry (t1 with its slot in the low 4 bits: an insertion is
//    16 v_med3_u32 and nothing else), the index word in LDS, the exponent in private scratch, exact
//    (t1, index) order restored on demand by bubble passes over entries with equal truncated t1 (parity
//    suite green, also with 10 low bits masked so that those paths ran all the time; commit e0e02ed):
//    SQ_INSTS_VALU 83.5 M -> 78.0 M per c3 launch, 0.1308 -> 0.1281 / 1.281 -> 1.344 -- and 137 MB of
//    scratch writes per 1080p frame reach HBM (WRITE_SIZE 24.9 -> 162 MB): a 2 % gain for six times the
//    memory traffic;
//  * 2 or 4 slab bounds in flight per wave in the expansion (Tune::kExpand) with that list's free registers:
//    0.1289 -> 0.1304 / 0.1307;
//  * whole frames alternating over two streams on ONE GPU (frame f+1's k_project + k_bin in the shadow of
//    frame f's k_trace): 0.1468 -> 0.1733 ms per frame at c3, 1.339 -> 1.369 at c5 (two resident k_trace
//    grids thrash each other's L2).
//  * supertile bins filled by k_project itself (every slab appends its id to the bins it touches with one
//    atomic per supertile; no k_bin launch; tiles rank-sort what they keep): project + bin 17.2 -> 15.0 us,
//    but k_trace 0.1325 -> 0.1465 / 1.304 -> 1.452 -- the per-tile rank sort (two barriers, n^2 LDS reads)
//    costs every tile what k_bin's one sort per supertile saves.
//  * the candidate loop in two instantiations, with and without the class-1 (general conic) code, chosen per
//    chunk from a class bit fetched with the distance bounds: the duplicated loop body costs the 72-VGPR
//    main instantiation 57 spilled registers (0.1313 -> 0.1920); the uncapped ones grow by 5-16 VGPRs.
// What the counters say (profiles/r03_c3_summary.json): of a wave's 16 k quad-cycles 28 % are spent
// issuing, 32 % stalled at issue behind the SIMD's other six waves, 40 % parked in s_waitcnt / barriers
// (scan, expansion, compositing gathers).  Instruction issue is 60 % of the time and the vector
// instructions two thirds of the issue, so a cut of x % of the VALU count returns about 0.4 x % of the
// time -- which is what each of the experiments above and the macro-tile bins (scan phase 7.7 k -> 7.2 k
// cycles: 0.1347 -> 0.1317) delivered.
// Round 4 (DESIGN.md section 4 has the numbers): new -- the tie fix-up at the end of k_trace (a ray whose 16-slot list
// lost entries to exactly tied twins peels on, `depth` DISTINCT entry distances as the reference composites them), the
// list form (kList: the sorted distinct entries of every ray go to memory, k_peel composites one per call: the viewer's
// sample() x 16 from one traversal), explicit tile lists and the region cull of k_project (any partition of the frame
// over the ranks; one compact sector per rank skips most of the projection), k_bin without scratch, k_project<1024>
// instead of k_super_bbox; SH staging per workgroup for the SH forms of scenes up to 262 k Gaussians (kStage: c3 + SH
// 0.2905 -> 0.2595 ms).  Measured and dropped again: SH coefficients staged per wave and list position (0.291 -> 0.325
// ms); the per-workgroup staging for the 1 M-Gaussian form (129 VGPRs: c5 1.293 -> 1.42 capped at 128) and with slots 48
// floats apart (four-way bank conflicts: no gain at all); explicit fmaf chains in the 1 M-Gaussian form's gather (141
// VGPRs); the next frame's projection on a stream of its own, also at the lowest priority (one library call with raw
// events, removed in round 5: 0.1502 -> 0.1542 ms per c3 frame: a dependency across two hardware queues costs more than
// the 17 us of kernels it hides).
#ifndef RTGS_STAMPS
#define RTGS_STAMPS 0     // diagnostic build: per-phase s_memtime stamps of wave 0 into the buffer
#endif                    // passed as attenuation_buf (8 x u64 per block); never quote its run time
#ifndef RTGS_COUNT
#define RTGS_COUNT 0      // counting build (librtgs_hip_count.so, bench.py's honesty columns): every wave adds its
#endif                    // candidate / hit / insertion counts to device counters read by rtgs_debug_counters()
#ifndef RTGS_ABLATE
#define RTGS_ABLATE 0     // timing-only builds (wrong images): 1 no insertion, 2 no hit path,
#endif                    // 3 no per-candidate loop, 4 no slab expansion either
#ifndef RTGS_WAVES_PER_SIMD
#define RTGS_WAVES_PER_SIMD 4   // __launch_bounds__ second argument: caps VGPRs at 512/this
#endif
#ifndef RTGS_WAVES_PER_SIMD_MAIN
#define RTGS_WAVES_PER_SIMD_MAIN 7   // the no-SH forms of scenes up to 4096 slabs (whole frame, list, batched frames): 72 VGPRs
#endif
#ifndef RTGS_WAVES_PER_SIMD_SH
#define RTGS_WAVES_PER_SIMD_SH 5   // the SH variant carries 15 basis values more per lane (c5: 1.54 ms at 5 with 2 gathers in flight, 1.60 at 4 / 4)
#endif
#ifndef RTGS_LOOKAHEAD
#define RTGS_LOOKAHEAD 1  // big scenes: touch the next candidate's record line one iteration ahead (scalar-cache
#endif                    // prefetch).  c5 (64 MB of records) 1.74 -> 1.58 ms; c3 (6.4 MB, L2-resident) 0.188 -> 0.192: off there

#ifndef RTGS_BINS
#define RTGS_BINS 1       // per-frame supertile bins of depth-sorted slab lists (k_bin); 0: every tile scans all slab bounds
#endif
#ifndef RTGS_HORIZON
#define RTGS_HORIZON 1    // chunk-level horizon test against the per-Gaussian distance bound (see k_trace)
#endif
#ifndef RTGS_TIER
#define RTGS_TIER 8       // insertions touch only the first RTGS_TIER slots while slot RTGS_TIER-1 is empty in every lane
#endif                    // (best of 4 / 6 / 8 / 10 / 12; without the tiers c3 0.1634 -> 0.1545 ms)

#ifndef RTGS_SH_STAGE
#define RTGS_SH_STAGE 1   // SH forms of scenes up to 262 k Gaussians: the 16x16 tile's referenced Gaussians' coefficients are
#endif                    // fetched ONCE per workgroup into LDS and read per lane from there (k_trace, "SH staging"): c3 + SH
                          // 0.2905 -> 0.2617 ms.  0: every lane gathers its entries' coefficients from memory
// (round 5: the tuning macros whose sweeps have stood for three rounds -- slab bounds in flight, list capacities, slabs
// per wave, gathers in flight, the scene size that switches to the two-level scan -- are the constants of Tune below;
// their measurements are in HISTORY.md)
constexpr int kSlab = 64;                                 // Gaussians per slab = one wave
// Batch sizes of the scan / expansion / colour-gather stages.  Measured optimum depends on the
// number of slabs: c3 (1.6 k slabs) 2/2/4 -> 0.293 ms vs 0.344 with 4/4/8; c5 (15.6 k slabs, SH)
// 4/4/8 -> 2.67 ms vs 3.11 with 2/2/4 (2.08 with the two-level scan below).  rtgs_trace picks
// the instantiation by scene size.
template <bool kBig>
struct Tune {
  static constexpr int kWaves = kWavesPerTile;              // waves per workgroup
  static constexpr int kBlock = kWaves * kWave;             // threads per workgroup
  static constexpr int kSlabIters = kBig ? 4 : 2;           // slab bounds in flight per thread
  static constexpr int kSlabStep = kSlabIters * kBlock;     // slabs scanned per barrier
  static constexpr int kSlabCap = kBig ? 2048 : 768;        // LDS list of surviving slabs
  static constexpr int kSlabsPerWave = kBig ? 4 : 2;        // slabs one wave expands per batch
  // LDS list of surviving Gaussians (indices): kWavesPerTile regions of kSlabsPerWave slabs' worth;
  // the same in both workgroup forms, so that both see the same list order
  static constexpr int kListCap = kSlabsPerWave * kWavesPerTile * kSlab;
  // colour/opacity gathers in flight (c5: the SH form's register pressure decides -- 8: 1.67 ms, 4: 1.53-1.60, 2: 1.54;
  // the SH forms of every size keep two)
  static constexpr int kGroup = kBig ? 2 : 4;
  static constexpr int kGroupSh = 2;
  // slabs one wave has in flight in the expansion (c3: 0.185 ms with 1, 0.189 with 2, 0.194 with 4; c5 does not care)
  static constexpr int kExpand = kBig ? 4 : 1;
  static_assert(kSlabCap >= kSlabStep + 64, "slab list too small");
};
constexpr int kBigSceneSlabs = 4096;   // scenes with more slabs than this (262 k Gaussians) use Tune<true>
// Tune<true> scans slab bounds in two levels: bounds of kSuper consecutive slabs first, then the
// slabs of the surviving groups only.  c5 (15.6 k slabs): the flat scan was 27 % of a workgroup's
// lifetime (every one of 32 400 workgroups streaming 250 KB of bounds through L2); c3 (1.6 k slabs)
// is faster flat (one barrier less), so Tune<false> keeps the flat scan.
constexpr int kSuper = 16;
// A tile's surviving slabs are visited nearest first (by k_project's depth key) when there are at
// most kSortMax of them (512 / 1024 for scenes above 262 k Gaussians: as many keys as fit the LDS
// they borrow; both workgroup forms use the same bound).  In stream (Morton) order near
// and far Gaussians interleave and nearly EVERY candidate inserts into some lane's list; nearest
// first, the lists fill with near entries and far candidates are rejected wave-wide: 56 -> 44
// inserting candidates on c3's central 8x8 tiles (simulated with the oracle; sorting single
// Gaussians exactly would give 43.5).  Pixels do not depend on the order (see list_insert2).
constexpr int kSortMaxAsk = 4096;   // capped by the LDS the depth keys can borrow: kListCap floats (512 / 1024)

constexpr int kBinSupertilesMax = 4096;   // supertile bins the workspace provides for (4 K at 64 px: 2 040), see k_bin
#ifndef RTGS_TILE_BINS
#define RTGS_TILE_BINS 1   // k_bin also writes every MACRO-TILE's own depth-sorted slab list (one load instead of the bin filter)
#endif
// per-macro-tile slab lists (k_bin): [count][ids ...], kTileBinCap ints = 1 KB per tile; a tile that lists more
// slabs gets count -1 and filters its supertile's bin; frames of more than kTileBinTilesMax tiles have none
constexpr int kTileBinCap = 256;
constexpr int kTileBinTilesMax = 65536;
constexpr int32_t kMaxGaussians = 1 << 26;   // 67 M: a record's byte offset (64 B each) fits 32 bits

struct TraceArgs {
  int32_t n;
  int32_t width, height;
  int32_t tiles_x;
  int32_t tile_first, tile_step;
  const int32_t* tile_list;   // optional: local tile l is macro-tile tile_list[l] (any partition of the frame), else tile_first + l * tile_step
  int32_t total_tiles;
  int32_t depth;
  float t_cutoff;
  uint32_t flags;
  float inv_fx, inv_fy;
  float rc[9];  // camera rotation, for the SH direction only
  float* sample;
  float* att;
  float* start;
  int32_t start_stride;
  const int32_t* tile_order;
  const FrameCam* pose;   // optional: camera pose in device memory (only its rotation is used here)
  // per-frame bins (k_bin): depth-sorted slab lists of the supertiles of 2^bin_shift macro-tiles
  const int32_t* bin_count;   // [supertiles], -1 = overflowed: that supertile's tiles scan for themselves
  const int32_t* bin_list;    // [supertiles][bin_stride]
  int32_t bin_stride, bin_shift, bin_sx;
  const int32_t* tile_bin;    // [tiles][kTileBinCap] or NULL: per-macro-tile lists, valid where the supertile's bin is
  // peel lists (rtgs_trace_lists) or NULL: instead of compositing, the launch stores every pixel's sorted DISTINCT
  // entries -- a count per pixel and [tile][RTGS_MAX_DEPTH][256 threads] Gaussian index, exponent, t1 -- for rtgs_peel
  // frames batched into ONE launch (rtgs_shard_frames, round 5): grid.y = frames; frame b's workspace (projected
  // records, bounds, bins, pose block) lies frame_stride BYTES behind frame b - 1's, its output out_stride FLOATS
  // behind.  Both 0 for the one-frame launches (grid.y = 1)
  size_t frame_stride;
  size_t out_stride;
  int32_t* list_cnt;  // [tile][256 threads]: how many entries the pixel's list holds (no terminators: slots beyond are never written)
  int32_t* list_idx;
  float* list_w;
  float* list_t;      // (only rtgs_peel_start reads it)
};

// frame b of a batched launch: the same object in frame b's workspace region (byte offset)
// (pointer arithmetic, NOT an integer round trip: through uintptr_t the result loses its provenance -- no longer
// "based on" the const __restrict__ kernel argument -- and the wave-uniform record loads of the candidate loop come
// out as vector loads instead of s_load_dwordx16: the batched form ran 1.5 x slower per tile that way)
template <typename T>
__device__ __forceinline__ T* at_frame(T* p, size_t off) {
  using Byte = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return reinterpret_cast<T*>(reinterpret_cast<Byte*>(p) + off);
}

#if RTGS_COUNT
// [0] candidates a wave evaluated (x 64 lanes = ray-Gaussian conic tests executed), [1] of those with a hit
// in some lane, [2] insertion-network runs, [3] lanes with a valid hit, [4] lanes that inserted,
// [5] Gaussian bounds tested by the expansion (lanes), [6] slab bounds tested (lanes), [7] waves,
// forms with the SH staging (kStage): [8] entries shaded from LDS, [9] from memory, [10] distinct Gaussians the
// workgroups' lists named (summed), [11] workgroups that named more than there are staging slots
__device__ unsigned long long g_count[12];
#endif

// ------------------------------------------------------------------------------------
// per-frame projection
// ------------------------------------------------------------------------------------

// kThreads = 256, or 1024 for scenes whose slab bounds are scanned in two levels (Tune<true>): a block is then
// exactly one group of kSuper = 16 slabs and writes the group's union bound itself (round 3 launched
// k_super_bbox for that: 5.2 us of c5's prologue)
template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_project(int32_t n, const float* __restrict__ pos,
                                                  const float* __restrict__ rot,
                                                  const float* __restrict__ scale, FrameCam cam,
                                                  const FrameCam* __restrict__ pose,
                                                  float4* __restrict__ bbox,
                                                  float4* __restrict__ slab_bbox,
                                                  float4* __restrict__ super_bbox,
                                                  float* __restrict__ slab_key,
                                                  float* __restrict__ rec, float* __restrict__ tail,
                                                  float* __restrict__ tmin, int32_t* __restrict__ bin_reset,
                                                  const float4 region, const int32_t region_on,
                                                  const size_t frame_stride) {
  // frames batched into one launch: blockIdx.y = frame, whose workspace region (and pose block) lies
  // blockIdx.y * frame_stride bytes behind the first one's
  if (frame_stride) {
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    pose = at_frame(pose, fo);
    bbox = at_frame(bbox, fo); slab_bbox = at_frame(slab_bbox, fo); slab_key = at_frame(slab_key, fo);
    if (super_bbox) super_bbox = at_frame(super_bbox, fo);
    rec = at_frame(rec, fo); tail = at_frame(tail, fo); tmin = at_frame(tmin, fo);
    if (bin_reset) bin_reset = at_frame(bin_reset, fo);
  }
  const int g0 = blockIdx.x * blockDim.x + threadIdx.x;
  // a frame projected WITHOUT bins must not leave last frame's in the workspace: -1 = "scan for yourself"
  // (bins are only used from 32 k Gaussians up, so the first kBinSupertilesMax threads exist)
  if (bin_reset && g0 < kBinSupertilesMax) bin_reset[g0] = -1;
  const bool live = g0 < n;
  const int g = live ? g0 : n - 1;   // padding lanes recompute the last record, store nothing
  const double inf = __builtin_huge_val();
  float4 bb = make_float4(kInf, kInf, -kInf, -kInf);  // empty: culled
  float r[kRecFloats], tl[kTailFloats];
#pragma unroll
  for (int k = 0; k < kRecFloats; ++k) r[k] = 0.f;
#pragma unroll
  for (int k = 0; k < kTailFloats; ++k) tl[k] = 0.f;

  if (pose) cam = *pose;   // camera pose from device memory (RtgsCamera.device_pose): wave-uniform loads
  const double mx = cam.o[0] - (double)pos[3 * g + 0];
  const double my = cam.o[1] - (double)pos[3 * g + 1];
  const double mz = cam.o[2] - (double)pos[3 * g + 2];
  // Region cull (round 4, RTGS_TRACE_REGION): a shard that is a compact part of the frame (one sector per rank,
  // rtgs.parallel.Partition.wedges) only needs the Gaussians that can reach its rectangle.  The ellipsoid lies in
  // the ball of radius sqrt(3) |q|^2 s_max about its centre; in front of the camera that ball projects inside
  // [f (x -+ r) / (d +- r)] per axis -- a conservative screen interval from a dozen operations, before the 600 of
  // the projection proper.  A Gaussian off the rectangle (by more than a pixel) is treated as culled: empty bounds,
  // no record.  Its distance still enters the slab's depth key, so slabs are visited in the same order and the
  // shard's pixels equal the whole frame's bit for bit.
  bool off = false;
  if (region_on) {
    const double k2r = (double)rot[4 * g] * rot[4 * g] + (double)rot[4 * g + 1] * rot[4 * g + 1] +
                       (double)rot[4 * g + 2] * rot[4 * g + 2] + (double)rot[4 * g + 3] * rot[4 * g + 3];
    const double smr = fmax(fabs((double)scale[3 * g]), fmax(fabs((double)scale[3 * g + 1]), fabs((double)scale[3 * g + 2])));
    // (camera-space lengths below are |q_cam|^2 times world lengths -- rc is |q_cam|^2 times a rotation -- so is the radius)
    const double ksr = sqrt(cam.rc[0] * cam.rc[0] + cam.rc[3] * cam.rc[3] + cam.rc[6] * cam.rc[6]);
    const double rr = 1.7320508075688774 * k2r * smr * (1.0 + 1e-9) * ksr;
    // camera-space centre: d_cam = rc^T (mu - o), looking along -z
    const double xc = -(cam.rc[0] * mx + cam.rc[3] * my + cam.rc[6] * mz);
    const double yc = -(cam.rc[1] * mx + cam.rc[4] * my + cam.rc[7] * mz);
    const double dc = cam.rc[2] * mx + cam.rc[5] * my + cam.rc[8] * mz;    // depth along the view direction
    if (dc - rr > 0.0 && rr < inf) {      // wholly in front of the camera plane (else: keep, the exact test decides)
      const double fx = 1.0 / cam.inv_fx, fy = 1.0 / cam.inv_fy;
      const double xh = xc + rr, xl = xc - rr, yh = yc + rr, yl = yc - rr;
      const double Xhi = fx * xh / (xh >= 0.0 ? dc - rr : dc + rr), Xlo = fx * xl / (xl >= 0.0 ? dc + rr : dc - rr);
      const double Yhi = fy * yh / (yh >= 0.0 ? dc - rr : dc + rr), Ylo = fy * yl / (yl >= 0.0 ? dc + rr : dc - rr);
      off = Xhi < (double)region.x - 1.0 || Xlo > (double)region.z + 1.0 || Yhi < (double)region.y - 1.0 ||
            Ylo > (double)region.w + 1.0;
    }
  }
  if (off) bb = make_float4(kInf, kInf, -kInf, -kInf);
  if (!off) {
  // quaternion + scale -> inverse covariance, fp64 (utils/quaternion.py:99-121, gaussian.py:86-102)
  double m6[6];
  cov_inverse6(rot + 4 * (size_t)g, scale + 3 * (size_t)g, m6);
  const double Mxx = m6[0], Mxy = m6[1], Mxz = m6[2], Myy = m6[3], Myz = m6[4], Mzz = m6[5];
  // u = M m ; C = m.M m - 3   (gaussian.py:217-219)
  const double u0 = Mxx * mx + Mxy * my + Mxz * mz;
  const double u1 = Mxy * mx + Myy * my + Myz * mz;
  const double u2 = Mxz * mx + Myz * my + Mzz * mz;
  const double C = mx * u0 + my * u1 + mz * u2 - 3.0;

  if (C > 0.0) {  // camera outside the ellipsoid (inside: t1 < 0 < t2, never a hit)
    // camera frame, focal lengths folded in: D = diag(1/fx, 1/fy, 1) E, d_world = rc D.
    const double sc[3] = {cam.inv_fx, cam.inv_fy, 1.0};
    double col[3][3];  // col[k] = k-th column of rc scaled by sc[k]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      col[k][0] = cam.rc[0 + k] * sc[k];
      col[k][1] = cam.rc[3 + k] * sc[k];
      col[k][2] = cam.rc[6 + k] * sc[k];
    }
    double v[3], Mc[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = col[k][0] * u0 + col[k][1] * u1 + col[k][2] * u2;
      const double t0 = Mxx * col[k][0] + Mxy * col[k][1] + Mxz * col[k][2];
      const double t1 = Mxy * col[k][0] + Myy * col[k][1] + Myz * col[k][2];
      const double t2 = Mxz * col[k][0] + Myz * col[k][1] + Mzz * col[k][2];
#pragma unroll
      for (int l = 0; l < 3; ++l) Mc[l][k] = col[l][0] * t0 + col[l][1] * t1 + col[l][2] * t2;
    }
    // Q = v v^T - C Mc ; delta/4 = E^T Q E, E = (X, Y, -1)
    const double Q00 = v[0] * v[0] - C * Mc[0][0];
    const double Q01 = v[0] * v[1] - C * Mc[0][1];
    const double Q11 = v[1] * v[1] - C * Mc[1][1];
    const double Q02 = v[0] * v[2] - C * Mc[0][2];
    const double Q12 = v[1] * v[2] - C * Mc[1][2];
    const double Q22 = v[2] * v[2] - C * Mc[2][2];
    const double det = Q00 * Q11 - Q01 * Q01;
    // Screen bound + visibility.  An ellipse (negative-definite 2x2 block) that reaches 0 and
    // whose cone opens towards -z gets its bounding box; one that never reaches 0 or is the
    // mirror cone behind the camera is culled; anything else (ellipsoid crossing the camera
    // plane: hyperbola / parabola) is kept without a bound and sorted out per pixel by b < 0.
    bool keep = true;
    double X0 = 0.0, Y0 = 0.0;
    bb = make_float4(-kInf, -kInf, kInf, kInf);
    if (Q00 < 0.0 && det > 0.0) {
      X0 = (Q11 * Q02 - Q01 * Q12) / det;
      Y0 = (Q00 * Q12 - Q01 * Q02) / det;
      const double K = Q22 - (Q02 * X0 + Q12 * Y0);
      const double b0 = v[0] * X0 + v[1] * Y0 - v[2];
      if (!(K > 0.0) || !(b0 < 0.0)) {
        keep = false;
      } else {
        const double hx = sqrt(K * (-Q11) / det), hy = sqrt(K * (-Q00) / det);
        const double pad_x = 1e-3 + 1e-5 * (fabs(X0) + hx), pad_y = 1e-3 + 1e-5 * (fabs(Y0) + hy);
        if (hx < inf && hy < inf && fabs(X0) < inf && fabs(Y0) < inf)
          bb = make_float4((float)(X0 - hx - pad_x), (float)(Y0 - hy - pad_y),
                           (float)(X0 + hx + pad_x), (float)(Y0 + hy + pad_y));
      }
    }
    // Expansion point.  For an ellipse: the point of the screen rectangle where delta is LARGEST
    // (the Mahalanobis-nearest point to the conic centre: a convex 2-D QP, solved exactly by
    // checking the centre and the minimiser of each edge).  Every on-screen pixel then has
    // 0 <= terms <= delta(there) <= 3 A, so the fp32 Taylor evaluation never cancels, however
    // far off-screen the centre is; and if even that maximum is <= 0 the Gaussian is invisible.
    // Other conics: see the else branch.
    const double hw = 0.5 * cam.width, hh = 0.5 * cam.height;
    double cx = 0.0, cy = 0.0;
    if (keep && Q00 < 0.0 && det > 0.0) {
      cx = X0; cy = Y0;
      if (!(fabs(X0) <= hw && fabs(Y0) <= hh)) {
        const double P00 = -Q00, P01 = -Q01, P11 = -Q11;
        double best = inf;
#pragma unroll
        for (int edge = 0; edge < 4; ++edge) {
          double px, py;
          if (edge < 2) {   // x = -hw, +hw
            px = edge == 0 ? -hw : hw;
            py = fmin(fmax(Y0 - P01 * (px - X0) / P11, -hh), hh);
          } else {          // y = -hh, +hh
            py = edge == 2 ? -hh : hh;
            px = fmin(fmax(X0 - P01 * (py - Y0) / P00, -hw), hw);
          }
          const double dx = px - X0, dy = py - Y0;
          const double f = P00 * dx * dx + 2.0 * P01 * dx * dy + P11 * dy * dy;
          if (f < best) { best = f; cx = px; cy = py; }
        }
        const double K = Q22 - (Q02 * X0 + Q12 * Y0);
        if (!(best < K)) keep = false;   // delta/4 = K - f <= 0 on the whole screen (NaN: cull)
      }
    } else if (keep) {
      // Not an ellipse (the ellipsoid crosses the camera plane): delta has no bounded maximum, so
      // the expansion starts from the screen centre (offsets bounded by half the screen) -- and,
      // if the conic has a dominant positive eigenvalue, i.e. it is a needle, slides across the
      // needle onto its ridge, so that pixels inside the needle see a small perpendicular offset
      // (tests/scenes.py fuzz seeds 8476 / 8982: 7e-5 without).  Chasing the largest delta over
      // the screen instead is wrong: it runs away along the open branch (errors up to 1.6e-2).
      const double P00 = -Q00, P01 = -Q01, P11 = -Q11;
      const double half_tr = 0.5 * (P00 + P11), rad = sqrt(0.25 * (P00 - P11) * (P00 - P11) + P01 * P01);
      const double lam = half_tr + rad;          // the larger eigenvalue of P
      if (lam > 0.0 && lam > 16.0 * fabs(half_tr - rad)) {
        double nx = P01, ny = lam - P00;         // its eigenvector, from the better-conditioned row
        if (fabs(lam - P11) > fabs(ny)) { nx = lam - P11; ny = P01; }
        const double nn = sqrt(nx * nx + ny * ny);
        if (nn > 0.0 && nn < inf) {
          nx /= nn; ny /= nn;
          // delta/4 (c + s n) = f(c) + 2 s n.q - s^2 lam, q = (Q E_c)_{xy}: largest at s = n.q / lam
          const double q0c = Q00 * cx + Q01 * cy - Q02, q1c = Q01 * cx + Q11 * cy - Q12;
          const double sstep = (nx * q0c + ny * q1c) / lam;
          const double tx = cx + sstep * nx, ty = cy + sstep * ny;
          if (fabs(tx) <= hw && fabs(ty) <= hh) { cx = tx; cy = ty; }
        }
      }
    }
    if (keep && C < inf) {
      // the expansion point as the kernel will see it: rounded to fp32, NOT to the pixel grid -- a
      // needle a hundredth of a pixel wide is tens of its widths away from the nearest half-
      // integer, and expanding there brings back the cancellation (tools/fuzz_needles.py seed
      // 9065: delta/4 = -2.2e9 + 4.4e9 - 2.2e9).  x = X - cx then rounds once, to 2^-24 relative.
      cx = (double)(float)cx;
      cy = (double)(float)cy;
      // E_c = (cx, cy, -1)
      const double q0 = Q00 * cx + Q01 * cy - Q02, q1 = Q01 * cx + Q11 * cy - Q12,
                   q2 = Q02 * cx + Q12 * cy - Q22;
      const double e0 = Mc[0][0] * cx + Mc[0][1] * cy - Mc[0][2];
      const double e1 = Mc[1][0] * cx + Mc[1][1] * cy - Mc[1][2];
      const double e2 = Mc[2][0] * cx + Mc[2][1] * cy - Mc[2][2];
      r[0] = (float)cx;              r[1] = (float)cy;
      r[2] = (float)(cx * q0 + cy * q1 - q2);      // delta/4 at the expansion point
      // quadratic part x^T P x, P = -Q(2x2), in LDL^T form about the larger diagonal entry:
      // d1 (x' + l y')^2 + d2 y'^2 with (x', y') = (x, y) or, flagged in the tail, (y, x).  For an
      // ellipse (P positive definite) that is a SUM OF SQUARES, so a needle a fraction of a pixel
      // wide does not lose its exponent to cancellation between p00 x^2, 2 p01 x y and p11 y^2 (a
      // 200:1 needle 0.06 px wide, 20 px from its expansion point: 2e-3 relative in delta with the
      // plain polynomial, 2e-5 this way; tools/fuzz_needles.py seed 9021).  Other conics take it
      // when the pivot dominates the off-diagonal entry (needles again); otherwise they keep the
      // polynomial: d1 = p00, l = 0, d2 = p11 and the cross term 2 p01 in the tail.
      bool swp = false;
      const bool ellipse = Q00 < 0.0 && det > 0.0;
      float cross = 0.f;
      {
        const double P00 = -Q00, P01 = -Q01, P11 = -Q11;
        swp = !ellipse && fabs(P11) > fabs(P00);   // an ellipse is a sum of squares either way
        const double piv = swp ? P11 : P00;
        if ((ellipse || fabs(piv) >= fabs(P01)) && fabs(piv) > 0.0 && fabs(piv) < inf) {
          r[5] = (float)piv;           r[6] = (float)(P01 / piv);   r[7] = (float)(det / piv);
        } else {
          swp = false;
          r[5] = (float)P00;           r[6] = 0.f;                  r[7] = (float)P11;
          cross = (float)(2.0 * P01);
        }
      }
      // with the flag set k_trace exchanges x and y once, so every polynomial is stored in (y, x)
      r[3] = (float)(2.0 * (swp ? q1 : q0));       r[4] = (float)(2.0 * (swp ? q0 : q1));   // gradient of delta/4
      r[8] = (float)(v[0] * cx + v[1] * cy - v[2]);  // b = B/2 at the expansion point
      r[9] = (float)(swp ? v[1] : v[0]);           r[10] = (float)(swp ? v[0] : v[1]);
      r[11] = (float)C;
      // A = d^T Sigma^-1 d is only needed for the exponent delta/(4A) of an inserted hit.  Since
      // delta/4 = b^2 - A C identically, A = (b^2 - delta/4) / C: two operations on values the hit
      // path has anyway, instead of a 6-coefficient polynomial -- and the record fits ONE 64-B
      // cache line (one s_load_dwordx16; the scalar cache holds twice as many).  The subtraction
      // costs 2^-22 (1 + 3/C) relative (delta/4 <= 3A on a hit), so only Gaussians with C >= 8
      // take it; the few the camera is within ~3.3 sigma of (C < 8) and all non-elliptic conics
      // (ellipsoids crossing the camera plane) are CLASS 1: they keep A's Taylor polynomial, the
      // cross term and the swap flag in a 32-B tail that k_trace loads behind a scalar branch.
      const bool compact = ellipse && C >= 8.0 && cross == 0.f && !swp;
      r[12] = (float)(1.0 / C);
      r[13] = compact ? 0.f : 1.f;
      if (!compact) {
        tl[0] = (float)(cx * e0 + cy * e1 - e2);      // A at the expansion point
        tl[1] = (float)(2.0 * (swp ? e1 : e0));       tl[2] = (float)(2.0 * (swp ? e0 : e1));
        tl[3] = (float)(swp ? Mc[1][1] : Mc[0][0]);   tl[4] = (float)(2.0 * Mc[0][1]);
        tl[5] = (float)(swp ? Mc[0][0] : Mc[1][1]);
        tl[6] = cross;
        tl[7] = swp ? 1.f : 0.f;
      }
    } else {
      bb = make_float4(kInf, kInf, -kInf, -kInf);
    }
  }
  }   // !off
  if (live && off) bbox[g] = bb;      // (empty bounds: the Gaussian is in no tile's list, its record is never read)
  if (live && !off) {
    // lower bound of the entry distance t1 over ALL rays: the ellipsoid {d^T Sigma^-1 d <= 3} lies in
    // the ball of radius sqrt(3) |q|^2 s_max about the centre.  k_trace's horizon test skips a
    // candidate whose bound is beyond the 16th entry of every ray of the tile.
    {
      const double k2 = (double)rot[4 * g] * rot[4 * g] + (double)rot[4 * g + 1] * rot[4 * g + 1] +
                        (double)rot[4 * g + 2] * rot[4 * g + 2] + (double)rot[4 * g + 3] * rot[4 * g + 3];
      const double smax = fmax(fabs((double)scale[3 * g]), fmax(fabs((double)scale[3 * g + 1]), fabs((double)scale[3 * g + 2])));
      // t1 counts in units of the WORLD direction's length, and the reference's rot_vec3 of a NON-UNIT camera
      // quaternion stretches that direction by |q_cam|^2 (camera.py:52, utils/quaternion.py:84-96: rc is |q|^2 times
      // a rotation): a world distance is |q_cam|^2 entry-distance units.  (Round 5: up to here the bound was left in
      // world units and the horizon test pruned real hits of every ray behind a camera quaternion longer than 1 --
      // found when the one-entry list, which has no horizon, was retired and its depth-1 test case met this kernel.)
      const double ks = sqrt(cam.rc[0] * cam.rc[0] + cam.rc[3] * cam.rc[3] + cam.rc[6] * cam.rc[6]);
      const double lo = (sqrt(mx * mx + my * my + mz * mz) - 1.7320508075688774 * k2 * smax) / ks;
      // (1e-4 relative: the t1 it stands in for is evaluated per ray in fp32 with v_rcp / v_sqrt on
      // fp32-rounded record coefficients, a few 1e-7 relative; the bound must stay below THAT value)
      tmin[g] = lo > 0.0 ? (float)(lo * (1.0 - 1e-4)) : 0.f;   // NaN -> 0: never skipped
    }
    bbox[g] = bb;
    float4* out = reinterpret_cast<float4*>(rec + (size_t)g * kRecFloats);
#pragma unroll
    for (int k = 0; k < kRecFloats / 4; ++k)
      out[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
    if (r[13] != 0.f) {   // the tail of a compact record is never read
      float4* to = reinterpret_cast<float4*>(tail + (size_t)g * kTailFloats);
      to[0] = make_float4(tl[0], tl[1], tl[2], tl[3]);
      to[1] = make_float4(tl[4], tl[5], tl[6], tl[7]);
    }
  }
  // union bound of the slab (= this wave's 64 Gaussians); culled members are empty boxes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bb.x = fminf(bb.x, __shfl_xor(bb.x, off));
    bb.y = fminf(bb.y, __shfl_xor(bb.y, off));
    bb.z = fmaxf(bb.z, __shfl_xor(bb.z, off));
    bb.w = fmaxf(bb.w, __shfl_xor(bb.w, off));
  }
  // depth key of the slab: summed squared distance of its members from the camera.  k_trace
  // visits a tile's surviving slabs nearest first, so that far candidates meet full lists.
  float d2 = (float)(mx * mx + my * my + mz * mz);
  d2 = d2 == d2 ? d2 : kInf;   // a NaN key would give two slabs the same rank in k_trace's sort
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d2 += __shfl_xor(d2, off);
  if ((threadIdx.x & (kWave - 1)) == 0 && live) {
    slab_bbox[g0 / kSlab] = bb;
    slab_key[g0 / kSlab] = d2;
  }
  if constexpr (kThreads == kSuper * kSlab) {
    // union bound of the block's kSuper slabs (slabs past the end contribute nothing)
    __shared__ float4 s_sb[kSuper];
    if ((threadIdx.x & (kWave - 1)) == 0) s_sb[threadIdx.x / kWave] = live ? bb : make_float4(kInf, kInf, -kInf, -kInf);
    __syncthreads();
    if (threadIdx.x < kWave) {
      float4 u = threadIdx.x < kSuper ? s_sb[threadIdx.x] : make_float4(kInf, kInf, -kInf, -kInf);
#pragma unroll
      for (int off = kSuper / 2; off > 0; off >>= 1) {
        u.x = fminf(u.x, __shfl_xor(u.x, off));
        u.y = fminf(u.y, __shfl_xor(u.y, off));
        u.z = fmaxf(u.z, __shfl_xor(u.z, off));
        u.w = fmaxf(u.w, __shfl_xor(u.w, off));
      }
      if (threadIdx.x == 0) super_bbox[blockIdx.x] = u;
    }
  }
}

// ------------------------------------------------------------------------------------
// trace
// ------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------
// per-frame binning: one workgroup per SUPERTILE (a square of 2^shift x 2^shift macro-tiles, 64 px
// and up) scans all slab bounds once, keeps the slabs that touch it and writes their ids sorted by
// k_project's depth key.  Every macro-tile inside then filters ~100 listed slabs (one coalesced
// read + one gather, order preserved: no sort) instead of scanning all of them in four dependent
// steps and rank-sorting its survivors -- a quarter of a k_trace workgroup's life at c3.
// ------------------------------------------------------------------------------------
constexpr int kBinCapMax = 1024;          // slabs a bin can list; more = overflow, the tiles fall back to scanning
// 16-slab groups a supertile can keep in k_bin<true> (16 k slabs = 1 M Gaussians' worth); more = overflow, its tiles scan
// for themselves.  (4096: 12 KB more LDS, a workgroup per CU fewer now that the listed slabs' bounds are kept too: c5
// project + bin 63.1 -> 58.6 us)
constexpr int kBinGroups = 1024;
constexpr int kBinIters = 8;   // slab bounds in flight per thread: c3's 1 563 slabs are ONE step (4: two; 16.3 -> 15.8 us)

__host__ __device__ inline int bin_stride_of(size_t nslabs) {
  const size_t c = nslabs < (size_t)kBinCapMax ? nslabs : (size_t)kBinCapMax;
  return (int)((c + 3) & ~(size_t)3);
}

// which macro-tiles a projection's bins are for (all of them: step <= 1 and no rectangle)
struct BinOwner { int32_t first, step; int32_t rect_on; float4 rect; };

template <bool kGroups>   // kGroups: bounds of kSuper-slab groups first (scenes with k_super_bbox's output)
__global__ __launch_bounds__(256) void k_bin(int32_t nslabs, const float4* __restrict__ slab_bbox,
                                              const float4* __restrict__ super_bbox,
                                              const float* __restrict__ slab_key, int32_t width, int32_t height,
                                              int32_t shift, int32_t bin_sx, int32_t stride,
                                              int32_t* __restrict__ bin_count, int32_t* __restrict__ bin_list,
                                              int32_t* __restrict__ tile_bin, int32_t tiles_x, int32_t tiles_y,
                                              const size_t frame_stride, const BinOwner own) {
  if (frame_stride) {      // frames batched into one launch: blockIdx.y = frame (see k_project)
    const size_t fo = (size_t)blockIdx.y * frame_stride;
    slab_bbox = at_frame(slab_bbox, fo); super_bbox = at_frame(super_bbox, fo); slab_key = at_frame(slab_key, fo);
    bin_count = at_frame(bin_count, fo); bin_list = at_frame(bin_list, fo);
    if (tile_bin) tile_bin = at_frame(tile_bin, fo);
  }
  __shared__ int s_id[kBinCapMax];
  __shared__ float s_key[kBinCapMax];
  __shared__ float4 s_bb[RTGS_TILE_BINS ? kBinCapMax : 1];      // bounds of the listed slabs (list order)
  __shared__ int s_perm[RTGS_TILE_BINS ? kBinCapMax : 1];        // sorted position -> list position
  __shared__ int s_wc[2][4];
  constexpr int kGrpCap = kBinGroups;
  __shared__ int s_grp[kGroups ? kGrpCap : 1];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int st = blockIdx.x;
  const int edge = kTile << shift;
  const int i0 = (st % bin_sx) * edge, j0 = (st / bin_sx) * edge;
  const float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
  // bounds of the pixel centres the supertile covers (the macro-tiles' own bounds lie inside)
  const float Xlo = ((float)i0 + 0.5f) - hw, Xhi = ((float)min(i0 + edge, width) - 0.5f) - hw;
  const float Ylo = ((float)j0 + 0.5f) - hh, Yhi = ((float)min(j0 + edge, height) - 0.5f) - hh;
  // A shard's own projection (rtgs_shard_frame(s): this frame's bins are read by THAT shard's trace and nobody else)
  // only needs the bins of the tiles the shard owns: macro-tile t with (t - first) mod step == 0, inside the shard's
  // region rectangle when it has one.  A supertile without an owned tile is skipped whole -- every second one of a
  // round-robin 1/8 shard, three quarters of the frame for one image sector in eight -- and so are the other tiles'
  // lists (k_bin was 13 % of a rank's 1080p frame at 8 ranks, 10 % of a 4K one).
  auto owned = [&](int tx, int ty) -> bool {
    if (tx >= tiles_x || ty >= tiles_y) return false;
    if (own.step > 1 && (ty * tiles_x + tx - own.first) % own.step != 0) return false;
    if (own.rect_on) {
      const float xl = ((float)(tx * kTile) + 0.5f) - hw, xh = ((float)min(tx * kTile + kTile, width) - 0.5f) - hw;
      const float yl = ((float)(ty * kTile) + 0.5f) - hh, yh = ((float)min(ty * kTile + kTile, height) - 0.5f) - hh;
      if (xl > own.rect.z || xh < own.rect.x || yl > own.rect.w || yh < own.rect.y) return false;
    }
    return true;
  };
  const bool restricted = own.step > 1 || own.rect_on != 0;
  if (restricted) {
    const int per_edge_o = 1 << shift;
    bool mine = false;
    for (int t = tid; t < per_edge_o * per_edge_o; t += 256)
      mine = mine || owned((st % bin_sx) * per_edge_o + (t % per_edge_o), (st / bin_sx) * per_edge_o + (t / per_edge_o));
    if (!__syncthreads_or(mine ? 1 : 0)) return;     // (uniform: nobody owns a tile here)
  }
  int cnt = 0, par = 0;
  bool over = false;
  int nitems = nslabs;
  if constexpr (kGroups) {
    // level 1: surviving groups, in stream order (ballot prefix per step)
    const int ngroups = (nslabs + kSuper - 1) / kSuper;
    int gcnt = 0;
    for (int base = 0; base < ngroups; base += kBinIters * 256, par ^= 1) {
      float4 gb[kBinIters];
#pragma unroll
      for (int it = 0; it < kBinIters; ++it) {
        const int u = base + (wave * kBinIters + it) * kWave + lane;   // wave-contiguous: the list comes out in item order
        gb[it] = make_float4(kInf, kInf, -kInf, -kInf);
        if (u < ngroups) gb[it] = super_bbox[u];
      }
      unsigned long long gm[kBinIters];
      int wc = 0;
#pragma unroll
      for (int it = 0; it < kBinIters; ++it) {
        gm[it] = __ballot(gb[it].x <= Xhi && gb[it].z >= Xlo && gb[it].y <= Yhi && gb[it].w >= Ylo);
        wc += __popcll(gm[it]);
      }
      if (lane == 0) s_wc[par][wave] = wc;
      __syncthreads();
      int off = gcnt;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int c = s_wc[par][w];
        off += (w < wave) ? c : 0;
        gcnt += c;
      }
      if (gcnt > kGrpCap) { over = true; break; }
#pragma unroll
      for (int it = 0; it < kBinIters; ++it) {
        if ((gm[it] >> lane) & 1ull)
          s_grp[off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(gm[it] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)gm[it], 0u))] = base + (wave * kBinIters + it) * kWave + lane;
        off += __popcll(gm[it]);
      }
    }
    __syncthreads();
    nitems = over ? 0 : gcnt * kSuper;
  }
  for (int base = 0; base < nitems; base += kBinIters * 256, par ^= 1) {
    float4 sb[kBinIters];
    float sk[kBinIters];
    int sid[kBinIters];
#pragma unroll
    for (int it = 0; it < kBinIters; ++it) {
      const int item = base + (wave * kBinIters + it) * kWave + lane;
      int sl = item;
      if constexpr (kGroups) sl = item < nitems ? s_grp[item / kSuper] * kSuper + (item % kSuper) : nslabs;
      sid[it] = sl;
      sb[it] = make_float4(kInf, kInf, -kInf, -kInf);
      sk[it] = 0.f;
      if (sl < nslabs) { sb[it] = slab_bbox[sl]; sk[it] = slab_key[sl]; }
    }
    unsigned long long km[kBinIters];
    int wc = 0;
#pragma unroll
    for (int it = 0; it < kBinIters; ++it) {
      const bool keep = sb[it].x <= Xhi && sb[it].z >= Xlo && sb[it].y <= Yhi && sb[it].w >= Ylo;
      km[it] = __ballot(keep);
      wc += __popcll(km[it]);
    }
    if (lane == 0) s_wc[par][wave] = wc;
    __syncthreads();
    int off = cnt;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = s_wc[par][w];
      off += (w < wave) ? c : 0;
      cnt += c;
    }
    if (cnt > stride) { over = true; break; }   // uniform
#pragma unroll
    for (int it = 0; it < kBinIters; ++it) {
      if ((km[it] >> lane) & 1ull) {
        const int at = off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km[it] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km[it], 0u));
        s_id[at] = sid[it];
        s_key[at] = sk[it];
        if constexpr (RTGS_TILE_BINS != 0) s_bb[at] = sb[it];
      }
      off += __popcll(km[it]);
    }
  }
  if (over) {
    if (tid == 0) bin_count[st] = -1;
    return;
  }
  __syncthreads();
  // nearest first: rank by (key, slab id) -- the ids were listed in increasing order.  The rank goes
  // straight to the bin (global) and, as the inverse permutation, to s_perm: nothing is carried in
  // registers from this loop to the tile lists below (round 3 kept rank[4] + bounds[4] per thread,
  // which the compiler put in 80 B of scratch per lane)
  for (int e = tid; e < cnt; e += 256) {
    const float my = s_key[e];
    int r = 0;
    for (int j = 0; j < cnt; ++j) {
      const float kj = s_key[j];
      r += (kj < my || (kj == my && j < e)) ? 1 : 0;
    }
    bin_list[(size_t)st * stride + r] = s_id[e];
    if constexpr (RTGS_TILE_BINS != 0) s_perm[r] = e;
  }
  if (tid == 0) bin_count[st] = cnt;
  if constexpr (RTGS_TILE_BINS != 0) {
    if (!tile_bin) return;
    // The macro-tiles' own lists: the sorted list filtered by each tile's bounds, order preserved (= what
    // the tile's filter of this bin would keep).  One wave per tile, 64 listed slabs per step.
    __syncthreads();
    const int per_edge = 1 << shift, ntl = per_edge * per_edge;
    // four tiles at a time per wave: one pass over the sorted list serves all four (the list entries are
    // read once, the four compaction chains are independent)
    for (int t0 = wave * 4; t0 < ntl; t0 += 16) {
      float tXlo[4], tXhi[4], tYlo[4], tYhi[4];
      int32_t* out[4];
      int n[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = t0 + q;
        const int tx = (st % bin_sx) * per_edge + (t % per_edge), ty = (st / bin_sx) * per_edge + (t / per_edge);
        const bool in = t < ntl && tx < tiles_x && ty < tiles_y && (!restricted || owned(tx, ty));
        const int mi0 = tx * kTile, mj0 = ty * kTile;
        // (a tile outside the frame, or not this shard's, gets no output and bounds NOTHING overlaps: NaN, every
        // compare false.  Round 4's (+inf, -inf) "empty" bounds were overlapped by the (-inf .. +inf) box of a slab
        // with a Gaussian around the camera (k_project), whose id then went to out[q] = NULL: a write to address 4
        // whenever such a slab met a frame whose tile grid is not whole supertiles -- found in round 5 by the
        // tile-parallel fuzzer, 200 x 1 pixels with the camera inside a Gaussian)
        const float none = __builtin_nanf("");
        tXlo[q] = in ? ((float)mi0 + 0.5f) - hw : none;
        tXhi[q] = in ? ((float)min(mi0 + kTile, width) - 0.5f) - hw : none;
        tYlo[q] = in ? ((float)mj0 + 0.5f) - hh : none;
        tYhi[q] = in ? ((float)min(mj0 + kTile, height) - 0.5f) - hh : none;
        out[q] = in ? tile_bin + (size_t)(ty * tiles_x + tx) * kTileBinCap : nullptr;
        n[q] = 0;
      }
      for (int base = 0; base < cnt; base += kWave) {
        const int e = base + lane;
        float4 b = make_float4(kInf, kInf, -kInf, -kInf);
        int id = 0;
        if (e < cnt) { const int at = s_perm[e]; b = s_bb[at]; id = s_id[at]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool keep = b.x <= tXhi[q] && b.z >= tXlo[q] && b.y <= tYhi[q] && b.w >= tYlo[q];
          const unsigned long long km2 = __ballot(keep);
          const int at = n[q] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km2, 0u));
          if (keep && out[q] && at < kTileBinCap - 1) out[q][1 + at] = id;
          n[q] += __popcll(km2);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lane == 0 && out[q]) out[q][0] = n[q] <= kTileBinCap - 1 ? n[q] : -1;
    }
  }
}

struct Rgb { float r, g, b; };
// wave-wide max / min of an unsigned value (DPP within rows of 16, then the four rows through SGPRs)
__device__ __forceinline__ unsigned wave_umax(unsigned h) {
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0xB1, 0xF, 0xF, false));    // quad_perm 1,0,3,2
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x4E, 0xF, 0xF, false));    // quad_perm 2,3,0,1
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x141, 0xF, 0xF, false));   // row_half_mirror
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x140, 0xF, 0xF, false));   // row_mirror
  return max(max((unsigned)__builtin_amdgcn_readlane((int)h, 0), (unsigned)__builtin_amdgcn_readlane((int)h, 16)),
             max((unsigned)__builtin_amdgcn_readlane((int)h, 32), (unsigned)__builtin_amdgcn_readlane((int)h, 48)));
}
__device__ __forceinline__ unsigned wave_umin(unsigned h) { return ~wave_umax(~h); }
// exp(w - 3) for the exponent w = delta/(4A) in [0, 3] of a hit (gaussian.py:197 at the chord midpoint): one
// multiply-add and v_exp_f32 (2^x, 1 ulp) -- the argument is in [-4.33, 0], nowhere near the
// denormal / overflow handling that libm's expf spends another eight instructions on
__device__ __forceinline__ float exp_m3(float w) {
  return __builtin_amdgcn_exp2f(fmaf(w, 1.4426950408889634f, -4.3280851226668902f));
}
// HIP's __any() goes through an int (v_cndmask 0/1 + v_cmp_ne): the ballot form is the compare itself
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// Per-lane list of the RTGS_MAX_DEPTH smallest entry distances: three register arrays (t1 bits,
// Gaussian index, exponent), sorted ascending, empty slots = +inf.  Keys are the t1 bits alone
// (t1 > 0, so unsigned order == float order); exactly equal t1 (a few dozen pixels per frame)
// keep their ARRIVAL order, and k_trace makes that order a function of the tile alone (slabs
// nearest first, lanes in index order): the same in every instantiation and for every sharding.
//
// Insertion is chain-free: the new key is compared with all 16 ORIGINAL slots at once (independent compares; the
// results are monotone because the list is sorted), then every slot picks its own value, its left neighbour's or the
// new element.  (Round 1 measured the alternatives on c3: serial bubble with v_cmpx + v_swap_b32 0.301 ms vs 0.293,
// select bubbles on 64- / 32-bit keys 0.374 / 0.372, LDS append lists behind a front-to-back sort 1.21: HISTORY.md.)

// The list keeps only TWO register arrays per lane: the t1 bits and a payload
// (index << 4 | slot).  The third field, the exponent delta/(4A), lives in LDS in one of the lane's
// 16 slots; the slots of a list are a permutation of 0..15 (slot k at start), and a new entry
// takes the slot of the entry that falls off the end, so the exponent never moves: 16 compares +
// 64 selects instead of 16 + 96, and 32 list registers instead of 48 (96 -> 80 VGPRs).
// kN < kK: only the first kN slots are touched -- valid while slot kN-1 is empty in every lane of
// the wave (the caller tracks that with one ballot per insertion): lists are short for most of
// a tile's candidates (c3: 8 hits per ray on average), and a short insertion is half the work.
// list_insert2<kLo, kHi> updates slots kLo .. kHi-1 from the ORIGINAL values of slots kLo-1 .. kHi-1:
// the upper range runs first (and only once the lists have grown into it), then the lower one.
template <int kLo, int kHi, int kK>
__device__ __forceinline__ void list_insert2(unsigned (&kt)[kK], unsigned (&kp)[kK], unsigned tn, unsigned pn) {
  static_assert(kLo >= 0 && kLo < kHi && kHi <= kK, "range");
  constexpr int kB = kLo > 0 ? kLo - 1 : 0;   // lowest slot compared
  bool lt[kHi - kB];
#pragma unroll
  for (int k = kB; k < kHi; ++k) lt[k - kB] = tn < kt[k];
#pragma unroll
  for (int k = kHi - 1; k > kB; --k) {
    // the list is sorted, so the new key of slot k is the median of (left neighbour, itself, new key):
    // one v_med3_u32 instead of two selects (c3 0.1506 -> 0.1388 ms)
    kt[k] = max(min(kt[k - 1], kt[k]), min(max(kt[k - 1], kt[k]), tn));
    kp[k] = lt[k - 1 - kB] ? kp[k - 1] : (lt[k - kB] ? pn : kp[k]);
  }
  if constexpr (kLo == 0) {
    kt[0] = min(kt[0], tn);
    kp[0] = lt[0] ? pn : kp[0];
  }
}

// gaussian.py:149-163, including the y_33 quirk (5 z^2 - 3 z).
__device__ __forceinline__ void sh_basis(float x, float y, float z, float (&Y)[RTGS_SH_COEFFS]) {
  constexpr float c0 = 0.97720502380583984f, c1 = 2.1850968611841584f, c2 = 1.2615662610100802f,
                  c3 = 2.3601743597065673f, c4 = 5.7812228852811081f, c5 = 1.8281831978578629f,
                  c6 = 1.4927053303604616f;
  Y[0] = 0.5f * c0 * y;
  Y[1] = 0.5f * c0 * z;
  Y[2] = 0.5f * c0 * x;
  Y[3] = 0.5f * c1 * x * y;
  Y[4] = 0.5f * c1 * y * z;
  Y[5] = 0.25f * c2 * (3.f * z * z - 1.f);
  Y[6] = 0.5f * c1 * x * z;
  Y[7] = 0.25f * c1 * (x * x - y * y);
  Y[8] = 0.25f * c3 * y * (3.f * x * x - y * y);
  Y[9] = 0.5f * c4 * x * y * z;
  Y[10] = 0.25f * c5 * y * (5.f * z * z - 1.f);
  Y[11] = 0.25f * c6 * (5.f * z * z - 3.f * z);
  Y[12] = 0.25f * c5 * x * (5.f * z * z - 1.f);
  Y[13] = 0.25f * c4 * (x * x - y * y) * z;
  Y[14] = 0.25f * c3 * x * (x * x - 3.f * y * y);
}

// One ray against one projected record, operation for operation what k_trace's candidate loop does
// with the record in SGPRs (same fmaf chains, same v_rcp / v_sqrt): t1 bits -- ~0u when the ray
// misses or t1 <= start -- and the exponent delta/(4A).  Used by the tie fix-up at the end of
// k_trace, where a lane holds a Gaussian and the pixel is wave-uniform; t1 MUST come out bit for bit
// as in the candidate loop, because `t1 > start` is what keeps an entry from being found twice.
__device__ __forceinline__ unsigned pair_eval(const float4* __restrict__ rec, const float4* __restrict__ tail, int g,
                                              float X, float Y, float dlen, float start, float& wn) {
  const float4* rp = rec + (size_t)g * (kRecFloats / 4);
  const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
  float x = X - q0.x;
  float y = Y - q0.y;
  float gx = q0.w;
  const bool gen = __float_as_uint(q3.y) != 0u;
  float A_gen = 0.f;
  if (gen) {
    const float4* tp = tail + (size_t)g * (kTailFloats / 4);
    const float4 t0 = tp[0], t1v = tp[1];
    if (t1v.w != 0.f) { const float t = x; x = y; y = t; }
    gx = fmaf(-t1v.z, y, q0.w);
    A_gen = fmaf(y, fmaf(t1v.y, y, t0.z), fmaf(x, fmaf(t1v.x, y, fmaf(t0.w, x, t0.y)), t0.x));
  }
  const float u = fmaf(q1.z, y, x);
  float e = fmaf(x, gx, q0.z);
  e = fmaf(y, fmaf(-q1.w, y, q1.x), e);
  e = fmaf(-(q1.y * u), u, e);
  const float bq = fmaf(q2.z, y, fmaf(q2.y, x, q2.x));
  const bool hit = e > 0.f && bq < 0.f;
  const float t1 = q2.w * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(e) - bq) * dlen;
  const float A = gen ? A_gen : fmaf(bq, bq, -e) * q3.x;
  wn = e * __builtin_amdgcn_rcpf(A);
  return (hit && t1 > start && __float_as_uint(t1) < 0x7F800000u) ? __float_as_uint(t1) : ~0u;
}

// colour += sum_q Y_q * sh_q (gaussian.py:165-181) in explicit fma chains: the single-entry paths (listed peel, tie
// fix-up) and the list forms of scenes up to 262 k Gaussians; the 1 M-Gaussian list forms keep `+=` (within 1 ulp per term of this)
__device__ __forceinline__ void sh_add(float4& co, const float (&Y)[RTGS_SH_COEFFS], const float* __restrict__ sp) {
#pragma unroll
  for (int q = 0; q < RTGS_SH_COEFFS; ++q) {
    co.x = fmaf(Y[q], sp[3 * q + 0], co.x);
    co.y = fmaf(Y[q], sp[3 * q + 1], co.y);
    co.z = fmaf(Y[q], sp[3 * q + 2], co.z);
  }
}

// kList (rtgs_trace_lists): the launch stores every ray's sorted distinct entries instead of compositing them
// (an instantiation of its own: as a run-time branch it cost the main one five spilled VGPRs).
// kMulti (rtgs_shard_frames): frames batched into one launch, blockIdx.y = frame (an instantiation of its own: the
// per-frame pointer offsets cost the 72-VGPR main form 20 spilled VGPRs when they were unconditional).
// Round 5 retired three forms the measurements had left without a case (HISTORY.md): two waves per 8x8 tile (the "split" form:
// <= 5 % for a LONE launch below 700 tiles, nothing with two frames in flight, and batched launches have no lone small
// launch), the traversal cut of t_cutoff (never paid on any scene tried; the compositing cut stays) and the
// one-entry list (kK = 1: single peels go through the peel lists, the fall-backs through this kernel with depth = 1).
template <bool kHasSh, bool kBig, bool kList = false, bool kMulti = false>
__global__ __launch_bounds__(kTraceBlock,
                             kHasSh ? (kBig ? RTGS_WAVES_PER_SIMD_SH : RTGS_WAVES_PER_SIMD)
                                    : (!kBig ? RTGS_WAVES_PER_SIMD_MAIN
                                             : RTGS_WAVES_PER_SIMD)) void k_trace(const TraceArgs a,
                                                          const float4* __restrict__ bbox,
                                                          const float4* __restrict__ slab_bbox,
                                                          const float4* __restrict__ super_bbox,
                                                          const float4* __restrict__ rec,
                                                          const float4* __restrict__ tail,
                                                          const float* __restrict__ tmin,
                                                          const float4* __restrict__ colop,
                                                          const float* __restrict__ sh) {
  constexpr int kK = RTGS_MAX_DEPTH;   // entries kept per ray
  using TN = Tune<kBig>;
  constexpr int kSlabIters = TN::kSlabIters, kSlabStep = TN::kSlabStep, kSlabCap = TN::kSlabCap,
                kSlabsPerWave = TN::kSlabsPerWave, kListCap = TN::kListCap,
                kWaves = TN::kWaves, kBlock = TN::kBlock;
  // the survivor list's LDS pool
  // happens after the last barrier of the candidate loop
  // (hand-over in two halves of kK/2 entries x {t1, payload}: 16 KB instead of 32)
  constexpr int kListF4 = kListCap / 4;          // int indices; the 16-B bounds are re-read (L1) by the wave test
  constexpr int kPoolF4 = kListF4;
  __shared__ float4 s_pool[kPoolF4];
  int* const s_idx = reinterpret_cast<int*>(s_pool);
  // exponents of the list entries, [slot][thread]: conflict-free whatever slots the lanes use
  static_assert(kK == 16, "the payload keeps the slot in 4 bits");
  // payload = entry << kSlotBits | slot field
  constexpr int kSlotBits = 4;
  constexpr unsigned kSlotMask = (1u << kSlotBits) - 1u;
  __shared__ float s_w[kK * kBlock];
  // surviving slab ids: 16 bits are enough below kBigSceneSlabs (4 096) slabs
  using SlabId = typename std::conditional<kBig, int, unsigned short>::type;
  static_assert(kBig || kBigSceneSlabs <= 65536, "slab ids of small scenes are stored in 16 bits");
  // SH staging (round 4, kStage): ONE staging pass per WORKGROUP for the SH forms of scenes up to 262 k Gaussians.  The
  // 16x16 tile's referenced Gaussians (40-60: neighbouring 8x8 tiles share half of theirs; up to kStageSlots of them)
  // are identified by their position in the tile's survivor list -- the list entries of these forms carry that
  // position instead of the Gaussian index, made absolute (kAbs) before a list refill --, deduplicated by a byte map
  // over the list, fetched once by LDS-DMA (lane q < 45 brings coefficient q) and read per lane by all four waves.
  // Slots are 52 floats apart: slot s starts at bank (52 s) mod 64 = 0, 52, 40, 28, ..., so sixteen slots' b128 reads
  // tile all 64 banks (48 apart they pile onto four: c3 + SH 0.2905 -> 0.2901 ms with 48, 0.2713 with 52, 0.2617 with 96
  // slots instead of 64).  The staging area shares its LDS with the slab list and the group list, which are dead once
  // a tile's last round of expansion is behind its barrier.  Not for the 1 M-Gaussian form (built, with colour + opacity
  // staged too and 80 slots = what four workgroups per CU leave: c5 1.304 -> 1.459 ms; its tiles name 66.8 distinct
  // Gaussians on average, 42 % of them more than 80, and a wave with one slot-less lane runs both sources).
  constexpr bool kStage = RTGS_SH_STAGE != 0 && kHasSh && !kBig && !kList;
  // (round 5, profiles/r05_c3sh_occupancy.txt: 64 slots + a launch bound of five waves per SIMD -- five workgroups per CU
  // by LDS and by registers -- needs 20 spilled VGPRs in the shading loop: c3 + SH k_trace 0.2397 -> 0.3679 ms;
  // 64 slots at four waves change nothing)
  constexpr int kStageSlots = 96, kStageStride = 52;
  constexpr unsigned kAbs = 1u << 31;     // kStage payloads: the id field is a Gaussian index (else: a survivor-list slot)
  constexpr int kStageBytes = kStageSlots * kStageStride * (int)sizeof(float);
  static_assert(kStageSlots < 255, "staging slots are bytes, 0xFF = none");
  constexpr int kSlabBytes = (kSlabCap * (int)sizeof(SlabId) + 15) & ~15;
  constexpr int kSuperInts = (kBig ? TN::kSlabIters * kBlock : 1);
  constexpr int kMiscBytes = kSlabBytes + kSuperInts * 4 > kStageBytes ? kSlabBytes + kSuperInts * 4 : kStageBytes;
  __shared__ float4 s_misc[kStage ? kMiscBytes / 16 + 1 : 1];
  __shared__ SlabId s_slab_own[kStage ? 1 : kSlabCap];
  SlabId* const s_slab = kStage ? reinterpret_cast<SlabId*>(s_misc) : s_slab_own;
  // kStage: which survivor-list slots some ray's list refers to (then: their staging slot, 0xFF = none), the
  // Gaussian of every staging slot
  __shared__ unsigned char s_ref[kStage ? kListCap : 1];
  __shared__ int s_stg_g[kStage ? kStageSlots : 1];
  __shared__ int s_nref[kStage ? 1 + kWavesPerTile : 1];
  // depth keys of the first kSortMax listed slabs: only alive between the scan and the sort, when
  // the survivor pool is idle (every batch ends with a barrier) -- they share its LDS
  constexpr int kSortMax = kSortMaxAsk < kListCap ? kSortMaxAsk : kListCap;   // kListCap floats fit the pool
  float* const s_skey = reinterpret_cast<float*>(s_pool);
  __shared__ int s_wcnt[2][kWaves];
  constexpr int kRegions = kWavesPerTile;            // expander waves = regions of the survivor list
  constexpr int kRegionCap = kListCap / kRegions;
  // slabs one wave has in flight: more only pay where latency counts (the two-wave form's small launches)
  constexpr int kExpand = TN::kExpand;
  static_assert(kRegionCap >= kSlab, "a region must hold one whole slab");
  __shared__ int s_rcnt[kRegions], s_rmore[kRegions], s_rfirst;
  constexpr int kSuperStep = kSlabIters * kBlock;   // slab groups scanned per outer step
  constexpr bool kTwo = kBig;   // two-level slab scan (small scenes: c3 0.260 vs 0.261 ms flat, one barrier more)
  static_assert(kSuperInts == (kTwo ? kSuperStep : 1), "group list size");
  __shared__ int s_super_own[kStage ? 1 : kSuperInts];
  int* const s_super = kStage ? reinterpret_cast<int*>(reinterpret_cast<char*>(s_misc) + kSlabBytes) : s_super_own;
  __shared__ int s_wcnt2[2][kWaves];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile bounds stay in SGPRs
  // frames batched into one launch (rtgs_shard_frames): blockIdx.y = frame; everything per-frame lies
  // blockIdx.y * frame_stride bytes behind frame 0's (zero offset for the one-frame launches)
  static_assert(!(kMulti && kList), "batched frames composite: the list form is one frame per launch");
  const int32_t* f_bin_count = a.bin_count;
  const int32_t* f_bin_list = a.bin_list;
  const int32_t* f_tile_bin = a.tile_bin;
  if constexpr (kMulti) {
    const size_t fo = (size_t)blockIdx.y * a.frame_stride;
    bbox = at_frame(bbox, fo); slab_bbox = at_frame(slab_bbox, fo); super_bbox = at_frame(super_bbox, fo);
    rec = at_frame(rec, fo); tail = at_frame(tail, fo); tmin = at_frame(tmin, fo);
    if (f_bin_count) f_bin_count = at_frame(f_bin_count, fo);
    f_bin_list = at_frame(f_bin_list, fo);
    if (f_tile_bin) f_tile_bin = at_frame(f_tile_bin, fo);
  }
  const int ltile = a.tile_order ? a.tile_order[blockIdx.x] : (int)blockIdx.x;
  if ((unsigned)ltile >= gridDim.x) return;   // a malformed caller-supplied order must not write out of bounds
  const int tile = a.tile_list ? a.tile_list[ltile] : a.tile_first + ltile * a.tile_step;
  if ((unsigned)tile >= (unsigned)a.total_tiles) return;   // (a malformed caller-supplied list)
  const int mi0 = (tile % a.tiles_x) * kTile;
  const int mj0 = (tile / a.tiles_x) * kTile;
  const int sub = wave % kWavesPerTile;    // which 8x8 tile of the macro-tile
  const int wi0 = mi0 + (sub % kWavesX) * kWaveTile;
  const int wj0 = mj0 + (sub / kWavesX) * kWaveTile;
  const int pi = wi0 + (lane >> 3);
  const int pj = wj0 + (lane & 7);
  const bool in_img = pi < a.width && pj < a.height;
  const bool wave_live = wi0 < a.width && wj0 < a.height;

  const float hw = 0.5f * (float)a.width, hh = 0.5f * (float)a.height;
  const float X = ((float)pi + 0.5f) - hw;   // exact: half-integers
  const float Y = ((float)pj + 0.5f) - hh;
  // bounds of pixel centres covered by the macro-tile and by this wave's tile
  const float mXlo = ((float)mi0 + 0.5f) - hw;
  const float mXhi = ((float)min(mi0 + kTile, a.width) - 0.5f) - hw;
  const float mYlo = ((float)mj0 + 0.5f) - hh;
  const float mYhi = ((float)min(mj0 + kTile, a.height) - 0.5f) - hh;
  const float wXlo = ((float)wi0 + 0.5f) - hw;
  const float wXhi = ((float)min(wi0 + kWaveTile, a.width) - 0.5f) - hw;
  const float wYlo = ((float)wj0 + 0.5f) - hh;
  const float wYhi = ((float)min(wj0 + kWaveTile, a.height) - 0.5f) - hh;

  // |D| of this pixel's unnormalised camera-space direction D = (X/fx, Y/fy, -1): the
  // reference's t is along normalize(D) (camera.py:49-50), ours along D.
  const float dcx = X * a.inv_fx, dcy = Y * a.inv_fy;
  const float dlen = sqrtf(dcx * dcx + dcy * dcy + 1.0f);

  float start = 0.f;
  if ((a.flags & RTGS_TRACE_RESUME) && in_img) start = a.start[((size_t)pi * a.height + pj) * a.start_stride];

  unsigned key[kK], kidx[kK];
  constexpr unsigned kEmpty = 0x7F800000u;  // +inf
#pragma unroll
  for (int k = 0; k < kK; ++k) {
    key[k] = kEmpty;
    kidx[k] = (unsigned)k;   // payload = entry << kSlotBits | slot: the slots of a list are a permutation of 0..15
  }
  [[maybe_unused]] float* const s_wl = s_w + tid;
  constexpr bool kHorizon = RTGS_HORIZON != 0 && kK == RTGS_MAX_DEPTH;
  constexpr int kChunksMax = kListCap / kWave;
  constexpr bool kLookahead = RTGS_LOOKAHEAD == 2 || (RTGS_LOOKAHEAD == 1 && kBig);
  static_assert(RTGS_TIER > 0 && RTGS_TIER < RTGS_MAX_DEPTH, "the tier boundary lies inside the list");
  constexpr int kTier = RTGS_TIER;
  [[maybe_unused]] bool tier_short = kTier < kK;   // wave-uniform: slot kTier-1 still empty in every lane

  // set bits of a wave mask below this lane: v_mbcnt_lo/hi straight from the SGPR pair
  auto below = [](unsigned long long m) -> int {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  };
  const int nslabs = (a.n + kSlab - 1) / kSlab;
  // depth keys live behind the slab and group bounds (see slab_key_of)
  const float* const slab_key = reinterpret_cast<const float*>(slab_bbox + nslabs + (nslabs + kSuper - 1) / kSuper);
#if RTGS_COUNT
  unsigned n_iter = 0, n_pass = 0, n_ins = 0, n_hit_lanes = 0, n_ins_lanes = 0, n_exp = 0, n_scan = 0;
  [[maybe_unused]] unsigned n_lds = 0, n_mem = 0, n_named = 0, n_over = 0;
#endif
#if RTGS_STAMPS
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
  int c_iter = 0, c_hit = 0, c_ins = 0, c_slabs = 0;
  unsigned long long t_ins = 0, t_loop0 = 0, t_front = 0, t_exp = 0, t_bar = 0, t_merge = 0, t_mark = 0;
  int c_rounds = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: calibrates the shader clock
  st[0] = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (kStage) {
#pragma unroll
    for (int q = 0; q < kListCap / kBlock; ++q) s_ref[q * kBlock + tid] = 0;   // (ordered before any mark by the scan's barriers)
  }
  int scnt = 0;  // surviving slabs listed so far: identical in every thread
  // this tile's supertile bin (k_bin), if the frame has bins and the bin did not overflow
  const int bin_at = ((tile / a.tiles_x) >> a.bin_shift) * a.bin_sx + ((tile % a.tiles_x) >> a.bin_shift);
  const int bin_n = f_bin_count ? __builtin_amdgcn_readfirstlane(f_bin_count[bin_at]) : -1;
  const bool binned = bin_n >= 0;
  const int32_t* const bin = f_bin_list + (size_t)bin_at * a.bin_stride;
  // this tile's own list (k_bin wrote it iff it wrote the supertile's bin): the count and every thread's entry
  // are loaded together with bin_n above -- ONE round trip instead of bin count -> bin ids -> slab bounds
  // (when the frame's bins were NOT built -- a small shard -- these two loads read whatever the region
  // holds: in bounds, and ignored, `binned` below is false then; guarding them with `binned` would put them behind
  // bin_n's round trip again)
  int tb_n = -1, tb_id = 0;
  if constexpr (RTGS_TILE_BINS != 0) {
    if (f_tile_bin) {
      const int32_t* const tb = f_tile_bin + (size_t)tile * kTileBinCap;
      tb_n = __builtin_amdgcn_readfirstlane(tb[0]);
      if (tid < kTileBinCap - 1) tb_id = tb[1 + tid];
    }
  }
  const bool tbinned = binned && tb_n >= 0;
  const int nsuper = (kTwo && !binned) ? (nslabs + kSuper - 1) / kSuper : 1;
  int par = 0;
  for (int ub = 0, upar = 0; ub < nsuper; ub += kSuperStep, upar ^= 1) {
  int nitems = tbinned ? 0 : (binned ? bin_n : nslabs);
  if constexpr (kTwo) {
  if (!binned) {
    // ---- level 1: bounds of kSuper-slab groups; survivors in stream order into s_super ----
    float4 gbb[kSlabIters];
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it) {
      const int u = ub + (wave * kSlabIters + it) * kWave + lane;
      gbb[it] = make_float4(kInf, kInf, -kInf, -kInf);
      if (u < nsuper) gbb[it] = super_bbox[u];
    }
    unsigned long long um[kSlabIters];
    int ucnt = 0;
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it) {
      const bool keep = gbb[it].x <= mXhi && gbb[it].z >= mXlo && gbb[it].y <= mYhi && gbb[it].w >= mYlo;
      um[it] = __ballot(keep);
      ucnt += __popcll(um[it]);
    }
    if (lane == 0) s_wcnt2[upar][wave] = ucnt;
    __syncthreads();
    int uoff = 0, utot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int c = s_wcnt2[upar][w];
      uoff += (w < wave) ? c : 0;
      utot += c;
    }
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it) {
      if ((um[it] >> lane) & 1ull) s_super[uoff + below(um[it])] = ub + (wave * kSlabIters + it) * kWave + lane;
      uoff += __popcll(um[it]);
    }
    __syncthreads();
    nitems = utot * kSuper;
  }
  }
  const bool last_outer = ub + kSuperStep >= nsuper;
  for (int sbase = 0; sbase < nitems || (last_outer && sbase == 0); sbase += kSlabStep, par ^= 1) {
    bool last = true;
    if (tbinned) {
      // the tile's list as k_bin wrote it: already filtered by this tile's bounds and nearest first
      if (tid < tb_n) s_slab[tid] = (SlabId)tb_id;
      scnt = tb_n;
    } else {
    // ---- scan slab bounds; keep the slabs that touch the macro-tile.  The list is built
    // without atomics: every wave publishes its count, one barrier later every thread derives
    // the same offsets (s_wcnt is double-buffered by step parity).
    float4 sb[kSlabIters];
    float sk[kSlabIters];
    int slid[kSlabIters];
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it) {
      // Every wave takes kSlabIters * 64 CONSECUTIVE items, so that the survivors -- listed wave by wave, step by step --
      // come out in item order: a bin's order (nearest slab first) survives the filter.  (Up to round 5 step `it` of
      // thread `tid` took item it * 256 + tid: wave 0 listed items 0-63 and 256-319 before wave 1's 64-127 -- harmless
      // where the list is sorted afterwards, but a tile that falls back to its supertile's bin met its slabs in a
      // permuted order: farther candidates first, and an exact tie across two slabs resolved against the list's order,
      // which the tie fix-up's find_next goes by.  Found with coloured twins, tools/debug_twin_seed.py 50062.)
      const int item = sbase + (wave * kSlabIters + it) * kWave + lane;
      int sl = item;
      if (binned) sl = item < nitems ? bin[item] : nslabs;          // listed nearest first by k_bin
      else if constexpr (kTwo) sl = item < nitems ? s_super[item / kSuper] * kSuper + (item % kSuper) : nslabs;
      slid[it] = sl;
      sb[it] = make_float4(kInf, kInf, -kInf, -kInf);
      sk[it] = 0.f;
      if (sl < nslabs) { sb[it] = slab_bbox[sl]; sk[it] = slab_key[sl]; }
    }
#if RTGS_COUNT
    n_scan += (unsigned)kSlabIters * kWave;
#endif
    unsigned long long keep_m[kSlabIters];
    int wcnt = 0;
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it) {
      const bool keep = sb[it].x <= mXhi && sb[it].z >= mXlo && sb[it].y <= mYhi && sb[it].w >= mYlo;
      keep_m[it] = __ballot(keep);
      wcnt += __popcll(keep_m[it]);
    }
    if (lane == 0) s_wcnt[par][wave] = wcnt;
    __syncthreads();
    int off = scnt;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int c = s_wcnt[par][w];
      off += (w < wave) ? c : 0;
      scnt += c;
    }
#pragma unroll
    for (int it = 0; it < kSlabIters; ++it) {
      if ((keep_m[it] >> lane) & 1ull) {
        const int at = off + below(keep_m[it]);
        s_slab[at] = (SlabId)slid[it];
        if (at < kSortMax) s_skey[at] = sk[it];
      }
      off += __popcll(keep_m[it]);
    }
    last = last_outer && sbase + kSlabStep >= nitems;
    if (scnt + kSlabStep <= kSlabCap && !last) continue;
    }
    __syncthreads();  // the slab list is complete
    // (a bin is sorted already; filtering keeps the order.  Marked unlikely: c3 0.1286 -> 0.1278 ms, c5 1.295 -> 1.2885)
    if (__builtin_expect(!binned && kSortMax > 0 && scnt > 1 && scnt <= kSortMax, 0)) {
      // nearest slabs first: rank by (key, list position), kSortPer listed slabs per thread (one for
      // lists up to the workgroup size, i.e. practically always: c3 28 slabs per tile on average,
      // 116 at most; c5 58 / 235).  (One wave exchanging the keys by v_readlane instead of LDS, and
      // skipping lists shorter than 8 or 16: all slower, 0.275-0.281 vs 0.270 ms on c3.)
      constexpr int kSortPer = (kSortMax + kBlock - 1) / kBlock;
      int my_slab[kSortPer], rank[kSortPer];
      // (laundered: the compiler formed (tid, tid + 256) at the top of the kernel, kept the pair through everything
      // and -- one register short in the main form -- spilled it to scratch for this cold block)
      int tid_s = tid;
      asm volatile("" : "+v"(tid_s));
#pragma unroll
      for (int q = 0; q < kSortPer; ++q) {
        const int e = tid_s + q * kBlock;
        my_slab[q] = 0; rank[q] = 0;
        if (e < scnt && (q == 0 || scnt > q * kBlock)) {
          my_slab[q] = s_slab[e];
          const float my_key = s_skey[e];
          for (int j = 0; j < scnt; ++j) {
            const float kj = s_skey[j];
            rank[q] += (kj < my_key || (kj == my_key && j < e)) ? 1 : 0;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kSortPer; ++q)
        if (tid_s + q * kBlock < scnt) s_slab[rank[q]] = (SlabId)my_slab[q];
      __syncthreads();
    }
#if RTGS_STAMPS
    if (st[1] == 0) st[1] = __builtin_amdgcn_s_memtime();
    c_slabs += scnt;
#endif

    // ---- expand the listed slabs into the survivor list, then trace it; in ROUNDS only when the
    // list overflows.  The kRegions expander waves each own a contiguous share of the (sorted)
    // slab list and a region of the LDS list: no cross-wave prefix, ONE barrier between
    // expansion and tracing, and the list order -- region by region, slab by slab, lane by lane
    // -- is a function of the tile alone (equal t1 keys resolve identically on every run, for
    // every sharding and in both workgroup forms).
    // Rounds keep that order too (round 5): a round TRACES the regions up to the first one that filled up before its
    // share was done; the regions behind it keep what they have expanded and wait (nothing is expanded twice), the full
    // one goes on with its share in the next round -- so the candidates always arrive in the slab list's order.  (Up to
    // here every round traced every region: round 1 = the head of every share, round 2 = the tails, and a Gaussian of the
    // second share could arrive before one of the first share's tail.  Only exact ties see the order: the tie fix-up's
    // find_next resolves them by the list's order, so a ray's 13th entry could be one twin in a one-pass trace and the
    // other in the 13th single-peel launch -- found with coloured twins in tiles of > 512 survivors,
    // tests/test_gpu_parallel_fuzz.py::twin_paths_case.)
    const int share = (scnt + kRegions - 1) / kRegions;
    int pos = min(scnt, wave * share);
    const int pend = wave < kRegions ? min(scnt, (wave + 1) * share) : pos;
    // (which region goes on is kept in LDS, not in a register: two more scalars alive through the candidate loop cost the
    // c3 form 2 % -- its SGPRs spill as it is.  After a batch's first round the ONLY region that expands is the one
    // that filled up: the ones before it are finished, the ones behind it still hold what they expanded in round one.
    // (A survivor list of 1024 instead of 512 -- fewer rounds -- changed nothing at c3 / cap / c2; 2048 costs a workgroup per CU.)
    // What this cost and why (A/B of library builds, gpurun_out/ab_*.txt): the first version was 1.8 % slower at c3 and
    // c2 -- not the rounds: the region counts, read under the new conditions, had become eight dependent 4-B LDS reads
    // with a wait each, ahead of the first chunk's loads on every tile's critical path.  Read unconditionally (two
    // 16-B reads as before) and combined as integer masks: k_trace as fast as before, the c3 frame within 0.8 %.)
    int more = 0;
    for (;;) {
#if RTGS_STAMPS
      t_mark = __builtin_amdgcn_s_memtime();
      ++c_rounds;
#endif
      // (a batch's first round -- for nearly every tile its only one -- reads no state: `more`, which the loop carries
      // anyway, says whether this is a later round)
      int first = 0;
      if (__builtin_expect(more != 0, 0)) first = __builtin_amdgcn_readfirstlane(s_rfirst);   // the region that goes on
      if (!more || wave == first) {
      int rcnt = 0;   // survivors this wave lists in its region this round (wave-uniform)
      while (pos < pend) {
        // kExpand slabs in flight: one lane per Gaussian, 16-B bounds, coalesced
        float4 gb[kExpand];
        int gbase[kExpand];
#pragma unroll
        for (int q = 0; q < kExpand; ++q) {
          // branch-free (clamped slab and lane: a repeated slab is not consumed below, lanes past n
          // are not kept), so that the loads really are in flight together
          gbase[q] = __builtin_amdgcn_readfirstlane((int)s_slab[min(pos + q, pend - 1)]) * kSlab;
          gb[q] = bbox[(unsigned)min(gbase[q] + lane, a.n - 1)];
        }
        bool full = false;
#pragma unroll
        for (int q = 0; q < kExpand; ++q) {
          if (pos < pend && !full) {
            const bool keep = gb[q].x <= mXhi && gb[q].z >= mXlo && gb[q].y <= mYhi && gb[q].w >= mYlo &&
                              gbase[q] + lane < a.n;
            const unsigned long long gm = __ballot(keep);
            const int c = __popcll(gm);
#if RTGS_COUNT
            n_exp += kWave;
#endif
            if (rcnt + c > kRegionCap) {
              full = true;   // this slab opens the region's next round
            } else {
              if (keep) s_idx[wave * kRegionCap + rcnt + below(gm)] = gbase[q] + lane;
              rcnt += c;
              ++pos;
            }
          }
        }
        if (full) break;
      }
      if (lane == 0 && wave < kRegions) { s_rcnt[wave] = rcnt; s_rmore[wave] = pos < pend ? 1 : 0; }
      }
      __syncthreads();
      int rc[kRegions];
      int total = 0;
      more = 0;
      {
        // (all eight values first, unconditionally -- two 16-B LDS reads as before; read under the conditions below
        // they became eight dependent 4-B reads, each with its own wait: 2 % of a c3 frame)
        int cnt_r[kRegions], more_r[kRegions];
#pragma unroll
        for (int r = 0; r < kRegions; ++r) {
          cnt_r[r] = __builtin_amdgcn_readfirstlane(s_rcnt[r]);     // wave-uniform: SGPRs
          more_r[r] = __builtin_amdgcn_readfirstlane(s_rmore[r]);
        }
        // regions below `first` are finished, regions behind the first full one wait.  (Integer masks: as bools the
        // compiler took these wave-uniform values through VGPRs and 60 instructions -- ahead of the first chunk's
        // loads, on every tile's critical path)
        int last_r = 0;
#pragma unroll
        for (int r = 0; r < kRegions; ++r) {
          const int on = (r >= first ? 1 : 0) & (more ^ 1);
          rc[r] = cnt_r[r] & -on;
          const int full = on & (more_r[r] != 0 ? 1 : 0);
          last_r |= r & -full;      // (at most one region is `full`)
          more |= full;
          total += rc[r];
        }
        // (every wave stores the same value; the next round reads it behind this round's closing barrier, and every
        // reader of the old value is past the barrier above)
        if (more && lane == 0) s_rfirst = last_r;
      }
#if RTGS_STAMPS
      if (st[2] == 0) st[2] = __builtin_amdgcn_s_memtime();
      t_exp += __builtin_amdgcn_s_memtime() - t_mark;
#endif
#if RTGS_ABLATE == 4
      total = 0;
#endif
      // ---- every wave: survivors vs its own 8x8 tile, then exact per-ray tests.  The regions
      // are walked as one list; the next chunk's bounds load while this chunk's candidates run.
      auto fetch = [&](int c, int& g, float4& wb, float& tm, [[maybe_unused]] int& ls) {
        const int v = c + lane;
        g = -1;
        ls = 0;
        tm = 0.f;
        wb = make_float4(kInf, kInf, -kInf, -kInf);
        if (v < total) {
          int slot = v, base = 0;
#pragma unroll
          for (int r = 0; r + 1 < kRegions; ++r) {
            base += rc[r];
            if (v >= base) slot = v - base + (r + 1) * kRegionCap;
          }
          g = s_idx[slot];
          ls = slot;
          wb = bbox[g];
          if constexpr (kHorizon) tm = tmin[g];
        }
      };
      if (wave_live && total > 0) {
        int g, gnext = -1;
        [[maybe_unused]] int ls, lsnext = 0;
        float4 wb, wbnext = make_float4(kInf, kInf, -kInf, -kInf);
        float tm, tmnext = 0.f;
        fetch(0, g, wb, tm, ls);
        for (int c = 0; c < total; c += kWave) {
          const bool pass = wb.x <= wXhi && wb.z >= wXlo && wb.y <= wYhi && wb.w >= wYlo;
          unsigned long long m = __ballot(pass);
          if constexpr (kHorizon) {
            // Horizon: once every ray of the tile holds kK entries, a candidate whose distance bound
            // lies at or beyond the farthest of their last entries cannot enter any list.  One wave
            // reduction per 64 candidates; the image does not depend on it (such a candidate would
            // fail `kn < key[kK-1]` in every lane).  Scenes with large background Gaussians
            // (capture-like, tools/cap_sweep.py) lose most of their candidates here.
            // (while some list is not full the horizon is +inf: one compare + ballot instead of the reduction)
            const bool open_list = wave_any(in_img && key[kK - 1] >= kEmpty);
            const unsigned hz = open_list ? kEmpty : wave_umax(in_img ? key[kK - 1] : 0u);   // t1 bits
            // (evaluated even while hz is still +inf: this chunk's loads are all consumed before
            // the next chunk's are issued, so nothing inside the candidate loop waits for memory)
            m &= __ballot(tm < __uint_as_float(min(hz, kEmpty)) || hz >= kEmpty);
          }
          if (c + kWave < total) fetch(c + kWave, gnext, wbnext, tmnext, lsnext);
#if RTGS_ABLATE >= 3
          asm volatile("" ::"s"(m));
          m = 0;
#endif
          [[maybe_unused]] float pf = 0.f;
          const unsigned long long img_m = __builtin_amdgcn_ballot_w64(in_img);
          while (m) {
            const int b = __builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));   // m &= m - 1 in one SALU
#if RTGS_COUNT
            ++n_iter;
#endif
#if RTGS_STAMPS
            ++c_iter;
            t_loop0 = __builtin_amdgcn_s_memtime();
#endif
            // the record's address is wave-uniform: scalar loads, operands straight from SGPRs
            const int gi = __builtin_amdgcn_readlane(g, b);
            // one 64-B line, ONE s_load_dwordx16 (loaded as a single vector so that the compiler
            // cannot sink part of it behind the first branch and wait twice)
            typedef float Rec16 __attribute__((ext_vector_type(16)));
            // 32-bit byte offset (n < 2^26, checked on the host): base + soffset, one SALU instead of four
            const Rec16 rq = *reinterpret_cast<const Rec16*>(reinterpret_cast<const char*>(rec) +
                                                             ((unsigned)gi * (unsigned)(kRecFloats * sizeof(float))));
            const float4 q0 = make_float4(rq[0], rq[1], rq[2], rq[3]), q1 = make_float4(rq[4], rq[5], rq[6], rq[7]),
                         q2 = make_float4(rq[8], rq[9], rq[10], rq[11]), q3 = make_float4(rq[12], rq[13], rq[14], rq[15]);
            // delta/4 = e_c + g.(x',y') - d1 (x' + l y')^2 - d2 y'^2 [- cr x y] about the record's
            // expansion point (k_project: LDL^T about the larger pivot)
            float x = X - q0.x;
            float y = Y - q0.y;
            float gx = q0.w;
            if constexpr (kLookahead) {
              // Look-ahead: one dword of the NEXT candidate's record, issued once this record has
              // arrived (the asm ties the address to its first use), so that the line is in the
              // scalar cache -- 55 % hit rate without -- when the next iteration asks for it.  The
              // value itself is only kept alive until then.
              asm volatile("" ::"s"(pf));
              if (m) {
                int gn = __builtin_amdgcn_readlane(g, __builtin_ctzll(m));
                asm volatile("" : "+s"(gn) : "v"(x));
                pf = *reinterpret_cast<const float*>(rec + (size_t)gn * (kRecFloats / 4));
              }
            }
            // class 1 (general conic, or the camera within ~3.3 sigma): cross term, (y, x) storage
            // and A's own polynomial come from the tail, behind a scalar branch
            const bool gen = __float_as_uint(q3.y) != 0u;   // scalar integer compare (the flag is 0.0f or 1.0f)
            float A_gen = 0.f;
            float invC_eff = q3.x;                          // (wave-uniform: an SGPR)
            if (__builtin_expect(gen, 0)) {
              const float4* tp = tail + (size_t)gi * (kTailFloats / 4);
              const float4 t0 = tp[0], t1v = tp[1];
              if (t1v.w != 0.f) asm volatile("v_swap_b32 %0, %1" : "+v"(x), "+v"(y));
              gx = fmaf(-t1v.z, y, q0.w);
              A_gen = fmaf(y, fmaf(t1v.y, y, t0.z), fmaf(x, fmaf(t1v.x, y, fmaf(t0.w, x, t0.y)), t0.x));
              invC_eff = 0.f;
            }
            const float u = fmaf(q1.z, y, x);
            float e = fmaf(x, gx, q0.z);
            e = fmaf(y, fmaf(-q1.w, y, q1.x), e);
            e = fmaf(-(q1.y * u), u, e);
            bool hit = e > 0.f && in_img;
#if RTGS_ABLATE == 2
            asm volatile("" ::"v"(e));
            continue;
#endif
            if ((__builtin_amdgcn_ballot_w64(e > 0.f) & img_m) == 0ull) continue;
#if RTGS_COUNT
            ++n_pass;
#endif
#if RTGS_STAMPS
            ++c_hit;
#endif
            // explicit fma chains: every instantiation (K = 1 / 16, one or two waves, SH or not)
            // must round t1 and the exponent identically, or two hits one ulp apart in t1 could
            // composite in a different order in a progressive sample() than in a one-pass trace
            const float bq = fmaf(q2.z, y, fmaf(q2.y, x, q2.x));
            hit = hit && bq < 0.f;   // the cone half in front of the camera
            // smaller root of A t^2 + 2 b t + C (gaussian.py:224) in its cancellation-free form
            // t1 = C / (sqrt(delta/4) - b): C > 0 and b < 0, so nothing is subtracted.
            // v_sqrt_f32 / v_rcp_f32 are 1 ulp: far inside the 1e-4 pixel tolerance (IEEE division / sqrt: +8 %).
            const float t1 = q2.w * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(e) - bq) * dlen;
            // scene.py:433: start < t1 < end, end = +inf for a camera ray (an overflowed t1 = +inf has the
            // bits of an empty slot and fails the insertion test below by itself; NaN fails `>`)
            const bool cand = hit && t1 > start;
            const unsigned kn = cand ? __float_as_uint(t1) : ~0u;
            const bool ins = kn < key[kK - 1];
#if RTGS_COUNT
            n_hit_lanes += (unsigned)__popcll(__ballot(cand));
            n_ins_lanes += (unsigned)__popcll(__ballot(ins));
            if (wave_any(ins)) ++n_ins;
#endif
            if (wave_any(ins)) {
#if RTGS_STAMPS
              ++c_ins;
              const unsigned long long ti0 = __builtin_amdgcn_s_memtime();
              t_front += ti0 - t_loop0;
#endif
              // A = d^T Sigma^-1 d from delta/4 = b^2 - A C (compact records), or its own polynomial
              // no select on the class (round 5: it kept the class test as a 64-bit mask -- compare, cselect, and, branch
              // per candidate; now compare + branch, c3 0.1288 -> 0.1273 ms): (b^2 - delta/4) * (1/C) + 0 for the compact
              // records (the product, rounded once: the same bits as the multiplication), 0 * (...) + A's own polynomial
              // for class 1
              const float A = fmaf(fmaf(bq, bq, -e), invC_eff, A_gen);
              const float wn = e * __builtin_amdgcn_rcpf(A);
#if RTGS_ABLATE == 1
              asm volatile("" ::"v"(kn), "v"(wn));
#else
              unsigned entry = (unsigned)gi;
              if constexpr (kStage) entry = (unsigned)__builtin_amdgcn_readlane(ls, b);   // the survivor-list slot, see kStage
              {
                // the new entry takes the exponent slot of the entry that falls off the end
                const unsigned slot = (tier_short ? kidx[kTier - 1] : kidx[kK - 1]) & kSlotMask;
                const unsigned pn = (entry << kSlotBits) | slot;
                if (ins) s_wl[(slot & 15u) * kBlock] = wn;
                // Slot k of a lane changes iff kn < key[k] (kn = ~0 where the lane does not insert), so the
                // lower half is untouched in the whole wave unless some lane has kn below key[kTier-1]:
                // one compare + ballot skips it -- candidates arrive nearest first, so the later ones
                // mostly land in the upper half (c3: 0.1634 -> 0.1551 ms).  Four blocks of four with a
                // ballot each lose again (0.211: too many small basic blocks), and three straight-line
                // cases (short / both halves / upper only) make the register allocator spill 107 VGPRs.
                // Fill stages of 4 / 8 / 12 / 16 slots instead of 8 / 16 (no extra ballots, same registers):
                // c3 0.1357-0.1370 -> 0.1376-0.1386, c5 1.299 -> 1.355 -- the branches cost more than the
                // 128 skipped instructions per wave.
                const bool low = kTier == kK || tier_short || wave_any(kn < key[kTier - 1]);
                if constexpr (kTier < kK) {
                  if (!tier_short) list_insert2<kTier, kK>(key, kidx, kn, pn);
                }
                if (low) list_insert2<0, kTier>(key, kidx, kn, pn);
                if constexpr (kTier < kK) {
                  if (tier_short) tier_short = !wave_any(key[kTier - 1] < kEmpty);
                }
              }
#endif
#if RTGS_STAMPS
              t_ins += __builtin_amdgcn_s_memtime() - ti0;
#endif
            }
          }
          g = gnext;
          wb = wbnext;
          tm = tmnext;
          if constexpr (kStage) ls = lsnext;
        }
      }
      // everyone is done with the list before the next round, the next scan or (two-wave form) the
      // list hand-over refills its LDS; after a tile's LAST round nothing does, and the waves go on
      // to their compositing without waiting for the slowest of the four
#if RTGS_STAMPS
      t_mark = __builtin_amdgcn_s_memtime();
#endif
      if constexpr (kStage) {
        if (more || !last) {   // (rare: a tile whose survivors overflow the list) the slots are about to be reused
#pragma unroll
          for (int k = 0; k < kK; ++k)
            if (!(kidx[k] & kAbs) && key[k] < kEmpty)
              kidx[k] = kAbs | ((unsigned)s_idx[(kidx[k] >> kSlotBits) & 0x3FFFFFFu] << kSlotBits) | (kidx[k] & kSlotMask);
        }
      }
      if (more || !last) __syncthreads();
#if RTGS_STAMPS
      t_bar += __builtin_amdgcn_s_memtime() - t_mark;
#endif
      if (!more) break;
    }
    scnt = 0;
  }
  }

#if RTGS_STAMPS
  t_mark = __builtin_amdgcn_s_memtime();
  st[3] = t_mark;      // (end of the candidate phase; the two-wave form's list merge used to sit between these two stamps)
#endif
  // ---- front-to-back compositing of the sorted list (ray_tracer.py:88-104) ----
  // The pixel's coordinates and linear index are formed again from the lane number (laundered, so
  // that the compiler cannot keep the prologue's copies alive through the candidate loop: they
  // were 5 of the 9 VGPRs it spilled to scratch, 49 MB of HBM writes per c3 frame).
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int pi_e = wi0 + (lane_e >> 3), pj_e = wj0 + (lane_e & 7);
  // The arguments the epilogue needs are read again from the kernarg segment behind a laundered
  // pointer: kept in SGPRs from the prologue they cost ~20 of the 102 through the whole candidate
  // loop (and that many v_writelane / v_readlane spills).
  using KArgs = const TraceArgs __attribute__((address_space(4)));
  KArgs* ea = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ea));
  const size_t pix = (size_t)pi_e * ea->height + pj_e;

  // The next entry of ONE ray beyond distance sf, found by the whole wave: every lane takes one Gaussian of a
  // slab that covers the pixel and evaluates it for THAT pixel with the candidate loop's own arithmetic
  // (pair_eval); of exactly tied entries the one that would have arrived first wins.  Wave-uniform results; false =
  // the ray misses.  Cold code: only the tie fix-up (below) calls it.
  [[maybe_unused]] auto find_next = [&](float Xf, float Yf, float dlf, float sf, unsigned& t_out, unsigned& g_out, float& w_out) -> bool {
    const int n_g = ea->n, nsl = (n_g + kSlab - 1) / kSlab;
    // the tile's own depth-sorted slab list when the frame has one, else every slab
    // (tile and bin recomputed from the kernarg segment: nothing of this path is kept alive through the candidate loop)
    const int32_t* tbl = nullptr;
    int nlist = nsl;
    if constexpr (RTGS_TILE_BINS != 0) {
      const int tile_f = ea->tile_list ? ea->tile_list[ltile] : ea->tile_first + ltile * ea->tile_step;
      const int bin_f = ((tile_f / ea->tiles_x) >> ea->bin_shift) * ea->bin_sx + ((tile_f % ea->tiles_x) >> ea->bin_shift);
      const size_t fo_f = kMulti ? (size_t)blockIdx.y * ea->frame_stride : 0;
      if (ea->tile_bin && ea->bin_count && __builtin_amdgcn_readfirstlane(at_frame(ea->bin_count, fo_f)[bin_f]) >= 0) {
        const int32_t* const tb = at_frame(ea->tile_bin, fo_f) + (size_t)tile_f * kTileBinCap;
        const int c = __builtin_amdgcn_readfirstlane(tb[0]);
        if (c >= 0) { tbl = tb + 1; nlist = c; }
      }
    }
    // Of exactly tied entries the one that would have ARRIVED first wins, as in the lists: slabs are visited by
    // (depth key, slab id), a slab's members in index order -- whatever order this search meets them in
    const float* const skey = reinterpret_cast<const float*>(slab_bbox + nsl + (nsl + kSuper - 1) / kSuper);
    unsigned bt = ~0u, bk = ~0u, bg = ~0u;   // this lane's best (t1 bits, slab key bits, Gaussian) beyond sf
    float bw = 0.f;
    for (int base = 0; base < nlist; base += kWave) {
      const int li = base + lane;
      int sl = -1;
      if (li < nlist) sl = tbl ? tbl[li] : li;
      bool touch = false;
      if (sl >= 0) {
        const float4 sb = slab_bbox[sl];
        touch = sb.x <= Xf && sb.z >= Xf && sb.y <= Yf && sb.w >= Yf;
      }
      unsigned long long sm = __builtin_amdgcn_ballot_w64(touch);
      while (sm) {
        const int sb_ = __builtin_ctzll(sm);
        sm &= sm - 1ull;
        const int slab_i = __builtin_amdgcn_readlane(sl, sb_);
        const int g = slab_i * kSlab + lane;
        if (g < n_g) {
          const float4 gb = bbox[g];
          if (gb.x <= Xf && gb.z >= Xf && gb.y <= Yf && gb.w >= Yf) {
            float w1;
            const unsigned tb1 = pair_eval(rec, tail, g, Xf, Yf, dlf, sf, w1);
            const unsigned kb1 = __float_as_uint(skey[slab_i]);     // (keys are >= 0 or +inf: their bits order like them)
            if (tb1 < bt || (tb1 == bt && (kb1 < bk || (kb1 == bk && (unsigned)g < bg)))) { bt = tb1; bk = kb1; bg = (unsigned)g; bw = w1; }
          }
        }
      }
    }
    const unsigned tmin1 = wave_umin(bt);
    if (tmin1 >= kEmpty) return false;
    const unsigned kmin = wave_umin(bt == tmin1 ? bk : ~0u);
    const unsigned gmin = wave_umin(bt == tmin1 && bk == kmin ? bg : ~0u);
    const int wl = __builtin_ctzll(__builtin_amdgcn_ballot_w64(bt == tmin1 && bk == kmin && bg == gmin));
    t_out = tmin1;
    g_out = gmin;
    w_out = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(bw), wl));
    return true;
  };
  [[maybe_unused]] auto lane_f = [](float v, int l) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); };

  if constexpr (kList) {
    {
      // ---- list form (rtgs_trace_lists): no compositing; the sorted DISTINCT entries of every ray go to the
      // peel lists, [tile][slot][thread], for rtgs_peel to composite one per call (the reference viewer's
      // sample() pattern, __main__.py:253-256: 16 calls, ONE traversal).  A tied twin is dropped here, and a
      // full list that came up short that way is topped up by the wave (the tie fix-up, list form).
      const int tile_l = ea->tile_list ? ea->tile_list[ltile] : ea->tile_first + ltile * ea->tile_step;
      const size_t lbase = (size_t)tile_l * (kK * kTraceBlock) + (size_t)(sub * kWave + lane_e);
      int cnt = 0;
      unsigned prev = 0u;
#pragma unroll
      for (int k = 0; k < kK; ++k) {
        const unsigned tb = key[k];
        const bool valid = in_img && tb < kEmpty && tb != prev;
        prev = tb;
        if (valid) {
          const size_t at = lbase + (size_t)cnt * kTraceBlock;
          ea->list_idx[at] = (int32_t)(kidx[k] >> kSlotBits);
          ea->list_w[at] = s_wl[(kidx[k] & 15u) * kBlock];
          ea->list_t[at] = __uint_as_float(tb);
          ++cnt;
        }
      }
      unsigned long long fix_l = __builtin_amdgcn_ballot_w64(in_img && cnt < kK && key[kK - 1] < kEmpty);
      while (__builtin_expect(fix_l != 0ull, 0)) {
        const int fl = __builtin_ctzll(fix_l);
        fix_l &= fix_l - 1ull;
        const float Xf = lane_f(X, fl), Yf = lane_f(Y, fl), dlf = lane_f(dlen, fl);
        float sf = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)key[kK - 1], fl));
        int left = kK - __builtin_amdgcn_readlane(cnt, fl);
        unsigned tn, gn;
        float wn;
        while (left > 0 && find_next(Xf, Yf, dlf, sf, tn, gn, wn)) {
          if (lane == fl) {
            const size_t at = lbase + (size_t)cnt * kTraceBlock;
            ea->list_idx[at] = (int32_t)gn;
            ea->list_w[at] = wn;
            ea->list_t[at] = __uint_as_float(tn);
            ++cnt;
          }
          sf = __uint_as_float(tn);
          --left;
        }
      }
      // the list's length instead of a terminator in every unused slot (round 5: 64 B of index plane per pixel were
      // written whatever the list held; now 4 B + 12 B per entry)
      ea->list_cnt[(size_t)tile_l * kTraceBlock + (size_t)(sub * kWave + lane_e)] = cnt;
      return;
    }
  }

  // ---- kStage: the tile's referenced Gaussians' SH coefficients into LDS, once per workgroup ----
  if constexpr (kStage) {
    constexpr unsigned kIdMask = 0x3FFFFFFu;
    // (a) every ray marks the survivor-list slots its entries name
#pragma unroll
    for (int k = 0; k < kK; ++k)
      if (key[k] < kEmpty && !(kidx[k] & kAbs)) s_ref[(kidx[k] >> kSlotBits) & kIdMask] = 1;
    __syncthreads();
    // (b) one wave numbers the marked slots (list order: a function of the tile alone) and notes their Gaussians
    if (wave == 0) {
      int n = 0;
      for (int base = 0; base < kListCap; base += kWave) {
        const bool r = s_ref[base + lane] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(r);
        const int at = n + below(m);
        s_ref[base + lane] = (r && at < kStageSlots) ? (unsigned char)at : (unsigned char)0xFF;
        if (r && at < kStageSlots) s_stg_g[at] = s_idx[base + lane];
        n += __popcll(m);
      }
      if (lane == 0) s_nref[0] = n < kStageSlots ? n : kStageSlots;
#if RTGS_COUNT
      if (lane == 0) { n_named = (unsigned)n; n_over = n > kStageSlots ? 1u : 0u; }
#endif
    }
    __syncthreads();
    // (c) the four waves fetch them: lane q < 45 brings coefficient q of one Gaussian per instruction, straight into LDS
    {
      float* const stage = reinterpret_cast<float*>(s_misc);
      const int n = __builtin_amdgcn_readfirstlane(s_nref[0]);
      for (int j = wave; j < n; j += kWaves) {
        const unsigned gl = (unsigned)__builtin_amdgcn_readfirstlane(s_stg_g[j]);
        if (lane_e < RTGS_SH_COEFFS * 3)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(sh + (size_t)gl * (RTGS_SH_COEFFS * 3) + lane_e),
              (__attribute__((address_space(3))) void*)(stage + j * kStageStride), 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (d) every entry learns its staging slot and becomes absolute for the compositing below:
    // payload = staging slot (0xFF = none) << 24 | Gaussian index << 4 | exponent slot (these forms have n <= 2^18)
    static_assert(kBigSceneSlabs * kSlab <= (1 << 18), "staged payloads keep the Gaussian index in 18 bits");
#pragma unroll
    for (int k = 0; k < kK; ++k) {
      if (key[k] < kEmpty && !(kidx[k] & kAbs)) {
        const unsigned id = (kidx[k] >> kSlotBits) & kIdMask;
        kidx[k] = ((unsigned)s_ref[id] << 24) | ((unsigned)s_idx[id] << kSlotBits) | (kidx[k] & kSlotMask);
      } else {
        kidx[k] |= 0xFF000000u;     // (absolute already, or an empty slot: no staging slot)
      }
    }
    __syncthreads();   // every wave's fetches have landed (each waited for its own above)
  }

  float T = 1.f;
#if !RTGS_STAMPS
  if ((ea->flags & RTGS_TRACE_RESUME) && in_img) T = ea->att[pix];
#endif
  float L0 = 0.f, L1 = 0.f, L2 = 0.f;
  float Ysh[RTGS_SH_COEFFS];
  if (kHasSh) {
    const float il = 1.0f / dlen;
    const float cx = dcx * il, cy = dcy * il, cz = -il;
    float rcf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
      rcf[k] = ea->pose ? (float)at_frame(ea->pose, kMulti ? (size_t)blockIdx.y * ea->frame_stride : 0)->rc[k] : ea->rc[k];
    float wx = rcf[0] * cx + rcf[1] * cy + rcf[2] * cz;
    float wy = rcf[3] * cx + rcf[4] * cy + rcf[5] * cz;
    float wz = rcf[6] * cx + rcf[7] * cy + rcf[8] * cz;
    const float in = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);  // gaussian.py:200
    sh_basis(wx * in, wy * in, wz * in, Ysh);
  }
  [[maybe_unused]] const unsigned idx_mask = kStage ? 0x3FFFFu : ~0u;
  int nhit = 0;
  float last_t = 0.f;
  unsigned prev_tbits = 0u;
  // colour/opacity gathers issued together; the SH variants carry 15 basis values per lane and keep
  // two in flight whatever the scene size (four spilled 56 VGPRs in the <= 4096-slab SH kernel)
  constexpr int kGroupN = kHasSh && TN::kGroup > TN::kGroupSh ? TN::kGroupSh : TN::kGroup;
  constexpr int kGroup = kGroupN < kK ? kGroupN : kK;
  bool done = false;
#pragma unroll
  for (int k0 = 0; k0 < kK; k0 += kGroup) {
    if (done) break;
    // slots beyond the first `depth` only matter to lanes that dropped an exact t1 tie on the way
    // (the reference's peel loop peels on until `depth` DISTINCT entry distances are composited)
    if (k0 >= ea->depth && !wave_any(nhit < ea->depth && key[k0] < kEmpty)) break;
    if (k0 > 0 && !wave_any(key[k0] < kEmpty)) break;   // every list of the wave ends before this group
    float4 cog[kGroup];
    [[maybe_unused]] float wl[kGroup];
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
      const int k = k0 + j;
      const unsigned tb = key[k];
      const unsigned gi = (kidx[k] >> kSlotBits) & idx_mask;
      cog[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tb < kEmpty) cog[j] = colop[gi];
      // the exponent: slot (kidx & 15) of this lane's column
      wl[j] = s_wl[(kidx[k] & 15u) * kBlock];
    }
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
      const int k = k0 + j;
      if (!done) {
        const unsigned tbits = key[k];
        const unsigned gi = (kidx[k] >> kSlotBits) & idx_mask;
        // an exact repeat of the previous entry distance is what the reference's strict
        // `t1 > ray.start` drops (scene.py:433 after ray_tracer.py:101)
        const bool valid = tbits < kEmpty && tbits != prev_tbits && nhit < ea->depth;
        // transmittance cut-off (extension, off by default): a ray below it stops compositing
        // -- as if it had missed, ray.start = +inf -- and a wave ballot ends the loop
        const bool live = valid && !(ea->t_cutoff > 0.f && T < ea->t_cutoff);
        // (the ballot asks whether any ray can still composite something at this slot OR LATER: a slot that
        // merely repeats the previous entry distance must not end the loop for the entries behind it)
        if (ea->t_cutoff > 0.f && !wave_any(tbits < kEmpty && nhit < ea->depth && !(T < ea->t_cutoff))) { done = true; continue; }
        prev_tbits = tbits;
        float4 co = cog[j];
        if (kHasSh) {
          // Per entry 45 coefficients, 180 B per lane.  (Records padded to 192 or 256 B and 16-B aligned: c3 + SH 0.310 ->
          // 0.333 / 0.381 ms.  SH per (wave, insertion event) through scalar loads: 0.289 -> 0.317, c5 1.30 -> 1.99.
          // Staged per wave and list position instead of per workgroup, commit b90b368: 0.291 -> 0.325 / c5 1.282 ->
          // 1.713 -- a serial chain per position.)  Forms with the staging (kStage) read them from LDS for the entries
          // that have a slot; the few that have none -- more than kStageSlots referenced Gaussians in the tile, or
          // a tile whose survivor list was refilled -- and all other forms gather from memory.  The small-scene forms
          // fold them in with explicit fma chains from either source, so that their staged and gathering (two-wave,
          // single-peel) instantiations agree bit for bit; the 1 M-Gaussian forms keep `+=`, which the compiler
          // contracts as it likes: with explicit fmaf they need 141 VGPRs instead of 125 (c5 1.303 -> 1.542 ms).
          if constexpr (kStage) {
            const unsigned ss = kidx[k] >> 24;
#if RTGS_COUNT
            n_lds += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(live && ss != 0xFFu));
            n_mem += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(live && ss == 0xFFu));
#endif
            if (live) {
              // four steps of 12 floats (4 coefficients x rgb) from the lane's source: all 45 in flight at once cost 34 VGPRs
              const float* const lp = reinterpret_cast<const float*>(s_misc) + ss * kStageStride;
              const float* const gp = sh + (size_t)gi * (RTGS_SH_COEFFS * 3);
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4) {
                float cf[12];
                if (ss != 0xFFu) {
                  const float4 v0 = reinterpret_cast<const float4*>(lp)[3 * c4], v1 = reinterpret_cast<const float4*>(lp)[3 * c4 + 1],
                               v2 = reinterpret_cast<const float4*>(lp)[3 * c4 + 2];
                  cf[0] = v0.x; cf[1] = v0.y; cf[2] = v0.z; cf[3] = v0.w; cf[4] = v1.x; cf[5] = v1.y; cf[6] = v1.z; cf[7] = v1.w;
                  cf[8] = v2.x; cf[9] = v2.y; cf[10] = v2.z; cf[11] = v2.w;
                } else {
#pragma unroll
                  for (int q = 0; q < 12; ++q) cf[q] = 12 * c4 + q < RTGS_SH_COEFFS * 3 ? gp[12 * c4 + q] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (4 * c4 + q < RTGS_SH_COEFFS) {
                    co.x = fmaf(Ysh[4 * c4 + q], cf[3 * q + 0], co.x);
                    co.y = fmaf(Ysh[4 * c4 + q], cf[3 * q + 1], co.y);
                    co.z = fmaf(Ysh[4 * c4 + q], cf[3 * q + 2], co.z);
                  }
                }
              }
            }
          } else if (live) {   // (gathering for every slot, valid or not, to batch the loads: 2.24 vs 2.08 ms on c5)
            const float* sp = sh + (size_t)gi * (RTGS_SH_COEFFS * 3);
            if constexpr (kBig) {
#pragma unroll
              for (int q = 0; q < RTGS_SH_COEFFS; ++q) {
                co.x += Ysh[q] * sp[3 * q + 0];
                co.y += Ysh[q] * sp[3 * q + 1];
                co.z += Ysh[q] * sp[3 * q + 2];
              }
            } else {
              sh_add(co, Ysh, sp);
            }
          }
        }
        float wk;
        wk = wl[j];
        const float alpha = co.w * exp_m3(wk);  // == exp(-d^T S^-1 d) at the chord midpoint
        if (live) {
          const float wa = T * alpha;
          L0 += wa * co.x; L1 += wa * co.y; L2 += wa * co.z;
          T *= 1.0f - alpha;
          last_t = __uint_as_float(tbits);
          ++nhit;
        }
      }
    }
  }

  // ---- tie fix-up (round 4).  The reference composites `depth` DISTINCT entry distances
  // (scene.py:433 `intersections.x > ray.start` after ray_tracer.py:101); this launch kept the kK
  // nearest ENTRIES, and an exactly tied twin took one of the slots.  A lane whose list is full but
  // which has composited fewer than `depth` distinct entries (a few dozen pixels of a 1080p frame)
  // peels on here, with the whole wave's help: every lane takes one Gaussian of a slab that covers the
  // pixel, evaluates it for THAT pixel with the candidate loop's own arithmetic (pair_eval), and the
  // nearest entry beyond the lane's last one is composited -- the reference's next peel, literally --
  // until `depth` are done or the ray misses.  Cold code: one ballot per wave when there is no tie.
  if constexpr (kK > 1) {
    unsigned long long fix_m = __builtin_amdgcn_ballot_w64(in_img && nhit < ea->depth && key[kK - 1] < kEmpty &&
                                                           !(ea->t_cutoff > 0.f && T < ea->t_cutoff));
    if (__builtin_expect(fix_m != 0ull, 0)) {
      while (fix_m) {
        const int fl = __builtin_ctzll(fix_m);
        fix_m &= fix_m - 1ull;
        const float Xf = lane_f(X, fl), Yf = lane_f(Y, fl), dlf = lane_f(dlen, fl);
        float sf = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)key[kK - 1], fl));
        int left = ea->depth - __builtin_amdgcn_readlane(nhit, fl);
        unsigned tmin1, gmin;
        float wk;
        while (left > 0 && find_next(Xf, Yf, dlf, sf, tmin1, gmin, wk)) {
          bool stop = false;
          if (lane == fl) {
            float4 co = colop[gmin];
            if (kHasSh) {
              sh_add(co, Ysh, sh + (size_t)gmin * (RTGS_SH_COEFFS * 3));
            }
            const float alpha = co.w * exp_m3(wk);
            const float wa = T * alpha;
            L0 += wa * co.x; L1 += wa * co.y; L2 += wa * co.z;
            T *= 1.0f - alpha;
            last_t = __uint_as_float(tmin1);
            ++nhit;
            stop = ea->t_cutoff > 0.f && T < ea->t_cutoff;   // below the cut-off: stopped for good
          }
          if (__builtin_amdgcn_ballot_w64(stop) != 0ull) break;
          sf = __uint_as_float(tmin1);
          --left;
        }
      }
    }
  }

#if RTGS_COUNT
  if (lane == 0) {
    const unsigned v[12] = {n_iter, n_pass, n_ins, n_hit_lanes, n_ins_lanes, n_exp, n_scan, 1u, n_lds, n_mem, n_named, n_over};
    for (int k = 0; k < 12; ++k)
      if (v[k]) atomicAdd(&g_count[k], (unsigned long long)v[k]);
  }
#endif
  if (ea->flags & RTGS_TRACE_COMPACT) {
    float* o = ea->sample + (kMulti ? (size_t)blockIdx.y * ea->out_stride : 0) +
               ((size_t)ltile * kTile * kTile + (size_t)(pi_e - mi0) * kTile + (pj_e - mj0)) * 3;
    if (ea->flags & RTGS_TRACE_ACCUMULATE) {
      const Rgb old = *reinterpret_cast<const Rgb*>(o);
      L0 += old.r; L1 += old.g; L2 += old.b;
    }
    *reinterpret_cast<Rgb*>(o) = Rgb{L0, L1, L2};   // one 12-B store per lane
  } else if (in_img) {
    float* o = ea->sample + (kMulti ? (size_t)blockIdx.y * ea->out_stride : 0) + pix * 3;
    if (ea->flags & RTGS_TRACE_ACCUMULATE) {
      const Rgb old = *reinterpret_cast<const Rgb*>(o);
      L0 += old.r; L1 += old.g; L2 += old.b;
    }
    *reinterpret_cast<Rgb*>(o) = Rgb{L0, L1, L2};
  }
#if RTGS_STAMPS
  st[4] = __builtin_amdgcn_s_memtime();
  if (tid == 0 && ea->att) {
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(ea->att) + (size_t)blockIdx.x * 14;
    sp[10] = t_exp; sp[11] = t_bar; sp[12] = t_merge; sp[13] = (unsigned long long)c_rounds;
    for (int k = 0; k < 5; ++k) sp[k] = st[k];
    sp[8] = rt0; sp[9] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    sp[5] = (unsigned long long)c_iter | ((unsigned long long)c_hit << 16) | ((unsigned long long)c_ins << 32) |
            ((unsigned long long)min(c_slabs, 65535) << 48);
    sp[6] = t_front; sp[7] = t_ins; (void)hwid; (void)xcc;
  }
#endif
  if (in_img) {
#if !RTGS_STAMPS
    if (ea->att) ea->att[pix] = T;
#endif
    // ray_tracer.py:99-104: start = t1 + EPS after a hit, +inf after a miss (fewer than `depth`
    // distinct entries: the tie fix-up above has peeled on until the ray really missed) or below the cut-off
    if (ea->start) ea->start[pix * ea->start_stride] = (nhit == ea->depth) ? last_t + 1e-8f : kInf;
  }
}

// ------------------------------------------------------------------------------------
// peel lists: the reference viewer's one-peel-per-call pattern from ONE traversal
// ------------------------------------------------------------------------------------
struct PeelArgs {
  int32_t width, height, tiles_x;
  int32_t k;            // which entry of every ray's list this call composites (0 = the nearest)
  uint32_t fresh;       // the sample starts here: attenuation 1 instead of attenuation_buf (clear_attenuation)
  float inv_fx, inv_fy;
  float rc[9];
  const FrameCam* pose;
  const int32_t* list_cnt;
  const int32_t* list_idx;
  const float* list_w;
  float* sample;
  float* att;
};

// RayTracer.sample_step (ray_tracer.py:79-104) for camera rays whose sorted entries k_trace has already
// listed: sample += attenuation * alpha * colour, attenuation *= 1 - alpha for entry k of every ray -- what
// the K = 1 resumed launch computes after re-traversing the tile.  Same thread <-> pixel map as k_trace, so
// a list slot is one coalesced 1 KB read per workgroup.
template <bool kHasSh>
__global__ __launch_bounds__(kTraceBlock) void k_peel(const PeelArgs a, const float4* __restrict__ colop,
                                                       const float* __restrict__ sh) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), sub = tid >> 6;
  const int tile = blockIdx.x;
  const int pi = (tile % a.tiles_x) * kTile + (sub % kWavesX) * kWaveTile + (lane >> 3);
  const int pj = (tile / a.tiles_x) * kTile + (sub / kWavesX) * kWaveTile + (lane & 7);
  if (pi >= a.width || pj >= a.height) return;
  const size_t at = ((size_t)tile * RTGS_MAX_DEPTH + a.k) * kTraceBlock + tid;
  const size_t pix = (size_t)pi * a.height + pj;
  if (a.k >= a.list_cnt[(size_t)tile * kTraceBlock + tid]) {   // a miss (ray_tracer.py:103-104): nothing to add
    if (a.fresh) a.att[pix] = 1.0f;
    return;
  }
  const int gi = a.list_idx[at];
  const float wk = a.list_w[at];
  float T = a.fresh ? 1.0f : a.att[pix];
  float4 co = colop[gi];
  if (kHasSh) {
    const float hw = 0.5f * (float)a.width, hh = 0.5f * (float)a.height;
    const float dcx = (((float)pi + 0.5f) - hw) * a.inv_fx, dcy = (((float)pj + 0.5f) - hh) * a.inv_fy;
    const float il = 1.0f / sqrtf(dcx * dcx + dcy * dcy + 1.0f);
    const float cx = dcx * il, cy = dcy * il, cz = -il;
    float rcf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) rcf[k] = a.pose ? (float)a.pose->rc[k] : a.rc[k];
    const float wx = rcf[0] * cx + rcf[1] * cy + rcf[2] * cz;
    const float wy = rcf[3] * cx + rcf[4] * cy + rcf[5] * cz;
    const float wz = rcf[6] * cx + rcf[7] * cy + rcf[8] * cz;
    const float in = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);  // gaussian.py:200
    float Ysh[RTGS_SH_COEFFS];
    sh_basis(wx * in, wy * in, wz * in, Ysh);
    sh_add(co, Ysh, sh + (size_t)gi * (RTGS_SH_COEFFS * 3));
  }
  const float alpha = co.w * exp_m3(wk);
  const float wa = T * alpha;
  float* o = a.sample + pix * 3;
  const Rgb old = *reinterpret_cast<const Rgb*>(o);
  *reinterpret_cast<Rgb*>(o) = Rgb{old.r + wa * co.x, old.g + wa * co.y, old.b + wa * co.z};
  a.att[pix] = T * (1.0f - alpha);
}

// cam_ray_field.start after `done` peels of the listed sample (ray_tracer.py:99-104): t1 of the last peel
// + 1e-8, +inf once a peel has missed, 0 before the first
__global__ __launch_bounds__(kTraceBlock) void k_peel_start(int32_t width, int32_t height, int32_t tiles_x, int32_t done,
                                                             const int32_t* __restrict__ list_cnt,
                                                             const float* __restrict__ list_t,
                                                             float* __restrict__ start, int32_t stride) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), sub = tid >> 6;
  const int tile = blockIdx.x;
  const int pi = (tile % tiles_x) * kTile + (sub % kWavesX) * kWaveTile + (lane >> 3);
  const int pj = (tile / tiles_x) * kTile + (sub / kWavesX) * kWaveTile + (lane & 7);
  if (pi >= width || pj >= height) return;
  float s_out = 0.f;
  if (done > 0) {
    const size_t at = ((size_t)tile * RTGS_MAX_DEPTH + (done - 1)) * kTraceBlock + tid;
    s_out = done <= list_cnt[(size_t)tile * kTraceBlock + tid] ? list_t[at] + 1e-8f : kInf;
  }
  start[((size_t)pi * height + pj) * stride] = s_out;
}

// rank_stride: tiles between consecutive ranks' blocks (tiles_per_rank for one frame per gather,
// batch x tiles_per_rank when every rank sends `batch` frames' shards in one block)
__global__ __launch_bounds__(256) void k_untile(const float* __restrict__ compact, int world,
                                                 int rank_stride, int width, int height,
                                                 int tiles_x, float* __restrict__ image) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= width * height) return;
  const int i = p / height, j = p % height;
  const int tile = (j / kTile) * tiles_x + i / kTile;
  const int rank = tile % world, local = tile / world;
  const float* s = compact + (((size_t)rank * rank_stride + local) * kTile * kTile +
                              (size_t)(i % kTile) * kTile + (j % kTile)) * 3;
  float* o = image + (size_t)p * 3;
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}

// Any partition of the frame (rtgs.parallel.Partition): tile_map[tile] = rank << 24 | compact slot in that rank's
// buffer; every rank has a buffer of its own (the destination's own shard is read where it was traced: no copy).
constexpr int kMaxRanks = 16;
struct RankBufs { const float* p[kMaxRanks]; };
__global__ __launch_bounds__(256) void k_untile_map(RankBufs bufs, const int32_t* __restrict__ tile_map, int width,
                                                     int height, int tiles_x, float* __restrict__ image) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= width * height) return;
  const int i = p / height, j = p % height;
  const int m = tile_map[(j / kTile) * tiles_x + i / kTile];
  // (a malformed caller-supplied map -- a rank beyond the buffers passed, or one without a buffer -- must not be followed)
  if ((unsigned)m >> 24 >= (unsigned)kMaxRanks || !bufs.p[m >> 24]) return;
  const float* s = bufs.p[m >> 24] + ((size_t)(m & 0xFFFFFF) * kTile * kTile + (size_t)(i % kTile) * kTile + (j % kTile)) * 3;
  float* o = image + (size_t)p * 3;
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}

// Frames batched into one launch take their camera from a pose block in device memory, one per frame region of the
// workspace (what RtgsCamera.device_pose is for one frame): this writes up to kMaxMerge of them from the kernel
// arguments -- no host buffer, no copy engine, stream-ordered like everything else here.
constexpr int kMaxMerge = 8;
struct PoseBatch { FrameCam c[kMaxMerge]; };
static_assert(sizeof(FrameCam) % sizeof(double) == 0 && sizeof(PoseBatch) <= 1024, "pose blocks travel as kernel arguments");
__global__ __launch_bounds__(64) void k_upload_poses(const PoseBatch batch, char* __restrict__ first_block, size_t frame_stride,
                                                     int32_t frames) {
  constexpr int kWords = (int)(sizeof(FrameCam) / sizeof(double));
  // (read through the kernarg segment: a dynamically indexed by-value struct would be copied to scratch first)
  using KDouble = const double __attribute__((address_space(4)));
  KDouble* src = (KDouble*)__builtin_amdgcn_kernarg_segment_ptr();
  const int t = threadIdx.x;
  if (t < kWords)
    for (int f = 0; f < frames; ++f)
      reinterpret_cast<double*>(first_block + (size_t)f * frame_stride)[t] = src[f * kWords + t];
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? RTGS_OK : RTGS_ELAUNCH; }

}  // namespace
}  // namespace rtgs

using namespace rtgs;

extern "C" {

// bins: [counts kBinSupertilesMax int][lists kBinSupertilesMax x bin_stride_of(nslabs) int], behind the depth keys,
// + the macro-tiles' own lists, 1 KB per tile of the frame (8 MB at 1080p, 33 MB at 4K; frames of more than
// kTileBinTilesMax tiles do without)
static size_t bins_bytes(size_t nslabs, size_t tiles) {
#if RTGS_BINS
  return (size_t)kBinSupertilesMax * sizeof(int32_t) * (1 + (size_t)bin_stride_of(nslabs)) +
         (RTGS_TILE_BINS && tiles <= (size_t)kTileBinTilesMax ? tiles * kTileBinCap * sizeof(int32_t) : 0);
#else
  (void)nslabs; (void)tiles;
  return 0;
#endif
}

// Bins pay once a tile's own scan is more than one step (k_bin is one more launch, ~3.5 us): c2's
// 157 slabs are faster without (0.0486 vs 0.0518 ms per frame), c3's 1 563 and c5's 15 625 with.
static bool use_bins(int nslabs) { return RTGS_BINS != 0 && nslabs >= 513; }

struct BinGeom { int shift, sx, supertiles; };
// smallest supertile of 2^shift x 2^shift macro-tiles (64 px and up) that covers the frame with <= kBinSupertilesMax bins
static BinGeom bin_geom(int width, int height) {
  const int tx = (width + kTile - 1) / kTile, ty = (height + kTile - 1) / kTile;
  BinGeom g{2, 0, 0};
  for (;; ++g.shift) {
    g.sx = (tx + (1 << g.shift) - 1) >> g.shift;
    const int sy = (ty + (1 << g.shift) - 1) >> g.shift;
    g.supertiles = g.sx * sy;
    if (g.supertiles <= kBinSupertilesMax) return g;
  }
}

static size_t workspace_bytes_tiles(int32_t n, size_t tiles) {
  if (n <= 0) return 0;
  const size_t nn = ((size_t)n + 63) & ~(size_t)63;
  const size_t nslabs = nn / kSlab, nsuper = (nslabs + kSuper - 1) / kSuper;
  return nn * sizeof(float4) + nn * (kRecFloats + kTailFloats + 1) * sizeof(float) + (nslabs + nsuper) * sizeof(float4) +
         ((nslabs + 3) & ~(size_t)3) * sizeof(float) + bins_bytes(nslabs, tiles);
}

// any frame size: provides for kTileBinTilesMax tiles (64 MiB of per-tile lists)
size_t rtgs_workspace_bytes(int32_t n) { return workspace_bytes_tiles(n, (size_t)kTileBinTilesMax); }

// what a width x height frame needs (ABI v6): the per-tile lists sized from the frame
size_t rtgs_workspace_bytes_frame(int32_t n, int32_t width, int32_t height) {
  if (width <= 0 || height <= 0) return 0;
  return workspace_bytes_tiles(n, (size_t)((width + kTile - 1) / kTile) * (size_t)((height + kTile - 1) / kTile));
}

// One frame's region of a workspace that holds several (rtgs_shard_frames batches consecutive frames into one launch
// when the stream's workspace has room): the frame's own workspace, then its pose block, rounded to 256 B.
static size_t frame_pose_offset(int32_t n, int32_t width, int32_t height) {
  return (rtgs_workspace_bytes_frame(n, width, height) + 15) & ~(size_t)15;
}
static size_t frame_region_bytes(int32_t n, int32_t width, int32_t height) {
  return (frame_pose_offset(n, width, height) + sizeof(FrameCam) + 255) & ~(size_t)255;
}
size_t rtgs_workspace_bytes_frames(int32_t n, int32_t width, int32_t height, int32_t frames) {
  if (n <= 0 || width <= 0 || height <= 0 || frames <= 0) return 0;
  return frames == 1 ? rtgs_workspace_bytes_frame(n, width, height) : (size_t)frames * frame_region_bytes(n, width, height);
}
int32_t rtgs_max_frames_per_launch(void) { return kMaxMerge; }

size_t rtgs_pose_bytes(void) { return sizeof(FrameCam); }

int rtgs_pose_pack(const RtgsCamera* cam, void* host_out) {
  if (!cam || !host_out || cam->width <= 0 || cam->height <= 0 || !(cam->fx > 0.f) || !(cam->fy > 0.f))
    return RTGS_EINVAL;
  const FrameCam fc = make_frame_cam(*cam);
  memcpy(host_out, &fc, sizeof(fc));
  return RTGS_OK;
}

int32_t rtgs_tile_size(void) { return kTile; }

int32_t rtgs_num_tiles(int32_t width, int32_t height) {
  if (width <= 0 || height <= 0) return 0;
  return ((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
}

int32_t rtgs_num_local_tiles(int32_t width, int32_t height, int32_t tile_first, int32_t tile_step) {
  const int32_t total = rtgs_num_tiles(width, height);
  if (tile_step <= 0 || tile_first < 0 || tile_first >= total) return 0;
  return (total - tile_first + tile_step - 1) / tile_step;
}

// Host only.  kind 0: most central tile first; kind 1: one image wedge per XCD (see rtgs_hip.h).
static int32_t launch_order_of(int32_t width, int32_t height, const std::vector<int32_t>& tiles, int32_t kind, int32_t* order) {
  const int32_t n = (int32_t)tiles.size();
  if (!order || n <= 0 || (kind != 0 && kind != 1)) return RTGS_EINVAL;
  const int tx = (width + kTile - 1) / kTile;
  std::vector<double> centr(n), angle(n);
  for (int32_t l = 0; l < n; ++l) {
    const int t = tiles[l];
    const double cx = ((t % tx) + 0.5) * kTile - 0.5 * width, cy = ((t / tx) + 0.5) * kTile - 0.5 * height;
    centr[l] = std::max(std::fabs(cx) / (0.5 * width), std::fabs(cy) / (0.5 * height));
    angle[l] = std::atan2(cy, cx);
  }
  std::vector<int32_t> by(n);
  for (int32_t l = 0; l < n; ++l) by[l] = l;
  constexpr int kXcds = 8;
  if (kind == 0 || n < 8 * kXcds) {
    std::stable_sort(by.begin(), by.end(), [&](int32_t p, int32_t q) { return centr[p] < centr[q]; });
    memcpy(order, by.data(), sizeof(int32_t) * (size_t)n);
    return n;
  }
  // equal-count wedges by angle rank, most central first inside a wedge, then dealt out round-robin:
  // position kXcds * i + x holds the i-th tile of wedge x
  std::stable_sort(by.begin(), by.end(), [&](int32_t p, int32_t q) { return angle[p] < angle[q]; });
  std::vector<std::vector<int32_t>> wedge(kXcds);
  for (int32_t r = 0; r < n; ++r) wedge[(int)(((int64_t)r * kXcds) / n)].push_back(by[r]);
  size_t rows = 0;
  for (auto& wv : wedge) {
    std::stable_sort(wv.begin(), wv.end(), [&](int32_t p, int32_t q) { return centr[p] < centr[q]; });
    rows = std::max(rows, wv.size());
  }
  int32_t pos = 0;
  for (size_t i = 0; i < rows; ++i)
    for (int x = 0; x < kXcds; ++x)
      if (i < wedge[x].size()) order[pos++] = wedge[x][i];
  return pos;
}

int32_t rtgs_launch_order(int32_t width, int32_t height, int32_t tile_first, int32_t tile_step, int32_t kind,
                          int32_t* order) {
  const int32_t n = rtgs_num_local_tiles(width, height, tile_first, tile_step);
  if (n <= 0) return RTGS_EINVAL;
  std::vector<int32_t> tiles(n);
  for (int32_t l = 0; l < n; ++l) tiles[l] = tile_first + l * tile_step;
  return launch_order_of(width, height, tiles, kind, order);
}

// the same for an explicit tile list (HOST array): RtgsTraceParams.tile_list's launch order
int32_t rtgs_launch_order_list(int32_t width, int32_t height, const int32_t* tiles, int32_t n_tiles, int32_t kind,
                               int32_t* order) {
  if (!tiles || n_tiles <= 0) return RTGS_EINVAL;
  const int32_t total = rtgs_num_tiles(width, height);
  std::vector<int32_t> tv(tiles, tiles + n_tiles);
  for (int32_t t : tv)
    if (t < 0 || t >= total) return RTGS_EINVAL;
  return launch_order_of(width, height, tv, kind, order);
}

// workspace: [bbox nn x float4][records nn x 16 float][tails nn x 8 float][distance bounds nn float][slab bounds][group bounds][slab depth keys]
static float* slab_key_of(float4* slab_bbox, size_t nn) {
  const size_t nslabs = nn / kSlab, nsuper = (nslabs + kSuper - 1) / kSuper;
  return reinterpret_cast<float*>(slab_bbox + nslabs + nsuper);
}

static int validate(const RtgsScene* scene, const RtgsCamera* cam, void* ws, size_t ws_bytes) {
  if (!scene || !cam || !ws) return RTGS_EINVAL;
  if (scene->n <= 0 || !scene->position || !scene->rotation || !scene->scale || !scene->color_opacity)
    return RTGS_EINVAL;
  if (scene->n > kMaxGaussians) return RTGS_EINVAL;   // 32-bit record offsets, index << 4 | slot payloads
  if (cam->width <= 0 || cam->height <= 0 || !(cam->fx > 0.f) || !(cam->fy > 0.f)) return RTGS_EINVAL;
  if (ws_bytes < rtgs_workspace_bytes_frame(scene->n, cam->width, cam->height)) return RTGS_EWORKSPACE;
  if (reinterpret_cast<uintptr_t>(ws) & 15u) return RTGS_EINVAL;
  return RTGS_OK;
}

}  // extern "C"

// the macro-tiles' own slab lists live behind the supertile bins; frames of more tiles than the workspace provides
// for (kTileBinTilesMax) have none
static int32_t* tile_bin_of(int32_t* bin_count, size_t nslabs, int width, int height) {
  if (!RTGS_TILE_BINS || rtgs_num_tiles(width, height) > kTileBinTilesMax) return nullptr;
  return bin_count + (size_t)kBinSupertilesMax * (1 + (size_t)bin_stride_of(nslabs));
}

// want_bins: build the supertile bins (rtgs_project always asks; rtgs_trace's own projection only when
// its shard is large enough for them to pay, see trace_impl)
// frames > 1: `frames` consecutive frames in ONE launch each of k_project / k_bin (grid.y = frames): frame b's region
// of the workspace lies b * frame_stride bytes behind the first and ends in its pose block (pose0 = frame 0's),
// which the kernels read instead of `cam`'s pose
static int project_impl(const RtgsScene* scene, const RtgsCamera* cam, void* workspace,
                        size_t workspace_bytes, rtgs_stream_t stream, bool want_bins, const float* region = nullptr,
                        int32_t frames = 1, size_t frame_stride = 0, const FrameCam* pose0 = nullptr,
                        int32_t own_first = 0, int32_t own_step = 1, bool own_region = false) {
  if (int rc = validate(scene, cam, workspace, workspace_bytes)) return rc;
  BinOwner own;
  own.first = own_first; own.step = own_step;
  own.rect_on = (own_region && region) ? 1 : 0;
  own.rect = region ? make_float4(region[0], region[1], region[2], region[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  const FrameCam* const pose = frames > 1 ? pose0 : static_cast<const FrameCam*>(cam->device_pose);
  const size_t fstride = frames > 1 ? frame_stride : 0;
  const size_t nn = ((size_t)scene->n + 63) & ~(size_t)63;
  float4* bbox = static_cast<float4*>(workspace);
  float* rec = reinterpret_cast<float*>(bbox + nn);
  float* tail = rec + nn * kRecFloats;
  float* tmin = tail + nn * kTailFloats;
  float4* slab_bbox = reinterpret_cast<float4*>(tmin + nn);
  const FrameCam fc = make_frame_cam(*cam);
  const int nslabs = (int)(nn / kSlab);
  const bool bins = use_bins(nslabs) && want_bins;
  int32_t* const bin_count = reinterpret_cast<int32_t*>(slab_key_of(slab_bbox, nn) + ((nslabs + 3) & ~3));
  const bool groups = nslabs > kBigSceneSlabs;   // the scenes k_trace / k_bin scan in two levels (Tune<true>)
  const float4 rgn = region ? make_float4(region[0], region[1], region[2], region[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  static_assert(kSuper * kSlab == 1024, "k_project<1024>: one block = one group of slabs");
  if (groups)
    hipLaunchKernelGGL(k_project<kSuper * kSlab>, dim3((scene->n + kSuper * kSlab - 1) / (kSuper * kSlab), frames), dim3(kSuper * kSlab), 0,
                       static_cast<hipStream_t>(stream), scene->n, scene->position, scene->rotation,
                       scene->scale, fc, pose, bbox, slab_bbox,
                       slab_bbox + nslabs, slab_key_of(slab_bbox, nn), rec, tail, tmin,
                       (use_bins(nslabs) && !bins) ? bin_count : nullptr, rgn, region ? 1 : 0, fstride);
  else
    hipLaunchKernelGGL(k_project<256>, dim3((scene->n + 255) / 256, frames), dim3(256), 0,
                       static_cast<hipStream_t>(stream), scene->n, scene->position, scene->rotation,
                       scene->scale, fc, pose, bbox, slab_bbox,
                       nullptr, slab_key_of(slab_bbox, nn), rec, tail, tmin,
                       (use_bins(nslabs) && !bins) ? bin_count : nullptr, rgn, region ? 1 : 0, fstride);
  if (bins) {
    const BinGeom g = bin_geom(cam->width, cam->height);
    float* keys = slab_key_of(slab_bbox, nn);
    const int tiles_x = (cam->width + kTile - 1) / kTile, tiles_y = (cam->height + kTile - 1) / kTile;
    int32_t* const tile_bin = tile_bin_of(bin_count, nslabs, cam->width, cam->height);
    if (groups)
      hipLaunchKernelGGL(k_bin<true>, dim3(g.supertiles, frames), dim3(256), 0, static_cast<hipStream_t>(stream), nslabs,
                         slab_bbox, slab_bbox + nslabs, keys, cam->width, cam->height, g.shift, g.sx,
                         bin_stride_of(nslabs), bin_count, bin_count + kBinSupertilesMax, tile_bin, tiles_x, tiles_y, fstride, own);
    else
      hipLaunchKernelGGL(k_bin<false>, dim3(g.supertiles, frames), dim3(256), 0, static_cast<hipStream_t>(stream), nslabs,
                         slab_bbox, slab_bbox + nslabs, keys, cam->width, cam->height, g.shift, g.sx,
                         bin_stride_of(nslabs), bin_count, bin_count + kBinSupertilesMax, tile_bin, tiles_x, tiles_y, fstride, own);
  }
  return check_launch();
}

extern "C" int rtgs_project(const RtgsScene* scene, const RtgsCamera* cam, void* workspace,
                            size_t workspace_bytes, rtgs_stream_t stream) {
  return project_impl(scene, cam, workspace, workspace_bytes, stream, true);
}

// Round 5: whether a projection builds the slab bins no longer depends on the SHARD (up to here a shard below a sixth
// of the frame scanned for itself: k_bin, one more launch in front of the trace, 7 us at c3 and 20 us at c5, did not
// pay for a lone small launch).  A tile's candidates arrive nearest slab first from its bin; a tile that scans for itself
// lists its survivors in BATCHES of up to kSlabStep scanned slabs when they do not fit one list, each batch sorted for
// itself -- another arrival order, and the arrival order decides which of two EXACTLY tied entries a ray composites
// (the reference's BVH order there is unspecified).  One shard with bins and one without could so differ in a pixel:
// found by the tile-parallel fuzzer (seed 37424: one pixel of a 100 x 16 frame, 40 000 Gaussians, 8.5e-2 between a
// merged batch -- bins -- and the same shard rendered alone -- none).  Now every launch of a scene large enough for bins
// uses them, restricted to the shard's own tiles where the projection is the shard's own (rtgs_shard_frame(s)): a
// tile's arrival order is a function of scene, camera, frame and tile alone again.  RTGS_TRACE_BINS_OFF remains for
// experiments.
extern "C" int rtgs_project_shard(const RtgsScene* scene, const RtgsCamera* cam, int32_t tile_first,
                                  int32_t tile_step, void* workspace, size_t workspace_bytes,
                                  rtgs_stream_t stream) {
  if (!cam || tile_step <= 0 || tile_first < 0) return RTGS_EINVAL;
  return project_impl(scene, cam, workspace, workspace_bytes, stream, true);
}

// peel lists: [count per pixel: tiles x 256 int32][index plane][exponent plane][t1 plane], the three planes
// [tile][RTGS_MAX_DEPTH][256 threads]
struct PeelLists { int32_t* cnt; int32_t* idx; float* w; float* t; };
static PeelLists peel_lists_of(void* lists, int32_t width, int32_t height) {
  const size_t tiles = (size_t)rtgs_num_tiles(width, height), slots = tiles * RTGS_MAX_DEPTH * kTraceBlock;
  PeelLists p;
  p.cnt = static_cast<int32_t*>(lists);
  p.idx = p.cnt + tiles * kTraceBlock;
  p.w = reinterpret_cast<float*>(p.idx + slots);
  p.t = p.w + slots;
  return p;
}

// tile_limit caps the number of (tile_first, tile_step) tiles the launch covers (rtgs_render's
// row ranges); rtgs_trace passes "all of them".
static int trace_impl(const RtgsScene* scene, const RtgsCamera* cam, const RtgsTraceParams* params,
                      float* sample_buf, float* attenuation_buf, float* ray_start,
                      int32_t ray_start_stride, void* workspace, size_t workspace_bytes,
                      int32_t projected, int32_t tile_limit, rtgs_stream_t stream, void* lists = nullptr,
                      int32_t frames = 1, size_t frame_stride = 0, size_t out_stride = 0, const FrameCam* pose0 = nullptr,
                      bool own_bins = false) {
  if (!params || (!sample_buf && !lists)) return RTGS_EINVAL;
  if (frames > 1 && (lists || attenuation_buf || ray_start || params->depth > RTGS_MAX_DEPTH)) return RTGS_EINVAL;
  if (int rc = validate(scene, cam, workspace, workspace_bytes)) return rc;
  if (params->depth < 1) return RTGS_EDEPTH;
  if (params->depth > RTGS_MAX_DEPTH && !lists) {
    // The reference's depth is unbounded (__main__.py:66-71: one peel per sample() call, as many
    // as asked for).  A launch keeps RTGS_MAX_DEPTH entries per ray; more peels are a CHAIN of
    // launches through the per-pixel state -- each resumes behind the last entry of the one before
    // (ray.start, attenuation) and accumulates into the same image, exactly the reference's loop
    // cut into runs of 16.  Needs the state buffers; one projection serves the whole chain.
    if (!attenuation_buf || !ray_start) return RTGS_EDEPTH;
    RtgsTraceParams p = *params;
    int32_t left = params->depth;
    for (int32_t link = 0; left > 0; ++link) {
      p.depth = left > RTGS_MAX_DEPTH ? RTGS_MAX_DEPTH : left;
      if (link > 0) p.flags |= RTGS_TRACE_RESUME | RTGS_TRACE_ACCUMULATE;
      if (int rc = trace_impl(scene, cam, &p, sample_buf, attenuation_buf, ray_start, ray_start_stride,
                              workspace, workspace_bytes, link > 0 ? 1 : projected, tile_limit, stream, nullptr, 1, 0, 0,
                              nullptr, own_bins))
        return rc;
      left -= p.depth;
    }
    return RTGS_OK;
  }
  if (!(params->t_cutoff >= 0.f)) return RTGS_EINVAL;
  // an explicit tile list is announced by its flag, not by a non-NULL pointer: a rank whose share is EMPTY has no
  // list to point at, and must render nothing -- not fall back to the arithmetic deal (that was a whole frame written
  // into a one-tile buffer, found by the two-rank rehearsal in round 4)
  if ((params->flags & RTGS_TRACE_REGION) &&
      !(params->region[0] <= params->region[2] && params->region[1] <= params->region[3])) return RTGS_EINVAL;   // (NaN too)
  const bool listed = (params->flags & RTGS_TRACE_TILE_LIST) != 0;
  if (listed ? (params->tile_count < 0 || (params->tile_count > 0 && !params->tile_list))
             : (params->tile_list != nullptr || params->tile_step <= 0 || params->tile_first < 0)) return RTGS_EINVAL;
  if ((params->flags & RTGS_TRACE_RESUME) && (!attenuation_buf || !ray_start)) return RTGS_EINVAL;
  if (ray_start && ray_start_stride <= 0) return RTGS_EINVAL;
  int32_t local = listed ? params->tile_count
                         : rtgs_num_local_tiles(cam->width, cam->height, params->tile_first, params->tile_step);
  if (local > tile_limit) local = tile_limit;
  if (local <= 0) return RTGS_OK;  // a shard with no tiles
  if (!projected) {
    // (bins whenever the scene is large enough for them, whatever the shard: see rtgs_project_shard)
    const bool want_bins = !(params->flags & RTGS_TRACE_BINS_OFF);
    // own_bins (rtgs_shard_frame(s): projection and trace of ONE shard, paired per frame): the bins of the shard's
    // own tiles only.  rtgs_trace keeps whole-frame bins -- a caller may trace other shards from this projection
    const bool listed_ = (params->flags & RTGS_TRACE_TILE_LIST) != 0;
    if (int rc = project_impl(scene, cam, workspace, workspace_bytes, stream, want_bins,
                              (params->flags & RTGS_TRACE_REGION) ? params->region : nullptr, frames, frame_stride, pose0,
                              (own_bins && !listed_) ? params->tile_first : 0, (own_bins && !listed_) ? params->tile_step : 1,
                              own_bins && listed_ && (params->flags & RTGS_TRACE_REGION)))
      return rc;
  }

  const size_t nn = ((size_t)scene->n + 63) & ~(size_t)63;
  const FrameCam fc = make_frame_cam(*cam);
  TraceArgs a;
  a.n = scene->n;
  a.width = cam->width;
  a.height = cam->height;
  a.tiles_x = (cam->width + kTile - 1) / kTile;
  a.tile_first = params->tile_first;
  a.tile_step = params->tile_step;
  a.tile_list = listed ? params->tile_list : nullptr;
  a.total_tiles = rtgs_num_tiles(cam->width, cam->height);
  a.depth = params->depth;
  a.t_cutoff = params->t_cutoff;
  a.flags = params->flags;
  a.inv_fx = (float)fc.inv_fx;
  a.inv_fy = (float)fc.inv_fy;
  for (int k = 0; k < 9; ++k) a.rc[k] = (float)fc.rc[k];
  a.pose = frames > 1 ? pose0 : static_cast<const FrameCam*>(cam->device_pose);
  a.frame_stride = frames > 1 ? frame_stride : 0;
  a.out_stride = frames > 1 ? out_stride : 0;
  a.bin_count = nullptr; a.bin_list = nullptr; a.bin_stride = 0; a.bin_shift = 0; a.bin_sx = 1; a.tile_bin = nullptr;
  if (use_bins((int)(nn / kSlab))) {
    const BinGeom g = bin_geom(cam->width, cam->height);
    const size_t nslabs_ = nn / kSlab, nsuper_ = (nslabs_ + kSuper - 1) / kSuper;
    const char* base = static_cast<const char*>(workspace) + nn * sizeof(float4) +
                       nn * (kRecFloats + kTailFloats + 1) * sizeof(float) + (nslabs_ + nsuper_) * sizeof(float4) +
                       ((nslabs_ + 3) & ~(size_t)3) * sizeof(float);
    a.bin_count = reinterpret_cast<const int32_t*>(base);
    a.bin_list = a.bin_count + kBinSupertilesMax;
    a.bin_stride = bin_stride_of(nslabs_);
    a.bin_shift = g.shift;
    a.bin_sx = g.sx;
    a.tile_bin = tile_bin_of(const_cast<int32_t*>(a.bin_count), nslabs_, cam->width, cam->height);
  }
  const float4* bbox = static_cast<const float4*>(workspace);
  const float4* rec = bbox + nn;
  const float4* tail = rec + nn * (kRecFloats / 4);
  const float* tmin = reinterpret_cast<const float*>(tail + nn * (kTailFloats / 4));
  const float4* slab_bbox = reinterpret_cast<const float4*>(tmin + nn);
  const float4* super_bbox = slab_bbox + nn / kSlab;
  const float4* colop = reinterpret_cast<const float4*>(scene->color_opacity);
  a.sample = sample_buf;
  a.att = attenuation_buf;
  a.start = ray_start;
  a.start_stride = ray_start ? ray_start_stride : 1;
  a.tile_order = params->tile_order;
  a.list_cnt = nullptr; a.list_idx = nullptr; a.list_w = nullptr; a.list_t = nullptr;
  if (lists) {
    const PeelLists pl = peel_lists_of(lists, cam->width, cam->height);
    a.list_cnt = pl.cnt; a.list_idx = pl.idx; a.list_w = pl.w; a.list_t = pl.t;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool big = (scene->n + kSlab - 1) / kSlab > kBigSceneSlabs;
  const float* sh = scene->sh;
#define RTGS_LAUNCH_K(HAS_SH, BIG)                                                  \
  do {                                                                              \
    if (lists) hipLaunchKernelGGL((k_trace<HAS_SH, BIG, true, false>), dim3(local), dim3(kTraceBlock), 0, s, a, \
                                  bbox, slab_bbox, super_bbox, rec, tail, tmin, colop, sh);                              \
    else if (frames > 1) hipLaunchKernelGGL((k_trace<HAS_SH, BIG, false, true>), dim3(local, frames), dim3(kTraceBlock), 0, s, a, \
                                            bbox, slab_bbox, super_bbox, rec, tail, tmin, colop, sh);                    \
    else hipLaunchKernelGGL((k_trace<HAS_SH, BIG, false, false>), dim3(local), dim3(kTraceBlock), 0, s, a, \
                            bbox, slab_bbox, super_bbox, rec, tail, tmin, colop, sh);                                    \
  } while (0)
  if (sh) { if (big) RTGS_LAUNCH_K(true, true); else RTGS_LAUNCH_K(true, false); }
  else    { if (big) RTGS_LAUNCH_K(false, true); else RTGS_LAUNCH_K(false, false); }
#undef RTGS_LAUNCH_K
  return check_launch();
}

extern "C" {

int rtgs_trace(const RtgsScene* scene, const RtgsCamera* cam, const RtgsTraceParams* params,
               float* sample_buf, float* attenuation_buf, float* ray_start,
               int32_t ray_start_stride, void* workspace, size_t workspace_bytes,
               int32_t projected, rtgs_stream_t stream) {
  return trace_impl(scene, cam, params, sample_buf, attenuation_buf, ray_start, ray_start_stride,
                    workspace, workspace_bytes, projected, INT32_MAX, stream);
}

// ---- peel lists (ABI v6): the reference viewer calls sample() `depth` times per sample (__main__.py:253-256,
// ray_tracer.py:39-54), one peel each; rtgs_trace_lists traverses ONCE and stores every ray's sorted distinct
// entries, rtgs_peel composites entry k of every ray, rtgs_peel_start writes the ray.start the peels so far imply.

size_t rtgs_peel_list_bytes(int32_t width, int32_t height) {
  return (size_t)rtgs_num_tiles(width, height) * kTraceBlock *
         (sizeof(int32_t) + RTGS_MAX_DEPTH * (sizeof(int32_t) + 2 * sizeof(float)));
}

int rtgs_trace_lists(const RtgsScene* scene, const RtgsCamera* cam, const int32_t* tile_order, void* lists,
                     size_t lists_bytes, void* workspace, size_t workspace_bytes, int32_t projected,
                     rtgs_stream_t stream) {
  if (!cam || !lists || (reinterpret_cast<uintptr_t>(lists) & 15u)) return RTGS_EINVAL;
  if (lists_bytes < rtgs_peel_list_bytes(cam->width, cam->height)) return RTGS_EWORKSPACE;
  RtgsTraceParams p;
  p.depth = RTGS_MAX_DEPTH;
  p.t_cutoff = 0.f;
  p.tile_first = 0;
  p.tile_step = 1;
  p.flags = 0;
  p.tile_order = tile_order;
  p.tile_list = nullptr;
  p.tile_count = 0;
  p.region[0] = p.region[1] = p.region[2] = p.region[3] = 0.f;
  return trace_impl(scene, cam, &p, nullptr, nullptr, nullptr, 1, workspace, workspace_bytes, projected, INT32_MAX,
                    stream, lists);
}

int rtgs_peel(const RtgsScene* scene, const RtgsCamera* cam, const void* lists, size_t lists_bytes, int32_t k,
              uint32_t flags, float* sample_buf, float* attenuation_buf, rtgs_stream_t stream) {
  if (!scene || !cam || !lists || !sample_buf || !attenuation_buf || !scene->color_opacity) return RTGS_EINVAL;
  if (cam->width <= 0 || cam->height <= 0 || !(cam->fx > 0.f) || !(cam->fy > 0.f)) return RTGS_EINVAL;
  if (k < 0 || k >= RTGS_MAX_DEPTH) return RTGS_EDEPTH;
  if (lists_bytes < rtgs_peel_list_bytes(cam->width, cam->height)) return RTGS_EWORKSPACE;
  const FrameCam fc = make_frame_cam(*cam);
  const int32_t tiles = rtgs_num_tiles(cam->width, cam->height);
  const PeelLists pl = peel_lists_of(const_cast<void*>(lists), cam->width, cam->height);
  PeelArgs a;
  a.width = cam->width; a.height = cam->height; a.tiles_x = (cam->width + kTile - 1) / kTile;
  a.k = k; a.fresh = (flags & RTGS_PEEL_FRESH) ? 1u : 0u;
  a.inv_fx = (float)fc.inv_fx; a.inv_fy = (float)fc.inv_fy;
  for (int q = 0; q < 9; ++q) a.rc[q] = (float)fc.rc[q];
  a.pose = static_cast<const FrameCam*>(cam->device_pose);
  a.list_cnt = pl.cnt; a.list_idx = pl.idx; a.list_w = pl.w;
  a.sample = sample_buf; a.att = attenuation_buf;
  const float4* colop = reinterpret_cast<const float4*>(scene->color_opacity);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (scene->sh) hipLaunchKernelGGL(k_peel<true>, dim3(tiles), dim3(kTraceBlock), 0, s, a, colop, scene->sh);
  else hipLaunchKernelGGL(k_peel<false>, dim3(tiles), dim3(kTraceBlock), 0, s, a, colop, scene->sh);
  return check_launch();
}

int rtgs_peel_start(const RtgsCamera* cam, const void* lists, size_t lists_bytes, int32_t peels_done,
                    float* ray_start, int32_t ray_start_stride, rtgs_stream_t stream) {
  if (!cam || !lists || !ray_start || ray_start_stride <= 0 || cam->width <= 0 || cam->height <= 0) return RTGS_EINVAL;
  if (peels_done < 0 || peels_done > RTGS_MAX_DEPTH) return RTGS_EDEPTH;
  if (lists_bytes < rtgs_peel_list_bytes(cam->width, cam->height)) return RTGS_EWORKSPACE;
  const int32_t tiles = rtgs_num_tiles(cam->width, cam->height);
  const PeelLists pl = peel_lists_of(const_cast<void*>(lists), cam->width, cam->height);
  hipLaunchKernelGGL(k_peel_start, dim3(tiles), dim3(kTraceBlock), 0, static_cast<hipStream_t>(stream), cam->width,
                     cam->height, (cam->width + kTile - 1) / kTile, peels_done, pl.cnt, pl.t, ray_start, ray_start_stride);
  return check_launch();
}

int rtgs_render(const RtgsScene* scene, const RtgsCamera* cam, int32_t depth, float t_cutoff,
                int32_t tile_y0, int32_t tile_y1, float* sample_out, float* transmittance_out,
                void* workspace, size_t workspace_bytes, rtgs_stream_t stream) {
  if (!cam || cam->width <= 0 || cam->height <= 0) return RTGS_EINVAL;
  const int32_t tiles_x = (cam->width + kTile - 1) / kTile, tiles_y = (cam->height + kTile - 1) / kTile;
  if (tile_y0 < 0 || tile_y1 > tiles_y || tile_y0 > tile_y1) return RTGS_EINVAL;
  if (tile_y0 == tile_y1) return RTGS_OK;
  RtgsTraceParams p;
  p.depth = depth;
  p.t_cutoff = t_cutoff;
  p.tile_first = tile_y0 * tiles_x;
  p.tile_step = 1;
  p.flags = 0;
  p.tile_order = nullptr;
  p.tile_list = nullptr;
  p.tile_count = 0;
  p.region[0] = p.region[1] = p.region[2] = p.region[3] = 0.f;
  return trace_impl(scene, cam, &p, sample_out, transmittance_out, nullptr, 1, workspace, workspace_bytes,
                    0, (tile_y1 - tile_y0) * tiles_x, stream);
}

// ---- the tile-parallel frame loop's stream plumbing, in ONE call per side -------------------------
// (rtgs.parallel spent 40 us of host time per frame on five torch / ctypes calls around two launches)

int rtgs_event_create(void** event) {
  if (!event) return RTGS_EINVAL;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return RTGS_EDEVICE;
  *event = e;
  return RTGS_OK;
}

int rtgs_event_destroy(void* event) {
  if (!event) return RTGS_EINVAL;
  return hipEventDestroy(static_cast<hipEvent_t>(event)) == hipSuccess ? RTGS_OK : RTGS_EDEVICE;
}

int rtgs_event_query(void* event) {
  if (!event) return RTGS_EINVAL;
  const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return RTGS_EDEVICE;
}

int rtgs_shard_frame(const RtgsScene* scene, const RtgsCamera* cam, const RtgsTraceParams* params,
                     float* out, float* attenuation_buf, float* ray_start, void* workspace, size_t workspace_bytes,
                     rtgs_stream_t render_stream, void* reuse_event, void* traced_event,
                     rtgs_stream_t consumer_stream, int32_t consumer_waits) {
  if (!traced_event) return RTGS_EINVAL;
  hipStream_t rs = static_cast<hipStream_t>(render_stream);
  // the buffer's previous reader: usually long done, and then the render stream is spared the barrier packet
  if (reuse_event && hipEventQuery(static_cast<hipEvent_t>(reuse_event)) != hipSuccess) {
    (void)hipGetLastError();
    if (hipStreamWaitEvent(rs, static_cast<hipEvent_t>(reuse_event), 0) != hipSuccess) return RTGS_EDEVICE;
  }
  if (int rc = trace_impl(scene, cam, params, out, attenuation_buf, ray_start, 1, workspace, workspace_bytes, 0,
                          INT32_MAX, render_stream, nullptr, 1, 0, 0, nullptr, true))
    return rc;
  if (hipEventRecord(static_cast<hipEvent_t>(traced_event), rs) != hipSuccess) return RTGS_EDEVICE;
  if (consumer_waits &&
      hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), static_cast<hipEvent_t>(traced_event), 0) != hipSuccess)
    return RTGS_EDEVICE;
  return RTGS_OK;
}

// B frames of one shard, render side, in ONE call (ABI v7; VERDICT r4 #1b): the tile-parallel loop is host-bound at 8
// ranks (a 1/8 shard of a 1080p frame traces in ~20 us of throughput while a Python frame loop costs ~30 us), so the
// caller hands over a HOST array of camera poses and the library queues, for frame k = 0 .. n_frames-1,
// rtgs_project + rtgs_trace of the shard on render_streams[(first_stream + k) % n_streams] with that stream's
// workspace (and per-pixel state for depth > RTGS_MAX_DEPTH) into outs[k].  Before its first frame every stream used
// waits for reuse_event (the previous reader of the buffers outs[] point into; skipped when complete or NULL); after
// its last one it records done_events[stream] and -- consumer_waits != 0 -- consumer_stream is made to wait for it.
// Frames on one stream are ordered, so n_streams events order the whole batch.
int rtgs_shard_frames(const RtgsScene* scene, const RtgsCamera* cams, int32_t n_frames, const RtgsTraceParams* params,
                      float* const* outs, float* const* attenuation_bufs, float* const* ray_starts,
                      void* const* workspaces, size_t workspace_bytes, const rtgs_stream_t* render_streams,
                      int32_t n_streams, int32_t first_stream, void* reuse_event, void* const* done_events,
                      rtgs_stream_t consumer_stream, int32_t consumer_waits) {
  if (!cams || !outs || !workspaces || !render_streams || !done_events || n_frames < 0 || n_streams <= 0 ||
      first_stream < 0)
    return RTGS_EINVAL;
  if (n_frames == 0) return RTGS_OK;
  const int32_t used = n_frames < n_streams ? n_frames : n_streams;
  for (int32_t u = 0; u < used; ++u)
    if (!done_events[(first_stream + u) % n_streams]) return RTGS_EINVAL;
  const bool reuse_pending = reuse_event && hipEventQuery(static_cast<hipEvent_t>(reuse_event)) != hipSuccess;
  if (reuse_pending) {
    (void)hipGetLastError();
    for (int32_t u = 0; u < used; ++u)
      if (hipStreamWaitEvent(static_cast<hipStream_t>(render_streams[(first_stream + u) % n_streams]),
                             static_cast<hipEvent_t>(reuse_event), 0) != hipSuccess)
        return RTGS_EDEVICE;
  }
  // an EMPTY share (an explicit tile list of no tiles; more ranks than the frame has tiles) renders nothing: its frames
  // only take part in the event chain
  const bool empty_share = params && ((params->flags & RTGS_TRACE_TILE_LIST)
                                          ? params->tile_count == 0
                                          : (cams[0].width > 0 && cams[0].height > 0 && params->tile_step > 0 && params->tile_first >= 0 &&
                                             rtgs_num_local_tiles(cams[0].width, cams[0].height, params->tile_first, params->tile_step) == 0));
  // Frames that can share a launch: consecutive frames of one stream whose outputs are equally spaced, when the
  // stream's workspace has room for their regions (rtgs_workspace_bytes_frames) -- ONE k_project, k_bin and k_trace
  // with grid.y = frames instead of one latency-bound chain of small launches per frame: a 1/8 shard of a 1080p frame
  // is one resident round of workgroups, eight of them in one grid fill the chip like a whole frame does.
  int32_t merge = 1;
  size_t region = 0, pose_off = 0;
  if (!empty_share && n_frames > 1 && scene && params && !attenuation_bufs && !ray_starts && params->depth <= RTGS_MAX_DEPTH &&
      cams[0].width > 0 && cams[0].height > 0) {
    region = frame_region_bytes(scene->n, cams[0].width, cams[0].height);
    pose_off = frame_pose_offset(scene->n, cams[0].width, cams[0].height);
    const size_t room = region ? workspace_bytes / region : 0;
    merge = (int32_t)(room < (size_t)kMaxMerge ? room : (size_t)kMaxMerge);
    if (merge < 1) merge = 1;
  }
  int32_t group = 0;
  for (int32_t k = 0; k < n_frames && !empty_share;) {
    // group `group` of this call runs on stream (first_stream + group) % n_streams
    const int32_t si = (first_stream + (merge > 1 ? group : k)) % n_streams;
    if (!outs[k]) return RTGS_EINVAL;
    int32_t g = 1;
    ptrdiff_t step = 0;
    if (merge > 1) {
      while (k + g < n_frames && g < merge) {
        if (!outs[k + g] || cams[k + g].device_pose || cams[k].device_pose || cams[k + g].width != cams[k].width ||
            cams[k + g].height != cams[k].height || cams[k + g].fx != cams[k].fx || cams[k + g].fy != cams[k].fy) break;
        const ptrdiff_t d = outs[k + g] - outs[k + g - 1];
        if (g == 1) step = d;
        if (d != step || d <= 0) break;
        ++g;
      }
    }
    if (g > 1) {
      PoseBatch batch;
      for (int32_t b = 0; b < g; ++b) batch.c[b] = make_frame_cam(cams[k + b]);
      for (int32_t b = g; b < kMaxMerge; ++b) batch.c[b] = batch.c[0];
      char* const ws = static_cast<char*>(workspaces[si]);
      if (!ws) return RTGS_EINVAL;
      hipLaunchKernelGGL(k_upload_poses, dim3(1), dim3(64), 0, static_cast<hipStream_t>(render_streams[si]), batch,
                         ws + pose_off, region, g);
      if (int rc = trace_impl(scene, &cams[k], params, outs[k], nullptr, nullptr, 1, ws, region, 0, INT32_MAX,
                              render_streams[si], nullptr, g, region, (size_t)step,
                              reinterpret_cast<const FrameCam*>(ws + pose_off), true))
        return rc;
    } else {
      if (int rc = trace_impl(scene, &cams[k], params, outs[k], attenuation_bufs ? attenuation_bufs[si] : nullptr,
                              ray_starts ? ray_starts[si] : nullptr, 1, workspaces[si], workspace_bytes, 0, INT32_MAX,
                              render_streams[si], nullptr, 1, 0, 0, nullptr, true))
        return rc;
    }
    k += g;
    ++group;
  }
  // (which streams carried work: with merged groups the first min(groups, n_streams) from first_stream on, without
  // merging the first min(n_frames, n_streams): `used` below covers both -- recording an event on an idle stream is harmless)
  for (int32_t u = 0; u < used; ++u) {
    const int32_t si = (first_stream + u) % n_streams;
    hipEvent_t ev = static_cast<hipEvent_t>(done_events[si]);
    if (hipEventRecord(ev, static_cast<hipStream_t>(render_streams[si])) != hipSuccess) return RTGS_EDEVICE;
    if (consumer_waits && hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), ev, 0) != hipSuccess)
      return RTGS_EDEVICE;
  }
  return RTGS_OK;
}

int rtgs_frames_land(const float* gathered, int32_t world, int32_t tiles_per_rank, int32_t batch, int32_t n_land,
                     int32_t width, int32_t height, float* const* images, rtgs_stream_t stream, void* sent_event) {
  if (world <= 0 || tiles_per_rank <= 0 || batch <= 0 || n_land < 0 || n_land > batch || width <= 0 || height <= 0)
    return RTGS_EINVAL;
  if ((int64_t)world * tiles_per_rank < rtgs_num_tiles(width, height)) return RTGS_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the gather queued before this call has read the senders' buffers: they may be traced into again
  if (sent_event && hipEventRecord(static_cast<hipEvent_t>(sent_event), s) != hipSuccess) return RTGS_EDEVICE;
  if (n_land == 0) return RTGS_OK;
  if (!gathered || !images) return RTGS_EINVAL;
  const int npx = width * height, block = 256;
  for (int32_t b = 0; b < n_land; ++b) {
    if (!images[b]) return RTGS_EINVAL;
    hipLaunchKernelGGL(k_untile, dim3((npx + block - 1) / block), dim3(block), 0, s,
                       gathered + (size_t)b * tiles_per_rank * kTile * kTile * 3, world, batch * tiles_per_rank,
                       width, height, (width + kTile - 1) / kTile, images[b]);
  }
  return check_launch();
}

int rtgs_frames_land_map(const float* const* rank_bufs, const int32_t* rank_tiles, int32_t world, const int32_t* tile_map,
                         int32_t batch, int32_t n_land, int32_t width, int32_t height, float* const* images,
                         rtgs_stream_t stream, void* sent_event) {
  if (world <= 0 || world > kMaxRanks || batch <= 0 || n_land < 0 || n_land > batch || width <= 0 || height <= 0)
    return RTGS_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // sent_event = "the buffers this step read may be traced into again": the exchange for the senders, and for the
  // destination -- whose OWN shard is scattered from where it was traced -- the scatter too: recorded last
  if (n_land == 0) {
    if (sent_event && hipEventRecord(static_cast<hipEvent_t>(sent_event), s) != hipSuccess) return RTGS_EDEVICE;
    return RTGS_OK;
  }
  if (!rank_bufs || !rank_tiles || !tile_map || !images) return RTGS_EINVAL;
  int64_t sum = 0;
  for (int32_t r = 0; r < world; ++r) {
    if (rank_tiles[r] < 0 || (rank_tiles[r] > 0 && !rank_bufs[r])) return RTGS_EINVAL;
    sum += rank_tiles[r];
  }
  if (sum != rtgs_num_tiles(width, height)) return RTGS_EINVAL;     // a partition: every tile exactly once
  const int npx = width * height, block = 256;
  for (int32_t b = 0; b < n_land; ++b) {
    if (!images[b]) return RTGS_EINVAL;
    RankBufs bufs;
    for (int r = 0; r < kMaxRanks; ++r)   // frame b of rank r's batch block: [frame][slot][T][T][3]
      bufs.p[r] = r < world && rank_bufs[r] ? rank_bufs[r] + (size_t)b * rank_tiles[r] * kTile * kTile * 3 : nullptr;
    hipLaunchKernelGGL(k_untile_map, dim3((npx + block - 1) / block), dim3(block), 0, s, bufs, tile_map, width, height,
                       (width + kTile - 1) / kTile, images[b]);
  }
  if (sent_event && hipEventRecord(static_cast<hipEvent_t>(sent_event), s) != hipSuccess) return RTGS_EDEVICE;
  return check_launch();
}

int rtgs_untile_map(const float* const* rank_bufs, const int32_t* rank_tiles, int32_t world, const int32_t* tile_map,
                    int32_t width, int32_t height, float* image, rtgs_stream_t stream) {
  float* const images[1] = {image};
  if (!image) return RTGS_EINVAL;
  return rtgs_frames_land_map(rank_bufs, rank_tiles, world, tile_map, 1, 1, width, height, images, stream, nullptr);
}

#if RTGS_COUNT
// counting build only (not part of include/rtgs_hip.h): copy the 8 counters to the host, optionally zero them
int rtgs_debug_counters(unsigned long long* host_out, int32_t reset) {
  if (hipDeviceSynchronize() != hipSuccess) return RTGS_EDEVICE;
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_count), sizeof(g_count)) != hipSuccess) return RTGS_EDEVICE;
  if (reset) {
    const unsigned long long z[sizeof(g_count) / sizeof(g_count[0])] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_count), z, sizeof(z)) != hipSuccess) return RTGS_EDEVICE;
  }
  return RTGS_OK;
}
#endif

int rtgs_untile(const float* compact, int32_t world, int32_t tiles_per_rank, int32_t width,
                int32_t height, float* image, rtgs_stream_t stream) {
  if (!compact || !image || world <= 0 || tiles_per_rank <= 0 || width <= 0 || height <= 0) return RTGS_EINVAL;
  const int32_t total = rtgs_num_tiles(width, height);
  if ((int64_t)world * tiles_per_rank < total) return RTGS_EINVAL;
  const int block = 256;
  const int npx = width * height;
  hipLaunchKernelGGL(k_untile, dim3((npx + block - 1) / block), dim3(block), 0,
                     static_cast<hipStream_t>(stream), compact, world, tiles_per_rank, width, height,
                     (width + kTile - 1) / kTile, image);
  return check_launch();
}

}  // extern "C"
