// crt_device.hpp — shared declarations between the C-ABI host layer (crt_api.cpp) and the
// gfx950 kernels (crt_kernels.hip).  MI355X only; compiled with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
// The no-contraction flag is part of the numerical contract: every fp32 operation of the
// reference is evaluated separately (mul, then add), never fused.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crt {

// Base triangle record, camera independent (scene pack, createScene.h:175-190):
//   v0, v1, v2 (reversed winding already applied), stored normal.   12 floats = 48 B.
constexpr int kBaseFloats = 12;

// Derived record for a pinhole camera at origin o: 16 floats = 64 B = one s_load_dwordx16.
//   [0..2]   n   stored normal                    (cull test, shapes/triangle.h:61)
//   [3..5]   e2  v2 - v0                          (triangle.h:70)
//   [6..8]   e1  v1 - v0                          (triangle.h:69)
//   [9..11]  s   o - v0                           (triangle.h:81; same for every ray)
//   [12..14] q   cross(s, e1)                     (triangle.h:87)
//   [15]     tq  dot(e2, q)                       (triangle.h:92)
// Everything is computed with the reference's own operation order, so the values are the
// ones Triangle::collision_detection recomputes for every ray.
constexpr int kRecFloats = 16;

// Classification record (tiled variant), 16 floats = 4 float4 planes (SoA, plane-major:
// cls[plane * n_pad + i]) so that a wavefront with lane = triangle reads 1 KiB per load:
//   plane 0: (A, B, C, M) of dn  = d.n                 (cull test, triangle.h:61)
//   plane 1: (A, B, C, M) of a   = d.(e2 x e1)         (triangle.h:71-72)
//   plane 2: (A, B, C, M) of su  = d.(e2 x s)          (triangle.h:82)
//   plane 3: (A, B, C, M) of dq  = d.q                 (triangle.h:88)
// F(u,v) = A + u*B + v*C is the exact value of the form on the un-normalised camera
// direction d0(u,v) = (llc - o) + u*horizontal + v*vertical; M bounds |G|_1-type magnitudes
// used for the rounding slack.
constexpr int kClsPlanes = 5;

// EPSILON of both Triangle ctors the scene pack can reach (shapes/triangle.h:9,22).
constexpr float kTriEpsilon = 0.000001f;

struct CameraPOD {
    float llc[3], horizontal[3], vertical[3], origin[3];
};

// The rows a launch renders ("local rows" 0..row_count) are strips of `strip_rows` image rows:
// local row ly is image row  row_begin + (ly / strip_rows) * strip_rows * strip_step + ly % strip_rows.
// strip_step == 1 is a contiguous band; strip_step == W deals strips round-robin to W ranks.
struct RenderParams {
    CameraPOD cam;
    float albedo[3];
    int nx, ny, samples;
    int row_begin, row_count;
    int strip_rows, strip_step;
    int n_tris;
    long long rng_count;         // number of states = row_count * nx (SoA stride); element = ly * nx + x
    float rnx, rny;              // 1.0f / float(nx), 1.0f / float(ny) (host, IEEE)
    float pos_du, pos_dv;        // tiled variant: position slack of the camera in (u,v) (host, camera_slacks)
    float K_slack;               // tiled variant: slack factor of the form values, whole screen (host, camera_slacks)
    float pix_hu, pix_hv;        // half extents of a 1-pixel cell in (u, v) incl. position slack (pixel_half_extents; host)
    float cam_hlen, cam_vlen;    // |horizontal|, |vertical| as tile_cone forms them (host)
    float cam_mag, cam_thr;      // sum of |component| over llc, horizontal, vertical, origin, and 2^-20 of it (host; the sky loop's range
                                 // guard: computed in the kernel they were loop invariants that hipcc hoisted to the entry and spilled)
};

__host__ __device__ inline int global_row(int row_begin, int strip_rows, int strip_step, int ly) {
    return row_begin + (ly / strip_rows) * strip_rows * strip_step + ly % strip_rows;
}

// XORWOW jump table: J_k = T^(2^(67+k)), k = 0..63; row-vector convention,
// 160 rows x 5 words per matrix.
constexpr int kXwWords = 5;
constexpr int kXwBits = 160;
constexpr int kXwJumps = 64;

// Launchers (defined in crt_kernels.hip).
// States of `count` pixels: element i is pixel x = i % nx of local row ly = i / nx, whose
// global pixel index is first_pixel + global_row(0, strip_rows, strip_step, ly) * nx + x.
hipError_t launch_rng_init(uint32_t* d_state, long long first_pixel, long long count, int nx, int strip_rows,
                           int strip_step, uint64_t seed, const uint32_t* d_jump, hipStream_t stream);
hipError_t launch_tri_prepass(const float* d_base, int n_tris, const float origin[3],
                              float* d_rec, hipStream_t stream);
hipError_t launch_update_pose(const float* d_points, const int* d_idx, int n_tris, float* d_base, hipStream_t stream);
hipError_t launch_cls_prepass(const float* d_rec, int n_tris, int n_pad, const RenderParams& p,
                              float4* d_cls, hipStream_t stream);
// variant 0: brute force (every ray x every triangle); 1: tiled (conservative region/tile
// classification + exact test on survivors); 2: tiled with tiny list capacities (tests the
// overflow fallbacks); 3: tiled with the scalar-operand walk also for light tiles (A/B of
// the LDS-staged records).  d_cls may be null for variant 0.
// Scratch of the tiled variant (device pointers; owned by the host layer, grow-only).
// What one wavefront of the binning walk found in one tile: the survivors among its 64 triangles.
struct BinRecord {
    uint32_t tile, base;             // tile id; first triangle of the wavefront
    unsigned long long mask, sure;   // lanes (triangles) that may be hit in the tile; those whose cull test cannot reject there
};

struct TiledWork {
    BinRecord* log;                  // the walk's records (bin_tri_kernel -> bin_fill_kernel)
    uint32_t log_cap;                // records
    uint32_t* pool;                  // survivor indices of all tiles, bump-allocated
    unsigned long long* counters64;  // [0] pool cursor (entries)
    uint32_t pool_cap;               // entries
    uint2* tile_desc;                // per tile {pool offset, count | kNoList}
    uint32_t* queue;                 // kClasses weight classes x tiles_total tile ids
    uint32_t* counters;              // kCtrWords words: queue lengths and consumer cursors (layout below)
    int tiles_x, tiles_y, tiles_total;
    uint32_t* tile_count;            // per tile: survivors (count pass), fill cursor (fill pass)
};
// Weight classes of a tile (by survivor count; heaviest first = longest-processing-time-first
// order for the persistent shading blocks): class c < kHeavyClasses holds tiles with more than
// kClassMin[c] survivors (class 0 also: no list), the last class the light tiles.
constexpr int kHeavyClasses = 9;
constexpr int kLightClasses = 6;      // light tiles by survivors: >48, >32, >16, >8, >0, none
constexpr int kClasses = kHeavyClasses + kLightClasses;
constexpr int kCtrQueueLen = 2;      // counters[0..1] alias counters64[0]; [2 + c]: length of class c's queue
// Consumer cursors of the shading kernel.  Returning atomics on one address serialise in L2
// (~60 ns each, measured), so every hot cursor has its own 128-byte line and the light queue
// is cut into kLightChunks chunks with one cursor each.
constexpr int kLogSegments = 32;                   // TiledWork::log is cut into this many segments of log_cap / kLogSegments records, each with
                                                   // its own cursor on its own 128-byte line (kCtrLogCursor + s * kCtrLine): a binning wavefront
                                                   // appends to segment (its block index) % kLogSegments.  One cursor for all (rounds 1-3) was one
                                                   // L2 line taking a returning atomic from every wavefront, 20 ns each, one after the other:
                                                   // 15 625 wavefronts = 0.32 ms of the C5 frame, 33 us of the bench frame.
constexpr int kCtrLogOverflow = 25;                // != 0: the log was full, every tile walks the whole list
constexpr int kCtrLine = 32;                       // words per 128-byte line
constexpr int kCtrHeavyCursor = 2 * kCtrLine;      // next item of the concatenated heavy classes
constexpr int kLightChunks = 64;
constexpr int kCtrLightCursor = 3 * kCtrLine;      // + k * kCtrLine: next position inside chunk k
constexpr int kCtrLogCursor = (3 + kLightChunks) * kCtrLine;   // + s * kCtrLine: records appended to segment s of TiledWork::log
constexpr int kCtrWords = (3 + kLightChunks + kLogSegments) * kCtrLine;

// Occupancy of the persistent shading kernel, queried once per launch context and kernel.
struct LaunchCache {
    const void* fn = nullptr;
    int resident = 0;
    size_t lds = 0;   // dynamic LDS the query was made with (render_generic_kernel)
};
// crt_material (include/crt.h) as the kernels read it, and the per-triangle material view of a launch
struct MaterialPOD {
    int kind;          // 0 Lambertian, 1 Metal, 2 Dielectric, 3 DiffuseLight
    float albedo[3];
    float param;       // Metal: fuzz (clamped to 1 like the ctor); Dielectric: ref_idx
};
struct MatView {
    const int* material_of = nullptr;        // per triangle (list order) or per slot (BVH world); null = all 0
    const MaterialPOD* materials = nullptr;  // null: the scene's single Lambertian (RenderParams::albedo)
    int normal_shader = 0;                   // != 0: shade_normal (render.h:90-103): a hit returns rec.normal, nothing is drawn
};
// (a HOST struct of device pointers: launch_render hands it to the kernel by value)
struct BvhTiles {
    const float4* nodes;     // flat tree (crt_bvh_kernels.hpp)
    int n_nodes;
    const int* partner;      // per slot s: r(s) = s with the two slots of a 2-triangle leaf swapped.  The tree keeps, among equal t,
                             // the LAST leaf and inside a leaf the FIRST triangle (bvh.h:172-179, hitable_list.h:84): the larger r wins.
    const int* slot_leaf;    // per slot: leaf ordinal (bits 0..23) | leaf depth << 24
    const int* anc;          // per leaf ordinal: node indices of the path root..leaf, anc_stride ints each
    int anc_stride;
    float rot[9];            // rotate(., 0) matrix
};
hipError_t launch_render(const RenderParams& p, const float* d_rec, const float4* d_cls, int n_pad,
                         const float* d_shade_n, uint32_t* d_state, float* d_rgb, int variant,
                         const TiledWork* work, int n_cus, bool fast_rcp, unsigned long long* d_stats,
                         hipEvent_t ev_mid, LaunchCache* cache, bool prezeroed, hipStream_t stream,
                         const BvhTiles* d_bvh_tiles = nullptr, bool bvh_force_general = false, const MatView* materials = nullptr);
// The per-frame prepass of the tiled variant in ONE launch: exact records (tri_prepass) + classification
// records (cls_prepass) + zeroing of this frame's binning scratch (w.tile_count, w.counters);
// launch_render is then called with prezeroed = true.
hipError_t launch_frame_prepass(const float* d_base, int n_tris, int n_pad, const RenderParams& p, float* d_rec, float4* d_cls,
                                const TiledWork& w, hipStream_t stream);
// Multi-GPU reassembly on rank 0 (dist: strips of `strip_rows` rows dealt round-robin to `world`
// ranks): parts[r * part_stride ...] holds rank r's strips one after the other; every strip is
// copied to its rows of the full nx x ny float3 frame.
hipError_t launch_assemble_strips(const float* d_parts, long long part_stride, int nx, int ny, int strip_rows,
                                  int world, float* d_full, hipStream_t stream);
hipError_t launch_audit(const RenderParams& p, const float* d_rec, const float4* d_cls, int n_pad,
                        const uint32_t* d_state, const TiledWork* work, unsigned long long* d_counters,
                        hipStream_t stream);
hipError_t launch_check_reciprocal(int mode, uint32_t exp_lo, uint32_t exp_hi, unsigned long long* d_bad,
                                   hipStream_t stream);
hipError_t launch_vec3_ops(int op, long long n, const float* d_a, const float* d_b, const float* d_s, float* d_out, hipStream_t stream);
constexpr int kStatsWords = 48;
constexpr int kStatsTimeline = 47;   // != 0: shade_tiles_kernel records only its timeline (below), at full speed
constexpr int kStatsWaveSlots = 8192;                 // timeline: per wave (block * waves + wave) four words, three 100 MHz time stamps:
constexpr int kStatsWaveLog = kStatsWords;            // start, end of the heavy phase, end; then entries << 32 | tile of the block's last shared tile
constexpr int kStatsSampleBlocks = 4, kStatsSampleSlots = 256;   // timeline: blocks 0..3 log (draws done, slices done) per sample
constexpr int kStatsSampleLog = kStatsWaveLog + 4 * kStatsWaveSlots;
constexpr int kStatsLightLog = kStatsSampleLog + 2 * kStatsSampleBlocks * kStatsSampleSlots;   // timeline: per wave, its last light tile with geometry:
constexpr int kStatsTotalWords = kStatsLightLog + 3 * kStatsWaveSlots;                          // start stamp; tile | entries << 32 | such tiles taken << 48

// BVH world (crt_bvh_kernels.hpp).  d_nodes: 2 float4 per node, preorder with skip links
// (layout there); d_rec / d_shade_n are in leaf ("slot") order.  fast: bounded scene (host check).
// force_mode: 0 = normal, 1 = general loop for every wavefront, 2 = general loop iterating the
// direction chain from its fourth element (test hooks: all must give the same pixels).
hipError_t launch_render_bvh(const RenderParams& p, const float4* d_nodes, int n_nodes, const float rot[9], const float* d_rec,
                             const float* d_shade_n, uint32_t* d_state, float* d_rgb, bool fast, int force_mode,
                             hipStream_t stream);
// pair records of the tree (crt_bvh_kernels.hpp: bvh_walk_pairs): boxes refreshed from the flat nodes
hipError_t launch_bvh_pairs(const float4* d_nodes, float4* d_pairs, const int* d_pair_flat, int n_pairs, hipStream_t stream);
constexpr float kConeScale = 1.02f / 254.0f;   // threshold byte of a pair record's normal cone -> threshold (crt_bvh.cpp: cone_of)
constexpr int kGenericQueueWords = 16;   // render_generic_kernel's queue cursor (launch context)
constexpr int kGenericChunkLog = 2;      // its pixel queue is dealt out in chunks of 4 positions
constexpr int kPairArea = 32;        // ints of LDS per lane for bvh_walk_pairs: the stack from the bottom, queued triangles from the top
constexpr int kPairStackDepth = 24;   // entries of the per-lane stack of bvh_walk_pairs (one per tree level); deeper trees keep the skip-link walk
// The generic render kernel (crt_shade_kernels.hpp): shader 0 LambertShade, 1 shade (recursive, depth
// <= max_depth < kMaxShadeDepth), 2 shade_normal; world = list (d_nodes null) or BVH; per-triangle
// materials (d_materials: array of {int kind; float albedo[3]; float param}).  d_base / d_shade_n /
// d_material_of are in list order for the list world, in slot order for the BVH world.
constexpr int kMaxShadeDepth = 31;
hipError_t launch_render_generic(const RenderParams& p, const float* d_base, const float* d_shade_n, const int* d_material_of,
                                 const void* d_materials, int n_tris, const float4* d_nodes, int n_nodes, const float rot[9],
                                 uint32_t* d_state, float* d_rgb, bool bvh, bool fast, int shader, int max_depth, int force_mode,
                                 hipStream_t stream, const float4* d_pairs, const int* d_partner, int walk_mode, unsigned* d_pixel_queue, int n_cus,
                                 unsigned long long* d_stats = nullptr,    // d_stats: tuning builds only (BvhView::walk_stats)
                                 LaunchCache* cache = nullptr);            // the occupancy query, kept per launch context
// walk_mode (BVH world with pair records): 2 the wavefront shares the walk (bvh_walk_coop), 1 one stack per lane (bvh_walk_pairs),
// 0 skip links, 3 the shared walk with a short list (test hook).  d_pixel_queue: kGenericQueueWords words of the launch context.
// What the tiled kernels need of the tree to serve the BVH world's PRIMARY rays:
// a triangle is hit through the tree iff it passes Triangle::collision_detection AND the boxes of its
// leaf and of all the leaf's ancestors pass AABB::hit — the tile lists (a superset of the first
// condition) give the candidates, the chain of boxes is checked for the winner of every pixel.

// One level of BVHNode::UpdateBVH: the nodes level_nodes[0..n_level) (all of one depth) recompute
// their boxes from their triangles (leaves) or children (the deeper levels are already done).
hipError_t launch_bvh_refit_level(float4* d_nodes, const int* d_level_nodes, int n_level, const float* d_base_slots,
                                  const float rot[9], hipStream_t stream);

// Sets the calling thread's crt_last_error() text and returns `code` (host layer, crt_api.cpp).
int set_error(int code, const char* fmt, ...);

}  // namespace crt
