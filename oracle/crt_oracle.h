/*
 * crt_oracle.h — CPU oracle for the list-render hot path of chocorone/CudaRaytracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / the timed CPU baseline.  The product path (cudaraytracer_amd/csrc) never
 * links, includes or calls it.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden images or known-answer vectors
 * (SURVEY.md §4), it is CUDA (needs nvcc, cuda_runtime.h, curand_kernel.h — all absent
 * from this image) and therefore cannot be built here without writing stand-ins for
 * those headers/libraries, which the build rules forbid.  This file is a plain-C,
 * line-by-line restatement of the reference's algorithm (each function cites the
 * reference file:line it follows, paths relative to /root/reference/CudaTest/src) and
 * of cuRAND's published XORWOW algorithm (CUDA Toolkit 12.2 curand_kernel.h, not in the
 * reference tree).  It defines "the reference on the host CPU" for this repository.
 * (Two pieces of the reference ARE plain C/C++ and build from their own files, into
 * oracle/_ref/ — `make ref`: the PNG encoder WritePng calls, ref_stb_png.c, and the
 * timing table's writer, ref_csv.cpp.  They pin the product's output files byte for byte,
 * tests/test_png.py; they say nothing about the render path.  Later rounds added, the same
 * way: core/vec3.h (ref_vec3.cpp), the hit-path headers compiled for gfx950 (ref_hit.hip),
 * rocThrust's thrust::sort (ref_thrust_sort.hip) and rocRAND's XORWOW engine for the
 * recurrence and the 2^67-step skips (ref_rocrand_xorwow.cpp).  DESIGN.md §2 lists what
 * each pins and what stays unpinned.)
 *
 * Floating point: every operation is IEEE-754 binary32 evaluated in source order with
 * no contraction (build with -ffp-contract=off), except the reference's double-typed
 * sites, which are kept literally.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* core/vec3.h:7-50 */
typedef struct { float e[3]; } o_vec3;

/* cuRAND XORWOW state (curand_kernel.h `curandStateXORWOW`): only d and v[5] matter on
 * this path (the Box-Muller fields are never touched). */
typedef struct { unsigned int d; unsigned int v[5]; } o_curand_state;

/* core/camera.h:16-53 */
typedef struct {
    o_vec3 lower_left_corner, horizontal, vertical, origin, x, y, z;
    float lens_radius;
} o_camera;

/* hitable/transform.h:6-36 */
typedef struct { o_vec3 position, rotation, scale; } o_transform;

/* shapes/triangle.h:5-54 (+ the per-object Transform of hitable/hitable.h:52) */
typedef struct {
    o_transform transform;
    float EPSILON;
    o_vec3 vertices[3];
    o_vec3 normal;
    int backCulling;
} o_triangle;

/* material/material.h:42-161 with ConstantTexture (material/texture.h:12-22) */
enum { O_LAMBERTIAN = 0, O_METAL = 1, O_DIELECTRIC = 2, O_DIFFUSE_LIGHT = 3 };
typedef struct {
    int kind;
    o_vec3 albedo;   /* Lambertian: albedo->value(); Metal: albedo; DiffuseLight: emit->value() */
    float param;     /* Metal: fuzz as given (the ctor clamps it to 1, material.h:79-81); Dielectric: ref_idx */
} o_material;

/* hitable/hitable_list.h:6-73.  The scene pack gives every triangle one shared
 * Lambertian(ConstantTexture(albedo)) (createScene.h:180): materials[0]; o_world_set_materials
 * replaces the table and the per-triangle Material* (as indices). */
typedef struct {
    o_transform transform;
    o_triangle *list;
    int list_size;
    o_vec3 albedo;
    o_material *materials;
    int n_materials;
    int *material_of;   /* per triangle, NULL = all 0 */
} o_world;

/* core/aabb.h:14-55 */
typedef struct { o_vec3 _min, _max; } o_aabb;

/* hitable/bvh.h:48-73 BVHNode, flattened: nodes are stored in preorder (a node, its left
 * subtree, its right subtree), children and leaves are indices.  A leaf's HitableList
 * (childList, 1 or 2 triangles) is the span [first, first+count) of the sorted pointer array. */
typedef struct {
    o_transform transform;        /* BVHNode's own Transform (reset to identity, bvh.h:81) */
    o_transform list_transform;   /* leaf: childList's Transform (reset, bvh.h:97,103) */
    o_aabb box;
    int isLeaf, left, right;      /* children: node indices, -1 in a leaf */
    int first, count;             /* span of the pointer array this node was built from */
    int depth, axis, skip;        /* depth of the node (root 0); axis drawn at bvh.h:84; preorder index after the subtree */
} o_bvh_node;
typedef struct o_bvh o_bvh;

/* ---- RNG: curand_init / curand / curand_uniform (core/deviceManage.h:127, render.h:116) */
void o_curand_init(unsigned long long seed, unsigned long long subsequence,
                   unsigned long long offset, o_curand_state *state);
unsigned int o_curand(o_curand_state *state);
float o_curand_uniform(o_curand_state *state);
/* test hook: subsequence * 2^67 steps, then offset steps, on a raw state (tests/test_ref_xorwow.py) */
void o_xorwow_skip_raw(o_curand_state *state, unsigned long long subsequence, unsigned long long offset);
/* core/deviceManage.h:120-136: state[pi] = curand_init(seed, pi, 0) for pi in [0,nx*ny) */
void o_set_curand_state(o_curand_state *state, int nx, int ny, unsigned long long seed);
/* same, for an arbitrary pixel range [first, first+count) written to state[0..count) */
void o_set_curand_state_range(o_curand_state *state, long long first, long long count,
                              unsigned long long seed);

/* ---- camera: core/camera.h:18-38, createScene.h:142-154 */
void o_camera_init(o_camera *cam, const float lookfrom[3], const float lookat[3],
                   const float vup[3], float vfov, float aspect, float aperture,
                   float focus_dist);

/* ---- scene pack: createScene.h:175-190 (add_mesh_withNormal).
 * idx3f are the reference's float-triplet indices; index_base is subtracted after the
 * int() conversion (0 for the FBX path, 1 for OBJ indices stored undecremented,
 * Loader/obj_loader.h:62-64).  normals == NULL selects the 5-argument Triangle ctor rule
 * (shapes/triangle.h:8-19). Returns NULL on bad indices. */
o_world *o_world_from_mesh(const float *points, int n_points, const float *idx3f,
                           int n_tris, const float *normals, int index_base,
                           const float albedo[3]);
/* Triangle::material per triangle (triangle.h:54): table + one index per triangle (NULL = all 0).
 * -1: bad index or kind. */
int o_world_set_materials(o_world *w, const o_material *table, int n_materials, const int *triangle_material);
/* createScene.h:99-110 update_pose: new vertex positions, normals kept.  -1: index out of range. */
int o_world_update_pose(o_world *w, const float *new_points, int n_points, const float *idx3f, int index_base);
void o_world_free(o_world *w);

/* ---- OBJ: Loader/obj_loader.h:20-133. Two-call protocol: pass NULL buffers to count. */
int o_parse_obj(const char *path, float *points, float *idx3f, int *n_points, int *n_tris);

/* ---- the render kernel, run serially pixel by pixel in index order
 * (core/render.h:105-129).  colorBuffer: nx*ny*3 floats, row 0 = image bottom.
 * state: nx*ny entries.  rows [row_begin,row_end) and columns [col_begin,col_end) are
 * rendered (the full image is row 0..ny, col 0..nx); untouched pixels are left alone. */
void o_render_window(float *colorBuffer, const o_world *world, const o_camera *camera,
                     o_curand_state *state, int nx, int ny, int samples, int max_depth,
                     int frameIndex, int row_begin, int row_end, int col_begin,
                     int col_end);
void o_render(float *colorBuffer, const o_world *world, const o_camera *camera,
              o_curand_state *state, int nx, int ny, int samples, int max_depth,
              int frameIndex);

/* ---- BVH: create_BVH (createScene.h:42-44) = new BVHNode(list->list, list_size, 0, 1, state)
 * (hitable/bvh.h:76-125).  `state` is the pointer the reference passes — the BASE of the
 * curandState array (kernel.cu:36, createScene.h:212-215), i.e. stream 0, which every node
 * advances by one draw for its split axis.  The world's triangles are referenced, not copied:
 * o_world_update_pose + o_bvh_update = updateFBXObj + Update_BVH (render.h:201-207).
 * NULL when the list is empty (the reference recurses without end). */
o_bvh *o_bvh_build(o_world *world, o_curand_state *state);
void o_bvh_update(o_bvh *bvh);                       /* BVHNode::UpdateBVH, bvh.h:135-158 */
void o_bvh_free(o_bvh *bvh);
int o_bvh_node_count(const o_bvh *bvh);
const o_bvh_node *o_bvh_nodes(const o_bvh *bvh);
const int *o_bvh_order(const o_bvh *bvh);            /* order[slot] = triangle index after the in-place sorts */
int o_bvh_hit(const o_bvh *bvh, const float origin[3], const float direction[3],
              float *t_out, int *index_out, float normal_out[3]);
/* the render kernel with world = the BVH (renderBVHAnimation, core/render.h:191-237) */
void o_render_window_bvh(float *colorBuffer, const o_bvh *bvh, const o_camera *camera,
                         o_curand_state *state, int nx, int ny, int samples, int max_depth,
                         int frameIndex, int row_begin, int row_end, int col_begin,
                         int col_end);

/* The render kernel with an explicit choice of the two things the reference selects by
 * (un)commenting source lines: the world (bvh == NULL: the list, render.h:179; else the BVH,
 * render.h:222) and the shader called at core/render.h:119-121 — O_SHADER_LAMBERT = LambertShade
 * (:70-87, the live one), O_SHADER_SHADE = the recursive path tracer `shade` (:48-67),
 * O_SHADER_NORMAL = shade_normal (:90-103). */
enum { O_SHADER_LAMBERT = 0, O_SHADER_SHADE = 1, O_SHADER_NORMAL = 2 };
void o_render_window_ex(float *colorBuffer, const o_world *world, const o_bvh *bvh, int shader,
                        const o_camera *camera, o_curand_state *state, int nx, int ny, int samples,
                        int max_depth, int frameIndex, int row_begin, int row_end, int col_begin,
                        int col_end);

/* Diagnostics for tests: closest hit of ONE ray through the world list
 * (core/render.h:75 -> hitable.h:29-38 -> hitable_list.h:76-91). Returns hit flag. */
int o_world_hit(const o_world *world, const float origin[3], const float direction[3],
                float *t_out, int *index_out, float normal_out[3]);

/* Batch forms of the two above plus rec.p, and the box arithmetic of core/aabb.h / hitable/transform.h, for
 * tests/test_ref_hit.py (which holds them to the reference's own headers running on the GPU, oracle/ref_hit.hip).
 * o_world_hit_many: `levels` extra one-element HitableList wrappers around the world list (each one more
 * Hitable::hit level, hitable/hitable.h:29-38). */
int o_world_hit_many(const o_world *world, int levels, long long n, const float *origin, const float *direction,
                     int *hit, float *t_out, int *index_out, float *normal_out, float *p_out);
int o_bvh_hit_many(const o_bvh *bvh, long long n, const float *origin, const float *direction,
                   int *hit, float *t_out, int *index_out, float *normal_out, float *p_out);
/* AABB::hit (core/aabb.h:30-43) */
int o_aabb_hit_many(long long n, const float *bmin, const float *bmax, const float *origin, const float *direction,
                    const float *t_min, const float *t_max, int *out);
/* op 0 surrounding_box(a, b) (aabb.h:58-66); op 1 Transform(position b.min, rotation b.max, scale 1).TransformAABB(a)
 * (transform.h:22-31) */
int o_aabb_op_many(int op, long long n, const float *amin, const float *amax, const float *bmin, const float *bmax,
                   float *omin, float *omax);
/* which 0: GetBV of every triangle; 1: GetBV of the world list (hitable_list.h:94-118).  Returns bounding_box's flag. */
int o_world_bounding_boxes(const o_world *world, int which, float *omin, float *omax);

/* ---- WritePng's pixel conversion (core/render.h:135-146): RGBA8, vertical flip,
 * char(255.99*c). rgba: nx*ny*4 bytes. */
void o_to_rgba8(const float *colorBuffer, int nx, int ny, unsigned char *rgba);

/* ---- thrust::sort(l, l + n, BoxCompare(axis)) as BVHNode's constructor calls it inside one device thread
 * (hitable/bvh.h:84-92): Thrust's sequential stable merge sort (insertion sort per 32 elements, pairwise merges) with
 * the comparator `val1 - val2 < 0.0 ? false : true` (bvh.h:36-41) on key[order[i]]; order is permuted in place.
 * Pinned against rocThrust's real thrust::sort (oracle/ref_thrust_sort.hip, tests/test_ref_thrust.py). */
int o_thrust_sort(int n, const float *key, int *order);

/* ---- one core/vec3.h operation at a time (op codes of oracle/ref_vec3.cpp); pinned against the reference's own
 * vec3.h by tests/test_vec3_ref.py.  Returns 1 for an operation the oracle does not restate. */
int o_vec3_op(int op, long long n, const float *a, const float *b, const float *s, float *out);

#ifdef __cplusplus
}
#endif
#endif
