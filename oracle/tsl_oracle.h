/*
 * tsl_oracle.h -- CPU ORACLE for the TaichiSLAM dense-mapping hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (taichislam_amd/) never does.
 *
 * It is a plain-C restatement of the reference algorithm (xuhao1/TaichiSLAM,
 * taichi_slam/mapping/ *.py); every function cites the reference file:line it follows.
 *
 * PARITY PINNED TO THE REFERENCE'S SOURCE (not to Taichi): the reference ships no golden vectors or
 * asserting tests and its execution engine (taichi, un-pinned, requirements.txt:4) is not installable
 * here.  Its own source is: dense_tsdf.py, mapping_common.py, taichi_octomap.py and
 * marching_cube_mesher.py are imported unmodified and RUN on a sequential stand-in for the Taichi subset
 * they use (tools/ti_seq, tools/gen_ref_golden.py), and ORA_FAITHFUL reproduces what they produce --
 * maps after depth / point-cloud / colour integration, the weight clamp, submap fusion, Octomap leaves,
 * mesh vertices and normals -- bit for bit (tests/golden/ref_*.npz, tests/test_ref_golden.py).  What stays
 * an assumption, shared by the stand-in and this file (A1-A10), is the behaviour of Taichi's own
 * back end: f16 arithmetic through f32, ti.round half away from zero, the cast in front of an atomic add,
 * no FMA contraction, and WHICH serialisation of a parallel struct-for a run corresponds to.
 *
 * Two update modes (DESIGN.md "Defined semantics"):
 *   ORA_FAITHFUL  sequential replay of the reference: f16 fields, per-update f16 rounding,
 *                 raster order for the depth accumulate, Taichi struct-for order for rays.
 *   ORA_BATCHED   identical per-ray / per-step arithmetic, but the per-frame contributions
 *                 (w*sd, w) are summed in exact 2^-24 fixed point per voxel and applied once
 *                 per touched voxel.  Order-free => what the GPU implements bit-for-bit.
 *   ORA_IDEAL     FAITHFUL's update sequence with TSDF / W kept in float64 between updates (f16 only on export): the
 *                 yardstick for "how far is each of the two from what the reference's formula means".
 */
#ifndef TSL_ORACLE_H
#define TSL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_FAITHFUL = 0, ORA_BATCHED = 1, ORA_IDEAL = 2 };

typedef struct {
    double map_size_xy, map_size_z;   /* DenseTSDF(map_scale=[xy,z])          dense_tsdf.py:13 */
    double voxel_scale;
    int    num_voxel_per_blk_axis;
    double max_ray_length, min_ray_length;
    int    internal_voxels;
    int    max_submap_num;
    int    is_global_map;
    int    texture_enabled;
    double disp_ceiling, disp_floor;
    int    recast_step;
    int    color_same_proj;
} ora_tsdf_cfg;

typedef struct {
    int64_t p_used;      /* pixels / points visited                                   */
    int64_t p_valid;     /* passed the range gate and inside the sensor-centred grid  */
    int64_t p_oob;       /* passed the gate but outside the sensor-centred grid       */
    int64_t v_pcl;       /* sensor-grid voxels with count>0 (= rays)                  */
    int64_t v_skipped;   /* degenerate rays skipped (len==0, z^2 not finite/zero)     */
    int64_t steps;       /* ray-steps applied  (S)                                    */
    int64_t steps_oob;   /* ray-steps skipped: voxel outside the map volume           */
    int64_t unique;      /* distinct map voxels touched this frame (U)                */
    int64_t bricks;      /* distinct 16^3 bricks touched this frame                   */
} ora_frame_stats;

typedef struct ora_tsdf ora_tsdf;

ora_tsdf* ora_tsdf_create(const ora_tsdf_cfg* cfg);
void      ora_tsdf_destroy(ora_tsdf* m);
void      ora_tsdf_get_dims(const ora_tsdf* m, int* N, int* Nz, int* pcl_lo, int* pcl_hi);
void      ora_tsdf_set_intrinsics(ora_tsdf* m, const double Kdep[9], const double Kcol[9]);
void      ora_tsdf_set_base_pose(ora_tsdf* m, const double R[9], const double T[3]);
void      ora_tsdf_set_base_pose_submap(ora_tsdf* m, int sid, const double R[9], const double T[3]);
int       ora_tsdf_get_active_submap(const ora_tsdf* m);
void      ora_tsdf_set_active_submap(ora_tsdf* m, int sid);
void      ora_tsdf_reset(ora_tsdf* m);

int ora_tsdf_integrate_depth(ora_tsdf* m, int mode, const double R[9], const double T[3],
                             const uint16_t* depth, int h, int w,
                             const uint8_t* tex, int th, int tw, ora_frame_stats* st);
/* all-core "port" of the BATCHED semantics (brick-binned like the GPU path; bit-identical map), for bench.py's cpu_baseline_allcore */
int ora_tsdf_integrate_depth_mt(ora_tsdf* m, const double R[9], const double T[3], const uint16_t* depth, int h, int w,
                                int nthreads, ora_frame_stats* st);
/* which legal serialisation of the reference's racy ray loop FAITHFUL / IDEAL replay: kind 0 struct-for order (default), 1 random ray
 * order (seed), 2 `param` threads over contiguous shares of the struct-for order, one step per turn (tsl_oracle.c, process_new_pcl) */
void ora_tsdf_set_schedule(ora_tsdf* m, int kind, int param, uint64_t seed);
int ora_tsdf_integrate_points(ora_tsdf* m, int mode, const double R[9], const double T[3],
                              const float* xyz, const uint8_t* rgb, int64_t n, ora_frame_stats* st);

int64_t ora_tsdf_count_active(const ora_tsdf* m);
/* sparse export of the ACTIVE submap, sorted by (i,j,k) ascending.  color_h may be NULL. */
int64_t ora_tsdf_export_sparse(const ora_tsdf* m, int16_t* idx, uint16_t* tsdf_h, uint16_t* w_h,
                               int8_t* occ, uint16_t* color_h, int64_t cap);
int     ora_tsdf_import_sparse(ora_tsdf* m, int sid, const int16_t* idx, const uint16_t* tsdf_h,
                               const uint16_t* w_h, const int8_t* occ, const uint16_t* color_h, int64_t n);
/* occupancy (occupy != 0) voxel list of the active submap, sorted; returns count */
int64_t ora_tsdf_export_occupied(const ora_tsdf* m, int16_t* idx, int8_t* occ, int64_t cap);

int64_t ora_tsdf_surface_voxels(const ora_tsdf* m, float* xyz, float* rgb, int64_t cap);
int64_t ora_tsdf_slice_voxels(const ora_tsdf* m, float z, float dz, float* xyz, float* val, float* rgb, int64_t cap);

int ora_tsdf_fuse_submaps(ora_tsdf* global, const ora_tsdf* sub, int mode);
int ora_tsdf_fuse_accumulate_dense(ora_tsdf* global, const ora_tsdf* sub, int64_t* acc, int32_t* cnt);
int ora_tsdf_fuse_finalize_dense(ora_tsdf* global, const int64_t* acc, const int32_t* cnt);

/* marching cubes over one map (active submap of a submap collection, or submap 0 of a global map).
 * verts/normals: [3*max_tri][3] f32, colors may be NULL.  returns triangle count (may exceed max_tri;
 * only the first max_tri are stored). */
int64_t ora_mesh_generate(const ora_tsdf* m, int step, float surface_thres, int64_t max_tri,
                          float* verts, float* normals, float* colors);

void ora_tsdf_query_points(const ora_tsdf* m, int mode, int param, const float* xyz, int64_t n, uint8_t* out);
void ora_tsdf_query_raycast(const ora_tsdf* m, const float* pos, const float* dir, float max_dist, int64_t n, uint8_t* hit, float* end_xyz, float* len);

/* ---- Octomap hit counter (taichi_octomap.py) ---- */
typedef struct {
    double map_size_xy, map_size_z, voxel_scale;
    double min_occupy_thres;
    int    texture_enabled;
    double min_ray_length, max_ray_length;
    int    K;
    int    max_submap_num;
    double disp_ceiling, disp_floor;
    int    is_global_map;
    int    recast_step;
    int    color_same_proj;
} ora_octo_cfg;
typedef struct ora_octo ora_octo;
ora_octo* ora_octo_create(const ora_octo_cfg* cfg);
void      ora_octo_destroy(ora_octo* m);
void      ora_octo_get_dims(const ora_octo* m, int* N, int* Nz, int* Rxy, int* Rz, double* voxel_scale);
void      ora_octo_set_intrinsics(ora_octo* m, const double Kdep[9], const double Kcol[9]);
void      ora_octo_set_base_pose_submap(ora_octo* m, int sid, const double R[9], const double T[3]);
void      ora_octo_set_active_submap(ora_octo* m, int sid);
void      ora_octo_reset(ora_octo* m);
int       ora_octo_integrate_depth(ora_octo* m, const double R[9], const double T[3],
                                   const uint16_t* depth, int h, int w, const uint8_t* tex, int th, int tw,
                                   ora_frame_stats* st);
int       ora_octo_integrate_points(ora_octo* m, const double R[9], const double T[3],
                                    const float* xyz, const uint8_t* rgb, int64_t n, ora_frame_stats* st);
/* leaf export (sorted by index) of the active submap: idx int32[n][3], count f32[n] */
int64_t   ora_octo_export_leaves(const ora_octo* m, int32_t* idx, float* cnt, float* rgb /* nullable, f32[n][3] */, int64_t cap);
/* taichi_octomap.py:90-102 at tree level `level` (0 = leaf); xyz f32[n][3] sorted by node index */
int64_t   ora_octo_occupied_voxels(const ora_octo* m, int level, float* xyz, float* rgb /* nullable */, int64_t cap);
int       ora_octo_fuse_submaps(ora_octo* global, const ora_octo* sub);

/* ---- ESDF (definitions from dense_esdf.py:228-333) ---- */
/* Full (non-incremental) 26-neighbourhood quasi-Euclidean ESDF of the active submap computed by
 * Dijkstra from the fixed band |TSDF| < gamma.  out_* sized by ora_tsdf_count_active(). */
int64_t ora_esdf_compute(const ora_tsdf* m, float gamma, float max_dist, int16_t* idx, float* esdf, int64_t cap);

/* f16 helpers exposed for the KATs */
uint16_t ora_f32_to_f16(float f);
float    ora_f16_to_f32(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
