/*
 * taichislam_hip.h -- C-ABI of the MI355X-native dense-mapping backend (libtaichislam_hip.so).
 *
 * The reference (xuhao1/TaichiSLAM) has no FFI seam: its callers use the Python classes of
 * taichi_slam/mapping directly.  This ABI is what taichislam_amd/mapping/ *.py (ctypes shims that
 * keep those class surfaces) binds, and what any C/C++ host would bind.  Each entry point cites the
 * reference method it replaces (paths relative to the reference root).
 *
 * Conventions: opaque handles; plain pointers and sizes; int status return (0 = TSL_OK, <0 = error,
 * text via tsl_last_error()); no exceptions cross the boundary.  One handle = one device + one HIP
 * stream; a handle is NOT thread-safe, distinct handles are independent.  Pointers named *_dev are
 * device pointers (e.g. torch tensor .data_ptr()); all others are caller-owned host buffers.
 * Integration calls are asynchronous on the handle's stream; every call that returns data to the
 * host synchronises.  R/T are row-major float64 camera-to-world poses (scripts/taichislam_node.py:381).
 */
#ifndef TAICHISLAM_HIP_H
#define TAICHISLAM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSL_OK              0
#define TSL_ERR_ARG        -1
#define TSL_ERR_HIP        -2
#define TSL_ERR_CAPACITY   -3   /* brick pool / frame scratch / output buffer exhausted */
#define TSL_ERR_NO_DEVICE  -4

typedef struct tsl_tsdf tsl_tsdf;   /* DenseTSDF                 taichi_slam/mapping/dense_tsdf.py:12      */
typedef struct tsl_octo tsl_octo;   /* Octomap                   taichi_slam/mapping/taichi_octomap.py:12  */

/* DenseTSDF.__init__ kwargs (dense_tsdf.py:13-16) + backend sizing knobs (0 = default). */
typedef struct {
    double  map_size_xy, map_size_z;
    double  voxel_scale;
    int32_t num_voxel_per_blk_axis;
    double  max_ray_length, min_ray_length;
    int32_t internal_voxels;
    int32_t max_submap_num;
    int32_t is_global_map;
    int32_t texture_enabled;
    double  disp_ceiling, disp_floor;
    int32_t recast_step;
    int32_t color_same_proj;
    int64_t max_disp_particles;
    /* backend sizing */
    int32_t max_bricks;          /* 16^3 brick pool capacity (24 KiB each)                */
    int32_t max_frame_bricks;    /* bricks one frame may touch (64 KiB scratch each)      */
    int32_t max_points;          /* pixels/points per integrate call                       */
} tsl_tsdf_cfg;

typedef struct {
    int64_t p_used;      /* pixels / points visited                                   */
    int64_t p_valid;     /* passed the range gate and inside the sensor-centred grid  */
    int64_t p_oob;       /* passed the gate but outside the sensor-centred grid       */
    int64_t v_pcl;       /* sensor-grid voxels with count>0 (= rays)                  */
    int64_t v_skipped;   /* degenerate rays skipped                                   */
    int64_t steps;       /* ray-steps applied (S)                                     */
    int64_t steps_oob;   /* ray-steps outside the map volume (skipped)                */
    int64_t unique;      /* distinct voxels touched this frame (U)                    */
    int64_t bricks;      /* distinct 16^3 bricks touched this frame                   */
} tsl_frame_stats;

/* kernel ids for tsl_tsdf_prof_query */
enum { TSL_K_VOXELIZE = 0, TSL_K_SORT = 1, TSL_K_RAYS = 2, TSL_K_INTEGRATE = 3, TSL_K_FINALIZE = 4,
       TSL_K_MESH = 5, TSL_K_SEGMENTS = 6, TSL_K_BIN = 7, TSL_K_ESDF = 8, TSL_K_FUSE = 9, TSL_K_COUNT };

const char* tsl_version(void);
const char* tsl_last_error(void);
int  tsl_device_count(int* n);
/* Exhaustive device check (all 2^32 float patterns) of an arithmetic shortcut the kernels rely on for bit-exactness:
   which = 0: three-instruction round-half-away == ti.round (mapping_common.py:263-266, dense_tsdf.py:254);
   which = 1: rescale-free correctly rounded sqrt == sqrtf on [2^-96, inf);
   which = 2 (2^32 pseudo-random operand pairs, not exhaustive): the division-free quotient of the sequential replay's saturated voxels
              (reciprocal product + two FMA residual corrections) == IEEE division.  *mismatches must come back 0. */
int  tsl_selftest(int which, int64_t* mismatches);

/* ---- lifecycle -------------------------------------------------------------------------- */
int  tsl_tsdf_create(const tsl_tsdf_cfg* cfg, int device, tsl_tsdf** out);        /* dense_tsdf.py:13-50,52-118 */
void tsl_tsdf_destroy(tsl_tsdf* m);
int  tsl_tsdf_get_dims(const tsl_tsdf* m, int32_t* N, int32_t* Nz, int32_t* block_num_xy, int32_t* block_num_z);
int  tsl_tsdf_sync(tsl_tsdf* m);
int  tsl_tsdf_reset(tsl_tsdf* m);                                                    /* dense_tsdf.py:309-310 */
int  tsl_tsdf_memory_bytes(const tsl_tsdf* m, int64_t* bytes);
int  tsl_tsdf_bricks_in_use(tsl_tsdf* m, int32_t* n);

/* ---- poses / camera ------------------------------------------------------------------------ */
int  tsl_tsdf_set_intrinsics(tsl_tsdf* m, const double Kdep[9], const double Kcol[9]);   /* mapping_common.py:25-29 */
int  tsl_tsdf_set_base_pose(tsl_tsdf* m, const double R[9], const double T[3]);           /* mapping_common.py:141-147 */
int  tsl_tsdf_set_base_pose_submap(tsl_tsdf* m, int sid, const double R[9], const double T[3]);   /* mapping_common.py:121-131 */
int  tsl_tsdf_get_active_submap(const tsl_tsdf* m, int32_t* sid);                         /* mapping_common.py:113-114 */
int  tsl_tsdf_set_active_submap(tsl_tsdf* m, int32_t sid);                                /* mapping_common.py:116-119 */
int  tsl_tsdf_set_colormap(tsl_tsdf* m, const float rgb[1024 * 3]);                       /* mapping_common.py:158-163 */

/* ---- integration (the hot path) --------------------------------------------------------------
 * recast_depth_to_map(R, T, depthmap, texture)  dense_tsdf.py:162-165,188-270
 * depth: uint16 millimetres [h][w] C-contiguous; tex: uint8 [th][tw][3] or NULL. */
int  tsl_tsdf_integrate_depth(tsl_tsdf* m, const double R[9], const double T[3],
                              const uint16_t* depth, int h, int w, const uint8_t* tex, int th, int tw);
int  tsl_tsdf_integrate_depth_dev(tsl_tsdf* m, const double R[9], const double T[3],
                                  const void* depth_dev, int h, int w, const void* tex_dev, int th, int tw);
/* recast_pcl_to_map(R, T, xyz_array, rgb_array)  dense_tsdf.py:157-160,167-186; xyz f32 [n][3] */
int  tsl_tsdf_integrate_points(tsl_tsdf* m, const double R[9], const double T[3],
                               const float* xyz, const uint8_t* rgb, int64_t n);
int  tsl_tsdf_integrate_points_dev(tsl_tsdf* m, const double R[9], const double T[3],
                                   const void* xyz_dev, const void* rgb_dev, int64_t n);
/* Stream that will read the device buffers of the next integrate_*_dev call (points: 0 depth image, 1 point cloud).  With `ordered`,
 * that stream is made to wait (when the frame's batch is issued) for everything queued so far on `producer` (the hipStream_t the caller fills its buffers on; NULL =
 * the default stream): the reference's recast_* calls are synchronous (dense_tsdf.py:157-165), so the Python shim does this for torch
 * tensors with torch's current stream. */
int  tsl_tsdf_input_stream(tsl_tsdf* m, int points, int ordered, void* producer, void** hip_stream);
/* The three calls a caller with device buffers makes per frame, as one: input_stream(ordered, producer) + integrate_depth_dev +
 * frames_consumed (the shim's per-frame path for torch tensors: one FFI crossing instead of three). */
int  tsl_tsdf_integrate_depth_stream(tsl_tsdf* m, const double R[9], const double T[3], const void* depth_dev, int h, int w,
                                     const void* tex_dev, int th, int tw, void* producer, int64_t* queued_total, int64_t* consumed);
/* frames queued by integrate_* calls and not yet issued to the device: a device buffer handed to integrate_*_dev is read by kernels
 * that are only enqueued once this has dropped back to 0 (or any synchronising call was made) */
int  tsl_tsdf_queued_frames(const tsl_tsdf* m, int32_t* n);
/* Lifetime of device input buffers without synchronising: *queued_total = frames handed to integrate_* since the handle was created
 * (the frame of the latest call has index queued_total - 1), *consumed = how many of them the device has certainly finished reading.
 * The buffers of frame i may be reused or freed once consumed > i.  Host-side bookkeeping only (no device query, never blocks): the
 * library never queues more than eight batches (64 frames) ahead of the device -- the integrate call that would exceed that waits for
 * the oldest batch -- and that wait, like every synchronising call, advances the count. */
int  tsl_tsdf_frames_consumed(tsl_tsdf* m, int64_t* queued_total, int64_t* consumed);
/* The integrate calls only QUEUE the frame (host buffers are copied before they return -- the visited pixels of a depth image, the points, the texture,
 * into a pinned buffer of the frame's working set, from where ONE copy kernel per batch takes them to device memory at the head of the batch;
 * device buffers must stay unchanged until the next call that returns data, or tsl_tsdf_sync).  A batch that could not be issued (an allocation
 * or launch failure) is reported by the next call that returns data or synchronises -- never by a map that silently lacks its frames.  Queued frames are issued eight at a time (four for the first two batches after the pipeline ran dry), or as soon
 * as any other call needs the map (option "adaptive": also as soon as the device is ready for them), so results never depend on the queueing; frames still queued when a handle is destroyed are dropped. */
/* counters of the most recent integrate call (synchronises) */
int  tsl_tsdf_last_frame_stats(tsl_tsdf* m, tsl_frame_stats* out);

/* ---- sparse export / import  (dense_tsdf.py:412-498) -------------------------------------------- */
int  tsl_tsdf_count_active(tsl_tsdf* m, int64_t* n);                                       /* :412-423 */
int  tsl_tsdf_export_sparse(tsl_tsdf* m, int16_t* idx, uint16_t* tsdf_h, uint16_t* w_h, int8_t* occ,
                            uint16_t* color_h, int64_t cap, int64_t* n);                   /* to_numpy :425-440 */
int  tsl_tsdf_import_sparse(tsl_tsdf* m, int sid, const int16_t* idx, const uint16_t* tsdf_h,
                            const uint16_t* w_h, const int8_t* occ, const uint16_t* color_h, int64_t n);   /* load_numpy :442-454 */
int  tsl_tsdf_export_occupied(tsl_tsdf* m, int16_t* idx, int8_t* occ, int64_t cap, int64_t* n);

/* ---- visualisation exports (device-resident result buffers, max_disp_particles rows) ---------- */
/* cvt_TSDF_surface_to_voxels[_to]  dense_tsdf.py:323-365.  add_to_cur: keep the current count and
 * append (the `_to` form); the true count is returned even when it exceeds the capacity. */
int  tsl_tsdf_surface_voxels(tsl_tsdf* m, tsl_tsdf* dst /* NULL = m itself */, int add_to_cur, int32_t* n);
/* cvt_TSDF_to_voxels_slice(z, dz, clear_last)  dense_tsdf.py:367-389 */
int  tsl_tsdf_slice_voxels(tsl_tsdf* m, float z, float dz, int clear_last, int32_t* n);
/* read back export_TSDF_xyz / export_color / export_TSDF (any may be NULL), rows [0, n) */
int  tsl_tsdf_read_exports(tsl_tsdf* m, float* xyz, float* rgb, float* val, int64_t n);
/* the same three buffers as DEVICE pointers (f32 [max_disp_particles][3] / [3] / [1], valid for the lifetime of the handle) + the count
 * of the last cvt_* call: the form taichislam_node.py:350-351 would use if its consumer stayed on the GPU.  Synchronises. */
int  tsl_tsdf_exports_dev(tsl_tsdf* m, void** xyz_dev, void** rgb_dev, void** val_dev, int32_t* n);
/* export_TSDF_xyz[row] = v (field 0) / export_color[row] = v (field 1): callers append marker points to the particle list (tests/gen_topo_graph.py:64-65) */
int  tsl_tsdf_set_export_row(tsl_tsdf* m, int field, int64_t row, const float v[3]);
/* the first n exported particles as the data block of a sensor_msgs/PointCloud2: interleaved float32 rows x y z [r g b], point_step 12 / 24
 * (what utils/ros_pcl_transfer.py:96-136 builds from the numpy copies, scripts/taichislam_node.py:420-425) -- interleaved on the device */
int  tsl_tsdf_pack_pointcloud2(tsl_tsdf* m, int has_rgb, int64_t n, void* out_host);
int  tsl_tsdf_num_particles(tsl_tsdf* m, int32_t* n);
int  tsl_tsdf_set_num_particles(tsl_tsdf* m, int32_t n);

/* ---- submap fusion  (dense_tsdf.py:272-318) ------------------------------------------------------ */
int  tsl_tsdf_fuse_submaps(tsl_tsdf* global, tsl_tsdf* submaps);
/* ---- multi-GPU global-map merge: one submap collection per GPU, one exchange at merge time -------------------------------------
 * Swarm counterpart of submap_mapping.py:226-253 + utils/communication.py:9-43 (agents ship zlib'd numpy submaps over LCM and every
 * agent fuses them, dense_tsdf.py:272-318): every rank splats its own submaps into exact 2^-24 fixed-point sums per touched 16^3
 * brick, the union of touched bricks is all-reduced (RCCL over xGMI) and every rank writes the same global TSDF.  Integer sums: the
 * result is bit-identical for any number of ranks and equal to one GPU fusing every submap.  The global map's pose table must hold the
 * base pose of every submap id used by any rank (tsl_tsdf_set_base_pose_submap), as for tsl_tsdf_fuse_submaps. */
typedef struct tsl_comm tsl_comm;                                  /* an RCCL communicator (RCCL is bound at run time, dlopen) */
int   tsl_comm_unique_id(char id[128]);                            /* ncclGetUniqueId on one rank; ship the 128 bytes to the others */
int   tsl_comm_create(const char id[128], int nranks, int rank, int device, tsl_comm** out);   /* ncclCommInitRank (collective) */
void  tsl_comm_destroy(tsl_comm* c);
void* tsl_comm_handle(tsl_comm* c);                                /* the ncclComm_t */
/* one call: splat, all-reduce the touched-brick mask (MAX) and the packed union bricks (SUM) on `rccl_comm` (an ncclComm_t from
 * tsl_comm_handle or the caller's own; NULL = this rank alone), finalise.  bytes_per_rank (nullable) = payload all-reduced. */
int   tsl_tsdf_allreduce_merge(tsl_tsdf* global, tsl_tsdf* submaps, void* rccl_comm, int64_t* bytes_per_rank);
/* the same in steps, for callers that run the two reductions themselves (torch.distributed, MPI); *_dev buffers are the caller's:
 *   merge_begin  : reset `global`, splat `submaps`, write the touched-brick byte mask (tsl_tsdf_merge_mask_bytes bytes)
 *   -> all-reduce(MAX) the mask
 *   merge_union  : ascending list of the union bricks, *nunion of them
 *   merge_pack   : acc_dev int64 [nunion][4096][2] = {sum w*tsdf, sum w}, cnt_dev int32 [nunion][4096] = contributions * 65536 + occupancy
 *   -> all-reduce(SUM) both
 *   merge_finish : write the global map from the sums */
int   tsl_tsdf_merge_mask_bytes(const tsl_tsdf* global, int64_t* n);
int   tsl_tsdf_merge_begin(tsl_tsdf* global, tsl_tsdf* submaps, void* mask_dev, int64_t mask_bytes);
int   tsl_tsdf_merge_union(tsl_tsdf* global, const void* mask_dev, int32_t* nunion);
int   tsl_tsdf_merge_pack(tsl_tsdf* global, void* acc_dev, void* cnt_dev);
int   tsl_tsdf_merge_finish(tsl_tsdf* global, const void* acc_dev, const void* cnt_dev);
/* The second form of the exchange: REDUCE-SCATTER the packed planes (pad them to whole bricks per rank with zeros), every rank turns the
 * `nbricks` bricks of sums it received into finalised records -- per brick tsl_tsdf_merge_record_bytes = 20 992 bytes: f16 {TSDF, W} words, occupancy bytes, a
 * "written" bit per voxel (dense_tsdf.py:272-280 applied once per voxel) -- and with the records of all union bricks ALL-GATHERED (union order) the map is written:
 *   merge_begin -> all-reduce(MAX) -> merge_union -> merge_pack -> reduce-scatter(SUM) x2 -> merge_finalize_slice -> all-gather -> merge_finish_records.
 * 5.1 bytes per union voxel travel in the second half instead of the 20 an all-reduce sends round again.  tsl_tsdf_allreduce_merge takes this form with
 * option "merge_exchange" = 1 on the global map (ncclReduceScatter / ncclAllGather). */
int   tsl_tsdf_merge_record_bytes(int64_t* n);
int   tsl_tsdf_merge_finalize_slice(tsl_tsdf* global, const void* acc_dev, const void* cnt_dev, int32_t nbricks, void* rec_dev);
int   tsl_tsdf_merge_finish_records(tsl_tsdf* global, const void* rec_dev);

/* ---- marching cubes  (marching_cube_mesher.py:127-193) ------------------------------------------- */
/* generate_mesh(step): result stays on the device in the map's mesh buffers (3*max_tri rows each);
 * n_tri is the true triangle count.  read with tsl_mesh_read. */
int  tsl_mesh_generate(tsl_tsdf* m, int step, float surface_thres, int64_t max_tri, int32_t* n_tri);
int  tsl_mesh_read(tsl_tsdf* m, float* verts, float* normals, float* colors, int64_t n_vertices);
/* mesh_vertices / mesh_normals / mesh_colors (marching_cube_mesher.py:16-22) as DEVICE pointers, f32 [3 * max_triangles][3] (colours NULL
 * for untextured maps), + num_facelets of the last generate: taichislam_node.py:342 without the host copy */
int  tsl_mesh_buffers_dev(tsl_tsdf* m, void** verts_dev, void** normals_dev, void** colors_dev, int32_t* n_tri);

/* ---- batched map queries  (mapping_common.py:165-204, dense_tsdf.py:148-155; consumers: topo_graph.py:444-507) ---------- */
/* mode 0: is_pos_occupy, 1: is_pos_unobserved, 2: is_near_pos_occupy(param voxels); xyz f32 [n][3] in the active submap's frame */
int  tsl_tsdf_query_points(tsl_tsdf* m, int mode, int param, const float* xyz, int64_t n, uint8_t* out);
/* raycast(pos, dir, max_dist) per query: hit flag, last evaluated position, length travelled */
int  tsl_tsdf_query_raycast(tsl_tsdf* m, const float* pos, const float* dir, float max_dist, int64_t n, uint8_t* hit, float* end_xyz, float* len);
/* the same with DEVICE buffers, asynchronous: launched on the handle's stream (behind every queued frame) after the work already queued
 * on `user_stream` (a hipStream_t, e.g. torch's current stream; NULL = the legacy default stream), and `user_stream` waits for the
 * result -- no host round trip per 64-128-ray node expansion (topo_graph.py:444-507). */
int  tsl_tsdf_query_points_dev(tsl_tsdf* m, int mode, int param, const void* xyz_dev, int64_t n, void* out_u8_dev, void* user_stream);
int  tsl_tsdf_query_raycast_dev(tsl_tsdf* m, const void* pos_dev, const void* dir_dev, float max_dist, int64_t n,
                                void* hit_u8_dev, void* end_xyz_dev, void* len_dev, void* user_stream);

/* ---- ESDF  (dense_esdf.py:228-333 as the definition) -----------------------------------------------------------------
 * |TSDF| < gamma: ESDF = TSDF; elsewhere the 26-neighbour quasi-Euclidean distance (edge cost |dir| * voxel) to that band along voxels
 * of the same sign, capped at max_dist.  tsl_esdf_update is INCREMENTAL: the integrate kernels mark the bricks they write, an update
 * re-initialises and relaxes only those bricks dilated by max_dist (device-side work lists, no host synchronisation inside) and
 * yields exactly the map a full recompute yields.  The first update, one after reset / import / fusion or with other parameters, and
 * every update when option "esdf_full" is set, covers all bricks.  n_relaxed (nullable) = brick relaxations performed. */
typedef struct {
    int32_t incremental;         /* 0: all bricks were recomputed */
    int32_t dirty_bricks;        /* bricks written since the previous update */
    int32_t changed_bricks;      /* ... of which the ESDF inputs (observed / sign / band membership / band value of a voxel) changed */
    int32_t region_bricks;       /* bricks re-initialised and relaxed (dirty bricks dilated by max_dist) */
    int32_t total_bricks;        /* bricks of the handle */
    int64_t brick_relaxations;   /* LDS relaxations run (a brick is revisited when its surroundings change) */
    int64_t voxel_pushes;        /* voxel values lowered by the sweeps (a voxel may be lowered more than once) */
    int32_t rounds, max_passes;  /* relaxation rounds that had work; most sweep sets (six concurrent directional sweeps) one brick relaxation needed */
    int32_t raise_sets;          /* esdf_mode 1: sets of raise sweeps (parent links re-derived) over all brick visits */
    int64_t passes;              /* sweep sets over all brick relaxations */
    int64_t voxels_raised;       /* esdf_mode 1: of voxel_pushes, the values re-derived through their parent link by the raise sweeps (the rest was lowered) */
    int32_t max_raise_sets, reserved_;
} tsl_esdf_stats;
/* sums over the updates of the handle that have completed */
typedef struct {
    int64_t updates, incremental, dirty_bricks, region_bricks, brick_relaxations, voxel_pushes, passes;
} tsl_esdf_totals_t;
/* n_relaxed != NULL: waits for the update and returns its brick relaxations.  n_relaxed == NULL: ASYNCHRONOUS -- the update is only
 * enqueued (behind everything queued on the handle so far; up to 4 may be in flight; the per-frame hook of dense_esdf.py:400-402 uses this form), nothing is
 * waited for.  last_stats / totals / export wait for the outstanding updates first, so what they return is always complete. */
int  tsl_esdf_update(tsl_tsdf* m, float gamma, float max_dist, int32_t* n_relaxed);
int  tsl_esdf_last_stats(tsl_tsdf* m, tsl_esdf_stats* out);
int  tsl_esdf_totals(tsl_tsdf* m, tsl_esdf_totals_t* out);
int  tsl_esdf_export(tsl_tsdf* m, int16_t* idx, float* esdf, int64_t cap, int64_t* n);
/* the same compaction left on the device: *idx_dev = int16 [min(n, cap)][3], *val_dev = f32 [min(n, cap)] in the handle's staging buffer
 * (valid until the next exporting / importing / host-buffer query call on the handle) */
int  tsl_esdf_export_dev(tsl_tsdf* m, int64_t cap, void** idx_dev, void** val_dev, int64_t* n);
/* cvt_ESDF_to_voxels_slice(z)  dense_esdf.py:498-509: ESDF of the voxel layer at height z of the active submap -> export_ESDF_xyz /
 * export_ESDF (max_disp_particles rows on the device), *n = num_export_ESDF_particles; read back with tsl_esdf_read_slice or take the
 * device pointers (valid for the lifetime of the handle) with tsl_esdf_slice_dev */
int  tsl_esdf_slice(tsl_tsdf* m, float z, int32_t* n);
int  tsl_esdf_read_slice(tsl_tsdf* m, float* xyz, float* val, int64_t n);
int  tsl_esdf_slice_dev(tsl_tsdf* m, void** xyz_dev, void** val_dev, int32_t* n);

/* backend knobs for A/B-ing kernel variants: name in
     "variant"  0|1: one global int64 atomic pair per ray step, 2 (default): brick-binned LDS accumulation
     "semantics" 0 (default): a frame's contributions to a voxel are summed exactly and applied once (order-free, oracle mode BATCHED);
                1: the reference-literal SEQUENTIAL replay of dense_tsdf.py:236-270 -- rays in Taichi's struct-for order, every ray step an f16
                read-modify-write with the W clamp, colours by last writer; on a GLOBAL map: tsl_tsdf_fuse_submaps replays fuse_submaps_kernel
                (dense_tsdf.py:272-318) the same way, submap cells in struct-for order, corners in loop order -- bit-exact with oracle FAITHFUL and
                with the maps the reference's own source produces on tools/ti_seq (tests/golden/ref_*.npz); needs variant 2 and group 1, at
                most 2^21 points per frame and maps of at most 2^17 bricks.  Integration and fusion both follow the reference's struct-for order for any
                num_voxel_per_blk_axis of the submaps, 4..32 (the fusion orders the source cells by the sorted list of the blocks their 16^3 storage bricks overlap).
                Switching the mode on allocates the replay scratch of "seq_impl" 1 (an allocation failure is reported by this call)
     "seq_impl" how semantics 1 integrates.  1 (default): on the brick pipeline, whole batches -- behind phase A every (frame, brick) gets its
                ray steps as 8-byte tuples, stably grouped by voxel in replay order (k_seq_group: persistent workgroups claim the items heavy-first; LDS sample
                sort of the brick's segments by ray rank, a counting walk, then a placing pass that evaluates every step at its replay position),
                phase B is one thread per voxel applying its runs frame after frame (k_seq_replay);
                memory: 8 bytes x "seq_tuple_cap" + 16 KiB x (max_frame_bricks + 1024) per working set, 24 working sets -- ~4.1 GB at the
                defaults -- allocated when "semantics" is set to 1 (round 5: a step is evaluated twice and written once; the first form also kept every
                step in a "stash" array in replay order, 1.5 GB more and 16 bytes of traffic per step).  0: round 3's form -- every ray step a 16-byte tuple, two global radix sorts, one frame per batch
     "seq_longest_run" (get only) the longest run of updates of one voxel in a frame, summed over the frames of the batch issued last (the voxel next
                to the sensor; a wave of its own settles it 64 updates per evaluation where its f16 state has stopped moving)
     "seq_long_voxels" (get only) voxels of the batch issued last that were replayed by a wave of their own (a run of >= 64 updates in a frame)
     "overlapped_launches" / "dry_launches" (get only) batches issued while phase B of the batch before was still pending / into a pipeline that had run
                dry: what the back-to-back parity tests assert on (tests/test_pipeline_overlap_gpu.py); "batch_shape_hash" (get only): FNV hash over the
                sizes of the batches issued so far (two runs that batched the stream alike have the same hash)
     "seq_verify_mismatches" (get only; developer aid, environment TSL_SEQ_VERIFY=1, else -1) literal mode: every (frame, brick) work item is recomputed by
                brute force behind k_seq_group and its offsets / tuples are checksummed behind k_seq_group, in front of the replay and behind it;
                the number of disagreements logged (records on stderr).  Found round 5's one-brick difference
     "seq_tuple_cap" ray steps one frame may produce under seq_impl 1 (default 2^23; a frame beyond it is dropped with TSL_ERR_CAPACITY);
                synchronises; scratch that exists already is rebuilt at the new size
     "overlap"  0 = one frame at a time on the main stream, n = frames per batch (default and maximum 8; three batch slots: phase A of up to two
                batches is in flight beside phase B of a third)
     "adaptive" 1 = queued frames are also issued as soon as phase A of the previous batch has completed (a slow sensor gets every frame
                integrated on arrival; full batches form by themselves when the producer outruns the device), 0 (default) = a batch is
                issued when it is full -- half full for the first two batches after the pipeline ran dry -- or when anything reads the map
     "ramp"     short batches issued after the pipeline ran dry before full ones are waited for (default 2; with 1 a 20-frame burst runs as 4 + 8 + 8: 18.5 k against 18.3 k frames/s in an A/B, inside the box-to-box spread, while the launches of the burst get longer per frame), "ramp_size" their length (default 4)
     "group"    1 (default) = hash grouping of the pixels of a sensor voxel, 0 = stable radix sort
     "split"    lanes per ray (divides 64; the brick-binned path uses at most 8), default 2
     "wg"       threads per workgroup of the brick integrate kernel: 512 (default) or 256 (two workgroups per CU)
     "spt"      segments per thread and step of the brick kernel: 4 (steps of 2048 segments, one 512-thread workgroup per CU) or 2 (steps of 1024,
                <= 128 VGPRs: two 512-thread workgroups = 16 waves per CU)
     "unit"     a brick whose segments of a whole batch number at most this is walked by ONE workgroup, frame after frame, with its voxels in
                registers; heavier bricks are split into parts that leave their sums in slab slots of their own, applied by k_apply_slab
     "unit_floor" the unit limit is quoted for a full batch and scaled with the frames of a shorter one, but not below this (default 4096)
     "unit_half" bricks with more segments per batch than this (and at most "unit") are walked half as a unit (their first frames) and half as
                parts (their later frames): halves the longest serial chain of a launch; >= "unit" disables the middle tier
     "chunks"   steps a part may hold (1..8, default 4)
     "bgrid"    resident workgroups of the brick kernel in percent of the slots (default 75: the rest is left to phase A of the next batch)
     "split_launch" 1 = full batches of the overlapped pipeline launch the brick kernel twice: the PARTS of the heavy bricks (they read
                their frame's rays and write slab slots of their own, never the map) on the batch's phase-A stream, beside the previous
                batch's phase B, and the UNITS + k_apply_slab on the handle's stream; 0 (default) = one launch over the whole work list.
                Measured slower (27 k against 30 k frames/s): the brick kernel is bound by the SIMDs' issue rate, two launches side by side
                slow each other by more than the shorter chain saves
     "ugrid" / "pgrid" resident workgroups of the units / parts launch in percent of the slots (defaults 75 / 25)
     "mesh_gather" 1 = marching cubes reads every value through the brick table also at step 1 (default: brick + halo staged in LDS)
     "esdf_full" 1 = every tsl_esdf_update recomputes all bricks (the reference for the incremental update)
     "esdf_mode" 0 (default) = regional recompute: the bricks whose ESDF inputs changed, dilated by max_dist, are re-initialised and relaxed;
                 1 = raise / lower wavefront with a parent direction per voxel (dense_esdf.py:96, :255-333): only the voxels whose parent chain passes
                 through a changed voxel are re-derived, then lowered -- a third of the voxel writes, the same map bit for bit, ~1.45x the time on the
                 benchmark stream (tsl_esdf.hip has the measurements); read-only "esdf_orphans" = lowered voxels without a supporting neighbour (always 0)
     "esdf_grid" n > 0 = workgroups of a relaxation round (default 0 = four per CU; developer A/B)
     "merge_exchange" (on the GLOBAL map) 0 (default) = tsl_tsdf_allreduce_merge all-reduces the packed sums of the union bricks; 1 = reduce-scatter of the
                 sums, every rank finalises its slice, all-gather of the finalised 5.1-byte voxels (the step form: tsl_tsdf_merge_finalize_slice / _finish_records)
     "fuse_window_misses" (read-only, on the GLOBAL map) corner splats of the LDS-window fusion that fell outside their 15^3 window and went straight to memory
                 (0 for every pose tested: the window is sized for the worst rotation; a miss costs time, not correctness)
     "fuse_direct" (on the GLOBAL map) 1 = the fusion splat of round 5, one set of global atomics per corner (A/B); 0 (default) = sums gathered per 8^3 source
                 block in a 15^3 LDS window first.  Textured maps always take the direct form
     "esdf_overlap" 1 (default) = an update's kernels run on one of the handle's phase-A streams: the relaxation rounds of update n overlap
                    the integration of frame n + 1 (which waits only until the update has read the TSDF); 0 = on the handle's stream
     "esdf_round_cap" n > 0 = launch at most n relaxation rounds per update (test knob: an update that stops early must be repaired)
     "fastdiv"  0 = force IEEE division
     "phases"   developer timing aid: 1 = phase A only, 2 = phase B only (the map contents are then meaningless), 3 = both */
int  tsl_tsdf_set_option(tsl_tsdf* m, const char* name, int value);
int  tsl_tsdf_get_option(tsl_tsdf* m, const char* name, int* value);   /* also "fastdiv": 1 when the device verified the fma-refined division;
                                                                           read-only: "last_heavy_bricks" / "last_slab_slots" = bricks walked in parts and
                                                                           merge-slab slots handed out by the batch issued last (synchronises) */

/* ---- environment switches ------------------------------------------------------------------------
 * The product library (lib/libtaichislam_hip.so) reads ONE environment variable:
 *   TSL_SEQ_VERIFY=1   semantics = 1 only: every work item of the literal mode is recounted by brute force from its segment list and an order-free
 *                      checksum of its tuples is compared behind k_seq_group, in front of the replay and behind it; mismatches are counted in
 *                      get_option("seq_verify_mismatches").  Slows the mode down by ~10x; never changes a result.
 * Everything else is compiled only into the developer build lib/libtaichislam_hip_testhooks.so (-DTSL_TEST_HOOKS, built beside the product library
 * by taichislam_amd/build.py; load it with TSL_LIB=<path> in the Python shims):
 *   TSL_FAULT_NO_BDONE_WAIT=1   FAULT INJECTION: phase A of a batch does not wait for the phase B that still reads the batch slot's working sets --
 *                               the map becomes wrong; tests/test_pipeline_overlap_gpu.py uses it to show that the parity tests see the overlap
 *   TSL_PIN_LEGACY=1            host staging buffers without hipHostMallocCoherent (A/B of round 4's allocation)
 *   TSL_EV_SYS=1                pipeline events with the default system-scope fence (A/B of round 5's hipEventDisableSystemFence)
 *   TSL_SEQ_SPLIT_ROLES=1       the two roles of k_seq_replay as separate launches (profiling aid)
 * (the sort-grouped phase A that TSL_GROUP_SORT selected is the backend option "group" = 0.) */

/* ---- profiling: HIP-event timing of the per-frame kernels on the handle's stream ----------------- */
int  tsl_tsdf_prof_enable(tsl_tsdf* m, int on);   /* 0 off, 1 every kernel, 2*mask: only the kernel ids whose bit is set in mask */
int  tsl_tsdf_prof_query(tsl_tsdf* m, int kernel_id, double* total_ms, int64_t* launches);   /* synchronises, resets */

/* ======== Octomap hit counter (taichi_octomap.py) =================================================== */
typedef struct {
    double  map_size_xy, map_size_z, voxel_scale;
    double  min_occupy_thres;
    int32_t texture_enabled;
    double  min_ray_length, max_ray_length;
    int32_t K;
    int32_t max_submap_num;
    double  disp_ceiling, disp_floor;
    int32_t is_global_map;
    int32_t recast_step;
    int32_t color_same_proj;
    int64_t max_disp_particles;
    int32_t max_bricks;
    int32_t max_points;
} tsl_octo_cfg;

int  tsl_octo_create(const tsl_octo_cfg* cfg, int device, tsl_octo** out);        /* taichi_octomap.py:14-84 */
void tsl_octo_destroy(tsl_octo* m);
int  tsl_octo_get_dims(const tsl_octo* m, int32_t* N, int32_t* Nz, int32_t* Rxy, int32_t* Rz, double* voxel_scale);
int  tsl_octo_sync(tsl_octo* m);
int  tsl_octo_reset(tsl_octo* m);                                                    /* :210-211 */
int  tsl_octo_set_intrinsics(tsl_octo* m, const double Kdep[9], const double Kcol[9]);
int  tsl_octo_set_base_pose_submap(tsl_octo* m, int sid, const double R[9], const double T[3]);
int  tsl_octo_get_active_submap(const tsl_octo* m, int32_t* sid);
int  tsl_octo_set_active_submap(tsl_octo* m, int32_t sid);
/* host image.  Untextured: the visited pixels are copied into a pinned, device-mapped slot of a ring at once (the image may be reused on return) and the frame is
 * queued like a device-resident one: no copy call, no synchronisation per frame (30 k frames/s from numpy images).  Textured: staged and inserted at once. */
int  tsl_octo_integrate_depth(tsl_octo* m, const double R[9], const double T[3], const uint16_t* depth, int h, int w,
                              const uint8_t* tex, int th, int tw);                   /* :130-132,147-169; tex u8[th][tw][3] BGR (:120-124) or NULL */
/* device-resident depth (and texture).  Untextured frames are only QUEUED: up to eight are inserted by ONE launch (the insert is an order-free count), issued when
 * eight are queued or as soon as any other call on the handle needs the map or its stream -- invisible except in timing.  The buffers must stay unchanged until
 * tsl_octo_sync (or any call that returns map contents). */
int  tsl_octo_integrate_depth_dev(tsl_octo* m, const double R[9], const double T[3], const void* depth_dev, int h, int w,
                                  const void* tex_dev, int th, int tw);
int  tsl_octo_integrate_points(tsl_octo* m, const double R[9], const double T[3], const float* xyz, const uint8_t* rgb, int64_t n);   /* :126-128,134-145 */
int  tsl_octo_integrate_points_dev(tsl_octo* m, const double R[9], const double T[3], const void* xyz_dev, const void* rgb_dev, int64_t n);   /* device buffers: f32 [n][3], u8 [n][3] or NULL */
int  tsl_octo_last_frame_stats(tsl_octo* m, tsl_frame_stats* out);
/* every touched leaf of the active submap: index, hit count and (textured maps; else zeros) colour f32[n][3]; rgb may be NULL */
int  tsl_octo_export_leaves(tsl_octo* m, int32_t* idx, float* cnt, float* rgb, int64_t cap, int64_t* n);
/* cvt_occupy_to_voxels(level) / cvt_occupy_voxels_to(...)  :90-114 */
int  tsl_octo_occupied_voxels(tsl_octo* m, tsl_octo* dst /* NULL = m */, int level, int add_to_cur, int32_t* n);
int  tsl_octo_read_exports(tsl_octo* m, float* xyz, float* rgb, int64_t n);
int  tsl_octo_exports_dev(tsl_octo* m, void** xyz_dev, void** rgb_dev, int32_t* n);   /* export_x / export_color as device pointers + num_export_particles (taichislam_node.py:330-333) */
/* the first n rows of export_x [+ export_color] as a sensor_msgs/PointCloud2 data block (interleaved f32 x y z [r g b]): the publisher of
 * scripts/taichislam_node.py:330-333 + utils/ros_pcl_transfer.py:96-136, interleaved on the device */
int  tsl_octo_pack_pointcloud2(tsl_octo* m, int has_rgb, int64_t n, void* out_host);
int  tsl_octo_num_particles(tsl_octo* m, int32_t* n);
int  tsl_octo_fuse_submaps(tsl_octo* global, tsl_octo* submaps);                     /* :171-199 */

#ifdef __cplusplus
}
#endif
#endif
