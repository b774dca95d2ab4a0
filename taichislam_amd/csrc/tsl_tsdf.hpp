// tsl_tsdf.hpp -- host-side state of one DenseTSDF handle (internal; the public surface is include/taichislam_hip.h)
#pragma once
#include "tsl_common.hpp"

namespace tsl {

// per-call constants handed to the per-frame kernels (dense_tsdf.py:188-270 uses them as ti.static constants)
struct FrameParams {
    float R[9], T[3];                  // input_R / input_T  (mapping_common.py:12-13,149-156)
    float fx, fy, cx, cy;              // depth intrinsics   (mapping_common.py:31-41)
    float fxc, fyc, cxc, cyc;          // colour intrinsics  (mapping_common.py:43-58)
    float vs;                          // voxel_scale as f32
    float rvs; int fastdiv;            // RN(1/vs) and whether x/vs == fma-refined x*rvs was verified for every float (div_vs)
    float thr_max, thr_min;            // max/min_ray_length*1000  (dense_tsdf.py:198)
    float max_ray_f;                   // dense_tsdf.py:177
    float max_steps_f;                 // max_ray_length/voxel_scale (dense_tsdf.py:249)
    float internal_f;                  // internal_voxels
    int   pcl_lo, pcl_ext, pcl_bits;   // sensor-centred grid: index range [lo, lo+ext), bits per axis
    int   step, hh, ww, H, W;          // recast_step, visited rows/cols, image size
    int   rstride;                     // elements between two VISITED rows of the depth buffer: step * W for the caller's image, W / step for a staged host image (only its visited pixels are copied)
    int   cstride;                     // ... and between two visited pixels of a row: step, or 1 for a staged host image
    int   th, tw, tex, same_proj;
    int   slot;                        // map-side submap slot written by this frame
    int   variant, split;              // integrate kernel variant / lanes per ray
    int   group;                       // 1: group pixels per sensor voxel through a hash table, 0: stable radix sort (rocPRIM)
    int   seq, pcl_blk;                // sequential semantics (tsl_sequential.hip): k_segments leaves every ray's struct-for key and step count; block size of the sensor grid
    int   hlog2;                       // log2 of the part of the set's hash table this frame uses (>= 2 slots per visited pixel: the table stays cache-resident)
    const void* input; int total;      // device pointer of the depth image / point array of this frame, pixels or points to visit
    const uint8_t* tex_input; int points;   // texture [th][tw][3] (depth input) or rgb [n][3] (point input); input kind
};

// Device-side view of ONE frame's working set.  Everything up to the brick-sorted ray segments depends only on the
// depth image, the pose and the map geometry -- not on the map contents -- so it is computed per frame in its own set
// on its own stream ("phase A"), several frames in flight; only the LDS integration + finalise ("phase B") runs in
// frame order on the main stream.
struct FrameDev {
    // ---- per set ----
    uint32_t *keys, *keys_s, *vals, *vals_s;     // sensor-grid Morton key + pixel id, unsorted / sorted
    uint2* pix;                                  // per pixel: f16 bits {x,y | z,depth}
    uint4* rayA;                                 // per ray: {p01, p2|d0, d12, w bits}
    int*   rayN;                                 // per ray: step count
    uint32_t* rayFirst;                          // per ray: index of the first pixel / point of its sensor voxel (texture: colour winner order)
    struct HSlot* htab; int hlog2;               // sensor-voxel hash table (open addressing, 64-byte slots with the pixel ids inline); hlog2 = log2 of the allocated slots
    int*   act;                                  // sensor voxels opened in this frame (arrival order), count in counters[6]
    int*   actx;                                 // overflow slots (pixels 14, 15, ... of a crowded sensor voxel) opened in this frame, count in counters[7]; listed to be cleared
    uint2* colpix;                               // [pixel] f16 colour {r|g<<16, b} of the ray opened by that pixel (texture)
    tsl_frame_stats* stats;                      // header: stats | nrays | counters[8], zeroed by one memset per frame
    int*   nrays;                                // ray count of this frame
    int*   counters;                             // [1] active bricks [2] appended segments [3] segments laid out [6] sensor voxels [7] overflow slots of crowded voxels [11] frame overflow bits
                                                 // [16..19] parts per class; in the header of a batch's FIRST frame: [12] next rank to claim [13] merge-slab slots [20..23] units per class
    unsigned long long *seg, *seg_sorted;        // ray segments (one brick each), unsorted / sorted by brick
    int   *bhist, *bcursor, *boffset, *bnseg;    // [nb3] per-brick segment count / scatter cursor (zero between uses) / first segment, segment count (valid for this frame's bricks)
    int   *bslab;                                // [nb3] heavy bricks of the frame: first slab slot | parts << 20 (k_plan; valid for the bricks the batch lists as heavy)
    int   *act_b;                                // [max_frame_bricks] active bricks of the frame, in the order they were listed
    int4  *part_tab; int part_cap;               // integrate work list of the frame's heavy bricks, one table per length class (k_plan)
    int4  *unit_tab; int unit_cap;               // first set of a batch: the batch's units (bricks integrated for all frames by one workgroup), one table per class
    int4  *heavy_tab;                            // first set of a batch: [max_frame_bricks] the batch's heavy bricks { brick id, pool index, frames with segments, - } (k_plan -> k_apply_slab)
    // ---- shared by the sets of a batch (acc / accw) or by all sets: only touched in phase B, i.e. serially on the main stream ----
    int*   slot_tab;                             // variants 0/1: [nb3] brick id -> frame scratch slot, EMPTY between frames
    int*   touched;                              // variants 0/1: [max_frame_bricks] -> pool brick
    int*   touched_b;                            // variants 0/1: [max_frame_bricks] -> brick id
    unsigned long long* acc;                     // merge slab [max_frame_bricks][4096][2]  {num, den} 2^-24 fixed point: one slot per part of a heavy brick (variant 2), written whole by that part;
                                                 //   variants 0/1: per-frame brick scratch, zero between launches
    uint32_t* accw;                              // [max_frame_bricks][4096] colour winner (first pixel + 1) per slab slot (texture)
    long long* dbg;                              // developer timing counters (TSL_TIMING builds)
    int    seg_cap;
    int    max_frame_bricks;
    int    max_points;
};

// ray segment (k_segments -> k_scatter -> brick kernels): [0,6) step count  [6,18) first step  [18,40) ray id  [40,64) brick id
// (variants 0/1 staging: [18,42) ray id  [42,58) frame slot of the brick)
#define SEG_CNT_BITS 6
#define SEG_J_BITS   12
#define SEG_RAY_BITS 24
#define SEG_SLOT_SHIFT (SEG_CNT_BITS + SEG_J_BITS + SEG_RAY_BITS)
#define SEG_MAX_CNT 63
#define STG_RAY_BITS 22
#define STG_B_SHIFT (SEG_CNT_BITS + SEG_J_BITS + STG_RAY_BITS)
#define PART_NP_BITS 12
#define SLAB_SLOT_BITS 20         // bslab word: first slot | parts << 20
#define PLAN_NCLS 4
#define HDR_FAIL 11            // header words (FrameDev.counters): frame overflow bits
#define HDR_CLAIM 12           //   batch (first frame's header): next rank to claim
#define HDR_SLAB 13            //   batch: merge-slab slots handed out
#define HDR_HEAVY 14           //   batch: heavy bricks listed
#define HDR_PARTS 16           //   [16..19] parts per class
#define HDR_UNITS 20           //   batch: [20..23] units per class
#define HDR_CLAIM2 24          //   batch: next rank to claim of the parts-only launch (split launches)
#define HDR_SEQ_TUPLES 26      //   [26..27] sequential semantics: ray-step tuples reserved in the frame's tuple arrays (one 64-bit counter)
#define HDR_SEQ_LONG 28        //   batch: sequential semantics: voxels listed for the long role of k_seq_replay
#define HDR_SEQ_XLONG 30       //   batch: ... of those, the ones with the longest chains (walked first)
#define HDR_SEQ_MAXRUN 31      //   sequential semantics: the longest run of a voxel in this frame (updates)
#define HDR_SEQ_CLAIM 35       //   sequential semantics: next item of the frame a workgroup of k_seq_group claims (two passes over the slots: heavy items, then the others)
#define HDR_SEQ_SLOTS 29       //   sequential semantics: slots (= k_seq_group work items) of the frame

// A frame that runs out of its own scratch (bit 1: frame bricks / parts, bit 2: ray segments) is not integrated at all: the flag
// lives in the frame's header (counters[11], cleared by the frame's prologue), so it cannot leak into other frames, and it is
// mirrored into the handle's sticky word, which the host reports (TSL_ERR_CAPACITY) from the next call that synchronises.
__device__ __forceinline__ void frame_fail(const MapDev& M, const FrameDev& F, int bits) { atomicOr(&F.counters[11], bits); atomicOr(M.err, bits); }

// ---- sensor voxel -> ray  (process_point dense_tsdf.py:230-234, process_new_pcl :242-249); citations are to dense_tsdf.py ----
// Sensor-voxel hash table of a frame's working set (hash grouping of the pixels, k_voxelize_* -> k_segments).  One 64-byte slot = one cache
// line: the voxel's key, its pixel count and the ids of its first H_INL pixels in arrival order.  Pixels H_INL, H_INL + 1, ... of a crowded voxel
// go to further slots keyed (voxel, block = rank / H_INL): same table, same probing, no second pass over the pixels and nothing to wait for.
// An empty slot is all zero (key 0, count 0): the table starts as a zero fill, the slots a frame opened are listed (act / actx) and zeroed
// again by k_scatter, so the table is empty between frames.
#define H_INL 13
#define H_BLOCK_SHIFT 44      // key = (voxel key (3 x <= 14 bits, Morton) | block << 44) + 1
#define H_EMPTY 0ull
__device__ __forceinline__ unsigned long long h_key(unsigned long long vkey, int block) { return (vkey | ((unsigned long long)block << H_BLOCK_SHIFT)) + 1ull; }
#define GROUP_BIG_CAP 16384   // pixels one sensor voxel may hold (beyond: error bit 3, the ray is dropped)
struct HSlot { unsigned long long key; int cnt; uint32_t pix[H_INL]; };
static_assert(sizeof(HSlot) == 64, "one slot per 64 bytes");
__device__ __forceinline__ uint32_t h_hash64(unsigned long long key, int log2n) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2n)); }

struct PixAcc { int cnt; h16 sx, sy, sz, zs, cr, cg, cb; };

// new_pcl_sum_color += rgb  (:234)
__device__ __forceinline__ void acc_colour(const FrameParams& P, const FrameDev& F, uint32_t pid, PixAcc& A)
{
    const uint8_t* rgb;
    if (P.points) rgb = P.tex_input + (size_t)pid * 3;                           // :179-183
    else {
        const int jj = (int)pid / P.ww, ii = (int)pid - jj * P.ww;
        const int pj = jj * P.step, pi = ii * P.step;
        if (P.same_proj) rgb = P.tex_input + ((size_t)pj * P.tw + pi) * 3;       // :206
        else {                                                                   // color_ind_from_depth_pt  mapping_common.py:43-58
            int ci = (int)((((float)pi - P.cx) / P.fx) * P.fxc + P.cxc);
            int cj = (int)((((float)pj - P.cy) / P.fy) * P.fyc + P.cyc);
            if (ci < 0 || ci >= P.th || cj < 0 || cj >= P.tw) { ci = 0; cj = 0; }          // the reference tests column against rows (:56)
            if (cj >= P.th || ci >= P.tw) { ci = 0; cj = 0; }                            // keep the read inside the buffer
            rgb = P.tex_input + ((size_t)cj * P.tw + ci) * 3;
        }
    }
    A.cr = hadd(A.cr, f2h((float)rgb[0])); A.cg = hadd(A.cg, f2h((float)rgb[1])); A.cb = hadd(A.cb, f2h((float)rgb[2]));
}
// the pixel's f16 payload {x,y | z,depth} joins the voxel's sums: per-add f16 rounding, so the ORDER of the calls is part of the result
__device__ __forceinline__ void acc_payload(const uint2 pl, PixAcc& A)
{
    A.sx = hadd(A.sx, (h16)(pl.x & 0xffffu)); A.sy = hadd(A.sy, (h16)(pl.x >> 16));              // :231
    A.sz = hadd(A.sz, (h16)(pl.y & 0xffffu)); A.zs = hadd(A.zs, (h16)(pl.y >> 16));              // :232
    ++A.cnt;                                                                                       // :230
}
__device__ __forceinline__ void acc_pixel(const FrameParams& P, const FrameDev& F, uint32_t pid, PixAcc& A)
{
    if (P.tex) acc_colour(P, F, pid, A);
    acc_payload(F.pix[pid], A);
}

// mean point -> ray record; false for degenerate rays (zero length / z^2 not in (0, inf))
__device__ __forceinline__ bool finish_ray(const FrameParams& P, const FrameDev& F, const PixAcc& A, uint32_t first, uint4* rec, int* nsteps, bool write_colour = true)
{
    const h16 c = f2h((float)A.cnt);                                                 // :242
    const h16 px = hdiv(A.sx, c), py = hdiv(A.sy, c), pz = hdiv(A.sz, c);            // :243
    const h16 len = hsqrt(hadd(hadd(hmul(px, px), hmul(py, py)), hmul(pz, pz)));     // :244
    const h16 zbar = hdiv(A.zs, c);                                                  // :247
    const float lenf = h2f(len), zzf = h2f(hmul(zbar, zbar));
    if (!((lenf > 0.0f) && isfinite(lenf) && (zzf > 0.0f) && isfinite(zzf))) return false;
    const h16 dx = hdiv(px, len), dy = hdiv(py, len), dz = hdiv(pz, len);            // :245
    float nf = lenf / P.vs + P.internal_f;                                           // :249
    if (P.max_steps_f < nf) nf = P.max_steps_f;
    *nsteps = (int)nf;
    float w = 1.0f / zzf;                                                            // w_x_p :216-225 (d >= 0 always, Q3)
    if (w > TSL_W_CLAMP) w = TSL_W_CLAMP;
    rec->x = (uint32_t)px | ((uint32_t)py << 16);
    rec->y = (uint32_t)pz | ((uint32_t)dx << 16);
    rec->z = (uint32_t)dy | ((uint32_t)dz << 16);
    rec->w = __float_as_uint(w);
    if (P.tex && write_colour) {                                                     // color = sum_color/c/255  (:269)
        const h16 r16 = f2h(h2f(hdiv(A.cr, c)) / 255.0f), g16 = f2h(h2f(hdiv(A.cg, c)) / 255.0f), b16 = f2h(h2f(hdiv(A.cb, c)) / 255.0f);
        F.colpix[first] = make_uint2((uint32_t)r16 | ((uint32_t)g16 << 16), (uint32_t)b16);
    }
    return true;
}


// Sequential semantics on the brick pipeline (tsl_sequential.hip): per frame working set, the ray steps of every (frame, brick) stably grouped by voxel
// as 8-byte replay tuples { w, w * sd } (tup) with the run offsets of the brick's 4096 voxels (csr).  (`stash`: round 4's intermediate copy of every step
// in replay order, { signed distance f32 | voxel 12 | z^2 f16 }; only allocated by -DTSL_SEQ_STASH builds.)  Lives in device memory (the batch's working
// sets already fill the 4 KiB of kernel arguments).
#define SQ_CSR_STRIDE 4104            // words per (frame, brick): 4097 run offsets | first tuple of the brick's region | "a tuple outside the fast path's range"
#define SQ_CSR_BASE 4097
#define SQ_CSR_UNSAFE 4098
struct SeqDev {
    unsigned long long* stash;        // [cap]
    float2* tup;                      // [cap + 16]
    uint32_t* csr;                    // [slot_cap][SQ_CSR_STRIDE]
    int4* items; int slot_cap;        // k_seq_group's work items of the frame, one per slot: { first segment, segments, slot, 1: in the spare segment array }
    uint32_t* stash_ray;              // textured maps: [cap] ray of every stashed tuple
    uint32_t* tup_ray;                // textured maps: [cap] ray of every replay tuple (a run's last one colours the voxel)
    long long cap;
};

struct ProfSlot { hipEvent_t a, b; int kid; int count; };

// Frames are queued and processed TSL_NB at a time: phase A of a whole batch runs as one sequence of launches (grid.y = frame)
// on the batch's stream while phase B of the previous batch runs on the main stream; two batches are in flight.
#define TSL_NB 8
#define TSL_NBATCH 3          // batch slots: phase A of up to two batches is in flight beside phase B of a third
#define TSL_NSTREAMS 3        // phase-A streams shared by the batch slots (the device runs main + 3 queues efficiently)
#define TSL_NSETS (TSL_NB * TSL_NBATCH)
struct FSet {
    FrameDev F; void* sort_temp; void* header; size_t header_bytes;
    void* stage_in; size_t stage_in_bytes; void* stage_tex; size_t stage_tex_bytes;      // staging of host-pointer inputs of the frame queued into this set
    void* pin; size_t pin_bytes; void* pin_dev;      // pinned, device-mapped host buffer the host copies the visited rows (and the texture) of a host-pointer input into
    size_t copy_in, copy_tex, copy_tex_off;          // bytes of the queued frame's input / texture waiting in `pin` for the batch's copy kernel (0: a device input)
    FrameParams* Pd;                       // this frame's parameters in device memory (written by the frame's prologue kernel)
    std::vector<void*> owned;
};
// kernel argument of the batched phase-A kernels: the working sets and (device) parameter blocks of the frames of one batch
struct BatchDev { FrameDev f[TSL_NB]; const FrameParams* p[TSL_NB]; int n; };
struct ParamPack { FrameParams p[TSL_NB]; };
struct SetPtrs { FrameParams* p[TSL_NB]; int* header[TSL_NB]; };        // k_set_params: where the parameters go, the headers to clear
struct BatchHost { hipStream_t st; hipEvent_t a_done, b_done, p_done, c_done; bool b_pending, a_recorded, c_recorded; };      // p_done: the batch's parts-only brick launch (split launches); c_done: its host inputs have left the pinned buffers
struct StageCopy { const uint4* src[2 * TSL_NB]; uint4* dst[2 * TSL_NB]; int n16[2 * TSL_NB]; };      // k_stage_host: per frame of the batch its input and its texture, in 16-byte units
#define TSL_INFLIGHT 8          // batches the host may run ahead of the device

int fuse_submaps_sequential(tsl_tsdf* g, tsl_tsdf* sub, const float* pose_dev, int nsrc);      // tsl_sequential.hip
int esdf_finish(tsl_tsdf* m);            // tsl_esdf.hip: wait for the ESDF updates in flight (repairing one that stopped early)
void esdf_release(tsl_tsdf* m);
}  // namespace tsl

#define TSL_ESDF_SLOTS 4
struct EsdfSlot { hipEvent_t ev; int* host; int rounds; tsl_esdf_stats st; };      // one ESDF update in flight: its counters land in `host`
struct tsl_tsdf {
    tsl_tsdf_cfg cfg;
    int device;
    hipStream_t stream_;                 // phase B + everything else; entry points take it through tsl::ms(), which first issues queued frames
    tsl::FSet fset[TSL_NSETS]; tsl::BatchHost batch[TSL_NBATCH]; hipStream_t copy_st;   // copy_st: H2D copies of host-pointer inputs
    int overlap;                         // frames per batch (0 = one frame at a time on the main stream)
    int cur, npend, pend_points; tsl::FrameParams pend[TSL_NB]; int deferred_rc;      // frames queued for batch `cur`
    int64_t frames_issued, frames_consumed;  // frames handed to the device so far / of those, frames whose inputs have been read
    int64_t batch_seq; hipEvent_t ring_ev[TSL_INFLIGHT]; int64_t ring_upto[TSL_INFLIGHT];      // back-pressure ring: end of phase B of the last TSL_INFLIGHT batches
    int last_set, last_batch_n;          // working set of the frame queued last; frames of the batch issued last
    hipStream_t producers[4]; int nproducers;   // producer streams of the queued device inputs (ordered before phase A when the batch is issued)
    hipEvent_t in_ev[8]; int in_ev_next;  // cached events ordering callers' producer streams before the input-reading stream (tsl_tsdf_input_stream)
    bool scratch_ready;                  // frame scratch allocated (first integrate call)
    int N, Nz, nbx, nbz, nb3, nsub, npose;
    int pcl_lo, pcl_ext, pcl_bits;
    tsl::MapDev M;
    tsl::FrameDev F;
    tsl::FrameParams P;                  // constants filled at create / set_intrinsics
    std::vector<double> baseR, baseT;    // submaps_base_R_np / T_np  (mapping_common.py:106-107)
    std::vector<float> baseRf, baseTf;   // f32 field copies          (mapping_common.py:104-105)
    double gbaseR[9], gbaseT[3];
    int active;
    float surf_thres, disp_floor, disp_ceiling;
    void* sort_temp; size_t sort_temp_bytes;
    tsl_frame_stats* h_stats;            // pinned
    int* h_ints;                         // pinned scratch (128 ints)
    // export buffers (export_TSDF_xyz / export_color / export_TSDF, num_TSDF_particles)  dense_tsdf.py:53-60
    float *exp_xyz, *exp_rgb, *exp_val; int* num_particles; int64_t max_disp;
    float* colormap;                     // [1024][3]
    float* pose_dev;                     // [npose][12] f32 (R row-major, T) for fusion
    // sparse export staging
    void* xbuf; size_t xbuf_bytes;
    // mesh buffers (mesh_vertices / mesh_normals / mesh_colors, num_facelets)  marching_cube_mesher.py:16-22
    uint32_t* mesh_flags;                        // per pool brick: bit 0 a stored TSDF value < 0, bit 1 one that is not (k_mc_summary)
    float *mesh_v, *mesh_n, *mesh_c; int* mesh_count; int64_t mesh_cap; int mesh_gather;      // mesh_gather: option, 1 = global-gather kernel also for step 1 (A/B)
    void *fuse_acc, *fuse_cnt, *fuse_cacc; bool fuse_dirty;      // global-map fusion scratch ({num,den} int64 pairs, count|occupancy, colour sums); dirty: a splat was not followed by its finalise / pack
    uint8_t* mrg_mask; int *mrg_list, *mrg_count; int mrg_nunion;      // multi-GPU merge: touched-brick mask, union list (tsl_merge.hip)
    void *mrg_pacc, *mrg_pcnt; size_t mrg_pacc_bytes, mrg_pcnt_bytes;  // packed union bricks of the one-call form
    void *mrg_racc, *mrg_rcnt, *mrg_rec; size_t mrg_racc_bytes, mrg_rcnt_bytes, mrg_rec_bytes; int merge_exchange;      // option "merge_exchange" 1: this rank's reduced slice, the gathered records
    // esdf
    float* esdf; uint8_t *esdf_fl, *esdf_region, *esdf_par, *esdf_ok; int esdf_mode, esdf_grid; bool fuse_direct; long long esdf_orphans; int *esdf_list, *esdf_queue, *esdf_ctr, *esdf_inq, *esdf_nbr; uint32_t* esdf_note; int esdf_qcap;      // tsl_esdf.hip
    float *esdf_exp_xyz, *esdf_exp_val; int* esdf_exp_count; int esdf_exp_n;      // export_ESDF_xyz / export_ESDF / num_export_ESDF_particles (dense_esdf.py:498-509), allocated by the first slice
    bool esdf_valid, esdf_force_full; int esdf_submap; float esdf_gamma, esdf_maxd; tsl_esdf_stats esdf_stats;
    hipEvent_t esdf_gate, esdf_gate_ev; bool esdf_gate_set; unsigned esdf_gate_mask;      // recorded behind the collect kernel of the latest ESDF update: phase A of later frames waits for it
                                                                                           // -- on EVERY phase-A stream (mask: the streams that have waited since the update was queued)
    hipEvent_t esdf_in, esdf_read, esdf_last;      // option "esdf_overlap": the relaxation rounds of update n run beside the integration of frame n + 1.
                                                   // esdf_in: the TSDF an update starts from; esdf_read: the update has read it; esdf_last: the latest update
    bool esdf_overlap; int esdf_ctr_idx;
    void *fseq_keys[2], *fseq_vals[2], *fseq_temp; size_t fseq_bytes[2], fseq_vbytes[2], fseq_tbytes; void* fseq_ctr;      // sequential fusion (tsl_sequential.hip)
    EsdfSlot esdf_slot[TSL_ESDF_SLOTS]; int esdf_tail, esdf_npend, esdf_rounds_seen, esdf_round_cap; bool esdf_short; tsl_esdf_totals_t esdf_tot;   // updates in flight (tsl_esdf.hip)
    // profiling
    bool prof_on, prof_open, prof_group; unsigned prof_mask; std::vector<tsl::ProfSlot> prof; std::vector<hipEvent_t> prof_free;
    double prof_ms[TSL_K_COUNT]; int64_t prof_n[TSL_K_COUNT];
    int semantics;                       // 0: BATCHED (exact per-frame sums applied once), 1: the reference-literal sequential replay (tsl_sequential.hip)
    int seq_impl;                        // 1: per-brick replay runs built on the brick pipeline (default), 0: round 3's two global radix sorts (one frame per batch; kept as a cross-check)
    bool seq_ready; int64_t seq_bytes0;      /* bytes of the literal scratch inside m->bytes: seq_release takes exactly those out of the account again */
    tsl::SeqDev seq_h[TSL_NSETS]; tsl::SeqDev* seq_d;      // seq_impl 1: tuple arrays of every working set (allocated by the first sequential batch)
    void *seqb_keys[TSL_NBATCH][2], *seqb_vals[TSL_NBATCH][2], *seqb_temp[TSL_NBATCH], *seqb_long[TSL_NBATCH], *seqb_lmask[TSL_NBATCH], *seqb_perm[TSL_NBATCH]; size_t seqb_temp_bytes; long long seq_tuple_cap;      // per batch slot: the rays' struct-for keys of all its frames, sorted in one call
    unsigned long long *seq_keys[2], *seq_vals[2], *seq_ctr; void* seq_temp; size_t seq_temp_bytes; long long seq_cap;
    int variant, split, phases, wg, spt, ncu, chunks, unit_max, unit_half, unit_floor, bgrid, ugrid, pgrid, split_launch, adaptive, ramp, ramp_batches, ramp_size; bool clean; uint64_t batch_gen;
    int64_t bytes;
    void* seqv_sum[TSL_NBATCH]; int* seqv_log;      // TSL_SEQ_VERIFY (developer aid, tsl_sequential.hip): per-item checksums, mismatch log
    uint32_t shape_hash; int dry_launches;      // developer statistics: FNV hash over the sizes of the batches issued so far; batches issued into a dry pipeline
};

namespace tsl {
int  grow(void** p, size_t* have, size_t need);
hipStream_t ms(tsl_tsdf* m);                                                 // main stream, after issuing the queued frames
int  flush_pending(tsl_tsdf* m);
void prof_begin(tsl_tsdf* m, int kid, hipStream_t st = nullptr, int count = 1);
void prof_end(tsl_tsdf* m, hipStream_t st = nullptr);
bool prof_slot(tsl_tsdf* m, int kid, int count, hipEvent_t* a, hipEvent_t* b);
void convert_pose(const double* Rb, const double* Tb, const double* R, const double* T, float* outR, float* outT);
int  dev_alloc(tsl_tsdf* m, void** p, size_t bytes, int fill);
int  check_variant2(tsl_tsdf* m);
int  launch_segments(tsl_tsdf* m, const BatchDev& B, const FrameParams* hp, int total, hipStream_t st);      // phase A tail: rays -> brick-sorted segments
int  launch_apply(tsl_tsdf* m, FSet& S, int total);                          // phase B, variants 0/1: apply one frame to the map
int  launch_apply_batch(tsl_tsdf* m, const BatchDev& B, const FrameParams& P, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
int  launch_brick(tsl_tsdf* m, const BatchDev& B, const FrameParams& P, int kind, hipStream_t st, hipEvent_t start, hipEvent_t stop);
int  launch_slab_apply(tsl_tsdf* m, const BatchDev& B, const FrameParams& P);
int  launch_seq_group(tsl_tsdf* m, const BatchDev& B, const FrameParams* hp, int bi, hipStream_t st);      // tsl_sequential.hip, phase A tail: replay ranks of the rays, per-brick replay runs
int  launch_seq_apply(tsl_tsdf* m, const BatchDev& B, const FrameParams& P, int bi);                       // tsl_sequential.hip, phase B of a batch: every voxel's runs applied in frame order
void seq_release(tsl_tsdf* m);
int  seq_prepare(tsl_tsdf* m);
int  selftest_seqdiv(unsigned long long* bad_dev);
int  seq_verify_report(tsl_tsdf* m, int* out, int cap);
int  launch_apply_sequential(tsl_tsdf* m, const BatchDev& B, const FrameParams& P);      // tsl_sequential.hip: phase B of one frame, sequential semantics      // phase B, variant 2: apply a batch of frames (one launch)
}
