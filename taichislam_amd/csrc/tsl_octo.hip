// tsl_octo.hip -- Octomap hit counter on MI355X.  Replaces taichi_slam/mapping/taichi_octomap.py:14-211 (reference root).
//
// The reference stores one f32 hit count per leaf of a K-ary pointer tree (11 pointer levels at 1024^3) and walks the
// tree for every insert.  Here the leaves live in 16^3 bricks of f32 behind a direct-mapped brick table per submap
// (tables are created lazily by the host when a submap becomes active); an insert is one table lookup + one f32 atomic.
// Counts are small integers in f32, so the sums are exact and independent of the order of the atomics.  Textured maps keep one
// f32 colour per leaf; who colours a leaf is decided by an atomicMax over the writer's index instead of the reference's race.
#include "tsl_common.hpp"
#include <cmath>

namespace tsl {

struct OctoDev {
    int N, Nz, hN, hNz, ext_xy, ext_z, nbx, nbz, nb3;
    int max_bricks;
    int** tables;            // [nsub] -> brick table of the submap (nullptr until the host creates it)
    float* cnt;              // [max_bricks][4096]
    float* col;              // [max_bricks][4096][3] leaf colour (texture_enabled only)
    unsigned long long* win; // [max_bricks][4096] winner of the current frame / fusion (zero in between)
    int* owner_s; int* owner_b;
    int* pool_top; int* err;
};

struct OctoParams {
    float R[9], T[3]; float fx, fy, cx, cy; float fxc, fyc, cxc, cyc; float vs; float thr_max, thr_min; int step, hh, ww, W;
    int th, tw, same_proj;
    int packed;                // the depth buffer holds the VISITED pixels only, hh rows of ww (host images handed over through the pinned slots)
};

__device__ __forceinline__ bool octo_in_tree(const OctoDev& M, int i, int j, int k)
{ return i >= -M.hN && i < M.ext_xy - M.hN && j >= -M.hN && j < M.ext_xy - M.hN && k >= -M.hNz && k < M.ext_z - M.hNz; }
__device__ __forceinline__ int octo_brick_of(const OctoDev& M, int i, int j, int k, int* local)
{
    const int ui = i + M.hN, uj = j + M.hN, uk = k + M.hNz;
    *local = ((ui & 15) << 8) | ((uj & 15) << 4) | (uk & 15);
    return ((ui >> 4) * M.nbx + (uj >> 4)) * M.nbz + (uk >> 4);
}
__device__ __forceinline__ int octo_claim(const OctoDev& M, int s, int b)
{
    int* tab = M.tables[s];
    if (!tab) return -1;
    int v = __hip_atomic_load(tab + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= 0) return v;
    v = claim_index_1(tab + b, M.pool_top, M.max_bricks);
    if (v >= 0) { M.owner_s[v] = s; M.owner_b[v] = b; } else atomicOr(M.err, 1);
    return v;
}
// process_point  taichi_octomap.py:116-124.  Returns the leaf (pool brick * 4096 + cell) or -1.  The reference lets the last
// writer colour a leaf (:120-124, a race); here the pixel / point with the largest index wins: `id1` = index + 1 enters an
// atomicMax and k_octo_colour lets the winner write (the oracle inserts in index order, so its last writer is the same one).
__device__ __forceinline__ long long octo_point(const OctoDev& M, int s, float vs, float x, float y, float z, unsigned id1)
{
    const int ci = rnd_i(x / vs), cj = rnd_i(y / vs), ck = rnd_i(z / vs);                    // mapping_common.py:252-266
    if (!octo_in_tree(M, ci, cj, ck)) return -1;
    int l; const int b = octo_brick_of(M, ci, cj, ck, &l);
    const int p = octo_claim(M, s, b);
    if (p < 0) return -1;
    const long long leaf = (long long)p * TSL_BRK3 + l;
    atomicAdd(M.cnt + leaf, 1.0f);                                                           // :119
    if (id1) atomicMax(M.win + leaf, (unsigned long long)id1);
    return leaf;
}
// texel of a depth pixel (taichi_octomap.py:160-165 with the index mapping of mapping_common.py:43-58, as in the TSDF path)
__device__ __forceinline__ const uint8_t* octo_texel(const OctoParams& P, const uint8_t* tex, int i, int j)
{
    if (P.same_proj) return tex + ((size_t)j * P.tw + i) * 3;
    int ci = (int)((((float)i - P.cx) / P.fx) * P.fxc + P.cxc);
    int cj = (int)((((float)j - P.cy) / P.fy) * P.fyc + P.cyc);
    if (ci < 0 || ci >= P.th || cj < 0 || cj >= P.tw) { ci = 0; cj = 0; }
    if (cj >= P.th || ci >= P.tw) { ci = 0; cj = 0; }
    return tex + ((size_t)cj * P.tw + ci) * 3;
}
// second pass of a textured insert: the winner of every leaf writes its colour (BGR -> RGB, :121-124) and clears the mark
__global__ void __launch_bounds__(256) k_octo_colour(OctoDev M, OctoParams P, const uint8_t* __restrict__ tex, const long long* __restrict__ leaf_of, int total, int points)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const long long leaf = leaf_of[p];
    if (leaf < 0 || M.win[leaf] != (unsigned long long)(p + 1)) return;
    const uint8_t* rgb;
    if (points) rgb = tex + (size_t)p * 3;
    else { const int jj = p / P.ww, ii = p - jj * P.ww; rgb = octo_texel(P, tex, ii * P.step, jj * P.step); }
    M.col[leaf * 3] = (float)rgb[2] / 255.0f; M.col[leaf * 3 + 1] = (float)rgb[1] / 255.0f; M.col[leaf * 3 + 2] = (float)rgb[0] / 255.0f;
    M.win[leaf] = 0ull;
}

// recast_depth_to_map_kernel  taichi_octomap.py:147-169
__global__ void __launch_bounds__(256) k_octo_depth(OctoDev M, OctoParams P, int s, const uint16_t* __restrict__ depth, tsl_frame_stats* st, long long* leaf_of)
{
    const int total = P.hh * P.ww;
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool gate = false, ok = false;
    if (p < total) {
        long long leaf = -1;
        const int jj = p / P.ww, ii = p - jj * P.ww;
        const int j = jj * P.step, i = ii * P.step;
        const uint16_t d = depth[(size_t)j * P.W + i];
        const float df = (float)d;
        if (d != 0 && !(df > P.thr_max) && !(df < P.thr_min)) {                               // :155
            gate = true;
            const float dep = df / 1000.0f;                                                   // :157
            const float px = ((float)i - P.cx) * dep / P.fx, py = ((float)j - P.cy) * dep / P.fy, pz = dep;
            const float mx = ((P.R[0] * px + P.R[1] * py) + P.R[2] * pz) + P.T[0];            // :159
            const float my = ((P.R[3] * px + P.R[4] * py) + P.R[5] * pz) + P.T[1];
            const float mz = ((P.R[6] * px + P.R[7] * py) + P.R[8] * pz) + P.T[2];
            leaf = octo_point(M, s, P.vs, mx, my, mz, leaf_of ? (unsigned)(p + 1) : 0u);
            ok = leaf >= 0;
        }
        if (leaf_of) leaf_of[p] = leaf;
    }
    block_count_add(&st->p_valid, ok);
    block_count_add(&st->p_oob, gate && !ok);
}
// The same insert for up to OCTO_NB queued frames in ONE launch (round 6; blockIdx.y = frame).  The insert is a count: leaf += 1.0f, integers below 2^24 in
// f32, exact in any order, and a brick is claimed once whoever comes first -- so frames may share a launch like pixels do.  A frame is ~10 us of kernel
// behind ~6 us of launch; eight per launch take the launch out of the per-frame cost (65 k -> 200 k+ frames/s at 640 x 480 / recast_step 2).
#define OCTO_NB 8
#define OCTO_PIN_SLOTS 16
struct OctoBatchArgs { OctoParams P[OCTO_NB]; const uint16_t* depth[OCTO_NB]; tsl_frame_stats* st[OCTO_NB]; int s[OCTO_NB]; int n; };
__global__ void __launch_bounds__(256) k_octo_depth_batch(OctoDev M, OctoBatchArgs B)
{
    const int f = blockIdx.y;
    const OctoParams& P = B.P[f];
    // a workgroup per 16 x 16 tile of sampled pixels, a wave per 4 rows x 16 columns of it: at 5 cm and 3 m some ten neighbouring pixels fall into one leaf, and
    // 76 800 same-address f32 atomics per frame on ~8 000 leaves were what the one-pixel-one-atomic kernel took its 10 us for.  The lanes of a wave are grouped by
    // leaf (lane arithmetic, one iteration per distinct leaf) and the first lane of a group adds the group's size: an integer below 2^24, exact.
    const int tiles_x = (P.ww + 15) >> 4, tiles_y = (P.hh + 15) >> 4;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;              // (uniform: the grid is sized for the largest frame of the batch)
    const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
    const int ii = tx * 16 + (int)(threadIdx.x & 15u), jj = ty * 16 + (int)(threadIdx.x >> 4);
    bool gate = false, in = false;
    int ci = 0, cj = 0, ck = 0;
    if (ii < P.ww && jj < P.hh) {
        const int j = jj * P.step, i = ii * P.step;
        const uint16_t d = B.depth[f][P.packed ? (size_t)jj * P.ww + ii : (size_t)j * P.W + i];
        const float df = (float)d;
        if (d != 0 && !(df > P.thr_max) && !(df < P.thr_min)) {                               // :155
            gate = true;
            const float dep = df / 1000.0f;                                                   // :157
            const float px = ((float)i - P.cx) * dep / P.fx, py = ((float)j - P.cy) * dep / P.fy, pz = dep;
            const float mx = ((P.R[0] * px + P.R[1] * py) + P.R[2] * pz) + P.T[0];            // :159
            const float my = ((P.R[3] * px + P.R[4] * py) + P.R[5] * pz) + P.T[1];
            const float mz = ((P.R[6] * px + P.R[7] * py) + P.R[8] * pz) + P.T[2];
            ci = rnd_i(mx / P.vs); cj = rnd_i(my / P.vs); ck = rnd_i(mz / P.vs);               // process_point :116-119, mapping_common.py:252-266
            in = octo_in_tree(M, ci, cj, ck);
        }
    }
    const int lane = lane_id();
    int leader = lane, gsize = 1;
    for (unsigned long long todo = __ballot(in); todo; ) {
        const int l0 = (int)__builtin_ctzll(todo);                                            // (uniform)
        const int i0 = __builtin_amdgcn_readlane(ci, l0), j0 = __builtin_amdgcn_readlane(cj, l0), k0 = __builtin_amdgcn_readlane(ck, l0);
        const bool mine = in && ci == i0 && cj == j0 && ck == k0;
        const unsigned long long grp = __ballot(mine);
        if (mine) { leader = l0; gsize = popc64(grp); }
        todo &= ~grp;
    }
    int got = 0;
    if (in && lane == leader) {
        int l; const int b = octo_brick_of(M, ci, cj, ck, &l);
        const int p = octo_claim(M, B.s[f], b);
        if (p >= 0) { atomicAdd(M.cnt + (long long)p * TSL_BRK3 + l, (float)gsize); got = 1; }      // :119, gsize times
    }
    got = __shfl(got, leader);
    const bool ok = in && got != 0;
    block_count_add(&B.st[f]->p_valid, ok);
    block_count_add(&B.st[f]->p_oob, gate && !ok);
}
// recast_pcl_to_map_kernel  taichi_octomap.py:134-145 (no range gate)
__global__ void __launch_bounds__(256) k_octo_points(OctoDev M, OctoParams P, int s, const float* __restrict__ xyz, int n, tsl_frame_stats* st, long long* leaf_of)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool ok = false;
    if (p < n) {
        const float px = xyz[(size_t)p * 3], py = xyz[(size_t)p * 3 + 1], pz = xyz[(size_t)p * 3 + 2];
        const float mx = ((P.R[0] * px + P.R[1] * py) + P.R[2] * pz) + P.T[0];                // :141
        const float my = ((P.R[3] * px + P.R[4] * py) + P.R[5] * pz) + P.T[1];
        const float mz = ((P.R[6] * px + P.R[7] * py) + P.R[8] * pz) + P.T[2];
        const long long leaf = octo_point(M, s, P.vs, mx, my, mz, leaf_of ? (unsigned)(p + 1) : 0u);
        ok = leaf >= 0;
        if (leaf_of) leaf_of[p] = leaf;
    }
    block_count_add(&st->p_valid, ok);
    block_count_add(&st->p_oob, p < n && !ok);
}

__device__ __forceinline__ void octo_ijk(const OctoDev& M, int b, int l, int* i, int* j, int* k)
{
    const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
    *i = bi * 16 + (l >> 8) - M.hN; *j = bj * 16 + ((l >> 4) & 15) - M.hN; *k = bk * 16 + (l & 15) - M.hNz;
}

struct OctoPose { float R[9], T[3]; };
// mode 0: every touched leaf -> (idx, count); mode 1: cvt_occupy_to_voxels(level) taichi_octomap.py:90-114 -> xyz
__global__ void __launch_bounds__(256) k_octo_export(OctoDev M, int s, int nused, int mode, float thres, int gxy, int gz, OctoPose B, float vs,
                                                     int32_t* idx, float* cnt, float* xyz, float* rgb, long long cap, int* counter)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        if (M.owner_s[p] != s) continue;
        const int b = M.owner_b[p];
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const float c = M.cnt[(size_t)p * TSL_BRK3 + l];
            int i, j, k; octo_ijk(M, b, l, &i, &j, &k);
            bool pred;
            if (mode == 0) pred = c != 0.0f;
            else pred = (c > thres) && ((i + M.hN) % gxy == 0) && ((j + M.hN) % gxy == 0) && ((k + M.hNz) % gz == 0);   // :96-97, :86-88
            const int o = wave_reserve(counter, pred);
            if (pred && o < cap) {
                if (rgb && M.col) for (int a = 0; a < 3; ++a) rgb[(size_t)o * 3 + a] = M.col[((size_t)p * TSL_BRK3 + l) * 3 + a];      // :100-101
                if (mode == 0) { idx[(size_t)o * 3] = i; idx[(size_t)o * 3 + 1] = j; idx[(size_t)o * 3 + 2] = k; cnt[o] = c; }
                else {
                    const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;      // sijk_to_xyz mapping_common.py:234-238
                    for (int a = 0; a < 3; ++a) xyz[(size_t)o * 3 + a] = ((B.R[a * 3] * p0 + B.R[a * 3 + 1] * p1) + B.R[a * 3 + 2] * p2) + B.T[a];
                }
            }
        }
    }
}

// fuse_submaps_kernel  taichi_octomap.py:171-189
// pass 0 adds the counts (:186) and, when textured, lets the source leaf with the largest (submap, i, j, k) mark the target
// (the reference's `color = submap_color` :189 is a race between the sources); pass 1 lets the marked source copy its colour.
__global__ void __launch_bounds__(256) k_octo_fuse(OctoDev S, OctoDev G, int nused, const float* poses, int npose, float vs, float thres, int pass)
{
    const bool tex = S.col && G.col;
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int s = S.owner_s[p], b = S.owner_b[p];
        if (s >= npose) continue;
        const float* Rp = poses + (size_t)s * 12;
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const float c = S.cnt[(size_t)p * TSL_BRK3 + l];
            if (!(c > thres)) continue;                                                      // :181
            int i, j, k; octo_ijk(S, b, l, &i, &j, &k);
            const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;
            int cc[3];
            for (int a = 0; a < 3; ++a) { const float x = ((Rp[a * 3] * p0 + Rp[a * 3 + 1] * p1) + Rp[a * 3 + 2] * p2) + Rp[9 + a]; cc[a] = rnd_i(x / vs); }   // :182-183
            if (!octo_in_tree(G, cc[0], cc[1], cc[2])) continue;
            int gl; const int gb = octo_brick_of(G, cc[0], cc[1], cc[2], &gl);
            const unsigned long long key1 = 1ull + (((unsigned long long)s << 54) | ((unsigned long long)(i + S.hN) << 36) | ((unsigned long long)(j + S.hN) << 18) | (unsigned long long)(k + S.hNz));
            if (pass == 0) {
                const int gp = octo_claim(G, 0, gb);
                if (gp >= 0) {
                    atomicAdd(G.cnt + (size_t)gp * TSL_BRK3 + gl, c);                          // :186
                    if (tex) atomicMax(G.win + (size_t)gp * TSL_BRK3 + gl, key1);
                }
            } else {
                const int gp = G.tables[0][gb];
                if (gp >= 0 && G.win[(size_t)gp * TSL_BRK3 + gl] == key1) {
                    const size_t src = ((size_t)p * TSL_BRK3 + l) * 3, dst = ((size_t)gp * TSL_BRK3 + gl) * 3;
                    for (int a = 0; a < 3; ++a) G.col[dst + a] = S.col[src + a];
                    G.win[(size_t)gp * TSL_BRK3 + gl] = 0ull;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_octo_reset(OctoDev M, int nused)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        float* c = M.cnt + (size_t)p * TSL_BRK3;
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) c[l] = 0.0f;
        if (M.col) for (int l = threadIdx.x; l < TSL_BRK3 * 3; l += 256) M.col[(size_t)p * TSL_BRK3 * 3 + l] = 0.0f;
        if (threadIdx.x == 0) M.tables[M.owner_s[p]][M.owner_b[p]] = TSL_EMPTY;
    }
}

void convert_pose(const double* Rb, const double* Tb, const double* R, const double* T, float* outR, float* outT);

}  // namespace tsl

struct tsl_octo {
    tsl_octo_cfg cfg; int device; hipStream_t stream_;      // (use os(m): it issues the queued depth frames first)
    int Rxy, Rz, N, Nz, K, nsub;
    double voxel_scale_recomputed;
    tsl::OctoDev M; tsl::OctoParams P;
    std::vector<int*> tables;                 // host copy of M.tables
    std::vector<double> baseR, baseT; std::vector<float> baseRf, baseTf;
    int active; float occ_thres;
    tsl_frame_stats* stats; tsl_frame_stats* h_stats; int64_t p_used;      // stats: the CURRENT frame's slot of a ring of OCTO_STAT_RING (cleared half a ring at a time: one memset per 32 frames instead of one per frame)
    tsl_frame_stats* stats_ring; int stat_idx;
    float *exp_xyz, *exp_rgb; int* num_particles; int64_t max_disp;
    float* pose_dev;
    void* stage; size_t stage_bytes; void* xbuf; size_t xbuf_bytes;
    void* stage_tex; size_t stage_tex_bytes; long long* leaf_of; size_t leaf_of_n;      // texture staging, leaf of every pixel / point of the frame
    tsl::OctoBatchArgs* q; int qmax;
    // host images: a ring of pinned, device-mapped slots the host copies the visited pixels into (no copy call, no synchronisation per frame: tsl_octo_integrate_depth);
    // a slot is written again 16 frames later, after the batch that read it has finished (qev, recorded behind every batch launch)
    void* pin[OCTO_PIN_SLOTS]; void* pin_dev[OCTO_PIN_SLOTS]; size_t pin_bytes[OCTO_PIN_SLOTS]; int pin_guard[OCTO_PIN_SLOTS]; int pin_idx;
    hipEvent_t qev[4]; int qev_idx; int qslot[8]; int next_packed_slot;          // qslot: pinned slot of the k-th queued frame (-1: the caller's device buffer)          // untextured device-resident depth frames queued for one launch (q->n of them; OCTO_NB at most)
};

using namespace tsl;

// issue the queued depth frames: one launch for all of them
static void octo_flush(tsl_octo* m)
{
    if (!m->q || m->q->n == 0) return;
    hipLaunchKernelGGL(k_octo_depth_batch, dim3((unsigned)m->qmax, (unsigned)m->q->n), dim3(256), 0, m->stream_, m->M, *m->q);
    bool pinned = false;
    for (int k = 0; k < m->q->n; ++k) if (m->qslot[k] >= 0) { m->pin_guard[m->qslot[k]] = m->qev_idx; pinned = true; }
    if (pinned && m->qev[m->qev_idx]) { (void)hipEventRecord(m->qev[m->qev_idx], m->stream_); m->qev_idx = (m->qev_idx + 1) % 4; }
    m->q->n = 0; m->qmax = 0;
}
// the handle's stream behind everything queued on it: every entry point that reads or writes the map comes through here
static hipStream_t os(tsl_octo* m) { octo_flush(m); return m->stream_; }

static int octo_ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }
static int octo_ensure_table(tsl_octo* m, int s)
{
    if (m->tables[(size_t)s]) return TSL_OK;
    int* t = nullptr;
    TSL_HIP(hipMalloc((void**)&t, sizeof(int) * (size_t)m->M.nb3));
    TSL_HIP(hipMemsetAsync(t, 0xff, sizeof(int) * (size_t)m->M.nb3, os(m)));
    m->tables[(size_t)s] = t;
    TSL_HIP(hipMemcpyAsync(m->M.tables + s, &m->tables[(size_t)s], sizeof(int*), hipMemcpyHostToDevice, os(m)));
    TSL_HIP(hipStreamSynchronize(os(m)));
    return TSL_OK;
}
static int octo_read_int(tsl_octo* m, const int* dev, int* out)
{
    TSL_HIP(hipMemcpyAsync(m->h_stats, dev, sizeof(int), hipMemcpyDeviceToHost, os(m)));
    TSL_HIP(hipStreamSynchronize(os(m)));
    *out = *reinterpret_cast<int*>(m->h_stats);
    return TSL_OK;
}
static int octo_used(tsl_octo* m, int* n)
{ int v = 0; int rc = octo_read_int(m, m->M.pool_top, &v); if (rc) return rc; *n = v > m->M.max_bricks ? m->M.max_bricks : v; return TSL_OK; }
static int octo_check_err(tsl_octo* m)
{
    int e = 0; int rc = octo_read_int(m, m->M.err, &e); if (rc) return rc;
    if (e) { (void)hipMemsetAsync(m->M.err, 0, sizeof(int), os(m)); set_error("octomap brick pool exhausted (max_bricks)"); return TSL_ERR_CAPACITY; }
    return TSL_OK;
}

static int octo_stage_tex(tsl_octo* m, const uint8_t* tex, size_t bytes)
{
    if (m->stage_tex_bytes < bytes) { if (m->stage_tex) (void)hipFree(m->stage_tex); m->stage_tex = nullptr; TSL_HIP(hipMalloc(&m->stage_tex, bytes + 4096)); m->stage_tex_bytes = bytes + 4096; }
    TSL_HIP(hipMemcpyAsync(m->stage_tex, tex, bytes, hipMemcpyHostToDevice, os(m)));
    return TSL_OK;
}
static int octo_leaf_scratch(tsl_octo* m, size_t n)
{
    if (m->leaf_of_n < n) { if (m->leaf_of) (void)hipFree(m->leaf_of); m->leaf_of = nullptr; TSL_HIP(hipMalloc((void**)&m->leaf_of, sizeof(long long) * (n + 1024))); m->leaf_of_n = n + 1024; }
    return TSL_OK;
}

#define OCTO_STAT_RING 64
extern "C" {

int tsl_octo_create(const tsl_octo_cfg* cfg, int device, tsl_octo** out)
{
    TSL_REQUIRE(cfg && out, "tsl_octo_create: null argument");
    TSL_REQUIRE(cfg->voxel_scale > 0 && cfg->K >= 2 && cfg->recast_step >= 1, "tsl_octo_create: bad config");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available"); return TSL_ERR_NO_DEVICE; }
    TSL_REQUIRE(device >= 0 && device < ndev, "tsl_octo_create: bad device index");
    TSL_HIP(hipSetDevice(device));
    tsl_octo* m = new tsl_octo();
    m->cfg = *cfg; m->device = device;
    TSL_HIP(hipStreamCreateWithFlags(&m->stream_, hipStreamNonBlocking));
    m->q = new OctoBatchArgs(); m->q->n = 0; m->qmax = 0; m->pin_idx = 0; m->qev_idx = 0; m->next_packed_slot = -1;
    for (int k = 0; k < OCTO_PIN_SLOTS; ++k) { m->pin[k] = m->pin_dev[k] = nullptr; m->pin_bytes[k] = 0; m->pin_guard[k] = -1; }
    for (int k = 0; k < 4; ++k) TSL_HIP(hipEventCreateWithFlags(&m->qev[k], hipEventDisableTiming));
    m->P.packed = 0;
    m->K = cfg->K;
    m->Rxy = (int)std::ceil(std::log2(cfg->map_size_xy / cfg->voxel_scale) / std::log2((double)cfg->K));     // taichi_octomap.py:19
    m->Rz = (int)std::ceil(std::log2(cfg->map_size_z / cfg->voxel_scale) / std::log2((double)cfg->K));      // :20
    TSL_REQUIRE(m->Rxy >= 1 && m->Rz >= 1 && m->Rxy <= 20, "tsl_octo_create: bad tree depth");
    m->N = octo_ipow(m->K, m->Rxy); m->Nz = octo_ipow(m->K, m->Rz);                                          // :26-27
    m->voxel_scale_recomputed = cfg->map_size_xy / (double)m->N;                                             // :28 (Q15)
    OctoDev& M = m->M; std::memset(&M, 0, sizeof(M));
    M.N = m->N; M.Nz = m->Nz; M.hN = m->N / 2; M.hNz = m->Nz / 2;
    M.ext_xy = octo_ipow(m->K, m->Rxy + 1);                                                                  // tree extent, :65-70 (Q16)
    M.ext_z = octo_ipow(m->K, 1 + (m->Rz < m->Rxy ? m->Rz : m->Rxy));
    M.nbx = (M.ext_xy + 15) / 16; M.nbz = (M.ext_z + 15) / 16;
    const int64_t nb3 = (int64_t)M.nbx * M.nbx * M.nbz;
    TSL_REQUIRE(nb3 < (1ll << 30), "tsl_octo_create: map too large");
    M.nb3 = (int)nb3;
    m->nsub = cfg->max_submap_num > 0 ? cfg->max_submap_num : 1;
    M.max_bricks = cfg->max_bricks > 0 ? cfg->max_bricks : 65536;
    TSL_HIP(hipMalloc((void**)&M.tables, sizeof(int*) * (size_t)m->nsub));
    TSL_HIP(hipMemsetAsync(M.tables, 0, sizeof(int*) * (size_t)m->nsub, os(m)));
    m->tables.assign((size_t)m->nsub, nullptr);
    TSL_HIP(hipMalloc((void**)&M.cnt, sizeof(float) * (size_t)M.max_bricks * TSL_BRK3));
    TSL_HIP(hipMemsetAsync(M.cnt, 0, sizeof(float) * (size_t)M.max_bricks * TSL_BRK3, os(m)));
    M.col = nullptr; M.win = nullptr;
    if (cfg->texture_enabled) {
        TSL_REQUIRE(M.ext_xy <= (1 << 18) && m->nsub <= 1024, "tsl_octo_create: textured maps are limited to 2^18 cells per axis and 1024 submaps");
        TSL_HIP(hipMalloc((void**)&M.col, sizeof(float) * 3 * (size_t)M.max_bricks * TSL_BRK3));
        TSL_HIP(hipMemsetAsync(M.col, 0, sizeof(float) * 3 * (size_t)M.max_bricks * TSL_BRK3, os(m)));
        TSL_HIP(hipMalloc((void**)&M.win, sizeof(unsigned long long) * (size_t)M.max_bricks * TSL_BRK3));
        TSL_HIP(hipMemsetAsync(M.win, 0, sizeof(unsigned long long) * (size_t)M.max_bricks * TSL_BRK3, os(m)));
    }
    TSL_HIP(hipMalloc((void**)&M.owner_s, sizeof(int) * (size_t)M.max_bricks));
    TSL_HIP(hipMalloc((void**)&M.owner_b, sizeof(int) * (size_t)M.max_bricks));
    TSL_HIP(hipMalloc((void**)&M.pool_top, sizeof(int) * 4));
    TSL_HIP(hipMemsetAsync(M.pool_top, 0, sizeof(int) * 4, os(m)));
    M.err = M.pool_top + 1;
    OctoParams& P = m->P; std::memset(&P, 0, sizeof(P));
    for (int i = 0; i < 3; ++i) P.R[i * 4] = 1.0f;
    P.vs = (float)cfg->voxel_scale;                                                                          // voxel_scale_ cached from the ctor argument (Q15)
    P.thr_max = (float)(cfg->max_ray_length * 1000.0); P.thr_min = (float)(cfg->min_ray_length * 1000.0);
    P.step = cfg->recast_step; P.same_proj = cfg->color_same_proj;
    m->occ_thres = (float)cfg->min_occupy_thres;
    m->baseR.assign((size_t)m->nsub * 9, 0.0); m->baseT.assign((size_t)m->nsub * 3, 0.0);
    m->baseRf.assign((size_t)m->nsub * 9, 0.0f); m->baseTf.assign((size_t)m->nsub * 3, 0.0f);
    for (int s = 0; s < m->nsub; ++s) for (int i = 0; i < 3; ++i) { m->baseR[(size_t)s * 9 + i * 4] = 1.0; m->baseRf[(size_t)s * 9 + i * 4] = 1.0f; }   // identity default
    m->active = 0; m->p_used = 0;
    TSL_HIP(hipMalloc((void**)&m->stats_ring, sizeof(tsl_frame_stats) * OCTO_STAT_RING));
    TSL_HIP(hipMemsetAsync(m->stats_ring, 0, sizeof(tsl_frame_stats) * OCTO_STAT_RING, os(m)));
    m->stats = m->stats_ring; m->stat_idx = 0;
    TSL_HIP(hipHostMalloc((void**)&m->h_stats, sizeof(tsl_frame_stats), hipHostMallocDefault));
    m->max_disp = cfg->max_disp_particles > 0 ? cfg->max_disp_particles : 1000000;
    TSL_HIP(hipMalloc((void**)&m->exp_xyz, sizeof(float) * 3 * (size_t)m->max_disp));
    TSL_HIP(hipMalloc((void**)&m->exp_rgb, sizeof(float) * 3 * (size_t)m->max_disp));
    TSL_HIP(hipMalloc((void**)&m->num_particles, sizeof(int) * 4));
    TSL_HIP(hipMemsetAsync(m->num_particles, 0, sizeof(int) * 4, os(m)));
    TSL_HIP(hipMalloc((void**)&m->pose_dev, sizeof(float) * 12 * (size_t)m->nsub));
    m->stage = nullptr; m->stage_bytes = 0; m->xbuf = nullptr; m->xbuf_bytes = 0;
    m->stage_tex = nullptr; m->stage_tex_bytes = 0; m->leaf_of = nullptr; m->leaf_of_n = 0;
    int rc = octo_ensure_table(m, 0); if (rc) return rc;
    TSL_HIP(hipStreamSynchronize(os(m)));
    *out = m;
    return TSL_OK;
}

void tsl_octo_destroy(tsl_octo* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device); (void)hipStreamSynchronize(os(m));
    for (int* t : m->tables) if (t) (void)hipFree(t);
    void* ptrs[] = { m->M.col, m->M.win, m->stage_tex, m->leaf_of, m->M.tables, m->M.cnt, m->M.owner_s, m->M.owner_b, m->M.pool_top, m->stats_ring, m->exp_xyz, m->exp_rgb, m->num_particles, m->pose_dev, m->stage, m->xbuf };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (m->h_stats) (void)hipHostFree(m->h_stats);
    for (int k = 0; k < OCTO_PIN_SLOTS; ++k) if (m->pin[k]) (void)hipHostFree(m->pin[k]);
    for (int k = 0; k < 4; ++k) if (m->qev[k]) (void)hipEventDestroy(m->qev[k]);
    (void)hipStreamDestroy(m->stream_); delete m->q; m->q = nullptr;
    delete m;
}

int tsl_octo_get_dims(const tsl_octo* m, int32_t* N, int32_t* Nz, int32_t* Rxy, int32_t* Rz, double* vs)
{ TSL_REQUIRE(m, "null handle"); if (N) *N = m->N; if (Nz) *Nz = m->Nz; if (Rxy) *Rxy = m->Rxy; if (Rz) *Rz = m->Rz; if (vs) *vs = m->voxel_scale_recomputed; return TSL_OK; }
int tsl_octo_sync(tsl_octo* m) { TSL_REQUIRE(m, "null handle"); TSL_HIP(hipSetDevice(m->device)); TSL_HIP(hipStreamSynchronize(os(m))); return TSL_OK; }

int tsl_octo_reset(tsl_octo* m)                                                              // taichi_octomap.py:210-211
{
    TSL_REQUIRE(m, "null handle"); TSL_HIP(hipSetDevice(m->device));
    int used = 0; int rc = octo_used(m, &used); if (rc) return rc;
    if (used > 0) hipLaunchKernelGGL(k_octo_reset, dim3(used < 4096 ? used : 4096), dim3(256), 0, os(m), m->M, used);
    TSL_HIP(hipMemsetAsync(m->M.pool_top, 0, sizeof(int) * 2, os(m)));
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
int tsl_octo_set_intrinsics(tsl_octo* m, const double Kd[9], const double Kc[9])
{
    TSL_REQUIRE(m, "null handle");
    if (Kd) { m->P.fx = (float)Kd[0]; m->P.fy = (float)Kd[4]; m->P.cx = (float)Kd[2]; m->P.cy = (float)Kd[5]; }
    if (Kc) { m->P.fxc = (float)Kc[0]; m->P.fyc = (float)Kc[4]; m->P.cxc = (float)Kc[2]; m->P.cyc = (float)Kc[5]; }      // mapping_common.py:25-29
    return TSL_OK;
}
int tsl_octo_set_base_pose_submap(tsl_octo* m, int sid, const double R[9], const double T[3])
{
    TSL_REQUIRE(m && R && T, "null"); TSL_REQUIRE(sid >= 0 && sid < m->nsub, "set_base_pose_submap: submap id out of range");
    std::memcpy(&m->baseR[(size_t)sid * 9], R, 72); std::memcpy(&m->baseT[(size_t)sid * 3], T, 24);
    for (int a = 0; a < 9; ++a) m->baseRf[(size_t)sid * 9 + a] = (float)R[a];
    for (int a = 0; a < 3; ++a) m->baseTf[(size_t)sid * 3 + a] = (float)T[a];
    return TSL_OK;
}
int tsl_octo_get_active_submap(const tsl_octo* m, int32_t* sid) { TSL_REQUIRE(m && sid, "null"); *sid = m->active; return TSL_OK; }
int tsl_octo_set_active_submap(tsl_octo* m, int32_t sid)
{
    TSL_REQUIRE(m, "null"); TSL_REQUIRE(sid >= 0 && sid < m->nsub, "set_active_submap: id out of range");
    TSL_HIP(hipSetDevice(m->device));
    m->active = sid;
    return octo_ensure_table(m, sid);
}

// the next frame's statistics slot: zero already (half of the ring is cleared whenever the index enters it, so the slot of the frame
// before -- the one tsl_octo_last_frame_stats reads -- is never cleared under it)
static int octo_next_stats(tsl_octo* m)
{
    m->stat_idx = (m->stat_idx + 1) % OCTO_STAT_RING;
    const int half = OCTO_STAT_RING / 2;
    if (m->stat_idx % half == 0) TSL_HIP(hipMemsetAsync(m->stats_ring + m->stat_idx, 0, sizeof(tsl_frame_stats) * half, os(m)));
    m->stats = m->stats_ring + m->stat_idx;
    return TSL_OK;
}
static void octo_fill_pose(tsl_octo* m, const double R[9], const double T[3])
{ convert_pose(&m->baseR[(size_t)m->active * 9], &m->baseT[(size_t)m->active * 3], R, T, m->P.R, m->P.T); }

int tsl_octo_integrate_depth_dev(tsl_octo* m, const double R[9], const double T[3], const void* depth_dev, int h, int w, const void* tex_dev, int th, int tw)
{
    TSL_REQUIRE(m && R && T && depth_dev, "octo integrate_depth: null argument"); TSL_REQUIRE(h > 0 && w > 0, "octo integrate_depth: bad image size");
    TSL_HIP(hipSetDevice(m->device));
    octo_fill_pose(m, R, T);
    OctoParams& P = m->P;
    P.W = w; P.hh = (int)((float)h / (float)P.step); P.ww = (int)((float)w / (float)P.step);   // taichi_octomap.py:151,153
    const int total = P.hh * P.ww;
    m->p_used = total;
    const bool tex = m->M.col && tex_dev && th > 0 && tw > 0;
    if (tex) {
        TSL_REQUIRE(!P.same_proj || (th >= h && tw >= w), "octo integrate_depth: texture smaller than the depth image");
        P.th = th; P.tw = tw;
        int rc = octo_leaf_scratch(m, (size_t)total); if (rc) return rc;
    }
    if (!tex && m->q) {
        // only QUEUED: up to OCTO_NB frames are inserted by one launch, issued when the queue is full or as soon as anything else wants the map or the stream
        // (the slot's statistics words are cleared by a memset ordered in front of that launch: the ring's half is entered here, the launch comes later)
        m->stat_idx = (m->stat_idx + 1) % OCTO_STAT_RING;
        if (m->stat_idx % (OCTO_STAT_RING / 2) == 0) { octo_flush(m); TSL_HIP(hipMemsetAsync(m->stats_ring + m->stat_idx, 0, sizeof(tsl_frame_stats) * (OCTO_STAT_RING / 2), m->stream_)); }
        m->stats = m->stats_ring + m->stat_idx;
        if (total > 0) {
            OctoBatchArgs& Q = *m->q;
            Q.P[Q.n] = P; Q.P[Q.n].packed = m->next_packed_slot >= 0 ? 1 : 0; Q.depth[Q.n] = (const uint16_t*)depth_dev; Q.st[Q.n] = m->stats; Q.s[Q.n] = m->active;
            m->qslot[Q.n] = m->next_packed_slot; m->next_packed_slot = -1; ++Q.n;
            { const int tiles = ((P.ww + 15) >> 4) * ((P.hh + 15) >> 4); if (tiles > m->qmax) m->qmax = tiles; }          // workgroups of the largest frame
            if (Q.n == OCTO_NB) octo_flush(m);
        }
        return TSL_OK;
    }
    { const int rc = octo_next_stats(m); if (rc) return rc; }
    if (total > 0) {
        hipLaunchKernelGGL(k_octo_depth, dim3((total + 255) / 256), dim3(256), 0, os(m), m->M, P, m->active, (const uint16_t*)depth_dev, m->stats, tex ? m->leaf_of : nullptr);
        if (tex) hipLaunchKernelGGL(k_octo_colour, dim3((total + 255) / 256), dim3(256), 0, os(m), m->M, P, (const uint8_t*)tex_dev, (const long long*)m->leaf_of, total, 0);
    }
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
int tsl_octo_integrate_depth(tsl_octo* m, const double R[9], const double T[3], const uint16_t* depth, int h, int w, const uint8_t* tex, int th, int tw)
{
    TSL_REQUIRE(m && depth, "octo integrate_depth: null argument"); TSL_REQUIRE(h > 0 && w > 0, "octo integrate_depth: bad image size");
    TSL_HIP(hipSetDevice(m->device));
    const bool use_tex = m->M.col && tex && th > 0 && tw > 0;
    if (!use_tex && m->q) {
        // Round 6: no copy call and no synchronisation per frame.  The host copies the VISITED pixels (every recast_step-th of every recast_step-th row: a quarter
        // of the image at step 2) into a pinned, device-mapped slot of a ring and the frame is queued like a device-resident one; the insert kernel reads the slot
        // across the host link.  The caller may reuse its image on return; the slot is written again sixteen frames later, behind the batch that read it.
        // (17 k -> 25 k+ frames/s from numpy images, taichislam_node.py:381-382's calling convention.)
        const int step = m->P.step, hh = (int)((float)h / (float)step), ww = (int)((float)w / (float)step);
        const size_t need = (size_t)hh * ww * sizeof(uint16_t) + 64;
        const int k = m->pin_idx; m->pin_idx = (m->pin_idx + 1) % OCTO_PIN_SLOTS;
        if (m->pin_guard[k] >= 0) { TSL_HIP(hipEventSynchronize(m->qev[m->pin_guard[k]])); m->pin_guard[k] = -1; }
        for (int j = 0; j < m->q->n; ++j) if (m->qslot[j] == k) { octo_flush(m); TSL_HIP(hipStreamSynchronize(m->stream_)); break; }      // (cannot happen with 16 slots and batches of 8)
        if (m->pin_bytes[k] < need) {
            if (m->pin[k]) (void)hipHostFree(m->pin[k]);
            m->pin[k] = nullptr; m->pin_bytes[k] = 0;
            TSL_HIP(hipHostMalloc(&m->pin[k], need + need / 4, hipHostMallocMapped | hipHostMallocCoherent));
            m->pin_bytes[k] = need + need / 4;
            TSL_HIP(hipHostGetDevicePointer(&m->pin_dev[k], m->pin[k], 0));
        }
        uint16_t* dst = static_cast<uint16_t*>(m->pin[k]);
        for (int jj = 0; jj < hh; ++jj) {
            const uint16_t* src = depth + (size_t)jj * step * w;
            uint16_t* d = dst + (size_t)jj * ww;
            if (step == 1) std::memcpy(d, src, (size_t)ww * 2);
            else if (step == 2) { for (int i = 0; i < ww; ++i) d[i] = src[2 * i]; }
            else for (int i = 0; i < ww; ++i) d[i] = src[(size_t)i * step];
        }
        m->next_packed_slot = k;
        const int rc = tsl_octo_integrate_depth_dev(m, R, T, m->pin_dev[k], h, w, nullptr, 0, 0);
        m->next_packed_slot = -1;                               // (a frame without a visited pixel is not queued at all)
        return rc;
    }
    const size_t nb = (size_t)h * w * 2;
    if (m->stage_bytes < nb) { if (m->stage) (void)hipFree(m->stage); m->stage = nullptr; TSL_HIP(hipMalloc(&m->stage, nb + 4096)); m->stage_bytes = nb + 4096; }
    TSL_HIP(hipMemcpyAsync(m->stage, depth, nb, hipMemcpyHostToDevice, os(m)));
    if (use_tex) { int rc = octo_stage_tex(m, tex, (size_t)th * tw * 3); if (rc) return rc; }
    TSL_HIP(hipStreamSynchronize(os(m)));
    const int rc = tsl_octo_integrate_depth_dev(m, R, T, m->stage, h, w, use_tex ? m->stage_tex : nullptr, th, tw);
    octo_flush(m);                                          // (the staging buffer is written again by the next call)
    return rc;
}
/* recast_pcl_to_map with DEVICE buffers: xyz f32 [n][3], rgb u8 [n][3] or NULL (taichi_octomap.py:126-128,134-145).  Enqueued on the
 * handle's stream; the buffers must be complete (the caller's producing stream synchronised) and stay unchanged until tsl_octo_sync. */
int tsl_octo_integrate_points_dev(tsl_octo* m, const double R[9], const double T[3], const void* xyz_dev, const void* rgb_dev, int64_t n)
{
    TSL_REQUIRE(m && R && T, "octo integrate_points: null argument"); TSL_REQUIRE(n >= 0 && (n == 0 || xyz_dev) && n < (1ll << 31), "octo integrate_points: bad input");
    TSL_HIP(hipSetDevice(m->device));
    octo_fill_pose(m, R, T);
    m->p_used = n;
    const bool use_tex = m->M.col && rgb_dev && n > 0;
    { const int rc = octo_next_stats(m); if (rc) return rc; }
    if (n == 0) return TSL_OK;
    if (use_tex) { int rc = octo_leaf_scratch(m, (size_t)n); if (rc) return rc; }
    hipLaunchKernelGGL(k_octo_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, os(m), m->M, m->P, m->active, (const float*)xyz_dev, (int)n, m->stats, use_tex ? m->leaf_of : nullptr);
    if (use_tex) hipLaunchKernelGGL(k_octo_colour, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, os(m), m->M, m->P, (const uint8_t*)rgb_dev, (const long long*)m->leaf_of, (int)n, 1);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
int tsl_octo_integrate_points(tsl_octo* m, const double R[9], const double T[3], const float* xyz, const uint8_t* rgb, int64_t n)
{
    TSL_REQUIRE(m && R && T, "octo integrate_points: null argument"); TSL_REQUIRE(n >= 0 && (n == 0 || xyz) && n < (1ll << 31), "octo integrate_points: bad input");
    TSL_HIP(hipSetDevice(m->device));
    const bool use_tex = m->M.col && rgb && n > 0;
    if (n > 0) {
        const size_t nb = (size_t)n * 12;
        if (m->stage_bytes < nb) { if (m->stage) (void)hipFree(m->stage); m->stage = nullptr; TSL_HIP(hipMalloc(&m->stage, nb + 4096)); m->stage_bytes = nb + 4096; }
        TSL_HIP(hipMemcpyAsync(m->stage, xyz, nb, hipMemcpyHostToDevice, os(m)));
        if (use_tex) { int rc = octo_stage_tex(m, rgb, (size_t)n * 3); if (rc) return rc; }
        TSL_HIP(hipStreamSynchronize(os(m)));
    }
    return tsl_octo_integrate_points_dev(m, R, T, m->stage, use_tex ? m->stage_tex : nullptr, n);
}
int tsl_octo_last_frame_stats(tsl_octo* m, tsl_frame_stats* out)
{
    TSL_REQUIRE(m && out, "null"); TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipMemcpyAsync(m->h_stats, m->stats, sizeof(tsl_frame_stats), hipMemcpyDeviceToHost, os(m)));
    TSL_HIP(hipStreamSynchronize(os(m)));
    *out = *m->h_stats; out->p_used = m->p_used;
    return octo_check_err(m);
}

static int octo_export(tsl_octo* m, tsl_octo* dst, int mode, int level, int keep, int32_t* idx, float* cnt, float* rgb, int64_t cap, int64_t* n)
{
    TSL_HIP(hipSetDevice(m->device));
    int used = 0; int rc = octo_used(m, &used); if (rc) return rc;
    int gxy = 1, gz = 1;
    for (int up = 0; up < level - 1; ++up) { const int r = m->Rxy - 1 - up; if (r < 0) break; gxy *= m->K; if (r < m->Rz) gz *= m->K; }   // granularity of occupy.parent(level)
    OctoPose B;
    for (int a = 0; a < 9; ++a) B.R[a] = m->baseRf[(size_t)m->active * 9 + a];
    for (int a = 0; a < 3; ++a) B.T[a] = m->baseTf[(size_t)m->active * 3 + a];
    int* counter; int32_t* didx = nullptr; float* dcnt = nullptr; float* dxyz = nullptr; float* drgb = nullptr; long long dcap;
    if (mode == 0) {
        const size_t need = (size_t)cap * 28 + 64;
        if (m->xbuf_bytes < need) { if (m->xbuf) (void)hipFree(m->xbuf); m->xbuf = nullptr; TSL_HIP(hipMalloc(&m->xbuf, need + 4096)); m->xbuf_bytes = need + 4096; }
        didx = (int32_t*)m->xbuf; dcnt = (float*)((char*)m->xbuf + (size_t)cap * 12); drgb = (float*)((char*)m->xbuf + (size_t)cap * 16);
        counter = m->num_particles + 2; dcap = cap;
        TSL_HIP(hipMemsetAsync(counter, 0, sizeof(int), os(m)));
    } else {
        if (!dst) dst = m;
        counter = dst->num_particles; dxyz = dst->exp_xyz; drgb = dst->exp_rgb; dcap = dst->max_disp;
        if (!keep) TSL_HIP(hipMemsetAsync(counter, 0, sizeof(int), os(m)));                // :93
    }
    if (used > 0) hipLaunchKernelGGL(k_octo_export, dim3(used < 8192 ? used : 8192), dim3(256), 0, os(m), m->M, m->active, used, mode, m->occ_thres, gxy, gz, B, m->P.vs,
                                     didx, dcnt, dxyz, drgb, dcap, counter);
    int c = 0; rc = octo_read_int(m, counter, &c); if (rc) return rc;
    *n = c;
    if (mode == 0) {
        const size_t k = (size_t)(c < cap ? c : cap);
        if (k && idx) TSL_HIP(hipMemcpy(idx, didx, k * 12, hipMemcpyDeviceToHost));
        if (k && cnt) TSL_HIP(hipMemcpy(cnt, dcnt, k * 4, hipMemcpyDeviceToHost));
        if (k && rgb) { if (m->M.col) TSL_HIP(hipMemcpy(rgb, drgb, k * 12, hipMemcpyDeviceToHost)); else std::memset(rgb, 0, k * 12); }
    }
    return TSL_OK;
}
int tsl_octo_export_leaves(tsl_octo* m, int32_t* idx, float* cnt, float* rgb, int64_t cap, int64_t* n)
{ TSL_REQUIRE(m && n && cap >= 0, "octo export_leaves: bad argument"); return octo_export(m, nullptr, 0, 0, 0, idx, cnt, rgb, cap, n); }
int tsl_octo_occupied_voxels(tsl_octo* m, tsl_octo* dst, int level, int add_to_cur, int32_t* n)
{
    TSL_REQUIRE(m, "null handle"); TSL_REQUIRE(level >= 0, "bad level");
    int64_t c = 0; int rc = octo_export(m, dst, 1, level, add_to_cur, nullptr, nullptr, nullptr, 0, &c);
    if (n) *n = (int32_t)c;
    return rc;
}
int tsl_octo_read_exports(tsl_octo* m, float* xyz, float* rgb, int64_t n)
{
    TSL_REQUIRE(m, "null handle"); TSL_REQUIRE(n >= 0 && n <= m->max_disp, "read_exports: n out of range"); TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipStreamSynchronize(os(m)));
    if (n && xyz) TSL_HIP(hipMemcpy(xyz, m->exp_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    if (n && rgb) TSL_HIP(hipMemcpy(rgb, m->exp_rgb, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    return TSL_OK;
}
/* export_x / export_color as DEVICE pointers (f32 [max_disp_particles][3], valid for the lifetime of the handle) + num_export_particles */
int tsl_octo_exports_dev(tsl_octo* m, void** xyz_dev, void** rgb_dev, int32_t* n)
{
    TSL_REQUIRE(m, "null handle"); TSL_HIP(hipSetDevice(m->device));
    if (xyz_dev) *xyz_dev = m->exp_xyz; if (rgb_dev) *rgb_dev = m->exp_rgb;
    int v = 0; const int rc = octo_read_int(m, m->num_particles, &v); if (rc) return rc;
    if (n) *n = v;
    return TSL_OK;
}
/* the first n rows of export_x [+ export_color] as the data block of a sensor_msgs/PointCloud2 (interleaved f32 x y z [r g b], point_step
 * 12 / 24): what scripts/taichislam_node.py:330-333 + utils/ros_pcl_transfer.py:96-136 assemble from the numpy copies, interleaved on
 * the device */
__global__ void __launch_bounds__(256) k_octo_pack_pointcloud2(const float* __restrict__ xyz, const float* __restrict__ rgb, float* __restrict__ out, long long n, int stride)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n * stride) return;
    const long long row = q / stride; const int c = (int)(q - row * stride);
    out[q] = c < 3 ? xyz[row * 3 + c] : rgb[row * 3 + c - 3];
}
int tsl_octo_pack_pointcloud2(tsl_octo* m, int has_rgb, int64_t n, void* out_host)
{
    TSL_REQUIRE(m && (n == 0 || out_host), "octo pack_pointcloud2: null argument"); TSL_REQUIRE(n >= 0 && n <= m->max_disp, "octo pack_pointcloud2: n out of range");
    TSL_HIP(hipSetDevice(m->device));
    if (n == 0) return TSL_OK;
    const int stride = has_rgb ? 6 : 3;
    const size_t need = sizeof(float) * (size_t)stride * (size_t)n;
    if (m->xbuf_bytes < need) { if (m->xbuf) (void)hipFree(m->xbuf); m->xbuf = nullptr; TSL_HIP(hipMalloc(&m->xbuf, need + 4096)); m->xbuf_bytes = need + 4096; }
    hipLaunchKernelGGL(k_octo_pack_pointcloud2, dim3((unsigned)(((long long)n * stride + 255) / 256)), dim3(256), 0, os(m), m->exp_xyz, m->exp_rgb, (float*)m->xbuf, (long long)n, stride);
    TSL_HIP(hipMemcpyAsync(out_host, m->xbuf, need, hipMemcpyDeviceToHost, os(m)));
    TSL_HIP(hipStreamSynchronize(os(m)));
    return TSL_OK;
}
int tsl_octo_num_particles(tsl_octo* m, int32_t* n) { TSL_REQUIRE(m && n, "null"); TSL_HIP(hipSetDevice(m->device)); int v = 0; int rc = octo_read_int(m, m->num_particles, &v); *n = v; return rc; }

int tsl_octo_fuse_submaps(tsl_octo* g, tsl_octo* sub)
{
    TSL_REQUIRE(g && sub, "octo fuse_submaps: null handle"); TSL_REQUIRE(g->device == sub->device, "octo fuse_submaps: maps live on different devices");
    TSL_HIP(hipSetDevice(g->device));
    int rc = tsl_octo_sync(sub); if (rc) return rc;
    if ((rc = tsl_octo_reset(g))) return rc;                                                   // :196
    if ((rc = octo_ensure_table(g, 0))) return rc;
    for (int s = 0; s < sub->active && s < g->nsub; ++s) {                                     // :174-178
        for (int a = 0; a < 9; ++a) g->baseRf[(size_t)s * 9 + a] = (float)g->baseR[(size_t)s * 9 + a];
        for (int a = 0; a < 3; ++a) g->baseTf[(size_t)s * 3 + a] = (float)g->baseT[(size_t)s * 3 + a];
    }
    std::vector<float> tab((size_t)g->nsub * 12);
    for (int s = 0; s < g->nsub; ++s) { for (int a = 0; a < 9; ++a) tab[(size_t)s * 12 + a] = g->baseRf[(size_t)s * 9 + a]; for (int a = 0; a < 3; ++a) tab[(size_t)s * 12 + 9 + a] = g->baseTf[(size_t)s * 3 + a]; }
    TSL_HIP(hipMemcpyAsync(g->pose_dev, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, os(g)));
    TSL_HIP(hipStreamSynchronize(os(g)));
    int nsrc = 0; if ((rc = octo_used(sub, &nsrc))) return rc;
    if (nsrc > 0) {
        hipLaunchKernelGGL(k_octo_fuse, dim3(nsrc < 8192 ? nsrc : 8192), dim3(256), 0, os(g), sub->M, g->M, nsrc, g->pose_dev, g->nsub, g->P.vs, g->occ_thres, 0);
        if (sub->M.col && g->M.col) hipLaunchKernelGGL(k_octo_fuse, dim3(nsrc < 8192 ? nsrc : 8192), dim3(256), 0, os(g), sub->M, g->M, nsrc, g->pose_dev, g->nsub, g->P.vs, g->occ_thres, 1);
    }
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipStreamSynchronize(os(g)));
    return octo_check_err(g);
}

}  // extern "C"
