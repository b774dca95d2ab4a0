/*
 * tsl_oracle_octo.c -- CPU ORACLE, Octomap hit counter (test infrastructure only).
 * Restates taichi_slam/mapping/taichi_octomap.py; parity pinned to the reference's source run on tools/ti_seq (tests/golden/ref_octomap.npz; see tsl_oracle.h).
 */
#include "tsl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float rnd_f(float x)
{
    float r = truncf(x);
    float d = fabsf(x - r);
    if (d >= 0.5f) r += copysignf(1.0f, x);
    return r;
}
static inline int rnd_i(float x) { return (int)rnd_f(x); }

typedef struct { int32_t c[3]; float cnt; float col[3]; int used; int32_t wkey[4]; } leaf_t;      /* wkey: source of col after a fusion (submap, i, j, k) */
typedef struct { leaf_t* a; int nslots, n; } leafmap;

struct ora_octo {
    ora_octo_cfg cfg;
    int Rxy, Rz, N, Nz, K;
    int ext_xy, ext_z;                 /* tree extent in cells: K^(Rxy+1), K^(1+min(Rxy,Rz))  taichi_octomap.py:65-70 (Q16) */
    double voxel_scale_recomputed;     /* :28 (Q15) */
    float vs;                          /* voxel_scale_ cached from the ctor argument  mapping_common.py:22-23 */
    float fx, fy, cx, cy, fxc, fyc, cxc, cyc;
    float thr_max, thr_min, occ_thres;
    int nsub, active;
    leafmap* sub;
    double* baseR; double* baseT; float* baseRf; float* baseTf;
    float inR[9], inT[3];
};

static int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }

ora_octo* ora_octo_create(const ora_octo_cfg* cfg)
{
    ora_octo* m = (ora_octo*)calloc(1, sizeof(*m));
    m->cfg = *cfg; m->K = cfg->K;
    m->Rxy = (int)ceil(log2(cfg->map_size_xy / cfg->voxel_scale) / log2((double)cfg->K));   /* :19 */
    m->Rz = (int)ceil(log2(cfg->map_size_z / cfg->voxel_scale) / log2((double)cfg->K));     /* :20 */
    m->N = ipow(m->K, m->Rxy); m->Nz = ipow(m->K, m->Rz);                                     /* :26-27 */
    m->ext_xy = ipow(m->K, m->Rxy + 1);
    m->ext_z = ipow(m->K, 1 + (m->Rz < m->Rxy ? m->Rz : m->Rxy));
    m->voxel_scale_recomputed = cfg->map_size_xy / (double)m->N;
    m->vs = (float)cfg->voxel_scale;
    m->thr_max = (float)(cfg->max_ray_length * 1000.0); m->thr_min = (float)(cfg->min_ray_length * 1000.0);
    m->occ_thres = (float)cfg->min_occupy_thres;
    m->nsub = cfg->max_submap_num > 0 ? cfg->max_submap_num : 1;                              /* :65 (global maps keep the axis too) */
    m->sub = (leafmap*)calloc((size_t)m->nsub, sizeof(leafmap));
    m->baseR = (double*)calloc((size_t)m->nsub * 9, sizeof(double)); m->baseT = (double*)calloc((size_t)m->nsub * 3, sizeof(double));
    m->baseRf = (float*)calloc((size_t)m->nsub * 9, sizeof(float)); m->baseTf = (float*)calloc((size_t)m->nsub * 3, sizeof(float));
    for (int s = 0; s < m->nsub; ++s) for (int i = 0; i < 3; ++i) { m->baseR[s * 9 + i * 4] = 1.0; m->baseRf[s * 9 + i * 4] = 1.0f; }   /* identity default */
    for (int i = 0; i < 3; ++i) m->inR[i * 4] = 1.0f;
    return m;
}
void ora_octo_reset(ora_octo* m)       /* :210-211 */
{ for (int s = 0; s < m->nsub; ++s) { free(m->sub[s].a); memset(&m->sub[s], 0, sizeof(leafmap)); } }
void ora_octo_destroy(ora_octo* m)
{ if (!m) return; ora_octo_reset(m); free(m->sub); free(m->baseR); free(m->baseT); free(m->baseRf); free(m->baseTf); free(m); }
void ora_octo_get_dims(const ora_octo* m, int* N, int* Nz, int* Rxy, int* Rz, double* vs)
{ if (N) *N = m->N; if (Nz) *Nz = m->Nz; if (Rxy) *Rxy = m->Rxy; if (Rz) *Rz = m->Rz; if (vs) *vs = m->voxel_scale_recomputed; }
void ora_octo_set_intrinsics(ora_octo* m, const double Kd[9], const double Kc[9])
{
    if (Kd) { m->fx = (float)Kd[0]; m->fy = (float)Kd[4]; m->cx = (float)Kd[2]; m->cy = (float)Kd[5]; }
    if (Kc) { m->fxc = (float)Kc[0]; m->fyc = (float)Kc[4]; m->cxc = (float)Kc[2]; m->cyc = (float)Kc[5]; }     /* mapping_common.py:25-29 */
}
void ora_octo_set_base_pose_submap(ora_octo* m, int sid, const double R[9], const double T[3])
{
    memcpy(m->baseR + sid * 9, R, 72); memcpy(m->baseT + sid * 3, T, 24);
    for (int i = 0; i < 9; ++i) m->baseRf[sid * 9 + i] = (float)R[i];
    for (int i = 0; i < 3; ++i) m->baseTf[sid * 3 + i] = (float)T[i];
}
void ora_octo_set_active_submap(ora_octo* m, int sid) { m->active = sid; }

static int in_tree(const ora_octo* m, const int c[3])
{
    int h = m->N / 2, hz = m->Nz / 2;
    return c[0] >= -h && c[0] < m->ext_xy - h && c[1] >= -h && c[1] < m->ext_xy - h && c[2] >= -hz && c[2] < m->ext_z - hz;
}

static leaf_t* leaf_get(leafmap* lm, const int c[3], int create)
{
    if (!lm->a) { if (!create) return NULL; lm->nslots = 1 << 16; lm->a = (leaf_t*)calloc((size_t)lm->nslots, sizeof(leaf_t)); }
    if (create && lm->n * 2 > lm->nslots) {
        leafmap big = { (leaf_t*)calloc((size_t)lm->nslots * 4, sizeof(leaf_t)), lm->nslots * 4, 0 };
        for (int i = 0; i < lm->nslots; ++i) if (lm->a[i].used) { leaf_t* d = leaf_get(&big, lm->a[i].c, 1); *d = lm->a[i]; }
        free(lm->a); *lm = big;
    }
    uint64_t h = ((uint64_t)(uint32_t)c[0] * 0x9E3779B1u) ^ ((uint64_t)(uint32_t)c[1] * 0x85EBCA77u) ^ ((uint64_t)(uint32_t)c[2] * 0xC2B2AE3Du);
    h ^= h >> 31;
    int mask = lm->nslots - 1, s = (int)(h & (uint64_t)mask);
    for (;;) {
        leaf_t* l = &lm->a[s];
        if (!l->used) { if (!create) return NULL; l->used = 1; l->c[0] = c[0]; l->c[1] = c[1]; l->c[2] = c[2]; lm->n++; return l; }
        if (l->c[0] == c[0] && l->c[1] == c[1] && l->c[2] == c[2]) return l;
        s = (s + 1) & mask;
    }
}

static void set_pose(ora_octo* m, const double R[9], const double T[3])      /* mapping_common.py:91-100,149-156 */
{
    const double* Rb = m->baseR + m->active * 9; const double* Tb = m->baseT + m->active * 3;
    double d[3] = { T[0] - Tb[0], T[1] - Tb[1], T[2] - Tb[2] };
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { double acc = 0.0; for (int k = 0; k < 3; ++k) acc += Rb[k * 3 + i] * R[k * 3 + j]; m->inR[i * 3 + j] = (float)acc; }
        double acc = 0.0; for (int k = 0; k < 3; ++k) acc += Rb[k * 3 + i] * d[k]; m->inT[i] = (float)acc;
    }
}

/* process_point :116-124 */
static int octo_point(ora_octo* m, const float pt[3], const uint8_t* rgb)
{
    int c[3]; for (int a = 0; a < 3; ++a) c[a] = rnd_i(pt[a] / m->vs);
    if (!in_tree(m, c)) return 0;
    leaf_t* l = leaf_get(&m->sub[m->active], c, 1);
    l->cnt += 1.0f;
    if (rgb && m->cfg.texture_enabled) { l->col[0] = (float)rgb[2] / 255.0f; l->col[1] = (float)rgb[1] / 255.0f; l->col[2] = (float)rgb[0] / 255.0f; }
    return 1;
}

int ora_octo_integrate_depth(ora_octo* m, const double R[9], const double T[3], const uint16_t* depth, int h, int w,
                             const uint8_t* tex, int th, int tw, ora_frame_stats* st_out)     /* :130-132,147-169 */
{
    ora_frame_stats st; memset(&st, 0, sizeof(st));
    set_pose(m, R, T);
    const int step = m->cfg.recast_step;
    const int hh = (int)((float)h / (float)step), ww = (int)((float)w / (float)step);
    for (int jj = 0; jj < hh; ++jj) for (int ii = 0; ii < ww; ++ii) {
        int j = jj * step, i = ii * step;
        st.p_used++;
        uint16_t d = depth[(size_t)j * w + i];
        if (d == 0 || (float)d > m->thr_max || (float)d < m->thr_min) continue;               /* :155 */
        float dep = (float)d / 1000.0f;                                                        /* :157 */
        float pt[3] = { ((float)i - m->cx) * dep / m->fx, ((float)j - m->cy) * dep / m->fy, dep };
        float pm[3];
        for (int a = 0; a < 3; ++a) pm[a] = ((m->inR[a * 3] * pt[0] + m->inR[a * 3 + 1] * pt[1]) + m->inR[a * 3 + 2] * pt[2]) + m->inT[a];   /* :159 */
        const uint8_t* rgb = NULL;
        if (tex && m->cfg.texture_enabled) {
            if (m->cfg.color_same_proj) rgb = tex + ((size_t)j * tw + i) * 3;                 /* :161 */
            else {                                                                             /* :164-165, mapping_common.py:43-58 (as in the TSDF path) */
                int ci = (int)((((float)i - m->cx) / m->fx) * m->fxc + m->cxc);
                int cj = (int)((((float)j - m->cy) / m->fy) * m->fyc + m->cyc);
                if (ci < 0 || ci >= th || cj < 0 || cj >= tw) { ci = 0; cj = 0; }
                if (cj >= th || ci >= tw) { ci = 0; cj = 0; }
                rgb = tex + ((size_t)cj * tw + ci) * 3;
            }
        }
        if (octo_point(m, pm, rgb)) st.p_valid++; else st.p_oob++;                            /* pixels in raster order: the last one to hit a leaf colours it */
    }
    if (st_out) *st_out = st;
    return 0;
}

int ora_octo_integrate_points(ora_octo* m, const double R[9], const double T[3], const float* xyz, const uint8_t* rgb, int64_t n, ora_frame_stats* st_out)   /* :126-128,134-145 */
{
    ora_frame_stats st; memset(&st, 0, sizeof(st));
    set_pose(m, R, T);
    for (int64_t q = 0; q < n; ++q) {
        st.p_used++;
        const float* pt = xyz + q * 3; float pm[3];
        for (int a = 0; a < 3; ++a) pm[a] = ((m->inR[a * 3] * pt[0] + m->inR[a * 3 + 1] * pt[1]) + m->inR[a * 3 + 2] * pt[2]) + m->inT[a];   /* :141 */
        if (octo_point(m, pm, rgb ? rgb + q * 3 : NULL)) st.p_valid++; else st.p_oob++;
    }
    if (st_out) *st_out = st;
    return 0;
}

static int cmp_leaf(const void* a, const void* b)
{
    const leaf_t* x = (const leaf_t*)a; const leaf_t* y = (const leaf_t*)b;
    for (int d = 0; d < 3; ++d) if (x->c[d] != y->c[d]) return x->c[d] < y->c[d] ? -1 : 1;
    return 0;
}
static leaf_t* sorted_leaves(const leafmap* lm, int* n)
{
    leaf_t* out = (leaf_t*)malloc(sizeof(leaf_t) * (size_t)(lm->n > 0 ? lm->n : 1)); int k = 0;
    for (int i = 0; i < lm->nslots; ++i) if (lm->a && lm->a[i].used) out[k++] = lm->a[i];
    qsort(out, (size_t)k, sizeof(leaf_t), cmp_leaf); *n = k; return out;
}

int64_t ora_octo_export_leaves(const ora_octo* m, int32_t* idx, float* cnt, float* rgb, int64_t cap)
{
    int n; leaf_t* l = sorted_leaves(&m->sub[m->active], &n);
    for (int i = 0; i < n && i < cap; ++i) {
        idx[i * 3] = l[i].c[0]; idx[i * 3 + 1] = l[i].c[1]; idx[i * 3 + 2] = l[i].c[2]; cnt[i] = l[i].cnt;
        if (rgb) memcpy(rgb + (size_t)i * 3, l[i].col, sizeof(l[i].col));
    }
    free(l); return n;
}

/* cvt_occupy_to_voxels(level) :90-102.  occupy.parent(level) for level>=1 is the pointer SNode
 * `level-1` steps above the leaf cells; its active cells are reported with the coordinate of
 * their lowest leaf, and is_occupy() reads that leaf (:97, :86-88). */
int64_t ora_octo_occupied_voxels(const ora_octo* m, int level, float* xyz, float* rgb, int64_t cap)
{
    int gxy = 1, gz = 1;
    for (int up = 0; up < level - 1; ++up) {          /* tree level r = Rxy-1-up splits z iff r < Rz  (:66-70) */
        int r = m->Rxy - 1 - up; if (r < 0) break;
        gxy *= m->K; if (r < m->Rz) gz *= m->K;
    }
    int n; leaf_t* l = sorted_leaves(&m->sub[m->active], &n);
    int64_t cnt = 0;
    const float* R = m->baseRf + m->active * 9; const float* T = m->baseTf + m->active * 3;
    for (int i = 0; i < n; ++i) {
        int u0 = l[i].c[0] + m->N / 2, u1 = l[i].c[1] + m->N / 2, u2 = l[i].c[2] + m->Nz / 2;
        if (u0 % gxy || u1 % gxy || u2 % gz) continue;
        if (!(l[i].cnt > m->occ_thres)) continue;
        if (cnt < cap) {
            float p[3] = { (float)l[i].c[0] * m->vs, (float)l[i].c[1] * m->vs, (float)l[i].c[2] * m->vs };
            for (int a = 0; a < 3; ++a) xyz[cnt * 3 + a] = ((R[a * 3] * p[0] + R[a * 3 + 1] * p[1]) + R[a * 3 + 2] * p[2]) + T[a];   /* sijk_to_xyz mapping_common.py:234-238 */
            if (rgb) memcpy(rgb + (size_t)cnt * 3, l[i].col, sizeof(l[i].col));                /* :100-101 */
        }
        cnt++;
    }
    free(l); return cnt;
}

/* fuse_submaps_kernel :171-189 */
int ora_octo_fuse_submaps(ora_octo* g, const ora_octo* sub)
{
    ora_octo_reset(g);
    int nsub = sub->active;
    for (int s = 0; s < nsub && s < g->nsub; ++s) {
        for (int a = 0; a < 9; ++a) g->baseRf[s * 9 + a] = (float)g->baseR[s * 9 + a];
        for (int a = 0; a < 3; ++a) g->baseTf[s * 3 + a] = (float)g->baseT[s * 3 + a];
    }
    for (int s = 0; s < sub->nsub; ++s) {
        const leafmap* lm = &sub->sub[s]; if (!lm->a) continue;
        const float* R = g->baseRf + s * 9; const float* T = g->baseTf + s * 3;
        for (int i = 0; i < lm->nslots; ++i) {
            const leaf_t* l = &lm->a[i]; if (!l->used) continue;
            if (!(l->cnt > g->occ_thres)) continue;                                            /* :181 */
            float p[3] = { (float)l->c[0] * g->vs, (float)l->c[1] * g->vs, (float)l->c[2] * g->vs };
            int c[3];
            for (int a = 0; a < 3; ++a) { float x = ((R[a * 3] * p[0] + R[a * 3 + 1] * p[1]) + R[a * 3 + 2] * p[2]) + T[a]; c[a] = rnd_i(x / g->vs); }   /* :182-183 */
            if (!in_tree(g, c)) continue;
            leaf_t* d = leaf_get(&g->sub[0], c, 1);
            const int first = d->cnt == 0.0f;
            d->cnt += l->cnt;                                                                  /* :186 */
            if (g->cfg.texture_enabled) {       /* :189 is a race between source leaves: here the largest (submap, i, j, k) wins */
                const int32_t key[4] = { s, l->c[0], l->c[1], l->c[2] };
                int bigger = first;
                for (int a = 0; a < 4 && !bigger; ++a) { if (key[a] != d->wkey[a]) { bigger = key[a] > d->wkey[a]; break; } }
                if (bigger) { memcpy(d->col, l->col, sizeof(d->col)); memcpy(d->wkey, key, sizeof(key)); }
            }
        }
    }
    return 0;
}
