// tsl_esdf.hip -- ESDF of the active submap from its TSDF.
//
// The reference's ESDF (taichi_slam/mapping/dense_esdf.py:228-333, reference root) is a legacy module that cannot be
// constructed at HEAD and whose queue propagation is incomplete; it serves as the DEFINITION only:
//   * voxels with |TSDF| < gamma are "fixed": ESDF := TSDF                                   (:228-230, :313-317)
//   * every other observed voxel starts at sign(TSDF)*max_dist                              (:325, :329)
//   * distances are lowered in magnitude through the 26-neighbourhood, edge cost |dir|*voxel (:282-297), only between
//     voxels on the same side of the surface (:289, :295).
// The fixed point of that relaxation is unique, so instead of the reference's two serial queues every brick (16^3 + halo)
// is staged in LDS and relaxed there for several sweeps per launch; launches repeat until no brick changes.
#include "tsl_tsdf.hpp"

namespace tsl {

#define ESDF_T 18
#define ESDF_T3 (ESDF_T * ESDF_T * ESDF_T)

__global__ void __launch_bounds__(256) k_esdf_init(MapDev M, int s, int nused, float* esdf, float gamma, float max_dist)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        if (M.owner[p] / M.nb3 != s) continue;
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const float t = h2f((h16)(M.tw[v] & 0xffffu));
            esdf[v] = (M.obs[v] > 0 && fabsf(t) < gamma) ? fabsf(t) : max_dist;
        }
    }
}

// flags: 0 not a node, 1 positive side, 2 negative side, +4 fixed
__global__ void __launch_bounds__(256) k_esdf_relax(MapDev M, int s, int nused, float* esdf, float gamma, float vs, int sweeps, int* changed)
{
    __shared__ float s_mag[ESDF_T3];
    __shared__ unsigned char s_flag[ESDF_T3];
    const float c1 = 1.0f * vs, c2 = sqrtf(2.0f) * vs, c3 = sqrtf(3.0f) * vs;              // dense_esdf.py:286
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = M.owner[p];
        if (owner / M.nb3 != s) continue;
        const int b = owner - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        for (int t = threadIdx.x; t < ESDF_T3; t += 256) {
            const int tz = t % ESDF_T, ty = (t / ESDF_T) % ESDF_T, tx = t / (ESDF_T * ESDF_T);
            const int i = bi * 16 + tx - 1 - M.hN, j = bj * 16 + ty - 1 - M.hN, k = bk * 16 + tz - 1 - M.hNz;
            unsigned char fl = 0; float mg = 0.0f;
            if (in_volume(M, i, j, k)) {
                int l; const int nb = brick_of(M, i, j, k, &l);
                const int np = nb == b ? p : pool_lookup_ro(M, s, nb);
                if (np >= 0) {
                    const size_t v = (size_t)np * TSL_BRK3 + l;
                    if (M.obs[v] > 0) {
                        const float tv = h2f((h16)(M.tw[v] & 0xffffu));
                        fl = (tv < 0.0f ? 2 : 1) | (fabsf(tv) < gamma ? 4 : 0);
                        mg = esdf[v];
                    }
                }
            }
            s_flag[t] = fl; s_mag[t] = mg;
        }
        __syncthreads();
        bool any = false;
        for (int it = 0; it < sweeps; ++it) {
            bool ch = false;
            for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
                const int t = (((l >> 8) + 1) * ESDF_T + (((l >> 4) & 15) + 1)) * ESDF_T + ((l & 15) + 1);
                const unsigned char fl = s_flag[t];
                if (fl == 0 || (fl & 4)) continue;
                float best = s_mag[t];
                for (int di = -1; di <= 1; ++di) for (int dj = -1; dj <= 1; ++dj) for (int dk = -1; dk <= 1; ++dk) {
                    const int m2 = di * di + dj * dj + dk * dk;
                    if (m2 == 0) continue;
                    const int n = t + (di * ESDF_T + dj) * ESDF_T + dk;
                    if ((s_flag[n] & 3) != (fl & 3)) continue;
                    const float cand = s_mag[n] + (m2 == 1 ? c1 : (m2 == 2 ? c2 : c3));
                    if (cand < best) best = cand;
                }
                if (best < s_mag[t]) { s_mag[t] = best; ch = true; }
            }
            any |= ch;
            if (!__syncthreads_or(ch)) break;
        }
        if (__syncthreads_or(any)) {
            for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
                const int t = (((l >> 8) + 1) * ESDF_T + (((l >> 4) & 15) + 1)) * ESDF_T + ((l & 15) + 1);
                esdf[(size_t)p * TSL_BRK3 + l] = s_mag[t];
            }
            if (threadIdx.x == 0) *changed = 1;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_esdf_export(MapDev M, int s, int nused, const float* esdf, float gamma, int16_t* idx, float* out, long long cap, int* counter)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = M.owner[p];
        if (owner / M.nb3 != s) continue;
        const int b = owner - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const bool pred = M.obs[v] > 0;
            const int o = wave_reserve(counter, pred);
            if (pred && o < cap) {
                idx[(size_t)o * 3] = (int16_t)(bi * 16 + (l >> 8) - M.hN); idx[(size_t)o * 3 + 1] = (int16_t)(bj * 16 + ((l >> 4) & 15) - M.hN);
                idx[(size_t)o * 3 + 2] = (int16_t)(bk * 16 + (l & 15) - M.hNz);
                const float t = h2f((h16)(M.tw[v] & 0xffffu));
                out[o] = fabsf(t) < gamma ? t : (float)sgn_f(t) * esdf[v];
            }
        }
    }
}

}  // namespace tsl

using namespace tsl;

extern "C" {

int tsl_esdf_update(tsl_tsdf* m, float gamma, float max_dist, int32_t* n_iters)
{
    TSL_REQUIRE(m, "esdf_update: null handle"); TSL_REQUIRE(gamma > 0 && max_dist > 0, "esdf_update: gamma and max_dist must be positive");
    TSL_HIP(hipSetDevice(m->device));
    int rc;
    if (!m->esdf) {
        if ((rc = dev_alloc(m, (void**)&m->esdf, sizeof(float) * (size_t)m->M.max_bricks * TSL_BRK3, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_flag, sizeof(int) * 4, 0))) return rc;
    }
    int nused = 0; if ((rc = tsl_tsdf_bricks_in_use(m, &nused))) return rc;
    const int s = m->cfg.is_global_map ? 0 : m->active;
    int iters = 0;
    if (nused > 0) {
        const int grid = nused < 8192 ? nused : 8192;
        hipLaunchKernelGGL(k_esdf_init, dim3(grid), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, gamma, max_dist);
        for (;;) {
            TSL_HIP(hipMemsetAsync(m->esdf_flag, 0, sizeof(int), ms(m)));
            for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(k_esdf_relax, dim3(grid), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, gamma, m->P.vs, 24, m->esdf_flag);
            iters += 2;
            TSL_HIP(hipMemcpyAsync(m->h_ints, m->esdf_flag, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
            TSL_HIP(hipStreamSynchronize(ms(m)));
            if (m->h_ints[0] == 0 || iters > 4096) break;
        }
    }
    m->esdf_gamma = gamma;
    if (n_iters) *n_iters = iters;
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

int tsl_esdf_export(tsl_tsdf* m, int16_t* idx, float* esdf, int64_t cap, int64_t* n)
{
    TSL_REQUIRE(m && n && cap >= 0, "esdf_export: bad argument"); TSL_REQUIRE(m->esdf, "esdf_export: call tsl_esdf_update first");
    TSL_HIP(hipSetDevice(m->device));
    int nused = 0; int rc = tsl_tsdf_bricks_in_use(m, &nused); if (rc) return rc;
    const size_t need = (((size_t)cap * 6 + 15) / 16) * 16 + (size_t)cap * 4 + 64;
    rc = grow(&m->xbuf, &m->xbuf_bytes, need); if (rc) return rc;
    int16_t* didx = (int16_t*)m->xbuf; float* dval = (float*)((char*)m->xbuf + (((size_t)cap * 6 + 15) / 16) * 16);
    int* counter = m->num_particles + 2;
    TSL_HIP(hipMemsetAsync(counter, 0, sizeof(int), ms(m)));
    const int s = m->cfg.is_global_map ? 0 : m->active;
    if (nused > 0) hipLaunchKernelGGL(k_esdf_export, dim3(nused < 8192 ? nused : 8192), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, m->esdf_gamma, didx, dval, (long long)cap, counter);
    TSL_HIP(hipMemcpyAsync(m->h_ints, counter, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    const int c = m->h_ints[0];
    *n = c;
    const size_t k = (size_t)(c < cap ? c : cap);
    if (k && idx) TSL_HIP(hipMemcpy(idx, didx, k * 6, hipMemcpyDeviceToHost));
    if (k && esdf) TSL_HIP(hipMemcpy(esdf, dval, k * 4, hipMemcpyDeviceToHost));
    return TSL_OK;
}

}  // extern "C"
