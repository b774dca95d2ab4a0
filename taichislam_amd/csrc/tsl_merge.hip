// tsl_merge.hip -- multi-GPU global-map merge: one submap collection per GPU, ONE exchange at merge time.
// Replaces, for the swarm case, taichi_slam/mapping/submap_mapping.py:226-253 + taichi_slam/utils/communication.py:9-43 (submaps
// shipped between agents as zlib'd numpy dicts over LCM, then fused by every agent, dense_tsdf.py:272-318) with an all-reduce of the
// fusion's exact integer sums over RCCL / xGMI.
//
// Every rank splats its own submaps into the per-brick accumulators of its (reset) global map -- the same kernel as the single-GPU
// fusion (tsl_fuse.hip): {sum w*t, sum w} in 2^-24 fixed point (int64) + contribution | occupancy counts (int32) per voxel of every
// 16^3 brick it touches.  Only bricks travel:
//     1. a byte mask over the brick grid (nb^3 bytes: 32 KiB for 512^3) is all-reduced (MAX)  -> the union of touched bricks,
//     2. the union bricks are packed in ascending brick order (identical on every rank; bricks a rank did not touch are zeros),
//     3. the packed int64 / int32 planes are all-reduced (SUM),
//     4. every rank writes the same global TSDF from the sums.
// Integer sums make the merged map independent of the number of ranks and of the reduction order: bit-identical to one GPU fusing
// every submap (tests/test_merge_gpu.py).  RCCL is bound at run time (dlopen): the library has no link-time dependency on it, a caller
// that has its own collective (torch.distributed, MPI) uses the step functions and runs the two reductions itself.
#include "tsl_tsdf.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace tsl {

struct PoseTab { const float* p; };
int fuse_splat_into_global(tsl_tsdf* g, tsl_tsdf* sub, int* ndst, bool with_colour);          // tsl_fuse.hip

// ---- kernels ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_merge_mask(MapDev G, int nused, uint8_t* mask)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < nused) mask[G.owner[p]] = 1;                       // global map: one submap slot, owner = brick id
}

// ascending list of the bricks whose mask byte is set: one workgroup, block-wide prefix sums over the brick grid (deterministic order)
__global__ void __launch_bounds__(1024) k_merge_union(const uint8_t* mask, int nb3, int* list, int* count)
{
    __shared__ int s_w[16];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb3; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const bool on = b < nb3 && mask[b] != 0;
        const unsigned long long m = __ballot(on);
        const int wid = threadIdx.x >> 6;
        if (lane_id() == 0) s_w[wid] = popc64(m);
        __syncthreads();
        int off = s_base + rank_below(m);
        for (int q = 0; q < wid; ++q) off += s_w[q];
        if (on) list[off] = b;
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int q = 0; q < 16; ++q) tot += s_w[q]; s_base += tot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_base;
}

// packed planes of the union bricks (this rank's sums, zeros for bricks it did not touch); the local accumulators go back to zero
__global__ void __launch_bounds__(256) k_merge_pack(MapDev G, const int* list, int nunion, unsigned long long* acc, int* cnt,
                                                    ulonglong2* pacc, int* pcnt)
{
    for (int u = blockIdx.x; u < nunion; u += gridDim.x) {
        const int p = pool_lookup_ro(G, 0, list[u]);
        ulonglong2* a = reinterpret_cast<ulonglong2*>(acc) + (size_t)(p < 0 ? 0 : p) * TSL_BRK3;
        int* c = cnt + (size_t)(p < 0 ? 0 : p) * TSL_BRK3;
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const size_t o = (size_t)u * TSL_BRK3 + l;
            if (p >= 0) { pacc[o] = a[l]; pcnt[o] = c[l]; a[l] = make_ulonglong2(0ull, 0ull); c[l] = 0; }
            else { pacc[o] = make_ulonglong2(0ull, 0ull); pcnt[o] = 0; }
        }
    }
}

__global__ void __launch_bounds__(256) k_merge_finish(MapDev G, const int* list, int nunion, const ulonglong2* pacc, const int* pcnt)
{
    __shared__ int s_p;
    for (int u = blockIdx.x; u < nunion; u += gridDim.x) {
        if (threadIdx.x == 0) s_p = pool_claim<false>(G, 0, list[u]);       // bricks only other ranks touched are allocated now
        __syncthreads();
        const int p = s_p;
        if (p >= 0)
            for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
                const size_t o = (size_t)u * TSL_BRK3 + l;
                const int c = pcnt[o];
                if (c != 0) fuse_write_voxel(G, (size_t)p * TSL_BRK3 + l, (long long)pacc[o].x, (long long)pacc[o].y, c);
            }
        __syncthreads();
    }
}

// ---- the second form of the exchange (round 6): REDUCE-SCATTER the packed sums, every rank finalises the slice of union bricks it
// received, ALL-GATHER the finalised voxels.  A finalised voxel is its f16 {TSDF, W} word, its occupancy byte and one "written" bit: 5.125 bytes instead of the
// 20 bytes of sums an all-reduce sends round the ring a second time.  A brick's record: 4096 x u32 | 4096 x i8 | 4096 bits.
#define MRG_REC_BYTES (TSL_BRK3 * 4 + TSL_BRK3 + TSL_BRK3 / 8)
__global__ void __launch_bounds__(256) k_merge_final_slice(const ulonglong2* __restrict__ pacc, const int* __restrict__ pcnt, int nbricks, uint8_t* __restrict__ rec)
{
    for (int u = blockIdx.x; u < nbricks; u += gridDim.x) {
        uint8_t* const r = rec + (size_t)u * MRG_REC_BYTES;
        uint32_t* const tw = reinterpret_cast<uint32_t*>(r);
        int8_t* const oc = reinterpret_cast<int8_t*>(r + TSL_BRK3 * 4);
        unsigned long long* const bits = reinterpret_cast<unsigned long long*>(r + TSL_BRK3 * 5);
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const size_t o = (size_t)u * TSL_BRK3 + l;
            const int c = pcnt[o];
            uint32_t w = 0u; int8_t occ = 0;
            if (c != 0) {                                              // fuse_write_voxel (tsl_common.hpp), into the record instead of the map
                const float num = from_fix((long long)pacc[o].x), den = from_fix((long long)pacc[o].y);
                w = (uint32_t)f2h(num / den) | ((uint32_t)f2h(den) << 16);
                occ = (int8_t)(int)(int16_t)(c & 0xffff);
            }
            tw[l] = w; oc[l] = occ;
            const unsigned long long m = __ballot(c != 0);
            if (lane_id() == 0) bits[l >> 6] = m;
        }
    }
}
__global__ void __launch_bounds__(256) k_merge_finish_rec(MapDev G, const int* list, int nunion, const uint8_t* __restrict__ rec)
{
    __shared__ int s_p;
    for (int u = blockIdx.x; u < nunion; u += gridDim.x) {
        if (threadIdx.x == 0) s_p = pool_claim<false>(G, 0, list[u]);
        __syncthreads();
        const int p = s_p;
        if (p >= 0) {
            const uint8_t* const r = rec + (size_t)u * MRG_REC_BYTES;
            const uint32_t* const tw = reinterpret_cast<const uint32_t*>(r);
            const int8_t* const oc = reinterpret_cast<const int8_t*>(r + TSL_BRK3 * 4);
            const unsigned long long* const bits = reinterpret_cast<const unsigned long long*>(r + TSL_BRK3 * 5);
            for (int l = threadIdx.x; l < TSL_BRK3; l += 256)
                if ((bits[l >> 6] >> (l & 63)) & 1ull) { const size_t v = (size_t)p * TSL_BRK3 + l; G.tw[v] = tw[l]; G.obs[v] = 1; G.occ[v] = oc[l]; }
        }
        __syncthreads();
    }
}

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------------------------
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static int rccl_load()
{
    if (g_rccl.h) return TSL_OK;
    // a process that already carries an RCCL (torch ships one) keeps using that copy; otherwise ROCm's
    const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
    for (const char* n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!h) { set_error(std::string("RCCL not found: ") + (dlerror() ? dlerror() : "")); return TSL_ERR_HIP; }
    Rccl r; r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(h, "ncclReduceScatter");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(h, "ncclCommUserRank");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString || !r.ReduceScatter || !r.AllGather || !r.CommCount || !r.CommUserRank) { set_error("RCCL: missing symbols"); return TSL_ERR_HIP; }
    g_rccl = r;
    return TSL_OK;
}
#define TSL_NCCL(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { \
    tsl::set_error(std::string(#expr) + ": " + tsl::g_rccl.GetErrorString(_r)); return TSL_ERR_HIP; } } while (0)

static int merge_state(tsl_tsdf* g)            // per-handle exchange scratch: mask, union list, counter
{
    if (g->mrg_mask) return TSL_OK;
    int rc;
    if ((rc = dev_alloc(g, (void**)&g->mrg_mask, (size_t)g->nb3 + 16, 0))) return rc;      // + a status byte that travels with the mask
    if ((rc = dev_alloc(g, (void**)&g->mrg_list, sizeof(int) * (size_t)(g->nb3 + 4), 0))) return rc;
    g->mrg_count = g->mrg_list + g->nb3;
    return TSL_OK;
}

}  // namespace tsl

struct tsl_comm { ncclComm_t comm; int nranks, rank, device; };

using namespace tsl;

extern "C" {

// ---- step API -------------------------------------------------------------------------------------------------------------------------
int tsl_tsdf_merge_begin(tsl_tsdf* g, tsl_tsdf* sub, void* mask_dev, int64_t mask_bytes)
{
    TSL_REQUIRE(g && sub && mask_dev, "merge_begin: null argument");
    TSL_REQUIRE(g->cfg.is_global_map && g->device == sub->device, "merge_begin: destination must be a global map on the submaps' device");
    TSL_REQUIRE(mask_bytes >= g->nb3, "merge_begin: the mask needs one byte per 16^3 brick of the global grid (tsl_tsdf_merge_mask_bytes)");
    TSL_HIP(hipSetDevice(g->device));
    int rc = merge_state(g); if (rc) return rc;
    int ndst = 0;
    if ((rc = fuse_splat_into_global(g, sub, &ndst, false))) return rc;      // the exchange carries {sum w*t, sum w, count}: merged maps have no colour
    TSL_HIP(hipMemsetAsync(mask_dev, 0, (size_t)mask_bytes, ms(g)));
    if (ndst > 0) hipLaunchKernelGGL(k_merge_mask, dim3((ndst + 255) / 256), dim3(256), 0, ms(g), g->M, ndst, (uint8_t*)mask_dev);
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipStreamSynchronize(ms(g)));
    g->mrg_nunion = -1;
    return TSL_OK;
}

int tsl_tsdf_merge_mask_bytes(const tsl_tsdf* g, int64_t* n) { TSL_REQUIRE(g && n, "null"); *n = g->nb3; return TSL_OK; }

int tsl_tsdf_merge_union(tsl_tsdf* g, const void* mask_dev, int32_t* nunion)
{
    TSL_REQUIRE(g && mask_dev && nunion, "merge_union: null argument"); TSL_REQUIRE(g->mrg_mask, "merge_union: call merge_begin first");
    TSL_HIP(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_merge_union, dim3(1), dim3(1024), 0, ms(g), (const uint8_t*)mask_dev, g->nb3, g->mrg_list, g->mrg_count);
    TSL_HIP(hipMemcpyAsync(&g->h_ints[29], g->mrg_count, sizeof(int), hipMemcpyDeviceToHost, ms(g)));
    TSL_HIP(hipStreamSynchronize(ms(g)));
    g->mrg_nunion = g->h_ints[29];
    *nunion = g->mrg_nunion;
    return TSL_OK;
}

int tsl_tsdf_merge_pack(tsl_tsdf* g, void* acc_dev, void* cnt_dev)
{
    TSL_REQUIRE(g && g->mrg_nunion >= 0, "merge_pack: call merge_union first");
    TSL_REQUIRE(g->mrg_nunion == 0 || (acc_dev && cnt_dev), "merge_pack: null buffers");
    TSL_HIP(hipSetDevice(g->device));
    const int n = g->mrg_nunion;
    if (n > 0) hipLaunchKernelGGL(k_merge_pack, dim3(n < 4096 ? n : 4096), dim3(256), 0, ms(g), g->M, g->mrg_list, n,
                                  (unsigned long long*)g->fuse_acc, (int*)g->fuse_cnt, (ulonglong2*)acc_dev, (int*)cnt_dev);
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipStreamSynchronize(ms(g)));
    g->fuse_dirty = false;                          // every brick this rank splatted into is in the union: its sums were moved out and zeroed
    return TSL_OK;
}

int tsl_tsdf_merge_finish(tsl_tsdf* g, const void* acc_dev, const void* cnt_dev)
{
    TSL_REQUIRE(g && g->mrg_nunion >= 0, "merge_finish: call merge_union / merge_pack first");
    TSL_HIP(hipSetDevice(g->device));
    const int n = g->mrg_nunion;
    if (n > 0) hipLaunchKernelGGL(k_merge_finish, dim3(n < 4096 ? n : 4096), dim3(256), 0, ms(g), g->M, g->mrg_list, n, (const ulonglong2*)acc_dev, (const int*)cnt_dev);
    TSL_HIP(hipGetLastError());
    g->mrg_nunion = -1;
    return tsl_tsdf_sync(g);                        // reports an exhausted brick pool of the global map
}

// the reduce-scatter + all-gather form: a rank finalises the `nbricks` union bricks whose reduced sums it holds (any slice of the packed planes) into records of
// tsl_tsdf_merge_record_bytes each, and -- with all records gathered, union order -- writes the global map from them
int tsl_tsdf_merge_record_bytes(int64_t* n) { TSL_REQUIRE(n, "null"); *n = MRG_REC_BYTES; return TSL_OK; }
int tsl_tsdf_merge_finalize_slice(tsl_tsdf* g, const void* acc_dev, const void* cnt_dev, int32_t nbricks, void* rec_dev)
{
    TSL_REQUIRE(g && nbricks >= 0 && (nbricks == 0 || (acc_dev && cnt_dev && rec_dev)), "merge_finalize_slice: bad argument");
    TSL_HIP(hipSetDevice(g->device));
    if (nbricks > 0) hipLaunchKernelGGL(k_merge_final_slice, dim3(nbricks < 4096 ? nbricks : 4096), dim3(256), 0, ms(g), (const ulonglong2*)acc_dev, (const int*)cnt_dev, (int)nbricks, (uint8_t*)rec_dev);
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipStreamSynchronize(ms(g)));
    return TSL_OK;
}
int tsl_tsdf_merge_finish_records(tsl_tsdf* g, const void* rec_dev)
{
    TSL_REQUIRE(g && g->mrg_nunion >= 0, "merge_finish_records: call merge_union / merge_pack first");
    TSL_REQUIRE(g->mrg_nunion == 0 || rec_dev, "merge_finish_records: null records");
    TSL_HIP(hipSetDevice(g->device));
    const int n = g->mrg_nunion;
    if (n > 0) hipLaunchKernelGGL(k_merge_finish_rec, dim3(n < 4096 ? n : 4096), dim3(256), 0, ms(g), g->M, g->mrg_list, n, (const uint8_t*)rec_dev);
    TSL_HIP(hipGetLastError());
    g->mrg_nunion = -1;
    return tsl_tsdf_sync(g);
}

// ---- communicator + one-call form ---------------------------------------------------------------------------------------------------------
int tsl_comm_unique_id(char id[128])
{
    TSL_REQUIRE(id, "comm_unique_id: null"); int rc = rccl_load(); if (rc) return rc;
    ncclUniqueId u; TSL_NCCL(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, 128);
    return TSL_OK;
}
int tsl_comm_create(const char id[128], int nranks, int rank, int device, tsl_comm** out)
{
    TSL_REQUIRE(id && out && nranks >= 1 && rank >= 0 && rank < nranks, "comm_create: bad argument"); int rc = rccl_load(); if (rc) return rc;
    TSL_HIP(hipSetDevice(device));
    ncclUniqueId u; std::memcpy(u.internal, id, 128);
    ncclComm_t c = nullptr;
    TSL_NCCL(g_rccl.CommInitRank(&c, nranks, u, rank));
    *out = new tsl_comm{ c, nranks, rank, device };
    return TSL_OK;
}
void tsl_comm_destroy(tsl_comm* c) { if (!c) return; if (g_rccl.CommDestroy && c->comm) (void)g_rccl.CommDestroy(c->comm); delete c; }
void* tsl_comm_handle(tsl_comm* c) { return c ? (void*)c->comm : nullptr; }

// A rank that fails locally (capacity error of its submaps, an allocation, RCCL not loadable) must not leave the others inside a
// collective: every rank always runs the mask all-reduce -- a failed rank contributes an empty mask -- and a status byte travels behind
// the mask (MAX), so that after the first exchange every rank knows whether all of them can go on.  A failure after that point (the
// packed buffers) is exchanged the same way through a one-word all-reduce before the payload.  The error is returned on every rank.
int tsl_tsdf_allreduce_merge(tsl_tsdf* g, tsl_tsdf* sub, void* rccl_comm, int64_t* bytes_per_rank)
{
    TSL_REQUIRE(g && sub, "allreduce_merge: null handle");
    TSL_HIP(hipSetDevice(g->device));
    if (rccl_comm) { const int rc0 = rccl_load(); if (rc0) return rc0; }          // nothing collective can be issued without it
    ncclComm_t comm = (ncclComm_t)rccl_comm;
    int rc = merge_state(g);
    if (rc) return rc;                              // (out of memory before anything was exchanged: the peers see the communicator fail)
    std::string first_err;
    int local = tsl_tsdf_merge_begin(g, sub, g->mrg_mask, g->nb3);
    if (local) { first_err = tsl_last_error(); (void)hipMemsetAsync(g->mrg_mask, 0, (size_t)g->nb3, g->stream_); }
    hipStream_t st = g->stream_;
    int64_t bytes = 0;
    uint8_t status = local ? 1 : 0;
    TSL_HIP(hipMemcpyAsync(g->mrg_mask + g->nb3, &status, 1, hipMemcpyHostToDevice, st));
    TSL_HIP(hipStreamSynchronize(st));
    if (comm) { TSL_NCCL(g_rccl.AllReduce(g->mrg_mask, g->mrg_mask, (size_t)g->nb3 + 1, ncclUint8, ncclMax, comm, st)); bytes += g->nb3 + 1; }
    TSL_HIP(hipMemcpyAsync(&status, g->mrg_mask + g->nb3, 1, hipMemcpyDeviceToHost, st));
    TSL_HIP(hipStreamSynchronize(st));
    if (status) {
        g->mrg_nunion = -1;
        if (local) { set_error("allreduce_merge: " + first_err); return local; }
        set_error("allreduce_merge: another rank failed before the exchange; nothing was merged");
        return TSL_ERR_HIP;
    }
    int32_t n = 0;
    local = tsl_tsdf_merge_union(g, g->mrg_mask, &n);
    const size_t nv = (size_t)(local ? 0 : n) * TSL_BRK3;
    size_t nvpad = nv;                                              // merge_exchange 1: whole bricks per rank, the last slice padded with zero bricks
    int xranks = 1, xme = 0, xper = 0;                              // ... its communicator's size, this rank, union bricks per rank
    if (!local && g->merge_exchange == 1 && n > 0) {
        if (comm && (g_rccl.CommCount(comm, &xranks) != ncclSuccess || g_rccl.CommUserRank(comm, &xme) != ncclSuccess)) { xranks = 1; xme = 0; }
        xper = (n + xranks - 1) / xranks;
        nvpad = (size_t)xper * xranks * TSL_BRK3;
    }
    if (!local) local = grow(&g->mrg_pacc, &g->mrg_pacc_bytes, nvpad * 16 + 16);
    if (!local) local = grow(&g->mrg_pcnt, &g->mrg_pcnt_bytes, nvpad * 4 + 16);
    if (!local && xper > 0) {                                       // the second form's buffers are allocated HERE too: a failure travels with the status word below
        local = grow(&g->mrg_racc, &g->mrg_racc_bytes, (size_t)xper * TSL_BRK3 * 16 + 16);
        if (!local) local = grow(&g->mrg_rcnt, &g->mrg_rcnt_bytes, (size_t)xper * TSL_BRK3 * 4 + 16);
        if (!local) local = grow(&g->mrg_rec, &g->mrg_rec_bytes, (size_t)xper * xranks * MRG_REC_BYTES + 16);
    }
    if (!local && nvpad > nv) {
        (void)hipMemsetAsync((char*)g->mrg_pacc + nv * 16, 0, (nvpad - nv) * 16, st);
        (void)hipMemsetAsync((char*)g->mrg_pcnt + nv * 4, 0, (nvpad - nv) * 4, st);
    }
    if (!local) local = tsl_tsdf_merge_pack(g, g->mrg_pacc, g->mrg_pcnt);
    if (local) first_err = tsl_last_error();
    if (comm) {                                     // second status exchange: one word
        int* word = g->mrg_list + g->nb3 + 1;
        int hs = local ? 1 : 0;
        TSL_HIP(hipMemcpyAsync(word, &hs, sizeof(int), hipMemcpyHostToDevice, st));
        TSL_NCCL(g_rccl.AllReduce(word, word, 1, ncclInt32, ncclMax, comm, st));
        TSL_HIP(hipMemcpyAsync(&hs, word, sizeof(int), hipMemcpyDeviceToHost, st));
        TSL_HIP(hipStreamSynchronize(st));
        if (hs && !local) { set_error("allreduce_merge: another rank failed while packing; nothing was merged"); g->mrg_nunion = -1; return TSL_ERR_HIP; }
    }
    if (local) { set_error("allreduce_merge: " + first_err); g->mrg_nunion = -1; return local; }
    if (g->merge_exchange == 1 && n > 0) {
        // reduce-scatter + all-gather (option "merge_exchange" = 1): the packed planes are padded to nranks equal slices of whole bricks (the pad is zeros:
        // the buffers were grown for it and cleared above), rank r receives the sums of slice r, finalises it and every rank gathers the records
        const int nranks = xranks, me = xme, nper = xper;           // (the slice and record buffers were allocated with the packed planes, in front of the status exchange)
        const size_t sv = (size_t)nper * TSL_BRK3;                  // voxels of a slice
        if (comm) {
            TSL_NCCL(g_rccl.ReduceScatter(g->mrg_pacc, g->mrg_racc, sv * 2, ncclInt64, ncclSum, comm, st));
            TSL_NCCL(g_rccl.ReduceScatter(g->mrg_pcnt, g->mrg_rcnt, sv, ncclInt32, ncclSum, comm, st));
        } else {
            TSL_HIP(hipMemcpyAsync(g->mrg_racc, g->mrg_pacc, sv * 16, hipMemcpyDeviceToDevice, st));
            TSL_HIP(hipMemcpyAsync(g->mrg_rcnt, g->mrg_pcnt, sv * 4, hipMemcpyDeviceToDevice, st));
        }
        uint8_t* const mine = (uint8_t*)g->mrg_rec + (size_t)me * nper * MRG_REC_BYTES;
        hipLaunchKernelGGL(k_merge_final_slice, dim3(nper < 4096 ? nper : 4096), dim3(256), 0, st, (const ulonglong2*)g->mrg_racc, (const int*)g->mrg_rcnt, nper, mine);
        if (comm) TSL_NCCL(g_rccl.AllGather(mine, g->mrg_rec, (size_t)nper * MRG_REC_BYTES, ncclUint8, comm, st));
        // what a rank sends in a ring: (N - 1) / N of the scattered planes, then (N - 1) / N of the gathered records
        if (comm) bytes += (int64_t)((double)(nranks - 1) / nranks * ((double)nper * nranks * TSL_BRK3 * 20 + (double)nper * nranks * MRG_REC_BYTES));
        if (bytes_per_rank) *bytes_per_rank = bytes;
        TSL_HIP(hipGetLastError());
        return tsl_tsdf_merge_finish_records(g, g->mrg_rec);
    }
    if (comm && n > 0) {
        TSL_NCCL(g_rccl.AllReduce(g->mrg_pacc, g->mrg_pacc, nv * 2, ncclInt64, ncclSum, comm, st));
        TSL_NCCL(g_rccl.AllReduce(g->mrg_pcnt, g->mrg_pcnt, nv, ncclInt32, ncclSum, comm, st));
        bytes += (int64_t)nv * 20;
    }
    if (bytes_per_rank) *bytes_per_rank = bytes;
    return tsl_tsdf_merge_finish(g, g->mrg_pacc, g->mrg_pcnt);
}

}  // extern "C"
