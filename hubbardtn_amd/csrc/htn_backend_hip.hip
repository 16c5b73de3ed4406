// HIP backend of the sweep engine (libhubbardtn_hip.so): device memory pool, stream, kernel launches, RCCL.
// There is no CPU path in this library: htn_ctx_create(HTN_BACKEND_CPU, ...) fails here.
#include <dlfcn.h>

#include <map>
#include <mutex>

#include "htn_common.h"
#include "htn_core.h"

namespace {

__global__ __launch_bounds__(256) void k_scale_z(double2* __restrict__ x, int64_t n, double f) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = x[j];
        x[j] = make_double2(v.x * f, v.y * f);
    }
}

// device -> host-mapped pinned memory by a kernel: a small hipMemcpyAsync D2H goes through the SDMA engine and costs
// ~100 us end to end on this part (measured per bond for the ~10 KB of singular values); zero-copy stores cost ~10
__global__ __launch_bounds__(256) void k_copy_words(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) dst[j] = src[j];
    __threadfence_system();
}

// ---- RCCL through dlsym: torch (when it is the host's plumbing) has already loaded its own librccl; binding at run
// time keeps ONE copy of the library in the process, and a plain C caller gets /opt/rocm/lib/librccl.so ----------------
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    void* CommInitRank = nullptr;                                    // (ncclComm_t*, int, ncclUniqueId BY VALUE, int)
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
struct NcclId {
    char internal[128];
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
        r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    });
    return r;
}
typedef int (*nccl_init_fn)(void**, int, NcclId, int);

struct HipBackend : htn::Backend {
    int device = 0;
    hipStream_t st = nullptr;
    bool own_stream = false;
    // size -> block.  Reuse is STREAM ORDERED: a block released while kernels that use it are still queued may be handed out
    // again at once, because whatever touches it next is enqueued behind them on the SAME stream `st`.  Invariant that makes
    // this sound: every piece of device work of this backend is either on `st`, or on the SVD's forked stream between a
    // fork event recorded on `st` and a join event `st` waits for before htn_jacobi_svd_z returns -- and no block is
    // allocated or released while that fork is open (`fork_open`, asserted in alloc / release).
    std::multimap<size_t, void*> free_list;
    bool fork_open = false;
    std::map<void*, size_t> live;
    size_t pooled = 0;
    void* lan_scratch = nullptr;
    int64_t lan_scratch_elems = 0;
    void* comm = nullptr;
    char* stage = nullptr;                        // pinned host staging ring of upload()
    size_t stage_cap = (size_t)32 << 20, stage_pos = 0;
    char* land = nullptr;                         // pinned, device-mapped landing buffer of download()
    char* land_dev = nullptr;
    size_t land_cap = (size_t)4 << 20;

    ~HipBackend() override {
        (void)hipSetDevice(device);
        if (st) (void)hipStreamSynchronize(st);
        htn_krylov_release_stream(st);                // the Lanczos / SVD drivers' per-stream scratch, events, forked stream
        htn_svd_release_stream(st);
        if (comm && rccl().ok) rccl().CommDestroy(comm);
        for (auto& kv : free_list) (void)hipFree(kv.second);
        for (auto& kv : live) (void)hipFree(kv.first);
        if (lan_scratch) (void)hipFree(lan_scratch);
        if (stage) (void)hipHostFree(stage);
        if (land) (void)hipHostFree(land);
        if (own_stream && st) (void)hipStreamDestroy(st);
    }
    int kind() const override { return HTN_BACKEND_HIP; }
    // every ABI entry point that reaches the device goes through here first: allocations, events and launches of this
    // context belong to ITS device whatever the calling thread's current device was
    int activate() override {
        HIP_TRY(hipSetDevice(device));
        return 0;
    }

    // HTN_DEBUG_POISON: every block handed out is filled with 0xFF (NaN) first, stream ordered like its first use
    void* poisoned(void* p, size_t bytes) {
        if (p && htn_debug_poison() && hipMemsetAsync(p, 0xFF, bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            htn::set_error("HTN_DEBUG_POISON: hipMemsetAsync failed");
            live.erase(p);
            (void)hipFree(p);
            return nullptr;
        }
        return p;
    }
    void* alloc(size_t bytes) override {
        if (fork_open) {
            htn::set_error("pool invariant violated: allocation while the SVD's forked stream is open");
            return nullptr;
        }
        bytes = (bytes + 255) / 256 * 256;
        // best fit within 25 %: per-bond buffers recur with slightly different sizes as the sector tables move
        auto it = free_list.lower_bound(bytes);
        if (it != free_list.end() && it->first <= bytes + bytes / 4 + 4096) {
            void* p = it->second;
            live[p] = it->first;
            pooled -= it->first;
            free_list.erase(it);
            return poisoned(p, live[p]);
        }
        void* p = nullptr;
        const size_t want = bytes + bytes / 8;       // headroom so that the next, slightly larger request still fits
        if (hipMalloc(&p, want) != hipSuccess) {
            // give the pool back to the driver and retry once with the exact size
            (void)hipGetLastError();
            (void)hipStreamSynchronize(st);
            for (auto& kv : free_list) (void)hipFree(kv.second);
            free_list.clear();
            pooled = 0;
            if (hipMalloc(&p, bytes) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            live[p] = bytes;
            return poisoned(p, bytes);
        }
        live[p] = want;
        return poisoned(p, want);
    }
    void release(void* p) override {
        if (fork_open) {                              // (cannot fail: record it, the next alloc reports)
            htn::set_error("pool invariant violated: release while the SVD's forked stream is open");
            abort();
        }
        auto it = live.find(p);
        if (it == live.end()) return;
        const size_t sz = it->second;
        live.erase(it);
        if (pooled + sz > ((size_t)64 << 30)) {      // keep at most 64 GiB of the 288 parked in the pool
            (void)hipStreamSynchronize(st);
            (void)hipFree(p);
            return;
        }
        free_list.insert({sz, p});
        pooled += sz;
    }
    // Host -> device through a pinned staging ring owned by the backend: the caller's (pageable, often short-lived)
    // source is copied out before this returns, the DMA itself is asynchronous on the stream.  (hipMemcpyAsync from
    // pageable memory pins the user's pages and returns before the transfer for larger sizes: a plan's std::vector
    // freed right after the call would be read after free.)
    int upload(void* dst, const void* src, size_t bytes) override {
        if (!bytes) return 0;
        if (bytes > stage_cap / 2) {
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            stage_pos = 0;
            return 0;
        }
        if (!stage) HIP_TRY(hipHostMalloc((void**)&stage, stage_cap));
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (stage_pos + need > stage_cap) {
            HIP_TRY(hipStreamSynchronize(st));
            stage_pos = 0;
        }
        memcpy(stage + stage_pos, src, bytes);
        HIP_TRY(hipMemcpyAsync(dst, stage + stage_pos, bytes, hipMemcpyHostToDevice, st));
        stage_pos += need;
        return 0;
    }
    // Device -> host through a pinned landing buffer + polling wait: a copy into pageable memory goes through the runtime's
    // own staging and a sleeping wait (~0.15 ms per bond for the ~10 KB of singular values, measured in the kernel trace)
    int download(void* dst, const void* src, size_t bytes) override {
        if (!bytes) return 0;
        if (bytes > land_cap) {
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            stage_pos = 0;
            return 0;
        }
        if (!land) {
            HIP_TRY(hipHostMalloc((void**)&land, land_cap, hipHostMallocMapped | hipHostMallocCoherent));
            HIP_TRY(hipHostGetDevicePointer((void**)&land_dev, land, 0));
        }
        if (bytes <= ((size_t)256 << 10) && ((uintptr_t)src & 3) == 0) {        // (device blocks are 256-byte granular: the
            const size_t words = (bytes + 3) / 4;                                //  last word may be read past `bytes`)
            hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<size_t>((words + 255) / 256, 64)), dim3(256), 0, st,
                               (uint32_t*)land_dev, (const uint32_t*)src, words);
            HIP_TRY(hipGetLastError());
        } else
            HIP_TRY(hipMemcpyAsync(land, src, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(htn_stream_spin(st));
        memcpy(dst, land, bytes);
        stage_pos = 0;                        // everything enqueued before has completed: the ring is free again
        return 0;
    }
    int zero(void* p, size_t bytes) override {
        if (!bytes) return 0;
        HIP_TRY(hipMemsetAsync(p, 0, bytes, st));
        return 0;
    }
    int sync() override {
        HIP_TRY(htn_stream_spin(st));
        stage_pos = 0;
        return 0;
    }
    int grouped_gemm(const void* const* bufs, const htn_tile* tiles, int32_t n_tiles, const htn_seg* segs) override {
        return htn_grouped_gemm_z(bufs, tiles, n_tiles, segs, st);
    }
    int lanczos(const htn_gemm_launch* stages, int n_stages, int x_slot, int y_slot, void* V, int64_t n, int krylovdim,
                double tol, int max_restart, int zero_y, htn_exchange2_fn exchange, void* user, double* eig, int* n_matvec,
                double* residual, double* matvec_ms) override {
        const int64_t need = htn_lanczos_scratch_elems(krylovdim);
        if (need > lan_scratch_elems) {
            if (lan_scratch) HIP_TRY(hipFree(lan_scratch));
            HIP_TRY(hipMalloc(&lan_scratch, sizeof(double2) * need));
            if (htn_debug_poison()) HIP_TRY(hipMemsetAsync(lan_scratch, 0xFF, sizeof(double2) * need, st));
            lan_scratch_elems = need;
        }
        int32_t nmv = 0;
        const int rc = htn_lanczos_z(stages, n_stages, x_slot, y_slot, V, n, krylovdim, tol, max_restart, lan_scratch, zero_y,
                                     exchange, user, eig, &nmv, residual, matvec_ms, st);
        *n_matvec = nmv;
        return rc;
    }
    int jacobi_svd(void* G, void* Vj, double* S, const htn_svd_block* desc_dev, const htn_svd_block* desc_host, int n_blocks,
                   int max_m, int max_sweeps, double tol, int32_t* info_dev, const htn_svd_opts* opts) override {
        fork_open = true;          // the call forks a second stream and joins it into `st` before it returns
        const int rc = htn_jacobi_svd_z(G, Vj, S, desc_dev, desc_host, n_blocks, max_m, max_sweeps, tol, info_dev, opts, st);
        fork_open = false;
        return rc;
    }
    int batched_copy(void* dst, const void* src, const int32_t* idx, const double* scl, const htn_copy_item* items, int n_items,
                     double gscale) override {
        return htn_batched_copy_z(dst, src, idx, scl, items, n_items, gscale, st);
    }
    int scale(void* x, int64_t n, double f) override {
        if (n <= 0) return 0;
        const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(k_scale_z, dim3(grid), dim3(256), 0, st, (double2*)x, n, f);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    int set_comm(int rank, int world, const void* id) override {
        Rccl& r = rccl();
        if (!r.ok) return htn::set_error("htn_ctx_set_comm: librccl.so could not be loaded");
        if (comm) {
            r.CommDestroy(comm);
            comm = nullptr;
        }
        NcclId uid;
        memcpy(uid.internal, id, sizeof(uid.internal));
        const int rc = ((nccl_init_fn)r.CommInitRank)(&comm, world, uid, rank);
        if (rc != 0) return htn::set_error("ncclCommInitRank: %s", r.GetErrorString ? r.GetErrorString(rc) : "error");
        return 0;
    }
    bool has_comm() const override { return comm != nullptr; }
    int allreduce(void* y, int64_t n) override {
        if (!comm) return htn::set_error("allreduce: no communicator (htn_ctx_set_comm)");
        // ncclDouble = 8, ncclSum = 0 (nccl.h enums, stable across NCCL 2.x / RCCL)
        const int rc = rccl().AllReduce(y, y, (size_t)(2 * n), 8, 0, comm, st);
        if (rc != 0) return htn::set_error("ncclAllReduce: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "error");
        return 0;
    }
};

}  // namespace

char* htn_err_buf() { return htn::err_buf(); }

namespace htn {
Backend* make_backend(int backend, int device, void* stream) {
    if (backend != HTN_BACKEND_HIP) {
        set_error("libhubbardtn_hip.so has no CPU backend (backend %d requested): there is no CPU fallback", backend);
        return nullptr;
    }
    char name[256];
    int cus = 0;
    if (htn_device_init(device, name, &cus)) return nullptr;
    auto* b = new HipBackend();
    b->device = device;
    if (stream) b->st = (hipStream_t)stream;
    else {
        if (hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking) != hipSuccess) {
            set_error("hipStreamCreate failed");
            delete b;
            return nullptr;
        }
        b->own_stream = true;
    }
    return b;
}
}  // namespace htn

extern "C" int htn_comm_unique_id(void* id_host) {
    Rccl& r = rccl();
    if (!r.ok) return htn::set_error("htn_comm_unique_id: librccl.so could not be loaded");
    const int rc = r.GetUniqueId(id_host);
    if (rc != 0) return htn::set_error("ncclGetUniqueId: %s", r.GetErrorString ? r.GetErrorString(rc) : "error");
    return 0;
}
