// comm.hip - data-parallel collectives of the C ABI (fn_comm_*): RCCL called directly on the CALLER's stream.
//
// The reference has no distributed code; the step that must see the REDUCED gradient is clip_grad_norm_ + optimizer.step()
// (trainer_gmm.py:249-251).  One process per GPU; ranks exchange a 128-byte RCCL unique id out of band (the Python side uses the
// torch.distributed store / a gloo broadcast) and then call RCCL themselves: the collectives are ordinary stream work - they can
// be captured into the hipGraph of the training step, and no process-group watchdog thread touches the streams.
//
// librccl is bound at RUN TIME (dlopen of the SONAME, so a copy the process has already loaded - e.g. the one PyTorch ships - is
// reused): a single-GPU user needs no RCCL at all, and the library has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <class F>
bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* h = nullptr;
        for (const char* n : names) {                           // a copy that is already in the process first
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        for (int i = 0; i < 3 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        Rccl& r = g_rccl;
        r.handle = h;
        r.ok = bind(h, "ncclGetUniqueId", r.GetUniqueId) && bind(h, "ncclCommInitRank", r.CommInitRank) &&
               bind(h, "ncclCommDestroy", r.CommDestroy) && bind(h, "ncclAllReduce", r.AllReduce) && bind(h, "ncclAllGather", r.AllGather) &&
               bind(h, "ncclGetErrorString", r.GetErrorString) && bind(h, "ncclCommCount", r.CommCount) &&
               bind(h, "ncclCommUserRank", r.CommUserRank);
    });
    return g_rccl;
}

inline int rc(ncclResult_t r) { return r == ncclSuccess ? FN_OK : FN_COMM_ERROR_BASE + (int)r; }

static_assert(sizeof(ncclUniqueId) == FN_COMM_ID_BYTES, "fadernets.h: FN_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

}  // namespace

const char* fn_comm_strerror(int code) {
    const Rccl& r = rccl();
    if (code > FN_COMM_ERROR_BASE && r.ok) return r.GetErrorString((ncclResult_t)(code - FN_COMM_ERROR_BASE));
    return "RCCL error";
}

extern "C" {

int fn_comm_unique_id(void* id_out) {
    if (!id_out) return FN_E_NULL;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    return rc(r.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
}

int fn_comm_init(void** comm_out, int world, int rank, const void* id) {
    if (!comm_out || !id) return FN_E_NULL;
    if (world <= 0 || rank < 0 || rank >= world) return FN_E_SHAPE;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    const int e = rc(r.CommInitRank(&c, world, uid, rank));     // binds the communicator to the CURRENT device of the calling thread
    *comm_out = e == FN_OK ? (void*)c : nullptr;
    return e;
}

int fn_comm_destroy(void* comm) {
    if (!comm) return FN_OK;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    return rc(r.CommDestroy((ncclComm_t)comm));
}

int fn_comm_all_reduce_f32(void* comm, float* buf, size_t n, void* stream) {
    if (!comm || !buf) return FN_E_NULL;
    if (n == 0) return FN_OK;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    return rc(r.AllReduce(buf, buf, n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

int fn_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    if (!comm || !send || !recv) return FN_E_NULL;
    if (bytes_per_rank == 0) return FN_OK;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    return rc(r.AllGather(send, recv, bytes_per_rank, ncclInt8, (ncclComm_t)comm, (hipStream_t)stream));
}

// what RCCL itself says about the communicator (bench.py prints it: the proof that a multi-GPU line really ran over N RCCL ranks)
int fn_comm_count(void* comm, int* count_out) {
    if (!comm || !count_out) return FN_E_NULL;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    return rc(r.CommCount((ncclComm_t)comm, count_out));
}

int fn_comm_rank(void* comm, int* rank_out) {
    if (!comm || !rank_out) return FN_E_NULL;
    const Rccl& r = rccl();
    if (!r.ok) return FN_E_COMM;
    return rc(r.CommUserRank((ncclComm_t)comm, rank_out));
}

// ---- diagnostic: hold LDS on some compute units for a while (see fadernets.h) ----------------------------------------------
}  // extern "C"

namespace {
__global__ void occupy_kernel(long long cycles) {
    extern __shared__ int hold[];
    hold[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int acc = 0;
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) {
        acc += hold[(threadIdx.x + acc) & 63];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0x7fffffff) hold[0] = acc;        // practically never true: keeps the loop and the LDS alive
}
}  // namespace

extern "C" {

int fn_occupy_cus(int blocks, int lds_bytes, long long cycles, void* stream) {
    if (blocks <= 0 || lds_bytes < 256 || lds_bytes > 64 * 1024 || cycles < 0) return FN_E_SHAPE;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, cycles);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // extern "C"
