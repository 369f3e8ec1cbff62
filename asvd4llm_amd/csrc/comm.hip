// comm.hip — C1: the one collective of the multi-GPU path: all-gather of per-layer sensitivities over RCCL.
// RCCL is resolved at run time (dlsym on the process first: torch ships and loads its own librccl; dlopen("librccl.so") otherwise), so
// libasvd_hip.so carries no link-time dependency and never pulls a second copy of the library into a torch process.
// The unique id travels through a file: rank 0 writes it atomically (tmp + rename), the others poll for it.
#include "common.h"
#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

struct NcclUniqueId { char internal[128]; };
typedef void* ncclComm_t;
enum { ncclFloat32 = 7, ncclFloat64 = 8 };  // nccl.h ncclDataType_t

struct Api {
    int (*GetUniqueId)(NcclUniqueId*);
    int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int);
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
    int (*CommDestroy)(ncclComm_t);
    bool ok;
};

Api load_api() {
    Api a{};
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclGetUniqueId")) {
        h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
    }
    a.GetUniqueId = (int (*)(NcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(ncclComm_t*, int, NcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    a.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    a.CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy;
    return a;
}

struct Comm {
    Api api;
    ncclComm_t comm;
    int rank, nranks;
    std::string id_path;
};

}  // namespace

extern "C" {

int asvd_comm_init(void** comm_out, int rank, int nranks, int device, const char* id_path, int timeout_s) {
    if (!comm_out || nranks < 1 || rank < 0 || rank >= nranks || !id_path || !*id_path) return ASVD_E_BADARG;
    Api api = load_api();
    if (!api.ok) return ASVD_E_HIP;
    ASVD_HIP_CHECK(hipSetDevice(device));
    NcclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    if (rank == 0) {
        if (api.GetUniqueId(&id) != 0) return ASVD_E_HIP;
        const std::string tmp = std::string(id_path) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return ASVD_E_BADARG;
        const size_t w = std::fwrite(&id, 1, sizeof(id), f);
        std::fclose(f);
        if (w != sizeof(id) || std::rename(tmp.c_str(), id_path) != 0) return ASVD_E_BADARG;
    } else {
        const int tries = (timeout_s > 0 ? timeout_s : 60) * 20;
        bool got = false;
        for (int t = 0; t < tries && !got; ++t) {
            FILE* f = std::fopen(id_path, "rb");
            if (f) {
                got = std::fread(&id, 1, sizeof(id), f) == sizeof(id);
                std::fclose(f);
            }
            if (!got) usleep(50 * 1000);
        }
        if (!got) return ASVD_E_HIP;
    }
    ncclComm_t c = nullptr;
    if (api.CommInitRank(&c, nranks, id, rank) != 0) return ASVD_E_HIP;
    Comm* cm = new Comm{api, c, rank, nranks, id_path};
    *comm_out = cm;
    return ASVD_OK;
}

static int allgather(void* comm, const void* send, void* recv, int64_t count, int dt, void* stream) {
    if (!comm || !send || !recv || count < 1) return ASVD_E_BADARG;
    Comm* cm = (Comm*)comm;
    return cm->api.AllGather(send, recv, (size_t)count, dt, cm->comm, (hipStream_t)stream) == 0 ? ASVD_OK : ASVD_E_HIP;
}
int asvd_comm_allgather_f32(void* comm, const float* send, float* recv, int64_t count, void* stream) {
    return allgather(comm, send, recv, count, ncclFloat32, stream);
}
int asvd_comm_allgather_f64(void* comm, const double* send, double* recv, int64_t count, void* stream) {
    return allgather(comm, send, recv, count, ncclFloat64, stream);
}

int asvd_comm_destroy(void* comm) {
    if (!comm) return ASVD_E_BADARG;
    Comm* cm = (Comm*)comm;
    const int rc = cm->api.CommDestroy(cm->comm);
    if (cm->rank == 0) std::remove(cm->id_path.c_str());
    delete cm;
    return rc == 0 ? ASVD_OK : ASVD_E_HIP;
}

}  // extern "C"
