// Communicator owned by an engine handle (SURVEY.md 8b "Ownership": the engine owns its RCCL communicator; 8e: the halo
// exchange is a sparse all-to-all-v, one grouped ncclSend/ncclRecv pair per neighbour over its own xGMI link).
// Two transports behind one interface:
//   MGN_COMM_RCCL  RCCL (librccl.so.1 bound at run time: the copy that is already in the process -- PyTorch-ROCm wheels bundle
//                  one -- or ROCm's), collectives on a private communication stream, ordered against the compute stream by events.
//   MGN_COMM_HOST  POSIX shared memory on one node: rows are staged through the host.  Not the production wire; it exists so
//                  that several ranks can share ONE GPU (RCCL refuses two ranks on a device; the test boxes have one), so that
//                  host-only handles can exchange, and as a fallback where RCCL cannot initialise.
// There is no precedent in the reference (single device, src/MeshGraphNets.jl:255-263).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace mgn {

struct Comm {
    int rank = 0, nranks = 1, transport = 0;
    std::string err;
    virtual ~Comm() {}
    // Sparse all-to-all-v.  Peer q gets send[soff[q] .. soff[q] + sbytes[q]) and fills recv[roff[q] .. roff[q] + rbytes[q]).
    // Device form: `send` is complete in stream order of `compute` at the call; after a2a_finish returns, work enqueued on
    // `compute` sees `recv`.  Between the two calls the caller may enqueue independent work on `compute` (the overlap).
    virtual int a2a_start(const void* send, const size_t* sbytes, const size_t* soff, void* recv, const size_t* rbytes,
                          const size_t* roff, hipStream_t compute) = 0;
    virtual int a2a_finish(hipStream_t compute) = 0;
    // Host form (blocking), for host-only handles.  RCCL transport: unsupported.
    virtual int a2a_host(const void* send, const size_t* sbytes, const size_t* soff, void* recv, const size_t* rbytes,
                         const size_t* roff) = 0;
    // recv [nranks][bytes] on the device <- every rank's send [bytes]; complete in stream order of `compute`.
    virtual int allgather(const void* send, size_t bytes, void* recv, hipStream_t compute) = 0;
    // x[n] on the HOST, reduced over ranks in place (op 0 = sum, 1 = max); blocking; the same bits on every rank.
    virtual int allreduce_f64(double* x, int n, int op, hipStream_t compute) = 0;
    virtual int barrier(hipStream_t compute) = 0;
};

constexpr size_t COMM_ID_BYTES = 128;   // == NCCL_UNIQUE_ID_BYTES

// id: COMM_ID_BYTES bytes every rank of the communicator passes identically (made by comm_unique_id on one rank).
// Return nullptr and set `why` on failure.  device_ok == false: a host-only handle (HOST transport only).
int comm_unique_id(void* id, int transport, std::string& why);
Comm* comm_create(const void* id, int transport, int rank, int nranks, bool device_ok, std::string& why);

}  // namespace mgn
