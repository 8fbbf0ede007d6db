// comm.cpp -- the per-timestep exchange of a tile-sharded chip (one process per GPU), inside the product.
//
// The reference routes every spike message of a step in one serial loop over the source cores
// (process_messages / receive_message, src/chip.cpp:656-708).  With the tiles sharded over GPUs the only thing
// a rank needs from the others is WHICH of their neurons fired (a message is a firing neuron x a static axon),
// so the exchange is an in-place all-gather of the spike bitmap windows: 1 bit per neuron slot.
//
//   RCCL      ncclAllGather (equal windows) or grouped ncclBroadcast (unequal ones) on a communication stream,
//             directly on the device bitmap; librccl is dlopen'ed on first use (no link-time dependency, and
//             no clash with the copy another library in the process may carry).
//   callback  a caller-supplied host all-gather (tests on one GPU, MPI bindings): the windows go through
//             host memory.
//
// Per step:  neurons -> gather (in line; or on the comm stream beside the delivery of the slices fed by local neurons only)
//            -> delivery of the remaining slices.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "comm.hpp"

namespace sanafe_amd
{
namespace
{
struct Rccl
{
    void *lib{nullptr};
    decltype(&ncclGetUniqueId) get_unique_id{nullptr};
    decltype(&ncclCommInitRank) comm_init_rank{nullptr};
    decltype(&ncclCommDestroy) comm_destroy{nullptr};
    decltype(&ncclAllGather) all_gather{nullptr};
    decltype(&ncclAllReduce) all_reduce{nullptr};
    decltype(&ncclBroadcast) broadcast{nullptr};
    decltype(&ncclGroupStart) group_start{nullptr};
    decltype(&ncclGroupEnd) group_end{nullptr};
    decltype(&ncclGetErrorString) error_string{nullptr};
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    if (r.lib || !r.error.empty()) return r;
    std::vector<std::string> names;
    if (const char *env = std::getenv("SANAFE_RCCL_LIB")) names.push_back(env);
    names.push_back("librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string &n : names)
    {
        // DEEPBIND: the library's own calls must bind to itself even if another RCCL is loaded globally
        r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (r.lib) break;
    }
    if (!r.lib)
    {
        r.error = std::string("cannot load librccl: ") + dlerror();
        return r;
    }
#define SANAFE_SYM(field, name)                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));            \
    if (!r.field) r.error = std::string("librccl lacks ") + name;
    SANAFE_SYM(get_unique_id, "ncclGetUniqueId")
    SANAFE_SYM(comm_init_rank, "ncclCommInitRank")
    SANAFE_SYM(comm_destroy, "ncclCommDestroy")
    SANAFE_SYM(all_gather, "ncclAllGather")
    SANAFE_SYM(all_reduce, "ncclAllReduce")
    SANAFE_SYM(broadcast, "ncclBroadcast")
    SANAFE_SYM(group_start, "ncclGroupStart")
    SANAFE_SYM(group_end, "ncclGroupEnd")
    SANAFE_SYM(error_string, "ncclGetErrorString")
#undef SANAFE_SYM
    return r;
}
} // namespace

#define NCCLCHK(expr)                                                                          \
    do                                                                                         \
    {                                                                                          \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) return set_error(std::string(#expr) + ": " + rccl().error_string(r_)); \
    } while (0)
#define HIPCHK(expr)                                                                           \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int Exchange::set_error(const std::string &msg)
{
    error = msg;
    return -1;
}

int Exchange::unique_id(uint8_t *id)
{
    static_assert(SANAFE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    Rccl &r = rccl();
    if (!r.error.empty()) return -1;
    ncclUniqueId uid;
    if (r.get_unique_id(&uid) != ncclSuccess) return -1;
    std::memcpy(id, uid.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

std::string Exchange::library_error() { return rccl().error; }

int Exchange::init_rccl(const uint8_t *id, int device, void *compute_stream)
{
    Rccl &r = rccl();
    if (!r.error.empty()) return set_error(r.error);
    close();
    HIPCHK(hipSetDevice(device));
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    NCCLCHK(r.comm_init_rank(&c, n_ranks, uid, rank));
    comm = c;
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    comm_stream = s;
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    ev_neurons = a;
    ev_gathered = b;
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, static_cast<size_t>(n_ranks) * sizeof(sanafe_hip_totals)));
    d_gather = p;
    stream = compute_stream;
    if (const char *env = std::getenv("SANAFE_COMM_OVERLAP")) overlap = std::atoi(env) != 0;
    kind = Rccl_;
    return 0;
}

int Exchange::init_callback(sanafe_allgather_fn fn, void *ctx)
{
    close();
    if (!fn) return set_error("null all-gather callback");
    callback = fn;
    callback_ctx = ctx;
    kind = Callback;
    return 0;
}

void Exchange::close()
{
    if (kind == Rccl_)
    {
        if (comm) rccl().comm_destroy(static_cast<ncclComm_t>(comm));
        if (comm_stream) (void) hipStreamDestroy(static_cast<hipStream_t>(comm_stream));
        if (ev_neurons) (void) hipEventDestroy(static_cast<hipEvent_t>(ev_neurons));
        if (ev_gathered) (void) hipEventDestroy(static_cast<hipEvent_t>(ev_gathered));
        if (d_gather) (void) hipFree(d_gather);
    }
    if (d_stage) (void) hipFree(d_stage);
    d_stage = nullptr;
    stage_bytes = 0;
    comm = comm_stream = ev_neurons = ev_gathered = d_gather = nullptr;
    kind = None;
}

Exchange::~Exchange() { close(); }

// Enqueues the gather of this step's spike windows.  RCCL: on the communication stream, behind the neuron
// launch (event), leaving the compute stream free for the local delivery; the caller makes the compute stream
// wait for gathered() before the remaining delivery.
int Exchange::gather_spikes_rccl(void *global_bits)
{
    Rccl &r = rccl();
    hipStream_t s = static_cast<hipStream_t>(stream), cs = overlap ? static_cast<hipStream_t>(comm_stream) : s;
    if (overlap)
    {
        HIPCHK(hipEventRecord(static_cast<hipEvent_t>(ev_neurons), s));
        HIPCHK(hipStreamWaitEvent(cs, static_cast<hipEvent_t>(ev_neurons), 0));
    }
    char *base = static_cast<char *>(global_bits);
    bool equal = true;
    const uint32_t w0 = slot_begin[1] - slot_begin[0];
    for (int k = 0; k < n_ranks; k++) equal = equal && (slot_begin[k + 1] - slot_begin[k]) == w0;
    if (equal) // in place: this rank's window is already where the gather puts it
    {
        NCCLCHK(r.all_gather(base + slot_begin[rank] / 8, base, w0 / 8, ncclUint8, static_cast<ncclComm_t>(comm), cs));
    }
    else
    {
        NCCLCHK(r.group_start());
        for (int k = 0; k < n_ranks; k++)
        {
            char *w = base + slot_begin[k] / 8;
            NCCLCHK(r.broadcast(w, w, (slot_begin[k + 1] - slot_begin[k]) / 8, ncclUint8, k, static_cast<ncclComm_t>(comm), cs));
        }
        NCCLCHK(r.group_end());
    }
    if (overlap) HIPCHK(hipEventRecord(static_cast<hipEvent_t>(ev_gathered), cs));
    return 0;
}

int Exchange::wait_gathered()
{
    if (overlap) HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_gathered), 0));
    return 0;
}

// Host path: every rank contributes its window padded to the widest one; the gathered rows are cut back.
int Exchange::gather_spikes_host(const uint32_t *local_bits, uint32_t *global_bits)
{
    uint32_t widest = 0;
    for (int k = 0; k < n_ranks; k++) widest = std::max(widest, slot_begin[k + 1] - slot_begin[k]);
    const size_t row = widest / 8;
    h_send.assign(row, 0);
    h_recv.assign(row * n_ranks, 0);
    std::memcpy(h_send.data(), local_bits, (slot_begin[rank + 1] - slot_begin[rank]) / 8);
    if (callback(callback_ctx, h_send.data(), row, h_recv.data()) != 0) return set_error("the all-gather callback failed");
    for (int k = 0; k < n_ranks; k++)
        std::memcpy(reinterpret_cast<char *>(global_bits) + slot_begin[k] / 8, h_recv.data() + row * k, (slot_begin[k + 1] - slot_begin[k]) / 8);
    return 0;
}

// The run totals of every rank, in rank order (the caller adds them up in that order: reproducible).
int Exchange::gather_totals(const sanafe_hip_totals &mine, void *device_totals, std::vector<sanafe_hip_totals> &all)
{
    all.assign(n_ranks, sanafe_hip_totals{});
    if (kind == Rccl_)
    {
        hipStream_t s = static_cast<hipStream_t>(stream);
        NCCLCHK(rccl().all_gather(device_totals, d_gather, sizeof(sanafe_hip_totals), ncclUint8, static_cast<ncclComm_t>(comm), s));
        HIPCHK(hipMemcpyAsync(all.data(), d_gather, all.size() * sizeof(sanafe_hip_totals), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }
    if (callback(callback_ctx, &mine, sizeof(mine), all.data()) != 0) return set_error("the all-gather callback failed");
    return 0;
}

int Exchange::gather_bytes(const void *send, size_t bytes, std::vector<unsigned char> &recv)
{
    recv.assign(bytes * static_cast<size_t>(n_ranks), 0);
    if (bytes == 0) return 0;
    if (kind == Rccl_)
    {
        hipStream_t s = static_cast<hipStream_t>(stream);
        const size_t need = bytes * static_cast<size_t>(n_ranks + 1);
        if (need > stage_bytes)
        {
            HIPCHK(hipStreamSynchronize(s));
            if (d_stage) HIPCHK(hipFree(d_stage));
            d_stage = nullptr;
            stage_bytes = 0;
            HIPCHK(hipMalloc(&d_stage, need));
            stage_bytes = need;
        }
        char *d_send = static_cast<char *>(d_stage), *d_recv = d_send + bytes;
        HIPCHK(hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, s));
        NCCLCHK(rccl().all_gather(d_send, d_recv, bytes, ncclUint8, static_cast<ncclComm_t>(comm), s));
        HIPCHK(hipMemcpyAsync(recv.data(), d_recv, recv.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }
    if (kind != Callback) return set_error("no exchange set up");
    if (callback(callback_ctx, send, bytes, recv.data()) != 0) return set_error("the all-gather callback failed");
    return 0;
}

// Element-wise maximum over the ranks of `count` doubles: device memory with RCCL (in place), host memory
// with the callback.
int Exchange::max_over_ranks(double *device_values, double *host_values, size_t count)
{
    if (count == 0) return 0;
    if (kind == Rccl_)
    {
        hipStream_t s = static_cast<hipStream_t>(stream);
        NCCLCHK(rccl().all_reduce(device_values, device_values, count, ncclDouble, ncclMax, static_cast<ncclComm_t>(comm), s));
        HIPCHK(hipMemcpyAsync(host_values, device_values, count * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }
    std::vector<double> all(count * n_ranks);
    if (callback(callback_ctx, host_values, count * sizeof(double), all.data()) != 0) return set_error("the all-gather callback failed");
    for (size_t i = 0; i < count; i++)
    {
        double m = all[i];
        for (int k = 1; k < n_ranks; k++) m = std::max(m, all[static_cast<size_t>(k) * count + i]);
        host_values[i] = m;
    }
    return 0;
}
} // namespace sanafe_amd
