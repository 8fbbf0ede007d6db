// comm.hpp -- the per-timestep spike exchange of a tile-sharded chip (see comm.cpp).
#ifndef SANAFE_HOST_COMM_HPP
#define SANAFE_HOST_COMM_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sanafe_host.h"

namespace sanafe_amd
{
struct Exchange
{
    enum Kind { None = 0, Rccl_ = 1, Callback = 2 };
    Kind kind{None};
    int n_ranks{1}, rank{0};
    std::vector<uint32_t> slot_begin; // [n_ranks + 1] window of every rank in the global slot space
    std::string error;

    // RCCL (opaque here: ncclComm_t, hipStream_t, hipEvent_t)
    void *comm{nullptr}, *comm_stream{nullptr}, *ev_neurons{nullptr}, *ev_gathered{nullptr}, *d_gather{nullptr};
    void *stream{nullptr}; // the chip's compute stream
    // Gather in line on the compute stream (default), or on the communication stream beside the delivery of the
    // slices fed by local neurons only (SANAFE_COMM_OVERLAP=1).  Measured on MI355X with a one-rank communicator:
    // the two cross-stream dependencies of the overlapped form cost ~20 us per step, the in-line gather ~0.5-3 us;
    // and with random connectivity nearly every slice has a remote source, so there is little to overlap with.
    bool overlap{false};
    // host callback
    sanafe_allgather_fn callback{nullptr};
    void *callback_ctx{nullptr};
    std::vector<unsigned char> h_send, h_recv;

    ~Exchange();
    static int unique_id(uint8_t *id);
    static std::string library_error();
    int init_rccl(const uint8_t *id, int device, void *compute_stream);
    int init_callback(sanafe_allgather_fn fn, void *ctx);
    void close();
    int gather_spikes_rccl(void *global_bits);
    int wait_gathered();
    int gather_spikes_host(const uint32_t *local_bits, uint32_t *global_bits);
    int gather_totals(const sanafe_hip_totals &mine, void *device_totals, std::vector<sanafe_hip_totals> &all);
    int max_over_ranks(double *device_values, double *host_values, size_t count);
    // Host-memory all-gather of `bytes` per rank (records of a chunk of steps: totals, spike rows, status rows): recv holds
    // n_ranks blocks in rank order.  RCCL stages through device memory; not on the per-step path.
    int gather_bytes(const void *send, size_t bytes, std::vector<unsigned char> &recv);
    void *d_stage{nullptr};
    size_t stage_bytes{0};
    int set_error(const std::string &msg);
};
} // namespace sanafe_amd
#endif
