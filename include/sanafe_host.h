/* sanafe_host.h -- C API of libsanafe_host.so: the host side of the MI355X path.
 *
 * Mirrors the reference's SpikingChip surface (src/chip.hpp:56-107) over a
 * `sanafe_desc`: create = SpikingChip(arch) + load(net); sim = SpikingChip::sim;
 * the getters = get_spikes / get_potentials / get_traces / RunData.  The host
 * library maps the network (src/chip.cpp:129-408), lowers it to the device
 * image of sanafe_hip.h, drives libsanafe_hip.so and runs the `detailed`
 * NoC timing model (src/schedule.cpp:208-620) on the CPU, as the reference does.
 * All per-neuron outputs are indexed by the neuron's global id in desc order.
 */
#ifndef SANAFE_HOST_H
#define SANAFE_HOST_H

#include <stdint.h>
#include "sanafe_desc.h"
#include "sanafe_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sanafe_chip sanafe_chip;

enum { SANAFE_TIMING_SIMPLE = 0, SANAFE_TIMING_DETAILED = 1, SANAFE_TIMING_CYCLE = 2 };

typedef struct sanafe_chip_info
{
    uint32_t n_cores, n_local_cores, n_slots, n_global_slots, ring_slots, n_slices;
    uint64_t n_neurons, n_axons, n_synapses, mapped_tiles, mapped_cores;
    uint32_t n_soma_classes, n_cost_classes;
    double sync_delay;
    uint64_t image_bytes;
} sanafe_chip_info;

/* One spike message, the fields of `Message` that the traces print (src/message.hpp:19-62). */
typedef struct sanafe_message
{
    int64_t timestep, mid;
    int64_t src_neuron;       /* global neuron id (desc order): group + offset give src_neuron_group_id / _offset */
    int64_t src_tile, src_core_offset, src_core_id;
    int64_t dest_tile, dest_core_offset, dest_core_id, dest_axon_id;
    int64_t hops, spikes, placeholder;
    int64_t src_x, src_y, dest_x, dest_y; /* NoC coordinates of the tiles (src/message.cpp:20-59) */
    double generation_delay, processing_delay, network_delay, blocking_delay,
            min_hop_delay, sent_timestamp, received_timestamp,
            processed_timestamp, messages_along_route;
} sanafe_message;

const char *sanafe_last_error(void);
/* SpikingChip(arch) + load(net).  rank / n_ranks: tile-sharded multi-GPU runs (one process per GPU). */
int sanafe_chip_create(const sanafe_desc *desc, int device, int n_ranks, int rank, sanafe_chip **out);
void sanafe_chip_destroy(sanafe_chip *chip);
int sanafe_chip_get_info(sanafe_chip *chip, sanafe_chip_info *out);
sanafe_hip_chip *sanafe_chip_device(sanafe_chip *chip);
/* The lowered device image (host pointers, valid while the chip lives) and the
 * neuron -> global slot map; with device < 0 at create the chip is mapped only. */
int sanafe_chip_get_image(sanafe_chip *chip, sanafe_hip_image *out);
int sanafe_chip_get_slot_map(sanafe_chip *chip, uint32_t *slot_of_neuron);

/* `load(net, overwrite=false)` after timesteps have been simulated (src/chip.cpp:129-138): the caller lowers the programmed
 * networks plus the new one into `to` (ids, mapping order and per-core offsets continue, so the programmed neurons keep
 * their global ids) and this call moves the run-time state of the programmed neurons from `from` into it -- potentials,
 * LIF input currents, refractory counters, statuses, pending synaptic input (time-step buffers / delay lines), spike-train
 * cursors, the step counter and the run totals.  Chips with stochastic value streams, plugin / host-side units or `taps`
 * dendrites are refused (UnsupportedError). */
int sanafe_chip_carry_state(sanafe_chip *to, sanafe_chip *from);

/* SpikingChip::sim(timesteps, timing_model): returns the RunData of this call.
 * record bit 0 keeps per-step totals and spike lists for the getters, bit 2 (value 4) also the messages of every
 * step (the message trace, src/chip.cpp:440-460) -- under `detailed` timing with the scheduled timestamps, under
 * `simple` timing as schedule_messages_timestep_simple leaves them (network_delay = min_hop_delay, blocking_delay = 0,
 * timestamps -inf; src/schedule.cpp:61-102). */
#define SANAFE_RECORD_STEPS 1
#define SANAFE_RECORD_MESSAGES 4
#define SANAFE_RECORD_STATE 8 /* potentials / input currents of the neurons listed with sanafe_chip_set_state_log */
int sanafe_chip_sim(sanafe_chip *chip, int64_t timesteps, int timing_model, int record, sanafe_hip_totals *run_data);
/* Number of host threads that run the detailed NoC schedule of finished timesteps while the GPU
 * simulates ahead (SpikingChip::sim `scheduler_threads`, src/chip.cpp:291-349; 0 = inline). */
int sanafe_chip_set_scheduler_threads(sanafe_chip *chip, int n_threads);
/* MappedNeuron::set_attributes between sim() calls (src/mapped.cpp:113-166): forwards one attribute to the
 * neuron's soma unit, exactly as at load().  Built-in LIF / TrueNorth somas: every key of
 * <Model>::set_attribute_neuron (src/models.cpp:375-439, 664-722) -- the neuron moves to the parameter class with
 * the new value, `bias` / `potential` are patched in place; plugin somas: set_attribute_neuron of the plugin.
 * type is a SANAFE_ATTR_* (include/sanafe_desc.h); `str` is read for SANAFE_ATTR_STRING.  Changes reach the
 * device at the next sim()/step (or sanafe_chip_commit_attributes).  Neurons of other ranks are ignored. */
int sanafe_chip_set_attribute(sanafe_chip *chip, int64_t neuron, const char *key, int type, double num, const char *str);
/* The list-valued form (type SANAFE_ATTR_LIST): input neurons' `spikes` train, which also rewinds the train
 * (src/models.cpp:832-841).  Input neurons also take `rate`, and `poisson` when they had poisson > 0 at load(). */
int sanafe_chip_set_attribute_list(sanafe_chip *chip, int64_t neuron, const char *key, const double *values, int64_t count);
int sanafe_chip_commit_attributes(sanafe_chip *chip);
/* Generates the next `steps` rows of the external value streams (include/sanafe_hip.h: slot_ext,
 * sanafe_hip_write_ext) into out[steps][image.n_ext], advancing the host-side sources (Poisson
 * generators, the rand() sequence, noise files).  sim() does this itself; the call exists for mapped-only
 * chips and for callers that drive sanafe_hip_* directly. */
int sanafe_chip_generate_ext(sanafe_chip *chip, int64_t steps, int32_t *out);
/* Profiling hook for the host-side detailed NoC schedule: rebuilds the messages of one timestep from `status` (one
 * NeuronStatus byte per local slot of a single-rank chip, device or mapped-only) and schedules them `reps` times. */
int sanafe_test_schedule(sanafe_chip *chip, const uint8_t *status, int reps, double *sim_time, int64_t *n_messages,
        double *build_seconds, double *schedule_seconds);
/* The same for a chip whose message-pipeline somas run on the device (sanafe_hip_image::msg_*): `status` is what the NEURON LOOP
 * left (the step's status log), `msg_fired` per message into such a core how many of its synaptic events made the soma fire
 * (sanafe_hip_read_step_msg_fired) -- the message's processing delay depends on it (src/chip.cpp:738-789). */
int sanafe_test_schedule_msg(sanafe_chip *chip, const uint8_t *status, const uint16_t *msg_fired, double *sim_time, int64_t *n_messages);
/* Test hook: the optional perf columns (sanafe_chip_perf_columns) of one timestep from the same two logs. */
int sanafe_test_optional_columns(sanafe_chip *chip, const uint8_t *status, const uint16_t *msg_fired, double *out);
/* Self-check hook: the first n values of the host's restatement of glibc rand() for `seed`. */
void sanafe_test_glibc_rand(uint32_t seed, int64_t n, uint32_t *out);
int sanafe_chip_reset(sanafe_chip *chip);
double sanafe_chip_get_power(sanafe_chip *chip);

int sanafe_chip_get_status(sanafe_chip *chip, uint8_t *out);      /* [n_neurons] NeuronStatus */
int sanafe_chip_get_potentials(sanafe_chip *chip, double *out);   /* [n_neurons] */
int sanafe_chip_get_input_current(sanafe_chip *chip, double *out);/* [n_neurons] LIF trace "u" */
int sanafe_chip_get_step_totals(sanafe_chip *chip, int64_t first, int64_t count, sanafe_hip_totals *out);
/* Optional perf-trace columns (sim_trace_get_optional_traces, src/chip.cpp:1541-1579): `<tile>.energy`,
 * `<tile>.<core>.energy`, `<tile>.<core>.<unit>.energy` / `.latency` for every tile / core / unit whose description
 * sets log_energy / log_latency, in lexicographic order.  _perf_columns writes the NUL-separated names and returns
 * the column count; _get_step_optional copies out[count][columns] of the recorded steps (sim with record). */
int64_t sanafe_chip_perf_columns(sanafe_chip *chip, char *names, int64_t cap);
int sanafe_chip_get_step_optional(sanafe_chip *chip, int64_t first, int64_t count, double *out);
/* Potential and neuron traces (src/chip.cpp:1786-1831): the neurons whose potential (n_v) and LIF input current `u`
 * (n_u) every recorded step keeps, sampled on the device right after the neuron update (no per-step host round trip);
 * _get_step_state copies out[count][n_v + n_u] rows of a sim() run with SANAFE_RECORD_STATE. */
/* 1 when the architecture asks for optional perf columns (log_energy / log_latency flags): on a tile-sharded chip they need
 * the whole chip's tables (sanafe_chip_attach_whole) before a recorded sim(). */
int sanafe_chip_wants_perf_columns(sanafe_chip *chip);
int sanafe_chip_set_state_log(sanafe_chip *chip, int64_t n_v, const int64_t *neurons_v, int64_t n_u, const int64_t *neurons_u);
int sanafe_chip_get_step_state(sanafe_chip *chip, int64_t first, int64_t count, double *out);
/* fired flag per neuron (desc order) of recorded step `index` of the last sim */
int sanafe_chip_get_step_fired(sanafe_chip *chip, int64_t index, uint8_t *out);
/* messages of recorded step `index` (detailed timing + record only), per-source-core order */
int64_t sanafe_chip_get_step_messages(sanafe_chip *chip, int64_t index, sanafe_message *out, int64_t cap);

/* MappedNeuron::set_attributes between sim() calls (src/mapped.cpp:113-166): bias and potential */
int sanafe_chip_set_bias(sanafe_chip *chip, int64_t count, const int64_t *neurons, const double *bias);

/* ---- tile-sharded chips: one process per GPU, rank r of n_ranks holds a contiguous block of tiles ----
 * sanafe_chip_sim on such a chip exchanges the spike bitmap windows of all ranks once per timestep (the device
 * form of the reference's serial routing loop, src/chip.cpp:656-692) and returns the RunData of the WHOLE chip on
 * every rank: counters and energies summed over the ranks in rank order, sim_time from the per-step maximum over
 * all cores of all ranks (simple timing model, src/schedule.cpp:61-102).  The exchange must be set up first;
 * without one, sim() on a sharded chip fails.  Recorded runs (SANAFE_RECORD_STEPS: spike and perf traces) gather the
 * ranks' per-step records once per chunk of steps.  `detailed` timing and message traces are whole-chip host
 * algorithms (one global event queue): after sanafe_chip_attach_whole every rank gathers the NeuronStatus of all
 * neurons per chunk and replays the whole chip's messages on a mapped-only twin, as the reference does in its one
 * process.  Potential / neuron traces, optional perf columns and plugin units are refused on sharded chips.
 *   RCCL:     rank 0 calls sanafe_comm_unique_id, ships the 128 bytes to the others by whatever channel launched
 *             them, and every rank calls sanafe_chip_comm_init_rccl (collective: ncclCommInitRank).  The all-gather
 *             runs on a communication stream directly on the device bitmap, overlapped with the delivery of the
 *             slices fed by local neurons.
 *   callback: the caller supplies a blocking all-gather over host memory -- recv[r] (bytes each) = rank r's send;
 *             returns 0 on success.  For tests on one GPU and for MPI-style bindings. */
#define SANAFE_COMM_ID_BYTES 128
typedef int (*sanafe_allgather_fn)(void *ctx, const void *send, uint64_t bytes, void *recv);
int sanafe_comm_unique_id(uint8_t id[SANAFE_COMM_ID_BYTES]);
int sanafe_chip_comm_init_rccl(sanafe_chip *chip, const uint8_t id[SANAFE_COMM_ID_BYTES]);
int sanafe_chip_comm_init_callback(sanafe_chip *chip, sanafe_allgather_fn fn, void *ctx);
/* Maps the WHOLE chip a second time on the host (no device, one rank) from the complete description -- the tables the
 * detailed NoC schedule and the message trace of a tile-sharded chip run on (src/schedule.cpp:208-620 is one global
 * event queue over all messages of a step).  `desc` must describe every neuron and edge of the chip. */
int sanafe_chip_attach_whole(sanafe_chip *chip, const sanafe_desc *desc);

/* Split step for callers that drive the exchange themselves. */
int sanafe_chip_step_neurons(sanafe_chip *chip);
int sanafe_chip_step_deliver(sanafe_chip *chip, int timing_model);
int sanafe_chip_spike_buffers(sanafe_chip *chip, void **local_bits, uint64_t *local_bytes, void **global_bits,
        uint64_t *global_bytes, uint64_t *local_offset_bytes);
int sanafe_chip_synchronize(sanafe_chip *chip);
int sanafe_chip_read_totals(sanafe_chip *chip, sanafe_hip_totals *out);
/* SpikingChip::total_timesteps (src/chip.hpp:93): timesteps simulated since load(). */
int64_t sanafe_chip_total_timesteps(sanafe_chip *chip);

/* Synthetic random SNN edges for the benchmark configs: `out_degree` distinct
 * uniformly drawn targets per neuron, integer weights in {-8..8}\{0}; arrays hold
 * n_neurons * out_degree entries; global ids are offset by src_base / dst_base. */
int sanafe_generate_random_edges(int64_t n_neurons, int64_t out_degree, uint64_t seed, int n_threads,
        int64_t src_base, int64_t dst_base, int64_t *src, int64_t *dst, double *weight);

/* Sharded variant: only the edges whose source or destination neuron is in [lo, hi) (what one
 * rank of a tile-sharded run needs); same per-neuron streams as the unsharded generator. */
typedef struct sanafe_edge_set sanafe_edge_set;
int sanafe_generate_random_edges_sharded(int64_t n_neurons, int64_t out_degree, uint64_t seed, int n_threads,
        int64_t lo, int64_t hi, sanafe_edge_set **out, int64_t *count);
/* Locally connected variant (weak scaling): targets drawn uniformly from the `window` neurons centred on the source
 * (ids wrap); window == n_neurons is the uniform recipe.  Keeps the edges with source or destination in [lo, hi). */
int sanafe_generate_random_edges_windowed(int64_t n_neurons, int64_t out_degree, uint64_t seed, int n_threads,
        int64_t window, int64_t lo, int64_t hi, sanafe_edge_set **out, int64_t *count);
int sanafe_edge_set_copy(sanafe_edge_set *set, int64_t *src, int64_t *dst, double *weight);
void sanafe_edge_set_free(sanafe_edge_set *set);

#ifdef __cplusplus
}
#endif
#endif
