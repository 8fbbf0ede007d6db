/* sanafe_oracle.h -- C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a scalar CPU restatement of
 * SANA-FE's per-timestep loop used to check the MI355X path (single-threaded
 * unless oracle_set_threads asks for the reference's OpenMP-over-cores mode).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (sana-fe_amd/) never links or calls it.
 */
#ifndef SANAFE_ORACLE_H
#define SANAFE_ORACLE_H

#include <stdint.h>
#include "../include/sanafe_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_chip oracle_chip;

/* Per-timestep totals: the fields of `Timestep` (src/timestep.hpp:21-46). */
typedef struct oracle_ts
{
    int64_t timestep;
    int64_t spike_count;      /* synaptic events (src/chip.cpp:1039) */
    int64_t total_hops;
    int64_t packets_sent;
    int64_t neurons_updated;
    int64_t neurons_fired;
    int64_t n_messages;       /* including placeholders */
    double total_energy, synapse_energy, dendrite_energy, soma_energy,
            network_energy;
    double sim_time;
} oracle_ts;

/* One spike message: the traced fields of `Message` (src/message.hpp:19-62). */
typedef struct oracle_msg
{
    int64_t timestep, mid;
    int64_t src_neuron;       /* global neuron id in desc order */
    int64_t src_tile, src_core_offset, src_core_id;
    int64_t dest_tile, dest_core_offset, dest_core_id, dest_axon_id;
    int64_t hops, spikes, placeholder;
    int64_t src_x, src_y, dest_x, dest_y; /* tile coordinates (src/message.cpp:20-59) */
    double generation_delay, processing_delay, network_delay, blocking_delay,
            min_hop_delay, sent_timestamp, received_timestamp,
            processed_timestamp, messages_along_route;
} oracle_msg;

enum { ORACLE_TIMING_SIMPLE = 0, ORACLE_TIMING_DETAILED = 1 };

oracle_chip *oracle_create(const sanafe_desc *desc, char *err, int errlen);
void oracle_destroy(oracle_chip *chip);
/* One SpikingChip::step(); returns 0 on success, -1 and fills err otherwise. */
int oracle_step(oracle_chip *chip, int timing_model, oracle_ts *out, char *err,
        int errlen);
/* n_threads > 1: the two hot loops (and the counter reset) run as OpenMP `parallel for schedule(dynamic)` over cores,
 * as the reference's do (src/chip.cpp:629-632, 675-678, 1398-1401) -- the multithreaded CPU baseline of bench.py.
 * Message ids then depend on the thread schedule, as in the reference; parity tests use the default of 1. */
void oracle_set_threads(oracle_chip *chip, int n_threads);
/* NeuronStatus of every neuron after the last step, desc (global id) order. */
void oracle_get_status(const oracle_chip *chip, uint8_t *out);
/* soma get_potential() of every neuron, desc order. */
void oracle_get_potentials(const oracle_chip *chip, double *out);
/* named neuron trace (e.g. "u"), NaN where the neuron has no such trace. */
void oracle_get_trace(const oracle_chip *chip, const char *name, double *out);
/* Messages of the last step in per-source-core order; returns count. */
/* Optional perf-trace columns of the last step (sim_trace_get_optional_traces, src/chip.cpp:1541-1579):
 * NUL-separated names in lexicographic order + their values; returns the column count. */
int64_t oracle_optional_traces(const oracle_chip *chip, char *names, int64_t names_cap, double *values, int64_t cap);
int64_t oracle_get_messages(const oracle_chip *chip, oracle_msg *out,
        int64_t cap);
void oracle_reset(oracle_chip *chip);
/* MappedNeuron::set_attributes for one attribute (src/mapped.cpp:113-166). */
int oracle_set_neuron_attr(oracle_chip *chip, int64_t neuron, const char *key,
        int type, double num, const char *str, const double *list,
        int64_t list_len, int fwd, char *err, int errlen);
int64_t oracle_mapped_tiles(const oracle_chip *chip);

#ifdef __cplusplus
}
#endif
#endif
