/* sanafe_hip.h -- C ABI of libsanafe_hip.so: SANA-FE's per-timestep simulation
 * loop on one MI355X (gfx950).  Plain pointers and sizes only; no torch, no C++
 * types.  This is the boundary a SANA-FE maintainer binds (INTEGRATION.md shows
 * the stub): `SpikingChip::load()` lowers the mapped network to a
 * `sanafe_hip_image` once, then every entry point below replaces one piece of
 * the reference's CPU loop:
 *
 *   sanafe_hip_chip_create   <- SpikingChip::load / map_neurons / map_connections /
 *                               map_axons            (src/chip.cpp:129-408, 1263-1391)
 *   sanafe_hip_step          <- SpikingChip::step -> sim_hw_timestep: reset measurements,
 *                               process_neurons, process_messages, forced_updates,
 *                               energy + counters, simple timing model
 *                               (src/chip.cpp:549-560, 624-764, 1028-1108, 1171-1261,
 *                               1393-1445; src/schedule.cpp:61-102)
 *   sanafe_hip_read_*        <- get_spikes / get_potentials / get_traces
 *                               (src/chip.cpp:1766-1831), RunData (src/chip.hpp:215-233)
 *   sanafe_hip_write_*       <- MappedNeuron::set_attributes between sim() calls
 *                               (src/mapped.cpp:113-166)
 *   sanafe_hip_reset         <- SpikingChip::reset (src/chip.cpp:576-600)
 *   sanafe_hip_step_neurons / _spike_buffers / _{export,import}_spikes / _step_deliver
 *                            <- the inter-tile message exchange when tiles are sharded
 *                               over GPUs (no reference equivalent)
 *
 * Every function returns 0 on success, a negative sanafe_hip_status otherwise;
 * sanafe_hip_last_error() gives the text.  Nothing here falls back to the CPU:
 * without a gfx950 device the calls fail.
 *
 * ---------------------------------------------------------------------------
 * Index spaces
 *   neuron slot g : cores are laid out back to back, each padded to a multiple of
 *                   64 slots (one wavefront handles 64 consecutive slots of ONE
 *                   core): g = core_nbase[c] + offset_within_core.
 *   axon a        : inbound axons of all cores, concatenated in destination-core
 *                   order; inside a core in the reference's delivery order
 *                   (source core id, source neuron order) -- src/chip.cpp:661-690.
 *   synapse s     : synapses of an axon are contiguous, in connection order
 *                   (src/chip.cpp:748-761), axons in axon order.
 * ---------------------------------------------------------------------------
 */
#ifndef SANAFE_HIP_H
#define SANAFE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sanafe_hip_status
{
    SANAFE_HIP_OK = 0,
    SANAFE_HIP_ERR_NO_DEVICE = -1,
    SANAFE_HIP_ERR_INVALID = -2,
    SANAFE_HIP_ERR_HIP = -3,
    SANAFE_HIP_ERR_UNSUPPORTED = -4
} sanafe_hip_status;

/* soma models (src/models.cpp:933-967) */
enum { SANAFE_SOMA_NONE = 0, SANAFE_SOMA_LIF = 1, SANAFE_SOMA_TRUENORTH = 2,
       SANAFE_SOMA_INPUT = 3, SANAFE_SOMA_HOST = 4,
       SANAFE_SOMA_PERSIST = 5 /* a neuron of a core whose time-step buffer sits before axon_out (`buffer_position:
                                  axon_out`): its neuron pipeline holds NO unit (src/mapped.cpp:168-188) -- nothing is updated or
                                  costed in the neuron loop, the status the message pipeline's soma calls left persists and is
                                  what axon_out acts on (a `fired` neuron sends its spike messages); the soma parameters of its
                                  class are TrueNorth's, used by the per-event updates (msg_* below) */ };
/* how a neuron's synaptic input reaches its soma */
enum {
    SANAFE_IN_BUFFERED = 0,   /* accumulator + kernel time-step buffer, or delay line: read slot t % ring_slots */
    SANAFE_IN_ZERO = 1,       /* `accumulator` with the buffer inside the unit: always 0.0 */
    SANAFE_IN_GATED = 3,      /* `accumulator_with_delay` reached through the kernel's buffer (`buffer_position: soma`,
                                 outside): the charge maturing at step t is handed to the soma at t+1 only if some event
                                 reached the neuron at step t (the unit is only called on events and its output is what the
                                 buffer keeps, src/models.cpp:96-131, src/chip.cpp:759); otherwise it is lost */
    SANAFE_IN_TAPS = 4,       /* `taps` dendrite (MultiTapModel1D, src/models.cpp:167-348) behind the kernel's buffer: the
                                 synapse's delay field carries the tap index; tap state advances once per step, the
                                 step's charge is added per tap, and tap 0 is handed to the soma at t+1 if an event
                                 reached the neuron at t (slot_aux = entry of the tap tables) */
    SANAFE_IN_LAST_DELAY = 5, /* `accumulator_with_delay` with the buffer BEFORE the dendrite unit: the neuron-side call happens
                                 every step (the line shifts, the matured charge reaches the soma) and integrates only the LAST
                                 synaptic event's current, without a synapse address -- so with the delay of the unit's synapse
                                 address 0 (src/models.cpp:96-131, src/pipeline.hpp:460-508); slot_aux = that delay */
    SANAFE_IN_NONE = 6,       /* the soma is called from the neuron loop WITHOUT an input (`buffer_position: soma`, inside the
                                 unit: the neuron pipeline is the soma alone, src/mapped.cpp:168-188); its synaptic input
                                 reaches it per event in the message pipeline (msg_* below) */
    SANAFE_IN_LAST = 2        /* buffer before the dendrite unit (`buffer_position: dendrite`, outside): the kernel's
                                 time-step buffer keeps only the LAST synaptic event's current (src/chip.cpp:759), which the
                                 accumulator then integrates alone; always "has input" (the lazy clear leaves 0.0) */
};
/* NeuronResetModes (src/arch.hpp:61-68) */
enum { SANAFE_RESET_NONE = 0, SANAFE_RESET_SOFT = 1, SANAFE_RESET_HARD = 2, SANAFE_RESET_SATURATE = 3 };

/* Soma parameter class: neurons with identical parameters share one entry
 * (LoihiCompartment src/models.hpp:222-241, TrueNorthNeuron :307-322). */
typedef struct sanafe_hip_soma_class
{
    double threshold, reverse_threshold, reset, reverse_reset;
    double leak_decay;      /* LIF; TrueNorth: additive `leak` */
    double input_decay;     /* LIF only */
    int32_t refractory_delay;
    uint8_t reset_mode, reverse_reset_mode;
    uint8_t force_update;
    uint8_t leak_towards_zero; /* TrueNorth only */
} sanafe_hip_soma_class;

/* Cost class: default energy/latency of the neuron-processing pipeline
 * (src/pipeline.hpp:574-714).  Index 0/1/2 = idle / updated / fired. */
typedef struct sanafe_hip_cost_class
{
    double soma_energy[3], soma_latency[3];
    double dendrite_energy, dendrite_latency; /* 0 when the dendrite is not in the neuron pipeline */
} sanafe_hip_cost_class;

typedef struct sanafe_hip_image
{
    /* ---- sizes ---- */
    uint32_t n_cores;
    uint32_t n_slots;        /* padded neuron slots, multiple of 64 */
    uint32_t n_soma_classes, n_cost_classes;
    uint32_t ring_slots;     /* rows of the time-step buffer / delay ring: >= 1; the host mapper passes 2 without synaptic delays (this
                              * step's row and the next one's: what push delivery needs), 6 with (max_delay 5, src/models.hpp:158) */
    uint32_t n_slices;       /* delivery work items, >= number of cores with inbound axons */
    uint64_t n_axons, n_synapses;
    uint32_t n_input;        /* input-model neurons */
    uint64_t n_train_words;  /* 32-bit words of packed input spike trains */
    uint32_t slot_offset;    /* multi-GPU: first GLOBAL slot held by this chip (0 on one GPU) */
    uint32_t n_global_slots; /* multi-GPU: slots of the whole chip; == n_slots on one GPU */
    double sync_delay;       /* ts_sync_delay_table.get(mapped_tiles), src/chip.cpp:562-574 */

    /* ---- per core [n_cores] ---- */
    const uint32_t *core_nbase;   /* first slot, multiple of 64 (local) */
    const uint32_t *core_ncount;  /* mapped neurons */
    const double *core_axon_out_latency; /* AxonOutUnit::latency_access of unit 0 */

    /* ---- class tables ---- */
    const sanafe_hip_soma_class *soma_classes;
    const sanafe_hip_cost_class *cost_classes;

    /* ---- per slot [n_slots] ---- */
    const uint32_t *slot_cls;     /* soma model (3b) | input kind (3b) << 3 | cost class (10b) << 6 | soma class (16b) << 16 */
    const double *slot_bias;
    const double *slot_v0;        /* initial potential */
    const uint32_t *slot_aux;     /* input-model neurons: index into in_*; others: 0 */
    /* static per-neuron totals of everything one spike of this neuron causes
     * (its messages, their hops, the synaptic events behind them) */
    const uint32_t *slot_packets; /* out-axons = messages per spike */
    const uint32_t *slot_hops;    /* sum of hops of those messages */
    const uint32_t *slot_events;  /* sum of synaptic events (Message::spikes) */
    const double *slot_e_net;     /* axon-out + hop + axon-in energy */
    const double *slot_e_syn;     /* synapse energy of the events */
    const double *slot_e_dend;    /* message-side dendrite energy of the events */

    /* ---- input model [n_input] (src/models.cpp:863-903) ---- */
    const uint32_t *in_train_beg; /* first bit of the spike train in in_train_bits */
    const uint32_t *in_train_len; /* bits */
    const int64_t *in_rate_period;/* (long)(1.0/rate), 0 = no rate input */
    const uint32_t *in_train_bits;/* [n_train_words] packed LSB first */

    /* ---- `taps` dendrites [n_taps] (optional) ---- */
    uint32_t n_taps;
    const uint32_t *tap_slot;     /* slot of the neuron */
    const uint32_t *tap_count;    /* taps, 1..8 */
    const double *tap_tc;         /* [n_taps][8] time constants */
    const double *tap_sc;         /* [n_taps][8] space constants between tap k and k+1 */

    /* ---- external per-step value streams (optional; host-generated, sanafe_hip_write_ext) ----
     * Three models consume a sequential host-side source on every update, which makes the
     * values a static schedule of (timestep, neuron): the input model's Poisson draw
     * (std::mt19937 per unit, src/models.cpp:883), TrueNorth's `std::rand() & random_mask`
     * (src/models.cpp:749-759) and the LIF noise file (src/models.cpp:535-539, 589-651).
     * slot_ext[s] = column of slot s in a row of stream values, or 0xffffffff. */
    uint32_t n_ext;               /* columns per timestep; 0 = none */
    const uint32_t *slot_ext;     /* [n_slots], may be NULL when n_ext == 0 */

    /* ---- delivery slices [n_slices] ---- */
    const uint32_t *slice_core;   /* destination core (local) */
    const uint64_t *slice_axon_beg, *slice_axon_end; /* axon range */
    const uint64_t *core_syn_base;/* [n_cores] first synapse of the core */
    const double *core_axon_in_latency; /* [n_cores] AxonInUnit::latency_spike_message of unit 0 */

    /* ---- per axon [n_axons] ---- */
    const uint32_t *ax_pre;       /* GLOBAL slot of the pre-synaptic neuron */
    const uint32_t *ax_syn_beg;   /* first synapse, relative to core_syn_base[dest core] */
    const uint32_t *ax_nsyn;      /* synapses behind the axon (Message::spikes) */
    const double *ax_proc_delay;  /* processing delay of the message, src/chip.cpp:738-764 */
    /* optional compression hint: when every synaptic event behind an axon costs the same
     * latency, ax_lat_class[a] < 255 indexes lat_class_per_event and the device evaluates
     * axon-in latency + nsyn * per-event latency instead of reading ax_proc_delay[a]
     * (255, or a NULL array: read ax_proc_delay) */
    const uint8_t *ax_lat_class;
    const double *lat_class_per_event; /* [255] */

    /* ---- per synapse [n_synapses] ---- */
    const uint32_t *syn_meta;     /* post-neuron offset in core (16b) | delay (3b) << 16 | drop (1b) << 19 */
    const double *syn_weight;

    /* ---- cores whose SOMA is part of the message pipeline (buffer inside the soma unit or before axon_out:
     *      src/pipeline.cpp:268-310, src/mapped.cpp:27-58) with built-in units -- `current_based` synapse, `accumulator`
     *      dendrite, `truenorth` soma: every synaptic event runs synapse -> dendrite (running sum of the step's currents,
     *      src/models.cpp:71-94) -> soma update, in delivery order (src/chip.cpp:738-789).  Their inbound axons are NOT in
     *      the arrays above (their places in syn_meta are holes with the drop bit); they are listed here per core, in
     *      delivery order, and the device walks them per post-synaptic neuron after the step's delivery.  n_msg_cores = 0:
     *      none.  Cores of this kind with plugin units, other models or `taps` run on the host instead (sanafe_host.h). ---- */
    uint32_t n_msg_cores;
    const uint32_t *msg_core;      /* [n_msg_cores] local core */
    const uint32_t *msg_ax_beg;    /* [n_msg_cores + 1] the core's axons in msg_ax_* */
    const uint32_t *msg_ax_pre;    /* GLOBAL slot of the axon's source neuron */
    const uint32_t *msg_ax_nsyn;   /* synapses of the axon; they follow each other in msg_syn_* in axon order */
    const uint32_t *msg_syn_beg;   /* [n_msg_cores + 1] the core's synapses in msg_syn_* */
    const uint32_t *msg_syn_post;  /* post-synaptic neuron, offset within the core */
    const double *msg_syn_weight;
    const struct sanafe_hip_msg_core_costs *msg_costs; /* [n_msg_cores] */
} sanafe_hip_image;

/* Default costs of the units a message passes on such a core (src/pipeline.hpp:511-731): per message the axon-in latency,
 * per synaptic event the synapse's and the dendrite's costs and the soma's by the status its update returned
 * (access + update always -- an update with an input current is never idle --, spike_out when it fired). */
typedef struct sanafe_hip_msg_core_costs
{
    double axon_in_latency;
    double synapse_energy, synapse_latency;   /* energy_process_spike, latency_process_spike */
    double dendrite_energy, dendrite_latency; /* energy_update, latency_update */
    double soma_energy[3], soma_latency[3];   /* access_neuron, update_neuron, spike_out */
} sanafe_hip_msg_core_costs;

/* Totals of one timestep / of a run: `Timestep` (src/timestep.hpp:21-46),
 * `RunData` (src/chip.hpp:215-233). */
typedef struct sanafe_hip_totals
{
    int64_t timesteps;
    int64_t spikes;          /* synaptic events */
    int64_t packets_sent, neurons_updated, neurons_fired, total_hops;
    double total_energy, synapse_energy, dendrite_energy, soma_energy, network_energy;
    double sim_time;         /* simple timing model; 0 when timing is left to the host scheduler */
} sanafe_hip_totals;

typedef struct sanafe_hip_chip sanafe_hip_chip;

const char *sanafe_hip_last_error(void);
int sanafe_hip_device_count(void);

/* Uploads the image to `device` and allocates all run-time state there. */
int sanafe_hip_chip_create(const sanafe_hip_image *image, int device, sanafe_hip_chip **out);
void sanafe_hip_chip_destroy(sanafe_hip_chip *chip);

/* Runs `n_steps` timesteps back to back on the chip's stream (asynchronous).
 * `record` bit 0 keeps a per-step record (totals + spike bitmap) for sanafe_hip_read_step_*, bit 3 the logged state
 * (sanafe_hip_set_state_log), bit 1 additionally the
 * NeuronStatus of every slot (sanafe_hip_read_step_status; what the host needs to rebuild a step's messages for the
 * detailed timing model without a round trip per step); `simple_timing` != 0 evaluates the simple timing
 * model on the device (src/schedule.cpp:61-102). */
int sanafe_hip_step(sanafe_hip_chip *chip, int64_t n_steps, int simple_timing, int record);
/* Device-side compaction of the logged neurons (potential trace, src/chip.cpp:1786-1805; neuron traces :1807-1831):
 * with `record` bit 3 (value 8) every recorded step keeps the potentials of the n_v listed slots followed by the LIF
 * input currents of the n_u listed slots -- sanafe_hip_read_step_state copies out[count][n_v + n_u] rows. */
int sanafe_hip_set_state_log(sanafe_hip_chip *chip, uint32_t n_v, const uint32_t *slots_v, uint32_t n_u, const uint32_t *slots_u);
int sanafe_hip_read_step_state(sanafe_hip_chip *chip, int64_t first, int64_t count, double *out);
/* Queues the external stream values of the next n_steps timesteps: values[step][column], int32:
 * input model 1 = "poisson_probability > U(0,1)" held, TrueNorth `rand() & mask`, LIF the sign-extended
 * noise value.  Every step of a chip with n_ext > 0 consumes one row; stepping past the queued rows
 * fails with SANAFE_HIP_ERR_INVALID (no default values are invented). */
int sanafe_hip_write_ext(sanafe_hip_chip *chip, int64_t n_steps, const int32_t *values);
int sanafe_hip_synchronize(sanafe_hip_chip *chip);
/* How the image was packed for the device (diagnostics, tests): synapse format 0 = 4 bytes (axon code, accumulator
 * index, int8 weight), 3 = 4 bytes (12-bit integer weight), 4 = 4 + 8 bytes (fp64 weight), 6 = 2 bytes (first-synapse
 * bit, 5-bit code into a dictionary of the chip's <= 32 distinct weights, 10-bit accumulator index), 7 = the words of
 * 6 for a dictionary of integers, summed in 32-bit integer accumulators -- the streamable layouts -- or the gather-only
 * fall-backs 1 = 4 bytes (12-bit weight), 2 = 4 + 8 bytes for cores with too many accumulators; number of delivery
 * slices whose axon records use the 2-byte delta form instead of the 8-byte form.  The environment variable
 * SANAFE_SYN_FORMAT (0, 1, 2, 3, 4, 6) picks a wider layout than the default where the image allows it (tests). */
int sanafe_hip_get_layout(sanafe_hip_chip *chip, int *syn_format, uint32_t *n_compact_slices);
/* > 0: the delivery workgroups sum integer weights in 32-bit integer LDS accumulators, every event adding
 * weight + 2^shift (formats 7, and 0 / 3 when the per-accumulator bounds hold); 0: fp64 accumulators.  The result is the
 * same bit for bit -- sums of integers are exact in any order; SANAFE_INT_ACC=0 keeps the fp64 accumulators (tests). */
int sanafe_hip_get_acc_shift(sanafe_hip_chip *chip);
/* Delivery slices whose axon records are a SOURCE BITMAP (format 7, dense fan-in: one bit per source slot of the slice's
 * 256-slot windows + one synapse-count byte per axon; "which axons spiked" is an AND with the spike bitmap).
 * SANAFE_AXON_BITMAP=0 keeps the 2-byte delta records (tests, A/B runs). */
int sanafe_hip_get_bitmap_slices(sanafe_hip_chip *chip);
/* 1: the chip's delivery kernel keeps 16 sub-accumulators per post-synaptic neuron in LDS, picked by the word's weight code
 * (bitmap records, cores of at most 256 neurons, no synaptic delays): the accumulator address is the 2-byte word itself with
 * two bits masked.  Same results (sums of integers).  SANAFE_SUB_ACCUMULATORS=0 keeps one accumulator per neuron. */
int sanafe_hip_get_sub_accumulators(sanafe_hip_chip *chip);
/* Push delivery (the neuron launch of a step with few spikes delivers them itself, walking the fired neurons' out-synapses,
 * and the delivery launch -- one probe per inbound axon of the chip whatever the activity -- returns at once; same result,
 * chosen per step on the device from the events of an earlier step): enabled = 1 when the chip qualifies (integer weights, no
 * synaptic delays / last-event cores / taps / host units / lost charge, one latency class per core, one GPU, ring_slots >= 2;
 * SANAFE_PUSH=0 switches it off), 2 when the chip is push-ONLY (at most one out-synapse per neuron on average,
 * SANAFE_PUSH_ONLY_DEGREE / SANAFE_PUSH_ONLY=0|1: every step is pushed, one launch per step); pushed_steps = steps delivered
 * that way since create. */
int sanafe_hip_get_push_info(sanafe_hip_chip *chip, uint32_t *enabled, uint32_t *pushed_steps);
/* Event-driven delivery (replaces the per-message walk of process_messages / process_message, src/chip.cpp:656-764, for
 * steps with few spikes on chips too large for push tables): a second copy of the format-7 synapse words, regrouped
 * source-neuron-major per group of destination cores, so that a step reads the spike bitmap, one table entry per (fired
 * neuron, core group) and that neuron's blocks of words -- work in proportion to the step's synaptic events, LDS integer
 * accumulators like the streaming kernel, same results.  Which kernel delivers a step is decided on the device (few
 * events three steps earlier -> events) and counted in sanafe_hip_get_push_info's pushed_steps.
 *   info[0] core groups (0: the chip has no event layout)   [1] segments of the source space (grid = groups x segments)
 *   [2] 16-byte units of all blocks   [3] words per (neuron, group) block x 1000   [4] lanes per block   [5] weight-code bits
 *   [6] accumulator shift   [7] 1: every step goes by events (SANAFE_EVENT=2)   [8] event threshold of the decision
 *   [9] events per step up to which the neuron-major block table is used (the group-major one above)   [10] steps that used it
 * SANAFE_EVENT=0 off / 1 build whatever the block length / 2 build and always use; SANAFE_EVENT_SEGMENTS,
 * SANAFE_EVENT_GROUP_CORES, SANAFE_EVENT_LPB (4 | 8), SANAFE_EVENT_MAX_EVENTS, SANAFE_EVENT_SPARSE_EVENTS. */
/* Cores whose soma is part of the message pipeline and that run on the device (sanafe_hip_image::msg_*): their number. */
int sanafe_hip_get_msg_cores(sanafe_hip_chip *chip);
/* ... and, for recorded steps with the status log (record bit 1), per message INTO such a core (= inbound axon, in the order
 * of msg_ax_*) how many of its synaptic events made the soma fire: what the message's processing delay depends on
 * (process_message, src/chip.cpp:738-789: the soma's latency is by the status its update returned) -- the host's NoC
 * schedule and message trace need it per step.  out: [count][msg_ax_beg[n_msg_cores]]. */
int sanafe_hip_read_step_msg_fired(sanafe_hip_chip *chip, int64_t first, int64_t count, uint16_t *out);
#define SANAFE_HIP_EVENT_INFO_FIELDS 11
int sanafe_hip_get_event_info(sanafe_hip_chip *chip, uint64_t *info, int n);

/* Bytes of the device layout, for roofline bookkeeping (bench.py): what the design itself has to move.
 *   [0] synapse words (+ fp64 weights in format 2)   [1] axon records   [2] chunk tables   [3] slice descriptors
 *   [4] global spike bitmap   -- one delivery launch reads [0..4] once when every chunk is streamed, less when
 *       few axons spike (gather path), plus 8 + 8 + 1 bytes per post-synaptic neuron and delay value it touches;
 *   [5] / [6] bytes the neuron launch reads / writes per step for the per-slot state (all mapped neurons)
 *   [7] bytes read per FIRED neuron on top (static downstream totals of its spike);
 *   [8] part of the axon-record array that a launch which streams every chunk does NOT read: the one synapse-count byte
 *       per axon behind the bitmap records (slice mode 2), read only for 256-slot windows with so few spiking axons that
 *       they take the gather path.  [1] excludes it.
 *   [9] / [10] event layout (sanafe_hip_get_event_info): all blocks of synapse words / the per-neuron tables.  A step
 *       delivered by events reads, per fired neuron, its blocks (2 bytes per synaptic event + padding to 16 bytes per
 *       block) and 8 + 4 bytes of [10] per core group. */
#define SANAFE_HIP_LAYOUT_FIELDS 11
int sanafe_hip_layout_bytes(sanafe_hip_chip *chip, uint64_t *out, int n);

/* Split step for tile-sharded (multi-GPU) runs and for host-evaluated (plugin)
 * soma units: neurons -> [exchange spike bitmaps] -> deliver. */
int sanafe_hip_step_neurons(sanafe_hip_chip *chip);
/* Records for the split step: the next n_steps split steps keep what sanafe_hip_step's `record` bits keep (bit 0 totals +
 * spike bitmap, bit 1 the NeuronStatus of every slot, bit 3 the logged state), read back with sanafe_hip_read_step_*;
 * record 0 switches recording off.  A tile-sharded chip records its own window; the host library gathers the ranks. */
int sanafe_hip_record_begin(sanafe_hip_chip *chip, int64_t n_steps, int record);
int sanafe_hip_step_deliver(sanafe_hip_chip *chip, int simple_timing, int record);
/* Tile-sharded runs overlap the exchange with delivery: the slices whose axons all start on this chip are
 * delivered first (they need only the local window of the bitmap, which the neuron launch wrote in place),
 * the others once the gathered bitmap is there; `_remote` also finishes the step (like step_deliver).
 * The reference delivers in one pass after its serial routing loop, src/chip.cpp:656-692. */
int sanafe_hip_step_deliver_local(sanafe_hip_chip *chip);
int sanafe_hip_step_deliver_remote(sanafe_hip_chip *chip, int simple_timing);
int sanafe_hip_slice_split(sanafe_hip_chip *chip, uint32_t *n_local, uint32_t *n_remote);
/* Simple timing model across GPUs: sim_time of a step = max over ALL cores of the chip + sync delay
 * (src/schedule.cpp:61-102), so each rank logs the largest per-core delay of every step it simulates
 * (entry `timestep % real_capacity`) and the caller takes the maximum over the ranks.  Allocates / returns the
 * device log: at least `capacity` entries, never shrinking -- real_capacity is what the ring arithmetic uses;
 * next_index = entry the next simulated step will write. */
int sanafe_hip_delay_log(sanafe_hip_chip *chip, int64_t capacity, double **device_log, int64_t *real_capacity,
        int64_t *next_index);
int sanafe_hip_read_delay_log(sanafe_hip_chip *chip, int64_t first, int64_t count, double *out);
/* Device address of the run totals (sanafe_hip_totals), for a device-side gather over the ranks. */
void *sanafe_hip_run_totals_device(sanafe_hip_chip *chip);
/* Device pointer + size (bytes) of this chip's local spike bitmap and of the
 * global bitmap the delivery kernel reads.  The local bitmap IS this chip's window
 * of the global one (local_bits == global_bits + slot_offset / 8), so the RCCL
 * all-gather runs in place on these. */
int sanafe_hip_spike_buffers(sanafe_hip_chip *chip, void **local_bits, uint64_t *local_bytes, void **global_bits,
        uint64_t *global_bytes);
/* Host-staged variant of the exchange (tests, or no device-to-device path): copy the local
 * bitmap out / the gathered global bitmap in; both synchronise the chip's stream. */
int sanafe_hip_export_spikes(sanafe_hip_chip *chip, uint32_t *local_bits_out);
int sanafe_hip_import_spikes(sanafe_hip_chip *chip, const uint32_t *global_bits);
void *sanafe_hip_stream(sanafe_hip_chip *chip); /* hipStream_t the kernels run on */
/* Run on a caller-owned stream instead (e.g. the stream a collective library orders against). */
int sanafe_hip_set_stream(sanafe_hip_chip *chip, void *hip_stream);

/* Run totals since create/reset_totals, and per-step records of the last sim. */
int sanafe_hip_read_totals(sanafe_hip_chip *chip, sanafe_hip_totals *out);
int sanafe_hip_reset_totals(sanafe_hip_chip *chip);
int sanafe_hip_read_step_totals(sanafe_hip_chip *chip, int64_t first, int64_t count, sanafe_hip_totals *out);
/* Spike bitmap (1 bit per local slot, LSB first) of recorded step `index`. */
int sanafe_hip_read_step_spikes(sanafe_hip_chip *chip, int64_t index, uint32_t *bits_out);
/* The bitmaps of recorded steps [first, first + count) in ONE copy: bits_out[count][n_slots / 32]. */
int sanafe_hip_read_step_spike_rows(sanafe_hip_chip *chip, int64_t first, int64_t count, uint32_t *bits_out);
/* NeuronStatus (0..3) per local slot after the last step. */
int sanafe_hip_read_status(sanafe_hip_chip *chip, uint8_t *out);
/* NeuronStatus of recorded steps [first, first + count) of the last sanafe_hip_step(record & 2): out[count][n_slots] */
int sanafe_hip_read_step_status(sanafe_hip_chip *chip, int64_t first, int64_t count, uint8_t *out);
int sanafe_hip_read_potentials(sanafe_hip_chip *chip, double *out);
int sanafe_hip_read_input_current(sanafe_hip_chip *chip, double *out); /* LIF `u` trace */
/* Per-core sums of the last step: generation-delay sum and processing-delay sum (either may be NULL).  proc_sum is refused
 * (SANAFE_HIP_ERR_UNSUPPORTED) on chips with push delivery: pushed steps price a core's messages inside the step reduction. */
int sanafe_hip_read_core_delays(sanafe_hip_chip *chip, double *gen_sum, double *proc_sum);

/* Parameter patches between sim() calls (slot-indexed, local). */
int sanafe_hip_write_bias(sanafe_hip_chip *chip, uint32_t first_slot, uint32_t count, const double *bias);
int sanafe_hip_write_potential(sanafe_hip_chip *chip, uint32_t first_slot, uint32_t count, const double *v);
int sanafe_hip_write_slot_class(sanafe_hip_chip *chip, uint32_t first_slot, uint32_t count, const uint32_t *cls);
/* Replaces the input-model tables (same meaning as in_train_beg/len/in_rate_period/in_train_bits of the image;
 * n_input must equal the image's) and rewinds the spike-train cursor of every input with rewind[i] != 0:
 * InputModel::set_attribute_neuron "spikes" / "rate" after load() (src/models.cpp:832-853). */
int sanafe_hip_write_inputs(sanafe_hip_chip *chip, uint32_t n_input, const uint32_t *train_beg, const uint32_t *train_len,
        const int64_t *rate_period, const uint32_t *train_bits, uint64_t n_train_words, const uint8_t *rewind);
int sanafe_hip_write_soma_classes(sanafe_hip_chip *chip, uint32_t n, const sanafe_hip_soma_class *classes);
/* Host-evaluated soma units (plugins, `extern "C" PipelineUnit *create_<model>()`,
 * src/plugins.cpp:45-98), between step_neurons and step_deliver:
 *   read_host_inputs   the synaptic input of the listed SANAFE_SOMA_HOST slots for the step in
 *                      flight (read-and-clear of the time-step buffer, src/chip.cpp:717-723)
 *   write_host_status  status (0..3) the plugin returned, the LOCAL core of each slot and the
 *                      energy / latency of its neuron-processing pipeline (src/pipeline.hpp:631-714);
 *                      folded into the step's totals with the spike's downstream costs. */
int sanafe_hip_read_host_inputs(sanafe_hip_chip *chip, uint32_t count, const uint32_t *slots, double *current_out,
        uint8_t *has_out);
int sanafe_hip_write_host_status(sanafe_hip_chip *chip, uint32_t count, const uint32_t *slots, const uint8_t *status,
        const uint32_t *core, const double *energy, const double *latency);

/* Cores that run on the HOST (the soma is part of the message pipeline -- `buffer_position: soma` inside the unit, or
 * `axon_out`, src/mapped.cpp:27-58 -- or a synapse / dendrite unit is a plugin, src/plugins.cpp:45-98): their neurons are
 * SANAFE_SOMA_HOST slots and they have no inbound axons in the image; the host replays their pipelines per timestep.
 *   write_host_core_status  between step_neurons and step_deliver: NeuronStatus of the listed slots and their local
 *                           cores; sets the spike bits and adds the static totals of the spikes (messages, hops,
 *                           events, network energy) -- nothing else
 *   write_host_core_costs   after step_deliver of the SAME step, before anything else is launched: per core, what its
 *                           units returned in the step (energies by role, soma-activity counters, the neuron pipelines'
 *                           latency sum = message generation delay, the messages' processing-delay sum) */
typedef struct sanafe_hip_host_core_costs
{
    uint32_t core, pad;       /* local core */
    double synapse_energy, dendrite_energy, soma_energy;
    double neuron_latency, processing_delay;
    int64_t neurons_updated, neurons_fired;
} sanafe_hip_host_core_costs;
int sanafe_hip_write_host_core_status(sanafe_hip_chip *chip, uint32_t count, const uint32_t *slots, const uint8_t *status,
        const uint32_t *core);
int sanafe_hip_write_host_core_costs(sanafe_hip_chip *chip, uint32_t count, const sanafe_hip_host_core_costs *costs);

/* The run-time state of a chip, slot by slot, in caller-owned host buffers -- what `load(net, overwrite=false)` on a chip that
 * has already simulated timesteps carries from the old lowering into the new one (src/chip.cpp:129-138 maps the new
 * neurons next to the programmed ones and keeps every unit's state).  Arrays are [n_slots] except ring / ring_valid
 * ([ring_slots][n_slots]: row (t % ring_slots) is what step t consumes) and in_pos ([n_input] spike-train cursors);
 * arrived / ring_last may be NULL when the chip has none. */
typedef struct sanafe_hip_state
{
    int64_t timesteps;      /* steps simulated so far (Timestep::timestep of the next step - 1) */
    double *v, *icur;
    int32_t *refrac;
    uint8_t *status;
    double *ring;
    uint8_t *ring_valid;
    uint8_t *arrived;
    uint32_t *ring_last;
    uint32_t *in_pos;
} sanafe_hip_state;
int sanafe_hip_export_state(sanafe_hip_chip *chip, sanafe_hip_state *out);
int sanafe_hip_import_state(sanafe_hip_chip *chip, const sanafe_hip_state *in);

/* SpikingChip::reset: potentials, input currents and buffers to zero. */
int sanafe_hip_reset(sanafe_hip_chip *chip);

/* Name and launch statistics of the kernels, for bench.py's roofline block:
 * event-timed average duration (ms) of the delivery kernel over the last
 * sanafe_hip_step call with `timed` set (see sanafe_hip_set_timing). */
int sanafe_hip_set_timing(sanafe_hip_chip *chip, int enabled);
int sanafe_hip_read_timing(sanafe_hip_chip *chip, double *neuron_ms, double *deliver_ms, double *reduce_ms,
        int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* SANAFE_HIP_H */
