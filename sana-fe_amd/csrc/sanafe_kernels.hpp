// sanafe_kernels.hpp -- device side of libsanafe_hip: constants, device views of the image and state, and the
// kernels (neuron_kernel, deliver_kernel, the two-level step reduction, host-unit kernels).  Included by
// sanafe_hip.hip inside its anonymous namespace; see the header comment there for the map of the kernels.
#pragma once

constexpr int WAVE = 64;
constexpr int NEURON_BLOCK = 256;   // 4 wavefronts = up to 4 consecutive 64-slot chunks of ONE simulated core
constexpr int PARTS_PER_WG = NEURON_BLOCK / WAVE; // step partials: one per wavefront of a neuron workgroup (WgPart)
constexpr int DELIVER_BLOCK = 256;
constexpr int AX_PER_THREAD = 4;    // axon records per lane: one 8-byte (compact) or two 16-byte (wide) loads
constexpr int REDUCE_BLOCK = 256;
constexpr uint32_t L1_CORES = 16;   // cores one wavefront of level 1 of the step reduction folds (four lanes per core)
constexpr uint32_t SOMA_LDS_MAX = 128; // soma parameter classes staged in LDS by the neuron kernel (7 KiB)
constexpr uint32_t COST_LDS_MAX = 64;  // cost classes staged in LDS (4 KiB)

// One neuron workgroup: up to 4 chunks of one core (a 256-neuron TrueNorth core is one workgroup, a 1024-neuron
// Loihi core four).  One 16-byte scalar load tells the workgroup everything it needs before its slot loads.
#ifndef SANAFE_NEURON_SGPRS
#define SANAFE_NEURON_SGPRS 102 // (80 made the kernel reload kernel arguments a dozen times before its first vector load: +0.8 us per launch on the 1 M neuron chip)
#endif
#ifndef SANAFE_NEURON_ATTR
// the uniform TrueNorth instantiation fits 64 registers without spilling: 8 wavefronts per SIMD instead of 6 (+4 % on C4)
#define SANAFE_NEURON_ATTR __attribute__((amdgpu_waves_per_eu((UNI && MODEL == SANAFE_SOMA_TRUENORTH) ? 8 : 6, 8)))
#endif
struct WgDesc
{
    uint32_t slot0;   // first (local) slot
    uint32_t core;    // local core
    uint32_t nchunks; // 1..4
    uint32_t pad;
};
// Partial sums of one WAVEFRONT of a neuron workgroup (one writer), by step parity: PARTS_PER_WG per workgroup.
struct WgPart
{
    double e_soma, e_dend, e_syn, e_net, lat;
    long long updated, fired, packets, hops, events;
    long long counted; // uniform chips (DevImage::uni_costing): neurons the default costing covers; level 1 of the step
    long long pad;     // reduction prices them per core (idle = counted - updated, updated-only = updated - fired, fired)
};
// Level-1 result of the step reduction: L1_CORES consecutive cores folded by one wavefront.
struct GroupPart
{
    double e_soma, e_dend, e_syn, e_net, gmax, pmax;
    long long updated, fired, packets, hops, events;
};
// Everything one spike of a neuron causes downstream is static (its messages, their hops, the synaptic
// events behind them, their energy): one 40-byte record per slot, read only by lanes whose neuron fired.
struct SpikeStatic
{
    double e_net, e_syn, e_dend;
    uint32_t packets, hops, events, pad;
};

// One delivery slice: everything its workgroup needs before the first record load, in one 64-byte scalar load.
struct SliceDesc
{
    unsigned long long rec_off;  // byte offset of the slice's axon records in ax_bytes
    unsigned long long syn_base; // first synapse of the destination core
    unsigned long long a_beg;    // first axon of the slice (index into ax_proc_delay)
    double ain_lat;              // AxonInUnit::latency_spike_message of the core
    double slice_lat;            // per-event latency of the slice's latency class
    uint32_t n_ax, nbase, ncount, chunk0;
    uint32_t slice_id;           // index of the slice in core order (slice_proc, core_slice_beg); descriptors are in LAUNCH order
    uint8_t mode;                // 0 wide, 1 compact axon records, 2 bitmap records (a_beg = first 32-slot word of the slice's
                                 // source windows, n_ax = 256 x windows; see DevImage)
    uint8_t inkind;              // input kind of the core's neurons (the buffer position belongs to the core)
    uint8_t shared;              // the core has more than one slice: write back with atomics
    uint8_t pad;
};
static_assert(sizeof(SliceDesc) == 64, "one scalar load");

struct DevImage
{
    uint32_t n_cores, n_slots, ring_slots, n_slices, n_input, slot_offset, n_global_slots, max_core_slots, delay_slots;
    uint32_t n_wgs;          // neuron workgroups
    uint32_t n_groups;       // ceil(n_cores / L1_CORES): level-1 reduction groups
    uint32_t n_reduce_wgs;   // ceil(n_groups / 4): leading workgroups of a neuron launch that reduce earlier steps
    uint32_t n_soma_classes, n_cost_classes;
    int has_lif;             // some slot runs the LIF soma (its input current `icur` is state)
    int spike_energy;        // bit 0/1/2: some neuron's spike costs synapse / network / message-side dendrite energy
    int any_refrac;          // some soma class has refractory_delay > 0 (otherwise `refrac` is never touched)
    int uni_costing;         // every neuron carries the cost class `uni_cost` (UniformSoma chips): the neuron launch leaves
                             // COUNTS per wavefront and level 1 of the step reduction prices them once per core, instead of
                             // a dozen fp64 operations per wavefront (src/pipeline.hpp:574-731)
    sanafe_hip_cost_class uni_cost;
    double sync_delay;
    const WgDesc *wg_desc;          // [n_wgs]
    const uint32_t *core_wg_beg;    // [n_cores + 1] neuron workgroups of each core
    const uint32_t *core_nbase;
    const double *core_axon_out_latency;
    const sanafe_hip_soma_class *soma_classes;
    const sanafe_hip_cost_class *cost_classes;
    const uint32_t *slot_cls, *slot_aux;
    const SpikeStatic *slot_spike;
    const double *slot_bias;
    const uint32_t *in_train_beg, *in_train_len, *in_train_bits;
    const long long *in_rate_period;
    const uint32_t *slot_ext; // column of the slot in a row of external stream values (0xffffffff: none)
    uint32_t n_ext;
    uint32_t n_taps;          // neurons behind a `taps` dendrite (SANAFE_IN_TAPS); tables indexed by slot_aux
    const uint32_t *tap_slot, *tap_count;
    const double *tap_tc, *tap_sc; // [n_taps][8]
    const SliceDesc *slice_desc;    // [n_slices]
    const unsigned long long *core_syn_base;
    const uint32_t *core_slice_beg; // [n_cores+1]
    // Device layout of the inbound axons, chosen per delivery slice (slice_mode):
    //   wide    8 bytes/axon: bits 0-31 pre-synaptic GLOBAL slot | 32-47 synapse count | 48-55 latency class
    //           (255: read ax_proc_delay)
    //   compact 2 bytes/axon: bits 0-7 pre slot minus the previous axon's pre slot (0 for the first axon of a
    //           256-axon chunk, whose pre slot is chunk_pre0) | 8-15 synapse count; one latency class per slice.
    //           Used when the slice's axons are dense in pre-slot order (gaps < 256), have < 256 synapses each
    //           and share a latency class -- the normal case of a large recurrent network.
    //   bitmap  (format 7 only) the slice's source space cut into 256-slot windows (= its chunks): one bit per source
    //           slot (set: this core has an axon from that neuron; the axon order IS ascending source slot), then one byte
    //           per axon with its synapse count.  "Which axons spiked" becomes an AND with the spike bitmap, 32 axons per
    //           instruction.  Used when the slice's axons are compact-eligible, own their synapses (no lost charge) and fill
    //           at least a quarter of the source span -- large recurrent networks: every core hears from most neurons.
    const unsigned char *ax_bytes;            // all slices' records, each slice 16-byte aligned
    const uint32_t *chunk_syn0;               // per 256-axon chunk: first synapse (relative to the core)
    const uint32_t *chunk_pre0;               // per 256-axon chunk: pre slot of its first axon
    const double *ax_proc_delay;  // exact processing delays, only dereferenced for latency class 255
    const double *lat_class;      // [256] per-event latency of each class
    // Synapses, one of three formats (chip-wide):
    //   0: 4 bytes      axon code (11b) | accumulator index (13b) << 11 | int8 weight << 24
    //                   axon code = (chunk of the slice & 7) << 8 | index of the synapse's axon inside its 256-axon chunk: lets a chunk with many
    //                   spikes be STREAMED (every synapse word read once, in order, fired or not decided from the
    //                   word itself via a 256-byte table in LDS) instead of gathered;
    //                   accumulator index = delay * (npad + 1) + post-neuron offset, i.e. the LDS entry the weight is
    //                   added to; synapses whose charge is lost point at the trash entry `npad` of row 0
    //   1: 4 bytes      post (16b) | delay (3b) << 16 | drop << 19 | 12-bit signed weight << 20   (gather only)
    //   2: 4 + 8 bytes  syn_meta as in 1 without the weight, syn_weight = fp64                    (gather only)
    //   3: 4 bytes      axon code (8b) | accumulator index (12b) << 8 | 12-bit signed weight << 20: the streamable
    //                   form of 1, for chips whose cores need at most 4096 accumulators
    //   4: 4 + 8 bytes  axon code (11b) | accumulator index (15b) << 11, syn_weight = fp64: the streamable form of 2
    //   6: 2 bytes      first synapse of its axon (1b) | weight code (5b) << 1 | accumulator index (10b) << 6:
    //                   dictionary-coded weights (weight_lut, at most 32 distinct values on the chip) for cores with at
    //                   most 1024 accumulators.  No axon code: the words of a chunk are in axon order, so a word's axon
    //                   is the number of "first synapse" bits up to it (a per-lane popcount + one wave prefix sum per
    //                   16 bytes = 8 words).  Half the bytes of format 0 on the networks the benchmark recipe builds.
    //   7: 2 bytes      the words of 6 for dictionaries of INTEGERS: the LDS accumulators are 32-bit integers (an LDS
    //                   integer add costs a third of an fp64 one, profiles/micro/lds_ops.hip) and the sum of integers is
    //                   exact in any order, so the result equals the fp64 sum bit for bit.  Every event adds
    //                   weight + 2^acc_shift: a touched accumulator is never 0 (the buffer "holds a value, even a zero
    //                   one") and count and sum separate again at write-back; the host proves per slice and accumulator
    //                   that |sum| < 2^(acc_shift-1) and (count + 1) * 2^acc_shift <= 2^32 before it picks this format.
    // Formats 0, 3, 4, 6 and 7 give every 256-axon chunk a 16-byte aligned, padded run of words (stream layout).
    const uint32_t *syn_meta;     // padded by 256 words so the streaming loads may run past the end
    const double *syn_weight;
    const double *weight_lut;     // [32] formats 6, 7: the chip's distinct weight values
    int syn_format;               // 0 .. 4, 6, 7 as above
    int acc_shift;                // format 7: every event adds weight + 2^acc_shift
    int has_last;                 // some cores keep only the last event's current (SANAFE_IN_LAST)
    uint32_t bitmap_run_len;      // bitmap axon records: 256-slot windows a wavefront streams in one go (<= 8), chosen so that a
                                  // run's 16-byte groups fill whole 64-lane rows (the last row of a run is processed in full
                                  // however few of its lanes hold a group)
    // Push delivery for steps with FEW spikes (C4-like activity: 0.3 % of the neurons fire): the neuron launch delivers the
    // spikes itself -- the wavefront that updated a 64-slot chunk walks the static out-synapse lists of the neurons that fired
    // and adds their weights to the NEXT step's row of the time-step buffer (push chips keep two rows, so no wavefront of the
    // launch still reads what another one adds to) -- and the delivery launch of the step, which would probe every inbound
    // axon of the chip, is not launched at all.  Which path a step takes is decided ON THE HOST, deterministically, from the
    // synaptic events of the step DECISION_LAG before it: reduce_l2 publishes every step's event count in a pinned ring
    // (DevState::host_events), the launch loop reads the entry of step s - DECISION_LAG before it launches step s (it waits
    // for it if it has to: the host then runs at most DECISION_LAG - 2 steps ahead of the device) and passes the mode to the
    // neuron launch (StepArgs::pushed) and to level 1 of the step's reduction (PendStep::pushed).
    // Only built for chips where the result cannot depend on the order of the additions and the per-core message costs are
    // integers times a constant: integer weights, no synaptic delays / last-event cores / taps / host units, one latency
    // class per core, ring_slots >= 2.  On a tile-sharded chip the neuron launch pushes the rank's own spikes and
    // remote_push_kernel, after the all-gather, those of the other ranks.  push_cap == 0: not built.
    uint32_t push_cap;            // 0: not built; else the number of 64-slot neuron chunks
    uint32_t push_always;         // 1: every step is pushed and the chip has NO delivery launch (chips with so few synapses per
                                  // neuron that even a step in which every neuron fires costs about what one probe of all
                                  // inbound axons costs: C4 has one synapse per neuron)
    uint32_t push_max_events;     // a step is pushed when the step DECISION_LAG before it caused at most this many synaptic
                                  // events (the prediction only picks the faster path: both paths are exact for any activity)
    const uint32_t *push_ptr;     // [n_global_slots + 1] out-synapses INTO THIS CHIP of each neuron, by global slot
    const struct PushEntry *push_syn; // post slot, destination core | first-synapse-of-its-axon flag, weight
    const double *core_ain_lat;   // [n_cores] axon-in latency per message
    const double *core_event_lat; // [n_cores] latency per synaptic event (one latency class per core)
    // Ordered delivery (syn_format 8, non-integer weights): the synapses regrouped PER ACCUMULATOR (post neuron x delay
    // value), each list in the reference's delivery order; 64 lists side by side make one group (ordered_deliver_kernel).
    uint32_t ord_groups, ord_wgs; // groups of 64 accumulators; workgroups of the launch (4 wavefronts: a group each, then slices)
    uint32_t ord_walk_slices;     // n_slices (0: profiling without the processing-delay walk)
    const struct OrdGroup *ord_group;   // [ord_groups], longest lists first
    const uint32_t *ord_lane_slot;      // [ord_groups * 64] local slot of the lane's post neuron, 0xffffffff: none
    const uint8_t *ord_lane_delay;      // [ord_groups * 64] delay value (accumulator row) of the lane's list
    const uint32_t *ord_pre;            // entries [rows][64] per group: GLOBAL pre slot (| weight code << 27 with a dictionary);
                                        // padding entries point at bit n_global_slots, which never fires
    const double *ord_w;                // the entries' weights (no dictionary), same indexing
    int ord_dict;                       // weights are 5-bit codes into weight_lut
    // Event-driven delivery (event_deliver_kernel): the reference touches only the synapses behind the messages that arrived
    // (src/chip.cpp:738-764).  A second copy of the format-7 words, regrouped by SOURCE neuron: the destination cores are cut
    // into groups whose accumulators fit one workgroup's LDS (EvGroup), and the words a neuron sends into a group lie in one
    // contiguous, 16-byte aligned BLOCK; the blocks of a NEURON lie group after group (one ~5 KB region per neuron that all
    // the groups' workgroups read at about the same time: a 128-byte line comes out of HBM once, the Infinity Cache serves
    // the other XCDs).  A step with few spikes then reads the spike bitmap, per (fired neuron, group) two adjacent 8-byte
    // table entries, and that neuron's blocks -- work and bytes in proportion to the step's synaptic events instead of to
    // the chip's synapses.  Which kernel delivers a step is decided on the device
    // by the host like push delivery (push_max_events); both are exact for any activity.  ev_groups == 0: not built.
    uint32_t ev_groups;           // core groups
    uint32_t ev_segments;         // the source space is cut into this many segments of ev_seg_tiles tiles: grid = groups x segments
    uint32_t ev_seg_tiles;        // 1,024-slot tiles of the global source space per segment (<= 64: list entries are 16 bits)
    uint32_t ev_tiles;            // tiles in all: n_global_slots / 1024, rounded up
    uint32_t ev_always;           // 1: every step is delivered by events, the streaming kernel is never launched (tests)
    int ev_shift;                 // every event adds weight + 2^ev_shift; bounds proven per (segment, accumulator)
    const struct EvGroup *ev_group;   // [ev_groups]
    const unsigned long long *ev_meta_n; // the same entries [n_global_slots][ev_groups] (steps in which few neurons fire)
    const unsigned long long *ev_meta; // [ev_groups][n_global_slots]: bits 0-31 first 16-byte unit of block (neuron, group),
                                  // bits 32-47 its length in units, bits 48-63 which cores of the group the neuron reaches
                                  // (one message each).  Group-major: a workgroup reads its group's entries front to back.
    const uint16_t *ev_words;     // weight code (ev_code_bits) | accumulator index << ev_code_bits; padding words add into the
                                  // trash entries behind the group's accumulators
    const uint32_t *ev_chunk_core; // [n_slots / 64] local core of each 64-slot chunk
    const double *ev_lut;         // [32] the chip's distinct weight values, densely coded
    uint32_t ev_code_bits;        // 4 (<= 16 distinct weights: 4,096 accumulators per group) or 5 (2,048)
    // Cores whose soma is part of the MESSAGE pipeline (sanafe_hip_image::msg_*): msgsoma_kernel walks, per post-synaptic
    // neuron, its inbound synapses in delivery order and updates the soma once per synaptic event.
    uint32_t n_msg_cores, n_msg_chunks;
    const struct MsgCoreDev *msg_core_dev;  // [n_msg_cores]
    const uint32_t *msg_chunk_core;         // [n_msg_chunks] 64-slot chunk -> msg core
    const uint32_t *msg_chunk_slot0;        // [n_msg_chunks] first local slot of the chunk
    const uint32_t *msg_ptr;                // [n_slots + 1] a neuron's inbound synapses in msg_pre / msg_w, delivery order
    const uint32_t *msg_pre;                // GLOBAL slot of the synapse's source neuron
    const uint32_t *msg_ax;                 // the synapse's axon, numbered over all such cores (sanafe_hip_image::msg_ax_*)
    const double *msg_w;
    uint32_t n_msg_axons;
    const uint32_t *msg_ax_pre;             // the cores' inbound axons (message counting), core after core
};
struct MsgCoreDev
{
    uint32_t core, first_chunk, ax_beg, ax_end;
    sanafe_hip_msg_core_costs costs;
};
// One group of destination cores of the event layout: consecutive cores whose slots (each core padded to 64) span at most
// 2^(16 - code bits) - 64 accumulators; accumulator i of the group is local slot slot0 + i.
struct EvGroup
{
    uint32_t core0, n_cores; // <= 16 cores: one bit each in ev_meta
    uint32_t slot0, n_acc;   // n_acc: a multiple of 64
};
struct PushEntry
{
    uint32_t post;  // local slot of the post-synaptic neuron
    uint32_t core;  // destination core | 0x80000000: first synapse of its axon (= one message)
    double w;
};
// One group of the ordered layout: 64 accumulators whose lists lie side by side, entry k of lane l at off + 64 k + l.
struct OrdGroup
{
    unsigned long long off; // first entry
    uint32_t rows;          // entries per lane, padded to a multiple of ORD_UNROLL (dictionary: ORD_DICT_ROWS); with a dictionary
                            // entry k of lane l sits at off + 256 (k / 4) + 4 l + k % 4 (four rows per 16-byte load)
    uint32_t pad;
};
constexpr int ORD_UNROLL = 8;                         // loads in flight per lane while the previous ORD_UNROLL are folded
constexpr uint32_t ORD_DICT_ROWS = 4 * ORD_UNROLL;    // dictionary entries: four rows per 16-byte load
constexpr uint32_t ORD_PRE_BITS = 27; // dictionary entries: pre slot in 27 bits, weight code above

// The reduction of a step is split in two levels that ride in the leading workgroups of LATER neuron launches,
// so a timestep stays two launches and no launch waits on a serial reduction:
//   level 1 (step s, inside the neuron launch of step s+1, one wavefront per L1_CORES cores, four lanes per core): per core, the neuron
//           workgroups' partials and the delivery slices' processing delays are summed in a fixed order, then
//           folded over the 64 cores -> GroupPart[s & 1][group];
//   level 2 (step s, inside the neuron launch of step s+2, one wavefront): the groups -> Timestep totals,
//           simple timing model, run totals, records, t += 1.
// Partials live in two halves by step parity.  reduce_kernel flushes what is pending before any read-back.
struct PendStep
{
    int valid;          // 0: nothing to do
    int simple_timing, record, parity;
    int push_buf;        // step number % 3: the per-core push counters of that step
    int pushed;          // the step was delivered by the push path / the event kernel: level 1 prices the per-core counters
    long long rec_index; // record slot of that step
};

struct DevState
{
    double *v, *icur;
    int *refrac;
    uint8_t *status;
    uint32_t *in_pos;
    double *ring;          // [ring_slots][n_slots]
    uint8_t *ring_valid;   // [ring_slots][n_slots]
    uint8_t *arrived;      // [n_slots], SANAFE_IN_GATED / _TAPS neurons: an event reached the neuron in this step
    double *tap_v;         // [n_taps][8] tap voltages
    double *tap_in;        // [n_taps][8] charge delivered to each tap in the current step
    uint32_t *ring_last;   // [n_slots], SANAFE_IN_LAST cores: 1 + position (in the core's synapses) of the last event
    uint32_t *bits_local;  // [n_slots/32]; multi-GPU: this chip's window INSIDE bits_global
    uint32_t *bits_global; // [n_global_slots/32] (== bits_local on one GPU)
    WgPart *wg_part;       // [2][n_wgs][PARTS_PER_WG], by step parity
    double *slice_proc;    // [2][n_slices], by step parity: processing-delay sum of each delivery slice
    GroupPart *group_part; // [2][n_groups], by step parity
    long long *t;          // timesteps simulated so far
    long long *rec;        // records written so far in this sim
    sanafe_hip_totals *run;       // run totals
    sanafe_hip_totals *step_log;  // [log_cap]
    uint32_t *spike_log;          // [log_cap][n_slots/32]
    uint8_t *status_log;          // [log_cap][n_slots] NeuronStatus per step (record & 2), or NULL
    long long log_cap;
    // push / event delivery, by step number % 3
    uint32_t *push_core_cnt;      // [3][n_cores][2]: messages, events delivered to each core by the push path / the event kernel
    long long *host_events;       // pinned host memory, [HOST_EVENT_RING][2]: synaptic events and Timestep::timestep of the steps
                                  // reduced last (written by reduce_l2, read by the launch loop: the push / pull decision), or NULL
    double *host_proc;            // [2][n_cores] by step parity: message-processing delay of cores that run on the host, or NULL
    double *delay_log;            // [delay_log_cap] largest per-core delay of each step (multi-GPU simple timing), or NULL
    long long delay_log_cap;
    // event-driven delivery (DevImage::ev_*): what a step delivered by events leaves for the NEXT step's neuron launch
    uint32_t *msg_cnt;            // [n_msg_cores][4]: synaptic events, soma updates that fired, messages of the step (msgsoma_kernel)
    uint32_t *msg_ax_fired;       // [n_msg_axons] soma updates that fired, per message (= inbound axon) of the step; cleared by
                                  // msgsoma_finish_kernel
    uint16_t *msg_fired_log;      // [log_cap][n_msg_axons] the same per recorded step (record bit 1: status log), or NULL
    uint32_t *ev_part;            // [EV_MAX_SEGMENTS][n_slots]: per segment of the source space and neuron, count * 2^ev_shift + sum
                                  // of the weights that arrived (0: nothing); every workgroup stores all of its accumulators
};
constexpr uint32_t EV_MAX_SEGMENTS = 8;
constexpr long long DECISION_LAG = 16;    // step s is pushed / delivered by events when step s - DECISION_LAG caused few events
constexpr long long DECISION_STRIDE = 4;  // ... rounded down to a multiple of this: only every fourth step publishes its count
                                          // (a store across PCIe costs the launch it rides in ~1 us on the small configurations)
constexpr long long HOST_EVENT_RING = 64; // entries of DevState::host_events (the host runs < DECISION_LAG steps ahead)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Sum over the 64 lanes, returned in every lane.  DPP moves instead of LDS-crossbar shuffles: Hillis-Steele inside
// each 16-lane row, row_bcast:15 and row_bcast:31 across rows, total in lane 63.  Lanes without a source read
// 0 bits = +0.0 / 0.  Fixed order.
#define SANAFE_DPP_STEPS(STEP)                  \
    STEP(0x111, 0xf) /* row_shr:1 */            \
    STEP(0x112, 0xf) /* row_shr:2 */            \
    STEP(0x114, 0xf) /* row_shr:4 */            \
    STEP(0x118, 0xf) /* row_shr:8 */            \
    STEP(0x142, 0xa) /* row_bcast:15 */         \
    STEP(0x143, 0xc) /* row_bcast:31 */
__device__ __forceinline__ double wave_sum(double x)
{
#define SANAFE_STEP(CTRL, ROWS)                                                                   \
    {                                                                                             \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xf, false); \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xf, false); \
        x += __hiloint2double(hi, lo);                                                            \
    }
    SANAFE_DPP_STEPS(SANAFE_STEP)
#undef SANAFE_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
__device__ __forceinline__ long long wave_sum(long long x)
{
#define SANAFE_STEP(CTRL, ROWS)                                                                            \
    {                                                                                                      \
        const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) x, CTRL, ROWS, 0xf, false);          \
        const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) ((unsigned long long) x >> 32), CTRL, ROWS, 0xf, false); \
        x += (long long) (((unsigned long long) hi << 32) | lo);                                           \
    }
    SANAFE_DPP_STEPS(SANAFE_STEP)
#undef SANAFE_STEP
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) x, 63);
    const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) ((unsigned long long) x >> 32), 63);
    return (long long) (((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ double wave_max(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, WAVE));
    return x;
}

// static_cast<int>(double) as x86-64 performs it (cvttsd2si): out-of-range and NaN give INT_MIN.
// The reference quantises with it (src/models.cpp:447-455).
__device__ __forceinline__ int cvt_int_x86(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int) 0x80000000;
    return (int) x;
}

// Weight of the synapse at position `pos` of the device layout, whatever the format (slow paths only).
__device__ __forceinline__ double synapse_weight_at(const DevImage &im, unsigned long long pos)
{
    if (im.syn_format == 2 || im.syn_format == 4) return im.syn_weight[pos];
    if (im.syn_format == 6 || im.syn_format == 7) return im.weight_lut[(reinterpret_cast<const uint16_t *>(im.syn_meta)[pos] >> 1) & 31u];
    return (double) ((int) im.syn_meta[pos] >> (im.syn_format == 0 ? 24 : 20));
}

__device__ void reduce_l1(const DevImage &im, const DevState &st, int parity, uint32_t group, int push_buf, int pushed);
__device__ void reduce_l2(const DevImage &im, const DevState &st, const PendStep &prev);

// Per-launch values the host works out (no 64-bit division or row arithmetic on the device).
struct StepArgs
{
    double *ring;        // time-step buffer / delay-ring row of this step: st.ring + (t % ring_slots) * n_slots
    uint8_t *rvalid;     // the matching row of ring_valid
    uint32_t *slog;      // spike-log row of this step, or NULL
    uint8_t *stlog;      // status-log row of this step, or NULL
    const int *ext_row;  // external stream values of this step, or NULL
    long long t;         // Timestep::timestep of this step (steps simulated before it + 1)
    int parity;          // (t - 1) & 1: which half of the partials this step writes
    int push_buf;        // (t - 1) % 3: the per-core push counters of this step
    int pushed;          // this step's spikes are pushed by this launch (chips with push tables; decided by the host)
    double *ring_next;   // the NEXT step's row of the time-step buffer (push chips: ring_slots >= 2), where pushed spikes land
    uint8_t *rvalid_next;
    const uint32_t *ev_part; // the previous step was delivered by events: DevState::ev_part (else NULL), n_slots entries per row
};

// A chip whose neurons all carry the same class word (one soma model, one parameter set, one cost class, one input
// kind: the large synthetic configurations) gets its parameters as kernel arguments, i.e. in scalar registers:
// the neuron kernel then loads no class word and no class table at all.
struct UniformSoma
{
    sanafe_hip_soma_class p;
    sanafe_hip_cost_class c;
    uint32_t cls;        // the one slot class word
    uint32_t ncount;     // neurons of every core that has any
    uint32_t cpc;        // 64-slot chunks of such a core
    uint32_t wpc_shift;  // log2(neuron workgroups per core)
    uint32_t bias_uniform; // every live slot carries the bias `bias`: the per-slot array is not read (C4: no biases at all)
    double bias;
};

// Push delivery of the spikes of up to 64 source neurons (one per lane: fired_mask; lanes of fired neurons hold the bounds
// of their out-synapse lists in push_b / push_e): neuron by neuron, the lanes over its out-synapses (integer-valued weights:
// exact in any order), into the NEXT step's row of the time-step buffer.  The per-core message / event counters are bumped
// once per RUN of lanes with the same destination core (a neuron's synapses lie core by core), not once per synapse:
// thousands of same-address atomics would serialise at ~2.5 ns each.  Used by the neuron launch for its own chunk and, on
// tile-sharded chips, by remote_push_kernel for the neurons of the other ranks.
__device__ __forceinline__ void push_walk(const DevImage &im, uint32_t *cnt, double *ring_next, uint8_t *rvalid_next, unsigned long long fired_mask,
        uint32_t push_b, uint32_t push_e, uint32_t lane)
{
    for (unsigned long long m = fired_mask; m != 0ull; m &= m - 1ull)
    {
        const int j = __ffsll((long long) m) - 1;
        const uint32_t bi = (uint32_t) __builtin_amdgcn_readlane((int) push_b, j), ei = (uint32_t) __builtin_amdgcn_readlane((int) push_e, j);
        for (uint32_t k0 = bi; k0 < ei; k0 += WAVE) // (wave-uniform bounds)
        {
            const uint32_t k = k0 + lane;
            const bool act = k < ei; // lanes 0 .. n - 1
            uint32_t pc = 0xffffffffu;
            bool first = false;
            if (act)
            {
                const PushEntry pe = im.push_syn[k];
                atomicAdd(&ring_next[pe.post], pe.w);
                rvalid_next[pe.post] = 1;
                pc = pe.core & 0x7fffffffu;
                first = (pe.core >> 31) != 0u;
            }
            const uint32_t before = (uint32_t) __shfl_up((int) pc, 1, WAVE);
            const bool head = act && (lane == 0u || before != pc);
            const unsigned long long heads = __ballot(head), firsts = __ballot(first);
            if (head)
            {
                const unsigned long long later = heads & ~((2ull << lane) - 1ull); // (lane 63: the shift wraps to 0 - 1: no later head)
                const uint32_t n_act = min(ei - k0, (uint32_t) WAVE);
                const uint32_t end = (lane < 63u && later != 0ull) ? (uint32_t) __ffsll((long long) later) - 1u : n_act;
                const unsigned long long run = (end >= 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
                atomicAdd(&cnt[pc * 2u + 1u], end - lane);
                const uint32_t n_msgs = (uint32_t) __popcll(firsts & run);
                if (n_msgs != 0u) atomicAdd(&cnt[pc * 2u], n_msgs);
            }
        }
    }
}

// Tile-sharded chips with push tables: the spikes of the OTHER ranks' neurons, pushed by the receiving rank after the
// all-gather of the spike bitmap (the neuron launch has pushed this rank's own).  grid = ceil(n_global_slots / 2048), block =
// 64: a wavefront scans 64 words of the global bitmap (2,048 source slots; the local window reads as silent) and walks the
// out-synapse lists -- into THIS rank's neurons -- of the neurons that fired.  Nothing else touches the bitmap: a step with
// few spikes costs the scan (n_global_slots / 8 bytes) however many ranks the chip is cut into.
__global__ void __launch_bounds__(WAVE) remote_push_kernel(DevImage im, DevState st, double *ring_next, uint8_t *rvalid_next, int push_buf)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t wi = blockIdx.x * WAVE + lane; // word of the global bitmap
    const uint32_t local0 = im.slot_offset / 32u, local1 = (im.slot_offset + im.n_slots) / 32u;
    uint32_t w = 0;
    if (wi < im.n_global_slots / 32u && !(wi >= local0 && wi < local1)) w = st.bits_global[wi];
    uint32_t *cnt = st.push_core_cnt + (size_t) push_buf * im.n_cores * 2u;
    for (unsigned long long words = __ballot(w != 0u); words != 0ull; words &= words - 1ull) // (wave-uniform)
    {
        const int l = __ffsll((long long) words) - 1;
        const uint32_t ww = (uint32_t) __builtin_amdgcn_readlane((int) w, l);
        const uint32_t slot = (blockIdx.x * WAVE + (uint32_t) l) * 32u + lane; // lanes 0 .. 31: the word's source slots
        uint32_t pb = 0, pe = 0;
        if (lane < 32u && ((ww >> lane) & 1u))
        {
            pb = im.push_ptr[slot];
            pe = im.push_ptr[slot + 1u];
        }
        push_walk(im, cnt, ring_next, rvalid_next, (unsigned long long) ww, pb, pe, lane);
    }
}

// ---------------------------------------------------------------------------------------
// K1: neuron update.  grid = n_reduce_wgs + n_wgs, block = 256.
//   leading workgroups: level 1 / level 2 of the step reduction of the two previous steps (one wavefront per
//                       group of 64 cores), independent of everything else in the launch;
//   neuron workgroups:  one wavefront per 64-slot chunk (no loop), so the spike ballot maps 1:1 to bitmap words.
// The kernel is bound by vector-instruction issue before it is bound by HBM (16 wavefronts per SIMD on a 1 M
// neuron chip, fp64 at 8 cycles an instruction), so everything wave-uniform is kept in scalar registers: the
// per-wave base pointers (every array is indexed by the lane alone), the counters and class costs (ballots and
// popcounts), and -- UNI = true -- the soma parameters themselves.  MODEL = 0 decides the soma model per lane,
// 1 / 2 compile the LIF / TrueNorth update alone.
// The load chain is two round trips deep: (1) the workgroup descriptor (arithmetic when UNI), (2) every per-slot
// array at once -- class word, time-step buffer value AND valid byte, bias, potential, input current, refractory
// counter -- while the soma / cost class tables are staged in LDS, so no load depends on the class word;
// lanes whose neuron fired add (3) one 40-byte record of what the spike causes downstream.
// ---------------------------------------------------------------------------------------
template <int MODEL, bool UNI>
__global__ void __launch_bounds__(NEURON_BLOCK) __attribute__((amdgpu_num_sgpr(SANAFE_NEURON_SGPRS))) SANAFE_NEURON_ATTR
neuron_kernel(DevImage im, DevState st, StepArgs sa, UniformSoma us, PendStep l1, PendStep l2)
{
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); // scalar
    if (blockIdx.x < im.n_reduce_wgs) // workgroup-uniform
    {
        if (blockIdx.x == 0 && wave == 0 && l2.valid) reduce_l2(im, st, l2);
        const uint32_t group = blockIdx.x * (NEURON_BLOCK / WAVE) + wave;
        if (l1.valid && group < im.n_groups) reduce_l1(im, st, l1.parity, group, l1.push_buf, l1.pushed);
        return;
    }
    __shared__ sanafe_hip_soma_class s_soma[UNI ? 1 : SOMA_LDS_MAX];
    __shared__ sanafe_hip_cost_class s_cost[UNI ? 1 : COST_LDS_MAX];
    const uint32_t wg = blockIdx.x - im.n_reduce_wgs;
    uint32_t slot0, nchunks, nlive = WAVE, core = 0xffffffffu;
    if (UNI)
    {
        const uint32_t k = wg >> us.wpc_shift, q = (wg & ((1u << us.wpc_shift) - 1u)) * 4u;
        slot0 = (k * us.cpc + q) * WAVE;
        nchunks = (us.cpc - q < 4u) ? us.cpc - q : 4u;
        const uint32_t first = (q + wave) * WAVE; // offset of this wave's chunk inside its core
        nlive = (first < us.ncount) ? ((us.ncount - first < (uint32_t) WAVE) ? us.ncount - first : (uint32_t) WAVE) : 0u;
    }
    else
    {
        const WgDesc wd = im.wg_desc[wg];
        slot0 = wd.slot0;
        nchunks = wd.nchunks;
        core = wd.core;
    }
    const bool soma_lds = UNI || im.n_soma_classes <= SOMA_LDS_MAX, cost_lds = UNI || im.n_cost_classes <= COST_LDS_MAX;
    const bool active = wave < nchunks; // wave-uniform
    const long long t = sa.t;
    const uint32_t c0 = slot0 + wave * WAVE; // first slot of this wave's chunk (scalar): arrays are indexed by the lane
    double *p_ring = sa.ring + c0, *p_v = st.v + c0, *p_icur = st.icur + c0;
    uint8_t *p_rvalid = sa.rvalid + c0;
    int *p_refrac = st.refrac + c0;
    const double *p_bias = im.slot_bias + c0;

    // ---- every per-slot load, issued at once (padding slots exist in all per-slot arrays) ----
    uint32_t cls = 0, ext_col = 0xffffffffu;
    uint8_t in_valid = 0;
    double in_value = 0.0, bias = 0.0, v_in = 0.0, ic_in = 0.0;
    int rc_in = 0;
    if (active)
    {
        if (UNI) cls = (lane < nlive) ? us.cls : 0u;
        else cls = im.slot_cls[c0 + lane];
        in_valid = p_rvalid[lane];
        in_value = p_ring[lane];
        bias = (UNI && us.bias_uniform) ? us.bias : p_bias[lane];
        v_in = p_v[lane];
        if (MODEL != SANAFE_SOMA_TRUENORTH && im.has_lif) ic_in = p_icur[lane];
        if (MODEL != SANAFE_SOMA_TRUENORTH && im.any_refrac) rc_in = p_refrac[lane];
        if (!UNI && sa.ext_row != nullptr) ext_col = im.slot_ext[c0 + lane];
    }
    // the PREVIOUS step was delivered by events (the host knows: sa.ev_part): what event_deliver_kernel left per segment of
    // the source space, loaded with everything else
    uint32_t evp[EV_MAX_SEGMENTS];
#pragma unroll
    for (uint32_t q = 0; q < EV_MAX_SEGMENTS; q++) evp[q] = 0u;
    // (not in the uniform TrueNorth instantiation, which sits at its 64-register budget: such chips get no event layout)
    constexpr bool EV_OK = !(UNI && MODEL == SANAFE_SOMA_TRUENORTH);
    const bool ev_in = EV_OK && sa.ev_part != nullptr; // (a kernel argument: no load behind it)
    if (ev_in && active)
    {
#pragma unroll
        for (uint32_t q = 0; q < EV_MAX_SEGMENTS; q++) evp[q] = sa.ev_part[(size_t) q * im.n_slots + c0 + lane];
    }
    // push or pull for this step: decided by the host (DevImage::push_*), a kernel argument
    const bool push_now = sa.pushed != 0;
    if (!UNI)
    {
        // ---- class tables -> LDS, in flight together with the slot loads ----
        static_assert(sizeof(sanafe_hip_soma_class) % 8 == 0 && sizeof(sanafe_hip_cost_class) % 8 == 0, "copied in 8-byte words");
        if (soma_lds)
        {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(im.soma_classes);
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(s_soma);
            const uint32_t n = im.n_soma_classes * (uint32_t) (sizeof(sanafe_hip_soma_class) / 8);
            for (uint32_t i = threadIdx.x; i < n; i += NEURON_BLOCK) dst[i] = src[i];
        }
        if (cost_lds)
        {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(im.cost_classes);
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(s_cost);
            const uint32_t n = im.n_cost_classes * (uint32_t) (sizeof(sanafe_hip_cost_class) / 8);
            for (uint32_t i = threadIdx.x; i < n; i += NEURON_BLOCK) dst[i] = src[i];
        }
        __syncthreads(); // class tables staged
    }

    double e_soma = 0.0, e_dend = 0.0, e_syn = 0.0, e_net = 0.0, lat = 0.0;
    long long n_pack = 0, n_hops = 0, n_ev = 0, n_upd = 0, n_fire = 0, n_counted = 0;
    if (active)
    {
        const uint32_t model = cls & 7u; // padding slots carry SANAFE_SOMA_NONE
        int status = 0;
        // buffer before axon_out: the neuron pipeline holds no unit; the status the message pipeline's soma calls left persists
        // and axon_out acts on it (src/mapped.cpp:168-188, src/chip.cpp:710-736)
        if (!UNI && model == SANAFE_SOMA_PERSIST) status = st.status[c0 + lane];
        if (model != SANAFE_SOMA_NONE && model != SANAFE_SOMA_HOST && (UNI || model != SANAFE_SOMA_PERSIST))
        {
            const uint32_t inkind = (cls >> 3) & 7u;
            // ---- synaptic input from the time-step buffer / delay ring (read, then clear) ----
            bool has_in;
            double cur;
            if (inkind == SANAFE_IN_ZERO)
            {
                has_in = true;
                cur = 0.0;
            }
            else if (inkind == SANAFE_IN_NONE)
            {
                // buffer inside the soma unit: the neuron loop calls the soma without an input; its synaptic input reaches
                // it per event in the message pipeline (msgsoma_kernel)
                has_in = false;
                cur = 0.0;
            }
            else if (inkind == SANAFE_IN_GATED)
            {
                // the delay line's matured charge reaches the soma only through the buffer an event of the
                // previous step wrote; unobserved charge is consumed all the same (the line keeps shifting)
                has_in = false;
                cur = 0.0;
                if (in_valid != 0)
                {
                    p_ring[lane] = 0.0;
                    p_rvalid[lane] = 0;
                }
                if (st.arrived[c0 + lane] != 0)
                {
                    st.arrived[c0 + lane] = 0;
                    has_in = in_valid != 0;
                    cur = has_in ? in_value : 0.0;
                }
            }
            else if (inkind == SANAFE_IN_LAST_DELAY)
            {
                // accumulator_with_delay called from the NEURON pipeline every step: the line shifts (the charge maturing
                // now goes to the soma, if any), then the one current the time-step buffer kept is added with the delay
                // of the unit's synapse address 0 and matures delay + 1 steps later (src/models.cpp:96-131)
                has_in = in_valid != 0;
                cur = has_in ? in_value : 0.0;
                if (has_in)
                {
                    p_ring[lane] = 0.0;
                    p_rvalid[lane] = 0;
                }
                const uint32_t last = st.ring_last[c0 + lane];
                if (last != 0u)
                {
                    const uint32_t cr = (core != 0xffffffffu) ? core : im.wg_desc[wg].core;
                    const unsigned long long pos = im.core_syn_base[cr] + (last - 1u);
                    const double w = synapse_weight_at(im, pos);
                    const uint32_t row = (uint32_t) ((t + 1 + (long long) im.slot_aux[c0 + lane]) % im.ring_slots);
                    const size_t at = (size_t) row * im.n_slots + c0 + lane;
                    st.ring[at] = (st.ring_valid[at] ? st.ring[at] : 0.0) + w;
                    st.ring_valid[at] = 1;
                    st.ring_last[c0 + lane] = 0u;
                }
            }
            else if (inkind == SANAFE_IN_LAST)
            {
                // the accumulator integrates the one current the time-step buffer kept, after its lazy clear
                // (src/models.cpp:71-94): 0.0 + w_last, or plain 0.0 -- a value either way
                has_in = true;
                cur = 0.0;
                const uint32_t last = st.ring_last[c0 + lane];
                if (last != 0u)
                {
                    const uint32_t cr = (core != 0xffffffffu) ? core : im.wg_desc[wg].core;
                    const unsigned long long pos = im.core_syn_base[cr] + (last - 1u);
                    const double w = synapse_weight_at(im, pos);
                    cur = 0.0 + w;
                    st.ring_last[c0 + lane] = 0u;
                }
            }
            else
            {
                has_in = in_valid != 0;
                cur = 0.0;
                if (has_in)
                {
                    cur = in_value;
                    p_ring[lane] = 0.0;
                    p_rvalid[lane] = 0;
                }
                if (ev_in)
                {
                    // count * 2^shift + sum per segment (bounds proven per segment and accumulator): decoded one by one;
                    // the sum of integers is the fp64 sum of the reference whatever the order
                    long long tot = 0;
                    uint32_t any = 0;
#pragma unroll
                    for (uint32_t q = 0; q < EV_MAX_SEGMENTS; q++)
                    {
                        const uint32_t n_ev = (evp[q] + (1u << (im.ev_shift - 1))) >> im.ev_shift;
                        tot += (long long) (int) (evp[q] - (n_ev << im.ev_shift));
                        any |= evp[q];
                    }
                    if (any != 0u)
                    {
                        has_in = true;
                        cur = (double) tot; // (the buffer row of this step is empty: no other launch delivered into it)
                    }
                }
            }
            // host-generated value of a sequential source this neuron consumes at every update
            // (Poisson draw, std::rand() & mask, noise file): include/sanafe_hip.h, slot_ext
            const bool has_ext = !UNI && ext_col != 0xffffffffu;
            const int ext = has_ext ? sa.ext_row[ext_col] : 0;
            sanafe_hip_soma_class p;
            if (UNI) p = us.p;
            else if (model == SANAFE_SOMA_INPUT) p = sanafe_hip_soma_class{};
            else if (soma_lds) p = s_soma[cls >> 16];
            else p = im.soma_classes[cls >> 16];
            if ((MODEL == 0 || MODEL == SANAFE_SOMA_LIF) && model == SANAFE_SOMA_LIF)
            {
                // LoihiLifModel::update, src/models.cpp:497-567
                double v = v_in;
                double ic = ic_in;
                int rc = rc_in;
                status = 1;
                if (fabs(v) > 0.0 || has_in || fabs(bias) > 0.0 || p.force_update) status = 2;
                if (t > 1)
                {
                    ic *= p.input_decay;
                    v *= p.leak_decay;
                }
                v = (double) cvt_int_x86(v * 64.0) / 64.0;
                if (has_ext) v += (double) ext; // loihi_generate_noise, src/models.cpp:535-539
                if (!(rc > 0))
                {
                    v += bias;
                    ic += has_in ? cur : 0.0;
                    v += ic;
                    bool fired = false;
                    if (v > p.threshold)
                    {
                        if (p.reset_mode == SANAFE_RESET_HARD) v = p.reset;
                        else if (p.reset_mode == SANAFE_RESET_SOFT) v -= p.threshold;
                        rc = p.refractory_delay;
                        fired = true;
                    }
                    if (v < p.reverse_threshold)
                    {
                        if (p.reverse_reset_mode == SANAFE_RESET_SOFT) v -= p.reverse_threshold;
                        else if (p.reverse_reset_mode == SANAFE_RESET_HARD) v = p.reverse_reset;
                        else if (p.reverse_reset_mode == SANAFE_RESET_SATURATE) v = p.reverse_threshold;
                    }
                    if (fired) status = 3;
                }
                rc = rc - 1 > 0 ? rc - 1 : 0;
                p_v[lane] = v;
                p_icur[lane] = ic;
                if (im.any_refrac) p_refrac[lane] = rc;
            }
            else if ((MODEL == 0 || MODEL == SANAFE_SOMA_TRUENORTH) && model == SANAFE_SOMA_TRUENORTH)
            {
                // TrueNorthModel::update, src/models.cpp:724-830
                double v = v_in;
                status = 1;
                if (fabs(v) > 0.0 || has_in || fabs(bias) > 0.0 || p.force_update) status = 2;
                if (p.leak_towards_zero)
                {
                    if (v > 0.0) v -= p.leak_decay;
                    else if (v < 0.0) v += p.leak_decay;
                }
                else
                {
                    v += p.leak_decay;
                }
                v += bias;
                if (has_in) v += cur;
                // the randomised threshold test sees V + (rand() & mask); the resets act on V (src/models.cpp:745-797)
                const double vt = has_ext ? v + (double) ext : v;
                if (vt >= p.threshold)
                {
                    if (p.reset_mode == SANAFE_RESET_HARD) v = p.reset;
                    else if (p.reset_mode == SANAFE_RESET_SOFT) v -= p.threshold;
                    else if (p.reset_mode == SANAFE_RESET_SATURATE) v = p.threshold;
                    status = 3;
                }
                else if (vt <= p.reverse_threshold)
                {
                    if (p.reverse_reset_mode == SANAFE_RESET_HARD) v = p.reverse_reset;
                    else if (p.reverse_reset_mode == SANAFE_RESET_SOFT) v += p.reverse_threshold;
                    else if (p.reverse_reset_mode == SANAFE_RESET_SATURATE) v = p.reverse_threshold;
                }
                p_v[lane] = v;
            }
            else if (MODEL == 0) // SANAFE_SOMA_INPUT: InputModel::update, src/models.cpp:863-903
            {
                const uint32_t a = im.slot_aux[c0 + lane];
                const uint32_t pos = st.in_pos[a];
                bool send = false;
                if (pos < im.in_train_len[a])
                {
                    const uint32_t b = im.in_train_beg[a] + pos;
                    send = (im.in_train_bits[b >> 5] >> (b & 31u)) & 1u;
                    st.in_pos[a] = pos + 1;
                }
                if (ext != 0) send = true; // poisson_probability > uniform_distribution(gen)
                const long long period = im.in_rate_period[a];
                if (period > 0 && (t % period) == 0) send = true;
                status = send ? 3 : 1;
            }
        }
        // SANAFE_SOMA_HOST slots (plugin units) are evaluated by the host between
        // step_neurons and step_deliver; host_status_kernel sets their status and spike bits.
        const bool live = model != SANAFE_SOMA_NONE;
        if (live && model != SANAFE_SOMA_HOST && (UNI || model != SANAFE_SOMA_PERSIST)) st.status[c0 + lane] = (uint8_t) status;
        if (sa.stlog != nullptr && live) sa.stlog[c0 + lane] = (uint8_t) status;
        const unsigned long long fired_mask = __ballot(status == 3);
        // push delivery (see DevImage::push_*): this step was decided to be one of few spikes -- the wavefront delivers its
        // own neurons' spikes right here, into the NEXT step's row of the time-step buffer (push chips keep two rows)
        uint32_t push_b = 0, push_e = 0;
        const bool pushing = push_now && fired_mask != 0ull; // wave-uniform
        if (pushing && status == 3)
        {
            push_b = im.push_ptr[im.slot_offset + c0 + lane]; // (indexed by GLOBAL source slot: tile-sharded chips push too)
            push_e = im.push_ptr[im.slot_offset + c0 + lane + 1];
        }
        if (lane == 0)
        {
            const uint32_t w = c0 >> 5;
            st.bits_local[w] = (uint32_t) fired_mask;
            st.bits_local[w + 1] = (uint32_t) (fired_mask >> 32);
            if (sa.slog != nullptr)
            {
                sa.slog[w] = (uint32_t) fired_mask;
                sa.slog[w + 1] = (uint32_t) (fired_mask >> 32);
            }
        }
        // ---- default costing, src/pipeline.hpp:574-731.  Counters and class costs are taken per wavefront from
        //      ballots; only what depends on the individual neuron -- the static totals of a spike, or costs when
        //      the 64 neurons do not share one cost class -- is summed per lane and reduced once. ----
        const bool sender = (model != SANAFE_SOMA_NONE && model != SANAFE_SOMA_HOST); // (host slots: host_status_kernel)
        const bool counted = sender && (UNI || model != SANAFE_SOMA_PERSIST);         // a soma call of the neuron loop
        const unsigned long long m_cnt = __ballot(counted);
        const unsigned long long fired_cnt = fired_mask & m_cnt;
        if (m_cnt != 0ull) // wave-uniform
        {
            const unsigned long long m_upd = __ballot(counted && status >= 2);
            n_upd = __popcll(m_upd);
            n_fire = __popcll(fired_cnt);
            const uint32_t ccid = (cls >> 6) & 1023u;
            const uint32_t cc0 = UNI ? 0u : (uint32_t) __builtin_amdgcn_readlane((int) ccid, __ffsll((long long) m_cnt) - 1);
            if (UNI)
            {
                n_counted = __popcll(m_cnt); // priced per core by reduce_l1 (DevImage::uni_costing)
            }
            else if (__ballot(counted && ccid != cc0) == 0ull)
            {
                sanafe_hip_cost_class c0c;
                if (cost_lds) c0c = s_cost[cc0];
                else c0c = im.cost_classes[cc0];
                const double n_all = (double) __popcll(m_cnt), n_f = (double) __popcll(fired_cnt),
                             n_u = (double) __popcll(m_upd & ~fired_cnt), n_i = (double) __popcll(m_cnt & ~m_upd);
                e_soma = (n_i * c0c.soma_energy[0] + n_u * c0c.soma_energy[1]) + n_f * c0c.soma_energy[2];
                e_dend = n_all * c0c.dendrite_energy;
                lat = n_all * (0.0 + c0c.dendrite_latency) + ((n_i * c0c.soma_latency[0] + n_u * c0c.soma_latency[1]) + n_f * c0c.soma_latency[2]);
            }
            else
            {
                double le_soma = 0.0, le_dend = 0.0, l_lat = 0.0;
                if (counted)
                {
                    // (static indices + selects: a runtime index into a by-value copy would go to scratch)
                    const sanafe_hip_cost_class cc = cost_lds ? s_cost[ccid] : im.cost_classes[ccid];
                    le_dend = cc.dendrite_energy;
                    le_soma = status == 1 ? cc.soma_energy[0] : status == 2 ? cc.soma_energy[1] : cc.soma_energy[2];
                    l_lat = (0.0 + cc.dendrite_latency) +
                            (status == 1 ? cc.soma_latency[0] : status == 2 ? cc.soma_latency[1] : cc.soma_latency[2]);
                }
                e_soma = wave_sum(le_soma);
                e_dend = wave_sum(le_dend);
                lat = wave_sum(l_lat);
            }
        }
        if ((fired_mask & __ballot(sender)) != 0ull) // wave-uniform
        {
            // everything this spike causes downstream is static per neuron
            // (pipeline_process_axon_out, receive_message: src/chip.cpp:694-708, 802-834)
            SpikeStatic ss{};
            if (status == 3 && sender) ss = im.slot_spike[c0 + lane];
            // messages and hops of a chunk fit 24 + 40 bits: one integer reduction for both
            const long long ph = wave_sum((long long) (((unsigned long long) ss.packets << 40) | (unsigned long long) ss.hops));
            n_pack = (long long) ((unsigned long long) ph >> 40);
            n_hops = (long long) ((unsigned long long) ph & ((1ull << 40) - 1ull));
            n_ev = wave_sum((long long) ss.events);
            if (im.spike_energy & 1) e_syn = wave_sum(ss.e_syn);
            if (im.spike_energy & 2) e_net = wave_sum(ss.e_net);
            if (im.spike_energy & 4) e_dend += wave_sum(ss.e_dend);
        }
        if (pushing) push_walk(im, st.push_core_cnt + (size_t) sa.push_buf * im.n_cores * 2u, sa.ring_next, sa.rvalid_next, fired_mask, push_b, push_e, lane);
    }
    if (lane == 0)
    {
        // every wavefront leaves its own partial (no LDS, no workgroup barrier: a wavefront retires as soon as its chunk is
        // done); reduce_l1 adds a core's partials in a fixed order.  Wavefronts without a chunk write zeros.
        WgPart part;
        part.e_soma = e_soma;
        part.e_dend = e_dend;
        part.e_syn = e_syn;
        part.e_net = e_net;
        part.lat = lat;
        part.updated = n_upd;
        part.fired = n_fire;
        part.packets = n_pack;
        part.hops = n_hops;
        part.events = n_ev;
        part.counted = n_counted;
        part.pad = 0;
        st.wg_part[((size_t) sa.parity * im.n_wgs + wg) * PARTS_PER_WG + wave] = part;
    }
}

// ---------------------------------------------------------------------------------------
// K2: spike delivery.  grid = n_slices, block = 256 (4 independent wavefronts), dynamic LDS
//
// The four waves of a workgroup share only the LDS accumulators of their destination core.
// Each wave walks its own interleaved 256-axon chunks of the slice with NO workgroup
// barrier inside the loop (scan -> ballot/prefix compaction -> expansion all stay inside the
// wave, ordered by LDS issue order), so 16-32 waves per CU keep independent chains of
// global loads in flight: the loop is a latency-bound gather, not a bandwidth-bound stream.
// ---------------------------------------------------------------------------------------
constexpr int WAVE_CHUNK = WAVE * AX_PER_THREAD; // axons one wave scans per iteration
constexpr int EXPAND_UNROLL = 4;
constexpr uint32_t HEAD_WINDOW = 2048; // events covered by one 64-word head bitmap
#ifndef SANAFE_STREAM_DEPTH
#define SANAFE_STREAM_DEPTH 4
#endif
#ifndef SANAFE_DELIVER_WAVES_PER_EU
#define SANAFE_DELIVER_WAVES_PER_EU 5
#endif
#ifndef SANAFE_BITMAP_WAVES_PER_EU
#define SANAFE_BITMAP_WAVES_PER_EU 6 // bitmap axon records need fewer registers: with 3 instead of 4 groups in flight per lane the
                                     // kernel fits 80 (one spilled dword) and a sixth wavefront per SIMD hides more of the
                                     // vector-issue stalls: +5 % on 1,024 x 256, +0..2 % on 512 x 512 (7 or 8 wavefronts spill
                                     // 4-12 dwords and lose 3-8 %)
#endif
#ifndef SANAFE_BITMAP_STREAM_DEPTH
#define SANAFE_BITMAP_STREAM_DEPTH 3
#endif
constexpr int STREAM_DEPTH = SANAFE_STREAM_DEPTH; // 16-byte groups per lane in flight in the stream path
#ifndef SANAFE_STREAM_MIN_ACTIVE_LANES
#define SANAFE_STREAM_MIN_ACTIVE_LANES 8
#endif
constexpr uint32_t STREAM_MIN_ACTIVE_LANES = SANAFE_STREAM_MIN_ACTIVE_LANES; // lanes (4 axons each) with a spiking axon from which streaming a chunk beats gathering
constexpr unsigned long long ACC_UNTOUCHED = 0x8000000000000000ull; // -0.0: no sum of additions yields it

// Inclusive prefix sum over the 64 lanes with DPP moves (no LDS round trips): Hillis-Steele inside each
// 16-lane row (row_shr 1/2/4/8), then row_bcast:15 carries a row's total into the next row (rows 1 and 3) and
// row_bcast:31 the first half's total into rows 2 and 3.  Lanes without a source read `old` = 0.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xf, 0xf, false); // row_shr:1
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return x;
}

// LDS traffic of ONE wave is executed in issue order, so a ds_read issued after a ds_write of
// the same wave sees it even across lanes; this only has to stop the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

extern __shared__ __align__(16) unsigned char deliver_lds[];

// The synapse words are read exactly once per launch: non-temporal loads (global_load_dwordx4 ... nt) keep them from
// displacing the spike bitmap and the axon records in the caches.
// Keeps the stream loads in program order: the scheduler would otherwise reorder independent loads, and the wait
// for "the oldest group" (vmcnt) would again cover all of them.
__device__ __forceinline__ void keep_load_order() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ uint4 load_stream16(const uint4 *p)
{
#ifdef SANAFE_STREAM_PLAIN_LOADS
    return *p;
#else
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

// LAST: the chip has cores whose time-step buffer sits before the dendrite unit (SANAFE_IN_LAST): for those cores the
// workgroup keeps, per post-synaptic neuron, the position of the LAST event in delivery order (LDS atomic max)
// instead of a sum -- the buffer holds one pipeline result per neuron and later events overwrite earlier ones
// (src/chip.cpp:738-764).  Compiled out (LAST = false) for every chip without such cores.
// BLOCK: 256 threads (4 wavefronts share a slice's chunks), or 64 on chips whose slices hold one or two chunks (TrueNorth:
// ~256 axons per core) -- most of the four wavefronts would idle and hold wave slots.
// BITMAP: every compact slice of the chip is on bitmap axon records (SliceDesc::mode 2; format 7 only) -- compiled apart from
// the 2-byte delta records so that neither phase A pays for the other's registers.  PUSH: the chip has push-delivery tables
// (DevImage::push_*) -- no longer used: the host decides per step and does not launch this kernel on pushed steps.
// SUB: bitmap records on cores of at most SUB_MAX_NEURONS neurons -- every accumulator is spread over 16 sub-accumulators, picked by
// the upper four bits of the word's weight code, 64 bytes per neuron in STATIC LDS: the accumulator's LDS address is then the
// word itself with its low two bits masked (one instruction per word instead of a bit-field extract and a shift-add: 70 -> 62
// vector instructions per 16-byte group), same-address and bank conflicts thin out, and the write-back adds the sixteen
// (count * 2^shift + sum; the sub-sums obey the bounds the host proved for the whole).
constexpr uint32_t SUB_MAX_NEURONS = 256;
template <int SYN_FMT, bool HAS_DELAY, bool LAST, bool IACC = false, int BLOCK = DELIVER_BLOCK, bool BITMAP = false, bool PUSH = false, bool SUB = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BITMAP ? SANAFE_BITMAP_WAVES_PER_EU : SANAFE_DELIVER_WAVES_PER_EU, 8)))
deliver_kernel(DevImage im, DevState st, long long done /* steps simulated before this one */, uint32_t first_slice)
{
    static_assert(!SUB || (BITMAP && !HAS_DELAY), "sub-accumulators: bitmap records without synaptic delays");
    __shared__ uint32_t s_sub[SUB ? (SUB_MAX_NEURONS + 1u) * 16u : 1u]; // [neuron | trash entry][code >> 1]
    __shared__ uint32_t s_beg[BLOCK / WAVE][WAVE_CHUNK];       // first synapse of each active axon
    __shared__ __align__(256) uint32_t s_pref[BLOCK / WAVE][WAVE];   // head bitmap of the event window / spiked-axon mask
    __shared__ double s_red[BLOCK / WAVE];
    __shared__ long long s_redi[BLOCK / WAVE];
    constexpr bool DICT16 = (SYN_FMT == 6 || SYN_FMT == 7); // 2-byte words, dictionary-coded weights
    constexpr bool INT_ACC = (SYN_FMT == 7) || IACC;        // 32-bit integer accumulators (see DevImage): format 7 always, formats 0 / 3 when the bounds hold
    static_assert(!IACC || SYN_FMT == 0 || SYN_FMT == 3, "integer accumulators: formats 0, 3 and 7");
    __shared__ double s_lut[(SYN_FMT == 6) ? 32 : 1];      // format 6: the weight dictionary
    __shared__ uint16_t s_lut16[(SYN_FMT == 7) ? 32 : 2];  // format 7: weight + 2^acc_shift (acc_shift <= 15)
    // Formats 6, 7: a wave takes RUN_MAX consecutive chunks, notes which axons spiked in a bit table and streams the
    // words of all of them in one go (a "run"): the words carry no chunk-relative axon code, so nothing ties the
    // stream to 256 axons, and the fixed cost of starting a stream is paid once per run.
    // Formats 0, 3 and 4 take runs as well, with a byte table of 8 x 256 entries: the words of 0 and 4 carry an 11-bit axon
    // code = (chunk of the slice & 7) << 8 | axon of the chunk; format 3 has room for the 8-bit axon only and takes the
    // chunk from the word's POSITION (a chunk's words are 16-byte aligned: a group belongs to one chunk).
    constexpr bool RUNS = DICT16 || SYN_FMT == 0 || SYN_FMT == 3 || SYN_FMT == 4;
    constexpr bool CODE11 = (SYN_FMT == 0 || SYN_FMT == 4);
    constexpr bool BYTE_TABLE = RUNS && !DICT16;
    constexpr uint32_t RUN_MAX = 8;
    // Slices on BITMAP axon records (SliceDesc::mode 2): format 7 without last-event cores (see phase A below)
    constexpr bool BITMAP_RECORDS = BITMAP;
    static_assert(!BITMAP || (SYN_FMT == 7 && !LAST), "bitmap axon records: format 7 without last-event cores");
    // formats 6, 7: bit 32 + a = "axon a of the run spiked"; formats 0, 4: byte (code11) = "that axon spiked"
    __shared__ uint32_t s_bits[RUNS ? BLOCK / WAVE : 1][BYTE_TABLE ? RUN_MAX * 64 : RUNS ? RUN_MAX * 8 + 8 : 1];
    if (SYN_FMT == 6 && threadIdx.x < 32) s_lut[threadIdx.x] = im.weight_lut[threadIdx.x]; // visible after the barrier below
    if (SYN_FMT == 7 && threadIdx.x < 32) s_lut16[threadIdx.x] = (uint16_t) ((int) im.weight_lut[threadIdx.x] + (1 << im.acc_shift));

    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); // a scalar: chunk offsets and bases stay in SGPRs
    // Descriptors are in launch order: slices whose axons all start on this GPU first (delivered while the
    // spike bitmaps of the other GPUs are still being gathered), the rest after them.
    const SliceDesc sd = im.slice_desc[first_slice + blockIdx.x];
    const uint32_t slice = sd.slice_id;
    const uint32_t ncount = sd.ncount;
    const uint32_t npad = (ncount + 63u) & ~63u;
    const uint32_t nbase = sd.nbase;
    const uint32_t R = im.ring_slots;
    const uint32_t D = HAS_DELAY ? im.delay_slots : 1u; // LDS holds one accumulator row per delay value in use
    // Row stride of the accumulators.  Format 0 appends one "trash" entry per row: synapses whose charge is lost
    // (and the padding words) are packed with post == npad, so the stream path needs no test for them.
    constexpr bool STREAMABLE = (SYN_FMT == 0 || SYN_FMT == 3 || SYN_FMT == 4 || DICT16); // stream layout, index-coded words
    constexpr uint32_t GROUP_WORDS = DICT16 ? 8u : 4u; // synapse words in one 16-byte group
    constexpr bool FP_WEIGHTS = (SYN_FMT == 2 || SYN_FMT == 4);
    constexpr int SDEPTH = (SYN_FMT == 4) ? 2 : BITMAP ? SANAFE_BITMAP_STREAM_DEPTH : STREAM_DEPTH; // fp64 weights triple the registers of a group in flight
    const uint32_t RS = STREAMABLE ? npad + 1u : npad;
    const long long t = done + 1;
    const unsigned long long a_beg = sd.a_beg;
    const uint32_t n_ax = sd.n_ax; // slices hold < 2^32 axons
    const unsigned long long syn_base = sd.syn_base;
    const double ain_lat = sd.ain_lat;
    const bool compact = sd.mode != 0; // workgroup-uniform
    const bool bitmap = BITMAP && compact; // (on a BITMAP chip every compact slice keeps its records as a source bitmap)
    const unsigned char *rec = im.ax_bytes + sd.rec_off;
    const uint32_t *chunk_syn0 = im.chunk_syn0 + sd.chunk0;
    const uint32_t *chunk_pre0 = im.chunk_pre0 + sd.chunk0;
    const double slice_lat = sd.slice_lat;
    double *acc = reinterpret_cast<double *>(deliver_lds);                                  // [D][npad]
    uint32_t *acc32 = reinterpret_cast<uint32_t *>(deliver_lds);                            // the same in format 7
    // Which accumulators received a synaptic event (the buffer holds a value, even a zero one: src/chip.cpp:759)?
    // Integer-weight formats start every accumulator at -0.0, which no addition of weights can produce again;
    // fp64 weights (format 2) could be -0.0 themselves and keep a byte per accumulator instead.
    constexpr bool TOUCH_BYTES = FP_WEIGHTS;
    uint8_t *touched = deliver_lds + (size_t) im.delay_slots * (im.max_core_slots + 1u) * sizeof(double); // [D][RS]
    uint32_t *w_beg = s_beg[wave], *w_pref = s_pref[wave];
    const uint32_t core_inkind = sd.inkind;  // the buffer position belongs to the core
    const bool last_mode = LAST && (core_inkind == SANAFE_IN_LAST || core_inkind == SANAFE_IN_LAST_DELAY); // workgroup-uniform

    uint32_t *lastv = reinterpret_cast<uint32_t *>(deliver_lds);                          // [npad + 1] in last_mode

    double proc = 0.0;
    uint32_t stream_events = 0, stream_msgs = 0; // per lane: events / messages of the chunks this wave streamed
    const uint32_t *bits = st.bits_global;
    constexpr unsigned long long NONE = ~0ull; // wide record "past the end of the slice"
    // The 4 consecutive axon records of this lane.  Wide: two 16-byte loads (r[0..3]).  Compact: one 8-byte
    // load, kept in r[0] (4 x 16 bits; records past the end read as 0 = no synapses, no advance).
    auto load4 = [&](uint32_t a0, unsigned long long (&r)[AX_PER_THREAD]) {
        if (compact)
        {
            unsigned long long q = 0;
            if (a0 + AX_PER_THREAD <= n_ax) q = *reinterpret_cast<const unsigned long long *>(rec + 2ull * a0);
            else
                for (int k = 0; k < AX_PER_THREAD; k++)
                    if (a0 + k < n_ax) q |= (unsigned long long) *reinterpret_cast<const uint16_t *>(rec + 2ull * (a0 + k)) << (16 * k);
            r[0] = q;
            return;
        }
        const unsigned long long *wide = reinterpret_cast<const unsigned long long *>(rec);
        if (a0 + AX_PER_THREAD <= n_ax)
        {
            const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(wide + a0);
            const ulonglong2 hi = *reinterpret_cast<const ulonglong2 *>(wide + a0 + 2);
            r[0] = lo.x;
            r[1] = lo.y;
            r[2] = hi.x;
            r[3] = hi.y;
            return;
        }
#pragma unroll
        for (int k = 0; k < AX_PER_THREAD; k++) r[k] = (a0 + k < n_ax) ? wide[a0 + k] : NONE;
    };
    // Stream path state: the chunk's synapse words, STREAM_DEPTH 16-byte groups per lane loaded ahead.
    uint4 sq[SDEPTH];
    double2 sw[FP_WEIGHTS ? SDEPTH : 1][2]; // the four fp64 weights of each group (format 4)
    constexpr uint32_t stride = (BLOCK / WAVE) * WAVE_CHUNK;
    // Chunk order of a wave: every (BLOCK / WAVE)-th chunk (per-chunk loop), or `run_len` consecutive chunks at a time
    // (runs; a power of two, small enough that all waves get chunks of a short slice).
    uint32_t run_len = 1;
    if (RUNS && compact)
        while (run_len < RUN_MAX && run_len * (BLOCK / WAVE) * WAVE_CHUNK < n_ax) run_len *= 2;
    if (BITMAP && compact && run_len == RUN_MAX) run_len = im.bitmap_run_len; // (bitmap records: any length up to RUN_MAX, see DevImage)
    auto next_c = [&](uint32_t c) -> uint32_t { return c + stride; };
    uint32_t c0 = (uint32_t) wave * WAVE_CHUNK; // per-chunk loop: axon offset of the chunk inside the slice
    uint32_t *w_bits = s_bits[RUNS ? wave : 0];
    // The chunk loop is software-pipelined over three loads that depend on each other: axon records ->
    // (pre slots) -> spike-bitmap words -> (which axons spiked) -> synapse words.  While chunk i is delivered,
    // the bitmap words of chunk i+1 and the records of chunk i+2 are in flight.
    unsigned long long cur[AX_PER_THREAD] = {NONE, NONE, NONE, NONE}; // records of the chunk AFTER the decoded one
    uint32_t nx_pre[AX_PER_THREAD], nx_nsyn[AX_PER_THREAD], nx_lcls[AX_PER_THREAD], nx_word[AX_PER_THREAD], nx_valid = 0;
    // decode the records in `cur` (chunk at axon offset cc) and issue the loads of their bitmap words
    auto decode_and_probe = [&](uint32_t cc) {
        const uint32_t a0 = cc + (uint32_t) lane * AX_PER_THREAD;
        nx_valid = 0;
        if (compact)
        {
            const unsigned long long q = cur[0];
            uint32_t dl[AX_PER_THREAD];
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
            {
                dl[k] = (uint32_t) (q >> (16 * k)) & 0xffu;
                nx_nsyn[k] = (uint32_t) (q >> (16 * k + 8)) & 0xffu;
                nx_lcls[k] = 0u;
            }
            const uint32_t lane_d = dl[0] + dl[1] + dl[2] + dl[3];
            uint32_t pre = chunk_pre0[cc / WAVE_CHUNK] + wave_inclusive_scan(lane_d) - lane_d;
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
            {
                pre += dl[k];
                nx_pre[k] = pre;
                nx_valid |= (a0 + k < n_ax) ? (1u << k) : 0u;
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
            {
                const unsigned long long r = cur[k];
                nx_pre[k] = (r == NONE) ? 0u : (uint32_t) r;
                nx_nsyn[k] = (r == NONE) ? 0u : (uint32_t) ((r >> 32) & 0xffffu);
                nx_lcls[k] = (uint32_t) ((r >> 48) & 0xffu);
                nx_valid |= (r != NONE) ? (1u << k) : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < AX_PER_THREAD; k++) nx_word[k] = bits[nx_pre[k] >> 5]; // pad axons read word 0: in bounds
    };
    if (!(RUNS && compact) && c0 < n_ax)
    {
        load4(c0 + (uint32_t) lane * AX_PER_THREAD, cur);
        decode_and_probe(c0);
        if (next_c(c0) < n_ax) load4(next_c(c0) + (uint32_t) lane * AX_PER_THREAD, cur);
    }
    // The accumulators are cleared while the first records and bitmap words are in flight.
    for (uint32_t i = threadIdx.x; i < D * RS; i += BLOCK)
    {
        if (LAST && last_mode)
        {
            reinterpret_cast<unsigned long long *>(acc)[i] = 0ull; // two `lastv` entries
        }
        else if (INT_ACC)
        {
            if constexpr (SUB)
            {
#pragma unroll
                for (int q = 0; q < 16; q += 4) *reinterpret_cast<uint4 *>(&s_sub[i * 16u + q]) = make_uint4(0u, 0u, 0u, 0u);
            }
            else acc32[i] = 0u;
        }
        else if (TOUCH_BYTES)
        {
            acc[i] = 0.0;
            touched[i] = 0;
        }
        else
        {
            reinterpret_cast<unsigned long long *>(acc)[i] = ACC_UNTOUCHED;
        }
    }
    __syncthreads();
    // ---- GATHER: the spiking axons of one chunk (axon offset cg inside the slice; per lane the mask of its 4 axons,
    //      their synapse counts and latency classes) ----
    auto gather_chunk = [&](uint32_t cg, uint32_t amask, const uint32_t (&nsyn)[AX_PER_THREAD], const uint32_t (&lcls)[AX_PER_THREAD]) {
        // ---- GATHER: few spiking axons; touch only their synapses ----
        // the chunk's synapses are contiguous: one base + a prefix over ALL its axons' counts
        const uint32_t lane_syn = nsyn[0] + nsyn[1] + nsyn[2] + nsyn[3];
        uint32_t syn_off = chunk_syn0[cg / WAVE_CHUNK] + wave_inclusive_scan(lane_syn) - lane_syn;
        // ---- compact the active axons in axon (= reference delivery) order ----
        const uint32_t my_act = (uint32_t) __popc(amask);
        const uint32_t incl_act = wave_inclusive_scan(my_act);
        uint32_t my_ev = 0;
        uint32_t st4[AX_PER_THREAD], sb4[AX_PER_THREAD]; // MY active axons, packed to the front: event start, first synapse
#pragma unroll
        for (int k = 0; k < AX_PER_THREAD; k++) st4[k] = ~0u, sb4[k] = 0u;
        {
            uint32_t j = 0;
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
            {
                if (amask & (1u << k))
                {
                    // (static indices only: a runtime-indexed register array would go to scratch)
                    if (j == 0) st4[0] = my_ev, sb4[0] = syn_off;
                    else if (j == 1) st4[1] = my_ev, sb4[1] = syn_off;
                    else if (j == 2) st4[2] = my_ev, sb4[2] = syn_off;
                    else st4[3] = my_ev, sb4[3] = syn_off;
                    j++;
                    my_ev += nsyn[k];
                    if (compact) proc += ain_lat + (double) nsyn[k] * slice_lat;
                    else proc += (lcls[k] != 255u) ? ain_lat + (double) nsyn[k] * im.lat_class[lcls[k]] : im.ax_proc_delay[a_beg + cg + (uint32_t) lane * AX_PER_THREAD + k];
                }
                syn_off += nsyn[k];
            }
        }
        const uint32_t incl_ev = wave_inclusive_scan(my_ev);
        const uint32_t n_ev = __shfl(incl_ev, WAVE - 1, WAVE);
        const uint32_t lane_base = incl_ev - my_ev;
        // event e of the chunk belongs to the active axon i with start[i] <= e < start[i+1]; its synapse
        // is syn_base + (first_synapse[i] - start[i]) + e, so one word per active axon is enough
        {
            const uint32_t pos = incl_act - my_act;
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
                if ((uint32_t) k < my_act)
                {
                    st4[k] += lane_base;
                    w_beg[pos + k] = sb4[k] - st4[k];
                }
        }
        // ---- expand to synaptic events.  Ownership comes from a bitmap of axon starts ("heads"):
        //      owner(e) = (#heads at or before e) - 1, a popcount instead of a binary search. ----
        uint32_t heads_before = 0; // heads in earlier windows (wave-uniform)
        for (uint32_t w0 = 0; w0 < n_ev; w0 += HEAD_WINDOW)
        {
            w_pref[lane] = 0u; // 64 words = HEAD_WINDOW bits
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
            {
                const uint32_t rel = st4[k] - w0;
                if (st4[k] != ~0u && st4[k] >= w0 && rel < HEAD_WINDOW) atomicOr(&w_pref[rel >> 5], 1u << (rel & 31u));
            }
            wave_lds_fence();
            const uint32_t w_end = (n_ev - w0 < HEAD_WINDOW) ? n_ev - w0 : HEAD_WINDOW;
            uint32_t seen = heads_before; // heads before the current tile
            for (uint32_t e0 = 0; e0 < w_end; e0 += WAVE * EXPAND_UNROLL)
            {
                uint32_t meta[EXPAND_UNROLL]; // post (16b) | delay << 16 | drop << 19, whatever the stored format
                uint32_t spos[EXPAND_UNROLL]; // position among the core's synapses (used in last_mode)
                double wgt[EXPAND_UNROLL];
                uint32_t wgt32[EXPAND_UNROLL]; // format 7
#pragma unroll
                for (int u = 0; u < EXPAND_UNROLL; u++)
                {
                    const uint32_t tile = e0 + u * WAVE;
                    meta[u] = 1u << 19; // "drop": nothing to add
                    wgt[u] = 0.0;
                    wgt32[u] = 0u;
                    spos[u] = 0u;
                    if (tile < w_end) // wave-uniform
                    {
                        const unsigned long long h = (unsigned long long) w_pref[tile >> 5] | ((unsigned long long) w_pref[(tile >> 5) + 1] << 32);
                        const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                        const uint32_t owner = seen + (uint32_t) __popcll(h & le) - 1u;
                        seen += (uint32_t) __popcll(h);
                        const uint32_t e = w0 + tile + lane;
                        if (e < n_ev)
                        {
                            spos[u] = (uint32_t) (w_beg[owner] + e);
                            const unsigned long long s = syn_base + spos[u];
                            if (DICT16)
                            {
                                const uint32_t m = reinterpret_cast<const uint16_t *>(im.syn_meta)[s];
                                meta[u] = m >> 6;
                                if (INT_ACC) wgt32[u] = s_lut16[(m >> 1) & 31u];
                                else wgt[u] = s_lut[(m >> 1) & 31u];
                            }
                            else if (STREAMABLE)
                            {
                                const uint32_t m = im.syn_meta[s];
                                // the accumulator index itself (trash entry when the charge is lost)
                                meta[u] = SYN_FMT == 3 ? (m >> 8) & 0xfffu : (m >> 11) & (SYN_FMT == 0 ? 0x1fffu : 0x7fffu);
                                if (INT_ACC) wgt32[u] = (uint32_t) (((int) m >> (SYN_FMT == 0 ? 24 : 20)) + (1 << im.acc_shift));
                                else wgt[u] = SYN_FMT == 4 ? im.syn_weight[s] : (double) ((int) m >> (SYN_FMT == 0 ? 24 : 20));
                            }
                            else if (SYN_FMT == 1)
                            {
                                const uint32_t m = im.syn_meta[s];
                                meta[u] = m & 0xfffffu;
                                wgt[u] = (double) ((int) m >> 20);
                            }
                            else
                            {
                                meta[u] = im.syn_meta[s];
                                wgt[u] = im.syn_weight[s];
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < EXPAND_UNROLL; u++)
                    if (!((meta[u] >> 19) & 1u))
                    {
                        if (LAST && last_mode)
                        {
                            atomicMax(&lastv[meta[u] & 0xffffu], spos[u] + 1u); // (format 0: the index; no delays in this mode)
                            continue;
                        }
                        const uint32_t idx = STREAMABLE ? meta[u]
                                                            : (HAS_DELAY ? __umul24((meta[u] >> 16) & 7u, RS) : 0u) + (meta[u] & 0xffffu);
                        if (INT_ACC) atomicAdd(&acc32[idx], wgt32[u]); // ds_add_u32
                        else atomicAdd(&acc[idx], wgt[u]);           // ds_add_f64
                        if (TOUCH_BYTES) touched[idx] = 1;
                    }
            }
            heads_before = seen;
        }
        wave_lds_fence(); // the lists are rewritten by the next chunk
    };
    if (RUNS && compact)
    {
        // ================= dictionary formats, compact slices: runs of `run_len` consecutive chunks =================
        // Phase A, all chunks of the run at once: axon records -> pre slots -> spike-bitmap words -> which axons
        // spiked.  Every load is unconditional (clamped for chunks and lanes past the end), so the 8 record loads and then
        // the 32 bitmap probes are in flight together: two exposed memory round trips per run, not two per chunk.
        // Phase B: the words of the dense chunks of the run, streamed in one go.  Phase C: chunks with few spiking axons
        // go through the gather path.
        constexpr uint32_t NW = BLOCK / WAVE;
        // bitmap records: the two words of a lane for the NEXT run are loaded while this run streams (pf_*)
        // (+ where the run's chunks start among the core's synapse words, lane j: chunk j, lane n_here: the end -- a scalar load
        //  of two table entries at the start of phase B would put one more memory round trip in front of every run's stream)
        uint32_t pf_csyn = 0;
        auto bitmap_load = [&](uint32_t r, uint32_t &src_word, uint32_t &spk_word, bool with_csyn) {
            const uint32_t rr = r < n_ax ? r : 0u; // (past the end: any valid run, the values are not used)
            const uint32_t n_here = min(run_len, (n_ax - rr + WAVE_CHUNK - 1u) / WAVE_CHUNK);
            const bool have = (uint32_t) lane < n_here * 8u;
            const uint32_t w = (rr / WAVE_CHUNK) * 8u + (uint32_t) lane;
            src_word = have ? reinterpret_cast<const uint32_t *>(rec)[w] : 0u;
            spk_word = have ? bits[(uint32_t) a_beg + w] : 0u;
            if (with_csyn) pf_csyn = chunk_syn0[rr / WAVE_CHUNK + min((uint32_t) lane, n_here)];
        };
        uint32_t pf_src = 0, pf_spk = 0;
        if (BITMAP_RECORDS && bitmap && n_ax > 0) bitmap_load((uint32_t) wave * run_len * WAVE_CHUNK, pf_src, pf_spk, true);
        for (uint32_t r0 = (uint32_t) wave * run_len * WAVE_CHUNK; r0 < n_ax; r0 += NW * run_len * WAVE_CHUNK)
        {
            const uint32_t ci0 = r0 / WAVE_CHUNK;
            uint32_t amask_all = 0;                 // per lane: 4 bits per chunk of the run
            uint32_t dense_mask = 0, gather_mask = 0; // wave-uniform: chunks to stream / to gather
            // N = chunk slots compiled in: 8, or 1 for slices so short that every wave takes single chunks (small chips: no
            // masked-out decode work, no surplus probes)
            unsigned long long q_single = 0ull; // the records of a single-chunk run, kept for phase C
            auto phase_a = [&](auto n_slots) {
                constexpr uint32_t N = decltype(n_slots)::value, H = N >= 2 ? N / 2 : 1;
                unsigned long long q[N];
                // pre slot of each chunk's first axon: lane j fetches chunk j's (one load, broadcast by readlane below)
                const uint32_t n_here = min(run_len, (n_ax - r0 + WAVE_CHUNK - 1u) / WAVE_CHUNK); // chunks of this run inside the slice
                const uint32_t pre0_lane = chunk_pre0[ci0 + min((uint32_t) lane, n_here - 1u)];
                // one address, eight immediate offsets: the record array ends in 4 KB of padding, reads past the slice stay in
                // bounds and are zeroed below
                const unsigned long long *rec_lane = reinterpret_cast<const unsigned long long *>(rec + 2ull * (r0 + (uint32_t) lane * AX_PER_THREAD));
#pragma unroll
                for (uint32_t j = 0; j < N; j++)
                {
                    q[j] = rec_lane[j * (WAVE_CHUNK * 2u / 8u)];
                    keep_load_order();
                }
                // (N = 8: the bitmap probes in two halves of 16: 32 destination registers at once would spill)
#pragma unroll
                for (uint32_t half = 0; half < N; half += H)
                {
                if (half >= run_len) continue; // short runs (small slices): nothing in the second half
                uint32_t word[H][AX_PER_THREAD], shifts[H];
#pragma unroll
                for (uint32_t jj = 0; jj < H; jj++)
                {
                    const uint32_t j = half + jj;
                    const uint32_t a0 = r0 + j * WAVE_CHUNK + (uint32_t) lane * AX_PER_THREAD;
                    const bool have = j < run_len && a0 < n_ax;
                    if (!have) q[j] = 0ull;
                    const uint32_t lo = (uint32_t) q[j], hi = (uint32_t) (q[j] >> 32);
                    const uint32_t dl[AX_PER_THREAD] = {lo & 0xffu, (lo >> 16) & 0xffu, hi & 0xffu, (hi >> 16) & 0xffu};
                    const uint32_t lane_d = dl[0] + dl[1] + dl[2] + dl[3];
                    uint32_t pre = (uint32_t) __builtin_amdgcn_readlane((int) pre0_lane, (int) j) + wave_inclusive_scan(lane_d) - lane_d;
                    shifts[jj] = 0;
#pragma unroll
                    for (int k = 0; k < AX_PER_THREAD; k++)
                    {
                        pre += dl[k];
                        word[jj][k] = bits[pre >> 5]; // pad axons repeat the slot before them: in bounds
                        shifts[jj] |= (pre & 31u) << (8 * k);
                    }
                    keep_load_order();
                }
#pragma unroll
                for (uint32_t jj = 0; jj < H; jj++)
                {
                    const uint32_t j = half + jj;
                    // (an axon of these formats owns at least one synapse: a zero count marks the records past the end)
                    uint32_t amask = 0;
#pragma unroll
                    for (int k = 0; k < AX_PER_THREAD; k++)
                        amask |= min((word[jj][k] >> ((shifts[jj] >> (8 * k)) & 31u)) & 1u, (uint32_t) (q[j] >> (16 * k + 8)) & 0xffu) << k;
                    const uint32_t n_act_lanes = (uint32_t) __popcll(__ballot(amask != 0)); // lanes with a spiking axon
                    const bool dense = n_act_lanes >= STREAM_MIN_ACTIVE_LANES;
                    dense_mask |= dense ? (1u << j) : 0u;
                    gather_mask |= (!dense && n_act_lanes > 0) ? (1u << j) : 0u;
                    amask_all |= amask << (4 * j);
                    // Bit table: bit 32 + 256 j + a = "axon a of chunk j spiked" (chunks that are not streamed: zeros).  Eight
                    // lanes' masks make one dword: three DPP steps inside the rows, no LDS atomics.
                    if constexpr (BYTE_TABLE)
                    {
                        // byte table: lane L owns axons 4L..4L+3 of the chunk = one dword of the chunk's 256 bytes
                        w_bits[64u * ((ci0 + j) & 7u) + (uint32_t) lane] = dense ? (amask * 0x00204081u) & 0x01010101u : 0u;
                    }
                    else
                    {
                        uint32_t m8 = dense ? amask : 0u;
                        m8 |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m8, 0x101, 0xf, 0xf, true) << 4;  // row_shl:1
                        m8 |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m8, 0x102, 0xf, 0xf, true) << 8;  // row_shl:2
                        m8 |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m8, 0x104, 0xf, 0xf, true) << 16; // row_shl:4
                        if ((lane & 7) == 0) w_bits[1u + 8u * j + ((uint32_t) lane >> 3)] = m8;
                    }
                    if (dense)
                    {
                        // processing delay of the chunk's messages: axon-in latency per message + per-event latency; counted
                        // in integers here, priced once at the end of the slice
                        // (a 4 x 8-bit dot product: the synapse counts of the spiking axons)
                        const uint32_t lo = (uint32_t) q[j], hi = (uint32_t) (q[j] >> 32);
                        const uint32_t counts = __builtin_amdgcn_perm(hi, lo, 0x07050301u); // bytes 1, 3 of lo and of hi
                        stream_events = __builtin_amdgcn_udot4(counts, (amask * 0x00204081u) & 0x01010101u, stream_events, false);
                        stream_msgs += (uint32_t) __popc(amask);
                    }
                }
                }
                if (N == 1) q_single = q[0];
            };
            uint32_t before_bm = 0;                  // bitmap records: axons of the run before its first dense chunk
            uint32_t csyn_lane = 0;                  // bitmap records: first synapse word of chunk `lane` of this run (prefetched)
            // bitmap records: this lane's source word, its spiking axons, the axons of the run before it (phase C loads them
            // again rather than keeping three registers alive across the stream)
            auto bitmap_words = [&](uint32_t &bm_src, uint32_t &bm_f, uint32_t &bm_excl, uint32_t &fincl, bool prefetched) {
                const uint32_t n_here = min(run_len, (n_ax - r0 + WAVE_CHUNK - 1u) / WAVE_CHUNK);
                uint32_t spk;
                if (prefetched)
                {
                    bm_src = pf_src;
                    spk = pf_spk;
                }
                else bitmap_load(r0, bm_src, spk, false);
                bm_f = spk & bm_src;
                const uint32_t cnt = (uint32_t) __popc(bm_src);
                bm_excl = wave_inclusive_scan(cnt) - cnt;
                fincl = wave_inclusive_scan((uint32_t) __popc(bm_f));
                return n_here;
            };
            if (BITMAP_RECORDS && bitmap)
            {
                uint32_t bm_src, bm_f, bm_excl, fincl;
                // ---- Phase A on BITMAP records (slice mode 2): a chunk is a 256-slot window of the source space, one 32-bit
                //      word per lane, a run = 8 windows = the 64 lanes.  "Which axons spiked" is ONE AND of two coalesced
                //      loads -- the slice's source bitmap (bit = this core has an axon from that neuron) and the global spike
                //      bitmap at the same position -- instead of a record decode, a prefix sum and a random probe per axon.
                const uint32_t n_here = bitmap_words(bm_src, bm_f, bm_excl, fincl, true);
                csyn_lane = pf_csyn;
                bitmap_load(r0 + NW * run_len * WAVE_CHUNK, pf_src, pf_spk, true); // the next run's, in flight during the stream
                stream_msgs += (uint32_t) __popc(bm_f);
                // per chunk (8 lanes): spiking axons -> stream (many), gather (few) or nothing
#pragma unroll
                for (uint32_t j = 0; j < RUN_MAX; j++)
                {
                    if (j >= n_here) break;
                    const uint32_t hi8 = (uint32_t) __builtin_amdgcn_readlane((int) fincl, (int) (8u * j + 7u));
                    const uint32_t lo8 = j == 0u ? 0u : (uint32_t) __builtin_amdgcn_readlane((int) fincl, (int) (8u * j - 1u));
                    const uint32_t n_spiking = hi8 - lo8;
                    dense_mask |= (n_spiking >= STREAM_MIN_ACTIVE_LANES) ? (1u << j) : 0u;
                    gather_mask |= (n_spiking > 0u && n_spiking < STREAM_MIN_ACTIVE_LANES) ? (1u << j) : 0u;
                }
                if (dense_mask != 0u)
                {
                    // bit table for phase B: bit 32 + a = "axon a (counted from the first dense chunk) spiked".  The spiking
                    // bits of a lane are its word compressed by its source mask, placed at the lane's first axon.
                    const uint32_t first_dense = (uint32_t) __builtin_ctz(dense_mask);
                    if constexpr (DICT16)
                    {
                        // the stream's first loads go out BEFORE the table is built: its ~120 instructions and LDS round trips
                        // then run under the loads' latency instead of in front of it (same pattern as in phase B below)
                        const uint32_t last_d = 31u - (uint32_t) __builtin_clz(dense_mask);
                        const uint32_t p0 = (uint32_t) __builtin_amdgcn_readlane((int) csyn_lane, (int) first_dense);
                        const uint32_t ng = ((uint32_t) __builtin_amdgcn_readlane((int) csyn_lane, (int) (last_d + 1u)) - p0) / GROUP_WORDS;
                        const uint4 *s0 = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(im.syn_meta) + (syn_base + p0));
#pragma unroll
                        for (int u = 0; u < SDEPTH; u++)
                        {
                            const uint32_t g_u = (uint32_t) lane + (uint32_t) u * WAVE;
                            sq[u] = load_stream16(s0 + (g_u < ng ? g_u : ng - 1u));
                            keep_load_order();
                        }
                    }
                    before_bm = (uint32_t) __builtin_amdgcn_readlane((int) bm_excl, (int) (8u * first_dense));
                    w_bits[lane] = 0u;
                    if (lane < 8) w_bits[64 + lane] = 0u;
                    wave_lds_fence();
                    const bool my_dense = (dense_mask >> ((uint32_t) lane >> 3)) & 1u;
                    uint32_t x = my_dense ? bm_f : 0u;
                    if (x != 0u)
                    {
                        // compress x by the mask bm_src (Hacker's Delight 7-4: five parallel-suffix rounds)
                        uint32_t m = bm_src, mk = ~m << 1;
#pragma unroll
                        for (int i = 0; i < 5; i++)
                        {
                            uint32_t mp = mk ^ (mk << 1);
                            mp ^= mp << 2;
                            mp ^= mp << 4;
                            mp ^= mp << 8;
                            mp ^= mp << 16;
                            const uint32_t mv = mp & m;
                            m = (m ^ mv) | (mv >> (1 << i));
                            const uint32_t tt = x & mv;
                            x = (x ^ tt) | (tt >> (1 << i));
                            mk &= ~mp;
                        }
                        const uint32_t pos = 32u + bm_excl - before_bm, sh = pos & 31u;
                        atomicOr(&w_bits[pos >> 5], x << sh);
                        if (sh != 0u && (x >> (32u - sh)) != 0u) atomicOr(&w_bits[(pos >> 5) + 1u], x >> (32u - sh));
                    }
                }
            }
            else if constexpr (!BITMAP)
            {
                if (run_len == 1) phase_a(std::integral_constant<uint32_t, 1>{});
                else phase_a(std::integral_constant<uint32_t, RUN_MAX>{});
            }
            if (dense_mask != 0u)
            {
                // ---- Phase B: stream the words of chunks first_dense .. last_dense ----
                wave_lds_fence();
                const uint32_t first_dense = (uint32_t) __builtin_ctz(dense_mask), last_dense = 31u - (uint32_t) __builtin_clz(dense_mask);
                const uint32_t run_pos0 = (BITMAP_RECORDS && bitmap) ? (uint32_t) __builtin_amdgcn_readlane((int) csyn_lane, (int) first_dense)
                                                                     : chunk_syn0[ci0 + first_dense];
                const uint32_t run_end = (BITMAP_RECORDS && bitmap) ? (uint32_t) __builtin_amdgcn_readlane((int) csyn_lane, (int) (last_dense + 1u))
                                                                    : chunk_syn0[ci0 + last_dense + 1u];
                const uint32_t run_groups = (run_end - run_pos0) / GROUP_WORDS; // chunks are 16-byte aligned and padded
                const uint4 *src = DICT16 ? reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(im.syn_meta) + (syn_base + run_pos0))
                                          : reinterpret_cast<const uint4 *>(im.syn_meta + (syn_base + run_pos0));
                const double2 *wsrc = (SYN_FMT == 4) ? reinterpret_cast<const double2 *>(im.syn_weight + (syn_base + run_pos0)) : nullptr;
                // The first SDEPTH groups of every lane.  All loads of the stream are unconditional (past the end: the last
                // group again) and issued in one fixed pattern: only then can the loads in flight be counted, so that a
                // group waits for ITS load (vmcnt(SDEPTH - 1)) and not for all of them.
                if (!(BITMAP_RECORDS && bitmap)) // (bitmap records: issued before the bit table was built)
                {
#pragma unroll
                for (int u = 0; u < SDEPTH; u++)
                {
                    const uint32_t g_u = (uint32_t) lane + (uint32_t) u * WAVE;
                    const uint32_t g = g_u < run_groups ? g_u : run_groups - 1u;
                    sq[u] = load_stream16(src + g);
                    if (SYN_FMT == 4)
                    {
                        sw[u][0] = wsrc[2 * g];
                        sw[u][1] = wsrc[2 * g + 1];
                    }
                    keep_load_order();
                }
                }
                if constexpr (DICT16)
                {
                // 8 words per lane and group.  A word's axon = (first-synapse bits of the run up to and including it) - 1:
                // per group the lanes count their bits, one DPP prefix sum orders the lanes, a scalar carries the count
                // from group to group.  The eight axons of a lane are consecutive, so ONE 32-bit window of the bit
                // table (two dwords, funnel-shifted) answers "spiked?" for all eight words.
            uint32_t before = (BITMAP_RECORDS && bitmap) ? 0u : first_dense * WAVE_CHUNK; // first-synapse bits of the run before the current group-instruction (wave-uniform); bitmap records count axons from the first dense chunk
            // LDS address of the accumulators, hidden from constant folding: the compiler then forms
            // base + (index << 2) in one instruction instead of rebuilding it from shifted masks
            typedef __attribute__((address_space(3))) uint32_t lds_u32;
            uint32_t acc_base = (uint32_t) (uintptr_t) (lds_u32 *) acc32;
            asm volatile("" : "+v"(acc_base));
            auto add8 = [&](const uint4 &q, uint32_t pos0, bool live) {
                const uint32_t d4[4] = {q.x, q.y, q.z, q.w};
                // first-synapse bits up to and including word 1, 3, 5, 7 of the lane: one chained popcount each
                uint32_t upto[4];
                upto[0] = (uint32_t) __popc(d4[0] & 0x00010001u);
                upto[1] = upto[0] + (uint32_t) __popc(d4[1] & 0x00010001u);
                upto[2] = upto[1] + (uint32_t) __popc(d4[2] & 0x00010001u);
                upto[3] = upto[2] + (uint32_t) __popc(d4[3] & 0x00010001u);
                const uint32_t mine = upto[3];
                const uint32_t incl = wave_inclusive_scan(mine);
                const uint32_t t = 31u + before + incl - mine; // table bit of the axon before this lane's first first-synapse bit
                before += (uint32_t) __builtin_amdgcn_readlane((int) incl, WAVE - 1);
                // the window, bit-reversed: "axon spiked" becomes a sign test after one shift
                const uint32_t winr = __builtin_bitreverse32(__builtin_amdgcn_alignbit(w_bits[(t >> 5) + 1u], w_bits[t >> 5], t & 31u));
                bool fired[8];
                uint32_t x = winr;
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    fired[2 * j] = (int) (x << (d4[j] & 1u)) < 0;   // axon of the even word: one further if it starts one
                    x = winr << upto[j];
                    fired[2 * j + 1] = (int) x < 0;
                }
                if constexpr (INT_ACC)
                {
                    uint32_t wv[8]; // all eight dictionary reads in flight together
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        wv[2 * j] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(s_lut16) + (d4[j] & 0x3eu));
                        wv[2 * j + 1] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(s_lut16) + ((d4[j] >> 16) & 0x3eu));
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) asm volatile("" : "+v"(wv[k])); // keeps the zero-extension in the load (ds_read_u16), not an AND per use
                    if (live)
                    {
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (fired[k])
                            {
                                lds_u32 *slot;
                                if constexpr (SUB)
                                {
                                    // (LDS byte address = static array + the 16-bit word with its low two bits masked)
                                    const uint32_t half = (k & 1) ? (d4[k >> 1] >> 16) : d4[k >> 1];
                                    slot = (lds_u32 *) &s_sub[(half & 0xfffcu) >> 2];
                                }
                                else
                                {
                                // (LDS byte address = opaque base + 4 * index: a bit-field extract and one shift-add)
                                uint32_t idx = __builtin_amdgcn_ubfe(d4[k >> 1], (k & 1) ? 22u : 6u, 10u);
                                asm("" : "+v"(idx)); // (or the shift is folded back into the extract: three instructions)
                                slot = reinterpret_cast<lds_u32 *>(acc_base + (idx << 2));
                                }
                                if (LAST && last_mode) __hip_atomic_fetch_max(slot, pos0 + (uint32_t) k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                else __hip_atomic_fetch_add(slot, wv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // ds_add_u32 (lost charge lands in the trash entry)
                            }
                    }
                }
                else
                {
#pragma unroll
                    for (int h = 0; h < 8; h += 4)
                    {
                        double wv[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            wv[k] = *reinterpret_cast<const double *>(reinterpret_cast<const uint8_t *>(s_lut) + (((d4[(h + k) >> 1] >> ((k & 1) ? 16 : 0)) & 0x3eu) << 2));
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (live && fired[h + k])
                            {
                                const uint32_t idx = __builtin_amdgcn_ubfe(d4[(h + k) >> 1], (k & 1) ? 22u : 6u, 10u);
                                if (LAST && last_mode) atomicMax(&lastv[idx], pos0 + (uint32_t) (h + k) + 1u);
                                else atomicAdd(&acc[idx], wv[k]); // ds_add_f64 (lost charge lands in the trash entry)
                            }
                    }
                }
            };
            // Lanes past the end of the run keep whatever their registers hold: they are the highest lanes of the last
            // group-row, so their first-synapse counts reach no live lane, and `live` keeps them from adding.
            const int lane_groups = (int) run_groups - lane; // this lane has group (row + lane) while row < lane_groups
            for (uint32_t row = 0; row < run_groups; row += WAVE * SDEPTH) // row: first group of the wave's group-row (a scalar)
            {
#pragma unroll
                for (int u = 0; u < SDEPTH; u++)
                {
                    const uint32_t r = row + (uint32_t) u * WAVE;
                    if (r < run_groups) add8(sq[u], run_pos0 + 8u * (r + (uint32_t) lane), (int) r < lane_groups);
                    // The refill is unconditional (past the end: the last group again): with a load behind a branch
                    // the compiler cannot count the loads in flight and waits for ALL of them before every group,
                    // which leaves one group per wave in flight instead of SDEPTH.
                    const uint32_t nxt = r + (uint32_t) SDEPTH * WAVE + (uint32_t) lane;
                    sq[u] = load_stream16(src + (nxt < run_groups ? nxt : run_groups - 1u));
                    keep_load_order();
                }
            }
                }
                else
                {
                    // 4 words per lane and group: the word's low 11 bits (format 3: its low 8 bits + the chunk of its group)
                    // index the byte table, the next bits ARE the LDS accumulator index (lost charge lands in the trash entry)
                    const uint8_t *spiked = reinterpret_cast<const uint8_t *>(w_bits);
                    const int acc_bias = 1 << im.acc_shift; // integer accumulators: every event adds weight + 2^acc_shift
                    // format 3: where the chunks of the run end, in groups (lane j: chunk first_dense + j), and the chunk the
                    // current group-row starts in (scalars)
                    const uint32_t n_run = last_dense - first_dense + 1u;
                    const uint32_t end_lane = CODE11 ? 0u : (chunk_syn0[ci0 + first_dense + min((uint32_t) lane, n_run - 1u) + 1u] - run_pos0) / GROUP_WORDS;
                    uint32_t cur = 0, cur_end = CODE11 ? 0u : (uint32_t) __builtin_amdgcn_readlane((int) end_lane, 0);
                    auto add4 = [&](const uint4 &g, const double2 (&wq)[2], uint32_t pos0 /* position of g.x among the core's synapses */, uint32_t tab) {
                        const uint32_t w4[4] = {g.x, g.y, g.z, g.w};
                        const double f4[4] = {wq[0].x, wq[0].y, wq[1].x, wq[1].y};
                        uint32_t fired[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) fired[u] = spiked[CODE11 ? (w4[u] & 0x7ffu) : ((w4[u] & 0xffu) | tab)];
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (fired[u])
                            {
                                const uint32_t idx = SYN_FMT == 3 ? (w4[u] >> 8) & 0xfffu : (w4[u] >> 11) & (SYN_FMT == 0 ? 0x1fffu : 0x7fffu);
                                if (LAST && last_mode)
                                {
                                    atomicMax(&lastv[idx], pos0 + (uint32_t) u + 1u);
                                    continue;
                                }
                                if (INT_ACC) atomicAdd(&acc32[idx], (uint32_t) (((int) w4[u] >> (SYN_FMT == 3 ? 20 : 24)) + acc_bias)); // ds_add_u32
                                else atomicAdd(&acc[idx], SYN_FMT == 4 ? f4[u] : (double) ((int) w4[u] >> (SYN_FMT == 3 ? 20 : 24))); // ds_add_f64
                                if (TOUCH_BYTES) touched[idx] = 1;
                            }
                    };
                    for (uint32_t row = 0; row < run_groups; row += WAVE * SDEPTH) // row: first group of the wave's group-row (a scalar)
                    {
#pragma unroll
                        for (int u = 0; u < SDEPTH; u++)
                        {
                            // add, then refill the same registers (the other groups of the window are still in flight)
                            const uint32_t r = row + (uint32_t) u * WAVE; // first group of this row (scalar)
                            const uint32_t g = r + (uint32_t) lane;
                            uint32_t tab = 0; // format 3: byte-table offset of the group's chunk
                            if (!CODE11 && r < run_groups)
                            {
                                while (r >= cur_end) cur_end = (uint32_t) __builtin_amdgcn_readlane((int) end_lane, (int) ++cur); // (chunks hold words: it ends)
                                tab = ((ci0 + first_dense + cur) & 7u) << 8;
                                // chunks that begin inside this row
                                for (uint32_t c = cur, e = cur_end; e < r + WAVE && c + 1u < n_run;)
                                {
                                    c++;
                                    tab = (g >= e) ? ((ci0 + first_dense + c) & 7u) << 8 : tab;
                                    e = (uint32_t) __builtin_amdgcn_readlane((int) end_lane, (int) c);
                                }
                            }
                            if (g < run_groups) add4(sq[u], sw[FP_WEIGHTS ? u : 0], run_pos0 + 4u * g, tab);
                            const uint32_t nxt_g = g + (uint32_t) SDEPTH * WAVE;
                            const uint32_t nxt = nxt_g < run_groups ? nxt_g : run_groups - 1u;
                            sq[u] = load_stream16(src + nxt);
                            if (SYN_FMT == 4)
                            {
                                sw[u][0] = wsrc[2 * nxt];
                                sw[u][1] = wsrc[2 * nxt + 1];
                            }
                            keep_load_order();
                        }
                    }
                }
                wave_lds_fence(); // the table is rewritten by the next run
            }
            // ---- Phase C: chunks with a few spiking axons ----
            if (BITMAP_RECORDS && bitmap && gather_mask != 0u) // wave-uniform
            {
            uint32_t bm_src, bm_f, bm_excl, fincl_unused;
            bitmap_words(bm_src, bm_f, bm_excl, fincl_unused, false);
            while (gather_mask != 0u)
            {
                // bitmap records: a window with fewer than STREAM_MIN_ACTIVE_LANES spiking axons.  The whole wave takes them
                // one by one: first synapse = the chunk's first + the synapse counts of the axons before it (one byte per axon
                // behind the slice's bitmap words), then one lane per synapse.
                const uint32_t j = (uint32_t) __builtin_ctz(gather_mask);
                gather_mask &= gather_mask - 1u;
                const uint32_t ci = ci0 + j;
                const uint32_t ax0 = chunk_pre0[ci], axc = chunk_pre0[ci + 1u] - ax0; // first axon of the window (within the slice), axons in it
                const uint8_t *nsyn8 = rec + (size_t) (n_ax / WAVE_CHUNK) * 32u + ax0;
                uint32_t mine[AX_PER_THREAD], lane_sum = 0;
#pragma unroll
                for (int k = 0; k < AX_PER_THREAD; k++)
                {
                    const uint32_t a = (uint32_t) lane * AX_PER_THREAD + (uint32_t) k;
                    mine[k] = a < axc ? nsyn8[a] : 0u;
                    lane_sum += mine[k];
                }
                uint32_t run_sum = wave_inclusive_scan(lane_sum) - lane_sum;
#pragma unroll
                for (int k = 0; k < AX_PER_THREAD; k++)
                {
                    w_beg[(uint32_t) lane * AX_PER_THREAD + (uint32_t) k] = run_sum | (mine[k] << 24); // first synapse (in the chunk) | count
                    run_sum += mine[k];
                }
                wave_lds_fence();
                const uint32_t chunk_first = chunk_syn0[ci];
                const uint32_t excl0 = (uint32_t) __builtin_amdgcn_readlane((int) bm_excl, (int) (8u * j));
                for (uint32_t wv = 0; wv < 8u; wv++)
                {
                    uint32_t fw = (uint32_t) __builtin_amdgcn_readlane((int) bm_f, (int) (8u * j + wv));
                    const uint32_t sw = (uint32_t) __builtin_amdgcn_readlane((int) bm_src, (int) (8u * j + wv));
                    const uint32_t ex = (uint32_t) __builtin_amdgcn_readlane((int) bm_excl, (int) (8u * j + wv)) - excl0;
                    while (fw != 0u) // scalar loop over the spiking axons of this word
                    {
                        const uint32_t b = (uint32_t) __builtin_ctz(fw);
                        fw &= fw - 1u;
                        const uint32_t a = ex + (uint32_t) __builtin_popcount(sw & ((1u << b) - 1u)); // axon within the window
                        const uint32_t e = w_beg[a];
                        const uint32_t first = chunk_first + (e & 0xffffffu), n = e >> 24;
                        for (uint32_t k = (uint32_t) lane; k < n; k += WAVE)
                        {
                            const uint32_t word = reinterpret_cast<const uint16_t *>(im.syn_meta)[syn_base + first + k];
                            atomicAdd(SUB ? &s_sub[(word & 0xfffcu) >> 2] : &acc32[word >> 6], (uint32_t) s_lut16[(word >> 1) & 31u]); // ds_add_u32
                        }
                    }
                }
                wave_lds_fence(); // w_beg is rewritten by the next window
            }
            }
            while (gather_mask != 0u) // wave-uniform
            {
                const uint32_t j = (uint32_t) __builtin_ctz(gather_mask);
                gather_mask &= gather_mask - 1u;
                const uint32_t cg = r0 + j * WAVE_CHUNK;
                const uint32_t a0 = cg + (uint32_t) lane * AX_PER_THREAD;
                // (longer runs do not keep their records in registers: one more load, L2-hot)
                const unsigned long long qq = (run_len == 1) ? q_single : (a0 < n_ax) ? *reinterpret_cast<const unsigned long long *>(rec + 2ull * a0) : 0ull;
                const uint32_t lo = (uint32_t) qq, hi = (uint32_t) (qq >> 32);
                const uint32_t nsyn[AX_PER_THREAD] = {(lo >> 8) & 0xffu, lo >> 24, (hi >> 8) & 0xffu, hi >> 24};
                const uint32_t lcls[AX_PER_THREAD] = {0u, 0u, 0u, 0u};
                gather_chunk(cg, (amask_all >> (4u * j)) & 15u, nsyn, lcls);
            }
        }
    }
    else
    for (;; c0 = next_c(c0))
    {
        // ---- this chunk: take over what the previous iteration decoded and probed ----
        uint32_t amask = 0, nsyn[AX_PER_THREAD], lcls[AX_PER_THREAD];
        if (c0 < n_ax)
        {
#pragma unroll
            for (int k = 0; k < AX_PER_THREAD; k++)
            {
                nsyn[k] = nx_nsyn[k];
                lcls[k] = nx_lcls[k];
                amask |= ((nx_word[k] >> (nx_pre[k] & 31u)) & (nx_valid >> k) & 1u) << k;
            }
            // ---- next chunk: decode its records, probe the bitmap; then start the loads of the chunk after it ----
            const uint32_t c1 = next_c(c0);
            if (c1 < n_ax)
            {
                decode_and_probe(c1);
                const uint32_t c2 = next_c(c1);
                if (c2 < n_ax) load4(c2 + (uint32_t) lane * AX_PER_THREAD, cur);
            }
        }
        if (c0 >= n_ax) break;
        if (__ballot(amask != 0) == 0ull) continue; // wave-uniform
        gather_chunk(c0, amask, nsyn, lcls);
    }
    // ---- write the accumulated charge back (one access per touched neuron and delay value) ----
    __syncthreads(); // every wave's additions are in the accumulators
    long long wb_events = 0; // bitmap records: the synaptic events of the slice = the event counts of its integer accumulators
    const bool shared_core = sd.shared != 0;
    if (LAST && last_mode)
    {
        // the latest event over all slices of the core wins: positions grow in delivery order
        for (uint32_t n = threadIdx.x; n < npad; n += BLOCK)
            if (lastv[n] != 0u) atomicMax(&st.ring_last[nbase + n], lastv[n]);
    }
    else
    for (uint32_t i = threadIdx.x; i < D * RS; i += BLOCK)
    {
        double sum;
        if (INT_ACC)
        {
            // count * 2^shift + sum of weights, |sum| < 2^(shift-1) (checked by the host): 0 = no event arrived
            uint32_t v;
            if constexpr (SUB)
            {
                v = 0u;
#pragma unroll
                for (int q = 0; q < 16; q += 4)
                {
                    const uint4 p4 = *reinterpret_cast<const uint4 *>(&s_sub[i * 16u + q]);
                    v += (p4.x + p4.y) + (p4.z + p4.w);
                }
            }
            else v = acc32[i];
            if (v == 0u) continue;
            const uint32_t events = (v + (1u << (im.acc_shift - 1))) >> im.acc_shift;
            sum = (double) (int) (v - (events << im.acc_shift));
            if (BITMAP_RECORDS && bitmap && i % RS < npad) wb_events += events; // (not the trash entry: padding words land there)
        }
        else
        {
            if (TOUCH_BYTES ? !touched[i] : (reinterpret_cast<const unsigned long long *>(acc)[i] == ACC_UNTOUCHED)) continue;
            sum = acc[i];
        }
        const uint32_t d = i / RS, n = i - d * RS;
        if (n >= npad) continue; // trash entry
        // per neuron: a core may mix dendrite units
        const uint32_t post_kind = HAS_DELAY ? (im.slot_cls[nbase + n] >> 3) & 7u : (uint32_t) SANAFE_IN_BUFFERED;
        if (post_kind == SANAFE_IN_TAPS)
        {
            // row d is tap d of the neuron's dendrite: taps_kernel integrates it after this launch
            atomicAdd(&st.tap_in[(size_t) im.slot_aux[nbase + n] * 8u + d], sum);
            st.arrived[nbase + n] = 1;
            continue;
        }
        const bool gated = post_kind == SANAFE_IN_GATED; // a neuron behind a gated delay line
        const uint32_t wslot = (uint32_t) ((t + 1 + d + (gated ? 1 : 0)) % R);
        const size_t gi = (size_t) wslot * im.n_slots + nbase + n;
        // Without synaptic delays all charge for step t+1 arrives in step t and the neuron launch of step t left
        // every consumed entry at 0.0: a core with one slice stores, it need not read.
        if (shared_core) atomicAdd(&st.ring[gi], sum);
        else if (!HAS_DELAY) st.ring[gi] = sum;
        else st.ring[gi] += sum;
        st.ring_valid[gi] = 1;
        if (gated) st.arrived[nbase + n] = 1;
    }
    // ---- processing-delay sum of this slice (simple timing model): wave partials, combined after the barrier ----
    if (STREAMABLE) proc += (double) stream_events * slice_lat + (double) stream_msgs * ain_lat;
    proc = wave_sum(proc);
    if (BITMAP_RECORDS) wb_events = wave_sum(wb_events);
    if (lane == 0)
    {
        s_red[wave] = proc;
        if (BITMAP_RECORDS) s_redi[wave] = wb_events;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double p = 0.0;
        long long ev = 0;
        for (int w = 0; w < BLOCK / WAVE; w++)
        {
            p += s_red[w];
            if (BITMAP_RECORDS) ev += s_redi[w];
        }
        if (BITMAP_RECORDS && bitmap) p += (double) ev * slice_lat;
        // one value per slice; level 1 of the step reduction adds a core's slices in order (reproducible)
        st.slice_proc[(size_t) (done & 1) * im.n_slices + slice] = p;
    }
}

// ---------------------------------------------------------------------------------------
// K2e: EVENT-DRIVEN spike delivery (DevImage::ev_*): work in proportion to the step's synaptic events, like the reference's
// process_messages / process_message (src/chip.cpp:656-764), for steps in which a few percent of the neurons fire.
//
//   grid = 8 x ceil(groups / 8) x segments, block = 256.  A workgroup owns the LDS accumulators of ONE group of destination
//   cores and ONE segment of the source space.  Workgroups b and b + 8 share an XCD (observed round-robin placement; speed
//   only): consecutive workgroups of an XCD take neighbouring groups of the same segment, so the 128-byte lines that hold a
//   fired neuron's blocks for neighbouring groups are fetched into that XCD's L2 once.
//   Every wavefront (no barrier inside the loop)
//     1. scans 1,024-slot tiles of its segment of the spike bitmap and lists the neurons that fired (16-bit entries in LDS);
//     2. drains the list in batches of 64 / LPB neurons, LPB lanes per neuron: table entry of (neuron, group) -> where the
//        block starts, how many 16-byte units it holds, which cores it reaches; every lane takes one unit = 8 words and adds
//        weight + 2^shift into the group's 32-bit integer accumulators (sums of integers: exact in any order).  Three stages
//        in flight: the table entries of batch b + 2 and the words of batch b + 1 are loading while batch b is added.
//   Write-back: every accumulator (count * 2^shift + sum, untouched: 0) goes to the workgroup's segment row of
//   DevState::ev_part with plain coalesced stores; the NEXT neuron launch adds a neuron's rows up (neuron_kernel: ev_in).
//   The segments of a group share its neurons, and global fp64 atomics run memory-side at a fifth of the store rate: with
//   atomics into the time-step buffer the write-back alone was ~13 us of a 38 us launch at 2 % activity.
//   Messages and events per destination core go to the push counters that level 1 of the step reduction prices.
//   WAVES wavefronts per workgroup share the accumulators: more loads in flight per CU without more rows to write back.
//   UPL units per lane and batch: with LPB = 4 and UPL = 2 a batch covers 16 neurons instead of 8 with the same 8 unit
//   slots per block -- twice the table entries and words in flight per wavefront (the launch is bound by the bytes in
//   flight per CU, not by LDS or vector issue).
// ---------------------------------------------------------------------------------------
constexpr uint32_t EV_TILE = 1024;      // source slots per tile: 16 per lane
constexpr uint32_t EV_LIST_CAP = 1536;  // per wavefront (16-bit entries): a tile adds at most 1,024, the list is drained from 512 on
constexpr uint32_t EV_DRAIN_AT = 512;
constexpr uint32_t EV_TRASH = 64;       // accumulators behind a group's own that padding words add into
// (plain loads: consecutive fired neurons' blocks share 128-byte lines, which the caches should keep for the neighbour)
__device__ __forceinline__ uint4 ev_load16(const uint4 *p)
{
#ifdef SANAFE_EVENT_NT_LOADS
    return load_stream16(p);
#else
    return *p;
#endif
}
template <int LPB, int CODE_BITS, int WAVES, int UPL = 1, bool SPARSE = false /* the neuron-major copy of the block table */>
__global__ void __launch_bounds__(WAVES * WAVE)
event_deliver_kernel(DevImage im, DevState st, long long done /* steps simulated before this one */)
{
    constexpr uint32_t ACC_MAX = 1u << (16 - CODE_BITS);
    constexpr uint32_t CODE_MASK = (1u << CODE_BITS) - 1u;
    constexpr uint32_t NB = WAVE / LPB; // neurons per batch
    constexpr uint32_t BLOCK = WAVES * WAVE;
    __shared__ uint32_t s_acc[ACC_MAX];
    __shared__ uint16_t s_lut16[32];
    __shared__ uint16_t s_list[WAVES][EV_LIST_CAP];
    __shared__ uint32_t s_msgs[16], s_events[16];
    __shared__ uint8_t s_chunk_core[ACC_MAX / WAVE]; // core (within the group) of every 64-accumulator chunk
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    // workgroup -> (group, segment): blocks b and b + 8 share an XCD
    const uint32_t gpx = (im.ev_groups + 7u) / 8u;
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const uint32_t g = xcd * gpx + k % gpx, seg = k / gpx;
    if (g >= im.ev_groups) return; // (workgroup-uniform)
    const EvGroup eg = im.ev_group[g];
    for (uint32_t i = threadIdx.x; i < ACC_MAX; i += BLOCK) s_acc[i] = 0u;
    if (threadIdx.x < 32) s_lut16[threadIdx.x] = (uint16_t) ((int) im.ev_lut[threadIdx.x] + (1 << im.ev_shift));
    if (threadIdx.x < 16) s_msgs[threadIdx.x] = s_events[threadIdx.x] = 0u;
    // (staged here, under the barrier: a dependent global load per 64 accumulators in the write-back loop cost the launch
    //  a memory latency per iteration)
    if (threadIdx.x < eg.n_acc / WAVE) s_chunk_core[threadIdx.x] = (uint8_t) (im.ev_chunk_core[(eg.slot0 >> 6) + threadIdx.x] - eg.core0);
    __syncthreads();
    const uint32_t tile0 = seg * im.ev_seg_tiles, tile1 = min(tile0 + im.ev_seg_tiles, im.ev_tiles);
    // this group's table entries: every msn-th (the host picks the table, i.e. the instantiation, by the step's activity)
    const unsigned long long *mgroup = SPARSE ? im.ev_meta_n + g : im.ev_meta + (size_t) g * im.n_global_slots;
    const uint32_t msn = SPARSE ? im.ev_groups : 1u;
    uint16_t *list = s_list[wave];
    const uint32_t j = lane / LPB, q = lane % LPB; // this lane: neuron j of the batch, unit q (+ LPB, ...) of its block
    uint32_t msg_cnt[(16 + LPB - 1) / LPB];        // messages to core q, q + LPB, ... of the group, over this lane's neurons
#pragma unroll
    for (uint32_t m = 0; m < (16 + LPB - 1) / LPB; m++) msg_cnt[m] = 0u;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    auto add8 = [&](const uint4 &w4) {
        const uint32_t d4[4] = {w4.x, w4.y, w4.z, w4.w};
        uint32_t wv[8];
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            wv[2 * h] = s_lut16[d4[h] & CODE_MASK];
            wv[2 * h + 1] = s_lut16[(d4[h] >> 16) & CODE_MASK];
        }
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            __hip_atomic_fetch_add((lds_u32 *) &s_acc[(d4[h] & 0xffffu) >> CODE_BITS], wv[2 * h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add((lds_u32 *) &s_acc[d4[h] >> (16 + CODE_BITS)], wv[2 * h + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    struct Meta
    {
        unsigned long long m0;
        bool have;
    };
    struct Words
    {
        uint4 w[UPL]; // UPL units per lane and batch: lane q of a block holds units q, q + LPB, ...
        uint32_t unit0, units;
    };
    const uint4 *words = reinterpret_cast<const uint4 *>(im.ev_words);
    // list entries: source slot relative to the first slot of tile `list_tile0` (16 bits: the list is drained before it spans
    // 64 tiles)
    auto drain = [&](uint32_t len, uint32_t list_slot0) {
        wave_lds_fence(); // the list entries the lanes wrote
        auto fetch_meta = [&](uint32_t b, Meta &m) {
            m.have = b + j < len;
            const uint32_t f = list_slot0 + (uint32_t) list[m.have ? b + j : b];
            m.m0 = mgroup[(size_t) f * msn];
        };
        auto fetch_words = [&](const Meta &m, Words &w) {
            w.units = m.have ? (uint32_t) (m.m0 >> 32) & 0xffffu : 0u;
            w.unit0 = (uint32_t) m.m0;
            // (lanes without a unit read the block's first unit -- or, for an empty block, whatever follows: the array is padded)
#pragma unroll
            for (uint32_t u = 0; u < (uint32_t) UPL; u++)
                w.w[u] = ev_load16(words + (size_t) w.unit0 + (q + u * LPB < w.units ? q + u * LPB : 0u));
            if (m.have)
            {
                const uint32_t mask = (uint32_t) (m.m0 >> 48);
#pragma unroll
                for (uint32_t mm = 0; mm < (16 + LPB - 1) / LPB; mm++) msg_cnt[mm] += (mask >> (q + mm * LPB)) & 1u;
            }
        };
        Meta ma;
        Words wb;
        fetch_meta(0u, ma);
        fetch_words(ma, wb);
        if (NB < len) fetch_meta(NB, ma);
        for (uint32_t b = 0; b < len; b += NB) // (wave-uniform bounds)
        {
            const Words wc = wb;
            if (b + NB < len) fetch_words(ma, wb);
            if (b + 2u * NB < len) fetch_meta(b + 2u * NB, ma);
#pragma unroll
            for (uint32_t u = 0; u < (uint32_t) UPL; u++)
                if (q + u * LPB < wc.units) add8(wc.w[u]);
            // blocks of more than LPB x UPL units: the rest, LPB units at a time
            for (uint32_t u = q + UPL * LPB; __ballot(u < wc.units) != 0ull; u += LPB)
                if (u < wc.units) add8(ev_load16(words + (size_t) wc.unit0 + u));
        }
        wave_lds_fence(); // the list is rewritten
    };
    uint32_t len = 0, list_tile0 = tile0;
    // the wavefront's tiles, four at a time: their bitmap words are loaded together (one memory round trip, not four)
    constexpr uint32_t TB = 4;
    for (uint32_t tb = tile0 + wave; tb < tile1; tb += TB * (uint32_t) WAVES) // (wave-uniform bounds)
    {
        uint32_t w4[TB];
#pragma unroll
        for (uint32_t u = 0; u < TB; u++)
        {
            const uint32_t tile = tb + u * (uint32_t) WAVES;
            w4[u] = st.bits_global[(tile < tile1 ? tile : tb) * (EV_TILE / 32u) + (lane >> 1)];
        }
#pragma unroll
        for (uint32_t u = 0; u < TB; u++)
        {
            const uint32_t tile = tb + u * (uint32_t) WAVES;
            if (tile >= tile1) break;
            if (len > 0u && tile - list_tile0 >= 64u)
            {
                drain(len, list_tile0 * EV_TILE);
                len = 0;
            }
            if (len == 0u) list_tile0 = tile;
            // 16 slots per lane: lanes 2i and 2i + 1 share a word of the bitmap
            uint32_t half = (w4[u] >> (16u * (lane & 1u))) & 0xffffu;
            const uint32_t cnt = (uint32_t) __popc(half);
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, WAVE - 1);
            uint32_t pos = len + incl - cnt;
            const uint32_t rel0 = (tile - list_tile0) * EV_TILE + lane * 16u;
            while (half != 0u)
            {
                list[pos++] = (uint16_t) (rel0 + (uint32_t) __builtin_ctz(half));
                half &= half - 1u;
            }
            len += total;
            if (len >= EV_DRAIN_AT)
            {
                drain(len, list_tile0 * EV_TILE);
                len = 0;
            }
        }
    }
    if (len > 0u) drain(len, list_tile0 * EV_TILE);
    // messages per core of the group: lanes with the same q hold counts of the same cores
#pragma unroll
    for (uint32_t mm = 0; mm < (16 + LPB - 1) / LPB; mm++)
        if (msg_cnt[mm] != 0u && q + mm * LPB < 16u) atomicAdd(&s_msgs[q + mm * LPB], msg_cnt[mm]);
    __syncthreads(); // every wavefront's additions are in the accumulators
    // Every accumulator of the group, touched or not, goes to this segment's row of the partials (plain coalesced stores: the
    // segments of a group share its neurons, and global fp64 atomics run at a fifth of the store rate); the next neuron launch
    // adds a neuron's rows up.  Event counts per core: the accumulators hold them.
    uint32_t *part = st.ev_part + (size_t) seg * im.n_slots + eg.slot0;
    for (uint32_t i0 = wave * WAVE; i0 < eg.n_acc; i0 += BLOCK) // (n_acc: a multiple of 64 -> wave-uniform)
    {
        const uint32_t v = s_acc[i0 + lane];
        part[i0 + lane] = v;
        // count * 2^shift + sum of weights, |sum| < 2^(shift-1) (proven by the host per segment and accumulator)
        const uint32_t events = (v + (1u << (im.ev_shift - 1))) >> im.ev_shift;
        const long long ev = wave_sum((long long) events);
        if (lane == 0 && ev != 0) atomicAdd(&s_events[s_chunk_core[i0 >> 6]], (uint32_t) ev);
    }
    __syncthreads();
    if (threadIdx.x < eg.n_cores)
    {
        uint32_t *cnt = st.push_core_cnt + ((size_t) (uint32_t) (done % 3) * im.n_cores + eg.core0 + threadIdx.x) * 2u;
        if (s_msgs[threadIdx.x] != 0u) atomicAdd(&cnt[0], s_msgs[threadIdx.x]);
        if (s_events[threadIdx.x] != 0u) atomicAdd(&cnt[1], s_events[threadIdx.x]);
    }
}

// What a step delivered by events left in the partial rows, folded into the time-step buffer (state export, or any other
// reader of the buffer on the host): one thread per slot.
__global__ void event_fold_kernel(DevImage im, DevState st, long long t_done)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= im.n_slots) return;
    long long tot = 0;
    uint32_t any = 0;
    for (uint32_t q = 0; q < EV_MAX_SEGMENTS; q++)
    {
        const uint32_t v = st.ev_part[(size_t) q * im.n_slots + g];
        const uint32_t n_ev = (v + (1u << (im.ev_shift - 1))) >> im.ev_shift;
        tot += (long long) (int) (v - (n_ev << im.ev_shift));
        any |= v;
    }
    if (any == 0u) return;
    const size_t gi = (size_t) ((t_done + 1) % im.ring_slots) * im.n_slots + g;
    st.ring[gi] = (double) tot;
    st.ring_valid[gi] = 1;
}

// ---------------------------------------------------------------------------------------
// K2m: cores whose SOMA is part of the message pipeline (buffer inside the soma unit or before axon_out; DevImage::msg_*).
//
// The reference runs, for every synaptic event of such a core and in delivery order, synapse -> dendrite -> soma
// (process_message / execute_pipeline, src/chip.cpp:738-789; build_message_processing_pipeline, src/mapped.cpp:27-58): the
// `accumulator` dendrite returns the running sum of the step's currents (src/models.cpp:71-94) and the TrueNorth soma is
// updated with it -- leak, bias, input, threshold and reset, once PER EVENT (src/models.cpp:724-830).  A neuron's events
// are its inbound synapses whose source fired, in delivery order, and nothing couples two neurons of the core: one lane
// per post-synaptic neuron walks its own list against the step's spike bitmap.  grid = 64-slot chunks of these cores,
// block = 64, launched after the step's delivery:
//   * potential and status as the LAST update left them (what get_status reports at the end of a step; with the buffer
//     before axon_out it is also what the NEXT neuron launch sends spikes for: SANAFE_SOMA_PERSIST);
//   * the step's spike RECORD row of the chunk = neurons whose final status is `fired` (the reference's traces read the
//     status at the end of the step, src/pymodule.cpp:549-706); the bitmap that was delivered is left alone;
//   * per core: synaptic events, updates that fired, messages (the first chunk of a core walks its axon list) -- integers;
//     msgsoma_finish_kernel prices them once per core into the step's partials and the core's processing delay.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) msgsoma_kernel(DevImage im, DevState st, uint32_t *slog /* spike-record row of the step, or NULL */)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t k = im.msg_chunk_core[blockIdx.x];
    const uint32_t slot0 = im.msg_chunk_slot0[blockIdx.x], slot = slot0 + lane;
    const MsgCoreDev mcd = im.msg_core_dev[k];
    const uint32_t cls = im.slot_cls[slot];
    const bool live = (cls & 7u) != SANAFE_SOMA_NONE;
    const uint32_t b = live ? im.msg_ptr[slot] : 0u, e = live ? im.msg_ptr[slot + 1u] : 0u;
    sanafe_hip_soma_class p{};
    if (live) p = im.soma_classes[cls >> 16];
    const double bias = live ? im.slot_bias[slot] : 0.0;
    double v = live ? st.v[slot] : 0.0, acc = 0.0;
    uint32_t n_events = 0, n_fired = 0;
    int status = 0;
    for (uint32_t i = b; i < e; i++)
    {
        const uint32_t pre = im.msg_pre[i];
        if (!((st.bits_global[pre >> 5] >> (pre & 31u)) & 1u)) continue;
        acc = acc + im.msg_w[i]; // AccumulatorModel::update: cleared at the step's first call, then the running sum
        // TrueNorthModel::update with an input current: never idle
        status = 2;
        if (p.leak_towards_zero)
        {
            if (v > 0.0) v -= p.leak_decay;
            else if (v < 0.0) v += p.leak_decay;
        }
        else v += p.leak_decay;
        v += bias;
        v += acc;
        if (v >= p.threshold)
        {
            if (p.reset_mode == SANAFE_RESET_HARD) v = p.reset;
            else if (p.reset_mode == SANAFE_RESET_SOFT) v -= p.threshold;
            else if (p.reset_mode == SANAFE_RESET_SATURATE) v = p.threshold;
            status = 3;
        }
        else if (v <= p.reverse_threshold)
        {
            if (p.reverse_reset_mode == SANAFE_RESET_HARD) v = p.reverse_reset;
            else if (p.reverse_reset_mode == SANAFE_RESET_SOFT) v += p.reverse_threshold;
            else if (p.reverse_reset_mode == SANAFE_RESET_SATURATE) v = p.reverse_threshold;
        }
        n_events++;
        if (status == 3)
        {
            n_fired++;
            atomicAdd(&st.msg_ax_fired[im.msg_ax[i]], 1u); // (what the message's processing delay depends on: detailed timing)
        }
    }
    int final_status = 0;
    if (live)
    {
        if (n_events != 0u)
        {
            st.v[slot] = v;
            st.status[slot] = (uint8_t) status;
            final_status = status;
        }
        else final_status = st.status[slot]; // what the neuron loop left (or, before axon_out, an earlier step's events)
    }
    const unsigned long long fired_now = __ballot(final_status == 3);
    if (slog != nullptr && lane == 0)
    {
        slog[slot0 >> 5] = (uint32_t) fired_now;
        slog[(slot0 >> 5) + 1u] = (uint32_t) (fired_now >> 32);
    }
    const long long ev = wave_sum((long long) n_events), fi = wave_sum((long long) n_fired);
    if (lane == 0)
    {
        if (ev != 0) atomicAdd(&st.msg_cnt[k * 4u + 0u], (uint32_t) ev);
        if (fi != 0) atomicAdd(&st.msg_cnt[k * 4u + 1u], (uint32_t) fi);
    }
    if (blockIdx.x == mcd.first_chunk) // the core's messages: its inbound axons whose source fired
    {
        uint32_t cnt = 0;
        for (uint32_t a = mcd.ax_beg + lane; a < mcd.ax_end; a += WAVE)
        {
            const uint32_t pre = im.msg_ax_pre[a];
            cnt += (st.bits_global[pre >> 5] >> (pre & 31u)) & 1u;
        }
        const long long msgs = wave_sum((long long) cnt);
        if (lane == 0) st.msg_cnt[k * 4u + 2u] = (uint32_t) msgs;
    }
}

// One thread per such core, after msgsoma_kernel: the step's counts priced with the core's default costs
// (src/pipeline.hpp:511-731) into the partial of the core's first neuron workgroup -- unit energies by role, soma-activity
// counters -- and the core's message-processing delay (axon-in latency per message + the units' latencies per event).
__global__ void msgsoma_finish_kernel(DevImage im, DevState st, int parity, uint16_t *fired_log_row /* or NULL */)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    // per message: how many of its synaptic events made the soma fire -> the step's row of the log (recorded runs), cleared
    for (uint32_t a = k; a < im.n_msg_axons; a += gridDim.x * blockDim.x)
    {
        if (fired_log_row != nullptr) fired_log_row[a] = (uint16_t) min(st.msg_ax_fired[a], 65535u);
        st.msg_ax_fired[a] = 0u;
    }
    if (k >= im.n_msg_cores) return;
    const MsgCoreDev mcd = im.msg_core_dev[k];
    const sanafe_hip_msg_core_costs &c = mcd.costs;
    const double ev = (double) st.msg_cnt[k * 4u + 0u], fi = (double) st.msg_cnt[k * 4u + 1u], msgs = (double) st.msg_cnt[k * 4u + 2u];
    WgPart *cp = st.wg_part + ((size_t) parity * im.n_wgs + im.core_wg_beg[mcd.core]) * PARTS_PER_WG;
    cp->e_syn += ev * c.synapse_energy;
    cp->e_dend += ev * c.dendrite_energy;
    cp->e_soma += ev * (c.soma_energy[0] + c.soma_energy[1]) + fi * c.soma_energy[2];
    cp->updated += (long long) st.msg_cnt[k * 4u + 0u];
    cp->fired += (long long) st.msg_cnt[k * 4u + 1u];
    st.host_proc[(size_t) parity * im.n_cores + mcd.core] = msgs * c.axon_in_latency +
            ev * ((c.synapse_latency + c.dendrite_latency) + (c.soma_latency[0] + c.soma_latency[1])) + fi * c.soma_latency[2];
    st.msg_cnt[k * 4u + 0u] = st.msg_cnt[k * 4u + 1u] = st.msg_cnt[k * 4u + 2u] = 0u;
}

// ---------------------------------------------------------------------------------------
// K2o: ORDERED spike delivery, for chips with non-integer weights (syn_format 8).
//
// fp64 addition does not associate: the reference adds a step's synaptic currents into an accumulator one by one, in
// delivery order -- source core, source neuron, connection (src/chip.cpp:661-690, 748-761; `value_or(0.0) + current`,
// src/models.cpp:71-131) -- and any other association can flip a threshold-borderline spike.  The streaming kernel
// above adds with LDS / global atomics in arrival order, which is exact only for integers.  Here every accumulator
// (post neuron x delay value) is OWNED BY ONE LANE that walks the accumulator's own list of (pre slot, weight) in the
// reference's order and folds the weights of the pre neurons that spiked -- starting from the value the delay ring
// already holds (charge that earlier steps sent to the same future step): the same additions in the same order as the
// reference, so potentials and spikes are bit-identical to it and from run to run.  No atomics, no slices.
//
//   grid = ord_wgs, block = 256 = 4 independent wavefronts (no barrier after the staging), dynamic LDS = the whole spike
//   bitmap (LDS_BITS) -- a probe is one ds_read.  Every wavefront
//     1. folds ONE group of 64 accumulators, longest lists first.  The 64 lists of a group lie side by side
//        ([row][lane]): a row is one coalesced load, ORD_UNROLL loads are in flight while the previous ORD_UNROLL are
//        folded.  HBM-bound: every entry is read once per step, 4 bytes with a weight dictionary (<= 32 distinct
//        values; four rows per 16-byte load), 12 without;
//     2. then takes whole delivery slices (its number, + the wavefronts of the launch, ...): the processing-delay sum of
//        the slice's spiking axons (the simple timing model's per-core message processing time, src/chip.cpp:738-764)
//        from the axon records, runs of 8 chunks in flight, summed in a fixed order.
// ---------------------------------------------------------------------------------------
template <bool DICT, bool LDS_BITS, bool HAS_DELAY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) // (4 workgroups per CU hold the bitmap in LDS)
ordered_deliver_kernel(DevImage im, DevState st, long long done /* steps simulated before this one */)
{
    __shared__ double s_lut[DICT ? 32 : 1];
    uint32_t *sb = reinterpret_cast<uint32_t *>(deliver_lds);
    const uint32_t n_words = im.n_global_slots / 32;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    if (DICT && threadIdx.x < 32) s_lut[threadIdx.x] = im.weight_lut[threadIdx.x];
    if (LDS_BITS)
    {
        for (uint32_t i = threadIdx.x; i < n_words; i += 256) sb[i] = st.bits_global[i];
        if (threadIdx.x == 0) sb[n_words] = 0u; // the bit padding entries point at
    }
    if (DICT || LDS_BITS) __syncthreads();
    // (bits_global carries one zero word past its end for the same purpose)
    auto spiked = [&](uint32_t pre) -> bool { return ((LDS_BITS ? sb[pre >> 5] : st.bits_global[pre >> 5]) >> (pre & 31u)) & 1u; };
    const long long t = done + 1;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t) wave; // this wavefront among all of the launch
    if (gw < im.ord_groups)
    {
        const uint32_t g = gw;
        const OrdGroup og = im.ord_group[g];
        const uint32_t slot = im.ord_lane_slot[(size_t) g * WAVE + lane];
        const uint32_t d = im.ord_lane_delay[(size_t) g * WAVE + lane];
        const bool live = slot != 0xffffffffu;
        const uint32_t kind = (HAS_DELAY && live) ? (im.slot_cls[slot] >> 3) & 7u : (uint32_t) SANAFE_IN_BUFFERED;
        const bool taps = kind == SANAFE_IN_TAPS, gated = kind == SANAFE_IN_GATED;
        const uint32_t r0 = (uint32_t) ((t + 1) % (long long) im.ring_slots); // scalar
        uint32_t wrow = r0 + d + (gated ? 1u : 0u);
        while (wrow >= im.ring_slots) wrow -= im.ring_slots;
        const size_t gi = (size_t) wrow * im.n_slots + (live ? slot : 0u);
        // the accumulator continues what earlier steps sent to the same future step: value_or(0.0) + current
        // (consumed entries are left at 0.0 by the neuron launch)
        double acc = (live && !taps) ? st.ring[gi] : 0.0;
        bool touched = false;
        if constexpr (DICT)
        {
            // 4-byte entries (pre slot | weight code << 27), four consecutive rows of a lane side by side: one 16-byte load per
            // lane brings four entries, ORD_UNROLL loads = ORD_DICT_ROWS rows are in flight while the previous ones are folded
            const uint4 *pp = reinterpret_cast<const uint4 *>(im.ord_pre + og.off) + lane; // block b of 4 rows: pp[b * 64]
            uint4 e[ORD_UNROLL];
#pragma unroll
            for (int u = 0; u < ORD_UNROLL; u++) e[u] = load_stream16(pp + (size_t) u * WAVE);
            for (uint32_t r = 0; r < og.rows; r += ORD_DICT_ROWS)
            {
                uint4 ce[ORD_UNROLL];
#pragma unroll
                for (int u = 0; u < ORD_UNROLL; u++) ce[u] = e[u];
                // the next rows, unconditionally (past the end: these again)
                const size_t nb = (size_t) (((r + ORD_DICT_ROWS < og.rows) ? r + ORD_DICT_ROWS : r) / 4u) * WAVE;
#pragma unroll
                for (int u = 0; u < ORD_UNROLL; u++) e[u] = load_stream16(pp + nb + (size_t) u * WAVE);
                // eight entries at a time: their bitmap probes and dictionary reads in flight together, then the fold
#pragma unroll
                for (int h = 0; h < ORD_UNROLL; h += 2)
                {
                    const uint32_t q[8] = {ce[h].x, ce[h].y, ce[h].z, ce[h].w, ce[h + 1].x, ce[h + 1].y, ce[h + 1].z, ce[h + 1].w};
                    bool f[8];
                    double wv[8];
#pragma unroll
                    for (int j = 0; j < 8; j++)
                    {
                        f[j] = spiked(q[j] & ((1u << ORD_PRE_BITS) - 1u));
                        wv[j] = s_lut[q[j] >> ORD_PRE_BITS];
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) // the fold itself: strictly in list order
                        if (f[j])
                        {
                            acc += wv[j];
                            touched = true;
                        }
                }
            }
        }
        else
        {
            const uint32_t *pp = im.ord_pre + og.off + lane;
            const double *wp = im.ord_w + og.off + lane;
            uint32_t e[ORD_UNROLL];
            double w[ORD_UNROLL];
#pragma unroll
            for (int u = 0; u < ORD_UNROLL; u++)
            {
                e[u] = __builtin_nontemporal_load(pp + (size_t) u * WAVE);
                w[u] = __builtin_nontemporal_load(wp + (size_t) u * WAVE);
            }
            for (uint32_t r = 0; r < og.rows; r += ORD_UNROLL)
            {
                uint32_t ce[ORD_UNROLL];
                double cw[ORD_UNROLL];
#pragma unroll
                for (int u = 0; u < ORD_UNROLL; u++)
                {
                    ce[u] = e[u];
                    cw[u] = w[u];
                }
                // the next rows, unconditionally (past the end: these again), while the current ones are folded
                const size_t nr = (size_t) ((r + ORD_UNROLL < og.rows) ? r + ORD_UNROLL : r) * WAVE;
#pragma unroll
                for (int u = 0; u < ORD_UNROLL; u++)
                {
                    e[u] = __builtin_nontemporal_load(pp + nr + (size_t) u * WAVE);
                    w[u] = __builtin_nontemporal_load(wp + nr + (size_t) u * WAVE);
                }
                bool f[ORD_UNROLL];
#pragma unroll
                for (int u = 0; u < ORD_UNROLL; u++) f[u] = spiked(ce[u]);
#pragma unroll
                for (int u = 0; u < ORD_UNROLL; u++) // the fold itself: strictly in list order
                    if (f[u])
                    {
                        acc += cw[u];
                        touched = true;
                    }
            }
        }
        if (live && touched)
        {
            if (taps)
            {
                // row d is tap d of the neuron's dendrite: taps_kernel integrates it after this launch
                st.tap_in[(size_t) im.slot_aux[slot] * 8u + d] = acc;
                st.arrived[slot] = 1;
            }
            else
            {
                st.ring[gi] = acc;
                st.ring_valid[gi] = 1;
                if (gated) st.arrived[slot] = 1;
            }
        }
    }
    // ---- processing-delay sums: every wavefront takes whole delivery slices, gw, gw + (waves of the launch), ... ----
    for (uint32_t sl = gw; sl < im.ord_walk_slices; sl += gridDim.x * 4u)
    {
    const SliceDesc sd = im.slice_desc[sl];
    const uint32_t n_ax = sd.n_ax;
    const bool compact = sd.mode != 0;
    const unsigned char *rec = im.ax_bytes + sd.rec_off;
    const uint32_t *chunk_pre0 = im.chunk_pre0 + sd.chunk0;
    double proc = 0.0;
    if (compact)
    {
        // runs of 8 consecutive 256-axon chunks per wave, their record loads in flight together (one address, eight
        // immediate offsets; the record array ends in 4 KB of padding, reads past the slice are masked below).  One latency
        // class per compact slice: messages and events are counted in integers and priced once.
        constexpr uint32_t RUN = 8;
        uint32_t n_msgs = 0, n_events = 0;
        // (the records and first pre slots of the NEXT run are loaded before this run is decoded: two runs in flight)
        auto load_run = [&](uint32_t r0, unsigned long long (&qq)[RUN], uint32_t &pre0) {
            const uint32_t n_here = min(RUN, (n_ax - r0 + WAVE_CHUNK - 1u) / WAVE_CHUNK);
            pre0 = chunk_pre0[r0 / WAVE_CHUNK + min((uint32_t) lane, n_here - 1u)]; // lane j: first pre slot of chunk j
            const unsigned long long *rec_lane = reinterpret_cast<const unsigned long long *>(rec + 2ull * (r0 + (uint32_t) lane * AX_PER_THREAD));
#pragma unroll
            for (uint32_t j = 0; j < RUN; j++) qq[j] = rec_lane[j * (WAVE_CHUNK * 2u / 8u)];
        };
        unsigned long long qn[RUN];
        uint32_t pre0_next = 0;
        if (n_ax > 0) load_run(0u, qn, pre0_next);
        for (uint32_t r0 = 0; r0 < n_ax; r0 += RUN * WAVE_CHUNK) // wave-uniform bounds
        {
            const uint32_t n_here = min(RUN, (n_ax - r0 + WAVE_CHUNK - 1u) / WAVE_CHUNK);
            const uint32_t pre0_lane = pre0_next;
            unsigned long long q[RUN];
#pragma unroll
            for (uint32_t j = 0; j < RUN; j++) q[j] = qn[j];
            const uint32_t r1 = r0 + RUN * WAVE_CHUNK;
            load_run(r1 < n_ax ? r1 : r0, qn, pre0_next); // unconditional (past the end: this run again)
#pragma unroll
            for (uint32_t j = 0; j < RUN; j++)
            {
                const uint32_t a0 = r0 + j * WAVE_CHUNK + (uint32_t) lane * AX_PER_THREAD;
                const uint32_t n_valid = (j < n_here && a0 < n_ax) ? min((uint32_t) AX_PER_THREAD, n_ax - a0) : 0u;
                if (n_valid < (uint32_t) AX_PER_THREAD) q[j] &= (1ull << (16u * n_valid)) - 1ull; // records past the end: no advance
                uint32_t dl[AX_PER_THREAD];
#pragma unroll
                for (int k = 0; k < AX_PER_THREAD; k++) dl[k] = (uint32_t) (q[j] >> (16 * k)) & 0xffu;
                const uint32_t lane_d = dl[0] + dl[1] + dl[2] + dl[3];
                uint32_t pre = (uint32_t) __builtin_amdgcn_readlane((int) pre0_lane, (int) j) + wave_inclusive_scan(lane_d) - lane_d;
#pragma unroll
                for (int k = 0; k < AX_PER_THREAD; k++)
                {
                    pre += dl[k];
                    if ((uint32_t) k < n_valid && spiked(pre))
                    {
                        n_msgs += 1u;
                        n_events += (uint32_t) (q[j] >> (16 * k + 8)) & 0xffu;
                    }
                }
            }
        }
        proc = (double) n_events * sd.slice_lat + (double) n_msgs * sd.ain_lat;
    }
    else
    for (uint32_t c0 = 0; c0 < n_ax; c0 += WAVE_CHUNK) // wave-uniform bounds
    {
        const uint32_t a0 = c0 + (uint32_t) lane * AX_PER_THREAD;
        const unsigned long long *wide = reinterpret_cast<const unsigned long long *>(rec);
#pragma unroll
        for (int k = 0; k < AX_PER_THREAD; k++)
            if (a0 + k < n_ax)
            {
                const unsigned long long r = wide[a0 + k];
                const uint32_t nsyn = (uint32_t) ((r >> 32) & 0xffffu), lcls = (uint32_t) ((r >> 48) & 0xffu);
                if (spiked((uint32_t) r))
                    proc += (lcls != 255u) ? sd.ain_lat + (double) nsyn * im.lat_class[lcls] : im.ax_proc_delay[sd.a_beg + a0 + k];
            }
    }
    proc = wave_sum(proc); // fixed order
    if (lane == 0) st.slice_proc[(size_t) (done & 1) * im.n_slices + sd.slice_id] = proc;
    }
}

// ---------------------------------------------------------------------------------------
// K3: per-step reduction in two levels (see PendStep), every association fixed: results are reproducible
// run to run.  sim_calculate_ts_energy, sim_update_ts_counters, schedule_messages_timestep_simple
// (src/chip.cpp:1028-1051, 1171-1261; src/schedule.cpp:61-102), update_run_data (src/chip.cpp:462-475).
// ---------------------------------------------------------------------------------------
// Sum over the four lanes of a quad, in every lane: (x0 + x1) + (x2 + x3) -- two DPP quad_perm steps, fixed order.
__device__ __forceinline__ double quad_sum(double x)
{
#define SANAFE_QSTEP(CTRL)                                                                       \
    {                                                                                             \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false); \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false); \
        x += __hiloint2double(hi, lo);                                                            \
    }
    SANAFE_QSTEP(0xb1) // quad_perm:[1,0,3,2]
    SANAFE_QSTEP(0x4e) // quad_perm:[2,3,0,1]
#undef SANAFE_QSTEP
    return x;
}
__device__ __forceinline__ long long quad_sum(long long x)
{
#define SANAFE_QSTEP(CTRL)                                                                                                      \
    {                                                                                                                            \
        const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) x, CTRL, 0xf, 0xf, false);               \
        const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) ((unsigned long long) x >> 32), CTRL, 0xf, 0xf, false); \
        x += (long long) (((unsigned long long) hi << 32) | lo);                                                                 \
    }
    SANAFE_QSTEP(0xb1)
    SANAFE_QSTEP(0x4e)
#undef SANAFE_QSTEP
    return x;
}

// Level 1: one wavefront folds L1_CORES consecutive cores, FOUR LANES PER CORE.  Per core: its neuron workgroups' partials --
// lane q of the core's quad takes every fourth one, so a core's (up to 16) partials are four loads in flight per lane instead
// of one lane walking them one memory latency after the other: that walk sat on the critical path of the neuron launch it
// rides in (C2: 10.5 instead of 7 us per launch) -- the generation-delay sum of its messages incl. the placeholder
// (src/chip.cpp:640-652, 727-728, 821-823; src/schedule.cpp:81) and the processing-delay sum of its delivery slices, all in
// a fixed order: lane-strided partial sums, then (q0 + q1) + (q2 + q3).
__device__ void reduce_l1(const DevImage &im, const DevState &st, int parity, uint32_t group, int push_buf, int pushed_step)
{
    // (The order below keeps few values live at a time -- counters first, each folded and stored before the next; then the
    //  energies -- because this code shares the register budget of the neuron kernels it rides in: 64 for the uniform
    //  TrueNorth instantiation.)
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t q = (uint32_t) lane & 3u;
    const uint32_t c = group * L1_CORES + ((uint32_t) lane >> 2);
    const bool have = c < im.n_cores;     // (the same for the four lanes of a quad)
    const bool owner = have && q == 0u;   // the lane that speaks for the core in the fold over the cores
    const WgPart *part = st.wg_part + (size_t) parity * im.n_wgs * PARTS_PER_WG;
    const uint32_t w0 = have ? im.core_wg_beg[c] * PARTS_PER_WG : 0u, w1 = have ? im.core_wg_beg[c + 1] * PARTS_PER_WG : 0u;
    GroupPart *out = &st.group_part[(size_t) parity * im.n_groups + group];
    // ---- the counters of the core's wavefront partials: lane q takes every fourth partial ----
    uint32_t n_upd, n_fired, n_counted, n_packets;
    {
        long long upd = 0, fired = 0, packets = 0, hops = 0, events = 0, counted = 0;
#pragma unroll 1
        for (uint32_t w = w0 + q; w < w1; w += 4u) // (not unrolled: four lanes per core already keep four partials in flight)
        {
            const WgPart &p = part[w];
            upd += p.updated;
            fired += p.fired;
            packets += p.packets;
            hops += p.hops;
            events += p.events;
            counted += p.counted;
        }
        upd = quad_sum(upd);
        fired = quad_sum(fired);
        packets = quad_sum(packets);
        counted = quad_sum(counted);
        n_upd = (uint32_t) upd; // (per core and step: they fit 32 bits)
        n_fired = (uint32_t) fired;
        n_counted = (uint32_t) counted;
        n_packets = (uint32_t) packets;
        long long x = wave_sum(owner ? upd : 0LL);
        if (lane == 0) out->updated = x;
        x = wave_sum(owner ? fired : 0LL);
        if (lane == 0) out->fired = x;
        x = wave_sum(owner ? packets : 0LL);
        if (lane == 0) out->packets = x;
        hops = quad_sum(hops); // (every lane executes the quad exchange; then one lane per core counts)
        events = quad_sum(events);
        x = wave_sum(q == 0u ? hops : 0LL);
        if (lane == 0) out->hops = x;
        x = wave_sum(q == 0u ? events : 0LL);
        if (lane == 0) out->events = x;
    }
    asm volatile("" ::: "memory");
    // ---- processing delay: the core's delivery slices (lane q takes every fourth one, two running sums each), or the
    //      counters of the push path / the event kernel ----
    double proc = 0.0;
    const bool pushed = pushed_step != 0; // (the host's decision: the same value went to the step's own launches)
    if (!pushed)
    {
        const double *sp = st.slice_proc + (size_t) parity * im.n_slices;
        const uint32_t s0 = have ? im.core_slice_beg[c] : 0u, s1 = have ? im.core_slice_beg[c + 1] : 0u;
        double a0 = 0.0, a1 = 0.0;
        uint32_t s = s0 + q;
#pragma unroll 1
        for (; s + 4u < s1; s += 8u)
        {
            a0 += sp[s];
            a1 += sp[s + 4u];
        }
        if (s < s1) a0 += sp[s];
        proc = quad_sum(a0 + a1);
    }
    if (owner)
    {
        if (pushed)
        {
            // the push path / the event kernel counted this core's messages and events: integers times the core's constants
            uint32_t *cnt = st.push_core_cnt + ((size_t) push_buf * im.n_cores + c) * 2u;
            proc = (double) cnt[1] * im.core_event_lat[c] + (double) cnt[0] * im.core_ain_lat[c];
            cnt[0] = 0u;
            cnt[1] = 0u;
        }
        // a core that runs on the host has no delivery slices: its message-processing delay comes from the host's replay
        if (st.host_proc != nullptr) proc += st.host_proc[(size_t) parity * im.n_cores + c];
    }
    else proc = 0.0;
    {
        const double pmax = wave_max(proc);
        if (lane == 0) out->pmax = pmax;
    }
    asm volatile("" ::: "memory");
    // ---- the energies and the latency sum of the partials ----
    double e_soma = 0, e_dend = 0, e_syn = 0, e_net = 0, lat = 0;
#pragma unroll 1
    for (uint32_t w = w0 + q; w < w1; w += 4u)
    {
        const double *pd = reinterpret_cast<const double *>(part + w);
        const double2 d01 = *reinterpret_cast<const double2 *>(pd), d23 = *reinterpret_cast<const double2 *>(pd + 2);
        e_soma += d01.x;
        e_dend += d01.y;
        e_syn += d23.x;
        e_net += d23.y;
        lat += pd[4];
    }
    {
        e_syn = quad_sum(e_syn);
        e_net = quad_sum(e_net);
        double x = wave_sum(q == 0u ? e_syn : 0.0);
        if (lane == 0) out->e_syn = x;
        x = wave_sum(q == 0u ? e_net : 0.0);
        if (lane == 0) out->e_net = x;
    }
    e_soma = quad_sum(e_soma);
    e_dend = quad_sum(e_dend);
    lat = quad_sum(lat);
    double gen = 0.0;
    if (owner)
    {
        if (im.uni_costing && n_counted != 0u)
        {
            // default costing of the core's neurons (src/pipeline.hpp:574-731), from their counts by soma activity
            const sanafe_hip_cost_class &cc = im.uni_cost;
            const double n_all = (double) n_counted, n_f = (double) n_fired, n_u = (double) (n_upd - n_fired), n_i = (double) (n_counted - n_upd);
            e_soma += (n_i * cc.soma_energy[0] + n_u * cc.soma_energy[1]) + n_f * cc.soma_energy[2];
            e_dend += n_all * cc.dendrite_energy;
            lat += n_all * (0.0 + cc.dendrite_latency) + ((n_i * cc.soma_latency[0] + n_u * cc.soma_latency[1]) + n_f * cc.soma_latency[2]);
        }
        // generation delay of the core's messages incl. the placeholder (src/chip.cpp:640-652, 727-728, 821-823; src/schedule.cpp:81)
        gen = lat + (double) n_packets * im.core_axon_out_latency[c];
    }
    else e_soma = e_dend = 0.0;
    {
        double x = wave_sum(e_soma);
        if (lane == 0) out->e_soma = x;
        x = wave_sum(e_dend);
        if (lane == 0) out->e_dend = x;
        x = wave_max(gen);
        if (lane == 0) out->gmax = x;
    }
}

// Level 2: one wavefront folds the groups (lane = group, further groups in rounds of 64) into the Timestep
// totals, applies the simple timing model, accumulates RunData and writes the step record.
__device__ void reduce_l2(const DevImage &im, const DevState &st, const PendStep &prev)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const GroupPart *gp = st.group_part + (size_t) prev.parity * im.n_groups;
    double e_soma = 0, e_dend = 0, e_syn = 0, e_net = 0, gmax = 0, pmax = 0;
    long long upd = 0, fired = 0, packets = 0, hops = 0, events = 0;
#pragma unroll 1
    for (uint32_t g = (uint32_t) lane; g < im.n_groups; g += WAVE) // (not unrolled: it shares the neuron kernels' register budget)
    {
        const GroupPart p = gp[g];
        e_soma += p.e_soma;
        e_dend += p.e_dend;
        e_syn += p.e_syn;
        e_net += p.e_net;
        gmax = fmax(gmax, p.gmax);
        pmax = fmax(pmax, p.pmax);
        upd += p.updated;
        fired += p.fired;
        packets += p.packets;
        hops += p.hops;
        events += p.events;
    }
    e_soma = wave_sum(e_soma);
    e_dend = wave_sum(e_dend);
    e_syn = wave_sum(e_syn);
    e_net = wave_sum(e_net);
    gmax = wave_max(gmax);
    pmax = wave_max(pmax);
    upd = wave_sum(upd);
    fired = wave_sum(fired);
    packets = wave_sum(packets);
    hops = wave_sum(hops);
    events = wave_sum(events);
    if (lane == 0)
    {
        sanafe_hip_totals ts;
        ts.timesteps = 1;
        ts.spikes = events;
        ts.packets_sent = packets;
        ts.neurons_updated = upd;
        ts.neurons_fired = fired;
        ts.total_hops = hops;
        ts.synapse_energy = e_syn;
        ts.dendrite_energy = e_dend;
        ts.soma_energy = e_soma;
        ts.network_energy = e_net;
        ts.total_energy = ((e_net + e_syn) + e_dend) + e_soma;
        // multi-GPU: the largest per-core delay of THIS rank; the host library takes the maximum over the ranks
        // before adding the sync delay (record bit 2: keep the raw maximum in the step record)
        const double local_max = fmax(pmax, gmax);
        ts.sim_time = prev.simple_timing ? local_max + im.sync_delay : 0.0;
        sanafe_hip_totals r = *st.run; // update_run_data, src/chip.cpp:462-475
        r.timesteps += 1;
        r.spikes += ts.spikes;
        r.packets_sent += ts.packets_sent;
        r.neurons_updated += ts.neurons_updated;
        r.neurons_fired += ts.neurons_fired;
        r.total_hops += ts.total_hops;
        r.total_energy += ts.total_energy;
        r.synapse_energy += ts.synapse_energy;
        r.dendrite_energy += ts.dendrite_energy;
        r.soma_energy += ts.soma_energy;
        r.network_energy += ts.network_energy;
        r.sim_time += ts.sim_time;
        *st.run = r;
        if (prev.record)
        {
            ts.timesteps = *st.t + 1; // the record carries the timestep number
            st.step_log[prev.rec_index % st.log_cap] = ts;
            *st.rec = prev.rec_index + 1;
        }
        // the step's synaptic events, for the host's push / pull (event / stream) decision DECISION_LAG steps from now:
        // value first, then the step number it belongs to (the host waits for the number)
        // (count and step number in ONE 16-byte store to the pinned ring -- a single write on the bus, so the host never pairs a
        //  new number with an old count -- written THROUGH to system scope (sc0 sc1) so that the host sees it when this wavefront
        //  has stored it, not at some later cache write-back, and with no release fence: a system-scope release here writes the
        //  XCD's L2 back in the middle of the neuron launch, +3 us per step on the small configurations)
        if (st.host_events != nullptr && ((*st.t + 1) % DECISION_STRIDE) == 0)
        {
            const unsigned long long ev_u = (unsigned long long) events, n_u = (unsigned long long) (*st.t + 1);
            u32x4_t rec;
            rec.x = (uint32_t) ev_u;
            rec.y = (uint32_t) (ev_u >> 32);
            rec.z = (uint32_t) n_u;
            rec.w = (uint32_t) (n_u >> 32);
            long long *dst = st.host_events + 2 * ((*st.t + 1) % HOST_EVENT_RING);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(rec) : "memory");
        }
        if (st.delay_log != nullptr) st.delay_log[*st.t % st.delay_log_cap] = local_max;
        *st.t = *st.t + 1;
    }
}

// Device-side compaction of the logged neurons (potential / neuron traces, src/chip.cpp:1766-1831): after the
// neuron launch of a recorded step, the potentials of the `n_v` listed slots and the input currents of the `n_u`
// listed slots go into one row each of the state log -- no per-step host round trip, no full-state copy.
__global__ void state_log_kernel(const double *v, const double *icur, const uint32_t *slots_v, uint32_t n_v, const uint32_t *slots_u,
        uint32_t n_u, double *row_v, double *row_u)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_v) row_v[i] = v[slots_v[i]];
    else if (i - n_v < n_u) row_u[i - n_v] = icur[slots_u[i - n_v]];
}

// Flushes pending reductions when no further neuron launch follows: grid = n_reduce_wgs (or 1 for level 2 only).
__global__ void __launch_bounds__(REDUCE_BLOCK) reduce_kernel(DevImage im, DevState st, PendStep l1, PendStep l2)
{
    const int wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && wave == 0 && l2.valid) reduce_l2(im, st, l2);
    const uint32_t group = blockIdx.x * (REDUCE_BLOCK / WAVE) + (uint32_t) wave;
    if (l1.valid && group < im.n_groups) reduce_l1(im, st, l1.parity, group, l1.push_buf, l1.pushed);
}

// `taps` dendrites, after the delivery launch of step t (one thread per neuron): advance the RC line by one step
// (MultiTapModel1D::calculate_next_state, src/models.cpp:167-205 -- the reference catches up lazily at the first
// event, which is the same sequence of operations), add the charge delivered to each tap in this step, and hand
// tap 0 to the soma through the time-step buffer of step t+1 if an event reached the neuron (src/models.cpp:240-262).
__global__ void taps_kernel(DevImage im, DevState st, long long done /* steps simulated before this one */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= im.n_taps) return;
    const uint32_t taps = im.tap_count[i], g = im.tap_slot[i];
    double v[8], nv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = st.tap_v[(size_t) i * 8 + k];
    const double *tc = im.tap_tc + (size_t) i * 8, *sc = im.tap_sc + (size_t) i * 8;
    for (uint32_t k = 0; k < taps; k++) nv[k] = v[k] * tc[k];
    for (uint32_t s = 0; s < taps; s++)
    {
        if (s > 0)
        {
            const double c = v[s] * sc[s - 1];
            nv[s - 1] += c;
            nv[s] -= c;
        }
        if (s + 1 < taps)
        {
            const double c = v[s] * sc[s];
            nv[s + 1] += c;
            nv[s] -= c;
        }
    }
    for (uint32_t k = 0; k < taps; k++)
    {
        // the events of the step add to the advanced state one by one; their sum per tap arrives here
        v[k] = nv[k] + st.tap_in[(size_t) i * 8 + k];
        st.tap_in[(size_t) i * 8 + k] = 0.0;
        st.tap_v[(size_t) i * 8 + k] = v[k];
    }
    if (st.arrived[g] != 0)
    {
        st.arrived[g] = 0;
        const long long t = done + 1;
        const size_t gi = (size_t) ((t + 1) % im.ring_slots) * im.n_slots + g;
        st.ring[gi] = v[0];
        st.ring_valid[gi] = 1;
    }
}

__global__ void host_input_kernel(DevImage im, DevState st, uint32_t count, const uint32_t *slots, double *cur, uint8_t *has,
        long long done)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t g = slots[i];
    const long long t = done + 1;
    const size_t gi = (size_t) (t % im.ring_slots) * im.n_slots + g;
    const uint32_t inkind = (im.slot_cls[g] >> 3) & 7u;
    if (inkind == SANAFE_IN_ZERO)
    {
        has[i] = 1;
        cur[i] = 0.0;
        return;
    }
    if (inkind == SANAFE_IN_GATED)
    {
        const uint8_t valid = st.ring_valid[gi];
        const double value = st.ring[gi];
        if (valid)
        {
            st.ring[gi] = 0.0;
            st.ring_valid[gi] = 0;
        }
        const bool arr = st.arrived[g] != 0;
        st.arrived[g] = 0;
        has[i] = (arr && valid) ? 1 : 0;
        cur[i] = (arr && valid) ? value : 0.0;
        return;
    }
    if (inkind == SANAFE_IN_LAST)
    {
        has[i] = 1;
        double c = 0.0;
        const uint32_t last = st.ring_last[g];
        if (last != 0u)
        {
            uint32_t core = 0; // the slot's core: cores are few, slots of a core contiguous
            while (core + 1 < im.n_cores && im.core_nbase[core + 1] <= g) core++;
            const unsigned long long pos = im.core_syn_base[core] + (last - 1u);
            c = 0.0 + synapse_weight_at(im, pos);
            st.ring_last[g] = 0u;
        }
        cur[i] = c;
        return;
    }
    const uint8_t h = st.ring_valid[gi];
    has[i] = h;
    cur[i] = h ? st.ring[gi] : 0.0;
    if (h)
    {
        st.ring[gi] = 0.0;
        st.ring_valid[gi] = 0;
    }
}

// Status, energy and latency of the host-evaluated (plugin) somas of one step, folded into the partial of the
// first neuron workgroup of each neuron's core.  The neuron kernel skipped these slots entirely, so the
// dendrite's per-update cost of the neuron-processing pipeline (buffer before or inside the dendrite unit,
// src/pipeline.hpp:574-629) is added here as well.
// RAW (neurons of cores that run on the host, host/host_cores.cpp): only the status, the spike bit and the static totals of
// the spike; the core's unit energies, latencies and soma-activity counters arrive per core with host_core_costs_kernel.
__global__ void host_status_kernel(DevImage im, DevState st, uint32_t count, const uint32_t *slots, const uint8_t *status,
        const uint32_t *core, const double *energy, const double *latency, int parity, int raw)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t g = slots[i];
    const uint8_t s = status[i];
    st.status[g] = s;
    WgPart *cp = st.wg_part + ((size_t) parity * im.n_wgs + im.core_wg_beg[core[i]]) * PARTS_PER_WG;
    if (!raw)
    {
        const sanafe_hip_cost_class &cc = im.cost_classes[(im.slot_cls[g] >> 6) & 1023u];
        atomicAdd(&cp->e_soma, energy[i]);
        atomicAdd(&cp->e_dend, cc.dendrite_energy);
        atomicAdd(&cp->lat, (0.0 + cc.dendrite_latency) + latency[i]);
        if (s >= 2) atomicAdd((unsigned long long *) &cp->updated, 1ull);
    }
    if (s == 3)
    {
        const SpikeStatic ss = im.slot_spike[g];
        atomicOr(&st.bits_local[g >> 5], 1u << (g & 31u));
        if (!raw) atomicAdd((unsigned long long *) &cp->fired, 1ull);
        atomicAdd((unsigned long long *) &cp->packets, (unsigned long long) ss.packets);
        atomicAdd((unsigned long long *) &cp->hops, (unsigned long long) ss.hops);
        atomicAdd((unsigned long long *) &cp->events, (unsigned long long) ss.events);
        atomicAdd(&cp->e_net, ss.e_net);
        atomicAdd(&cp->e_syn, ss.e_syn);
        atomicAdd(&cp->e_dend, ss.e_dend);
    }
}

// What one timestep of a core that runs on the host adds to the step's totals (include/sanafe_hip.h:
// sanafe_hip_host_core_costs), after the delivery launch of the step: unit energies by role, the neuron pipelines' latency
// sum (generation delay), the soma-activity counters, and the core's message-processing delay for the simple timing model.
__global__ void host_core_costs_kernel(DevImage im, DevState st, uint32_t count, const sanafe_hip_host_core_costs *costs, int parity)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const sanafe_hip_host_core_costs k = costs[i];
    WgPart *cp = st.wg_part + ((size_t) parity * im.n_wgs + im.core_wg_beg[k.core]) * PARTS_PER_WG;
    cp->e_syn += k.synapse_energy; // (one thread per core; the neuron launch of the step wrote the partial long before)
    cp->e_dend += k.dendrite_energy;
    cp->e_soma += k.soma_energy;
    cp->lat += k.neuron_latency;
    cp->updated += k.neurons_updated;
    cp->fired += k.neurons_fired;
    st.host_proc[(size_t) parity * im.n_cores + k.core] = k.processing_delay;
}
