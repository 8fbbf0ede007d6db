// sanafe_hip.hip -- SANA-FE's per-timestep simulation loop for MI355X (gfx950, CDNA4).
//
// Implements the C ABI of include/sanafe_hip.h.  Two launches per timestep (one on push-only chips), all on one
// HIP stream, no host round trip between steps (the host reads the published event counts from pinned memory):
//
//   K1 neuron_kernel   one 256-thread workgroup per up to four 64-slot chunks of a simulated core, one wavefront per
//                      chunk; SoA neuron state, coalesced 8-byte loads; soma update (LIF / TrueNorth / input, incl.
//                      the host-generated stochastic value streams); wave ballot -> spike bitmap; cost / counter
//                      partials per wavefront.  On steps with few spikes the wavefront also DELIVERS its neurons'
//                      spikes (push delivery: static out-synapse lists, atomics into the next step's buffer row).
//                      The leading workgroups reduce the two previous steps (K3).
//                      Reference: process_neurons / process_neuron / execute_pipeline,
//                      src/chip.cpp:624-654, 710-736, 766-789; models src/models.cpp:441-903;
//                      default costing src/pipeline.hpp:631-731.
//   K2 deliver_kernel  one 256-thread workgroup (4 independent wavefronts) per delivery slice of a
//                      destination core: walks the core's static inbound-axon list against the spike bitmap
//                      ("pull": messages are never materialised) and adds the weights of the spiking axons'
//                      synapses into LDS accumulators before one write-back.  Axon records: a source bitmap per
//                      256-slot window (dense fan-in), 2-byte deltas, or 8-byte records; synapse words: 2-byte
//                      dictionary words or 4-byte words, streamed in runs of 8 chunks (every word is read once,
//                      in order, and says itself whether its axon spiked) or gathered (few spikes).
//                      Reference: process_messages / process_message, src/chip.cpp:656-764;
//                      AccumulatorModel / AccumulatorWithDelayModel src/models.cpp:71-131.
//   K2e event_deliver_kernel   steps in which few neurons fire (up to ~58 % on C3): a second, source-neuron-major copy of
//                      the 2-byte words; a workgroup owns the LDS accumulators of one group of destination cores and one
//                      segment of the source space, lists the neurons that fired and adds only their blocks -- work in
//                      proportion to the step's synaptic events, like the reference's loop.  The host picks the kernel (and
//                      which copy of the block table it reads) per step from the event counts the device publishes.
//   K2o ordered_deliver_kernel   chips with non-integer weights: one lane per accumulator folds the accumulator's own
//                      synapse list in the reference's delivery order (bit-equal fp64 sums, no atomics).
//   K2m msgsoma_kernel  cores whose SOMA is part of the message pipeline (buffer inside the soma unit / before axon_out):
//                      one lane per post-synaptic neuron walks its inbound synapses in delivery order and updates the
//                      TrueNorth soma once per synaptic event (src/mapped.cpp:27-58, src/chip.cpp:738-789).
//   K3 reduce_l1 / l2  fixed-order, two-level reduction of the per-wavefront partials and the slices' processing
//                      delays into the timestep totals, simple timing model, run totals, t += 1.  Rides in the
//                      leading workgroups of the NEXT two neuron launches; reduce_kernel flushes the last steps.
//                      Reference: sim_calculate_ts_energy, sim_update_ts_counters,
//                      schedule_messages_timestep_simple, src/chip.cpp:1028-1051, 1171-1261;
//                      src/schedule.cpp:61-102.
//
// No MFMA: the path is HBM-bound pointwise work plus an irregular gather / stream.
// Compile with -ffp-contract=off: the membrane arithmetic must round exactly like the
// reference's scalar C++ (no fused multiply-add).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sanafe_hip.h"

namespace
{
thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(SANAFE_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                                \
    } while (0)

#include "sanafe_kernels.hpp" // device code: constants, DevImage / DevState, the kernels

} // namespace

// ---------------------------------------------------------------------------------------
// Host side of the C ABI
// ---------------------------------------------------------------------------------------
struct sanafe_hip_chip
{
    int device{0};
    hipStream_t stream{nullptr};
    bool own_stream{true};
    bool has_delay{false}, force_delay_variant{false};
    long long t_host{0};      // timesteps launched so far (the device's *st.t trails it by the pending reduction)
    PendStep pend1{};         // the launched step that still needs level 1 of its reduction
    PendStep pend2{};         // the step before it: level 1 done (or launched), level 2 not yet launched
    std::vector<uint32_t> h_core_wg_beg, h_core_slice_beg; // host copies for sanafe_hip_read_core_delays
    std::vector<double> h_core_out_lat;
    uint64_t layout_bytes[SANAFE_HIP_LAYOUT_FIELDS]{};
    bool sub_accumulators{false}; // deliver_kernel<..., SUB>
    std::vector<double> weight_lut; // formats 6, 7: the chip's distinct weight values (<= 32)
    int acc_shift{0};               // format 7: see DevImage
    bool small_slices{false};       // every delivery slice holds at most two 256-axon chunks: 64-thread delivery workgroups
    // state log (record bit 3): potentials / input currents of listed slots, one row per recorded step
    uint32_t *d_log_slots_v{nullptr}, *d_log_slots_u{nullptr};
    uint32_t n_log_v{0}, n_log_u{0};
    double *d_state_log{nullptr}; // [state_log_cap][n_log_v + n_log_u]
    long long state_log_cap{0};
    int split_record{0};        // record bits of the split step (sanafe_hip_record_begin)
    long long split_index{0};   // record slot the next split step writes
    uint32_t n_local_slices{0}; // leading slice descriptors whose axons all start on this chip
    int neuron_model{0};      // soma model every live slot runs (SANAFE_SOMA_LIF / _TRUENORTH), 0 when they differ
    bool uni{false};          // every live slot carries the class word us.cls (UniformSoma)
    UniformSoma us{};
    std::vector<sanafe_hip_soma_class> h_soma_classes; // host copy of the class table the device uses
    int syn_format{2};        // 0..4, see DevImage
    uint32_t n_compact_slices{0};
    uint32_t n_bitmap_slices{0};  // of those: slices on bitmap axon records
    DevImage im{};
    DevState st{};
    std::vector<void *> allocs;
    size_t deliver_lds{0};
    bool ord_dict{false};           // format 8: the entries carry 5-bit codes into weight_lut instead of fp64 weights
    int ord_debug{0};               // SANAFE_ORDERED_ROLE=1/2: launch one role of ordered_deliver_kernel only (profiling)
    size_t ord_lds{0};              // format 8: dynamic LDS of ordered_deliver_kernel (the spike bitmap), 0: probe global memory
    const void *deliver_fn{nullptr}; // the deliver_kernel instantiation this chip launches (deliver_variants)
    const void *event_fn{nullptr};   // event_deliver_kernel instantiation (chips with the event layout, DevImage::ev_*)
    const void *event_fn_sparse{nullptr}; // ... reading the neuron-major copy of the block table (steps in which few neurons fire)
    int ev_lpb{4};                   // its lanes per block
    int ev_upl{1};                   // its 16-byte units per lane and batch
    int ev_waves{16};                // its wavefronts per workgroup
    // push / event decisions (DevImage::push_*): made here, on the host, from the events the device publishes
    long long *h_events{nullptr};    // pinned ring DevState::host_events points at
    long long epoch_first_step{1};   // Timestep::timestep of the first step whose events this chip's ring can hold
    int cur_pushed{0};               // mode of the step whose neuron launch went out last
    int cur_sparse{0};               // ... delivered by events: with the neuron-major table (few neurons fire)
    uint64_t ev_sparse_max_events{0}; // synaptic events per step up to which the neuron-major table is the faster one
    long long sparse_steps{0};
    long long pushed_steps{0};       // steps delivered by the push path / the event kernel since create
    long long ev_pending{-1};        // t_host after the step whose input still lies in DevState::ev_part (-1: none)
    uint32_t *cur_slog{nullptr};     // spike-record row of the step whose neuron launch went out last (msgsoma_kernel patches it)
    uint16_t *cur_msg_log{nullptr};  // its row of DevState::msg_fired_log (recorded status: detailed timing), or NULL
    long long dbg_waits{0}, dbg_fallbacks{0}; // decide_pushed: decisions that had to wait for the device / gave up (SANAFE_DEBUG_DECIDE)
    double dbg_wait_ms{0.0};
    double ev_avg_block{0.0};        // words per (source neuron, core group) block
    uint32_t ev_grid{0};
    uint32_t deliver_block{0};
    uint32_t neuron_grid{0};
    long long rec_host{0};
    bool timing{false};
    double t_neuron{0}, t_deliver{0}, t_reduce{0};
    std::vector<hipEvent_t> split_events;
    long long t_launches{0};
    std::vector<double> v0;
    // staging for host-evaluated units
    uint32_t *d_host_slots{nullptr}, *d_host_core{nullptr};
    uint8_t *d_host_status{nullptr};
    double *d_host_a{nullptr}, *d_host_b{nullptr};
    uint32_t host_cap{0};
    sanafe_hip_host_core_costs *d_host_costs{nullptr};
    uint32_t host_cost_cap{0};
    uint32_t *d_in_beg{nullptr}, *d_in_len{nullptr}, *d_in_bits{nullptr}; // input tables rewritten after create
    long long *d_in_period{nullptr};
    uint64_t in_bits_cap{0};
    sanafe_hip_soma_class *d_soma_classes{nullptr}; // table rewritten after create (sanafe_hip_write_soma_classes)
    uint32_t soma_class_cap{0};
    // external stream rows queued by sanafe_hip_write_ext
    int *d_ext{nullptr};
    long long ext_cap{0}, ext_rows{0}, ext_next{0};
};

namespace
{
// Host-side packing loops over up to ~10^9 synapses: contiguous blocks on a few threads.
template <typename F> void parallel_for(uint64_t n, const F &fn)
{
    unsigned T = std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    if (n < (1u << 16)) T = 1;
    if (T == 1)
    {
        fn(uint64_t{0}, n);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; t++) pool.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T); });
    for (std::thread &th : pool) th.join();
}

template <typename T> int upload(sanafe_hip_chip *c, const T *src, size_t n, const T **dst)
{
    *dst = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    if (n > 0)
    {
        if (src == nullptr) return fail(SANAFE_HIP_ERR_INVALID, "image array is NULL but its count is %zu", n);
        HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    }
    *dst = static_cast<const T *>(p);
    return 0;
}
template <typename T> int dalloc(sanafe_hip_chip *c, size_t n, T **dst, bool zero = true)
{
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    if (zero) HIPCHK(hipMemset(p, 0, bytes));
    *dst = static_cast<T *>(p);
    return 0;
}
#define TRY(expr)            \
    do                       \
    {                        \
        int rc_ = (expr);    \
        if (rc_ != 0) return rc_; \
    } while (0)

// Every deliver_kernel instantiation the library can launch, in ONE table: sanafe_hip_chip_create picks the chip's
// kernel from it, opts in to its dynamic LDS and checks its static LDS; launch_deliver launches what was picked.
struct DeliverVariant
{
    int fmt;
    bool delay, last, iacc;
    int block;
    bool bitmap, push, sub;
    const void *fn;
};
#define SANAFE_DV(F, D, L, I, B) {F, D, L, I, B, false, false, false, reinterpret_cast<const void *>(deliver_kernel<F, D, L, I, B>)}
#define SANAFE_DVX(F, D, I, B, BM, P) {F, D, false, I, B, BM, P, false, reinterpret_cast<const void *>(deliver_kernel<F, D, false, I, B, BM, P>)}
#define SANAFE_DVS(P) {7, false, false, false, DELIVER_BLOCK, true, P, true, reinterpret_cast<const void *>(deliver_kernel<7, false, false, false, DELIVER_BLOCK, true, P, true>)}
#define SANAFE_DV_FORMAT(F) SANAFE_DV(F, false, false, false, DELIVER_BLOCK), SANAFE_DV(F, true, false, false, DELIVER_BLOCK), \
                            SANAFE_DV(F, false, true, false, DELIVER_BLOCK), SANAFE_DV(F, false, false, false, 64)
#define SANAFE_DV_IACC(F) SANAFE_DV(F, false, false, true, DELIVER_BLOCK), SANAFE_DV(F, true, false, true, DELIVER_BLOCK), \
                          SANAFE_DV(F, false, true, true, DELIVER_BLOCK)
const DeliverVariant deliver_variants[] = {
        SANAFE_DV_FORMAT(0), SANAFE_DV_FORMAT(1), SANAFE_DV_FORMAT(2), SANAFE_DV_FORMAT(3), SANAFE_DV_FORMAT(4), SANAFE_DV_FORMAT(6),
        SANAFE_DV_FORMAT(7), SANAFE_DV_IACC(0), SANAFE_DV_IACC(3),
        // bitmap axon records (format 7)
        SANAFE_DVX(7, false, false, DELIVER_BLOCK, true, false), SANAFE_DVX(7, true, false, DELIVER_BLOCK, true, false),
        SANAFE_DVX(7, false, false, 64, true, false),
        // ... on sub-accumulators (cores of at most 256 neurons, no synaptic delays)
        SANAFE_DVS(false)};
// (no instantiation with the PUSH prologue any more: the host decides per step and does not launch the kernel on pushed steps)
#undef SANAFE_DV
#undef SANAFE_DVX
#undef SANAFE_DVS
#undef SANAFE_DV_FORMAT
#undef SANAFE_DV_IACC
// ordered delivery (format 8): dictionary entries x bitmap in LDS x per-neuron write-back rules
struct OrderedVariant
{
    bool dict, lds_bits, delay;
    const void *fn;
};
#define SANAFE_OV(D, L, Y) {D, L, Y, reinterpret_cast<const void *>(ordered_deliver_kernel<D, L, Y>)}
const OrderedVariant ordered_variants[] = {SANAFE_OV(false, false, false), SANAFE_OV(false, false, true), SANAFE_OV(false, true, false),
        SANAFE_OV(false, true, true), SANAFE_OV(true, false, false), SANAFE_OV(true, false, true), SANAFE_OV(true, true, false),
        SANAFE_OV(true, true, true)};
#undef SANAFE_OV
const DeliverVariant *find_deliver_variant(int fmt, bool delay, bool last, bool iacc, int block, bool bitmap, bool push, bool sub)
{
    for (const DeliverVariant &v : deliver_variants)
        if (v.fmt == fmt && v.delay == delay && v.last == last && v.iacc == iacc && v.block == block && v.bitmap == bitmap && v.push == push &&
                v.sub == sub)
            return &v;
    return nullptr;
}

// Ordered layout (format 8): the image's synapses regrouped per accumulator = (post neuron, delay value), every list in the
// image's synapse order, which IS the reference's delivery order at the destination core (axons by source core and source
// neuron, synapses of an axon in connection order; include/sanafe_hip.h "Index spaces").  Accumulators are sorted by list
// length, longest first, and cut into groups of 64 whose lists lie side by side ([row][lane]), padded to the group's
// longest list (rounded up to ORD_UNROLL rows) with entries that never fire.
int build_ordered(sanafe_hip_chip *c, const sanafe_hip_image &h)
{
    DevImage &im = c->im;
    // accumulator index space: core by core, (max delay + 1) rows of npad entries
    uint32_t rows = 1;
    {
        std::atomic<uint32_t> md{0};
        parallel_for(h.n_synapses, [&](uint64_t lo, uint64_t hi) {
            uint32_t m = 0;
            for (uint64_t k = lo; k < hi; k++) m = std::max(m, (h.syn_meta[k] >> 16) & 7u);
            uint32_t seen = md.load();
            while (seen < m && !md.compare_exchange_weak(seen, m)) {}
        });
        rows = md.load() + 1;
    }
    std::vector<uint64_t> acc_base(h.n_cores + 1, 0);
    for (uint32_t k = 0; k < h.n_cores; k++) acc_base[k + 1] = acc_base[k] + (uint64_t) rows * ((h.core_ncount[k] + 63u) & ~63u);
    const uint64_t n_acc = acc_base[h.n_cores];
    if (n_acc >= (1ull << 32)) return fail(SANAFE_HIP_ERR_UNSUPPORTED, "more than 2^32 accumulators");
    auto core_syn_end = [&](uint32_t k) { return (k + 1 < h.n_cores) ? h.core_syn_base[k + 1] : h.n_synapses; };
    auto acc_of = [&](uint32_t core, uint32_t meta) {
        return acc_base[core] + (uint64_t) ((meta >> 16) & 7u) * ((h.core_ncount[core] + 63u) & ~63u) + (meta & 0xffffu);
    };
    // the synapses of a core are one contiguous range of the image, in delivery order
    std::vector<uint32_t> count(n_acc, 0);
    parallel_for(h.n_cores, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; k++)
            for (uint64_t q = h.core_syn_base[k]; q < core_syn_end((uint32_t) k); q++)
                if (!((h.syn_meta[q] >> 19) & 1u)) count[acc_of((uint32_t) k, h.syn_meta[q])]++; // (lost charge: no entry at all)
    });
    std::vector<uint32_t> order;
    for (uint64_t a = 0; a < n_acc; a++)
        if (count[a] > 0) order.push_back((uint32_t) a);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return count[a] > count[b]; });
    const uint32_t n_groups = (uint32_t) ((order.size() + WAVE - 1) / WAVE);
    std::vector<OrdGroup> groups(n_groups);
    std::vector<uint32_t> lane_slot((size_t) n_groups * WAVE, 0xffffffffu);
    std::vector<uint8_t> lane_delay((size_t) n_groups * WAVE, 0);
    std::vector<uint32_t> place(n_acc, 0xffffffffu); // accumulator -> group * 64 + lane
    uint64_t n_entries = 0;
    for (uint32_t g = 0; g < n_groups; g++)
    {
        const uint32_t longest = count[order[(size_t) g * WAVE]];
        groups[g].off = n_entries;
        const uint32_t quantum = c->ord_dict ? ORD_DICT_ROWS : (uint32_t) ORD_UNROLL;
        groups[g].rows = (longest + quantum - 1) / quantum * quantum;
        groups[g].pad = 0;
        n_entries += (uint64_t) groups[g].rows * WAVE;
        for (uint32_t l = 0; l < (uint32_t) WAVE && (size_t) g * WAVE + l < order.size(); l++)
            place[order[(size_t) g * WAVE + l]] = g * WAVE + l;
    }
    // lane -> (slot, delay)
    for (uint32_t k = 0; k < h.n_cores; k++)
    {
        const uint32_t npad = (h.core_ncount[k] + 63u) & ~63u;
        for (uint64_t a = acc_base[k]; a < acc_base[k + 1]; a++)
        {
            if (place[a] == 0xffffffffu) continue;
            const uint64_t rel = a - acc_base[k];
            lane_slot[place[a]] = h.core_nbase[k] + (uint32_t) (rel % npad);
            lane_delay[place[a]] = (uint8_t) (rel / npad);
        }
    }
    const uint32_t pad_pre = h.n_global_slots; // its bit is always 0 (one spare zero word behind the bitmap)
    std::vector<uint32_t> pre(n_entries + (size_t) ORD_DICT_ROWS * WAVE, pad_pre);
    std::vector<double> wts;
    if (!c->ord_dict) wts.assign(n_entries + (size_t) ORD_UNROLL * WAVE, 0.0);
    auto lut_code = [&](double w) {
        for (uint32_t q = 0; q < 32; q++)
            if (std::memcmp(&c->weight_lut[q], &w, sizeof w) == 0) return q;
        return 0u;
    };
    std::fill(count.begin(), count.end(), 0u); // now: entries written so far
    std::vector<uint64_t> core_ax_beg(h.n_cores, 0), core_ax_end(h.n_cores, 0); // axons of a core: those of its slices
    for (uint32_t sl = h.n_slices; sl-- > 0;) core_ax_beg[h.slice_core[sl]] = h.slice_axon_beg[sl];
    for (uint32_t sl = 0; sl < h.n_slices; sl++) core_ax_end[h.slice_core[sl]] = h.slice_axon_end[sl];
    parallel_for(h.n_cores, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; k++)
        {
            // axon by axon: the pre slot belongs to the axon, its synapses follow in order
            for (uint64_t a = core_ax_beg[k]; a < core_ax_end[k]; a++)
            {
                const uint64_t src = h.core_syn_base[k] + h.ax_syn_beg[a];
                for (uint32_t q = 0; q < h.ax_nsyn[a]; q++)
                {
                    const uint32_t m = h.syn_meta[src + q];
                    if ((m >> 19) & 1u) continue;
                    const uint64_t acc = acc_of((uint32_t) k, m);
                    const uint32_t pl = place[acc], g = pl / WAVE, l = pl % WAVE;
                    const uint32_t row = count[acc]++;
                    const uint64_t at = c->ord_dict ? groups[g].off + (uint64_t) (row / 4u) * (4u * WAVE) + 4u * l + (row & 3u)
                                                    : groups[g].off + (uint64_t) row * WAVE + l;
                    pre[at] = c->ord_dict ? (h.ax_pre[a] | (lut_code(h.syn_weight[src + q]) << ORD_PRE_BITS)) : h.ax_pre[a];
                    if (!c->ord_dict) wts[at] = h.syn_weight[src + q];
                }
            }
        }
    });
    // (dictionary: n_global_slots < 2^27, so a padding entry reads as code 0 of a bit that never fires)
    im.ord_groups = n_groups;
    // one wavefront per group; at least enough wavefronts to spread the slices' processing-delay walk over the chip
    im.ord_wgs = std::max<uint32_t>((n_groups + 3) / 4, std::min<uint32_t>((h.n_slices + 3) / 4, 1024u));
    im.ord_walk_slices = h.n_slices;
    im.ord_dict = c->ord_dict ? 1 : 0;
    TRY(upload(c, groups.data(), groups.size(), &im.ord_group));
    TRY(upload(c, lane_slot.data(), lane_slot.size(), &im.ord_lane_slot));
    TRY(upload(c, lane_delay.data(), lane_delay.size(), &im.ord_lane_delay));
    TRY(upload(c, pre.data(), pre.size(), &im.ord_pre));
    im.ord_w = nullptr;
    if (!c->ord_dict) TRY(upload(c, wts.data(), wts.size(), &im.ord_w));
    im.weight_lut = nullptr;
    if (c->ord_dict) TRY(upload(c, c->weight_lut.data(), c->weight_lut.size(), &im.weight_lut));
    c->layout_bytes[0] = n_entries * (c->ord_dict ? 4ull : 12ull) + (uint64_t) n_groups * (sizeof(OrdGroup) + WAVE * 5ull);
    return 0;
}

// Push delivery tables (DevImage: push_*): per neuron, the synapses one spike of it reaches, with the post slot, the
// destination core and the weight -- built only where the push path is exact and cheap (see DevImage).
int build_push(sanafe_hip_chip *c, const sanafe_hip_image &h)
{
    DevImage &im = c->im;
    im.push_cap = 0;
    im.push_always = 0;
    if (std::getenv("SANAFE_PUSH") != nullptr && std::atoi(std::getenv("SANAFE_PUSH")) == 0) return 0; // tests / A-B runs
    // (the integer formats whose kernels have a PUSH instantiation: 7, and 0 / 3 on integer accumulators)
    const bool integer_weights = c->syn_format == 7 || ((c->syn_format == 0 || c->syn_format == 3) && c->acc_shift > 0);
    if (!integer_weights || c->has_delay || im.has_last || h.n_taps != 0 || h.n_ext != 0 ||
            h.n_synapses == 0 || h.n_synapses > (64ull << 20) || h.ax_lat_class == nullptr || h.ring_slots < 2)
        return 0; // (two rows of the time-step buffer: the neuron launch adds to the next step's row while it reads this step's)
    for (uint32_t g = 0; g < h.n_slots; g++)
        if ((h.slot_cls[g] & 7u) == SANAFE_SOMA_HOST) return 0; // their spikes are set after the neuron launch
    for (uint64_t k = 0; k < h.n_synapses; k++)
        if ((h.syn_meta[k] >> 19) & 1u) return 0; // lost charge still counts as an event: keep such chips on the pull path
    for (uint64_t a = 0; a < h.n_axons; a++)
        if (h.ax_nsyn[a] == 0u) return 0; // an axon without synapses is still a message (axon-in latency): it would have no entry here
    // one latency class per core
    std::vector<double> ev_lat(h.n_cores, 0.0);
    std::vector<int> cls(h.n_cores, -1);
    for (uint32_t sl = 0; sl < h.n_slices; sl++)
        for (uint64_t a = h.slice_axon_beg[sl]; a < h.slice_axon_end[sl]; a++)
        {
            const uint32_t core = h.slice_core[sl];
            if (h.ax_lat_class[a] == 255u) return 0;
            if (cls[core] < 0) cls[core] = h.ax_lat_class[a];
            else if (cls[core] != (int) h.ax_lat_class[a]) return 0;
        }
    for (uint32_t k = 0; k < h.n_cores; k++)
        if (cls[k] >= 0) ev_lat[k] = h.lat_class_per_event ? h.lat_class_per_event[cls[k]] : 0.0;
    // by GLOBAL source slot: on a tile-sharded chip the lists of the other ranks' neurons (their synapses into THIS chip) are
    // walked by remote_push_kernel after the all-gather of the spike bitmap
    std::vector<uint32_t> ptr((size_t) h.n_global_slots + 1, 0);
    for (uint64_t a = 0; a < h.n_axons; a++) ptr[h.ax_pre[a] + 1] += h.ax_nsyn[a];
    for (uint32_t g = 0; g < h.n_global_slots; g++) ptr[g + 1] += ptr[g];
    std::vector<PushEntry> syn(h.n_synapses);
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    for (uint32_t sl = 0; sl < h.n_slices; sl++)
    {
        const uint32_t core = h.slice_core[sl];
        for (uint64_t a = h.slice_axon_beg[sl]; a < h.slice_axon_end[sl]; a++)
        {
            const uint64_t src = h.core_syn_base[core] + h.ax_syn_beg[a];
            for (uint32_t k = 0; k < h.ax_nsyn[a]; k++)
            {
                PushEntry &e = syn[cur[h.ax_pre[a]]++];
                e.post = h.core_nbase[core] + (h.syn_meta[src + k] & 0xffffu);
                e.core = core | (k == 0 ? 0x80000000u : 0u);
                e.w = h.syn_weight[src + k];
            }
        }
    }
    TRY(upload(c, ptr.data(), ptr.size(), &im.push_ptr));
    TRY(upload(c, syn.data(), syn.size(), &im.push_syn));
    TRY(upload(c, h.core_axon_in_latency, h.n_cores, &im.core_ain_lat));
    TRY(upload(c, ev_lat.data(), ev_lat.size(), &im.core_event_lat));
    im.push_cap = h.n_slots / WAVE;
    // Push-only chips: a pushed event costs a few global atomics, a pull launch one probe per inbound axon whatever the
    // activity.  With at most SANAFE_PUSH_ONLY_DEGREE out-synapses per neuron on average the worst step (every neuron fires)
    // pushes in about the time of one pull launch, so the chip runs ONE launch per step and never decides anything.
    {
        double degree = 1.0;
        if (const char *env = std::getenv("SANAFE_PUSH_ONLY_DEGREE")) degree = std::atof(env);
        // (a rank of a tile-sharded chip holds the synapses INTO its neurons: about one per neuron, not exactly: 10 % slack)
        im.push_always = (double) h.n_synapses <= degree * (double) h.n_slots * (h.n_global_slots != h.n_slots ? 1.1 : 1.0) ? 1u : 0u;
        if (const char *env = std::getenv("SANAFE_PUSH_ONLY")) im.push_always = std::atoi(env) != 0 ? 1u : 0u;
    }
    // a pushed event costs three atomics inside the neuron launch; the pull path one probe per inbound axon of the chip, whatever the activity
    im.push_max_events = (uint32_t) std::min<uint64_t>(16384, std::max<uint64_t>(256, h.n_axons / 128));
    if (const char *env = std::getenv("SANAFE_PUSH_MAX_EVENTS")) im.push_max_events = (uint32_t) std::max(0L, std::atol(env));
    TRY(dalloc(c, 3 * (size_t) h.n_cores * 2, &c->st.push_core_cnt));
    return 0;
}

// Event-driven delivery layout (DevImage: ev_*): the chip's format-7 words a second time, regrouped source-neuron-major per
// group of destination cores, for steps with few spikes.  Built only where event_deliver_kernel is exact and pays:
// integer dictionary weights on bitmap axon records (one axon per source neuron and core, every axon owns its synapses, no
// lost charge), no synaptic delays / last-event cores / taps / host units, one latency class per core, and blocks long
// enough to be worth a table entry.  SANAFE_EVENT=0 switches it off, 1 builds it whatever the block length, 2 also delivers
// every step with it (tests); SANAFE_EVENT_SEGMENTS, SANAFE_EVENT_GROUP_CORES, SANAFE_EVENT_MAX_EVENTS tune it.
int build_event(sanafe_hip_chip *c, const sanafe_hip_image &h)
{
    DevImage &im = c->im;
    im.ev_groups = 0;
    im.ev_always = 0;
    int want = -1; // default: build when it pays
    if (const char *env = std::getenv("SANAFE_EVENT")) want = std::atoi(env);
    if (want == 0) return 0;
    if (c->syn_format != 7 || c->n_bitmap_slices == 0 || c->n_bitmap_slices != h.n_slices || c->has_delay || im.has_last || h.n_taps != 0 ||
            h.n_synapses == 0 || h.ax_lat_class == nullptr || h.n_slots / 64 == 0)
        return 0;
    if (c->uni && c->neuron_model == SANAFE_SOMA_TRUENORTH) return 0; // (its neuron kernel has no registers to spare for the partial rows)
    for (uint32_t g = 0; g < h.n_slots; g++)
        if ((h.slot_cls[g] & 7u) == SANAFE_SOMA_HOST) return 0; // their spikes are set after the neuron launch
    // the dictionary, densely coded
    std::vector<double> lut;
    for (double w : c->weight_lut)
    {
        bool known = false;
        for (double v : lut) known = known || v == w;
        if (!known) lut.push_back(w);
    }
    std::sort(lut.begin(), lut.end());
    const uint32_t code_bits = lut.size() <= 16 ? 4u : 5u;
    const uint32_t acc_max = 1u << (16u - code_bits);
    auto lut_code = [&](double w) {
        for (uint32_t q = 0; q < lut.size(); q++)
            if (lut[q] == w) return q;
        return 0u;
    };
    // groups of consecutive cores: at most 16, their slots within acc_max - EV_TRASH accumulators
    uint32_t max_group_cores = 16;
    if (const char *env = std::getenv("SANAFE_EVENT_GROUP_CORES")) max_group_cores = (uint32_t) std::min(16L, std::max(1L, std::atol(env)));
    std::vector<EvGroup> groups;
    std::vector<uint32_t> group_of(h.n_cores, 0xffffffffu);
    for (uint32_t k = 0; k < h.n_cores; k++)
    {
        const uint32_t npad = (h.core_ncount[k] + 63u) & ~63u;
        if (npad == 0) continue; // (no neurons: no inbound synapses either)
        if (npad > acc_max - EV_TRASH) return 0;
        const uint32_t end = h.core_nbase[k] + npad;
        if (groups.empty() || k - groups.back().core0 >= max_group_cores || end - groups.back().slot0 > acc_max - EV_TRASH)
            groups.push_back(EvGroup{k, 0u, h.core_nbase[k], 0u});
        groups.back().n_cores = k - groups.back().core0 + 1u;
        groups.back().n_acc = end - groups.back().slot0;
        group_of[k] = (uint32_t) groups.size() - 1u;
    }
    const uint32_t NG = (uint32_t) groups.size();
    if (NG == 0 || NG > 4096) return 0;
    // one latency class per core (the reduction prices a core's events with one constant)
    std::vector<double> ev_lat(h.n_cores, 0.0);
    {
        std::vector<int> cls(h.n_cores, -1);
        for (uint32_t sl = 0; sl < h.n_slices; sl++)
        {
            if (h.slice_axon_end[sl] == h.slice_axon_beg[sl]) continue;
            const uint32_t core = h.slice_core[sl];
            const int lc = h.ax_lat_class[h.slice_axon_beg[sl]]; // (bitmap slices: one class per slice)
            if (lc == 255) return 0;
            if (cls[core] < 0) cls[core] = lc;
            else if (cls[core] != lc) return 0;
        }
        for (uint32_t k = 0; k < h.n_cores; k++)
            if (cls[k] >= 0) ev_lat[k] = h.lat_class_per_event ? h.lat_class_per_event[cls[k]] : 0.0;
    }
    // slices of each group (slices are sorted by core)
    std::vector<uint32_t> group_slice_beg(NG + 1, h.n_slices);
    for (uint32_t sl = h.n_slices; sl-- > 0;) group_slice_beg[group_of[h.slice_core[sl]]] = sl;
    for (uint32_t g = NG; g-- > 0;)
        if (group_slice_beg[g] > group_slice_beg[g + 1]) group_slice_beg[g] = group_slice_beg[g + 1]; // (a group without slices)
    const uint64_t N = h.n_global_slots;
    // pass 1: words and core masks per (source neuron, group)
    std::vector<uint32_t> cnt(N * NG, 0u);
    std::vector<uint16_t> mask(N * NG, 0);
    std::atomic<bool> bad{false};
    parallel_for(NG, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t g = lo; g < hi; g++)
            for (uint32_t sl = group_slice_beg[g]; sl < group_slice_beg[g + 1]; sl++)
            {
                const uint16_t bit = (uint16_t) (1u << (h.slice_core[sl] - groups[g].core0));
                for (uint64_t a = h.slice_axon_beg[sl]; a < h.slice_axon_end[sl]; a++)
                {
                    const uint64_t at = (uint64_t) h.ax_pre[a] * NG + g;
                    if (mask[at] & bit) bad = true; // two axons from one neuron into one core: messages would be miscounted
                    mask[at] |= bit;
                    cnt[at] += h.ax_nsyn[a];
                }
            }
    });
    if (bad.load()) return 0;
    // is it worth it?  A block costs a table entry and starts a new 128-byte line: short blocks make the gather dearer than
    // the stream at any activity.
    {
        uint64_t blocks = 0;
        for (uint64_t i = 0; i < N * NG; i++) blocks += cnt[i] != 0u;
        if (blocks == 0 || (want < 1 && (double) h.n_synapses / (double) blocks < 6.0)) return 0;
        c->ev_avg_block = (double) h.n_synapses / (double) blocks;
    }
    // segments of the source space: 1,024-slot tiles, at most 64 per segment (16-bit list entries)
    const uint32_t n_tiles = (uint32_t) ((N + EV_TILE - 1) / EV_TILE);
    // grid = groups x segments workgroups, two resident per CU (65 KB of LDS each) and each running for the whole launch on a
    // busy step: the segment count in 4 .. 8 that fills the 256 CUs' slots most evenly (69 groups: 7 -> 483 of 512 slots,
    // where 8 -> 552 would leave a third of the CUs with three workgroups' work and the others with two)
    uint32_t segments = 8;
    {
        double best = 0.0;
        for (uint32_t sg = 4; sg <= EV_MAX_SEGMENTS; sg++)
        {
            const double wgs = (double) NG * sg, fill = wgs / (std::ceil(wgs / 512.0) * 512.0);
            if (fill >= best) best = fill, segments = sg;
        }
    }
    if (const char *env = std::getenv("SANAFE_EVENT_SEGMENTS")) segments = (uint32_t) std::max(1L, std::atol(env));
    segments = std::min(std::min(segments, n_tiles), EV_MAX_SEGMENTS); // (one row of partials per segment, DevState::ev_part)
    // pass 2: offsets, NEURON-major: the blocks of a neuron lie group after group, neuron after neuron (the ~70 workgroups that
    // need a fired neuron's blocks read one contiguous 5 KB region at about the same time: each 128-byte line comes out of
    // HBM once and is served to the other XCDs by the Infinity Cache; with the blocks of a GROUP together -- tried -- every
    // line is private to one workgroup and the launch is HBM-bound on half-used lines: 97 instead of 71 us at 10 % activity).
    // The TABLE -- meta = first 16-byte unit of block (n, g) | units << 32 | core mask << 48, one self-contained 8-byte entry
    // per lookup -- exists twice.  Group-major [g][n]: a workgroup, which walks the fired neurons of its segment in ascending
    // order, reads its group's entries front to back; at the headline's activity a 128-byte line serves ~5 lookups and a
    // wavefront's 16 lookups touch 3-4 lines.  Neuron-major [n][g]: the ~9 workgroups of an XCD that take neighbouring groups
    // of the same segment share one or two lines per fired neuron, where the group-major table costs each of them a line of
    // its own for one 8-byte entry -- the better table while few neurons fire (measured on C3: +11 % timesteps/s at 2 %
    // activity, +7 % at 10 %, -9 % at 34 %: every lookup of a wavefront is a line of its own).  The host picks per step
    // (decide_pushed: ev_sparse_max_events).
    const uint64_t msn = 1u, msg = N; // strides of the (group-major) table by neuron and group
    std::vector<uint64_t> meta((uint64_t) NG * N + 1u, 0ull);
    std::vector<uint64_t> neuron_units(N + 1, 0ull);
    parallel_for(N, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++)
        {
            uint64_t units = 0;
            for (uint32_t g = 0; g < NG; g++) units += (cnt[n * NG + g] + 7u) / 8u;
            neuron_units[n + 1] = units;
        }
    });
    for (uint64_t n = 0; n < N; n++) neuron_units[n + 1] += neuron_units[n];
    const uint64_t total_units = neuron_units[N];
    if (total_units >= (1ull << 32)) return 0;
    parallel_for(N, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++)
        {
            uint64_t off = neuron_units[n];
            for (uint32_t g = 0; g < NG; g++)
            {
                const uint64_t units = (cnt[n * NG + g] + 7u) / 8u;
                if (units > 0xffffu) bad = true;
                meta[g * msg + n * msn] = off | (units << 32) | ((uint64_t) mask[n * NG + g] << 48);
                off += units;
            }
        }
    });
    if (bad.load()) return 0;
    // pass 3: the words, group by group (a block is written by one thread); cnt becomes the write cursor.  Per (segment,
    // accumulator): events and |weight| sums, for the bounds of the integer accumulators.
    std::vector<uint16_t> words((total_units + 64u) * 8u, 0);
    std::atomic<uint64_t> max_count{0}, max_abs{0};
    for (uint64_t i = 0; i < N * NG; i++) cnt[i] = 0u;
    const uint32_t seg_tiles = (n_tiles + segments - 1u) / segments;
    parallel_for(NG, [&](uint64_t lo, uint64_t hi) {
        std::vector<uint32_t> count((size_t) segments * acc_max);
        std::vector<uint64_t> abs_sum((size_t) segments * acc_max);
        uint64_t mc = 0, ma = 0;
        for (uint64_t g = lo; g < hi; g++)
        {
            std::fill(count.begin(), count.end(), 0u);
            std::fill(abs_sum.begin(), abs_sum.end(), 0ull);
            for (uint32_t sl = group_slice_beg[g]; sl < group_slice_beg[g + 1]; sl++)
            {
                const uint32_t core = h.slice_core[sl];
                const uint32_t acc0 = h.core_nbase[core] - groups[g].slot0;
                for (uint64_t a = h.slice_axon_beg[sl]; a < h.slice_axon_end[sl]; a++)
                {
                    const uint64_t n = h.ax_pre[a];
                    const uint64_t src = h.core_syn_base[core] + h.ax_syn_beg[a];
                    const size_t sg = (size_t) std::min<uint32_t>((uint32_t) (n / EV_TILE) / seg_tiles, segments - 1u) * acc_max;
                    uint64_t at = (meta[g * msg + n * msn] & 0xffffffffull) * 8u + cnt[n * NG + g];
                    for (uint32_t k = 0; k < h.ax_nsyn[a]; k++)
                    {
                        const uint32_t idx = acc0 + (h.syn_meta[src + k] & 0xffffu);
                        words[at++] = (uint16_t) (lut_code(h.syn_weight[src + k]) | (idx << code_bits));
                        count[sg + idx]++;
                        abs_sum[sg + idx] += (uint64_t) std::fabs(h.syn_weight[src + k]);
                    }
                    cnt[n * NG + g] += h.ax_nsyn[a];
                }
            }
            for (size_t q = 0; q < count.size(); q++) mc = std::max<uint64_t>(mc, count[q]), ma = std::max(ma, abs_sum[q]);
            // padding words of every block: code 0 into the trash entries behind the group's accumulators
            for (uint64_t n = 0; n < N; n++)
            {
                const uint64_t b0 = (meta[g * msg + n * msn] & 0xffffffffull) * 8u, b1 = b0 + ((meta[g * msg + n * msn] >> 32) & 0xffffull) * 8u;
                for (uint64_t p = b0 + cnt[n * NG + g]; p < b1; p++) words[p] = (uint16_t) ((groups[g].n_acc + (uint32_t) (p & (EV_TRASH - 1u))) << code_bits);
            }
        }
        uint64_t seen = max_count.load();
        while (seen < mc && !max_count.compare_exchange_weak(seen, mc)) {}
        seen = max_abs.load();
        while (seen < ma && !max_abs.compare_exchange_weak(seen, ma)) {}
    });
    int shift = 1;
    while ((1ull << (shift - 1)) <= max_abs.load() && shift < 32) shift++;
    if (shift > 15 || ((max_count.load() + 1ull) << shift) > (1ull << 32)) return 0; // (weight + 2^shift: 16 bits in the kernel's table)
    std::vector<uint32_t> chunk_core(h.n_slots / 64, 0u);
    for (uint32_t k = 0; k < h.n_cores; k++)
        for (uint32_t q = 0; q < (h.core_ncount[k] + 63u) / 64u; q++) chunk_core[h.core_nbase[k] / 64u + q] = k;
    lut.resize(32, 0.0);
    TRY(upload(c, groups.data(), groups.size(), &im.ev_group));
    TRY(upload(c, reinterpret_cast<const unsigned long long *>(meta.data()), meta.size(), &im.ev_meta));
    // (between 10 % and 34 % activity on C3, see above)
    c->ev_sparse_max_events = (uint64_t) ((double) h.n_synapses * 0.2);
    if (const char *env = std::getenv("SANAFE_EVENT_SPARSE_EVENTS")) c->ev_sparse_max_events = (uint64_t) std::max(0LL, std::atoll(env));
    TRY(upload(c, words.data(), words.size(), &im.ev_words));
    TRY(upload(c, chunk_core.data(), chunk_core.size(), &im.ev_chunk_core));
    TRY(upload(c, lut.data(), lut.size(), &im.ev_lut));
    TRY(upload(c, h.core_axon_in_latency, h.n_cores, &im.core_ain_lat));
    TRY(upload(c, ev_lat.data(), ev_lat.size(), &im.core_event_lat));
    im.ev_groups = NG;
    im.ev_segments = segments;
    im.ev_seg_tiles = seg_tiles;
    im.ev_tiles = n_tiles;
    im.ev_shift = shift;
    im.ev_code_bits = code_bits;
    im.ev_always = want >= 2 ? 1u : 0u;
    // the per-step decision rides on the push machinery (reduce_l2 decides, reduce_l1 prices the per-core counters)
    im.push_cap = h.n_slots / WAVE;
    im.push_always = im.ev_always;
    // Measured on C3 1,024 x 256 (profiles/r04_c3_activity.json): the event kernel takes ~11 us + 0.6 us per million events,
    // the streaming kernel 245 us whatever the activity (and up to 0.8 ms when so few axons spike that its windows fall back
    // to the gather path): they cross at ~59 % of the neurons firing, 0.59 events per synapse and step
    im.push_max_events = (uint32_t) std::min<uint64_t>(0xffffffffu, (uint64_t) ((double) h.n_synapses * 0.58));
    if (const char *env = std::getenv("SANAFE_EVENT_MAX_EVENTS")) im.push_max_events = (uint32_t) std::max(0LL, std::atoll(env));
    TRY(dalloc(c, 3 * (size_t) h.n_cores * 2, &c->st.push_core_cnt));
    TRY(dalloc(c, (size_t) EV_MAX_SEGMENTS * h.n_slots, &c->st.ev_part)); // (rows of unused segments stay zero)
    c->ev_waves = 16;
    if (const char *env = std::getenv("SANAFE_EVENT_WAVES")) c->ev_waves = std::atoi(env) == 8 ? 8 : std::atoi(env) == 4 ? 4 : 16;
    c->layout_bytes[9] = total_units * 16ull;
    c->layout_bytes[10] = meta.size() * 8ull; // (one of the two tables: a step reads one)
    {
        // (uploaded last: what the busy steps read -- group-major table, blocks -- keeps the addresses it had without this copy)
        std::vector<uint64_t> meta_n(meta.size(), 0ull);
        parallel_for(N, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t n = lo; n < hi; n++)
                for (uint32_t g = 0; g < NG; g++) meta_n[n * NG + g] = meta[(uint64_t) g * N + n];
        });
        TRY(upload(c, reinterpret_cast<const unsigned long long *>(meta_n.data()), meta_n.size(), &im.ev_meta_n));
    }
    c->ev_grid = 8u * ((NG + 7u) / 8u) * segments;
    // 8 unit slots per block and batch from ~32 words per block on (4 lanes x 2 units: 16 neurons per batch), else 4
    c->ev_lpb = 4;
    c->ev_upl = c->ev_avg_block > 32.0 ? 2 : 1;
    // (2 lanes x 4 units -- 32 neurons per batch -- needs 77 registers: one workgroup per CU instead of two, 175 instead of
    //  142 us at the headline's activity)
    if (const char *env = std::getenv("SANAFE_EVENT_LPB")) c->ev_lpb = std::atoi(env) == 8 ? 8 : 4;
    if (const char *env = std::getenv("SANAFE_EVENT_UPL")) c->ev_upl = std::atoi(env) == 2 ? 2 : 1;
    if (c->ev_lpb == 8) c->ev_upl = 1;
    return 0;
}

int validate(const sanafe_hip_image *im)
{
    if (im->n_cores == 0) return fail(SANAFE_HIP_ERR_INVALID, "image has no cores");
    if (im->n_slots % 64 != 0) return fail(SANAFE_HIP_ERR_INVALID, "n_slots must be a multiple of 64");
    if (im->ring_slots < 1 || im->ring_slots > 8) return fail(SANAFE_HIP_ERR_INVALID, "ring_slots out of range");
    if (im->n_cost_classes == 0 || im->n_cost_classes > 1024)
        return fail(SANAFE_HIP_ERR_INVALID, "cost classes must be 1..1024");
    if (im->n_soma_classes == 0 || im->n_soma_classes > 65536)
        return fail(SANAFE_HIP_ERR_INVALID, "soma classes must be 1..65536");
    if (im->n_global_slots % 64 != 0 || im->slot_offset % 64 != 0 ||
            (uint64_t) im->slot_offset + im->n_slots > im->n_global_slots)
        return fail(SANAFE_HIP_ERR_INVALID, "bad global slot window");
    uint64_t next = 0;
    for (uint32_t c = 0; c < im->n_cores; c++)
    {
        if (im->core_nbase[c] % 64 != 0 || im->core_nbase[c] < next)
            return fail(SANAFE_HIP_ERR_INVALID, "core %u: slots must be 64-aligned and ascending", c);
        next = (uint64_t) im->core_nbase[c] + ((im->core_ncount[c] + 63u) & ~63u);
        if (next > im->n_slots) return fail(SANAFE_HIP_ERR_INVALID, "core %u overruns n_slots", c);
        if (im->core_ncount[c] > 65536) return fail(SANAFE_HIP_ERR_UNSUPPORTED, "core %u has more than 65536 neurons", c);
        // the buffer position is a property of the core: SANAFE_IN_LAST for all of its neurons or for none
        bool any_last = false, any_other = false;
        for (uint32_t k = 0; k < im->core_ncount[c]; k++)
        {
            const uint32_t cl = im->slot_cls[im->core_nbase[c] + k];
            if ((cl & 7u) == SANAFE_SOMA_NONE) continue;
            const uint32_t kind = (cl >> 3) & 7u;
            ((kind == SANAFE_IN_LAST || kind == SANAFE_IN_LAST_DELAY) ? any_last : any_other) = true;
        }
        if (any_last && any_other) return fail(SANAFE_HIP_ERR_INVALID, "core %u mixes SANAFE_IN_LAST with other input kinds", c);
    }
    for (uint32_t s = 0; s < im->n_slices; s++)
    {
        if (im->slice_core[s] >= im->n_cores) return fail(SANAFE_HIP_ERR_INVALID, "slice %u: bad core", s);
        if (s > 0 && im->slice_core[s] < im->slice_core[s - 1])
            return fail(SANAFE_HIP_ERR_INVALID, "slices must be sorted by core");
        if (im->slice_axon_beg[s] > im->slice_axon_end[s] || im->slice_axon_end[s] > im->n_axons)
            return fail(SANAFE_HIP_ERR_INVALID, "slice %u: bad axon range", s);
    }
    // Bounds of every index the kernels dereference: a bad image must fail here, not fault on the GPU.
    {
        std::mutex err_mutex;
        uint64_t err_slice = ~0ull;
        std::string err_text;
        auto report = [&](uint64_t s, const char *what, unsigned long long idx) {
            std::lock_guard<std::mutex> lock(err_mutex);
            if (s < err_slice)
            {
                err_slice = s;
                char buf[160];
                std::snprintf(buf, sizeof(buf), what, idx);
                err_text = buf;
            }
        };
        parallel_for(im->n_slices, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t s = lo; s < hi; s++)
            {
                const uint32_t core = im->slice_core[s];
                const uint64_t core_syn_end = (core + 1 < im->n_cores) ? im->core_syn_base[core + 1] : im->n_synapses;
                bool bad = false;
                for (uint64_t a = im->slice_axon_beg[s]; a < im->slice_axon_end[s] && !bad; a++)
                {
                    bad = true;
                    if (im->ax_pre[a] >= im->n_global_slots) report(s, "axon %llu: bad pre slot", a);
                    else if (im->ax_nsyn[a] > 0xffffu) report(s, "axon %llu has more than 65535 synapses", a);
                    else if (a > im->slice_axon_beg[s] && im->ax_syn_beg[a] != im->ax_syn_beg[a - 1] + im->ax_nsyn[a - 1])
                        report(s, "axon %llu: synapses of a slice must be contiguous in axon order", a);
                    else if (im->core_syn_base[core] + im->ax_syn_beg[a] + im->ax_nsyn[a] > core_syn_end)
                        report(s, "axon %llu: synapse range leaves its core", a);
                    else bad = false;
                    if (bad) break;
                    const uint64_t sb = im->core_syn_base[core] + im->ax_syn_beg[a];
                    for (uint32_t k = 0; k < im->ax_nsyn[a]; k++)
                        if ((im->syn_meta[sb + k] & 0xffffu) >= im->core_ncount[core])
                        {
                            report(s, "synapse %llu: post neuron outside its core", sb + k);
                            bad = true;
                            break;
                        }
                }
            }
        });
        if (err_slice != ~0ull) return fail(SANAFE_HIP_ERR_INVALID, "%s", err_text.c_str());
    }
    for (uint32_t g = 0; g < im->n_slots; g++)
    {
        const uint32_t cls = im->slot_cls[g], model = cls & 7u;
        if (model > SANAFE_SOMA_PERSIST) return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad soma model", g);
        if (model == SANAFE_SOMA_NONE) continue;
        if (((cls >> 6) & 1023u) >= im->n_cost_classes) return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad cost class", g);
        if ((model == SANAFE_SOMA_LIF || model == SANAFE_SOMA_TRUENORTH) && (cls >> 16) >= im->n_soma_classes)
            return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad soma class", g);
        if (model == SANAFE_SOMA_INPUT && im->slot_aux[g] >= im->n_input)
            return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad input index", g);
        if (((cls >> 3) & 7u) > SANAFE_IN_NONE) return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad input kind", g);
        if (model == SANAFE_SOMA_PERSIST && (cls >> 16) >= im->n_soma_classes) return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad soma class", g);
        if (((cls >> 3) & 7u) == SANAFE_IN_LAST_DELAY && im->slot_aux[g] + 2u > im->ring_slots)
            return fail(SANAFE_HIP_ERR_INVALID, "slot %u: delay %u needs %u ring slots", g, im->slot_aux[g], im->slot_aux[g] + 2u);
        if (((cls >> 3) & 7u) == SANAFE_IN_TAPS && (im->slot_aux[g] >= im->n_taps || im->tap_count[im->slot_aux[g]] < 1 || im->tap_count[im->slot_aux[g]] > 8))
            return fail(SANAFE_HIP_ERR_INVALID, "slot %u: bad tap table entry", g);
    }
    for (uint32_t i = 0; i < im->n_taps; i++)
        if (im->tap_slot[i] >= im->n_slots) return fail(SANAFE_HIP_ERR_INVALID, "tap table entry %u: bad slot", i);
    if (im->n_ext > 0)
    {
        if (!im->slot_ext) return fail(SANAFE_HIP_ERR_INVALID, "n_ext > 0 but slot_ext is NULL");
        for (uint32_t g = 0; g < im->n_slots; g++)
            if (im->slot_ext[g] != 0xffffffffu && im->slot_ext[g] >= im->n_ext)
                return fail(SANAFE_HIP_ERR_INVALID, "slot %u: external stream column out of range", g);
    }
    for (uint32_t i = 0; i < im->n_input; i++)
        if ((uint64_t) im->in_train_beg[i] + im->in_train_len[i] > im->n_train_words * 32ull)
            return fail(SANAFE_HIP_ERR_INVALID, "input %u: spike train outside in_train_bits", i);
    // cores whose soma is part of the message pipeline (msg_*)
    for (uint32_t k = 0; k < im->n_msg_cores; k++)
    {
        if (!im->msg_core || !im->msg_ax_beg || !im->msg_syn_beg || !im->msg_costs) return fail(SANAFE_HIP_ERR_INVALID, "msg_* arrays are NULL");
        const uint32_t core = im->msg_core[k];
        if (core >= im->n_cores || (k > 0 && core <= im->msg_core[k - 1])) return fail(SANAFE_HIP_ERR_INVALID, "msg core %u: bad or unsorted core", k);
        if (im->msg_ax_beg[k] > im->msg_ax_beg[k + 1] || im->msg_syn_beg[k] > im->msg_syn_beg[k + 1])
            return fail(SANAFE_HIP_ERR_INVALID, "msg core %u: bad table ranges", k);
        uint64_t syn = im->msg_syn_beg[k];
        for (uint32_t a = im->msg_ax_beg[k]; a < im->msg_ax_beg[k + 1]; a++)
        {
            if (im->msg_ax_pre[a] >= im->n_global_slots) return fail(SANAFE_HIP_ERR_INVALID, "msg core %u: bad pre slot", k);
            syn += im->msg_ax_nsyn[a];
        }
        if (syn != im->msg_syn_beg[k + 1]) return fail(SANAFE_HIP_ERR_INVALID, "msg core %u: the axons' synapse counts do not add up", k);
        for (uint32_t q = im->msg_syn_beg[k]; q < im->msg_syn_beg[k + 1]; q++)
            if (im->msg_syn_post[q] >= im->core_ncount[core]) return fail(SANAFE_HIP_ERR_INVALID, "msg core %u: post neuron outside its core", k);
        for (uint32_t n = 0; n < im->core_ncount[core]; n++)
        {
            const uint32_t cl = im->slot_cls[im->core_nbase[core] + n], m = cl & 7u;
            if (m == SANAFE_SOMA_NONE) continue;
            if (!((m == SANAFE_SOMA_TRUENORTH && ((cl >> 3) & 7u) == SANAFE_IN_NONE) || m == SANAFE_SOMA_PERSIST))
                return fail(SANAFE_HIP_ERR_INVALID, "msg core %u: its neurons must be TrueNorth somas with SANAFE_IN_NONE, or SANAFE_SOMA_PERSIST", k);
        }
    }
    return 0;
}
} // namespace

extern "C" const char *sanafe_hip_last_error(void) { return g_err.c_str(); }

extern "C" int sanafe_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sanafe_hip_chip_create(const sanafe_hip_image *image, int device, sanafe_hip_chip **out)
{
    if (!image || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    *out = nullptr;
    TRY(validate(image)); // a malformed image is reported as such, with or without a device
    if (device < 0 || sanafe_hip_device_count() <= device)
        return fail(SANAFE_HIP_ERR_NO_DEVICE, "no HIP device %d (libsanafe_hip has no CPU fallback)", device);
    auto *c = new sanafe_hip_chip();
    c->device = device;
    auto bail = [&](int rc) {
        sanafe_hip_chip_destroy(c);
        return rc;
    };
#define TRYC(expr)                    \
    do                                \
    {                                 \
        int rc2_ = (expr);            \
        if (rc2_ != 0) return bail(rc2_); \
    } while (0)
#define HIPC(expr)                                                                                       \
    do                                                                                                   \
    {                                                                                                    \
        hipError_t e2_ = (expr);                                                                         \
        if (e2_ != hipSuccess)                                                                           \
            return bail(fail(SANAFE_HIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e2_)));       \
    } while (0)
    HIPC(hipSetDevice(device));
    HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const sanafe_hip_image &h = *image;
    DevImage &im = c->im;
    im.n_cores = h.n_cores;
    im.n_slots = h.n_slots;
    im.ring_slots = h.ring_slots;
    im.n_slices = h.n_slices;
    im.n_input = h.n_input;
    im.slot_offset = h.slot_offset;
    im.n_global_slots = h.n_global_slots;
    im.sync_delay = h.sync_delay;
    uint32_t max_pad = 64;
    for (uint32_t k = 0; k < h.n_cores; k++) max_pad = std::max(max_pad, (h.core_ncount[k] + 63u) & ~63u);
    im.max_core_slots = max_pad;
    TRYC(upload(c, h.core_nbase, h.n_cores, &im.core_nbase));
    TRYC(upload(c, h.core_axon_out_latency, h.n_cores, &im.core_axon_out_latency));
    TRYC(upload(c, h.soma_classes, h.n_soma_classes, &im.soma_classes));
    TRYC(upload(c, h.cost_classes, h.n_cost_classes, &im.cost_classes));
    TRYC(upload(c, h.slot_cls, h.n_slots, &im.slot_cls));
    TRYC(upload(c, h.slot_bias, h.n_slots, &im.slot_bias));
    TRYC(upload(c, h.slot_aux, h.n_slots, &im.slot_aux));
    {
        // what one spike of each neuron causes downstream: one 40-byte record per slot
        std::vector<SpikeStatic> spike(h.n_slots);
        for (uint32_t g = 0; g < h.n_slots; g++)
        {
            spike[g].e_net = h.slot_e_net[g];
            spike[g].e_syn = h.slot_e_syn[g];
            spike[g].e_dend = h.slot_e_dend[g];
            spike[g].packets = h.slot_packets[g];
            spike[g].hops = h.slot_hops[g];
            spike[g].events = h.slot_events[g];
            spike[g].pad = 0;
        }
        TRYC(upload(c, spike.data(), spike.size(), &im.slot_spike));
    }
    {
        // neuron workgroups: up to 4 consecutive 64-slot chunks of one core each
        std::vector<WgDesc> wgs;
        std::vector<uint32_t> wg_beg(h.n_cores + 1, 0);
        for (uint32_t k = 0; k < h.n_cores; k++)
        {
            wg_beg[k] = (uint32_t) wgs.size();
            const uint32_t chunks = (h.core_ncount[k] + 63u) / 64u;
            for (uint32_t q = 0; q < chunks; q += NEURON_BLOCK / WAVE)
                wgs.push_back(WgDesc{h.core_nbase[k] + q * 64u, k, std::min<uint32_t>(NEURON_BLOCK / WAVE, chunks - q), 0u});
        }
        wg_beg[h.n_cores] = (uint32_t) wgs.size();
        // Which neuron kernel: one soma model on the whole chip -> that model's kernel; and when every live slot
        // carries the same class word on equal cores laid out back to back (the large synthetic configurations),
        // the parameters travel as kernel arguments and the workgroup -> slot mapping is arithmetic (UniformSoma).
        {
            uint32_t cls0 = 0, ncount0 = 0, next_slot = 0;
            bool same_cls = true, same_model = true, equal_cores = true;
            for (uint32_t k = 0; k < h.n_cores; k++)
            {
                if (h.core_ncount[k] == 0) continue;
                if (ncount0 == 0) ncount0 = h.core_ncount[k];
                equal_cores = equal_cores && h.core_ncount[k] == ncount0 && h.core_nbase[k] == next_slot;
                next_slot = h.core_nbase[k] + ((h.core_ncount[k] + 63u) & ~63u);
                for (uint32_t q = 0; q < h.core_ncount[k]; q++)
                {
                    const uint32_t cl = h.slot_cls[h.core_nbase[k] + q];
                    if (cls0 == 0) cls0 = cl;
                    same_cls = same_cls && cl == cls0;
                    same_model = same_model && (cl & 7u) == (cls0 & 7u);
                }
            }
            const uint32_t model0 = cls0 & 7u;
            c->neuron_model = (same_model && (model0 == SANAFE_SOMA_LIF || model0 == SANAFE_SOMA_TRUENORTH)) ? (int) model0 : 0;
            const uint32_t cpc = (ncount0 + 63u) / 64u, wpc = (cpc + 3u) / 4u;
            c->uni = c->neuron_model != 0 && same_cls && equal_cores && h.n_ext == 0 && (wpc & (wpc - 1u)) == 0u &&
                    ((cls0 >> 3) & 7u) != SANAFE_IN_LAST_DELAY && // its per-slot delay lives in slot_aux
                    (cls0 >> 16) < h.n_soma_classes && ((cls0 >> 6) & 1023u) < h.n_cost_classes &&
                    std::getenv("SANAFE_NEURON_GENERIC") == nullptr; // tests: force the table-driven kernel
            if (c->uni)
            {
                c->us.p = h.soma_classes[cls0 >> 16];
                c->us.c = h.cost_classes[(cls0 >> 6) & 1023u];
                c->us.cls = cls0;
                c->us.ncount = ncount0;
                c->us.cpc = cpc;
                c->us.wpc_shift = 0;
                while ((1u << c->us.wpc_shift) < wpc) c->us.wpc_shift++;
                // one bias for every live slot: a kernel argument instead of an 8-byte load per slot
                bool same_bias = true;
                double bias0 = 0.0;
                bool have_bias = false;
                for (uint32_t k = 0; k < h.n_cores && same_bias; k++)
                    for (uint32_t q = 0; q < h.core_ncount[k] && same_bias; q++)
                    {
                        const double bq = h.slot_bias[h.core_nbase[k] + q];
                        if (!have_bias) bias0 = bq, have_bias = true;
                        same_bias = std::memcmp(&bq, &bias0, sizeof bq) == 0;
                    }
                c->us.bias_uniform = same_bias ? 1u : 0u;
                c->us.bias = bias0;
            }
            im.uni_costing = c->uni ? 1 : 0;
            if (c->uni) im.uni_cost = c->us.c;
            else std::memset(&im.uni_cost, 0, sizeof im.uni_cost);
            c->h_soma_classes.assign(h.soma_classes, h.soma_classes + h.n_soma_classes);
        }
        im.spike_energy = 0;
        for (uint32_t g = 0; g < h.n_slots; g++)
        {
            im.spike_energy |= (h.slot_e_syn[g] != 0.0 ? 1 : 0) | (h.slot_e_net[g] != 0.0 ? 2 : 0) | (h.slot_e_dend[g] != 0.0 ? 4 : 0);
            if (h.slot_packets[g] >= (1u << 18)) return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "slot %u: a spike fans out to 2^18 cores or more", g));
        }
        im.n_wgs = (uint32_t) wgs.size();
        im.n_groups = (h.n_cores + L1_CORES - 1) / L1_CORES;
        im.n_reduce_wgs = (im.n_groups + (NEURON_BLOCK / WAVE) - 1) / (NEURON_BLOCK / WAVE);
        TRYC(upload(c, wgs.data(), wgs.size(), &im.wg_desc));
        TRYC(upload(c, wg_beg.data(), wg_beg.size(), &im.core_wg_beg));
        c->h_core_wg_beg = wg_beg;
        c->h_core_out_lat.assign(h.core_axon_out_latency, h.core_axon_out_latency + h.n_cores);
    }
    im.n_soma_classes = h.n_soma_classes;
    im.n_cost_classes = h.n_cost_classes;
    im.has_lif = 0;
    for (uint32_t g = 0; g < h.n_slots && !im.has_lif; g++) im.has_lif = (h.slot_cls[g] & 7u) == SANAFE_SOMA_LIF;
    im.any_refrac = 0;
    for (uint32_t k = 0; k < h.n_soma_classes; k++) im.any_refrac |= h.soma_classes[k].refractory_delay > 0;
    TRYC(upload(c, h.in_train_beg, h.n_input, &im.in_train_beg));
    TRYC(upload(c, h.in_train_len, h.n_input, &im.in_train_len));
    TRYC(upload(c, reinterpret_cast<const long long *>(h.in_rate_period), h.n_input, &im.in_rate_period));
    TRYC(upload(c, h.in_train_bits, h.n_train_words, &im.in_train_bits));
    im.n_taps = h.n_taps;
    im.tap_slot = im.tap_count = nullptr;
    im.tap_tc = im.tap_sc = nullptr;
    if (h.n_taps > 0)
    {
        TRYC(upload(c, h.tap_slot, h.n_taps, &im.tap_slot));
        TRYC(upload(c, h.tap_count, h.n_taps, &im.tap_count));
        TRYC(upload(c, h.tap_tc, (size_t) h.n_taps * 8, &im.tap_tc));
        TRYC(upload(c, h.tap_sc, (size_t) h.n_taps * 8, &im.tap_sc));
    }
    im.n_ext = h.n_ext;
    im.slot_ext = nullptr;
    if (h.n_ext > 0) TRYC(upload(c, h.slot_ext, h.n_slots, &im.slot_ext));
    // ---- cores whose soma is part of the message pipeline: per post-synaptic neuron, its inbound synapses in delivery order ----
    im.n_msg_cores = h.n_msg_cores;
    im.n_msg_chunks = 0;
    c->st.msg_cnt = nullptr;
    c->st.msg_ax_fired = nullptr;
    c->st.msg_fired_log = nullptr;
    im.n_msg_axons = 0;
    if (h.n_msg_cores > 0)
    {
        std::vector<MsgCoreDev> cores(h.n_msg_cores);
        std::vector<uint32_t> chunk_core, chunk_slot0, ptr((size_t) h.n_slots + 1, 0u);
        for (uint32_t k = 0; k < h.n_msg_cores; k++)
        {
            const uint32_t core = h.msg_core[k];
            cores[k].core = core;
            cores[k].first_chunk = (uint32_t) chunk_core.size();
            cores[k].ax_beg = h.msg_ax_beg[k];
            cores[k].ax_end = h.msg_ax_beg[k + 1];
            cores[k].costs = h.msg_costs[k];
            for (uint32_t q = 0; q < (h.core_ncount[core] + 63u) / 64u; q++)
            {
                chunk_core.push_back(k);
                chunk_slot0.push_back(h.core_nbase[core] + 64u * q);
            }
            for (uint32_t q = h.msg_syn_beg[k]; q < h.msg_syn_beg[k + 1]; q++) ptr[h.core_nbase[core] + h.msg_syn_post[q] + 1u]++;
        }
        for (uint32_t g = 0; g < h.n_slots; g++) ptr[g + 1] += ptr[g];
        const uint32_t n_syn = ptr[h.n_slots];
        std::vector<uint32_t> pre(std::max<uint32_t>(n_syn, 1u), 0u), axn(std::max<uint32_t>(n_syn, 1u), 0u), cur(ptr.begin(), ptr.end() - 1);
        std::vector<double> w(std::max<uint32_t>(n_syn, 1u), 0.0);
        for (uint32_t k = 0; k < h.n_msg_cores; k++)
        {
            const uint32_t core = h.msg_core[k];
            uint32_t q = h.msg_syn_beg[k];
            for (uint32_t a = h.msg_ax_beg[k]; a < h.msg_ax_beg[k + 1]; a++) // axon by axon = delivery order, per neuron as well
                for (uint32_t j = 0; j < h.msg_ax_nsyn[a]; j++, q++)
                {
                    const uint32_t at = cur[h.core_nbase[core] + h.msg_syn_post[q]]++;
                    pre[at] = h.msg_ax_pre[a];
                    axn[at] = a;
                    w[at] = h.msg_syn_weight[q];
                }
        }
        im.n_msg_chunks = (uint32_t) chunk_core.size();
        TRYC(upload(c, cores.data(), cores.size(), &im.msg_core_dev));
        TRYC(upload(c, chunk_core.data(), chunk_core.size(), &im.msg_chunk_core));
        TRYC(upload(c, chunk_slot0.data(), chunk_slot0.size(), &im.msg_chunk_slot0));
        TRYC(upload(c, ptr.data(), ptr.size(), &im.msg_ptr));
        TRYC(upload(c, pre.data(), pre.size(), &im.msg_pre));
        TRYC(upload(c, axn.data(), axn.size(), &im.msg_ax));
        im.n_msg_axons = h.msg_ax_beg[h.n_msg_cores];
        TRYC(dalloc(c, std::max<uint32_t>(im.n_msg_axons, 1u), &c->st.msg_ax_fired));
        TRYC(upload(c, w.data(), w.size(), &im.msg_w));
        TRYC(upload(c, h.msg_ax_pre, h.msg_ax_beg[h.n_msg_cores], &im.msg_ax_pre));
        TRYC(dalloc(c, (size_t) h.n_msg_cores * 4, &c->st.msg_cnt));
    }
    // ---- synapse format: the narrowest that holds every weight exactly and every accumulator index (see DevImage) ----
    //   streamable: 0 int8 weights, 3 12-bit integer weights, 4 fp64 weights; gather-only fall-backs: 1 (12-bit), 2 (fp64)
    {
        std::atomic<int> need{0};
        std::atomic<bool> fractional{false}; // some weight is not an integer (of moderate size): fp64 sums depend on their order
        parallel_for(h.n_synapses, [&](uint64_t lo, uint64_t hi) {
            int fmt = 0;
            bool frac = false;
            for (uint64_t k = lo; k < hi && !(fmt == 2 && frac); k++)
            {
                const double w = h.syn_weight[k];
                const bool integral = (w >= -2048.0 && w <= 2047.0 && w == (double) (int) w && !(w == 0.0 && std::signbit(w)));
                if (!integral) fmt = 2;
                else if (w < -128.0 || w > 127.0) fmt = std::max(fmt, 1);
                // sums of integers below 2^40 stay exact in fp64 whatever the order (2^13 events per accumulator and step)
                if (!(std::fabs(w) <= 1099511627776.0 && w == std::nearbyint(w))) frac = true;
            }
            int seen = need.load();
            while (seen < fmt && !need.compare_exchange_weak(seen, fmt)) {}
            if (frac) fractional = true;
        });
        int fmt = need.load();
        // format 0 addresses the LDS accumulators with 15 bits: (max delay + 1) rows of npad + 1 entries
        uint32_t fmt0_rows = 1, fmt0_max_pad = 64;
        {
            std::atomic<uint32_t> md{0};
            parallel_for(h.n_synapses, [&](uint64_t lo, uint64_t hi) {
                uint32_t m = 0;
                for (uint64_t k = lo; k < hi; k++) m = std::max(m, (h.syn_meta[k] >> 16) & 7u);
                uint32_t seen = md.load();
                while (seen < m && !md.compare_exchange_weak(seen, m)) {}
            });
            fmt0_rows = md.load() + 1;
            for (uint32_t k = 0; k < h.n_cores; k++) fmt0_max_pad = std::max(fmt0_max_pad, (h.core_ncount[k] + 63u) & ~63u);
        }
        // the index-coded words address (max delay + 1) rows of npad + 1 accumulators: 15 bits (12 with 12-bit weights)
        const uint64_t n_acc = (uint64_t) fmt0_rows * (fmt0_max_pad + 1u);
        const int weights = fmt; // 0 int8, 1 12-bit integers, 2 fp64
        if (weights == 0) fmt = n_acc <= 8192ull ? 0 : 1;
        else if (weights == 1) fmt = n_acc <= 4096ull ? 3 : 1;
        else fmt = n_acc <= 32768ull ? 4 : 2;
        // format 6 (2-byte words, dictionary-coded weights): at most 32 distinct weight values on the chip, none of them
        // -0.0 (the untouched-accumulator sentinel), at most 1024 accumulators per core
        // At most 32 distinct weight values on the chip: a dictionary (formats 6, 7; ordered delivery with 4-byte entries)
        bool lut_ok = h.n_synapses > 0;
        {
            std::mutex lut_mutex;
            std::vector<double> lut;
            std::atomic<bool> too_many{false};
            parallel_for(h.n_synapses, [&](uint64_t lo, uint64_t hi) {
                std::vector<double> local;
                for (uint64_t k = lo; k < hi && !too_many.load(std::memory_order_relaxed); k++)
                {
                    const double w = h.syn_weight[k];
                    bool known = false;
                    for (double v : local) known = known || (std::memcmp(&v, &w, sizeof w) == 0);
                    if (!known)
                    {
                        local.push_back(w);
                        if (local.size() > 32) too_many = true;
                    }
                }
                std::lock_guard<std::mutex> lock(lut_mutex);
                for (double w : local)
                {
                    bool known = false;
                    for (double v : lut) known = known || (std::memcmp(&v, &w, sizeof w) == 0);
                    if (!known) lut.push_back(w);
                }
                if (lut.size() > 32) too_many = true;
            });
            lut_ok = lut_ok && !too_many.load();
            if (lut_ok)
            {
                std::sort(lut.begin(), lut.end()); // thread-count independent codes
                if (lut.size() <= 16)
                {
                    // up to 16 values take the EVEN codes (the odd ones repeat them): the delivery kernel on sub-accumulators
                    // picks the sub-accumulator -- and with it the LDS bank -- by code >> 1, so the values spread over all 16
                    std::vector<double> spread(32, 0.0);
                    for (size_t i = 0; i < lut.size(); i++) spread[2 * i] = spread[2 * i + 1] = lut[i];
                    lut = spread;
                }
                lut.resize(32, 0.0);
                c->weight_lut = lut;
            }
        }
        // format 6 (2-byte words, dictionary-coded weights): none of the values -0.0 (the untouched-accumulator sentinel), at
        // most 1024 accumulators per core
        bool dict_ok = lut_ok && n_acc <= 1024ull;
        if (dict_ok) // the words carry no axon code: every axon must own at least one of them
        {
            std::atomic<bool> empty_axon{false};
            parallel_for(h.n_axons, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t a = lo; a < hi; a++)
                    if (h.ax_nsyn[a] == 0) empty_axon = true;
            });
            dict_ok = !empty_axon.load();
        }
        if (dict_ok)
            for (double w : c->weight_lut) dict_ok = dict_ok && !(w == 0.0 && std::signbit(w));
        // Integer accumulators (format 7; formats 0 and 3 when they can): integer weights, and per slice and accumulator the
        // bounds that keep "events * 2^shift + sum of weights" decodable (see DevImage)
        bool int_ok = dict_ok;
        if (int_ok)
            for (double w : c->weight_lut) int_ok = int_ok && std::fabs(w) <= 1048576.0 && w == (double) (long long) w;
        int acc_shift = 0; // 0: the bounds do not hold
        if ((int_ok || weights <= 1) && n_acc <= 32768ull && h.n_synapses > 0)
        {
            std::atomic<uint64_t> max_count{0}, max_abs{0};
            parallel_for(h.n_slices, [&](uint64_t lo, uint64_t hi) {
                std::vector<uint32_t> count(n_acc);
                std::vector<uint64_t> abs_sum(n_acc);
                uint64_t mc = 0, ma = 0;
                for (uint64_t sl = lo; sl < hi; sl++)
                {
                    const uint32_t core = h.slice_core[sl];
                    const uint32_t row = ((h.core_ncount[core] + 63u) & ~63u) + 1u;
                    std::fill(count.begin(), count.end(), 0u);
                    std::fill(abs_sum.begin(), abs_sum.end(), 0ull);
                    for (uint64_t a = h.slice_axon_beg[sl]; a < h.slice_axon_end[sl]; a++)
                    {
                        const uint64_t src = h.core_syn_base[core] + h.ax_syn_beg[a];
                        for (uint32_t k = 0; k < h.ax_nsyn[a]; k++)
                        {
                            const uint32_t m = h.syn_meta[src + k];
                            if ((m >> 19) & 1u) continue; // lost charge: the trash entry may wrap
                            const uint32_t idx = ((m >> 16) & 7u) * row + (m & 0xffffu);
                            count[idx]++;
                            abs_sum[idx] += (uint64_t) std::fabs(h.syn_weight[src + k]);
                        }
                    }
                    for (uint64_t q = 0; q < n_acc; q++) mc = std::max<uint64_t>(mc, count[q]), ma = std::max(ma, abs_sum[q]);
                }
                uint64_t seen = max_count.load();
                while (seen < mc && !max_count.compare_exchange_weak(seen, mc)) {}
                seen = max_abs.load();
                while (seen < ma && !max_abs.compare_exchange_weak(seen, ma)) {}
            });
            int shift = 1;
            while ((1ull << (shift - 1)) <= max_abs.load() && shift < 32) shift++;
            if (shift <= 30 && ((max_count.load() + 1ull) << shift) <= (1ull << 32)) acc_shift = shift;
        }
        if (const char *env = std::getenv("SANAFE_INT_ACC")) // tests: 0 keeps the fp64 accumulators
            if (std::atoi(env) == 0) acc_shift = 0;
        int_ok = int_ok && acc_shift > 0 && acc_shift <= 15; // format 7: weight + 2^shift has 16 bits in the kernel's table
        const int plain = fmt;
        // integer dictionary: 7.  Other dictionaries: 6 beats the 12-byte fp64 layouts; integers that fail the bounds
        // of 7 keep their 4-byte streamable form when they have one (fp64 LDS adds bound both, 4 bytes has less ALU work).
        if (int_ok) fmt = 7;
        else if (dict_ok && !(plain == 0 || plain == 3)) fmt = 6;
        if (h.n_synapses == 0) fmt = 2;
        if (const char *env = std::getenv("SANAFE_SYN_FORMAT")) // tests / experiments: 1 or 2 force the gather-only
        {                                                       // layouts, 0 / 3 / 4 the wider streamable ones
            const int want = std::atoi(env);
            if (want == 0) fmt = plain; // no dictionary coding
            else if (want == 6 && dict_ok) fmt = 6; // dictionary with fp64 accumulators
            else if (want == 1) fmt = weights <= 1 ? 1 : 2;
            else if (want == 2) fmt = 2;
            else if (want == 3 && weights <= 1 && n_acc <= 4096ull) fmt = 3;
            else if (want == 4 && n_acc <= 32768ull) fmt = 4;
        }
        // Non-integer weights: fp64 sums depend on the order of the additions, and only the reference's order gives the
        // reference's bits -- ordered delivery (format 8, ordered_deliver_kernel).  Integer weights are exact in any order
        // and keep the streaming formats.  Cores that keep only the LAST event of a step (buffer before the dendrite unit)
        // are order-free by themselves but share the chip-wide kernel choice: such chips stay on the streaming kernels.
        bool any_last = false;
        for (uint32_t g = 0; g < h.n_slots && !any_last; g++)
            any_last = (h.slot_cls[g] & 7u) != SANAFE_SOMA_NONE &&
                       (((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_LAST || ((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_LAST_DELAY);
        bool ordered = fractional.load() && !any_last && h.n_synapses > 0 && std::getenv("SANAFE_SYN_FORMAT") == nullptr;
        if (const char *env = std::getenv("SANAFE_SYN_FORMAT")) // tests: 8 forces the ordered layout on integer weights too
            if (std::atoi(env) == 8 && !any_last && h.n_synapses > 0) ordered = true;
        if (const char *env = std::getenv("SANAFE_ORDERED")) // 0: non-integer weights through the streaming kernels (A/B runs)
            if (std::atoi(env) == 0) ordered = false;
        if (ordered) fmt = 8;
        c->ord_dict = ordered && lut_ok && h.n_global_slots < (1u << ORD_PRE_BITS) && std::getenv("SANAFE_ORDERED_NO_DICT") == nullptr;
        c->syn_format = fmt;
        c->acc_shift = (fmt == 7 || ((fmt == 0 || fmt == 3) && weights <= 1)) ? acc_shift : 0;
    }
    const bool dict16 = (c->syn_format == 6 || c->syn_format == 7); // 2-byte dictionary-coded words
    const bool stream_layout = (c->syn_format == 0 || c->syn_format == 3 || c->syn_format == 4 || dict16);
    const uint64_t group_words = dict16 ? 8ull : 4ull; // words per 16-byte group
    {
        // ---- axon records: per slice, compact (2 B/axon) when its axons allow it, else wide (8 B/axon) ----
        std::vector<unsigned long long> rec_off(h.n_slices, 0);
        std::vector<uint8_t> mode(h.n_slices, 0), slat(h.n_slices, 0);
        std::vector<uint32_t> chunk0(h.n_slices, 0);
        std::vector<uint64_t> n_slice_chunks(h.n_slices, 0);
        uint64_t n_chunks = 0, n_bytes = 0, n_gather_only = 0;
        bool any_exact = false;
        bool any_last_cores = false;
        for (uint32_t g = 0; g < h.n_slots && !any_last_cores; g++)
            any_last_cores = (h.slot_cls[g] & 7u) != SANAFE_SOMA_NONE &&
                             (((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_LAST || ((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_LAST_DELAY);
        const bool runs_format = dict16 || c->syn_format == 0 || c->syn_format == 3 || c->syn_format == 4; // deliver_kernel: RUNS
        bool bitmap_records = c->syn_format == 7 && !any_last_cores; // deliver_kernel: BITMAP_RECORDS
        if (const char *env = std::getenv("SANAFE_AXON_BITMAP")) // tests / A-B runs: 0 keeps the 2-byte delta records
            if (std::atoi(env) == 0) bitmap_records = false;
        parallel_for(h.n_slices, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t sl = lo; sl < hi; sl++)
            {
                const uint64_t b0 = h.slice_axon_beg[sl], e0 = h.slice_axon_end[sl];
                bool ok = e0 > b0 && h.ax_lat_class != nullptr && h.ax_lat_class[b0] != 255u;
                for (uint64_t a = b0; a < e0 && ok; a++)
                {
                    ok = h.ax_nsyn[a] < 256u && h.ax_lat_class[a] == h.ax_lat_class[b0] &&
                         (h.ax_nsyn[a] > 0u || !runs_format); // (the run formats mark records past the end with a zero count)
                    if (ok && ((a - b0) % WAVE_CHUNK) != 0) ok = h.ax_pre[a] >= h.ax_pre[a - 1] && h.ax_pre[a] - h.ax_pre[a - 1] < 256u;
                }
                mode[sl] = ok ? 1 : 0;
                slat[sl] = ok ? h.ax_lat_class[b0] : 0;
                // bitmap records: compact-eligible, strictly ascending source slots, no lost charge, a quarter of the span in use
                if (ok && bitmap_records)
                {
                    bool bm = true;
                    for (uint64_t a = b0 + 1; a < e0 && bm; a++) bm = h.ax_pre[a] > h.ax_pre[a - 1];
                    const uint64_t windows = (uint64_t) (h.ax_pre[e0 - 1] >> 8) - (h.ax_pre[b0] >> 8) + 1ull;
                    bm = bm && (e0 - b0) * 4ull >= windows * 256ull;
                    const uint32_t core = h.slice_core[sl];
                    for (uint64_t a = b0; a < e0 && bm; a++)
                        for (uint32_t k = 0; k < h.ax_nsyn[a] && bm; k++)
                            bm = !((h.syn_meta[h.core_syn_base[core] + h.ax_syn_beg[a] + k] >> 19) & 1u);
                    if (bm) mode[sl] = 2;
                }
            }
        });
        // one kernel per chip: bitmap records only when EVERY compact slice can take them (deliver_kernel<..., BITMAP>)
        if (bitmap_records)
        {
            bool all = true;
            for (uint32_t sl = 0; sl < h.n_slices; sl++) all = all && mode[sl] != 1;
            if (!all)
                for (uint32_t sl = 0; sl < h.n_slices; sl++)
                    if (mode[sl] == 2) mode[sl] = 1;
        }
        for (uint32_t sl = 0; sl < h.n_slices; sl++)
        {
            const uint64_t n = h.slice_axon_end[sl] - h.slice_axon_beg[sl];
            if (n >= (1ull << 32)) return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "slice %u holds 2^32 axons or more", sl));
            rec_off[sl] = n_bytes;
            // chunks of a slice: 256 consecutive axons -- or, on bitmap records, the 256-slot windows of its source span
            n_slice_chunks[sl] = (mode[sl] == 2) ? (uint64_t) (h.ax_pre[h.slice_axon_end[sl] - 1] >> 8) - (h.ax_pre[h.slice_axon_beg[sl]] >> 8) + 1ull
                                                 : (n + WAVE_CHUNK - 1) / WAVE_CHUNK;
            n_bytes += (((mode[sl] == 2) ? n_slice_chunks[sl] * 32ull + n : n * (mode[sl] ? 2ull : 8ull)) + 15ull) & ~15ull;
            if (mode[sl] == 2) n_gather_only += n; // the synapse-count bytes: read for windows on the gather path only
            chunk0[sl] = (uint32_t) n_chunks;
            n_chunks += n_slice_chunks[sl] + 1; // + end entry: first synapse after the slice
            if (n_chunks >= (1ull << 32)) return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "too many axon chunks"));
        }
        // Where each chunk's synapses live on the device.  Formats 1 and 2 keep the image's order.  Format 0 gives
        // every chunk a 16-byte aligned start and pads its end to 16 bytes with words that drop their charge, so
        // a chunk can be streamed with aligned 16-byte loads and no boundary tests.
        std::vector<uint64_t> chunk_pos(n_chunks, 0);  // absolute device position of the chunk's first synapse
        std::vector<uint64_t> dev_core_base(h.core_syn_base, h.core_syn_base + h.n_cores);
        uint64_t n_dev_syn = h.n_synapses;
        // first axon of every chunk (+ the end entry of each slice)
        std::vector<uint64_t> chunk_ax(n_chunks, 0);
        parallel_for(h.n_slices, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t sl = lo; sl < hi; sl++)
            {
                const uint64_t b0 = h.slice_axon_beg[sl], e0 = h.slice_axon_end[sl], nck = n_slice_chunks[sl];
                if (mode[sl] == 2)
                {
                    const uint32_t w0 = h.ax_pre[b0] >> 8;
                    uint64_t a = b0;
                    for (uint64_t k = 0; k <= nck; k++)
                    {
                        while (a < e0 && (uint64_t) (h.ax_pre[a] >> 8) - w0 < k) a++;
                        chunk_ax[chunk0[sl] + k] = a;
                    }
                }
                else
                    for (uint64_t k = 0; k <= nck; k++) chunk_ax[chunk0[sl] + k] = std::min<uint64_t>(b0 + k * WAVE_CHUNK, e0);
            }
        });
        auto chunk_syn_count = [&](uint32_t sl, uint64_t k) { // synapses of chunk k of slice sl
            const uint64_t a0 = chunk_ax[chunk0[sl] + k], a1 = chunk_ax[chunk0[sl] + k + 1];
            if (a1 == a0) return uint64_t{0}; // (a window no axon starts in)
            return (uint64_t) (h.ax_syn_beg[a1 - 1] + h.ax_nsyn[a1 - 1] - h.ax_syn_beg[a0]);
        };
        if (stream_layout)
        {
            uint64_t pos = 0;
            uint32_t prev_core = 0xffffffffu;
            for (uint32_t k = 0; k < h.n_cores; k++) dev_core_base[k] = 0;
            for (uint32_t sl = 0; sl < h.n_slices; sl++)
            {
                const uint32_t core = h.slice_core[sl];
                if (core != prev_core) dev_core_base[core] = pos;
                prev_core = core;
                const uint64_t nck = n_slice_chunks[sl];
                for (uint64_t k = 0; k < nck; k++)
                {
                    chunk_pos[chunk0[sl] + k] = pos;
                    pos += (chunk_syn_count(sl, k) + group_words - 1ull) / group_words * group_words;
                }
                chunk_pos[chunk0[sl] + nck] = pos;
                if (pos - dev_core_base[core] > 0xffffffffull)
                    return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "more than 2^32 synapses on core %u", core));
            }
            n_dev_syn = pos;
        }
        else
        {
            for (uint32_t sl = 0; sl < h.n_slices; sl++)
            {
                const uint64_t b0 = h.slice_axon_beg[sl], e0 = h.slice_axon_end[sl];
                const uint64_t cbase = h.core_syn_base[h.slice_core[sl]];
                const uint64_t nck = (e0 - b0 + WAVE_CHUNK - 1) / WAVE_CHUNK;
                for (uint64_t k = 0; k < nck; k++) chunk_pos[chunk0[sl] + k] = cbase + h.ax_syn_beg[b0 + k * WAVE_CHUNK];
                chunk_pos[chunk0[sl] + nck] = (e0 > b0) ? cbase + h.ax_syn_beg[e0 - 1] + h.ax_nsyn[e0 - 1] : cbase;
            }
        }
        TRYC(upload(c, reinterpret_cast<const unsigned long long *>(dev_core_base.data()), h.n_cores, &im.core_syn_base));
        std::vector<unsigned char> bytes(n_bytes + 4096 + 16, 0); // (the delivery kernel reads up to 8 chunks of records at fixed offsets)
        std::vector<uint32_t> csyn(n_chunks, 0), cpre(n_chunks, 0);
        std::vector<uint8_t> exact(h.n_slices, 0);
        std::vector<uint32_t> meta; // device synapse words (all formats but 2, which keeps the image's arrays)
        std::vector<double> wdev;   // format 4: the fp64 weights at the device positions of their words
        std::vector<uint16_t> meta16; // formats 6, 7: 2-byte words
        if (dict16) meta16.assign(n_dev_syn + 512, 0);
        else if (c->syn_format != 2 && c->syn_format != 8) meta.assign(n_dev_syn + 256, 0u);
        if (c->syn_format == 4) wdev.assign(n_dev_syn + 256, 0.0);
        auto lut_code = [&](double w) {
            for (uint32_t q = 0; q < 32; q++)
                if (std::memcmp(&c->weight_lut[q], &w, sizeof w) == 0) return q;
            return 0u;
        };
        parallel_for(h.n_slices, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t sl = lo; sl < hi; sl++)
            {
                const uint64_t b0 = h.slice_axon_beg[sl], e0 = h.slice_axon_end[sl];
                const uint32_t core = h.slice_core[sl];
                const uint32_t trash_post = (h.core_ncount[core] + 63u) & ~63u; // == npad of the core
                unsigned char *dst = bytes.data() + rec_off[sl];
                const uint64_t nck = n_slice_chunks[sl];
                const bool bm = mode[sl] == 2;
                const uint32_t w0 = (bm && e0 > b0) ? h.ax_pre[b0] >> 8 : 0u; // bitmap records: first 256-slot window of the slice
                for (uint64_t k = 0; k <= nck; k++) csyn[chunk0[sl] + k] = (uint32_t) (chunk_pos[chunk0[sl] + k] - dev_core_base[core]);
                if (bm) // the chunk table's second column: first axon of each window, relative to the slice (+ the end entry)
                    for (uint64_t k = 0; k <= nck; k++) cpre[chunk0[sl] + k] = (uint32_t) (chunk_ax[chunk0[sl] + k] - b0);
                if (stream_layout) // padding words of every chunk: weight 0 into the trash entry
                    for (uint64_t k = 0; k < nck; k++)
                        for (uint64_t q = chunk_pos[chunk0[sl] + k] + chunk_syn_count((uint32_t) sl, k); q < chunk_pos[chunk0[sl] + k + 1]; q++)
                        {
                            if (dict16) meta16[q] = (uint16_t) (trash_post << 6); // no first-synapse bit, trash entry
                            else meta[q] = trash_post << (c->syn_format == 3 ? 8 : 11); // code 0, weight 0: adds nothing wherever it lands
                        }
                for (uint64_t a = b0; a < e0; a++)
                {
                    const uint64_t rel = a - b0;
                    // the axon's chunk: 256 consecutive axons, or (bitmap records) the 256-slot window its source lies in
                    const uint64_t chunk_k = bm ? (uint64_t) (h.ax_pre[a] >> 8) - w0 : rel / WAVE_CHUNK;
                    const uint32_t in_chunk = bm ? (uint32_t) (a - chunk_ax[chunk0[sl] + chunk_k]) : (uint32_t) (rel % WAVE_CHUNK);
                    if (!bm && in_chunk == 0) cpre[chunk0[sl] + rel / WAVE_CHUNK] = h.ax_pre[a];
                    if (bm)
                    {
                        // one bit per source slot of the slice's windows, then one synapse-count byte per axon
                        const uint32_t bit = h.ax_pre[a] - (w0 << 8);
                        reinterpret_cast<uint32_t *>(dst)[bit >> 5] |= 1u << (bit & 31u);
                        dst[nck * 32ull + rel] = (unsigned char) h.ax_nsyn[a];
                    }
                    else if (mode[sl])
                    {
                        const uint32_t delta = (in_chunk == 0) ? 0u : h.ax_pre[a] - h.ax_pre[a - 1];
                        const uint16_t r16 = (uint16_t) (delta | (h.ax_nsyn[a] << 8));
                        std::memcpy(dst + 2 * rel, &r16, 2);
                    }
                    else
                    {
                        const uint32_t cls = h.ax_lat_class ? h.ax_lat_class[a] : 255u;
                        if (cls == 255u) exact[sl] = 1;
                        const unsigned long long r64 = (unsigned long long) h.ax_pre[a] | ((unsigned long long) h.ax_nsyn[a] << 32) | ((unsigned long long) cls << 48);
                        std::memcpy(dst + 8 * rel, &r64, 8);
                    }
                    if (c->syn_format == 2 || c->syn_format == 8) continue;
                    // synapse words: image position -> device position (same offset inside the chunk)
                    const uint64_t src = h.core_syn_base[core] + h.ax_syn_beg[a];
                    const uint64_t chunk_first = h.core_syn_base[core] + h.ax_syn_beg[chunk_ax[chunk0[sl] + chunk_k]];
                    const uint64_t dpos = chunk_pos[chunk0[sl] + chunk_k] + (src - chunk_first);
                    // index of the axon inside its 256-axon chunk; formats 0 and 4 add the chunk's place in a run of 8
                    const uint32_t code = in_chunk | (c->syn_format == 3 ? 0u : (uint32_t) (chunk_k & 7u) << 8);
                    for (uint32_t k = 0; k < h.ax_nsyn[a]; k++)
                    {
                        const uint32_t m = h.syn_meta[src + k];
                        const int w = (int) h.syn_weight[src + k];
                        if (stream_layout)
                        {
                            const bool drop = (m >> 19) & 1u;
                            const uint32_t idx = drop ? trash_post : ((m >> 16) & 7u) * (trash_post + 1u) + (m & 0xffffu);
                            if (dict16)
                                meta16[dpos + k] = (uint16_t) ((idx << 6) | (lut_code(h.syn_weight[src + k]) << 1) | (k == 0 ? 1u : 0u));
                            else if (c->syn_format == 0) meta[dpos + k] = code | (idx << 11) | ((uint32_t) (w & 0xff) << 24);
                            else if (c->syn_format == 3) meta[dpos + k] = code | (idx << 8) | ((uint32_t) (w & 0xfff) << 20);
                            else
                            {
                                meta[dpos + k] = code | (idx << 11);
                                wdev[dpos + k] = h.syn_weight[src + k];
                            }
                        }
                        else
                            meta[dpos + k] = (m & 0xfffffu) | ((uint32_t) (w & 0xfff) << 20);
                    }
                }
            }
        });
        for (uint8_t x : exact) any_exact |= (x != 0);
        TRYC(upload(c, bytes.data(), bytes.size(), &im.ax_bytes));
        {
            std::vector<double> lat255(256, 0.0);
            if (h.lat_class_per_event) std::copy(h.lat_class_per_event, h.lat_class_per_event + 255, lat255.begin());
            std::vector<uint32_t> n_core_slices(h.n_cores, 0);
            for (uint32_t sl = 0; sl < h.n_slices; sl++) n_core_slices[h.slice_core[sl]]++;
            std::vector<SliceDesc> desc(h.n_slices);
            for (uint32_t sl = 0; sl < h.n_slices; sl++)
            {
                const uint32_t core = h.slice_core[sl];
                SliceDesc &d = desc[sl];
                d.rec_off = rec_off[sl];
                d.syn_base = dev_core_base[core];
                d.a_beg = (mode[sl] == 2) ? (unsigned long long) (h.ax_pre[h.slice_axon_beg[sl]] >> 8) * 8ull // first 32-slot word of its windows
                                          : h.slice_axon_beg[sl];
                d.ain_lat = h.core_axon_in_latency[core];
                d.slice_lat = lat255[slat[sl]];
                d.n_ax = (mode[sl] == 2) ? (uint32_t) (n_slice_chunks[sl] * WAVE_CHUNK) // 256 source slots per chunk
                                         : (uint32_t) (h.slice_axon_end[sl] - h.slice_axon_beg[sl]);
                d.nbase = h.core_nbase[core];
                d.ncount = h.core_ncount[core];
                d.chunk0 = chunk0[sl];
                d.slice_id = sl;
                d.mode = mode[sl];
                d.inkind = (uint8_t) ((h.slot_cls[h.core_nbase[core]] >> 3) & 7u);
                d.shared = n_core_slices[core] > 1 ? 1 : 0;
                d.pad = 0;
            }
            // launch order: slices fed by this chip's own neurons only, then the ones that need the gathered bitmap
            std::vector<uint8_t> local_only(h.n_slices, 1);
            if (h.n_global_slots != h.n_slots)
                parallel_for(h.n_slices, [&](uint64_t lo, uint64_t hi) {
                    for (uint64_t sl = lo; sl < hi; sl++)
                        for (uint64_t a = h.slice_axon_beg[sl]; a < h.slice_axon_end[sl] && local_only[sl]; a++)
                            local_only[sl] = h.ax_pre[a] >= h.slot_offset && h.ax_pre[a] < h.slot_offset + h.n_slots;
                });
            std::vector<SliceDesc> ordered;
            ordered.reserve(h.n_slices);
            for (uint32_t sl = 0; sl < h.n_slices; sl++)
                if (local_only[sl]) ordered.push_back(desc[sl]);
            c->n_local_slices = (c->syn_format == 8) ? 0u : (uint32_t) ordered.size(); // ordered delivery runs in one launch, after the gather
            for (uint32_t sl = 0; sl < h.n_slices; sl++)
                if (!local_only[sl]) ordered.push_back(desc[sl]);
            TRYC(upload(c, ordered.data(), ordered.size(), &im.slice_desc));
        }
        TRYC(upload(c, csyn.data(), csyn.size(), &im.chunk_syn0));
        TRYC(upload(c, cpre.data(), cpre.size(), &im.chunk_pre0));
        TRYC(upload(c, h.ax_proc_delay, any_exact ? h.n_axons : 0, &im.ax_proc_delay));
        std::vector<double> lat(256, 0.0);
        if (h.lat_class_per_event) std::copy(h.lat_class_per_event, h.lat_class_per_event + 255, lat.begin());
        TRYC(upload(c, lat.data(), lat.size(), &im.lat_class));
        for (uint8_t m : mode) c->n_compact_slices += (m != 0), c->n_bitmap_slices += (m == 2);
        c->small_slices = h.n_slices > 0;
        for (uint32_t sl = 0; sl < h.n_slices; sl++) c->small_slices = c->small_slices && (h.slice_axon_end[sl] - h.slice_axon_beg[sl]) <= 2 * WAVE_CHUNK;
        if (const char *env = std::getenv("SANAFE_DELIVER_SMALL")) // tests: 0 keeps the 256-thread workgroups
            if (std::atoi(env) == 0) c->small_slices = false;
        // what one delivery launch reads when every chunk is streamed (sanafe_hip_layout_bytes)
        c->layout_bytes[0] = (c->syn_format == 2) ? h.n_synapses * 12ull : n_dev_syn * (c->syn_format == 4 ? 12ull : dict16 ? 2ull : 4ull);
        // (format 8: build_ordered below replaces [0] with the bytes of the per-accumulator lists)
        im.bitmap_run_len = 8;
        if (const char *env = std::getenv("SANAFE_BITMAP_RUN_LEN")) im.bitmap_run_len = (uint32_t) std::min(8L, std::max(1L, std::atol(env)));
        c->layout_bytes[1] = n_bytes - n_gather_only;
        c->layout_bytes[8] = n_gather_only;
        c->layout_bytes[2] = n_chunks * 8ull;
        c->layout_bytes[3] = (uint64_t) h.n_slices * sizeof(SliceDesc);
        c->layout_bytes[4] = h.n_global_slots / 8;
        im.syn_meta = nullptr;
        im.weight_lut = nullptr;
        im.syn_weight = nullptr;
        if (c->syn_format == 8)
        {
            TRYC(build_ordered(c, h)); // per-accumulator lists instead of per-axon synapse words
        }
        else if (c->syn_format == 2)
        {
            TRYC(upload(c, h.syn_meta, h.n_synapses, &im.syn_meta));
            TRYC(upload(c, h.syn_weight, h.n_synapses, &im.syn_weight));
        }
        else if (dict16)
        {
            const uint16_t *d16 = nullptr;
            TRYC(upload(c, meta16.data(), meta16.size(), &d16));
            im.syn_meta = reinterpret_cast<const uint32_t *>(d16);
            TRYC(upload(c, c->weight_lut.data(), c->weight_lut.size(), &im.weight_lut));
        }
        else
        {
            TRYC(upload(c, meta.data(), meta.size(), &im.syn_meta));
            if (c->syn_format == 4) TRYC(upload(c, wdev.data(), wdev.size(), &im.syn_weight));
        }
    }
    {
        std::vector<uint32_t> beg(h.n_cores + 1, 0);
        for (uint32_t s = 0; s < h.n_slices; s++) beg[h.slice_core[s] + 1]++;
        for (uint32_t k = 0; k < h.n_cores; k++) beg[k + 1] += beg[k];
        TRYC(upload(c, beg.data(), beg.size(), &im.core_slice_beg));
        c->h_core_slice_beg = beg;
    }
    DevState &st = c->st;
    TRYC(dalloc(c, h.n_slots, &st.v));
    TRYC(dalloc(c, h.n_slots, &st.icur));
    TRYC(dalloc(c, h.n_slots, &st.refrac));
    TRYC(dalloc(c, h.n_slots, &st.status));
    TRYC(dalloc(c, h.n_input, &st.in_pos));
    TRYC(dalloc(c, (size_t) h.ring_slots * h.n_slots, &st.ring));
    TRYC(dalloc(c, (size_t) h.ring_slots * h.n_slots, &st.ring_valid));
    im.has_last = 0;
    for (uint32_t g = 0; g < h.n_slots && !im.has_last; g++)
        im.has_last = ((h.slot_cls[g] & 7u) != SANAFE_SOMA_NONE &&
                              (((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_LAST || ((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_LAST_DELAY))
                ? 1 : 0;
    st.ring_last = nullptr;
    if (im.has_last) TRYC(dalloc(c, h.n_slots, &st.ring_last));
    st.arrived = nullptr;
    {
        bool any_gated = false;
        for (uint32_t g = 0; g < h.n_slots && !any_gated; g++)
            any_gated = (h.slot_cls[g] & 7u) != SANAFE_SOMA_NONE && ((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_GATED;
        if (any_gated && h.ring_slots < 7) return bail(fail(SANAFE_HIP_ERR_INVALID, "SANAFE_IN_GATED neurons need ring_slots >= 7"));
        if (any_gated || h.n_taps > 0)
        {
            TRYC(dalloc(c, h.n_slots, &st.arrived));
            c->force_delay_variant = true; // per-neuron write-back rules live in the HAS_DELAY kernels
        }
        st.tap_v = st.tap_in = nullptr;
        if (h.n_taps > 0)
        {
            TRYC(dalloc(c, (size_t) h.n_taps * 8, &st.tap_v));
            TRYC(dalloc(c, (size_t) h.n_taps * 8, &st.tap_in));
        }
    }
    // the local spike bitmap is this chip's window of the global one: local delivery can start right after the
    // neuron launch, and the multi-GPU exchange gathers in place
    // (+ zero words past the end: the last 256-slot window of bitmap axon records, the bit padding entries of the ordered layout probe)
    TRYC(dalloc(c, h.n_global_slots / 32 + 64, &st.bits_global)); // (+ the rest of the event kernel's last 1,024-slot tile)
    st.bits_local = st.bits_global + h.slot_offset / 32;
    TRYC(dalloc(c, 2 * (size_t) im.n_wgs * PARTS_PER_WG, &st.wg_part));
    TRYC(dalloc(c, 2 * (size_t) h.n_slices, &st.slice_proc));
    TRYC(dalloc(c, 2 * (size_t) im.n_groups, &st.group_part));
    st.delay_log = nullptr;
    st.delay_log_cap = 0;
    st.host_proc = nullptr;
    if (im.n_msg_cores > 0) TRYC(dalloc(c, 2 * (size_t) h.n_cores, &st.host_proc)); // their processing delays (msgsoma_finish_kernel)
    TRYC(dalloc(c, 1, &st.t));
    TRYC(dalloc(c, 1, &st.rec));
    TRYC(dalloc(c, 1, &st.run));
    st.step_log = nullptr;
    st.spike_log = nullptr;
    st.log_cap = 1;
    if (h.slot_v0)
    {
        c->v0.assign(h.slot_v0, h.slot_v0 + h.n_slots);
        HIPC(hipMemcpy(st.v, h.slot_v0, (size_t) h.n_slots * sizeof(double), hipMemcpyHostToDevice));
    }
    c->neuron_grid = im.n_reduce_wgs + im.n_wgs;
    {
        // per neuron slot and step: bytes the neuron kernel reads / writes (see neuron_kernel)
        uint64_t live = 0;
        for (uint32_t k = 0; k < h.n_cores; k++) live += h.core_ncount[k];
        const uint64_t rd = (c->uni ? 0u : 4u) + 8u + ((c->uni && c->us.bias_uniform) ? 0u : 8u) + 8u + 1u + (im.has_lif ? 8u : 0u) + (im.any_refrac ? 4u : 0u) + (h.n_ext ? 4u : 0u);
        const uint64_t wr = 8u + 1u + (im.has_lif ? 8u : 0u) + (im.any_refrac ? 4u : 0u);
        c->layout_bytes[5] = live * rd;
        c->layout_bytes[6] = live * wr;
        c->layout_bytes[7] = sizeof(SpikeStatic);
    }
    {
        // LDS accumulator rows: one per synaptic delay value actually present in the image
        std::atomic<uint32_t> max_delay_seen{0};
        parallel_for(h.n_synapses, [&](uint64_t lo, uint64_t hi) {
            uint32_t m = 0;
            for (uint64_t k = lo; k < hi; k++) m = std::max(m, (h.syn_meta[k] >> 16) & 7u);
            uint32_t seen = max_delay_seen.load();
            while (seen < m && !max_delay_seen.compare_exchange_weak(seen, m)) {}
        });
        const uint32_t max_delay = max_delay_seen.load();
        if (max_delay >= h.ring_slots) return bail(fail(SANAFE_HIP_ERR_INVALID, "synaptic delay %u needs more than %u ring slots", max_delay, h.ring_slots));
        im.delay_slots = max_delay + 1;
    }
    {
        // a gated delay line matures one slot later than a plain one: writes go to slot (t + 1 + d + 1) % R
        bool gated = false;
        for (uint32_t g = 0; g < h.n_slots && !gated; g++)
            gated = (h.slot_cls[g] & 7u) != SANAFE_SOMA_NONE && ((h.slot_cls[g] >> 3) & 7u) == SANAFE_IN_GATED;
        if (gated && im.delay_slots + 1 > h.ring_slots)
            return bail(fail(SANAFE_HIP_ERR_INVALID, "SANAFE_IN_GATED neurons with synaptic delay %u need %u ring slots (have %u)",
                    im.delay_slots - 1, im.delay_slots + 1, h.ring_slots));
    }
    c->deliver_lds = (size_t) im.delay_slots * (max_pad + 1) * (sizeof(double) + 1);
    c->has_delay = im.delay_slots > 1 || c->force_delay_variant;
    im.syn_format = c->syn_format;
    im.acc_shift = c->acc_shift;
    if (im.has_last && c->has_delay)
        return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "cores with the buffer before the dendrite unit cannot be mixed with synaptic delays"));
    // The one delivery kernel this chip launches, picked from the table every instantiation lives in (deliver_variants):
    // opt in to its dynamic LDS and check dynamic + STATIC shared memory against the 160 KiB of a CU here, not at the
    // first launch.
    st.push_core_cnt = nullptr;
    st.host_events = nullptr;
    st.ev_part = nullptr;
    im.ev_groups = 0;
    im.ev_always = 0;
    {
        // (before the kernel is picked: chips with push tables or the event layout run the PUSH instantiations)
        const bool force_event = std::getenv("SANAFE_EVENT") != nullptr && std::atoi(std::getenv("SANAFE_EVENT")) >= 1;
        if (force_event) TRYC(build_event(c, h));
        if (im.ev_groups == 0u) TRYC(build_push(c, h));
        // chips too big for push tables (a global atomic per event) get the event layout (LDS accumulators) when it pays
        // (SANAFE_PUSH=0 -- the A/B switch for "streaming delivery only" -- keeps the event layout away as well)
        const bool push_off = std::getenv("SANAFE_PUSH") != nullptr && std::atoi(std::getenv("SANAFE_PUSH")) == 0;
        if (im.ev_groups == 0u && im.push_cap == 0u && !force_event && !push_off) TRYC(build_event(c, h));
        if (im.push_cap != 0u && im.push_always == 0u)
        {
            // the ring the device publishes every step's event count in (reduce_l2), read by the launch loop
            HIPC(hipHostMalloc(reinterpret_cast<void **>(&c->h_events), (size_t) HOST_EVENT_RING * 2 * sizeof(long long), hipHostMallocMapped));
            std::memset(c->h_events, 0, (size_t) HOST_EVENT_RING * 2 * sizeof(long long));
            HIPC(hipHostGetDevicePointer(reinterpret_cast<void **>(&st.host_events), c->h_events, 0));
        }
    }
    if (c->syn_format == 8)
    {
        // the whole spike bitmap in LDS when it leaves room for four workgroups per CU
        const size_t bitmap = (size_t) h.n_global_slots / 8 + 16;
        bool lds_bits = bitmap <= 36 * 1024;
        if (const char *env = std::getenv("SANAFE_ORDERED_LDS_BITS")) // tests: 0 probes the bitmap in global memory
            if (std::atoi(env) == 0) lds_bits = false;
        c->ord_lds = lds_bits ? bitmap : 0;
        if (const char *env = std::getenv("SANAFE_ORDERED_ROLE")) c->ord_debug = std::atoi(env);
        for (const OrderedVariant &v : ordered_variants)
            if (v.dict == c->ord_dict && v.lds_bits == lds_bits && v.delay == c->has_delay) c->deliver_fn = v.fn;
        if (c->ord_lds > 0) HIPC(hipFuncSetAttribute(c->deliver_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->ord_lds));
        c->deliver_block = 256;
    }
    else
    {
        const bool iacc = (c->syn_format == 0 || c->syn_format == 3) && c->acc_shift > 0;
        // (the 64-thread variant is built for the plain kernels only: no synaptic delays, no last-event cores, fp64 or
        //  dictionary accumulators)
        const bool use_small = c->small_slices && !im.has_last && !c->has_delay && !iacc;
        // sub-accumulators (deliver_kernel<..., SUB>): bitmap records, cores of at most 256 neurons, no synaptic delays
        bool sub = c->n_bitmap_slices > 0 && !use_small && !c->has_delay && max_pad <= SUB_MAX_NEURONS;
        if (const char *env = std::getenv("SANAFE_SUB_ACCUMULATORS")) // tests / A-B runs: 0 keeps one accumulator per neuron
            if (std::atoi(env) == 0) sub = false;
        c->sub_accumulators = sub;
        const DeliverVariant *v = find_deliver_variant(c->syn_format, !im.has_last && c->has_delay, im.has_last != 0, iacc, use_small ? 64 : DELIVER_BLOCK,
                c->n_bitmap_slices > 0, false, sub);
        if (v == nullptr)
            return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "no delivery kernel for format %d (delay %d, last %d, integer accumulators %d)",
                    c->syn_format, (int) c->has_delay, im.has_last, (int) iacc));
        hipFuncAttributes fa{};
        HIPC(hipFuncGetAttributes(&fa, v->fn));
        if (c->deliver_lds + fa.sharedSizeBytes > 160 * 1024)
            return bail(fail(SANAFE_HIP_ERR_UNSUPPORTED, "a core with %u neurons x %u delay values needs %zu + %zu B of LDS (> 160 KiB)",
                    max_pad, im.delay_slots, c->deliver_lds, (size_t) fa.sharedSizeBytes));
        HIPC(hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->deliver_lds));
        c->deliver_fn = v->fn;
        c->deliver_block = (uint32_t) v->block;
        if (im.ev_groups != 0u)
        {
#define SANAFE_EV(L, B, W, U) \
    if (c->ev_lpb == L && im.ev_code_bits == B && c->ev_waves == W && c->ev_upl == U) \
        c->event_fn = reinterpret_cast<const void *>(event_deliver_kernel<L, B, W, U, false>), \
        c->event_fn_sparse = reinterpret_cast<const void *>(event_deliver_kernel<L, B, W, U, true>)
#define SANAFE_EVW(L, B, U) SANAFE_EV(L, B, 4, U); SANAFE_EV(L, B, 8, U); SANAFE_EV(L, B, 16, U)
            SANAFE_EVW(4, 4, 1); SANAFE_EVW(8, 4, 1); SANAFE_EVW(4, 4, 2); SANAFE_EVW(4, 5, 1); SANAFE_EVW(8, 5, 1); SANAFE_EVW(4, 5, 2);
#undef SANAFE_EVW
#undef SANAFE_EV
        }
    }
    HIPC(hipDeviceSynchronize());
    *out = c;
    return 0;
}

extern "C" void sanafe_hip_chip_destroy(sanafe_hip_chip *c)
{
    if (!c) return;
    (void) hipSetDevice(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    for (void *p : c->allocs) (void) hipFree(p);
    if (c->st.step_log) (void) hipFree(c->st.step_log);
    if (c->st.spike_log) (void) hipFree(c->st.spike_log);
    if (c->st.status_log) (void) hipFree(c->st.status_log);
    if (c->st.msg_fired_log) (void) hipFree(c->st.msg_fired_log);
    if (c->st.delay_log) (void) hipFree(c->st.delay_log);
    for (void *p : {(void *) c->d_log_slots_v, (void *) c->d_log_slots_u, (void *) c->d_state_log})
        if (p) (void) hipFree(p);
    for (void *p : {(void *) c->d_host_slots, (void *) c->d_host_core, (void *) c->d_host_status, (void *) c->d_host_a,
                 (void *) c->d_host_b, (void *) c->d_host_costs, (void *) c->d_ext, (void *) c->d_soma_classes,
                 (void *) c->d_in_beg, (void *) c->d_in_len, (void *) c->d_in_bits, (void *) c->d_in_period})
        if (p) (void) hipFree(p);
    if (c->h_events && std::getenv("SANAFE_DEBUG_DECIDE"))
        std::fprintf(stderr, "[sanafe_hip] decide: %lld steps, %lld pushed, %lld waited for the device (%.2f ms in all), %lld gave up\n",
                c->t_host, c->pushed_steps, c->dbg_waits, c->dbg_wait_ms, c->dbg_fallbacks);
    if (c->h_events) (void) hipHostFree(c->h_events);
    if (c->stream && c->own_stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

static int ensure_log(sanafe_hip_chip *c, long long steps, bool with_status)
{
    if (c->st.step_log && c->st.log_cap >= steps && (!with_status || c->st.status_log)) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    steps = std::max(steps, c->st.step_log ? c->st.log_cap : 1LL);
    if (c->st.step_log) HIPCHK(hipFree(c->st.step_log));
    if (c->st.spike_log) HIPCHK(hipFree(c->st.spike_log));
    if (c->st.status_log) HIPCHK(hipFree(c->st.status_log));
    c->st.step_log = nullptr;
    c->st.spike_log = nullptr;
    c->st.status_log = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->st.step_log), (size_t) steps * sizeof(sanafe_hip_totals)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->st.spike_log), (size_t) steps * (c->im.n_slots / 32) * sizeof(uint32_t)));
    if (with_status)
    {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->st.status_log), (size_t) steps * c->im.n_slots));
        HIPCHK(hipMemset(c->st.status_log, 0, (size_t) steps * c->im.n_slots)); // padding slots stay 0
        if (c->st.msg_fired_log) HIPCHK(hipFree(c->st.msg_fired_log));
        c->st.msg_fired_log = nullptr;
        if (c->im.n_msg_axons > 0) HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->st.msg_fired_log), (size_t) steps * c->im.n_msg_axons * sizeof(uint16_t)));
    }
    c->st.log_cap = steps;
    return 0;
}

static int ensure_state_log(sanafe_hip_chip *c, long long steps)
{
    const size_t row = (size_t) c->n_log_v + c->n_log_u;
    if (row == 0) return fail(SANAFE_HIP_ERR_INVALID, "record bit 3 needs the slots to log (sanafe_hip_set_state_log)");
    if (c->d_state_log && c->state_log_cap >= steps) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->d_state_log) HIPCHK(hipFree(c->d_state_log));
    c->d_state_log = nullptr;
    c->state_log_cap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_state_log), (size_t) steps * row * sizeof(double)));
    c->state_log_cap = steps;
    return 0;
}
static int launch_state_log(sanafe_hip_chip *c, long long rec_index)
{
    const uint32_t n = c->n_log_v + c->n_log_u;
    double *row = c->d_state_log + (size_t) (rec_index % c->state_log_cap) * n;
    hipLaunchKernelGGL(state_log_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->st.v, c->st.icur, c->d_log_slots_v, c->n_log_v,
            c->d_log_slots_u, c->n_log_u, row, row + c->n_log_v);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int sanafe_hip_set_state_log(sanafe_hip_chip *c, uint32_t n_v, const uint32_t *slots_v, uint32_t n_u, const uint32_t *slots_u)
{
    if (!c || (n_v > 0 && !slots_v) || (n_u > 0 && !slots_u)) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n_v; i++)
        if (slots_v[i] >= c->im.n_slots) return fail(SANAFE_HIP_ERR_INVALID, "logged slot %u out of range", i);
    for (uint32_t i = 0; i < n_u; i++)
        if (slots_u[i] >= c->im.n_slots) return fail(SANAFE_HIP_ERR_INVALID, "logged slot %u out of range", i);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (uint32_t **p : {&c->d_log_slots_v, &c->d_log_slots_u})
    {
        if (*p) HIPCHK(hipFree(*p));
        *p = nullptr;
    }
    if (c->d_state_log) HIPCHK(hipFree(c->d_state_log));
    c->d_state_log = nullptr;
    c->state_log_cap = 0;
    c->n_log_v = c->n_log_u = 0;
    if (n_v > 0)
    {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_log_slots_v), (size_t) n_v * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(c->d_log_slots_v, slots_v, (size_t) n_v * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (n_u > 0)
    {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_log_slots_u), (size_t) n_u * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(c->d_log_slots_u, slots_u, (size_t) n_u * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->n_log_v = n_v;
    c->n_log_u = n_u;
    return 0;
}
extern "C" int sanafe_hip_read_step_state(sanafe_hip_chip *c, int64_t first, int64_t count, double *out)
{
    if (!c || !out || first < 0 || count < 0 || first + count > c->state_log_cap || !c->d_state_log)
        return fail(SANAFE_HIP_ERR_INVALID, "state records [%lld, %lld) not available", (long long) first, (long long) (first + count));
    const size_t row = (size_t) c->n_log_v + c->n_log_u;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->d_state_log + (size_t) first * row, (size_t) count * row * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Push / pull (event / stream) for the step with Timestep::timestep `t`: pushed when step t - DECISION_LAG (rounded down to a
// multiple of DECISION_STRIDE: every fourth step publishes) caused at most push_max_events synaptic events.  A pure function of the simulation's own history, so every run decides alike; the count
// comes from the pinned ring reduce_l2 publishes in.  Level 2 of step t - DECISION_LAG rode in a neuron launch enqueued
// DECISION_LAG - 2 steps ago (or in a flush): if the device has not got there yet, wait -- the host then leads the device by
// at most DECISION_LAG - 2 steps, which is queue enough to hide the launch costs.
static int decide_pushed(sanafe_hip_chip *c, long long t)
{
    if (c->im.push_cap == 0u) return 0;
    if (c->im.push_always != 0u)
    {
        c->cur_sparse = c->ev_sparse_max_events != 0u ? 1 : 0; // (forced modes know no event counts: the table by the environment)
        return 1;
    }
    c->cur_sparse = 0;
    const long long n = (t - DECISION_LAG) / DECISION_STRIDE * DECISION_STRIDE; // the last step at or before t - LAG that published
    if (n < c->epoch_first_step || c->h_events == nullptr) return 0; // no history yet: pull
    volatile long long *e = c->h_events + 2 * (n % HOST_EVENT_RING);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long long spins = 0;; spins++)
    {
        if (__atomic_load_n(&e[1], __ATOMIC_ACQUIRE) == n)
        {
            if (spins > 0)
            {
                c->dbg_waits++;
                c->dbg_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            const long long events = __atomic_load_n(&e[0], __ATOMIC_RELAXED);
            c->cur_sparse = events <= (long long) c->ev_sparse_max_events ? 1 : 0;
            return events <= (long long) c->im.push_max_events ? 1 : 0;
        }
        if ((spins & 0x3fffu) == 0x3fffu)
        {
            // Waited long (the device normally publishes within a few steps' time): is the stream idle and the entry not there
            // (a step reduced before this chip's ring existed), or the device stuck (the next synchronising call reports it)?
            // Then pull.  (No hipStreamQuery in the common case: on this runtime every query costs the device ~3 us of the
            // step it interrupts -- C2 ran 17 % slower with a query every few thousand spins.)
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited < std::chrono::milliseconds(50)) continue;
            const bool idle = hipStreamQuery(c->stream) == hipSuccess;
            if (idle && __atomic_load_n(&e[1], __ATOMIC_ACQUIRE) == n) continue;
            if (idle || waited > std::chrono::seconds(20))
            {
                c->dbg_fallbacks++;
                c->dbg_wait_ms += std::chrono::duration<double, std::milli>(waited).count();
                if (std::getenv("SANAFE_DEBUG_DECIDE"))
                    std::fprintf(stderr, "[sanafe_hip] decide: step %lld wants the events of step %lld, ring holds step %lld (idle %d)\n", t, n,
                            (long long) e[1], (int) idle);
                return 0;
            }
        }
    }
}

// ---- one timestep = neuron launch (which first reduces the previous step) + delivery launch ----
static int launch_neurons(sanafe_hip_chip *c, int record, long long rec_index)
{
    StepArgs sa{};
    if (c->im.n_ext > 0)
    {
        if (c->ext_next >= c->ext_rows)
            return fail(SANAFE_HIP_ERR_INVALID, "external value streams exhausted: queue rows with sanafe_hip_write_ext before stepping");
        sa.ext_row = c->d_ext + (size_t) c->ext_next * c->im.n_ext;
        c->ext_next++;
    }
    sa.t = c->t_host + 1;
    sa.parity = (int) (c->t_host & 1);
    sa.push_buf = (int) (c->t_host % 3);
    c->cur_pushed = decide_pushed(c, sa.t);
    sa.pushed = (c->cur_pushed && c->im.ev_groups == 0u) ? 1 : 0; // (event chips: event_deliver_kernel delivers, not this launch)
    const size_t rslot = (size_t) (sa.t % c->im.ring_slots);
    sa.ring = c->st.ring + rslot * c->im.n_slots;
    sa.rvalid = c->st.ring_valid + rslot * c->im.n_slots;
    const size_t nslot = (size_t) ((sa.t + 1) % c->im.ring_slots);
    sa.ring_next = c->st.ring + nslot * c->im.n_slots;
    sa.rvalid_next = c->st.ring_valid + nslot * c->im.n_slots;
    sa.ev_part = (c->im.ev_groups != 0u && c->ev_pending == c->t_host) ? c->st.ev_part : nullptr; // the previous step went by events
    if (record) sa.slog = c->st.spike_log + (size_t) (rec_index % c->st.log_cap) * (c->im.n_slots / 32);
    c->cur_slog = sa.slog;
    c->cur_msg_log = ((record & 2) && c->st.msg_fired_log) ? c->st.msg_fired_log + (size_t) (rec_index % c->st.log_cap) * c->im.n_msg_axons : nullptr;
    if (record & 2) sa.stlog = c->st.status_log + (size_t) (rec_index % c->st.log_cap) * c->im.n_slots;
    const dim3 grid(c->neuron_grid), block(NEURON_BLOCK);
#define SANAFE_LAUNCH_NEURON(M, U) \
    hipLaunchKernelGGL((neuron_kernel<M, U>), grid, block, 0, c->stream, c->im, c->st, sa, c->us, c->pend1, c->pend2)
    if (c->uni && c->neuron_model == SANAFE_SOMA_LIF) SANAFE_LAUNCH_NEURON(SANAFE_SOMA_LIF, true);
    else if (c->uni && c->neuron_model == SANAFE_SOMA_TRUENORTH) SANAFE_LAUNCH_NEURON(SANAFE_SOMA_TRUENORTH, true);
    else if (c->neuron_model == SANAFE_SOMA_LIF) SANAFE_LAUNCH_NEURON(SANAFE_SOMA_LIF, false);
    else if (c->neuron_model == SANAFE_SOMA_TRUENORTH) SANAFE_LAUNCH_NEURON(SANAFE_SOMA_TRUENORTH, false);
    else SANAFE_LAUNCH_NEURON(0, false);
#undef SANAFE_LAUNCH_NEURON
    HIPCHK(hipGetLastError());
    c->pend2 = c->pend1; // level 1 rides in that launch; level 2 in the next one
    c->pend1.valid = 0;
    return 0;
}
static int launch_deliver_slices(sanafe_hip_chip *c, uint32_t first, uint32_t count);
// Delivers the slices [first, first + count) of the launch order; with the last of them, the cores whose soma is part of the
// message pipeline (msgsoma_kernel: one soma update per synaptic event, after everything else of the step).
static int launch_deliver(sanafe_hip_chip *c, uint32_t first, uint32_t count)
{
    TRY(launch_deliver_slices(c, first, count));
    if (c->im.n_msg_cores != 0u && first + count == c->im.n_slices)
    {
        hipLaunchKernelGGL(msgsoma_kernel, dim3(c->im.n_msg_chunks), dim3(WAVE), 0, c->stream, c->im, c->st, c->cur_slog);
        hipLaunchKernelGGL(msgsoma_finish_kernel, dim3(std::max((c->im.n_msg_cores + 255u) / 256u, std::min(64u, (c->im.n_msg_axons + 255u) / 256u))), dim3(256), 0, c->stream, c->im,
                c->st,
                (int) (c->t_host & 1), c->cur_msg_log);
        HIPCHK(hipGetLastError());
    }
    return 0;
}
static int launch_deliver_slices(sanafe_hip_chip *c, uint32_t first, uint32_t count)
{
    if (c->im.ev_groups != 0u && c->cur_pushed)
    {
        // a step the host decided to deliver by events: the event kernel takes the whole source space at once -- it goes with
        // the last slices (tile-sharded chips deliver in two calls, local slices first) -- and leaves the next step's input in
        // DevState::ev_part
        if (first + count == c->im.n_slices)
        {
            long long done = c->t_host;
            c->sparse_steps += c->cur_sparse;
            void *args[] = {&c->im, &c->st, &done};
            HIPCHK(hipLaunchKernel(c->cur_sparse ? c->event_fn_sparse : c->event_fn, dim3(c->ev_grid), dim3(64u * (uint32_t) c->ev_waves), args, 0, c->stream));
            c->ev_pending = c->t_host + 1;
        }
        return 0;
    }
    if (c->cur_pushed)
    {
        // the neuron launch delivered this chip's own spikes itself; on a tile-sharded chip the other ranks' spikes are in the
        // gathered bitmap by now (this call follows the exchange) and are pushed here, through the same tables
        if (c->im.n_global_slots != c->im.n_slots && first + count == c->im.n_slices)
        {
            const size_t nslot = (size_t) ((c->t_host + 2) % c->im.ring_slots); // the NEXT step's row, as in launch_neurons
            hipLaunchKernelGGL(remote_push_kernel, dim3((c->im.n_global_slots / 32u + WAVE - 1) / WAVE), dim3(WAVE), 0, c->stream, c->im, c->st,
                    c->st.ring + nslot * c->im.n_slots, c->st.ring_valid + nslot * c->im.n_slots, (int) (c->t_host % 3));
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    if (c->syn_format == 8)
    {
        // ordered delivery: one launch for the whole chip -- every wavefront folds an accumulator group, then walks
        // delivery slices for their processing-delay sums (n_local_slices is 0: nothing runs before the gathered bitmap is there)
        if (count == 0 || first + count != c->im.n_slices) return 0;
        long long done = c->t_host;
        DevImage im = c->im;
        if (c->ord_debug == 1) im.ord_walk_slices = 0; // measurements only (SANAFE_ORDERED_ROLE): accumulator groups alone
        else if (c->ord_debug == 2) im.ord_groups = 0; // ... or the processing-delay walk alone
        void *args[] = {&im, &c->st, &done};
        HIPCHK(hipLaunchKernel(c->deliver_fn, dim3(im.ord_wgs), dim3(256), args, c->ord_lds, c->stream));
        return 0;
    }
    if (count > 0)
    {
        long long done = c->t_host;
        void *args[] = {&c->im, &c->st, &done, &first};
        HIPCHK(hipLaunchKernel(c->deliver_fn, dim3(count), dim3(c->deliver_block), args, c->deliver_lds, c->stream));
    }
    return 0;
}
static int launch_taps(sanafe_hip_chip *c)
{
    if (c->im.n_taps == 0) return 0;
    hipLaunchKernelGGL(taps_kernel, dim3((c->im.n_taps + 255) / 256), dim3(256), 0, c->stream, c->im, c->st, c->t_host);
    HIPCHK(hipGetLastError());
    return 0;
}
// The reduction of the step just launched is left pending: the next neuron launch performs it, or flush_pending.
static void finish_step(sanafe_hip_chip *c, int simple_timing, int record, long long rec_index)
{
    c->pend1.valid = 1;
    c->pend1.simple_timing = simple_timing;
    c->pend1.record = record;
    c->pend1.parity = (int) (c->t_host & 1);
    c->pend1.push_buf = (int) (c->t_host % 3);
    c->pend1.pushed = c->cur_pushed;
    c->pushed_steps += c->cur_pushed ? 1 : 0;
    c->pend1.rec_index = rec_index;
    c->t_host += 1;
}
static int flush_pending(sanafe_hip_chip *c)
{
    const bool flushed = c->pend1.valid || c->pend2.valid;
    while (c->pend1.valid || c->pend2.valid)
    {
        const uint32_t grid = c->pend1.valid ? std::max(1u, c->im.n_reduce_wgs) : 1u;
        hipLaunchKernelGGL(reduce_kernel, dim3(grid), dim3(REDUCE_BLOCK), 0, c->stream, c->im, c->st, c->pend1, c->pend2);
        HIPCHK(hipGetLastError());
        c->pend2 = c->pend1;
        c->pend1.valid = 0;
    }
    // (push / pull decisions need no reset here: level 2 of every step -- in a neuron launch or in reduce_kernel above --
    //  renews the word of the step three after it, reduce_l2)
    (void) flushed;
    return 0;
}

// Chips with the event layout: a step delivered by events leaves the next step's input in DevState::ev_part; readers of the
// time-step buffer on the host get it folded into the buffer row first.
static int fold_event_partials(sanafe_hip_chip *c)
{
    if (c->im.ev_groups == 0u || c->ev_pending != c->t_host) return 0;
    hipLaunchKernelGGL(event_fold_kernel, dim3((c->im.n_slots + 255) / 256), dim3(256), 0, c->stream, c->im, c->st, (long long) c->t_host);
    HIPCHK(hipGetLastError());
    c->ev_pending = -1;
    return 0;
}

extern "C" int sanafe_hip_step(sanafe_hip_chip *c, int64_t n_steps, int simple_timing, int record)
{
    if (!c || n_steps < 0) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (record)
    {
        TRY(flush_pending(c)); // an earlier step's record must not land in the resized log
        TRY(ensure_log(c, n_steps, (record & 2) != 0));
        if (record & 8) TRY(ensure_state_log(c, n_steps));
        HIPCHK(hipMemsetAsync(c->st.rec, 0, sizeof(long long), c->stream));
    }
    if (!c->timing)
    {
        for (int64_t s = 0; s < n_steps; s++)
        {
            TRY(launch_neurons(c, record, s));
            if (record & 8) TRY(launch_state_log(c, s));
            TRY(launch_deliver(c, 0, c->im.n_slices));
            TRY(launch_taps(c));
            finish_step(c, simple_timing, record, s);
        }
        return 0;
    }
    // Timed mode (bench.py roofline block): HIP events on the kernels' own stream.
    struct Events // destroyed on every exit path
    {
        std::vector<hipEvent_t> v;
        ~Events()
        {
            for (hipEvent_t e : v)
                if (e) (void) hipEventDestroy(e);
        }
    } events;
    events.v.assign((size_t) n_steps * 4, nullptr);
    std::vector<hipEvent_t> &ev = events.v;
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (int64_t s = 0; s < n_steps; s++)
    {
        TRY(flush_pending(c)); // timed steps keep the three phases apart
        HIPCHK(hipEventRecord(ev[s * 4 + 0], c->stream));
        TRY(launch_neurons(c, record, s));
        HIPCHK(hipEventRecord(ev[s * 4 + 1], c->stream));
        TRY(launch_deliver(c, 0, c->im.n_slices));
        TRY(launch_taps(c));
        HIPCHK(hipEventRecord(ev[s * 4 + 2], c->stream));
        finish_step(c, simple_timing, record, s);
        TRY(flush_pending(c));
        HIPCHK(hipEventRecord(ev[s * 4 + 3], c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int64_t s = 0; s < n_steps; s++)
    {
        float a = 0, b = 0, d = 0;
        HIPCHK(hipEventElapsedTime(&a, ev[s * 4 + 0], ev[s * 4 + 1]));
        HIPCHK(hipEventElapsedTime(&b, ev[s * 4 + 1], ev[s * 4 + 2]));
        HIPCHK(hipEventElapsedTime(&d, ev[s * 4 + 2], ev[s * 4 + 3]));
        c->t_neuron += a;
        c->t_deliver += b;
        c->t_reduce += d;
    }
    c->t_launches += n_steps;
    return 0;
}

extern "C" int sanafe_hip_write_ext(sanafe_hip_chip *c, int64_t n_steps, const int32_t *values)
{
    if (!c || n_steps < 0 || (n_steps > 0 && !values)) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    if (c->im.n_ext == 0) return fail(SANAFE_HIP_ERR_INVALID, "this chip has no external value streams");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream)); // steps still in flight read the rows being replaced
    if (n_steps > c->ext_cap)
    {
        if (c->d_ext) HIPCHK(hipFree(c->d_ext));
        c->d_ext = nullptr;
        c->ext_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_ext), (size_t) n_steps * c->im.n_ext * sizeof(int)));
        c->ext_cap = n_steps;
    }
    if (n_steps > 0) HIPCHK(hipMemcpy(c->d_ext, values, (size_t) n_steps * c->im.n_ext * sizeof(int), hipMemcpyHostToDevice));
    c->ext_rows = n_steps;
    c->ext_next = 0;
    return 0;
}

extern "C" int sanafe_hip_get_bitmap_slices(sanafe_hip_chip *c) { return c ? (int) c->n_bitmap_slices : 0; }
extern "C" int sanafe_hip_get_sub_accumulators(sanafe_hip_chip *c) { return (c && c->sub_accumulators) ? 1 : 0; }
extern "C" int sanafe_hip_get_acc_shift(sanafe_hip_chip *c)
{
    return c ? c->acc_shift : 0;
}
extern "C" int sanafe_hip_get_push_info(sanafe_hip_chip *c, uint32_t *enabled, uint32_t *pushed_steps)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (enabled) *enabled = c->im.push_cap != 0u ? (c->im.push_always != 0u ? 2u : 1u) : 0u;
    if (pushed_steps)
    {
        *pushed_steps = 0;
        if (c->im.push_cap != 0u) *pushed_steps = (uint32_t) c->pushed_steps; // (counted where the decision is made: on the host)
    }
    return 0;
}
extern "C" int sanafe_hip_get_msg_cores(sanafe_hip_chip *c) { return c ? (int) c->im.n_msg_cores : 0; }
extern "C" int sanafe_hip_get_event_info(sanafe_hip_chip *c, uint64_t *info, int n)
{
    if (!c || !info || n < SANAFE_HIP_EVENT_INFO_FIELDS) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    info[0] = c->im.ev_groups;
    info[1] = c->im.ev_groups ? c->im.ev_segments : 0;
    info[2] = c->layout_bytes[9] / 16u;
    info[3] = (uint64_t) (c->ev_avg_block * 1000.0);
    info[4] = c->im.ev_groups ? (uint64_t) c->ev_lpb : 0;
    info[5] = c->im.ev_groups ? c->im.ev_code_bits : 0;
    info[6] = c->im.ev_groups ? (uint64_t) c->im.ev_shift : 0;
    info[7] = c->im.ev_groups ? c->im.ev_always : 0;
    info[8] = c->im.ev_groups ? c->im.push_max_events : 0;
    info[9] = c->im.ev_groups ? c->ev_sparse_max_events : 0;
    info[10] = (uint64_t) c->sparse_steps;
    return 0;
}
extern "C" int sanafe_hip_get_layout(sanafe_hip_chip *c, int *syn_format, uint32_t *n_compact_slices)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (syn_format) *syn_format = c->syn_format;
    if (n_compact_slices) *n_compact_slices = c->n_compact_slices;
    return 0;
}

extern "C" int sanafe_hip_layout_bytes(sanafe_hip_chip *c, uint64_t *out, int n)
{
    if (!c || !out || n < SANAFE_HIP_LAYOUT_FIELDS) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    for (int k = 0; k < SANAFE_HIP_LAYOUT_FIELDS; k++) out[k] = c->layout_bytes[k];
    return 0;
}

extern "C" int sanafe_hip_synchronize(sanafe_hip_chip *c)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Timed mode of the split step (multi-GPU loop): events queue up here and are resolved by sanafe_hip_read_timing.
static int timed_event(sanafe_hip_chip *c)
{
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventRecord(e, c->stream));
    c->split_events.push_back(e);
    return 0;
}

extern "C" int sanafe_hip_record_begin(sanafe_hip_chip *c, int64_t n_steps, int record)
{
    if (!c || n_steps < 0) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    c->split_record = 0;
    c->split_index = 0;
    if (record == 0) return 0;
    TRY(flush_pending(c)); // an earlier step's record must not land in the resized log
    TRY(ensure_log(c, std::max<int64_t>(n_steps, 1), (record & 2) != 0));
    if (record & 8) TRY(ensure_state_log(c, std::max<int64_t>(n_steps, 1)));
    HIPCHK(hipMemsetAsync(c->st.rec, 0, sizeof(long long), c->stream));
    c->split_record = record;
    return 0;
}

extern "C" int sanafe_hip_step_neurons(sanafe_hip_chip *c)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    if (c->split_record && c->split_index >= c->st.log_cap) return fail(SANAFE_HIP_ERR_INVALID, "more split steps than sanafe_hip_record_begin announced");
    if (!c->timing)
    {
        TRY(launch_neurons(c, c->split_record, c->split_index));
        if (c->split_record & 8) TRY(launch_state_log(c, c->split_index));
        return 0;
    }
    TRY(flush_pending(c));
    TRY(timed_event(c));
    TRY(launch_neurons(c, c->split_record, c->split_index));
    if (c->split_record & 8) TRY(launch_state_log(c, c->split_index));
    return timed_event(c);
}

extern "C" int sanafe_hip_step_deliver(sanafe_hip_chip *c, int simple_timing, int record)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    if (record)
    {
        TRY(flush_pending(c));
        TRY(ensure_log(c, 1, false));
    }
    if (!c->timing)
    {
        TRY(launch_deliver(c, 0, c->im.n_slices));
        TRY(launch_taps(c));
        finish_step(c, simple_timing, c->split_record, c->split_index);
        if (c->split_record) c->split_index++;
        return 0;
    }
    TRY(timed_event(c));
    TRY(launch_deliver(c, 0, c->im.n_slices));
    TRY(launch_taps(c));
    TRY(timed_event(c));
    finish_step(c, simple_timing, c->split_record, c->split_index);
    if (c->split_record) c->split_index++;
    TRY(flush_pending(c));
    return timed_event(c);
}

extern "C" int sanafe_hip_step_deliver_local(sanafe_hip_chip *c)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    return launch_deliver(c, 0, c->n_local_slices);
}
extern "C" int sanafe_hip_step_deliver_remote(sanafe_hip_chip *c, int simple_timing)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    TRY(launch_deliver(c, c->n_local_slices, c->im.n_slices - c->n_local_slices));
    TRY(launch_taps(c));
    finish_step(c, simple_timing, c->split_record, c->split_index);
    if (c->split_record) c->split_index++;
    return 0;
}
extern "C" int sanafe_hip_slice_split(sanafe_hip_chip *c, uint32_t *n_local, uint32_t *n_remote)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (n_local) *n_local = c->n_local_slices;
    if (n_remote) *n_remote = c->im.n_slices - c->n_local_slices;
    return 0;
}
extern "C" int sanafe_hip_delay_log(sanafe_hip_chip *c, int64_t capacity, double **log, int64_t *real_capacity, int64_t *next_index)
{
    if (!c || capacity < 0) return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (capacity > c->st.delay_log_cap)
    {
        if (c->st.delay_log) HIPCHK(hipFree(c->st.delay_log));
        c->st.delay_log = nullptr;
        c->st.delay_log_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->st.delay_log), (size_t) capacity * sizeof(double)));
        c->st.delay_log_cap = capacity;
    }
    if (log) *log = c->st.delay_log;
    if (real_capacity) *real_capacity = c->st.delay_log_cap; // the log never shrinks: ring arithmetic uses THIS, not the request
    if (next_index) *next_index = c->st.delay_log_cap > 0 ? c->t_host % c->st.delay_log_cap : 0;
    return 0;
}
extern "C" int sanafe_hip_read_delay_log(sanafe_hip_chip *c, int64_t first, int64_t count, double *out)
{
    if (!c || !out || first < 0 || count < 0 || first + count > c->st.delay_log_cap)
        return fail(SANAFE_HIP_ERR_INVALID, "delay log entries [%lld, %lld) not available", (long long) first, (long long) (first + count));
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    HIPCHK(hipMemcpyAsync(out, c->st.delay_log + first, (size_t) count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" void *sanafe_hip_run_totals_device(sanafe_hip_chip *c) { return c ? c->st.run : nullptr; }

template <typename T> static int d2h(sanafe_hip_chip *c, T *dst, const T *src, size_t n);

extern "C" int sanafe_hip_spike_buffers(sanafe_hip_chip *c, void **local_bits, uint64_t *local_bytes, void **global_bits,
        uint64_t *global_bytes)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (local_bits) *local_bits = c->st.bits_local;
    if (local_bytes) *local_bytes = (uint64_t) c->im.n_slots / 8;
    if (global_bits) *global_bits = c->st.bits_global;
    if (global_bytes) *global_bytes = (uint64_t) c->im.n_global_slots / 8;
    return 0;
}

extern "C" int sanafe_hip_export_spikes(sanafe_hip_chip *c, uint32_t *local_bits_out)
{
    if (!c || !local_bits_out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    return d2h(c, local_bits_out, c->st.bits_local, (size_t) c->im.n_slots / 32);
}
extern "C" int sanafe_hip_import_spikes(sanafe_hip_chip *c, const uint32_t *global_bits)
{
    if (!c || !global_bits) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->st.bits_global, global_bits, (size_t) c->im.n_global_slots / 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" void *sanafe_hip_stream(sanafe_hip_chip *c) { return c ? c->stream : nullptr; }

extern "C" int sanafe_hip_set_stream(sanafe_hip_chip *c, void *stream)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->own_stream) HIPCHK(hipStreamDestroy(c->stream));
    c->stream = static_cast<hipStream_t>(stream);
    c->own_stream = false;
    return 0;
}

template <typename T> static int d2h(sanafe_hip_chip *c, T *dst, const T *src, size_t n)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
template <typename T> static int h2d(sanafe_hip_chip *c, T *dst, const T *src, size_t n)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sanafe_hip_read_totals(sanafe_hip_chip *c, sanafe_hip_totals *out)
{
    if (!c || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    return d2h(c, out, c->st.run, 1);
}
extern "C" int sanafe_hip_reset_totals(sanafe_hip_chip *c)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    HIPCHK(hipMemsetAsync(c->st.run, 0, sizeof(sanafe_hip_totals), c->stream));
    return 0;
}
extern "C" int sanafe_hip_read_step_totals(sanafe_hip_chip *c, int64_t first, int64_t count, sanafe_hip_totals *out)
{
    if (!c || !out || first < 0 || count < 0 || first + count > c->st.log_cap || !c->st.step_log)
        return fail(SANAFE_HIP_ERR_INVALID, "step records [%lld, %lld) not available", (long long) first, (long long) (first + count));
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    return d2h(c, out, c->st.step_log + first, (size_t) count);
}
extern "C" int sanafe_hip_read_step_spikes(sanafe_hip_chip *c, int64_t index, uint32_t *bits_out)
{
    if (!c || !bits_out || index < 0 || index >= c->st.log_cap || !c->st.spike_log)
        return fail(SANAFE_HIP_ERR_INVALID, "spike record %lld not available", (long long) index);
    const size_t words = c->im.n_slots / 32;
    return d2h(c, bits_out, c->st.spike_log + (size_t) index * words, words);
}
extern "C" int sanafe_hip_read_step_spike_rows(sanafe_hip_chip *c, int64_t first, int64_t count, uint32_t *bits_out)
{
    if (!c || !bits_out || first < 0 || count < 0 || first + count > c->st.log_cap || !c->st.spike_log)
        return fail(SANAFE_HIP_ERR_INVALID, "spike records [%lld, %lld) not available", (long long) first, (long long) (first + count));
    const size_t words = c->im.n_slots / 32; // the log is contiguous on the device: one copy for all rows
    return d2h(c, bits_out, c->st.spike_log + (size_t) first * words, (size_t) count * words);
}
extern "C" int sanafe_hip_read_step_status(sanafe_hip_chip *c, int64_t first, int64_t count, uint8_t *out)
{
    if (!c || !out || first < 0 || count < 0 || first + count > c->st.log_cap || !c->st.status_log)
        return fail(SANAFE_HIP_ERR_INVALID, "status records [%lld, %lld) not available", (long long) first, (long long) (first + count));
    return d2h(c, out, c->st.status_log + (size_t) first * c->im.n_slots, (size_t) count * c->im.n_slots);
}

extern "C" int sanafe_hip_read_step_msg_fired(sanafe_hip_chip *c, int64_t first, int64_t count, uint16_t *out)
{
    if (!c || !out || first < 0 || count < 0 || first + count > c->st.log_cap || !c->st.msg_fired_log)
        return fail(SANAFE_HIP_ERR_INVALID, "per-message fired counts of steps [%lld, %lld) not available", (long long) first, (long long) (first + count));
    return d2h(c, out, c->st.msg_fired_log + (size_t) first * c->im.n_msg_axons, (size_t) count * c->im.n_msg_axons);
}

extern "C" int sanafe_hip_read_status(sanafe_hip_chip *c, uint8_t *out)
{
    if (!c || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    return d2h(c, out, c->st.status, c->im.n_slots);
}
extern "C" int sanafe_hip_read_potentials(sanafe_hip_chip *c, double *out)
{
    if (!c || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    return d2h(c, out, c->st.v, c->im.n_slots);
}
extern "C" int sanafe_hip_read_input_current(sanafe_hip_chip *c, double *out)
{
    if (!c || !out) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    return d2h(c, out, c->st.icur, c->im.n_slots);
}
extern "C" int sanafe_hip_read_core_delays(sanafe_hip_chip *c, double *gen_sum, double *proc_sum)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    const size_t last_parity = (size_t) ((c->t_host > 0 ? c->t_host - 1 : 0) & 1); // the step launched last
    if (gen_sum) // as level 1 of the step reduction forms it
    {
        const size_t n_parts = (size_t) c->im.n_wgs * PARTS_PER_WG;
        std::vector<WgPart> wp(std::max<size_t>(1, n_parts));
        TRY(d2h(c, wp.data(), c->st.wg_part + last_parity * n_parts, n_parts));
        for (uint32_t k = 0; k < c->im.n_cores; k++)
        {
            double lat4[4] = {0.0, 0.0, 0.0, 0.0};
            long long packets = 0, counted = 0, upd = 0, fired = 0;
            const size_t wb = (size_t) c->h_core_wg_beg[k] * PARTS_PER_WG;
            for (size_t w = wb; w < (size_t) c->h_core_wg_beg[k + 1] * PARTS_PER_WG; w++)
                lat4[(w - wb) & 3u] += wp[w].lat, packets += wp[w].packets, counted += wp[w].counted, upd += wp[w].updated, fired += wp[w].fired;
            double lat = (lat4[0] + lat4[1]) + (lat4[2] + lat4[3]); // (reduce_l1: four lane-strided sums per core)
            if (c->im.uni_costing && counted != 0) // uniform chips: the default costing is priced per core (reduce_l1)
            {
                const sanafe_hip_cost_class &cc = c->im.uni_cost;
                const double n_all = (double) counted, n_f = (double) fired, n_u = (double) (upd - fired), n_i = (double) (counted - upd);
                lat += n_all * (0.0 + cc.dendrite_latency) + ((n_i * cc.soma_latency[0] + n_u * cc.soma_latency[1]) + n_f * cc.soma_latency[2]);
            }
            gen_sum[k] = lat + (double) packets * c->h_core_out_lat[k];
        }
    }
    if (proc_sum)
    {
        // pushed steps price a core's messages inside the step reduction (reduce_l1, from counters it clears): nothing to read back
        if (c->im.push_cap != 0u)
            return fail(SANAFE_HIP_ERR_UNSUPPORTED, "per-core processing delays are not kept on chips with push delivery "
                                                    "(create the chip with SANAFE_PUSH=0 to read them)");
        std::vector<double> sp(std::max<uint32_t>(1, c->im.n_slices));
        TRY(d2h(c, sp.data(), c->st.slice_proc + last_parity * c->im.n_slices, c->im.n_slices));
        for (uint32_t k = 0; k < c->im.n_cores; k++)
        {
            // the association of reduce_l1: lane q of the core's quad takes slices q, q + 4, ... in two running sums
            const uint32_t s0 = c->h_core_slice_beg[k], s1 = c->h_core_slice_beg[k + 1];
            double lane_sum[4];
            for (uint32_t q = 0; q < 4u; q++)
            {
                double a0 = 0.0, a1 = 0.0;
                uint32_t s = s0 + q;
                for (; s + 4u < s1; s += 8u) a0 += sp[s], a1 += sp[s + 4u];
                if (s < s1) a0 += sp[s];
                lane_sum[q] = a0 + a1;
            }
            proc_sum[k] = (lane_sum[0] + lane_sum[1]) + (lane_sum[2] + lane_sum[3]);
        }
    }
    return 0;
}

extern "C" int sanafe_hip_write_bias(sanafe_hip_chip *c, uint32_t first, uint32_t count, const double *bias)
{
    if (!c || !bias || (uint64_t) first + count > c->im.n_slots) return fail(SANAFE_HIP_ERR_INVALID, "bad slot range");
    c->us.bias_uniform = 0u; // the per-slot array is read again from now on
    return h2d(c, const_cast<double *>(c->im.slot_bias) + first, bias, count);
}
extern "C" int sanafe_hip_write_potential(sanafe_hip_chip *c, uint32_t first, uint32_t count, const double *v)
{
    if (!c || !v || (uint64_t) first + count > c->im.n_slots) return fail(SANAFE_HIP_ERR_INVALID, "bad slot range");
    return h2d(c, c->st.v + first, v, count);
}
extern "C" int sanafe_hip_write_slot_class(sanafe_hip_chip *c, uint32_t first, uint32_t count, const uint32_t *cls)
{
    if (!c || !cls || (uint64_t) first + count > c->im.n_slots) return fail(SANAFE_HIP_ERR_INVALID, "bad slot range");
    for (uint32_t k = 0; k < count; k++)
    {
        if (c->uni && cls[k] != c->us.cls) c->uni = false; // the chip is no longer one class word: table-driven kernel
        if (c->neuron_model != 0 && (cls[k] & 7u) != (uint32_t) c->neuron_model) c->neuron_model = 0, c->uni = false;
    }
    return h2d(c, const_cast<uint32_t *>(c->im.slot_cls) + first, cls, count);
}
extern "C" int sanafe_hip_write_inputs(sanafe_hip_chip *c, uint32_t n_input, const uint32_t *train_beg, const uint32_t *train_len,
        const int64_t *rate_period, const uint32_t *train_bits, uint64_t n_train_words, const uint8_t *rewind)
{
    if (!c || n_input != c->im.n_input) return fail(SANAFE_HIP_ERR_INVALID, "input count differs from the image's");
    if (n_input == 0) return 0;
    if (!train_beg || !train_len || !rate_period || (n_train_words > 0 && !train_bits) || !rewind)
        return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n_input; i++)
        if ((uint64_t) train_beg[i] + train_len[i] > n_train_words * 32ull)
            return fail(SANAFE_HIP_ERR_INVALID, "input %u: spike train outside train_bits", i);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream)); // steps in flight still read the old tables
    if (!c->d_in_beg)
    {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_in_beg), (size_t) n_input * sizeof(uint32_t)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_in_len), (size_t) n_input * sizeof(uint32_t)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_in_period), (size_t) n_input * sizeof(long long)));
    }
    if (n_train_words > c->in_bits_cap)
    {
        if (c->d_in_bits) HIPCHK(hipFree(c->d_in_bits));
        c->d_in_bits = nullptr;
        c->in_bits_cap = 0;
        const uint64_t cap = n_train_words + n_train_words / 2 + 64;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_in_bits), cap * sizeof(uint32_t)));
        c->in_bits_cap = cap;
    }
    HIPCHK(hipMemcpy(c->d_in_beg, train_beg, (size_t) n_input * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_in_len, train_len, (size_t) n_input * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_in_period, rate_period, (size_t) n_input * sizeof(long long), hipMemcpyHostToDevice));
    if (n_train_words > 0) HIPCHK(hipMemcpy(c->d_in_bits, train_bits, n_train_words * sizeof(uint32_t), hipMemcpyHostToDevice));
    // rewind the cursors: read, patch, write (n_input words)
    std::vector<uint32_t> pos(n_input);
    HIPCHK(hipMemcpy(pos.data(), c->st.in_pos, (size_t) n_input * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_input; i++)
        if (rewind[i]) pos[i] = 0;
    HIPCHK(hipMemcpy(c->st.in_pos, pos.data(), (size_t) n_input * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->im.in_train_beg = c->d_in_beg;
    c->im.in_train_len = c->d_in_len;
    c->im.in_rate_period = c->d_in_period;
    if (c->d_in_bits) c->im.in_train_bits = c->d_in_bits;
    return 0;
}

extern "C" int sanafe_hip_write_soma_classes(sanafe_hip_chip *c, uint32_t n, const sanafe_hip_soma_class *classes)
{
    if (!c || !classes || n == 0 || n > 65536) return fail(SANAFE_HIP_ERR_INVALID, "bad class table");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream)); // steps in flight still read the old table
    if (n > c->soma_class_cap)
    {
        // grow geometrically into a table of our own (the one uploaded at create stays in `allocs`)
        const uint32_t cap = std::min<uint32_t>(65536u, std::max<uint32_t>(n, 2u * c->soma_class_cap + 64u));
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, (size_t) cap * sizeof(sanafe_hip_soma_class)));
        if (c->d_soma_classes) (void) hipFree(c->d_soma_classes);
        c->d_soma_classes = static_cast<sanafe_hip_soma_class *>(p);
        c->soma_class_cap = cap;
    }
    HIPCHK(hipMemcpy(c->d_soma_classes, classes, (size_t) n * sizeof(sanafe_hip_soma_class), hipMemcpyHostToDevice));
    c->im.soma_classes = c->d_soma_classes;
    c->im.n_soma_classes = n;
    c->im.any_refrac = 0;
    for (uint32_t k = 0; k < n; k++) c->im.any_refrac |= classes[k].refractory_delay > 0;
    c->h_soma_classes.assign(classes, classes + n);
    if (c->uni)
    {
        if ((c->us.cls >> 16) < n) c->us.p = classes[c->us.cls >> 16];
        else c->uni = false;
    }
    return 0;
}
static int ensure_host_staging(sanafe_hip_chip *c, uint32_t count)
{
    if (c->host_cap >= count) return 0;
    for (void *p : {(void *) c->d_host_slots, (void *) c->d_host_core, (void *) c->d_host_status, (void *) c->d_host_a,
                 (void *) c->d_host_b})
        if (p) HIPCHK(hipFree(p));
    c->d_host_slots = c->d_host_core = nullptr;
    c->d_host_status = nullptr;
    c->d_host_a = c->d_host_b = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_host_slots), (size_t) count * sizeof(uint32_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_host_core), (size_t) count * sizeof(uint32_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_host_status), (size_t) count));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_host_a), (size_t) count * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_host_b), (size_t) count * sizeof(double)));
    c->host_cap = count;
    return 0;
}

extern "C" int sanafe_hip_read_host_inputs(sanafe_hip_chip *c, uint32_t count, const uint32_t *slots, double *current_out,
        uint8_t *has_out)
{
    if (!c || (count > 0 && (!slots || !current_out || !has_out))) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; i++)
        if (slots[i] >= c->im.n_slots) return fail(SANAFE_HIP_ERR_INVALID, "bad host slot %u", i);
    TRY(ensure_host_staging(c, count));
    HIPCHK(hipMemcpyAsync(c->d_host_slots, slots, (size_t) count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(host_input_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->im, c->st, count,
            c->d_host_slots, c->d_host_a, c->d_host_status, c->t_host);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(current_out, c->d_host_a, (size_t) count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(has_out, c->d_host_status, (size_t) count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sanafe_hip_write_host_status(sanafe_hip_chip *c, uint32_t count, const uint32_t *slots, const uint8_t *status,
        const uint32_t *core, const double *energy, const double *latency)
{
    if (!c || (count > 0 && (!slots || !status || !core || !energy || !latency))) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; i++)
        if (slots[i] >= c->im.n_slots || status[i] > 3 || core[i] >= c->im.n_cores)
            return fail(SANAFE_HIP_ERR_INVALID, "bad host status entry %u", i);
    TRY(ensure_host_staging(c, count));
    HIPCHK(hipMemcpyAsync(c->d_host_slots, slots, (size_t) count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_host_core, core, (size_t) count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_host_status, status, (size_t) count, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_host_a, energy, (size_t) count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_host_b, latency, (size_t) count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(host_status_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->im, c->st, count,
            c->d_host_slots, c->d_host_status, c->d_host_core, c->d_host_a, c->d_host_b, (int) (c->t_host & 1), 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sanafe_hip_write_host_core_status(sanafe_hip_chip *c, uint32_t count, const uint32_t *slots, const uint8_t *status,
        const uint32_t *core)
{
    if (!c || (count > 0 && (!slots || !status || !core))) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; i++)
        if (slots[i] >= c->im.n_slots || status[i] > 3 || core[i] >= c->im.n_cores || (c->im.slot_cls == nullptr))
            return fail(SANAFE_HIP_ERR_INVALID, "bad host status entry %u", i);
    TRY(ensure_host_staging(c, count));
    HIPCHK(hipMemcpyAsync(c->d_host_slots, slots, (size_t) count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_host_core, core, (size_t) count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_host_status, status, (size_t) count, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(host_status_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->im, c->st, count,
            c->d_host_slots, c->d_host_status, c->d_host_core, c->d_host_a, c->d_host_b, (int) (c->t_host & 1), 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sanafe_hip_write_host_core_costs(sanafe_hip_chip *c, uint32_t count, const sanafe_hip_host_core_costs *costs)
{
    if (!c || (count > 0 && !costs)) return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    if (c->t_host == 0) return fail(SANAFE_HIP_ERR_INVALID, "no timestep has been launched yet");
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; i++)
        if (costs[i].core >= c->im.n_cores) return fail(SANAFE_HIP_ERR_INVALID, "bad host core entry %u", i);
    if (c->timing)
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "per-kernel timing (sanafe_hip_set_timing) closes every step's reduction at once: switch it off on "
                                                "chips with cores that run on the host");
    if (c->pend1.valid == 0) return fail(SANAFE_HIP_ERR_INVALID, "the step's reduction has already run: write the host cores' costs right after its delivery");
    if (c->st.host_proc == nullptr) TRY(dalloc(c, 2 * (size_t) c->im.n_cores, &c->st.host_proc));
    if (count > c->host_cost_cap)
    {
        if (c->d_host_costs) HIPCHK(hipFree(c->d_host_costs));
        c->d_host_costs = nullptr;
        c->host_cost_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_host_costs), (size_t) count * sizeof(sanafe_hip_host_core_costs)));
        c->host_cost_cap = count;
    }
    HIPCHK(hipMemcpyAsync(c->d_host_costs, costs, (size_t) count * sizeof(sanafe_hip_host_core_costs), hipMemcpyHostToDevice, c->stream));
    // the step just launched: its partials live in the half t_host - 1 selects
    hipLaunchKernelGGL(host_core_costs_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->im, c->st, count, c->d_host_costs,
            (int) ((c->t_host - 1) & 1));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sanafe_hip_export_state(sanafe_hip_chip *c, sanafe_hip_state *o)
{
    if (!c || !o || !o->v || !o->icur || !o->refrac || !o->status || !o->ring || !o->ring_valid || (c->im.n_input > 0 && !o->in_pos))
        return fail(SANAFE_HIP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    TRY(fold_event_partials(c));
    const size_t n = c->im.n_slots, r = (size_t) c->im.ring_slots * n;
    o->timesteps = c->t_host;
    TRY(d2h(c, o->v, c->st.v, n));
    TRY(d2h(c, o->icur, c->st.icur, n));
    TRY(d2h(c, o->refrac, c->st.refrac, n));
    TRY(d2h(c, o->status, c->st.status, n));
    TRY(d2h(c, o->ring, c->st.ring, r));
    TRY(d2h(c, o->ring_valid, c->st.ring_valid, r));
    if (o->arrived && c->st.arrived) TRY(d2h(c, o->arrived, c->st.arrived, n));
    if (o->ring_last && c->st.ring_last) TRY(d2h(c, o->ring_last, c->st.ring_last, n));
    if (c->im.n_input > 0) TRY(d2h(c, o->in_pos, c->st.in_pos, (size_t) c->im.n_input));
    return 0;
}

extern "C" int sanafe_hip_import_state(sanafe_hip_chip *c, const sanafe_hip_state *in)
{
    if (!c || !in || !in->v || !in->icur || !in->refrac || !in->status || !in->ring || !in->ring_valid || (c->im.n_input > 0 && !in->in_pos) ||
            in->timesteps < 0)
        return fail(SANAFE_HIP_ERR_INVALID, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    TRY(flush_pending(c));
    c->ev_pending = -1; // the imported buffer rows hold all pending input
    const size_t n = c->im.n_slots, r = (size_t) c->im.ring_slots * n;
    TRY(h2d(c, c->st.v, in->v, n));
    TRY(h2d(c, c->st.icur, in->icur, n));
    TRY(h2d(c, c->st.refrac, in->refrac, n));
    TRY(h2d(c, c->st.status, in->status, n));
    TRY(h2d(c, c->st.ring, in->ring, r));
    TRY(h2d(c, c->st.ring_valid, in->ring_valid, r));
    if (in->arrived && c->st.arrived) TRY(h2d(c, c->st.arrived, in->arrived, n));
    if (in->ring_last && c->st.ring_last) TRY(h2d(c, c->st.ring_last, in->ring_last, n));
    if (c->im.n_input > 0) TRY(h2d(c, c->st.in_pos, in->in_pos, (size_t) c->im.n_input));
    const long long t = in->timesteps;
    TRY(h2d(c, c->st.t, &t, 1));
    c->t_host = t; // the step numbering (Timestep::timestep, ring rows, step parity) continues from there
    c->epoch_first_step = t + 1; // (the event counts of earlier steps are not in this chip's ring)
    return 0;
}

extern "C" int sanafe_hip_reset(sanafe_hip_chip *c)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    HIPCHK(hipSetDevice(c->device));
    const size_t n = c->im.n_slots;
    HIPCHK(hipMemsetAsync(c->st.v, 0, n * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->st.icur, 0, n * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->st.status, 0, n, c->stream));
    HIPCHK(hipMemsetAsync(c->st.ring, 0, (size_t) c->im.ring_slots * n * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->st.ring_valid, 0, (size_t) c->im.ring_slots * n, c->stream));
    if (c->st.ring_last) HIPCHK(hipMemsetAsync(c->st.ring_last, 0, n * sizeof(uint32_t), c->stream));
    if (c->st.arrived) HIPCHK(hipMemsetAsync(c->st.arrived, 0, n, c->stream));
    c->ev_pending = -1; // pending input of an event step is dropped like the buffer rows
    if (c->st.tap_v)
    {
        HIPCHK(hipMemsetAsync(c->st.tap_v, 0, (size_t) c->im.n_taps * 8 * sizeof(double), c->stream));
        HIPCHK(hipMemsetAsync(c->st.tap_in, 0, (size_t) c->im.n_taps * 8 * sizeof(double), c->stream));
    }
    return 0;
}

extern "C" int sanafe_hip_set_timing(sanafe_hip_chip *c, int enabled)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    // timed steps close every step's reduction at once; cores that run on the host write their costs after the delivery
    // launch (sanafe_hip_write_host_core_costs) and would find it closed
    if (enabled && c->st.host_proc != nullptr)
        return fail(SANAFE_HIP_ERR_UNSUPPORTED, "per-kernel timing is not available on chips with cores that run on the host");
    c->timing = enabled != 0;
    c->t_neuron = c->t_deliver = c->t_reduce = 0.0;
    c->t_launches = 0;
    for (hipEvent_t e : c->split_events) (void) hipEventDestroy(e);
    c->split_events.clear();
    return 0;
}
extern "C" int sanafe_hip_read_timing(sanafe_hip_chip *c, double *neuron_ms, double *deliver_ms, double *reduce_ms,
        int64_t *launches)
{
    if (!c) return fail(SANAFE_HIP_ERR_INVALID, "null chip");
    if (!c->split_events.empty())
    {
        // events of timed split steps: [n0, n1] per step_neurons, [d0, d1, r1] per step_deliver, in call order
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        size_t i = 0;
        const size_t n_ev = c->split_events.size();
        while (i + 5 <= n_ev)
        {
            float a = 0, b = 0, d = 0;
            HIPCHK(hipEventElapsedTime(&a, c->split_events[i], c->split_events[i + 1]));
            HIPCHK(hipEventElapsedTime(&b, c->split_events[i + 2], c->split_events[i + 3]));
            HIPCHK(hipEventElapsedTime(&d, c->split_events[i + 3], c->split_events[i + 4]));
            c->t_neuron += a;
            c->t_deliver += b;
            c->t_reduce += d;
            c->t_launches += 1;
            i += 5;
        }
        for (hipEvent_t e : c->split_events) (void) hipEventDestroy(e);
        c->split_events.clear();
    }
    const double n = c->t_launches > 0 ? (double) c->t_launches : 1.0;
    if (neuron_ms) *neuron_ms = c->t_neuron / n;
    if (deliver_ms) *deliver_ms = c->t_deliver / n;
    if (reduce_ms) *reduce_ms = c->t_reduce / n;
    if (launches) *launches = c->t_launches;
    return 0;
}
