/* sanafe_desc.h -- plain-C, columnar description of an architecture plus a
 * mapped spiking network: the input of SpikingChip::load().
 *
 * This is a DATA FORMAT, not code from the reference.  It flattens what the
 * reference holds in `Architecture` (src/arch.hpp:70-205) and `SpikingNetwork`
 * (src/network.hpp:90-199) into arrays so that (a) the MI355X host mapper
 * (sana-fe_amd/host/mapper.cpp) and (b) the CPU oracle (oracle/) can both
 * consume exactly the same input without sharing any mapping code.
 *
 * Conventions
 *  - every name is an index into `strings` (-1 == empty string)
 *  - tiles, cores, units, groups, neurons and edges appear in CREATION order,
 *    which is what the reference's ordering rules are defined on
 *    (src/chip.cpp:186-234, 334-408; src/network.cpp:85-92)
 *  - attributes are generic (key, type, value[, list]) records, the columnar
 *    twin of `ModelAttribute` (src/attribute.hpp:41-176)
 */
#ifndef SANAFE_DESC_H
#define SANAFE_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ModelAttribute variant tags (src/attribute.hpp:170) */
enum { SANAFE_ATTR_BOOL = 0, SANAFE_ATTR_INT = 1, SANAFE_ATTR_DOUBLE = 2,
       SANAFE_ATTR_STRING = 3, SANAFE_ATTR_LIST = 4 };
/* forward_to_* flags (src/attribute.hpp:172-175) */
enum { SANAFE_FWD_SYNAPSE = 1, SANAFE_FWD_DENDRITE = 2, SANAFE_FWD_SOMA = 4 };
/* HardwareBitfield (src/pipeline.hpp:25-31) */
enum { SANAFE_IMPL_SYNAPSE = 1, SANAFE_IMPL_DENDRITE = 2, SANAFE_IMPL_SOMA = 4 };
/* unit flags */
enum { SANAFE_UNIT_LOG_ENERGY = 1, SANAFE_UNIT_LOG_LATENCY = 2,
       SANAFE_UNIT_UPDATE_EVERY_TIMESTEP = 4 };
/* BufferPosition (src/arch.hpp:41-49) */
enum { SANAFE_BUF_BEFORE_DENDRITE = 0, SANAFE_BUF_INSIDE_DENDRITE = 1,
       SANAFE_BUF_BEFORE_SOMA = 2, SANAFE_BUF_INSIDE_SOMA = 3,
       SANAFE_BUF_BEFORE_AXON_OUT = 4 };
/* hop directions: index into tile_hop_energy/latency rows */
enum { SANAFE_DIR_NORTH = 0, SANAFE_DIR_EAST = 1, SANAFE_DIR_SOUTH = 2,
       SANAFE_DIR_WEST = 3 };

typedef struct sanafe_attr_table
{
    int64_t n;               /* number of attribute records                 */
    const int32_t *key;      /* [n] string id                               */
    const uint8_t *type;     /* [n] SANAFE_ATTR_*                           */
    const uint8_t *fwd;      /* [n] SANAFE_FWD_* bits                       */
    const double *num;       /* [n] value for bool/int/double               */
    const int32_t *str;      /* [n] string id for SANAFE_ATTR_STRING        */
    const int64_t *list_ptr; /* [n+1] range into list_num for _LIST, else empty */
    const double *list_num;  /* flattened numeric list payloads             */
} sanafe_attr_table;

typedef struct sanafe_desc
{
    /* ---- string table ---- */
    int32_t n_strings;
    const char *const *strings;

    /* ---- network-on-chip (src/arch.hpp:122-129) ---- */
    int32_t noc_width, noc_height, noc_buffer_size;
    int32_t n_sync;            /* ts_sync_delay_table entries (src/utils.hpp:19-44) */
    const int64_t *sync_key;   /* [n_sync] ascending */
    const double *sync_val;    /* [n_sync] */

    /* ---- tiles ---- */
    int32_t n_tiles;
    const int32_t *tile_name;        /* [n_tiles] */
    const double *tile_hop_energy;   /* [n_tiles*4] N,E,S,W */
    const double *tile_hop_latency;  /* [n_tiles*4] N,E,S,W */
    const uint8_t *tile_log_energy;  /* [n_tiles] */

    /* ---- cores, global id order (tile order, then offset in tile) ---- */
    int32_t n_cores;
    const int32_t *core_name;        /* [n_cores] */
    const int32_t *core_tile;        /* [n_cores] parent tile id */
    const int32_t *core_buffer_pos;  /* [n_cores] SANAFE_BUF_* */
    const int64_t *core_max_neurons; /* [n_cores] */
    const uint8_t *core_log_energy;  /* [n_cores] */
    const int32_t *core_template;    /* [n_cores] index of the core's unit template */

    /* ---- core templates: the axon units and pipeline units of a core.  Cores
     * replicated from one description entry (`name[a..b]`) share a template;
     * every core still gets its OWN unit instances, created in template order
     * (src/chip.cpp:70-93). ---- */
    int32_t n_templates;
    const int32_t *tmpl_axon_in_ptr; /* [n_templates+1] */
    const double *axon_in_energy;    /* energy_message_in  */
    const double *axon_in_latency;   /* latency_message_in */
    const int32_t *tmpl_axon_out_ptr;/* [n_templates+1] */
    const double *axon_out_energy;   /* energy_message_out  */
    const double *axon_out_latency;  /* latency_message_out */
    const int32_t *tmpl_unit_ptr;    /* [n_templates+1] range into unit_* */

    /* ---- pipeline units of each template, creation order ---- */
    int32_t n_units;
    const int32_t *unit_name;        /* [n_units] */
    const int32_t *unit_model;       /* [n_units] model name (string id) */
    const int32_t *unit_plugin;      /* [n_units] plugin library path or -1 */
    const uint8_t *unit_implements;  /* [n_units] SANAFE_IMPL_* */
    const uint8_t *unit_flags;       /* [n_units] SANAFE_UNIT_* */
    const int64_t *unit_attr_ptr;    /* [n_units+1] range into unit_attrs */
    sanafe_attr_table unit_attrs;    /* ModelInfo::model_attributes */

    /* ---- neuron groups, creation order ---- */
    int32_t n_groups;
    const int32_t *group_name;       /* [n_groups] */
    const int64_t *group_ptr;        /* [n_groups+1] range of global neuron ids */

    /* ---- neurons: global id = group_ptr[g] + offset ---- */
    int64_t n_neurons;
    const int32_t *neuron_core;      /* [n] global core id, -1 = not mapped */
    const int64_t *neuron_map_order; /* [n] Neuron::mapping_order */
    const int32_t *neuron_soma_hw;   /* [n] soma_hw_name or -1 */
    const int32_t *neuron_dendrite_hw;/* [n] dendrite_hw_name or -1 */
    const int32_t *neuron_synapse_hw;/* [n] default_synapse_hw_name or -1 */
    const uint8_t *neuron_log_spikes;
    const uint8_t *neuron_log_potential;
    const int64_t *neuron_attr_ptr;  /* [n+1] range into neuron_attrs (key-sorted) */
    sanafe_attr_table neuron_attrs;  /* Neuron::model_attributes */

    /* ---- edges, creation order ---- */
    int64_t n_edges;
    const int64_t *edge_src;         /* [e] global neuron id */
    const int64_t *edge_dst;         /* [e] global neuron id */
    const int32_t *edge_synapse_hw;  /* [e] Connection::synapse_hw_name or -1 */
    const double *edge_weight;       /* [e] "w"/"weight" (0.0 if absent) */
    const int8_t *edge_delay;        /* [e] the edge's dendrite attribute: "d"/"delay" (0..5), or 64 + "tap" (tap index
                                        of a `taps` dendrite, src/models.cpp:330-342), or -1 if absent; may be NULL */
    const int64_t *edge_attr_ptr;    /* [e+1] extra attributes, may be NULL */
    sanafe_attr_table edge_attrs;
} sanafe_desc;

#ifdef __cplusplus
}
#endif
#endif /* SANAFE_DESC_H */
