// mapped.hpp -- `NeuronStatus` and the mapped-object views a plugin can see.
// The enum values are the reference's (src/mapped.hpp:22-28).  On the MI355X host the mapped
// network is columnar, so MappedNeuron / MappedConnection are thin views carrying the members
// the reference documents as plugin-visible.
#ifndef SANAFE_AMD_PLUGIN_MAPPED_HPP
#define SANAFE_AMD_PLUGIN_MAPPED_HPP
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "attribute.hpp"
#include "fwd.hpp"

namespace sanafe
{
enum NeuronStatus : uint8_t
{
    neuron_state_unset = 0U,
    idle = 1U,
    updated = 2U,
    fired = 3U
};

class HardwareMappingError : public std::runtime_error
{
public:
    explicit HardwareMappingError(const std::string &message) : std::runtime_error(message) {}
};

class MappedConnection
{
public:
    PipelineUnit *synapse_hw{nullptr};
    size_t mapped_synapse_hw_address{0UL};
};

class MappedNeuron
{
public:
    std::string parent_group_name;
    size_t offset{0UL};
    size_t id{0UL};
    PipelineUnit *dendrite_hw{nullptr};
    PipelineUnit *soma_hw{nullptr};
    size_t mapped_dendrite_hw_address{0UL};
    size_t mapped_soma_hw_address{0UL};
    NeuronStatus status{neuron_state_unset};
};
}
#endif
