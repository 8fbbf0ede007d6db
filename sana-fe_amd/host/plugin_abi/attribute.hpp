// attribute.hpp -- `sanafe::ModelAttribute`, the typed value a plugin receives in
// set_attribute_hw / _neuron / _edge.  Source-compatible with the reference's
// src/attribute.hpp:41-176: same variant alternatives in the same order, same conversion
// operators (int -> double and int -> bool are allowed, everything else throws), same public
// members `value`, `name`, `forward_to_synapse/dendrite/soma`.
#ifndef SANAFE_AMD_PLUGIN_ATTRIBUTE_HPP
#define SANAFE_AMD_PLUGIN_ATTRIBUTE_HPP
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "print.hpp"

namespace sanafe
{
struct ModelAttribute
{
    using Variant = std::variant<bool, int, double, std::string, std::vector<ModelAttribute>>;

    operator bool() const
    {
        if (const bool *b = std::get_if<bool>(&value)) return *b;
        if (const int *i = std::get_if<int>(&value)) return *i != 0;
        throw std::runtime_error("Error: Attribute " + name.value_or("") + " cannot be cast to a bool ()");
    }
    operator int() const { return std::get<int>(value); }
    operator double() const
    {
        if (const double *d = std::get_if<double>(&value)) return *d;
        if (const int *i = std::get_if<int>(&value)) return static_cast<double>(*i);
        throw std::runtime_error("Error: Attribute " + name.value_or("") + " cannot be cast to a double");
    }
    operator std::string() const { return std::get<std::string>(value); }
    template <typename T> operator std::vector<T>() const
    {
        std::vector<T> out;
        for (const ModelAttribute &e : std::get<std::vector<ModelAttribute>>(value)) out.push_back(static_cast<T>(e));
        return out;
    }
    bool is_list() const { return std::holds_alternative<std::vector<ModelAttribute>>(value); }
    bool operator==(const ModelAttribute &o) const
    {
        return value == o.value && forward_to_synapse == o.forward_to_synapse &&
                forward_to_dendrite == o.forward_to_dendrite && forward_to_soma == o.forward_to_soma;
    }
    bool operator!=(const ModelAttribute &o) const { return !(*this == o); }

    Variant value;
    std::optional<std::string> name;
    bool forward_to_synapse{true};
    bool forward_to_dendrite{true};
    bool forward_to_soma{true};
};
using AttributeVariant = ModelAttribute::Variant;
}
#endif
