// ref_models_driver.cpp -- thin C driver over the REFERENCE's own unit models.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
// the reference's model translation units taken unmodified, in place, from
// /root/reference/src (see oracle/Makefile) into oracle/_ref/, and lets the
// tests drive the reference's `PipelineUnit::update()` implementations
// (src/models.cpp, plugins/hodgkin_huxley.cpp) with arbitrary call sequences so
// the oracle's restated models can be pinned against them.
#include <cstdio>
#include <cstring>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "attribute.hpp"
#include "models.hpp"
#include "pipeline.hpp"
#include "plugins.hpp"

struct refm_unit
{
    std::shared_ptr<sanafe::PipelineUnit> hw;
};

struct refm_result
{
    int has_current;
    double current;
    int status;
    int has_energy;
    double energy;
    int has_latency;
    double latency;
};

static void fill(refm_result *out, const sanafe::PipelineResult &r)
{
    out->has_current = r.current.has_value();
    out->current = r.current.value_or(0.0);
    out->status = static_cast<int>(r.status);
    out->has_energy = r.energy.has_value();
    out->energy = r.energy.value_or(0.0);
    out->has_latency = r.latency.has_value();
    out->latency = r.latency.value_or(0.0);
}

static sanafe::ModelAttribute make_attr(const char *key, int type, double num, const char *str, const double *list,
        long list_len)
{
    sanafe::ModelAttribute a;
    a.name = std::string(key);
    switch (type)
    {
    case 0: a.value = (num != 0.0); break;
    case 1: a.value = static_cast<int>(num); break;
    case 2: a.value = num; break;
    case 3: a.value = std::string(str ? str : ""); break;
    default:
    {
        std::vector<sanafe::ModelAttribute> v;
        for (long i = 0; i < list_len; i++)
        {
            sanafe::ModelAttribute e;
            // lists of integral values are int attributes, like a YAML `[1, 0, 1]`
            if (list[i] == static_cast<double>(static_cast<int>(list[i]))) e.value = static_cast<int>(list[i]);
            else e.value = list[i];
            v.push_back(e);
        }
        a.value = v;
    }
    }
    return a;
}

#define GUARD(body)                                              \
    try                                                          \
    {                                                            \
        body;                                                    \
        return 0;                                                \
    }                                                            \
    catch (const std::exception &e)                              \
    {                                                            \
        if (err && errlen > 0) std::snprintf(err, errlen, "%s", e.what()); \
        return -1;                                               \
    }

extern "C" refm_unit *refm_create(const char *model, const char *plugin_path, char *err, int errlen)
{
    try
    {
        auto u = std::make_unique<refm_unit>();
        if (plugin_path && plugin_path[0]) u->hw = sanafe::plugin_get_hw(model, plugin_path);
        else u->hw = sanafe::model_get_pipeline_unit(model);
        return u.release();
    }
    catch (const std::exception &e)
    {
        if (err && errlen > 0) std::snprintf(err, errlen, "%s", e.what());
        return nullptr;
    }
}
extern "C" void refm_destroy(refm_unit *u) { delete u; }
extern "C" int refm_set_attr_hw(refm_unit *u, const char *key, int type, double num, const char *str,
        const double *list, long n, char *err, int errlen)
{
    GUARD(u->hw->set_attribute_hw(key, make_attr(key, type, num, str, list, n)))
}
extern "C" int refm_set_attr_neuron(refm_unit *u, long addr, const char *key, int type, double num, const char *str,
        const double *list, long n, char *err, int errlen)
{
    GUARD(u->hw->set_attribute_neuron(addr, key, make_attr(key, type, num, str, list, n)))
}
extern "C" int refm_set_attr_edge(refm_unit *u, long addr, const char *key, int type, double num, const char *str,
        const double *list, long n, char *err, int errlen)
{
    GUARD(u->hw->set_attribute_edge(addr, key, make_attr(key, type, num, str, list, n)))
}
extern "C" int refm_update_syn(refm_unit *u, long addr, int read, long t, refm_result *out, char *err, int errlen)
{
    GUARD(fill(out, u->hw->update(static_cast<size_t>(addr), read != 0, t)))
}
extern "C" int refm_update_dend(refm_unit *u, long naddr, int has_cur, double cur, int has_syn, long syn, long t,
        refm_result *out, char *err, int errlen)
{
    GUARD(fill(out,
            u->hw->update(static_cast<size_t>(naddr), has_cur ? std::optional<double>(cur) : std::nullopt,
                    has_syn ? std::optional<size_t>(syn) : std::nullopt, t)))
}
extern "C" int refm_update_soma(refm_unit *u, long naddr, int has_cur, double cur, long t, refm_result *out, char *err,
        int errlen)
{
    GUARD(fill(out, u->hw->update(static_cast<size_t>(naddr), has_cur ? std::optional<double>(cur) : std::nullopt, t)))
}
extern "C" double refm_get_potential(refm_unit *u, long addr) { return u->hw->get_potential(addr); }
extern "C" int refm_get_trace(refm_unit *u, long addr, const char *name, double *out)
{
    auto tr = u->hw->get_neuron_traces(addr);
    auto it = tr.find(name);
    if (it == tr.end()) return 0;
    *out = it->second;
    return 1;
}
extern "C" void refm_reset(refm_unit *u) { u->hw->reset(); }
