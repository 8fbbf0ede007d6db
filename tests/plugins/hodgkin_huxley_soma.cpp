// hodgkin_huxley_soma.cpp -- example soma plugin for the MI355X host's plugin ABI.
//
// Same model as the reference's plugins/hodgkin_huxley.cpp (config C5), written against
// host/plugin_abi: exponential-Euler Hodgkin-Huxley membrane with dt = 0.1, one neuron per unit
// instance, ignores its synaptic current, fires when V crosses 25 mV upwards and reports
// `updated` otherwise; returns no energy/latency, so the architecture must give the soma unit
// default costs.  Exports `create_hodgkin_huxley`, the factory name the arch description's
// `model: hodgkin_huxley` resolves to.
#include <cmath>

#include "attribute.hpp"
#include "mapped.hpp"
#include "pipeline.hpp"
#include "print.hpp"

namespace
{
class HodgkinHuxleySoma : public sanafe::SomaUnit
{
public:
    HodgkinHuxleySoma() { register_attributes({"m", "n", "h", "current"}); }

    void set_attribute_hw(const std::string &, const sanafe::ModelAttribute &) override {}
    void set_attribute_neuron(size_t, const std::string &key, const sanafe::ModelAttribute &value) override
    {
        if (key == "m") m_ = static_cast<double>(value);
        else if (key == "n") n_ = static_cast<double>(value);
        else if (key == "h") h_ = static_cast<double>(value);
        else if (key == "current") i_ = static_cast<double>(value);
    }
    void reset() override { v_ = prev_v_ = m_ = n_ = h_ = 0.0; }
    double get_potential(size_t) override { return v_; }

    sanafe::PipelineResult update(size_t, std::optional<double>, long int) override
    {
        // gating rates
        const double an = (0.01 * (v_ + 55)) / (1 - exp(-0.1 * (v_ + 55)));
        const double am = (0.1 * (v_ + 40)) / (1 - exp(-0.1 * (v_ + 40)));
        const double ah = 0.07 * exp(-0.05 * (v_ + 65));
        const double bn = 0.125 * exp(-0.01125 * (v_ + 55));
        const double bm = 4 * exp(-0.05556 * (v_ + 65));
        const double bh = 1 / (1 + exp(-0.1 * (v_ + 35)));
        const double tau_n = 1 / (an + bn), tau_m = 1 / (am + bm), tau_h = 1 / (ah + bh);
        const double pm = am / (am + bm), pn = an / (an + bn), ph = ah / (ah + bh);
        // membrane: conductances, time constant, steady state
        const double den = g_l + g_k * (pow(n_, 4)) + g_na * (pow(m_, 3) * h_);
        const double tau_v = c_m / den;
        const double v_inf = ((g_l) *v_l + g_k * (pow(n_, 4)) * v_k + g_na * (pow(m_, 3)) * h_ * v_na + i_) / den;
        prev_v_ = v_;
        v_ = v_inf + (v_ - v_inf) * exp(-1 * dt / tau_v);
        m_ = pm + (m_ - pm) * exp(-1 * dt / tau_m);
        n_ = pn + (n_ - pn) * exp(-1 * dt / tau_n);
        h_ = ph + (h_ - ph) * exp(-1 * dt / tau_h);
        const sanafe::NeuronStatus status = ((prev_v_ < 25) && (v_ > 25)) ? sanafe::fired : sanafe::updated;
        return {std::nullopt, status, std::nullopt, std::nullopt};
    }

private:
    static constexpr double c_m = 10.0, g_na = 1200.0, g_k = 360.0, g_l = 3.0, v_na = 50.0, v_k = -77.0, v_l = 54.387, dt = 0.1;
    double v_{0.0}, prev_v_{0.0}, i_{0.0}, m_{0.0}, n_{0.0}, h_{0.0};
};
}

extern "C" sanafe::PipelineUnit *create_hodgkin_huxley() { return new HodgkinHuxleySoma(); }
