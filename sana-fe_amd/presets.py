"""Programmatic architecture presets.

``/root/reference`` does not travel to the GPU box, so the architectures the
benchmark configs name are restated here through the description API.  The
numeric constants are the ones the reference's architecture descriptions give
(cited per preset); tests/test_presets.py checks each preset against the
reference YAML whenever the reference is present.
"""


def _api(api):
    """The description API to build with: the C++/PyBind11 module by default (tests pass their Python twin)."""
    if api is not None:
        return api
    from .chip import cpp
    if cpp is None:
        raise ImportError("sanafecpp_amd is missing: build it with `make -C sana-fe_amd`")
    return cpp

# arch/loihi.yaml:19-27, arch/loihi_large.yaml:12-20 -- Loihi hop costs (Davies et al. 2018)
_LOIHI_TILE = dict(energy_north_hop=4.2e-12, latency_north_hop=6.5e-9, energy_east_hop=3.0e-12, latency_east_hop=4.1e-9,
                   energy_south_hop=4.2e-12, latency_south_hop=6.5e-9, energy_west_hop=3.0e-12, latency_west_hop=4.1e-9)
_LOIHI_SYNC = {1: 0.6e-6, 2: 1.0e-6, 4: 1.4e-6, 29: 1.8e-6}  # arch/loihi.yaml:16


def _f(**kw):
    return {k: float(v) for k, v in kw.items()}


def _loihi_core(arch, tile_id, idx, buffer_position, buffer_inside_unit, shared, n_inputs=1024, noise_file=None,
                noise_bits=None):
    core = arch.create_core("loihi_core[%d]" % idx, tile_id, buffer_position, buffer_inside_unit, 1024,
                            share_units_with=shared.get("c"))
    if "c" in shared:
        return core
    shared["c"] = core
    core.create_axon_in("loihi_in", 0.0e-12, 16.0e-9)
    # unit sections are always visited synapse, dendrite, soma (src/yaml_arch.cpp:260-266);
    # inside a section the file order is kept, which differs between loihi.yaml and loihi_large.yaml
    core.create_synapse("loihi_dense_synapse", "current_based", _f(energy_process_spike=35.5e-12, latency_process_spike=3.8e-9))
    core.create_synapse("loihi_sparse_synapse", "current_based", _f(energy_process_spike=33.6e-12, latency_process_spike=4.7e-9))
    core.create_synapse("loihi_conv_synapse", "current_based", _f(latency_process_spike=3.1e-9, energy_process_spike=24.0e-12))
    core.create_dendrite("loihi_dendrites", "accumulator", _f(energy_update=0.0, latency_update=0.0))
    core.create_dendrite("loihi_dendrites_delay", "accumulator_with_delay", _f(energy_update=0.0, latency_update=0.0))
    core.create_soma("loihi_lif", "leaky_integrate_fire",
                     _f(energy_access_neuron=51.2e-12, latency_access_neuron=6.0e-9, energy_update_neuron=21.6e-12,
                        latency_update_neuron=3.7e-9, energy_spike_out=69.3e-12, latency_spike_out=30.0e-9))
    if noise_file is not None:
        # arch/loihi_with_noise.yaml:46-61 (its `noise:` path is private to the authors' machine; pass your own)
        at = _f(energy_access_neuron=51.2e-12, latency_access_neuron=6.0e-9, energy_update_neuron=27.6e-12,
                latency_update_neuron=3.7e-9, energy_spike_out=69.3e-12, latency_spike_out=30.0e-9)
        at["noise"] = str(noise_file)
        if noise_bits is not None:
            at["noise_bits"] = int(noise_bits)
        core.create_soma("loihi_stochastic_lif", "leaky_integrate_fire", at)
    zero = _f(energy_access_neuron=0.0, latency_access_neuron=0.0, energy_update_neuron=0.0, latency_update_neuron=0.0,
              energy_spike_out=0.0, latency_spike_out=0.0)
    for i in range(n_inputs):
        core.create_soma("loihi_inputs[%d]" % i, "input", zero)
    core.create_axon_out("loihi_out", 111.0e-12, 5.1e-9)
    return core


def loihi(n_inputs=1024, api=None, noise_file=None, noise_bits=None):
    """arch/loihi.yaml: 8x4 mesh, 32 tiles x 4 cores, buffer before soma.  With ``noise_file`` the cores also get the
    `loihi_stochastic_lif` soma of arch/loihi_with_noise.yaml reading that file."""
    arch = _api(api).Architecture("loihi_chip", 8, 4, 16, _LOIHI_SYNC)
    shared = {}
    for t in range(32):
        tile = arch.create_tile("loihi_tile[%d]" % t, **_LOIHI_TILE)
        for c in range(4):
            _loihi_core(arch, tile.id, c, "soma", False, shared, n_inputs, noise_file, noise_bits)
    return arch


def loihi_large(n_tiles=1024, n_inputs=1024, width=256, height=128, api=None, buffer_inside_unit=True):
    """arch/loihi_large.yaml: 256x128 mesh, 1024 tiles x 4 cores, buffer inside the dendrite unit
    (``buffer_inside_unit=False``: the same chip with the time-step buffer BEFORE the dendrite unit)."""
    arch = _api(api).Architecture("loihi_chip", width, height, 16, _LOIHI_SYNC)
    shared = {}
    for t in range(n_tiles):
        tile = arch.create_tile("loihi_tile[%d]" % t, **_LOIHI_TILE)
        for c in range(4):
            _loihi_core(arch, tile.id, c, "dendrite", buffer_inside_unit, shared, n_inputs)
    return arch


def truenorth(n_tiles=4096, width=64, height=64, api=None):
    """arch/truenorth.yaml: 64x64 mesh, one 256-neuron core per tile, all costs zero."""
    arch = _api(api).Architecture("truenorth_chip", width, height, 1, {0: 0.0})
    first = None
    for t in range(n_tiles):
        tile = arch.create_tile("truenorth_tile[%d]" % t)
        core = arch.create_core("truenorth_core[0]", tile.id, "soma", False, 256, share_units_with=first)
        if first is not None:
            continue
        first = core
        core.create_axon_in("core_in", 0.0, 0.0)
        core.create_synapse("core_synapses", "current_based", _f(energy_process_spike=0.0, latency_process_spike=0.0))
        core.create_dendrite("core_dendrites", "accumulator", _f(energy_update=0.0, latency_update=0.0))
        core.create_soma("core_soma", "truenorth",
                         _f(energy_access_neuron=0.0, latency_access_neuron=0.0, energy_update_neuron=0.0,
                            latency_update_neuron=0.0, energy_spike_out=0.0, latency_spike_out=0.0))
        core.create_axon_out("core_out", 0.0, 0.0)
    return arch


def example_chip(api=None):
    """arch/example_chip.yaml: 2 tiles x 4 cores demo chip."""
    arch = _api(api).Architecture("demo", 2, 1, 4, {0: 0.0})
    first = None
    for t in range(2):
        tile = arch.create_tile("demo_tile[%d]" % t, energy_north_hop=2.0e-12, latency_north_hop=1.4e-9,
                                energy_east_hop=2.5e-12, latency_east_hop=1.2e-9, energy_south_hop=2.0e-12,
                                latency_south_hop=1.5e-9, energy_west_hop=1.8e-12, latency_west_hop=2.0e-9)
        for c in range(4):
            core = arch.create_core("demo_core[%d]" % c, tile.id, "soma", False, 100, share_units_with=first)
            if first is not None:
                continue
            first = core
            core.create_axon_in("demo_in", 0.0, 0.0)
            core.create_synapse("demo_synapse", "current_based", _f(energy_process_spike=20.0e-12, latency_process_spike=3.0e-9))
            core.create_dendrite("demo_dendrite", "accumulator", _f(energy_update=0.0, latency_update=0.0),
                                 update_every_timestep=True)
            core.create_soma("demo_soma_default", "leaky_integrate_fire",
                             _f(energy_access_neuron=20.0e-12, latency_access_neuron=3.0e-9, energy_update_neuron=10.0e-12,
                                latency_update_neuron=1.0e-9, energy_spike_out=60.0e-12, latency_spike_out=30.0e-9))
            core.create_soma("demo_soma_alt", "leaky_integrate_fire",
                             _f(energy_access_neuron=50.0e-12, latency_access_neuron=5.0e-9, energy_update_neuron=60.0e-12,
                                latency_update_neuron=10.0e-9, energy_spike_out=30.0e-12, latency_spike_out=3.0e-9))
            core.create_soma("demo_input", "input",
                             _f(energy_access_neuron=0.0, latency_access_neuron=0.0, energy_update_neuron=0.0,
                                latency_update_neuron=0.0, energy_spike_out=0.0, latency_spike_out=0.0))
            core.create_axon_out("demo_out", 100.0e-12, 5.0e-9)
    return arch


def loihi_with_plugin_somas(k, plugin_path, model="hodgkin_huxley", n_inputs=4, api=None):
    """Config C5: the Loihi architecture plus `k` plugin soma units `hh[0..k-1]` on every core
    (one neuron per unit instance, like plugins/hodgkin_huxley.cpp), with the soma default costs the
    plugin does not simulate itself (src/pipeline.hpp:698-713)."""
    arch = loihi(n_inputs=n_inputs, api=api)
    core = arch.cores()[0]
    costs = _f(energy_access_neuron=51.2e-12, latency_access_neuron=6.0e-9, energy_update_neuron=21.6e-12,
               latency_update_neuron=3.7e-9, energy_spike_out=69.3e-12, latency_spike_out=30.0e-9)
    for i in range(k):
        core.create_soma("hh[%d]" % i, model, dict(costs), plugin=plugin_path)
    return arch
