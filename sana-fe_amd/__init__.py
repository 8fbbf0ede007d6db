"""sanafe_amd -- MI355X-native drop-in for SANA-FE's per-timestep simulation loop.

The directory is called ``sana-fe_amd`` (not importable by name); load it with
``_sanafe_pkg.load()`` at the repo root, which registers it as ``sanafe_amd``.

Everything on the path is compiled: the C++17 / PyBind11 module ``sanafecpp_amd`` provides ``Architecture``,
``Network``, ``load_arch``, ``load_net`` and ``SpikingChip`` with the reference's Python names and signatures over
libsanafe_host.so / libsanafe_hip.so.  ``chip.SpikingChip`` is that class plus the diagnostics the tests and
bench.py use; ``presets`` restates the reference's architecture files through the API.
"""
from . import presets  # noqa: F401
from .chip import SpikingChip, BackendMissingError, HardwareMappingError, map_only, cpp  # noqa: F401
from . import chip  # noqa: F401

if cpp is not None:
    Architecture, Network, Tile, Core = cpp.Architecture, cpp.Network, cpp.Tile, cpp.Core
    NeuronGroup, Neuron = cpp.NeuronGroup, cpp.Neuron
    load_arch, load_net = cpp.load_arch, cpp.load_net
