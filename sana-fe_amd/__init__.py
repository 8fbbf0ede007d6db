"""sanafe_amd -- MI355X-native drop-in for SANA-FE's per-timestep simulation loop.

The directory is called ``sana-fe_amd`` (not importable by name); load it with
``_sanafe_pkg.load()`` at the repo root, which registers it as ``sanafe_amd``.

Front-end: the C++17 / PyBind11 module ``sanafecpp_amd`` provides ``Architecture``, ``Network``,
``load_arch`` and ``load_net`` with the reference's Python names and signatures; ``SpikingChip``
drives libsanafe_host.so / libsanafe_hip.so.  ``description`` / ``yaml_io`` are a pure-Python twin of
the description layer: the tests build their networks with it and cross-check the C++ front-end
against it; ``presets`` restates the reference's architecture files through the API.
"""
from . import description, presets, yaml_io  # noqa: F401
from .description import HardwareMappingError, to_desc  # noqa: F401
from .chip import SpikingChip, BackendMissingError, map_only, cpp  # noqa: F401
from . import chip  # noqa: F401

if cpp is not None:
    Architecture, Network, Tile, Core = cpp.Architecture, cpp.Network, cpp.Tile, cpp.Core
    NeuronGroup, Neuron = cpp.NeuronGroup, cpp.Neuron
    load_arch, load_net = cpp.load_arch, cpp.load_net
else:  # pragma: no cover
    from .description import Architecture, Network, NeuronGroup, Neuron, Tile, Core  # noqa: F401
    from .yaml_io import load_arch, load_net  # noqa: F401
