"""sanafe_amd -- MI355X-native drop-in for SANA-FE's per-timestep simulation loop.

The directory is called ``sana-fe_amd`` (not importable by name); load it with
``_sanafe_pkg.load()`` at the repo root, which registers it as ``sanafe_amd``.
"""
from .description import (Architecture, Network, NeuronGroup, Neuron, Tile, Core, HardwareMappingError,  # noqa: F401
                          to_desc)
from .yaml_io import load_arch, load_net  # noqa: F401
from . import description, presets, yaml_io  # noqa: F401
from .chip import SpikingChip, BackendMissingError, map_only  # noqa: F401,E402
from . import chip  # noqa: F401,E402
