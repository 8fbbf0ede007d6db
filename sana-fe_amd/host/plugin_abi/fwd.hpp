// fwd.hpp -- forward declarations of the plugin-facing types (plugin ABI of the MI355X host).
#ifndef SANAFE_AMD_PLUGIN_FWD_HPP
#define SANAFE_AMD_PLUGIN_FWD_HPP
namespace sanafe
{
struct ModelAttribute;
struct PipelineResult;
class PipelineUnit;
class SynapseUnit;
class DendriteUnit;
class SomaUnit;
class MappedNeuron;
class MappedConnection;
class Core;
class AxonOutUnit;
struct Timestep;
}
#endif
