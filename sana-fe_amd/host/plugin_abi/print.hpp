// print.hpp -- logging macros a plugin may use (same names and arity as the reference's
// src/print.hpp:28-108).  INFO prints when ENABLE_DEBUG_PRINTS is defined; TRACEn(category, ...)
// additionally needs DEBUG_LEVEL_<category> >= n.  Both compile to nothing by default.
#ifndef SANAFE_AMD_PLUGIN_PRINT_HPP
#define SANAFE_AMD_PLUGIN_PRINT_HPP
#include <cstdio>

#define SANAFE_AMD_DEFINE_LEVEL(cat) 0
#ifndef DEBUG_LEVEL_ARCH
#define DEBUG_LEVEL_ARCH 0
#endif
#ifndef DEBUG_LEVEL_NET
#define DEBUG_LEVEL_NET 0
#endif
#ifndef DEBUG_LEVEL_PYMODULE
#define DEBUG_LEVEL_PYMODULE 0
#endif
#ifndef DEBUG_LEVEL_DESCRIPTION
#define DEBUG_LEVEL_DESCRIPTION 0
#endif
#ifndef DEBUG_LEVEL_MODELS
#define DEBUG_LEVEL_MODELS 0
#endif
#ifndef DEBUG_LEVEL_PLUGINS
#define DEBUG_LEVEL_PLUGINS 0
#endif
#ifndef DEBUG_LEVEL_SCHEDULER
#define DEBUG_LEVEL_SCHEDULER 0
#endif
#ifndef DEBUG_LEVEL_CHIP
#define DEBUG_LEVEL_CHIP 0
#endif

#ifdef ENABLE_DEBUG_PRINTS
#define INFO(...) std::fprintf(stdout, __VA_ARGS__)
#define SANAFE_AMD_TRACE(n, category, ...)                              \
    do                                                                  \
    {                                                                   \
        if (DEBUG_LEVEL_##category >= (n)) std::fprintf(stdout, __VA_ARGS__); \
    } while (0)
#else
#define INFO(...) \
    do            \
    {             \
    } while (0)
#define SANAFE_AMD_TRACE(n, category, ...) \
    do                                     \
    {                                      \
    } while (0)
#endif
#define TRACE1(category, ...) SANAFE_AMD_TRACE(1, category, __VA_ARGS__)
#define TRACE2(category, ...) SANAFE_AMD_TRACE(2, category, __VA_ARGS__)
#define TRACE3(category, ...) SANAFE_AMD_TRACE(3, category, __VA_ARGS__)
#endif
