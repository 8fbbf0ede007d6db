// pyconv.hpp -- Python value -> AttrValue with the reference's rules (src/pymodule.cpp:118-175), shared by the
// translation units of the PyBind11 module.
#ifndef SANAFE_HOST_PYCONV_HPP
#define SANAFE_HOST_PYCONV_HPP

#include <pybind11/pybind11.h>

#include "description.hpp"

sanafe_amd::AttrValue py_to_attr(const pybind11::handle &v, bool narrow);
void bind_spiking_chip(pybind11::module_ &m);

#endif
