"""Pure-Python twin of the product's description layer (C++: sana-fe_amd/host/description.cpp, yaml_subset.cpp).

TEST INFRASTRUCTURE: the tests build most of their networks with it and cross-check the C++ front-end against it
(PyYAML reader, typed setters).  The product never imports it; `_sanafe_pkg.load()` attaches it to the loaded package
as `S.description` / `S.yaml_io` / `S.to_desc` for the tests and for bench.py's small configurations."""
