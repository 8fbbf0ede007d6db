"""Regenerates the input fixtures under tests/golden/ from the reference's DATA files.

Run in the build container (where /root/reference exists):  python tests/golden/make_fixtures.py

* dvs_challenge.npz -- the DVS-gesture weights/thresholds/input biases the reference ships as a
  data file for its own tutorial (sanafe/examples/dvs_challenge.npz; tutorial/tutorial_5_dvs.ipynb
  checks results["neurons_fired"] == 365277 after 1000 steps on it).  Copied byte for byte.
* dvs_yaml.npz -- snn/dvs.yaml (config C2) reduced to arrays: per-group soma parameters, the
  per-neuron input biases, the five hyper-edges (conv2d/dense parameters + weights) and the
  neuron -> core mapping.  tests/nets.py rebuilds the network from these arrays.
"""
import json
import os
import shutil
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _py(v):
    """typed Python value of a raw YAML scalar (int -> float -> bool -> str)."""
    v = str(v)
    for conv in (int, float):
        try:
            return conv(v)
        except ValueError:
            pass
    return {"true": True, "false": False}.get(v, v)


def main():
    shutil.copyfile(os.path.join(REF, "sanafe/examples/dvs_challenge.npz"), os.path.join(HERE, "dvs_challenge.npz"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import _sanafe_pkg
    loader = _sanafe_pkg.load().yaml_io._Loader  # keeps scalars as text: `5.10` must not become the float 5.1
    with open(os.path.join(REF, "snn/dvs.yaml")) as f:
        top = yaml.load(f, Loader=loader)
    net = top["network"]
    out = {}
    meta = {"name": net["name"], "groups": [], "edges": []}
    for g in net["groups"]:
        name = str(g["name"])
        attrs = {k: _py(v) for k, v in g["attributes"].items()}
        n = 0
        bias = {}
        for entry in g["neurons"]:
            for k, v in entry.items():
                k = str(k)
                if ".." in k:
                    a, b = (int(x) for x in k.split(".."))
                else:
                    a = b = int(k)
                n = max(n, b + 1)
                if v and "bias" in v:
                    for i in range(a, b + 1):
                        bias[i] = float(v["bias"])
        meta["groups"].append({"name": name, "count": n, "attributes": attrs})
        if bias:
            arr = np.zeros(n, dtype=np.int64)
            for i, b in bias.items():
                assert float(b) == int(b)
                arr[i] = int(b)
            out["bias_" + name] = arr
    for i, entry in enumerate(net["edges"]):
        for desc, attrs in entry.items():
            a = {k: (v if k == "weight" else _py(v)) for k, v in attrs.items()}
            w = np.asarray([float(x) for x in a.pop("weight")], dtype=np.float64)
            assert (w == np.round(w)).all()
            out["w_%d" % i] = w.astype(np.int16) if np.abs(w).max() < 32768 else w
            meta["edges"].append({"desc": desc, "params": a})
    cores = {}
    order = []
    for m in top["mappings"]:
        for addr, info in m.items():
            fields = {}
            for e in (info if isinstance(info, list) else [info]):
                fields.update(e)
            order.append((str(addr), str(fields["core"])))
    meta["mappings"] = order
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "dvs_yaml.npz"), **out)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    sys.exit(main())
