"""Delivery-launch duration against the network's activity, step by step (C3 recipe): profiles/step_profile.py [steps]
Prints, per block of 2 steps: spikes fired, synaptic events, average delivery / neuron launch (HIP events)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _sanafe_pkg  # noqa: E402
import bench  # noqa: E402

S = _sanafe_pkg.load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
arch, net = bench.build_workload(S, 1, 512, 512, 2621, 0.1, 1)
chip = S.SpikingChip(arch)
chip.load(net)
H = S.chip.hip_lib()
dev = chip.device_handle()
for rep in range(2):  # the second pass (after reset) separates the network's transient from the clocks' ramp
  if rep:
      chip.reset()
      print("---- reset, again ----")
  H.sanafe_hip_set_timing(dev, 1)
  prev = chip.read_totals()
  for t in range(0, steps, 2):
      if H.sanafe_hip_step(dev, 2, 1, 0) != 0:
          raise RuntimeError(H.sanafe_hip_last_error().decode())
      cur = chip.read_totals()
      nm, dm, rm, ln = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
      H.sanafe_hip_read_timing(dev, C.byref(nm), C.byref(dm), C.byref(rm), C.byref(ln))
      H.sanafe_hip_set_timing(dev, 1)  # restart the averages
      print("steps %3d-%3d  fired/step %7.0f  events/step %.4g  deliver %.1f us  neuron %.1f us" % (
          t + 1, t + 2, (cur["neurons_fired"] - prev["neurons_fired"]) / 2, (cur["spikes"] - prev["spikes"]) / 2,
          1e3 * dm.value, 1e3 * nm.value), flush=True)
      prev = cur
