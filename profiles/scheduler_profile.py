"""Host `detailed` NoC scheduler (src/schedule.cpp:208-620 restated in host/chip.cpp): messages/s per thread and the C2
end-to-end rate at 0 (in line) / 1 / 8 / 16 scheduler threads.  Host work only -- no GPU credit is claimed for it.

    python3 profiles/scheduler_profile.py > profiles/r03_scheduler.json       (on the GPU box)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import ctypes as C  # noqa: E402
import _sanafe_pkg  # noqa: E402
import bench  # noqa: E402
import nets  # noqa: E402

S = _sanafe_pkg.load()
out = {"host": bench.host_info(), "c2": {}, "schedule_only": {}}
arch, net = nets.dvs_yaml(S)
for threads in (0, 1, 8, 16):
    chip = S.SpikingChip(arch)
    chip.load(net)
    L = S.chip.lib()
    L.sanafe_chip_set_scheduler_threads(chip._h, threads)
    chip.run(50, "detailed")
    t0 = time.perf_counter()
    r = chip.run(3000, "detailed")
    dt = time.perf_counter() - t0
    out["c2"][str(threads)] = {"timesteps_per_s": 3000 / dt, "messages_per_s": r["packets_sent"] / dt,
                               "messages_per_step": r["packets_sent"] / 3000}
    del chip
# the scheduler alone, one thread: rebuild + schedule the messages of one busy step over and over
chip = S.SpikingChip(arch)
chip.load(net)
chip.run(200, "simple")
status = np.ascontiguousarray(chip.status())
info = chip.info()
# statuses in slot order for sanafe_test_schedule
slot_map = np.zeros(info["n_neurons"], dtype=np.uint32)
L = S.chip.lib()
L.sanafe_chip_get_slot_map.argtypes = [C.c_void_p, C.c_void_p]
L.sanafe_chip_get_slot_map(chip._h, slot_map.ctypes.data)
by_slot = np.zeros(info["n_slots"], dtype=np.uint8)
by_slot[slot_map] = status
sim_time, n_msgs, t_build, t_sched = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
L.sanafe_test_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
reps = 200
L.sanafe_test_schedule(chip._h, by_slot.ctypes.data, reps, C.byref(sim_time), C.byref(n_msgs), C.byref(t_build), C.byref(t_sched))
out["schedule_only"] = {"messages_per_step": n_msgs.value, "reps": reps,
                        "build_ns_per_message": 1e9 * t_build.value / max(1, reps * n_msgs.value),
                        "schedule_ns_per_message": 1e9 * t_sched.value / max(1, reps * n_msgs.value),
                        "messages_per_s_one_thread": reps * n_msgs.value / max(1e-9, t_build.value + t_sched.value)}
print(json.dumps(out, indent=1))
