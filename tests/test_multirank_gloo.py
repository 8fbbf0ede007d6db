"""N > 1 path on CPU: two processes (gloo) each lower their tile shard (`n_ranks=2`), run the numpy
emulation of the device kernels on it and exchange spike bitmaps with an all-gather, exactly like
bench.py does with RCCL.  The aggregated result must equal the oracle's single-chip run."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 12
# (arch kind, synaptic delays, dendrite unit): delay line inside the dendrite unit; last-event buffer before the
# dendrite unit; gated delay line behind the buffer of arch/loihi.yaml
CONFIGS = {"inside_delay": ("large", True, None), "before_dendrite": ("before_dendrite", False, None),
           "gated_delay": ("loihi", True, "loihi_dendrites_delay")}


def _network(S, nets, config):
    kind, delays, dendrite = CONFIGS[config]
    return nets.random_loihi(S, n_tiles=4, neurons_per_core=64, out_degree=20, arch_kind=kind, delays=delays, seed=9,
                             dendrite=dendrite)


def _worker(rank, world, port, out, config):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import _sanafe_pkg
    import nets
    from image_emulator import ImageEmulator
    S = _sanafe_pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arch, net = _network(S, nets, config)
    im, slot_of = S.map_only(arch, net, n_ranks=world, rank=rank)
    emu = ImageEmulator(im)
    n_local, n_global, off = im["n_slots"], im["n_global_slots"], im["slot_offset"]
    assert n_global == n_local * world, "equal shards expected"
    rows = []
    for t in range(STEPS):
        tot, fired = emu.step_neurons()
        local = torch.from_numpy(fired.astype(np.uint8))
        glob = torch.zeros(n_global, dtype=torch.uint8)
        dist.all_gather_into_tensor(glob, local)
        tot = emu.step_deliver(tot, glob.numpy().astype(bool))
        vec = torch.tensor([tot[k] for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired", "total_hops",
                                             "total_energy")], dtype=torch.float64)
        dist.all_reduce(vec)
        st = torch.tensor([tot["sim_time"]], dtype=torch.float64)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        rows.append(vec.tolist() + st.tolist())
    # potentials of the local neurons, by global neuron id
    mine = (slot_of >= off) & (slot_of < off + n_local)
    v = np.full(len(slot_of), np.nan)
    v[mine] = emu.v[slot_of[mine] - off]
    if rank == 0:
        np.save(out + ".rows.npy", np.array(rows))
    np.save(out + ".v%d.npy" % rank, v)
    dist.destroy_process_group()


@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_two_ranks_match_oracle(S, tmp_path, config):
    import nets
    from oracle.oracle import OracleChip
    out = str(tmp_path / "mr")
    port = 29500 + (os.getpid() % 2000) + 7 * sorted(CONFIGS).index(config)
    mp.spawn(_worker, args=(2, port, out, config), nprocs=2, join=True)
    rows = np.load(out + ".rows.npy")
    v = np.where(np.isnan(np.load(out + ".v0.npy")), np.load(out + ".v1.npy"), np.load(out + ".v0.npy"))
    arch, net = _network(S, nets, config)
    orc = OracleChip(S.to_desc(arch, net))
    for t in range(STEPS):
        b = orc.step("simple")
        got = rows[t]
        assert got[:5].tolist() == [b["spike_count"], b["packets_sent"], b["neurons_updated"], b["neurons_fired"], b["total_hops"]], t
        assert got[5] == pytest.approx(b["total_energy"], rel=1e-9)
        assert got[6] == pytest.approx(b["sim_time"], rel=1e-9)
    assert np.array_equal(v, orc.potentials())
