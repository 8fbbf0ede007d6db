"""numpy emulation of the three device kernels over a lowered image (tests only).

It follows sana-fe_amd/csrc/sanafe_hip.hip statement for statement, so that the mapper and the
lowered-image semantics can be checked against the oracle on a machine without a GPU.  It is NOT
a fallback: nothing in the product imports it."""
import numpy as np


class ImageEmulator:
    def __init__(self, im):
        self.im = im
        n = im["n_slots"]
        self.v = im["slot_v0"].astype(np.float64).copy()
        self.icur = np.zeros(n)
        self.refrac = np.zeros(n, dtype=np.int64)
        self.status = np.zeros(n, dtype=np.uint8)
        self.R = im["ring_slots"]
        self.ring = np.zeros((self.R, n))
        self.valid = np.zeros((self.R, n), dtype=bool)
        self.in_pos = np.zeros(max(1, im["n_input"]), dtype=np.int64)
        self.t = 0
        self.ext_next = 0
        self.last_w = np.zeros(n)           # SANAFE_IN_LAST cores: current of the last event of the previous step
        self.last_set = np.zeros(n, dtype=bool)
        self.arrived = np.zeros(n, dtype=bool)  # SANAFE_IN_GATED / _TAPS: an event reached the neuron in the previous step
        nt = int(im.get("n_taps", 0))
        self.tap_v = np.zeros((nt, 8))
        self.tap_in = np.zeros((nt, 8))
        cls = im["slot_cls"]
        self.model = cls & 7
        self.inkind = (cls >> 3) & 7
        self.ccls = (cls >> 6) & 1023
        self.pcls = cls >> 16
        self.core_of_slot = np.zeros(n, dtype=np.int64)
        for c in range(im["n_cores"]):
            b, k = im["core_nbase"][c], im["core_ncount"][c]
            self.core_of_slot[b:b + ((k + 63) // 64) * 64] = c
        self.slot_core_off = np.arange(n) - im["core_nbase"][self.core_of_slot]

    @staticmethod
    def _cvt(x):
        y = np.where((x > -2147483649.0) & (x < 2147483648.0), np.trunc(x), -2147483648.0)
        return y

    def step(self):
        tot, fired = self.step_neurons()
        return self.step_deliver(tot, fired)

    def step_neurons(self):
        """K1: returns (partial totals, local fired flags per local slot)."""
        im = self.im
        done, t = self.t, self.t + 1
        rs = t % self.R
        n = im["n_slots"]
        status = np.zeros(n, dtype=np.uint8)
        has_in = np.where((self.inkind == 1) | (self.inkind == 2), True, self.valid[rs])  # kinds 0 and 4 read the buffer
        cur = np.where(self.inkind == 1, 0.0, np.where(self.valid[rs], self.ring[rs], 0.0))
        none = self.inkind == 6  # SANAFE_IN_NONE: the soma of a message-pipeline core, called without an input by the neuron loop
        has_in = np.where(none, False, has_in)
        cur = np.where(none, 0.0, cur)
        cur = np.where(self.inkind == 2, np.where(self.last_set, 0.0 + self.last_w, 0.0), cur)
        self.last_set[:] = False
        gated = self.inkind == 3
        has_in = np.where(gated, self.arrived & self.valid[rs], has_in)
        cur = np.where(gated, np.where(self.arrived & self.valid[rs], self.ring[rs], 0.0), cur)
        self.arrived[gated] = False
        buf = ((self.inkind == 0) | (self.inkind == 4) | gated) & self.valid[rs] & (self.model != 0)
        self.ring[rs][buf] = 0.0
        self.valid[rs][buf] = False
        bias = im["slot_bias"]
        sc = im["soma_classes"]
        # external value streams: one row per step (sanafe_hip_write_ext); stepping past the rows is an error
        has_ext = np.zeros(n, dtype=bool)
        ext = np.zeros(n, dtype=np.int64)
        if im.get("n_ext", 0):
            rows = im["ext_rows"]
            if self.ext_next >= len(rows):
                raise RuntimeError("external value streams exhausted")
            has_ext = im["slot_ext"] != 0xffffffff
            ext[has_ext] = rows[self.ext_next][im["slot_ext"][has_ext]]
            self.ext_next += 1

        def col(name):
            return np.array([c[name] for c in sc])[np.minimum(self.pcls, len(sc) - 1)]

        with np.errstate(all="ignore"):
            # ---- LIF ----
            m = self.model == 1
            if m.any():
                v, ic, rc = self.v.copy(), self.icur.copy(), self.refrac.copy()
                st = np.where((np.abs(v) > 0) | has_in | (np.abs(bias) > 0) | (col("force_update") != 0), 2, 1)
                if done > 0:
                    ic = ic * col("input_decay")
                    v = v * col("leak_decay")
                v = self._cvt(v * 64.0) / 64.0
                v = np.where(has_ext, v + ext, v)
                active = ~(rc > 0)
                v2 = v + bias
                ic2 = ic + np.where(has_in, cur, 0.0)
                v2 = v2 + ic2
                th, rth = col("threshold"), col("reverse_threshold")
                fired = v2 > th
                rm, rrm = col("reset_mode"), col("reverse_reset_mode")
                v3 = np.where(fired & (rm == 2), col("reset"), np.where(fired & (rm == 1), v2 - th, v2))
                rc2 = np.where(fired, col("refractory_delay"), rc)
                below = v3 < rth
                v4 = np.where(below & (rrm == 1), v3 - rth,
                              np.where(below & (rrm == 2), col("reverse_reset"), np.where(below & (rrm == 3), rth, v3)))
                v = np.where(active, v4, v)
                ic = np.where(active, ic2, ic)
                rc = np.where(active, rc2, rc)
                st = np.where(active & fired, 3, st)
                rc = np.maximum(rc - 1, 0)
                self.v[m], self.icur[m], self.refrac[m] = v[m], ic[m], rc[m]
                status[m] = st[m]
            # ---- TrueNorth ----
            m = self.model == 2
            if m.any():
                v = self.v.copy()
                st = np.where((np.abs(v) > 0) | has_in | (np.abs(bias) > 0) | (col("force_update") != 0), 2, 1)
                leak, ltz = col("leak_decay"), col("leak_towards_zero") != 0
                v = np.where(ltz, np.where(v > 0, v - leak, np.where(v < 0, v + leak, v)), v + leak)
                v = v + bias
                v = np.where(has_in, v + cur, v)
                th, rth = col("threshold"), col("reverse_threshold")
                rm, rrm = col("reset_mode"), col("reverse_reset_mode")
                vt = np.where(has_ext, v + ext, v)
                fired = vt >= th
                low = (~fired) & (vt <= rth)
                vf = np.where(rm == 2, col("reset"), np.where(rm == 1, v - th, np.where(rm == 3, th, v)))
                vl = np.where(rrm == 2, col("reverse_reset"), np.where(rrm == 1, v + rth, np.where(rrm == 3, rth, v)))
                v = np.where(fired, vf, np.where(low, vl, v))
                st = np.where(fired, 3, st)
                self.v[m] = v[m]
                status[m] = st[m]
        # ---- input ----
        for g in np.nonzero(self.model == 3)[0]:
            a = im["slot_aux"][g]
            pos = self.in_pos[a]
            send = False
            if pos < im["in_train_len"][a]:
                b = int(im["in_train_beg"][a]) + int(pos)
                send = bool((int(im["in_train_bits"][b >> 5]) >> (b & 31)) & 1)
                self.in_pos[a] = pos + 1
            if ext[g] != 0:
                send = True
            period = im["in_rate_period"][a]
            if period > 0 and t % period == 0:
                send = True
            status[g] = 3 if send else 1
        # SANAFE_SOMA_PERSIST (buffer before axon_out): the neuron pipeline holds no unit -- the status the message pipeline
        # left persists (and is what axon_out sends spikes for); not a soma call of the neuron loop
        persist = self.model == 5
        status[persist] = self.status[persist]
        self.status = status
        self.loop_status = status.copy()  # what the step's status log holds: the statuses the neuron loop left
        live = (self.model != 0) & ~persist
        cc = im["cost_classes"]
        se = np.array([c["soma_energy"] for c in cc])[self.ccls]
        sl = np.array([c["soma_latency"] for c in cc])[self.ccls]
        de = np.array([c["dendrite_energy"] for c in cc])[self.ccls]
        dl = np.array([c["dendrite_latency"] for c in cc])[self.ccls]
        idx = np.maximum(status.astype(np.int64) - 1, 0)
        ar = np.arange(n)
        fired = status == 3
        tot = dict(timesteps=1, spikes=int(im["slot_events"][fired].sum()), packets_sent=int(im["slot_packets"][fired].sum()),
                   neurons_updated=int(((status >= 2) & live).sum()), neurons_fired=int((fired & live).sum()),
                   total_hops=int(im["slot_hops"][fired].sum()))
        tot["soma_energy"] = float(se[ar, idx][live].sum())
        tot["dendrite_energy"] = float(de[live].sum() + im["slot_e_dend"][fired].sum())
        tot["synapse_energy"] = float(im["slot_e_syn"][fired].sum())
        tot["network_energy"] = float(im["slot_e_net"][fired].sum())
        tot["total_energy"] = tot["network_energy"] + tot["synapse_energy"] + tot["dendrite_energy"] + tot["soma_energy"]
        lat = np.where(live, (0.0 + dl) + sl[ar, idx], 0.0)
        gen = np.zeros(im["n_cores"])
        np.add.at(gen, self.core_of_slot, lat)
        pk = np.zeros(im["n_cores"])
        np.add.at(pk, self.core_of_slot[fired], im["slot_packets"][fired])
        gen += pk * im["core_axon_out_latency"]
        self._gen = gen
        return tot, fired

    def step_deliver(self, tot, fired_global):
        """K2 + K3 with the GLOBAL fired flags (all ranks' slots concatenated)."""
        im = self.im
        t = self.t + 1
        gen = self._gen
        # ---- delivery ----
        proc = np.zeros(im["n_cores"])
        for s in range(im["n_slices"]):
            c = im["slice_core"][s]
            a0, a1 = int(im["slice_axon_beg"][s]), int(im["slice_axon_end"][s])
            act = np.nonzero(fired_global[im["ax_pre"][a0:a1]])[0] + a0
            if len(act) == 0:
                continue
            proc[c] += im["ax_proc_delay"][act].sum()
            base = int(im["core_syn_base"][c])
            nb = im["core_nbase"][c]
            for a in act:
                s0 = base + int(im["ax_syn_beg"][a])
                for k in range(int(im["ax_nsyn"][a])):
                    meta = int(im["syn_meta"][s0 + k])
                    if (meta >> 19) & 1:
                        continue
                    post, d = meta & 0xffff, (meta >> 16) & 7
                    if self.inkind[nb + post] == 2:  # the buffer keeps the last event only (delivery order = array order)
                        self.last_w[nb + post] = im["syn_weight"][s0 + k]
                        self.last_set[nb + post] = True
                        continue
                    if self.inkind[nb + post] == 4:  # tap d of the neuron's dendrite
                        self.tap_in[im["slot_aux"][nb + post], d] += im["syn_weight"][s0 + k]
                        self.arrived[nb + post] = True
                        continue
                    if self.inkind[nb + post] == 3:
                        d += 1
                        self.arrived[nb + post] = True
                    ws = (t + 1 + d) % self.R
                    self.ring[ws][nb + post] += im["syn_weight"][s0 + k]
                    self.valid[ws][nb + post] = True
        # ---- taps_kernel ----
        for i in range(int(im.get("n_taps", 0))):
            taps, g = int(im["tap_count"][i]), int(im["tap_slot"][i])
            v, tc, sc = self.tap_v[i].copy(), im["tap_tc"][i * 8:i * 8 + 8], im["tap_sc"][i * 8:i * 8 + 8]
            nv = np.zeros(8)
            for k in range(taps):
                nv[k] = v[k] * tc[k]
            for s_ in range(taps):
                if s_ > 0:
                    c_ = v[s_] * sc[s_ - 1]
                    nv[s_ - 1] += c_
                    nv[s_] -= c_
                if s_ + 1 < taps:
                    c_ = v[s_] * sc[s_]
                    nv[s_ + 1] += c_
                    nv[s_] -= c_
            for k in range(taps):
                self.tap_v[i][k] = nv[k] + self.tap_in[i][k]
                self.tap_in[i][k] = 0.0
            if self.arrived[g]:
                self.arrived[g] = False
                ws = (t + 1) % self.R
                self.ring[ws][g] = self.tap_v[i][0]
                self.valid[ws][g] = True
        # ---- msgsoma_kernel: cores whose soma is part of the message pipeline -- one TrueNorth update per synaptic event, in
        #      delivery order, with the running sum of the step's currents (AccumulatorModel::update) ----
        sc = im["soma_classes"]
        for k in range(int(im.get("n_msg_cores", 0))):
            core = int(im["msg_core"][k])
            nb = int(im["core_nbase"][core])
            costs = im["msg_costs"][k]
            ain_l, syn_e, syn_l, dend_e, dend_l = costs[0:5]
            se, sl = costs[5:8], costs[8:11]
            acc, events, fired_updates, msgs = {}, 0, 0, 0
            if k == 0:
                self.msg_fired = np.zeros(len(im["msg_ax_pre"]), dtype=np.uint16)  # per message: updates that fired
            q = int(im["msg_syn_beg"][k])
            for a in range(int(im["msg_ax_beg"][k]), int(im["msg_ax_beg"][k + 1])):
                nsyn = int(im["msg_ax_nsyn"][a])
                if not fired_global[im["msg_ax_pre"][a]]:
                    q += nsyn
                    continue
                msgs += 1
                for j in range(q, q + nsyn):
                    g = nb + int(im["msg_syn_post"][j])
                    p = sc[min(int(self.pcls[g]), len(sc) - 1)]
                    acc[g] = acc.get(g, 0.0) + im["msg_syn_weight"][j]
                    v = self.v[g]
                    if p["leak_towards_zero"]:
                        v = v - p["leak_decay"] if v > 0 else (v + p["leak_decay"] if v < 0 else v)
                    else:
                        v = v + p["leak_decay"]
                    v = v + im["slot_bias"][g]
                    v = v + acc[g]
                    st = 2
                    if v >= p["threshold"]:
                        v = p["reset"] if p["reset_mode"] == 2 else (v - p["threshold"] if p["reset_mode"] == 1 else
                                                                     (p["threshold"] if p["reset_mode"] == 3 else v))
                        st = 3
                    elif v <= p["reverse_threshold"]:
                        rr = p["reverse_reset_mode"]
                        v = p["reverse_reset"] if rr == 2 else (v + p["reverse_threshold"] if rr == 1 else
                                                                (p["reverse_threshold"] if rr == 3 else v))
                    self.v[g] = v
                    self.status[g] = st
                    events += 1
                    fired_updates += st == 3
                    self.msg_fired[a] += st == 3
                q += nsyn
            tot["synapse_energy"] += events * syn_e
            tot["dendrite_energy"] += events * dend_e
            soma = events * (se[0] + se[1]) + fired_updates * se[2]
            tot["soma_energy"] += soma
            tot["total_energy"] += events * syn_e + events * dend_e + soma
            tot["neurons_updated"] += events
            tot["neurons_fired"] += int(fired_updates)
            proc[core] = msgs * ain_l + events * ((syn_l + dend_l) + (sl[0] + sl[1])) + fired_updates * sl[2]
        tot["sim_time"] = float(max(proc.max(), gen.max()) + im["sync_delay"])
        self.t = t
        return tot
