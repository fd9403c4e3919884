"""Who is in the tail of a c2 step?  (round-3 verdict, item 2.)  From the chip-wide phase stamps of the diagnostic build
(profiles/_diag/build_variant.sh stamps):  NAVSIM_LIB=build/libnavsim_stamps.so python profiles/_diag/tail_rescan.py
For each of the last 12 of 40 steps: the launch span, the share of the workgroup-time inside its last 30 % that belongs
to workgroups that took the re-scan branch (phase 5: crash revert / respawn), where those workgroups sat in the launch
order, and whether the workgroup that ends the launch is one of them."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
wl = dict(bench.WORKLOADS[os.environ.get("NAVSIM_WL", "c2")]); wl["field"] = "u16t"
if os.environ.get("NAVSIM_ENVS"):
    wl["envs"] = int(os.environ["NAVSIM_ENVS"])
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
sim.t["scan_noise_std"].fill_(0.02); sim.cfg.add_scan_noise = 1
E = cfg.n_envs
L = lib.load()
buf = torch.zeros((E, 8), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
T = 40
acts = torch.rand((T, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
rows = []
for t in range(T):
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
    if t < T - 12:
        continue
    b = buf.cpu().numpy().astype(np.float64)
    s, e = b[:, 0], b[:, 6]
    t0, t1 = s.min(), e.max()
    span = t1 - t0
    resc = (b[:, 5] - b[:, 4]) > 0.35 * (b[:, 3] - b[:, 2])          # phase 4-5 holds a second scan
    cut = t1 - 0.3 * span
    in_tail = np.clip(e - np.maximum(s, cut), 0, None)               # workgroup-time inside the last 30 %
    life = e - s
    last = int(np.argmax(e))
    pos = np.where(resc)[0]
    rows.append(dict(span_us=span / 100.0, n_rescan=int(resc.sum()), tail_share_rescan=float(in_tail[resc].sum() / in_tail.sum()),
                     tail_occupancy=float(in_tail.sum() / (0.3 * span) / 2048.0),
                     full_occupancy=float((life.sum() - in_tail.sum()) / (0.7 * span) / 2048.0),
                     last_is_rescan=bool(resc[last]), last_slot=last, last_life_us=life[last] / 100.0,
                     median_life_us=float(np.median(life)) / 100.0, rescan_life_us=float(np.median(life[resc])) / 100.0 if resc.any() else 0.0,
                     rescan_slots_quartiles=[int(np.percentile(pos, q)) for q in (0, 25, 50, 75, 100)] if resc.any() else [],
                     # what the launch would last if the slot-time were perfectly packed
                     packed_us=float(life.sum() / 2048.0) / 100.0,
                     phase_us=[round(float(x) / 100.0, 2) for x in np.median(np.diff(b[:, :7], axis=1), axis=0)],
                     # stamp 7 (builds with the job board): end of job_help; how many workgroups helped (> 1 us there) and for how long
                     help_n=int(((b[:, 7] - b[:, 6]) > 100).sum()) if b[:, 7].max() > 0 else 0,
                     help_us_mean=float(np.mean((b[:, 7] - b[:, 6])[(b[:, 7] - b[:, 6]) > 100])) / 100.0 if b[:, 7].max() > 0 and ((b[:, 7] - b[:, 6]) > 100).any() else 0.0,
                     end_us=float((np.maximum(b[:, 7], b[:, 6]).max() - t0) / 100.0),
                     rescan_phase5_us=float(np.median((b[:, 5] - b[:, 4])[resc])) / 100.0 if resc.any() else 0.0))
for r in rows:
    print(r)
if os.environ.get("NAVSIM_SLOWEST"):          # the last step's slowest workgroups, phase by phase (start offset | phases | life)
    order = np.argsort(-(b[:, 6] - b[:, 0]))
    starts = (b[:, 0] - b[:, 0].min()) / 100.0
    print("start offsets (us): median %.2f p90 %.2f max %.2f" % (np.median(starts), np.percentile(starts, 90), starts.max()))
    for w in list(order[:6]) + list(order[len(order) // 2: len(order) // 2 + 2]):
        print("wg %4d start %.2f phases %s life %.2f%s" % (w, starts[w], [round(float(x) / 100.0, 2) for x in np.diff(b[w, :7])],
                                                          (b[w, 6] - b[w, 0]) / 100.0, "  (re-scan)" if resc[w] else ""))
if "counters" in sim.t:
    cn = sim.t["counters"].cpu().numpy()
    print("helper attempts over the %d steps: exhausted at the first look %d, empty descriptor %d, no chunk left at the add %d, chunks marched by helpers %d"
          % (T, cn[6] & 0xFFFFFFFF, cn[6] >> 32, cn[7] & 0xFFFFFFFF, cn[7] >> 32))
print("mean span %.1f us; mean tail share of re-scanning workgroups %.3f; launches ended by one: %d of %d; tail occupancy %.2f, before %.2f"
      % (np.mean([r["span_us"] for r in rows]), np.mean([r["tail_share_rescan"] for r in rows]),
         sum(r["last_is_rescan"] for r in rows), len(rows), np.mean([r["tail_occupancy"] for r in rows]),
         np.mean([r["full_occupancy"] for r in rows])))
