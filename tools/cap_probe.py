"""The free-running driver's batch cap: expansions/s of bench.py's timed region (4096 8x8 games x 100 sims, staggered) at several caps.
    python tools/cap_probe.py [steps] [cap ...]       (0 = no cap: every live game's leaf in each batch)"""
import sys, time, json
sys.path.insert(0, ".")
import torch
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.training import SelfPlayEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
caps = [int(a) for a in sys.argv[2:]] or [3640, 0]
n, G, sims = 8, 4096, 100
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f16x2")
for rep in range(2):
    for cap in caps:
        eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, game_id_stride=G, q_mode=_lib.QMODE_F64, refill=True,
                             record_cap=int(G * (steps + 2 + 66) * 1.25), dedup=False, batch_cap=cap)
        eng.stagger(sims)
        eng.run_steps(2 * sims, sync=True); eng.sync()
        net.profile_kernels(reset=True); net.profile(1)
        a = eng.stats(); torch.cuda.synchronize(); t = time.perf_counter()
        eng.run_steps(steps * sims, sync=False); eng.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t; b = eng.stats()
        ms, cnt = net.profile_read(); net.profile(0)
        print(json.dumps({"cap": cap, "expansions_per_s": round((b["expansions"] - a["expansions"]) / dt), "games_per_s": round((b["games_completed"] - a["games_completed"]) / dt, 1),
                          "ms_per_100_batches": round(dt / steps * 1e3, 1), "leaves_per_batch": round((b["leaves_evaluated"] - a["leaves_evaluated"]) / (steps * sims), 1),
                          "conv3_tile": net.conv3_tile_rows()}), flush=True)
        del eng
