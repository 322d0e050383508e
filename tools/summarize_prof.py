"""Turn rocprofv3 output directories into the summaries kept under profiles/.

  python tools/summarize_prof.py trace  <rocprof_dir> <out_csv>  "<header comment>" [leaves_per_launch] [last_n]
  python tools/summarize_prof.py pmc    <out_csv> "<header comment>" [last=N] <rocprof_dir> [<rocprof_dir> ...]

last_n / last=N: the window of the run that is bench.py's TIMED region -- the last N launches of every (kernel, grid) in
dispatch order (N = steps x sims; run bench.py with --no-compare so that nothing is launched after the timed region).
The untimed stagger / warm-up launches before it run on partly filled batches; the `trace` summary lists both windows.
  python tools/summarize_prof.py traffic <pmc_by_shape_csv> <out_json> <precision> <kernel substring> <grid_threads> [layer=conv2] [leaves_per_launch]

`trace`   : per (kernel, grid) averages from *_kernel_trace.csv; the OthelloNN layers are recognised by grid size
            (launched for 4096 slots) and get their algorithmic fp32 TFLOP/s for `leaves_per_launch` positions actually
            evaluated per launch -- take it from the bench line of the same run (leaves_evaluated_rank0 / roofline.launches;
            ~3760 in whole-game self-play, because ~8 % of the simulations end on finished boards); default 4096 = full batches.
`pmc`     : per (kernel, grid, counter) per-launch averages from one or more *_counter_collection.csv (separate passes).
  python tools/summarize_prof.py tree   <pmc_by_shape_csv> <out_json> <simulations_per_launch>
`tree`    : FETCH_SIZE / WRITE_SIZE of k_select / k_expand_backup / k_compact per simulation -> profiles/tree_traffic.json
`traffic` : conv2's HBM bytes per launch -> the json bench.py reads for roofline.traffic
            (gfx950: FETCH_SIZE is in KB and counts half of wide coalesced reads -> x2; WRITE_SIZE in KB).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

LEAVES = 4096
# layer -> (fp32 FLOP per leaf); K, N of the implicit GEMM x output pixels per leaf x 2
FLOP = {"conv2": 2 * 64 * 4608 * 512, "conv3": 2 * 36 * 4608 * 512, "conv4": 2 * 16 * 4608 * 512,
        "fc1": 2 * 8192 * 1024, "fc2": 2 * 1024 * 512}


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]


def trace(d, out, header, leaves=LEAVES, last_n=0):
    rows = defaultdict(list)
    with open(find(d, "_kernel_trace.csv")) as f:
        for r in sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"])):
            k = short(r["Kernel_Name"])
            if k.startswith("__amd") or "at::" in k or "elementwise" in k:
                continue
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            rows[(k, grid, wg, int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]),
                  int(r["Scratch_Size"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # name the GEMM launches.  precision f16x2 (k_gemm_h2): by tile configuration and grid at 4096 leaves per launch;
    # precision f32: the gemm shapes ordered by total time (conv2 > conv3 > conv4 > fc1 > fc2 holds for this net)
    names = {}
    h2 = {("H2BigPPLut", 1048576): "conv2", ("H2BigPP>", 1048576): "conv2", ("H2MidPP", 786432): "conv3", ("H2BigPP>", 262144): "conv4",
          ("H2BigPP>", 131072): "fc1", ("H2Cfg<1, 2, 2, 2>", 32768): "fc2",
          # precision f32: k_gemm_f32, 128 x 128 tiles, 256 threads
          ("k_gemm_f32", 2097152): "conv2", ("k_gemm_f32", 1179648): "conv3", ("k_gemm_f32", 524288): "conv4",
          ("k_gemm_f32", 65536): "fc1", ("k_gemm_f32", 32768): "fc2"}
    for k in rows:
        for (sub, grid), lay in h2.items():
            if "k_gemm" in k[0] and sub in k[0] and k[1] == grid and len(rows[k]) >= 50:
                names[k] = lay
    if not names:
        gemm = sorted((k for k in rows if "k_gemm" in k[0] and len(rows[k]) >= 50), key=lambda k: -sum(rows[k]))
        for k, lay in zip(gemm, ("conv2", "conv3", "conv4", "fc1", "fc2")):
            names[k] = lay
    with open(out, "w") as f:
        f.write(f"# {header}\n")
        f.write(f"# per (kernel, grid) averages from the kernel trace; {leaves:g} leaves evaluated per launch; TFLOP_per_s = ALGORITHMIC fp32 FLOP "
                "(precision f16x2 executes 3x that on the matrix pipe)\n")
        if last_n:
            f.write(f"# timed_* columns: the last {last_n} launches of the kernel in dispatch order = bench.py's timed region (full batches); "
                    "calls / avg_us / total_ms: every launch of the run incl. the untimed stagger and warm-up rounds (partly filled batches) -- "
                    "the population rocprofv3's own --stats file averages over\n")
        f.write("kernel,grid_threads,wg,lds_bytes,vgpr,agpr,scratch,calls,avg_us,min_us,max_us,total_ms,timed_calls,timed_avg_us,layer,algorithmic_TFLOP_per_s\n")
        for k in sorted(rows, key=lambda k: -sum(rows[k])):
            t = rows[k]
            avg = sum(t) / len(t)
            lay = names.get(k, "")
            tw = t[-last_n:] if last_n and len(t) >= last_n else []
            tavg = sum(tw) / len(tw) if tw else avg
            tf = f"{FLOP[lay] * leaves / (tavg * 1e-9) / 1e12:.1f}" if lay else ""
            f.write(f"\"{k[0]}\",{k[1]},{k[2]},{k[3]},{k[4]},{k[5]},{k[6]},{len(t)},{avg / 1e3:.1f},{min(t) / 1e3:.1f},"
                    f"{max(t) / 1e3:.1f},{sum(t) / 1e6:.1f},{len(tw) if tw else ''},{tavg / 1e3 if tw else 0:.1f},{lay},{tf}\n")


def pmc(out, header, dirs, last_n=0):
    acc = defaultdict(lambda: [0.0, 0])
    for d in dirs:
        per = defaultdict(list)
        with open(find(d, "_counter_collection.csv")) as f:
            for r in sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"])):
                k = short(r["Kernel_Name"])
                if k.startswith("__amd") or "at::" in k or "elementwise" in k:
                    continue
                per[(k, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for key, vals in per.items():
            if last_n and len(vals) >= last_n:
                vals = vals[-last_n:]                       # the timed region's launches only
            acc[key][0] += sum(vals)
            acc[key][1] += len(vals)
    with open(out, "w") as f:
        f.write(f"# {header}\n")
        f.write("# per-launch averages; FETCH_SIZE/WRITE_SIZE in KB as reported (gfx950: FETCH_SIZE counts 1/2 of wide coalesced "
                "reads -> x2 when converted to bytes)\n")
        f.write("kernel,grid_threads,counter,avg_per_launch,launches\n")
        for (k, g, c), (s, n) in sorted(acc.items()):
            f.write(f"\"{k}\",{g},{c},{s / n:.6g},{n}\n")


def tree(src, out, games):
    """HBM-side bytes per simulation of the tree kernels from the pmc-by-shape summary of ONE timed move round
    (`games` simulations per launch).  FETCH_SIZE is reported raw and x2 (the gfx950 correction the guide prescribes for wide
    16-byte-per-lane reads -- the node records are read that way; the 4-byte table window and the scattered 8/16-byte reads
    are uncalibrated, so x2 is an upper bound)."""
    vals = defaultdict(dict)
    with open(src) as f:
        for r in csv.DictReader(l for l in f if not l.startswith("#")):
            vals[r["kernel"]][r["counter"]] = float(r["avg_per_launch"])
    j = {"round": 2, "simulations_per_launch": float(games), "algorithmic_bytes_per_sim": 1300, "source": src, "kernels": {}}
    tot_raw = tot_w = 0.0
    # (k_backup_select = expand + backup of the previous simulation fused with the descent: 99 of the 100 launches of a round;
    #  k_select / k_expand_backup = the first descent and the closing backup of a round, one launch each)
    # free-running driver (bench.py's default): k_backup_advance = expand + backup of the previous batch fused with every game's advance
    # (99 of the 100 launches of a step), k_advance / k_expand_backup = the first advance and the closing backup of a call; `games` = the
    # simulations a launch completes on average (passed by the caller: simulations of the step / launches)
    for k in ("k_backup_advance", "k_advance", "k_backup_select", "k_select", "k_expand_backup", "k_compact"):
        if k not in vals or "FETCH_SIZE" not in vals[k]:
            continue
        fr, wr = vals[k]["FETCH_SIZE"] * 1024 / float(games), vals[k]["WRITE_SIZE"] * 1024 / float(games)
        j["kernels"][k] = {"fetch_bytes_per_sim_raw": fr, "fetch_bytes_per_sim_x2": 2 * fr, "write_bytes_per_sim": wr}
        if k in ("k_backup_advance", "k_backup_select", "k_compact"):               # the per-batch kernels of a steady step
            tot_raw += fr; tot_w += wr
    j["tree_side_bytes_per_sim"] = {"fetch_raw": tot_raw, "fetch_x2": 2 * tot_raw, "write": tot_w, "total_with_x2_fetch": 2 * tot_raw + tot_w}
    with open(out, "w") as f:
        json.dump(j, f, indent=1)
    print(json.dumps(j))


def traffic(src, out, precision, kernel_sub, grid, layer="conv2", leaves=LEAVES):
    leaves = float(leaves)
    vals = {}
    with open(src) as f:
        for r in csv.DictReader(l for l in f if not l.startswith("#")):
            if kernel_sub in r["kernel"] and int(r["grid_threads"]) == int(grid):
                vals[r["counter"]] = float(r["avg_per_launch"])
    fetch = vals["FETCH_SIZE"] * 1024 * 2
    write = vals["WRITE_SIZE"] * 1024
    # algorithmic bytes per leaf (h2 activations are 4 B per value, like fp32): input pixels + output pixels + the layer's weights once per launch
    alg = {"conv2": 64 * 2048 + 64 * 2048 + 9 * 512 * 512 * 4 / leaves, "conv3": 64 * 2048 + 36 * 2048 + 9 * 512 * 512 * 4 / leaves}[layer]
    if precision == "f32" and layer == "conv2":
        alg = 262144.0 + 18 * 512 * 512 * 4 / LEAVES           # as first committed: the transposed copy of the kernel counted too
    j = {"round": 2, "precision": precision, "kernel": f"{kernel_sub} {layer}", "leaves_per_launch": leaves,
         "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
         "hbm_bytes_per_leaf": (fetch + write) / leaves,
         "algorithmic_bytes_per_leaf": alg,
         "source": src}
    if "GRBM_GUI_ACTIVE" in vals:                  # summed over the 8 XCDs
        j["gpu_cycles_per_launch"] = vals["GRBM_GUI_ACTIVE"] / 8
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:     # summed over 256 CUs x 4 SIMDs
            j["mfma_busy_frac"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_LDS_BANK_CONFLICT" in vals:
        j["lds_bank_conflict_cycles"] = vals["SQ_LDS_BANK_CONFLICT"]
    with open(out, "w") as f:
        json.dump(j, f, indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "trace":
        trace(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]) if len(sys.argv) > 5 else LEAVES,
              int(sys.argv[6]) if len(sys.argv) > 6 else 0)
    elif mode == "pmc":
        rest = sys.argv[4:]
        last = int(rest.pop(0).split("=")[1]) if rest and rest[0].startswith("last=") else 0
        pmc(sys.argv[2], sys.argv[3], rest, last)
    elif mode == "traffic":
        traffic(*sys.argv[2:9])
    elif mode == "tree":
        tree(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
