"""Turn rocprofv3 output directories into the summaries kept under profiles/.

  python tools/summarize_prof.py trace  <rocprof_dir> <out_csv>  "<header comment>" [leaves_per_launch] [last_n]
  python tools/summarize_prof.py pmc    <out_csv> "<header comment>" [last=N] <rocprof_dir> [<rocprof_dir> ...]

last_n / last=N: the window of the run that is bench.py's TIMED region -- the last N launches of every (kernel, grid) in
dispatch order (N = steps x sims; run bench.py with --no-compare so that nothing is launched after the timed region).
The untimed stagger / warm-up launches before it run on partly filled batches; the `trace` summary lists both windows.
  python tools/summarize_prof.py traffic <pmc_by_shape_csv> <out_json> <precision> <kernel substring> <grid_threads> [layer=conv2] [leaves_per_launch]

`trace`   : per (kernel, grid, layer) averages from *_kernel_trace.csv; the OthelloNN layers are recognised by the ORDER of the launches
            inside a forward (k_lut_ids / k_conv1* starts one) and -- in the timed window -- get their algorithmic fp32 TFLOP/s and
            fraction of the matrix peak for `leaves_per_launch` positions per launch: take it from the bench line of the same run
            (leaves_evaluated_rank0 / roofline.launches; 3640 under bench.py's batch cap); default 4096 = full batches.
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


PEAK = {"k_gemm_h2": 2500.0, "k_gemm_b3": 2500.0, "k_gemm_f32": 157.3}          # dense fp16 / bf16 / fp32 matrix peaks, TFLOP/s (MI355X_MICROARCH.md)
NET_ORDER_TABLES = ("conv3", "conv4", "fc1", "fc2")          # GEMM launches of one forward, in launch order, conv1 + conv2 from the tables
NET_ORDER_GEMM = ("conv2", "conv3", "conv4", "fc1", "fc2")   # ... with conv2 as a GEMM (oz_net_set_tables 0 / 1)


def label_layers(records):
    """records: dispatch-ordered (kernel, ...) tuples.  Returns one layer name (or "") per record: a forward starts at k_lut_ids / k_conv1*,
    whether it uses the table gather decides which GEMM comes first, and the n-th k_gemm launch after the start is the n-th layer of
    OthelloNN -- launch ORDER, not grid size, so every batch size, tile choice and board is labelled the same way.  The table build at
    oz_net_commit (nine GEMMs in a row without a forward start) stays unlabelled."""
    out = [""] * len(records)
    order, idx, open_fw = None, 0, False
    for i, k in enumerate(records):
        if k.startswith("k_lut_ids") or k.startswith("k_conv1"):
            open_fw, idx = True, 0
            order = None if k.startswith("k_lut_ids") else NET_ORDER_GEMM      # k_lut_ids: decided by what follows
            continue
        if not open_fw:
            continue
        if k.startswith("k_conv2_lut"):
            order = NET_ORDER_TABLES
            out[i] = "conv1+conv2 (table gather)"
            continue
        if k.startswith("k_gemm"):
            if order is None:
                order = NET_ORDER_GEMM                        # k_lut_ids followed by a GEMM: conv1 from its table inside conv2's gather
            if idx < len(order):
                out[i] = order[idx]
            idx += 1
            if idx >= len(order):
                open_fw = False
        elif k.startswith("k_heads"):
            open_fw = False
    return out


DESCENT = ("k_select", "k_backup_select", "k_advance", "k_backup_advance")      # the tree launch that opens a batch


def batch_ids(kernels):
    """dispatch-ordered kernel names -> the running number of the network batch each launch belongs to (a batch opens with its descent kernel)"""
    out, b = [], 0
    for k in kernels:
        if k in DESCENT:
            b += 1
        out.append(b)
    return out


def trace(d, out, header, leaves=LEAVES, last_n=0):
    recs = []
    with open(find(d, "_kernel_trace.csv")) as f:
        for r in sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"])):
            k = short(r["Kernel_Name"])
            if k.startswith("__amd") or "at::" in k or "elementwise" in k:
                continue
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            recs.append((k, grid, wg, int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["Scratch_Size"]),
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    layers = label_layers([r[0] for r in recs])
    bid = batch_ids([r[0] for r in recs])
    first_timed = (bid[-1] - last_n + 1) if last_n and bid else 0           # the timed region = the last `last_n` batches of the run
    rows, timed = defaultdict(list), defaultdict(list)
    for r, lay, b in zip(recs, layers, bid):
        rows[r[:7] + (lay,)].append(r[7])
        if last_n and b >= first_timed:
            timed[r[:7] + (lay,)].append(r[7])
    with open(out, "w") as f:
        f.write(f"# {header}\n")
        f.write(f"# per (kernel, grid, layer) averages from the kernel trace; layer = position of the launch inside its forward (launch order); "
                f"{leaves:g} leaves evaluated per launch in the timed window; TFLOP_per_s = ALGORITHMIC fp32 FLOP (precision f16x2 executes 3x, bf16x3 6x that on "
                "the matrix pipe); frac = TFLOP_per_s / the dense matrix peak of the kernel's arithmetic (2500 fp16, 157.3 fp32)\n")
        if last_n:
            f.write(f"# timed_* columns: the launches of the row inside the last {last_n} network batches of the run = bench.py's timed region (full batches); "
                    "calls / avg_us / total_ms: every launch of the run incl. the untimed stagger and warm-up rounds (partly filled batches) -- "
                    "the population rocprofv3's own --stats file averages over\n")
        f.write("kernel,grid_threads,wg,lds_bytes,vgpr,agpr,scratch,calls,avg_us,min_us,max_us,total_ms,timed_calls,timed_avg_us,layer,leaves_per_launch,"
                "algorithmic_TFLOP_per_s,frac\n")
        for k in sorted(rows, key=lambda k: -sum(rows[k])):
            t = rows[k]
            avg = sum(t) / len(t)
            lay = k[7]
            tw = timed.get(k, [])
            tavg = sum(tw) / len(tw) if tw else avg
            tf = frac = lv = ""
            if lay in FLOP and tw:                            # a layer of the timed window: its leaves per launch are known
                val = FLOP[lay] * leaves / (tavg * 1e-9) / 1e12
                peak = next((p for n, p in PEAK.items() if k[0].startswith(n)), None)
                tf, lv = f"{val:.1f}", f"{leaves:g}"
                frac = f"{val / peak:.3f}" if peak else ""
            f.write(f"\"{k[0]}\",{k[1]},{k[2]},{k[3]},{k[4]},{k[5]},{k[6]},{len(t)},{avg / 1e3:.1f},{min(t) / 1e3:.1f},"
                    f"{max(t) / 1e3:.1f},{sum(t) / 1e6:.1f},{len(tw) if tw else ''},{tavg / 1e3 if tw else 0:.1f},{lay},{lv},{tf},{frac}\n")


def pmc(out, header, dirs, last_n=0):
    acc = defaultdict(lambda: [0.0, 0])
    for d in dirs:
        per = defaultdict(list)
        with open(find(d, "_counter_collection.csv")) as f:
            rows = [r for r in sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
                    if not (short(r["Kernel_Name"]).startswith("__amd") or "at::" in r["Kernel_Name"] or "elementwise" in r["Kernel_Name"])]
        # one row per (dispatch, counter): label the dispatches by launch order, then fan the label out to their counter rows
        disp = []
        for r in rows:
            if not disp or disp[-1][0] != r["Dispatch_Id"]:
                disp.append((r["Dispatch_Id"], short(r["Kernel_Name"])))
        lab = dict(zip((x[0] for x in disp), label_layers([x[1] for x in disp])))
        bid = dict(zip((x[0] for x in disp), batch_ids([x[1] for x in disp])))
        first_timed = (max(bid.values()) - last_n + 1) if last_n and bid else 0     # the timed region = the last `last_n` batches of the run
        for r in rows:
            if bid[r["Dispatch_Id"]] < first_timed:
                continue
            k = short(r["Kernel_Name"])
            name = k + (f" [{lab[r['Dispatch_Id']]}]" if lab.get(r["Dispatch_Id"]) else "")
            per[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for key, vals in per.items():
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
    j = {"round": 6, "simulations_per_launch": float(games), "algorithmic_bytes_per_sim": 1300, "source": src, "kernels": {}}
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
            if kernel_sub in r["kernel"] and (int(grid) <= 0 or int(r["grid_threads"]) == int(grid)):
                vals[r["counter"]] = float(r["avg_per_launch"])
    fetch = vals["FETCH_SIZE"] * 1024 * 2
    write = vals["WRITE_SIZE"] * 1024
    # algorithmic bytes per leaf (h2 activations are 4 B per value, like fp32): input pixels + output pixels + the layer's weights once per launch
    alg = {"conv2": 64 * 2048 + 64 * 2048 + 9 * 512 * 512 * 4 / leaves, "conv3": 64 * 2048 + 36 * 2048 + 9 * 512 * 512 * 4 / leaves}[layer]
    if precision == "f32" and layer == "conv2":
        alg = 262144.0 + 18 * 512 * 512 * 4 / LEAVES           # as first committed: the transposed copy of the kernel counted too
    j = {"round": 6, "precision": precision, "kernel": f"{kernel_sub} {layer}", "leaves_per_launch": leaves,
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
