"""CPU-only checks: the C-ABI library loads and exports every declared symbol (no compute), host-side
logic, the oracle's NN legs agree with each other and with torch, and the multi-rank record pooling
works over gloo with world_size 2."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from oracle import nn_numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from othellozero_amd import _lib
    lib = _lib.load()                      # raises OzLibraryError if the .so is missing
    hdr = open(os.path.join(ROOT, "include", "othellozero_amd.h")).read()
    declared = set(re.findall(r"^\s*(?:const char\*|int)\s+(oz_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) > 40
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    bound = set(_lib.SIGNATURES) | {"oz_last_error"}
    assert declared == bound, (declared ^ bound)
    assert lib.oz_version() >= 200                  # 200: the dead edge_cap arguments left the ABI


def test_missing_library_fails_loudly(tmp_path):
    from othellozero_amd import _lib
    with pytest.raises(_lib.OzLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_no_gpu_means_error_not_fallback():
    from othellozero_amd import _lib
    lib = _lib.load()
    if lib.oz_device_count() > 0:
        pytest.skip("a GPU is visible")
    from othellozero_amd.NNet import NNetWrapper
    with pytest.raises(_lib.OzLibraryError):
        NNetWrapper((6, 6), num_channels_1=128)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "othellozero_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
                assert "liboz_oracle" not in src, f


def test_bench_uses_the_oracle_only_as_checker_and_cpu_baseline():
    """bench.py (launcher + timed region) never imports the oracle; bench_legs.py imports it only inside the legs that ARE the checker or the
    CPU baseline (cpu_baseline*, host_cpu_share = how many threads the CPU legs may use, parity_sample, config5_arena's replay, dropin_config0's CPU port) -- never at module level, never in run_secondary"""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M)
    tree = ast.parse(open(os.path.join(ROOT, "bench_legs.py")).read())
    allowed = {"cpu_baseline", "cpu_baseline_config", "parity_sample", "config5_arena", "dropin_config0", "host_threads", "host_cpu_share"}
    for node in tree.body:
        names = []
        for sub in ast.walk(node):
            if isinstance(sub, ast.Import):
                names += [a.name.split(".")[0] for a in sub.names]
            elif isinstance(sub, ast.ImportFrom) and sub.module:
                names.append(sub.module.split(".")[0])
        if "oracle" in names:
            assert isinstance(node, ast.FunctionDef) and node.name in allowed, getattr(node, "name", type(node).__name__)


def test_record_dtype_and_pack_roundtrip():
    from othellozero_amd import _lib
    assert _lib.RECORD_DTYPE.itemsize == 48
    rs = np.random.RandomState(0)
    for n in (4, 6, 8):
        b = rs.rand(n, n, 2) < 0.3
        b[:, :, 1] &= ~b[:, :, 0]
        c0, c1 = _lib.pack_board(b)
        assert (c0, c1) == oracle.pack_board(b)
        assert np.array_equal(_lib.unpack_board(c0, c1, n), b)


def test_shard_games():
    from othellozero_amd.distributed import shard_games
    for total, world in ((32768, 8), (10, 3), (5, 8)):
        seen = []
        for r in range(world):
            first, cnt = shard_games(total, r, world)
            seen += list(range(first, first + cnt))
        assert seen == list(range(total))


def test_weight_shapes_match_survey_parameter_counts():
    from othellozero_amd.weights import init_weights, onn_shapes
    assert sum(int(np.prod(s)) for s in onn_shapes(8)) == 16051265        # params incl. BN statistics
    assert sum(int(np.prod(s)) for s in onn_shapes(6)) == 9745445
    w = init_weights(6, seed=1, channels=128)
    assert len(w) == 40 and w[0].shape == (3, 3, 2, 128) and w[24].shape == (4 * 128, 1024)


def _torch_forward(weights, own, opp, n):
    import torch
    import torch.nn.functional as F
    w = [torch.from_numpy(np.asarray(a, dtype=np.float64)) for a in weights]
    x = torch.from_numpy(nn_numpy.planes(own, opp, n)).permute(0, 3, 1, 2)          # NCHW
    for layer, pad in enumerate((1, 1, 0, 0)):
        k, bias, g, b, mu, var = w[6 * layer:6 * layer + 6]
        x = F.conv2d(x, k.permute(3, 2, 0, 1), bias, padding=pad)
        x = F.batch_norm(x, mu, var, g, b, training=False, eps=1e-3).relu()
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                               # keras Flatten of NHWC
    for layer in (4, 5):
        k, bias, g, b, mu, var = w[6 * layer:6 * layer + 6]
        x = F.batch_norm(x @ k + bias, mu, var, g, b, training=False, eps=1e-3).relu()
    pi = torch.softmax(x @ w[36] + w[37], dim=1)
    v = torch.tanh(x @ w[38] + w[39])[:, 0]
    return pi.numpy(), v.numpy()


def _random_boards(n, count, seed):
    rs = np.random.RandomState(seed)
    own, opp = [], []
    for _ in range(count):
        a = rs.rand(n, n) < 0.35
        b = (rs.rand(n, n) < 0.35) & ~a
        brd = np.stack([a, b], axis=2)
        o, p = oracle.pack_board(brd)
        own.append(o); opp.append(p)
    return np.array(own, np.uint64), np.array(opp, np.uint64)


@pytest.mark.parametrize("n", [6, 8])
def test_oracle_nn_legs_agree(n):
    """float64 NumPy restatement == torch-CPU float64 (independent conv/BN implementation) to 1e-12,
    and the float32 C restatement is within 1e-5 of both."""
    from othellozero_amd.weights import init_weights
    C = 64
    w = init_weights(n, seed=3, channels=C, randomize_all=True)
    for i in (0, 6, 12, 18, 24, 30, 36, 38):
        w[i] = w[i] * 3.0                      # lift the logits away from uniform
    own, opp = _random_boards(n, 5, seed=n)
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    pit, vt = _torch_forward(w, own, opp, n)
    assert np.abs(pi64 - pit).max() < 1e-12 and np.abs(v64 - vt).max() < 1e-12
    assert pi64.std() > 1e-3
    cnet = oracle.CNet(w, n, channels=C, nthreads=2)
    pi32, v32 = cnet.forward(own, opp)
    assert np.abs(pi32 - pi64).max() < 1e-5 and np.abs(v32 - v64).max() < 1e-5


def test_oracle_episode_with_c_net_runs():
    """the cpu_baseline path: oracle search driven by the float32 C net (batch-1 leaf evaluation)"""
    from othellozero_amd.weights import init_weights
    w = init_weights(6, seed=0, channels=64)
    cnet = oracle.CNet(w, 6, channels=64, nthreads=2)
    m = oracle.Mcts(6, 1.0, oracle.QMODE_F64, evaluator=cnet.evaluator())
    ep = m.episode(8, 1.0, 0.9, 1234, 0, max_moves=3)
    assert ep["n_moves"] == 3 and ep["stats"]["expansions"] > 10 and cnet.ctx.calls == ep["stats"]["expansions"]


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from othellozero_amd import _lib
from othellozero_amd.distributed import gather_records, records_to_tensor, tensor_to_records, shard_games
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
first, cnt = shard_games(7, rank, world)           # ragged: 4 + 3 games
rec = np.zeros(0, dtype=_lib.RECORD_DTYPE)
rows = []
for g in range(first, first + cnt):
    k = 3 + g                                       # variable number of plies per game
    r = np.zeros(k, dtype=_lib.RECORD_DTYPE)
    r["game_id"] = g; r["ply"] = np.arange(k); r["black"] = 1000 * g + np.arange(k); r["z"] = 1 - 2 * (g & 1)
    rows.append(r)
rec = np.concatenate(rows) if rows else rec
pooled = tensor_to_records(gather_records(records_to_tensor(rec)))
pooled = pooled[np.lexsort((pooled["ply"], pooled["game_id"]))]
exp = sum(3 + g for g in range(7))
assert pooled.size == exp, (pooled.size, exp)
assert np.array_equal(np.unique(pooled["game_id"]), np.arange(7))
for g in range(7):
    sel = pooled[pooled["game_id"] == g]
    assert np.array_equal(sel["ply"], np.arange(3 + g)) and np.all(sel["black"] == 1000 * g + np.arange(3 + g))
# an empty rank must not break the gather
pooled2 = tensor_to_records(gather_records(records_to_tensor(rec if rank == 0 else rec[:0])))
assert pooled2.size == (sum(3 + g for g in range(4)))
# data-parallel training: a rank whose step failed still joins the gradient all-reduce (poisoned arena) and EVERY rank raises (ADVICE r2)
from othellozero_amd.distributed import GradientAllReduce
class FakeTrainer:
    def sync(self): pass
ar = GradientAllReduce(6, 128, 2, device="cpu")
ar.flat.fill_(float(rank + 1))
ar(FakeTrainer())
assert float(ar.flat[5]) == 1.5                     # the average of the two arenas
err = _lib.OzError(_lib.OZ_ERR_STATE, "range guard") if rank == 1 else None
try:
    ar(FakeTrainer(), failed=err)
    raise SystemExit("the all-reduce with a failed rank did not raise on rank %d" % rank)
except _lib.OzError as e:
    assert e.code == _lib.OZ_ERR_STATE and (("range guard" in str(e)) == (rank == 1))
# ... any exception, not only the library's (ADVICE r3): a bad batch shape on rank 0
try:
    ar(FakeTrainer(), failed=ValueError("bad batch") if rank == 0 else None)
    raise SystemExit("a ValueError on one rank did not end the step on rank %d" % rank)
except ValueError:
    assert rank == 0
except _lib.OzError as e:
    assert rank == 1 and "other rank" in str(e)
# the status word is its own element: a genuinely diverged gradient (NaN in element 0) is NOT reported as a peer's failure
ar.flat.fill_(1.0)
if rank == 0:
    ar.flat[0] = float("nan")
ar(FakeTrainer())
assert bool(torch.isnan(ar.flat[0])) and float(ar.flat[1]) == 1.0
# arena games sharded over the ranks: ONE all-gather of 16-byte results, every rank ends with the whole match sorted by game id
import othellozero_amd.agents as agents
from othellozero_amd.distributed import arena_sharded
def fake_arena(net_a, net_b, board_size, num_games, num_simulations, degree_exploration, seed=0, first_game_id=0, **kw):
    ids = first_game_id + np.arange(num_games)
    return dict(winner=(1 - 2 * (ids & 1)).astype(np.int8), points=(30 + ids).astype(np.int32), n_moves=(50 + ids).astype(np.int32))
agents.arena_batch = fake_arena
res = arena_sharded(None, None, 6, 7, 10)            # 7 games: 4 + 3
assert np.array_equal(res["game_id"], np.arange(7)) and np.array_equal(res["winner"], 1 - 2 * (np.arange(7) & 1))
assert np.array_equal(res["points"], 30 + np.arange(7)) and np.array_equal(res["n_moves"], 50 + np.arange(7))
res1 = arena_sharded(None, None, 6, 1, 10)           # fewer games than ranks: rank 1 plays none
assert np.array_equal(res1["game_id"], [0]) and res1["points"][0] == 30
# a rank whose games raise ends the match on EVERY rank instead of leaving the others in the gather (ADVICE r3)
def failing_arena(*a, first_game_id=0, **kw):
    if first_game_id > 0:
        raise RuntimeError("rank-local failure")
    return fake_arena(*a, first_game_id=first_game_id, **kw)
agents.arena_batch = failing_arena
try:
    arena_sharded(None, None, 6, 7, 10)
    raise SystemExit("arena_sharded did not raise on rank %d" % rank)
except RuntimeError as e:
    assert ("rank-local failure" in str(e)) == (rank == 1) and ("another rank" in str(e)) == (rank == 0)
dist.barrier()
print("RANK_OK", rank)
"""


def test_gloo_world_size_2_record_pooling(tmp_path):
    script = tmp_path / "gloo_worker.py"
    script.write_text(GLOO_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {rank}" in out, out


GLOO8_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from othellozero_amd._lib import RECORD_DTYPE
from othellozero_amd.distributed import gather_records, records_to_tensor, tensor_to_records, shard_games
rank, world = int(os.environ["RANK"]), 8
dist.init_process_group("gloo", rank=rank, world_size=world)
# ragged and empty ranks: rank r contributes COUNTS[r] records; ranks 2 and 5 none, rank 7 far more than the others
COUNTS = [3, 40, 0, 7, 1, 0, 12, 200]
def records_of(r):
    rec = np.zeros(COUNTS[r], dtype=RECORD_DTYPE)
    rec["game_id"] = 1000 * r + np.arange(COUNTS[r]) // 5
    rec["ply"] = np.arange(COUNTS[r]) % 5
    rec["black"] = 7 * r + np.arange(COUNTS[r])
    rec["z"] = 1 - 2 * (np.arange(COUNTS[r]) & 1)
    return rec
pooled = tensor_to_records(gather_records(records_to_tensor(records_of(rank))))
want = np.concatenate([records_of(r) for r in range(world)])             # rank order, every rank's own order kept
assert pooled.size == sum(COUNTS) and pooled.tobytes() == want.tobytes(), (rank, pooled.size)
# every rank empty: still a collective, still fine
none = tensor_to_records(gather_records(records_to_tensor(records_of(2))))
assert none.size == 0
# only the LAST rank has records
last = tensor_to_records(gather_records(records_to_tensor(records_of(7) if rank == 7 else records_of(2))))
assert last.tobytes() == records_of(7).tobytes()
# the 8-way split of a job's game ids: contiguous blocks that cover [0, total) exactly once, ragged and with fewer games than ranks
for total in (32768, 4096, 13, 5, 0):
    blocks = [shard_games(total, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
    assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
dist.barrier()
print("RANK_OK", rank)
"""


def test_gloo_world_size_8_ragged_and_empty_ranks(tmp_path):
    """the exchange step 8 ways (BASELINE configs[2]'s rank count; workers.py:180-184 replaced by one all-gather): ragged contributions,
    empty ranks in the middle, everybody empty, only the last rank non-empty -- pooled == concatenation in rank order on every rank"""
    script = tmp_path / "gloo8_worker.py"
    script.write_text(GLOO8_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {rank}" in out, out


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    """bench.py --gpus 2 inside a launcher environment of ONE rank must fail (exit 2) before touching any GPU -- never a
    silent 1-GPU run reported as the N-GPU number (VERDICT r1 / ADVICE r1)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_parent_reports_failing_ranks():
    """without a GPU the self-launched ranks fail loudly; the parent relays a non-zero exit code and prints no JSON line"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "rank" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_preferred_batch_cap_makes_conv3_whole_grid_rounds():
    """the cap bench.py gives the free-running driver: the largest batch whose conv3 grid (256 x 256 tiles, 256 CUs) is a whole number of
    rounds -- 3640 leaves for 4096 8x8 games on 512 or 256 filters; nothing to cap where the games themselves already are (6x6, 8192 games)"""
    from othellozero_amd.training import preferred_batch_cap
    assert preferred_batch_cap(8, 4096, 512) == 3640 and preferred_batch_cap(8, 4096, 256) == 3640
    for n, G, C in ((8, 4096, 512), (8, 2048, 512), (8, 6000, 512), (8, 4096, 256)):
        cap = preferred_batch_cap(n, G, C)
        tiles = -(-cap * (n - 2) ** 2 // 256) * (C // 256)
        assert 0 < cap < G and tiles % 256 == 0
        assert (-(-(cap + 1) * (n - 2) ** 2 // 256) * (C // 256)) > tiles          # one more leaf would open another round
    assert preferred_batch_cap(6, 4096, 512) == 0 and preferred_batch_cap(8, 8192, 512) == 0
    assert preferred_batch_cap(8, 512, 512) == 0 and preferred_batch_cap(8, 4096, 128) == 0


def test_profile_summaries_label_layers_by_launch_order(tmp_path):
    """tools/summarize_prof.py: the OthelloNN layers are labelled by the ORDER of the GEMM launches inside a forward, not
    by hard-coded grid sizes; the timed columns come from the launches inside the last N network batches only -- a kernel shape that only
    the untimed stagger phase launched gets no timed average and no TFLOP/s (round 2 printed an impossible 1976 TFLOP/s for it)"""
    import csv
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("summarize_prof", os.path.join(root, "tools", "summarize_prof.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    commit = ["k_gemm_h2<H2BigPP>"] * 9
    stagger = ["k_select", "k_compact", "k_lut_ids", "k_conv2_lut_xcd<8, true>", "k_gemm_h2<H2MidPP>", "k_gemm_h2<H2BigPP>", "k_gemm_h2<H2BigPP>",
               "k_splitk_reduce_h2", "k_gemm_h2<H2Thin4w>", "k_heads_lds"]
    timed = ["k_backup_advance", "k_compact", "k_lut_ids", "k_conv2_lut_xcd<8, true>", "k_gemm_h2<H2BigPP>", "k_gemm_h2<H2BigPP>", "k_gemm_h2<H2BigPP>",
             "k_splitk_reduce_h2", "k_gemm_h2<H2Thin4w>", "k_heads_lds"]
    seq = commit + stagger * 5 + timed * 3
    lab = sp.label_layers(seq)
    assert lab[:9] == [""] * 9                                            # the table build at commit is not a forward
    assert [l for l in lab[9:19] if l] == ["conv1+conv2 (table gather)", "conv3", "conv4", "fc1", "fc2"]
    assert sp.label_layers(["k_conv1_h2", "k_gemm_h2<H2BigPP>", "k_gemm_h2<H2MidPP>"])[1:] == ["conv2", "conv3"]
    assert sp.label_layers(["k_lut_ids", "k_gemm_h2<H2BigPPLut>", "k_gemm_h2<H2MidPP>"])[1:] == ["conv2", "conv3"]
    # a synthetic kernel trace: conv3 on the 192-row tile in the stagger phase (200 us), on the 256-row tile in the timed window (1000 us)
    d = tmp_path / "prof"
    d.mkdir()
    with open(d / "run_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z",
                    "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Scratch_Size", "Start_Timestamp", "End_Timestamp"])
        t = 0
        for i, k in enumerate(seq):
            dur = 1000000 if (i >= 9 + 50 and lab[i] == "conv3") else 200000
            grid = 524288 if "BigPP" in k else 786432
            w.writerow([i + 1, f"void {k}(args)", grid, 1, 1, 512, 1, 1, 0, 128, 0, 0, t, t + dur])
            t += dur + 1000
    out = tmp_path / "by_shape.csv"
    sp.trace(str(d), str(out), "synthetic", 3640.0, 3)
    rows = list(csv.DictReader(l for l in open(out) if not l.startswith("#")))
    conv3 = {r["kernel"]: r for r in rows if r["layer"] == "conv3"}
    big, mid = conv3["k_gemm_h2<H2BigPP>"], conv3["k_gemm_h2<H2MidPP>"]
    assert mid["timed_calls"] == "" and mid["algorithmic_TFLOP_per_s"] == ""          # stagger-only shape: no timed average, no TFLOP/s
    assert big["timed_calls"] == "3" and abs(float(big["timed_avg_us"]) - 1000.0) < 1e-6
    want = 2 * 36 * 4608 * 512 * 3640 / 1e-3 / 1e12
    assert abs(float(big["algorithmic_TFLOP_per_s"]) - want) < 0.1 and abs(float(big["frac"]) - want / 2500.0) < 1e-3


def _build_c_host(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "othellozero_amd", "lib")
    exe = str(tmp_path / "selfplay_host")
    r = subprocess.run(["gcc", "-std=c11", "-D_DEFAULT_SOURCE", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "selfplay_host.c"),
                        "-o", exe, "-L" + libdir, "-lothellozero_amd", "-Wl,-rpath," + libdir], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_host_example_compiles_against_the_header(tmp_path):
    """examples/selfplay_host.c: the hot path and the exchange step driven from plain C -- include/othellozero_amd.h is valid C11 (no
    C++-isms), every entry point it uses links against the library; without a GPU the program says so and exits 2"""
    import subprocess
    exe = _build_c_host(tmp_path)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_host_example_plays_and_pools_records(tmp_path):
    """the same program on the GPU: OthelloNN built through oz_net_set_weight, 64 concurrent 6x6 games played to the end by the batched
    engine, the records pooled through the library's own RCCL communicator (one rank here) -- no Python, no torch on the path"""
    import subprocess
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SELFPLAY_HOST_OK" in r.stdout, r.stdout + r.stderr


def test_elf_reader_sees_the_library_s_hip_dependency():
    """_lib._elf_dynamic_strings (no external tools): the built library needs libamdhip64.so.<major>, and a HIP runtime file's SONAME has the same
    form -- the two strings _one_hip_runtime compares before it preloads PyTorch's bundled runtime"""
    from othellozero_amd import _lib
    soname, needed = _lib._elf_dynamic_strings(_lib.LIB_PATH)
    hip = [x for x in needed if x.startswith("libamdhip64.so")]
    assert len(hip) == 1 and re.fullmatch(r"libamdhip64\.so\.\d+", hip[0]), needed
    sys_rt = "/opt/rocm/lib/libamdhip64.so"
    if os.path.exists(sys_rt):
        so, _ = _lib._elf_dynamic_strings(os.path.realpath(sys_rt))
        assert so == hip[0]                                      # the runtime it was linked against
    with pytest.raises(ValueError):
        _lib._elf_dynamic_strings(os.path.join(ROOT, "README.md"))


def test_library_and_pytorch_share_one_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as the system one) and load it by path: with the library loaded first a process
    used to hold TWO HIP runtimes, whose streams / device pointers are invalid in each other (torch tensors handed to the library; RCCL's
    `unhandled cuda error` in oz_comm_create).  _lib.load() now brings PyTorch's runtime in first where PyTorch is installed: either import
    order ends with exactly one libamdhip64 mapped."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = ("; print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))")
    for head in ("from othellozero_amd import _lib; _lib.load(); import torch", "import torch; from othellozero_amd import _lib; _lib.load()"):
        r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {root!r}); " + head + tail], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        libs = eval(r.stdout.strip().splitlines()[-1])
        assert len(libs) == 1, (head, libs)


def test_bench_line_fits_the_drivers_tail_at_any_rank_count():
    """bench_legs.compact_line on the committed round-6 record (profiles/r6_bench_detail_driver_style.json), also blown up to 8 ranks with every
    optional leg present: strict JSON, contract keys first, < 6000 bytes (the driver keeps an 8 KB tail; round 5's 33 KB line was unparsed)"""
    import copy
    import json
    import bench_legs
    det = json.load(open(os.path.join(ROOT, "profiles", "r6_bench_detail_driver_style.json")))
    for world in (1, 8):
        d = copy.deepcopy(det)
        d["n_gpus"] = d["rccl_ranks"] = world
        d["per_rank_records"] = [81234 + r for r in range(world)]
        d["per_rank"]["ms_per_step"] = [423.0320192 + r for r in range(world)]
        d["detail_file"] = "/some/long/scratch/path/of/the/driver/box/repo/bench_detail.json"
        text = bench_legs.compact_line(d)
        assert len(text) < 6000, len(text)

        def no_constants(name):
            raise ValueError(name)
        line = json.loads(text, parse_constant=no_constants)
        keys = list(line)
        assert keys[:15] == ["metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                             "dtype", "data", "config", "roofline"]
        assert keys[15] == "cpu_baseline" and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
        rf = line["roofline"]
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["bound"] == "mfma" and rf["traffic"] > rf["algorithmic_bytes_per_launch"]
        assert line["value"] == det["value"] and line["dtype"] == "f32 (3xbf16 split)" and line["roofline_f32"]["peak"] == 157.3
        assert len(line["per_rank_records"]) == world and "kernels" not in line and "config5" not in line and "precisions" not in line


def test_precision_modes_and_batch_cap():
    from othellozero_amd.NNet import PRECISION_MODES
    from othellozero_amd.training import preferred_batch_cap
    assert PRECISION_MODES == {"f32": 0, "f16x2": 1, "bf16x3": 2}
    # bf16x3's one tile is 128 x 256: 4096 leaves are whole rounds of conv3 AND conv4 -> no cap; the other precisions cap at 4.0 rounds of 256-row tiles
    assert preferred_batch_cap(8, 4096, 512, "bf16x3") == 0 and preferred_batch_cap(8, 4096, 512, "f32") == 3640 and preferred_batch_cap(8, 4096, 512) == 3640


def test_lds_swizzles_are_conflict_free_for_the_16x16x32_operand_map():
    """the two XOR swizzles of the GEMM kernels' LDS images, checked as arithmetic: a ds_read_b128 is served in 16-lane groups
    {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32 for the upper half, MI355X_MICROARCH.md); with the 16x16x32 operand map -- lane l reads row l & 15,
    k-group l >> 4 -- the 16 lanes of a group must hit 16 different 16-byte bank quads (address / 16 mod 16).
    h2 layout (oz_net_h2.h): rows of 8 chunks, chunk (2 kg + plane) ^ h2_swz(row), h2_swz(r) = bit1(r) | 6 bit3(r);
    b3 layout (oz_net_b3.h): rows of 12 chunks, chunk 4 plane + (kg ^ g(row)), g(r) = (-(r >> 2)) & 3."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]

    def h2_chunk(lane, plane):
        r, kg = lane & 15, lane >> 4
        return 8 * r + ((2 * kg + plane) ^ (((r >> 1) & 1) | (((r >> 3) & 1) * 6)))

    def b3_chunk(lane, plane):
        r, kg = lane & 15, lane >> 4
        return 12 * r + 4 * plane + (kg ^ ((4 - (r >> 2)) & 3))
    for g in groups:
        assert len(g) == 16
        for plane in (0, 1):
            assert len({h2_chunk(l, plane) % 16 for l in g}) == 16, ("h2", plane, g)
        for plane in (0, 1, 2):
            assert len({b3_chunk(l, plane) % 16 for l in g}) == 16, ("b3", plane, g)
        # (what the swizzle is for: without it the b3 image would put rows r and r + 4 k of a group on the same quad)
        assert len({(12 * (l & 15) + (l >> 4)) % 16 for l in g}) < 16
    # every (row, k-group) chunk of a 16-row block stays inside its plane's four slots: the image is a permutation of the 192 chunks
    assert sorted(b3_chunk(l, p) for l in range(64) for p in range(3)) == list(range(192))
    assert sorted(h2_chunk(l, p) for l in range(64) for p in range(2)) == list(range(128))


def test_b3_staging_map_matches_the_fragment_reads():
    """k_gemm_b3's LDS-DMA staging map against its fragment reads, as arithmetic: DMA instruction i of a 16-row block, lane l, writes LDS chunk
    64 i + l of the block and fetches the source chunk (plane, k-group) the kernel computes for it; the MFMA fragment read of lane (row, k-group) for
    plane p must find exactly (p, k-group) of that row there.  (A DMA instruction spans 5.3 rows of 12 chunks: the map is per lane, not per row.)"""
    holds = {}
    for i in range(3):
        for lane in range(64):
            cidx = 64 * i + lane
            r, pos = divmod(cidx, 12)
            plane, kg_src = pos >> 2, (pos & 3) ^ ((4 - (r >> 2)) & 3)
            assert cidx not in holds
            holds[cidx] = (r, plane, kg_src)                  # what the kernel's src_chunk = plane * 4 + kg_src of row r denotes
    assert len(holds) == 192 and {v[0] for v in holds.values()} == set(range(16))
    for r16 in range(16):
        for kg in range(4):
            for p in range(3):
                lofs_chunks = 12 * r16 + (kg ^ ((4 - (r16 >> 2)) & 3))        # lofs / 16
                assert holds[lofs_chunks + 4 * p] == (r16, p, kg)
