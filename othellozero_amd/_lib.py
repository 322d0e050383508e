"""ctypes binding of libothellozero_amd.so (include/othellozero_amd.h).

There is NO fallback: if the HIP library cannot be loaded, importing anything that
computes raises `OzLibraryError`.  The library is built in-tree by
`python -m othellozero_amd.build` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OZ_LIB_PATH") or os.path.join(_HERE, "lib", "libothellozero_amd.so")

OZ_OK, OZ_ERR_HIP, OZ_ERR_ARG, OZ_ERR_CAPACITY, OZ_ERR_KEY, OZ_ERR_STATE = range(6)
QMODE_NEP50, QMODE_F64 = 0, 1
DEDUP_DEFAULT, DEDUP_ON, DEDUP_OFF = 0, 1, 2                                          # oz_selfplay_config.dedup
NET_OPT_SIMPLE_LOOP, NET_OPT_ACT_TARGET_LOG2, NET_OPT_LOW_GUARD_LOG2, NET_OPT_SELF_CHECK, NET_OPT_W_TARGET_LOG2, NET_OPT_F32_STD_TILE, NET_OPT_LATENCY_SPLITS = 1, 2, 3, 4, 5, 6, 7      # oz_net_set_option
NET_OPT_CONV3_TILE = 8
NET_OPT_LOW_LOOP_PHASES = 9
NET_OPT_B3_TILE = 10
NET_INFO_CONV3_TILE_ROWS, NET_INFO_SELF_CHECK_GUARD, NET_INFO_ARITHMETIC = 1, 2, 3                                                          # oz_net_get_info
LEAF_IDLE, LEAF_TERMINAL, LEAF_EVAL = 0, 1, 2
VT_INT, VT_F32, VT_F64 = 0, 1, 2
NET_KERNELS = ("input", "conv2", "conv3", "conv4", "fc1", "fc2", "heads")           # OZ_NET_KERNELS slots
TREE_KERNELS = ("select", "compact", "network", "expand_backup", "roots_move")       # OZ_TREE_KERNELS slots


class OzLibraryError(RuntimeError):
    pass


class OzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[oz error {code}] {msg}")
        self.code = code


class SelfplayConfig(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("num_games", C.c_int32), ("sims", C.c_int32), ("q_mode", C.c_int32),
        ("c", C.c_double), ("temperature", C.c_double), ("e_greedy", C.c_double),
        ("seed", C.c_uint64), ("first_game_id", C.c_uint64), ("game_id_stride", C.c_uint64),
        ("refill", C.c_int32), ("node_cap", C.c_int32), ("reserved0", C.c_int32), ("record_cap", C.c_int32),
        ("dedup", C.c_int32), ("batch_cap", C.c_int32), ("eval_cache", C.c_int32), ("reserved", C.c_int32),
    ]


class SelfplayStats(C.Structure):
    _fields_ = [
        ("simulations", C.c_int64), ("node_visits", C.c_int64), ("expansions", C.c_int64),
        ("terminal_hits", C.c_int64), ("fallbacks", C.c_int64),
        ("moves", C.c_int64), ("games_completed", C.c_int64), ("records", C.c_int64),
        ("live_games", C.c_int32), ("overflow", C.c_int32),
        ("leaves_evaluated", C.c_int64),
    ]


# numpy view of oz_record (48 bytes)
RECORD_DTYPE = np.dtype([
    ("black", "<u8"), ("white", "<u8"), ("final_black", "<u8"), ("final_white", "<u8"), ("game_id", "<u8"),
    ("ply", "u1"), ("action", "u1"), ("player", "i1"), ("z", "i1"), ("greedy", "u1"), ("pad", "u1", (3,)),
])
assert RECORD_DTYPE.itemsize == 48

_u64p, _i32p, _f32p, _f64p = (C.POINTER(t) for t in (C.c_uint64, C.c_int32, C.c_float, C.c_double))
_u8p, _i8p, _i64p, _vp = C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_int64), C.c_void_p

# name -> argtypes: every symbol include/othellozero_amd.h declares (tests check the export list)
SIGNATURES = {
    "oz_version": [], "oz_device_count": [], "oz_set_device": [C.c_int],
    "oz_rules_legal_moves": [_u64p, _u64p, C.c_int, C.c_int, _u64p],
    "oz_rules_apply_moves": [_u64p, _u64p, _u8p, C.c_int, C.c_int, _u64p, _u64p],
    "oz_rules_status": [_u64p, _u64p, C.c_int, C.c_int, _u8p, _i32p, _i32p, _i8p],
    "oz_rules_play": [_u64p, _u64p, _i8p, _u8p, C.c_int, C.c_int, _u64p, _u64p, _i8p, _u8p],
    "oz_net_create": [C.POINTER(_vp), C.c_int, C.c_int, C.c_int],
    "oz_net_create_bnn": [C.POINTER(_vp), C.c_int, C.c_int, C.c_int],
    "oz_net_create_stub": [C.POINTER(_vp), C.c_int, C.c_uint64, C.c_uint64, C.c_int],
    "oz_net_destroy": [_vp], "oz_net_num_weights": [_vp],
    "oz_net_weight_size": [_vp, C.c_int, _i64p],
    "oz_net_set_weight": [_vp, C.c_int, _f32p, C.c_int64],
    "oz_net_get_weight": [_vp, C.c_int, _f32p, C.c_int64],
    "oz_net_commit": [_vp],
    "oz_net_set_precision": [_vp, C.c_int], "oz_net_get_precision": [_vp], "oz_net_check": [_vp],
    "oz_net_init_random": [_vp, C.c_uint64],
    "oz_net_predict": [_vp, _u64p, _u64p, C.c_int, _f32p, _f32p],
    "oz_net_predict_boards": [_vp, _u8p, C.c_int, _f32p, _f32p],
    "oz_net_time_forward": [_vp, C.c_int, C.c_int, _f32p],
    "oz_net_profile": [_vp, C.c_int], "oz_net_profile_read": [_vp, _f64p, _i64p],
    "oz_net_profiled_layer": [_vp, C.POINTER(C.c_int)], "oz_net_set_tables": [_vp, C.c_int],
    "oz_net_profile_kernels": [_vp, _f64p, _i64p, C.c_int],
    "oz_net_set_eval_cache": [_vp, C.c_int64], "oz_net_eval_cache_stats": [_vp, _i64p, _i64p, _i64p, _i64p],
    "oz_net_set_option": [_vp, C.c_int, C.c_int], "oz_net_get_info": [_vp, C.c_int, C.POINTER(C.c_int)],
    "oz_net_get_scaling": [_vp, C.c_int, _i32p, C.c_int64],
    "oz_net_self_check": [_vp, _f64p, _f64p, C.POINTER(C.c_int)],
    "oz_mcts_create": [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int],
    "oz_mcts_destroy": [_vp], "oz_mcts_reset": [_vp, C.c_int], "oz_mcts_set_dedup": [_vp, C.c_int],
    "oz_mcts_set_roots": [_vp, _u64p, _u64p, _u8p],
    "oz_mcts_simulate": [_vp, _vp, C.c_int], "oz_mcts_select": [_vp],
    "oz_mcts_leaves": [_vp, _i32p, _u64p, _u64p],
    "oz_mcts_backup": [_vp, _f32p, _f32p],
    "oz_mcts_last_value": [_vp, _f64p, _i32p, _i32p],
    "oz_mcts_root_counts": [_vp, _i32p, _u64p, _i32p],
    "oz_mcts_policy": [_vp, C.c_double, _u64p, C.POINTER(C.c_double), _i32p],
    "oz_mcts_num_nodes": [_vp, _i32p],
    "oz_mcts_dump_node": [_vp, C.c_int, C.c_int, _u64p, _u64p, _i32p, _u64p, _i32p, _f64p, _u8p, _f64p],
    "oz_mcts_stats": [_vp, _i64p],
    "oz_selfplay_create": [C.POINTER(_vp), C.POINTER(SelfplayConfig), _vp],
    "oz_selfplay_destroy": [_vp], "oz_selfplay_run": [_vp, C.c_int], "oz_selfplay_run_steps": [_vp, C.c_int], "oz_selfplay_sync": [_vp],
    "oz_selfplay_stagger": [_vp, C.c_int], "oz_selfplay_profile": [_vp, C.c_int], "oz_selfplay_set_batch_cap": [_vp, C.c_int], "oz_selfplay_set_dedup": [_vp, C.c_int],
    "oz_selfplay_profile_read": [_vp, _f64p, _i64p, C.c_int],
    "oz_selfplay_get_stats": [_vp, C.POINTER(SelfplayStats)],
    "oz_selfplay_state": [_vp, _u64p, _u64p, _i8p, _u8p, _i32p, _u64p],
    "oz_selfplay_records": [_vp, _vp, C.c_int64, _i64p],
    "oz_selfplay_records_device": [_vp, _vp, C.c_int64, _i64p],
    "oz_selfplay_last_counts": [_vp, _i32p],
    "oz_selfplay_eval_time": [_vp, _f64p, _i64p, _i64p],
    "oz_comm_unique_id": [_u8p], "oz_comm_create": [C.POINTER(_vp), _u8p, C.c_int, C.c_int], "oz_comm_destroy": [_vp],
    "oz_selfplay_gather_records": [_vp, _vp, C.c_int64, _vp, C.c_int64, _i64p, _i64p],
    "oz_arena_create": [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_uint64, _vp, _vp, C.c_int],
    "oz_arena_destroy": [_vp], "oz_arena_run": [_vp], "oz_arena_run_rounds": [_vp, C.c_int], "oz_arena_stats": [_vp, _i64p, _i64p],
    "oz_arena_set_dedup": [_vp, C.c_int], "oz_arena_set_eval_cache": [_vp, C.c_int], "oz_arena_profile": [_vp, C.c_int], "oz_arena_profile_read": [_vp, _f64p, _i64p, C.c_int], "oz_arena_leaves_evaluated": [_vp, _i64p, _i64p],
    "oz_arena_results": [_vp, _i8p, _i32p, _i32p, _u8p, _i8p, _u64p, _u64p],
    "oz_examples_expand": [_vp, C.c_int64, C.c_int, C.c_int, _u8p, _i32p, _i8p],
    "oz_symmetry_table": [C.c_int, _i32p],
    "oz_selftest_arith": [_f64p, _f64p, C.c_int, _f64p, _f64p, _f32p, _f32p],
    "oz_selftest_mfma_rate": [C.c_int, C.c_double, _f64p, _f64p, _f64p],
    "oz_selftest_b3_split": [_f32p, C.c_int64, _f32p, _f32p],
    "oz_trainer_arena_size": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)],
    "oz_trainer_create": [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                          C.c_uint64, _vp],
    "oz_trainer_destroy": [_vp],
    "oz_trainer_set_weight": [_vp, C.c_int, _f32p, C.c_int64],
    "oz_trainer_get_weight": [_vp, C.c_int, _f32p, C.c_int64],
    "oz_trainer_get_grad": [_vp, C.c_int, _f32p, C.c_int64],
    "oz_trainer_grad_arena": [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)],
    "oz_trainer_forward_backward": [_vp, _u64p, _u64p, _f32p, _f32p, C.c_int, _f32p],
    "oz_trainer_apply": [_vp],
    "oz_trainer_set_dataset": [_vp, _u64p, _u64p, _f32p, _f32p, C.c_int64],
    "oz_trainer_fit_epoch": [_vp, _i32p, C.c_int64, C.c_int, _f32p],
    "oz_trainer_outputs": [_vp, C.c_int, _f32p, _f32p],
    "oz_trainer_get_activation": [_vp, C.c_int, C.c_int, _f32p, C.c_int64],
    "oz_trainer_sync": [_vp],
    "oz_trainer_set_precision": [_vp, C.c_int],
    "oz_trainer_step_count": [_vp, C.POINTER(C.c_int64)],
}

_LIB = None
PRELOAD_TORCH_HIP = True          # set to False before the first load() to bind the library to the system ROCm runtime even where PyTorch is installed


HIP_RUNTIME_BOUND = None          # after load(): "torch:<path>" (PyTorch's bundled runtime was preloaded) or "system"


def _elf_dynamic_strings(path):
    """(DT_SONAME or None, [DT_NEEDED ...]) of a little-endian ELF64 shared object, read without external tools"""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError("not a little-endian ELF64 file")
    phoff, = struct.unpack_from("<Q", data, 0x20)
    phentsize, phnum = struct.unpack_from("<HH", data, 0x36)
    loads, dyn = [], None
    for k in range(phnum):
        p_type, _flags, p_offset, p_vaddr, _paddr, p_filesz = struct.unpack_from("<IIQQQQ", data, phoff + k * phentsize)
        if p_type == 1:
            loads.append((p_vaddr, p_offset, p_filesz))
        elif p_type == 2:
            dyn = (p_offset, p_filesz)
    if dyn is None:
        return None, []
    entries = [struct.unpack_from("<qQ", data, dyn[0] + 16 * k) for k in range(dyn[1] // 16)]
    strtab = next((v for t, v in entries if t == 5), None)
    if strtab is None:
        return None, []
    off = next((strtab - va + fo for va, fo, sz in loads if va <= strtab < va + sz), strtab)

    def name(idx):
        end = data.index(b"\0", off + idx)
        return data[off + idx:end].decode()
    soname = next((name(v) for t, v in entries if t == 14), None)
    return soname, [name(v) for t, v in entries if t == 1]


def _one_hip_runtime(lib_path):
    """PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so) and load it by path.  If this library is loaded first it
    binds to the system runtime, a later `import torch` brings the bundled one in as well, and the process holds TWO HIP / HSA runtimes:
    streams and device pointers of one are invalid handles in the other (torch tensors handed to the library, the library's RCCL
    exchange step on torch's RCCL -- `ncclCommInitRank: unhandled cuda error`).  So where PyTorch is installed but not imported yet, its
    runtime is loaded first and the library's DT_NEEDED entry resolves to it -- ONLY when that is the runtime the library was linked
    against: the bundled file's SONAME must equal the library's DT_NEEDED name (libamdhip64.so.<major>); otherwise the system runtime is
    used and a warning says so (a silent major-version mix fails, or loads two runtimes anyway, with no message).  Where
    PyTorch is absent (a C host, examples/selfplay_host.c) the system runtime and the system RCCL are used together.
    `othellozero_amd._lib.HIP_RUNTIME_BOUND` records what happened; PRELOAD_TORCH_HIP = False (before the first load) opts out."""
    global HIP_RUNTIME_BOUND
    import importlib.util
    import sys
    import warnings
    HIP_RUNTIME_BOUND = "system"
    if not PRELOAD_TORCH_HIP:
        return
    try:
        spec = importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
    except Exception:            # no PyTorch, or an unusual layout: the system runtime it is
        return
    if not cand or not os.path.exists(cand):
        return
    try:
        bundled_soname, _ = _elf_dynamic_strings(cand)
        _, needed = _elf_dynamic_strings(lib_path)
        wanted = next((x for x in needed if x.startswith("libamdhip64.so")), None)
    except Exception as e:
        warnings.warn(f"othellozero_amd: could not compare PyTorch's bundled HIP runtime with the library's ({e}); using the system runtime")
        return
    if "torch" in sys.modules:
        HIP_RUNTIME_BOUND = "torch:" + cand          # already in the process (loaded by path): DT_NEEDED resolves to it by SONAME, or not at all
        if wanted and bundled_soname != wanted:
            warnings.warn(f"othellozero_amd: PyTorch's HIP runtime ({bundled_soname}) is not the one this library was linked against ({wanted}): "
                          "two HIP runtimes in one process; do not hand torch device pointers or streams to the library")
            HIP_RUNTIME_BOUND = "system"
        return
    if wanted and bundled_soname == wanted:
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        HIP_RUNTIME_BOUND = "torch:" + cand
    else:
        warnings.warn(f"othellozero_amd: PyTorch bundles HIP runtime {bundled_soname}, the library was linked against {wanted}: not preloading it; "
                      "a later `import torch` will put two HIP runtimes in this process (keep torch device pointers away from the library)")


def load(path=None):
    """Load the HIP library (no compute).  Raises OzLibraryError if it is missing or unloadable."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OzLibraryError(
            f"{p} not found: the HIP extension is required (no CPU fallback). Build it with "
            "`python -m othellozero_amd.build` (hipcc --offload-arch=gfx950).")
    _one_hip_runtime(p)
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise OzLibraryError(f"cannot load {p}: {e}") from e
    lib.oz_last_error.restype = C.c_char_p
    lib.oz_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OzLibraryError(f"{p} does not export {name}") from e
        fn.argtypes = args
        fn.restype = C.c_int
    if path is None:
        _LIB = lib
    return lib


def check(rc):
    if rc != OZ_OK:
        msg = load().oz_last_error().decode("utf-8", "replace")
        if rc == OZ_ERR_KEY:
            raise KeyError(msg)
        raise OzError(rc, msg)


def require_gpu():
    lib = load()
    if lib.oz_device_count() <= 0:
        raise OzLibraryError("no HIP device visible: othellozero_amd computes on an MI355X only (no CPU fallback)")
    return lib


def mfma_rate(kind, target_ms=50.0):
    """oz_selftest_mfma_rate: {"tflops", "clock_ghz", "ms"} of a pure-MFMA loop on the current device; kind "f32" / "f16" / "bf16" """
    t, g, m = C.c_double(), C.c_double(), C.c_double()
    check(require_gpu().oz_selftest_mfma_rate({"f32": 0, "f16": 1, "bf16": 2}[kind], float(target_ms), C.byref(t), C.byref(g), C.byref(m)))
    return {"tflops": t.value, "clock_ghz": g.value, "ms": m.value}


# numpy -> pointer helpers
def p_u64(a): return a.ctypes.data_as(_u64p)
def p_i32(a): return a.ctypes.data_as(_i32p)
def p_f32(a): return a.ctypes.data_as(_f32p)
def p_f64(a): return a.ctypes.data_as(_f64p)
def p_u8(a): return a.ctypes.data_as(_u8p)
def p_i8(a): return a.ctypes.data_as(_i8p)
def p_i64(a): return a.ctypes.data_as(_i64p)


def pack_board(board):
    """(n,n,2) array-like (channel 0, channel 1) -> two python ints, bit = row*8 + col."""
    b = np.asarray(board).astype(bool)
    n = b.shape[0]
    w = (np.uint64(1) << (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(n, dtype=np.uint64)[None, :]))
    return int(w[b[:, :, 0]].sum(dtype=np.uint64)), int(w[b[:, :, 1]].sum(dtype=np.uint64))


def unpack_board(c0, c1, n):
    sq = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(n, dtype=np.uint64)[None, :])
    out = np.zeros((n, n, 2), dtype=bool)
    out[:, :, 0] = (np.uint64(c0) >> sq) & np.uint64(1)
    out[:, :, 1] = (np.uint64(c1) >> sq) & np.uint64(1)
    return out
