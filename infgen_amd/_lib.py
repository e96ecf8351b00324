"""ctypes binding of libinfgen_hip.so (C ABI: include/infgen_hip.h).

The product path has NO fallback: if the shared library is missing or an entry point fails,
an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libinfgen_hip.so')
MAX_LAYERS = 8

_p = C.c_void_p
_i = C.c_int
_f = C.c_float


class LinearDesc(C.Structure):
    """InfgenLinearDesc of include/infgen_hip.h"""
    _fields_ = [('X', C.c_void_p), ('ldx', C.c_int), ('gather', C.c_void_p), ('rows', C.c_int), ('K', C.c_int),
                ('Wp', C.c_void_p), ('Np', C.c_int), ('bias', C.c_void_p), ('N', C.c_int),
                ('pre_g', C.c_void_p), ('pre_b', C.c_void_p), ('post_g', C.c_void_p), ('post_b', C.c_void_p), ('relu', C.c_int),
                ('Y', C.c_void_p), ('ldy', C.c_int)]


class EdgeBuf(C.Structure):
    _fields_ = [('off', _p), ('cnt', _p), ('src', _p), ('raw', _p), ('rhat', _p), ('total', _p),
                ('cap', _i), ('_pad', _i)]


class RadiusEdges(C.Structure):
    """InfgenRadiusEdges (include/infgen_hip.h)"""
    _fields_ = [('n_q', _i), ('_pad0', _i),
                ('q_node', _p), ('q_pt', _p), ('q_c0', _p), ('q_c1', _p), ('q_self', _p), ('q_pair_off', _p),
                ('p_pos', _p), ('p_head', _p), ('p_inv', _p),
                ('c_pos', _p), ('c_head', _p), ('c_inv', _p), ('c_ok', _p), ('c_src', _p), ('pair_ok', _p),
                ('radius', C.c_float), ('K', _i), ('gap_rule', _i), ('index_diff', _i), ('e_base', _i), ('_pad1', _i)]


class Insertion(C.Structure):
    """InfgenInsertion (include/infgen_hip.h)"""
    _fields_ = [('attn_occ2sa', _p * 3), ('attn_pt2sa', _p * 3), ('attn_a2sa', _p * 3),
                ('four_a2sa', _p), ('four_pt2sa', _p),
                ('head_state', _p), ('head_type', _p), ('head_shape', _p), ('head_pos', _p), ('head_heading', _p),
                ('head_offset', _p), ('occ_embed', _p), ('shape_emb', _p), ('type_a_emb', _p), ('f_seed', _p),
                ('occ', _p), ('occ_emb', _p), ('Kocc', _p * 3), ('Vocc', _p * 3), ('mapK', _p * 3), ('mapV', _p * 3),
                ('Ksa', _p * 3), ('Vsa', _p * 3), ('Kh', _p * 3), ('Vh', _p * 3), ('Xc', _p),
                ('zero_agg', _p), ('zero_z', _p), ('zero_sig', _p),
                ('XS', _p), ('QS', _p), ('US', _p), ('AGGS', _p), ('ZS', _p), ('SIGS', _p), ('KN', _p), ('VN', _p),
                ('ea_s', EdgeBuf), ('em_s', EdgeBuf), ('ea_h', EdgeBuf), ('em_h', EdgeBuf),
                ('occ_off', _p), ('occ_cnt', _p), ('occ_src', _p),
                ('active', _p), ('n_new', _p), ('inserted', _p), ('new_row', _p), ('new_cell', _p), ('new_local', _p), ('new_shape', _p),
                ('prev_row', _p), ('prev_mask', _p), ('pend_row', _p), ('pend_mask', _p),
                ('hv_ovr', _p), ('shape_all', _p),
                ('hid', _p), ('lg_state', _p), ('lg_type', _p), ('shape', _p), ('lg_pos', _p), ('lg_heading', _p), ('offset', _p),
                ('t1', _p), ('t2', _p), ('shp', _p), ('host_dec', _p),
                ('r_seed', C.c_float), ('r_a2sa', C.c_float), ('r_pl2sa', C.c_float), ('angle_interval', C.c_float),
                ('n_heading', _i), ('force_enter', _i), ('insert_k', _i), ('max_new', _i)]


class Options(C.Structure):
    _fields_ = [('use', _i), ('attn_mode', _i), ('gemm_terms', _i), ('fourier_mode', _i), ('edge_fuse', _i), ('edge_loop', _i),
                ('overlap', _i), ('row_group_margin', _i), ('layers_p', _i), ('rhat_format', _i), ('edge_kernel', _i), ('_pad0', _i), ('row_groups', _p), ('n_row_groups', _p)]


OPTIONS_VALUE_BYTES = C.sizeof(_i) * 12       # the integer switches of Options (the two pointers follow)


class Rollout(C.Structure):
    _fields_ = [
        ('S', _i), ('A_cap', _i), ('T', _i), ('M_cap', _i), ('W', _i), ('ring', _i), ('R', _i),
        ('token_size', _i), ('grid_size', _i), ('num_layers', _i), ('force_valid', _i), ('store_logits', _i),
        ('r_map', _f), ('r_agent', _f),
        ('n_agents', _p), ('n_map', _p), ('av_index', _p),
        ('pos', _p), ('head', _p), ('state', _p), ('token', _p), ('grid', _p),
        ('tmask', _p), ('imask', _p), ('catflag', _p), ('type', _p), ('bos', _p),
        ('map_pos', _p), ('map_orient', _p),
        ('attn_t', _p * MAX_LAYERS), ('attn_m', _p * MAX_LAYERS), ('attn_a', _p * MAX_LAYERS),
        ('four_t', _p), ('four_m', _p), ('four_a', _p), ('four_xa', _p),
        ('fusion_pack', _p), ('tok_head_pack', _p), ('st_head_pack', _p),
        ('tok_tab', _p), ('grid_tab', _p), ('state_emb', _p), ('cat_agent', _p), ('cat_seed', _p),
        ('vocab', _p), ('grid_xy', _p),
        ('ringK', _p * MAX_LAYERS), ('ringV', _p * MAX_LAYERS), ('mapK', _p * MAX_LAYERS), ('mapV', _p * MAX_LAYERS),
        ('X', _p), ('Q', _p), ('U', _p), ('Ka', _p), ('Va', _p), ('AGG', _p), ('Z', _p), ('SIG', _p),
        ('et', EdgeBuf), ('em', EdgeBuf), ('ea', EdgeBuf),
        ('raw2', _p), ('cat', _p), ('fus_in', _p), ('tmp1', _p), ('tmp2', _p),
        ('next_token', _p), ('next_state', _p), ('logits', _p),
        ('teacher_token', _p), ('teacher_state', _p),
        ('pred_traj', _p), ('pred_head', _p), ('pred_state', _p),
        ('first_new', _p), ('hv_ovr', _p),
        ('sample_k', _i), ('_pad1', _i), ('sample_u', _p), ('logits_scratch', _p),
        ('opts', Options),
        ('teacher_grid', _p),
        ('four_t_dt', _p),
        ('teacher_pos', _p), ('teacher_head', _p),
        ('map_scene', _p),
        ('tap_x', _p),
    ]


# symbol -> (restype, argtypes); every symbol include/infgen_hip.h declares
SYMBOLS = {
    'infgen_layout_query': (_i, [_i]),
    'infgen_attn_pack_offset': (_i, [C.c_char_p]),
    'infgen_fourier_pack_offset': (_i, [C.c_char_p, _i, _i]),
    'infgen_last_error': (C.c_char_p, []),
    'infgen_linear': (_i, [_p, _i, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p]),
    'infgen_linear_multi': (_i, [_p, _i, _p]),
    'infgen_radius_edges': (_i, [_p, _p, _p]),
    'infgen_motion_features': (_i, [_p, _p, _p, _p, _i, _i, _p, _p]),
    'infgen_layernorm': (_i, [_p, _i, _p, _p, _p, _p]),
    'infgen_fourier_embed': (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _p]),
    'infgen_set_fourier_mode': (_i, [_i]),
    'infgen_set_attn_mode': (_i, [_i]),
    'infgen_set_edge_fuse': (_i, [_i]),
    'infgen_set_layers_p': (_i, [_i]),
    'infgen_layers_p_capacity': (_i, []),
    'infgen_rollout_validate': (_i, [_p]),
    'infgen_set_edge_loop': (_i, [_i]),
    'infgen_set_rhat_format': (_i, [_i]),
    'infgen_set_edge_kernel': (_i, [_i]),
    'infgen_mlp_embedding': (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    'infgen_get_options': (_i, [C.POINTER(Options)]),
    'infgen_thread_options': (_i, [C.POINTER(Options)]),
    'infgen_get_effective_options': (_i, [C.POINTER(Options)]),
    'infgen_edge_fused_occupancy': (_i, []),
    'infgen_debug_stream_read': (_i, [_p, C.c_ulonglong, _i, _p, _p]),
    'infgen_set_overlap': (_i, [_i]),
    'infgen_edge_attn_fused': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'infgen_edge_attn_fused_r24': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'infgen_fourier_embed_r24': (_i, [_p, _i, _p, _i, _p, _p, _p]),
    'infgen_embedding_sum4': (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _p, _p]),
    'infgen_fourier_last_dim_table': (_i, [_p, _i, _p, _p]),
    'infgen_fourier_embed_tab': (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _p]),
    'infgen_distance_to_nearest_object': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, C.c_float, _p, _p, _p]),
    'infgen_kinematic_features': (_i, [_p, _p, _p, _p, _i, _i, C.c_float, _p, _p, _p, _p, _p]),
    'infgen_time_to_collision': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    'infgen_set_gemm_terms': (_i, [_i]),
    'infgen_active_row_groups': (_i, [_p, _i, _i, _i, _p, _p, _p]),
    'infgen_set_row_groups': (_i, [_p, _p, _i]),
    'infgen_set_row_limits': (_i, [_p, _i, _i]),
    'infgen_fetch_enterings': (_i, [_p] * 5 + [_i] * 3 + [_p, _i, _f, _f, _i, _i] + [_p] * 9 + [_i, _p, _i, _p, _p]),
    'infgen_tokenize_agent': (_i, [_p] * 9 + [_i] * 10 + [_p] * 8),
    'infgen_distance_to_road_edge': (_i, [_p] * 9 + [_i] * 4 + [_p, _p, _p, _i, _f, _p, _p]),
    'infgen_window_log_likelihood': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p]),
    'infgen_placement_features': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    'infgen_match_map_tokens': (_i, [_p, _p, _p, _i, _i, _p, _p]),
    'infgen_match_agent_tokens': (_i, [_p, _p, _p, _p, _p, _p, C.c_longlong, _i, _i, _i, _i, _p, _p, _p]),
    'infgen_attn_pre': (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p]),
    'infgen_edge_attn': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'infgen_edge_attn_mode': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    'infgen_attn_post': (_i, [_p, _i, _p, _p, _p, _p, _i, _p]),
    'infgen_attn_post_pre': (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    'infgen_heads': (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _p]),
    'infgen_map_graph': (_i, [_i, _i, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _i, _p]),
    'infgen_build_edges': (_i, [C.POINTER(Rollout), _i, _i, _p]),
    'infgen_integrate': (_i, [C.POINTER(Rollout), _i, _p]),
    'infgen_raw_feature': (_i, [C.POINTER(Rollout), _i, _p]),
    'infgen_raw_feature_rows': (_i, [_p, _i, _p, _p, _i, _p]),
    'infgen_decode_layers': (_i, [C.POINTER(Rollout), _i, _i, _p]),
    'infgen_decode_step': (_i, [C.POINTER(Rollout), _i, _p]),
    'infgen_rollout_run': (_i, [C.POINTER(Rollout), _i, _i, _p]),
    'infgen_sample_topk': (_i, [_p, _i, _i, _i, _p, _p, _p]),
    'infgen_occupancy': (_i, [C.POINTER(Rollout), _i, _p, _p]),
    'infgen_occupancy_embed': (_i, [C.POINTER(Rollout), _i, _p, _p, _p, _p]),
    'infgen_point_edges': (_i, [C.POINTER(Rollout), _i, _p, _p, _i, _i, _f, _i, _f, _i, C.POINTER(EdgeBuf), C.POINTER(EdgeBuf), _p]),
    'infgen_insert_decide': (_i, [C.POINTER(Rollout), _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'infgen_insert_seed': (_i, [C.POINTER(Rollout), C.POINTER(Insertion), _i, _i, _i, _p, _p]),
    'infgen_insert_heading': (_i, [C.POINTER(Rollout), C.POINTER(Insertion), _i, _i, _i, _p]),
    'infgen_insert_decide_topk': (_i, [C.POINTER(Rollout), _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p]),
    'infgen_insert_finalize': (_i, [C.POINTER(Rollout), _i, _f, _p, _p, _p, _i, _p, _p, _p]),
    'infgen_prof_enable': (_i, [C.c_uint, _i]),
    'infgen_prof_collect': (_i, [C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
    'infgen_prof_collect_steps': (_i, [C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_ulonglong),
                                       C.POINTER(C.c_double), C.POINTER(_i)]),
    'infgen_prof_set_stride': (_i, [_i]),
    'infgen_prof_seen': (_i, [C.POINTER(_i), C.POINTER(_i)]),
}

Q_ATTN_PACK_SIZE, Q_FOURIER_N2, Q_FOURIER_N3, Q_FOURIER_N4, Q_TILE_ROWS, Q_EDGE_ATTN_CAP, Q_MAX_AGENTS, \
    Q_ABI_VERSION, Q_SIZEOF_ROLLOUT = range(9)

KERNEL_IDS = ['k_linear', 'k_fourier', 'k_attn_pre', 'k_edge_attn', 'k_attn_post', 'k_heads', 'k_build_edges',
              'k_integrate', 'k_rawfeat_prep', 'k_map_graph']

_lib: Optional[C.CDLL] = None


class InfgenHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libinfgen_hip.so; raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise InfgenHipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C infgen_amd/csrc` (there is no CPU fallback for the product path)')
    # torch first: its wheel bundles its own libamdhip64 (ROCm 7.0), libinfgen_hip.so is linked against the system's (/opt/rocm).
    # Whichever HIP runtime is mapped first serves both (same SONAME); with the library loaded BEFORE torch the process ends up with
    # two runtimes and the library's sees no device ("no ROCm-capable device is detected" from the first launch - found with
    # `python __graft_entry__.py smoke`, whose build() loads the library before smoke() imports torch)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.infgen_layout_query(Q_SIZEOF_ROLLOUT) != C.sizeof(Rollout):
        raise InfgenHipError('InfgenRollout layout mismatch between include/infgen_hip.h and infgen_amd/_lib.py: '
                             f'{lib.infgen_layout_query(Q_SIZEOF_ROLLOUT)} != {C.sizeof(Rollout)}')
    _lib = lib
    return lib


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = load().infgen_last_error()
        raise InfgenHipError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')


def ptr(t) -> Optional[int]:
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return t.data_ptr()


_tl = threading.local()


class thread_options:
    """context manager: the calling thread's option block for operator-level entries (infgen_thread_options, include/infgen_hip.h);
    nests (the outer block is restored on exit)"""

    def __init__(self, opts: 'Options'):
        self.opts, self.prev = opts, None

    def __enter__(self):
        lib = load()
        depth = getattr(_tl, 'depth', 0)
        if depth > 0:
            self.prev = Options()
            check(lib.infgen_get_effective_options(C.byref(self.prev)), 'infgen_get_effective_options')
        check(lib.infgen_thread_options(C.byref(self.opts)), 'infgen_thread_options')
        _tl.depth = depth + 1
        return self

    def __exit__(self, *exc):
        _tl.depth -= 1
        check(load().infgen_thread_options(C.byref(self.prev) if self.prev is not None else None), 'infgen_thread_options')
        return False


_prof_mask = 0


def prof_enable(mask: int, max_launches: int = 20000) -> None:
    global _prof_mask
    check(load().infgen_prof_enable(mask, max_launches), 'infgen_prof_enable')
    _prof_mask = int(mask)


def prof_set_stride(stride: int) -> None:
    """after prof_enable: only every stride-th launch inside decode steps carries an event pair (bench.py's timed region)"""
    check(load().infgen_prof_set_stride(int(stride)), 'infgen_prof_set_stride')


def prof_seen():
    """-> {kernel: dict(seen, seen_step)}: launches of the selected kernels since prof_enable, bracketed or not"""
    n = len(KERNEL_IDS)
    seen, seen_step = (_i * n)(), (_i * n)()
    check(load().infgen_prof_seen(seen, seen_step), 'infgen_prof_seen')
    return {k: dict(seen=seen[i], seen_step=seen_step[i]) for i, k in enumerate(KERNEL_IDS)}


def prof_active() -> bool:
    """per-kernel HIP-event profiling is on (bench.py's roofline legs): launches must be issued eagerly, not replayed"""
    return _prof_mask != 0


def prof_collect():
    """-> {kernel: dict(ms, calls, macs, step_ms, step_calls)} of the launches recorded since prof_enable; synchronises"""
    n = len(KERNEL_IDS)
    ms = (C.c_double * n)()
    calls = (_i * n)()
    macs = (C.c_double * n)()
    rows = (C.c_ulonglong * 16)()
    sms = (C.c_double * n)()
    scalls = (_i * n)()
    check(load().infgen_prof_collect_steps(ms, calls, macs, rows, sms, scalls), 'infgen_prof_collect_steps')
    # step_ms / step_calls: the launches issued inside decode steps (the rest: prologue and operator-level calls)
    out = {k: dict(ms=ms[i], calls=calls[i], macs=macs[i], step_ms=sms[i], step_calls=scalls[i]) for i, k in enumerate(KERNEL_IDS)}
    # edges built per set; every decode step's sets are consumed by one edge-attention launch per layer
    out['k_edge_attn']['edges_built'] = dict(temporal=int(rows[8]), map=int(rows[9]), agent=int(rows[10]))
    # FourierEmbedding: n x (129x128 + 128x128) + 128x128 MACs per row (reference layers.py:126-141)
    out['k_fourier']['macs'] = float(sum(rows[nd] * (nd * 32896 + 16384) for nd in range(8)))
    out['k_fourier']['rows'] = {nd: int(rows[nd]) for nd in range(8) if rows[nd]}
    return out
