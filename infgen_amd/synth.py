"""Synthetic Waymo-shaped scenes, formula vocabularies and closed-form weights.

Everything here is *input generation*: it is fed identically to the reference
(golden generation, this container only), to the CPU oracle and to the HIP
path, so none of it has to agree with anything in the reference bit-for-bit —
only the schema does.  Schema = the ``data`` dict read by
``InfGenAgentDecoder.inference`` (reference infgen/modules/agent_decoder.py:1609-1628,
1648-1650) and ``InfGenMapDecoder.forward`` (infgen/modules/map_decoder.py:71-93),
produced in the reference by ``TokenProcessor._tokenize_agent``
(infgen/datasets/preprocess.py:534-548) and ``InfGen.match_token_map`` /
``_fetch_enterings`` (infgen/model/infgen.py:972-979, 1083-1092).

No torch RNG is used: numpy ``default_rng(seed)`` (PCG64) so that every side
sees the same numbers.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

# reference infgen/datasets/preprocess.py:13-20
SHIFT = 5
AGENT_SHAPE = {
    'vehicle': [4.3, 1.8, 1.0],
    'pedstrain': [0.5, 0.5, 1.0],
    'cyclist': [1.9, 0.5, 1.0],
}
AGENT_TYPE = ['veh', 'ped', 'cyc', 'seed']
AGENT_STATE = ['invalid', 'valid', 'enter', 'exit']
INVALID, VALID, ENTER, EXIT = 0, 1, 2, 3


@dataclass
class RolloutConfig:
    """Hyper-parameters of the path (reference configs/ours_standard.yaml:2-21,55-85)."""
    input_dim: int = 2
    hidden_dim: int = 128
    num_heads: int = 8
    head_dim: int = 16
    num_freq_bands: int = 64
    num_map_layers: int = 3
    num_agent_layers: int = 6
    num_historical_steps: int = 11
    token_size: int = 2048
    a2a_radius: float = 60.0
    pl2a_radius: float = 30.0
    pl2pl_radius: float = 10.0
    a2sa_radius: float = 10.0
    pl2sa_radius: float = 10.0
    pl2seed_radius: float = 75.0
    time_span: int = 60
    grid_range: float = 150.0
    grid_interval: float = 3.0
    angle_interval: float = 3.0
    seed_size: int = 1
    buffer_size: int = 128
    num_recurrent_steps_val: int = 80
    disable_insertion: bool = True
    state_token: Dict[str, int] = field(default_factory=lambda: dict(invalid=0, valid=1, enter=2, exit=3))

    @property
    def shift(self) -> int:
        return SHIFT

    @property
    def window(self) -> int:
        # reference agent_decoder.py:586-587  (time_span / shift)
        return int(self.time_span // SHIFT)

    @property
    def num_columns(self) -> int:
        # reference agent_decoder.py:1637
        return (self.num_recurrent_steps_val + self.num_historical_steps) // SHIFT

    @property
    def num_decode_steps(self) -> int:
        return self.num_recurrent_steps_val // SHIFT

    @property
    def hist_columns(self) -> int:
        # (num_historical_steps - 1) // shift == 2
        return (self.num_historical_steps - 1) // SHIFT


def smart_config(**kw) -> RolloutConfig:
    """BASELINE config C1: configs/smart.yaml model keys (time_span 30) + the keys it
    lacks taken from ours_standard.yaml."""
    d = dict(time_span=30, num_recurrent_steps_val=50, disable_insertion=True)
    d.update(kw)
    return RolloutConfig(**d)


def standard_config(**kw) -> RolloutConfig:
    """BASELINE configs C2/C3: configs/ours_standard.yaml."""
    d = dict(time_span=60, num_recurrent_steps_val=80)
    d.update(kw)
    return RolloutConfig(**d)


# --------------------------------------------------------------------------------------
# attribute grid (numpy replica of Attr_Tokenizer._prepare_grid, attr_tokenizer.py:24-43)
# --------------------------------------------------------------------------------------

def build_grid(grid_range: float = 150.0, grid_interval: float = 3.0, radius: float = 75.0) -> np.ndarray:
    num_grid = int(grid_range / grid_interval) + 1
    x = np.arange(num_grid, dtype=np.float32)
    gx, gy = np.meshgrid(x, x, indexing='xy')
    grid = np.stack([gx.reshape(-1), gy.reshape(-1)], axis=-1)
    grid = grid.reshape(num_grid, num_grid, 2)[::-1].reshape(-1, 2)
    grid = (grid - np.float32(num_grid // 2)) * np.float32(grid_interval)
    dist = np.sqrt((grid.astype(np.float32) ** 2).sum(-1, dtype=np.float32))
    mask = (dist <= np.float32(radius))
    return np.ascontiguousarray(grid[mask].astype(np.float32))


def _rot_right(x: np.ndarray, theta: np.ndarray) -> np.ndarray:
    """x @ [[cos, sin], [-sin, cos]] (reference attr_tokenizer.py:45-55), float32."""
    c, s = np.cos(theta, dtype=np.float32), np.sin(theta, dtype=np.float32)
    out = np.empty_like(x)
    out[..., 0] = x[..., 0] * c - x[..., 1] * s
    out[..., 1] = x[..., 0] * s + x[..., 1] * c
    return out


def encode_pos_np(grid: np.ndarray, x: np.ndarray, y: np.ndarray, theta_y: np.ndarray) -> np.ndarray:
    """numpy stand-in for Attr_Tokenizer.encode_pos (attr_tokenizer.py:77-89); used only to
    generate *input* history grid tokens."""
    cx = (x - y).astype(np.float32)
    ang = (-(theta_y - np.float32(np.pi / 2))).astype(np.float32)
    cx = _rot_right(cx, ang)
    d = np.sqrt(((cx[:, None, :] - grid[None]) ** 2).sum(-1, dtype=np.float32))
    return d.argmin(-1).astype(np.int64)


# --------------------------------------------------------------------------------------
# vocabularies (formula tables with the shapes/dtypes of infgen/tokens/*.pkl)
# --------------------------------------------------------------------------------------

def make_agent_vocab(token_size: int = 2048) -> Dict[str, np.ndarray]:
    """(token_size, 6, 4, 2) float32 contour templates per type: a box swept along a
    constant speed / constant yaw-rate arc for 0.5 s, corners ordered
    front-left, front-right, rear-right, rear-left like agent_vocab_555_s2.pkl."""
    out = {}
    nv = 64
    nw = token_size // nv
    for name, key, vmax, wmax in (('veh', 'vehicle', 12.0, 0.9), ('ped', 'pedstrain', 2.0, 1.5),
                                  ('cyc', 'cyclist', 6.0, 1.2)):
        length, width, _ = AGENT_SHAPE[key]
        k = np.arange(token_size)
        # displacement over 0.5 s and heading change over 0.5 s
        disp = (k // nw).astype(np.float64) / (nv - 1) * vmax - 0.05 * vmax
        dth = ((k % nw).astype(np.float64) / (nw - 1) - 0.5) * 2.0 * wmax
        tab = np.zeros((token_size, 6, 4, 2), dtype=np.float64)
        corners = np.array([[length / 2, width / 2], [length / 2, -width / 2],
                            [-length / 2, -width / 2], [-length / 2, width / 2]])
        for s in range(6):
            f = s / 5.0
            th = dth * f
            # arc integration of a unicycle
            with np.errstate(divide='ignore', invalid='ignore'):
                px = np.where(np.abs(dth) < 1e-9, disp * f, disp * np.sin(th) / np.where(dth == 0, 1, dth))
                py = np.where(np.abs(dth) < 1e-9, 0.0, disp * (1 - np.cos(th)) / np.where(dth == 0, 1, dth))
            c, sn = np.cos(th), np.sin(th)
            for j in range(4):
                tab[:, s, j, 0] = px + corners[j, 0] * c - corners[j, 1] * sn
                tab[:, s, j, 1] = py + corners[j, 0] * sn + corners[j, 1] * c
        out[name] = tab.astype(np.float32)
    return out


def make_map_vocab(num_tokens: int = 1024) -> np.ndarray:
    """(num_tokens, 11, 2) float32 polyline templates like map_traj_token5.pkl['traj_src']."""
    k = np.arange(num_tokens)
    curv = ((k % 32) / 31.0 - 0.5) * 0.4
    length = 2.0 + (k // 32) / 31.0 * 3.5
    s = np.linspace(0, 1, 11)[None, :] * length[:, None]
    th = curv[:, None] * s
    x = np.where(np.abs(curv[:, None]) < 1e-9, s, np.sin(th) / np.where(curv[:, None] == 0, 1, curv[:, None]))
    y = np.where(np.abs(curv[:, None]) < 1e-9, 0 * s, (1 - np.cos(th)) / np.where(curv[:, None] == 0, 1, curv[:, None]))
    return np.stack([x, y], -1).astype(np.float32)


# --------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------

def scene_seed(config_id: int, scene_idx: int) -> int:
    return 1000 * config_id + scene_idx


def make_scene(seed: int, num_agents: int, num_map: int, cfg: RolloutConfig,
               half_extent: float = 60.0, ego_last: bool = True,
               edge_cases: bool = False, vocab: Optional[Dict[str, np.ndarray]] = None,
               grid: Optional[np.ndarray] = None, slip: float = 0.0) -> Dict[str, Dict[str, np.ndarray]]:
    """One synthetic scene in the reference's input schema (numpy).

    ``slip`` (radians) turns each agent's velocity away from its heading by a per-agent angle drawn
    from ±[slip/2, slip].  With ``slip == 0`` every agent moves exactly along its heading, so the
    relative-position angle of its temporal edges (agent_decoder.py:600-604) is ±pi up to rounding
    noise — the sign then depends on the last ulp of cos/sin(heading), which no two libm agree on
    (DESIGN.md "parity caveats").  Parity tests at scale use slip > 0; the committed golden
    fixtures were generated with slip == 0 and keep their seeds.

    ``edge_cases`` injects: an agent entering at column 1, an agent that exits at column 1,
    a row with ``valid_mask[:, 10] == False`` and (if ego_last) a row filtered out before
    the ego (state invalid at column 1).
    """
    rng = np.random.default_rng(seed)
    A, M, L = num_agents, num_map, half_extent
    T0 = min(18, cfg.num_columns)
    if vocab is None:
        vocab = make_agent_vocab(cfg.token_size)
    if grid is None:
        grid = build_grid(cfg.grid_range, cfg.grid_interval, cfg.pl2seed_radius)
    n_raw = 91

    pos0 = rng.uniform(-L, L, size=(A, 2))
    head0 = rng.uniform(-np.pi, np.pi, size=(A,))
    speed = rng.uniform(0.0, 10.0, size=(A,))
    atype = rng.integers(0, 3, size=(A,))
    av = A - 1 if ego_last else 0
    pos0[av] = 0.0
    head0[av] = rng.uniform(-0.3, 0.3)
    atype[av] = 0
    speed[atype == 1] *= 0.2
    vdir = head0
    if slip > 0.0:
        srng = np.random.default_rng([seed, 977])
        vdir = head0 + srng.uniform(0.5 * slip, slip, size=(A,)) * srng.choice([-1.0, 1.0], size=(A,))
    vel = np.stack([np.cos(vdir), np.sin(vdir)], -1) * speed[:, None]

    col = np.arange(T0)
    token_pos = (pos0[:, None, :] + vel[:, None, :] * (0.5 * col)[None, :, None]).astype(np.float32)
    token_heading = np.repeat(head0[:, None], T0, 1).astype(np.float32)
    state = np.full((A, T0), VALID, dtype=np.int64)
    state[:, 0] = ENTER
    token_idx = rng.integers(0, cfg.token_size, size=(A, T0)).astype(np.int64)
    raw_valid = np.ones((A, T0), dtype=bool)
    raw_valid[:, 0] = False
    valid_mask = np.ones((A, n_raw), dtype=bool)

    if edge_cases and A >= 8:
        cand = [i for i in range(A) if i != av]
        a_plain, a_enter1, a_exit1, a_novalid = cand[0], cand[1], cand[2], cand[3]
        state[a_plain, 0] = VALID                      # no bos inside the window
        raw_valid[a_plain, 0] = True
        state[a_enter1, 0] = INVALID                   # enters at the current column
        state[a_enter1, 1] = ENTER
        raw_valid[a_enter1, :2] = False
        state[a_exit1, 1] = EXIT
        valid_mask[a_novalid, 10] = False
        if ego_last:
            a_filtered = cand[4]
            state[a_filtered, 1] = INVALID             # dropped by filter_mask (agent_decoder.py:1609)
            state[a_filtered, 0] = INVALID
            raw_valid[a_filtered, :2] = False

    inv = state == INVALID
    token_idx[state == ENTER] = -2
    token_idx[inv] = -1
    token_pos[inv] = 0.0
    token_heading[inv] = 0.0

    grid_idx = np.full((A, T0), -1, dtype=np.int64)
    for j in range(T0):
        g = encode_pos_np(grid, token_pos[:, j], np.repeat(token_pos[av:av + 1, j], A, 0),
                          np.repeat(token_heading[av:av + 1, j], A, 0))
        grid_idx[:, j] = np.where(inv[:, j], -1, g)

    shape = np.zeros((A, n_raw, 3), dtype=np.float32)
    for i, key in enumerate(('vehicle', 'pedstrain', 'cyclist')):
        shape[atype == i] = np.asarray(AGENT_SHAPE[key], dtype=np.float32)
    steps = np.arange(n_raw)
    position = np.zeros((A, n_raw, 3), dtype=np.float32)
    position[..., :2] = pos0[:, None, :] + vel[:, None, :] * (0.1 * steps)[None, :, None]
    heading = np.repeat(head0[:, None], n_raw, 1).astype(np.float32)

    type_names = ['veh', 'ped', 'cyc']
    agent = {
        'num_nodes': A,
        'av_index': np.array([av], dtype=np.int64),
        'id': np.arange(A, dtype=np.int64),
        'type': atype.astype(np.uint8),
        'state_idx': state,
        'token_idx': token_idx,
        'token_pos': token_pos,
        'token_heading': token_heading,
        'raw_agent_valid_mask': raw_valid,
        'agent_valid_mask': raw_valid.copy(),
        'valid_mask': valid_mask,
        'grid_token_idx': grid_idx,
        'shape': shape,
        'position': position,
        'heading': heading,
        'category': np.full((A,), 2, dtype=np.uint8),
        'trajectory_token_veh': vocab['veh'],
        'trajectory_token_ped': vocab['ped'],
        'trajectory_token_cyc': vocab['cyc'],
    }

    npoly = max(1, M // 8)
    pt = {
        'num_nodes': M,
        'position': np.concatenate([rng.uniform(-L, L, size=(M, 2)), np.zeros((M, 1))], -1).astype(np.float32),
        'orientation': rng.uniform(-np.pi, np.pi, size=(M,)).astype(np.float32),
        'type': rng.integers(0, 17, size=(M,)).astype(np.uint8),
        'pl_type': rng.integers(0, 4, size=(M,)).astype(np.uint8),
        'token_idx': rng.integers(0, 1024, size=(M,)).astype(np.int64),
        'pt_valid_mask': np.ones((M,), dtype=bool),
        'pt_pred_mask': np.zeros((M,), dtype=bool),
        'pt_target_mask': np.zeros((M,), dtype=bool),
        'batch': np.zeros((M,), dtype=np.int64),
    }
    poly = {'num_nodes': npoly, 'light_type': rng.integers(0, 4, size=(npoly,)).astype(np.uint8)}
    tok2pl = np.stack([np.arange(M), rng.integers(0, npoly, size=(M,))]).astype(np.int64)
    return {
        'agent': agent,
        'pt_token': pt,
        'map_polygon': poly,
        'pt_token__to__map_polygon': {'edge_index': tok2pl},
        'batch_size_a': np.array([A], dtype=np.int64),
        'batch_size_pl': np.array([M], dtype=np.int64),
        'scenario_id': ['synth_%d' % seed],
        'num_graphs': 1,
    }


# --------------------------------------------------------------------------------------
# closed-form weights
# --------------------------------------------------------------------------------------

def fill_state_dict(shapes: Dict[str, tuple], seed: int = 0, rich: bool = True,
                    head_gain: float = 1.0, freq_std: float = 0.02) -> Dict[str, np.ndarray]:
    """Deterministic weights keyed by tensor name (order independent).

    Distributions follow ``weight_init`` (reference infgen/utils/func.py:177-196): Linear
    xavier-uniform / zero bias, Embedding N(0, 0.02), LayerNorm 1 / 0.  With ``rich`` the
    biases and LayerNorm affine parameters are perturbed so that a missing bias / gamma /
    beta shows up in parity tests (a trained checkpoint has them non-trivial too).
    ``head_gain`` scales ``token_predict_head.mlp.3.weight`` (sharpened logits for
    free-running token parity).
    """
    out = {}
    for name, shape in shapes.items():
        # non-bipartite AttentionLayers register ONE LayerNorm under two names
        # (reference infgen/modules/layers.py:52-53): both keys must carry the same values
        key = name
        if any(t in name for t in _SHARED_PRENORM) and '.attn_prenorm_x_dst.' in name:
            key = name.replace('.attn_prenorm_x_dst.', '.attn_prenorm_x_src.')
        rng = np.random.default_rng((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.split('.')[-1]
        is_ln = _is_layernorm(name, shapes)
        if name.endswith('attr_tokenizer.grid') or leaf in ('grid', 'dist', 'dir'):
            continue
        if len(shape) == 2 and leaf == 'weight' and not is_ln:
            if _is_embedding(name):
                std = freq_std if name.endswith('freqs.weight') else 0.02
                w = rng.normal(0.0, std, size=shape)
            else:
                fan_out, fan_in = shape
                bound = np.sqrt(6.0 / (fan_in + fan_out))
                w = rng.uniform(-bound, bound, size=shape)
                if name.endswith('token_predict_head.mlp.3.weight') and 'map_encoder' not in name:
                    w = w * head_gain
        elif is_ln and leaf == 'weight':
            w = 1.0 + (rng.uniform(-0.2, 0.2, size=shape) if rich else 0.0)
        elif is_ln and leaf == 'bias':
            w = rng.uniform(-0.1, 0.1, size=shape) if rich else np.zeros(shape)
        elif leaf == 'bias':
            w = rng.uniform(-0.05, 0.05, size=shape) if rich else np.zeros(shape)
        else:
            w = rng.normal(0.0, 0.02, size=shape)
        out[name] = np.asarray(w, dtype=np.float32).reshape(shape)
    return out


_SHARED_PRENORM = ('.t_attn_layers.', '.a2a_attn_layers.', '.a2sa_attn_layers.', '.pt2pt_layers.')
_EMB_LEAVES = ('type_a_emb', 'state_a_emb', 'no_token_emb', 'bos_token_emb', 'invalid_offset_token_emb',
               'type_pt_emb', 'side_pt_emb', 'polygon_type_emb', 'light_pl_emb', 'freqs')


def _is_embedding(name: str) -> bool:
    parts = name.split('.')
    return len(parts) >= 2 and parts[-2] in _EMB_LEAVES


def _is_layernorm(name: str, shapes: Dict[str, tuple]) -> bool:
    """LayerNorm params are 1-D 'weight'/'bias' pairs whose 'weight' is 1-D."""
    base = name.rsplit('.', 1)[0]
    w = shapes.get(base + '.weight')
    return w is not None and len(w) == 1
