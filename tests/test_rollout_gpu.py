"""GPU: the HIP rollout (through the C ABI) against the golden fixtures produced by the
reference's own modules, and against the CPU oracle on fresh seeded scenes.

Bars (BASELINE.json north_star): greedy token indices bit-exact; logits within 1e-3 (fp32).
Fixtures with a sharpened token head (``head_gain`` 64) scale the logits — and their error —
by the gain; until round 5 their tolerance was 1e-3 * gain / 16.  Since the round-to-nearest operand
split (round 6) the flat 1e-3 holds on them too (measured 9e-5 - 3.2e-4, the reference's own fp32
noise at that gain), and that is what is asserted."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_case, make_weights

pytestmark = pytest.mark.gpu


def _engine(case, teacher=None, scenes=None, **kw):
    from infgen_amd import engine
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(case['sd'], case['cfg'], dev)
    eng = engine.RolloutEngine(w, scenes or [case['scene']], case['vocab'], case['map_vocab'], case['grid'],
                               store_logits=True, live_state=case['meta']['live_state'], teacher=teacher, **kw)
    eng.prologue()
    eng.edge_counts = []
    if eng.insertion:
        eng.run()
    else:
        for t in range(case['cfg'].num_decode_steps):      # step by step so that the per-step edge totals can be read
            eng.step(t)
            eng.edge_counts.append(eng.edge_totals())
    return eng, eng.outputs()


@pytest.fixture(params=[(2, 1), (1, 1), (0, 1), (2, 2), (1, 2), (0, 0)],
                ids=['by-size', 'split16', 'fp32mfma', 'by-size+edge-tile', 'split16+edge-tile', 'fp32mfma+unfused'])
def attn_mode(request):
    """node-side attention kernels: chosen by row count (default), forced fp16 three-term split, forced fp32 MFMA;
    edge side: k_edge_fused from 257 rows (default: single-scene fixtures take the unfused kernels), always, never"""
    from infgen_amd import _lib
    lib = _lib.load()
    _lib.check(lib.infgen_set_attn_mode(request.param[0]))
    _lib.check(lib.infgen_set_edge_fuse(request.param[1]))
    yield request.param
    _lib.check(lib.infgen_set_attn_mode(2))
    _lib.check(lib.infgen_set_edge_fuse(1))


# Fixtures whose smallest top-1 / top-2 logit margin is far enough above the kernels' error that a free-running rollout must
# reproduce every token (the expectation is explicit: a fixture that drifts out of this set fails instead of being checked
# weakly).  c2_a32_m512 has the reference's unsharpened head (margin 1e-4, the size of fp32 noise): first-step logits free
# running, everything else teacher-forced.
STRICT_CASES = {'c1_a8_m128': True, 'a24_m256_edge': True, 'a16_m128_egofirst_state': True, 'c2_a32_m512': False,
                'c3_a64_m1024': True}      # BASELINE C3's scene shape (64 agents, 1024 map tokens, R = 80), free-running


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_free_running_rollout_matches_reference_fixture(name, attn_mode):
    c = load_case(name)
    z, m = c['z'], c['meta']
    eng, outs = _engine(c)
    o = outs[0]
    assert np.abs(o['x_pt'] - z['x_pt']).max() <= 1e-4, 'map encoder'
    gain = max(1.0, m['head_gain'])
    tol = 1e-3
    steps = z['logits'].shape[0]
    assert (z['margin'].min() > tol) == STRICT_CASES[name], 'fixture changed class: regenerate STRICT_CASES deliberately'
    # per-step edge totals of the device's edge-set builder against the reference's own edge lists (a6 - a8)
    assert np.array_equal(np.asarray(eng.edge_counts), z['edge_count'][:, [0, 2, 1]]), 'temporal / map / agent edge totals'
    if STRICT_CASES[name]:
        assert np.array_equal(o['next_token_idx'], z['next_token_idx']), 'greedy tokens must be bit-exact'
        assert np.array_equal(o['next_state_idx'], z['next_state_idx'])
        assert np.abs(o['logits'][:steps] - z['logits']).max() <= tol
        if 'logit_max' in z.files:           # (fixtures that keep the full logits of their first steps only: per-row maxima of all)
            assert np.abs(o['logits'].max(-1) - z['logit_max']).max() <= tol
            assert np.array_equal(o['logits'].argmax(-1), z['logit_argmax'])
        assert np.abs(o['pos_a'] - z['pos_a']).max() <= 1e-3
        assert np.abs(o['head_a'] - z['head_a']).max() <= 1e-4
        assert np.abs(o['pred_traj'] - z['pred_traj']).max() <= 1e-3
        assert np.abs(o['pred_head'] - z['pred_head']).max() <= 1e-4
        assert np.array_equal(o['pred_state'], z['pred_state'])
        assert np.array_equal(o['pred_valid'], z['pred_valid'])
    else:
        # tiny argmax margins (unsharpened head): a flip is legitimate fp32 noise; compare the first
        # step free-running and everything else teacher-forced (next test)
        assert np.abs(o['logits'][0] - z['logits'][0]).max() <= tol
    assert np.array_equal(o['agent_id'], z['agent_id'])
    assert o['ego_index'] == int(z['ego_index'])


@pytest.mark.parametrize('name,copies', [('c1_a8_m128', 1), ('c1_a8_m128', 336), ('a24_m256_edge', 1), ('a24_m256_edge', 336),
                                         ('c3_a64_m1024', 1), ('c3_a64_m1024', 72)],
                         ids=['c1-single', 'c1-336 copies', 'a24-single', 'a24-336 copies', 'c3-single', 'c3-72 copies (4,608 rows)'])
def test_edge_lists_and_triple_outputs_match_the_reference(name, copies):
    """below the logits level: after every decode step the device's three edge sets (k_build_edges: CSR by
    destination row) decoded to sorted (step, destination agent, source column | agent | map token) triples are the REFERENCE's
    own edge lists (agent_decoder.py:540-758, hooked by tests/golden/make_golden_internals.py), and the residual stream after the
    first and the last (temporal, map, agent) triple of steps 0..2 (InfgenRollout.tap_x) equals the outputs of the reference's
    a2a_attn_layers[0] / [L-1] (:2133-2158) within 1e-4.  Checked for the first and the last scene of the batch; the batches of copies
    take the big-launch kernels (k_edge_fused3 etc.), c3 is BASELINE C3's scene shape (64 agents, 1024 map tokens: 26,726 agent edges)."""
    import os
    from conftest import GOLDEN
    from infgen_amd import engine
    c = load_case(name)
    zi = np.load(os.path.join(GOLDEN, name + '_internals.npz'))
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    eng = engine.RolloutEngine(w, [c['scene']] * copies, c['vocab'], c['map_vocab'], c['grid'], store_logits=False,
                               live_state=c['meta']['live_state'], tap_layers=True, use_graph=False)
    eng.prologue()
    A, A_cap, M_cap, rows, ring, L = eng.hosts[0]['A'], eng.A_cap, eng.M_cap, eng.rows, eng.ring, cfg.num_agent_layers
    n_act = zi['act_first'].shape[0]
    scenes = sorted({0, copies - 1})
    got = {s: {k: [] for k in 'tma'} for s in scenes}
    for t in range(cfg.num_decode_steps):
        eng.step(t)
        torch.cuda.synchronize()
        col = cfg.hist_columns - 1 + t
        for kind in 'tma':
            e = eng.edges[kind]
            off, cnt, src = (e[k].cpu().numpy() for k in ('off', 'cnt', 'src'))
            for s in scenes:
                for a in range(A_cap):
                    r = s * A_cap + a
                    if cnt[r] == 0:
                        continue
                    assert a < A, (kind, t, s, a)
                    ss = src[off[r]:off[r] + cnt[r]].astype(np.int64)
                    if kind == 't':                   # src = (column % ring) * rows + row, column in [col - 12, col - 1]
                        assert np.all(ss % rows == r)
                        slot = ss // rows
                        j = col - ((col - slot) % ring)
                        assert np.all((j >= col - (ring - 1)) & (j < col))
                        val = j
                    elif kind == 'm':                 # src = scene * M_cap + token
                        assert np.all(ss // M_cap == s)
                        val = ss % M_cap
                    else:                             # src = scene * A_cap + agent
                        assert np.all(ss // A_cap == s)
                        val = ss % A_cap
                    got[s][kind] += [(t, a, int(v)) for v in val]
        if t < n_act:
            tap = eng.tap_x.cpu().numpy()
            for s in scenes:
                x0, x1 = tap[0, s * A_cap:s * A_cap + A], tap[L - 1, s * A_cap:s * A_cap + A]
                assert np.abs(x0 - zi['act_first'][t]).max() <= 1e-4, (t, s, 'first triple')
                assert np.abs(x1 - zi['act_last'][t]).max() <= 1e-4, (t, s, 'last triple')
    for s in scenes:
        for kind, key in (('t', 'edges_t'), ('a', 'edges_a'), ('m', 'edges_m')):
            mine = np.asarray(sorted(got[s][kind]), np.int32).reshape(-1, 3)
            ref = zi[key]
            ref = ref[np.lexsort((ref[:, 2], ref[:, 1], ref[:, 0]))]
            assert np.array_equal(mine, ref), (s, kind, len(mine), len(ref))
    assert np.array_equal(eng.outputs()[0]['next_token_idx'], zi['next_token_idx'])


@pytest.mark.parametrize('name', ['c2_a32_m512', 'a16_m128_egofirst_state', 'a24_m256_edge'])
def test_teacher_forced_logits(name, attn_mode):
    """feed the reference's tokens/states back in: every step's logits within 1e-3 (fp32)"""
    c = load_case(name)
    z, m = c['z'], c['meta']
    eng, outs = _engine(c, teacher=[(z['next_token_idx'], z['next_state_idx'])])
    o = outs[0]
    tol = 1e-3
    assert np.abs(o['logits'] - z['logits']).max() <= tol
    assert np.abs(o['pos_a'] - z['pos_a']).max() <= 1e-3
    # argmax agreement wherever the reference's own margin exceeds the tolerance
    ok = z['margin'] > 4 * tol
    mine = o['logits'].argmax(-1)
    ref = z['logits'].argmax(-1)
    assert np.array_equal(mine[ok], ref[ok])


def test_batched_scenes_equal_single_scene_runs():
    """scenes are independent units: a batch must reproduce each scene run alone,
    bit for bit in tokens and to round-off in poses, with ragged agent/map counts."""
    from infgen_amd import synth
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    scenes = [synth.make_scene(7000 + i, a, m, cfg, ego_last=(i % 2 == 0), edge_cases=(i == 1), vocab=c['vocab'],
                               grid=c['grid']) for i, (a, m) in enumerate([(24, 256), (9, 100), (40, 300)])]
    engb, outb = _engine(c, scenes=scenes)
    for i, sc in enumerate(scenes):
        _, outs = _engine(c, scenes=[sc])
        assert np.array_equal(outs[0]['next_token_idx'], outb[i]['next_token_idx'])
        assert np.abs(outs[0]['pos_a'] - outb[i]['pos_a']).max() <= 1e-5
        assert np.abs(outs[0]['logits'] - outb[i]['logits']).max() <= 1e-4


@pytest.mark.parametrize('wseed,sseed', [(14, 4251), (16, 4253)])
def test_rollout_vs_oracle_fresh_seed(wseed, sseed):
    """scene / weight seeds that have no committed fixture: HIP vs the CPU oracle run in-process.  The seeds were picked so that
    the oracle's own arg-max margin clears the bar at every (step, row) - 0.081 / 0.060 against 4 x 1e-3 - which is asserted,
    so the token comparison below is unconditional over all 16 free-running steps (it used to sit behind an
    `if` that a seed with one near-tie silently skipped)"""
    from infgen_amd import synth
    from oracle import rollout_oracle as ro
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    sd = make_weights(seed=wseed, head_gain=64.0)
    scene = synth.make_scene(sseed, 20, 200, cfg, ego_last=True, edge_cases=True, vocab=c['vocab'], grid=c['grid'])
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = ro.run_scene(tsd, scene, cfg, c['vocab'], c['map_vocab'], c['grid'])
    case = dict(c, sd=sd, scene=scene)
    _, outs = _engine(case)
    o = outs[0]
    lg = ref['logits'].numpy()
    part = np.partition(lg, -2, axis=-1)
    margin = part[..., -1] - part[..., -2]
    tol = 1e-3
    assert margin.min() > 4 * tol, f'seed no longer has a clear margin: {margin.min()}'
    assert np.array_equal(o['next_token_idx'], ref['next_token_idx'].numpy())
    assert np.array_equal(o['next_state_idx'], ref['next_state_idx'].numpy())
    err = float(np.abs(o['logits'] - lg).max())
    print(f'fresh seeds ({wseed}, {sseed}): min arg-max margin {margin.min():.3f}, max logits error over all steps {err:.2e}')
    assert err <= tol, err


def test_batch_of_fresh_scenes_through_the_big_launch_kernels_vs_oracle():
    """24 scenes no fixture knows - 8 to 64 agents, 128 to 1024 map tokens, ego first / last, history edge cases - as ONE batch of 192
    (8 copies each: 12,288 rows, i.e. the kernels of the headline batch: k_edge_fused3, k_attn_h, k_fourier_h(12), k_heads_h), every
    scene against its own CPU-oracle rollout (reference agent_decoder.py:1605-2389), step by step: as long as the tokens agreed so
    far the logits of a step must agree within 1e-3, and a token may differ ONLY in a row whose arg-max margin in the oracle is below
    4 x 1e-3 (a near-tie; the comparison of that rollout ends there).  The first and the last copy of every scene are checked; at
    least 40 of the 48 checked rollouts must agree over all 16 steps"""
    from infgen_amd import engine, synth
    from oracle import rollout_oracle as ro
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    sd = make_weights(seed=21, head_gain=64.0)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    A_of, M_of = (8, 11, 17, 24, 33, 40, 48, 64), (128, 300, 700, 1024)
    scenes = [synth.make_scene(7300 + i, A_of[i % 8], M_of[(i // 2) % 4], cfg, ego_last=bool(i % 2), edge_cases=bool((i // 3) % 2),
                               vocab=c['vocab'], grid=c['grid'], slip=0.2) for i in range(24)]
    copies = 8
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(sd, cfg, dev)
    eng = engine.RolloutEngine(w, [sc for sc in scenes for _ in range(copies)], c['vocab'], c['map_vocab'], c['grid'], store_logits=True,
                               use_graph=False)
    assert eng.rows > 10240
    eng.rollout()
    outs = eng.outputs()
    torch.set_num_threads(16)
    full, worst, steps_checked = 0, 0.0, 0
    hc = cfg.hist_columns
    for i, sc in enumerate(scenes):
        ref = ro.run_scene(tsd, sc, cfg, c['vocab'], c['map_vocab'], c['grid'])
        lg = ref['logits'].numpy()                                   # [steps][A][2048]
        part = np.partition(lg, -2, axis=-1)
        margin = part[..., -1] - part[..., -2]                       # [steps][A]
        rtok = ref['next_token_idx'].numpy()
        for o in (outs[i * copies], outs[i * copies + copies - 1]):
            done = lg.shape[0]
            for t in range(lg.shape[0]):
                # same inputs so far: the logits of step t must agree; a token may differ only where the oracle's own margin is a near-tie
                err = float(np.abs(o['logits'][t] - lg[t]).max())
                worst = max(worst, err)
                assert err <= 1e-3, (i, t, err)
                diff = o['next_token_idx'][:, hc + t] != rtok[:, hc + t]
                assert not (diff & (margin[t] >= 4e-3)).any(), (i, t, 'a token with a clear margin differs')
                steps_checked += 1
                if diff.any():                                       # (legitimate near-tie: the rollouts part ways here)
                    done = t
                    break
            full += done == lg.shape[0]
    print(f'fresh batch: {full} of 48 checked rollouts agree over all 16 steps, {steps_checked} steps compared, worst logits error {worst:.2e}')
    assert full >= 40 and steps_checked >= 700, (full, steps_checked)


@pytest.mark.parametrize('name', ['ins_forced_a16_m256', 'ins_natural_a20_m256', 'ins_sampled_a16_m256'])
def test_insertion_rollout_matches_reference_fixture(name):
    """scenario insertion (agent_decoder.py:1773-2105): same agents inserted at the same steps with the same
    cells / types / headings as the reference, tokens bit-exact, logits within tolerance"""
    from infgen_amd import engine
    c = load_case(name)
    z, m = c['z'], c['meta']
    cfg = c['cfg']
    cfg.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    sampled = m.get('insert_k', 1) > 1          # the fixture's uniforms replay the reference's top-10 cell draws (:1900-1909)
    eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True,
                               force_enter=(m['insertion'] == 'forced'), insert_k=m.get('insert_k', 1),
                               insert_uniforms=z['insert_uniforms'][:, :, None] if sampled else None, seed_outputs=True)
    eng.rollout()
    o = eng.outputs()[0]
    assert o['pos_a'].shape[0] == z['pos_a'].shape[0], (o['pos_a'].shape, z['pos_a'].shape)
    assert np.array_equal(o['next_state_idx'], z['next_state_idx'])
    assert np.array_equal(o['next_token_idx'], z['next_token_idx'])
    assert np.array_equal(o['agent_id'], z['agent_id'])
    assert np.array_equal(o['pred_type'], z['pred_type'])
    assert np.abs(o['pred_shape'] - z['pred_shape']).max() <= 1e-4
    tol = 1e-3
    for i, n in enumerate(z['n_agents_step']):
        assert np.abs(o['logits'][i, :n] - z['logits'][i, :n]).max() <= tol, i
    assert np.abs(o['pos_a'] - z['pos_a']).max() <= 1e-3
    assert np.abs(o['head_a'] - z['head_a']).max() <= 1e-4
    assert np.abs(o['pred_traj'] - z['pred_traj']).max() <= 1e-3
    assert np.abs(o['pred_head'] - z['pred_head']).max() <= 1e-4
    assert np.array_equal(o['pred_state'], z['pred_state'])
    # the seed node's per-insertion outputs of the return dict (agent_decoder.py:2099-2113, :2364-2386) and the labels (:1996-1999)
    assert np.array_equal(o['next_state_prob_seed'] > 0, z['seed_state_prob'] > 0)
    assert np.abs(o['next_state_prob_seed'] - z['seed_state_prob']).max() <= 1e-4
    assert np.abs(o['next_pos_rel_prob_seed'] - z['seed_pos_prob']).max() <= 1e-4
    assert np.abs(o['grid_agent_occ_seed'] - z['seed_occ_a']).max() <= 1e-3
    assert np.abs(o['grid_pt_occ_seed'] - z['seed_occ_p']).max() <= 1e-3
    assert np.array_equal(o['grid_agent_occ_gt_seed'].astype(np.int8), z['seed_occ_gt'])
    lab = np.asarray([[int(l[1:]) if l else 0 for l in row] for row in o['agent_labels']], np.int16)
    assert np.array_equal(lab, z['agent_label_k'])
    # a second rollout of the same engine (state reset) reproduces the first
    eng.rollout()
    o2 = eng.outputs()[0]
    assert np.array_equal(o2['next_token_idx'], o['next_token_idx'])
    assert np.array_equal(o2['next_state_prob_seed'], o['next_state_prob_seed'])


@pytest.mark.parametrize('name', ['ins_forced_a16_m256', 'ins_natural_a20_m256'])
def test_insertion_fixture_as_a_big_batch(name):
    """the insertion fixtures as 128 copies in one batch (128 x 96 rows = 12,288: the big-launch kernels - k_edge_fused3 and k_attn_h
    over the 16-row GROUP LISTS of the padded layout, the row limits of the edge kernel): the first, a middle and the last copy
    insert the reference's agents at the reference's steps (agent_decoder.py:1773-2105) with its tokens, states, ids and types, poses
    within 1e-3, and a second rollout is bitwise the first"""
    from infgen_amd import engine
    c = load_case(name)
    z, m = c['z'], c['meta']
    cfg = c['cfg']
    cfg.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    copies = 128
    eng = engine.RolloutEngine(w, [c['scene']] * copies, c['vocab'], c['map_vocab'], c['grid'], store_logits=False,
                               force_enter=(m['insertion'] == 'forced'), a_cap=96)
    assert eng.rows > 10240
    eng.rollout()
    outs = eng.outputs()
    tok0 = eng.token.clone()
    for i in (0, copies // 2, copies - 1):
        o = outs[i]
        assert o['pos_a'].shape[0] == z['pos_a'].shape[0], (i, o['pos_a'].shape, z['pos_a'].shape)
        assert np.array_equal(o['next_state_idx'], z['next_state_idx']), i
        assert np.array_equal(o['next_token_idx'], z['next_token_idx']), i
        assert np.array_equal(o['agent_id'], z['agent_id']) and np.array_equal(o['pred_type'], z['pred_type']), i
        assert np.abs(o['pos_a'] - z['pos_a']).max() <= 1e-3 and np.abs(o['head_a'] - z['head_a']).max() <= 1e-4, i
        assert np.array_equal(o['pred_state'], z['pred_state']), i
    eng.rollout()
    assert torch.equal(tok0, eng.token)


def _first_ill_conditioned_step(z, grid, ego, hist=2):
    """first decode step whose input column is ill-conditioned in the reference itself, so that everything downstream may
    legitimately differ between two correct implementations (the reference's CPU and GPU builds disagree there as well):
    (a) a row moving exactly against its heading - angle_between_2d_vectors is +pi or -pi by the last ulp of cos / sin
    (DESIGN.md "parity caveat"); (b) a row whose position sits on the border of two cells of the ego-centric grid - the
    arg-min of encode_pos (attr_tokenizer.py:77-89) flips with the last bits of the pose, and with it the row's grid
    embedding"""
    pos, head, st = z['pos_a'], z['head_a'], z['next_state_idx']
    for col in range(hist, pos.shape[1] - 1):
        live = st[:, col] != 0
        mv = pos[:, col] - pos[:, col - 1]
        hv = np.stack([np.cos(head[:, col]), np.sin(head[:, col])], -1)
        cross = hv[:, 0] * mv[:, 1] - hv[:, 1] * mv[:, 0]
        dot = (hv * mv).sum(-1)
        against = live & (st[:, col - 1] != 0) & (dot < 0) & (np.abs(cross) < 1e-5 * np.maximum(np.abs(dot), 1e-3))
        phi = -(head[ego, col] - np.pi / 2)
        rel = pos[:, col] - pos[ego, col]
        loc = np.stack([rel[:, 0] * np.cos(phi) - rel[:, 1] * np.sin(phi), rel[:, 0] * np.sin(phi) + rel[:, 1] * np.cos(phi)], -1)
        d = np.sqrt(((loc[:, None, :] - grid[None]) ** 2).sum(-1))
        two = np.partition(d, 1, axis=1)[:, :2]
        border = live & (two[:, 1] - two[:, 0] < 2e-4)
        if against.any() or border.any():
            return col - 1            # decode step t reads column 1 + t
    return pos.shape[1] - 2


def test_long_insertion_fixture_and_row_headroom():
    """80 decode steps with forced insertion on the reference (24 -> 129 agents; tests/golden/make_golden.py
    ins_forced_long_a24_m256), motion tokens teacher-forced (80 free-running steps are beyond what the logit margins
    guarantee): (a) with too few rows the engine says so instead of dropping insertions; (b) with enough rows the same agents
    are inserted at the same steps with the same cells / types / poses, and every row's top logit matches at every step, up
    to the first step that is ill-conditioned in the reference itself (a row moving exactly against its heading)"""
    from infgen_amd import engine
    c = load_case('ins_forced_long_a24_m256')
    z, m, cfg = c['z'], c['meta'], c['cfg']
    cfg.disable_insertion = False
    n_final = z['agent_id'].shape[0]
    assert n_final - m['A'] >= 100 and cfg.num_decode_steps == 80
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    teacher = [(z['next_token_idx'], z['next_state_idx'])]
    small = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], force_enter=True, teacher=teacher,
                                 insert_headroom=40)
    assert small.A_cap == 64
    with pytest.raises(engine.InsertionHeadroomError):
        small.rollout()
    t_ok = _first_ill_conditioned_step(z, c['grid'], int(z['ego_index']))
    assert 40 <= t_ok <= 80, t_ok
    eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, force_enter=True,
                               teacher=teacher, insert_headroom=n_final - m['A'] + 8)
    eng.prologue()
    eng.run(0, t_ok)
    o = eng.outputs()[0]
    n = int(z['n_agents_step'][t_ok - 1])
    assert n - m['A'] >= 60 and o['pos_a'].shape[0] == n, 'same number of agents inserted'
    cols = slice(0, 1 + t_ok)
    assert np.array_equal(o['agent_id'], z['agent_id'][:n]) and np.array_equal(o['pred_type'], z['pred_type'][:n])
    assert np.array_equal(o['next_state_idx'][:, cols], z['next_state_idx'][:n, cols])
    assert np.abs(o['pos_a'][:, cols] - z['pos_a'][:n, cols]).max() <= 2e-3
    assert np.abs(o['head_a'][:, cols] - z['head_a'][:n, cols]).max() <= 1e-3
    tol = 1e-3
    for t in range(t_ok):
        k = int(z['n_agents_step'][t])
        assert np.abs(o['logits'][t, :k].max(-1) - z['logit_max'][t, :k]).max() <= tol, t
        sure = z['margin'][t, :k] > 4 * tol
        assert np.array_equal(o['logits'][t, :k].argmax(-1)[sure], z['logit_argmax'][t, :k][sure]), t
    for t in range(z['logits'].shape[0]):
        k = int(z['n_agents_step'][t])
        assert np.abs(o['logits'][t, :k] - z['logits'][t, :k]).max() <= tol


def test_batched_insertion_equals_single_scene_runs():
    """insertion is per-scene state: a batch with ragged agent counts and different insertion histories must
    reproduce each scene decoded alone"""
    from infgen_amd import engine, synth
    c = load_case('ins_natural_a20_m256')
    cfg = c['cfg']
    cfg.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(8100 + i, a, m, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid'])
                             for i, (a, m) in enumerate([(12, 128), (30, 300)])]
    engb = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=False, a_cap=128)
    engb.rollout()
    outb = engb.outputs()
    assert np.array_equal(outb[0]['next_token_idx'], c['z']['next_token_idx'])
    n_ins = []
    for i, sc in enumerate(scenes):
        e1 = engine.RolloutEngine(w, [sc], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, a_cap=128)
        e1.rollout()
        o1 = e1.outputs()[0]
        assert o1['pos_a'].shape == outb[i]['pos_a'].shape
        assert np.array_equal(o1['next_token_idx'], outb[i]['next_token_idx'])
        assert np.abs(o1['pos_a'] - outb[i]['pos_a']).max() <= 1e-5
        n_ins.append(o1['num_inserted'])
    assert max(n_ins) > 0


@pytest.mark.parametrize('insertion', [False, True])
def test_device_epilogue_equals_host_epilogue(insertion):
    """RolloutEngine.outputs_device (the return dict of agent_decoder.py:2303-2389 built on the device for all scenes at once)
    against outputs() (numpy, per scene)"""
    from infgen_amd import engine, synth
    c = load_case('ins_natural_a20_m256' if insertion else 'a24_m256_edge')
    cfg = c['cfg']
    cfg.disable_insertion = not insertion
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(8300 + i, 9 + 5 * i, 150 + 30 * i, cfg, ego_last=(i % 2 == 0), edge_cases=(i == 1),
                                              vocab=c['vocab'], grid=c['grid']) for i in range(3)]
    eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, seed_outputs=insertion)
    eng.rollout()
    host, devo = eng.outputs(), eng.outputs_device()
    assert insertion == (sum(o['num_inserted'] for o in host) > 0)
    for h, d in zip(host, devo):
        assert set(h) == set(d), set(h) ^ set(d)
        for k, v in h.items():
            g = d[k]
            if isinstance(v, np.ndarray):
                g = g.cpu().numpy()
                assert g.shape == v.shape and g.dtype == v.dtype, (k, g.shape, v.shape, g.dtype, v.dtype)
                if v.dtype.kind == 'f':
                    assert np.abs(g - v).max() <= 1e-5 if v.size else True, k
                else:
                    assert np.array_equal(g, v), k
            else:
                assert g == v, k


def test_graph_replay_equals_eager_rollout():
    """RolloutEngine(use_graph=True): the decode steps captured in a HIP graph (second rollout) and replayed (third) give the
    eager rollout bit for bit"""
    from infgen_amd import engine, synth
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(8200 + i, 20 + i, 200, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid']) for i in range(3)]
    ref = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True)
    ref.rollout()
    r = ref.outputs()
    eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=True)
    for i in range(3):
        eng.rollout()
        torch.cuda.synchronize()
        assert (eng._graph is not None) == (i >= 1)
        for a, b in zip(eng.outputs(), r):
            assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
            assert np.array_equal(a['logits'], b['logits'])
            assert np.array_equal(a['pos_a'], b['pos_a'])
    assert np.array_equal(r[0]['next_token_idx'], c['z']['next_token_idx'])


def test_whole_rollout_graph_equals_eager_rollout():
    """RolloutEngine(use_graph='all'): reset, map encoder, column-0 chain and every decode step as ONE HIP graph (captured at the
    second rollout, replayed from then on; dropped by reload()) give the eager rollout bit for bit - also as several engines on
    several streams (rollout_many replays one graph per engine)"""
    from infgen_amd import engine, synth
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(8300 + i, 20 + i, 200, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid']) for i in range(3)]
    ref = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True)
    ref.rollout()
    r = ref.outputs()
    eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph='all')
    for i in range(3):
        eng.rollout()
        torch.cuda.synchronize()
        assert (eng._wgraph is not None) == (i >= 1)
        for a, b in zip(eng.outputs(), r):
            assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
            assert np.array_equal(a['logits'], b['logits'])
            assert np.array_equal(a['pos_a'], b['pos_a'])
    assert np.array_equal(r[0]['next_token_idx'], c['z']['next_token_idx'])
    eng.reload(scenes[::-1])
    assert eng._wgraph is None
    eng.rollout(); eng.rollout()
    torch.cuda.synchronize()
    for a, b in zip(eng.outputs(), r[::-1]):
        assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
    halves = [engine.RolloutEngine(w, scenes[i:i + 2], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph='all')
              for i in (0, 2)]
    streams = [torch.cuda.Stream(device=dev) for _ in halves]
    for _ in range(3):
        engine.rollout_many(halves, streams)
    torch.cuda.synchronize()
    outs = halves[0].outputs() + halves[1].outputs()
    for a, b in zip(outs, r):
        assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
        assert np.array_equal(a['logits'], b['logits'])


def test_warm_workgroups_are_bitwise_neutral(monkeypatch):
    """small launches carry workgroups that only read the next kernels' weights (tile.cuh: WarmArgs): same results with and
    without them (INFGEN_WARM_MAX_GROUPS is read once per process, so the comparison is against the reference fixture and a
    batch that is too large to carry any)"""
    from infgen_amd import engine, synth, _lib
    lib = _lib.load()
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    _lib.check(lib.infgen_set_edge_fuse(2))                     # the fused edge kernel's one-group variant carries them
    _lib.check(lib.infgen_set_layers_p(0))                      # (the per-sublayer launches: k_layers_p has no warm workgroups)
    try:
        few = [c['scene']] + [synth.make_scene(8400 + i, 24, 256, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid']) for i in range(7)]
        many = few + [synth.make_scene(8500 + i, 24, 256, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid']) for i in range(120)]
        # (edge_kernel 0 in both: the big batch's map encoder would otherwise take k_edge_fused3, whose fp32 summation order differs)
        a = engine.RolloutEngine(w, few, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, options=dict(edge_kernel=0, edge_fuse=2, layers_p=0))      # 8 x 32 rows: 16 groups
        b = engine.RolloutEngine(w, many, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, options=dict(edge_kernel=0, edge_fuse=2, layers_p=0))     # 256 groups: none
        a.rollout(); b.rollout()
        torch.cuda.synchronize()
        for x, y in zip(a.outputs(), b.outputs()[:8]):
            assert np.array_equal(x['next_token_idx'], y['next_token_idx'])
            assert np.array_equal(x['logits'], y['logits'])
        assert np.array_equal(a.outputs()[0]['next_token_idx'], c['z']['next_token_idx'])
    finally:
        _lib.check(lib.infgen_set_edge_fuse(1))
        _lib.check(lib.infgen_set_layers_p(1))


def test_few_scene_kernel_shapes_are_bitwise_neutral():
    """up to 128 scenes k_integrate runs one workgroup per 16 rows (every workgroup integrates the ego's step for itself; the arg-max
    keys are then cleared by the next k_build_edges) and k_build_edges 16 waves per workgroup; larger batches keep one workgroup
    per scene / 4 waves (csrc/api.hip: integrate_groups, INFGEN_BE_WIDE_SCENES - read once per process, so the comparison is
    between a batch below and a batch above the limit).  Same rows in, bitwise the same rollouts out - and the fixture's."""
    from infgen_amd import engine, synth, _lib
    lib = _lib.load()
    c = load_case('a24_m256_edge')
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    _lib.check(lib.infgen_set_layers_p(0))                      # (both batches through the per-sublayer launches)
    try:
        few = [c['scene']] + [synth.make_scene(8600 + i, 24, 256, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid']) for i in range(7)]
        many = few + [synth.make_scene(8700 + i, 20, 256, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid']) for i in range(125)]
        a = engine.RolloutEngine(w, few, c['vocab'], c['map_vocab'], c['grid'], store_logits=True)      # 8 scenes: the few-scene shapes
        b = engine.RolloutEngine(w, many, c['vocab'], c['map_vocab'], c['grid'], store_logits=True)     # 133 scenes: the big-batch shapes
        assert a.S <= 128 < b.S
        a.rollout(); b.rollout()
        torch.cuda.synchronize()
        assert a.edge_totals() != b.edge_totals()
        for x, y in zip(a.outputs(), b.outputs()[:8]):
            # what the two kernels produce - poses, states, grid cells behind them, the edge sets behind the tokens - bitwise;
            # the logits also pass through kernels chosen by the row count (256 rows: k_edge_attn with U / Z in memory; 4,256 rows:
            # k_edge_fused): rounding-level differences on logits of magnitude ~30
            for k in ('next_token_idx', 'next_state_idx', 'pos_a', 'head_a', 'pred_traj', 'pred_head'):
                assert np.array_equal(x[k], y[k]), k
            assert np.abs(x['logits'] - y['logits']).max() <= 1e-3
        assert np.array_equal(a.outputs()[0]['next_token_idx'], c['z']['next_token_idx'])
        assert np.array_equal(a.gridtok.cpu().numpy(), b.gridtok.cpu().numpy()[:8])
    finally:
        _lib.check(lib.infgen_set_layers_p(1))


def test_rollout_many_streams_equals_single_engine():
    """engine.rollout_many: engines on their own streams, sequenced cooperatively by one host thread (each yields where it
    needs the device's insertion decisions) - the same scenes give the same rollouts as one engine after the other"""
    from infgen_amd import engine, synth
    c = load_case('ins_natural_a20_m256')
    cfg = c['cfg']
    cfg.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    sets = [[c['scene'], synth.make_scene(8100, 12, 128, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid'])],
            [synth.make_scene(8101, 30, 300, cfg, ego_last=False, vocab=c['vocab'], grid=c['grid'])],
            [synth.make_scene(8102 + i, 16 + 4 * i, 256, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid']) for i in range(3)]]
    mk = lambda: [engine.RolloutEngine(w, sc, c['vocab'], c['map_vocab'], c['grid'], store_logits=False, a_cap=128) for sc in sets]
    ref = mk()
    for e in ref:
        e.rollout()
    ref_out = [e.outputs() for e in ref]
    many = mk()
    streams = [torch.cuda.Stream(device=dev) for _ in many]
    for _ in range(2):                       # twice: the second pass reuses the engines' buffers and events
        engine.rollout_many(many, streams)
        torch.cuda.synchronize()
    assert np.array_equal(ref_out[0][0]['next_token_idx'], c['z']['next_token_idx'])
    n_ins = 0
    for e, ro_ in zip(many, ref_out):
        for o, r in zip(e.outputs(), ro_):
            assert o['pos_a'].shape == r['pos_a'].shape
            assert np.array_equal(o['next_token_idx'], r['next_token_idx'])
            assert np.array_equal(o['next_state_idx'], r['next_state_idx'])
            assert np.array_equal(o['pos_a'], r['pos_a'])
            n_ins += o['num_inserted']
    assert n_ins > 0


def test_long_horizon_rollout_vs_oracle():
    """ours_long_term-style horizon (R = 300 -> 62 columns, 60 decode steps): the temporal ring wraps several
    times; HIP vs the CPU oracle (teacher-forced after the first steps to stay on the same trajectory)"""
    from infgen_amd import engine, synth
    from oracle import rollout_oracle as ro
    c = load_case('a24_m256_edge')
    cfg = synth.standard_config(num_recurrent_steps_val=300)
    scene = synth.make_scene(5151, 14, 128, cfg, ego_last=True, vocab=c['vocab'], grid=c['grid'])
    sd = make_weights(seed=6, head_gain=64.0)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = ro.run_scene(tsd, scene, cfg, c['vocab'], c['map_vocab'], c['grid'])
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(sd, cfg, dev)
    teacher = [(ref['next_token_idx'].numpy(), ref['next_state_idx'].numpy())]
    eng = engine.RolloutEngine(w, [scene], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, teacher=teacher)
    eng.rollout()
    o = eng.outputs()[0]
    lg = ref['logits'].numpy()
    assert lg.shape[0] == 60
    assert np.abs(o['logits'] - lg).max() <= 1e-3           # head sharpened x64
    assert np.abs(o['pos_a'] - ref['pos_a'].numpy()).max() <= 2e-3
    part = np.partition(lg, -2, axis=-1)
    ok = (part[..., -1] - part[..., -2]) > 2e-2
    assert np.array_equal(o['logits'].argmax(-1)[ok], lg.argmax(-1)[ok])


def test_topk_sampling_with_supplied_uniforms():
    """the reference's default decode is top-5 multinomial (motion_beam_size = 5, agent_decoder.py:300,2163,2194);
    torch RNG cannot be bit-matched, so the sampler is defined as inverse-CDF over the top-k probabilities with
    caller-supplied uniforms — HIP vs oracle with the same uniforms"""
    from infgen_amd import engine, synth
    from oracle import rollout_oracle as ro
    c = load_case('c1_a8_m128')
    cfg = c['cfg']
    rng = np.random.default_rng(99)
    A = c['z']['pos_a'].shape[0]
    u = rng.uniform(0, 1, size=(cfg.num_decode_steps, 1, A)).astype(np.float32)
    tsd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
    ref = ro.run_scene(tsd, c['scene'], cfg, c['vocab'], c['map_vocab'], c['grid'], sample_k=5, sample_uniforms=u[:, 0])
    assert not np.array_equal(ref['next_token_idx'].numpy(), c['z']['next_token_idx'])      # it really samples
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    teacher = [(ref['next_token_idx'].numpy(), ref['next_state_idx'].numpy())]
    # free-running sampled rollout
    eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, sample_k=5,
                               sample_uniforms=u)
    eng.rollout()
    o = eng.outputs()[0]
    margins = np.stack(ref['sample_margin'])            # distance of u to the nearest CDF edge, per (step, agent)
    if margins.min() > 1e-4:
        assert np.array_equal(o['next_token_idx'], ref['next_token_idx'].numpy())
        assert np.abs(o['pos_a'] - ref['pos_a'].numpy()).max() <= 1e-3
    else:
        first_bad = int(np.argwhere(margins <= 1e-4)[0][0])
        hc = cfg.hist_columns
        assert np.array_equal(o['next_token_idx'][:, :hc + first_bad], ref['next_token_idx'].numpy()[:, :hc + first_bad])
    # the sampler alone, without a stored logits buffer
    eng2 = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, sample_k=5,
                                sample_uniforms=u)
    eng2.rollout()
    assert np.array_equal(eng2.outputs()[0]['next_token_idx'], o['next_token_idx'])


def _oracle_vs_engine(cfg, scene, sd, c, tol=1e-3, min_margin=2e-2, **eng_kw):
    from infgen_amd import engine
    from oracle import rollout_oracle as ro
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ref = ro.run_scene(tsd, scene, cfg, c['vocab'], c['map_vocab'], c['grid'])
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(sd, cfg, dev)
    teacher = [(ref['next_token_idx'].numpy(), ref['next_state_idx'].numpy())]
    eng = engine.RolloutEngine(w, [scene], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, teacher=teacher,
                               **eng_kw)
    eng.rollout()
    o = eng.outputs()[0]
    lg = ref['logits'].numpy()
    assert np.abs(o['logits'] - lg).max() <= tol
    assert np.abs(o['pos_a'] - ref['pos_a'].numpy()).max() <= 2e-3
    part = np.partition(lg, -2, axis=-1)
    ok = (part[..., -1] - part[..., -2]) > min_margin
    assert np.array_equal(o['logits'].argmax(-1)[ok], lg.argmax(-1)[ok])
    if x_pt_ok := (ref['x_pt'].shape[0] > 0):
        assert np.abs(o['x_pt'] - ref['x_pt'].numpy()).max() <= 1e-4
    return o, ref


def test_degenerate_scenes():
    """empty / ragged inputs: a scene with the ego alone, a scene without any map token within reach, and a
    scene whose other agents are all far outside every radius (no a2a / map edges at all)"""
    from infgen_amd import synth
    c = load_case('c1_a8_m128')
    cfg = synth.standard_config(num_recurrent_steps_val=20)
    sd = make_weights(seed=7, head_gain=64.0)
    # ego alone
    scene = synth.make_scene(31, 1, 64, cfg, vocab=c['vocab'], grid=c['grid'])
    _oracle_vs_engine(cfg, scene, sd, c)
    # map tokens all far away (no map->agent edge, map encoder still runs)
    scene = synth.make_scene(32, 6, 40, cfg, vocab=c['vocab'], grid=c['grid'])
    scene['pt_token']['position'][:, :2] += 5000.0
    _oracle_vs_engine(cfg, scene, sd, c)
    # no map tokens at all / a single one (empty map encoder input, empty map edge set)
    for m in (0, 1):
        scene = synth.make_scene(35, 5, m, cfg, vocab=c['vocab'], grid=c['grid'])
        _oracle_vs_engine(cfg, scene, sd, c)
    # agents spread over kilometres: no agent<->agent edges
    scene = synth.make_scene(33, 12, 64, cfg, half_extent=5000.0, vocab=c['vocab'], grid=c['grid'])
    o, ref = _oracle_vs_engine(cfg, scene, sd, c)
    assert (ref['edge_count'][:, 1] == 0).all()


def test_maximum_sizes_stress_shape():
    """BASELINE config C5 shapes (256 agents, 4096 map tokens) in fp32, two decode steps: the per-scene
    kernels at their agent cap (A_cap = 256), a2a degree > 64, map LDS staging at 4096 tokens"""
    from infgen_amd import synth
    c = load_case('c1_a8_m128')
    cfg = synth.standard_config(num_recurrent_steps_val=10)
    sd = make_weights(seed=8, head_gain=64.0)
    scene = synth.make_scene(41, 256, 4096, cfg, half_extent=60.0, vocab=c['vocab'], grid=c['grid'], slip=0.2)
    o, ref = _oracle_vs_engine(cfg, scene, sd, c)
    assert o['pos_a'].shape[0] == 256
    assert ref['edge_count'][:, 1].max() > 256 * 64      # dense agent<->agent neighbourhoods


def test_side_stream_overlap_is_bitwise_neutral():
    """infgen_set_overlap(1): Fourier embeddings of the map / agent sets on a side stream; same kernels, same results"""
    from infgen_amd import engine, synth, _lib
    c = load_case('c2_a32_m512')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    scenes = [synth.make_scene(700 + i, 20 + i, 256, c['cfg'], vocab=c['vocab'], grid=c['grid'], slip=0.2) for i in range(8)]
    lib = _lib.load()
    outs = []
    _lib.check(lib.infgen_set_layers_p(0))       # (the per-sublayer launches: k_layers_p does not run with the side stream)
    try:
        for mode in (0, 1):
            _lib.check(lib.infgen_set_overlap(mode))
            eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True)
            eng.rollout()
            outs.append(eng.outputs())
    finally:
        _lib.check(lib.infgen_set_overlap(0))
        _lib.check(lib.infgen_set_layers_p(1))
    for a, b in zip(*outs):
        assert np.array_equal(a['logits'], b['logits']) and np.array_equal(a['pos_a'], b['pos_a'])


def test_fused_edge_attention_rollout_matches_default():
    """infgen_set_edge_fuse(1) (default): the absorbed query and the positional aggregate stay on chip inside k_edge_fused
    (no U / Z / SIG arrays); tokens identical, logits within fp32 noise of the unfused sequence (mode 0)"""
    from infgen_amd import engine, synth, _lib
    c = load_case('c2_a32_m512')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    scenes = [synth.make_scene(600 + i, 24 + i, 300, c['cfg'], vocab=c['vocab'], grid=c['grid'], slip=0.2) for i in range(12)]
    outs = []
    lib = _lib.load()
    try:
        for mode in (0, 1):
            _lib.check(lib.infgen_set_edge_fuse(mode))
            eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True)
            eng.rollout()
            outs.append(eng.outputs())
    finally:
        _lib.check(lib.infgen_set_edge_fuse(1))
    for a, b in zip(*outs):
        assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
        assert np.abs(a['logits'] - b['logits']).max() <= 2e-4


def test_packed_rhat_rows_match_fp32_rows():
    """the rollout can keep the normalised relative-position embeddings of its own edge sets as packed 24-bit rows (kernels.h
    R24_ROW_BYTES: k_fourier_h writes, k_edge_fused reads; relative error 2^-17; InfgenOptions.rhat_format = 1, a per-context
    option) - against the default, fp32 rows: tokens identical, logits within 5e-5.  Both engines are alive at the same time
    and run alternately: the option lives in the context, not in the process"""
    from infgen_amd import engine, synth
    c = load_case('c2_a32_m512')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    scenes = [synth.make_scene(700 + i, 24 + i, 300, c['cfg'], vocab=c['vocab'], grid=c['grid'], slip=0.2) for i in range(12)]
    engs = [engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True,
                                 options=dict(edge_fuse=2, rhat_format=fmt)) for fmt in (0, 1)]
    for _ in range(2):
        for e in engs:
            e.rollout()
    outs = [e.outputs() for e in engs]
    worst = 0.0
    for a, b in zip(*outs):
        assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
        worst = max(worst, float(np.abs(a['logits'] - b['logits']).max()))
    assert 0.0 < worst <= 5e-5, worst          # (> 0: the packed path really ran)


def test_time_gap_lookup_matches_per_edge_evaluation():
    """the temporal edges' time-gap branch of r_t_emb looked up (InfgenRollout.four_t_dt) against the same rollout with
    the branch evaluated per edge (RolloutEngine(flags={'dt_table': False})): tokens identical, logits within 2e-5 (fp32 summation
    order only) - and different at all, i.e. the lookup really ran.  A per-engine flag: both engines exist side by side"""
    from infgen_amd import engine, synth
    c = load_case('c2_a32_m512')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    scenes = [synth.make_scene(900 + i, 20 + i, 300, c['cfg'], vocab=c['vocab'], grid=c['grid'], slip=0.2) for i in range(12)]
    engs = [engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, flags={'dt_table': on})
            for on in (False, True)]
    outs = []
    for eng, on in zip(engs, (False, True)):
        eng.rollout()
        assert bool(eng._ctx.four_t_dt) == on
        outs.append(eng.outputs())
    worst = 0.0
    for a, b in zip(*outs):
        assert np.array_equal(a['next_token_idx'], b['next_token_idx'])
        worst = max(worst, float(np.abs(a['logits'] - b['logits']).max()))
    assert 0.0 < worst <= 2e-5, worst


def test_bench_size_batch_properties():
    """BASELINE C3 shapes at the bench's full size - the headline batch: 1024 scenes x 64 agents x 1024 map tokens, R = 80 -
    checked through size-independent properties: a second rollout of the same batch is bitwise identical; scenes are independent
    units, so sampled scenes run alone with the same kernels (split forced through the engine's options: the by-size choice would
    pick the fp32 kernels for a single scene) reproduce their rows of the batch bit for bit; every decoded token is a valid id and
    the poses finite"""
    from infgen_amd import engine, synth
    c = load_case('c1_a8_m128')
    cfg = synth.standard_config()
    S = 1024
    scenes = [synth.make_scene(synth.scene_seed(3, i), 64, 1024, cfg, vocab=c['vocab'], grid=c['grid'], slip=0.2)
              for i in range(S)]
    sd = make_weights(seed=1, head_gain=1.0)
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(sd, cfg, dev)
    opt = dict(attn_mode=1)                 # (the Fourier kernels are the split ones at every size already)
    eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=False, options=opt)
    eng.rollout()
    first = eng.outputs()
    eng.rollout()
    second = eng.outputs()
    for k in ('next_token_idx', 'next_state_idx', 'pos_a', 'head_a', 'pred_traj'):
        assert all(np.array_equal(a[k], b[k]) for a, b in zip(first, second)), k
    tok = np.stack([o['next_token_idx'] for o in first])
    assert tok.shape == (S, 64, 18) and tok[:, :, 2:].min() >= 0 and tok.max() < cfg.token_size
    assert all(np.isfinite(o['pred_traj']).all() for o in first)
    assert eng.agent_steps() == S * 64 * 80
    for i in (0, 137, 511, 1023):
        one = engine.RolloutEngine(w, [scenes[i]], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, options=opt)
        one.rollout()
        o = one.outputs()[0]
        for k in ('next_token_idx', 'pos_a', 'head_a', 'pred_traj'):
            assert np.array_equal(o[k], first[i][k]), (i, k)


def test_reduced_precision_mode_stays_close():
    """infgen_set_gemm_terms(1): the GEMM kernels on plain fp16 operands (the reduced-precision mode for BASELINE config C5,
    "bf16" there; fp16 keeps 11 bits).  Not inside the 1e-3 bar by design: teacher-forced logits stay within 5e-3 of the
    fp32-accurate split (unsharpened head), >= 99 % of the arg-max decisions agree, and switching back restores the
    default bit for bit"""
    from infgen_amd import _lib, engine
    lib = _lib.load()
    c = load_case('c2_a32_m512')
    z = c['z']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    teacher = [(z['next_token_idx'], z['next_state_idx'])]

    def run():
        eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, teacher=teacher)
        eng.rollout()
        return eng.outputs()[0]['logits']
    _lib.check(lib.infgen_set_attn_mode(1))
    try:
        full = run()
        _lib.check(lib.infgen_set_gemm_terms(1))
        half = run()
        _lib.check(lib.infgen_set_gemm_terms(3))
        again = run()
    finally:
        _lib.check(lib.infgen_set_gemm_terms(3))
        _lib.check(lib.infgen_set_attn_mode(2))
    assert np.array_equal(full, again)
    err = np.abs(half - full)
    assert 1e-6 < err.max() <= 5e-3
    assert (half.argmax(-1) == full.argmax(-1)).mean() >= 0.99
    assert lib.infgen_set_gemm_terms(4) != 0 and lib.infgen_set_gemm_terms(0) != 0


def test_more_than_256_rows_per_scene():
    """A_cap > 256 (long rollouts with insertion need the head-room): the per-scene kernels switch to 1024-thread workgroups.
    (a) 320 agents against the CPU oracle; (b) the natural-insertion fixture with a 288-row layout reproduces the reference
    fixture exactly like the 128-row layout does (same kernels, other workgroup size)"""
    from infgen_amd import engine, synth
    c = load_case('c1_a8_m128')
    cfg = synth.standard_config(num_recurrent_steps_val=10)
    sd = make_weights(seed=8, head_gain=64.0)
    scene = synth.make_scene(43, 320, 512, cfg, half_extent=90.0, vocab=c['vocab'], grid=c['grid'], slip=0.2)
    o, ref = _oracle_vs_engine(cfg, scene, sd, c)
    assert o['pos_a'].shape[0] == 320
    # (a') 330 agents inside one 60 m neighbourhood: radius_graph's max_num_neighbors = 300 binds (agent_decoder.py:632-633):
    # every destination keeps the candidates among the first 301 rows in range only
    dense = synth.make_scene(44, 330, 256, cfg, half_extent=18.0, vocab=c['vocab'], grid=c['grid'], slip=0.2)
    o2, ref2 = _oracle_vs_engine(cfg, dense, sd, c)
    assert o2['pos_a'].shape[0] == 330 and ref2['edge_count'][:, 1].max() < 330 * 329
    ci = load_case('ins_natural_a20_m256')
    cfg_i = ci['cfg']
    cfg_i.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(ci['sd'], cfg_i, dev)
    eng = engine.RolloutEngine(w, [ci['scene']], ci['vocab'], ci['map_vocab'], ci['grid'], store_logits=False, a_cap=288)
    assert eng.A_cap == 288
    eng.rollout()
    assert eng.scenes_at_row_cap() == 0
    out = eng.outputs()[0]
    z = ci['z']
    assert np.array_equal(out['next_token_idx'], z['next_token_idx'])
    assert np.array_equal(out['next_state_idx'], z['next_state_idx'])
    assert np.array_equal(out['agent_id'], z['agent_id'])
    assert np.abs(out['pos_a'] - z['pos_a']).max() <= 1e-3


def test_reference_internal_invariants_hold():
    """the reference's own runtime asserts as properties of the device state after a batch rollout with insertion
    (agent_decoder.py:1785-1789: the interact mask is exactly "state != invalid" on every processed column; :2351: an
    invalid step has an all-zero position; :2033-2034: one row per initial or inserted agent)"""
    from infgen_amd import engine, synth
    from infgen_amd.synth import INVALID
    c = load_case('ins_natural_a20_m256')
    cfg = c['cfg']
    cfg.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [synth.make_scene(8300 + i, a, m, cfg, ego_last=(i % 2 == 0), edge_cases=(i % 3 == 0 and a >= 8),
                               vocab=c['vocab'], grid=c['grid'], slip=0.2)
              for i, (a, m) in enumerate([(20, 256), (9, 100), (33, 300), (16, 128), (40, 256), (12, 64)])]
    eng = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=False)
    eng.rollout()
    outs = eng.outputs()
    assert sum(o['num_inserted'] for o in outs) > 0
    state = eng.state.cpu().numpy()          # [S][T][A_cap]
    imask = eng.imask.cpu().numpy().astype(bool)
    n = eng.n_agents.cpu().numpy()
    for s, o in enumerate(outs):
        A = int(n[s])
        assert o['pos_a'].shape[0] == o['agent_id'].shape[0] == o['next_state_idx'].shape[0]
        # decoded columns (the history columns carry the masks of the input scene, which synthetic edge cases decouple)
        assert np.array_equal(imask[s, 2:, :A], state[s, 2:, :A] != INVALID)
        assert not imask[s, :, A:].any()
        inv = o['next_state_idx'] == INVALID
        assert (o['pos_a'][inv] == 0).all() and (o['head_a'][inv] == 0).all()


def test_folded_step_tail_equals_stepwise_decode():
    """infgen_rollout_run folds a step's tail for few rows (next column's edge sets early, x_a_emb in the multi-set Fourier launch,
    k_integrate decoding / clearing the arg-max keys, clearing the edge totals and gathering the raw features): bit-identical to
    the step-by-step sequence of infgen_decode_step, free-running, on a ragged three-scene batch and on the C3-sized fixture"""
    from infgen_amd import engine, synth
    dev = torch.device('cuda:0')
    for name in ('a24_m256_edge', 'c3_a64_m1024'):
        c = load_case(name)
        cfg = c['cfg']
        scenes = [c['scene']] + [synth.make_scene(8100 + i, a, m, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid'])
                                 for i, (a, m) in enumerate([(9, 100), (40, 300)])]
        w = engine.PackedWeights(c['sd'], cfg, dev)
        mk = lambda: engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=False)
        a, b = mk(), mk()
        a.prologue(); b.prologue()
        a.run()                                             # one infgen_rollout_run: the folded sequence
        for t in range(cfg.num_decode_steps):
            b.step(t)                                       # infgen_decode_step per step
        for k in ('pos', 'head', 'state', 'token', 'gridtok', 'X', 'logits', 'pred_traj', 'pred_head', 'next_token'):
            assert torch.equal(getattr(a, k), getattr(b, k)), (name, k)
        # both sequences leave the LAST step's edge counts in the context (the folded one used to leave zeros)
        assert a.edge_totals() == b.edge_totals() and sum(a.edge_totals()) > 0, (a.edge_totals(), b.edge_totals())
        z = c['z']
        assert np.array_equal(a.outputs()[0]['next_token_idx'], z['next_token_idx'])
        a.rollout()                                         # a second rollout on the same engine starts from cleared keys again
        assert np.array_equal(a.outputs()[0]['next_token_idx'], z['next_token_idx'])


def test_layers_p_matches_per_sublayer_launches_and_reruns_bitwise():
    """k_layers_p (csrc/layers_p.hip: all 18 sublayers of a decode step in ONE launch, a resident workgroup per 16-row group, the
    scene's workgroups meeting at a counter before every agent sublayer) against the per-sublayer launches of k_edge_fused +
    k_attn_hs on a ragged 8-scene batch: same tokens / states / poses free-running over 16 steps, logits within the fp32 noise of
    two roundings of the same operators (reference layers.py:61-113); two runs of the new path bitwise equal (fixed-order
    reductions only, no atomics in the arithmetic); single-scene fixtures (2 / 4 workgroups) still reproduce the reference"""
    from infgen_amd import engine, synth, _lib
    lib = _lib.load()
    c = load_case('c3_a64_m1024')
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(8600 + i, a, m, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid'], slip=0.3)
                             for i, (a, m) in enumerate([(64, 1024), (9, 100), (40, 300), (64, 700), (33, 512), (17, 64), (50, 900)])]
    runs = {}
    try:
        for mode in (0, 1, 1):
            _lib.check(lib.infgen_set_layers_p(mode))
            e = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=False)
            e.rollout()
            torch.cuda.synchronize()
            runs.setdefault(mode, []).append({k: getattr(e, k).clone() for k in ('pos', 'head', 'state', 'token', 'X', 'logits')})
            if mode == 1 and len(runs[1]) == 1:
                assert np.array_equal(e.outputs()[0]['next_token_idx'], c['z']['next_token_idx'])
    finally:
        _lib.check(lib.infgen_set_layers_p(1))
    old, new, again = runs[0][0], runs[1][0], runs[1][1]
    for k in new:
        assert torch.equal(new[k], again[k]), k                  # bitwise reproducible
    assert torch.equal(old['token'], new['token']) and torch.equal(old['state'], new['state'])
    assert torch.allclose(old['pos'], new['pos'], atol=1e-4) and torch.allclose(old['head'], new['head'], atol=1e-5)
    gain = 64.0                                                   # the fixture's sharpened token head (make_golden.py: head_gain)
    err = (old['logits'] - new['logits']).abs().max().item()
    assert err <= 1e-3, err
    assert not torch.equal(old['logits'], new['logits'])          # (it IS another kernel: the switch took effect)


def test_layers_p_one_launch_per_decode_step():
    """with the default switches a small batch spends ONE edge / node launch per decode step (profiling ids of the per-sublayer
    kernels: k_edge_attn 1 instead of 18, k_attn_post / k_attn_pre 0 instead of 19 inside the steps)"""
    from infgen_amd import engine, synth, _lib
    lib = _lib.load()
    c = load_case('a24_m256_edge')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    e = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, use_graph=False)
    e.rollout()
    counts = {}
    try:
        for mode in (1, 0):
            _lib.check(lib.infgen_set_layers_p(mode))
            _lib.prof_enable((1 << len(_lib.KERNEL_IDS)) - 1)
            e.rollout()
            counts[mode] = _lib.prof_collect()
    finally:
        _lib.prof_enable(0)
        _lib.check(lib.infgen_set_layers_p(1))
    steps = c['cfg'].num_decode_steps
    assert counts[1]['k_edge_attn']['step_calls'] == steps and counts[1]['k_attn_post']['step_calls'] == 0
    assert counts[0]['k_edge_attn']['step_calls'] == 18 * steps and counts[0]['k_attn_post']['step_calls'] == 18 * steps


def test_profiler_stride_samples_the_step_launches_evenly():
    """bench.py's timed region brackets every fifth decode-step launch of the dominant kernel (infgen_prof_set_stride; an event
    pair costs launch-stream time): the launches are all COUNTED (infgen_prof_seen), the bracketed ones are every stride-th of the
    launches inside decode steps - all positions of a step's 18 sublayers equally often - and every launch outside them"""
    from infgen_amd import engine, _lib
    lib = _lib.load()
    c = load_case('a24_m256_edge')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    e = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, use_graph=False)
    e.rollout()
    steps = c['cfg'].num_decode_steps
    kid = 1 << _lib.KERNEL_IDS.index('k_edge_attn')
    try:
        _lib.check(lib.infgen_set_layers_p(0))                 # 18 edge launches per step
        _lib.prof_enable(kid)
        e.rollout()
        full, seen_full = _lib.prof_collect()['k_edge_attn'], _lib.prof_seen()['k_edge_attn']
        _lib.prof_enable(kid)
        _lib.prof_set_stride(5)
        for _ in range(5):
            e.rollout()
        part, seen_part = _lib.prof_collect()['k_edge_attn'], _lib.prof_seen()['k_edge_attn']
    finally:
        _lib.prof_enable(0)
        _lib.check(lib.infgen_set_layers_p(1))
    assert full['step_calls'] == seen_full['seen_step'] == 18 * steps and full['calls'] == seen_full['seen']
    assert seen_part['seen_step'] == 5 * 18 * steps and seen_part['seen'] == 5 * seen_full['seen']
    assert part['step_calls'] == 18 * steps                                        # a fifth of the 5 rollouts' step launches
    assert part['calls'] - part['step_calls'] == 5 * (full['calls'] - full['step_calls'])      # outside the steps: every launch
    # the same average duration within noise (single scene: ~20 us launches)
    a, b = full['step_ms'] / full['step_calls'], part['step_ms'] / part['step_calls']
    assert abs(a - b) <= 0.25 * a, (a, b)


def _eight_scene_engine(c, dev, seed0=8600, **kw):
    from infgen_amd import engine, synth
    cfg = c['cfg']
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(seed0 + i, a, m, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid'], slip=0.3)
                             for i, (a, m) in enumerate([(64, 1024), (9, 100), (40, 300), (64, 700), (33, 512), (17, 64), (50, 900)])]
    return engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=False, **kw)


@pytest.mark.parametrize('mode', [1, 2])
def test_layers_p_next_to_a_long_kernel_on_another_stream(mode):
    """VERDICT r4 item 3 / ADVICE r4: k_layers_p's workgroups meet at counters in global memory, so all of them must become
    resident.  A launch never exceeds the device's resident capacity and waits without a limit (mode 1; mode 2: the same through
    hipLaunchCooperativeKernel): an 8-scene engine (32 - 128 workgroups per launch) runs 20 rollouts while a second stream keeps
    most CUs busy with long kernels of another library (torch matmuls, ~10 ms each) - no trap, and tokens / states / poses /
    logits bitwise equal to the quiet run"""
    from infgen_amd import _lib
    lib = _lib.load()
    assert lib.infgen_layers_p_capacity() > 0, 'k_layers_p would never run on this device'
    c = load_case('c3_a64_m1024')
    dev = torch.device('cuda:0')
    e = _eight_scene_engine(c, dev, options={'layers_p': mode})
    keys = ('pos', 'head', 'state', 'token', 'X', 'logits')
    e.rollout()
    torch.cuda.synchronize()
    quiet = {k: getattr(e, k).clone() for k in keys}
    assert np.array_equal(e.outputs()[0]['next_token_idx'], c['z']['next_token_idx'])
    # the kernel in question really is the one that runs (one edge-side launch per decode step)
    _lib.prof_enable(1 << _lib.KERNEL_IDS.index('k_edge_attn'))
    e.rollout()
    assert _lib.prof_collect()['k_edge_attn']['step_calls'] == c['cfg'].num_decode_steps
    _lib.prof_enable(0)
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(8192, 8192, device=dev)
    b = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    for it in range(20):
        with torch.cuda.stream(side):
            for _ in range(6):                      # fp32 8192^3: ~10 ms each on every CU, queued ahead of and beside the rollout
                a = torch.mm(a, b) * 1e-2
        e.rollout()
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(getattr(e, k), quiet[k]), (it, k)


@pytest.mark.parametrize('mode', [1, 2])
def test_layers_p_engines_on_two_streams_from_two_host_threads(mode):
    """two engines whose decode steps are k_layers_p launches, driven concurrently from two host threads on two streams (the
    documented use of contexts, include/infgen_hip.h): the library orders the k_layers_p launches of different streams behind each
    other, so both finish and reproduce their single-engine results bitwise - 10 rollouts each"""
    import threading
    from infgen_amd import _lib
    c = load_case('c3_a64_m1024')
    dev = torch.device('cuda:0')
    keys = ('pos', 'head', 'state', 'token', 'X', 'logits')
    engs = [_eight_scene_engine(c, dev, options={'layers_p': mode}), _eight_scene_engine(c, dev, seed0=9100, options={'layers_p': mode})]
    quiet = []
    for e in engs:
        e.rollout()
        torch.cuda.synchronize()
        quiet.append({k: getattr(e, k).clone() for k in keys})
    streams = [torch.cuda.Stream(device=dev) for _ in engs]
    errors = []

    def work(i):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                for it in range(10):
                    engs[i].rollout()
                    streams[i].synchronize()
                    for k in keys:
                        if not torch.equal(getattr(engs[i], k), quiet[i][k]):
                            errors.append((i, it, k))
        except Exception as ex:          # noqa: BLE001
            errors.append((i, repr(ex)))
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths), 'an engine did not finish (k_layers_p launches waiting for each other?)'
    assert not errors, errors[:5]
    # the same two engines through rollout_many on two streams (one host thread)
    from infgen_amd import engine
    engine.rollout_many(engs, streams)
    torch.cuda.synchronize()
    for i, e in enumerate(engs):
        for k in keys:
            assert torch.equal(getattr(e, k), quiet[i][k]), (i, k)


def test_layers_p_is_refused_for_packs_without_the_layernorm_bounds():
    """k_layers_p scales a GEMM operand by a bound it reads from header slots 10..13 of the attention pack; a pack from
    an older packer has zeros there (scale 2^126 -> inf / NaN).  The library checks the header version (slot 14) and gives such a
    context the per-sublayer launches: same tokens, 18 edge launches per step instead of one"""
    from infgen_amd import engine, _lib
    lib = _lib.load()
    c = load_case('a24_m256_edge')
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], c['cfg'], dev)
    hdr = lib.infgen_attn_pack_offset(b'h_hdr')
    assert hdr > 0
    for pk in list(w.attn_t) + list(w.attn_m) + list(w.attn_a):
        assert float(pk[hdr + 14]) == 2.0                       # the current packer's version stamp
    w.attn_a[0] = w.attn_a[0].clone()                           # a fresh pointer: the library caches its verdict per pack
    w.attn_a[0][hdr + 10:hdr + 15] = 0.0                        # what the previous packer wrote
    e = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, use_graph=False)
    _lib.prof_enable(1 << _lib.KERNEL_IDS.index('k_edge_attn'))
    try:
        e.rollout()
        n = _lib.prof_collect()['k_edge_attn']['step_calls']
    finally:
        _lib.prof_enable(0)
    assert n == 18 * c['cfg'].num_decode_steps, n
    assert np.array_equal(e.outputs()[0]['next_token_idx'], c['z']['next_token_idx'])


def test_copies_share_one_map_encoding_and_equal_single_scene_runs():
    """RolloutEngine(scenes, copies=n) decodes every scene n times in lockstep over ONE map encoding (the
    reference runs n_rollout_close_val rollouts per scene, infgen/model/infgen.py:704-706, and offers inference_no_map(data,
    map_enc), infgen_decoder.py:132-134, so that the map is encoded once).  Three ragged scenes x 4 copies with DIFFERENT
    uniforms for the top-5 token draw: every copy equals the single-scene run with that copy's uniforms bit for bit (tokens,
    states, poses; logits bitwise for the same batch shape is not required - compared within 1e-4), the copies differ from each
    other, and the map-side buffers (x_pt, map K / V, the pt <-> pt graph) are sized for 3 scenes, not 12"""
    from infgen_amd import engine, synth
    c = load_case('c2_a32_m512')
    cfg = c['cfg']
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(9300 + i, a, m, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid'], slip=0.3)
                             for i, (a, m) in enumerate([(20, 300), (32, 512)])]
    n = 4
    steps = cfg.num_decode_steps
    rng = np.random.default_rng(7)
    amax = max(int(np.asarray(s_['agent']['state_idx']).shape[0]) for s_ in scenes)
    u = rng.random((steps, len(scenes) * n, amax)).astype(np.float32)
    e = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=False,
                             sample_k=5, sample_uniforms=u, copies=n)
    e.rollout()
    outs = e.outputs()
    assert len(outs) == len(scenes) * n
    assert e.x_pt.shape[0] == len(scenes) * e.M_cap and e.mapK[0].shape[0] == len(scenes) * e.M_cap
    assert e._mg['off'].shape[0] == len(scenes) * e.M_cap
    for i, sc in enumerate(scenes):
        toks = []
        for k in range(n):
            s = i * n + k
            one = engine.RolloutEngine(w, [sc], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, use_graph=False,
                                       sample_k=5, sample_uniforms=u[:, s:s + 1], a_cap=e.A_cap, m_cap=e.M_cap)
            one.rollout()
            o1, on = one.outputs()[0], outs[s]
            for key in ('next_token_idx', 'next_state_idx'):
                assert np.array_equal(o1[key], on[key]), (i, k, key)
            assert np.allclose(o1['pos_a'], on['pos_a'], atol=1e-4) and np.allclose(o1['logits'], on['logits'], atol=1e-4 * 4)
            assert np.array_equal(o1['x_pt'], on['x_pt'])
            toks.append(on['next_token_idx'])
        assert any(not np.array_equal(toks[0], t) for t in toks[1:]), 'the copies drew the same tokens: uniforms not per copy?'
    # a second rollout of the same engine (state reset from the device snapshot) repeats the first bit for bit
    e.rollout()
    again = e.outputs()
    for a_, b_ in zip(outs, again):
        assert np.array_equal(a_['next_token_idx'], b_['next_token_idx']) and np.array_equal(a_['pos_a'], b_['pos_a'])


def test_copies_with_insertion_share_the_map_rows_of_the_seed_layers():
    """copies + scenario insertion: the map K / V rows of the pt -> seed layers (InfgenInsertion.mapK / mapV) and the map -> seed
    edges (k_point_edges) go through the same map_scene indirection - 3 ragged scenes x 2 greedy copies reproduce each scene
    decoded alone (agents inserted at the same steps, same tokens; reference agent_decoder.py:1773-2105)"""
    from infgen_amd import engine, synth
    c = load_case('ins_natural_a20_m256')
    cfg = c['cfg']
    cfg.disable_insertion = False
    dev = torch.device('cuda:0')
    w = engine.PackedWeights(c['sd'], cfg, dev)
    scenes = [c['scene']] + [synth.make_scene(8100 + i, a, m, cfg, ego_last=(i % 2 == 0), vocab=c['vocab'], grid=c['grid'])
                             for i, (a, m) in enumerate([(12, 128), (30, 300)])]
    e = engine.RolloutEngine(w, scenes, c['vocab'], c['map_vocab'], c['grid'], store_logits=False, a_cap=128, copies=2)
    e.rollout()
    outs = e.outputs()
    assert len(outs) == 6 and e.ins['mapK'][0].shape[0] == 3 * e.M_cap
    assert np.array_equal(outs[0]['next_token_idx'], c['z']['next_token_idx'])
    n_ins = []
    for i, sc in enumerate(scenes):
        e1 = engine.RolloutEngine(w, [sc], c['vocab'], c['map_vocab'], c['grid'], store_logits=False, a_cap=128)
        e1.rollout()
        o1 = e1.outputs()[0]
        for k in range(2):
            ob = outs[2 * i + k]
            assert o1['pos_a'].shape == ob['pos_a'].shape and o1['num_inserted'] == ob['num_inserted']
            assert np.array_equal(o1['next_token_idx'], ob['next_token_idx'])
            assert np.abs(o1['pos_a'] - ob['pos_a']).max() <= 1e-5
        n_ins.append(o1['num_inserted'])
    assert max(n_ins) > 0
