"""profiles/traffic.json from the two PMC summaries of tools/prof_round.sh (FETCH_SIZE / WRITE_SIZE per kernel, KB summed over
the dispatches of the pass): bytes per launch, FETCH corrected by the factors tools/calibrate_fetch.sh measured.
usage: make_traffic_json.py <pmc_FETCH_SIZE.summary.csv> <pmc_WRITE_SIZE.summary.csv> <fetch_factor_8B> <fetch_factor_16B> <tag>
       [<pmc_FETCH_SIZE.bygrid.csv> <pmc_WRITE_SIZE.bygrid.csv>]   -> also `k_edge_attn_step`: the decode-step launches alone"""
import csv, json, os, sys
fetch, write, f8, f16, tag = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
names = {'k_edge_fused': ('k_edge_attn', f8), 'k_attn_h': ('k_attn_post', f16), 'k_fourier_h': ('k_fourier', f16)}


def load(path, col):
    out = {}
    for line in list(open(path))[1:]:
        kernel, disp, val = line.rstrip('\n').rsplit(',', 2)      # (kernel names contain commas: k_attn_h<4, 3>)
        for k, (kid, fac) in names.items():
            if k in kernel:                                          # (template variants of one kernel add up: k_edge_fused<6, true | false>)
                b, n, _ = out.get(kid, (0.0, 0, fac))
                out[kid] = (b + float(val) * 1024.0, n + int(disp), fac)
    return out


def load_step(path):
    """the decode-step launches of k_edge_fused from a --by-grid summary: the (kernel, grid) group with the most dispatches
    (18 launches per decode step, all of one grid size; the map encoder's three pt <-> pt launches have another)"""
    best = None
    for line in list(open(path))[1:]:
        kernel, disp, val = line.rstrip('\n').rsplit(',', 2)
        if 'k_edge_fused' in kernel and '@grid=' in kernel and (best is None or int(disp) > best[1]):
            best = (float(val) * 1024.0, int(disp), kernel)
    return best


fe, wr = load(fetch, 'FETCH_SIZE'), load(write, 'WRITE_SIZE')
kern = {}
if len(sys.argv) > 7:            # <pmc_FETCH_SIZE.bygrid.csv> <pmc_WRITE_SIZE.bygrid.csv>
    sf, sw = load_step(sys.argv[6]), load_step(sys.argv[7])
    if sf and sw:
        # the group holds, per rollout, the 16 x 18 decode-step launches AND the 18 edgeless launches of the column-0 chain (same
        # grid; they read q and write agg only, ~2 x 512 B per row).  All of the group's bytes are charged to the decode-step
        # launches: a slight over-estimate of their traffic, never an under-estimate
        # (since round 4 the edgeless chain skips its edge launches - api.hip: skip_edges - and the group is the decode-step launches
        # alone: a multiple of 288; older builds: 306 per rollout)
        n_step = sf[1] if sf[1] % 288 == 0 else sf[1] * 288 // 306
        kern['k_edge_attn_step'] = dict(fetch_bytes_per_launch=sf[0] / n_step / f8, fetch_counter_bytes_per_launch=sf[0] / n_step,
                                        fetch_counter_factor=f8, write_bytes_per_launch=sw[0] / n_step, dispatches=n_step,
                                        group=sf[2], group_dispatches=sf[1])
for kid in fe:
    fb, n, fac = fe[kid]
    wb, nw, _ = wr[kid]
    kern[kid] = dict(fetch_bytes_per_launch=fb / n / fac, fetch_counter_bytes_per_launch=fb / n, fetch_counter_factor=fac,
                     write_bytes_per_launch=wb / nw, dispatches=n)
json.dump(dict(source=f'{tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --no-cpu-baseline '
                      f'--no-parity --steps 1 --warmup 1` (tools/prof_round.sh); FETCH_SIZE divided by the counted / actual factor '
                      f'of a known-size streaming read of the same width (tools/calibrate_fetch.sh), WRITE_SIZE as counted',
               scenes_per_gpu=int(os.environ.get('SCENES', 1024)), agents=64, map_tokens=1024, insertion=False, rollout_steps=80, kernels=kern),
          open('profiles/traffic.json', 'w'), indent=1)
print(json.dumps(kern, indent=1))
