"""The numbers DESIGN.md / README.md / profiles/README.md quote for the headline line are the committed files' (retargeted at round 6's closing files: docs
that disagree with the files of the last commit).  Parses profiles/bench_r06f_s1024.json, profiles/r06f_kernel_stats_s1024.csv and
profiles/traffic.json and looks for their figures, rounded the way the documents write them."""
import csv
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(name):
    with open(os.path.join(REPO, name)) as f:
        return f.read()


def test_headline_numbers_in_the_documents_are_the_committed_line():
    d = json.loads(_read('profiles/bench_r06f_s1024.json'))
    r = d['roofline']
    design, readme, prof = _read('DESIGN.md'), _read('README.md'), _read('profiles/README.md')
    value = f"{d['value'] / 1e6:.1f} M"                       # 28.6 M
    ms = f"{d['ms_per_step']:.1f}"                            # 183.5
    launch = f"{r['avg_launch_us']:.1f}"                      # 212.3
    frac = f"{r['frac']:.3f}"                                 # 0.165
    traffic_mb = f"{r['traffic'] / 1e6:.0f} MB"               # 706 MB (looked up from the previous suite's passes)
    ratio = f"{r['traffic_ratio']:.2f} x"                     # 2.25 x
    multi = f"{d['multi_rollout_value'] / 1e6:.1f} M"         # 34.2 M
    value2 = f"{d['value'] / 1e6:.2f} M"                      # 28.56 M (profiles/README.md writes two decimals)
    for text, name in ((design, 'DESIGN.md'), (readme, 'README.md'), (prof, 'profiles/README.md')):
        assert value in text or value2 in text, f'{name} does not quote {value!r} / {value2!r} of profiles/bench_r06f_s1024.json'
        for needle in (ms, launch, frac, ratio, multi):
            assert needle in text, f'{name} does not quote {needle!r} of profiles/bench_r06f_s1024.json'
    assert traffic_mb in design and traffic_mb in readme
    assert f"{d['prologue_ms']:.1f}" in design and f"{d['decode_ms']:.1f}" in design
    # the flat keys the line must carry (a consumer that keeps top-level scalars only)
    for k in ('fp32_mfma_value', 'rhat24_value', 'strict_fp32_value', 'two_streams_value', 'multi_rollout_value', 'c3_literal_value', 'c3_8scene_value', 'prologue_ms',
              'decode_ms', 'roofline_frac', 'roofline_traffic_ratio'):
        assert isinstance(d[k], float), k
    assert d['roofline_frac'] == r['frac'] and d['parity']['ok']


def test_rocprof_average_and_traffic_quoted_from_the_files():
    design = _read('DESIGN.md')
    with open(os.path.join(REPO, 'profiles', 'r06f_kernel_stats_s1024.csv')) as f:
        rows = list(csv.DictReader(f))
    edge = next(r_ for r_ in rows if 'k_edge_fused' in r_['Name'])
    avg_us = float(edge['AverageNs']) / 1e3
    assert f'{avg_us:.2f}' in design, f'DESIGN.md does not quote the rocprofv3 average {avg_us:.2f} us of k_edge_fused'
    assert f"{int(edge['Calls']):,}" in design
    t = json.loads(_read('profiles/traffic.json'))['kernels']['k_edge_attn_step']
    assert f"{t['fetch_bytes_per_launch'] / 1e6:.1f} MB read" in design and f"{t['write_bytes_per_launch'] / 1e6:.1f} MB written" in design
    # the event average of the line and the rocprofv3 figure without the map encoder's launches agree (same box)
    d = json.loads(_read('profiles/bench_r06f_s1024.json'))
    total_ms = float(edge['TotalDurationNs']) / 1e6
    step_us = (total_ms - 15 * 4.07) * 1e3 / 1440
    assert abs(step_us - d['roofline']['avg_launch_us']) / d['roofline']['avg_launch_us'] < 0.05
