"""kernel-trace CSV of rocprofv3 -> per-kernel durations and the idle gaps between consecutive kernels of the busiest stream
(what a small-batch rollout spends between launches).  python tools/trace_gaps.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
# the last third of the trace = the timed rollouts
t_lo = ev[0][0] + 2 * (ev[-1][1] - ev[0][0]) // 3
ev = [e for e in ev if e[0] >= t_lo]
dur = collections.defaultdict(lambda: [0, 0.0])
gap_after = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for i, (s, e, k) in enumerate(ev):
    name = k.split('(')[0][:60]
    dur[name][0] += 1; dur[name][1] += (e - s) / 1e3
    busy += (e - s) / 1e3
    if i + 1 < len(ev):
        g = (ev[i + 1][0] - e) / 1e3
        gap_after[name][0] += 1; gap_after[name][1] += g
span = (ev[-1][1] - ev[0][0]) / 1e3
print(f'span {span:.1f} us, kernels busy {busy:.1f} us ({100*busy/span:.1f} %), {len(ev)} launches, mean gap {(span-busy)/max(1,len(ev)-1):.2f} us')
print(f'{"kernel":62s} {"n":>6s} {"avg us":>9s} {"total us":>10s} {"avg gap after":>14s}')
for k, (n, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    ga = gap_after[k]
    print(f'{k:62s} {n:6d} {t/n:9.2f} {t:10.1f} {ga[1]/max(1,ga[0]):14.2f}')
