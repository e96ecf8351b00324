#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (sum over dispatches).
--by-grid: one line per (kernel, grid size) - separates e.g. the decode-step launches of k_edge_fused (S x A rows) from the map
encoder's pt <-> pt launches (S x M rows) and the small-batch variants."""
import csv, sys, collections
by_grid = '--by-grid' in sys.argv
path = [a for a in sys.argv[1:] if not a.startswith('--')][0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
with open(path) as f:
    for row in csv.DictReader(f):
        k = row['Kernel_Name'].split('(')[0]
        if by_grid:
            k = k + ' @grid=' + str(row.get('Grid_Size', row.get('Grid_Size_X', '?')))
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        n[(k, row['Counter_Name'])] += 1
names = sorted({c for v in acc.values() for c in v})
print('kernel,dispatches,' + ','.join(names))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    disp = max(n[(k, c)] for c in names if (k, c) in n)
    print(k + ',' + str(disp) + ',' + ','.join(f'{v.get(c, 0):.0f}' for c in names))
