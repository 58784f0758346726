"""Per-kernel sums of the counters of one rocprofv3 --pmc pass (the LARGEST launch of every kernel -- bench.py's pass 2 re-runs the
kernels on the few triggered events after the timed steps --, summed over XCDs / SEs).

    python tools/pmc_counters.py <dir>/**/x_counter_collection.csv [kernel-substring]
"""
import sys
import pandas as pd

d = pd.read_csv(sys.argv[1])
d['k'] = d.Kernel_Name.str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
if len(sys.argv) > 2:
    d = d[d.k.str.contains(sys.argv[2])]
g = d.groupby(['k', 'Dispatch_Id', 'Counter_Name']).Counter_Value.sum().reset_index()
tot = g.groupby(['k', 'Dispatch_Id']).Counter_Value.sum().reset_index()
last = tot.sort_values('Counter_Value').groupby('k').Dispatch_Id.last()
g = g[g.apply(lambda r: r.Dispatch_Id == last[r.k], axis=1)]
t = g.pivot(index='k', columns='Counter_Name', values='Counter_Value')
meta = d.groupby('k').agg(vgpr=('VGPR_Count', 'last'), sgpr=('SGPR_Count', 'last'), lds=('LDS_Block_Size', 'last'),
                          scratch=('Scratch_Size', 'last'), wg=('Workgroup_Size', 'last'), grid=('Grid_Size', 'last'))
t = t.join(meta)
pd.set_option('display.width', 250, 'display.max_columns', 50, 'display.max_colwidth', 60)
print(t.sort_values(t.columns[0], ascending=False).to_csv())
