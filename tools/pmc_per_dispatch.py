"""Per-dispatch HBM counters of one kernel (rocprofv3 --kernel-trace --pmc <counter> CSV): value per dispatch, grid, and the totals.
    python tools/pmc_per_dispatch.py <..._counter_collection.csv> <counter> <kernel substring> [max rows]"""
import re
import sys
import pandas as pd
d = pd.read_csv(sys.argv[1])
d = d[(d.Counter_Name == sys.argv[2]) & d.Kernel_Name.str.contains(sys.argv[3], regex=False)]
g = d.groupby(['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'LDS_Block_Size', 'Scratch_Size'] if 'Scratch_Size' in d.columns else ['Dispatch_Id', 'Kernel_Name', 'Grid_Size']).Counter_Value.sum().reset_index()
g['k'] = [re.sub(r'\(.*', '', k)[-60:] for k in g.Kernel_Name]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
print(g.drop(columns=['Kernel_Name']).head(n).to_string())
print('dispatches', len(g), 'sum KB', g.Counter_Value.sum(), 'mean KB', g.Counter_Value.mean())
