"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes of the LAST launch of each kernel.

    python tools/pmc_summary.py <dir_fetch>/x_counter_collection.csv <dir_write>/x_counter_collection.csv > profiles/...csv

gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section) -> hbm = (2 FETCH + WRITE) KB * 1024.
"""
import sys
import pandas as pd


def last_launch(path, counter):
    d = pd.read_csv(path)
    d = d[d.Counter_Name == counter]
    d['k'] = d.Kernel_Name.str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
    # sum over the counter's dimensions (XCDs) per dispatch, then keep the last dispatch of each kernel
    g = d.groupby(['k', 'Dispatch_Id']).Counter_Value.sum().reset_index()
    return g.sort_values('Dispatch_Id').groupby('k').Counter_Value.last()


f = last_launch(sys.argv[1], 'FETCH_SIZE')
w = last_launch(sys.argv[2], 'WRITE_SIZE')
print('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 1 --warmup 1, 1e6 events, last launch')
print('# hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128 B request, MI355X_MICROARCH.md section HBM)')
print('kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected')
rows = sorted(set(f.index) | set(w.index), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))
for k in rows:
    print('%s,%.1f,%.1f,%d' % (k, f.get(k, 0), w.get(k, 0), (2 * f.get(k, 0) + w.get(k, 0)) * 1024))
