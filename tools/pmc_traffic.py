"""Two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; each with --kernel-trace only) of `python3 bench.py --steps 1 --warmup 3`
-> HBM bytes of the LARGEST launch of every kernel (bench.py's pass 2 re-runs the kernels on the few triggered events after the
timed steps: the last launch is not the timed one), as JSON for bench.py's `roofline.traffic` and as CSV for reading.

    python tools/pmc_traffic.py <fetch>/..._counter_collection.csv <write>/..._counter_collection.csv profiles/r02_pmc_traffic

gfx950: FETCH_SIZE tallies 64 B per 128-B request (MI355X_MICROARCH.md, section HBM): bytes = (2 FETCH_SIZE + WRITE_SIZE) KB x 1024.
The JSON carries the hash of the kernel sources it was taken on (bench.source_hash); bench.py quotes it only on those sources.
"""
import json
import os
import re
import sys
import pandas as pd

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def largest_launch(path, counter):
    d = pd.read_csv(path)
    d = d[d.Counter_Name == counter].copy()
    d['k'] = [re.sub(r'^.*::', '', re.sub(r'[<(].*', '', k.replace('void ', ''))) for k in d.Kernel_Name]
    g = d.groupby(['k', 'Dispatch_Id']).Counter_Value.sum().reset_index()   # sum over XCDs
    return g.groupby('k').Counter_Value.max(), g.groupby('k').Dispatch_Id.nunique()


def summed_launches(path, counter, n_steps):
    """arrays (--per-step K: the profiled command made K passes over the list, warm-up included): all launches of a kernel summed,
    divided by K -- the kernel's HBM bytes per step"""
    d = pd.read_csv(path)
    d = d[d.Counter_Name == counter].copy()
    d['k'] = [re.sub(r'^.*::', '', re.sub(r'[<(].*', '', k.replace('void ', ''))) for k in d.Kernel_Name]
    return d.groupby('k').Counter_Value.sum() / n_steps, d.groupby('k').Dispatch_Id.nunique()


def main():
    import bench
    per_step = None
    if '--per-step' in sys.argv:
        i = sys.argv.index('--per-step')
        per_step = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    if per_step:
        f, nf = summed_launches(sys.argv[1], 'FETCH_SIZE', per_step)
        w, _ = summed_launches(sys.argv[2], 'WRITE_SIZE', per_step)
    else:
        f, nf = largest_launch(sys.argv[1], 'FETCH_SIZE')
        w, _ = largest_launch(sys.argv[2], 'WRITE_SIZE')
    names = sorted(set(f.index) | set(w.index), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))
    rows = [(k, f.get(k, 0.), w.get(k, 0.), (2 * f.get(k, 0.) + w.get(k, 0.)) * 1024, int(nf.get(k, 0))) for k in names]
    base = sys.argv[3]
    with open(base + '.csv', 'w') as o:
        if per_step:
            o.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of a bench.py command that makes %g pass(es) over the list\n' % per_step)
            o.write('# ALL launches of each kernel summed, per pass; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 counts 64 B per 128-B read request)\n')
        else:
            o.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 3\n')
            o.write('# largest launch of each kernel; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 counts 64 B per 128-B read request)\n')
        o.write('# kernel sources %s\n' % bench.source_hash())
        o.write('kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected,launches_seen\n')
        for r in rows:
            o.write('%s,%.1f,%.1f,%d,%d\n' % r)
    json.dump({'source_hash': bench.source_hash(), 'file': os.path.basename(base) + '.csv',
               'unit': ('GB per step (all launches of the kernel in one pass over the list)' if per_step else 'GB per launch (largest launch)') +
                       ', (2 FETCH_SIZE + WRITE_SIZE) KB x 1024',
               'kernels': {r[0]: round(r[3] / 1e9, 4) for r in rows}}, open(base + '.json', 'w'), indent=1)
    print(open(base + '.csv').read())


if __name__ == '__main__':
    main()
