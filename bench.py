#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE config 2: 1e6 events, 1 EeV-class showers, one 5-channel dipole
station (S5), South-Pole exponential ice, SP1 attenuation, Alvarez2009, 4096-sample traces at 2 GHz.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one pass of the whole per-event hot path (ray tracing -> delta_C cut -> attenuation -> Askaryan ->
candidate cut -> antenna + filter response on the event's common time grid -> threshold trigger) over one batch of
synthetic events that is already resident in HBM.  Events shard across ranks (weak scaling: every rank owns
`--events` events); the only collective is one all-gather of the per-rank triggered masks (RCCL via
torch.distributed, backend nccl) after the timed loop's barrier -- there is no exchange inside the compute.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step; `cpu_baseline` times the
oracle (C ray tracer + numpy spectral chain, one thread) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ICE = (1.78, 0.423, 77.)           # southpole_2015 (NuRadioMC/utilities/medium.py:69)
N_SAMPLES, FS = 4096, 2.0
CHANNELS = np.array([[0., 0., -100. - i] for i in range(5)])
ENERGY = 3e17                       # shower energy [eV] of a 1 EeV neutrino at <y> ~ 0.3
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s
# algorithmic HBM bytes of the un-fused formulation, N = 4096, L = 5296
B_RAY, B_CHANNEL, B_PAIR = 601216, 169504, 320


def make_events(n, seed):
    """uniform in r^2 <= (4 km)^2 and z in [-2.7 km, 0], isotropic directions, hadronic showers"""
    rng = np.random.default_rng(seed)
    r = np.sqrt(rng.uniform(0, 4000. ** 2, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    vertex = np.stack([r * np.cos(phi), r * np.sin(phi), rng.uniform(-2700., 0., n)], axis=1)
    zenith = np.arccos(rng.uniform(-1, 1, n))
    azimuth = rng.uniform(0, 2 * np.pi, n)
    return vertex, zenith, azimuth


def cpu_baseline(n_sample, seed):
    """The oracle on one host thread over the first n_sample events of the same synthetic list."""
    from oracle import spectral_oracle as so  # checker / baseline only
    vertex, zenith, azimuth = make_events(n_sample, seed)
    st = so.Station(CHANNELS, n_samples=N_SAMPLES, fs=FS)
    vrms, vrms_e = so.vrms_from_filters(FS)
    t0 = time.time()
    n_trig = 0
    for i in range(n_sample):
        o = so.simulate_event(vertex[i], zenith[i], azimuth[i], ENERGY, 'HAD', None, st, ICE, vrms, vrms_e)
        n_trig += o['triggered']
        if time.time() - t0 > 40:
            n_sample = i + 1
            break
    dt = time.time() - t0
    return dict(value=n_sample / dt, unit="events/s", cores=1, kind="port",
                sample="%d events of the same synthetic list (oracle: C ray tracer + numpy chain), %.1f s, %d triggered"
                       % (n_sample, dt, n_trig))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)   # clocks and caches settle over the first calls (73 -> 69.5 ms)
    ap.add_argument('--events', type=int, default=1000000, help='events per rank and step')
    ap.add_argument('--cpu-sample', type=int, default=3000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    import nuradiomc_amd
    ctx = nuradiomc_amd.Context(ICE, 'SP1', device=local_rank)
    st = nuradiomc_amd.Station(ctx, CHANNELS, antenna='analytic_VPol', n_samples=N_SAMPLES, sampling_rate=FS, n_freq=25)
    n = args.events
    vertex, zenith, azimuth = make_events(n, 10 + rank)
    d_in = [ctx.to_device(a) for a in (vertex, zenith, azimuth, np.full(n, ENERGY), np.zeros(n, np.int32), np.ones(n))]
    if world > 1:
        trig_t = torch.zeros(n, dtype=torch.uint8, device='cuda')
        d_trig = trig_t.data_ptr()
    else:
        d_trig = ctx.malloc(n)

    def barrier():
        ctx.synchronize()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step(want_stats):
        return st.simulate_events_dev(n, *d_in, d_trig, askaryan_model='Alvarez2009', want_stats=want_stats)

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    stats = None
    for k in range(args.steps):
        stats = step(k == args.steps - 1)
    barrier()
    elapsed = time.perf_counter() - t0

    n_trig_total = stats['n_triggered']
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        gathered = torch.empty(world * n, dtype=torch.uint8, device='cuda')
        dist.all_gather_into_tensor(gathered, trig_t)  # the one collective: triggered masks over xGMI
        n_trig_total = int(gathered.sum().item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        value = world * n * args.steps / elapsed
        sm = stats['stage_ms']
        dom = max((k for k in sm if k != 'total'), key=lambda k: sm[k])
        kernel_of = {'raytrace': 'raytrace_kernel', 'ray_setup': 'select/scan/ray_setup kernels',
                     'amp_bound': 'amp_bound_kernel', 'attenuation': 'attenuation_group_kernel<32, 1>',
                     'efield_max': 'efield_bound_kernel + efield_max_kernel', 'event_grid': 'event_grid_kernel + candidate lists',
                     'length_tables': 'length_tables_kernel', 'channel': 'channel_prefilter_kernel + channel_conv_kernel'}
        # algorithmic HBM bytes per launch
        b_field = 2 * 2049 * 16 + 2 * 2049 * 16 + 2 * 4096 * 8     # write spec_N, c2r N in/out per ray
        alg = {'raytrace': B_PAIR * stats['n_pairs'],
               'amp_bound': 136 * stats['n_rays'] + 8 * stats['n_rays'],
               'attenuation': (32 + 8 * 25) * stats['n_active_rays'],
               'efield_max': 232 * stats['n_active_rays'] + b_field * stats['n_efield_transforms'],
               # only the transforms actually carried out are priced (pruned items move no algorithmic bytes)
               'channel': (B_RAY - b_field) * stats['n_ray_transforms'] + B_CHANNEL * stats['n_channel_transforms']}
        alg_bytes = alg.get(dom, 0)
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, see the file header);
        # only meaningful for the workload they were measured on
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'r01_rocprofv3_pmc_hbm_traffic_final.csv')
        if n == 1000000 and os.path.exists(pmc):
            for line in open(pmc):
                name, _, _, hbm = line.strip().rsplit(',', 3) if line.count(',') >= 3 else ('', 0, 0, 0)
                if name.startswith('nrhip::') and name.split('::')[1].split('<')[0] in kernel_of[dom].split(' + ')[-1]:
                    traffic = float(hbm) / 1e9
        achieved = alg_bytes / (sm[dom] * 1e-3) / 1e9 if sm[dom] > 0 else 0.
        b_event = B_RAY * stats['n_rays'] + B_CHANNEL * stats['n_channel_items'] + B_PAIR * stats['n_pairs']
        # FP64 view of the attenuation quadrature: one integrand evaluation = frequency-independent node part (shared
        # by the 25 lanes of a ray, ~110 flop incl. exp, 2 sqrt, 2 div) / 25 + per-lane exp + div (~45 flop)
        flop_per_eval = 110. / 25. + 45.
        fp64 = stats['n_integrand_evals'] * flop_per_eval / (sm['attenuation'] * 1e-3) / 1e12 if sm['attenuation'] > 0 else 0.
        out = {
            "metric": "simulated events/sec (1e6-evt 1 EeV SP survey)", "value": value, "unit": "events/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d events/GPU, 3e17 eV hadronic showers, 5-ch analytic_VPol "
                                   "station at -100..-104 m, southpole_2015 ice, SP1, Alvarez2009, 4096 samples @ 2 GHz, "
                                   "Butterworth 80-500 MHz, 3 Vrms threshold" % n,
                       "events_per_gpu": n, "n_pairs": stats['n_pairs'], "n_rays": stats['n_rays'],
                       "n_active_rays": stats['n_active_rays'], "n_candidate_events": stats['n_candidate_events'],
                       "n_channel_items": stats['n_channel_items'], "n_channel_transforms": stats['n_channel_transforms'],
                       "n_ray_transforms": stats['n_ray_transforms'], "n_efield_transforms": stats['n_efield_transforms'],
                       "n_triggered_rank0": stats['n_triggered'], "n_triggered_all": n_trig_total,
                       "n_distinct_trace_lengths": stats['n_distinct_lengths'],
                       "triggered_events_per_s": n_trig_total / (elapsed / max(args.steps, 1)),
                       "stage_ms_last_step": {k: round(v, 3) for k, v in sm.items()}},
            "roofline": {"bound": "hbm", "kernel": kernel_of[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "GB per launch (rocprofv3 PMC, profiles/r01_rocprofv3_pmc_hbm_traffic_final.csv)",
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": sm[dom],
                         "whole_step_equivalent_GBs": b_event / (sm['total'] * 1e-3) / 1e9 if sm['total'] > 0 else 0.,
                         "whole_step_equivalent_frac": b_event / (sm['total'] * 1e-3) / 1e9 / HBM_PEAK_GBS if sm['total'] > 0 else 0.,
                         "attenuation_integrand_evals": stats['n_integrand_evals'],
                         "attenuation_fp64_tflops_est": fp64, "fp64_vector_peak_tflops": 78.6,
                         "note": "every kernel of the path is FP64-VALU/LDS bound: the fused kernels move ~1e-3 of the "
                                 "un-fused algorithmic bytes of SURVEY 8(d) that 'achieved' is priced on"},
        }
        if dom == 'attenuation':
            # the quadrature kernel is FP64 bound: price it in flops against the dense FP64 peak of the
            # MI355X (78.6 TFLOP/s, vector and matrix alike; the kernel issues VALU FP64, there is no contraction for MFMA).
            # algorithmic flops per launch = integrand evaluations (counted by the kernel) x flop_per_eval
            out["roofline"].update({"bound": "mfma", "achieved": fp64, "peak": 78.6, "unit": "TFLOP/s", "frac": fp64 / 78.6,
                                    "algorithmic_flops_per_launch": stats['n_integrand_evals'] * flop_per_eval,
                                    "hbm_view_GBs": achieved,
                                    "note": "FP64 VALU kernel priced against the dense FP64 peak (no MFMA-shaped work on this "
                                            "path); 'traffic' = HBM bytes of the launch from the PMC passes"})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, 10)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
