#!/usr/bin/env python3
"""bench.py -- generator fwd+bwd images/sec at 256^2 on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...      (no launcher: starts that same torch.distributed.run command itself, as a child process)

One step = one generator training step of the `--model stylegan3` path on one batch of synthetic MR-like slices:
mapping + encoder + co-modulated decoder forward (HIP filtered_lrelu / bias_act / MFMA convs), lambda_L1 * L1 loss,
backward, gradient all-reduce across ranks (RCCL), NaN scrub, Adam step -- StyleGAN3GeneratorStep.optimize_parameters.
Workload at N=1: BASELINE.json configs[1] (256x256, bf16, batch 16 on one GPU); for N>1 the same per-GPU batch on
every rank (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_HBM_GBS = 8000.0            # HBM3E spec; 6.29 TB/s measured float4 copy
PEAK_MFMA_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('AFCM_BENCH_BATCH', 16)), help='per-GPU batch')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--cpu-baseline', default='auto', choices=['auto', 'off'])
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--graph-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-threads', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-timed', type=int, default=2, help=argparse.SUPPRESS)
    ap.add_argument('--host-threads', type=int, default=0,
                    help='intra-op threads of this rank (torch.set_num_threads); 0 = cpu_count // (8 * ranks): the host side of a step is a '
                         'stream of launches from ONE Python thread, and a pool of 128 OpenMP workers per rank (torch\'s default on a 256-thread '
                         'host) only adds wake-ups -- eight ranks share that host')
    ap.add_argument('--no-kernel-timing', action='store_true', help='skip the HIP-event spans around the hot kernels')
    ap.add_argument('--kernel-timing-every', type=int, default=4,
                    help='bracket the hot launches with HIP events in every n-th timed step only (every event fences its launch: all steps '
                         'instrumented cost 0.9-1.0 ms of a 41 ms step, measured; every 4th: 0.25 ms)')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL and run the gradient buckets even with one rank (path check)')
    ap.add_argument('--comm-dtype', default='fp32', choices=['fp32', 'bf16'], help='dtype of the gradient buckets on the wire')
    ap.add_argument('--no-homogeneous-dot', action='store_true', help='ablation: the demodulation gradient from real plane dot products everywhere '
                                                                      '(torch_utils/ops/fused_layer.py HOMOGENEOUS_DOT)')
    ap.add_argument('--watchdog', type=int, default=int(os.environ.get('AFCM_BENCH_WATCHDOG', 0)),
                    help='seconds after which a rank that is still running dumps every thread\'s stack to stderr and exits (0: off); a rank stuck in a '
                         'collective otherwise burns the launcher\'s whole time limit without a word')
    ap.add_argument('--fp32-conv', default='f16x3', choices=['f16x3', 'bf16x6', 'bf16x663', 'bf16x633', 'bf16x3', 'native'],
                    help='--dtype fp32 only: the 3x3 convs on the 16-bit matrix pipe from split operands (f16x3: scaled float16 parts, 3 terms, '
                         'fp32-grade; bf16x6: 6 terms, fp32-grade; bf16x663 / 633: 3 terms in the weight / both gradients; bf16x3: ~16 bits) '
                         'or on the native fp32 MFMA kernels')
    ap.add_argument('--also', default='auto', choices=['auto', 'off'],
                    help='auto: on the default one-GPU workload, append `also: {fp32, d_plus_g}` -- the same step in the reference\'s own precision '
                         '(fp32) and the full D + G iteration, --also-steps timed steps each, measured in this process after the headline')
    ap.add_argument('--also-steps', type=int, default=10)
    ap.add_argument('--host-calibration', default='auto', choices=['auto', 'off'], help='200 empty launches: host and device microseconds per launch')
    ap.add_argument('--no-launch-census', action='store_true',
                    help='skip the extra untimed step under the framework\'s kernel tracer (tools/pmc*.sh pass it: no second tracer beside rocprofv3, '
                         'no extra step in the kernel statistics)')
    ap.add_argument('--with-discriminator', action='store_true',
                    help='time the FULL iteration (D update with R1, then G update with the GAN term; SURVEY.md row f1) instead of the '
                         'generator step that BASELINE.json\'s metric names')
    ap.add_argument('--lean', action='store_true',
                    help='the timed steps and nothing else: --cpu-baseline off --also off --host-calibration off --no-launch-census (A/B tools and '
                         'every rocprofv3 script: one tracer, no extra steps in the kernel statistics)')
    a = ap.parse_args()
    if a.lean:
        a.cpu_baseline = a.also = a.host_calibration = 'off'
        a.no_launch_census = True
    return a


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def cpu_baseline_worker(res, batch=2, timed=2, threads=0):
    """Child process: time the CPU oracle (pure-aten restatement of the reference's impl='ref' path, oracle/) as SURVEY.md
    section 8(d) / BASELINE.md section 4 plan it: full-width generator, batch 2, fwd + L1 loss + bwd on the same synthetic tensors
    the GPU run uses (afcm_amd.synthetic, seed 0), `threads` host threads (0 = every core), 1 warm-up + `timed` timed iterations."""
    import torch
    from oracle import generator as ogen
    threads = threads or (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    from afcm_amd import synthetic
    pl = ogen.plan(res, 4, 1, {})
    sd = ogen.random_state_dict(pl, 512, 1, 512, 8, seed=0)
    params = [v.requires_grad_(True) for k, v in sd.items() if not k.endswith(('magnitude_ema', 'w_avg', 'up_filter', 'down_filter'))]
    real_A, real_B, z, c = synthetic.generator_inputs(batch, size=res, seed=0)
    mask = (torch.rand(batch, 1024, generator=torch.Generator().manual_seed(0)) > 0.5).float() * 2
    times = []
    for it in range(1 + timed):
        for p in params:
            p.grad = None
        t0 = time.time()
        y = ogen.generator(sd, pl, z, c, real_A, mapping_layers=8, dropout_mask=mask)
        loss = (y - real_B).abs().mean() * 100.0
        loss.backward()
        times.append(time.time() - t0)
        print(json.dumps(dict(progress=it, seconds=times[-1])), flush=True)
    print(json.dumps(dict(seconds=sum(times[1:]) / timed, warmup_seconds=times[0], images=batch, batch=batch, timed=timed, cores=threads,
                          cpu=_cpu_model(), res=res)))


def run_cpu_baseline(res, budget=230.0):
    """Bounded CPU samples, each in a child process under its own time limit so the GPU number is never blocked: the bench resolution
    on 32 threads (1 warm-up + 2 timed iterations; the oracle's small aten ops stop scaling beyond a few dozen threads), config-1's
    shape (128^2, BASELINE.json configs[0]: the reference's own CPU-runnable case; 1 + 1).  `value` is the point at the bench resolution;
    every point is listed with its thread count, every sample that did not finish or was not taken is named."""
    points, notes = [], []
    t_start = time.time()
    ncores = os.cpu_count() or 1

    def sample(r, threads, timed, limit):
        try:
            env = dict(os.environ, OMP_NUM_THREADS=str(threads))          # (this process capped its own pool: the sample sets its own)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--res', str(r), '--cpu-threads', str(threads),
                                  '--cpu-timed', str(timed)], capture_output=True, text=True, timeout=limit, cwd=ROOT, env=env)
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"seconds"')][-1])
            points.append(dict(resolution=r, batch=d['batch'], images_per_sec=d['images'] / d['seconds'], s_per_iteration=d['seconds'],
                               warmup_s=d['warmup_seconds'], timed_iterations=d['timed'], cores=d['cores'], cpu=d['cpu']))
        except Exception as e:  # timeout or failure: say so, keep what was measured
            notes.append(f'{r}x{r} on {threads} threads: {type(e).__name__} (limit {limit:.0f} s)')

    few = min(ncores, 32)
    sample(res, few, 2, min(130.0, budget))
    at_res = [p for p in points if p['resolution'] == res]
    if res != 128 and budget - (time.time() - t_start) > 50.0:
        sample(128, few, 1, min(70.0, budget - (time.time() - t_start)))
    if ncores > few:
        # (r03 / r04 also tried every core -- SURVEY.md 8(d) "all cores": on the GPU boxes' 256-thread hosts the oracle's small aten ops run
        # slower than on 32 threads and the sample never finished inside 80-90 s against 31-33 s per iteration on 32; it cost every bench
        # run ~87 s for no number, so r05 dropped it and says so here)
        notes.append(f'{res}x{res} on all {ncores} threads: not sampled (r03/r04: did not finish in 80-90 s; 32 threads is the stated value)')
    at_res = [p for p in points if p['resolution'] == res]
    if not points:
        return dict(value=None, unit='images/sec', cores=ncores, kind='port', sample='not measured: ' + '; '.join(notes))
    h = max(at_res, key=lambda p: p['images_per_sec']) if at_res else points[-1]
    return dict(value=h['images_per_sec'], unit='images/sec', cores=h['cores'], kind='port',
                sample=f'oracle/ (aten restatement of the reference impl=ref path), full-width {h["resolution"]}x{h["resolution"]} generator, '
                       f'batch {h["batch"]}, fwd + L1 + bwd on the GPU run\'s synthetic tensors, 1 warm-up ({h["warmup_s"]:.1f} s) + '
                       f'{h["timed_iterations"]} timed iteration(s) ({h["s_per_iteration"]:.1f} s each) on {h["cores"]} threads of {h["cpu"]} '
                       f'({ncores} cores; the faster of the thread counts in `points`)'
                       + ('' if not notes else '; ' + '; '.join(notes)),
                points=points)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a CHILD process -- this parent has made no GPU call (torch is not even imported yet) and only relays the child's
    output and exit code; rank 0's JSON line is the last line the child prints to stdout."""
    # --standalone: the launcher picks a free rendezvous port itself and holds it (a port found by bind-and-close here could be taken
    # by another process before the ranks connect); 127.0.0.1 because the container's hostname may not resolve
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1', f'--nproc-per-node={n}',
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8 * n) // (8 * n))))      # the rule of --host-threads (main())
    print(f'bench.py: --gpus {n} without a launcher, starting: {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def _thread_cpu_ns():
    """{tid: (comm, on-CPU nanoseconds)} of this process's threads (/proc/self/task/*/schedstat; utime + stime ticks where that file is absent)."""
    out = {}
    base = '/proc/self/task'
    try:
        tids = os.listdir(base)
    except OSError:
        return out
    tick_ns = 1e9 / os.sysconf('SC_CLK_TCK')
    for tid in tids:
        try:
            comm = open(f'{base}/{tid}/comm').read().strip()
            try:
                ns = int(open(f'{base}/{tid}/schedstat').read().split()[0])
            except (OSError, ValueError, IndexError):
                f = open(f'{base}/{tid}/stat').read().rsplit(')', 1)[1].split()
                ns = (int(f[11]) + int(f[12])) * tick_ns
            out[int(tid)] = (comm, ns)
        except OSError:
            pass
    return out


def host_calibration(dev, launches=200):
    """What ONE launch costs the host here, with no work behind it: `launches` empty kernels through the C ABI (ctypes ->
    hipLaunchKernel) and the same number of one-element framework ops (the Python -> ATen -> hipLaunchKernel path), wall time per call
    on an idle stream, and the device-side span of the empty launches (HIP events).  A box whose host is slow shows it here."""
    import torch
    from afcm_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    t = torch.zeros(1, device=dev)
    out = {}
    for name, fn in (('c_abi_noop', lambda: lib.afcm_noop(st)), ('torch_add_', lambda: t.add_(1.0))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(launches):
            fn()
        host = time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        out[name] = dict(host_us_per_launch=host / launches * 1e6, device_us_per_launch=e0.elapsed_time(e1) * 1e3 / launches)
    out['launches'] = launches
    return out


def build_step(args, dev, dtype_name, with_discriminator, use_dist, fp32_conv, capturable=False):
    """The module(s), the step object and the synthetic batch of one workload; sets the fp32 conv route for it."""
    import torch
    from afcm_amd import layer_schedule as sched
    from afcm_amd import synthetic
    from afcm_amd.networks_stylegan3 import Stylegan3Generator
    from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep
    from afcm_amd.torch_utils.ops import conv2d as conv_ops
    bf = torch.bfloat16
    conv_ops.FP32_SPLIT = {'f16x3': (torch.float16, 3, 3, 3), 'bf16x6': (bf, 6, 6, 6), 'bf16x663': (bf, 6, 6, 3), 'bf16x633': (bf, 6, 3, 3),
                           'bf16x3': (bf, 3, 3, 3), 'native': None}[fp32_conv]
    dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[dtype_name]
    comm_dtype = torch.bfloat16 if args.comm_dtype == 'bf16' else None
    rank = int(os.environ.get('RANK', 0))
    torch.manual_seed(0)      # identical init on every rank (the step also broadcasts from rank 0)
    kw = dict(sched.DEFAULT_SYNTHESIS_KWARGS)
    G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=args.res, img_channels_in=4, img_channels_out=1,
                           mapping_kwargs=dict(num_layers=8), synthesis_kwargs=dict(kw, compute_dtype=dtype)).to(dev).train()
    if with_discriminator:
        from afcm_amd.networks_discriminator import CoModDiscriminator
        from afcm_amd.stylegan3_model import StyleGAN3Step
        # the four highest-resolution blocks in the compute dtype (the reference's num_fp16_res switch, generator.py:808; conv_clamp
        # 256 as its 16-bit configurations use), the rest fp32
        n16 = 0 if dtype == torch.float32 else 4
        D = CoModDiscriminator(c_dim=0, img_resolution=args.res, img_channels=5, channel_base=int(0.5 * 32768), channel_max=512,
                               num_fp16_res=n16, conv_clamp=(256 if n16 else None), block_kwargs=dict(fp16_dtype=dtype if n16 else torch.float16),
                               epilogue_kwargs=dict(mbstd_group_size=16)).to(dev)
        step = StyleGAN3Step(G, D, lr_G=0.0025, lr_D=0.0025, lambda_L1=100.0, lambda_r1=10.0, distributed=use_dist,
                             force_collectives=args.force_dist, comm_dtype=comm_dtype, capturable=capturable)
    else:
        step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0, distributed=use_dist, force_collectives=args.force_dist,
                                      comm_dtype=comm_dtype, capturable=capturable)
    inputs = synthetic.generator_inputs(args.batch, size=args.res, seed=rank, device=dev)
    return step, inputs


def kernel_table(fams, dtype_name, fp32_conv, timed_with_events):
    """Per hot family: algorithmic work / HIP-event launch durations of this run against the guide's peaks."""
    from afcm_amd.torch_utils.ops import conv2d as conv_ops
    kernels = {}
    for fam, d in fams.items():
        avg_ms = d['total_ms'] / d['launches']
        if fam == 'filtered_lrelu':
            ach = d['work'] / (d['total_ms'] * 1e-3) / 1e9
            kernels[fam] = dict(bound='hbm', achieved=ach, peak=PEAK_HBM_GBS, unit='GB/s', frac=ach / PEAK_HBM_GBS, traffic=None,
                                launches=d['launches'], avg_launch_ms=avg_ms, total_ms=d['total_ms'], steps_timed=timed_with_events,
                                ms_per_step=d['total_ms'] / max(1, timed_with_events))
        else:
            ach = d['work'] / (d['total_ms'] * 1e-3) / 1e12
            peak, pipe = PEAK_MFMA_TFLOPS[dtype_name], None
            if dtype_name == 'fp32' and conv_ops.FP32_SPLIT is not None:
                # fp32 products from split 16-bit operands: `achieved` stays the ALGORITHMIC (fp32) flop rate; the pipe that
                # executes them is the 16-bit one, and an fp32 product costs `terms` of its multiplications
                terms = conv_ops.FP32_SPLIT[3] if fam == 'conv2d_wgrad' else max(conv_ops.FP32_SPLIT[1:3])
                peak = PEAK_MFMA_TFLOPS['fp16'] / terms
                pipe = f'{fp32_conv}: 16-bit MFMA pipe (2500 TFLOP/s dense), {terms} product terms per fp32 product'
            kernels[fam] = dict(bound='mfma', achieved=ach, peak=peak, unit='TFLOP/s', frac=ach / peak, traffic=None,
                                launches=d['launches'], avg_launch_ms=avg_ms, total_ms=d['total_ms'], steps_timed=timed_with_events,
                                ms_per_step=d['total_ms'] / max(1, timed_with_events), **({'pipe': pipe} if pipe else {}))
    return kernels


def timed_steps(step, inputs, steps, warmup, world, use_dist, kernel_timing, every, dev):
    """`warmup` untimed steps, then EXACTLY `steps` timed ones between barrier + synchronize on both sides (max over ranks).
    Returns dict(elapsed, host_wall, host_cpu, timed_with_events, families, threads, one_step)."""
    import torch
    import torch.distributed as dist
    from afcm_amd import profiling
    real_A, real_B, z, c = inputs

    def one_step():
        step.set_input(real_A, real_B, z, c)
        step.optimize_parameters()

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timed_with_events = 0
    if kernel_timing:
        profiling.start()
    host_wall = host_cpu = 0.0
    th0 = _thread_cpu_ns()
    t0 = time.perf_counter()
    for i in range(steps):
        if kernel_timing:
            profiling.enabled = (i % max(1, every) == 0)
            timed_with_events += int(profiling.enabled)
        if use_dist and step.buckets is not None and i == steps - 1:
            # the LAST timed step carries the bucket timeline: CUDA events at the phase boundaries and at every bucket's launch
            # (records only, nothing waits on them)
            step.buckets.trace, step.phase_events = [], {}
        h0, c0 = time.perf_counter(), time.process_time()
        one_step()
        host_wall += time.perf_counter() - h0      # wall time of the Python call: the host's own work while it does not run ahead into a full queue
        host_cpu += time.process_time() - c0       # CPU time of this process (all its threads) inside the call
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    th1 = _thread_cpu_ns()
    profiling.stop()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    # which threads of this process were on a CPU during the timed region (45 ms of CPU time inside a 23.6 ms call)
    busy = sorted(((th1[t][1] - th0.get(t, (None, 0))[1], th1[t][0], t) for t in th1), reverse=True)
    main_tid = threading.get_native_id()
    threads = dict(python_threads=threading.active_count(), os_threads=len(th1),
                   cpu_ms_per_step_by_thread=[dict(thread=('main (Python)' if t == main_tid else name), cpu_ms_per_step=round(ns / 1e6 / steps, 3))
                                              for ns, name, t in busy[:6] if ns > 0])
    return dict(elapsed=elapsed, host_wall=host_wall, host_cpu=host_cpu, timed_with_events=timed_with_events,
                families=profiling.summary() if kernel_timing else {}, threads=threads, one_step=one_step)


def launch_census(one_step, rank, world, skip):
    """Launches per step (rank 0, one extra untimed step under the framework's kernel tracer): total, the three hot families, the rest.
    EVERY rank runs the step whatever the tracer does on rank 0 (it carries the collectives: ADVICE r05) -- the tracer is set up first,
    and a failure to set it up falls back to the bare step."""
    import torch
    if skip:
        return dict(skipped='--no-launch-census')
    if rank != 0:
        one_step()
        torch.cuda.synchronize()
        return None
    import collections
    prof = None
    try:
        from torch.profiler import profile, ProfilerActivity
        prof = profile(activities=[ProfilerActivity.CUDA])
        prof.__enter__()
    except Exception as e:
        prof, err = None, f'{type(e).__name__}: {e}'
    one_step()                          # outside any try: an error in the step itself is the bench's error, on every rank alike
    torch.cuda.synchronize()
    if prof is None:
        return dict(error=err)
    try:
        prof.__exit__(None, None, None)
        names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and 'Memcpy' not in e.name and 'Memset' not in e.name]
        fam = collections.Counter('conv2d' if ('conv2d_fwd' in n or 'conv2d_direct' in n) else 'conv2d_wgrad' if 'conv2d_wgrad' in n else
                                  'filtered_lrelu' if ('flrelu_wave' in n or 'flrelu_mfma_kernel' in n or 'flrelu_strip' in n or 'flrelu_sep' in n) else
                                  'other_afcm' if 'afcm' in n else 'framework' for n in names)
        return dict(total=len(names), **fam, outside_the_three_families=fam['other_afcm'] + fam['framework'])
    except Exception as e:              # the tracer is an aid: never lose the bench line to it
        return dict(error=f'{type(e).__name__}: {e}')


def also_record(args, dev, name, dtype_name, with_discriminator, steps, warmup):
    """One sub-record of the driver-run line: the same step in the reference's own precision (fp32: NET:619,653) or the
    full D + G iteration, measured in this process after the headline; the headline fields do not depend on it."""
    import gc
    import torch
    try:
        step, inputs = build_step(args, dev, dtype_name, with_discriminator, False, args.fp32_conv)
        r = timed_steps(step, inputs, steps, warmup, 1, False, not args.no_kernel_timing, 2, dev)
        kernels = kernel_table(r['families'], dtype_name, args.fp32_conv, r['timed_with_events'])
        ms = r['elapsed'] / steps * 1e3
        out = dict(metric=('full GAN iteration (D update with R1 + G update) images/sec' if with_discriminator else 'generator fwd+bwd images/sec')
                   + f' @{args.res}^2', images_per_sec=args.batch * steps / r['elapsed'], ms_per_step=ms, steps=steps, warmup=warmup, dtype=dtype_name,
                   per_gpu_batch=args.batch, host_ms_per_step=r['host_wall'] / steps * 1e3,
                   kernels={k: {f: v[f] for f in ('bound', 'achieved', 'peak', 'unit', 'frac', 'ms_per_step', 'launches', 'pipe') if f in v}
                            for k, v in kernels.items()},
                   outside_the_three_families_ms=ms - sum(v['ms_per_step'] for v in kernels.values()) if kernels else None,
                   **({'fp32_conv': args.fp32_conv} if dtype_name == 'fp32' else {}))
        del step, inputs, r
    except Exception as e:              # a sub-record never takes the headline down with it
        out = dict(error=f'{type(e).__name__}: {e}')
    gc.collect()
    torch.cuda.empty_cache()
    return out


def graph_worker(args):
    """Child process of `graph_record`: build the bf16 generator step with a capturable optimizer, capture it into ONE hipGraph
    (stylegan3_model.capture_step: warm-up and capture on one side stream, before the step ever ran elsewhere), time replays.  One JSON line."""
    import torch
    from afcm_amd.stylegan3_model import capture_step
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // 8))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    steps = args.also_steps
    step, inputs = build_step(args, dev, 'bf16', args.with_discriminator, False, args.fp32_conv, capturable=True)
    graph = capture_step(step, inputs, warmup=3)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        graph.replay()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps(dict(graph_record=True, ms_per_step=el / steps * 1e3, images_per_sec=args.batch * steps / el, host_ms_per_replay=host / steps * 1e3,
                          steps=steps, optimizer_steps_on_device=step.optimizer_G.device_step(), loss_G_L1=float(step.loss_G_L1.detach()))), flush=True)


def graph_record(args, limit=240.0, with_discriminator=False):
    """The N = 1 bf16 step captured into ONE hipGraph and replayed (asked for when the eager step's host side takes >= 50 % of
    the step on the driver's box): ms per replayed step and the host's wall time per replay.  The step is GPU-bound either way at N = 1;
    what the graph removes is the host's ~20 ms of Python / autograd / launch work per step -- the margin eight ranks sharing one host live
    on.  Measured in a CHILD process (a fresh context: the capture must be the step's first use of autograd, and a capture that fails must
    not take the headline line with it)."""
    try:
        cmd = [sys.executable, os.path.abspath(__file__), '--graph-worker', '--also-steps', str(args.also_steps), '--batch', str(args.batch), '--res', str(args.res)]
        if with_discriminator:
            cmd.append('--with-discriminator')
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith('{"graph_record"')]
        if not lines:
            return dict(error=f'child exited with code {out.returncode}', stderr_tail=out.stderr[-400:])
        d = json.loads(lines[-1])
        d.pop('graph_record')
        return d
    except Exception as e:
        return dict(error=f'{type(e).__name__}: {e}'[:400])


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.res, timed=args.cpu_timed, threads=args.cpu_threads)
        return
    if args.graph_worker:
        graph_worker(args)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))
    world_env = int(os.environ.get('WORLD_SIZE', 1))
    host_threads = args.host_threads or max(1, (os.cpu_count() or 8) // (8 * world_env))
    os.environ.setdefault('OMP_NUM_THREADS', str(host_threads))      # before torch creates its pools
    import torch
    import torch.distributed as dist
    torch.set_num_threads(host_threads)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE {world}')
    # rehearsal aid for a one-GPU box: AFCM_BENCH_REHEARSE=1 puts every rank on device 0 and exchanges through gloo (RCCL refuses
    # two ranks on one device) -- exercises the multi-process path (broadcast, bucket hooks, reduced-gradient Adam), not a benchmark
    rehearse = os.environ.get('AFCM_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    have = torch.cuda.device_count()
    if local_rank >= have:
        raise SystemExit(f'bench.py rank {rank}: --gpus {args.gpus} needs {args.gpus} devices, this host shows {have} '
                         f'(AFCM_BENCH_REHEARSE=1 puts every rank on device 0 over gloo: a path check, not a benchmark)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    if args.no_homogeneous_dot:
        from afcm_amd.torch_utils.ops import fused_layer
        fused_layer.HOMOGENEOUS_DOT = False
    step, inputs = build_step(args, dev, args.dtype, args.with_discriminator, use_dist, args.fp32_conv)
    r = timed_steps(step, inputs, args.steps, args.warmup, world, use_dist, not args.no_kernel_timing, args.kernel_timing_every, dev)
    elapsed, host_wall, host_cpu, timed_with_events = r['elapsed'], r['host_wall'], r['host_cpu'], r['timed_with_events']
    r_threads, families = r['threads'], r['families']

    bucket_timeline = None
    if use_dist and step.buckets is not None and getattr(step, 'phase_events', None):
        # where in the backward pass of the last TIMED step each gradient bucket's all-reduce was issued (GPU timeline of this rank;
        # a one-rank run has no peer to exchange with, the N-rank runs overlap from these points on)
        pe = step.phase_events
        bwd_ms = pe['backward'].elapsed_time(pe['finish'])
        bucket_timeline = dict(step='last timed step', forward_ms=pe['forward'].elapsed_time(pe['backward']), backward_ms=bwd_ms,
                               finish_and_optimizer_ms=pe['finish'].elapsed_time(pe['end']),
                               buckets=[dict(bucket=i, mbytes=round(nb / 1e6, 1), issued_at_ms=round(pe['backward'].elapsed_time(e), 2),
                                             backward_left_ms=round(bwd_ms - pe['backward'].elapsed_time(e), 2)) for i, nb, e in step.buckets.trace])
        step.buckets.trace = step.phase_events = None
    launches = launch_census(r['one_step'], rank, world, args.no_launch_census)
    if rank == 0:
        kernels = kernel_table(families, args.dtype, args.fp32_conv, timed_with_events)
        dominant = max(kernels, key=lambda k: kernels[k]['total_ms']) if kernels else None
        roofline = dict(kernels[dominant], kernel=dominant) if dominant else None
        # HBM traffic: NOT measured in this run -- bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (tools/pmc_traffic.sh -> profiles/pmc_traffic.json, which names the build it was taken on); said so in `traffic_source`
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        default_workload = args.dtype == 'bf16' and args.batch == 16 and args.res == 256 and not args.with_discriminator
        if os.path.exists(tpath) and default_workload:        # the PMC passes were taken on the default command only
            try:
                t = json.load(open(tpath))
                src = f'profiles/pmc_traffic.json ({t.get("_source", "rocprofv3 --pmc passes of bench.py")}), not measured in this run'
                for fam, d in kernels.items():
                    d['traffic'] = t.get(fam)
                    d['traffic_source'] = src if t.get(fam) is not None else None
                if roofline:
                    roofline['traffic'] = t.get(dominant)
                    roofline['traffic_source'] = src if t.get(dominant) is not None else None
            except Exception:
                pass
        calibration = None
        if world == 1 and args.host_calibration == 'auto':
            try:
                calibration = host_calibration(dev)
            except Exception as e:
                calibration = dict(error=f'{type(e).__name__}: {e}')
        gradient_buckets = step.buckets.num_buckets if step.buckets is not None else 0
        bucket_mbytes = [round(b['flat'].numel() * b['flat'].element_size() / 1e6, 2) for b in step.buckets._buckets] if step.buckets is not None else []
        # the sub-records: same process, after the headline (whose model is released first)
        also = None
        if world == 1 and default_workload and args.also == 'auto' and not args.force_dist:
            del step, inputs, r
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            also = dict(fp32=also_record(args, dev, 'fp32', 'fp32', False, args.also_steps, 2),
                        d_plus_g=also_record(args, dev, 'd_plus_g', 'bf16', True, args.also_steps, 2),
                        graph=graph_record(args),
                        d_plus_g_graph=graph_record(args, with_discriminator=True))
        cpu = run_cpu_baseline(args.res) if (world == 1 and args.cpu_baseline == 'auto') else None
        images = world * args.batch * args.steps
        out = {
            'metric': f'generator fwd+bwd images/sec @{args.res}^2' if not args.with_discriminator else f'full GAN iteration (D + G update) images/sec @{args.res}^2',
            'value': images / elapsed,
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            # host side of a step (rank 0): wall time of the Python call that issues it and the CPU time the process spent in it;
            # both well below ms_per_step = the GPU is the limiter and the host runs ahead
            'host_ms_per_step': host_wall / args.steps * 1e3,
            'host_cpu_ms_per_step': host_cpu / args.steps * 1e3,
            'host_threads_busy': r_threads,
            'host_calibration': calibration,
            'launches_per_step': launches,
            'host_threads': dict(omp_num_threads=os.environ.get('OMP_NUM_THREADS'), torch_num_threads=torch.get_num_threads(),
                                 cpu_count=os.cpu_count(), ranks_on_host=world, rule='--host-threads, default cpu_count // (8 * ranks)'),
            # every AFCM_* variable set in this process's environment (they select libraries / layouts: a stray one changes what is measured)
            'env': {k: v for k, v in sorted(os.environ.items()) if k.startswith('AFCM_')},
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': ('AFCM --model stylegan3 FULL iteration (D update with R1 + G update with GAN and L1 terms), ' if args.with_discriminator else '') +
                                   f'AFCM --model stylegan3 generator training step (fwd + L1 loss + bwd + grad all-reduce + Adam), '
                                   f'IXI T1->T2 shape: {args.res}x{args.res}, 4->1 channels, full-width 58.5M-param generator, random init',
                       'global_batch': world * args.batch, 'per_gpu_batch': args.batch, 'resolution': args.res,
                       'parallelism': f'dp{world}',
                       # what the process group itself reports (a SCALE record can check that RCCL saw N ranks)
                       'world_size': dist.get_world_size() if use_dist else 1,
                       'backend': dist.get_backend() if use_dist else None,
                       'gradient_buckets': gradient_buckets, 'gradient_bucket_mbytes': bucket_mbytes,
                       'comm_dtype': args.comm_dtype, 'homogeneous_dot': not args.no_homogeneous_dot,
                       **({'fp32_conv': args.fp32_conv} if args.dtype == 'fp32' else {})},
            'roofline': roofline,
            'kernels': kernels,
            # milliseconds of the step outside the three hot families
            'outside_the_three_families_ms': (elapsed / args.steps * 1e3 - sum(k['ms_per_step'] for k in kernels.values())) if kernels else None,
            'cpu_baseline': cpu,
        }
        if also is not None:
            out['also'] = also
        if bucket_timeline is not None:
            out['bucket_timeline'] = bucket_timeline
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
