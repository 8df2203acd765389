#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the StyleGAN2 few-shot loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5]

One "step" is one training iteration of the few-shot loop on synthetic data: Gmain
and Dmain every iteration, Greg (path length, half batch) every 4th, Dreg (R1) every
16th, Adam steps, generator EMA - at a fixed per-GPU batch (weak scaling; the global
batch is per-GPU batch x N).  Default workload: FFHQ-256 Affine+ at batch 32 per GPU
with the four highest resolutions in bfloat16.  Inputs are resident in HBM before
the timed region starts.  Rank 0 prints ONE JSON line.

The same line carries the 1024^2 half of the metric under "workloads" (FFHQ-1024 AffineLight+, batch 16 per GPU,
timed the same way after the headline workload).

For N > 1 either launch with torch.distributed.run (one rank per GPU, RCCL), or run `python bench.py --gpus N`
as is: without WORLD_SIZE in the environment the process starts the N ranks itself, as the reference's entry
point spawns one process per GPU (train.py:691-696), and exits with the first non-zero status among them.
"""
import argparse
import json
import os
import platform
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak
# dense matrix-core peaks.  float32 convolutions run on the 16-bit matrix instruction, three partial products per float
# product (both operands cut into two half-precision pieces, the low x low product dropped): their ceiling in
# float-equivalent flops is a third of the 16-bit peak.
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'float': 2500.0 / 3}

CONFIGS = {
    # name: (description, cfg preset, resolution, per-GPU batch, parameterisation)
    'c2': ('FFHQ-256 Affine+ bf16, batch=32 per GPU', 'paper256', 256, 32, 'AFFINE_PLUS'),
    'c3': ('FFHQ-1024 AffineLight+ bf16, batch=16 per GPU', 'stylegan2', 1024, 16, 'AFFINE_LIGHT_PLUS'),
    'c5': ('AFHQ-512 Affine+ bf16, batch=8 per GPU', 'stylegan2', 512, 8, 'AFFINE_PLUS'),
    'c4': ('FFHQ-1024 full-generator fine-tune bf16, batch=8 per GPU', 'stylegan2', 1024, 8, 'FULL'),
}


# Counter passes of a profiler that serialises every dispatch: keep the queue short (the figure such a run prints is not a result).
SYNC_EVERY_STEP = bool(os.environ.get('SG_BENCH_SYNC_STEPS'))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--also', default='c3', help='second workload timed after the headline one and reported under '
                    '"workloads" (the 1024^2 half of the metric); "none" to skip')
    ap.add_argument('--also-steps', type=int, default=0, help='timed steps of the second workload (default: --steps)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--ada', action='store_true',
                    help='adaptive discriminator augmentation (bgc pipe, target 0.6) in front of every D call; the '
                         'headline figure is quoted without it')
    ap.add_argument('--graphs', action='store_true',
                    help='record the forward/backward of every phase into a HIP graph at the first step and replay it '
                         'afterwards (same speed on one GPU, where the step is GPU-bound; spares the host)')
    return ap.parse_args()


def make_trainer(args, rank, world, config=None):
    from ga_gan_amd.training import training_loop
    desc, cfg, res, batch_gpu, par = CONFIGS[config or args.config]
    opts = training_loop.setup_training_options(cfg=cfg, resolution=res, batch_size=batch_gpu * world, num_gpus=world,
                                                batch_gpu=batch_gpu, glrate=0.02 if par != 'FULL' else None,
                                                **getattr(training_loop, par))
    extra = {}
    if getattr(args, 'ada', False):
        extra = dict(augment_pipe=training_loop.make_augment_pipe('bgc', 0.3, torch.device('cuda', torch.cuda.current_device())),
                     ada_target=0.6)
    return (training_loop.Trainer(rank=rank, random_seed=0, use_graphs=bool(getattr(args, 'graphs', False)), **extra, **opts),
            desc + (' + ADA (bgc)' if extra else ''), res, batch_gpu)


def traffic_file(config):
    """The newest committed counter summary of the workload (profiles/rN_traffic_<config>.json), or None."""
    names = sorted(n for n in os.listdir(os.path.join(ROOT, 'profiles')) if n.endswith('_traffic_%s.json' % config)) \
        if os.path.isdir(os.path.join(ROOT, 'profiles')) else []
    return os.path.join(ROOT, 'profiles', names[-1]) if names else None


def measured_traffic(label, config):
    """HBM-side bytes per launch of the kernel `label` (symbol<dtype>) from the committed counter runs of this same
    workload (profiles/rN_traffic_<config>.json, written by tools/summarize_traffic.py from rocprofv3 FETCH_SIZE /
    WRITE_SIZE passes); None when the file has no entry.  All template instances of the symbol and dtype are pooled."""
    path = traffic_file(config)
    if path is None:
        return None
    symbol, _, dtype = label.partition('<')
    dtype = dtype.rstrip('>').split(',')[0]
    total, launches = 0.0, 0
    for name, v in json.load(open(path))['kernels'].items():
        if name.startswith(symbol + '<') and name[len(symbol) + 1:].startswith(dtype):
            total += v['launches'] * (v['fetch_bytes_per_launch'] + v['write_bytes_per_launch'])
            launches += v['launches']
    return int(total / launches) if launches else None


def traffic_source(label, config):
    """Where `roofline.traffic` comes from: it is not measured in this run (counters need rocprofv3 passes of their own)."""
    path = traffic_file(config)
    if path is None or measured_traffic(label, config) is None:
        return None
    return 'profiles/%s: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this workload, per launch' % os.path.basename(path)


def kernel_roofline(trainer, real, real_c, steps, config):
    """Instrumented pass: HIP events around every launch of the hand-written kernels (recorded on the
    stream they are launched on), grouped by kernel symbol.  The kernel with the largest total time is the
    dominant one: for the matrix-core convolutions it is priced against the dense MFMA peak of its dtype
    (algorithmic flops per launch / average launch time), for the streaming kernels against HBM
    (algorithmic bytes per launch = operands read once + result written once)."""
    from ga_gan_amd import _lib
    graphs, trainer.use_graphs = trainer.use_graphs, False   # events are recorded around host launches
    _lib.profile_begin()
    for _ in range(steps):
        trainer.step(real, real_c)
    records = _lib.profile_end()
    trainer.use_graphs = graphs
    if not records:
        return None, {}
    by_kernel = {}
    for name, nbytes, ms, flops in records:
        k = by_kernel.setdefault(name, dict(launches=0, bytes=0, ms=0.0, flops=0.0))
        k['launches'] += 1
        k['bytes'] += nbytes
        k['ms'] += ms
        k['flops'] += flops
    name, k = max(by_kernel.items(), key=lambda kv: kv[1]['ms'])
    avg_ms = k['ms'] / k['launches']
    if k['flops'] > 0:
        peak = MFMA_PEAK_TFLOPS['float' if '<float' in name else 'bf16']
        achieved = (k['flops'] / k['launches']) / (avg_ms * 1e-3) / 1e12
        roofline = dict(bound='mfma', kernel=name, achieved=round(achieved, 2), peak=peak, unit='TFLOP/s',
                        frac=round(achieved / peak, 4), traffic=measured_traffic(name, config), traffic_source=traffic_source(name, config), launches=k['launches'],
                        avg_launch_us=round(avg_ms * 1e3, 2), algorithmic_flops_per_launch=int(k['flops'] / k['launches']),
                        algorithmic_bytes_per_launch=int(k['bytes'] / k['launches']))
    else:
        achieved = (k['bytes'] / k['launches']) / (avg_ms * 1e-3) / 1e9
        roofline = dict(bound='hbm', kernel=name, achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                        frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=measured_traffic(name, config), traffic_source=traffic_source(name, config), launches=k['launches'],
                        avg_launch_us=round(avg_ms * 1e3, 2), algorithmic_bytes_per_launch=int(k['bytes'] / k['launches']))
    summary = {n: dict(launches_per_step=v['launches'] // steps, ms_per_step=round(v['ms'] / steps, 3),
                       gbps=round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1),
                       tflops=round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 2))
               for n, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]['ms'])[:8]}   # the eight heaviest symbols
    return roofline, summary


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def cpu_baseline():
    """The oracle (CPU restatement of the reference's fallback path) on this host's cores: FFHQ-256 Affine+,
    batch 4, float32 (the CPU-runnable configuration of the reference).  Bounded sample: 3 warm-up passes, then 3 timed passes of each of the
    four phases on all cores (an average iteration costs Gmain + Dmain + Greg/4 + Dreg/16), and one Gmain pass on
    a single thread, from which the single-thread rate is scaled."""
    from ga_gan_amd.training import networks, training_loop
    from oracle import objectives
    from oracle import stylegan2 as sg
    cores = min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', '16')))
    torch.set_num_threads(cores)
    opts = training_loop.setup_training_options(cfg='paper256', resolution=256, batch_size=4, fp32=True,
                                                **training_loop.AFFINE_PLUS)
    torch.manual_seed(0)
    G = networks.Generator(**opts['G_kwargs'])
    D = networks.Discriminator(**opts['D_kwargs'])
    gp = {k: v.detach().clone() for k, v in G.state_dict().items()}
    dp = {k: v.detach().clone() for k, v in D.state_dict().items()}
    g_names = objectives.select_trainable([n for n, _ in G.named_parameters()], opts['requires_grad_parts']['Gmain'])
    d_names = [n for n, _ in D.named_parameters()]
    del G, D
    gcfg = sg.GeneratorConfig(img_resolution=256, channel_base=16384, use_domain_modulation=True,
                              domain_modulation_parametrization='out_in_additive')
    dcfg = sg.DiscriminatorConfig(img_resolution=256, channel_base=16384, mbstd_group_size=4)
    obj = objectives.TrainingObjective(gp, gcfg, dp, dcfg, r1_gamma=1.0)
    z = torch.randn(4, 512, generator=torch.Generator().manual_seed(0))
    c = torch.zeros(4, 0)
    real = torch.randint(0, 256, [4, 3, 256, 256], generator=torch.Generator().manual_seed(1)).float() / 127.5 - 1

    def one_pass(name, gain):
        params, names = (gp, g_names) if name[0] == 'G' else (dp, d_names)
        for t in list(gp.values()) + list(dp.values()):
            t.requires_grad_(False)
            t.grad = None
        for n in names:
            params[n].requires_grad_(True)
        t0 = time.perf_counter()
        obj.accumulate_gradients(name, real, c, z, c, gain)
        return time.perf_counter() - t0

    gains = dict(Gmain=1, Dmain=1, Greg=4, Dreg=16)
    for name in ('Gmain', 'Dmain', 'Gmain'):
        one_pass(name, gains[name])
    timed = 3
    times = {name: sum(one_pass(name, gain) for _ in range(timed)) / timed for name, gain in gains.items()}
    per_iter = times['Gmain'] + times['Dmain'] + times['Greg'] / 4 + times['Dreg'] / 16
    torch.set_num_threads(1)
    gmain_1t = one_pass('Gmain', 1)
    torch.set_num_threads(cores)
    return dict(value=round(4 / per_iter, 4), unit='img/s', cores=cores, kind='port',
                value_1_thread=round(4 / (per_iter * gmain_1t / times['Gmain']), 4), cpu=cpu_model(), torch=torch.__version__,
                sample='FFHQ-256 Affine+ batch 4 fp32 on CPU: 3 warm-up passes + %d timed passes of each phase on %d threads '
                       '(Gmain %.2fs, Dmain %.2fs, Greg %.2fs, Dreg %.2fs); iteration = Gmain+Dmain+Greg/4+Dreg/16; '
                       '1 thread: one Gmain pass (%.2fs), iteration scaled by the same ratio'
                       % (timed, cores, times['Gmain'], times['Dmain'], times['Greg'], times['Dreg'], gmain_1t))


def visible_gpus():
    """GPUs this process may use, counted WITHOUT the HIP runtime (the parent must stay free of it: it starts the
    ranks): the nodes of the kernel driver's topology that have SIMDs, narrowed by the *_VISIBLE_DEVICES lists."""
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if os.environ.get(var) is not None:
            return len([d for d in os.environ[var].split(',') if d.strip() != ''])
    count = 0
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get('simd_count', 0)) > 0:
                count += 1
    except OSError:
        return None            # no topology to read: let the ranks find out
    return count


def launch_ranks(args):
    """`--gpus N` without a launcher: start one rank per GPU from this process, which never touches the HIP
    runtime (every rank checks for itself that its GPU exists), wait for all of them, exit with the first failure."""
    have = visible_gpus()
    if have is not None and have < args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} needs {args.gpus} GPUs: {have} visible')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', str(port)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    for proc in procs:
        rc = proc.wait()
        if rc != 0 and status == 0:
            status = rc
            for other in procs:          # a rank that died leaves the others waiting in a collective
                if other.poll() is None:
                    other.terminate()
    return status


def run_workload(args, config, steps, warmup, rank, world, local_rank, note, with_roofline):
    """Warm up, time `steps` iterations between barriers, MAX over ranks.  -> dict for the JSON line."""
    trainer, desc, res, batch_gpu = make_trainer(args, rank, world, config)
    gen = torch.Generator(device=trainer.device).manual_seed(1 + rank)
    real, real_c = trainer.synthetic_batch(gen)          # resident in HBM before timing

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    note(f'{desc}; world {world}; warm-up {warmup} steps')
    for i in range(warmup):
        trainer.step(real, real_c)
        torch.cuda.synchronize()
        note(f'warm-up step {i} done')
    barrier()
    note(f'timing {steps} steps')
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(real, real_c)
        if SYNC_EVERY_STEP:
            torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=trainer.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    note(f'timed region: {elapsed:.3f}s')
    roofline, kernels = (None, {})
    if with_roofline:
        roofline, kernels = kernel_roofline(trainer, real, real_c, min(steps, 4), config)
        barrier()
    global_batch = batch_gpu * world
    out = dict(value=round(steps * global_batch / elapsed, 3), unit='img/s', steps=steps, warmup=warmup,
               ms_per_step=round(elapsed / steps * 1e3, 3),
               config=dict(workload=desc, resolution=res, global_batch=global_batch, per_gpu_batch=batch_gpu,
                           parallelism=f'dp{world}', hip_graphs=bool(trainer.use_graphs),
                           iteration='Gmain+Dmain every step, Greg every 4th, Dreg every 16th',
                           activations='bf16 in the four highest resolutions (images within 3e-2 relative L2 of the float32 '
                                       'reference path; float32 runs within 1e-3)',
                           kernels='conv2d / conv3x3 / band3x3 / wgrad (MFMA), 1x1 to-RGB / from-RGB streaming kernels, upfirdn2d, '
                                   'layer epilogues, bias_act, minibatch-std: hand-written HIP; small GEMMs: torch'),
               roofline=roofline, kernels=kernels,
               grad_exchange_bytes_per_step=int(trainer.exchanged_bytes / max(trainer.batch_idx, 1)))
    del trainer, real, real_c
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE is {world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no GPU is visible')
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'bench.py --gpus {args.gpus} needs {args.gpus} GPUs: {torch.cuda.device_count()} visible')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    torch.backends.cudnn.benchmark = os.environ.get('SG_MIOPEN_FIND', '0') == '1'

    from ga_gan_amd.torch_utils import training_stats
    t_start = time.perf_counter()

    def note(msg):   # progress on stderr (the JSON line is the only thing on stdout)
        if rank == 0:
            print(f'[bench +{time.perf_counter() - t_start:7.1f}s] {msg}', file=sys.stderr, flush=True)

    main_wl = run_workload(args, args.config, args.steps, args.warmup, rank, world, local_rank, note, not args.no_roofline)
    workloads = {}
    if args.also != 'none' and args.also != args.config:
        also_steps = args.also_steps or args.steps
        wl = run_workload(args, args.also, also_steps, min(args.warmup, 3), rank, world, local_rank, note, not args.no_roofline)
        workloads[args.also] = wl
    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        training_stats.enable(False)
        note('CPU baseline (oracle) ...')
        baseline = cpu_baseline()
        note('CPU baseline done')

    if rank == 0:
        out = dict(metric='training images/sec', value=main_wl['value'], unit='img/s', n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=main_wl['ms_per_step'], higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='bf16', data='synthetic', config=main_wl['config'], roofline=main_wl['roofline'], cpu_baseline=baseline,
                   kernels=main_wl['kernels'], workloads=workloads,
                   grad_exchange_bytes_per_step=main_wl['grad_exchange_bytes_per_step'])
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
