#!/usr/bin/env python3
"""Headline benchmark: StyleGAN-V G+D training-step throughput at 256^2 on N MI355X (one node).

    python bench.py --gpus 1 --steps 16 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "FFS 256x256 full G+D train step + PL/R1 reg, bs=32, fp32"):
FaceForensics-config generator/discriminator (cfg=auto: fmaps 0.5, mapping depth 2, 3 frames per
video), random-init weights, synthetic frames and latents, fp32.  A "step" is one iteration of the
reference's phase schedule (src/training/training_loop.py:351-404): Gmain every iteration, Greg every
4th (a no-op because configs/model/stylegan-v.yaml sets pl_weight 0 -- and the reference's PL term
cannot run with 3 frames per video), Dmain every iteration, Dreg (R1, double backward)
every 16th, each followed by nan_to_num + Adam, then the G_ema update.  Per-GPU batch is fixed
(weak scaling): `--batch-gpu` videos x 3 frames per rank; DDP all-reduces G/D gradients over RCCL.

The single JSON line carries, besides the contract fields:
  roofline      the step's dominant hand-written kernel (largest summed time: conv3x3_ws_kernel, the stride-1 3x3 convolution): algorithmic flops /
                HIP-event time summed over its launches inside the second half of the timed steps (events recorded by the C ABI on the launch stream;
                on every step they cost 2.5-4 % of the step, profiles/r02_bench_prof_overhead.log)
  roofline_conv_family  the same over the whole 3x3 convolution family (what earlier rounds' lines reported under `roofline`)
  roofline_upfirdn2d  the same for the upfirdn2d lane-exchange kernel against the HBM roofline (second half of BASELINE.json's metric)
  kernels       the same accounting for every native kernel family
  cpu_baseline  the same training step on the host CPU through the plain-PyTorch op path (a restatement
                of the reference's CPU fallback ops), on a bounded sample (1 video = 3 frames per step): one main iteration and one R1 iteration are timed
                and weighted by the schedule (Dreg every 16th)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4-copy ceiling 6290 GB/s
HBM_COPY_GBPS = 6290.0
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md); the fp32-input MFMA peaks at 157.3
MFMAS_PER_PRODUCT = {1: 1, 3: 3, 4: 3}   # conv2d_gradfix.native_conv_terms -> 16-bit MFMAs per fp32 product


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class PowerSampler:
    """Engine clock, socket power and hot-spot temperature of ONE device while a region runs: the amdgpu hwmon files of the card whose PCI address is the HIP device's,
    read every `period` seconds on a thread.  Evidence for the regime the kernels run in (the step sits at the package power cap, far below the
    2.4 GHz the MFMA peak is quoted at); never an input to `value`.  Silent when the files are not there."""

    FILES = (('sclk_MHz', 'freq1_input', 1e-6), ('socket_W', 'power1_input', 1e-6), ('socket_W', 'power1_average', 1e-6), ('hotspot_C', 'temp2_input', 1e-3), ('cap_W', 'power1_cap', 1e-6))

    def __init__(self, device_index=0, period=0.02):
        import glob
        self.period, self.files, self.rows, self.on, self.card = period, {}, [], False, None
        try:
            p = torch.cuda.get_device_properties(device_index)
            want = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}'
        except Exception:      # noqa
            return
        for card in sorted(glob.glob('/sys/class/drm/card*/device')):
            if not os.path.basename(os.path.realpath(card)).lower().startswith(want):
                continue
            for hw in glob.glob(os.path.join(card, 'hwmon', 'hwmon*')):
                for key, name, scale in self.FILES:
                    path = os.path.join(hw, name)
                    if key not in self.files and os.path.exists(path):
                        self.files[key] = (path, scale)
            self.card = os.path.basename(os.path.dirname(card)) + ' @ ' + os.path.basename(os.path.realpath(card))

    def _loop(self):
        while self.on:
            row = {}
            for key, (path, scale) in self.files.items():
                try:
                    with open(path) as fh:
                        row[key] = float(fh.read().strip()) * scale
                except Exception:      # noqa
                    pass
            self.rows.append(row)
            time.sleep(self.period)

    def __enter__(self):
        import threading
        self.rows, self.on = [], bool(self.files)
        self.thread = threading.Thread(target=self._loop, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.on = False
        self.thread.join()

    def summary(self):
        """{key: median, key_min, key_max} over the samples of the last region; None without samples."""
        if not self.rows or not self.files:
            return None
        out = dict(samples=len(self.rows), card=self.card)
        for key in self.files:
            v = sorted(r[key] for r in self.rows if key in r)
            if v:
                out[key] = round(v[len(v) // 2], 1)
                if key != 'cap_W':
                    out[key + '_min'], out[key + '_max'] = round(v[0], 1), round(v[-1], 1)
        return out


def _round_floats(o, sig=6):
    """Floats to `sig` significant digits (the compact line must survive an 8 KB tail: 17-digit floats are a third of it)."""
    if isinstance(o, float):
        return float(f'{o:.{sig}g}')
    if isinstance(o, dict):
        return {k: _round_floats(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_floats(v, sig) for v in o]
    return o


ROOFLINE_KEYS = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'algorithmic_flops_per_launch', 'launches', 'avg_launch_us',
                 'frac_of_measured_copy_peak')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'cpu', 'sample')
COMPACT_LIMIT = 6000     # bytes; the driver keeps an 8 KB tail of stdout (the 21.8 KB line of round 4 did not parse)


def compact_line(out, detail_path=None):
    """The ONE JSON line of the contract, built from the full result `out`: contract keys, `dtype`, `config`, the `roofline` objects (numbers only),
    `cpu_baseline` and the companion values as scalars.  Everything else (per-variant / per-size / per-family tables, the companions' details, notes and
    provenance strings) goes to the side file `detail_path` and to stderr.  Guaranteed < COMPACT_LIMIT bytes: optional parts are dropped in a fixed
    order if a future field makes it grow, and the contract keys alone are a few hundred bytes."""
    def pick(d, keys):
        return None if d is None else {k: d[k] for k in keys if k in d}
    line = {k: out.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    cfg = dict(out.get('config') or {})
    line['config'] = {k: cfg[k] for k in ('workload', 'videos_per_gpu', 'frames_per_video', 'clips_per_gpu', 'frames_per_clip', 'global_batch_videos', 'parallelism', 'phases_run',
                                          'hip_graphs', 'headline_mode', 'conv_terms', 'native_launches_per_step') if k in cfg}
    line['roofline'] = pick(out.get('roofline'), ROOFLINE_KEYS)
    cpu = pick(out.get('cpu_baseline'), CPU_KEYS)
    if cpu and len(str(cpu.get('sample', ''))) > 400:
        cpu['sample'] = cpu['sample'][:397] + '...'
    line['cpu_baseline'] = cpu
    optional = [('roofline_upfirdn2d', pick(out.get('roofline_upfirdn2d'), ROOFLINE_KEYS)), ('roofline_conv_family', pick(out.get('roofline_conv_family'), ROOFLINE_KEYS))]
    optional += [(k, out[k]) for k in sorted(out) if k.startswith('value_') and not isinstance(out[k], (dict, list))]
    if out.get('multi_gpu'):
        optional.append(('multi_gpu', out['multi_gpu']))
    if out.get('power'):       # engine clock / socket power sampled over the timed region (hwmon of the device's own card)
        optional.append(('power', {k: v for k, v in out['power'].items() if k in ('sclk_MHz', 'socket_W', 'cap_W', 'hotspot_C', 'samples')}))
    optional.append(('detail', detail_path))
    for k, v in optional:
        line[k] = v
    line = _round_floats(line)
    text = json.dumps(line, separators=(',', ':'))
    for k, _ in reversed(optional):            # never reached with today's fields (~2.5 KB); the limit is a hard guarantee all the same
        if len(text) < COMPACT_LIMIT:
            break
        line.pop(k, None)
        text = json.dumps(line, separators=(',', ':'))
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


def emit(out):
    """Write the full result to bench_detail.json (repo root; also gpurun_out/ when it exists, which travels back from the GPU box) and to stderr, then
    print the compact contract line as the LAST line of stdout."""
    name = 'bench_detail.json'
    paths = [os.path.join(ROOT, name)]
    if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        paths.append(os.path.join(ROOT, 'gpurun_out', name))
    full = json.dumps(out)
    written = None
    for path in paths:
        try:
            with open(path, 'w') as fh:
                fh.write(full + '\n')
            written = written or os.path.relpath(path, ROOT)
        except OSError as err:
            log(f'[bench] could not write {path}: {err}')
    log('[bench] full result (tables, companions, provenance):')
    log(full)
    sys.stdout.flush()
    print(compact_line(out, written), flush=True)


TIMED_FAMILIES = ('conv3x3_s1', 'upfirdn2d_lanes')      # kernel families whose launches are bracketed by HIP events inside the timed region (eager headline)
CAPTURED_FAMILIES = ('conv3x3_s1',)                     # captured headline: the family whose kernel writes its own timestamps inside the replayed graphs
PMC_FILES = ['r06_pmc_bench_step_FETCH_WRITE.json']   # collected by tools/gpu_recipes/r06_final.sh (separate --pmc passes of this command), stamped with the kernel sources' digest


PMC_NOT_THIS_WORKLOAD = None      # set when the run is not the workload the committed counter passes were collected on


def pmc_traffic(prefixes, dword_read_prefixes=(), files=None):
    """HBM-side (L2 miss) bytes per launch of a kernel family from the newest committed rocprofv3 PMC summary
    (tools/pmc_summary.py over separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command: bench.py cannot
    collect counters on itself).  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced read,
    so reads are doubled for kernels that read with 16-B loads (calibrated on a float4 copy,
    profiles/r01_pmc_headline_call_FETCH_WRITE.json); kernels reading with dword loads are taken as is; WRITE_SIZE is exact.
    Returns (bytes per launch or None, description of the source incl. the commit the counters were collected on)."""
    from stylegan_v_amd.torch_utils import custom_ops
    if PMC_NOT_THIS_WORKLOAD and files is None:
        return None, PMC_NOT_THIS_WORKLOAD
    stale = None
    for fname in (files or PMC_FILES):
        path = os.path.join(ROOT, 'profiles', fname)
        try:
            with open(path) as fh:
                d = json.load(fh)
            # counters of OTHER kernels are not this run's traffic: the file must have been collected on the kernel sources that are
            # being timed -- md5 over csrc/*.hip, *.h and include/sgv_ops.h, as the in-tree build's own staleness check
            if d.get('csrc_digest') != custom_ops.source_digest():
                stale = f"profiles/{fname} was collected on other kernel sources (digest {str(d.get('csrc_digest'))[:8]} != {custom_ops.source_digest()[:8]}): not used"
                continue
            kb = n = 0.0
            for name, e in d['FETCH_SIZE'].items():
                if name.startswith(tuple(prefixes)):
                    kb += (1.0 if name.startswith(tuple(dword_read_prefixes)) and dword_read_prefixes else 2.0) * e['total_KB'] + d['WRITE_SIZE'][name]['total_KB']
                    n += e['launches']
            if n:
                src = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, profiles/{fname}, collected on commit {d.get('commit', 'of round 1 (before the XCD-aware tile order)')}"
                return kb * 1024.0 / n, src
        except (OSError, KeyError, ValueError):
            continue
    return None, stale or 'no PMC summary committed'


def pmc_traffic_conv_family():
    # every member reads with 16-byte loads now (the one-role-per-wave strided kernel, SGV_S2_WS=0, read dwords)
    return pmc_traffic(('conv3x3_ws_kernel', 'conv3x3_kernel', 'conv3x3_small_kernel', 'convT3x3_s2_ws_kernel', 'convT3x3_s2_kernel', 'conv3x3_s2_pairs_kernel', 'conv3x3_s2_ws_kernel',
                        'conv3x3_s2_kernel'), dword_read_prefixes=('conv3x3_s2_kernel',))


def pmc_traffic_per_launch(prefixes=('upfirdn2d_lanes', 'upfirdn2d_tile', 'upfirdn2d_down2_tile', 'upfirdn2d_up2_tile'), files=None):
    return pmc_traffic(tuple(prefixes), files=files)


def cpu_baseline(res, frames, seconds_cap):
    """The same training step on the host cores through the plain-PyTorch op path, batch of ONE video (3 frames) per step.
    `kind: port`: this repo's CPU path (the reference's own CPU behaviour restated: `impl='ref'` ops + ATen convolutions, tests/test_compat_reference.py) --
    the reference checkout itself cannot travel to the GPU box.  Protocol: all physical cores, one warm-up
    iteration, then >= 3 timed main iterations (median); the R1 iteration of the lazy-regularisation schedule is timed once (its own warm-up would double
    the leg) and enters with its weight 1/16."""
    from stylegan_v_amd.training import config as cfgs
    from stylegan_v_amd.training.train_step import TrainStep
    logical = os.cpu_count() or 1
    allowed = logical
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    smt = 1
    try:
        with open('/sys/devices/system/cpu/cpu0/topology/thread_siblings_list') as fh:
            smt = max(1, len(fh.read().replace('-', ',').split(',')))
    except OSError:
        pass
    physical = max(1, min(allowed, logical // smt))      # one thread per physical core
    t_start = time.time()
    # The reference ITSELF where its checkout exists (the build container: tools/cpu_reference_step.py imports its modules and drives them as training_loop.py does);
    # on the GPU box it does not, and the leg is this repo's port of the same step.  profiles/r06_cpu_baseline_reference_vs_port_container.json: both, one process.
    kind = 'port'
    one = None
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import cpu_reference_step as _ref
        if _ref.available() and res == 256:
            one = _ref.make_reference_step(res, frames)
            kind = 'reference'
    except Exception as exc:      # noqa: BLE001
        log(f'[bench] the reference checkout is present but could not be driven ({type(exc).__name__}: {exc}); timing the port')
        one = None
    if one is None:
        g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=res, batch_size=1, num_gpus=1, fp32=True, num_frames_per_video=frames)
        ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device='cpu', batch_gpu=1, world_size=1, ddp=False)

        def one(batch_idx):
            ts.batch_idx = batch_idx
            t1 = time.time()
            phases = ts.step()
            return time.time() - t1, phases
    # Thread count: every physical core is the protocol -- but at one video per step more threads are not faster on a two-socket host (measured on the
    # pool's 2 x 64-core EPYC 9575F: 9.7 s per main iteration with 128 threads, 5.7 s with 64).  Both are tried once (after the warm-up iterations) and the
    # faster one is the baseline: the CPU gets its best configuration.  TWO warm-up iterations (with one, the first timed sample was still warming:
    # 5.1 / 4.58 / 4.52 s), and the probe samples are not reused as timed ones.
    candidates = [physical] + ([64] if physical > 64 else [])
    torch.set_num_threads(candidates[0])
    t_warm, phases_main = one(1)                       # warm-up: main phases (Gmain + Dmain; Greg is a no-op at pl_weight 0)
    t_warm2 = one(1)[0]
    probe = {}
    for th in candidates:
        torch.set_num_threads(th)
        probe[th] = one(1)[0]
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    reps = []
    while len(reps) < 3 or (len(reps) < 5 and time.time() - t_start + 3.5 * reps[0] < seconds_cap):
        reps.append(one(1)[0])
    t_main = sorted(reps)[len(reps) // 2]
    t_reg, phases_reg = None, None
    if time.time() - t_start + 3.2 * t_main < 2.5 * seconds_cap:
        t_reg, phases_reg = one(16)
    per_iter = t_main + (max(t_reg - t_main, 0.0) / 16.0 if t_reg is not None else 0.0)
    model = ''
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    spent = time.time() - t_start
    return dict(value=frames / per_iter, unit='img/s', cores=threads, host_physical_cores=physical, host_logical_cpus=logical, threads_per_core=smt, kind=kind, cpu=model,
                seconds_main_iteration_by_threads={str(k): round(v, 3) for k, v in probe.items()},
                kind_note=("the reference's own modules, loss and update order (tools/cpu_reference_step.py), its pure-Python fallback ops" if kind == 'reference' else
                           "this repo's plain-PyTorch op path (= the reference's impl='ref' fallback ops + ATen CPU convolutions); the reference checkout does not exist on the GPU box "
                           "(both timed side by side in the build container: profiles/r06_cpu_baseline_reference_vs_port_container.json)"),
                seconds_main_iteration_median=t_main, main_iteration_samples=[round(v, 3) for v in reps], seconds_warmup_iterations=[round(t_warm, 3), round(t_warm2, 3)], seconds_reg_iteration=t_reg,
                sample=f'batch 1 video x {frames} frames, {res}x{res}, fp32, {threads} threads (the faster of {candidates} on {physical} physical cores): 2 warm-ups + {len(reps)} timed main iterations '
                       f'({"+".join(phases_main)}), median {t_main:.2f} s' + (f'; one R1 iteration ({"+".join(phases_reg)}) {t_reg:.1f} s, weighted 1/16: rate = frames / (t_main + (t_16 - t_main) / 16)'
                                                                       if t_reg is not None else '; the R1 iteration did not fit the budget and is not charged') + f'; {spent:.0f} s of CPU time')


def synthesis_workload(args, world, rank, device):
    """BASELINE configs[4] (`g1024`: SkyTimelapse 1024x1024 synthesis, 16-frame clips, min_period_len 256, 8 clips over 8 GPUs = one clip per
    GPU and step) / configs[1] (`g256`: FFS 256x256 generator forward, 32 videos x 3 frames): the reference's generation loop
    (src/scripts/generate.py:43-145 -> Generator.forward in eval mode, networks.py:370-401) on synthetic latents, replicas only (no collective).
    A step = one batch of clips through G; value = frames/s over all ranks.  `roofline` = the upfirdn2d chain (HBM-bound; 16 calls per forward
    at 1024^2), with a per-size table; per-family accounting as in the training workload."""
    from stylegan_v_amd.torch_utils import custom_ops
    from stylegan_v_amd.training import config as cfgs
    from stylegan_v_amd.training.networks import Generator
    res, frames = (1024, 16) if args.workload == 'g1024' else (256, 3)
    clips = args.clips_gpu if args.workload == 'g1024' else args.batch_gpu
    g_kwargs, _, _ = cfgs.model_kwargs(resolution=res, batch_size=clips * world, num_gpus=world, min_period_len=256 if res == 1024 else 16, num_frames_per_video=frames)
    torch.manual_seed(rank)
    G = Generator(**g_kwargs).to(device).eval().requires_grad_(False)
    z, c = torch.randn([clips, 512], device=device), torch.zeros([clips, 0], device=device)
    t = torch.arange(frames, device=device, dtype=torch.float32).unsqueeze(0).repeat(clips, 1) if res == 1024 else \
        torch.sort(torch.rand([clips, frames], device=device) * 100, dim=1).values

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(args.warmup):
            G(z, c, t)
        launches0 = custom_ops.launch_count()
        if not args.no_prof:
            custom_ops.prof_enable(1 << 16)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = G(z, c, t)
        barrier()
        elapsed = time.perf_counter() - t0
        records = []
        if not args.no_prof:
            custom_ops.prof_disable()
            records = custom_ops.prof_collect_records(1 << 16)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                G(z, c, t)
            barrier()
            clean = time.perf_counter() - t1
        else:
            clean = elapsed
    assert img.shape == (clips * frames, 3, res, res) and bool(torch.isfinite(img).all())
    t_max = torch.tensor([elapsed, clean], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    elapsed, clean = float(t_max[0]), float(t_max[1])
    if rank == 0:
        fam, sizes = {}, {}
        for name, ms, nbytes, nflops in records:
            e = fam.setdefault(name, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
            e['launches'] += 1; e['ms'] += ms; e['bytes'] += nbytes; e['flops'] += nflops
            if name == 'upfirdn2d_lanes':
                g = sizes.setdefault(int(nbytes), [0, 0.0])
                g[0] += 1; g[1] += ms
        kernels = {n: dict(launches=e['launches'], ms_total=e['ms'], avg_us=1e3 * e['ms'] / e['launches'],
                           **({'GBps': e['bytes'] / (e['ms'] * 1e-3) / 1e9} if e['bytes'] > 0 else {}), **({'TFLOPs': e['flops'] / (e['ms'] * 1e-3) / 1e12} if e['flops'] > 0 else {}))
                   for n, e in fam.items() if e['launches']}
        roofline, by_size = None, None
        if 'upfirdn2d_lanes' in fam:
            r = fam['upfirdn2d_lanes']
            achieved = r['bytes'] / (r['ms'] * 1e-3) / 1e9
            pmc_g = pmc_traffic_per_launch(files=[f'r06_pmc_{args.workload}_FETCH_WRITE.json'])     # this workload's own counter passes, when committed
            roofline = dict(kernel='upfirdn2d_tile_kernel / upfirdn2d_lanes_kernel (the FIR / 2x up-sampling chain of the synthesis network)', bound='hbm', achieved=achieved, peak=HBM_PEAK_GBPS,
                            unit='GB/s', frac=achieved / HBM_PEAK_GBPS, frac_of_measured_copy_peak=achieved / HBM_COPY_GBPS, traffic=pmc_g[0], traffic_source=pmc_g[1] + ' (reads x2, gfx950 correction)', launches=r['launches'],
                            launches_per_forward=r['launches'] / args.steps, algorithmic_bytes_per_forward=r['bytes'] / args.steps, avg_launch_us=1e3 * r['ms'] / r['launches'],
                            share_of_forward_time=r['ms'] / (1e3 * elapsed),
                            note='size-weighted over every upfirdn2d launch of the timed forwards (HIP events recorded by the C ABI on the launch stream)')
            by_size = [dict(algorithmic_MB=b / 1e6, launches=n, avg_us=1e3 * ms / n, GBps=b * n / (ms * 1e-3) / 1e9, share_of_family_time=ms / r['ms']) for b, (n, ms) in sorted(sizes.items(), reverse=True)]
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            # CPU leg: the same generator on the host cores through the plain-PyTorch op path, ONE frame (the clip is 16 of them)
            log('[bench] timing the CPU baseline leg ...')
            threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 64))
            torch.set_num_threads(threads)
            Gc = Generator(**g_kwargs).eval().requires_grad_(False)
            zc, cc, tc = torch.randn([1, 512]), torch.zeros([1, 0]), torch.zeros([1, 1])
            with torch.no_grad():
                tcpu = time.perf_counter()
                Gc(zc, cc, tc)
                first = time.perf_counter() - tcpu
                done, spent = 1, first
                while spent + first < args.cpu_seconds and done < 8:
                    tcpu = time.perf_counter(); Gc(zc, cc, tc); spent += time.perf_counter() - tcpu; done += 1
            cpu = dict(value=done / spent, unit='img/s', cores=threads, kind='port', sample=f'{done} single-frame forward(s) of the same generator at {res}x{res} on the host, fp32, {spent:.1f} s')
        workload = (f'SkyTimelapse-config generator synthesis {res}x{res} (fmaps 1, min_period_len 256), {clips} clip(s) x {frames} frames per GPU and step, eval mode, fp32'
                    if res == 1024 else f'FFS-config generator forward {res}x{res}, {clips} videos x {frames} frames per GPU and step, eval mode, fp32')
        out = dict(metric=f'G synthesis images/sec at {res}^2 ({frames}-frame clips)', value=clips * frames * world * args.steps / elapsed, unit='img/s', n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32 (fp32 tensors + accumulators; 3x3 products = block-scaled 2-way fp16 split on MFMA, vendor-fp32 class, on every layer the hand-written kernels serve (c_out % 32 == 0); vendor fp32 below)',
                   data='synthetic',
                   config=dict(workload=workload, clips_per_gpu=clips, frames_per_clip=frames, parallelism=f'replicas x{world} (no collective)',
                               native_launches_per_forward=(custom_ops.launch_count() - launches0) / (2 * args.steps if not args.no_prof else args.steps)),
                   value_no_prof=clips * frames * world * args.steps / clean, roofline=roofline, upfirdn2d_by_size=by_size, kernels=kernels, cpu_baseline=cpu)
        emit(out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def build_headline_step(make_step, world, rank, want_captured, warmup, dev_sync, between_capture_and_region=None):
    """Build the step the timed region runs, warm it up, and make every rank run the SAME kind of step.  -> (ts, mode)

    want_captured: False (eager step: DDP buckets at world > 1), True (Gmain / Dmain replayed as hipGraphs; at world > 1 one flat eager all-reduce of the captured
    gradient buffers between the two graphs of a phase, TrainStep(ddp_manual)), or 'emulate' (the host-side stand-in of a replayed graph: CPU tests of this path).
    A capture that fails raises on its rank before that rank's first collective of the step; the ranks then agree (MIN over a flag) to rebuild the EAGER step -- the
    line says so in `headline_mode` -- instead of mixing modes.  (A rank that failed AFTER its peers entered a collective would leave them waiting: `--eager` is the
    way around a box where that happens.)
    mode: 'captured' | 'emulated capture' | 'eager' | 'eager (capture failed: ...)'."""
    import torch
    ts, err = None, ''
    if want_captured:
        try:
            ts = make_step(want_captured)
            for i in range(max(warmup, 1)):         # the first step holds the captures: never inside the timed region
                tw = time.perf_counter()
                ts.step()
                dev_sync()
                if rank == 0:
                    log(f'[bench] warm-up iteration {i}: {time.perf_counter() - tw:.2f} s (captures in the first one)')
            if between_capture_and_region is not None:
                between_capture_and_region(ts)
            if warmup < 2:     # the first R1 iteration behind a capture pays 0.5-0.8 s once (profiles/r05_c17b_captured_steps.log): not inside the region
                ts.batch_idx = 0
                ts.step()
                dev_sync()
        except Exception as exc:      # noqa: BLE001  (whatever the runtime raised: the fallback is the documented behaviour)
            err = f'{type(exc).__name__}: {exc}'[:200]
            log(f'[bench] rank {rank}: the captured step failed ({err}); falling back to the eager step on every rank')
            ts = None
        ok = torch.tensor([1.0 if ts is not None else 0.0], device=getattr(ts, 'device', None) or ('cuda' if torch.cuda.is_available() else 'cpu'))
        if world > 1:
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if float(ok.item()) > 0:
            return ts, ('emulated capture' if want_captured == 'emulate' else 'captured')
        del ts
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    ts = make_step(False)
    for i in range(warmup):
        tw = time.perf_counter()
        ts.step()
        dev_sync()
        if rank == 0:
            log(f'[bench] warm-up iteration {i}: {time.perf_counter() - tw:.2f} s (includes MIOpen kernel compilation on a cold cache)')
    return ts, ('eager' if not want_captured else f'eager (capture failed on a rank{": " + err if err else ""})')


def timed_steps(ts, steps, world, dev_sync, device, before_step=None, after_step=None):
    """The contract's timed region: barrier + device sync on both sides of exactly `steps` iterations, MAX over ranks.  -> (seconds, phases_run, per-rank seconds tensor)"""
    import torch

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        dev_sync()
    phases_run = {}
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if before_step is not None:
            before_step(i)
        for name in ts.step():
            phases_run[name] = phases_run.get(name, 0) + 1
        if after_step is not None:
            after_step(i)
    barrier()
    mine = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    t_max = mine.clone()
    if world > 1:
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    return float(t_max.item()), phases_run, mine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-gpu', type=int, default=32, help='videos per GPU (x3 frames each)')
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--cpu-seconds', type=float, default=30.0, help='budget of the CPU baseline leg (1 warm-up + >= 3 timed main iterations + one R1 iteration); 0 disables it')
    ap.add_argument('--workload', choices=['train256', 'g1024', 'g256'], default='train256',
                    help="train256: the G+D training step (BASELINE configs[2], the default); g1024 / g256: generator synthesis of 16-frame clips at "
                         "1024^2 (BASELINE configs[4], SkyTimelapse config: fmaps 1, min_period_len 256) / 3-frame clips at 256^2 (configs[1])")
    ap.add_argument('--clips-gpu', type=int, default=1, help='g1024 / g256: clips per GPU and step (configs[4]: 8 clips over 8 GPUs)')
    ap.add_argument('--no-prof', action='store_true', help='skip the per-launch HIP-event accounting')
    ap.add_argument('--clean-steps', type=int, default=-1, help='steps of the un-instrumented repeat of the timed window (value_no_prof); -1 = --steps, 0 disables it')
    ap.add_argument('--ada-steps', type=int, default=None, help="steps of the aug=ada companion measurement (the reference's default augmentation, bgc pipeline); 0 disables it")
    ap.add_argument('--split3-steps', type=int, default=None, help='steps of the bf16-split companion (terms = 3: the arithmetic of earlier rounds); 0 disables it')
    ap.add_argument('--bf16-steps', type=int, default=None, help='steps of the bf16-products companion measurement (fp32 tensors, one bf16 MFMA per product); 0 disables it')
    ap.add_argument('--strict-steps', type=int, default=None, help='steps of the strict-fp32 companion measurement (all convolutions on the vendor fp32 path); 0 disables it')
    ap.add_argument('--lowp-steps', type=int, default=None, help="steps of the mixed-precision companion (bf16 tensors in the blocks >= 32^2: the reference's num_fp16_res=4 with bf16, BASELINE config 4); 0 disables it")
    ap.add_argument('--pl-steps', type=int, default=None, help='steps of the path-length-regularisation companion (F=1, pl_weight=2); 0 disables it')
    ap.add_argument('--graph-steps', type=int, default=None, help='steps of the captured-step companion (Gmain / Dmain replayed as hipGraphs, single GPU); 0 disables it')
    ap.add_argument('--aug', choices=['noaug', 'ada'], default='noaug', help="discriminator augmentation: the reference's default is ada (bgc pipeline, adaptive p)")
    ap.add_argument('--graphs', action='store_true', help='replay Gmain / Dmain as hipGraphs WITHOUT any per-launch events (no roofline objects)')
    ap.add_argument('--eager', action='store_true', help='single GPU: time the eager step instead of the captured one (the default headline of a one-GPU run is the captured step, with per-launch event nodes inside the graphs)')
    ap.add_argument('--lowp', choices=['none', 'fp16', 'bf16'], default='none',
                    help='mixed precision in the 4 highest resolutions (reference: fp16; BASELINE config 4: bf16). Default: full fp32')
    args = ap.parse_args()
    # Companion measurements (same models, other arithmetic / augmentation / regulariser): part of the default single-GPU line; a multi-GPU run measures the
    # scaling of the headline step and leaves them out unless they are asked for (each builds and warms up further models on every rank).
    multi = args.gpus > 1
    # 10 steps each from iteration 0: one R1 iteration in ten, the share the 20-step headline window has (an R1 iteration costs ~2x a plain one: with 8 steps a
    # companion would carry 1/8 and read 1.5 % low against the headline); the PL companion a whole period of its schedule (Greg every 4th, Dreg every 16th)
    for name, dflt in (('ada_steps', 10), ('bf16_steps', 10), ('strict_steps', 10), ('lowp_steps', 10), ('pl_steps', 16), ('split3_steps', 10), ('graph_steps', 10)):
        if getattr(args, name) is None:
            setattr(args, name, 0 if multi else dflt)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: re-launch as N ranks (one per GPU) under torch.distributed.run; rank 0 prints the line
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log('[bench] spawning', ' '.join(cmd))
        os.execvp(sys.executable, cmd)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from stylegan_v_amd.torch_utils import custom_ops
    from stylegan_v_amd.torch_utils.ops import conv2d_gradfix
    from stylegan_v_amd.training import config as cfgs
    from stylegan_v_amd.training.train_step import TrainStep
    custom_ops.verbosity = 'none' if rank else 'brief'
    # fail loudly here if the HIP library is missing; rank 0 goes first so that a stale in-tree .so is rebuilt once
    if rank == 0:
        custom_ops.get_native()
    if world > 1:
        torch.distributed.barrier()
    custom_ops.get_native()

    # The reference sets cudnn.benchmark=True (training_loop.py:140).  MIOpen in this image ships no gfx950 find-db, so
    # "find" would time every solver (incl. the naive one) on full-size tensors for many minutes: use immediate mode.
    import stylegan_v_amd
    stylegan_v_amd.configure_miopen(immediate=os.environ.get('SGV_MIOPEN_FIND', '0') != '1')
    if args.workload != 'train256':
        return synthesis_workload(args, world, rank, device)
    global PMC_NOT_THIS_WORKLOAD
    if (args.batch_gpu, args.frames, args.res, args.lowp, args.aug) != (32, 3, 256, 'none', 'noaug'):
        PMC_NOT_THIS_WORKLOAD = 'the committed counter passes are of the default workload (32 videos x 3 frames at 256^2, fp32, noaug): bytes per launch of another workload are not this run\'s'
    global_batch = args.batch_gpu * world
    lowp = {'none': None, 'fp16': torch.float16, 'bf16': torch.bfloat16}[args.lowp]
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=args.res, batch_size=global_batch, num_gpus=world, fp32=(lowp is None),
                                                      num_frames_per_video=args.frames, lowp_dtype=lowp)
    # One GPU: the headline step is the CAPTURED one: Gmain / Dmain replayed as hipGraphs, the reg phases eager at their schedule --
    # the same work, without the dispatch latency between ~2,000 dependent launches per iteration (and at the clocks the chip reaches when those gaps are gone).  The
    # roofline object needs the dominant kernel's per-launch duration from INSIDE the timed region, and a replay launches nothing from the host: the stride-1
    # convolution writes its own timestamp pair when it is recorded during a capture (sgv_launch_scope::kernel_stamps, conv_ws_params::stamp: workgroup 0 stores the
    # 100-MHz device clock as it starts, the consumer waves atomicMax it as they leave) -- no node is added to the graph, no gap opened, so EVERY replay carries the
    # timing and the durations of the last iteration are read after the region.  (Round 5 first bracketed the launches with one-thread timestamp kernels in a second graph set that only the last iteration replayed: 2-4 ms
    # of their own, and an odd iteration that ran in another clock state than its neighbours in three of four records.)  The upfirdn2d family's in-region
    # sample, the per-variant tables and `value_eager` come from the eager step behind the region.  Several GPUs (DDP's reducer cannot be captured) or --eager: the
    # eager step with HIP events, as in rounds 1-4.
    captured_headline = not args.eager and not args.graphs and not args.no_prof      # (every N: the scaling curve compares like with like; --eager: the eager DDP step)
    if captured_headline:
        import contextlib
        from stylegan_v_amd.training import train_step as _tsmod
        custom_ops.prof_families(CAPTURED_FAMILIES)
        custom_ops.prof_enable(1 << 15)      # allocates the pools and starts a new record list ...
        custom_ops.prof_disable()            # ... which only the captures append to

        @contextlib.contextmanager
        def _capture_events():
            custom_ops.prof_resume()
            try:
                yield
            finally:
                custom_ops.prof_disable()
        _tsmod._HipGraph.capture_hook = _capture_events
    def make_step(graphs):
        return TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=args.batch_gpu, world_size=world, rank=rank, use_graphs=graphs, augment=args.aug)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def _captures_done(_ts):
        _tsmod._HipGraph.capture_hook = None
    ts, headline_mode = build_headline_step(make_step, world, rank, bool(args.graphs or captured_headline), args.warmup, torch.cuda.synchronize,
                                            between_capture_and_region=_captures_done if captured_headline else None)
    if captured_headline and headline_mode != 'captured':      # the agreed fallback: the eager step with HIP events, as --eager
        _tsmod._HipGraph.capture_hook = None
        custom_ops.prof_disable()
        custom_ops.prof_families(None)
        captured_headline = False
    headline_sync = ('one flat all-reduce per phase between the two hipGraphs' if ts.ddp_manual else 'DDP buckets (eager)') if world > 1 else None
    # Start the timed window on an iteration that runs the regularisation phases, whatever the warm-up was.
    ts.batch_idx = 0
    launches0 = custom_ops.launch_count()
    if args.graphs:
        args.no_prof = True   # a replayed graph launches nothing from the host: there are no per-launch events to record
    if os.environ.get('SGV_TORCH_PROFILE'):
        # developer aid: aten-level table of two steps with input shapes (who issues the element-wise kernels), to the given file
        import torch.profiler as tprof
        ts.batch_idx = int(os.environ.get('SGV_TORCH_PROFILE_FROM', '0'))      # 0: starts on an iteration with the regularisation phases; 1: two plain iterations
        with tprof.profile(activities=[tprof.ProfilerActivity.CPU, tprof.ProfilerActivity.CUDA], record_shapes=True) as prof_t:
            ts.step(); ts.step()
            torch.cuda.synchronize()
        if rank == 0:
            with open(os.environ['SGV_TORCH_PROFILE'], 'w') as fh:
                fh.write(prof_t.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=120, max_name_column_width=60, max_shapes_column_width=90))
                # every op / kernel by call count (two steps): who issues the small launches
                avgs = sorted(prof_t.key_averages(), key=lambda e: -e.count)
                fh.write('\n\n%-90s %8s %12s %12s\n' % ('name (all ops and kernels of 2 steps, by call count)', 'calls', 'cpu ms', 'device ms'))
                for e in avgs[:300]:
                    fh.write('%-90s %8d %12.3f %12.3f\n' % (e.key[:90], e.count, e.cpu_time_total / 1e3, getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0.0)) / 1e3))
                # the element-wise / copy / reduction ops of torch itself, by input shape: which tensors still take a separate pass
                fh.write('\n\n%-24s %-110s %6s %10s\n' % ('torch op (2 steps)', 'input shapes', 'calls', 'device ms'))
                rows = [e for e in prof_t.key_averages(group_by_input_shape=True)
                        if e.key in ('aten::add_', 'aten::add', 'aten::mul', 'aten::mul_', 'aten::copy_', 'aten::clone', 'aten::sum', 'aten::cat', 'aten::fill_', 'aten::div', 'aten::sub', 'aten::pow', 'aten::mean')]
                for e in sorted(rows, key=lambda e: -getattr(e, 'device_time_total', 0.0))[:60]:
                    fh.write('%-24s %-110s %6d %10.3f\n' % (e.key, str(e.input_shapes)[:110], e.count, getattr(e, 'device_time_total', 0.0) / 1e3))
        ts.batch_idx = 0
    # Per-launch HIP events (two event packets around a native launch) cost 2.5-4 % of the step when every one of the ~550 launches per iteration carries them on
    # every step (profiles/r02_bench_prof_overhead.log), 1.2-1.4 % on half of the steps (round 4: 580.4 vs 588.5 img/s).  Inside the timed region they are therefore
    # recorded (i) on the SECOND HALF of the timed steps only -- with the default 20 steps that is steps 10-19, which hold one R1 iteration in ten like the whole
    # window (steps 0 and 16), so the kernel mix of the sample is the mix of the window -- and (ii) for the two kernel families the roofline objects are about
    # (TIMED_FAMILIES: the dominant stride-1 3x3 kernel and the upfirdn2d family, ~120 launches per iteration).  The full per-family / per-variant tables come from
    # a fully instrumented pass of their own behind the timed region.  `value` is the wall time of all K steps.
    prof_from = args.steps // 2
    power = PowerSampler(device.index or 0) if rank == 0 else None
    if power is not None:
        power.__enter__()
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # one event per iteration boundary: the per-iteration device times in the side file

    def _before(i_step):
        if i_step == 0:
            step_marks[0].record()
        if not args.no_prof and not captured_headline and i_step == prof_from:
            custom_ops.prof_families(TIMED_FAMILIES)      # inside the timed region: the dominant kernel and the FIR family only (all 554 launches per iteration: 1.4 % of the step)
            custom_ops.prof_enable(1 << 17)
    elapsed, phases_run, t_mine = timed_steps(ts, args.steps, world, torch.cuda.synchronize, device, before_step=_before, after_step=lambda i_step: step_marks[i_step + 1].record())
    if power is not None:
        power.__exit__()
    step_ms = [round(step_marks[i].elapsed_time(step_marks[i + 1]), 3) for i in range(args.steps)]
    log(f'[bench] per-iteration device time of the timed region (ms): {step_ms}')
    launches = custom_ops.launch_count() - launches0
    def collect_tables():
        """Stop recording and fold the per-launch records: per family, per kernel variant, and the upfirdn2d launches grouped by size."""
        custom_ops.prof_disable()
        records = custom_ops.prof_collect_records(1 << 17, with_variant=True)
        custom_ops.prof_families(None)
        table = {name: dict(launches=0, ms=0.0, bytes=0.0, flops=0.0) for name in custom_ops.SGV_K_NAMES}
        sizes, variants_ = {}, {}
        for fam, ms, nbytes, nflops, variant in records:
            e = table[fam]
            e['launches'] += 1; e['ms'] += ms; e['bytes'] += nbytes; e['flops'] += nflops
            v = variants_.setdefault(variant or fam, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
            v['launches'] += 1; v['ms'] += ms; v['bytes'] += nbytes; v['flops'] += nflops
            if fam == 'upfirdn2d_lanes':
                g = sizes.setdefault(int(nbytes), [0, 0.0])
                g[0] += 1; g[1] += ms
        for key in ('launches', 'ms', 'bytes', 'flops'):      # 'conv3x3' is the whole family incl. its largest member 'conv3x3_s1' (as custom_ops.prof_collect)
            table['conv3x3'][key] += table['conv3x3_s1'][key]
        # the upfirdn2d launches of the sample grouped by their algorithmic byte count (= by layer size and fused mode): where the size-weighted mean comes from
        by_size = [dict(algorithmic_MB=b / 1e6, launches=n, avg_us=1e3 * ms / n, GBps=b * n / (ms * 1e-3) / 1e9, share_of_family_time=ms / max(table['upfirdn2d_lanes']['ms'], 1e-9))
                   for b, (n, ms) in sorted(sizes.items(), reverse=True)]
        return table, variants_, by_size

    prof = None
    prof_timed = None          # the families recorded INSIDE the timed region (TIMED_FAMILIES): what `roofline` / `roofline_upfirdn2d` are computed from
    ufd_by_size = None
    by_variant = None
    if not args.no_prof:
        prof_timed, _, ufd_by_size = collect_tables()      # (captured headline: the event nodes' recordings of the LAST replayed Gmain / Dmain iteration)
    hip_graphs_headline = bool(ts.use_graphs)
    headline_launches_per_step = launches / args.steps
    if captured_headline:
        # Everything behind the headline -- the eager step as a companion, the instrumented per-variant tables, the arithmetic / augmentation companions that switch
        # dispatch flags at run time (a captured graph has them baked in) -- runs on an EAGER step with its own models; the captured one is released first.
        _tsmod._HipGraph.capture_hook = None
        del ts
        torch.cuda.empty_cache()
        ts = make_step(False)      # (world > 1: the eager DDP step -- `value_eager` at every N)
        ts.batch_idx = 1
        ts.step(); ts.step()
        torch.cuda.synchronize()

    t_max = t_mine      # this rank's own time (`elapsed` is already the MAX over ranks)
    multi_gpu = None
    if world > 1:
        # what the ranks saw: every rank's own time, the number of ranks RCCL reduced over, and the flat gradient all-reduce of each phase on its own
        per_rank = [torch.zeros_like(t_max) for _ in range(world)]
        torch.distributed.all_gather(per_rank, t_max)
        ones = torch.ones([1], device=device)
        torch.distributed.all_reduce(ones)
        allreduce_ms = {}
        for label, module in (('G', ts.G), ('D', ts.D)):
            flat = torch.zeros([sum(p.numel() for p in module.parameters())], device=device)
            torch.distributed.all_reduce(flat)       # warm-up (communicator set-up for this size)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                torch.distributed.all_reduce(flat)
            torch.cuda.synchronize()
            allreduce_ms[label] = dict(MB=flat.numel() * 4 / 1e6, ms=(time.perf_counter() - t1) / 5 * 1e3)
        multi_gpu = dict(rccl_ranks_seen=int(ones.item()), ms_per_step_by_rank=[1e3 * float(v.item()) / args.steps for v in per_rank], flat_gradient_allreduce=allreduce_ms,
                         gradient_sync=headline_sync, headline_mode=headline_mode)
    frames_total = global_batch * args.frames * args.steps
    value = frames_total / elapsed

    # The same K steps once more WITHOUT any per-launch event recording (`value` carries the recorder's cost for two kernel families on half of its steps):
    # same schedule start, same bracket, MAX over ranks -> `value_no_prof`.
    value_no_prof = None
    if not args.no_prof and args.clean_steps != 0:
        k_clean = args.steps if args.clean_steps < 0 else args.clean_steps
        ts.batch_idx = 0
        barrier()
        t1 = time.perf_counter()
        for _ in range(k_clean):
            ts.step()
        barrier()
        t_c = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(t_c, op=torch.distributed.ReduceOp.MAX)
        value_no_prof = dict(value=global_batch * args.frames * k_clean / float(t_c.item()), ms_per_step=1e3 * float(t_c.item()) / k_clean, steps=k_clean)
    elif args.no_prof:
        value_no_prof = dict(value=value, ms_per_step=1e3 * elapsed / args.steps, steps=args.steps)
    value_eager = None
    if captured_headline:      # the K steps just timed WERE the eager step (no events): the headline's eager twin, not a "no_prof" repeat of it
        value_eager, value_no_prof = value_no_prof, None
    # The full per-family / per-variant tables (`kernels`, `kernels_by_variant`, `roofline_conv_family`): every launch bracketed, in a pass of its own over the
    # same schedule positions as the recorded half of the timed window (iterations prof_from .. steps - 1), outside every timed region.
    if not args.no_prof:
        ts.batch_idx = prof_from
        barrier()
        custom_ops.prof_enable(1 << 17)
        for _ in range(args.steps - prof_from):
            ts.step()
        barrier()
        prof, by_variant, ufd_eager = collect_tables()
        if captured_headline:
            ufd_by_size = ufd_eager      # (no upfirdn2d launch is timed inside the replayed graphs)
        for fam in (() if captured_headline else TIMED_FAMILIES):      # the two families of the timed region keep their in-region figures in the family table
            if prof_timed[fam]['launches'] and prof_timed[fam]['ms'] > 0:   # (captured headline: the tables stay the eager step's, the roofline objects take the in-graph sample)
                for key in ('launches', 'ms', 'bytes', 'flops'):
                    if fam == 'conv3x3_s1':
                        prof['conv3x3'][key] += prof_timed[fam][key] - prof[fam][key]
                    prof[fam][key] = prof_timed[fam][key]

    # Strict-fp32 companion (the reference's fp32 mode is allow_tf32=False, training_loop.py:129,141-142): the same step with every
    # 3x3 convolution on the vendor library's fp32 kernels instead of the split-bf16 matrix-pipe kernels.  Same schedule start
    # (iteration 0 runs all four phases), same barrier / synchronize bracket, MAX over ranks.
    strict = None
    default_terms = (conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms)
    if args.strict_steps > 0 and lowp is None and default_terms != (0, 0):
        conv2d_gradfix.native_conv_terms = conv2d_gradfix.native_wrw_terms = 0
        try:
            tw = time.perf_counter()
            ts.batch_idx = 1
            ts.step()                      # MIOpen kernel selection / compilation for the shapes that were on the native path
            ts.batch_idx = 0
            ts.step()                      # ... and for the R1 shapes
            torch.cuda.synchronize()
            if rank == 0:
                log(f'[bench] strict-fp32 companion: warm-up {time.perf_counter() - tw:.1f} s')
            ts.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            strict_phases = {}
            for _ in range(args.strict_steps):
                for name in ts.step():
                    strict_phases[name] = strict_phases.get(name, 0) + 1
            barrier()
            t_s = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t_s, op=torch.distributed.ReduceOp.MAX)
            strict = dict(value=global_batch * args.frames * args.strict_steps / float(t_s.item()), ms_per_step=1e3 * float(t_s.item()) / args.strict_steps,
                          steps=args.strict_steps, phases_run=strict_phases,
                          what='same step, SGV_CONV_TERMS=0 SGV_WRW_TERMS=0: every 3x3 convolution and weight gradient on the vendor library (MIOpen fp32)')
        finally:
            conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms = default_terms

    # bf16-split companion: the arithmetic of rounds 1-3's headline (2-way bf16 split, 16-bit operands, 4.4e-6 -- inside north_star's 1e-3, not fp32-grade)
    # on the same models: what the fp32-grade default costs against it (the bound passes + the fp16 operands' lower clock).
    split3 = None
    if args.split3_steps > 0 and lowp is None and default_terms == (4, 4):
        conv2d_gradfix.native_conv_terms = conv2d_gradfix.native_wrw_terms = 3
        try:
            ts.batch_idx = 0
            ts.step(); ts.step()
            ts.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.split3_steps):
                ts.step()
            barrier()
            t_s = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t_s, op=torch.distributed.ReduceOp.MAX)
            split3 = dict(value=global_batch * args.frames * args.split3_steps / float(t_s.item()), ms_per_step=1e3 * float(t_s.item()) / args.split3_steps, steps=args.split3_steps,
                          what='same step, SGV_CONV_TERMS=3 SGV_WRW_TERMS=3: 2-way bf16 split (the headline arithmetic of rounds 1-3; 4.4e-6 rel. error per convolution)')
        finally:
            conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms = default_terms

    # bf16-products companion (BASELINE config 4 says "bf16 compute"): the same step, fp32 tensors and fp32 accumulation, but ONE bf16 MFMA per
    # product in the 3x3 family (terms = 1: operands rounded to bf16, rel-L2 2e-3 per convolution) instead of the three of the split.
    bf16c = None
    if args.bf16_steps > 0 and lowp is None and default_terms in ((3, 3), (4, 4)):
        conv2d_gradfix.native_conv_terms = conv2d_gradfix.native_wrw_terms = 1
        try:
            ts.batch_idx = 0
            ts.step(); ts.step()
            ts.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.bf16_steps):
                ts.step()
            barrier()
            t_s = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t_s, op=torch.distributed.ReduceOp.MAX)
            bf16c = dict(value=global_batch * args.frames * args.bf16_steps / float(t_s.item()), ms_per_step=1e3 * float(t_s.item()) / args.bf16_steps, steps=args.bf16_steps,
                         what='same step, SGV_CONV_TERMS=1 SGV_WRW_TERMS=1: fp32 tensors, bf16-rounded operands (one MFMA per product), fp32 accumulate; '
                              'outside the 1e-3 fp32 parity bar (2e-3 rel-L2 per 3x3 convolution), reported as the bf16-compute reading of BASELINE config 4')
        finally:
            conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms = default_terms

    # aug=ada companion: the reference's default discriminator augmentation (bgc pipeline: reflect-pad -> 2x up -> affine resample -> 2x down +
    # colour matrix on every D input, adaptive p) on the same models; same bracket, schedule restarted at iteration 0.
    ada = None
    if args.ada_steps > 0 and args.aug == 'noaug' and not args.graphs:
        ts.set_augment('ada')
        try:
            ts.batch_idx = 0
            ts.step(); ts.step()
            ts.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.ada_steps):
                ts.step()
            barrier()
            t_a = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t_a, op=torch.distributed.ReduceOp.MAX)
            ada = dict(value=global_batch * args.frames * args.ada_steps / float(t_a.item()), ms_per_step=1e3 * float(t_a.item()) / args.ada_steps, steps=args.ada_steps,
                       p_final=float(ts.augment_pipe.p), what='same step with aug=ada (augpipe bgc, one transform per video, ADA target 0.6 / interval 4 / 500 kimg)')
        finally:
            ts.set_augment('noaug')
        # the control: the same step object, aug=noaug, the same schedule window, straight behind the aug=ada window -- the chip's clock state drifts by several per
        # cent over a bench run, so the cost of the augmentation is `value / noaug_control_value` of THIS pair, not `value_aug_ada / value`
        ts.batch_idx = 0
        ts.step(); ts.step()
        ts.batch_idx = 0
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.ada_steps):
            ts.step()
        barrier()
        t_c = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(t_c, op=torch.distributed.ReduceOp.MAX)
        ada['noaug_control_value'] = global_batch * args.frames * args.ada_steps / float(t_c.item())
        ada['ratio_to_control'] = ada['value'] / ada['noaug_control_value']

    # ... and the same step the way the headline runs it: Gmain / Dmain replayed as hipGraphs (static worst-case reflect margin: nothing is read back to the host; the
    # geometric block is one kernel per direction, so the wider virtual padding costs nothing).  Its own models; same bracket and schedule.  `value_aug_ada_captured / value`
    # is the like-for-like cost of the augmentation when `headline_mode` is "captured".
    ada_cap = None
    if ada is not None and headline_mode == 'captured' and world == 1:
        ts_a = TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=args.batch_gpu, world_size=world, rank=rank, use_graphs=True, augment='ada')
        try:
            ts_a.batch_idx = 1
            for _ in range(4):
                ts_a.step()            # eager warm-up on a side stream + the captures + their first replays
            torch.cuda.synchronize()
            ts_a.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.ada_steps):
                ts_a.step()
            barrier()
            t_a = time.perf_counter() - t1
            ada_cap = dict(value=global_batch * args.frames * args.ada_steps / t_a, ms_per_step=1e3 * t_a / args.ada_steps, steps=args.ada_steps, p_final=float(ts_a.augment_pipe.p),
                           graphs=sorted(ts_a._graphs), what='aug=ada with Gmain / Dmain replayed as hipGraphs (the mode of the headline); static reflect margin')
        except Exception as err:      # (a companion: report, do not fail the line)
            ada_cap = dict(value=None, error=str(err).splitlines()[0][:200])
        finally:
            del ts_a
            torch.cuda.empty_cache()

    # Path-length companion ("time PL in a separate F=1 run"): the reference's PL term only runs with one frame per video
    # (loss.py:117), so this is config 3 with num_frames_per_video = 1 and pl_weight = 2 (the StyleGAN2 default, train.py:189): Gmain, Greg (PL:
    # second order through G) every 4th, Dmain, Dreg every 16th; its own models (the frame count changes D's input layer), same bracket.
    plc = None
    if args.pl_steps > 0 and lowp is None and not args.graphs:
        g_kw1, d_kw1, train_cfg1 = cfgs.model_kwargs(resolution=args.res, batch_size=global_batch, num_gpus=world, fp32=True, num_frames_per_video=1)
        train_cfg1.pl_weight = 2.0
        ts1 = TrainStep(g_kw1, d_kw1, train_cfg1, device=device, batch_gpu=args.batch_gpu, world_size=world, rank=rank, augment='noaug')
        try:
            tw = time.perf_counter()
            ts1.step(); ts1.step()
            torch.cuda.synchronize()
            if rank == 0:
                log(f'[bench] PL / F=1 companion: warm-up {time.perf_counter() - tw:.1f} s')
            ts1.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            pl_phases = {}
            for _ in range(args.pl_steps):
                for name in ts1.step():
                    pl_phases[name] = pl_phases.get(name, 0) + 1
            barrier()
            t_p = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t_p, op=torch.distributed.ReduceOp.MAX)
            plc = dict(value=global_batch * 1 * args.pl_steps / float(t_p.item()), ms_per_step=1e3 * float(t_p.item()) / args.pl_steps, steps=args.pl_steps, phases_run=pl_phases,
                       pl_penalty=float(ts1.last_losses.get('G/reg', float('nan'))),
                       what='config 3 with num_frames_per_video=1 and pl_weight=2: Gmain + Greg (path-length regularisation, double backward through G, batch shrink 2) '
                            'every 4th + Dmain + Dreg (R1) every 16th; frames/s = videos/s here')
        finally:
            del ts1
            torch.cuda.empty_cache()

    # Mixed-precision companion (BASELINE config 4, "bf16 compute"): config 3 with the reference's own mixed-precision switch (train.py:173-174:
    # num_fp16_res = 4, conv_clamp = 256 -> the blocks at 32^2 .. 256^2 hold 16-bit activations) and bf16 as the 16-bit format: 16-bit tensor I/O on
    # the hand-written kernels, fp32 master weights, fp32 accumulate, fp32 weight gradients.  Its own models; same bracket and schedule.
    lowpc = None
    if args.lowp_steps > 0 and lowp is None and not args.graphs and args.workload == 'train256':
        g_kw2, d_kw2, train_cfg2 = cfgs.model_kwargs(resolution=args.res, batch_size=global_batch, num_gpus=world, fp32=False, num_frames_per_video=args.frames,
                                                     lowp_dtype=torch.bfloat16)
        ts2 = TrainStep(g_kw2, d_kw2, train_cfg2, device=device, batch_gpu=args.batch_gpu, world_size=world, rank=rank, augment=args.aug)
        try:
            tw = time.perf_counter()
            ts2.step(); ts2.batch_idx = 0; ts2.step()
            torch.cuda.synchronize()
            if rank == 0:
                log(f'[bench] mixed-precision (bf16) companion: warm-up {time.perf_counter() - tw:.1f} s')
            ts2.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            lp_phases = {}
            for _ in range(args.lowp_steps):
                for name in ts2.step():
                    lp_phases[name] = lp_phases.get(name, 0) + 1
            barrier()
            t_l = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(t_l, op=torch.distributed.ReduceOp.MAX)
            lowpc = dict(value=global_batch * args.frames * args.lowp_steps / float(t_l.item()), ms_per_step=1e3 * float(t_l.item()) / args.lowp_steps, steps=args.lowp_steps,
                         dtype='bf16 (blocks >= 32^2; f32 accumulate, f32 master weights, f32 weight gradients)', phases_run=lp_phases,
                         what='config 3 with num_fp16_res=4, conv_clamp=256 (train.py:173-174) and bf16 as the 16-bit format: 16-bit tensor I/O on the hand-written '
                              'convolution / FIR / bias_act kernels; parity at the stated 16-bit tolerance 1e-2 (tests/test_conv_lowp_gpu.py)')
        finally:
            del ts2
            torch.cuda.empty_cache()

    # Captured-step companion: the same models' configuration with the Gmain / Dmain phases replayed as hipGraphs (TrainStep(use_graphs=True): two graphs
    # per phase, the reg phases eager).  The eager step issues ~2,000 launches per iteration from Python and leaves the device idle for ~9 of its 158 ms
    # (profiles/r05_bench_step_kernel_stats_final.csv: 144.7 ms of kernels per iteration); a replay has no such gaps.  Its own models (captures bind their tensors);
    # same bracket and schedule.  The headline stays the eager step: per-launch HIP events -- the `roofline` objects -- cannot be recorded inside a replay.
    graphc = None
    if args.graph_steps > 0 and lowp is None and not args.graphs and world == 1 and not captured_headline:
        ts3 = TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=args.batch_gpu, world_size=world, rank=rank, use_graphs=True, augment=args.aug)
        try:
            tw = time.perf_counter()
            ts3.batch_idx = 1
            ts3.step(); ts3.step()            # eager warm-up on a side stream + the captures + their first replays
            ts3.batch_idx = 0
            ts3.step()                        # ... and the eager reg phases
            torch.cuda.synchronize()
            if rank == 0:
                log(f'[bench] captured-step companion: warm-up + capture {time.perf_counter() - tw:.1f} s')
            ts3.batch_idx = 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.graph_steps):
                ts3.step()
            barrier()
            t_g = time.perf_counter() - t1
            graphc = dict(value=global_batch * args.frames * args.graph_steps / t_g, ms_per_step=1e3 * t_g / args.graph_steps, steps=args.graph_steps,
                          what='same step with Gmain / Dmain replayed as hipGraphs (two graphs per phase: gradients | sanitise + Adam; Greg / Dreg eager)')
        finally:
            del ts3
            torch.cuda.empty_cache()

    eager_note = (f'the eager step behind the timed region (HIP events around every launch of a pass of its own)' if captured_headline
                  else f'all launches inside the second half of the timed steps (steps {args.steps // 2}..{args.steps - 1})')
    sample_note = (f'the launches of the LAST iteration of the timed steps (step {args.steps - 1}: Gmain + Dmain replayed from hipGraphs; the kernel writes its own start / end in the 100-MHz device clock, '
                   f'conv_ws_params::stamp -- no node added to the graphs)' if captured_headline else eager_note)
    F32_LABEL = 'f32' if default_terms == (0, 0) else \
        ('f32 (fp32 tensors + accumulators; products = block-scaled 2-way fp16 split on MFMA: 22-bit operands, 2.7e-7 vs fp64 = vendor-fp32 class)' if default_terms == (4, 4) else
         'fp32 tensors + accumulators; products = 2-way bf16 split (16-bit operands, 4.4e-6 vs fp64: NOT fp32-grade)')
    if rank == 0:
        kernels = {}
        roofline = None
        roofline_ufd = None
        roofline_family = None
        if prof is not None:
            for name, e in prof.items():
                if e['launches'] == 0:
                    continue
                k = dict(launches=e['launches'], ms_total=e['ms'], avg_us=1e3 * e['ms'] / e['launches'])
                if e['bytes'] > 0:
                    k['GBps'] = e['bytes'] / (e['ms'] * 1e-3) / 1e9
                    k['bytes_per_launch'] = e['bytes'] / e['launches']
                if e['flops'] > 0:
                    k['TFLOPs'] = e['flops'] / (e['ms'] * 1e-3) / 1e12
                kernels[name] = k
            live = captured_headline and prof_timed is not None and prof_timed['conv3x3_s1']['ms'] > 0
            in_region = prof_timed if live else prof     # what the `roofline` object is computed from
            if not live:
                sample_note = eager_note
            r = prof['upfirdn2d_lanes'] if captured_headline else in_region['upfirdn2d_lanes']      # (captured headline: no timestamps inside this family's kernels -> the eager pass)
            if r['launches']:
                achieved = r['bytes'] / (r['ms'] * 1e-3) / 1e9
                roofline_ufd = dict(kernel='upfirdn2d_tile_kernel / upfirdn2d_down2_tile_kernel / upfirdn2d_up2_tile_kernel (+ the lane-exchange forms)', bound='hbm', achieved=achieved, peak=HBM_PEAK_GBPS, unit='GB/s', frac=achieved / HBM_PEAK_GBPS,
                                    frac_of_measured_copy_peak=achieved / HBM_COPY_GBPS, traffic=pmc_traffic_per_launch()[0], launches=r['launches'],
                                    traffic_source=pmc_traffic_per_launch()[1] + ' (reads x2, gfx950 correction)',
                                    avg_launch_us=1e3 * r['ms'] / r['launches'], algorithmic_bytes_per_launch=r['bytes'] / r['launches'],
                                    note=(eager_note if captured_headline else sample_note) + ' (every layer size, fwd+bwd), size-weighted')
            # The contract's `roofline` is the step's dominant hand-written kernel (largest summed HIP-event time): the stride-1 producer / consumer
            # 3x3 kernel, accounted on its own (SGV_K_CONV3X3_S1); `roofline_conv_family` keeps the figure of the whole 3x3 family that earlier rounds'
            # lines reported under `roofline` (stride 1 + stride 2 + transposed + 16^2 / 8^2 + edge-strip members, flop-weighted).
            def mfma_roofline(e, terms, kernel, traffic):
                achieved = e['flops'] / (e['ms'] * 1e-3) / 1e12
                terms = MFMAS_PER_PRODUCT.get(terms, 3)        # MFMAs per product: 3 for both 2-way splits (terms = 3: bf16, terms = 4: block-scaled fp16)
                peak = MFMA_BF16_PEAK_TFLOPS / terms
                return dict(kernel=kernel, bound='mfma', achieved=achieved, peak=peak, unit='TFLOP/s', frac=achieved / peak, traffic=traffic[0],
                            traffic_source=traffic[1] + ' (L2-miss bytes: Infinity-Cache hits included)',
                            algorithmic_bytes_per_launch=e['bytes'] / e['launches'], launches=e['launches'],
                            avg_launch_us=1e3 * e['ms'] / e['launches'], algorithmic_flops_per_launch=e['flops'] / e['launches'],
                            executed_bf16_TFLOPs=achieved * terms, bf16_dense_peak_TFLOPs=MFMA_BF16_PEAK_TFLOPS, fp32_mfma_peak_TFLOPs=157.3,
                            note=f'algorithmic flops = 2*N*H*W*Cin*Cout*9 (fp32-equivalent); the kernel issues {terms} 16-bit MFMAs per product (hi/lo split, fp32 accumulate), '
                                 f'so its ceiling is the 16-bit dense peak / {terms}; ' + sample_note + ', flop-weighted')
            roofline_family = None
            if prof['conv3x3']['launches']:
                roofline_family = mfma_roofline(prof['conv3x3'], conv2d_gradfix.native_conv_terms,
                                                'conv3x3_ws_kernel / conv3x3_s2_pairs_kernel / convT3x3_s2_ws_kernel (+ 16x16 / 8x8 and edge-strip members)', pmc_traffic_conv_family())
            singles = {n: e for n, e in prof.items() if e['launches'] and n != 'conv3x3'}     # 'conv3x3' is a sum that contains 'conv3x3_s1'
            dom = max(singles, key=lambda n: singles[n]['ms'], default=None)
            if dom == 'conv3x3_s1':
                roofline = mfma_roofline(in_region[dom], conv2d_gradfix.native_conv_terms, 'conv3x3_ws_kernel (stride 1, forward and data gradient, incl. the variants with the layer tail in the store)',
                                         pmc_traffic(('conv3x3_ws_kernel',)))
            elif dom == 'conv_wrw':
                roofline = mfma_roofline(prof[dom], conv2d_gradfix.native_wrw_terms, 'wrw3x3_ws_kernel / wrw3x3_s2_ws_kernel', pmc_traffic(('wrw3x3',)))
            elif dom == 'gemm':
                e = prof[dom]
                achieved = e['flops'] / (e['ms'] * 1e-3) / 1e12
                roofline = dict(kernel='gemm_f32_kernel', bound='mfma', achieved=achieved, peak=157.3, unit='TFLOP/s', frac=achieved / 157.3, traffic=None, launches=e['launches'])
            elif dom is not None and dom != 'upfirdn2d_lanes':
                e = prof[dom]
                achieved = e['bytes'] / (e['ms'] * 1e-3) / 1e9
                roofline = dict(kernel=dom, bound='hbm', achieved=achieved, peak=HBM_PEAK_GBPS, unit='GB/s', frac=achieved / HBM_PEAK_GBPS, traffic=None, launches=e['launches'])
            else:
                roofline = roofline_ufd
        # the same sample by kernel variant (which member of a family a call took; a call = the kernel + its auxiliary launches): TFLOP/s against the
        # split-bf16 ceiling for the matrix-pipe members, GB/s against HBM for the streams
        variants = None
        if by_variant:
            total_ms = sum(v['ms'] for v in by_variant.values())
            variants = {}
            for name, v in sorted(by_variant.items(), key=lambda kv: -kv[1]['ms']):
                row = dict(launches=v['launches'], ms_per_step=v['ms'] / max(args.steps - prof_from, 1), avg_us=1e3 * v['ms'] / v['launches'], share_of_native_time=v['ms'] / total_ms)
                if v['flops'] > 0:
                    row['TFLOPs'] = v['flops'] / (v['ms'] * 1e-3) / 1e12
                    if name.startswith(('conv_', 'convT_', 'wrw_')):
                        one_term = name.endswith('_lowp') or conv2d_gradfix.native_conv_terms == 1
                        row['frac_of_ceiling'] = row['TFLOPs'] / (MFMA_BF16_PEAK_TFLOPS if one_term else MFMA_BF16_PEAK_TFLOPS / 3)
                if v['bytes'] > 0:
                    row['GBps'] = v['bytes'] / (v['ms'] * 1e-3) / 1e9
                variants[name] = row
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            log('[bench] timing the CPU baseline leg ...')
            cpu = cpu_baseline(args.res, args.frames, args.cpu_seconds)
        out = dict(metric='G+D train-step images/sec at 256^2', value=value, unit='img/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype={'none': F32_LABEL, 'fp16': 'f16 (blocks >= 32^2; f32 accumulate, f32 master weights)', 'bf16': 'bf16 (blocks >= 32^2; f32 accumulate, f32 master weights)'}[args.lowp], data='synthetic',
                   config=dict(workload=f'FFS {args.res}x{args.res} full G+D train step (Gmain+Greg+Dmain+Dreg/R1), cfg=auto fmaps 0.5, ' + ('fp32' if lowp is None else args.lowp + ' mixed precision') + ', aug=' + args.aug
                               + (', Gmain / Dmain replayed as hipGraphs' if hip_graphs_headline else ''),
                               videos_per_gpu=args.batch_gpu, frames_per_video=args.frames, frames_per_gpu=args.batch_gpu * args.frames,
                               global_batch_videos=global_batch, parallelism=f'dp{world}', phases_run=phases_run,
                               pl_reg='off (reference config pl_weight=0; Greg phase is a no-op)', r1_gamma=train_cfg.r1_gamma,
                               native_launches_per_step=headline_launches_per_step, hip_graphs=hip_graphs_headline,
                               headline_mode=headline_mode,      # captured | eager | eager (capture failed ...): what `value` measures, at every N (`value_eager`: the eager step's figure beside it)
                               # companions of the same run as scalars (each is also a top-level value_* key with its details next to it)
                               value_no_prof=value_no_prof['value'] if value_no_prof else None, value_eager=value_eager['value'] if value_eager else None,
                               value_bf16_split=split3['value'] if split3 else None, value_vendor_fp32_convs=strict['value'] if strict else None,
                               value_aug_ada=ada['value'] if ada else None, value_aug_ada_captured=ada_cap['value'] if ada_cap else None, value_bf16_products=bf16c['value'] if bf16c else None,
                               value_lowp_bf16=lowpc['value'] if lowpc else None, value_pl_f1=plc['value'] if plc else None, value_hip_graphs=graphc['value'] if graphc else None,
                               conv_terms=default_terms[0], upfirdn2d_in_step_GBps=roofline_ufd['achieved'] if roofline_ufd else None,
                               upfirdn2d_in_step_frac=roofline_ufd['frac'] if roofline_ufd else None),
                   multi_gpu=multi_gpu, step_ms=step_ms, power=power.summary() if power is not None else None, value_bf16_split=split3['value'] if split3 else None, bf16_split=split3,
                   value_no_prof=value_no_prof['value'] if value_no_prof else None, no_prof=value_no_prof, value_eager=value_eager['value'] if value_eager else None, eager=value_eager,
                   value_fp32_grade=value if (default_terms in ((0, 0), (4, 4)) and lowp is None) else None,     # the headline's products are fp32-GRADE (22-bit split operands, 2.7e-7), not strict fp32: the strict-fp32 figure is value_vendor_fp32_convs
                   value_vendor_fp32_convs=strict['value'] if strict else None, vendor_fp32_convs=strict, value_aug_ada=ada['value'] if ada else None, aug_ada=ada, value_aug_ada_over_noaug_control=ada.get('ratio_to_control') if ada else None, value_aug_ada_captured=ada_cap['value'] if ada_cap else None, aug_ada_captured=ada_cap, value_bf16_products=bf16c['value'] if bf16c else None, bf16_products=bf16c, value_pl_f1=plc['value'] if plc else None, pl_f1=plc,
                   value_lowp_bf16=lowpc['value'] if lowpc else None, lowp_bf16=lowpc, value_hip_graphs=graphc['value'] if graphc else None, hip_graphs=graphc,
                   roofline=roofline, roofline_conv_family=roofline_family, roofline_upfirdn2d=roofline_ufd, upfirdn2d_by_size=ufd_by_size, kernels=kernels, kernels_by_variant=variants, cpu_baseline=cpu)
        emit(out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
