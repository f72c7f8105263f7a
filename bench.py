"""bench.py -- PAVE-Net forward throughput on MI355X (clips/s), the BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a launcher (WORLD_SIZE unset): this process starts `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD
    process before anything touches the GPU (as the reference's tools/dist_test.sh:8-10 does with
    torch.distributed.launch), relays its JSON line and exits with its return code.  Launched under
    torch.distributed.run already (WORLD_SIZE set) it is one rank of that job.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): PAVE-Net
R-50, T = 7 frames, batch = 4 clips of synthetic 800x1344 video per GPU, 300 pose queries,
K = 15 keypoints, max_per_img = 20, random weights, fp32 -- the whole ``simple_test`` path:
backbone -> neck -> 6-layer deformable encoder -> proposals/top-k -> 3-layer pose-aware T-frame
decoder -> 2-layer joint decoder -> OKS-NMS, results copied to the host.  One "step" = one such
batch.  Multi-GPU (``--shard clips``, default) is clip-parallel: independent clips per rank, weak
scaling, one RCCL all-gather of the fixed-shape results per step.  ``--shard frames`` is the
long-clip mode of BASELINE configs[4]: ONE clip of ``--frames`` frames per step, frame t on rank
t % N, the T-frame attentions merged with one small all-gather each (strong scaling; not the
headline).

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel class of the step, the
split-operand MFMA GEMM / convolution launches (2*M*N*K counted per launch by the wrappers in
pavenet_amd/ops.py, every launch bracketed by HIP events on its own stream inside the timed
region: achieved = sum of FLOP / sum of launch times against the dense bf16 MFMA peak / 6 products);
``roofline_hbm`` is the encoder deformable-attention launch (the dominant HBM-bound kernel) measured
the same way; ``cpu_baseline`` is the CPU oracle (a port of the reference's CPU path) timed on rank 0
at N = 1 on a bounded sample; ``parity`` compares clip 0 of the timed batch with that oracle run.
``roofline.clock`` is the in-kernel clock of that class (delta s_memtime / delta s_memrealtime around sampled
workgroups, taken on the -DPAVE_DIAG build of the same sources in a pass of its own after the timed region) and
``frac_of_sustained_clock_peak`` the same achieved figure against the MFMA peak AT that clock;
``fallback_ops_per_step`` / ``aten_launches_per_step`` are a census of the ATen operators one extra, un-timed step
executed on device tensors (pavenet_amd/census.py): a non-zero fallback count means a gate left the hand-written path.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
S_TOKENS = sum(h * w for h, w in LEVELS)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# dense MFMA peak of the arithmetic each --gemm mode runs the projections in (TFLOP/s, spec)
MFMA_PEAK = {'native': 157.3, 'bf16x3': 2500.0 / 6, 'bf16x2': 2500.0 / 3, 'bf16': 2500.0,
             'fp16': 2500.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frames', type=int, default=None, help='default 7 (15 with --shard frames)')
    ap.add_argument('--clips', type=int, default=None, help='clips per GPU per step (default 4; '
                                                            '1 clip in all with --shard frames)')
    ap.add_argument('--shard', choices=('clips', 'frames'), default='clips')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--max-per-img', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', choices=('nccl', 'gloo'), default=None,
                    help='create the process group and take every `world > 1` branch (rank census, result '
                         'all-gather on device tensors, max-over-ranks all-reduce, barrier, teardown) even '
                         'with --gpus 1: executes RCCL on a one-GPU box')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the short runs of the other BASELINE configurations (`extra`)')
    ap.add_argument('--no-native-side', action='store_true',
                    help='skip the --gemm native side measurement (profiling runs: one mode per trace)')
    ap.add_argument('--cpu-baseline-clips', type=int, default=3,
                    help='timed oracle clips after one small warm-up (~20 s each)')
    ap.add_argument('--cpu-threads', type=int, default=16,
                    help='torch intra-op threads of the CPU oracle leg (a 1-GPU box is a 16-CPU share of its host; '
                         "torch's default of half the host's hardware threads runs the oracle ~3 x slower)")
    ap.add_argument('--graph', type=int, default=0,
                    help='replay the forward as one hipGraph (opt-in: pays off for small batches; '
                         'the 28-frame headline batch is GPU-bound without it)')
    ap.add_argument('--tail-graph', type=int, default=0,
                    help='replay everything behind the encoder (proposals, decoders, post-processing: ~170 '
                         'launch-bound dispatches) as ONE hipGraph; backbone / neck / encoder stay eager and '
                         'keep their per-launch events (pavenet_amd.graph.TailGraphedForward; single stream, '
                         'clip-parallel only; falls back to eager if the capture fails its self-check).  Off by '
                         'default -- measured, no gain: on the 28-frame headline batch the host has queued the tail '
                         'long before the GPU reaches it (70.97 ms replayed against 70.82 eager), and on configs[1] the '
                         'replay of the ~170-node graph is SLOWER than the eager launches (12.14 against 11.29 ms; '
                         'with it, `extra` times configs[1] both ways)')
    ap.add_argument('--pipeline', type=int, default=1,
                    help='steps in flight: P > 1 runs consecutive steps (independent batches) on P HIP '
                         'streams, so the launch-bound decoder / post-processing tail of step i overlaps '
                         'the backbone of step i + 1; results are still copied to the host per step')
    ap.add_argument('--no-extra-passes', action='store_true',
                    help='skip the un-timed passes behind the timed region (ATen census, free-selection forward, '
                         'in-kernel clock pass on the diagnostic build): profiling runs, whose traces should hold the '
                         'timed steps only')
    ap.add_argument('--no-events', action='store_true',
                    help='do not bracket the tagged kernels with HIP events (A/B of the measurement overhead)')
    ap.add_argument('--gemm-select', choices=('tuned', 'default', 'tune'), default='tuned',
                    help="vendor GEMM kernel per shape: 'tuned' = the shipped TunableOp selections "
                         "(pavenet_amd/data/tunableop_gfx950.csv, no tuning at run time), 'default' = "
                         "library heuristic, 'tune' = measure now and write gpurun_out/tunableop_gfx950.csv")
    ap.add_argument('--backbone', choices=('r50', 'hrnet_w48', 'swin_l'), default='r50',
                    help="'hrnet_w48' = BASELINE configs[3] (HRNet-w48 backbone under the MulFrames head); 'swin_l' = "
                         "the reference's 2025-2-7 Swin-L config (use --frames 3 --clips 1)")
    ap.add_argument('--gemm', choices=('native', 'bf16x3', 'bf16x2', 'bf16', 'fp16'), default='bf16x3',
                    help="dense projections / convolutions: 'bf16x3' (the headline) = hand-written "
                         "exact 3-term bf16 split on the bf16 MFMA: fp32 in, fp32 accumulate, "
                         "fp32-level accuracy -- the whole golden / oracle GPU suite runs in this "
                         "mode at the fp32 tolerances; 'native' = vendor fp32-MFMA kernels, measured "
                         "beside it at N = 1 and printed as `native_fp32_mfma`; 'fp16' = fp16 "
                         "operands (BASELINE config 5's reduced-precision projections, never the "
                         "headline); see DESIGN.md")
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 15 if args.shard == 'frames' else 7
    if args.clips is None:
        args.clips = 1 if args.shard == 'frames' else 4
    return args


def spawn_ranks(args):
    """`--gpus N` (N > 1) run without a launcher: start the N ranks as ONE child process tree
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) -- never an exec,
    and before this process has imported torch or touched the GPU.  The child's stdout (rank 0's
    JSON line) is relayed; a failed, killed or hung child gives a non-zero return code."""
    import signal
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    limit = float(os.environ.get('PAVE_BENCH_CHILD_TIMEOUT', 3000))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)   # the exact process group started above
        out, _ = proc.communicate()
        sys.stdout.write(out or '')
        print(f'bench.py: the {args.gpus}-rank child did not finish within {limit:.0f} s; killed',
              file=sys.stderr)
        return 124
    # rank 0's JSON line goes to stdout, anything else the launcher or the ranks printed to stderr
    for ln in (out or '').splitlines():
        print(ln, file=sys.stdout if ln.startswith('{') else sys.stderr)
    sys.stdout.flush()
    if proc.returncode != 0:
        print(f'bench.py: the {args.gpus}-rank child exited with code {proc.returncode}', file=sys.stderr)
        return proc.returncode if proc.returncode > 0 else 1
    if not any(ln.startswith('{') for ln in (out or '').splitlines()):
        print('bench.py: the child printed no JSON line', file=sys.stderr)
        return 1
    return 0


ARGS = None
if __name__ == '__main__':
    ARGS = parse()
    if ARGS.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(ARGS))

import pavenet_amd  # noqa: E402,F401  (sets the runtime flag hipGraph replay needs, before HIP starts)
import torch  # noqa: E402


def file_git_blob_sha1(path):
    """`git hash-object` of a file (sha1 over "blob <len>\\0" + bytes), without git."""
    data = open(path, 'rb').read()
    return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


# launches of the split-operand MFMA GEMM family (pavenet_amd/csrc/pave_gemm_dma.hip, pave_gemm_split.hip)
SPLIT_GEMM_TAGS = ('gemm_bf16x3', 'gemm_bf16x3_ln', 'conv3x3_split', 'conv1x1_strided', 'conv7x7_stem',
                   'bottleneck_chain')
# the same kernels on the decoders' / heads' few-hundred-row Linears (M < ops.SMALL_ROWS = 8192: 10 row
# tiles for 256 CUs, latency-bound by shape): timed too, reported beside the class, not inside it
SMALL_GEMM_TAGS = ('gemm_bf16x3_small', 'gemm_bf16x3_ln_small')


def algorithmic_bytes_gemm_launch(tag, shape):
    """Algorithmic HBM bytes of one GEMM / convolution launch from the wrapper's shape note: the fp32
    activation rows read once, the output written once, a residual / identity read once (weights: a few
    hundred KB, not counted).  3x3 / stride s: the input map = M s^2 Cin values; the layer1 chain: c1 + identity
    (or the downsample input) in, out + the next conv1 out (its 64-channel scratch stays in L2); the stem: 4 input
    pixels x 3 channels per output pixel."""
    if not shape:
        return 0
    M, K, N = int(shape[0]), int(shape[1]), int(shape[2])
    notes = [str(x) for x in shape[3:]]
    if tag == 'bottleneck_chain':     # (M, 64, 256, cn, 'k2=..', 'tail' | '')
        cn, k2 = int(shape[3]), int(notes[1][3:])
        return 4 * M * (64 + (k2 if k2 else 256) + 256 + cn)
    if tag == 'conv7x7_stem':
        return M * (4 * 3 + 64) * 4
    a = M * K * (2 if 'a16' in notes else 4)       # (fp16 mode: an fp16 activation in / out of the launch)
    if 'o16' in notes:
        return a + M * N * 2
    for n in notes:
        if n.startswith('3x3 s'):
            s_ = int(n[5])
            a = M * s_ * s_ * (K // 9) * 4
    res = M * 32 if 'encproj' in notes else (M * N * 4 if 'res' in notes else 0)
    return a + M * N * 4 + res


def tokens_per_frame(height, width):
    """S of the 4-level pyramid the R-50 / ChannelMapper stack makes of a height x width frame: strides 8, 16, 32,
    64, every halving rounded up (22 323 at 800 x 1344, 20 906 at 750 x 1333)."""
    h, w = (height + 7) // 8, (width + 7) // 8
    s = 0
    for _ in range(4):
        s += h * w
        h, w = (h + 1) // 2, (w + 1) // 2
    return s


def algorithmic_bytes_encoder_launch(n_frames, height=800, width=1344):
    """4 B x [value S*256 + offsets/logits S*8*16*3 + out S*256] per frame-layer
    = 80.0 MB at S = 22 323; one launch covers all frames of the batch."""
    return 4 * n_frames * tokens_per_frame(height, width) * (256 + 8 * 16 * 3 + 256)


def census_of_step(model, img, metas, **kw):
    """One extra, un-timed forward under pavenet_amd.census.LaunchCensus (the caches are warm: it runs after the
    timed steps): which ATen operators still ran on device tensors."""
    from pavenet_amd.census import LaunchCensus
    with torch.no_grad(), LaunchCensus() as c:
        model.forward_device(img, metas, **kw)
    torch.cuda.synchronize()
    d = c.summary()
    return dict(fallback_ops_per_step=d['fallback_ops'], fallback_op_names=d['fallback_op_names'],
                aten_launches_per_step=d['aten_launches'], aten_launch_names=d['aten_launch_names'],
                host_syncs=d['host_syncs'], slow_paths=d['slow_paths'])


CLOCK_KINDS = ('gemm_q', 'gemm_w', 'gemm_wn', 'gemm_wn_enc', 'bottleneck_chain', 'gemm_q_ln', 'gemm_w_ln', 'stem7x7')


def class_clock_pass(run_step, steps=3):
    """The in-kernel clock of the split GEMM class under THIS workload (MI355X_MICROARCH.md, 'DVFS give-back' item 6):
    `steps` more steps on the -DPAVE_DIAG build of the same sources, whose GEMM / convolution kernels stamp
    s_memtime (shader clock) and s_memrealtime (100 MHz) around every 64th workgroup; clock = sum of shader ticks /
    sum of 100 MHz ticks x 100 MHz, over the class and per kernel.  Runs after the timed region (the chip has been
    under this load for seconds); the shipped library executes no stamp."""
    import ctypes
    from pavenet_amd import native
    try:
        with native.diag_build() as lib:
            lib.pave_diag_clock_reset.restype = ctypes.c_int
            lib.pave_diag_clock_read.restype = ctypes.c_int
            lib.pave_diag_clock_read.argtypes = [ctypes.c_void_p]
            run_step()                       # (first launches of the other binary: code objects load here)
            torch.cuda.synchronize()
            if lib.pave_diag_clock_reset() != 0:
                return dict(error='pave_diag_clock_reset failed')
            for _ in range(steps):
                run_step()
            buf = (ctypes.c_ulonglong * 16)()
            if lib.pave_diag_clock_read(ctypes.cast(buf, ctypes.c_void_p)) != 0:
                return dict(error='pave_diag_clock_read failed')
    except Exception as e:      # a measurement aid must not cost the line
        return dict(error=f'{type(e).__name__}: {e}'[:200])
    tot_s = sum(buf[2 * k] for k in range(8))
    tot_r = sum(buf[2 * k + 1] for k in range(8))
    if tot_r == 0:
        return dict(error='no stamped workgroup ran')
    by = {CLOCK_KINDS[k]: round(buf[2 * k] / buf[2 * k + 1] * 100.0, 1) for k in range(8) if buf[2 * k + 1]}
    return dict(sustained_clock_mhz=round(tot_s / tot_r * 100.0, 1), by_kernel_mhz=by, nominal_mhz=2400,
                measured='delta s_memtime / delta s_memrealtime x 100 MHz around every 64th workgroup of the class\'s '
                         f'kernels, summed over {steps} steps of this workload on the -DPAVE_DIAG build '
                         '(time-weighted over the class), after the timed region')


def clip0_image(args, frames):
    """Clip 0 of rank 0 is generated on the host (seed 0) so that the CPU oracle and the device
    path see the same frames."""
    g = torch.Generator().manual_seed(0)
    return torch.randn(1, frames, 3, args.height, args.width, generator=g)


def cpu_baseline_and_parity(model, args, frames, clip0, free_result, free_selection=None):
    """The oracle (CPU restatement of the reference's path with the torch-CPU sampler the
    reference's own CPU fallback uses) on clip 0: one small warm-up, then `cpu_baseline_clips`
    timed runs.  Its output is the parity reference for clip 0 of the timed batch."""
    from oracle import pavenet_ref as R
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cfg = dict(num_frames=frames, num_keypoints=15, num_query=300, max_per_img=args.max_per_img)
    # torch's intra-op pool defaults to half the HOST's hardware threads (128 on a GPU box, which is a 16-CPU
    # share of that host): the oracle then runs 2.8 x SLOWER than with 16 - 32 threads (tests/oracle_threads.py,
    # profiles/r05_oracle_threads.txt), so the baseline is timed on the share it has
    default_threads = torch.get_num_threads()
    threads = max(1, min(default_threads, args.cpu_threads))
    torch.set_num_threads(threads)
    R.SAMPLER = 'torch'
    with torch.no_grad():
        R.videopose_simple_test(sd, dict(cfg, num_frames=1), clip0[:, :1])   # warm-up (thread pools)
        times, taps, exp = [], {}, None
        for _ in range(max(1, args.cpu_baseline_clips)):
            taps = {}
            t0 = time.time()
            exp = R.videopose_simple_test(sd, cfg, clip0, taps=taps)
            times.append(time.time() - t0)
    torch.set_num_threads(default_threads)
    dt = sum(times) / len(times)
    base = dict(value=round(1.0 / dt, 5), unit='clips/s', cores=threads, kind='port',
                sample=f'{len(times)} timed clips (after a 1-frame warm-up) of T={frames}, '
                       f'{args.height}x{args.width}, max_per_img={args.max_per_img}, oracle/'
                       f'pavenet_ref.py with the torch-CPU grid_sample sampler, {threads} threads; '
                       f'per clip {", ".join(f"{t:.1f}" for t in times)} s')
    # ---- parity of clip 0: (a) the free-running timed batch, (b) selections pinned to the oracle's
    eb, el, ek = exp
    dev = next(model.parameters()).device
    N = args.max_per_img
    metas = [dict(batch_input_shape=(args.height, args.width),
                  img_shape=(args.height, args.width, 3), scale_factor=(1., 1., 1., 1.))]
    with torch.no_grad():
        res = model.forward_device(clip0.to(dev), metas,
                                   force_topk_proposals=taps['topk_idx'].to(dev),
                                   force_score_topk=taps['score_topk_idx'].view(1, -1).to(dev))
        (gb, gl, gk), = model.bbox_head.results_to_list(res)
    gk = gk.cpu()
    keep_equal = tuple(gk.shape) == tuple(ek.shape)
    max_px = float((gk[..., :2] - ek[..., :2]).abs().max()) if keep_equal else None
    free_sel_equal = None
    if free_selection is not None:
        free_sel_equal = bool(torch.equal(free_selection[0].cpu().view(-1), taps['topk_idx'].view(-1))
                              and torch.equal(free_selection[1].cpu().view(-1), taps['score_topk_idx'].view(-1)))
    # free run: how many of the oracle's kept poses the un-forced timed batch reproduced (within
    # 0.05 px); under random weights the two top-k selections sit on near-ties, so this is
    # reported, not asserted
    fk = free_result
    matched = 0
    for pose in ek[..., :2]:
        if fk.numel() and float((fk[..., :2] - pose).abs().amax(dim=(1, 2)).min()) < 5e-2:
            matched += 1
    free_forced = None
    if free_sel_equal is False:
        # the free run made other (near-tie) selections than the oracle: one more oracle clip with the DEVICE's
        # selections forced -- the free timed batch must then be reproduced pose for pose
        R.SAMPLER = 'torch'
        torch.set_num_threads(threads)
        with torch.no_grad():
            t2 = {'force_topk_idx': free_selection[0].cpu().view(1, -1),
                  'force_score_topk_idx': free_selection[1].cpu().view(-1)}
            _, _, ek2 = R.videopose_simple_test(sd, cfg, clip0, taps=t2)
        torch.set_num_threads(default_threads)
        same_n = tuple(ek2.shape) == tuple(fk.shape)
        free_forced = dict(keep_equal=bool(same_n), oracle_poses=int(ek2.shape[0]),
                           max_px=round(float((fk[..., :2] - ek2[..., :2]).abs().max()), 5) if same_n else None,
                           note='the oracle re-run with the free timed batch\'s own two selections forced, against '
                                'clip 0 of that batch')
    parity = dict(max_px=None if max_px is None else round(max_px, 5), keep_equal=bool(keep_equal),
                  oracle_poses=int(ek.shape[0]), free_run_poses_matched=matched,
                  free_run_selection_is_the_oracles=free_sel_equal, free_run_vs_oracle_with_its_selections=free_forced,
                  note='clip 0 vs the CPU oracle; max_px with the oracle\'s top-k selections pinned.  The free '
                       '(un-pinned) timed run reproduces the oracle\'s poses only where it makes the oracle\'s '
                       'selections: the 300 proposals are a sorted list and query i adds its own embedding to '
                       'proposal i, so two logits ~1e-6 apart that swap places give two queries other inputs '
                       '(tests/test_model_gpu.py::_full_size_vs_oracle re-runs the oracle with the device\'s '
                       'selections forced in that case)')
    return base, parity


def secondary_workloads(args, dev, budget_s=110.0):
    """The other BASELINE configurations on the driver's clock (`extra` of the JSON line): short
    single-stream runs -- 3 warm-up + 10 timed steps each, results copied to the host per step as in
    the headline, the split-GEMM class timed by HIP events over the last 3 steps -- of configs[1]
    (R-50, T = 3, one clip), configs[3] on ONE GPU (HRNet-w48, T = 7, 4 clips) and the configs[4]
    shape on ONE GPU (R-50, T = 15, one clip; exact split GEMMs and fp16-operand projections).
    No vendor-kernel side, no oracle; a workload that would start after `budget_s` is skipped and
    says so."""
    from pavenet_amd import ops
    from pavenet_amd.bricks import set_gemm_mode
    from pavenet_amd.models import build_model, videopose_r50_cfg, with_hrnet_w48
    from pavenet_amd.weights import init_random_weights
    # the headline workload as a PADDED batch (what the reference's test pipeline produces: valid 800 x 1333 /
    # 750 x 1333 images in the 800 x 1344 batch, configs/_base_/datasets/coco_keypoint.py:79): two runs of clips
    # with their own masks, positional tables and valid ratios
    pad = [(args.height, args.width - 11)] * 2 + [(args.height - 50, args.width - 11)] * 2
    todo = [('configs[2] as a padded batch (img_shape 2 x 800x1333, 2 x 750x1333)', 'r50', 7, 4, 'bf16x3', pad, None),
            ('configs[1]', 'r50', 3, 1, 'bf16x3', None, None),
            # the canvas the reference's own video test pipeline produces: keep-ratio resize to (1333, 800) and
            # Pad(size_divisor=1), i.e. NO padding -- a 1080p frame is a 750 x 1333 batch, odd width
            # (configs/_base_/datasets/posetrack17_video_keypoint.py:68-84)
            ('configs[1] on the reference\'s PoseTrack test canvas (750x1333, size_divisor=1)', 'r50', 3, 1, 'bf16x3',
             None, (750, 1333)),
            ('configs[3] on one GPU', 'hrnet_w48', 7, 4, 'bf16x3', None, None),
            ('configs[4] shape on one GPU', 'r50', 15, 1, 'bf16x3', None, None),
            ('configs[4] shape on one GPU, fp16-operand projections', 'r50', 15, 1, 'fp16', None, None),
            # the reference's flagship backbone (configs/videopose/2025-2-7/2025_2_7_swin_num_frames_3_posetrack17.py)
            ('Swin-L T=3 (the reference\'s 2025-2-7 config)', 'swin_l', 3, 1, 'bf16x3', None, None)]
    out, t_begin, model, key = [], time.perf_counter(), None, None
    steps, warmup, ev_steps = 10, 3, 3
    for name, backbone, T, B, gemm, shapes, canvas in todo:
        bname = {'r50': 'R-50', 'hrnet_w48': 'HRNet-w48', 'swin_l': 'Swin-L'}[backbone]
        H_, W_ = canvas if canvas is not None else (args.height, args.width)
        label = f'{name}: PAVE-Net {bname} T={T}, batch={B} clips, {H_}x{W_}, gemm={gemm}'
        if time.perf_counter() - t_begin > budget_s:
            out.append(dict(workload=label, skipped=f'secondary budget of {budget_s:.0f} s used up'))
            continue
        if key != (backbone, T):
            model = None
            torch.cuda.empty_cache()
            mcfg = videopose_r50_cfg(num_frames=T, max_per_img=args.max_per_img)
            if backbone == 'hrnet_w48':
                mcfg = with_hrnet_w48(mcfg)
            elif backbone == 'swin_l':
                from pavenet_amd.models import with_swin_l
                mcfg = with_swin_l(mcfg, num_frames=T)
            model = init_random_weights(build_model(mcfg), seed=0).to(dev).eval()
            key = (backbone, T)
        set_gemm_mode(gemm)
        metas = [dict(batch_input_shape=(H_, W_), img_shape=(shapes[i] if shapes else (H_, W_)) + (3,),
                      scale_factor=(1., 1., 1., 1.)) for i in range(B)]
        g = torch.Generator(device=dev).manual_seed(4321)
        img = torch.randn(B, T, 3, H_, W_, device=dev, generator=g)
        host = None
        tail, tail_note = None, 'off'
        use_tail = [False]
        if args.tail_graph and B * T <= 3:      # a latency-shaped workload, also timed with the tail replayed
            from pavenet_amd import GRAPH_REPLAY_SAFE
            from pavenet_amd.graph import TailGraphedForward
            try:
                if GRAPH_REPLAY_SAFE:
                    tail = TailGraphedForward(model, img, metas)
                    with torch.no_grad():
                        a_, b_ = model.forward_device(img, metas), tail(img)
                        torch.cuda.synchronize()
                    if all(torch.equal(a_[k], b_[k]) for k in ('bboxes', 'kpts', 'keep')):
                        tail_note = 'on'
                    else:
                        tail, tail_note = None, 'off: the replay differed from the eager forward'
            except Exception as e:
                tail, tail_note = None, f'off: capture failed ({type(e).__name__}: {e})'[:200]

        def step():
            nonlocal host
            res = tail(img) if (tail is not None and use_tail[0]) else model.forward_device(img, metas)
            packed = torch.cat([res['bboxes'].flatten(1), res['kpts'].flatten(1), res['keep'].float()], dim=1)
            if host is None:
                host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            host.copy_(packed, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        try:
            with torch.no_grad():
                for _ in range(warmup):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps - ev_steps):
                    step()
                ops.KERNEL_EVENT_TAGS = SPLIT_GEMM_TAGS      # (rows >= 8192, as the headline's class)
                ops.KERNEL_EVENTS, ops.KERNEL_EVENT_SHAPES = [], []
                for _ in range(ev_steps):
                    step()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        except Exception as e:      # an `extra` workload must not take the headline line with it
            ops.KERNEL_EVENTS = ops.KERNEL_EVENT_SHAPES = None
            out.append(dict(workload=label, error=f'{type(e).__name__}: {e}'[:300]))
            del img
            continue
        ev, shapes = ops.KERNEL_EVENTS, ops.KERNEL_EVENT_SHAPES
        ops.KERNEL_EVENTS = ops.KERNEL_EVENT_SHAPES = None
        tt = sum(s_.elapsed_time(e_) for _, s_, e_, _ in ev) * 1e-3
        fl = sum(f for _, _, _, f in ev)
        nbytes = sum(algorithmic_bytes_gemm_launch(e_[0], sh) for e_, sh in zip(ev, shapes))
        peak = MFMA_PEAK[gemm]
        rec = dict(workload=label, steps=steps, warmup=warmup,
                   ms_per_step=round(dt / steps * 1e3, 3), clips_per_s=round(B * steps / dt, 3),
                   split_class_tflops=round(fl / tt / 1e12, 1) if tt > 0 else None,
                   split_class_ms_per_step=round(tt / ev_steps * 1e3, 3),
                   split_class_tflop_per_step=round(fl / ev_steps / 1e12, 3),
                   split_class_gb_per_step=round(nbytes / ev_steps / 1e9, 2),
                   split_class_gbps=round(nbytes / tt / 1e9, 1) if tt > 0 else None)
        if gemm == 'fp16':
            # one fp16 product per tile: these launches move fp32 activations and are HBM-bound, so the class is
            # scored against the HBM peak on its algorithmic bytes (the MFMA fraction is printed for reference)
            rec.update(bound='hbm', peak_gbps=HBM_PEAK_GBS,
                       frac=round(nbytes / tt / 1e9 / HBM_PEAK_GBS, 4) if tt > 0 else None,
                       mfma_frac=round(fl / tt / 1e12 / peak, 4) if tt > 0 else None,
                       bytes_are='fp32 activation rows read once + output written once + residual read once per '
                                 'launch (bench.algorithmic_bytes_gemm_launch)')
        else:
            rec.update(bound='mfma', peak_tflops=round(peak, 1),
                       frac=round(fl / tt / 1e12 / peak, 4) if tt > 0 else None)
        try:
            rec.update(census_of_step(model, img, metas))
        except Exception as e:
            rec['census_error'] = f'{type(e).__name__}: {e}'[:200]
        if tail is not None:
            # the same workload with everything behind the encoder replayed as ONE hipGraph (TailGraphedForward)
            use_tail[0] = True
            try:
                with torch.no_grad():
                    for _ in range(warmup):
                        step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        step()
                    torch.cuda.synchronize()
                    rec['tail_graph'] = dict(ms_per_step=round((time.perf_counter() - t0) / steps * 1e3, 3),
                                             note='backbone / neck / encoder eager, proposals + decoders + '
                                                  'post-processing replayed as one hipGraph per step')
            except Exception as e:
                rec['tail_graph'] = dict(error=f'{type(e).__name__}: {e}'[:200])
        elif tail_note != 'off':
            rec['tail_graph'] = dict(note=tail_note)
        out.append(rec)
        tail = None
        del img
    model = None
    torch.cuda.empty_cache()
    set_gemm_mode(args.gemm)
    return out


def main():
    args = ARGS if ARGS is not None else parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    # PAVE_BENCH_ONE_DEVICE=1 (+ gloo): every rank on GPU 0, to exercise the N > 1 path on a 1-GPU box
    one_device = os.environ.get('PAVE_BENCH_ONE_DEVICE', '0') == '1'
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # MIOpen's per-shape solver search only where MIOpen runs at all: the vendor-kernel modes (the headline mode
    # launches no library convolution -- `fallback_ops_per_step` on the line says so; the `native_fp32_mfma` side
    # measurement switches the search on for its own steps)
    torch.backends.cudnn.benchmark = args.gemm == 'native'
    dev = torch.device('cuda', local_rank)
    dist = None
    host_collectives = False
    backend = None
    ranks_seen, devices = [0], [f'cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}']
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        backend = args.force_dist or ('gloo' if one_device else 'nccl')   # RCCL refuses two ranks on one device
        if 'WORLD_SIZE' not in os.environ:   # --force-dist with no launcher: a one-rank rendezvous
            import socket
            sock = socket.socket()
            sock.bind(('127.0.0.1', 0))
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                              MASTER_PORT=str(sock.getsockname()[1]))
            sock.close()
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
        host_collectives = backend == 'gloo'
        # who is in the job: one all-gather of the rank ids on the data-path backend, one of the
        # device each rank drives
        ids = torch.full((1,), rank, dtype=torch.int64, device='cpu' if host_collectives else dev)
        seen = torch.empty((world,), dtype=torch.int64, device=ids.device)
        dist.all_gather_into_tensor(seen, ids)
        ranks_seen = [int(v) for v in seen.cpu().tolist()]
        devices = [None] * world
        dist.all_gather_object(devices, f'cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}')
        # the job is what the line will say it is: every rank 0 .. world-1 present exactly once, and -- outside the
        # one-device rehearsal mode -- every rank on a device of its own (every rank sees the same lists, so all of
        # them leave together; the reference's launcher gives rank i GPU i: tools/dist_test.sh:8-10)
        if sorted(ranks_seen) != list(range(world)):
            raise SystemExit(f'bench.py: rank census {ranks_seen} is not a permutation of 0..{world - 1}')
        if not one_device and len(set(d.split(' ')[0] for d in devices)) != world:
            raise SystemExit(f'bench.py: {world} ranks but the devices are not pairwise distinct: {devices}')

    from pavenet_amd import ops
    from pavenet_amd.models import build_model, videopose_r50_cfg
    from pavenet_amd.weights import init_random_weights

    T, B = args.frames, args.clips
    frame_sharded = args.shard == 'frames'
    mcfg = videopose_r50_cfg(num_frames=T, max_per_img=args.max_per_img)
    if args.backbone == 'hrnet_w48':
        from pavenet_amd.models import with_hrnet_w48
        mcfg = with_hrnet_w48(mcfg)
    elif args.backbone == 'swin_l':
        from pavenet_amd.models import with_swin_l
        mcfg = with_swin_l(mcfg, num_frames=T)
    model = build_model(mcfg)
    init_random_weights(model, seed=0)
    model = model.to(dev).eval()
    from pavenet_amd.bricks import set_gemm_mode
    set_gemm_mode(args.gemm)
    if args.gemm_select != 'default':
        from pavenet_amd import tuning
        if args.gemm_select == 'tune':
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            tuning.use_tuned_gemms(os.path.join(ROOT, 'gpurun_out', 'tunableop_gfx950.csv'), tune=True)
        else:
            tuning.use_tuned_gemms()
    metas = [dict(batch_input_shape=(args.height, args.width),
                  img_shape=(args.height, args.width, 3), scale_factor=(1., 1., 1., 1.))
             for _ in range(B)]
    N, K = args.max_per_img, 15
    shard = None
    if frame_sharded:
        # ONE clip for the whole job (same seed on every rank); this rank keeps frames t % world == rank
        from pavenet_amd.dist import FrameShard
        g = torch.Generator(device=dev).manual_seed(1234)
        img = torch.randn(B, T, 3, args.height, args.width, device=dev, generator=g)
        if dist is not None:
            shard = FrameShard(T, rank, world)
            img = img[:, shard.local].contiguous()
        clip0 = None
    else:
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        img = torch.randn(B, T, 3, args.height, args.width, device=dev, generator=g)
        clip0 = clip0_image(args, T) if rank == 0 else None
        if clip0 is not None:
            img[0].copy_(clip0[0])

    graphed = None
    if args.graph:
        from pavenet_amd.graph import GraphedForward
        graphed = GraphedForward(model, img, metas)
    tail_graph, tail_note = None, None
    if args.tail_graph and not args.graph and shard is None and args.pipeline == 1 and not frame_sharded:
        from pavenet_amd import GRAPH_REPLAY_SAFE
        from pavenet_amd.graph import TailGraphedForward
        if not GRAPH_REPLAY_SAFE:
            tail_note = 'off: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not set before HIP started'
        else:
            try:
                tail_graph = TailGraphedForward(model, img, metas)
                # self-check on the timed batch: the replay must give what the eager forward gives
                with torch.no_grad():
                    ref_res = model.forward_device(img, metas)
                    got_res = tail_graph(img)
                    torch.cuda.synchronize()
                same = all(torch.equal(ref_res[k], got_res[k]) for k in ('bboxes', 'kpts', 'keep'))
                if not same:
                    tail_graph, tail_note = None, 'off: the replay differed from the eager forward'
                else:
                    tail_note = 'on: proposals / decoders / post-processing replayed as one hipGraph per step'
            except Exception as e:      # a capture problem must not cost the run
                tail_graph, tail_note = None, f'off: capture failed ({type(e).__name__}: {e})'[:200]

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(2, args.pipeline))]
    TAIL_ON = [True]     # (the two-batches-in-flight side run is eager: one capture serves one stream)

    host_bufs = {}

    def to_host(packed, slot):
        """results on the host, as simple_test returns them, in a pinned buffer; slot >= 0: the copy
        is asynchronous (the caller waits on the stream's event)"""
        buf = host_bufs.get(slot)
        if buf is None or buf.shape != packed.shape:
            buf = host_bufs[slot] = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        if slot < 0:    # single stream: the step ends when its results are on the host (pinned
            torch.cuda.current_stream().synchronize()   # buffer: no staging copy through pageable memory)
        return buf

    def step(slot=-1):
        if graphed is not None:
            res = graphed(img)
        elif tail_graph is not None and TAIL_ON[0]:
            res = tail_graph(img)
        elif shard is not None:
            res = model.forward_device(img, metas, frame_shard=shard)
        else:
            res = model.forward_device(img, metas)
        packed = torch.cat([res['bboxes'].flatten(1), res['kpts'].flatten(1),
                            res['keep'].float()], dim=1)  # [B, N*5 + N*K*3 + N]
        if dist is not None and not frame_sharded:   # clip-parallel: gather every rank's clips
            if host_collectives:
                packed = packed.cpu()
                out = torch.empty((world * packed.shape[0], packed.shape[1]))
                dist.all_gather_into_tensor(out, packed)
                return out
            out = torch.empty((world * packed.shape[0], packed.shape[1]), device=dev)
            dist.all_gather_into_tensor(out, packed)   # RCCL over xGMI
            packed = out
        return to_host(packed, slot)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n, P):
        """n steps; P > 1: step k runs on HIP stream k % P and its (asynchronous, pinned) host copy is
        awaited only when the stream is needed again or at the end -- P independent batches in
        flight, the launch-bound decoder / post-processing tail of one overlapping the backbone
        of the next."""
        if P == 1:
            out = None
            for _ in range(n):
                out = step()
            return out
        pending, out = [None] * P, None
        cur = torch.cuda.current_stream()
        for k in range(n):
            st = streams[k % P]
            if pending[k % P] is not None:
                pending[k % P].synchronize()
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out = step(k % P)     # asynchronous host copy into the slot's pinned buffer
                ev = torch.cuda.Event()
                ev.record(st)
            pending[k % P] = ev
        for ev in pending:
            if ev is not None:
                ev.synchronize()
        return out

    RANK_DT = []     # per-rank wall time of the last timed region (N > 1)

    def timed(record_events, P=None):
        P = max(1, args.pipeline) if P is None else P
        run_steps(args.warmup, P)
        sync()
        # tagged launches record (start, end) HIP events on the stream they launch on -- during the
        # LAST `ev_steps` of the K timed steps (bracketing ~90 launches per step costs ~0.5 ms a step)
        ev_on = record_events and graphed is None and not args.no_events
        ev_steps = min(3, args.steps) if ev_on else 0
        t0 = time.perf_counter()
        if args.steps - ev_steps > 0:
            out = run_steps(args.steps - ev_steps, P)
        if ev_on:
            ops.KERNEL_EVENT_TAGS = ('enc_tile', 'enc_grid_T1') + SPLIT_GEMM_TAGS + SMALL_GEMM_TAGS
            ops.KERNEL_EVENTS, ops.KERNEL_EVENT_SHAPES = [], []
            out = run_steps(ev_steps, P)
        sync()
        dt = time.perf_counter() - t0
        ev = ops.KERNEL_EVENTS or []
        # (one shape note per recorded launch, in the same order)
        ev = [e_ + (sh,) for e_, sh in zip(ev, ops.KERNEL_EVENT_SHAPES or [None] * len(ev))]
        ops.KERNEL_EVENTS = ops.KERNEL_EVENT_SHAPES = None
        if dist is not None:
            tt = torch.tensor([dt], device='cpu' if host_collectives else dev, dtype=torch.float64)
            every = torch.empty((world,), device=tt.device, dtype=torch.float64)
            dist.all_gather_into_tensor(every, tt)         # each rank's own clock: a straggler shows on the line
            RANK_DT[:] = [float(v) for v in every.cpu().tolist()]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, ev, out

    native_dt = None
    if args.gemm != 'native' and graphed is None and not args.no_native_side and world == 1:
        # the same workload on the vendor fp32-MFMA kernels, printed beside the headline
        set_gemm_mode('native')
        TAIL_ON[0] = False          # (the captured tail holds the headline mode's kernels)
        torch.backends.cudnn.benchmark = True       # (MIOpen times its solvers once, during this leg's warm-up)
        native_dt, _, _ = timed(False)
        torch.backends.cudnn.benchmark = False
        TAIL_ON[0] = True
        set_gemm_mode(args.gemm)
    dt, events, last = timed(True)
    last = last.clone()     # (the pinned result buffer is reused by later runs)
    rank_dt = list(RANK_DT)
    ev_steps_main = (min(3, args.steps) if (graphed is None and not args.no_events) else 0)
    pipe_dt = None
    if args.pipeline == 1 and world == 1 and graphed is None and not args.no_native_side:
        # the same K steps with two batches in flight (two HIP streams): throughput of a serving loop;
        # reported beside the headline, whose kernels run alone (clean per-kernel event times)
        TAIL_ON[0] = False
        pipe_dt, _, _ = timed(False, 2)
        TAIL_ON[0] = True
    # un-timed passes behind the timed regions, on every rank (the steps hold the job's collectives):
    # (a) the ATen-operator census of one forward, (b) the class's in-kernel clock on the diagnostic build
    # (neither pass may leave the ranks of a multi-rank job with different collective counts if it fails on one of
    # them: the census runs where the forward holds no collective -- every mode but the frame-sharded one --, the
    # clock pass, whose steps include the result all-gather, at N = 1 only)
    free_selection = None
    if rank == 0 and shard is None and not args.no_extra_passes \
            and hasattr(model.bbox_head.transformer, 'last_topk_proposals'):
        with torch.no_grad():      # (the un-pinned selections of clip 0, as the timed steps made them)
            r_ = model.forward_device(img, metas)
        free_selection = (model.bbox_head.transformer.last_topk_proposals[0].clone(), r_['score_index'][0].clone())
    census = None
    if shard is None and not args.no_extra_passes:
        try:
            census = census_of_step(model, img, metas)
        except Exception as e:
            census = dict(census_error=f'{type(e).__name__}: {e}'[:200])
    clock = None
    if dist is None and args.gemm != 'native' and graphed is None and tail_graph is None and not args.no_events \
            and not args.no_extra_passes:
        clock = class_clock_pass(lambda: step())
    timed_ev = [(tag, s.elapsed_time(e) * 1e-3, fl) for tag, s, e, fl, _ in events]
    shaped_ev = [(tag, s.elapsed_time(e) * 1e-3, fl, sh) for tag, s, e, fl, sh in events]
    enc = [(tag, t) for tag, t, _ in timed_ev if tag in ('enc_tile', 'enc_grid_T1')]
    n_frames = img.shape[0] * img.shape[1]   # frames this rank encodes per step
    roofline_mfma = None
    gm = [(tag, t, fl) for tag, t, fl in timed_ev if tag in SPLIT_GEMM_TAGS]
    if gm and args.gemm != 'native':
        tot_t, tot_f = sum(t for _, t, _ in gm), sum(fl for _, _, fl in gm)
        peak = MFMA_PEAK[args.gemm]
        by = {}
        for tag, t, fl in gm:
            d = by.setdefault(tag, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += t
            d[2] += fl
        roofline_mfma = dict(
            bound='mfma',
            kernel='gemm_q / gemm_w / gemm_wn(_enc) / gemm_w_ln / gemm_q_ln / bottleneck_chain / stem7x7_qr '
                   'kernels (pave_gemm_dma.hip, the LDS-DMA split GEMM): every Linear / FFN / 1x1, 3x3, 7x7 '
                   'convolution launch of the step with >= 8192 rows (the decoders\' few-hundred-row '
                   'launches of the same kernels: small_row_launches)' if args.gemm == 'bf16x3' else
                   'split-operand GEMM family (pave_gemm_split.hip): every Linear / FFN / 1x1, 3x3, 7x7 '
                   'convolution launch of the step',
            achieved=round(tot_f / tot_t / 1e12, 1), peak=round(peak, 1), unit='TFLOP/s',
            frac=round(tot_f / tot_t / 1e12 / peak, 4), traffic=None,
            achieved_is='sum of 2*M*N*K over the launches (real K of the stem: 147) / sum of their '
                        'HIP-event durations, inside the timed steps; fp32-equivalent FLOP',
            peak_is='2500 TFLOP/s dense bf16 MFMA / 6 products per fp32 product' if args.gemm == 'bf16x3'
                    else 'dense MFMA peak of the --gemm mode',
            launches_per_step=round(len(gm) / max(1, ev_steps_main), 1),
            ms_per_step=round(tot_t / max(1, ev_steps_main) * 1e3, 3),
            tflop_per_step=round(tot_f / max(1, ev_steps_main) / 1e12, 3),
            measured_over=f'the last {ev_steps_main} of the {args.steps} timed steps',
            by_entry_point={k: dict(launches_per_step=round(v[0] / max(1, ev_steps_main), 1),
                                    ms_per_step=round(v[1] / max(1, ev_steps_main) * 1e3, 3),
                                    tflops=round(v[2] / v[1] / 1e12, 1)) for k, v in sorted(by.items())})
        # the single dominant kernel of the class on its own: the launches of ONE shape that take the most time
        # (the wide-tile row GEMM gemm_w_kernel<0> on the encoder's FFN1 shape in the headline workload)
        groups = {}
        for tag, t, fl, sh in shaped_ev:
            if tag in SPLIT_GEMM_TAGS and sh:
                d = groups.setdefault((tag,) + tuple(str(x) for x in sh), [0, 0.0, 0.0])
                d[0] += 1
                d[1] += t
                d[2] += fl
        if groups:
            gk, gv = max(groups.items(), key=lambda kv: kv[1][1])
            roofline_mfma['dominant_kernel'] = dict(
                launch=f'{gk[0]} M x K x N = {gk[1]} x {gk[2]} x {gk[3]} ' + ' '.join(x for x in gk[4:] if x),
                kernel='gemm_w_kernel<0> (wide tile form: 32 rows x 256 columns per wave)'
                       if gk[0] == 'gemm_bf16x3' and int(gk[3]) % 256 == 0 else gk[0],
                launches_per_step=round(gv[0] / max(1, ev_steps_main), 1),
                avg_us=round(gv[1] / gv[0] * 1e6, 1), ms_per_step=round(gv[1] / max(1, ev_steps_main) * 1e3, 3),
                achieved=round(gv[2] / gv[1] / 1e12, 1), peak=round(peak, 1), unit='TFLOP/s',
                frac=round(gv[2] / gv[1] / 1e12 / peak, 4))
        if clock is not None:
            roofline_mfma['clock'] = clock
            mhz = clock.get('sustained_clock_mhz')
            if mhz:
                # "bound by power" as a number on this line: the class's own clock under this workload, and the same
                # achieved TFLOP/s against the MFMA peak AT that clock (peak x clock / 2400 MHz)
                roofline_mfma['sustained_clock_mhz'] = mhz
                roofline_mfma['frac_of_sustained_clock_peak'] = round(tot_f / tot_t / 1e12 / (peak * mhz / 2400.0), 4)
                dk = roofline_mfma.get('dominant_kernel')
                dmhz = clock['by_kernel_mhz'].get('gemm_w' if (dk and 'gemm_w_kernel' in dk.get('kernel', ''))
                                                  else 'gemm_w_ln')
                if dk and dmhz:
                    dk['sustained_clock_mhz'] = dmhz
                    dk['frac_of_sustained_clock_peak'] = round(dk['achieved'] / (peak * dmhz / 2400.0), 4)
        sm = [(t, fl) for tag, t, fl in timed_ev if tag in SMALL_GEMM_TAGS]
        if sm:
            st_, sf_ = sum(t for t, _ in sm), sum(fl for _, fl in sm)
            roofline_mfma['small_row_launches'] = dict(
                note='the same kernels at M < 8192 rows (decoder / head Linears): latency-bound by shape, '
                     'reported beside the class',
                launches_per_step=round(len(sm) / max(1, ev_steps_main), 1),
                ms_per_step=round(st_ / max(1, ev_steps_main) * 1e3, 3),
                tflop_per_step=round(sf_ / max(1, ev_steps_main) / 1e12, 4),
                tflops=round(sf_ / st_ / 1e12, 1))
    if graphed is not None:
        roofline = dict(skipped='graph replay: kernels are not launched through the timed wrappers')
    elif enc:
        avg = sum(t for _, t in enc) / len(enc)
        alg = algorithmic_bytes_encoder_launch(n_frames, args.height, args.width)
        achieved = alg / avg / 1e9
        tile = enc[0][0] == 'enc_tile'
        roofline = dict(bound='hbm',
                        kernel=('enc_tile_kernel' if tile else 'enc_head_major_kernel') +
                               ' (encoder MSDA: fused softmax + locations + bilinear sampling, T=1)',
                        achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
                        achieved_is='algorithmic bytes / measured launch time',
                        launches=len(enc), avg_us=round(avg * 1e6, 1),
                        algorithmic_bytes_per_launch=alg)
        tj = os.path.join(ROOT, 'profiles', 'enc_kernel_traffic.json')
        if os.path.exists(tj):   # HBM bytes per launch from rocprofv3 PMC passes of THIS workload
            t = json.load(open(tj))
            src = os.path.join(ROOT, 'pavenet_amd', 'csrc', 'pave_enc_tile.hip')
            # the figure is kept only while the kernel source it was measured on is unchanged
            if t.get('kernel', '').startswith(roofline['kernel'].split(' ')[0]) and \
                    t.get('frames_per_launch') == n_frames and \
                    t.get('kernel_source_git_blob') == file_git_blob_sha1(src):
                roofline['traffic'] = t.get('hbm_bytes_per_launch')
                roofline['traffic_source'] = 'profiles/enc_kernel_traffic.json (offline rocprofv3 ' \
                                             '--pmc passes over bench.py, tools/pmc_bench_enc.sh; ' \
                                             'pave_enc_tile.hip blob ' + t['kernel_source_git_blob'][:10] + ')'
    else:
        roofline = None
    if rank == 0:
        clips = (B if frame_sharded else B * world) * args.steps
        unit_note = 'frame-sharded x%d (1 clip in all)' % world if frame_sharded else f'clip-parallel x{world}'
        line = dict(metric=f'clips/sec (T={T}, {args.height}x{args.width}) fwd', value=round(clips / dt, 4),
                    unit='clips/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                    scaling='strong' if frame_sharded else 'weak', vs_baseline=None,
                    dtype={'native': 'f32', 'bf16x3': 'f32 (exact 3-term bf16 split on the bf16 MFMA)',
                           'bf16x2': 'bf16x2 operands, f32 accumulate', 'bf16': 'bf16 operands, f32 '
                           'accumulate', 'fp16': 'f16 operands, f32 accumulate'}[args.gemm],
                    data='synthetic',
                    config=dict(workload=f'PAVE-Net {dict(r50="R-50", hrnet_w48="HRNet-w48", swin_l="Swin-L")[args.backbone]} T={T} frames, '
                                         f'batch={B} clips{"" if frame_sharded else "/GPU"}, '
                                         f'{args.height}x{args.width}, Q=300, K=15, '
                                         f'max_per_img={N}, fwd simple_test incl. OKS-NMS',
                                parallelism=unit_note, gemm=args.gemm, tail_graph=tail_note,
                                detections_last_step=int(last[..., -N:].sum().item())),
                    backend=backend, ranks_seen=ranks_seen, devices=devices,
                    roofline=roofline_mfma if roofline_mfma is not None else roofline,
                    roofline_hbm=roofline)
        if args.gemm == 'native':      # (the only mode that runs TunableOp-selected vendor GEMMs)
            line['config']['gemm_select'] = args.gemm_select
        if census is not None:
            line.update(census)
        if rank_dt:
            line['rank_ms_per_step'] = dict(min=round(min(rank_dt) / args.steps * 1e3, 3),
                                            max=round(max(rank_dt) / args.steps * 1e3, 3),
                                            by_rank=[round(v / args.steps * 1e3, 3) for v in rank_dt],
                                            note='each rank\'s own wall clock over the timed region; `ms_per_step` '
                                                 'is the max')
        if args.pipeline > 1:
            line['config']['pipeline'] = f'{args.pipeline} steps in flight on {args.pipeline} HIP streams'
        if pipe_dt is not None:
            line['two_batches_in_flight'] = dict(
                value=round(clips / pipe_dt, 4), unit='clips/s',
                ms_per_step=round(pipe_dt / args.steps * 1e3, 3),
                note='same run, same K steps issued alternately on 2 HIP streams (--pipeline 2): the '
                     'launch-bound decoder / post-processing tail of one batch overlaps the backbone '
                     'of the next; kernel event times are not meaningful in this mode, so the '
                     'headline and its roofline are the single-stream figures')
        if native_dt is not None:
            line['native_fp32_mfma'] = dict(value=round(clips / native_dt, 4), unit='clips/s',
                                            ms_per_step=round(native_dt / args.steps * 1e3, 3),
                                            gemm_select=args.gemm_select,
                                            note='same run, same inputs, --gemm native (vendor '
                                                 'fp32-MFMA GEMM / convolution kernels)')
        if args.backbone == 'r50' and (args.height, args.width) == (800, 1344):
            # SURVEY 8d dense (MFMA) work: per frame R-50 176 + neck 9 + encoder 201 GFLOP, per clip
            # proposals 31 + T x (8.8 + 5.9) decoder value projections
            flops = B * (T * (176 + 9 + 201) + 31 + T * 14.7) * 1e9
            if frame_sharded:
                flops /= world
            tf = flops * args.steps / dt / 1e12
            peak = MFMA_PEAK[args.gemm]
            line['dense_mfma'] = dict(tflop_per_step_per_gpu=round(flops / 1e12, 3),
                                      achieved=round(tf, 1), peak=round(peak, 1), unit='TFLOP/s',
                                      frac=round(tf / peak, 4),
                                      note='whole step incl. the non-MFMA kernels; peak = dense MFMA '
                                           'peak of the --gemm mode (in the split modes the Cin, Cout % 64 == 0 3x3 convolutions run on the same split kernel)')
        if world == 1 and dist is None and not args.no_secondary and graphed is None and \
                args.backbone == 'r50' and (T, B) == (7, 4) and args.gemm == 'bf16x3':
            line['extra'] = secondary_workloads(args, dev)
        if world == 1 and not args.no_cpu_baseline and clip0 is not None:
            kept = last[0, -N:] > 0.5
            free_kpts = last[0, N * 5:N * 5 + N * K * 3].view(N, K, 3)[kept]
            line['cpu_baseline'], line['parity'] = cpu_baseline_and_parity(model, args, T, clip0,
                                                                           free_kpts, free_selection)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
