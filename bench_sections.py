"""bench_sections.py - the measured sections of bench.py, each run in a CHILD process (`bench.py --child <group>`).

bench.py itself is the GPU-free parent (spawns children, assembles the contract line); everything that imports numpy /
torch / the HIP library lives here. Groups: render | cpu | train | attack | extras | selftest.
"""
import json
import os
import sys
import time


def _usable_cpus():
    """CPUs this process may actually use: the cgroup quota (cpu.max) if there is one, else the affinity mask."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# The GPU boxes report 256 CPUs but grant a 16-CPU cgroup quota. OpenMP / OpenBLAS pools sized for 256 then spin through
# the quota and CFS throttles the whole process in 100 ms periods - seen as training steps of 30-110 ms instead of 8
# (the launch thread simply did not run). Size the pools to what the process may use, BEFORE numpy / torch create them.
# (NERFAIL_BENCH_CPUS: set by the parent for the `cpu` child that runs beside the render child.)
N_CPU = max(1, _usable_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))     # per rank of this node
if os.environ.get('NERFAIL_BENCH_CPUS'):
    N_CPU = max(1, int(os.environ['NERFAIL_BENCH_CPUS']))
    for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[_v] = str(N_CPU)
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, str(N_CPU))

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

np = torch = dist = synth = None         # bound by _heavy_imports() in a child; the parent never imports torch


def _heavy_imports(gpu=True):
    """numpy / torch / the synthetic-scene helpers, in a CHILD only. gpu=True also installs the guard-page allocator when
    NERFAIL_GUARD_ALLOC=1 (tests/guard: debugging aid, must precede the first device allocation)."""
    global np, torch, dist, synth
    import numpy as _np
    import torch as _torch
    import torch.distributed as _dist
    import synth as _synth
    np, torch, dist, synth = _np, _torch, _dist, _synth
    if gpu and os.environ.get('NERFAIL_GUARD_ALLOC') == '1':       # debugging aid under tests/: never a product dependency
        import guard
        guard.install_if_wanted()


H = W = 800
N_SAMPLES, N_IMPORTANCE = 64, 128
NET_D, NET_W = 8, 256
# MACs per sample = 63*256 + 4*256^2 + 319*256 + 2*256^2 (pts) + 256 (alpha)
# + 65536 (feature) + 283*128 (views) + 384 (rgb) = 593 408
FLOP_PER_SAMPLE = 2 * 593408
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; the contract is ONE JSON line there. While the
    process group is set up (and the first collective creates the communicator) fd 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def kernel_source_hashes():
    """sha256 (16 hex digits) of every file of nerfail_amd/csrc: ties a stored PMC measurement to the kernel sources it was
    taken from, file by file (a change to the kNN kernel does not invalidate the MLP kernel's counters)."""
    import hashlib
    d = os.path.join(ROOT, 'nerfail_amd', 'csrc')
    return {f: hashlib.sha256(open(os.path.join(d, f), 'rb').read()).hexdigest()[:16]
            for f in sorted(os.listdir(d)) if f.endswith(('.hip', '.h'))}


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f, v in kernel_source_hashes().items():
        h.update(f.encode())
        h.update(v.encode())
    return h.hexdigest()[:16]


# the translation units (and the headers they include) a profiled kernel is compiled from
KERNEL_SOURCES = {'nerf_mlp_fwd_lds_kernel': ('mlp_lds.hip', 'mlp_layout.h', 'common.h'),
                  'gauss_': ('gauss.hip', 'gauss_csr.hip', 'common.h'), 'igsm_': ('gauss.hip', 'common.h'),
                  'seg_': ('gauss_csr.hip', 'common.h')}
PMC_FILE = os.path.join(ROOT, 'profiles', 'r06_pmc_hbm_traffic.json')
PMC_SQ_FILE = os.path.join(ROOT, 'profiles', 'r06_pmc_sq_render.json')


def _kernel_key(name):
    """'void nerfail::k<8, 1, false>' -> 'k<8,1,false>': the form kernel names are compared in."""
    name = name.strip()
    if name.startswith('void '):
        name = name[5:]
    return name.replace('nerfail::', '').replace(' ', '')


def pmc_mfma_busy(kernel):
    """Matrix-pipe utilisation of a kernel from the stored SQ counter pass (tools/r06_pmc_sq.sh: SQ_VALU_MFMA_BUSY_CYCLES over the
    SIMD-cycles of the dispatch, and the clock the chip held) - reported only while the kernel's sources are unchanged."""
    try:
        pmc = json.load(open(PMC_SQ_FILE))
    except (OSError, ValueError):
        return None
    now, then = kernel_source_hashes(), pmc.get('csrc_files', {})
    files = next((v for k, v in KERNEL_SOURCES.items() if k in kernel), tuple(now))
    if any(now.get(f) != then.get(f) for f in files):
        return None
    want = _kernel_key(kernel)
    for name, v in pmc.get('kernels', {}).items():
        if _kernel_key(name) == want and 'mfma_busy' in v:
            return {'mfma_busy': v['mfma_busy'], 'clock_GHz': v.get('clock_GHz'), 'file': 'profiles/' + os.path.basename(PMC_SQ_FILE)}
    return None


def pmc_traffic(kernel_substr, which='avg'):
    """HBM(+Infinity Cache) bytes per launch of a kernel from the separate rocprofv3 --pmc passes (FETCH_SIZE x2 +
    WRITE_SIZE, MI355X_MICROARCH.md; tools/r03_prof_b.sh + tools/pmc_summary.py write the file). PMC counters cannot be
    read from inside this process, so the number is a stored measurement: it is reported ONLY if the file was taken from
    the very sources this run compiles that kernel from (per-file hashes) - otherwise null, with the reason. which = 'avg'
    (mean over the profiled command's launches) or 'max' (its largest launch: for kernels that also run at smaller sizes)."""
    try:
        pmc = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None, 'no PMC file (%s)' % os.path.basename(PMC_FILE)
    now, then = kernel_source_hashes(), pmc.get('csrc_files', {})
    files = next((v for k, v in KERNEL_SOURCES.items() if k in kernel_substr), tuple(now))
    changed = [f for f in files if now.get(f) != then.get(f)]
    if changed:
        return None, 'stale: %s was measured before %s changed' % (os.path.basename(PMC_FILE), ', '.join(changed))
    want = _kernel_key(kernel_substr)
    for name, v in pmc.get('kernels', {}).items():
        if _kernel_key(name) == want:                        # the exact kernel, template arguments included
            f = v['fetch_bytes_per_launch_corrected' if which == 'avg' else 'fetch_bytes_max_launch_corrected']
            w = v['write_bytes_per_launch' if which == 'avg' else 'write_bytes_max_launch']
            if f is None or w is None:
                return None, 'kernel missing from one of the two counter passes'
            return f + w, {'file': 'profiles/' + os.path.basename(PMC_FILE), 'command': pmc.get('command'),
                           'sources': {x: now[x] for x in files}, 'launch': which}
    return None, 'kernel not in ' + os.path.basename(PMC_FILE)


def make_net(seed, dev):
    from nerfail_amd.run_nerf_helpers import NeRF
    sd = synth.nerf_state_dict(D=NET_D, W=NET_W, seed=seed)
    m = NeRF(D=NET_D, W=NET_W, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd, m.to(dev)


def cpu_baseline(seconds_budget=20.0):
    """The numpy oracle (CPU port of the reference path) on rays 320000:320000+n of pose 0, n sized to ~20 s."""
    from oracle import nerf as O
    sc, sf = synth.nerf_state_dict(seed=21), synth.nerf_state_dict(seed=22)
    focal, K = synth.lego_intrinsics(H, W)
    c2w = synth.pose_spherical(-180., -30., 4.)[:3, :4]
    ro, rd = O.get_rays(H, W, K, c2w)
    rays = O.pack_rays(ro, rd, 2., 6.)
    O.render_rays(rays[320000:320128], sc, N_SAMPLES, N_IMPORTANCE, sf, white_bkgd=True)      # warm-up (BLAS threads)
    t = time.time()
    O.render_rays(rays[320128:320384], sc, N_SAMPLES, N_IMPORTANCE, sf, white_bkgd=True)
    per_ray = (time.time() - t) / 256
    n = int(max(256, min(32768, seconds_budget / per_ray)) // 256 * 256)
    t = time.time()
    for s in range(0, n, 1024):
        O.render_rays(rays[320000 + s:320000 + min(n, s + 1024)], sc, N_SAMPLES, N_IMPORTANCE, sf, white_bkgd=True)
    dt = time.time() - t
    return {'value': n / dt, 'unit': 'rays/s', 'cores': N_CPU, 'kind': 'port',
            'sample': '%d rays (pixels 320000..) of the same 800x800 view, 64+128 samples, D=8 W=256, numpy oracle '
                      '(oracle/nerf.py, OpenBLAS sgemm on the %d CPUs of the cgroup quota; the host reports %d), %.1f s'
                      % (n, N_CPU, os.cpu_count() or 0, dt)}


def cpu_baseline_fwd_bwd(seconds_budget=15.0):
    """The training step (RN:776-791: render with perturb = 1, the two MSE terms, loss.backward()) as a PyTorch-CPU fp32
    port (oracle/torch_port.py, pinned to the reference's own gradients by fixture g7) on a bounded ray sample of the
    1024-ray step, D=8 W=256, 64+128 samples, torch.set_num_threads(cores)."""
    from oracle import torch_port as TP
    torch.set_num_threads(N_CPU)
    sc, sf = synth.nerf_state_dict(seed=31), synth.nerf_state_dict(seed=32)
    rs = np.random.RandomState(0)

    def run(n, seed):
        rays = synth.ray_batch(n, seed=seed)
        target = rs.uniform(size=(n, 3)).astype(np.float32)
        t_rand, u = rs.uniform(size=(n, N_SAMPLES)).astype(np.float32), rs.uniform(size=(n, N_IMPORTANCE)).astype(np.float32)
        t = time.time()
        TP.train_step(rays, sc, sf, target, t_rand, u, N_SAMPLES, N_IMPORTANCE, NET_D)
        return time.time() - t
    run(64, 1)                                              # warm-up (thread pools, allocator)
    per_ray = run(128, 2) / 128
    n = int(max(128, min(2048, seconds_budget / per_ray)) // 128 * 128)
    dt = 0.0
    for s_ in range(0, n, 1024):                            # whole 1024-ray steps (the reference's batch), then the rest
        dt += run(min(1024, n - s_), 3 + s_)
    return {'value': n / dt, 'unit': 'rays/s (fwd+bwd)', 'cores': N_CPU, 'kind': 'port',
            'sample': '%d rays in steps of <= 1024 (64+128 samples, D=8 W=256, perturb=1, coarse+fine MSE, loss.backward(); no '
                      'optimizer), PyTorch-CPU fp32 port oracle/torch_port.py, torch.set_num_threads(%d), %.1f s' % (n, N_CPU, dt)}


def cpu_baseline_attack(seconds_budget=20.0):
    """The numpy oracle's gauss path of ONE NeRFail-S iteration (oracle/gauss.py: gauss_forward + gauss_backward for each view
    of the batch, then igsm_step on the [3,800,800,4] perturbation) at full size, on a bounded number of the 8 views."""
    from oracle import gauss as OG
    rs = np.random.RandomState(0)
    P = 3
    Ns = P * H * W
    s = np.zeros((P, H, W, 4), np.float32)
    s[..., 3] = synth.disc_alpha_image(P, H, W, seed=200)[..., 3]
    G = rs.normal(size=(1, H, W, 4)).astype(np.float32)

    def one_view(seed):
        # (index / weight maps with the statistics of a K8+K9-built map are not needed for a CPU time: random neighbours)
        idx = rs.randint(0, Ns, (1, H, W, 8)).astype(np.float32)
        w = rs.uniform(size=(1, H, W, 8)).astype(np.float32)
        w /= w.sum(-1, keepdims=True)
        wi = np.stack([w, idx], 1)
        ori = synth.disc_alpha_image(1, H, W, seed=seed)
        t = time.time()
        OG.gauss_forward(s, wi, ori, None)
        g = OG.gauss_backward(s, wi, ori, np.zeros_like(G), G, None)
        return time.time() - t, g
    t1, g = one_view(1)
    n_views = int(max(1, min(8, (seconds_budget - t1) / t1)))
    t_views = t1
    for v in range(1, n_views):
        t_views += one_view(1 + v)[0]
    t = time.time()
    OG.igsm_step(s, g, s, 2.0, 32.0, False)
    t_step = time.time() - t
    per_iter = t_views / n_views * 8 + t_step
    return {'value': 1.0 / per_iter, 'unit': 'iterations/s (gauss path, batch of 8 views)', 'cores': 1, 'kind': 'port',
            'sample': '%d of the 8 views of one iteration at 800x800, P=3 (oracle/gauss.py gauss_forward + gauss_backward per view: '
                      '%.2f s per view, single-threaded numpy gathers / np.add.at) + igsm_step %.2f s; extrapolated to 8 views'
                      % (n_views, t_views / n_views, t_step)}


def train_bench(dev, steps=10, warmup=2, n_rand=1024, precision='f32'):
    """NeRF training step (RN:776-801) at the shipped config (configs/lego.txt: N_rand=1024, 64+128 samples,
    D=8 W=256, perturb=1, white_bkgd): render -> mse(rgb)+mse(rgb0) -> backward -> Adam. rays/s (fwd+bwd)."""
    from nerfail_amd import run_nerf as RN
    from nerfail_amd.run_nerf import ray_gen
    _, coarse = make_net(31, dev)
    _, fine = make_net(32, dev)
    coarse.precision = fine.precision = precision
    params = list(coarse.parameters()) + list(fine.parameters())
    for p in params:
        p.requires_grad_(True)
    from nerfail_amd.optim import Adam
    opt = Adam(params, lr=5e-4, betas=(0.9, 0.999))              # RN:207 on the fused K13 kernel
    focal, K = synth.lego_intrinsics(H, W)
    c2w = synth.pose_spherical(-180., -30., 4.)[:3, :4]
    gen = torch.Generator(device=dev).manual_seed(0)
    host_rng = np.random.default_rng(0)

    def step():
        # RN:752: get_rays of the step's view on the FULL image, inside the step as in the reference's loop (round 6, VERDICT r5
        # item 2: until then the rays came from an all_rays tensor made before the timed region) - one ray_gen launch, 28 MB
        all_rays = ray_gen(H, W, K, c2w, 2., 6.)
        # RN:768 draws the batch with the LEGACY np.random.choice(H*W, N_rand, replace=False): a 640 000-element host
        # permutation (6-15 ms, longer than the whole GPU step). Same distribution from numpy's Generator.choice (Floyd's
        # algorithm: 27 us on the host, which runs ahead of the GPU anyway); the 4 KB of indices go up asynchronously.
        # (Round 2 drew them with torch.randperm on the device: a 640 000-key sort, ~0.2 ms of GPU time per step.)
        sel = torch.from_numpy(host_rng.choice(H * W, n_rand, replace=False)).pin_memory().to(dev, non_blocking=True)
        rays = all_rays[sel].contiguous()
        target = torch.rand((n_rand, 3), device=dev, generator=gen)
        t_rand = torch.rand((n_rand, N_SAMPLES), device=dev, generator=gen)
        u = torch.rand((n_rand, N_IMPORTANCE), device=dev, generator=gen)
        r = RN.render_rays(rays, coarse, None, N_SAMPLES, N_importance=N_IMPORTANCE, network_fine=fine, white_bkgd=True,
                           perturb=1., t_rand=t_rand, u=u)
        loss = RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    # warm-up: at least `warmup` steps AND one second, and then UNTIL THE STEP TIME IS STEADY (5 consecutive steps within
    # 1.3x of their fastest; at most 10 s). Right behind another heavy GPU section (or process) every step of this short
    # section has been seen to take 60-115 ms instead of 8 for one to two seconds - every kernel at its normal duration,
    # the launch thread simply not running (CFS throttling of the container's CPU quota); a single render view is longer
    # than such an episode, this section is not, so it waits the episode out instead of timing it.
    t_w = time.time()
    n_w, recent = 0, []
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        n_w += 1
        recent = (recent + [e0.elapsed_time(e1)])[-5:]
        steady = len(recent) == 5 and max(recent) < 1.3 * min(recent)
        if n_w >= warmup and time.time() - t_w >= 1.0 and (steady or time.time() - t_w > 10.0):
            break
    warmup_info = {'steps': n_w, 'seconds': time.time() - t_w, 'steady': bool(steady)}

    def timed_block():
        t = time.time()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            loss_ = step()
            marks[i + 1].record()
            if i >= 1:
                marks[i].synchronize()      # the host stays at most one step ahead of the GPU (see the note below)
        torch.cuda.synchronize()
        return (time.time() - t) / steps, [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)], loss_
    # A host-side episode (see the warm-up note) can be perfectly STEADY while it lasts - round 3 saw 17 consecutive steps of
    # 77.5 ms (GPU kernels: 7 ms) right behind the render section, which the steadiness test above accepts. The timed block is
    # therefore repeated (at most 4 times) while its steps disagree by more than 1.5x or a later block is faster; every block
    # is reported, the section's rate is the block with the lowest median.
    attempts = []
    for _ in range(4):
        attempts.append(timed_block())
        ps = attempts[-1][1]
        if max(ps) <= 1.5 * min(ps) and (len(attempts) == 1 or float(np.median(ps)) >= 0.9 * min(float(np.median(a[1])) for a in attempts[:-1])):
            break
    best = min(attempts, key=lambda a: float(np.median(a[1])))
    dt, per_step, loss = best
    warmup_info['timed_blocks_median_ms'] = [round(float(np.median(a[1])), 3) for a in attempts]
    evals = n_rand * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE)                           # coarse + fine network evaluations
    flop = evals * FLOP_PER_SAMPLE * 3                                                # fwd + bwd-data + bwd-weights
    # Before the CPU pools were sized to the cgroup quota (top of this file) single steps sporadically took 30-110 ms:
    # CFS throttling of the launch thread. The section's rate is the MEDIAN step; the whole-loop mean and every step's
    # time are reported next to it so that such an episode stays visible.
    mean_dt = dt
    dt = float(np.median(per_step)) * 1e-3
    out = {'train_rays_per_sec_fwd_bwd': n_rand / dt, 'ms_per_step': dt * 1e3, 'rays_per_step': n_rand, 'precision': precision,
           'final_loss': float(loss.detach()), 'fp32_equivalent_tflops_whole_step': flop / dt / 1e12,
           'statistic': 'median of the %d timed steps' % steps, 'ms_per_step_mean_whole_loop': mean_dt * 1e3, 'warmup': warmup_info,
           'note': 'the whole loop body RN:752-801: get_rays of the full 800x800 image (one ray_gen launch) -> batch of 1024 pixels '
                   '(drawn on the host like RN:768, but with numpy Generator.choice (27 us) instead of the legacy np.random.choice, '
                   'a 6-15 ms permutation) -> render -> loss -> backward -> Adam',
           'ms_per_step_each': [round(v, 3) for v in per_step]}
    if precision == 'f32':      # the three GEMM families run on the exact-f32 MFMA: that pipe bounds the step
        out['roofline'] = {'bound': 'mfma', 'achieved': flop / dt / 1e12, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': flop / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                           'note': 'whole step incl. sampling, compositing, Adam and host launch gaps; 3x forward FLOPs'}
    else:                       # split precision: the saved activations / gradients (HBM) bound the step, not the MFMAs
        # per network evaluation (D=8 W=256): forward writes 79 activation slots of 128 B, backward-data
        # writes 77 gradient slots, the weight-gradient pass reads 82 + 89 slots (some operands serve two layers)
        nbytes = evals * 128 * (79 + 77 + 82 + 89)
        out['roofline'] = {'bound': 'hbm', 'achieved': nbytes / dt / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': nbytes / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                           'note': 'whole step; algorithmic bytes of the saved activations and layer gradients only'}
    return out


def render_f16x3_bench(dev, steps=2):
    """The same 800x800 render with the opt-in split-precision MLP kernel (NeRF.precision = 'f16x3': every product as
    a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the fp16 matrix cores, fp32 accumulation; parity-tested at the same 1e-4 bound).
    Reports rays/s and the max deviation of the rendered maps from the exact-f32 kernel on the same view."""
    from nerfail_amd import nerf_to_coord as NC
    nets = {}
    for prec in ('f32', 'f16x3'):
        _, c = make_net(21, dev)
        _, f = make_net(22, dev)
        c.precision = f.precision = prec
        nets[prec] = (c, f)
    focal, K = synth.lego_intrinsics(H, W)
    c2w = torch.from_numpy(synth.pose_spherical(-117., -30., 4.)[:3, :4])

    def run(prec):
        c, f = nets[prec]
        kw = dict(network_query_fn=None, perturb=0., N_importance=N_IMPORTANCE, network_fine=f, N_samples=N_SAMPLES,
                  network_fn=c, use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)
        with torch.no_grad():
            return NC.render(H, W, K, chunk=H * W, c2w=c2w, near=2., far=6., **kw)
    ref = run('f32')
    out = run('f16x3')
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(steps):
        out = run('f16x3')
    torch.cuda.synchronize()
    dt = (time.time() - t) / steps
    dev_rgb = float((out[0] - ref[0]).abs().max())
    dev_acc = float((out[2] - ref[2]).abs().max())
    samples = H * W * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE)
    return {'rays_per_sec': H * W / dt, 'ms_per_view': dt * 1e3, 'speedup_vs_f32_kernel_this_run': None,
            'fp32_equivalent_tflops_whole_step': samples * FLOP_PER_SAMPLE / dt / 1e12,
            'max_abs_dev_rgb_vs_f32_kernel': dev_rgb, 'max_abs_dev_acc_vs_f32_kernel': dev_acc,
            'note': 'opt-in mode; the headline value above is the exact-f32 kernel'}


def knn_bench(dev, reps=2):
    """8-NN index build of ONE view (create_index_and_dist.py:126-145): 640 000 queries (the view's pts_max) against the
    1 920 000-point set of 3 base views. Exact (d2, index) ordering. Two geometries: the synthetic shell points of
    SURVEY.md section 8d, and the analytic pts_max of a rendered sphere (40 % surface hits, 60 % background pixels on
    the near plane, far from every set point). The set's grid is built once per scene (CI:57-61) and searched per view:
    `ms_per_view` is the per-view search, `grid_build_ms` the one-off build.
    Roofline: the search is neither HBM- nor MFMA-bound; what it spends is distance evaluations (3 sub, 3 mul, 2 add,
    no FMA - the bit-exact definition forbids contraction) and the dependent loads that feed them. Reported against the
    non-FMA vector rate (157.3 / 2 TFLOP/s -> 9.8e12 evaluations/s) with the number of candidates examined per query from
    the kernel's own counters (nerfail_knn8_grid_stats); the brute-force scan would examine 1 920 000 per query."""
    from nerfail_amd import _lib, create_index_and_dist as CID
    from nerfail_amd.create_index_and_dist import index_and_dist
    lib = _lib.load()
    out = {}
    geo = {'shell_points': (synth.sphere_shell_points(3 * H * W, seed=0), synth.sphere_shell_points(H * W, seed=1).reshape(H, W, 3)),
           'rendered_view_geometry': (np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3),
                                      synth.sphere_view_points(H, W, 45.).reshape(H, W, 3))}
    for name, (S_, Q_) in geo.items():
        S, Q = torch.from_numpy(np.ascontiguousarray(S_, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(Q_, np.float32)).to(dev)
        CID._GRID.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        CID._grid_for(S)                                         # the build alone
        e1.record()
        torch.cuda.synchronize()
        build_ms = e0.elapsed_time(e1)
        index_and_dist(Q, S)
        blocks = []
        for _ in range(3):                                       # fastest of 3 blocks of `reps` searches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                index_and_dist(Q, S)
            e1.record()
            torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) / reps * 1e-3)
        dt = min(blocks)
        stats = torch.zeros((2,), dtype=torch.int64, device=dev)
        lib.nerfail_knn8_grid_stats(_lib.dev(stats))
        try:                                                     # never leave the library counting into a tensor that may be freed
            index_and_dist(Q, S)
            torch.cuda.synchronize()
        finally:
            lib.nerfail_knn8_grid_stats(None)
        cand, nfar = [int(v) for v in stats.cpu().tolist()]
        out[name] = {'views_per_sec': 1.0 / dt, 'ms_per_view': dt * 1e3, 'queries_per_sec': H * W / dt, 'grid_build_ms': build_ms,
                     'candidates_examined_per_query': cand / float(H * W), 'far_search_queries': nfar,
                     'brute_force_equivalent_pairs_per_sec': float(H * W) * float(3 * H * W) / dt,
                     'candidate_evaluations_per_sec': cand / dt,
                     'compulsory_hbm_bytes_per_view': 71.7e6, 'compulsory_hbm_GBps_at_this_rate': 71.7e6 / dt / 1e9,
                     # a pruned search has no single bounding roofline (instruction issue + the slowest tiles decide):
                     # the work counters above are the evidence, not a fraction of a brute-force ALU peak
                     'roofline': None}
    out.update(out['shell_points'])                          # round-1 keys keep their meaning (shell points)
    return out


def victim_cnn(num_classes=8):
    """Stand-in victim with the shape of the reference's 800x800 classifier (model/MyModel.py:5-52: seven
    3x3 conv + ReLU + 2x2 max-pool stages 3-32-64-128-256-256-128-64, then 1024-512-classes). Stock PyTorch
    (MIOpen) - the classifier is outside the hot path."""
    chans = [3, 32, 64, 128, 256, 256, 128, 64]
    layers = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        layers += [torch.nn.Conv2d(cin, cout, 3), torch.nn.ReLU(), torch.nn.MaxPool2d(2)]
    layers += [torch.nn.Flatten(), torch.nn.Linear(1024, 512), torch.nn.ReLU(), torch.nn.Linear(512, num_classes)]
    return torch.nn.Sequential(*layers)


class _Emitting(dict):
    """dict that reports every top-level assignment through `emit` as it happens: what a section measured before a later
    leg faulted is already in the child's result file."""

    def __init__(self, emit):
        super().__init__()
        self._emit = emit

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        self._emit({k: v})


def attack_bench(dev, iters=5, out=None):
    """NeRFail-S iteration (AS:304-392) on one batch of 8 views, 800x800, P=3 base views.
    (i) gauss path only: K10 fwd + K11 bwd (deterministic inverted-index form) + K12, the classifier replaced by a
        fixed upstream gradient; against the 1.60 GB/iteration HBM roofline of SURVEY.md section 8(d).
    (ii) end to end through gauss_net.forward with the stand-in victim CNN (2 classifier forwards + 1 backward per
        step, as the reference does)."""
    from nerfail_amd.GaussNet import gauss_gather, create_gauss_w, gauss_net
    from nerfail_amd.attack import igsm_step, nerfail_s_step
    rs = np.random.RandomState(0)
    P, B = 3, 8
    wi, ori, s_init = _attack_inputs(dev, B, seed=0)          # maps built by K8 + K9 on the analytic view geometry
    G = torch.from_numpy(rs.normal(size=(B, H, W, 4)).astype(np.float32)).to(dev)
    out = out if out is not None else {}

    def timed(fn, s, blocks=5):
        # median of `blocks` timed blocks of `iters` iterations each: these sections last milliseconds, and a host-side
        # stall (see train_bench's warm-up note) inside a single short block would otherwise be the reported number
        s = fn(s)                       # warm-up (builds the inverted index / MIOpen plans once)
        torch.cuda.synchronize()
        per_block = []
        for _ in range(blocks):
            t = time.time()
            for _ in range(iters):
                s = fn(s)
            torch.cuda.synchronize()
            per_block.append((time.time() - t) / iters)
        timed.blocks_ms = [round(v * 1e3, 4) for v in per_block]
        return float(np.median(per_block))

    from nerfail_amd.GaussNet import resolve_views, hot_forward, hot_backward_rgb, register_view, _VIEW_MAPS, _VIEW_ORI, _VIEW_CACHE
    from nerfail_amd.attack import igsm_step_rgb
    alg_bytes = 8 * (102.4e6 + 81.9e6) + 122.9e6        # 1.60 GB / iteration
    Ns = s_init.numel() // 4
    ori_u8 = ori.to(torch.uint8)                        # what cv2.imread hands the reference's dataset (MyDataset.py:200)

    def leg(dt):
        return {'iters_per_sec': 1.0 / dt, 'ms_per_iter': dt * 1e3, 'statistic': 'median of 5 blocks of %d iterations' % iters,
                'ms_per_iter_each_block': timed.blocks_ms,
                'roofline': {'bound': 'hbm', 'achieved': alg_bytes / dt / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': alg_bytes / dt / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                             'note': 'algorithmic 1.60 GB per iteration as SURVEY 8(d) counts it (fp32 x and ori, all four '
                                     'gradient channels); this form moves fewer bytes, see attack.gauss_kernels'}}

    # (1) the path nerfail_s_step takes (round 3): forward without the x tensor, uint8 images, alpha + 3-bit mask handed to an
    # rgb-gradient-only backward that writes [Ns,3], sign step on [Ns,3] (AS:357-392 reads grad[..., :3] only)
    views = resolve_views(s_init, wi, ori_u8)
    buf = torch.empty((3 * Ns + 1,), device=dev)

    def one_iter_rgb(s):
        _, x_rgba, aux = hot_forward(s, views, None, None, need_x=False, need_aux=True)
        hot_backward_rgb(aux, G, views, buf)
        return igsm_step_rgb(s, buf, s_init, 2.0, 32.0, False)
    out['gauss_path_deterministic'] = leg(timed(one_iter_rgb, s_init.clone()))
    out['gauss_path_deterministic']['form'] = ('rgb-gradient-only step path with the gradient materialised (attack.nerfail_s_step on more '
                                               'than one rank: the [Ns,3] buffer is what the all-reduce moves): K10 no-x/uint8-ori/aux, K11 rgb, K12 rgb')

    # (1b) round 6: the path nerfail_s_step takes on ONE rank - the sign step is the epilogue of K11's last launch
    # (nerfail_gauss_bwd_views_rgb_step): no [Ns,3] gradient written and read back, one launch less; bit-identical iterates
    from nerfail_amd.GaussNet import hot_backward_rgb_step

    def one_iter_fused(s):
        _, x_rgba, aux = hot_forward(s, views, None, None, need_x=False, need_aux=True)
        return hot_backward_rgb_step(aux, G, views, s, s_init, 2.0, 32.0, False)
    out['gauss_path_fused_step'] = leg(timed(one_iter_fused, s_init.clone()))
    out['gauss_path_fused_step']['form'] = 'attack.nerfail_s_step on one rank: K10, then K11 with the sign step K12 as its epilogue'

    # (2) the full autograd form (all four gradient channels, x materialised): what gauss_net.forward + loss.backward() run
    for det in (True, False):
        def one_iter(s, det=det):
            st = s.detach().requires_grad_(True)
            x, x_rgba = gauss_gather(st, wi, ori, None, None, det)
            x_rgba.backward(G)
            return igsm_step(st.detach(), st.grad, s_init, 2.0, 32.0, False)
        out['gauss_path_full_gradient_' + ('deterministic' if det else 'atomics')] = leg(timed(one_iter, s_init.clone()))

    # (3) VERDICT r2 item 2 - the loop as the reference feeds it: a Dataset over the files on disk (index_and_weight/<i>.pth,
    # <i>.png) behind a DataLoader(batch_size=8, num_workers=0) that is iterated every step (MyDataset.py:187-204, AS:222-231,
    # AS:304-317). nerfail_amd.MyDataset.gauss_dataset reads a view ONCE and keeps it on the device by view id; (a) with its
    # collate_views the batch is a list of those resident tensors (nothing copied), (b) with torch's default collate the eight
    # resident maps are stacked into a fresh 328 MB device tensor per iteration, as the reference's loader does,
    # (c) the reference's own behaviour - torch.load + imread of every view in every iteration - for comparison.
    import shutil
    import tempfile
    from PIL import Image
    from nerfail_amd.MyDataset import gauss_dataset
    light = os.environ.get('NERFAIL_BENCH_LIGHT', '0') == '1'      # counter passes: kernels only, no host-bound legs
    tmp = tempfile.mkdtemp(prefix='nf_bench_ds_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        if light:
            raise StopIteration
        maps, pngs = [], []
        for b in range(B):
            maps.append(os.path.join(tmp, '%d.pth' % b))
            pngs.append(os.path.join(tmp, '%d.png' % b))
            torch.save(wi[b].cpu(), maps[-1])
            Image.fromarray(ori_u8[b].cpu().numpy()[..., [2, 1, 0, 3]], 'RGBA').save(pngs[-1])    # BGRA tensor -> RGBA file
        names = [''] * B
        for tag, resident, own_collate in (('resident_collate_views', True, True), ('resident_default_collate', True, False),
                                           ('reload_every_iteration', False, False)):
            _VIEW_MAPS.clear(); _VIEW_ORI.clear(); _VIEW_CACHE.clear()
            ds = gauss_dataset(maps, pngs, names, names, dev, Ns=Ns if resident else None)
            loader = torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False, num_workers=0,
                                                 collate_fn=ds.collate_views if own_collate else None)

            def one_iter_host(s):
                for idx, ori_b, wi_b, _, _ in loader:             # one batch = all 8 views
                    v = resolve_views(s, wi_b, ori_b)
                    _, x_rgba, aux = hot_forward(s, v, None, None, need_x=False, need_aux=True)
                    hot_backward_rgb(aux, G, v, buf)
                    s = igsm_step_rgb(s, buf, s_init, 2.0, 32.0, False)
                return s
            dt = timed(one_iter_host, s_init.clone(), blocks=3)
            out['gauss_path_host_dataloader_' + tag] = leg(dt)
            out['gauss_path_host_dataloader_' + tag]['ratio_to_resident_path'] = dt * 1e3 / out['gauss_path_deterministic']['ms_per_iter']
        out['gauss_path_host_dataloader_note'] = (
            'DataLoader(num_workers=0) over index_and_weight/<i>.pth + <i>.png iterated every step. resident_collate_views: '
            'nerfail_amd.MyDataset.gauss_dataset keeps each view on the device by id, the batch is a list of resident tensors; '
            'resident_default_collate: same dataset, torch default_collate stacks 328 MB on the device per step; '
            'reload_every_iteration: the reference behaviour (torch.load + imread per view and step, files in /dev/shm), '
            'maps fingerprinted to find their cached inverted index')
    except StopIteration:
        pass
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        _VIEW_MAPS.clear(); _VIEW_ORI.clear()
    out['gauss_kernels'] = gk = gauss_kernel_rooflines(dev, wi, ori, s_init, G)
    # the same iteration against the bytes THIS form has to move (8 views: K10 + K11, once K12), not SURVEY's nominal 1.60 GB
    comp = sum(gk[k]['compulsory_bytes_per_call'] for k in ('K10_gauss_fwd', 'K11_gauss_bwd_views', 'K12_igsm_step'))
    rl = dict(out['gauss_path_deterministic']['roofline'])
    rl['compulsory_bytes_per_iter'] = comp
    rl['frac_compulsory'] = comp / (out['gauss_path_deterministic']['ms_per_iter'] * 1e-3) / 1e9 / HBM_PEAK_GBS
    out['gauss_path_deterministic'] = dict(out['gauss_path_deterministic'], roofline=rl)
    # the fused form: K11 writes no [Ns,3] gradient, the step reads s and s_init and writes s' (K12's bytes without the gradient read)
    comp_f = comp - Ns * 12 - Ns * 12
    rl = dict(out['gauss_path_fused_step']['roofline'])
    rl['compulsory_bytes_per_iter'] = comp_f
    rl['frac_compulsory'] = comp_f / (out['gauss_path_fused_step']['ms_per_iter'] * 1e-3) / 1e9 / HBM_PEAK_GBS
    out['gauss_path_fused_step'] = dict(out['gauss_path_fused_step'], roofline=rl)
    if light:
        out['batch_views'] = B
        return out

    # end to end. The victim runs with MIOpen allowed to pick its solvers (torch.backends.cudnn.benchmark) and channels-last
    # weights: without that MIOpen falls back to naive_conv_* kernels for some of these 800x800 layers (17 % of the profiled
    # GPU time in round 2). Identical arithmetic; the untuned configuration is reported as the secondary line.
    torch.manual_seed(0)
    victim = victim_cnn(8).to(dev)
    victim.requires_grad_(False)                   # the attack differentiates w.r.t. the perturbation only
    label = torch.tensor(4, device=dev)
    # Round 4: the solver search is OPT-IN. torch.backends.cudnn.benchmark makes MIOpen RUN every candidate solver of every
    # layer; under the guard-page allocator (tests/guard) one of its candidates, igemm_bwd_gtcx35_nhwc_fp32_*, reads past the
    # end of a tensor - harmless while something is mapped behind it, a "Memory access fault by GPU" at the first byte past an
    # allocator segment when nothing is (BENCH_r03). The victim CNN is outside the hot path.
    tuned = os.environ.get('NERFAIL_BENCH_TUNE_VICTIM', '0') == '1'
    victim.train(False)                            # AS:281-282, as every attack loop of the reference does
    net_u = gauss_net(dev, 0.02, victim, 'my_model', epsilon=None)
    net_u.cache_ori_cla = False                    # this leg: the reference's shape, original-image logits recomputed per step
    dt = timed(lambda s: nerfail_s_step(net_u, s, s_init, wi, ori_u8, label, 2.0, 32.0, False)[0], s_init.clone())
    untuned = {'iters_per_sec': 1.0 / dt, 'ms_per_iter': dt * 1e3,
               'note': 'default MIOpen solver choice, NCHW victim: gauss_net forward (2 classifier forwards) + CE + backward + sign step'}
    prev = torch.backends.cudnn.benchmark
    if tuned:
        t_tune = time.time()
        torch.backends.cudnn.benchmark = True
        victim_t = victim_cnn(8).to(dev).to(memory_format=torch.channels_last).requires_grad_(False)
        victim_t.load_state_dict(victim.state_dict())
        victim_t.train(False)
        net = gauss_net(dev, 0.02, victim_t, 'my_model', epsilon=None)
        net.cache_ori_cla = False
        dt = timed(lambda s: nerfail_s_step(net, s, s_init, wi, ori_u8, label, 2.0, 32.0, False)[0], s_init.clone())
        out['end_to_end_victim_cnn'] = {
            'iters_per_sec': 1.0 / dt, 'ms_per_iter': dt * 1e3, 'solver_search_seconds': time.time() - t_tune - dt * (iters + 1),
            'note': 'torch.backends.cudnn.benchmark = True + channels_last victim (solver selection only; same arithmetic): '
                    'gauss_net forward (2 classifier forwards) + CE + backward + sign step'}
        out['end_to_end_victim_cnn_untuned_miopen'] = untuned
    else:
        net = net_u
        out['end_to_end_victim_cnn'] = untuned
    # Round 6: the DEFAULT configuration (cache_ori_cla = None) with the views named by id, as INTEGRATION.md
    # section 1 writes the loop: the logits of the unperturbed images are kept per set of view ids while the classifier is frozen
    # and in eval() mode - one classifier forward per step instead of two, identical results.
    net.cache_ori_cla = None
    ids = [('bench-e2e', b) for b in range(B)]
    dt = timed(lambda s: nerfail_s_step(net, s, s_init, wi, ori_u8, label, 2.0, 32.0, False, view_ids=ids)[0], s_init.clone())
    out['end_to_end_victim_cnn_cached_original_logits'] = {
        'iters_per_sec': 1.0 / dt, 'ms_per_iter': dt * 1e3,
        'note': 'default settings, views named by id (INTEGRATION.md section 1): the original images\' logits are computed once '
                '(1 classifier forward per step; identical results)'}
    # NeRFail's per-view inner loop (deepfool.py:44-107): one view, 8 classes, untargeted (7 competing classes per
    # iteration); m1 is set so that the loop never stops early
    from nerfail_amd.deepfool import deepfool
    net.cache_ori_cla = False
    n_it = 6
    deepfool((s_init, wi[:1], ori[:1]), 1.0, net, num_classes=8, max_iter=2, m1=1e6, m2=30)          # warm-up
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):                  # median of 3 runs of the 6-iteration loop (same reason as `timed`)
        t = time.time()
        _, loop_i, _, _, _ = deepfool((s_init, wi[:1], ori[:1]), 1.0, net, num_classes=8, max_iter=n_it, m1=1e6, m2=30)
        torch.cuda.synchronize()
        runs.append((time.time() - t) / max(loop_i, 1))
    dt = float(np.median(runs))
    out['deepfool_inner_loop'] = {'iters_per_sec': 1.0 / dt, 'ms_per_iter': dt * 1e3,
                                  'note': 'one 800x800 view, 8 class gradients per iteration: victim CNN fwd + 8 bwd (stock '
                                          'PyTorch) + one multi-RHS pass over the inverted index'}
    torch.backends.cudnn.benchmark = prev
    out['batch_views'] = B
    out['unit'] = 'NeRFail-S iterations/s (batch of 8 views, 800x800, P=3)'
    return out


def gauss_kernel_rooflines(dev, wi, ori, s_init, G, n=10):
    """K10 / K11 / K12 of the step path (rgb-gradient form) one by one through the C-ABI, HIP events on the launch stream.
    Bytes: `compulsory_bytes_per_call` = what THIS form has to move once (counted from the maps: background pixels read
    their 32 bytes of weights only; the index stream of K11 is part of it), `survey_bytes_per_call` = SURVEY.md section 8d's
    nominal figure (fp32 x and ori, four gradient channels) for reference. roofline.achieved / frac = compulsory bytes / time
    (round 4; algorithmic bytes, as for every other roofline of this file); roofline.traffic = the kernels' PMC bytes when
    the stored counters belong to these sources; K12's tables (3 x 30.7 MB) would sit in the 256 MB Infinity Cache across
    back-to-back calls, so its calls rotate over 4 table sets (368 MB) and use distinct s / s_init buffers."""
    from nerfail_amd import _lib
    from nerfail_amd.GaussNet import resolve_views, view_table
    lib = _lib.load()
    B, P, Ns = wi.shape[0], H * W, s_init.numel() // 4
    s = s_init.reshape(-1, 4).contiguous()
    ori_u8 = ori.to(torch.uint8)
    views = resolve_views(s_init, wi, ori_u8)
    vtab = views.table()
    xr = torch.empty((B, H, W, 4), device=dev)
    aux_a, aux_m = torch.empty((B, H, W), device=dev), torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    vis = views.indices()                                     # the per-view inverted indices (N2; built once, cached)
    table, floats = view_table(vis)
    scratch = torch.empty((floats,), device=dev)
    g3 = torch.empty((3 * Ns + 1,), device=dev)
    st = _lib.stream()
    # compulsory bytes of this form, from the data
    fg = float((wi[:, 0].abs().sum(-1) > 0).float().mean())                     # pixels with a non-zero weight
    entries = float(sum(vi.n_entries for vi in vis))
    rows = float(sum(vi.n_rows for vi in vis))
    k10 = B * P * (32 + 32 * fg + 4 + 16 + 5) + Ns * 16                         # weights, indices (foreground), u8 ori, x_rgba, aux; table once
    _lib.check(lib.nerfail_gauss_fwd_views(_lib.dev(s), Ns, vtab, B, P, 1, -1.0, None, _lib.dev(xr), _lib.dev(aux_a), _lib.dev(aux_m), None, st))
    passing = float((aux_m != 0).float().mean())
    k11 = (B * P * (1 + 20 * passing + 16) + B * P * 16                         # mask, alpha + G where it passes, g_pix write; g_pix read once
           + entries * 8 + rows * 16 * 2 + B * Ns * 4 + Ns * 12)                # index stream, row sums w + r, pos per view, grad3
    k12 = Ns * (16 + 12 + 16 + 16)
    sets = [(torch.randn((Ns, 4), device=dev), torch.randn((3 * Ns + 1,), device=dev), torch.randn((Ns, 4), device=dev),
             torch.empty((Ns, 4), device=dev)) for _ in range(4)]
    rot = [0]

    def k12_call():
        a_, g_, i_, o_ = sets[rot[0] % 4]
        rot[0] += 1
        return lib.nerfail_igsm_step_rgb(_lib.dev(a_), _lib.dev(g_), _lib.dev(i_), Ns, 2.0, 32.0, 0, _lib.dev(o_), st)
    calls = {
        'K10_gauss_fwd': (lambda: lib.nerfail_gauss_fwd_views(_lib.dev(s), Ns, vtab, B, P, 1, -1.0, None, _lib.dev(xr), _lib.dev(aux_a),
                                                              _lib.dev(aux_m), None, st),
                          k10, B * 102.4e6, ('gauss_fwd_views_kernel<true>',)),
        'K11_gauss_bwd_views': (lambda: lib.nerfail_gauss_bwd_views_rgb(_lib.dev(aux_a), _lib.dev(aux_m), _lib.dev(G), table, B, Ns, P,
                                                                       _lib.dev(scratch), _lib.dev(g3), st),
                                k11, B * 81.9e6, ('gauss_pixel_grad_rgb_kernel', 'gauss_seg_reduce_views_kernel<false>', 'gauss_seg_combine_views_kernel',
                                                  'gauss_rows_sum3_kernel<false>')),
        'K12_igsm_step': (k12_call, k12, 122.9e6, ('igsm_step_rgb_kernel',)),
    }
    out = {}
    for name, (fn, comp, survey, kernels) in calls.items():
        for _ in range(2):
            _lib.check(fn())
        blocks = []
        for _ in range(3):                                    # fastest of 3 blocks of n back-to-back calls (a host stall
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # inside a block leaves
            e0.record()                                                                            # the GPU idle between events)
            for _ in range(n):
                _lib.check(fn())
            e1.record()
            torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) / n)
        ms = min(blocks)
        traffic, src = 0.0, None
        for k in kernels:                                     # PMC bytes per launch (every kernel launches once per call)
            t, src = pmc_traffic(k, 'max')                    # the 8-view batch is these kernels' largest launch
            traffic = None if (t is None or traffic is None) else traffic + t
        rate = comp / ms / 1e6                                # ALGORITHMIC (compulsory) bytes of this form per second
        out[name] = {'ms_per_call': ms, 'statistic': 'fastest of 3 blocks of %d calls' % n, 'kernels': list(kernels),
                     'compulsory_bytes_per_call': comp, 'survey_bytes_per_call': survey,
                     'roofline': {'bound': 'hbm', 'achieved': rate, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': rate / HBM_PEAK_GBS,
                                  'bytes_used': 'compulsory bytes of this form',
                                  'traffic': traffic, 'traffic_source': src,
                                  'traffic_note': 'PMC FETCH_SIZE x2 + WRITE_SIZE per call; the x2 of MI355X_MICROARCH.md is calibrated for wide '
                                                  'coalesced reads, not for the 16-byte gathers of these kernels: an upper estimate',
                                  'survey_nominal_GBps': survey / ms / 1e6}}
    out['inverted_index_bytes_per_view'] = int(sum(vi.nbytes() for vi in vis) / len(vis))
    out['foreground_pixel_fraction'] = fg
    out['gradient_passing_pixel_fraction'] = passing
    return out


def _attack_inputs(dev, n_views, seed=0):
    """Synthetic NeRFail-S inputs at full size. Geometry = the analytic `pts_max` of a rough unit
    sphere (synth.sphere_view_points: hit pixels on the surface, miss pixels on the near plane, ARRAY ORDER = PIXEL ORDER
    as in the real pipeline): the 1.92 M-point set is 3 base views (CI:57-61), each attack view's 640 000 query points
    come from its own pose, and every view's [2,800,800,8] weight/index map is BUILT BY THE PATH ITSELF (K8
    nerfail_knn8_grid -> K9 nerfail_gauss_weight), exactly what create_index_and_dist + dist_to_weight write. Images are
    uint8-valued BGRA with alpha = 255 inside a centred disc (about the sphere's silhouette); zero-init perturbation
    with alpha = base-view alpha (AS:259-263)."""
    from nerfail_amd.create_index_and_dist import index_and_dist
    from nerfail_amd.GaussNet import create_gauss_w
    P = 3
    S = torch.from_numpy(np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3)).to(dev)
    cw = create_gauss_w(dev, 0.02)
    maps = []
    for v in range(n_views):
        Q = torch.from_numpy(synth.sphere_view_points(H, W, -171. + 360. * ((v * 7 + seed) % 40) / 40.)).to(dev)
        maps.append(cw(index_and_dist(Q, S).unsqueeze(0))[0][0])
    wi = torch.stack(maps)
    ori = torch.from_numpy(synth.disc_alpha_image(n_views, H, W, seed=seed + 100)).to(dev)
    s_init = torch.zeros((P, H, W, 4), device=dev)
    s_init[..., 3] = torch.from_numpy(synth.disc_alpha_image(P, H, W, seed=seed + 200)[..., 3]).to(dev)
    return wi, ori, s_init


def cfg3_bench(dev, iters=20, n_views=16, batch=8):
    """BASELINE.json configs[2]: the 20-iteration IGSM loop of attack_NeRFail_S.py (AS:278-392) through the 8-NN Gaussian
    scatter over 16 views = 2 batches of 8, perturbation updated after every batch, stand-in 800x800 victim CNN."""
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.attack import nerfail_s_loop
    t = time.time()
    wi, ori, s_init = _attack_inputs(dev, n_views, seed=40)
    torch.cuda.synchronize()
    t_build = time.time() - t
    torch.manual_seed(0)
    victim = victim_cnn(8).to(dev).requires_grad_(False)
    net = gauss_net(dev, 0.02, victim, 'my_model', epsilon=None)
    net.cache_ori_cla = True
    batches = [(wi[b:b + batch].contiguous(), ori[b:b + batch].contiguous()) for b in range(0, n_views, batch)]
    label = torch.tensor(4, device=dev)
    nerfail_s_loop(net, s_init, s_init, batches, label, 1)                  # warm-up: inverted indices, MIOpen plans
    torch.cuda.synchronize()
    t = time.time()
    s = nerfail_s_loop(net, s_init, s_init, batches, label, iters)
    torch.cuda.synchronize()
    dt = time.time() - t
    steps = iters * len(batches)
    return {'iters_per_sec': steps / dt, 'ms_per_iter': dt / steps * 1e3, 'loop_seconds': dt, 'loop': '%d iterations x %d batches of %d views'
            % (iters, len(batches), batch), 'map_build_seconds_16_views_knn8_plus_weights': t_build,
            'moved_fraction': float((s[..., :3] != 0).float().mean()),
            'note': 'one "iter" = one batch step (gauss_net forward, CE, backward, sign step); maps built by K8+K9 on '
                    'the analytic sphere pts_max; original-image logits cached (identical results)'}


def multi_gpu_legs(dev, world, rank, steps, nets, K, legs=('render', 'attack')):
    """N > 1 only. (a) strong-scaling render: ONE view per step, its 640 000 rays cut into contiguous per-rank ranges
    (sharding.render_shard; no collective). (b) NeRFail-S attack step, cfg5 shape: the 8 views of ONE batch split over
    ranks, one all-reduce (C1, RCCL over xGMI) of the 30.72 MB perturbation gradient, identical sign step everywhere."""
    from nerfail_amd import sharding
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.attack import nerfail_s_step
    coarse, fine = nets
    kw = dict(network_query_fn=None, perturb=0., N_importance=N_IMPORTANCE, network_fine=fine, N_samples=N_SAMPLES,
              network_fn=coarse, white_bkgd=True, raw_noise_std=0.)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n, warm=1):
        for i in range(warm):
            fn(i)
        barrier()
        t0 = time.time()
        for i in range(n):
            fn(warm + i)
        barrier()
        t = torch.tensor([time.time() - t0], device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    thetas = np.linspace(-180, 180, 41)[:-1]

    def strong(i):
        c2w = torch.from_numpy(synth.pose_spherical(float(thetas[i % len(thetas)]), -30., 4.)[:3, :4])
        with torch.no_grad():
            return sharding.render_shard(H, W, K, c2w, 2., 6., rank, world, chunk=H * W, **kw)
    out = {}
    if 'render' in legs:
        dt = timed(strong, steps)
        out['render_strong'] = {'rays_per_sec': steps * H * W / dt, 'ms_per_view': dt / steps * 1e3, 'scaling': 'strong',
                                'rays_per_rank': [hi - lo for lo, hi in sharding.shard_ranges(H * W, world)],
                                'note': 'one 800x800 view per step, contiguous ray ranges per rank, no collective'}
    if 'attack' not in legs:
        return out

    B = 8
    wi, ori, s_init = _attack_inputs(dev, B, seed=60)
    torch.manual_seed(0)
    victim = victim_cnn(8).to(dev).requires_grad_(False)
    net = gauss_net(dev, 0.02, victim, 'my_model', epsilon=None)
    net.cache_ori_cla = True
    label = torch.tensor(4, device=dev)
    timing = {}
    state = {'s': s_init.clone()}

    def attack(i):
        state['s'], _ = nerfail_s_step(net, state['s'], s_init, wi, ori, label, 2.0, 32.0, False, timing=timing)
    attack(0)
    timing.clear()
    n_it = max(5, steps)
    dt = timed(attack, n_it, warm=1)
    ev = timing.get('allreduce_events', [])[1:]                     # (the first is the warm-up iteration inside timed)
    ar_ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
    nbytes = ev[0][2] if ev else 0
    ar = float(np.median(ar_ms)) if ar_ms else None
    seen = [None] * world
    dist.all_gather_object(seen, (rank, torch.cuda.current_device()))
    # all ranks must hold the identical perturbation after the loop (the property the sharding preserves)
    chk = state['s'].double().sum().reshape(1).to(dev if dist.get_backend() == 'nccl' else 'cpu')
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    out['attack'] = {'iters_per_sec': n_it / dt, 'ms_per_iter': dt / n_it * 1e3, 'batch_views': B,
                     'views_per_rank': [hi - lo for lo, hi in sharding.shard_ranges(B, world)],
                     'allreduce_ms': ar, 'allreduce_bytes': nbytes, 'allreduce_backend': dist.get_backend(),
                     'allreduce_bus_GBps': (2.0 * (world - 1) / world * nbytes / (ar * 1e-3) / 1e9) if ar else None,
                     'xgmi_per_link_peak_GBps': 153.0, 'ranks_seen': seen,
                     'perturbation_identical_on_all_ranks': bool(float(lo_[0]) == float(hi_[0])),
                     'expected_allreduce_ms_8_gpus': '0.05 (direct reduce-scatter + all-gather over 7 links) .. 0.35 (ring, one link), SURVEY section 5',
                     'note': 'NeRFail-S step (AS:304-392) end to end with the stand-in 800x800 victim CNN; ONE collective per step: '
                             'the [Ns,3] gradient with the loss in its tail (23.04 MB + 4 B); bus GB/s = 2(N-1)/N x bytes / time '
                             '(ring-equivalent), to compare with one xGMI link'}
    return out


# ------------------------------------------------------------------------------------------------------------ children
def _emitter(path):
    """emit(obj): one JSON object per line, appended to `path` (flushed + fsynced: survives a GPU fault of this process) or
    printed to stdout when the child was started by hand without --out."""
    def emit(obj):
        line = json.dumps(obj)
        if path is None:
            print(line, flush=True)
            return
        with open(path, 'a') as f:
            f.write(line + '\n')
            f.flush()
            os.fsync(f.fileno())
    return emit


def _device(args):
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU path)')
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    return torch.device('cuda', dev_index)


def child_render(args, emit):
    """The contract's step: full 800x800 renders, HIP-event timing of the MLP and composite launches inside the timed region.
    N > 1: every rank is one of these children (env from torch.distributed.run); rank 0 emits the line, then all ranks run
    the strong-scaling render leg and the sharded attack leg (emitted afterwards: a failure there cannot cost the line)."""
    _heavy_imports()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    dev = _device(args)
    rccl_ranks = None
    if world > 1:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        with _StdoutToStderr():
            # (a rank that dies leaves the others in a collective: they give up after the timeout instead of hanging)
            to = datetime.timedelta(seconds=int(os.environ.get('NERFAIL_BENCH_PG_TIMEOUT', '180')))
            if args.dist_backend == 'nccl':
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=to)
                ones = torch.ones(1, device=dev)
                dist.all_reduce(ones)                                # creates the communicator now (banner -> stderr)
                torch.cuda.synchronize()
                rccl_ranks = int(ones.item())                        # ranks that really took part in an RCCL collective
                # asserted BEFORE the timed region: a line that says n_gpus N was measured on N RCCL ranks
                assert rccl_ranks == world, 'RCCL all-reduce of ones returned %d on a %d-rank group' % (rccl_ranks, world)
            else:
                dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=to)
    assert world == args.gpus, '--gpus %d but WORLD_SIZE %d (bench.py spawns the ranks itself when no launcher did)' % (args.gpus, world)

    from nerfail_amd import nerf_to_coord as NC, run_nerf as RN
    _, coarse = make_net(21, dev)
    _, fine = make_net(22, dev)
    coarse.packed(), fine.packed()
    focal, K = synth.lego_intrinsics(H, W)
    thetas = np.linspace(-180, 180, 41)[:-1]
    kw = dict(network_query_fn=None, perturb=0., N_importance=N_IMPORTANCE, network_fine=fine, N_samples=N_SAMPLES,
              network_fn=coarse, use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)

    mlp_events = []
    orig_mlp = RN._mlp_points

    def timed_mlp(fn, pts, viewdirs):       # HIP events on the stream the kernel is launched on (torch's current)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_mlp(fn, pts, viewdirs)
        e1.record()
        mlp_events.append((e0, e1, pts.shape[0] * pts.shape[1]))
        return out
    RN._mlp_points = timed_mlp
    orig_mlp_rays = RN._mlp_rays

    def timed_mlp_rays(fn, rays_, z_vals, acts=None):     # the same kernel, points formed inside (round 3: the path render takes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_mlp_rays(fn, rays_, z_vals, acts)
        e1.record()
        mlp_events.append((e0, e1, z_vals.shape[0] * z_vals.shape[1]))
        return out
    RN._mlp_rays = timed_mlp_rays
    comp_events = []
    orig_comp = RN._composite

    def timed_composite(raw, z_vals, rays, noise, white_bkgd, pts=None, want_pts_max=None):   # the compositing scan (K5), same event scheme
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_comp(raw, z_vals, rays, noise, white_bkgd, pts, want_pts_max)
        e1.record()
        comp_events.append((e0, e1, z_vals.shape[0] * (24 * z_vals.shape[1] + 36)))     # 24N + 36 B per ray
        return out
    RN._composite = timed_composite

    def step(i):
        c2w = synth.pose_spherical(float(thetas[(i * world + rank) % len(thetas)]), -30., 4.)[:3, :4]
        return NC.render(H, W, K, chunk=H * W, c2w=torch.from_numpy(c2w), near=2., far=6., **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        with torch.no_grad():
            step(i)
    barrier()
    mlp_events.clear()
    comp_events.clear()
    t0 = time.time()
    for i in range(args.steps):
        with torch.no_grad():                      # render-only, as nerf_to_coord.py:619 does
            out = step(args.warmup + i)
    barrier()
    elapsed = time.time() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    assert torch.isfinite(out[0]).all()

    # HBM traffic of the dominant kernel from the separate rocprofv3 --pmc passes of this same command (corrected as
    # MI355X_MICROARCH.md prescribes); profiles/ travels with the repo, the counters cannot be read from inside bench.py
    traffic, traffic_source = pmc_traffic('nerf_mlp_fwd_lds_kernel<8,1,false>')
    comp_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in comp_events)
    comp_bytes = sum(b for _, _, b in comp_events)
    mlp_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in mlp_events)
    mlp_samples = sum(n for _, _, n in mlp_events)
    achieved = mlp_samples * FLOP_PER_SAMPLE / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
    RN._mlp_points, RN._composite, RN._mlp_rays = orig_mlp, orig_comp, orig_mlp_rays

    if rank == 0:
        rays_total = world * args.steps * H * W
        emit({
            'metric': 'rays/sec', 'value': rays_total / elapsed, 'unit': 'rays/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'Blender-lego-shaped 800x800 full render incl. per-pixel argmax point '
                                   '(nerf_to_coord.render), 64+128 samples, D=8 W=256 coarse+fine, white_bkgd, '
                                   'one view per step per GPU (BASELINE.json configs[1])',
                       'rays_per_step_per_gpu': H * W, 'chunk': H * W, 'pass': 'forward (render)',
                       'parallelism': 'view-per-rank, no collective'},
            'roofline': {'bound': 'mfma', 'kernel': 'nerf_mlp_fwd_lds_kernel<8,1>', 'achieved': achieved,
                         'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS,
                         'traffic': traffic, 'traffic_unit': 'HBM+IC bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)',
                         'traffic_source': traffic_source,
                         'mfma_busy': (pmc_mfma_busy('nerf_mlp_fwd_lds_kernel<8,1,false>') or {}).get('mfma_busy'),
                         'clock_GHz': (pmc_mfma_busy('nerf_mlp_fwd_lds_kernel<8,1,false>') or {}).get('clock_GHz'),   # the clock the chip held in that counter pass (the peak assumes 2.4)
                         'mfma_busy_source': pmc_mfma_busy('nerf_mlp_fwd_lds_kernel<8,1,false>'),
                         'launches': len(mlp_events), 'avg_launch_ms': mlp_ms / max(1, len(mlp_events)),
                         'flop_per_sample': FLOP_PER_SAMPLE, 'mlp_share_of_step': mlp_ms * 1e-3 / elapsed},
            # the other roofline the north star asks for: achieved HBM rate of the compositing scan (K5 + K7)
            'composite_scan': {'bound': 'hbm', 'kernel': 'composite2_kernel<2|6>', 'achieved': comp_bytes / max(comp_ms, 1e-9) / 1e6,
                               'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': comp_bytes / max(comp_ms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                               'launches': len(comp_events), 'ms_total': comp_ms,
                               'note': 'algorithmic 24N+36 B per ray (coarse N=64 and fine N=192 launches together)'},
            'rccl_ranks': rccl_ranks,
        })

    if world > 1 and not args.no_attack:
        # The legs hold collectives: an exception on ONE rank would leave the others waiting. Every rank reports its own
        # outcome, then all of them agree (MAX of an error flag) before anyone trusts the numbers.
        err = None
        try:
            legs = multi_gpu_legs(dev, world, rank, args.steps, (coarse, fine), K)
        except Exception as e:
            legs, err = {}, '%s: %s' % (type(e).__name__, str(e)[:300])
        flag = torch.tensor([1.0 if err else 0.0], device=dev if args.dist_backend == 'nccl' else 'cpu')
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        except Exception as e:                                           # the group itself is gone
            err = err or '%s: %s' % (type(e).__name__, str(e)[:300])
        if rank == 0:
            if err or float(flag[0]) > 0:
                emit({'multi_gpu_legs_error': err or 'a peer rank failed inside the legs'})
            else:
                emit(legs)
    if world == 1 and os.environ.get('NERFAIL_BENCH_DRYRUN_NCCL', '0') == '1':
        # RCCL dry run on one GPU: a 1-rank nccl group bound to the device, the attack leg's gradient all-reduce issued
        # through it (sum over one rank = identity) - communicator, stream semantics and HIP-event timing executed once
        try:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29517')
            with _StdoutToStderr():
                dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
                dist.all_reduce(torch.zeros(1, device=dev))
                torch.cuda.synchronize()
            os.environ['NERFAIL_FORCE_COLLECTIVE'] = '1'
            emit({'attack_nccl_dryrun': multi_gpu_legs(dev, 1, 0, args.steps, (coarse, fine), K, legs=('attack',))['attack']})
        except Exception as e:
            emit({'attack_nccl_dryrun_error': '%s: %s' % (type(e).__name__, str(e)[:300])})
        finally:
            os.environ['NERFAIL_FORCE_COLLECTIVE'] = '0'
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _guarded(emit, name, fn):
    """One leg of a child: its result, or `<name>_error` - a Python exception in one leg never costs the others."""
    try:
        emit({name: fn()})
    except Exception as e:
        emit({name + '_error': '%s: %s' % (type(e).__name__, str(e)[:300])})


def child_cpu(args, emit):
    """The three CPU baselines (no GPU is touched), on all N_CPU cores of the cgroup quota, nothing else running."""
    _heavy_imports(gpu=False)
    _guarded(emit, 'cpu_baseline', lambda: cpu_baseline(float(os.environ.get('NERFAIL_BENCH_CPU_SECONDS', '12'))))
    _guarded(emit, 'cpu_baseline_fwd_bwd', lambda: cpu_baseline_fwd_bwd(6.0))
    _guarded(emit, 'cpu_baseline_attack', lambda: cpu_baseline_attack(6.0))


def child_train(args, emit):
    _heavy_imports()
    dev = _device(args)
    _guarded(emit, 'train', lambda: train_bench(dev))
    if 'f16x3' in args.section_set:
        _guarded(emit, 'train_f16x3', lambda: train_bench(dev, precision='f16x3'))


def child_attack(args, emit):
    _heavy_imports()
    dev = _device(args)
    acc = _Emitting(lambda part: emit({'attack': part, '_merge': True}))
    try:
        attack_bench(dev, out=acc)
    except Exception as e:
        emit({'attack_error': '%s: %s' % (type(e).__name__, str(e)[:300])})
    emit({'attack': dict(acc), '_merge': True})          # nested fields set after a leg's first assignment
    if os.environ.get('NERFAIL_BENCH_LIGHT', '0') != '1':
        try:
            emit({'attack': {'cfg3_loop': cfg3_bench(dev)}, '_merge': True})
        except Exception as e:
            emit({'attack': {'cfg3_loop_error': '%s: %s' % (type(e).__name__, str(e)[:300])}, '_merge': True})


def child_extras(args, emit):
    _heavy_imports()
    dev = _device(args)
    if 'knn' in args.section_set:
        _guarded(emit, 'knn', lambda: knn_bench(dev))
    if 'f16x3' in args.section_set:
        _guarded(emit, 'render_f16x3', lambda: render_f16x3_bench(dev))


def child_selftest(args, emit):
    """No GPU, no torch: lets tests/test_bench_parent.py check the parent's containment (a child killed mid-way keeps what it
    had emitted and becomes `selftest_error`)."""
    emit({'selftest': {'before': 1}})
    if os.environ.get('NERFAIL_BENCH_SELFTEST_DIE', '0') == '1':
        os.kill(os.getpid(), 9)
    emit({'selftest': {'after': 2}, '_merge': True})


def child_fake_render(args, emit):
    """tests/test_bench_parent.py only (NERFAIL_BENCH_FAKE_RENDER=<json file>): a render child without a GPU. It emits the
    objects of a stored, fully populated run, with n_gpus = the number of ranks that really met in a gloo all-reduce under
    the environment the parent (or torch.distributed.run) set - so the spawn / env logic of `--gpus N` is what is tested."""
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    met = 1
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        t = torch.ones(1)
        dist.all_reduce(t)
        met = int(t.item())
    if os.environ.get('NERFAIL_BENCH_FAKE_RENDER_DIE') == str(rank):
        os.kill(os.getpid(), 9)                                  # before the line's value exists
    if rank == 0:
        stored = json.load(open(os.environ['NERFAIL_BENCH_FAKE_RENDER']))
        stored.pop('sections', None)
        stored.update({'n_gpus': met, 'local_rank_seen': int(os.environ.get('LOCAL_RANK', '-1'))})
        emit(stored)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


CHILDREN = {'selftest': child_selftest, 'render': child_render, 'cpu': child_cpu, 'train': child_train, 'attack': child_attack, 'extras': child_extras}




def run_child(args):
    """`bench.py --child <group> [--out file]`: run ONE section group in this process."""
    emit = _emitter(args.out)
    if os.environ.get('NERFAIL_BENCH_KILL_GROUP') == args.child:
        # containment drill: this child SIGKILLs itself right after its first result object
        plain = emit

        def emit(obj):
            plain(obj)
            os.kill(os.getpid(), 9)
    if args.child == 'render' and os.environ.get('NERFAIL_BENCH_FAKE_RENDER'):
        return child_fake_render(args, emit)
    CHILDREN[args.child](args, emit)
