"""Benchmark of the generation hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

metric  : audio seconds generated / wall-second (real-time factor), MusicGen-medium 30 s @ 32 kHz
workload: BASELINE.json configs[2] -- MusicGen-medium (1.5B) bf16 weights + bf16 KV cache, batch 8
          prompts x 30 s PER GPU (weak scaling), CFG on (16 rows), top-k 250 sampling, 1503
          autoregressive positions, then EnCodec-32k decode of the [8, 4, 1500] tokens to [8, 1, 960000].
          Random-init weights of that architecture, synthetic T5 stand-in (16 x 768 per prompt): no
          checkpoints exist offline.
step    : one full generate (conditioning given -> tokens -> waveform on device).  N > 1: rank 0
          builds + broadcasts the global conditioning, every rank generates its 8 prompts, tokens are
          all-gathered (both collectives are inside the timed region).

Extra objects on the JSON line:
  roofline     dominant kernel = lin_tiled_kernel (weight-streaming skinny GEMM, HBM bound): algorithmic
               weight bytes of one decode position / number of its launches, divided by the
               average launch duration measured here with HIP events over one decode position's worth
               of launches (all layers, real shapes) on the launch stream.
  cpu_baseline the reference's CPU path timed on this box's host cores on a bounded sample of the same workload:
               the imported reference itself where /root/reference exists (kind "reference"), else the oracle, a
               port of the same algorithm incl. its torch.cat KV cache (kind "port": the GPU box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0  # what a float4 copy kernel reaches on this chip (same guide): the practical ceiling of a stream


def lm_algorithmic_bytes(lm, B_eff: int, n_positions: int, Lc: int, prefix: int = 0):
    """per position W = bw*(14|12)*d^2*L + bw*4*d*card, KV read 2*B_eff*d*L*bk*t,
    KV write 2*B_eff*d*L*bk, cross-KV 2*B_eff*Lc*d*L*bk."""
    d, L, card, K = lm.dim, lm.num_layers, lm.card, lm.n_q
    bw = 2 if lm.weight_dtype == torch.bfloat16 else 4
    bk = 2 if lm.kv_dtype == torch.bfloat16 else 4
    per_layer = (14 if lm.has_cross_attention else 12) * d * d
    w_step = bw * (per_layer * L + K * d * card)
    kv_total = sum(2 * B_eff * d * L * bk * (prefix + t + 1) for t in range(n_positions))
    kv_write = 2 * B_eff * d * L * bk * n_positions
    cross = 2 * B_eff * Lc * d * L * bk * n_positions if lm.has_cross_attention else 0
    return dict(w_step=w_step, total=w_step * n_positions + kv_total + kv_write + cross)


def measure_lin_kernel(model, B_eff: int, reps: int = 3):
    """Average lin_tiled_kernel launch duration over one decode position's worth of launches (HIP events on the
    launch stream), and the algorithmic bytes those launches stream."""
    from audiocraft_amd import _C
    lm = model.lm
    pk = lm._packed
    dev = lm.device
    d, ffn, wd = lm.dim, lm.ffn_dim, lm.weight_dtype
    rnd = lambda n: torch.randn(B_eff, n, device=dev)  # noqa: E731
    kt = 32 if wd == torch.bfloat16 else 16
    dp = -(-d // kt) * kt
    rbs = 2 * dp // kt
    hid = _C.tile_matrix(rnd(ffn), wd)
    catt = _C.tile_matrix(rnd(d), wd)                      # cross-attention output
    # the residual stream: f32 row-major (x), raw in A-fragment order as a bf16 hi / lo pair, the hi buffer
    # holding [x | self-attention output] side by side (two pairs: the paired launch reads one, writes the
    # other) and its per-row (mean, M2) partials (stats): exactly what acmi_lm_step keeps
    x, x0 = torch.zeros(B_eff, d, device=dev), rnd(d)       # x = x0 + a @ W^T keeps the synthetic stream bounded
    cat = torch.zeros(B_eff, 2 * dp, device=dev)
    cat[:, :d], cat[:, dp:dp + d] = rnd(d), rnd(d)
    xh = [_C.tile_matrix(cat, wd), _C.tile_matrix(cat, wd)]
    # bf16: single-term fragments with a per-row shift (the default of acmi_lm_step), or the hi / lo pair (ACMI_LN_LO=1)
    hilo = wd == torch.bfloat16 and os.environ.get('ACMI_LN_LO', '') == '1'
    xl = [_C.tile_matrix(rnd(d) * 2.0 ** -9, wd) if hilo else None for _ in range(2)]
    shifts = [torch.zeros(B_eff, device=dev), torch.zeros(B_eff, device=dev)] if (wd == torch.bfloat16 and not hilo) else None
    # single-term + shift mode, <= 32 rows: the LayerNorm-consuming GEMMs take their row statistics from the fragments
    # (no a_stats) and only the paired out projection still writes partials (for the cross-attention query hook)
    gram = shifts is not None and B_eff <= 32 and os.environ.get('ACMI_LN_GRAM', '') != '0'
    layer_no = 0
    np_ = max(1, d // 16)                                     # statistics partials of the current x (cnt elements each)
    stats = torch.zeros(B_eff, max(1, d // 8), 2, device=dev)
    stats[..., 1] = 16.0
    q = torch.empty(B_eff, d, device=dev)
    qkv = torch.empty(B_eff, 4 * d, device=dev)
    h = _C.tiled_activation_buffer(B_eff, ffn, wd, dev)
    logits = torch.empty(B_eff, lm.n_q * lm.card, device=dev)
    launches = 0
    nbytes = 0
    cur = 0

    def wbytes(w):
        return w.N * w.K * w.data.element_size()

    def sh_in():    # shift the current x fragments were stored with / the one this layer's producers use
        return None if shifts is None else shifts[layer_no & 1]

    def sh_out():
        return None if shifts is None else shifts[(layer_no + 1) & 1]

    def consume(w, out, out_mode, colsum, bias, act=0, first_of_layer=False, produced=True, publish=True):
        # LayerNorm(x) @ W'^T: raw fragments + statistics + column sums; the QKV launch also publishes the row means
        nonlocal launches, nbytes
        _C.linear_ex(xh[cur], w, out, B_eff, _C.A_TILED, out_mode, a_stats=None if gram else stats, np_=np_, cnt=d // np_, bias=bias, act=act,
                     a_lo=xl[cur], colsum=colsum, a_rbs=rbs, a_shift=sh_out() if produced and not first_of_layer else sh_in(),
                     mean_out=sh_out() if (first_of_layer and publish) else None)
        launches += 1
        nbytes += wbytes(w)

    def produce_desc(a, w, a_rbs, dst, stats_needed=False):
        return _C.linear_desc(a, w, x, B_eff, _C.A_TILED, _C.OUT_F32, residual=x0,
                              stats_out=None if (gram and not stats_needed) else stats, xt_hi=xh[dst],
                              xt_lo=xl[dst], a_rbs=a_rbs, xt_rbs=rbs, xt_shift=sh_out())

    def produce(a, w):
        # x = x0 + a @ W^T, also written as raw fragments and statistics partials (of 16 elements; of 8 when the weight
        # is in half-tile order: FFN2 on 8-feature workgroups)
        nonlocal launches, nbytes, np_
        _C.linear_launch(produce_desc(a, w, 0, cur))
        np_ = d // (8 if w.half else 16)
        launches += 1
        nbytes += wbytes(w)

    def one_position():
        # the same GEMM launches acmi_lm_step issues for one position (same shapes, operand layouts, weights)
        nonlocal launches, nbytes, cur, np_, layer_no
        for layer_no, ent in enumerate(pk['per_layer']):
            if 'w_qkvx' in ent:
                # QKV + the x0 part of the cross-attention query (a fourth block of features), then the out projection
                # and the att part of that query in one launch (acmi_linear_pair); acmi_lm_layer.w_qkvx / w_mq
                consume(ent['w_qkvx'], qkv, _C.OUT_F32, ent['cs_qkvx'], ent['b_qkvx'], first_of_layer=True)
                att_half = xh[cur].view(-1)[(dp // kt) * 64 * (16 // xh[cur].element_size()):]
                p0 = produce_desc(att_half, ent['w_out'], rbs, cur ^ 1, stats_needed=True)
                p1 = _C.linear_desc(att_half, ent['w_mq'], q, B_eff, _C.A_TILED, _C.OUT_F32, residual=q, a_rbs=rbs)
                _C.linear_pair(p0, p1)
                launches += 1
                # consume() counted w_qkvx = W_qkv + W_cq, as SURVEY.md section 8(d) does; w_mq (= W_cq' W_out, +d^2
                # elements per layer) is extra traffic of this formulation and shows up in the PMC bytes instead
                nbytes += wbytes(ent['w_out'])
                np_ = d // 16
                cur ^= 1
                produce(catt, ent['w_cout'])
            else:
                consume(ent['w_qkv'], qkv, _C.OUT_F32, ent['cs_qkv'], ent['b_qkv'], first_of_layer=True)
                att_half = xh[cur].view(-1)[(dp // kt) * 64 * (16 // xh[cur].element_size()):]
                _C.linear_launch(produce_desc(att_half, ent['w_out'], rbs, cur))
                launches += 1
                nbytes += wbytes(ent['w_out'])
                np_ = d // 16
            consume(ent['w_ff1'], h, _C.OUT_TILED, ent['cs_ff1'], ent['b_ff1'], act=1)
            produce(hid, ent['w_ff2h'] if 'w_ff2h' in ent and B_eff <= 32 else ent['w_ff2'])
        layer_no += 1
        consume(pk['w_head'], logits, _C.OUT_F32, pk['cs_head'], pk['b_head'], first_of_layer=True, publish=False)   # shift = the last layer's

    one_position()  # warm
    torch.cuda.synchronize()
    # capture the position's launches into a hipGraph so that the host (python + ctypes, ~10 us per call) is
    # out of the measurement, exactly like in generate(); HIP events bracket the replays on the launch stream
    launches = nbytes = 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        try:
            one_position()
        finally:
            graph.capture_end()
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            graph.replay()
        e1.record(side)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    ms = e0.elapsed_time(e1)
    return dict(avg_us=ms * 1e3 / (launches * reps), bytes_per_launch=nbytes / launches, launches_per_position=launches)


def measure_attn_kernel(model, B_eff: int, T: int, reps: int = 3, prefix: int = 0):
    """Average duration of the self-attention decode launch at the MEAN context of a T-frame generate: one launch per layer on
    that layer's own KV cache (cold, as in a decode position), captured into a hipGraph, HIP events on the launch stream.
    Algorithmic bytes: K and V rows [0, context) of every (row, head), each read once."""
    from audiocraft_amd import _C
    lm = model.lm
    run = lm._run
    k, v = run['k'], run['v']                     # [L, Beff, H, Tmax, hd], what the last generate left
    L, _, H, Tmax, hd = k.shape
    ctx = max(1, min(Tmax, prefix + (T + 3) // 2))
    q = torch.randn(B_eff, H * hd, device=k.device)
    out = _C.tiled_activation_buffer(B_eff, H * hd, lm.weight_dtype, k.device)
    pos = torch.tensor([ctx - 1, 0, 0, 0], dtype=torch.int32, device=k.device)   # the length is a device word, as in generate()

    def one_pass():
        for li in range(L):
            _C.attn_decode(q, k[li], v[li], out, 0, len_dev=pos, len_bias=1, out_tiled=True)

    one_pass()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        try:
            one_pass()
        finally:
            graph.capture_end()
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            graph.replay()
        e1.record(side)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    nbytes = 2 * B_eff * H * ctx * hd * k.element_size()
    return dict(avg_us=e0.elapsed_time(e1) * 1e3 / (L * reps), bytes_per_launch=nbytes, context=ctx, launches=L)


def pmc_traffic_per_launch():
    """-> (HBM bytes per lin_tiled_kernel launch, source) from the committed rocprofv3 PMC passes over the same launch
    chain (profiles/r*_lin_chain_pmc_{FETCH,WRITE}_SIZE.csv, separate --pmc passes, scripts/dbg_chain.py):
    FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half size
    (MI355X_MICROARCH.md, HBM section), hence the x2.  (None, reason) if no PMC summary is present.  The counters cannot be
    collected inside this process; `source` says which files the figure was read from and how old they are, so that a
    stale figure is visible in the bench line."""
    import csv
    import glob
    vals = {}
    used = []
    for name in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_lin_chain_pmc_{name}.csv')))
        if not files:
            return None, 'no PMC summary under profiles/'
        total = n = 0.0  # dispatch-weighted mean over the kernel's template variants (plain / folded LayerNorm)
        used.append(files[-1])
        for row in csv.DictReader(open(files[-1])):
            if ('lin_tiled_kernel' in row['kernel'] or 'lin_pair_kernel' in row['kernel']) and row['counter'] == name:
                total += float(row['mean_per_dispatch']) * float(row['dispatches'])
                n += float(row['dispatches'])
        if n == 0:
            return None, f'no lin_tiled_kernel rows in {os.path.basename(files[-1])}'
        vals[name] = total / n
    age_days = (time.time() - min(os.path.getmtime(f) for f in used)) / 86400.0
    src = ' + '.join(os.path.basename(f) for f in used) + f' (committed rocprofv3 --pmc passes of scripts/dbg_chain.py; file age {age_days:.1f} d)'
    return int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), src


def _run_child(cmd, env, timeout_s: int):
    """Run a profiler child in its OWN process group and end the whole group on a time-out (rocprofv3 forks the python
    process: killing only the parent would leave a grandchild on the GPU during the measurements that follow)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        rc = p.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.wait()
        raise
    if rc != 0:
        raise subprocess.CalledProcessError(rc, cmd)


def in_situ_kernel_stats(model_name: str, batch: int, timeout_s: int = 420, duration: float = 8.0, greedy: bool = False,
                         melody_seconds: float = 0.0):
    """-> ({family: (calls, avg_us)}, source) of the decode kernels INSIDE a real generate: a `rocprofv3 --kernel-trace --stats`
    child process over scripts/short_generate.py (the bench model, one 8 s generate: every GEMM launch sits between its real
    neighbours -- attention kernels, sampler -- in the captured decode graph).  `measure_lin_kernel` times the GEMM launches of a
    position as an isolated chain and came out ~6 % kinder than the same kernels in situ (round 4); the bench line's
    `roofline.frac` is therefore computed from THIS average, the isolated one is reported next to it.  (None, reason) when the
    pass cannot run (no rocprofv3, ACMI_BENCH_INSITU=0, already under a profiler, failure / timeout)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('ACMI_BENCH_INSITU', '1') == '0':
        return None, 'ACMI_BENCH_INSITU=0'
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCPROFILER')) for k in os.environ):
        return None, 'this process runs under a profiler'
    tmp = tempfile.mkdtemp(prefix='acmi_insitu_', dir='/tmp')
    t0 = time.time()
    try:
        env = dict(os.environ, TMPDIR='/tmp')
        cmd = [exe, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', tmp, '--', sys.executable,
               os.path.join(ROOT, 'scripts', 'short_generate.py'), model_name, str(batch), str(duration), '1' if greedy else '0',
               str(melody_seconds)]
        _run_child(cmd, env, timeout_s)
        files = glob.glob(os.path.join(tmp, '**', '*kernel_stats.csv'), recursive=True)
        if not files:
            return None, 'the kernel-trace pass wrote no kernel_stats.csv'
        fam = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                n = row['Name']
                key = ('fused' if 'qkv_attn_kernel' in n else
                       'gemm' if ('lin_tiled_kernel' in n or 'lin_pair_kernel' in n) else
                       'self_attn' if 'attn_decode_kernel' in n and ', false>' in n else
                       'cross_attn' if ('cross_q_kernel' in n or ('attn_decode_kernel' in n and ', true>' in n)) else None)
                if key is None:
                    continue
                c, tot = fam.get(key, (0, 0.0))
                fam[key] = (c + int(row['Calls']), tot + float(row['TotalDurationNs']))
        if 'gemm' not in fam:
            return None, 'no GEMM dispatches in the kernel-trace pass'
        keep = os.environ.get('ACMI_BENCH_INSITU_KEEP')   # copy the summary where the caller wants it (profiles/)
        if keep:
            shutil.copyfile(files[0], keep)
        out = {k: (c, tot / c / 1e3) for k, (c, tot) in fam.items()}
        return out, (f'rocprofv3 --kernel-trace --stats over scripts/short_generate.py ({model_name}, {batch} x {duration:g} s after a '
                     f'2 s warm-up generate) in this run, {time.time() - t0:.0f} s: {out["gemm"][0]} GEMM launches')
    except Exception as e:   # noqa: BLE001
        return None, f'the kernel-trace pass failed: {type(e).__name__}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic_live(model_name: str = 'facebook/musicgen-medium', B_eff: int = 16, timeout_s: int = 200):
    """-> (HBM bytes per GEMM launch, source) MEASURED in this run: two `rocprofv3 --pmc` child processes (FETCH_SIZE and
    WRITE_SIZE in separate passes, counters only -- no trace domains -- as MI355X_MICROARCH.md prescribes) over
    scripts/dbg_chain.py, the same launch chain `measure_lin_kernel` times; same arithmetic as pmc_traffic_per_launch.
    (None, reason) when rocprofv3 is absent, this process is itself being profiled, ACMI_BENCH_PMC=0, or a pass fails / times
    out: the caller then falls back to the committed passes and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('ACMI_BENCH_PMC', '1') == '0':
        return None, 'ACMI_BENCH_PMC=0'
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCPROFILER')) for k in os.environ):
        return None, 'this process runs under a profiler'
    vals = {}
    t0 = time.time()
    for name in ('FETCH_SIZE', 'WRITE_SIZE'):
        tmp = tempfile.mkdtemp(prefix=f'acmi_pmc_{name}_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            cmd = [exe, '--pmc', name, '--output-format', 'csv', '-d', tmp, '--', sys.executable,
                   os.path.join(ROOT, 'scripts', 'dbg_chain.py'), model_name, str(B_eff)]
            _run_child(cmd, env, timeout_s)
            files = glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True)
            if not files:
                return None, f'the {name} pass wrote no counter_collection.csv'
            total = n = 0.0
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row['Counter_Name'] == name and ('lin_tiled_kernel' in row['Kernel_Name'] or 'lin_pair_kernel' in row['Kernel_Name']):
                        total += float(row['Counter_Value'])
                        n += 1
            if n == 0:
                return None, f'no GEMM dispatches in the {name} pass'
            vals[name] = total / n
        except Exception as e:   # noqa: BLE001  (a failed profiler pass must never cost the bench line)
            return None, f'the {name} pass failed: {type(e).__name__}'
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    src = f'measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/dbg_chain.py {model_name} {B_eff}, {time.time() - t0:.0f} s'
    return int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), src


def pmc_traffic_fused(model_name: str, batch: int, duration: float = 3.0, greedy: bool = False, timeout_s: int = 420):
    """-> (HBM bytes per qkv_attn_kernel launch, source), MEASURED in this run: two `rocprofv3 --pmc` child passes (FETCH_SIZE,
    WRITE_SIZE; counters only) over scripts/fused_chain.py (the bench model's LM with 4 layers -- the per-launch traffic does not
    depend on the depth --, tokens only: a 2 s warm-up + `duration` s), averaged over the fused QKV + self-attention dispatches;
    same gfx950 arithmetic as pmc_traffic_per_launch.  The caller prices the same dispatches' algorithmic bytes (weights + the
    mean K / V stream of those positions).  (A counter pass over a whole MusicGen.generate incl. the codec crashes rocprofv3.)"""
    import csv
    import glob
    import shutil
    import tempfile
    if os.environ.get('ACMI_BENCH_PMC', '1') == '0':
        return None, 'ACMI_BENCH_PMC=0'
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCPROFILER')) for k in os.environ):
        return None, 'this process runs under a profiler'
    vals = {}
    t0 = time.time()
    for name in ('FETCH_SIZE', 'WRITE_SIZE'):
        tmp = tempfile.mkdtemp(prefix=f'acmi_pmcf_{name}_', dir='/tmp')
        try:
            cmd = [exe, '--pmc', name, '--output-format', 'csv', '-d', tmp, '--', sys.executable,
                   os.path.join(ROOT, 'scripts', 'fused_chain.py'), model_name, str(batch), str(duration), '4']
            _run_child(cmd, dict(os.environ, TMPDIR='/tmp'), timeout_s)
            files = glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True)
            if not files:
                return None, f'the {name} pass wrote no counter_collection.csv'
            total = n = 0.0
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row['Counter_Name'] == name and 'qkv_attn_kernel' in row['Kernel_Name']:
                        total += float(row['Counter_Value'])
                        n += 1
            if n == 0:
                return None, f'no qkv_attn_kernel dispatches in the {name} pass'
            vals[name] = total / n
        except Exception as e:   # noqa: BLE001
            return None, f'the {name} pass failed: {type(e).__name__}'
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), (
        f'measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/fused_chain.py {model_name} '
        f'{batch} {duration:g} 4, qkv_attn_kernel dispatches only, {time.time() - t0:.0f} s')


def attn_traffic_committed(context: int, bytes_per_launch: int = 73826304):
    """HBM bytes per self-attention launch from the COMMITTED PMC passes (profiles/r05_attn_pmc_t751.txt: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate passes, over scripts/attn_bench.py --contexts 751 at the configs[2] geometry; same
    gfx950 arithmetic as pmc_traffic_per_launch) -- not measured live; None for another context."""
    path = os.path.join(ROOT, 'profiles', 'r05_attn_pmc_t751.txt')
    try:
        vals = {}
        for line in open(path):
            if 'attn_decode_kernel' in line:
                f = line.strip().split(',')
                vals[f[-3]] = float(f[-2])
        if context != 751 or bytes_per_launch != 73826304 or 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
            return {"traffic": None, "traffic_source": f"the committed PMC passes are of the configs[2] geometry at context 751 "
                                                       f"(this run: context {context}, {bytes_per_launch} B per launch)"}
        return {"traffic": int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024),
                "traffic_source": "committed passes (round 5, profiles/r05_attn_pmc_t751.txt): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over scripts/attn_bench.py --contexts 751"}
    except OSError:
        return {"traffic": None, "traffic_source": "profiles/r05_attn_pmc_t751.txt not found"}


def cpu_baseline(model, B: int, duration: float, Lc: int, top_k: int, early_steps: int = 16, late_steps: int = 14,
                 late_context: int = 1400):
    """The reference's CPU path on the host cores, on a bounded sample of the same workload, as SURVEY.md section 8(d)
    specifies: `early_steps` decode positions at the start of the stream AND `late_steps` positions at context
    `late_context` (KV state of that length: the cost of a position depends on the shapes only), both at the full batch;
    the per-position cost is taken as linear in the context between the two measurements and integrated over all
    positions; plus the EnCodec decode of 1 s of audio at the full batch, scaled to the duration.
    kind "reference": the UNMODIFIED reference imported from /root/reference (oracle/ref_baseline.py) -- where that tree
    exists (the build container); kind "port": the oracle, a restatement of the same algorithm incl. its torch.cat KV
    cache -- on the GPU box, where the reference does not travel."""
    from oracle import codec as ocodec
    from oracle import lm as olm
    from oracle import ref_baseline
    use_ref = ref_baseline.available() and os.environ.get('ACMI_BENCH_CPU_KIND', 'reference') != 'port'
    log = lambda msg: print(f"[cpu_baseline {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)  # noqa: E731
    default_threads = torch.get_num_threads()
    log(f"copying weights to the host (torch default: {default_threads} threads)")
    lm = model.lm
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=lm.dim, num_heads=lm.num_heads, num_layers=lm.num_layers, n_q=lm.n_q, card=lm.card,
                      cross_attention=lm.has_cross_attention)
    g = torch.Generator().manual_seed(0)
    cross = torch.randn(2 * B, Lc, lm.dim, generator=g)
    cross[B:] = 0
    T = int(duration * model.frame_rate)
    n_pos = T + 3

    ref_lm = None
    if use_ref:
        log("building the imported reference LMModel (kind: reference)")
        ref_lm = ref_baseline.build_reference_lm(sd, lm.dim, lm.num_heads, lm.num_layers, lm.n_q, lm.card,
                                                 lm.has_cross_attention)

    def run(steps):
        if ref_lm is not None:
            return ref_baseline.time_reference_positions(ref_lm, B, cross, top_k, steps, 1, 8)[0]
        t0 = time.perf_counter()
        olm.generate(sd, oc, None, B, cross, max_gen_len=T, top_k=top_k, max_steps=steps)
        return (time.perf_counter() - t0) / steps

    # give the CPU its best shot: the skinny (16-row) GEMMs of a decode position do not scale to every SMT
    # thread of a big host, so probe a few pool sizes (2 positions each) and keep the fastest
    cands = sorted({default_threads, max(1, default_threads // 2), 32, 16} & set(range(1, (os.cpu_count() or 1) + 1)))
    best_t, cores = None, default_threads
    for nthr in cands:
        torch.set_num_threads(nthr)
        run(1)
        t = run(2)
        log(f"{nthr} threads: {t * 1e3:.0f} ms/position")
        if best_t is None or t < best_t:
            best_t, cores = t, nthr
    torch.set_num_threads(cores)
    log(f"timing {early_steps} early-context and {late_steps} late-context ({late_context}) positions with {cores} threads")
    ctx_early = (early_steps - 1) / 2.0 + 1
    if ref_lm is not None:
        t_early, t_late = ref_baseline.time_reference_positions(ref_lm, B, cross, top_k, early_steps, late_steps, late_context)
        del ref_lm
    else:
        t_early = run(early_steps)
        # late context: a streaming state that already holds `late_context` positions per layer
        H, hd = lm.num_heads, lm.dim // lm.num_heads
        st = olm.LMState(lm.num_layers)
        for li in range(lm.num_layers):
            st.past_k[li] = torch.randn(2 * B, H, late_context, hd, generator=g)
            st.past_v[li] = torch.randn(2 * B, H, late_context, hd, generator=g)
        st.offset, st.first_step = late_context, False
        tok = torch.randint(0, lm.card, (2 * B, lm.n_q, 1), generator=g)
        with torch.no_grad():
            olm.lm_forward(sd, oc, tok, cross, None, st)   # untimed first call
            t0 = time.perf_counter()
            for _ in range(late_steps):
                logits = olm.lm_forward(sd, oc, tok, cross, None, st)
                olm.sample_next_token(olm.cfg_mix(logits, 3.0)[:, :, -1], True, 1.0, top_k, 0.0)
            t_late = (time.perf_counter() - t0) / late_steps
        del st
    ctx_late = late_context + 1 + (late_steps - 1) / 2.0
    slope = (t_late - t_early) / (ctx_late - ctx_early)
    t_lm = sum(t_early + slope * (t - ctx_early) for t in range(n_pos))
    log(f"{t_late * 1e3:.0f} ms/position at context {late_context}; EnCodec decode sample")
    del sd
    csd = {k: v.detach().float().cpu() for k, v in model.compression_model.state_dict().items()}
    cc = ocodec.CodecConfig(channels=1, dimension=128, n_filters=64, n_residual_layers=1, ratios=[8, 5, 4, 4],
                            causal=False, pad_mode='constant', lstm=2, norm='weight_norm', n_q=4, bins=2048,
                            sample_rate=32000, frame_rate=50)
    if use_ref:
        t_codec_1s = ref_baseline.time_reference_codec_decode(csd, B, 50, g)
    else:
        codes = torch.randint(0, 2048, (B, 4, 50), generator=g)
        t0 = time.perf_counter()
        ocodec.encodec_decode(csd, cc, codes, fast_lstm=True)
        t_codec_1s = time.perf_counter() - t0
    wall = t_lm + t_codec_1s * duration
    # port vs reference on ONE host (scripts/cpu_calibration.py, container only; the file travels with the repo): what makes
    # kind "port" a stated stand-in for the reference's speed rather than an assumption
    calib = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r05_cpu_calibration.json')) as f:
            cj = json.load(f)
        calib = (f"; calibration (same host, {cj['threads']} threads, same {cj['early_positions']} + {cj['late_positions']} positions): "
                 f"port / unmodified reference = {cj['port_over_reference']:.3f} in LM seconds for 1503 positions "
                 f"(reference {cj['reference_ms_per_position']['early']:.0f} / {cj['reference_ms_per_position']['late']:.0f} ms, "
                 f"port {cj['port_ms_per_position']['early']:.0f} / {cj['port_ms_per_position']['late']:.0f} ms per position early / late)")
    except (OSError, KeyError, ValueError):
        pass
    return dict(value=round(B * duration / wall, 4), unit='audio-s / wall-s', cores=cores, threads=cores,
                port_over_reference=None if calib is None else round(cj['port_over_reference'], 4),
                logical_cores=os.cpu_count(), threads_tried=cands, kind='reference' if use_ref else 'port',
                sample=f"{early_steps} decode positions at context <= {early_steps} ({t_early * 1e3:.0f} ms/position) and "
                       f"{late_steps} positions at context {late_context} ({t_late * 1e3:.0f} ms/position), batch {B} "
                       f"(CFG rows {2 * B}); per-position cost linear in the context between the two, integrated over "
                       f"{n_pos} positions = {t_lm:.0f} s; + EnCodec decode of 1 s ({t_codec_1s:.2f} s) x {duration:.0f}"
                       + (calib if (calib is not None and not use_ref) else ''))


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, RCCL over
    xGMI through torch.distributed, rendezvous on 127.0.0.1) and relay rank 0's JSON line."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared = os.environ.get('ACMI_ALLOW_SHARED_DEVICE', '') == '1' and have > 0   # test switch: ranks share devices (gloo)
    if have < n and not shared:
        print(f"bench.py: --gpus {n} needs {n} visible MI355X devices, this node shows {have} "
              f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}); nothing was run", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def config_tag(args) -> str:
    """Which BASELINE.json configuration the command line is (the default: configs[2], the one `metric` is quoted on)."""
    key = (args.model.split('/')[-1], args.batch, int(args.duration), bool(args.greedy))
    return {('musicgen-medium', 8, 30, False): 'BASELINE.json configs[2]',
            ('musicgen-small', 1, 10, True): 'BASELINE.json configs[1]',
            ('musicgen-large', 8, 30, False): "BASELINE.json configs[3], one GPU's shard of 8 prompts",
            ('musicgen-melody', 16, 30, False): 'BASELINE.json configs[4]'}.get(key, 'not a BASELINE.json configuration')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--model', default='facebook/musicgen-medium')
    ap.add_argument('--batch', type=int, default=8, help='prompts per GPU')
    ap.add_argument('--duration', type=float, default=30.0)
    ap.add_argument('--top-k', type=int, default=250)
    ap.add_argument('--text-len', type=int, default=16)
    ap.add_argument('--greedy', action='store_true', help='argmax decoding (BASELINE.json configs[1])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--dump-tokens', default='', help='rank 0 saves the gathered tokens of the last step here (tests)')
    ap.add_argument('--melody-seconds', type=float, default=10.0, help='melody clip per prompt for a melody model (configs[4])')
    ap.add_argument('--insitu-duration', type=float, default=-1.0,
                    help='seconds generated by the in-situ kernel-trace child (default: the bench duration, at most 30)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:   # no external launcher: spawn the ranks here
        sys.exit(_spawn_ranks(args.gpus))

    from audiocraft_amd import distributed as adist
    from audiocraft_amd.models.musicgen import MusicGen
    rank, world, local_rank = adist.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (there is no CPU fallback)")
    try:   # one process per GPU: LOCAL_RANK r drives HIP device r (ACMI_ALLOW_SHARED_DEVICE=1, tests: r % device_count)
        dev_idx = adist.device_index(local_rank)
    except RuntimeError as e:
        sys.exit(f"bench.py: {e}; --gpus {args.gpus} needs {args.gpus} MI355X on this node")
    torch.cuda.set_device(dev_idx)
    assert torch.cuda.current_device() == dev_idx
    dev = torch.device('cuda', dev_idx)

    model = MusicGen.get_random_init(args.model, dev, torch.bfloat16, text_len=args.text_len, seed=0)
    model.set_generation_params(use_sampling=not args.greedy, top_k=args.top_k, duration=args.duration)
    B = args.batch
    B_global = B * world
    T = int(args.duration * model.frame_rate)
    descriptions = [f"synthetic prompt {i}" for i in range(B_global)]

    melody = 'melody' in args.model
    if melody and world > 1:
        sys.exit("bench.py: the melody configuration (configs[4]) is a single-GPU line")
    mel = (torch.randn(B, 1, int(32000 * args.melody_seconds), generator=torch.Generator().manual_seed(5)).to(dev)
           if melody else None)

    def step(i):
        if melody:   # configs[4]: chroma front-end on the device + the prepended prefix are inside the step
            wav, tokens = model.generate_with_chroma(descriptions, mel, 32000, return_tokens=True)
            return tokens, wav
        tokens, wav = adist.generate_sharded(model, descriptions if rank == 0 else None, B_global, T,
                                             decode=True, base_seed=1000 * i)
        return tokens, wav

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    vlog = lambda msg: print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)  # noqa: E731
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        if rank == 0:
            vlog(f"warmup step {i}: {time.perf_counter() - t_w:.2f} s")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tokens, wav = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        # every rank's own clock over the same barrier-bracketed region: the MAX is the job's time, the list localises a straggler
        cdev = dev if torch.distributed.get_backend() != 'gloo' else 'cpu'
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        per_rank = [float(e.item()) for e in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert tokens.shape == (B_global, 4, T) and wav.shape == (B, 1, T * 640)
    assert int(tokens.min()) >= 0 and int(tokens.max()) < 2048 and bool(torch.isfinite(wav).all())

    ms_per_step = elapsed / args.steps * 1e3
    value = B_global * args.duration * args.steps / elapsed
    out = {
        "metric": "audio seconds generated / wall-sec (real-time factor), MusicGen-medium 30 s @ 32 kHz",
        "value": round(value, 3), "unit": "audio-s / wall-s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, synthetic T5 stand-in)",
        "per_rank_s": [round(e / args.steps, 4) for e in per_rank],
        "config": {"workload": f"{args.model} bf16, batch {B} prompts x {args.duration:.0f} s per GPU, CFG, "
                               f"{'greedy' if args.greedy else f'top-k {args.top_k}'}, {T + 3} AR positions + EnCodec-32k decode "
                               f"({config_tag(args)})",
                   "global_batch": B_global, "seq_len": T, "parallelism": f"dp{world} (prompt sharding)"},
    }
    if rank == 0 and args.dump_tokens:
        torch.save(tokens.cpu(), args.dump_tokens)
    if rank == 0:
        lm = model.lm
        n_pos = T + 3
        prefix = int(getattr(lm, '_last_n_prepend', 0))   # prepended condition rows (melody: chroma frames + text), already in the caches
        alg = lm_algorithmic_bytes(lm, 2 * B, n_pos, args.text_len, prefix=prefix)
        job_bytes = alg['total'] * world   # every rank streams its own replica / KV shard
        out["step_roofline"] = {"bound": "hbm", "algorithmic_bytes": job_bytes,
                                "achieved": round(job_bytes * args.steps / elapsed / 1e9, 1),
                                "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                "frac": round(job_bytes * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world), 4),
                                "note": "whole generate incl. EnCodec decode and collectives, all GPUs"}
        if not args.no_roofline:
            r = measure_lin_kernel(model, 2 * B)
            ach = r['bytes_per_launch'] / (r['avg_us'] * 1e-6) / 1e9
            on_cfg2 = config_tag(args).endswith('configs[2]')
            traffic, traffic_src = pmc_traffic_live(args.model, 2 * B) if world == 1 else (None, 'multi-GPU run')
            if traffic is None and on_cfg2:   # fall back to the committed passes (they are of the configs[2] chain), and say why
                why = traffic_src
                traffic, traffic_src = pmc_traffic_per_launch()
                traffic_src = f'{traffic_src}; not measured live: {why}'
            # the roofline fraction is that of the kernels IN SITU (inside the real decode graph, rocprofv3 kernel trace of a
            # short generate of this model); the isolated GEMM-only chain (HIP events) is reported next to it
            is_dur = min(args.duration, 30.0) if args.insitu_duration <= 0 else args.insitu_duration
            insitu, insitu_src = (in_situ_kernel_stats(args.model, B, duration=is_dur, greedy=args.greedy,
                                                       melody_seconds=args.melody_seconds if melody else 0.0)
                                  if world == 1 else (None, 'multi-GPU run'))
            avg_us = insitu['gemm'][1] if insitu is not None else r['avg_us']
            # the traced population against the one the numerator describes: decode positions of the warm-up (2 s) and the main
            # generate x the GEMM launches of a position (a prefix goes through the one-forward prefill: other kernels)
            is_positions = (int(2.0 * model.frame_rate) + 3) + (int(is_dur * model.frame_rate) + 3)
            # the default decode step runs the QKV GEMM inside the fused QKV + self-attention launch (qkv_attn_kernel) where the
            # geometry allows: its launches then leave the GEMM family's population, and its bytes the family's numerator
            fused_on = insitu is not None and 'fused' in insitu
            L_ = lm.num_layers
            w_qkv = (2 if lm.weight_dtype == torch.bfloat16 else 4) * (4 if lm.has_cross_attention else 3) * lm.dim * lm.dim
            lpp = r['launches_per_position'] - (L_ if fused_on else 0)
            gemm_bytes = (r['bytes_per_launch'] * r['launches_per_position'] - (L_ * w_qkv if fused_on else 0)) / lpp
            calls_expected = lpp * is_positions
            ach_is = gemm_bytes / (avg_us * 1e-6) / 1e9
            gemm_obj = {"kernel": "lin_tiled_kernel + lin_pair_kernel (weight-streaming skinny GEMM, LayerNorm folded into its epilogue)", "bound": "hbm",
                               "achieved": round(ach_is, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach_is / HBM_PEAK_GBS, 4), "frac_of_copy_bw": round(ach_is / HBM_COPY_GBS, 4),
                               "traffic": traffic, "traffic_source": traffic_src,
                               "bytes_per_launch": int(gemm_bytes), "avg_launch_us": round(avg_us, 3),
                               "avg_launch_source": ("in situ: " + insitu_src) if insitu is not None else
                                                    f"isolated GEMM chain (HIP events); in-situ pass unavailable: {insitu_src}",
                               "frac_isolated_chain": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us_isolated_chain": round(r['avg_us'], 3),
                               "launches_per_position": lpp}
            if insitu is not None:
                gemm_obj["in_situ_us"] = {k: round(v[1], 3) for k, v in insitu.items()}
                gemm_obj["in_situ_gemm_launches"] = {"traced": insitu['gemm'][0], "expected_decode_launches": calls_expected,
                                                     "note": "numerator and denominator describe the same launches when these agree"}
            if fused_on:
                # the DOMINANT kernel of the decode step (by time): QKV GEMM + self-attention as one launch.  Algorithmic bytes of a
                # launch = the layer's QKV (+ cross-query) weights + the K / V stream of its position (every (row, head) read once)
                gemm_obj["note"] = ("the GEMM launches OUTSIDE the fused QKV + self-attention launch; frac_isolated_chain / traffic are of the "
                                    "isolated chain, which still runs the QKV GEMM as a launch of its own")
                kv = model.lm._run['k']
                Hh, hd_, bk_ = kv.shape[2], kv.shape[4], kv.element_size()
                ctx_f = [prefix + t + 1 for n in (int(2.0 * model.frame_rate) + 3, int(is_dur * model.frame_rate) + 3) for t in range(n)]
                f_bytes = w_qkv + 2 * (2 * B) * Hh * hd_ * bk_ * (sum(ctx_f) / len(ctx_f))
                c_f, us_f = insitu['fused']
                pmc_dur = 3.0   # (rocprofv3 --pmc segfaults on longer generates of this chain: 10 s and 30 s both, profiles/r06_pmc_fused_chain_durations.txt)
                f_traffic, f_src = pmc_traffic_fused(args.model, B, pmc_dur, args.greedy) if (world == 1 and not melody) else (None, 'not collected')
                ctx_p = [t + 1 for n in (int(2.0 * model.frame_rate) + 3, int(pmc_dur * model.frame_rate) + 3) for t in range(n)]
                f_bytes_pmc = w_qkv + 2 * (2 * B) * Hh * hd_ * bk_ * (sum(ctx_p) / len(ctx_p))
                out["roofline"] = {"kernel": "qkv_attn_kernel (the layer's QKV GEMM with the LayerNorm folded in + the self-attention that consumes it, ONE launch: "
                                             "weights and the K / V stream overlap; sentinel hand-off per (row, head))", "bound": "hbm",
                                   "achieved": round(f_bytes / (us_f * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(f_bytes / (us_f * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                   "frac_of_copy_bw": round(f_bytes / (us_f * 1e-6) / 1e9 / HBM_COPY_GBS, 4),
                                   "traffic": f_traffic, "traffic_source": f_src,
                                   "traffic_algorithmic_bytes_of_those_launches": int(f_bytes_pmc),
                                   "traffic_note": ("the counter passes run a 2 s + 3 s generate (mean context ~67: the profiler crashes on longer ones), so `traffic` "
                                                    "compares with traffic_algorithmic_bytes_of_those_launches, not with bytes_per_launch; the difference is a "
                                                    "context-independent overhead (hand-off polls and stores, activation re-reads per XCD)"),
                                   "traffic_overhead_bytes": None if f_traffic is None else int(f_traffic - f_bytes_pmc),
                                   "traffic_over_algorithmic_at_the_bench_context_derived": None if f_traffic is None else
                                       round((f_bytes + (f_traffic - f_bytes_pmc)) / f_bytes, 3),
                                   "bytes_per_launch": int(f_bytes), "avg_launch_us": round(us_f, 3),
                                   "mean_context": round(sum(ctx_f) / len(ctx_f), 1),
                                   "avg_launch_source": "in situ: " + insitu_src,
                                   "launches": {"traced": c_f, "expected": len(ctx_f) * L_},
                                   "share_of_decode_time": "the largest single kernel of the step (DESIGN.md section 5.11)"}
                out["roofline_gemm"] = gemm_obj
            else:
                out["roofline"] = gemm_obj
            ra = measure_attn_kernel(model, 2 * B, T, prefix=prefix)
            out["roofline_attn"] = {"kernel": "attn_decode_kernel (single-query self-attention over the bf16 KV cache)", "bound": "hbm",
                                    "achieved": round(ra['bytes_per_launch'] / (ra['avg_us'] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(ra['bytes_per_launch'] / (ra['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                    "frac_of_copy_bw": round(ra['bytes_per_launch'] / (ra['avg_us'] * 1e-6) / 1e9 / HBM_COPY_GBS, 4),
                                    "bytes_per_launch": int(ra['bytes_per_launch']), "avg_launch_us": round(ra['avg_us'], 3),
                                    "context": ra['context'], "launches": ra['launches'],
                                    "note": "one launch per layer at the mean context of the generate, every layer's own (cold) cache; the stand-alone kernel "
                                            "(the decode step uses it where the fused QKV + self-attention launch does not apply)"}
            out["roofline_attn"].update(attn_traffic_committed(ra['context'], int(ra['bytes_per_launch'])))
            if insitu is not None and 'self_attn' in insitu:
                # in situ: the same kernel inside the decode graph of the traced generates; bytes = the mean K / V stream of
                # those launches (context t + 1 at position t, + the prefix), every (row, head) read once
                kv = model.lm._run['k']
                Hh, hd_, bk_ = kv.shape[2], kv.shape[4], kv.element_size()
                ctxs = [prefix + t + 1 for n in (int(2.0 * model.frame_rate) + 3, int(is_dur * model.frame_rate) + 3) for t in range(n)]
                mean_bytes = 2 * (2 * B) * Hh * hd_ * bk_ * (sum(ctxs) / len(ctxs))
                c_sa, us_sa = insitu['self_attn']
                out["roofline_attn"]["in_situ"] = {
                    "avg_launch_us": round(us_sa, 3), "launches": c_sa, "expected_launches": len(ctxs) * kv.shape[0],
                    "mean_context": round(sum(ctxs) / len(ctxs), 1), "bytes_per_launch": int(mean_bytes),
                    "achieved": round(mean_bytes / (us_sa * 1e-6) / 1e9, 1),
                    "frac": round(mean_bytes / (us_sa * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "rocprofv3 kernel trace of the in-situ child (2 s warm-up + the main generate); frac above is the isolated launch at the mean context"}
        if world == 1 and not args.no_cpu_baseline:
            if melody:
                out["cpu_baseline"] = None   # (the port's timed sample has no prepended prefix; configs[2] carries the CPU line)
            else:
                out["cpu_baseline"] = cpu_baseline(model, B, args.duration, args.text_len, args.top_k)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
