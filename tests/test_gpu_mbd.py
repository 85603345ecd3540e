"""-m gpu: MultiBandDiffusion -- the diffusion kernels through the C ABI, the U-Net, the reverse
process and the sample processor against the CPU oracle (oracle/mbd.py) and the goldens computed by the unmodified reference
(tests/golden/mbd_*.npz, made by tests/golden/make_mbd_golden.py).

Tolerances: all f32; the kernels sum in a different order from torch's CPU kernels (and the convolutions run on
v_mfma_f32_16x16x4_f32), so outputs agree to f32 round-off accumulated over the network depth: 2e-5 absolute / 1e-4 relative on
O(1) outputs at the golden sizes, relative 2-norm error <= 2e-5 at the released geometry.
"""
import json

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mbd as ombd  # noqa: E402

from conftest import load_golden  # noqa: E402


@pytest.fixture(scope='module')
def C():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from audiocraft_amd import _C
    return _C


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------ kernels

@pytest.mark.parametrize('B,Cc,T,G,relu', [(2, 8, 51, 4, True), (3, 48, 8000, 4, True), (1, 3072, 125, 4, False), (2, 16, 1, 4, True),
                                           (1, 192, 2003, 4, True)])
def test_group_norm_vs_torch(C, B, Cc, T, G, relu):
    g = torch.Generator().manual_seed(Cc + T)
    x = torch.randn(B, Cc, T, generator=g) * 2 + 0.5
    w, b = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    ref = F.group_norm(x.double(), G, w.double(), b.double(), 1e-5)
    ref = ref.relu() if relu else ref
    xd = x.cuda()
    got = C.group_norm(xd, w.cuda(), b.cuda(), G, 1e-5, relu=relu)
    assert (got.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    inplace = C.group_norm(xd, w.cuda(), b.cuda(), G, 1e-5, relu=relu, out=xd)
    assert inplace.data_ptr() == xd.data_ptr() and torch.equal(inplace, got)


def test_channel_add_cropped_interp(C):
    g = torch.Generator().manual_seed(3)
    z = torch.randn(3, 10, 77, generator=g)
    table = torch.randn(100, 10, generator=g)
    steps = torch.tensor([5, 99, 0])
    got = C.channel_add(z.cuda().clone(), table.cuda(), steps.cuda()).cpu()
    assert torch.equal(got, z + table[steps][:, :, None])
    a, s = torch.randn(3, 10, 80, generator=g), torch.randn(3, 10, 77, generator=g)
    assert torch.equal(C.add_cropped(a.cuda(), s.cuda()).cpu(), a[..., :77] + s)
    for Tc in (77, 9, 40, 150, 1):     # F.interpolate(mode='nearest'), up- and down-sampling
        ce = torch.randn(3, 10, Tc, generator=g)
        ref = z + F.interpolate(ce, size=77)
        got = C.interp_add(z.cuda().clone(), ce.cuda()).cpu()
        assert torch.equal(got, ref), Tc
    for T, Tc in ((500, 125), (1999, 500), (32000, 1600), (31999, 1500)):
        zz, ce = torch.zeros(1, 1, T), torch.arange(Tc, dtype=torch.float32).view(1, 1, Tc)
        assert torch.equal(C.interp_add(zz.cuda(), ce.cuda()).cpu(), F.interpolate(ce, size=T)), (T, Tc)


def test_ddpm_step_vs_formula(C):
    g = torch.Generator().manual_seed(4)
    cur, est, noise = (torch.randn(2, 1, 1001, generator=g) * 3 for _ in range(3))
    c_est, sa, sigma, clip, es, osc = 0.37, 0.93, 0.21, 5.0, 0.9, 1.5
    ref = (((cur - c_est * (est * es)) / sa + sigma * noise).clamp(-clip, clip)) * osc
    got = C.ddpm_step(cur.cuda(), est.cuda(), noise.cuda(), torch.empty_like(cur).cuda(), c_est, sa, sigma, clip, es, osc).cpu()
    assert torch.allclose(got, ref, atol=1e-6, rtol=1e-6)
    ref0 = (cur - c_est * est) / sa
    got0 = C.ddpm_step(cur.cuda(), est.cuda(), None, torch.empty_like(cur).cuda(), c_est, sa, 0.0, 0.0, 1.0, 1.0).cpu()
    assert torch.allclose(got0, ref0, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize('sr,n_bands,shape', [(16000, 6, (2, 1, 4000)), (24000, 4, (2, 1, 160)), (32000, 32, (1, 1, 9001)),
                                              (24000, 8, (3, 2, 777))])
def test_split_bands_vs_oracle(C, sr, n_bands, shape):
    """acmi_fir_bank / acmi_band_stats / acmi_band_mix against the oracle's julius.SplitBands restatement; the host-built
    filters are the oracle's bit for bit."""
    from audiocraft_amd.modules.diffusion_schedule import SplitBands, band_filters
    g = torch.Generator().manual_seed(n_bands)
    x = torch.randn(*shape, generator=g)
    cut = ombd.mel_frequencies(n_bands + 1, 0, sr / 2)[1:-1] / sr
    filt, half = ombd.lowpass_filters(cut)
    mine, mhalf = band_filters(sr, n_bands)
    assert mhalf == half and torch.equal(mine, filt)
    ref = ombd.split_bands(x, sr, n_bands)
    split = SplitBands(sr, n_bands)
    got = split(x.cuda()).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-5, (got - ref).abs().max()
    st = split.stats(x.cuda(), split.lows(x.cuda()))
    assert st.dtype == torch.float64 and st.shape == (n_bands, 2)
    rd = ref.double().reshape(n_bands, -1)
    assert torch.allclose(st[:, 0], rd.sum(1), atol=2e-3, rtol=1e-4) and torch.allclose(st[:, 1], (rd * rd).sum(1), atol=1e-3, rtol=1e-4)
    gains = torch.rand(n_bands, generator=g) + 0.5
    mix = C.band_mix(x.cuda(), split.lows(x.cuda()), gains.cuda(), offset=0.25).cpu()
    want = (ref * gains.view(-1, 1, 1, 1)).sum(0) + 0.25
    assert (mix - want).abs().max() < 3e-5


# ------------------------------------------------------------------------------------------ the U-Net

def _unet(cfg, sd):
    from audiocraft_amd.models.unet import DiffusionUnet
    kw = {k: cfg[k] for k in ('chin', 'hidden', 'depth', 'growth', 'max_channels', 'num_steps', 'emb_all_layers', 'bilstm', 'codec_dim',
                              'kernel', 'stride', 'norm_groups', 'res_blocks')}
    m = DiffusionUnet(**kw)
    m.load_state_dict(sd, strict=True)      # the reference's own parameter names
    return m.cuda()


@pytest.mark.parametrize('name', ['mbd_unet', 'mbd_unet_bilstm'])
def test_unet_vs_reference_golden(C, name):
    cfg, sd, a = load_golden(name)
    m = _unet(cfg, sd)
    est = m(a['x'].cuda(), a['step'].cuda(), a['condition'].cuda()).sample.cpu()
    assert est.shape == a['estimate'].shape
    assert torch.allclose(est, a['estimate'], atol=2e-5, rtol=1e-4), (est - a['estimate']).abs().max()
    est1 = m(a['x'][:1].cuda(), 42, condition=a['condition'][:1].cuda()).sample.cpu()
    assert torch.allclose(est1, a['estimate_step42'], atol=2e-5, rtol=1e-4)


def _random_unet_sd(uc, seed):
    """Reference-format state dict with O(1) activations throughout (PyTorch default init scaled up for the convolutions)."""
    from audiocraft_amd.models.unet import DiffusionUnet
    torch.manual_seed(seed)
    m = DiffusionUnet(chin=uc.chin, hidden=uc.hidden, depth=uc.depth, growth=uc.growth, max_channels=uc.max_channels,
                      num_steps=uc.num_steps, emb_all_layers=uc.emb_all_layers, bilstm=uc.bilstm, codec_dim=uc.codec_dim,
                      kernel=uc.kernel, stride=uc.stride, norm_groups=uc.norm_groups, res_blocks=uc.res_blocks)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith('norm.weight') or 'norm1.weight' in k or 'norm2.weight' in k:
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        elif 'norm' in k and k.endswith('bias'):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize('bilstm,T,B', [(False, 32000, 2), (False, 24001, 1), (True, 8000, 2)])
def test_unet_released_geometry_vs_oracle(C, bilstm, T, B):
    """config/model/score/basic.yaml (hidden 48, depth 4, kernel 8, stride 4, growth 4, all-layer embeddings) with the codec
    condition of MusicGen's EnCodec (128 x 50 Hz), one second of audio; random weights."""
    uc = ombd.UnetConfig(chin=1, hidden=48, depth=4, growth=4., max_channels=10_000, num_steps=1000, emb_all_layers=True,
                         bilstm=bilstm, codec_dim=128, kernel=8, stride=4, norm_groups=4, res_blocks=1)
    if bilstm:
        uc.depth, uc.growth = 3, 2.      # 192-wide recurrence (the 3072-wide one of depth 4 takes the oracle minutes)
    m, sd = _random_unet_sd(uc, 11)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, 1, T, generator=g)
    cond = torch.randn(B, 128, -(-T // 640), generator=g)
    step = torch.randint(0, 1000, (B,), generator=g)
    ref = ombd.unet_forward(sd, uc, x, step, cond)
    got = m.cuda()(x.cuda(), step.cuda(), cond.cuda()).sample.cpu()
    assert got.shape == ref.shape == (B, 1, T)
    assert rel(got, ref) < 2e-5, rel(got, ref)


def test_unet_refuses_cpu_and_transformer(C):
    from audiocraft_amd.models.unet import DiffusionUnet
    with pytest.raises(NotImplementedError):
        DiffusionUnet(transformer=True)
    m = DiffusionUnet(chin=1, hidden=8, depth=2)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 1, 64), 3)


# ------------------------------------------------------------------------------------------ reverse process + processor

def _schedule_and_processor(cfg, a, device='cuda'):
    from audiocraft_amd.models import builders
    from audiocraft_amd.modules.diffusion_schedule import NoiseSchedule
    pc = dict(cfg['processor'])
    proc = builders.get_processor({'use': True, 'name': 'multi_band_processor', 'n_bands': pc['n_bands'], 'num_samples': pc.get('num_samples', 10),
                                   'power_std': pc['power_std']}, sample_rate=pc['sample_rate'])
    proc.load_state_dict({'counts': a['proc_counts'], 'sum_x': a['proc_sum_x'], 'sum_x2': a['proc_sum_x2'],
                          'sum_target_x2': a['proc_sum_target_x2']})
    proc.to(device)
    return NoiseSchedule(**cfg['schedule'], sample_processor=proc, device=device), proc


def test_reverse_process_vs_reference_golden(C):
    """NoiseSchedule.generate_subsampled + MultiBandProcessor.return_sample / project_sample == the reference, with its
    torch.randn_like draws replayed through `noise_source`."""
    cfg, sd, a = load_golden('mbd_process')
    m = _unet(cfg, sd)
    sched, proc = _schedule_and_processor(cfg, a)
    noises = [n.cuda() for n in a['noises']]
    sched.noise_source = lambda like: noises.pop(0)
    out = sched.generate_subsampled(m, a['initial'].cuda(), step_list=cfg['step_list'], condition=a['condition'].cuda()).cpu()
    assert torch.allclose(out, a['sample'], atol=3e-5, rtol=1e-4), (out - a['sample']).abs().max()
    proj = proc.project_sample(a['sample'].cuda()).cpu()
    assert torch.allclose(proj, a['projected'], atol=3e-5, rtol=1e-4), (proj - a['projected']).abs().max()
    # return_list: every iterate, on the host
    noises.extend(n.cuda() for n in a['noises'])
    its = sched.generate_subsampled(m, a['initial'].cuda(), step_list=cfg['step_list'], condition=a['condition'].cuda(), return_list=True)
    assert len(its) == len(cfg['step_list']) and all(i.shape == a['initial'].shape for i in its)


def test_full_reverse_process_and_training_item_vs_oracle(C):
    """NoiseSchedule.generate (every step of a 12-step schedule, both variance choices) against the same recurrence written
    with the oracle's U-Net; get_training_item's mixing."""
    from audiocraft_amd.modules.diffusion_schedule import NoiseSchedule
    cfg, sd, a = load_golden('mbd_unet')
    cfg = dict(cfg, num_steps=12)
    sd = {k: (v[:12] if k.startswith('embedding') else v) for k, v in sd.items()}
    m = _unet(cfg, sd)
    uc = ombd.UnetConfig(**{k: cfg[k] for k in ('chin', 'hidden', 'depth', 'growth', 'max_channels', 'num_steps', 'emb_all_layers',
                                                'bilstm', 'codec_dim', 'kernel', 'stride', 'norm_groups', 'res_blocks')})
    g = torch.Generator().manual_seed(5)
    init, cond = torch.randn(2, 1, 203, generator=g), a['condition'][:2]
    for variance in ('beta', 'beta_tilde'):
        sched = NoiseSchedule(beta_t0=1e-4, beta_t1=0.3, num_steps=12, variance=variance, clip=3., rescale=0.8, noise_scale=0.9)
        draws = [torch.randn(2, 1, 203, generator=g) for _ in range(11)]
        pending = [d.cuda() for d in draws]
        sched.noise_source = lambda like: pending.pop(0)
        got = sched.generate(m, init.cuda(), condition=cond.cuda()).cpu()
        betas = sched.betas
        alpha_bar, cur = (1 - betas).prod(), init
        for step in range(12)[::-1]:
            est = ombd.unet_forward(sd, uc, cur, step, cond)
            alpha = 1 - betas[step]
            prev = (cur - (1 - alpha) / (1 - alpha_bar).sqrt() * est) / alpha.sqrt()
            pab = (1 - betas[:step]).prod()
            s2 = 0 if step == 0 else (1 - alpha if variance == 'beta' else (1 - pab) / (1 - alpha_bar) * (1 - alpha))
            if s2 > 0:
                prev = prev + s2 ** 0.5 * draws[11 - step] * 0.9
            cur = prev.clamp(-3., 3.)
            alpha_bar = pab
        assert not pending
        assert torch.allclose(got, cur * 0.8, atol=5e-5, rtol=1e-4), (variance, (got - cur * 0.8).abs().max())
    x = torch.randn(2, 1, 203, generator=g)
    nz = torch.randn(2, 1, 203, generator=g)
    sched.noise_source = lambda like: nz.cuda()
    item = sched.get_training_item(x.cuda())
    ab = (1 - sched.betas[:item.step + 1]).prod()
    want = (ab.sqrt() / 0.8) * x + (1 - ab).sqrt() * nz * 0.9
    assert torch.allclose(item.noisy.cpu(), want, atol=1e-5, rtol=1e-5) and torch.equal(item.noise.cpu(), nz)


def test_processor_statistics_update(C):
    """MultiBandProcessor.project_sample while counts < num_samples: the running sums the reference accumulates
    (diffusion_schedule.py:94-99), from acmi_band_stats."""
    from audiocraft_amd.modules.diffusion_schedule import MultiBandProcessor
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 1, 3000, generator=g) * torch.linspace(0.5, 2, 3).view(3, 1, 1) + 0.1
    proc = MultiBandProcessor(n_bands=5, sample_rate=16000, num_samples=100, power_std=[1., 0.9, 0.8, 0.7, 0.6]).cuda()
    out = proc.project_sample(x.cuda()).cpu()
    bands = ombd.split_bands(x, 16000, 5)
    assert proc.counts.item() == 3
    assert torch.allclose(proc.sum_x.cpu(), bands.mean(dim=(2, 3)).sum(dim=1), atol=1e-5)
    assert torch.allclose(proc.sum_x2.cpu(), bands.pow(2).mean(dim=(2, 3)).sum(dim=1), atol=1e-5, rtol=1e-4)
    assert (proc.sum_target_x2 > 0).all()
    ps = ombd.ProcessorState(n_bands=5, sample_rate=16000, power_std=torch.tensor([1., 0.9, 0.8, 0.7, 0.6]), counts=proc.counts.cpu(),
                             sum_x=proc.sum_x.cpu(), sum_x2=proc.sum_x2.cpu(), sum_target_x2=proc.sum_target_x2.cpu())
    assert torch.allclose(out, ombd.project_sample(ps, x), atol=3e-5, rtol=1e-4)
    back = proc.return_sample(out.cuda()).cpu()
    assert torch.allclose(back, ombd.return_sample(ps, out), atol=3e-5, rtol=1e-4)


# ------------------------------------------------------------------------------------------ MultiBandDiffusion

def _make_mbd(codec, n_dp, sample_rate, unet_kw, sched_kw, seed, proc_bands=None):
    """n_dp diffusion processes (random U-Nets of `unet_kw`, loaded processor statistics) around `codec`; also the oracle's view
    of every part: (UnetConfig, state dict, ScheduleConfig, ProcessorState)."""
    from audiocraft_amd.models.multibanddiffusion import DiffusionProcess, MultiBandDiffusion
    from audiocraft_amd.modules.diffusion_schedule import MultiBandProcessor, NoiseSchedule
    dim = codec.decode_latent(torch.zeros(1, codec.num_codebooks, 2, dtype=torch.long, device='cuda')).shape[1]
    nb = proc_bands or n_dp
    DPs, parts = [], []
    for i in range(n_dp):
        uc = ombd.UnetConfig(chin=1, codec_dim=dim, **unet_kw)
        m, sd = _random_unet_sd(uc, seed + i)
        proc = MultiBandProcessor(n_bands=nb, sample_rate=sample_rate, num_samples=1)
        g = torch.Generator().manual_seed(seed + 10 + i)
        st = {'counts': torch.tensor([4.]), 'sum_x': torch.randn(nb, generator=g) * 0.01,
              'sum_x2': torch.rand(nb, generator=g) + 0.5, 'sum_target_x2': torch.rand(nb, generator=g) + 0.5}
        proc.load_state_dict(st)
        DPs.append(DiffusionProcess(m.cuda(), NoiseSchedule(**sched_kw, sample_processor=proc.cuda())))
        parts.append((uc, sd, ombd.ScheduleConfig(**sched_kw), ombd.ProcessorState(n_bands=nb, sample_rate=sample_rate, power_std=1., **st)))
    return MultiBandDiffusion(DPs, codec), parts


def _debug_mbd(n_dp=2, seed=21):
    from audiocraft_amd.models import builders
    codec = builders.get_debug_compression_model('cuda', sample_rate=16000)
    unet_kw = dict(hidden=8, depth=2, growth=2., max_channels=10_000, num_steps=40, emb_all_layers=True, bilstm=False, kernel=8,
                   stride=4, norm_groups=4, res_blocks=1)
    sched_kw = dict(beta_t0=1e-4, beta_t1=0.1, num_steps=40, variance='beta', clip=5., rescale=1., noise_scale=1.0)
    return _make_mbd(codec, n_dp, 16000, unet_kw, sched_kw, seed)


def test_multibanddiffusion_released_shape_tokens_to_wav_vs_oracle(C):
    """The whole token -> waveform path at the RELEASED shape (config/model/score/basic.yaml, config/solver/diffusion/
    default.yaml, MultiBandDiffusion.get_mbd_musicgen): the 32 kHz EnCodec geometry (128-d latents at 50 Hz), four bands,
    U-Nets of hidden 48 / depth 4 / growth 4 / kernel 8 / stride 4, the 1000-step schedule (beta 1e-5 .. 2.9e-2, exponent
    7.5) sub-sampled to six steps, 8-band sample processors, then the 32-band re_eq against the codec's own decode -- on
    1 s of audio, initial noise and every step's noise replayed, against oracle/mbd.py (multibanddiffusion.py:112-191,
    diffusion_schedule.py:205-272).  Random weights: what is checked is the arithmetic, end to end, at the size that ships."""
    from audiocraft_amd.models import builders
    torch.manual_seed(5)
    codec = builders.get_compression_model(builders.ENCODEC_32KHZ, 'cuda')
    unet_kw = dict(hidden=48, depth=4, growth=4., max_channels=10_000, num_steps=1000, emb_all_layers=True, bilstm=False, kernel=8,
                   stride=4, norm_groups=4, res_blocks=1)
    sched_kw = dict(beta_t0=1e-5, beta_t1=2.9e-2, beta_exp=7.5, num_steps=1000, variance='beta', clip=5., rescale=1., noise_scale=1.0)
    mbd, parts = _make_mbd(codec, 4, 32000, unet_kw, sched_kw, seed=40, proc_bands=8)
    g = torch.Generator().manual_seed(6)
    tokens = torch.randint(0, codec.cardinality, (1, codec.num_codebooks, 50), generator=g).cuda()     # 1 s at 50 Hz
    wav_codec = codec.decode(tokens)
    assert wav_codec.shape == (1, 1, 32000)
    emb = mbd.get_emb(tokens)
    assert emb.shape == (1, 128, 50)
    steps = [999, 799, 599, 399, 199, 0]
    size = wav_codec.shape
    inits = [torch.randn(*size, generator=g) for _ in parts]
    draws = [[torch.randn(*size, generator=g) for _ in steps[:-2]] for _ in parts]
    pend_init = [t.cuda() for t in inits]
    mbd.noise_source = lambda like: pend_init.pop(0)
    for dp, dr in zip(mbd.DPs, draws):
        pend = [t.cuda() for t in dr]
        dp.schedule.noise_source = (lambda p: (lambda like: p.pop(0)))(pend)
    wav = mbd.generate(emb, size=size, step_list=steps)
    ref = torch.zeros(size)
    for (uc, sd, sc, ps), init, dr in zip(parts, inits, draws):
        model = (lambda sd, uc: (lambda x, step, cond: ombd.unet_forward(sd, uc, x, step, cond)))(sd, uc)
        ref = ref + ombd.generate_subsampled(model, sc, init, steps, emb.cpu(), list(dr), ps)
    r = rel(wav.cpu(), ref)
    assert wav.shape == size and r < 5e-5, f"generate at the released shape: rel-L2 {r}, max abs {(wav.cpu() - ref).abs().max()}"
    eq = mbd.re_eq(wav, wav_codec, n_bands=32).cpu()
    want = ombd.re_eq(ref, wav_codec.cpu(), 32000, n_bands=32)
    assert rel(eq, want) < 1e-4, rel(eq, want)
    # tokens_to_wav = generate + re_eq with fresh noise: shape and finiteness (its arithmetic is the two calls checked above)
    mbd.noise_source = None
    for dp in mbd.DPs:
        dp.schedule.noise_source = None
    out = mbd.tokens_to_wav(tokens, n_bands=32)
    assert out.shape == size and torch.isfinite(out).all()


def test_multibanddiffusion_tokens_to_wav_vs_oracle(C):
    """generate (sum over the bands' processes) + re_eq against the oracle, draws replayed; get_condition / regenerate shapes."""
    mbd, parts = _debug_mbd()
    codec = mbd.codec_model
    g = torch.Generator().manual_seed(2)
    tokens = torch.randint(0, codec.cardinality, (2, codec.num_codebooks, 12), generator=g).cuda()
    wav_codec = codec.decode(tokens)
    emb = mbd.get_emb(tokens)
    steps = [39, 26, 13, 0]
    size = wav_codec.shape
    inits = [torch.randn(*size, generator=g) for _ in parts]
    draws = [[torch.randn(*size, generator=g) for _ in steps[:-2]] for _ in parts]
    pend_init = [t.cuda() for t in inits]
    mbd.noise_source = lambda like: pend_init.pop(0)
    for dp, dr in zip(mbd.DPs, draws):
        pend = [t.cuda() for t in dr]
        dp.schedule.noise_source = (lambda p: (lambda like: p.pop(0)))(pend)
    wav = mbd.generate(emb, size=size, step_list=steps)
    ref = torch.zeros(size)
    for (uc, sd, sc, ps), init, dr in zip(parts, inits, draws):
        model = (lambda sd, uc: (lambda x, step, cond: ombd.unet_forward(sd, uc, x, step, cond)))(sd, uc)
        ref = ref + ombd.generate_subsampled(model, sc, init, steps, emb.cpu(), list(dr), ps)
    assert wav.shape == size and torch.allclose(wav.cpu(), ref, atol=5e-5, rtol=1e-4), (wav.cpu() - ref).abs().max()
    eq = mbd.re_eq(wav, wav_codec, n_bands=8).cpu()
    want = ombd.re_eq(ref, wav_codec.cpu(), 16000, n_bands=8)
    assert rel(eq, want) < 1e-4, rel(eq, want)
    # the band standard deviations of the result are the codec output's
    sb, sr_ = ombd.split_bands(eq, 16000, 8), ombd.split_bands(wav_codec.cpu(), 16000, 8)
    for i in range(8):
        assert abs(sb[i].std() / sr_[i].std() - 1) < 0.05
    mbd.noise_source = None
    for dp in mbd.DPs:
        dp.schedule.noise_source = None
    x = torch.randn(1, 1, 8000, generator=g)
    cond = mbd.get_condition(x, 8000)
    assert cond.shape[0] == 1 and cond.shape[1] == emb.shape[1]
    out = mbd.generate(cond, step_list=steps)
    assert out.shape[0] == 1 and out.shape[-1] == cond.shape[-1] * int(codec.sample_rate / codec.frame_rate) and torch.isfinite(out).all()


def test_multibanddiffusion_vs_reference_golden(C):
    """MultiBandDiffusion.generate + re_eq against the fixture made by the reference's own multibanddiffusion.py
    (tests/golden/mbd_model.npz): two bands, initial and step noise replayed through the `noise_source` hooks."""
    from audiocraft_amd.models.multibanddiffusion import DiffusionProcess, MultiBandDiffusion
    from audiocraft_amd.modules.diffusion_schedule import MultiBandProcessor, NoiseSchedule
    cfg, sd, a = load_golden('mbd_model')
    n, pc = cfg['draws_per_band'], cfg['processor']

    class Codec(torch.nn.Module):   # what MultiBandDiffusion reads from the compression model
        sample_rate, frame_rate, channels = 16000, 50, 1

        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

    DPs, inits = [], []
    for i in range(2):
        m = _unet(cfg, {k[len(f'dp{i}.'):]: v for k, v in sd.items() if k.startswith(f'dp{i}.')})
        proc = MultiBandProcessor(n_bands=pc['n_bands'], sample_rate=pc['sample_rate'], num_samples=1, power_std=pc['power_std'])
        proc.load_state_dict({k: a[f'proc{i}_{k}'] for k in ('counts', 'sum_x', 'sum_x2', 'sum_target_x2')})
        sched = NoiseSchedule(**cfg['schedule'], sample_processor=proc.cuda())
        d = [t.cuda() for t in a['draws'][i * n:(i + 1) * n]]
        inits.append(d[0])
        sched.noise_source = (lambda p: (lambda like: p.pop(0)))(d[1:])
        DPs.append(DiffusionProcess(m, sched))
    mbd = MultiBandDiffusion(DPs, Codec().cuda())
    mbd.noise_source = lambda like: inits.pop(0)
    wav = mbd.generate(a['emb'].cuda(), step_list=cfg['step_list'])
    assert wav.shape == a['generated'].shape
    assert torch.allclose(wav.cpu(), a['generated'], atol=5e-5, rtol=1e-4), (wav.cpu() - a['generated']).abs().max()
    eq = mbd.re_eq(a['generated'].cuda(), a['reference_wav'].cuda(), n_bands=8).cpu()
    assert rel(eq, a['re_eq']) < 1e-5, rel(eq, a['re_eq'])
    eq_half = mbd.re_eq(a['generated'].cuda(), a['reference_wav'].cuda(), n_bands=8, strictness=0.5).cpu()
    assert rel(eq_half, a['re_eq_half']) < 1e-5


def test_load_diffusion_models_roundtrip(C, tmp_path):
    """loaders.load_diffusion_models on a package in the release layout ({'sample_rate', 'n_bands', i: {cfg, model_state,
    processor_state}}) written from reference-format state dicts."""
    from audiocraft_amd.models import loaders
    cfg, sd, a = load_golden('mbd_process')
    pc = cfg['processor']
    xp = {'channels': 1, 'schedule': cfg['schedule'],
          'diffusion_unet': {k: cfg[k] for k in ('hidden', 'depth', 'growth', 'max_channels', 'emb_all_layers', 'bilstm', 'codec_dim',
                                                 'kernel', 'stride', 'norm_groups', 'res_blocks')},
          'processor': {'use': True, 'name': 'multi_band_processor', 'n_bands': pc['n_bands'], 'num_samples': pc.get('num_samples', 10),
                        'power_std': pc['power_std']}}
    pstate = {'counts': a['proc_counts'], 'sum_x': a['proc_sum_x'], 'sum_x2': a['proc_sum_x2'], 'sum_target_x2': a['proc_sum_target_x2']}
    # the release layout also carries julius' low-pass bank (a buffer of the reference's SplitBands)
    from audiocraft_amd.modules.diffusion_schedule import band_filters
    pstate['split_bands.lowpass.filters'] = band_filters(pc['sample_rate'], pc['n_bands'])[0][:, None, :]
    pkg = {'sample_rate': pc['sample_rate'], 'n_bands': 2,
           0: {'cfg': xp, 'model_state': sd, 'processor_state': pstate}, 1: {'cfg': json.dumps(xp), 'model_state': sd, 'processor_state': pstate}}
    path = tmp_path / 'mbd_test.th'
    torch.save(pkg, path)
    models, processors, cfgs = loaders.load_diffusion_models(str(path), device='cuda')
    assert len(models) == len(processors) == len(cfgs) == 2
    from audiocraft_amd.modules.diffusion_schedule import NoiseSchedule
    sched = NoiseSchedule(**cfgs[1]['schedule'], sample_processor=processors[1])
    noises = [n.cuda() for n in a['noises']]
    sched.noise_source = lambda like: noises.pop(0)
    out = sched.generate_subsampled(models[1], a['initial'].cuda(), step_list=cfg['step_list'], condition=a['condition'].cuda()).cpu()
    assert torch.allclose(out, a['sample'], atol=3e-5, rtol=1e-4)
